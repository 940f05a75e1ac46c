"""Import shim: the package directory is `mm2-gb_amd/` (hyphenated, as the project is named), which the import
statement cannot spell.  `import mm2gb_amd` gives the same module object."""
import importlib
import sys

_pkg = importlib.import_module("mm2-gb_amd")
sys.modules[__name__] = _pkg
