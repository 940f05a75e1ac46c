#!/usr/bin/env python3
"""Counts of the in-tile phases on the bench's reads (instrumented library of make_intile_count_build.py)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MM2GB_LIB_PATH"] = os.path.join(ROOT, "mm2-gb_amd", "ab", "libintile.so")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ROOT)
import mm2gb_amd as mm
names = ["in_tile_phases", "plain_steps_calls", "sources_needed", "rows_computed", "calls_free", "calls_checked", "calls_clamped", "fast_path_undone", "entry_or_full_steps",
         "phases_rescue_build", "phases_nothing_to_do"]
for name, lo, hi, n_reads in (("100-300kb", 100_000, 300_000, 1800), ("10-30kb", 10_000, 30_000, 20000)):
    a, off = mm.synth_reads(2024, 0, n_reads, lo, hi, threads=16)
    cnt = (C.c_ulonglong * 16)()
    with mm.Engine() as e:
        e.set_misc(mm.default_misc())
        mm.lib().mm2gb_debug_intile_counts(cnt, 1)
        f, p, st = e.score(a, off)
        mm.lib().mm2gb_debug_intile_counts(cnt, 1)
    d = dict(zip(names, list(cnt)))
    d.update(anchors=int(len(a)), tiles=int(len(a)) // 64, pairs=int(st["n_pairs"]))
    print(name, json.dumps(d))
