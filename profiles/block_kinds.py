#!/usr/bin/env python3
"""How the score kernel's tile-blocks (64 sources x 64 targets) split over its sweep builds, on the bench's synthetic reads: needs the
instrumented library (variants/libcount.so = chain_kernels.hip + profiles/experiments/r02x_block_kinds_instrumentation.patch).  Prints JSON."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MM2GB_LIB_PATH"] = os.path.join(ROOT, "mm2-gb_amd", "variants", "libcount.so")
sys.path.insert(0, ROOT)
import numpy as np
import mm2gb_amd as mm
out = {}
for name, lo, hi, n_reads in (("100-300kb", 100_000, 300_000, 1800), ("10-100kb", 10_000, 100_000, 6000)):
    a, off = mm.synth_reads(2024, 0, n_reads, lo, hi, threads=16)
    cnt = (C.c_ulonglong * 8)()
    with mm.Engine() as e:
        e.set_misc(mm.default_misc())
        mm.lib().mm2gb_debug_block_counts(cnt, 1)
        f, p, st = e.score(a, off)
        mm.lib().mm2gb_debug_block_counts(cnt, 1)
    c = list(cnt)
    names = ("two_tiles_unchecked", "two_tiles_range_test", "one_tile_unchecked", "one_tile_range_test", "one_tile_window_tests", "in_tile_phases")
    tot = sum(c[:6])
    out[name] = {"anchors": int(len(a)), "pairs": int(st["n_pairs"]), "tile_blocks": dict(zip(names, c[:6])), "share": {k: round(v / tot, 4) for k, v in zip(names, c[:6])}}
print(json.dumps(out))
