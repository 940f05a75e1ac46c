#!/usr/bin/env python3
"""How the score kernel's tile-blocks (64 sources x 64 targets) split over its sweep builds, on the bench's synthetic reads, and what that
makes in vector instructions per 64 pairs: needs the instrumented library (python profiles/experiments/make_block_kinds_build.py ->
mm2-gb_amd/ab/libcount.so).  Prints JSON."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MM2GB_LIB_PATH"] = os.path.join(ROOT, "mm2-gb_amd", "ab", "libcount.so")
sys.path.insert(0, ROOT)
import numpy as np
import mm2gb_amd as mm
out = {}
for name, lo, hi, n_reads in (("100-300kb", 100_000, 300_000, 1800), ("10-100kb", 10_000, 100_000, 6000)):
    a, off = mm.synth_reads(2024, 0, n_reads, lo, hi, threads=16)
    cnt = (C.c_ulonglong * 16)()
    with mm.Engine() as e:
        e.set_misc(mm.default_misc())
        mm.lib().mm2gb_debug_block_counts(cnt, 1)
        f, p, st = e.score(a, off)
        mm.lib().mm2gb_debug_block_counts(cnt, 1)
    c = list(cnt)
    names = ("two_tiles_far", "two_tiles_unchecked", "two_tiles_range_test", "one_tile_far", "one_tile_unchecked", "one_tile_range_test", "one_tile_window_tests", "in_tile_phases")
    # vector instructions per source step (64 pairs) of each build, from the sweeps' source: FAR sad + add + half a max3; unchecked 2 sub,
    # sad, min3, shift-add, add, half a max3; range test + cmpx and a whole max; window tests: four compares and a select more; in-tile ~13 per step
    per_step = (2.5, 6.5, 8.0, 2.5, 6.5, 8.0, 11.0, 13.0)
    tot = sum(c[:8])
    budget = sum(v * k for v, k in zip(c[:8], per_step)) / tot
    out[name] = {"anchors": int(len(a)), "pairs": int(st["n_pairs"]), "tile_blocks": dict(zip(names, c[:8])), "share": {k: round(v / tot, 4) for k, v in zip(names, c[:8])},
                 "valu_instructions_per_64_pairs_by_kind": {k: round(v * q / tot, 3) for k, v, q in zip(names, c[:8], per_step)},
                 "valu_instructions_per_64_pairs_from_shares": round(budget, 2),
                 "rescue_build_in_tile": {"tiles": c[11], "steps_entry_mode": c[8], "steps_after_an_update_inside_the_tile": c[9], "steps_full_state_machine": c[10],
                                          "share_of_all_in_tile_phases": round(c[11] / max(1, c[7]), 3)},
                 "note": "measured overall: roofline.valu_insts_per_64_pairs on the bench line (counters of the shipped build); the difference is block staging, loop control in vector registers and lanes of partly filled tiles"}
print(json.dumps(out))
