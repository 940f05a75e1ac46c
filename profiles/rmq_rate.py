#!/usr/bin/env python3
"""Throughput of RMQ re-chaining: mm2gb_rmq_chain_gpu on a batch of reads vs the reference's own mg_lchain_rmq
(oracle/_ref/libmm2ref.so, one read per call on all usable host threads) on the same inputs -- the anchors the first chaining
keeps, re-sorted by x.   python profiles/rmq_rate.py [--reads 2000] [--out gpurun_out/rmq_rate.json]"""
import argparse, ctypes as C, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, mm2gb_amd as mm, orc

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=2000)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rmq_rate.json"))
args = ap.parse_args()
threads = bench.cpu_quota() or os.cpu_count() or 8
a, off = mm.synth_reads(2024, 0, args.reads, 100_000, 300_000, threads=threads)
with mm.Engine() as e:
    t0 = time.perf_counter(); first, st1 = e.chain(a, off, threads=threads); t_first = time.perf_counter() - t0
    reads = [orc.radix_sort_x(x[1]) if len(x[1]) else x[1] for x in first]          # what post_chaining_helper re-sorts and passes on (map.c:449)
    o2 = np.zeros(len(reads) + 1, np.int64); o2[1:] = np.cumsum([len(x) for x in reads])
    allr = np.concatenate(reads)
    prm = mm.default_rmq_param()
    best = best_c = best_xc = None
    for _ in range(3):
        t0 = time.perf_counter(); res, tied, st = e.rmq_chain(allr, o2, prm); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        best_c = st["ms_total"] / 1e3 if best_c is None else min(best_c, st["ms_total"] / 1e3)      # the C call alone (its own clock): without the wrapper's per-read numpy copies
    # the call the product answers with: device and host threads at the same time, exact for every read (csrc/rmq_hybrid.cpp)
    best_x, deal = None, {}
    for _ in range(3):
        t0 = time.perf_counter(); _, where, deal = e.rmq_chain_exact(allr, o2, prm, threads=threads); dt = time.perf_counter() - t0
        best_x = dt if best_x is None else min(best_x, dt)
        best_xc = deal["total_s"] if best_xc is None else min(best_xc, deal["total_s"])
    # and the host form alone on the same threads
    t0 = time.perf_counter(); mm.rmq_chain_host(allr, o2, prm, threads=threads); t_host_form = time.perf_counter() - t0
doc = {"exact_call_s": round(best_x, 4), "exact_call_anchors_per_s": o2[-1] / best_x, "exact_call_deal": deal, "host_form_s": round(t_host_form, 4), "host_threads": threads,
       "reads": args.reads, "anchors_first_pass": int(off[-1]), "anchors_rechained": int(o2[-1]), "first_pass_chain_host_s": round(t_first, 3),
       "gpu_rmq_call_s": round(best_c, 4), "exact_call_c_s": round(best_xc, 4),
       "gpu_rmq_chain_s": round(best, 4), "gpu_anchors_per_s": o2[-1] / best, "reads_with_a_tie": int((tied > 0).sum()), "ms_post": st["ms_post"]}
ref_path = os.path.join(orc.REF_DIR, "libmm2ref.so")
if os.path.exists(ref_path):
    ref = C.CDLL(ref_path)
    ref.mg_lchain_rmq.restype = C.c_void_p
    ref.mg_lchain_rmq.argtypes = [C.c_int] * 7 + [C.c_float, C.c_float, C.c_int64, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_void_p]
    libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]; libc.free.argtypes = [C.c_void_p]
    def one(r):
        x = reads[r]
        if not len(x): return
        buf = libc.malloc(x.nbytes); C.memmove(buf, x.ctypes.data, x.nbytes)
        n_u, u = C.c_int(0), C.c_void_p(0)
        out = ref.mg_lchain_rmq(prm.max_dist, prm.max_dist_inner, prm.bw, prm.max_chn_skip, prm.cap_rmq_size, prm.min_cnt, prm.min_sc, prm.chn_pen_gap, prm.chn_pen_skip,
                                len(x), buf, C.byref(n_u), C.byref(u), None)
        if out: libc.free(out)
        if u.value: libc.free(u)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex: list(ex.map(one, range(len(reads))))
    dt = time.perf_counter() - t0
    doc.update({"reference_cpu_s": round(dt, 3), "reference_cpu_threads": threads, "reference_anchors_per_s": o2[-1] / dt, "gpu_over_reference": dt / best, "exact_call_over_reference": dt / best_x,
                "gpu_call_over_reference": dt / best_c, "exact_call_c_over_reference": dt / best_xc,
                "note": "gpu_rmq_chain_s / exact_call_s are timed around the Python wrappers, which copy every read's result into numpy arrays; *_call_s / *_c_s are the C calls alone"})
json.dump(doc, open(args.out, "w"), indent=1)
print(json.dumps(doc))
