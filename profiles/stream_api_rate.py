#!/usr/bin/env python3
"""Throughput of the drop-in boundary (chain_stream_gpu, deferred hand-back) driven without a minimap2 host: batches of
chain_read_t records of a given size, one stream.  What a batch costs end to end (pack, H2D, kernels, D2H, host post-pass,
results into malloc'd arrays) and where the time goes.   python profiles/stream_api_rate.py [--reads-per-batch 64 ...]"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mm2gb_amd as mm
from test_gpu_stream_api import ChainRead, libc

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stream_api_rate.json"))
ap.add_argument("--threads", type=int, default=0, help="also drive this many streams at once (host threads, one stream id each; needs num_streams >= threads in the config)")
ap.add_argument("--leave-queues-to-the-library", action="store_true", help="do not export GPU_MAX_HW_QUEUES here: init_stream_gpu sets it (if the HIP runtime has not started)")
args = ap.parse_args()
if args.threads > 1 and not args.leave_queues_to_the_library:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(4 * args.threads + 2))      # four HIP streams per stream id; before the runtime starts
L = mm.lib()
L.init_stream_gpu.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, mm.Misc]
L.chain_stream_gpu.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_void_p]
L.finish_stream_gpu.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_void_p]
L.free_stream_gpu.argtypes = [C.c_int]
# The library frees every read's input anchors (as compact_a does, lchain.c:108-109) with the host's allocator -- here glibc's.
# Keep freed memory in the process (no trimming, no mmap per array): returning a GB to the kernel at the last free of a pass
# cost 0.18 s and is this harness's allocator, not the path measured (the reference host recycles a kalloc arena).
libc.mallopt.argtypes = [C.c_int, C.c_int]
libc.mallopt(-1, 2**31 - 1)      # M_TRIM_THRESHOLD
libc.mallopt(-3, 2**30)          # M_MMAP_THRESHOLD
mt, mr, mn = C.c_size_t(0), C.c_int(0), C.c_int(0)
cfg = json.load(open(os.path.join(ROOT, "mm2-gb_amd", "mi355x_config.json")))
cfg["num_streams"] = max(1, args.threads)
cfg_path = os.path.join(ROOT, "gpurun_out", "stream_api_rate_cfg.json")
os.makedirs(os.path.dirname(cfg_path), exist_ok=True)
json.dump(cfg, open(cfg_path, "w"))
L.init_stream_gpu(C.byref(mt), C.byref(mr), C.byref(mn), cfg_path.encode(), mm.default_misc())


def release(ptr, n):
    arr = C.cast(ptr, C.POINTER(ChainRead))
    chains = 0
    for k in range(n):
        r = arr[k]
        chains += r.n_u
        if r.n_u > 0:
            libc.free(r.u); libc.free(r.a)
    return chains


rows = []
for reads_per_batch, lo, hi, n_batches in ((64, 10_000, 100_000, 24), (512, 10_000, 100_000, 8), (4096, 10_000, 100_000, 3), (64, 100_000, 300_000, 12)):
    anchors, off = mm.synth_reads(77, 0, reads_per_batch * n_batches, lo, hi, threads=16)
    batches = []
    for bi in range(n_batches):
        arr = (ChainRead * reads_per_batch)()
        for k in range(reads_per_batch):
            r = bi * reads_per_batch + k
            a = anchors[off[r]:off[r + 1]]
            buf = libc.malloc(max(a.nbytes, 16))
            C.memmove(buf, a.ctypes.data, a.nbytes)
            arr[k].a, arr[k].n, arr[k].n_seg = buf, len(a), 1
        batches.append(arr)
    n_anch = int(off[-1])

    def fresh_batches():
        out = []
        for bi in range(n_batches):
            arr = (ChainRead * reads_per_batch)()
            for k in range(reads_per_batch):
                r = bi * reads_per_batch + k
                a = anchors[off[r]:off[r + 1]]
                buf = libc.malloc(max(a.nbytes, 16))
                C.memmove(buf, a.ctypes.data, a.nbytes)
                arr[k].a, arr[k].n, arr[k].n_seg = buf, len(a), 1
            out.append(arr)
        return out

    # two warm-up passes: the two staging sets of the stream alternate, both must have seen the largest batch (page-locking a
    # grown buffer is a one-time cost of ~0.4 s per GB); then the timed pass on fresh copies
    for rep in range(2):
        for arr in batches:
            ptr, n = C.c_void_p(C.addressof(arr)), C.c_int(reads_per_batch)
            L.chain_stream_gpu(None, None, C.byref(ptr), C.byref(n), 0, None)
            if ptr.value:
                release(ptr.value, n.value)
        ptr, n = C.c_void_p(0), C.c_int(0)
        L.finish_stream_gpu(None, None, C.byref(ptr), C.byref(n), 0, None)
        release(ptr.value, n.value)
        batches = fresh_batches()
    t0 = time.perf_counter()
    chains = 0
    for arr in batches:
        ptr, n = C.c_void_p(C.addressof(arr)), C.c_int(reads_per_batch)
        L.chain_stream_gpu(None, None, C.byref(ptr), C.byref(n), 0, None)
        if ptr.value:
            chains += release(ptr.value, n.value)
    ptr, n = C.c_void_p(0), C.c_int(0)
    L.finish_stream_gpu(None, None, C.byref(ptr), C.byref(n), 0, None)
    chains += release(ptr.value, n.value)
    dt = time.perf_counter() - t0
    rows.append({"reads_per_batch": reads_per_batch, "read_len": [lo, hi], "batches": n_batches, "anchors": n_anch, "anchors_per_batch": n_anch // n_batches,
                 "seconds": round(dt, 4), "ms_per_batch": round(dt * 1e3 / n_batches, 2), "anchors_per_s": n_anch / dt, "chains": chains})
    print(rows[-1], flush=True)
mt_rows = []
if args.threads > 1:
    import threading
    reads_per_batch, lo, hi, n_batches = 64, 10_000, 100_000, 16
    work = []
    for t in range(args.threads):
        anchors, off = mm.synth_reads(100 + t, 0, reads_per_batch * n_batches, lo, hi, threads=16)
        work.append((anchors, off))

    def make(anchors, off):
        out = []
        for bi in range(n_batches):
            arr = (ChainRead * reads_per_batch)()
            for k in range(reads_per_batch):
                r = bi * reads_per_batch + k
                a = anchors[off[r]:off[r + 1]]
                buf = libc.malloc(max(a.nbytes, 16))
                C.memmove(buf, a.ctypes.data, a.nbytes)
                arr[k].a, arr[k].n, arr[k].n_seg = buf, len(a), 1
            out.append(arr)
        return out

    def drive(tid, batches, res):
        chains = 0
        for arr in batches:
            ptr, n = C.c_void_p(C.addressof(arr)), C.c_int(reads_per_batch)
            L.chain_stream_gpu(None, None, C.byref(ptr), C.byref(n), tid, None)
            if ptr.value:
                chains += release(ptr.value, n.value)
        ptr, n = C.c_void_p(0), C.c_int(0)
        L.finish_stream_gpu(None, None, C.byref(ptr), C.byref(n), tid, None)
        chains += release(ptr.value, n.value)
        res[tid] = chains

    for n_thr in (1, args.threads):
        for rep in range(2):                       # first pass warms the staging buffers of every stream
            sets = [make(*work[t]) for t in range(n_thr)]
            res = [0] * n_thr
            t0 = time.perf_counter()
            ths = [threading.Thread(target=drive, args=(t, sets[t], res)) for t in range(n_thr)]
            [t.start() for t in ths]; [t.join() for t in ths]
            dt = time.perf_counter() - t0
        n_anch = sum(int(work[t][1][-1]) for t in range(n_thr))
        mt_rows.append({"streams": n_thr, "reads_per_batch": reads_per_batch, "batches_per_stream": n_batches, "anchors": n_anch,
                        "seconds": round(dt, 4), "anchors_per_s": n_anch / dt, "chains": sum(res)})
        print(mt_rows[-1], flush=True)
L.free_stream_gpu(1)
json.dump({"post_threads": os.environ.get("MM2GB_POST_THREADS", "default"), "rows": rows, "several_streams": mt_rows}, open(args.out, "w"), indent=1)
