"""One mm2gb_chain_gpu call (page-locked host anchors -> chains, everything on the device) at the host legs' size, alone, for a timeline:
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/hp_trace -- python3 profiles/host_path_timeline.py 200000000
prints the call's seconds (three calls: the first makes arenas and result blocks) and, with MM2GB_DEBUG_PHASES=1, the library's own line."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
import mm2gb_amd as mm
import bench

n_target = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
nr = bench.pick_reads(mm, 2024, 0, n_target, 100_000, 300_000)
a, off = mm.synth_reads(2024, 0, nr, 100_000, 300_000, threads=16)
n = int(off[-1])
h_a = torch.empty((n, 2), dtype=torch.int64).pin_memory()
h_a.numpy()[:] = a.view(np.int64)
del a
eng = mm.Engine(device=0)
for k in range(3):
    ch, st = mm.Chains(), mm.Stats()
    t0 = time.perf_counter()
    rc = mm.lib().mm2gb_chain_gpu(eng._h, nr, off.ctypes.data, h_a.data_ptr(), ctypes.byref(ch), ctypes.byref(st))
    dt = time.perf_counter() - t0
    assert rc == 0, mm.lib().mm2gb_last_error().decode()
    print(f"call {k}: {n} anchors, {nr} reads: {dt * 1e3:.1f} ms; H2D alone at 57.6 GB/s would be {n * 16 / 57.6e9 * 1e3:.1f} ms", flush=True)
    mm.lib().mm2gb_chains_free(ctypes.byref(ch))
eng.close()
