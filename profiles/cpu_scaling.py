import os, sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, mm2gb_amd as mm, orc
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: pass
a, off = mm.synth_reads(2024, 0, 256, 100_000, 300_000, threads=32)
prm = orc.default_param()
for th in (1, 2, 4, 8, 16, 32, 64, 128):
    nr = min(256, max(4, th * 2))
    t0 = time.perf_counter()
    _, _, pairs = orc.chain_fill_many(a[: off[nr]], off[: nr + 1], prm, threads=th)
    dt = time.perf_counter() - t0
    print(th, "threads", nr, "reads", round(pairs / dt / 1e9, 3), "G pairs/s", round(dt, 2), "s", flush=True)
