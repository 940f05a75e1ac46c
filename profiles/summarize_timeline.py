"""rocprofv3 --kernel-trace --memory-copy-trace CSVs of profiles/host_path_timeline.py -> a text timeline of its LAST call: every copy and
kernel with start / end in ms relative to the call's first event, busy time per engine.   python3 profiles/summarize_timeline.py gpurun_out/hp_trace"""
import csv, glob, os, sys
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel", r["Kernel_Name"].split("(")[0][:48]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "copy"), ""))
ev.sort()
# the last call: events after the last gap of more than 30 ms
cut = 0
for k in range(1, len(ev)):
    if ev[k][0] - max(e[1] for e in ev[max(0, k - 50):k]) > 30e6:
        cut = k
ev = ev[cut:]
t0 = ev[0][0]
busy = {}
print(f"{len(ev)} events; times in ms since the call's first event")
merged = []
for s, e, kind, name in ev:
    key = kind if kind != "kernel" else name
    if merged and merged[-1][2] == key and s - merged[-1][1] < 0.2e6:
        merged[-1][1] = e; merged[-1][3] += 1; merged[-1][4] += e - s
    else:
        merged.append([s, e, key, 1, e - s])
    busy[key] = busy.get(key, 0) + e - s
for s, e, key, cnt, b in merged:
    if e - s > 0.3e6:
        print(f"{(s - t0) / 1e6:8.2f} -> {(e - t0) / 1e6:8.2f}  {key}" + (f"  x{cnt}" if cnt > 1 else ""))
print("busy ms by kind:", {k: round(v / 1e6, 2) for k, v in sorted(busy.items(), key=lambda kv: -kv[1])[:12]})
print(f"whole: {(max(e[1] for e in ev) - t0) / 1e6:.2f} ms")
