#!/usr/bin/env python3
"""What the compiler made of k_score's loops (VERDICT r04 item 6 i): gfx950 assembly of csrc/chain_kernels.hip with line tables, every loop of a
kernel instantiation (a backward branch and its target) with its instruction mix -- vector ALU (of which the kinds that issue at half rate on
gfx950, profiles/ubench/r02_valu_rate.txt), scalar, LDS, memory, waits -- the source function most of its instructions come from, and any
spill traffic inside it (scratch_load / scratch_store, v_readlane / v_writelane marked "Reload" / "Spill" by the compiler).
    python3 profiles/isa_loops.py [kernel substring, default 'k_scoreILi0ELb0ELb0E'] [min instructions, default 6]  > profiles/r05_isa_loops.txt
Needs hipcc (cross-compiles without a GPU)."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mm2-gb_amd", "csrc", os.environ.get("MM2GB_ISA_SRC", "chain_kernels.hip"))      # MM2GB_ISA_SRC=post_kernels.hip for the post-pass / RMQ kernels
want = sys.argv[1] if len(sys.argv) > 1 else "k_scoreILi0ELb0ELb0E"
min_len = int(sys.argv[2]) if len(sys.argv) > 2 else 6
HALF = re.compile(r"^v_(sad_u32|min3|max3|lshl_add|add3|and_or|min_|max_|cmp|cmpx|readlane|readfirstlane|writelane|cvt_|mul_lo|mul_hi|mad_|bfe|perm|lshl_or|or3|xad|add_lshl|med3)")

with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-gline-tables-only", "-I" + os.path.join(ROOT, "include"),
                    "--cuda-device-only", "-S", SRC, "-o", asm], check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().splitlines()

# source functions by line (definitions at column 0 of the .hip file: "__device__ ... name(" / "template" blocks)
src = open(SRC).read().splitlines()
fn_at = []
for i, ln in enumerate(src, 1):
    m = re.match(r"^(?:template\s*<[^>]*>\s*)?(?:static\s+)?(?:__device__|__global__)[^()]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", ln)
    if m:
        fn_at.append((i, m.group(1)))
    else:
        m = re.match(r"^\s+__device__ __forceinline__ [^()]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", ln)      # member functions
        if m:
            fn_at.append((i, m.group(1)))


def fn_of(line):
    name = "?"
    for at, n in fn_at:
        if at <= line:
            name = n
        else:
            break
    return name


# the kernel's text
start = next(i for i, ln in enumerate(lines) if ln.startswith("_ZN5mm2gb") and want in ln and ln.rstrip().split(":")[0].endswith(ln.split(":")[0]) and ":" in ln)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
file_ids = {}
for ln in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m:
        file_ids[int(m.group(1))] = (m.group(3) or m.group(2))
body = []          # (kind, text, srcline, is_main_file)
labels = {}
cur = (0, False)
for i in range(start + 1, end):
    ln = lines[i]
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
    if m:
        cur = (int(m.group(2)), file_ids.get(int(m.group(1)), "").endswith(os.path.basename(SRC)))
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", ln)
    if m:
        labels[m.group(1)] = len(body)
        continue
    t = ln.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    body.append((t.split()[0], t, cur[0] if cur[1] else 0))

# registers that exist only to hold spilled scalars: written by v_writelane_b32 and touched by nothing but v_readlane_b32 / v_writelane_b32
reg_users = collections.defaultdict(set)
for op, text, _ in body:
    for r in re.findall(r"\bv(\d+)\b", text):
        reg_users[int(r)].add(op)
spill_vgprs = {r for r, ops in reg_users.items() if "v_writelane_b32" in ops and ops <= {"v_writelane_b32", "v_readlane_b32"}}


def is_spill(o):
    if "Spill" in o[1] or "Reload" in o[1] or o[0].startswith("scratch_"):
        return True
    if o[0] == "v_writelane_b32":
        m = re.match(r"v_writelane_b32 v(\d+)", o[1]); return bool(m) and int(m.group(1)) in spill_vgprs
    if o[0] == "v_readlane_b32":
        m = re.match(r"v_readlane_b32 s\d+, v(\d+)", o[1]); return bool(m) and int(m.group(1)) in spill_vgprs
    return False


def mix(ops):
    valu = [o for o in ops if o[0].startswith("v_") and not is_spill(o)]
    return {"n": len(ops), "valu": len(valu), "half": sum(1 for o in valu if HALF.match(o[0])),
            "salu": sum(1 for o in ops if o[0].startswith("s_") and not o[0].startswith(("s_waitcnt", "s_nop"))),
            "lds": sum(1 for o in ops if o[0].startswith("ds_")), "mem": sum(1 for o in ops if o[0].startswith(("global_", "buffer_", "flat_"))),
            "wait": sum(1 for o in ops if o[0].startswith("s_waitcnt")), "spill": sum(1 for o in ops if is_spill(o)),
            "bread": sum(1 for o in ops if o[0] in ("ds_read_b128", "ds_read2_b64", "ds_read_b64") )}


# ---- table A: the sweeps (fully unrolled, so no loop of their own): instructions the compiler attributes to each sweep function's source lines,
# per broadcast read of a staged source (ds_read_b128: one per source and 64 -- two-tile forms: 128 -- targets)
fn_ranges = []
for k, (at, name) in enumerate(fn_at):
    fn_ranges.append((at, fn_at[k + 1][0] - 1 if k + 1 < len(fn_at) else len(src), name))
by_fn = collections.defaultdict(list)
for o in body:
    if o[2]:
        by_fn[fn_of(o[2])].append(o)
print(f"# {want}: {len(body)} instructions; registers that only hold spilled scalars: {sorted(spill_vgprs)}")
print("# A. per source function (line tables; inlined copies summed): instructions | vector ALU (half-rate kinds) | scalar | LDS | memory | waits | spill traffic | per broadcast read of a source: vector / scalar")
for name, ops in sorted(by_fn.items(), key=lambda kv: -len(kv[1])):
    m = mix(ops)
    if m["n"] < 40:
        continue
    per = f"{m['valu'] / m['bread']:.2f} / {m['salu'] / m['bread']:.2f}" if m["bread"] >= 8 else "-"
    print(f"{name:28s} {m['n']:6d} | {m['valu']:5d} ({m['half']:5d}) | {m['salu']:5d} | {m['lds']:4d} | {m['mem']:3d} | {m['wait']:4d} | {m['spill']:3d} | {per}")

loops = []
for k, (op, text, _) in enumerate(body):
    if op.startswith("s_cbranch") or op == "s_branch":
        tgt = text.split()[1].rstrip(",")
        if tgt in labels and labels[tgt] <= k:
            loops.append((labels[tgt], k))
loops = sorted(set(loops))
print(f"\n# B. loops (a backward branch and its target): {len(loops)}; the innermost ones of >= {min_len} instructions, most spill traffic first, then largest (at most 40 shown)")
print("# source function (most instructions) : instructions | vector ALU (half-rate) | scalar | LDS | memory | waits | spill traffic inside")
rows = []
for a, b in loops:
    n = b - a + 1
    ops = body[a:b + 1]
    m = mix(ops)
    inner = any(a <= a2 and b2 <= b and (a2, b2) != (a, b) for a2, b2 in loops)
    if inner or n < min_len:
        continue
    srcl = [o[2] for o in ops if o[2] > 60]
    fns = collections.Counter(fn_of(x) for x in srcl)
    top = ", ".join(f"{f} {c}" for f, c in fns.most_common(2))
    rows.append((-m["spill"], -n, f"{top:48s} : {n:5d} | {m['valu']:4d} ({m['half']:4d}) | {m['salu']:4d} | {m['lds']:3d} | {m['mem']:3d} | {m['wait']:3d} | {m['spill']:3d}" + ("  (has inner loops)" if inner else "")))
for _, _, r in sorted(rows)[:40]:
    print(r)
tot_spill = [o for o in body if is_spill(o)]
in_loops = set()
for a, b in loops:
    in_loops.update(range(a, b + 1))
innermost = [(a, b) for a, b in loops if not any(a <= a2 and b2 <= b and (a2, b2) != (a, b) for a2, b2 in loops)]
in_inner = set()
for a, b in innermost:
    in_inner.update(range(a, b + 1))
sp_in = [k for k, o in enumerate(body) if is_spill(o) and k in in_loops]
sp_inner = [k for k in sp_in if k in in_inner]
print(f"\n# spill instructions in the whole kernel: {len(tot_spill)} ({sum(1 for o in tot_spill if o[0].startswith('scratch_'))} scratch_*, the rest v_writelane / v_readlane of scalar registers); inside any loop: {len(sp_in)}; inside an INNERMOST loop: {len(sp_inner)}")
by_fn = collections.Counter(fn_of(body[k][2]) for k in sp_in if body[k][2])
print("# spill instructions inside loops, by source function:", dict(by_fn.most_common(12)))
