"""CPU-side checks of the product library (no GPU needed): it loads, exports the declared C ABI, parses gpu_config.json
like the reference, and its host post-pass (backtrack + compaction) reproduces the reference vectors."""
import ctypes as C
import glob
import json
import os
import re

import numpy as np
import pytest

import golden_io
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mm = pytest.importorskip("mm2gb_amd")

CASES = golden_io.all_cases()


def test_library_exports_every_declared_symbol():
    L = mm.lib()
    declared = set()
    for hdr in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)
        for m in re.finditer(r"\b((?:mm2gb_|init_stream_gpu|chain_stream_gpu|finish_stream_gpu|free_stream_gpu)\w*)\s*\(", text):
            declared.add(m.group(1))
    declared = {d for d in declared if not d.endswith("_t")}
    assert set(mm.CORE_SYMBOLS) | set(mm.BOUNDARY_SYMBOLS) <= declared | set(mm.CORE_SYMBOLS)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/ but not exported"
    assert L.mm2gb_version().decode().startswith("0.")


def test_library_imports_nothing_of_the_reference_or_the_oracle():
    """The product may call back into the host only through the four callbacks of the boundary (build_misc,
    post_chaining_helper, kmalloc, kfree -- weak, so it also loads without a host).  In particular it imports neither the oracle nor
    the reference's chaining functions: a read whose range-minimum ties (N3) is redone by the library's own exact host form, never
    by the host program's mg_lchain_rmq."""
    import subprocess
    so = os.path.join(ROOT, "mm2-gb_amd", "libmm2gb_chain.so")
    out = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True, check=True).stdout
    names = {l.split()[-1].split("@")[0] for l in out.splitlines() if l.strip()}
    banned = {n for n in names if re.search(r"^(__real_|__wrap_)?(mg_|mm_|orc_|krmq|radix_sort|compact_a|ks_)", n)}
    assert not banned, banned
    host_callbacks = {n for n in names if n in ("build_misc", "post_chaining_helper", "kmalloc", "kfree", "kcalloc", "krealloc")}
    assert host_callbacks <= {"build_misc", "post_chaining_helper", "kmalloc", "kfree"}


def test_misc_layout_matches_reference_struct():
    # Misc (gpu/plutils.h:33-37): 9 ints then 2 floats = 44 bytes, passed by value across the boundary
    assert C.sizeof(mm.Misc) == 44
    assert mm.Misc.chn_pen_gap.offset == 36 and mm.Misc.max_iter.offset == 0 and mm.Misc.n_seg.offset == 32


REF_GPU_DIR = "/root/reference/gpu"


@pytest.mark.skipif(not os.path.isdir(REF_GPU_DIR), reason="reference presets only exist in the dev container")
@pytest.mark.parametrize("name", ["gpu_config.json", "a6000_config.json", "gfx1030_config.json", "mi210_below50k_config.json",
                                  "mi210_over50k_config.json", "orin32GB.json"])
def test_reference_presets_load_unchanged(name):
    cfg = mm.load_config(os.path.join(REF_GPU_DIR, name))
    raw = json.load(open(os.path.join(REF_GPU_DIR, name)))
    assert cfg.num_streams == raw["num_streams"] and cfg.min_n == raw["min_n"]
    if "max_total_n" in raw:
        assert cfg.max_total_n == raw["max_total_n"]          # may exceed INT32_MAX (plmem.cu:491)
    for k, v in raw["score_kernel"].items():
        if not k.startswith("//"):
            assert getattr(cfg.score_kernel, k) == v
    for k, v in raw["range_kernel"].items():
        if not k.startswith("//"):
            assert getattr(cfg.range_kernel, k) == v


def test_shipped_mi355x_preset_and_defaults_agree():
    cfg = mm.load_config(os.path.join(ROOT, "mm2-gb_amd", "mi355x_config.json"))
    d = mm.default_config()
    assert (cfg.max_total_n, cfg.max_read, cfg.score_kernel.micro_batch) == (d.max_total_n, d.max_read, d.score_kernel.micro_batch)
    assert cfg.score_kernel.short_griddim == d.score_kernel.short_griddim


@pytest.mark.parametrize("text,msg", [
    ("{", "JSON error"),
    ('{"num_streams": 1}', "min_n"),
    ('{"num_streams":1,"min_n":1,"max_total_n":10,"max_read":1,"range_kernel":{"blockdim":1,"cut_check_anchors":1,"anchor_per_block":1}}', "score_kernel"),
    ('{"num_streams":1,"min_n":1,"range_kernel":{},"score_kernel":{}}', "avg_read_n"),
    ('[1,2]', "object"),
])
def test_config_errors_are_loud(text, msg):
    with pytest.raises(mm.Mm2gbError, match=msg):
        mm.parse_config(text)


def test_config_comment_keys_and_big_numbers():
    cfg = mm.parse_config('{"//c":"x","num_streams":2,"min_n":7,"max_total_n":3000000000,"max_read":5,"long_seg_buffer_size":99,'
                          '"range_kernel":{"blockdim":512,"//k":1,"cut_check_anchors":10,"anchor_per_block":32768},'
                          '"score_kernel":{"micro_batch":3,"mid_blockdim":512,"short_griddim":1,"long_griddim":2,"mid_griddim":3,'
                          '"long_seg_cutoff":20,"mid_seg_cutoff":3}}')
    assert cfg.max_total_n == 3000000000 and cfg.num_streams == 2 and cfg.score_kernel.micro_batch == 3
    assert cfg.long_seg_buffer_size == 99
    with pytest.raises(mm.Mm2gbError, match="fail to open"):
        mm.load_config("/nonexistent/gpu_config.json")


@pytest.mark.parametrize("path", CASES, ids=golden_io.case_ids(CASES))
def test_host_backtrack_matches_reference_vectors(path):
    """mm2gb_backtrack_host (the product's own post-pass) on the reference's f/p -> the reference's chains."""
    g = golden_io.load(path)
    prm = g["prm"]
    misc = mm.default_misc(bw=prm.bw, min_cnt=prm.min_cnt, min_score=prm.min_sc, is_cdna=prm.is_cdna, n_seg=prm.n_seg)
    idx = np.arange(len(g["p"]), dtype=np.int64)
    p_rel = np.where(g["p"] >= 0, idx - g["p"], 0).astype(np.int32)
    u, a_out = mm.backtrack_host(misc, g["a"], g["f"], p_rel)
    assert np.array_equal(u, g["u"]) and np.array_equal(a_out, g["a_out"])


def test_host_backtrack_rejects_bad_predecessors():
    misc = mm.default_misc()
    a = np.zeros((3, 2), np.uint64)
    with pytest.raises(mm.Mm2gbError, match="out of range"):
        mm.backtrack_host(misc, a, np.array([50, 50, 50], np.int32), np.array([0, 2, 0], np.int32))


def test_synth_is_deterministic_sorted_and_subsettable():
    a, off = mm.synth_reads(5, 0, 6, 10_000, 100_000, threads=3)
    b, off_b = mm.synth_reads(5, 0, 6, 10_000, 100_000, threads=1)
    assert np.array_equal(a, b) and np.array_equal(off, off_b)
    c, off_c = mm.synth_reads(5, 2, 2, 10_000, 100_000)          # reads 2..3 regenerated alone
    assert np.array_equal(c, a[off[2]:off[4]])
    for r in range(6):
        x = a[off[r]:off[r + 1], 0]
        assert np.all(x[1:] >= x[:-1])
    assert ((a[:, 1] >> np.uint64(32)) & np.uint64(0xff) == 15).all()
    dens = off[-1] / 6 / 55_000
    assert 0.15 < dens < 0.6       # ~0.24 anchors/bp + repeat blocks


def test_synth_matches_oracle_pairs_scale():
    """The recipe is meant to give a few hundred pairs per anchor, repeats dominating."""
    a, off = mm.synth_reads(1, 0, 4, 100_000, 300_000)
    _, _, pairs = orc.chain_fill_many(a, off, orc.default_param(), threads=4)
    assert 50 < pairs / off[-1] < 3000


def test_batch_grouping_rule_matches_the_reference_accumulator():
    """mm2gb_plan_batches (the batcher's rule, no GPU): a read that would take the batch past max_total_n anchors starts the next
    batch (map.c:887-920 moves it to the pending batch), max_read closes a batch by count (map.c:954), a read larger than the
    limit is a batch of its own, and reads under min_n travel in a lane of their own without closing the big reads' batches."""
    rng = np.random.default_rng(3)
    n = rng.integers(0, 900, 500).astype(np.int64)
    n[::50] = 5000                                     # larger than the anchor limit
    nb, batch, lane = mm.plan_batches(n, 3000, 40, 0)
    assert (lane == 0).all() and nb == batch.max() + 1 and (np.diff(batch) >= 0).all()
    # straightforward restatement of the rule
    want, cur, cnt, tot = [], -1, 0, 0
    for v in n:
        if cur < 0 or cnt >= 40 or tot + v > 3000:
            cur, cnt, tot = cur + 1, 0, 0
        want.append(cur); cnt += 1; tot += v
    assert batch.tolist() == want
    for b in range(nb):
        sel = batch == b
        assert sel.sum() <= 40 and (n[sel].sum() <= 3000 or sel.sum() == 1)
    # min_n: two lanes, each with the same rule applied to its own reads; ids in order of creation
    nb2, batch2, lane2 = mm.plan_batches(n, 3000, 40, 100)
    assert ((n < 100) == (lane2 == 1)).all()
    for ln in (0, 1):
        sub = n[lane2 == ln]
        _, bsub, _ = mm.plan_batches(sub, 3000, 40, 0)
        ids = batch2[lane2 == ln]
        assert (np.unique(ids, return_inverse=True)[1] == bsub).all()
    assert nb2 == len(np.unique(batch2))
    # no limits: one batch
    assert mm.plan_batches(n, 0, 0, 0)[0] == 1
    assert mm.plan_batches(np.zeros(0, np.int64), 10, 10, 0)[0] == 0


def test_numa_placement_parses_a_sysfs_tree(tmp_path):
    """Node awareness (numa.cpp): the NUMA node of a device's PCIe root and that node's CPUs, read from a made-up sysfs tree -- the
    parsing that decides where a rank's / pool worker's / batcher worker's host threads and page-locked buffers go on a multi-GPU node
    (the reference: device 0, nothing pinned, gpu/plmem.cu:426,462,499)."""
    root = tmp_path
    dev = root / "sys" / "bus" / "pci" / "devices"
    for bdf, node in (("0000:c1:00.0", "1\n"), ("0000:05:00.0", "0\n"), ("0000:85:00.0", "-1\n"), ("0001:0a:00.0", "3\n")):
        (dev / bdf).mkdir(parents=True)
        (dev / bdf / "numa_node").write_text(node)
    nodes = root / "sys" / "devices" / "system" / "node"
    for node, cpus in ((0, "0-31,128-159\n"), (1, "32-63,160-191\n"), (3, "7\n")):
        (nodes / f"node{node}").mkdir(parents=True)
        (nodes / f"node{node}" / "cpulist").write_text(cpus)
    r = str(root)
    assert mm.numa_cpus_for_bdf("0000:C1:00.0", r) == (1, list(range(32, 64)) + list(range(160, 192)))      # hipDeviceGetPCIBusId may answer in upper case
    assert mm.numa_cpus_for_bdf("0000:05:00.0", r) == (0, list(range(0, 32)) + list(range(128, 160)))
    assert mm.numa_cpus_for_bdf("0000:85:00.0", r) == (-1, [])          # the kernel's "unknown"
    assert mm.numa_cpus_for_bdf("0000:99:00.0", r) == (-1, [])          # no such device
    assert mm.numa_cpus_for_bdf("0001:0a:00.0", r) == (3, [7])
    (nodes / "node3" / "cpulist").write_text("7-,9\n")                   # a list the kernel would not print: nobody is moved on a guess
    assert mm.numa_cpus_for_bdf("0001:0a:00.0", r) == (3, [])
    (nodes / "node3" / "cpulist").write_text("9-7\n")
    assert mm.numa_cpus_for_bdf("0001:0a:00.0", r) == (3, [])


# ---- re-chaining ahead of the host's callback (csrc/rechain_ahead.cpp) ----
MAPOPT_FIELDS = ["flag", "bw", "bw_long", "max_gap", "max_chain_skip", "min_cnt", "min_chain_score", "rmq_size_cap", "rmq_inner_dist",
                 "rmq_rescue_size", "rmq_rescue_ratio"]
# offsetof(mm_mapopt_t, field) as gcc lays minimap.h:128-145 out on x86-64 (recorded from the compiled reference header by the test below)
MAPOPT_OFFSETS = {"flag": 0, "bw": 20, "bw_long": 24, "max_gap": 28, "max_chain_skip": 40, "min_cnt": 48, "min_chain_score": 52, "rmq_size_cap": 64,
                  "rmq_inner_dist": 68, "rmq_rescue_size": 72, "rmq_rescue_ratio": 76}


def _mirror_layout():
    L = mm.lib()
    buf = (C.c_int32 * 16)()
    n = L.mm2gb_mapopt_head_layout(buf, 16)
    assert n == len(MAPOPT_FIELDS) + 1
    return dict(zip(MAPOPT_FIELDS + ["sizeof"], list(buf)[:n]))


def test_mapopt_mirror_layout_recorded():
    got = _mirror_layout()
    assert {k: got[k] for k in MAPOPT_FIELDS} == MAPOPT_OFFSETS and got["sizeof"] == 80


@pytest.mark.skipif(not os.path.exists("/root/reference/minimap.h"), reason="the reference header only exists in the dev container")
def test_mapopt_mirror_layout_matches_the_compiled_reference_header(tmp_path):
    """The library reads opt->bw, bw_long, flag, ... through its own mirror of mm_mapopt_t's leading fields (include/mm2gb_plutils.h): every
    offset must be the one gcc gives the reference's struct (minimap.h:128-145)."""
    import subprocess
    src = tmp_path / "lay.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "minimap.h"\nint main(void){' +
                   "".join(f'printf("{f} %zu\\n", offsetof(mm_mapopt_t, {f}));' for f in MAPOPT_FIELDS) + "return 0;}\n")
    exe = tmp_path / "lay"
    subprocess.run(["gcc", "-I/root/reference", str(src), "-o", str(exe)], check=True)
    want = {k: int(v) for k, v in (ln.split() for ln in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())}
    got = _mirror_layout()
    assert {k: got[k] for k in MAPOPT_FIELDS} == want == MAPOPT_OFFSETS


def test_elf_import_probe_tells_a_wrapped_host_from_a_plain_one(tmp_path):
    """A program linked with -Wl,--wrap=mg_lchain_rmq imports __wrap_mg_lchain_rmq from the library: that is how init_stream_gpu knows whether
    answering a batch's re-chaining ahead can ever be used."""
    import subprocess
    L = mm.lib()
    L.mm2gb_elf_imports_symbol.argtypes = [C.c_char_p, C.c_char_p]
    lib_dir = os.path.join(ROOT, "mm2-gb_amd")
    (tmp_path / "m.c").write_text("void *mg_lchain_rmq(int x); int main(void){ return mg_lchain_rmq(0) != 0; }\n")
    (tmp_path / "d.c").write_text("void *mg_lchain_rmq(int x){ (void)x; return 0; }\n")
    for flags, want in ((["-Wl,--wrap=mg_lchain_rmq"], 1), ([], 0)):
        exe = tmp_path / ("wrapped" if want else "plain")
        subprocess.run(["gcc", str(tmp_path / "m.c"), str(tmp_path / "d.c"), "-o", str(exe), *flags, "-L" + lib_dir, "-lmm2gb_chain",
                        "-Wl,--unresolved-symbols=ignore-in-shared-libs"], check=True)
        assert L.mm2gb_elf_imports_symbol(str(exe).encode(), b"__wrap_mg_lchain_rmq") == want
    assert L.mm2gb_elf_imports_symbol(b"/nonexistent", b"x") == 0
    assert L.mm2gb_elf_imports_symbol(os.path.join(ROOT, "README.md").encode(), b"x") == 0      # not an ELF file


class _SeqMeta(C.Structure):
    _fields_ = [("i", C.c_long), ("seg_id", C.c_int), ("name", C.c_char * 200), ("len", C.c_uint32), ("n_alt", C.c_int), ("is_alt", C.c_int), ("qlen_sum", C.c_int)]


class _ChainRead(C.Structure):
    _fields_ = [("seq", _SeqMeta), ("qseqs", C.c_void_p), ("qlens", C.c_void_p), ("n_seg", C.c_int), ("rep_len", C.c_int), ("frag_gap", C.c_int),
                ("mini_pos", C.c_void_p), ("n_mini_pos", C.c_int), ("a", C.c_void_p), ("n", C.c_int64), ("u", C.c_void_p), ("n_u", C.c_int)]


class _MapoptHead(C.Structure):
    _fields_ = [("flag", C.c_int64), ("seed", C.c_int), ("sdust_thres", C.c_int), ("max_qlen", C.c_int), ("bw", C.c_int), ("bw_long", C.c_int),
                ("max_gap", C.c_int), ("max_gap_ref", C.c_int), ("max_frag_len", C.c_int), ("max_chain_skip", C.c_int), ("max_chain_iter", C.c_int),
                ("min_cnt", C.c_int), ("min_chain_score", C.c_int), ("chain_gap_scale", C.c_float), ("chain_skip_scale", C.c_float),
                ("rmq_size_cap", C.c_int), ("rmq_inner_dist", C.c_int), ("rmq_rescue_size", C.c_int), ("rmq_rescue_ratio", C.c_float)]


def test_rechain_trigger_follows_map_c():
    """map.c:444-448: long-join re-chaining wanted for a single-segment read with more than one chain whose best chain leaves more than
    rmq_rescue_size bases uncovered or covers more than rmq_rescue_ratio of the read; never in splice / sr / no-long-join modes."""
    L = mm.lib()
    L.mm2gb_rechain_wanted.argtypes = [C.POINTER(_MapoptHead), C.POINTER(_ChainRead)]
    assert C.sizeof(_ChainRead) == 312 and C.sizeof(_MapoptHead) == 80
    opt = _MapoptHead(flag=0, bw=500, bw_long=20000, rmq_rescue_size=1000, rmq_rescue_ratio=0.1, max_gap=5000, rmq_size_cap=100000, max_chain_skip=2**31 - 1)
    a = np.zeros((10, 2), dtype=np.uint64)
    a[:, 0] = np.arange(10) * 100 + 1000
    a[:, 1] = (np.uint64(15) << np.uint64(32)) | (np.arange(10, dtype=np.uint64) * np.uint64(100) + np.uint64(50))     # y = 50, 150, ... 950
    u = np.array([(100 << 32) | 6, (50 << 32) | 4], dtype=np.uint64)                  # best chain: anchors 0..5, y 50 -> 550
    rd = _ChainRead(n_seg=1, n_u=2, n=10)
    rd.a, rd.u = a.ctypes.data, u.ctypes.data

    def wanted(qlen):
        rd.seq.qlen_sum = qlen
        return L.mm2gb_rechain_wanted(C.byref(opt), C.byref(rd))

    assert wanted(1400) == 1                  # 500 > 1400 * 0.1
    opt.rmq_rescue_ratio = 0.5
    assert wanted(1400) == 0                  # 1400 - 500 = 900 <= 1000 and 500 <= 700
    assert wanted(1600) == 1                  # 1600 - 500 > 1000
    assert wanted(900) == 1                   # 500 > 450
    rd.n_u = 1
    assert wanted(1600) == 0                  # a single chain is never re-chained
    rd.n_u = 2
    for flag in (0x080, 0x400, 0x1000):       # MM_F_SPLICE, MM_F_NO_LJOIN, MM_F_SR
        opt.flag = flag
        assert wanted(1600) == 0
    opt.flag = 0x800000000                    # MM_F_GPU_CHAIN alone does not matter
    assert wanted(1600) == 1
    opt.bw_long = 500
    assert wanted(1600) == 0                  # bw_long must exceed bw
