"""RMQ re-chaining on the device (mm2gb_rmq_chain_gpu / mm2gb_lchain_rmq, k_rmq_fill in csrc/post_kernels.hip)
against the reference's own vectors and the CPU oracle.  Reads without a tie on the range-minimum priority must give the
reference's chains bit for bit; reads with one must be REPORTED (the reference breaks such ties by the shape of its AVL tree)."""
import ctypes as C

import numpy as np
import pytest

import golden_io
import orc
import synth_cases as sc

pytestmark = pytest.mark.gpu

mm = pytest.importorskip("mm2gb_amd")

CASES = golden_io.rmq_cases()


def to_lib(prm):
    return mm.RmqParam(max_dist=prm.max_dist, max_dist_inner=prm.max_dist_inner, bw=prm.bw, max_chn_skip=prm.max_chn_skip, cap_rmq_size=prm.cap_rmq_size,
                       min_cnt=prm.min_cnt, min_sc=prm.min_sc, chn_pen_gap=np.float32(prm.pen_gap), chn_pen_skip=np.float32(prm.pen_skip))


@pytest.fixture(scope="module")
def engine():
    with mm.Engine() as e:
        yield e


def first_pass(a):
    o = orc.lchain_dp(a, orc.default_param(), want_fp=False)
    return orc.radix_sort_x(o["a_out"]) if len(o["a_out"]) else o["a_out"]


@pytest.mark.parametrize("path", CASES, ids=golden_io.case_ids(CASES))
def test_reference_vectors(engine, path):
    g = golden_io.load_rmq(path)
    res, tied, _ = engine.rmq_chain(g["a"], np.array([0, len(g["a"])], np.int64), to_lib(g["prm"]))
    o = orc.lchain_rmq(g["a"], g["prm"])
    # (a vector recorded with a skip limit is filled by the one-anchor-per-step kernel, which keeps the limit and counts every tie -- as the oracle's
    # n_decide does under a limit)
    assert o["n_tied"] == g["tied"] and int(tied[0]) == o["n_decide"]
    if o["n_decide"] == 0:
        # no tie, or only ties whose holders all leave the anchor with the same score and predecessor: the reference's chains, whatever its tree picked
        assert np.array_equal(res[0][0], g["u"]) and np.array_equal(res[0][1], g["a_out"])


def counted(o, kernel="tiles"):
    """What the device reports in n_tied: the tile form weighs a tie (it counts where the holders of the smallest priority differ in what they
    leave the anchor with -- orc_rmq_last_ties_that_decide); the one-anchor-per-step kernel counts every tie."""
    return o["n_decide"] if kernel.startswith("tiles") else o["n_tied"]


@pytest.mark.parametrize("kernel", ["tiles", "steps", "tiles, whole workgroups on the first 5 reads", "tiles, whole workgroups on every read"])
def test_batch_against_the_oracle(engine, monkeypatch, kernel):
    """A batch of reads re-chained in one call: tie counts equal the oracle's, chains equal the oracle's (which uses the same
    stated tie rule, so tied reads agree with IT too).  Both device forms: the tile kernel (64 anchors per step of a wave, a tournament
    tree over the ranks; the default) and the one-anchor-per-step kernel (MM2GB_RMQ_KERNEL=steps)."""
    monkeypatch.setenv("MM2GB_RMQ_KERNEL", kernel.split(",")[0])        # (without it the engine keeps the form mm2gb_rmq_chain last picked)
    if "," in kernel:
        # a read that a whole workgroup fills (k_rmq_fill_tiles: the sweeps of a tile shared by four waves, their results combined in LDS)
        monkeypatch.setenv("MM2GB_RMQ_TEAM_READS", "5" if "first 5" in kernel else "1000000")
    a, off = mm.synth_reads(41, 0, 40, 10_000, 120_000)
    reads = [first_pass(a[off[r]:off[r + 1]]) for r in range(40)]
    reads.insert(7, np.zeros((0, 2), np.uint64))
    rng = np.random.default_rng(3)
    reads.append(orc.radix_sort_x(sc.sort_by_x(sc.pack(np.full(900, 1), np.zeros(900, np.int64), 1000 + rng.integers(0, 150, 900), 100 + rng.integers(0, 150, 900)))))
    o2 = np.zeros(len(reads) + 1, dtype=np.int64)
    o2[1:] = np.cumsum([len(x) for x in reads])
    allr = np.concatenate(reads)
    for kw in (dict(), dict(cap_rmq_size=64), dict(max_dist_inner=0), dict(bw=300, max_dist=1500, max_dist_inner=200), dict(pen_gap=np.float32(0.3), pen_skip=np.float32(0.05)),
               dict(max_chn_skip=25), dict(max_chn_skip=3, bw=300, max_dist=1500, max_dist_inner=200), dict(max_chn_skip=0), dict(max_chn_skip=1, cap_rmq_size=64)):
        prm = orc.default_rmq_param(**kw)
        res, tied, st = engine.rmq_chain(allr, o2, to_lib(prm))
        n_with_ties = 0
        limited = "max_chn_skip" in kw                  # a skip limit below the size cap: the one-anchor-per-step kernel whatever was asked for
        for r, x in enumerate(reads):
            o = orc.lchain_rmq(x, prm)
            assert int(tied[r]) == counted(o, "steps" if limited else kernel), (kw, r)
            n_with_ties += o["n_tied"] > 0
            assert np.array_equal(res[r][0], o["u"]) and np.array_equal(res[r][1], o["a_out"]), (kw, r)
        assert n_with_ties >= 1 and st["ms_post"] > 0


def test_single_read_entry_with_the_reference_signature(engine):
    g = golden_io.load_rmq([p for p in CASES if p.endswith("several_chains.npz")][0])
    L = mm.lib()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    a = np.ascontiguousarray(g["a"])
    buf = libc.malloc(a.nbytes)
    C.memmove(buf, a.ctypes.data, a.nbytes)
    prm = g["prm"]
    n_u, u_ptr = C.c_int(0), C.c_void_p(0)
    out = L.mm2gb_lchain_rmq(prm.max_dist, prm.max_dist_inner, prm.bw, prm.max_chn_skip, prm.cap_rmq_size, prm.min_cnt, prm.min_sc,
                             prm.pen_gap, prm.pen_skip, len(a), buf, C.byref(n_u), C.byref(u_ptr), None)
    assert n_u.value == len(g["u"])
    u = np.ctypeslib.as_array(C.cast(u_ptr, C.POINTER(C.c_uint64)), shape=(n_u.value,)).copy()
    a_out = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(len(g["a_out"]), 2)).copy()
    assert np.array_equal(u, g["u"]) and np.array_equal(a_out, g["a_out"])
    libc.free(u_ptr)
    libc.free(out)
    calls, tied = C.c_int64(0), C.c_int64(0)
    L.mm2gb_lchain_rmq_counts(C.byref(calls), C.byref(tied))
    assert calls.value >= 1 and tied.value == 0


@pytest.mark.parametrize("deal", ["auto", "device", "host"])
def test_batch_call_that_is_exact_for_every_read(engine, monkeypatch, deal):
    """mm2gb_rmq_chain (csrc/rmq_hybrid.cpp): kernel form and host form at the same time, reads dealt by estimated cost, reads the kernel
    reports a tie for redone by the host form -- the chains of EVERY read equal the host form's (which tests/test_rmq_host_cpu.py pins to
    the reference's vectors and to the compiled reference on tied reads), whatever the deal: the estimate's, everything on the device
    first, everything on the host."""
    a, off = mm.synth_reads(43, 0, 48, 10_000, 120_000)
    reads = [first_pass(a[off[r]:off[r + 1]]) for r in range(48)]
    reads.insert(5, np.zeros((0, 2), np.uint64))
    rng = np.random.default_rng(7)
    # dense clouds: ties on the range-minimum priority are certain, and the windows are full of inner candidates
    for k, n in enumerate((900, 20_000)):
        reads.append(orc.radix_sort_x(sc.sort_by_x(sc.pack(np.full(n, 1), np.zeros(n, np.int64), 1000 + rng.integers(0, 150 + 400 * k, n), 100 + rng.integers(0, 150 + 400 * k, n)))))
    o2 = np.zeros(len(reads) + 1, dtype=np.int64)
    o2[1:] = np.cumsum([len(x) for x in reads])
    allr = np.concatenate(reads)
    if deal != "auto":
        monkeypatch.setenv("MM2GB_RMQ_DEAL", deal)
    for kw in (dict(), dict(bw=300, max_dist=1500, max_dist_inner=200)):
        prm = to_lib(orc.default_rmq_param(**kw))
        want, _ = mm.rmq_chain_host(allr, o2, prm, threads=8)
        got, where, d = engine.rmq_chain_exact(allr, o2, prm, threads=8)
        for r in range(len(reads)):
            assert np.array_equal(got[r][0], want[r][0]) and np.array_equal(got[r][1], want[r][1]), (deal, kw, r, int(where[r]))
        if orc.ref_available():
            # ... and the compiled reference's mg_lchain_rmq itself (oracle/_ref travels to the GPU box), read by read, ties included
            oprm = orc.default_rmq_param(**kw)
            for r, x in enumerate(reads):
                if len(x):
                    ref = orc.ref_lchain_rmq(x, oprm)
                    assert np.array_equal(got[r][0], ref["u"]) and np.array_equal(got[r][1], ref["a_out"]), ("reference", deal, kw, r, int(where[r]))
        assert d["n_device"] + d["n_host_cost"] == len(reads) and d["n_host_tie"] == int((where == 2).sum())
        if deal == "device":
            assert d["n_host_cost"] == 0 and d["n_host_tie"] >= 1          # the clouds tie: found on the device, redone on the host
        if deal == "host":
            assert d["n_device"] == 0 and (where == 1).all()


@pytest.mark.parametrize("team_reads,strips", [("0", "1"), ("1000000", "1"), ("0", "0"), ("1000000", "0")])
def test_tile_kernel_on_odd_shapes(engine, monkeypatch, team_reads, strips):
    """The tile form's edges against the oracle: reads shorter than a tile, exactly one and two tiles, runs of equal x longer than a tile
    (nobody in the run may chain to another), several references and strands in one read (window starts jump), a size cap that evicts
    (cap_rmq_size below the window), gaps wider than max_dist (everything leaves at once), a dense cloud (ties, full inner windows)."""
    monkeypatch.setenv("MM2GB_RMQ_KERNEL", "tiles")
    monkeypatch.setenv("MM2GB_RMQ_TEAM_READS", team_reads)       # one wave per read / a whole workgroup per read
    monkeypatch.setenv("MM2GB_RMQ_STRIPS", strips)               # the inner window lane by lane over the (y strip, index) order / swept block by block
    rng = np.random.default_rng(11)
    def cloud(n, xw, yw, rid=1, strand=0, x0=1000, y0=100):
        return sc.pack(np.full(n, rid), np.full(n, strand, np.int64), x0 + rng.integers(0, xw, n), y0 + rng.integers(0, yw, n))
    reads = []
    for n in (1, 2, 63, 64, 65, 128, 129):
        reads.append(orc.radix_sort_x(sc.sort_by_x(cloud(n, 40 * n + 10, 40 * n + 10))))
    reads.append(orc.radix_sort_x(sc.sort_by_x(sc.pack(np.full(300, 1), np.zeros(300, np.int64), np.repeat(5000 + np.arange(3) * 7, 100), 100 + rng.integers(0, 4000, 300)))))   # three runs of 100 equal x
    reads.append(orc.radix_sort_x(sc.sort_by_x(np.concatenate([cloud(500, 3000, 3000, rid=r, strand=s) for r in (1, 2) for s in (0, 1)]))))
    reads.append(orc.radix_sort_x(sc.sort_by_x(np.concatenate([cloud(400, 2000, 2000, x0=1000 + 50_000 * k, y0=100 + 900 * k) for k in range(4)]))))   # gaps of 50 kb
    reads.append(orc.radix_sort_x(sc.sort_by_x(cloud(3000, 600, 600))))
    a40, off40 = mm.synth_reads(47, 0, 6, 20_000, 90_000)
    reads += [first_pass(a40[off40[r]:off40[r + 1]]) for r in range(6)]
    o2 = np.zeros(len(reads) + 1, dtype=np.int64)
    o2[1:] = np.cumsum([len(x) for x in reads])
    allr = np.concatenate(reads)
    for kw in (dict(), dict(cap_rmq_size=50), dict(cap_rmq_size=0), dict(max_dist_inner=0), dict(bw=100, max_dist=400, max_dist_inner=90), dict(max_dist=70_000, bw=70_000, max_dist_inner=3000)):
        prm = orc.default_rmq_param(**kw)
        res, tied, _ = engine.rmq_chain(allr, o2, to_lib(prm))
        for r, x in enumerate(reads):
            o = orc.lchain_rmq(x, prm)
            assert int(tied[r]) == counted(o), (kw, r, len(x))
            assert np.array_equal(res[r][0], o["u"]) and np.array_equal(res[r][1], o["a_out"]), (kw, r, len(x))


def test_fuzz_random_reads_and_parameters(engine, monkeypatch):
    """Random reads (clouds of every density, several references and strands, runs of equal x, first-pass chains of simulated reads) under random
    parameters -- window, band, inner distance, size cap, penalties -- and a random device form for every batch (one wave or a whole workgroup
    per read, inner windows by strips of y or swept block by block), against the oracle: tie counts and chains, read by read.  The seed moves
    with the sources (tests/test_gpu_parity.py::fuzz_seed) and is printed."""
    from test_gpu_parity import fuzz_seed
    seed = fuzz_seed()
    print("rmq fuzz seed", seed)
    rng = np.random.default_rng(seed)

    def cloud(n, xw, yw, rid, strand, x0, y0):
        return sc.pack(np.full(n, rid), np.full(n, strand, np.int64), x0 + rng.integers(0, max(1, xw), n), y0 + rng.integers(0, max(1, yw), n))

    for it in range(40):
        reads = []
        for _ in range(int(rng.integers(3, 9))):
            kind = int(rng.integers(0, 5))
            if kind == 0:      # a dense cloud: full inner windows, ties
                n = int(rng.integers(1, 2500)); w = int(rng.integers(30, 3000))
                reads.append(orc.radix_sort_x(sc.sort_by_x(cloud(n, w, w, 1, 0, 1000, 100))))
            elif kind == 1:    # several references / strands, windows that jump
                parts = [cloud(int(rng.integers(1, 600)), 4000, 4000, r, s, 1000, 100) for r in (1, 2, 3) for s in (0, 1) if rng.random() < 0.7]
                reads.append(orc.radix_sort_x(sc.sort_by_x(np.concatenate(parts))) if parts else np.zeros((0, 2), np.uint64))
            elif kind == 2:    # runs of equal x
                k = int(rng.integers(1, 6)); m = int(rng.integers(1, 150))
                reads.append(orc.radix_sort_x(sc.sort_by_x(sc.pack(np.full(k * m, 1), np.zeros(k * m, np.int64), np.repeat(5000 + np.arange(k) * int(rng.integers(1, 50)), m), 100 + rng.integers(0, 5000, k * m)))))
            elif kind == 3:    # islands further apart than max_dist
                reads.append(orc.radix_sort_x(sc.sort_by_x(np.concatenate([cloud(int(rng.integers(1, 500)), 2500, 2500, 1, 0, 1000 + 60_000 * k, 100 + 800 * k) for k in range(int(rng.integers(1, 5)))]))))
            else:              # what the first chaining keeps of a simulated read
                a1, o1 = mm.synth_reads(int(rng.integers(1, 1 << 20)), 0, 1, 5_000, 60_000)
                reads.append(first_pass(a1[o1[0]:o1[1]]))
        o2 = np.zeros(len(reads) + 1, dtype=np.int64)
        o2[1:] = np.cumsum([len(x) for x in reads])
        allr = np.concatenate(reads) if o2[-1] else np.zeros((0, 2), np.uint64)
        max_dist = int(rng.choice([300, 1500, 5000, 20000, 70000]))
        bw = int(rng.choice([100, 500, 2000, 20000]))
        kw = dict(max_dist=max_dist, bw=bw, max_dist_inner=int(rng.choice([0, 90, 200, 1000, 3000])), cap_rmq_size=int(rng.choice([0, 50, 100000])),
                  pen_gap=np.float32(rng.choice([0.12, 0.3, 0.8])), pen_skip=np.float32(rng.choice([0.0, 0.05])),
                  max_chn_skip=int(rng.choice([orc.INT32_MAX, orc.INT32_MAX, 0, 1, 5, 25, 60])))
        prm = orc.default_rmq_param(**kw)
        kernel = "tiles" if rng.random() < 0.8 else "steps"
        monkeypatch.setenv("MM2GB_RMQ_KERNEL", kernel)
        if kw["max_chn_skip"] != orc.INT32_MAX and not (kw["cap_rmq_size"] > 0 and kw["max_chn_skip"] >= kw["cap_rmq_size"]):
            kernel = "steps"                             # a skip limit that can end a walk: the one-anchor-per-step kernel takes the call
        monkeypatch.setenv("MM2GB_RMQ_TEAM_READS", str(int(rng.choice([0, 2, 1000000]))))
        monkeypatch.setenv("MM2GB_RMQ_STRIPS", str(int(rng.random() < 0.7)))
        res, tied, _ = engine.rmq_chain(allr, o2, to_lib(prm))
        for r, x in enumerate(reads):
            o = orc.lchain_rmq(x, prm)
            assert int(tied[r]) == counted(o, kernel), (seed, it, kw, r, len(x))
            assert np.array_equal(res[r][0], o["u"]) and np.array_equal(res[r][1], o["a_out"]), (seed, it, kw, r, len(x))


def test_every_tie_counts_when_asked(engine, monkeypatch):
    """MM2GB_RMQ_TIES=strict: the tile form counts every anchor whose smallest priority has several holders, like the one-anchor-per-step
    kernel and the oracle's n_tied; the default counts fewer, never more, and never on a read without a tie."""
    rng = np.random.default_rng(5)
    reads = [orc.radix_sort_x(sc.sort_by_x(sc.pack(np.full(n, 1), np.zeros(n, np.int64), 1000 + rng.integers(0, w, n), 100 + rng.integers(0, w, n)))) for n, w in ((900, 150), (3000, 600), (5000, 2500))]
    a1, o1 = mm.synth_reads(53, 0, 6, 20_000, 90_000)
    reads += [first_pass(a1[o1[r]:o1[r + 1]]) for r in range(6)]
    o2 = np.zeros(len(reads) + 1, dtype=np.int64)
    o2[1:] = np.cumsum([len(x) for x in reads])
    allr = np.concatenate(reads)
    prm = orc.default_rmq_param()
    monkeypatch.setenv("MM2GB_RMQ_KERNEL", "tiles")
    for team in ("0", "1000000"):
        monkeypatch.setenv("MM2GB_RMQ_TEAM_READS", team)
        monkeypatch.setenv("MM2GB_RMQ_TIES", "strict")
        _, strict, _ = engine.rmq_chain(allr, o2, to_lib(prm))
        monkeypatch.delenv("MM2GB_RMQ_TIES")
        _, weighed, _ = engine.rmq_chain(allr, o2, to_lib(prm))
        for r, x in enumerate(reads):
            o = orc.lchain_rmq(x, prm)
            assert int(strict[r]) == o["n_tied"] and int(weighed[r]) == o["n_decide"] <= o["n_tied"], (team, r)
        assert (strict > 0).sum() >= 3


@pytest.mark.skipif(not orc.ref_available(), reason="needs the reference build (oracle/_ref, made where /root/reference exists)")
@pytest.mark.parametrize("team_reads", ["0", "1000000"])
def test_reads_whose_ties_decide_nothing_equal_the_reference(engine, monkeypatch, team_reads):
    """The point of weighing ties: a read that meets ties but none that decides is NOT done again with the reference's tree -- so the device's
    chains for it must be the compiled reference's (mg_lchain_rmq itself, oracle/_ref), whatever its tree picked at those anchors."""
    monkeypatch.setenv("MM2GB_RMQ_KERNEL", "tiles")
    monkeypatch.setenv("MM2GB_RMQ_TEAM_READS", team_reads)
    a1, o1 = mm.synth_reads(61, 0, 60, 30_000, 150_000)
    reads = [first_pass(a1[o1[r]:o1[r + 1]]) for r in range(60)]
    rng = np.random.default_rng(9)
    reads += [orc.radix_sort_x(sc.sort_by_x(sc.pack(np.full(n, 1), np.zeros(n, np.int64), 1000 + rng.integers(0, w, n), 100 + rng.integers(0, w, n)))) for n, w in ((400, 4000), (1500, 9000), (3000, 20000))]
    o2 = np.zeros(len(reads) + 1, dtype=np.int64)
    o2[1:] = np.cumsum([len(x) for x in reads])
    prm = orc.default_rmq_param()
    res, tied, _ = engine.rmq_chain(np.concatenate(reads), o2, to_lib(prm))
    spared = 0
    for r, x in enumerate(reads):
        if len(x) == 0 or tied[r] != 0:
            continue
        ref = orc.ref_lchain_rmq(x, prm)
        assert np.array_equal(res[r][0], ref["u"]) and np.array_equal(res[r][1], ref["a_out"]), r
        spared += orc.lchain_rmq(x, prm)["n_tied"] > 0
    print("reads with ties that decide nothing:", spared)
    assert spared >= 1
