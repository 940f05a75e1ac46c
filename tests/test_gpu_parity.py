"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the committed reference vectors.
Bit-exact on every per-anchor score f[] and predecessor p[], and on chains u[] / compacted anchors."""
import numpy as np
import pytest

import golden_io
import orc
import synth_cases as sc

pytestmark = pytest.mark.gpu

mm = pytest.importorskip("mm2gb_amd")

CASES = golden_io.all_cases()


def misc_from(prm):
    return mm.default_misc(max_iter=prm.max_iter, max_dist_x=prm.max_dist_x, max_dist_y=prm.max_dist_y, max_skip=orc.INT32_MAX,
                           bw=prm.bw, min_cnt=prm.min_cnt, min_score=prm.min_sc, is_cdna=prm.is_cdna, n_seg=prm.n_seg,
                           chn_pen_gap=np.float32(prm.pen_gap), chn_pen_skip=np.float32(prm.pen_skip))


def rel(p):
    idx = np.arange(len(p), dtype=np.int64)
    return np.where(p >= 0, idx - p, 0).astype(np.int32)


@pytest.fixture(scope="module")
def engine():
    with mm.Engine() as e:
        yield e


def check_batch(engine, a, off, prm, threads=4):
    engine.set_misc(misc_from(prm))
    f, p, st = engine.score(a, off)
    fo, po, pairs = orc.chain_fill_many(a, off, prm, threads=threads)
    po_rel = np.concatenate([rel(po[off[r]:off[r + 1]]) for r in range(len(off) - 1)]) if len(a) else np.zeros(0, np.int32)
    bad = np.flatnonzero((f != fo) | (p != po_rel))
    assert bad.size == 0, f"{bad.size} anchors differ, first at {bad[:5]}: gpu f/p {f[bad[:5]]}/{p[bad[:5]]} oracle {fo[bad[:5]]}/{po_rel[bad[:5]]}"
    assert st["n_pairs"] == pairs
    return st


def every_anchor_against_the_oracle(a, off, f, p, prm, n_pairs):
    """f / p of a whole batch against the oracle's, every anchor: the oracle fills all reads on as many threads as the process may use
    (the port does ~3 G pairs/s on the GPU box's 16 CPUs: a 2.7e11-pair batch in ~90 s).  Compared read by read to keep the
    temporaries small."""
    import bench
    fo, po, pairs = orc.chain_fill_many(a, off, prm, threads=max(1, bench.cpu_quota()))
    assert pairs == n_pairs
    assert np.array_equal(f, fo), f"{np.count_nonzero(f != fo)} scores differ, first at anchor {int(np.flatnonzero(f != fo)[0])}"
    del fo
    for r in range(len(off) - 1):
        lo, hi = off[r], off[r + 1]
        assert np.array_equal(p[lo:hi], rel(po[lo:hi])), f"predecessors of read {r} differ"


def device_post_pass_against_the_host_post_pass(e, a, off):
    """Chains and compacted anchors of every read of the batch: backtrack + compaction on the device (mm2gb_chain_gpu) against the
    host post-pass (mm2gb_chain_host, which tests/test_host_cpu.py pins to the reference's vectors) on the same scores."""
    import bench
    dev, _ = e.chain_gpu(a, off)
    host, _ = e.chain(a, off, threads=max(1, bench.cpu_quota()))
    assert len(dev) == len(host) == len(off) - 1
    n_chains = 0
    for r in range(len(off) - 1):
        assert np.array_equal(dev[r][0], host[r][0]) and np.array_equal(dev[r][1], host[r][1]), f"chains of read {r} differ"
        n_chains += len(dev[r][0])
    return n_chains


@pytest.mark.parametrize("path", CASES, ids=golden_io.case_ids(CASES))
def test_reference_vectors(engine, path):
    """Every committed reference vector.  Vectors recorded with a finite max_skip are re-derived with the oracle at
    max_skip = INT32_MAX (the GPU path is exhaustive by contract); all others are compared to the reference's own f/p."""
    g = golden_io.load(path)
    prm = g["prm"]
    a = g["a"]
    off = np.array([0, len(a)], dtype=np.int64)
    engine.set_misc(misc_from(prm))
    f, p, st = engine.score(a, off)
    if prm.max_skip == orc.INT32_MAX:
        assert np.array_equal(f, g["f"]) and np.array_equal(p, rel(g["p"]))
        res, _ = engine.chain(a, off)
        assert np.array_equal(res[0][0], g["u"]) and np.array_equal(res[0][1], g["a_out"])
    else:
        prm.max_skip = orc.INT32_MAX
        fo, po, _ = orc.chain_fill(a, prm)
        assert np.array_equal(f, fo) and np.array_equal(p, rel(po))


def test_multi_read_batch(engine):
    a, off = sc.multi_read_batch(40, 3)
    st = check_batch(engine, a, off, orc.default_param())
    assert st["n_chunks"] >= 1 and st["n_reads"] == 40


def test_reads_never_chain_across_each_other(engine):
    """Two copies of one read back to back share rid/strand/positions; windows must stop at the read boundary."""
    one = sc.read_like(15000, 5)
    a = np.concatenate([one, one, one])
    off = np.array([0, len(one), 2 * len(one), 3 * len(one)], dtype=np.int64)
    check_batch(engine, a, off, orc.default_param())


def test_empty_and_tiny_inputs(engine):
    prm = orc.default_param()
    engine.set_misc(misc_from(prm))
    f, p, st = engine.score(np.zeros((0, 2), np.uint64), np.array([0], np.int64))
    assert len(f) == 0 and st["n_pairs"] == 0
    f, p, st = engine.score(np.zeros((0, 2), np.uint64), np.array([0, 0, 0], np.int64))      # reads without anchors
    assert len(f) == 0
    one = sc.noise(1, 3)
    f, p, _ = engine.score(one, np.array([0, 1], np.int64))
    assert f.tolist() == [15] and p.tolist() == [0]
    # ragged: empty reads between non-empty ones
    r = sc.read_like(4000, 9)
    a = np.concatenate([r, r])
    off = np.array([0, 0, len(r), len(r), len(r), 2 * len(r), 2 * len(r)], dtype=np.int64)
    check_batch(engine, a, off, prm)


@pytest.mark.parametrize("kw", [dict(is_cdna=1), dict(n_seg=2), dict(n_seg=2, is_cdna=1), dict(pen_skip=np.float32(0.05)),
                                dict(bw=100, max_dist_x=50, max_dist_y=60), dict(max_iter=64), dict(max_iter=1), dict(max_dist_y=200),
                                dict(pen_gap=np.float32(0.0)), dict(pen_gap=np.float32(1.7), pen_skip=np.float32(0.3))])
def test_parameter_variants(engine, kw):
    prm = orc.default_param(**kw)
    a1 = sc.two_segments(500, 21) if kw.get("n_seg", 1) > 1 else sc.read_like(9000, 22)
    a2 = sc.variable_span(700, 23)
    a = np.concatenate([a1, a2])
    off = np.array([0, len(a1), len(a)], dtype=np.int64)
    check_batch(engine, a, off, prm)


def test_saturated_windows_and_rescue(engine):
    """Windows cut by max_iter and the max_ii rescue, several shapes in one batch."""
    parts = [sc.rescue_case(n_noise=700, n_chain=40, seed=3), sc.sort_by_x(np.concatenate([sc.repeat_block(900, 31), sc.colinear(300, 32)])),
             sc.rescue_case(n_noise=260, n_chain=80, seed=4), sc.read_like(6000, 33)]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in parts])
    st = check_batch(engine, np.concatenate(parts), off, orc.default_param(max_iter=200))
    assert st["n_tracked_chunks"] >= 1 and st["n_clamped_blocks"] >= 1


def test_rescue_state_restarts_at_every_read(engine):
    """The remembered best anchor (max_ii, lchain.c:156,189-205) belongs to one read.  Reads that map to the same region
    put the next read's first anchors within max_dist_x of the previous read's remembered anchor, on the same strand and
    reference; the state must not carry over.  First case: the batch a seeded fuzz run found (730 anchors, two reads,
    max_iter 7); second: several copies of one repeat-rich read back to back."""
    import json, os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regress", "fuzz_keep_across_reads.npz"))
    kw = json.loads(str(z["kw"]))
    prm = orc.default_param(**{k: (np.float32(v) if k.startswith("pen") else int(v)) for k, v in kw.items()})
    st = check_batch(engine, z["a"], z["off"], prm)
    assert st["n_reads"] == 2 and st["n_chunks"] == 1 and st["n_tracked_chunks"] == 1
    one = sc.sort_by_x(np.concatenate([sc.repeat_block(500, 61, xwin=300, ywin=400), sc.colinear(120, 62, max_gap=20)]))
    for max_iter in (7, 63, 200):
        reads = [one, one[: len(one) // 2], one, one[len(one) // 3:]]
        off = np.zeros(len(reads) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(r) for r in reads])
        st = check_batch(engine, np.concatenate(reads), off, orc.default_param(max_iter=max_iter, max_dist_x=500, bw=100))
        assert st["n_tracked_chunks"] >= 1


def test_unchecked_sweep_and_tiles_that_span_reads(engine):
    """A chunk may hold several reads; a tile pair whose second tile reaches into the next read has a 'last anchor' whose position says
    nothing about the first tile's distances.  A seeded fuzz run found it (max_dist_y = 10, bw = 1: a source 32 bases left of its
    targets was accepted by the unchecked sweep of the first tile).  Then the same shape made on purpose: short reads over one region,
    read boundaries at every offset inside tile pairs, bounds that make almost every pair fail the range test."""
    import json, os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regress", "fuzz_tiny_dist_y.npz"), allow_pickle=True)
    kw = json.loads(str(z["kw"]))
    prm = orc.default_param(**{k: (np.float32(v) if k.startswith("pen") else int(v)) for k, v in kw.items()})
    check_batch(engine, z["a"], z["off"], prm)
    rng = np.random.default_rng(77)
    reads = []
    for r in range(40):
        n = int(rng.integers(30, 260))
        x = 1_000_000 + np.sort(rng.integers(0, 6000, n)) + int(rng.integers(-40000, 40000)) * (r % 3 == 0)
        y = 500 + (x - x.min()) + rng.integers(-12, 13, n)
        reads.append(sc.sort_by_x(sc.pack(np.full(n, 2), np.full(n, r & 1), x, y)))
    off = np.zeros(len(reads) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in reads])
    a = np.concatenate(reads)
    for kw in (dict(max_dist_y=10, bw=1, max_dist_x=20000, max_iter=200), dict(max_dist_y=40, bw=8, max_iter=500), dict(max_dist_y=300, bw=100), dict()):
        check_batch(engine, a, off, orc.default_param(**kw))


@pytest.mark.parametrize("env", [{}, {"MM2GB_LONG_MIN_COST": "1", "MM2GB_LONG_MIN_WINDOW": "1", "MM2GB_WIDE_WINDOW": "1", "MM2GB_WHOLE_WG_PCT": "0"},
                                 {"MM2GB_LONG_MIN_COST": "1", "MM2GB_LONG_MIN_WINDOW": "1", "MM2GB_WIDE_WINDOW": "1", "MM2GB_WHOLE_WG_PCT": "1"},
                                 {"MM2GB_LONG_MIN_COST": "1", "MM2GB_LONG_MIN_WINDOW": "1", "MM2GB_WIDE_WINDOW": "100000000"}],
                         ids=["planner", "teams-of-8", "whole-workgroup", "teams-of-4"])
def test_chunk_lengths_around_tile_pairs(monkeypatch, env):
    """A wave works on two tiles of 64 anchors at a time: chunks that end inside the first tile, exactly between the two,
    inside the second, and one anchor into the next pair -- dense (every window reaches back across tiles) and sparse, with
    windows cut by a small max_iter so that the rescue state crosses the tile boundaries too."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    lengths = [1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 320]
    dense = [sc.sort_by_x(sc.repeat_block(n, 100 + n, xwin=300, ywin=400)) for n in lengths]
    sparse = [sc.colinear(n, 200 + n, max_gap=30) for n in lengths]
    reads = dense + sparse
    off = np.zeros(len(reads) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(r) for r in reads])
    a = np.concatenate(reads)
    with mm.Engine() as e:
        for kw in (dict(), dict(max_iter=40), dict(max_iter=100, bw=50)):
            check_batch(e, a, off, orc.default_param(**kw))
        # the planner joins short reads into one chunk; alone in its batch a read IS the chunk, so the lengths above are chunk lengths
        prm = orc.default_param(max_iter=100)
        for r in dense + sparse[::3]:
            check_batch(e, r, np.array([0, len(r)], np.int64), prm, threads=1)


def test_default_max_iter_repeat_block(engine):
    a = sc.sort_by_x(np.concatenate([sc.repeat_block(7000, 41), sc.colinear(800, 42), sc.noise(3000, 43)]))
    st = check_batch(engine, a, np.array([0, len(a)], np.int64), orc.default_param())
    assert st["n_tracked_chunks"] >= 1


def test_equal_reference_positions(engine):
    """Runs of anchors sharing one reference position (dr == 0 is rejected, lchain.c:120): short runs, runs longer than a
    tile, runs longer than the kernel's bounded scalar walk, inside windows that other anchors chain through."""
    rng = np.random.default_rng(5)
    parts = []
    for run in (1, 2, 3, 70, 200):
        base = sc.colinear(300, 100 + run, r0=500_000, q0=200, max_gap=12)
        xs = int(base[150, 0] & np.uint64(0xffffffff))
        dup = sc.pack(np.full(run, 3), np.zeros(run, np.int64), np.full(run, xs), np.sort(rng.integers(200, 6000, run)))
        parts.append(sc.sort_by_x(np.concatenate([base, dup])))
    # a dense block where most positions repeat (12k anchors on 4k positions), wide enough for cooperative mode
    parts.append(sc.sort_by_x(sc.repeat_block(6000, 77, xwin=1500, ywin=5000)))
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in parts])
    st = check_batch(engine, np.concatenate(parts), off, orc.default_param())
    assert st["n_long_chunks"] >= 1


def test_cooperative_and_wave_modes_agree(monkeypatch):
    """Heavy chunks through the team modes (every team size) and, with them switched off, through the wave mode."""
    parts = [sc.sort_by_x(sc.repeat_block(4500, 51)), sc.sort_by_x(np.concatenate([sc.repeat_block(7000, 52), sc.colinear(900, 53)])),
             sc.read_like(9000, 54), sc.rescue_case(n_noise=6000, n_chain=50, seed=9)]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in parts])
    a = np.concatenate(parts)
    prm = orc.default_param()
    with mm.Engine() as e1:
        st1 = check_batch(e1, a, off, prm)
    assert st1["n_long_chunks"] >= 3 and st1["n_tracked_chunks"] >= 2
    # every way of running the big-team list: all chunks by whole workgroups, none (8-wave teams only), 16-wave teams only
    for env in ({"MM2GB_WHOLE_WG_PCT": "1"}, {"MM2GB_WHOLE_WG_PCT": "0"}, {"MM2GB_BIG_TEAM": "16"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with mm.Engine() as e:
            st = check_batch(e, a, off, prm)
        assert st["n_long_chunks"] == st1["n_long_chunks"]
        for k in env:
            monkeypatch.delenv(k)
    monkeypatch.setenv("MM2GB_NO_COOP", "1")
    with mm.Engine() as e2:
        st2 = check_batch(e2, a, off, prm)
    assert st2["n_long_chunks"] == 0 and st2["n_tracked_chunks"] == st1["n_tracked_chunks"]


@pytest.mark.parametrize("kw", [dict(bw=9000, max_dist_x=20000, max_dist_y=20000), dict(pen_skip=np.float32(0.02), max_iter=20000)])
def test_modes_without_the_penalty_table(kw):
    """bw too large for the LDS table, or chn_pen_skip != 0: per-pair float path (MODE_FAST), also in cooperative mode."""
    a = sc.sort_by_x(np.concatenate([sc.repeat_block(5200, 61), sc.colinear(700, 62)]))
    with mm.Engine() as e:
        st = check_batch(e, a, np.array([0, len(a)], np.int64), orc.default_param(**kw))
    assert st["n_pairs"] > 1_000_000


def test_many_tiny_reads(engine):
    """50 000 reads of 0-40 anchors packed into one micro-batch: read boundaries every few anchors (several per planning block
    and per tile), reads without anchors in between."""
    rng = np.random.default_rng(99)
    sizes = rng.integers(0, 41, 50_000)
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    off[1:] = np.cumsum(sizes)
    n = int(off[-1])
    # every read: a short colinear run on one of two reference sequences + a few stray anchors, sorted by x within the read
    rid = rng.integers(0, 2, n)
    read_of = np.repeat(np.arange(len(sizes)), sizes)
    pos_in_read = np.arange(n) - off[read_of]
    x = 10_000 + pos_in_read * rng.integers(5, 30, n) + rng.integers(0, 3, n)
    y = 100 + pos_in_read * 17 + rng.integers(0, 9, n)
    a = sc.pack(rid, np.zeros(n, np.int64), x, y)
    order = np.lexsort((a[:, 0], read_of))
    a = a[order]
    st = check_batch(engine, a, off, orc.default_param(min_cnt=2, min_sc=20), threads=8)
    assert st["n_reads"] == 50_000


def test_four_wave_teams(engine):
    """Heavy chunks with narrow windows are pipelined over 4-wave teams with their share of the LDS ring each: plain chains,
    and max_iter-clamped windows (rescue state handed from wave to wave inside a team)."""
    prm = orc.default_param()
    chains = [sc.sort_by_x(np.concatenate([sc.colinear(9000, 300 + k, r0=1_000_000 + 7 * k), sc.noise(3000, 400 + k)])) for k in range(6)]
    off = np.zeros(len(chains) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in chains])
    st = check_batch(engine, np.concatenate(chains), off, prm)
    assert st["n_mid_chunks"] >= 4 and st["n_long_chunks"] == 0
    clamped = [sc.sort_by_x(np.concatenate([sc.repeat_block(6000, 500 + k), sc.colinear(500, 600 + k)])) for k in range(3)]
    clamped.append(sc.rescue_case(n_noise=5000, n_chain=60, seed=12))
    off = np.zeros(len(clamped) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in clamped])
    st = check_batch(engine, np.concatenate(clamped), off, orc.default_param(max_iter=160))
    assert st["n_mid_chunks"] >= 3 and st["n_tracked_chunks"] >= 3


@pytest.mark.parametrize("max_iter", [9000, 40000, orc.INT32_MAX])
def test_max_iter_beyond_the_lds_ring(engine, max_iter):
    """max_iter larger than the LDS score ring can hold: chunks whose widest window still fits go to teams, the others to
    single waves; results unchanged (this is also the `-x sr`-style max_iter = INT32_MAX)."""
    parts = [sc.sort_by_x(np.concatenate([sc.repeat_block(12000, 81, xwin=4500), sc.colinear(600, 82)])),   # windows up to ~12000
             sc.sort_by_x(np.concatenate([sc.repeat_block(5000, 83), sc.colinear(600, 84)]))]                # windows up to ~5000
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in parts])
    st = check_batch(engine, np.concatenate(parts), off, orc.default_param(max_iter=max_iter))
    assert st["n_long_chunks"] + st["n_mid_chunks"] >= 1


def test_inputs_outside_the_table_sweep_domain(engine):
    """The table sweep works on coordinates x4 and needs query positions < 2^27 and q_span > 0; anything else must be
    detected on the device and scored by the per-pair build, with identical results."""
    base = sc.sort_by_x(np.concatenate([sc.repeat_block(5200, 71), sc.colinear(900, 72)]))
    far = base.copy()
    far[:, 1] += np.uint64((1 << 27) + 12345)                      # query positions beyond 2^27
    zero = base.copy()
    zero[::7, 1] &= ~(np.uint64(0xff) << np.uint64(32))            # some anchors with q_span == 0
    huge_x = base.copy()
    huge_x[:, 0] += np.uint64(0x7ff00000)                          # reference positions near 2^31 (x4 wraps; differences do not)
    for a in (far, zero, huge_x):
        check_batch(engine, a, np.array([0, len(a)], np.int64), orc.default_param())


def test_synthetic_ont_batch_full_chain(engine):
    """The bench workload at small scale: 24 reads of 10-100 kb; scores, then chains, against the oracle."""
    a, off = mm.synth_reads(7, 0, 24, 10_000, 100_000)
    prm = orc.default_param()
    check_batch(engine, a, off, prm, threads=8)
    res, _ = engine.chain(a, off, threads=4)
    for r in range(len(off) - 1):
        o = orc.lchain_dp(a[off[r]:off[r + 1]], prm, want_fp=False)
        assert np.array_equal(res[r][0], o["u"]) and np.array_equal(res[r][1], o["a_out"]), f"read {r}"


def test_ultralong_reads_properties(engine):
    """configs[3]-shaped reads (100-300 kb): oracle on a sample of reads, size-independent properties on all."""
    a, off = mm.synth_reads(11, 0, 12, 100_000, 300_000)
    prm = orc.default_param()
    engine.set_misc(misc_from(prm))
    f, p, st = engine.score(a, off)
    # properties: f >= q_span; a predecessor is in the same read, same strand|rid, within max_dist_x, and within the window
    span = ((a[:, 1] >> np.uint64(32)) & np.uint64(0xff)).astype(np.int64)
    assert np.all(f >= span)
    idx = np.arange(len(a))
    has = p > 0
    j = idx - p
    read_of = np.searchsorted(off, idx, side="right") - 1
    assert np.all(j[has] >= off[read_of[has]])
    assert np.all((a[idx[has], 0] >> np.uint64(32)) == (a[j[has], 0] >> np.uint64(32)))
    assert np.all(a[idx[has], 0] - a[j[has], 0] <= 5000)
    assert np.all(f[has] > span[has]) and np.all(f[~has] == span[~has])
    # idempotence: same inputs, same outputs
    f2, p2, _ = engine.score(a, off)
    assert np.array_equal(f, f2) and np.array_equal(p, p2)
    # oracle on three reads
    for r in (0, 5, 11):
        fo, po, _ = orc.chain_fill(a[off[r]:off[r + 1]], prm)
        assert np.array_equal(f[off[r]:off[r + 1]], fo) and np.array_equal(p[off[r]:off[r + 1]], rel(po)), f"read {r}"


def test_sliced_host_call_overlapped_streams(monkeypatch):
    """mm2gb_score_host cuts a large batch into slices (H2D / kernels / D2H on three streams, two staging sets):
    same results as one piece, counters add up, staging sets are reused more than once."""
    a, off = mm.synth_reads(21, 0, 40, 10_000, 60_000)
    prm = orc.default_param()
    with mm.Engine() as e:
        f1, p1, st1 = e.score(a, off)
    monkeypatch.setenv("MM2GB_SLICE_ANCHORS", "60000")
    with mm.Engine() as e:
        f2, p2, st2 = e.score(a, off)
        f3, p3, st3 = e.score(a, off)          # second call on the same engine: slots and sets start over
    assert np.array_equal(f1, f2) and np.array_equal(p1, p2) and np.array_equal(f1, f3) and np.array_equal(p1, p3)
    assert st2["n_pairs"] == st1["n_pairs"] == st3["n_pairs"] and st2["n_anchors"] == len(a) and st2["n_reads"] == 40
    assert st2["n_chunks"] >= st1["n_chunks"]
    fo, po, pairs = orc.chain_fill_many(a, off, prm, threads=4)
    assert np.array_equal(f2, fo) and st2["n_pairs"] == pairs


def fuzz_seed():
    """Seed of the in-suite fuzz: MM2GB_FUZZ_SEED, else derived from a hash of everything under mm2-gb_amd/csrc (kernels, engine,
    host code), so the batches move whenever the product does and stay reproducible for a given tree."""
    import glob
    import hashlib
    import os
    v = os.environ.get("MM2GB_FUZZ_SEED")
    if v:
        return int(v)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(root, "mm2-gb_amd", "csrc", "*"))):
        with open(path, "rb") as fh:
            h.update(fh.read())
    return int(h.hexdigest()[:8], 16) % (2 ** 31)


def test_fuzz_random_batches_and_parameters(engine):
    """Seeded fuzz: random mixtures of chains, noise, repeat blocks, duplicated positions, empty reads, and random chaining
    parameters (window limits around tile and ring sizes, both score builds).  The seed follows the kernel sources (fuzz_seed): every
    build that changes a kernel is fuzzed with batches no earlier build has seen."""
    import os
    seed = fuzz_seed()
    print(f"fuzz seed {seed} (hash of mm2-gb_amd/csrc; MM2GB_FUZZ_SEED overrides)")
    rng = np.random.default_rng(seed)
    for it in range(int(os.environ.get("MM2GB_FUZZ_ITERS", 40))):
        a, off, kw = sc.fuzz_case(rng)
        prm = orc.default_param(**kw)
        check_batch(engine, a, off, prm, threads=2)
        if it % 8 == 0 and off[-1]:
            res, _ = engine.chain(a, off, threads=2)
            for r in range(len(off) - 1):
                o = orc.lchain_dp(a[off[r]:off[r + 1]], prm, want_fp=False)
                assert np.array_equal(res[r][0], o["u"]) and np.array_equal(res[r][1], o["a_out"]), (it, r, kw)


def test_fuzz_soak_six_engine_configurations():
    """tests/fuzz_soak.py inside the suite: one seed (from the kernel sources' hash) x 100 batches x six
    engine configurations -- default planner, every chunk on 8-wave teams, on whole-workgroup teams, on 4-wave teams, on 4-wave teams
    with windows wider than the ring share, on gangs of workgroups -- each batch against the oracle on every anchor, the device post-pass against the host
    post-pass, 2 larger bench-like batches with random parameters, and one batch of >= 20 M anchors (30-300 kb reads: team modes, gangs and
    the planner's lists under real load) -- every anchor of it against the oracle under all six configurations.  The bug that mattered in round 2 (the unchecked sweep judging a
    tile by an anchor of the next read) was found by exactly this tool."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    seed = fuzz_seed() ^ 0x5eed
    r = subprocess.run([sys.executable, os.path.join(here, "fuzz_soak.py"), str(seed), "--iters", os.environ.get("MM2GB_SOAK_ITERS", "100"), "--teams", "--post", "--big", "2", "--huge", "1"],
                       capture_output=True, text=True, timeout=1500)
    print(r.stdout[-1500:])
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert f"seed {seed} clean over" in r.stdout


def test_large_batch_properties_and_mode_agreement(monkeypatch):
    """Bench-shaped batch at scale (40 M anchors, ~2.4e10 pairs; the oracle would need minutes): size-independent
    properties, idempotence, and agreement of independent code paths -- team modes vs one-wave-per-chunk, table sweep vs
    per-pair arithmetic -- through checksums of f and p."""
    a, off = mm.synth_reads(2024, 0, 720, 100_000, 300_000, threads=32)
    n = len(a)
    assert n > 30_000_000

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with mm.Engine() as e:
            f, p, st = e.score(a, off)
        for k in env:
            monkeypatch.delenv(k)
        return f, p, st

    f, p, st = run({})
    assert st["n_long_chunks"] > 100 and st["n_pairs"] > 10_000_000_000
    span = ((a[:, 1] >> np.uint64(32)) & np.uint64(0xff)).astype(np.int32)
    has = p > 0
    assert np.all(f >= span) and np.all(f[~has] == span[~has]) and np.all(f[has] > span[has])
    idx = np.arange(n, dtype=np.int64)
    j = idx - p
    read_of = np.searchsorted(off, idx, side="right") - 1
    assert np.all(j[has] >= off[read_of[has]])
    assert np.all((a[idx[has], 0] >> np.uint64(32)) == (a[j[has], 0] >> np.uint64(32)))
    assert np.all(a[idx[has], 0] - a[j[has], 0] <= 5000)
    # a predecessor's score plus at most q_span reaches the successor's score: f[i] <= f[j] + span[j]
    assert np.all(f[has] <= f[j[has]] + span[j[has]])
    ck = (int(f.astype(np.int64).sum()), int(p.astype(np.int64).sum()), int((f.astype(np.int64) * (idx % 1009)).sum()))
    # (the last two: wide-window chunks on 4-wave teams, which take the scores their ring share no longer holds from global memory --
    # the rule of batches from 250 M anchors on, here for all chunks but the dominant ones and for all chunks)
    for env in ({"MM2GB_NO_COOP": "1"}, {"MM2GB_WIDE_WINDOW": "100000000"}, {"MM2GB_LONG_MIN_COST": "50000000"},
                {"MM2GB_TEAM4_MIN_ANCHORS": "0"}, {"MM2GB_TEAM4_ALL": "1"}):
        f2, p2, st2 = run(env)
        assert st2["n_pairs"] == st["n_pairs"]
        ck2 = (int(f2.astype(np.int64).sum()), int(p2.astype(np.int64).sum()), int((f2.astype(np.int64) * (idx % 1009)).sum()))
        assert ck2 == ck, env
        assert np.array_equal(f2, f) and np.array_equal(p2, p), env
    # per-pair float build (chn_pen_skip tiny but non-zero disables the table): scores may differ only where the skip term
    # changes a truncation, so compare against the same build without teams instead
    misc = mm.default_misc(chn_pen_skip=np.float32(1e-9))
    with mm.Engine(misc=misc) as e:
        f3, p3, _ = e.score(a, off)
    monkeypatch.setenv("MM2GB_NO_COOP", "1")
    with mm.Engine(misc=misc) as e:
        f4, p4, _ = e.score(a, off)
    monkeypatch.delenv("MM2GB_NO_COOP")
    assert np.array_equal(f3, f4) and np.array_equal(p3, p4)
    # oracle on a few reads of this batch
    prm = orc.default_param()
    for r in (0, 359, 719):
        fo, po, _ = orc.chain_fill(a[off[r]:off[r + 1]], prm)
        assert np.array_equal(f[off[r]:off[r + 1]], fo) and np.array_equal(p[off[r]:off[r + 1]], rel(po))


def test_bench_size_batch_properties():
    """The micro-batch the bench is quoted on (BASELINE configs[3]: 100-300 kb reads, 500 M anchors, 2.7e11 pairs): size-independent
    properties of f and p on all of it, idempotence, then EVERY anchor's f and p against the oracle (all host threads, ~90 s) and every
    read's chains from the device post-pass against the host post-pass.
    Set MM2GB_TEST_FULL_ANCHORS to run a smaller batch on a machine with less host memory."""
    import os
    target = int(os.environ.get("MM2GB_TEST_FULL_ANCHORS", 500_000_000))
    import bench
    _, n_reads, a, off = bench.shard_for_rank(mm, 0, 1, 2024, target, 100_000, 300_000, threads=32)
    n = len(a)
    assert n >= 0.99 * target
    with mm.Engine() as e:
        f, p, st = e.score(a, off)
        assert st["n_anchors"] == n and st["n_reads"] == n_reads and st["n_pairs"] > 400 * n
        span = ((a[:, 1] >> np.uint64(32)) & np.uint64(0xff)).astype(np.int32)
        assert (f >= span).all()
        has = p > 0
        assert (f[has] > span[has]).all() and (f[~has] == span[~has]).all()
        # predecessors: inside the read, same strand | rid, within max_dist_x, at most max_iter back (or the rescue: further, but in reach)
        idx = np.flatnonzero(has)
        j = idx - p[idx]
        read_of = np.searchsorted(off, idx, side="right") - 1
        assert (j >= off[read_of]).all()
        assert ((a[idx, 0] >> np.uint64(32)) == (a[j, 0] >> np.uint64(32))).all()
        assert (a[idx, 0] - a[j, 0] <= np.uint64(5000)).all()
        del idx, j, read_of, has
        # same inputs, same outputs (checksums: xor-folded 64-bit sums are order independent, enough for "identical arrays")
        f2, p2, st2 = e.score(a, off)
        assert st2["n_pairs"] == st["n_pairs"] and np.array_equal(f, f2) and np.array_equal(p, p2)
        # backtrack + compaction of the whole batch: device post-pass == host post-pass, every read
        assert device_post_pass_against_the_host_post_pass(e, a, off) > n_reads
    # and EVERY anchor of the batch the headline number is quoted on against the oracle
    every_anchor_against_the_oracle(a, off, f, p, orc.default_param(), st["n_pairs"])


def test_config2_size_batch_properties():
    """BASELINE configs[2] at its full size: synthetic ONT 10-100 kb reads, one 500 M-anchor micro-batch.  Same checks as the
    configs[3] test above: size-independent properties on every anchor, idempotence, every anchor against the oracle, every read's
    chains from the device post-pass against the host post-pass."""
    import os
    target = int(os.environ.get("MM2GB_TEST_FULL_ANCHORS", 500_000_000))
    import bench
    _, n_reads, a, off = bench.shard_for_rank(mm, 0, 1, 2024, target, 10_000, 100_000, threads=32)
    n = len(a)
    assert n >= 0.99 * target
    with mm.Engine() as e:
        f, p, st = e.score(a, off)
        assert st["n_anchors"] == n and st["n_reads"] == n_reads and st["n_pairs"] > 100 * n
        span = ((a[:, 1] >> np.uint64(32)) & np.uint64(0xff)).astype(np.int32)
        assert (f >= span).all()
        has = p > 0
        assert (f[has] > span[has]).all() and (f[~has] == span[~has]).all()
        idx = np.flatnonzero(has)
        j = idx - p[idx]
        read_of = np.searchsorted(off, idx, side="right") - 1
        assert (j >= off[read_of]).all()
        assert ((a[idx, 0] >> np.uint64(32)) == (a[j, 0] >> np.uint64(32))).all()
        assert (a[idx, 0] - a[j, 0] <= np.uint64(5000)).all()
        assert (f[idx] <= f[j] + span[j]).all()
        del idx, j, read_of, has
        f2, p2, st2 = e.score(a, off)
        assert st2["n_pairs"] == st["n_pairs"] and np.array_equal(f, f2) and np.array_equal(p, p2)
        assert device_post_pass_against_the_host_post_pass(e, a, off) > n_reads
    every_anchor_against_the_oracle(a, off, f, p, orc.default_param(), st["n_pairs"])


def band_cloud(n, seed, xwin, jitter, r0=3_000_000, q0=20_000, spans=None):
    """Dense cloud around one diagonal: reference positions uniform in a window, query position = the diagonal +- jitter.  Among the
    pairs inside a window every case the range test exists for occurs in numbers: dq <= 0 with |dr - dq| <= bw (sources just left of a
    target but above it on the query), dq > max_dist_y with dr far out, and dq > 0 with |dr - dq| beyond the penalty table."""
    rng = np.random.default_rng(seed)
    x = r0 + rng.integers(0, xwin, n)
    y = q0 + (x - r0) + rng.integers(-jitter, jitter + 1, n)
    qspan = 15 if spans is None else rng.choice(np.asarray(spans), n)
    return sc.sort_by_x(sc.pack(np.full(n, 5), np.zeros(n, np.int64), x, y, qspan=qspan))


@pytest.mark.parametrize("kw", [dict(), dict(max_dist_y=3000), dict(max_dist_x=3000, max_dist_y=4500), dict(bw=2000), dict(bw=2400, max_dist_y=4900),
                                dict(max_dist_x=1200, max_dist_y=1200, bw=100), dict(max_iter=900)],
                         ids=["defaults", "dist_y_3000", "dist_x_3000", "bw_2000", "bw_near_half_dist", "dist_1200", "iter_900"])
def test_unchecked_sweep_rejects_what_the_range_test_would(monkeypatch, kw):
    """The table sweep leaves out the test 0 < dq <= min(max_dist_x, max_dist_y) for source blocks at most dq_lim - bw bases left of
    their targets: there a gather beyond the penalty table (= beyond the workgroup's LDS, which reads 0) or a saturated table address
    (dq <= 0) rejects the pair instead (chain_kernels.hip, sweep_block_lut2_free).  Dense clouds where such pairs abound, under
    parameter sets that move every bound; with the unchecked build switched off (MM2GB_FREE_SWEEP=0) the results are the same."""
    parts = [band_cloud(9000, 301, xwin=9000, jitter=700), band_cloud(7000, 302, xwin=3500, jitter=6500, r0=5_000_000),
             band_cloud(6000, 303, xwin=12000, jitter=250, r0=7_000_000), sc.read_like(9000, 304),
             # spans of every size (the FAR build of the unchecked sweep takes min(q_span, dr, dq) = q_span where dr >= bw + q_span), and
             # query positions far above the reference positions (negative diagonals in its table address)
             band_cloud(9000, 307, xwin=10000, jitter=600, r0=9_000_000, spans=[1, 10, 15, 19, 60, 200, 255]),
             band_cloud(8000, 308, xwin=9000, jitter=900, r0=20_000, q0=3_000_000, spans=[15, 21, 128]),
             sc.sort_by_x(np.concatenate([sc.repeat_block(7000, 305), sc.colinear(600, 306)]))]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in parts])
    a = np.concatenate(parts)
    prm = orc.default_param(**kw)
    with mm.Engine() as e:
        st = check_batch(e, a, off, prm)
        assert st["n_long_chunks"] + st["n_mid_chunks"] >= 1          # team modes (two tiles per wave) took part
        f1, p1, _ = e.score(a, off)
    monkeypatch.setenv("MM2GB_FREE_SWEEP", "0")
    with mm.Engine() as e:
        check_batch(e, a, off, prm)
        f0, p0, _ = e.score(a, off)
    assert np.array_equal(f0, f1) and np.array_equal(p0, p1)
    monkeypatch.delenv("MM2GB_FREE_SWEEP")
    monkeypatch.setenv("MM2GB_NO_COOP", "1")                          # every chunk by one wave
    with mm.Engine() as e:
        check_batch(e, a, off, prm)


def test_edge_blocks_with_the_window_test_from_a_prefix_mask(monkeypatch):
    """MM2GB_EDGE=new (off by default: profiles/r06_narrow_ab.txt): edge blocks of tiles whose window starts rise from lane to lane take "source inside this
    target's window" from a scalar prefix mask per source instead of a vector compare (chain_kernels.hip, sweep_block_lut_edge_sorted).  Narrow windows --
    every block of a 60-anchor window is an edge block --, several reads per chunk (starts jump), runs of equal positions, and the dense clouds of the
    test above: same f / p as the default build, and both equal the oracle's."""
    a1, o1 = mm.synth_reads(71, 0, 60, 10_000, 30_000)
    parts = [a1[o1[r]:o1[r + 1]] for r in range(60)]
    parts += [band_cloud(9000, 321, xwin=9000, jitter=700), band_cloud(6000, 323, xwin=900, jitter=250, r0=7_000_000), sc.read_like(9000, 324),
              sc.sort_by_x(np.concatenate([sc.repeat_block(5000, 325), sc.colinear(600, 326)]))]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in parts])
    a = np.concatenate(parts)
    prm = orc.default_param()
    with mm.Engine() as e:
        check_batch(e, a, off, prm)
        f0, p0, _ = e.score(a, off)
    monkeypatch.setenv("MM2GB_EDGE", "new")
    for no_coop in (False, True):
        if no_coop:
            monkeypatch.setenv("MM2GB_NO_COOP", "1")                  # every chunk by one wave
        with mm.Engine() as e:
            check_batch(e, a, off, prm)
            f1, p1, _ = e.score(a, off)
        assert np.array_equal(f0, f1) and np.array_equal(p0, p1)


def test_without_the_lds_contract_the_checked_builds_run(monkeypatch):
    """Engine::init probes what the unchecked sweeps rely on (reads beyond a workgroup's LDS return 0, v_sad_u32 clamp saturates).
    MM2GB_LDS_PROBE=0 stands for a device where the probe fails: clamped table, every range test, same results."""
    monkeypatch.setenv("MM2GB_LDS_PROBE", "0")
    parts = [band_cloud(8000, 311, xwin=8000, jitter=900), sc.read_like(7000, 312)]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in parts])
    with mm.Engine() as e:
        check_batch(e, np.concatenate(parts), off, orc.default_param())


@pytest.mark.timeout(300)
def test_one_chunk_on_several_workgroups(monkeypatch):
    """A batch too small to fill the GPU ends with its largest chunk, which one workgroup scores at the pace of one CU.  In micro-batches
    of up to MM2GB_SPLIT_MAX_ANCHORS anchors (0 = never, the default: the build is exact but measured slower) such chunks
    are scored strip by strip (1 024 anchors), the sweeps over the sources before a strip cut into items that idle workgroups take
    (chain_kernels.hip, split_chunk).  Heavy chunks of every kind -- windows cut by max_iter (the rescue state machine runs), wide and
    narrow windows, a chunk that ends inside a strip, several owners at once -- against the oracle, every chunk of the big-team list
    split (MM2GB_WHOLE_WG_PCT=1), and against the same batch with the build switched off."""
    if not mm.lib().mm2gb_has_split_build():
        pytest.skip("the SPLIT instantiation of k_score is a build option (make -C mm2-gb_amd SPLIT=1): exact, measured slower, not in the default library")
    monkeypatch.setenv("MM2GB_SPLIT_MAX_ANCHORS", "200000000")
    parts = [sc.sort_by_x(np.concatenate([sc.repeat_block(23000, 401, xwin=4500, ywin=7000), sc.colinear(900, 402)])),
             sc.sort_by_x(sc.repeat_block(9000, 403, xwin=9000, ywin=9000, r0=4_000_000)),
             band_cloud(12345, 404, xwin=11000, jitter=800), sc.read_like(9000, 405),
             sc.rescue_case(n_noise=9000, n_chain=60, seed=23),
             sc.sort_by_x(sc.repeat_block(17000 + 64 * 3 + 7, 406, xwin=3000, ywin=4000, r0=6_000_000))]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in parts])
    a = np.concatenate(parts)
    prm = orc.default_param()
    with mm.Engine() as e:
        st = check_batch(e, a, off, prm)
        f1, p1, _ = e.score(a, off)
        chunks, helped = e.split_counts()
        assert chunks >= 3 and helped > 0, (chunks, helped, st)
        for _ in range(5):                                            # who takes which item changes from run to run; the results must not
            f2, p2, _ = e.score(a, off)
            assert np.array_equal(f1, f2) and np.array_equal(p1, p2)
        check_batch(e, a, off, orc.default_param(max_iter=700))
        check_batch(e, a, off, orc.default_param(max_dist_x=2000, max_dist_y=2000, bw=300))
    monkeypatch.setenv("MM2GB_WHOLE_WG_PCT", "1")
    with mm.Engine() as e:
        for _ in range(3):
            check_batch(e, a, off, prm)
    monkeypatch.delenv("MM2GB_WHOLE_WG_PCT")
    monkeypatch.delenv("MM2GB_SPLIT_MAX_ANCHORS")
    with mm.Engine() as e:
        f0, p0, _ = e.score(a, off)
        assert e.split_counts() == (0, 0)
    assert np.array_equal(f0, f1) and np.array_equal(p0, p1)


@pytest.mark.timeout(300)
def test_gangs_several_workgroups_on_one_chunk(monkeypatch):
    """A batch too small to fill the GPU ends with its largest chunks, and a team is bounded by one CU: a chunk whose share of the batch's
    pairs is worth two workgroups or more is cut into strips of 16 tile pairs that several workgroups take in turn, scores travelling
    through global memory (chain_kernels.hip, gang_chunk_pairs; plan_gangs).  Heavy chunks of every kind -- windows cut by max_iter
    (the rescue state crosses workgroups), wide and narrow windows, a chunk that ends inside a strip, several gangs at once -- against
    the oracle, with the default gang size, gangs of 2 and of 64, and against the same batch with gangs switched off."""
    parts = [sc.sort_by_x(np.concatenate([sc.repeat_block(23000, 401, xwin=4500, ywin=7000), sc.colinear(900, 402)])),
             sc.sort_by_x(sc.repeat_block(9000, 403, xwin=9000, ywin=9000, r0=4_000_000)),
             band_cloud(12345, 404, xwin=11000, jitter=800), sc.read_like(9000, 405),
             sc.rescue_case(n_noise=9000, n_chain=60, seed=23),
             sc.sort_by_x(sc.repeat_block(17000 + 64 * 3 + 7, 406, xwin=3000, ywin=4000, r0=6_000_000))]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in parts])
    a = np.concatenate(parts)
    prm = orc.default_param()
    results = []
    for gang_max, pct in (("8", "100"), ("2", "100"), ("64", "400")):
        monkeypatch.setenv("MM2GB_GANG_MAX", gang_max)
        monkeypatch.setenv("MM2GB_GANG_PCT", pct)
        with mm.Engine() as e:
            st = check_batch(e, a, off, prm)
            f1, p1, _ = e.score(a, off)
            chunks, wgs = e.gang_counts()
            assert chunks >= 2 and wgs >= 2 * chunks, (chunks, wgs, st)
            for _ in range(5):                                        # which workgroup takes which strip changes from run to run; the results must not
                f2, p2, _ = e.score(a, off)
                assert np.array_equal(f1, f2) and np.array_equal(p1, p2)
            check_batch(e, a, off, orc.default_param(max_iter=700))
            check_batch(e, a, off, orc.default_param(max_dist_x=2000, max_dist_y=2000, bw=300))
            check_batch(e, a, off, orc.default_param(max_iter=12000))   # windows wider than the LDS ring
            results.append((f1, p1))
    monkeypatch.setenv("MM2GB_GANG_MAX", "0")
    with mm.Engine() as e:
        f0, p0, _ = e.score(a, off)
        assert e.gang_counts() == (0, 0)
    for f1, p1 in results:
        assert np.array_equal(f0, f1) and np.array_equal(p0, p1)


def test_clamped_penalty_table_build(monkeypatch):
    """MM2GB_LUT_CLAMP=1: the bw+2-entry penalty table with a clamped index (no LDS read ever leaves the table) instead of
    the wide unclamped one.  Same results on saturated windows, ties, the rescue, team and wave modes."""
    monkeypatch.setenv("MM2GB_LUT_CLAMP", "1")
    parts = [sc.sort_by_x(np.concatenate([sc.repeat_block(7000, 141), sc.colinear(800, 142), sc.noise(3000, 143)])),
             sc.rescue_case(n_noise=6000, n_chain=50, seed=19), sc.read_like(9000, 144),
             sc.sort_by_x(sc.repeat_block(6000, 177, xwin=1500, ywin=5000))]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in parts])
    a = np.concatenate(parts)
    with mm.Engine() as e:
        st = check_batch(e, a, off, orc.default_param())
        assert st["n_long_chunks"] >= 2 and st["n_tracked_chunks"] >= 1
        check_batch(e, a, off, orc.default_param(max_iter=160))
        for path in [c for c in CASES if "ties" in c or "rescue" in c]:
            g = golden_io.load(path)
            if g["prm"].max_skip != orc.INT32_MAX:
                continue
            e.set_misc(misc_from(g["prm"]))
            f, p, _ = e.score(g["a"], np.array([0, len(g["a"])], np.int64))
            assert np.array_equal(f, g["f"]) and np.array_equal(p, rel(g["p"])), path
    monkeypatch.setenv("MM2GB_NO_COOP", "1")
    with mm.Engine() as e:
        check_batch(e, a, off, orc.default_param())


@pytest.mark.parametrize("dist", [(1 << 28) - 1, 1 << 28, 1 << 29, (1 << 31) - 1])
def test_huge_max_dist_leaves_the_table_sweep_domain(dist):
    """A user -g / -r of 2^28 and more (max_dist_x / max_dist_y): the x4 coordinates of the table sweep would wrap, so the
    engine must pick the per-pair build by itself; windows are then bounded by max_iter alone."""
    a = sc.sort_by_x(np.concatenate([sc.repeat_block(3000, 151), sc.colinear(900, 152), sc.noise(2000, 153)]))
    with mm.Engine() as e:
        check_batch(e, a, np.array([0, len(a)], np.int64), orc.default_param(max_dist_x=dist, max_dist_y=dist, max_iter=700))
        check_batch(e, a, np.array([0, len(a)], np.int64), orc.default_param(max_dist_x=dist, max_dist_y=5000))


@pytest.mark.timeout(120)
def test_unsorted_anchors_do_not_hang_or_crash(engine):
    """Anchors sorted by x are the caller's contract (map.c:329).  Broken input must still come back: window starts stay inside
    [read start, i] by construction, and no wait in the kernel depends on the data.  The values are unspecified."""
    rng = np.random.default_rng(5)
    parts = [sc.sort_by_x(np.concatenate([sc.repeat_block(6000, 91), sc.colinear(500, 92)])), sc.read_like(8000, 93)]
    a = np.concatenate(parts)
    off = np.array([0, len(parts[0]), len(a)], np.int64)
    shuffled = a.copy()
    rng.shuffle(shuffled[: len(parts[0])])                  # first read in random order, second untouched
    engine.set_misc(mm.default_misc())
    f, p, st = engine.score(shuffled, off)
    assert len(f) == len(a) and st["n_anchors"] == len(a)
    idx = np.arange(len(a))
    assert ((p >= 0) & (p <= idx - off[np.searchsorted(off, idx, side="right") - 1])).all()      # predecessors stay inside the read
    # the untouched read is unaffected by its neighbour
    fo, po, _ = orc.chain_fill(parts[1], orc.default_param())
    assert np.array_equal(f[off[1]:], fo) and np.array_equal(p[off[1]:], rel(po))


def test_lchain_dp_signature_entry():
    """mm2gb_lchain_dp: same call shape as mg_lchain_dp (lchain.c:148), input consumed, outputs malloc'd."""
    import ctypes as C
    g = golden_io.load([p for p in CASES if p.endswith("real_mt_inf_0.npz")][0])
    L = mm.lib()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    a = np.ascontiguousarray(g["a"])
    buf = libc.malloc(a.nbytes)
    C.memmove(buf, a.ctypes.data, a.nbytes)
    prm = g["prm"]
    n_u = C.c_int(0)
    u_ptr = C.c_void_p(0)
    out = L.mm2gb_lchain_dp(prm.max_dist_x, prm.max_dist_y, prm.bw, prm.max_skip, prm.max_iter, prm.min_cnt, prm.min_sc,
                            prm.pen_gap, prm.pen_skip, prm.is_cdna, prm.n_seg, len(a), buf, C.byref(n_u), C.byref(u_ptr), None)
    assert n_u.value == len(g["u"])
    u = np.ctypeslib.as_array(C.cast(u_ptr, C.POINTER(C.c_uint64)), shape=(n_u.value,)).copy()
    a_out = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(len(g["a_out"]), 2)).copy()
    assert np.array_equal(u, g["u"]) and np.array_equal(a_out, g["a_out"])
    libc.free(u_ptr)
    libc.free(out)
