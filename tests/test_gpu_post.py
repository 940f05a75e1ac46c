"""backtrack + compaction on the device (mm2gb_chain_gpu, csrc/post_kernels.hip) against the committed
reference vectors, the CPU oracle and the host post-pass -- chains u[] and compacted anchors bit for bit, including the
order radix_sort_128x leaves among equal scores (which decides what chain an anchor ends up in)."""
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


@pytest.fixture(scope="module")
def engine():
    with mm.Engine() as e:
        yield e


def check_against_oracle(engine, a, off, prm):
    engine.set_misc(misc_from(prm))
    res, st = engine.chain_gpu(a, off)
    assert len(res) == len(off) - 1
    for r in range(len(off) - 1):
        o = orc.lchain_dp(a[off[r]:off[r + 1]], prm, want_fp=False)
        assert np.array_equal(res[r][0], o["u"]), f"read {r}: chains differ ({len(res[r][0])} vs {len(o['u'])})"
        assert np.array_equal(res[r][1], o["a_out"]), f"read {r}: compacted anchors differ"
    return st


@pytest.mark.parametrize("path", [p for p in CASES], ids=golden_io.case_ids(CASES))
def test_reference_vectors(engine, path):
    g = golden_io.load(path)
    prm = g["prm"]
    if prm.max_skip != orc.INT32_MAX:
        pytest.skip("recorded with a finite max_skip; the GPU path is exhaustive by contract")
    a = g["a"]
    engine.set_misc(misc_from(prm))
    res, _ = engine.chain_gpu(a, np.array([0, len(a)], dtype=np.int64))
    assert np.array_equal(res[0][0], g["u"]) and np.array_equal(res[0][1], g["a_out"])


def test_empty_and_chainless_reads(engine):
    prm = orc.default_param()
    engine.set_misc(misc_from(prm))
    res, _ = engine.chain_gpu(np.zeros((0, 2), np.uint64), np.array([0], np.int64))
    assert res == []
    res, _ = engine.chain_gpu(np.zeros((0, 2), np.uint64), np.array([0, 0, 0], np.int64))
    assert all(len(u) == 0 and len(a) == 0 for u, a in res)
    # reads that score but never reach min_cnt / min_sc, between reads that chain, and empty reads in between
    r = sc.read_like(5000, 9)
    lone = sc.noise(40, 3)
    a = np.concatenate([lone, r, lone[:1], r])
    off = np.array([0, 0, len(lone), len(lone) + len(r), len(lone) + len(r), len(lone) + len(r) + 1, len(a)], dtype=np.int64)
    check_against_oracle(engine, a, off, prm)


def test_synthetic_ont_batch(engine):
    a, off = mm.synth_reads(7, 0, 24, 10_000, 100_000)
    st = check_against_oracle(engine, a, off, orc.default_param())
    assert st["ms_post"] > 0


def test_equal_scores_follow_the_hosts_sort_order(engine):
    """Grids and repeat blocks give thousands of chain ends with equal scores; which of them is visited first comes from the
    in-place radix passes (ksort.h:116-146).  Runs of every size class: <= 64 (insertion sort), > 64 (passes), several levels."""
    prm = orc.default_param()
    parts = [sc.grid_ties(), sc.grid_ties(nx=90, ny=40, step=17), sc.sort_by_x(sc.repeat_block(9000, 5, xwin=2500, ywin=3000)),
             sc.sort_by_x(np.concatenate([sc.repeat_block(3000, 6, xwin=300, ywin=300), sc.colinear(4000, 7)])),
             sc.sort_by_x(sc.repeat_block(70, 8, xwin=200, ywin=200)), sc.sort_by_x(sc.repeat_block(64, 9, xwin=200, ywin=200)),
             sc.sort_by_x(sc.repeat_block(65, 10, xwin=200, ywin=200))]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in parts])
    a = np.concatenate(parts)
    for kw in (dict(), dict(min_cnt=1, min_sc=15), dict(min_cnt=2, min_sc=1), dict(bw=100), dict(is_cdna=1)):
        check_against_oracle(engine, a, off, orc.default_param(**kw))


def test_many_tiny_reads(engine):
    rng = np.random.default_rng(99)
    sizes = rng.integers(0, 41, 20_000)
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    off[1:] = np.cumsum(sizes)
    n = int(off[-1])
    read_of = np.repeat(np.arange(len(sizes)), sizes)
    pos_in_read = np.arange(n) - off[read_of]
    x = 10_000 + pos_in_read * rng.integers(5, 30, n) + rng.integers(0, 3, n)
    y = 100 + pos_in_read * 17 + rng.integers(0, 9, n)
    a = sc.pack(rng.integers(0, 2, n), np.zeros(n, np.int64), x, y)
    a = a[np.lexsort((a[:, 0], read_of))]
    prm = orc.default_param(min_cnt=2, min_sc=20)
    engine.set_misc(misc_from(prm))
    res, _ = engine.chain_gpu(a, off)
    host, _ = engine.chain(a, off, threads=4)          # the host post-pass, itself checked against the oracle elsewhere
    for r in range(len(sizes)):
        assert np.array_equal(res[r][0], host[r][0]) and np.array_equal(res[r][1], host[r][1]), f"read {r}"
    for r in range(0, len(sizes), 997):
        o = orc.lchain_dp(a[off[r]:off[r + 1]], prm, want_fp=False)
        assert np.array_equal(res[r][0], o["u"]) and np.array_equal(res[r][1], o["a_out"])


def test_ultralong_reads_against_host_post_pass_and_oracle(engine):
    """configs[3]-shaped reads (100-300 kb, tens of thousands of candidates per read, several radix levels)."""
    a, off = mm.synth_reads(11, 0, 16, 100_000, 300_000)
    prm = orc.default_param()
    engine.set_misc(misc_from(prm))
    res, st = engine.chain_gpu(a, off)
    host, _ = engine.chain(a, off, threads=8)
    for r in range(len(off) - 1):
        assert np.array_equal(res[r][0], host[r][0]) and np.array_equal(res[r][1], host[r][1]), f"read {r}"
    for r in (0, 7, 15):
        o = orc.lchain_dp(a[off[r]:off[r + 1]], prm, want_fp=False)
        assert np.array_equal(res[r][0], o["u"]) and np.array_equal(res[r][1], o["a_out"])


@pytest.mark.parametrize("lens,n_reads", [((100_000, 300_000), 600), ((10_000, 100_000), 2500)], ids=["config3_reads", "config2_reads"])
def test_large_batches_device_post_equals_host_post(engine, lens, n_reads):
    """Tens of millions of anchors, every size class of read, the largest reads of the bench mix: the chains of the device post-pass
    against the host post-pass (itself checked against the oracle and the reference vectors), every read, every element."""
    a, off = mm.synth_reads(23, 0, n_reads, lens[0], lens[1], threads=16)
    engine.set_misc(misc_from(orc.default_param()))
    res, st = engine.chain_gpu(a, off)
    host, _ = engine.chain(a, off, threads=16)
    assert len(res) == len(host) == n_reads
    n_chains = 0
    for r in range(n_reads):
        assert np.array_equal(res[r][0], host[r][0]) and np.array_equal(res[r][1], host[r][1]), f"read {r}"
        n_chains += len(res[r][0])
    assert n_chains > n_reads


def test_fuzz(engine):
    import os
    rng = np.random.default_rng(int(os.environ.get("MM2GB_FUZZ_SEED", 777)))
    for it in range(int(os.environ.get("MM2GB_FUZZ_ITERS", 25))):
        a, off, kw = sc.fuzz_case(rng)
        check_against_oracle(engine, a, off, orc.default_param(**kw))


def test_repeated_calls_reuse_arenas(engine):
    a1, off1 = mm.synth_reads(5, 0, 6, 10_000, 40_000)
    a2, off2 = mm.synth_reads(5, 6, 9, 10_000, 60_000)
    prm = orc.default_param()
    engine.set_misc(misc_from(prm))
    first, _ = engine.chain_gpu(a1, off1)
    engine.chain_gpu(a2, off2)
    again, _ = engine.chain_gpu(a1, off1)
    for r in range(len(off1) - 1):
        assert np.array_equal(first[r][0], again[r][0]) and np.array_equal(first[r][1], again[r][1])


def test_sliced_call_overlaps_copies_and_kernels_same_chains(engine, monkeypatch):
    """A large batch goes into mm2gb_chain_gpu in slices of reads (the score kernels of slice k under the H2D of slice k+1, one post-pass
    over the whole batch at the end, results in page-locked blocks of the result cache): forced here with a tiny slice size.  Same chains as
    the host post-pass read by read, same pair count, and a second call (which gets the first call's blocks back from the cache) the same again."""
    a, off = mm.synth_reads(31, 0, 40, 8_000, 40_000)
    want, st_h = engine.chain(a, off, threads=4)
    monkeypatch.setenv("MM2GB_CHAIN_SLICE_ANCHORS", str(max(50_000, len(a) // 7)))
    for _ in range(2):
        got, st = engine.chain_gpu(a, off)
        assert st["n_pairs"] == st_h["n_pairs"] and st["n_anchors"] == len(a)
        assert len(got) == len(want)
        for r in range(len(want)):
            assert np.array_equal(got[r][0], want[r][0]) and np.array_equal(got[r][1], want[r][1]), r
    monkeypatch.setenv("MM2GB_CHAIN_SLICE_ANCHORS", "1000")       # a slice per read or so, two result sets alternating many times
    got, _ = engine.chain_gpu(a, off)
    for r in range(len(want)):
        assert np.array_equal(got[r][0], want[r][0]) and np.array_equal(got[r][1], want[r][1]), r
    # a batch with empty reads, a read without chains and a last slice of one short read, sliced per read
    parts = [a[off[r]:off[r + 1]] for r in range(6)]
    parts = [parts[0], parts[1][:0], sc.noise(50, 5), parts[2], parts[3][:0], parts[4], parts[5][:3]]
    off2 = np.zeros(len(parts) + 1, dtype=np.int64)
    off2[1:] = np.cumsum([len(x) for x in parts])
    a2 = np.concatenate(parts)
    want2, _ = engine.chain(a2, off2, threads=2)
    got2, _ = engine.chain_gpu(a2, off2)
    for r in range(len(parts)):
        assert np.array_equal(got2[r][0], want2[r][0]) and np.array_equal(got2[r][1], want2[r][1]), r


def test_a_further_engine_on_the_device_has_two_streams_and_the_same_chains(engine, monkeypatch):
    """Every engine but the first alive on a device gets two HIP streams instead of four (kernels and H2D on one, D2H on the other:
    Engine::init, profiles/r05_hw_queues.txt).  The host-buffer paths are written for four -- copies in on one stream, kernels on another,
    events between them -- and must give the same chains when two of those streams are one: the sliced call and the plain one on a second
    engine, against the first engine's host post-pass."""
    a, off = mm.synth_reads(33, 0, 30, 8_000, 40_000)
    want, st_h = engine.chain(a, off, threads=4)
    second = mm.Engine(device=engine.device)
    try:
        for slice_anchors in (str(max(50_000, len(a) // 5)), "1000", None):
            if slice_anchors is None:
                monkeypatch.delenv("MM2GB_CHAIN_SLICE_ANCHORS", raising=False)
            else:
                monkeypatch.setenv("MM2GB_CHAIN_SLICE_ANCHORS", slice_anchors)
            got, st = second.chain_gpu(a, off)
            assert st["n_pairs"] == st_h["n_pairs"] and len(got) == len(want)
            for r in range(len(want)):
                assert np.array_equal(got[r][0], want[r][0]) and np.array_equal(got[r][1], want[r][1]), (slice_anchors, r)
        got, _ = second.chain(a, off, threads=4)                # scores through the host buffers, host post-pass
        for r in range(len(want)):
            assert np.array_equal(got[r][0], want[r][0]) and np.array_equal(got[r][1], want[r][1]), r
    finally:
        second.close()


def test_post_pass_of_one_engine_beside_the_scores_of_another(engine):
    """mm2gb_post_device_enqueue / _totals: the post-pass of a scored batch enqueued on a SECOND engine (its own stream) while the first engine
    scores again into another pair of arrays -- what bench.py's through_backtrace_pipelined measures.  Same totals as the synchronous call
    on the first engine, twice in a row (the second engine's arenas are reused)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    H2D, D2H = 1, 2

    def dev(nbytes, src=None):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), max(nbytes, 16)) == 0
        if src is not None:
            assert hip.hipMemcpy(p, src.ctypes.data, nbytes, H2D) == 0
        return p

    a, off = mm.synth_reads(41, 0, 24, 8_000, 50_000)
    a = np.ascontiguousarray(a)
    n, nr = len(a), len(off) - 1
    d_a, d_off = dev(n * 16, a), dev((nr + 1) * 8, off)
    fp = [(dev(n * 4), dev(n * 4)) for _ in range(2)]
    L = mm.lib()
    try:
        engine.score_device(nr, d_off.value, d_a.value, n, fp[0][0].value, fp[0][1].value)
        engine.sync()
        n_ch, n_kept, ms = C.c_int64(0), C.c_int64(0), C.c_float(0)
        assert L.mm2gb_post_device(engine._h, nr, d_off, d_a, n, fp[0][0], fp[0][1], C.byref(n_ch), C.byref(n_kept), C.byref(ms)) == 0
        want_chains, _ = engine.chain(a, off, threads=2)
        assert n_ch.value == sum(len(u) for u, _ in want_chains) and n_kept.value == sum(len(x) for _, x in want_chains)
        with mm.Engine(device=0) as eng_b:
            for k in (1, 2):
                assert L.mm2gb_post_device_enqueue(eng_b._h, nr, d_off, d_a, n, fp[(k - 1) & 1][0], fp[(k - 1) & 1][1]) == 0
                engine.score_device(nr, d_off.value, d_a.value, n, fp[k & 1][0].value, fp[k & 1][1].value)
                engine.sync()
                c2, k2, m2 = C.c_int64(0), C.c_int64(0), C.c_float(0)
                assert L.mm2gb_post_device_totals(eng_b._h, C.byref(c2), C.byref(k2), C.byref(m2)) == 0
                assert (c2.value, k2.value) == (n_ch.value, n_kept.value) and m2.value > 0
        f0, f1 = np.empty(n, np.int32), np.empty(n, np.int32)
        assert hip.hipMemcpy(f0.ctypes.data, fp[0][0], n * 4, D2H) == 0 and hip.hipMemcpy(f1.ctypes.data, fp[1][0], n * 4, D2H) == 0
        assert np.array_equal(f0, f1)
    finally:
        for p in [d_a, d_off] + [x for pair in fp for x in pair]:
            hip.hipFree(p)


def _chains_under(env, a, off, prm):
    """The device post-pass of a fresh engine made under these environment settings (the form is read when an engine is made)."""
    import os
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with mm.Engine() as e:
            e.set_misc(misc_from(prm))
            res, _ = e.chain_gpu(a, off)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return res


def test_the_forms_of_the_post_pass_agree(engine):
    """Round 6 split the post-pass: the sort level by level over the whole batch (a task per run), the walks per (read, class of trees) on
    waves of their own, the chains put back in the host's order from their ends' sorted positions.  The earlier forms are still in the
    library (MM2GB_POST_SORT=reads: one wave sorts a read from top to bottom; MM2GB_POST_FORM=fused: one wave sorts AND walks a read):
    all three must give the same chains, element for element, on reads of every shape -- equal scores (the sort's order decides), repeats
    (hundreds of trees per read), forests of single anchors, reads with one tree only."""
    prm = orc.default_param()
    rng = np.random.default_rng(2026)
    parts = [sc.read_like(60_000, 3), sc.grid_ties(), sc.sort_by_x(sc.repeat_block(9000, 5, xwin=2500, ywin=3000)), sc.colinear(7000, 21), sc.noise(3000, 4),
             sc.sort_by_x(np.concatenate([sc.repeat_block(3000, 6, xwin=300, ywin=300), sc.colinear(4000, 7)])), sc.colinear(3, 5), sc.colinear(70, 6)]
    for _ in range(12):                                     # the fuzz batches' reads (their own parameters are not used here)
        fa, foff, _ = sc.fuzz_case(rng)
        parts += [fa[foff[r]:foff[r + 1]] for r in range(len(foff) - 1)]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in parts])
    a = np.concatenate(parts)
    for kw in (dict(), dict(min_cnt=1, min_sc=15), dict(bw=100)):
        p2 = orc.default_param(**kw)
        engine.set_misc(misc_from(p2))
        want, _ = engine.chain(a, off, threads=4)          # the host post-pass
        for env in ({}, {"MM2GB_POST_SORT": "reads"}, {"MM2GB_POST_FORM": "fused"}):
            got = _chains_under(env, a, off, p2)
            for r in range(len(parts)):
                assert np.array_equal(got[r][0], want[r][0]) and np.array_equal(got[r][1], want[r][1]), f"{env} {kw}: read {r}"
    for r in (0, 1, 2, 5):
        o = orc.lchain_dp(a[off[r]:off[r + 1]], prm, want_fp=False)
        got = _chains_under({}, a[off[r]:off[r + 1]], np.array([0, off[r + 1] - off[r]], dtype=np.int64), prm)
        assert np.array_equal(got[0][0], o["u"]) and np.array_equal(got[0][1], o["a_out"])


def test_scores_beyond_24_bits_reach_the_fourth_level(engine):
    """A chain of 1.5 M anchors scores past 2^24: the candidates' keys then differ in their top byte, the sort by levels runs all four of its launches
    (key bytes 3 .. 0) and the top pass has few values of its byte (the form that runs where the candidates are collected); next to it reads whose top
    byte takes many values (a first-level task of the ordinary kind) and a short read.  Against the host post-pass, element for element."""
    prm = orc.default_param()
    big = sc.sort_by_x(np.concatenate([sc.colinear(1_500_000, 31, r0=1_000_000, max_gap=33), sc.repeat_block(6000, 32, r0=9_000_000, xwin=3000, ywin=4000)]))
    parts = [big, sc.read_like(40_000, 33), sc.colinear(90, 34)]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in parts])
    a = np.concatenate(parts)
    engine.set_misc(misc_from(prm))
    f, _, _ = engine.score(a, off)
    assert int(f[:len(big)].max()) >= 1 << 24, "the test's chain is too short to score past 2^24"
    res, _ = engine.chain_gpu(a, off)
    host, _ = engine.chain(a, off, threads=8)
    for r in range(len(parts)):
        assert np.array_equal(res[r][0], host[r][0]) and np.array_equal(res[r][1], host[r][1]), f"read {r}"
