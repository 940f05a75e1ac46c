"""The host form of RMQ re-chaining (csrc/rmq_host.cpp: mg_lchain_rmq's fill with a segment tree over (y, index) ranks, lchain.c:250-369)
against the reference's own vectors and the CPU oracle.  No GPU needed."""
import numpy as np
import pytest

import golden_io
import orc
import synth_cases as sc

mm = pytest.importorskip("mm2gb_amd")

CASES = golden_io.rmq_cases()


def to_lib(prm):
    return mm.RmqParam(max_dist=prm.max_dist, max_dist_inner=prm.max_dist_inner, bw=prm.bw, max_chn_skip=prm.max_chn_skip, cap_rmq_size=prm.cap_rmq_size,
                       min_cnt=prm.min_cnt, min_sc=prm.min_sc, chn_pen_gap=np.float32(prm.pen_gap), chn_pen_skip=np.float32(prm.pen_skip))


def first_pass(a):
    o = orc.lchain_dp(a, orc.default_param(), want_fp=False)
    return orc.radix_sort_x(o["a_out"]) if len(o["a_out"]) else o["a_out"]


@pytest.mark.parametrize("path", CASES, ids=golden_io.case_ids(CASES))
def test_reference_vectors(path):
    """Every recorded call of the reference's mg_lchain_rmq, the ones whose range-minimum met ties included: this form keeps the
    reference's tree, so its chains are the reference's."""
    g = golden_io.load_rmq(path)                        # some were recorded with a finite max_chn_skip: this form honours it
    res, _ = mm.rmq_chain_host(g["a"], np.array([0, len(g["a"])], np.int64), to_lib(g["prm"]), threads=1)
    assert np.array_equal(res[0][0], g["u"]) and np.array_equal(res[0][1], g["a_out"])


def batch_of_reads():
    a, off = mm.synth_reads(41, 0, 24, 10_000, 90_000)
    reads = [first_pass(a[off[r]:off[r + 1]]) for r in range(24)]
    reads.insert(7, np.zeros((0, 2), np.uint64))
    rng = np.random.default_rng(3)
    for k in range(4):   # dense clouds: many equal priorities, i.e. ties
        reads.append(orc.radix_sort_x(sc.sort_by_x(sc.pack(np.full(900, 1), np.zeros(900, np.int64), 1000 + rng.integers(0, 150 + 200 * k, 900), 100 + rng.integers(0, 150 + 100 * k, 900)))))
    return reads


PARAMS = (dict(), dict(max_chn_skip=25), dict(max_chn_skip=3, bw=20000, max_dist=5000, max_dist_inner=1000), dict(cap_rmq_size=64), dict(max_dist_inner=0), dict(bw=300, max_dist=1500, max_dist_inner=200), dict(bw=20000, max_dist=5000, max_dist_inner=1000),
          dict(pen_gap=np.float32(0.3), pen_skip=np.float32(0.05)))


def test_batch_against_the_oracle_where_there_is_no_tie():
    """Reads re-chained in one call on several threads, under the parameter sets the device test uses plus the real call's (bw = bw_long
    = 20 000, map.c:704).  The oracle scans ranges and only KNOWS the answer where the range-minimum is unique: those reads must agree."""
    reads = batch_of_reads()
    o2 = np.zeros(len(reads) + 1, dtype=np.int64)
    o2[1:] = np.cumsum([len(x) for x in reads])
    allr = np.concatenate(reads)
    for kw in PARAMS:
        prm = orc.default_rmq_param(**kw)
        res, _ = mm.rmq_chain_host(allr, o2, to_lib(prm), threads=4)
        untied = 0
        for r, x in enumerate(reads):
            o = orc.lchain_rmq(x, prm)
            if o["n_tied"] == 0:
                untied += 1
                assert np.array_equal(res[r][0], o["u"]) and np.array_equal(res[r][1], o["a_out"]), (kw, r)
        assert untied >= 10


@pytest.mark.skipif(not orc.ref_available(), reason="needs the reference build (oracle/_ref, made where /root/reference exists)")
def test_ties_are_broken_like_the_reference():
    """Against the compiled reference's mg_lchain_rmq itself, on reads where many elements share the smallest priority: which one the
    reference returns follows from the shape of its AVL tree (krmq.h); ShapeTree must be that tree."""
    reads = batch_of_reads()
    n_tied_reads = 0
    for kw in PARAMS:
        prm = orc.default_rmq_param(**kw)
        o2 = np.zeros(len(reads) + 1, dtype=np.int64)
        o2[1:] = np.cumsum([len(x) for x in reads])
        res, _ = mm.rmq_chain_host(np.concatenate(reads), o2, to_lib(prm), threads=4)
        for r, x in enumerate(reads):
            if len(x) == 0:
                continue
            ref = orc.ref_lchain_rmq(x, prm)
            n_tied_reads += orc.lchain_rmq(x, prm)["n_tied"] > 0
            assert np.array_equal(res[r][0], ref["u"]) and np.array_equal(res[r][1], ref["a_out"]), (kw, r)
    assert n_tied_reads >= 8


def test_vector_and_scalar_inner_scans_agree(monkeypatch):
    """The exhaustive inner scan reads eight candidates per instruction where AVX2 is there (scan_bucket_avx2); MM2GB_RMQ_NO_SIMD=1 keeps the
    scalar loop.  Same chains from both, on reads with full inner windows (dense clouds, many interleaved chains) under the parameter sets
    that take the vector path (no skip limit, chn_pen_skip == 0)."""
    rng = np.random.default_rng(17)
    reads = batch_of_reads()
    for n, w in ((5000, 900), (12000, 2500)):
        reads.append(orc.radix_sort_x(sc.sort_by_x(sc.pack(np.full(n, 1), np.zeros(n, np.int64), 1000 + rng.integers(0, w, n), 100 + rng.integers(0, w, n)))))
    o2 = np.zeros(len(reads) + 1, dtype=np.int64)
    o2[1:] = np.cumsum([len(x) for x in reads])
    allr = np.concatenate(reads)
    for kw in (dict(), dict(bw=20000, max_dist=5000, max_dist_inner=1000), dict(bw=300, max_dist=1500, max_dist_inner=200), dict(bw=7, max_dist=900, max_dist_inner=300)):
        prm = to_lib(orc.default_rmq_param(**kw))
        monkeypatch.delenv("MM2GB_RMQ_NO_SIMD", raising=False)
        fast, _ = mm.rmq_chain_host(allr, o2, prm, threads=4)
        monkeypatch.setenv("MM2GB_RMQ_NO_SIMD", "1")
        slow, _ = mm.rmq_chain_host(allr, o2, prm, threads=4)
        for r in range(len(reads)):
            assert np.array_equal(fast[r][0], slow[r][0]) and np.array_equal(fast[r][1], slow[r][1]), (kw, r)


def test_ties_are_weighed_before_a_read_is_done_again(monkeypatch):
    """The tournament tree cannot say which holder of a shared smallest priority the reference's tree returns -- but where every holder leaves the
    anchor with the same score and predecessor it need not (rmq_fill_one): the read is done again with the reference's tree exactly when the
    oracle finds a tie that decides (orc_rmq_last_ties_that_decide), and with MM2GB_RMQ_TIES=strict whenever it finds a tie at all.  The chains
    are the same either way."""
    reads = batch_of_reads()
    o2 = np.zeros(len(reads) + 1, dtype=np.int64)
    o2[1:] = np.cumsum([len(x) for x in reads])
    allr = np.concatenate(reads)
    fewer = 0
    for kw in PARAMS:
        prm = orc.default_rmq_param(**kw)
        monkeypatch.delenv("MM2GB_RMQ_TIES", raising=False)
        res, again = mm.rmq_chain_host(allr, o2, to_lib(prm), threads=4)
        monkeypatch.setenv("MM2GB_RMQ_TIES", "strict")
        res_s, again_s = mm.rmq_chain_host(allr, o2, to_lib(prm), threads=4)
        for r, x in enumerate(reads):
            o = orc.lchain_rmq(x, prm)
            assert (again[r] != 0) == (o["n_decide"] > 0) and (again_s[r] != 0) == (o["n_tied"] > 0), (kw, r)
            assert o["n_decide"] <= o["n_tied"] and (prm.max_chn_skip == orc.INT32_MAX or o["n_decide"] == o["n_tied"])
            assert np.array_equal(res[r][0], res_s[r][0]) and np.array_equal(res[r][1], res_s[r][1]), (kw, r)
        fewer += int(again_s.sum() - again.sum())
    assert fewer >= 3
