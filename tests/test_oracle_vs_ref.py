"""Oracle vs the reference compiled in this container (oracle/_ref, built from /root/reference by oracle/Makefile).
Skipped where the reference build is absent (e.g. the GPU box) -- there test_oracle_golden.py carries the pin."""
import numpy as np
import pytest

import orc
import synth_cases as sc

pytestmark = pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


def _same(a, prm):
    r = orc.ref_lchain_dp(a, prm)
    o = orc.lchain_dp(a, prm)
    assert np.array_equal(r["f"], o["f"]) and np.array_equal(r["p"], o["p"])
    assert np.array_equal(r["u"], o["u"]) and np.array_equal(r["a_out"], o["a_out"])


@pytest.mark.parametrize("seed", range(6))
def test_random_reads(seed):
    rng = np.random.default_rng(100 + seed)
    _same(sc.read_like(int(rng.integers(3000, 40000)), 200 + seed), orc.default_param())


@pytest.mark.parametrize("kw", [dict(max_skip=25), dict(max_skip=0), dict(max_iter=50), dict(bw=50), dict(is_cdna=1),
                                dict(pen_skip=np.float32(0.03)), dict(min_cnt=1, min_sc=10), dict(max_dist_x=300, max_dist_y=200)])
def test_parameter_variants(kw):
    _same(sc.read_like(12000, 77), orc.default_param(**kw))


def test_saturated_repeat_and_rescue():
    _same(sc.sort_by_x(np.concatenate([sc.repeat_block(8000, 31), sc.colinear(600, 32)])), orc.default_param())
    _same(sc.rescue_case(n_noise=700, n_chain=40, seed=3), orc.default_param(max_iter=200))


def test_multi_segment():
    _same(sc.two_segments(600, 41), orc.default_param(n_seg=2))
    _same(sc.two_segments(600, 42), orc.default_param(n_seg=2, is_cdna=1))


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 66, 300, 5000, 70000])
def test_radix_sort_order_matches(n):
    """ksort.h:98-151 is unstable; equal keys must come out in the reference's order."""
    rng = np.random.default_rng(n)
    keys = rng.integers(0, max(2, n // 8 + 2), max(n, 0)).astype(np.uint64) << np.uint64(8 * int(rng.integers(0, 7)))
    arr = np.stack([keys, np.arange(n, dtype=np.uint64)], axis=1) if n else np.zeros((0, 2), np.uint64)
    assert np.array_equal(orc.ref_radix_sort(arr), orc.radix_sort_x(arr))


def test_pair_score_exhaustive_small_grid():
    """Every (dr, dq) on a grid through both implementations' DP: covered indirectly; here check the
    float path on the dd range the penalty can see (0..bw) against a direct numpy float32 evaluation."""
    prm = orc.default_param()
    gap = np.float32(prm.pen_gap)
    for dd in list(range(0, 501, 7)) + [1, 2, 3, 499, 500]:
        cur = np.array([5000 + 100 + dd, (15 << 32) | (2000 + 100)], dtype=np.uint64)
        prev = np.array([5000, (15 << 32) | 2000], dtype=np.uint64)
        sc_ = orc.pair_score(cur, prev, prm)
        lin = np.float32(gap * np.float32(dd))
        lg = np.float32(orc.lib().orc_log2_approx(float(dd + 1))) if dd >= 1 else np.float32(0)
        exp = 15 - int(np.float32(lin + np.float32(0.5) * lg)) if dd else 15
        assert sc_ == exp


def test_rmq_oracle_vs_reference_on_rechaining_inputs():
    """The RMQ restatement against the reference's mg_lchain_rmq, in-process, on what post_chaining_helper hands it: anchors kept
    by the first chaining, re-sorted by x (map.c:444-451).  Every case without a tie must agree in f, p, chains and anchors; with
    finite max_chn_skip, a small tree cap, no inner tree and other distances too."""
    import mm2gb_amd as mm
    n_cases = n_tied = 0
    for seed in range(4):
        a, off = mm.synth_reads(100 + seed, 0, 3, 10_000, 60_000)
        for r in range(3):
            o1 = orc.lchain_dp(a[off[r]:off[r + 1]], orc.default_param(), want_fp=False)
            if not len(o1["a_out"]):
                continue
            x = orc.ref_radix_sort(o1["a_out"])
            for kw in (dict(), dict(max_chn_skip=25), dict(cap_rmq_size=40), dict(max_dist_inner=0), dict(bw=500, max_dist=2000, max_dist_inner=300)):
                prm = orc.default_rmq_param(**kw)
                o, rf = orc.lchain_rmq(x, prm), orc.ref_lchain_rmq(x, prm)
                n_cases += 1
                if o["n_tied"]:
                    n_tied += 1
                    continue
                assert np.array_equal(o["f"], rf["f"]) and np.array_equal(o["p"], rf["p"]), (seed, r, kw)
                assert np.array_equal(o["u"], rf["u"]) and np.array_equal(o["a_out"], rf["a_out"]), (seed, r, kw)
    assert n_cases >= 40 and n_tied < n_cases // 2


def test_gen_regs_oracle_vs_reference():
    """orc_gen_regs against the reference's mm_gen_regs (hit.c:52-88, libminimap2ref.so): the leading 72 bytes of every mm_reg1_t,
    for few and for many chains (> 64: the radix passes of the score sort), both strand conventions."""
    import mm2gb_amd as mm
    import synth_cases as sc
    rng = np.random.default_rng(2)
    a, off = mm.synth_reads(9, 0, 6, 10_000, 120_000)
    cases = [(a[off[r]:off[r + 1]], orc.default_param()) for r in range(6)]
    cases.append((sc.sort_by_x(np.concatenate([sc.repeat_block(3000, 5, xwin=900, ywin=900), sc.colinear(300, 6)])), orc.default_param(min_cnt=2, min_sc=20)))
    n_many = 0
    for x, prm in cases:
        o = orc.lchain_dp(x, prm, want_fp=False)
        for is_q in (0, 1):
            h, qlen = int(rng.integers(0, 2**32)), int(rng.integers(100_000, 200_000))
            got, want = orc.gen_regs(o["u"], o["a_out"], qlen, h, is_q), orc.ref_gen_regs(o["u"], o["a_out"], qlen, h, is_q)
            assert np.array_equal(got, want)
            n_many += len(want) > 64
    assert n_many >= 1
