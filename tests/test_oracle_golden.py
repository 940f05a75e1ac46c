"""The CPU oracle (oracle/chain_oracle.c) against the committed reference vectors in tests/golden/.

Every fixture was produced by the reference's own mg_lchain_dp (lchain.c:148-217); this test runs wherever the
repo is checked out (no /root/reference needed) and is what pins the oracle on the GPU box."""
import numpy as np
import pytest

import golden_io
import orc

CASES = golden_io.all_cases()


def test_fixture_inventory():
    names = golden_io.case_ids(CASES)
    assert "real_mt_inf_0" in names and "real_inv_inf_1" in names and "synth_rescue9000" in names
    assert len(names) >= 20


@pytest.mark.parametrize("path", CASES, ids=golden_io.case_ids(CASES))
def test_oracle_matches_reference_vectors(path):
    g = golden_io.load(path)
    o = orc.lchain_dp(g["a"], g["prm"])
    assert np.array_equal(o["f"], g["f"]), "f[] differs from the reference"
    assert np.array_equal(o["p"], g["p"]), "p[] differs from the reference"
    assert np.array_equal(o["u"], g["u"]), "chains u[] differ from the reference"
    assert np.array_equal(o["a_out"], g["a_out"]), "compacted anchors differ from the reference"


@pytest.mark.parametrize("path", CASES, ids=golden_io.case_ids(CASES))
def test_backtrack_compact_from_reference_fp(path):
    """backtrack + compaction alone, fed the reference's own f/p."""
    g = golden_io.load(path)
    u, a_out = orc.backtrack_compact(g["a"], g["f"], g["p"], g["prm"])
    assert np.array_equal(u, g["u"])
    assert np.array_equal(a_out, g["a_out"])


def test_mt_golden_is_the_survey_chain():
    """configs[0]: MT-human x MT-orang -> 346 anchors, one chain of 342."""
    g = golden_io.load([p for p in CASES if p.endswith("real_mt_inf_0.npz")][0])
    assert len(g["a"]) == 346
    assert len(g["u"]) == 1 and int(g["u"][0] & 0xffffffff) == 342
    paf = open(golden_io.GOLD + "/real_mt_inf.paf").read().split("\t")
    assert paf[0] == "MT_orang" and paf[5] == "MT_human" and "cm:i:342" in paf and "s1:i:3187" in paf


def test_rescue_fixture_exercises_max_ii():
    """the crafted case must actually take the lchain.c:196-201 branch."""
    g = golden_io.load([p for p in CASES if p.endswith("synth_rescue9000.npz")][0])
    _, _, st = orc.chain_fill(g["a"], g["prm"])
    assert st["n_clamped"] > 0 and st["n_rescue_eval"] > 0 and st["n_rescue_taken"] >= 1
    far = (np.arange(len(g["p"])) - g["p"])[g["p"] >= 0].max()
    assert far > g["prm"].max_iter, "some predecessor must lie further back than max_iter"


def test_log2_known_values():
    # mmpriv.h:118-126 is a quadratic fit: exact at powers of two up to the fit's constant offset
    L = orc.lib()
    for k in range(1, 20):
        assert abs(L.orc_log2_approx(float(2 ** k)) - k) < 0.01
    assert abs(L.orc_log2_approx(3.0) - np.log2(3.0)) < 0.01


def test_empty_and_single():
    prm = orc.default_param()
    o = orc.lchain_dp(np.zeros((0, 2), np.uint64), prm)
    assert len(o["u"]) == 0 and len(o["a_out"]) == 0
    one = np.array([[5 << 32 | 100, 15 << 32 | 50]], dtype=np.uint64)
    f, p, st = orc.chain_fill(one, prm)
    assert f.tolist() == [15] and p.tolist() == [-1] and st["n_pairs"] == 0


@pytest.mark.parametrize("path", golden_io.rmq_cases(), ids=golden_io.case_ids(golden_io.rmq_cases()))
def test_rmq_oracle_matches_reference_vectors(path):
    """orc_lchain_rmq (brute-force restatement of mg_lchain_rmq, lchain.c:250-369) against what the reference computed.  Where the
    reference had to break a tie by the shape of its tree the oracle must SAY so (n_tied > 0) and is not compared."""
    g = golden_io.load_rmq(path)
    o = orc.lchain_rmq(g["a"], g["prm"])
    assert o["n_tied"] == g["tied"]
    if g["tied"] == 0:
        assert np.array_equal(o["f"], g["f"]) and np.array_equal(o["p"], g["p"])
        assert np.array_equal(o["u"], g["u"]) and np.array_equal(o["a_out"], g["a_out"])


@pytest.mark.parametrize("path", golden_io.seed_cases(), ids=golden_io.case_ids(golden_io.seed_cases()))
def test_seed_collection_oracle_matches_reference_vectors(path):
    """orc_collect_seeds (collect_seed_hits + skip_seed, map.c:205-227,295-331) against the anchors the reference made of the same
    matches: both strands, --for-only / --rev-only, the name tests of -X (diagonal dropped, self flag, dual pairs)."""
    g = golden_io.load_seeds(path)
    a = orc.collect_seeds(g["flag"], g["qlen"], g["seeds"], g["hit_off"], g["hits"], q_rank=g["q_rank"], ref_len=g["ref_len"], ref_rank=g["ref_rank"])
    assert a.shape == g["a"].shape and np.array_equal(a, g["a"])
