"""Whole-batch calls above the engine (pool.cpp): several engines in one process (no exchange between devices) and
the host post-pass overlapped with scoring.  The GPU box has one GPU, so the multi-engine cases put two or three engines
on device 0 -- same code path (one host thread, arenas and streams per engine), same dealing of reads."""
import numpy as np
import pytest

import mm2gb_amd as mm
import orc
import synth_cases as sc
from test_gpu_parity import misc_from, rel

pytestmark = pytest.mark.gpu


def oracle_chains(a, off, prm):
    return [orc.lchain_dp(a[off[r]:off[r + 1]], prm, want_fp=False) for r in range(len(off) - 1)]


def same_chains(res, want):
    assert len(res) == len(want)
    for r, (got, o) in enumerate(zip(res, want)):
        assert np.array_equal(got[0], o["u"]) and np.array_equal(got[1], o["a_out"]), f"read {r}"


def test_pool_deals_reads_evenly_and_matches_the_oracle():
    a, off = mm.synth_reads(31, 0, 30, 10_000, 80_000)
    prm = orc.default_param()
    fo, po, pairs = orc.chain_fill_many(a, off, prm, threads=4)
    po_rel = np.concatenate([rel(po[off[r]:off[r + 1]]) for r in range(len(off) - 1)])
    with mm.Pool(devices=[0, 0, 0], misc=misc_from(prm)) as pool:
        assert len(pool) == 3 and pool.devices() == [0, 0, 0]
        f, p, st, first = pool.score(a, off)
        assert np.array_equal(f, fo) and np.array_equal(p, po_rel)
        assert st["n_pairs"] == pairs and st["n_anchors"] == len(a) and st["n_reads"] == 30
        # contiguous, ordered shares of about a third of the anchors each (within one read of even)
        assert first[0] == 0 and first[-1] == 30 and np.all(np.diff(first) >= 0)
        share = np.diff(off[first])
        longest = int(np.diff(off).max())
        assert np.all(np.abs(share - len(a) / 3) <= longest)
        same_chains(pool.chain(a, off, threads=3)[0], oracle_chains(a, off, prm))


def test_pool_with_more_engines_than_reads_and_empty_batches():
    prm = orc.default_param()
    with mm.Pool(devices=[0, 0, 0], misc=misc_from(prm)) as pool:
        a = sc.read_like(3000, 5)
        f, p, st, first = pool.score(a, np.array([0, len(a)], np.int64))
        fo, po, _ = orc.chain_fill(a, prm)
        assert np.array_equal(f, fo) and np.array_equal(p, rel(po)) and st["n_reads"] == 1
        assert list(np.diff(first)).count(1) == 1 and first[-1] == 1
        f, p, st, first = pool.score(np.zeros((0, 2), np.uint64), np.zeros(1, np.int64))
        assert len(f) == 0 and st["n_anchors"] == 0 and list(first) == [0, 0, 0, 0]
        res, _ = pool.chain(np.zeros((0, 2), np.uint64), np.zeros(4, np.int64), threads=2)      # three empty reads
        assert len(res) == 3 and all(len(u) == 0 and len(ao) == 0 for u, ao in res)


def test_pool_parameters_follow_set_misc():
    a = sc.sort_by_x(np.concatenate([sc.repeat_block(900, 31), sc.colinear(300, 32)]))
    parts = [a, sc.read_like(5000, 33), a[:400]]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in parts])
    allv = np.concatenate(parts)
    with mm.Pool(devices=[0, 0]) as pool:
        for kw in (dict(max_iter=100), dict(bw=100, min_cnt=2), dict(is_cdna=1)):
            prm = orc.default_param(**kw)
            pool.set_misc(misc_from(prm))
            same_chains(pool.chain(allv, off, threads=2)[0], oracle_chains(allv, off, prm))


def test_post_pass_overlaps_slices(monkeypatch):
    """Chains of a batch cut into many slices (the post-pass threads start on slice k while slice k+1 is on the device)
    equal those of the unsliced call and the oracle's, for one engine and for a pool."""
    a, off = mm.synth_reads(41, 0, 36, 10_000, 50_000)
    prm = orc.default_param()
    want = oracle_chains(a, off, prm)
    with mm.Engine(misc=misc_from(prm)) as e:
        whole, st0 = e.chain(a, off, threads=4)
    same_chains(whole, want)
    monkeypatch.setenv("MM2GB_SLICE_ANCHORS", "40000")
    with mm.Engine(misc=misc_from(prm)) as e:
        for threads in (1, 5):
            res, st = e.chain(a, off, threads=threads)
            same_chains(res, want)
            assert st["n_pairs"] == st0["n_pairs"] and st["n_reads"] == 36
    with mm.Pool(devices=[0, 0], misc=misc_from(prm)) as pool:
        res, st = pool.chain(a, off, threads=4)
        same_chains(res, want)
        assert st["n_pairs"] == st0["n_pairs"] and st["n_anchors"] == len(a)


def test_pool_rejects_bad_arguments():
    with pytest.raises(mm.Mm2gbError):
        mm.Pool(devices=[0, 99])
    with mm.Pool(devices=[0]) as pool:
        a = sc.read_like(500, 1)
        with pytest.raises(mm.Mm2gbError):
            pool.score(a, np.array([1, len(a)], np.int64))          # offsets[0] != 0
