"""SURVEY 8(f) N4, the two formats either side of the path that do not need an index: the seed sort upstream
(radix_sort_128x of collect_seed_hits, map.c:329) and chains -> hit records downstream (mm_gen_regs, hit.c:52-88), on the
device, against the oracle (pinned to the compiled reference in tests/test_oracle_vs_ref.py)."""
import numpy as np
import pytest

import orc
import synth_cases as sc

pytestmark = pytest.mark.gpu

mm = pytest.importorskip("mm2gb_amd")


@pytest.fixture(scope="module")
def engine():
    with mm.Engine() as e:
        yield e


def test_seed_sort_equals_radix_sort_128x(engine):
    """Unsorted seeds of many reads: equal x are frequent (repeats), runs of every size class, empty and tiny reads."""
    rng = np.random.default_rng(4)
    reads = [np.zeros((0, 2), np.uint64), sc.noise(1, 1), sc.noise(64, 2), sc.noise(65, 3)]
    a, off = mm.synth_reads(5, 0, 10, 10_000, 150_000)
    reads += [a[off[r]:off[r + 1]] for r in range(10)]
    dup = sc.pack(np.full(5000, 2), rng.integers(0, 2, 5000), 7000 + rng.integers(0, 300, 5000), rng.integers(0, 50_000, 5000))     # ~16 anchors per x
    reads.append(dup)
    shuffled = []
    for x in reads:
        x = x.copy(); rng.shuffle(x); shuffled.append(x)
    o2 = np.zeros(len(shuffled) + 1, np.int64); o2[1:] = np.cumsum([len(x) for x in shuffled])
    got = engine.sort_seeds(np.concatenate(shuffled), o2)
    for r, x in enumerate(shuffled):
        assert np.array_equal(got[o2[r]:o2[r + 1]], orc.radix_sort_x(x)), f"read {r} ({len(x)} anchors)"


def test_unsorted_seeds_to_hit_records(engine):
    """The widened path end to end on the device: unsorted seeds -> sort -> chaining -> backtrack + compaction -> hit records,
    against the oracle's radix_sort_128x + mg_lchain_dp + mm_gen_regs."""
    rng = np.random.default_rng(8)
    a, off = mm.synth_reads(12, 0, 16, 10_000, 120_000)
    reads = [a[off[r]:off[r + 1]].copy() for r in range(16)]
    reads.append(sc.sort_by_x(np.concatenate([sc.repeat_block(3000, 5, xwin=900, ywin=900), sc.colinear(300, 6)])))     # many chains, equal scores
    for x in reads:
        rng.shuffle(x)
    o2 = np.zeros(len(reads) + 1, np.int64); o2[1:] = np.cumsum([len(x) for x in reads])
    prm = orc.default_param(min_cnt=2, min_sc=20)
    engine.set_misc(mm.default_misc(min_cnt=2, min_score=20))
    srt = engine.sort_seeds(np.concatenate(reads), o2)
    chains, _ = engine.chain_gpu(srt, o2)
    qlen = rng.integers(100_000, 200_000, len(reads)).astype(np.int32)
    hashes = rng.integers(0, 2**32, len(reads), dtype=np.uint64).astype(np.uint32)
    for is_q in (0, 1):
        regs = engine.gen_regs(chains, qlen, hashes, is_q)
        n_many = 0
        for r, x in enumerate(reads):
            o = orc.lchain_dp(orc.radix_sort_x(x), prm, want_fp=False)
            assert np.array_equal(chains[r][0], o["u"]) and np.array_equal(chains[r][1], o["a_out"]), f"read {r}"
            want = orc.gen_regs(o["u"], o["a_out"], int(qlen[r]), int(hashes[r]), is_q)
            assert np.array_equal(regs[r], want), f"read {r}: hit records differ"
            n_many += len(want) > 64
        assert n_many >= 1
