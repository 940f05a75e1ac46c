"""SURVEY 8(f) N4, the formats either side of the path that do not need an index: seed matches -> anchors and the seed sort
upstream (collect_seed_hits, map.c:295-331) and chains -> hit records downstream (mm_gen_regs, hit.c:52-88), on the device,
against the reference's vectors and the oracle (pinned to the compiled reference in tests/test_oracle_golden.py and
tests/test_oracle_vs_ref.py)."""
import numpy as np
import pytest

import golden_io
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


def test_seed_matches_to_anchors_reference_vectors(engine):
    """Every recorded call of the reference's collect_seed_hits, all in ONE batch (reads with different option bits go in separate
    calls: the bits are per call): anchors bit-identical, order of equal x included."""
    cases = [golden_io.load_seeds(p) for p in golden_io.seed_cases()]
    assert cases
    groups = {}
    for g in cases:
        key = (g["flag"], tuple(g["ref_rank"] or ()), tuple(g["ref_len"] or ()))
        groups.setdefault(key, []).append(g)
    for (flag, ref_rank, ref_len), gs in groups.items():
        reads = [dict(seeds=g["seeds"], hits=g["hits"], qlen=g["qlen"], **({"q_rank": g["q_rank"]} if ref_rank else {})) for g in gs]
        got = engine.collect_seeds(flag, reads, ref_len=list(ref_len) or None, ref_rank=list(ref_rank) or None)
        for g, a in zip(gs, got):
            assert a.shape == g["a"].shape and np.array_equal(a, g["a"]), g["name"]


def random_matches(rng, n_seeds, qlen, n_ref, max_hits, ref_len):
    """Seed matches as mm_collect_matches could return them: any q_pos / strand, spans, segment ids, tandem bits, hit lists of any
    length incl. empty, hits on the diagonal of a same-length reference sequence."""
    seeds = np.zeros((n_seeds, 4), np.uint32)
    n = rng.integers(0, max_hits + 1, n_seeds)
    n[rng.random(n_seeds) < 0.2] = 0
    span = rng.integers(5, 29, n_seeds)
    qp = rng.integers(span, qlen, n_seeds)                    # last base of the k-mer
    seeds[:, 0] = n
    seeds[:, 1] = qp << 1 | rng.integers(0, 2, n_seeds)
    seeds[:, 2] = span | (rng.integers(0, 2, n_seeds) << 31)
    seeds[:, 3] = rng.integers(0, 3, n_seeds) | (rng.integers(0, 2, n_seeds) << 31)
    hits = []
    for k in range(n_seeds):
        rid = rng.integers(0, n_ref, n[k])
        pos = np.array([rng.integers(int(span[k]), ref_len[r]) for r in rid], dtype=np.int64)
        diag = rng.random(n[k]) < 0.15
        pos[diag] = qp[k]                                       # same position as in the query: what NO_DIAG looks for
        hits.append((rid.astype(np.uint64) << np.uint64(32)) | (pos.astype(np.uint64) << np.uint64(1)) | rng.integers(0, 2, n[k]).astype(np.uint64))
    return seeds, (np.concatenate(hits) if hits else np.zeros(0, np.uint64)).astype(np.uint64)


@pytest.mark.parametrize("flag", [0, orc.MM_F_FOR_ONLY, orc.MM_F_REV_ONLY, orc.MM_F_QSTRAND, orc.MM_F_NO_DIAG, orc.MM_F_NO_DUAL,
                                  orc.MM_F_NO_DIAG | orc.MM_F_NO_DUAL | orc.MM_F_QSTRAND, orc.MM_F_NO_DIAG | orc.MM_F_FOR_ONLY])
def test_seed_matches_to_anchors_fuzz(engine, flag):
    """Random matches under every combination of option bits the function looks at, reads of 0 .. 40 k hits, against the oracle."""
    rng = np.random.default_rng(100 + (flag & 0xffff) + (flag >> 20))
    n_ref = 7
    ref_len = [int(v) for v in rng.integers(4000, 60000, n_ref)]
    ref_rank = [int(v) for v in rng.permutation(n_ref)]
    ref_rank[3] = ref_rank[5]                                   # two sequences with the same name
    reads = []
    for r in range(40):
        qlen = ref_len[r % n_ref] if r % 3 == 0 else int(rng.integers(500, 70000))
        n_seeds = 0 if r == 7 else int(rng.integers(1, 1500))
        seeds, hits = random_matches(rng, n_seeds, qlen, n_ref, 1 if r == 11 else 40, ref_len)
        reads.append(dict(seeds=seeds, hits=hits, qlen=qlen, q_rank=ref_rank[r % n_ref] if r % 2 == 0 else int(rng.integers(0, n_ref))))
    got = engine.collect_seeds(flag, reads, ref_len=ref_len, ref_rank=ref_rank)
    kept = 0
    for r, rd in enumerate(reads):
        hit_off = np.zeros(len(rd["seeds"]) + 1, np.int64); np.cumsum(rd["seeds"][:, 0], out=hit_off[1:])
        want = orc.collect_seeds(flag, rd["qlen"], rd["seeds"], hit_off, rd["hits"], q_rank=rd["q_rank"], ref_len=ref_len, ref_rank=ref_rank)
        assert got[r].shape == want.shape and np.array_equal(got[r], want), f"read {r}"
        kept += len(want)
    assert kept > 0


def test_sequence_to_sorted_anchors(engine):
    """The whole producer side without the reference: FASTA -> minimizers -> index look-up -> matches (host, csrc/seeding.cpp) -> anchors,
    sorted (device) == the anchors the reference handed to its chaining for the same sequences."""
    import os
    from test_seeding_cpu import read_fasta, DATA
    for case, tgt, qry in (("mt", "MT-human.fa", "MT-orang.fa"), ("inv", "t-inv.fa", "q-inv.fa")):
        with mm.SeedIndex([s for _, s in read_fasta(os.path.join(DATA, tgt))]) as ix:
            reads = [ix.matches(s, ix.mid_occ()) for _, s in read_fasta(os.path.join(DATA, qry))]
            got = engine.collect_seeds(0, reads)
            for k, a in enumerate(got):
                g = golden_io.load_seeds(os.path.join(golden_io.GOLD, "seeds", f"{case}_{k}.npz"))
                assert np.array_equal(a, g["a"]), f"{case} read {k}"
