"""From sequence to seed matches on the host (csrc/seeding.cpp): minimizer sketch, index look-up, occurrence threshold,
match collection -- against what the reference's mm_collect_matches (seed.c:98) returned for the same sequences, recorded through
oracle/capture_hooks.c into tests/golden/seeds (oracle/gen_golden_seeds.py).  No GPU needed."""
import os
import tempfile

import numpy as np
import pytest

import golden_io
import sim_reads

mm = pytest.importorskip("mm2gb_amd")

DATA = os.path.join(golden_io.GOLD, "data")
SEEDS = os.path.join(golden_io.GOLD, "seeds")


def read_fasta(path):
    recs, name, seq = [], None, []
    with open(path, "rb") as fh:
        for ln in fh:
            ln = ln.strip()
            if ln.startswith(b">"):
                if name is not None:
                    recs.append((name, b"".join(seq)))
                name, seq = ln[1:].split()[0].decode(), []
            elif ln:
                seq.append(ln)
    if name is not None:
        recs.append((name, b"".join(seq)))
    return recs


def check_reads(index, reads, case):
    mid_occ = index.mid_occ()
    n = 0
    for k, (_, seq) in enumerate(reads):
        path = os.path.join(SEEDS, f"{case}_{k}.npz")
        if not os.path.exists(path):
            continue                                   # the reference chained nothing for this read: no record
        g = golden_io.load_seeds(path)
        assert g["qlen"] == len(seq)
        m = index.matches(seq, mid_occ)
        assert np.array_equal(m["seeds"], g["seeds"]), f"{case} read {k}: seeds"
        assert np.array_equal(m["hits"], g["hits"]), f"{case} read {k}: hits"
        assert m["rep_len"] == g["rep_len"] and np.array_equal(m["mini_pos"], g["mini_pos"]), f"{case} read {k}: rep_len / mini_pos"
        n += 1
    return n


@pytest.mark.parametrize("case,tgt,qry", [("mt", "MT-human.fa", "MT-orang.fa"), ("inv", "t-inv.fa", "q-inv.fa"),
                                          ("mt_x_self", "MT-human.fa", "MT-human.fa"), ("mt_x_smaller", "MT-orang.fa", "MT-human.fa")])
def test_reference_test_pairs(case, tgt, qry):
    with mm.SeedIndex([s for _, s in read_fasta(os.path.join(DATA, tgt))]) as ix:
        assert ix.mid_occ() == 10                       # options.c:81-82: never below min_mid_occ
        assert check_reads(ix, read_fasta(os.path.join(DATA, qry)), case) >= 1


def test_simulated_reads_on_a_genome_with_repeats():
    """3 Mbp with interspersed repeat families and tandem arrays: minimizers above mid_occ, streaks of them thinned out (seed.c:58-96),
    repeat length, reads on both strands."""
    with tempfile.TemporaryDirectory() as td:
        ref_fa, reads_fa = os.path.join(td, "ref.fa"), os.path.join(td, "reads.fa")
        sim_reads.simulate(ref_fa, reads_fa, seed=5, n_reads=150, len_lo=3_000, len_hi=20_000)     # as oracle/gen_golden_seeds.py
        with mm.SeedIndex([s for _, s in read_fasta(ref_fa)]) as ix:
            mid_occ = ix.mid_occ()
            assert check_reads(ix, read_fasta(reads_fa), "sim") == 9
            g = [golden_io.load_seeds(p) for p in golden_io.seed_cases() if os.path.basename(p).startswith("sim_") and "for" not in p]
            assert sum(x["rep_len"] > 0 for x in g) >= 5 and max(int(x["seeds"][:, 0].max()) for x in g) > mid_occ   # the filters were exercised


def test_small_genome_against_itself():
    rng = np.random.default_rng(9)                          # as oracle/gen_golden_seeds.py
    chrs = sim_reads.make_genome(rng, n_chr=3, chr_len=20_000, n_rep_families=2, rep_len=300, copies=9, tandem=0)
    seqs = [c.tobytes() for c in chrs]
    with mm.SeedIndex(seqs) as ix:
        assert check_reads(ix, [(None, s) for s in seqs], "self_x") == 3


@pytest.mark.parametrize("k,w", [(4, 3), (7, 5), (11, 10), (15, 10), (21, 11), (28, 19)])
def test_index_lookup_against_a_dictionary_of_the_sketch(k, w):
    """The index answers a look-up through a table of key prefixes (csrc/seeding.cpp, SeedIndex::bucket): for every k-mer size -- few bits per
    key and many keys per prefix, or many bits and empty prefixes -- what a read's minimizers find is what a dictionary built from the
    reference's own sketch holds, occurrence for occurrence and in the index's order (ascending by reference and position)."""
    rng = np.random.default_rng(100 + k)
    refs = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n)) for n in (6000, 2500, 1)]
    table = {}
    for rid, s in enumerate(refs):
        for x, y in mm.sketch(s, w, k, rid=rid):
            table.setdefault(int(x) >> 8, []).append(int(y))
    reads = [refs[0][1000:3000], refs[1][::-1], bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 1500)), refs[0][:k], b""]
    with mm.SeedIndex(refs, k=k, w=w, threads=2) as ix:
        assert ix.size()[0] == len(table) and ix.size()[1] == sum(len(v) for v in table.values())
        for rd in reads:
            m = ix.matches(rd, mid_occ=1 << 30, max_max_occ=1 << 30, occ_dist=0, q_occ_frac=0.0)
            want_seeds, want_hits = [], []
            for x, y in mm.sketch(rd, w, k):
                occ = table.get(int(x) >> 8)
                if occ:
                    want_seeds.append((len(occ), int(y) & 0xffffffff))
                    want_hits += sorted(occ)
            assert [(int(a), int(b)) for a, b in m["seeds"][:, :2]] == want_seeds
            assert m["hits"].tolist() == want_hits


def test_sketch_edge_cases():
    assert mm.sketch(b"", 10, 15).shape == (0, 2)
    assert mm.sketch(b"ACGTACGTAC", 10, 15).shape == (0, 2)                      # shorter than k
    a = mm.sketch(b"ACGTTGCATGCCATGA" * 20 + b"NNNN" + b"GATTACAGATTACAGGATC" * 10, 10, 15, rid=3)
    assert len(a) > 0 and np.all(a[:, 1] >> 32 == 3) and np.all((a[:, 0] & 0xff) == 15)
    pos = (a[:, 1] & 0xffffffff) >> 1
    assert np.all(np.diff(pos.astype(np.int64)) >= 0)                               # minimizers come out in order of position
    lower = mm.sketch((b"ACGTTGCATGCCATGA" * 20).lower(), 10, 15)
    upper = mm.sketch(b"ACGTTGCATGCCATGA" * 20, 10, 15)
    assert np.array_equal(lower, upper)


def test_matches_to_anchors_on_host_threads_reference_vectors():
    """mm2gb_collect_seeds_host = collect_seed_hits + skip_seed (map.c:205-227,295-331) on host threads: every recorded call of the
    reference (both strands, --for-only / --rev-only, the name tests of -X), batched by option set, anchors bit-identical."""
    cases = [golden_io.load_seeds(p) for p in golden_io.seed_cases()]
    groups = {}
    for g in cases:
        groups.setdefault((g["flag"], tuple(g["ref_rank"] or ()), tuple(g["ref_len"] or ())), []).append(g)
    assert len(groups) >= 4
    for (flag, ref_rank, ref_len), gs in groups.items():
        reads = [dict(seeds=g["seeds"], hits=g["hits"], qlen=g["qlen"], **({"q_rank": g["q_rank"]} if ref_rank else {})) for g in gs]
        got = mm.collect_seeds_host(flag, reads, ref_len=list(ref_len) or None, ref_rank=list(ref_rank) or None, threads=3)
        for g, a in zip(gs, got):
            assert a.shape == g["a"].shape and np.array_equal(a, g["a"]), g["name"]
