"""Reads in, PAF out without the reference's sources (csrc/seeding.cpp + csrc/mapper.cpp + the device path), against the
PAF the reference's CPU path printed for the same sequences at max-chain-skip = infinity (tests/golden/*.paf): BASELINE configs[0]/[1]
(MT-human x MT-orang), the reference's other test pairs, and 160 simulated long reads on a 3 Mbp genome with repeats."""
import hashlib
import json
import os

import pytest

import golden_io
import sim_reads
from test_seeding_cpu import DATA, read_fasta

pytestmark = pytest.mark.gpu

mm = pytest.importorskip("mm2gb_amd")
GOLD = golden_io.GOLD


@pytest.fixture(scope="module")
def engine():
    with mm.Engine() as e:
        yield e


def map_files(engine, ref_fa, reads_fa, **opt):
    refs, reads = read_fasta(ref_fa), read_fasta(reads_fa)
    with mm.SeedIndex([s for _, s in refs]) as ix:
        return mm.map_reads(engine, ix, [n for n, _ in refs], reads, opt=mm.map_opt(**opt))


@pytest.mark.parametrize("seeds_on_device", [1, -1], ids=["anchors_on_device", "anchors_on_host"])
@pytest.mark.parametrize("case,tgt,qry", [("mt", "MT-human.fa", "MT-orang.fa"), ("inv", "t-inv.fa", "q-inv.fa"), ("q2", "t2.fa", "q2.fa")])
def test_reference_test_pairs_paf_identical(engine, case, tgt, qry, seeds_on_device):
    paf, st = map_files(engine, os.path.join(DATA, tgt), os.path.join(DATA, qry), seeds_on_device=seeds_on_device)
    assert paf == open(os.path.join(GOLD, f"real_{case}_inf.paf")).read()
    assert st["n_rmq_tied"] == 0


@pytest.mark.parametrize("rechain_on_device", [1, 0, -1], ids=["rechain_on_device", "rechain_dealt_device_and_host", "rechain_on_host"])
@pytest.mark.parametrize("seeds_on_device", [1, -1], ids=["anchors_on_device", "anchors_on_host"])
def test_simulated_long_reads_paf(engine, tmp_path, seeds_on_device, rechain_on_device):
    """Everything a long-read run exercises: minimizers above mid_occ, reads on both strands, secondary hits, re-chaining of most reads
    through mg_lchain_rmq: k_rmq_fill on the device, with the reads that met a tie on the range-minimum priority redone by the host form
    (the reference's tree: ties break as they do there), or the host form for all of them."""
    meta = json.load(open(os.path.join(GOLD, "sim160.json")))
    ref, reads = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fa")
    sim_reads.simulate(ref, reads, seed=meta["seed"], n_reads=meta["n_reads"], len_lo=meta["len_lo"], len_hi=meta["len_hi"], tandem=meta["tandem"])
    assert hashlib.md5(open(ref, "rb").read()).hexdigest() == meta["ref_md5"], "simulator drifted: regenerate the golden"
    assert hashlib.md5(open(reads, "rb").read()).hexdigest() == meta["reads_md5"]
    paf, st = map_files(engine, ref, reads, seeds_on_device=seeds_on_device, rechain_on_device=rechain_on_device)
    want = open(os.path.join(GOLD, "sim160_inf.paf")).read()
    assert st["n_reads"] == meta["n_reads"] and st["n_mapped"] >= 150 and st["n_rechained"] >= 100
    if paf != want:
        g, w = paf.splitlines(), want.splitlines()
        bad = [k for k in range(min(len(g), len(w))) if g[k] != w[k]]
        raise AssertionError(f"{len(bad)} of {len(w)} PAF lines differ (got {len(g)}); first: {g[bad[0]] if bad else None} vs {w[bad[0]] if bad else None}")


@pytest.mark.parametrize("name,tgt,qry,flag", [("mt_for", "MT-human.fa", "MT-orang.fa", 0x100000), ("mt_rev", "MT-human.fa", "MT-orang.fa", 0x200000),
                                               ("inv_rev", "t-inv.fa", "q-inv.fa", 0x200000)])
def test_strand_restricted_runs(engine, name, tgt, qry, flag):
    """--for-only / --rev-only (MM_F_FOR_ONLY / MM_F_REV_ONLY reach skip_seed, map.c:220-226): the reference's PAF for the same run."""
    paf, _ = map_files(engine, os.path.join(DATA, tgt), os.path.join(DATA, qry), flag=flag)
    assert paf == open(os.path.join(GOLD, "seeds", name + ".paf")).read()


def test_reads_that_map_nowhere_and_odd_input(engine):
    """Empty reads, reads shorter than a k-mer, runs of N, lower case, random sequence, names with blanks cut by the caller: no line for
    what does not map, the same line for the same sequence in either case."""
    import numpy as np
    refs = read_fasta(os.path.join(DATA, "MT-human.fa"))
    good = read_fasta(os.path.join(DATA, "MT-orang.fa"))[0][1]
    rng = np.random.default_rng(2)
    rand = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 3000))
    reads = [("empty", b""), ("short", b"ACGTACGTAC"), ("n_only", b"N" * 500), ("random", rand), ("upper", good), ("lower", good.lower()),
             ("with_n", good[:8000] + b"N" * 50 + good[8050:])]
    with mm.SeedIndex([s for _, s in refs]) as ix:
        paf, st = mm.map_reads(engine, ix, [n for n, _ in refs], reads)
        assert mm.map_reads(engine, ix, [n for n, _ in refs], [])[0] == ""
    lines = paf.splitlines()
    by = {ln.split("\t")[0]: ln for ln in lines}
    assert set(by) == {"upper", "lower", "with_n"} and st["n_mapped"] == 3 and st["n_reads"] == 7
    assert by["upper"].split("\t")[1:] == by["lower"].split("\t")[1:]
    assert by["upper"] == open(os.path.join(GOLD, "real_mt_inf.paf")).read().strip().replace("MT_orang", "upper", 1)


def test_batches_of_different_sizes_on_one_engine_and_scratch_release(engine, tmp_path):
    """An engine keeps the largest host arrays of its mapping calls from call to call (matches, anchors, the re-chaining gathers, the spliced
    chains) without clearing them: a large batch, a small one, the large one again, and once more after mm2gb_engine_release_host_scratch -- the
    same PAF every time."""
    meta = json.load(open(os.path.join(GOLD, "sim160.json")))
    ref, reads = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fa")
    sim_reads.simulate(ref, reads, seed=meta["seed"], n_reads=meta["n_reads"], len_lo=meta["len_lo"], len_hi=meta["len_hi"], tandem=meta["tandem"])
    refs, rd = read_fasta(ref), read_fasta(reads)
    want = open(os.path.join(GOLD, "sim160_inf.paf")).read()
    names = [n for n, _ in refs]
    with mm.SeedIndex([s for _, s in refs]) as ix:
        assert mm.map_reads(engine, ix, names, rd)[0] == want
        few = mm.map_reads(engine, ix, names, rd[5:9])[0]
        assert few == "".join(ln + "\n" for ln in want.splitlines() if ln.split("\t")[0] in {n.decode() if isinstance(n, bytes) else n for n, _ in rd[5:9]})
        assert mm.map_reads(engine, ix, names, rd)[0] == want
        engine.release_host_scratch()
        assert mm.map_reads(engine, ix, names, rd[5:9])[0] == few
        assert mm.map_reads(engine, ix, names, rd)[0] == want


def test_reads_sharded_over_several_engines(engine, tmp_path):
    """mm2gb_map_reads_multi with three engines on the one GPU (the multi-GPU path without a multi-GPU box, as tests/test_gpu_pool.py does for
    the chaining calls): same PAF as one engine, in read order."""
    meta = json.load(open(os.path.join(GOLD, "sim160.json")))
    ref, reads = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fa")
    sim_reads.simulate(ref, reads, seed=meta["seed"], n_reads=meta["n_reads"], len_lo=meta["len_lo"], len_hi=meta["len_hi"], tandem=meta["tandem"])
    refs, rd = read_fasta(ref), read_fasta(reads)
    with mm.SeedIndex([s for _, s in refs]) as ix, mm.Engine() as e2, mm.Engine() as e3:
        paf, st = mm.map_reads_multi([engine, e2, e3], ix, [n for n, _ in refs], rd, opt=mm.map_opt(host_threads=12))
    assert paf == open(os.path.join(GOLD, "sim160_inf.paf")).read()
    assert st["n_reads"] == meta["n_reads"]


def test_a_run_as_a_stream_of_chunks(engine, tmp_path):
    """mm2gb_map_reads_stream: the run cut into chunks of ~600 kb of reads (about 15 of them), three engines on the one GPU taking chunks as
    they become free -- one chunk's host stages overlap another's kernels (role of worker_for's batch rotation,
    map.c:924-1153).  Same PAF as one batch, in read order; counts add up."""
    meta = json.load(open(os.path.join(GOLD, "sim160.json")))
    ref, reads = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fa")
    sim_reads.simulate(ref, reads, seed=meta["seed"], n_reads=meta["n_reads"], len_lo=meta["len_lo"], len_hi=meta["len_hi"], tandem=meta["tandem"])
    refs, rd = read_fasta(ref), read_fasta(reads)
    want = open(os.path.join(GOLD, "sim160_inf.paf")).read()
    with mm.SeedIndex([s for _, s in refs]) as ix, mm.Engine() as e2, mm.Engine() as e3:
        paf, st = mm.map_reads_stream([engine, e2, e3], ix, [n for n, _ in refs], rd, opt=mm.map_opt(host_threads=12), chunk_bases=600_000)
        assert paf == want
        assert st["n_reads"] == meta["n_reads"] and st["n_rechained"] >= 100
        paf1, _ = mm.map_reads_stream([engine], ix, [n for n, _ in refs], rd, opt=mm.map_opt(host_threads=8), chunk_bases=10**12)   # one chunk, one engine
        assert paf1 == want
        assert mm.map_reads_stream([engine, e2], ix, [n for n, _ in refs], [])[0] == ""
