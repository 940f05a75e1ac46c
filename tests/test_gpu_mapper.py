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


@pytest.mark.parametrize("seeds_on_device", [1, -1], ids=["anchors_on_device", "anchors_on_host"])
def test_simulated_long_reads_paf(engine, tmp_path, seeds_on_device):
    """Everything a long-read run exercises: minimizers above mid_occ, reads on both strands, secondary hits, re-chaining of most reads
    through mg_lchain_rmq (host form with the reference's tree: ties on the range-minimum priority break as they do there)."""
    meta = json.load(open(os.path.join(GOLD, "sim160.json")))
    ref, reads = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fa")
    sim_reads.simulate(ref, reads, seed=meta["seed"], n_reads=meta["n_reads"], len_lo=meta["len_lo"], len_hi=meta["len_hi"], tandem=meta["tandem"])
    assert hashlib.md5(open(ref, "rb").read()).hexdigest() == meta["ref_md5"], "simulator drifted: regenerate the golden"
    assert hashlib.md5(open(reads, "rb").read()).hexdigest() == meta["reads_md5"]
    paf, st = map_files(engine, ref, reads, seeds_on_device=seeds_on_device)
    want = open(os.path.join(GOLD, "sim160_inf.paf")).read()
    assert st["n_reads"] == meta["n_reads"] and st["n_mapped"] >= 150 and st["n_rechained"] >= 100
    if paf != want:
        g, w = paf.splitlines(), want.splitlines()
        bad = [k for k in range(min(len(g), len(w))) if g[k] != w[k]]
        raise AssertionError(f"{len(bad)} of {len(w)} PAF lines differ (got {len(g)}); first: {g[bad[0]] if bad else None} vs {w[bad[0]] if bad else None}")
