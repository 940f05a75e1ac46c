"""configs[0]/[1]: the reference's own minimap2 host (built unmodified from /root/reference with -D__AMD_SPLIT_KERNELS__,
oracle/Makefile target `gpuhost`) running --gpu-chain on top of OUR libmm2gb_chain.so, PAF compared with the PAF the
reference's CPU path printed for the same inputs (tests/golden/real_*_inf.paf, made with --max-chain-skip=2147483647)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "oracle", "_ref", "minimap2_gpuhost")
GOLD = os.path.join(ROOT, "tests", "golden")
CFG = os.path.join(ROOT, "mm2-gb_amd", "mi355x_config.json")
PAIRS = {"mt": ("MT-human.fa", "MT-orang.fa"), "inv": ("t-inv.fa", "q-inv.fa"), "q2": ("t2.fa", "q2.fa")}



def _require_host():
    """These tests are the configs[0]/[1] gate: on a box with a GPU a missing host binary is a FAILURE, not a skip (it is built
    where /root/reference exists and travels with the snapshot).  MM2GB_ALLOW_NO_GPUHOST=1 turns the failure back into a skip."""
    if os.path.exists(HOST):
        return
    if os.environ.get("MM2GB_ALLOW_NO_GPUHOST") == "1":
        pytest.skip("oracle/_ref/minimap2_gpuhost not built (MM2GB_ALLOW_NO_GPUHOST=1)")
    pytest.fail("oracle/_ref/minimap2_gpuhost is missing: the PAF gate cannot run.  Build it with `make -C oracle gpuhost` where the "
                "reference checkout exists (it travels to the GPU box), or set MM2GB_ALLOW_NO_GPUHOST=1 to skip on purpose.")


@pytest.fixture(autouse=True)
def _host_binary():
    _require_host()


def needs_host(fn):        # kept as a marker of which tests drive the reference host; the autouse fixture enforces it
    return fn


def run_host(tgt, qry, *extra):
    cmd = [HOST, "-t", "1", "--gpu-chain", "--gpu-cfg", CFG, *extra, os.path.join(GOLD, "data", tgt), os.path.join(GOLD, "data", qry)]
    r = subprocess.run(cmd, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout.decode()


@needs_host
@pytest.mark.parametrize("name", sorted(PAIRS))
def test_paf_diff_empty_vs_cpu_reference(name):
    tgt, qry = PAIRS[name]
    want = open(os.path.join(GOLD, f"real_{name}_inf.paf")).read()
    got = run_host(tgt, qry)
    assert got == want


@needs_host
def test_paf_with_cigar_alignment_stage_downstream():
    """-c runs the base-level alignment on the chains we hand back: the compacted a[] / u[] must be usable downstream."""
    out = run_host("MT-human.fa", "MT-orang.fa", "-c")
    assert "cg:Z:" in out and out.split("\t")[0] == "MT_orang"


@needs_host
def test_small_micro_batches_and_multiple_batches(tmp_path):
    """Force several host batches / micro-batches: tiny max_total_n so every read is its own micro-batch."""
    import json
    cfg = json.load(open(CFG))
    cfg["max_total_n"] = 1000
    cfg["max_read"] = 1
    cfg["score_kernel"]["micro_batch"] = 1
    p = tmp_path / "tiny.json"
    p.write_text(json.dumps(cfg))
    cmd = [HOST, "-t", "1", "--gpu-chain", "--gpu-cfg", str(p), os.path.join(GOLD, "data", "t-inv.fa"), os.path.join(GOLD, "data", "q-inv.fa")]
    r = subprocess.run(cmd, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout.decode() == open(os.path.join(GOLD, "real_inv_inf.paf")).read()


@needs_host
def test_simulated_long_reads_paf_diff_empty(tmp_path):
    """160 simulated 5-60 kb reads (8 % error) on a 3 Mbp genome with interspersed and tandem repeats: real seeding, several
    host batches through chain_stream_gpu/finish_stream_gpu with the host's kalloc, post_chaining_helper's RMQ re-chain,
    then alignment-free PAF.  Expected PAF was printed by the reference CPU path (tests/golden/sim160.json)."""
    import hashlib
    import json
    import sim_reads
    meta = json.load(open(os.path.join(GOLD, "sim160.json")))
    ref, reads = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fa")
    sim_reads.simulate(ref, reads, seed=meta["seed"], n_reads=meta["n_reads"], len_lo=meta["len_lo"], len_hi=meta["len_hi"], tandem=meta["tandem"])
    assert hashlib.md5(open(ref, "rb").read()).hexdigest() == meta["ref_md5"], "simulator drifted: regenerate the golden"
    assert hashlib.md5(open(reads, "rb").read()).hexdigest() == meta["reads_md5"]
    r = subprocess.run([HOST, "-t", "1", "--gpu-chain", "--gpu-cfg", CFG, ref, reads], capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    want = open(os.path.join(GOLD, "sim160_inf.paf")).read()
    got = r.stdout.decode()
    if got != want:
        w, g = want.splitlines(), got.splitlines()
        bad = [k for k in range(min(len(w), len(g))) if w[k] != g[k]]
        raise AssertionError(f"{len(bad)} of {len(w)} PAF lines differ (got {len(g)} lines); first: {g[bad[0]] if bad else None} vs {w[bad[0]] if bad else None}")


@needs_host
@pytest.mark.parametrize("threads", [1, 3])
def test_device_post_pass_through_the_boundary(tmp_path, threads):
    """MM2GB_POST=gpu: backtrack + compaction of every batch run as kernels behind its score kernel and only chains
    come back through chain_stream_gpu / finish_stream_gpu -- no host post-pass threads.  Same PAF as the reference CPU path,
    single-threaded and with one stream id per host thread."""
    import json
    import sim_reads
    meta = json.load(open(os.path.join(GOLD, "sim160.json")))
    ref, reads = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fa")
    sim_reads.simulate(ref, reads, seed=meta["seed"], n_reads=meta["n_reads"], len_lo=meta["len_lo"], len_hi=meta["len_hi"], tandem=meta["tandem"])
    cfg = json.load(open(CFG))
    cfg["num_streams"] = threads
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps(cfg))
    env = dict(os.environ, MM2GB_POST="gpu", MM2GB_DEBUG_PHASES="1")
    r = subprocess.run([HOST, "-t", str(threads), "--gpu-chain", "--gpu-cfg", str(p), ref, reads], capture_output=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert "finish (device post-pass)" in r.stderr.decode()
    want = open(os.path.join(GOLD, "sim160_inf.paf")).read()
    if threads == 1:
        assert r.stdout.decode() == want
    else:
        assert sorted(r.stdout.decode().splitlines()) == sorted(want.splitlines())
    for name, (tgt, qry) in PAIRS.items():
        r = subprocess.run([HOST, "-t", "1", "--gpu-chain", "--gpu-cfg", CFG, os.path.join(GOLD, "data", tgt), os.path.join(GOLD, "data", qry)],
                           capture_output=True, timeout=600, env=env)
        assert r.returncode == 0 and r.stdout.decode() == open(os.path.join(GOLD, f"real_{name}_inf.paf")).read(), name


@needs_host
@pytest.mark.parametrize("form", ["host_tree", "kernel"])
def test_rmq_rechaining_through_the_library_paf_diff_empty(tmp_path, form):
    """The same host linked with -Wl,--wrap=mg_lchain_rmq (oracle/Makefile target gpuhost_rmq; sources untouched): every
    re-chaining call of post_chaining_helper (map.c:450) lands in the library.  Default: the host form that keeps the
    reference's tree rules, all 160 reads.  MM2GB_RMQ=gpu: the kernel, one read per call (far slower per call, so 48 reads); a read whose
    range-minimum meets a tie is redone by the library's own exact host form.  Neither form ever calls the host program's
    mg_lchain_rmq: the library does not import it (tests/test_host_cpu.py checks the symbol table), and the report line says so.
    Same PAF as the reference CPU path either way."""
    import json
    import re
    import sim_reads
    host_rmq = HOST + "_rmq"
    if not os.path.exists(host_rmq):
        pytest.fail("oracle/_ref/minimap2_gpuhost_rmq is missing (make -C oracle gpuhost_rmq where the reference checkout exists)")
    meta = json.load(open(os.path.join(GOLD, "sim160.json")))
    ref, reads = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fa")
    sim_reads.simulate(ref, reads, seed=meta["seed"], n_reads=meta["n_reads"], len_lo=meta["len_lo"], len_hi=meta["len_hi"], tandem=meta["tandem"])
    n_take = 160 if form == "host_tree" else 48                # reads map independently of each other
    recs = open(reads).read().split(">")[1:n_take + 1]
    some = str(tmp_path / "some.fa")
    open(some, "w").write("".join(">" + x for x in recs))
    names = {x.split()[0] for x in recs}
    env = dict(os.environ, MM2GB_RMQ_REPORT="1")
    if form == "kernel":
        env["MM2GB_RMQ"] = "gpu"
    r = subprocess.run([host_rmq, "-t", "1", "--gpu-chain", "--gpu-cfg", CFG, ref, some], capture_output=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    want = "".join(l + "\n" for l in open(os.path.join(GOLD, "sim160_inf.paf")).read().splitlines() if l.split("\t")[0] in names)
    assert want.count("\n") >= n_take and r.stdout.decode() == want
    m = re.search(r"mg_lchain_rmq calls answered by the library: (\d+), of which redone by the library's exact host form because of a tie \(device form only\): (\d+); "
                  r"handed to the host program: (\d+)", r.stderr.decode())
    assert m and int(m.group(1)) > 10, r.stderr.decode()[-500:]
    assert int(m.group(3)) == 0
    if form == "host_tree":
        assert int(m.group(2)) == 0
    undefined = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(ROOT, "mm2-gb_amd", "libmm2gb_chain.so")], capture_output=True, text=True).stdout
    assert "mg_lchain_rmq" not in undefined


@needs_host
def test_multithreaded_host_one_stream_per_thread(tmp_path):
    """-t 3 with num_streams = 3: every host thread drives its own engine/stream through the boundary at the same time
    (the reference supports -t 1 only, README.md:46-47).  Same PAF as the single-threaded CPU path."""
    import json
    import sim_reads
    meta = json.load(open(os.path.join(GOLD, "sim160.json")))
    ref, reads = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fa")
    sim_reads.simulate(ref, reads, seed=meta["seed"], n_reads=meta["n_reads"], len_lo=meta["len_lo"], len_hi=meta["len_hi"], tandem=meta["tandem"])
    cfg = json.load(open(CFG))
    cfg["num_streams"] = 3
    cfg["max_total_n"] = 300_000           # small batches so that every thread launches several
    cfg["max_read"] = 8
    p = tmp_path / "mt.json"
    p.write_text(json.dumps(cfg))
    r = subprocess.run([HOST, "-t", "3", "--gpu-chain", "--gpu-cfg", str(p), ref, reads], capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    want = sorted(open(os.path.join(GOLD, "sim160_inf.paf")).read().splitlines())
    got = sorted(r.stdout.decode().splitlines())
    assert got == want


@needs_host
@pytest.mark.parametrize("threads,post", [(1, "host"), (3, "host"), (3, "gpu")])
def test_rechaining_answered_ahead_for_the_whole_batch(tmp_path, threads, post):
    """The wrapped host at --max-chain-skip=2147483647 (the setting the path's results are defined at): chain_stream_gpu / finish_stream_gpu
    answer a batch's mg_lchain_rmq calls TOGETHER on the device before the host's callback asks for them read by read
    (csrc/rechain_ahead.cpp; map.c:444-451), and every call is then served from that batch after a byte-for-byte comparison of its input.
    Same PAF as the reference CPU path; nearly every call must have been answered ahead (a wrong guess of map.c's trigger would show here)."""
    import json
    import re
    import sim_reads
    host_rmq = HOST + "_rmq"
    if not os.path.exists(host_rmq):
        pytest.fail("oracle/_ref/minimap2_gpuhost_rmq is missing (make -C oracle gpuhost_rmq where the reference checkout exists)")
    meta = json.load(open(os.path.join(GOLD, "sim160.json")))
    ref, reads = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fa")
    sim_reads.simulate(ref, reads, seed=meta["seed"], n_reads=meta["n_reads"], len_lo=meta["len_lo"], len_hi=meta["len_hi"], tandem=meta["tandem"])
    cfg = json.load(open(CFG))
    cfg["num_streams"] = threads
    if threads > 1:
        cfg["max_total_n"] = 300_000       # small batches so that every thread hands several back
        cfg["max_read"] = 8
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps(cfg))
    env = dict(os.environ, MM2GB_REPORT="1", MM2GB_POST=post)
    env.pop("MM2GB_PRECHAIN", None)        # the library finds the wrap in the program's symbol table
    r = subprocess.run([host_rmq, "-t", str(threads), "--max-chain-skip=2147483647", "--gpu-chain", "--gpu-cfg", str(p), ref, reads],
                       capture_output=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    want = open(os.path.join(GOLD, "sim160_inf.paf")).read()
    if threads == 1:
        assert r.stdout.decode() == want
    else:
        assert sorted(r.stdout.decode().splitlines()) == sorted(want.splitlines())
    m = re.search(r"mg_lchain_rmq answered by the library [0-9.]+ s in (\d+) calls \| re-chaining ahead of the callback [0-9.]+ s for (\d+) reads, (\d+) calls answered from it",
                  r.stderr.decode())
    assert m, r.stderr.decode()[-800:]
    calls, ahead, served = int(m.group(1)), int(m.group(2)), int(m.group(3))
    assert calls > 50 and served >= 0.95 * calls and ahead >= served
    # the plain host (no --wrap) never reaches the library's re-chaining entry: nothing is answered ahead for it
    r = subprocess.run([HOST, "-t", "1", "--max-chain-skip=2147483647", "--gpu-chain", "--gpu-cfg", CFG, ref, reads], capture_output=True, timeout=900, env=env)
    assert r.returncode == 0 and r.stdout.decode() == want
    m = re.search(r"re-chaining ahead of the callback [0-9.]+ s for (\d+) reads", r.stderr.decode())
    assert m and int(m.group(1)) == 0


@needs_host
def test_rechaining_ahead_at_a_finite_max_chain_skip(tmp_path):
    """max_chain_skip below the tree's size cap (minimap2's default is 25): the reference's skip counter can end an inner walk early (lchain.c:329-333).
    Round 6: the one-anchor-per-step kernel keeps the counter, so a batch's calls are answered ahead here too -- the PAF equals the one of a run in
    which every call is answered on the spot by the host form --, and with MM2GB_RMQ_SKIP=ignore (no such walk on the device) nothing is answered ahead."""
    import json
    import re
    import sim_reads
    host_rmq = HOST + "_rmq"
    meta = json.load(open(os.path.join(GOLD, "sim160.json")))
    ref, reads = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fa")
    sim_reads.simulate(ref, reads, seed=meta["seed"], n_reads=48, len_lo=meta["len_lo"], len_hi=meta["len_hi"], tandem=meta["tandem"])
    pat = r"in (\d+) calls \| re-chaining ahead of the callback [0-9.]+ s for (\d+) reads, (\d+) calls answered from it"
    out = {}
    for flag in ("--max-chain-skip=25", "--max-chain-skip=0"):
        for mode in ("ahead", "off"):
            env = dict(os.environ, MM2GB_REPORT="1", MM2GB_PRECHAIN="1")
            if mode == "off":
                env["MM2GB_RMQ_SKIP"] = "ignore"
            r = subprocess.run([host_rmq, "-t", "1", flag, "--gpu-chain", "--gpu-cfg", CFG, ref, reads], capture_output=True, timeout=900, env=env)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            m = re.search(pat, r.stderr.decode())
            assert m and int(m.group(1)) > 10, r.stderr.decode()[-800:]
            out[flag, mode] = (r.stdout.decode(), int(m.group(1)), int(m.group(2)), int(m.group(3)))
        assert out[flag, "off"][2] == 0 and out[flag, "off"][3] == 0
        assert out[flag, "ahead"][3] >= 0.9 * out[flag, "ahead"][1]
        assert out[flag, "ahead"][0] == out[flag, "off"][0]
