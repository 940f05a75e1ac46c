"""The drop-in boundary (init/chain/finish/free_stream_gpu, gpu/plutils.h:98-104) driven the way map.c:924-1153 drives it,
without a minimap2 host: batches of chain_read_t records, deferred hand-back, several micro-batches per batch."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
mm = pytest.importorskip("mm2gb_amd")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class SeqMeta(C.Structure):      # mm_seq_meta_t, gpu/plutils.h:19-31
    _fields_ = [("i", C.c_long), ("seg_id", C.c_int), ("name", C.c_char * 200), ("len", C.c_uint32), ("n_alt", C.c_int),
                ("is_alt", C.c_int), ("qlen_sum", C.c_int)]


class ChainRead(C.Structure):    # chain_read_t, gpu/plutils.h:45-73
    _fields_ = [("seq", SeqMeta), ("qseqs", C.c_void_p), ("qlens", C.c_void_p), ("n_seg", C.c_int), ("rep_len", C.c_int),
                ("frag_gap", C.c_int), ("mini_pos", C.c_void_p), ("n_mini_pos", C.c_int), ("a", C.c_void_p), ("n", C.c_int64),
                ("u", C.c_void_p), ("n_u", C.c_int)]


libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
libc.free.argtypes = [C.c_void_p]


def test_struct_layout_matches_reference_header():
    # offsets a C compiler gives gpu/plutils.h on x86-64 (checked against the reference header in the dev container)
    assert C.sizeof(SeqMeta) == 232 and SeqMeta.name.offset == 12 and SeqMeta.len.offset == 212 and SeqMeta.qlen_sum.offset == 224
    assert ChainRead.a.offset == 280 and ChainRead.n.offset == 288 and ChainRead.u.offset == 296 and ChainRead.n_u.offset == 304
    assert C.sizeof(ChainRead) == 312


def make_batch(reads):
    arr = (ChainRead * len(reads))()
    for k, a in enumerate(reads):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        buf = libc.malloc(max(a.nbytes, 16))
        C.memmove(buf, a.ctypes.data, a.nbytes)
        arr[k].a, arr[k].n, arr[k].n_seg = buf, len(a), 1
        arr[k].seq.i = k
    return arr


def collect(ptr, n):
    out = []
    arr = C.cast(ptr, C.POINTER(ChainRead))
    for k in range(n):
        r = arr[k]
        if r.n_u > 0:
            u = np.ctypeslib.as_array(C.cast(r.u, C.POINTER(C.c_uint64)), shape=(r.n_u,)).copy()
            na = int((u & 0xffffffff).sum())
            a = np.ctypeslib.as_array(C.cast(r.a, C.POINTER(C.c_uint64)), shape=(na, 2)).copy()
            libc.free(r.u)
            libc.free(r.a)
        else:
            assert not r.a, "a must be 0 when nothing chains (plchain.cu:134-137)"
            u, a = np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64)
        out.append((r.seq.i, u, a))
    return out


@pytest.mark.parametrize("max_total_n,micro_batch", [(500_000_000, 4), (40_000, 2)])
def test_batched_deferred_protocol(tmp_path, max_total_n, micro_batch):
    L = mm.lib()
    cfg = json.load(open(os.path.join(ROOT, "mm2-gb_amd", "mi355x_config.json")))
    cfg["max_total_n"], cfg["score_kernel"]["micro_batch"], cfg["max_read"] = max_total_n, micro_batch, 1000
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cfg))
    misc = mm.default_misc()
    prm = orc.default_param()
    mt, mr, mn = C.c_size_t(0), C.c_int(0), C.c_int(0)
    L.init_stream_gpu.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, mm.Misc]
    L.chain_stream_gpu.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_void_p]
    L.finish_stream_gpu.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_void_p]
    L.free_stream_gpu.argtypes = [C.c_int]
    L.init_stream_gpu(C.byref(mt), C.byref(mr), C.byref(mn), str(path).encode(), misc)
    assert mt.value == max_total_n * micro_batch and mr.value == 1000 * micro_batch and mn.value == cfg["min_n"]

    anchors, off = mm.synth_reads(31, 0, 30, 5_000, 40_000)
    reads = [anchors[off[r]:off[r + 1]] for r in range(30)] + [np.zeros((0, 2), np.uint64)]    # and one read without anchors
    batches = [reads[0:9], reads[9:10], reads[10:24], reads[24:31]]
    expect = {}
    done = []
    keep_alive = []
    base = 0
    for b in batches:
        arr = make_batch(b)
        for k in range(len(b)):
            arr[k].seq.i = base + k
            expect[base + k] = orc.lchain_dp(b[k], prm, want_fp=False) if len(b[k]) else dict(u=np.zeros(0, np.uint64), a_out=np.zeros((0, 2), np.uint64))
        base += len(b)
        keep_alive.append(arr)
        ptr, n = C.c_void_p(C.addressof(arr)), C.c_int(len(b))
        L.chain_stream_gpu(None, None, C.byref(ptr), C.byref(n), 0, None)
        # first call hands back nothing; later calls hand back the previous batch, complete
        if len(keep_alive) == 1:
            assert not ptr.value and n.value == 0
        else:
            assert ptr.value == C.addressof(keep_alive[-2]) and n.value == len(keep_alive[-2])
            done += collect(ptr.value, n.value)
    ptr, n = C.c_void_p(0), C.c_int(0)
    L.finish_stream_gpu(None, None, C.byref(ptr), C.byref(n), 0, None)
    assert ptr.value == C.addressof(keep_alive[-1]) and n.value == len(batches[-1])
    done += collect(ptr.value, n.value)
    L.finish_stream_gpu(None, None, C.byref(ptr), C.byref(n), 0, None)          # nothing in flight any more
    assert not ptr.value and n.value == 0
    L.free_stream_gpu(1)
    assert sorted(i for i, _, _ in done) == list(range(31))
    for i, u, a in done:
        assert np.array_equal(u, expect[i]["u"]) and np.array_equal(a, expect[i]["a_out"]), f"read {i}"


def drive_boundary(tmp_path, batches, cfg_edit=None, env=None):
    """Push `batches` (lists of anchor arrays) through init / chain / finish / free_stream_gpu on stream id 0 and check every read's
    chains against the oracle."""
    L = mm.lib()
    cfg = json.load(open(os.path.join(ROOT, "mm2-gb_amd", "mi355x_config.json")))
    cfg.update(cfg_edit or {})
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cfg))
    prm = orc.default_param()
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        mt, mr, mn = C.c_size_t(0), C.c_int(0), C.c_int(0)
        L.init_stream_gpu.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, mm.Misc]
        L.chain_stream_gpu.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_void_p]
        L.finish_stream_gpu.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_void_p]
        L.free_stream_gpu.argtypes = [C.c_int]
        L.init_stream_gpu(C.byref(mt), C.byref(mr), C.byref(mn), str(path).encode(), mm.default_misc())
        expect, done, keep_alive, base = {}, [], [], 0
        for b in batches:
            arr = make_batch(b)
            for k in range(len(b)):
                arr[k].seq.i = base + k
                expect[base + k] = orc.lchain_dp(b[k], prm, want_fp=False) if len(b[k]) else dict(u=np.zeros(0, np.uint64), a_out=np.zeros((0, 2), np.uint64))
            base += len(b)
            keep_alive.append(arr)
            ptr, n = C.c_void_p(C.addressof(arr)), C.c_int(len(b))
            L.chain_stream_gpu(None, None, C.byref(ptr), C.byref(n), 0, None)
            if len(keep_alive) > 1:
                assert ptr.value == C.addressof(keep_alive[-2]) and n.value == len(keep_alive[-2])
                done += collect(ptr.value, n.value)
        ptr, n = C.c_void_p(0), C.c_int(0)
        L.finish_stream_gpu(None, None, C.byref(ptr), C.byref(n), 0, None)
        done += collect(ptr.value, n.value)
        L.free_stream_gpu(1)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert sorted(i for i, _, _ in done) == list(range(base))
    for i, u, a in done:
        assert np.array_equal(u, expect[i]["u"]) and np.array_equal(a, expect[i]["a_out"]), f"read {i}"


def test_device_post_pass_with_growing_batches(tmp_path):
    """MM2GB_POST=gpu, every batch much larger than all before it: batch k+1 is launched -- and sizes its result
    set -- while batch k's chains still wait, unfetched, in the other set.  Growing the arenas for k+1 must not free or move k's
    results (they did: set 0's buffers were re-allocated by every growth)."""
    anchors, off = mm.synth_reads(77, 0, 63, 4_000, 30_000)
    reads = [anchors[off[r]:off[r + 1]] for r in range(63)]
    batches = [reads[0:1], reads[1:3], reads[3:7], reads[7:15], reads[15:31], reads[31:63]]
    sizes = [sum(len(a) for a in b) for b in batches]
    assert all(sizes[k + 1] > 1.5 * sizes[k] for k in range(len(sizes) - 1))
    drive_boundary(tmp_path, batches, env={"MM2GB_POST": "gpu"})


def test_device_post_pass_mixed_with_multi_micro_batch_batches(tmp_path):
    """MM2GB_POST=gpu with a small max_total_n: batches that need several micro-batches take the host post-pass (scores copied out
    of the two staging sets on the D2H stream), single-micro-batch ones the device post-pass, which writes scores into the same
    staging sets (the kernels must wait for the earlier copy-out of the set they reuse)."""
    anchors, off = mm.synth_reads(78, 0, 40, 4_000, 30_000)
    reads = [anchors[off[r]:off[r + 1]] for r in range(40)]
    cap = max(len(a) for a in reads) + 1
    batches = [reads[0:9], reads[9:10], reads[10:20], reads[20:21], reads[21:22], reads[22:33], reads[33:34], reads[34:40]]
    drive_boundary(tmp_path, batches, cfg_edit={"max_total_n": cap, "max_read": 1000}, env={"MM2GB_POST": "gpu"})


def test_auto_sized_batch_limits(tmp_path):
    """A config without max_total_n / max_read (only avg_read_n, plmem.cu:497-539): limits are derived from the device's
    memory and this engine's per-anchor footprint."""
    L = mm.lib()
    cfg = json.load(open(os.path.join(ROOT, "mm2-gb_amd", "mi355x_config.json")))
    del cfg["max_total_n"], cfg["max_read"]
    cfg["avg_read_n"] = 20000
    path = tmp_path / "auto.json"
    path.write_text(json.dumps(cfg))
    mt, mr, mn = C.c_size_t(0), C.c_int(0), C.c_int(0)
    L.init_stream_gpu.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, mm.Misc]
    L.free_stream_gpu.argtypes = [C.c_int]
    L.init_stream_gpu(C.byref(mt), C.byref(mr), C.byref(mn), str(path).encode(), mm.default_misc())
    per_mb = mt.value // cfg["score_kernel"]["micro_batch"]
    assert 100_000_000 <= per_mb <= 2_000_000_000          # 288 GB device: capped at 2 G anchors per micro-batch
    assert mr.value >= per_mb // 20000
    L.free_stream_gpu(1)


def test_parked_streams_are_taken_over_by_the_next_init(tmp_path, capfd):
    """free_stream_gpu parks the streams (nothing released: page-locked staging and arenas are seconds to give back and to make again);
    init_stream_gpu with the same configuration takes them over as they are, another configuration replaces them.  Results are the same
    either way, and the second run of the same configuration makes no stream anew."""
    env_dbg = os.environ.get("MM2GB_DEBUG_PHASES")
    os.environ["MM2GB_DEBUG_PHASES"] = "1"
    try:
        anchors, off = mm.synth_reads(77, 0, 12, 5_000, 30_000)
        reads = [anchors[off[r]:off[r + 1]] for r in range(12)]
        for cfg_edit in (None, None, {"max_read": 777}):
            drive_boundary(tmp_path, [reads[:7], reads[7:]], cfg_edit=cfg_edit)
        err = capfd.readouterr().err
        made = [int(x) for x in re.findall(r"init_stream_gpu: entered at epoch [0-9.]+, returns after [0-9.]+ s: \d+ stream\(s\), (\d+) of them still being made", err)]
        assert len(made) == 3 and err.count("streams parked") == 3
        # made | taken over as they were | another configuration: made anew.  (How many engines of a fresh set are still being made when
        # init returns depends on how fast their maker thread is, and is none under MM2GB_INIT=wait: only the takeover is a fixed number.)
        assert made[1] == 0, made
    finally:
        if env_dbg is None:
            del os.environ["MM2GB_DEBUG_PHASES"]
        else:
            os.environ["MM2GB_DEBUG_PHASES"] = env_dbg


def test_parked_streams_are_released_after_the_grace_period(tmp_path, capfd):
    """Parked streams do not stay for ever: with nobody taking them over within MM2GB_PARK_SECONDS the library releases them (arenas, page-locked
    staging, engines); the next init_stream_gpu makes its streams anew and the results are the same."""
    import time
    saved = {k: os.environ.get(k) for k in ("MM2GB_DEBUG_PHASES", "MM2GB_PARK_SECONDS")}
    os.environ["MM2GB_DEBUG_PHASES"] = "1"
    os.environ["MM2GB_PARK_SECONDS"] = "0.2"
    try:
        anchors, off = mm.synth_reads(78, 0, 8, 5_000, 20_000)
        reads = [anchors[off[r]:off[r + 1]] for r in range(8)]
        first = drive_boundary(tmp_path, [reads[:5], reads[5:]])
        time.sleep(1.5)                                      # the grace period passes: the reaper lets everything go
        second = drive_boundary(tmp_path, [reads[:5], reads[5:]])
        assert first is None or first == second
        err = capfd.readouterr().err
        assert "parked stream(s) released: nobody took them over" in err
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
