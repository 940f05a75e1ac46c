"""mm2-gb chaining on MI355X: thin ctypes plumbing over libmm2gb_chain.so (HIP kernels + C ABI, include/mm2gb_chain.h).

The directory name carries a hyphen, so load it with
    importlib.import_module("mm2-gb_amd")
(or use the `mm2gb_amd` shim module at the repository root).

There is no CPU fallback: if the shared library is missing or no GPU is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# An engine's copy and compute streams must not share a hardware queue (csrc/engine.hip, Engine::init): the HIP runtime reads this once, when
# it starts, so a Python user who imports the package before torch / before the first HIP call gets 16 queues without doing anything.  A value
# that is already set is the user's.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
LIB_PATH = os.environ.get("MM2GB_LIB_PATH") or os.path.join(_HERE, "libmm2gb_chain.so")   # override: A/B runs of kernel builds
INT32_MAX = 2**31 - 1


class Misc(C.Structure):
    """mm2gb_misc_t == Misc (gpu/plutils.h:33-37)."""
    _fields_ = [("max_iter", C.c_int), ("max_dist_x", C.c_int), ("max_dist_y", C.c_int), ("max_skip", C.c_int),
                ("bw", C.c_int), ("min_cnt", C.c_int), ("min_score", C.c_int), ("is_cdna", C.c_int), ("n_seg", C.c_int),
                ("chn_pen_gap", C.c_float), ("chn_pen_skip", C.c_float)]


class _RangeCfg(C.Structure):
    _fields_ = [("blockdim", C.c_int), ("cut_check_anchors", C.c_int), ("anchor_per_block", C.c_int)]


class _ScoreCfg(C.Structure):
    _fields_ = [("micro_batch", C.c_int), ("mid_blockdim", C.c_int), ("short_griddim", C.c_int), ("long_griddim", C.c_int),
                ("mid_griddim", C.c_int), ("long_seg_cutoff", C.c_int), ("mid_seg_cutoff", C.c_int)]


class Config(C.Structure):
    """mm2gb_config_t: the gpu_config.json schema (gpu/gpu_config.json)."""
    _fields_ = [("num_streams", C.c_int), ("min_n", C.c_int), ("long_seg_buffer_size", C.c_int64), ("max_total_n", C.c_int64),
                ("max_read", C.c_int), ("avg_read_n", C.c_int),
                ("has_max_total_n", C.c_int), ("has_max_read", C.c_int), ("has_avg_read_n", C.c_int),
                ("range_kernel", _RangeCfg), ("score_kernel", _ScoreCfg)]


class Stats(C.Structure):
    _fields_ = [("n_anchors", C.c_int64), ("n_reads", C.c_int64), ("n_pairs", C.c_int64), ("n_chunks", C.c_int64),
                ("n_long_chunks", C.c_int64), ("n_mid_chunks", C.c_int64), ("n_tracked_chunks", C.c_int64), ("n_clamped_blocks", C.c_int64),
                ("ms_h2d", C.c_float), ("ms_prep", C.c_float), ("ms_score", C.c_float), ("ms_d2h", C.c_float), ("ms_total", C.c_float),
                ("ms_post", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Chains(C.Structure):
    _fields_ = [("u_off", C.POINTER(C.c_int64)), ("u", C.POINTER(C.c_uint64)), ("a_off", C.POINTER(C.c_int64)), ("a", C.c_void_p)]


class RmqParam(C.Structure):
    """mm2gb_rmq_param_t == the leading arguments of mg_lchain_rmq (lchain.c:250-251)."""
    _fields_ = [("max_dist", C.c_int), ("max_dist_inner", C.c_int), ("bw", C.c_int), ("max_chn_skip", C.c_int), ("cap_rmq_size", C.c_int),
                ("min_cnt", C.c_int), ("min_sc", C.c_int), ("chn_pen_gap", C.c_float), ("chn_pen_skip", C.c_float)]


READ_DONE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_uint64), C.c_int64, C.c_void_p)


class BatcherStats(C.Structure):
    _fields_ = [("reads", C.c_int64), ("anchors", C.c_int64), ("reads_per_lane", C.c_int64 * 2), ("batches", C.c_int64 * 2),
                ("batches_per_engine", C.c_int64 * 16), ("n_engines", C.c_int)]


class Mm2gbError(RuntimeError):
    pass


_lib = None

# every symbol include/mm2gb_chain.h and include/mm2gb_plutils.h declare
CORE_SYMBOLS = ["mm2gb_last_error", "mm2gb_version", "mm2gb_config_defaults", "mm2gb_config_parse", "mm2gb_config_load",
                "mm2gb_device_count", "mm2gb_device_numa_node", "mm2gb_pin_thread_to_device", "mm2gb_numa_cpus_for_bdf", "mm2gb_engine_create", "mm2gb_engine_destroy", "mm2gb_engine_set_misc", "mm2gb_engine_device", "mm2gb_engine_split_counts", "mm2gb_engine_gang_counts", "mm2gb_has_gang_build",
                "mm2gb_engine_reserve", "mm2gb_score_host", "mm2gb_score_device", "mm2gb_engine_sync", "mm2gb_engine_stats",
                "mm2gb_engine_stream", "mm2gb_engine_last_kernel_ms", "mm2gb_chain_host", "mm2gb_chain_gpu", "mm2gb_post_device", "mm2gb_post_device_enqueue", "mm2gb_post_device_totals", "mm2gb_post_device_digest", "mm2gb_chains_free", "mm2gb_backtrack_host",
                "mm2gb_free", "mm2gb_lchain_dp", "mm2gb_synth_count", "mm2gb_synth_fill",
                "mm2gb_pool_create", "mm2gb_pool_destroy", "mm2gb_pool_size", "mm2gb_pool_device", "mm2gb_pool_set_misc",
                "mm2gb_pool_score_host", "mm2gb_pool_chain_host",
                "mm2gb_batcher_create", "mm2gb_batcher_add", "mm2gb_batcher_feed", "mm2gb_batcher_flush", "mm2gb_batcher_stats", "mm2gb_batcher_destroy",
                "mm2gb_plan_batches", "mm2gb_rmq_chain_gpu", "mm2gb_lchain_rmq", "mm2gb_lchain_rmq_counts",
                "mm2gb_sort_seeds_gpu", "mm2gb_gen_regs_gpu", "mm2gb_collect_seeds_gpu",
                "mm2gb_sketch", "mm2gb_index_build", "mm2gb_index_destroy", "mm2gb_index_size", "mm2gb_index_mid_occ", "mm2gb_collect_matches", "mm2gb_matches_free", "mm2gb_map_opt_init", "mm2gb_map_reads", "mm2gb_engine_release_host_scratch", "mm2gb_rmq_chain_host", "mm2gb_rmq_chain_host_tied", "mm2gb_rmq_chain", "mm2gb_engine_set_rmq_kernel", "mm2gb_engine_set_rmq_team_reads", "mm2gb_has_split_build", "mm2gb_collect_seeds_host", "mm2gb_map_reads_multi", "mm2gb_map_reads_stream"]
BOUNDARY_SYMBOLS = ["init_stream_gpu", "chain_stream_gpu", "finish_stream_gpu", "free_stream_gpu"]


def lib():
    """Load libmm2gb_chain.so (built in-tree by `make -C mm2-gb_amd` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mm2gbError(f"{LIB_PATH} is missing: build it with `make -C {_HERE}` (hipcc, --offload-arch=gfx950); "
                             "there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.mm2gb_last_error.restype = C.c_char_p
        L.mm2gb_version.restype = C.c_char_p
        L.mm2gb_config_defaults.argtypes = [C.POINTER(Config)]
        L.mm2gb_config_defaults.restype = None
        L.mm2gb_config_parse.argtypes = [C.c_char_p, C.POINTER(Config)]
        L.mm2gb_config_load.argtypes = [C.c_char_p, C.POINTER(Config)]
        L.mm2gb_engine_create.restype = C.c_void_p
        L.mm2gb_engine_create.argtypes = [C.POINTER(Config), C.POINTER(Misc), C.c_int]
        L.mm2gb_engine_destroy.argtypes = [C.c_void_p]
        L.mm2gb_engine_destroy.restype = None
        L.mm2gb_engine_set_misc.argtypes = [C.c_void_p, C.POINTER(Misc)]
        L.mm2gb_engine_device.argtypes = [C.c_void_p]
        L.mm2gb_engine_reserve.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        L.mm2gb_score_host.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.mm2gb_score_device.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.mm2gb_engine_sync.argtypes = [C.c_void_p]
        L.mm2gb_engine_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.mm2gb_engine_stream.argtypes = [C.c_void_p]
        L.mm2gb_engine_stream.restype = C.c_void_p
        L.mm2gb_engine_last_kernel_ms.argtypes = [C.c_void_p]
        L.mm2gb_engine_last_kernel_ms.restype = C.c_float
        L.mm2gb_chain_host.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Chains), C.POINTER(Stats)]
        L.mm2gb_chain_gpu.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(Chains), C.POINTER(Stats)]
        L.mm2gb_post_device_enqueue.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.mm2gb_post_device_totals.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_float)]
        L.mm2gb_post_device_digest.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.mm2gb_post_device.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_float)]
        L.mm2gb_chains_free.argtypes = [C.POINTER(Chains)]
        L.mm2gb_chains_free.restype = None
        L.mm2gb_backtrack_host.argtypes = [C.POINTER(Misc), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.mm2gb_free.argtypes = [C.c_void_p]
        L.mm2gb_free.restype = None
        L.mm2gb_lchain_dp.restype = C.c_void_p
        L.mm2gb_lchain_dp.argtypes = [C.c_int] * 7 + [C.c_float, C.c_float, C.c_int, C.c_int, C.c_int64, C.c_void_p,
                                                      C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_void_p]
        L.mm2gb_synth_count.restype = C.c_int64
        L.mm2gb_synth_count.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.mm2gb_synth_fill.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.mm2gb_pool_create.restype = C.c_void_p
        L.mm2gb_pool_create.argtypes = [C.POINTER(Config), C.POINTER(Misc), C.c_int, C.c_void_p]
        L.mm2gb_pool_destroy.argtypes = [C.c_void_p]
        L.mm2gb_pool_destroy.restype = None
        L.mm2gb_pool_size.argtypes = [C.c_void_p]
        L.mm2gb_pool_device.argtypes = [C.c_void_p, C.c_int]
        L.mm2gb_pool_set_misc.argtypes = [C.c_void_p, C.POINTER(Misc)]
        L.mm2gb_pool_score_host.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats), C.c_void_p]
        L.mm2gb_pool_chain_host.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Chains), C.POINTER(Stats)]
        L.mm2gb_rmq_chain_gpu.argtypes = [C.c_void_p, C.POINTER(RmqParam), C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(Chains), C.c_void_p, C.POINTER(Stats)]
        L.mm2gb_lchain_rmq.restype = C.c_void_p
        L.mm2gb_lchain_rmq.argtypes = [C.c_int] * 7 + [C.c_float, C.c_float, C.c_int64, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_void_p]
        L.mm2gb_lchain_rmq_counts.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.mm2gb_lchain_rmq_counts.restype = None
        L.mm2gb_sort_seeds_gpu.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.mm2gb_sketch.argtypes = [C.c_char_p, C.c_int32, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.mm2gb_index_build.restype = C.c_void_p
        L.mm2gb_index_build.argtypes = [C.c_int, C.c_int, C.c_int32, C.POINTER(C.c_char_p), C.c_void_p, C.c_int]
        L.mm2gb_index_destroy.restype = None
        L.mm2gb_index_destroy.argtypes = [C.c_void_p]
        L.mm2gb_index_size.restype = C.c_int64
        L.mm2gb_index_size.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.mm2gb_index_mid_occ.restype = C.c_int32
        L.mm2gb_index_mid_occ.argtypes = [C.c_void_p, C.c_float, C.c_int32, C.c_int32]
        L.mm2gb_collect_matches.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.mm2gb_matches_free.restype = None
        L.mm2gb_matches_free.argtypes = [C.c_void_p]
        L.mm2gb_collect_seeds_gpu.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mm2gb_gen_regs_gpu.argtypes = [C.c_void_p, C.c_int64, C.POINTER(Chains), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.mm2gb_batcher_create.restype = C.c_void_p
        L.mm2gb_batcher_create.argtypes = [C.POINTER(Config), C.POINTER(Misc), C.c_int, C.c_void_p, C.c_int, READ_DONE_FN, C.c_void_p]
        L.mm2gb_batcher_add.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.mm2gb_batcher_flush.argtypes = [C.c_void_p]
        L.mm2gb_batcher_stats.argtypes = [C.c_void_p, C.POINTER(BatcherStats)]
        L.mm2gb_batcher_destroy.argtypes = [C.c_void_p]
        L.mm2gb_batcher_destroy.restype = None
        L.mm2gb_plan_batches.restype = C.c_int64
        L.mm2gb_plan_batches.argtypes = [C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise Mm2gbError(lib().mm2gb_last_error().decode())


def default_misc(**kw):
    """build_misc() for map-ont / no preset (map.c:393-426; options.c:24-36; k = 15)."""
    d = dict(max_iter=5000, max_dist_x=5000, max_dist_y=5000, max_skip=INT32_MAX, bw=500, min_cnt=3, min_score=40,
             is_cdna=0, n_seg=1, chn_pen_gap=np.float32(0.8 * 0.01 * 15), chn_pen_skip=np.float32(0.0))
    d.update(kw)
    return Misc(**d)


def default_config():
    c = Config()
    lib().mm2gb_config_defaults(C.byref(c))
    return c


def parse_config(text):
    c = Config()
    _check(lib().mm2gb_config_parse(text.encode(), C.byref(c)))
    return c


def load_config(path):
    c = Config()
    _check(lib().mm2gb_config_load(os.fsencode(path), C.byref(c)))
    return c


def device_count():
    return lib().mm2gb_device_count()


def device_numa_node(device):
    """NUMA node of the device's PCIe root (-1: unknown)."""
    return lib().mm2gb_device_numa_node(int(device))


def pin_thread_to_device(device):
    """Move the calling thread (and the threads it starts afterwards) onto the usable CPUs of the device's NUMA node; 0 = nothing changed."""
    return lib().mm2gb_pin_thread_to_device(int(device))


def numa_cpus_for_bdf(bdf, sysfs_root=""):
    """(node, cpus) of a PCI device from a sysfs tree (tests hand in a made-up one)."""
    L = lib()
    L.mm2gb_numa_cpus_for_bdf.restype = C.c_int
    L.mm2gb_numa_cpus_for_bdf.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32]
    node = C.c_int32(-1)
    buf = (C.c_int32 * 4096)()
    n = L.mm2gb_numa_cpus_for_bdf(bdf.encode(), sysfs_root.encode(), C.byref(node), buf, 4096)
    return node.value, list(buf[:min(n, 4096)])


class Engine:
    """One chaining engine on one GPU (mm2gb_engine_t)."""

    def __init__(self, misc=None, config=None, device=0):
        L = lib()
        self.misc = misc if misc is not None else default_misc()
        self.config = config if config is not None else default_config()
        self.device = int(device)
        self._h = L.mm2gb_engine_create(C.byref(self.config), C.byref(self.misc), device)
        if not self._h:
            raise Mm2gbError(L.mm2gb_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().mm2gb_engine_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_misc(self, misc):
        _check(lib().mm2gb_engine_set_misc(self._h, C.byref(misc)))
        self.misc = misc

    def split_counts(self):
        """Of the last completed call: (chunks scored strip by strip with other workgroups' help, items those others took)."""
        c, h = C.c_int64(), C.c_int64()
        lib().mm2gb_engine_split_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib().mm2gb_engine_split_counts.restype = None
        lib().mm2gb_engine_split_counts(self._h, C.byref(c), C.byref(h))
        return c.value, h.value

    def gang_counts(self):
        """Of the last completed call: (chunks scored by a gang of workgroups, workgroups that started in a gang)."""
        c, h = C.c_int64(), C.c_int64()
        lib().mm2gb_engine_gang_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib().mm2gb_engine_gang_counts.restype = None
        lib().mm2gb_engine_gang_counts(self._h, C.byref(c), C.byref(h))
        return c.value, h.value

    def score(self, anchors, offsets):
        """Host buffers in, host buffers out.  anchors: (n,2) uint64; offsets: (R+1,) int64.
        Returns f int32[n], p int32[n] (distance back to the predecessor, 0 = none), stats dict."""
        a = np.ascontiguousarray(anchors, dtype=np.uint64)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        n = int(off[-1])
        assert a.shape == (n, 2)
        f = np.empty(n, dtype=np.int32)
        p = np.empty(n, dtype=np.int32)
        st = Stats()
        _check(lib().mm2gb_score_host(self._h, len(off) - 1, off.ctypes.data, a.ctypes.data, f.ctypes.data, p.ctypes.data, C.byref(st)))
        return f, p, st.as_dict()

    def score_device(self, n_reads, d_offsets, d_anchors, n_anchors, d_f, d_p):
        """Raw device pointers (ints); asynchronous on the engine's stream."""
        _check(lib().mm2gb_score_device(self._h, n_reads, d_offsets, d_anchors, n_anchors, d_f, d_p))

    def sync(self):
        _check(lib().mm2gb_engine_sync(self._h))

    def stats(self):
        st = Stats()
        _check(lib().mm2gb_engine_stats(self._h, C.byref(st)))
        return st.as_dict()

    def stream(self):
        return lib().mm2gb_engine_stream(self._h)

    def chain(self, anchors, offsets, threads=1):
        """Full chaining of a batch: returns list of (u, a_out) per read plus stats."""
        a = np.ascontiguousarray(anchors, dtype=np.uint64)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        R = len(off) - 1
        out = Chains()
        st = Stats()
        _check(lib().mm2gb_chain_host(self._h, R, off.ctypes.data, a.ctypes.data, threads, C.byref(out), C.byref(st)))
        return _take_chains(out, R), st.as_dict()


def _engine_chain_gpu(self, anchors, offsets):
    """Full chaining of a batch with backtrack + compaction on the device too (mm2gb_chain_gpu): list of (u, a_out) per read."""
    a = np.ascontiguousarray(anchors, dtype=np.uint64)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    R = len(off) - 1
    out = Chains()
    st = Stats()
    _check(lib().mm2gb_chain_gpu(self._h, R, off.ctypes.data, a.ctypes.data, C.byref(out), C.byref(st)))
    return _take_chains(out, R), st.as_dict()


Engine.chain_gpu = _engine_chain_gpu


def default_rmq_param(**kw):
    """What post_chaining_helper passes to mg_lchain_rmq for map-ont defaults (map.c:450-451), max_chain_skip = infinity."""
    d = dict(max_dist=5000, max_dist_inner=1000, bw=20000, max_chn_skip=INT32_MAX, cap_rmq_size=100000, min_cnt=3, min_sc=40,
             chn_pen_gap=np.float32(0.8 * 0.01 * 15), chn_pen_skip=np.float32(0.0))
    d.update(kw)
    return RmqParam(**d)


def _engine_rmq_chain(self, anchors, offsets, prm):
    """RMQ re-chaining of a batch on the device (mm2gb_rmq_chain_gpu): list of (u, a_out) per read, n_tied per read, stats."""
    a = np.ascontiguousarray(anchors, dtype=np.uint64)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    R = len(off) - 1
    out = Chains()
    st = Stats()
    tied = np.zeros(max(R, 1), dtype=np.int32)
    _check(lib().mm2gb_rmq_chain_gpu(self._h, C.byref(prm), R, off.ctypes.data, a.ctypes.data, C.byref(out), tied.ctypes.data, C.byref(st)))
    return _take_chains(out, R), tied[:R], st.as_dict()


Engine.rmq_chain = _engine_rmq_chain


class RmqDeal(C.Structure):
    _fields_ = [("n_device", C.c_int64), ("n_host_cost", C.c_int64), ("n_host_tie", C.c_int64), ("est_device_s", C.c_double), ("est_host_s", C.c_double),
                ("device_s", C.c_double), ("host_s", C.c_double), ("tie_s", C.c_double), ("total_s", C.c_double), ("device_kernel", C.c_int32), ("n_team", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def _engine_rmq_chain_exact(self, anchors, offsets, prm, threads=4):
    """mm2gb_rmq_chain: device and host threads at the same time, exact for every read: list of (u, a_out) per read, where each read was
    done (0 device, 1 host by cost, 2 host after a tie), the deal."""
    L = lib()
    L.mm2gb_rmq_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    a = np.ascontiguousarray(anchors, dtype=np.uint64)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    R = len(off) - 1
    out = Chains()
    where = np.zeros(max(R, 1), dtype=np.int32)
    deal = RmqDeal()
    _check(L.mm2gb_rmq_chain(self._h, C.byref(prm), R, off.ctypes.data, a.ctypes.data, int(threads), C.byref(out), where.ctypes.data, C.byref(deal)))
    return _take_chains(out, R), where[:R], deal.as_dict()


Engine.rmq_chain_exact = _engine_rmq_chain_exact


def rmq_chain_host(anchors, offsets, prm, threads=4):
    """mm2gb_rmq_chain_host: the same re-chaining on host threads (segment tree): list of (u, a_out) per read, n_tied per read."""
    L = lib()
    L.mm2gb_rmq_chain_host.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    a = np.ascontiguousarray(anchors, dtype=np.uint64)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    R = len(off) - 1
    out = Chains()
    tied = np.zeros(max(R, 1), dtype=np.int32)
    _check(L.mm2gb_rmq_chain_host(C.byref(prm), R, off.ctypes.data, a.ctypes.data, int(threads), C.byref(out), tied.ctypes.data))
    return _take_chains(out, R), tied[:R]

REG_DTYPE = np.dtype([(k, "<i4") for k in "id cnt rid score qs qe rs re parent subsc as_ mlen blen n_sub score0".split()] +
                     [("flags", "<u4"), ("hash", "<u4"), ("div", "<f4")])      # mm2gb_reg_t


def _engine_sort_seeds(self, anchors, offsets):
    """mm2gb_sort_seeds_gpu: every read's anchors sorted by x the way radix_sort_128x leaves them; returns the sorted copy."""
    a = np.ascontiguousarray(anchors, dtype=np.uint64).copy()
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    _check(lib().mm2gb_sort_seeds_gpu(self._h, len(off) - 1, off.ctypes.data, a.ctypes.data))
    return a


def _engine_collect_seeds(self, flag, reads, ref_len=None, ref_rank=None):
    """mm2gb_collect_seeds_gpu.  reads: list of dicts with seeds (n,4) uint32, hits (uint64), qlen and optionally q_rank.
    Returns one sorted anchor array (m,2) uint64 per read."""
    R = len(reads)
    seed_off = np.zeros(R + 1, np.int64)
    seed_off[1:] = np.cumsum([len(r["seeds"]) for r in reads])
    seeds = np.ascontiguousarray(np.concatenate([np.asarray(r["seeds"], np.uint32).reshape(-1, 4) for r in reads]) if R else np.zeros((0, 4), np.uint32), dtype=np.uint32)
    hit_off = np.zeros(len(seeds) + 1, np.int64)
    np.cumsum(seeds[:, 0], out=hit_off[1:])
    hits = np.ascontiguousarray(np.concatenate([np.asarray(r["hits"], np.uint64) for r in reads]) if R else np.zeros(0, np.uint64), dtype=np.uint64)
    qlen = np.ascontiguousarray([r["qlen"] for r in reads], dtype=np.int32)
    have_rank = any("q_rank" in r for r in reads)
    q_rank = np.ascontiguousarray([r.get("q_rank", 0) for r in reads], dtype=np.int32) if have_rank else None
    rl = np.ascontiguousarray(ref_len, dtype=np.int32) if ref_len is not None else None
    rr = np.ascontiguousarray(ref_rank, dtype=np.int32) if ref_rank is not None else None
    n_ref = len(rl) if rl is not None else (len(rr) if rr is not None else 0)
    a_off = np.zeros(R + 1, np.int64)
    out = np.zeros((max(len(hits), 1), 2), np.uint64)
    _check(lib().mm2gb_collect_seeds_gpu(self._h, int(flag), R, seed_off.ctypes.data, seeds.ctypes.data, hit_off.ctypes.data, hits.ctypes.data, qlen.ctypes.data,
                                         q_rank.ctypes.data if q_rank is not None else None, n_ref, rl.ctypes.data if rl is not None else None,
                                         rr.ctypes.data if rr is not None else None, a_off.ctypes.data, out.ctypes.data))
    return [out[a_off[r]:a_off[r + 1]].copy() for r in range(R)]


def _engine_gen_regs(self, chains, qlen, hashes, is_qstrand=0):
    """mm2gb_gen_regs_gpu on a list of (u, a_out) per read: list of REG_DTYPE arrays."""
    R = len(chains)
    u_off = np.zeros(R + 1, np.int64); a_off = np.zeros(R + 1, np.int64)
    u_off[1:] = np.cumsum([len(u) for u, _ in chains]); a_off[1:] = np.cumsum([len(a) for _, a in chains])
    u_all = np.ascontiguousarray(np.concatenate([u for u, _ in chains]) if R else np.zeros(0, np.uint64), dtype=np.uint64)
    a_all = np.ascontiguousarray(np.concatenate([a for _, a in chains]) if R else np.zeros((0, 2), np.uint64), dtype=np.uint64)
    ch = Chains(u_off.ctypes.data_as(C.POINTER(C.c_int64)), u_all.ctypes.data_as(C.POINTER(C.c_uint64)), a_off.ctypes.data_as(C.POINTER(C.c_int64)), a_all.ctypes.data)
    ql = np.ascontiguousarray(qlen, dtype=np.int32); hs = np.ascontiguousarray(hashes, dtype=np.uint32)
    regs = np.zeros(int(u_off[-1]), dtype=REG_DTYPE)
    _check(lib().mm2gb_gen_regs_gpu(self._h, R, C.byref(ch), ql.ctypes.data, hs.ctypes.data, int(is_qstrand), regs.ctypes.data))
    return [regs[u_off[r]:u_off[r + 1]] for r in range(R)]


Engine.sort_seeds = _engine_sort_seeds
Engine.collect_seeds = _engine_collect_seeds


def collect_seeds_host(flag, reads, ref_len=None, ref_rank=None, threads=4):
    """mm2gb_collect_seeds_host: same arguments and results as Engine.collect_seeds, on host threads."""
    L = lib()
    L.mm2gb_collect_seeds_host.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    R = len(reads)
    seed_off = np.zeros(R + 1, np.int64)
    seed_off[1:] = np.cumsum([len(r["seeds"]) for r in reads])
    seeds = np.ascontiguousarray(np.concatenate([np.asarray(r["seeds"], np.uint32).reshape(-1, 4) for r in reads]) if R else np.zeros((0, 4), np.uint32), dtype=np.uint32)
    hit_off = np.zeros(len(seeds) + 1, np.int64)
    np.cumsum(seeds[:, 0], out=hit_off[1:])
    hits = np.ascontiguousarray(np.concatenate([np.asarray(r["hits"], np.uint64) for r in reads]) if R else np.zeros(0, np.uint64), dtype=np.uint64)
    qlen = np.ascontiguousarray([r["qlen"] for r in reads], dtype=np.int32)
    have_rank = any("q_rank" in r for r in reads)
    q_rank = np.ascontiguousarray([r.get("q_rank", 0) for r in reads], dtype=np.int32) if have_rank else None
    rl = np.ascontiguousarray(ref_len, dtype=np.int32) if ref_len is not None else None
    rr = np.ascontiguousarray(ref_rank, dtype=np.int32) if ref_rank is not None else None
    n_ref = len(rl) if rl is not None else (len(rr) if rr is not None else 0)
    a_off = np.zeros(R + 1, np.int64)
    out = np.zeros((max(len(hits), 1), 2), np.uint64)
    _check(L.mm2gb_collect_seeds_host(int(flag), R, seed_off.ctypes.data, seeds.ctypes.data, hit_off.ctypes.data, hits.ctypes.data, qlen.ctypes.data,
                                      q_rank.ctypes.data if q_rank is not None else None, n_ref, rl.ctypes.data if rl is not None else None,
                                      rr.ctypes.data if rr is not None else None, int(threads), a_off.ctypes.data, out.ctypes.data))
    return [out[a_off[r]:a_off[r + 1]].copy() for r in range(R)]
Engine.gen_regs = _engine_gen_regs


def _take_chains(out, R):
    """Copy a mm2gb_chains_t into per-read (u, a_out) arrays and release it."""
    try:
        u_off = np.ctypeslib.as_array(out.u_off, shape=(R + 1,)).copy()
        a_off = np.ctypeslib.as_array(out.a_off, shape=(R + 1,)).copy()
        u_all = np.ctypeslib.as_array(out.u, shape=(int(u_off[-1]),)).copy() if u_off[-1] else np.zeros(0, np.uint64)
        a_all = (np.ctypeslib.as_array(C.cast(out.a, C.POINTER(C.c_uint64)), shape=(int(a_off[-1]), 2)).copy()
                 if a_off[-1] else np.zeros((0, 2), np.uint64))
    finally:
        lib().mm2gb_chains_free(C.byref(out))
    return [(u_all[u_off[r]:u_off[r + 1]], a_all[a_off[r]:a_off[r + 1]]) for r in range(R)]


class Pool:
    """Several engines in one process (mm2gb_pool_t): reads are dealt to the devices as contiguous runs with about the same
    number of anchors, nothing is exchanged between devices.  devices=None: every visible GPU; ids may repeat."""

    def __init__(self, devices=None, misc=None, config=None):
        L = lib()
        self.misc = misc if misc is not None else default_misc()
        self.config = config if config is not None else default_config()
        ids = np.ascontiguousarray(devices, dtype=np.int32) if devices is not None else None
        self._h = L.mm2gb_pool_create(C.byref(self.config), C.byref(self.misc), 0 if ids is None else len(ids),
                                      None if ids is None else ids.ctypes.data)
        if not self._h:
            raise Mm2gbError(L.mm2gb_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().mm2gb_pool_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __len__(self):
        return lib().mm2gb_pool_size(self._h)

    def devices(self):
        return [lib().mm2gb_pool_device(self._h, k) for k in range(len(self))]

    def set_misc(self, misc):
        _check(lib().mm2gb_pool_set_misc(self._h, C.byref(misc)))
        self.misc = misc

    def score(self, anchors, offsets):
        """Like Engine.score; also returns first_read_of_device (len(pool)+1,) = how the reads were dealt."""
        a = np.ascontiguousarray(anchors, dtype=np.uint64)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        n = int(off[-1])
        assert a.shape == (n, 2)
        f = np.empty(n, dtype=np.int32)
        p = np.empty(n, dtype=np.int32)
        first = np.zeros(len(self) + 1, dtype=np.int64)
        st = Stats()
        _check(lib().mm2gb_pool_score_host(self._h, len(off) - 1, off.ctypes.data, a.ctypes.data, f.ctypes.data, p.ctypes.data,
                                           C.byref(st), first.ctypes.data))
        return f, p, st.as_dict(), first

    def chain(self, anchors, offsets, threads=1):
        a = np.ascontiguousarray(anchors, dtype=np.uint64)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        R = len(off) - 1
        out = Chains()
        st = Stats()
        _check(lib().mm2gb_pool_chain_host(self._h, R, off.ctypes.data, a.ctypes.data, threads, C.byref(out), C.byref(st)))
        return _take_chains(out, R), st.as_dict()


def plan_batches(n_anchors, max_total_n, max_read, min_n):
    """The batcher's grouping rule alone (no GPU): (number of batches, batch id per read, lane per read)."""
    n = np.ascontiguousarray(n_anchors, dtype=np.int64)
    batch = np.empty(len(n), dtype=np.int32)
    lane = np.empty(len(n), dtype=np.int32)
    nb = lib().mm2gb_plan_batches(len(n), n.ctypes.data, max_total_n, max_read, min_n, batch.ctypes.data, lane.ctypes.data)
    if nb < 0:
        raise Mm2gbError(lib().mm2gb_last_error().decode())
    return int(nb), batch, lane


class Batcher:
    """mm2gb_batcher_t: feed reads one at a time, get every read's chains back (dict read_id -> (u, a_out))."""

    def __init__(self, devices=None, misc=None, config=None, post_threads=2, keep_results=True):
        L = lib()
        self.misc = misc if misc is not None else default_misc()
        self.config = config if config is not None else default_config()
        self.results = {}
        self.order = []
        self.n_chains = 0
        self.n_kept = 0

        def on_done(_user, read_id, n_u, u, n_a, a):
            if not keep_results:                     # rate measurements: count, do not copy
                self.n_chains += n_u
                self.n_kept += n_a
                return
            uu = np.ctypeslib.as_array(u, shape=(n_u,)).copy() if n_u else np.zeros(0, np.uint64)
            aa = (np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_uint64)), shape=(n_a, 2)).copy() if n_a else np.zeros((0, 2), np.uint64))
            self.results[read_id] = (uu, aa)
            self.order.append(read_id)

        self._cb = READ_DONE_FN(on_done)          # keep alive
        ids = np.ascontiguousarray(devices, dtype=np.int32) if devices is not None else None
        self._h = L.mm2gb_batcher_create(C.byref(self.config), C.byref(self.misc), 0 if ids is None else len(ids),
                                         None if ids is None else ids.ctypes.data, post_threads, self._cb, None)
        if not self._h:
            raise Mm2gbError(L.mm2gb_last_error().decode())

    def add(self, read_id, anchors):
        a = np.ascontiguousarray(anchors, dtype=np.uint64)
        _check(lib().mm2gb_batcher_add(self._h, read_id, a.ctypes.data, len(a)))

    def feed(self, first_id, anchors, offsets, producers=1):
        """mm2gb_batcher_feed: the reads of a packed batch added one at a time by `producers` native threads (read r gets id first_id + r)."""
        a = np.ascontiguousarray(anchors, dtype=np.uint64)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        L = lib()
        L.mm2gb_batcher_feed.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        _check(L.mm2gb_batcher_feed(self._h, len(off) - 1, int(first_id), off.ctypes.data, a.ctypes.data, int(producers)))

    def flush(self):
        _check(lib().mm2gb_batcher_flush(self._h))

    def stats(self):
        st = BatcherStats()
        _check(lib().mm2gb_batcher_stats(self._h, C.byref(st)))
        return {"reads": st.reads, "anchors": st.anchors, "reads_per_lane": list(st.reads_per_lane), "batches": list(st.batches),
                "batches_per_engine": list(st.batches_per_engine)[: st.n_engines], "n_engines": st.n_engines}

    def close(self):
        if getattr(self, "_h", None):
            lib().mm2gb_batcher_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def backtrack_host(misc, anchors, f, p_rel):
    """mm2gb_backtrack_host: backtrack + compaction for one read from f / relative p."""
    a = np.ascontiguousarray(anchors, dtype=np.uint64)
    f = np.ascontiguousarray(f, dtype=np.int32)
    p = np.ascontiguousarray(p_rel, dtype=np.int32)
    u_ptr = C.c_void_p(0)
    a_ptr = C.c_void_p(0)
    L = lib()
    n_u = L.mm2gb_backtrack_host(C.byref(misc), len(f), a.ctypes.data, f.ctypes.data, p.ctypes.data, C.byref(u_ptr), C.byref(a_ptr))
    if n_u < 0:
        raise Mm2gbError(L.mm2gb_last_error().decode())
    if n_u == 0:
        return np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64)
    u = np.ctypeslib.as_array(C.cast(u_ptr, C.POINTER(C.c_uint64)), shape=(n_u,)).copy()
    n_out = int((u & 0xffffffff).sum())
    a_out = np.ctypeslib.as_array(C.cast(a_ptr, C.POINTER(C.c_uint64)), shape=(n_out, 2)).copy()
    L.mm2gb_free(u_ptr)
    L.mm2gb_free(a_ptr)
    return u, a_out


def synth_count(seed, first_read, n_reads, len_lo, len_hi):
    """How many anchors synth_reads would make for these reads (no anchors are generated)."""
    L = lib()
    off = np.zeros(n_reads + 1, dtype=np.int64)
    n = L.mm2gb_synth_count(seed, first_read, n_reads, len_lo, len_hi, off.ctypes.data)
    if n < 0:
        raise Mm2gbError(L.mm2gb_last_error().decode())
    return int(n)


def synth_reads(seed, first_read, n_reads, len_lo, len_hi, threads=8):
    """Deterministic synthetic reads.  Returns anchors (n,2) uint64 and offsets (R+1,) int64."""
    L = lib()
    off = np.zeros(n_reads + 1, dtype=np.int64)
    n = L.mm2gb_synth_count(seed, first_read, n_reads, len_lo, len_hi, off.ctypes.data)
    if n < 0:
        raise Mm2gbError(L.mm2gb_last_error().decode())
    a = np.empty((n, 2), dtype=np.uint64)
    _check(L.mm2gb_synth_fill(seed, first_read, n_reads, len_lo, len_hi, off.ctypes.data, a.ctypes.data, threads))
    return a, off


# ---- from sequence to seed matches on the host (csrc/seeding.cpp) ------------------------------------------------------------
class SeedOpt(C.Structure):
    _fields_ = [("mid_occ", C.c_int32), ("max_max_occ", C.c_int32), ("occ_dist", C.c_int32), ("q_occ_frac", C.c_float)]


class Matches(C.Structure):
    _fields_ = [("n_seeds", C.c_int32), ("rep_len", C.c_int32), ("n_mini_pos", C.c_int32), ("pad_", C.c_int32), ("n_hits", C.c_int64),
                ("seeds", C.c_void_p), ("hits", C.c_void_p), ("mini_pos", C.c_void_p)]


def sketch(seq, w=10, k=15, rid=0):
    """mm2gb_sketch: the (w,k)-minimizers of a sequence (bytes) as an (n,2) uint64 array (x = hash << 8 | span, y = rid << 32 | pos << 1 | strand)."""
    ptr, n = C.c_void_p(), C.c_int64()
    _check(lib().mm2gb_sketch(seq, len(seq), w, k, rid, C.byref(ptr), C.byref(n)))
    out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(max(n.value, 1) * 2,))[:n.value * 2].reshape(-1, 2).copy()
    lib().mm2gb_free(ptr)
    return out


class SeedIndex:
    """Minimizer index of reference sequences (list of bytes) with the look-up semantics of the reference's mm_idx_get."""

    def __init__(self, seqs, k=15, w=10, threads=4):
        self._seqs = [bytes(s) for s in seqs]
        arr = (C.c_char_p * len(self._seqs))(*self._seqs)
        lens = np.ascontiguousarray([len(s) for s in self._seqs], dtype=np.int32)
        self.lens = lens
        self._h = lib().mm2gb_index_build(k, w, len(self._seqs), arr, lens.ctypes.data, threads)
        if not self._h:
            raise Mm2gbError(lib().mm2gb_last_error().decode())

    def close(self):
        if self._h:
            lib().mm2gb_index_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def size(self):
        occ = C.c_int64()
        return int(lib().mm2gb_index_size(self._h, C.byref(occ))), int(occ.value)

    def mid_occ(self, frac=2e-4, min_mid_occ=10, max_mid_occ=1000000):
        return int(lib().mm2gb_index_mid_occ(self._h, frac, min_mid_occ, max_mid_occ))

    def matches(self, seq, mid_occ, max_max_occ=4095, occ_dist=500, q_occ_frac=0.01):
        """mm2gb_collect_matches for one read: dict(seeds (n,4) uint32, hits uint64, qlen, rep_len, mini_pos) -- the record Engine.collect_seeds takes."""
        opt = SeedOpt(int(mid_occ), int(max_max_occ), int(occ_dist), float(q_occ_frac))
        m = Matches()
        _check(lib().mm2gb_collect_matches(self._h, bytes(seq), len(seq), C.byref(opt), C.byref(m)))
        def take(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dt).itemsize,)).view(dt).copy()
        out = dict(seeds=take(m.seeds, m.n_seeds * 4, np.uint32).reshape(-1, 4), hits=take(m.hits, m.n_hits, np.uint64), qlen=len(seq),
                   rep_len=int(m.rep_len), mini_pos=take(m.mini_pos, m.n_mini_pos, np.uint64))
        lib().mm2gb_matches_free(C.byref(m))
        return out


class MapOpt(C.Structure):
    _fields_ = [("flag", C.c_int64), ("seed", C.c_int32), ("mid_occ", C.c_int32), ("min_mid_occ", C.c_int32), ("max_mid_occ", C.c_int32),
                ("max_max_occ", C.c_int32), ("occ_dist", C.c_int32), ("mid_occ_frac", C.c_float), ("q_occ_frac", C.c_float),
                ("min_cnt", C.c_int32), ("min_chain_score", C.c_int32), ("bw", C.c_int32), ("bw_long", C.c_int32), ("max_gap", C.c_int32),
                ("max_gap_ref", C.c_int32), ("max_chain_iter", C.c_int32), ("rmq_inner_dist", C.c_int32), ("rmq_size_cap", C.c_int32),
                ("rmq_rescue_size", C.c_int32), ("rmq_rescue_ratio", C.c_float), ("chain_gap_scale", C.c_float), ("chain_skip_scale", C.c_float),
                ("mask_level", C.c_float), ("mask_len", C.c_int32), ("pri_ratio", C.c_float), ("best_n", C.c_int32), ("host_threads", C.c_int32), ("seeds_on_device", C.c_int32), ("rechain_on_device", C.c_int32)]


class MapStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("n_reads", "n_mapped", "n_anchors", "n_chains", "n_rechained", "n_rmq_tied")] + \
               [(k, C.c_double) for k in ("s_seed", "s_anchors", "s_chain", "s_rechain", "s_regs", "s_post")]

    def as_dict(self):
        return {k: (round(getattr(self, k), 4) if k.startswith("s_") else int(getattr(self, k))) for k, _ in self._fields_}


def map_opt(**kw):
    o = MapOpt()
    lib().mm2gb_map_opt_init(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def map_reads(engine, index, ref_names, reads, opt=None, k=15):
    """mm2gb_map_reads: reads = list of (name, sequence bytes); returns (PAF text, stats dict).  index: a SeedIndex of the references."""
    L = lib()
    L.mm2gb_map_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                  C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p]
    opt = opt or map_opt()
    rn = (C.c_char_p * len(ref_names))(*[n.encode() for n in ref_names])
    names = (C.c_char_p * len(reads))(*[n.encode() for n, _ in reads])
    seqs_b = [bytes(s) for _, s in reads]
    seqs = (C.c_char_p * len(reads))(*seqs_b)
    lens = np.ascontiguousarray([len(s) for s in seqs_b], dtype=np.int32)
    out, n, st = C.c_void_p(), C.c_int64(), MapStats()
    _check(L.mm2gb_map_reads(engine._h, index._h, k, rn, index.lens.ctypes.data, len(ref_names), C.byref(opt), len(reads), names, seqs, lens.ctypes.data,
                             C.byref(out), C.byref(n), C.byref(st)))
    text = C.string_at(out, n.value).decode()
    L.mm2gb_free(out)
    return text, st.as_dict()


def map_reads_multi(engines, index, ref_names, reads, opt=None, k=15):
    """mm2gb_map_reads_multi: like map_reads over a list of engines (one per device): reads shard, PAF in read order."""
    L = lib()
    L.mm2gb_map_reads_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                        C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p]
    opt = opt or map_opt()
    hs = (C.c_void_p * len(engines))(*[e._h for e in engines])
    rn = (C.c_char_p * len(ref_names))(*[n.encode() for n in ref_names])
    names = (C.c_char_p * len(reads))(*[n.encode() for n, _ in reads])
    seqs_b = [bytes(s) for _, s in reads]
    seqs = (C.c_char_p * len(reads))(*seqs_b)
    lens = np.ascontiguousarray([len(s) for s in seqs_b], dtype=np.int32)
    out, n, st = C.c_void_p(), C.c_int64(), MapStats()
    _check(L.mm2gb_map_reads_multi(hs, len(engines), index._h, k, rn, index.lens.ctypes.data, len(ref_names), C.byref(opt), len(reads), names, seqs, lens.ctypes.data,
                                   C.byref(out), C.byref(n), C.byref(st)))
    text = C.string_at(out, n.value).decode()
    L.mm2gb_free(out)
    return text, st.as_dict()


def _engine_release_host_scratch(self):
    """Give back the large host arrays the engine's mapping / re-chaining calls keep between calls (mm2gb_engine_release_host_scratch)."""
    L = lib()
    L.mm2gb_engine_release_host_scratch.argtypes = [C.c_void_p]
    _check(L.mm2gb_engine_release_host_scratch(self._h))


Engine.release_host_scratch = _engine_release_host_scratch


def map_reads_stream(engines, index, ref_names, reads, opt=None, k=15, chunk_bases=0):
    """mm2gb_map_reads_stream: a run of any size as a stream of chunks of about chunk_bases bases; every engine (several per device overlap
    host stages with kernels, engines on several devices shard the reads) takes the next chunk.  Returns (PAF text in read order, stats:
    counts summed, s_* summed over chunks)."""
    L = lib()
    L.mm2gb_map_reads_stream.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                         C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p]
    opt = opt or map_opt()
    hs = (C.c_void_p * len(engines))(*[e._h for e in engines])
    rn = (C.c_char_p * len(ref_names))(*[n.encode() for n in ref_names])
    names = (C.c_char_p * len(reads))(*[n.encode() for n, _ in reads])
    seqs_b = [s if isinstance(s, bytes) else bytes(s) for _, s in reads]
    seqs = (C.c_char_p * len(reads))(*seqs_b)
    lens = np.ascontiguousarray([len(s) for s in seqs_b], dtype=np.int32)
    out, n, st = C.c_void_p(), C.c_int64(), MapStats()
    _check(L.mm2gb_map_reads_stream(hs, len(engines), index._h, k, rn, index.lens.ctypes.data, len(ref_names), C.byref(opt), len(reads), names, seqs, lens.ctypes.data,
                                    int(chunk_bases), C.byref(out), C.byref(n), C.byref(st)))
    text = C.string_at(out, n.value).decode()
    L.mm2gb_free(out)
    return text, st.as_dict()
