"""ctypes bindings for the CPU oracle (oracle/libchain_oracle.so) and, when built, the reference itself
(oracle/_ref/libmm2ref.so + the capture hooks).  Test infrastructure only: imported from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the product package."""
import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
INT32_MAX = 2**31 - 1


class Param(C.Structure):
    """orc_param_t (oracle/chain_oracle.h) == the argument list of mg_lchain_dp (lchain.c:148-149)."""
    _fields_ = [("max_dist_x", C.c_int32), ("max_dist_y", C.c_int32), ("bw", C.c_int32),
                ("max_skip", C.c_int32), ("max_iter", C.c_int32),
                ("min_cnt", C.c_int32), ("min_sc", C.c_int32),
                ("pen_gap", C.c_float), ("pen_skip", C.c_float),
                ("is_cdna", C.c_int32), ("n_seg", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in
                ("n_pairs", "n_scored", "n_clamped", "n_rescue_eval", "n_rescue_taken", "n_rescan")]


class RmqParam(C.Structure):
    """orc_rmq_param_t == the leading arguments of mg_lchain_rmq (lchain.c:250-251)."""
    _fields_ = [("max_dist", C.c_int32), ("max_dist_inner", C.c_int32), ("bw", C.c_int32), ("max_chn_skip", C.c_int32),
                ("cap_rmq_size", C.c_int32), ("min_cnt", C.c_int32), ("min_sc", C.c_int32), ("pen_gap", C.c_float), ("pen_skip", C.c_float)]


def default_rmq_param(**kw):
    """What post_chaining_helper passes for map-ont defaults (map.c:450-451, options.c:24-55, k = 15), max_chain_skip = infinity."""
    d = dict(max_dist=5000, max_dist_inner=1000, bw=20000, max_chn_skip=INT32_MAX, cap_rmq_size=100000, min_cnt=3, min_sc=40,
             pen_gap=np.float32(0.8 * 0.01 * 15), pen_skip=np.float32(0.0))
    d.update(kw)
    return RmqParam(**d)


class Reg(C.Structure):
    """orc_reg_t: the leading 72 bytes of mm_reg1_t (minimap.h:104-119)."""
    _fields_ = [(k, C.c_int32) for k in "id cnt rid score qs qe rs re parent subsc as_ mlen blen n_sub score0".split()] + \
               [("flags", C.c_uint32), ("hash", C.c_uint32), ("div", C.c_float)]


REG_DTYPE = np.dtype([(k, "<i4") for k in "id cnt rid score qs qe rs re parent subsc as_ mlen blen n_sub score0".split()] +
                     [("flags", "<u4"), ("hash", "<u4"), ("div", "<f4")])


def default_param(**kw):
    """map-ont / no-preset defaults (options.c:24-36, map.c:408-409 with k=15), max_skip = infinity."""
    d = dict(max_dist_x=5000, max_dist_y=5000, bw=500, max_skip=INT32_MAX, max_iter=5000,
             min_cnt=3, min_sc=40, pen_gap=np.float32(0.8 * 0.01 * 15), pen_skip=np.float32(0.0),
             is_cdna=0, n_seg=1)
    d.update(kw)
    return Param(**d)


def param_to_dict(p):
    return {k: getattr(p, k) for k, _ in Param._fields_}


_lib = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "libchain_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.orc_pair_score.restype = C.c_int32
        L.orc_pair_score.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Param)]
        L.orc_log2_approx.restype = C.c_float
        L.orc_log2_approx.argtypes = [C.c_float]
        L.orc_chain_fill.restype = None
        L.orc_chain_fill.argtypes = [C.POINTER(Param), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.orc_radix_sort_x.restype = None
        L.orc_radix_sort_x.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_lchain_dp.restype = C.c_void_p
        L.orc_lchain_dp.argtypes = [C.POINTER(Param), C.c_int64, C.c_void_p, C.POINTER(C.c_int32),
                                    C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.orc_backtrack.restype = C.c_void_p
        L.orc_backtrack.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_compact.restype = C.c_void_p
        L.orc_compact.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.orc_chain_fill_reads_mt.restype = C.c_int64
        L.orc_chain_fill_reads_mt.argtypes = [C.POINTER(Param), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_free.restype = None
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_gen_regs.restype = None
        L.orc_gen_regs.argtypes = [C.c_uint32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_rmq_fill.restype = C.c_int64
        L.orc_rmq_fill.argtypes = [C.POINTER(RmqParam), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.orc_lchain_rmq.restype = C.c_void_p
        L.orc_lchain_rmq.argtypes = [C.POINTER(RmqParam), C.c_int64, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_void_p),
                                     C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.orc_collect_seeds.restype = C.c_int64
        L.orc_collect_seeds.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def as_anchors(a):
    """-> C-contiguous uint64 array of shape (n, 2): columns x, y (mm128_t layout)."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.ndim == 2 and a.shape[1] == 2
    return a


def pair_score(cur, prev, prm):
    cur = np.ascontiguousarray(cur, dtype=np.uint64)
    prev = np.ascontiguousarray(prev, dtype=np.uint64)
    return lib().orc_pair_score(cur.ctypes.data, prev.ctypes.data, C.byref(prm))


def chain_fill(a, prm):
    """f[], p[] of lchain.c:169-207 for one read.  Returns (f int32[n], p int64[n], stats dict)."""
    a = as_anchors(a)
    n = a.shape[0]
    f = np.empty(n, dtype=np.int32)
    p = np.empty(n, dtype=np.int64)
    st = Stats()
    lib().orc_chain_fill(C.byref(prm), n, a.ctypes.data, f.ctypes.data, p.ctypes.data, C.byref(st))
    return f, p, {k: getattr(st, k) for k, _ in Stats._fields_}


def lchain_dp(a, prm, want_fp=True):
    """Full mg_lchain_dp restatement.  Returns dict(u, a_out, f, p, stats)."""
    a = as_anchors(a)
    n = a.shape[0]
    f = np.empty(n, dtype=np.int32)
    p = np.empty(n, dtype=np.int64)
    n_u = C.c_int32(0)
    u_ptr = C.c_void_p(0)
    st = Stats()
    L = lib()
    out = L.orc_lchain_dp(C.byref(prm), n, a.ctypes.data, C.byref(n_u), C.byref(u_ptr),
                          f.ctypes.data if want_fp else None, p.ctypes.data if want_fp else None, C.byref(st))
    nu = n_u.value
    if nu > 0:
        u = np.ctypeslib.as_array(C.cast(u_ptr, C.POINTER(C.c_uint64)), shape=(nu,)).copy()
        n_out = int((u & 0xffffffff).sum())
        a_out = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(n_out, 2)).copy()
    else:
        u = np.zeros(0, dtype=np.uint64)
        a_out = np.zeros((0, 2), dtype=np.uint64)
    if u_ptr.value:
        L.orc_free(u_ptr)
    if out:
        L.orc_free(out)
    return dict(u=u, a_out=a_out, f=f if want_fp else None, p=p if want_fp else None,
                stats={k: getattr(st, k) for k, _ in Stats._fields_})


def backtrack_compact(a, f, p, prm):
    """mg_chain_backtrack + compact_a restatement on given f/p (p: int64 absolute, -1 none)."""
    a = as_anchors(a)
    n = a.shape[0]
    f = np.ascontiguousarray(f, dtype=np.int32)
    p = np.ascontiguousarray(p, dtype=np.int64)
    v = np.empty(max(n, 1), dtype=np.int32)
    n_u = C.c_int32(0)
    n_v = C.c_int32(0)
    max_drop = INT32_MAX if prm.is_cdna else prm.bw
    L = lib()
    u_ptr = L.orc_backtrack(n, f.ctypes.data, p.ctypes.data, v.ctypes.data, prm.min_cnt, prm.min_sc, max_drop,
                            C.byref(n_u), C.byref(n_v))
    if n_u.value == 0:
        return np.zeros(0, dtype=np.uint64), np.zeros((0, 2), dtype=np.uint64)
    out = L.orc_compact(n_u.value, u_ptr, n_v.value, v.ctypes.data, a.ctypes.data)
    u = np.ctypeslib.as_array(C.cast(u_ptr, C.POINTER(C.c_uint64)), shape=(n_u.value,)).copy()
    a_out = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(n_v.value, 2)).copy()
    L.orc_free(u_ptr)
    L.orc_free(out)
    return u, a_out


def lchain_rmq(a, prm):
    """Full mg_lchain_rmq restatement.  Returns dict(u, a_out, f, p, n_tied)."""
    a = as_anchors(a)
    n = a.shape[0]
    f = np.empty(n, dtype=np.int32)
    p = np.empty(n, dtype=np.int64)
    n_u = C.c_int32(0)
    u_ptr = C.c_void_p(0)
    tied = C.c_int64(0)
    L = lib()
    out = L.orc_lchain_rmq(C.byref(prm), n, a.ctypes.data, C.byref(n_u), C.byref(u_ptr), f.ctypes.data, p.ctypes.data, C.byref(tied))
    nu = n_u.value
    if nu > 0:
        u = np.ctypeslib.as_array(C.cast(u_ptr, C.POINTER(C.c_uint64)), shape=(nu,)).copy()
        n_out = int((u & 0xffffffff).sum())
        a_out = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(n_out, 2)).copy()
    else:
        u = np.zeros(0, dtype=np.uint64)
        a_out = np.zeros((0, 2), dtype=np.uint64)
    if u_ptr.value:
        L.orc_free(u_ptr)
    if out:
        L.orc_free(out)
    L.orc_rmq_last_ties_that_decide.restype = C.c_int64
    return dict(u=u, a_out=a_out, f=f, p=p, n_tied=tied.value, n_decide=int(L.orc_rmq_last_ties_that_decide()))


def gen_regs(u, a_out, qlen, hash_, is_qstrand=0):
    """mm_gen_regs restatement: structured array (REG_DTYPE), one record per chain."""
    u = np.ascontiguousarray(u, dtype=np.uint64)
    a = as_anchors(a_out)
    r = np.zeros(len(u), dtype=REG_DTYPE)
    if len(u):
        lib().orc_gen_regs(int(hash_) & 0xffffffff, int(qlen), len(u), u.ctypes.data, a.ctypes.data, int(is_qstrand), r.ctypes.data)
    return r


def ref_gen_regs(u, a_out, qlen, hash_, is_qstrand=0):
    """The REFERENCE's mm_gen_regs (hit.c:52, from oracle/_ref/libminimap2ref.so): the leading 72 bytes of every mm_reg1_t."""
    ref = C.CDLL(os.path.join(REF_DIR, "libminimap2ref.so"))
    ref.mm_gen_regs.restype = C.c_void_p
    ref.mm_gen_regs.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    u = np.ascontiguousarray(u, dtype=np.uint64).copy()
    a = as_anchors(a_out).copy()
    r = np.zeros(len(u), dtype=REG_DTYPE)
    if len(u) == 0:
        return r
    out = ref.mm_gen_regs(None, int(hash_) & 0xffffffff, int(qlen), len(u), u.ctypes.data, a.ctypes.data, int(is_qstrand))
    raw = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(len(u) * 80,)).reshape(len(u), 80)     # sizeof(mm_reg1_t) = 80
    r[:] = np.frombuffer(np.ascontiguousarray(raw[:, :72]).tobytes(), dtype=REG_DTYPE)
    _libc.free(out)
    return r


def radix_sort_x(arr):
    arr = as_anchors(arr).copy()
    lib().orc_radix_sort_x(arr.ctypes.data, arr.ctypes.data + arr.nbytes)
    return arr


def chain_fill_many(anchors, offsets, prm, threads=1):
    """cpu_baseline helper: orc_chain_fill per read on `threads` pthreads inside the oracle library.
    Returns (f, p_local, total_pairs)."""
    anchors = as_anchors(anchors)
    offsets = np.asarray(offsets, dtype=np.int64)
    n = anchors.shape[0]
    f = np.empty(n, dtype=np.int32)
    p = np.empty(n, dtype=np.int64)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    pairs = lib().orc_chain_fill_reads_mt(C.byref(prm), len(off) - 1, off.ctypes.data, anchors.ctypes.data, f.ctypes.data, p.ctypes.data,
                                          max(1, int(threads)))
    return f, p, pairs


# ---------------------------------------------------------------- the reference itself (dev container only)

def ref_available():
    return os.path.exists(os.path.join(REF_DIR, "libmm2ref.so")) and os.path.exists(os.path.join(REF_DIR, "libcapture.so"))


_ref = None


def ref_libs():
    """(capture, ref) -- capture hooks loaded RTLD_GLOBAL first so the reference's PLT calls land in them."""
    global _ref
    if _ref is None:
        ref_path = os.path.join(REF_DIR, "libmm2ref.so")
        os.environ["MM2GB_REF_LIB"] = ref_path
        cap = C.CDLL(os.path.join(REF_DIR, "libcapture.so"), mode=C.RTLD_GLOBAL)
        ref = C.CDLL(ref_path)
        cap.cap_last_n.restype = C.c_int64
        cap.cap_last_f.restype = C.POINTER(C.c_int32)
        cap.cap_last_p.restype = C.POINTER(C.c_int64)
        # call through the hook's mg_lchain_dp: it arms the f/p recorder and forwards to the reference
        cap.mg_lchain_dp.restype = C.c_void_p
        cap.mg_lchain_dp.argtypes = [C.c_int] * 7 + [C.c_float, C.c_float, C.c_int, C.c_int, C.c_int64, C.c_void_p,
                                                     C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_void_p]
        ref.radix_sort_128x.restype = None
        ref.radix_sort_128x.argtypes = [C.c_void_p, C.c_void_p]
        _ref = (cap, ref)
    return _ref


_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]
_libc.free.argtypes = [C.c_void_p]


def ref_lchain_dp(a, prm):
    """Run the REFERENCE's mg_lchain_dp (lchain.c:148) on a copy of `a`; f/p come from the capture hooks."""
    cap, ref = ref_libs()
    a = as_anchors(a)
    n = a.shape[0]
    if n == 0:
        return dict(u=np.zeros(0, np.uint64), a_out=np.zeros((0, 2), np.uint64), f=np.zeros(0, np.int32), p=np.zeros(0, np.int64))
    buf = _libc.malloc(a.nbytes)           # the reference frees its input with kfree(0, a) == free(a)
    C.memmove(buf, a.ctypes.data, a.nbytes)
    n_u = C.c_int(0)
    u_ptr = C.c_void_p(0)
    os.environ.pop("MM2GB_CAPTURE", None)
    out = cap.mg_lchain_dp(prm.max_dist_x, prm.max_dist_y, prm.bw, prm.max_skip, prm.max_iter, prm.min_cnt, prm.min_sc,
                           prm.pen_gap, prm.pen_skip, prm.is_cdna, prm.n_seg, n, buf, C.byref(n_u), C.byref(u_ptr), None)
    assert cap.cap_last_n() == n
    f = np.ctypeslib.as_array(cap.cap_last_f(), shape=(n,)).copy()
    p = np.ctypeslib.as_array(cap.cap_last_p(), shape=(n,)).copy()
    nu = n_u.value
    if nu > 0:
        u = np.ctypeslib.as_array(C.cast(u_ptr, C.POINTER(C.c_uint64)), shape=(nu,)).copy()
        n_out = int((u & 0xffffffff).sum())
        a_out = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(n_out, 2)).copy()
        _libc.free(out)
    else:
        u = np.zeros(0, dtype=np.uint64)
        a_out = np.zeros((0, 2), dtype=np.uint64)
    if u_ptr.value:
        _libc.free(u_ptr)
    return dict(u=u, a_out=a_out, f=f, p=p)


def ref_lchain_rmq(a, prm):
    """Run the REFERENCE's mg_lchain_rmq (lchain.c:250) on a copy of `a`; f/p come from the capture hooks."""
    cap, ref = ref_libs()
    a = as_anchors(a)
    n = a.shape[0]
    if n == 0:
        return dict(u=np.zeros(0, np.uint64), a_out=np.zeros((0, 2), np.uint64), f=np.zeros(0, np.int32), p=np.zeros(0, np.int64))
    ref.mg_lchain_rmq.restype = C.c_void_p
    ref.mg_lchain_rmq.argtypes = [C.c_int] * 7 + [C.c_float, C.c_float, C.c_int64, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_void_p]
    cap.cap_arm.argtypes = [C.c_int]
    buf = _libc.malloc(a.nbytes)           # consumed by the reference
    C.memmove(buf, a.ctypes.data, a.nbytes)
    n_u = C.c_int(0)
    u_ptr = C.c_void_p(0)
    cap.cap_arm(1)
    out = ref.mg_lchain_rmq(prm.max_dist, prm.max_dist_inner, prm.bw, prm.max_chn_skip, prm.cap_rmq_size, prm.min_cnt, prm.min_sc,
                            prm.pen_gap, prm.pen_skip, n, buf, C.byref(n_u), C.byref(u_ptr), None)
    cap.cap_arm(0)
    assert cap.cap_last_n() == n
    f = np.ctypeslib.as_array(cap.cap_last_f(), shape=(n,)).copy()
    p = np.ctypeslib.as_array(cap.cap_last_p(), shape=(n,)).copy()
    nu = n_u.value
    if nu > 0:
        u = np.ctypeslib.as_array(C.cast(u_ptr, C.POINTER(C.c_uint64)), shape=(nu,)).copy()
        n_out = int((u & 0xffffffff).sum())
        a_out = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(n_out, 2)).copy()
        _libc.free(out)
    else:
        u = np.zeros(0, dtype=np.uint64)
        a_out = np.zeros((0, 2), dtype=np.uint64)
    if u_ptr.value:
        _libc.free(u_ptr)
    return dict(u=u, a_out=a_out, f=f, p=p)


def ref_radix_sort(arr):
    _, ref = ref_libs()
    arr = as_anchors(arr).copy()
    ref.radix_sort_128x(arr.ctypes.data, arr.ctypes.data + arr.nbytes)
    return arr


def read_capture(path):
    """Parse the records written by oracle/capture_hooks.c."""
    recs = []
    with open(path, "rb") as fh:
        blob = fh.read()
    off = 0
    while off < len(blob):
        assert blob[off:off + 7] == b"MMCAP01", "bad capture magic"
        off += 8
        ints = np.frombuffer(blob, dtype="<i4", count=9, offset=off); off += 36
        fl = np.frombuffer(blob, dtype="<f4", count=2, offset=off); off += 8
        have_fp = int(np.frombuffer(blob, dtype="<i4", count=1, offset=off)[0]); off += 4
        n = int(np.frombuffer(blob, dtype="<i8", count=1, offset=off)[0]); off += 8
        a = np.frombuffer(blob, dtype="<u8", count=2 * n, offset=off).reshape(n, 2).copy(); off += 16 * n
        f = p = None
        if have_fp:
            f = np.frombuffer(blob, dtype="<i4", count=n, offset=off).copy(); off += 4 * n
            p = np.frombuffer(blob, dtype="<i8", count=n, offset=off).copy(); off += 8 * n
        n_u = int(np.frombuffer(blob, dtype="<i4", count=1, offset=off)[0]); off += 4
        u = np.frombuffer(blob, dtype="<u8", count=n_u, offset=off).copy(); off += 8 * n_u
        n_out = int(np.frombuffer(blob, dtype="<i8", count=1, offset=off)[0]); off += 8
        a_out = np.frombuffer(blob, dtype="<u8", count=2 * n_out, offset=off).reshape(n_out, 2).copy(); off += 16 * n_out
        prm = Param(max_dist_x=int(ints[0]), max_dist_y=int(ints[1]), bw=int(ints[2]), max_skip=int(ints[3]),
                    max_iter=int(ints[4]), min_cnt=int(ints[5]), min_sc=int(ints[6]),
                    pen_gap=float(fl[0]), pen_skip=float(fl[1]), is_cdna=int(ints[7]), n_seg=int(ints[8]))
        recs.append(dict(prm=prm, a=a, f=f, p=p, u=u, a_out=a_out))
    return recs


# minimap.h:8-9,28-29,40: the option bits collect_seed_hits / skip_seed look at
MM_F_NO_DIAG, MM_F_NO_DUAL, MM_F_FOR_ONLY, MM_F_REV_ONLY, MM_F_QSTRAND = 0x001, 0x002, 0x100000, 0x200000, 0x100000000


def collect_seeds(flag, qlen, seeds, hit_off, hits, q_rank=0, ref_len=None, ref_rank=None):
    """The oracle's collect_seed_hits (map.c:295-331) for one read: seeds (n,4) uint32 = leading 16 bytes of mm_seed_t,
    hits = the concatenated cr arrays, hit_off (n+1,) int64.  Returns the sorted anchors (m,2) uint64."""
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32).reshape(-1, 4)
    hit_off = np.ascontiguousarray(hit_off, dtype=np.int64)
    hits = np.ascontiguousarray(hits, dtype=np.uint64)
    out = np.empty((max(len(hits), 1), 2), dtype=np.uint64)
    rl = np.ascontiguousarray(ref_len, dtype=np.int32) if ref_len is not None else None
    rr = np.ascontiguousarray(ref_rank, dtype=np.int32) if ref_rank is not None else None
    n = lib().orc_collect_seeds(int(flag), int(qlen), int(q_rank), len(seeds), seeds.ctypes.data, hit_off.ctypes.data, hits.ctypes.data,
                                rl.ctypes.data if rl is not None else None, rr.ctypes.data if rr is not None else None, out.ctypes.data)
    return out[:n].copy()


def read_seed_capture(path):
    """Parse $MM2GB_CAPTURE_SEEDS (oracle/capture_hooks.c): one dict per mm_collect_matches call that reached chaining."""
    recs = []
    with open(path, "rb") as fh:
        blob = fh.read()
    off = 0
    while off < len(blob):
        magic = blob[off:off + 7]; off += 8
        if magic == b"MMSEED1":
            qlen, n_m, rep_len, n_mp = (int(v) for v in np.frombuffer(blob, dtype="<i4", count=4, offset=off)); off += 16
            seeds = np.frombuffer(blob, dtype="<u4", count=4 * n_m, offset=off).reshape(n_m, 4).copy(); off += 16 * n_m
            nh = int(seeds[:, 0].sum())
            hits = np.frombuffer(blob, dtype="<u8", count=nh, offset=off).copy(); off += 8 * nh
            mini_pos = np.frombuffer(blob, dtype="<u8", count=n_mp, offset=off).copy(); off += 8 * n_mp
            hit_off = np.zeros(n_m + 1, dtype=np.int64); np.cumsum(seeds[:, 0], out=hit_off[1:])
            recs.append(dict(qlen=qlen, seeds=seeds, hits=hits, hit_off=hit_off, rep_len=rep_len, mini_pos=mini_pos, a=None))
        else:
            assert magic == b"MMANCH1", "bad seed capture magic"
            n = int(np.frombuffer(blob, dtype="<i8", count=1, offset=off)[0]); off += 8
            recs[-1]["a"] = np.frombuffer(blob, dtype="<u8", count=2 * n, offset=off).reshape(n, 2).copy(); off += 16 * n
    return [r for r in recs if r["a"] is not None]
