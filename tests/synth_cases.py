"""Seeded synthetic anchor sets for parity tests (numpy; small/medium sizes).

Anchor packing follows lchain.c:140-143 / mmpriv.h:18-24:
    x = rev<<63 | rid<<32 | ref_pos        y = seg_id<<48 | flags<<40 | q_span<<32 | query_pos
Every generator returns an (n, 2) uint64 array sorted by x, which is what map.c:329 hands to chaining.
"""
import numpy as np


def pack(rid, rev, rpos, qpos, qspan=15, sid=0, flags=0):
    rid = np.asarray(rid, dtype=np.uint64)
    rev = np.asarray(rev, dtype=np.uint64)
    rpos = np.asarray(rpos, dtype=np.uint64)
    qpos = np.asarray(qpos, dtype=np.uint64)
    qspan = np.broadcast_to(np.asarray(qspan, dtype=np.uint64), rpos.shape)
    sid = np.broadcast_to(np.asarray(sid, dtype=np.uint64), rpos.shape)
    flags = np.broadcast_to(np.asarray(flags, dtype=np.uint64), rpos.shape)
    x = (rev << np.uint64(63)) | (rid << np.uint64(32)) | rpos
    y = (sid << np.uint64(48)) | (flags << np.uint64(40)) | (qspan << np.uint64(32)) | qpos
    return np.stack([x, y], axis=1)


def sort_by_x(a):
    return a[np.argsort(a[:, 0], kind="stable")]


def noise(n, seed, n_rid=24, span=100_000_000, qlen=50_000):
    rng = np.random.default_rng(seed)
    return sort_by_x(pack(rng.integers(0, n_rid, n), rng.integers(0, 2, n), rng.integers(0, span, n),
                          rng.integers(15, qlen, n)))


def colinear(n, seed, rid=3, rev=0, r0=1_000_000, q0=100, max_gap=33, indel_p=0.125, qspan=15):
    """A true chain: gaps U[1,max_gap], occasional small indels on either axis."""
    rng = np.random.default_rng(seed)
    step = rng.integers(1, max_gap + 1, n)
    dx = step + np.where(rng.random(n) < indel_p, rng.integers(0, 20, n), 0)
    dy = step + np.where(rng.random(n) < indel_p, rng.integers(0, 20, n), 0)
    return pack(np.full(n, rid), np.full(n, rev), r0 + np.cumsum(dx), q0 + np.cumsum(dy), qspan=qspan)


def repeat_block(n, seed, rid=3, rev=0, r0=2_000_000, q0=5_000, xwin=4000, ywin=6000):
    """Dense block: many anchors inside one max_dist_x window (saturates max_iter when n > max_iter)."""
    rng = np.random.default_rng(seed)
    return pack(np.full(n, rid), np.full(n, rev), r0 + rng.integers(0, xwin, n), q0 + rng.integers(0, ywin, n))


def read_like(length, seed, with_repeat=True):
    """One ONT-like read: chain + 3x noise + optional repeat block (small-scale SURVEY 8d recipe)."""
    rng = np.random.default_rng(seed)
    n_true = max(4, int(0.06 * length))
    parts = [colinear(n_true, seed + 1, q0=50), noise(3 * n_true, seed + 2, qlen=max(length, 32))]
    if with_repeat:
        parts.append(repeat_block(int(rng.integers(200, 1500)), seed + 3, r0=1_000_000 + int(rng.integers(0, 17 * n_true)),
                                  q0=int(rng.integers(15, max(16, length - 6000)))))
    return sort_by_x(np.concatenate(parts))


def rescue_case(n_noise=9000, seed=7, n_chain=60):
    """a chain, then > max_iter noise anchors packed inside max_dist_x, then the chain goes on.
    With max_iter < n_noise the window of the first post-noise chain anchor no longer reaches the chain's
    last anchor, and only the max_ii rescue (lchain.c:190-201) links across."""
    rng = np.random.default_rng(seed)
    first = colinear(n_chain, seed, r0=1_000_000, q0=1000, max_gap=20, indel_p=0.0)
    xe = int(first[-1, 0] & np.uint64(0xffffffff))
    ye = int(first[-1, 1] & np.uint64(0xffffffff))
    # noise sits just after the chain end in x but far away in y, so it scores low and never chains with it
    nx = xe + 1 + np.sort(rng.integers(0, 2500, n_noise))
    ny = rng.integers(20_000, 60_000, n_noise)
    mid = pack(np.full(n_noise, 3), np.zeros(n_noise, np.int64), nx, ny)
    second = colinear(n_chain, seed + 1, r0=xe + 2600, q0=ye + 2600, max_gap=20, indel_p=0.0)
    return sort_by_x(np.concatenate([first, mid, second]))


def grid_ties(nx=40, ny=12, step=20):
    """Anchors on a regular lattice: lots of exactly equal candidate scores (tie-break coverage)."""
    gx, gy = np.meshgrid(np.arange(nx) * step + 5000, np.arange(ny) * step + 100, indexing="ij")
    return sort_by_x(pack(np.full(gx.size, 1), np.zeros(gx.size, np.int64), gx.ravel(), gy.ravel()))


def two_segments(n, seed):
    """Paired-end-like: seg_id 0/1 mixed on the same reference region (n_seg = 2 branches of comput_sc)."""
    rng = np.random.default_rng(seed)
    a = colinear(n, seed, r0=50_000, q0=10, max_gap=9)
    sid = rng.integers(0, 2, n).astype(np.uint64)
    a[:, 1] |= sid << np.uint64(48)
    # duplicate a few reference positions across segments to hit the dr == 0 bonus branch (lchain.c:132)
    dup = a[:: max(1, n // 10)].copy()
    dup[:, 1] ^= np.uint64(1) << np.uint64(48)
    dup[:, 1] += np.uint64(3)
    return sort_by_x(np.concatenate([a, dup]))


def variable_span(n, seed):
    """HPC-like seeds: q_span varies per anchor."""
    rng = np.random.default_rng(seed)
    a = colinear(n, seed, max_gap=25)
    span = rng.integers(9, 40, n).astype(np.uint64)
    a[:, 1] = (a[:, 1] & ~(np.uint64(0xff) << np.uint64(32))) | (span << np.uint64(32))
    return sort_by_x(np.concatenate([a, noise(n // 2, seed + 5, n_rid=4, span=200_000, qlen=30_000)]))


def multi_read_batch(n_reads, seed, min_len=2_000, max_len=30_000):
    """Concatenated reads + offsets (the micro-batch layout handed to the GPU path)."""
    rng = np.random.default_rng(seed)
    reads = [read_like(int(rng.integers(min_len, max_len)), seed * 1000 + r, with_repeat=(r % 3 == 0)) for r in range(n_reads)]
    off = np.zeros(n_reads + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(r) for r in reads])
    return np.concatenate(reads), off


def fuzz_case(rng):
    """One random batch and one random set of chaining parameters for the seeded fuzz test and tests/fuzz_soak.py:
    mixtures of chains, noise, repeat blocks, duplicated positions, empty reads, reads repeated or overlapping an earlier
    read of the batch; window limits around the tile (64) and ring sizes.  Returns (anchors, read offsets, parameter dict)."""
    reads = []
    for _ in range(int(rng.integers(1, 7))):
        kind = int(rng.integers(0, 7))
        seed = int(rng.integers(1, 1 << 30))
        if kind == 0:
            reads.append(np.zeros((0, 2), np.uint64))
        elif kind == 1:
            reads.append(noise(int(rng.integers(1, 400)), seed, n_rid=int(rng.integers(1, 4)), span=int(rng.integers(2_000, 200_000)), qlen=20_000))
        elif kind == 2:
            reads.append(read_like(int(rng.integers(1_000, 25_000)), seed))
        elif kind == 3:
            reads.append(sort_by_x(np.concatenate([repeat_block(int(rng.integers(100, 3000)), seed, xwin=int(rng.integers(50, 5000)), ywin=int(rng.integers(50, 7000))),
                                                   colinear(int(rng.integers(10, 800)), seed + 1, max_gap=int(rng.integers(2, 60)))])))
        elif kind == 4:
            reads.append(variable_span(int(rng.integers(50, 900)), seed))
        elif kind == 5:
            reads.append(grid_ties(nx=int(rng.integers(3, 50)), ny=int(rng.integers(2, 14)), step=int(rng.integers(1, 40))))
        elif reads and len(reads[-1]):
            prev = reads[-1]                       # a read over the same region: a slice of the previous one
            lo = int(rng.integers(0, len(prev)))
            reads.append(prev[lo:lo + int(rng.integers(1, len(prev) - lo + 1))].copy())
        else:
            reads.append(colinear(int(rng.integers(1, 300)), seed))
    off = np.zeros(len(reads) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(r) for r in reads])
    a = np.concatenate(reads) if off[-1] else np.zeros((0, 2), np.uint64)
    kw = dict(max_iter=int(rng.choice([1, 7, 63, 64, 65, 200, 1000, 5000])), bw=int(rng.choice([0, 1, 50, 500, 3000])),
              max_dist_x=int(rng.choice([10, 500, 5000, 20000])), max_dist_y=int(rng.choice([10, 500, 5000, 20000])),
              pen_gap=np.float32(rng.choice([0.0, 0.12, 0.19, 1.5])), pen_skip=np.float32(rng.choice([0.0, 0.0, 0.01, 0.3])),
              min_cnt=int(rng.integers(1, 5)), min_sc=int(rng.choice([1, 20, 40, 100])))
    return a, off, kw
