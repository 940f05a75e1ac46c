// post_kernels.hip -- backtrack + compaction on the device: chains out of (f, p) without a trip to the host.
//
// What is computed is fixed by the reference's CPU code: mg_chain_backtrack / mg_chain_bk_end (lchain.c:9-76), compact_a
// (lchain.c:78-111) and -- because the order in which equal scores are visited decides which chain an anchor ends up in --
// the exact element order radix_sort_128x leaves (ksort.h:98-151: in-place most-significant-byte radix passes with a cycle
// permutation, insertion sort for runs of <= 64).  None of these steps has a parallel form with the same results (the
// permutation is a pointer chase through 256 bucket heads, a chain walk follows predecessor links and stops at anchors an
// earlier, better chain has taken), so the parallelism is ACROSS reads: one wave per read, thousands of reads in flight,
// the wave's 64 lanes used wherever a step is data-parallel (candidate collection, histograms, prefix sums, small-run
// sorts, segment discovery, the final gather) and lane 0 for the two chases.  A low-occupancy, latency-bound kernel by
// design; it runs on the engine's compute stream behind the score kernel of its micro-batch.
//
//   k_post_chains   per read: candidates z = (f, i) with f >= min_sc  ->  the host's sort order  ->  chain walks  ->
//                   picked[] (anchor indices, chain by chain), u_tmp[] (score<<32 | count), n_u, n_kept
//   k_post_scan     exclusive scans of n_u / n_kept over the reads -> u_off, a_off (+ totals)
//   k_post_emit     per read: chains ordered by the reference position of their first anchor (same sort), u[] and the
//                   compacted anchors written to their final place
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <limits.h>
#include <algorithm>
#include <rocprim/device/device_segmented_radix_sort.hpp>   // the (strip, index) order of the RMQ fill's inner windows: a plain segmented key sort
#include "chain_dev.h"
#include "post_dev.h"

namespace mm2gb {

namespace {

constexpr int W = 64;
constexpr int POST_THREADS = 256;               // 4 waves, one read each
constexpr int SMALL_RUN = 64;                   // RS_MIN_SIZE, ksort.h:98
#ifndef MM2GB_POST_SPEC
#define MM2GB_POST_SPEC 4
#endif
constexpr int SPEC = MM2GB_POST_SPEC;
static_assert(SPEC == 4, "the walk resolution spells out four look-ahead steps");                         // steps of a chain walk every candidate of a group takes ahead of its turn

__device__ __forceinline__ int lane() { return threadIdx.x & (W - 1); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// the same for LDS alone: a wave's LDS instructions execute in order, so all it takes is that the compiler keeps them in order (and that what was
// read has arrived); outstanding global stores are NOT waited for
__device__ __forceinline__ void lds_sync()
{
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void wave_sync()
{
	// LDS and global accesses of one wave are issued in order; the fence keeps the compiler from moving them across phases
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	__builtin_amdgcn_wave_barrier();
}

// Per-wave LDS scratch of a radix pass: bucket bounds, and a cache of the elements at every bucket's head -- the cycle
// permutation reads one element per step from 256 slowly advancing heads.  Every bucket owns a line of the cache whose length
// grows with the bucket's share of the keys (a bucket that takes half of the steps gets half of the spare space), holding the
// elements that follow `anchor` (a position the head is at or has passed).  Most steps are then an LDS read instead of a memory
// round trip, and when a head runs off its line all lines are fetched again together: one round trip for the lot, every few
// hundred steps whatever the distribution of the keys.
#ifndef MM2GB_POST_LINE_BYTES
#define MM2GB_POST_LINE_BYTES 7168
#endif
constexpr int LINE_STORE_BYTES = MM2GB_POST_LINE_BYTES;
static_assert(LINE_STORE_BYTES % 1024 == 0 && LINE_STORE_BYTES / 8 >= 2 * 256 + 64, "line slots come in whole rounds of 64 for both element kinds (8 and 16 bytes), and every bucket owns at least two (8-byte elements): fewer and the lines cannot be laid out");
// k_post_chains: registers are budgeted for this many waves per SIMD (= workgroups per CU; its LDS allows 3 at 6 KB of lines).  The kernel is a
// sum of latencies: 168 registers and three waves per SIMD (6 spilled) against 184 and two: 58.5 -> 55.7 ms per 500 M anchors
#ifndef MM2GB_POST_WAVES_PER_SIMD
#define MM2GB_POST_WAVES_PER_SIMD 3
#endif
constexpr int POST_WAVES_PER_SIMD = MM2GB_POST_WAVES_PER_SIMD;
#ifndef MM2GB_POST_MARKS_IN_MEMORY
#define MM2GB_POST_MARKS_IN_MEMORY 0            // 1: the walks' "taken" flags in the anchors' records for every read (round 4; A/B builds)
#endif
constexpr bool POST_MARKS_IN_MEMORY = MM2GB_POST_MARKS_IN_MEMORY != 0;
struct alignas(16) PassLds {
	int where[256];              // histogram first; then line start | line length << 16
	int head[256], tail[256], anchor[256];
	unsigned long long line[LINE_STORE_BYTES / 8];
	unsigned char owner[LINE_STORE_BYTES / 8];   // bucket of every line slot
};

// ---- the two element kinds that get sorted the host's way ----------------------------------------------------------
// Z: candidates of the backtrack, key = score f (lchain.c:38-41: z[k].x = f[i], z[k].y = i), packed f<<32 | i.
// H: chain heads of the compaction, key = x of the chain's first anchor, value = offset<<32 | chain (lchain.c:94-99).
struct ZElem {
	using T = unsigned long long;
	[[maybe_unused]] static constexpr int SLOTS = LINE_STORE_BYTES / 8, LINE_MIN = 2;   // line slots of a wave; shortest line
	static __device__ __forceinline__ unsigned long long key(T e) { return e >> 32; }
};
struct HElem {
	using T = ulonglong2;
	static constexpr int SLOTS = LINE_STORE_BYTES / 16, LINE_MIN = 1;
	static __device__ __forceinline__ unsigned long long key(const T &e) { return e.x; }
};

// index of the lowest / highest set bit of a non-zero 64-bit mask, from 32-bit halves (see leading_out below for why)
__device__ __forceinline__ int first_set(unsigned long long m)
{
	const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
	return lo ? __builtin_ctz(lo) : 32 + __builtin_ctz(hi);
}
__device__ __forceinline__ int first_set_from_top(unsigned long long m)       // number of leading zeros
{
	const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
	return hi ? __builtin_clz(hi) : 32 + __builtin_clz(lo);
}

__device__ __forceinline__ unsigned long long shfl_up64(unsigned long long v, int by)
{
	return (unsigned long long)(unsigned)__shfl_up((int)(unsigned)v, by) | (unsigned long long)(unsigned)__shfl_up((int)(unsigned)(v >> 32), by) << 32;
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int src)   // src wave-uniform
{
	return (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src) | (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src) << 32;
}
[[maybe_unused]] __device__ __forceinline__ unsigned long long bcast_elem(unsigned long long e, int src) { return __shfl(e, src); }
__device__ __forceinline__ ulonglong2 bcast_elem(const ulonglong2 &e, int src) { return make_ulonglong2(__shfl(e.x, src), __shfl(e.y, src)); }

// Stable sort of a run of at most 64 elements by key == what rs_insertsort leaves (ksort.h:105-115): every lane holds one
// element and counts the elements that must come before it.
template <class E>
__device__ __forceinline__ void small_run_sort(typename E::T *g, int lo, int len)
{
	const int l = lane();
	const bool in = l < len;
	typename E::T e = g[lo + (in ? l : 0)];
	const unsigned long long k = E::key(e);
	const unsigned klo = (unsigned)k, khi = (unsigned)(k >> 32);
	// already in order (the common case at the lower levels: a run sorted one level up is seen again)
	const unsigned plo = __shfl_up(klo, 1), phi = __shfl_up(khi, 1);
	const unsigned long long prev = (unsigned long long)phi << 32 | plo;
	if (__ballot(in && l > 0 && k < prev) == 0) return;
	int rank = 0;
	for (int m = 0; m < len; ++m) {
		const unsigned long long km = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)khi, m) << 32 | (unsigned)__builtin_amdgcn_readlane((int)klo, m);
		rank += (km < k) || (km == k && m < l);
	}
	wave_sync();
	if (in) g[lo + rank] = e;
	wave_sync();
}

// Every bucket's line filled from its head on: all lanes, one memory round trip (every load is issued before the first one is
// waited for).  Positions at or beyond a head still hold what they held when the pass began -- only a consumed head position is
// ever written -- so a line stays valid until the head has moved past it.  Positions past the end of a bucket are read too
// (clamped to the array) and never used: a head stops at its bucket's tail.
template <class E>
__device__ __forceinline__ void fetch_all_lines(const typename E::T *g, int last, int used, PassLds &L)
{
	typename E::T *line = (typename E::T*)L.line;
	const int l = lane();
	constexpr int ITERS = E::SLOTS / W;
	typename E::T v[ITERS];
	int hd[ITERS], d[ITERS], j[ITERS];
#pragma unroll
	for (int it = 0; it < ITERS; ++it) {
		const int slot = min(it * W + l, used - 1);
		d[it] = L.owner[slot];
		j[it] = slot - (L.where[d[it]] & 0xffff);
		hd[it] = L.head[d[it]];
		v[it] = g[min(hd[it] + j[it], last)];
	}
#pragma unroll
	for (int it = 0; it < ITERS; ++it) {
		const int slot = it * W + l;
		if (slot < used) {
			line[slot] = v[it];
			if (j[it] == 0) L.anchor[d[it]] = hd[it];
		}
	}
	wave_sync();
}

// One pass of rs_sort (ksort.h:116-146) over g[lo, hi) on key byte `shift`: histogram and bucket bounds with all lanes, then
// the in-place cycle permutation exactly as the host does it -- element by element, each placement evicting the element
// that decides the next one.  Returns false when every key has the same byte (the pass moves nothing).
template <class E>
__device__ __forceinline__ bool radix_pass(typename E::T *g, int lo, int hi, int shift, PassLds &L, long long *dbg = nullptr)
{
	const int l = lane();
	for (int k = l; k < 256; k += W) L.where[k] = 0;
	wave_sync();
	for (int base = lo; base < hi; base += 4 * W) {
		unsigned long long key[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) key[u] = E::key(g[min(base + u * W + l, hi - 1)]);
#pragma unroll
		for (int u = 0; u < 4; ++u) if (base + u * W + l < hi) atomicAdd(&L.where[(int)(key[u] >> shift) & 255], 1);
	}
	wave_sync();
	// lane l owns buckets 4l .. 4l+3
	const int c0 = L.where[4 * l], c1 = L.where[4 * l + 1], c2 = L.where[4 * l + 2], c3 = L.where[4 * l + 3];
	const int len = hi - lo;
	if (__ballot(c0 == len || c1 == len || c2 == len || c3 == len) != 0) return false;
	int inc = c0 + c1 + c2 + c3;
	const int own = inc;
	for (int off = 1; off < W; off <<= 1) { const int o = __shfl_up(inc, off); if (l >= off) inc += o; }
	int at = lo + inc - own;
	L.head[4 * l] = at; at += c0; L.tail[4 * l] = at;
	L.head[4 * l + 1] = at; at += c1; L.tail[4 * l + 1] = at;
	L.head[4 * l + 2] = at; at += c2; L.tail[4 * l + 2] = at;
	L.head[4 * l + 3] = at; at += c3; L.tail[4 * l + 3] = at;
	// line lengths: LINE_MIN each, the rest of the slots by share of the keys
	constexpr int SPARE = E::SLOTS - 256 * E::LINE_MIN;
	const int cap[4] = { E::LINE_MIN + (int)((long long)c0 * SPARE / len), E::LINE_MIN + (int)((long long)c1 * SPARE / len),
	                     E::LINE_MIN + (int)((long long)c2 * SPARE / len), E::LINE_MIN + (int)((long long)c3 * SPARE / len) };
	int cinc = cap[0] + cap[1] + cap[2] + cap[3];
	const int cown = cinc;
	for (int off = 1; off < W; off <<= 1) { const int o = __shfl_up(cinc, off); if (l >= off) cinc += o; }
	const int used = __builtin_amdgcn_readlane(cinc, W - 1);
	int cat = cinc - cown;
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		L.where[4 * l + q] = cat | cap[q] << 16;
		for (int j = 0; j < cap[q]; ++j) L.owner[cat + j] = (unsigned char)(4 * l + q);
		cat += cap[q];
	}
	wave_sync();
	fetch_all_lines<E>(g, hi - 1, used, L);
	int since = 0;                                          // cycle steps since all lines were fetched together
	int d_steps = 0, d_one = 0, d_all = 1, d_cycles = 0;
	// The host's loop, bucket by bucket: elements at the head of bucket k that already belong to k are passed over (all 64
	// lanes look at the next 64 of them at once); the first one that does not starts a cycle, which is followed exactly as
	// the host does -- place the carried element at the head of its bucket, pick up what was there -- until an element of
	// bucket k turns up (ksort.h:128-139).
	for (int k = 0; k < 256; ++k) {
		int hk = uni(L.head[k]);
		const int tk = uni(L.tail[k]);
		for (; hk < tk; hk += W) {
			// the next 64 elements of bucket k: those that belong elsewhere start a cycle each, in order.  Nothing but the end of such a
			// cycle writes into this part of the array (at the position the cycle started from), so one load serves them all.
			const int i = hk + l;
			const bool in = i < tk;
			typename E::T e = g[in ? i : hk];
			unsigned long long moves = __ballot(in && ((int)(E::key(e) >> shift) & 255) != k);
			while (((unsigned)moves | (unsigned)(moves >> 32)) != 0) {
			const int skip = first_set(moves);
			moves &= moves - 1;
			++d_cycles;
			typename E::T carry = bcast_elem(e, skip);
			// the cycle, wave-uniform: the element at the head of the destination bucket comes from that bucket's line
			typename E::T *line = (typename E::T*)L.line;
			int d = uni((int)(E::key(carry) >> shift) & 255);
			while (d != k) {
				const int hd = uni(L.head[d]), wh = uni(L.where[d]);
				const int l0 = wh & 0xffff, ln = wh >> 16;
				int at_line = hd - uni(L.anchor[d]);
				if (at_line >= ln) {
					// off the line: all lines again, unless that was done a moment ago (then this line alone)
					if (since >= 64) { fetch_all_lines<E>(g, hi - 1, used, L); since = 0; ++d_all; }
					else {
						++d_one;
						for (int j = l; j < ln; j += W) line[l0 + j] = g[min(hd + j, hi - 1)];
						if (l == 0) L.anchor[d] = hd;
						wave_sync();
					}
					at_line = 0;
				}
				const typename E::T next = line[l0 + at_line];
				if (l == 0) { g[hd] = carry; L.head[d] = hd + 1; }
				carry = next;
				++since; ++d_steps;
				d = uni((int)(E::key(carry) >> shift) & 255);
			}
			if (l == 0) g[hk + skip] = carry;
			}
			wave_sync();
		}
	}
	wave_sync();
	if (dbg && l == 0) {
		atomicAdd((unsigned long long*)&dbg[0], (unsigned long long)d_steps);
		atomicAdd((unsigned long long*)&dbg[1], (unsigned long long)d_one);
		atomicAdd((unsigned long long*)&dbg[2], (unsigned long long)d_all);
		atomicAdd((unsigned long long*)&dbg[3], (unsigned long long)d_cycles);
		atomicAdd((unsigned long long*)&dbg[4], (unsigned long long)(hi - lo));
		atomicAdd((unsigned long long*)&dbg[5], 1ull);
	}
	return true;
}

// ---- the same pass as a walk over BYTES (round 5; candidates' sort only) ---------------------------------------------------------------
// What the cycle permutation does next depends on one thing only: the destination bucket of the element it has just picked up, i.e. one key
// byte.  And a position at or beyond a bucket's head keeps its original occupant until the head gets there.  So the pass is run on the
// sequence of destination bytes S[i] = byte of g[i] (written once, in parallel), and what the walk produces is the permutation: perm[i] = where
// the element that was at i ends up (identity for the elements that are accepted in place); the elements are moved afterwards by all lanes.
// The walk's state per bucket is ONE 16-byte record in LDS -- next byte of the bucket's line, next position, line end, bucket end -- and a
// step is: read the record of the bucket the carried element goes to, read the byte of the occupant it displaces, write the record back, store
// one perm entry; nothing else, on one lane, no scalar round trips.  A line holds BYTES: 8 positions per 8 bytes of LDS where the element form
// holds one, so lines run out an eighth as often (level 1 of the 500 M-anchor batch: 10.8 M single-line refills in 87 M steps before).
struct SortScratch {
	unsigned char *S;            // one byte per position of the array being sorted
	int *perm;                   // one entry per position
	unsigned long long *tmp;     // the elements' way station
};

constexpr int GROUP = 8;                                   // bytes per line slot
constexpr int GROUPS = LINE_STORE_BYTES / GROUP;           // 896 line slots per wave (PassLds::owner has one entry each)
static_assert(GROUPS >= 2 * 256, "every bucket owns at least one slot, and there is spare to hand out");

__device__ __forceinline__ unsigned long long load8_unaligned(const unsigned char *p)
{
	unsigned long long v;
	__builtin_memcpy(&v, p, 8);
	return v;
}

// every bucket's line filled with the bytes that follow its current position (all lanes, one memory round trip)
__device__ __forceinline__ void fetch_all_byte_lines(const unsigned char *S, int used, PassLds &L)
{
	int4 *rec = (int4*)L.where;
	unsigned long long *line = L.line;
	const int l = lane();
	constexpr int ITERS = GROUPS / W;
	unsigned long long v[ITERS];
	int d[ITERS], j[ITERS], l0[ITERS];
#pragma unroll
	for (int it = 0; it < ITERS; ++it) {
		const int slot = min(it * W + l, used - 1);
		d[it] = L.owner[slot];
		const int4 r = rec[d[it]];
		l0[it] = (int)((unsigned)r.z >> 16);
		j[it] = slot - l0[it] / GROUP;
		v[it] = load8_unaligned(S + r.y + GROUP * j[it]);      // (S has slack at its end: bytes past a bucket's end are read and never used)
	}
	wave_sync();
#pragma unroll
	for (int it = 0; it < ITERS; ++it) {
		const int slot = it * W + l;
		if (slot < used) {
			line[slot] = v[it];
			if (j[it] == 0) rec[d[it]].x = l0[it];
		}
	}
	wave_sync();
}

// A run of at most LINE_STORE_BYTES elements (nearly every run below the top level: 1 800 elements on average at the lowest byte) keeps its
// whole byte sequence in LDS: no lines, no refills, and the scan of a bucket for the elements that must move reads LDS instead of memory --
// with 256 buckets and a few elements in each, those 256 dependent memory round trips were most of a small pass's time.
// (few-bucket passes, below, report their buckets: the run's children are known without another look at the elements)
struct FewBuckets { int n = 0; int start[4] = { 0, 0, 0, 0 }, end[4] = { 0, 0, 0, 0 }; };
// FEW_ONLY: only the few-bucket form (below) and nothing of PassLds but 256 words for the histogram -- the pass a read's candidates get right where
// they were collected (k_post_classes).  Returns 0: every key has the same byte (nothing moved), 1: done, 2 (FEW_ONLY): more than four values, not done.
template <bool FEW_ONLY>
__device__ __forceinline__ int radix_pass_bytes_t(unsigned long long *g, int lo, int hi, int shift, PassLds *Lp, int *where, const SortScratch &sc, long long *dbg, long long *ph, FewBuckets *fb)
{
	long long tp = ph ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
	auto phase = [&](int k) { if (ph) { const long long tn = (long long)__builtin_amdgcn_s_memrealtime(); ph[k] += tn - tp; tp = tn; } };
	const int l = lane();
	const int len = hi - lo;
	const bool resident = !FEW_ONLY && len <= LINE_STORE_BYTES;
	unsigned char *lineb = FEW_ONLY ? nullptr : (unsigned char*)Lp->line;
	for (int k = l; k < 256; k += W) where[k] = 0;
	wave_sync();
	// histogram, the byte sequence, the identity permutation: one pass over the run, eight loads in flight per lane
	for (int base = lo; base < hi; base += 8 * W) {
		int byte[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) byte[u] = (int)(ZElem::key(g[min(base + u * W + l, hi - 1)]) >> shift) & 255;
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int i = base + u * W + l;
			if (i < hi) {
				atomicAdd(&where[byte[u]], 1);
				if (resident) lineb[i - lo] = (unsigned char)byte[u]; else sc.S[i] = (unsigned char)byte[u];
				sc.perm[i] = i;
			}
		}
	}
	wave_sync();
	phase(0);
	// lane l owns buckets 4l .. 4l+3
	const int c[4] = { where[4 * l], where[4 * l + 1], where[4 * l + 2], where[4 * l + 3] };
	if (__ballot(c[0] == len || c[1] == len || c[2] == len || c[3] == len) != 0) return 0;
	int inc = c[0] + c[1] + c[2] + c[3];
	const int own = inc;
	for (int off = 1; off < W; off <<= 1) { const int o = __shfl_up(inc, off); if (l >= off) inc += o; }
	int at = lo + inc - own;
	// buckets that hold anything, as four masks (bucket 4 l + q: bit l of mask q): the walk only visits those
	unsigned long long full[4];
#pragma unroll
	for (int q = 0; q < 4; ++q) full[q] = __ballot(c[q] > 0);
	int4 *rec = (int4*)where;
	int used = 0;
	int since = 0;
	int d_steps = 0, d_one = 0, d_all = 1, d_cycles = 0;
	// A run too long for LDS whose keys take at most four values of this byte -- the top pass of a read: its scores span two to four values of
	// their highest byte that differs -- needs no records and no lines: the walk's whole state is four positions, and what it reads are four
	// sequences of bytes that it consumes front to back.  Positions live in scalar registers, the next 64 bytes of every sequence across the
	// lanes of one vector register (a step reads one with v_readlane; 64 more are loaded when they run out): a step is a handful of scalar
	// instructions instead of two trips to LDS behind each other (0.33 us a step before, and such passes were what a level's launch ended with).
	const int n_full = __popcll(full[0]) + __popcll(full[1]) + __popcll(full[2]) + __popcll(full[3]);
	const bool few = !resident && n_full <= 4;
	if (FEW_ONLY && !few) return 2;
	if (few) {
		int v0 = -1, v1 = -1, v2 = -1, v3 = -1, h0 = 0, h1 = 0, h2 = 0, h3 = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0, nf = 0;
		{
			const int st_q[4] = { at, at + c[0], at + c[0] + c[1], at + c[0] + c[1] + c[2] };
			unsigned long long lanes = full[0] | full[1] | full[2] | full[3];
			while (((unsigned)lanes | (unsigned)(lanes >> 32)) != 0) {
				const int src_l = first_set(lanes);
				lanes &= lanes - 1;
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const int cq = __builtin_amdgcn_readlane(c[q], src_l), sq = __builtin_amdgcn_readlane(st_q[q], src_l);
					if (cq > 0) {
						const int val = 4 * src_l + q;
						if (nf == 0) { v0 = val; h0 = sq; e0 = sq + cq; } else if (nf == 1) { v1 = val; h1 = sq; e1 = sq + cq; }
						else if (nf == 2) { v2 = val; h2 = sq; e2 = sq + cq; } else { v3 = val; h3 = sq; e3 = sq + cq; }
						++nf;
					}
				}
			}
		}
		if (fb) { fb->n = nf; fb->start[0] = h0; fb->end[0] = e0; fb->start[1] = h1; fb->end[1] = e1; fb->start[2] = h2; fb->end[2] = e2; fb->start[3] = h3; fb->end[3] = e3; }
		phase(1);
		// the sequences of buckets 1 .. 3 (a cycle never goes to the first one: nothing that belongs below the bucket being done is left): 64 bytes
		// across the lanes.  (Tried: the next 64 asked for when the first are half used -- the extra test per step cost more than the waits it saved.)
		int w1 = 0, w2 = 0, w3 = 0, b1 = INT_MIN / 2, b2 = INT_MIN / 2, b3 = INT_MIN / 2;
		int d = 0, src = 0;
		// (where the elements go is collected 64 entries at a time across the lanes of two registers and stored by all lanes at once: a store per
		// step would be waited for by the next load of bytes -- loads and stores are counted together, in order)
		int out_src = 0, out_pos = 0, n_out = 0;
#define MM2GB_FEW_PUT(SRC, POS) { if (l == n_out) { out_src = SRC; out_pos = POS; } \
                                  if (++n_out == W) { sc.perm[out_src] = out_pos; n_out = 0; } }
// (Tried: a run of J's own elements that moves up by one found with a ballot and moved with one store -- the runs are short, a step each is faster.)
#define MM2GB_FEW_STEP(J) { unsigned off_ = (unsigned)(h##J - b##J); if (off_ >= (unsigned)W) { w##J = (int)sc.S[h##J + l]; b##J = h##J; off_ = 0; } \
                            const int nb_ = __builtin_amdgcn_readlane(w##J, (int)off_); MM2GB_FEW_PUT(src, h##J) src = h##J; ++h##J; d = nb_; }
#define MM2GB_FEW_BUCKET(K) if (K < nf) { \
			const int tk = e##K, vk = v##K; \
			int nx4[4]; \
			_Pragma("unroll") for (int u = 0; u < 4; ++u) nx4[u] = (int)sc.S[min(h##K + u * W + l, max(tk - 1, h##K))]; \
			for (int hq = h##K; hq < tk; hq += 4 * W) { \
				int by4[4]; \
				_Pragma("unroll") for (int u = 0; u < 4; ++u) by4[u] = nx4[u]; \
				if (hq + 4 * W < tk) { _Pragma("unroll") for (int u = 0; u < 4; ++u) nx4[u] = (int)sc.S[min(hq + (4 + u) * W + l, tk - 1)]; }   /* the next four blocks, under this one's cycles */ \
				_Pragma("unroll") for (int u = 0; u < 4; ++u) { \
					const int hk = hq + u * W; \
					if (hk < tk) { \
						unsigned long long moves = __ballot(hk + l < tk && by4[u] != vk); \
						while (((unsigned)moves | (unsigned)(moves >> 32)) != 0) { \
							const int skip = first_set(moves); \
							moves &= moves - 1; \
							++d_cycles; \
							const int home = hk + skip; \
							d = __builtin_amdgcn_readlane(by4[u], skip); src = home; \
							do { if (d == v1) MM2GB_FEW_STEP(1) else if (d == v2) MM2GB_FEW_STEP(2) else MM2GB_FEW_STEP(3) ++d_steps; } while (d != vk); \
							MM2GB_FEW_PUT(src, home) \
						} \
					} \
				} \
			} \
		}
		MM2GB_FEW_BUCKET(0)
		MM2GB_FEW_BUCKET(1)
		MM2GB_FEW_BUCKET(2)
		MM2GB_FEW_BUCKET(3)
		if (l < n_out) sc.perm[out_src] = out_pos;
#undef MM2GB_FEW_BUCKET
#undef MM2GB_FEW_STEP
#undef MM2GB_FEW_PUT
	} else if constexpr (!FEW_ONLY) {
	PassLds &L = *Lp;
	if (resident) {
		wave_sync();                                            // (the counts are in registers: the records take their place)
		// record of a bucket: x = its next position, w = its end
#pragma unroll
		for (int q = 0; q < 4; ++q) { rec[4 * l + q] = make_int4(at, 0, 0, at + c[q]); at += c[q]; }
		wave_sync();
		// the map of the elements that are not in their bucket (one bit per position, in the line slots' owner bytes): what the walking lane looks
		// for -- with it the lane finds a bucket's next element to move by itself, and the other 63 are not asked 256 times over
		for (int base = 0; base < len; base += W) {
			const int i = base + l;
			bool out = false;
			if (i < len) { const int4 r = rec[lineb[i]]; out = lo + i < r.x || lo + i >= r.w; }
			const unsigned long long m = __ballot(out);
			if (l == 0) { ((unsigned*)L.owner)[base / 32] = (unsigned)m; ((unsigned*)L.owner)[base / 32 + 1] = (unsigned)(m >> 32); }
		}
		wave_sync();
	} else {
		// line slots: one each, the spare ones by share of the keys
		constexpr int SPARE = GROUPS - 256;
		int cap[4];
#pragma unroll
		for (int q = 0; q < 4; ++q) cap[q] = 1 + (int)((long long)c[q] * SPARE / len);
		int cinc = cap[0] + cap[1] + cap[2] + cap[3];
		const int cown = cinc;
		for (int off = 1; off < W; off <<= 1) { const int o = __shfl_up(cinc, off); if (l >= off) cinc += o; }
		used = __builtin_amdgcn_readlane(cinc, W - 1);
		int cat = cinc - cown;
		wave_sync();
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const int l0 = cat * GROUP, lend = (cat + cap[q]) * GROUP;
			rec[4 * l + q] = make_int4(lend, at, lend | l0 << 16, at + c[q]);      // x == lend: "line empty" until the first fetch sets x = l0
			cat += cap[q]; at += c[q];
		}
		wave_sync();
		// owner of every line slot: all lanes, by bisection over the buckets' first slots (a bucket with most of the keys owns hundreds)
		for (int slot = l; slot < used; slot += W) {
			int a = 0, bq = 255;
			while (a < bq) { const int m = (a + bq + 1) >> 1; if ((int)((unsigned)rec[m].z >> 16) <= slot * GROUP) a = m; else bq = m - 1; }
			L.owner[slot] = (unsigned char)a;
		}
		wave_sync();
		fetch_all_byte_lines(sc.S, used, L);
	}
	phase(1);
	if (resident) {
		// the whole walk on lane 0 (ksort.h:128-139): bucket by bucket, the next element out of place from the map, its cycle -- the next position of
		// the bucket arrived at, the byte of its occupant, the position advanced, a perm entry per step.  (Tried: the occupant's byte kept in the
		// bucket's record, refreshed off the dependent chain -- one dependent LDS read per step instead of two, twice the instructions: slower.)
		int steps = 0, cycles = 0;
		if (l == 0) {
			const unsigned *bm = (const unsigned*)L.owner;
			for (int k = 0; k < 256; ++k) {
				const int4 rk = rec[k];
				int i = rk.x - lo;
				const int end = rk.w - lo;
				while (i < end) {
					const unsigned w = bm[i >> 5] >> (i & 31);
					if (w == 0) { i = (i | 31) + 1; continue; }
					i += __builtin_ctz(w);
					if (i >= end) break;
					const int home = lo + i;
					int d = lineb[i], src = home;
					++cycles;
					do {
						const int pos = rec[d].x;
						const int nb = lineb[pos - lo];
						sc.perm[src] = pos;
						rec[d].x = pos + 1;
						src = pos; d = nb; ++steps;
					} while (d != k);
					sc.perm[src] = home;
					++i;
				}
			}
		}
		d_steps = uni(steps); d_cycles = uni(cycles);
	} else
	for (int k = 0; k < 256; ++k) {
		if (!((full[k & 3] >> (k >> 2)) & 1)) continue;
		const int4 rk = rec[k];
		int hk = uni(rk.y);
		const int tk = uni(rk.w);
		int by4[4];
		for (int hq = hk; hq < tk; hq += 4 * W) {
			// the bucket's bytes, four blocks of 64 per round trip (they never change during the pass: read ahead at will)
#pragma unroll
			for (int u = 0; u < 4; ++u) { const int i = min(hq + u * W + l, tk - 1); by4[u] = (int)sc.S[i]; }
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			hk = hq + u * W;
			if (hk >= tk) break;
			const int i = hk + l;
			const bool in = i < tk;
			const int by = by4[u];
			unsigned long long moves = __ballot(in && by != k);
			while (((unsigned)moves | (unsigned)(moves >> 32)) != 0) {
				const int skip = first_set(moves);
				moves &= moves - 1;
				++d_cycles;
				const int home = hk + skip;
				int d = __builtin_amdgcn_readlane(by, skip), src = home;
				for (;;) {
					// the cycle on lane 0, until it closes or a line runs out
					int status = 0, steps = 0;                // 0 closed, 1 line of bucket d empty
					if (l == 0) {
						for (;;) {
							const int4 r = rec[d];
							if (r.x == (r.z & 0xffff)) { status = 1; break; }
							const int nb = lineb[r.x];
							sc.perm[src] = r.y;
							*(int2*)&rec[d] = make_int2(r.x + 1, r.y + 1);
							src = r.y; d = nb; ++steps;
							if (d == k) break;
						}
					}
					status = uni(status); d = uni(d); src = uni(src); steps = uni(steps);
					since += steps; d_steps += steps;
					if (status == 0) break;
					if (since >= 64) { fetch_all_byte_lines(sc.S, used, L); since = 0; ++d_all; }
					else {
						++d_one;
						const int4 r = rec[d];
						const int l0 = (int)((unsigned)r.z >> 16), ln = (r.z & 0xffff) - l0;
						for (int j = l * GROUP; j < ln; j += W * GROUP) *(unsigned long long*)(L.line + (l0 + j) / GROUP) = load8_unaligned(sc.S + r.y + j);
						if (l == 0) rec[d].x = l0;
						wave_sync();
					}
				}
				if (l == 0) sc.perm[src] = home;
			}
		}
		}
	}
	}
	wave_sync();
	phase(2);
	// the elements follow the permutation: out of place first, then back (all lanes, eight loads in flight each)
	for (int base = lo; base < hi; base += 8 * W) {
		unsigned long long e[8]; int to[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) { const int i = min(base + u * W + l, hi - 1); e[u] = g[i]; to[u] = sc.perm[i]; }
#pragma unroll
		for (int u = 0; u < 8; ++u) if (base + u * W + l < hi) sc.tmp[to[u]] = e[u];
	}
	wave_sync();
	for (int base = lo; base < hi; base += 8 * W) {
		unsigned long long e[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) e[u] = sc.tmp[min(base + u * W + l, hi - 1)];
#pragma unroll
		for (int u = 0; u < 8; ++u) if (base + u * W + l < hi) g[base + u * W + l] = e[u];
	}
	wave_sync();
	phase(3);
	if (ph) ph[5] += d_steps;
	if (dbg && l == 0) {
		atomicAdd((unsigned long long*)&dbg[0], (unsigned long long)d_steps);
		atomicAdd((unsigned long long*)&dbg[1], (unsigned long long)d_one);
		atomicAdd((unsigned long long*)&dbg[2], (unsigned long long)d_all);
		atomicAdd((unsigned long long*)&dbg[3], (unsigned long long)d_cycles);
		atomicAdd((unsigned long long*)&dbg[4], (unsigned long long)(hi - lo));
		atomicAdd((unsigned long long*)&dbg[5], 1ull);
	}
	return 1;
}
__device__ __forceinline__ bool radix_pass_bytes(unsigned long long *g, int lo, int hi, int shift, PassLds &L, const SortScratch &sc, long long *dbg = nullptr, long long *ph = nullptr, FewBuckets *fb = nullptr)
{
	return radix_pass_bytes_t<false>(g, lo, hi, shift, &L, L.where, sc, dbg, ph, fb) == 1;
}

// ---- TWO short runs walked side by side (round 6) ---------------------------------------------------------------------------------------
// A pass's walk is one lane's chain of dependent LDS reads: 63 lanes wait.  Runs of up to half the lines' bytes are taken in PAIRS: their histograms,
// byte sequences and records go into one wave's LDS side by side (2 KB of records, 3 584 bytes and a 448-byte map of the elements that are not in
// their bucket, each), lane 0 walks the first run and lane 1 the second at the same time -- the same instructions, each on its own run -- and the
// elements of both follow their permutations with all lanes afterwards.  The walk is the reference's (ksort.h:128-139) on each run, so the
// arrays come out the same; what differs from the single form is only that a bucket's elements which must move are found from the map, bit by
// bit, by the walking lane itself (the single form's wave looks at 64 positions at a time).
constexpr int PAIR_MAX = LINE_STORE_BYTES / 2;
static_assert(PAIR_MAX % 64 == 0 && PAIR_MAX / 8 * 2 <= LINE_STORE_BYTES / 8, "two maps of PAIR_MAX bits fit in PassLds::owner");
struct PairRun { unsigned long long *g; int len, shift; SortScratch sc; };
// returns bit j set when run j was moved by its pass (clear: all its keys share the byte -- nothing was done to it)
__device__ __forceinline__ int radix_pass_pair(const PairRun &A, const PairRun &B, PassLds &L, long long *dbg)
{
	const int l = lane();
	int moved = 0;
	int d_steps = 0, d_cycles = 0;
#pragma unroll
	for (int j = 0; j < 2; ++j) {
		const PairRun &R = j == 0 ? A : B;
		int *cnt = L.where + 512 * j;
		int2 *rec = (int2*)cnt;
		unsigned char *lb = (unsigned char*)L.line + PAIR_MAX * j;
		unsigned *bm = (unsigned*)L.owner + (PAIR_MAX / 32) * j;
		const int len = R.len;
		for (int k = l; k < 256; k += W) cnt[k] = 0;
		wave_sync();
		for (int base = 0; base < len; base += 8 * W) {
			int byte[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) byte[u] = (int)(ZElem::key(R.g[min(base + u * W + l, len - 1)]) >> R.shift) & 255;
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const int i = base + u * W + l;
				if (i < len) { atomicAdd(&cnt[byte[u]], 1); lb[i] = (unsigned char)byte[u]; R.sc.perm[i] = i; }
			}
		}
		wave_sync();
		const int c[4] = { cnt[4 * l], cnt[4 * l + 1], cnt[4 * l + 2], cnt[4 * l + 3] };
		const bool same = __ballot(c[0] == len || c[1] == len || c[2] == len || c[3] == len) != 0;
		int inc = c[0] + c[1] + c[2] + c[3];
		const int own = inc;
		for (int off = 1; off < W; off <<= 1) { const int o = __shfl_up(inc, off); if (l >= off) inc += o; }
		int at = inc - own;
		wave_sync();                                            // (the counts are in registers: the records take their place)
#pragma unroll
		for (int q = 0; q < 4; ++q) { rec[4 * l + q] = make_int2(at, at + c[q]); at += c[q]; }
		wave_sync();
		// the map: an element is where it belongs when its position lies inside its own bucket
		for (int base = 0; base < len; base += W) {
			const int i = base + l;
			bool out = false;
			if (i < len) { const int2 r = rec[lb[i]]; out = i < r.x || i >= r.y; }
			const unsigned long long m = __ballot(out);
			if (l == 0) { bm[base / 32] = (unsigned)m; bm[base / 32 + 1] = (unsigned)(m >> 32); }
		}
		wave_sync();
		if (!same) moved |= 1 << j;
	}
	// the two walks, lane j on run j
	if (l < 2 && ((moved >> l) & 1)) {
		int2 *rec = (int2*)(L.where + 512 * l);
		const unsigned char *lb = (const unsigned char*)L.line + PAIR_MAX * l;
		const unsigned *bm = (const unsigned*)L.owner + (PAIR_MAX / 32) * l;
		int32_t *perm = l == 0 ? A.sc.perm : B.sc.perm;
		int steps = 0, cycles = 0;
		for (int k = 0; k < 256; ++k) {
			const int2 rk = rec[k];
			int i = rk.x;
			const int end = rk.y;
			while (i < end) {
				const unsigned w = bm[i >> 5] >> (i & 31);
				if (w == 0) { i = (i | 31) + 1; continue; }
				i += __builtin_ctz(w);
				if (i >= end) break;
				const int home = i;
				int d = lb[home], src = home;
				++cycles;
				do {
					const int pos = rec[d].x;
					const int nb = lb[pos];
					perm[src] = pos;
					rec[d].x = pos + 1;
					src = pos; d = nb; ++steps;
				} while (d != k);
				perm[src] = home;
				i = home + 1;
			}
		}
		d_steps = steps; d_cycles = cycles;
	}
	wave_sync();
	// the elements follow the permutations: out of place first, then back (all lanes)
#pragma unroll
	for (int j = 0; j < 2; ++j) {
		if (!((moved >> j) & 1)) continue;
		const PairRun &R = j == 0 ? A : B;
		for (int base = 0; base < R.len; base += 8 * W) {
			unsigned long long e[8]; int to[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) { const int i = min(base + u * W + l, R.len - 1); e[u] = R.g[i]; to[u] = R.sc.perm[i]; }
#pragma unroll
			for (int u = 0; u < 8; ++u) if (base + u * W + l < R.len) R.sc.tmp[to[u]] = e[u];
		}
	}
	wave_sync();
#pragma unroll
	for (int j = 0; j < 2; ++j) {
		if (!((moved >> j) & 1)) continue;
		const PairRun &R = j == 0 ? A : B;
		for (int base = 0; base < R.len; base += 8 * W) {
			unsigned long long e[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) e[u] = R.sc.tmp[min(base + u * W + l, R.len - 1)];
#pragma unroll
			for (int u = 0; u < 8; ++u) if (base + u * W + l < R.len) R.g[base + u * W + l] = e[u];
		}
	}
	wave_sync();
	if (dbg) {
		const int s0 = __builtin_amdgcn_readlane(d_steps, 0) + __builtin_amdgcn_readlane(d_steps, 1), c0 = __builtin_amdgcn_readlane(d_cycles, 0) + __builtin_amdgcn_readlane(d_cycles, 1);
		if (l == 0) {
			atomicAdd((unsigned long long*)&dbg[0], (unsigned long long)s0);
			atomicAdd((unsigned long long*)&dbg[3], (unsigned long long)c0);
			atomicAdd((unsigned long long*)&dbg[4], (unsigned long long)((moved & 1 ? A.len : 0) + (moved & 2 ? B.len : 0)));
			atomicAdd((unsigned long long*)&dbg[5], (unsigned long long)__popc(moved));
		}
	}
	return moved;
}

// radix_sort_128x (ksort.h:147-151) of g[0, n) by key, same final element order as the host's.
// The host recurses bucket by bucket; buckets are independent, so the same work is done here level by level: at the level of
// key byte `shift` the array is made of runs of elements that agree on all higher key bytes; a run longer than 64 gets a
// radix pass on this byte, a run of 2..64 is insertion-sorted (stable: rs_insertsort, ksort.h:105-115; a run that was sorted
// one level up is seen again: a no-op).
// Passes on bytes in which all keys of the run agree move nothing, so starting at the highest byte in which any two keys
// differ equals the host's start at byte 7.
#ifndef MM2GB_POST_SORT_ELEMENTS
#define MM2GB_POST_SORT_ELEMENTS 0              // 1: the candidates' sort moves elements step by step too (round 4; A/B builds)
#endif
template <class E, bool BYTES>
__device__ __forceinline__ bool one_radix_pass(typename E::T *g, int lo, int hi, int shift, PassLds &L, const SortScratch *sc, long long *dbg, long long *ph = nullptr, FewBuckets *fb = nullptr)
{
	if constexpr (BYTES && !MM2GB_POST_SORT_ELEMENTS) return radix_pass_bytes(g, lo, hi, shift, L, *sc, dbg, ph, fb);
	else return radix_pass<E>(g, lo, hi, shift, L, dbg);
}

// One level of the sort over g[0, n): runs of equal key >> (shift + 8) (at the top level the whole array is one run by construction).  The
// array is taken 64 elements at a time from the start of a run: a run that does not end within them is a LONG run -- long_run(first, end) is
// called for it (the radix pass on this byte, now or as a task of its own) --; otherwise ALL the runs that end within the 64 are
// insertion-sorted together -- every lane ranks its element among those of its own run -- with one load and one store for the lot (short runs
// are many: one round trip each would be the whole cost).  Returns the number of such groups that had to be reordered.
template <class E, class LongRun>
__device__ __forceinline__ int sort_level(typename E::T *g, int n, int shift, LongRun &&long_run)
{
	const int l = lane();
	int d_small = 0;
	int pos = 0;
	while (pos < n) {
		const int i = pos + l, n_in = min(W, n - pos);
		const bool in = i < n;
		const typename E::T e = g[in ? i : n - 1];
		const unsigned long long k = E::key(e);
		const unsigned long long pk = shift < 56 ? k >> (shift + 8) : 0;
		const bool has_after = pos + W < n;
		const unsigned long long pk_after = has_after && shift < 56 ? E::key(g[pos + W]) >> (shift + 8) : 0;
		const unsigned long long pk_prev = shfl_up64(pk, 1), k_prev = shfl_up64(k, 1);
		const bool start = in && (l == 0 || pk != pk_prev);
		const unsigned long long starts = __ballot(start);
		const bool tail_open = has_after && pk_after == readlane64(pk, n_in - 1);   // the last run goes on beyond these 64
		if (tail_open && (starts & (starts - 1)) == 0) {
			// a single run of more than 64 elements: where it ends
			const unsigned long long pk0 = readlane64(pk, 0);
			int q = pos + W, adv;
			do {
				// four blocks of 64 per round trip (a run of 25 000 elements is 400 of them)
				unsigned long long pk4[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) { const int i2 = q + u * W + l; pk4[u] = i2 < n && shift < 56 ? E::key(g[i2]) >> (shift + 8) : 0; }
				adv = W;
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					if (adv == W) {
						const int i2 = q + l;
						const unsigned long long out = __ballot(i2 >= n || pk4[u] != pk0);
						adv = ((unsigned)out | (unsigned)(out >> 32)) ? first_set(out) : W;
						q += adv;
					}
				}
			} while (adv == W);
			long_run(pos, q);
			pos = q;
			continue;
		}
		const int end_c = tail_open ? 63 - first_set_from_top(starts) : n_in;   // the open run (if any) starts the next 64
		const int rs = 63 - __clzll(starts & ((2ull << l) - 1));                // where this lane's run starts
		const bool act = l < end_c;
		if (__ballot(act && !start && k < k_prev) != 0) {
			int rank = 0;
			for (int m = 0; m < end_c; ++m) {
				const unsigned long long km = readlane64(k, m);
				const int rsm = __builtin_amdgcn_readlane(rs, m);
				rank += (rsm == rs) & ((km < k) | ((km == k) & (m < l)));
			}
			wave_sync();
			if (act) g[pos + rs + rank] = e;
			++d_small;
		}
		wave_sync();
		pos += end_c;
	}
	wave_sync();
	return d_small;
}

// the highest key byte in which any two of g[0, n) differ (the sort starts there: passes on bytes in which all keys agree move nothing), -8 if none
template <class E>
__device__ __forceinline__ int top_byte_shift(const typename E::T *g, int n)
{
	const int l = lane();
	unsigned long long any = 0, all = ~0ull;
	for (int base = 0; base < n; base += 8 * W) {           // (eight loads in flight per lane: a wave on its own pays every round trip in full)
		unsigned long long k[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) k[u] = E::key(g[min(base + u * W + l, n - 1)]);
#pragma unroll
		for (int u = 0; u < 8; ++u) { any |= k[u]; all &= k[u]; }
	}
	for (int off = W / 2; off > 0; off >>= 1) {
		any |= (unsigned long long)(unsigned)__shfl_xor((int)(unsigned)any, off) | (unsigned long long)(unsigned)__shfl_xor((int)(unsigned)(any >> 32), off) << 32;
		all &= (unsigned long long)(unsigned)__shfl_xor((int)(unsigned)all, off) | (unsigned long long)(unsigned)__shfl_xor((int)(unsigned)(all >> 32), off) << 32;
	}
	const unsigned long long diff = any ^ all;
	if (diff == 0) return -8;
	int top = 56;
	while (top > 0 && ((diff >> top) & 255) == 0) top -= 8;
	return top;
}

template <class E, bool BYTES = false>
__device__ __forceinline__ void sort_like_host(typename E::T *g, int n, PassLds &L, long long *dbg = nullptr, const SortScratch *sc = nullptr)
{
	if (n <= 1) return;
	if (n <= SMALL_RUN) { small_run_sort<E>(g, 0, n); return; }
	const int top = top_byte_shift<E>(g, n);
	if (top < 0) return;
	long long tlev = dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0, d_pass = 0, d_elems = 0, d_small = 0;
	for (int shift = top; shift >= 0; shift -= 8) {
		d_small += sort_level<E>(g, n, shift, [&](int first, int end) {
			one_radix_pass<E, BYTES>(g, first, end, shift, L, sc, dbg ? dbg + 24 + 6 * min((top - shift) / 8, 2) : nullptr);
			++d_pass; d_elems += end - first;
		});
		if (dbg && lane() == 0) {
			const long long tn = (long long)__builtin_amdgcn_s_memrealtime();
			const int lvl = (top - shift) / 8;
			atomicAdd((unsigned long long*)&dbg[13 + (lvl < 3 ? lvl : 3)], (unsigned long long)(tn - tlev));
			tlev = tn;
		}
	}
	if (dbg && lane() == 0) {
		atomicAdd((unsigned long long*)&dbg[17], (unsigned long long)d_pass);
		atomicAdd((unsigned long long*)&dbg[18], (unsigned long long)d_elems);
		atomicAdd((unsigned long long*)&dbg[19], (unsigned long long)d_small);
	}
}

// chains of read r hold at least max(1, min_cnt) anchors each, so at most n_r / mc of them: slot of read r in the per-chain arrays
__device__ __forceinline__ int64_t chain_slot(int64_t off_r, int64_t r, int mc) { return off_r / mc + r; }

} // namespace

// --------------------------------------------------------------------------------------------------------------
// reads by size, largest first: N_SIZE_CLASSES classes, eight per power of two, so reads of one class differ by at most an eighth (order
// inside a class is arbitrary: reads are independent, results do not depend on it).  The kernel ends with its largest reads: they start
// first, and the very largest are the ones whole workgroups start on (k_post_chains).
// --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int size_class(int64_t n)
{
	if (n <= 0) return 0;
	const int msb = 63 - __clzll((unsigned long long)n);                      // 0..62
	const int frac = msb >= 3 ? (int)((n >> (msb - 3)) & 7) : (int)((n << (3 - msb)) & 7);
	return min(N_SIZE_CLASSES - 1, 1 + msb * 8 + frac);
}

// Items into bins with the bins' counters in memory: a workgroup first counts its own items per bin in LDS and then takes a range of every bin it
// needs with ONE atomic (atomics on one address are served one after the other, ~30 ns each: a level's tens of thousands of tasks crowd a few
// dozen bins).  count: bins[c] += the workgroup's items of bin c.  scatter: returns the item's slot (bins[c] is the bin's cursor).
// Every thread of the workgroup calls; `active` says whether it has an item.  s_cnt / s_base: N_SIZE_CLASSES ints of LDS each.
__device__ __forceinline__ void block_bin_count(int32_t *bins, int c, bool active, int *s_cnt)
{
	for (int k = threadIdx.x; k < N_SIZE_CLASSES; k += blockDim.x) s_cnt[k] = 0;
	__syncthreads();
	if (active) atomicAdd(&s_cnt[c], 1);
	__syncthreads();
	for (int k = threadIdx.x; k < N_SIZE_CLASSES; k += blockDim.x) if (s_cnt[k]) atomicAdd(&bins[k], s_cnt[k]);
	__syncthreads();
}
__device__ __forceinline__ int block_bin_slot(int32_t *bins, int c, bool active, int *s_cnt, int *s_base)
{
	for (int k = threadIdx.x; k < N_SIZE_CLASSES; k += blockDim.x) s_cnt[k] = 0;
	__syncthreads();
	int mine = 0;
	if (active) mine = atomicAdd(&s_cnt[c], 1);
	__syncthreads();
	for (int k = threadIdx.x; k < N_SIZE_CLASSES; k += blockDim.x) if (s_cnt[k]) s_base[k] = atomicAdd(&bins[k], s_cnt[k]);
	__syncthreads();
	const int at = active ? s_base[c] + mine : 0;
	__syncthreads();
	return at;
}

__global__ __launch_bounds__(256) void k_post_size_count(PostBatch b)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r < b.n_reads) atomicAdd(&b.size_bins[size_class(b.offsets[r + 1] - b.offsets[r])], 1);
}

// counts -> first slot of every class (largest class first), one wave
__global__ __launch_bounds__(64) void k_post_size_bases(PostBatch b)
{
	if (threadIdx.x != 0) return;
	int acc = 0;
	for (int k = N_SIZE_CLASSES - 1; k >= 0; --k) { const int c = b.size_bins[k]; b.size_bins[N_SIZE_CLASSES + k] = acc; acc += c; }
}

__global__ __launch_bounds__(256) void k_post_size_scatter(PostBatch b)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= b.n_reads) return;
	const int c = size_class(b.offsets[r + 1] - b.offsets[r]);
	b.order[atomicAdd(&b.size_bins[N_SIZE_CLASSES + c], 1)] = (int)r;
}

// --------------------------------------------------------------------------------------------------------------
// lifting tables of the predecessor links: up4[i] / up16[i] = distance from anchor i to the anchor 4 / 16 links down its path
// (0 = the path is shorter).  p is relative and never leaves the read, so this is one pass over all anchors of the batch per level.
// The first level also writes the walks' records fp[i] = (f[i], p[i]): a walk asks for an anchor's score, link and "taken" flag
// together, at an address no other lane of the wave is near -- one 8-byte load instead of three loads from three arrays (the flag is
// the top bit of .y; p is never negative).
// --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_post_lift(PostBatch b, int level)
{
	const int32_t *src = level == 0 ? b.p : b.up4;
	int32_t *dst = level == 0 ? b.up4 : b.up16;
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < b.n; g += (int64_t)gridDim.x * blockDim.x) {
		const int r0 = src[g];
		if (level == 0) b.fp[g] = make_int2(b.f[g], r0);           // the walks' record of the anchor, nothing taken yet (lchain.c:43)
		int64_t t = r0 ? g - r0 : -1;
		for (int k = 1; k < 4 && t >= 0; ++k) { const int rj = src[t]; t = rj ? t - rj : -1; }
		dst[g] = t < 0 ? 0 : (int32_t)(g - t);
	}
}

// --------------------------------------------------------------------------------------------------------------
// per read: candidates, host order, chain walks (lchain.c:27-76)
// --------------------------------------------------------------------------------------------------------------
namespace {

struct WalkDbg { long long load = 0, longt = 0, groups = 0, open = 0, nlong = 0, iters = 0; };

// The chain walks of one read (lchain.c:44-74) over its sorted candidates z[0, n_z): one wave.
// `bits` (round 5): one "taken" bit per anchor of the read in the wave's LDS (the sort's scratch, free by now) instead of the top bit of the
// anchor's record in memory.  60 % of all candidates are found taken on their first probe (their chain's best end came earlier): that probe,
// the marks of every walk and the second looks of a group become LDS traffic, and nothing but scores and links is ever loaded.  Null for a
// read with more anchors than the scratch has bits: the flag then lives in the record as before.
// Split form (round 6, k_post_walk): z is ONE CLASS of the read's sorted candidates (the trees whose roots hash to it; sorted order kept), kp
// their positions in the whole read's order; a chain that is kept is recorded under its end's position (what.endslot: its chain slot, read-relative,
// what.u_loc: where its anchors start in the read's picked[]), so that the read's chains can be put in the host's order afterwards.
struct WalkSplit { const int32_t *kp = nullptr; int32_t *endslot = nullptr; int32_t *u_loc = nullptr; int picked_base = 0, slot_base = 0; };
__device__ __forceinline__ void post_walk_read(const PostBatch &b, const int64_t off, const int n_z, const unsigned long long *z, int2 *fp,
                                               int32_t *picked, unsigned long long *u_tmp, int &n_u_out, int &n_v_out, WalkDbg &wd, unsigned *bits,
                                               const WalkSplit &what = WalkSplit())
{
	const int l = lane();
	long long &dbg_load = wd.load, &dbg_longt = wd.longt, &dbg_groups = wd.groups, &dbg_open = wd.open, &dbg_long = wd.nlong;
	// best-scoring end first; every anchor walked is consumed even if its chain is dropped (lchain.c:59-71)
	// A walk (mg_chain_bk_end, lchain.c:9-25: back from the chain end until an anchor that is taken, the start of the path, or an
	// X-drop of more than max_drop below the best prefix) is a chain of dependent loads, one memory round trip per anchor, and
	// walks depend on each other through the marks.  Candidates are handled 64 at a time, in the host's order.
	const int32_t *up4 = b.up4 + off, *up16 = b.up16 + off;
	constexpr int TAKEN = INT_MIN;                           // top bit of fp[i].y
	int *fpw = (int*)fp;                                     // fpw[2 i + 1] = fp[i].y
	auto taken = [&](int i) { return (int)((bits[i >> 5] >> (i & 31)) & 1u) != 0; };
	auto take = [&](int i) { atomicOr(&bits[i >> 5], 1u << (i & 31)); };
	int n_u = 0, n_v = 0;
	for (int kb = n_z - 1; kb >= 0; kb -= W) {
		const int k_l = kb - l;
		const unsigned long long z_l = k_l >= 0 ? z[k_l] : 0;
		const int kp_l = what.kp && k_l >= 0 ? what.kp[k_l] : 0;
		const int n0 = (int)(unsigned)z_l, top_l = (int)(z_l >> 32);
		unsigned long long pending = __ballot(k_l >= 0);
		int nx[SPEC], sx[SPEC];                                 // the group's look-ahead: anchors SPEC steps down every candidate's path, and their score drops
		bool fresh = true;
		while (pending) {
			// Every pending candidate that is still free takes the first SPEC steps of its own walk, all lanes at once: SPEC + 1 memory
			// round trips for the group instead of for each candidate.  Scores and links never change; marks only ever get set, which
			// can only end a walk earlier -- so a lane may stop loading where its walk would end with the marks it sees now.
			// After a long walk of the group the lanes that are left look again: the paths are the same, only marks can have
			// changed (a walk can only END earlier than it did), so the second look reads the marks of the anchors it already
			// knows -- one round trip of independent loads instead of SPEC + 1 dependent ones.
			wave_sync();
			const long long ta = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
			int p0 = TAKEN, pn[SPEC];                            // link | taken of the candidate and of the anchors down its path
			int own_kept = 0, own_best = 0, own_ended = 0, touched = 0;   // the lane's own walk as far as it can tell now
#pragma unroll
			for (int j = 0; j < SPEC; ++j) pn[j] = TAKEN;
			if (fresh) {
#pragma unroll
				for (int j = 0; j < SPEC; ++j) { nx[j] = -1; sx[j] = top_l; }
			}
			if ((pending >> l) & 1) {
				p0 = bits ? (taken(n0) ? TAKEN : fpw[2 * n0 + 1]) : fpw[2 * n0 + 1];
				if (fresh) {
					int pc = p0;
					if (p0 >= 0) {
						int cur = n0, best = 0, kept = 0;
						bool ended = false;
#pragma unroll
						for (int j = 0; j < SPEC; ++j) {
							if (!ended) {
								const int next = pc ? cur - pc : -1;
								nx[j] = next;
								if (next >= 0) { const int2 rec = fp[next]; sx[j] = top_l - rec.x; pn[j] = bits && taken(next) ? rec.y | TAKEN : rec.y; pc = rec.y; }
								if (sx[j] > best) { best = sx[j]; kept = j + 1; }
								else if (best - sx[j] > b.max_drop) ended = true;
								if (pn[j] < 0) ended = true;
								cur = next;
							}
						}
						own_kept = kept; own_best = best; own_ended = ended;
					}
				} else {
#pragma unroll
					for (int j = 0; j < SPEC; ++j) if (nx[j] >= 0) pn[j] = bits ? (taken(nx[j]) ? TAKEN : 0) : fpw[2 * nx[j] + 1];   // (only the sign is looked at when the marks are bits)
					if (p0 >= 0) {
						int best = 0, kept = 0;
						bool ended = false;
#pragma unroll
						for (int j = 0; j < SPEC; ++j) {
							if (!ended) {
								if (sx[j] > best) { best = sx[j]; kept = j + 1; }
								else if (best - sx[j] > b.max_drop) ended = true;
								if (pn[j] < 0) ended = true;
							}
						}
						own_kept = kept; own_best = best; own_ended = ended;
					}
				}
			}
			fresh = false;
			// The candidates in order, on wave-uniform copies of their lane's values -- no memory round trip for a walk that ends
			// within SPEC steps.  What an earlier walk of the group takes is set in the later lanes' copies of the marks.
			unsigned long long open = __ballot(p0 >= 0);
			bool stale = false;
			if (b.dbg) { dbg_load += (long long)__builtin_amdgcn_s_memrealtime() - ta; ++dbg_groups; dbg_open += __popcll(open); }
			while (open != 0 && !stale) {
				const int src = first_set(open);                 // lowest lane = highest k
				open &= open - 1;
				if (__builtin_amdgcn_readlane(p0, src) < 0) continue;    // taken by an earlier walk of this group
				const int c0 = __builtin_amdgcn_readlane(n0, src);
				if (__builtin_amdgcn_readlane(own_ended, src) != 0 && __builtin_amdgcn_readlane(touched, src) == 0) {
					// no earlier walk of the group took any anchor this lane looked at: its own evaluation stands (most candidates:
					// one or two anchors, then an anchor that was taken long ago)
					const int kept = __builtin_amdgcn_readlane(own_kept, src), best = __builtin_amdgcn_readlane(own_best, src);
#pragma unroll
					for (int j = 0; j < SPEC; ++j) {
						if (j < kept) {
							const int v = j == 0 ? c0 : __builtin_amdgcn_readlane(nx[j > 0 ? j - 1 : 0], src);
							const int link = j == 0 ? __builtin_amdgcn_readlane(p0, src) : __builtin_amdgcn_readlane(pn[j > 0 ? j - 1 : 0], src);
							if (l == 0) { picked[n_v + j] = v; if (bits) take(v); else fpw[2 * v + 1] = link | TAKEN; }
							const bool hit = (n0 == v) | (nx[0] == v) | (nx[1] == v) | (nx[2] == v) | (nx[3] == v);
							touched |= hit;
							p0 |= n0 == v ? TAKEN : 0;
#pragma unroll
							for (int i = 0; i < SPEC; ++i) pn[i] |= nx[i] == v ? TAKEN : 0;
						}
					}
					if (best >= b.min_sc && kept > 0 && kept >= b.min_cnt) {
						if (l == 0) u_tmp[n_u] = (unsigned long long)(unsigned)best << 32 | (unsigned)kept;
						if (what.kp) { const int kp = __builtin_amdgcn_readlane(kp_l, src); if (l == 0) { what.u_loc[n_u] = what.picked_base + n_v; what.endslot[kp] = what.slot_base + n_u; } }
						++n_u; n_v += kept;
					}
					continue;
				}
				const int top = __builtin_amdgcn_readlane(top_l, src);
				int cn[SPEC], cs[SPEC], cm[SPEC];
#pragma unroll
				for (int j = 0; j < SPEC; ++j) {
					cn[j] = __builtin_amdgcn_readlane(nx[j], src); cs[j] = __builtin_amdgcn_readlane(sx[j], src); cm[j] = __builtin_amdgcn_readlane(pn[j], src);
				}
				// `kept` is how many of the visited anchors lie before the one the best prefix stops at
				int cur = c0, kept = 0, visited = 0, best = 0;
				bool ended = false;
#pragma unroll
				for (int j = 0; j < SPEC; ++j) {
					if (!ended) {
						if (l == 0) picked[n_v + visited] = cur;
						++visited;
						if (cs[j] > best) { best = cs[j]; kept = visited; }
						else if (best - cs[j] > b.max_drop) ended = true;
						if (cm[j] < 0) ended = true;
						cur = cn[j];
					}
				}
				if (ended) {
					// at most SPEC anchors taken: marked from the registers, and noted in the lanes that come later
#pragma unroll
					for (int j = 0; j < SPEC; ++j) {
						if (j < kept) {
							const int v = j == 0 ? c0 : cn[j > 0 ? j - 1 : 0];
							const int link = j == 0 ? __builtin_amdgcn_readlane(p0, src) : cm[j > 0 ? j - 1 : 0];
							if (l == 0) { if (bits) take(v); else fpw[2 * v + 1] = link | TAKEN; }
							const bool hit = (n0 == v) | (nx[0] == v) | (nx[1] == v) | (nx[2] == v) | (nx[3] == v);
							touched |= hit;
							p0 |= n0 == v ? TAKEN : 0;
#pragma unroll
							for (int i = 0; i < SPEC; ++i) pn[i] |= nx[i] == v ? TAKEN : 0;
						}
					}
				} else {
					// a walk that goes on: 64 anchors at a time, lane j finds the j-th anchor down the path through the lifting tables
					// (p, p^4, p^16: at most 9 dependent loads instead of j); what the sequential loop decides step by step -- the running
					// best prefix, the first step that ends the walk -- becomes a prefix maximum and a ballot over the wave.  It may take
					// anchors the other lanes have looked at: they look again afterwards -- unless the walk ended within its first round (most do):
					// then everything it took is in registers, and the later lanes' copies of the marks are set from there as after a short walk.
					wave_sync();
					const long long tl = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
					++dbg_long;
					// the first round looks at 16 anchors only: most walks that go on end within a dozen steps (the chains of a repeat), and a lane
					// beyond the walk's end costs its loads all the same -- divergent loads are what this kernel is made of
					int rw = W / 4, n_rounds = 0, t_first = -1, pt_first = 0;
					while (!ended) {
						++wd.iters; ++n_rounds;
						int t = l < rw ? cur : -1;
						for (int k = 0; k < 3; ++k) if (k < (l >> 4) && t >= 0) { const int rj = up16[t]; t = rj ? t - rj : -1; }
						for (int k = 0; k < 3; ++k) if (k < ((l >> 2) & 3) && t >= 0) { const int rj = up4[t]; t = rj ? t - rj : -1; }
						for (int k = 0; k < 3; ++k) if (k < (l & 3) && t >= 0) { const int rj = fpw[2 * t + 1] & ~TAKEN; t = rj ? t - rj : -1; }
						const bool valid = t >= 0;
						// Every lane loads the record of ITS anchor, once: its link is the lane's step, and score and mark of the anchor the step
						// arrives at are the NEXT lane's record (lane l + 1 holds p^(l+1) = the predecessor of lane l's anchor) -- a shuffle instead
						// of a second load behind the first.  The round's last lane only lends its record: a round decides rw - 1 steps.
						const int rd = rw - 1;
						int2 own = make_int2(0, TAKEN);
						if (valid) own = fp[t];
						const int pt = valid ? own.y & ~TAKEN : 0, next = pt ? t - pt : -1;
						if (n_rounds == 1) { t_first = t; pt_first = pt; }
						const int nx_f = __shfl_down(own.x, 1), nx_y = __shfl_down(own.y, 1);
						int s = top, m = 1;
						if (next >= 0) { s = top - nx_f; m = bits ? (int)taken(next) : (int)(nx_y < 0); }
						// best prefix BEFORE this lane's step
						int inc = valid && l < rd ? s : INT_MIN;
						for (int o = 1; o < W; o <<= 1) { const int v = __shfl_up(inc, o); if (l >= o) inc = max(inc, v); }
						int before = __shfl_up(inc, 1);
						before = l == 0 ? best : max(best, before);
						const bool newmax = valid && s > before;
						const bool ends = !valid || (!newmax && before - s > b.max_drop) || m != 0;
						const unsigned long long endm = __ballot(ends && l < rd);
						const int jb = endm ? first_set(endm) : rd;            // the step that ends the walk (all of it is still taken)
						if (valid && l <= jb && l < rd) picked[n_v + visited + l] = t;
						const unsigned long long nm = __ballot(newmax && l <= jb && l < rd);
						if (nm) {
							const int last = 63 - first_set_from_top(nm);
							best = __shfl(s, last);
							kept = visited + last + 1;
						}
						if (jb < rd) { visited += jb + 1; ended = true; }
						else { visited += rd; cur = __shfl(t, rd); rw = W; }
					}
					wave_sync();
					if (n_rounds == 1 && !bits) {
						// the first SPEC anchors it took are the look-ahead's, the others the round's (one per lane)
#pragma unroll
						for (int j = 0; j < SPEC; ++j) {
							if (j < kept) {
								const int v = j == 0 ? c0 : cn[j > 0 ? j - 1 : 0];
								const int link = j == 0 ? __builtin_amdgcn_readlane(p0, src) : cm[j > 0 ? j - 1 : 0];
								if (l == 0) fpw[2 * v + 1] = link | TAKEN;
								const bool hit = (n0 == v) | (nx[0] == v) | (nx[1] == v) | (nx[2] == v) | (nx[3] == v);
								touched |= hit;
								p0 |= n0 == v ? TAKEN : 0;
#pragma unroll
								for (int i = 0; i < SPEC; ++i) pn[i] |= nx[i] == v ? TAKEN : 0;
							}
						}
						if (l < kept - SPEC) fpw[2 * t_first + 1] = pt_first | TAKEN;
						for (int q = SPEC; q < kept; ++q) {
							const int v = __builtin_amdgcn_readlane(t_first, q - SPEC);
							const bool hit = (n0 == v) | (nx[0] == v) | (nx[1] == v) | (nx[2] == v) | (nx[3] == v);
							touched |= hit;
							p0 |= n0 == v ? TAKEN : 0;
#pragma unroll
							for (int i = 0; i < SPEC; ++i) pn[i] |= nx[i] == v ? TAKEN : 0;
						}
					} else {
						stale = true;
						if (bits) for (int q = l; q < kept; q += W) take(picked[n_v + q]);
						else for (int q = l; q < kept; q += W) { int *w = &fpw[2 * picked[n_v + q] + 1]; *w |= TAKEN; }
					}
					if (b.dbg) dbg_longt += (long long)__builtin_amdgcn_s_memrealtime() - tl;
				}
				// the chain's score is the best prefix itself (lchain.c:66: f of the end minus f of where it stops)
				if (best >= b.min_sc && kept > 0 && kept >= b.min_cnt) {
					if (l == 0) u_tmp[n_u] = (unsigned long long)(unsigned)best << 32 | (unsigned)kept;
					if (what.kp) { const int kp = __builtin_amdgcn_readlane(kp_l, src); if (l == 0) { what.u_loc[n_u] = what.picked_base + n_v; what.endslot[kp] = what.slot_base + n_u; } }
					++n_u; n_v += kept;
				}
			}
			pending = stale ? open : 0;
		}
	}
	n_u_out = n_u; n_v_out = n_v;
}

// candidates of anchors [i_lo, i_hi) of a read, in index order (lchain.c:35-41), appended at z[at...] (nothing is taken yet: k_post_lift wrote the records).
// Returns how many; any / all: OR and AND of their keys (which key bytes differ at all).
__device__ __forceinline__ int post_collect(const PostBatch &b, const int32_t *f, unsigned long long *z, int i_lo, int i_hi, int at, bool write,
                                            unsigned &any, unsigned &all)
{
	const int l = lane();
	int n_z = 0;
	for (int base = i_lo; base < i_hi; base += W) {
		const int i = base + l;
		const bool in = i < i_hi;
		const int fi = in ? f[i] : INT_MIN;
		const bool take = in && fi >= b.min_sc;
		const unsigned long long m = __ballot(take);
		if (take) { any |= (unsigned)fi; all &= (unsigned)fi; }
		if (take && write) z[at + n_z + __popcll(m & ((1ull << l) - 1))] = (unsigned long long)(unsigned)fi << 32 | (unsigned)i;
		n_z += __popcll(m);
	}
	return n_z;
}

} // namespace

// One wave per read, except at the very start of the launch: reads come largest first, and the kernel ends with the largest ones -- a
// read is one wave's serial work (the sort's cycle walks, the chain walks), ~0.45 us per anchor, while most of the chip has long run out of
// reads.  So the FIRST read a workgroup takes (the `team_reads` largest of the batch) is shared by its four waves as far as the
// reference's algorithm allows: candidates are collected by all four (each a quarter of the anchors), the top radix pass -- one token walk
// through 256 bucket heads, inherently serial (ksort.h:116-146) -- is wave 0's, and its buckets, which the host sorts independently of each
// other (rs_sort's recursion, ksort.h:140-145), are dealt to the four waves, largest first.  The chain walks are wave 0's again (they
// depend on each other through the marks); the other three waves go on to reads of their own.
// SORT_ONLY (round 6): the same up to the sorted candidates, whose number goes to read_nz[]; the walks are k_post_walk's.
template <bool SORT_ONLY>
__device__ __forceinline__ void post_chains_body(const PostBatch &b, int team_reads)
{
	__shared__ PassLds lds[POST_THREADS / W];
	__shared__ int s_team[8];                                // [0] the team's read (position in the order), [1..4] candidates per wave, [5] next task
	__shared__ unsigned s_bits[2 * (POST_THREADS / W)];      // per wave: OR / AND of its candidates' keys
	__shared__ int s_bound[257];                             // ends of the top pass's buckets
	__shared__ int s_task[256];                              // buckets worth a task, largest first
	PassLds &L = lds[threadIdx.x / W];
	const int l = lane(), w = uni(threadIdx.x / W);
	const int mc = b.min_cnt > 1 ? b.min_cnt : 1;
	// ---- the team's read ----
	bool walker = false;                                     // wave 0 of a team: sorted candidates wait for its walks
	int team_r = -1, team_nz = 0, solo_q = -1;
	if (team_reads > 0) {
		if (threadIdx.x == 0) s_team[0] = atomicAdd(b.cursor, 1);
		__syncthreads();
		const int q = uni(s_team[0]);
		if (q < team_reads && q < b.n_reads) {
			const int r = uni(b.order[q]);
			const int64_t off = b.offsets[r];
			const int n = (int)(b.offsets[r + 1] - off);
			const int32_t *f = b.f + off;
			unsigned long long *z = b.z + off;
			// candidates: every wave a quarter of the anchors (whole groups of 64), counted first, then written behind the earlier quarters'
			const int per = ((n + 4 * W - 1) / (4 * W)) * W;
			const int i_lo = min(n, w * per), i_hi = min(n, (w + 1) * per);
			unsigned any = 0, all = ~0u;
			const int mine = post_collect(b, f, z, i_lo, i_hi, 0, false, any, all);
			for (int o = W / 2; o > 0; o >>= 1) { any |= __shfl_xor(any, o); all &= __shfl_xor(all, o); }
			if (l == 0) { s_team[1 + w] = mine; s_bits[2 * w] = any; s_bits[2 * w + 1] = all; }
			__syncthreads();
			int at = 0, n_z = 0;
			for (int k = 0; k < POST_THREADS / W; ++k) { const int c = uni(s_team[1 + k]); if (k < w) at += c; n_z += c; any |= s_bits[2 * k]; all &= s_bits[2 * k + 1]; }
			{ unsigned a2 = 0, b2 = ~0u; post_collect(b, f, z, i_lo, i_hi, at, true, a2, b2); }
			__threadfence_block();
			__syncthreads();
			// the top pass: the highest key byte in which any two candidates differ (sort_like_host), wave 0 alone
			const unsigned diff = uni((int)(any ^ all));
			int n_tasks = 0;
			if (n_z > SMALL_RUN && diff != 0) {
				int top = 24;                                      // of the key = the score: byte 3 .. 0 (sort_like_host's `top`)
				while (top > 0 && ((diff >> top) & 255u) == 0) top -= 8;
				if (w == 0) {
					const SortScratch sc = { b.sort_s + off, b.sort_perm + off, b.sort_tmp + off };
					one_radix_pass<ZElem, true>(z, 0, n_z, top, L, &sc, b.dbg ? b.dbg + 24 : nullptr);
					for (int k = l; k < 256; k += W) s_bound[k + 1] = MM2GB_POST_SORT_ELEMENTS ? L.tail[k] : ((const int4*)L.where)[k].w;   // the buckets' ends: the byte form keeps them in its records
					if (l == 0) s_bound[0] = 0;
					wave_sync();
					// buckets of more than one element are tasks, largest first (a few hold nearly everything: the scores of a read span two
					// to four values of the top byte): rank of every bucket by (size, number), all 256 against all 256
					int sz[4], rank[4] = { 0, 0, 0, 0 }, tasks = 0;
#pragma unroll
					for (int q = 0; q < 4; ++q) { sz[q] = s_bound[4 * l + q + 1] - s_bound[4 * l + q]; tasks += sz[q] > 1; }
					for (int k = 0; k < 256; ++k) {
						const int other = uni(s_bound[k + 1]) - uni(s_bound[k]);
#pragma unroll
						for (int q = 0; q < 4; ++q) rank[q] += (other > sz[q]) | ((other == sz[q]) & (k < 4 * l + q));
					}
#pragma unroll
					for (int q = 0; q < 4; ++q) s_task[rank[q]] = 4 * l + q;
					for (int o = W / 2; o > 0; o >>= 1) tasks += __shfl_xor(tasks, o);
					if (l == 0) { s_team[6] = tasks; s_team[5] = 0; }
				}
				__threadfence_block();
				__syncthreads();
				n_tasks = uni(s_team[6]);
				for (;;) {
					int t = 0;
					if (l == 0) t = atomicAdd(&s_team[5], 1);
					t = uni(t);
					if (t >= n_tasks) break;
					const int k = uni(s_task[t]);
					const int lo = uni(s_bound[k]), hi = uni(s_bound[k + 1]);
					const SortScratch sc = { b.sort_s + off + lo, b.sort_perm + off + lo, b.sort_tmp + off + lo };
					sort_like_host<ZElem, true>(z + lo, hi - lo, L, b.dbg, &sc);
				}
				__threadfence_block();
				__syncthreads();
			} else if (w == 0) { const SortScratch sc = { b.sort_s + off, b.sort_perm + off, b.sort_tmp + off }; sort_like_host<ZElem, true>(z, n_z, L, b.dbg, &sc); }
			if (w == 0) { walker = true; team_r = r; team_nz = n_z; }
		} else if (w == 0) solo_q = q;                       // not a team read (any more): wave 0's first read, on its own
	}
	for (bool first = true;; first = false) {
		int r = 0, n_z = 0;
		const bool team = first && walker;
		if (team) { r = team_r; n_z = team_nz; }
		else {
			if (first && solo_q >= 0) r = solo_q;
			else { if (l == 0) r = atomicAdd(b.cursor, 1); r = uni(r); }
			if (r >= b.n_reads) break;
			r = uni(b.order[r]);
		}
		const int64_t off = b.offsets[r];
		const int n = (int)(b.offsets[r + 1] - off);
		const int32_t *f = b.f + off;
		unsigned long long *z = b.z + off;
		int2 *fp = b.fp + off;
		int32_t *picked = b.picked + off;
		unsigned long long *u_tmp = b.u_tmp + chain_slot(off, r, mc);
		const long long t0 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
		long long t1 = t0, t2 = t0;
		if (!team) {
			unsigned any = 0, all = ~0u;
			n_z = post_collect(b, f, z, 0, n, 0, true, any, all);
			wave_sync();
			t1 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
			const SortScratch sc = { b.sort_s + off, b.sort_perm + off, b.sort_tmp + off };
			sort_like_host<ZElem, true>(z, n_z, L, b.dbg, &sc);
			t2 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
		}
		if constexpr (SORT_ONLY) {
			if (l == 0) {
				b.read_nz[r] = n_z;
				if (b.dbg) {
					if (b.dbg_reads) { b.dbg_reads[4 * r] = t0; b.dbg_reads[4 * r + 1] = t1; b.dbg_reads[4 * r + 2] = t2; b.dbg_reads[4 * r + 3] = team ? (long long)__builtin_amdgcn_s_memrealtime() : t2; }
					atomicAdd((unsigned long long*)&b.dbg[0], (unsigned long long)(t1 - t0));
					atomicAdd((unsigned long long*)&b.dbg[1], (unsigned long long)(t2 - t1));
					atomicMax((unsigned long long*)&b.dbg[4], (unsigned long long)(t2 - t1));
					atomicAdd((unsigned long long*)&b.dbg[12], (unsigned long long)n_z);
				}
			}
			wave_sync();
			continue;
		}
		int n_u = 0, n_v = 0;
		WalkDbg wd;
		// the marks of the walks: a bit per anchor in this wave's LDS (the sort is done with it) when the read fits
		constexpr int MARK_WORDS = (int)(sizeof(PassLds) / 4);
		unsigned *bits = n <= MARK_WORDS * 32 && !POST_MARKS_IN_MEMORY ? (unsigned*)&L : nullptr;
		if (bits) { wave_sync(); for (int k = l; k < (n + 31) / 32; k += W) bits[k] = 0; wave_sync(); }
		post_walk_read(b, off, n_z, z, fp, picked, u_tmp, n_u, n_v, wd, bits);
		wave_sync();
		if (l == 0) {
			b.n_u[r] = n_u;
			b.n_kept[r] = n_v;
			if (b.dbg) {
				const long long t3 = (long long)__builtin_amdgcn_s_memrealtime();
				if (b.dbg_reads) { b.dbg_reads[4 * r] = t0; b.dbg_reads[4 * r + 1] = t1; b.dbg_reads[4 * r + 2] = t2; b.dbg_reads[4 * r + 3] = t3; }
				atomicAdd((unsigned long long*)&b.dbg[0], (unsigned long long)(t1 - t0));
				atomicAdd((unsigned long long*)&b.dbg[1], (unsigned long long)(t2 - t1));
				atomicAdd((unsigned long long*)&b.dbg[2], (unsigned long long)(t3 - t2));
				atomicMax((unsigned long long*)&b.dbg[4], (unsigned long long)(t2 - t1));
				atomicMax((unsigned long long*)&b.dbg[5], (unsigned long long)(t3 - t2));
				atomicMax((unsigned long long*)&b.dbg[6], (unsigned long long)(t3 - t0));
				atomicAdd((unsigned long long*)&b.dbg[7], (unsigned long long)wd.load);
				atomicAdd((unsigned long long*)&b.dbg[8], (unsigned long long)wd.longt);
				atomicAdd((unsigned long long*)&b.dbg[9], (unsigned long long)wd.groups);
				atomicAdd((unsigned long long*)&b.dbg[10], (unsigned long long)wd.open);
				atomicAdd((unsigned long long*)&b.dbg[11], (unsigned long long)wd.nlong);
				atomicAdd((unsigned long long*)&b.dbg[12], (unsigned long long)n_z);
				atomicAdd((unsigned long long*)&b.dbg[20], (unsigned long long)wd.iters);
			}
		}
		wave_sync();
	}
}

__global__ __launch_bounds__(POST_THREADS, POST_WAVES_PER_SIMD) void k_post_chains(PostBatch b, int team_reads) { post_chains_body<false>(b, team_reads); }
__global__ __launch_bounds__(POST_THREADS, POST_WAVES_PER_SIMD) void k_post_sort(PostBatch b, int team_reads) { post_chains_body<true>(b, team_reads); }

// --------------------------------------------------------------------------------------------------------------
// The sort by levels (round 6).  A read's candidates are collected in index order (lchain.c:35-41) by the pass over its anchors that also gives
// them their classes (k_post_classes, below); a read of more than 64 candidates whose scores differ becomes the first task.  k_post_sort_level: one wave per task -- the radix pass of a run on its key byte (on the next byte down
// while all keys share it: such a pass moves nothing), then the run's buckets: those of more than 64 elements are the next level's tasks, the
// others are insertion-sorted on the spot (rs_sort's recursion, ksort.h:140-145: buckets are sorted independently of each other, so any order
// of the tasks gives the host's result).  Tasks are taken longest first.
// --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_post_stask_count(PostBatch b, int level)
{
	const int n_t = b.cursor[8 + level];
	const int4 *list = b.stask[level & 1];
	__shared__ int s_cnt[N_SIZE_CLASSES];
	for (int t0 = blockIdx.x * blockDim.x; t0 < n_t; t0 += gridDim.x * blockDim.x) {
		const int t = t0 + threadIdx.x;
		block_bin_count(b.size_bins, t < n_t ? size_class(list[t].z) : 0, t < n_t, s_cnt);
	}
}
__global__ __launch_bounds__(256) void k_post_stask_scatter(PostBatch b, int level)
{
	const int n_t = b.cursor[8 + level];
	const int4 *list = b.stask[level & 1];
	__shared__ int s_cnt[N_SIZE_CLASSES], s_base[N_SIZE_CLASSES];
	for (int t0 = blockIdx.x * blockDim.x; t0 < n_t; t0 += gridDim.x * blockDim.x) {
		const int t = t0 + threadIdx.x;
		const int at = block_bin_slot(b.size_bins + N_SIZE_CLASSES, t < n_t ? size_class(list[t].z) : 0, t < n_t, s_cnt, s_base);
		if (t < n_t) b.stask_order[at] = t;
	}
}

__global__ __launch_bounds__(POST_THREADS, POST_WAVES_PER_SIMD) void k_post_sort_level(PostBatch b, int level)
{
	__shared__ PassLds lds[POST_THREADS / W];
	PassLds &L = lds[threadIdx.x / W];
	const int l = lane();
	const int n_t = b.cursor[8 + level];
	const int4 *list = b.stask[level & 1];
	int4 *next = b.stask[(level + 1) & 1];
	const int lvl = level < 3 ? level : 2;
	const bool pairs = b.sort_pairs != 0;
	for (;;) {
		int q = 0;
		if (l == 0) q = atomicAdd(b.cursor + 16 + level, 1);
		q = uni(q);
		if (q >= n_t) break;
		int4 todo[2];
		todo[0] = list[uni(b.stask_order[q])];
		int n_todo = 1;
		if (pairs && uni(todo[0].z) <= PAIR_MAX) {
			// a short run: it is walked beside the next one when that is short too (the tasks come longest first, so it nearly always is)
			int q2 = 0;
			if (l == 0) q2 = atomicAdd(b.cursor + 16 + level, 1);
			q2 = uni(q2);
			if (q2 < n_t) {
				todo[1] = list[uni(b.stask_order[q2])];
				n_todo = 2;
				if (uni(todo[1].z) <= PAIR_MAX) {
					const long long t0 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
					PairRun R[2];
#pragma unroll
					for (int j = 0; j < 2; ++j) {
						const int64_t off = b.offsets[uni(todo[j].x)] + uni(todo[j].y);
						R[j].g = b.z + off; R[j].len = uni(todo[j].z); R[j].shift = uni(todo[j].w);
						R[j].sc = SortScratch{ b.sort_s + off, b.sort_perm + off, b.sort_tmp + off };
					}
					const int moved = radix_pass_pair(R[0], R[1], L, b.dbg ? b.dbg + 24 + 6 * lvl : nullptr);
					n_todo = 0;
					int d_small = 0;
					for (int j = 0; j < 2; ++j) {
						const int4 t = j == 0 ? todo[0] : todo[1];
						const int r = uni(t.x), lo = uni(t.y), len = uni(t.z), shift = uni(t.w) - 8;
						if (shift < 0) continue;
						if (!((moved >> j) & 1)) { const int4 again = make_int4(r, lo, len, shift); if (n_todo == 0) todo[0] = again; else todo[1] = again; ++n_todo; continue; }
						d_small += sort_level<ZElem>(j == 0 ? R[0].g : R[1].g, len, shift, [&](int first, int end) {
							if (l == 0) next[atomicAdd(b.cursor + 8 + level + 1, 1)] = make_int4(r, lo + first, end - first, shift);
						});
					}
					if (b.dbg && l == 0) {
						const long long t1 = (long long)__builtin_amdgcn_s_memrealtime();
						atomicAdd((unsigned long long*)&b.dbg[13 + lvl], (unsigned long long)(t1 - t0));
						atomicAdd((unsigned long long*)&b.dbg[1], (unsigned long long)(t1 - t0));
						atomicMax((unsigned long long*)&b.dbg[4], (unsigned long long)(t1 - t0));
						atomicAdd((unsigned long long*)&b.dbg[17], 2ull);
						atomicAdd((unsigned long long*)&b.dbg[18], (unsigned long long)(R[0].len + R[1].len));
						atomicAdd((unsigned long long*)&b.dbg[19], (unsigned long long)d_small);
						if (b.dbg_stasks) {
							const unsigned long long at = atomicAdd((unsigned long long*)&b.dbg[42], 1ull);
							if (at < 262144) { b.dbg_stasks[4 * at] = t0; b.dbg_stasks[4 * at + 1] = t1; b.dbg_stasks[4 * at + 2] = (long long)level << 32 | (R[0].len + R[1].len); b.dbg_stasks[4 * at + 3] = 0; }
						}
					}
					wave_sync();
				}
			}
		}
		for (int k = 0; k < n_todo; ++k) {
		const int4 t = k == 0 ? todo[0] : todo[1];
		const int r = uni(t.x), lo = uni(t.y), len = uni(t.z);
		const int64_t off = b.offsets[r];
		unsigned long long *g = b.z + off + lo;
		const SortScratch sc = { b.sort_s + off + lo, b.sort_perm + off + lo, b.sort_tmp + off + lo };
		const long long t0 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
		int shift = uni(t.w);
		bool moved = false;
		long long ph[6] = { 0, 0, 0, 0, 0, 0 };
		FewBuckets fb;
		for (; shift >= 0 && !moved; shift -= 8) { fb.n = 0; moved = one_radix_pass<ZElem, true>(g, 0, len, shift, L, &sc, b.dbg ? b.dbg + 24 + 6 * lvl : nullptr, b.dbg_stasks ? ph : nullptr, &fb); }
		// (shift is now one byte below the pass that moved the run)
		int d_small = 0;
		if (moved && shift >= 0 && fb.n > 0) {
			// the pass knew its (at most four) buckets: they are the run's children
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				if (j < fb.n) {
					const int first = fb.start[j], n_j = fb.end[j] - fb.start[j];
					if (n_j > SMALL_RUN) { if (l == 0) next[atomicAdd(b.cursor + 8 + level + 1, 1)] = make_int4(r, lo + first, n_j, shift); }
					else if (n_j > 1) small_run_sort<ZElem>(g, first, n_j);
				}
			}
		} else if (moved && shift >= 0)
			d_small = sort_level<ZElem>(g, len, shift, [&](int first, int end) {
				if (l == 0) next[atomicAdd(b.cursor + 8 + level + 1, 1)] = make_int4(r, lo + first, end - first, shift);
			});
		if (b.dbg && l == 0) {
			const long long t1 = (long long)__builtin_amdgcn_s_memrealtime();
			atomicAdd((unsigned long long*)&b.dbg[13 + lvl], (unsigned long long)(t1 - t0));
			atomicAdd((unsigned long long*)&b.dbg[1], (unsigned long long)(t1 - t0));
			atomicMax((unsigned long long*)&b.dbg[4], (unsigned long long)(t1 - t0));
			atomicAdd((unsigned long long*)&b.dbg[17], 1ull);
			atomicAdd((unsigned long long*)&b.dbg[18], (unsigned long long)len);
			atomicAdd((unsigned long long*)&b.dbg[19], (unsigned long long)d_small);
			if (b.dbg_stasks) {
				const unsigned long long at = atomicAdd((unsigned long long*)&b.dbg[42], 1ull);
				// [3]: ticks of the pass's phases, 12 bits each (9 of value, 3 of exponent: value << 2 * exponent): histogram | set-up | walk | the elements' move
				auto q12 = [](long long v) { int e = 0; while ((v >> (2 * e)) > 511 && e < 7) ++e; const long long m = v >> (2 * e); return (m > 511 ? 511LL : m) | (long long)e << 9; };
				if (at < 262144) { b.dbg_stasks[4 * at] = t0; b.dbg_stasks[4 * at + 1] = t1; b.dbg_stasks[4 * at + 2] = (long long)level << 32 | len;
				                   b.dbg_stasks[4 * at + 3] = q12(ph[0]) | q12(ph[1]) << 12 | q12(ph[2]) << 24 | q12(ph[3]) << 36 | (ph[5] & 0xfffff) << 48; }
			}
		}
		wave_sync();
		}
	}
}

// --------------------------------------------------------------------------------------------------------------
// The walks shared out by tree (round 6).  k_post_classes: every anchor's class -- a hash of the root of its tree -- by one pass over the read in
// index order (a predecessor always lies before its anchor): 64 anchors per step, predecessors inside the 64 resolved by pointer jumping
// between lanes, the most recent classes kept in LDS (most predecessors are near), older ones read back from memory.  k_post_partition: a read's
// sorted candidates dealt to their classes, order kept.  k_post_walk: one wave per (read, class).
// --------------------------------------------------------------------------------------------------------------
namespace {
constexpr int CLS_RING = 1024;                  // most recent classes kept in LDS, per wave
__device__ __forceinline__ int tree_class(int root) { return (int)(((unsigned)root * 0x9E3779B1u) >> 28); }
static_assert(N_TREE_CLASSES == 16, "tree_class takes the top four bits of the hash; tasks pack the class into four bits");
} // namespace

__global__ __launch_bounds__(POST_THREADS) void k_post_classes(PostBatch b)
{
	__shared__ int s_ring[POST_THREADS / W][CLS_RING / 4];       // the class ring (bytes); afterwards the histogram of the read's top pass (256 words)
	__shared__ int s_cnt[POST_THREADS / W][N_TREE_CLASSES];
	static_assert(CLS_RING / 4 >= 256, "the ring's memory doubles as the 256 counters of the few-bucket pass");
	const int l = lane(), w = uni(threadIdx.x / W);
	unsigned char *ring = (unsigned char*)s_ring[w];
	int *cnt = s_cnt[w];
	for (;;) {
		int q = 0;
		if (l == 0) q = atomicAdd(b.cursor + 2, 1);
		q = uni(q);
		if (q >= b.n_reads) break;
		const int r = uni(b.order[q]);
		const int64_t off = b.offsets[r];
		const int n = (int)(b.offsets[r + 1] - off);
		const int32_t *p = b.p + off, *f = b.f + off;
		unsigned char *cls = b.cls + off;
		// the same pass collects the read's candidates for the sort by levels (lchain.c:35-41: z[k] = (f[i], i) for f[i] >= min_sc, in index order)
		const bool collect = b.stask[0] != nullptr;
		unsigned long long *z = b.z + off;
		int n_z = 0;
		unsigned any = 0, all = ~0u;
		const long long t0 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
		if (l < N_TREE_CLASSES) cnt[l] = 0;
		wave_sync();
		// (the links of four blocks per round trip, asked for four blocks ahead: a block's classes wait for the block before it, its links need
		// not -- and a wait for loads that were issued BEFORE the last blocks' stores does not wait for those stores)
		int pn4[4], fn4[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) { pn4[u] = u * W + l < n ? p[u * W + l] : 0; fn4[u] = collect && u * W + l < n ? f[u * W + l] : INT_MIN; }
		for (int base0 = 0; base0 < n; base0 += 4 * W) {
		int pl4[4], fl4[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) { pl4[u] = pn4[u]; fl4[u] = fn4[u]; }
#pragma unroll
		for (int u = 0; u < 4; ++u) { const int i2 = base0 + (4 + u) * W + l; pn4[u] = i2 < n ? p[i2] : 0; fn4[u] = collect && i2 < n ? f[i2] : INT_MIN; }
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int base = base0 + u * W;
			if (base >= n) break;
			const int i = base + l;
			const bool in = i < n;
			const int pl = pl4[u];
			const int pred = i - pl;
			int c = tree_class(i);                                 // a root's own
			bool known = !in || pl == 0;
			const bool far = !known && pred < base - CLS_RING;       // (the ring holds the CLS_RING anchors before this block; it is read before the block's own classes go in)
			if (__ballot(far) != 0) {
				wave_sync();                                       // what this wave stored blocks ago has landed
				if (far) { c = cls[pred]; known = true; }
			}
			if (!known && pred < base) { c = ring[pred & (CLS_RING - 1)]; known = true; }
			// predecessors inside the block: pointer jumping between lanes (a chain of at most 64 links: six rounds)
			int ptr = known ? l : pred - base;
			for (int it = 0; it < 6; ++it) {
				if (__ballot(!known) == 0) break;
				const int c2 = __shfl(c, ptr), k2 = __shfl((int)known, ptr), p2 = __shfl(ptr, ptr);
				if (!known) { if (k2) { c = c2; known = true; } else ptr = p2; }
			}
			lds_sync();                                            // the ring's reads before its writes (the stores to memory drain behind: nothing near reads them back)
			if (in) { cls[i] = (unsigned char)c; ring[i & (CLS_RING - 1)] = (unsigned char)c; atomicAdd(&cnt[c], 1); }
			lds_sync();
			const int fi = fl4[u];
			const bool take = in && fi >= b.min_sc;                // (never when nothing is collected: fl4 is INT_MIN then)
			const unsigned long long m = __ballot(take);
			if (take) { any |= (unsigned)fi; all &= (unsigned)fi; z[n_z + __popcll(m & ((1ull << l) - 1))] = (unsigned long long)(unsigned)fi << 32 | (unsigned)i; }
			n_z += __popcll(m);
		}
		}
		wave_sync();
		if (l < N_TREE_CLASSES) b.cls_cnt[(int64_t)r * N_TREE_CLASSES + l] = cnt[l];
		if (collect) {
			for (int o = W / 2; o > 0; o >>= 1) { any |= __shfl_xor(any, o); all &= __shfl_xor(all, o); }
			if (l == 0) {
				b.read_nz[r] = n_z;
				if (b.dbg) { atomicAdd((unsigned long long*)&b.dbg[0], (unsigned long long)((long long)__builtin_amdgcn_s_memrealtime() - t0)); atomicAdd((unsigned long long*)&b.dbg[12], (unsigned long long)n_z); }
			}
			const unsigned diff = uni((int)(any ^ all));
			if (n_z > SMALL_RUN) {
				if (diff != 0) {
					int top = 24;                                  // of the key = the score: byte 3 .. 0 (sort_like_host's `top`)
					while (top > 0 && ((diff >> top) & 255u) == 0) top -= 8;
					// A long read's top pass right here: its scores take two to four values of their highest byte that differs, and such a pass
					// needs nothing of the sort's LDS but a histogram (radix_pass_bytes_t<true>: positions in scalar registers, bytes across the
					// lanes) -- the wave that collected the candidates has them in its caches, and the pass does not wait for the whole batch's
					// classes.  Its buckets are tasks of the SECOND level; a run that is not of that kind becomes a first-level task as before.
					int status = 2, shift = top;
					FewBuckets fb;
					if (n_z > LINE_STORE_BYTES) {
						wave_sync();
						const SortScratch sc = { b.sort_s + off, b.sort_perm + off, b.sort_tmp + off };
						for (status = 0; shift >= 0 && status == 0; shift -= 8) { fb.n = 0; status = radix_pass_bytes_t<true>(z, 0, n_z, shift, nullptr, s_ring[w], sc, b.dbg ? b.dbg + 24 : nullptr, nullptr, &fb); }
						// (shift is one byte below the pass that ran last)
					}
					if (status == 1) {
						if (shift >= 0) {
#pragma unroll
							for (int j = 0; j < 4; ++j) {
								if (j < fb.n) {
									const int first = fb.start[j], n_j = fb.end[j] - fb.start[j];
									if (n_j > SMALL_RUN) { if (l == 0) b.stask[1][atomicAdd(b.cursor + 9, 1)] = make_int4(r, first, n_j, shift); }
									else if (n_j > 1) small_run_sort<ZElem>(z, first, n_j);
								}
							}
						}
					} else if (status == 2) { if (l == 0) b.stask[0][atomicAdd(b.cursor + 8, 1)] = make_int4(r, 0, n_z, n_z > LINE_STORE_BYTES ? shift + 8 : top); }
				}
			} else if (n_z > 1) small_run_sort<ZElem>(z, 0, n_z);
		}
		wave_sync();
	}
}

__global__ __launch_bounds__(POST_THREADS) void k_post_partition(PostBatch b)
{
	const int l = lane();
	for (;;) {
		int q = 0;
		if (l == 0) q = atomicAdd(b.cursor + 3, 1);
		q = uni(q);
		if (q >= b.n_reads) break;
		const int r = uni(b.order[q]);
		const int64_t off = b.offsets[r];
		const int n_z = b.read_nz[r];
		const unsigned long long *z = b.z + off;
		const unsigned char *cls = b.cls + off;
		unsigned long long *zc = b.zc + off;
		int32_t *kpos = b.kpos + off, *endslot = (int32_t*)(b.z + off);     // (endslot takes the place of the read's own z: n entries of 4 bytes in its 8 n bytes)
		// a class's candidates go where its anchors' share of the read's slots begins (candidates of a class <= anchors of a class)
		int inc = l < N_TREE_CLASSES ? b.cls_cnt[(int64_t)r * N_TREE_CLASSES + l] : 0;
		const int own = inc;
		for (int o = 1; o < N_TREE_CLASSES; o <<= 1) { const int v = __shfl_up(inc, o); if (l >= o) inc += v; }
		const int sub_l = inc - own;
		int sub[N_TREE_CLASSES], cur[N_TREE_CLASSES];
#pragma unroll
		for (int c = 0; c < N_TREE_CLASSES; ++c) { sub[c] = __builtin_amdgcn_readlane(sub_l, c); cur[c] = 0; }
		for (int base = 0; base < n_z; base += 4 * W) {
			// four blocks of 64 per round trip (one wave streams a read: every round trip is paid in full)
			unsigned long long e4[4];
			int c4[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) e4[u] = z[min(base + u * W + l, n_z - 1)];
#pragma unroll
			for (int u = 0; u < 4; ++u) c4[u] = base + u * W + l < n_z ? (int)cls[(int)(unsigned)e4[u]] : -1;
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const int k = base + u * W + l;
				const unsigned long long e = e4[u];
				const int c = c4[u];
#pragma unroll
				for (int cc = 0; cc < N_TREE_CLASSES; ++cc) {
					const unsigned long long m = __ballot(c == cc);
					if (c == cc) { const int at = sub[cc] + cur[cc] + __popcll(m & ((1ull << l) - 1)); zc[at] = e; kpos[at] = k; }
					cur[cc] += __popcll(m);
				}
			}
			// (endslot lives in z's memory: entries [base, base + 256) lie inside z[0, base / 2 + 128), which has been read)
#pragma unroll
			for (int u = 0; u < 4; ++u) if (base + u * W + l < n_z) endslot[base + u * W + l] = -1;
		}
		int mine = 0;
#pragma unroll
		for (int c = 0; c < N_TREE_CLASSES; ++c) if (l == c) mine = cur[c];
		if (l < N_TREE_CLASSES) {
			b.cls_nz[(int64_t)r * N_TREE_CLASSES + l] = mine;
			if (mine > 0) b.wtask[atomicAdd(b.cursor + 5, 1)] = r << 4 | l;
		}
	}
}

// walk tasks by number of candidates, most first (the same classes of sizes as the reads')
__global__ __launch_bounds__(256) void k_post_task_count(PostBatch b)
{
	const int n_t = b.cursor[5];
	__shared__ int s_cnt[N_SIZE_CLASSES];
	for (int t0 = blockIdx.x * blockDim.x; t0 < n_t; t0 += gridDim.x * blockDim.x) {
		const int t = t0 + threadIdx.x;
		const int task = t < n_t ? b.wtask[t] : 0;
		block_bin_count(b.size_bins, t < n_t ? size_class(b.cls_nz[(int64_t)(task >> 4) * N_TREE_CLASSES + (task & 15)]) : 0, t < n_t, s_cnt);
	}
}
__global__ __launch_bounds__(256) void k_post_task_scatter(PostBatch b)
{
	const int n_t = b.cursor[5];
	__shared__ int s_cnt[N_SIZE_CLASSES], s_base[N_SIZE_CLASSES];
	for (int t0 = blockIdx.x * blockDim.x; t0 < n_t; t0 += gridDim.x * blockDim.x) {
		const int t = t0 + threadIdx.x;
		const int task = t < n_t ? b.wtask[t] : 0;
		const int at = block_bin_slot(b.size_bins + N_SIZE_CLASSES, t < n_t ? size_class(b.cls_nz[(int64_t)(task >> 4) * N_TREE_CLASSES + (task & 15)]) : 0, t < n_t, s_cnt, s_base);
		if (t < n_t) b.wtask_order[at] = task;
	}
}

#ifndef MM2GB_WALK_WAVES_PER_SIMD
#define MM2GB_WALK_WAVES_PER_SIMD 4
#endif
__global__ __launch_bounds__(POST_THREADS, MM2GB_WALK_WAVES_PER_SIMD) void k_post_walk(PostBatch b)
{
	const int l = lane();
	const int mc = b.min_cnt > 1 ? b.min_cnt : 1;
	const int n_t = b.cursor[5];
	for (;;) {
		int q = 0;
		if (l == 0) q = atomicAdd(b.cursor + 4, 1);
		q = uni(q);
		if (q >= n_t) break;
		const int task = uni(b.wtask_order[q]);
		const int r = task >> 4, c = task & 15;
		const int64_t off = b.offsets[r];
		// this class's share of the read's slots
		const int cc = l < N_TREE_CLASSES ? b.cls_cnt[(int64_t)r * N_TREE_CLASSES + l] : 0;
		int inc = cc, uinc = cc / mc;
		for (int o = 1; o < N_TREE_CLASSES; o <<= 1) { const int v = __shfl_up(inc, o), uv = __shfl_up(uinc, o); if (l >= o) { inc += v; uinc += uv; } }
		const int sub = __builtin_amdgcn_readlane(inc - cc, c), usub = __builtin_amdgcn_readlane(uinc - cc / mc, c);
		const int n_zc = b.cls_nz[(int64_t)r * N_TREE_CLASSES + c];
		const int64_t slot0 = chain_slot(off, r, mc);
		const long long t0 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
		WalkSplit what;
		what.kp = b.kpos + off + sub; what.endslot = (int32_t*)(b.z + off); what.u_loc = b.u_loc + slot0 + usub; what.picked_base = sub; what.slot_base = usub;
		int n_u = 0, n_v = 0;
		WalkDbg wd;
		post_walk_read(b, off, n_zc, b.zc + off + sub, b.fp + off, b.picked + off + sub, b.u_tmp + slot0 + usub, n_u, n_v, wd, nullptr, what);
		wave_sync();
		if (l == 0) {
			if (n_u) atomicAdd(&b.n_u[r], n_u);
			if (n_v) atomicAdd(&b.n_kept[r], n_v);
			if (b.dbg) {
				const long long t3 = (long long)__builtin_amdgcn_s_memrealtime();
				atomicAdd((unsigned long long*)&b.dbg[2], (unsigned long long)(t3 - t0));
				atomicMax((unsigned long long*)&b.dbg[5], (unsigned long long)(t3 - t0));
				atomicAdd((unsigned long long*)&b.dbg[7], (unsigned long long)wd.load);
				atomicAdd((unsigned long long*)&b.dbg[8], (unsigned long long)wd.longt);
				atomicAdd((unsigned long long*)&b.dbg[9], (unsigned long long)wd.groups);
				atomicAdd((unsigned long long*)&b.dbg[10], (unsigned long long)wd.open);
				atomicAdd((unsigned long long*)&b.dbg[11], (unsigned long long)wd.nlong);
				atomicAdd((unsigned long long*)&b.dbg[20], (unsigned long long)wd.iters);
				atomicAdd((unsigned long long*)&b.dbg[21], 1ull);
				atomicMin((unsigned long long*)&b.dbg[22], (unsigned long long)t0);      // first task's start (the slot is set to ~0 before the launch)
				atomicMax((unsigned long long*)&b.dbg[23], (unsigned long long)t3);
				if (b.dbg_tasks) { b.dbg_tasks[8 * q] = t0; b.dbg_tasks[8 * q + 1] = t3; b.dbg_tasks[8 * q + 2] = task; b.dbg_tasks[8 * q + 3] = n_zc; b.dbg_tasks[8 * q + 4] = wd.load; b.dbg_tasks[8 * q + 5] = wd.longt; b.dbg_tasks[8 * q + 6] = wd.nlong; b.dbg_tasks[8 * q + 7] = wd.iters; }
			}
		}
		wave_sync();
	}
}

// --------------------------------------------------------------------------------------------------------------
// offsets of every read's chains / kept anchors in the compacted outputs
// --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_post_scan(PostBatch b)
{
	__shared__ long long s_tmp[2][1024 / W];
	long long carry_u = 0, carry_a = 0;
	const int w = threadIdx.x / W, l = lane();
	for (int64_t base = 0; base < b.n_reads; base += 1024) {
		const int64_t r = base + threadIdx.x;
		const long long vu = r < b.n_reads ? b.n_u[r] : 0, va = r < b.n_reads ? b.n_kept[r] : 0;
		long long iu = vu, ia = va;
		for (int off = 1; off < W; off <<= 1) {
			const long long ou = __shfl_up(iu, off), oa = __shfl_up(ia, off);
			if (l >= off) { iu += ou; ia += oa; }
		}
		__syncthreads();
		if (l == W - 1) { s_tmp[0][w] = iu; s_tmp[1][w] = ia; }
		__syncthreads();
		long long bu = 0, ba = 0, tu = 0, ta = 0;
		for (int k = 0; k < 1024 / W; ++k) { if (k < w) { bu += s_tmp[0][k]; ba += s_tmp[1][k]; } tu += s_tmp[0][k]; ta += s_tmp[1][k]; }
		if (r < b.n_reads) { b.u_off[r] = carry_u + bu + iu - vu; b.a_off[r] = carry_a + ba + ia - va; }
		carry_u += tu; carry_a += ta;
	}
	if (threadIdx.x == 0) { b.u_off[b.n_reads] = carry_u; b.a_off[b.n_reads] = carry_a; b.totals[0] = carry_u; b.totals[1] = carry_a; }
}

// --------------------------------------------------------------------------------------------------------------
// per read: order of compaction (lchain.c:84-110) and the gather into the final arrays
// --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(POST_THREADS) void k_post_emit(PostBatch b)
{
	__shared__ PassLds lds[POST_THREADS / W];
	PassLds &L = lds[threadIdx.x / W];
	const int l = lane();
	const int mc = b.min_cnt > 1 ? b.min_cnt : 1;
	for (;;) {
		int r = 0;
		if (l == 0) r = atomicAdd(b.cursor + 1, 1);
		r = uni(r);
		if (r >= b.n_reads) break;
		r = uni(b.order[r]);
		const int n_u = b.n_u[r];
		if (n_u == 0) continue;
		const long long t0 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
		const int64_t off = b.offsets[r];
		const uint4 *raw = b.raw + off;
		const int32_t *picked = b.picked + off;
		const unsigned long long *u_tmp = b.u_tmp + chain_slot(off, r, mc);
		ulonglong2 *heads = b.heads + chain_slot(off, r, mc);
		unsigned long long *u_out = b.u_out + b.u_off[r];
		uint4 *a_out = b.a_out + b.a_off[r];
		// (x of the chain's first anchor, offset << 32 | chain): the first anchor is the last one picked (lchain.c:88-99)
		int k_at = 0;
		if (b.cls) {
			// split form: the chains in the order the host finds them = the candidates from the best one down; endslot[] says which of them ended a
			// chain that was kept, and where k_post_walk left it (its slot in u_tmp / u_loc; u_loc: its anchors in picked)
			const int32_t *endslot = (const int32_t*)(b.z + off);
			const int32_t *u_loc = b.u_loc + chain_slot(off, r, mc);
			const int n_z = b.read_nz[r];
			for (int kb = n_z - 1; kb >= 0; kb -= 4 * W) {
				int slot[4];
#pragma unroll
				for (int q = 0; q < 4; ++q) { const int k = kb - q * W - l; slot[q] = k >= 0 ? endslot[k] : -1; }
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const unsigned long long m = __ballot(slot[q] >= 0);
					if (slot[q] >= 0) {
						const int c = k_at + __popcll(m & ((1ull << l) - 1));
						const int cnt = (int)(unsigned)u_tmp[slot[q]], k0 = u_loc[slot[q]];
						const uint4 first = raw[picked[k0 + cnt - 1]];
						heads[c] = make_ulonglong2((unsigned long long)first.y << 32 | first.x, (unsigned long long)(unsigned)k0 << 32 | (unsigned)slot[q]);
					}
					k_at += __popcll(m);
				}
			}
		} else
		for (int base = 0; base < n_u; base += W) {
			const int c = base + l;
			const int cnt = c < n_u ? (int)(unsigned)u_tmp[c] : 0;
			int inc = cnt;
			for (int o = 1; o < W; o <<= 1) { const int v = __shfl_up(inc, o); if (l >= o) inc += v; }
			const int k0 = k_at + inc - cnt;
			if (c < n_u) {
				const uint4 first = raw[picked[k0 + cnt - 1]];
				heads[c] = make_ulonglong2((unsigned long long)first.y << 32 | first.x, (unsigned long long)(unsigned)k0 << 32 | (unsigned)c);
			}
			k_at += __builtin_amdgcn_readlane(inc, W - 1);
		}
		wave_sync();
		sort_like_host<HElem>(heads, n_u, L);
		// chains in that order; each chain's anchors from its first to its last (picked[] holds them last to first).  64 chains at a time: their
		// records and counts with all lanes (one round trip for the lot, not one per chain); chains of 64 anchors or more are then copied by the
		// whole wave, four blocks of 64 in flight; the others -- a repeat's chains of a dozen anchors, hundreds per read -- share the wave: output
		// positions are dealt to the lanes in order, a lane finds its chain by bisection over the lanes' running counts
		wave_sync();
		int out_at = 0;
		for (int cb = 0; cb < n_u; cb += W) {
			const int c = cb + l;
			const bool in = c < n_u;
			const ulonglong2 h = heads[in ? c : n_u - 1];
			const int k0 = (int)(h.y >> 32), ci = (int)(unsigned)h.y;
			const unsigned long long u = u_tmp[ci];
			const int cnt = in ? (int)(unsigned)u : 0, small = cnt < W ? cnt : 0;
			if (in) u_out[c] = u;
			int inc = cnt, sinc = small;
			for (int o = 1; o < W; o <<= 1) { const int v = __shfl_up(inc, o), sv = __shfl_up(sinc, o); if (l >= o) { inc += v; sinc += sv; } }
			const int at = out_at + inc - cnt;                       // where this lane's chain goes
			// the long ones
			unsigned long long big = __ballot(cnt >= W);
			while (((unsigned)big | (unsigned)(big >> 32)) != 0) {
				const int src = first_set(big);
				big &= big - 1;
				const int bk = __builtin_amdgcn_readlane(k0, src), bc = __builtin_amdgcn_readlane(cnt, src), ba = __builtin_amdgcn_readlane(at, src);
				for (int j0 = 0; j0 < bc; j0 += 4 * W) {
					int from[4];
					uint4 a4[4];
#pragma unroll
					for (int q = 0; q < 4; ++q) { const int j = j0 + q * W + l; from[q] = picked[bk + (bc - 1 - min(j, bc - 1))]; }
#pragma unroll
					for (int q = 0; q < 4; ++q) a4[q] = raw[from[q]];
#pragma unroll
					for (int q = 0; q < 4; ++q) { const int j = j0 + q * W + l; if (j < bc) a_out[ba + j] = a4[q]; }
				}
			}
			// the short ones
			const int total_small = __builtin_amdgcn_readlane(sinc, W - 1);
			for (int p0 = 0; p0 < total_small; p0 += W) {
				const int pp = p0 + l;
				int lo = 0, hi = W - 1;                               // first lane whose running count of short-chain anchors exceeds pp
#pragma unroll
				for (int it = 0; it < 6; ++it) { const int mid = (lo + hi) >> 1; const int v = __shfl(sinc, mid); if (v <= pp) lo = mid + 1; else hi = mid; }
				const int sk = __shfl(k0, lo), sc2 = __shfl(small, lo), sa = __shfl(at, lo), before = __shfl(sinc, lo) - sc2;
				if (pp < total_small) { const int j = pp - before; a_out[sa + j] = raw[picked[sk + (sc2 - 1 - j)]]; }
			}
			out_at += __builtin_amdgcn_readlane(inc, W - 1);
		}
		if (b.dbg && l == 0) atomicAdd((unsigned long long*)&b.dbg[3], (unsigned long long)((long long)__builtin_amdgcn_s_memrealtime() - t0));
	}
}

// --------------------------------------------------------------------------------------------------------------
// RMQ re-chaining: the score fill of mg_lchain_rmq (lchain.c:273-350), one wave per read.
// The reference keeps the anchors in reach in two balanced trees ordered by (query position, index) and asks them for the
// element of minimal pri = -(f + 0.5 * gap * (x + y)) in a query-position range (krmq.h).  What the trees hold when anchor i
// is processed are index ranges -- [st, i0) and [st_inner, i0), with i0 the first anchor sharing a[i].x and st / st_inner
// advanced by the eviction loops of lchain.c:293-310 -- so the same definition is evaluated here by the wave's 64 lanes
// scanning those ranges: an arg-max of the (double) key over the outer range, then, unless that pair is an exact diagonal
// extension, the best-scoring pair over the inner range in descending (y, index) order (lchain.c:320-341 with
// max_chn_skip = infinity, as everywhere on the GPU path).
// One thing is not a function of the definition: WHICH element krmq_rmq returns when several in-range elements share the
// minimal pri depends on the shape of the reference's AVL tree (krmq.h:110-147).  Such anchors are counted in n_tied[read];
// the host side does not use the device's answer for a read with n_tied != 0 (stream_api.cpp: mm2gb_lchain_rmq).
// --------------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float rmq_log2(float v)       // mg_log2, mmpriv.h:118-126
{
	unsigned u = __float_as_uint(v);
	float r = (float)(((u >> 23) & 255u) - 128u);
	u = (u & ~(255u << 23)) + (127u << 23);
	const float m = __uint_as_float(u);
	r += (-0.34484843f * m + 2.02466578f) * m - 0.67487759f;
	return r;
}

// comput_sc_simple, lchain.c:232-248
__device__ __forceinline__ int rmq_pair_score(unsigned xi, int yi, unsigned xj, int yj, int q_span_j, const RmqParams &P, bool &exact, int &width)
{
	const int dq = yi - yj, dr = (int)(xi - xj);
	const int dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq;
	int sc = q_span_j < dg ? q_span_j : dg;
	width = dd;
	exact = dd == 0 && dg <= q_span_j;
	if (dd != 0 || dq > q_span_j) {
		const float lin = P.pen_gap * (float)dd + P.pen_skip * (float)dg;
		const float lg = dd >= 1 ? rmq_log2((float)(dd + 1)) : 0.0f;
		sc -= (int)(lin + .5f * lg);
	}
	return sc;
}

// number of leading lanes (from lane 0) whose bit is CLEAR in `stay`, 64 if none is set.  Written with 32-bit halves on purpose:
// with __builtin_ctzll the loop test on the result was compiled to a 64-bit scalar compare and the kernel hung (ROCm 7.2).
__device__ __forceinline__ int leading_out(unsigned long long stay)
{
	const unsigned lo = (unsigned)stay, hi = (unsigned)(stay >> 32);
	return lo ? __builtin_ctz(lo) : hi ? 32 + __builtin_ctz(hi) : W;
}

// ordered image of a double: signed 64-bit comparison of the images orders the values (no NaN here)
__device__ __forceinline__ long long key_order(double k)
{
	const long long bits = __double_as_longlong(k);
	return bits ^ ((bits >> 63) & 0x7fffffffffffffffLL);
}

// maximum over the wave, the same in every lane: rotations inside the four rows of 16 lanes by DPP, then the rows' values through
// scalar registers.  Every lane must be active.
__device__ __forceinline__ int wave_max_i32(int v)
{
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false));   // row_ror:1
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false));   // row_ror:2
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));   // row_ror:4
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));   // row_ror:8
	return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_sum_i32(int v) { for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) { return (unsigned)wave_max_i32((int)(v ^ 0x80000000u)) ^ 0x80000000u; }
__device__ __forceinline__ long long wave_max_i64(long long v)
{
	const int hi = wave_max_i32((int)(v >> 32));
	const unsigned lo = wave_max_u32((int)(v >> 32) == hi ? (unsigned)v : 0u);
	return (long long)((unsigned long long)(unsigned)hi << 32 | lo);
}

constexpr long long RMQ_NONE = LLONG_MIN;   // key of a slot nothing has been put in
constexpr int RMQ_RING = 256;               // per wave: the most recent blocks' score bounds kept in LDS

// a candidate of the range-minimum query.  rank: position in the read's (y, index) order -- among equal keys the stated rule takes the
// largest.  check: >= 0 when the candidate stands for a block of 64 ranks whose summary says "several anchors share this key"; the
// summary may say so after one of them has left the window, so the block is looked at before the tie is believed.
struct RmqCand { long long ord; int j, rank, tie, check; };
__device__ __forceinline__ void rmq_offer(RmqCand &c, bool in, long long ord, int j, int rank, int tie, int check)
{
	if (!in) return;
	if (ord > c.ord) { c.ord = ord; c.j = j; c.rank = rank; c.tie = tie; c.check = check; }
	else if (ord == c.ord) {
		if (rank > c.rank) { c.j = j; c.rank = rank; }
		c.tie = 1; c.check = -1;
	}
}

// the largest key among the anchors of ranks [64 * blk, 64 * blk + 64) that are in the tree now (put in, index >= st), as a summary:
// (key low, key high, holder's index | several holders << 31, holder's rank & 63); written back, and returned in every lane
__device__ __forceinline__ uint4 rmq_block_summary(int blk, int n, int st, const long long *key, const int32_t *ord_idx, uint4 *l1)
{
	const int rk = (blk << 6) + lane();
	long long k = RMQ_NONE;
	int id = -1;
	if (rk < n) { k = key[rk]; id = ord_idx[rk]; }
	const bool act = k != RMQ_NONE && id >= st;
	const long long top = wave_max_i64(act ? k : RMQ_NONE);
	uint4 e = make_uint4((unsigned)top, (unsigned)((unsigned long long)top >> 32), 0x7fffffffu, 0u);
	if (top != RMQ_NONE) {
		const unsigned long long win = __ballot(act && k == top);
		const int who = 63 - first_set_from_top(win);
		e.z = (unsigned)__builtin_amdgcn_readlane(id, who) | (__popcll(win) > 1 ? 0x80000000u : 0u);
		e.w = (unsigned)who;
	}
	if (lane() == 0) l1[blk] = e;
	return e;
}

// ---- the inner walk with a skip limit (lchain.c:328-341), 64 candidates at a time in walking order (lane 0 first) ----
// Its scans over the lanes go by DPP inside the rows of 16 lanes (a shift is a vector instruction, not a trip through the LDS crossbar -- a round of
// the walk had ~30 of those behind each other) and by three scalar reads of the rows' last lanes across them.
template <int CTRL> __device__ __forceinline__ int dpp_or(int v, int otherwise) { return __builtin_amdgcn_update_dpp(otherwise, v, CTRL, 0xf, 0xf, false); }   // lanes without a source keep `otherwise`
constexpr int DPP_ROW_SHR = 0x110, DPP_WAVE_SHR1 = 0x138;
// largest value among the lanes BELOW this one (INT_MIN for lane 0)
__device__ __forceinline__ int wave_max_below(int v)
{
	v = max(v, dpp_or<DPP_ROW_SHR + 1>(v, INT_MIN)); v = max(v, dpp_or<DPP_ROW_SHR + 2>(v, INT_MIN));
	v = max(v, dpp_or<DPP_ROW_SHR + 4>(v, INT_MIN)); v = max(v, dpp_or<DPP_ROW_SHR + 8>(v, INT_MIN));
	const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = max(t0, __builtin_amdgcn_readlane(v, 31)), t2 = max(t1, __builtin_amdgcn_readlane(v, 47));
	const int row = lane() >> 4;
	v = max(v, row == 0 ? INT_MIN : row == 1 ? t0 : row == 2 ? t1 : t2);
	return dpp_or<DPP_WAVE_SHR1>(v, INT_MIN);
}
__device__ __forceinline__ unsigned wave_or_u32(unsigned x)
{
	int v = (int)x;
	v |= dpp_or<DPP_ROW_SHR + 1>(v, 0); v |= dpp_or<DPP_ROW_SHR + 2>(v, 0); v |= dpp_or<DPP_ROW_SHR + 4>(v, 0); v |= dpp_or<DPP_ROW_SHR + 8>(v, 0);
	return (unsigned)(__builtin_amdgcn_readlane(v, 15) | __builtin_amdgcn_readlane(v, 31) | __builtin_amdgcn_readlane(v, 47) | __builtin_amdgcn_readlane(v, 63));
}
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) { return (unsigned long long)wave_or_u32((unsigned)(v >> 32)) << 32 | wave_or_u32((unsigned)v); }
// The skip counter is a chain of x -> max(x - 1, 0) (a better score was met), x -> x + 1 (a candidate whose chain had been offered) and
// x -> x: all of the form x -> max(x + a, b), closed under composition -- (a1, b1) then (a2, b2) is (a1 + a2, max(b1 + a2, b2)) --, so the
// counter after every lane comes from a prefix scan of the lanes' (a, b).  NONE stands for "no lower bound" (far below any count).
constexpr int SKIP_NONE = INT_MIN / 4;
__device__ __forceinline__ int wave_skip_counts(int a, int bnd, int before)
{
	// (the earlier lanes' step first, then this one's)
#define MM2GB_SKIP_STEP(N) { const int oa = dpp_or<DPP_ROW_SHR + N>(a, 0), ob = dpp_or<DPP_ROW_SHR + N>(bnd, SKIP_NONE); bnd = max(ob + a, bnd); a = oa + a; }
	MM2GB_SKIP_STEP(1) MM2GB_SKIP_STEP(2) MM2GB_SKIP_STEP(4) MM2GB_SKIP_STEP(8)
#undef MM2GB_SKIP_STEP
	// the rows before this lane's: their last lanes' (a, b), composed in order
	const int a0 = __builtin_amdgcn_readlane(a, 15), b0 = __builtin_amdgcn_readlane(bnd, 15);
	const int ra1 = __builtin_amdgcn_readlane(a, 31), rb1 = __builtin_amdgcn_readlane(bnd, 31);
	const int ra2 = __builtin_amdgcn_readlane(a, 47), rb2 = __builtin_amdgcn_readlane(bnd, 47);
	const int a1 = a0 + ra1, b1 = max(b0 + ra1, rb1);               // rows 0 and 1
	const int a2 = a1 + ra2, b2 = max(b1 + ra2, rb2);               // rows 0 .. 2
	const int row = lane() >> 4;
	const int pa = row == 0 ? 0 : row == 1 ? a0 : row == 2 ? a1 : a2, pb = row == 0 ? SKIP_NONE : row == 1 ? b0 : row == 2 ? b1 : b2;
	bnd = max(pb + a, bnd); a = pa + a;
	return max(before + a, max(bnd, SKIP_NONE));
}

} // namespace

// ---- preparation, one thread per anchor: the (y, index) order of every read, and what each anchor's query looks like in it ----
__device__ __forceinline__ int64_t rmq_read_of(const int64_t *offsets, int64_t n_reads, int64_t g)
{
	int64_t lo = 0, hi = n_reads;                      // last r with offsets[r] <= g
	while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (offsets[mid] <= g) lo = mid; else hi = mid; }
	return lo;
}
__global__ __launch_bounds__(256) void k_rmq_prep_keys(RmqBatch b)
{
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < b.n; g += (int64_t)gridDim.x * blockDim.x) {
		const int64_t r = rmq_read_of(b.offsets, b.n_reads, g);
		const unsigned long long k = (unsigned long long)b.raw[g].z << 32 | (unsigned long long)(g - b.offsets[r]);
		if (b.skey_in) b.skey_in[g] = k; else b.by_y[g] = make_ulonglong2(k, 0ULL);
		((long long*)b.key)[g] = RMQ_NONE;
	}
	const int64_t nb = (b.n >> 6) + b.n_reads + 1;
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < nb; g += (int64_t)gridDim.x * blockDim.x)
		b.l1[g] = make_uint4(0u, 0x80000000u, 0x7fffffffu, 0u);   // RMQ_NONE, nobody
}
__global__ __launch_bounds__(256) void k_rmq_keys_to_by_y(RmqBatch b)
{
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < b.n; g += (int64_t)gridDim.x * blockDim.x) b.by_y[g] = make_ulonglong2(b.skey[g], 0ULL);
}
__global__ __launch_bounds__(256) void k_rmq_prep_ranks(RmqBatch b)
{
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < b.n; g += (int64_t)gridDim.x * blockDim.x) {
		const int64_t off = b.offsets[rmq_read_of(b.offsets, b.n_reads, g)];
		const int idx = (int)(unsigned)b.by_y[g].x;
		b.ord_idx[g] = idx;
		b.meta[off + idx].x = (int)(g - off);
	}
}
__global__ __launch_bounds__(256) void k_rmq_prep_ranges(RmqBatch b, int max_dist)
{
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < b.n; g += (int64_t)gridDim.x * blockDim.x) {
		const int64_t r = rmq_read_of(b.offsets, b.n_reads, g);
		const int64_t off = b.offsets[r];
		const int n = (int)(b.offsets[r + 1] - off);
		const ulonglong2 *z = b.by_y + off;
		const int yi = (int)b.raw[g].z;
		auto first_not_below = [&](long long y) {          // first rank whose y is >= y
			int lo = 0, hi = n;
			while (lo < hi) { const int mid = (lo + hi) >> 1; if ((long long)(z[mid].x >> 32) < y) lo = mid + 1; else hi = mid; }
			return lo;
		};
		// closed interval [(yi - max_dist, INT32_MAX), (yi, 0)] of (y, index) (lchain.c:311-313): y in (yi - max_dist, yi), plus anchor 0 when its y is yi
		b.meta[g].y = first_not_below((long long)yi - max_dist + 1);
		b.meta[g].z = first_not_below((long long)yi) - 1 + ((int)b.raw[off].z == yi ? 1 : 0);
	}
}

// with a skip limit: the anchors by rank, and every anchor's inner walk as a range of ranks (k_rmq_fill)
__global__ __launch_bounds__(256) void k_rmq_prep_skip(RmqBatch b, int max_inner)
{
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < b.n; g += (int64_t)gridDim.x * blockDim.x) {
		const int64_t r = rmq_read_of(b.offsets, b.n_reads, g);
		const int64_t off = b.offsets[r];
		const int n = (int)(b.offsets[r + 1] - off);
		const ulonglong2 *z = b.by_y + off;
		{	// g as a rank of read r
			const int idx = (int)(unsigned)b.by_y[g].x;
			const uint4 e = b.raw[off + idx];
			b.rk_a[g] = make_uint4(e.x, e.z, e.w & 0xffu, (unsigned)idx);
			b.rk_f[g] = 0; b.rk_p[g] = -1; b.rk_mark[g] = -1;
		}
		{	// g as an anchor of read r
			const int yi = (int)b.raw[g].z;
			auto first_not_below = [&](long long y) {
				int lo = 0, hi = n;
				while (lo < hi) { const int mid = (lo + hi) >> 1; if ((long long)(z[mid].x >> 32) < y) lo = mid + 1; else hi = mid; }
				return lo;
			};
			b.rk_in[g] = make_int2(first_not_below((long long)yi - max_inner), first_not_below((long long)yi) - 1);
		}
	}
}

// One WAVE per read, one anchor per step.  The reference's tree is ordered by (y, index) and answers "smallest priority with y in a
// range"; here every read's anchors are ranked in that order beforehand, key[] holds the (negated) priorities by RANK -- set when an
// anchor enters the tree -- and l1[] the largest key of every 64 consecutive ranks with its holder.  A query reads the two blocks of
// ranks at the ends of its interval anchor by anchor and up to 64 block summaries per instruction in between.  Nothing is ever taken
// out: an anchor that left the window is recognised by its index (< st), and a summary whose holder has left is rebuilt when a query
// meets it.  Anchors enter one per step, some steps after they were scored; until then they are candidates straight from the
// registers that hold the current block of 64 anchors and the one before it (one anchor per lane), so that no step waits for its
// predecessor's stores.  The inner window's exhaustive scan (lchain.c:320-341) goes over indices, block by block, and skips a
// block whose largest f + span cannot beat the score already found (comput_sc_simple never returns more than the span).
__global__ __launch_bounds__(POST_THREADS) void k_rmq_fill(RmqBatch b, RmqParams P)
{
	__shared__ int ring[POST_THREADS / W][RMQ_RING];       // largest f + span of the most recent finished blocks (by index)
	const int l = lane(), w = uni(threadIdx.x / W);
	const int max_dist = P.max_dist < P.bw ? P.bw : P.max_dist;                                       // lchain.c:264
	const int max_inner = (P.max_dist_inner <= 0 || P.max_dist_inner >= max_dist) ? 0 : P.max_dist_inner;   // lchain.c:265
	const double half_gap = 0.5 * (double)P.pen_gap;
	for (;;) {
		int r = 0;
		if (l == 0) r = atomicAdd(b.cursor, 1);
		r = uni(r);
		if (r >= b.n_reads) break;
		const int64_t off = b.offsets[r];
		const int n = (int)(b.offsets[r + 1] - off);
		const uint4 *a = b.raw + off;
		const int4 *meta = b.meta + off;
		const int32_t *ord_idx = b.ord_idx + off;
		int32_t *f = b.f + off, *p = b.p + off;
		long long *key = (long long*)b.key + off;
		uint4 *l1 = b.l1 + (off >> 6) + r;                 // n / 64 + 1 entries of this read
		int32_t *bound = b.bound + (off >> 6) + r;
		const bool skip_limit = P.max_skip != INT_MAX && b.rk_a != nullptr;
		const uint4 *rk_a = skip_limit ? b.rk_a + off : nullptr;
		int32_t *rk_f = skip_limit ? b.rk_f + off : nullptr, *rk_p = skip_limit ? b.rk_p + off : nullptr, *rk_mark = skip_limit ? b.rk_mark + off : nullptr;
		const int2 *rk_in = skip_limit ? b.rk_in + off : nullptr;
		int2 cin = make_int2(0, -1);                       // current block: first and last rank of the inner walk (skip limit only)
		int i0 = 0, st = 0, st_in = 0, ins = 0, tied = 0;
		int d_late = 0, d_stale = 0, d_check = 0, d_l1 = 0, d_inner = 0, d_far = 0, d_evict = 0;   // MM2GB_DEBUG_PHASES
		unsigned x0_lo = 0, x0_hi = 0;                     // x of anchor i0
		uint4 ca = make_uint4(0, 0, 0, 0), pa = ca;        // anchors of the current block and of the one before
		int4 cm = make_int4(0, 0, 0, 0);                   // current block: rank, first and last rank of the query
		int cf = 0, cp = 0, pf = 0, prank = 0;
		long long ck = 0, pk = 0;
		int hblk = -1, iblk = -1;
		unsigned hxl = 0, hxh = 0, ixl = 0, ixh = 0;       // x of the blocks st and st_in are in
		for (int i = 0; i < n; ++i) {
			const int k = i & 63, B = i >> 6;
			if (k == 0) {
				if (i > 0) {
					const int base = (B - 1) << 6;
					f[base + l] = cf; p[base + l] = cp;
					const int top = wave_max_i32(cf + (int)(ca.w & 0xffu));
					if (l == 0) { ring[w][(B - 1) & (RMQ_RING - 1)] = top; bound[B - 1] = top; }
					pa = ca; pf = cf; pk = ck; prank = cm.x;
				}
				ca = i + l < n ? a[i + l] : make_uint4(0, 0, 0, 0);
				cm = i + l < n ? meta[i + l] : make_int4(0, 0, 0, 0);
				if (skip_limit) cin = i + l < n ? rk_in[i + l] : make_int2(0, -1);
				cf = 0; cp = 0; ck = 0;
			}
			const unsigned xi_lo = (unsigned)__builtin_amdgcn_readlane((int)ca.x, k), xi_hi = (unsigned)__builtin_amdgcn_readlane((int)ca.y, k);
			const int yi = __builtin_amdgcn_readlane((int)ca.z, k), q_i = __builtin_amdgcn_readlane((int)ca.w, k) & 0xff;
			const int q_lo = __builtin_amdgcn_readlane(cm.y, k), q_hi = __builtin_amdgcn_readlane(cm.z, k);
			if (i == 0) { x0_lo = xi_lo; x0_hi = xi_hi; }
			if (i0 < i && (x0_lo != xi_lo || x0_hi != xi_hi)) { i0 = i; x0_lo = xi_lo; x0_hi = xi_hi; }     // lchain.c:279-292
			// eviction (lchain.c:293-310): the conditions hold for a prefix of [st, i); a block of 64 candidates is tested at once
			for (;;) {
				const int sb = st >> 6;
				unsigned xl, xh;
				if (sb == B) { xl = ca.x; xh = ca.y; }
				else if (sb == B - 1) { xl = pa.x; xh = pa.y; }
				else {
					if (hblk != sb) { const uint2 t = *(const uint2*)&a[(sb << 6) + l]; hxl = t.x; hxh = t.y; hblk = sb; }
					xl = hxl; xh = hxh;
				}
				const int j = (sb << 6) + l;
				++d_evict;
				const bool out = j < st || (j < i && (xh != xi_hi || (unsigned long long)xi_lo > (unsigned long long)xl + (unsigned)max_dist || (i0 > j ? i0 - j : 0) > P.cap_rmq_size));
				const int adv = uni(leading_out(~__ballot(out)));
				st = (sb << 6) + adv;
				if (adv < W) break;
			}
			if (max_inner > 0)
				for (;;) {
					const int sb = st_in >> 6;
					unsigned xl, xh;
					if (sb == B) { xl = ca.x; xh = ca.y; }
					else if (sb == B - 1) { xl = pa.x; xh = pa.y; }
					else {
						if (iblk != sb) { const uint2 t = *(const uint2*)&a[(sb << 6) + l]; ixl = t.x; ixh = t.y; iblk = sb; }
						xl = ixl; xh = ixh;
					}
					const int j = (sb << 6) + l;
					const bool out = j < st_in || (j < i && (xh != xi_hi || (unsigned long long)xi_lo > (unsigned long long)xl + (unsigned)max_inner || (i0 > j ? i0 - j : 0) > P.cap_rmq_size));
					const int adv = uni(leading_out(~__ballot(out)));
					st_in = (sb << 6) + adv;
					if (adv < W) break;
				}
			// anchors that wait to enter and are no longer in registers (a run of equal x longer than a block, or the block change): now, one by one
			if (ins < i0 && ins < (B - 1) * W) wave_sync();
			while (ins < i0 && ins < (B - 1) * W) {
				const uint4 e = a[ins];
				const int rank = meta[ins].x;
				const long long kk = key_order((double)f[ins] + half_gap * (double)((int)e.x + (int)e.z));
				if (l == 0) key[rank] = kk;
				wave_sync();
				rmq_block_summary(rank >> 6, n, st, key, ord_idx, l1);
				++ins; ++d_late;
			}
			wave_sync();                                       // the stores of earlier steps before this step's loads
			// the anchor that enters in this step (if any): its block's summary is read together with the query's data and rewritten after it
			const bool enter = ins < i0;
			int e_rank = 0;
			long long e_key = 0;
			uint4 e_sum = make_uint4(0, 0, 0, 0);
			if (enter) {
				const int src = ins & 63;
				if ((ins >> 6) == B) { e_rank = __builtin_amdgcn_readlane(cm.x, src); e_key = (long long)readlane64((unsigned long long)ck, src); }
				else { e_rank = __builtin_amdgcn_readlane(prank, src); e_key = (long long)readlane64((unsigned long long)pk, src); }
				e_sum = l1[e_rank >> 6];
			}
			int max_f = q_i, max_j = -1, max_rank = -1;
			RmqCand c;
			c.ord = RMQ_NONE; c.j = -1; c.rank = -1; c.tie = 0; c.check = -1;
			{	// not yet in the tree's arrays: straight from the registers
				const int j = (B << 6) + l;
				rmq_offer(c, j >= ins && j >= st && j < i0 && cm.x >= q_lo && cm.x <= q_hi, ck, j, cm.x, 0, -1);
			}
			if (B > 0) {
				const int j = ((B - 1) << 6) + l;
				rmq_offer(c, j >= ins && j >= st && j < i0 && prank >= q_lo && prank <= q_hi, pk, j, prank, 0, -1);
			}
			if (q_lo <= q_hi && ins > st) {
				const int lb = q_lo >> 6, hb = q_hi >> 6;
				{
					const int rk = (lb << 6) + l;
					if (rk >= q_lo && rk <= q_hi) { const long long kk = key[rk]; const int id = ord_idx[rk]; rmq_offer(c, kk != RMQ_NONE && id >= st, kk, id, rk, 0, -1); }
				}
				if (hb > lb) {
					const int rk = (hb << 6) + l;
					if (rk <= q_hi) { const long long kk = key[rk]; const int id = ord_idx[rk]; rmq_offer(c, kk != RMQ_NONE && id >= st, kk, id, rk, 0, -1); }
				}
				for (int bb0 = lb + 1; bb0 < hb; bb0 += W) {
					++d_l1;
					const int bb = bb0 + l;
					bool stale = false;
					if (bb < hb) {
						const uint4 e = l1[bb];
						const long long kk = (long long)((unsigned long long)e.y << 32 | e.x);
						const int id = (int)(e.z & 0x7fffffffu);
						stale = kk != RMQ_NONE && id < st;
						rmq_offer(c, kk != RMQ_NONE && !stale, kk, id, (bb << 6) + (int)e.w, 0, (e.z >> 31) ? bb : -1);
					}
					unsigned long long m = __ballot(stale);
					while (m) {                                    // the holder of a block's largest key has left the window: look at the block
						++d_stale;
						const int s = first_set(m);
						m &= m - 1;
						const uint4 e = rmq_block_summary(bb0 + s, n, st, key, ord_idx, l1);
						const long long kk = (long long)((unsigned long long)e.y << 32 | e.x);
						if (l == s) rmq_offer(c, kk != RMQ_NONE, kk, (int)(e.z & 0x7fffffffu), ((bb0 + s) << 6) + (int)e.w, (int)(e.z >> 31), -1);
					}
				}
			}
			if (enter) {                                       // the entering anchor's key and its block's summary, while the query is reduced
				const long long cur = (long long)((unsigned long long)e_sum.y << 32 | e_sum.x);
				const int holder = (int)(e_sum.z & 0x7fffffffu), r6 = e_rank & 63;
				if (l == 0) key[e_rank] = e_key;
				if (cur != RMQ_NONE && holder < st) {              // that summary's holder has left: rebuilt with the new anchor in it
					wave_sync();
					rmq_block_summary(e_rank >> 6, n, st, key, ord_idx, l1);
				} else if (cur == RMQ_NONE || e_key > cur) {
					if (l == 0) l1[e_rank >> 6] = make_uint4((unsigned)e_key, (unsigned)((unsigned long long)e_key >> 32), (unsigned)ins, (unsigned)r6);
				} else if (e_key == cur) {
					const bool later = r6 > (int)e_sum.w;
					if (l == 0) l1[e_rank >> 6] = make_uint4(e_sum.x, e_sum.y, (unsigned)(later ? ins : holder) | 0x80000000u, (unsigned)(later ? r6 : (int)e_sum.w));
				}
				++ins;
			}
			long long top = wave_max_i64(c.ord);
			bool inner = false;
			if (top != RMQ_NONE) {
				unsigned long long m = __ballot(c.ord == top && c.check >= 0);
				if (m) wave_sync();
				while (m) {                                        // "several holders" on a winning summary: believed after a look at the block
					const int s = first_set(m);
					m &= m - 1;
					const int blk = __builtin_amdgcn_readlane(c.check, s);
					++d_check;
					const uint4 e = rmq_block_summary(blk, n, st, key, ord_idx, l1);
					if (l == s) { c.j = (int)(e.z & 0x7fffffffu); c.rank = (blk << 6) + (int)e.w; c.tie = (int)(e.z >> 31); c.check = -1; }
				}
				unsigned long long win = __ballot(c.ord == top);
				if (__popcll(win) > 1 || __ballot(c.ord == top && c.tie) != 0) {
					++tied;                                        // the stated rule among equal keys: largest (y, index), i.e. largest rank
					const int rr = wave_max_i32(c.ord == top ? c.rank : -1);
					win = __ballot(c.ord == top && c.rank == rr);
				}
				const int bj = __builtin_amdgcn_readlane(c.j, first_set(win)), brank = __builtin_amdgcn_readlane(c.rank, first_set(win));
				unsigned xj; int yj, sj, fj;
				if ((bj >> 6) == B) {
					const int s = bj & 63;
					xj = (unsigned)__builtin_amdgcn_readlane((int)ca.x, s); yj = __builtin_amdgcn_readlane((int)ca.z, s);
					sj = __builtin_amdgcn_readlane((int)ca.w, s) & 0xff; fj = __builtin_amdgcn_readlane(cf, s);
				} else if ((bj >> 6) == B - 1) {
					const int s = bj & 63;
					xj = (unsigned)__builtin_amdgcn_readlane((int)pa.x, s); yj = __builtin_amdgcn_readlane((int)pa.z, s);
					sj = __builtin_amdgcn_readlane((int)pa.w, s) & 0xff; fj = __builtin_amdgcn_readlane(pf, s);
				} else {
					++d_far;
					const uint4 e = a[bj];
					xj = e.x; yj = (int)e.z; sj = (int)(e.w & 0xffu); fj = f[bj];
				}
				bool exact; int width;
				const int sc = fj + rmq_pair_score(xi_lo, yi, xj, yj, sj, P, exact, width);
				if (width <= P.bw && sc > max_f) { max_f = sc; max_j = bj; max_rank = brank; }
				max_f = uni(max_f); max_j = uni(max_j); max_rank = uni(max_rank);
				inner = uni((int)(!exact && max_inner > 0 && st_in < i0 && yi > 0)) != 0;
			}
			if (inner && skip_limit) {
				// lchain.c:320-341 as written: the inner tree's elements with y in [yi - max_inner, yi - 1] from the largest (y, index) down -- ranks
				// [r_lo, r_hi] downwards, 64 per round, lane 0 the first met; those still in the inner tree (index in [st_in, i0)) and inside the
				// band take part.  A strictly better score replaces the best and takes one off the skip counter; a candidate that does not, and whose
				// chain this anchor has been offered already (the mark: some candidate met earlier has it as predecessor), adds one, and the walk
				// ends when the counter passes the limit; every candidate that takes part marks its predecessor.  All of that per round by scans
				// over the lanes (the counter: wave_skip_counts; marks inside the round: one 64-bit mask; marks for later rounds: rk_mark).
				const int r_lo = __builtin_amdgcn_readlane(cin.x, k), r_hi = __builtin_amdgcn_readlane(cin.y, k);
				int n_skip = 0;
				for (int top = r_hi; top >= r_lo; top -= W) {
					const int rk = top - l;
					const bool have = rk >= r_lo;
					// (scores, predecessors and marks change under the walks: they are read past the CU's cache.  Tried: the first round's four loads
					// asked for at the start of the step, under the outer query -- 266 -> 264 ms on profiles/experiments/rmq_skip_rate.py: not kept)
					const uint4 e = have ? rk_a[rk] : make_uint4(0, 0, 0, 0);
					const int fj = have ? __hip_atomic_load(rk_f + rk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
					const int pr = have ? __hip_atomic_load(rk_p + rk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
					const int mk = have ? __hip_atomic_load(rk_mark + rk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
					const int j = (int)e.w;
					bool ex2; int w2;
					const int s2 = fj + rmq_pair_score(xi_lo, yi, e.x, (int)e.y, (int)e.z, P, ex2, w2);
					const bool part = have && j >= st_in && j < i0 && w2 <= P.bw;
					const int below = wave_max_below(part ? s2 : INT_MIN);         // (every lane takes part in the scan: not inside a short-circuit)
					const bool better = part && s2 > max(max_f, below);
					const int to = top - pr;                                       // the lane of this round that holds the predecessor (if any: to > l)
					const unsigned long long offered = wave_or_u64(part && pr >= 0 && to < W ? 1ull << to : 0ull);
					const bool again = part && !better && (mk == i || ((offered >> l) & 1));
					const int count = wave_skip_counts(better ? -1 : again ? 1 : 0, better ? 0 : SKIP_NONE, n_skip);
					const unsigned long long ends = __ballot(again && count > P.max_skip);
					const int upto = ends ? first_set(ends) : W;                   // lanes below `upto` were met before the walk ended
					const unsigned long long got = __ballot(better) & (upto < W ? (1ull << upto) - 1 : ~0ull);
					if (got) {
						const int last = 63 - first_set_from_top(got);             // every better one replaced the one before: the last stands
						max_f = __builtin_amdgcn_readlane(s2, last); max_j = __builtin_amdgcn_readlane(j, last); max_rank = __builtin_amdgcn_readlane(rk, last);
					}
					if (ends) break;
					n_skip = __builtin_amdgcn_readlane(count, W - 1);
					if (part && pr >= 0 && to >= W) __hip_atomic_store(rk_mark + pr, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a predecessor has a lower rank: it comes in a later round, or never)
					wave_sync();
				}
			} else if (inner) {
				// lchain.c:320-341: every inner-tree element with y in [yi - max_inner, yi - 1], from the largest (y, index) down; strict '>'
				// keeps the first of equal scores, i.e. the largest (y, index), and nothing replaces the outer result without beating it
				int bs = max_f, cj = -1, cy = 0;
				const int y_top = yi - 1, y_bot = yi - max_inner;
				auto offer = [&](bool in, unsigned xj, int yj, int sj, int fj, int j) {
					if (!in || yj > y_top || yj < y_bot) return;
					bool ex2; int w2;
					const int s2 = fj + rmq_pair_score(xi_lo, yi, xj, yj, sj, P, ex2, w2);
					if (w2 > P.bw) return;
					if (s2 > bs || (s2 == bs && cj >= 0 && (yj > cy || (yj == cy && j > cj)))) { bs = s2; cj = j; cy = yj; }
				};
				{ const int j = (B << 6) + l; offer(j >= st_in && j < i0, ca.x, (int)ca.z, (int)(ca.w & 0xffu), cf, j); }
				if (B > 0) { const int j = ((B - 1) << 6) + l; offer(j >= st_in && j < i0, pa.x, (int)pa.z, (int)(pa.w & 0xffu), pf, j); }
				const int e_hi = min(i0, (B - 1) * W);
				if (st_in < e_hi) {
					const int b_lo = st_in >> 6, b_hi = (e_hi + 63) >> 6;        // blocks b_lo .. b_hi - 1, all finished
					for (int bb0 = b_lo; bb0 < b_hi; bb0 += W) {
						const int bb = bb0 + l;
						bool look = false;
						if (bb < b_hi) look = (bb >= B - RMQ_RING ? ring[w][bb & (RMQ_RING - 1)] : bound[bb]) > max_f;
						unsigned long long m = __ballot(look);
						while (m) {
							// up to four blocks at a time: their loads are issued together (one round trip), then scored
							int jj[4];
							uint4 ee[4];
							int ff[4];
#pragma unroll
							for (int u = 0; u < 4; ++u) {
								jj[u] = -1;
								if (m) { jj[u] = ((bb0 + first_set(m)) << 6) + l; m &= m - 1; ++d_inner; }
							}
#pragma unroll
							for (int u = 0; u < 4; ++u) {
								const bool in = jj[u] >= st_in && jj[u] < e_hi;        // (jj = -1: no block)
								ee[u] = in ? a[jj[u]] : make_uint4(0, 0, 0, 0);
								ff[u] = in ? f[jj[u]] : 0;
							}
#pragma unroll
							for (int u = 0; u < 4; ++u) offer(jj[u] >= st_in && jj[u] < e_hi, ee[u].x, (int)ee[u].z, (int)(ee[u].w & 0xffu), ff[u], jj[u]);
						}
					}
				}
				const int best = wave_max_i32(cj >= 0 ? bs : INT_MIN);
				if (best != INT_MIN) {
					unsigned long long win = __ballot(cj >= 0 && bs == best);
					if (__popcll(win) > 1) {
						const int yy = wave_max_i32(cj >= 0 && bs == best ? cy : INT_MIN);
						const int jj = wave_max_i32(cj >= 0 && bs == best && cy == yy ? cj : -1);
						win = __ballot(cj >= 0 && bs == best && cy == yy && cj == jj);
					}
					max_f = best; max_j = __builtin_amdgcn_readlane(cj, first_set(win));
				}
			}
			if (l == k) {                                                                              // lchain.c:346 (+ the key of lchain.c:284)
				cf = max_f;
				cp = max_j < 0 ? 0 : i - max_j;
				ck = key_order((double)max_f + half_gap * (double)((int)xi_lo + yi));
				if (skip_limit) {                                                                      // (the next step's loads come after a wave_sync)
					__hip_atomic_store(rk_f + cm.x, max_f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(rk_p + cm.x, max_j < 0 ? -1 : max_rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
		}
		if (n > 0) {
			const int base = ((n - 1) >> 6) << 6;
			if (base + l < n) { f[base + l] = cf; p[base + l] = cp; }
		}
		if (l == 0) b.n_tied[r] = tied;
		if (b.dbg && l == 0) {
			const long long v[8] = { n, d_late, d_stale, d_check, d_l1, d_inner, d_far, d_evict };
			for (int q = 0; q < 8; ++q) atomicAdd((unsigned long long*)&b.dbg[q], (unsigned long long)v[q]);
		}
		wave_sync();
	}
}

// --------------------------------------------------------------------------------------------------------------
// RMQ re-chaining, tile form (round 3).  k_rmq_fill above takes one anchor per step: a chain of dependent memory round trips per
// anchor (query the summaries, fetch the winner, scan the inner window), microseconds each.  Here a wave takes 64 consecutive anchors
// per step -- a TILE, one anchor per lane -- the way the chaining DP does (chain_kernels.hip):
//   * what every anchor of the tile may look at splits into (1) anchors that are settled for the whole tile -- in reach of every
//     lane, evicted for none: index in [lo, hi), lo = the LAST lane's window start, hi = the first lane's i0 -- which live in a binary
//     tournament tree over the read's (y, index) ranks in global memory (key = the negated priority of lchain.c:284, a node = the best
//     key below it, its rank, "shared by several"): every lane asks it for the best key in its own rank interval, O(log n) INDEPENDENT
//     loads, all 64 queries side by side; (2) the few anchors around the edges -- leaving the window during the tile, [st of lane 0,
//     lo), or waiting to enter, [hi, tile start) -- and every anchor of the inner window (lchain.c:320-341 scores ALL of them): these
//     are read 64 at a time, coalesced, and broadcast one by one to all lanes, each lane testing its own window, rank interval and
//     query range; (3) the tile's own anchors, which depend on each other: 64 serial steps, all in registers -- lane t's result is
//     final at step t and is broadcast to the lanes above it.
//   * the tree is kept exact by batches: before a tile's queries, the leaves of the anchors that left [lo, hi) are emptied and those
//     that entered are set, one leaf per lane, and the lanes walk up to the root together, each recomputing the nodes on its path from
//     their children; a lane that is a level behind another on a common path rewrites it with final children, so the last write of
//     every node is right.
// Same definition of the result as k_rmq_fill and the oracle (orc_rmq_fill), incl. the stated tie rule and the count of tied anchors.
// --------------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ uint4 tnode_none() { return make_uint4(0u, 0x80000000u, 0u, 0u); }      // key = RMQ_NONE
__device__ __forceinline__ long long tnode_key(const uint4 &e) { return (long long)((unsigned long long)e.y << 32 | e.x); }
__device__ __forceinline__ uint4 tnode_comb(const uint4 &a, const uint4 &b)
{
	const long long ka = tnode_key(a), kb = tnode_key(b);
	if (ka > kb) return a;
	if (kb > ka) return b;
	if (ka == RMQ_NONE) return a;                          // both empty
	const unsigned ra = a.z & 0x7fffffffu, rb = b.z & 0x7fffffffu;
	return make_uint4(a.x, a.y, (ra > rb ? ra : rb) | 0x80000000u, 0u);   // equal keys: the larger (y, index) rank, and "several share it"
}

// the best candidate of the outer query a lane has seen so far (lchain.c:311-318): its key, its place in the tie rule, and what the lane
// gets from it -- the pair's score, whether it is an exact diagonal extension, the diagonal distance -- computed when it is adopted
struct TileCand { long long key; int rank, tie, j, sc, exact, width; };
// (selects only, no branches: a lone wave on its SIMD pays every taken branch and every change of the execution mask in full)
__device__ __forceinline__ void tile_offer(TileCand &c, bool in, long long key, int rank, int tie, int j, int sc, int exact, int width)
{
	const bool better = in & (key > c.key), same = in & (key == c.key);
	const bool take = better | (same & (rank > c.rank));
	c.tie = better ? tie : same ? 1 : c.tie;
	c.key = take ? key : c.key;
	c.rank = take ? rank : c.rank; c.j = take ? j : c.j; c.sc = take ? sc : c.sc; c.exact = take ? exact : c.exact; c.width = take ? width : c.width;
}
// the best of the inner window (lchain.c:328-341 with max_chn_skip = infinity): largest score, then largest (y, index)
struct TileInner { int s, y, j; };
__device__ __forceinline__ void tile_offer_inner(TileInner &c, bool in, int s2, int yj, int j)
{
	const bool take = in & ((c.j < 0) | (s2 > c.s) | ((s2 == c.s) & ((yj > c.y) | ((yj == c.y) & (j > c.j)))));
	c.s = take ? s2 : c.s; c.y = take ? yj : c.y; c.j = take ? j : c.j;
}
// comput_sc_simple (lchain.c:232-248) without a branch: same arithmetic as rmq_pair_score, the penalty selected rather than skipped
__device__ __forceinline__ int tile_pair_score(unsigned xi, int yi, unsigned xj, int yj, int q_span_j, const RmqParams &P, int &exact, int &width)
{
	const int dq = yi - yj, dr = (int)(xi - xj);
	const int dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq;
	const int sc = q_span_j < dg ? q_span_j : dg;
	width = dd;
	exact = (dd == 0) & (dg <= q_span_j);
	const float lin = P.pen_gap * (float)dd + P.pen_skip * (float)dg;
	const float lg = dd >= 1 ? rmq_log2((float)(dd + 1)) : 0.0f;
	const int pen = (int)(lin + .5f * lg);
	return sc - (((dd != 0) | (dq > q_span_j)) ? pen : 0);
}


// Several anchors hold the smallest priority of an anchor's query, and which of them the reference returns follows from the shape of its
// tree.  That need not be known where every holder leaves the anchor with the same score and predecessor: the tree's content does not
// depend on the pick (priorities come from f[] alone) and -- there is no skip limit here -- neither does the inner walk, whose result is
// the first candidate, in walking order, to reach the walk's largest score wherever the walk started from below that.  So: the anchor's
// whole inner window for its best (nothing passed over on the strength of one holder's score), then every holder of the key in the
// anchor's window and rank interval followed through lchain.c:316-341 and compared with what the tile made of the anchor.  Rare (a few
// anchors in a hundred thousand), so by the plain means: the whole wave, 64 anchors of the window at a time, from memory; the tile's own
// anchors below the one in question are still in registers, one per lane (lanes < n_own).  Returns the same in every lane.
// (csrc/rmq_host.cpp weighs its ties the same way; the oracle counts them: orc_rmq_last_ties_that_decide.)
__device__ __noinline__ bool rmq_tie_decides(const uint4 *a, const int32_t *f, const int4 *meta, double half_gap, RmqParams P, int max_inner,
                                             unsigned xt, int yt, int q_t, long long key_t, int rk_lo, int rk_hi, int st, int st_in, int i0, int res_f, int res_j,
                                             int tb, int n_own, uint4 A_own, int f_own, long long k_own, int rk_own)
{
	const int l = lane();
	const bool inner_there = max_inner > 0 && st_in < i0 && yt > 0;
	const int y_top = yt - 1, y_bot = yt - max_inner, mem_end = min(i0, tb);
	TileInner best; best.s = 0; best.y = 0; best.j = -1;
	if (inner_there) {
		for (int base = st_in; base < mem_end; base += W) {
			const int j = base + l;
			if (j < mem_end) {
				const uint4 e = a[j];
				int ex, wd;
				const int s2 = f[j] + tile_pair_score(xt, yt, e.x, (int)e.z, (int)(e.w & 0xffu), P, ex, wd);
				tile_offer_inner(best, ((int)e.z <= y_top) & ((int)e.z >= y_bot) & (wd <= P.bw), s2, (int)e.z, j);
			}
		}
		{
			const int j = tb + l;
			int ex, wd;
			const int s2 = f_own + tile_pair_score(xt, yt, A_own.x, (int)A_own.z, (int)(A_own.w & 0xffu), P, ex, wd);
			tile_offer_inner(best, (l < n_own) & (j >= st_in) & (j < i0) & ((int)A_own.z <= y_top) & ((int)A_own.z >= y_bot) & (wd <= P.bw), s2, (int)A_own.z, j);
		}
		for (int o = W / 2; o > 0; o >>= 1) tile_offer_inner(best, __shfl_xor(best.j, o) >= 0, __shfl_xor(best.s, o), __shfl_xor(best.y, o), __shfl_xor(best.j, o));
	}
	bool differs = false;
	auto weigh = [&](bool holds, unsigned xj, int yj, int sj, int fj, int j) {
		int ex, wd;
		const int sc = fj + tile_pair_score(xt, yt, xj, yj, sj, P, ex, wd);
		int o_f = q_t, o_j = -1;
		if (wd <= P.bw && sc > o_f) { o_f = sc; o_j = j; }
		if (!ex && inner_there && best.j >= 0 && best.s > o_f) { o_f = best.s; o_j = best.j; }
		differs |= holds & ((o_f != res_f) | (o_j != res_j));
	};
	for (int base = st; base < mem_end; base += W) {
		const int j = base + l;
		if (j < mem_end) {
			const uint4 e = a[j];
			const int fj = f[j], rk = meta[j].x;
			const long long kj = key_order((double)fj + half_gap * (double)((int)e.x + (int)e.z));
			weigh((kj == key_t) & (rk >= rk_lo) & (rk <= rk_hi), e.x, (int)e.z, (int)(e.w & 0xffu), fj, j);
		}
	}
	{
		const int j = tb + l;
		weigh((l < n_own) & (j >= st) & (j < i0) & (k_own == key_t) & (rk_own >= rk_lo) & (rk_own <= rk_hi), A_own.x, (int)A_own.z, (int)(A_own.w & 0xffu), f_own, j);
	}
	return __ballot(differs) != 0;
}

} // namespace

// per anchor: its window starts and the start of its run of equal x, closed forms of the carried values of lchain.c:279-310 (the
// conditions of the eviction loops hold for a prefix of the anchors before i, so the loops end at the largest of the three bounds)
__global__ __launch_bounds__(256) void k_rmq_prep_windows(RmqBatch b, int max_dist, int max_inner, int cap)
{
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < b.n; g += (int64_t)gridDim.x * blockDim.x) {
		const int64_t r = rmq_read_of(b.offsets, b.n_reads, g);
		const int64_t off = b.offsets[r];
		const uint4 *a = b.raw + off;
		const int i = (int)(g - off);
		const unsigned xl = a[i].x, xh = a[i].y;
		// first index of the run of equal x that holds i (anchors are sorted by x)
		int lo = 0, hi = i;
		while (lo < hi) { const int mid = (lo + hi) >> 1; const uint2 v = *(const uint2*)&a[mid]; if (v.y < xh || (v.y == xh && v.x < xl)) lo = mid + 1; else hi = mid; }
		const int i0 = lo;
		auto first_in_reach = [&](int dist) {                 // first j <= i on the same strand | reference with x_i <= x_j + dist
			int l2 = 0, h2 = i;
			while (l2 < h2) {
				const int mid = (l2 + h2) >> 1;
				const uint2 v = *(const uint2*)&a[mid];
				const bool in = v.y == xh && (unsigned long long)xl <= (unsigned long long)v.x + (unsigned)dist;
				if (in) h2 = mid; else l2 = mid + 1;
			}
			return l2;
		};
		const int by_cap = i0 > cap ? i0 - cap : 0;
		const int st = min(i, max(first_in_reach(max_dist), by_cap));
		const int st_in = max_inner > 0 ? min(i, max(first_in_reach(max_inner), by_cap)) : i;
		b.win[g] = make_int4(st, st_in, i0, 0);
	}
}

__global__ __launch_bounds__(256) void k_rmq_tree_init(RmqBatch b)
{
	const uint4 none = tnode_none();
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < 2 * b.n; g += (int64_t)gridDim.x * blockDim.x) b.tree[g] = none;
}

// ---- the inner window by strips of y ---------------------------------------------------------------------------------------
// The inner scan of lchain.c:323-341 takes, for anchor i, the anchors still in the inner tree -- indices [st_inner, i0) -- whose y lies in
// [y_i - max_dist_inner, y_i - 1].  Where a read crosses a tandem array thousands of anchors share that index window and a few hundred
// that y range: sweeping the window (every anchor against all 64 lanes of a tile) was 96 % of the slowest reads' time.  So every read's
// anchors are also kept sorted by (y >> shift, index) with 2^shift >= max_dist_inner: a lane's candidates then lie in at most two strips, and
// within a strip its index window is ONE range of that order -- found here, for every anchor, by bisection.  The fill kernel walks the two
// ranges lane by lane: what it scores is what the reference's tree iteration visits, times at most two.
__global__ __launch_bounds__(256) void k_rmq_strip_keys(RmqBatch b)
{
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < b.n; g += (int64_t)gridDim.x * blockDim.x) {
		const int64_t off = b.offsets[rmq_read_of(b.offsets, b.n_reads, g)];
		b.skey_in[g] = (unsigned long long)(b.raw[g].z >> b.strip_shift) << 32 | (unsigned)(g - off);
	}
}
__global__ __launch_bounds__(256) void k_rmq_strip_fill(RmqBatch b)
{
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < b.n; g += (int64_t)gridDim.x * blockDim.x) {
		const int64_t off = b.offsets[rmq_read_of(b.offsets, b.n_reads, g)];
		const unsigned j = (unsigned)b.skey[g];
		const uint4 e = b.raw[off + j];
		b.sa[g] = make_uint4(e.x, e.z, j, e.w & 0xffu);
		((int32_t*)b.skey_in)[b.n + off + j] = (int32_t)(g - off);      // spos (k_rmq_fill_tiles): the keys' unsorted copy is not read again
	}
}
__global__ __launch_bounds__(256) void k_rmq_strip_ranges(RmqBatch b, int max_inner)
{
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < b.n; g += (int64_t)gridDim.x * blockDim.x) {
		const int64_t r = rmq_read_of(b.offsets, b.n_reads, g);
		const int64_t off = b.offsets[r];
		const int n = (int)(b.offsets[r + 1] - off);
		const unsigned long long *k = b.skey + off;
		const int4 wn = b.win[g];                                 // st, st_inner, i0
		const int yi = (int)b.raw[g].z;
		int4 out = make_int4(0, 0, 0, 0);
		if (max_inner > 0 && yi > 0 && wn.y < wn.z) {
			auto first_not_below = [&](unsigned long long key) {
				int lo = 0, hi = n;
				while (lo < hi) { const int mid = (lo + hi) >> 1; if (k[mid] < key) lo = mid + 1; else hi = mid; }
				return lo;
			};
			const unsigned long long s_hi = (unsigned long long)((unsigned)(yi - 1) >> b.strip_shift);
			const int y_bot = yi - max_inner;
			const unsigned long long s_lo = y_bot > 0 ? (unsigned long long)((unsigned)y_bot >> b.strip_shift) : 0ull;
			out.z = first_not_below(s_hi << 32 | (unsigned)wn.y); out.w = first_not_below(s_hi << 32 | (unsigned)wn.z);
			if (s_lo < s_hi) { out.x = first_not_below(s_lo << 32 | (unsigned)wn.y); out.y = first_not_below(s_lo << 32 | (unsigned)wn.z); }
		}
		b.srange[g] = out;
	}
}

constexpr int RMQ_THREADS = 1024;              // k_rmq_fill_tiles: 16 waves -- one read each, or all 16 on one read
constexpr int RMQ_TEAM = RMQ_THREADS / W;
constexpr int RMQ_MERGE_WORDS = 12;
// A team's read keeps the top of its tournament tree -- nodes 1 .. RMQ_TOP_NODES - 1, eleven levels -- in LDS (`s_top`; only the tree wave
// touches it): a tile's update climbs every level behind a store and a load of the level below, ~1.6 us per level in global memory, 31 us per
// tile of a 450 k-anchor read (19 levels), a third of such a read's time (profiles/r03_rmq_teams.txt).
#ifndef MM2GB_RMQ_TOP_NODES
#define MM2GB_RMQ_TOP_NODES 2048
#endif
constexpr int RMQ_TOP_NODES = MM2GB_RMQ_TOP_NODES;
// One read by ONE wave (the reads of a call that are not worth a whole workgroup; rmq_fill_read_team below has the others).
__device__ __forceinline__ void rmq_fill_read_tiles(const RmqBatch &b, const RmqParams &P, const int r)
{
	const int l = lane();
	const int max_dist = P.max_dist < P.bw ? P.bw : P.max_dist;                                       // lchain.c:264
	const int max_inner = (P.max_dist_inner <= 0 || P.max_dist_inner >= max_dist) ? 0 : P.max_dist_inner;   // lchain.c:265
	const double half_gap = 0.5 * (double)P.pen_gap;
	{
		const int64_t off = b.offsets[r];
		const int n = (int)(b.offsets[r + 1] - off);
		const uint4 *a = b.raw + off;
		const int4 *meta = b.meta + off, *win = b.win + off;
		const bool strips = b.sa != nullptr && max_inner > 0;     // the inner window by strips of y (k_rmq_strip_ranges) instead of block sweeps
		const uint4 *sa = b.sa + off;
		const int4 *srange = b.srange + off;
		// strips: every anchor's score also at its place in the strip order (the memory of the sort's input keys, free once the order stands):
		// sf[position] for the inner scans, spos[index] = the position, for whoever settles the anchor
		int32_t *sf = (int32_t*)b.skey_in + off;
		const int32_t *spos = (const int32_t*)b.skey_in + b.n + off;
		const int32_t *ord_idx = b.ord_idx + off;
		int32_t *f = b.f + off, *p = b.p + off;
		uint4 *tree = b.tree + 2 * off;                      // node q of this read: tree[q], leaves at n + rank, root 1
		auto tld = [&](int q) -> uint4 { return tree[q]; };
		auto tst = [&](int q, const uint4 &v) { tree[q] = v; };
		int32_t *bound = b.bound + (off >> 6) + r;           // per block of 64 anchors (by index) the largest f + span, once its tile is done
		int ev = 0, ins = 0, tied = 0;                       // the tree holds the anchors of index [ev, ins)
		const long long t_read0 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
		long long d_tiles = 0, d_upd = 0, d_levels = 0, d_qloads = 0, d_bcast = 0, d_skip = 0, d_t3 = 0, d_redo = 0, d_single = 0;   // MM2GB_DEBUG_PHASES: 100 MHz ticks of the tree update, the queries, the broadcasts, the in-tile steps
		for (int tb = 0; tb < n; tb += W) {
			const int n_here = min(W, n - tb), i = tb + l;
			const bool live = l < n_here;
			const uint4 A = live ? a[i] : make_uint4(0, 0, 0, 0);
			const int4 M = live ? meta[i] : make_int4(0, 1, 0, 0);             // rank, first and last rank of the query (dead lanes: empty)
			const int4 Wn = live ? win[i] : make_int4(INT_MAX, INT_MAX, 0, 0);   // st, st_inner, i0 (dead lanes: nothing is in reach, nothing came before)
			const int sp_i = (strips && live) ? spos[i] : 0;
			const unsigned xi = A.x;
			const int yi = (int)A.z, q_i = (int)(A.w & 0xffu);
			const int st_first = __builtin_amdgcn_readlane(Wn.x, 0), st_last = __builtin_amdgcn_readlane(Wn.x, n_here - 1);
			const int stin_first = __builtin_amdgcn_readlane(Wn.y, 0);
			const int hi = __builtin_amdgcn_readlane(Wn.z, 0);                   // every anchor before the first lane's run of equal x has entered for all lanes
			const int lo = min(st_last, hi);                                       // ... and none from here on has left for any
			++d_tiles;
			const long long ts0 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
			// ---- the tree: out with [ev, min(lo, ins)), in with [max(ins, lo), hi) ----
			{
				int e0 = ev, n0 = max(ins, lo);
				const int e1 = min(lo, ins), n1 = hi;
				while (e0 < e1 || n0 < n1) {
					const int je = e0 + l, ji = n0 + l;
					int pe = 0, pi = 0;
					if (je < e1) { pe = n + meta[je].x; tst(pe, tnode_none()); }
					if (ji < n1) {
						const uint4 e = a[ji];
						const int rk = meta[ji].x;
						const long long kk = key_order((double)f[ji] + half_gap * (double)((int)e.x + (int)e.z));
						pi = n + rk;
						tst(pi, make_uint4((unsigned)kk, (unsigned)((unsigned long long)kk >> 32), (unsigned)rk, 0u));
					}
					wave_sync();
					while (__ballot(pe > 1 || pi > 1) != 0) {
						uint4 c0 = tnode_none(), c1 = c0, c2 = c0, c3 = c0;
						if (pe > 1) { pe >>= 1; c0 = tld(2 * pe); c1 = tld(2 * pe + 1); } else pe = 0;
						if (pi > 1) { pi >>= 1; c2 = tld(2 * pi); c3 = tld(2 * pi + 1); } else pi = 0;
						if (pe > 0) tst(pe, tnode_comb(c0, c1));
						if (pi > 0 && pi != pe) tst(pi, tnode_comb(c2, c3));
						wave_sync();
					}
					e0 += W; n0 += W;
				}
				ev = lo; ins = hi;
			}
			const long long ts1 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
			// ---- (1) every lane's query of the tree: the nodes that tile its rank interval, bottom up; the loads do not depend on each other ----
			TileCand c;
			c.key = RMQ_NONE; c.rank = -1; c.tie = 0; c.j = -1; c.sc = 0; c.exact = 0; c.width = 0;
			{
				uint4 best = tnode_none();
				int ql = n + M.y, qr = n + M.z + 1;                                // [ql, qr) over the leaves
				bool go = live && M.y <= M.z && lo < hi;
				while (__ballot(go && ql < qr) != 0) {
					uint4 vl = tnode_none(), vr = vl;
					if (go && ql < qr) {
						if (ql & 1) vl = tld(ql++);
						if (qr & 1) vr = tld(--qr);
						ql >>= 1; qr >>= 1;
					}
					best = tnode_comb(best, tnode_comb(vl, vr));
				}
				if (tnode_key(best) != RMQ_NONE) {
					const int rk = (int)(best.z & 0x7fffffffu);
					const int j = ord_idx[rk];
					const uint4 e = a[j];
					int ex, wd;
					const int sc = f[j] + tile_pair_score(xi, yi, e.x, (int)e.z, (int)(e.w & 0xffu), P, ex, wd);
					c.key = tnode_key(best); c.rank = rk; c.tie = (int)(best.z >> 31); c.j = j; c.sc = sc; c.exact = ex; c.width = wd;
				}
			}
			const long long ts2 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
			// ---- (2) anchors before the tile that are not in the tree for every lane, and the inner window: broadcast one by one ----
			TileInner in;
			in.s = 0; in.y = 0; in.j = -1;
			int relied = INT_MIN;                                 // the largest bound of a block this lane passed over on the strength of its outer candidate's score alone
			const int y_top = yi - 1, y_bot = yi - max_inner;
			auto sweep_range = [&](int from, int to, bool outer_all) {
				// outer_all: every anchor of the range is outside the tree (it left the window for some lanes); else only those from hi on are
				for (int base = from; base < to; base += W) {
					const int j_l = base + l;
					const bool have = j_l < to;
					const uint4 e_l = have ? a[j_l] : make_uint4(0, 0, 0, 0);
					const int f_l = have ? f[j_l] : 0, rk_l = have ? meta[j_l].x : 0;
					const long long k_l = key_order((double)f_l + half_gap * (double)((int)e_l.x + (int)e_l.z));
					const int cnt = min(W, to - base);
					for (int k = 0; k < cnt; k += 2) {
						// two anchors per round, straight-line: their arithmetic interleaves (the wave is alone on its SIMD: nothing else hides latency)
						const int ja = base + k, jb2 = ja + 1;
						const bool two = k + 1 < cnt;
						const int kb = two ? k + 1 : k;
						const bool outer_a = outer_all || ja >= hi, outer_b = two && (outer_all || jb2 >= hi);     // wave-uniform: in the tree otherwise
						const bool inner_a = !strips && max_inner > 0 && ja >= stin_first, inner_b = !strips && two && max_inner > 0 && jb2 >= stin_first;   // wave-uniform
						d_bcast += two ? 2 : 1;
						if (!(outer_a || outer_b || inner_a || inner_b)) continue;
						const int ya = __builtin_amdgcn_readlane((int)e_l.z, k), yb = __builtin_amdgcn_readlane((int)e_l.z, kb);
						const int rka = __builtin_amdgcn_readlane(rk_l, k), rkb = __builtin_amdgcn_readlane(rk_l, kb);
						const bool out_a = outer_a & (ja < Wn.z) & (ja >= Wn.x) & (rka >= M.y) & (rka <= M.z);
						const bool out_b = outer_b & (jb2 < Wn.z) & (jb2 >= Wn.x) & (rkb >= M.y) & (rkb <= M.z);
						const bool in_a = inner_a & (ja < Wn.z) & (ja >= Wn.y) & (ya <= y_top) & (ya >= y_bot);
						const bool in_b = inner_b & (jb2 < Wn.z) & (jb2 >= Wn.y) & (yb <= y_top) & (yb >= y_bot);
						const unsigned long long any_out = __ballot(out_a | out_b);
						if ((any_out | __ballot(in_a | in_b)) == 0) continue;
						const unsigned xa = (unsigned)__builtin_amdgcn_readlane((int)e_l.x, k), xb = (unsigned)__builtin_amdgcn_readlane((int)e_l.x, kb);
						const int sa = __builtin_amdgcn_readlane((int)e_l.w, k) & 0xff, sb = __builtin_amdgcn_readlane((int)e_l.w, kb) & 0xff;
						const int fa = __builtin_amdgcn_readlane(f_l, k), fb = __builtin_amdgcn_readlane(f_l, kb);
						int exa, wa, exb, wb;
						const int s2a = fa + tile_pair_score(xi, yi, xa, ya, sa, P, exa, wa);     // the same pair score serves the outer query and the inner scan
						const int s2b = fb + tile_pair_score(xi, yi, xb, yb, sb, P, exb, wb);
						if (any_out != 0) {
							tile_offer(c, out_a, (long long)readlane64((unsigned long long)k_l, k), rka, 0, ja, s2a, exa, wa);
							tile_offer(c, out_b, (long long)readlane64((unsigned long long)k_l, kb), rkb, 0, jb2, s2b, exb, wb);
						}
						tile_offer_inner(in, in_a & (wa <= P.bw), s2a, ya, ja);
						tile_offer_inner(in, in_b & (wb <= P.bw), s2b, yb, jb2);
					}
				}
			};
			sweep_range(st_first, lo, true);
			{
				// From the newest block of 64 down: the nearest anchors carry the highest scores, so the lanes' inner bests rise at once, and a
				// block whose largest f + span (`bound`, written when its tile was finished) cannot beat ANY lane's is passed over --
				// comput_sc_simple never returns more than the span, the inner scan only ever replaces a result it BEATS (lchain.c:331), and an
				// equal score wins only by a larger (y, index).  Without this every anchor of a dense window was scored against every lane:
				// 5 266 broadcasts per tile on the mapper's reads (profiles/r03h_*), twenty times the work of the kernel above.
				const int from = (max_inner > 0 && !strips) ? max(min(stin_first, tb), lo) : max(hi, lo);
				for (int bb = (tb >> 6) - 1; bb >= 0 && ((bb + 1) << 6) > from; --bb) {
					const int b_lo = max(bb << 6, from), b_hi = (bb + 1) << 6;
					if (b_lo >= hi || max_inner <= 0) { sweep_range(b_lo, b_hi, false); continue; }     // holds anchors the outer query needs (not in the tree): no skipping
					if (b_hi > hi) { sweep_range(b_lo, b_hi, false); continue; }
					const int bnd = uni(bound[bb]);
					// what holds whatever happens: the block lies outside the lane's inner window, or cannot beat the span every anchor starts from, or the
					// best the lane's inner scan has found.  And what holds as things stand: it cannot beat what the lane's outer candidate scores --
					// the inner scan only replaces a result it beats (lchain.c:331), and that candidate is nearly always final or replaced by a
					// better-scoring one (the previous anchor of the chain).  A lane that passes a block over on that ground alone remembers the
					// largest such bound; if its outer result ends up below it, its inner window is scanned again, in full, when its turn comes.
					const bool sure = !live || bnd <= q_i || (in.j >= 0 && bnd < in.s) || b_hi <= Wn.y || b_lo >= Wn.z;
					const int spec = (c.key != RMQ_NONE && c.width <= P.bw) ? c.sc : INT_MIN;
					const bool idle = sure || bnd <= spec;
					const unsigned long long need = __ballot(!idle);
					if (need == 0) { ++d_skip; relied = (!sure && bnd > relied) ? bnd : relied; continue; }
					if (__popcll(need) >= 24) { sweep_range(b_lo, b_hi, false); continue; }
					// Few lanes need this block (a read's chains interleave along x: a block bounded by the best chain's scores is of no use to
					// the anchors of that chain, which sit far above it, but cannot be ruled out for the anchors of a weak one): those lanes
					// take it one at a time, the block's 64 anchors side by side -- ~100 instructions per lane instead of ~2 600 for a broadcast
					relied = (idle && !sure && bnd > relied) ? bnd : relied;
					const int j_c = (bb << 6) + l;
					const bool have = j_c >= b_lo && j_c < b_hi;
					const uint4 e_c = have ? a[j_c] : make_uint4(0, 0, 0, 0);
					const int f_c = have ? f[j_c] : 0;
					for (unsigned long long m = need; m != 0; m &= m - 1) {
						const int u = first_set(m);
						const unsigned xu = (unsigned)__builtin_amdgcn_readlane((int)xi, u);
						const int yu = __builtin_amdgcn_readlane(yi, u), from_u = __builtin_amdgcn_readlane(Wn.y, u), to_u = __builtin_amdgcn_readlane(Wn.z, u);
						int ex2, w2;
						const int s2 = f_c + tile_pair_score(xu, yu, e_c.x, (int)e_c.z, (int)(e_c.w & 0xffu), P, ex2, w2);
						const bool ok = have & (j_c >= from_u) & (j_c < to_u) & ((int)e_c.z <= yu - 1) & ((int)e_c.z >= yu - max_inner) & (w2 <= P.bw);
						const unsigned long long oks = __ballot(ok);
						++d_single;
						if (oks == 0) continue;
						const int bs = wave_max_i32(ok ? s2 : INT_MIN);
						const int by = wave_max_i32((ok & (s2 == bs)) ? (int)e_c.z : INT_MIN);
						const int bj = wave_max_i32((ok & (s2 == bs) & ((int)e_c.z == by)) ? j_c : -1);
						if (l == u) tile_offer_inner(in, true, bs, by, bj);
					}
				}
			}
			if (strips) {
				// the inner window, lane by lane: the lane's two ranges of its read's (strip, index) order, up to the anchors of this tile (theirs come
				// with the in-tile steps); four candidates per round, their loads side by side.  A team's waves take a part of every range each.
				const int4 R4 = live ? srange[i] : make_int4(0, 0, 0, 0);
				for (int pass = 0; pass < 2; ++pass) {
					const int r_lo = pass ? R4.z : R4.x, r_hi = pass ? R4.w : R4.y;
					int pp = r_lo;
					const int pe = r_hi;
					while (__ballot(pp < pe) != 0) {
						if (pp < pe) {
							uint4 cq[4];
							int fq[4];
							bool okq[4];
							// (the candidates' scores come from their copy in this order, `sf`: a round is ONE round trip to memory, not the candidate and
							// then f[its index] -- a lone wave has nothing else to hide a dependent load behind)
#pragma unroll
							for (int q = 0; q < 4; ++q) { const int at = min(pp + q, pe - 1); cq[q] = sa[at]; fq[q] = sf[at]; }
#pragma unroll
							for (int q = 0; q < 4; ++q) okq[q] = (pp + q < pe) & ((int)cq[q].z < tb);
#pragma unroll
							for (int q = 0; q < 4; ++q) {
								int ex, wd;
								const int s2 = fq[q] + tile_pair_score(xi, yi, cq[q].x, (int)cq[q].y, (int)cq[q].w, P, ex, wd);
								tile_offer_inner(in, okq[q] & ((int)cq[q].y <= y_top) & ((int)cq[q].y >= y_bot) & (wd <= P.bw), s2, (int)cq[q].y, (int)cq[q].z);
							}
							++d_single;
							pp = (okq[0] & okq[1] & okq[2] & okq[3]) ? pp + 4 : pe;   // indices rise along a range: from the first anchor of this tile on, nothing is final
						}
					}
				}
			}
			const long long ts3 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
			// ---- (3) the tile's own anchors, one after the other ----
			int f_l = q_i, p_l = 0;
			long long k_l = 0;
			for (int t = 0; t < n_here; ++t) {
				// what lane t gets from its candidates as they stand (every lane computes -- a handful of selects; lane t's is final)
				const bool has = c.key != RMQ_NONE;
				int max_f = q_i, max_j = -1;
				{ const bool use_out = has & (c.width <= P.bw) & (c.sc > max_f); max_f = use_out ? c.sc : max_f; max_j = use_out ? c.j : max_j; }
				const bool inner_on = has & !c.exact & (max_inner > 0) & (Wn.y < Wn.z) & (yi > 0);
				if (__builtin_amdgcn_readlane((int)(inner_on & (relied > max_f)), t) != 0) {
					// lane t passed blocks over that its final outer result does not rule out: its whole inner window before the tile again, 64
					// candidates at a time (the tile's own anchors have all been offered to it already)
					++d_redo;
					const unsigned xt = (unsigned)__builtin_amdgcn_readlane((int)xi, t);
					const int yt = __builtin_amdgcn_readlane(yi, t), from_t = __builtin_amdgcn_readlane(Wn.y, t), to_t = min(__builtin_amdgcn_readlane(Wn.z, t), tb);
					TileInner best; best.s = 0; best.y = 0; best.j = -1;
					for (int base = from_t; base < to_t; base += W) {
						const int j = base + l;
						if (j < to_t) {
							const uint4 e = a[j];
							int ex2, w2;
							const int s2 = f[j] + tile_pair_score(xt, yt, e.x, (int)e.z, (int)(e.w & 0xffu), P, ex2, w2);
							tile_offer_inner(best, ((int)e.z <= yt - 1) & ((int)e.z >= yt - max_inner) & (w2 <= P.bw), s2, (int)e.z, j);
						}
					}
					for (int o = W / 2; o > 0; o >>= 1) tile_offer_inner(best, __shfl_xor(best.j, o) >= 0, __shfl_xor(best.s, o), __shfl_xor(best.y, o), __shfl_xor(best.j, o));
					if (l == t) tile_offer_inner(in, best.j >= 0, best.s, best.y, best.j);
				}
				const bool use_in = inner_on & (in.j >= 0) & (in.s > max_f);
				max_f = use_in ? in.s : max_f; max_j = use_in ? in.j : max_j;
				bool counts = __builtin_amdgcn_readlane((int)(has && c.tie), t) != 0;
				if (counts && P.weigh_ties)
					counts = rmq_tie_decides(a, f, meta, half_gap, P, max_inner, (unsigned)__builtin_amdgcn_readlane((int)xi, t), __builtin_amdgcn_readlane(yi, t), __builtin_amdgcn_readlane(q_i, t),
					                         (long long)readlane64((unsigned long long)c.key, t), __builtin_amdgcn_readlane(M.y, t), __builtin_amdgcn_readlane(M.z, t),
					                         __builtin_amdgcn_readlane(Wn.x, t), __builtin_amdgcn_readlane(Wn.y, t), __builtin_amdgcn_readlane(Wn.z, t),
					                         __builtin_amdgcn_readlane(max_f, t), __builtin_amdgcn_readlane(max_j, t), tb, t, A, f_l, k_l, M.x);
				if (l == t) {
					f_l = max_f; p_l = max_j < 0 ? 0 : i - max_j;
					k_l = key_order((double)max_f + half_gap * (double)((int)xi + yi));
					tied += counts;
				}
				// lane t's anchor to the lanes above it
				const int j = tb + t;
				const int yj = __builtin_amdgcn_readlane((int)A.z, t), rkj = __builtin_amdgcn_readlane(M.x, t);
				const bool before = j < Wn.z;
				const bool out_ok = before & (j >= Wn.x) & (rkj >= M.y) & (rkj <= M.z);
				const bool in_ok = (max_inner > 0) & before & (j >= Wn.y) & (yj <= y_top) & (yj >= y_bot);
				if (__ballot(out_ok | in_ok) == 0) continue;
				const unsigned xj = (unsigned)__builtin_amdgcn_readlane((int)A.x, t);
				const int sj = __builtin_amdgcn_readlane((int)A.w, t) & 0xff, fj = __builtin_amdgcn_readlane(f_l, t);
				int ex2, w2;
				const int s2 = fj + tile_pair_score(xi, yi, xj, yj, sj, P, ex2, w2);
				tile_offer(c, out_ok, (long long)readlane64((unsigned long long)k_l, t), rkj, 0, j, s2, ex2, w2);
				tile_offer_inner(in, in_ok & (w2 <= P.bw), s2, yj, j);
			}
			if (live) { f[i] = f_l; p[i] = p_l; if (strips) sf[sp_i] = f_l; }
			if (b.dbg) { const long long ts4 = (long long)__builtin_amdgcn_s_memrealtime(); d_upd += ts1 - ts0; d_levels += ts2 - ts1; d_qloads += ts3 - ts2; d_t3 += ts4 - ts3; }
			{
				const int top = wave_max_i32(live ? f_l + q_i : INT_MIN);
				if (l == 0) bound[tb >> 6] = top;
			}
			wave_sync();
			// a tie in this tile: the caller redoes the read with the reference's tree whatever comes of the rest -- stop here (abandon_tied)
			if (b.abandon_tied && __ballot(tied != 0) != 0) {
				for (int k = l; k < n; k += W) { f[k] = INT_MIN; p[k] = 0; }   // nothing of this read for the post-pass
				break;
			}
		}
		tied = (int)wave_sum_i32(tied);
		if (l == 0) b.n_tied[r] = tied;
		if (b.dbg && l == 0) {
			const long long v[8] = { n, d_tiles, d_upd, d_levels, d_qloads, d_bcast + (d_single << 36), d_skip + (d_redo << 32), d_t3 };
			for (int q = 0; q < 8; ++q) atomicAdd((unsigned long long*)&b.dbg[q], (unsigned long long)v[q]);
			if (b.dbg_reads) {
				long long *o = b.dbg_reads + 8 * (int64_t)r;
				o[0] = n; o[1] = 1; o[2] = (long long)__builtin_amdgcn_s_memrealtime() - t_read0; o[3] = d_upd; o[4] = d_levels; o[5] = d_qloads; o[6] = d_t3; o[7] = d_bcast;
			}
		}
		wave_sync();
	}
}

// One read by a whole workgroup, round 5.  A tile of a large read used to be a chain -- the tree's update, the queries, the waves' sweeps and
// inner scans, the combination, the tile's own 64 steps -- of which only the sweeps were shared (96 us per tile of a 450 k-anchor read: 0.68 s,
// and a call to the device is over when its longest read is).  Only the 64 steps need the previous tile's scores at once, so the team is SKEWED,
// three roles with a loop of their own each (and registers of their own: one loop with the roles as branches spilled 48 of them):
//   wave 0      the combination and the 64 steps of tile T, nothing else (team_steps);
//   wave 1      the tree, one tile behind (team_tree) -- while wave 0 walks tile T it puts the tree in the state tile T + 1 needs (what leaves
//               before that tile's last window start goes out; what has entered up to the END OF TILE T - 1, all final, goes in), and asks it
//               tile T + 1's queries;
//   waves 2..   for tile T + 1 (team_helper): while wave 0 walks tile T everything that does not need tile T's scores -- the sweep of the anchors
//               leaving the window, the inner scans up to tile T - 1's last anchor (P1) -- and, once tile T is out, its 64 anchors by a broadcast
//               sweep and the end of the inner scans (P2); tile T's anchors are not in the tree for tile T + 1, the sweep offers them like any
//               anchor from `hi` on.
// Per tile: P2, a barrier, { the 64 steps | the tree's update and queries | P1 }, a barrier (measured on the mapper's largest reads: 4 us, then
// 26 | 35 | 47 us -- the helpers' inner scans are what a tile waits for now).  Results per lane meet in LDS (`s_m`, wave 1's and
// the helpers') and are combined by the rules of tile_offer / tile_offer_inner, which do not depend on the order of the offers.
namespace {

constexpr int TEAM_HELPERS = RMQ_TEAM - 2;                   // waves 2 .. RMQ_TEAM - 1

// what the three roles share of a read
struct TeamRead {
	int n, max_inner;
	bool strips;
	double half_gap;
	const uint4 *a, *sa;
	const int4 *meta, *win, *srange;
	int32_t *sf, *f, *p, *bound;
	const int32_t *spos, *ord_idx;
	uint4 *tree;
};
// the tile a wave works FOR: wave 0 the one whose 64 steps come next, the others the one after it
struct TeamTile {
	int tb, n_here, i, yi, q_i, st_first, st_last, stin_first, hi, lo, y_top, y_bot;
	bool live;
	unsigned xi;
	uint4 A;
	int4 M, Wn;
	__device__ __forceinline__ void load(const TeamRead &R, int tb_)
	{
		const int l = lane();
		tb = tb_; n_here = min(W, R.n - tb); i = tb + l; live = l < n_here;
		A = live ? R.a[i] : make_uint4(0, 0, 0, 0);
		M = live ? R.meta[i] : make_int4(0, 1, 0, 0);             // rank, first and last rank of the query (dead lanes: empty)
		Wn = live ? R.win[i] : make_int4(INT_MAX, INT_MAX, 0, 0);   // st, st_inner, i0 (dead lanes: nothing is in reach, nothing came before)
		xi = A.x; yi = (int)A.z; q_i = (int)(A.w & 0xffu);
		st_first = __builtin_amdgcn_readlane(Wn.x, 0); st_last = __builtin_amdgcn_readlane(Wn.x, n_here - 1);
		stin_first = __builtin_amdgcn_readlane(Wn.y, 0);
		hi = max(0, min(__builtin_amdgcn_readlane(Wn.z, 0), tb - W));   // the tree for this tile: [lo, hi) -- nothing of the tile before it
		lo = min(st_last, hi);
		y_top = yi - 1; y_bot = yi - R.max_inner;
	}
};
__device__ __forceinline__ void team_cand_reset(TileCand &c, TileInner &in, int &relied)
{
	c.key = RMQ_NONE; c.rank = -1; c.tie = 0; c.j = -1; c.sc = 0; c.exact = 0; c.width = 0; in.s = 0; in.y = 0; in.j = -1; relied = INT_MIN;
}
__device__ __forceinline__ void team_cand_out(int (*m)[W], const TileCand &c, const TileInner &in, int relied)
{
	const int l = lane();
	m[0][l] = (int)(unsigned)(unsigned long long)c.key; m[1][l] = (int)(unsigned)((unsigned long long)c.key >> 32); m[2][l] = c.rank; m[3][l] = c.tie; m[4][l] = c.j; m[5][l] = c.sc;
	m[6][l] = c.exact; m[7][l] = c.width; m[8][l] = in.s; m[9][l] = in.y; m[10][l] = in.j; m[11][l] = relied;
}
// what every role does when a tile is out: a tie in it and the caller redoes the read with the reference's tree anyway (abandon_tied) -- the
// read's scores are cleared (nothing of it for the post-pass) and the team is through with it
__device__ __forceinline__ bool team_gives_up(const RmqBatch &b, const TeamRead &R, int w, const int *s_flag)
{
	if (!b.abandon_tied || uni(*s_flag) == 0) return false;
	__syncthreads();                                          // (everybody has read the flag before the next read of the team resets it)
	for (int k = w * W + lane(); k < R.n; k += RMQ_TEAM * W) { R.f[k] = INT_MIN; R.p[k] = 0; }
	return true;
}

// ---- wave 0 ----
__device__ __forceinline__ void team_steps(const RmqBatch &b, const RmqParams &P, const TeamRead &R, const int r, int (*s_m)[RMQ_MERGE_WORDS][W], int *s_flag)
{
	const int l = lane(), n = R.n, max_inner = R.max_inner;
	const long long t_read0 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
	long long d_tiles = 0, d_wait = 0, d_t3 = 0, d_redo = 0;
	int tied = 0;
	TeamTile T;
	for (int tb = 0; tb < n; tb += W) {
		const long long tx0 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
		T.load(R, tb);
		const int sp_i = (R.strips && T.live) ? R.spos[T.i] : 0;
		++d_tiles;
		__syncthreads();                                       // every wave's part of this tile is in LDS
		const long long tx1 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
		TileCand c;
		TileInner in;
		int relied;
		team_cand_reset(c, in, relied);
		for (int k = 0; k < RMQ_TEAM - 1; ++k) {
			int (*m)[W] = s_m[k];
			const long long key = (long long)((unsigned long long)(unsigned)m[1][l] << 32 | (unsigned)m[0][l]);
			tile_offer(c, key != RMQ_NONE, key, m[2][l], m[3][l], m[4][l], m[5][l], m[6][l], m[7][l]);
			tile_offer_inner(in, m[10][l] >= 0, m[8][l], m[9][l], m[10][l]);
			relied = max(relied, m[11][l]);
		}
		// ---- the tile's own anchors, one after the other (rmq_fill_read_tiles) ----
		const unsigned xi = T.xi;
		const int yi = T.yi, q_i = T.q_i, i = T.i, y_top = T.y_top, y_bot = T.y_bot;
		const int4 Wn = T.Wn, M = T.M;
		const uint4 A = T.A;
		int f_l = q_i, p_l = 0;
		long long k_l = 0;
		for (int t = 0; t < T.n_here; ++t) {
			const bool has = c.key != RMQ_NONE;
			int max_f = q_i, max_j = -1;
			{ const bool use_out = has & (c.width <= P.bw) & (c.sc > max_f); max_f = use_out ? c.sc : max_f; max_j = use_out ? c.j : max_j; }
			const bool inner_on = has & !c.exact & (max_inner > 0) & (Wn.y < Wn.z) & (yi > 0);
			if (__builtin_amdgcn_readlane((int)(inner_on & (relied > max_f)), t) != 0) {
				// lane t passed blocks over that its final outer result does not rule out: its whole inner window before the tile again
				++d_redo;
				const unsigned xt = (unsigned)__builtin_amdgcn_readlane((int)xi, t);
				const int yt = __builtin_amdgcn_readlane(yi, t), from_t = __builtin_amdgcn_readlane(Wn.y, t), to_t = min(__builtin_amdgcn_readlane(Wn.z, t), tb);
				TileInner best; best.s = 0; best.y = 0; best.j = -1;
				for (int base = from_t; base < to_t; base += W) {
					const int j = base + l;
					if (j < to_t) {
						const uint4 e = R.a[j];
						int ex2, w2;
						const int s2 = R.f[j] + tile_pair_score(xt, yt, e.x, (int)e.z, (int)(e.w & 0xffu), P, ex2, w2);
						tile_offer_inner(best, ((int)e.z <= yt - 1) & ((int)e.z >= yt - max_inner) & (w2 <= P.bw), s2, (int)e.z, j);
					}
				}
				for (int o = W / 2; o > 0; o >>= 1) tile_offer_inner(best, __shfl_xor(best.j, o) >= 0, __shfl_xor(best.s, o), __shfl_xor(best.y, o), __shfl_xor(best.j, o));
				if (l == t) tile_offer_inner(in, best.j >= 0, best.s, best.y, best.j);
			}
			const bool use_in = inner_on & (in.j >= 0) & (in.s > max_f);
			max_f = use_in ? in.s : max_f; max_j = use_in ? in.j : max_j;
			bool counts = __builtin_amdgcn_readlane((int)(has && c.tie), t) != 0;
			if (counts && P.weigh_ties)
				counts = rmq_tie_decides(R.a, R.f, R.meta, R.half_gap, P, max_inner, (unsigned)__builtin_amdgcn_readlane((int)xi, t), __builtin_amdgcn_readlane(yi, t), __builtin_amdgcn_readlane(q_i, t),
				                         (long long)readlane64((unsigned long long)c.key, t), __builtin_amdgcn_readlane(M.y, t), __builtin_amdgcn_readlane(M.z, t),
				                         __builtin_amdgcn_readlane(Wn.x, t), __builtin_amdgcn_readlane(Wn.y, t), __builtin_amdgcn_readlane(Wn.z, t),
				                         __builtin_amdgcn_readlane(max_f, t), __builtin_amdgcn_readlane(max_j, t), tb, t, A, f_l, k_l, M.x);
			if (l == t) {
				f_l = max_f; p_l = max_j < 0 ? 0 : i - max_j;
				k_l = key_order((double)max_f + R.half_gap * (double)((int)xi + yi));
				tied += counts;
			}
			// lane t's anchor to the lanes above it
			const int j = tb + t;
			const int yj = __builtin_amdgcn_readlane((int)A.z, t), rkj = __builtin_amdgcn_readlane(M.x, t);
			const bool before = j < Wn.z;
			const bool out_ok = before & (j >= Wn.x) & (rkj >= M.y) & (rkj <= M.z);
			const bool in_ok = (max_inner > 0) & before & (j >= Wn.y) & (yj <= y_top) & (yj >= y_bot);
			if (__ballot(out_ok | in_ok) == 0) continue;
			const unsigned xj = (unsigned)__builtin_amdgcn_readlane((int)A.x, t);
			const int sj = __builtin_amdgcn_readlane((int)A.w, t) & 0xff, fj = __builtin_amdgcn_readlane(f_l, t);
			int ex2, w2;
			const int s2 = fj + tile_pair_score(xi, yi, xj, yj, sj, P, ex2, w2);
			tile_offer(c, out_ok, (long long)readlane64((unsigned long long)k_l, t), rkj, 0, j, s2, ex2, w2);
			tile_offer_inner(in, in_ok & (w2 <= P.bw), s2, yj, j);
		}
		if (T.live) { R.f[i] = f_l; R.p[i] = p_l; if (R.strips) R.sf[sp_i] = f_l; }
		{
			const int top = wave_max_i32(T.live ? f_l + q_i : INT_MIN);
			if (l == 0) R.bound[tb >> 6] = top;
		}
		if (b.dbg) { const long long tx2 = (long long)__builtin_amdgcn_s_memrealtime(); d_wait += tx1 - tx0; d_t3 += tx2 - tx1; }
		if (b.abandon_tied) { const bool any = __ballot(tied != 0) != 0; if (l == 0) *s_flag = any; }
		__threadfence_block();
		__syncthreads();                                       // the tile's scores and its bound are out, the tree stands for the next tile
		if (team_gives_up(b, R, 0, s_flag)) break;
	}
	tied = (int)wave_sum_i32(tied);
	if (l == 0) b.n_tied[r] = tied;
	if (b.dbg && l == 0) {
		const long long v[8] = { n, d_tiles, 0, 0, d_wait, 0, d_redo << 32, d_t3 };
		for (int q = 0; q < 8; ++q) if (v[q]) atomicAdd((unsigned long long*)&b.dbg[q], (unsigned long long)v[q]);
		if (b.dbg_reads) {
			long long *o = b.dbg_reads + 8 * (int64_t)r;
			o[0] = n; o[1] = RMQ_TEAM; o[2] = (long long)__builtin_amdgcn_s_memrealtime() - t_read0; o[5] = d_wait; o[6] = d_t3; o[7] = 0;
		}
	}
}

// ---- wave 1 ----
__device__ __forceinline__ void team_tree(const RmqBatch &b, const RmqParams &P, const TeamRead &R, const int r, int (*s_m)[RMQ_MERGE_WORDS][W], uint4 *s_top, const int *s_flag)
{
	const int l = lane(), n = R.n;
	constexpr int top_nodes = RMQ_TOP_NODES > 1 ? RMQ_TOP_NODES : 0;     // nodes below this index live in s_top for the read's time
	uint4 *tree = R.tree;
	auto tld = [&](int q) -> uint4 { if (q < top_nodes) return s_top[q]; return tree[q]; };
	auto tst = [&](int q, const uint4 &v) { if (q < top_nodes) s_top[q] = v; else tree[q] = v; };
	int ev = 0, ins = 0;                                       // the tree holds the anchors of index [ev, ins)
	long long d_upd = 0, d_levels = 0;
	TeamTile T;
	TileCand c;
	TileInner in;
	int relied;
	team_cand_reset(c, in, relied);                            // (tile 0 has nothing before it: nothing in the tree to ask for)
	for (int tb = 0; tb < n; tb += W) {
		team_cand_out(s_m[0], c, in, relied);                     // tile tb's queries, answered beside the previous tile's 64 steps
		__syncthreads();
		if (tb + W < n) {
			// ---- the tree for tile tb + W: out with [ev, min(lo, ins)), in with [max(ins, lo), hi) (rmq_fill_read_tiles) ----
			const long long tu0 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
			T.load(R, tb + W);
			const int lo = T.lo, hi = T.hi;
			int e0 = ev, n0 = max(ins, lo);
			const int e1 = min(lo, ins), n1 = hi;
			while (e0 < e1 || n0 < n1) {
				const int je = e0 + l, ji = n0 + l;
				int pe = 0, pi = 0;
				if (je < e1) { pe = n + R.meta[je].x; tst(pe, tnode_none()); }
				if (ji < n1) {
					const uint4 e = R.a[ji];
					const int rk = R.meta[ji].x;
					const long long kk = key_order((double)R.f[ji] + R.half_gap * (double)((int)e.x + (int)e.z));
					pi = n + rk;
					tst(pi, make_uint4((unsigned)kk, (unsigned)((unsigned long long)kk >> 32), (unsigned)rk, 0u));
				}
				wave_sync();
				while (__ballot(pe > 1 || pi > 1) != 0) {
					uint4 c0 = tnode_none(), c1 = c0, c2 = c0, c3 = c0;
					if (pe > 1) { pe >>= 1; c0 = tld(2 * pe); c1 = tld(2 * pe + 1); } else pe = 0;
					if (pi > 1) { pi >>= 1; c2 = tld(2 * pi); c3 = tld(2 * pi + 1); } else pi = 0;
					if (pe > 0) tst(pe, tnode_comb(c0, c1));
					if (pi > 0 && pi != pe) tst(pi, tnode_comb(c2, c3));
					wave_sync();
				}
				e0 += W; n0 += W;
			}
			ev = lo; ins = hi;
			// ---- every lane's query of it, bottom up; the loads do not depend on each other ----
			const long long tq0 = b.dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
			team_cand_reset(c, in, relied);
			uint4 best = tnode_none();
			int ql = n + T.M.y, qr = n + T.M.z + 1;                         // [ql, qr) over the leaves
			const bool go = T.live && T.M.y <= T.M.z && lo < hi;
			while (__ballot(go && ql < qr) != 0) {
				uint4 vl = tnode_none(), vr = vl;
				if (go && ql < qr) {
					if (ql & 1) vl = tld(ql++);
					if (qr & 1) vr = tld(--qr);
					ql >>= 1; qr >>= 1;
				}
				best = tnode_comb(best, tnode_comb(vl, vr));
			}
			if (tnode_key(best) != RMQ_NONE) {
				const int rk = (int)(best.z & 0x7fffffffu);
				const int j = R.ord_idx[rk];
				const uint4 e = R.a[j];
				int ex, wd;
				const int sc = R.f[j] + tile_pair_score(T.xi, T.yi, e.x, (int)e.z, (int)(e.w & 0xffu), P, ex, wd);
				c.key = tnode_key(best); c.rank = rk; c.tie = (int)(best.z >> 31); c.j = j; c.sc = sc; c.exact = ex; c.width = wd;
			}
			if (b.dbg) { d_upd += tq0 - tu0; d_levels += (long long)__builtin_amdgcn_s_memrealtime() - tq0; }
		}
		__threadfence_block();
		__syncthreads();
		if (team_gives_up(b, R, 1, s_flag)) break;
	}
	if (b.dbg && l == 0) {
		atomicAdd((unsigned long long*)&b.dbg[2], (unsigned long long)d_upd);
		atomicAdd((unsigned long long*)&b.dbg[3], (unsigned long long)d_levels);
		if (b.dbg_reads) { b.dbg_reads[8 * (int64_t)r + 3] = d_upd; b.dbg_reads[8 * (int64_t)r + 4] = d_levels; }
	}
}

// ---- waves 2 .. ----
__device__ __forceinline__ void team_helper(const RmqBatch &b, const RmqParams &P, const TeamRead &R, const int w, int (*s_m)[RMQ_MERGE_WORDS][W], const int *s_flag)
{
	constexpr int NH = TEAM_HELPERS;
	const int l = lane(), h = w - 2, n = R.n, max_inner = R.max_inner;
	const bool strips = R.strips;
	const uint4 *a = R.a;
	const int32_t *f = R.f;
	long long d_bcast = 0, d_single = 0, d_skip = 0;
	TeamTile T;
	TileCand c;
	TileInner in;
	int relied;
	int spp[2] = { 0, 0 }, spe[2] = { 0, 0 };                  // this wave's part of the lane's two strip ranges: where the walk stands, where it ends
	// anchors broadcast one by one to all lanes (rmq_fill_read_tiles); in_turn: the helpers take eighths of a block of 64 in turn
	auto sweep_range = [&](int from, int to, bool outer_all, bool in_turn, int blk0) {
		for (int base = from; base < to; base += W) {
			const int blk = blk0 + ((base - from) >> 6);
			if (in_turn && ((h - blk * 8) % NH + NH) % NH >= 8) continue;    // none of this block's eight turns is this wave's
			const int j_l = base + l;
			const bool have = j_l < to;
			const uint4 e_l = have ? a[j_l] : make_uint4(0, 0, 0, 0);
			const int f_l = have ? f[j_l] : 0, rk_l = have ? R.meta[j_l].x : 0;
			const long long k_l = key_order((double)f_l + R.half_gap * (double)((int)e_l.x + (int)e_l.z));
			const int cnt = min(W, to - base);
			for (int k = 0; k < cnt; k += 2) {
				if (in_turn && (blk * 8 + (k >> 3)) % NH != h) continue;
				const int ja = base + k, jb2 = ja + 1;
				const bool two = k + 1 < cnt;
				const int kb = two ? k + 1 : k;
				const bool outer_a = outer_all || ja >= T.hi, outer_b = two && (outer_all || jb2 >= T.hi);     // wave-uniform: in the tree otherwise
				const bool inner_a = !strips && max_inner > 0 && ja >= T.stin_first, inner_b = !strips && two && max_inner > 0 && jb2 >= T.stin_first;   // wave-uniform
				d_bcast += two ? 2 : 1;
				if (!(outer_a || outer_b || inner_a || inner_b)) continue;
				const int ya = __builtin_amdgcn_readlane((int)e_l.z, k), yb = __builtin_amdgcn_readlane((int)e_l.z, kb);
				const int rka = __builtin_amdgcn_readlane(rk_l, k), rkb = __builtin_amdgcn_readlane(rk_l, kb);
				const bool out_a = outer_a & (ja < T.Wn.z) & (ja >= T.Wn.x) & (rka >= T.M.y) & (rka <= T.M.z);
				const bool out_b = outer_b & (jb2 < T.Wn.z) & (jb2 >= T.Wn.x) & (rkb >= T.M.y) & (rkb <= T.M.z);
				const bool in_a = inner_a & (ja < T.Wn.z) & (ja >= T.Wn.y) & (ya <= T.y_top) & (ya >= T.y_bot);
				const bool in_b = inner_b & (jb2 < T.Wn.z) & (jb2 >= T.Wn.y) & (yb <= T.y_top) & (yb >= T.y_bot);
				const unsigned long long any_out = __ballot(out_a | out_b);
				if ((any_out | __ballot(in_a | in_b)) == 0) continue;
				const unsigned xa = (unsigned)__builtin_amdgcn_readlane((int)e_l.x, k), xb = (unsigned)__builtin_amdgcn_readlane((int)e_l.x, kb);
				const int sa2 = __builtin_amdgcn_readlane((int)e_l.w, k) & 0xff, sb2 = __builtin_amdgcn_readlane((int)e_l.w, kb) & 0xff;
				const int fa = __builtin_amdgcn_readlane(f_l, k), fb = __builtin_amdgcn_readlane(f_l, kb);
				int exa, wa, exb, wb;
				const int s2a = fa + tile_pair_score(T.xi, T.yi, xa, ya, sa2, P, exa, wa);     // the same pair score serves the outer query and the inner scan
				const int s2b = fb + tile_pair_score(T.xi, T.yi, xb, yb, sb2, P, exb, wb);
				if (any_out != 0) {
					tile_offer(c, out_a, (long long)readlane64((unsigned long long)k_l, k), rka, 0, ja, s2a, exa, wa);
					tile_offer(c, out_b, (long long)readlane64((unsigned long long)k_l, kb), rkb, 0, jb2, s2b, exb, wb);
				}
				tile_offer_inner(in, in_a & (wa <= P.bw), s2a, ya, ja);
				tile_offer_inner(in, in_b & (wb <= P.bw), s2b, yb, jb2);
			}
		}
	};
	// the blocks of 64 from bb_top down to `from` (rmq_fill_read_tiles: bounds, lanes one at a time where few need a block)
	auto sweep_blocks = [&](int bb_top, int from) {
		for (int bb = bb_top; bb >= 0 && ((bb + 1) << 6) > from; --bb) {
			const int b_lo = max(bb << 6, from), b_hi = (bb + 1) << 6;
			if (b_lo >= T.hi || max_inner <= 0) { sweep_range(b_lo, b_hi, false, true, bb); continue; }     // holds anchors the outer query needs (not in the tree): no skipping
			if (bb % NH != h) continue;
			if (b_hi > T.hi) { sweep_range(b_lo, b_hi, false, false, bb); continue; }
			const int bnd = uni(R.bound[bb]);
			const bool sure = !T.live || bnd <= T.q_i || (in.j >= 0 && bnd < in.s) || b_hi <= T.Wn.y || b_lo >= T.Wn.z;
			const int spec = (c.key != RMQ_NONE && c.width <= P.bw) ? c.sc : INT_MIN;
			const bool idle = sure || bnd <= spec;
			const unsigned long long need = __ballot(!idle);
			if (need == 0) { ++d_skip; relied = (!sure && bnd > relied) ? bnd : relied; continue; }
			if (__popcll(need) >= 24) { sweep_range(b_lo, b_hi, false, false, bb); continue; }
			relied = (idle && !sure && bnd > relied) ? bnd : relied;
			const int j_c = (bb << 6) + l;
			const bool have = j_c >= b_lo && j_c < b_hi;
			const uint4 e_c = have ? a[j_c] : make_uint4(0, 0, 0, 0);
			const int f_c = have ? f[j_c] : 0;
			for (unsigned long long m = need; m != 0; m &= m - 1) {
				const int u = first_set(m);
				const unsigned xu = (unsigned)__builtin_amdgcn_readlane((int)T.xi, u);
				const int yu = __builtin_amdgcn_readlane(T.yi, u), from_u = __builtin_amdgcn_readlane(T.Wn.y, u), to_u = __builtin_amdgcn_readlane(T.Wn.z, u);
				int ex2, w2;
				const int s2 = f_c + tile_pair_score(xu, yu, e_c.x, (int)e_c.z, (int)(e_c.w & 0xffu), P, ex2, w2);
				const bool ok = have & (j_c >= from_u) & (j_c < to_u) & ((int)e_c.z <= yu - 1) & ((int)e_c.z >= yu - max_inner) & (w2 <= P.bw);
				const unsigned long long oks = __ballot(ok);
				++d_single;
				if (oks == 0) continue;
				const int bs = wave_max_i32(ok ? s2 : INT_MIN);
				const int by = wave_max_i32((ok & (s2 == bs)) ? (int)e_c.z : INT_MIN);
				const int bj = wave_max_i32((ok & (s2 == bs) & ((int)e_c.z == by)) ? j_c : -1);
				if (l == u) tile_offer_inner(in, true, bs, by, bj);
			}
		}
	};
	// the inner window lane by lane over the (strip, index) order, this wave's part of either range, as far as index `upto` (indices rise along a
	// range: the walk stops at the first candidate that is not final yet and goes on from there in the next phase)
	auto strips_walk = [&](int upto) {
		constexpr int G = 4;                                   // candidates per round: a round is one round trip to memory, and the wave waits it out
#pragma unroll
		for (int pass = 0; pass < 2; ++pass) {
			bool act = spp[pass] < spe[pass];
			while (__ballot(act) != 0) {
				if (act) {
					const int pp = spp[pass], pe = spe[pass];
					uint4 cq[G];
					int fq[G];
#pragma unroll
					for (int q = 0; q < G; ++q) { const int at = min(pp + q, pe - 1); cq[q] = R.sa[at]; fq[q] = R.sf[at]; }
					int adv = 0;
					bool run = true;
#pragma unroll
					for (int q = 0; q < G; ++q) {
						const bool ok = run & (pp + q < pe) & ((int)cq[q].z < upto);
						int ex, wd;
						const int s2 = fq[q] + tile_pair_score(T.xi, T.yi, cq[q].x, (int)cq[q].y, (int)cq[q].w, P, ex, wd);
						tile_offer_inner(in, ok & ((int)cq[q].y <= T.y_top) & ((int)cq[q].y >= T.y_bot) & (wd <= P.bw), s2, (int)cq[q].y, (int)cq[q].z);
						adv += ok; run = ok;
					}
					++d_single;
					spp[pass] = pp + adv;
					act = (adv == G) & (pp + G < pe);
				}
			}
		}
	};
	auto next_tile = [&](int tb) {
		T.load(R, tb);
		team_cand_reset(c, in, relied);
		if (strips) {
			const int4 R4 = T.live ? R.srange[T.i] : make_int4(0, 0, 0, 0);
#pragma unroll
			for (int pass = 0; pass < 2; ++pass) {
				const int r_lo = pass ? R4.z : R4.x, r_hi = pass ? R4.w : R4.y;
				const int part = (r_hi - r_lo + NH - 1) / NH;   // a helper's part of the lane's range
				spp[pass] = r_lo + h * part; spe[pass] = min(r_hi, spp[pass] + part);
			}
		}
	};
	next_tile(0);                                             // tile 0 has nothing before it
	for (int tb = 0; tb < n; tb += W) {
		// P2: the tile that has just been finished -- its anchors for the outer query (and the inner window where it is swept), the inner scans' end
		if (tb > 0) sweep_range(tb - W, tb, false, true, (tb >> 6) - 1);
		if (strips) strips_walk(tb);
		team_cand_out(s_m[w - 1], c, in, relied);
		__syncthreads();
		if (tb + W < n) {
			// P1 for the next tile, beside wave 0's 64 steps: what does not need this tile's scores
			next_tile(tb + W);
			sweep_range(T.st_first, T.lo, true, true, 0);
			// (everything from `hi` on is outside the tree and swept for the outer query, wherever the inner windows start)
			const int from = max((max_inner > 0 && !strips) ? min(min(T.stin_first, T.tb), T.hi) : T.hi, T.lo);
			sweep_blocks((T.tb >> 6) - 2, from);
			if (strips) strips_walk(tb);
		}
		__threadfence_block();
		__syncthreads();
		if (team_gives_up(b, R, w, s_flag)) break;
	}
	if (b.dbg && l == 0) atomicAdd((unsigned long long*)&b.dbg[5], (unsigned long long)(d_bcast + (d_single << 36)));
	if (b.dbg && l == 0 && d_skip) atomicAdd((unsigned long long*)&b.dbg[6], (unsigned long long)d_skip);
}

} // namespace

__device__ __forceinline__ void rmq_fill_read_team(const RmqBatch &b, const RmqParams &P, const int r, const int w, int (*s_m)[RMQ_MERGE_WORDS][W], uint4 *s_top, int *s_flag)
{
	TeamRead R;
	const int max_dist = P.max_dist < P.bw ? P.bw : P.max_dist;                                       // lchain.c:264
	R.max_inner = (P.max_dist_inner <= 0 || P.max_dist_inner >= max_dist) ? 0 : P.max_dist_inner;      // lchain.c:265
	R.half_gap = 0.5 * (double)P.pen_gap;
	const int64_t off = b.offsets[r];
	R.n = (int)(b.offsets[r + 1] - off);
	R.a = b.raw + off; R.meta = b.meta + off; R.win = b.win + off;
	R.strips = b.sa != nullptr && R.max_inner > 0;
	R.sa = b.sa + off; R.srange = b.srange + off;
	R.sf = (int32_t*)b.skey_in + off;                         // the scores in strip order, and every anchor's place in it (rmq_fill_read_tiles)
	R.spos = (const int32_t*)b.skey_in + b.n + off;
	R.ord_idx = b.ord_idx + off;
	R.f = b.f + off; R.p = b.p + off;
	R.tree = b.tree + 2 * off;
	R.bound = b.bound + (off >> 6) + r;
	if (w == 0) for (int q = lane(); q < (RMQ_TOP_NODES > 1 ? RMQ_TOP_NODES : 0); q += W) s_top[q] = tnode_none();
	__syncthreads();
	if (w == 0) team_steps(b, P, R, r, s_m, s_flag);
	else if (w == 1) team_tree(b, P, R, r, s_m, s_top, s_flag);
	else team_helper(b, P, R, w, s_m, s_flag);
	wave_sync();
}

// The first b.n_team reads of the batch (the caller puts the most expensive first) are a whole workgroup's each, the rest one wave's.
__global__ __launch_bounds__(RMQ_THREADS) void k_rmq_fill_tiles(RmqBatch b, RmqParams P)
{
	__shared__ int s_m[RMQ_TEAM - 1][RMQ_MERGE_WORDS][W];
	__shared__ uint4 s_top[RMQ_TOP_NODES];
	__shared__ int s_read, s_flag;
	const int l = lane(), w = uni(threadIdx.x / W);
	const int n_team = (int)min((int64_t)b.n_team, b.n_reads);
	for (;;) {
		if (n_team <= 0) break;
		if (threadIdx.x == 0) s_read = atomicAdd(b.cursor + 1, 1);
		__syncthreads();
		const int r = uni(s_read);
		__syncthreads();
		if (r >= n_team) break;
		rmq_fill_read_team(b, P, r, w, s_m, s_top, &s_flag);
	}
	for (;;) {
		int r = 0;
		if (l == 0) r = atomicAdd(b.cursor, 1);
		r = uni(r) + n_team;
		if (r >= b.n_reads) break;
		rmq_fill_read_tiles(b, P, r);
	}
}

// --------------------------------------------------------------------------------------------------------------
// Either side of the path (N4).  k_sort_x: the seed sort of collect_seed_hits (map.c:329), one wave per read, the same exact
// device form of radix_sort_128x that orders the chains.  k_gen_regs: mm_gen_regs (hit.c:52-88): chains ordered by
// (score, hash of the first anchor) with the same sort, best first, then coordinates and fuzzy lengths (hit.c:8-38), one lane
// per chain.
// --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(POST_THREADS) void k_sort_x(SortBatch b)
{
	__shared__ PassLds lds[POST_THREADS / W];
	PassLds &L = lds[threadIdx.x / W];
	const int per = POST_THREADS / W;
	for (int64_t r = (int64_t)blockIdx.x * per + uni(threadIdx.x / W); r < b.n_reads; r += (int64_t)gridDim.x * per) {
		const int64_t off = b.offsets[r];
		sort_like_host<HElem>(b.a + off, (int)(b.offsets[r + 1] - off), L);
	}
}

// --------------------------------------------------------------------------------------------------------------
// Seed matches -> anchors (collect_seed_hits, map.c:295-331).  k_seed_reads: which read a seed belongs to.  k_seed_expand: one
// thread per hit -- its seed by bisection of hit_off, the tests of skip_seed (map.c:205-227), the anchor (map.c:311-324) written at
// the hit's own position.  k_seed_compact: one wave per read closes the gaps of dropped hits (order kept: the sort that follows is
// not stable and depends on it) and counts; k_seed_offsets: exclusive scan of the counts; k_seed_pack: the reads back to back.
// --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_seed_reads(SeedBatch b)
{
	for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < b.n_seeds; k += (int64_t)gridDim.x * blockDim.x) {
		int64_t lo = 0, hi = b.n_reads;                   // invariant: seed_off[lo] <= k < seed_off[hi]
		while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (b.seed_off[mid] <= k) lo = mid; else hi = mid; }
		b.seed_read[k] = (int32_t)lo;
	}
}

__global__ __launch_bounds__(256) void k_seed_expand(SeedBatch b)
{
	constexpr long long F_NO_DIAG = 0x001, F_NO_DUAL = 0x002, F_FOR_ONLY = 0x100000, F_REV_ONLY = 0x200000, F_QSTRAND = 0x100000000LL;   // minimap.h:8-9,28-29,40
	for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < b.n_hits; h += (int64_t)gridDim.x * blockDim.x) {
		int64_t lo = 0, hi = b.n_seeds;                   // invariant: hit_off[lo] <= h < hit_off[hi]  (seeds without hits are passed over)
		while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (b.hit_off[mid] <= h) lo = mid; else hi = mid; }
		const SeedRecord q = b.seeds[lo];
		const int rd = b.seed_read[lo];
		const int qlen = b.qlen[rd];
		const unsigned long long r = b.hits[h];
		const unsigned q_span = q.span_flt & 0x7fffffffu, seg_id = q.seg_tandem & 0x7fffffffu;
		const int rpos = (int)((unsigned)r >> 1);
		const bool same_strand = (r & 1) == (q.q_pos & 1);
		bool skip = false, is_self = false;
		if (b.q_rank && (b.flag & (F_NO_DIAG | F_NO_DUAL))) {                          // map.c:208-219
			const int rid = (int)(r >> 32), qr = b.q_rank[rd], rr = b.ref_rank[rid];
			if ((b.flag & F_NO_DIAG) && qr == rr && b.ref_len[rid] == qlen) {
				if ((unsigned)r >> 1 == (q.q_pos >> 1)) skip = true;
				else if (same_strand) is_self = true;
			}
			if (!skip && (b.flag & F_NO_DUAL) && qr > rr) skip = true;
		}
		if (!skip && (b.flag & (F_FOR_ONLY | F_REV_ONLY))) {                            // map.c:220-226
			if (same_strand) skip = (b.flag & F_REV_ONLY) != 0;
			else skip = (b.flag & F_FOR_ONLY) != 0;
		}
		ulonglong2 a;
		if (skip) a = make_ulonglong2(~0ull, ~0ull);
		else {
			if (same_strand) {                                                          // map.c:311-313
				a.x = (r & 0xffffffff00000000ULL) | (unsigned)rpos;
				a.y = (unsigned long long)q_span << 32 | q.q_pos >> 1;
			} else if (!(b.flag & F_QSTRAND)) {                                         // map.c:314-316
				a.x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (unsigned)rpos;
				a.y = (unsigned long long)q_span << 32 | (unsigned)(qlen - (int)((q.q_pos >> 1) + 1 - q_span) - 1);
			} else {                                                                    // map.c:317-321
				const int len = b.ref_len[r >> 32];
				a.x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (unsigned)(len - (rpos + 1 - (int)q_span) - 1);
				a.y = (unsigned long long)q_span << 32 | q.q_pos >> 1;
			}
			a.y |= (unsigned long long)seg_id << 48;                                    // MM_SEED_SEG_SHIFT
			if (q.seg_tandem >> 31) a.y |= 1ULL << 42;                                  // MM_SEED_TANDEM
			if (is_self) a.y |= 1ULL << 43;                                             // MM_SEED_SELF
		}
		b.tmp[h] = a;
	}
}

__global__ __launch_bounds__(POST_THREADS) void k_seed_compact(SeedBatch b)
{
	const int per = POST_THREADS / W, l = lane();
	for (int64_t r = (int64_t)blockIdx.x * per + uni(threadIdx.x / W); r < b.n_reads; r += (int64_t)gridDim.x * per) {
		const int64_t h0 = b.hit_off[b.seed_off[r]], h1 = b.hit_off[b.seed_off[r + 1]];
		int64_t at = h0;
		for (int64_t base = h0; base < h1; base += W) {
			const int64_t i = base + l;
			const bool in = i < h1;
			const ulonglong2 a = b.tmp[in ? i : h0];
			const bool keep = in && !(a.x == ~0ull && a.y == ~0ull);
			const unsigned long long m = __ballot(keep);
			wave_sync();                                  // every lane has read its element before any lane writes (at <= base)
			if (keep) b.tmp[at + __popcll(m & ((1ull << l) - 1))] = a;
			at += __popcll(m);
			wave_sync();
		}
		if (l == 0) b.n_kept[r] = (int32_t)(at - h0);
	}
}

__global__ __launch_bounds__(1024) void k_seed_offsets(SeedBatch b)
{
	__shared__ long long s_tmp[1024 / W];
	long long carry = 0;
	const int w = threadIdx.x / W, l = lane();
	for (int64_t base = 0; base < b.n_reads; base += 1024) {
		const int64_t r = base + threadIdx.x;
		const long long v = r < b.n_reads ? b.n_kept[r] : 0;
		long long inc = v;
		for (int off = 1; off < W; off <<= 1) { const long long o = __shfl_up(inc, off); if (l >= off) inc += o; }
		__syncthreads();
		if (l == W - 1) s_tmp[w] = inc;
		__syncthreads();
		long long before = 0, total = 0;
		for (int k = 0; k < 1024 / W; ++k) { if (k < w) before += s_tmp[k]; total += s_tmp[k]; }
		if (r < b.n_reads) b.anchor_off[r] = carry + before + inc - v;
		carry += total;
	}
	if (threadIdx.x == 0) b.anchor_off[b.n_reads] = carry;
}

__global__ __launch_bounds__(POST_THREADS) void k_seed_pack(SeedBatch b)
{
	const int per = POST_THREADS / W, l = lane();
	for (int64_t r = (int64_t)blockIdx.x * per + uni(threadIdx.x / W); r < b.n_reads; r += (int64_t)gridDim.x * per) {
		const int64_t h0 = b.hit_off[b.seed_off[r]], o0 = b.anchor_off[r];
		const int n = b.n_kept[r];
		for (int j = l; j < n; j += W) b.out[o0 + j] = b.tmp[h0 + j];
	}
}

void launch_collect_seeds(const SeedBatch &b, hipStream_t s)
{
	if (b.n_reads <= 0) return;
	const int per = POST_THREADS / W;
	const unsigned rgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((b.n_reads + per - 1) / per, ((int64_t)b.grid_waves + per - 1) / per));
	if (b.n_seeds > 0) hipLaunchKernelGGL(k_seed_reads, dim3((unsigned)std::min<int64_t>((b.n_seeds + 255) / 256, 65536)), dim3(256), 0, s, b);
	if (b.n_hits > 0) hipLaunchKernelGGL(k_seed_expand, dim3((unsigned)std::min<int64_t>((b.n_hits + 255) / 256, 262144)), dim3(256), 0, s, b);
	hipLaunchKernelGGL(k_seed_compact, dim3(rgrid), dim3(POST_THREADS), 0, s, b);
	hipLaunchKernelGGL(k_seed_offsets, dim3(1), dim3(1024), 0, s, b);
	hipLaunchKernelGGL(k_seed_pack, dim3(rgrid), dim3(POST_THREADS), 0, s, b);
	SortBatch sb;
	sb.a = b.out; sb.offsets = b.anchor_off; sb.n_reads = b.n_reads; sb.cursor = nullptr; sb.grid_waves = b.grid_waves;
	launch_sort_x(sb, s);
}

namespace {
__device__ __forceinline__ unsigned long long hit_hash64(unsigned long long key)   // hit.c:40-50
{
	key = (~key + (key << 21));
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8));
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4));
	key = key ^ key >> 28;
	key = (key + (key << 31));
	return key;
}
} // namespace

__global__ __launch_bounds__(POST_THREADS) void k_gen_regs(RegBatch b)
{
	__shared__ PassLds lds[POST_THREADS / W];
	PassLds &L = lds[threadIdx.x / W];
	const int l = lane();
	const int per = POST_THREADS / W;
	for (int64_t r = (int64_t)blockIdx.x * per + uni(threadIdx.x / W); r < b.n_reads; r += (int64_t)gridDim.x * per) {
		const int64_t uo = b.u_off[r];
		const int n_u = (int)(b.u_off[r + 1] - uo);
		if (n_u == 0) continue;
		const unsigned long long *u = b.u + uo;
		const uint4 *a = b.a + b.a_off[r];
		ulonglong2 *z = b.z + uo;
		RegRecord *regs = b.regs + uo;
		const unsigned hash = b.hash[r];
		const int qlen = b.qlen[r];
		// hit.c:63-69: key = chain record with the hash of its first anchor folded into the low half; value = offset << 32 | count
		int k_at = 0;
		for (int base = 0; base < n_u; base += W) {
			const int c = base + l;
			const unsigned long long uc = c < n_u ? u[c] : 0;
			const int cnt = (int)(unsigned)uc;
			int inc = cnt;
			for (int o = 1; o < W; o <<= 1) { const int v = __shfl_up(inc, o); if (l >= o) inc += v; }
			const int k0 = k_at + inc - cnt;
			if (c < n_u) {
				const uint4 f0 = a[k0];
				const unsigned long long ax = (unsigned long long)f0.y << 32 | f0.x, ay = (unsigned long long)f0.w << 32 | f0.z;
				const unsigned h = (unsigned)hit_hash64((hit_hash64(ax) + hit_hash64(ay)) ^ hash);
				z[c] = make_ulonglong2(uc ^ h, (unsigned long long)(unsigned)k0 << 32 | (unsigned)cnt);
			}
			k_at += __builtin_amdgcn_readlane(inc, W - 1);
		}
		wave_sync();
		sort_like_host<HElem>(z, n_u, L);
		wave_sync();
		// hit.c:70-86 (largest key first), 22-38, 8-20
		for (int i = l; i < n_u; i += W) {
			const ulonglong2 zi = z[n_u - 1 - i];
			const int as = (int)(zi.y >> 32), cnt = (int)(unsigned)zi.y;
			RegRecord ri = {};
			ri.id = i; ri.parent = -1;
			ri.score = ri.score0 = (int)(zi.x >> 32);
			ri.hash = (unsigned)zi.x;
			ri.cnt = cnt; ri.as = as; ri.div = -1.0f;
			const uint4 first = a[as], last = a[as + cnt - 1];
			const int q_span = (int)(first.w & 0xffu);
			const bool rev = first.y >> 31;
			if (rev) ri.flags |= 1u << 10;
			ri.rid = (int)(first.y & 0x7fffffffu);
			ri.rs = (int)first.x + 1 > q_span ? (int)first.x + 1 - q_span : 0;
			ri.re = (int)last.x + 1;
			if (!rev || b.is_qstrand) { ri.qs = (int)first.z + 1 - q_span; ri.qe = (int)last.z + 1; }
			else { ri.qs = qlen - ((int)last.z + 1); ri.qe = qlen - ((int)first.z + 1 - q_span); }
			int mlen = q_span, blen = q_span;
			uint4 prev = first;
			for (int j = as + 1; j < as + cnt; ++j) {
				const uint4 cur = a[j];
				const int span = (int)(cur.w & 0xffu), tl = (int)cur.x - (int)prev.x, ql = (int)cur.z - (int)prev.z;
				blen += tl > ql ? tl : ql;
				mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
				prev = cur;
			}
			ri.mlen = cnt > 0 ? mlen : 0; ri.blen = cnt > 0 ? blen : 0;
			regs[i] = ri;
		}
		wave_sync();
	}
}

void launch_sort_x(const SortBatch &b, hipStream_t s)
{
	if (b.n_reads <= 0) return;
	const int per = POST_THREADS / W;
	const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((b.n_reads + per - 1) / per, ((int64_t)b.grid_waves + per - 1) / per));
	hipLaunchKernelGGL(k_sort_x, dim3(grid), dim3(POST_THREADS), 0, s, b);
}

void launch_gen_regs(const RegBatch &b, hipStream_t s)
{
	if (b.n_reads <= 0) return;
	const int per = POST_THREADS / W;
	const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((b.n_reads + per - 1) / per, ((int64_t)b.grid_waves + per - 1) / per));
	hipLaunchKernelGGL(k_gen_regs, dim3(grid), dim3(POST_THREADS), 0, s, b);
}

void launch_post(const PostBatch &b, hipStream_t s, hipStream_t aux, hipEvent_t fork, hipEvent_t join)
{
	if (b.n_reads <= 0) return;
	// MM2GB_DEBUG_LAUNCH=1: wait after every launch and say which one it was (finding a kernel that does not come back)
	static const bool step = [] { const char *v = getenv("MM2GB_DEBUG_LAUNCH"); return v && *v && *v != '0'; }();
	auto done = [&](const char *what) { if (step) { fprintf(stderr, "[mm2gb post-pass] %s ...", what); const hipError_t e = hipStreamSynchronize(s); fprintf(stderr, " %s\n", e == hipSuccess ? "done" : hipGetErrorString(e)); } };
	(void)hipMemsetAsync(b.cursor, 0, 32 * sizeof(int32_t), s);
	(void)hipMemsetAsync(b.size_bins, 0, 2 * N_SIZE_CLASSES * sizeof(int32_t), s);
	const unsigned rgrid = (unsigned)((b.n_reads + 255) / 256);
	hipLaunchKernelGGL(k_post_size_count, dim3(rgrid), dim3(256), 0, s, b); done("k_post_size_count");
	hipLaunchKernelGGL(k_post_size_bases, dim3(1), dim3(64), 0, s, b); done("k_post_size_bases");
	hipLaunchKernelGGL(k_post_size_scatter, dim3(rgrid), dim3(256), 0, s, b); done("k_post_size_scatter");
	const int64_t waves = (int64_t)b.grid_waves;
	unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((b.n_reads + POST_THREADS / W - 1) / (POST_THREADS / W), (waves + POST_THREADS / W - 1) / (POST_THREADS / W)));
	if (b.team_reads > 0) grid = (unsigned)std::max<int64_t>(grid, std::min<int64_t>(b.team_reads, (waves + POST_THREADS / W - 1) / (POST_THREADS / W)));   // a workgroup per team read
	const unsigned lgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((b.n + 255) / 256, 256 * 64));
	// split form: only the WALKS need the lifting tables and the walks' records -- where a second stream is given, the two passes that make them
	// (bandwidth: 24 bytes per anchor) run beside the classes' pass and the sort (waves that wait for LDS and for each other: the sort's levels
	// end with a few long tasks and an idle chip)
	const bool beside = b.cls && aux && fork && join && !step && hipEventRecord(fork, s) == hipSuccess && hipStreamWaitEvent(aux, fork, 0) == hipSuccess;
	{
		hipStream_t ls = beside ? aux : s;
		hipLaunchKernelGGL(k_post_lift, dim3(lgrid), dim3(256), 0, ls, b, 0); done("k_post_lift");
		hipLaunchKernelGGL(k_post_lift, dim3(lgrid), dim3(256), 0, ls, b, 1); done("k_post_lift");
		if (beside) (void)hipEventRecord(join, aux);
	}
	if (b.cls) {
		// split form (round 6): sort | classes of the trees | candidates dealt to their classes | walks per (read, class)
		(void)hipMemsetAsync(b.n_u, 0, (size_t)b.n_reads * sizeof(int32_t), s);
		(void)hipMemsetAsync(b.n_kept, 0, (size_t)b.n_reads * sizeof(int32_t), s);
		hipLaunchKernelGGL(k_post_classes, dim3(grid), dim3(POST_THREADS), 0, s, b); done("k_post_classes");
		if (b.stask[0]) {
			for (int level = 0; level < 4; ++level) {           // key bytes 3 .. 0 of the score
				(void)hipMemsetAsync(b.size_bins, 0, 2 * N_SIZE_CLASSES * sizeof(int32_t), s);
				hipLaunchKernelGGL(k_post_stask_count, dim3(256), dim3(256), 0, s, b, level); done("k_post_stask_count");
				hipLaunchKernelGGL(k_post_size_bases, dim3(1), dim3(64), 0, s, b); done("k_post_size_bases");
				hipLaunchKernelGGL(k_post_stask_scatter, dim3(256), dim3(256), 0, s, b, level); done("k_post_stask_scatter");
				hipLaunchKernelGGL(k_post_sort_level, dim3((unsigned)std::max<int64_t>(1, ((int64_t)b.grid_waves + POST_THREADS / W - 1) / (POST_THREADS / W))), dim3(POST_THREADS), 0, s, b, level); done("k_post_sort_level");
			}
		} else
		hipLaunchKernelGGL(k_post_sort, dim3(grid), dim3(POST_THREADS), 0, s, b, b.team_reads); done("k_post_sort");
		hipLaunchKernelGGL(k_post_partition, dim3(grid), dim3(POST_THREADS), 0, s, b); done("k_post_partition");
		(void)hipMemsetAsync(b.size_bins, 0, 2 * N_SIZE_CLASSES * sizeof(int32_t), s);
		const unsigned tgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((b.n_reads * N_TREE_CLASSES + 255) / 256, 512));
		hipLaunchKernelGGL(k_post_task_count, dim3(tgrid), dim3(256), 0, s, b); done("k_post_task_count");
		hipLaunchKernelGGL(k_post_size_bases, dim3(1), dim3(64), 0, s, b); done("k_post_size_bases");
		hipLaunchKernelGGL(k_post_task_scatter, dim3(tgrid), dim3(256), 0, s, b); done("k_post_task_scatter");
		if (beside) (void)hipStreamWaitEvent(s, join, 0);            // the tables are there
		const int64_t wwaves = std::max<int64_t>(b.walk_grid_waves, 4);
		const unsigned wgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((b.n_reads * N_TREE_CLASSES + POST_THREADS / W - 1) / (POST_THREADS / W), (wwaves + POST_THREADS / W - 1) / (POST_THREADS / W)));
		hipLaunchKernelGGL(k_post_walk, dim3(wgrid), dim3(POST_THREADS), 0, s, b); done("k_post_walk");
	} else
	hipLaunchKernelGGL(k_post_chains, dim3(grid), dim3(POST_THREADS), 0, s, b, b.team_reads); done("k_post_chains");
	hipLaunchKernelGGL(k_post_scan, dim3(1), dim3(1024), 0, s, b); done("k_post_scan");
	hipLaunchKernelGGL(k_post_emit, dim3(grid), dim3(POST_THREADS), 0, s, b); done("k_post_emit");
}

size_t rmq_strip_sort_temp_bytes(int64_t n, int64_t n_reads)
{
	size_t bytes = 0;
	const int64_t *no_offsets = nullptr;
	(void)rocprim::segmented_radix_sort_keys(nullptr, bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned)std::max<int64_t>(n, 1), (unsigned)std::max<int64_t>(n_reads, 1),
	                                         no_offsets, no_offsets, 0, 64, (hipStream_t)0);
	return bytes + 256;
}
int rmq_strip_shift(const RmqParams &P)
{
	const int max_dist = P.max_dist < P.bw ? P.bw : P.max_dist;
	const int max_inner = (P.max_dist_inner <= 0 || P.max_dist_inner >= max_dist) ? 0 : P.max_dist_inner;
	if (max_inner <= 0) return 0;
	int sh = 0;
	while ((1 << sh) < max_inner && sh < 30) ++sh;
	return sh;
}

int launch_rmq_fill(const RmqBatch &b, const RmqParams &P, hipStream_t s)
{
	if (b.n_reads <= 0 || b.n <= 0) { if (b.n_reads > 0) (void)hipMemsetAsync(b.n_tied, 0, (size_t)b.n_reads * sizeof(int32_t), s); return 0; }
	(void)hipMemsetAsync(b.cursor, 0, 2 * sizeof(int32_t), s);
	const unsigned wide = (unsigned)std::min<int64_t>((b.n + 255) / 256, (int64_t)b.grid_waves * 4);
	hipLaunchKernelGGL(k_rmq_prep_keys, dim3(wide), dim3(256), 0, s, b);
	if (b.skey_in && b.skey && b.sort_tmp) {
		// every read's anchors by (y, index): the keys are all different, any sort will do -- a segmented radix sort of the whole batch (the order
		// of chains and seeds, where equal keys must fall as the host's unstable sort leaves them, is what k_sort_x is for: one wave per read)
		size_t tmp = b.sort_tmp_bytes;
		if (rocprim::segmented_radix_sort_keys(b.sort_tmp, tmp, b.skey_in, b.skey, (unsigned)b.n, (unsigned)b.n_reads, b.offsets, b.offsets + 1, 0, 64, s) != hipSuccess) return -1;
		hipLaunchKernelGGL(k_rmq_keys_to_by_y, dim3(wide), dim3(256), 0, s, b);
	} else {
		SortBatch sb;
		sb.a = b.by_y; sb.offsets = b.offsets; sb.n_reads = b.n_reads; sb.cursor = nullptr; sb.grid_waves = b.grid_waves;
		launch_sort_x(sb, s);
	}
	hipLaunchKernelGGL(k_rmq_prep_ranks, dim3(wide), dim3(256), 0, s, b);
	const int max_dist = P.max_dist < P.bw ? P.bw : P.max_dist;
	hipLaunchKernelGGL(k_rmq_prep_ranges, dim3(wide), dim3(256), 0, s, b, max_dist);
	const int per = POST_THREADS / W;
	const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((b.n_reads + per - 1) / per, (int64_t)b.grid_waves / per));
	if (b.tree && b.win) {
		// tile form: 64 anchors per step of a wave, a tournament tree over the ranks for what is settled for a whole tile
		const int max_inner = (P.max_dist_inner <= 0 || P.max_dist_inner >= max_dist) ? 0 : P.max_dist_inner;
		hipLaunchKernelGGL(k_rmq_prep_windows, dim3(wide), dim3(256), 0, s, b, max_dist, max_inner, P.cap_rmq_size);
		hipLaunchKernelGGL(k_rmq_tree_init, dim3(wide), dim3(256), 0, s, b);
		if (b.sa && max_inner > 0) {
			hipLaunchKernelGGL(k_rmq_strip_keys, dim3(wide), dim3(256), 0, s, b);
			size_t tmp = b.sort_tmp_bytes;
			if (rocprim::segmented_radix_sort_keys(b.sort_tmp, tmp, b.skey_in, b.skey, (unsigned)b.n, (unsigned)b.n_reads, b.offsets, b.offsets + 1, 0, 64, s) != hipSuccess) return -1;
			hipLaunchKernelGGL(k_rmq_strip_fill, dim3(wide), dim3(256), 0, s, b);
			hipLaunchKernelGGL(k_rmq_strip_ranges, dim3(wide), dim3(256), 0, s, b, max_inner);
		}
		const int per_t = RMQ_THREADS / W;
		const int64_t singles = std::max<int64_t>(0, b.n_reads - b.n_team);
		const unsigned grid_t = (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::max<int64_t>((singles + per_t - 1) / per_t, b.n_team), (int64_t)b.grid_waves / per_t));   // a workgroup per team read
		hipLaunchKernelGGL(k_rmq_fill_tiles, dim3(grid_t), dim3(RMQ_THREADS), 0, s, b, P);
	} else {
		// one anchor per step (MM2GB_RMQ_KERNEL=steps; and every call with a skip limit: its inner walk goes through the candidates by rank)
		if (P.max_skip != INT_MAX && b.rk_a) {
			const int max_inner = (P.max_dist_inner <= 0 || P.max_dist_inner >= max_dist) ? 0 : P.max_dist_inner;
			hipLaunchKernelGGL(k_rmq_prep_skip, dim3(wide), dim3(256), 0, s, b, max_inner);
		}
		hipLaunchKernelGGL(k_rmq_fill, dim3(grid), dim3(POST_THREADS), 0, s, b, P);
	}
	return 0;
}

} // namespace mm2gb
