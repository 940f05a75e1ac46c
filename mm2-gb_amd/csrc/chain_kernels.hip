// chain_kernels.hip -- gfx950 kernels of the chaining hot path.
//
// What the path computes is fixed by the reference's CPU code (lchain.c:113-207); how it is computed here is not
// taken from the reference's gpu/*.cu:
//   * k_window      the predecessor-window start of every anchor by bisection in LDS (role of plrange.cu:38-76), the input flags and
//                   the planner's per-block reductions (cuts, pair counts, widest windows, max_iter clamps) in one pass over the
//                   16-byte mm128_t records, which every kernel reads IN PLACE: there is no AoS -> SoA copy (the reference makes
//                   one on a CPU thread, plmem.cu:154-198)
//   * plan_*        turn cuts into independent, cost-ordered work items ("chunks") without host round trips
//                   (role of the cut/long_seg/mid_seg bookkeeping, plscore.cu:314-385, and the host pairsort, plchain.cu:30-42)
//   * k_score<MODE> the DP: persistent workgroups; 64 anchors per tile held one-per-lane in registers; predecessors are staged
//                   64 at a time in per-wave LDS and broadcast-read, so every step scores ONE predecessor against the
//                   wave's 64 anchors.  Heavy chunks are pipelined over teams of 4, 8 or 16 waves through an LDS ring of
//                   scores (the sliding predecessor window).  No block barrier per anchor, no global read-modify-write
//                   (role of plscore.cu:109-187, 290-451).
// Arithmetic follows lchain.c:113-138 + mmpriv.h:118-126 bit for bit: compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <algorithm>
#include "chain_dev.h"

namespace mm2gb {

#define WAVE 64

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int bcast(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
__device__ __forceinline__ int first_lane(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Wave-wide maximum and the previous lane's value without the LDS crossbar: DPP row shifts and row broadcasts (six dependent vector
// instructions where six ds_bpermute trips take ~10x as long -- they sit in the in-tile phase of the rescue build, on a team's chain)
__device__ __forceinline__ int wave_max_i32_dpp(int v)
{
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x111, 0xf, 0xf, false));   // row_shr:1
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x112, 0xf, 0xf, false));   // row_shr:2
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x114, 0xf, 0xf, false));   // row_shr:4
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x118, 0xf, 0xf, false));   // row_shr:8  -> lane 15 of every row holds the row's maximum
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3
	return __builtin_amdgcn_readlane(v, WAVE - 1);
}
__device__ __forceinline__ int prev_lane(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }   // wave_shr:1 (lane 0 keeps its own)
// An anchor's fields straight from the caller's array (mm128_t as four dwords: x.lo x.hi y.lo y.hi): reference position = x.lo, query position
// = y.lo, q_span = y.hi & 0xff (lchain.c:125), segment id = y.hi >> 16 & 0xff (lchain.c:116).  Until round 3 k_window copied them into
// arrays of their own (12 B written and read again per anchor: a third of that kernel's traffic) -- the score kernel reads a source block once
// per 128 targets, where the stride of 16 bytes costs nothing that shows.
__device__ __forceinline__ int a_x(const DevBatch &b, int i) { return ((const int*)b.raw)[(size_t)i * 4]; }
__device__ __forceinline__ int a_y(const DevBatch &b, int i) { return ((const int*)b.raw)[(size_t)i * 4 + 2]; }
__device__ __forceinline__ int a_span(const DevBatch &b, int i) { return (int)(((const unsigned*)b.raw)[(size_t)i * 4 + 3] & 0xffu); }
__device__ __forceinline__ int tag_of(unsigned y_hi) { return (int)(((y_hi >> 8) & 0xff00u) | (y_hi & 0xffu)); }      // seg_id << 8 | q_span
__device__ __forceinline__ int a_tag(const DevBatch &b, int i) { return tag_of(((const unsigned*)b.raw)[(size_t)i * 4 + 3]); }


// --------------------------------------------------------------------------------------------------------------
// predecessor window start (lchain.c:172-173), input flags, planner reductions
//   the block's own anchors are read once as mm128_t (16 B, coalesced); nothing but st[] and the flags is written (the score kernel
//   reads the same records in place: no SoA copy, role of plmem.cu:154-198 not needed); look-back probes read the raw anchors too
//   st[i] = max( first j <= i in the same read with xhi[j]==xhi[i] and x[i] <= x[j]+max_dist_x ,  i - max_iter )
// The CPU carries st across iterations; because validity is monotone in both i and j (anchors sorted by x) the
// carried value equals this closed form ("window start").
// --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool in_reach(const DevBatch &b, int j, int hi_i, unsigned x_i, unsigned dist)
{
	const uint2 v = *(const uint2*)&b.raw[j];                    // x.lo = reference position, x.hi = strand | rid
	return (int)v.y == hi_i && x_i <= v.x + dist;                // positions < 2^31, dist < 2^31: no wrap
}

constexpr int WIN_SAMPLE = 32;                         // anchors per sample = ints per 128-byte line
constexpr int WIN_MAX_SAMPLES = 512;                   // look-back of 16 K anchors; a larger max_iter probes memory beyond it
// (Tried, round 6: the 1 024 anchors right before the block in LDS as well, so that the probes between two samples are LDS reads instead of
// dependent trips to L2 -- 24 KB of LDS per workgroup instead of 16, 8 KB more to load per block: the step went from 48.8 to 49.3 ms at
// 500 M anchors and from 3.33 to 3.44 ms on 10-30 kb reads.  The kernel is not waiting for those probes.)

// Read that owns the first anchor of every planning block: one bisection of the read offsets per block, all blocks at once
// (done by the first thread of each k_window workgroup it put ~13 dependent loads in front of every workgroup).
__global__ __launch_bounds__(256) void k_block_reads(DevBatch b)
{
	const int64_t blk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (blk >= b.n_blocks) return;
	const int64_t base = blk * PLAN_BLOCK;
	int64_t lo = 0, hi = b.n_reads;                   // invariant: offsets[lo] <= base < offsets[hi]
	while (hi - lo > 1) {
		const int64_t mid = (lo + hi) >> 1;
		if (b.offsets[mid] <= base) lo = mid; else hi = mid;
	}
	b.blk_read[blk] = (int32_t)lo;
}

// A block of PLAN_BLOCK anchors first puts into LDS (a) its own anchors' reference position and strand|rid (a.x) and (b) one sample per 32 anchors (one per
// 128-byte line of x) of the max_iter anchors before it.  A window start is then found with LDS probes only -- bisection
// over the block's own anchors, or over the samples followed by at most five probes inside the one line the answer lies in,
// a line the block has just touched -- instead of ~25 dependent trips to L2/HBM per search.

__global__ __launch_bounds__(PLAN_THREADS) void k_window(DevBatch b, DevParams P)
{
	__shared__ int s_cut[PLAN_THREADS / WAVE];
	__shared__ unsigned long long s_pairs[PLAN_THREADS / WAVE];
	__shared__ int s_clamp[PLAN_THREADS / WAVE];
	__shared__ int s_wmax[2 * PLAN_THREADS / WAVE];
	__shared__ int own_x[PLAN_BLOCK], own_hi[PLAN_BLOCK];
	__shared__ int smp_x[WIN_MAX_SAMPLES], smp_hi[WIN_MAX_SAMPLES];      // sample k-1 = anchor base - 32 k
	__shared__ int s_st[PLAN_BLOCK];                                    // results, written out coalesced at the end
	// Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8 share an XCD and its L2, MI355X_MICROARCH.md): planning block =
	// f(blockIdx) is chosen so that every XCD works through ONE contiguous eighth of the batch, in order.  The max_iter anchors a
	// block looks back at were then read moments earlier by workgroups of the same XCD and are still in that L2; with the
	// identity mapping they had been read by the other seven XCDs and every look-back line came from HBM again.  Speed only.
	const unsigned nb = gridDim.x, per = nb / 8, rem = nb % 8, xcd = blockIdx.x % 8, kth = blockIdx.x / 8;
	const unsigned blk = nb < 64 ? blockIdx.x : xcd * per + (xcd < rem ? xcd : rem) + kth;
	const int64_t base = (int64_t)blk * PLAN_BLOCK;
	const unsigned dist = (unsigned)P.max_dist_x;
	const int n_samples = (int)min((int64_t)WIN_MAX_SAMPLES, min(base, (int64_t)P.max_iter + WIN_SAMPLE - 1) / WIN_SAMPLE);
	// (the block's first read and its bounds are asked for before the anchors: two dependent loads that every thread's first search waited for)
	const int64_t rd_first = b.blk_read[blk];
	const int64_t rd_first_s = b.offsets[rd_first], rd_first_e = b.offsets[rd_first + 1];
	bool any_seg = false, big_y = false;
	for (int k = threadIdx.x; k < PLAN_BLOCK; k += PLAN_THREADS) {
		const int64_t g = base + k;
		uint4 v = make_uint4(0, 0, 0, 0);                 // x.lo x.hi y.lo y.hi
		if (g < b.n) {
			v = b.raw[g];
			const unsigned span = v.w & 0xffu;            // y>>32 & 0xff      (lchain.c:125)
			const unsigned seg = (v.w >> 16) & 0xffu;     // (y & MM_SEED_SEG_MASK) >> 48 (lchain.c:116)
			any_seg |= seg != 0;
			big_y |= v.z >= (1u << 22) || span == 0;
		}
		own_x[k] = (int)v.x; own_hi[k] = (int)v.y;
	}
	for (int k = threadIdx.x; k < n_samples; k += PLAN_THREADS) {
		const uint2 v = *(const uint2*)&b.raw[base - (int64_t)(k + 1) * WIN_SAMPLE];
		smp_x[k] = (int)v.x; smp_hi[k] = (int)v.y;
	}
	if (__ballot(any_seg) != 0 && lane_id() == 0) atomicOr(b.flags, FLAG_ANY_SEGID);
	if (__ballot(big_y) != 0 && lane_id() == 0) atomicOr(b.flags, FLAG_NO_LUT);
	__syncthreads();
	int my_cut = INT_MAX, my_clamp = 0;
	unsigned long long my_pairs = 0;

	// Each thread owns PLAN_BLOCK / PLAN_THREADS CONSECUTIVE anchors: the first gets a full search, and because window starts
	// are monotone (st[i+1] >= st[i]) the others continue forward from their neighbour's start, usually one or two probes.
	constexpr int PER = PLAN_BLOCK / PLAN_THREADS;
	const int64_t i_first = base + (int64_t)threadIdx.x * PER;
	const int base32 = (int)base;
	int64_t rd = rd_first;                            // read of the current anchor
	int rs = 0, re = 0, st_prev = 0;
	int win[PER];
#pragma unroll
	for (int k = 0; k < PER; ++k) win[k] = -1;
#pragma unroll
	for (int k = 0; k < PER; ++k) {
		const int64_t i64 = i_first + k;
		if (i64 >= b.n) break;
		const int i = (int)i64;
		bool fresh = k == 0;
		if (k == 0 || i >= re) {
			// read that owns anchor i: last r with offsets[r] <= i (gallop forward from the last known read, then bisect)
			int64_t lo = rd, hi = b.n_reads;
			if (k == 0 && rd_first_e > i64) { rs = (int)rd_first_s; re = (int)rd_first_e; fresh = true; }
			else {
			if (b.offsets[lo + 1] > i64) hi = lo + 1;
			else {
				int64_t step = 1;
				while (lo + step < b.n_reads && b.offsets[lo + step] <= i64) { lo += step; step <<= 1; }
				hi = min(lo + step, b.n_reads);
			}
			while (hi - lo > 1) {
				const int64_t mid = (lo + hi) >> 1;
				if (b.offsets[mid] <= i64) lo = mid; else hi = mid;
			}
			rd = lo; rs = (int)b.offsets[lo]; re = (int)b.offsets[lo + 1];
			fresh = true;
			}
		}
		int lb = i - P.max_iter;                      // may be negative
		if (lb < rs) lb = rs;
		const int hi_i = own_hi[i - base32];
		const unsigned x_i = (unsigned)own_x[i - base32];
		auto reach_own = [&](int j) { return own_hi[j - base32] == hi_i && x_i <= (unsigned)own_x[j - base32] + dist; };
		auto reach_smp = [&](int s) { return smp_hi[s] == hi_i && x_i <= (unsigned)smp_x[s] + dist; };   // anchor base - 32 (s + 1)
		// validity is monotone over [lb, i): false ... false true ... true.  st = first valid index, i if none.
		auto reach_at = [&](int j) { return j >= base32 ? reach_own(j) : in_reach(b, j, hi_i, x_i, dist); };
		int st = i;
		if (i > lb && !fresh) {
			// first valid index >= max(neighbour's start, lb); validity is monotone, so gallop forward and bisect
			int l = st_prev > lb ? st_prev : lb, h = i;
			if (reach_at(l)) h = l;
			else {
				for (int step = 1; l + step < h; step <<= 1) {
					if (reach_at(l + step)) { h = l + step; break; }
					l += step;
				}
				while (h - l > 1) {
					const int mid = (l + h) >> 1;
					if (in_reach(b, mid, hi_i, x_i, dist)) h = mid; else l = mid;
				}
			}
			st = h;
			if (st == lb && lb > rs && lb == i - P.max_iter && in_reach(b, lb - 1, hi_i, x_i, dist)) my_clamp = 1;
		} else if (i > lb) {
			int l, h;                                   // invariant: l is out of reach (or lb - 1), h is in reach (or i)
			bool in_block = true;
			if (lb >= base32) { l = lb - 1; h = i; }
			else if (i > base32 && !reach_own(base32)) { l = base32; h = i; }
			else {
				// the window starts at or before the block's first anchor: bisect the samples (closest first: in reach ...
				// in reach, out of reach ...), then the <= 32 anchors between two samples, which share one line of x
				in_block = false;
				const int s_cnt = min(n_samples, (base32 - lb) / WIN_SAMPLE);          // samples at positions >= lb
				int sl = -1, sh = s_cnt;
				while (sh - sl > 1) {
					const int mid = (sl + sh) >> 1;
					if (reach_smp(mid)) sl = mid; else sh = mid;
				}
				h = sl >= 0 ? base32 - (sl + 1) * WIN_SAMPLE : (i > base32 ? base32 : i);
				l = sh < s_cnt ? base32 - (sh + 1) * WIN_SAMPLE : lb - 1;
			}
			if (in_block) {
				while (h - l > 1) {
					const int mid = (l + h) >> 1;
					if (reach_own(mid)) h = mid; else l = mid;
				}
			} else {
				while (h - l > 1) {
					const int mid = (l + h) >> 1;
					if (in_reach(b, mid, hi_i, x_i, dist)) h = mid; else l = mid;
				}
			}
			st = h;
			// the max_iter clamp bit (lchain.c:173): window would have reached further back
			if (st == lb && lb > rs && lb == i - P.max_iter && in_reach(b, lb - 1, hi_i, x_i, dist)) my_clamp = 1;
		}
		s_st[i - base32] = st;
		st_prev = st;
		my_pairs += (unsigned)(i - st);
		win[k] = i - st;
		if (st == i && i < my_cut) my_cut = i;
	}
	// wave then block reductions
	for (int off = WAVE / 2; off > 0; off >>= 1) {
		my_cut = min(my_cut, __shfl_xor(my_cut, off));
		my_pairs += __shfl_xor(my_pairs, off);
		my_clamp |= __shfl_xor(my_clamp, off);
	}
	const int w = threadIdx.x / WAVE;
	if (lane_id() == 0) { s_cut[w] = my_cut; s_pairs[w] = my_pairs; s_clamp[w] = my_clamp; }
	__syncthreads();
	for (int k = threadIdx.x; k < PLAN_BLOCK && base + k < b.n; k += PLAN_THREADS) b.st[base + k] = s_st[k];
	int blk_cut = s_cut[0];
	for (int k = 1; k < PLAN_THREADS / WAVE; ++k) blk_cut = min(blk_cut, s_cut[k]);
	// widest window before the block's first cut (belongs to the chunk that started earlier) and from it on (belongs to
	// the chunk that starts here)
	int head = 0, tail = 0;
#pragma unroll
	for (int k = 0; k < PER; ++k) {
		if (win[k] < 0) continue;
		if (i_first + k < blk_cut) head = max(head, win[k]); else tail = max(tail, win[k]);
	}
	for (int off = WAVE / 2; off > 0; off >>= 1) { head = max(head, __shfl_xor(head, off)); tail = max(tail, __shfl_xor(tail, off)); }
	if (lane_id() == 0) { s_wmax[w] = head; s_wmax[PLAN_THREADS / WAVE + w] = tail; }
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int k = 1; k < PLAN_THREADS / WAVE; ++k) { my_pairs += s_pairs[k]; my_clamp |= s_clamp[k]; }
		head = tail = 0;
		for (int k = 0; k < PLAN_THREADS / WAVE; ++k) { head = max(head, s_wmax[k]); tail = max(tail, s_wmax[PLAN_THREADS / WAVE + k]); }
		b.blk_wmax[2 * blk] = head;
		b.blk_wmax[2 * blk + 1] = tail;
		b.blk_firstcut[blk] = blk_cut;
		b.blk_pairs[blk] = (int64_t)my_pairs;
		b.blk_clamped[blk] = my_clamp;
	}
}

// --------------------------------------------------------------------------------------------------------------
// Planner.  A chunk starts at the first cut of every planning block that has one and runs to the next such cut, so
// chunks are independent DP problems of >= ~PLAN_BLOCK anchors (or one long segment).  Chunks are bucket-sorted by
// estimated cost, most expensive first, into the wave-mode list and the cooperative-mode list.  Six small launches, every
// one parallel over planning blocks or chunks; nothing returns to the host.
//   plan_tile_sums   per tile of 1024 planning blocks: #chunks, sum of pairs, #clamped blocks
//   plan_tile_scan   one workgroup: exclusive scan of the tile sums, totals
//   plan_emit        per tile: chunk start + prefix values at each chunk's first block
//   plan_finish      per chunk: end, cost, flags; histogram of cost bins
//   plan_bins        one workgroup: bin bases (descending cost), counters
//   plan_scatter     per chunk: slot in its list
// --------------------------------------------------------------------------------------------------------------
constexpr int PLANNER_THREADS = 1024;
constexpr int N_SMALL_TEAMS_PLAN = 4;                  // 4-wave teams per score workgroup (k_score: 16 waves)
#ifndef MM2GB_COST_BINS_PER_OCTAVE
#define MM2GB_COST_BINS_PER_OCTAVE 16
#endif
constexpr int COST_SUB = MM2GB_COST_BINS_PER_OCTAVE;   // bins per power of two of a chunk's cost: the lists are served bin by bin, most expensive first
constexpr int COST_SUB_LOG = COST_SUB == 16 ? 4 : COST_SUB == 8 ? 3 : 2;
constexpr int COST_BINS = 64 * COST_SUB;
static_assert(COST_BINS <= PLAN_COST_BINS, "the engine allocates PLAN_COST_BINS bins per list");
enum { LIST_WAVE = 0, LIST_BIG = 1, LIST_TEAM4 = 2, N_LISTS = 3 };

__device__ __forceinline__ int cost_bin(int64_t c)
{
	if (c <= 0) return 0;
	const int msb = 63 - __clzll(c);
	const int frac = msb >= COST_SUB_LOG ? (int)((c >> (msb - COST_SUB_LOG)) & (COST_SUB - 1)) : 0;
	return min(COST_BINS - 1, msb * COST_SUB + frac);
}

template <typename T>
__device__ T block_exclusive_scan(T v, T *s_tmp /* blockDim/64 */, T *total)
{
	T inc = v;
	for (int off = 1; off < WAVE; off <<= 1) {
		T o = __shfl_up(inc, off);
		if (lane_id() >= off) inc += o;
	}
	const int w = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
	__syncthreads();
	if (lane_id() == WAVE - 1) s_tmp[w] = inc;
	__syncthreads();
	T wave_base = 0, tot = 0;
	for (int k = 0; k < nw; ++k) { if (k < w) wave_base += s_tmp[k]; tot += s_tmp[k]; }
	*total = tot;
	return wave_base + inc - v;
}

__global__ __launch_bounds__(PLANNER_THREADS) void plan_tile_sums(DevBatch b)
{
	__shared__ long long s_tmp[PLANNER_THREADS / WAVE];
	const int64_t k = (int64_t)blockIdx.x * PLANNER_THREADS + threadIdx.x;
	const bool in = k < b.n_blocks;
	long long cnt = in && b.blk_firstcut[k] != INT_MAX, pairs = in ? b.blk_pairs[k] : 0, clamps = in ? b.blk_clamped[k] : 0;
	long long t0, t1, t2;
	block_exclusive_scan<long long>(cnt, s_tmp, &t0);
	block_exclusive_scan<long long>(pairs, s_tmp, &t1);
	block_exclusive_scan<long long>(clamps, s_tmp, &t2);
	if (threadIdx.x == 0) { b.tile_sums[blockIdx.x * 3 + 0] = t0; b.tile_sums[blockIdx.x * 3 + 1] = t1; b.tile_sums[blockIdx.x * 3 + 2] = t2; }
}

__global__ __launch_bounds__(PLANNER_THREADS) void plan_tile_scan(DevBatch b, int n_tiles)
{
	__shared__ long long s_tmp[PLANNER_THREADS / WAVE];
	// n_tiles <= 2^31 / 2^20 = 2048: two per thread
	long long carry[3] = { 0, 0, 0 };
	for (int base = 0; base < n_tiles; base += PLANNER_THREADS) {
		const int t = base + threadIdx.x;
		for (int q = 0; q < 3; ++q) {
			const long long v = t < n_tiles ? b.tile_sums[t * 3 + q] : 0;
			long long tot;
			const long long ex = block_exclusive_scan<long long>(v, s_tmp, &tot);
			if (t < n_tiles) b.tile_base[t * 3 + q] = carry[q] + ex;
			carry[q] += tot;
		}
	}
	if (threadIdx.x == 0) {
		b.counters[CNT_NCHUNK] = (int)carry[0];
		b.counters[CNT_NCLAMP] = (int)carry[2];
		b.totals[0] = carry[1];
		b.totals[1] = carry[2];
	}
}

__global__ __launch_bounds__(PLANNER_THREADS) void plan_emit(DevBatch b)
{
	__shared__ long long s_tmp[PLANNER_THREADS / WAVE];
	const int64_t k = (int64_t)blockIdx.x * PLANNER_THREADS + threadIdx.x;
	const bool in = k < b.n_blocks;
	const int fc = in ? b.blk_firstcut[k] : INT_MAX;
	const long long cnt = fc != INT_MAX, pairs = in ? b.blk_pairs[k] : 0, clamps = in ? b.blk_clamped[k] : 0;
	long long tot;
	const long long c = b.tile_base[blockIdx.x * 3 + 0] + block_exclusive_scan<long long>(cnt, s_tmp, &tot);
	const long long pp = b.tile_base[blockIdx.x * 3 + 1] + block_exclusive_scan<long long>(pairs, s_tmp, &tot);
	const long long kk = b.tile_base[blockIdx.x * 3 + 2] + block_exclusive_scan<long long>(clamps, s_tmp, &tot);
	if (cnt) {
		b.chunk_start[c] = fc;
		b.chunk_pp[c] = pp;                  // pairs in blocks before this chunk's first block
		b.chunk_kk[c] = (int)kk;             // clamped blocks before it
		b.chunk_blk[c] = (int)k;
	}
}

__global__ __launch_bounds__(256) void plan_finish(DevBatch b, LaunchCfg cfg)
{
	__shared__ int s_hist[N_LISTS][COST_BINS];
	for (int k = threadIdx.x; k < N_LISTS * COST_BINS; k += blockDim.x) (&s_hist[0][0])[k] = 0;
	__syncthreads();
	const int n_chunks = b.counters[CNT_NCHUNK];
	int n_track = 0;
	long long big_cost = 0;
	for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (int64_t)gridDim.x * blockDim.x) {
		const bool last = c + 1 >= n_chunks;
		const int start = b.chunk_start[c], end = last ? (int)b.n : b.chunk_start[c + 1];
		const long long pp_next = last ? b.totals[0] : b.chunk_pp[c + 1];
		// the chunk covers its own block .. part of the next chunk's block: count clamps inclusively (a superset is safe)
		const long long kk_next = last ? b.totals[1] : (long long)b.chunk_kk[c + 1] + b.blk_clamped[b.chunk_blk[c + 1]];
		const bool track = kk_next - b.chunk_kk[c] > 0;
		const long long cost = (pp_next - b.chunk_pp[c]) + (long long)(end - start) * COST_PER_ANCHOR;
		// How the chunk will be scored (bits 1-2 of chunk_track): LIST_WAVE one wave; LIST_BIG a big team (8 waves, or the whole workgroup for the largest);
		// LIST_TEAM4 a 4-wave team (four chunks per workgroup at a time).  A chunk with mean window W keeps about
		// (W + 128) / 128 waves busy, so wide windows get the big team and a few blocks of window the small one, which also
		// needs the chunk's widest window to fit its share of the LDS ring.
		const long long len = end - start;
		const bool heavy = cfg.ring_slots > 0 && cost >= cfg.long_min_cost && len * cfg.long_min_window <= cost;
		int list = LIST_WAVE;
		if (heavy) {
			// widest window of the chunk: the part of its first block from the first cut on, then whole (cut-free) blocks,
			// then the part of the next chunk's block before that block's first cut
			const int blk0 = b.chunk_blk[c], blk_end = last ? (int)b.n_blocks : b.chunk_blk[c + 1];
			int wmax = b.blk_wmax[2 * blk0 + 1];
			for (int k = blk0 + 1; k < blk_end; ++k) wmax = max(wmax, b.blk_wmax[2 * k]);
			if (!last) wmax = max(wmax, b.blk_wmax[2 * blk_end]);
			// a team's share of the LDS ring must hold the tiles its widest window reaches back to, plus the one being written
			const int need_slots = (wmax + WAVE - 1) / WAVE + 1;
			const int big_slots = cfg.ring_slots / (16 / cfg.big_team), small_slots = cfg.ring_slots / 4;   // k_score: 16 waves, four small teams
			// A team is bounded by the chain of its chunk's in-tile phases, which keeps four waves busy but not eight: wide windows go to
			// 4-wave teams too (table build only: its teams take the scores their ring share no longer holds from global memory) unless
			// the chunk is so large that on four waves it alone would outlast the batch -- more than team4_share_pct % of what each of the
			// launch's 4-wave teams gets of all pairs.  Such chunks, and every wide-window chunk of a small batch, keep the big team.
			const bool lut_build = cfg.host_mode == SCORE_MODE_LUT && !(b.flags[0] & (FLAG_ANY_SEGID | FLAG_NO_LUT));
			const bool small_share = lut_build && (cfg.team4_all || (cfg.team4_share_pct > 0 &&
			                         cost * 100 * cfg.score_grid * N_SMALL_TEAMS_PLAN <= (long long)cfg.team4_share_pct * b.totals[0]));
			if (len * cfg.wide_window > cost && need_slots <= small_slots) list = LIST_TEAM4;
			else if (need_slots <= big_slots) list = small_share ? LIST_TEAM4 : LIST_BIG;   // else: window wider than the ring can hold -> one wave
		}
		b.chunk_end[c] = end;
		b.chunk_cost[c] = cost;
		b.chunk_track[c] = (uint8_t)((track ? 1 : 0) | (list << 1));
		n_track += track;
		if (list == LIST_BIG) big_cost += cost;
		atomicAdd(&s_hist[list][cost_bin(cost)], 1);
	}
	__syncthreads();
	for (int k = threadIdx.x; k < N_LISTS * COST_BINS; k += blockDim.x) {
		const int v = (&s_hist[0][0])[k];
		if (v) atomicAdd(&b.bins[k], v);
	}
	for (int off = WAVE / 2; off > 0; off >>= 1) { n_track += __shfl_xor(n_track, off); big_cost += __shfl_xor(big_cost, off); }
	if (lane_id() == 0 && n_track) atomicAdd(&b.counters[CNT_NTRACK], n_track);
	if (lane_id() == 0 && big_cost) atomicAdd((unsigned long long*)&b.totals[2], (unsigned long long)big_cost);
}

__global__ __launch_bounds__(64) void plan_bins(DevBatch b)
{
	if (threadIdx.x < N_LISTS) {
		int acc = 0;
		int *bins = b.bins + threadIdx.x * COST_BINS;
		for (int k = COST_BINS - 1; k >= 0; --k) { const int v = bins[k]; bins[k] = acc; acc += v; }   // count -> base
		if (threadIdx.x == LIST_BIG) b.counters[CNT_NLONG] = acc;
		if (threadIdx.x == LIST_TEAM4) b.counters[CNT_NMID] = acc;
	}
	if (threadIdx.x == 0) { b.counters[CNT_CURSOR] = 0; b.counters[CNT_LCURSOR] = 0; b.counters[CNT_MCURSOR] = 0; }
}

__global__ __launch_bounds__(256) void plan_scatter(DevBatch b)
{
	// most chunks fall into a handful of bins: rank them inside the workgroup first (LDS), then reserve one range per bin
	// per workgroup in global memory
	__shared__ int s_cnt[N_LISTS * COST_BINS], s_base[N_LISTS * COST_BINS];
	const int n_chunks = b.counters[CNT_NCHUNK];
	for (int64_t c0 = (int64_t)blockIdx.x * blockDim.x; c0 < n_chunks; c0 += (int64_t)gridDim.x * blockDim.x) {
		for (int k = threadIdx.x; k < N_LISTS * COST_BINS; k += blockDim.x) s_cnt[k] = 0;
		__syncthreads();
		const int64_t c = c0 + threadIdx.x;
		int key = -1, rank = 0;
		if (c < n_chunks) {
			key = ((b.chunk_track[c] >> 1) & 3) * COST_BINS + cost_bin(b.chunk_cost[c]);
			rank = atomicAdd(&s_cnt[key], 1);
		}
		__syncthreads();
		for (int k = threadIdx.x; k < N_LISTS * COST_BINS; k += blockDim.x) if (s_cnt[k]) s_base[k] = atomicAdd(&b.bins[k], s_cnt[k]);
		__syncthreads();
		if (key >= 0) {
			int *dst = key >= LIST_TEAM4 * COST_BINS ? b.mid_list : key >= LIST_BIG * COST_BINS ? b.long_list : b.order;
			dst[s_base[key] + rank] = (int)c;
		}
		__syncthreads();
	}
}

// Gangs (chain_dev.h, GangSlot): the chunks at the head of the big-team list whose share of the batch's pairs is worth two workgroups or
// more.  A chunk that costs c of the batch's C gets c / C of the launch's workgroups (x gang_pct %), at most gang_max and never more
// than it has strips; the workgroups [first_wg, first_wg + n_wg) start on it.  One workgroup; the list is ordered by cost bin, so the
// candidates are among its first entries.
__global__ __launch_bounds__(256) void plan_gangs(DevBatch b, LaunchCfg cfg)
{
	__shared__ int s_want[256], s_pos[256], s_first[256];
	const int n_long = b.counters[CNT_NLONG];
	const int t = threadIdx.x;
	int want = 0, ci = -1;
	if (t < n_long && cfg.gang_max >= 2 && !(b.flags[0] & (FLAG_ANY_SEGID | FLAG_NO_LUT))) {
		ci = b.long_list[t];
		const long long cost = b.chunk_cost[ci], total = b.totals[0] + (long long)b.n * COST_PER_ANCHOR;
		const int n_tiles = (b.chunk_end[ci] - b.chunk_start[ci] + WAVE - 1) / WAVE;
		const int strip_tiles = (cfg.gang_pairs ? 2 : 1) * GANG_STRIP_PAIRS;
		const int n_strips = (n_tiles + strip_tiles - 1) / strip_tiles;
		const long long share = cost * cfg.score_grid * cfg.gang_pct / (max(1ll, total) * 100);
		want = (int)min((long long)min(cfg.gang_max, n_strips), share);
		if (want < 2) want = 0;
	}
	s_want[t] = want;
	__syncthreads();
	if (t == 0) {
		int n = 0, wg = 0;
		for (int k = 0; k < 256; ++k) {
			s_pos[k] = -1;
			if (s_want[k] == 0 || n >= GANG_MAX_CHUNKS) continue;
			const int w = min(s_want[k], cfg.score_grid - wg);
			if (w < 2) continue;
			s_pos[k] = n++; s_first[k] = wg; s_want[k] = w; wg += w;
		}
		b.counters[CNT_NGANG] = n;
		b.counters[CNT_GANG_WGS] = wg;
	}
	__syncthreads();
	if (s_pos[t] >= 0) {
		GangSlot g;
		g.next_strip = 0; g.done = 0;
		for (int k = 0; k < 6; ++k) g.keep[k] = 0;
		g.chunk = ci; g.first_wg = s_first[t]; g.n_wg = s_want[t]; g.tiles_per_wave = cfg.gang_pairs ? 2 : 1;
		for (int k = 0; k < 20; ++k) g.pad_[k] = 0;
		b.gang_slots[s_pos[t]] = g;
		b.chunk_track[ci] |= 8;                       // the team phases pass it over
	}
}

// --------------------------------------------------------------------------------------------------------------
// Pair score, lchain.c:113-138 (+ mmpriv.h:118-126).  Three builds of the same arithmetic:
//   MODE_LUT      single query segment, not cDNA, chn_pen_skip == 0 (what --gpu-chain runs with stock presets,
//                 plchain.cu:499-500): the penalty (int)(gap*dd + .5*log2(dd+1)) depends on dd only and dd <= bw for
//                 every accepted pair, so it is tabulated once per parameter set BY THE SAME DEVICE CODE as MODE_FAST
//                 (k_build_lut) and looked up in LDS.  Entry bw+1 is a huge penalty: "dd > bw" rejects by itself.
//   MODE_FAST     same restriction except chn_pen_skip may be non-zero: penalty computed per pair.
//   MODE_GENERAL  every branch of comput_sc (segment ids, cDNA, n_seg > 1).
// --------------------------------------------------------------------------------------------------------------
enum { MODE_LUT = 0, MODE_FAST = 1, MODE_GENERAL = 2 };
constexpr int SCORE_THREADS = 1024;
#ifndef MM2GB_POLL_SLEEP
#define MM2GB_POLL_SLEEP 32            // x 64 cycles between two looks at a team's tile counter
#endif
#ifndef MM2GB_GANG_POLL_SLEEP
#define MM2GB_GANG_POLL_SLEEP 4        // x 64 cycles, in a gang (its waves wait for the chain through the chunk's tiles and little else)
#endif
#ifndef MM2GB_QUARTER_POLL_SLEEP
#define MM2GB_QUARTER_POLL_SLEEP 8     // x 64 cycles between two looks at the quarters a tile has handed out (every look is an LDS trip the in-tile wave waits behind)
#endif
#ifndef MM2GB_INTILE_PRIO
#define MM2GB_INTILE_PRIO 3
#endif
#ifndef MM2GB_SWEEP_GROUP
#define MM2GB_SWEEP_GROUP 4
#endif
constexpr int SWEEP_GROUP = MM2GB_SWEEP_GROUP;
#ifndef MM2GB_PAIR_SWEEP_GROUP
#define MM2GB_PAIR_SWEEP_GROUP 2
#endif
constexpr int PAIR_SWEEP_GROUP = MM2GB_PAIR_SWEEP_GROUP;   // sources per unrolled group of the two-tile sweep
// in the x128 score domain of the LUT sweep 128*(f+16) < 2^30 is guaranteed (FLAG_NO_LUT); LUT_BIAS (chain_dev.h) lies above that

__device__ __forceinline__ float log2_fit(float v)   // mmpriv.h:118-126
{
	unsigned u = __float_as_uint(v);
	float r = (float)(((u >> 23) & 255u) - 128u);
	u = (u & ~(255u << 23)) + (127u << 23);
	const float m = __uint_as_float(u);
	r += (-0.34484843f * m + 2.02466578f) * m - 0.67487759f;
	return r;
}

// lchain.c:129-130,135 for the single-segment, non-cDNA case
__device__ __forceinline__ int gap_penalty(int dd, int dg, const DevParams &P)
{
	const float lin = P.gap * (float)dd + P.skip * (float)dg;
	const float lg = dd >= 1 ? log2_fit((float)(dd + 1)) : 0.0f;
	return (int)(lin + .5f * lg);
}

__global__ void k_build_lut(int *lut, DevParams P)
{
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= LUT_ENTRIES) return;
	// LUT_BIAS - 128 * penalty, so that users ADD it to a score term that was lowered by LUT_BIAS; 0 rejects: what is left is then
	// LUT_BIAS below anything a pair can score.  (skip == 0 here: the dg term is +0.0f)
	lut[k] = k > P.bw || k > P.lut_last ? 0 : LUT_BIAS - 128 * gap_penalty(k, 0, P);
}

__device__ __forceinline__ unsigned abs_diff_u32(int a, int b)
{
	unsigned r;
	asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

// LDS address of the penalty table's entry for a pair, unclamped sweeps: base + |a - b| with a, b taken as UNSIGNED and the sum
// saturating at 2^32 - 1 (integer clamp).  For dq >= 1 (b >= 0) that is the entry of dd = |dr - dq|, or an address beyond the table =
// beyond the workgroup's LDS, which reads 0 = "reject".  For dq <= 0, b is "huge": |a - b| = 2^32 - (dr - dq)*4, and with the base added
// the sum would wrap around into LDS again -- the clamp keeps it at the top of the address space, beyond LDS: such a pair reads 0 too,
// and lanes that read beyond LDS cost the pipe no bank cycles (why the table at the END of LDS, with the address formed here rather than
// in the gather's offset field: with the base in the offset field the wrapped addresses fell on the ring and scratch, 4 ms per 500 M anchors).
__device__ __forceinline__ unsigned lut_address(int a, int b, unsigned base)
{
	unsigned r;
	asm("v_sad_u32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "s"(base));
	return r;
}

template <int MODE>
__device__ __forceinline__ bool pair_score(int xi, int yi, int segi, int xj, int yj, int tagj, const DevParams &P, const int *lut, int &sc_out)
{
	const int dq = yi - yj;
	const int dr = xi - xj;                                    // low 32 bits of the 64-bit difference (lchain.c:119)
	const int spanj = tagj & 0xff;
	const int dg = dr < dq ? dr : dq;
	int sc = spanj < dg ? spanj : dg;
	if (MODE == MODE_LUT) {
		// accepted pairs have dr >= 0 (same strand|rid, sorted by x) and dq >= 1, so the unsigned |dr-dq| is dd
		const unsigned dd = abs_diff_u32(dr, dq);
		const unsigned idx = dd < (unsigned)P.lut_last ? dd : (unsigned)P.lut_last;
		sc_out = sc + ((lut[idx] - LUT_BIAS) >> 7);   // the table stores LUT_BIAS - 128*penalty for the block sweep, 0 = reject
		return (unsigned)(dq - 1) < (unsigned)P.dq_lim && dr != 0;
	}
	const int ddiff = (int)((unsigned)dr - (unsigned)dq);
	const int dd = ddiff < 0 ? -ddiff : ddiff;
	if (MODE == MODE_FAST) {
		const bool ok = (unsigned)(dq - 1) < (unsigned)P.dq_lim && dr != 0 && dd <= P.bw;
		if (dd != 0 || dg > spanj) sc -= gap_penalty(dd, dg, P);
		sc_out = sc;
		return ok;
	} else {
		const int segj = tagj >> 8;
		const bool same = segi == segj;
		const float lin = P.gap * (float)dd + P.skip * (float)dg;
		const float lg = dd >= 1 ? log2_fit((float)(dd + 1)) : 0.0f;
		bool ok = dq > 0 && dq <= P.max_dist_x;
		if (same && (dr == 0 || dq > P.max_dist_y)) ok = false;
		if (same && dd > P.bw) ok = false;
		if (P.n_seg > 1 && !P.is_cdna && same && dr > P.max_dist_y) ok = false;
		if (dd != 0 || dg > spanj) {
			if (P.is_cdna || !same) {
				if (!same && dr == 0) ++sc;
				else if (dr > dq || !same) sc -= (int)(lin < lg ? lin : lg);
				else sc -= (int)(lin + .5f * lg);
			} else sc -= (int)(lin + .5f * lg);
		}
		sc_out = sc;
		return ok;
	}
}

// --------------------------------------------------------------------------------------------------------------
// Tile machinery.  A tile is 64 consecutive anchors of a chunk, lane L owning anchor i0+L ("target").
//   best / arg : running maximum in "threshold" form: best starts at q_span+1 with arg=-1, so "cand >= best" is the
//                CPU's strict '>' against q_span first and "latest j wins ties" afterwards (predecessors are visited in
//                ascending j here; descending with strict '>' on the CPU, lchain.c:174-181).
// Predecessors ("sources") come 64 at a time and are broadcast to all lanes, so each step scores ONE source against the
// wave's 64 targets: 64x reuse of every source.  MODE_LUT broadcasts through LDS (sweep_block_lut); the other builds take
// x/y/tag from the scalar path and the score with v_readlane (sweep_block).
// --------------------------------------------------------------------------------------------------------------
struct Target { int x, y, tag, seg, q, st, hi; bool live; };
struct Keep { int idx, x, hi, y, tag, f; };   // the remembered best anchor ("max_ii", lchain.c:189-205), wave-uniform

// Read-only inputs fetched through the scalar path: a wave-uniform index into constant address space makes the
// compiler emit s_load_dword* (scalar cache -> SGPRs), which costs no vector-ALU issue slot at all.  Only arrays that no
// kernel in flight writes may be read this way (x, y, tag: written by k_window in an earlier launch).
typedef const int __attribute__((address_space(4))) *scalar_i32_ptr;
typedef const int __attribute__((address_space(3))) *lds_i32_ptr;
__device__ __forceinline__ scalar_i32_ptr as_scalar(const void *p) { return (scalar_i32_ptr)(uintptr_t)p; }

// Sources jb+k, k in [k_from, 64), all final.  CHECK: some target windows start inside this block.
// MODE_FAST / MODE_GENERAL sweep: x, y and the span/segment tag of each source come through the scalar path; its score f
// (written by this very kernel, so not readable that way) sits in lane k of `sf` and is broadcast with v_readlane.
struct SrcGroup { int x[4], y[4], t[4]; };
__device__ __forceinline__ SrcGroup load_group(const DevBatch &b, int j0)
{
	const scalar_i32_ptr sr = as_scalar(b.raw);
	SrcGroup g;
#pragma unroll
	for (int u = 0; u < 4; ++u) { const size_t at = (size_t)(j0 + u) * 4; g.x[u] = sr[at]; g.y[u] = sr[at + 2]; g.t[u] = tag_of((unsigned)sr[at + 3]); }
	return g;
}
// MODE_LUT sweep.  A block's 64 sources are first written to this wave's LDS scratch; each step then takes ONE LDS
// broadcast read (ds_read_b128, same address in every lane) for the source and one LDS gather for the penalty, so every
// vector-ALU operand is a VGPR (ops with an SGPR operand issue at half rate on gfx950, profiles/ubench).
// Coordinates are kept multiplied by 4 and shifted by one: the scratch holds 4x, 4y and 4(q_span-1), the target registers
// 4(x-1), 4(y-1).  Then 4(dr-1) - 4(dq-1) = 4(dr-dq), so |.| is directly the BYTE offset into the penalty table (no shift)
// and "(dq-1) <u lim" needs no decrement.
// Scores are kept multiplied by 128 with the source's position in the block, k+1 in 1..64, in the low 7 bits:
//   V = 128*(f_j + min(q_span, dr, dq) - penalty) + (k+1) = (min3 << 5) + scratch.x - table[|dr-dq|]
// with scratch.x = 128(f+1) + (k+1) and the table in units of 128.  The running best enters a block as 128*best (low bits
// clear), so "V > best" is exactly "cand >= best" for the first acceptance and "cand > or (cand == and later source)" after
// it -- the same choices as comparing (cand, j) pairs -- and the arg-max needs no instruction per source: whoever won left
// its k+1 in the low bits.  Per source: two subtractions, |dr-dq|, min3, shift-add, subtraction, the dq range test, the
// running-max test, one select.
// CLAMP: the table has bw+2 entries and the index is clamped to the last ("reject") one.  Without it the table covers every
// distance a pair that passes the range tests can have (|dr-dq| <= max(dr, dq) <= max_dist_x inside a window) and the clamp
// instruction goes; the index of a pair that fails them may point anywhere -- reads beyond the LDS allocation return 0 on
// gfx950 (profiles/ubench) and whatever is read is discarded by the range test.
// CHECK adds "source inside this target's window" and "dr != 0" (lchain.c:120), which can only fail in the first and last
// blocks of a sweep: sources of interior blocks lie inside every target's window and strictly left of every target's x.
// Exact while 128*(f+16) < 2^30 and coordinate differences fit 30 bits: k_window raises FLAG_NO_LUT for query positions
// >= 2^22 (f <= y + q_span always) and the batch then runs the MODE_FAST build.
template <bool CHECK, bool CLAMP>
__device__ __forceinline__ void sweep_block_lut(int t_st, int tx4, int ty4, int jb, int k_from, const int4 *stage,
                                                const DevParams &P, int &bestv)
{
	const unsigned base = (unsigned)P.lut_base, last_at = base + ((unsigned)P.lut_last << 2), lim4 = (unsigned)P.dq_lim << 2;
	constexpr int G = SWEEP_GROUP;                          // sources per unrolled group: G broadcasts + G gathers in flight
	// The v_cmpx statements save the execution mask and put it back INSIDE the same asm statement (s_mov to a scalar pair the
	// statement owns), so they are correct under whatever mask the caller runs with and the compiler never sees exec change.  (An
	// earlier build restored -1, which is only right while every caller is wave-uniform; a copy read in a SEPARATE statement has no
	// data dependence that keeps it in place, and the compiler runs wave-uniform code under whatever mask is at hand.)
	for (int kg = k_from & ~(G - 1); kg < WAVE; kg += G) {
		const int j0 = jb + kg;
		int dqm[G], drm[G], pen[G];
		int4 s4[G];
#pragma unroll
		for (int u = 0; u < G; ++u) s4[u] = stage[kg + u];
#pragma unroll
		for (int u = 0; u < G; ++u) {
			dqm[u] = ty4 - s4[u].w; drm[u] = tx4 - s4[u].z;
			// CLAMP: the saturated address of a pair with dq <= 0 ends at the rejecting entry like any distance beyond bw
			const unsigned at = lut_address(drm[u], dqm[u], base);
			pen[u] = *(lds_i32_ptr)(uintptr_t)(CLAMP ? (at < last_at ? at : last_at) : at);
		}
#pragma unroll
		for (int u = 0; u < G; ++u) {
			const int dg = drm[u] < dqm[u] ? drm[u] : dqm[u];
			int v = ((s4[u].y < dg ? s4[u].y : dg) << 5) + s4[u].x;      // v_lshl_add_u32 ...
			asm("" : "+v"(v));                                            // ... then a plain add (not shift + v_add3, which measured slower)
			v += pen[u];
			if (!CHECK) {
				// "bestv = max(bestv, v) in the lanes whose dq is in range": the range test goes straight into the execution
				// mask (v_cmpx), so no select is needed; the mask is put back within the same statement
				unsigned long long saved;
				asm volatile("s_mov_b64 %[sv], exec\n\tv_cmpx_gt_u32_e32 vcc, %[lim], %[dq]\n\tv_max_i32_e32 %[b], %[v], %[b]\n\ts_mov_b64 exec, %[sv]"
				             : [b] "+v"(bestv), [sv] "=&s"(saved) : [lim] "s"(lim4), [dq] "v"(dqm[u]), [v] "v"(v) : "vcc");
			} else {
				// bitwise on purpose: short-circuit '&&' makes the compiler fork the wave on the first test
				const bool take = ((unsigned)dqm[u] < lim4) & (v > bestv) & (drm[u] != -4) & (j0 + u >= t_st);
				bestv = take ? v : bestv;
			}
		}
	}
}

template <int MODE, bool CHECK>
__device__ __forceinline__ void sweep_block(const DevBatch &b, const Target &T, int jb, int k_from, int sf,
                                            const DevParams &P, const int *lut, int &best, int &arg)
{
	int kg = k_from & ~3;
	SrcGroup nxt = load_group(b, jb + kg);
	for (; kg < WAVE; kg += 4) {
		const SrcGroup g = nxt;
		const int j0 = jb + kg;
		nxt = load_group(b, kg + 4 < WAVE ? j0 + 4 : j0);
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int j = j0 + u;
			const int uf = bcast(sf, kg + u);
			int sc;
			bool take = pair_score<MODE>(T.x, T.y, T.seg, g.x[u], g.y[u], g.t[u], P, lut, sc);
			const int cand = sc + uf;
			take = take && cand >= best;
			if (CHECK) take = take && j >= T.st;
			if (take) { best = cand; arg = j; }
		}
	}
}

// MODE_LUT: a block's 64 sources go to this wave's LDS scratch once ...
// far: the block goes through the FAR build of the unchecked sweep only, which wants the span term folded into the score term and, in
// place of the span, the source's DIAGONAL 4 * ((x - x0) - (y - y0)) with the top bit flipped (d0 = 4 * (x0 - y0), x0 / y0 the block's
// first source: small numbers whatever the positions; the flipped bit makes their unsigned order the signed one, for v_sad_u32)
__device__ __forceinline__ void stage_block_lut(int xs, int ys, int sf, int sq, int4 *stage, bool far = false, int d0 = 0)
{
	const int k = lane_id();
	const int x4 = (int)((unsigned)xs << 2), y4 = (int)((unsigned)ys << 2);
	stage[k] = make_int4(((sf + 1) << 7) + k + 1 - LUT_BIAS + (far ? (sq - 1) << 7 : 0), far ? (int)((unsigned)(x4 - y4 - d0) ^ 0x80000000u) : (sq - 1) * 4, x4, y4);
	__builtin_amdgcn_wave_barrier();                        // LDS is in-order per wave; keep the compiler from reordering
}
__device__ __forceinline__ void stage_block_lut(const DevBatch &b, int jb, int sf, int sq, int4 *stage)
{
	const int js = jb + lane_id();
	stage_block_lut(a_x(b, js), a_y(b, js), sf, sq, stage);
}
// ... and are swept against one tile ...
struct TileXY { int x, y, st; };          // what a sweep needs of a tile: position, query position, window start (INT_MAX: dead lane)
// one tile, a whole block, no test at all: sweep_block_lut2_free (below) explains when and why, and what FAR is
template <bool FAR>
__device__ __forceinline__ void sweep_block_lut_free(int tx4, int ty4, const int4 *stage, const unsigned base, const int d0, int &bestv)
{
	constexpr int G = 4;
	const int td = (int)((unsigned)(tx4 - ty4 - d0) ^ 0x80000000u);   // FAR: the target's diagonal, like the staged ones
	for (int kg = 0; kg < WAVE; kg += G) {
		int4 s4[G];
		int dqm[G], drm[G], pen[G], v[G];
#pragma unroll
		for (int u = 0; u < G; ++u) {
			if (FAR) { const int2 h = *(const int2*)&stage[kg + u]; s4[u].x = h.x; s4[u].y = h.y; }
			else s4[u] = stage[kg + u];
		}
#pragma unroll
		for (int u = 0; u < G; ++u) {
			if (FAR) pen[u] = *(lds_i32_ptr)(uintptr_t)lut_address(td, s4[u].y, base);
			else {
				dqm[u] = ty4 - s4[u].w; drm[u] = tx4 - s4[u].z;
				pen[u] = *(lds_i32_ptr)(uintptr_t)lut_address(drm[u], dqm[u], base);
			}
		}
#pragma unroll
		for (int u = 0; u < G; ++u) {
			if (FAR) v[u] = s4[u].x + pen[u];
			else {
				const int dg = drm[u] < dqm[u] ? drm[u] : dqm[u];
				v[u] = ((s4[u].y < dg ? s4[u].y : dg) << 5) + s4[u].x;
				asm("" : "+v"(v[u]));
				v[u] += pen[u];
			}
		}
		asm("v_max3_i32 %0, %1, %2, %0" : "+v"(bestv) : "v"(v[0]), "v"(v[1]));
		asm("v_max3_i32 %0, %1, %2, %0" : "+v"(bestv) : "v"(v[2]), "v"(v[3]));
	}
}
// An EDGE block -- some target windows start inside it, or it reaches into the run of anchors that share the tile's first position.
// Two tests are left, both straight into the execution mask (v_cmpx; the second narrows the first) around a plain v_max: the query
// distance's range and "source inside this target's window".  dr <= 0 -- a source AT the target's position (lchain.c:120), or one of
// another read of the chunk that happens to lie right of it -- rejects through the table address, with the query distance's sign bit
// cleared as in plain_steps; no compare-and-select chain, no scalar mask arithmetic.  11 vector instructions per source where
// sweep_block_lut<true, .> takes 13 and three scalar ones (round 4: edge blocks are 8 % of the tile-blocks of the bench mix and most of
// what a batch of 10-30 kb reads does).  Unclamped table only.
__device__ __forceinline__ void sweep_block_lut_edge(int t_st, int tx4, int ty4, int jb, int k_from, const int4 *stage, const DevParams &P, int &bestv, const int k_to = WAVE)
{
	constexpr int G = 4;
	const unsigned base = (unsigned)P.lut_base, lim4 = (unsigned)P.dq_lim << 2;
	int pos = 0x7fffffff;
	asm volatile("" : "+v"(pos));
	for (int kg = k_from & ~(G - 1); kg < k_to; kg += G) {
		int4 s4[G];
		int dqm[G], drm[G], pen[G];
#pragma unroll
		for (int u = 0; u < G; ++u) s4[u] = stage[kg + u];
#pragma unroll
		for (int u = 0; u < G; ++u) {
			dqm[u] = ty4 - s4[u].w; drm[u] = tx4 - s4[u].z;
			pen[u] = *(lds_i32_ptr)(uintptr_t)lut_address(drm[u], dqm[u] & pos, base);
		}
#pragma unroll
		for (int u = 0; u < G; ++u) {
			const int dg = drm[u] < dqm[u] ? drm[u] : dqm[u];
			int v = ((s4[u].y < dg ? s4[u].y : dg) << 5) + s4[u].x;
			asm("" : "+v"(v));
			v += pen[u];
			const int j = jb + kg + u;
			unsigned long long saved;                               // (the mask on entry is saved and put back inside the statement: sweep_block_lut)
			asm volatile("s_mov_b64 %[sv], exec\n\tv_cmpx_gt_u32_e32 vcc, %[lim], %[dq]\n\tv_cmpx_ge_i32_e32 vcc, %[j], %[st]\n\tv_max_i32_e32 %[b], %[v], %[b]\n\ts_mov_b64 exec, %[sv]"
			             : [b] "+v"(bestv), [sv] "=&s"(saved) : [lim] "s"(lim4), [dq] "v"(dqm[u]), [st] "v"(t_st), [j] "s"(j), [v] "v"(v) : "vcc");
		}
	}
}
// The same block where the targets' window starts do not fall from lane to lane -- every tile of a chunk (anchors come sorted by position, a
// window start never moves back; dead lanes sit at the end with INT_MAX) --: the lanes whose window holds source k are then a PREFIX of the
// wave, so "source inside this target's window" needs no vector compare per source: how many lanes hold source k is found for all 64 sources
// at once (lane k bisects the lanes' starts, six LDS permutes), four counts are packed per register, and the execution mask of a source is
// made from its count by scalar instructions, which issue beside another wave's vector ones.  What is left per source: the two distances,
// the table address, min3, shift-add, add, the query distance's range straight into the mask, max -- 8 vector instructions (11 above; the
// narrow-window regime, 10-30 kb reads, is made of these blocks).  The caller checks the order of the starts (edge_starts_sorted).
__device__ __forceinline__ bool edge_starts_sorted(int t_st)
{
	const int before = __shfl_up(t_st, 1);
	return __ballot(lane_id() > 0 && t_st < before) == 0;
}
__device__ __forceinline__ void sweep_block_lut_edge_sorted(int t_st, int tx4, int ty4, int jb, int k_from, const int4 *stage, const DevParams &P, int &bestv, const int k_to = WAVE)
{
	constexpr int G = 4;
	const unsigned base = (unsigned)P.lut_base, lim4 = (unsigned)P.dq_lim << 2;
	// lane k: how many lanes' windows hold source jb + k
	int c = 0;
	{
		const int j = jb + lane_id();
#pragma unroll
		for (int step = WAVE / 2; step > 0; step >>= 1) c += __shfl(t_st, c + step - 1) <= j ? step : 0;
		c += __shfl(t_st, c) <= j ? 1 : 0;                        // (c <= 63 here)
	}
	// four counts per register: lane 4 m holds those of sources 4 m .. 4 m + 3
	unsigned packed = (unsigned)c;
	packed |= (unsigned)__builtin_amdgcn_update_dpp(0, c, 0x101, 0xf, 0xf, true) << 8;     // row_shl:1: lane i takes lane i + 1's
	packed |= (unsigned)__builtin_amdgcn_update_dpp(0, c, 0x102, 0xf, 0xf, true) << 16;
	packed |= (unsigned)__builtin_amdgcn_update_dpp(0, c, 0x103, 0xf, 0xf, true) << 24;
	for (int kg = k_from & ~(G - 1); kg < k_to; kg += G) {
		int4 s4[G];
		int dqm[G], drm[G], pen[G], v[G];
		unsigned long long m[G];
		const unsigned c4 = (unsigned)__builtin_amdgcn_readlane((int)packed, kg);
#pragma unroll
		for (int u = 0; u < G; ++u) s4[u] = stage[kg + u];
#pragma unroll
		for (int u = 0; u < G; ++u) {
			dqm[u] = ty4 - s4[u].w; drm[u] = tx4 - s4[u].z;
			pen[u] = *(lds_i32_ptr)(uintptr_t)lut_address(drm[u], dqm[u], base);     // (a query distance <= 0 may read anything: its lane is masked below)
			const unsigned cu = (c4 >> (8 * u)) & 0xffu;
			m[u] = cu ? ~0ull >> (64u - cu) : 0ull;                 // the first cu lanes
		}
#pragma unroll
		for (int u = 0; u < G; ++u) {
			const int dg = drm[u] < dqm[u] ? drm[u] : dqm[u];
			v[u] = ((s4[u].y < dg ? s4[u].y : dg) << 5) + s4[u].x;
			asm("" : "+v"(v[u]));
			v[u] += pen[u];
		}
		unsigned long long saved;
		asm volatile("s_mov_b64 %[sv], exec\n\t"
		             "s_mov_b64 exec, %[m0]\n\tv_cmpx_gt_u32_e32 vcc, %[lim], %[q0]\n\tv_max_i32_e32 %[b], %[v0], %[b]\n\t"
		             "s_mov_b64 exec, %[m1]\n\tv_cmpx_gt_u32_e32 vcc, %[lim], %[q1]\n\tv_max_i32_e32 %[b], %[v1], %[b]\n\t"
		             "s_mov_b64 exec, %[m2]\n\tv_cmpx_gt_u32_e32 vcc, %[lim], %[q2]\n\tv_max_i32_e32 %[b], %[v2], %[b]\n\t"
		             "s_mov_b64 exec, %[m3]\n\tv_cmpx_gt_u32_e32 vcc, %[lim], %[q3]\n\tv_max_i32_e32 %[b], %[v3], %[b]\n\t"
		             "s_mov_b64 exec, %[sv]"
		             : [b] "+v"(bestv), [sv] "=&s"(saved)
		             : [lim] "s"(lim4), [q0] "v"(dqm[0]), [q1] "v"(dqm[1]), [q2] "v"(dqm[2]), [q3] "v"(dqm[3]), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]),
		               [m0] "s"(m[0]), [m1] "s"(m[1]), [m2] "s"(m[2]), [m3] "s"(m[3]) : "vcc");
	}
}
__device__ __forceinline__ void sweep_staged_lut(const TileXY &T, int jb, int k_from, bool no_check, bool free_block, const int4 *stage, const DevParams &P,
                                                 int &best, int &arg, bool far_block = false, int d0 = 0)
{
	const int tx4 = (int)(((unsigned)T.x - 1u) << 2), ty4 = (int)(((unsigned)T.y - 1u) << 2);
	int bestv = best << 7;
	if (far_block) sweep_block_lut_free<true>(tx4, ty4, stage, (unsigned)P.lut_base, d0, bestv);
	else if (free_block) sweep_block_lut_free<false>(tx4, ty4, stage, (unsigned)P.lut_base, 0, bestv);
	else if (!no_check && !P.lut_clamp) {
		if (P.edge_prefix && edge_starts_sorted(T.st)) sweep_block_lut_edge_sorted(T.st, tx4, ty4, jb, k_from, stage, P, bestv);
		else sweep_block_lut_edge(T.st, tx4, ty4, jb, k_from, stage, P, bestv);
	}
	else if (P.lut_clamp) {
		if (no_check) sweep_block_lut<false, true>(T.st, tx4, ty4, jb, k_from, stage, P, bestv);
		else sweep_block_lut<true, true>(T.st, tx4, ty4, jb, k_from, stage, P, bestv);
	} else {
		if (no_check) sweep_block_lut<false, false>(T.st, tx4, ty4, jb, k_from, stage, P, bestv);
		else sweep_block_lut<true, false>(T.st, tx4, ty4, jb, k_from, stage, P, bestv);
	}
	const int won = bestv & 127;                            // k+1 of the source that holds the best, 0 = none of this block
	arg = (unsigned)(won - 1) < (unsigned)WAVE ? jb + won - 1 : arg;   // (1..64 by construction; see sweep_staged_lut2 for the test)
	best = bestv >> 7;
}
// ... or against two tiles at once: one LDS broadcast read per source serves 128 targets (the LDS pipe, one per CU, is as
// busy as the vector ALUs in the one-tile sweep).  Only for blocks that need neither window nor equal-position tests.
template <bool CLAMP>
__device__ __forceinline__ void sweep_block_lut2(int txa, int tya, int txb, int tyb, const int4 *stage, const DevParams &P, int &bva, int &bvb)
{
	const unsigned base = (unsigned)P.lut_base, last_at = base + ((unsigned)P.lut_last << 2), lim4 = (unsigned)P.dq_lim << 2;
	constexpr int G = PAIR_SWEEP_GROUP;                         // (execution mask: see sweep_block_lut)
	for (int kg = 0; kg < WAVE; kg += G) {
		int4 s4[G];
		int dqa[G], dra[G], pa[G], dqb[G], drb[G], pb[G];
#pragma unroll
		for (int u = 0; u < G; ++u) s4[u] = stage[kg + u];
#pragma unroll
		for (int u = 0; u < G; ++u) {
			dqa[u] = tya - s4[u].w; dra[u] = txa - s4[u].z;
			dqb[u] = tyb - s4[u].w; drb[u] = txb - s4[u].z;
			const unsigned aa = lut_address(dra[u], dqa[u], base), ab = lut_address(drb[u], dqb[u], base);
			pa[u] = *(lds_i32_ptr)(uintptr_t)(CLAMP ? (aa < last_at ? aa : last_at) : aa);
			pb[u] = *(lds_i32_ptr)(uintptr_t)(CLAMP ? (ab < last_at ? ab : last_at) : ab);
		}
#pragma unroll
		for (int u = 0; u < G; ++u) {
			const int ga = dra[u] < dqa[u] ? dra[u] : dqa[u], gb = drb[u] < dqb[u] ? drb[u] : dqb[u];
			int va = ((s4[u].y < ga ? s4[u].y : ga) << 5) + s4[u].x;
			asm("" : "+v"(va));
			va += pa[u];
			int vb = ((s4[u].y < gb ? s4[u].y : gb) << 5) + s4[u].x;
			asm("" : "+v"(vb));
			vb += pb[u];
			// one statement for both tiles: the mask on entry is saved and put back INSIDE the statement, so it is whatever the caller
			// runs under (every caller today: all 64 lanes) and the compiler never sees a changed mask
			unsigned long long saved;
			asm volatile("s_mov_b64 %[sv], exec\n\tv_cmpx_gt_u32_e32 vcc, %[lim], %[dqa]\n\tv_max_i32_e32 %[ba], %[va], %[ba]\n\ts_mov_b64 exec, %[sv]\n\t"
			             "v_cmpx_gt_u32_e32 vcc, %[lim], %[dqb]\n\tv_max_i32_e32 %[bb], %[vb], %[bb]\n\ts_mov_b64 exec, %[sv]"
			             : [ba] "+v"(bva), [bb] "+v"(bvb), [sv] "=&s"(saved) : [lim] "s"(lim4), [dqa] "v"(dqa[u]), [va] "v"(va), [dqb] "v"(dqb[u]), [vb] "v"(vb) : "vcc");
		}
	}
}
// The same without any range test, for blocks in which every (source, target) pair has  dr + bw <= dq_lim  (most of a window: with the
// defaults, all sources but those more than 4 500 bases left of the targets).  There the gather alone rejects what the range test
// would: dq > dq_lim means dd = dq - dr > dq_lim - dr >= bw, an index beyond bw, which reads 0 = "reject" from the table or, beyond the
// table, from beyond the workgroup's LDS allocation (out-of-range LDS reads return 0, profiles/ubench/lds_oob.hip; the table ends where
// the allocation ends); and dq <= 0 makes lut_address saturate, which reads 0 as well.  No v_cmpx, no exec juggling, and two sources
// share one v_max3: 6.5 vector instructions per pair instead of 8.  Unclamped table only; dr >= 1 is the caller's (no_check blocks).
// FAR: every pair of the block also has dr >= bw + q_span of its source.  A pair that survives the gather has |dr - dq| <= bw, so
// dq >= q_span too: min(q_span, dr, dq) IS the source's q_span, and the block was staged with 128 * (q_span - 1) already in the score
// term (stage_block_lut) -- no v_min3, no shift-add.  And dq <= 0 needs no saturation trick there: it means dd = dr - dq >= dr > bw,
// which the TRUE |dr - dq| rejects by itself -- so the table address comes from the two diagonals, |dr - dq| = |(x_i - y_i) - (x_j -
// y_j)|, staged per source and formed once per tile and block: ONE v_sad_u32 per pair, no subtractions.  sad, add per pair and one
// v_max3 per two sources: 2.5 vector instructions per pair.  (A pair the gather rejects keeps its LUT_BIAS-low value whatever the
// span term.)
template <bool FAR>
__device__ __forceinline__ void sweep_block_lut2_free(int txa, int tya, int txb, int tyb, const int4 *stage, const unsigned base, const int d0, int &bva, int &bvb)
{
	constexpr int G = 2;
	const int tda = (int)((unsigned)(txa - tya - d0) ^ 0x80000000u), tdb = (int)((unsigned)(txb - tyb - d0) ^ 0x80000000u);   // FAR
	for (int kg = 0; kg < WAVE; kg += G) {
		int4 s4[G];
		int dqa[G], dra[G], pa[G], dqb[G], drb[G], pb[G], va[G], vb[G];
#pragma unroll
		for (int u = 0; u < G; ++u) {
			// FAR needs the score term and the diagonal only: an 8-byte broadcast costs the LDS pipe half of a 16-byte one
			if (FAR) { const int2 h = *(const int2*)&stage[kg + u]; s4[u].x = h.x; s4[u].y = h.y; }
			else s4[u] = stage[kg + u];
		}
#pragma unroll
		for (int u = 0; u < G; ++u) {
			if (FAR) {
				pa[u] = *(lds_i32_ptr)(uintptr_t)lut_address(tda, s4[u].y, base);
				pb[u] = *(lds_i32_ptr)(uintptr_t)lut_address(tdb, s4[u].y, base);
			} else {
				dqa[u] = tya - s4[u].w; dra[u] = txa - s4[u].z;
				dqb[u] = tyb - s4[u].w; drb[u] = txb - s4[u].z;
				pa[u] = *(lds_i32_ptr)(uintptr_t)lut_address(dra[u], dqa[u], base);
				pb[u] = *(lds_i32_ptr)(uintptr_t)lut_address(drb[u], dqb[u], base);
			}
		}
#pragma unroll
		for (int u = 0; u < G; ++u) {
			if (FAR) { va[u] = s4[u].x + pa[u]; vb[u] = s4[u].x + pb[u]; }
			else {
				const int ga = dra[u] < dqa[u] ? dra[u] : dqa[u], gb = drb[u] < dqb[u] ? drb[u] : dqb[u];
				va[u] = ((s4[u].y < ga ? s4[u].y : ga) << 5) + s4[u].x;
				asm("" : "+v"(va[u]));
				va[u] += pa[u];
				vb[u] = ((s4[u].y < gb ? s4[u].y : gb) << 5) + s4[u].x;
				asm("" : "+v"(vb[u]));
				vb[u] += pb[u];
			}
		}
		asm("v_max3_i32 %0, %1, %2, %0" : "+v"(bva) : "v"(va[0]), "v"(va[1]));
		asm("v_max3_i32 %0, %1, %2, %0" : "+v"(bvb) : "v"(vb[0]), "v"(vb[1]));
	}
}
__device__ __forceinline__ void sweep_staged_lut2(const TileXY &TA, const TileXY &TB, int jb, bool free_block, bool far_block, int d0, const int4 *stage, const DevParams &P,
                                                  int &best_a, int &arg_a, int &best_b, int &arg_b)
{
	const int txa = (int)(((unsigned)TA.x - 1u) << 2), tya = (int)(((unsigned)TA.y - 1u) << 2);
	const int txb = (int)(((unsigned)TB.x - 1u) << 2), tyb = (int)(((unsigned)TB.y - 1u) << 2);
	int bva = best_a << 7, bvb = best_b << 7;
	if (far_block) sweep_block_lut2_free<true>(txa, tya, txb, tyb, stage, (unsigned)P.lut_base, d0, bva, bvb);
	else if (free_block) sweep_block_lut2_free<false>(txa, tya, txb, tyb, stage, (unsigned)P.lut_base, 0, bva, bvb);
	else if (P.lut_clamp) sweep_block_lut2<true>(txa, tya, txb, tyb, stage, P, bva, bvb);
	else sweep_block_lut2<false>(txa, tya, txb, tyb, stage, P, bva, bvb);
	// (k+1 of the winner is 1..64 for any input the caller's contract allows; anchors that are not sorted by position can make the
	// unchecked sweep read outside the table, and whatever comes back must not become an index outside the block)
	const int wa = bva & 127, wb = bvb & 127;
	arg_a = (unsigned)(wa - 1) < (unsigned)WAVE ? jb + wa - 1 : arg_a; best_a = bva >> 7;
	arg_b = (unsigned)(wb - 1) < (unsigned)WAVE ? jb + wb - 1 : arg_b; best_b = bvb >> 7;
}

// Sweep of one full source block for either build.  `stage` is this wave's 64-entry LDS scratch (MODE_LUT only).
// no_check: every source of the block is inside every live target's window and left of every target's x.
template <int MODE>
__device__ __forceinline__ void sweep_any(const DevBatch &b, const Target &T, int jb, int k_from, int sf, int sq, bool no_check,
                                          int4 *stage, const DevParams &P, const int *lut, int &best, int &arg)
{
	if (MODE == MODE_LUT) {
		const int xs = a_x(b, jb + lane_id()), ys = a_y(b, jb + lane_id());
		const TileXY xy = { T.x, T.y, T.st };
		// (dead lanes of a tile repeat its last live anchor; sweep_block_lut2_free for the conditions)
		const bool free_block = no_check && P.free_sweep && (unsigned)(bcast(T.x, WAVE - 1) - first_lane(xs)) <= (unsigned)(P.dq_lim - P.bw);
		// FAR: dr >= bw + q_span for every source of the block (its last source and the tile's first anchor give the smallest dr)
		const bool far_block = free_block && first_lane(T.x) > bcast(xs, WAVE - 1) && __ballot(sq + P.bw > first_lane(T.x) - bcast(xs, WAVE - 1)) == 0;
		const int d0 = (first_lane(xs) - first_lane(ys)) * 4;
		stage_block_lut(xs, ys, sf, sq, stage, far_block, d0);
		sweep_staged_lut(xy, jb, k_from, no_check, free_block, stage, P, best, arg, far_block, d0);
		__builtin_amdgcn_wave_barrier();
	} else {
		// pair_score tests dr != 0 itself; only the window start needs the CHECK build
		if (no_check) sweep_block<MODE, false>(b, T, jb, k_from, sf, P, lut, best, arg);
		else sweep_block<MODE, true>(b, T, jb, k_from, sf, P, lut, best, arg);
	}
}

// First anchor of the run of equal reference positions that ends at anchor i0 (scalar walk, bounded).
__device__ __forceinline__ int equal_x_run_start(const DevBatch &b, int cs, int i0, int x0)
{
	const scalar_i32_ptr sr = as_scalar(b.raw);
	int e = i0;
	for (int n = 0; n < 64 && e > cs && sr[(size_t)(e - 1) * 4] == x0; ++n) --e;
	if (e > cs && sr[(size_t)(e - 1) * 4] == x0) e = cs;                  // longer than the bound: check everything
	return e;
}

__device__ __forceinline__ Target load_target(const DevBatch &b, int i0, int ce, bool want_hi)
{
	Target T;
	const int i = i0 + lane_id();
	T.live = i < ce;
	const int il = T.live ? i : ce - 1;
	T.x = a_x(b, il); T.y = a_y(b, il); T.tag = a_tag(b, il);
	T.hi = want_hi ? (int)((const uint2*)&b.raw[il])->y : 0;   // strand | rid, only the rescue state machine looks at it
	T.st = T.live ? b.st[il] : INT_MAX;          // dead lanes never activate
	T.q = T.tag & 0xff; T.seg = T.tag >> 8;
	return T;
}

// ---- inside the tile, table build -----------------------------------------------------------------------------
// Same packed arithmetic as sweep_block_lut with the tile's own anchors as sources: scratch entry t = {-, 4(q_span-1), 4x, 4y}
// of lane t; its score is not known until step t, so 128(f_t+1) + (t+1) is formed on the scalar side from one v_readlane of
// lane t's running value.  Lane state: V = 128*cand + (k+1) as in the sweep; a lane without predecessor so far enters as
// 128*q_span + 127 (one below the acceptance threshold 128(q_span+1)), so that f = V >> 7 holds for every lane at any time.
// The acceptance tests are ballots combined on the scalar side with "lane > t", and one v_cndmask takes the mask.
struct TileLut {
	int tx4, ty4, lo;
	unsigned lim4, base, last_at;   // table: address of entry 0, of the rejecting entry
	const int4 *stage;
	bool edges;              // some window starts inside the tile, or two of its anchors share a reference position
	int kind;                // ROWS_*: what plain_steps has to test in this tile
};

// Source t against the lanes above it, in two parts.  tile_pre: everything that does not depend on scores -- LDS broadcast of
// the source, distances, penalty gather, the three tests that only involve coordinates -- issued one step AHEAD, so that its
// two LDS round trips overlap the previous step.  tile_fin: the dependent chain, one v_readlane of lane t's final value ->
// scalar or/add -> v_add -> v_cmp -> select.  The chain is what bounds a team: tile t+1 cannot finish before tile t has.
struct StepPre { int basev; unsigned long long ok; };

__device__ __forceinline__ StepPre tile_pre(const TileLut &tl, int t)
{
	const int4 s4 = tl.stage[t];
	const int dqm = tl.ty4 - s4.w, drm = tl.tx4 - s4.z;
	const unsigned at = lut_address(drm, dqm, tl.base);
	const int pen = *(lds_i32_ptr)(uintptr_t)(at < tl.last_at ? at : tl.last_at);   // always clamped here: one code path, one instruction more
	const int dg = drm < dqm ? drm : dqm;
	StepPre pre;
	pre.basev = ((s4.y < dg ? s4.y : dg) << 5) + pen;          // one shift-add: the table holds LUT_BIAS - 128*penalty
	const unsigned long long above = t < WAVE - 1 ? ~0ull << (t + 1) : 0ull;
	pre.ok = __ballot((unsigned)dqm < tl.lim4) & above;
	if (tl.edges) pre.ok &= __ballot(drm != -4) & __ballot(tl.lo <= t);   // wave-uniform: most tiles of wide-window chunks skip both
	return pre;
}

__device__ __forceinline__ void tile_fin(const StepPre &pre, int t, int s_bv, int &bestv)
{
	const int fx = (s_bv | 127) + (t + 2 - LUT_BIAS);       // 128(f_t + 1) + (t + 1) - LUT_BIAS, scalar
	const int v = pre.basev + fx;
	// one select under the combined mask (making the mask the execution mask of a v_max instead measured slower here: the
	// scalar write of exec sits in the dependent chain)
	const unsigned long long take = pre.ok & __ballot(v > bestv);
	asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(bestv) : "v"(bestv), "v"(v), "s"(take));
}

// The in-tile steps of the sources in `need` (bit t: source t).  Round 4.  A lone wave issues one instruction every ~5 cycles whatever its
// kind, a dependent vector instruction after ~8, and a hop through the scalar unit (v_readlane -> s_or -> v_add) costs ~20 more
// (profiles/ubench/lone_wave.hip) -- so what bounds a chunk's chain of in-tile phases is the NUMBER of instructions the one wave issues
// per source and the scalar hops of its chain, and both LDS trips of a source must be asked for long before they are needed.  Per source:
//   row (independent of every score; two sources per group, a group's broadcast reads asked for two groups ahead, its gathers one):
//       M_t = 128 * min3' + (LUT_BIAS - 128 pen) + (128 - LUT_BIAS)    -- 2 sub, and, sad, gather, min3, shift, add3
//     The table address rejects by itself whatever a source may not reach: with the query distance's sign bit cleared, dq <= 0 or
//     dr <= 0 -- a lane at or below the source, a lane that shares its position (lchain.c:120) -- puts |4(dr-1) - 4(dq-1)| near 2^31 or
//     above, beyond the table and beyond LDS: the gather reads 0 and the pair ends ~2^30 below any score (lut_address, sweep_block_lut2_free).
//     FREE tiles (no window starts inside the tile, it spans at most dq_lim - bw bases, unclamped table) need nothing else: dq > dq_lim
//     is a distance beyond bw there.  Other tiles: dq range and window start as ballots, one v_cndmask puts -2^30 where they fail.
//   step (the dependent chain): v_readlane lane t's packed value -> v_add M_t -> v_and_or (low 7 bits := t + 1) -> v_max
//     (max, not compare + select: two sources never tie, their codes differ; the code of lane t's own winner rides along in the sum
//     and is overwritten: no scalar instruction in the chain, no execution-mask change, nothing the compiler cannot interleave).
// 11 vector instructions per source (13 in tiles with tests) and ~4 scalar, against 13-15 and ~14 of the step-by-step form (tile_pre /
// tile_fin, which the rescue state machine still uses); the steps alone on a dense tile: profiles/experiments/steps_alone.py.
// A group without a needed source is skipped; the other row of a group that holds one is computed with it (a source no later lane
// reaches fails `lo <= t` in every lane).  Applying a row twice changes nothing (max), so a caller may ask again for rows that ran before.
// Progressive publication (round 4): in a team whose waves take single tiles, the wave of tile t+1 needs tile t's scores for the LAST block
// of its sweep -- a whole block sweep on the chunk's critical path, between two in-tile phases.  A lane of tile t is final long before the
// tile is (lane L after source L - 1): the in-tile phase hands out its scores a quarter of the tile at a time (ring slot, then a counter:
// part = 4 t + quarters), and the next wave sweeps the first three quarters of that block while the phase is still running.
struct Progress {
	int *ring_slot = nullptr;                               // this tile's 64 scores in the team's LDS ring; null: nothing is handed out early
	int *part = nullptr;                                    // the team's counter
	int base = 0;                                           // 4 * (tile of the chunk)
	__device__ __forceinline__ void publish(const int q, const int bestv) const
	{
		// Scores and counter are both in LDS, and the LDS executes one wave's instructions in the order they were issued: the counter's store
		// follows the scores' without a fence (a release here waits for every LDS read the in-tile phase has asked for ahead: 2 k cycles per
		// quarter in a gang's trace).  The reader's acquire load of the counter precedes its loads of the scores the same way.
		if ((lane_id() >> 4) == q) ring_slot[lane_id()] = bestv >> 7;   // (packed value >> 7: the score, also of a lane without predecessor)
		// The hardware order is only worth something if the COMPILER keeps the two stores in program order, and a plain store followed by a
		// relaxed atomic store to another address is nothing it has to keep: a compiler-only release fence (no instruction, no s_waitcnt)
		// says so.  Correctness of the gang path depends on it.
		__atomic_signal_fence(__ATOMIC_RELEASE);
		__builtin_amdgcn_wave_barrier();
		if (lane_id() == 0) __hip_atomic_store(part, base + q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		__builtin_amdgcn_wave_barrier();
	}
};

enum { ROWS_FREE = 0, ROWS_CHECKED = 1, ROWS_CLAMPED = 2 };   // no test at all | dq range and window start | those and a clamped table index

template <int KIND>
__device__ __forceinline__ void plain_steps_impl(const TileLut &tl, const unsigned long long need, int &bestv)
{
	int negv = INT_MIN / 2, keep_hi = ~127, pos = 0x7fffffff;
	asm volatile("" : "+v"(negv), "+v"(keep_hi), "+v"(pos));      // VGPRs, not literals per row
	struct Row { int t1; unsigned at; unsigned long long ok; };
	auto row = [&](const int4 s4, const int t) {
		Row r;
		const int dqm = tl.ty4 - s4.w, drm = tl.tx4 - s4.z;
		r.at = lut_address(drm, dqm & pos, tl.base);
		if (KIND == ROWS_CLAMPED) r.at = r.at < tl.last_at ? r.at : tl.last_at;   // else: an address beyond the table reads 0 = reject
		const int dg = drm < dqm ? drm : dqm;
		r.t1 = ((s4.y < dg ? s4.y : dg) << 5) + s4.x;             // s4.x = 128 - LUT_BIAS (in_tile_lut stages it)
		// (a tile with a window start inside it may hold several reads: only there can a lane BELOW the source lie right of it, and needs the mask)
		r.ok = KIND == ROWS_FREE ? 0ull : __ballot((unsigned)dqm < tl.lim4) & __ballot(tl.lo <= t) & (t < WAVE - 1 ? ~0ull << (t + 1) : 0ull);
		return r;
	};
	auto finish = [&](const Row &r, const int pen) {
		const int sum = r.t1 + pen;
		if (KIND == ROWS_FREE) return sum;
		int m;
		asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(m) : "v"(negv), "v"(sum), "s"(r.ok));
		return m;
	};
	auto step = [&](const int t, const int m) {
		int v = bcast(bestv, t) + m;
		asm("v_and_or_b32 %0, %0, %1, %2" : "+v"(v) : "v"(keep_hi), "s"(t + 1));     // (the compiler splits it into v_and + v_or3 when it likes the scalar side better)
		bestv = v > bestv ? v : bestv;
	};
	// the groups (two sources each) that hold a needed source, in order: bit 2 g of gm.  A tile's needed sources come in runs, and half
	// of a typical tile's sources are needed by nobody (profiles/experiments/intile_counts.py)
	unsigned long long gm = (need | need >> 1) & 0x5555555555555555ull;
	auto next_group = [&]() { const int t = gm ? (int)__builtin_ctzll(gm) : -1; gm &= gm - 1; return t; };
	int tc = next_group();                                        // the group whose steps run in this turn of the loop
	int tn = next_group();                                        // the one after it: its rows are computed meanwhile
	int4 sa = tl.stage[tc], sb = tl.stage[tc + 1];
	Row ra = row(sa, tc), rb = row(sb, tc + 1);
	int pa = *(lds_i32_ptr)(uintptr_t)ra.at, pb = *(lds_i32_ptr)(uintptr_t)rb.at;
	// (loads that nobody will use read the current group again: an unconditional load keeps the registers of sa / sb in place, a
	// conditional one makes the compiler copy eight registers per turn.  A third group in flight -- the gathers a whole turn ahead too --
	// was tried: the rotation costs copies that wait for the gathers, two turns per trip with swapped registers 45 spilled registers)
	{ const int tl0 = tn >= 0 ? tn : tc; sa = tl.stage[tl0]; sb = tl.stage[tl0 + 1]; }
	while (tn >= 0) {
		const int m0 = finish(ra, pa), m1 = finish(rb, pb);
		const int tnn = next_group();
		step(tc, m0);                                               // (its chain has room for the rows' instructions)
		ra = row(sa, tn); rb = row(sb, tn + 1);
		pa = *(lds_i32_ptr)(uintptr_t)ra.at; pb = *(lds_i32_ptr)(uintptr_t)rb.at;
		const int tl1 = tnn >= 0 ? tnn : tn;
		sa = tl.stage[tl1]; sb = tl.stage[tl1 + 1];
		__builtin_amdgcn_sched_barrier(0);                          // the reads stay HERE, a whole turn ahead of their use (the scheduler likes them next to it)
		step(tc + 1, m1);
		tc = tn; tn = tnn;
	}
	step(tc, finish(ra, pa));
	step(tc + 1, finish(rb, pb));
}

__device__ __forceinline__ void plain_steps(const TileLut &tl, unsigned long long need, int &bestv)
{
	if (!need) return;
	if (tl.kind == ROWS_FREE) plain_steps_impl<ROWS_FREE>(tl, need, bestv);
	else if (tl.kind == ROWS_CHECKED) plain_steps_impl<ROWS_CHECKED>(tl, need, bestv);
	else plain_steps_impl<ROWS_CLAMPED>(tl, need, bestv);
}
// The same a quarter of the tile at a time, on_quarter(q, bestv) after the sources of quarter q < 3 (then the lanes up to 16 q + 16 are final):
// Progress.  Four runs of the loop rather than a test per turn inside it -- with the test the compiler spreads the callback's arithmetic over
// every turn (a gang's in-tile phase: 11 k -> 18 k cycles).
template <class Q>
__device__ __forceinline__ void plain_steps_by_quarters(const TileLut &tl, const unsigned long long need, int &bestv, Q &&on_quarter)
{
#pragma nounroll
	for (int q = 0; q < 4; ++q) {
		plain_steps(tl, need & (0xffffull << (16 * q)), bestv);
		if (q < 3) on_quarter(q, bestv);
	}
}

// lchain.c:113-138 for one pair with every input wave-uniform (single segment, no cDNA, chn_pen_skip == 0: the MODE_LUT
// conditions).  The penalty is computed (same function that filled the table) rather than read: an LDS read would put a
// memory round trip into the serial chain of the tile.
__device__ __forceinline__ bool pair_score_uniform(const DevParams &P, int xi, int yi, int xj, int yj, int tagj, int &sc_out)
{
	const int dq = yi - yj, dr = xi - xj, span = tagj & 0xff;
	const int dg = dr < dq ? dr : dq;
	const int diff = dr - dq;
	const int dd = diff < 0 ? -diff : diff;
	sc_out = (span < dg ? span : dg) - gap_penalty(dd, 0, P);
	return (unsigned)(dq - 1) < (unsigned)P.dq_lim && dr != 0 && dd <= P.bw;
}

template <bool TRACK, typename FOld>
__device__ __forceinline__ void in_tile_lut(const DevBatch &b, const Target &T, int i0, int n_here, const DevParams &P, int4 *stage,
                                            int &best, int &arg, Keep &keep, FOld f_old, const Progress &prog = Progress())
{
	const int lane = lane_id(), i = i0 + lane;
	// The in-tile phase is a chain of dependent instructions, and in a team every other wave's next tile waits for it: while it lasts
	// this wave goes first among the 8 waves of its SIMD (the sweeping ones have independent work to fill the slots it leaves).
	__builtin_amdgcn_s_setprio(MM2GB_INTILE_PRIO);
	stage[lane] = make_int4(128 - LUT_BIAS, (T.q - 1) * 4, (int)((unsigned)T.x << 2), (int)((unsigned)T.y << 2));   // .x: plain_steps
	__builtin_amdgcn_wave_barrier();
	TileLut tl;
	tl.tx4 = (int)(((unsigned)T.x - 1u) << 2); tl.ty4 = (int)(((unsigned)T.y - 1u) << 2);
	tl.lo = T.live ? (T.st > i0 ? T.st - i0 : 0) : WAVE;    // first in-tile source inside this lane's window
	tl.lim4 = (unsigned)P.dq_lim << 2; tl.base = (unsigned)P.lut_base; tl.last_at = tl.base + ((unsigned)P.lut_last << 2);
	tl.stage = stage;
	// "source inside this lane's window" and "dr != 0" (lchain.c:120) cannot fail when every window starts before the tile
	// (then all its anchors share strand|rid and are sorted by position, so equal positions are neighbours) and no two
	// neighbours are equal.  Lanes past the end of the chunk may then accept anything: they are never stored nor broadcast.
	const int x_prev = prev_lane(T.x);
	tl.edges = __ballot(T.live && (T.st > i0 || (lane > 0 && T.x == x_prev))) != 0;
	// (dead lanes repeat the tile's last live anchor: what they accept is never stored nor broadcast)
	// plain_steps: no test at all where no window starts inside the tile, the tile spans at most dq_lim - bw bases (dq > dq_lim is then
	// beyond bw: the table rejects it; and at most 2^22, whatever the user's distances: a lane BELOW a source has dr < 0, and the span term
	// 128 * 4(dr - 1) of its rejected pair must not wrap around into the scores) and the table is the unclamped one
	tl.kind = P.lut_clamp ? ROWS_CLAMPED : (P.free_sweep && __ballot(T.live && T.st > i0) == 0 &&
	          (unsigned)(bcast(T.x, WAVE - 1) - first_lane(T.x)) <= (unsigned)min(P.dq_lim - P.bw, 1 << 22)) ? ROWS_FREE : ROWS_CHECKED;
	// (dead lanes repeat the tile's last live anchor: what they accept is never stored nor broadcast)
	int bestv = (best << 7) - (arg < 0 ? 1 : 0);
	if (!TRACK) {
		// source t matters only if anchor t+1 reaches back to it (window starts are monotone)
		if (prog.ring_slot) plain_steps_by_quarters(tl, __ballot(T.live && T.st < i) >> 1, bestv, [&](int q, int bv) { prog.publish(q, bv); });
		else plain_steps(tl, __ballot(T.live && T.st < i) >> 1, bestv);
	} else {
		// the state machine of lchain.c:189-205 runs on the scalar side.  Anchor t's fields come by v_readlane: scalar loads
		// would share the wave's lgkm counter with the LDS reads of every step and expose their latency
		const scalar_i32_ptr sr = as_scalar(b.raw);
		// In a tile whose windows all start before the tile (the usual case where windows are cut by max_iter) only the anchor
		// remembered at the tile's START can ever be out of reach or an extra candidate: one remembered later is an anchor of
		// this tile, hence inside every later window.  What the state machine needs about the entry anchor is computed for all
		// 64 anchors at once -- which find it out of reach, which have it as extra candidate, that candidate's score -- and a
		// step costs a few bit tests; after the first update there is nothing to test at all.  The full state machine below
		// takes over from the first step that needs a rescan (or a cut) until the end of the tile, and runs narrow tiles.
		enum { ENTRY = 0, IN_TILE = 1, FULL = 2 };
		unsigned long long slow = 0ull, extra = 0ull;
		int extra_v = 0;
		const int keep0 = keep.idx;
		int mode = FULL;
		if (keep0 >= 0 && bcast(T.st, n_here - 1) <= i0) {
			mode = ENTRY;
			slow = __ballot(!T.live || T.st == i || T.hi != keep.hi || (unsigned)(T.x - keep.x) > (unsigned)P.max_dist_x);
			// lchain.c:113-138 for (entry anchor -> every anchor of the tile); a distance beyond bw reads the table's "reject"
			// entry, which keeps the candidate far below any score
			const int dq = T.y - keep.y, dr = T.x - keep.x, span = keep.tag & 0xff;
			const int dg = dr < dq ? dr : dq;
			const unsigned dd = abs_diff_u32(dr, dq);
			const unsigned idx = dd < (unsigned)P.lut_last ? dd : (unsigned)P.lut_last;
			const int sc = (span < dg ? span : dg) + ((*(lds_i32_ptr)(uintptr_t)((idx << 2) + (unsigned)P.lut_base) - LUT_BIAS) >> 7);
			extra = __ballot((unsigned)(dq - 1) < (unsigned)P.dq_lim && dr != 0 && keep0 < T.st - 1);
			extra_v = (sc + keep.f) << 7;
		}
		// The whole tile in entry mode, at the cost of the plain build (round 3: entry steps were a third of this build's steps and cost 2.6 times a plain
		// one).  Where the state machine has nothing to rescan (`slow` empty), all it does until an anchor of the tile becomes the remembered
		// one -- step t_s, the first with a final score above the entry anchor's -- is to offer the entry anchor to lane t when lane t's turn comes,
		// strictly ('>' of lchain.c:199).  A maximum does not care in which order its candidates arrive, and the packed score keeps the tie rule
		// (an anchor of the window beats the extra candidate at equal score: its low bits are not zero), so the offer is made to every lane that
		// has it BEFORE the steps, the plain steps run, and t_s is read off the final scores (lanes up to t_s have had nothing they should not).
		// A lane beyond t_s has no extra candidate in the reference (the remembered anchor is inside its window by then): if one of them ENDS with
		// it, the tile is done again the long way -- a remembered anchor far behind the window that still beats everything inside it, after a
		// better anchor has turned up in the tile: rare.  The remembered anchor afterwards: the first anchor holding the tile's largest score, if
		// that beats the entry anchor's (what the running '<' of lchain.c:204-205 leaves).
		bool done_fast = false;
		const unsigned long long live_m = n_here >= WAVE ? ~0ull : ((1ull << n_here) - 1ull);
		if (mode == ENTRY && (slow & live_m) == 0ull) {
			const int bestv0 = bestv, arg0 = arg;
			const bool took = ((extra >> lane) & 1ull) != 0ull && ((bestv | 127) < extra_v);
			bestv = took ? extra_v : bestv;
			arg = took ? keep0 : arg;
			// (handing out a quarter early: its lanes are final, and they are RIGHT unless one of them, beyond the first anchor that beats the
			// entry anchor, has ended with the extra candidate -- the very test made for the whole tile below, on the lanes that are final so far;
			// once it fails nothing more is handed out, the tile is done again the long way, and what was handed out stays right: a lane's
			// value does not depend on the lanes above it)
			bool handed_wrong = false;
			const bool any_took = __ballot(took) != 0ull;               // (no lane took the entry anchor: nothing to test)
			auto hand_out = [&](int q, int bv) {
				if (!any_took) { prog.publish(q, bv); return; }
				if (handed_wrong) return;
				const int last = 16 * q + 15;
				const int f_q = lane < n_here && lane <= last ? bv >> 7 : INT_MIN;
				const unsigned long long ab = __ballot(f_q > keep.f);
				const int ts_q = ab ? (int)__builtin_ctzll(ab) : WAVE;
				if (__ballot(took && lane > ts_q && lane <= last && lane < n_here && bv == extra_v) != 0ull) { handed_wrong = true; return; }
				prog.publish(q, bv);
			};
			if (prog.ring_slot) plain_steps_by_quarters(tl, __ballot(T.live && T.st < i) >> 1, bestv, hand_out);
			else plain_steps(tl, __ballot(T.live && T.st < i) >> 1, bestv);
			const int f_l = lane < n_here ? bestv >> 7 : INT_MIN;
			const unsigned long long above = __ballot(f_l > keep.f);
			const int t_s = above ? (int)__builtin_ctzll(above) : WAVE;
			if (__ballot(took && lane > t_s && lane < n_here && bestv == extra_v) == 0ull) {
				if (above) {
					int top = f_l;
					top = wave_max_i32_dpp(top);
					keep.idx = i0 + (int)__builtin_ctzll(__ballot(f_l == top)); keep.f = top;
					mode = IN_TILE;                                   // (its other fields below)
				}
				done_fast = true;
			} else { bestv = bestv0; arg = arg0; }
		}
		StepPre cur; cur.basev = 0; cur.ok = 0ull;
		if (!done_fast) cur = tile_pre(tl, 0);                                     // (two LDS trips nobody needs once the tile is done)
		int t = done_fast ? n_here : 0;
		for (; t < n_here; ++t) {
			if (mode == IN_TILE) break;                                              // the rest of the tile: plain steps, below
			const int j = i0 + t;
			const StepPre nxt = tile_pre(tl, t + 1 < n_here ? t + 1 : t);
			if (mode == ENTRY && !(slow >> t & 1)) {
				// lchain.c:196-201 with the precomputed candidate; strict: (V | 127) < 128*cand  <=>  V >> 7 < cand
				if (mode == ENTRY && (extra >> t & 1)) {
					const bool take = (lane == t) & ((bestv | 127) < extra_v);
					bestv = take ? extra_v : bestv;
					arg = take ? keep0 : arg;
				}
				const int s_bv = bcast(bestv, t);
				const int ft = s_bv >> 7;                                        // lchain.c:202
				if (keep.f < ft) { keep.idx = j; keep.f = ft; mode = IN_TILE; }  // lchain.c:204-205; its other fields when the tile ends
				tile_fin(cur, t, s_bv, bestv);
				cur = nxt;
				continue;
			}
			mode = FULL;                                                         // (only ever from ENTRY: IN_TILE stays in the branch above)
			const int xt = bcast(T.x, t), yt = bcast(T.y, t), tgt = bcast(T.tag, t), ht = bcast(T.hi, t), stt = bcast(T.st, t);
			// lchain.c:190-195: the remembered anchor fell out of reach (or none yet): arg-max of f over the window,
			// largest index among equals.  An empty window (stt == j) is a natural cut, where the host's scan finds nothing
			// too, or the first anchor of a read, where the host starts over with max_ii = -1 (lchain.c:156) even if the
			// previous read's remembered anchor lies within max_dist_x on the same strand and reference.
			if (keep.idx < 0 || stt == j || ht != keep.hi || (unsigned)(xt - keep.x) > (unsigned)P.max_dist_x) {
				int bf = INT_MIN, bi = -1;
				for (int jj = stt + lane; jj < i0; jj += WAVE) {              // earlier tiles (ascending per lane)
					const int v = f_old(jj);
					if (v >= bf) { bf = v; bi = jj; }
				}
				if (lane < t && i >= stt) {                                      // finished lanes of this tile
					const int v = bestv >> 7;
					if (v >= bf) { bf = v; bi = i; }
				}
				for (int off = WAVE / 2; off > 0; off >>= 1) {
					const int of = __shfl_xor(bf, off), oi = __shfl_xor(bi, off);
					if (of > bf || (of == bf && oi > bi)) { bf = of; bi = oi; }
				}
				keep.idx = first_lane(bi);
				if (keep.idx >= 0) {
					keep.f = first_lane(bf);
					{ const size_t at = (size_t)keep.idx * 4; keep.x = sr[at]; keep.y = sr[at + 2]; keep.tag = tag_of((unsigned)sr[at + 3]); keep.hi = ht; }
				}
			}
			// lchain.c:196-201: one more candidate if the scan stopped before reaching it (end_j = st-1 at max_skip=inf)
			if (keep.idx >= 0 && keep.idx < stt - 1) {
				int sc;
				if (pair_score_uniform(P, xt, yt, keep.x, keep.y, keep.tag, sc)) {
					const int candv = (sc + keep.f) << 7;
					const bool take = (lane == t) & ((bestv | 127) < candv);
					bestv = take ? candv : bestv;
					arg = take ? keep.idx : arg;
				}
			}
			const int s_bv = bcast(bestv, t);
			const int ft = s_bv >> 7;                                            // lchain.c:202
			// lchain.c:204-205 (in reach is guaranteed after the refresh above)
			if (keep.idx < 0 || keep.f < ft) { keep.idx = j; keep.x = xt; keep.hi = ht; keep.y = yt; keep.tag = tgt; keep.f = ft; }
			tile_fin(cur, t, s_bv, bestv);
			cur = nxt;
		}
		if (mode == IN_TILE && t < n_here) {
			// Once an anchor of this tile is the remembered one, nothing the state machine tests can happen before the tile ends: the
			// remembered anchor lies inside every later window of the tile (no extra candidate, lchain.c:196), it is in reach, and it only
			// changes to a later anchor of the tile with a higher score (lchain.c:204-205) -- which nobody looks at before the tile ends.
			// So the remaining sources take the plain steps (those no later anchor reaches are skipped, as in the plain build), and the
			// remembered anchor is brought up to date afterwards: the first anchor that holds the largest final score from here on, if that
			// beats the one remembered now -- what updating at every step with a strict '<' leaves.  (Three quarters of the rescue
			// build's in-tile steps are of this kind on the bench's reads: profiles/r03_block_kinds.json.)
			const int t_from = t;
			plain_steps(tl, (__ballot(T.live && T.st < i) >> 1) & (t_from < WAVE ? ~0ull << t_from : 0ull), bestv);
			const int f_l = (lane >= t_from && lane < n_here) ? bestv >> 7 : INT_MIN;
			int top = f_l;
			top = wave_max_i32_dpp(top);
			if (top > keep.f) { keep.idx = i0 + __builtin_ctzll(__ballot(f_l == top)); keep.f = top; }
		}
		if (mode == IN_TILE) {                                                   // the anchor remembered now is one of this tile
			const int k = keep.idx - i0;
			keep.x = bcast(T.x, k); keep.y = bcast(T.y, k); keep.tag = bcast(T.tag, k); keep.hi = bcast(T.hi, k);
		}
	}
	const int won = bestv & 127;                            // 1..64: source won-1 of this tile; 0: an earlier anchor; 127: none
	arg = (unsigned)(won - 1) < (unsigned)WAVE ? i0 + won - 1 : arg;
	best = (bestv + 1) >> 7;
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_s_setprio(0);
}

// Predecessors inside the tile: lane t becomes final at step t and is pushed to the lanes above it.
// F(j) returns the final score of an anchor of an EARLIER tile (global memory for the wave kernel, LDS ring for the
// cooperative one); only the rescue state machine needs it.
template <int MODE, bool TRACK, typename FOld>
__device__ __forceinline__ void in_tile(const DevBatch &b, const Target &T, int i0, int n_here, const DevParams &P, const int *lut, int4 *stage,
                                        int &best, int &arg, Keep &keep, FOld f_old, const Progress &prog = Progress())
{
	if (MODE == MODE_LUT) {
		in_tile_lut<TRACK>(b, T, i0, n_here, P, stage, best, arg, keep, f_old, prog);
		return;
	}
	const int lane = lane_id(), i = i0 + lane;
	if (!TRACK) {
		// source t matters only if anchor t+1 reaches back to it (window starts are monotone)
		unsigned long long need = __ballot(T.live && T.st < i) >> 1;
		while (need) {
			const int t = __builtin_ctzll(need);
			need &= need - 1;
			const int j = i0 + t;
			const int ft = bcast(arg < 0 ? T.q : best, t);
			const int ux = bcast(T.x, t), uy = bcast(T.y, t), ut = bcast(T.tag, t);
			int sc;
			const bool ok = pair_score<MODE>(T.x, T.y, T.seg, ux, uy, ut, P, lut, sc);
			const int cand = sc + ft;
			if (ok && lane > t && j >= T.st && cand >= best) { best = cand; arg = j; }
		}
		return;
	}
	for (int t = 0; t < n_here; ++t) {
		const int j = i0 + t;
		const int xt = bcast(T.x, t), yt = bcast(T.y, t), tgt = bcast(T.tag, t), ht = bcast(T.hi, t), stt = bcast(T.st, t);
		// lchain.c:190-195: the remembered anchor fell out of reach (or none yet): arg-max of f over the window,
		// largest index among equals.  An empty window (stt == j) is a natural cut, where the host's scan finds nothing
		// too, or the first anchor of a read, where the host starts over with max_ii = -1 (lchain.c:156) even if the
		// previous read's remembered anchor lies within max_dist_x on the same strand and reference.
		if (keep.idx < 0 || stt == j || ht != keep.hi || (unsigned)(xt - keep.x) > (unsigned)P.max_dist_x) {
			int bf = INT_MIN, bi = -1;
			for (int jj = stt + lane; jj < i0; jj += WAVE) {              // earlier tiles (ascending per lane)
				const int v = f_old(jj);
				if (v >= bf) { bf = v; bi = jj; }
			}
			if (lane < t && i >= stt) {                                      // finished lanes of this tile
				const int v = arg < 0 ? T.q : best;
				if (v >= bf) { bf = v; bi = i; }
			}
			for (int off = WAVE / 2; off > 0; off >>= 1) {
				const int of = __shfl_xor(bf, off), oi = __shfl_xor(bi, off);
				if (of > bf || (of == bf && oi > bi)) { bf = of; bi = oi; }
			}
			keep.idx = first_lane(bi);
			if (keep.idx >= 0) {
				keep.f = first_lane(bf);
				keep.x = a_x(b, keep.idx); keep.y = a_y(b, keep.idx); keep.tag = a_tag(b, keep.idx); keep.hi = ht;
			}
		}
		// lchain.c:196-201: one more candidate if the scan stopped before reaching it (end_j = st-1 at max_skip=inf)
		if (keep.idx >= 0 && keep.idx < stt - 1) {
			int sc;
			const bool ok = pair_score<MODE>(xt, yt, tgt >> 8, keep.x, keep.y, keep.tag, P, lut, sc);
			if (ok && lane == t) {
				const int cur = arg < 0 ? T.q : best;
				if (cur < sc + keep.f) { best = sc + keep.f; arg = keep.idx; }
			}
		}
		const int ft = bcast(arg < 0 ? T.q : best, t);                       // lchain.c:202
		// lchain.c:204-205 (in reach is guaranteed after the refresh above)
		if (keep.idx < 0 || keep.f < ft) { keep.idx = j; keep.x = xt; keep.hi = ht; keep.y = yt; keep.tag = tgt; keep.f = ft; }
		int sc;
		const bool ok = pair_score<MODE>(T.x, T.y, T.seg, xt, yt, tgt, P, lut, sc);
		const int cand = sc + ft;
		if (ok && lane > t && j >= T.st && cand >= best) { best = cand; arg = j; }
	}
}

// ---- wave mode: one wave owns the chunk [cs, ce) --------------------------------------------------------------
template <int MODE, bool TRACK>
__device__ __forceinline__ void run_chunk(const DevBatch &b, const DevParams &P, const int *lut, int4 *stage, const int cs, const int ce)
{
	const int lane = lane_id();
	Keep keep; keep.idx = -1; keep.x = keep.hi = keep.y = keep.tag = keep.f = 0;
	for (int i0 = cs; i0 < ce; i0 += WAVE) {
		const Target T = load_target(b, i0, ce, TRACK);
		const int n_here = min(WAVE, ce - i0);
		int best = T.q + 1, arg = -1;
		const int tile_lo = first_lane(T.st);                 // lane 0 has the smallest window start
		const int st_hi = bcast(T.st, n_here - 1);            // the last live lane the largest
		int jb = cs + ((tile_lo - cs) & ~(WAVE - 1));
		if (jb < i0) {
			const int eq_lo = MODE == MODE_LUT ? equal_x_run_start(b, cs, i0, first_lane(T.x)) : i0;
			int sf = b.f[jb + lane], sq = MODE == MODE_LUT ? a_span(b, jb + lane) : 0;
			for (; jb < i0; jb += WAVE) {
				// next block's scores are requested before this block is consumed
				const int jn = jb + WAVE < i0 ? jb + WAVE + lane : jb + lane;
				const int nf = b.f[jn], nq = MODE == MODE_LUT ? a_span(b, jn) : 0;
				const int k_from = tile_lo > jb ? tile_lo - jb : 0;
				sweep_any<MODE>(b, T, jb, k_from, sf, sq, jb >= st_hi && jb + WAVE <= eq_lo, stage, P, lut, best, arg);
				sf = nf; sq = nq;
			}
		}
		in_tile<MODE, TRACK>(b, T, i0, n_here, P, lut, stage, best, arg, keep, [&](int jj) { return b.f[jj]; });
		if (T.live) {
			const int i = i0 + lane;
			b.f[i] = arg < 0 ? T.q : best;
			b.p[i] = arg < 0 ? 0 : i - arg;
		}
	}
}

// ---- table build: a wave works on TWO consecutive tiles at a time (A: anchors i0.., B: the 64 after them) ----------------
// Blocks that lie inside every window of both tiles -- all but the edges of a sweep -- are swept against both at once, one
// LDS broadcast read per source for 128 targets.  Edge blocks are staged once and swept per tile with the checked build.
// Then tile A's in-tile phase, tile A as a source block for tile B, tile B's in-tile phase.
// Registers are what limits the pair (64 VGPRs for 8 waves per SIMD): during the sweeps a tile is three values per lane plus
// its running best; the rest of an anchor (span, strand|rid) is fetched again when the tile's in-tile phase starts.
struct TilePair {
	TileXY A, B;
	int n_a, n_b;            // live anchors (n_b = 0: the chunk ends within A)
	int lo_a, hi_a, lo_b, hi_b;   // smallest / largest window start of each tile
	int x_first, x_last_a, x_last;   // reference position of the pair's first anchor, of tile A's last live one, of the pair's last live one
	int best_a, arg_a, best_b, arg_b;
};

__device__ __forceinline__ TileXY load_xy(const DevBatch &b, int i0, int ce, int &best)
{
	const int i = i0 + lane_id();
	const bool live = i < ce;
	const int il = live ? i : ce - 1;
	TileXY t;
	t.x = a_x(b, il); t.y = a_y(b, il);
	t.st = live ? b.st[il] : INT_MAX;
	best = a_span(b, il) + 1;                  // threshold form: nothing beats q_span without exceeding it
	return t;
}

__device__ __forceinline__ TilePair load_pair(const DevBatch &b, int i0, int ce)
{
	TilePair t;
	t.n_a = min(WAVE, ce - i0);
	t.n_b = max(0, min(WAVE, ce - i0 - WAVE));
	t.A = load_xy(b, i0, ce, t.best_a);
	t.B = load_xy(b, t.n_b ? i0 + WAVE : i0, ce, t.best_b);
	t.lo_a = first_lane(t.A.st); t.hi_a = bcast(t.A.st, t.n_a - 1);
	t.lo_b = t.n_b ? first_lane(t.B.st) : INT_MAX; t.hi_b = t.n_b ? bcast(t.B.st, t.n_b - 1) : INT_MAX;
	t.x_first = first_lane(t.A.x); t.x_last_a = bcast(t.A.x, t.n_a - 1);
	t.x_last = t.n_b ? bcast(t.B.x, t.n_b - 1) : t.x_last_a;            // (dead lanes repeat the last live one)
	t.arg_a = -1; t.arg_b = -1;
	return t;
}

// one block of sources before tile A (scores sf, spans sq, one per lane): staged, then swept against the pair
__device__ __forceinline__ void sweep_pair_block(const DevBatch &b, TilePair &t, int jb, int eq_lo, int sf, int sq, int4 *stage, const DevParams &P)
{
	const bool nc_a = jb >= t.hi_a && jb + WAVE <= eq_lo;
	const bool use_b = t.n_b > 0 && jb + WAVE > t.lo_b;                      // the block reaches into B's windows
	const bool nc_b = use_b && jb >= t.hi_b && jb + WAVE <= eq_lo;           // (sources left of A are left of B, or share A's first x)
	const int xs = a_x(b, jb + lane_id()), ys = a_y(b, jb + lane_id());
	// every pair of this block has dr + bw <= dq_lim (sources are sorted by position: the block's first source and the pair's last
	// anchor give the largest dr): the gather rejects by itself (sweep_block_lut2_free)
	// A chunk may hold several runs of anchors (reads, strands): positions are only comparable between a block and the tiles whose
	// windows it lies in, so a tile that takes the block alone is judged by its own last anchor, and a difference that is not a
	// distance (negative) never passes (unsigned compare).  found by tests/fuzz_soak.py: tests/golden/regress/fuzz_tiny_dist_y.npz
	const unsigned free_span = (unsigned)(P.dq_lim - P.bw);
	const bool free_block = P.free_sweep && (unsigned)(t.x_last - first_lane(xs)) <= free_span;
	const bool free_a = P.free_sweep && (unsigned)(t.x_last_a - first_lane(xs)) <= free_span;
	// ... and dr >= bw + q_span for every source too (the block's last source and the pair's first anchor give the smallest dr): the FAR
	// build, for which the block is staged differently -- so only where both tiles take it
	// (a pair without a tile B -- a gang's single tiles, a chunk's last odd tile -- takes the block alone and may take it the FAR way too)
	const bool single = t.n_b == 0;
	const bool far_block = free_block && nc_a && (nc_b || single) && t.x_first > bcast(xs, WAVE - 1) && __ballot(sq + P.bw > t.x_first - bcast(xs, WAVE - 1)) == 0;
	const int d0 = (first_lane(xs) - first_lane(ys)) * 4;
	stage_block_lut(xs, ys, sf, sq, stage, far_block, d0);
	if (nc_a && nc_b) sweep_staged_lut2(t.A, t.B, jb, free_block, far_block, d0, stage, P, t.best_a, t.arg_a, t.best_b, t.arg_b);
	else {
		sweep_staged_lut(t.A, jb, t.lo_a > jb ? t.lo_a - jb : 0, nc_a, nc_a && free_a, stage, P, t.best_a, t.arg_a, far_block, d0);
		if (use_b) sweep_staged_lut(t.B, jb, t.lo_b > jb ? t.lo_b - jb : 0, nc_b, nc_b && free_block, stage, P, t.best_b, t.arg_b);
	}
	__builtin_amdgcn_wave_barrier();
}

// tile A (final scores f_a, one per lane) as the last source block of tile B
__device__ __forceinline__ void sweep_a_into_b(const DevBatch &b, TilePair &t, int cs, int i0, int f_a, int q_a, int4 *stage, const DevParams &P)
{
	if (i0 + WAVE <= t.lo_b) return;                                           // no window of B reaches into A
	stage_block_lut(b, i0, f_a, q_a, stage);
	const int eq_lo = equal_x_run_start(b, cs, i0 + WAVE, first_lane(t.B.x));
	// (this sweep sits on the team's critical path, between the two in-tile phases of the pair: where tile A lies inside every window of B, left
	// of B's first position and within dq_lim - bw of its last, the unchecked sweep does it in 6.5 instead of 8 instructions per source)
	const bool no_check = i0 >= t.hi_b && i0 + WAVE <= eq_lo;
	const bool span_ab = P.free_sweep && (unsigned)(t.x_last - t.x_first) <= (unsigned)(P.dq_lim - P.bw);
	sweep_staged_lut(t.B, i0, t.lo_b > i0 ? t.lo_b - i0 : 0, no_check, no_check && span_ab, stage, P, t.best_b, t.arg_b);
	__builtin_amdgcn_wave_barrier();
}

template <bool TRACK>
__device__ __forceinline__ void run_chunk_pairs(const DevBatch &b, const DevParams &P, const int *lut, int4 *stage, const int cs, const int ce)
{
	const int lane = lane_id();
	Keep keep; keep.idx = -1; keep.x = keep.hi = keep.y = keep.tag = keep.f = 0;
	auto f_old = [&](int jj) { return b.f[jj]; };
	for (int i0 = cs; i0 < ce; i0 += 2 * WAVE) {
		TilePair t = load_pair(b, i0, ce);
		int jb = cs + ((t.lo_a - cs) & ~(WAVE - 1));
		if (jb < i0) {
			const int eq_lo = equal_x_run_start(b, cs, i0, first_lane(t.A.x));
			int sf = b.f[jb + lane], sq = a_span(b, jb + lane);
			for (; jb < i0; jb += WAVE) {
				// next block's scores are requested before this block is consumed
				const int jn = jb + WAVE < i0 ? jb + WAVE + lane : jb + lane;
				const int nf = b.f[jn], nq = a_span(b, jn);
				sweep_pair_block(b, t, jb, eq_lo, sf, sq, stage, P);
				sf = nf; sq = nq;
			}
		}
		const Target TA = load_target(b, i0, ce, TRACK);
		in_tile<MODE_LUT, TRACK>(b, TA, i0, t.n_a, P, lut, stage, t.best_a, t.arg_a, keep, f_old);
		const int f_a = t.arg_a < 0 ? TA.q : t.best_a;
		if (TA.live) {
			const int i = i0 + lane;
			b.f[i] = f_a;
			b.p[i] = t.arg_a < 0 ? 0 : i - t.arg_a;
		}
		if (t.n_b == 0) break;
		sweep_a_into_b(b, t, cs, i0, f_a, TA.q, stage, P);
		const Target TB = load_target(b, i0 + WAVE, ce, TRACK);
		in_tile<MODE_LUT, TRACK>(b, TB, i0 + WAVE, t.n_b, P, lut, stage, t.best_b, t.arg_b, keep, f_old);
		if (TB.live) {
			const int i = i0 + WAVE + lane;
			b.f[i] = t.arg_b < 0 ? TB.q : t.best_b;
			b.p[i] = t.arg_b < 0 ? 0 : i - t.arg_b;
		}
	}
}

// ---- team mode: the waves of a team (4, 8 or 16) pipeline the tiles of ONE heavy chunk ------------------------
// Wave w takes tiles w, w+16, ...  A tile's sources are swept oldest first, so the only sources that may not be final
// yet are the most recent tiles': the sweep reaches them last, and by then the waves ahead have normally finished
// (a tile needs ~(window/64 + 2) block sweeps, its critical dependency is 2 of them).  Final scores travel between
// waves through an LDS ring indexed by anchor number (the sliding predecessor window, max_iter + slack entries);
// "tiles done" is a release/acquire counter in LDS.  No block barrier inside a chunk.
struct CoopShared { int done; int keep[6]; int chunk; int bar_count; int bar_gen; int part; };   // one per team (part: Progress)
// Gangs (several workgroups on one chunk, gang_chunk_pairs) are an instantiation of their own, k_score<MODE_LUT, false, true>: the gang code
// costs the plain kernel registers (250 -> 283 spilled scalars) and 1-3 % at 500 M anchors, where no chunk gets a gang anyway; the host
// launches it for the micro-batches small enough to end with their largest chunks (Engine: gang_max_n).
// what crosses workgroups (gangs, the SPLIT build): agent-scope accesses -- they go past the caches that are not coherent between CUs / XCDs
#define MM2GB_AGENT __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ int  gload(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, MM2GB_AGENT); }
__device__ __forceinline__ void gstore(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, MM2GB_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
constexpr int SMALL_TEAM = 4, N_SMALL_TEAMS = SCORE_THREADS / WAVE / SMALL_TEAM;
constexpr int N_TEAM_RECORDS = N_SMALL_TEAMS + 3;      // four small teams, two big ones, the whole workgroup
constexpr int TAB_INTS = 36;                           // behind the team records: 24 ints of the SPLIT build's strip table, 9 of a gang's turns

// Barrier among the waves of one small team (a workgroup barrier would stall the other teams): sense-reversing counter
// in LDS, one lane per wave takes part.
__device__ __forceinline__ void team_barrier(CoopShared *sh, int team_size)
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	if (lane_id() == 0) {
		const int gen = __hip_atomic_load(&sh->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (__hip_atomic_fetch_add(&sh->bar_count, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == team_size - 1) {
			__hip_atomic_store(&sh->bar_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			__hip_atomic_fetch_add(&sh->bar_gen, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		} else {
			while (__hip_atomic_load(&sh->bar_gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == gen) __builtin_amdgcn_s_sleep(2);
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The last block of a single tile's sweep while the tile it holds is still in its in-tile phase (Progress): a quarter of its sources at a
// time, as the owner hands them out; the edge sweep (window start, dq range; dr <= 0 through the table address) does every quarter.
// sf_of(): this wave's lane of the block's scores as they stand in the ring.  wait_q(q): until quarter q is out (or the whole tile).
template <class TT, class SF, class WQ>
__device__ __forceinline__ void sweep_block_by_quarters(const DevBatch &b, const TT &T, const int jb, const int k_from, const int sq, int4 *stage, const DevParams &P,
                                                        int &best, int &arg, SF &&sf_of, WQ &&wait_q)
{
	const int xs = a_x(b, jb + lane_id()), ys = a_y(b, jb + lane_id());
	const int tx4 = (int)(((unsigned)T.x - 1u) << 2), ty4 = (int)(((unsigned)T.y - 1u) << 2);
	for (int q = 0; q < 4; ++q) {
		wait_q(q);
		const int lo = max(k_from, 16 * q), hi = 16 * q + 16;
		if (lo >= hi) continue;
		stage_block_lut(xs, ys, sf_of(), sq, stage);          // (the quarters that are not out yet hold stale scores: nobody reads them)
		int bestv = best << 7;
		if (P.edge_prefix && edge_starts_sorted(T.st)) sweep_block_lut_edge_sorted(T.st, tx4, ty4, jb, lo, stage, P, bestv, hi);
		else sweep_block_lut_edge(T.st, tx4, ty4, jb, lo, stage, P, bestv, hi);
		const int won = bestv & 127;
		arg = (unsigned)(won - 1) < (unsigned)WAVE ? jb + won - 1 : arg;
		best = bestv >> 7;
		__builtin_amdgcn_wave_barrier();
	}
}

template <int MODE, bool TRACK>
__device__ __forceinline__ void coop_chunk(const DevBatch &b, const DevParams &P, const int *lut, int4 *stage, int *ring, const int n_slots, CoopShared *sh,
                           const int cs, const int ce, const int wave, const int n_waves)
{
	const int lane = lane_id();
	const int n_tiles = (ce - cs + WAVE - 1) / WAVE;
	auto wait_done = [&](int need) {
		// a waiting wave must not steal issue slots from the waves it waits for: poll rarely (s_sleep 32 = 2048 cycles,
		// a few percent of the shortest tile)
		while (first_lane(__hip_atomic_load(&sh->done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) __builtin_amdgcn_s_sleep(MM2GB_POLL_SLEEP);
	};
	for (int t = wave; t < n_tiles; t += n_waves) {
		const int i0 = cs + t * WAVE;
		const Target T = load_target(b, i0, ce, TRACK);
		const int n_here = min(WAVE, ce - i0);
		int best = T.q + 1, arg = -1;
		const int tile_lo = first_lane(T.st);
		const int st_hi = bcast(T.st, n_here - 1);
		int jb = cs + ((tile_lo - cs) & ~(WAVE - 1));
		const int eq_lo = MODE == MODE_LUT && jb < i0 ? equal_x_run_start(b, cs, i0, first_lane(T.x)) : i0;
		// tile k of the chunk lives in ring slot k mod n_slots (64 scores per slot; the planner made sure the slots cover
		// this chunk's widest window plus the tile being written)
		int slot = (int)((unsigned)((jb - cs) / WAVE) % (unsigned)n_slots);
		for (; jb < i0; jb += WAVE) {
			const int sq = MODE == MODE_LUT ? a_span(b, jb + lane) : 0;
			const int k = (jb - cs) / WAVE, k_from = tile_lo > jb ? tile_lo - jb : 0;
			wait_done(k + 1);                                          // that tile's scores are in the ring
			const int sf = ring[slot * WAVE + lane];
			slot = slot + 1 == n_slots ? 0 : slot + 1;
			sweep_any<MODE>(b, T, jb, k_from, sf, sq, jb >= st_hi && jb + WAVE <= eq_lo, stage, P, lut, best, arg);
		}
		const int my_slot = (int)((unsigned)t % (unsigned)n_slots);
		wait_done(t);                                                    // every earlier tile is final
		Keep keep;
		if (TRACK) { keep.idx = first_lane(sh->keep[0]); keep.x = first_lane(sh->keep[1]); keep.hi = first_lane(sh->keep[2]); keep.y = first_lane(sh->keep[3]); keep.tag = first_lane(sh->keep[4]); keep.f = first_lane(sh->keep[5]); }
		else { keep.idx = -1; keep.x = keep.hi = keep.y = keep.tag = keep.f = 0; }
		// (scores are not handed out a quarter at a time here, as a gang does: a chunk on ONE workgroup is bound by what its CU can sweep, the
		// wave that would take them is busy, and the extra code costs every other path registers -- measured: no gain, 20 M anchors 7.3 -> 8.3 ms)
		in_tile<MODE, TRACK>(b, T, i0, n_here, P, lut, stage, best, arg, keep,
		                     [&](int jj) { const unsigned d = (unsigned)(jj - cs); return ring[(d / WAVE) % (unsigned)n_slots * WAVE + d % WAVE]; });
		const int i = i0 + lane;
		const int fi = arg < 0 ? T.q : best;
		if (T.live) {
			ring[my_slot * WAVE + lane] = fi;
			b.f[i] = fi;
			b.p[i] = arg < 0 ? 0 : i - arg;
		}
		if (TRACK && lane == 0) { sh->keep[0] = keep.idx; sh->keep[1] = keep.x; sh->keep[2] = keep.hi; sh->keep[3] = keep.y; sh->keep[4] = keep.tag; sh->keep[5] = keep.f; }
		// publish: ring + keep writes above are ordered before the counter by the release
		if (lane == 0) __hip_atomic_store(&sh->done, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
}

// Team mode of the table build: waves take PAIRS of tiles round-robin (run_chunk_pairs explains the pair).  Tile A is published as
// soon as it is final, so the chain through the tiles of the chunk is as long as with single tiles.
template <bool TRACK>
__device__ __forceinline__ void coop_chunk_pairs(const DevBatch &b, const DevParams &P, const int *lut, int4 *stage, int *ring, const int n_slots, CoopShared *sh,
                                 const int cs, const int ce, const int wave, const int n_waves)
{
	const int lane = lane_id();
	const int n_tiles = (ce - cs + WAVE - 1) / WAVE;
	auto wait_done = [&](int need) {
		while (first_lane(__hip_atomic_load(&sh->done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) __builtin_amdgcn_s_sleep(MM2GB_POLL_SLEEP);
	};
	// The ring holds the team's most recent n_slots tiles (tile k in slot k mod n_slots; nothing newer than the tile being worked on can
	// have been written: a tile is final only after all before it).  A window that reaches further back takes the older scores from
	// global memory, where every final score goes as well -- written by waves of this workgroup before they published the tile, read
	// behind the acquire of that publication, one block ahead of its use.
	int cur = 0;                                                 // the tile whose in-tile phase is running (for f_old)
	// (the ring through an LDS pointer, not a generic one: a choice between a generic LDS address and a global one becomes a flat load of a
	// selected address, whose cast the compiler gets wrong -- "Illegal instruction detected: Operand has incorrect register class", ROCm 7.2)
	const lds_i32_ptr ring_l = (lds_i32_ptr)(uintptr_t)(unsigned)(uintptr_t)ring;
	auto f_old = [&](int jj) {
		const unsigned d = (unsigned)(jj - cs);
		const int k = (int)(d / WAVE);
		return k >= cur - n_slots ? ring_l[(unsigned)k % (unsigned)n_slots * WAVE + d % WAVE] : b.f[jj];
	};
	for (int pr = wave; 2 * pr < n_tiles; pr += n_waves) {
		const int ta = 2 * pr, i0 = cs + ta * WAVE;              // tile A = tile ta of the chunk, tile B = ta + 1
		TilePair t = load_pair(b, i0, ce);
		int jb = cs + ((t.lo_a - cs) & ~(WAVE - 1));
		const int eq_lo = jb < i0 ? equal_x_run_start(b, cs, i0, first_lane(t.A.x)) : i0;
		int slot = (int)((unsigned)((jb - cs) / WAVE) % (unsigned)n_slots);
		const int first_in_ring = ta - n_slots;                      // tiles from this one on are in the ring
		int nf = 0;
		if (jb < i0 && (jb - cs) / WAVE < first_in_ring) { wait_done((jb - cs) / WAVE + 1); nf = b.f[jb + lane]; }
		for (; jb < i0; jb += WAVE) {
			const int sq = a_span(b, jb + lane);
			const int k = (jb - cs) / WAVE;
			wait_done(k + 1);                                          // that tile's scores are final (and in the ring, if recent)
			int sf;
			if (k < first_in_ring) {
				sf = nf;
				if (k + 1 < first_in_ring) { wait_done(k + 2); nf = b.f[jb + WAVE + lane]; }
			} else sf = ring[slot * WAVE + lane];
			slot = slot + 1 == n_slots ? 0 : slot + 1;
			sweep_pair_block(b, t, jb, eq_lo, sf, sq, stage, P);
		}
		const int slot_a = (int)((unsigned)ta % (unsigned)n_slots), slot_b = slot_a + 1 == n_slots ? 0 : slot_a + 1;
		wait_done(ta);                                               // every earlier tile is final
		Keep keep;
		if (TRACK) { keep.idx = first_lane(sh->keep[0]); keep.x = first_lane(sh->keep[1]); keep.hi = first_lane(sh->keep[2]); keep.y = first_lane(sh->keep[3]); keep.tag = first_lane(sh->keep[4]); keep.f = first_lane(sh->keep[5]); }
		else { keep.idx = -1; keep.x = keep.hi = keep.y = keep.tag = keep.f = 0; }
		const Target TA = load_target(b, i0, ce, TRACK);
		cur = ta;
		in_tile<MODE_LUT, TRACK>(b, TA, i0, t.n_a, P, lut, stage, t.best_a, t.arg_a, keep, f_old);
		__builtin_amdgcn_s_setprio(MM2GB_INTILE_PRIO);                 // still on the team's critical path: publish A, A into B, B, publish B
		const int f_a = t.arg_a < 0 ? TA.q : t.best_a;
		if (TA.live) {
			const int i = i0 + lane;
			ring[slot_a * WAVE + lane] = f_a;
			b.f[i] = f_a;
			b.p[i] = t.arg_a < 0 ? 0 : i - t.arg_a;
		}
		if (t.n_b > 0) {
			// tile A is final: let the other waves go on (the rescue state stays here, nobody needs it before tile B is done)
			if (lane == 0) __hip_atomic_store(&sh->done, ta + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
			sweep_a_into_b(b, t, cs, i0, f_a, TA.q, stage, P);
			const Target TB = load_target(b, i0 + WAVE, ce, TRACK);
			cur = ta + 1;
			in_tile<MODE_LUT, TRACK>(b, TB, i0 + WAVE, t.n_b, P, lut, stage, t.best_b, t.arg_b, keep, f_old);
			__builtin_amdgcn_s_setprio(MM2GB_INTILE_PRIO);
			if (TB.live) {
				const int i = i0 + WAVE + lane;
				const int f_b = t.arg_b < 0 ? TB.q : t.best_b;
				ring[slot_b * WAVE + lane] = f_b;
				b.f[i] = f_b;
				b.p[i] = t.arg_b < 0 ? 0 : i - t.arg_b;
			}
		}
		if (TRACK && lane == 0) { sh->keep[0] = keep.idx; sh->keep[1] = keep.x; sh->keep[2] = keep.hi; sh->keep[3] = keep.y; sh->keep[4] = keep.tag; sh->keep[5] = keep.f; }
		// publish: ring + keep writes above are ordered before the counter by the release
		if (lane == 0) __hip_atomic_store(&sh->done, ta + (t.n_b > 0 ? 2 : 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		__builtin_amdgcn_s_setprio(0);
	}
}


// ---- a gang: the team pipeline of coop_chunk_pairs across workgroups (chain_dev.h, GangSlot) -----------------------------------
// The chunk's tile pairs are dealt in strips of 16 (pair 16 s + w of strip s is wave w's); a workgroup's wave 0 takes the next strip from
// the slot's counter for all 16 waves (`tab`: the strips of this workgroup's last 8 turns, tab[8] = turns announced so far; `seq` counts
// the workgroup's turns over the whole launch).  Scores: a tile this workgroup wrote in one of its last three strips is read from its
// LDS ring, every other one from global memory, where all scores are stored with agent scope.  "Tiles done": the slot's counter (global,
// published in tile order: a wave makes sure its predecessor's publication is out before its own) and the team record's (LDS: what
// this workgroup knows to be final -- its own tiles the moment they are, the others' when a wave has seen the global counter).
// A wave remembers the largest count it has seen (`known`), so tiles that have long been final cost no look at either.
// The rescue state travels like the scores: through LDS inside a strip, through the slot between strips.
template <bool TRACK>
__device__ __forceinline__ void gang_chunk_pairs(const DevBatch &b, const DevParams &P, const int *lut, int4 *stage, int *ring, const int n_slots, CoopShared *sh,
                                 GangSlot *gs, int *tab, int &seq, const int cs, const int ce, const int wave)
{
	const int lane = lane_id();
	const int n_tiles = (ce - cs + WAVE - 1) / WAVE;
	// One tile per wave and turn, or a pair (GangSlot::tiles_per_wave).  A gang exists because its chunk is bound by the chain through its
	// tiles, and per tile a pair's chain is the longer one: it sweeps TWO fresh blocks (the previous pair's tiles) and tile A into B where a
	// single tile sweeps one -- the pair's halved LDS broadcasts are worth nothing on workgroups that wait for the chain anyway.
	const int tpu = first_lane(gload(&gs->tiles_per_wave)) == 2 ? 2 : 1;
	const int STRIP_TILES = tpu * GANG_STRIP_PAIRS;
	int known = 0, known_global = 0;                               // leading tiles known final (from anywhere / from the global counter itself)
	auto look_global = [&]() { known_global = max(known_global, first_lane(gload(&gs->done))); known = max(known, known_global); };
	auto wait_done = [&](int need) {
		if (known >= need) return;
		for (;;) {
			known = max(known, first_lane(__hip_atomic_load(&sh->done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)));
			if (known >= need) return;
			look_global();
			if (known >= need) { if (lane == 0) __hip_atomic_fetch_max(&sh->done, known, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); return; }
			__builtin_amdgcn_s_sleep(MM2GB_GANG_POLL_SLEEP);           // (a gang's workgroups wait for the chain, not for issue slots: look often)
		}
	};
	auto wait_done_global = [&](int need) { while (known_global < need) { look_global(); if (known_global < need) __builtin_amdgcn_s_sleep(MM2GB_GANG_POLL_SLEEP); } };
	int s0 = -1, s1 = -1, s2 = -1;                                  // this workgroup's strips, newest first
	int cur = 0;
	const lds_i32_ptr ring_l = (lds_i32_ptr)(uintptr_t)(unsigned)(uintptr_t)ring;
	auto in_ring = [&](int k) { const int sk = k / STRIP_TILES; return (sk == s0 || sk == s1 || sk == s2) && k >= cur - n_slots; };
	auto f_old = [&](int jj) {
		const unsigned d = (unsigned)(jj - cs);
		const int k = (int)(d / WAVE);
		return in_ring(k) ? ring_l[(unsigned)k % (unsigned)n_slots * WAVE + d % WAVE] : gload(&b.f[jj]);
	};
	for (;; ++seq) {
		int s = 0;
		if (wave == 0) {
			if (lane == 0) {
				s = __hip_atomic_fetch_add(&gs->next_strip, 1, __ATOMIC_RELAXED, MM2GB_AGENT);
				tab[seq & 7] = s;
				__hip_atomic_store(&tab[8], seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
			s = first_lane(s);
		} else {
			while (first_lane(__hip_atomic_load(&tab[8], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) <= seq) __builtin_amdgcn_s_sleep(2);
			s = first_lane(tab[seq & 7]);
		}
		if (s * STRIP_TILES >= n_tiles) { ++seq; break; }            // no strip left (every wave of the workgroup sees the same turn)
		s2 = s1; s1 = s0; s0 = s;
		const int pr = s * GANG_STRIP_PAIRS + wave;
		if (tpu * pr >= n_tiles) continue;                           // the chunk's last strip is a short one
		const int ta = tpu * pr, i0 = cs + ta * WAVE;                // tile A = tile ta of the chunk, tile B = ta + 1 (pairs only)
		cur = ta;
		TilePair t = load_pair(b, i0, tpu == 2 ? ce : min(ce, i0 + WAVE));   // (one tile per wave: a pair whose tile B is empty)
		int jb = cs + ((t.lo_a - cs) & ~(WAVE - 1));
		const int eq_lo = jb < i0 ? equal_x_run_start(b, cs, i0, first_lane(t.A.x)) : i0;
		bool have = false;
		int nf = 0;
		for (; jb < i0; jb += WAVE) {
			const int sq = a_span(b, jb + lane);
			const int k = (jb - cs) / WAVE;
			if (tpu == 1 && wave > 0 && jb + WAVE == i0 && !P.lut_clamp && known <= k) {
				// The tile before this one is the previous wave's of this workgroup and (as far as this wave knows) still in its in-tile phase:
				// its scores a quarter at a time as they are handed out (Progress) -- three quarters of this block's sweep leave the chain.
				const int rs = (int)((unsigned)k % (unsigned)n_slots);
				sweep_block_by_quarters(b, t.A, jb, t.lo_a > jb ? t.lo_a - jb : 0, sq, stage, P, t.best_a, t.arg_a, [&]() { return ring[rs * WAVE + lane]; },
				                        [&](int q) {
					while (known <= k) {
						known = max(known, first_lane(__hip_atomic_load(&sh->done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)));
						if (known > k || first_lane(__hip_atomic_load(&sh->part, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >= 4 * k + q + 1) break;
						__builtin_amdgcn_s_sleep(MM2GB_QUARTER_POLL_SLEEP);
					}
				});
				have = false;
				continue;
			}
			wait_done(k + 1);                                          // that tile's scores are final
			int sf;
			if (in_ring(k)) sf = ring[(unsigned)k % (unsigned)n_slots * WAVE + lane];
			else sf = have ? nf : gload(&b.f[jb + lane]);
			// the next block's scores one block ahead, if they come from memory and are final already (no waiting ahead: the last
			// blocks of a window are the chunk's newest tiles)
			have = jb + WAVE < i0 && known >= k + 2 && !in_ring(k + 1);
			if (have) nf = gload(&b.f[jb + WAVE + lane]);
			sweep_pair_block(b, t, jb, eq_lo, sf, sq, stage, P);
		}
		const int slot_a = (int)((unsigned)ta % (unsigned)n_slots), slot_b = slot_a + 1 == n_slots ? 0 : slot_a + 1;
		const Target TA = load_target(b, i0, ce, TRACK);             // (asked for before the wait: a memory round trip off the gang's chain)
		wait_done(ta);                                               // every earlier tile is final
		Keep keep;
		keep.idx = -1; keep.x = keep.hi = keep.y = keep.tag = keep.f = 0;
		if (TRACK && ta > 0) {
			if (wave > 0 || s1 == s - 1) {                             // the pair before this one was this workgroup's: its state is in LDS
				keep.idx = first_lane(sh->keep[0]); keep.x = first_lane(sh->keep[1]); keep.hi = first_lane(sh->keep[2]); keep.y = first_lane(sh->keep[3]); keep.tag = first_lane(sh->keep[4]); keep.f = first_lane(sh->keep[5]);
			} else {
				wait_done_global(ta);                                    // the slot's copy was stored before that count
				keep.idx = first_lane(gload(&gs->keep[0])) - 1; keep.x = first_lane(gload(&gs->keep[1])); keep.hi = first_lane(gload(&gs->keep[2]));
				keep.y = first_lane(gload(&gs->keep[3])); keep.tag = first_lane(gload(&gs->keep[4])); keep.f = first_lane(gload(&gs->keep[5]));
			}
		}
		Progress prog;
		if (tpu == 1 && !P.lut_clamp) { prog.ring_slot = ring + slot_a * WAVE; prog.part = &sh->part; prog.base = 4 * ta; }
		in_tile<MODE_LUT, TRACK>(b, TA, i0, t.n_a, P, lut, stage, t.best_a, t.arg_a, keep, f_old, prog);
		__builtin_amdgcn_s_setprio(MM2GB_INTILE_PRIO);
		const int f_a = t.arg_a < 0 ? TA.q : t.best_a;
		if (TA.live) {
			const int i = i0 + lane;
			ring[slot_a * WAVE + lane] = f_a;
			gstore(&b.f[i], f_a);
			b.p[i] = t.arg_a < 0 ? 0 : i - t.arg_a;
		}
		if (t.n_b > 0) {
			if (lane == 0) __hip_atomic_fetch_max(&sh->done, ta + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
			sweep_a_into_b(b, t, cs, i0, f_a, TA.q, stage, P);
			const Target TB = load_target(b, i0 + WAVE, ce, TRACK);
			cur = ta + 1;
			in_tile<MODE_LUT, TRACK>(b, TB, i0 + WAVE, t.n_b, P, lut, stage, t.best_b, t.arg_b, keep, f_old);
			__builtin_amdgcn_s_setprio(MM2GB_INTILE_PRIO);
			if (TB.live) {
				const int i = i0 + WAVE + lane;
				const int f_b = t.arg_b < 0 ? TB.q : t.best_b;
				ring[slot_b * WAVE + lane] = f_b;
				gstore(&b.f[i], f_b);
				b.p[i] = t.arg_b < 0 ? 0 : i - t.arg_b;
			}
		}
		const int now_done = ta + (t.n_b > 0 ? 2 : 1);
		if (TRACK && lane == 0) { sh->keep[0] = keep.idx; sh->keep[1] = keep.x; sh->keep[2] = keep.hi; sh->keep[3] = keep.y; sh->keep[4] = keep.tag; sh->keep[5] = keep.f; }
		if (lane == 0) __hip_atomic_fetch_max(&sh->done, now_done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // this workgroup's waves go on
		__builtin_amdgcn_s_setprio(0);
		// the other workgroups: scores (and the state, at the end of a strip) are out, the publications before this one too, then the count
		if (TRACK && wave == GANG_STRIP_PAIRS - 1 && lane == 0) {
			gstore(&gs->keep[0], keep.idx + 1); gstore(&gs->keep[1], keep.x); gstore(&gs->keep[2], keep.hi); gstore(&gs->keep[3], keep.y); gstore(&gs->keep[4], keep.tag); gstore(&gs->keep[5], keep.f);
		}
		drain_stores();
		wait_done_global(ta);
		if (lane == 0) gstore(&gs->done, now_done);
		known = max(known, now_done); known_global = max(known_global, now_done);
	}
}


// ---- one chunk on several workgroups (SPLIT build) ------------------------------------------------------------------------
// A team is bounded by one CU; the largest chunks of a batch that cannot fill the machine decide when it ends.  Such a chunk (one that
// the planner gives a whole workgroup, its OWNER) is scored strip by strip, a strip being 16 tiles, one per wave.  For every strip:
//   phase A  the sweeps over the sources BEFORE the strip -- all final -- are cut into items (a tile pair x some source blocks) in a
//            slot in global memory; the owner's waves take items, and so does every wave of the launch that has nothing else left to
//            do (help_split_chunks, at the end of the kernel).  An item leaves per target the best score and its source;
//   phase B  wave w combines tile w's items in source order (a later source wins ties, as everywhere), then goes through the tiles
//            of its own strip that precede it -- scores through the LDS ring -- and the in-tile phase, as a team does (coop_chunk).
// Nothing waits for a workgroup that may not be running: the owner takes items itself and only waits for items somebody HAS taken,
// i.e. for waves that are executing them; helpers leave when no workgroup is in the whole-workgroup phase any more.
// What crosses workgroups: the chunk's scores (owner -> whoever takes an item) and the items' partial results (-> owner).  Both are
// PLAIN stores, drained by every storing wave, then ONE agent-scope release by the lane that signals (owner: before it opens a strip;
// item: before it counts itself done) and an agent-scope acquire on the reading side before its plain loads -- the form
// cdna_hip_programming.md Guideline 16 gives as always valid.  (A cheaper form, write-through sc1 stores and sc1 loads with no
// fences, was what the first build used; it was replaced while chasing wrong results that turned out to be the compiler problem noted at
// split_do_item_owner, and has not been tried again since.)  The slot's words are agent-scope atomics.  x, y, tag, st are not written
// in this launch.
__device__ __forceinline__ unsigned long long uni64(unsigned long long v)
{
	return (unsigned long long)(unsigned)first_lane((int)(v >> 32)) << 32 | (unsigned)first_lane((int)v);
}

// Item `it` of workgroup `wg`'s open strip.  Any wave of the launch, whole-wave (uniform) control flow only.
__device__ __forceinline__ void split_do_item(const DevBatch &b, const DevParams &P, int4 *stage, const int wg, const int it)
{
	SplitSlot *slot = b.split_slots + wg;
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");            // the scores the owner released when it opened this strip
	// the slot's fields belong to this strip until every item that was handed out is done
	// (plain loads behind the acquire, like the scores: the owner wrote them with plain stores before its release)
	const volatile SplitSlot *vs = slot;
	const int cs = first_lane(vs->cs), ce = first_lane(vs->ce), i_s = first_lane(vs->i_s), per = first_lane(vs->blocks_per_item);
	int p = 0;
	for (int q = 1; q < 8; ++q) if (it >= first_lane(vs->base[q])) p = q;
	const int k = it - first_lane(vs->base[p]);
	const int jb0 = first_lane(vs->jbs[p]) + k * per * WAVE, jb1 = min(i_s, jb0 + per * WAVE);
	const int i0 = i_s + p * 2 * WAVE;
	TilePair t = load_pair(b, i0, ce);
	const int eq_lo = equal_x_run_start(b, cs, i0, first_lane(t.A.x));
	for (int jb = jb0; jb < jb1; jb += WAVE) {
		const int sf = b.f[jb + lane_id()], sq = a_span(b, jb + lane_id());
		sweep_pair_block(b, t, jb, eq_lo, sf, sq, stage, P);
	}
	unsigned long long *part = b.split_part + ((size_t)wg * SPLIT_MAX_ITEMS + it) * 2 * WAVE;
	part[lane_id()] = (unsigned long long)(unsigned)t.best_a << 32 | (unsigned)t.arg_a;
	part[WAVE + lane_id()] = (unsigned long long)(unsigned)t.best_b << 32 | (unsigned)t.arg_b;
	drain_stores();
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
	drain_stores();
	if (lane_id() == 0) {
		__hip_atomic_fetch_add(&slot->done, 1, __ATOMIC_RELAXED, MM2GB_AGENT);
		if (wg != (int)blockIdx.x) atomicAdd(&b.counters[CNT_HELPED], 1);
	}
}
// The same for the owner's own waves, as a function of its own.  Inlined into split_chunk the item code was miscompiled (ROCm 7.2): items
// that contain a range-tested block came out with lanes missing -- every run of profiles/split_soak.py with a build without helpers
// wrong from the first such strip on, whatever the memory protocol -- while the copy inlined into help_split_chunks and this
// out-of-line copy give the oracle's results.  The call costs the owner little: with helpers around it takes few items itself.
__device__ __attribute__((noinline)) void split_do_item_owner(const DevBatch &b, const DevParams &P, int4 *stage, const int it)
{
	split_do_item(b, P, stage, (int)blockIdx.x, it);
}
// Take the next item of workgroup `wg`'s open strip: its number, or -1 when none is left.  (The same shape as the work cursors of
// k_score: one lane adds, the value is made wave-uniform.)
__device__ __forceinline__ int split_claim(const DevBatch &b, const int wg)
{
	unsigned long long w = 0;
	if (lane_id() == 0) w = __hip_atomic_fetch_add(&b.split_slots[wg].word, 1ull, __ATOMIC_RELAXED, MM2GB_AGENT);
	w = uni64(w);
	const int it = (int)(unsigned)w, total = (int)(w >> 32);
	return it < total ? it : -1;
}

// A wave with nothing else to do: items of any workgroup's open strip, until no workgroup is in the whole-workgroup phase any more.
__device__ __forceinline__ void help_split_chunks(const DevBatch &b, const DevParams &P, int4 *stage)
{
	unsigned idle = 0;
	for (;;) {
		const int open = first_lane(gload(&b.counters[CNT_SPLIT_OPEN]));       // read BEFORE the look at the slots
		bool any = false;
		// one word says whether any strip is open at all: that is all an idle wave reads, every few microseconds
		if (first_lane(gload(&b.counters[CNT_SPLIT_ANY])) > 0)
		for (int base = 0; base < (int)gridDim.x; base += WAVE) {
			const int wg = base + lane_id();
			unsigned long long w = 0;
			if (wg < (int)gridDim.x) w = __hip_atomic_load(&b.split_slots[wg].word, __ATOMIC_RELAXED, MM2GB_AGENT);
			unsigned long long has = __ballot((unsigned)w < (unsigned)(w >> 32));
			while (has) {
				const int sel = base + __builtin_ctzll(has);
				const int it = split_claim(b, sel);
				if (it >= 0) { split_do_item(b, P, stage, sel, it); any = true; }
				else has &= has - 1;
			}
		}
		if (!any) {
			if (open <= 0 || ++idle > (1u << 22)) return;             // (the bound: seconds; helping is optional, hanging is not)
			__builtin_amdgcn_s_sleep(64);
		}
	}
}

// The owner: all 16 waves of the workgroup, chunk [cs, ce).  tab: 24 ints of LDS.
template <bool TRACK>
__device__ __forceinline__ void split_chunk(const DevBatch &b, const DevParams &P, const int *lut, int4 *stage, int *ring, CoopShared *sh, int *tab,
                                            const int cs, const int ce, const int wave)
{
	constexpr int S = SPLIT_STRIP_TILES;
	const int lane = lane_id();
	SplitSlot *slot = b.split_slots + blockIdx.x;
	const unsigned long long *part = b.split_part + (size_t)blockIdx.x * SPLIT_MAX_ITEMS * 2 * WAVE;
	const int n_tiles = (ce - cs + WAVE - 1) / WAVE;
	auto wait_done = [&](int need) {
		while (first_lane(__hip_atomic_load(&sh->done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) __builtin_amdgcn_s_sleep(MM2GB_POLL_SLEEP);
	};
	if (wave == 0 && lane == 0) atomicAdd(&b.counters[CNT_NSPLIT], 1);
	for (int s0 = 0; s0 < n_tiles; s0 += S) {
		const int i_s = cs + s0 * WAVE;
		// ---- phase A: the sources before the strip ----
		if (wave < 8 && lane == 0) {
			const int i0 = i_s + wave * 2 * WAVE;
			int jbs = i_s, nblk = 0;
			if (s0 > 0 && i0 < ce) {
				const int lo = b.st[i0];                           // window starts are monotone: the pair's first anchor has the smallest
				jbs = cs + ((lo - cs) & ~(WAVE - 1));
				nblk = jbs < i_s ? (i_s - jbs) / WAVE : 0;
			}
			tab[wave] = jbs; tab[8 + wave] = nblk;
		}
		if (wave == 0 && lane == 0) __hip_atomic_store(&sh->done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		team_barrier(sh, S);
		int total = 0, per = 8;
		{
			int widest = 0;
			for (int q = 0; q < 8; ++q) widest = max(widest, first_lane(tab[8 + q]));
			per = max(8, (widest + 15) / 16);                       // at most 16 items per pair
			for (int q = 0; q < 8; ++q) total += (first_lane(tab[8 + q]) + per - 1) / per;
		}
		if (total > 0) {
			if (wave == 0 && lane == 0) {
				volatile SplitSlot *vs = slot;
				vs->cs = cs; vs->ce = ce; vs->i_s = i_s; vs->blocks_per_item = per; gstore(&slot->done, 0);
				int acc = 0;
				for (int q = 0; q < 8; ++q) { vs->jbs[q] = tab[q]; vs->base[q] = acc; acc += (tab[8 + q] + per - 1) / per; }
				vs->base[8] = acc;
				drain_stores();
				// the scores of the strips before this one: every wave drained its stores before the barrier above
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
				drain_stores();
				__hip_atomic_store(&slot->word, (unsigned long long)total << 32, __ATOMIC_RELAXED, MM2GB_AGENT);    // open
				atomicAdd(&b.counters[CNT_SPLIT_ANY], 1);
			}
			team_barrier(sh, S);
			for (;;) {
				const int it = split_claim(b, (int)blockIdx.x);
				if (it < 0) break;
				split_do_item_owner(b, P, stage, it);
			}
			// every item is handed out; wait for the ones still being worked on (by waves that are running: no wait for anybody's turn)
			if (lane == 0) {
				unsigned spins = 0;
				while (__hip_atomic_load(&slot->done, __ATOMIC_RELAXED, MM2GB_AGENT) < total) {
					__builtin_amdgcn_s_sleep(8);
					if (++spins > (1u << 26)) __builtin_trap();       // minutes: something is broken; fail loudly rather than hang
				}
			}
			team_barrier(sh, S);
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the partial results (every wave reads some)
			if (wave == 0 && lane == 0) { __hip_atomic_store(&slot->word, 0ull, __ATOMIC_RELAXED, MM2GB_AGENT); atomicAdd(&b.counters[CNT_SPLIT_ANY], -1); }   // closed
		}
		// ---- phase B: the strip itself, wave w its tile w ----
		const int tix = s0 + wave;
		if (tix < n_tiles) {
			const int i0 = cs + tix * WAVE;
			const Target T = load_target(b, i0, ce, TRACK);
			const int n_here = min(WAVE, ce - i0);
			int best = T.q + 1, arg = -1;
			if (total > 0) {
				const int q = wave >> 1, side = wave & 1;
				int first = 0;
				for (int r = 0; r < q; ++r) first += (first_lane(tab[8 + r]) + per - 1) / per;
				const int n_it = (first_lane(tab[8 + q]) + per - 1) / per;
				for (int k = 0; k < n_it; ++k) {                     // ascending sources: a later one wins ties
					const unsigned long long v = part[((size_t)(first + k) * 2 + side) * WAVE + lane];
					const int pb = (int)(unsigned)(v >> 32), pa = (int)(unsigned)v;
					if (pa >= 0 && pb >= best) { best = pb; arg = pa; }
				}
			}
			const int tile_lo = first_lane(T.st), st_hi = bcast(T.st, n_here - 1);
			int jb = cs + ((tile_lo - cs) & ~(WAVE - 1));
			if (jb < i_s) jb = i_s;
			const int eq_lo = jb < i0 ? equal_x_run_start(b, cs, i0, first_lane(T.x)) : i0;
			for (; jb < i0; jb += WAVE) {
				const int src = (jb - i_s) / WAVE;                  // tile of this strip, one of the waves before this one
				const int sq = a_span(b, jb + lane);
				wait_done(src + 1);
				const int sf = ring[src * WAVE + lane];
				const int k_from = tile_lo > jb ? tile_lo - jb : 0;
				sweep_any<MODE_LUT>(b, T, jb, k_from, sf, sq, jb >= st_hi && jb + WAVE <= eq_lo, stage, P, lut, best, arg);
			}
			wait_done(wave);                                         // every earlier tile of the strip is final (earlier strips: the barrier)
			Keep keep;
			if (TRACK) { keep.idx = first_lane(sh->keep[0]); keep.x = first_lane(sh->keep[1]); keep.hi = first_lane(sh->keep[2]); keep.y = first_lane(sh->keep[3]); keep.tag = first_lane(sh->keep[4]); keep.f = first_lane(sh->keep[5]); }
			else { keep.idx = -1; keep.x = keep.hi = keep.y = keep.tag = keep.f = 0; }
			in_tile<MODE_LUT, TRACK>(b, T, i0, n_here, P, lut, stage, best, arg, keep, [&](int jj) { return b.f[jj]; });
			const int i = i0 + lane;
			const int fi = arg < 0 ? T.q : best;
			if (T.live) {
				ring[wave * WAVE + lane] = fi;
				b.f[i] = fi;                                          // other workgroups read it in later strips' items (released when a strip opens)
				b.p[i] = arg < 0 ? 0 : i - arg;
			}
			if (TRACK && lane == 0) { sh->keep[0] = keep.idx; sh->keep[1] = keep.x; sh->keep[2] = keep.hi; sh->keep[3] = keep.y; sh->keep[4] = keep.tag; sh->keep[5] = keep.f; }
			drain_stores();
			if (lane == 0) __hip_atomic_store(&sh->done, wave + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		team_barrier(sh, S);
	}
}

// One phase of team work: the workgroup's waves form teams of team_size (4, 8 or 16) that pull chunks from `list`.
// first: a chunk (position in the list) already pulled for team 0 by the previous phase, or -1.
// min_cost: a chunk cheaper than this ends the phase for the team that pulled it; its position is returned (else -1) so
// that the next phase can start with it.  Only meaningful for a one-team phase (whole workgroup).
template <int MODE, bool SPLIT, bool GANG>
__device__ __forceinline__ int team_phase(const DevBatch &b, const DevParams &P, const int *lut, int4 *stage, int *ring, const int ring_slots, CoopShared *teams,
                          const int32_t *list, const int n_list, const int cursor, const int wave, const int team_size,
                          int first = -1, const long long min_cost = 0, int *split_tab = nullptr)
{
	const int n_teams = SCORE_THREADS / WAVE / team_size;
	const int team = wave / team_size, team_wave = wave - team * team_size;
	const int slots = ring_slots / n_teams;
	int *my_ring = ring + team * slots * WAVE;
	CoopShared *sh = &teams[team];
	while (true) {
		const bool given = first >= 0 && team == 0;
		if (team_wave == 0 && lane_id() == 0) { sh->chunk = given ? first : atomicAdd(&b.counters[cursor], 1); sh->done = 0; sh->part = 0; sh->keep[0] = -1; }
		first = -1;
		team_barrier(sh, team_size);
		const int c = first_lane(sh->chunk);
		if (c >= n_list) return -1;
		const int ci = first_lane(list[c]);
		if (min_cost > 0 && b.chunk_cost[ci] < min_cost) return c;
		if (GANG && (b.chunk_track[ci] & 8)) { team_barrier(sh, team_size); continue; }   // a gang's chunk (phase 0 of k_score)
		const int cs = first_lane(b.chunk_start[ci]), ce = first_lane(b.chunk_end[ci]);
		// whole-workgroup teams keep one tile per wave: with two, 32 tiles of one chunk would be in flight and the largest
		// chunks -- the ones that decide when a small batch ends -- ran 6 % slower
		if (SPLIT && MODE == MODE_LUT && team_size == SCORE_THREADS / WAVE) {
			if (b.chunk_track[ci] & 1) split_chunk<true>(b, P, lut, stage, my_ring, sh, split_tab, cs, ce, team_wave);
			else split_chunk<false>(b, P, lut, stage, my_ring, sh, split_tab, cs, ce, team_wave);
		} else if (MODE == MODE_LUT && team_size < SCORE_THREADS / WAVE) {
			if (b.chunk_track[ci] & 1) coop_chunk_pairs<true>(b, P, lut, stage, my_ring, slots, sh, cs, ce, team_wave, team_size);
			else coop_chunk_pairs<false>(b, P, lut, stage, my_ring, slots, sh, cs, ce, team_wave, team_size);
		} else {
			if (b.chunk_track[ci] & 1) coop_chunk<MODE, true>(b, P, lut, stage, my_ring, slots, sh, cs, ce, team_wave, team_size);
			else coop_chunk<MODE, false>(b, P, lut, stage, my_ring, slots, sh, cs, ce, team_wave, team_size);
		}
		team_barrier(sh, team_size);
	}
}

// --------------------------------------------------------------------------------------------------------------
// The score kernel: persistent 1024-thread workgroups (16 waves).  Phase 1a: the big-team list (wide-window heavy chunks) --
// first the chunks that are larger than a workgroup's fair share of it with all 16 waves, then two 8-wave teams per
// workgroup.  Phase 1b: four 4-wave teams per workgroup pull narrower heavy chunks.  Phase 2: every wave pulls ordinary
// chunks on its own.  All lists most expensive first; a team enters the next phase as soon as its list is empty.
// Exactly one MODE instance does the work of a batch (mode_sel picks it from the host's parameters and the
// "some anchor carries a segment id" flag found on the device by k_window).
// LDS layout (dynamic): [ ring : ring_slots x 64 ints ][ stage : 16 waves x 64 x int4 ][ CoopShared x 7 ] and, MODE_LUT only, the penalty
// table (bw + 2 entries) from P.lut_base to the end of the allocation, LUT_LDS_TOTAL (chain_dev.h)
// --------------------------------------------------------------------------------------------------------------

template <int MODE, bool SPLIT, bool GANG>
__global__ __launch_bounds__(SCORE_THREADS, 8) void k_score(DevBatch b, DevParams P, int host_mode, int ring_slots, int big_team, int whole_wg_pct)
{
	extern __shared__ __attribute__((aligned(16))) int smem[];
	const unsigned fl = b.flags[0];
	const int mode = (fl & FLAG_ANY_SEGID) ? MODE_GENERAL : (host_mode == MODE_LUT && (fl & FLAG_NO_LUT)) ? MODE_FAST : host_mode;
	if (mode != MODE) return;
	int *lut = smem + (MODE == MODE_LUT ? P.lut_base / 4 : 0);
	// the table sweep addresses the penalty table by raw LDS offset: the dynamic allocation must start at LDS address 0, and what
	// comes before the table must end before it
	if (MODE == MODE_LUT && ((unsigned)(uintptr_t)(__attribute__((address_space(3))) int*)smem != 0u ||
	                         (size_t)ring_slots * WAVE * 4 + SCORE_THREADS * sizeof(int4) + N_TEAM_RECORDS * sizeof(CoopShared) + TAB_INTS * 4 > (size_t)LUT_LDS_BASE ||
	                         P.lut_base < LUT_LDS_BASE || P.lut_base + 4 * (P.lut_last + 1) != LUT_LDS_TOTAL)) __builtin_trap();
	int *ring = smem;
	int4 *stage = (int4*)(ring + ring_slots * WAVE) + (threadIdx.x / WAVE) * WAVE;   // this wave's scratch
	CoopShared *teams = (CoopShared*)((int4*)(ring + ring_slots * WAVE) + SCORE_THREADS);   // N_SMALL_TEAMS of them
	if (MODE == MODE_LUT) for (int k = threadIdx.x; k <= P.lut_last; k += SCORE_THREADS) lut[k] = b.lut[k];    // entry lut_last = bw + 1 is 0: reject
	if (threadIdx.x < N_TEAM_RECORDS) { teams[threadIdx.x].bar_count = 0; teams[threadIdx.x].bar_gen = 0; }
	int *split_tab = (int*)(teams + N_TEAM_RECORDS);           // TAB_INTS ints between the team records and the table
	if (threadIdx.x == 0) split_tab[24 + 8] = 0;               // a gang's turns announced so far
	if (SPLIT && threadIdx.x == 0) atomicAdd(&b.counters[CNT_SPLIT_OPEN], 1);   // this workgroup is in the whole-workgroup phase
	__syncthreads();

	const int n_long = first_lane(b.counters[CNT_NLONG]), n_mid = first_lane(b.counters[CNT_NMID]);
	// wave-uniform values are told to the compiler as such (v_readfirstlane): tile loops, window bounds and the rescue state
	// machine then run on the scalar unit and read-only inputs can come through the scalar cache
	const int wave = first_lane(threadIdx.x / WAVE);
	// optional phase stamps (MM2GB_DEBUG_PHASES): 100 MHz wall clock at start / end of 1a / end of 1b / end, per workgroup
	if (b.dbg && threadIdx.x == 0) b.dbg[blockIdx.x * 4 + 0] = (long long)__builtin_amdgcn_s_memrealtime();
	// phase 0: gangs -- the chunks that would outlast the batch on one workgroup are scored by several (gang_chunk_pairs); this workgroup
	// starts on the chunk the planner gave it, if any.
	int gang_seq = 0;
	int *gang_tab = split_tab + 24;
	const int n_gang = (GANG && MODE == MODE_LUT && !SPLIT && b.gang_slots && ring_slots > 0) ? first_lane(b.counters[CNT_NGANG]) : 0;
	if constexpr (GANG && MODE == MODE_LUT && !SPLIT) {
	auto gang_run = [&](int e, bool late) __attribute__((always_inline)) {
		CoopShared *sh = teams + N_SMALL_TEAMS + 2;
		GangSlot *gs = b.gang_slots + e;
		const int ci = first_lane(gs->chunk);
		const int cs = first_lane(b.chunk_start[ci]), ce = first_lane(b.chunk_end[ci]);
		if (threadIdx.x == 0) {
			// a late helper only joins a chunk whose strips are not already handed out far ahead of what is final (those workgroups wait as it is)
			int go = 1;
			if (late) {
				const int next = gload(&gs->next_strip), done = gload(&gs->done), n_tiles = (ce - cs + WAVE - 1) / WAVE;
				const int st_tiles = (gload(&gs->tiles_per_wave) == 2 ? 2 : 1) * GANG_STRIP_PAIRS;
				go = next * st_tiles < n_tiles && next * st_tiles - done < 4 * st_tiles;
			}
			sh->chunk = go; sh->done = 0; sh->part = 0; sh->keep[0] = -1;
		}
		team_barrier(sh, SCORE_THREADS / WAVE);
		const int go = first_lane(sh->chunk);
		if (go) {
			if (b.chunk_track[ci] & 1) gang_chunk_pairs<true>(b, P, lut, stage, ring, ring_slots, sh, gs, gang_tab, gang_seq, cs, ce, wave);
			else gang_chunk_pairs<false>(b, P, lut, stage, ring, ring_slots, sh, gs, gang_tab, gang_seq, cs, ce, wave);
		}
		team_barrier(sh, SCORE_THREADS / WAVE);
	};
	if (n_gang > 0) {
		for (int e = 0; e < n_gang; ++e) {
			const int fw = first_lane(b.gang_slots[e].first_wg), nw = first_lane(b.gang_slots[e].n_wg);
			if ((int)blockIdx.x >= fw && (int)blockIdx.x < fw + nw) { gang_run(e, false); break; }
		}
	}
	}
	if (ring_slots > 0) {
		// phase 1a: big teams (the whole workgroup, or two 8-wave teams) on wide-window heavy chunks; phase 1b: four 4-wave
		// teams on narrower ones.  The ring is split hierarchically (a big team's share is made of its small teams' shares)
		// and every phase has its own team records, so a team moves on without waiting for the rest of the workgroup.
		// The big-team list is served most expensive first.  A chunk that alone is more than whole_wg_pct % of a workgroup's fair
		// share of that list gets the whole workgroup (the largest chunks decide when a small batch ends, and a team's speed
		// is bounded by what one CU's LDS pipe and issue slots give it); the rest go to teams of big_team waves, two at a time.
		// (One loop over the three team phases rather than three calls: the team code is large and is inlined once.)
		int first = -1;
		for (int phase = 0; phase < 3; ++phase) {
			const bool whole = phase == 0, small = phase == 2;
			if (whole && !(big_team < SCORE_THREADS / WAVE && whole_wg_pct > 0)) { if (SPLIT && threadIdx.x == 0) atomicAdd(&b.counters[CNT_SPLIT_OPEN], -1); continue; }
			const long long share = whole ? max(1ll, (long long)(b.totals[2] / gridDim.x * whole_wg_pct / 100)) : 0;
			CoopShared *records = small ? teams : whole ? teams + N_SMALL_TEAMS + 2 : teams + N_SMALL_TEAMS;
			const int got = team_phase<MODE, SPLIT, GANG>(b, P, lut, stage, ring, ring_slots, records, small ? b.mid_list : b.long_list, small ? n_mid : n_long,
			                                        small ? CNT_MCURSOR : CNT_LCURSOR, wave, small ? SMALL_TEAM : whole ? SCORE_THREADS / WAVE : big_team,
			                                        whole || small ? -1 : first, share, split_tab);
			if (whole) first = got;
			if (SPLIT && whole && threadIdx.x == 0) atomicAdd(&b.counters[CNT_SPLIT_OPEN], -1);   // (all 16 waves return together: the team's barrier)
			if (phase == 1 && b.dbg && lane_id() == 0) atomicMax((unsigned long long*)&b.dbg[blockIdx.x * 4 + 1], (unsigned long long)__builtin_amdgcn_s_memrealtime());
		}
	}
	else if (SPLIT && threadIdx.x == 0) atomicAdd(&b.counters[CNT_SPLIT_OPEN], -1);
	if (b.dbg && lane_id() == 0) atomicMax((unsigned long long*)&b.dbg[blockIdx.x * 4 + 2], (unsigned long long)__builtin_amdgcn_s_memrealtime());
	// phase 2: one wave per chunk
	const int n_chunks = b.counters[CNT_NCHUNK] - n_long - n_mid;
	while (true) {
		int c = 0;
		if (lane_id() == 0) c = atomicAdd(&b.counters[CNT_CURSOR], 1);
		c = first_lane(c);
		if (c >= n_chunks) break;
		const int ci = first_lane(b.order[c]);
		const int cs = first_lane(b.chunk_start[ci]), ce = first_lane(b.chunk_end[ci]);
		if (MODE == MODE_LUT) {
			if (b.chunk_track[ci] & 1) run_chunk_pairs<true>(b, P, lut, stage, cs, ce);
			else run_chunk_pairs<false>(b, P, lut, stage, cs, ce);
		} else {
			if (b.chunk_track[ci] & 1) run_chunk<MODE, true>(b, P, lut, stage, cs, ce);
			else run_chunk<MODE, false>(b, P, lut, stage, cs, ce);
		}
	}
	// nothing left of its own: items of chunks that other workgroups score strip by strip
	if (SPLIT && MODE == MODE_LUT) help_split_chunks(b, P, stage);
	if (b.dbg && lane_id() == 0) atomicMax((unsigned long long*)&b.dbg[blockIdx.x * 4 + 3], (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

// --------------------------------------------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------------------------------------------
void launch_window(const DevBatch &b, const DevParams &P, hipStream_t s)
{
	if (b.n <= 0) return;
	hipLaunchKernelGGL(k_block_reads, dim3((unsigned)((b.n_blocks + 255) / 256)), dim3(256), 0, s, b);
	hipLaunchKernelGGL(k_window, dim3((unsigned)b.n_blocks), dim3(PLAN_THREADS), 0, s, b, P);
}

void launch_plan(const DevBatch &b, const LaunchCfg &cfg, hipStream_t s)
{
	const int n_tiles = (int)((b.n_blocks + PLANNER_THREADS - 1) / PLANNER_THREADS);
	const int chunk_grid = (int)std::min<int64_t>(1024, (b.n_blocks + 255) / 256);
	(void)hipMemsetAsync(b.bins, 0, N_LISTS * COST_BINS * sizeof(int), s);
	hipLaunchKernelGGL(plan_tile_sums, dim3(n_tiles), dim3(PLANNER_THREADS), 0, s, b);
	hipLaunchKernelGGL(plan_tile_scan, dim3(1), dim3(PLANNER_THREADS), 0, s, b, n_tiles);
	hipLaunchKernelGGL(plan_emit, dim3(n_tiles), dim3(PLANNER_THREADS), 0, s, b);
	hipLaunchKernelGGL(plan_finish, dim3(chunk_grid), dim3(256), 0, s, b, cfg);
	hipLaunchKernelGGL(plan_bins, dim3(1), dim3(64), 0, s, b);
	hipLaunchKernelGGL(plan_scatter, dim3(chunk_grid), dim3(256), 0, s, b);
	if (b.gang_slots) hipLaunchKernelGGL(plan_gangs, dim3(1), dim3(256), 0, s, b, cfg);
}

void launch_build_lut(int *d_lut, const DevParams &P, hipStream_t s)
{
	hipLaunchKernelGGL(k_build_lut, dim3((LUT_ENTRIES + 255) / 256), dim3(256), 0, s, d_lut, P);
}

size_t score_lds_bytes(const DevParams &P, int host_mode, int ring_slots)
{
	const size_t front = (size_t)ring_slots * WAVE * 4 + (size_t)SCORE_THREADS * sizeof(int4) + N_TEAM_RECORDS * sizeof(CoopShared) + TAB_INTS * 4;
	if (host_mode != MODE_LUT) return front;
	return front <= (size_t)LUT_LDS_BASE ? (size_t)LUT_LDS_TOTAL : (size_t)1 << 30;     // the table's place is fixed: what does not fit before it does not fit
}

// The SPLIT instantiation of k_score (one chunk on several workgroups: exact, and measured slower at every batch size)
// is only compiled into builds that ask for it (make SPLIT=1): it is the heaviest instantiation of the kernel (hundreds of spilled
// registers) and nothing selects it by default.
#ifdef MM2GB_WITH_SPLIT
constexpr bool HAVE_SPLIT = true;
#else
constexpr bool HAVE_SPLIT = false;
#endif
bool score_has_split_build() { return HAVE_SPLIT; }

int score_set_lds_limit(size_t bytes)
{
	hipError_t e = hipFuncSetAttribute((const void*)k_score<MODE_LUT, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
	if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_score<MODE_LUT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
	if constexpr (HAVE_SPLIT) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_score<MODE_LUT, HAVE_SPLIT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
	if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_score<MODE_FAST, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
	if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_score<MODE_GENERAL, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
	return e == hipSuccess ? 0 : -1;
}

void launch_score(const DevBatch &b, const DevParams &P, const LaunchCfg &cfg, hipStream_t s)
{
	if (b.n <= 0) return;
	const size_t lds = score_lds_bytes(P, cfg.host_mode, cfg.ring_slots);
	const size_t lds_general = score_lds_bytes(P, MODE_GENERAL, cfg.ring_slots);
	if (cfg.host_mode == MODE_LUT) {
		if (HAVE_SPLIT && cfg.split && b.split_slots) hipLaunchKernelGGL((k_score<MODE_LUT, HAVE_SPLIT, false>), dim3(cfg.score_grid), dim3(SCORE_THREADS), lds, s, b, P, cfg.host_mode, cfg.ring_slots, cfg.big_team, cfg.whole_wg_pct);
		else if (b.gang_slots) hipLaunchKernelGGL((k_score<MODE_LUT, false, true>), dim3(cfg.score_grid), dim3(SCORE_THREADS), lds, s, b, P, cfg.host_mode, cfg.ring_slots, cfg.big_team, cfg.whole_wg_pct);
		else hipLaunchKernelGGL((k_score<MODE_LUT, false, false>), dim3(cfg.score_grid), dim3(SCORE_THREADS), lds, s, b, P, cfg.host_mode, cfg.ring_slots, cfg.big_team, cfg.whole_wg_pct);
	}
	if (cfg.host_mode == MODE_FAST || cfg.host_mode == MODE_LUT) hipLaunchKernelGGL((k_score<MODE_FAST, false, false>), dim3(cfg.score_grid), dim3(SCORE_THREADS), lds, s, b, P, cfg.host_mode, cfg.ring_slots, cfg.big_team, cfg.whole_wg_pct);
	hipLaunchKernelGGL((k_score<MODE_GENERAL, false, false>), dim3(cfg.score_grid), dim3(SCORE_THREADS), lds_general, s, b, P, cfg.host_mode, cfg.ring_slots, cfg.big_team, cfg.whole_wg_pct);
}

} // namespace mm2gb
