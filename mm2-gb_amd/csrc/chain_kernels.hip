// chain_kernels.hip -- gfx950 kernels of the chaining hot path.
//
// What the path computes is fixed by the reference's CPU code (lchain.c:113-207); how it is computed here is not
// taken from the reference's gpu/*.cu:
//   * k_split_soa   mm128_t AoS -> SoA on the device (the reference does this on one CPU thread, plmem.cu:154-198)
//   * k_window      predecessor-window start per anchor by galloping + binary search (role of plrange.cu:38-76),
//                   fused with the planner's per-block reductions (cuts, pair counts, max_iter clamps)
//   * k_plan        turns cuts into independent, cost-ordered work items ("chunks") without host round trips
//                   (role of the cut/long_seg/mid_seg bookkeeping, plscore.cu:314-385, and the host pairsort, plchain.cu:30-42)
//   * k_score_wave  the DP: one wave64 per chunk, 64 anchors per tile held one-per-lane in registers; predecessors are
//                   broadcast lane->SGPR (v_readlane) so every lane scores the same predecessor against its own anchor.
//                   No block barriers, no global read-modify-write (role of plscore.cu:109-187, 290-451).
// Arithmetic follows lchain.c:113-138 + mmpriv.h:118-126 bit for bit: compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "chain_dev.h"

namespace mm2gb {

#define WAVE 64

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int bcast(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
__device__ __forceinline__ int first_lane(int v) { return __builtin_amdgcn_readfirstlane(v); }

// --------------------------------------------------------------------------------------------------------------
// AoS -> SoA
// --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_split_soa(DevBatch b)
{
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	bool any_seg = false;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < b.n; i += stride) {
		const uint4 v = b.raw[i];                 // x.lo x.hi y.lo y.hi
		b.x[i] = (int32_t)v.x;
		b.xhi[i] = (int32_t)v.y;
		b.y[i] = (int32_t)v.z;
		const unsigned span = v.w & 0xffu;        // y>>32 & 0xff      (lchain.c:125)
		const unsigned seg = (v.w >> 16) & 0xffu; // (y & MM_SEED_SEG_MASK) >> 48 (lchain.c:116)
		b.tag[i] = (uint16_t)(seg << 8 | span);
		any_seg |= seg != 0;
	}
	if (__ballot(any_seg) != 0 && lane_id() == 0) atomicOr(b.flags, FLAG_ANY_SEGID);
}

// --------------------------------------------------------------------------------------------------------------
// Predecessor window start (lchain.c:172-173) + planner reductions
//   st[i] = max( first j <= i in the same read with xhi[j]==xhi[i] and x[i] <= x[j]+max_dist_x ,  i - max_iter )
// The CPU carries st across iterations; because validity is monotone in both i and j (anchors sorted by x) the
// carried value equals this closed form ("window start").
// --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool in_reach(const DevBatch &b, int j, int hi_i, unsigned x_i, unsigned dist)
{
	return b.xhi[j] == hi_i && x_i <= (unsigned)b.x[j] + dist;   // positions < 2^31, dist < 2^31: no wrap
}

__global__ __launch_bounds__(PLAN_THREADS) void k_window(DevBatch b, DevParams P)
{
	__shared__ int s_cut[PLAN_THREADS / WAVE];
	__shared__ unsigned long long s_pairs[PLAN_THREADS / WAVE];
	__shared__ int s_clamp[PLAN_THREADS / WAVE];
	const int64_t base = (int64_t)blockIdx.x * PLAN_BLOCK;
	const unsigned dist = (unsigned)P.max_dist_x;
	int my_cut = INT_MAX, my_clamp = 0;
	unsigned long long my_pairs = 0;

	for (int it = 0; it < PLAN_BLOCK / PLAN_THREADS; ++it) {
		const int64_t i64 = base + it * PLAN_THREADS + threadIdx.x;
		if (i64 >= b.n) break;
		const int i = (int)i64;
		// read that owns anchor i: last r with offsets[r] <= i
		int64_t lo = 0, hi = b.n_reads;           // invariant: offsets[lo] <= i < offsets[hi]
		while (hi - lo > 1) {
			const int64_t mid = (lo + hi) >> 1;
			if (b.offsets[mid] <= i64) lo = mid; else hi = mid;
		}
		const int rs = (int)b.offsets[lo];
		int lb = i - P.max_iter;                  // may be negative
		if (lb < rs) lb = rs;
		const int hi_i = b.xhi[i];
		const unsigned x_i = (unsigned)b.x[i];
		int st = i;
		if (i > lb && in_reach(b, i - 1, hi_i, x_i, dist)) {
			// gallop back from i until out of reach or at lb, then bisect
			int good = i - 1, step = 2, bad = -1;
			while (true) {
				int probe = i - step;
				if (probe <= lb) { probe = lb; if (in_reach(b, probe, hi_i, x_i, dist)) good = lb; else bad = lb; break; }
				if (in_reach(b, probe, hi_i, x_i, dist)) { good = probe; step <<= 1; }
				else { bad = probe; break; }
			}
			if (bad >= 0) {
				int l = bad, h = good;            // l out of reach, h in reach
				while (h - l > 1) {
					const int mid = (l + h) >> 1;
					if (in_reach(b, mid, hi_i, x_i, dist)) h = mid; else l = mid;
				}
				good = h;
			}
			st = good;
			// the max_iter clamp bit (lchain.c:173): window would have reached further back
			if (st == lb && lb > rs && lb == i - P.max_iter && in_reach(b, lb - 1, hi_i, x_i, dist)) my_clamp = 1;
		}
		b.st[i] = st;
		my_pairs += (unsigned)(i - st);
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
	if (threadIdx.x == 0) {
		for (int k = 1; k < PLAN_THREADS / WAVE; ++k) { my_cut = min(my_cut, s_cut[k]); my_pairs += s_pairs[k]; my_clamp |= s_clamp[k]; }
		b.blk_firstcut[blockIdx.x] = my_cut;
		b.blk_pairs[blockIdx.x] = (int64_t)my_pairs;
		b.blk_clamped[blockIdx.x] = my_clamp;
	}
}

// --------------------------------------------------------------------------------------------------------------
// Planner: one workgroup.  A chunk starts at the first cut of every planning block that has one and runs to the
// next such cut, so chunks are independent DP problems of >= ~PLAN_BLOCK anchors (or one long segment).
// Chunks are then bucket-sorted by estimated cost, most expensive first.
// --------------------------------------------------------------------------------------------------------------
constexpr int PLANNER_THREADS = 1024;
constexpr int COST_BINS = 256;

__device__ __forceinline__ int cost_bin(int64_t c)
{
	if (c <= 0) return 0;
	const int msb = 63 - __clzll(c);
	const int frac = msb >= 2 ? (int)((c >> (msb - 2)) & 3) : 0;
	return min(COST_BINS - 1, msb * 4 + frac);
}

template <typename T>
__device__ T block_exclusive_scan(T v, T *s_tmp /* PLANNER_THREADS/64 */, T *total)
{
	// inclusive scan inside the wave
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

__global__ __launch_bounds__(PLANNER_THREADS) void k_plan(DevBatch b, LaunchCfg cfg)
{
	__shared__ long long s_tmp[PLANNER_THREADS / WAVE];
	__shared__ int s_hist[COST_BINS];
	__shared__ int s_binbase[COST_BINS];
	const int64_t nb = b.n_blocks;
	const int tid = threadIdx.x;
	const int64_t per = (nb + PLANNER_THREADS - 1) / PLANNER_THREADS;
	const int64_t b0 = min(nb, (int64_t)tid * per), b1 = min(nb, b0 + per);

	// pass 1: per-thread counts over its slice of planning blocks
	long long cnt = 0, pairs = 0, clamps = 0;
	for (int64_t k = b0; k < b1; ++k) {
		cnt += b.blk_firstcut[k] != INT_MAX;
		pairs += b.blk_pairs[k];
		clamps += b.blk_clamped[k];
	}
	long long n_chunks, tot_pairs, tot_clamps;
	long long c_base = block_exclusive_scan<long long>(cnt, s_tmp, &n_chunks);
	long long p_base = block_exclusive_scan<long long>(pairs, s_tmp, &tot_pairs);
	long long k_base = block_exclusive_scan<long long>(clamps, s_tmp, &tot_clamps);

	// pass 2: chunk starts; stash the pair / clamp prefix at each chunk's first block in chunk_cost / chunk_end
	{
		long long c = c_base, pp = p_base, kk = k_base;
		for (int64_t k = b0; k < b1; ++k) {
			const int fc = b.blk_firstcut[k];
			if (fc != INT_MAX) {
				b.chunk_start[c] = fc;
				b.chunk_cost[c] = pp;           // pairs in blocks before this one
				b.chunk_end[c] = (int)kk;       // clamped blocks before this one (temporarily)
				b.order[c] = (int)k;            // block id (temporarily)
				++c;
			}
			pp += b.blk_pairs[k];
			kk += b.blk_clamped[k];
		}
	}
	for (int k = tid; k < COST_BINS; k += PLANNER_THREADS) s_hist[k] = 0;
	__threadfence_block();
	__syncthreads();

	// pass 3: ends, costs, track flags, histogram.  Two sweeps because pass 3 overwrites what neighbours read.
	const int64_t cper = (n_chunks + PLANNER_THREADS - 1) / PLANNER_THREADS;
	const int64_t c0 = min((int64_t)n_chunks, (int64_t)tid * cper), c1 = min((int64_t)n_chunks, c0 + cper);
	// values of the chunk after my last one, read before anyone overwrites them
	long long nxt_pp = tot_pairs, nxt_kk = tot_clamps;
	int nxt_start = (int)b.n, nxt_blk_clamped = 0;
	if (c1 < n_chunks) {
		nxt_pp = b.chunk_cost[c1]; nxt_kk = b.chunk_end[c1]; nxt_start = b.chunk_start[c1];
		nxt_blk_clamped = b.blk_clamped[b.order[c1]];
	}
	__syncthreads();
	int n_track = 0;
	for (int64_t c = c1 - 1; c >= c0; --c) {   // backwards so "next" values are still the stashed ones
		const long long pp = b.chunk_cost[c], kk = b.chunk_end[c];
		const int start = b.chunk_start[c], blk = b.order[c];
		const int end = nxt_start;
		// the chunk covers its own block .. part of the next chunk's block: count clamps inclusively (superset is safe)
		const bool track = (nxt_kk + nxt_blk_clamped - kk) > 0;
		const long long cost = (nxt_pp - pp) + (long long)(end - start) * COST_PER_ANCHOR;
		b.chunk_end[c] = end;
		b.chunk_cost[c] = cost;
		b.chunk_track[c] = track;
		n_track += track;
		atomicAdd(&s_hist[cost_bin(cost)], 1);
		nxt_pp = pp; nxt_kk = kk; nxt_start = start; nxt_blk_clamped = b.blk_clamped[blk];
	}
	__syncthreads();
	// descending bin order -> base offsets
	if (tid == 0) {
		int acc = 0;
		for (int k = COST_BINS - 1; k >= 0; --k) { s_binbase[k] = acc; acc += s_hist[k]; }
	}
	__syncthreads();
	for (int64_t c = c0; c < c1; ++c) {
		const int slot = atomicAdd(&s_binbase[cost_bin(b.chunk_cost[c])], 1);
		b.order[slot] = (int)c;
	}
	if (n_track) atomicAdd(&b.counters[CNT_NTRACK], n_track);
	if (tid == 0) {
		b.counters[CNT_NCHUNK] = (int)n_chunks;
		b.counters[CNT_CURSOR] = 0;
		b.counters[CNT_NLONG] = 0;
		b.counters[CNT_LCURSOR] = 0;
		b.counters[CNT_NCLAMP] = (int)tot_clamps;
		b.totals[0] = tot_pairs;
	}
	(void)cfg;
}

// --------------------------------------------------------------------------------------------------------------
// Pair score, lchain.c:113-138.  FAST = single query segment, not cDNA (what --gpu-chain runs: plchain.cu:499-500).
// --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float log2_fit(float v)   // mmpriv.h:118-126
{
	unsigned u = __float_as_uint(v);
	float r = (float)(((u >> 23) & 255u) - 128u);
	u = (u & ~(255u << 23)) + (127u << 23);
	const float m = __uint_as_float(u);
	r += (-0.34484843f * m + 2.02466578f) * m - 0.67487759f;
	return r;
}

template <bool FAST>
__device__ __forceinline__ bool pair_score(int xi, int yi, int segi, int xj, int yj, int spanj, int segj, const DevParams &P, int &sc_out)
{
	const int dq = yi - yj;
	const int dr = xi - xj;                                    // low 32 bits of the 64-bit difference (lchain.c:119)
	const int ddiff = (int)((unsigned)dr - (unsigned)dq);
	const int dd = ddiff < 0 ? -ddiff : ddiff;
	const int dg = dr < dq ? dr : dq;
	int sc = spanj < dg ? spanj : dg;
	const float lin = P.gap * (float)dd + P.skip * (float)dg;
	const float lg = dd >= 1 ? log2_fit((float)(dd + 1)) : 0.0f;
	if (FAST) {
		const bool ok = (unsigned)(dq - 1) < (unsigned)P.dq_lim && dr != 0 && dd <= P.bw;
		if (dd != 0 || dg > spanj) sc -= (int)(lin + .5f * lg);
		sc_out = sc;
		return ok;
	} else {
		const bool same = segi == segj;
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
// One wave works through a chunk [cs, ce) in tiles of 64 anchors, lane L owning anchor i0+L.
//   best / arg : running maximum in the "threshold" form: best starts at q_span+1 with arg=-1, so "cand >= best" is
//                the CPU's strict '>' against q_span first and "latest j wins ties" afterwards (predecessors are
//                visited in ascending j here, descending with strict '>' on the CPU: lchain.c:174-181).
//   TRACK      : carry the max_ii state machine of lchain.c:189-205 (only chunks with max_iter-clamped windows need it;
//                elsewhere max_ii always lies inside the window and the extra candidate is a no-op).
// --------------------------------------------------------------------------------------------------------------
struct Keep { int idx, x, hi, y, tag, f; };   // the remembered best anchor ("max_ii") and its fields, wave-uniform

template <bool FAST, bool TRACK>
__device__ void run_chunk(const DevBatch &b, const DevParams &P, const int cs, const int ce)
{
	const int lane = lane_id();
	Keep keep; keep.idx = -1; keep.x = keep.hi = keep.y = keep.tag = keep.f = 0;

	for (int i0 = cs; i0 < ce; i0 += WAVE) {
		const int i = i0 + lane;
		const bool live = i < ce;
		const int il = live ? i : ce - 1;
		const int xi = b.x[il], yi = b.y[il], tgi = b.tag[il];
		const int hii = TRACK ? b.xhi[il] : 0;
		const int sti = live ? b.st[il] : INT_MAX;        // dead lanes never activate
		const int qi = tgi & 0xff, segi = tgi >> 8;
		int best = qi + 1, arg = -1;

		// ---- predecessors in earlier tiles: all final, broadcast one by one ----
		const int tile_lo = first_lane(sti);               // lane 0 has the smallest window start
		for (int jb = cs + ((tile_lo - cs) & ~(WAVE - 1)); jb < i0; jb += WAVE) {
			const int js = jb + lane;                      // < i0 <= ce
			const int sx = b.x[js], sy = b.y[js], stg = b.tag[js], sf = b.f[js];
			int kg = tile_lo - jb; kg = kg < 0 ? 0 : (kg & ~3);
			for (; kg < WAVE; kg += 4) {
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const int k = kg + u, j = jb + k;
					const int ux = bcast(sx, k), uy = bcast(sy, k), ut = bcast(stg, k), uf = bcast(sf, k);
					int sc;
					const bool ok = pair_score<FAST>(xi, yi, segi, ux, uy, ut & 0xff, ut >> 8, P, sc);
					const int cand = sc + uf;
					if (ok && j >= sti && cand >= best) { best = cand; arg = j; }
				}
			}
		}

		// ---- predecessors inside the tile: lane t becomes final at step t and is pushed to the lanes above it ----
		const int n_here = min(WAVE, ce - i0);
		if (!TRACK) {
			// source t matters only if anchor t+1 reaches back to it (window starts are monotone)
			unsigned long long need = __ballot(live && sti < i) >> 1;
			while (need) {
				const int t = __builtin_ctzll(need);
				need &= need - 1;
				const int j = i0 + t;
				const int ft = bcast(arg < 0 ? qi : best, t);
				const int ux = bcast(xi, t), uy = bcast(yi, t), ut = bcast(tgi, t);
				int sc;
				const bool ok = pair_score<FAST>(xi, yi, segi, ux, uy, ut & 0xff, ut >> 8, P, sc);
				const int cand = sc + ft;
				if (ok && lane > t && j >= sti && cand >= best) { best = cand; arg = j; }
			}
		} else {
			for (int t = 0; t < n_here; ++t) {
				const int j = i0 + t;
				const int xt = bcast(xi, t), yt = bcast(yi, t), tgt = bcast(tgi, t), ht = bcast(hii, t), stt = bcast(sti, t);
				// lchain.c:190-195: the remembered anchor fell out of reach (or none yet): arg-max of f over the window,
				// largest index among equals
				if (keep.idx < 0 || ht != keep.hi || (unsigned)(xt - keep.x) > (unsigned)P.max_dist_x) {
					int bf = INT_MIN, bi = -1;
					for (int jj = stt + lane; jj < i0; jj += WAVE) {      // earlier tiles (ascending per lane)
						const int v = b.f[jj];
						if (v >= bf) { bf = v; bi = jj; }
					}
					if (lane < t && i >= stt) {                              // finished lanes of this tile
						const int v = arg < 0 ? qi : best;
						if (v >= bf) { bf = v; bi = i; }
					}
					for (int off = WAVE / 2; off > 0; off >>= 1) {
						const int of = __shfl_xor(bf, off), oi = __shfl_xor(bi, off);
						if (of > bf || (of == bf && oi > bi)) { bf = of; bi = oi; }
					}
					keep.idx = first_lane(bi);
					if (keep.idx >= 0) {
						keep.f = first_lane(bf);
						keep.x = b.x[keep.idx]; keep.y = b.y[keep.idx]; keep.tag = b.tag[keep.idx]; keep.hi = ht;
					}
				}
				// lchain.c:196-201: one more candidate if the scan stopped before reaching it (end_j = st-1 at max_skip=inf)
				if (keep.idx >= 0 && keep.idx < stt - 1) {
					int sc;
					const bool ok = pair_score<FAST>(xt, yt, tgt >> 8, keep.x, keep.y, keep.tag & 0xff, keep.tag >> 8, P, sc);
					if (ok && lane == t) {
						const int cur = arg < 0 ? qi : best;
						if (cur < sc + keep.f) { best = sc + keep.f; arg = keep.idx; }
					}
				}
				const int ft = bcast(arg < 0 ? qi : best, t);               // lchain.c:202
				// lchain.c:204-205 (in reach is guaranteed after the refresh above)
				if (keep.idx < 0 || keep.f < ft) { keep.idx = j; keep.x = xt; keep.hi = ht; keep.y = yt; keep.tag = tgt; keep.f = ft; }
				// push
				int sc;
				const bool ok = pair_score<FAST>(xi, yi, segi, xt, yt, tgt & 0xff, tgt >> 8, P, sc);
				const int cand = sc + ft;
				if (ok && lane > t && j >= sti && cand >= best) { best = cand; arg = j; }
			}
		}
		if (live) {
			b.f[i] = arg < 0 ? qi : best;
			b.p[i] = arg < 0 ? 0 : i - arg;
		}
	}
}

// FAST kernel runs only when no anchor carries a segment id; the general one only when some does (or always,
// when the host already knows the parameters need it).  Exactly one of the two does the work of a batch.
template <bool FAST>
__global__ __launch_bounds__(256) void k_score_wave(DevBatch b, DevParams P, int general_always)
{
	const bool any_seg = (b.flags[0] & FLAG_ANY_SEGID) != 0;
	if (FAST ? any_seg : (!general_always && !any_seg)) return;
	const int n_chunks = b.counters[CNT_NCHUNK];
	while (true) {
		int c = 0;
		if (lane_id() == 0) c = atomicAdd(&b.counters[CNT_CURSOR], 1);
		c = first_lane(c);
		if (c >= n_chunks) break;
		const int ci = b.order[c];
		const int cs = b.chunk_start[ci], ce = b.chunk_end[ci];
		if (b.chunk_track[ci]) run_chunk<FAST, true>(b, P, cs, ce);
		else run_chunk<FAST, false>(b, P, cs, ce);
	}
}

// --------------------------------------------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------------------------------------------
void launch_split_soa(const DevBatch &b, hipStream_t s)
{
	if (b.n <= 0) return;
	int64_t blocks = (b.n + 255) / 256;
	if (blocks > 256 * 16) blocks = 256 * 16;
	hipLaunchKernelGGL(k_split_soa, dim3((unsigned)blocks), dim3(256), 0, s, b);
}

void launch_window(const DevBatch &b, const DevParams &P, hipStream_t s)
{
	if (b.n <= 0) return;
	hipLaunchKernelGGL(k_window, dim3((unsigned)b.n_blocks), dim3(PLAN_THREADS), 0, s, b, P);
}

void launch_plan(const DevBatch &b, const LaunchCfg &cfg, hipStream_t s)
{
	hipLaunchKernelGGL(k_plan, dim3(1), dim3(PLANNER_THREADS), 0, s, b, cfg);
}

void launch_score(const DevBatch &b, const DevParams &P, const LaunchCfg &cfg, hipStream_t s)
{
	if (b.n <= 0) return;
	const bool host_general = P.is_cdna || P.n_seg > 1;
	if (!host_general) hipLaunchKernelGGL(k_score_wave<true>, dim3(cfg.wave_grid), dim3(256), 0, s, b, P, 0);
	hipLaunchKernelGGL(k_score_wave<false>, dim3(cfg.wave_grid), dim3(256), 0, s, b, P, host_general ? 1 : 0);
}

} // namespace mm2gb
