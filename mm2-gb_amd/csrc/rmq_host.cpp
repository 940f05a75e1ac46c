// rmq_host.cpp -- the score fill of mg_lchain_rmq (lchain.c:250-350) on the host, one thread per read.
//
// The reference keeps the anchors within max_dist of the current one in a balanced tree ordered by (y, index) and asks it for the
// element of smallest priority in a y-range (krmq.h): O(log n) per anchor.  The device kernel k_rmq_fill answers the same question by
// scanning the window -- fine for windows of a few hundred anchors, hopeless for the re-chaining call of map.c:697-708, whose window
// is bw_long = 20 000 bases wide (thousands of anchors, scanned once per anchor, one anchor after the other).  Here the range
// question goes to a segment tree over the anchors' ranks in (y, index) order: a leaf is live while its anchor is inside the window;
// a node keeps the best key below it, where it sits (larger rank wins ties) and how many leaves share it.  Same answers as the
// reference whenever the best key in range is unique; when several elements tie the reference's choice depends on the shape of its
// tree, and the read is counted in n_tied, exactly like the device form.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "engine.h"
#include "host_chain.h"

namespace mm2gb {
namespace {

inline float log2_fit(float v)                 // mg_log2, mmpriv.h:118-126
{
	union { float f; uint32_t i; } z = { v };
	float r = (float)(((z.i >> 23) & 255u) - 128u);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	r += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return r;
}

// comput_sc_simple, lchain.c:232-248
inline int pair_score(const mm2gb_anchor_t &ai, const mm2gb_anchor_t &aj, float pen_gap, float pen_skip, bool *exact, int *width)
{
	const int dq = (int32_t)ai.y - (int32_t)aj.y, dr = (int32_t)(ai.x - aj.x);
	const int dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq, span = (int)(aj.y >> 32 & 0xff);
	int sc = span < dg ? span : dg;
	*width = dd;
	if (exact) *exact = dd == 0 && dg <= span;
	if (dd || dq > span) {
		const float lin = pen_gap * (float)dd + pen_skip * (float)dg;
		const float lg = dd >= 1 ? log2_fit((float)(dd + 1)) : 0.0f;
		sc -= (int)(lin + .5f * lg);
	}
	return sc;
}

struct Node { double key; int rank, cnt; };    // cnt == 0: nothing live below

inline Node better(const Node &a, const Node &b)
{
	if (a.cnt == 0) return b;
	if (b.cnt == 0) return a;
	if (a.key > b.key) return a;
	if (b.key > a.key) return b;
	Node r = a.rank > b.rank ? a : b;
	r.cnt = a.cnt + b.cnt;
	return r;
}

struct FillScratch { std::vector<Node> tree; std::vector<int> rank_of, by_rank; std::vector<int> ys; std::vector<uint64_t> inner; };   // inner: y << 32 | index of the inner window's anchors, ascending

// f[n], p_rel[n] (i - predecessor, 0 = none); returns the number of anchors whose range-minimum was shared by several elements
int64_t rmq_fill_one(const mm2gb_rmq_param_t &P, int64_t n64, const mm2gb_anchor_t *a, int32_t *f, int32_t *p_rel, FillScratch &ws)
{
	const int n = (int)n64;
	const int max_dist = P.max_dist < P.bw ? P.bw : P.max_dist;                                                   // lchain.c:264
	const int max_inner = (P.max_dist_inner <= 0 || P.max_dist_inner >= max_dist) ? 0 : P.max_dist_inner;         // lchain.c:265
	const double half_gap = 0.5 * (double)P.chn_pen_gap;
	// ranks in (y, index) order
	ws.by_rank.resize((size_t)n); ws.rank_of.resize((size_t)n); ws.ys.resize((size_t)n);
	for (int j = 0; j < n; ++j) ws.by_rank[(size_t)j] = j;
	std::sort(ws.by_rank.begin(), ws.by_rank.end(), [&](int u, int v) { const int yu = (int32_t)a[u].y, yv = (int32_t)a[v].y; return yu != yv ? yu < yv : u < v; });
	for (int k = 0; k < n; ++k) { ws.rank_of[(size_t)ws.by_rank[(size_t)k]] = k; ws.ys[(size_t)k] = (int32_t)a[ws.by_rank[(size_t)k]].y; }
	int leaves = 1;
	while (leaves < n) leaves <<= 1;
	ws.tree.assign((size_t)2 * leaves, Node{ 0.0, -1, 0 });
	auto set_leaf = [&](int rank, const Node &v) {
		size_t at = (size_t)(leaves + rank);
		ws.tree[at] = v;
		for (at >>= 1; at >= 1; at >>= 1) ws.tree[at] = better(ws.tree[2 * at], ws.tree[2 * at + 1]);
	};
	auto best_in = [&](int lo, int hi) {                     // ranks [lo, hi)
		Node res{ 0.0, -1, 0 };
		for (size_t l = (size_t)(leaves + lo), r = (size_t)(leaves + hi); l < r; l >>= 1, r >>= 1) {
			if (l & 1) res = better(res, ws.tree[l++]);
			if (r & 1) res = better(res, ws.tree[--r]);
		}
		return res;
	};
	int64_t tied = 0;
	int i0 = 0, st = 0, st_in = 0;
	ws.inner.clear();
	auto inner_key = [&](int j) { return (uint64_t)(uint32_t)(int32_t)a[j].y << 32 | (uint32_t)j; };   // query positions are non-negative
	for (int i = 0; i < n; ++i) {
		const int yi = (int32_t)a[i].y, q_i = (int)(a[i].y >> 32 & 0xff);
		if (i0 < i && a[i0].x != a[i].x) {                     // lchain.c:279-292: the anchors before the run of equal x that holds i go in
			for (int j = i0; j < i; ++j) {
				set_leaf(ws.rank_of[(size_t)j], Node{ (double)f[j] + half_gap * (double)((int32_t)a[j].x + (int32_t)a[j].y), ws.rank_of[(size_t)j], 1 });
				if (max_inner > 0 && j >= st_in) { const uint64_t k = inner_key(j); ws.inner.insert(std::lower_bound(ws.inner.begin(), ws.inner.end(), k), k); }
			}
			i0 = i;
		}
		// lchain.c:293-310: out of reach (other strand | reference, too far back) or too many in the tree
		while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + (uint64_t)max_dist || (i0 > st ? i0 - st : 0) > P.cap_rmq_size)) {
			if (st < i0) set_leaf(ws.rank_of[(size_t)st], Node{ 0.0, -1, 0 });
			++st;
		}
		if (max_inner > 0)
			while (st_in < i && (a[i].x >> 32 != a[st_in].x >> 32 || a[i].x > a[st_in].x + (uint64_t)max_inner || (i0 > st_in ? i0 - st_in : 0) > P.cap_rmq_size)) {
				if (st_in < i0) { const uint64_t k = inner_key(st_in); ws.inner.erase(std::lower_bound(ws.inner.begin(), ws.inner.end(), k)); }
				++st_in;
			}
		int max_f = q_i, max_j = -1;
		// lchain.c:311-315: the closed interval [(yi - max_dist, INT32_MAX), (yi, 0)] of (y, index): yi - max_dist < y < yi, or anchor 0 itself at y == yi
		const int lo = (int)(std::upper_bound(ws.ys.begin(), ws.ys.end(), yi - max_dist) - ws.ys.begin());
		const int hi = (int)(std::lower_bound(ws.ys.begin(), ws.ys.end(), yi) - ws.ys.begin());
		Node best = lo < hi ? best_in(lo, hi) : Node{ 0.0, -1, 0 };
		if ((int32_t)a[0].y == yi && st == 0 && i0 > 0) best = better(best, ws.tree[(size_t)(leaves + ws.rank_of[0])]);
		if (best.cnt > 0) {
			if (best.cnt > 1) ++tied;
			const int j = ws.by_rank[(size_t)best.rank];
			bool exact; int width;
			const int sc = f[j] + pair_score(a[i], a[j], P.chn_pen_gap, P.chn_pen_skip, &exact, &width);
			if (width <= P.bw && sc > max_f) { max_f = sc; max_j = j; }
			if (!exact && max_inner > 0 && st_in < i0 && yi > 0) {
				// lchain.c:320-341 at max_chn_skip = infinity: the best of the inner window's anchors with y in [yi - max_inner, yi - 1];
				// the walk goes from the largest (y, index) down and only a strictly better score replaces the best
				int bs = INT32_MIN, cj = -1;
				const auto from = std::lower_bound(ws.inner.begin(), ws.inner.end(), (uint64_t)(uint32_t)std::max(yi - max_inner, 0) << 32);
				auto it = std::lower_bound(from, ws.inner.end(), (uint64_t)(uint32_t)yi << 32);
				while (it != from) {                              // (y, index) descending
					--it;
					const int j2 = (int)(uint32_t)*it;
					int w2;
					const int s2 = f[j2] + pair_score(a[i], a[j2], P.chn_pen_gap, P.chn_pen_skip, nullptr, &w2);
					if (w2 <= P.bw && s2 > bs) { bs = s2; cj = j2; }
				}
				if (cj >= 0 && bs > max_f) { max_f = bs; max_j = cj; }
			}
		}
		f[i] = max_f;
		p_rel[i] = max_j < 0 ? 0 : i - max_j;
	}
	return tied;
}

} // namespace
} // namespace mm2gb

using namespace mm2gb;

extern "C" {

int mm2gb_rmq_chain_host(const mm2gb_rmq_param_t *prm, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, int n_threads,
                         mm2gb_chains_t *out, int32_t *n_tied)
{
	if (!prm || !out || n_reads < 0 || !offsets || offsets[0] != 0) return fail("mm2gb_rmq_chain_host: offsets[0] must be 0");
	memset(out, 0, sizeof(*out));
	if (prm->max_chn_skip != INT32_MAX) return fail("mm2gb_rmq_chain_host: max_chn_skip must be INT32_MAX (the exhaustive scan of the GPU path's contract)");
	for (int64_t r = 0; r < n_reads; ++r) {
		if (offsets[r + 1] < offsets[r]) return fail("mm2gb_rmq_chain_host: offsets must be non-decreasing");
		if (offsets[r + 1] - offsets[r] >= ((int64_t)1 << 30)) return fail("mm2gb_rmq_chain_host: a read is limited to 2^30 anchors");
	}
	if (offsets[n_reads] > 0 && !anchors) return fail("mm2gb_rmq_chain_host: null buffer");
	const size_t R = (size_t)n_reads;
	std::vector<uint64_t*> u_of(R, nullptr);
	std::vector<mm2gb_anchor_t*> a_of(R, nullptr);
	std::vector<int> nu_of(R, 0);
	std::vector<int64_t> na_of(R, 0);
	mm2gb_misc_t misc = {};
	misc.min_cnt = prm->min_cnt; misc.min_score = prm->min_sc; misc.bw = prm->bw; misc.is_cdna = 0; misc.n_seg = 1;   // max_drop = bw (lchain.c:253,355)
	std::atomic<int64_t> next(0);
	HostAlloc libc_mem;
	auto work = [&]() {
		FillScratch ws;
		BacktrackScratch bs;
		std::vector<int32_t> f, p;
		for (;;) {
			const int64_t r = next.fetch_add(1);
			if (r >= n_reads) break;
			const int64_t n = offsets[r + 1] - offsets[r];
			if (n == 0) continue;
			f.resize((size_t)n); p.resize((size_t)n);
			const int64_t t = rmq_fill_one(*prm, n, anchors + offsets[r], f.data(), p.data(), ws);
			if (n_tied) n_tied[r] = (int32_t)std::min<int64_t>(t, INT32_MAX);
			nu_of[(size_t)r] = backtrack_compact(misc, n, anchors + offsets[r], f.data(), p.data(), libc_mem, bs, &u_of[(size_t)r], &a_of[(size_t)r]);
			for (int c = 0; c < nu_of[(size_t)r]; ++c) na_of[(size_t)r] += (uint32_t)u_of[(size_t)r][c];
		}
	};
	if (n_tied) for (int64_t r = 0; r < n_reads; ++r) n_tied[r] = 0;
	const int nt = std::max(1, n_threads);
	if (nt == 1) work();
	else { std::vector<std::thread> pool; for (int t = 0; t < nt; ++t) pool.emplace_back(work); for (auto &th : pool) th.join(); }
	out->u_off = (int64_t*)malloc((R + 1) * 8);
	out->a_off = (int64_t*)malloc((R + 1) * 8);
	if (!out->u_off || !out->a_off) { mm2gb_chains_free(out); return fail("mm2gb_rmq_chain_host: out of memory"); }
	out->u_off[0] = out->a_off[0] = 0;
	for (size_t r = 0; r < R; ++r) { out->u_off[r + 1] = out->u_off[r] + nu_of[r]; out->a_off[r + 1] = out->a_off[r] + na_of[r]; }
	out->u = (uint64_t*)malloc(((size_t)out->u_off[R] + 1) * 8);
	out->a = (mm2gb_anchor_t*)malloc(((size_t)out->a_off[R] + 1) * 16);
	if (!out->u || !out->a) { mm2gb_chains_free(out); return fail("mm2gb_rmq_chain_host: out of memory"); }
	for (size_t r = 0; r < R; ++r) {
		if (nu_of[r] > 0) { memcpy(out->u + out->u_off[r], u_of[r], (size_t)nu_of[r] * 8); memcpy(out->a + out->a_off[r], a_of[r], (size_t)na_of[r] * 16); }
		free(u_of[r]); free(a_of[r]);
	}
	return 0;
}

} // extern "C"
