// rmq_host.cpp -- the score fill of mg_lchain_rmq (lchain.c:250-350) on the host, one thread per read.
//
// The reference keeps the anchors within max_dist of the current one in a balanced tree ordered by (y, index) and asks it for the
// element of smallest priority in a y-range (krmq.h): O(log n) per anchor.  The device kernel k_rmq_fill answers the same question by
// scanning the window -- fine for windows of a few hundred anchors, hopeless for the re-chaining call of map.c:697-708, whose window
// is bw_long = 20 000 bases wide (thousands of anchors, scanned once per anchor, one anchor after the other).
// When several elements in range share the smallest priority, WHICH of them the reference returns depends on the shape of its tree
// and on how it keeps each subtree's minimum (krmq.h:146-150: a node, then its left subtree's, then its right subtree's, each
// replacing the other only if strictly smaller) and walks the two search paths (krmq.h:108-148).  So the tree here is that tree: an
// AVL tree with the same insertion, deletion and rotation rules and the same bookkeeping, on arrays indexed by anchor number
// (ShapeTree below) -- the chains are then the reference's for every read, ties included; nothing is left to report.
#include <algorithm>
#include <atomic>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <cstdio>
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

// The reference's tree (krmq.h) on arrays.  Node = anchor index; node n is scratch (the stand-in root of a deletion, krmq.h:262).
// kid[0] / kid[1]: left / right child (-1: none); low[v]: the node of smallest priority in v's subtree, ties as described above;
// tilt: AVL balance factor (right height - left height); count: nodes in the subtree.
struct ShapeTree {
	struct Node { int kid[2], low; unsigned count; double pri; int y; signed char tilt; };   // 32 bytes: a visit touches one line, not five arrays
	std::vector<Node> nd;
	int root = -1, scratch = 0;
	static constexpr int MAX_DEPTH = 64;

	void reset(const mm2gb_anchor_t *anchors, int n)
	{
		root = -1; scratch = n;
		nd.resize((size_t)n + 1);
		for (int v = 0; v < n; ++v) nd[(size_t)v].y = (int32_t)anchors[v].y;
	}
	// order of (y, index) pairs (lchain.c:225): the key of node v is (y of anchor v, v)
	int order(int y, int64_t i, int v) const
	{
		const int yv = nd[(size_t)v].y;
		return y < yv ? -1 : y > yv ? 1 : (i > v) - (i < v);
	}
	bool lower(int u, int v) const { return nd[(size_t)(u)].pri < nd[(size_t)(v)].pri; }              // lchain.c:226
	unsigned below(int v, int side) const { const int c = nd[(size_t)(v)].kid[side]; return c < 0 ? 0u : nd[(size_t)(c)].count; }
	// krmq.h:146-150
	void refresh_low(int v, int first, int second)
	{
		int m = (first < 0 || lower(v, nd[(size_t)(first)].low)) ? v : nd[(size_t)(first)].low;
		m = (second < 0 || lower(m, nd[(size_t)(second)].low)) ? m : nd[(size_t)(second)].low;
		nd[(size_t)(v)].low = m;
	}
	// krmq.h:151-163: (a,(b,c)q)p => ((a,b)p,c)q for side == 0, mirrored for side == 1
	int rotate_once(int p, int side)
	{
		const int other = 1 - side, q = nd[(size_t)(p)].kid[other], keep = nd[(size_t)(p)].low;
		const unsigned was = nd[(size_t)(p)].count;
		nd[(size_t)(p)].count -= nd[(size_t)(q)].count - below(q, side);
		nd[(size_t)(q)].count = was;
		refresh_low(p, nd[(size_t)(p)].kid[side], nd[(size_t)(q)].kid[side]);
		nd[(size_t)(q)].low = keep;
		nd[(size_t)(p)].kid[other] = nd[(size_t)(q)].kid[side];
		nd[(size_t)(q)].kid[side] = p;
		return q;
	}
	// krmq.h:164-187: (a,((b,c)r,d)q)p => ((a,b)p,(c,d)q)r
	int rotate_twice(int p, int side)
	{
		const int other = 1 - side, q = nd[(size_t)(p)].kid[other], r = nd[(size_t)(q)].kid[side], keep = nd[(size_t)(p)].low;
		const unsigned inner = below(r, side);
		nd[(size_t)(r)].count = nd[(size_t)(p)].count;
		nd[(size_t)(p)].count -= nd[(size_t)(q)].count - inner;
		nd[(size_t)(q)].count -= inner + 1;
		refresh_low(p, nd[(size_t)(p)].kid[side], nd[(size_t)(r)].kid[side]);
		refresh_low(q, nd[(size_t)(q)].kid[other], nd[(size_t)(r)].kid[other]);
		nd[(size_t)(r)].low = keep;
		nd[(size_t)(p)].kid[other] = nd[(size_t)(r)].kid[side];
		nd[(size_t)(r)].kid[side] = p;
		nd[(size_t)(q)].kid[side] = nd[(size_t)(r)].kid[other];
		nd[(size_t)(r)].kid[other] = q;
		const int lean = side == 0 ? +1 : -1;
		if (nd[(size_t)(r)].tilt == lean) { nd[(size_t)(q)].tilt = 0; nd[(size_t)(p)].tilt = (signed char)-lean; }
		else if (nd[(size_t)(r)].tilt == 0) nd[(size_t)(q)].tilt = nd[(size_t)(p)].tilt = 0;
		else { nd[(size_t)(q)].tilt = (signed char)lean; nd[(size_t)(p)].tilt = 0; }
		nd[(size_t)(r)].tilt = 0;
		return r;
	}
	// krmq.h:189-240
	void insert(int x, double priority)
	{
		unsigned char turn[MAX_DEPTH];
		int path[MAX_DEPTH];
		int pivot = root, above_pivot = -1, p = root, q = -1, top = 0, len = 0, side = 0;
		for (; p >= 0; q = p, p = nd[(size_t)(p)].kid[side]) {
			const int c = order(nd[(size_t)x].y, x, p);
			if (nd[(size_t)(p)].tilt != 0) { above_pivot = q; pivot = p; top = 0; }
			turn[top++] = (unsigned char)(side = c > 0);
			path[len++] = p;
		}
		nd[(size_t)(x)].pri = priority;
		nd[(size_t)(x)].tilt = 0; nd[(size_t)(x)].count = 1; nd[(size_t)(x)].kid[0] = nd[(size_t)(x)].kid[1] = -1; nd[(size_t)(x)].low = x;
		if (q < 0) root = x; else nd[(size_t)(q)].kid[side] = x;
		if (pivot < 0) return;
		for (int i = 0; i < len; ++i) ++nd[(size_t)(path[i])].count;
		for (int i = len - 1; i >= 0; --i) {
			refresh_low(path[i], nd[(size_t)(path[i])].kid[0], nd[(size_t)(path[i])].kid[1]);
			if (nd[(size_t)(path[i])].low != x) break;
		}
		top = 0;
		for (p = pivot; p != x; p = nd[(size_t)(p)].kid[turn[top]], ++top) nd[(size_t)(p)].tilt += turn[top] == 0 ? -1 : +1;
		if (nd[(size_t)(pivot)].tilt > -2 && nd[(size_t)(pivot)].tilt < 2) return;
		side = nd[(size_t)(pivot)].tilt < 0;
		const int lean = side == 0 ? +1 : -1;
		q = nd[(size_t)(pivot)].kid[1 - side];
		int r;
		if (nd[(size_t)(q)].tilt == lean) { r = rotate_once(pivot, side); nd[(size_t)(q)].tilt = nd[(size_t)(pivot)].tilt = 0; }
		else r = rotate_twice(pivot, side);
		if (above_pivot < 0) root = r; else nd[(size_t)(above_pivot)].kid[pivot != nd[(size_t)(above_pivot)].kid[0]] = r;
	}
	// krmq.h:242-325 for a node that is in the tree
	void erase(int x)
	{
		if (root < 0) return;
		int path[MAX_DEPTH];
		unsigned char turn[MAX_DEPTH];
		const int f = scratch;
		nd[(size_t)(f)].kid[0] = root; nd[(size_t)(f)].kid[1] = -1; nd[(size_t)(f)].low = nd[(size_t)(root)].low; nd[(size_t)(f)].tilt = nd[(size_t)(root)].tilt;
		nd[(size_t)(f)].count = nd[(size_t)(root)].count; nd[(size_t)(f)].pri = nd[(size_t)(root)].pri;
		int d = 0, p = f;
		for (int c = -1; c != 0; c = order(nd[(size_t)x].y, x, p)) {
			const int side = c > 0;
			turn[d] = (unsigned char)side; path[d++] = p;
			p = nd[(size_t)(p)].kid[side];
			if (p < 0) return;
		}
		for (int i = 1; i < d; ++i) --nd[(size_t)(path[i])].count;
		if (nd[(size_t)(p)].kid[1] < 0) nd[(size_t)(path[d - 1])].kid[turn[d - 1]] = nd[(size_t)(p)].kid[0];
		else {
			int q = nd[(size_t)(p)].kid[1];
			if (nd[(size_t)(q)].kid[0] < 0) {
				nd[(size_t)(q)].kid[0] = nd[(size_t)(p)].kid[0];
				nd[(size_t)(q)].tilt = nd[(size_t)(p)].tilt;
				nd[(size_t)(path[d - 1])].kid[turn[d - 1]] = q;
				path[d] = q; turn[d++] = 1;
				nd[(size_t)(q)].count = nd[(size_t)(p)].count - 1;
			} else {
				int r;
				const int e = d++;
				for (;;) {
					turn[d] = 0; path[d++] = q;
					r = nd[(size_t)(q)].kid[0];
					if (nd[(size_t)(r)].kid[0] < 0) break;
					q = r;
				}
				nd[(size_t)(r)].kid[0] = nd[(size_t)(p)].kid[0];
				nd[(size_t)(q)].kid[0] = nd[(size_t)(r)].kid[1];
				nd[(size_t)(r)].kid[1] = nd[(size_t)(p)].kid[1];
				nd[(size_t)(r)].tilt = nd[(size_t)(p)].tilt;
				nd[(size_t)(path[e - 1])].kid[turn[e - 1]] = r;
				path[e] = r; turn[e] = 1;
				for (int i = e + 1; i < d; ++i) --nd[(size_t)(path[i])].count;
				nd[(size_t)(r)].count = nd[(size_t)(p)].count - 1;
			}
		}
		for (int i = d - 1; i >= 0; --i) refresh_low(path[i], nd[(size_t)(path[i])].kid[0], nd[(size_t)(path[i])].kid[1]);
		while (--d > 0) {
			const int q = path[d], side = turn[d], other = 1 - side;
			const int b1 = side ? -1 : 1, b2 = side ? -2 : 2;
			nd[(size_t)(q)].tilt += (signed char)b1;
			if (nd[(size_t)(q)].tilt == b1) break;
			if (nd[(size_t)(q)].tilt == b2) {
				const int r = nd[(size_t)(q)].kid[other];
				if (nd[(size_t)(r)].tilt == -b1) nd[(size_t)(path[d - 1])].kid[turn[d - 1]] = rotate_twice(q, side);
				else {
					nd[(size_t)(path[d - 1])].kid[turn[d - 1]] = rotate_once(q, side);
					if (nd[(size_t)(r)].tilt == 0) { nd[(size_t)(r)].tilt = (signed char)-b1; nd[(size_t)(q)].tilt = (signed char)b1; break; }
					nd[(size_t)(r)].tilt = nd[(size_t)(q)].tilt = 0;
				}
			}
		}
		root = nd[(size_t)(f)].kid[0];
	}
	// krmq.h:108-148: the node of smallest priority among those with (y_lo, i_lo) <= key <= (y_hi, i_hi), -1 if none
	int lowest_between(int y_lo, int64_t i_lo, int y_hi, int64_t i_hi) const
	{
		if (root < 0) return -1;
		int path[2][MAX_DEPTH], went[2][MAX_DEPTH], len[2] = { 0, 0 };
		for (int side = 0; side < 2; ++side)
			for (int p = root; p >= 0;) {
				const int c = side == 0 ? order(y_lo, i_lo, p) : order(y_hi, i_hi, p);
				path[side][len[side]] = p; went[side][len[side]++] = c;
				if (c < 0) p = nd[(size_t)(p)].kid[0]; else if (c > 0) p = nd[(size_t)(p)].kid[1]; else break;
			}
		int fork = 0;
		for (; fork < len[0] && fork < len[1]; ++fork)
			if (path[0][fork] == path[1][fork] && went[0][fork] <= 0 && went[1][fork] >= 0) break;
		if (fork == len[0] || fork == len[1]) return -1;
		int m = path[0][fork];
		for (int i = fork + 1; i < len[0]; ++i)
			if (went[0][i] <= 0) {
				const int v = path[0][i], c = nd[(size_t)(v)].kid[1];
				if (lower(v, m)) m = v;
				if (c >= 0 && lower(nd[(size_t)(c)].low, m)) m = nd[(size_t)(c)].low;
			}
		for (int i = fork + 1; i < len[1]; ++i)
			if (went[1][i] >= 0) {
				const int v = path[1][i], c = nd[(size_t)(v)].kid[0];
				if (lower(v, m)) m = v;
				if (c >= 0 && lower(nd[(size_t)(c)].low, m)) m = nd[(size_t)(c)].low;
			}
		return m;
	}
	unsigned size() const { return root < 0 ? 0u : nd[(size_t)(root)].count; }
};

// The inner window (lchain.c:286-290, 301-310, 320-341): its anchors are only ever walked in (y, index) order over a y-range, which
// does not depend on a tree's shape.  They are kept in buckets of 64 query positions, each a small array in (y, index) order holding
// what scoring a pair needs -- position, span, score -- so that the walk reads memory front to back instead of chasing indices.
// A tournament tree of fixed shape over the read's anchors in (y, index) order: the same answer as the reference's tree wherever ONE
// anchor in range holds the smallest priority, at a fraction of the cost (no rebalancing, no pointers; an update stops at the first
// ancestor it does not change).  Where several anchors share the smallest priority the reference's answer depends on the shape of its
// tree: this one only reports the tie, and the read is done again with ShapeTree.
struct RankTree {
	// Round 6: eight children per node instead of two -- six levels for a read of 100 000 anchors where the binary tree had seventeen, and a
	// node's children sit side by side in memory (an update or a query touched ~70 nodes of a 5 MB tree before, a cache line each: 0.4 us an anchor).
	// Level 0 = the priorities by rank; a node of level k >= 1 sums up its eight children of level k - 1: smallest priority, the rank of ONE
	// leaf that holds it (the first), and whether several leaves below it do.
	static constexpr int FAN_LOG = 3, FAN = 1 << FAN_LOG;
	struct Sum { double pri; int at, tie; };
	std::vector<double> leaf;                       // level 0, padded to a multiple of FAN with EMPTY
	std::vector<std::vector<Sum>> up;               // up[k - 1] = level k
	std::vector<uint64_t> by_y;
	std::vector<int> rank, q_lo, q_hi;
	unsigned held = 0;
	static constexpr double EMPTY = 1e300;

	void reset(const mm2gb_anchor_t *a, int n, int max_dist)
	{
		by_y.resize((size_t)n); rank.resize((size_t)n); q_lo.resize((size_t)n); q_hi.resize((size_t)n);
		for (int j = 0; j < n; ++j) by_y[(size_t)j] = (uint64_t)(uint32_t)(int32_t)a[j].y << 32 | (uint32_t)j;
		sort_by_y();
		for (int r = 0; r < n; ++r) rank[(size_t)(uint32_t)by_y[(size_t)r]] = r;
		// the closed interval [(y - max_dist, INT32_MAX), (y, 0)] of (y, index) (lchain.c:311-313) in ranks: y' in (y - max_dist, y), and
		// anchor 0 -- the first of its y -- when its y is y
		const int64_t y0 = n > 0 ? (int32_t)a[0].y : 0;
		int lo = 0, hi = 0;
		for (int r = 0; r < n; ++r) {
			const int64_t y = (int32_t)(by_y[(size_t)r] >> 32);
			while (lo < n && (int64_t)(int32_t)(by_y[(size_t)lo] >> 32) <= y - max_dist) ++lo;
			while (hi < n && (int64_t)(int32_t)(by_y[(size_t)hi] >> 32) < y) ++hi;
			const int j = (int)(uint32_t)by_y[(size_t)r];
			q_lo[(size_t)j] = lo; q_hi[(size_t)j] = hi - 1 + (y0 == y ? 1 : 0);
		}
		size_t width = ((size_t)std::max(n, 1) + FAN - 1) & ~(size_t)(FAN - 1);
		leaf.assign(width, EMPTY);
		size_t levels = 0;
		for (size_t w = width; w > 1; w = ((w >> FAN_LOG) + FAN - 1) & ~(size_t)(FAN - 1)) { if (up.size() <= levels) up.emplace_back(); up[levels++].assign(((w >> FAN_LOG) + FAN - 1) & ~(size_t)(FAN - 1), Sum{ EMPTY, -1, 0 }); if ((w >> FAN_LOG) <= 1) break; }
		up.resize(levels);
		held = 0;
	}
	// the keys are (y << 32 | index) with the indices already rising: a stable sort on y alone, least significant byte first, skipping the
	// bytes every y shares (query positions are below 2^31 and mostly below 2^24; std::sort took 55 ns an anchor of the fill's 470)
	void sort_by_y()
	{
		const size_t n = by_y.size();
		if (n < 256) { std::sort(by_y.begin(), by_y.end()); return; }
		uint32_t all_or = 0, all_and = ~0u;
		for (size_t k = 0; k < n; ++k) { const uint32_t y = (uint32_t)(by_y[k] >> 32); all_or |= y; all_and &= y; }
		sort_tmp.resize(n);
		uint64_t *src = by_y.data(), *dst = sort_tmp.data();
		for (int byte = 0; byte < 4; ++byte) {
			const int shift = 32 + 8 * byte;
			if ((((all_or ^ all_and) >> (8 * byte)) & 0xffu) == 0) continue;        // the same in every key
			size_t count[257] = { 0 };
			for (size_t k = 0; k < n; ++k) ++count[((src[k] >> shift) & 0xff) + 1];
			for (int v = 0; v < 256; ++v) count[v + 1] += count[v];
			for (size_t k = 0; k < n; ++k) dst[count[(src[k] >> shift) & 0xff]++] = src[k];
			std::swap(src, dst);
		}
		if (src != by_y.data()) memcpy(by_y.data(), src, n * sizeof(uint64_t));
	}
	std::vector<uint64_t> sort_tmp;
	static Sum lower_of(const Sum &u, const Sum &v)
	{
		if (v.pri < u.pri) return v;
		if (v.pri == u.pri && v.pri != EMPTY) return Sum{ u.pri, u.at, 1 };
		return u;
	}
	Sum sum_of(size_t level, size_t at) const { return level == 0 ? Sum{ leaf[at], (int)at, 0 } : up[level - 1][at]; }
	void settle(size_t pos)
	{
		size_t at = pos;
		for (size_t level = 1; level <= up.size(); ++level) {
			at >>= FAN_LOG;
			const size_t first = at << FAN_LOG;
			Sum now{ EMPTY, -1, 0 };
			if (level == 1) { for (size_t c = first; c < first + FAN; ++c) now = lower_of(now, Sum{ leaf[c], (int)c, 0 }); }
			else { const Sum *kid = up[level - 2].data() + first; for (int c = 0; c < FAN; ++c) now = lower_of(now, kid[c]); }
			Sum &was = up[level - 1][at];
			if (now.pri == was.pri && now.at == was.at && now.tie == was.tie) break;
			was = now;
		}
	}
	void insert(int j, double priority) { const size_t pos = (size_t)rank[(size_t)j]; leaf[pos] = priority; ++held; settle(pos); }
	void erase(int j) { const size_t pos = (size_t)rank[(size_t)j]; leaf[pos] = EMPTY; --held; settle(pos); }
	unsigned size() const { return held; }
	// the nodes that tile the ranks [lo, hi]: at every level the few at either end that do not fill a parent, then one level up
	template <class F> void tiles(size_t lo, size_t hi, F &&visit) const
	{
		size_t l = lo, r = hi + 1;
		for (size_t level = 0; l < r; ++level) {
			if (level == up.size()) { for (; l < r; ++l) visit(level, l); break; }
			while (l < r && (l & (FAN - 1))) visit(level, l++);
			while (l < r && (r & (FAN - 1))) visit(level, --r);
			l >>= FAN_LOG; r >>= FAN_LOG;
		}
	}
	// the anchor of smallest priority in anchor i's interval, -1 if none; *tied when several hold it
	int lowest_for(int i, bool *tied) const
	{
		if (q_lo[(size_t)i] > q_hi[(size_t)i]) return -1;
		Sum best{ EMPTY, -1, 0 };
		tiles((size_t)q_lo[(size_t)i], (size_t)q_hi[(size_t)i], [&](size_t level, size_t at) { best = lower_of(best, sum_of(level, at)); });
		*tied = best.tie != 0;
		return best.at < 0 ? -1 : (int)(uint32_t)by_y[(size_t)best.at];
	}
	// every anchor of i's interval that holds the priority `pri` (the smallest one, when lowest_for reported a tie)
	void holders_of(int i, double pri, std::vector<int> &out) const
	{
		out.clear();
		if (q_lo[(size_t)i] > q_hi[(size_t)i]) return;
		tiles((size_t)q_lo[(size_t)i], (size_t)q_hi[(size_t)i], [&](size_t level, size_t at) { below(level, at, pri, out); });
	}
private:
	void below(size_t level, size_t at, double pri, std::vector<int> &out) const
	{
		if (sum_of(level, at).pri != pri) return;                // (the node lies inside the interval: its smallest priority is pri or above it)
		if (level == 0) { out.push_back((int)(uint32_t)by_y[at]); return; }
		for (size_t c = at << FAN_LOG; c < (at << FAN_LOG) + FAN; ++c) below(level - 1, c, pri, out);
	}
public:
};

struct InnerCand { int32_t y, j, x, f; int32_t span; };
// a bucket's anchors in (y, index) order, one array per field: the scan below reads eight candidates per instruction where AVX2 is there
struct InnerBucket {
	std::vector<int32_t> y, j, x, f, span;
	size_t gone = 0;                 // entries whose anchor has left the window (InnerWindow::erase)
	size_t size() const { return y.size(); }
	void clear() { y.clear(); j.clear(); x.clear(); f.clear(); span.clear(); gone = 0; }
};
struct InnerWindow {
	static constexpr int SHIFT = 6;
	std::vector<InnerBucket> bucket;
	std::vector<int32_t> top;      // per bucket: the largest f + span in it: no pair with one of its anchors can score more (lchain.c:237: sc <= q_span)
	int y0 = 0;
	size_t count = 0;
	void reset(int y_min, int y_max)
	{
		y0 = y_min; count = 0;
		const size_t nb = (size_t)((y_max - y_min) >> SHIFT) + 1;
		if (bucket.size() < nb) { bucket.resize(nb); top.resize(nb); }
		for (size_t b = 0; b < nb; ++b) { bucket[b].clear(); top[b] = INT32_MIN; }
	}
	void insert(const InnerCand &c)
	{
		const size_t b = (size_t)((c.y - y0) >> SHIFT);
		InnerBucket &v = bucket[b];
		size_t at = v.size();
		while (at > 0 && !(v.y[at - 1] != c.y ? v.y[at - 1] < c.y : v.j[at - 1] < c.j)) --at;       // arrivals come in order of x; within a bucket that is mostly near the end (an entry that has gone carries index -1: it sorts first among its y, which is as good as anywhere)
		v.y.insert(v.y.begin() + (ptrdiff_t)at, c.y); v.j.insert(v.j.begin() + (ptrdiff_t)at, c.j); v.x.insert(v.x.begin() + (ptrdiff_t)at, c.x);
		v.f.insert(v.f.begin() + (ptrdiff_t)at, c.f); v.span.insert(v.span.begin() + (ptrdiff_t)at, c.span);
		top[b] = std::max(top[b], c.f + c.span);
		++count;
	}
	// An anchor that leaves is not taken out at once: its index becomes -1 (the scans pass such entries over) and a bucket is packed when
	// half of it has gone -- taking one element out of five arrays and finding the bucket's largest f + span again was 60 ns an anchor.
	// `top` stays what it was until then: an upper bound.
	void erase(int y, int j)
	{
		const size_t b = (size_t)((y - y0) >> SHIFT);
		InnerBucket &v = bucket[b];
		const size_t n = v.size();
		const int32_t *vj = v.j.data();
		for (size_t at = 0; at < n; ++at)
			if (vj[at] == j) {
				v.j[at] = -1;
				--count;
				if (++v.gone * 2 > n) pack(b);
				return;
			}
	}
	void pack(size_t b)
	{
		InnerBucket &v = bucket[b];
		size_t to = 0;
		int32_t t = INT32_MIN;
		for (size_t at = 0; at < v.size(); ++at)
			if (v.j[at] >= 0) {
				v.y[to] = v.y[at]; v.j[to] = v.j[at]; v.x[to] = v.x[at]; v.f[to] = v.f[at]; v.span[to] = v.span[at];
				t = std::max(t, v.f[to] + v.span[to]);
				++to;
			}
		v.y.resize(to); v.j.resize(to); v.x.resize(to); v.f.resize(to); v.span.resize(to);
		v.gone = 0;
		top[b] = t;
	}
};

// The exhaustive inner scan of one bucket (lchain.c:328-341 with no skip limit and the penalty a function of the diagonal distance alone):
// its candidates from the largest (y, index) down, eight at a time; a strictly better score replaces the best, so among equal scores the
// first met -- the largest (y, index) -- stays.  AVX2; same integer arithmetic as the scalar loop (the penalty comes from the table).
#if defined(__x86_64__)
__attribute__((target("avx2"))) void scan_bucket_avx2(const InnerBucket &v, int xi, int yi, int y_top, int y_bot, int bw, const int32_t *pen, int &max_f, int &max_j)
{
	const __m256i vxi = _mm256_set1_epi32(xi), vyi = _mm256_set1_epi32(yi), vtop = _mm256_set1_epi32(y_top), vbot = _mm256_set1_epi32(y_bot), vbw = _mm256_set1_epi32(bw);
	const __m256i zero = _mm256_setzero_si256(), lowest = _mm256_set1_epi32(INT32_MIN);
	size_t hi = v.size();
	while (hi > 0) {
		const size_t lo = hi >= 8 ? hi - 8 : 0, cnt = hi - lo;
		// lanes 0 .. cnt-1 hold candidates lo .. hi-1 in ascending (y, index) order: the first met going down is the HIGHEST lane
		const __m256i lane = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
		const __m256i in_array = _mm256_cmpgt_epi32(_mm256_set1_epi32((int)cnt), lane);
		const __m256i cy = _mm256_maskload_epi32(v.y.data() + lo, in_array), cx = _mm256_maskload_epi32(v.x.data() + lo, in_array);
		const __m256i have = _mm256_andnot_si256(_mm256_cmpgt_epi32(zero, _mm256_maskload_epi32(v.j.data() + lo, in_array)), in_array);   // (index -1: the anchor has left the window)
		const __m256i cf = _mm256_maskload_epi32(v.f.data() + lo, have), cs = _mm256_maskload_epi32(v.span.data() + lo, have);
		const __m256i dq = _mm256_sub_epi32(vyi, cy), dr = _mm256_sub_epi32(vxi, cx);
		const __m256i dd = _mm256_abs_epi32(_mm256_sub_epi32(dr, dq)), dg = _mm256_min_epi32(dr, dq);
		// in range: y_bot <= y <= y_top, dd <= bw
		__m256i ok = _mm256_andnot_si256(_mm256_cmpgt_epi32(cy, vtop), have);
		ok = _mm256_andnot_si256(_mm256_cmpgt_epi32(vbot, cy), ok);
		ok = _mm256_andnot_si256(_mm256_cmpgt_epi32(dd, vbw), ok);
		if (_mm256_testz_si256(ok, ok)) { if (_mm256_movemask_ps(_mm256_castsi256_ps(_mm256_and_si256(in_array, _mm256_cmpgt_epi32(vbot, cy)))) == (int)((1u << cnt) - 1)) return; hi = lo; continue; }   // (entries that have gone keep their y: the order stands)
		const __m256i p = _mm256_mask_i32gather_epi32(zero, pen, dd, ok, 4);
		// sc = min(span, dg) - (dd != 0 || dq > span ? pen[dd] : 0)
		const __m256i charged = _mm256_or_si256(_mm256_xor_si256(_mm256_cmpeq_epi32(dd, zero), _mm256_set1_epi32(-1)), _mm256_cmpgt_epi32(dq, cs));
		const __m256i sc = _mm256_sub_epi32(_mm256_min_epi32(cs, dg), _mm256_and_si256(p, charged));
		const __m256i s2 = _mm256_blendv_epi8(lowest, _mm256_add_epi32(cf, sc), ok);
		// the largest score of the eight, and the highest lane that holds it
		__m256i m = _mm256_max_epi32(s2, _mm256_permute2x128_si256(s2, s2, 1));
		m = _mm256_max_epi32(m, _mm256_shuffle_epi32(m, 0x4e));
		m = _mm256_max_epi32(m, _mm256_shuffle_epi32(m, 0xb1));
		const int best = _mm256_cvtsi256_si32(m);
		if (best > max_f) {
			const unsigned who = (unsigned)_mm256_movemask_ps(_mm256_castsi256_ps(_mm256_and_si256(_mm256_cmpeq_epi32(s2, m), ok)));
			max_f = best; max_j = v.j[lo + (31 - (size_t)__builtin_clz(who))];
		}
		// every candidate below these is below the range too once the lowest of them is (a bucket is sorted by y)
		if (v.y[lo] < y_bot) return;
		hi = lo;
	}
}
#endif

struct FillScratch { ShapeTree tree; RankTree flat; InnerWindow inner; std::vector<int32_t> seen; std::vector<int> holders; long long ties_met = 0, ties_that_decide = 0; };   // seen: the reference's t[] (lchain.c:333-338)

// f[n], p_rel[n] (i - predecessor, 0 = none) of one read
// pen: (int)(gap * dd + .5 * mg_log2(dd + 1)) for dd = 0 .. bw when chn_pen_skip == 0 (the penalty then depends on dd alone), else null
// EXACT_SHAPE: with the reference's tree (always complete); otherwise with the tournament tree, giving up -- false -- at the first tie
template<bool EXACT_SHAPE>
bool rmq_fill_one(const mm2gb_rmq_param_t &P, int64_t n64, const mm2gb_anchor_t *a, int32_t *f, int32_t *p_rel, FillScratch &ws, const int32_t *pen)
{
	const int n = (int)n64;
	const int max_dist = P.max_dist < P.bw ? P.bw : P.max_dist;                                                   // lchain.c:264
	const int max_inner = (P.max_dist_inner <= 0 || P.max_dist_inner >= max_dist) ? 0 : P.max_dist_inner;         // lchain.c:265
	const double half_gap = 0.5 * (double)P.chn_pen_gap;
#if defined(__x86_64__)
	const bool use_avx2 = __builtin_cpu_supports("avx2") && !getenv("MM2GB_RMQ_NO_SIMD");   // (MM2GB_RMQ_NO_SIMD: the scalar scan, for A/B runs and tests; read once per read)
#endif
	auto &tree = [&]() -> auto& { if constexpr (EXACT_SHAPE) return ws.tree; else return ws.flat; }();
	const int weigh_ties = []() { const char *v = getenv("MM2GB_RMQ_TIES"); return v && !strcmp(v, "strict") ? 0 : 1; }();   // strict: every tie sends the read to the reference's tree (A/B runs, tests; read once per read)
	if constexpr (EXACT_SHAPE) tree.reset(a, n); else tree.reset(a, n, max_dist);
	ws.seen.assign((size_t)n, 0);
	if (max_inner > 0 && n > 0) {
		int y_min = INT32_MAX, y_max = INT32_MIN;
		for (int j = 0; j < n; ++j) { const int y = (int32_t)a[j].y; y_min = std::min(y_min, y); y_max = std::max(y_max, y); }
		ws.inner.reset(y_min, y_max);
	} else ws.inner.count = 0;
	int i0 = 0, st = 0, st_in = 0;
	for (int i = 0; i < n; ++i) {
		const int yi = (int32_t)a[i].y, q_i = (int)(a[i].y >> 32 & 0xff);
		if (i0 < i && a[i0].x != a[i].x) {                     // lchain.c:279-292: the anchors before the run of equal x that holds i go in
			for (int j = i0; j < i; ++j) {
				tree.insert(j, -((double)f[j] + half_gap * (double)((int32_t)a[j].x + (int32_t)a[j].y)));          // lchain.c:284
				if (max_inner > 0) ws.inner.insert(InnerCand{ (int32_t)a[j].y, j, (int32_t)a[j].x, f[j], (int32_t)(a[j].y >> 32 & 0xff) });
			}
			i0 = i;
		}
		// lchain.c:293-310: out of reach (other strand | reference, too far back) or too many in the tree
		while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + (uint64_t)max_dist || (int64_t)tree.size() > P.cap_rmq_size)) {
			if (st < i0) tree.erase(st);
			++st;
		}
		if (max_inner > 0)
			while (st_in < i && (a[i].x >> 32 != a[st_in].x >> 32 || a[i].x > a[st_in].x + (uint64_t)max_inner || (int64_t)ws.inner.count > P.cap_rmq_size)) {
				if (st_in < i0) ws.inner.erase((int32_t)a[st_in].y, st_in);
				++st_in;
			}
		int max_f = q_i, max_j = -1;
		// lchain.c:320-341: the inner window's anchors with y in [yi - max_inner, yi - 1], from the largest (y, index) down; a strictly better
		// score replaces the best; the walk gives up after max_chn_skip anchors whose own predecessor chain this anchor has already been
		// offered (the marks in seen[], lchain.c:333-338)
		const bool inner_there = max_inner > 0 && ws.inner.count > 0 && yi > 0;
		auto inner_scan = [&](int &max_f, int &max_j) {
			int n_skip = 0;
			const bool exhaustive = P.max_chn_skip == INT32_MAX;
			const int y_top = yi - 1, y_bot = yi - max_inner;
			const int xi = (int32_t)a[i].x;
			const int b_top = std::min<int>((int)ws.inner.bucket.size() - 1, (std::max(y_top, ws.inner.y0) - ws.inner.y0) >> InnerWindow::SHIFT);
			const int b_bot = (std::max(y_bot, ws.inner.y0) - ws.inner.y0) >> InnerWindow::SHIFT;
			bool stop = y_top < ws.inner.y0;
			for (int b = b_top; b >= b_bot && !stop; --b) {
				const InnerBucket &v = ws.inner.bucket[(size_t)b];
				// without a skip limit a candidate matters only if it beats the best so far, and none of this bucket can
				if (exhaustive && ws.inner.top[(size_t)b] <= max_f) continue;
#if defined(__x86_64__)
				if (exhaustive && pen && use_avx2) { scan_bucket_avx2(v, xi, yi, y_top, y_bot, P.bw, pen, max_f, max_j); continue; }
#endif
				for (size_t at = v.size(); at-- > 0;) {
					const int cy = v.y[at];
					if (cy > y_top) continue;
					if (cy < y_bot) break;
					// comput_sc_simple (lchain.c:232-248) on the copies
					const int cj = v.j[at], cspan = v.span[at];
					if (cj < 0) continue;                      // has left the window (InnerWindow::erase)
					const int dq = yi - cy, dr = xi - v.x[at], dd = dr > dq ? dr - dq : dq - dr;
					if (dd > P.bw) continue;
					const int dg = dr < dq ? dr : dq;
					int sc = cspan < dg ? cspan : dg;
					if (dd || dq > cspan) {
						if (pen) sc -= pen[dd];
						else {
							const float lin = P.chn_pen_gap * (float)dd + P.chn_pen_skip * (float)dg;
							const float lg = dd >= 1 ? log2_fit((float)(dd + 1)) : 0.0f;
							sc -= (int)(lin + .5f * lg);
						}
					}
					const int s2 = v.f[at] + sc;
					if (exhaustive) {                          // no skip limit: the marks of lchain.c:333-338 decide nothing -- no look at p[] and seen[] per candidate
						if (s2 > max_f) { max_f = s2; max_j = cj; }
						continue;
					}
					if (s2 > max_f) { max_f = s2; max_j = cj; if (n_skip > 0) --n_skip; }
					else if (ws.seen[(size_t)cj] == i) { if (++n_skip > P.max_chn_skip) { stop = true; break; } }
					if (p_rel[cj]) ws.seen[(size_t)(cj - p_rel[cj])] = i;
				}
			}
		};
		// lchain.c:311-315: the closed interval [(yi - max_dist, INT32_MAX), (yi, 0)] of (y, index)
		int j;
		bool weighed = false;
		if constexpr (EXACT_SHAPE) j = tree.lowest_between(yi - max_dist, INT32_MAX, yi, 0);
		else {
			bool tied = false;
			j = tree.lowest_for(i, &tied);
			if (tied) {
				// Several anchors hold the smallest priority and the reference's pick follows from the shape of its tree.  It need not be known
				// when every one of them leaves this anchor with the same score and predecessor: the tree's content does not depend on the
				// pick (priorities come from f[] alone), and without a skip limit neither does the inner walk -- its best is the first
				// candidate, in walking order, to reach the walk's largest score, wherever the walk started from below that.
				++ws.ties_met;
				// (a skip limit that can end a walk -- one below the tree's size cap, lchain.c:301-310, 329-333 -- makes the walk depend on where it started)
				const bool limit_can_end_a_walk = P.max_chn_skip != INT32_MAX && !(P.cap_rmq_size > 0 && P.max_chn_skip >= P.cap_rmq_size);
				if (limit_can_end_a_walk || weigh_ties == 0) { ++ws.ties_that_decide; return false; }
				int in_f = q_i, in_j = -1;
				if (inner_there) inner_scan(in_f, in_j);
				tree.holders_of(i, -((double)f[j] + half_gap * (double)((int32_t)a[j].x + (int32_t)a[j].y)), ws.holders);
				bool first = true, same = true;
				int res_f = q_i, res_j = -1;
				for (int q : ws.holders) {
					bool exact; int width;
					const int sc = f[q] + pair_score(a[i], a[q], P.chn_pen_gap, P.chn_pen_skip, &exact, &width);
					int o_f = q_i, o_j = -1;
					if (width <= P.bw && sc > o_f) { o_f = sc; o_j = q; }
					if (!exact && inner_there && in_f > o_f) { o_f = in_f; o_j = in_j; }
					if (first) { res_f = o_f; res_j = o_j; first = false; }
					else if (o_f != res_f || o_j != res_j) { same = false; break; }
				}
				if (!same || first) { ++ws.ties_that_decide; return false; }
				max_f = res_f; max_j = res_j; weighed = true;
			}
		}
		if (j >= 0 && !weighed) {
			bool exact; int width;
			const int sc = f[j] + pair_score(a[i], a[j], P.chn_pen_gap, P.chn_pen_skip, &exact, &width);
			if (width <= P.bw && sc > max_f) { max_f = sc; max_j = j; }
			if (!exact && inner_there) inner_scan(max_f, max_j);
		}
		f[i] = max_f;
		p_rel[i] = max_j < 0 ? 0 : i - max_j;
	}
	return true;
}

} // namespace
} // namespace mm2gb

using namespace mm2gb;

static int rmq_chain_host_impl(const mm2gb_rmq_param_t *prm, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, int n_threads,
                               mm2gb_chains_t *out, int32_t *n_tied, bool reference_tree_only)
{
	if (!prm || !out || n_reads < 0 || !offsets || offsets[0] != 0) return fail("mm2gb_rmq_chain_host: offsets[0] must be 0");
	memset(out, 0, sizeof(*out));
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
	// the pair penalty as a table when it depends on the diagonal distance alone (entries by the same arithmetic as pair_score)
	std::vector<int32_t> pen;
	if (prm->chn_pen_skip == 0.0f && prm->bw >= 0 && prm->bw < (1 << 22)) {
		pen.resize((size_t)prm->bw + 1);
		for (int dd = 0; dd <= prm->bw; ++dd) {
			const float lin = prm->chn_pen_gap * (float)dd + prm->chn_pen_skip * 0.0f;
			const float lg = dd >= 1 ? log2_fit((float)(dd + 1)) : 0.0f;
			pen[(size_t)dd] = (int)(lin + .5f * lg);
		}
	}
	// largest reads first: one read is one thread's work, and a read inside a tandem array can be a hundred times the median
	std::vector<int64_t> order((size_t)n_reads);
	for (int64_t r = 0; r < n_reads; ++r) order[(size_t)r] = r;
	std::sort(order.begin(), order.end(), [&](int64_t u, int64_t v) { const int64_t nu = offsets[u + 1] - offsets[u], nv = offsets[v + 1] - offsets[v]; return nu != nv ? nu > nv : u < v; });
	std::atomic<int64_t> next(0);
	HostAlloc libc_mem;
	if (n_tied) for (int64_t r = 0; r < n_reads; ++r) n_tied[r] = 0;   // becomes 1 for a read that met a tie and was done with the reference's tree
	const char *force = getenv("MM2GB_RMQ_TREE");
	const bool exact_only = reference_tree_only || (force && !strcmp(force, "avl"));   // MM2GB_RMQ_TREE=avl: the reference's tree for every read
	std::atomic<long long> ties_met(0), ties_that_decide(0);
	auto work = [&]() {
		FillScratch ws;
		BacktrackScratch bs;
		std::vector<int32_t> f, p;
		for (;;) {
			const int64_t at = next.fetch_add(1);
			if (at >= n_reads) break;
			const int64_t r = order[(size_t)at];
			const int64_t n = offsets[r + 1] - offsets[r];
			if (n == 0) continue;
			f.resize((size_t)n); p.resize((size_t)n);
			// the quick tree first; a read in which two anchors in range share the smallest priority -- few -- is done again with the reference's
			if (exact_only || !rmq_fill_one<false>(*prm, n, anchors + offsets[r], f.data(), p.data(), ws, pen.empty() ? nullptr : pen.data())) {
				rmq_fill_one<true>(*prm, n, anchors + offsets[r], f.data(), p.data(), ws, pen.empty() ? nullptr : pen.data());
				if (n_tied && !exact_only) n_tied[r] = 1;
			}
			nu_of[(size_t)r] = backtrack_compact(misc, n, anchors + offsets[r], f.data(), p.data(), libc_mem, bs, &u_of[(size_t)r], &a_of[(size_t)r]);
			for (int c = 0; c < nu_of[(size_t)r]; ++c) na_of[(size_t)r] += (uint32_t)u_of[(size_t)r][c];
		}
		ties_met += ws.ties_met; ties_that_decide += ws.ties_that_decide;
	};
	const int nt = std::max(1, n_threads);
	if (nt == 1) work();
	else { std::vector<std::thread> pool; for (int t = 0; t < nt; ++t) pool.emplace_back(work); for (auto &th : pool) th.join(); }
	if (getenv("MM2GB_DEBUG_PHASES") && !exact_only) {
		long long redone = 0;
		if (n_tied) for (int64_t r = 0; r < n_reads; ++r) redone += n_tied[r];
		fprintf(stderr, "[mm2gb rmq host] %lld reads: %lld ties met (several anchors on the smallest priority), %lld of them decide something; %lld reads done again with the reference's tree\n",
		        (long long)n_reads, ties_met.load(), ties_that_decide.load(), redone);
	}
	out->u_off = (int64_t*)malloc((R + 1) * 8);
	out->a_off = (int64_t*)malloc((R + 1) * 8);
	if (!out->u_off || !out->a_off) { mm2gb_chains_free(out); return fail("mm2gb_rmq_chain_host: out of memory"); }
	out->u_off[0] = out->a_off[0] = 0;
	for (size_t r = 0; r < R; ++r) { out->u_off[r + 1] = out->u_off[r] + nu_of[r]; out->a_off[r + 1] = out->a_off[r] + na_of[r]; }
	out->u = (uint64_t*)malloc(((size_t)out->u_off[R] + 1) * 8);
	out->a = (mm2gb_anchor_t*)mm2gb::result_alloc(((size_t)out->a_off[R] + 1) * 16);
	if (!out->u || !out->a) { mm2gb_chains_free(out); return fail("mm2gb_rmq_chain_host: out of memory"); }
	for (size_t r = 0; r < R; ++r) {
		if (nu_of[r] > 0) { memcpy(out->u + out->u_off[r], u_of[r], (size_t)nu_of[r] * 8); memcpy(out->a + out->a_off[r], a_of[r], (size_t)na_of[r] * 16); }
		free(u_of[r]); free(a_of[r]);
	}
	return 0;
}

extern "C" {

int mm2gb_rmq_chain_host(const mm2gb_rmq_param_t *prm, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, int n_threads,
                         mm2gb_chains_t *out, int32_t *n_tied)
{
	return rmq_chain_host_impl(prm, n_reads, offsets, anchors, n_threads, out, n_tied, false);
}

// Reads that are KNOWN to meet a tie (the device form counted it): the reference's tree at once, without the attempt that would stop at the tie
int mm2gb_rmq_chain_host_tied(const mm2gb_rmq_param_t *prm, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, int n_threads, mm2gb_chains_t *out)
{
	return rmq_chain_host_impl(prm, n_reads, offsets, anchors, n_threads, out, nullptr, true);
}

} // extern "C"
