// synth.cpp -- deterministic synthetic reads for benchmarks and large parity checks (recipe: SURVEY.md 8(d)).
// Every read is generated from its own xorshift64* stream seeded with  seed ^ 0x9E3779B97F4A7C15 ^ read_id, so any subset
// of reads can be regenerated anywhere (GPU bench ranks, the CPU baseline sample) and comes out identical.
//   (i)   true chain : floor(0.06 L) anchors on one (rid, strand); gaps U[1,33]; 1/8 chance each of an extra indel U[0,20) on x or y
//   (ii)  noise      : 3x as many anchors; rid U[0,24), strand U{0,1}, x U[0,1e8), y U[15,L)
//   (iii) repeats    : per 50 kb of read, with p = 0.3, a block of U[2000,12000] anchors with x inside a 4 kb window on the
//                      chain's (rid, strand) and y inside a 6 kb window  (windows saturate max_iter when the block has > 5000)
//   q_span = 15, seg_id = 0, sorted by x.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "engine.h"

namespace mm2gb {
namespace {

struct Rng {
	uint64_t s;
	explicit Rng(uint64_t seed) : s(seed ? seed : 0x2545F4914F6CDD1DULL) { for (int k = 0; k < 4; ++k) next(); }
	uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1DULL; }
	// uniform integer in [lo, hi)
	int64_t range(int64_t lo, int64_t hi) { return hi > lo ? lo + (int64_t)((next() >> 11) % (uint64_t)(hi - lo)) : lo; }
	double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

struct ReadPlan {
	int64_t length, n_true;
	std::vector<int> block_sizes;
	int64_t total() const { int64_t t = 4 * n_true; for (int b : block_sizes) t += b; return t; }
};

// Repeat-block probability per 50 kb of read: 0.3 in the SURVEY 8(d) recipe.  MM2GB_SYNTH_REPEAT_P overrides it for
// experiments (e.g. 0 = chains and noise only, windows of a few hundred anchors); read once.
double repeat_probability()
{
	static const double p = [] { const char *v = getenv("MM2GB_SYNTH_REPEAT_P"); return v && *v ? atof(v) : 0.3; }();
	return p;
}

// The header draws: everything that decides the anchor count.  The body continues from the same stream.
ReadPlan plan_read(Rng &rng, int len_lo, int len_hi)
{
	ReadPlan pl;
	pl.length = rng.range(len_lo, (int64_t)len_hi + 1);
	pl.n_true = std::max<int64_t>(1, (int64_t)(0.06 * (double)pl.length));
	const int n_win = (int)(pl.length / 50000);
	for (int w = 0; w < n_win; ++w) {
		const double u = rng.unit();
		const int sz = (int)rng.range(2000, 12001);
		if (u < repeat_probability()) pl.block_sizes.push_back(sz);
	}
	return pl;
}

inline mm2gb_anchor_t make_anchor(uint64_t rid, uint64_t rev, uint64_t rpos, uint64_t qpos)
{
	mm2gb_anchor_t a;
	a.x = rev << 63 | rid << 32 | (rpos & 0x7fffffffULL);
	a.y = (uint64_t)15 << 32 | (qpos & 0x7fffffffULL);
	return a;
}

void fill_read(uint64_t seed, int64_t read_id, int len_lo, int len_hi, mm2gb_anchor_t *out, int64_t expect)
{
	Rng rng(seed ^ 0x9E3779B97F4A7C15ULL ^ (uint64_t)read_id);
	const ReadPlan pl = plan_read(rng, len_lo, len_hi);
	int64_t k = 0;
	const uint64_t c_rid = (uint64_t)rng.range(0, 24), c_rev = (uint64_t)rng.range(0, 2);
	const int64_t c_x0 = rng.range(0, 90000000);
	int64_t x = c_x0, y = 15;
	for (int64_t t = 0; t < pl.n_true; ++t) {
		const int64_t gap = rng.range(1, 34);
		int64_t dx = gap, dy = gap;
		if ((rng.next() >> 20 & 7) == 0) dx += rng.range(0, 20);
		if ((rng.next() >> 20 & 7) == 0) dy += rng.range(0, 20);
		x += dx; y += dy;
		out[k++] = make_anchor(c_rid, c_rev, (uint64_t)x, (uint64_t)y);
	}
	const int64_t c_x1 = x, q_hi = std::max<int64_t>(pl.length, 16);
	for (int64_t t = 0; t < 3 * pl.n_true; ++t) {
		const uint64_t rid = (uint64_t)rng.range(0, 24), rev = (uint64_t)rng.range(0, 2);
		out[k++] = make_anchor(rid, rev, (uint64_t)rng.range(0, 100000000), (uint64_t)rng.range(15, q_hi));
	}
	for (int sz : pl.block_sizes) {
		const int64_t bx = rng.range(c_x0, std::max(c_x0 + 1, c_x1)), by = rng.range(15, std::max<int64_t>(16, pl.length - 6000));
		for (int t = 0; t < sz; ++t)
			out[k++] = make_anchor(c_rid, c_rev, (uint64_t)(bx + rng.range(0, 4000)), (uint64_t)(by + rng.range(0, 6000)));
	}
	(void)expect;
	std::sort(out, out + k, [](const mm2gb_anchor_t &a, const mm2gb_anchor_t &b) { return a.x != b.x ? a.x < b.x : a.y < b.y; });
}

} // namespace
} // namespace mm2gb

using namespace mm2gb;

extern "C" {

int64_t mm2gb_synth_count(uint64_t seed, int64_t first_read, int64_t n_reads, int len_lo, int len_hi, int64_t *offsets)
{
	if (n_reads < 0 || len_lo < 1 || len_hi < len_lo || !offsets) { fail("mm2gb_synth_count: bad arguments"); return -1; }
	offsets[0] = 0;
	for (int64_t r = 0; r < n_reads; ++r) {
		Rng rng(seed ^ 0x9E3779B97F4A7C15ULL ^ (uint64_t)(first_read + r));
		offsets[r + 1] = offsets[r] + plan_read(rng, len_lo, len_hi).total();
	}
	return offsets[n_reads];
}

int mm2gb_synth_fill(uint64_t seed, int64_t first_read, int64_t n_reads, int len_lo, int len_hi, const int64_t *offsets,
                     mm2gb_anchor_t *anchors, int n_threads)
{
	if (n_reads < 0 || !offsets || (offsets[n_reads] > 0 && !anchors)) return fail("mm2gb_synth_fill: bad arguments");
	if (n_threads < 1) n_threads = 1;
	std::atomic<int64_t> next(0);
	auto work = [&]() {
		for (;;) {
			const int64_t r = next.fetch_add(1);
			if (r >= n_reads) break;
			fill_read(seed, first_read + r, len_lo, len_hi, anchors + offsets[r], offsets[r + 1] - offsets[r]);
		}
	};
	if (n_threads == 1) work();
	else {
		std::vector<std::thread> pool;
		for (int t = 0; t < n_threads; ++t) pool.emplace_back(work);
		for (auto &th : pool) th.join();
	}
	return 0;
}

} // extern "C"
