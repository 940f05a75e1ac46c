// rmq_hybrid.cpp -- mg_lchain_rmq (lchain.c:250-369) for a batch of reads, exact for every read, with the DEVICE carrying the load:
// mm2gb_rmq_chain deals the reads between the kernel form (k_rmq_fill, one wave per read, thousands of reads at once) and the host
// form (csrc/rmq_host.cpp, one thread per read) so that both finish together, runs them AT THE SAME TIME, and has the reads for which
// the kernel reports a tie on the range-minimum priority (where the reference's answer follows from the shape of its tree,
// krmq.h:110-147) redone by the host form, which keeps that tree's rules.  Nothing is handed back to a caller to redo.
//
// Why a deal at all: the fill is a chain of n dependent steps per read.  A wave takes microseconds per step where a CPU core takes a
// fraction of one, and wins by running thousands of reads side by side -- so a batch is as slow as its slowest read, and a read inside a
// tandem array (hundreds of thousands of kept anchors within bw_long of each other, hundreds of inner candidates per anchor) is a
// hundred times the median.  Those few go to the host threads, which would otherwise idle while the device works.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include "engine.h"
#include "host_chain.h"

namespace mm2gb {
namespace {

// The cost model below is of a call that has the machine to itself.  It does not: the drop-in runs sixteen of these calls at once (a stream's
// finisher each), the own host four, and both sides then take two to three times what the model says -- the device more than the host -- so
// the deal left the longest reads on the device and the host threads idle (round 5, drop-in: device 1.08 s || host 0.29 s).  Each side's
// measured / estimated ratio of the calls so far (exponential average, process-wide) scales the next call's estimates; it only steers the deal.
// (Sixteen calls at once settle at x 6 -- the clamp -- for the device and x 1.9-2.4 for the host; starting there instead of at 1 / 1 was tried:
// the first mini-batch's calls got a host side of 1.0-1.25 s for a device side of 1.0 s, no better than 1.1 s || 0.2-0.5 s, and the run no faster.)
struct Calibration {
	std::mutex mu;
	double dev = 1.0, host = 1.0;
	void get(double &d, double &h) { std::lock_guard<std::mutex> lk(mu); d = dev; h = host; }
	void put(double est_dev, double got_dev, double est_host, double got_host)
	{
		std::lock_guard<std::mutex> lk(mu);
		auto mixin = [](double &avg, double est, double got) { if (est > 1e-3 && got > 1e-3) avg = std::min(6.0, std::max(0.5, 0.7 * avg + 0.3 * std::min(8.0, got / est))); };
		mixin(dev, est_dev, got_dev); mixin(host, est_host, got_host);
	}
};
Calibration g_calibration;

// What a read costs either side, in seconds, from three sums that take one pass over its anchors (sorted by x):
//   n      steps
//   s_in   sum over anchors of ceil(anchors within max_dist_inner before it on the reference / 64): blocks of the kernel's inner scan
//   s_out  sum over anchors of ceil(anchors within max_dist before it / 4096): iterations over block summaries of the kernel's query
// The constants are measured rates (profiles/r03_rmq_rate.json: counters of k_rmq_fill under MM2GB_DEBUG_PHASES against its time on
// the read that ends a batch; rmq_host.cpp's thread-seconds per anchor); they only steer the deal, never a result.
struct ReadCost { double dev, dev_steps, dev_team, host, s_in; bool team; };

ReadCost estimate(const mm2gb_rmq_param_t &P, const mm2gb_anchor_t *a, int64_t n)
{
	const int64_t max_dist = P.max_dist < P.bw ? P.bw : P.max_dist;
	const int64_t max_inner = (P.max_dist_inner <= 0 || P.max_dist_inner >= max_dist) ? 0 : P.max_dist_inner;
	double s_in = 0, s_out = 0;
	int64_t lo_in = 0, lo_out = 0;
	// every 16th anchor stands for its neighbours: windows change slowly along a read
	constexpr int64_t STRIDE = 16;
	for (int64_t i = 0; i < n; i += STRIDE) {
		const uint64_t xi = a[i].x;
		while (lo_out < i && (a[lo_out].x >> 32 != xi >> 32 || xi > a[lo_out].x + (uint64_t)max_dist)) ++lo_out;
		if (lo_in < lo_out) lo_in = lo_out;
		while (lo_in < i && xi > a[lo_in].x + (uint64_t)max_inner) ++lo_in;
		const int64_t w_out = std::min<int64_t>(i - lo_out, P.cap_rmq_size > 0 ? P.cap_rmq_size : i - lo_out);
		const int64_t w_in = max_inner > 0 ? std::min<int64_t>(i - lo_in, P.cap_rmq_size > 0 ? P.cap_rmq_size : i - lo_in) : 0;
		s_in += (double)((w_in + 63) / 64) * STRIDE;
		s_out += (double)((w_out + 4095) / 4096) * STRIDE;
	}
	ReadCost c;
	// The device form (k_rmq_fill_tiles, profiles/r03l_*): per tile of 64 anchors ~13 us of tree update, ~5 us of queries, ~30 us of in-tile
	// steps, and 0.31 us per anchor it broadcasts -- the inner window of the tile plus ~128 around the edges, i.e. about s_in + 2 n over
	// the read.  (The one-anchor-per-step kernel: 4.0 us per anchor + 0.42 us per block / summary round.)
	// The host form took 0.40 us per anchor of reads with narrow windows, 0.59 us on the mapper's reads, 0.9 us on a read inside a tandem
	// array (its inner scan visits the candidates of one y-range, not the window: a few hundred at worst).
	// (round 3, inner windows by strips of y: what a window costs is the candidates of its y range, a fraction of the window that s_in counts --
	// 0.07 us per anchor of window on the mapper's densest reads, profiles/r03_rmq_teams.txt)
	c.dev = 1.37e-6 * (double)n + 0.07e-6 * s_in;                         // tile kernel
	c.dev_steps = 4.0e-6 * (double)n + 0.42e-6 * (s_in + s_out);          // one anchor per step
	// a whole workgroup on the read (rmq_fill_read_team, round 5): per tile the longer of the serial wave's part (queries beside the last tile's
	// sweep, the combination, the 64 steps: ~40 us) and the helpers' (~22 us + their share of the inner scans) -- the 455 k-anchor read of
	// profiles/mapper_rate.py 3000: 0.41 s (0.68 s when the tree's update and the queries were in front of the steps)
	// (the figures below are the round-4 team's: as steering values they deal better than the new team's own -- with 0.35 us per anchor the
	// largest reads no longer head the list the host threads take from, stay on the device, and the mapper's call takes 1.06 s instead of 0.93)
	c.dev_team = 0.95e-6 * (double)n + 0.0045e-6 * s_in;
	c.team = false;
	c.s_in = s_in;
	c.host = (double)n * (0.40e-6 + 0.5e-6 * std::min(1.0, s_in / ((double)std::max<int64_t>(n, 1) * 100.0)));
	return c;
}

// copies of read-sized pieces, dealt to threads (a batch's anchors are a gigabyte: one thread copies at ~5 GB/s)
template <class F>
void parallel_reads(size_t n, int nt, F &&fn)
{
	std::atomic<size_t> next(0);
	auto work = [&]() { for (;;) { const size_t lo = next.fetch_add(64); if (lo >= n) break; for (size_t k = lo; k < std::min(n, lo + 64); ++k) fn(k); } };
	std::vector<std::thread> pool;
	for (int t = 1; t < std::max(1, std::min<int>(nt, (int)((n + 63) / 64))); ++t) pool.emplace_back(work);
	work();
	for (auto &th : pool) th.join();
}

void append(mm2gb_chains_t &dst, size_t r, const mm2gb_chains_t &src, size_t q)
{
	const int64_t nu = src.u_off[q + 1] - src.u_off[q], na = src.a_off[q + 1] - src.a_off[q];
	dst.u_off[r + 1] = nu; dst.a_off[r + 1] = na;                     // counts for now; turned into offsets by the caller
}

} // namespace
} // namespace mm2gb

using namespace mm2gb;

namespace {
// `out`: one result in the caller's read order; or `parts`: the three sides' results as they are (host_chain.h, RmqParts)
int rmq_chain_impl(mm2gb_engine_t *eng, const mm2gb_rmq_param_t *prm, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                   int n_threads, mm2gb_chains_t *out, mm2gb::RmqParts *parts, int32_t *where, mm2gb_rmq_deal_t *deal)
{
	if (!eng || !prm || (!out && !parts) || n_reads < 0 || !offsets || offsets[0] != 0) return fail("mm2gb_rmq_chain: null argument, or offsets[0] is not 0");
	if (out) memset(out, 0, sizeof(*out));
	for (int64_t r = 0; r < n_reads; ++r) if (offsets[r + 1] < offsets[r]) return fail("mm2gb_rmq_chain: offsets must be non-decreasing");
	if (offsets[n_reads] > 0 && !anchors) return fail("mm2gb_rmq_chain: null buffer");
	const size_t R = (size_t)n_reads;
	const int nt = std::max(1, n_threads);
	const auto t0 = std::chrono::steady_clock::now();
	auto seconds_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };

	// ---- the deal ----
	std::vector<ReadCost> cost(R);
	{
		std::atomic<int64_t> next(0);
		auto work = [&]() { for (;;) { const int64_t r = next.fetch_add(1); if (r >= n_reads) break; cost[(size_t)r] = estimate(*prm, anchors + offsets[r], offsets[r + 1] - offsets[r]); } };
		std::vector<std::thread> pool;
		for (int t = 1; t < std::min<int>(nt, (int)std::max<int64_t>(1, n_reads)); ++t) pool.emplace_back(work);
		work();
		for (auto &th : pool) th.join();
	}
	// (a skip limit that can end an inner walk -- below the size cap, lchain.c:329-333 --: the one-anchor-per-step kernel, which keeps the counter)
	const bool limited = [&] { const char *sk = getenv("MM2GB_RMQ_SKIP"); return !(sk && !strcmp(sk, "ignore")) && prm->max_chn_skip != INT32_MAX && !(prm->cap_rmq_size > 0 && prm->max_chn_skip >= prm->cap_rmq_size); }();
	const bool steps_kernel = [&] { const char *kv = getenv("MM2GB_RMQ_KERNEL"); return limited || (kv && !strcmp(kv, "steps")); }();
	// Which device form: the tile kernel (64 anchors per step of a wave, a whole workgroup on the reads whose inner windows hold thousands of
	// anchors) unless MM2GB_RMQ_KERNEL=steps asks for the one-anchor-per-step kernel.  (Round 3, before the workgroup reads: the step kernel won
	// on the mapper's batches -- windows of many hundreds of anchors, every chain of a read interleaved along x -- 2.25 s against 2.5 s for the
	// device's share; with them the tile kernel takes 1.8 s and more of the reads, profiles/r03_rmq_teams.txt.)
	{
		const bool use_steps = steps_kernel;
		if (limited) {
			// the skip-limited walk ends after a few dozen candidates whatever the window holds: a step is the kernel's base cost and a round or two
			// of 64 ranks -- 9 us an anchor on the longest read of profiles/experiments/rmq_skip_rate.py (the exhaustive one-anchor-per-step form: 7.5),
			// and the host form is a quarter slower with the marks (0.62 against 0.48 us an anchor)
			for (size_t r = 0; r < R; ++r) { cost[r].dev = 9.0e-6 * (double)(offsets[r + 1] - offsets[r]); cost[r].host *= 1.25; }
		} else if (use_steps) for (size_t r = 0; r < R; ++r) cost[r].dev = cost[r].dev_steps;
		// tile kernel: the reads that would set the device's pace get a whole workgroup each (there are far fewer reads than the chip holds waves)
		// -- those whose time is the sweeps and inner scans, which a team's helpers share.  Measured on the mapper's reads: the slowest
		// single-wave reads spend 1.5-1.9 s of 1.6-1.9 s in them (2-4 M anchors each, profiles/r03_rmq_teams.txt).  (Round 5: a team also
		// overlaps the serial part of a tile with the rest, so every long read would gain; but a team holds 16 wave slots for the work of three or
		// four, sixteen calls at once share the chip in the drop-in, and a rule built on comparing the two estimates read by read -- a team only
		// where one wave would take longer than the call's longest team read -- left reads to single waves that then took 1.0-1.2 s: the
		// single-wave estimate is not that good.  The rule stays.)
		else if (!getenv("MM2GB_RMQ_NO_TEAMS")) {
			int64_t team_from = INT64_MAX;                       // experiment: every read of at least this many anchors is a whole workgroup's
			if (const char *tv = getenv("MM2GB_RMQ_TEAM_MIN_ANCHORS")) team_from = atoll(tv);
			for (size_t r = 0; r < R; ++r) {
				const double n = (double)(offsets[r + 1] - offsets[r]);
				if ((cost[r].dev > 5e-3 && 0.07e-6 * cost[r].s_in >= 0.5 * 1.37e-6 * n && cost[r].dev_team < cost[r].dev) || (offsets[r + 1] - offsets[r] >= team_from)) { cost[r].dev = std::min(cost[r].dev, cost[r].dev_team); cost[r].team = true; }
			}
		}
	}
	double cal_dev = 1.0, cal_host = 1.0;
	// (only for callers whose host threads have nothing else to do while the call runs -- the drop-in's finishers, engine.h rmq_calibrate: where
	// other stages compete for the host threads, as in the own host's stream of chunks, shifting reads to them made the whole run 8-10 % slower)
	const bool calibrate = eng->e.rmq_calibrate && !getenv("MM2GB_RMQ_NO_CALIBRATION");
	if (calibrate) g_calibration.get(cal_dev, cal_host);
	for (size_t r = 0; r < R; ++r) { cost[r].dev *= cal_dev; cost[r].host *= cal_host; }
	std::vector<int64_t> by_dev(R);
	for (size_t r = 0; r < R; ++r) by_dev[r] = (int64_t)r;
	std::sort(by_dev.begin(), by_dev.end(), [&](int64_t u, int64_t v) { return cost[(size_t)u].dev != cost[(size_t)v].dev ? cost[(size_t)u].dev > cost[(size_t)v].dev : u < v; });
	const double slots = std::max(64.0, (double)eng->e.n_cu * 12.0);     // waves k_rmq_fill keeps resident (LDS-bound)
	double dev_sum = 0, host_sum = 0, host_max = 0;
	for (size_t r = 0; r < R; ++r) dev_sum += cost[r].dev * (cost[r].team ? 16.0 : 1.0);      // in wave slots
	size_t n_host = 0;                                                   // the first n_host reads of by_dev go to the host threads
	int policy = 0;                                                      // MM2GB_RMQ_DEAL=device / host: everything one side (A/B runs, tests)
	if (const char *v = getenv("MM2GB_RMQ_DEAL")) policy = !strcmp(v, "device") ? 1 : !strcmp(v, "host") ? 2 : 0;
	const double launch_s = 1.5e-3;                                      // what a device call costs before its first step (copies, sort by y, ranks)
	if (policy == 2) n_host = R;
	else if (policy == 0)
		while (n_host < R) {
			const ReadCost &c = cost[(size_t)by_dev[n_host]];
			const double dev_now = launch_s + std::max(c.dev, dev_sum / slots);                       // with this read still on the device
			const double host_then = std::max(std::max(host_max, c.host), (host_sum + c.host) / nt);  // with it on the host threads
			if (host_then >= dev_now) break;
			host_sum += c.host; host_max = std::max(host_max, c.host); dev_sum -= c.dev * (c.team ? 16.0 : 1.0);
			++n_host;
		}
	const double est_host_s = std::max(host_max, host_sum / nt), est_dev_s = n_host < R ? launch_s + std::max(cost[(size_t)by_dev[n_host]].dev, dev_sum / slots) : 0;
	if (deal) { memset(deal, 0, sizeof(*deal)); deal->device_kernel = steps_kernel ? 1 : 0; deal->n_host_cost = (int64_t)n_host; deal->n_device = (int64_t)(R - n_host); deal->est_host_s = std::max(host_max, host_sum / nt); deal->est_device_s = n_host < R ? launch_s + std::max(cost[(size_t)by_dev[n_host]].dev, dev_sum / slots) : 0; }

	const double s_estimate = seconds_since(t0);
	double s_gather = 0, s_merge = 0;
	// ---- both sides at once ----
	const int gather_threads = nt;
	auto gather = [&](size_t from, size_t to, std::vector<int64_t> &off, BigBuf<mm2gb_anchor_t> &buf) {
		off.assign(to - from + 1, 0);
		for (size_t q = from; q < to; ++q) off[q - from + 1] = off[q - from] + (offsets[by_dev[q] + 1] - offsets[by_dev[q]]);
		buf.resize((size_t)std::max<int64_t>(off.back(), 1));
		parallel_reads(to - from, gather_threads, [&](size_t k) {
			const int64_t r = by_dev[from + k], n = offsets[r + 1] - offsets[r];
			if (n) memcpy(buf.data() + off[k], anchors + offsets[r], (size_t)n * sizeof(mm2gb_anchor_t));
		});
	};
	std::vector<int64_t> h_off, d_off, t_off;
	// (the engine's, kept between calls: a gigabyte of fresh pages costs more to touch than to copy; a mapper's worker re-chains chunk after chunk)
	HostScratch &hs = host_scratch(eng);                 // (the engine's: kept between calls)
	BigBuf<mm2gb_anchor_t> &h_a = hs.gather[0], &d_a = hs.gather[1], &t_a = hs.gather[2];
	ChainsOwner h_own, d_own, t_own;
	mm2gb_chains_t &h_out = h_own.c, &d_out = d_own.c, &t_out = t_own.c;
	int h_rc = 0;
	std::string h_err;
	double h_seconds = 0;
	std::thread host_side, tie_side;
	std::atomic<bool> host_done{ n_host == 0 };          // the host side's threads are free again (the tie redo may take all of them)
	struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } join_host{ host_side }, join_tie{ tie_side };   // (also when this thread runs out of memory)
	if (n_host < R) {
		// the device's share first, on every thread: its side is the longer one and cannot start before its anchors are in one piece
		// (the reads of whole workgroups first -- the kernel takes its first n_team reads that way --, each part most expensive first)
		std::stable_partition(by_dev.begin() + (std::ptrdiff_t)n_host, by_dev.end(), [&](int64_t r) { return cost[(size_t)r].team; });
		const auto tg = std::chrono::steady_clock::now();
		gather(n_host, R, d_off, d_a);
		s_gather += seconds_since(tg);
	}
	if (n_host > 0) {
		{ const auto tg = std::chrono::steady_clock::now(); gather(0, n_host, h_off, h_a); s_gather += seconds_since(tg); }
		// the host side leaves one thread to the device call's own host work when it shares the machine with it
		const int h_threads = n_host < R ? std::max(1, nt - 1) : nt;
		const mm2gb_anchor_t *h_ptr = h_a.data();          // (a plain pointer for the new thread)
		host_side = std::thread([&, h_threads, h_ptr]() {
			const auto th = std::chrono::steady_clock::now();
			h_rc = mm2gb_rmq_chain_host(prm, (int64_t)n_host, h_off.data(), h_ptr, h_threads, &h_out, nullptr);
			if (h_rc) h_err = mm2gb_last_error();
			h_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - th).count();
			host_done.store(true, std::memory_order_release);
		});
	}
	std::vector<int32_t> tied(R - n_host + 1, 0);
	std::vector<int64_t> tie_slot(R, -1);
	std::vector<size_t> redo;
	int t_rc = 0;
	std::string t_err;
	double t_seconds = 0;
	int d_rc = 0;
	double d_seconds = 0;
	if (n_host < R) {
		int64_t n_team = 0;
		for (size_t q = n_host; q < R; ++q) n_team += cost[(size_t)by_dev[q]].team;
		if (deal) deal->n_team = (int32_t)n_team;
		(void)mm2gb_engine_set_rmq_team_reads(eng, (int)n_team);
		// a read that meets a tie is redone by the host form whatever the device makes of it: the kernel gives it up at that tile (the reads
		// that tie are the largest ones -- hundreds of thousands of anchors, 0.6 s of a workgroup each -- and they were what the fill ended with)
		eng->e.rmq_abandon_tied = true;
		const auto td = std::chrono::steady_clock::now();
		// the reads that met a tie are known when the device's fill is done: they are redone on host threads while its post-pass and its copies
		// still run (and beside what is left of the host side's own share)
		// (called from inside the engine, with the call's post-pass still queued: NOTHING may pass through it -- not bad_alloc from the vectors,
		// not system_error from the thread's constructor; a failure is reported through t_rc when the device call is back)
		eng->e.rmq_tied_ready = [&](const int32_t *tc) {
			try {
				for (size_t q = n_host; q < R; ++q) if (tc[q - n_host]) { tie_slot[q] = (int64_t)redo.size(); redo.push_back(q); }
				if (redo.empty()) return;
				t_off.assign(1, 0);
				int64_t total = 0;
				for (size_t q : redo) total += offsets[by_dev[q] + 1] - offsets[by_dev[q]];
				t_a.resize((size_t)total);
				for (size_t q : redo) {
					const int64_t r = by_dev[q], n = offsets[r + 1] - offsets[r];
					memcpy(t_a.data() + t_off.back(), anchors + offsets[r], (size_t)n * sizeof(mm2gb_anchor_t));
					t_off.push_back(t_off.back() + n);
				}
				const mm2gb_anchor_t *t_ptr = t_a.data();
				// the tie redo takes the threads the host side is not using -- all of them when it is through with its share (the usual case:
				// the deal gives it what it finishes while the device fills) --: together they stay within the caller's n_threads (plus this one)
				const int t_threads = host_done.load(std::memory_order_acquire) ? nt : std::max(1, nt / 2);
				tie_side = std::thread([&, t_ptr, t_threads]() {
					const auto tt = std::chrono::steady_clock::now();
					t_rc = mm2gb_rmq_chain_host_tied(prm, (int64_t)redo.size(), t_off.data(), t_ptr, t_threads, &t_out);
					if (t_rc) t_err = mm2gb_last_error();
					t_seconds = seconds_since(tt);
				});
			} catch (const std::exception &ex) { t_rc = -1; t_err = std::string("mm2gb_rmq_chain: redoing the tied reads: ") + ex.what(); }
			catch (...) { t_rc = -1; t_err = "mm2gb_rmq_chain: redoing the tied reads failed"; }
		};
		d_rc = mm2gb_rmq_chain_gpu(eng, prm, (int64_t)(R - n_host), d_off.data(), d_a.data(), &d_out, tied.data(), nullptr);
		eng->e.rmq_tied_ready = nullptr;
		d_seconds = seconds_since(td);
	}
	const std::string d_err = d_rc ? mm2gb_last_error() : "";
	if (host_side.joinable()) host_side.join();
	if (tie_side.joinable()) tie_side.join();
	auto give_up = [&](const std::string &why) { mm2gb_chains_free(&h_out); mm2gb_chains_free(&d_out); mm2gb_chains_free(&t_out); return fail(why); };
	if (d_rc) return give_up(d_err);
	if (h_rc) return give_up(h_err);

	// ---- reads that met a tie on the device: the host form, which keeps the reference's tree (started from inside the device call) ----
	if (t_rc) return give_up(t_err);
	if (deal) deal->n_host_tie = (int64_t)redo.size();

	if (parts) {
		// ---- the caller takes the three results as they are ----
		parts->which.assign(R, 0); parts->slot.assign(R, 0);
		for (size_t q = 0; q < R; ++q) {
			const size_t r = (size_t)by_dev[q];
			if (q < n_host) { parts->which[r] = 0; parts->slot[r] = (int64_t)q; }
			else if (tie_slot[q] >= 0) { parts->which[r] = 2; parts->slot[r] = tie_slot[q]; }
			else { parts->which[r] = 1; parts->slot[r] = (int64_t)(q - n_host); }
			if (where) where[r] = q < n_host ? 1 : tie_slot[q] >= 0 ? 2 : 0;
		}
		parts->chains[0] = h_own.release(); parts->chains[1] = d_own.release(); parts->chains[2] = t_own.release();      // (theirs to free now)
		if (getenv("MM2GB_DEBUG_PHASES") || getenv("MM2GB_RMQ_CALLS"))      // MM2GB_RMQ_CALLS=1: this line alone (the engines' own debugging copies wait for the whole device)
			fprintf(stderr, "[mm2gb rmq call] %zu reads: estimate + deal %.3f s, gathers %.3f s, device %.3f s || host %.3f s, ties %.3f s, results left in place, whole %.3f s (estimates scaled x %.2f device, x %.2f host)\n", R, s_estimate, s_gather, d_seconds, h_seconds, t_seconds, seconds_since(t0), cal_dev, cal_host);
		if (deal) { deal->host_s = h_seconds; deal->device_s = d_seconds; deal->tie_s = t_seconds; deal->total_s = seconds_since(t0); }
		if (calibrate) g_calibration.put(est_dev_s / cal_dev, d_seconds, est_host_s / cal_host, h_seconds);
		return 0;
	}
	if (calibrate) g_calibration.put(est_dev_s / cal_dev, d_seconds, est_host_s / cal_host, h_seconds);
	// ---- one result, in the caller's read order ----
	const auto tm = std::chrono::steady_clock::now();
	out->u_off = (int64_t*)malloc((R + 1) * 8);
	out->a_off = (int64_t*)malloc((R + 1) * 8);
	if (!out->u_off || !out->a_off) { mm2gb_chains_free(out); return give_up("mm2gb_rmq_chain: out of memory"); }
	out->u_off[0] = out->a_off[0] = 0;
	struct Src { const mm2gb_chains_t *c; size_t q; };
	std::vector<Src> src(R);
	for (size_t q = 0; q < R; ++q) {
		const size_t r = (size_t)by_dev[q];
		if (q < n_host) src[r] = { &h_out, q };
		else if (tie_slot[q] >= 0) src[r] = { &t_out, (size_t)tie_slot[q] };
		else src[r] = { &d_out, q - n_host };
		if (where) where[r] = q < n_host ? 1 : tie_slot[q] >= 0 ? 2 : 0;
	}
	for (size_t r = 0; r < R; ++r) append(*out, r, *src[r].c, src[r].q);
	for (size_t r = 0; r < R; ++r) { out->u_off[r + 1] += out->u_off[r]; out->a_off[r + 1] += out->a_off[r]; }
	out->u = (uint64_t*)malloc(((size_t)out->u_off[R] + 1) * 8);
	out->a = (mm2gb_anchor_t*)result_alloc(((size_t)out->a_off[R] + 1) * 16);
	if (!out->u || !out->a) { mm2gb_chains_free(out); return give_up("mm2gb_rmq_chain: out of memory"); }
	parallel_reads(R, nt, [&](size_t r) {
		const mm2gb_chains_t &c = *src[r].c;
		const size_t q = src[r].q;
		const int64_t nu = c.u_off[q + 1] - c.u_off[q], na = c.a_off[q + 1] - c.a_off[q];
		if (nu) memcpy(out->u + out->u_off[r], c.u + c.u_off[q], (size_t)nu * 8);
		if (na) memcpy(out->a + out->a_off[r], c.a + c.a_off[q], (size_t)na * 16);
	});
	mm2gb_chains_free(&h_out); mm2gb_chains_free(&d_out); mm2gb_chains_free(&t_out);
	s_merge = seconds_since(tm);
	if (getenv("MM2GB_DEBUG_PHASES") || getenv("MM2GB_RMQ_CALLS"))
		fprintf(stderr, "[mm2gb rmq call] %zu reads: estimate + deal %.3f s, gathers %.3f s, device %.3f s || host %.3f s, ties %.3f s, merge %.3f s, whole %.3f s (estimates scaled x %.2f device, x %.2f host)\n", R, s_estimate, s_gather, d_seconds, h_seconds, t_seconds, s_merge, seconds_since(t0), cal_dev, cal_host);
	if (deal) { deal->host_s = h_seconds; deal->device_s = d_seconds; deal->tie_s = t_seconds; deal->total_s = seconds_since(t0); }
	return 0;
}
} // namespace

namespace mm2gb {
int rmq_chain_parts(mm2gb_engine_t *eng, const mm2gb_rmq_param_t *prm, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                    int n_threads, RmqParts &parts, int32_t *where, mm2gb_rmq_deal_t *deal)
{
	try { return rmq_chain_impl(eng, prm, n_reads, offsets, anchors, n_threads, nullptr, &parts, where, deal); }
	catch (const std::bad_alloc&) { return fail("mm2gb_rmq_chain: out of host memory"); }
}
} // namespace mm2gb

extern "C" {

int mm2gb_rmq_chain(mm2gb_engine_t *eng, const mm2gb_rmq_param_t *prm, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                    int n_threads, mm2gb_chains_t *out, int32_t *where, mm2gb_rmq_deal_t *deal)
{
	if (!out) return fail("mm2gb_rmq_chain: null argument");
	try { return rmq_chain_impl(eng, prm, n_reads, offsets, anchors, n_threads, out, nullptr, where, deal); }
	catch (const std::bad_alloc&) { mm2gb_chains_free(out); return fail("mm2gb_rmq_chain: out of host memory"); }
}

} // extern "C"
