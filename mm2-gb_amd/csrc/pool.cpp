// pool.cpp -- whole batches on host buffers: one or several devices score, a pool of host threads backtracks.
//   * mm2gb_chain_host      : one engine (one device)
//   * mm2gb_pool_*          : several engines in one process, reads dealt to devices as contiguous runs balanced by anchor
//                             count; no data moves between devices (reads are independent, no collective)
// Both run the host post-pass (host_chain.cpp) on reads as soon as the slice that holds them is back from the device, while
// later slices are still being copied and scored, so the devices and the host threads work at the same time.
// Replaces plchain_cal_score_async + plchain_post_gpu_helper for callers that own whole batches (plchain.cu:201-464).
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "engine.h"
#include "host_chain.h"

struct mm2gb_pool {
	std::vector<mm2gb_engine_t*> engines;     // owned
};

namespace mm2gb {
namespace {

// Runs of reads whose scores are in host memory, consumed a few reads at a time by the post-pass threads.
class ReadyReads {
public:
	explicit ReadyReads(const int64_t *offsets) : off_(offsets) {}
	void push(int64_t r0, int64_t r1)
	{
		if (r1 <= r0) return;
		{ std::lock_guard<std::mutex> g(mu_); runs_.emplace_back(r0, r1); }
		cv_.notify_all();
	}
	void close()
	{
		{ std::lock_guard<std::mutex> g(mu_); closed_ = true; }
		cv_.notify_all();
	}
	// next group of reads [r0, r1): about GRAIN anchors, so that tiny reads do not pay one lock each
	bool pop(int64_t &r0, int64_t &r1)
	{
		std::unique_lock<std::mutex> g(mu_);
		cv_.wait(g, [&] { return closed_ || !runs_.empty(); });
		if (runs_.empty()) return false;
		auto &run = runs_.front();
		r0 = run.first;
		r1 = r0 + 1;
		while (r1 < run.second && off_[r1] - off_[r0] < GRAIN) ++r1;
		run.first = r1;
		if (run.first >= run.second) runs_.pop_front();
		return true;
	}
private:
	static constexpr int64_t GRAIN = 1 << 16;
	const int64_t *off_;
	std::mutex mu_;
	std::condition_variable cv_;
	std::deque<std::pair<int64_t, int64_t>> runs_;
	bool closed_ = false;
};

void add_stats(mm2gb_stats_t &sum, const mm2gb_stats_t &s)
{
	sum.n_anchors += s.n_anchors; sum.n_reads += s.n_reads; sum.n_pairs += s.n_pairs; sum.n_chunks += s.n_chunks;
	sum.n_long_chunks += s.n_long_chunks; sum.n_mid_chunks += s.n_mid_chunks; sum.n_tracked_chunks += s.n_tracked_chunks;
	sum.n_clamped_blocks += s.n_clamped_blocks;
	// devices run side by side: the slowest one is the time of the call
	sum.ms_h2d = std::max(sum.ms_h2d, s.ms_h2d); sum.ms_prep = std::max(sum.ms_prep, s.ms_prep); sum.ms_score = std::max(sum.ms_score, s.ms_score);
	sum.ms_d2h = std::max(sum.ms_d2h, s.ms_d2h); sum.ms_total = std::max(sum.ms_total, s.ms_total);
}

// Contiguous runs of reads, one per device, with about the same number of anchors each: first[d] .. first[d+1].
std::vector<int64_t> deal_reads(int64_t n_reads, const int64_t *offsets, int n_dev)
{
	std::vector<int64_t> first((size_t)n_dev + 1, n_reads);
	first[0] = 0;
	const int64_t n = offsets[n_reads];
	int64_t r = 0;
	for (int d = 1; d < n_dev; ++d) {
		const int64_t want = (int64_t)((__int128)n * d / n_dev);
		while (r < n_reads && offsets[r] < want) ++r;
		// the boundary read goes to whichever side leaves the shares closer to even
		if (r > first[d - 1] && offsets[r] - want > want - offsets[r - 1]) --r;
		first[d] = r;
	}
	return first;
}

// Scores on every engine (one host thread each when there are several), slice_done forwarded with batch-level read ids.
int score_on_engines(const std::vector<mm2gb_engine_t*> &engines, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                     int32_t *f, int32_t *p, mm2gb_stats_t *stats, int64_t *first_read_of_device, ReadyReads *ready)
{
	const int n_dev = (int)engines.size();
	const std::vector<int64_t> first = deal_reads(n_reads, offsets, n_dev);
	if (first_read_of_device) memcpy(first_read_of_device, first.data(), ((size_t)n_dev + 1) * sizeof(int64_t));
	std::vector<int> rc((size_t)n_dev, 0);
	std::vector<std::string> err((size_t)n_dev);
	auto run = [&](int d) {
		const int64_t r0 = first[d], r1 = first[d + 1];
		std::function<void(int64_t, int64_t)> forward = [&, r0](int64_t a, int64_t b) { ready->push(r0 + a, r0 + b); };
		rc[d] = engines[d]->e.score_host(r1 - r0, offsets + r0, anchors, f, p, ready ? &forward : nullptr);
		if (rc[d]) err[d] = mm2gb_last_error();          // the error text is per thread: carry it to the caller's
	};
	if (n_dev == 1) run(0);
	else {
		std::vector<std::thread> th;
		for (int d = 0; d < n_dev; ++d) th.emplace_back(run, d);
		for (auto &t : th) t.join();
	}
	mm2gb_stats_t sum = {};
	for (int d = 0; d < n_dev; ++d) {
		if (rc[d]) return fail("device " + std::to_string(engines[d]->e.device) + ": " + err[d]);
		add_stats(sum, engines[d]->e.last);
	}
	if (stats) *stats = sum;
	return 0;
}

int chain_on_engines(const std::vector<mm2gb_engine_t*> &engines, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                     int n_threads, mm2gb_chains_t *out, mm2gb_stats_t *stats)
{
	if (!out || !offsets || n_reads < 0 || engines.empty()) return fail("mm2gb_chain_host: null argument");
	memset(out, 0, sizeof(*out));
	if (offsets[0] != 0) return fail("mm2gb_chain_host: offsets[0] must be 0");
	if (n_threads < 1) n_threads = 1;
	const int64_t n = offsets[n_reads];
	// scores land here; no need to clear 8 bytes per anchor first
	std::unique_ptr<int32_t[]> f(new int32_t[(size_t)(n > 0 ? n : 1)]), p(new int32_t[(size_t)(n > 0 ? n : 1)]);
	std::vector<uint64_t*> u_of((size_t)n_reads, nullptr);
	std::vector<mm2gb_anchor_t*> a_of((size_t)n_reads, nullptr);
	std::vector<int> nu_of((size_t)n_reads, 0);
	const mm2gb_misc_t misc = engines[0]->e.misc;
	const HostAlloc mem;
	ReadyReads ready(offsets);
	std::vector<std::thread> post;
	for (int t = 0; t < n_threads; ++t)
		post.emplace_back([&]() {
			BacktrackScratch ws;
			int64_t r0, r1;
			while (ready.pop(r0, r1))
				for (int64_t r = r0; r < r1; ++r)
					nu_of[r] = backtrack_compact(misc, offsets[r + 1] - offsets[r], anchors + offsets[r], f.get() + offsets[r], p.get() + offsets[r],
					                             mem, ws, &u_of[r], &a_of[r]);
		});
	const int rc = score_on_engines(engines, n_reads, offsets, anchors, f.get(), p.get(), stats, nullptr, &ready);
	ready.close();
	for (auto &t : post) t.join();
	if (rc) {
		for (int64_t r = 0; r < n_reads; ++r) { free(u_of[r]); free(a_of[r]); }
		return -1;
	}
	out->u_off = (int64_t*)malloc((size_t)(n_reads + 1) * 8);
	out->a_off = (int64_t*)malloc((size_t)(n_reads + 1) * 8);
	out->u_off[0] = out->a_off[0] = 0;
	for (int64_t r = 0; r < n_reads; ++r) {
		int64_t na = 0;
		for (int k = 0; k < nu_of[r]; ++k) na += (uint32_t)u_of[r][k];
		out->u_off[r + 1] = out->u_off[r] + nu_of[r];
		out->a_off[r + 1] = out->a_off[r] + na;
	}
	out->u = (uint64_t*)malloc((size_t)(out->u_off[n_reads] + 1) * 8);
	out->a = (mm2gb_anchor_t*)malloc((size_t)(out->a_off[n_reads] + 1) * 16);
	std::atomic<int64_t> next(0);
	auto gather = [&]() {
		for (;;) {
			const int64_t r0 = next.fetch_add(64), r1 = std::min(n_reads, r0 + 64);
			if (r0 >= n_reads) break;
			for (int64_t r = r0; r < r1; ++r) {
				if (nu_of[r]) {
					memcpy(out->u + out->u_off[r], u_of[r], (size_t)nu_of[r] * 8);
					memcpy(out->a + out->a_off[r], a_of[r], (size_t)(out->a_off[r + 1] - out->a_off[r]) * 16);
				}
				free(u_of[r]); free(a_of[r]);
			}
		}
	};
	if (n_threads == 1 || n_reads < 128) gather();
	else {
		std::vector<std::thread> th;
		for (int t = 0; t < n_threads; ++t) th.emplace_back(gather);
		for (auto &t : th) t.join();
	}
	return 0;
}

} // namespace
} // namespace mm2gb

using namespace mm2gb;

extern "C" {

int mm2gb_chain_host(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                     int n_threads, mm2gb_chains_t *out, mm2gb_stats_t *stats)
{
	if (!eng) return fail("mm2gb_chain_host: null argument");
	return chain_on_engines(std::vector<mm2gb_engine_t*>(1, eng), n_reads, offsets, anchors, n_threads, out, stats);
}

void mm2gb_chains_free(mm2gb_chains_t *out)
{
	if (!out) return;
	free(out->u_off); free(out->u); free(out->a_off); free(out->a);
	memset(out, 0, sizeof(*out));
}

mm2gb_pool_t *mm2gb_pool_create(const mm2gb_config_t *cfg, const mm2gb_misc_t *misc, int n_devices, const int *devices)
{
	if (!misc) { set_error("mm2gb_pool_create: misc is required"); return nullptr; }
	const int visible = mm2gb_device_count();
	if (n_devices <= 0) { n_devices = visible; devices = nullptr; }
	if (n_devices <= 0) { set_error("mm2gb_pool_create: no device visible"); return nullptr; }
	std::unique_ptr<mm2gb_pool> pool(new mm2gb_pool());
	for (int k = 0; k < n_devices; ++k) {
		mm2gb_engine_t *e = mm2gb_engine_create(cfg, misc, devices ? devices[k] : k);
		if (!e) { mm2gb_pool_destroy(pool.release()); return nullptr; }
		pool->engines.push_back(e);
	}
	return pool.release();
}

void mm2gb_pool_destroy(mm2gb_pool_t *pool)
{
	if (!pool) return;
	for (mm2gb_engine_t *e : pool->engines) mm2gb_engine_destroy(e);
	delete pool;
}

int mm2gb_pool_size(const mm2gb_pool_t *pool) { return pool ? (int)pool->engines.size() : 0; }

int mm2gb_pool_device(const mm2gb_pool_t *pool, int k)
{
	return pool && k >= 0 && k < (int)pool->engines.size() ? pool->engines[(size_t)k]->e.device : -1;
}

int mm2gb_pool_set_misc(mm2gb_pool_t *pool, const mm2gb_misc_t *misc)
{
	if (!pool || !misc) return fail("mm2gb_pool_set_misc: null argument");
	for (mm2gb_engine_t *e : pool->engines) if (mm2gb_engine_set_misc(e, misc)) return -1;
	return 0;
}

int mm2gb_pool_score_host(mm2gb_pool_t *pool, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                          int32_t *f, int32_t *p, mm2gb_stats_t *stats, int64_t *first_read_of_device)
{
	if (!pool || !offsets || n_reads < 0) return fail("mm2gb_pool_score_host: null argument");
	if (offsets[0] != 0) return fail("mm2gb_pool_score_host: offsets[0] must be 0");
	return score_on_engines(pool->engines, n_reads, offsets, anchors, f, p, stats, first_read_of_device, nullptr);
}

int mm2gb_pool_chain_host(mm2gb_pool_t *pool, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                          int n_threads, mm2gb_chains_t *out, mm2gb_stats_t *stats)
{
	if (!pool) return fail("mm2gb_pool_chain_host: null argument");
	return chain_on_engines(pool->engines, n_reads, offsets, anchors, n_threads, out, stats);
}

} // extern "C"
