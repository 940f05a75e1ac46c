// pool.cpp -- whole batches on host buffers: one or several devices score, a pool of host threads backtracks.
//   * mm2gb_chain_host      : one engine (one device)
//   * mm2gb_pool_*          : several engines in one process, reads dealt to devices as contiguous runs balanced by anchor
//                             count; no data moves between devices (reads are independent, no collective)
// Both run the host post-pass (host_chain.cpp) on reads as soon as the slice that holds them is back from the device, while
// later slices are still being copied and scored, so the devices and the host threads work at the same time.
// Replaces plchain_cal_score_async + plchain_post_gpu_helper for callers that own whole batches (plchain.cu:201-464).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include "engine.h"
#include "host_chain.h"

struct mm2gb_pool {
	std::vector<mm2gb_engine_t*> engines;     // owned
};

namespace mm2gb {
namespace {

// A run of reads whose scores are in host memory: f[0] / p[0] belong to the first anchor of read `shift_read`.
struct ScoredRun {
	int64_t r0, r1;
	const int32_t *f, *p;
	int64_t shift;          // offsets[shift_read]
};

// Scored runs, consumed a few reads at a time by the post-pass threads.
class ReadyReads {
public:
	explicit ReadyReads(const int64_t *offsets) : off_(offsets) {}
	void push(const ScoredRun &run)
	{
		if (run.r1 <= run.r0) return;
		{ std::lock_guard<std::mutex> g(mu_); runs_.push_back(run); }
		cv_.notify_all();
	}
	void close()
	{
		{ std::lock_guard<std::mutex> g(mu_); closed_ = true; }
		cv_.notify_all();
	}
	// next group of reads: about GRAIN anchors, so that tiny reads do not pay one lock each
	bool pop(ScoredRun &part)
	{
		std::unique_lock<std::mutex> g(mu_);
		cv_.wait(g, [&] { return closed_ || !runs_.empty(); });
		if (runs_.empty()) return false;
		ScoredRun &run = runs_.front();
		part = run;
		part.r1 = part.r0 + 1;
		while (part.r1 < run.r1 && off_[part.r1] - off_[part.r0] < GRAIN) ++part.r1;
		run.r0 = part.r1;
		if (run.r0 >= run.r1) runs_.pop_front();
		return true;
	}
private:
	static constexpr int64_t GRAIN = 1 << 16;
	const int64_t *off_;
	std::mutex mu_;
	std::condition_variable cv_;
	std::deque<ScoredRun> runs_;
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

// Scores on every engine (one host thread each when there are several).  f / p given: results go there, indexed like
// anchors.  f == NULL: results stay in each engine's page-locked result buffers (true asynchronous D2H, no page faults on
// fresh memory, reused by the next call) and are only announced through `ready`.
int score_on_engines(const std::vector<mm2gb_engine_t*> &engines, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                     int32_t *f, int32_t *p, mm2gb_stats_t *stats, int64_t *first_read_of_device, ReadyReads *ready)
{
	const int n_dev = (int)engines.size();
	const std::vector<int64_t> first = deal_reads(n_reads, offsets, n_dev);
	if (first_read_of_device) memcpy(first_read_of_device, first.data(), ((size_t)n_dev + 1) * sizeof(int64_t));
	std::vector<int> rc((size_t)n_dev, 0);
	std::vector<std::string> err((size_t)n_dev);
	auto run = [&](int d) {
		Engine &e = engines[d]->e;
		// a device's own host thread sits on the CPUs next to it (numa.cpp; nothing happens on a one-node box or when the caller runs it itself)
		if (n_dev > 1) (void)mm2gb_pin_thread_to_device(e.device);
		const int64_t r0 = first[d], r1 = first[d + 1], shift = offsets[r0], share = offsets[r1] - shift;
		int32_t *fd = f ? f + shift : nullptr, *pd = p ? p + shift : nullptr;
		if (!f) {
			if (hipSetDevice(e.device) != hipSuccess || e.h_res_f.ensure((size_t)std::max<int64_t>(share, 1) * 4) || e.h_res_p.ensure((size_t)std::max<int64_t>(share, 1) * 4)) {
				rc[d] = -1; err[d] = mm2gb_last_error(); return;
			}
			fd = (int32_t*)e.h_res_f.ptr; pd = (int32_t*)e.h_res_p.ptr;
		}
		std::function<void(int64_t, int64_t)> forward = [&, r0, fd, pd, shift](int64_t a, int64_t b) { ready->push(ScoredRun{ r0 + a, r0 + b, fd, pd, shift }); };
		rc[d] = e.score_host(r1 - r0, offsets + r0, anchors, fd, pd, ready ? &forward : nullptr);
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

// What a post-pass thread keeps of the reads it handled: chain lists and, for every anchor kept, its index in the read.
struct PostSlab {
	std::vector<uint64_t> u;
	std::vector<int32_t> idx;
};
struct ReadChains { int32_t slab = 0, n_u = 0; int64_t n_kept = 0; size_t u_at = 0, idx_at = 0; };

template <typename F>
void on_threads(int n_threads, F fn)
{
	if (n_threads <= 1) { fn(0); return; }
	std::vector<std::thread> th;
	for (int t = 0; t < n_threads; ++t) th.emplace_back(fn, t);
	for (auto &t : th) t.join();
}

int chain_on_engines(const std::vector<mm2gb_engine_t*> &engines, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                     int n_threads, mm2gb_chains_t *out, mm2gb_stats_t *stats)
{
	if (!out || !offsets || n_reads < 0 || engines.empty()) return fail("mm2gb_chain_host: null argument");
	memset(out, 0, sizeof(*out));
	if (offsets[0] != 0) return fail("mm2gb_chain_host: offsets[0] must be 0");
	if (n_threads < 1) n_threads = 1;
	const mm2gb_misc_t misc = engines[0]->e.misc;
	ReadyReads ready(offsets);
	const char *dbg_env = getenv("MM2GB_DEBUG_PHASES");
	const bool dbg = dbg_env && *dbg_env && *dbg_env != '0';
	const auto t0 = std::chrono::steady_clock::now();
	auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
	// Post-pass: chains of each read as soon as its slice is back.  Only the order is kept (4 bytes per anchor kept); the
	// anchors themselves are copied once, from the caller's array straight to their final place, when every size is known.
	std::vector<PostSlab> slabs((size_t)n_threads);
	std::vector<ReadChains> of((size_t)n_reads);
	std::vector<std::thread> post;
	for (int t = 0; t < n_threads; ++t)
		post.emplace_back([&, t]() {
			BacktrackScratch ws;
			PostSlab &slab = slabs[(size_t)t];
			ScoredRun part;
			while (ready.pop(part))
				for (int64_t r = part.r0; r < part.r1; ++r) {
					ReadChains &rc = of[(size_t)r];
					const int64_t at = offsets[r] - part.shift;
					rc.n_u = backtrack_order(misc, offsets[r + 1] - offsets[r], anchors + offsets[r], part.f + at, part.p + at, ws, &rc.n_kept);
					if (rc.n_u == 0) continue;
					rc.slab = t; rc.u_at = slab.u.size(); rc.idx_at = slab.idx.size();
					slab.u.resize(rc.u_at + (size_t)rc.n_u);
					slab.idx.resize(rc.idx_at + (size_t)rc.n_kept);
					emit_chain_list(ws, slab.u.data() + rc.u_at);
					emit_anchor_order(ws, slab.idx.data() + rc.idx_at);
				}
		});
	const int rc = score_on_engines(engines, n_reads, offsets, anchors, nullptr, nullptr, stats, nullptr, &ready);
	const double t_scored = since();
	ready.close();
	for (auto &t : post) t.join();
	const double t_post = since();
	if (rc) return -1;
	out->u_off = (int64_t*)malloc((size_t)(n_reads + 1) * 8);
	out->a_off = (int64_t*)malloc((size_t)(n_reads + 1) * 8);
	out->u_off[0] = out->a_off[0] = 0;
	for (int64_t r = 0; r < n_reads; ++r) {
		out->u_off[r + 1] = out->u_off[r] + of[(size_t)r].n_u;
		out->a_off[r + 1] = out->a_off[r] + of[(size_t)r].n_kept;
	}
	out->u = (uint64_t*)malloc((size_t)(out->u_off[n_reads] + 1) * 8);
	out->a = (mm2gb_anchor_t*)result_alloc((size_t)(out->a_off[n_reads] + 1) * 16);   // (huge pages if it is large)
	std::atomic<int64_t> next(0);
	on_threads(n_reads < 128 ? 1 : n_threads, [&](int) {
		for (;;) {
			const int64_t r0 = next.fetch_add(16), r1 = std::min(n_reads, r0 + 16);
			if (r0 >= n_reads) break;
			for (int64_t r = r0; r < r1; ++r) {
				const ReadChains &rc = of[(size_t)r];
				if (rc.n_u == 0) continue;
				const PostSlab &slab = slabs[(size_t)rc.slab];
				memcpy(out->u + out->u_off[r], slab.u.data() + rc.u_at, (size_t)rc.n_u * 8);
				const mm2gb_anchor_t *src = anchors + offsets[r];
				const int32_t *idx = slab.idx.data() + rc.idx_at;
				mm2gb_anchor_t *dst = out->a + out->a_off[r];
				for (int64_t j = 0; j < rc.n_kept; ++j) dst[j] = src[idx[j]];
			}
		}
	});
	if (dbg) fprintf(stderr, "[mm2gb chain_host] scores back %.1f ms | post-pass done %.1f ms | chains gathered %.1f ms (%lld chains, %lld anchors kept)\n",
	                 t_scored, t_post, since(), (long long)out->u_off[n_reads], (long long)out->a_off[n_reads]);
	return 0;
}

} // namespace

int chain_batch_on_engine(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, int n_threads, mm2gb_chains_t *out, mm2gb_stats_t *stats)
{
	return chain_on_engines(std::vector<mm2gb_engine_t*>(1, eng), n_reads, offsets, anchors, n_threads, out, stats);
}

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
	free(out->u_off); result_release(out->u); free(out->a_off); result_release(out->a);
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
