// batcher.cpp -- the batch accumulator and dispatcher above the engines: what mm_trbuf_t / mm_trbuf_is_full /
// the batch rotation of worker_for do in the reference host (map.c:23-157, 886-922, 1026-1075), as a library component any host
// can feed reads into, from any number of threads, for any number of GPUs.
//   * reads are appended to the accumulating batch of their lane until the next read would take it past max_total_n anchors or
//     max_read reads; the batch is then closed and the read starts the next one (the reference moves the overflowing read to its
//     pending batch the same way, map.c:887-920).  A single read larger than max_total_n is a batch by itself (the reference
//     cannot take it at all, SURVEY 8c probe D).
//   * min_n (gpu_config.json) is honoured as a routing rule without a CPU fallback: reads with fewer than min_n anchors go to a
//     lane of their own, so that thousands of them fill one launch instead of closing the big reads' batches early by read count.
//   * closed batches are dealt to the devices' workers (one engine, one host thread each) as they become free; a worker chains
//     its batch -- scores on the GPU, post-pass on host threads overlapped with the device, or (post_threads == 0) on the device
//     too -- and hands every read's chains to the caller's callback.  Reads never move between devices: no collective.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "engine.h"
#include "trace.h"

namespace mm2gb {
namespace {

enum { LANE_BIG = 0, LANE_SMALL = 1, N_LANES = 2 };

// The grouping rule, on its own so that it can be checked without a GPU (mm2gb_plan_batches) and is the only place it lives.
struct Grouping {
	int64_t max_total_n;
	int     max_read, min_n;
	int lane_of(int64_t n) const { return min_n > 0 && n < min_n ? LANE_SMALL : LANE_BIG; }
	// does a batch holding `count` reads / `total` anchors have to be closed before a read of n anchors is added?
	bool closes(int64_t count, int64_t total, int64_t n) const
	{
		if (count == 0) return false;
		if (max_read > 0 && count >= max_read) return true;
		return max_total_n > 0 && total + n > max_total_n;
	}
};

// Page-locked buffer that grows keeping its contents.
struct GrowPinned {
	void *ptr = nullptr;
	size_t bytes = 0, first = 0;                   // first: size of the first allocation
	int reserve_keep(size_t need, size_t used)
	{
		if (need <= bytes) return 0;
		// page-locking costs ~0.4 s per GB and a grown buffer is a new one: the first allocation is already `first` bytes (the batch limit, up to
		// 256 MB), later ones double -- growing from 1 MB by halves re-pinned and re-copied a 640 MB batch sixteen times (1 s per batch)
		size_t want = bytes ? bytes : std::max<size_t>(first, (size_t)1 << 20);
		while (want < need) want += want;
		void *fresh = nullptr;
		if (hipHostMalloc(&fresh, want, hipHostMallocDefault) != hipSuccess) {
			(void)hipGetLastError();
			want = need;
			if (hipHostMalloc(&fresh, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail("mm2gb_batcher: out of page-locked host memory"); }
		}
		if (used) memcpy(fresh, ptr, used);
		if (ptr) (void)hipHostFree(ptr);
		ptr = fresh; bytes = want;
		return 0;
	}
	void release() { if (ptr) (void)hipHostFree(ptr); ptr = nullptr; bytes = 0; }
};

struct Batch {
	GrowPinned anchors;
	std::vector<int64_t> offsets, ids;
	int lane = LANE_BIG;
	int writers = 0;                               // producers that have reserved a place in `anchors` and are still copying into it (guarded by the batcher's mutex)
	int64_t count() const { return (int64_t)ids.size(); }
	int64_t total() const { return offsets.empty() ? 0 : offsets.back(); }
	void clear() { offsets.assign(1, 0); ids.clear(); }
};

} // namespace
} // namespace mm2gb

using namespace mm2gb;

struct mm2gb_batcher {
	Grouping rule;
	mm2gb_misc_t misc;
	int post_threads = 0;
	mm2gb_read_done_fn done = nullptr;
	void *user = nullptr;
	std::vector<mm2gb_engine_t*> engines;
	std::vector<std::thread> workers;
	std::mutex mu;                                 // accumulators, queues, counters
	std::condition_variable cv_ready, cv_free, cv_idle, cv_copy;   // cv_copy: some batch's last writer has finished
	std::vector<std::unique_ptr<Batch>> all;
	std::deque<Batch*> free_list, ready;
	Batch *acc[N_LANES] = { nullptr, nullptr };
	int in_flight = 0;
	bool stop = false;
	std::string error;                             // first failure of a worker
	mm2gb_batcher_stats_t stats = {};

	Batch *take_free(std::unique_lock<std::mutex> &lk)
	{
		cv_free.wait(lk, [&] { return !free_list.empty() || !error.empty(); });
		if (free_list.empty()) return nullptr;
		Batch *b = free_list.front(); free_list.pop_front();
		b->clear();
		return b;
	}
	void close(int lane)
	{
		Batch *b = acc[lane];
		acc[lane] = nullptr;
		if (!b) return;
		if (b->count() == 0) { free_list.push_back(b); cv_free.notify_one(); return; }
		b->lane = lane;
		ready.push_back(b);
		stats.batches[lane] += 1;
		cv_ready.notify_one();
	}
	void work(int k)
	{
		mm2gb_engine_t *eng = engines[(size_t)k];
		// this worker feeds one device: it, and the post-pass threads it starts, run on the CPUs next to that device (numa.cpp) -- when there
		// are several workers to keep apart (as pool.cpp: a lone engine's worker and its post-pass threads keep the whole machine)
		if (engines.size() > 1) (void)mm2gb_pin_thread_to_device(mm2gb_engine_device(eng));
		for (;;) {
			Batch *b = nullptr;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_ready.wait(lk, [&] { return stop || !ready.empty(); });
				if (ready.empty()) return;
				b = ready.front(); ready.pop_front();
				++in_flight;
				cv_copy.wait(lk, [&] { return b->writers == 0; });   // a batch can be closed while producers are still copying their reads into it
			}
			mm2gb_chains_t out;
			int rc;
			{
				TraceRange range("mm2gb:batcher_chain_batch");
				rc = post_threads > 0 ? chain_batch_on_engine(eng, b->count(), b->offsets.data(), (const mm2gb_anchor_t*)b->anchors.ptr, post_threads, &out, nullptr)
				                      : mm2gb_chain_gpu(eng, b->count(), b->offsets.data(), (const mm2gb_anchor_t*)b->anchors.ptr, &out, nullptr);
			}
			if (rc == 0) {
				TraceRange range("mm2gb:batcher_deliver");
				for (int64_t r = 0; r < b->count(); ++r)
					done(user, b->ids[(size_t)r], (int)(out.u_off[r + 1] - out.u_off[r]), out.u + out.u_off[r], out.a_off[r + 1] - out.a_off[r], out.a + out.a_off[r]);
				mm2gb_chains_free(&out);
			}
			{
				std::lock_guard<std::mutex> g(mu);
				if (rc != 0 && error.empty()) error = mm2gb_last_error();
				stats.batches_per_engine[k < MM2GB_BATCHER_MAX_ENGINES ? k : MM2GB_BATCHER_MAX_ENGINES - 1] += 1;
				--in_flight;
				free_list.push_back(b);
			}
			cv_free.notify_one();
			cv_idle.notify_all();
		}
	}
};

extern "C" {

int64_t mm2gb_plan_batches(int64_t n_reads, const int64_t *n_anchors, int64_t max_total_n, int max_read, int min_n,
                           int32_t *batch_of_read, int32_t *lane_of_read)
{
	if (n_reads < 0 || (n_reads > 0 && !n_anchors)) return fail("mm2gb_plan_batches: null argument");
	const Grouping rule = { max_total_n, max_read, min_n };
	int64_t count[N_LANES] = { 0, 0 }, total[N_LANES] = { 0, 0 }, id[N_LANES] = { -1, -1 }, next_id = 0;
	for (int64_t r = 0; r < n_reads; ++r) {
		const int64_t n = n_anchors[r] > 0 ? n_anchors[r] : 0;
		const int lane = rule.lane_of(n);
		if (id[lane] < 0 || rule.closes(count[lane], total[lane], n)) { id[lane] = next_id++; count[lane] = 0; total[lane] = 0; }
		count[lane] += 1; total[lane] += n;
		if (batch_of_read) batch_of_read[r] = (int32_t)id[lane];
		if (lane_of_read) lane_of_read[r] = lane;
	}
	return next_id;
}

mm2gb_batcher_t *mm2gb_batcher_create(const mm2gb_config_t *cfg, const mm2gb_misc_t *misc, int n_devices, const int *devices,
                                      int post_threads, mm2gb_read_done_fn done, void *user)
{
	if (!misc || !done) { set_error("mm2gb_batcher_create: misc and the callback are required"); return nullptr; }
	mm2gb_config_t def;
	if (!cfg) { mm2gb_config_defaults(&def); cfg = &def; }
	const int visible = mm2gb_device_count();
	if (n_devices <= 0) { n_devices = visible; devices = nullptr; }
	if (n_devices <= 0) { set_error("mm2gb_batcher_create: no device visible"); return nullptr; }
	std::unique_ptr<mm2gb_batcher> b(new mm2gb_batcher());
	b->rule = Grouping{ cfg->max_total_n * (int64_t)std::max(1, cfg->score_kernel.micro_batch), cfg->max_read * std::max(1, cfg->score_kernel.micro_batch), cfg->min_n };
	b->misc = *misc; b->post_threads = post_threads; b->done = done; b->user = user;
	for (int k = 0; k < n_devices; ++k) {
		mm2gb_engine_t *e = mm2gb_engine_create(cfg, misc, devices ? devices[k] : k);
		if (!e) { for (mm2gb_engine_t *x : b->engines) mm2gb_engine_destroy(x); return nullptr; }
		b->engines.push_back(e);
	}
	// two batches per worker (one on the device, one being filled / waiting) and one per lane on top
	for (int k = 0; k < 2 * n_devices + N_LANES; ++k) {
		b->all.emplace_back(new Batch());
		b->all.back()->anchors.first = (size_t)std::min<int64_t>(b->rule.max_total_n > 0 ? b->rule.max_total_n : (int64_t)1 << 24, (int64_t)1 << 24) * 16;
		b->free_list.push_back(b->all.back().get());
	}
	for (int k = 0; k < n_devices; ++k) b->workers.emplace_back([p = b.get(), k] { p->work(k); });
	return b.release();
}

int mm2gb_batcher_add(mm2gb_batcher_t *b, int64_t read_id, const mm2gb_anchor_t *a, int64_t n)
{
	if (!b || n < 0 || (n > 0 && !a)) return fail("mm2gb_batcher_add: bad argument");
	std::unique_lock<std::mutex> lk(b->mu);
	if (!b->error.empty()) return fail(b->error);
	const int lane = b->rule.lane_of(n);
	for (;;) {
		if (!b->error.empty()) return fail(b->error);
		Batch *cur = b->acc[lane];
		if (cur && b->rule.closes(cur->count(), cur->total(), n)) { b->close(lane); continue; }
		if (cur) break;
		Batch *fresh = b->take_free(lk);                  // may wait (back-pressure on the producers) and lets other producers in meanwhile
		if (!fresh) return fail(b->error);
		if (b->acc[lane]) { b->free_list.push_back(fresh); b->cv_free.notify_one(); continue; }   // another producer opened the lane's batch first
		b->acc[lane] = fresh;
	}
	for (;;) {                                                // room for this read: the buffer grows only while no producer is copying into it
		Batch *cur = b->acc[lane];
		if (!cur || b->rule.closes(cur->count(), cur->total(), n)) { lk.unlock(); return mm2gb_batcher_add(b, read_id, a, n); }   // closed under our feet while we waited: start over
		const size_t need = (size_t)(cur->total() + n) * 16;
		if (need <= cur->anchors.bytes) break;
		if (cur->writers > 0) { b->cv_copy.wait(lk); continue; }
		if (cur->anchors.reserve_keep(need, (size_t)cur->total() * 16)) return -1;
		break;
	}
	// A place in the batch is reserved under the lock; the read is copied into it OUTSIDE the lock, so producers copy side by side
	// (with the copy inside, three producer threads fed the batcher no faster than one).  The buffer may only move while nobody copies.
	Batch &acc = *b->acc[lane];
	const int64_t at = acc.total();
	acc.offsets.push_back(at + n);
	acc.ids.push_back(read_id);
	b->stats.reads += 1; b->stats.anchors += n; b->stats.reads_per_lane[lane] += 1;
	if (n == 0) return 0;
	mm2gb_anchor_t *dst = (mm2gb_anchor_t*)acc.anchors.ptr + at;
	++acc.writers;
	lk.unlock();
	memcpy(dst, a, (size_t)n * 16);
	lk.lock();
	if (--acc.writers == 0) b->cv_copy.notify_all();
	return 0;
}

// The reads of a packed batch (offsets[n_reads + 1] into anchors), added one at a time by n_producers threads that deal the reads among
// themselves: what a multi-threaded host's seeding threads do, from one call (read r gets the id first_id + r).
int mm2gb_batcher_feed(mm2gb_batcher_t *b, int64_t n_reads, int64_t first_id, const int64_t *offsets, const mm2gb_anchor_t *anchors, int n_producers)
{
	if (!b || n_reads < 0 || !offsets) return fail("mm2gb_batcher_feed: bad argument");
	std::atomic<int64_t> next(0);
	std::atomic<int> bad(0);
	std::string why;
	std::mutex why_lock;
	auto work = [&]() {
		for (;;) {
			const int64_t r = next.fetch_add(1);
			if (r >= n_reads || bad.load()) break;
			if (mm2gb_batcher_add(b, first_id + r, anchors + offsets[r], offsets[r + 1] - offsets[r])) {
				std::lock_guard<std::mutex> g(why_lock);
				if (!bad.exchange(1)) why = mm2gb_last_error();
			}
		}
	};
	std::vector<std::thread> pool;
	for (int t = 1; t < std::max(1, n_producers); ++t) pool.emplace_back(work);
	work();
	for (auto &th : pool) th.join();
	return bad.load() ? fail(why) : 0;
}

int mm2gb_batcher_flush(mm2gb_batcher_t *b)
{
	if (!b) return fail("mm2gb_batcher_flush: null argument");
	std::unique_lock<std::mutex> lk(b->mu);
	for (int lane = 0; lane < N_LANES; ++lane) b->close(lane);
	b->cv_idle.wait(lk, [&] { return (b->ready.empty() && b->in_flight == 0) || !b->error.empty(); });
	if (!b->error.empty()) return fail(b->error);
	return 0;
}

int mm2gb_batcher_stats(mm2gb_batcher_t *b, mm2gb_batcher_stats_t *out)
{
	if (!b || !out) return fail("mm2gb_batcher_stats: null argument");
	std::lock_guard<std::mutex> g(b->mu);
	*out = b->stats;
	out->n_engines = (int)b->engines.size();
	return 0;
}

void mm2gb_batcher_destroy(mm2gb_batcher_t *b)
{
	if (!b) return;
	(void)mm2gb_batcher_flush(b);
	{ std::lock_guard<std::mutex> g(b->mu); b->stop = true; }
	b->cv_ready.notify_all();
	for (auto &t : b->workers) t.join();
	for (auto &bt : b->all) bt->anchors.release();
	for (mm2gb_engine_t *e : b->engines) mm2gb_engine_destroy(e);
	delete b;
}

} // extern "C"
