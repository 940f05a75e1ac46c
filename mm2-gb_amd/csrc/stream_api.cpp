// stream_api.cpp -- batch-level host logic above the engine:
//   * mm2gb_lchain_dp      : single-read entry with the mg_lchain_dp signature (mmpriv.h:84-85)
//   * init/chain/finish/free_stream_gpu : the reference's drop-in boundary (gpu/plutils.h:98-104; plchain.cu:466-561),
//     same deferred hand-back protocol (launch batch k, return batch k-1 finished), one engine per stream/thread id.
#include <hip/hip_runtime.h>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/mm2gb_plutils.h"
#include "engine.h"
#include "host_chain.h"
#include "rechain_ahead.h"
#include "trace.h"
#include <sched.h>
#include <map>

// ---- host callbacks (map.c:393, map.c:428); weak so the library also loads without a minimap2 host ----
extern "C" {
mm2gb_Misc build_misc(const struct mm_idx_s *mi, const struct mm_mapopt_s *opt, const int64_t qlen_sum, const int n_seg) __attribute__((weak));
void post_chaining_helper(const struct mm_idx_s *mi, const struct mm_mapopt_s *opt, mm2gb_chain_read_t *read, mm2gb_Misc misc, void *km) __attribute__((weak));
}

// Layout of the records shared with the host, as a C compiler lays out gpu/plutils.h:19-73 on x86-64 (checked against the
// reference header with offsetof in the dev container; tests/test_gpu_stream_api.py repeats the check from Python).
static_assert(sizeof(mm2gb_Misc) == 44, "Misc layout");
static_assert(sizeof(mm2gb_seq_meta_t) == 232 && offsetof(mm2gb_seq_meta_t, name) == 12 && offsetof(mm2gb_seq_meta_t, qlen_sum) == 224, "mm_seq_meta_t layout");
static_assert(sizeof(mm2gb_chain_read_t) == 312 && offsetof(mm2gb_chain_read_t, a) == 280 && offsetof(mm2gb_chain_read_t, n) == 288 &&
              offsetof(mm2gb_chain_read_t, u) == 296 && offsetof(mm2gb_chain_read_t, n_u) == 304, "chain_read_t layout");

namespace mm2gb {

// Run fn(r, scratch) for r in [0, n) on n_threads threads, dynamic dealing.
template <typename F>
static void parallel_reads(int64_t n, int n_threads, F fn)
{
	if (n_threads < 1) n_threads = 1;
	if (n_threads == 1 || n < 2) {
		BacktrackScratch ws;
		for (int64_t r = 0; r < n; ++r) fn(r, ws);
		return;
	}
	std::atomic<int64_t> next(0);
	std::vector<std::thread> pool;
	for (int t = 0; t < n_threads; ++t)
		pool.emplace_back([&]() {
			BacktrackScratch ws;
			for (;;) {
				const int64_t r = next.fetch_add(1);
				if (r >= n) break;
				fn(r, ws);
			}
		});
	for (auto &th : pool) th.join();
}

// ---------------------------------------------------------------------------------------------------------------
// drop-in boundary state: one slot per stream / host thread id
// ---------------------------------------------------------------------------------------------------------------
// Host side of one batch in flight: page-locked staging and the reads it belongs to.
struct HostStage {
	std::vector<int64_t> goff;             // offsets of every read in h_raw / h_f / h_p, size n_read + 1
	mm2gb_chain_read_t *reads = nullptr;   // owned by the host
	mm2gb_misc_t misc = {};                // the parameters this batch was LAUNCHED with: its post-pass must use the same ones
	const struct mm_mapopt_s *opt = nullptr;   // the host's options at launch (for re-chaining ahead; the host keeps one record for the whole run)
	int  n_read = 0;
	bool busy = false;
	bool device_post = false;              // this batch's backtrack + compaction run as kernels; its chains wait in engine post set `post_set`
	int  post_set = 0;
	hipEvent_t done = nullptr;             // all f/p of this batch are back in h_f / h_p
	// ---- what the stream's finisher thread makes of the batch while the host thread is back at seeding the next one (finish_compute):
	// chains per read in libc memory (host post-pass) or one block (device post-pass), and the batch's re-chaining answered ahead ----
	bool computed = false;                 // guarded by StreamSlot::mu
	std::string err;                       // non-empty: the finisher failed (reported by the host thread that picks the batch up)
	std::vector<uint64_t*> u_of;
	std::vector<mm2gb_anchor_t*> a_of;
	std::vector<int> nu_of;
	mm2gb_chains_t ch = {};
	RechainAhead ahead;
	bool have_ahead = false;
	double ms_wait = 0, ms_post = 0, ms_ahead = 0, t_computed_ms = 0;
};

// One stream / host thread id.  Two stages alternate, and each stream has a FINISHER thread: the host thread packs and launches batch k and
// goes back to its own work (seeding batch k+1); the finisher waits for k's scores, turns them into chains (host post-pass threads, or fetches
// the device post-pass's), answers the batch's re-chaining ahead of the host's callback (rechain_ahead.cpp) -- so that when the host thread
// comes back with batch k+1 everything it needs of batch k is there, and what is left is the hand-over into its kalloc arena and its own callback.
// The engine is used by one thread at a time: the host thread touches it only while no batch of the slot is with the finisher.
struct StreamSlot {
	Engine eng;
	HostStage stage[2];
	int  cur = 0;                          // stage of the batch in flight
	bool live = false;
	// Page-locked staging, ONE set for both stages: a batch is launched only after the finisher is through with the previous one (its anchors
	// copied in, its scores turned into chains), so the previous batch needs none of it any more -- half the memory to pin (~0.4 s per GB, and
	// sixteen streams pin at the same moment) and to give back.
	PinnedBuf h_raw, h_f, h_p, h_off;      // h_off: per-micro-batch offsets, each from 0
	// re-chaining ahead: an engine of its own, made when first needed (a device re-chaining call owns its engine's streams and arenas)
	mm2gb_engine_t *rmq_eng = nullptr;
	std::mutex rmq_mu;                     // whoever makes rmq_eng (the maker thread ahead of time, else the finisher at its first need)
	std::thread finisher;
	std::mutex mu;
	std::condition_variable cv;
	std::vector<HostStage*> jobs;          // batches waiting for the finisher (at most two)
	bool stop = false;
};

static struct {
	std::vector<StreamSlot*> slots;
	mm2gb_config_t cfg;
	mm2gb_misc_t   misc;
	int  post_threads = 1;
	bool post_on_device = false;           // MM2GB_POST=gpu (or MM2GB_POST_THREADS=0): no host post-pass threads at all
	bool ready = false;
	bool debug = false;                    // MM2GB_DEBUG_PHASES: where a batch's host time goes, on stderr
	int  ahead_threads = 1;                // host threads of a batch's re-chaining ahead (beside the device)
	bool rechain_ahead = false;            // answer a batch's mg_lchain_rmq calls before the host's callback asks (host linked with --wrap, or MM2GB_PRECHAIN=1)
	// The streams' engines are made by a thread of their own while the host reads and seeds its first batch (init_stream_gpu returns once
	// the sizes are known): a slot is `live` when its engine and its finisher are there; whoever needs it waits for that (slot_for).
	std::thread *maker = nullptr;          // (on the heap: a process that ends without free_stream_gpu must not run a joinable thread's destructor)
	std::mutex mk_mu;
	std::condition_variable mk_cv;
	std::string mk_err;
} g_streams;

static double now_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static inline int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int64_t g_t_init_ns = 0;

[[noreturn]] static void die(const std::string &msg)
{
	fprintf(stderr, "[Error] %s\n", msg.c_str());   // the reference's style for fatal configuration problems (plmem.cu:390-412)
	exit(1);
}
#define MM2GB_HIP_DIE(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) die(std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

static int devices_for_streams(std::vector<int> &out)
{
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return fail("no HIP device visible");
	// MM2GB_DEVICES=0,1,2 restricts / orders the devices streams are dealt to; default: all visible devices
	const char *env = getenv("MM2GB_DEVICES");
	out.clear();
	if (env && *env) {
		for (const char *p = env; *p;) {
			char *end; long d = strtol(p, &end, 10);
			if (end == p) break;
			if (d < 0 || d >= n_dev) return fail("MM2GB_DEVICES names device " + std::to_string(d) + " but only " + std::to_string(n_dev) + " are visible");
			out.push_back((int)d);
			p = *end == ',' ? end + 1 : end;
		}
	}
	if (out.empty()) for (int d = 0; d < n_dev; ++d) out.push_back(d);
	return 0;
}

static std::atomic<int64_t> g_reads_ahead(0), g_ns_ahead(0);     // reads whose re-chaining was answered ahead, and the finishers' time in it

// First half of finishing the batch held by `st`, on the stream's finisher thread: wait for its scores, make the chains of every read
// (libc memory), answer the batch's re-chaining ahead.  Touches nothing of the host's (kalloc arenas are not thread-safe).
static int finish_compute(StreamSlot &slot, HostStage &st)
{
	TraceRange range("mm2gb:finish_batch");
	MM2GB_HIP(hipSetDevice(slot.eng.device));
	const double t0 = now_ms();
	const int n_read = st.n_read;
	const int64_t *off = st.goff.data();
	mm2gb_chain_read_t *reads = st.reads;
	std::vector<RechainRead> view;
	const bool want_ahead = g_streams.rechain_ahead && st.opt && n_read > 0 && rechain_ahead_is_exact(*(const mm2gb_mapopt_head_t*)st.opt);
	if (st.device_post) {
		// the chains were made on the device (post_kernels.hip): fetch them (exact sizes)
		if (slot.eng.fetch_chains(st.post_set, n_read, &st.ch)) return -1;
		st.ms_wait = now_ms() - t0; st.ms_post = 0;
		if (want_ahead) {
			view.resize((size_t)n_read);
			for (int r = 0; r < n_read; ++r)
				view[(size_t)r] = RechainRead{ st.ch.a + st.ch.a_off[r], st.ch.u + st.ch.u_off[r], (int)(st.ch.u_off[r + 1] - st.ch.u_off[r]), reads[r].n_seg, reads[r].seq.qlen_sum };
		}
	} else {
		{ TraceRange wait("mm2gb:wait_scores"); MM2GB_HIP(hipEventSynchronize(st.done)); }
		const double t_wait = now_ms();
		const int32_t *f = (const int32_t*)slot.h_f.ptr, *p = (const int32_t*)slot.h_p.ptr;
		const mm2gb_misc_t misc = st.misc;                    // not the engine's current ones: a newer batch may already be in flight
		HostAlloc libc_mem;                                   // worker threads allocate from libc only
		st.u_of.assign((size_t)n_read, nullptr); st.a_of.assign((size_t)n_read, nullptr); st.nu_of.assign((size_t)n_read, 0);
		{
			TraceRange post("mm2gb:backtrack_compact");
			parallel_reads(n_read, g_streams.post_threads, [&](int64_t r, BacktrackScratch &ws) {
				st.nu_of[r] = backtrack_compact(misc, reads[r].n, reads[r].a, f + off[r], p + off[r], libc_mem, ws, &st.u_of[r], &st.a_of[r]);
			});
		}
		st.ms_wait = t_wait - t0; st.ms_post = now_ms() - t_wait;
		if (want_ahead) {
			view.resize((size_t)n_read);
			for (int r = 0; r < n_read; ++r) view[(size_t)r] = RechainRead{ st.a_of[r], st.u_of[r], st.nu_of[r], reads[r].n_seg, reads[r].seq.qlen_sum };
		}
	}
	st.have_ahead = false;
	st.ms_ahead = 0;
	if (want_ahead) {
		const int64_t ta = now_ns();
		{
			// (normally there already: init_stream_gpu's maker makes the re-chaining engines too, beside the host's first batch)
			std::lock_guard<std::mutex> lk(slot.rmq_mu);
			if (!slot.rmq_eng) {
				slot.rmq_eng = mm2gb_engine_create(&g_streams.cfg, &st.misc, slot.eng.device);
				if (!slot.rmq_eng) return -1;
				slot.rmq_eng->e.rmq_calibrate = true;          // the host thread waits for this finisher: its CPU, and the call's threads, are the deal's to use
			}
		}
		if (rechain_ahead(slot.rmq_eng, *(const mm2gb_mapopt_head_t*)st.opt, st.misc, view.data(), n_read, g_streams.ahead_threads, st.ahead)) return -1;
		st.have_ahead = st.ahead.n_ahead() > 0;
		g_reads_ahead.fetch_add(st.ahead.n_ahead());
		g_ns_ahead.fetch_add(now_ns() - ta);
		st.ms_ahead = (now_ns() - ta) * 1e-6;
		if (g_streams.debug)
			fprintf(stderr, "[mm2gb stream] re-chaining ahead (%.3f -> %.3f s since init): %lld of %d reads, %lld anchors | select %.1f ms | copy + sort %.1f ms | one call %.1f ms (device %lld reads %.1f ms || host %lld reads %.1f ms, %lld tied redone %.1f ms)\n",
			        (ta - g_t_init_ns) * 1e-9, (now_ns() - g_t_init_ns) * 1e-9, (long long)st.ahead.n_ahead(), n_read, (long long)(st.have_ahead ? st.ahead.off.back() : 0), st.ahead.s_select * 1e3, st.ahead.s_sort * 1e3, st.ahead.s_call * 1e3,
			        (long long)st.ahead.deal.n_device, st.ahead.deal.device_s * 1e3, (long long)st.ahead.deal.n_host_cost, st.ahead.deal.host_s * 1e3, (long long)st.ahead.deal.n_host_tie, st.ahead.deal.tie_s * 1e3);
	}
	st.t_computed_ms = now_ms();
	return 0;
}

// The stream's finisher thread: batches in launch order.
static void finisher_main(StreamSlot *slot)
{
	for (;;) {
		HostStage *st = nullptr;
		{
			std::unique_lock<std::mutex> lk(slot->mu);
			slot->cv.wait(lk, [&] { return slot->stop || !slot->jobs.empty(); });
			if (slot->jobs.empty()) return;
			st = slot->jobs.front();
		}
		std::string err;
		try { if (finish_compute(*slot, *st)) err = mm2gb_last_error(); }
		catch (const std::exception &ex) { err = std::string("finishing a batch: ") + ex.what(); }
		{
			std::lock_guard<std::mutex> lk(slot->mu);
			st->err = err;
			st->computed = true;
			slot->jobs.erase(slot->jobs.begin());
		}
		slot->cv.notify_all();
	}
}

// Wait until the finisher is through with `st` (the host thread may then use the slot's engine again).
static void wait_computed(StreamSlot &slot, HostStage &st)
{
	if (!st.busy) return;
	std::unique_lock<std::mutex> lk(slot.mu);
	slot.cv.wait(lk, [&] { return st.computed; });
	if (!st.err.empty()) die(st.err);
}

// Second half, on the CALLING thread: results move into the host's kalloc arena (not thread-safe, so here), the reads go back.
static int finish_handover(StreamSlot &slot, HostStage &st, void *km, mm2gb_chain_read_t **reads_out, int *n_out, mm2gb_misc_t *misc_out)
{
	*reads_out = nullptr; *n_out = 0;
	if (!st.busy) return 0;
	(void)slot;
	const double t0 = now_ms();
	mm2gb_chain_read_t *reads = st.reads;
	const int n_read = st.n_read;
	const int64_t *off = st.goff.data();
	if (misc_out) *misc_out = st.misc;
	HostAlloc host_mem; host_mem.km = km; host_mem.use_kalloc = host_kalloc_present();
	if (st.device_post) {
		mm2gb_chains_t &ch = st.ch;
		for (int r = 0; r < n_read; ++r) {
			mm2gb_chain_read_t &rd = reads[r];
			const int n_u = (int)(ch.u_off[r + 1] - ch.u_off[r]);
			const int64_t n_a = ch.a_off[r + 1] - ch.a_off[r];
			uint64_t *u = nullptr; mm2gb_anchor_t *a_new = nullptr;
			if (n_u > 0) {
				u = (uint64_t*)host_mem.alloc((size_t)n_u * 8);
				a_new = (mm2gb_anchor_t*)host_mem.alloc((size_t)n_a * 16);
				memcpy(u, ch.u + ch.u_off[r], (size_t)n_u * 8); memcpy(a_new, ch.a + ch.a_off[r], (size_t)n_a * 16);
			}
			host_mem.release(rd.a);
			rd.a = a_new; rd.u = u; rd.n_u = n_u;
		}
		mm2gb_chains_free(&ch);
	} else {
		for (int r = 0; r < n_read; ++r) {
			mm2gb_chain_read_t &rd = reads[r];
			uint64_t *u = st.u_of[r]; mm2gb_anchor_t *a_new = st.a_of[r];
			if (host_mem.use_kalloc && st.nu_of[r] > 0) {
				size_t na = 0;
				for (int k = 0; k < st.nu_of[r]; ++k) na += (uint32_t)u[k];
				uint64_t *ku = (uint64_t*)host_mem.alloc((size_t)st.nu_of[r] * 8);
				mm2gb_anchor_t *ka = (mm2gb_anchor_t*)host_mem.alloc(na * 16);
				memcpy(ku, u, (size_t)st.nu_of[r] * 8); memcpy(ka, a_new, na * 16);
				free(u); free(a_new);
				u = ku; a_new = ka;
			}
			host_mem.release(rd.a);                 // compact_a frees the oversized input (lchain.c:108-109, plchain.cu:135)
			rd.a = a_new; rd.u = u; rd.n_u = st.nu_of[r];   // a = 0 when nothing chained (plchain.cu:134-137)
		}
		st.u_of.clear(); st.a_of.clear(); st.nu_of.clear();
	}
	if (g_streams.debug)
		fprintf(stderr, "[mm2gb stream] finish%s: %d reads, %lld anchors | finisher: wait %.2f ms, post-pass %.2f ms, re-chaining ahead %.2f ms, ready %.2f ms before it was asked for | hand-over %.2f ms\n",
		        st.device_post ? " (device post-pass)" : "", n_read, (long long)off[n_read], st.ms_wait, st.ms_post, st.ms_ahead, t0 - st.t_computed_ms, now_ms() - t0);
	st.busy = false; st.reads = nullptr; st.n_read = 0; st.device_post = false;
	*reads_out = reads; *n_out = n_read;
	return 0;
}

// Launch `reads` through `st`: pack anchors into pinned memory, enqueue one or more micro-batches, return at once.
static int launch_stage(StreamSlot &slot, HostStage &st, mm2gb_chain_read_t *reads, int n_read, const struct mm_mapopt_s *opt)
{
	int64_t total = 0;
	for (int r = 0; r < n_read; ++r) total += reads[r].n > 0 ? reads[r].n : 0;
	const double t0 = now_ms();
	TraceRange range("mm2gb:launch_batch");
	MM2GB_HIP(hipSetDevice(slot.eng.device));
	st.misc = slot.eng.misc;
	slot.eng.io_seq = 0;                                  // nothing of an earlier batch is in flight (wait_computed): start with staging set 0 again, no second set to allocate
	if (!st.done) MM2GB_HIP(hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
	if (slot.h_raw.ensure((size_t)(total + 1) * 16)) return -1;
	st.goff.resize((size_t)n_read + 1);
	int64_t *off = st.goff.data();
	mm2gb_anchor_t *raw = (mm2gb_anchor_t*)slot.h_raw.ptr;
	off[0] = 0;
	for (int r = 0; r < n_read; ++r) off[r + 1] = off[r] + (reads[r].n > 0 ? reads[r].n : 0);
	// pack the reads' anchor arrays into the pinned staging buffer (MM2GB_POST_THREADS host threads)
	{
		TraceRange pack("mm2gb:pack_anchors");
		parallel_reads(n_read, g_streams.post_threads, [&](int64_t r, BacktrackScratch &) {
			const int64_t n = off[r + 1] - off[r];
			if (n) memcpy(raw + off[r], reads[r].a, (size_t)n * 16);
		});
	}
	const double t_pack = now_ms();
	// micro-batches: greedy split so each holds at most max_total_n anchors (plchain.cu:356-366); unlike the reference
	// nothing is ever sent back to the CPU -- a batch simply takes as many micro-batches as it needs
	const int64_t cap = g_streams.cfg.max_total_n > 0 ? g_streams.cfg.max_total_n : total;
	std::vector<int64_t> mb_first(1, 0);
	int64_t acc = 0;
	for (int r = 0; r < n_read; ++r) {
		const int64_t n = off[r + 1] - off[r];
		if (acc > 0 && acc + n > cap) { mb_first.push_back(r); acc = 0; }
		acc += n;
	}
	mb_first.push_back(n_read);
	// every micro-batch needs offsets that start at 0: build them after the global ones
	const size_t n_mb = mb_first.size() - 1;
	if (slot.h_off.ensure(((size_t)n_read + n_mb + 1) * 8)) return -1;
	int64_t *local_off = (int64_t*)slot.h_off.ptr;
	size_t w = 0;
	for (size_t m = 0; m < n_mb; ++m) {
		const int64_t r0 = mb_first[m], r1 = mb_first[m + 1];
		const size_t base = w;
		for (int64_t r = r0; r <= r1; ++r) local_off[w++] = off[r] - off[r0];
		const int64_t n = off[r1] - off[r0];
		TraceRange mb("mm2gb:enqueue_microbatch");
		if (g_streams.post_on_device && n_mb == 1) {
			// scores never leave the device: post kernels follow the score kernel, the chains are fetched when the batch is finished
			st.post_set = slot.cur ^ 1;                              // == the index of this stage: stage k's results live in post set k
			if (slot.eng.enqueue_host_chains(r1 - r0, local_off + base, raw + off[r0], n, st.post_set)) return -1;
			st.device_post = true;
			continue;
		}
		if (slot.h_f.ensure((size_t)(total + 1) * 4) || slot.h_p.ensure((size_t)(total + 1) * 4)) return -1;   // (scores come back only on this path)
		if (slot.eng.enqueue_host(r1 - r0, local_off + base, raw + off[r0], n, (int32_t*)slot.h_f.ptr + off[r0], (int32_t*)slot.h_p.ptr + off[r0], false)) return -1;
	}
	if (!st.device_post && slot.eng.record_outputs_done(st.done)) return -1;
	st.reads = reads; st.n_read = n_read; st.busy = true; st.opt = opt;
	{
		std::lock_guard<std::mutex> lk(slot.mu);
		st.computed = false; st.err.clear();
		slot.jobs.push_back(&st);                      // the finisher takes it from here
	}
	slot.cv.notify_all();
	if (g_streams.debug)
		fprintf(stderr, "[mm2gb stream] launch: %d reads, %lld anchors, %zu micro-batch(es) | pack %.2f ms | enqueue %.2f ms\n",
		        n_read, (long long)total, n_mb, t_pack - t0, now_ms() - t_pack);
	return 0;
}

static StreamSlot &slot_for(int thread_id)
{
	if (!g_streams.ready) die("chain_stream_gpu called before init_stream_gpu");
	if (thread_id < 0 || thread_id >= (int)g_streams.slots.size())
		die("thread id " + std::to_string(thread_id) + " has no GPU stream: raise num_streams in the gpu config (have " +
		    std::to_string(g_streams.slots.size()) + ")");
	StreamSlot &slot = *g_streams.slots[thread_id];
	{
		std::unique_lock<std::mutex> lk(g_streams.mk_mu);
		g_streams.mk_cv.wait(lk, [&] { return slot.live || !g_streams.mk_err.empty(); });
		if (!slot.live) die(g_streams.mk_err);
	}
	return slot;
}
// every stream is there (or the process ends with what went wrong making one)
static void wait_for_all_streams()
{
	if (g_streams.maker) { g_streams.maker->join(); delete g_streams.maker; g_streams.maker = nullptr; }
	if (!g_streams.mk_err.empty()) die(g_streams.mk_err);
}

// CPUs this process may use at once: affinity mask and cgroup quota, whichever is smaller (containers give 16 of 128 here)
int usable_cpus()
{
	int n = (int)std::thread::hardware_concurrency();
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n > 0 ? n : CPU_COUNT(&set), CPU_COUNT(&set));
	if (FILE *fp = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char quota[64]; long long period = 0;
		if (fscanf(fp, "%63s %lld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
			n = std::min<long long>(n, std::max<long long>(1, atoll(quota) / period));
		fclose(fp);
	}
	return std::max(1, n);
}

// Engines of the synchronous single-read surface (mm2gb_lchain_dp / mm2gb_lchain_rmq): a bounded pool, leased per CALL.  A caller
// takes a free engine (one is created while the pool is below its bound, dealt round-robin over the visible devices, MM2GB_DEVICES),
// or waits for one; the lease's destructor hands it back.  So a host with many or short-lived threads holds at most
// MM2GB_SINGLE_ENGINES engines (default: 8 per device, at most the CPUs the process may use; made on demand) instead of one per thread id ever seen
// -- each is 4 streams, pinned staging and two work arenas -- and a recycled thread id can never share an engine with a live thread.
static std::mutex g_single_mu;
static std::condition_variable g_single_cv;
static std::vector<mm2gb_engine_t*> g_single_free;
static int g_single_made = 0, g_single_max = 0;

struct EngineLease {
	mm2gb_engine_t *eng = nullptr;
	explicit EngineLease(const mm2gb_misc_t &misc)
	{
		std::unique_lock<std::mutex> lock(g_single_mu);
		for (;;) {
			if (!g_single_free.empty()) { eng = g_single_free.back(); g_single_free.pop_back(); return; }
			std::vector<int> devs;
			if (devices_for_streams(devs)) return;
			if (g_single_max == 0) {
				g_single_max = std::max(1, std::min(8 * (int)devs.size(), usable_cpus()));   // a call is a synchronous round trip: a host with 8-16 threads wants an engine each
				if (const char *v = getenv("MM2GB_SINGLE_ENGINES")) g_single_max = std::max(1, atoi(v));
			}
			if (g_single_made < g_single_max) {
				const int dev = devs[(size_t)g_single_made % devs.size()];
				++g_single_made;                               // reserved; creation happens outside the lock (it takes ~0.1 s)
				lock.unlock();
				eng = mm2gb_engine_create(nullptr, &misc, dev);
				if (!eng) { lock.lock(); --g_single_made; g_single_cv.notify_one(); }
				return;
			}
			g_single_cv.wait(lock);
		}
	}
	~EngineLease()
	{
		if (!eng) return;
		{ std::lock_guard<std::mutex> lock(g_single_mu); g_single_free.push_back(eng); }
		g_single_cv.notify_one();
	}
	EngineLease(const EngineLease&) = delete;
	EngineLease &operator=(const EngineLease&) = delete;
};

static void free_single_read_engines()
{
	std::lock_guard<std::mutex> lock(g_single_mu);
	for (mm2gb_engine_t *e : g_single_free) mm2gb_engine_destroy(e);   // (engines on lease belong to calls still running: theirs to return)
	g_single_made -= (int)g_single_free.size();
	g_single_free.clear();
}

// streams that free_stream_gpu parked instead of releasing (see there), and the configuration they were made for
static std::vector<StreamSlot*> g_parked;
static mm2gb_config_t g_parked_cfg;
// Parked streams do not stay parked for ever: a reaper thread releases them -- arenas, page-locked staging, re-chaining engines,
// finisher threads, the leased single-read engines -- when no init_stream_gpu has taken them over within MM2GB_PARK_SECONDS (default 3) of the
// free_stream_gpu that parked them.  A host that exits right after free_stream_gpu (minimap2 does) never pays for the release; a process that
// goes on using the GPU gets the memory back.  MM2GB_FREE=now: released inside free_stream_gpu; MM2GB_FREE=park: kept until the process ends.
static std::mutex g_park_mu;                    // g_parked, g_parked_cfg, g_park_epoch
static std::condition_variable g_park_cv;
static uint64_t g_park_epoch = 0;
static bool g_reaper_started = false, g_reaper_stop = false;
static std::thread g_reaper;
static void destroy_slot(StreamSlot *slot);
static void free_single_read_engines();
static double g_grace_s = 3.0;
static void reaper_main()
{
	std::unique_lock<std::mutex> lk(g_park_mu);
	for (;;) {
		g_park_cv.wait(lk, [] { return g_reaper_stop || !g_parked.empty(); });
		if (g_reaper_stop) return;
		const uint64_t seen = g_park_epoch;
		if (g_park_cv.wait_for(lk, std::chrono::duration<double>(g_grace_s), [&] { return g_reaper_stop || g_park_epoch != seen || g_parked.empty(); })) {
			if (g_reaper_stop) return;
			continue;                               // taken over, or parked anew: the clock starts again
		}
		std::vector<StreamSlot*> mine;
		mine.swap(g_parked);
		lk.unlock();
		for (StreamSlot *slot : mine) destroy_slot(slot);
		free_single_read_engines();
		if (g_streams.debug) fprintf(stderr, "[mm2gb stream] %zu parked stream(s) released: nobody took them over within the grace period\n", mine.size());
		lk.lock();
	}
}
static void stop_reaper_at_exit()
{
	{ std::lock_guard<std::mutex> lk(g_park_mu); g_reaper_stop = true; }
	g_park_cv.notify_all();
	if (g_reaper.joinable()) g_reaper.join();       // (what is still parked goes with the process)
}

static void destroy_slot(StreamSlot *slot)
{
	{ std::lock_guard<std::mutex> lk(slot->mu); slot->stop = true; }
	slot->cv.notify_all();
	if (slot->finisher.joinable()) slot->finisher.join();        // (it drains what is still queued first)
	(void)slot->eng.sync();
	for (HostStage &st : slot->stage) {
		mm2gb_chains_free(&st.ch);
		for (size_t r = 0; r < st.u_of.size(); ++r) { free(st.u_of[r]); free(st.a_of[r]); }
		if (st.done) (void)hipEventDestroy(st.done);
	}
	slot->h_raw.release(); slot->h_f.release(); slot->h_p.release(); slot->h_off.release();
	slot->eng.shutdown();
	if (slot->rmq_eng) mm2gb_engine_destroy(slot->rmq_eng);
	delete slot;
}

} // namespace mm2gb

using namespace mm2gb;

extern "C" {

// ---------------------------------------------------------------------------------------------------------------
// core: whole batch, host buffers
// ---------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// core: one read, mg_lchain_dp's signature (lchain.c:148-217)
// ---------------------------------------------------------------------------------------------------------------
mm2gb_anchor_t *mm2gb_lchain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc,
                                float chn_pen_gap, float chn_pen_skip, int is_cdna, int n_seg, int64_t n, mm2gb_anchor_t *a,
                                int *n_u_, uint64_t **_u, void *km)
{
	HostAlloc mem; mem.km = km; mem.use_kalloc = host_kalloc_present();
	if (!mem.use_kalloc && km) { fprintf(stderr, "[Error] mm2gb_lchain_dp: a kalloc arena was passed but the host allocator is not linked\n"); exit(1); }
	if (_u) *_u = 0, *n_u_ = 0;
	if (n == 0 || a == 0) { mem.release(a); return 0; }                       // lchain.c:156-159
	mm2gb_misc_t misc;
	misc.max_iter = max_iter; misc.max_dist_x = max_dist_x; misc.max_dist_y = max_dist_y; misc.max_skip = max_skip; misc.bw = bw;
	misc.min_cnt = min_cnt; misc.min_score = min_sc; misc.is_cdna = is_cdna; misc.n_seg = n_seg;
	misc.chn_pen_gap = chn_pen_gap; misc.chn_pen_skip = chn_pen_skip;
	EngineLease lease(misc);                               // an engine of the pool for this call: no lock held while the GPU works
	mm2gb_engine_t *eng = lease.eng;
	if (!eng) { fprintf(stderr, "[Error] mm2gb_lchain_dp: %s\n", mm2gb_last_error()); exit(1); }
	const int64_t off[2] = { 0, n };
	std::vector<int32_t> f((size_t)n), p((size_t)n);
	if (mm2gb_engine_set_misc(eng, &misc) || mm2gb_score_host(eng, 1, off, a, f.data(), p.data(), nullptr)) {
		fprintf(stderr, "[Error] mm2gb_lchain_dp: %s\n", mm2gb_last_error()); exit(1);
	}
	BacktrackScratch ws;
	uint64_t *u = nullptr; mm2gb_anchor_t *out = nullptr;
	const int n_u = backtrack_compact(misc, n, a, f.data(), p.data(), mem, ws, &u, &out);
	mem.release(a);                                                          // input is consumed (lchain.c:146,213,109)
	*n_u_ = n_u; *_u = u;
	return out;
}

// ---------------------------------------------------------------------------------------------------------------
// one read, mg_lchain_rmq's signature (lchain.c:250-369)
// ---------------------------------------------------------------------------------------------------------------
static std::atomic<int64_t> g_rmq_calls(0), g_rmq_tied_calls(0), g_rmq_from_ahead(0);
// the answers of the batch whose reads the calling thread is handing to post_chaining_helper right now, and the read it is at (one call per read, map.c:450)
static thread_local const RechainAhead *t_ahead = nullptr;
static thread_local int32_t t_ahead_slot = -1;
// what the library held of a run (MM2GB_REPORT=1 prints it when the host frees the streams): nanoseconds summed over the host's threads
static std::atomic<int64_t> g_ns_chain(0), g_ns_helper(0), g_ns_rmq(0), g_ns_waited(0), g_tot_batches(0), g_tot_reads(0), g_tot_anchors(0);

mm2gb_anchor_t *mm2gb_lchain_rmq(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc,
                                 float chn_pen_gap, float chn_pen_skip, int64_t n, mm2gb_anchor_t *a, int *n_u_, uint64_t **_u, void *km)
{
	HostAlloc mem; mem.km = km; mem.use_kalloc = host_kalloc_present();
	if (!mem.use_kalloc && km) { fprintf(stderr, "[Error] mm2gb_lchain_rmq: a kalloc arena was passed but the host allocator is not linked\n"); exit(1); }
	struct Clock { int64_t t0 = now_ns(); ~Clock() { g_ns_rmq.fetch_add(now_ns() - t0); } } clock;
	if (_u) *_u = 0, *n_u_ = 0;
	if (n == 0 || a == 0) { mem.release(a); return 0; }                       // lchain.c:260-263
	const mm2gb_rmq_param_t prm = { max_dist, max_dist_inner, bw, max_chn_skip, cap_rmq_size, min_cnt, min_sc, chn_pen_gap, chn_pen_skip };
	if (t_ahead && t_ahead_slot >= 0) {
		// this read was re-chained with its whole batch before the callback started (rechain_ahead.cpp): the answer is the call's answer iff the
		// call IS the one that was answered -- same parameters, same anchors in the same order, byte for byte
		const RechainAhead &ah = *t_ahead;
		const int32_t s = t_ahead_slot;
		t_ahead_slot = -1;
		if (s + 1 < (int32_t)ah.off.size() && n == ah.off[s + 1] - ah.off[s] && memcmp(&prm, &ah.prm, sizeof prm) == 0 &&
		    memcmp(a, ah.sorted.data() + ah.off[s], (size_t)n * sizeof(mm2gb_anchor_t)) == 0) {
			const mm2gb_chains_t &c = ah.res.c;
			const int n_u = (int)(c.u_off[s + 1] - c.u_off[s]);
			const int64_t n_a = c.a_off[s + 1] - c.a_off[s];
			uint64_t *u = nullptr; mm2gb_anchor_t *res = nullptr;
			if (n_u > 0) {
				u = (uint64_t*)mem.alloc((size_t)n_u * 8);
				res = (mm2gb_anchor_t*)mem.alloc((size_t)n_a * 16);
				memcpy(u, c.u + c.u_off[s], (size_t)n_u * 8); memcpy(res, c.a + c.a_off[s], (size_t)n_a * 16);
			}
			mem.release(a);                                                      // input is consumed (lchain.c:357-360,109)
			g_rmq_calls.fetch_add(1); g_rmq_from_ahead.fetch_add(1);
			*n_u_ = n_u; *_u = u;
			return res;
		}
	}
	const int64_t off[2] = { 0, n };
	mm2gb_chains_t out;
	int32_t tied = 0;
	// One read at a time: the fill is a chain of n dependent steps either way.  On the host it is O(log n) per step (segment tree,
	// csrc/rmq_host.cpp); the kernel scans the window at every step and is for batches of reads with narrow windows (MM2GB_RMQ=gpu forces it).
	static const bool on_device = [] { const char *v = getenv("MM2GB_RMQ"); return v && strcmp(v, "gpu") == 0; }();
	if (!on_device) {
		// exact for every read: a read that meets a tie is done again with the reference's tree inside the call, nothing goes back to the host
		if (mm2gb_rmq_chain_host(&prm, 1, off, a, 1, &out, nullptr)) { fprintf(stderr, "[Error] mm2gb_lchain_rmq: %s\n", mm2gb_last_error()); exit(1); }
	} else {
		mm2gb_misc_t misc = {};                                               // the engine wants one; the re-chaining call carries its own thresholds
		misc.max_iter = 5000; misc.max_dist_x = max_dist; misc.max_dist_y = max_dist; misc.max_skip = max_chn_skip; misc.bw = std::min(bw, 8000);
		misc.min_cnt = min_cnt; misc.min_score = min_sc; misc.n_seg = 1; misc.chn_pen_gap = chn_pen_gap; misc.chn_pen_skip = chn_pen_skip;
		EngineLease lease(misc);
		mm2gb_engine_t *eng = lease.eng;
		if (!eng) { fprintf(stderr, "[Error] mm2gb_lchain_rmq: %s\n", mm2gb_last_error()); exit(1); }
		if (mm2gb_rmq_chain_gpu(eng, &prm, 1, off, a, &out, &tied, nullptr)) { fprintf(stderr, "[Error] mm2gb_lchain_rmq: %s\n", mm2gb_last_error()); exit(1); }
	}
	g_rmq_calls.fetch_add(1);
	if (tied != 0) {
		// the reference's answer depends on the shape of its tree here (krmq.h:110-147): the read is done again by the library's own
		// exact host form (csrc/rmq_host.cpp: the reference's tree rules on index arrays).  Nothing is ever handed to the host program.
		mm2gb_chains_free(&out);
		g_rmq_tied_calls.fetch_add(1);
		if (mm2gb_rmq_chain_host(&prm, 1, off, a, 1, &out, nullptr)) { fprintf(stderr, "[Error] mm2gb_lchain_rmq: %s\n", mm2gb_last_error()); exit(1); }
	}
	const int n_u = (int)out.u_off[1];
	uint64_t *u = nullptr; mm2gb_anchor_t *res = nullptr;
	if (n_u > 0) {
		u = (uint64_t*)mem.alloc((size_t)n_u * 8);
		res = (mm2gb_anchor_t*)mem.alloc((size_t)out.a_off[1] * 16);
		memcpy(u, out.u, (size_t)n_u * 8); memcpy(res, out.a, (size_t)out.a_off[1] * 16);
	}
	mm2gb_chains_free(&out);
	mem.release(a);                                                          // input is consumed (lchain.c:357-360,109)
	*n_u_ = n_u; *_u = u;
	return res;
}

// link-time interposition for an unmodified host: -Wl,--wrap=mg_lchain_rmq sends map.c:450's call here
mm2gb_anchor_t *__wrap_mg_lchain_rmq(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc,
                                     float chn_pen_gap, float chn_pen_skip, int64_t n, mm2gb_anchor_t *a, int *n_u_, uint64_t **_u, void *km)
{
	return mm2gb_lchain_rmq(max_dist, max_dist_inner, bw, max_chn_skip, cap_rmq_size, min_cnt, min_sc, chn_pen_gap, chn_pen_skip, n, a, n_u_, _u, km);
}

// how many single-read re-chaining calls there were, and how many of them (device form only) met a tie and were redone by the exact host form
void mm2gb_lchain_rmq_counts(int64_t *calls, int64_t *tied_calls)
{
	if (calls) *calls = g_rmq_calls.load();
	if (tied_calls) *tied_calls = g_rmq_tied_calls.load();
}

void mm2gb_rechain_ahead_counts(int64_t *calls, int64_t *answered_ahead, int64_t *reads_ahead)
{
	if (calls) *calls = g_rmq_calls.load();
	if (answered_ahead) *answered_ahead = g_rmq_from_ahead.load();
	if (reads_ahead) *reads_ahead = g_reads_ahead.load();
}

// The host's callback for every read of a batch that a boundary call hands back (plchain.cu:502-507, 539-541).  The batch's re-chaining
// calls were answered together by the stream's finisher where that applies (rechain_ahead.cpp): each read's call finds its answer through
// the thread-local view set here.
static void hand_to_host_callback(HostStage &st, const struct mm_idx_s *mi, const struct mm_mapopt_s *opt, mm2gb_chain_read_t *done, int n_done,
                                  const mm2gb_misc_t &misc, void *km)
{
	if (done && post_chaining_helper) {
		TraceRange range("mm2gb:post_chaining_helper");
		const bool have = st.have_ahead && (int)st.ahead.slot_of_read.size() == n_done;
		for (int i = 0; i < n_done; ++i) {
			if (have) { t_ahead = &st.ahead; t_ahead_slot = st.ahead.slot_of_read[(size_t)i]; }
			post_chaining_helper(mi, opt, &done[i], misc, km);
		}
		t_ahead = nullptr; t_ahead_slot = -1;
	}
	if (st.have_ahead) { st.ahead.clear(); st.have_ahead = false; }
}

// ---------------------------------------------------------------------------------------------------------------
// the reference's boundary
// ---------------------------------------------------------------------------------------------------------------
void init_stream_gpu(size_t *max_total_n, int *max_reads, int *min_n, char gpu_config_file[], mm2gb_Misc misc)
{
	if (g_streams.ready) free_stream_gpu((int)g_streams.slots.size());
	g_t_init_ns = now_ns();
	const double epoch_in = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
	if (mm2gb_config_load(gpu_config_file, &g_streams.cfg)) die(mm2gb_last_error());
	mm2gb_config_t &cfg = g_streams.cfg;
	// Every stream id is an engine with four HIP streams (copy in, two compute, copy out), its re-chaining engine (rechain_ahead.cpp) four more,
	// and single-read calls lease up to eight engines per device.  The HIP runtime multiplexes streams onto 4 hardware queues unless told
	// otherwise, and streams that share a queue run one after the other: four host threads driving 64-read batches reach 0.28-0.49 G anchors/s
	// with 8 queues, 0.80 with 16 (one thread: 0.26; profiles/r03_small_batches.txt).  But MORE is not better: past what the device has slots for
	// the queues themselves are time-sliced -- the drop-in at -t 16 (128 HIP streams) maps 1.05 Gbp in 11.8-12.0 s with 8, 11.1-11.3 with 12,
	// 10.2 with 16, 10.1-10.9 with 20, 10.4-10.6 with 24, 11.1-11.6 with 32, 11.6-12.4 with 64 (which rounds 4-5 asked for), and the own host's
	// four engines are fastest at 16 too.  So: 8 for a single stream, 16 from two streams on.  Only effective before the runtime starts, i.e. when
	// this is the first HIP call of the process, as it is in the minimap2 host; a value that is already set is the host's to choose, but one
	// below 8 is worth a line.
	const int want_queues = cfg.num_streams > 1 ? 16 : 8;
	if (const char *q = getenv("GPU_MAX_HW_QUEUES")) {
		if (atoi(q) < 8)
			fprintf(stderr, "[mm2gb] GPU_MAX_HW_QUEUES=%s with num_streams=%d: streams will share hardware queues and serialise (this library would ask for %d)\n", q, cfg.num_streams, want_queues);
	} else setenv("GPU_MAX_HW_QUEUES", std::to_string(want_queues).c_str(), 0);
	if (!(cfg.has_max_total_n && cfg.has_max_read)) {
		// auto-size from avg_read_n like plmem.cu:497-539, against this device's memory and this engine's footprint per anchor:
		// 16 B of work arrays + two staging sets of 24 B (raw in, f and p out); with the device post-pass (MM2GB_POST=gpu) its 41 B of work
		// arrays and two result sets of 24 B on top; a stream's re-chaining engine (rechain_ahead.cpp, ~200 B per KEPT anchor, a third of them) beside it
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) die("cannot query device memory");
		{ const char *pm = getenv("MM2GB_POST"); const char *ptz = getenv("MM2GB_POST_THREADS"); g_streams.post_on_device = (pm && strcmp(pm, "gpu") == 0) || (ptz && atoi(ptz) == 0); }
		const double per_anchor = 64.0 + 16.0 / 1024 + (g_streams.post_on_device ? 41.0 + 48.0 : 0.0) + 70.0;
		const double budget = (double)total_b / cfg.num_streams * 0.8;
		int64_t n = (int64_t)(budget / per_anchor);
		if (n > 2000000000LL) n = 2000000000LL;
		cfg.max_total_n = n;
		cfg.max_read = (int)std::min<int64_t>(n / std::max(1, cfg.avg_read_n) + 1, 2147483647LL / std::max(1, cfg.score_kernel.micro_batch));
	}
	g_streams.misc = misc;
	const char *pt = getenv("MM2GB_POST_THREADS");
	// host threads for packing anchors and for backtrack + compaction; results enter the host's kalloc arena on the calling
	// thread only, so this is safe with the single-threaded reference host
	// default: this stream's share of the CPUs the process may use (one GPU's post-pass needs ~13 CPU-equivalents to stay hidden)
	g_streams.post_threads = pt ? std::max(1, atoi(pt)) : std::max(1, std::min(32, usable_cpus() / std::max(1, cfg.num_streams)));
	{ const char *dbg = getenv("MM2GB_DEBUG_PHASES"); g_streams.debug = dbg && *dbg && *dbg != '0'; }
	{ const char *pm = getenv("MM2GB_POST"); g_streams.post_on_device = (pm && strcmp(pm, "gpu") == 0) || (pt && atoi(pt) == 0); }
	// re-chaining ahead only helps a host whose mg_lchain_rmq calls reach this library: one linked with -Wl,--wrap=mg_lchain_rmq imports
	// __wrap_mg_lchain_rmq (an undefined dynamic symbol of the program); MM2GB_PRECHAIN=0 / 1 overrides what its symbol table says
	{
		const char *pc = getenv("MM2GB_PRECHAIN");
		g_streams.rechain_ahead = pc && *pc ? atoi(pc) != 0 : elf_imports_symbol("/proc/self/exe", "__wrap_mg_lchain_rmq");
		// the stream's post-pass threads, plus the host thread's own CPU: it waits for the finisher while the batch it just launched is finished
		const char *at = getenv("MM2GB_AHEAD_THREADS");
		g_streams.ahead_threads = at ? std::max(1, atoi(at)) : g_streams.post_threads + 1;
	}
	std::vector<int> devs;
	if (devices_for_streams(devs)) die(mm2gb_last_error());
	std::unique_lock<std::mutex> park_lk(g_park_mu);
	++g_park_epoch;
	if (!g_parked.empty()) {
		// streams parked by free_stream_gpu: taken over as they are when they were made for this configuration and these devices
		bool same = memcmp(&g_parked_cfg, &cfg, sizeof cfg) == 0 && (int)g_parked.size() == cfg.num_streams;
		for (int s = 0; same && s < cfg.num_streams; ++s) same = g_parked[(size_t)s]->eng.device == devs[(size_t)s % devs.size()];
		if (same) {
			for (StreamSlot *slot : g_parked) { if (slot->eng.set_misc(&misc)) die(mm2gb_last_error()); slot->cur = 0; g_streams.slots.push_back(slot); }
			g_parked.clear();
		} else {
			for (StreamSlot *slot : g_parked) destroy_slot(slot);
			g_parked.clear();
		}
	}
	park_lk.unlock();
	g_park_cv.notify_all();
	{
		// The engines that are not there yet -- streams, events, the penalty table's kernel, ~40 ms each and one after the other inside the
		// runtime whoever asks (sixteen: 0.6 s) -- are made by a thread of their own, in stream order: the host goes on to read and seed its first
		// batch (seconds), and a stream's first boundary call waits for its engine only if it comes earlier than that.  MM2GB_INIT=wait keeps
		// init_stream_gpu until every engine stands (tests that time the call; a host that wants configuration errors from init itself).
		const int have = (int)g_streams.slots.size();
		std::vector<StreamSlot*> fresh;
		for (int s = have; s < cfg.num_streams; ++s) { fresh.push_back(new StreamSlot()); g_streams.slots.push_back(fresh.back()); }
		for (int s = 0; s < have; ++s) g_streams.slots[(size_t)s]->live = true;
		g_streams.mk_err.clear();
		if (!fresh.empty()) {
			const mm2gb_config_t cfg_copy = cfg;
			const mm2gb_misc_t misc_copy = misc;
			g_streams.maker = new std::thread([fresh, devs, have, cfg_copy, misc_copy]() {
				for (size_t k = 0; k < fresh.size(); ++k) {
					std::string err;
					try {
						if (fresh[k]->eng.init(&cfg_copy, &misc_copy, devs[(size_t)(have + (int)k) % devs.size()])) err = mm2gb_last_error();
						else fresh[k]->finisher = std::thread(finisher_main, fresh[k]);
					} catch (const std::exception &ex) { err = std::string("init_stream_gpu: making a stream: ") + ex.what(); }
					std::lock_guard<std::mutex> lk(g_streams.mk_mu);
					if (err.empty()) fresh[k]->live = true; else g_streams.mk_err = err;
					g_streams.mk_cv.notify_all();
					if (!err.empty()) return;
				}
				// ... and the streams' re-chaining engines (rechain_ahead.cpp), which the finishers would otherwise all make at the same moment -- the end
				// of the first batch's chaining, one after the other inside the runtime, 40 ms each.  A failure here is not fatal: the finisher tries again.
				const char *ra = getenv("MM2GB_RMQ_ENGINES_AHEAD");
				if (g_streams.rechain_ahead && !(ra && *ra == '0'))
					for (StreamSlot *slot : fresh) {
						std::lock_guard<std::mutex> lk(slot->rmq_mu);
						if (!slot->rmq_eng) {
							slot->rmq_eng = mm2gb_engine_create(&cfg_copy, &misc_copy, slot->eng.device);
							if (slot->rmq_eng) slot->rmq_eng->e.rmq_calibrate = true;
						}
					}
			});
			const char *iw = getenv("MM2GB_INIT");
			if (iw && strcmp(iw, "wait") == 0) wait_for_all_streams();
		}
	}
	g_streams.ready = true;
	if (g_streams.debug) {
		int n_new = 0;
		{ std::lock_guard<std::mutex> lk(g_streams.mk_mu); for (StreamSlot *slot : g_streams.slots) n_new += !slot->live; }
		fprintf(stderr, "[mm2gb stream] init_stream_gpu: entered at epoch %.3f, returns after %.3f s: %d stream(s), %d of them still being made (beside the host's first batch)\n",
		        epoch_in, (now_ns() - g_t_init_ns) * 1e-9, cfg.num_streams, n_new);
	}
	// what the host accumulates to (plmem.cu:616-617)
	*max_total_n = (size_t)cfg.max_total_n * (size_t)cfg.score_kernel.micro_batch;
	*max_reads = (int)std::min<int64_t>((int64_t)cfg.max_read * cfg.score_kernel.micro_batch, 2147483647LL);
	*min_n = cfg.min_n;
}

void chain_stream_gpu(const struct mm_idx_s *mi, const struct mm_mapopt_s *opt, mm2gb_chain_read_t **in_arr_ptr, int *n_read_ptr,
                      int thread_id, void *km)
{
	const int64_t t_in = now_ns();
	StreamSlot &slot = slot_for(thread_id);
	mm2gb_Misc misc = build_misc ? build_misc(mi, opt, 0, 1) : g_streams.misc;   // plchain.cu:500
	mm2gb_chain_read_t *new_reads = *in_arr_ptr;
	const int n_new = *n_read_ptr;
	if (new_reads && n_new > 0) {
		int64_t na = 0;
		for (int i = 0; i < n_new; ++i) na += new_reads[i].n;
		g_tot_batches.fetch_add(1); g_tot_reads.fetch_add(n_new); g_tot_anchors.fetch_add(na);
	}
	HostStage &prev = slot.stage[slot.cur], &next = slot.stage[slot.cur ^ 1];
	// The previous batch has been with the stream's finisher since it was launched (scores -> chains -> its re-chaining answered ahead),
	// while this thread seeded the batch it brings now: normally it is through.  Wait for it -- the engine serves one thread at a time --,
	// launch the new batch, then hand the previous one back.  (The reference finishes the previous batch on the calling thread before
	// it launches, plchain.cu:300-305, and leaves the GPU idle during its post-pass.)
	const int64_t t_w0 = now_ns();
	wait_computed(slot, prev);
	g_ns_waited.fetch_add(now_ns() - t_w0);
	if (slot.eng.set_misc(&misc)) die(mm2gb_last_error());
	const bool launched = new_reads && n_new > 0;
	if (launched && launch_stage(slot, next, new_reads, n_new, opt)) die(mm2gb_last_error());
	mm2gb_chain_read_t *done = nullptr; int n_done = 0;
	mm2gb_misc_t done_misc = misc;                       // the batch handed back is finished with the parameters IT was launched with
	if (finish_handover(slot, prev, km, &done, &n_done, &done_misc)) die(mm2gb_last_error());
	if (launched) slot.cur ^= 1;
	*in_arr_ptr = done; *n_read_ptr = n_done;
	const int64_t t_mid = now_ns();
	hand_to_host_callback(prev, mi, opt, done, n_done, done_misc, km);                         // plchain.cu:502-507
	const int64_t t_out = now_ns();
	g_ns_chain.fetch_add(t_mid - t_in); g_ns_helper.fetch_add(t_out - t_mid);
}

void finish_stream_gpu(const struct mm_idx_s *mi, const struct mm_mapopt_s *opt, mm2gb_chain_read_t **batches, int *num_reads,
                       int num_batch, void *km)
{
	const int64_t t_in = now_ns();
	StreamSlot &slot = slot_for(num_batch);
	mm2gb_Misc misc = build_misc ? build_misc(mi, opt, 0, 1) : g_streams.misc;
	mm2gb_chain_read_t *done = nullptr; int n_done = 0;
	HostStage &st = slot.stage[slot.cur];
	wait_computed(slot, st);
	g_ns_waited.fetch_add(now_ns() - t_in);
	if (finish_handover(slot, st, km, &done, &n_done, &misc)) die(mm2gb_last_error());
	const int64_t t_mid = now_ns();
	hand_to_host_callback(st, mi, opt, done, n_done, misc, km);                                // plchain.cu:539-541
	g_ns_chain.fetch_add(t_mid - t_in); g_ns_helper.fetch_add(now_ns() - t_mid);
	*batches = done; *num_reads = n_done;
}

void free_stream_gpu(int n_threads)
{
	(void)n_threads;
	const int64_t t_free0 = now_ns();
	const double epoch_free = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
	// Giving 16 streams' engines back one by one costs 2.5 s of a 16 s program (profiles/r05_dropin_notes.md; the process's end does it in
	// 0.6 s, profiles/r05_exit_cost.txt), and the host calls this as the last thing before it exits (main.c:465).  So the streams are PARKED:
	// work drained, engines idle, nothing released -- a later init_stream_gpu with the same configuration takes them over as they are (no
	// re-pinning either), another configuration or MM2GB_FREE=now releases them for real, and the process's end releases what is parked.
	static const bool release_now = [] { const char *v = getenv("MM2GB_FREE"); return v && strcmp(v, "now") == 0; }();
	static const bool park_for_ever = [] { const char *v = getenv("MM2GB_FREE"); return v && strcmp(v, "park") == 0; }();
	const double grace_s = [] { const char *v = getenv("MM2GB_PARK_SECONDS"); return v && *v ? std::max(0.0, atof(v)) : 3.0; }();
	wait_for_all_streams();                                       // (a host with nothing to map gets here before the engines stand)
	for (StreamSlot *slot : g_streams.slots) {
		{
			std::unique_lock<std::mutex> lk(slot->mu);
			slot->cv.wait(lk, [&] { return slot->jobs.empty(); });        // the finisher is through with what was launched
		}
		(void)slot->eng.sync();
		for (HostStage &st : slot->stage) {
			// a batch the host never came back for: what its finisher ran into is still said (the host is about to exit; nothing is thrown away silently)
			if (!st.err.empty()) fprintf(stderr, "[Error] free_stream_gpu: a batch that was launched and never collected had failed: %s\n", st.err.c_str());
			st.err.clear();
			mm2gb_chains_free(&st.ch);
			for (size_t r = 0; r < st.u_of.size(); ++r) { free(st.u_of[r]); free(st.a_of[r]); }
			st.u_of.clear(); st.a_of.clear(); st.nu_of.clear();
			st.ahead.clear(); st.have_ahead = false;
			st.busy = false; st.reads = nullptr; st.n_read = 0; st.device_post = false;
		}
		if (release_now) destroy_slot(slot);
		else { std::lock_guard<std::mutex> lk(g_park_mu); g_parked.push_back(slot); }
	}
	if (!release_now) {
		std::lock_guard<std::mutex> lk(g_park_mu);
		g_parked_cfg = g_streams.cfg;
		++g_park_epoch;
		g_grace_s = grace_s;
		if (!park_for_ever && !g_reaper_started) {
			g_reaper_started = true;
			g_reaper = std::thread(reaper_main);
			atexit(stop_reaper_at_exit);                              // registered after the runtime's own handlers: runs before them
		}
	}
	g_park_cv.notify_all();
	g_streams.slots.clear();
	g_streams.ready = false;
	if (release_now) free_single_read_engines();
	if (g_streams.debug) {
		// what the process holds when the host is about to exit (its end is not the library's to time, but most of it is memory going away)
		long rss_kb = 0, hwm_kb = 0, huge_kb = 0;
		if (FILE *fp = fopen("/proc/self/status", "r")) {
			char line[256];
			while (fgets(line, sizeof line, fp)) { (void)sscanf(line, "VmRSS: %ld", &rss_kb); (void)sscanf(line, "VmHWM: %ld", &hwm_kb); }
			fclose(fp);
		}
		if (FILE *fp = fopen("/proc/self/smaps_rollup", "r")) {
			char line[256];
			while (fgets(line, sizeof line, fp)) (void)sscanf(line, "AnonHugePages: %ld", &huge_kb);
			fclose(fp);
		}
		fprintf(stderr, "[mm2gb stream] free_stream_gpu: entered at epoch %.3f, streams %s after %.3f s; the process holds %.1f GB of memory (peak %.1f), %.1f GB of it in huge pages\n",
		        epoch_free, release_now ? "released" : park_for_ever ? "parked" : "parked (released if not taken over within the grace period)", (now_ns() - t_free0) * 1e-9, rss_kb / 1048576.0, hwm_kb / 1048576.0, huge_kb / 1048576.0);
	}
	// What the library held of the run, for whoever times the drop-in (bench.py's e2e.reference_host_at_scale): seconds are summed over the
	// host's threads; the host's own callback (post_chaining_helper, map.c:428: RMQ re-chaining, mm_gen_regs, ...) runs inside the boundary
	// calls and is listed apart, and of it what mg_lchain_rmq calls answered by the library took (hosts linked with --wrap=mg_lchain_rmq).
	if (const char *v = getenv("MM2GB_REPORT"))
		if (*v && *v != '0') fprintf(stderr, "[mm2gb totals] batches %lld reads %lld anchors %lld | inside chain_stream_gpu / finish_stream_gpu without the host's callback %.3f s | "
		                             "host callback post_chaining_helper %.3f s | of the callback: mg_lchain_rmq answered by the library %.3f s in %lld calls | "
		                             "re-chaining ahead of the callback %.3f s for %lld reads, %lld calls answered from it | of the boundary time: waiting for the stream's finisher %.3f s\n",
		                             (long long)g_tot_batches.load(), (long long)g_tot_reads.load(), (long long)g_tot_anchors.load(), g_ns_chain.load() * 1e-9,
		                             g_ns_helper.load() * 1e-9, g_ns_rmq.load() * 1e-9, (long long)g_rmq_calls.load(),
		                             g_ns_ahead.load() * 1e-9, (long long)g_reads_ahead.load(), (long long)g_rmq_from_ahead.load(), g_ns_waited.load() * 1e-9);
	if (const char *v = getenv("MM2GB_RMQ_REPORT"))
		if (*v && *v != '0') fprintf(stderr, "[mm2gb] mg_lchain_rmq calls answered by the library: %lld, of which redone by the library's exact host form because of a tie (device form only): %lld; handed to the host program: 0\n",
		                             (long long)g_rmq_calls.load(), (long long)g_rmq_tied_calls.load());
}

} // extern "C"
