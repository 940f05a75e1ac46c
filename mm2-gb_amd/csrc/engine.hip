// engine.hip -- device arenas, stream and the per-micro-batch launch sequence.
// Replaces plmem_stream_initialize / plmem_*_memcpy (plmem.cu:12-359, 558-641) and the kernel sequencing of
// plchain_cal_score_async (plchain.cu:292-464) with a design that never leaves the stream between stages:
// split -> window(+planner reductions) -> plan -> score are all enqueued back to back, no host sort, no hipMalloc per batch.
#include <atomic>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <mutex>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "engine.h"
#include "host_chain.h"

namespace mm2gb {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int fail(const std::string &msg) { g_err = msg; return -1; }
const char *last_error_cstr() { return g_err.c_str(); }

// Buffers only grow.  A buffer that has to grow again takes a quarter more than asked for: batches of slowly increasing size
// (reads differ) would otherwise reallocate -- and, for page-locked memory, re-pin at ~0.4 s per GB -- at every new maximum.
static size_t grown(size_t need, size_t have)
{
	return have == 0 ? need : std::max(need, have + have / 4);
}

// hipFree and hipHostFree wait for the WHOLE device to go idle.  One engine on its own never noticed; sixteen engines working side by side (the
// drop-in at -t 16: a stream id and a re-chaining engine per host thread) did: an arena that had to grow in one engine waited for every kernel
// of every other engine, and the calls of a mini-batch ran one after the other (10 s for what takes 1.6 s, profiles/r05_dropin_notes.md).  So a
// buffer that is REPLACED because it must grow is not freed but retired: it goes to a process-wide list that is emptied when an engine shuts down,
// when the list holds more than MM2GB_RETIRE_LIMIT_MB (default 16 GB of device memory, 4 GB page-locked), or when an allocation fails.
// (Growth is geometric, by a quarter: what a buffer retires over its life is a geometric series of its earlier sizes -- up to four times its
// final size if nothing flushed the list in between; the limits above bound it.)  A retired DEVICE buffer stays valid for work enqueued
// before the growth (hipFree waits for the device); retired page-locked blocks are recycled only after every device has gone idle (below).
namespace { void pinned_free(void *p); }
namespace {
std::atomic<unsigned long long> g_devices_with_engines{0};   // bit d: an engine was made on device d
struct Retired { void *ptr; size_t bytes; bool pinned; };
std::mutex g_retired_mu;
std::vector<Retired> g_retired;
size_t g_retired_dev = 0, g_retired_host = 0;

void flush_retired_locked()
{
	// a page-locked block on the list may still be the source or target of a copy some engine enqueued before it grew: nothing is handed out again
	// before those copies are through (the list is process-wide: every device that is in use)
	bool any_pinned = false;
	for (const Retired &r : g_retired) any_pinned |= r.pinned;
	if (any_pinned) {
		int cur = 0;
		const unsigned long long used = g_devices_with_engines.load();        // (only devices this process has engines on: touching another one would make a context there)
		if (hipGetDevice(&cur) == hipSuccess) {
			for (int d = 0; d < 64; ++d) if ((used >> d) & 1ull) if (hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();
			(void)hipSetDevice(cur);
		}
		(void)hipGetLastError();
	}
	for (const Retired &r : g_retired) { if (r.pinned) pinned_free(r.ptr); else (void)hipFree(r.ptr); }
	g_retired.clear();
	g_retired_dev = g_retired_host = 0;
}
void retire(void *ptr, size_t bytes, bool pinned)
{
	if (!ptr) return;
	static const size_t limit_mb = [] { const char *v = getenv("MM2GB_RETIRE_LIMIT_MB"); return v ? (size_t)std::max(0LL, atoll(v)) : (size_t)16384; }();
	std::lock_guard<std::mutex> lock(g_retired_mu);
	g_retired.push_back({ ptr, bytes, pinned });
	(pinned ? g_retired_host : g_retired_dev) += bytes;
	if (g_retired_dev > (limit_mb << 20) || g_retired_host > (limit_mb << 18)) flush_retired_locked();
}
} // namespace
void flush_retired_buffers()
{
	std::lock_guard<std::mutex> lock(g_retired_mu);
	flush_retired_locked();
}

int DevBuf::ensure(size_t need)
{
	if (need <= bytes) return 0;
	// The new buffer is allocated BEFORE the old one is freed: a failed growth then leaves the buffer as it was (contents are
	// never carried over: every arena is rewritten by the batch that needs it).  Only when there is no room for both is the old
	// one given up first; if even that fails the buffer is empty (ptr null, bytes 0) and the caller must treat its capacity as lost.
	const size_t want = grown(need, bytes);
	void *fresh = nullptr;
	if (hipMalloc(&fresh, want) == hipSuccess) { retire(ptr, bytes, false); ptr = fresh; bytes = want; return 0; }
	(void)hipGetLastError();
	flush_retired_buffers();
	if (hipMalloc(&fresh, need) == hipSuccess) { retire(ptr, bytes, false); ptr = fresh; bytes = need; return 0; }
	(void)hipGetLastError();
	release();
	MM2GB_HIP(hipMalloc(&ptr, need));
	bytes = need;
	return 0;
}
void DevBuf::release()
{
	if (ptr) (void)hipFree(ptr);
	ptr = nullptr; bytes = 0;
}

// Page-locked host memory.  hipHostMalloc takes ~0.16 s per GB and is serialised across threads (16 streams pinning 256 MB each at the same
// moment: 0.77 s, and 0.79 s to give it back; profiles/ubench/pin_rate.hip), because it pins 4 KB pages one by one.  A 2 MB-aligned block
// with MADV_HUGEPAGE, touched and then registered (hipHostRegister), is pinned in 2 MB pages: 0.06 s per GB, 0.026 s for the same 16 x 256 MB,
// and copies from it run at the link's rate all the same.  MM2GB_PIN=hipmalloc keeps the runtime's allocator.
namespace {
bool pin_by_register() { static const bool v = [] { const char *e = getenv("MM2GB_PIN"); return !(e && strcmp(e, "hipmalloc") == 0); }(); return v; }
// A block that is given back goes to a free list and is handed out again (best fit, any idle block that is large enough).  What the list holds
// beyond MM2GB_PIN_IDLE_MB (default 2048) is really let go, largest first: unregistered, and its pages returned to the system -- but its
// ADDRESS RANGE stays reserved (mapped again PROT_NONE, no memory behind it) for as long as the process lives.  Unregistering is what went
// wrong once: the blocks came from malloc, glibc reused a freed block's addresses for ordinary allocations, and a later copy from such an
// allocation faulted on the GPU ("Memory access fault ... Reason: Unknown", an address inside a former block).  So the blocks are mapped
// directly (mmap), outside malloc's heap, and an address that once was page-locked is never handed to anybody again.
struct PinBlock { void *ptr; size_t bytes; };
std::mutex g_pin_mu;
std::vector<PinBlock> g_pin_idle, g_pin_all;
size_t g_pin_idle_bytes = 0;
// idle blocks over the cap: out of the lists under the lock, let go outside it (hipHostUnregister may wait for the device)
void trim_idle_locked(std::vector<PinBlock> &out)
{
	static const size_t cap = [] { const char *v = getenv("MM2GB_PIN_IDLE_MB"); return (size_t)(v && *v ? std::max(0LL, atoll(v)) : 2048LL) << 20; }();
	while (g_pin_idle_bytes > cap && !g_pin_idle.empty()) {
		size_t big = 0;
		for (size_t k = 1; k < g_pin_idle.size(); ++k) if (g_pin_idle[k].bytes > g_pin_idle[big].bytes) big = k;
		const PinBlock blk = g_pin_idle[big];
		g_pin_idle.erase(g_pin_idle.begin() + (std::ptrdiff_t)big);
		g_pin_idle_bytes -= blk.bytes;
		for (size_t k = 0; k < g_pin_all.size(); ++k) if (g_pin_all[k].ptr == blk.ptr) { g_pin_all.erase(g_pin_all.begin() + (std::ptrdiff_t)k); break; }
		out.push_back(blk);
	}
}
void let_go(const std::vector<PinBlock> &blocks)
{
	for (const PinBlock &blk : blocks) {
		(void)hipHostUnregister(blk.ptr);
		(void)hipGetLastError();
		// the pages go back to the system, the addresses stay taken: a fresh PROT_NONE mapping over the same range
		if (mmap(blk.ptr, blk.bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED | MAP_NORESERVE, -1, 0) == MAP_FAILED) (void)madvise(blk.ptr, blk.bytes, MADV_DONTNEED);
	}
}
void *pinned_alloc(size_t bytes)
{
	if (!pin_by_register()) { void *p = nullptr; if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; } return p; }
	constexpr size_t HUGE = (size_t)2 << 20;
	const size_t rounded = (std::max<size_t>(bytes, 1) + (HUGE - 1)) & ~(HUGE - 1);
	{
		std::lock_guard<std::mutex> lock(g_pin_mu);
		size_t best = g_pin_idle.size();
		for (size_t k = 0; k < g_pin_idle.size(); ++k)
			if (g_pin_idle[k].bytes >= rounded && (best == g_pin_idle.size() || g_pin_idle[k].bytes < g_pin_idle[best].bytes)) best = k;
		if (best != g_pin_idle.size()) { void *p = g_pin_idle[best].ptr; g_pin_idle_bytes -= g_pin_idle[best].bytes; g_pin_idle.erase(g_pin_idle.begin() + (std::ptrdiff_t)best); return p; }
	}
	char *raw = (char*)mmap(nullptr, rounded + HUGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	if (raw == (char*)MAP_FAILED) return nullptr;
	char *p = (char*)(((uintptr_t)raw + (HUGE - 1)) & ~(uintptr_t)(HUGE - 1));
	if (p > raw) (void)munmap(raw, (size_t)(p - raw));
	if (p + rounded < raw + rounded + HUGE) (void)munmap(p + rounded, (size_t)(raw + rounded + HUGE - (p + rounded)));
	(void)madvise(p, rounded, MADV_HUGEPAGE);
	for (size_t off = 0; off < rounded; off += 4096) ((volatile char*)p)[off] = 0;       // first touch: the pages exist (as huge pages where the system grants them) before they are pinned
	if (hipHostRegister(p, rounded, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); (void)munmap(p, rounded); return nullptr; }
	std::lock_guard<std::mutex> lock(g_pin_mu);
	g_pin_all.push_back({ p, rounded });
	return p;
}
void pinned_free(void *p)
{
	if (!p) return;
	if (!pin_by_register()) { (void)hipHostFree(p); return; }
	std::vector<PinBlock> over;
	{
		std::lock_guard<std::mutex> lock(g_pin_mu);
		for (const PinBlock &b : g_pin_all) if (b.ptr == p) { g_pin_idle.push_back(b); g_pin_idle_bytes += b.bytes; break; }
		trim_idle_locked(over);
	}
	let_go(over);
}
} // namespace

int PinnedBuf::ensure(size_t need)
{
	if (need <= bytes) return 0;
	const size_t want = grown(need, bytes);
	void *fresh = pinned_alloc(want);                    // as DevBuf::ensure: a failed growth keeps the old buffer
	if (fresh) { retire(ptr, bytes, true); ptr = fresh; bytes = want; return 0; }
	flush_retired_buffers();
	if ((fresh = pinned_alloc(need)) != nullptr) { retire(ptr, bytes, true); ptr = fresh; bytes = need; return 0; }
	release();
	if ((ptr = pinned_alloc(need)) == nullptr) return fail("mm2gb: cannot allocate " + std::to_string(need) + " bytes of page-locked host memory");
	bytes = need;
	return 0;
}
void PinnedBuf::release()
{
	pinned_free(ptr);
	ptr = nullptr; bytes = 0;
}

// ---- page-locked result blocks, cached (host_chain.h) ----
namespace {
struct ResultBlock { void *ptr; size_t bytes; bool in_use; };
std::mutex g_result_mu;
std::vector<ResultBlock> g_result_blocks;
constexpr size_t RESULT_CACHE_MIN = (size_t)8 << 20;       // smaller results are plain allocations
} // namespace

void *result_alloc_pinned(size_t bytes)
{
	if (bytes < RESULT_CACHE_MIN) return result_alloc(bytes);
	{
		std::lock_guard<std::mutex> lock(g_result_mu);
		ResultBlock *best = nullptr;
		for (ResultBlock &b : g_result_blocks)
			if (!b.in_use && b.bytes >= bytes && b.bytes / 4 <= bytes && (!best || b.bytes < best->bytes)) best = &b;
		if (best) { best->in_use = true; return best->ptr; }
	}
	const size_t want = bytes + bytes / 8;                      // the next batch of about this size fits too
	void *p = pin_by_register() ? pinned_alloc(want) : nullptr;
	if (!p) return result_alloc(bytes);
	std::lock_guard<std::mutex> lock(g_result_mu);
	g_result_blocks.push_back({ p, (want + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1), true });
	return p;
}

void result_release(void *ptr)
{
	if (!ptr) return;
	static const size_t cap = [] { const char *v = getenv("MM2GB_RESULT_CACHE_MB"); return (size_t)(v ? std::max(0LL, atoll(v)) : 8192) << 20; }();
	std::vector<void*> drop;
	{
		std::lock_guard<std::mutex> lock(g_result_mu);
		bool mine = false;
		size_t idle = 0;
		for (ResultBlock &b : g_result_blocks) { if (b.ptr == ptr) { b.in_use = false; mine = true; } if (!b.in_use) idle += b.bytes; }
		if (!mine) { free(ptr); return; }
		// over the cap: the largest idle blocks go first
		while (idle > cap) {
			size_t at = g_result_blocks.size();
			for (size_t k = 0; k < g_result_blocks.size(); ++k)
				if (!g_result_blocks[k].in_use && (at == g_result_blocks.size() || g_result_blocks[k].bytes > g_result_blocks[at].bytes)) at = k;
			if (at == g_result_blocks.size()) break;
			idle -= g_result_blocks[at].bytes;
			drop.push_back(g_result_blocks[at].ptr);
			g_result_blocks.erase(g_result_blocks.begin() + (std::ptrdiff_t)at);
		}
	}
	for (void *p : drop) pinned_free(p);
}

static DevParams make_params(const mm2gb_misc_t &m)
{
	DevParams P;
	P.max_dist_x = m.max_dist_x; P.max_dist_y = m.max_dist_y; P.bw = m.bw;
	if (P.max_dist_x < P.bw) P.max_dist_x = P.bw;                      // lchain.c:160
	if (P.max_dist_y < P.bw && !m.is_cdna) P.max_dist_y = P.bw;        // lchain.c:161
	P.max_iter = m.max_iter; P.n_seg = m.n_seg; P.is_cdna = m.is_cdna;
	P.dq_lim = std::min(P.max_dist_x, P.max_dist_y);
	P.lut_last = P.bw + 1; P.lut_base = LUT_LDS_TOTAL - 4 * (P.lut_last + 1); P.lut_clamp = 1; P.free_sweep = 0; P.edge_prefix = 0;
	P.gap = m.chn_pen_gap; P.skip = m.chn_pen_skip;
	return P;
}

int Engine::set_misc(const mm2gb_misc_t *m)
{
	if (!m) return fail("mm2gb: null misc");
	if (m->max_iter < 0 || m->bw < 0 || m->max_dist_x < 0 || m->max_dist_y < 0) return fail("mm2gb: negative chaining parameter");
	if (misc_valid && memcmp(&misc, m, sizeof(misc)) == 0) return 0;      // params / table / LDS split stay as configured
	misc = *m;
	misc_valid = true;
	params = make_params(misc);
	if (!stream) return 0;
	return configure_score();
}

// What the unclamped table sweeps of k_score take for granted about the hardware (chain_kernels.hip, lut_address / sweep_block_lut2_free),
// checked on the device itself before they are used: an LDS read beyond the workgroup's allocation returns 0, the allocation of
// LUT_LDS_TOTAL bytes ends exactly where the penalty table ends (no rounded-up tail with stale contents), and v_sad_u32 with the integer
// clamp saturates instead of wrapping.  Every workgroup fills all of its LDS with a pattern, then reads at the addresses the sweep
// would form for: dq <= 0 (two sizes), a distance beyond the table, the table's last entry, the first word after it.
__global__ __launch_bounds__(1024) void k_probe_lds_contract(int *bad)
{
	extern __shared__ unsigned probe_lds[];
	for (unsigned k = threadIdx.x; k < (unsigned)LUT_LDS_TOTAL / 4; k += blockDim.x) probe_lds[k] = 0x7f7f7f7fu;
	__syncthreads();
	if (threadIdx.x < 5) {
		const int a[5] = { 4 * 10, 0, 4 * 100, 4 * 100, 4 * 100 };
		const int b[5] = { -4, -4 * 20000, 4 * 6000, 4 * (100 + LUT_ENTRIES - 1), 4 * (100 + LUT_ENTRIES) };
		unsigned addr, v;
		asm volatile("v_sad_u32 %0, %1, %2, %3 clamp" : "=v"(addr) : "v"(a[threadIdx.x]), "v"(b[threadIdx.x]), "s"((unsigned)LUT_LDS_BASE));
		asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
		const unsigned want = threadIdx.x == 3 ? 0x7f7f7f7fu : 0u;
		if (v != want) atomicOr(bad, 1 << threadIdx.x);
	}
}

static bool probe_lds_contract(int n_cu, hipStream_t s)
{
	int *d_bad = nullptr, bad = -1;
	if (hipMalloc(&d_bad, sizeof(int)) != hipSuccess) return false;
	bool ok = hipMemsetAsync(d_bad, 0, sizeof(int), s) == hipSuccess &&
	          hipFuncSetAttribute((const void*)k_probe_lds_contract, hipFuncAttributeMaxDynamicSharedMemorySize, LUT_LDS_TOTAL) == hipSuccess;
	if (ok) {
		hipLaunchKernelGGL(k_probe_lds_contract, dim3((unsigned)n_cu * 4), dim3(1024), LUT_LDS_TOTAL, s, d_bad);   // two resident per CU, twice over
		ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
	}
	(void)hipFree(d_bad);
	if (ok && bad != 0) fprintf(stderr, "[mm2gb] this device does not read 0 beyond a workgroup's LDS / saturate v_sad_u32 clamp (probe mask %#x): "
	                                    "the score kernel keeps its range tests and the clamped penalty table\n", bad);
	return ok && bad == 0;
}

// Pick the scoring build and the LDS budget for these parameters; (re)build the penalty table on the device.
int Engine::configure_score()
{
	MM2GB_HIP(hipSetDevice(device));                    // kernel attributes and the table build below belong to this engine's device
	const bool single = !params.is_cdna && params.n_seg == 1;
	constexpr int LUT_MAX = LUT_ENTRIES;                // entries (chain_dev.h: the table's place in LDS is fixed); bw above this falls back to per-pair arithmetic
	constexpr int LUT_MAX_DIST = 1 << 28;
	if (!single) launch.host_mode = SCORE_MODE_GENERAL;
	// the table sweep keeps coordinates x4 (and compares (unsigned)dq_lim << 2): exact only while every distance a pair can pass
	// the range tests with stays below 2^28; a larger max_dist (user -g / -r) runs the per-pair build
	else if (params.skip == 0.0f && params.lut_last + 1 <= LUT_MAX && params.max_dist_x < LUT_MAX_DIST && params.max_dist_y < LUT_MAX_DIST) launch.host_mode = SCORE_MODE_LUT;
	else launch.host_mode = SCORE_MODE_FAST;
	constexpr size_t LDS_BUDGET = 80 * 1024 - 256;      // two 1024-thread workgroups per CU (160 KB LDS); MODE_LUT takes LUT_LDS_TOTAL
	// Team modes: a team's share of the LDS ring holds the scores of the most recent tiles of its chunk and must cover the
	// chunk's widest predecessor window plus the tile being written (64 scores per slot).  The ring takes whatever the
	// LDS budget of two workgroups per CU leaves; the planner sends a chunk to a team only if its widest window fits
	// that team's share.
	launch.big_team = 8;
	if (const char *v = getenv("MM2GB_BIG_TEAM")) launch.big_team = atoi(v) == 16 ? 16 : 8;
	launch.whole_wg_pct = 100;
	if (const char *v = getenv("MM2GB_WHOLE_WG_PCT")) launch.whole_wg_pct = std::max(0, atoi(v));
	const int n_big = 16 / launch.big_team;
	auto fit_slots = [&](const DevParams &prm) {
		int64_t slots = 1024;                                          // 64 K scores: more than any budget
		while (slots > 0 && score_lds_bytes(prm, launch.host_mode, (int)slots) > LDS_BUDGET) slots -= 4;   // four small teams share it evenly
		return slots;
	};
	// Penalty table: bw + 1 entries and a rejecting one, ending where the workgroup's LDS ends (chain_dev.h).  On a device that reads 0
	// beyond a workgroup's LDS (probed in init) the index is not clamped -- a distance beyond bw is an address beyond LDS -- and source
	// blocks far enough inside a window are swept without any range test (MM2GB_FREE_SWEEP=0 turns that off, for A/B runs).
	params.lut_last = params.bw + 1; params.lut_base = LUT_LDS_TOTAL - 4 * (params.lut_last + 1); params.lut_clamp = 1; params.free_sweep = 0; params.edge_prefix = 0;
	if (launch.host_mode == SCORE_MODE_LUT && lds_contract_ok && !getenv("MM2GB_LUT_CLAMP")) {
		params.lut_clamp = 0;
		{ const char *e = getenv("MM2GB_EDGE"); params.edge_prefix = e && !strcmp(e, "new"); }   // MM2GB_EDGE=new: the window test of edge blocks from a scalar prefix mask (measured: +1 % on 10-30 kb reads, -1 % on 30-100 kb: profiles/r06_narrow_ab.txt; off)
		const char *v = getenv("MM2GB_FREE_SWEEP");
		params.free_sweep = !(v && atoi(v) == 0) && params.dq_lim > 2 * params.bw;
	}
	(void)n_big;
	int64_t slots = fit_slots(params);
	if (slots < 8) slots = 0;                                        // less than two tiles of window per small team: not worth it
	launch.ring_slots = coop_disabled ? 0 : (int)slots;
	// the limit is an attribute of the kernel, shared by every engine of the process: always the whole budget, so that engines
	// with different parameters cannot lower it under each other's feet
	if (score_set_lds_limit(LDS_BUDGET)) return fail("mm2gb: cannot raise the dynamic LDS limit of the score kernel");
	if (launch.host_mode == SCORE_MODE_LUT) {
		// the table is shared by both compute streams: nothing that reads the old one may be in flight, and the new one must be
		// complete before either stream launches again (parameters change between runs, not between batches)
		for (WorkSet &w : work) MM2GB_HIP(hipStreamSynchronize(w.stream));
		if (lut.ensure((size_t)LUT_ENTRIES * 4)) return -1;
		launch_build_lut((int*)lut.ptr, params, stream);
		MM2GB_HIP(hipGetLastError());
		MM2GB_HIP(hipStreamSynchronize(stream));
	}
	return 0;
}

constexpr int MAX_COUNTED_DEVICES = 64;
static std::atomic<int> g_engines_on_device[MAX_COUNTED_DEVICES];      // engines alive per device (Engine::init: how many HIP streams an engine gets)

int Engine::init(const mm2gb_config_t *c, const mm2gb_misc_t *m, int dev)
{
	// An engine has up to four HIP streams (H2D, two compute, D2H) whose whole point is to run at the same time.  The HIP runtime
	// multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default, and streams that share a queue run one after the
	// other: next to a framework's own streams the engine's copies and kernels then serialise (200 M anchors through
	// mm2gb_score_host: 146 ms instead of 92 ms, profiles/README.md).  Only effective before the runtime starts, so hosts
	// export it themselves before their first HIP call; bench.py and tests/conftest.py do.
	// The library does not touch the environment here (setenv races with getenv in threaded hosts, and is too late once the runtime
	// runs): it says so once when the variable is missing.  INTEGRATION.md documents it; init_stream_gpu, the one entry point that IS
	// the process's first HIP call and runs before the host starts threads, still exports it for the C host.
	{
		static std::once_flag warned;
		const char *q = getenv("GPU_MAX_HW_QUEUES");
		if (!q || atoi(q) < 8)
			std::call_once(warned, [] { fprintf(stderr, "[mm2gb] GPU_MAX_HW_QUEUES is unset or below 8: the engine's copy and compute streams may share a hardware queue and "
			                                            "serialise (export GPU_MAX_HW_QUEUES=8 before the process starts HIP)\n"); });
	}
	int n_dev = 0;
	MM2GB_HIP(hipGetDeviceCount(&n_dev));
	if (dev < 0 || dev >= n_dev) return fail("mm2gb: device " + std::to_string(dev) + " not present (" + std::to_string(n_dev) + " visible)");
	device = dev;
	if (dev >= 0 && dev < 64) g_devices_with_engines.fetch_or(1ull << dev);
	MM2GB_HIP(hipSetDevice(device));
	hipDeviceProp_t prop;
	MM2GB_HIP(hipGetDeviceProperties(&prop, device));
	n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	cfg = *c;
	// persistent grid: short_griddim counts 256-thread workgroups in the reference schema; k_score uses 1024-thread ones
	launch.score_grid = cfg.score_kernel.short_griddim > 0 ? std::max(1, cfg.score_kernel.short_griddim / 4) : n_cu * 2;
	// Chunks whose DP would keep a single wave busy for long are pipelined over a whole workgroup (cooperative mode).
	// A chunk with mean window W can keep about (W + 128) / 128 waves busy, so the mode pays from a few blocks of window.
	// long_seg_cutoff / mid_seg_cutoff keep their reference meaning of "cut units" (range_kernel.blockdim anchors,
	// plscore.cu:330,378): cost threshold = mid_seg_cutoff units of anchors at the minimum window.
	launch.long_min_window = 128;
	launch.long_min_cost = (int64_t)std::max(1, cfg.score_kernel.mid_seg_cutoff) * std::max(64, cfg.range_kernel.blockdim) * 1024;
	launch.wide_window = 1024;
	if (const char *v = getenv("MM2GB_LONG_MIN_WINDOW")) launch.long_min_window = std::max(1, atoi(v));
	if (const char *v = getenv("MM2GB_WIDE_WINDOW")) launch.wide_window = std::max(1, atoi(v));
	if (const char *v = getenv("MM2GB_LONG_MIN_COST")) launch.long_min_cost = std::max<int64_t>(1, atoll(v));
	// Wide-window heavy chunks on 4-wave teams (chain_kernels.hip, plan_finish): pays once a micro-batch keeps the machine full for
	// long -- 500 M anchors: 59.3 -> 47.6 ms, 300 M: 39.2 -> 36.2 -- and costs a few per cent where the largest chunks decide when the
	// batch ends (200 M: 27.1 -> 28.5 ms), so it is tied to the batch size (profiles/earlier/r02w_ab.txt)
	launch.team4_all = 0; launch.team4_share_pct = 60; team4_min_n = 250000000;
	if (const char *v = getenv("MM2GB_TEAM4_ALL")) launch.team4_all = atoi(v) != 0;
	if (const char *v = getenv("MM2GB_TEAM4_SHARE_PCT")) launch.team4_share_pct = std::max(0, atoi(v));
	if (const char *v = getenv("MM2GB_TEAM4_MIN_ANCHORS")) team4_min_n = std::max<int64_t>(0, atoll(v));
	if (const char *v = getenv("MM2GB_DEBUG_PHASES")) debug_phases = *v && *v != '0';
	if (const char *v = getenv("MM2GB_POST_FORM")) post_split = strcmp(v, "fused") != 0;
	if (const char *v = getenv("MM2GB_POST_SORT")) post_levels = strcmp(v, "reads") != 0;
	// Gangs: a chunk whose share of the batch's pairs is worth two workgroups or more is scored by several (chain_kernels.hip, plan_gangs):
	// batches that cannot fill the machine end with their largest chunks.  MM2GB_GANG_MAX=0 turns them off.
	// Large micro-batches keep the kernel without the gang code (MM2GB_GANG_MAX_ANCHORS: the largest batch that gets gangs).
	launch.gang_max = 8; launch.gang_pct = 150; gang_max_n = 150000000;   // (150 %: a gang's workgroups wait for the chain part of the time; 20 M anchors 7.1 -> 6.6 ms)
	if (const char *v = getenv("MM2GB_GANG_MAX")) launch.gang_max = std::max(0, std::min(64, atoi(v)));
	if (const char *v = getenv("MM2GB_GANG_PCT")) launch.gang_pct = std::max(1, atoi(v));
	launch.gang_pairs = 0;
	if (const char *v = getenv("MM2GB_GANG_PAIRS")) launch.gang_pairs = *v && *v != '0';
	if (const char *v = getenv("MM2GB_GANG_MAX_ANCHORS")) gang_max_n = std::max<int64_t>(0, atoll(v));
	split_max_n = 0;                                           // off: measured slower at every batch size (profiles/earlier/r02y_split_rate.json)
	if (const char *v = getenv("MM2GB_SPLIT_MAX_ANCHORS")) split_max_n = std::max<int64_t>(0, atoll(v));
	if (debug_phases && dbg.ensure((size_t)launch.score_grid * 32)) return -1;
	const char *env = getenv("MM2GB_NO_COOP");
	coop_disabled = env && *env && *env != '0';
	// Four HIP streams -- H2D, two compute, D2H -- let ONE engine overlap its copies with its kernels (the host-buffer paths).  Where several
	// engines share a device they overlap each other, and every stream more is one more thing the runtime multiplexes onto the hardware queues:
	// the drop-in at -t 16 (32 engines) maps 1.05 Gbp in 9.2-10.3 s with two streams per engine against 10.7-10.9 s with four, the own host's four
	// engines 4.3-4.5 s against 4.6-4.8 s (profiles/r05_hw_queues.txt).  So the first engine alive on a device gets four, every further one two:
	// kernels and H2D on one, D2H on the other.  MM2GB_LEAN_STREAMS=0 / 1 forces either.
	{
		const int before = dev >= 0 && dev < MAX_COUNTED_DEVICES ? g_engines_on_device[dev].fetch_add(1) : 0;
		counted_on_device = dev >= 0 && dev < MAX_COUNTED_DEVICES;
		const char *v = getenv("MM2GB_LEAN_STREAMS");
		lean_streams = v && *v ? *v != '0' : before > 0;
	}
	MM2GB_HIP(hipStreamCreateWithFlags(&work[0].stream, hipStreamNonBlocking));
	if (lean_streams) work[1].stream = work[0].stream; else MM2GB_HIP(hipStreamCreateWithFlags(&work[1].stream, hipStreamNonBlocking));
	stream = work[0].stream;
	{ const char *v = getenv("MM2GB_ONE_COMPUTE_STREAM"); one_compute_stream = v && *v && *v != '0'; }
	if (const char *v = getenv("MM2GB_DUAL_STREAM_MAX")) dual_stream_max_n = std::max<int64_t>(0, atoll(v));
	if (lean_streams) s_in = work[0].stream; else MM2GB_HIP(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
	MM2GB_HIP(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
	for (IoSet &s : io)
		for (hipEvent_t *e : { &s.in_start, &s.in_done, &s.comp_done, &s.out_start, &s.out_done }) MM2GB_HIP(hipEventCreate(e));
	{
		// the engine's small page-locked read-back areas out of ONE registered block (hipHostMalloc is a process-wide lock: sixteen engines made at
		// the same time spent most of their 40 ms each queueing for three of these)
		const size_t need = (size_t)MAX_SLOTS * CNT_WORDS * sizeof(int32_t) + (size_t)MAX_SLOTS * 2 * sizeof(int64_t) + 2 * 64;
		h_small = (char*)pinned_alloc(need);
		if (!h_small) return fail("mm2gb: cannot allocate the engine's page-locked read-back block");
		h_counters = (int32_t*)h_small;
		h_totals = (int64_t*)(h_small + (size_t)MAX_SLOTS * CNT_WORDS * sizeof(int32_t));
		post_out[0].h_totals = (int64_t*)((char*)h_totals + (size_t)MAX_SLOTS * 2 * sizeof(int64_t));
		post_out[1].h_totals = post_out[0].h_totals + 8;
	}
	for (WorkSet &w : work) if (w.counters.ensure(CNT_WORDS * sizeof(int32_t)) || w.totals.ensure(4 * sizeof(int64_t)) || w.flags.ensure(4 * sizeof(unsigned))) return -1;
	{
		// once per device and process (MM2GB_LDS_PROBE=0: take the contract as broken, i.e. clamped table and every range test)
		static std::mutex probe_lock;
		static int probed[64];                               // 0 not yet, 1 holds, 2 does not
		std::lock_guard<std::mutex> g(probe_lock);
		const char *v = getenv("MM2GB_LDS_PROBE");
		int &state = probed[device & 63];
		if (v && atoi(v) == 0) lds_contract_ok = false;
		else { if (!state) state = probe_lds_contract(n_cu, stream) ? 1 : 2; lds_contract_ok = state == 1; }
	}
	if (set_misc(m)) return -1;
	return 0;
}

void Engine::shutdown()
{
	if (counted_on_device) { g_engines_on_device[device].fetch_sub(1); counted_on_device = false; }
	(void)hipSetDevice(device);
	for (hipStream_t s : { s_in, work[0].stream, work[1].stream, s_out }) if (s) (void)hipStreamSynchronize(s);
	flush_retired_buffers();
	for (WorkSet &w : work) { for (DevBuf *b : w.all()) b->release(); w.cap_n = w.cap_reads = w.cap_blocks = 0; }
	lut.release(); dbg.release();
	for (IoSet &s : io) {
		s.raw.release(); s.offsets.release(); s.f.release(); s.p.release();
		for (hipEvent_t *e : { &s.in_start, &s.in_done, &s.comp_done, &s.out_start, &s.out_done }) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
		s.used = false;
	}
	for (BatchSlot &b : slots)
		for (hipEvent_t *e : { &b.prep0, &b.prep1, &b.score1 }) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
	h_slice_off.release(); h_res_f.release(); h_res_p.release();
	for (hipEvent_t &e : slice_in) if (e) (void)hipEventDestroy(e);
	slice_in.clear();
	for (DevBuf *b : { &post_dbg_reads, &post_dbg_tasks, &post_dbg_stasks, &rmq_dbg_reads, &rmq_skey_in, &rmq_skey, &rmq_sa, &rmq_srange, &rmq_sort_tmp, &post_z, &post_fp, &post_picked, &post_utmp, &post_heads, &post_nu, &post_nkept, &post_misc, &post_bins, &post_order, &post_up4, &post_up16, &post_sort_s, &post_sort_perm, &post_sort_tmp, &post_cls, &post_cls_cnt, &post_cls_nz, &post_read_nz, &post_uloc, &post_wtask, &post_stask, &rmq_tied, &rmq_sum, &rmq_by_y, &rmq_ord, &rmq_meta, &rmq_win, &rmq_tree, &reg_out,
	                   &sd_seeds, &sd_seed_off, &sd_hit_off, &sd_hits, &sd_qlen, &sd_q_rank, &sd_ref_len, &sd_ref_rank, &sd_seed_read, &sd_tmp, &sd_n_kept, &sd_a_off, &sd_out,
	                   &post_out[0].u_off, &post_out[0].a_off, &post_out[0].u_out, &post_out[0].a_out, &post_out[1].u_off, &post_out[1].a_off, &post_out[1].u_out, &post_out[1].a_out })
		b->release();
	cap_post_n = cap_post_reads = 0;
	for (PostOut &po : post_out) {
		po.h_totals = nullptr;
		if (po.done) { (void)hipEventDestroy(po.done); po.done = nullptr; }
	}
	for (hipEvent_t *e : { &post0, &post1, &post_fork, &post_join, &rmq_fill_done }) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
	pinned_free(h_small);
	h_small = nullptr;
	if (lean_streams) { s_in = nullptr; work[1].stream = nullptr; }
	for (hipStream_t *s : { &s_in, &work[0].stream, &work[1].stream, &s_out }) if (*s) { (void)hipStreamDestroy(*s); *s = nullptr; }
	stream = nullptr;
	h_counters = nullptr; h_totals = nullptr;
}

int Engine::reserve(int64_t n, int64_t n_reads, int set)
{
	MM2GB_HIP(hipSetDevice(device));
	if (n >= ((int64_t)1 << 31) - 2 * PLAN_BLOCK) return fail("mm2gb: a micro-batch is limited to 2^31 anchors (got " + std::to_string(n) + ")");
	WorkSet &w = work[set];
	if (n > w.cap_n || n_reads > w.cap_reads) {
		// wait for anything in flight on this set before its arenas move
		MM2GB_HIP(hipStreamSynchronize(w.stream));
		const int64_t nn = std::max<int64_t>(std::max(n, w.cap_n), 1024);
		const int64_t nb = (nn + PLAN_BLOCK - 1) / PLAN_BLOCK + 1;
		// A growth that fails part-way leaves some arenas grown, some as they were and possibly one empty: the recorded capacity
		// is dropped first and only restored when every arena has its size, so the next call reserves again instead of launching
		// on a short (or null) buffer.
		const int64_t had_reads = w.cap_reads;
		w.cap_n = w.cap_blocks = w.cap_reads = 0;
		if (w.st.ensure(nn * 4)) return -1;
		if (w.blk_firstcut.ensure(nb * 4) || w.blk_pairs.ensure(nb * 8) || w.blk_clamped.ensure(nb * 4) || w.blk_wmax.ensure(nb * 8) || w.blk_read.ensure(nb * 4)) return -1;
		if (w.chunk_start.ensure(nb * 4) || w.chunk_end.ensure(nb * 4) || w.chunk_cost.ensure(nb * 8) || w.chunk_track.ensure(nb) ||
		    w.order.ensure(nb * 4) || w.long_list.ensure(nb * 4) || w.mid_list.ensure(nb * 4) || w.chunk_pp.ensure(nb * 8) || w.chunk_kk.ensure(nb * 4) || w.chunk_blk.ensure(nb * 4) ||
		    w.tile_sums.ensure((nb / 1024 + 2) * 24) || w.tile_base.ensure((nb / 1024 + 2) * 24) || w.bins.ensure(3 * PLAN_COST_BINS * 4)) return -1;
		w.cap_n = nn; w.cap_blocks = nb; w.cap_reads = std::max(had_reads, n_reads);
	}
	return 0;
}

int Engine::begin_call()
{
	if (n_slots > 0 && sync()) return -1;     // a previous call was never collected
	n_slots = 0;
	last = mm2gb_stats_t();
	last_split_chunks = last_helped_items = 0; last_gang_chunks = last_gang_wgs = 0;
	return 0;
}

int Engine::enqueue(int64_t n_reads, const int64_t *d_offsets, const mm2gb_anchor_t *d_anchors, int64_t n, int32_t *d_f, int32_t *d_p, bool want_stats, int set)
{
	if (n < 0 || n_reads < 0) return fail("mm2gb: negative batch size");
	if (want_stats && n_slots >= MAX_SLOTS && sync()) return -1;      // fold what is done so far into `last`, keep counting
	if (reserve(n, n_reads, set)) return -1;
	WorkSet &w = work[set];
	hipStream_t stream = w.stream;                                    // (shadows the member: everything below belongs to this set)
	DevBuf &st = w.st, &blk_firstcut = w.blk_firstcut, &blk_pairs = w.blk_pairs, &blk_clamped = w.blk_clamped, &blk_wmax = w.blk_wmax,
	       &blk_read = w.blk_read, &chunk_start = w.chunk_start, &chunk_end = w.chunk_end, &chunk_cost = w.chunk_cost, &chunk_track = w.chunk_track, &order = w.order,
	       &long_list = w.long_list, &mid_list = w.mid_list, &chunk_pp = w.chunk_pp, &chunk_kk = w.chunk_kk, &chunk_blk = w.chunk_blk, &tile_sums = w.tile_sums,
	       &tile_base = w.tile_base, &bins = w.bins, &counters = w.counters, &totals = w.totals, &flags = w.flags;
	const int slot = want_stats ? n_slots++ : 0;
	BatchSlot &bs = slots[slot];
	if (want_stats) for (hipEvent_t *e : { &bs.prep0, &bs.prep1, &bs.score1 }) if (!*e) MM2GB_HIP(hipEventCreate(e));
	DevBatch b;
	b.raw = (const uint4*)d_anchors; b.offsets = d_offsets; b.n = n; b.n_reads = n_reads;
	b.st = (int32_t*)st.ptr;
	b.f = d_f; b.p = d_p;
	b.blk_firstcut = (int32_t*)blk_firstcut.ptr; b.blk_pairs = (int64_t*)blk_pairs.ptr; b.blk_clamped = (int32_t*)blk_clamped.ptr; b.blk_wmax = (int32_t*)blk_wmax.ptr; b.blk_read = (int32_t*)blk_read.ptr;
	b.n_blocks = (n + PLAN_BLOCK - 1) / PLAN_BLOCK;
	b.chunk_start = (int32_t*)chunk_start.ptr; b.chunk_end = (int32_t*)chunk_end.ptr; b.chunk_cost = (int64_t*)chunk_cost.ptr;
	b.chunk_track = (uint8_t*)chunk_track.ptr; b.order = (int32_t*)order.ptr; b.long_list = (int32_t*)long_list.ptr; b.mid_list = (int32_t*)mid_list.ptr;
	b.chunk_pp = (int64_t*)chunk_pp.ptr; b.chunk_kk = (int32_t*)chunk_kk.ptr; b.chunk_blk = (int32_t*)chunk_blk.ptr;
	b.tile_sums = (int64_t*)tile_sums.ptr; b.tile_base = (int64_t*)tile_base.ptr; b.bins = (int32_t*)bins.ptr;
	b.counters = (int32_t*)counters.ptr; b.totals = (int64_t*)totals.ptr; b.flags = (unsigned*)flags.ptr;
	b.lut = (const int32_t*)lut.ptr;
	b.dbg = debug_phases ? (int64_t*)dbg.ptr : nullptr;
	if (debug_phases) MM2GB_HIP(hipMemsetAsync(dbg.ptr, 0, (size_t)launch.score_grid * 32, stream));
	// A batch that cannot fill the machine ends with its largest chunks: the SPLIT build scores those strip by strip with the help of
	// the workgroups that have run out of work (chain_kernels.hip, split_chunk).  Large batches keep the plain build.
	LaunchCfg cfg_now = launch;
	if (n < team4_min_n) cfg_now.team4_share_pct = 0;
	cfg_now.split = score_has_split_build() && launch.host_mode == SCORE_MODE_LUT && launch.ring_slots > 0 && n > 0 && n <= split_max_n;
	b.split_slots = nullptr; b.split_part = nullptr;
	if (cfg_now.split) {
		if (w.split_slots.ensure((size_t)launch.score_grid * sizeof(SplitSlot)) || w.split_part.ensure((size_t)launch.score_grid * SPLIT_MAX_ITEMS * 2 * 64 * 8)) return -1;
		b.split_slots = (SplitSlot*)w.split_slots.ptr; b.split_part = (unsigned long long*)w.split_part.ptr;
		MM2GB_HIP(hipMemsetAsync(w.split_slots.ptr, 0, (size_t)launch.score_grid * sizeof(SplitSlot), stream));
	}

	b.gang_slots = nullptr;
	if (launch.gang_max >= 2 && launch.host_mode == SCORE_MODE_LUT && launch.ring_slots > 0 && !cfg_now.split && n > 0 && n <= gang_max_n) {
		if (w.gang_slots.ensure((size_t)GANG_MAX_CHUNKS * sizeof(GangSlot))) return -1;
		b.gang_slots = (GangSlot*)w.gang_slots.ptr;                  // (filled by plan_gangs; CNT_NGANG says how many)
	}

	MM2GB_HIP(hipMemsetAsync(counters.ptr, 0, CNT_WORDS * sizeof(int32_t), stream));
	MM2GB_HIP(hipMemsetAsync(totals.ptr, 0, 4 * sizeof(int64_t), stream));
	MM2GB_HIP(hipMemsetAsync(flags.ptr, 0, 4 * sizeof(unsigned), stream));
	if (want_stats) MM2GB_HIP(hipEventRecord(bs.prep0, stream));
	if (n > 0) {
		launch_window(b, params, stream);
		launch_plan(b, cfg_now, stream);
	}
	if (want_stats) MM2GB_HIP(hipEventRecord(bs.prep1, stream));
	if (n > 0) launch_score(b, params, cfg_now, stream);
	if (want_stats) {
		MM2GB_HIP(hipEventRecord(bs.score1, stream));
		MM2GB_HIP(hipMemcpyAsync(h_counters + (size_t)slot * CNT_WORDS, counters.ptr, CNT_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
		MM2GB_HIP(hipMemcpyAsync(h_totals + (size_t)slot * 2, totals.ptr, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
		last.n_anchors += n; last.n_reads += n_reads;
	}
	MM2GB_HIP(hipGetLastError());
	return 0;
}

int Engine::reserve_post(int64_t n, int64_t n_reads)
{
	MM2GB_HIP(hipSetDevice(device));
	for (PostOut &po : post_out) {
		if (!po.h_totals) return fail("mm2gb: engine not initialised");
		if (!po.done) MM2GB_HIP(hipEventCreateWithFlags(&po.done, hipEventDisableTiming));
	}
	for (hipEvent_t *e : { &post0, &post1 }) if (!*e) MM2GB_HIP(hipEventCreate(e));
	for (hipEvent_t *e : { &post_fork, &post_join }) if (!*e) MM2GB_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
	if (n <= cap_post_n && n_reads <= cap_post_reads) return 0;
	// Only the WORK arrays of the post kernels live here: they are dead once the kernels of the batch that used them have run, so
	// waiting for the compute stream is all it takes to move them.  What a batch leaves for the host (PostOut) is sized by
	// reserve_post_out for the one set that is about to be written: the other set may hold a batch that has not been fetched yet
	// (the boundary launches batch k+1 before it fetches batch k), and a buffer that grows is a new, uninitialised buffer.
	MM2GB_HIP(hipStreamSynchronize(stream));
	const int64_t nn = std::max<int64_t>(std::max(n, cap_post_n), 1024), nr = std::max<int64_t>(std::max(n_reads, cap_post_reads), 16);
	cap_post_n = cap_post_reads = 0;                    // as in reserve(): only restored when every buffer has its size
	// per-chain arrays are sized for min_cnt = 1 (a chain per anchor): min_cnt is a per-call parameter and may drop
	const size_t chains = (size_t)(nn + nr);
	if (post_sort_s.ensure((size_t)nn + 16384) || post_sort_perm.ensure((size_t)nn * 4) || post_sort_tmp.ensure((size_t)nn * 8)) return -1;   // radix_pass_bytes: 13 B per anchor
	if (post_z.ensure((size_t)nn * 8) || post_fp.ensure((size_t)nn * 8) || post_picked.ensure((size_t)nn * 4) || post_utmp.ensure(chains * 8) ||
	    post_heads.ensure(chains * 16) || post_nu.ensure((size_t)nr * 4) || post_nkept.ensure((size_t)nr * 4) || post_misc.ensure(2048) || post_bins.ensure(2 * N_SIZE_CLASSES * 4) || post_order.ensure((size_t)nr * 4) ||
	    post_up4.ensure((size_t)nn * 4) || post_up16.ensure((size_t)nn * 4)) return -1;
	// split form: a class per anchor, per read and class the anchors / candidates, per chain slot where its anchors are, the walk tasks and their order
	if (post_cls.ensure((size_t)nn) || post_cls_cnt.ensure((size_t)nr * N_TREE_CLASSES * 4) || post_cls_nz.ensure((size_t)nr * N_TREE_CLASSES * 4) || post_read_nz.ensure((size_t)nr * 4) ||
	    post_uloc.ensure(chains * 4) || post_wtask.ensure((size_t)nr * N_TREE_CLASSES * 4 * 2)) return -1;
	// the sort's tasks: runs of more than 64 candidates, at most n / 65 of them at a level (+ a read's first): two lists and an order
	if (post_stask.ensure(((size_t)nn / 64 + (size_t)nr + 64) * (16 + 16 + 4))) return -1;
	cap_post_n = nn; cap_post_reads = nr;
	return 0;
}

// Result buffers of ONE post set, for a batch of n anchors / n_reads reads that is about to be written into it.  The caller's
// contract (boundary: stage k <-> set k, a stage is finished before it is launched again) is that the set's previous results have
// been fetched; the other set is never touched.  Growth waits for the compute stream (kernels that wrote the set) and the D2H
// stream (a fetch of it that is still copying).
int Engine::reserve_post_out(int set, int64_t n, int64_t n_reads)
{
	PostOut &po = post_out[set];
	const int64_t nn = std::max<int64_t>(n, 1024), nr = std::max<int64_t>(n_reads, 16);
	const size_t need_off = (size_t)(nr + 1) * 8, need_u = (size_t)(nn + nr) * 8, need_a = (size_t)nn * 16;
	if (po.u_off.bytes >= need_off && po.a_off.bytes >= need_off && po.u_out.bytes >= need_u && po.a_out.bytes >= need_a) return 0;
	MM2GB_HIP(hipStreamSynchronize(stream));
	MM2GB_HIP(hipStreamSynchronize(s_out));
	if (po.u_off.ensure(need_off) || po.a_off.ensure(need_off) || po.u_out.ensure(need_u) || po.a_out.ensure(need_a)) return -1;
	return 0;
}

int Engine::enqueue_post(int64_t n_reads, const int64_t *d_offsets, const mm2gb_anchor_t *d_anchors, int64_t n, const int32_t *d_f, const int32_t *d_p,
                         const mm2gb_rmq_param_t *rmq, int out_set)
{
	if (out_set < 0 || out_set > 1) return fail("mm2gb: post set out of range");
	if (reserve_post(n, n_reads) || reserve_post_out(out_set, n, n_reads)) return -1;
	PostOut &po = post_out[out_set];
	PostBatch b;
	b.raw = (const uint4*)d_anchors; b.offsets = d_offsets; b.n = n; b.n_reads = n_reads; b.f = d_f; b.p = d_p;
	b.z = (unsigned long long*)post_z.ptr; b.fp = (int2*)post_fp.ptr; b.picked = (int32_t*)post_picked.ptr;
	b.up4 = (int32_t*)post_up4.ptr; b.up16 = (int32_t*)post_up16.ptr;
	b.sort_s = (unsigned char*)post_sort_s.ptr; b.sort_perm = (int32_t*)post_sort_perm.ptr; b.sort_tmp = (unsigned long long*)post_sort_tmp.ptr;
	b.u_tmp = (unsigned long long*)post_utmp.ptr; b.heads = (ulonglong2*)post_heads.ptr;
	b.n_u = (int32_t*)post_nu.ptr; b.n_kept = (int32_t*)post_nkept.ptr; b.u_off = (int64_t*)po.u_off.ptr; b.a_off = (int64_t*)po.a_off.ptr;
	b.u_out = (unsigned long long*)po.u_out.ptr; b.a_out = (uint4*)po.a_out.ptr;
	b.totals = (int64_t*)post_misc.ptr; b.cursor = (int32_t*)((char*)post_misc.ptr + 16);
	b.order = (int32_t*)post_order.ptr; b.size_bins = (int32_t*)post_bins.ptr;
	b.cls = post_split ? (unsigned char*)post_cls.ptr : nullptr;
	b.cls_cnt = (int32_t*)post_cls_cnt.ptr; b.cls_nz = (int32_t*)post_cls_nz.ptr; b.read_nz = (int32_t*)post_read_nz.ptr;
	b.zc = (unsigned long long*)post_sort_tmp.ptr; b.kpos = (int32_t*)post_sort_perm.ptr; b.u_loc = (int32_t*)post_uloc.ptr;
	b.wtask = (int32_t*)post_wtask.ptr; b.wtask_order = b.wtask + (size_t)std::max<int64_t>(cap_post_reads, 16) * N_TREE_CLASSES;
	{
		const size_t cap = (size_t)cap_post_n / 64 + (size_t)cap_post_reads + 64;
		b.stask[0] = post_levels ? (int4*)post_stask.ptr : nullptr; b.stask[1] = b.stask[0] ? b.stask[0] + cap : nullptr;
		b.stask_order = b.stask[0] ? (int32_t*)(b.stask[0] + 2 * cap) : nullptr;
	}
	b.walk_grid_waves = n_cu * 4 * 8;
	if (const char *v = getenv("MM2GB_WALK_WAVES")) b.walk_grid_waves = std::max(4, atoi(v));
	b.sort_pairs = 1;
	if (const char *v = getenv("MM2GB_SORT_PAIRS")) b.sort_pairs = atoi(v) != 0;
	b.dbg = debug_phases ? (long long*)((char*)post_misc.ptr + 1024) : nullptr;
	if (debug_phases) { MM2GB_HIP(hipMemsetAsync((char*)post_misc.ptr + 1024, 0, 512, stream)); MM2GB_HIP(hipMemsetAsync((char*)post_misc.ptr + 1024 + 22 * 8, 0xff, 8, stream)); }   // ([22]: a minimum)
	b.dbg_reads = nullptr;
	if (debug_phases) {
		if (post_dbg_reads.ensure((size_t)std::max<int64_t>(n_reads, 1) * 32)) return -1;
		MM2GB_HIP(hipMemsetAsync(post_dbg_reads.ptr, 0, (size_t)std::max<int64_t>(n_reads, 1) * 32, stream));
		b.dbg_reads = (long long*)post_dbg_reads.ptr;
		if (post_dbg_tasks.ensure((size_t)std::max<int64_t>(n_reads, 1) * N_TREE_CLASSES * 64)) return -1;
		MM2GB_HIP(hipMemsetAsync(post_dbg_tasks.ptr, 0, (size_t)std::max<int64_t>(n_reads, 1) * N_TREE_CLASSES * 64, stream));
	}
	b.dbg_tasks = debug_phases ? (long long*)post_dbg_tasks.ptr : nullptr;
	if (debug_phases && post_dbg_stasks.ensure((size_t)262144 * 32)) return -1;
	b.dbg_stasks = debug_phases ? (long long*)post_dbg_stasks.ptr : nullptr;
	b.min_cnt = misc.min_cnt; b.min_sc = misc.min_score;
	b.max_drop = misc.is_cdna ? INT_MAX : misc.bw;                    // lchain.c:151,162
	if (rmq) { b.min_cnt = rmq->min_cnt; b.min_sc = rmq->min_sc; b.max_drop = rmq->bw; }   // lchain.c:253,355
	// one read per wave at a time: as many waves as the chip holds (latency-bound pointer chases; parallelism is across reads)
	b.grid_waves = n_cu * 32;
	if (const char *v = getenv("MM2GB_POST_WAVES")) b.grid_waves = std::max(4, atoi(v));
	// Whole workgroups start on the largest reads together (k_post_chains: collection and the buckets of the sort's top pass shared by the
	// four waves): every read of a batch that leaves the chip idle anyway (at most one read per workgroup the chip holds at a time, 3 per CU by
	// the kernel's LDS), the 64 largest of any other.  The kernel ends with its largest reads (sort + walks of the largest: 49 ms on one wave, 45
	// with a workgroup's start), and since it runs three waves per SIMD the helpers' wait for the serial top pass no longer costs what it saves:
	// k_post_chains + lift + emit at 500 M anchors / 9 016 reads 54.1-54.3 ms with no teams, 51.5-52.4 with 16 ... 128, 52.9 with 256, 55.1-55.7
	// with 640 or 1 500 (profiles/r03_post_teams.txt; at two waves per SIMD every setting lost).
	b.team_reads = n_reads <= (int64_t)n_cu * 3 ? (int)n_reads : 64;
	if (const char *v = getenv("MM2GB_POST_TEAM_READS")) b.team_reads = std::max(0, atoi(v));
	MM2GB_HIP(hipEventRecord(post0, stream));
	launch_post(b, stream, !lean_streams && work[1].stream && work[1].stream != stream ? work[1].stream : nullptr, post_fork, post_join);
	MM2GB_HIP(hipEventRecord(post1, stream));
	MM2GB_HIP(hipMemcpyAsync(po.h_totals, post_misc.ptr, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
	MM2GB_HIP(hipEventRecord(po.done, stream));
	MM2GB_HIP(hipGetLastError());
	return 0;
}

int Engine::enqueue_host_chains(int64_t n_reads, const int64_t *h_offsets, const mm2gb_anchor_t *h_anchors, int64_t n, int out_set, bool want_stats)
{
	MM2GB_HIP(hipSetDevice(device));
	IoSet &s = io[io_seq++ & 1];
	const size_t nn = (size_t)std::max<int64_t>(n, 1);
	if (s.raw.bytes < nn * 16 || s.f.bytes < nn * 4 || s.p.bytes < nn * 4 || s.offsets.bytes < (size_t)(n_reads + 1) * 8) {
		for (hipStream_t q : { s_in, work[0].stream, work[1].stream, s_out }) MM2GB_HIP(hipStreamSynchronize(q));
		if (s.raw.ensure(nn * 16) || s.f.ensure(nn * 4) || s.p.ensure(nn * 4) || s.offsets.ensure((size_t)(n_reads + 1) * 8)) return -1;
	}
	// always compute stream 0: the post-pass's work arrays exist once
	if (s.used) MM2GB_HIP(hipStreamWaitEvent(s_in, s.comp_done, 0));
	MM2GB_HIP(hipMemcpyAsync(s.offsets.ptr, h_offsets, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, s_in));
	if (n > 0) MM2GB_HIP(hipMemcpyAsync(s.raw.ptr, h_anchors, (size_t)n * 16, hipMemcpyHostToDevice, s_in));
	MM2GB_HIP(hipEventRecord(s.in_done, s_in));
	MM2GB_HIP(hipStreamWaitEvent(stream, s.in_done, 0));
	// the score kernels overwrite s.f / s.p: a batch that went through enqueue_host (several micro-batches) may still be copying
	// them out of this set on the D2H stream, and its kernels may have run on the other compute stream
	if (s.used) { MM2GB_HIP(hipStreamWaitEvent(stream, s.out_done, 0)); MM2GB_HIP(hipStreamWaitEvent(stream, s.comp_done, 0)); }
	if (enqueue(n_reads, (const int64_t*)s.offsets.ptr, (const mm2gb_anchor_t*)s.raw.ptr, n, (int32_t*)s.f.ptr, (int32_t*)s.p.ptr, want_stats, 0)) return -1;
	if (enqueue_post(n_reads, (const int64_t*)s.offsets.ptr, (const mm2gb_anchor_t*)s.raw.ptr, n, (const int32_t*)s.f.ptr, (const int32_t*)s.p.ptr, nullptr, out_set)) return -1;
	MM2GB_HIP(hipEventRecord(s.comp_done, stream));
	MM2GB_HIP(hipEventRecord(s.out_done, stream));                       // (no D2H of f / p on this path: "outputs done" == kernels done)
	s.used = true;
	return 0;
}

int Engine::fetch_chains(int out_set, int64_t n_reads, mm2gb_chains_t *out)
{
	memset(out, 0, sizeof(*out));
	MM2GB_HIP(hipSetDevice(device));
	PostOut &po = post_out[out_set];
	MM2GB_HIP(hipEventSynchronize(po.done));
	const int64_t n_u = n_reads > 0 ? po.h_totals[0] : 0, n_a = n_reads > 0 ? po.h_totals[1] : 0;
	out->u_off = (int64_t*)malloc((size_t)(n_reads + 1) * 8);
	out->a_off = (int64_t*)malloc((size_t)(n_reads + 1) * 8);
	out->u = (uint64_t*)malloc((size_t)(n_u + 1) * 8);
	out->a = (mm2gb_anchor_t*)result_alloc_pinned((size_t)(n_a + 1) * 16);
	if (!out->u_off || !out->a_off || !out->u || !out->a) { free(out->u_off); free(out->a_off); free(out->u); result_release(out->a); memset(out, 0, sizeof(*out)); return fail("mm2gb: out of host memory"); }
	out->u_off[0] = out->a_off[0] = 0;
	if (n_reads > 0) {
		// on the D2H stream: the compute stream may already be busy with the next batch
		MM2GB_HIP(hipMemcpyAsync(out->u_off, po.u_off.ptr, (size_t)(n_reads + 1) * 8, hipMemcpyDeviceToHost, s_out));
		MM2GB_HIP(hipMemcpyAsync(out->a_off, po.a_off.ptr, (size_t)(n_reads + 1) * 8, hipMemcpyDeviceToHost, s_out));
		if (n_u > 0) MM2GB_HIP(hipMemcpyAsync(out->u, po.u_out.ptr, (size_t)n_u * 8, hipMemcpyDeviceToHost, s_out));
		if (n_a > 0) MM2GB_HIP(hipMemcpyAsync(out->a, po.a_out.ptr, (size_t)n_a * 16, hipMemcpyDeviceToHost, s_out));
		MM2GB_HIP(hipStreamSynchronize(s_out));
	}
	return 0;
}

// The same for a LARGE batch: the anchors go in in slices of reads, each slice's score kernels running under the next slice's H2D (the
// link is never idle; round 4 copied everything in, then ran the kernels, then copied the chains into fresh pageable memory: 170 ms for 200 M
// anchors against 55.6 ms of H2D alone).  The post-pass runs ONCE, over the whole batch, when the last slice is scored: k_post_chains is as
// long as its largest read takes (28 ms for a 64 M-anchor slice, 30 ms for the whole 200 M: four slices' post-passes one after the other were
// no faster than the round-4 call, profiles/r05_host_path_timeline.md), so it is paid once and only the chains' D2H follows it.
// Results land in page-locked blocks of the result cache (exact sizes: the totals are known before the copies start).
int Engine::chain_gpu_sliced(int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, mm2gb_chains_t *out, int64_t slice)
{
	const int64_t n = offsets[n_reads];
	if (begin_call()) return -1;
	const auto t0 = std::chrono::steady_clock::now();
	for (hipStream_t q : { s_in, work[0].stream, work[1].stream, s_out }) MM2GB_HIP(hipStreamSynchronize(q));
	// the whole batch resident: raw anchors, offsets, f, p in staging set 0 (sized for all of it); slices are copied into it piece by piece
	IoSet &s = io[0];
	io_seq = 1;
	const size_t nn = (size_t)std::max<int64_t>(n, 1);
	if (s.raw.ensure(nn * 16) || s.f.ensure(nn * 4) || s.p.ensure(nn * 4) || s.offsets.ensure((size_t)(n_reads + 1) * 8)) return -1;
	if (reserve_post(n, n_reads) || reserve_post_out(0, n, n_reads)) return -1;     // before anything is in flight: growing arenas waits
	// slices at read boundaries; the last one small (its kernels are exposed)
	std::vector<int64_t> first(1, 0);
	{
		const int64_t tail = slice / 4;
		int64_t acc = 0;
		bool tail_cut = false;
		for (int64_t r = 0; r + 1 < n_reads; ++r) {
			acc += offsets[r + 1] - offsets[r];
			const int64_t left = n - offsets[r + 1];
			if (tail_cut) continue;
			if (left <= tail && acc > tail) { first.push_back(r + 1); acc = 0; tail_cut = true; }
			else if (acc >= slice && left > tail + slice / 4) { first.push_back(r + 1); acc = 0; }
		}
	}
	first.push_back(n_reads);
	const size_t n_sl = first.size() - 1;
	auto give_up = [&](const std::string &why) {
		for (hipStream_t q : { s_in, work[0].stream, work[1].stream, s_out }) (void)hipStreamSynchronize(q);
		n_slots = 0;
		s.used = false;
		return fail(why);
	};
	// per-slice read offsets (each from 0): page-locked, the engine's; on the device they sit in a second offsets array (set 1's)
	if (h_slice_off.ensure(((size_t)n_reads + n_sl + 1) * 8) || io[1].offsets.ensure(((size_t)n_reads + n_sl + 1) * 8)) return give_up(last_error_cstr());
	int64_t *lo = (int64_t*)h_slice_off.ptr;
	std::vector<size_t> at(n_sl + 1, 0);
	for (size_t k = 0; k < n_sl; ++k) {
		at[k + 1] = at[k] + (size_t)(first[k + 1] - first[k]) + 1;
		for (int64_t r = first[k]; r <= first[k + 1]; ++r) lo[at[k] + (size_t)(r - first[k])] = offsets[r] - offsets[first[k]];
	}
	if (slice_in.size() < n_sl) { const size_t had = slice_in.size(); slice_in.resize(n_sl, nullptr); for (size_t k = had; k < n_sl; ++k) MM2GB_HIP(hipEventCreateWithFlags(&slice_in[k], hipEventDisableTiming)); }
	MM2GB_HIP(hipMemcpyAsync(s.offsets.ptr, offsets, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, s_in));
	MM2GB_HIP(hipMemcpyAsync(io[1].offsets.ptr, lo, at[n_sl] * 8, hipMemcpyHostToDevice, s_in));
	for (size_t k = 0; k < n_sl; ++k) {
		const int64_t a0 = offsets[first[k]], na = offsets[first[k + 1]] - a0;
		if (na > 0) MM2GB_HIP(hipMemcpyAsync((mm2gb_anchor_t*)s.raw.ptr + a0, anchors + a0, (size_t)na * 16, hipMemcpyHostToDevice, s_in));
		MM2GB_HIP(hipEventRecord(slice_in[k], s_in));
		MM2GB_HIP(hipStreamWaitEvent(stream, slice_in[k], 0));
		if (enqueue(first[k + 1] - first[k], (const int64_t*)io[1].offsets.ptr + at[k], (const mm2gb_anchor_t*)s.raw.ptr + a0, na, (int32_t*)s.f.ptr + a0, (int32_t*)s.p.ptr + a0, true, 0))
			return give_up(last_error_cstr());
	}
	if (enqueue_post(n_reads, (const int64_t*)s.offsets.ptr, (const mm2gb_anchor_t*)s.raw.ptr, n, (const int32_t*)s.f.ptr, (const int32_t*)s.p.ptr, nullptr, 0)) return give_up(last_error_cstr());
	PostOut &po = post_out[0];
	MM2GB_HIP(hipEventSynchronize(po.done));
	const int64_t n_u = po.h_totals[0], n_a = po.h_totals[1];
	out->u_off = (int64_t*)malloc((size_t)(n_reads + 1) * 8);
	out->a_off = (int64_t*)malloc((size_t)(n_reads + 1) * 8);
	out->u = (uint64_t*)result_alloc_pinned((size_t)(n_u + 1) * 8);
	out->a = (mm2gb_anchor_t*)result_alloc_pinned((size_t)(n_a + 1) * 16);
	if (!out->u_off || !out->a_off || !out->u || !out->a) { free(out->u_off); free(out->a_off); result_release(out->u); result_release(out->a); memset(out, 0, sizeof(*out)); return give_up("mm2gb_chain_gpu: out of host memory"); }
	MM2GB_HIP(hipMemcpyAsync(out->u_off, po.u_off.ptr, (size_t)(n_reads + 1) * 8, hipMemcpyDeviceToHost, s_out));
	MM2GB_HIP(hipMemcpyAsync(out->a_off, po.a_off.ptr, (size_t)(n_reads + 1) * 8, hipMemcpyDeviceToHost, s_out));
	if (n_u > 0) MM2GB_HIP(hipMemcpyAsync(out->u, po.u_out.ptr, (size_t)n_u * 8, hipMemcpyDeviceToHost, s_out));
	if (n_a > 0) MM2GB_HIP(hipMemcpyAsync(out->a, po.a_out.ptr, (size_t)n_a * 16, hipMemcpyDeviceToHost, s_out));
	MM2GB_HIP(hipStreamSynchronize(s_out));
	if (sync()) { mm2gb_chains_free(out); return -1; }
	s.used = false;                                     // nothing of this set is in flight once the call returns
	last.n_anchors = n; last.n_reads = n_reads;
	float ms = 0;
	if (hipEventElapsedTime(&ms, post0, post1) == hipSuccess) last.ms_post = ms;
	last.ms_total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
	if (debug_phases) fprintf(stderr, "[mm2gb chain_gpu] %lld anchors in %zu slices, %lld kept, %lld chains: %.1f ms (post-pass %.1f ms)\n", (long long)n, n_sl, (long long)n_a, (long long)n_u, last.ms_total, ms);
	return 0;
}

// Whole batch from host buffers to chains with every stage on the device: H2D of the anchors, score kernels, post-pass
// kernels, then only offsets + chains + compacted anchors come back (the host post-pass path returns 8 B per anchor).
// lchain.c:329-333: the skip counter grows by at most one per element of the inner tree visited, and that tree never holds more than
// cap_rmq_size elements when it is walked (lchain.c:301-310): a limit at or above the cap can never end a walk -- INT32_MAX then, the
// exhaustive walk both kernels have; anything below is the limit the one-anchor-per-step kernel keeps.  MM2GB_RMQ_SKIP=ignore: always
// exhaustive, as the device path was defined through round 5 (A/B runs).
static int rmq_skip_limit(const mm2gb_rmq_param_t &p)
{
	static const bool ignore = [] { const char *v = getenv("MM2GB_RMQ_SKIP"); return v && !strcmp(v, "ignore"); }();
	if (ignore || p.max_chn_skip == INT32_MAX || (p.cap_rmq_size > 0 && p.max_chn_skip >= p.cap_rmq_size)) return INT32_MAX;
	return p.max_chn_skip < 0 ? 0 : p.max_chn_skip;
}

int Engine::chain_gpu(int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, mm2gb_chains_t *out,
                      const mm2gb_rmq_param_t *rmq, int32_t *n_tied)
{
	memset(out, 0, sizeof(*out));
	if (!offsets || n_reads < 0 || offsets[0] != 0) return fail("mm2gb_chain_gpu: offsets[0] must be 0");
	for (int64_t r = 0; r < n_reads; ++r) if (offsets[r + 1] < offsets[r]) return fail("mm2gb_chain_gpu: offsets must be non-decreasing");
	const int64_t n = offsets[n_reads];
	if (n > 0 && !anchors) return fail("mm2gb_chain_gpu: null buffer");
	MM2GB_HIP(hipSetDevice(device));
	if (!rmq) {
		int64_t slice = 64 * 1000 * 1000;
		if (const char *v = getenv("MM2GB_CHAIN_SLICE_ANCHORS")) slice = std::max<int64_t>(1, atoll(v));
		if (n > slice + slice / 2 && n_reads > 1) return chain_gpu_sliced(n_reads, offsets, anchors, out, slice);
	}
	if (begin_call()) return -1;
	const auto t0 = std::chrono::steady_clock::now();
	IoSet &s = io[io_seq++ & 1];
	const size_t nn = (size_t)std::max<int64_t>(n, 1);
	for (hipStream_t q : { s_in, work[0].stream, work[1].stream, s_out }) MM2GB_HIP(hipStreamSynchronize(q));
	if (s.raw.ensure(nn * 16) || s.f.ensure(nn * 4) || s.p.ensure(nn * 4) || s.offsets.ensure((size_t)(n_reads + 1) * 8)) return -1;
	MM2GB_HIP(hipMemcpyAsync(s.offsets.ptr, offsets, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, stream));
	if (n > 0) MM2GB_HIP(hipMemcpyAsync(s.raw.ptr, anchors, (size_t)n * 16, hipMemcpyHostToDevice, stream));
	if (!rmq) {
		if (enqueue(n_reads, (const int64_t*)s.offsets.ptr, (const mm2gb_anchor_t*)s.raw.ptr, n, (int32_t*)s.f.ptr, (int32_t*)s.p.ptr)) return -1;
	} else {
		// mg_lchain_rmq's fill instead of the chaining DP; its key scratch shares the post-pass's candidate array (used after it)
		if (reserve_post(n, n_reads) || rmq_tied.ensure((size_t)std::max<int64_t>(n_reads, 1) * 4)) return -1;
		RmqBatch rb;
		rb.raw = (const uint4*)s.raw.ptr; rb.offsets = (const int64_t*)s.offsets.ptr; rb.n = n; rb.n_reads = n_reads;
		rb.f = (int32_t*)s.f.ptr; rb.p = (int32_t*)s.p.ptr; rb.key = (double*)post_z.ptr; rb.n_tied = (int32_t*)rmq_tied.ptr;
		const size_t n_sum = (size_t)(n >> 6) + (size_t)n_reads + 1;
		if (rmq_by_y.ensure(nn * 16) || rmq_ord.ensure(nn * 4) || rmq_meta.ensure(nn * 16) || rmq_sum.ensure(n_sum * 20)) return -1;
		rb.by_y = (ulonglong2*)rmq_by_y.ptr; rb.ord_idx = (int32_t*)rmq_ord.ptr; rb.meta = (int4*)rmq_meta.ptr;
		rb.l1 = (uint4*)rmq_sum.ptr; rb.bound = (int32_t*)((char*)rmq_sum.ptr + n_sum * 16);
		rb.win = nullptr; rb.tree = nullptr;
		{
			// tile form (k_rmq_fill_tiles) unless MM2GB_RMQ_KERNEL=steps asks for the one-anchor-per-step kernel (A/B runs, tests)
			const char *v = getenv("MM2GB_RMQ_KERNEL");
			// (a skip limit below the size cap can end an inner walk early, lchain.c:329-333: only the one-anchor-per-step kernel walks in the reference's order)
			const bool steps = rmq_skip_limit(*rmq) != INT32_MAX || (v ? !strcmp(v, "steps") : rmq_kernel == 1);
			rmq_tiles_last = !steps;
			if (!steps) {
				if (rmq_win.ensure(nn * 16) || rmq_tree.ensure(nn * 32)) return -1;
				rb.win = (int4*)rmq_win.ptr; rb.tree = (uint4*)rmq_tree.ptr;
			}
		}
		rb.skey_in = rb.skey = nullptr; rb.sa = nullptr; rb.srange = nullptr; rb.sort_tmp = nullptr; rb.sort_tmp_bytes = 0;
		rb.rk_a = nullptr; rb.rk_f = rb.rk_p = rb.rk_mark = nullptr; rb.rk_in = nullptr;
		const RmqParams rp0 = { rmq->max_dist, rmq->max_dist_inner, rmq->bw, rmq->cap_rmq_size, rmq->chn_pen_gap, rmq->chn_pen_skip, 0, INT32_MAX };
		rb.strip_shift = rmq_strip_shift(rp0);
		{
			// scratch of the batch's key sorts (ranks by y; the strip order), and -- tile kernel, MM2GB_RMQ_STRIPS=0 turns it off for A/B runs and
			// tests -- the inner window by strips of y (k_rmq_strip_ranges)
			const size_t tmp = rmq_strip_sort_temp_bytes(n, n_reads);
			if (rmq_skey_in.ensure(nn * 8) || rmq_skey.ensure(nn * 8) || rmq_sort_tmp.ensure(tmp)) return -1;
			rb.skey_in = (unsigned long long*)rmq_skey_in.ptr; rb.skey = (unsigned long long*)rmq_skey.ptr; rb.sort_tmp = rmq_sort_tmp.ptr; rb.sort_tmp_bytes = tmp;
			const char *v = getenv("MM2GB_RMQ_STRIPS");
			if (rb.tree && rb.strip_shift > 0 && !(v && atoi(v) == 0)) {
				if (rmq_sa.ensure(nn * 16) || rmq_srange.ensure(nn * 16)) return -1;
				rb.sa = (uint4*)rmq_sa.ptr; rb.srange = (int4*)rmq_srange.ptr;
			}
			if (!rb.tree && rmq_skip_limit(*rmq) != INT32_MAX) {
				// the skip-limited walk's arrays by rank, in the tile form's scratch (not in use here): anchors in `sa`'s, scores and predecessors'
				// ranks where the sort's input keys were (free once the (y, index) order stands), marks and walk ranges in `srange`'s
				if (rmq_sa.ensure(nn * 16) || rmq_srange.ensure(nn * 16)) return -1;
				rb.rk_a = (uint4*)rmq_sa.ptr; rb.rk_f = (int32_t*)rmq_skey_in.ptr; rb.rk_p = (int32_t*)rmq_skey_in.ptr + nn;
				rb.rk_in = (int2*)rmq_srange.ptr; rb.rk_mark = (int32_t*)((char*)rmq_srange.ptr + nn * 8);
			}
		}
		rb.cursor = (int32_t*)((char*)post_misc.ptr + 24); rb.grid_waves = n_cu * 32;
		rb.n_team = rmq_team_reads;
		rb.abandon_tied = rmq_abandon_tied && !getenv("MM2GB_RMQ_NO_ABANDON") ? 1 : 0;
		rmq_abandon_tied = false;
		rmq_team_reads = 0;
		if (const char *v = getenv("MM2GB_RMQ_TEAM_READS")) rb.n_team = std::max(0, atoi(v));
		rb.dbg = debug_phases ? (long long*)((char*)post_misc.ptr + 1536) : nullptr;
		if (debug_phases) MM2GB_HIP(hipMemsetAsync((char*)post_misc.ptr + 1536, 0, 64, stream));
		rb.dbg_reads = nullptr;
		if (debug_phases) {
			if (rmq_dbg_reads.ensure((size_t)std::max<int64_t>(n_reads, 1) * 64)) return -1;
			MM2GB_HIP(hipMemsetAsync(rmq_dbg_reads.ptr, 0, (size_t)std::max<int64_t>(n_reads, 1) * 64, stream));
			rb.dbg_reads = (long long*)rmq_dbg_reads.ptr;
		}
		const char *ties = getenv("MM2GB_RMQ_TIES");                   // strict: every tie counts (A/B runs, and what the one-anchor-per-step kernel always does)
		const RmqParams rp = { rmq->max_dist, rmq->max_dist_inner, rmq->bw, rmq->cap_rmq_size, rmq->chn_pen_gap, rmq->chn_pen_skip, ties && !strcmp(ties, "strict") ? 0 : 1, rmq_skip_limit(*rmq) };
		if (launch_rmq_fill(rb, rp, stream)) { (void)hipStreamSynchronize(stream); return fail("mm2gb_rmq_chain_gpu: the segmented sort of the batch's keys failed"); }
		MM2GB_HIP(hipGetLastError());
		last.n_anchors += n; last.n_reads += n_reads;
		if (rmq_tied_ready && n_tied && n_reads > 0) {
			if (!rmq_fill_done) MM2GB_HIP(hipEventCreateWithFlags(&rmq_fill_done, hipEventDisableTiming));
			MM2GB_HIP(hipEventRecord(rmq_fill_done, stream));
		}
	}
	if (enqueue_post(n_reads, (const int64_t*)s.offsets.ptr, (const mm2gb_anchor_t*)s.raw.ptr, n, (const int32_t*)s.f.ptr, (const int32_t*)s.p.ptr, rmq)) return -1;
	if (rmq && rmq_tied_ready) {
		// the tie counts are known when the fill is done: hand them over while the post-pass runs (the copy goes by the D2H stream, past the kernels)
		std::function<void(const int32_t*)> cb;
		cb.swap(rmq_tied_ready);
		if (n_tied && n_reads > 0) {
			MM2GB_HIP(hipStreamWaitEvent(s_out, rmq_fill_done, 0));
			MM2GB_HIP(hipMemcpyAsync(n_tied, rmq_tied.ptr, (size_t)n_reads * 4, hipMemcpyDeviceToHost, s_out));
			MM2GB_HIP(hipStreamSynchronize(s_out));
			cb(n_tied);
		}
	}
	s.used = false;                                     // nothing of this set is in flight once the call returns
	const double s_enqueued = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (sync()) return -1;
	const double s_synced = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (rmq && debug_phases && n > 0) {
		long long t[8] = { 0 };
		if (hipMemcpy(t, (char*)post_misc.ptr + 1536, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess) {
			if (rmq_tiles_last && rmq_dbg_reads.ptr && n_reads > 0) {
				std::vector<long long> tr((size_t)n_reads * 8);
				if (hipMemcpy(tr.data(), rmq_dbg_reads.ptr, tr.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
					std::vector<int64_t> idx((size_t)n_reads);
					for (int64_t r = 0; r < n_reads; ++r) idx[(size_t)r] = r;
					std::sort(idx.begin(), idx.end(), [&](int64_t u, int64_t v) { return tr[8 * u + 2] > tr[8 * v + 2]; });
					fprintf(stderr, "[mm2gb rmq fill, tiles] the slowest reads (position in the call | anchors | waves | ms whole | tree update | queries | broadcasts (incl. waiting for the team) | in-tile steps | anchors broadcast by wave 0):\n");
					for (size_t k = 0; k < std::min<size_t>(idx.size(), 8); ++k) {
						const long long *o = tr.data() + 8 * idx[k];
						fprintf(stderr, "    %6lld | %8lld | %2lld | %8.2f | %8.2f | %8.2f | %8.2f | %8.2f | %lld\n", (long long)idx[k], o[0], o[1], o[2] / 1e5, o[3] / 1e5, o[4] / 1e5, o[5] / 1e5, o[6] / 1e5, o[7] & 0xfffffffffLL);
					}
				}
			}
			if (rmq_tiles_last)
				fprintf(stderr, "[mm2gb rmq fill, tiles] %lld anchors, %lld tiles; wave time in 10 ns ticks, summed over reads: tree update %lld, queries %lld, broadcasts %lld (%lld anchors broadcast, %lld blocks scanned for one lane, "
				                "%lld inner blocks passed over, %lld lanes scanned their inner window again), in-tile steps %lld\n", t[0], t[1], t[2], t[3], t[4], t[5] & 0xfffffffffLL, t[5] >> 36, t[6] & 0xffffffffLL, t[6] >> 32, t[7]);
			else
				fprintf(stderr, "[mm2gb rmq fill] %lld steps: late entries %lld, summaries rebuilt %lld, ties looked at %lld, summary loads %lld, inner blocks read %lld, winners read from memory %lld, eviction tests %lld\n",
				        t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
		}
	}
	const int64_t n_u = n_reads > 0 ? h_post_totals[0] : 0, n_a = n_reads > 0 ? h_post_totals[1] : 0;
	out->u_off = (int64_t*)malloc((size_t)(n_reads + 1) * 8);
	out->a_off = (int64_t*)malloc((size_t)(n_reads + 1) * 8);
	out->u = (uint64_t*)malloc((size_t)(n_u + 1) * 8);
	out->a = (mm2gb_anchor_t*)result_alloc_pinned((size_t)(n_a + 1) * 16);
	if (!out->u_off || !out->a_off || !out->u || !out->a) { free(out->u_off); free(out->a_off); free(out->u); result_release(out->a); memset(out, 0, sizeof(*out)); return fail("mm2gb_chain_gpu: out of host memory"); }
	out->u_off[0] = out->a_off[0] = 0;
	if (n_reads > 0) {
		// on the engine's own D2H stream, never the null stream: every engine of the process would queue behind the same one
		MM2GB_HIP(hipMemcpyAsync(out->u_off, post_uoff.ptr, (size_t)(n_reads + 1) * 8, hipMemcpyDeviceToHost, s_out));
		MM2GB_HIP(hipMemcpyAsync(out->a_off, post_aoff.ptr, (size_t)(n_reads + 1) * 8, hipMemcpyDeviceToHost, s_out));
		if (n_u > 0) MM2GB_HIP(hipMemcpyAsync(out->u, post_uout.ptr, (size_t)n_u * 8, hipMemcpyDeviceToHost, s_out));
		if (n_a > 0) MM2GB_HIP(hipMemcpyAsync(out->a, post_aout.ptr, (size_t)n_a * 16, hipMemcpyDeviceToHost, s_out));
		if (rmq && n_tied) MM2GB_HIP(hipMemcpyAsync(n_tied, rmq_tied.ptr, (size_t)n_reads * 4, hipMemcpyDeviceToHost, s_out));
		MM2GB_HIP(hipStreamSynchronize(s_out));
	}
	float ms = 0;
	if (n_reads > 0 && hipEventElapsedTime(&ms, post0, post1) == hipSuccess) last.ms_post = ms;
	last.ms_total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
	if (debug_phases && n > 0) print_post_debug(n_reads, (const int64_t*)s.offsets.ptr);
	if (debug_phases && n > 0)
		fprintf(stderr, "[mm2gb chain_gpu%s] %lld anchors in, %lld kept: enqueued at %.1f ms, device done at %.1f ms (post-pass %.1f ms), results copied at %.1f ms\n", rmq ? ", rmq" : "",
		        (long long)n, (long long)n_a, s_enqueued * 1e3, s_synced * 1e3, ms, last.ms_total);
	return 0;
}

// Seed sort (map.c:329) of every read's anchors, in place in the caller's host array: H2D, one wave per read, D2H.
int Engine::sort_seeds(int64_t n_reads, const int64_t *offsets, mm2gb_anchor_t *anchors)
{
	if (!offsets || n_reads < 0 || offsets[0] != 0) return fail("mm2gb_sort_seeds_gpu: offsets[0] must be 0");
	for (int64_t r = 0; r < n_reads; ++r) {
		if (offsets[r + 1] < offsets[r]) return fail("mm2gb_sort_seeds_gpu: offsets must be non-decreasing");
		if (offsets[r + 1] - offsets[r] >= ((int64_t)1 << 31)) return fail("mm2gb_sort_seeds_gpu: a read is limited to 2^31 anchors");
	}
	const int64_t n = offsets[n_reads];
	if (n == 0 || n_reads == 0) return 0;
	if (!anchors) return fail("mm2gb_sort_seeds_gpu: null buffer");
	MM2GB_HIP(hipSetDevice(device));
	IoSet &s = io[0];
	for (hipStream_t q : { s_in, work[0].stream, work[1].stream, s_out }) MM2GB_HIP(hipStreamSynchronize(q));
	if (s.raw.ensure((size_t)n * 16) || s.offsets.ensure((size_t)(n_reads + 1) * 8)) return -1;
	MM2GB_HIP(hipMemcpyAsync(s.offsets.ptr, offsets, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, stream));
	MM2GB_HIP(hipMemcpyAsync(s.raw.ptr, anchors, (size_t)n * 16, hipMemcpyHostToDevice, stream));
	SortBatch sb;
	sb.a = (ulonglong2*)s.raw.ptr; sb.offsets = (const int64_t*)s.offsets.ptr; sb.n_reads = n_reads; sb.cursor = nullptr; sb.grid_waves = n_cu * 32;
	launch_sort_x(sb, stream);
	MM2GB_HIP(hipGetLastError());
	MM2GB_HIP(hipMemcpyAsync(anchors, s.raw.ptr, (size_t)n * 16, hipMemcpyDeviceToHost, stream));
	MM2GB_HIP(hipStreamSynchronize(stream));
	return 0;
}

// collect_seed_hits (map.c:295-331) for every read of a batch: matches in, sorted anchors out.
int Engine::collect_seeds(int64_t opt_flag, int64_t n_reads, const int64_t *seed_off, const mm2gb_seed_t *seeds, const int64_t *hit_off, const uint64_t *hits,
                          const int32_t *qlen, const int32_t *q_rank, int32_t n_ref, const int32_t *ref_len, const int32_t *ref_rank, int64_t *anchor_off, mm2gb_anchor_t *anchors)
{
	constexpr int64_t F_NO_DIAG = 0x001, F_NO_DUAL = 0x002, F_QSTRAND = 0x100000000LL;
	if (n_reads < 0 || !seed_off || !anchor_off || seed_off[0] != 0) return fail("mm2gb_collect_seeds_gpu: seed_off[0] must be 0");
	anchor_off[0] = 0;
	if (n_reads == 0) return 0;
	for (int64_t r = 0; r < n_reads; ++r) if (seed_off[r + 1] < seed_off[r]) return fail("mm2gb_collect_seeds_gpu: seed_off must be non-decreasing");
	const int64_t n_seeds = seed_off[n_reads];
	if (!qlen || !hit_off || (n_seeds > 0 && !seeds) || hit_off[0] != 0) return fail("mm2gb_collect_seeds_gpu: null argument, or hit_off[0] is not 0");
	for (int64_t k = 0; k < n_seeds; ++k) if (hit_off[k + 1] - hit_off[k] != (int64_t)seeds[k].n) return fail("mm2gb_collect_seeds_gpu: hit_off does not match the seeds' hit counts");
	const int64_t n_hits = hit_off[n_seeds];
	for (int64_t r = 0; r < n_reads; ++r) if (hit_off[seed_off[r + 1]] - hit_off[seed_off[r]] >= ((int64_t)1 << 31)) return fail("mm2gb_collect_seeds_gpu: a read is limited to 2^31 hits");
	const bool names = (opt_flag & (F_NO_DIAG | F_NO_DUAL)) != 0 && q_rank != nullptr;
	if (names && (!ref_rank || n_ref <= 0)) return fail("mm2gb_collect_seeds_gpu: NO_DIAG / NO_DUAL need ref_rank");
	if (((opt_flag & F_QSTRAND) || (names && (opt_flag & F_NO_DIAG))) && (!ref_len || n_ref <= 0)) return fail("mm2gb_collect_seeds_gpu: QSTRAND / NO_DIAG need ref_len");
	if (n_hits == 0) { for (int64_t r = 0; r < n_reads; ++r) anchor_off[r + 1] = 0; return 0; }
	if (!hits || !anchors) return fail("mm2gb_collect_seeds_gpu: null buffer");
	if (ref_len || ref_rank)
		for (int64_t h = 0; h < n_hits; ++h) if ((int64_t)(hits[h] >> 32) >= n_ref) return fail("mm2gb_collect_seeds_gpu: a hit names a reference sequence >= n_ref");
	MM2GB_HIP(hipSetDevice(device));
	MM2GB_HIP(hipStreamSynchronize(stream));
	const size_t nr = (size_t)n_reads, ns = (size_t)std::max<int64_t>(n_seeds, 1), nh = (size_t)n_hits, nf = (size_t)std::max<int32_t>(n_ref, 1);
	if (sd_seeds.ensure(ns * 16) || sd_seed_off.ensure((nr + 1) * 8) || sd_hit_off.ensure((ns + 1) * 8) || sd_hits.ensure(nh * 8) || sd_qlen.ensure(nr * 4) ||
	    sd_q_rank.ensure(nr * 4) || sd_ref_len.ensure(nf * 4) || sd_ref_rank.ensure(nf * 4) || sd_seed_read.ensure(ns * 4) || sd_tmp.ensure(nh * 16) ||
	    sd_n_kept.ensure(nr * 4) || sd_a_off.ensure((nr + 1) * 8) || sd_out.ensure(nh * 16)) return -1;
	MM2GB_HIP(hipMemcpyAsync(sd_seeds.ptr, seeds, (size_t)n_seeds * 16, hipMemcpyHostToDevice, stream));
	MM2GB_HIP(hipMemcpyAsync(sd_seed_off.ptr, seed_off, (nr + 1) * 8, hipMemcpyHostToDevice, stream));
	MM2GB_HIP(hipMemcpyAsync(sd_hit_off.ptr, hit_off, (size_t)(n_seeds + 1) * 8, hipMemcpyHostToDevice, stream));
	MM2GB_HIP(hipMemcpyAsync(sd_hits.ptr, hits, nh * 8, hipMemcpyHostToDevice, stream));
	MM2GB_HIP(hipMemcpyAsync(sd_qlen.ptr, qlen, nr * 4, hipMemcpyHostToDevice, stream));
	if (names) MM2GB_HIP(hipMemcpyAsync(sd_q_rank.ptr, q_rank, nr * 4, hipMemcpyHostToDevice, stream));
	if (ref_len) MM2GB_HIP(hipMemcpyAsync(sd_ref_len.ptr, ref_len, (size_t)n_ref * 4, hipMemcpyHostToDevice, stream));
	if (names) MM2GB_HIP(hipMemcpyAsync(sd_ref_rank.ptr, ref_rank, (size_t)n_ref * 4, hipMemcpyHostToDevice, stream));
	SeedBatch sb;
	sb.seeds = (const SeedRecord*)sd_seeds.ptr; sb.seed_off = (const int64_t*)sd_seed_off.ptr; sb.hit_off = (const int64_t*)sd_hit_off.ptr;
	sb.hits = (const unsigned long long*)sd_hits.ptr; sb.qlen = (const int32_t*)sd_qlen.ptr; sb.q_rank = names ? (const int32_t*)sd_q_rank.ptr : nullptr;
	sb.ref_len = ref_len ? (const int32_t*)sd_ref_len.ptr : nullptr; sb.ref_rank = names ? (const int32_t*)sd_ref_rank.ptr : nullptr;
	sb.n_reads = n_reads; sb.n_seeds = n_seeds; sb.n_hits = n_hits; sb.flag = (long long)opt_flag;
	sb.seed_read = (int32_t*)sd_seed_read.ptr; sb.tmp = (ulonglong2*)sd_tmp.ptr; sb.n_kept = (int32_t*)sd_n_kept.ptr;
	sb.anchor_off = (int64_t*)sd_a_off.ptr; sb.out = (ulonglong2*)sd_out.ptr; sb.grid_waves = n_cu * 32;
	static_assert(sizeof(SeedRecord) == sizeof(mm2gb_seed_t) && sizeof(mm2gb_seed_t) == 16, "seed record layout");
	launch_collect_seeds(sb, stream);
	MM2GB_HIP(hipGetLastError());
	MM2GB_HIP(hipMemcpyAsync(anchor_off, sd_a_off.ptr, (nr + 1) * 8, hipMemcpyDeviceToHost, stream));
	MM2GB_HIP(hipStreamSynchronize(stream));
	if (anchor_off[n_reads] > 0) { MM2GB_HIP(hipMemcpyAsync(anchors, sd_out.ptr, (size_t)anchor_off[n_reads] * 16, hipMemcpyDeviceToHost, s_out)); MM2GB_HIP(hipStreamSynchronize(s_out)); }
	return 0;
}

// mm_gen_regs (hit.c:52-88) for every read of a batch of chains.
int Engine::gen_regs(int64_t n_reads, const mm2gb_chains_t *ch, const int32_t *qlen, const uint32_t *hash, int is_qstrand, mm2gb_reg_t *regs)
{
	if (!ch || n_reads < 0 || (n_reads > 0 && (!ch->u_off || !ch->a_off || !qlen || !hash))) return fail("mm2gb_gen_regs_gpu: null argument");
	if (n_reads == 0) return 0;
	const int64_t n_u = ch->u_off[n_reads], n_a = ch->a_off[n_reads];
	if (n_u == 0) return 0;
	if (!regs || !ch->u || !ch->a) return fail("mm2gb_gen_regs_gpu: null buffer");
	static_assert(sizeof(RegRecord) == sizeof(mm2gb_reg_t) && sizeof(mm2gb_reg_t) == 72, "hit record layout");
	MM2GB_HIP(hipSetDevice(device));
	if (reserve_post(std::max<int64_t>(n_a, n_u), n_reads) || reserve_post_out(0, std::max<int64_t>(n_a, n_u), n_reads) || reg_out.ensure((size_t)n_u * sizeof(RegRecord))) return -1;
	MM2GB_HIP(hipStreamSynchronize(stream));
	MM2GB_HIP(hipMemcpyAsync(post_uoff.ptr, ch->u_off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, stream));
	MM2GB_HIP(hipMemcpyAsync(post_aoff.ptr, ch->a_off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, stream));
	MM2GB_HIP(hipMemcpyAsync(post_uout.ptr, ch->u, (size_t)n_u * 8, hipMemcpyHostToDevice, stream));
	MM2GB_HIP(hipMemcpyAsync(post_aout.ptr, ch->a, (size_t)n_a * 16, hipMemcpyHostToDevice, stream));
	MM2GB_HIP(hipMemcpyAsync(post_nu.ptr, qlen, (size_t)n_reads * 4, hipMemcpyHostToDevice, stream));
	MM2GB_HIP(hipMemcpyAsync(post_nkept.ptr, hash, (size_t)n_reads * 4, hipMemcpyHostToDevice, stream));
	RegBatch rb;
	rb.u_off = (const int64_t*)post_uoff.ptr; rb.a_off = (const int64_t*)post_aoff.ptr; rb.u = (const unsigned long long*)post_uout.ptr; rb.a = (const uint4*)post_aout.ptr;
	rb.qlen = (const int32_t*)post_nu.ptr; rb.hash = (const uint32_t*)post_nkept.ptr; rb.n_reads = n_reads;
	rb.z = (ulonglong2*)post_heads.ptr; rb.regs = (RegRecord*)reg_out.ptr; rb.cursor = nullptr; rb.is_qstrand = is_qstrand; rb.grid_waves = n_cu * 32;
	launch_gen_regs(rb, stream);
	MM2GB_HIP(hipGetLastError());
	MM2GB_HIP(hipMemcpyAsync(regs, reg_out.ptr, (size_t)n_u * sizeof(RegRecord), hipMemcpyDeviceToHost, stream));
	MM2GB_HIP(hipStreamSynchronize(stream));
	return 0;
}

int Engine::record_outputs_done(hipEvent_t ev)
{
	MM2GB_HIP(hipSetDevice(device));
	MM2GB_HIP(hipEventRecord(ev, s_out));
	return 0;
}

int Engine::enqueue_host(int64_t n_reads, const int64_t *h_offsets, const mm2gb_anchor_t *h_anchors, int64_t n, int32_t *h_f, int32_t *h_p, bool want_stats)
{
	MM2GB_HIP(hipSetDevice(device));
	// Small micro-batches alternate between the two compute streams: one of them cannot fill the GPU (it ends at the pace of
	// its largest chunk, run by one workgroup), so the next one's kernels start beside its tail -- 64-read batches through the
	// boundary went from 6.2 to 4.1 ms each.  Large ones stay on one stream: with ~100 M-anchor slices alternating was a wash
	// (400 M anchors: 148 vs 153 ms; 200 M: 87 vs 93 ms; but 64 M-anchor slices 178 vs 160 ms and the chains call 146 vs 136 ms).
	const int set = (!one_compute_stream && n <= dual_stream_max_n) ? (int)(io_seq & 1) : 0;
	IoSet &s = io[io_seq++ & 1];
	hipStream_t stream = work[set].stream;
	const size_t nn = (size_t)std::max<int64_t>(n, 1);
	if (s.raw.bytes < nn * 16 || s.f.bytes < nn * 4 || s.p.bytes < nn * 4 || s.offsets.bytes < (size_t)(n_reads + 1) * 8) {
		// growing a set frees its old buffers: nothing may be in flight on it
		for (hipStream_t q : { s_in, work[0].stream, work[1].stream, s_out }) MM2GB_HIP(hipStreamSynchronize(q));
		if (s.raw.ensure(nn * 16) || s.f.ensure(nn * 4) || s.p.ensure(nn * 4) || s.offsets.ensure((size_t)(n_reads + 1) * 8)) return -1;
	}
	// H2D may overwrite raw/offsets only after the kernels that last read this set are done
	if (s.used) MM2GB_HIP(hipStreamWaitEvent(s_in, s.comp_done, 0));
	MM2GB_HIP(hipEventRecord(s.in_start, s_in));
	MM2GB_HIP(hipMemcpyAsync(s.offsets.ptr, h_offsets, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, s_in));
	if (n > 0) MM2GB_HIP(hipMemcpyAsync(s.raw.ptr, h_anchors, (size_t)n * 16, hipMemcpyHostToDevice, s_in));
	MM2GB_HIP(hipEventRecord(s.in_done, s_in));
	// kernels need the inputs, and may overwrite f/p only after the previous D2H from this set is done
	MM2GB_HIP(hipStreamWaitEvent(stream, s.in_done, 0));
	if (s.used) MM2GB_HIP(hipStreamWaitEvent(stream, s.out_done, 0));
	if (enqueue(n_reads, (const int64_t*)s.offsets.ptr, (const mm2gb_anchor_t*)s.raw.ptr, n, (int32_t*)s.f.ptr, (int32_t*)s.p.ptr, want_stats, set)) return -1;
	MM2GB_HIP(hipEventRecord(s.comp_done, stream));
	MM2GB_HIP(hipStreamWaitEvent(s_out, s.comp_done, 0));
	MM2GB_HIP(hipEventRecord(s.out_start, s_out));
	if (n > 0) {
		MM2GB_HIP(hipMemcpyAsync(h_f, s.f.ptr, (size_t)n * 4, hipMemcpyDeviceToHost, s_out));
		MM2GB_HIP(hipMemcpyAsync(h_p, s.p.ptr, (size_t)n * 4, hipMemcpyDeviceToHost, s_out));
	}
	MM2GB_HIP(hipEventRecord(s.out_done, s_out));
	s.used = true;
	return 0;
}

int Engine::sync()
{
	MM2GB_HIP(hipSetDevice(device));
	for (hipStream_t q : { s_in, work[0].stream, work[1].stream, s_out }) MM2GB_HIP(hipStreamSynchronize(q));
	return collect_stats();
}

int Engine::collect_stats()
{
	if (debug_phases && n_slots > 0) {
		std::vector<int64_t> h((size_t)launch.score_grid * 4);
		if (hipMemcpy(h.data(), dbg.ptr, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
			int64_t t0 = INT64_MAX, e1 = 0, e2 = 0, e3 = 0, s1 = 0, s2 = 0, s3 = 0; int n = 0;
			for (int w = 0; w < launch.score_grid; ++w) if (h[4 * w]) t0 = std::min(t0, h[4 * w]);
			for (int w = 0; w < launch.score_grid; ++w) {
				if (!h[4 * w]) continue;
				const int64_t a = h[4 * w + 1] ? h[4 * w + 1] - t0 : 0, b2 = h[4 * w + 2] - t0, c = h[4 * w + 3] - t0;
				e1 = std::max(e1, a); e2 = std::max(e2, b2); e3 = std::max(e3, c); s1 += a; s2 += b2; s3 += c; ++n;
			}
			if (n) fprintf(stderr, "[mm2gb phases] workgroups %d | end of big-team phase: mean %.2f max %.2f ms | end of 4-wave phase: mean %.2f max %.2f ms | end: mean %.2f max %.2f ms\n",
			               n, s1 / 1e5 / n, e1 / 1e5, s2 / 1e5 / n, e2 / 1e5, s3 / 1e5 / n, e3 / 1e5);
			if (n && e3 > 0) {                                   // how many workgroups are still at work at eighths of the kernel's time
				int busy[8] = { 0 };
				for (int w = 0; w < launch.score_grid; ++w) if (h[4 * w]) for (int k = 0; k < 8; ++k) busy[k] += h[4 * w + 3] - t0 > e3 * k / 8;
				fprintf(stderr, "[mm2gb phases] workgroups still at work after 0/8 ... 7/8 of the kernel's time: %d %d %d %d %d %d %d %d\n", busy[0], busy[1], busy[2], busy[3], busy[4], busy[5], busy[6], busy[7]);
			}
		}
	}
	for (int k = 0; k < n_slots; ++k) {
		const int32_t *c = h_counters + (size_t)k * CNT_WORDS;
		last.n_pairs += h_totals[(size_t)k * 2];
		last.n_chunks += c[CNT_NCHUNK];
		last.n_long_chunks += c[CNT_NLONG];
		last.n_mid_chunks += c[CNT_NMID];
		last.n_tracked_chunks += c[CNT_NTRACK];
		last.n_clamped_blocks += c[CNT_NCLAMP];
		last_split_chunks += c[CNT_NSPLIT]; last_helped_items += c[CNT_HELPED];
		last_gang_chunks += c[CNT_NGANG]; last_gang_wgs += c[CNT_GANG_WGS];
		float ms = 0;
		if (hipEventElapsedTime(&ms, slots[k].prep0, slots[k].prep1) == hipSuccess) last.ms_prep += ms;
		if (hipEventElapsedTime(&ms, slots[k].prep1, slots[k].score1) == hipSuccess) last.ms_score += ms;
	}
	if (n_slots > 0) last.ms_total = last.ms_prep + last.ms_score;     // host-buffer calls overwrite this with wall time
	n_slots = 0;
	return 0;
}

// Scores for reads [0, n_reads) whose anchors are anchors[offsets[0] .. offsets[n_reads]) (offsets[0] need not be 0: a
// device of a pool gets a run of reads out of a larger batch); f[0] / p[0] belong to anchor offsets[0].
// Large batches are cut at read boundaries into slices so that the H2D of slice k+1, the kernels of slice k and the D2H of
// slice k-1 overlap (three streams, two device staging sets).  MM2GB_SLICE_ANCHORS sets the slice size.
// slice_done(r0, r1), if given, is called on this thread as soon as the scores of reads [r0, r1) are in host memory, in
// read order, while later slices are still in flight: the caller's post-pass can start on them.
int Engine::score_host(int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, int32_t *f, int32_t *p,
                       const std::function<void(int64_t, int64_t)> *slice_done)
{
	const int64_t base = offsets[0], n = offsets[n_reads] - base;
	for (int64_t r = 0; r < n_reads; ++r) if (offsets[r + 1] < offsets[r]) return fail("mm2gb_score_host: offsets must be non-decreasing");
	if (n > 0 && (!anchors || !f || !p)) return fail("mm2gb_score_host: null buffer");
	MM2GB_HIP(hipSetDevice(device));
	if (begin_call()) return -1;
	const auto t0 = std::chrono::steady_clock::now();
	// Slice size: the H2D copy (57 GB/s measured, profiles/ubench/h2d_rate.hip) is what a large batch is bound by, as long as the
	// kernels of a slice are faster than its copy -- and a slice too small runs at the pace of its largest chunks.
	// ~100 M anchors balance the two.  What follows the last copy is exposed (its kernels, its D2H), so the last slice is a small one.
	int64_t slice = 96 * 1000 * 1000;
	if (const char *v = getenv("MM2GB_SLICE_ANCHORS")) slice = std::max<int64_t>(1, atoll(v));
	std::vector<int64_t> first(1, 0);
	if (n > slice + slice / 2) {
		int64_t tail = slice / 4;                          // anchors kept for the last slice
		if (const char *v = getenv("MM2GB_SLICE_TAIL_ANCHORS")) tail = std::max<int64_t>(1, atoll(v));
		// nothing runs before the first slice has arrived: a smaller one starts the kernels earlier
		int64_t first_slice = slice;
		if (const char *v = getenv("MM2GB_SLICE_FIRST_ANCHORS")) first_slice = std::max<int64_t>(1, atoll(v));
		int64_t acc = 0;
		bool tail_cut = false;
		for (int64_t r = 0; r < n_reads; ++r) {
			acc += offsets[r + 1] - offsets[r];
			const int64_t left = offsets[n_reads] - offsets[r + 1];
			if (r + 1 >= n_reads || tail_cut) continue;
			if (left <= tail && acc > tail) { first.push_back(r + 1); acc = 0; tail_cut = true; }             // ... [rest of a slice][tail]
			else if (acc >= (first.size() == 1 ? first_slice : slice) && left > tail + slice / 4) { first.push_back(r + 1); acc = 0; }   // a full slice, enough left for more
		}
	}
	first.push_back(n_reads);
	const size_t n_sl = first.size() - 1;
	auto drain = [&]() {                 // error path: keep the error text, wait for whatever was enqueued
		const std::string why = last_error_cstr();
		for (hipStream_t q : { s_in, work[0].stream, work[1].stream, s_out }) (void)hipStreamSynchronize(q);
		n_slots = 0;
		set_error(why);
	};
	if (h_slice_off.ensure(((size_t)n_reads + n_sl + 1) * 8)) return -1;
	int64_t *lo = (int64_t*)h_slice_off.ptr;
	size_t w = 0;
	for (size_t k = 0; k < n_sl; ++k) {
		const int64_t r0 = first[k], r1 = first[k + 1];
		const size_t at = w;
		for (int64_t r = r0; r <= r1; ++r) lo[w++] = offsets[r] - offsets[r0];
		// on failure: copies of earlier slices may still be writing into the caller's f / p -- never return with those in flight
		if (enqueue_host(r1 - r0, lo + at, anchors + offsets[r0], offsets[r1] - offsets[r0], f + (offsets[r0] - base), p + (offsets[r0] - base))) { drain(); return -1; }
		if (slice_done && k > 0) {           // slice k is queued behind it: hand slice k-1 over once its D2H has landed
			if (hipEventSynchronize(io[(io_seq - 2) & 1].out_done) != hipSuccess) { drain(); return fail("mm2gb_score_host: waiting for a slice failed"); }
			(*slice_done)(first[k - 1], first[k]);
		}
	}
	if (sync()) return -1;
	if (slice_done) (*slice_done)(first[n_sl - 1], first[n_sl]);
	last.ms_total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
	{   // transfer times of the last slice (events on the copy streams)
		const IoSet &s = io[(io_seq - 1) & 1];
		float ms = 0;
		if (hipEventElapsedTime(&ms, s.in_start, s.in_done) == hipSuccess) last.ms_h2d = ms;
		if (hipEventElapsedTime(&ms, s.out_start, s.out_done) == hipSuccess) last.ms_d2h = ms;
	}
	return 0;
}

} // namespace mm2gb

using namespace mm2gb;

extern "C" {

const char *mm2gb_last_error(void) { return last_error_cstr(); }
// of the engine's last completed call: chunks scored strip by strip (k_score's SPLIT build) and the items of such chunks that
// workgroups other than the owner took
void mm2gb_engine_split_counts(const mm2gb_engine_t *eng, int64_t *chunks, int64_t *helped_items)
{
	if (chunks) *chunks = eng ? eng->e.last_split_chunks : 0;
	if (helped_items) *helped_items = eng ? eng->e.last_helped_items : 0;
}
// of the engine's last completed call: chunks that a gang of workgroups scored (k_score's phase 0) and the workgroups that started in one
void mm2gb_engine_gang_counts(const mm2gb_engine_t *eng, int64_t *chunks, int64_t *workgroups)
{
	if (chunks) *chunks = eng ? eng->e.last_gang_chunks : 0;
	if (workgroups) *workgroups = eng ? eng->e.last_gang_wgs : 0;
}
const char *mm2gb_version(void) { return MM2GB_VERSION; }
int mm2gb_has_split_build(void) { return score_has_split_build() ? 1 : 0; }
int mm2gb_has_gang_build(void) { return 1; }

int mm2gb_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

mm2gb_engine_t *mm2gb_engine_create(const mm2gb_config_t *cfg, const mm2gb_misc_t *misc, int device)
{
	mm2gb_config_t def;
	if (!cfg) { mm2gb_config_defaults(&def); cfg = &def; }
	if (!misc) { set_error("mm2gb_engine_create: misc is required"); return nullptr; }
	mm2gb_engine_t *eng = new mm2gb_engine_t();
	if (eng->e.init(cfg, misc, device)) { eng->e.shutdown(); delete eng; return nullptr; }
	return eng;
}

void mm2gb_engine_destroy(mm2gb_engine_t *eng)
{
	if (!eng) return;
	eng->e.shutdown();
	if (eng->host_scratch && eng->host_scratch_free) eng->host_scratch_free(eng->host_scratch);
	delete eng;
}

int mm2gb_engine_set_misc(mm2gb_engine_t *eng, const mm2gb_misc_t *misc) { return eng ? eng->e.set_misc(misc) : fail("mm2gb: null engine"); }
int mm2gb_engine_device(const mm2gb_engine_t *eng) { return eng ? eng->e.device : -1; }
int mm2gb_engine_set_rmq_team_reads(mm2gb_engine_t *eng, int n) { if (!eng || n < 0) return fail("mm2gb_engine_set_rmq_team_reads: bad argument"); eng->e.rmq_team_reads = n; return 0; }
int mm2gb_engine_set_rmq_kernel(mm2gb_engine_t *eng, int kind) { if (!eng || kind < 0 || kind > 1) return fail("mm2gb_engine_set_rmq_kernel: bad argument"); eng->e.rmq_kernel = kind; return 0; }
int mm2gb_engine_reserve(mm2gb_engine_t *eng, int64_t n_anchors, int64_t n_reads) { return eng ? eng->e.reserve(n_anchors, n_reads) : fail("mm2gb: null engine"); }
void *mm2gb_engine_stream(mm2gb_engine_t *eng) { return eng ? (void*)eng->e.stream : nullptr; }
float mm2gb_engine_last_kernel_ms(mm2gb_engine_t *eng) { return eng ? eng->e.last.ms_prep + eng->e.last.ms_score : 0.f; }

int mm2gb_score_device(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *d_offsets, const mm2gb_anchor_t *d_anchors,
                       int64_t n_anchors, int32_t *d_f, int32_t *d_p)
{
	if (!eng) return fail("mm2gb: null engine");
	Engine &e = eng->e;
	MM2GB_HIP(hipSetDevice(e.device));
	if (e.begin_call()) return -1;
	return e.enqueue(n_reads, d_offsets, d_anchors, n_anchors, d_f, d_p);
}

int mm2gb_engine_sync(mm2gb_engine_t *eng) { return eng ? eng->e.sync() : fail("mm2gb: null engine"); }

int mm2gb_chain_gpu(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, mm2gb_chains_t *out, mm2gb_stats_t *stats)
{
	if (!eng || !out) return fail("mm2gb_chain_gpu: null argument");
	if (eng->e.chain_gpu(n_reads, offsets, anchors, out)) return -1;
	if (stats) *stats = eng->e.last;
	return 0;
}

// MM2GB_DEBUG_PHASES: what the last post-pass kernel recorded (to stderr)
void Engine::print_post_debug(int64_t n_reads, const int64_t *d_offsets)
{
	long long t[64] = { 0 };
	if (hipMemcpy(t, (char*)post_misc.ptr + 1024, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess)
		for (int lv = 0; lv < 3; ++lv)
			fprintf(stderr, "[mm2gb post-pass] sort level %d: %.1f ms; %lld radix passes over %lld elements: %lld cycles, %lld steps, refills of one line %lld, of all lines %lld\n",
			        lv, t[13 + lv] / 1e5, t[24 + 6 * lv + 5], t[24 + 6 * lv + 4], t[24 + 6 * lv + 3], t[24 + 6 * lv], t[24 + 6 * lv + 1], t[24 + 6 * lv + 2]);
	if (t[21])
		fprintf(stderr, "[mm2gb post-pass] split form: %lld walk tasks, first starts at 0, last ends at %.2f ms; summed %.1f ms, longest task %.2f ms; spec loads %.1f ms, long walks %.1f ms; groups %lld, open %lld, long %lld (%lld rounds)\n",
		        t[21], (t[23] - t[22]) / 1e5, t[2] / 1e5, t[5] / 1e5, t[7] / 1e5, t[8] / 1e5, t[9], t[10], t[11], t[20]);
	if (t[6])
		fprintf(stderr, "[mm2gb post-pass] walks: spec loads %.1f ms, long walks %.1f ms; groups %lld, open %lld, long %lld (%lld rounds of 64 anchors), candidates %lld\n", t[7] / 1e5, t[8] / 1e5, t[9], t[10], t[11], t[20], t[12]);
	if (t[6] || t[21])
		fprintf(stderr, "[mm2gb post-pass] wave-time summed over reads: collect %.1f ms | sort %.1f ms | chain walks %.1f ms | emit %.1f ms  (%lld reads) | slowest read: sort %.2f ms, walks %.2f ms, whole %.2f ms\n",
		        t[0] / 1e5, t[1] / 1e5, t[2] / 1e5, t[3] / 1e5, (long long)n_reads, t[4] / 1e5, t[5] / 1e5, t[6] / 1e5);
	// the schedule: when the reads that finish last were started, and how long their parts took
	if (n_reads > 0 && post_dbg_reads.ptr) {
		std::vector<long long> tr((size_t)n_reads * 4);
		std::vector<int64_t> off((size_t)n_reads + 1);
		if (hipMemcpy(tr.data(), post_dbg_reads.ptr, tr.size() * 8, hipMemcpyDeviceToHost) == hipSuccess &&
		    hipMemcpy(off.data(), d_offsets, off.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
			long long t_first = LLONG_MAX, t_last = 0;
			for (int64_t r = 0; r < n_reads; ++r) if (tr[4 * r]) { t_first = std::min(t_first, tr[4 * r]); t_last = std::max(t_last, tr[4 * r + 3]); }
			std::vector<int64_t> idx;
			for (int64_t r = 0; r < n_reads; ++r) if (tr[4 * r]) idx.push_back(r);
			std::sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b2) { return tr[4 * a + 3] > tr[4 * b2 + 3]; });
			fprintf(stderr, "[mm2gb post-pass] k_post_chains: first read starts at 0, last ends at %.2f ms; the reads that end last (anchors | start | collect | sort | walks | end, ms):\n", (t_last - t_first) / 1e5);
			for (size_t k = 0; k < std::min<size_t>(idx.size(), 12); ++k) {
				const int64_t r = idx[k];
				fprintf(stderr, "    read %6lld  %7lld | %6.2f | %5.2f | %6.2f | %6.2f | %6.2f\n", (long long)r, (long long)(off[r + 1] - off[r]), (tr[4 * r] - t_first) / 1e5,
				        (tr[4 * r + 1] - tr[4 * r]) / 1e5, (tr[4 * r + 2] - tr[4 * r + 1]) / 1e5, (tr[4 * r + 3] - tr[4 * r + 2]) / 1e5, (tr[4 * r + 3] - t_first) / 1e5);
			}
			// reads in flight over time (how many wave slots still work at t)
			const int n_bins = 14;
			std::vector<int> busy(n_bins, 0);
			const double span = std::max(1.0, (double)(t_last - t_first));
			for (int64_t r : idx) for (int k = 0; k < n_bins; ++k) { const double at = t_first + span * (k + 0.5) / n_bins; if (tr[4 * r] <= at && at < tr[4 * r + 3]) ++busy[k]; }
			fprintf(stderr, "    reads in flight at %d points of the kernel's time:", n_bins);
			for (int k = 0; k < n_bins; ++k) fprintf(stderr, " %d", busy[k]);
			fprintf(stderr, "\n");
		}
	}
	if (t[42] > 0 && post_dbg_stasks.ptr) {
		const size_t n_t = (size_t)std::min<long long>(t[42], 262144);
		std::vector<long long> tk(n_t * 4);
		if (hipMemcpy(tk.data(), post_dbg_stasks.ptr, tk.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
			auto ph12 = [](long long v) { return (double)((v & 511) << (2 * (v >> 9 & 7))); };   // ticks of a phase as k_post_sort_level packed them
			for (int lv = 0; lv < 4; ++lv) {
				std::vector<size_t> idx;
				long long first = LLONG_MAX, last = 0, sum = 0;
				for (size_t k = 0; k < n_t; ++k) if ((int)(tk[4 * k + 2] >> 32) == lv) { idx.push_back(k); first = std::min(first, tk[4 * k]); last = std::max(last, tk[4 * k + 1]); sum += tk[4 * k + 1] - tk[4 * k]; }
				if (idx.empty()) continue;
				std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b2) { return tk[4 * a + 1] - tk[4 * a] > tk[4 * b2 + 1] - tk[4 * b2]; });
				fprintf(stderr, "[mm2gb post-pass] k_post_sort_level %d: %zu tasks, %.2f ms from the first start to the last end, %.1f ms summed; the longest (elements | start | ms):", lv, idx.size(), (last - first) / 1e5, sum / 1e5);
				for (size_t k = 0; k < std::min<size_t>(idx.size(), 6); ++k) {
					const long long phv = tk[4 * idx[k] + 3];
					fprintf(stderr, "  %lld | %.2f | %.2f (histogram %.2f, set-up %.2f, walk %.2f of %lld steps, move %.2f)", tk[4 * idx[k] + 2] & 0xffffffffLL, (tk[4 * idx[k]] - first) / 1e5, (tk[4 * idx[k] + 1] - tk[4 * idx[k]]) / 1e5,
					        ph12(phv) / 1e5, ph12(phv >> 12) / 1e5, ph12(phv >> 24) / 1e5, phv >> 48 & 0xfffff, ph12(phv >> 36) / 1e5);
				}
				{
					// by size: tasks, elements, steps, and where the waves' time went
					const long long edge[4] = { 1792, 3584, 7168, LLONG_MAX };
					const char *name[4] = { "up to 1 792 elements", "up to 3 584", "up to 7 168 (resident)", "longer" };
					for (int c = 0; c < 4; ++c) {
						long long sdt = 0, sel = 0, sst = 0, sp[4] = { 0, 0, 0, 0 }; size_t nn = 0;
						for (size_t k : idx) {
							const long long len = tk[4 * k + 2] & 0xffffffffLL, dt = tk[4 * k + 1] - tk[4 * k], phv = tk[4 * k + 3];
							if (len > edge[c] || (c > 0 && len <= edge[c - 1])) continue;
							++nn; sdt += dt; sel += len; sst += phv >> 48 & 0xfffff;
							for (int q = 0; q < 4; ++q) sp[q] += ph12(phv >> (12 * q));
						}
						if (nn) fprintf(stderr, "\n    %s: %zu tasks, %lld elements, %lld steps, %.1f ms summed (histogram %.1f, set-up %.1f, walk %.1f, move %.1f)", name[c], nn, sel, sst, sdt / 1e5, sp[0] / 1e5, sp[1] / 1e5, sp[2] / 1e5, sp[3] / 1e5);
					}
				}
				fprintf(stderr, "\n    tasks in flight at 14 points of its time:");
				for (int kb = 0; kb < 14; ++kb) { const double at = first + (double)(last - first) * (kb + 0.5) / 14; int busy = 0; for (size_t k : idx) if (tk[4 * k] <= at && at < tk[4 * k + 1]) ++busy; fprintf(stderr, " %d", busy); }
				fprintf(stderr, "\n");
			}
		}
	}
	if (t[21] > 0 && post_dbg_tasks.ptr) {
		const size_t n_t = (size_t)t[21];
		std::vector<long long> tk(n_t * 8);
		if (hipMemcpy(tk.data(), post_dbg_tasks.ptr, tk.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
			std::vector<size_t> idx(n_t);
			for (size_t k = 0; k < n_t; ++k) idx[k] = k;
			std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b2) { return tk[8 * a + 1] - tk[8 * a] > tk[8 * b2 + 1] - tk[8 * b2]; });
			fprintf(stderr, "[mm2gb post-pass] k_post_walk: the longest tasks (taken as | read | class | candidates | start | ms | of which look-ahead loads | long walks | how many | rounds):\n");
			for (size_t k = 0; k < std::min<size_t>(n_t, 16); ++k) {
				const size_t q = idx[k];
				fprintf(stderr, "    %6zu | %6lld | %2lld | %7lld | %6.2f | %6.2f | %6.2f | %6.2f | %lld | %lld\n", q, tk[8 * q + 2] >> 4, tk[8 * q + 2] & 15, tk[8 * q + 3], (tk[8 * q] - t[22]) / 1e5, (tk[8 * q + 1] - tk[8 * q]) / 1e5, tk[8 * q + 4] / 1e5, tk[8 * q + 5] / 1e5, tk[8 * q + 6], tk[8 * q + 7]);
			}
			const int n_bins = 14;
			std::vector<int> busy(n_bins, 0);
			const double span = std::max(1.0, (double)(t[23] - t[22]));
			for (size_t q = 0; q < n_t; ++q) for (int k = 0; k < n_bins; ++k) { const double at = t[22] + span * (k + 0.5) / n_bins; if (tk[8 * q] <= at && at < tk[8 * q + 1]) ++busy[k]; }
			fprintf(stderr, "    tasks in flight at %d points of the kernel's time:", n_bins);
			for (int k = 0; k < n_bins; ++k) fprintf(stderr, " %d", busy[k]);
			fprintf(stderr, "\n");
		}
	}
}

int mm2gb_rmq_chain_gpu(mm2gb_engine_t *eng, const mm2gb_rmq_param_t *prm, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                        mm2gb_chains_t *out, int32_t *n_tied, mm2gb_stats_t *stats)
{
	if (!eng || !out || !prm) return fail("mm2gb_rmq_chain_gpu: null argument");
	if (prm->max_dist < 0 || prm->bw < 0 || prm->cap_rmq_size < 0) return fail("mm2gb_rmq_chain_gpu: negative parameter");
	if (eng->e.chain_gpu(n_reads, offsets, anchors, out, prm, n_tied)) return -1;
	if (stats) *stats = eng->e.last;
	return 0;
}

int mm2gb_sort_seeds_gpu(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *offsets, mm2gb_anchor_t *anchors)
{
	return eng ? eng->e.sort_seeds(n_reads, offsets, anchors) : fail("mm2gb: null engine");
}

int mm2gb_collect_seeds_gpu(mm2gb_engine_t *eng, int64_t opt_flag, int64_t n_reads, const int64_t *seed_off, const mm2gb_seed_t *seeds,
                            const int64_t *hit_off, const uint64_t *hits, const int32_t *qlen, const int32_t *q_rank,
                            int32_t n_ref, const int32_t *ref_len, const int32_t *ref_rank, int64_t *anchor_off, mm2gb_anchor_t *anchors)
{
	return eng ? eng->e.collect_seeds(opt_flag, n_reads, seed_off, seeds, hit_off, hits, qlen, q_rank, n_ref, ref_len, ref_rank, anchor_off, anchors) : fail("mm2gb: null engine");
}

int mm2gb_gen_regs_gpu(mm2gb_engine_t *eng, int64_t n_reads, const mm2gb_chains_t *chains, const int32_t *qlen, const uint32_t *hash,
                       int is_qstrand, mm2gb_reg_t *regs)
{
	return eng ? eng->e.gen_regs(n_reads, chains, qlen, hash, is_qstrand, regs) : fail("mm2gb: null engine");
}

int mm2gb_post_device(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *d_offsets, const mm2gb_anchor_t *d_anchors, int64_t n_anchors,
                      const int32_t *d_f, const int32_t *d_p, int64_t *n_chains, int64_t *n_kept, float *ms)
{
	if (!eng) return fail("mm2gb: null engine");
	Engine &e = eng->e;
	MM2GB_HIP(hipSetDevice(e.device));
	if (e.enqueue_post(n_reads, d_offsets, d_anchors, n_anchors, d_f, d_p)) return -1;
	MM2GB_HIP(hipStreamSynchronize(e.stream));
	if (n_chains) *n_chains = e.h_post_totals[0];
	if (n_kept) *n_kept = e.h_post_totals[1];
	if (e.debug_phases) e.print_post_debug(n_reads, d_offsets);
	float t = 0;
	if (ms && hipEventElapsedTime(&t, e.post0, e.post1) == hipSuccess) *ms = t;
	return 0;
}

// the same, enqueued only: nothing is waited for (mm2gb_engine_sync does; the totals are read with mm2gb_post_device_totals afterwards).  For a
// caller that runs the post-pass of batch k on one engine beside the score kernels of batch k+1 on another.
int mm2gb_post_device_enqueue(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *d_offsets, const mm2gb_anchor_t *d_anchors, int64_t n_anchors,
                              const int32_t *d_f, const int32_t *d_p)
{
	if (!eng) return fail("mm2gb: null engine");
	MM2GB_HIP(hipSetDevice(eng->e.device));
	return eng->e.enqueue_post(n_reads, d_offsets, d_anchors, n_anchors, d_f, d_p);
}

int mm2gb_post_device_totals(mm2gb_engine_t *eng, int64_t *n_chains, int64_t *n_kept, float *ms)
{
	if (!eng) return fail("mm2gb: null engine");
	Engine &e = eng->e;
	if (!e.h_post_totals) return fail("mm2gb_post_device_totals: no post-pass has run on this engine");
	MM2GB_HIP(hipSetDevice(e.device));
	MM2GB_HIP(hipStreamSynchronize(e.stream));
	if (n_chains) *n_chains = e.h_post_totals[0];
	if (n_kept) *n_kept = e.h_post_totals[1];
	float t = 0;
	if (ms && hipEventElapsedTime(&t, e.post0, e.post1) == hipSuccess) *ms = t;
	return 0;
}

// A digest of what the last post-pass on this engine left on the device (u_off, a_off, u[], a[] of result set 0), for comparing two builds or two
// settings of the post-pass on batches too large to take through the oracle: copied to the host and folded there (position-dependent, so a
// permutation of chains or anchors changes it).  digest[0..3]: offsets of chains, offsets of anchors, chains, anchors.
// for tests and debugging: the scores and predecessor distances (i - p, 0 = none) the engine's LAST mm2gb_chain_gpu / mm2gb_rmq_chain_gpu call left on the device
int mm2gb_debug_last_fill(mm2gb_engine_t *eng, int64_t n, int32_t *f, int32_t *p)
{
	if (!eng || n < 0 || (n > 0 && (!f || !p))) return fail("mm2gb_debug_last_fill: null argument");
	Engine &e = eng->e;
	const IoSet &s = e.io[(e.io_seq - 1) & 1];
	if (e.io_seq == 0 || !s.f.ptr || !s.p.ptr) return fail("mm2gb_debug_last_fill: no call has run on this engine");
	if (hipSetDevice(e.device) != hipSuccess || hipMemcpy(f, s.f.ptr, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(p, s.p.ptr, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess)
		return fail("mm2gb_debug_last_fill: copy failed");
	return 0;
}

int mm2gb_post_device_digest(mm2gb_engine_t *eng, int64_t n_reads, uint64_t *digest)
{
	if (!eng || !digest) return fail("mm2gb: null argument");
	Engine &e = eng->e;
	if (!e.h_post_totals) return fail("mm2gb_post_device_digest: no post-pass has run on this engine");
	MM2GB_HIP(hipSetDevice(e.device));
	MM2GB_HIP(hipStreamSynchronize(e.stream));
	const int64_t n_u = e.h_post_totals[0], n_a = e.h_post_totals[1];
	auto fold = [](const void *dev, size_t words, uint64_t &out) -> int {
		std::vector<uint64_t> h(words ? words : 1);
		if (words) MM2GB_HIP(hipMemcpy(h.data(), dev, words * 8, hipMemcpyDeviceToHost));
		uint64_t acc = 0x9E3779B97F4A7C15ULL;
		for (size_t k = 0; k < words; ++k) { uint64_t v = h[k] + 0x9E3779B97F4A7C15ULL * (k + 1); v ^= v >> 29; v *= 0xBF58476D1CE4E5B9ULL; v ^= v >> 32; acc += v; }
		out = acc;
		return 0;
	};
	if (fold(e.post_uoff.ptr, (size_t)n_reads + 1, digest[0]) || fold(e.post_aoff.ptr, (size_t)n_reads + 1, digest[1]) ||
	    fold(e.post_uout.ptr, (size_t)n_u, digest[2]) || fold(e.post_aout.ptr, (size_t)n_a * 2, digest[3])) return -1;
	return 0;
}

int mm2gb_engine_stats(mm2gb_engine_t *eng, mm2gb_stats_t *stats)
{
	if (!eng || !stats) return fail("mm2gb: null argument");
	*stats = eng->e.last;
	return 0;
}

int mm2gb_score_host(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                     int32_t *f, int32_t *p, mm2gb_stats_t *stats)
{
	if (!eng || !offsets || n_reads < 0) return fail("mm2gb_score_host: null argument");
	if (offsets[0] != 0) return fail("mm2gb_score_host: offsets[0] must be 0");
	if (eng->e.score_host(n_reads, offsets, anchors, f, p, nullptr)) return -1;
	if (stats) *stats = eng->e.last;
	return 0;
}

} // extern "C"
