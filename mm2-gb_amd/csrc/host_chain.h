// host_chain.h -- internal: host-side post-pass (backtrack + compaction) and allocator plumbing.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <new>
#include <vector>
#include "../../include/mm2gb_chain.h"

namespace mm2gb {

// Where results that outlive the call are allocated: the host's kalloc arena when the library is linked into
// minimap2 (kmalloc/kfree, kalloc.h:15-18), libc otherwise.
struct HostAlloc {
	void *km = nullptr;
	bool use_kalloc = false;
	void *alloc(size_t bytes) const;
	void release(void *ptr) const;
};
bool host_kalloc_present();

// Reusable scratch for one worker thread.
struct BacktrackScratch {
	std::vector<mm2gb_anchor_t> keyed;     // (score, index) pairs to sort -- the reference's z[] (lchain.c:38-41)
	std::vector<uint8_t> mark;             // the reference's t[] (lchain.c:43), a byte per anchor is enough (values 0,1,2)
	std::vector<int32_t> picked;           // the reference's v[] (lchain.c:65)
	std::vector<int32_t> path;             // nodes visited by the current walk
	std::vector<uint64_t> chains;          // u[] before it is copied out
	std::vector<mm2gb_anchor_t> heads;     // (first x, offset<<32|chain) to order chains (lchain.c:94-99)
};

// Sort by .x exactly as radix_sort_128x does (ksort.h:98-151 instantiated at misc.c:167-168): equal keys end up in
// that implementation's order, which decides chain priority.
void sort_by_x_like_host(mm2gb_anchor_t *beg, mm2gb_anchor_t *end);

// Backtrack (lchain.c:27-76) and the order of compaction (lchain.c:78-111) for one read, left in `ws`: returns the number of
// chains, *n_kept = anchors in them.  emit_chain_list writes u[] (score<<32 | count) in output order; emit_anchor_order writes,
// for every output position, the index of the input anchor that goes there.  `a` is only read.
int  backtrack_order(const mm2gb_misc_t &misc, int64_t n, const mm2gb_anchor_t *a, const int32_t *f, const int32_t *p_rel,
                     BacktrackScratch &ws, int64_t *n_kept);
void emit_chain_list(const BacktrackScratch &ws, uint64_t *u);
void emit_anchor_order(const BacktrackScratch &ws, int32_t *idx);

// Both steps and the copy, results allocated from `mem`.
// p_rel[i] = i - predecessor, 0 = none.  Returns the number of chains; *u_out / *a_out come from `mem`
// (both NULL when there is no chain).  `a` is only read.
int backtrack_compact(const mm2gb_misc_t &misc, int64_t n, const mm2gb_anchor_t *a, const int32_t *f, const int32_t *p_rel,
                      const HostAlloc &mem, BacktrackScratch &ws, uint64_t **u_out, mm2gb_anchor_t **a_out);

// CPUs this process may use at once: affinity mask and cgroup quota, whichever is smaller (stream_api.cpp)
int usable_cpus();

// A large result array (kept anchors: a gigabyte per batch): huge pages where the system offers them on request -- first touch is most of what
// filling it costs, and giving it back is cheaper too; free() applies as to any malloc'd block.
inline void *result_alloc(size_t bytes)
{
	void *mem = nullptr;
	if (bytes >= ((size_t)64 << 20) && posix_memalign(&mem, (size_t)2 << 20, bytes) == 0) { (void)madvise(mem, bytes, MADV_HUGEPAGE); return mem; }
	return malloc(bytes ? bytes : 1);
}

// Results that the DEVICE writes (the chains of a large batch, D2H): a page-locked block out of a small process-wide cache (engine.hip) --
// a fresh gigabyte costs its first touch and its pinning, a block that mm2gb_chains_free gave back costs nothing and is copied into at the
// link's rate.  Falls back to result_alloc when nothing can be registered.  result_release: a cached block goes back to the cache
// (MM2GB_RESULT_CACHE_MB, default 8192, of idle blocks are kept), anything else to free().
void *result_alloc_pinned(size_t bytes);
void  result_release(void *ptr);

// A host thread's large scratch array, kept from call to call (static thread_local at its users): grows, never shrinks, nothing initialised.
template <class T> struct BigBuf {
	T *p = nullptr;
	size_t n = 0, cap = 0;
	BigBuf() {}
	BigBuf(const BigBuf&) = delete;
	BigBuf &operator=(const BigBuf&) = delete;
	~BigBuf() { free(p); }
	void release() { free(p); p = nullptr; n = cap = 0; }
	void resize(size_t want) { if (want > cap) { free(p); cap = want + want / 8; p = (T*)result_alloc(cap * sizeof(T)); if (!p) { cap = n = 0; throw std::bad_alloc(); } } n = want; }
	T *data() { return p; }
	const T *data() const { return p; }
	T *begin() { return p; }
	size_t size() const { return n; }
};

// mm2gb_collect_matches (seeding.cpp) without the copy of the occurrences: refs[s] points at kept seed s's occurrences in the index, out->hits stays null
int collect_matches_refs(const mm2gb_index_t *ix, const char *seq, int32_t len, const mm2gb_seed_opt_t *opt, mm2gb_matches_t *out, std::vector<const uint64_t*> *refs);

// The large host arrays of an engine's mapping and re-chaining calls, kept from call to call (first touch of a gigabyte of fresh pages costs
// more than filling it) and owned by the engine: matches, anchors, the re-chained reads' anchors, the spliced chains (mapper.cpp); the gathers of
// the hybrid re-chaining call -- host share, device share, tie redo (rmq_hybrid.cpp).  host_scratch() makes it on first use (mapper.cpp);
// mm2gb_engine_release_host_scratch gives the memory back, mm2gb_engine_destroy the object.
struct HostScratch {
	BigBuf<uint64_t> hits, nu;
	BigBuf<mm2gb_anchor_t> anchors, ra, nc;
	BigBuf<mm2gb_anchor_t> gather[3];
	void release() { hits.release(); nu.release(); anchors.release(); ra.release(); nc.release(); for (auto &g : gather) g.release(); }
};
HostScratch &host_scratch(mm2gb_engine_t *eng);

// a result set that is freed on every way out of its scope
struct ChainsOwner {
	mm2gb_chains_t c;
	ChainsOwner() { memset(&c, 0, sizeof c); }
	ChainsOwner(const ChainsOwner&) = delete;
	ChainsOwner &operator=(const ChainsOwner&) = delete;
	~ChainsOwner() { mm2gb_chains_free(&c); }
	mm2gb_chains_t release() { const mm2gb_chains_t r = c; memset(&c, 0, sizeof c); return r; }
};

// mm2gb_rmq_chain (rmq_hybrid.cpp) without its last step: the results stay where the three sides left them -- host threads, device, reads
// redone after a tie -- and read r's chains are chains[which[r]] at position slot[r].  For a caller that copies them on anyway (the mapper
// splices them into its own arrays: one copy of a gigabyte instead of two).
struct RmqParts {
	mm2gb_chains_t chains[3];
	std::vector<unsigned char> which;
	std::vector<int64_t> slot;
	RmqParts() { memset(chains, 0, sizeof chains); }
	~RmqParts() { for (mm2gb_chains_t &c : chains) mm2gb_chains_free(&c); }
	RmqParts(const RmqParts&) = delete;
	RmqParts &operator=(const RmqParts&) = delete;
};
int rmq_chain_parts(mm2gb_engine_t *eng, const mm2gb_rmq_param_t *prm, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                    int n_threads, RmqParts &parts, int32_t *where, mm2gb_rmq_deal_t *deal);

} // namespace mm2gb
