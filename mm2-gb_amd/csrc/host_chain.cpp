// host_chain.cpp -- host post-pass of the chaining path: chain extraction from (f, p) and compaction.
// Behaviour follows mg_chain_backtrack / compact_a (lchain.c:27-111); the ordering of equal scores follows
// radix_sort_128x (ksort.h:98-151).  Written for reuse across reads by a pool of worker threads: all temporaries live in
// a BacktrackScratch, only u[] and the compacted a[] are allocated per read (from the host's arena when there is one).
#include <cstdlib>
#include <cstring>
#include <limits.h>
#include "host_chain.h"
#include "engine.h"

// Host allocator of minimap2 (kalloc.h:15-18).  Weak: present when linked into the reference host, absent otherwise.
extern "C" {
void *kmalloc(void *km, size_t size) __attribute__((weak));
void  kfree(void *km, void *ptr) __attribute__((weak));
}

namespace mm2gb {

bool host_kalloc_present() { return kmalloc != nullptr && kfree != nullptr; }

void *HostAlloc::alloc(size_t bytes) const
{
	if (use_kalloc) return kmalloc(km, bytes);
	return malloc(bytes ? bytes : 1);
}
void HostAlloc::release(void *ptr) const
{
	if (!ptr) return;
	if (use_kalloc) kfree(km, ptr); else free(ptr);
}

// ---- the host's sort order --------------------------------------------------------------------------------
namespace {

constexpr size_t SMALL_RUN = 64;   // RS_MIN_SIZE, ksort.h:98

inline void insertion_run(mm2gb_anchor_t *v, size_t lo, size_t hi)   // ksort.h:105-115
{
	for (size_t i = lo + 1; i < hi; ++i) {
		if (v[i].x < v[i - 1].x) {
			const mm2gb_anchor_t hold = v[i];
			size_t j = i;
			while (j > lo && hold.x < v[j - 1].x) { v[j] = v[j - 1]; --j; }
			v[j] = hold;
		}
	}
}

// One most-significant-byte-first pass with in-place cycle permutation, then recursion (ksort.h:116-146).
void flag_pass(mm2gb_anchor_t *v, size_t lo, size_t hi, int shift)
{
	size_t head[256], tail[256], first[256];
	size_t count[256] = { 0 };
	for (size_t i = lo; i < hi; ++i) ++count[(v[i].x >> shift) & 255];
	size_t at = lo;
	for (int k = 0; k < 256; ++k) { first[k] = head[k] = at; at += count[k]; tail[k] = at; }
	for (int k = 0; k < 256;) {
		if (head[k] == tail[k]) { ++k; continue; }
		int dst = (int)((v[head[k]].x >> shift) & 255);
		if (dst == k) { ++head[k]; continue; }
		mm2gb_anchor_t carry = v[head[k]];
		do {
			const mm2gb_anchor_t moved = carry;
			carry = v[head[dst]];
			v[head[dst]++] = moved;
			dst = (int)((carry.x >> shift) & 255);
		} while (dst != k);
		v[head[k]++] = carry;
	}
	if (shift == 0) return;
	const int next = shift > 8 ? shift - 8 : 0;
	for (int k = 0; k < 256; ++k) {
		const size_t len = tail[k] - first[k];
		if (len > SMALL_RUN) flag_pass(v, first[k], tail[k], next);
		else if (len > 1) insertion_run(v, first[k], tail[k]);
	}
}

} // namespace

void sort_by_x_like_host(mm2gb_anchor_t *beg, mm2gb_anchor_t *end)
{
	const size_t n = (size_t)(end - beg);
	if (n <= SMALL_RUN) { insertion_run(beg, 0, n); return; }   // ksort.h:149
	// The host starts at the top key byte (shift 56).  A pass in which every key has the same byte moves nothing (one
	// non-empty bucket: every element is already "in place") and recurses on the whole range with the next byte, so the
	// outcome is the same as starting at the highest byte in which the keys actually differ.
	uint64_t all_or = 0, all_and = ~(uint64_t)0;
	for (size_t i = 0; i < n; ++i) { all_or |= beg[i].x; all_and &= beg[i].x; }
	const uint64_t diff = all_or ^ all_and;
	if (diff == 0) return;                                       // all keys equal: no pass moves anything
	int shift = 56;
	while (shift > 0 && ((diff >> shift) & 255) == 0) shift -= 8;
	flag_pass(beg, 0, n, shift);
}

// ---- backtrack ----------------------------------------------------------------------------------------------
namespace {

inline int64_t pred_of(const int32_t *p_rel, int64_t i) { return p_rel[i] ? i - p_rel[i] : -1; }

// lchain.c:9-25: walk back from the chain end until an anchor that is taken, the start of the path, or an X-drop of more
// than max_drop below the best prefix; returns where the kept part stops.  The host marks the nodes it visits (t = 2) and
// walks the same links again to clear the marks, and its caller walks them a third time to collect the chain; here the
// visited nodes are remembered in `path` (in walk order), so the links are chased once.
int64_t kept_until(int32_t max_drop, int32_t top_score, int64_t start, const int32_t *f, const int32_t *p_rel, const uint8_t *mark,
                   std::vector<int32_t> &path)
{
	int64_t i = start, best_i = start;
	int32_t best = 0;
	path.clear();
	if (i < 0 || mark[i] != 0) return i;
	do {
		path.push_back((int32_t)i);                 // the host's t[i] = 2: a node is never reached twice within one walk
		i = pred_of(p_rel, i);
		const int32_t s = i < 0 ? top_score : top_score - f[i];
		if (s > best) { best = s; best_i = i; }
		else if (best - s > max_drop) break;
	} while (i >= 0 && mark[i] == 0);
	return best_i;
}

} // namespace

int backtrack_order(const mm2gb_misc_t &misc, int64_t n, const mm2gb_anchor_t *a, const int32_t *f, const int32_t *p_rel,
                    BacktrackScratch &ws, int64_t *n_kept)
{
	*n_kept = 0;
	if (n <= 0) return 0;
	const int32_t min_sc = misc.min_score, min_cnt = misc.min_cnt;
	const int32_t max_drop = misc.is_cdna ? INT_MAX : misc.bw;                 // lchain.c:151,162

	// candidates: anchors scoring at least min_sc, ordered by score the way the host orders them (lchain.c:35-41)
	size_t n_z = 0;
	for (int64_t i = 0; i < n; ++i) n_z += f[i] >= min_sc;
	if (n_z == 0) return 0;
	ws.keyed.resize(n_z);
	n_z = 0;
	for (int64_t i = 0; i < n; ++i)
		if (f[i] >= min_sc) ws.keyed[n_z++] = mm2gb_anchor_t{ (uint64_t)(int64_t)f[i], (uint64_t)i };
	sort_by_x_like_host(ws.keyed.data(), ws.keyed.data() + ws.keyed.size());

	ws.mark.assign((size_t)n, 0);
	ws.picked.clear();
	ws.chains.clear();
	// best-scoring end first; every anchor walked is consumed even if its chain is dropped (lchain.c:59-71)
	for (int64_t k = (int64_t)ws.keyed.size() - 1; k >= 0; --k) {
		const int64_t start = (int64_t)ws.keyed[k].y;
		if (ws.mark[start] != 0) continue;
		const int32_t top = (int32_t)ws.keyed[k].x;
		const size_t n_before = ws.picked.size();
		const int64_t stop = kept_until(max_drop, top, start, f, p_rel, ws.mark.data(), ws.path);
		// the chain: visited nodes from the end up to (not including) `stop`; `stop` is a visited node, or the node the
		// walk ended on (taken / none), in which case every visited node belongs to the chain
		int64_t i = stop;
		for (size_t q = 0; q < ws.path.size(); ++q) {
			if (ws.path[q] == stop) break;
			ws.picked.push_back(ws.path[q]); ws.mark[ws.path[q]] = 1;
		}
		const int32_t sc = i < 0 ? top : top - f[i];
		const size_t cnt = ws.picked.size() - n_before;
		if (sc >= min_sc && cnt > 0 && (int64_t)cnt >= min_cnt) ws.chains.push_back((uint64_t)sc << 32 | (uint64_t)cnt);
		else ws.picked.resize(n_before);
	}
	const int n_u = (int)ws.chains.size();
	if (n_u == 0) return 0;

	// order of compaction (lchain.c:84-110): chains by the x of their first anchor.  The host packs the anchors into a
	// temporary and copies again after sorting; here the order is decided first and every anchor is copied once.
	ws.heads.resize((size_t)n_u);
	size_t k = 0;
	for (int c = 0; c < n_u; ++c) {
		const size_t cnt = (size_t)(uint32_t)ws.chains[c];
		ws.heads[c].x = a[ws.picked[k + cnt - 1]].x;            // first anchor of the chain = last one picked
		ws.heads[c].y = (uint64_t)k << 32 | (uint64_t)c;
		k += cnt;
	}
	sort_by_x_like_host(ws.heads.data(), ws.heads.data() + n_u);
	*n_kept = (int64_t)ws.picked.size();
	return n_u;
}

void emit_chain_list(const BacktrackScratch &ws, uint64_t *u)
{
	for (size_t c = 0; c < ws.heads.size(); ++c) u[c] = ws.chains[(size_t)(uint32_t)ws.heads[c].y];
}

void emit_anchor_order(const BacktrackScratch &ws, int32_t *idx)
{
	size_t k = 0;
	for (size_t c = 0; c < ws.heads.size(); ++c) {
		const size_t cnt = (size_t)(uint32_t)ws.chains[(size_t)(uint32_t)ws.heads[c].y], k0 = (size_t)(ws.heads[c].y >> 32);
		for (size_t j = 0; j < cnt; ++j) idx[k + j] = ws.picked[k0 + (cnt - j - 1)];          // each chain start -> end
		k += cnt;
	}
}

int backtrack_compact(const mm2gb_misc_t &misc, int64_t n, const mm2gb_anchor_t *a, const int32_t *f, const int32_t *p_rel,
                      const HostAlloc &mem, BacktrackScratch &ws, uint64_t **u_out, mm2gb_anchor_t **a_out)
{
	*u_out = nullptr; *a_out = nullptr;
	int64_t n_v = 0;
	const int n_u = backtrack_order(misc, n, a, f, p_rel, ws, &n_v);
	if (n_u == 0) return 0;
	uint64_t *u = (uint64_t*)mem.alloc((size_t)n_u * sizeof(uint64_t));
	mm2gb_anchor_t *out = (mm2gb_anchor_t*)mem.alloc((size_t)n_v * sizeof(mm2gb_anchor_t));
	emit_chain_list(ws, u);
	size_t k = 0;
	for (int c = 0; c < n_u; ++c) {
		const size_t cnt = (size_t)(uint32_t)u[c], k0 = (size_t)(ws.heads[c].y >> 32);
		for (size_t j = 0; j < cnt; ++j) out[k + j] = a[ws.picked[k0 + (cnt - j - 1)]];
		k += cnt;
	}
	*u_out = u; *a_out = out;
	return n_u;
}

} // namespace mm2gb

using namespace mm2gb;

extern "C" {

int mm2gb_backtrack_host(const mm2gb_misc_t *misc, int64_t n, const mm2gb_anchor_t *a, const int32_t *f, const int32_t *p_rel,
                         uint64_t **u_out, mm2gb_anchor_t **a_out)
{
	if (!misc || !u_out || !a_out || (n > 0 && (!a || !f || !p_rel))) return fail("mm2gb_backtrack_host: null argument");
	for (int64_t i = 0; i < n; ++i)
		if (p_rel[i] < 0 || p_rel[i] > i) return fail("mm2gb_backtrack_host: predecessor offset out of range at anchor " + std::to_string(i));
	BacktrackScratch ws;
	HostAlloc mem;   // libc: the caller frees with mm2gb_free
	return backtrack_compact(*misc, n, a, f, p_rel, mem, ws, u_out, a_out);
}

void mm2gb_free(void *ptr) { free(ptr); }

} // extern "C"
