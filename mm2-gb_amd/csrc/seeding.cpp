// seeding.cpp -- from sequence to seed matches, written from scratch (the producer side of the path).
//
//   sketch            (w,k)-minimizers of a sequence, the reference's definition to the letter (sketch.c:77-143, non-HPC): the hash of
//                     the canonical k-mer (sketch.c:29-39), ties inside a window kept, the special first window, ambiguous bases
//   Index             every minimizer of the reference sequences -> its occurrences  rid << 32 | last_pos << 1 | strand  in ascending
//                     order: what mm_idx_get hands out (index.c:81-98; the order is that of radix_sort_64 in index.c:251), and the
//                     occurrence threshold mid_occ of mm_mapopt_update (options.c:78-84 with index.c:186-211)
//   collect_matches   mm_collect_matches (seed.c:98-131) for one read: query minimizers, the over-represented ones dropped
//                     (mm_seed_mz_flt, seed.c:5-30), every minimizer the index knows (mm_seed_collect_all, seed.c:32-54), the high-occurrence
//                     ones thinned out (mm_seed_select, seed.c:58-96), repeat length and minimizer positions
// The output of collect_matches is the input of mm2gb_collect_seeds_gpu.  One query segment per read (n_segs = 1).
// Host code: an index look-up per minimizer is a pointer chase through a structure the size of the genome -- not a kernel.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "engine.h"
#include "host_chain.h"

namespace mm2gb {
namespace {

struct Mini { uint64_t x, y; };            // x = hash << 8 | span, y = rid << 32 | last_pos << 1 | strand   (sketch.c:66-71)

inline int base_code(unsigned char c)      // A C G T (either case, U as T) -> 0..3, anything else 4
{
	switch (c) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': case 'U': case 'u': return 3;
	default: return 4;
	}
}

// invertible integer hash of a 2k-bit k-mer (sketch.c:29-39)
inline uint64_t mix(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key ^= key >> 24;
	key = (key + (key << 3) + (key << 8)) & mask;
	key ^= key >> 14;
	key = (key + (key << 2) + (key << 4)) & mask;
	key ^= key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

// sketch.c:77-143 without homopolymer compression.  A ring of the last w k-mers (slot = k-mer number mod w); the current minimum and
// its slot; what is emitted when, in this order: the k-mers equal to the minimum of the FIRST full window; the old minimum when a
// k-mer at least as small arrives or when it leaves the window; after a rescan, the other k-mers equal to the new minimum.
void sketch(const char *seq, int len, int w, int k, uint32_t rid, std::vector<Mini> &out)
{
	const uint64_t none = ~0ull, mask = (1ull << 2 * k) - 1, top_shift = 2 * (k - 1);
	uint64_t fwd = 0, rev = 0;
	std::vector<Mini> ring((size_t)w, Mini{ none, none });
	Mini best{ none, none };
	int run = 0, slot = 0, best_slot = 0;       // run: valid bases since the last ambiguous one (k-mers ending here: run - k + 1)
	for (int i = 0; i < len; ++i) {
		const int c = base_code((unsigned char)seq[i]);
		Mini cur{ none, none };
		if (c < 4) {
			const int span = run + 1 < k ? run + 1 : k;
			fwd = (fwd << 2 | (uint64_t)c) & mask;
			rev = rev >> 2 | (uint64_t)(3 ^ c) << top_shift;
			if (fwd == rev) continue;               // its own reverse complement: no strand, and it does not count as a base seen
			const int strand = fwd < rev ? 0 : 1;
			++run;
			if (run >= k && span < 256) {
				cur.x = mix(strand ? rev : fwd, mask) << 8 | (uint64_t)span;
				cur.y = (uint64_t)rid << 32 | (uint64_t)(uint32_t)i << 1 | (uint64_t)strand;
			}
		} else run = 0;
		ring[(size_t)slot] = cur;
		auto twins = [&](int from, int to) {     // k-mers in slots [from, to) with the minimum's value at another position
			for (int j = from; j < to; ++j)
				if (ring[(size_t)j].x == best.x && ring[(size_t)j].y != best.y) out.push_back(ring[(size_t)j]);
		};
		if (run == w + k - 1 && best.x != none) { twins(slot + 1, w); twins(0, slot); }     // the first full window
		if (cur.x <= best.x) {                     // at least as small: the old minimum is done
			if (run >= w + k && best.x != none) out.push_back(best);
			best = cur; best_slot = slot;
		} else if (slot == best_slot) {            // the minimum leaves the window
			if (run >= w + k - 1 && best.x != none) out.push_back(best);
			best.x = none;
			for (int j = slot + 1; j < w; ++j) if (ring[(size_t)j].x <= best.x) { best = ring[(size_t)j]; best_slot = j; }   // oldest first, later wins ties
			for (int j = 0; j <= slot; ++j) if (ring[(size_t)j].x <= best.x) { best = ring[(size_t)j]; best_slot = j; }
			if (run >= w + k - 1 && best.x != none) { twins(slot + 1, w); twins(0, slot + 1); }
		}
		if (++slot == w) slot = 0;
	}
	if (best.x != none) out.push_back(best);    // sketch.c:141-142
}

} // namespace

struct SeedIndex {
	int k = 15, w = 10;
	std::vector<int32_t> lens;
	std::vector<uint64_t> keys;             // distinct minimizers (x >> 8), ascending
	std::vector<int64_t> first;             // keys.size() + 1: where each one's occurrences begin
	std::vector<uint64_t> where;            // occurrences, ascending within a minimizer
	// keys by their top bits: bucket[b] = first key with (key >> bucket_shift) >= b.  The hash spreads minimizers evenly over their 2k bits,
	// so a bucket holds a key or two and a look-up is one probe of this table and a search among those (a binary search over all keys:
	// ~20 dependent probes of a table that does not fit the cache, ~200 ns per minimizer of a read, most of what seeding cost)
	std::vector<uint32_t> bucket;
	int bucket_shift = 0;
	void build_buckets()
	{
		int bits = 1;
		while (bits < 2 * k && ((size_t)1 << bits) < keys.size()) ++bits;   // about one key per bucket
		bits = std::min(bits, 26);
		bucket_shift = 2 * k - bits;
		bucket.assign(((size_t)1 << bits) + 1, 0);
		size_t at = 0;
		for (size_t b = 0; b <= (size_t)1 << bits; ++b) {
			while (at < keys.size() && (keys[at] >> bucket_shift) < b) ++at;
			bucket[b] = (uint32_t)at;
		}
	}
	const uint64_t *find(uint64_t minier, int *n) const
	{
		const uint64_t b = minier >> bucket_shift;
		if (b + 1 >= bucket.size()) { *n = 0; return nullptr; }
		const auto lo = keys.begin() + bucket[(size_t)b], hi = keys.begin() + bucket[(size_t)b + 1];
		const auto it = std::lower_bound(lo, hi, minier);
		if (it == hi || *it != minier) { *n = 0; return nullptr; }
		const size_t at = (size_t)(it - keys.begin());
		*n = (int)(first[at + 1] - first[at]);
		return where.data() + first[at];
	}
};

namespace {

// max-heap of 64-bit values, sift-down as in ksort.h:43-59 (which element sits where decides which seed is replaced, seed.c:80-85)
void sift_down(size_t i, size_t n, uint64_t *h)
{
	const uint64_t v = h[i];
	for (size_t c = 2 * i + 1; c < n; c = 2 * i + 1) {
		if (c + 1 < n && h[c] < h[c + 1]) ++c;
		if (h[c] < v) break;
		h[i] = h[c]; i = c;
	}
	h[i] = v;
}

struct Match { uint32_t n, q_pos, q_span, seg_id; bool flt, tandem; const uint64_t *cr; };

// seed.c:58-96: inside every streak of minimizers that occur more than max_occ times, keep the max_high_occ rarest -- one per
// `dist` bases of the streak -- and never one that occurs more than max_max_occ times
void thin_out(std::vector<Match> &m, int qlen, int max_occ, int max_max_occ, int dist)
{
	const int n = (int)m.size();
	if (n < 2) return;
	bool any = false;
	for (const Match &q : m) any |= q.n > (uint32_t)max_occ;
	if (!any) return;
	uint64_t heap[128];
	for (int i = 0, last_low = -1; i <= n; ++i) {
		if (i < n && m[(size_t)i].n > (uint32_t)max_occ) continue;
		if (i - last_low > 1) {                    // the streak (last_low, i)
			const int from = last_low + 1, to = i;
			const int ps = last_low < 0 ? 0 : (int)(m[(size_t)last_low].q_pos >> 1), pe = i == n ? qlen : (int)(m[(size_t)i].q_pos >> 1);
			int keep = (int)((double)(pe - ps) / dist + .499);
			if (keep > 0) {
				if (keep > 128) keep = 128;
				int j = from, filled = 0;
				for (; j < to && filled < keep; ++j, ++filled) heap[filled] = (uint64_t)m[(size_t)j].n << 32 | (uint32_t)j;
				for (size_t s = (size_t)filled / 2; s-- > 0;) sift_down(s, (size_t)filled, heap);
				for (; j < to; ++j)
					if ((int32_t)m[(size_t)j].n < (int32_t)(heap[0] >> 32)) { heap[0] = (uint64_t)m[(size_t)j].n << 32 | (uint32_t)j; sift_down(0, (size_t)filled, heap); }
				for (int s = 0; s < filled; ++s) m[(size_t)(uint32_t)heap[s]].flt = true;
			}
			for (int j = from; j < to; ++j) m[(size_t)j].flt = !m[(size_t)j].flt;
			for (int j = from; j < to; ++j) if (m[(size_t)j].n > (uint32_t)max_max_occ) m[(size_t)j].flt = true;
		}
		last_low = i;
	}
}

} // namespace
} // namespace mm2gb

using namespace mm2gb;

extern "C" {

int mm2gb_sketch(const char *seq, int32_t len, int w, int k, uint32_t rid, uint64_t **out_xy, int64_t *n_out)
{
	if (!seq || !out_xy || !n_out || len < 0 || w < 1 || w > 255 || k < 1 || k > 28) return fail("mm2gb_sketch: bad arguments (0 < w < 256, 0 < k <= 28)");
	std::vector<Mini> v;
	if (len > 0) sketch(seq, len, w, k, rid, v);
	*n_out = (int64_t)v.size();
	*out_xy = (uint64_t*)malloc((v.size() + 1) * 16);
	if (!*out_xy) return fail("mm2gb_sketch: out of memory");
	if (!v.empty()) memcpy(*out_xy, v.data(), v.size() * 16);
	return 0;
}

mm2gb_index_t *mm2gb_index_build(int k, int w, int32_t n_seq, const char *const *seqs, const int32_t *lens, int n_threads)
{
	if (n_seq < 0 || (n_seq > 0 && (!seqs || !lens)) || w < 1 || w > 255 || k < 1 || k > 28) { fail("mm2gb_index_build: bad arguments (0 < w < 256, 0 < k <= 28)"); return nullptr; }
	SeedIndex *ix = new SeedIndex;
	ix->k = k; ix->w = w;
	ix->lens.assign(lens, lens + n_seq);
	std::vector<std::vector<Mini>> per((size_t)n_seq);
	std::atomic<int32_t> next(0);
	auto work = [&]() { for (;;) { const int32_t s = next.fetch_add(1); if (s >= n_seq) break; if (lens[s] > 0) sketch(seqs[s], lens[s], w, k, (uint32_t)s, per[(size_t)s]); } };
	if (n_threads < 2) work();
	else { std::vector<std::thread> pool; for (int t = 0; t < n_threads; ++t) pool.emplace_back(work); for (auto &th : pool) th.join(); }
	size_t total = 0;
	for (auto &v : per) total += v.size();
	std::vector<Mini> all;
	all.reserve(total);
	for (auto &v : per) { all.insert(all.end(), v.begin(), v.end()); std::vector<Mini>().swap(v); }
	// by minimizer (span apart: x >> 8, index.c:229), then by occurrence (index.c:251)
	std::sort(all.begin(), all.end(), [](const Mini &a, const Mini &b) { return (a.x >> 8) != (b.x >> 8) ? (a.x >> 8) < (b.x >> 8) : a.y < b.y; });
	ix->where.resize(all.size());
	for (size_t i = 0; i < all.size(); ++i) {
		if (i == 0 || (all[i].x >> 8) != (all[i - 1].x >> 8)) { ix->keys.push_back(all[i].x >> 8); ix->first.push_back((int64_t)i); }
		ix->where[i] = all[i].y;
	}
	ix->first.push_back((int64_t)all.size());
	if (ix->keys.size() >= ((size_t)1 << 32)) { delete ix; fail("mm2gb_index_build: more than 2^32 distinct minimizers"); return nullptr; }
	ix->build_buckets();
	return reinterpret_cast<mm2gb_index_t*>(ix);
}

void mm2gb_index_destroy(mm2gb_index_t *ix) { delete reinterpret_cast<SeedIndex*>(ix); }

int64_t mm2gb_index_size(const mm2gb_index_t *ix_, int64_t *n_occurrences)
{
	const SeedIndex *ix = reinterpret_cast<const SeedIndex*>(ix_);
	if (!ix) return -1;
	if (n_occurrences) *n_occurrences = (int64_t)ix->where.size();
	return (int64_t)ix->keys.size();
}

// options.c:78-84 with index.c:186-211: one more than the (1 - frac) quantile of the occurrence counts, within [min_mid_occ, max_mid_occ]
int32_t mm2gb_index_mid_occ(const mm2gb_index_t *ix_, float frac, int32_t min_mid_occ, int32_t max_mid_occ)
{
	const SeedIndex *ix = reinterpret_cast<const SeedIndex*>(ix_);
	if (!ix) return -1;
	int32_t occ = INT32_MAX;
	if (frac > 0.f && !ix->keys.empty()) {
		const size_t n = ix->keys.size();
		std::vector<uint32_t> cnt(n);
		for (size_t i = 0; i < n; ++i) cnt[i] = (uint32_t)(ix->first[i + 1] - ix->first[i]);
		const size_t kth = (size_t)(uint32_t)((1. - frac) * n);
		std::nth_element(cnt.begin(), cnt.begin() + (ptrdiff_t)std::min(kth, n - 1), cnt.end());
		occ = (int32_t)(cnt[std::min(kth, n - 1)] + 1);
	}
	if (occ < min_mid_occ) occ = min_mid_occ;
	if (max_mid_occ > min_mid_occ && occ > max_mid_occ) occ = max_mid_occ;
	return occ;
}

int mm2gb_collect_matches(const mm2gb_index_t *ix, const char *seq, int32_t len, const mm2gb_seed_opt_t *opt, mm2gb_matches_t *out)
{
	return collect_matches_refs(ix, seq, len, opt, out, nullptr);
}

} // extern "C"

// refs == nullptr: out->hits is a copy of every kept seed's occurrences; otherwise out->hits stays null and refs[s] points at seed s's
// occurrences in the index (the mapper copies them straight into the batch's array: host_chain.h)
int mm2gb::collect_matches_refs(const mm2gb_index_t *ix_, const char *seq, int32_t len, const mm2gb_seed_opt_t *opt, mm2gb_matches_t *out, std::vector<const uint64_t*> *refs)
{
	const SeedIndex *ix = reinterpret_cast<const SeedIndex*>(ix_);
	if (!ix || !opt || !out || len < 0 || (len > 0 && !seq)) return fail("mm2gb_collect_matches: null argument");
	memset(out, 0, sizeof(*out));
	std::vector<Mini> mv;
	if (len > 0) sketch(seq, len, ix->w, ix->k, 0, mv);                       // map.c:186-199, one segment
	// seed.c:5-30: a minimizer that makes up more than q_occ_frac of the read's minimizers (and more than mid_occ of them) goes
	if (opt->q_occ_frac > 0.0f && opt->mid_occ > 0 && (int64_t)mv.size() > opt->mid_occ) {
		// occurrences of every minimizer value in the read, counted in an open-addressed table (the reference sorts a copy, seed.c:12-16; what
		// is dropped depends on the counts only)
		size_t cap = 64;
		while (cap < 2 * mv.size()) cap <<= 1;
		std::vector<uint64_t> key(cap, ~0ull);
		std::vector<int32_t> cnt(cap, 0);
		auto slot_of = [&](uint64_t x) { size_t h = (size_t)((x * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1); while (key[h] != ~0ull && key[h] != x) h = (h + 1) & (cap - 1); return h; };
		bool any = false;
		for (const Mini &q : mv) { const size_t h = slot_of(q.x); key[h] = q.x; any |= ++cnt[h] > opt->mid_occ; }
		if (any) {
			size_t kept = 0;
			for (size_t i = 0; i < mv.size(); ++i) {
				const int32_t c = cnt[slot_of(mv[i].x)];
				if (!(c > opt->mid_occ && c > mv.size() * opt->q_occ_frac)) mv[kept++] = mv[i];
			}
			mv.resize(kept);
		}
	}
	// seed.c:32-54
	std::vector<Match> m;
	m.reserve(mv.size());
	for (size_t i = 0; i < mv.size(); ++i) {
		int n = 0;
		const uint64_t *cr = ix->find(mv[i].x >> 8, &n);
		if (n == 0) continue;
		Match q;
		q.n = (uint32_t)n; q.q_pos = (uint32_t)mv[i].y; q.q_span = (uint32_t)(mv[i].x & 0xff); q.seg_id = (uint32_t)(mv[i].y >> 32); q.cr = cr;
		q.flt = false;
		q.tandem = (i > 0 && (mv[i].x >> 8) == (mv[i - 1].x >> 8)) || (i + 1 < mv.size() && (mv[i].x >> 8) == (mv[i + 1].x >> 8));
		m.push_back(q);
	}
	// seed.c:105-111
	if (opt->occ_dist > 0 && opt->max_max_occ > opt->mid_occ) thin_out(m, len, opt->mid_occ, opt->max_max_occ, opt->occ_dist);
	else for (Match &q : m) if (q.n > (uint32_t)opt->mid_occ) q.flt = true;
	// seed.c:112-130: repeat length = bases covered by dropped minimizers; the kept ones, their hits and positions
	int64_t n_hits = 0;
	size_t n_keep = 0;
	for (const Match &q : m) if (!q.flt) { n_hits += q.n; ++n_keep; }
	out->seeds = (mm2gb_seed_t*)malloc((n_keep + 1) * sizeof(mm2gb_seed_t));
	out->hits = refs ? nullptr : (uint64_t*)malloc(((size_t)n_hits + 1) * 8);
	out->mini_pos = (uint64_t*)malloc((n_keep + 1) * 8);
	if (refs) { refs->clear(); refs->reserve(n_keep); }
	if (!out->seeds || (!refs && !out->hits) || !out->mini_pos) { mm2gb_matches_free(out); return fail("mm2gb_collect_matches: out of memory"); }
	int rep_st = 0, rep_en = 0, rep_len = 0;
	int64_t at = 0;
	for (const Match &q : m) {
		if (q.flt) {
			const int en = (int)(q.q_pos >> 1) + 1, st = en - (int)q.q_span;
			if (st > rep_en) { rep_len += rep_en - rep_st; rep_st = st; rep_en = en; }
			else rep_en = en;
		} else {
			mm2gb_seed_t &s = out->seeds[out->n_seeds];
			s.n = q.n; s.q_pos = q.q_pos; s.span_flt = q.q_span; s.seg_tandem = q.seg_id | (q.tandem ? 1u << 31 : 0u);
			if (refs) refs->push_back(q.cr);
			else memcpy(out->hits + at, q.cr, (size_t)q.n * 8);
			at += q.n;
			out->mini_pos[out->n_seeds++] = (uint64_t)q.q_span << 32 | q.q_pos >> 1;
		}
	}
	out->rep_len = rep_len + (rep_en - rep_st);
	out->n_hits = n_hits;
	out->n_mini_pos = out->n_seeds;
	return 0;
}

extern "C" {

// collect_seed_hits (map.c:295-331) with skip_seed (map.c:205-227) on host threads: the same function as mm2gb_collect_seeds_gpu, for
// batches small enough that the trip over the link and one wave sorting the largest read cost more than they save
int mm2gb_collect_seeds_host(int64_t opt_flag, int64_t n_reads, const int64_t *seed_off, const mm2gb_seed_t *seeds, const int64_t *hit_off,
                             const uint64_t *hits, const int32_t *qlen, const int32_t *q_rank, int32_t n_ref, const int32_t *ref_len,
                             const int32_t *ref_rank, int n_threads, int64_t *anchor_off, mm2gb_anchor_t *anchors)
{
	constexpr int64_t F_NO_DIAG = 0x001, F_NO_DUAL = 0x002, F_FOR_ONLY = 0x100000, F_REV_ONLY = 0x200000, F_QSTRAND = 0x100000000LL;   // minimap.h:8-9,28-29,40
	if (n_reads < 0 || !seed_off || !anchor_off || seed_off[0] != 0) return fail("mm2gb_collect_seeds_host: seed_off[0] must be 0");
	anchor_off[0] = 0;
	if (n_reads == 0) return 0;
	const int64_t n_seeds = seed_off[n_reads];
	if (!qlen || !hit_off || (n_seeds > 0 && !seeds) || hit_off[0] != 0) return fail("mm2gb_collect_seeds_host: null argument, or hit_off[0] is not 0");
	const bool names = (opt_flag & (F_NO_DIAG | F_NO_DUAL)) != 0 && q_rank != nullptr;
	if (names && (!ref_rank || n_ref <= 0)) return fail("mm2gb_collect_seeds_host: NO_DIAG / NO_DUAL need ref_rank");
	if (((opt_flag & F_QSTRAND) || (names && (opt_flag & F_NO_DIAG))) && (!ref_len || n_ref <= 0)) return fail("mm2gb_collect_seeds_host: QSTRAND / NO_DIAG need ref_len");
	if (hit_off[n_seeds] > 0 && (!hits || !anchors)) return fail("mm2gb_collect_seeds_host: null buffer");
	// every read writes at its hits' offset first (an upper bound of where it ends up), sorts there, and is moved down afterwards
	std::vector<int64_t> kept((size_t)n_reads, 0);
	std::atomic<int64_t> next(0);
	auto work = [&]() {
		for (;;) {
			const int64_t r = next.fetch_add(1);
			if (r >= n_reads) break;
			mm2gb_anchor_t *out = anchors + hit_off[seed_off[r]];
			int64_t n_a = 0;
			for (int64_t k = seed_off[r]; k < seed_off[r + 1]; ++k) {
				const mm2gb_seed_t &q = seeds[k];
				const uint32_t q_span = q.span_flt & 0x7fffffffu, seg_id = q.seg_tandem & 0x7fffffffu;
				for (int64_t h = hit_off[k]; h < hit_off[k + 1]; ++h) {
					const uint64_t rr = hits[h];
					const int32_t rpos = (int32_t)((uint32_t)rr >> 1);
					const bool same_strand = (rr & 1) == (q.q_pos & 1);
					bool skip = false, is_self = false;
					if (names) {                                                                      // map.c:208-219
						const int32_t rid = (int32_t)(rr >> 32);
						if ((opt_flag & F_NO_DIAG) && q_rank[r] == ref_rank[rid] && ref_len[rid] == qlen[r]) {
							if ((uint32_t)rr >> 1 == (q.q_pos >> 1)) skip = true;
							else if (same_strand) is_self = true;
						}
						if (!skip && (opt_flag & F_NO_DUAL) && q_rank[r] > ref_rank[rid]) skip = true;
					}
					if (!skip && (opt_flag & (F_FOR_ONLY | F_REV_ONLY))) skip = same_strand ? (opt_flag & F_REV_ONLY) != 0 : (opt_flag & F_FOR_ONLY) != 0;   // map.c:220-226
					if (skip) continue;
					mm2gb_anchor_t &p = out[n_a++];
					if (same_strand) {                                                                // map.c:311-313
						p.x = (rr & 0xffffffff00000000ULL) | (uint32_t)rpos;
						p.y = (uint64_t)q_span << 32 | q.q_pos >> 1;
					} else if (!(opt_flag & F_QSTRAND)) {                                             // map.c:314-316
						p.x = 1ULL << 63 | (rr & 0xffffffff00000000ULL) | (uint32_t)rpos;
						p.y = (uint64_t)q_span << 32 | (uint32_t)(qlen[r] - (int32_t)((q.q_pos >> 1) + 1 - q_span) - 1);
					} else {                                                                          // map.c:317-321
						p.x = 1ULL << 63 | (rr & 0xffffffff00000000ULL) | (uint32_t)(ref_len[rr >> 32] - (rpos + 1 - (int32_t)q_span) - 1);
						p.y = (uint64_t)q_span << 32 | q.q_pos >> 1;
					}
					p.y |= (uint64_t)seg_id << 48;                                                     // MM_SEED_SEG_SHIFT
					if (q.seg_tandem >> 31) p.y |= 1ULL << 42;                                         // MM_SEED_TANDEM
					if (is_self) p.y |= 1ULL << 43;                                                    // MM_SEED_SELF
				}
			}
			sort_by_x_like_host(out, out + n_a);                                                       // map.c:329
			kept[(size_t)r] = n_a;
		}
	};
	const int nt = std::max(1, n_threads);
	if (nt == 1) work();
	else { std::vector<std::thread> pool; for (int t = 0; t < nt; ++t) pool.emplace_back(work); for (auto &th : pool) th.join(); }
	for (int64_t r = 0; r < n_reads; ++r) {
		anchor_off[r + 1] = anchor_off[r] + kept[(size_t)r];
		const int64_t from = hit_off[seed_off[r]];
		if (from != anchor_off[r] && kept[(size_t)r] > 0) memmove(anchors + anchor_off[r], anchors + from, (size_t)kept[(size_t)r] * sizeof(mm2gb_anchor_t));
	}
	return 0;
}

void mm2gb_matches_free(mm2gb_matches_t *m)
{
	if (!m) return;
	free(m->seeds); free(m->hits); free(m->mini_pos);
	memset(m, 0, sizeof(*m));
}

} // extern "C"
