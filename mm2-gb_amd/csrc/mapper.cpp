// mapper.cpp -- reads in, PAF out, without the reference's sources: the host side around the device path, written
// from scratch to give the reference's output for single-segment reads mapped without base-level alignment (no -a / -c).
//
//   per batch of reads:   matches (seeding.cpp, host threads)  ->  anchors, sorted (collect_seed_hits: device for large batches, host threads otherwise)  ->  chains (device:
//   chaining DP + backtrack)  ->  re-chaining of reads whose chains look broken (host threads: mg_lchain_rmq with the reference's tree, csrc/rmq_host.cpp, map.c:697-708)
//   ->  hit records (device: mm_gen_regs)  ->  per read on the host: primary / secondary (mm_set_parent, hit.c:125-198), which
//   secondaries stay (mm_select_sub, hit.c:272-295, mm_sync_regs hit.c:247-270), divergence estimate (mm_est_err, esterr.c:31-64),
//   mm_filter_strand_retained (hit.c:297-309), mapping quality (mm_set_mapq, hit.c:420-466), PAF line (format.c:274-321).
//
// Not reproduced: more than one query segment, base-level alignment and everything that depends on it (inversions, cs/MD, SAM),
// ALT contigs, the heap variant of seed collection, homopolymer-compressed indexes, --qstrand, multi-part indexes.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "engine.h"
#include "host_chain.h"
#include "trace.h"

namespace mm2gb {
namespace {

struct Hit {                                   // mm_reg1_t without the alignment (minimap.h:104-119)
	int id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, mlen, blen, n_sub, score0;
	int mapq;
	bool rev, strand_retained;
	uint32_t hash;
	float div;
};

enum { PARENT_UNSET = -1, PARENT_TMP_PRI = -2 };   // mmpriv.h:13-14

// khash.h:383-409
uint32_t name_hash(const char *s) { uint32_t h = (uint32_t)(unsigned char)*s; if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)(unsigned char)*s; return h; }
uint32_t wang(uint32_t k) { k += ~(k << 15); k ^= k >> 10; k += k << 3; k ^= k >> 6; k += ~(k << 11); k ^= k >> 16; return k; }

// hit.c:125-198 without alignments and ALT: hits come best first; a hit is secondary to the first earlier primary it overlaps by
// more than mask_level of the shorter of the two (less what earlier primaries leave uncovered of it)
void set_parent(float mask_level, int mask_len, std::vector<Hit> &r, bool hard_mask_level)
{
	const int n = (int)r.size();
	if (n == 0) return;
	for (int i = 0; i < n; ++i) r[(size_t)i].id = i;
	std::vector<int> prim{ 0 };
	std::vector<uint64_t> cov;
	r[0].parent = 0;
	for (int i = 1; i < n; ++i) {
		Hit &ri = r[(size_t)i];
		const int si = ri.qs, ei = ri.qe;
		int uncov = 0;
		bool overlaps_any = true;
		if (!hard_mask_level) {
			cov.clear();
			for (int w : prim) {
				int sj = r[(size_t)w].qs, ej = r[(size_t)w].qe;
				if (ej <= si || sj >= ei) continue;
				cov.push_back((uint64_t)std::max(sj, si) << 32 | (uint32_t)std::min(ej, ei));
			}
			overlaps_any = !cov.empty();
			if (overlaps_any) {                          // length of [si, ei) no earlier primary covers
				std::sort(cov.begin(), cov.end());
				int x = si;
				for (uint64_t c : cov) {
					if ((int)(c >> 32) > x) uncov += (int)(c >> 32) - x;
					x = std::max(x, (int)(int32_t)c);
				}
				if (ei > x) uncov += ei - x;
			}
		}
		bool secondary = false;
		if (overlaps_any) {
			for (int w : prim) {
				Hit &rp = r[(size_t)w];
				const int sj = rp.qs, ej = rp.qe;
				if (ej <= si || sj >= ei) continue;
				const int lmin = std::min(ej - sj, ei - si), lmax = std::max(ej - sj, ei - si);
				const int ol = std::min(ei, ej) - std::max(si, sj);
				if ((float)ol / lmin - (float)uncov / lmax > mask_level && uncov <= mask_len) {
					ri.parent = rp.parent;
					rp.subsc = std::max(rp.subsc, ri.score);
					if (ri.cnt >= rp.cnt) ++rp.n_sub;
					secondary = true;
					break;
				}
			}
		}
		if (!secondary) { prim.push_back(i); ri.parent = i; ri.n_sub = 0; }
	}
}

// hit.c:247-270 (+ mm_set_sam_pri, which has no effect on PAF)
void sync_hits(std::vector<Hit> &r)
{
	int max_id = -1;
	for (const Hit &h : r) max_id = std::max(max_id, h.id);
	std::vector<int> now((size_t)(max_id + 1), -1);
	for (size_t i = 0; i < r.size(); ++i) if (r[i].id >= 0) now[(size_t)r[i].id] = (int)i;
	for (size_t i = 0; i < r.size(); ++i) {
		Hit &h = r[i];
		h.id = (int)i;
		if (h.parent == PARENT_TMP_PRI) h.parent = (int)i;
		else if (h.parent >= 0 && now[(size_t)h.parent] >= 0) h.parent = now[(size_t)h.parent];
		else h.parent = PARENT_UNSET;
	}
}

// hit.c:272-295
void select_sub(float pri_ratio, int min_diff, int best_n, bool check_strand, int min_strand_sc, std::vector<Hit> &r)
{
	if (!(pri_ratio > 0.0f) || r.empty()) return;
	const size_t n = r.size();
	size_t k = 0;
	int n_2nd = 0;
	for (size_t i = 0; i < n; ++i) {
		const int p = r[i].parent;
		if (p == (int)i) { r[k++] = r[i]; continue; }
		const Hit &rp = r[(size_t)p];                 // parents precede their secondaries and are never dropped: still at index p? see below
		if ((r[i].score >= rp.score * pri_ratio || r[i].score + min_diff >= rp.score) && n_2nd < best_n) {
			if (!(r[i].qs == rp.qs && r[i].qe == rp.qe && r[i].rid == rp.rid && r[i].rs == rp.rs && r[i].re == rp.re)) { r[k++] = r[i]; ++n_2nd; }
		} else if (check_strand && n_2nd < best_n && r[i].score > min_strand_sc && r[i].rev != rp.rev) {
			r[i].strand_retained = true;
			r[k++] = r[i]; ++n_2nd;
		}
	}
	if (k != n) { r.resize(k); sync_hits(r); }
}

// esterr.c:9-64
int forward_qpos(int qlen, const mm2gb_anchor_t &a)
{
	const int x = (int32_t)a.y, span = (int)(a.y >> 32 & 0xff);
	return a.x >> 63 ? qlen - 1 - (x + 1 - span) : x;
}

void estimate_divergence(int qlen, const std::vector<int32_t> &ref_len, std::vector<Hit> &r, const mm2gb_anchor_t *a, int n_mini, const uint64_t *mini_pos)
{
	if (n_mini == 0) return;
	uint64_t sum_k = 0;
	for (int i = 0; i < n_mini; ++i) sum_k += mini_pos[i] >> 32 & 0xff;
	const float avg_k = (float)sum_k / n_mini;
	for (Hit &h : r) {
		h.div = -1.0f;
		if (h.cnt == 0) continue;
		auto anchor = [&](int k) -> const mm2gb_anchor_t & { return h.rev ? a[h.as + h.cnt - 1 - k] : a[h.as + k]; };   // in query order
		const int x0 = forward_qpos(qlen, anchor(0));
		int lo = 0, hi = n_mini - 1, st = -1;
		while (lo <= hi) {
			const int mid = (int)(((uint64_t)lo + (uint64_t)hi) >> 1), y = (int32_t)mini_pos[mid];
			if (y < x0) lo = mid + 1; else if (y > x0) hi = mid - 1; else { st = mid; break; }
		}
		if (st < 0) continue;
		int en = st, n_match = 1;
		for (int k = 1, j = st + 1; j < n_mini && k < h.cnt; ++j)
			if (forward_qpos(qlen, anchor(k)) == (int32_t)mini_pos[j]) { ++k; en = j; ++n_match; }
		int n_tot = en - st + 1;
		if (h.qs > avg_k && h.rs > avg_k) ++n_tot;
		if (qlen - h.qs > avg_k && ref_len[(size_t)h.rid] - h.re > avg_k) ++n_tot;
		h.div = n_match >= n_tot ? 0.0f : (float)(1.0 - pow((double)n_match / n_tot, 1.0 / avg_k));
	}
}

// hit.c:297-309
void filter_strand_retained(std::vector<Hit> &r)
{
	size_t k = 0;
	for (size_t i = 0; i < r.size(); ++i) {
		const int p = r[i].parent;
		if (!r[i].strand_retained || r[i].div < r[(size_t)p].div * 5.0f || r[i].div < 0.01f) r[k++] = r[i];
	}
	r.resize(k);
}

// hit.c:420-466 without alignments
void set_mapq(std::vector<Hit> &r, int min_chain_sc, int rep_len)
{
	if (r.empty()) return;
	int64_t sum_sc = 0;
	for (const Hit &h : r) if (h.parent == h.id) sum_sc += h.score;
	const float uniq_ratio = (float)sum_sc / (sum_sc + rep_len);
	for (Hit &h : r) {
		if (h.parent != h.id) { h.mapq = 0; continue; }
		const float pen_s1 = (h.score > 100 ? 1.0f : 0.01f * h.score) * uniq_ratio;
		float pen_cm = h.cnt > 10 ? 1.0f : 0.1f * h.cnt;
		pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
		const int subsc = h.subsc > min_chain_sc ? h.subsc : min_chain_sc;
		const float x = (float)subsc / h.score0;
		int mapq = (int)(pen_cm * 40.0f * (1.0f - x) * logf((float)h.score));
		mapq -= (int)(4.343f * logf((float)(h.n_sub + 1)) + .499f);
		mapq = mapq > 0 ? mapq : 0;
		h.mapq = mapq < 60 ? mapq : 60;
	}
}

void append_int(std::string &s, long long v) { char buf[24]; snprintf(buf, sizeof buf, "%lld", v); s += buf; }

// format.c:274-321
void write_paf(std::string &out, const char *qname, int qlen, const Hit &h, const char *rname, int rlen, int rep_len)
{
	out += qname; out += '\t'; append_int(out, qlen); out += '\t'; append_int(out, h.qs); out += '\t'; append_int(out, h.qe); out += '\t';
	out += h.rev ? '-' : '+'; out += '\t'; out += rname; out += '\t'; append_int(out, rlen); out += '\t'; append_int(out, h.rs); out += '\t';
	append_int(out, h.re); out += '\t'; append_int(out, h.mlen); out += '\t'; append_int(out, h.blen); out += '\t'; append_int(out, h.mapq);
	out += "\ttp:A:"; out += h.id == h.parent ? 'P' : 'S';
	out += "\tcm:i:"; append_int(out, h.cnt);
	out += "\ts1:i:"; append_int(out, h.score);
	if (h.parent == h.id) { out += "\ts2:i:"; append_int(out, h.subsc); }
	if (h.div >= 0.0f && h.div <= 1.0f) {
		out += "\tdv:f:";
		if (h.div == 0.0f) out += '0';
		else { char buf[16]; snprintf(buf, sizeof buf, "%.4f", h.div); out += buf; }
	}
	out += "\trl:i:"; append_int(out, rep_len);
	out += '\n';
}

} // namespace
} // namespace mm2gb

using namespace mm2gb;

namespace mm2gb {
HostScratch &host_scratch(mm2gb_engine_t *eng)
{
	if (!eng->host_scratch) {
		eng->host_scratch = new HostScratch;
		eng->host_scratch_free = [](void *p) { delete static_cast<HostScratch*>(p); };
	}
	return *static_cast<HostScratch*>(eng->host_scratch);
}
}

extern "C" {

void mm2gb_map_opt_init(mm2gb_map_opt_t *o)       // mm_mapopt_init (options.c:15-75), the fields this path looks at
{
	memset(o, 0, sizeof(*o));
	o->seed = 11; o->mid_occ_frac = 2e-4f; o->min_mid_occ = 10; o->max_mid_occ = 1000000; o->q_occ_frac = 0.01f;
	o->min_cnt = 3; o->min_chain_score = 40; o->bw = 500; o->bw_long = 20000; o->max_gap = 5000; o->max_gap_ref = -1;
	o->max_chain_iter = 5000; o->rmq_inner_dist = 1000; o->rmq_size_cap = 100000; o->rmq_rescue_size = 1000; o->rmq_rescue_ratio = 0.1f;
	o->chain_gap_scale = 0.8f; o->chain_skip_scale = 0.0f; o->max_max_occ = 4095; o->occ_dist = 500;
	o->mask_level = 0.5f; o->mask_len = INT32_MAX; o->pri_ratio = 0.8f; o->best_n = 5;
	o->host_threads = 0;           // 0: as many as the process may use, at most 32
}

int mm2gb_engine_release_host_scratch(mm2gb_engine_t *eng)
{
	if (!eng) return fail("mm2gb: null engine");
	if (eng->host_scratch) static_cast<HostScratch*>(eng->host_scratch)->release();
	return 0;
}

static int map_reads_body(mm2gb_engine_t *eng, const mm2gb_index_t *ix, int k, const char *const *ref_names, const int32_t *ref_lens, int32_t n_ref,
                          const mm2gb_map_opt_t *opt_in, int32_t n_reads, const char *const *names, const char *const *seqs, const int32_t *lens,
                          char **paf_out, int64_t *paf_len, mm2gb_map_stats_t *stats);

int mm2gb_map_reads(mm2gb_engine_t *eng, const mm2gb_index_t *ix, int k, const char *const *ref_names, const int32_t *ref_lens, int32_t n_ref,
                    const mm2gb_map_opt_t *opt_in, int32_t n_reads, const char *const *names, const char *const *seqs, const int32_t *lens,
                    char **paf_out, int64_t *paf_len, mm2gb_map_stats_t *stats)
{
	// a batch's arrays are gigabytes: running out of host memory on this thread is an error of the call, not the end of the process
	try { return map_reads_body(eng, ix, k, ref_names, ref_lens, n_ref, opt_in, n_reads, names, seqs, lens, paf_out, paf_len, stats); }
	catch (const std::bad_alloc&) { return fail("mm2gb_map_reads: out of host memory"); }
}

static int map_reads_body(mm2gb_engine_t *eng, const mm2gb_index_t *ix, int k, const char *const *ref_names, const int32_t *ref_lens, int32_t n_ref,
                          const mm2gb_map_opt_t *opt_in, int32_t n_reads, const char *const *names, const char *const *seqs, const int32_t *lens,
                          char **paf_out, int64_t *paf_len, mm2gb_map_stats_t *stats)
{
	if (!eng || !ix || !opt_in || !paf_out || !paf_len || n_reads < 0 || n_ref <= 0 || !ref_names || !ref_lens || (n_reads > 0 && (!names || !seqs || !lens)))
		return fail("mm2gb_map_reads: null argument");
	mm2gb_map_opt_t opt = *opt_in;
	if (opt.flag & ~(int64_t)(0x100000 | 0x200000)) return fail("mm2gb_map_reads: of mm_mapopt_t::flag only MM_F_FOR_ONLY and MM_F_REV_ONLY are supported");
	if (opt.mid_occ <= 0) opt.mid_occ = mm2gb_index_mid_occ(ix, opt.mid_occ_frac, opt.min_mid_occ, opt.max_mid_occ);   // options.c:78-84
	if (opt.bw_long < opt.bw) opt.bw_long = opt.bw;
	if (opt.host_threads <= 0) opt.host_threads = std::min(32, usable_cpus());
	mm2gb_map_stats_t st_local; memset(&st_local, 0, sizeof st_local);
	auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double t_mark = now();
	auto lap = [&](double &slot) { const double t = now(); slot += t - t_mark; t_mark = t; };
	*paf_out = nullptr; *paf_len = 0;
	if (n_reads == 0) { *paf_out = (char*)calloc(1, 1); if (stats) *stats = st_local; return *paf_out ? 0 : fail("mm2gb_map_reads: out of memory"); }
	const size_t R = (size_t)n_reads;
	std::vector<int32_t> ref_len_v(ref_lens, ref_lens + n_ref);

	// 1. matches on host threads
	std::unique_ptr<TraceRange> tr(new TraceRange("mm2gb:map_seed"));   // stage ranges for rocprofv3 --marker-trace / rocprof-sys
	struct MatchSet { std::vector<mm2gb_matches_t> v; ~MatchSet() { for (auto &m : v) mm2gb_matches_free(&m); } } mt_own;   // (freed on every way out)
	std::vector<mm2gb_matches_t> &mt = mt_own.v;
	mt.resize(R);
	std::vector<std::vector<const uint64_t*>> occ(R);     // per read and kept seed: its occurrences, where the index holds them (copied once, into the batch's array)
	for (auto &m : mt) memset(&m, 0, sizeof m);
	const mm2gb_seed_opt_t so = { opt.mid_occ, opt.max_max_occ, opt.occ_dist, opt.q_occ_frac };
	{
		std::atomic<int32_t> next(0);
		std::atomic<int> bad(0);
		std::string why;                                      // error text is per thread: carry the first one over
		std::mutex why_lock;
		auto work = [&]() {
			for (;;) {
				const int32_t r = next.fetch_add(1);
				if (r >= n_reads) break;
				if (lens[r] > 0 && collect_matches_refs(ix, seqs[r], lens[r], &so, &mt[(size_t)r], &occ[(size_t)r])) {
					std::lock_guard<std::mutex> g(why_lock);
					if (!bad.exchange(1)) why = mm2gb_last_error();
				}
			}
		};
		const int nt = std::max(1, opt.host_threads);
		std::vector<std::thread> pool;
		for (int t = 0; t < nt; ++t) pool.emplace_back(work);
		for (auto &th : pool) th.join();
		if (bad) { for (auto &m : mt) mm2gb_matches_free(&m); return fail(why); }
	}
	auto free_matches = [&]() { for (auto &m : mt) mm2gb_matches_free(&m); };
	lap(st_local.s_seed);
	tr.reset(); tr.reset(new TraceRange("mm2gb:map_anchors"));

	// 2. anchors, sorted, on the device
	std::vector<int64_t> seed_off(R + 1, 0);
	for (size_t r = 0; r < R; ++r) seed_off[r + 1] = seed_off[r] + mt[r].n_seeds;
	std::vector<mm2gb_seed_t> seeds((size_t)seed_off[R]);
	std::vector<int64_t> hit_off((size_t)seed_off[R] + 1, 0);
	int64_t n_hits = 0;
	for (size_t r = 0; r < R; ++r) {
		if (mt[r].n_seeds) memcpy(seeds.data() + seed_off[r], mt[r].seeds, (size_t)mt[r].n_seeds * sizeof(mm2gb_seed_t));
		for (int s = 0; s < mt[r].n_seeds; ++s) { hit_off[(size_t)(seed_off[r] + s) + 1] = hit_off[(size_t)(seed_off[r] + s)] + mt[r].seeds[s].n; }
		n_hits += mt[r].n_hits;
	}
	// (the batch's two largest arrays are the engine's from call to call: a gigabyte of fresh pages costs more to touch than to fill)
	BigBuf<uint64_t> &hits = host_scratch(eng).hits;
	BigBuf<mm2gb_anchor_t> &anchors = host_scratch(eng).anchors;
	hits.resize((size_t)n_hits);
	{
		uint64_t *const hits_ptr = hits.data();
		std::atomic<size_t> next(0);
		auto work = [&]() {
			for (;;) {
				const size_t lo = next.fetch_add(16);
				if (lo >= R) break;
				for (size_t r = lo; r < std::min(R, lo + 16); ++r)
					for (int32_t q = 0; q < mt[r].n_seeds; ++q)
						memcpy(hits_ptr + hit_off[(size_t)seed_off[r] + (size_t)q], occ[r][(size_t)q], (size_t)mt[r].seeds[q].n * 8);
			}
		};
		std::vector<std::thread> pool;
		for (int t = 1; t < std::max(1, opt.host_threads); ++t) pool.emplace_back(work);
		work();
		for (auto &th : pool) th.join();
	}
	std::vector<int32_t> qlen(lens, lens + n_reads);
	std::vector<int64_t> a_off(R + 1, 0);
	anchors.resize((size_t)std::max<int64_t>(n_hits, 1));
	// on the device for large batches (mm2gb_collect_seeds_gpu: matches up, anchors down, one wave sorts a read); below that the host
	// threads are quicker: the largest read's sort alone is hundreds of milliseconds for one wave, milliseconds for a core
	const bool seeds_on_device = opt.seeds_on_device > 0 || (opt.seeds_on_device == 0 && n_hits >= 400000000);
	if (seeds_on_device ? mm2gb_collect_seeds_gpu(eng, opt.flag, n_reads, seed_off.data(), seeds.data(), hit_off.data(), hits.data(), qlen.data(), nullptr, n_ref, nullptr, nullptr,
	                                              a_off.data(), anchors.data())
	                    : mm2gb_collect_seeds_host(opt.flag, n_reads, seed_off.data(), seeds.data(), hit_off.data(), hits.data(), qlen.data(), nullptr, n_ref, nullptr, nullptr,
	                                               std::max(1, opt.host_threads), a_off.data(), anchors.data())) { free_matches(); return -1; }
	st_local.n_anchors = a_off[R];
	lap(st_local.s_anchors);
	tr.reset(); tr.reset(new TraceRange("mm2gb:map_chain"));

	// 3. chains on the device; map.c:393-426 for the parameters (the GPU path chains with max-chain-skip = infinity)
	mm2gb_misc_t misc;
	misc.max_iter = opt.max_chain_iter; misc.max_dist_y = opt.max_gap; misc.max_dist_x = opt.max_gap_ref > 0 ? opt.max_gap_ref : opt.max_gap;
	misc.max_skip = INT32_MAX; misc.bw = opt.bw; misc.min_cnt = opt.min_cnt; misc.min_score = opt.min_chain_score; misc.is_cdna = 0; misc.n_seg = 1;
	misc.chn_pen_gap = (float)(opt.chain_gap_scale * 0.01 * k); misc.chn_pen_skip = (float)(opt.chain_skip_scale * 0.01 * k);
	if (mm2gb_engine_set_misc(eng, &misc)) { free_matches(); return -1; }
	ChainsOwner ch_own;                                  // (freed on every way out)
	mm2gb_chains_t &ch = ch_own.c;
	// backtrack + compaction as kernels for large batches; below that on host threads, overlapped with the device: a single huge read (a
	// tandem array) keeps one wave busy for hundreds of milliseconds where a core needs tens
	if (a_off[R] >= 200000000 ? mm2gb_chain_gpu(eng, n_reads, a_off.data(), anchors.data(), &ch, nullptr)
	                          : mm2gb_chain_host(eng, n_reads, a_off.data(), anchors.data(), std::max(1, opt.host_threads), &ch, nullptr)) { free_matches(); return -1; }

	// 4. re-chaining of long reads whose best chain leaves much of the read uncovered (map.c:697-708): the chained anchors, sorted
	//    again, through mg_lchain_rmq's fill
	std::vector<int64_t> u_off(ch.u_off, ch.u_off + R + 1), c_off(ch.a_off, ch.a_off + R + 1);
	// (the chains are read where the chaining call left them -- a gigabyte of kept anchors per batch is not copied again)
	const uint64_t *u = ch.u;
	const mm2gb_anchor_t *ca = ch.a;
	lap(st_local.s_chain);
	tr.reset(); tr.reset(new TraceRange("mm2gb:map_rechain"));
	std::vector<int32_t> redo;
	const auto t_rechain = std::chrono::steady_clock::now();
	if (opt.bw_long > opt.bw) {
		for (size_t r = 0; r < R; ++r) {
			if (u_off[r + 1] - u_off[r] <= 1) continue;
			const mm2gb_anchor_t *a = ca + c_off[r];
			const int st = (int32_t)a[0].y, en = (int32_t)a[(int32_t)u[(size_t)u_off[r]] - 1].y;
			if (lens[r] - (en - st) > opt.rmq_rescue_size || en - st > lens[r] * opt.rmq_rescue_ratio) redo.push_back((int32_t)r);
		}
	}
	st_local.n_rechained = (int64_t)redo.size();
	// largest first: a read is one wave's (or one thread's) work from start to end, and the call ends with its longest read
	std::sort(redo.begin(), redo.end(), [&](int32_t u, int32_t v) { const int64_t nu = c_off[(size_t)u + 1] - c_off[(size_t)u], nv = c_off[(size_t)v + 1] - c_off[(size_t)v]; return nu != nv ? nu > nv : u < v; });
	if (!redo.empty()) {
		std::vector<int64_t> ro(redo.size() + 1, 0);
		for (size_t q = 0; q < redo.size(); ++q) ro[q + 1] = ro[q] + (c_off[(size_t)redo[q] + 1] - c_off[(size_t)redo[q]]);
		// (the engine's, kept between calls like the gathers of mm2gb_rmq_chain: fresh pages cost more to touch than to fill)
		BigBuf<mm2gb_anchor_t> &ra = host_scratch(eng).ra;
		ra.resize((size_t)ro.back());
		mm2gb_anchor_t *const ra_ptr = ra.data();               // (for the threads below: `ra` names each thread's own)
		{
			std::atomic<int64_t> next(0);
			auto work = [&]() {
				for (;;) {
					const int64_t q = next.fetch_add(1);
					if (q >= (int64_t)redo.size()) break;
					const size_t r = (size_t)redo[(size_t)q];
					memcpy(ra_ptr + ro[(size_t)q], ca + c_off[r], (size_t)(ro[(size_t)q + 1] - ro[(size_t)q]) * sizeof(mm2gb_anchor_t));
					sort_by_x_like_host(ra_ptr + ro[(size_t)q], ra_ptr + ro[(size_t)q + 1]);
				}
			};
			const int nt = std::max(1, std::min<int>(opt.host_threads, (int)redo.size()));
			std::vector<std::thread> pool;
			for (int t = 1; t < nt; ++t) pool.emplace_back(work);
			work();
			for (auto &th : pool) th.join();
		}
		const bool verbose = getenv("MM2GB_DEBUG_PHASES") != nullptr;
		if (const char *path = getenv("MM2GB_DUMP_RECHAIN")) {       // the re-chaining call's input, for profiling csrc/rmq_host.cpp off the box: offsets, then anchors
			if (FILE *fp = fopen(path, "wb")) {
				const int64_t nr = (int64_t)redo.size();
				fwrite(&nr, 8, 1, fp); fwrite(ro.data(), 8, ro.size(), fp); fwrite(ra.data(), sizeof(mm2gb_anchor_t), ra.size(), fp);
				fclose(fp);
			}
		}
		const auto t_sorted = std::chrono::steady_clock::now();
		const mm2gb_rmq_param_t rp = { opt.max_gap, opt.rmq_inner_dist, opt.bw_long, INT32_MAX, opt.rmq_size_cap, opt.min_cnt, opt.min_chain_score, misc.chn_pen_gap, misc.chn_pen_skip };
		ChainsOwner rc_own, rc_tie_own;
		mm2gb_chains_t &rc = rc_own.c, &rc_tie = rc_tie_own.c;
		std::vector<int32_t> tied(redo.size(), 0);
		std::vector<int> tie_slot(redo.size(), -1);          // reads the device reported a tie for: their place in the host call that follows
		RmqParts parts;                                      // the default path: the results of the call's three sides, spliced from where they are
		std::vector<unsigned char> q_side(redo.size(), 0);   // per re-chained read: which result holds it (0: rc, 1: rc_tie, 2 + k: parts.chains[k]) ...
		std::vector<int64_t> q_slot(redo.size(), 0);         // ... and where
		// mg_lchain_rmq's fill.  Default: mm2gb_rmq_chain (csrc/rmq_hybrid.cpp) -- the kernel form takes the bulk of the reads, the host
		// threads, at the same time, the few whose windows are so dense that one wave would still be on them long after the rest of the
		// batch is done, and afterwards the reads the kernel reported a tie for (where the reference's answer follows from the shape of
		// its tree; the host form keeps that tree's rules).  rechain_on_device = 1: every read on the device first; -1: host threads only.
		if (opt.rechain_on_device == 0) {
			std::vector<int32_t> where(redo.size(), 0);
			mm2gb_rmq_deal_t deal = {};
			if (rmq_chain_parts(eng, &rp, (int64_t)redo.size(), ro.data(), ra.data(), std::max(1, opt.host_threads), parts, where.data(), &deal)) { free_matches(); return -1; }
			for (size_t q = 0; q < redo.size(); ++q) { tied[q] = where[q] == 2; q_side[q] = (unsigned char)(2 + parts.which[q]); q_slot[q] = parts.slot[q]; }
			if (verbose) fprintf(stderr, "[mm2gb] re-chaining deal: %lld reads on the device, %d of them a whole workgroup's (%.3f s, estimated %.3f), %lld on host threads by cost (%.3f s, estimated %.3f), %lld redone after a tie (%.3f s)\n",
			                     (long long)deal.n_device, (int)deal.n_team, deal.device_s, deal.est_device_s, (long long)deal.n_host_cost, deal.host_s, deal.est_host_s, (long long)deal.n_host_tie, deal.tie_s);
		} else if (opt.rechain_on_device > 0) {
			if (mm2gb_rmq_chain_gpu(eng, &rp, (int64_t)redo.size(), ro.data(), ra.data(), &rc, tied.data(), nullptr)) { free_matches(); return -1; }
			std::vector<int64_t> to(1, 0);
			std::vector<mm2gb_anchor_t> ta;
			for (size_t q = 0; q < redo.size(); ++q)
				if (tied[q]) {
					tie_slot[q] = (int)to.size() - 1;
					ta.insert(ta.end(), ra.begin() + ro[q], ra.begin() + ro[q + 1]);
					to.push_back((int64_t)ta.size());
				}
			if (to.size() > 1 && mm2gb_rmq_chain_host(&rp, (int64_t)to.size() - 1, to.data(), ta.data(), std::max(1, opt.host_threads), &rc_tie, nullptr)) { mm2gb_chains_free(&rc); free_matches(); return -1; }
		} else if (mm2gb_rmq_chain_host(&rp, (int64_t)redo.size(), ro.data(), ra.data(), std::max(1, opt.host_threads), &rc, tied.data())) { free_matches(); return -1; }
		if (opt.rechain_on_device != 0)
			for (size_t q = 0; q < redo.size(); ++q) { q_side[q] = tie_slot[q] >= 0 ? 1 : 0; q_slot[q] = tie_slot[q] >= 0 ? tie_slot[q] : (int64_t)q; }
		const mm2gb_chains_t *const side[5] = { &rc, &rc_tie, &parts.chains[0], &parts.chains[1], &parts.chains[2] };
		const auto t_filled = std::chrono::steady_clock::now();
		// splice the re-chained reads back in
		std::vector<int64_t> nu_off(R + 1, 0), nc_off(R + 1, 0);
		std::vector<int> which(R, -1);
		for (size_t q = 0; q < redo.size(); ++q) { which[(size_t)redo[q]] = (int)q; if (tied[q]) ++st_local.n_rmq_tied; }
		for (size_t r = 0; r < R; ++r) {
			const int q = which[r];
			const mm2gb_chains_t &from = *side[q >= 0 ? q_side[(size_t)q] : 0];
			const int64_t qq = q >= 0 ? q_slot[(size_t)q] : 0;
			nu_off[r + 1] = nu_off[r] + (q < 0 ? u_off[r + 1] - u_off[r] : from.u_off[qq + 1] - from.u_off[qq]);
			nc_off[r + 1] = nc_off[r] + (q < 0 ? c_off[r + 1] - c_off[r] : from.a_off[qq + 1] - from.a_off[qq]);
		}
		// (the engine's from call to call: touched pages)
		BigBuf<uint64_t> &nu = host_scratch(eng).nu;
		BigBuf<mm2gb_anchor_t> &nc = host_scratch(eng).nc;
		nu.resize((size_t)nu_off[R]); nc.resize((size_t)nc_off[R]);
		uint64_t *const nu_ptr = nu.data();
		mm2gb_anchor_t *const nc_ptr = nc.data();
		{
			// (a batch's kept anchors are a gigabyte: the copies go to all host threads)
			std::atomic<size_t> next(0);
			auto work = [&]() {
				for (;;) {
					const size_t lo = next.fetch_add(32);
					if (lo >= R) break;
					for (size_t r = lo; r < std::min(R, lo + 32); ++r) {
						const int q = which[r];
						const mm2gb_chains_t &from = *side[q >= 0 ? q_side[(size_t)q] : 0];
						const int64_t qq = q >= 0 ? q_slot[(size_t)q] : 0;
						const uint64_t *su = q < 0 ? u + u_off[r] : from.u + from.u_off[qq];
						const mm2gb_anchor_t *sa = q < 0 ? ca + c_off[r] : from.a + from.a_off[qq];
						if (nu_off[r + 1] > nu_off[r]) memcpy(nu_ptr + nu_off[r], su, (size_t)(nu_off[r + 1] - nu_off[r]) * 8);
						if (nc_off[r + 1] > nc_off[r]) memcpy(nc_ptr + nc_off[r], sa, (size_t)(nc_off[r + 1] - nc_off[r]) * sizeof(mm2gb_anchor_t));
					}
				}
			};
			std::vector<std::thread> pool;
			for (int t = 1; t < std::max(1, opt.host_threads); ++t) pool.emplace_back(work);
			work();
			for (auto &th : pool) th.join();
		}
		mm2gb_chains_free(&rc); mm2gb_chains_free(&rc_tie);
		u = nu_ptr; ca = nc_ptr; u_off.swap(nu_off); c_off.swap(nc_off);
		mm2gb_chains_free(&ch);
		if (verbose) fprintf(stderr, "[mm2gb] re-chaining %zu reads (%lld redone on the host after a tie), %lld anchors: sort %.3f s, fill %.3f s, splice %.3f s\n", redo.size(), (long long)st_local.n_rmq_tied, (long long)ro.back(), std::chrono::duration<double>(t_sorted - t_rechain).count(),
		                     std::chrono::duration<double>(t_filled - t_sorted).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t_filled).count());
	}
	st_local.n_chains = u_off[R];
	lap(st_local.s_rechain);
	tr.reset(); tr.reset(new TraceRange("mm2gb:map_hit_records"));

	// 5. hit records on the device (hit.c:52-88); the hash of map.c:660-662
	std::vector<uint32_t> hash(R);
	for (size_t r = 0; r < R; ++r) {
		uint32_t h = names[r] ? name_hash(names[r]) : 0;
		h ^= wang((uint32_t)lens[r]) + wang((uint32_t)opt.seed);
		hash[r] = wang(h);
	}
	std::vector<mm2gb_reg_t> regs((size_t)std::max<int64_t>(u_off[R], 1));
	{
		mm2gb_chains_t view; view.u_off = u_off.data(); view.u = const_cast<uint64_t*>(u); view.a_off = c_off.data(); view.a = const_cast<mm2gb_anchor_t*>(ca);
		if (mm2gb_gen_regs_gpu(eng, n_reads, &view, qlen.data(), hash.data(), 0, regs.data())) { free_matches(); return -1; }
	}

	lap(st_local.s_regs);
	tr.reset(); tr.reset(new TraceRange("mm2gb:map_hits_to_paf"));
	// 6. per read on the host
	std::vector<std::string> lines(R);
	{
		std::atomic<int32_t> next(0);
		auto work = [&]() {
			std::vector<Hit> hs;
			for (;;) {
				const int32_t ri = next.fetch_add(1);
				if (ri >= n_reads) break;
				const size_t r = (size_t)ri;
				hs.clear();
				for (int64_t j = u_off[r]; j < u_off[r + 1]; ++j) {
					const mm2gb_reg_t &g = regs[(size_t)j];
					Hit h;
					h.id = g.id; h.cnt = g.cnt; h.rid = g.rid; h.score = g.score; h.qs = g.qs; h.qe = g.qe; h.rs = g.rs; h.re = g.re; h.parent = g.parent;
					h.subsc = g.subsc; h.as = g.as; h.mlen = g.mlen; h.blen = g.blen; h.n_sub = g.n_sub; h.score0 = g.score0;
					h.mapq = 0; h.rev = (g.flags >> 10) & 1; h.strand_retained = false; h.hash = g.hash; h.div = g.div;
					hs.push_back(h);
				}
				if (hs.empty()) continue;
				set_parent(opt.mask_level, opt.mask_len, hs, false);                                     // map.c:336
				select_sub(opt.pri_ratio, k * 2, opt.best_n, true, (int)(opt.max_gap * 0.8), hs);          // map.c:337
				estimate_divergence(lens[r], ref_len_v, hs, ca + c_off[r], mt[r].n_mini_pos, mt[r].mini_pos);   // map.c:751
				filter_strand_retained(hs);                                                              // map.c:752
				set_mapq(hs, opt.min_chain_score, mt[r].rep_len);                                        // map.c:758
				for (const Hit &h : hs) write_paf(lines[r], names[r] ? names[r] : "*", lens[r], h, ref_names[h.rid], ref_lens[h.rid], mt[r].rep_len);
			}
		};
		const int nt = std::max(1, opt.host_threads);
		std::vector<std::thread> pool;
		for (int t = 0; t < nt; ++t) pool.emplace_back(work);
		for (auto &th : pool) th.join();
	}
	free_matches();
	lap(st_local.s_post);
	tr.reset();
	size_t total = 0;
	for (const auto &l : lines) total += l.size();
	char *buf = (char*)malloc(total + 1);
	if (!buf) return fail("mm2gb_map_reads: out of memory");
	size_t at = 0;
	for (const auto &l : lines) { memcpy(buf + at, l.data(), l.size()); at += l.size(); if (!l.empty()) ++st_local.n_mapped; }
	buf[total] = 0;
	*paf_out = buf; *paf_len = (int64_t)total;
	st_local.n_reads = n_reads;
	if (stats) *stats = st_local;
	return 0;
}

// Several devices (reads shard, no exchange): the batch is cut into contiguous runs of reads balanced by bases, every engine
// maps its run on its own host thread (the host threads of opt are shared out among them), the PAF comes back in read order.
int mm2gb_map_reads_multi(mm2gb_engine_t *const *engines, int n_engines, const mm2gb_index_t *ix, int k, const char *const *ref_names, const int32_t *ref_lens,
                          int32_t n_ref, const mm2gb_map_opt_t *opt_in, int32_t n_reads, const char *const *names, const char *const *seqs, const int32_t *lens,
                          char **paf_out, int64_t *paf_len, mm2gb_map_stats_t *stats)
{
	if (!engines || n_engines < 1 || !opt_in || !paf_out || !paf_len || n_reads < 0 || (n_reads > 0 && !lens)) return fail("mm2gb_map_reads_multi: null argument");
	if (n_engines == 1) return mm2gb_map_reads(engines[0], ix, k, ref_names, ref_lens, n_ref, opt_in, n_reads, names, seqs, lens, paf_out, paf_len, stats);
	*paf_out = nullptr; *paf_len = 0;
	int64_t total = 0;
	for (int32_t r = 0; r < n_reads; ++r) total += lens[r];
	std::vector<int32_t> cut((size_t)n_engines + 1, n_reads);
	cut[0] = 0;
	{ int64_t acc = 0; int e = 1; for (int32_t r = 0; r < n_reads && e < n_engines; ++r) { acc += lens[r]; if (acc * n_engines >= total * e) cut[(size_t)e++] = r + 1; } }
	mm2gb_map_opt_t opt = *opt_in;
	opt.host_threads = std::max(1, (opt_in->host_threads > 0 ? opt_in->host_threads : std::min(32, usable_cpus())) / n_engines);
	if (opt.mid_occ <= 0) opt.mid_occ = mm2gb_index_mid_occ(ix, opt.mid_occ_frac, opt.min_mid_occ, opt.max_mid_occ);     // once, not per engine
	std::vector<char*> part((size_t)n_engines, nullptr);
	std::vector<int64_t> part_len((size_t)n_engines, 0);
	std::vector<mm2gb_map_stats_t> st((size_t)n_engines);
	std::vector<int> rc((size_t)n_engines, 0);
	std::vector<std::string> err((size_t)n_engines);
	std::vector<std::thread> pool;
	for (int e = 0; e < n_engines; ++e)
		pool.emplace_back([&, e]() {
			const int32_t from = cut[(size_t)e], n = cut[(size_t)e + 1] - from;
			rc[(size_t)e] = mm2gb_map_reads(engines[e], ix, k, ref_names, ref_lens, n_ref, &opt, n, names + from, seqs + from, lens + from, &part[(size_t)e], &part_len[(size_t)e], &st[(size_t)e]);
			if (rc[(size_t)e]) err[(size_t)e] = mm2gb_last_error();
		});
	for (auto &th : pool) th.join();
	int bad = -1;
	for (int e = 0; e < n_engines; ++e) if (rc[(size_t)e] && bad < 0) bad = e;
	if (bad >= 0) { for (char *p : part) free(p); return fail("mm2gb_map_reads_multi: engine " + std::to_string(bad) + ": " + err[(size_t)bad]); }
	int64_t all = 0;
	for (int64_t l : part_len) all += l;
	char *buf = (char*)malloc((size_t)all + 1);
	if (!buf) { for (char *p : part) free(p); return fail("mm2gb_map_reads_multi: out of memory"); }
	int64_t at = 0;
	mm2gb_map_stats_t sum; memset(&sum, 0, sizeof sum);
	for (int e = 0; e < n_engines; ++e) {
		memcpy(buf + at, part[(size_t)e], (size_t)part_len[(size_t)e]); at += part_len[(size_t)e]; free(part[(size_t)e]);
		const mm2gb_map_stats_t &q = st[(size_t)e];
		sum.n_reads += q.n_reads; sum.n_mapped += q.n_mapped; sum.n_anchors += q.n_anchors; sum.n_chains += q.n_chains; sum.n_rechained += q.n_rechained; sum.n_rmq_tied += q.n_rmq_tied;
		sum.s_seed = std::max(sum.s_seed, q.s_seed); sum.s_anchors = std::max(sum.s_anchors, q.s_anchors); sum.s_chain = std::max(sum.s_chain, q.s_chain);
		sum.s_rechain = std::max(sum.s_rechain, q.s_rechain); sum.s_regs = std::max(sum.s_regs, q.s_regs); sum.s_post = std::max(sum.s_post, q.s_post);
	}
	buf[all] = 0;
	*paf_out = buf; *paf_len = all;
	if (stats) *stats = sum;
	return 0;
}

// A run of any size as a stream of batches (role of the batch rotation of worker_for, map.c:924-1153: seed batch k+1 while batch k is
// chained and batch k-1 is finished): the reads are cut into consecutive chunks of about chunk_bases bases, and every engine -- several
// per device are the point: each has its own streams and arenas -- has a host thread that takes the next chunk and maps it from
// seeding to PAF.  A chunk's stages alternate between host threads and the device, so with two or three engines on a GPU one chunk is
// being seeded or post-processed while another one's kernels run; with engines on several GPUs the reads shard (no
// exchange).  The host threads of opt are shared out with over-subscription (2.5 x), because a chunk's threads idle while its kernels run.
// PAF in read order; stats: counts summed, s_* = seconds of each stage SUMMED over chunks (they overlap: not wall time).
int mm2gb_map_reads_stream(mm2gb_engine_t *const *engines, int n_engines, const mm2gb_index_t *ix, int k, const char *const *ref_names, const int32_t *ref_lens,
                           int32_t n_ref, const mm2gb_map_opt_t *opt_in, int32_t n_reads, const char *const *names, const char *const *seqs, const int32_t *lens,
                           int64_t chunk_bases, char **paf_out, int64_t *paf_len, mm2gb_map_stats_t *stats)
{
	if (!engines || n_engines < 1 || !ix || !opt_in || !paf_out || !paf_len || n_reads < 0 || (n_reads > 0 && (!lens || !names || !seqs)) || (n_ref > 0 && (!ref_names || !ref_lens)))
		return fail("mm2gb_map_reads_stream: null argument");
	for (int e = 0; e < n_engines; ++e) if (!engines[e]) return fail("mm2gb_map_reads_stream: null engine");
	*paf_out = nullptr; *paf_len = 0;
	if (chunk_bases <= 0) chunk_bases = 96 * 1000 * 1000;
	std::vector<int32_t> cut(1, 0);
	{ int64_t acc = 0; for (int32_t r = 0; r < n_reads; ++r) { acc += lens[r]; if (acc >= chunk_bases && r + 1 < n_reads) { cut.push_back(r + 1); acc = 0; } } }
	cut.push_back(n_reads);
	const size_t n_chunks = cut.size() - 1;
	mm2gb_map_opt_t opt = *opt_in;
	const int all_threads = opt_in->host_threads > 0 ? opt_in->host_threads : std::min(32, usable_cpus());
	const int workers = (int)std::min<size_t>((size_t)n_engines, std::max<size_t>(1, n_chunks));
	// host threads of all workers together, in % of opt's (MM2GB_STREAM_THREADS_PCT): a chunk's threads idle while its kernels run, and the reads its
	// re-chaining gives to host threads want a core each when they come.  1.05 Gbp, four engines, 16 threads: 100 % 8.0 s, 150 % 6.8-7.1, 250 % 6.3-6.5, 400 % 6.6-6.9
	int oversub_pct = 250;
	if (const char *v = getenv("MM2GB_STREAM_THREADS_PCT")) oversub_pct = std::max(25, atoi(v));
	opt.host_threads = std::max(1, workers == 1 ? all_threads : (all_threads * oversub_pct / 100 + workers - 1) / workers);
	if (opt.mid_occ <= 0) opt.mid_occ = mm2gb_index_mid_occ(ix, opt.mid_occ_frac, opt.min_mid_occ, opt.max_mid_occ);     // once, not per chunk
	std::vector<char*> part(n_chunks, nullptr);
	std::vector<int64_t> part_len(n_chunks, 0);
	std::vector<mm2gb_map_stats_t> st(n_chunks);
	std::atomic<size_t> next(0);
	std::atomic<int> failed(0);
	std::string why;
	std::mutex why_lock;
	auto work = [&](int e) {
		for (;;) {
			const size_t c = next.fetch_add(1);
			if (c >= n_chunks || failed.load()) break;
			const int32_t from = cut[c], n = cut[c + 1] - from;
			memset(&st[c], 0, sizeof(st[c]));
			if (mm2gb_map_reads(engines[e], ix, k, ref_names, ref_lens, n_ref, &opt, n, names + from, seqs + from, lens + from, &part[c], &part_len[c], &st[c])) {
				std::lock_guard<std::mutex> g(why_lock);
				if (!failed.exchange(1)) why = "chunk " + std::to_string(c) + " on engine " + std::to_string(e) + ": " + mm2gb_last_error();
			}
		}
	};
	{
		std::vector<std::thread> pool;
		for (int e = 1; e < workers; ++e) pool.emplace_back(work, e);
		work(0);
		for (auto &th : pool) th.join();
	}
	if (failed.load()) { for (char *p : part) free(p); return fail("mm2gb_map_reads_stream: " + why); }
	int64_t all = 0;
	for (int64_t l : part_len) all += l;
	char *buf = (char*)malloc((size_t)all + 1);
	if (!buf) { for (char *p : part) free(p); return fail("mm2gb_map_reads_stream: out of memory"); }
	int64_t at = 0;
	mm2gb_map_stats_t sum; memset(&sum, 0, sizeof sum);
	for (size_t c = 0; c < n_chunks; ++c) {
		if (part_len[c]) memcpy(buf + at, part[c], (size_t)part_len[c]);
		at += part_len[c]; free(part[c]);
		const mm2gb_map_stats_t &q = st[c];
		sum.n_reads += q.n_reads; sum.n_mapped += q.n_mapped; sum.n_anchors += q.n_anchors; sum.n_chains += q.n_chains; sum.n_rechained += q.n_rechained; sum.n_rmq_tied += q.n_rmq_tied;
		sum.s_seed += q.s_seed; sum.s_anchors += q.s_anchors; sum.s_chain += q.s_chain; sum.s_rechain += q.s_rechain; sum.s_regs += q.s_regs; sum.s_post += q.s_post;
	}
	buf[all] = 0;
	*paf_out = buf; *paf_len = all;
	if (stats) *stats = sum;
	return 0;
}

} // extern "C"
