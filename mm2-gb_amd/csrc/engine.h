// engine.h -- internal: the chaining engine (device arenas, streams, micro-batch pipeline).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>
#include "../../include/mm2gb_chain.h"
#include "chain_dev.h"
#include "post_dev.h"

namespace mm2gb {

void set_error(const std::string &msg);
int  fail(const std::string &msg);            // sets the error text, returns -1

#define MM2GB_HIP(call)                                                                                        \
	do {                                                                                                       \
		hipError_t e_ = (call);                                                                                \
		if (e_ != hipSuccess)                                                                                  \
			return ::mm2gb::fail(std::string(#call) + ": " + hipGetErrorString(e_));                           \
	} while (0)

// One device buffer that only ever grows.
struct DevBuf {
	void  *ptr = nullptr;
	size_t bytes = 0;
	int ensure(size_t need);
	void release();
};

// Buffers replaced by a growth are retired, not freed (engine.hip: hipFree waits for the whole device): this frees them now.
void flush_retired_buffers();

// Page-locked host buffer that only ever grows.
struct PinnedBuf {
	void  *ptr = nullptr;
	size_t bytes = 0;
	int ensure(size_t need);
	void release();
};

// Device-side staging of one host micro-batch: raw anchors + offsets in, f/p out.  Two sets alternate so that the H2D of
// micro-batch k+1, the kernels of k and the D2H of k-1 run at the same time on three streams.
struct IoSet {
	DevBuf raw, offsets, f, p;
	hipEvent_t in_start = nullptr, in_done = nullptr, comp_done = nullptr, out_start = nullptr, out_done = nullptr;
	bool used = false;
};

// Timing + result counters of one enqueued micro-batch (a host call may enqueue several).
struct BatchSlot { hipEvent_t prep0 = nullptr, prep1 = nullptr, score1 = nullptr; };
constexpr int MAX_SLOTS = 256;

struct Engine {
	int device = 0;
	mm2gb_config_t cfg;
	mm2gb_misc_t   misc;
	DevParams      params;
	LaunchCfg      launch;
	hipStream_t    stream = nullptr;       // kernels (== work[0].stream)
	hipStream_t    s_in = nullptr, s_out = nullptr;   // H2D / D2H of the host-buffer paths
	bool           lean_streams = false;   // two HIP streams instead of four: kernels and H2D share one (Engine::init: every engine but the first on its device)
	bool           counted_on_device = false;
	int            n_cu = 256;

	// Work arenas of the kernels of one micro-batch (sized by cap_n / cap_blocks, grow-only).  Two sets, each with its own
	// compute stream: consecutive micro-batches of the host-buffer paths alternate between them, so the tail of micro-batch k
	// (a persistent kernel ends at the pace of its last few chunks) overlaps the body of micro-batch k+1 instead of idling the GPU.
	// Device-pointer calls (mm2gb_score_device) always use set 0 = `stream`.
	struct WorkSet {
		hipStream_t stream = nullptr;
		int64_t cap_n = 0, cap_reads = 0, cap_blocks = 0;
		DevBuf st;
		DevBuf blk_firstcut, blk_pairs, blk_clamped, blk_wmax, blk_read;
		DevBuf chunk_start, chunk_end, chunk_cost, chunk_track, order, long_list, mid_list;
		DevBuf chunk_pp, chunk_kk, chunk_blk, tile_sums, tile_base, bins;
		DevBuf counters, totals, flags;
		DevBuf split_slots, split_part;       // one chunk on several workgroups (k_score's SPLIT build): allocated when first used
		DevBuf gang_slots;                    // one chunk on several workgroups (gangs, k_score's phase 0): GANG_MAX_CHUNKS slots
		std::vector<DevBuf*> all() { return { &st, &blk_firstcut, &blk_pairs, &blk_clamped, &blk_wmax, &blk_read, &chunk_start, &chunk_end, &chunk_cost,
		                                      &chunk_track, &order, &long_list, &mid_list, &chunk_pp, &chunk_kk, &chunk_blk, &tile_sums, &tile_base, &bins, &counters, &totals, &flags, &split_slots, &split_part, &gang_slots }; }
	};
	WorkSet work[2];
	DevBuf lut, dbg;
	// device post-pass (post_kernels.hip), allocated on first use: 41 B/anchor of work arrays (candidates 8, walk records 8, picked 4, two lifting tables 8, the sort's bytes / permutation / way station 13), + chains' arrays, + outputs
	int64_t cap_post_n = 0, cap_post_reads = 0;
	DevBuf post_dbg_reads, post_dbg_tasks, post_dbg_stasks, rmq_dbg_reads, rmq_skey_in, rmq_skey, rmq_sa, rmq_srange, rmq_sort_tmp, post_z, post_fp, post_picked, post_utmp, post_heads, post_nu, post_nkept, post_misc, post_bins, post_order, post_up4, post_up16, post_sort_s, post_sort_perm, post_sort_tmp, post_cls, post_cls_cnt, post_cls_nz, post_read_nz, post_uloc, post_wtask, post_stask, rmq_tied, rmq_sum, rmq_by_y, rmq_ord, rmq_meta, rmq_win, rmq_tree, reg_out;
	DevBuf sd_seeds, sd_seed_off, sd_hit_off, sd_hits, sd_qlen, sd_q_rank, sd_ref_len, sd_ref_rank, sd_seed_read, sd_tmp, sd_n_kept, sd_a_off, sd_out;   // mm2gb_collect_seeds_gpu
	// what the post-pass leaves for the host, two sets: the boundary keeps two batches in flight (the results of batch k are
	// fetched after batch k+1 has been launched)
	struct PostOut {
		DevBuf u_off, a_off, u_out, a_out;
		int64_t *h_totals = nullptr;       // pinned: [0] chains [1] anchors kept
		hipEvent_t done = nullptr;         // post kernels of the batch that owns this set have finished, totals are in h_totals
	} post_out[2];
	DevBuf &post_uoff = post_out[0].u_off, &post_aoff = post_out[0].a_off, &post_uout = post_out[0].u_out, &post_aout = post_out[0].a_out;
	int64_t *&h_post_totals = post_out[0].h_totals;
	hipEvent_t post0 = nullptr, post1 = nullptr;
	hipEvent_t post_fork = nullptr, post_join = nullptr;   // the classes' pass on the second compute stream beside the lifting tables' (engines with streams of their own)
	IoSet io[2];
	uint64_t io_seq = 0;
	PinnedBuf h_slice_off;                 // per-slice read offsets of mm2gb_score_host
	std::vector<hipEvent_t> slice_in;      // sliced mm2gb_chain_gpu: slice k's anchors have arrived
	PinnedBuf h_res_f, h_res_p;            // scores of whole-batch chaining calls (pool.cpp): page-locked, reused, grow-only
	// per-slot read-back (pinned)
	char    *h_small = nullptr;            // one page-locked block: h_counters, h_totals, both post sets' totals
	int32_t *h_counters = nullptr;         // MAX_SLOTS x CNT_WORDS
	int64_t *h_totals = nullptr;           // MAX_SLOTS x 2
	BatchSlot slots[MAX_SLOTS];
	int n_slots = 0;

	mm2gb_stats_t last = {};
	bool misc_valid = false, coop_disabled = false, debug_phases = false, one_compute_stream = false;
	bool post_levels = true;        // the split form's sort level by level over the whole batch, a task per run (MM2GB_POST_SORT=reads: one wave sorts a read from top to bottom)
	bool post_split = true;         // device post-pass: a read's walks shared out by tree over several waves (MM2GB_POST_FORM=fused: one wave sorts and walks a read)
	int64_t team4_min_n = 0;        // micro-batches from this many anchors on send wide-window heavy chunks to 4-wave teams (launch.team4_share_pct)
	int64_t split_max_n = 0;        // micro-batches up to this many anchors run the SPLIT build of k_score (0: never)
	int64_t gang_max_n = 0;         // micro-batches up to this many anchors run the instantiation of k_score with the gang phase
	int64_t last_gang_chunks = 0, last_gang_wgs = 0;        // of the last call: chunks scored by a gang of workgroups, workgroups that started in one
	int64_t last_split_chunks = 0, last_helped_items = 0;   // of the last call: chunks scored strip by strip, items other workgroups took
	bool rmq_tiles_last = false;    // which form the last RMQ call ran (for the debug print)
	std::function<void(const int32_t*)> rmq_tied_ready;   // the NEXT device re-chaining call only: called on the calling thread as soon as the fill's tie counts are on the host, while the call's post-pass and copies still run
	hipEvent_t rmq_fill_done = nullptr;
	int  rmq_team_reads = 0;        // tile form, the NEXT device call only: its first reads that get a whole workgroup each (mm2gb_rmq_chain puts the costliest first); MM2GB_RMQ_TEAM_READS overrides
	bool rmq_abandon_tied = false;  // the NEXT device re-chaining call only: give a read up at its first tie (mm2gb_rmq_chain redoes tied reads on the host whatever the device made of them)
	bool rmq_calibrate = false;     // mm2gb_rmq_chain on this engine scales its cost model by the measured / estimated times of earlier calls (rmq_hybrid.cpp)
	int  rmq_kernel = 0;            // device form of the RMQ fill: 0 tiles (k_rmq_fill_tiles), 1 one anchor per step (k_rmq_fill); MM2GB_RMQ_KERNEL=steps|tiles overrides
	bool lds_contract_ok = false;   // this device reads 0 beyond a workgroup's LDS and saturates v_sad_u32 ... clamp (probed in init)
	int64_t dual_stream_max_n = 16 * 1000 * 1000;   // micro-batches up to this many anchors alternate between the two compute streams

	int  init(const mm2gb_config_t *cfg, const mm2gb_misc_t *misc, int device);
	void shutdown();
	int  set_misc(const mm2gb_misc_t *m);
	int  configure_score();
	int  reserve(int64_t n_anchors, int64_t n_reads, int set = 0);
	int  begin_call();                     // start of a host-level call: resets slots and `last`
	// host buffers (pinned for true asynchrony): H2D, kernels, D2H enqueued on three streams; returns without waiting
	// want_stats = false: no timing events / counter read-back for this micro-batch (the drop-in boundary keeps two host
	// batches in flight and never asks for statistics)
	int  enqueue_host(int64_t n_reads, const int64_t *h_offsets, const mm2gb_anchor_t *h_anchors, int64_t n, int32_t *h_f, int32_t *h_p, bool want_stats = true);
	int  enqueue(int64_t n_reads, const int64_t *d_offsets, const mm2gb_anchor_t *d_anchors, int64_t n, int32_t *d_f, int32_t *d_p, bool want_stats = true, int set = 0);
	int  score_host(int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, int32_t *f, int32_t *p,
	                const std::function<void(int64_t, int64_t)> *slice_done);   // sliced + overlapped, waits for the end
	// backtrack + compaction of a scored micro-batch on the device (all pointers device pointers; d_f / d_p as enqueue() left
	// them); enqueued on the compute stream.  Results stay in post_uoff / post_aoff / post_uout / post_aout; totals land in
	// h_post_totals once the stream has been synchronised.
	int  reserve_post(int64_t n_anchors, int64_t n_reads);                 // work arrays of the post kernels (shared by both result sets)
	int  reserve_post_out(int set, int64_t n_anchors, int64_t n_reads);    // result buffers of one set; never touches the other
	void print_post_debug(int64_t n_reads, const int64_t *d_offsets);   // MM2GB_DEBUG_PHASES: the last post-pass kernel's records, to stderr
	int  enqueue_post(int64_t n_reads, const int64_t *d_offsets, const mm2gb_anchor_t *d_anchors, int64_t n, const int32_t *d_f, const int32_t *d_p,
	                  const mm2gb_rmq_param_t *rmq = nullptr, int out_set = 0);   // rmq given: thresholds of the re-chaining call (lchain.c:355) instead of misc's
	// the boundary's device post-pass: host anchors in (page-locked), H2D + score kernels + post kernels enqueued, nothing waited
	// for; fetch_chains() waits for that batch and copies its chains out (exact sizes)
	int  enqueue_host_chains(int64_t n_reads, const int64_t *h_offsets, const mm2gb_anchor_t *h_anchors, int64_t n, int out_set, bool want_stats = false);
	int  fetch_chains(int out_set, int64_t n_reads, mm2gb_chains_t *out);
	// whole batch on host buffers, chains back, nothing but the chains crosses the link on the way back
	// rmq given: the score fill is mg_lchain_rmq's (k_rmq_fill) instead of the chaining DP; n_tied (optional, n_reads entries)
	// receives, per read, the number of anchors whose range-minimum was tied (results for such a read are not the reference's)
	int  chain_gpu(int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, mm2gb_chains_t *out,
	               const mm2gb_rmq_param_t *rmq = nullptr, int32_t *n_tied = nullptr);
	int  chain_gpu_sliced(int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, mm2gb_chains_t *out, int64_t slice);
	int  collect_seeds(int64_t opt_flag, int64_t n_reads, const int64_t *seed_off, const mm2gb_seed_t *seeds, const int64_t *hit_off, const uint64_t *hits,
	                   const int32_t *qlen, const int32_t *q_rank, int32_t n_ref, const int32_t *ref_len, const int32_t *ref_rank, int64_t *anchor_off, mm2gb_anchor_t *anchors);
	int  sort_seeds(int64_t n_reads, const int64_t *offsets, mm2gb_anchor_t *anchors);
	int  gen_regs(int64_t n_reads, const mm2gb_chains_t *chains, const int32_t *qlen, const uint32_t *hash, int is_qstrand, mm2gb_reg_t *regs);
	int  record_outputs_done(hipEvent_t ev);   // fires when every D2H enqueued so far has landed
	int  sync();
	int  collect_stats();
};

} // namespace mm2gb

// host_scratch: what the engine's host-side callers keep from call to call (the mapper's and the re-chaining call's large arrays, host_chain.h
// HostScratch): an engine serves one thread at a time, so its scratch needs no lock, and it outlives the threads that use it
struct mm2gb_engine { mm2gb::Engine e; void *host_scratch = nullptr; void (*host_scratch_free)(void*) = nullptr; };

namespace mm2gb {
// pool.cpp: one engine, host buffers in, chains out, post-pass on n_threads host threads overlapped with the device
int chain_batch_on_engine(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, int n_threads, mm2gb_chains_t *out, mm2gb_stats_t *stats);
}
