/*
 * mm2gb_chain.h -- C ABI of the MI355X-native chaining library (libmm2gb_chain.so).
 *
 * Two layers are exported:
 *   1. the host-independent core declared here (anchors in, per-anchor score/predecessor or finished chains out);
 *   2. the reference's own drop-in boundary (init/chain/finish/free_stream_gpu) declared in mm2gb_plutils.h.
 *
 * Plain C types only: pointers, sizes, PODs.  "Reference" citations are file:line in the mm2-gb checkout.
 */
#ifndef MM2GB_CHAIN_H
#define MM2GB_CHAIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM2GB_VERSION "0.1-mi355x"

/* One anchor; bit-identical to mm128_t (minimap.h:44):
 *   x = rev<<63 | rid<<32 | ref_pos     y = seg_id<<48 | flags<<40 | q_span<<32 | query_pos   (lchain.c:140-143) */
typedef struct { uint64_t x, y; } mm2gb_anchor_t;

/* Chaining parameter block; field order and types are those of `Misc` (gpu/plutils.h:33-37) so a Misc can be
 * passed by pointer cast.  Values are what build_misc() (map.c:393-426) produces. */
typedef struct {
	int max_iter, max_dist_x, max_dist_y, max_skip, bw, min_cnt, min_score, is_cdna, n_seg;
	float chn_pen_gap, chn_pen_skip;
} mm2gb_misc_t;

/* gpu_config.json, same schema as the reference's presets (gpu/gpu_config.json; parsed at plmem.cu:416-540).
 * Keys that start with "//" are comments.  Unknown keys are ignored.  Fields absent from a file keep their default. */
typedef struct {
	int     num_streams;            /* top level */
	int     min_n;
	int64_t long_seg_buffer_size;
	int64_t max_total_n;            /* may exceed INT32_MAX (plmem.cu:491 reads it as a double) */
	int     max_read;
	int     avg_read_n;             /* optional in the reference schema (plmem.cu:497-539) */
	int     has_max_total_n, has_max_read, has_avg_read_n;
	struct { int blockdim, cut_check_anchors, anchor_per_block; } range_kernel;
	struct { int micro_batch, mid_blockdim, short_griddim, long_griddim, mid_griddim,
	             long_seg_cutoff, mid_seg_cutoff; } score_kernel;
} mm2gb_config_t;

/* Per-call measurements (device times from HIP events on the engine's stream). */
typedef struct {
	int64_t n_anchors, n_reads;
	int64_t n_pairs;            /* sum of predecessor-window sizes == the reference's "anchor pairs" (planalyze.cu:69-83) */
	int64_t n_chunks;           /* independent DP work items found by the planner */
	int64_t n_long_chunks;      /* of those, pipelined over a big team: 8 waves, or a whole workgroup (LDS score ring) */
	int64_t n_mid_chunks;       /* of those, pipelined over a 4-wave team */
	int64_t n_tracked_chunks;   /* of those, run with the max_ii rescue state machine (lchain.c:190-205) */
	int64_t n_clamped_blocks;   /* planner blocks containing a window cut by max_iter (lchain.c:173) */
	float   ms_h2d, ms_prep, ms_score, ms_d2h, ms_total;
	float   ms_post;            /* device post-pass kernels (mm2gb_chain_gpu), 0 when the post-pass ran on the host */
} mm2gb_stats_t;

typedef struct mm2gb_engine mm2gb_engine_t;

/* ---- errors: every int-returning call gives 0 on success, negative on failure; text via mm2gb_last_error() ---- */
const char *mm2gb_last_error(void);
const char *mm2gb_version(void);

/* ---- configuration (replaces plmem_parse_gpu_config / plmem_config_kernels / plmem_config_batch, plmem.cu:373-540) ---- */
void mm2gb_config_defaults(mm2gb_config_t *cfg);                       /* tuned for MI355X, see mm2-gb_amd/mi355x_config.json */
int  mm2gb_config_parse(const char *json_text, mm2gb_config_t *cfg);   /* starts from defaults */
int  mm2gb_config_load(const char *path, mm2gb_config_t *cfg);

/* ---- engine: one per (process, device); owns streams and device arenas (replaces plmem_stream_initialize,
 *      plmem.cu:558-624, and the constant uploads plrange.cu:225-239 / plscore.cu:491-502) ---- */
int  mm2gb_device_count(void);
/* Node awareness (the reference uses device 0 and pins nothing, gpu/plmem.cu:426,462,499).  The NUMA node of a device's PCIe root
 * (/sys/bus/pci/devices/<bdf>/numa_node; -1 unknown) and a move of the CALLING thread -- and of the threads it starts afterwards -- onto the
 * CPUs of that node which the process may use: 0 when nothing changed (node unknown, none of its CPUs usable, MM2GB_NUMA=0), else the
 * CPUs in the new mask.  Pool workers, batcher workers and bench.py's ranks call it before they allocate page-locked staging (first
 * touch then lands on that node).  mm2gb_numa_cpus_for_bdf is the parsing alone, against a given sysfs root (tests use a made-up tree):
 * returns the number of CPUs of the node (cpus[] filled up to max_cpus), 0 when unknown. */
int  mm2gb_device_numa_node(int device);
int  mm2gb_pin_thread_to_device(int device);
int  mm2gb_numa_cpus_for_bdf(const char *bdf, const char *sysfs_root_dir, int32_t *node_out, int32_t *cpus, int32_t max_cpus);
mm2gb_engine_t *mm2gb_engine_create(const mm2gb_config_t *cfg, const mm2gb_misc_t *misc, int device);
void mm2gb_engine_destroy(mm2gb_engine_t *eng);
int  mm2gb_engine_set_misc(mm2gb_engine_t *eng, const mm2gb_misc_t *misc);
int  mm2gb_engine_device(const mm2gb_engine_t *eng);
/* of the engine's last completed call: heavy chunks that were scored strip by strip with the help of idle workgroups (micro-batches of
 * up to MM2GB_SPLIT_MAX_ANCHORS anchors; 0 = never, the default: the build is exact but measured slower), and the items
 * of such chunks that workgroups other than the owner took */
void mm2gb_engine_split_counts(const mm2gb_engine_t *eng, int64_t *chunks, int64_t *helped_items);
/* of the engine's last completed call: chunks that a GANG of workgroups scored -- a chunk whose share of the micro-batch's pairs is worth two
 * workgroups or more is cut into strips that several workgroups take in turn (what plscore.cu's long-segment kernel does with one block per
 * segment, plscore.cu:187-290, cannot: a segment there is one block's) -- and the workgroups that started in a gang.  MM2GB_GANG_MAX (default 8,
 * 0 = off) bounds the workgroups per chunk; micro-batches of more than MM2GB_GANG_MAX_ANCHORS anchors (default 150 M) run the kernel without
 * the gang phase: they fill the machine with whole chunks */
void mm2gb_engine_gang_counts(const mm2gb_engine_t *eng, int64_t *chunks, int64_t *workgroups);
int  mm2gb_has_gang_build(void);    /* 1 (round 3: the gang phase was a build option) */
int  mm2gb_has_split_build(void);   /* 1 if the library was built with `make SPLIT=1` (k_score's one-chunk-on-several-workgroups build; MM2GB_SPLIT_MAX_ANCHORS is ignored otherwise) */
/* make sure arenas can take a micro-batch of this size (grows, never shrinks) */
int  mm2gb_engine_reserve(mm2gb_engine_t *eng, int64_t n_anchors, int64_t n_reads);

/* ---- score generation: range selection + DP (replaces plrange_async_range_selection, plscore_async_*;
 *      call chain plchain.cu:346-461).  Reads are concatenated: read r owns anchors [offsets[r], offsets[r+1]).
 *      f[i]  = best chain score ending at anchor i (lchain.c:202)
 *      p[i]  = distance back to its predecessor, i - j (0 = no predecessor); int32 because the max_ii rescue
 *              (lchain.c:196-201) can reach further than the reference GPU path's uint16 (plmem.cuh:30). ---- */
int mm2gb_score_host(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                     int32_t *f, int32_t *p, mm2gb_stats_t *stats);
/* Same, all pointers are DEVICE pointers in the engine's device; enqueued on the engine's compute stream and
 * returns without synchronising.  Use mm2gb_engine_sync() then mm2gb_engine_stats(). */
int mm2gb_score_device(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *d_offsets, const mm2gb_anchor_t *d_anchors,
                       int64_t n_anchors, int32_t *d_f, int32_t *d_p);
int mm2gb_engine_sync(mm2gb_engine_t *eng);
int mm2gb_engine_stats(mm2gb_engine_t *eng, mm2gb_stats_t *stats);
/* HIP stream handle (hipStream_t) the engine launches on -- for callers that time with their own HIP events. */
void *mm2gb_engine_stream(mm2gb_engine_t *eng);
/* time of the score kernels alone for the last mm2gb_score_device call (ms, valid after mm2gb_engine_sync) */
float mm2gb_engine_last_kernel_ms(mm2gb_engine_t *eng);

/* ---- full chaining of a batch on host buffers: scores on the GPU, then backtrack + compaction
 *      (replaces plchain_cal_score_async + plchain_post_gpu_helper, plchain.cu:201-464).
 *      The post-pass starts on each slice of reads as soon as its scores are back, while later slices are on the device.
 *      Outputs are malloc'd by the library: u_off[n_reads+1] indexes u[], a_off[n_reads+1] indexes a[].
 *      Free with mm2gb_chains_free().  Stats of a multi-device call: counts are summed, times are the slowest device's. ---- */
typedef struct {
	int64_t *u_off;   uint64_t *u;            /* chains per read: score<<32 | n_anchors (lchain.c:145) */
	int64_t *a_off;   mm2gb_anchor_t *a;      /* compacted anchors per read, chain by chain (lchain.c:78-111) */
} mm2gb_chains_t;
int  mm2gb_chain_host(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                      int n_threads, mm2gb_chains_t *out, mm2gb_stats_t *stats);
void mm2gb_chains_free(mm2gb_chains_t *out);

/* ---- the same with the post-pass on the device too: backtrack (mg_chain_backtrack, lchain.c:27-76) and
 *      compaction (compact_a, lchain.c:78-111) run as kernels behind the score kernel -- one wave per read, the reference's
 *      sort order (radix_sort_128x, ksort.h:98-151) reproduced element for element -- and only chains and compacted anchors
 *      come back over the link.  Same outputs as mm2gb_chain_host, no host threads.  One micro-batch per call.
 *      mm2gb_post_device: the post-pass alone on device-resident scores (as mm2gb_score_device left them); results stay on the
 *      device, the totals and the kernels' time are reported.  Synchronises the engine's stream. ---- */
int  mm2gb_chain_gpu(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                     mm2gb_chains_t *out, mm2gb_stats_t *stats);
int  mm2gb_post_device(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *d_offsets, const mm2gb_anchor_t *d_anchors, int64_t n_anchors,
                       const int32_t *d_f, const int32_t *d_p, int64_t *n_chains, int64_t *n_kept, float *ms);
/* the same, enqueued only (mm2gb_post_device_totals waits and reads the totals): the post-pass of batch k on one engine beside the score kernels of
 * batch k+1 on another -- a stream of micro-batches pays the longer of the two, not their sum, where the kernels can share the chip */
int  mm2gb_post_device_enqueue(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *d_offsets, const mm2gb_anchor_t *d_anchors, int64_t n_anchors,
                               const int32_t *d_f, const int32_t *d_p);
int  mm2gb_post_device_totals(mm2gb_engine_t *eng, int64_t *n_chains, int64_t *n_kept, float *ms);
/* a digest of the chains the last post-pass on this engine left on the device (offsets of chains, offsets of anchors, chains, anchors: four
 * position-dependent sums folded on the host), for comparing two settings or builds on batches too large for the oracle */
int  mm2gb_post_device_digest(mm2gb_engine_t *eng, int64_t n_reads, uint64_t *digest);
/* for tests and debugging: the per-anchor scores f[] and predecessor distances p[] (i - predecessor, 0 = none) that the engine's LAST
 * mm2gb_chain_gpu / mm2gb_rmq_chain_gpu call left on the device (n = that call's anchors) */
int  mm2gb_debug_last_fill(mm2gb_engine_t *eng, int64_t n, int32_t *f, int32_t *p);

/* ---- RMQ re-chaining (mg_lchain_rmq, lchain.c:250-369, called per read from post_chaining_helper, map.c:444-456,
 *      on the anchors the first chaining kept, sorted by x).  Parameters in the order of mg_lchain_rmq's argument list.
 *      mm2gb_rmq_chain_gpu: a batch of reads; score fill, backtrack and compaction all on the device.  The reference resolves
 *      ties between equal range-minimum priorities by the shape of its AVL tree (krmq.h:110-147), which no closed form
 *      reproduces: n_tied[r] (optional) counts the anchors of read r where that happened AND the tied elements do not all leave
 *      the anchor with the same score and predecessor (the tile kernel weighs a tie on the spot; the one-anchor-per-step
 *      kernel and MM2GB_RMQ_TIES=strict count every tie); the chains of a read with n_tied[r] != 0 may differ from the reference's
 *      and are for the caller to discard (mm2gb_lchain_rmq does).
 *      max_chn_skip: at or above cap_rmq_size (or INT32_MAX) the inner walk can never be cut short (lchain.c:329-333) and either kernel
 *      form fills the batch; below it the one-anchor-per-step kernel does, whose inner walk meets the candidates in the reference's
 *      order and keeps the skip counter and its marks (round 6; MM2GB_RMQ_SKIP=ignore: exhaustive whatever the value, as before).
 *      mm2gb_rmq_chain: the batch call that is exact for EVERY read with the device carrying the load (csrc/rmq_hybrid.cpp): reads are
 *      dealt between the kernel form and the host form by estimated cost so that both finish together (a read inside a tandem array is
 *      a hundred times the median and would be one wave's alone), both run at the same time, and reads the kernel reports a tie for
 *      are redone by the host form.  where[r] (optional): 0 device, 1 host threads (cost), 2 host threads (tie).
 *      mm2gb_lchain_rmq: one read, signature and ownership of mg_lchain_rmq; answered by the host form below (exact for every read);
 *      with MM2GB_RMQ=gpu by the kernel, and then a read that met a tie is redone by the host form.  The host PROGRAM's mg_lchain_rmq
 *      is never called: the library does not import it. ---- */
typedef struct {
	int max_dist, max_dist_inner, bw, max_chn_skip, cap_rmq_size, min_cnt, min_sc;
	float chn_pen_gap, chn_pen_skip;
} mm2gb_rmq_param_t;
int  mm2gb_rmq_chain_gpu(mm2gb_engine_t *eng, const mm2gb_rmq_param_t *prm, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                         mm2gb_chains_t *out, int32_t *n_tied, mm2gb_stats_t *stats);
typedef struct { int64_t n_device, n_host_cost, n_host_tie; double est_device_s, est_host_s, device_s, host_s, tie_s, total_s; int32_t device_kernel /* 0 tiles, 1 steps */, n_team /* device reads a whole workgroup filled */; } mm2gb_rmq_deal_t;
/* device form of the fill for this engine's later calls: 0 the tile kernel (64 anchors per step of a wave; best where inner windows hold up to a few hundred
 * anchors), 1 one anchor per step (best where they hold many: its inner scan passes blocks over per anchor).  mm2gb_rmq_chain picks per call. */
int  mm2gb_engine_set_rmq_kernel(mm2gb_engine_t *eng, int kind);
/* tile kernel, this engine's NEXT mm2gb_rmq_chain_gpu call only: its first n reads are filled by a whole workgroup each (the sweeps over a tile's inner
 * window shared by its waves) instead of one wave -- for the few reads of a batch that would otherwise set its pace.  mm2gb_rmq_chain sets it. */
int  mm2gb_engine_set_rmq_team_reads(mm2gb_engine_t *eng, int n);
int  mm2gb_rmq_chain(mm2gb_engine_t *eng, const mm2gb_rmq_param_t *prm, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                     int n_threads, mm2gb_chains_t *out, int32_t *where, mm2gb_rmq_deal_t *deal);
/* The same on host threads, O(log n) per anchor, the reference's answer for EVERY read at any max_chn_skip (csrc/rmq_host.cpp): a read
 * is first done with a tournament tree of fixed shape over its anchors' (y, index) ranks, which gives the reference's answer as long as
 * one anchor in range holds the smallest priority, or all that hold it leave the anchor with the same score and predecessor; at the
 * first tie that decides something (which element the reference returns then follows from its tree's shape; with a skip limit: at the
 * first tie) the read is done again with the reference's own tree -- an AVL tree with krmq.h's insertion, deletion, rotation and
 * subtree-minimum rules, on arrays (MM2GB_RMQ_TREE=avl: that tree for every read).  n_tied[r] (may be NULL) = 1 for a read that was
 * done again, else 0: information only, the chains are exact either way. */
int  mm2gb_rmq_chain_host(const mm2gb_rmq_param_t *prm, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, int n_threads,
                          mm2gb_chains_t *out, int32_t *n_tied);
/* the same for reads that are known to meet a tie (a device call counted it): the reference's tree for every read at once */
int  mm2gb_rmq_chain_host_tied(const mm2gb_rmq_param_t *prm, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors, int n_threads, mm2gb_chains_t *out);
mm2gb_anchor_t *mm2gb_lchain_rmq(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc,
                                 float chn_pen_gap, float chn_pen_skip, int64_t n, mm2gb_anchor_t *a, int *n_u_, uint64_t **_u, void *km);
void mm2gb_lchain_rmq_counts(int64_t *calls, int64_t *tied_calls);   /* single-read calls so far, and how many of them met a tie */

/* ---- the formats either side of the path, on the device with the reference's exact orders:
 *      mm2gb_sort_seeds_gpu: the seed sort of collect_seed_hits (map.c:329): every read's anchors sorted by x IN PLACE exactly as
 *      radix_sort_128x leaves them (order of equal x included), so unsorted seeds can go straight into the chaining calls;
 *      mm2gb_gen_regs_gpu: mm_gen_regs (hit.c:52-88): one hit record per chain -- best score first, ties by the hash of the first
 *      anchor and the read's hash (map.c:590-592), coordinates and fuzzy lengths of hit.c:8-38.  mm2gb_reg_t is the leading
 *      72 bytes of mm_reg1_t (minimap.h:104-119), i.e. everything but the alignment pointer; regs must hold
 *      chains->u_off[n_reads] records.  qlen / hash: one per read. ---- */
typedef struct {
	int32_t id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, mlen, blen, n_sub, score0;
	uint32_t flags;          /* mm_reg1_t's bit-field word: rev = bit 10 */
	uint32_t hash;
	float div;
} mm2gb_reg_t;
int  mm2gb_sort_seeds_gpu(mm2gb_engine_t *eng, int64_t n_reads, const int64_t *offsets, mm2gb_anchor_t *anchors);
/*      mm2gb_collect_seeds_gpu: collect_seed_hits (map.c:295-331) with skip_seed (map.c:205-227): the matches mm_collect_matches
 *      (seed.c:98) returned for every read -- seeds[] = the leading 16 bytes of each mm_seed_t (mmpriv.h:40-46), read by read
 *      (seed_off: n_reads + 1), hits[] = the arrays mm_seed_t::cr points at, seed by seed (hit_off: n_seeds + 1) -- become anchors
 *      (map.c:311-324), dropped hits removed, every read's anchors sorted as radix_sort_128x leaves them (map.c:329): the input of the
 *      chaining calls.  opt_flag: the MM_F_* bits of mm_mapopt_t::flag that the function looks at (NO_DIAG, NO_DUAL, FOR_ONLY,
 *      REV_ONLY, QSTRAND; others ignored).  qlen: per read.  Name tests (strcmp(qname, name), map.c:211) are given as ranks in one
 *      common order -- q_rank per read, ref_rank per reference sequence, equal names <=> equal ranks; both may be NULL when neither
 *      NO_DIAG nor NO_DUAL is set.  ref_len (per reference sequence) is needed for NO_DIAG and QSTRAND.  anchors must hold
 *      hit_off[n_seeds] elements; anchor_off (n_reads + 1) receives where each read's anchors begin. ---- */
typedef struct { uint32_t n, q_pos, span_flt, seg_tandem; } mm2gb_seed_t;
int  mm2gb_collect_seeds_gpu(mm2gb_engine_t *eng, int64_t opt_flag, int64_t n_reads, const int64_t *seed_off, const mm2gb_seed_t *seeds,
                             const int64_t *hit_off, const uint64_t *hits, const int32_t *qlen, const int32_t *q_rank,
                             int32_t n_ref, const int32_t *ref_len, const int32_t *ref_rank, int64_t *anchor_off, mm2gb_anchor_t *anchors);
int  mm2gb_gen_regs_gpu(mm2gb_engine_t *eng, int64_t n_reads, const mm2gb_chains_t *chains, const int32_t *qlen, const uint32_t *hash,
                        int is_qstrand, mm2gb_reg_t *regs);

/* ---- from sequence to seed matches on the host (csrc/seeding.cpp), the reference's definitions to the letter:
 *      mm2gb_sketch: mm_sketch (sketch.c:77-143, no homopolymer compression): pairs (hash << 8 | span, rid << 32 | last_pos << 1 | strand);
 *      mm2gb_index_*: every minimizer of the reference sequences -> its occurrences in ascending order, what mm_idx_get returns
 *      (index.c:81-98, 213-262); mid_occ as mm_mapopt_update computes it (options.c:78-84, index.c:186-211);
 *      mm2gb_collect_matches: mm_collect_matches (seed.c:98-131, with seed.c:5-96) for one read of one segment: the arrays
 *      mm2gb_collect_seeds_gpu takes, plus rep_len and the minimizer positions the host's mapq / divergence estimates use.
 *      Free a matches record with mm2gb_matches_free, a sketch with mm2gb_free. ---- */
typedef struct mm2gb_index mm2gb_index_t;
typedef struct { int32_t mid_occ, max_max_occ, occ_dist; float q_occ_frac; } mm2gb_seed_opt_t;   /* mm_mapopt_t: mid_occ, max_max_occ, occ_dist, q_occ_frac */
typedef struct {
	int32_t n_seeds, rep_len, n_mini_pos, pad_;
	int64_t n_hits;
	mm2gb_seed_t *seeds;      /* n_seeds */
	uint64_t *hits;           /* n_hits: seed 0's, seed 1's, ... */
	uint64_t *mini_pos;       /* n_mini_pos: q_span << 32 | position of the minimizer's last base (seed.c:125) */
} mm2gb_matches_t;
int  mm2gb_sketch(const char *seq, int32_t len, int w, int k, uint32_t rid, uint64_t **out_xy, int64_t *n_out);
mm2gb_index_t *mm2gb_index_build(int k, int w, int32_t n_seq, const char *const *seqs, const int32_t *lens, int n_threads);
void mm2gb_index_destroy(mm2gb_index_t *ix);
int64_t mm2gb_index_size(const mm2gb_index_t *ix, int64_t *n_occurrences);     /* distinct minimizers */
int32_t mm2gb_index_mid_occ(const mm2gb_index_t *ix, float mid_occ_frac, int32_t min_mid_occ, int32_t max_mid_occ);
int  mm2gb_collect_seeds_host(int64_t opt_flag, int64_t n_reads, const int64_t *seed_off, const mm2gb_seed_t *seeds, const int64_t *hit_off,
                              const uint64_t *hits, const int32_t *qlen, const int32_t *q_rank, int32_t n_ref, const int32_t *ref_len,
                              const int32_t *ref_rank, int n_threads, int64_t *anchor_off, mm2gb_anchor_t *anchors);   /* = mm2gb_collect_seeds_gpu, on host threads */
int  mm2gb_collect_matches(const mm2gb_index_t *ix, const char *seq, int32_t len, const mm2gb_seed_opt_t *opt, mm2gb_matches_t *out);
void mm2gb_matches_free(mm2gb_matches_t *m);

/* ---- reads in, PAF out (csrc/mapper.cpp): seeding on host threads, anchors / chaining / re-chaining / hit records on the
 *      device, primary-secondary decisions, divergence, mapping quality and the PAF line on the host, written from scratch after
 *      mm_map_frag (map.c:630-790) for single-segment reads without base-level alignment.  Options: the fields of mm_mapopt_t this
 *      path looks at, mm2gb_map_opt_init sets the defaults of mm_mapopt_init (options.c:15-75); chaining runs at max-chain-skip =
 *      infinity (the GPU path's contract).  paf: malloc'd text, one line per hit in read order (free with mm2gb_free). ---- */
typedef struct {
	int64_t flag;              /* MM_F_FOR_ONLY | MM_F_REV_ONLY only */
	int32_t seed, mid_occ, min_mid_occ, max_mid_occ, max_max_occ, occ_dist;
	float   mid_occ_frac, q_occ_frac;
	int32_t min_cnt, min_chain_score, bw, bw_long, max_gap, max_gap_ref, max_chain_iter;
	int32_t rmq_inner_dist, rmq_size_cap, rmq_rescue_size;
	float   rmq_rescue_ratio, chain_gap_scale, chain_skip_scale;
	float   mask_level; int32_t mask_len; float pri_ratio; int32_t best_n;
	int32_t host_threads;      /* 0: every CPU the process may use, at most 32 */
	int32_t seeds_on_device;   /* matches -> sorted anchors: 1 on the device, -1 on host threads, 0 by batch size */
	int32_t rechain_on_device; /* mg_lchain_rmq's fill: 0 (default) mm2gb_rmq_chain -- device and host threads at the same time, reads dealt by cost, ties redone on the host; 1 every read on the device first (ties redone on host threads); -1 host threads only */
} mm2gb_map_opt_t;
typedef struct { int64_t n_reads, n_mapped, n_anchors, n_chains, n_rechained, n_rmq_tied; double s_seed, s_anchors, s_chain, s_rechain, s_regs, s_post; } mm2gb_map_stats_t;   /* s_*: seconds per stage */
void mm2gb_map_opt_init(mm2gb_map_opt_t *opt);
int  mm2gb_map_reads(mm2gb_engine_t *eng, const mm2gb_index_t *ix, int k, const char *const *ref_names, const int32_t *ref_lens, int32_t n_ref,
                     const mm2gb_map_opt_t *opt, int32_t n_reads, const char *const *names, const char *const *seqs, const int32_t *lens,
                     char **paf_out, int64_t *paf_len, mm2gb_map_stats_t *stats);

/* the same over several engines (one per device; reads shard, no exchange): contiguous runs of reads balanced by bases, one host thread per
 * engine, PAF in read order; counts are summed, stage times are the slowest engine's */
int  mm2gb_map_reads_multi(mm2gb_engine_t *const *engines, int n_engines, const mm2gb_index_t *ix, int k, const char *const *ref_names, const int32_t *ref_lens,
                           int32_t n_ref, const mm2gb_map_opt_t *opt, int32_t n_reads, const char *const *names, const char *const *seqs, const int32_t *lens,
                           char **paf_out, int64_t *paf_len, mm2gb_map_stats_t *stats);
/* A run of any size as a stream of batches (role of worker_for's batch rotation, map.c:924-1153): reads cut into chunks of about
 * chunk_bases bases (<= 0: 96 Mbp), every engine's host thread takes the next chunk and maps it from seeding to PAF.  Several engines
 * per device overlap one chunk's host stages with another's kernels; engines on several devices shard the reads.  PAF in read order;
 * stats->s_*: seconds per stage summed over chunks (stages overlap: not wall time). */
int  mm2gb_map_reads_stream(mm2gb_engine_t *const *engines, int n_engines, const mm2gb_index_t *ix, int k, const char *const *ref_names, const int32_t *ref_lens,
                            int32_t n_ref, const mm2gb_map_opt_t *opt, int32_t n_reads, const char *const *names, const char *const *seqs, const int32_t *lens,
                            int64_t chunk_bases, char **paf_out, int64_t *paf_len, mm2gb_map_stats_t *stats);
/* An engine that maps (or re-chains: mm2gb_rmq_chain) keeps the largest host arrays of those calls -- matches, anchors, the re-chaining
 * gathers and the spliced chains, about 70 bytes per anchor of its largest batch so far, several gigabytes at 100 M anchors -- from call to
 * call, because touching fresh pages costs more than filling them.  They go with the engine; this gives the memory back at once. */
int  mm2gb_engine_release_host_scratch(mm2gb_engine_t *eng);

/* ---- several devices in one process: reads are independent, so a batch is dealt to the devices as contiguous
 *      runs of reads with about the same number of anchors; each device has its own engine (arenas, three streams) and host
 *      thread, nothing is exchanged between devices, results come back in read order.  devices == NULL: 0..n_devices-1;
 *      n_devices <= 0: every visible device.  A device id may repeat (two engines sharing one GPU).
 *      mm2gb_pool_score_host: first_read_of_device (optional, n_devices+1 entries) reports how the reads were dealt.
 *      Replaces the reference's one-GPU stream_setup (plmem.cu:370, plchain.cu:299) for hosts that own whole batches. ---- */
typedef struct mm2gb_pool mm2gb_pool_t;
mm2gb_pool_t *mm2gb_pool_create(const mm2gb_config_t *cfg, const mm2gb_misc_t *misc, int n_devices, const int *devices);
void mm2gb_pool_destroy(mm2gb_pool_t *pool);
int  mm2gb_pool_size(const mm2gb_pool_t *pool);
int  mm2gb_pool_device(const mm2gb_pool_t *pool, int k);              /* HIP device of engine k, -1 if out of range */
int  mm2gb_pool_set_misc(mm2gb_pool_t *pool, const mm2gb_misc_t *misc);
int  mm2gb_pool_score_host(mm2gb_pool_t *pool, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                           int32_t *f, int32_t *p, mm2gb_stats_t *stats, int64_t *first_read_of_device);
int  mm2gb_pool_chain_host(mm2gb_pool_t *pool, int64_t n_reads, const int64_t *offsets, const mm2gb_anchor_t *anchors,
                           int n_threads, mm2gb_chains_t *out, mm2gb_stats_t *stats);

/* ---- batch accumulator + dispatcher (role of mm_trbuf_t, mm_trbuf_is_full and the batch rotation of worker_for,
 *      map.c:23-157, 886-922, 1026-1075): feed reads one at a time from any number of host threads; they are grouped into batches
 *      of at most max_total_n x micro_batch anchors / max_read x micro_batch reads (a read that would overflow the batch starts
 *      the next one, as in map.c:887-920; a read larger than the limit is a batch of its own), closed batches are dealt to the
 *      devices' workers as they become free (one engine + one host thread per device, nothing moves between devices), and every
 *      read's chains come back through `done` (called on worker threads, reads of one batch in the order they were added).
 *      min_n (gpu_config.json) routes reads with fewer anchors to a lane of their own -- batched separately, still on the GPU;
 *      there is no CPU fallback.  post_threads > 0: host threads per device for backtrack + compaction, overlapped with the
 *      device; 0: the device post-pass (mm2gb_chain_gpu).  mm2gb_batcher_add blocks while every batch buffer is in flight.
 *      mm2gb_plan_batches: the grouping rule alone, for a sequence of reads added in order by one thread (no GPU needed):
 *      writes the batch id (in order of creation) and lane (0 big, 1 small) of every read, returns the number of batches. ---- */
typedef struct mm2gb_batcher mm2gb_batcher_t;
typedef void (*mm2gb_read_done_fn)(void *user, int64_t read_id, int n_u, const uint64_t *u, int64_t n_a, const mm2gb_anchor_t *a);
#define MM2GB_BATCHER_MAX_ENGINES 16
typedef struct {
	int64_t reads, anchors;
	int64_t reads_per_lane[2], batches[2];                         /* [0] reads of at least min_n anchors, [1] smaller ones */
	int64_t batches_per_engine[MM2GB_BATCHER_MAX_ENGINES];
	int     n_engines;
} mm2gb_batcher_stats_t;
mm2gb_batcher_t *mm2gb_batcher_create(const mm2gb_config_t *cfg, const mm2gb_misc_t *misc, int n_devices, const int *devices,
                                      int post_threads, mm2gb_read_done_fn done, void *user);
int  mm2gb_batcher_add(mm2gb_batcher_t *b, int64_t read_id, const mm2gb_anchor_t *a, int64_t n);
/* the reads of a packed batch added one at a time by n_producers threads of the library (read r gets the id first_id + r) */
int  mm2gb_batcher_feed(mm2gb_batcher_t *b, int64_t n_reads, int64_t first_id, const int64_t *offsets, const mm2gb_anchor_t *anchors, int n_producers);
int  mm2gb_batcher_flush(mm2gb_batcher_t *b);                     /* close the partial batches, wait until every read is delivered */
int  mm2gb_batcher_stats(mm2gb_batcher_t *b, mm2gb_batcher_stats_t *out);
void mm2gb_batcher_destroy(mm2gb_batcher_t *b);
int64_t mm2gb_plan_batches(int64_t n_reads, const int64_t *n_anchors, int64_t max_total_n, int max_read, int min_n,
                           int32_t *batch_of_read, int32_t *lane_of_read);

/* ---- host post-pass on given f / relative p for ONE read (restates mg_chain_backtrack + compact_a,
 *      lchain.c:27-111, including the radix_sort_128x order, ksort.h:98-151).  Returns number of chains;
 *      *u_out / *a_out are malloc'd (NULL when 0).  Exposed for tests and for hosts that keep their own loop. ---- */
int  mm2gb_backtrack_host(const mm2gb_misc_t *misc, int64_t n, const mm2gb_anchor_t *a, const int32_t *f, const int32_t *p_rel,
                          uint64_t **u_out, mm2gb_anchor_t **a_out);
void mm2gb_free(void *ptr);

/* ---- synchronous single-read entry with the signature of mg_lchain_dp (mmpriv.h:84-85, lchain.c:148-149):
 *      consumes a[] (freed with the host's kfree when linked into minimap2, free() otherwise), returns the compacted
 *      anchors and *_u allocated the same way.  max_skip is ignored: the GPU path is exhaustive (== INT32_MAX). ---- */
mm2gb_anchor_t *mm2gb_lchain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc,
                                float chn_pen_gap, float chn_pen_skip, int is_cdna, int n_seg, int64_t n, mm2gb_anchor_t *a,
                                int *n_u_, uint64_t **_u, void *km);

/* ---- deterministic synthetic reads for benchmarks (SURVEY 8d recipe; xorshift64* seeded per read id).
 *      Two-call protocol: count, then fill.  len_lo/len_hi in bases. ---- */
int64_t mm2gb_synth_count(uint64_t seed, int64_t first_read, int64_t n_reads, int len_lo, int len_hi, int64_t *offsets /* n_reads+1 */);
int     mm2gb_synth_fill(uint64_t seed, int64_t first_read, int64_t n_reads, int len_lo, int len_hi, const int64_t *offsets,
                         mm2gb_anchor_t *anchors, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
