/*
 * mm2gb_plutils.h -- the drop-in boundary: the four entry points minimap2's batched host calls for --gpu-chain, with the
 * record layouts they exchange.  A host built from the mm2-gb sources with -D__AMD_SPLIT_KERNELS__ links against
 * libmm2gb_chain.so unchanged; it keeps using its own gpu/plutils.h -- this header describes the
 * same ABI for everyone else (tests, other hosts).
 *
 * Replaces: gpu/plutils.h:98-104 (prototypes), implemented in the reference at gpu/plchain.cu:466-561.
 * Callers in the reference: main.c:445-447 (init), main.c:465 (free), map.c:1026 (chain), map.c:1069 (finish).
 */
#ifndef MM2GB_PLUTILS_H
#define MM2GB_PLUTILS_H

#include <stddef.h>
#include <stdint.h>
#include "mm2gb_chain.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as Misc (gpu/plutils.h:33-37). */
typedef mm2gb_misc_t mm2gb_Misc;

/* Per-sequence bookkeeping the host keeps next to every read; layout of mm_seq_meta_t (gpu/plutils.h:19-31). */
typedef struct {
	long     i;            /* read index inside the mini-batch */
	int      seg_id;
	char     name[200];
	uint32_t len;
	int      n_alt;
	int      is_alt;
	int      qlen_sum;
} mm2gb_seq_meta_t;

/* One read travelling through seed -> chain -> align; layout of chain_read_t (gpu/plutils.h:45-73, non-debug build).
 * The chaining step reads a/n and writes a, u, n_u (a is replaced by the compacted array, or freed and set to 0 when
 * nothing chains: plchain.cu:129-145). */
typedef struct mm2gb_chain_read_s {
	mm2gb_seq_meta_t seq;
	const char **qseqs;
	int         *qlens;
	int          n_seg;
	int          rep_len;
	int          frag_gap;
	uint64_t    *mini_pos;
	int          n_mini_pos;
	mm2gb_anchor_t *a;     /* anchors sorted by x, allocated from the batch's kalloc arena */
	int64_t      n;
	uint64_t    *u;        /* chains: score<<32 | count */
	int          n_u;
} mm2gb_chain_read_t;

/* The host's index and option records are opaque here; they are handed back to the host callbacks below.  One exception: to answer a
 * batch's re-chaining calls ahead of the host's callback (below, "re-chaining ahead") the library reads the LEADING fields of mm_mapopt_t
 * (minimap.h:128-145) through this mirror -- the ones map.c:444-451 reads when it decides whether a read is re-chained and with what.
 * tests/test_host_cpu.py compiles the reference header and compares every offset. */
struct mm_idx_s;
struct mm_mapopt_s;
typedef struct {
	int64_t flag;
	int seed, sdust_thres, max_qlen;
	int bw, bw_long;
	int max_gap, max_gap_ref;
	int max_frag_len;
	int max_chain_skip, max_chain_iter;
	int min_cnt, min_chain_score;
	float chain_gap_scale, chain_skip_scale;
	int rmq_size_cap, rmq_inner_dist;
	int rmq_rescue_size;
	float rmq_rescue_ratio;
} mm2gb_mapopt_head_t;

/* gpu/plutils.h:98-99, plchain.cu:470-486.  Reads gpu_config_file, creates one engine per configured stream, reports
 * the batch limits the host should accumulate to: *max_total_n = max_total_n x micro_batch anchors,
 * *max_reads = max_read x micro_batch, *min_n = min_n (plmem.cu:616-617).  Fatal problems print to stderr and exit(1),
 * as the reference does (plmem.cu:390-412). */
void init_stream_gpu(size_t *max_total_n, int *max_reads, int *min_n, char gpu_config_file[], mm2gb_Misc misc);

/* gpu/plutils.h:104, plchain.cu:496-509.  Launches *in_arr_ptr (count *n_read_ptr) asynchronously on stream `thread_id`
 * and hands back, through the same two pointers, the batch launched by the previous call on that stream with chaining
 * finished and post_chaining_helper() applied (NULL / 0 when there was none).  `km` is the kalloc arena of the batch being
 * handed back (map.c:1026 passes launched_batch.km). */
void chain_stream_gpu(const struct mm_idx_s *mi, const struct mm_mapopt_s *opt, mm2gb_chain_read_t **in_arr_ptr, int *n_read_ptr,
                      int thread_id, void *km);

/* gpu/plutils.h:100-101, plchain.cu:518-546.  Completes the batch still in flight on stream `num_batch` (= thread id). */
void finish_stream_gpu(const struct mm_idx_s *mi, const struct mm_mapopt_s *opt, mm2gb_chain_read_t **batches, int *num_reads,
                       int num_batch, void *km);

/* gpu/plutils.h:102, plchain.cu:549-557. */
void free_stream_gpu(int n_threads);

/* ---- re-chaining ahead (csrc/rechain_ahead.cpp).  post_chaining_helper (map.c:428-456), which chain_stream_gpu / finish_stream_gpu call for
 * every read they hand back (plchain.cu:502-507, 539-541), re-chains most long reads with mg_lchain_rmq (lchain.c:250-369), one read per call on
 * the calling thread.  In a host linked with -Wl,--wrap=mg_lchain_rmq those calls arrive at __wrap_mg_lchain_rmq below; the
 * boundary calls then answer the whole batch's re-chaining FIRST -- map.c:444-448's trigger evaluated per read, copies of the kept anchors
 * sorted as radix_sort_128x will sort them, one mm2gb_rmq_chain on the device beside the stream's host threads -- and a call is answered from
 * that batch when its input equals the stored input byte for byte (it is done on the spot otherwise).  Active when the host program imports
 * __wrap_mg_lchain_rmq (read from its ELF symbol table) and opt->max_chain_skip >= opt->rmq_size_cap (the device form is exhaustive);
 * MM2GB_PRECHAIN=0 / 1 overrides the first condition. ---- */
mm2gb_anchor_t *__wrap_mg_lchain_rmq(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc,
                                     float chn_pen_gap, float chn_pen_skip, int64_t n, mm2gb_anchor_t *a, int *n_u_, uint64_t **_u, void *km);
/* single-read re-chaining calls so far: all, those answered from a batch's answers made ahead, reads answered ahead in all */
void mm2gb_rechain_ahead_counts(int64_t *calls, int64_t *answered_ahead, int64_t *reads_ahead);
/* test hooks: offsets of the mirror's fields (order in csrc/rechain_ahead.cpp), ELF import probe, map.c:444-448's trigger */
int  mm2gb_mapopt_head_layout(int32_t *out, int max_out);
int  mm2gb_elf_imports_symbol(const char *path, const char *name);
int  mm2gb_rechain_wanted(const mm2gb_mapopt_head_t *opt, const struct mm2gb_chain_read_s *read);

/* ---- what the library imports from the host (resolved at link/load time; weak, so the library also loads alone) ----
 *   Misc  build_misc(const mm_idx_t*, const mm_mapopt_t*, const int64_t qlen_sum, const int n_seg);      map.c:393
 *   void  post_chaining_helper(const mm_idx_t*, const mm_mapopt_t*, chain_read_t*, Misc, void *km);      map.c:428
 *   void *kmalloc(void *km, size_t size);  void kfree(void *km, void *ptr);                              kalloc.h:15-18
 * mg_chain_backtrack / compact_a (lchain.c:27,78) are NOT imported: the library carries its own restatement
 * (mm2gb_backtrack_host) so the post-pass can run on several host threads. */

#ifdef __cplusplus
}
#endif
#endif
