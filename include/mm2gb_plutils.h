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
typedef struct {
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

/* The host's index and option records are opaque here; they are only handed back to the host callbacks below. */
struct mm_idx_s;
struct mm_mapopt_s;

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
