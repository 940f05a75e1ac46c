// post_dev.h -- internal: device-side data of the kernels that follow the score kernel (post_kernels.hip): backtrack + compaction,
// and the score fill of RMQ re-chaining.  Kept apart from chain_dev.h, whose text identifies the score-kernel build a counter profile
// belongs to (bench.py: kernel_sha16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mm2gb {

// Everything the device post-pass (post_kernels.hip: backtrack + compaction, lchain.c:9-111) needs for one micro-batch.
// Chains of a read hold at least mc = max(1, min_cnt) anchors, so read r has at most n_r / mc of them: the per-chain arrays
// (u_tmp, heads) give read r the slots from offsets[r] / mc + r on.
constexpr int N_SIZE_CLASSES = 512;
struct PostBatch {
	const uint4   *raw;        // anchors (mm128_t)
	const int64_t *offsets;    // n_reads + 1
	int64_t        n, n_reads;
	const int32_t *f, *p;      // scores and relative predecessors, as k_score leaves them
	unsigned long long *z;     // n: candidates (f << 32 | i), sorted in place
	int2     *fp;              // n: (f, p | taken << 31) per anchor, written by k_post_lift: what a chain walk needs of an anchor in ONE 8-byte load; the top bit of .y is the host's t[] (lchain.c:43)
	int32_t  *picked;          // n: the host's v[] (lchain.c:65)
	unsigned char *sort_s;     // n + slack: the sort's destination bytes; sort_perm n: its permutation; sort_tmp n: the elements' way station (radix_pass_bytes; all null: the element form)
	int32_t  *sort_perm;
	unsigned long long *sort_tmp;
	int32_t  *up4, *up16;      // n each: distance to the anchor 4 / 16 predecessor links down the path (0 = path ends before)
	unsigned long long *u_tmp; // n / mc + n_reads: chains in the order they were found
	ulonglong2 *heads;         // n / mc + n_reads: (x of first anchor, offset << 32 | chain) for the order of compaction
	int32_t  *n_u, *n_kept;    // per read
	int64_t  *u_off, *a_off;   // n_reads + 1: exclusive scans of the two
	unsigned long long *u_out; // chains, read by read, in output order (lchain.c:145: score << 32 | count)
	uint4    *a_out;           // compacted anchors, read by read (lchain.c:78-111)
	int64_t  *totals;          // [0] chains [1] anchors kept
	int32_t  *cursor;          // work cursors: [0] reads of the sort [1] of the emit [2] of the classes [3] of the partition [4] walk tasks taken [5] walk tasks made [6] of the collection [8 + level] sort tasks of a level [16 + level] taken (32 words)
	int32_t  *order;           // n_reads: reads, largest first (the kernel ends with its longest read: start those first)
	int32_t  *size_bins;       // 2 x N_SIZE_CLASSES: reads per size class (eight classes per power of two), and the fill cursors of the scatter
	int       min_cnt, min_sc, max_drop;
	int       grid_waves;      // waves to launch (one read per wave at a time)
	int       team_reads;      // the largest reads of the batch that a whole workgroup starts on together (k_post_chains)
	long long *dbg;            // optional (MM2GB_DEBUG_PHASES): summed 100 MHz ticks of [0] candidate collection [1] sort [2] chain walks [3] emit
	long long *dbg_reads;      // optional (MM2GB_DEBUG_PHASES): per read 4 ticks: start, end of collection, end of sort, end of walks (k_post_chains)
	// The walks of a read shared out by TREE (round 6; null: one wave sorts and walks a read, k_post_chains).  p[] is a forest and a walk never
	// leaves its tree (lchain.c:9-25 follows p; marks are only ever set along it), so walks that start in different trees never meet: every tree
	// gets a class -- a hash of its root --, a read's sorted candidates are dealt to their classes in order, and (read, class) pairs are walked by
	// different waves.  What the host appends to u[] / v[] candidate by candidate is put together again from the candidates' sorted positions.
	unsigned char *cls;        // n: class of every anchor's tree
	int32_t  *cls_cnt;         // n_reads x N_TREE_CLASSES: anchors per class (a class's share of the read's slots in picked / zc / kpos; of u_tmp / u_loc: count / mc)
	int32_t  *cls_nz;          // n_reads x N_TREE_CLASSES: candidates per class
	int32_t  *read_nz;         // n_reads: candidates
	unsigned long long *zc;    // n: the sorted candidates class by class (aliases sort_tmp: the sort is over)
	int32_t  *kpos;            // n: their positions in the read's sorted order (aliases sort_perm)
	                           // (endslot: by sorted position, the chain slot of the chain that candidate ended, -1 none: n entries of 4 bytes in the read's own part of z, whose candidates live in zc by then)
	int32_t  *u_loc;           // per chain slot: where the chain's anchors start in the read's picked[]
	int32_t  *wtask, *wtask_order;   // up to n_reads x N_TREE_CLASSES each: read << 4 | class of every pair with candidates; the same, most candidates first
	// The sort by LEVELS (round 6): every run of more than 64 candidates that needs a radix pass is a task of its level's launch (k_post_sort_level)
	// -- the host's recursion bucket by bucket (ksort.h:140-145), breadth first over the whole batch -- instead of one wave sorting a read from top to bottom
	int4     *stask[2];        // two lists (a level reads one and fills the other): read, first candidate, length, key byte's shift; counts in cursor[8 + level], work cursors in cursor[16 + level]
	int32_t  *stask_order;     // the tasks of the level that runs, longest first
	int       walk_grid_waves; // waves of k_post_walk (it holds no LDS: more fit than of the sort)
	int       sort_pairs;      // 1: k_post_sort_level walks two short runs side by side (MM2GB_SORT_PAIRS=0: one at a time)
	long long *dbg_stasks;     // optional (MM2GB_DEBUG_PHASES): per sort task (up to 262144, in the order finished; count in dbg[40]) 4 values: start tick, end tick, level << 32 | length, steps of its pass
	long long *dbg_tasks;      // optional (MM2GB_DEBUG_PHASES): per walk task (in the order taken) 4 values: start tick, end tick, read << 4 | class, candidates
};
constexpr int N_TREE_CLASSES = 16;
// aux (may be null): a second stream for the pass that gives the anchors their classes, beside the lifting tables' (both only read the scores); fork / join: its events
void launch_post(const PostBatch &b, hipStream_t s, hipStream_t aux = nullptr, hipEvent_t fork = nullptr, hipEvent_t join = nullptr);

// RMQ re-chaining (mg_lchain_rmq, lchain.c:250-369) of reads whose anchors are already chained once: score fill on the device.
struct RmqParams { int max_dist, max_dist_inner, bw, cap_rmq_size; float pen_gap, pen_skip;
                   int weigh_ties;      // tile form: a tie on the smallest priority counts (n_tied) only if its holders differ in what they leave the anchor with; 0: every tie counts
                   int max_skip; };     // INT_MAX: the inner walk is exhaustive (both kernels); else lchain.c:329-333's limit, one-anchor-per-step kernel only (RmqBatch::rk_*)
struct RmqBatch {
	const uint4   *raw;        // anchors, sorted by x within each read
	const int64_t *offsets;
	int64_t        n, n_reads;
	int32_t *f, *p;            // out: score, i - predecessor (0 = none)
	double  *key;              // scratch, n: the (negated, order-preserving) priority of lchain.c:284 by RANK in the read's (y, index) order, once the anchor is in the tree
	ulonglong2 *by_y;          // scratch, n: (y << 32 | index, -) sorted: the (y, index) order of every read
	int32_t *ord_idx;          // scratch, n: index of the anchor at every rank
	int4    *meta;             // scratch, n: per anchor its rank, and the first and last rank of its range-minimum query
	uint4   *l1;               // scratch, n / 64 + n_reads + 1: per 64 ranks the largest key (low, high), its holder | several << 31, holder's rank & 63
	int32_t *bound;            // scratch, n / 64 + n_reads + 1: per 64 anchors (by index) the largest f + span
	int4    *win;              // tile form, scratch, n: per anchor its window start, inner window start, start of its run of equal x (null: the one-anchor-per-step kernel)
	uint4   *tree;             // tile form, scratch, 2 n: per read a binary tournament tree over its ranks (key low, key high, rank | several << 31, -)
	long long *dbg;            // optional (MM2GB_DEBUG_PHASES): summed over reads [0] steps [1] late entries [2] summaries rebuilt [3] ties looked at [4] summary loads [5] inner blocks read [6] winners read from memory [7] eviction tests (tile form: [0] anchors [1] tiles [2] ticks of tree updates [3] of queries [4] of broadcasts [5] anchors broadcast [6] inner blocks passed over [7] ticks of in-tile steps)
	int32_t *n_tied;           // out, per read: anchors whose range-minimum was shared by several elements (see post_kernels.hip)
	int32_t *cursor;           // two work cursors: reads taken by single waves, reads taken by teams
	int      grid_waves;
	// tile form, the inner window by strips of y (null: the inner window is swept block by block): every read's anchors sorted by (y >> strip_shift, index) --
	// a lane's inner candidates, y within max_dist_inner below its own, lie in at most two strips, and inside a strip its index window is one range
	unsigned long long *skey_in, *skey;   // scratch, n each: (strip << 32 | index), unsorted / sorted within each read; once sorted, skey_in's memory holds
	                                      // two int32 arrays of n: the scores in strip order, and every anchor's position in that order (k_rmq_fill_tiles: sf, spos)
	uint4   *sa;               // scratch, n: the anchors in that order (x, y, index, q_span)
	int4    *srange;           // scratch, n: per anchor [begin, end) in the lower strip and [begin, end) in the upper one (positions in its read's order)
	void    *sort_tmp;         // scratch of the segmented sort
	size_t   sort_tmp_bytes;
	int      strip_shift;
	long long *dbg_reads;      // optional (MM2GB_DEBUG_PHASES), tile form: per read 8 values: anchors, waves, ticks whole / tree update / queries / broadcasts / in-tile steps, anchors broadcast (wave 0 of a team)
	int      abandon_tied;     // tile form: a read is given up at the first tile that met a tie (its f / p are cleared: the post-pass finds nothing in it; n_tied != 0 tells the caller, who redoes the read anyway)
	int      n_team;           // tile form: the first n_team reads of the batch are filled by a whole workgroup each (k_rmq_fill_tiles)
	// one-anchor-per-step kernel with a skip limit (RmqParams::max_skip): the inner walk goes through the candidates in the reference's order,
	// (y, index) downwards, so every read's anchors and what the walk needs of them are kept BY RANK (null: no skip limit)
	uint4   *rk_a;             // scratch, n: (x, y, q_span, index) of the anchor at every rank
	int32_t *rk_f, *rk_p;      // scratch, n each: its score and the RANK of its predecessor (-1: none), written when the anchor is settled
	int32_t *rk_mark;          // scratch, n: lchain.c:333-338's t[], by rank: the anchor whose walk last offered this one's chain
	int2    *rk_in;            // scratch, n: per anchor (by index) the first and last rank of its inner walk: y in [y_i - max_dist_inner, y_i - 1]
};
int  launch_rmq_fill(const RmqBatch &b, const RmqParams &P, hipStream_t s);   // -1: a library sort refused (nothing usable was launched after it)
size_t rmq_strip_sort_temp_bytes(int64_t n, int64_t n_reads);   // what RmqBatch::sort_tmp must hold
int    rmq_strip_shift(const RmqParams &P);                      // 2^shift >= max_dist_inner (0: no inner window)

// Formats either side of the path: the seed sort upstream (radix_sort_128x of the collected anchors, map.c:329) and the
// conversion of chains into hit records downstream (mm_gen_regs, hit.c:52-88).
struct SortBatch {
	ulonglong2    *a;          // anchors (x, y), sorted in place by x within each read, exactly as radix_sort_128x leaves them
	const int64_t *offsets;
	int64_t        n_reads;
	int32_t       *cursor;
	int            grid_waves;
};
void launch_sort_x(const SortBatch &b, hipStream_t s);

// Seed matches -> anchors (collect_seed_hits, map.c:295-331, with skip_seed, map.c:205-227): every hit of every seed becomes an anchor
// unless the options drop it; a read's anchors keep the order (seed, hit) and are then sorted like the host's (launch_sort_x).
struct SeedRecord { uint32_t n, q_pos, span_flt, seg_tandem; };   // the leading 16 bytes of mm_seed_t (mmpriv.h:40-46)
struct SeedBatch {
	const SeedRecord *seeds;           // all reads' seeds, read by read
	const int64_t *seed_off;           // n_reads + 1
	const int64_t *hit_off;            // n_seeds + 1: seed k owns hits[hit_off[k] .. hit_off[k+1])
	const unsigned long long *hits;    // rid << 32 | pos << 1 | strand (what mm_seed_t::cr points at)
	const int32_t *qlen, *q_rank;      // per read (q_rank may be null: no name tests)
	const int32_t *ref_len, *ref_rank; // per reference sequence (may be null unless the options need them)
	int64_t        n_reads, n_seeds, n_hits;
	long long      flag;               // MM_F_NO_DIAG | MM_F_NO_DUAL | MM_F_FOR_ONLY | MM_F_REV_ONLY | MM_F_QSTRAND (minimap.h)
	int32_t       *seed_read;          // scratch, n_seeds: read of every seed
	ulonglong2    *tmp;                // scratch, n_hits: anchors at their hit's position (x = ~0: dropped)
	int32_t       *n_kept;             // scratch, n_reads
	int64_t       *anchor_off;         // out, n_reads + 1
	ulonglong2    *out;                // out, n_hits: the reads' anchors back to back, sorted
	int            grid_waves;
};
void launch_collect_seeds(const SeedBatch &b, hipStream_t s);

struct RegRecord {             // the leading 72 bytes of mm_reg1_t (minimap.h:104-119)
	int32_t id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, mlen, blen, n_sub, score0;
	uint32_t flags, hash;
	float div;
};
struct RegBatch {
	const int64_t *u_off, *a_off;      // n_reads + 1 each
	const unsigned long long *u;       // chains, read by read (score << 32 | count)
	const uint4   *a;                  // compacted anchors, read by read
	const int32_t *qlen;               // per read
	const uint32_t *hash;              // per read (map.c:590-592)
	int64_t        n_reads;
	ulonglong2    *z;                  // scratch: one per chain
	RegRecord     *regs;               // out: one per chain, at u_off
	int32_t       *cursor;
	int            is_qstrand, grid_waves;
};
void launch_gen_regs(const RegBatch &b, hipStream_t s);

} // namespace mm2gb
