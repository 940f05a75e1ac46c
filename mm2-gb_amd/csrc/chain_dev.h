// chain_dev.h -- internal: device-side data layout and kernel launchers of the chaining engine (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mm2gb {

// Scoring constants as the kernels see them (clamps of lchain.c:160-161 already applied by the host).
struct DevParams {
	int   max_dist_x, max_dist_y, bw, max_iter, n_seg, is_cdna;
	int   dq_lim;            // min(max_dist_x, max_dist_y): the single-segment fast path's dq bound
	int   lut_last;          // last index of the penalty table: bw + 1, the "reject" entry (what a clamped index ends at)
	int   lut_base;          // LDS byte address of the table's entry 0: LUT_LDS_TOTAL - 4 * (lut_last + 1), so that it ends where LDS ends
	int   lut_clamp;         // 1: the sweeps clamp the table index to lut_last; 0: an index beyond it reads beyond LDS, i.e. 0 = reject
	int   free_sweep;        // 1 (only without lut_clamp): source blocks whose every pair has bw < dr <= dq_lim - bw are swept without range test
	int   edge_prefix;       // 1: edge blocks of tiles whose window starts rise from lane to lane take their window test from a scalar prefix mask (sweep_block_lut_edge_sorted); only with MM2GB_EDGE=new
	float gap, skip;
};

// bits of DevBatch::flags[0]
enum : unsigned { FLAG_ANY_SEGID = 1u,     // some anchor carries a segment id -> MODE_GENERAL
                  FLAG_NO_LUT = 2u };      // a query position >= 2^22 or a zero q_span -> the table sweep is not exact, use MODE_FAST

// MODE_LUT keeps its penalty table -- entries 0 .. bw and one rejecting entry -- at the very END of the workgroup's LDS, up to the last
// byte the hardware allocates (LUT_LDS_TOTAL is a multiple of every LDS granule in use: 512 and 1 280 bytes), so that an index beyond
// the table is an address beyond the allocation, which reads as 0 = "reject" and costs the LDS pipe no bank cycles
// (chain_kernels.hip, lut_address / sweep_block_lut2_free).  Where the table starts depends on bw (DevParams::lut_base).
constexpr int LUT_LDS_TOTAL = 31 * 2560;               // 79 360 bytes: two 1024-thread workgroups per CU (160 KB LDS)
constexpr int LUT_ENTRIES   = 5120;                    // the most the table may have (bw + 2 <= LUT_ENTRIES), what the global copy holds
constexpr int LUT_LDS_BASE  = LUT_LDS_TOTAL - LUT_ENTRIES * 4;   // 58 880: the lowest address the table can start at; ring, scratch and team records end below it
// a valid entry is LUT_BIAS - 128 * penalty, a rejecting one 0; sources are staged with their score term lowered by LUT_BIAS
constexpr int LUT_BIAS      = (1 << 30) + (1 << 16);

// Planner granularity: anchors per planning block (one k_window workgroup).
constexpr int PLAN_BLOCK = 1024;
#ifndef MM2GB_PLAN_THREADS
#define MM2GB_PLAN_THREADS 256
#endif
constexpr int PLAN_THREADS = MM2GB_PLAN_THREADS;   // k_window: each thread owns PLAN_BLOCK / PLAN_THREADS consecutive anchors
// cost charged per anchor on top of its pairs when ordering chunks (tile bookkeeping is not free)
constexpr int COST_PER_ANCHOR = 16;
// cost bins per work list of the planner (chain_kernels.hip: 16 per power of two; the engine sizes DevBatch::bins by it)
constexpr int PLAN_COST_BINS = 1024;

// A heavy chunk scored by its owner workgroup strip by strip (16 tiles = 1 024 anchors), the sweeps over the sources BEFORE the strip
// cut into items that any idle workgroup may take (chain_kernels.hip, split_chunk).  Every word is accessed with agent-scope atomics.
struct SplitSlot {
	unsigned long long word;   // total items of the open strip << 32 | next item to hand out (0: nothing open)
	int done;                  // items of the open strip that are finished
	int cs, ce, i_s;           // the chunk, the strip's first anchor
	int blocks_per_item;
	int jbs[8];                // per tile pair of the strip: first source block of its window
	int base[9];               // per tile pair: its first item; base[8] = total
	int pad_[8];
};
static_assert(sizeof(SplitSlot) == 128, "one slot per 128-byte line");
constexpr int SPLIT_MAX_ITEMS = 128;                   // per strip: 8 tile pairs x at most 16 items
constexpr int SPLIT_STRIP_TILES = 16;                  // = waves of a score workgroup

// A GANG: several workgroups on ONE chunk (k_score's first phase; chain_kernels.hip, gang_chunk_pairs).  A batch that cannot fill
// the machine ends with its largest chunks, and a team is bounded by one CU.  The chunk is cut into strips of GANG_STRIP_PAIRS tile pairs
// (one pair per wave of a workgroup); the workgroups the planner assigned to the chunk take strips from a counter and run the team
// pipeline across workgroups: final scores travel through global memory (agent-scope stores / loads), "tiles done" is a counter in
// the chunk's slot.  Nothing waits for a workgroup that may not be running: a strip is only ever waited for by workgroups that took a
// LATER strip, and it was taken by a workgroup that is executing.
struct GangSlot {
	int next_strip;            // strips handed out so far
	int done;                  // leading tiles of the chunk whose scores are final and visible
	int keep[6];               // the remembered anchor (rescue state) behind tile `done`: keep[0] = index + 1 (0: none)
	int chunk;                 // chunk id
	int first_wg, n_wg;        // workgroups [first_wg, first_wg + n_wg) start on this chunk
	int tiles_per_wave;        // 1: a wave takes one tile per turn (strips of 16 tiles); 2: a pair (strips of 32)
	int pad_[20];
};
static_assert(sizeof(GangSlot) == 128, "one slot per 128-byte line");
constexpr int GANG_MAX_CHUNKS = 64;                    // chunks per batch that may get a gang
constexpr int GANG_STRIP_PAIRS = 16;                   // turns per strip = waves of a score workgroup (a turn: one tile, or a pair)

// Everything one micro-batch needs in HBM: the caller's anchors (16 B each, all reads concatenated) and one array per derived field.
struct DevBatch {
	// inputs
	const uint4   *raw;        // mm128_t as 4 dwords: x.lo x.hi y.lo y.hi          16 B/anchor
	const int64_t *offsets;    // n_reads + 1
	int64_t        n;          // anchors
	int64_t        n_reads;
	// (reference position, query position, span and segment id are read straight from `raw`: x.lo, y.lo, y.hi -- chain_kernels.hip, a_x / a_y / a_tag)
	// range selection (written by k_window)
	int32_t  *st;              // first predecessor index of each anchor (lchain.c:172-173)   4 B
	// outputs
	int32_t  *f;               // score                                                4 B
	int32_t  *p;               // i - predecessor, 0 = none                            4 B
	// planner (per PLAN_BLOCK anchors)
	int32_t  *blk_firstcut;    // smallest i in block with st[i] == i, INT32_MAX if none
	int64_t  *blk_pairs;       // sum of window sizes in block
	int32_t  *blk_clamped;     // 1 if any window in block was cut by max_iter
	int32_t  *blk_read;        // read that owns the block's first anchor (k_block_reads)
	int32_t  *blk_wmax;        // two per block: widest window before the block's first cut, and from it on
	int64_t   n_blocks;
	// chunks (independent runs of anchors between cuts), at most n_blocks of them
	int32_t  *chunk_start, *chunk_end;
	int64_t  *chunk_cost;
	uint8_t  *chunk_track;     // needs the max_ii state machine
	int64_t  *chunk_pp;        // planner scratch: pairs before the chunk's first block
	int32_t  *chunk_kk, *chunk_blk; // planner scratch: clamped blocks before it; its first block
	int64_t  *tile_sums, *tile_base; // planner scratch: 3 values per tile of 1024 planning blocks
	int32_t  *bins;            // planner scratch: 3 x 256 cost bins
	int32_t  *order;           // chunk ids, most expensive first
	int32_t  *long_list;       // chunk ids scored by a whole workgroup
	int32_t  *mid_list;        // chunk ids scored by a 4-wave team
	// scalars
	int32_t  *counters;        // [0] n_chunks [1] work cursor (wave kernel) [2] n_long [3] work cursor (long kernel) [4] n_tracked [5] n_clamped_blocks
	int64_t  *totals;          // [0] total pairs [1] clamped windows [2] summed cost of the chunks on the big-team list
	unsigned *flags;           // FLAG_*
	const int32_t *lut;        // penalty table by dd, LUT_ENTRIES entries (MODE_LUT only)
	int64_t  *dbg;             // optional: 4 time stamps per score workgroup (MM2GB_DEBUG_PHASES), else null
	// one chunk on several workgroups (k_score's SPLIT build): a slot per score workgroup, zeroed before every launch, and room
	// for the partial results of one strip's items per workgroup; null when the build is not used
	SplitSlot          *split_slots;
	unsigned long long *split_part;
	GangSlot           *gang_slots;   // GANG_MAX_CHUNKS of them, set up by plan_gangs; null: no gangs
};
enum { CNT_NCHUNK = 0, CNT_CURSOR = 1, CNT_NLONG = 2, CNT_LCURSOR = 3, CNT_NTRACK = 4, CNT_NCLAMP = 5, CNT_NMID = 6, CNT_MCURSOR = 7,
       CNT_SPLIT_OPEN = 8,      // score workgroups that have started and not yet left the whole-workgroup phase
       CNT_NSPLIT = 9,          // chunks scored strip by strip with other workgroups' help
       CNT_HELPED = 10,         // items of such chunks that a workgroup other than the chunk's owner took
       CNT_SPLIT_ANY = 11,      // strips that are open for items right now (what idle waves poll)
       CNT_NGANG = 12,          // chunks scored by a gang of workgroups
       CNT_GANG_STRIPS = 13,    // strips of such chunks
       CNT_GANG_WGS = 14,       // workgroups that started in a gang
       CNT_WORDS = 16 };

struct LaunchCfg {
	int score_grid;          // persistent 1024-thread workgroups of k_score
	int host_mode;           // MODE_* the host's parameters allow (the device may still fall back to MODE_GENERAL)
	int ring_slots;          // LDS ring of the team modes: slots of 64 scores, shared out among the teams of a phase; 0 = team modes off
	int big_team;            // waves per team in the first phase: 16 (one team per workgroup) or 8 (two)
	int whole_wg_pct;        // a big-team chunk costing more than this % of a workgroup's fair share of that list gets all 16 waves; 0 = never
	int team4_all;           // 1: every heavy chunk of the table build goes to a 4-wave team, whatever its widest window (for A/B runs)
	int team4_share_pct;     // a wide-window heavy chunk goes to a 4-wave team unless it costs more than this % of a 4-wave team's fair share
	                         // of the batch's pairs (then: a big team); 0 = wide windows always go to big teams
	int gang_max;            // most workgroups the planner gives one chunk (0: no gangs)
	int gang_pct;            // a chunk gets gang_pct % of the workgroups its share of the batch's pairs would give it, if that is at least two
	int gang_pairs;          // 1: a gang's waves take pairs of tiles (as 4- and 8-wave teams do); 0: single tiles, the shorter chain per tile
	int split;               // 1: launch the SPLIT build (such chunks strip by strip, idle workgroups help); the host's choice by batch size
	int64_t long_min_cost;   // chunks at least this expensive ...
	int     long_min_window; // ... whose mean window is at least this are candidates for the cooperative mode
	int     wide_window;     // mean window from which a big team pays; narrower heavy chunks get 4-wave teams
};

void launch_window(const DevBatch &b, const DevParams &P, hipStream_t s);
void launch_plan(const DevBatch &b, const LaunchCfg &cfg, hipStream_t s);
void launch_score(const DevBatch &b, const DevParams &P, const LaunchCfg &cfg, hipStream_t s);
void launch_build_lut(int *d_lut, const DevParams &P, hipStream_t s);
size_t score_lds_bytes(const DevParams &P, int host_mode, int ring_slots);
int  score_set_lds_limit(size_t bytes);     // hipFuncSetAttribute on every k_score instance
bool score_has_split_build();               // compiled with -DMM2GB_WITH_SPLIT (make SPLIT=1)
enum { SCORE_MODE_LUT = 0, SCORE_MODE_FAST = 1, SCORE_MODE_GENERAL = 2 };

} // namespace mm2gb
