// engine.h -- internal: the chaining engine (device arenas, streams, micro-batch pipeline).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/mm2gb_chain.h"
#include "chain_dev.h"

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

// Page-locked host buffer that only ever grows.
struct PinnedBuf {
	void  *ptr = nullptr;
	size_t bytes = 0;
	int ensure(size_t need);
	void release();
};

struct Engine {
	int device = 0;
	mm2gb_config_t cfg;
	mm2gb_misc_t   misc;
	DevParams      params;
	LaunchCfg      launch;
	hipStream_t    stream = nullptr;       // compute (and, for now, copies)
	hipEvent_t     ev[6] = {};             // start, h2d done, prep done, score done, d2h done, spare
	int            n_cu = 256;

	// work arenas (sized by capacity_n / capacity_blocks)
	int64_t cap_n = 0, cap_reads = 0, cap_blocks = 0;
	DevBuf x, y, xhi, tag, st;
	DevBuf blk_firstcut, blk_pairs, blk_clamped;
	DevBuf chunk_start, chunk_end, chunk_cost, chunk_track, order, long_list;
	DevBuf chunk_pp, chunk_kk, chunk_blk, tile_sums, tile_base, bins;
	DevBuf counters, totals, flags, lut;
	// staging for the host-buffer API
	DevBuf raw, offsets, f, p;
	// pinned scalars for stats read-back
	int32_t *h_counters = nullptr;
	int64_t *h_totals = nullptr;

	mm2gb_stats_t last = {};
	bool stats_pending = false;
	bool timed_h2d = false, timed_d2h = false;
	bool misc_valid = false, coop_disabled = false;

	int  init(const mm2gb_config_t *cfg, const mm2gb_misc_t *misc, int device);
	void shutdown();
	int  set_misc(const mm2gb_misc_t *m);
	int  configure_score();
	int  reserve(int64_t n_anchors, int64_t n_reads, bool host_staging);
	// host buffers (pinned for true asynchrony): H2D, kernels, D2H enqueued; returns without waiting
	int  enqueue_host(int64_t n_reads, const int64_t *h_offsets, const mm2gb_anchor_t *h_anchors, int64_t n, int32_t *h_f, int32_t *h_p);
	int  enqueue(int64_t n_reads, const int64_t *d_offsets, const mm2gb_anchor_t *d_anchors, int64_t n, int32_t *d_f, int32_t *d_p);
	int  sync();
	int  collect_stats();
};

} // namespace mm2gb

struct mm2gb_engine { mm2gb::Engine e; };
