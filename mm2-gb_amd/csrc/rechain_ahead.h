// rechain_ahead.h -- internal: a batch's mg_lchain_rmq calls answered before the host asks for them (csrc/rechain_ahead.cpp).
#pragma once
#include <stdint.h>
#include <vector>
#include "../../include/mm2gb_plutils.h"
#include "host_chain.h"

namespace mm2gb {

// The answers of one batch of reads that is about to go through the host's post_chaining_helper (map.c:428-456).
// Read r of the batch was answered ahead iff slot_of_read[r] >= 0; its call's INPUT (the kept anchors in radix_sort_128x order, map.c:449)
// is sorted[off[s] .. off[s + 1]), its OUTPUT chains res.c.u / res.c.a at position s.  A call is answered from here only when its whole
// input equals the stored one byte for byte and its parameters equal prm: a wrong guess costs time, never a result.
struct RechainAhead {
	std::vector<int32_t> slot_of_read;
	std::vector<int64_t> off;
	BigBuf<mm2gb_anchor_t> sorted;
	mm2gb_rmq_param_t prm = {};
	ChainsOwner res;
	mm2gb_rmq_deal_t deal = {};
	double s_select = 0, s_sort = 0, s_call = 0;
	int64_t n_ahead() const { return off.empty() ? 0 : (int64_t)off.size() - 1; }
	void clear() { slot_of_read.clear(); off.clear(); mm2gb_chains_free(&res.c); }
};

// What map.c:444-449 looks at of a read that has been chained: its chains u[0..n_u) over the compacted anchors a[], n_seg, qlen_sum.
struct RechainRead { const mm2gb_anchor_t *a; const uint64_t *u; int n_u, n_seg, qlen_sum; };

// map.c:444-446: does post_chaining_helper re-chain this read?  (single-segment long reads whose best chain covers little of the read)
bool rechain_wanted(const mm2gb_mapopt_head_t &opt, const RechainRead &rd);

// True when the device form of the fill gives mg_lchain_rmq's answer for these options.  Round 6: always -- the reference's skip counter
// (lchain.c:329-333) can never pass a max_chain_skip that is at least the tree's size cap (the exhaustive walk, either kernel), and below
// the cap the one-anchor-per-step kernel keeps the counter -- unless MM2GB_RMQ_SKIP=ignore takes that walk away.
bool rechain_ahead_is_exact(const mm2gb_mapopt_head_t &opt);

// Decide, gather, sort as the host will (radix_sort_128x order, equal keys included), then ONE mm2gb_rmq_chain for the batch
// (device || n_threads host threads, tied reads redone with the reference's tree).  Returns 0 / -1 (mm2gb_last_error).
int rechain_ahead(mm2gb_engine_t *eng, const mm2gb_mapopt_head_t &opt, const mm2gb_misc_t &misc, const RechainRead *reads, int n_reads,
                  int n_threads, RechainAhead &out);

// Does the ELF file at `path` import (undefined dynamic symbol) `name`?  A host linked with -Wl,--wrap=mg_lchain_rmq imports
// __wrap_mg_lchain_rmq from this library; one linked without it never calls the library's re-chaining entry, and answering ahead would be wasted.
bool elf_imports_symbol(const char *path, const char *name);

} // namespace mm2gb
