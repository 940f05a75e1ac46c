// rechain_ahead.cpp -- the re-chaining calls of a whole batch, answered on the device BEFORE the host's callback asks for them read by read.
//
// The reference host calls post_chaining_helper (map.c:428-456) for every read a boundary call hands back; for a long read whose best chain
// covers little of it (map.c:444-446: nearly every ONT read over a repeat-rich genome) that callback sorts the kept anchors by x
// (radix_sort_128x, map.c:449) and calls mg_lchain_rmq (lchain.c:250-369) -- one read at a time, on the calling thread.  A host linked with
// -Wl,--wrap=mg_lchain_rmq sends those calls to mm2gb_lchain_rmq, which could only answer them one at a time too: 154 thread-seconds of CPU
// code at 1.05 Gbp while the GPU idled (profiles/r04_bench_default.json).  Here the library evaluates map.c's trigger itself for every read of
// the batch, sorts copies of those reads' anchors exactly as the host is about to, runs ONE mm2gb_rmq_chain (device || host threads, tied
// reads redone with the reference's tree: exact for every read) and keeps the answers; mm2gb_lchain_rmq hands an answer out when the call's
// input equals the stored input byte for byte, and does the read itself otherwise.
#include <elf.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>
#include "engine.h"
#include "rechain_ahead.h"

namespace mm2gb {

namespace {
constexpr int64_t F_SPLICE = 0x080, F_NO_LJOIN = 0x400, F_SR = 0x1000;   // minimap.h:15,18,20

template <class F>
void for_reads(size_t n, int nt, F &&fn)
{
	std::atomic<size_t> next(0);
	auto work = [&]() { for (;;) { const size_t k = next.fetch_add(1); if (k >= n) break; fn(k); } };
	std::vector<std::thread> pool;
	for (int t = 1; t < std::max(1, std::min<int>(nt, (int)n)); ++t) pool.emplace_back(work);
	work();
	for (auto &th : pool) th.join();
}
double seconds_since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); }
} // namespace

bool rechain_wanted(const mm2gb_mapopt_head_t &opt, const RechainRead &rd)
{
	if (!(opt.bw_long > opt.bw && (opt.flag & (F_SPLICE | F_SR | F_NO_LJOIN)) == 0 && rd.n_seg == 1 && rd.n_u > 1)) return false;   // map.c:444-446
	if (!rd.a || !rd.u) return false;
	const int32_t st = (int32_t)rd.a[0].y, en = (int32_t)rd.a[(int32_t)rd.u[0] - 1].y;                                               // map.c:447
	const int qlen_sum = rd.qlen_sum;
	return qlen_sum - (en - st) > opt.rmq_rescue_size || (float)(en - st) > (float)qlen_sum * opt.rmq_rescue_ratio;                   // map.c:448
}

bool rechain_ahead_is_exact(const mm2gb_mapopt_head_t &opt)
{
	// lchain.c:329-333: n_skip grows by at most one per element of the inner tree visited, and that tree never holds more than cap_rmq_size
	// elements when it is queried (lchain.c:296-304): with max_chn_skip >= the cap the break cannot happen and the scan is exhaustive, which
	// is what the kernel computes.  (--max-chain-skip=infinity parses to 0 in the reference, SURVEY F2: not exact, answered per call by the host form.)
	// Round 6: below the cap the one-anchor-per-step kernel keeps the counter (its inner walk goes through the candidates in the reference's
	// order, post_kernels.hip k_rmq_fill), so a batch is answered ahead at any max_chain_skip; MM2GB_RMQ_SKIP=ignore takes that walk away again.
	static const bool no_limit_walk = [] { const char *v = getenv("MM2GB_RMQ_SKIP"); return v && !strcmp(v, "ignore"); }();
	return !no_limit_walk || (opt.rmq_size_cap > 0 && opt.max_chain_skip >= opt.rmq_size_cap);
}

int rechain_ahead(mm2gb_engine_t *eng, const mm2gb_mapopt_head_t &opt, const mm2gb_misc_t &misc, const RechainRead *reads, int n_reads,
                  int n_threads, RechainAhead &out)
{
	out.clear();
	out.slot_of_read.assign((size_t)std::max(0, n_reads), -1);
	const auto t0 = std::chrono::steady_clock::now();
	std::vector<int32_t> picked;
	out.off.assign(1, 0);
	for (int r = 0; r < n_reads; ++r) {
		if (!rechain_wanted(opt, reads[r])) continue;
		int64_t n_a = 0;
		for (int i = 0; i < reads[r].n_u; ++i) n_a += (int32_t)reads[r].u[i];      // map.c:449
		if (n_a <= 0) continue;
		out.slot_of_read[(size_t)r] = (int32_t)picked.size();
		picked.push_back(r);
		out.off.push_back(out.off.back() + n_a);
	}
	out.s_select = seconds_since(t0);
	if (picked.empty()) { out.off.clear(); return 0; }
	// what mg_lchain_rmq will be called with (map.c:450-451)
	out.prm = mm2gb_rmq_param_t{ opt.max_gap, opt.rmq_inner_dist, opt.bw_long, opt.max_chain_skip, opt.rmq_size_cap, opt.min_cnt, opt.min_chain_score,
	                             misc.chn_pen_gap, misc.chn_pen_skip };
	const auto t1 = std::chrono::steady_clock::now();
	try { out.sorted.resize((size_t)out.off.back()); } catch (const std::bad_alloc&) { out.clear(); return fail("rechain_ahead: out of host memory"); }
	for_reads(picked.size(), n_threads, [&](size_t s) {
		const RechainRead &rd = reads[picked[s]];
		mm2gb_anchor_t *dst = out.sorted.data() + out.off[s];
		const size_t n = (size_t)(out.off[s + 1] - out.off[s]);
		memcpy(dst, rd.a, n * sizeof(mm2gb_anchor_t));
		sort_by_x_like_host(dst, dst + n);                                       // the order the host's radix_sort_128x will leave (map.c:449)
	});
	out.s_sort = seconds_since(t1);
	const auto t2 = std::chrono::steady_clock::now();
	if (mm2gb_rmq_chain(eng, &out.prm, (int64_t)picked.size(), out.off.data(), out.sorted.data(), n_threads, &out.res.c, nullptr, &out.deal)) { out.clear(); return -1; }
	out.s_call = seconds_since(t2);
	return 0;
}

bool elf_imports_symbol(const char *path, const char *name)
{
	const int fd = open(path, O_RDONLY);
	if (fd < 0) return false;
	struct stat sb;
	if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < sizeof(Elf64_Ehdr)) { close(fd); return false; }
	const size_t size = (size_t)sb.st_size;
	void *map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (map == MAP_FAILED) return false;
	bool found = false;
	const unsigned char *base = (const unsigned char*)map;
	const Elf64_Ehdr *eh = (const Elf64_Ehdr*)base;
	auto inside = [&](uint64_t off, uint64_t len) { return off <= size && len <= size - off; };
	if (memcmp(eh->e_ident, ELFMAG, SELFMAG) == 0 && eh->e_ident[EI_CLASS] == ELFCLASS64 && eh->e_shentsize == sizeof(Elf64_Shdr) &&
	    inside(eh->e_shoff, (uint64_t)eh->e_shnum * sizeof(Elf64_Shdr))) {
		const Elf64_Shdr *sh = (const Elf64_Shdr*)(base + eh->e_shoff);
		for (unsigned s = 0; s < eh->e_shnum && !found; ++s) {
			if (sh[s].sh_type != SHT_DYNSYM || sh[s].sh_link >= eh->e_shnum || sh[s].sh_entsize != sizeof(Elf64_Sym)) continue;
			const Elf64_Shdr &str = sh[sh[s].sh_link];
			if (!inside(sh[s].sh_offset, sh[s].sh_size) || !inside(str.sh_offset, str.sh_size) || str.sh_size == 0) continue;
			const Elf64_Sym *sym = (const Elf64_Sym*)(base + sh[s].sh_offset);
			const char *names = (const char*)(base + str.sh_offset);
			if (names[str.sh_size - 1] != 0) continue;                          // a string table ends with NUL
			const size_t n_sym = sh[s].sh_size / sizeof(Elf64_Sym);
			for (size_t k = 0; k < n_sym; ++k)
				if (sym[k].st_shndx == SHN_UNDEF && sym[k].st_name < str.sh_size && strcmp(names + sym[k].st_name, name) == 0) { found = true; break; }
		}
	}
	munmap(map, size);
	return found;
}

} // namespace mm2gb

extern "C" {

// for tests: the offsets of the fields the library reads through its mirror of mm_mapopt_t (minimap.h:128-145), in the order
// flag, bw, bw_long, max_gap, max_chain_skip, min_cnt, min_chain_score, rmq_size_cap, rmq_inner_dist, rmq_rescue_size, rmq_rescue_ratio, sizeof
int mm2gb_mapopt_head_layout(int32_t *out, int max_out)
{
	const int32_t v[] = { (int32_t)offsetof(mm2gb_mapopt_head_t, flag), (int32_t)offsetof(mm2gb_mapopt_head_t, bw), (int32_t)offsetof(mm2gb_mapopt_head_t, bw_long),
	                      (int32_t)offsetof(mm2gb_mapopt_head_t, max_gap), (int32_t)offsetof(mm2gb_mapopt_head_t, max_chain_skip), (int32_t)offsetof(mm2gb_mapopt_head_t, min_cnt),
	                      (int32_t)offsetof(mm2gb_mapopt_head_t, min_chain_score), (int32_t)offsetof(mm2gb_mapopt_head_t, rmq_size_cap), (int32_t)offsetof(mm2gb_mapopt_head_t, rmq_inner_dist),
	                      (int32_t)offsetof(mm2gb_mapopt_head_t, rmq_rescue_size), (int32_t)offsetof(mm2gb_mapopt_head_t, rmq_rescue_ratio), (int32_t)sizeof(mm2gb_mapopt_head_t) };
	const int n = (int)(sizeof(v) / sizeof(v[0]));
	for (int i = 0; i < n && i < max_out; ++i) out[i] = v[i];
	return n;
}

int mm2gb_elf_imports_symbol(const char *path, const char *name) { return path && name && mm2gb::elf_imports_symbol(path, name) ? 1 : 0; }

int mm2gb_rechain_wanted(const mm2gb_mapopt_head_t *opt, const mm2gb_chain_read_t *read)
{
	if (!opt || !read) return 0;
	const mm2gb::RechainRead rd = { read->a, read->u, read->n_u, read->n_seg, read->seq.qlen_sum };
	return mm2gb::rechain_wanted(*opt, rd) ? 1 : 0;
}

} // extern "C"
