/*
 * capture_hooks.c -- symbol-interposition hooks that record what the REFERENCE's chaining saw and produced.
 *
 * TEST INFRASTRUCTURE ONLY (used by oracle/gen_golden.py and tests/test_oracle_vs_ref.py).
 *
 * The reference is built unmodified as position-independent shared objects (oracle/Makefile), so its calls
 * to mg_lchain_dp (map.c:523 -> lchain.c:148) and mg_chain_backtrack (lchain.c:209 -> lchain.c:27) go through
 * the PLT.  Loading this object first (LD_PRELOAD for the minimap2_cpu binary, RTLD_GLOBAL from ctypes) lets
 * us see the per-anchor f[]/p[] arrays, which the reference never exposes, without touching its sources.
 *
 * Record layout appended to $MM2GB_CAPTURE (all little-endian, packed):
 *   char    magic[8] = "MMCAP01\0"
 *   int32   max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, is_cdna, n_seg
 *   float   pen_gap, pen_skip
 *   int32   have_fp
 *   int64   n
 *   u64     a_in[n][2]
 *   int32   f[n]      (present iff have_fp)
 *   int64   p[n]      (present iff have_fp)
 *   int32   n_u
 *   u64     u[n_u]
 *   int64   n_out
 *   u64     a_out[n_out][2]
 *
 * Seed matches (mm_collect_matches, seed.c:98, called by collect_seed_hits map.c:301) go to $MM2GB_CAPTURE_SEEDS:
 *   char    magic[8] = "MMSEED1\0"
 *   int32   qlen, n_m, rep_len, n_mini_pos
 *   u32     seed[n_m][4]        the leading 16 bytes of every mm_seed_t (mmpriv.h:40-46)
 *   u64     hits[sum of seed.n] the arrays mm_seed_t::cr points at, concatenated
 *   u64     mini_pos[n_mini_pos]
 *   -- followed, when the matches reach chaining, by the anchors collect_seed_hits made of them (map.c:329 -> :523):
 *   char    magic[8] = "MMANCH1\0"
 *   int64   n
 *   u64     a[n][2]
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t x, y; } cap128_t;

typedef cap128_t *(*lchain_dp_fn)(int, int, int, int, int, int, int, float, float, int, int, int64_t, cap128_t*, int*, uint64_t**, void*);
typedef uint64_t *(*backtrack_fn)(void*, int64_t, const int32_t*, const int64_t*, int32_t*, int32_t*, int32_t, int32_t, int32_t, int32_t*, int32_t*);

static __thread int      tl_seed_pending = 0;
static __thread int      tl_in_dp = 0;
static __thread int32_t *tl_f = 0;
static __thread int64_t *tl_p = 0;
static __thread int64_t  tl_n = 0;

static void *next_symbol(const char *name)
{
	void *fn = dlsym(RTLD_NEXT, name);
	if (!fn) { /* loaded from ctypes: the reference object is RTLD_LOCAL, reach it by path */
		const char *path = getenv("MM2GB_REF_LIB");
		void *h = path ? dlopen(path, RTLD_LAZY | RTLD_NOLOAD) : 0;
		if (h) fn = dlsym(h, name);
	}
	if (!fn) { fprintf(stderr, "[capture_hooks] cannot resolve the reference's %s\n", name); abort(); }
	return fn;
}

/* accessors for in-process users (ctypes) */
/* arm the f/p recorder for a direct call into the reference that reaches mg_chain_backtrack (mg_lchain_rmq, lchain.c:355) */
void cap_arm(int on) { tl_in_dp = on; if (on) tl_n = -1; }
int64_t cap_last_n(void) { return tl_n; }
const int32_t *cap_last_f(void) { return tl_f; }
const int64_t *cap_last_p(void) { return tl_p; }

uint64_t *mg_chain_backtrack(void *km, int64_t n, const int32_t *f, const int64_t *p, int32_t *v, int32_t *t,
                             int32_t min_cnt, int32_t min_sc, int32_t max_drop, int32_t *n_u_, int32_t *n_v_)
{
	static backtrack_fn real = 0;
	if (!real) real = (backtrack_fn)next_symbol("mg_chain_backtrack");
	if (tl_in_dp) {
		free(tl_f); free(tl_p);
		tl_f = (int32_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
		tl_p = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
		memcpy(tl_f, f, (size_t)n * sizeof(int32_t));
		memcpy(tl_p, p, (size_t)n * sizeof(int64_t));
		tl_n = n;
	}
	return real(km, n, f, p, v, t, min_cnt, min_sc, max_drop, n_u_, n_v_);
}

cap128_t *mg_lchain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc,
                       float chn_pen_gap, float chn_pen_skip, int is_cdna, int n_seg, int64_t n, cap128_t *a,
                       int *n_u_, uint64_t **_u, void *km)
{
	static lchain_dp_fn real = 0;
	const char *path = getenv("MM2GB_CAPTURE");
	cap128_t *in_copy = 0, *out;
	int64_t n_out = 0, i;
	if (!real) real = (lchain_dp_fn)next_symbol("mg_lchain_dp");
	if (tl_seed_pending) {                                  /* the anchors made of the matches recorded last */
		const char *spath = getenv("MM2GB_CAPTURE_SEEDS");
		FILE *sp = spath ? fopen(spath, "ab") : 0;
		tl_seed_pending = 0;
		if (sp) {
			fwrite("MMANCH1", 1, 8, sp);
			fwrite(&n, 8, 1, sp);
			if (n > 0) fwrite(a, sizeof(cap128_t), (size_t)n, sp);
			fclose(sp);
		}
	}
	if (path && n > 0 && a) {
		in_copy = (cap128_t*)malloc((size_t)n * sizeof(cap128_t));
		memcpy(in_copy, a, (size_t)n * sizeof(cap128_t));
	}
	tl_in_dp = 1; tl_n = -1;
	out = real(max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, chn_pen_gap, chn_pen_skip, is_cdna, n_seg, n, a, n_u_, _u, km);
	tl_in_dp = 0;
	if (path && in_copy) {
		FILE *fp = fopen(path, "ab");
		int32_t ints[9] = { max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, is_cdna, n_seg };
		float fl[2] = { chn_pen_gap, chn_pen_skip };
		int32_t have_fp = tl_n == n, n_u = *n_u_;
		if (!fp) { perror("[capture_hooks] MM2GB_CAPTURE"); abort(); }
		for (i = 0; i < n_u; ++i) n_out += (int32_t)(*_u)[i];
		fwrite("MMCAP01", 1, 8, fp);
		fwrite(ints, 4, 9, fp); fwrite(fl, 4, 2, fp);
		fwrite(&have_fp, 4, 1, fp); fwrite(&n, 8, 1, fp);
		fwrite(in_copy, sizeof(cap128_t), (size_t)n, fp);
		if (have_fp) { fwrite(tl_f, 4, (size_t)n, fp); fwrite(tl_p, 8, (size_t)n, fp); }
		fwrite(&n_u, 4, 1, fp);
		if (n_u > 0) fwrite(*_u, 8, (size_t)n_u, fp);
		fwrite(&n_out, 8, 1, fp);
		if (n_out > 0) fwrite(out, sizeof(cap128_t), (size_t)n_out, fp);
		fclose(fp);
	}
	free(in_copy);
	return out;
}

/* the leading fields of mm_seed_t (mmpriv.h:40-46), x86-64 layout: 16 bytes of counters and bit-fields, then the pointer */
typedef struct { uint32_t w[4]; const uint64_t *cr; } cap_seed_t;
typedef cap_seed_t *(*collect_matches_fn)(void*, int*, int, int, int, int, const void*, const void*, int64_t*, int*, int*, uint64_t**);

cap_seed_t *mm_collect_matches(void *km, int *n_m_, int qlen, int max_occ, int max_max_occ, int dist, const void *mi, const void *mv,
                               int64_t *n_a, int *rep_len, int *n_mini_pos, uint64_t **mini_pos)
{
	static collect_matches_fn real = 0;
	const char *path = getenv("MM2GB_CAPTURE_SEEDS");
	cap_seed_t *m;
	if (!real) real = (collect_matches_fn)next_symbol("mm_collect_matches");
	m = real(km, n_m_, qlen, max_occ, max_max_occ, dist, mi, mv, n_a, rep_len, n_mini_pos, mini_pos);
	if (path) {
		FILE *fp = fopen(path, "ab");
		int32_t hdr[4] = { qlen, *n_m_, *rep_len, *n_mini_pos }, i;
		if (!fp) { perror("[capture_hooks] MM2GB_CAPTURE_SEEDS"); abort(); }
		fwrite("MMSEED1", 1, 8, fp);
		fwrite(hdr, 4, 4, fp);
		for (i = 0; i < *n_m_; ++i) fwrite(m[i].w, 4, 4, fp);
		for (i = 0; i < *n_m_; ++i) if (m[i].w[0]) fwrite(m[i].cr, 8, m[i].w[0], fp);
		if (*n_mini_pos > 0) fwrite(*mini_pos, 8, (size_t)*n_mini_pos, fp);
		fclose(fp);
		tl_seed_pending = 1;
	}
	return m;
}
