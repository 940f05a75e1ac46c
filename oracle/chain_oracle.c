/*
 * chain_oracle.c -- CPU oracle for the mm2-gb chaining hot path (see chain_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: a restatement, in plain C, of what the reference's CPU
 * chaining computes.  Every function names the reference lines it follows
 * (paths relative to the mm2-gb checkout).  Build with -ffp-contract=off and
 * without -ffast-math: the reference is built for baseline x86-64 (SSE2, no FMA,
 * Makefile:1), and the penalty arithmetic below must round the same way.
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include <pthread.h>
#include "chain_oracle.h"

#define ORC_REJECT INT32_MIN

/* field accessors for the packed anchor (lchain.c:140-143, mmpriv.h:23-24) */
static inline int32_t a_qspan(const orc_anchor_t *a) { return (int32_t)(a->y >> 32 & 0xff); }
static inline int32_t a_segid(const orc_anchor_t *a) { return (int32_t)((a->y >> 48) & 0xff); }
static inline uint64_t a_group(const orc_anchor_t *a) { return a->x >> 32; } /* strand|rid */

/* mmpriv.h:118-126: exponent from the float's bits, quadratic fit on the mantissa.
 * The subtraction is done in uint32 (as in the reference) and then converted. */
float orc_log2_approx(float x)
{
	union { float f; uint32_t u; } bits;
	uint32_t e;
	float r, m;
	bits.f = x;
	e = ((bits.u >> 23) & 255u) - 128u;
	r = (float)e;
	bits.u &= ~(255u << 23);
	bits.u += 127u << 23;
	m = bits.f;
	r += (-0.34484843f * m + 2.02466578f) * m - 0.67487759f;
	return r;
}

/* lchain.c:113-138 */
int32_t orc_pair_score(const orc_anchor_t *cur, const orc_anchor_t *prev, const orc_param_t *prm)
{
	const int32_t dq = (int32_t)cur->y - (int32_t)prev->y;
	const int same_seg = a_segid(cur) == a_segid(prev);
	int32_t dr, gap, diag, span, sc;

	if (dq <= 0 || dq > prm->max_dist_x) return ORC_REJECT;                 /* :118 */
	dr = (int32_t)(cur->x - prev->x);                                       /* :119 */
	if (same_seg && (dr == 0 || dq > prm->max_dist_y)) return ORC_REJECT;   /* :120 */
	gap = dr > dq ? dr - dq : dq - dr;                                      /* :121 dd */
	if (same_seg && gap > prm->bw) return ORC_REJECT;                       /* :122 */
	if (prm->n_seg > 1 && !prm->is_cdna && same_seg && dr > prm->max_dist_y) return ORC_REJECT; /* :123 */
	diag = dr < dq ? dr : dq;                                               /* :124 dg */
	span = a_qspan(prev);                                                   /* :125 */
	sc = span < diag ? span : diag;                                         /* :126 */
	if (gap != 0 || diag > span) {                                          /* :127 */
		const float lin = prm->pen_gap * (float)gap + prm->pen_skip * (float)diag;       /* :129 */
		const float lg = gap >= 1 ? orc_log2_approx((float)(gap + 1)) : 0.0f;            /* :130 */
		if (prm->is_cdna || !same_seg) {                                    /* :131 */
			if (!same_seg && dr == 0) ++sc;                                 /* :132 */
			else if (dr > dq || !same_seg) sc -= (int)(lin < lg ? lin : lg);/* :133 */
			else sc -= (int)(lin + .5f * lg);                               /* :134 */
		} else sc -= (int)(lin + .5f * lg);                                 /* :135 */
	}
	return sc;
}

/* lchain.c:155-207 */
static void chain_fill_ws(const orc_param_t *prm_in, int64_t n, const orc_anchor_t *a,
                          int32_t *f, int64_t *p, orc_stats_t *stats, int32_t *mark_ws)
{
	orc_param_t prm = *prm_in;
	orc_stats_t s;
	int32_t *mark;            /* the reference's t[] (lchain.c:166); caller-provided scratch of n ints, or NULL */
	int64_t i, lo = 0, keep = -1; /* lo = "st" (lchain.c:152), keep = "max_ii" */

	memset(&s, 0, sizeof(s));
	if (stats) *stats = s;
	if (n <= 0 || a == 0) return;
	if (prm.max_dist_x < prm.bw) prm.max_dist_x = prm.bw;                   /* :160 */
	if (prm.max_dist_y < prm.bw && !prm.is_cdna) prm.max_dist_y = prm.bw;   /* :161 */
	if (mark_ws) { mark = mark_ws; memset(mark, 0, (size_t)n * sizeof(int32_t)); }
	else mark = (int32_t*)calloc((size_t)n, sizeof(int32_t));

	for (i = 0; i < n; ++i) {
		int32_t best = a_qspan(&a[i]), n_skip = 0;                          /* :171 */
		int64_t arg = -1, j, stop;
		/* :172 window start: same strand|rid and within max_dist_x on the full 64-bit x */
		while (lo < i && (a_group(&a[i]) != a_group(&a[lo]) || a[i].x > a[lo].x + (uint64_t)(int64_t)prm.max_dist_x)) ++lo;
		if (i - lo > prm.max_iter) { lo = i - prm.max_iter; ++s.n_clamped; }/* :173 */
		s.n_pairs += i - lo;
		for (j = i - 1; j >= lo; --j) {                                     /* :174-188 */
			int32_t sc = orc_pair_score(&a[i], &a[j], &prm);
			++s.n_scored;
			if (sc == ORC_REJECT) continue;
			sc += f[j];
			if (sc > best) {
				best = sc, arg = j;
				if (n_skip > 0) --n_skip;
			} else if (mark[j] == (int32_t)i) {
				if (++n_skip > prm.max_skip) break;
			}
			if (p[j] >= 0) mark[p[j]] = (int32_t)i;
		}
		stop = j;                                                           /* :189 end_j */
		/* :190-195 refresh the remembered best anchor when it fell out of reach */
		if (keep < 0 || a[i].x - a[keep].x > (uint64_t)(int64_t)prm.max_dist_x) {
			int32_t top = INT32_MIN;
			keep = -1;
			++s.n_rescan;
			for (j = i - 1; j >= lo; --j)
				if (top < f[j]) top = f[j], keep = j;
		}
		/* :196-201 try it as one more predecessor when the scan did not reach it */
		if (keep >= 0 && keep < stop) {
			int32_t sc = orc_pair_score(&a[i], &a[keep], &prm);
			++s.n_rescue_eval;
			if (sc != ORC_REJECT && best < sc + f[keep]) {
				best = sc + f[keep], arg = keep;
				++s.n_rescue_taken;
			}
		}
		f[i] = best, p[i] = arg;                                            /* :202 */
		/* :204-205 */
		if (keep < 0 || (a[i].x - a[keep].x <= (uint64_t)(int64_t)prm.max_dist_x && f[keep] < f[i]))
			keep = i;
	}
	if (!mark_ws) free(mark);
	if (stats) *stats = s;
}

void orc_chain_fill(const orc_param_t *prm, int64_t n, const orc_anchor_t *a, int32_t *f, int64_t *p, orc_stats_t *stats)
{
	chain_fill_ws(prm, n, a, f, p, stats, 0);
}

/* ---- ksort.h:98-151, key = .x, 8 key bytes, 8 bits per pass, <=64 -> insertion sort ---- */

#define ORC_RS_SMALL 64

static void rs_insertion(orc_anchor_t *beg, orc_anchor_t *end) /* ksort.h:105-115 */
{
	orc_anchor_t *i;
	for (i = beg + 1; i < end; ++i) {
		if (i->x < (i - 1)->x) {
			orc_anchor_t *j, hold = *i;
			for (j = i; j > beg && hold.x < (j - 1)->x; --j) *j = *(j - 1);
			*j = hold;
		}
	}
}

typedef struct { orc_anchor_t *head, *tail; } rs_bin_t;

static void rs_pass(orc_anchor_t *beg, orc_anchor_t *end, int shift) /* ksort.h:116-146 */
{
	rs_bin_t bin[256], *k, *const bin_end = bin + 256;
	orc_anchor_t *it;
	for (k = bin; k != bin_end; ++k) k->head = k->tail = beg;
	for (it = beg; it != end; ++it) ++bin[it->x >> shift & 255].tail;          /* histogram */
	for (k = bin + 1; k != bin_end; ++k)                                         /* prefix -> [head,tail) */
		k->tail += (k - 1)->tail - beg, k->head = (k - 1)->tail;
	for (k = bin; k != bin_end;) {                                               /* in-place cycle permutation */
		if (k->head != k->tail) {
			rs_bin_t *dst = bin + (k->head->x >> shift & 255);
			if (dst != k) {
				orc_anchor_t carry = *k->head, moved;
				do {
					moved = carry; carry = *dst->head; *dst->head++ = moved;
					dst = bin + (carry.x >> shift & 255);
				} while (dst != k);
				*k->head++ = carry;
			} else ++k->head;
		} else ++k;
	}
	for (bin->head = beg, k = bin + 1; k != bin_end; ++k) k->head = (k - 1)->tail;
	if (shift) {
		shift = shift > 8 ? shift - 8 : 0;
		for (k = bin; k != bin_end; ++k) {
			if (k->tail - k->head > ORC_RS_SMALL) rs_pass(k->head, k->tail, shift);
			else if (k->tail - k->head > 1) rs_insertion(k->head, k->tail);
		}
	}
}

void orc_radix_sort_x(orc_anchor_t *beg, orc_anchor_t *end) /* ksort.h:147-151 */
{
	if (end - beg <= ORC_RS_SMALL) rs_insertion(beg, end);
	else rs_pass(beg, end, 56);
}

/* ---- lchain.c:9-25 ---- */
static int64_t bk_chain_end(int32_t max_drop, const orc_anchor_t *z, const int32_t *f, const int64_t *p, int32_t *t, int64_t k)
{
	int64_t i = (int64_t)z[k].y, last = -1, peak_i = i;
	int32_t peak = 0;
	if (i < 0 || t[i] != 0) return i;
	do {
		int32_t s;
		t[i] = 2;
		last = i = p[i];
		s = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
		if (s > peak) peak = s, peak_i = i;
		else if (peak - s > max_drop) break;
	} while (i >= 0 && t[i] == 0);
	for (i = (int64_t)z[k].y; i >= 0 && i != last; i = p[i]) t[i] = 0;
	return peak_i;
}

/* ---- lchain.c:27-76 ---- */
uint64_t *orc_backtrack(int64_t n, const int32_t *f, const int64_t *p, int32_t *v,
                        int32_t min_cnt, int32_t min_sc, int32_t max_drop,
                        int32_t *n_u_, int32_t *n_v_)
{
	orc_anchor_t *z;
	uint64_t *u = 0;
	int32_t *t;
	int64_t i, k, n_z = 0, n_v = 0;
	int32_t n_u = 0;
	int pass;

	*n_u_ = *n_v_ = 0;
	for (i = 0; i < n; ++i) if (f[i] >= min_sc) ++n_z;                        /* :35-36 */
	if (n_z == 0) return 0;
	z = (orc_anchor_t*)malloc((size_t)n_z * sizeof(*z));
	for (i = 0, k = 0; i < n; ++i)                                            /* :39-40 */
		if (f[i] >= min_sc) z[k].x = (uint64_t)(int64_t)f[i], z[k++].y = (uint64_t)i;
	orc_radix_sort_x(z, z + n_z);                                             /* :41 */
	t = (int32_t*)malloc((size_t)n * sizeof(int32_t));

	/* the reference walks the identical loop twice: once to size u[] (:44-56), once to fill (:59-71) */
	for (pass = 0; pass < 2; ++pass) {
		memset(t, 0, (size_t)n * sizeof(int32_t));
		n_v = 0, n_u = 0;
		for (k = n_z - 1; k >= 0; --k) {
			int64_t start = (int64_t)z[k].y, n_v0 = n_v, end_i;
			int32_t sc;
			if (t[start] != 0) continue;
			end_i = bk_chain_end(max_drop, z, f, p, t, k);
			for (i = start; i != end_i; i = p[i]) {
				if (pass) v[n_v] = (int32_t)i;
				++n_v, t[i] = 1;
			}
			sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
			if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt) {
				if (pass) u[n_u] = (uint64_t)sc << 32 | (uint64_t)(n_v - n_v0);
				++n_u;
			} else n_v = n_v0;
		}
		if (pass == 0) u = (uint64_t*)malloc((size_t)(n_u > 0 ? n_u : 1) * sizeof(uint64_t));
	}
	free(z); free(t);
	assert(n_v < INT32_MAX);
	*n_u_ = n_u, *n_v_ = (int32_t)n_v;
	if (n_u == 0) { free(u); return 0; }
	return u;
}

/* ---- lchain.c:78-111 ---- */
orc_anchor_t *orc_compact(int32_t n_u, uint64_t *u, int32_t n_v, const int32_t *v, const orc_anchor_t *a)
{
	orc_anchor_t *b, *w, *out;
	uint64_t *u2;
	int64_t i, j, k;

	b = (orc_anchor_t*)malloc((size_t)(n_v > 0 ? n_v : 1) * sizeof(*b));
	for (i = 0, k = 0; i < n_u; ++i) {                                        /* :86-90 chains end->start reversed */
		int32_t k0 = (int32_t)k, ni = (int32_t)u[i];
		for (j = 0; j < ni; ++j) b[k++] = a[v[k0 + (ni - j - 1)]];
	}
	w = (orc_anchor_t*)malloc((size_t)(n_u > 0 ? n_u : 1) * sizeof(*w));
	for (i = k = 0; i < n_u; ++i) {                                           /* :95-98 */
		w[i].x = b[k].x, w[i].y = (uint64_t)k << 32 | (uint64_t)i;
		k += (int32_t)u[i];
	}
	orc_radix_sort_x(w, w + n_u);                                             /* :99 */
	u2 = (uint64_t*)malloc((size_t)(n_u > 0 ? n_u : 1) * sizeof(uint64_t));
	out = (orc_anchor_t*)malloc((size_t)(n_v > 0 ? n_v : 1) * sizeof(*out));
	for (i = k = 0; i < n_u; ++i) {                                           /* :101-106 */
		int32_t src = (int32_t)w[i].y, cnt = (int32_t)u[src];
		u2[i] = u[src];
		memcpy(&out[k], &b[w[i].y >> 32], (size_t)cnt * sizeof(*out));
		k += cnt;
	}
	memcpy(u, u2, (size_t)n_u * 8);
	free(b); free(w); free(u2);
	return out;
}

/* ---- lchain.c:148-217 ---- */
orc_anchor_t *orc_lchain_dp(const orc_param_t *prm, int64_t n, const orc_anchor_t *a,
                            int32_t *n_u_, uint64_t **u_out,
                            int32_t *f_out, int64_t *p_out, orc_stats_t *stats)
{
	int32_t *f, *v, n_u = 0, n_v = 0, max_drop = prm->bw;
	int64_t *p;
	uint64_t *u;
	orc_anchor_t *res = 0;

	*n_u_ = 0; *u_out = 0;
	if (stats) memset(stats, 0, sizeof(*stats));
	if (n <= 0 || a == 0) return 0;                                            /* :156-159 */
	if (prm->is_cdna) max_drop = INT32_MAX;                                   /* :162 */
	f = (int32_t*)malloc((size_t)n * sizeof(int32_t));
	p = (int64_t*)malloc((size_t)n * sizeof(int64_t));
	v = (int32_t*)malloc((size_t)n * sizeof(int32_t));
	orc_chain_fill(prm, n, a, f, p, stats);
	if (f_out) memcpy(f_out, f, (size_t)n * sizeof(int32_t));
	if (p_out) memcpy(p_out, p, (size_t)n * sizeof(int64_t));
	u = orc_backtrack(n, f, p, v, prm->min_cnt, prm->min_sc, max_drop, &n_u, &n_v); /* :209 */
	if (n_u > 0) res = orc_compact(n_u, u, n_v, v, a);                       /* :216 */
	free(f); free(p); free(v);
	*n_u_ = n_u; *u_out = u;
	return res;
}

void orc_free(void *ptr) { free(ptr); }

/* ---- many reads on several host threads: the CPU baseline of bench.py (reads dealt dynamically, one call of
 *      orc_chain_fill per read, like kt_for over reads in map.c:1323) ---- */
typedef struct {
	const orc_param_t *prm; const int64_t *off; const orc_anchor_t *a; int32_t *f; int64_t *p;
	int64_t n_reads; volatile int64_t next; int64_t pairs; pthread_mutex_t mu;
} orc_mt_t;

static void *orc_mt_worker(void *arg)
{
	orc_mt_t *w = (orc_mt_t*)arg;
	int64_t mine = 0, cap = 0;
	int32_t *ws = 0;          /* per-thread scratch, like the reference's per-thread kalloc arena (no allocator contention) */
	for (;;) {
		int64_t r = __sync_fetch_and_add(&w->next, 1), n;
		orc_stats_t st;
		if (r >= w->n_reads) break;
		n = w->off[r + 1] - w->off[r];
		if (n > cap) { free(ws); cap = n + n / 4 + 1024; ws = (int32_t*)malloc((size_t)cap * sizeof(int32_t)); }
		chain_fill_ws(w->prm, n, w->a + w->off[r], w->f + w->off[r], w->p + w->off[r], &st, ws);
		mine += st.n_pairs;
	}
	free(ws);
	pthread_mutex_lock(&w->mu); w->pairs += mine; pthread_mutex_unlock(&w->mu);
	return 0;
}

int64_t orc_chain_fill_reads_mt(const orc_param_t *prm, int64_t n_reads, const int64_t *offsets, const orc_anchor_t *a,
                                int32_t *f, int64_t *p, int n_threads)
{
	orc_mt_t w;
	pthread_t *tid;
	int t;
	if (n_threads < 1) n_threads = 1;
	w.prm = prm; w.off = offsets; w.a = a; w.f = f; w.p = p; w.n_reads = n_reads; w.next = 0; w.pairs = 0;
	pthread_mutex_init(&w.mu, 0);
	tid = (pthread_t*)malloc((size_t)n_threads * sizeof(pthread_t));
	for (t = 0; t < n_threads; ++t) pthread_create(&tid[t], 0, orc_mt_worker, &w);
	for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
	free(tid);
	pthread_mutex_destroy(&w.mu);
	return w.pairs;
}
