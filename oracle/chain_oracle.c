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

/* ---- seed matches -> anchors: collect_seed_hits (map.c:295-331) with skip_seed (map.c:205-227) ------------------------------
 * One read.  seeds[k] is the first 16 bytes of the k-th mm_seed_t (mmpriv.h:40-46: n, q_pos, q_span:31|flt:1, seg_id:31|is_tandem:1),
 * its reference hits are hits[hit_off[k] .. hit_off[k+1]) (what mm_seed_t::cr points at: rid<<32 | pos<<1 | strand, index.c).
 * Name comparisons (strcmp(qname, s->name), map.c:211) are given as ranks in a common order: equal names <=> equal ranks,
 * qname > name <=> q_rank > ref_rank[rid].  Returns the number of anchors written to out[] (capacity: all hits), sorted as
 * radix_sort_128x leaves them (map.c:329). */
int64_t orc_collect_seeds(int64_t flag, int32_t qlen, int32_t q_rank, int64_t n_seeds, const orc_seed_t *seeds, const int64_t *hit_off,
                          const uint64_t *hits, const int32_t *ref_len, const int32_t *ref_rank, orc_anchor_t *out)
{
	const int64_t F_NO_DIAG = 0x001, F_NO_DUAL = 0x002, F_FOR_ONLY = 0x100000, F_REV_ONLY = 0x200000, F_QSTRAND = 0x100000000LL; /* minimap.h:8-9,28-29,40 */
	int64_t n_a = 0, k, h;
	for (k = 0; k < n_seeds; ++k) {
		const orc_seed_t *q = &seeds[k];
		const uint32_t q_span = q->span_flt & 0x7fffffffu, seg_id = q->seg_tandem & 0x7fffffffu, is_tandem = q->seg_tandem >> 31;
		for (h = hit_off[k]; h < hit_off[k + 1]; ++h) {
			const uint64_t r = hits[h];
			const int32_t rpos = (int32_t)((uint32_t)r >> 1);                       /* map.c:307 */
			const int same_strand = (r & 1) == (q->q_pos & 1);
			int is_self = 0, skip = 0;
			if (ref_rank && (flag & (F_NO_DIAG | F_NO_DUAL))) {                      /* map.c:208-219 (qname != NULL) */
				const int32_t rid = (int32_t)(r >> 32);
				const int cmp = q_rank < ref_rank[rid] ? -1 : q_rank > ref_rank[rid];
				if ((flag & F_NO_DIAG) && cmp == 0 && ref_len[rid] == qlen) {
					if ((uint32_t)r >> 1 == (q->q_pos >> 1)) skip = 1;
					else if (same_strand) is_self = 1;
				}
				if (!skip && (flag & F_NO_DUAL) && cmp > 0) skip = 1;
			}
			if (!skip && (flag & (F_FOR_ONLY | F_REV_ONLY))) {                      /* map.c:220-226 */
				if (same_strand) { if (flag & F_REV_ONLY) skip = 1; }
				else if (flag & F_FOR_ONLY) skip = 1;
			}
			if (skip) continue;
			orc_anchor_t *p = &out[n_a++];
			if (same_strand) {                                                       /* map.c:311-313 */
				p->x = (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
				p->y = (uint64_t)q_span << 32 | q->q_pos >> 1;
			} else if (!(flag & F_QSTRAND)) {                                        /* map.c:314-316 */
				p->x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
				p->y = (uint64_t)q_span << 32 | (uint32_t)(qlen - (int32_t)((q->q_pos >> 1) + 1 - q_span) - 1);
			} else {                                                                 /* map.c:317-321 */
				const int32_t len = ref_len[r >> 32];
				p->x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint32_t)(len - (rpos + 1 - (int32_t)q_span) - 1);
				p->y = (uint64_t)q_span << 32 | q->q_pos >> 1;
			}
			p->y |= (uint64_t)seg_id << 48;                                          /* map.c:322, MM_SEED_SEG_SHIFT */
			if (is_tandem) p->y |= 1ULL << 42;                                       /* MM_SEED_TANDEM */
			if (is_self) p->y |= 1ULL << 43;                                         /* MM_SEED_SELF */
		}
	}
	orc_radix_sort_x(out, out + n_a);                                                /* map.c:329 */
	return n_a;
}


/* ------------------------------------------------------------------------------------------------
 * chains -> hit records (hit.c:8-88)
 * ------------------------------------------------------------------------------------------------ */
static uint64_t orc_hash64(uint64_t key)   /* hit.c:40-50 */
{
	key = (~key + (key << 21));
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8));
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4));
	key = key ^ key >> 28;
	key = (key + (key << 31));
	return key;
}

void orc_gen_regs(uint32_t hash, int32_t qlen, int32_t n_u, const uint64_t *u, const orc_anchor_t *a, int is_qstrand, orc_reg_t *r)
{
	orc_anchor_t *z, tmp;
	int32_t i, k;
	if (n_u == 0) return;
	z = (orc_anchor_t*)malloc((size_t)n_u * sizeof(orc_anchor_t));
	for (i = k = 0; i < n_u; ++i) {                                              /* hit.c:63-69 */
		uint32_t h = (uint32_t)orc_hash64((orc_hash64(a[k].x) + orc_hash64(a[k].y)) ^ hash);
		z[i].x = u[i] ^ h;
		z[i].y = (uint64_t)k << 32 | (int32_t)u[i];
		k += (int32_t)u[i];
	}
	orc_radix_sort_x(z, z + n_u);
	for (i = 0; i < n_u >> 1; ++i) tmp = z[i], z[i] = z[n_u - 1 - i], z[n_u - 1 - i] = tmp;   /* larger score first */
	memset(r, 0, (size_t)n_u * sizeof(orc_reg_t));
	for (i = 0; i < n_u; ++i) {                                                  /* hit.c:75-86, 22-38, 8-20 */
		orc_reg_t *ri = &r[i];
		int32_t as = (int32_t)(z[i].y >> 32), cnt = (int32_t)z[i].y, q_span, j;
		ri->id = i; ri->parent = -1;
		ri->score = ri->score0 = (int32_t)(z[i].x >> 32);
		ri->hash = (uint32_t)z[i].x;
		ri->cnt = cnt; ri->as = as; ri->div = -1.0f;
		q_span = a_qspan(&a[as]);
		if (a[as].x >> 63) ri->flags |= 1u << 10;
		ri->rid = (int32_t)(a[as].x << 1 >> 33);
		ri->rs = (int32_t)a[as].x + 1 > q_span ? (int32_t)a[as].x + 1 - q_span : 0;
		ri->re = (int32_t)a[as + cnt - 1].x + 1;
		if (!(a[as].x >> 63) || is_qstrand) {
			ri->qs = (int32_t)a[as].y + 1 - q_span;
			ri->qe = (int32_t)a[as + cnt - 1].y + 1;
		} else {
			ri->qs = qlen - ((int32_t)a[as + cnt - 1].y + 1);
			ri->qe = qlen - ((int32_t)a[as].y + 1 - q_span);
		}
		ri->mlen = ri->blen = 0;
		if (cnt > 0) {
			ri->mlen = ri->blen = q_span;
			for (j = as + 1; j < as + cnt; ++j) {
				int span = a_qspan(&a[j]);
				int tl = (int32_t)a[j].x - (int32_t)a[j - 1].x, ql = (int32_t)a[j].y - (int32_t)a[j - 1].y;
				ri->blen += tl > ql ? tl : ql;
				ri->mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
			}
		}
	}
	free(z);
}

/* ------------------------------------------------------------------------------------------------
 * RMQ re-chaining (lchain.c:219-369)
 * ------------------------------------------------------------------------------------------------ */
int32_t orc_rmq_pair_score(const orc_anchor_t *ai, const orc_anchor_t *aj, float pen_gap, float pen_skip, int32_t *exact, int32_t *width)
{   /* comput_sc_simple, lchain.c:232-248 */
	int32_t dq = (int32_t)ai->y - (int32_t)aj->y, dr, dd, dg, q_span, sc;
	dr = (int32_t)(ai->x - aj->x);
	*width = dd = dr > dq ? dr - dq : dq - dr;
	dg = dr < dq ? dr : dq;
	q_span = a_qspan(aj);
	sc = q_span < dg ? q_span : dg;
	if (exact) *exact = (dd == 0 && dg <= q_span);
	if (dd || dq > q_span) {
		float lin_pen = pen_gap * (float)dd + pen_skip * (float)dg;
		float log_pen = dd >= 1 ? orc_log2_approx(dd + 1) : 0.0f;
		sc -= (int)(lin_pen + .5f * log_pen);
	}
	return sc;
}

typedef struct { int32_t y; int64_t j; } rmq_cand_t;
static int rmq_cand_desc(const void *pa, const void *pb)
{   /* descending (y, index): the order krmq_itr_prev walks the inner tree (lchain.c:227, 323-341) */
	const rmq_cand_t *a = (const rmq_cand_t*)pa, *b = (const rmq_cand_t*)pb;
	if (a->y != b->y) return a->y < b->y ? 1 : -1;
	return a->j < b->j ? 1 : a->j > b->j ? -1 : 0;
}

/* Does the pick among several holders of the smallest priority change anything for anchor i?  Each holder is followed through
 * lchain.c:316-341 as if it were the one the tree returned; without a skip limit the inner walk's result is the first candidate, in
 * walking order, to reach the walk's largest score -- whatever the walk started from below that -- so one walk (from q_span) serves
 * every holder.  With a skip limit the marks of lchain.c:333-338 depend on where the walk started: every tie is taken to decide. */
static int rmq_tie_decides(const orc_rmq_param_t *prm, const orc_anchor_t *a, const int32_t *f, int64_t i, int64_t st, int64_t st_inner, int64_t i0,
                           int32_t max_dist, int32_t max_dist_inner, double best_key, rmq_cand_t *cand)
{
	const int32_t yi = (int32_t)a[i].y, q_span = a_qspan(&a[i]);
	const int inner_there = max_dist_inner > 0 && st_inner < i0 && yi > 0;
	int32_t in_f = q_span, width, exact, sc, res_f = 0;
	int64_t in_j = -1, res_j = -1, j, nc = 0, k;
	int first = 1;
	/* (a limit at or above the tree's size cap can never end a walk: the counter grows by at most one per element visited, and the inner tree
	 * holds at most cap_rmq_size elements when it is walked, lchain.c:301-310) */
	if (prm->max_chn_skip != INT32_MAX && !(prm->cap_rmq_size > 0 && prm->max_chn_skip >= prm->cap_rmq_size)) return 1;
	if (inner_there) {
		for (j = st_inner; j < i0; ++j) {
			const int32_t yj = (int32_t)a[j].y;
			if (yj <= yi - 1 && yj >= yi - max_dist_inner) { cand[nc].y = yj; cand[nc].j = j; ++nc; }
		}
		qsort(cand, nc, sizeof(rmq_cand_t), rmq_cand_desc);
		for (k = 0; k < nc; ++k) {
			j = cand[k].j;
			sc = f[j] + orc_rmq_pair_score(&a[i], &a[j], prm->pen_gap, prm->pen_skip, 0, &width);
			if (width <= prm->bw && sc > in_f) in_f = sc, in_j = j;
		}
	}
	for (j = st; j < i0; ++j) {
		const int32_t yj = (int32_t)a[j].y;
		int32_t o_f = q_span;
		int64_t o_j = -1;
		if (!((yj > yi - max_dist && yj < yi) || (yj == yi && j == 0))) continue;
		if (f[j] + 0.5 * prm->pen_gap * ((int32_t)a[j].x + (int32_t)a[j].y) != best_key) continue;
		sc = f[j] + orc_rmq_pair_score(&a[i], &a[j], prm->pen_gap, prm->pen_skip, &exact, &width);
		if (width <= prm->bw && sc > o_f) o_f = sc, o_j = j;
		if (!exact && inner_there && in_f > o_f) o_f = in_f, o_j = in_j;
		if (first) res_f = o_f, res_j = o_j, first = 0;
		else if (o_f != res_f || o_j != res_j) return 1;
	}
	return 0;
}

static int64_t g_rmq_ties_that_decide;            /* of the last orc_rmq_fill (one thread at a time: test infrastructure) */
int64_t orc_rmq_last_ties_that_decide(void) { return g_rmq_ties_that_decide; }

int64_t orc_rmq_fill(const orc_rmq_param_t *prm, int64_t n, const orc_anchor_t *a, int32_t *f, int64_t *p, int64_t *n_tied)
{
	int32_t max_dist = prm->max_dist, max_dist_inner = prm->max_dist_inner;
	int64_t i, i0 = 0, st = 0, st_inner = 0, n_scored = 0, tied = 0, decide = 0;
	int32_t *t = (int32_t*)calloc(n > 0 ? n : 1, sizeof(int32_t));
	rmq_cand_t *cand = (rmq_cand_t*)malloc((n > 0 ? n : 1) * sizeof(rmq_cand_t));
	if (max_dist < prm->bw) max_dist = prm->bw;                                                   /* lchain.c:264 */
	if (max_dist_inner <= 0 || max_dist_inner >= max_dist) max_dist_inner = 0;                    /* lchain.c:265 */
	for (i = 0; i < n; ++i) {
		int64_t max_j = -1, j;
		int32_t q_span = a_qspan(&a[i]), max_f = q_span, yi = (int32_t)a[i].y;
		/* lchain.c:279-292: every anchor before the run of equal x that holds i is in the trees (until evicted) */
		if (i0 < i && a[i0].x != a[i].x) i0 = i;
		/* lchain.c:293-300 and 301-310: trees hold [st, i0) and [st_inner, i0); their sizes are i0 - st */
		while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + (uint64_t)max_dist || (i0 > st ? i0 - st : 0) > prm->cap_rmq_size)) ++st;
		if (max_dist_inner > 0)
			while (st_inner < i && (a[i].x >> 32 != a[st_inner].x >> 32 || a[i].x > a[st_inner].x + (uint64_t)max_dist_inner || (i0 > st_inner ? i0 - st_inner : 0) > prm->cap_rmq_size)) ++st_inner;
		/* lchain.c:311-315: closed interval [(yi - max_dist, INT32_MAX), (yi, 0)] in (y, index) order */
		{
			int64_t best_j = -1, n_best = 0;
			double best_key = 0.0;
			for (j = st; j < i0; ++j) {
				const int32_t yj = (int32_t)a[j].y;
				double key;
				if (!((yj > yi - max_dist && yj < yi) || (yj == yi && j == 0))) continue;
				key = f[j] + 0.5 * prm->pen_gap * ((int32_t)a[j].x + (int32_t)a[j].y);                /* -pri, lchain.c:284 */
				if (best_j < 0 || key > best_key) { best_key = key; best_j = j; n_best = 1; }
				else if (key == best_key) { ++n_best; if (yj > (int32_t)a[best_j].y || (yj == (int32_t)a[best_j].y && j > best_j)) best_j = j; }
			}
			if (best_j >= 0) {
				int32_t sc, exact, width, n_skip = 0;
				if (n_best > 1) { ++tied; decide += rmq_tie_decides(prm, a, f, i, st, st_inner, i0, max_dist, max_dist_inner, best_key, cand); }
				j = best_j;
				sc = f[j] + orc_rmq_pair_score(&a[i], &a[j], prm->pen_gap, prm->pen_skip, &exact, &width);
				++n_scored;
				if (width <= prm->bw && sc > max_f) max_f = sc, max_j = j;
				if (!exact && max_dist_inner > 0 && st_inner < i0 && yi > 0) {                          /* lchain.c:320: root_inner != 0 <=> the inner tree is not empty */
					int64_t nc = 0, k;
					for (j = st_inner; j < i0; ++j) {
						const int32_t yj = (int32_t)a[j].y;
						if (yj <= yi - 1 && yj >= yi - max_dist_inner) { cand[nc].y = yj; cand[nc].j = j; ++nc; }
					}
					qsort(cand, nc, sizeof(rmq_cand_t), rmq_cand_desc);
					for (k = 0; k < nc; ++k) {                                                              /* lchain.c:328-341 */
						j = cand[k].j;
						sc = f[j] + orc_rmq_pair_score(&a[i], &a[j], prm->pen_gap, prm->pen_skip, 0, &width);
						++n_scored;
						if (width <= prm->bw) {
							if (sc > max_f) {
								max_f = sc, max_j = j;
								if (n_skip > 0) --n_skip;
							} else if (t[j] == (int32_t)i) {
								if (++n_skip > prm->max_chn_skip) break;
							}
							if (p[j] >= 0) t[p[j]] = i;
						}
					}
				}
			}
		}
		f[i] = max_f, p[i] = max_j;                                                                  /* lchain.c:346 */
	}
	free(t); free(cand);
	if (n_tied) *n_tied = tied;
	g_rmq_ties_that_decide = decide;
	return n_scored;
}

orc_anchor_t *orc_lchain_rmq(const orc_rmq_param_t *prm, int64_t n, const orc_anchor_t *a, int32_t *n_u_, uint64_t **u_out,
                             int32_t *f_out, int64_t *p_out, int64_t *n_tied)
{
	int32_t *f, *v, n_u = 0, n_v = 0;
	int64_t *p;
	uint64_t *u;
	orc_anchor_t *out;
	*n_u_ = 0; *u_out = 0;
	if (n_tied) *n_tied = 0;
	if (n == 0 || a == 0) return 0;                                                                /* lchain.c:260-263 */
	f = (int32_t*)malloc(n * sizeof(int32_t)); p = (int64_t*)malloc(n * sizeof(int64_t)); v = (int32_t*)malloc(n * sizeof(int32_t));
	orc_rmq_fill(prm, n, a, f, p, n_tied);
	if (f_out) memcpy(f_out, f, n * sizeof(int32_t));
	if (p_out) memcpy(p_out, p, n * sizeof(int64_t));
	u = orc_backtrack(n, f, p, v, prm->min_cnt, prm->min_sc, prm->bw, &n_u, &n_v);                 /* max_drop = bw, lchain.c:253,355 */
	free(f); free(p);
	*n_u_ = n_u; *u_out = u;
	if (n_u == 0) { free(v); return 0; }
	out = orc_compact(n_u, u, n_v, v, a);
	free(v);
	return out;
}

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
