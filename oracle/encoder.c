/* encoder.c — oracle restatement of CEncoder (src/colord/encoder.{h,cpp}), the edit-script helpers
 * (edit_script.h), the gap aligner's observable behaviour (libs/edlib/edlib.cpp as called from
 * edit_script.h:272-413) and the cost estimators (utils.h:700-1131).  TEST INFRASTRUCTURE ONLY.
 *
 * edlib is restated through its published semantics rather than its bit-vector machinery:
 *   - NW distance = D[n][m]; SHW distance = min_c D[n][c], end = first c attaining it (edlib.cpp:547-700);
 *   - the path is the traceback that prefers up (consume a query symbol), then left (consume a target
 *     symbol), then the diagonal (edlib.cpp:1021-1147); cells outside edlib's Ukkonen band can never satisfy
 *     those equalities, so the full matrix gives the same path;
 *   - alignments whose traceback state would reach 1 MiB are split on the target's middle column: the
 *     smallest query prefix 1..n-1 whose forward + backward scores add up to the optimum, then the empty
 *     prefix, then the whole query (edlib.cpp:1190-1215,1230-1400), and both halves recurse.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>

/* ---------------------------------------------------------------------------------------------------- */
/* small containers                                                                                     */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct { char* p; size_t n, cap; } str_t;
static void str_reserve(str_t* s, size_t need) { if (need > s->cap) { s->cap = need * 2 + 64; s->p = (char*)realloc(s->p, s->cap); } }
static void str_push(str_t* s, char c) { str_reserve(s, s->n + 1); s->p[s->n++] = c; }
static void str_append_n(str_t* s, size_t n, char c) { str_reserve(s, s->n + n); memset(s->p + s->n, c, n); s->n += n; }
static void str_append(str_t* s, const char* p, size_t n) { str_reserve(s, s->n + n); memcpy(s->p + s->n, p, n); s->n += n; }
static void str_free(str_t* s) { free(s->p); s->p = NULL; s->n = s->cap = 0; }

typedef struct { uint32_t len, pos_enc, pos_ref; } anchor_t;
typedef struct { int rev; uint32_t ref_id; anchor_t* a; uint32_t n, cap; uint32_t tot; } cand_t;
static void cand_push(cand_t* c, anchor_t a) { if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 16; c->a = (anchor_t*)realloc(c->a, c->cap * sizeof(anchor_t)); } c->a[c->n++] = a; }
static cand_t cand_clone(const cand_t* c) { cand_t r = *c; r.cap = c->n ? c->n : 1; r.a = (anchor_t*)malloc(r.cap * sizeof(anchor_t)); memcpy(r.a, c->a, c->n * sizeof(anchor_t)); return r; }

typedef struct { uint8_t* b; uint32_t len; } rr_t;

/* ---------------------------------------------------------------------------------------------------- */
/* cost estimators (utils.h:700-1131)                                                                   */
/* ---------------------------------------------------------------------------------------------------- */
static const char ES_SYM[11] = { 'A', 'C', 'D', 'G', 'M', 'T', 'X', 'Y', 'Z', 'S', 'R' };
static double entropy_dna(const uint8_t* s, size_t n)                             /* :706-727 */
{
	uint32_t h[4] = { 0, 0, 0, 0 };
	for (size_t i = 0; i < n; ++i) ++h[s[i]];
	double sum = 0; for (int i = 0; i < 4; ++i) sum += h[i];
	double rec = 1.0 / sum, e = 0;
	for (int c = 0; c < 4; ++c) if (h[c]) { double p = (double)h[c] * rec; e += log2(p) * p; }
	return -e;
}
static double entropy_es(const char* s, size_t n)                                 /* :730-752 */
{
	uint32_t h[128]; memset(h, 0, sizeof(h));
	for (size_t i = 0; i < n; ++i) ++h[(unsigned char)s[i]];
	double sum = 0; for (int i = 0; i < 11; ++i) sum += h[(int)ES_SYM[i]];
	double rec = 1.0 / sum, e = 0;
	for (int c = 0; c < 11; ++c) if (h[(int)ES_SYM[c]]) { double p = (double)h[(int)ES_SYM[c]] * rec; e += log2(p) * p; }
	return -e;
}

typedef struct {
	uint32_t dna[4], es[12], dec[2];
	double dna_logs[4], es_logs[12], dec_logs[2];
	uint32_t dna_sum, es_sum, dec_sum;
	uint32_t codes[128];
} estim_t;
static void est_rescale(uint32_t* a, int n, uint32_t* sum, uint32_t mx) { while (*sum > mx) { *sum = 0; for (int i = 0; i < n; ++i) { a[i] = (a[i] + 1) / 2; *sum += a[i]; } } }
static void est_logs(const uint32_t* st, double* lg, int n, uint32_t sum)
{
	double rec = 1.0 / sum;
	for (int i = 0; i < n; ++i) lg[i] = st[i] ? -log2((double)st[i] * rec) : 0.0;
}
static void est_reset(estim_t* e)                                                /* :877-889 */
{
	for (int i = 0; i < 4; ++i) e->dna[i] = 1;
	e->dna_sum = 4;
	for (int i = 0; i < 12; ++i) e->es[i] = 1;
	e->es_sum = 12;
	e->dec[0] = e->dec[1] = 1; e->dec_sum = 2;
	est_logs(e->dna, e->dna_logs, 4, e->dna_sum); est_logs(e->es, e->es_logs, 12, e->es_sum); est_logs(e->dec, e->dec_logs, 2, e->dec_sum);
}
static void est_init(estim_t* e)                                                 /* :914-930 */
{
	for (int i = 0; i < 128; ++i) e->codes[i] = 11;
	e->codes['A'] = 0; e->codes['C'] = 1; e->codes['G'] = 2; e->codes['T'] = 3; e->codes['D'] = 4; e->codes['M'] = 5;
	e->codes['X'] = 6; e->codes['Y'] = 7; e->codes['Z'] = 8; e->codes['S'] = 9; e->codes['R'] = 10;
	est_reset(e);
}
static void est_log_read(estim_t* e, const uint8_t* r, uint32_t len)             /* :946-955 */
{
	for (uint32_t i = 0; i < len; ++i) ++e->dna[r[i]];
	e->dna_sum += len;
	est_rescale(e->dna, 4, &e->dna_sum, 1u << 20);
	est_logs(e->dna, e->dna_logs, 4, e->dna_sum);
}
static uint64_t bitlen(uint64_t x) { uint64_t r = 0; for (; x; ++r) x >>= 1; return r; }
/* CEntropyEstimator::EncodeWithEditScript (:1060-1130) */
static int est_encode_with_es(estim_t* e, const char* es, size_t n_es, const uint8_t* plain, size_t n_plain, size_t ref_len)
{
	uint32_t loc[12], rd[12], pl[4] = { 0, 0, 0, 0 }, loc_sum = e->es_sum;
	memcpy(loc, e->es, sizeof(loc)); memset(rd, 0, sizeof(rd));
	double es_cost = e->dec_logs[0], plain_cost = e->dec_logs[1];
	uint32_t* lens = (uint32_t*)malloc((n_es + 1) * sizeof(uint32_t)); size_t n_lens = 0;
	char c = ' '; uint32_t len = 0;                                               /* analyze_es (:819-874) */
	for (size_t i = 0; i <= n_es; ++i)
	{
		char x = i < n_es ? es[i] : ' ';
		if (x == c) { ++len; continue; }
		if (c == 'D')
		{
			if (len >= 10) { ++loc[9]; ++loc_sum; ++rd[9]; lens[n_lens++] = len; }
			else { loc[4] += len; loc_sum += len; rd[4] += len; }
		}
		else if (c == 'M')
		{
			if (len >= 15) { ++loc[10]; ++loc_sum; ++rd[10]; lens[n_lens++] = len; }
			else { loc[5] += len; loc_sum += len; rd[5] += len; }
		}
		else if (c != ' ') { uint32_t k = e->codes[(unsigned char)c]; ++loc[k]; ++loc_sum; ++rd[k]; }
		c = x; len = 1;
	}
	for (size_t i = 0; i < n_plain; ++i) ++pl[plain[i]];
	est_logs(loc, e->es_logs, 12, loc_sum);
	for (int i = 0; i < 12; ++i) es_cost += rd[i] * e->es_logs[i];
	for (size_t i = 0; i < n_lens; ++i) es_cost += bitlen(lens[i]) + 1;
	for (int i = 0; i < 4; ++i) plain_cost += pl[i] * e->dna_logs[i];
	plain_cost += bitlen(ref_len) + 1;
	free(lens);
	int choose_plain = plain_cost < es_cost;
	if (choose_plain) { ++e->dec[1]; est_rescale(e->es, 12, &e->es_sum, 1u << 20); }
	else { ++e->dec[0]; memcpy(e->es, loc, sizeof(loc)); e->es_sum = loc_sum; est_rescale(e->es, 12, &e->es_sum, 1u << 20); }
	++e->dec_sum;
	est_rescale(e->dec, 2, &e->dec_sum, 1u << 20);
	est_logs(e->dec, e->dec_logs, 2, e->dec_sum);
	return !choose_plain;
}

/* ---------------------------------------------------------------------------------------------------- */
/* gap alignment                                                                                        */
/* ---------------------------------------------------------------------------------------------------- */
static char mismatch_sym(uint8_t ref, uint8_t nw)                                /* utils.h:341-352 */
{
	static const char mm[4][4] = { {'M','X','Y','Z'}, {'X','M','Y','Z'}, {'X','Y','M','Z'}, {'X','Y','Z','M'} };
	return mm[ref][nw];
}
static int is_mismatch(char c) { return c == 'X' || c == 'Y' || c == 'Z'; }

/* find_edit_dist (edit_script.h:156-239): plain DP, traceback prefers up ('D'), then left (insert), then diagonal */
static uint32_t dp_edit_script(const uint8_t* s1, uint32_t n1, const uint8_t* s2, uint32_t n2, str_t* out)
{
	uint32_t W = n2 + 1;
	uint32_t* c = (uint32_t*)malloc((size_t)(n1 + 1) * W * sizeof(uint32_t));
	for (uint32_t i = 0; i <= n1; ++i) c[(size_t)i * W] = i;
	for (uint32_t j = 0; j <= n2; ++j) c[j] = j;
	for (uint32_t i = 1; i <= n1; ++i)
		for (uint32_t j = 1; j <= n2; ++j)
		{
			uint32_t a = c[(size_t)i * W + j - 1] + 1, b = c[(size_t)(i - 1) * W + j] + 1, d = c[(size_t)(i - 1) * W + j - 1] + (s1[i - 1] != s2[j - 1]);
			uint32_t m = a < b ? a : b; c[(size_t)i * W + j] = m < d ? m : d;
		}
	str_t rev = { 0, 0, 0 };
	uint32_t i = n1, j = n2;
	while (i > 0 && j > 0)
	{
		uint32_t cur = c[(size_t)i * W + j];
		if (c[(size_t)(i - 1) * W + j] + 1 == cur) { str_push(&rev, 'D'); --i; }
		else if (c[(size_t)i * W + j - 1] + 1 == cur) { str_push(&rev, "ACGT"[s2[j - 1]]); --j; }
		else { str_push(&rev, s1[i - 1] == s2[j - 1] ? 'M' : mismatch_sym(s1[i - 1], s2[j - 1])); --i; --j; }
	}
	while (i > 0) { str_push(&rev, 'D'); --i; }
	while (j > 0) { str_push(&rev, "ACGT"[s2[j - 1]]); --j; }
	for (size_t k = rev.n; k-- > 0;) str_push(out, rev.p[k]);
	uint32_t dist = c[(size_t)n1 * W + n2];
	str_free(&rev); free(c);
	return dist;
}

/* last column of the NW matrix of q (rows) against t[0..m): col[i] = D[i][m] */
static void nw_last_column(const uint8_t* q, uint32_t n, const uint8_t* t, uint32_t m, int q_rev, int t_rev, uint32_t* col)
{
	for (uint32_t i = 0; i <= n; ++i) col[i] = i;
	for (uint32_t j = 1; j <= m; ++j)
	{
		uint8_t tc = t_rev ? *(t - (j - 1)) : t[j - 1];
		uint32_t diag = col[0]; col[0] = j;
		for (uint32_t i = 1; i <= n; ++i)
		{
			uint8_t qc = q_rev ? *(q - (i - 1)) : q[i - 1];
			uint32_t up = col[i - 1] + 1, left = col[i] + 1, d = diag + (qc != tc);
			diag = col[i];
			uint32_t v = up < left ? up : left; col[i] = v < d ? v : d;
		}
	}
}

/* edlib operations: 0 match, 1 consume query symbol (up), 2 consume target symbol (left), 3 mismatch */
typedef struct { uint8_t* p; size_t n, cap; } ops_t;
static void ops_push(ops_t* o, uint8_t v) { if (o->n == o->cap) { o->cap = o->cap ? 2 * o->cap : 256; o->p = (uint8_t*)realloc(o->p, o->cap); } o->p[o->n++] = v; }

static void nw_traceback(const uint8_t* q, uint32_t n, const uint8_t* t, uint32_t m, ops_t* out)
{
	uint32_t W = m + 1;
	uint32_t* D = (uint32_t*)malloc((size_t)(n + 1) * W * sizeof(uint32_t));
	for (uint32_t i = 0; i <= n; ++i) D[(size_t)i * W] = i;
	for (uint32_t j = 0; j <= m; ++j) D[j] = j;
	for (uint32_t i = 1; i <= n; ++i)
		for (uint32_t j = 1; j <= m; ++j)
		{
			uint32_t up = D[(size_t)(i - 1) * W + j] + 1, left = D[(size_t)i * W + j - 1] + 1, d = D[(size_t)(i - 1) * W + j - 1] + (q[i - 1] != t[j - 1]);
			uint32_t v = up < left ? up : left; D[(size_t)i * W + j] = v < d ? v : d;
		}
	ops_t rev = { 0, 0, 0 };
	uint32_t i = n, j = m;
	while (i > 0 && j > 0)                                                       /* edlib.cpp:1021-1147 */
	{
		uint32_t cur = D[(size_t)i * W + j];
		if (D[(size_t)(i - 1) * W + j] + 1 == cur) { ops_push(&rev, 1); --i; }
		else if (D[(size_t)i * W + j - 1] + 1 == cur) { ops_push(&rev, 2); --j; }
		else { ops_push(&rev, D[(size_t)(i - 1) * W + j - 1] == cur ? 0 : 3); --i; --j; }
	}
	while (i > 0) { ops_push(&rev, 1); --i; }
	while (j > 0) { ops_push(&rev, 2); --j; }
	for (size_t k = rev.n; k-- > 0;) ops_push(out, rev.p[k]);
	free(rev.p); free(D);
}

/* obtainAlignment (edlib.cpp:1164-1215) */
static void nw_path(const uint8_t* q, uint32_t n, const uint8_t* t, uint32_t m, uint32_t best, ops_t* out)
{
	if (n == 0 || m == 0) { for (uint32_t i = 0; i < n + m; ++i) ops_push(out, n == 0 ? 2 : 1); return; }
	long long blocks = (n + 63) / 64;
	long long sz = (2ll * 8 + 4) * blocks * m + 2ll * 4 * m;
	if (sz < 1024 * 1024) { nw_traceback(q, n, t, m, out); return; }
	/* Hirschberg on the target's middle column (edlib.cpp:1230-1400) */
	uint32_t L = m / 2, R = m - L;
	uint32_t* left = (uint32_t*)malloc((n + 1) * sizeof(uint32_t));
	uint32_t* right = (uint32_t*)malloc((n + 1) * sizeof(uint32_t));
	nw_last_column(q, n, t, L, 0, 0, left);                 /* left[i]  = D(q[0..i), t[0..L)) */
	nw_last_column(q + n - 1, n, t + m - 1, R, 1, 1, right); /* right[i] = D(reverse suffix of length i of q, reverse of t[L..m)) */
	int found = -1; uint32_t ls = 0, rs = 0;
	for (uint32_t i = 1; i + 1 <= n; ++i)                    /* queryIdx = i-1 in 0..n-2 */
		if (left[i] + right[n - i] == best) { found = (int)i; ls = left[i]; rs = right[n - i]; break; }
	if (found < 0 && L + right[n] == best) { found = 0; ls = L; rs = right[n]; }
	if (found < 0 && left[n] + R == best) { found = (int)n; ls = left[n]; rs = R; }
	free(left); free(right);
	if (found < 0) { fprintf(stderr, "oracle: hirschberg split not found\n"); abort(); }
	nw_path(q, (uint32_t)found, t, L, ls, out);
	nw_path(q + found, n - (uint32_t)found, t + L, R, rs, out);
}

typedef struct { str_t es; uint32_t dist; } edres_t;

/* find_edit_dist_with_edlib_ex (edit_script.h:272-336): inner gaps, query = ref, target = enc, NW */
static edres_t ed_inner(const uint8_t* ref, uint32_t nr, const uint8_t* enc, uint32_t ne)
{
	edres_t r; memset(&r, 0, sizeof(r));
	if (nr < 2 || ne < 2 || (nr < 15 && ne < 15)) { r.dist = dp_edit_script(ref, nr, enc, ne, &r.es); return r; }
	uint32_t* col = (uint32_t*)malloc((nr + 1) * sizeof(uint32_t));
	nw_last_column(ref, nr, enc, ne, 0, 0, col);
	uint32_t best = col[nr]; free(col);
	ops_t ops = { 0, 0, 0 };
	nw_path(ref, nr, enc, ne, best, &ops);
	uint32_t pr = 0, pe = 0;
	for (size_t i = 0; i < ops.n; ++i)
		switch (ops.p[i])
		{
		case 0: str_push(&r.es, 'M'); ++pr; ++pe; break;
		case 1: str_push(&r.es, 'D'); ++pr; break;                                 /* consumes the query = ref */
		case 2: str_push(&r.es, "ACGT"[enc[pe++]]); break;
		default: str_push(&r.es, mismatch_sym(ref[pr], enc[pe])); ++pr; ++pe;
		}
	free(ops.p);
	r.dist = best;
	return r;
}
/* find_edit_dist_with_edlib_ex_odwr (edit_script.h:341-400): flanks, query = enc, target = ref, SHW */
static edres_t ed_flank(const uint8_t* ref, uint32_t nr, const uint8_t* enc, uint32_t ne, uint32_t* ref_end)
{
	edres_t r; memset(&r, 0, sizeof(r));
	if (nr < 2 || ne < 2) { r.dist = dp_edit_script(ref, nr, enc, ne, &r.es); *ref_end = nr - 1; return r; }
	/* last row of D(enc rows, ref columns): score of end position c is D[ne][c+1]; first minimum.  edlib reads the
	 * scores off the bottom of the last 64-row block, W = 64*ceil(ne/64) - ne rows below the query end, where column
	 * c stands for end position c - W (edlib.cpp:666-681): when W > 0 the end position -1 (no reference symbol used,
	 * score ne) is a candidate too, and being the first it wins ties. */
	uint32_t* col = (uint32_t*)malloc((ne + 1) * sizeof(uint32_t));
	for (uint32_t i = 0; i <= ne; ++i) col[i] = i;
	uint32_t best = 0xffffffffu; int64_t end = 0;
	if (ne % 64 != 0) { best = ne; end = -1; }
	for (uint32_t j = 1; j <= nr; ++j)
	{
		uint32_t diag = col[0]; col[0] = j;
		for (uint32_t i = 1; i <= ne; ++i)
		{
			uint32_t up = col[i - 1] + 1, left = col[i] + 1, d = diag + (enc[i - 1] != ref[j - 1]);
			diag = col[i];
			uint32_t v = up < left ? up : left; col[i] = v < d ? v : d;
		}
		if (col[ne] < best) { best = col[ne]; end = (int64_t)j - 1; }
	}
	free(col);
	*ref_end = (uint32_t)end;                     /* -1 wraps exactly like the reference's uint32_t ref_end (edit_script.h:352) */
	ops_t ops = { 0, 0, 0 };
	nw_path(enc, ne, ref, (uint32_t)(end + 1), best, &ops);
	uint32_t pr = 0, pe = 0;
	for (size_t i = 0; i < ops.n; ++i)
		switch (ops.p[i])
		{
		case 0: str_push(&r.es, 'M'); ++pr; ++pe; break;
		case 1: str_push(&r.es, "ACGT"[enc[pe++]]); break;                         /* consumes the query = enc */
		case 2: str_push(&r.es, 'D'); ++pr; break;
		default: str_push(&r.es, mismatch_sym(ref[pr], enc[pe])); ++pr; ++pe;
		}
	free(ops.p);
	r.dist = best;
	return r;
}

/* refactor_edit_script (edit_script.h:416-446,591-671) */
static void fix_in_range(char* es, uint64_t start, uint64_t end)
{
	if (end < start + 2) return;
	--end;
	for (;;)
	{
		while (start < end && es[start] == 'M') ++start;
		while (start < end && es[end] != 'M') --end;
		if (start == end) break;
		char t = es[start]; es[start] = es[end]; es[end] = t;
	}
}
static void refactor_es(const uint8_t* ref, const uint8_t* enc, str_t* s)
{
	uint32_t st = 0, pos = 0, es_start = 0;
	for (uint32_t p = 0; p < s->n; ++p)                                           /* _dels */
	{
		char c = s->p[p];
		int mis = is_mismatch(c), ins = c == 'A' || c == 'C' || c == 'G' || c == 'T';
		if (ins || mis || ref[st] != ref[pos]) { fix_in_range(s->p, es_start, p); es_start = p; if (ins || mis) ++es_start; st = pos; }
		if (!ins) ++pos;
	}
	fix_in_range(s->p, es_start, s->n);
	st = 0; pos = 0; es_start = 0;
	for (uint32_t p = 0; p < s->n; ++p)                                           /* _ins */
	{
		char c = s->p[p];
		int mis = is_mismatch(c), del = c == 'D';
		if (del || mis || enc[st] != enc[pos]) { fix_in_range(s->p, es_start, p); es_start = p; if (del || mis) ++es_start; st = pos; }
		if (!del) ++pos;
	}
	fix_in_range(s->p, es_start, s->n);
}

/* ---------------------------------------------------------------------------------------------------- */
/* encoder object                                                                                        */
/* ---------------------------------------------------------------------------------------------------- */
struct orc_encoder {
	uint32_t m, k, modulo, min_part_alt, max_rec, min_anchors; int source;
	double frac_always, frac_min, max_matches_mult, cost_mult;
	rr_t* refs; size_t n_refs, cap_refs;
	estim_t est;
};
orc_encoder* orc_encoder_new(uint32_t anchor_len, uint32_t kmer_len, uint32_t modulo, int source, double frac_always, double frac_min,
                             double max_matches_mult, double cost_mult, uint32_t min_part_alt, uint32_t max_rec, uint32_t min_anchors)
{
	orc_encoder* e = (orc_encoder*)calloc(1, sizeof(*e));
	e->m = anchor_len; e->k = kmer_len; e->modulo = modulo; e->source = source; e->frac_always = frac_always; e->frac_min = frac_min;
	e->max_matches_mult = max_matches_mult; e->cost_mult = cost_mult; e->min_part_alt = min_part_alt; e->max_rec = max_rec; e->min_anchors = min_anchors;
	est_init(&e->est);
	return e;
}
void orc_encoder_free(orc_encoder* e) { if (!e) return; for (size_t i = 0; i < e->n_refs; ++i) free(e->refs[i].b); free(e->refs); free(e); }
void orc_encoder_add_ref(orc_encoder* e, const uint8_t* b, uint32_t len)
{
	if (e->n_refs == e->cap_refs) { e->cap_refs = e->cap_refs ? 2 * e->cap_refs : 64; e->refs = (rr_t*)realloc(e->refs, e->cap_refs * sizeof(rr_t)); }
	rr_t* r = &e->refs[e->n_refs++]; r->len = len; r->b = (uint8_t*)malloc(len + 1); memcpy(r->b, b, len); r->b[len] = 255;
}
void orc_encoder_new_pack(orc_encoder* e) { est_reset(&e->est); }                 /* encoder.cpp:1677 */
static uint8_t* ref_oriented(const orc_encoder* e, uint32_t id, int rev)           /* GetRefRead(id, rev): len+1 bytes incl. guard */
{
	const rr_t* r = &e->refs[id];
	uint8_t* o = (uint8_t*)malloc(r->len + 1);
	if (!rev) memcpy(o, r->b, r->len);
	else for (uint32_t i = 0; i < r->len; ++i) o[i] = (uint8_t)(3 - r->b[r->len - 1 - i]);
	o[r->len] = 255;
	return o;
}

/* ---- m-mer index of the read to encode -------------------------------------------------------------- */
typedef struct { uint64_t mmer; uint32_t pos; } mp_t;
static int cmp_mp(const void* a, const void* b)
{
	const mp_t* x = (const mp_t*)a; const mp_t* y = (const mp_t*)b;
	if (x->mmer != y->mmer) return x->mmer < y->mmer ? -1 : 1;
	return x->pos < y->pos ? -1 : x->pos > y->pos;
}
static int cmp_pos_mp(const void* a, const void* b) { const mp_t* x = (const mp_t*)a; const mp_t* y = (const mp_t*)b; return x->pos < y->pos ? -1 : x->pos > y->pos; }
typedef struct { mp_t* v; uint32_t n; uint64_t* uniq; uint32_t* first; uint32_t n_uniq; } mindex_t;
static void mindex_build(mindex_t* ix, const uint8_t* read, uint32_t len, uint32_t m)
{
	memset(ix, 0, sizeof(*ix));
	if (len < m) return;
	ix->n = len - m + 1; ix->v = (mp_t*)malloc(ix->n * sizeof(mp_t));
	uint64_t mask = (1ULL << (2 * m)) - 1, x = 0;
	for (uint32_t p = 0; p < len; ++p) { x = ((x << 2) + read[p]) & mask; if (p + 1 >= m) { ix->v[p + 1 - m].mmer = x; ix->v[p + 1 - m].pos = p + 1 - m; } }
	qsort(ix->v, ix->n, sizeof(mp_t), cmp_mp);
	ix->uniq = (uint64_t*)malloc(ix->n * 8); ix->first = (uint32_t*)malloc((ix->n + 1) * 4);
	for (uint32_t i = 0; i < ix->n; ++i) if (i == 0 || ix->v[i].mmer != ix->v[i - 1].mmer) { ix->uniq[ix->n_uniq] = ix->v[i].mmer; ix->first[ix->n_uniq++] = i; }
	ix->first[ix->n_uniq] = ix->n;
}
static void mindex_free(mindex_t* ix) { free(ix->v); free(ix->uniq); free(ix->first); }
static int mindex_find(const mindex_t* ix, uint64_t x)
{
	uint32_t lo = 0, hi = ix->n_uniq;
	while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (ix->uniq[mid] < x) lo = mid + 1; else hi = mid; }
	return (lo < ix->n_uniq && ix->uniq[lo] == x) ? (int)lo : -1;
}

/* LIS (utils.cpp:114-209): strictly increasing, patience with predecessor links */
static int lis_search(const int* tmp_first, int size, int value)                   /* fast_upper_bound (:114-155) */
{
	int low = 0;
	while (size > 0)
	{
		int half = size / 2, other_half = size - half, probe = low + half, other_low = low + other_half;
		int v = tmp_first[probe];
		size = half;
		low = value > v ? other_low : low;
	}
	return low;
}
static uint32_t lis(const int* in, uint32_t n, int* out)
{
	if (!n) return 0;
	int* pred = (int*)malloc(n * sizeof(int)); int* tf = (int*)malloc(n * sizeof(int)); int* ts = (int*)malloc(n * sizeof(int));
	for (uint32_t i = 0; i < n; ++i) pred[i] = -1;
	tf[0] = in[0]; ts[0] = 0; int out_len = 1;
	for (int i = 1; i < (int)n; ++i)
	{
		int x = in[i], pos;
		if (tf[out_len - 1] < x) pos = out_len; else pos = lis_search(tf, out_len, x);
		if (pos == out_len) { tf[out_len] = x; ts[out_len] = i; ++out_len; } else { tf[pos] = x; ts[pos] = i; }
		pred[i] = pos > 0 ? ts[pos - 1] : -1;
	}
	int cur = ts[out_len - 1];
	for (int i = out_len - 1; i >= 0; --i) { out[i] = in[cur]; cur = pred[cur]; }
	free(pred); free(tf); free(ts);
	return (uint32_t)out_len;
}

enum { AR_EMPTY, AR_TOO_MANY, AR_TOO_LOW, AR_ACCEPT };
/* AnalyseRefRead (encoder.cpp:1016-1056) */
static int analyse_ref(const orc_encoder* e, const mindex_t* ix, uint32_t enc_len, const uint8_t* ref, uint32_t ref_len, cand_t* c, int decision)
{
	const uint32_t m = e->m;
	if (ref_len < m || ix->n == 0) return AR_EMPTY;
	/* reference positions whose m-mer occurs in the read to encode */
	uint32_t nrp = 0; mp_t* rp = (mp_t*)malloc((ref_len - m + 1) * sizeof(mp_t));
	uint64_t mask = (1ULL << (2 * m)) - 1, x = 0;
	for (uint32_t p = 0; p < ref_len; ++p)
	{
		x = ((x << 2) + ref[p]) & mask;
		if (p + 1 >= m && mindex_find(ix, x) >= 0) { rp[nrp].mmer = x; rp[nrp].pos = p + 1 - m; ++nrp; }
	}
	if (!nrp) { free(rp); return AR_EMPTY; }
	mp_t* rs = (mp_t*)malloc(nrp * sizeof(mp_t)); memcpy(rs, rp, nrp * sizeof(mp_t));
	qsort(rs, nrp, sizeof(mp_t), cmp_mp);                                          /* by m-mer, positions ascending (:1024-1028) */
	/* shared m-mers: enc positions in sorted-by-position order */
	uint32_t n_enc = 0; uint64_t matches = 0;
	for (uint32_t i = 0; i < nrp;)
	{
		uint32_t j = i; while (j < nrp && rs[j].mmer == rs[i].mmer) ++j;
		int u = mindex_find(ix, rs[i].mmer);
		uint32_t ce = ix->first[u + 1] - ix->first[u];
		n_enc += ce; matches += (uint64_t)ce * (j - i);
		i = j;
	}
	if (decision != 0) decision = (double)matches > e->max_matches_mult * (double)(enc_len + 1);   /* enc_read.size() counts the guard (:1037) */
	if (decision == 1) { free(rp); free(rs); return AR_TOO_MANY; }
	mp_t* se = (mp_t*)malloc(n_enc * sizeof(mp_t)); uint32_t k = 0;
	for (uint32_t i = 0; i < nrp;)
	{
		uint32_t j = i; while (j < nrp && rs[j].mmer == rs[i].mmer) ++j;
		int u = mindex_find(ix, rs[i].mmer);
		for (uint32_t t = ix->first[u]; t < ix->first[u + 1]; ++t) se[k++] = ix->v[t];
		i = j;
	}
	qsort(se, n_enc, sizeof(mp_t), cmp_pos_mp);                                    /* Convert (:697-729) */
	/* LIS input (:617-642): per enc occurrence, the ref positions of its m-mer in descending order */
	int* lin = (int*)malloc((size_t)(matches ? matches : 1) * sizeof(int)); size_t nl = 0;
	for (uint32_t i = 0; i < n_enc; ++i)
	{
		uint32_t lo = 0, hi = nrp;
		while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (rs[mid].mmer < se[i].mmer) lo = mid + 1; else hi = mid; }
		uint32_t j = lo; while (j < nrp && rs[j].mmer == se[i].mmer) ++j;
		for (uint32_t t = j; t-- > lo;) lin[nl++] = (int)rs[t].pos;
	}
	int* lout = (int*)malloc((nl ? nl : 1) * sizeof(int));
	uint32_t n_lis = lis(lin, (uint32_t)nl, lout);
	/* map the chain back to (enc pos, ref pos) pairs (:644-658) and merge runs into anchors (:731-776) */
	c->n = 0; c->tot = 0;
	uint32_t ep = 0, rpp = 0, start_e = 0, start_r = 0, prev_e = 0, prev_r = 0, run = 0;
	for (uint32_t i = 0; i < n_lis; ++i)
	{
		uint32_t pr = (uint32_t)lout[i];
		while (rp[rpp++].pos != pr) ;
		uint64_t mm = rp[rpp - 1].mmer;
		while (se[ep++].mmer != mm) ;
		uint32_t pe = se[ep - 1].pos;
		if (run && prev_e == pe - 1 && prev_r == pr - 1) ++run;
		else
		{
			if (run) { anchor_t a = { run + m - 1, start_e, start_r }; cand_push(c, a); c->tot += a.len; }
			run = 1; start_e = pe; start_r = pr;
		}
		prev_e = pe; prev_r = pr;
	}
	if (run) { anchor_t a = { run + m - 1, start_e, start_r }; cand_push(c, a); c->tot += a.len; }
	free(rp); free(rs); free(se); free(lin); free(lout);
	if (c->n < e->min_anchors) return AR_TOO_LOW;
	return AR_ACCEPT;
}

/* ---- HiFi k-mer anchors (encoder.cpp:870-1013) -------------------------------------------------------- */
static uint64_t revcomp_k(uint64_t x, uint32_t k) { uint64_t r = 0; for (uint32_t i = 0; i < k; ++i) { r = (r << 2) + (3 - (x & 3)); x >>= 2; } return r; }
/* position of the unique occurrence of forward k-mer x in read, -1 if absent or repeated.  restrict != NULL: only
 * positions whose canonical k-mer is in that ascending list and passes the modulo test take part (CKmersHashMapLP, :557-597) */
static int64_t unique_pos(const uint8_t* read, uint32_t len, uint32_t k, uint64_t x, const uint64_t* restrict_, uint32_t n_restrict, uint32_t modulo)
{
	if (len < k) return -1;
	uint64_t mask = (1ULL << (2 * k)) - 1, f = 0, r = 0; int64_t found = -1;
	for (uint32_t p = 0; p < len; ++p)
	{
		f = ((f << 2) + read[p]) & mask; r = (r >> 2) + ((uint64_t)(3 - read[p]) << (2 * (k - 1)));
		if (p + 1 < k || f != x) continue;
		if (restrict_)
		{
			uint64_t can = f < r ? f : r;
			if (orc_hash_mm(can) % modulo) continue;
			uint32_t lo = 0, hi = n_restrict; while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (restrict_[mid] < can) lo = mid + 1; else hi = mid; }
			if (lo >= n_restrict || restrict_[lo] != can) continue;
		}
		if (found >= 0) return -1;
		found = p + 1 - k;
	}
	return found;
}
static int cmp_u64(const void* a, const void* b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }
static int cmp_anchor_enc(const void* a, const void* b) { const anchor_t* x = (const anchor_t*)a; const anchor_t* y = (const anchor_t*)b; return x->pos_enc < y->pos_enc ? -1 : x->pos_enc > y->pos_enc; }
static void anchors_erase(cand_t* c, uint32_t i) { memmove(c->a + i, c->a + i + 1, (c->n - i - 1) * sizeof(anchor_t)); --c->n; }
enum { KR_NO_ANCHORS, KR_INCOMPATIBLE, KR_ACCEPT };
static int analyse_ref_kmers(const orc_encoder* e, const uint8_t* enc, uint32_t enc_len, const uint8_t* ref, uint32_t ref_len,
                             const uint64_t* common_sorted, uint32_t n_common, cand_t* c)
{
	const uint32_t k = e->k;
	c->n = 0; c->tot = 0;
	for (uint32_t i = 0; i < n_common; ++i)
	{
		uint64_t km = common_sorted[i];
		int64_t ie = unique_pos(enc, enc_len, k, km, NULL, 0, 0), ir = unique_pos(ref, ref_len, k, km, common_sorted, n_common, e->modulo);
		if (ie == -1 || ir == -1) { km = revcomp_k(km, k); ie = unique_pos(enc, enc_len, k, km, NULL, 0, 0); ir = unique_pos(ref, ref_len, k, km, common_sorted, n_common, e->modulo); }
		if (ie != -1 && ir != -1) { anchor_t a = { k, (uint32_t)ie, (uint32_t)ir }; cand_push(c, a); }
	}
	if (c->n == 0) return KR_NO_ANCHORS;
	qsort(c->a, c->n, sizeof(anchor_t), cmp_anchor_enc);
	for (uint32_t i = 1; i < c->n; ++i) if (c->a[i].pos_ref < c->a[i - 1].pos_ref) return KR_INCOMPATIBLE;
	for (uint64_t i = 0; i + 1 < c->n; ++i)                                          /* drop overlapping k-mers (:912-920) */
		if (c->a[i].pos_enc + c->a[i].len > c->a[i + 1].pos_enc || c->a[i].pos_ref + c->a[i].len > c->a[i + 1].pos_ref) { anchors_erase(c, (uint32_t)i + 1); --i; }
	anchor_t* f = &c->a[0];
	while (f->pos_enc > 0 && f->pos_ref > 0 && enc[f->pos_enc - 1] == ref[f->pos_ref - 1]) { --f->pos_enc; --f->pos_ref; ++f->len; }
	for (uint64_t i = 0; i < c->n; ++i)
	{
		if (i > 0)
		{
			uint32_t pe = c->a[i - 1].pos_enc + c->a[i - 1].len, pr = c->a[i - 1].pos_ref + c->a[i - 1].len;
			for (;;)
			{
				int re = c->a[i].pos_enc == pe, rr = c->a[i].pos_ref == pr;
				if (re && rr) { c->a[i].len += c->a[i - 1].len; anchors_erase(c, (uint32_t)i - 1); break; }   /* literal: positions are not moved (:944-948) */
				if (re || rr) break;
				if (enc[c->a[i].pos_enc - 1] != ref[c->a[i].pos_ref - 1]) break;
				++c->a[i].len; --c->a[i].pos_enc; --c->a[i].pos_ref;
			}
		}
		if (i != c->n - 1)
		{
			uint32_t ne_ = c->a[i + 1].pos_enc, nr_ = c->a[i + 1].pos_ref;
			uint32_t pe = c->a[i].pos_enc + c->a[i].len, pr = c->a[i].pos_ref + c->a[i].len;
			for (;;)
			{
				int re = pe == ne_, rr = pr == nr_;
				if (re && rr) { c->a[i].len += c->a[i + 1].len; anchors_erase(c, (uint32_t)i + 1); --i; break; }
				else if (re || rr) break;
				if (enc[pe] != ref[pr]) break;
				++pe; ++pr; ++c->a[i].len;
			}
		}
	}
	anchor_t* l = &c->a[c->n - 1];
	uint32_t pe = l->pos_enc + l->len, pr = l->pos_ref + l->len;
	while (pe < enc_len && pr < ref_len && enc[pe] == ref[pr]) { ++pe; ++pr; ++l->len; }
	for (uint32_t i = 0; i < c->n; ++i) c->tot += c->a[i].len;
	return KR_ACCEPT;
}

/* fixOverlaping* (encoder.cpp:1577-1622) */
static void fix_overlaps(cand_t* c)
{
	for (uint32_t i = 0; i + 1 < c->n; ++i)
	{
		uint32_t end = c->a[i].pos_ref + c->a[i].len;
		if (c->a[i + 1].pos_ref < end) { uint32_t d = end - c->a[i + 1].pos_ref; c->a[i + 1].pos_ref += d; c->a[i + 1].len -= d; c->a[i + 1].pos_enc += d; }
	}
	for (uint32_t i = 0; i + 1 < c->n; ++i)
	{
		uint32_t end = c->a[i].pos_enc + c->a[i].len;
		if (c->a[i + 1].pos_enc < end) { uint32_t d = end - c->a[i + 1].pos_enc; c->a[i + 1].pos_enc += d; c->a[i + 1].len -= d; c->a[i + 1].pos_ref += d; }
	}
}
/* AdjustAnchors (encoder.cpp:778-868) */
static uint32_t adjust_anchors(const orc_encoder* e, cand_t* c, uint32_t ns, uint32_t ne)
{
	const uint32_t G = 0xffffffffu; uint32_t first = G, last = G, tot = 0;
	for (uint32_t i = 0; i < c->n; ++i) if (c->a[i].pos_enc + c->a[i].len > ns) { first = i; break; }
	if (first == G) { c->n = 0; return 0; }
	if (c->a[first].pos_enc < ns && (c->a[first].pos_enc + c->a[first].len) - ns < e->m) ++first;
	for (int32_t i = (int32_t)c->n - 1; i >= 0; --i) if (c->a[i].pos_enc < ne) { last = (uint32_t)i; break; }
	if (last == G) { c->n = 0; return 0; }
	if (c->a[last].pos_enc + c->a[last].len > ne && ne - c->a[last].pos_enc < e->m) { if (last == 0) { c->n = 0; return 0; } --last; }
	if (first > last) { c->n = 0; return 0; }
	c->n = last + 1;
	memmove(c->a, c->a + first, (c->n - first) * sizeof(anchor_t)); c->n -= first;
	if (!c->n) return 0;
	anchor_t* b = &c->a[c->n - 1];
	if (b->pos_enc + b->len > ne) b->len -= (b->pos_enc + b->len - ne);
	for (uint32_t i = 0; i < c->n; ++i)
	{
		if (i == 0 && c->a[0].pos_enc < ns) { uint32_t d = ns - c->a[0].pos_enc; c->a[0].len -= d; c->a[0].pos_enc = 0; c->a[0].pos_ref += d; }
		else c->a[i].pos_enc -= ns;
		tot += c->a[i].len;
	}
	return tot;
}
/* std::sort on <= 16 elements is libstdc++'s insertion sort, i.e. stable (SURVEY §7 hard part 6) */
static void sort_cands_desc(cand_t* c, uint32_t n)
{
	if (n > 16) { fprintf(stderr, "oracle: more than 16 candidates, std::sort order not restated\n"); abort(); }
	for (uint32_t i = 1; i < n; ++i) { cand_t x = c[i]; uint32_t j = i; while (j > 0 && x.tot > c[j - 1].tot) { c[j] = c[j - 1]; --j; } c[j] = x; }
}

/* ---- tuple emission (encoder.cpp:1348-1443) ----------------------------------------------------------- */
typedef struct { uint8_t* p; size_t n, cap; uint32_t n_tuples; } esb_t;
static void esb_byte(esb_t* b, uint8_t v) { if (b->n == b->cap) { b->cap = b->cap ? 2 * b->cap : 1024; b->p = (uint8_t*)realloc(b->p, b->cap); } b->p[b->n++] = v; }
static void esb_t1(esb_t* b, uint32_t type, uint32_t val) { esb_byte(b, (uint8_t)((type << 4) + val)); ++b->n_tuples; }
static void esb_t28(esb_t* b, uint32_t type, uint32_t v) { esb_byte(b, (uint8_t)((type << 4) + (v >> 24))); esb_byte(b, (v >> 16) & 0xff); esb_byte(b, (v >> 8) & 0xff); esb_byte(b, v & 0xff); ++b->n_tuples; }
static void esb_id(esb_t* b, uint32_t type, uint32_t id, uint32_t rev) { esb_byte(b, (uint8_t)((type << 4) + rev)); esb_byte(b, id >> 24); esb_byte(b, (id >> 16) & 0xff); esb_byte(b, (id >> 8) & 0xff); esb_byte(b, id & 0xff); ++b->n_tuples; }
static void store_symbol_run(esb_t* b, char s, uint32_t rep)                       /* singleEditScriptSymbolStore */
{
	if (s == 'M') { if (rep >= 15) esb_t28(b, 4, rep); else for (uint32_t i = 0; i < rep; ++i) esb_t1(b, 2, 0); }
	else if (s == 'D') { if (rep > 16) esb_t28(b, 5, rep); else for (uint32_t i = 0; i < rep; ++i) esb_t1(b, 1, 0); }
	else if (is_mismatch(s)) { for (uint32_t i = 0; i < rep; ++i) esb_t1(b, 3, (uint32_t)(s - 'X')); }
	else { uint32_t code = s == 'A' ? 0 : s == 'C' ? 1 : s == 'G' ? 2 : 3; for (uint32_t i = 0; i < rep; ++i) esb_t1(b, 0, code); }
}
static void big_es_to_tuples(esb_t* b, const char* s, size_t n)
{
	char sym = s[0]; uint32_t rep = 1;
	for (size_t i = 1; i < n; ++i) { if (s[i] != sym) { store_symbol_run(b, sym, rep); sym = s[i]; rep = 1; } else ++rep; }
	store_symbol_run(b, sym, rep);
}
/* StoreFrag (encoder.cpp:1414-1443) */
static void store_frag(str_t* big, uint32_t level, esb_t* out, uint32_t ref_id, uint32_t main_id, uint32_t* last_pos_in_ref, uint32_t cur_pos_in_ref, int* first, int rev)
{
	if (big->n)
	{
		if (level == 0)
		{
			if (ref_id != main_id) esb_id(out, 6, ref_id, (uint32_t)rev);
			else if (!*first) esb_t1(out, 7, 0);
			big_es_to_tuples(out, big->p, big->n);
		}
		else
		{
			if (ref_id != main_id) esb_id(out, 6, ref_id, (uint32_t)rev); else esb_t1(out, 7, 0);
			str_t tmp = { 0, 0, 0 };
			str_append_n(&tmp, *last_pos_in_ref, 'D'); str_append(&tmp, big->p, big->n);
			big_es_to_tuples(out, tmp.p, tmp.n);
			str_free(&tmp);
		}
		*last_pos_in_ref = cur_pos_in_ref;
		*first = 0;
	}
	big->n = 0;
}

static void add_encoded(orc_encoder* e, const uint8_t* enc, uint32_t enc_len, cand_t* cands, uint32_t n_cands, uint32_t level, esb_t* out, uint32_t main_id, int* first);

/* GetEditDist (encoder.cpp:1255-1283) */
static edres_t get_edit_dist(const uint8_t* ref, uint32_t nr, const uint8_t* enc, uint32_t ne, uint32_t frag, uint32_t n_frag)
{
	edres_t r; memset(&r, 0, sizeof(r));
	if (nr == 0 || ne == 0)                                                          /* get_edit_dist_on_seq_empty */
	{
		if (nr == 0) { r.dist = ne; for (uint32_t i = 0; i < ne; ++i) str_push(&r.es, "ACGT"[enc[i]]); }
		else { r.dist = nr; str_append_n(&r.es, nr, 'D'); }
		return r;
	}
	uint32_t max_flank = ne * 2;
	if (frag == 0)
	{	/* find_edit_dist_with_edlib_ex_odwr_reverse (edit_script.h:405-419) */
		uint8_t* rr = (uint8_t*)malloc(nr); uint8_t* re = (uint8_t*)malloc(ne);
		for (uint32_t i = 0; i < nr; ++i) rr[i] = ref[nr - 1 - i];
		for (uint32_t i = 0; i < ne; ++i) re[i] = enc[ne - 1 - i];
		uint32_t use = max_flank < nr ? max_flank : nr, ref_end = 0;
		r = ed_flank(rr, use, re, ne, &ref_end);
		for (size_t a = 0, b = r.es.n; a + 1 < b; ++a) { --b; char t = r.es.p[a]; r.es.p[a] = r.es.p[b]; r.es.p[b] = t; }
		uint32_t ref_offset = (nr - 1) - ref_end;
		refactor_es(ref + ref_offset, enc, &r.es);
		str_t full = { 0, 0, 0 };
		str_append_n(&full, ref_offset, 'D'); str_append(&full, r.es.p, r.es.n);
		str_free(&r.es); r.es = full;
		free(rr); free(re);
	}
	else if (frag == n_frag - 1)
	{
		uint32_t use = max_flank < nr ? max_flank : nr, tmp = 0;
		r = ed_flank(ref, use, enc, ne, &tmp);
		refactor_es(ref, enc, &r.es);
	}
	else
	{
		r = ed_inner(ref, nr, enc, ne);
		refactor_es(ref, enc, &r.es);
	}
	return r;
}

/* EncodePart (encoder.cpp:1445-1511) */
static void encode_part(orc_encoder* e, uint32_t level, const uint8_t* enc, uint32_t frag, uint32_t n_frag, uint32_t end_enc, uint32_t end_ref,
                        cand_t* cands, uint32_t n_cands, uint32_t ref_id, const uint8_t* ref_read, uint32_t ref_len, uint32_t main_id,
                        uint32_t cur_ref, uint32_t cur_enc, str_t* big, esb_t* out, uint32_t* last_pos_in_ref, int* first)
{
	/* read_view::substr clamps the length to what is left of the reference (utils.h:52-56) */
	uint32_t want = end_ref - cur_ref, avail = ref_len - cur_ref;
	uint32_t nr = want < avail ? want : avail;
	const uint8_t* refp = ref_read + cur_ref; const uint8_t* encp = enc + cur_enc; uint32_t ne = end_enc - cur_enc;
	edres_t ed = get_edit_dist(refp, nr, encp, ne, frag, n_frag);
	int decision;
	if (ne < e->min_part_alt) decision = est_encode_with_es(&e->est, ed.es.p, ed.es.n, encp, ne, nr);
	else
	{	/* EncodeWithEditScript (:1315-1327) with GetEditScriptEntropyInput (:1299-1311) */
		size_t nd = 0; while (nd < ed.es.n && ed.es.p[nd] == 'D') ++nd;
		const char* p = ed.es.p; size_t n = ed.es.n;
		if (nd >= 10) { p += nd; n -= nd; }
		decision = entropy_es(p, n) * (double)n * e->cost_mult < entropy_dna(encp, ne) * (double)ne;
	}
	if (decision) str_append(big, ed.es.p, ed.es.n);
	else
	{
		/* EncodeWithAlternativeRead (:1329-1346) */
		int use_alt = 0; cand_t* alt = NULL;
		if (!(n_cands <= level + 1 || ne < e->min_part_alt || level >= e->max_rec))
		{
			alt = (cand_t*)malloc(n_cands * sizeof(cand_t));
			for (uint32_t i = 0; i < n_cands; ++i) alt[i] = cand_clone(&cands[i]);
			for (uint32_t i = level + 1; i < n_cands; ++i) alt[i].tot = adjust_anchors(e, &alt[i], cur_enc, end_enc);
			sort_cands_desc(alt + level + 1, n_cands - level - 1);
			use_alt = alt[level + 1].tot != 0;
		}
		if (use_alt)
		{
			store_frag(big, level, out, ref_id, main_id, last_pos_in_ref, cur_ref, first, cands[level].rev);
			add_encoded(e, encp, ne, alt, n_cands, level + 1, out, main_id, first);
			if (frag != n_frag - 1) str_append_n(big, end_ref - cur_ref, 'D');
		}
		else
		{
			for (uint32_t i = 0; i < ne; ++i) str_push(big, "ACGT"[encp[i]]);
			if (frag != n_frag - 1) str_append_n(big, end_ref - cur_ref, 'D');
		}
		if (alt) { for (uint32_t i = 0; i < n_cands; ++i) free(alt[i].a); free(alt); }
	}
	str_free(&ed.es);
}

/* AddEncodedReadWithCandidates (encoder.cpp:1513-1575) */
static void add_encoded(orc_encoder* e, const uint8_t* enc, uint32_t enc_len, cand_t* cands, uint32_t n_cands, uint32_t level, esb_t* out, uint32_t main_id, int* first)
{
	const uint32_t ref_id = cands[level].ref_id;
	const cand_t* c = &cands[level];
	uint32_t last_pos_in_ref = 0;
	if (level == 0) esb_id(out, 10, main_id, (uint32_t)cands[0].rev);
	str_t big = { 0, 0, 0 };
	uint8_t* ref_read = ref_oriented(e, ref_id, c->rev); const uint32_t ref_len = e->refs[ref_id].len;
	const uint32_t n_frag = c->n * 2 + 1;
	uint32_t anch = 0, cur_ref = 0, cur_enc = 0;
	for (uint32_t i = 0; i < n_frag; ++i)
	{
		if (i % 2 == 0)
		{
			uint32_t end_enc = i == n_frag - 1 ? enc_len : c->a[anch].pos_enc;
			uint32_t end_ref = i == n_frag - 1 ? ref_len : c->a[anch].pos_ref;
			encode_part(e, level, enc, i, n_frag, end_enc, end_ref, cands, n_cands, ref_id, ref_read, ref_len, main_id, cur_ref, cur_enc, &big, out, &last_pos_in_ref, first);
		}
		else
		{
			str_append_n(&big, c->a[anch].len, 'M');
			cur_ref = c->a[anch].pos_ref + c->a[anch].len; cur_enc = c->a[anch].pos_enc + c->a[anch].len;
			++anch;
		}
	}
	store_frag(&big, level, out, ref_id, main_id, &last_pos_in_ref, cur_ref, first, c->rev);
	str_free(&big); free(ref_read);
}

/* prepareEncodeCandidates[HiFi] (encoder.cpp:1058-1111,1194-1253) + fixOverlaping* (:1651-1655).
 * Returns the number of candidates (0 = the read is stored plain). */
static uint32_t prepare_candidates(orc_encoder* e, const uint8_t* read, uint32_t len, const uint32_t* neighbours, uint32_t n_nb,
                                   const uint64_t* common, const uint32_t* common_off, cand_t** out)
{
	cand_t* cands = NULL; uint32_t n_cands = 0;
	*out = NULL;
	if (n_nb == 0) return 0;
	mindex_t ix; mindex_build(&ix, read, len, e->m);
	int decision = -1;
	if ((double)ix.n_uniq > e->frac_always * (double)len) decision = 0;
	else if ((double)ix.n_uniq < e->frac_min * (double)len) decision = 1;
	if (decision != 1)
	{
		cands = (cand_t*)calloc(n_nb, sizeof(cand_t));
		for (uint32_t i = 0; i < n_nb; ++i)
		{
			const uint32_t id = neighbours[i];
			uint8_t* fwd = ref_oriented(e, id, 0); uint8_t* rc = ref_oriented(e, id, 1); const uint32_t rl = e->refs[id].len;
			cand_t cf, cr; memset(&cf, 0, sizeof(cf)); memset(&cr, 0, sizeof(cr));
			cf.ref_id = cr.ref_id = id; cr.rev = 1;
			int chosen = -1;                           /* 0 fwd, 1 rc */
			if (e->source == 2 && common)
			{	/* KmerBasedAnchors (:1113-1147) */
				uint32_t nc = common_off[i + 1] - common_off[i];
				uint64_t* cs = (uint64_t*)malloc((nc ? nc : 1) * 8); memcpy(cs, common + common_off[i], nc * 8);
				qsort(cs, nc, 8, cmp_u64);
				int rr = analyse_ref_kmers(e, read, len, rc, rl, cs, nc, &cr);
				int rf = analyse_ref_kmers(e, read, len, fwd, rl, cs, nc, &cf);
				free(cs);
				if (rr == KR_ACCEPT && rf == KR_ACCEPT) chosen = cf.tot > cr.tot ? 0 : 1;
				else if (rr == KR_ACCEPT) chosen = 1;
				else if (rf == KR_ACCEPT) chosen = 0;
			}
			if (chosen < 0)
			{	/* MmerBasedAnchors (:1149-1192): reverse complement analysed first, wins ties */
				int rr = analyse_ref(e, &ix, len, rc, rl, &cr, decision);
				int rf = analyse_ref(e, &ix, len, fwd, rl, &cf, decision);
				if (rr == AR_ACCEPT && rf == AR_ACCEPT) chosen = cf.tot > cr.tot ? 0 : 1;
				else if (rr == AR_ACCEPT) chosen = 1;
				else if (rf == AR_ACCEPT) chosen = 0;
			}
			if (chosen == 0) { cands[n_cands++] = cf; free(cr.a); }
			else if (chosen == 1) { cands[n_cands++] = cr; free(cf.a); }
			else { free(cf.a); free(cr.a); }
			free(fwd); free(rc);
		}
		sort_cands_desc(cands, n_cands);
	}
	mindex_free(&ix);
	for (uint32_t i = 0; i < n_cands; ++i) fix_overlaps(&cands[i]);
	if (n_cands == 0) { free(cands); cands = NULL; }
	*out = cands;
	return n_cands;
}

/* The candidates of one read as the encoder will use them (after orientation choice, sort and overlap fixing):
 * out_cand[i] = {ref_id, rev, tot_anchor_len, n_anchors}, anchors appended to out_anchors as {len, pos_enc, pos_ref}. */
uint32_t orc_encoder_candidates(orc_encoder* e, const uint8_t* read, uint32_t len, const uint32_t* neighbours, uint32_t n_nb,
                                const uint64_t* common, const uint32_t* common_off, uint32_t* out_cand, uint32_t* out_anchors, size_t cap_anchors, size_t* n_anchors)
{
	cand_t* c = NULL;
	uint32_t n = prepare_candidates(e, read, len, neighbours, n_nb, common, common_off, &c);
	size_t na = 0;
	for (uint32_t i = 0; i < n; ++i)
	{
		out_cand[4 * i] = c[i].ref_id; out_cand[4 * i + 1] = (uint32_t)c[i].rev; out_cand[4 * i + 2] = c[i].tot; out_cand[4 * i + 3] = c[i].n;
		for (uint32_t j = 0; j < c[i].n; ++j, ++na)
			if (na < cap_anchors) { out_anchors[3 * na] = c[i].a[j].len; out_anchors[3 * na + 1] = c[i].a[j].pos_enc; out_anchors[3 * na + 2] = c[i].a[j].pos_ref; }
		free(c[i].a);
	}
	free(c);
	*n_anchors = na;
	return n;
}

/* processComprElem (encoder.cpp:1625-1661) for one read.  neighbours: candidate reference ids from the graph;
 * common / common_off: HiFi shared k-mers per neighbour (NULL otherwise).  Returns the tuple stream. */
size_t orc_encoder_encode(orc_encoder* e, const uint8_t* read, uint32_t len, int has_n, const uint32_t* neighbours, uint32_t n_nb,
                          const uint64_t* common, const uint32_t* common_off, uint8_t* out, size_t cap, uint32_t* n_tuples)
{
	esb_t b = { 0, 0, 0, 0 };
	int plain = 0;
	cand_t* cands = NULL; uint32_t n_cands = 0;
	if (has_n) { esb_t1(&b, 11, 0); for (uint32_t i = 0; i < len; ++i) esb_t1(&b, 8, read[i]); goto done; }
	est_log_read(&e->est, read, len);
	n_cands = prepare_candidates(e, read, len, neighbours, n_nb, common, common_off, &cands);
	if (n_cands == 0) plain = 1;
	if (plain) { esb_t1(&b, 9, 0); for (uint32_t i = 0; i < len; ++i) esb_t1(&b, 8, read[i]); }
	else
	{
		int first = 1;
		add_encoded(e, read, len, cands, n_cands, 0, &b, cands[0].ref_id, &first);
	}
done:
	if (cands) { for (uint32_t i = 0; i < n_cands; ++i) free(cands[i].a); free(cands); }
	if (n_tuples) *n_tuples = b.n_tuples;
	size_t n = b.n;
	if (out && cap >= n) memcpy(out, b.p, n);
	free(b.p);
	return n;
}
