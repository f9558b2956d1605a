/* qual.c — oracle restatement of CQualityCoder (src/colord/quality_coder.cpp, quality_coder_impl.cpp)
 * and of the per-pack framing of CEntrComprQuals (entr_qual.h:68-79,100-135).  TEST INFRASTRUCTURE ONLY.
 *
 * One uniform context numbering is used for encoder and decoder (the reference's two sides number the
 * flag bits differently, quality_coder_impl.cpp:112-114 vs :489-490; contexts are only identities).
 */
#include "oracle.h"
#include "rc.h"
#include <math.h>

enum { QM_ORIGINAL = 0, QM_QUINARY_AVG, QM_QUAD_AVG, QM_BINARY_AVG, QM_QUINARY_THR, QM_QUAD_THR, QM_BINARY_THR, QM_AVERAGE, QM_NONE };   /* params.h:33-43 */

struct orc_qual {
	int compress, mode, source, level;
	uint32_t map_fwd[96], map_rev[96], quant[96];
	uint32_t n_ctx_sym, bits_per_sym, ctx_bits; uint64_t ctx_mask;
	uint32_t n_bins;
	orc_ctxmap sym_map;     /* per-base symbol models */
	orc_ctxmap byte_map;    /* 256-symbol models for the averages (Fenwick<256,2^18,8>) */
	orc_bytes out; orc_rce enc;
	orc_rcd dec;
};

static void fill(uint32_t* a, int lo, int hi, uint32_t v) { for (int i = lo; i < hi; ++i) a[i] = v; }

/* adjust_quality_map_symbols (quality_coder.cpp:250-270) */
static void map_bins(orc_qual* q, uint32_t n, const uint32_t* fwd, int n_fwd, const uint32_t* rev, int n_rev)
{
	if (n_fwd > 0)
	{
		fill(q->map_fwd, 0, (int)fwd[0], 0);
		for (uint32_t bin = 1; bin + 1 < n; ++bin) fill(q->map_fwd, (int)fwd[bin - 1], (int)fwd[bin], bin);
		fill(q->map_fwd, (int)fwd[n - 2], 96, n - 1);
	}
	if (n_rev > 0) for (uint32_t i = 0; i < n; ++i) q->map_rev[i] = rev[i];
}
/* quantisation of the previous-quality context in Original mode (quality_coder.cpp:276-504) */
static void map_lossless(orc_qual* q)
{
	for (int i = 0; i < 96; ++i) q->map_fwd[i] = q->map_rev[i] = (uint32_t)i;
	uint32_t* p = q->quant;
	memset(p, 0, sizeof(q->quant));
	static const int ont3[] = { 0, 1, 2, 4, 7, 11, 16, 22, 29, 37, 46, 56, 67, 79, 90, 96 };
	static const int ont12[] = { 0, 1, 2, 5, 10, 15, 20, 25, 35, 50, 70, 96 };
	static const int pb3[] = { 0, 1, 10, 20, 30, 39, 45, 51, 57, 63, 69, 75, 81, 87, 93, 94 };
	static const int pb12[] = { 0, 1, 15, 29, 41, 53, 63, 72, 80, 87, 93, 94 };
	const int* t; int n;
	if (q->source == 0) { if (q->level == 3) { t = ont3; n = 15; } else { t = ont12; n = 11; } }
	else { if (q->level == 3) { t = pb3; n = 15; } else { t = pb12; n = 11; } }
	for (int b = 0; b < n; ++b) fill(p, t[b], t[b + 1], (uint32_t)b);
	if (q->source == 2)                /* HiFi: classes shifted by one, Q93 gets class 0 (:456-504) */
	{
		for (int i = 0; i < 93; ++i) p[i] += 1;
		p[93] = 0;
	}
}

orc_qual* orc_qual_new(int compress, int mode, int source, int level, const uint32_t* fwd, int n_fwd, const uint32_t* rev, int n_rev)
{
	orc_qual* q = (orc_qual*)calloc(1, sizeof(*q));
	q->compress = compress; q->mode = mode; q->source = source; q->level = level;
	uint32_t n_sym = 0;
	switch (mode)                      /* quality_coder.cpp:58-240 */
	{
	case QM_ORIGINAL: map_lossless(q); q->bits_per_sym = 4; q->n_ctx_sym = 2; n_sym = 96; break;
	case QM_QUINARY_AVG: case QM_QUINARY_THR:
		q->n_bins = 5; map_bins(q, 5, fwd, n_fwd, rev, n_rev); q->bits_per_sym = 3; q->n_ctx_sym = 3; n_sym = 5; break;
	case QM_QUAD_AVG: case QM_QUAD_THR:
		q->n_bins = 4; map_bins(q, 4, fwd, n_fwd, rev, n_rev); q->bits_per_sym = 3; q->n_ctx_sym = 3; n_sym = 4; break;
	case QM_BINARY_AVG: case QM_BINARY_THR:
		q->n_bins = 2; map_bins(q, 2, fwd, n_fwd, rev, n_rev); q->bits_per_sym = 2; q->n_ctx_sym = 6; n_sym = 2; break;
	case QM_AVERAGE: q->bits_per_sym = 8; q->n_ctx_sym = 2; n_sym = 2; break;
	case QM_NONE: if (n_rev > 0) q->map_rev[0] = rev[0]; n_sym = 2; break;
	}
	q->ctx_bits = q->bits_per_sym * q->n_ctx_sym;
	q->ctx_mask = (1ULL << q->ctx_bits) - 1;
	if (mode == QM_ORIGINAL) orc_ctxmap_init(&q->sym_map, 96, 1u << 20, 32);     /* quality_coder.h:36 */
	else orc_ctxmap_init(&q->sym_map, n_sym, 1u << 18, 8);                       /* :37-39 */
	orc_ctxmap_init(&q->byte_map, 256, 1u << 18, 8);                             /* :41 */
	q->enc.out = &q->out;
	if (compress) orc_rce_start(&q->enc);
	return q;
}
void orc_qual_free(orc_qual* q)
{
	if (!q) return;
	orc_ctxmap_free(&q->sym_map); orc_ctxmap_free(&q->byte_map); free(q->out.p); free(q);
}

static inline uint64_t vs(uint8_t x) { return (uint64_t)(x & 3); }               /* valid_sym: N aliases A */
static inline uint64_t flag_bits(const orc_qual* q, const uint8_t* flags, uint32_t i)
{
	if (q->level <= 1 || !flags) return 0;
	return (uint64_t)(flags[i] == 'M') | ((uint64_t)(flags[i] == 'A') << 1);
}

/* per-base context of the *-avg / *-fix families (quality_coder_impl.cpp:203-224): previous bins |
 * bases i-2..i+1 | flags */
static inline uint64_t ctx_bins(const orc_qual* q, uint64_t hist, uint64_t dna_ctx, const uint8_t* flags, uint32_t i)
{
	return hist + (dna_ctx << q->ctx_bits) + (flag_bits(q, flags, i) << (q->ctx_bits + 8));
}
/* threshold families build the base context from explicit neighbours (:323-339) */
static inline uint64_t ctx_thr(const orc_qual* q, uint64_t hist, const uint8_t* b, uint32_t len, uint32_t i, const uint8_t* flags)
{
	uint64_t c = hist; uint32_t sh = q->ctx_bits;
	c += vs(b[i]) << sh;
	sh += 2;
	if (i > 0) c += vs(b[i - 1]) << sh;
	sh += 2;
	if (i > 1) c += vs(b[i - 2]) << sh;
	sh += 2;
	if (i + 1 < len) c += vs(b[i + 1]) << sh;
	sh += 2;
	return c + (flag_bits(q, flags, i) << sh);
}
static inline uint64_t ctx_org(const orc_qual* q, uint64_t hist, const uint8_t* b, uint32_t len, uint32_t i, const uint8_t* flags)
{	/* :88-112 */
	uint64_t c = hist; uint32_t sh = q->ctx_bits;
	c += vs(b[i]) << sh;
	sh += 2;
	if (i > 0) c += vs(b[i - 1]) << sh;
	sh += 2;
	if (q->level == 3)
	{
		if (i > 1) c += vs(b[i - 2]) << sh;
		sh += 2;
	}
	else
	{
		if (i > 1) c += (uint64_t)(vs(b[i - 2]) == vs(b[i - 1])) << sh;
		sh += 1;
	}
	if (i + 1 < len) c += vs(b[i + 1]) << sh;
	sh += 2;
	return c + (flag_bits(q, flags, i) << sh);
}

static void enc_avg(orc_qual* q, uint64_t ctx_base, double x)                    /* :821-834 */
{
	uint32_t a = (uint32_t)(x * 256), a1 = a >> 8, a2 = a & 0xff;
	orc_encode_sym(&q->enc, &q->byte_map, ctx_base, a1, -1, -1);
	orc_encode_sym(&q->enc, &q->byte_map, a1 + 0x100ULL, a2, -1, -1);
}
static double dec_avg(orc_qual* q, uint64_t ctx_base)                            /* :837-849 */
{
	uint32_t a1 = orc_decode_sym(&q->dec, &q->byte_map, ctx_base, -1, -1);
	uint32_t a2 = orc_decode_sym(&q->dec, &q->byte_map, a1 + 0x100ULL, -1, -1);
	return (double)((a1 << 8) + a2) / 256.0;
}

/* bases: codes 0..4 (+ the reference's guard is not needed: len is explicit; read[0] of an empty read is
 * the guard 255, whose two low bits are 3).  flags: per base 'A'/'M'/' '/'P' (analyze_es) or NULL. */
void orc_qual_encode(orc_qual* q, const uint8_t* b, const uint8_t* qual, uint32_t len, const uint8_t* flags)
{
	if (q->mode == QM_NONE) return;
	uint64_t hist = q->ctx_mask;                                                  /* reset_context */
	if (q->mode == QM_ORIGINAL)
	{
		for (uint32_t i = 0; i < len; ++i)
		{
			uint32_t s = q->map_fwd[qual[i] - 33u];
			orc_encode_sym(&q->enc, &q->sym_map, ctx_org(q, hist, b, len, i, flags), s, -1, -1);
			hist = ((hist << q->bits_per_sym) + q->quant[s]) & q->ctx_mask;
		}
		return;
	}
	if (q->mode == QM_AVERAGE)                                                    /* :438-450 */
	{
		double avg = 0.0;
		for (uint32_t i = 0; i < len; ++i) avg += qual[i] - 33u;
		avg /= len;
		enc_avg(q, 0ULL, avg);
		return;
	}
	const int is_avg = q->mode == QM_QUINARY_AVG || q->mode == QM_QUAD_AVG || q->mode == QM_BINARY_AVG;
	if (is_avg)
	{	/* :138-166: per-bin averages over a 128-entry histogram, accumulated in symbol order */
		double sum[5] = { 0, 0, 0, 0, 0 }; uint32_t cnt[5] = { 0, 0, 0, 0, 0 }; uint32_t h[128];
		memset(h, 0, sizeof(h));
		for (uint32_t i = 0; i < len; ++i) ++h[qual[i]];
		for (uint32_t i = 33; i < 128; ++i) { uint32_t bin = q->map_fwd[i - 33u]; sum[bin] += (double)(i - 33u) * h[i]; cnt[bin] += h[i]; }
		uint64_t ctx_p = 0;
		for (uint32_t i = 0; i < q->n_bins; ++i)
		{
			double avg = cnt[i] ? sum[i] / cnt[i] : 0.0;
			enc_avg(q, (1ULL << 30) + ((uint64_t)i << 24) + (ctx_p << 16), avg);
			ctx_p = (uint64_t)avg;
		}
		uint64_t dna = len ? vs(b[0]) : 3;
		for (uint32_t i = 0; i < len; ++i)
		{
			dna <<= 2; if (i + 1 < len) dna += vs(b[i + 1]); dna &= 0xff;
			uint32_t s = q->map_fwd[qual[i] - 33];
			orc_encode_sym(&q->enc, &q->sym_map, ctx_bins(q, hist, dna, flags, i), s, -1, -1);
			hist = ((hist << q->bits_per_sym) + s) & q->ctx_mask;
		}
		return;
	}
	for (uint32_t i = 0; i < len; ++i)                                            /* *-fix (:313-435) */
	{
		uint32_t s = q->map_fwd[qual[i] - 33];
		orc_encode_sym(&q->enc, &q->sym_map, ctx_thr(q, hist, b, len, i, flags), s, -1, -1);
		hist = ((hist << q->bits_per_sym) + s) & q->ctx_mask;
	}
}

/* Finish + GetOutput + Restart (entr_qual.h:68-79): returns the part payload and starts a new one.
 * Models persist. */
size_t orc_qual_finish_part(orc_qual* q, uint8_t* dst, size_t cap)
{
	if (!dst) return q->out.n + 8;      /* size query: payload so far + the 8 flush bytes */
	orc_rce_end(&q->enc);
	size_t n = q->out.n;
	if (cap >= n) memcpy(dst, q->out.p, n);
	q->out.n = 0; orc_rce_start(&q->enc);
	return n;
}

void orc_qual_set_input(orc_qual* q, const uint8_t* data, size_t n)
{
	q->dec.in = data; q->dec.n = n; q->dec.pos = 0;
	orc_rcd_start(&q->dec);
}
void orc_qual_decode(orc_qual* q, const uint8_t* b, uint32_t len, const uint8_t* flags, uint8_t* out)
{
	if (q->mode == QM_NONE) { for (uint32_t i = 0; i < len; ++i) out[i] = (uint8_t)(33 + q->map_rev[0]); return; }   /* quality_coder.cpp:611-617 */
	uint64_t hist = q->ctx_mask;
	if (q->mode == QM_ORIGINAL)
	{
		for (uint32_t i = 0; i < len; ++i)
		{
			uint32_t d = orc_decode_sym(&q->dec, &q->sym_map, ctx_org(q, hist, b, len, i, flags), -1, -1);
			uint32_t v = q->map_rev[d];
			out[i] = (uint8_t)(v + 33);
			hist = ((hist << q->bits_per_sym) + q->quant[v]) & q->ctx_mask;
		}
		return;
	}
	if (q->mode == QM_AVERAGE)                                                    /* :800-817 */
	{
		double avg = dec_avg(q, 0ULL), as = 0.0, qs = 0.0;
		for (uint32_t i = 0; i < len; ++i) { as += avg; uint32_t v = (uint32_t)(as - qs); qs += v; out[i] = (uint8_t)(v + 33); }
		return;
	}
	const int is_avg = q->mode == QM_QUINARY_AVG || q->mode == QM_QUAD_AVG || q->mode == QM_BINARY_AVG;
	if (is_avg)
	{	/* :506-559: error-diffusion reconstruction in IEEE double */
		double avg[5], as[5] = { 0, 0, 0, 0, 0 }, qs[5] = { 0, 0, 0, 0, 0 };
		uint64_t ctx_p = 0;
		for (uint32_t i = 0; i < q->n_bins; ++i) { avg[i] = dec_avg(q, (1ULL << 30) + ((uint64_t)i << 24) + (ctx_p << 16)); ctx_p = (uint64_t)avg[i]; }
		uint64_t dna = len ? vs(b[0]) : 3;
		for (uint32_t i = 0; i < len; ++i)
		{
			dna <<= 2; if (i + 1 < len) dna += vs(b[i + 1]); dna &= 0xff;
			uint32_t d = orc_decode_sym(&q->dec, &q->sym_map, ctx_bins(q, hist, dna, flags, i), -1, -1);
			as[d] += avg[d];
			uint32_t v = (uint32_t)(as[d] - qs[d]);
			qs[d] += v;
			out[i] = (uint8_t)(v + 33);
			hist = ((hist << q->bits_per_sym) + d) & q->ctx_mask;
		}
		return;
	}
	for (uint32_t i = 0; i < len; ++i)
	{
		uint32_t d = orc_decode_sym(&q->dec, &q->sym_map, ctx_thr(q, hist, b, len, i, flags), -1, -1);
		out[i] = (uint8_t)(q->map_rev[d] + 33);                                   /* quality_coder.cpp:267-269 */
		hist = ((hist << q->bits_per_sym) + d) & q->ctx_mask;
	}
}

/* analyze_es (quality_coder_impl.cpp:25-75): per-base class from the read's own tuple stream (App. A
 * byte layout): 'P' plain read, 'A' inside an anchor, 'M' unit match, ' ' insertion/substitution. */
void orc_es_flags(const uint8_t* es, size_t n, uint32_t read_len, uint8_t* flags)
{
	if (n == 0) return;
	uint32_t t0 = es[0] >> 4;
	if (t0 == 9 || t0 == 11) { memset(flags, 'P', read_len); return; }
	memset(flags, ' ', read_len);
	size_t p = (t0 == 10) ? 5 : 1;   /* start_es carries a 32-bit id */
	uint32_t o = 0;
	while (p < n)
	{
		uint32_t t = es[p] >> 4;
		switch (t)
		{
		case 4: { uint32_t v = ((uint32_t)(es[p] & 0xf) << 24) | ((uint32_t)es[p + 1] << 16) | ((uint32_t)es[p + 2] << 8) | es[p + 3];
		          memset(flags + o, 'A', v); o += v; p += 4; break; }
		case 5: p += 4; break;
		case 2: flags[o++] = 'M'; p += 1; break;
		case 0: case 3: ++o; p += 1; break;
		case 6: p += 5; break;
		default: p += 1;
		}
	}
}
