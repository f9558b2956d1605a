/* dna.c — oracle restatement of CDNACoder (src/colord/dna_coder.{h,cpp}) and of the part framing of
 * CEntrComprReads (entr_read.h:56-80).  TEST INFRASTRUCTURE ONLY.
 *
 * Input is the reference's tuple stream es_t in its byte layout (utils.h:56-273).  Context
 * values are computed with the same sums as the reference, so the (family, context) partition of the
 * symbols is the reference's.
 */
#include "oracle.h"
#include "rc.h"

enum { T_INS = 0, T_DEL, T_MATCH, T_SUBST, T_ANCHOR, T_SKIP, T_ALT_ID, T_MAIN_REF, T_PLAIN, T_START_PLAIN, T_START_ES, T_START_PLAIN_N, T_NONE };

typedef struct { uint8_t* b; uint32_t len; } refread_t;

struct orc_dna {
	int compress, level, max_alt;
	uint32_t cur_read_id;
	int no_tuples_in_mask, no_symbols_in_mask;
	uint64_t mask_tuple, mask_symbol;
	uint64_t ctx_read_type, ctx_rev_comp, ctx_tuple_type, ctx_symbol;
	int cur_ref_delta;
	/* per-read orientation cache (uo_rev_comp) */
	int rc_ids[64]; int rc_val[64]; int n_rc;
	orc_ctxmap m_read_type, m_rev_comp, m_seen, m_len_bits, m_len_data, m_symbols, m_symbols_n, m_read_id, m_read_id_short,
		m_anchor_len, m_skip_local, m_skip_distant, m_tuple_type;
	refread_t* refs; size_t n_refs, cap_refs;
	orc_bytes out; orc_rce enc; orc_rcd dec;
};

static uint64_t ilog2_(uint64_t x) { uint64_t r = 0; for (; x; ++r) x >>= 1; return r; }      /* basic_coder.h:39-47 */
static uint64_t no_bytes_(uint64_t x) { uint64_t r = 1; x >>= 8; for (; x; ++r) x >>= 8; return r; }   /* :50-60 */

orc_dna* orc_dna_new(int compress, int max_alt_refs, int level, uint32_t start_read_id)
{
	orc_dna* d = (orc_dna*)calloc(1, sizeof(*d));
	d->compress = compress; d->level = level; d->max_alt = max_alt_refs; d->cur_read_id = start_read_id;
	switch (level)                                                                 /* dna_coder.cpp:1253-1280 */
	{
	case 3: d->no_tuples_in_mask = 4; d->no_symbols_in_mask = 8; break;
	case 2: d->no_tuples_in_mask = 3; d->no_symbols_in_mask = 7; break;
	case 1: d->no_tuples_in_mask = 2; d->no_symbols_in_mask = 5; break;
	default: d->no_tuples_in_mask = 1; d->no_symbols_in_mask = 1;
	}
	d->mask_tuple = (1ULL << (3 * d->no_tuples_in_mask)) - 1;
	d->mask_symbol = (1ULL << (2 * d->no_symbols_in_mask)) - 1;
	/* model families, dna_coder.h:48-60: <symbols, MAX_TOTAL, ADDER> */
	orc_ctxmap_init(&d->m_rev_comp, 2, 1u << 15, 1);
	orc_ctxmap_init(&d->m_read_type, 3, 1u << 15, 1);
	orc_ctxmap_init(&d->m_seen, 2, 1u << 15, 1);
	orc_ctxmap_init(&d->m_len_bits, 32, 1u << 18, 8);
	orc_ctxmap_init(&d->m_len_data, 256, 1u << 18, 8);
	orc_ctxmap_init(&d->m_symbols, 4, 1u << 10, 1);
	orc_ctxmap_init(&d->m_symbols_n, 5, 1u << 10, 1);
	orc_ctxmap_init(&d->m_read_id, 256, 1u << 13, 1);
	orc_ctxmap_init(&d->m_skip_distant, 256, 1u << 15, 1);
	orc_ctxmap_init(&d->m_tuple_type, 8, 1u << 15, 1);
	orc_ctxmap_init(&d->m_read_id_short, (uint32_t)max_alt_refs, 1u << 13, 1);
	orc_ctxmap_init(&d->m_anchor_len, 24, 1u << 15, 1);
	orc_ctxmap_init(&d->m_skip_local, 256, 1u << 15, 1);
	d->enc.out = &d->out;
	if (compress) orc_rce_start(&d->enc);
	return d;
}
void orc_dna_free(orc_dna* d)
{
	if (!d) return;
	orc_ctxmap* ms[] = { &d->m_read_type, &d->m_rev_comp, &d->m_seen, &d->m_len_bits, &d->m_len_data, &d->m_symbols, &d->m_symbols_n,
		&d->m_read_id, &d->m_read_id_short, &d->m_anchor_len, &d->m_skip_local, &d->m_skip_distant, &d->m_tuple_type };
	for (size_t i = 0; i < sizeof(ms) / sizeof(ms[0]); ++i) orc_ctxmap_free(ms[i]);
	for (size_t i = 0; i < d->n_refs; ++i) free(d->refs[i].b);
	free(d->refs); free(d->out.p); free(d);
}
/* CReferenceReads::Add (reference_reads.h:209-212); reads are kept as plain codes here */
void orc_dna_add_ref(orc_dna* d, const uint8_t* bases, uint32_t len)
{
	if (d->n_refs == d->cap_refs) { d->cap_refs = d->cap_refs ? 2 * d->cap_refs : 64; d->refs = (refread_t*)realloc(d->refs, d->cap_refs * sizeof(refread_t)); }
	refread_t* r = &d->refs[d->n_refs++];
	r->len = len; r->b = (uint8_t*)malloc(len ? len : 1);
	memcpy(r->b, bases, len);
}
/* GetRefRead(id, rev)[pos] with the trailing guard 255 (reference_reads.h:142-207,214-220) */
static inline uint32_t ref_at(const orc_dna* d, int id, int rev, int64_t pos)
{
	const refread_t* r = &d->refs[id];
	if (pos < 0 || pos >= (int64_t)r->len) return 255;
	return rev ? 3u - r->b[r->len - 1 - pos] : r->b[pos];
}

/* ---- field coders (dna_coder.cpp:440-1239), encoder side ---- */
static void enc_read_flag(orc_dna* d, uint32_t flag)                             /* :440-463 */
{
	orc_encode_sym(&d->enc, &d->m_read_type, d->ctx_read_type, flag, -1, -1);
	d->ctx_read_type = ((d->ctx_read_type << 2) + flag) & 0xff;
}
static void enc_read_len(orc_dna* d, uint32_t len)                               /* :1004-1056 */
{
	int nb = (int)ilog2_(len);
	orc_encode_sym(&d->enc, &d->m_len_bits, 0, (uint32_t)nb, -1, -1);
	if (nb < 2) return;
	uint64_t ctx = (uint64_t)nb << 3;
	len -= 1u << (nb - 1);
	uint32_t prefix, suffix;
	if (nb <= 9) { prefix = len; suffix = 0; }
	else { prefix = len >> (nb - 9); suffix = len - (prefix << (nb - 9)); }
	orc_encode_sym(&d->enc, &d->m_len_data, ctx, prefix, -1, -1);
	if (nb <= 9) return;
	nb -= 9; ctx += 1ULL << 2;
	for (; nb > 0; nb -= 8) { orc_encode_sym(&d->enc, &d->m_len_data, ctx, suffix & 0xff, -1, -1); suffix >>= 8; ++ctx; }
}
static void enc_read_id(orc_dna* d, uint32_t id)                                 /* :535-551 */
{
	int n = (int)no_bytes_(d->cur_read_id);
	for (int i = n - 1; i >= 0; --i)
	{
		uint64_t add = (i == n - 2) ? ((id >> (8 * (n - 1))) & 0xff) : 0;
		orc_encode_sym(&d->enc, &d->m_read_id, (uint64_t)i + (add << 3), (id >> (8 * i)) & 0xff, -1, -1);
	}
}
static void enc_rev_comp(orc_dna* d, int read_id, int rc)                        /* :489-509 */
{
	for (int i = 0; i < d->n_rc; ++i) if (d->rc_ids[i] == read_id) return;
	orc_encode_sym(&d->enc, &d->m_rev_comp, d->ctx_rev_comp, (uint32_t)rc, -1, -1);
	if (d->n_rc < 64) { d->rc_ids[d->n_rc] = read_id; d->rc_val[d->n_rc] = rc; ++d->n_rc; }
	d->ctx_rev_comp = ((d->ctx_rev_comp << 2) + (uint64_t)rc) & 0xf;
}
static void enc_tuple_type(orc_dna* d, uint32_t type, uint32_t ref_symbol, uint32_t last, int first)   /* :651-710 */
{
	uint32_t shift = 3 * d->no_tuples_in_mask;
	uint64_t ctx = d->ctx_tuple_type;
	ctx += (d->ctx_symbol & 0xf) << shift; shift += 4;
	ctx += (uint64_t)ref_symbol << shift; shift += 2;
	if (d->cur_ref_delta < -10) ctx += 1ULL << shift;
	else if (d->cur_ref_delta < -1) ctx += 2ULL << shift;
	else if (d->cur_ref_delta > 10) ctx += 3ULL << shift;
	else if (d->cur_ref_delta > 1) ctx += 4ULL << shift;
	int e1 = -1, e2 = -1;
	if (!first)
		switch (last)
		{
		case T_MATCH: e1 = T_ANCHOR; break;
		case T_DEL: e1 = T_SKIP; break;
		case T_ANCHOR: e1 = T_ANCHOR; e2 = T_MATCH; break;
		case T_SKIP: e1 = T_DEL; e2 = T_SKIP; break;
		case T_MAIN_REF: case T_ALT_ID: e1 = T_ALT_ID; e2 = T_MAIN_REF; break;
		default: break;
		}
	orc_encode_sym(&d->enc, &d->m_tuple_type, ctx, type, e1, e2);
	d->ctx_tuple_type = ((d->ctx_tuple_type << 3) + type) & d->mask_tuple;
}
static uint64_t ctx_insertion(const orc_dna* d, uint32_t base)                   /* :772-811 */
{
	uint32_t shift = 2; uint64_t ctx = 2;
	if (d->level == 1) { ctx += (d->ctx_symbol & 0xff) << shift; shift += 8; }
	else if (d->level == 2) { ctx += (d->ctx_symbol & 0x3ff) << shift; shift += 10; }
	else
	{
		ctx += (d->ctx_symbol & 0x3ff) << shift; shift += 10;
		ctx += (uint64_t)(((d->ctx_symbol >> 10) & 3) == ((d->ctx_symbol >> 8) & 3)) << shift; ++shift;
	}
	ctx += (uint64_t)base << shift; shift += 2;
	ctx += (d->ctx_tuple_type & 0777) << shift;
	return ctx;
}
static uint64_t ctx_substitution(const orc_dna* d, uint32_t base)                /* :889-922 */
{
	uint32_t shift = 2; uint64_t ctx = 1;
	ctx += (d->ctx_symbol & 0x3f) << shift; shift += 6;
	if (d->level == 3) { ctx += (uint64_t)(((d->ctx_symbol >> 6) & 3) == ((d->ctx_symbol >> 4) & 3)) << shift; ++shift; }
	ctx += (uint64_t)base << shift; shift += 2;
	ctx += (d->ctx_tuple_type & 07777) << shift;
	return ctx;
}
static void enc_anchor_len(orc_dna* d, uint32_t len)                             /* :958-978 */
{
	for (uint64_t part = 0; len; ++part)
	{
		if (len < 23) { orc_encode_sym(&d->enc, &d->m_anchor_len, part, len, -1, -1); break; }
		orc_encode_sym(&d->enc, &d->m_anchor_len, part, 23, -1, -1);
		len -= 22;
	}
}
static void enc_skip_len(orc_dna* d, uint32_t len, int local)                    /* :1109-1137 */
{
	if (local)
	{
		for (uint64_t part = 0; len; ++part)
		{
			if (len < 255) { orc_encode_sym(&d->enc, &d->m_skip_local, part, len, -1, -1); break; }
			orc_encode_sym(&d->enc, &d->m_skip_local, part, 255, -1, -1);
			len -= 254;
		}
		return;
	}
	uint32_t encoded = 0;
	for (int i = 3; i >= 0; --i)
	{
		uint32_t x = (len >> (8 * i)) & 0xff;
		orc_encode_sym(&d->enc, &d->m_skip_distant, (uint64_t)i * 64 + ilog2_(encoded), x, -1, -1);
		encoded = (encoded << 8) + x;
	}
}

/* tuple reader over the App. A byte layout */
typedef struct { const uint8_t* p; const uint8_t* e; } esr_t;
static int es_next(esr_t* r, uint32_t* type, uint32_t* v1, uint32_t* v2)
{
	if (r->p >= r->e) return 0;
	uint32_t t = r->p[0] >> 4; *type = t;
	switch (t)
	{
	case T_INS: case T_SUBST: case T_PLAIN: *v1 = r->p[0] & 0xf; r->p += 1; break;
	case T_ANCHOR: case T_SKIP: *v2 = ((uint32_t)(r->p[0] & 0xf) << 24) | ((uint32_t)r->p[1] << 16) | ((uint32_t)r->p[2] << 8) | r->p[3]; r->p += 4; break;
	case T_ALT_ID: case T_START_ES: *v2 = r->p[0] & 0xf; *v1 = ((uint32_t)r->p[1] << 24) | ((uint32_t)r->p[2] << 16) | ((uint32_t)r->p[3] << 8) | r->p[4]; r->p += 5; break;
	default: r->p += 1;
	}
	return 1;
}

static const int subst_to_code[4][4] = { {1, 0, 0, 0}, {2, 2, 1, 1}, {3, 3, 3, 2}, {3, 3, 3, 3} };   /* dna_coder.h:37 */

/* CDNACoder::Encode (dna_coder.cpp:26-231) */
void orc_dna_encode(orc_dna* d, const uint8_t* es, size_t n_bytes, uint32_t n_tuples)
{
	d->ctx_tuple_type = d->mask_tuple; d->ctx_symbol = d->mask_symbol; d->ctx_rev_comp = 0xf; d->n_rc = 0;
	esr_t r = { es, es + n_bytes };
	uint32_t type = T_NONE, v1 = 0, v2 = 0;
	es_next(&r, &type, &v1, &v2);
	enc_read_flag(d, type == T_START_PLAIN ? 0 : type == T_START_PLAIN_N ? 1 : 2);
	enc_read_len(d, n_tuples - 1);
	if (type == T_START_PLAIN)
	{
		while (es_next(&r, &type, &v1, &v2))                                     /* :1178-1196 */
		{
			orc_encode_sym(&d->enc, &d->m_symbols, d->ctx_symbol << 2, v1, -1, -1);
			d->ctx_symbol = ((d->ctx_symbol << 2) + v1) & d->mask_symbol;
		}
		++d->cur_read_id; return;
	}
	if (type == T_START_PLAIN_N)
	{
		while (es_next(&r, &type, &v1, &v2))                                     /* :1213-1227 */
		{
			orc_encode_sym(&d->enc, &d->m_symbols_n, d->ctx_symbol, v1, -1, -1);
			d->ctx_symbol = ((d->ctx_symbol << 4) + v1) & d->mask_symbol;
		}
		++d->cur_read_id; return;
	}
	/* alternative-reference bookkeeping: ids in order of first use, last position, cached orientation */
	int alt_ids[64], alt_pos_of[64], alt_rev_of[64], n_alt = 0;
	int ref_id = (int)v1, ref_rev = (int)v2, alt_id = -1, alt_rev = 0, alt_slot = -1;
	int64_t ref_pos = 0, alt_pos = 0;
	uint32_t last_type = T_NONE, last_flag = T_NONE;
	int is_main = 1, first = 1;
	d->cur_ref_delta = 0;
	enc_read_id(d, (uint32_t)ref_id);
	enc_rev_comp(d, ref_id, ref_rev);
	while (es_next(&r, &type, &v1, &v2))
	{
		uint32_t ref_symbol = is_main ? ref_at(d, ref_id, ref_rev, ref_pos) : ref_at(d, alt_id, alt_rev, alt_pos);
		enc_tuple_type(d, type, ref_symbol, last_flag, first);
		first = 0; last_flag = type;
		switch (type)
		{
		case T_ALT_ID:
		{
			if (!is_main && alt_slot >= 0) alt_pos_of[alt_slot] = (int)alt_pos;      /* :103-105 */
			/* encode_alt_read_id (:572-615) */
			int is_new = 0, slot = -1;
			if (n_alt == 0) { enc_read_id(d, v1); alt_ids[0] = (int)v1; alt_pos_of[0] = 0; n_alt = 1; slot = 0; is_new = 1; }
			else
			{
				int seen = n_alt;
				for (int i = 0; i < n_alt; ++i) if (alt_ids[i] == (int)v1) slot = i;
				int short_id = slot;
				if (slot < 0) { slot = n_alt; alt_ids[n_alt] = (int)v1; alt_pos_of[n_alt] = 0; ++n_alt; is_new = 1; }
				orc_encode_sym(&d->enc, &d->m_seen, (uint64_t)seen, short_id >= 0, -1, -1);
				if (short_id < 0) enc_read_id(d, v1);
				else orc_encode_sym(&d->enc, &d->m_read_id_short, (uint64_t)seen, (uint32_t)short_id, -1, -1);
			}
			if (is_new) alt_rev_of[slot] = (int)v2;                               /* m_alt_read caches the first orientation (:109-110) */
			enc_rev_comp(d, (int)v1, (int)v2);
			alt_id = (int)v1; alt_slot = slot; alt_rev = alt_rev_of[slot];
			alt_pos = 0; is_main = 0; d->cur_ref_delta = 0;
			break;
		}
		case T_ANCHOR:
		{
			enc_anchor_len(d, v2);
			if (is_main) ref_pos += v2; else alt_pos += v2;
			for (int i = d->no_symbols_in_mask; i > 0; --i)                       /* :126-128 */
				d->ctx_symbol = (d->ctx_symbol << 2) + (is_main ? ref_at(d, ref_id, ref_rev, ref_pos - i) : ref_at(d, alt_id, alt_rev, alt_pos - i));
			d->ctx_symbol &= d->mask_symbol;
			d->cur_ref_delta = 0;
			break;
		}
		case T_MATCH:
			d->ctx_symbol = ((d->ctx_symbol << 2) + ref_symbol) & d->mask_symbol;
			if (is_main) ++ref_pos; else ++alt_pos;
			break;
		case T_INS:
			orc_encode_sym(&d->enc, &d->m_symbols, ctx_insertion(d, ref_symbol), v1, -1, -1);
			d->ctx_symbol = ((d->ctx_symbol << 2) + v1) & d->mask_symbol;
			++d->cur_ref_delta;
			break;
		case T_DEL:
			if (is_main) ++ref_pos; else ++alt_pos;
			--d->cur_ref_delta;
			break;
		case T_SUBST:
		{
			uint32_t sym = (uint32_t)subst_to_code[v1][ref_symbol & 3];           /* :158 (ref_symbol < 4 asserted by the reference) */
			orc_encode_sym(&d->enc, &d->m_symbols, ctx_substitution(d, ref_symbol), sym, (int)ref_symbol, -1);
			d->ctx_symbol = ((d->ctx_symbol << 2) + sym) & d->mask_symbol;
			if (is_main) ++ref_pos; else ++alt_pos;
			break;
		}
		case T_SKIP:
		{
			int skip_len = (int)v2;
			d->cur_ref_delta -= skip_len;
			if (!is_main && last_type == T_ALT_ID)                               /* :187-201 */
			{
				int mod = skip_len - alt_pos_of[alt_slot];
				if (mod > 0) enc_skip_len(d, (uint32_t)mod, 0);
				else { enc_skip_len(d, 0, 0); enc_skip_len(d, (uint32_t)(-mod), 0); }
			}
			else enc_skip_len(d, (uint32_t)skip_len, last_type != T_ALT_ID && last_type != T_NONE);
			if (is_main) ref_pos += v2; else alt_pos += v2;
			break;
		}
		case T_MAIN_REF:
			is_main = 1;
			if (alt_slot >= 0) alt_pos_of[alt_slot] = (int)alt_pos;               /* :212 */
			d->cur_ref_delta = 0;
			break;
		default: break;
		}
		last_type = type;
	}
	++d->cur_read_id;
}

/* Finish + GetOutput + Restart (entr_read.h:69-77) */
size_t orc_dna_finish_part(orc_dna* d, uint8_t* dst, size_t cap)
{
	if (!dst) return d->out.n + 8;
	orc_rce_end(&d->enc);
	size_t n = d->out.n;
	if (cap >= n) memcpy(dst, d->out.p, n);
	d->out.n = 0; orc_rce_start(&d->enc);
	return n;
}
