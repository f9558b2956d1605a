/* rc.h — oracle restatement of the 64-bit carry-less range coder and the adaptive frequency models
 * (src/colord/sub_rc.h:44-212,216-392; rc.h:34-220,225-480,487-764).  TEST INFRASTRUCTURE ONLY.
 *
 * All three model classes of the reference (CSimpleModel, CSimpleModelFixedSize, CFenwickTreeModel-
 * FixedSize) implement the same arithmetic: counters start at 1 (or at a template), Update adds ADDER to
 * the coded symbol and to the total, and when total >= MAX_TOTAL every counter becomes (c+1)/2 until the
 * total is below MAX_TOTAL again.  One plain-array model therefore restates all of them.
 */
#ifndef COLORD_ORACLE_RC_H
#define COLORD_ORACLE_RC_H
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint8_t* p; size_t n, cap; } orc_bytes;
static inline void orc_put(orc_bytes* b, uint8_t v)
{
	if (b->n == b->cap) { b->cap = b->cap ? 2 * b->cap : 4096; b->p = (uint8_t*)realloc(b->p, b->cap); }
	b->p[b->n++] = v;
}

#define ORC_TOP 0x00ffffffffffffULL
#define ORC_MASK 0xff00000000000000ULL

/* ---- encoder (sub_rc.h:44-212) ---- */
typedef struct { uint64_t low, range; orc_bytes* out; } orc_rce;
static inline void orc_rce_start(orc_rce* e) { e->low = 0; e->range = ORC_MASK; }          /* :72-76 */
static inline void orc_rce_encode(orc_rce* e, uint64_t freq, uint64_t cum, uint64_t tot)   /* :83-100 */
{
	e->range /= tot;
	e->low += e->range * cum;
	e->range *= freq;
	while (e->range <= ORC_TOP)
	{
		if ((e->low ^ (e->low + e->range)) & ORC_MASK)
		{
			uint64_t r = e->low;
			e->range = (r | ORC_TOP) - r;
		}
		orc_put(e->out, (uint8_t)(e->low >> 56));
		e->low <<= 8; e->range <<= 8;
	}
}
static inline void orc_rce_end(orc_rce* e)                                                   /* :203-210 */
{
	for (int i = 0; i < 8; ++i) { orc_put(e->out, (uint8_t)(e->low >> 56)); e->low <<= 8; }
}

/* ---- decoder (sub_rc.h:216-392) ---- */
typedef struct { uint64_t low, range, buffer; const uint8_t* in; size_t n, pos; } orc_rcd;
static inline uint8_t orc_rcd_byte(orc_rcd* d) { return d->pos < d->n ? d->in[d->pos++] : 0; }
static inline void orc_rcd_start(orc_rcd* d)                                                 /* :249-262 */
{
	if (d->n - d->pos < 8) return;
	d->buffer = 0;
	for (int i = 1; i <= 8; ++i) d->buffer |= (uint64_t)orc_rcd_byte(d) << (64 - 8 * i);
	d->low = 0; d->range = ORC_MASK;
}
static inline uint64_t orc_rcd_cum(orc_rcd* d, uint64_t tot) { return d->buffer / (d->range /= tot); }   /* :264-268 */
static inline void orc_rcd_update(orc_rcd* d, uint64_t freq, uint64_t cum)                   /* :270-287 */
{
	uint64_t r = cum * d->range;
	d->buffer -= r; d->low += r; d->range *= freq;
	while (d->range <= ORC_TOP)
	{
		if ((d->low ^ (d->low + d->range)) & ORC_MASK)
		{
			uint64_t q = d->low;
			d->range = (q | ORC_TOP) - q;
		}
		d->buffer = (d->buffer << 8) + orc_rcd_byte(d);
		d->low <<= 8; d->range <<= 8;
	}
}

/* ---- adaptive model (rc.h:34-220 / 225-480 / 487-764) ---- */
typedef struct { uint32_t* stats; uint32_t total; } orc_model;
static inline void orc_model_rescale(orc_model* m, uint32_t n, uint32_t max_total)           /* rc.h:41-52 */
{
	while (m->total >= max_total)
	{
		m->total = 0;
		for (uint32_t i = 0; i < n; ++i) { m->stats[i] = (m->stats[i] + 1) / 2; m->total += m->stats[i]; }
	}
}
static inline void orc_model_update(orc_model* m, uint32_t n, uint32_t sym, uint32_t max_total, uint32_t adder)   /* rc.h:178-185 */
{
	m->stats[sym] += adder; m->total += adder;
	if (m->total >= max_total) orc_model_rescale(m, n, max_total);
}

/* ---- context -> model container: only identity semantics matter (context_hm.h:34-417) ---- */
typedef struct {
	uint32_t n_sym, max_total, adder;
	uint64_t* keys; uint32_t* vals; size_t tsz, used;   /* open addressing, key ~0 = empty */
	uint32_t* pool; size_t n_models, cap_models;        /* stats of model i at pool[i*n_sym], total at totals[i] */
	uint32_t* totals;
} orc_ctxmap;
void orc_ctxmap_init(orc_ctxmap* c, uint32_t n_sym, uint32_t max_total, uint32_t adder);
void orc_ctxmap_free(orc_ctxmap* c);
orc_model orc_ctxmap_get(orc_ctxmap* c, uint64_t ctx, uint32_t** total_slot);   /* creates an all-ones model on first use */

/* encode / decode one symbol in context ctx, optionally excluding up to two symbols (rc.h:810-842,880-925) */
void orc_encode_sym(orc_rce* e, orc_ctxmap* c, uint64_t ctx, uint32_t sym, int exc1, int exc2);
uint32_t orc_decode_sym(orc_rcd* d, orc_ctxmap* c, uint64_t ctx, int exc1, int exc2);
#endif
