/* rc.c — oracle: context->model container and symbol coding with exclusions (see rc.h). */
#include "rc.h"

static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

void orc_ctxmap_init(orc_ctxmap* c, uint32_t n_sym, uint32_t max_total, uint32_t adder)
{
	memset(c, 0, sizeof(*c));
	c->n_sym = n_sym; c->max_total = max_total; c->adder = adder;
	c->tsz = 1024; c->keys = (uint64_t*)malloc(c->tsz * 8); c->vals = (uint32_t*)malloc(c->tsz * 4);
	memset(c->keys, 0xff, c->tsz * 8);
}
void orc_ctxmap_free(orc_ctxmap* c) { free(c->keys); free(c->vals); free(c->pool); free(c->totals); memset(c, 0, sizeof(*c)); }

static void ctxmap_grow(orc_ctxmap* c)
{
	uint64_t* ok = c->keys; uint32_t* ov = c->vals; size_t osz = c->tsz;
	c->tsz *= 2; c->keys = (uint64_t*)malloc(c->tsz * 8); c->vals = (uint32_t*)malloc(c->tsz * 4);
	memset(c->keys, 0xff, c->tsz * 8);
	for (size_t i = 0; i < osz; ++i) if (ok[i] != ~0ULL)
	{
		size_t h = mix64(ok[i]) & (c->tsz - 1);
		while (c->keys[h] != ~0ULL) h = (h + 1) & (c->tsz - 1);
		c->keys[h] = ok[i]; c->vals[h] = ov[i];
	}
	free(ok); free(ov);
}

/* find_rc_context / find_rce_context (basic_coder.h:116-137): a context seen for the first time gets a
 * copy of the template model; every template in the coders is the all-ones model (Init(nullptr)). */
orc_model orc_ctxmap_get(orc_ctxmap* c, uint64_t ctx, uint32_t** total_slot)
{
	size_t h = mix64(ctx) & (c->tsz - 1);
	while (c->keys[h] != ~0ULL && c->keys[h] != ctx) h = (h + 1) & (c->tsz - 1);
	uint32_t idx;
	if (c->keys[h] == ~0ULL)
	{
		if (c->n_models == c->cap_models)
		{
			c->cap_models = c->cap_models ? 2 * c->cap_models : 256;
			c->pool = (uint32_t*)realloc(c->pool, c->cap_models * c->n_sym * 4);
			c->totals = (uint32_t*)realloc(c->totals, c->cap_models * 4);
		}
		idx = (uint32_t)c->n_models++;
		for (uint32_t i = 0; i < c->n_sym; ++i) c->pool[(size_t)idx * c->n_sym + i] = 1;
		c->totals[idx] = c->n_sym;
		c->keys[h] = ctx; c->vals[h] = idx; ++c->used;
		if (c->used * 2 > c->tsz) ctxmap_grow(c);
	}
	else idx = c->vals[h];
	orc_model m; m.stats = c->pool + (size_t)idx * c->n_sym; m.total = c->totals[idx];
	*total_slot = &c->totals[idx];
	return m;
}

/* CRangeCoderModel*::Encode / EncodeExcluding (rc.h:810-842, 880-925): cumulative and total skip the
 * excluded symbols; the update is the ordinary one. */
void orc_encode_sym(orc_rce* e, orc_ctxmap* c, uint64_t ctx, uint32_t sym, int exc1, int exc2)
{
	uint32_t* tslot; orc_model m = orc_ctxmap_get(c, ctx, &tslot);
	uint32_t cum = 0, tot = m.total;
	for (uint32_t i = 0; i < sym; ++i) if ((int)i != exc1 && (int)i != exc2) cum += m.stats[i];
	if (exc1 >= 0) tot -= m.stats[exc1];
	if (exc2 >= 0) tot -= m.stats[exc2];
	orc_rce_encode(e, m.stats[sym], cum, tot);
	orc_model_update(&m, c->n_sym, sym, c->max_total, c->adder);
	*tslot = m.total;
}
uint32_t orc_decode_sym(orc_rcd* d, orc_ctxmap* c, uint64_t ctx, int exc1, int exc2)
{
	uint32_t* tslot; orc_model m = orc_ctxmap_get(c, ctx, &tslot);
	uint32_t tot = m.total;
	if (exc1 >= 0) tot -= m.stats[exc1];
	if (exc2 >= 0) tot -= m.stats[exc2];
	uint32_t target = (uint32_t)orc_rcd_cum(d, tot);
	uint32_t t = 0, sym = 0, cum = 0;
	for (uint32_t i = 0; i < c->n_sym; ++i)
	{
		if ((int)i != exc1 && (int)i != exc2) t += m.stats[i];
		if (t > target) { sym = i; cum = t - m.stats[i]; break; }
	}
	orc_rcd_update(d, m.stats[sym], cum);
	orc_model_update(&m, c->n_sym, sym, c->max_total, c->adder);
	*tslot = m.total;
	return sym;
}
