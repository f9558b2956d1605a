/* kmer_graph.c — oracle restatement of stages a1–a7 (see oracle.h; TEST INFRASTRUCTURE ONLY). */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* hash_filter.h:8-16 == filter_kmers.cpp:24-32 == murmur64_hash.h:65-75 (MurmurHash3 fmix64) */
uint64_t orc_hash_mm(uint64_t x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
	x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
	x ^= x >> 33;
	return x;
}

/* Rolling canonical k-mer (in_reads.h:30-74) restarted after every N (splitter.cpp:569-602). */
size_t orc_kmer_scan(const uint8_t* b, size_t len, uint32_t k, uint32_t f, uint64_t* out, size_t cap)
{
	const uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
	const uint32_t rev_off = 2 * (k - 1);
	uint64_t fwd = 0, rev = 0;
	uint32_t run = 0;           /* number of consecutive non-N bases ending here */
	size_t n = 0;
	for (size_t i = 0; i < len; ++i)
	{
		uint8_t c = b[i];
		if (c > 3) { run = 0; fwd = rev = 0; continue; }
		fwd = ((fwd << 2) | c) & mask;
		rev = (rev >> 2) | ((uint64_t)(3 - c) << rev_off);
		if (++run >= k)
		{
			uint64_t can = fwd < rev ? fwd : rev;
			if (orc_hash_mm(can) % f == 0)
			{
				if (out && n < cap) out[n] = can;
				++n;
			}
		}
	}
	return n;
}

static int cmp_u64(const void* a, const void* b)
{
	uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
	return x < y ? -1 : x > y;
}

/* kb_sorter.h:1000-1060: per distinct k-mer: n_total += count, ++n_unique, below ci -> dropped;
 * stored counter saturates at cs ("-cs"), upper cutoff cx stays at 1e9 (kmer_counter.cpp:559-601). */
size_t orc_count_filter(uint64_t* kmers, size_t n, uint32_t ci, uint32_t cs,
                        uint64_t* keys, uint32_t* counts, orc_kmer_stats* st)
{
	qsort(kmers, n, sizeof(uint64_t), cmp_u64);
	size_t m = 0; uint64_t uniq = 0, filt = 0;
	for (size_t i = 0; i < n;)
	{
		size_t j = i + 1;
		while (j < n && kmers[j] == kmers[i]) ++j;
		uint64_t c = j - i;
		++uniq;
		if (c >= ci)
		{
			uint32_t sc = c > cs ? cs : (uint32_t)c;
			keys[m] = kmers[i]; counts[m] = sc; ++m; filt += sc;
		}
		i = j;
	}
	if (st) { st->tot_kmers = n; st->n_unique = uniq; st->n_unique_counted = m; st->total_count_filtered = filt; }
	return m;
}

static int kept_has(const uint64_t* kept, size_t n, uint64_t x)
{
	size_t lo = 0, hi = n;
	while (lo < hi) { size_t mid = (lo + hi) >> 1; if (kept[mid] < x) lo = mid + 1; else hi = mid; }
	return lo < n && kept[lo] == x;
}

/* reads_sim_graph.cpp:128-169.  The reference dedups with an open-addressing set (insert_fast returns
 * false for a k-mer already seen in this read, *whether or not* it is in the kept set); only the set
 * semantics matter, so a small private linear-probing table is used here. */
size_t orc_accepted_kmers(const uint8_t* b, size_t len, uint32_t k, uint32_t f,
                          const uint64_t* kept, size_t n_kept, uint64_t* out, size_t cap)
{
	if (len < k) return 0;
	for (size_t i = 0; i < len; ++i) if (b[i] > 3) return 0;          /* hasN: whole read skipped (:142-143) */
	size_t tsz = 16; while (tsz < 3 * (len - k + 1)) tsz <<= 1;
	uint64_t* tab = (uint64_t*)malloc(tsz * sizeof(uint64_t));
	memset(tab, 0xff, tsz * sizeof(uint64_t));
	const uint64_t mask = (1ULL << (2 * k)) - 1; const uint32_t rev_off = 2 * (k - 1);
	uint64_t fwd = 0, rev = 0; size_t n = 0;
	for (size_t i = 0; i < len; ++i)
	{
		uint8_t c = b[i];
		fwd = ((fwd << 2) | c) & mask;
		rev = (rev >> 2) | ((uint64_t)(3 - c) << rev_off);
		if (i + 1 < k) continue;
		uint64_t can = fwd < rev ? fwd : rev;
		if (orc_hash_mm(can) % f) continue;                              /* Possible() (:156) */
		size_t h = orc_hash_mm(can) & (tsz - 1); int seen = 0;
		while (tab[h] != ~0ULL) { if (tab[h] == can) { seen = 1; break; } h = (h + 1) & (tsz - 1); }
		if (seen) continue;
		tab[h] = can;
		if (kept_has(kept, n_kept, can)) { if (out && n < cap) out[n] = can; ++n; }
	}
	free(tab);
	return n;
}

/* ---- a6 --------------------------------------------------------------------------------------- */
/* std::mt19937 (32-bit Mersenne twister, default seed 5489u) */
typedef struct { uint32_t s[624]; int idx; } mt_t;
static void mt_seed(mt_t* m, uint32_t seed)
{
	m->s[0] = seed;
	for (int i = 1; i < 624; ++i) m->s[i] = 1812433253u * (m->s[i - 1] ^ (m->s[i - 1] >> 30)) + (uint32_t)i;
	m->idx = 624;
}
static uint32_t mt_next(mt_t* m)
{
	if (m->idx >= 624)
	{
		for (int i = 0; i < 624; ++i)
		{
			uint32_t y = (m->s[i] & 0x80000000u) | (m->s[(i + 1) % 624] & 0x7fffffffu);
			m->s[i] = m->s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
		}
		m->idx = 0;
	}
	uint32_t y = m->s[m->idx++];
	y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
	return y;
}
/* libstdc++ generate_canonical<double,53>(mt19937): two draws, sum = d0 + d1*2^32 (double), / 2^64 */
static double mt_canonical(mt_t* m)
{
	double d0 = (double)mt_next(m);
	double d1 = (double)mt_next(m);
	double r = (d0 + d1 * 4294967296.0) / 18446744073709551616.0;
	if (r >= 1.0) r = nextafter(1.0, 0.0);
	return r;
}
void orc_ref_accept(uint32_t n_reads, uint32_t n_pseudo, uint32_t range, double exponent, uint8_t* out)
{
	mt_t m; mt_seed(&m, 5489u);
	for (uint32_t i = 0; i < n_reads + n_pseudo; ++i)
	{
		if (i < n_pseudo) { out[i] = 1; continue; }                          /* ref_reads_accepter.h:53-54 */
		uint32_t range_no = (i - n_pseudo) / range;
		double p = pow(1.0 / ((unsigned long)range_no + 1ul), exponent);    /* :56 */
		out[i] = mt_canonical(&m) <= p;
	}
}

/* ---- a5 ---------------------------------------------------------------------------------------
 * Sequential restatement: a k-mer -> growing list of reference ids.  The reference's container is a
 * prefix-partitioned compact multimap (hm_compact.h:545-579,872-918); only multimap semantics matter:
 * find() yields every value inserted for the key, in insertion order. */
typedef struct { uint64_t key; uint32_t* v; uint32_t n, cap; } kl_t;
struct orc_graph {
	kl_t* tab; size_t tsz, used;
	uint32_t max_cand, max_kc, n_refs;
	/* scratch vote map */
	uint32_t* vkey; uint32_t* vcnt; uint32_t* vfirst; size_t vsz, vused;
};
orc_graph* orc_graph_new(uint32_t max_candidates, uint32_t max_kmer_count)
{
	orc_graph* g = (orc_graph*)calloc(1, sizeof(*g));
	g->tsz = 1024; g->tab = (kl_t*)calloc(g->tsz, sizeof(kl_t));
	for (size_t i = 0; i < g->tsz; ++i) g->tab[i].key = ~0ULL;
	g->max_cand = max_candidates; g->max_kc = max_kmer_count;
	g->vsz = 1024; g->vkey = (uint32_t*)malloc(g->vsz * 4); g->vcnt = (uint32_t*)malloc(g->vsz * 4); g->vfirst = (uint32_t*)malloc(g->vsz * 4);
	return g;
}
void orc_graph_free(orc_graph* g)
{
	if (!g) return;
	for (size_t i = 0; i < g->tsz; ++i) free(g->tab[i].v);
	free(g->tab); free(g->vkey); free(g->vcnt); free(g->vfirst); free(g);
}
uint32_t orc_graph_n_refs(const orc_graph* g) { return g->n_refs; }

static kl_t* g_slot(orc_graph* g, uint64_t key)
{
	size_t h = orc_hash_mm(key) & (g->tsz - 1);
	while (g->tab[h].key != ~0ULL && g->tab[h].key != key) h = (h + 1) & (g->tsz - 1);
	return &g->tab[h];
}
static void g_grow(orc_graph* g)
{
	kl_t* old = g->tab; size_t osz = g->tsz;
	g->tsz *= 2; g->tab = (kl_t*)calloc(g->tsz, sizeof(kl_t));
	for (size_t i = 0; i < g->tsz; ++i) g->tab[i].key = ~0ULL;
	for (size_t i = 0; i < osz; ++i) if (old[i].key != ~0ULL) *g_slot(g, old[i].key) = old[i];
	free(old);
}
static void g_insert(orc_graph* g, uint64_t key, uint32_t val)
{
	if ((g->used + 1) * 2 > g->tsz) g_grow(g);
	kl_t* s = g_slot(g, key);
	if (s->key == ~0ULL) { s->key = key; ++g->used; }
	if (s->n == s->cap) { s->cap = s->cap ? 2 * s->cap : 4; s->v = (uint32_t*)realloc(s->v, s->cap * 4); }
	s->v[s->n++] = val;
}
void orc_graph_add_pseudo(orc_graph* g, const uint64_t* kmers, size_t n)
{
	uint32_t id = g->n_refs++;                                  /* :306-307 */
	for (size_t i = 0; i < n; ++i) g_insert(g, kmers[i], id);    /* uncapped (:314-317) */
}
static void v_reset(orc_graph* g, size_t need)
{
	size_t want = 64; while (want < 2 * need + 2) want <<= 1;
	if (want > g->vsz)
	{
		g->vsz = want;
		g->vkey = (uint32_t*)realloc(g->vkey, want * 4); g->vcnt = (uint32_t*)realloc(g->vcnt, want * 4); g->vfirst = (uint32_t*)realloc(g->vfirst, want * 4);
	}
	memset(g->vkey, 0xff, g->vsz * 4); g->vused = 0;
}
typedef struct { uint32_t id, votes; } cand_t;
static int cmp_cand(const void* a, const void* b)
{
	const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
	if (x->votes != y->votes) return x->votes > y->votes ? -1 : 1;     /* :411-415 votes desc */
	return x->id < y->id ? -1 : x->id > y->id;                          /* then id asc */
}
uint32_t orc_graph_next_read(orc_graph* g, const uint64_t* kmers, size_t n, int accept,
                             uint32_t* out_refs, uint32_t* out_votes, uint64_t* out_common, uint32_t* out_common_off)
{
	uint32_t my_id = 0;
	if (accept) my_id = g->n_refs++;                                    /* :343-347 */
	/* upper bound on distinct neighbours: sum of list lengths */
	size_t tot = 0;
	for (size_t i = 0; i < n; ++i) { kl_t* s = g_slot(g, kmers[i]); if (s->key != ~0ULL) tot += s->n; }
	v_reset(g, tot);
	for (size_t i = 0; i < n; ++i)
	{
		kl_t* s = g_slot(g, kmers[i]);
		uint32_t card = 0;
		if (s->key != ~0ULL)
		{
			card = s->n;
			for (uint32_t j = 0; j < s->n; ++j)
			{
				uint32_t r = s->v[j];
				size_t h = (r * 2654435761u) & (g->vsz - 1);
				while (g->vkey[h] != 0xffffffffu && g->vkey[h] != r) h = (h + 1) & (g->vsz - 1);
				if (g->vkey[h] == 0xffffffffu) { g->vkey[h] = r; g->vcnt[h] = 0; ++g->vused; }
				++g->vcnt[h];
			}
		}
		if (accept && card < g->max_kc) g_insert(g, kmers[i], my_id);   /* :390-393 */
	}
	cand_t* c = (cand_t*)malloc((g->vused + 1) * sizeof(cand_t)); size_t nc = 0;
	for (size_t h = 0; h < g->vsz; ++h) if (g->vkey[h] != 0xffffffffu) { c[nc].id = g->vkey[h]; c[nc].votes = g->vcnt[h]; ++nc; }
	qsort(c, nc, sizeof(cand_t), cmp_cand);                              /* total order => partial_sort-equivalent */
	if (nc > g->max_cand) nc = g->max_cand;
	for (size_t i = 0; i < nc; ++i) { out_refs[i] = c[i].id; if (out_votes) out_votes[i] = c[i].votes; }
	if (out_common)
	{	/* HiFi (:473-486,518-522): per chosen candidate, the shared k-mers in read order.  The own
		 * k-mers inserted above carry my_id, which is never among the candidates. */
		uint32_t off = 0;
		for (size_t ci = 0; ci < nc; ++ci)
		{
			out_common_off[ci] = off;
			for (size_t i = 0; i < n; ++i)
			{
				kl_t* s = g_slot(g, kmers[i]);
				if (s->key == ~0ULL) continue;
				for (uint32_t j = 0; j < s->n; ++j) if (s->v[j] == c[ci].id) { out_common[off++] = kmers[i]; break; }
			}
		}
		out_common_off[nc] = off;
	}
	free(c);
	return (uint32_t)nc;
}

/* reference_reads.h:35-72: 4 bases per byte, first base in bits 7..6; last partial byte left aligned;
 * one trailing byte = len % 4. */
size_t orc_refread_compact(const uint8_t* b, size_t len, uint8_t* out)
{
	size_t nb = (len + 3) / 4;
	memset(out, 0, nb + 1);
	for (size_t i = 0; i < len; ++i) out[i >> 2] |= (uint8_t)((b[i] & 3) << (6 - 2 * (i & 3)));
	out[nb] = (uint8_t)(len % 4);
	return nb + 1;
}
