// encode_es.hip — edit-script encoding of reads against their anchored candidates (a10 + a11 + a12):
// CEncoder::AddEncodedReadWithCandidates / EncodePart / GetEditDist / EncodeWithEditScript /
// EncodeWithAlternativeRead / AdjustAnchors / StoreFrag / bigEditScriptToTuples (src/colord/encoder.cpp:778-868,
// 1255-1575), refactor_edit_script (edit_script.h:416-446,591-671), CEntropy / CEntropyEstimator
// (utils.h:700-1131).
//
// Three passes, because the only cross-read state of the reference encoder is the adaptive cost estimator, and it
// is consulted only for gaps shorter than minPartLenToConsiderAltRead, whose outcome never changes WHICH work
// exists (no recursion starts from a short gap):
//   pass 1 (one lane per read, all reads concurrently): walks the fragment tree of the read — anchors, gaps,
//           recursion into alternative references for rejected long gaps (static entropy test) — aligns every gap
//           (align_dev.hpp), canonicalises indels, and records an ordered item list plus, for every short gap, the
//           statistics the estimator needs;
//   pass 2 (one lane per reader pack, the estimator's reset domain, encoder.cpp:1677): replays the estimator over
//           the short gaps in encoding order and decides edit script vs literal;
//   pass 3 (one lane per read): interprets the item list with those decisions into the tuple stream (run-length
//           rules of singleEditScriptSymbolStore, segment headers of StoreFrag).
#include "common.hpp"
#include "objects.hpp"
#include "align_dev.hpp"
#include <algorithm>

struct cl_anchors;   // anchors.hip
extern "C" const uint32_t* cl_anchors_n_cands(const cl_anchors* a);
extern "C" const uint32_t* cl_anchors_cands(const cl_anchors* a);
extern "C" const uint64_t* cl_anchors_cand_offsets(const cl_anchors* a);
extern "C" const uint32_t* cl_anchors_data(const cl_anchors* a);

namespace {
struct ArenaV { const uint64_t* packed; const uint64_t* word_off; const uint32_t* lens; };
CL_DEV inline uint32_t arena_base_at(const ArenaV& A, uint64_t wb, uint32_t p) { return (uint32_t)(A.packed[wb + (p >> 5)] >> (62 - 2 * (p & 31))) & 3u; }
CL_DEV inline uint32_t ref_base_at(const ArenaV& R, uint32_t id, bool rev, uint32_t pos)
{
	const uint32_t len = R.lens[id];
	const uint32_t p = rev ? len - 1 - pos : pos;
	const uint32_t b = arena_base_at(R, R.word_off[id], p);
	return rev ? 3u - b : b;
}

struct EncCfg {
	uint32_t c, m, min_part_alt, max_rec; double cost_mult; uint32_t scale, max_ref_len;
};
struct AnchorsV { const uint32_t* n_cands; const uint32_t* cand; const uint64_t* cand_off; const uint32_t* data; };

// item kinds (top 4 bits of a 64-bit item)
enum : uint64_t { IT_RUN = 1, IT_GAP = 2, IT_STORE = 3, IT_STORE2 = 4, IT_BEGIN = 5, IT_END = 6 };
CL_DEV inline uint64_t it_run(char sym, uint32_t n) { return (IT_RUN << 60) | ((uint64_t)(uint8_t)sym << 32) | n; }
struct GapRec { uint32_t es_off, es_len, enc_start, ne, d_after, state, pend, d_before; };   // script = d_before x 'D' + es   // state: 0 edit script, 1 literal, 2 pending (estimator)
struct PendRec { uint32_t rd[12]; uint32_t len_cost; uint32_t pl[4]; uint32_t ref_len; };

struct ReadOut {          // per-read chunk bookkeeping (filled at the end of pass 1)
	uint64_t item_off, gap_off, pend_off, es_off;
	uint32_t n_items, n_gaps, n_pend, es_len;
	uint32_t dna[4];      // base counts (estimator LogRead)
	uint32_t plain;       // 1: stored plain (no candidates), 2: plain with N
	uint32_t pad;
};

struct Frame {
	uint32_t level, enc_off, enc_len, n_cands;
	uint32_t* a_len; uint32_t* a_pe; uint32_t* a_pr;      // anchors of all candidates, concatenated
	uint32_t* c_first; uint32_t* c_n; uint32_t* c_tot; uint32_t* c_ref; uint8_t* c_rev;
	uint32_t i, n_frag, anch, cur_ref, cur_enc, after_child, a_span;   // a_span: length of the anchor arrays (slots keep their offsets)
	uint64_t pool_mark;
};

struct Sink {            // lane-local growing outputs in the lane pool's tail region
	uint64_t* items; uint32_t n_items, cap_items;
	GapRec* gaps; uint32_t n_gaps, cap_gaps;
	PendRec* pend; uint32_t n_pend, cap_pend;
	char* es; uint32_t n_es, cap_es;
	bool overflow; uint32_t why;
	CL_DEV inline void item(uint64_t v) { if (n_items < cap_items) items[n_items++] = v; else { overflow = true; why |= 8u; } }
};

CL_DEV inline char mismatch_sym(uint32_t ref, uint32_t nw)            // utils.h:341-352
{
	const uint32_t rank = nw - (nw > ref ? 1u : 0u);                      // rank of the new base among the three others
	return (char)('X' + rank);
}
CL_DEV inline bool is_mismatch(char c) { return c == 'X' || c == 'Y' || c == 'Z'; }
CL_DEV inline char base_letter(uint32_t b) { return b == 0 ? 'A' : b == 1 ? 'C' : b == 2 ? 'G' : 'T'; }

// refactor_edit_script (edit_script.h:416-446,591-671)
CL_DEV inline void fix_in_range(char* es, uint64_t start, uint64_t end)
{
	if (end < start + 2) return;
	--end;
	for (;;)
	{
		while (start < end && es[start] == 'M') ++start;
		while (start < end && es[end] != 'M') --end;
		if (start == end) break;
		const char t = es[start]; es[start] = es[end]; es[end] = t;
	}
}
CL_DEV inline void refactor_es(const uint8_t* ref, const uint8_t* enc, char* s, uint32_t n)
{
	uint32_t st = 0, pos = 0, es_start = 0;
	for (uint32_t p = 0; p < n; ++p)
	{
		const char c = s[p];
		const bool mis = is_mismatch(c), ins = c == 'A' || c == 'C' || c == 'G' || c == 'T';
		if (ins || mis || ref[st] != ref[pos]) { fix_in_range(s, es_start, p); es_start = p; if (ins || mis) ++es_start; st = pos; }
		if (!ins) ++pos;
	}
	fix_in_range(s, es_start, n);
	st = 0; pos = 0; es_start = 0;
	for (uint32_t p = 0; p < n; ++p)
	{
		const char c = s[p];
		const bool mis = is_mismatch(c), del = c == 'D';
		if (del || mis || enc[st] != enc[pos]) { fix_in_range(s, es_start, p); es_start = p; if (del || mis) ++es_start; st = pos; }
		if (!del) ++pos;
	}
	fix_in_range(s, es_start, n);
}

// edit script of one gap into dst (capacity >= nr + ne + 2); returns its length.  GetEditDist (encoder.cpp:1255-1283)
// rbuf: the reference symbols the alignment can touch — the whole part for an inner gap, its first `use` symbols for
// the right flank, its LAST `use` symbols for the left flank (use = min(2*ne, nr)).
CL_DEV uint32_t gap_edit_script(LanePool& pool, const uint8_t* rbuf, uint32_t nr, const uint8_t* enc, uint32_t ne, uint32_t frag, uint32_t n_frag, char* dst, uint32_t* d_before)
{
	uint32_t n = 0;
	const uint8_t* ref = rbuf;
	*d_before = 0;
	if (nr == 0 || ne == 0)                                                    // get_edit_dist_on_seq_empty (edit_script.h:250-267)
	{
		if (nr == 0) for (uint32_t i = 0; i < ne; ++i) dst[n++] = base_letter(enc[i]);
		else *d_before = nr;
		return n;
	}
	const uint64_t mk = pool.mark();
	const bool inner = frag != 0 && frag != n_frag - 1;
	const uint32_t use = inner ? nr : (2 * ne < nr ? 2 * ne : nr);
	uint8_t* opsbuf = (uint8_t*)pool.alloc((uint64_t)use + ne + 16);
	uint8_t* r2 = (uint8_t*)pool.alloc(use + 16ull); uint8_t* e2 = (uint8_t*)pool.alloc(ne + 16ull);
	if (pool.overflow) { pool.release(mk); return 0; }
	OpsOut ops{ opsbuf, 0 };
	const bool left = frag == 0, right = !left && frag == n_frag - 1;
	if (!left && !right)
	{	// find_edit_dist_with_edlib_ex: query = ref, target = enc, global.  (The small-input DP of the reference has the
		// same move preference in this orientation, edit_script.h:156-239, so one code path covers both.)
		const uint32_t best = nw_distance(pool, Seq{ ref, 1 }, nr, Seq{ enc, 1 }, ne);
		nw_path(pool, ref, nr, enc, ne, best, ops);
		uint32_t pr = 0, pe = 0;
		for (uint64_t i = 0; i < ops.n; ++i)
			switch (ops.p[i])
			{
			case 0: dst[n++] = 'M'; ++pr; ++pe; break;
			case 1: dst[n++] = 'D'; ++pr; break;
			case 2: dst[n++] = base_letter(enc[pe++]); break;
			default: dst[n++] = mismatch_sym(ref[pr], enc[pe]); ++pr; ++pe;
			}
		refactor_es(ref, enc, dst, n);
		pool.release(mk);
		return n;
	}
	// flanks: prefix-free alignment of the whole read part against <= 2*|enc| reference symbols
	if (left) { for (uint32_t i = 0; i < use; ++i) r2[i] = rbuf[use - 1 - i]; for (uint32_t i = 0; i < ne; ++i) e2[i] = enc[ne - 1 - i]; }
	else { for (uint32_t i = 0; i < use; ++i) r2[i] = ref[i]; for (uint32_t i = 0; i < ne; ++i) e2[i] = enc[i]; }
	uint32_t ref_end;
	if (use < 2 || ne < 2)
	{	// find_edit_dist (edit_script.h:156-239): global, rows = ref: prefers deleting a reference symbol
		const uint32_t best = nw_distance(pool, Seq{ r2, 1 }, use, Seq{ e2, 1 }, ne);
		nw_path(pool, r2, use, e2, ne, best, ops);
		uint32_t pr = 0, pe = 0;
		for (uint64_t i = 0; i < ops.n; ++i)
			switch (ops.p[i])
			{
			case 0: dst[n++] = 'M'; ++pr; ++pe; break;
			case 1: dst[n++] = 'D'; ++pr; break;
			case 2: dst[n++] = base_letter(e2[pe++]); break;
			default: dst[n++] = mismatch_sym(r2[pr], e2[pe]); ++pr; ++pe;
			}
		ref_end = use - 1;
	}
	else
	{	// edlib SHW, query = enc, target = ref (edit_script.h:341-400)
		uint32_t best; int64_t end;
		shw_distance(pool, Seq{ e2, 1 }, ne, Seq{ r2, 1 }, use, &best, &end);
		ref_end = (uint32_t)end;
		nw_path(pool, e2, ne, r2, (uint32_t)(end + 1), best, ops);
		uint32_t pr = 0, pe = 0;
		for (uint64_t i = 0; i < ops.n; ++i)
			switch (ops.p[i])
			{
			case 0: dst[n++] = 'M'; ++pr; ++pe; break;
			case 1: dst[n++] = base_letter(e2[pe++]); break;
			case 2: dst[n++] = 'D'; ++pr; break;
			default: dst[n++] = mismatch_sym(r2[pr], e2[pe]); ++pr; ++pe;
			}
	}
	if (left)
	{	// find_edit_dist_with_edlib_ex_odwr_reverse (edit_script.h:405-419) + the D prefix (encoder.cpp:1263-1269)
		for (uint32_t a = 0, b = n; a + 1 < b; ++a) { --b; const char t = dst[a]; dst[a] = dst[b]; dst[b] = t; }
		const uint32_t ref_offset = (nr - 1) - ref_end;                       // uint32 wrap for end = -1, as in the reference
		refactor_es(rbuf + (ref_offset - (nr - use)), enc, dst, n);      // = ref part + ref_offset
		*d_before = ref_offset;
	}
	else refactor_es(ref, enc, dst, n);
	pool.release(mk);
	return n;
}

// static entropies (utils.h:706-752)
CL_DEV inline double entropy_dna_dev(const uint8_t* s, uint32_t n)
{
	uint32_t h[4] = { 0, 0, 0, 0 };
	for (uint32_t i = 0; i < n; ++i) ++h[s[i]];
	double sum = 0; for (int i = 0; i < 4; ++i) sum += h[i];
	const double rec = 1.0 / sum; double e = 0;
	for (int c = 0; c < 4; ++c) if (h[c]) { const double p = (double)h[c] * rec; e += log2(p) * p; }
	return -e;
}
CL_DEV inline uint32_t es_class(char c)       // order of CEntropy::es_sym = A C D G M T X Y Z (S, R never occur in a script)
{
	switch (c) { case 'A': return 0; case 'C': return 1; case 'D': return 2; case 'G': return 3; case 'M': return 4; case 'T': return 5; case 'X': return 6; case 'Y': return 7; default: return 8; }
}
CL_DEV inline double entropy_es_dev(const char* s, uint32_t n, uint32_t extra_d)
{
	uint32_t h[9] = { 0, 0, extra_d, 0, 0, 0, 0, 0, 0 };
	for (uint32_t i = 0; i < n; ++i) ++h[es_class(s[i])];
	double sum = 0; for (int i = 0; i < 9; ++i) sum += h[i];
	const double rec = 1.0 / sum; double e = 0;
	for (int c = 0; c < 9; ++c) if (h[c]) { const double p = (double)h[c] * rec; e += log2(p) * p; }
	return -e;
}
CL_DEV inline uint32_t bitlen32(uint64_t x) { return x ? 64u - (uint32_t)__builtin_clzll(x) : 0u; }
// estimator alphabet (utils.h:914-930): A C G T D M X Y Z S R
CL_DEV inline uint32_t est_code(char c)
{
	switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'D': return 4; case 'M': return 5; case 'X': return 6; case 'Y': return 7; case 'Z': return 8; default: return 11; }
}
// analyze_es (utils.h:819-874) reduced to what EncodeWithEditScript consumes
CL_DEV inline void analyze_es_dev(const char* es, uint32_t n, uint32_t d_before, PendRec& p)
{
	for (int i = 0; i < 12; ++i) p.rd[i] = 0;
	p.len_cost = 0;
	char c = d_before ? 'D' : ' '; uint32_t len = d_before;
	for (uint32_t i = 0; i <= n; ++i)
	{
		const char x = i < n ? es[i] : ' ';
		if (x == c) { ++len; continue; }
		if (c == 'D') { if (len >= 10) { ++p.rd[9]; p.len_cost += bitlen32(len) + 1; } else p.rd[4] += len; }
		else if (c == 'M') { if (len >= 15) { ++p.rd[10]; p.len_cost += bitlen32(len) + 1; } else p.rd[5] += len; }
		else if (c != ' ') ++p.rd[est_code(c)];
		c = x; len = 1;
	}
}

// AdjustAnchors (encoder.cpp:778-868) on one candidate's anchors (arrays of length *n), returns total length
CL_DEV uint32_t adjust_anchors(uint32_t* al, uint32_t* ape, uint32_t* apr, uint32_t* n_io, uint32_t ns, uint32_t ne, uint32_t m)
{
	uint32_t n = *n_io; const uint32_t G = 0xffffffffu; uint32_t first = G, last = G, tot = 0;
	for (uint32_t i = 0; i < n; ++i) if (ape[i] + al[i] > ns) { first = i; break; }
	if (first == G) { *n_io = 0; return 0; }
	if (ape[first] < ns && (ape[first] + al[first]) - ns < m) ++first;
	for (int32_t i = (int32_t)n - 1; i >= 0; --i) if (ape[i] < ne) { last = (uint32_t)i; break; }
	if (last == G) { *n_io = 0; return 0; }
	if (last < n && ape[last] + al[last] > ne && ne - ape[last] < m) { if (last == 0) { *n_io = 0; return 0; } --last; }
	if (first > last) { *n_io = 0; return 0; }
	n = last + 1 - first;
	for (uint32_t i = 0; i < n; ++i) { al[i] = al[first + i]; ape[i] = ape[first + i]; apr[i] = apr[first + i]; }
	if (!n) { *n_io = 0; return 0; }
	if (ape[n - 1] + al[n - 1] > ne) al[n - 1] -= (ape[n - 1] + al[n - 1] - ne);
	for (uint32_t i = 0; i < n; ++i)
	{
		if (i == 0 && ape[0] < ns) { const uint32_t d = ns - ape[0]; al[0] -= d; ape[0] = 0; apr[0] += d; }
		else ape[i] -= ns;
		tot += al[i];
	}
	*n_io = n;
	return tot;
}

// ---- pass 1 ---------------------------------------------------------------------------------------------------
// One read: fills ro (always) and the lane-local sink (when the read has candidates).  Returns true when the sink
// holds the read's items; false for plain reads or when the lane's pool / sink ran out (pool.overflow / sk.overflow).
CL_DEV bool expand_read(LanePool& pool, const ArenaV& A, const ArenaV& R, const AnchorsV& AV, const EncCfg& cfg, const uint8_t* has_n, uint32_t r, ReadOut& ro, Sink& sk)
{
	memset(&ro, 0, sizeof(ro));
	const uint32_t len = A.lens[r]; const uint64_t wb = A.word_off[r];
	if (has_n[r]) { ro.plain = 2; return false; }
	for (uint32_t i = 0; i < len; ++i) ++ro.dna[arena_base_at(A, wb, i)];
	const uint32_t nc = AV.n_cands[r];
	if (nc == 0) { ro.plain = 1; return false; }
	// lane-local output areas (generous: a read's scripts cannot exceed a few times its length)
	memset(&sk, 0, sizeof(sk));
	uint32_t tot_anch = 0;
	for (uint32_t j = 0; j < nc; ++j) tot_anch += AV.cand[((uint64_t)r * cfg.c + j) * 4 + 3];
	uint32_t max_anch = 0;
	for (uint32_t j = 0; j < nc; ++j) { const uint32_t a = AV.cand[((uint64_t)r * cfg.c + j) * 4 + 3]; max_anch = a > max_anch ? a : max_anch; }
	sk.cap_gaps = ((max_anch + 2) * (cfg.max_rec + 1) + len / 64 + 16) * cfg.scale;
	sk.cap_items = sk.cap_gaps * 4 + 64;
	sk.cap_pend = sk.cap_gaps;
	sk.cap_es = (4 * len + 4096) * cfg.scale;
	sk.items = (uint64_t*)pool.alloc((uint64_t)sk.cap_items * 8); sk.gaps = (GapRec*)pool.alloc((uint64_t)sk.cap_gaps * sizeof(GapRec));
	sk.pend = (PendRec*)pool.alloc((uint64_t)sk.cap_pend * sizeof(PendRec)); sk.es = (char*)pool.alloc(sk.cap_es);
	uint8_t* encb = (uint8_t*)pool.alloc(len + 16ull);
	Frame* frames = (Frame*)pool.alloc(sizeof(Frame) * 10);
	uint32_t* f0 = (uint32_t*)pool.alloc((tot_anch * 3ull + nc * 5ull + 16) * 4);
	int fp = 0;
	if (pool.overflow) fp = -1;
	else
	{	// level-0 frame: the candidates as the anchor stage delivered them
		for (uint32_t i = 0; i < len; ++i) encb[i] = (uint8_t)arena_base_at(A, wb, i);
		Frame& F = frames[0]; memset(&F, 0, sizeof(F));
		F.level = 0; F.enc_off = 0; F.enc_len = len; F.n_cands = nc; F.a_span = tot_anch;
		F.a_len = f0; F.a_pe = f0 + tot_anch; F.a_pr = f0 + 2ull * tot_anch;
		F.c_first = f0 + 3ull * tot_anch; F.c_n = F.c_first + nc; F.c_tot = F.c_n + nc; F.c_ref = F.c_tot + nc; F.c_rev = (uint8_t*)(F.c_ref + nc);
		uint32_t o = 0;
		for (uint32_t j = 0; j < nc; ++j)
		{
			const uint64_t s = (uint64_t)r * cfg.c + j;
			F.c_ref[j] = AV.cand[s * 4]; F.c_rev[j] = (uint8_t)AV.cand[s * 4 + 1]; F.c_tot[j] = AV.cand[s * 4 + 2]; F.c_n[j] = AV.cand[s * 4 + 3]; F.c_first[j] = o;
			const uint64_t a0 = AV.cand_off[s];
			for (uint32_t t = 0; t < F.c_n[j]; ++t, ++o) { F.a_len[o] = AV.data[3 * (a0 + t)]; F.a_pe[o] = AV.data[3 * (a0 + t) + 1]; F.a_pr[o] = AV.data[3 * (a0 + t) + 2]; }
		}
		F.n_frag = F.c_n[0] * 2 + 1;
		F.pool_mark = pool.mark();
		sk.item((IT_BEGIN << 60) | ((uint64_t)0 << 56) | ((uint64_t)F.c_rev[0] << 32) | F.c_ref[0]);
	}
	while (fp >= 0 && !pool.overflow && !sk.overflow)
	{
		Frame& F = frames[fp];
		const uint32_t lv = F.level;
		const uint32_t ref_id = F.c_ref[lv]; const bool rev = F.c_rev[lv] != 0;
		const uint32_t ref_len = R.lens[ref_id];
		if (F.i == F.n_frag)
		{	// end of AddEncodedReadWithCandidates: final StoreFrag (encoder.cpp:1574)
			sk.item((IT_STORE << 60) | ((uint64_t)lv << 56) | ((uint64_t)(rev ? 1 : 0) << 32) | ref_id);
			sk.item((IT_STORE2 << 60) | F.cur_ref);
			sk.item(IT_END << 60);
			pool.release(F.pool_mark);
			--fp;
			if (fp >= 0)
			{
				Frame& Pf = frames[fp];
				pool.release(Pf.pool_mark);
				if (Pf.after_child) sk.item(it_run('D', Pf.after_child));
				Pf.after_child = 0;
			}
			continue;
		}
		const uint32_t* al = F.a_len + F.c_first[lv]; const uint32_t* ape = F.a_pe + F.c_first[lv]; const uint32_t* apr = F.a_pr + F.c_first[lv];
		if (F.i & 1)
		{	// anchor
			sk.item(it_run('M', al[F.anch]));
			F.cur_ref = apr[F.anch] + al[F.anch]; F.cur_enc = ape[F.anch] + al[F.anch];
			++F.anch; ++F.i;
			continue;
		}
		// gap (EncodePart, encoder.cpp:1445-1511)
		const bool last_frag = F.i == F.n_frag - 1;
		const uint32_t end_enc = last_frag ? F.enc_len : ape[F.anch];
		const uint32_t end_ref = last_frag ? ref_len : apr[F.anch];
		const uint32_t want = end_ref - F.cur_ref, avail = ref_len - F.cur_ref;
		const uint32_t nr = want < avail ? want : avail;
		const uint32_t ne = end_enc - F.cur_enc;
		if (end_enc < F.cur_enc || end_enc > F.enc_len || F.cur_ref > ref_len) { sk.overflow = true; sk.why |= 64u; break; }   // inconsistent anchors: refuse rather than run away
		const uint64_t mk = pool.mark();
		const bool flank = F.i == 0 || last_frag;
		const uint32_t use = flank ? (2 * ne < nr ? 2 * ne : nr) : nr;          // reference symbols an alignment of this gap can touch
		const uint32_t lo = F.i == 0 ? nr - use : 0;
		uint8_t* refp = (uint8_t*)pool.alloc(use + 16ull);
		if (!pool.overflow) for (uint32_t i = 0; i < use; ++i) refp[i] = (uint8_t)ref_base_at(R, ref_id, rev, F.cur_ref + lo + i);
		const uint8_t* encp = encb + F.enc_off + F.cur_enc;
		if (pool.overflow || sk.n_es + (uint64_t)use + ne + 8 > sk.cap_es || sk.n_gaps >= sk.cap_gaps || sk.n_pend >= sk.cap_pend) {
#ifdef CL_HOST_DEBUG
			printf("read %u: sink full at level %u frag %u/%u: nr %u ne %u use %u n_es %u/%u gaps %u/%u pend %u/%u cur_ref %u cur_enc %u end_ref %u end_enc %u ref_len %u\n", r, lv, F.i, F.n_frag, nr, ne, use, sk.n_es, sk.cap_es, sk.n_gaps, sk.cap_gaps, sk.n_pend, sk.cap_pend, F.cur_ref, F.cur_enc, end_ref, end_enc, ref_len);
#endif
			sk.overflow = true; sk.why |= 16u; break;
		}
		char* es = sk.es + sk.n_es;
		uint32_t d_before;
		const uint32_t n_es = gap_edit_script(pool, refp, nr, encp, ne, F.i, F.n_frag, es, &d_before);
		GapRec g; memset(&g, 0, sizeof(g));
		g.es_off = sk.n_es; g.es_len = n_es; g.enc_start = F.enc_off + F.cur_enc; g.ne = ne; g.d_after = last_frag ? 0 : end_ref - F.cur_ref; g.d_before = d_before;
		sk.n_es += n_es;
		bool accept = false, pending = false;
		if (ne < cfg.min_part_alt)
		{	// adaptive estimator decides in pass 2
			pending = true;
			PendRec& p = sk.pend[sk.n_pend];
			analyze_es_dev(es, n_es, d_before, p);
			p.pl[0] = p.pl[1] = p.pl[2] = p.pl[3] = 0;
			for (uint32_t i = 0; i < ne; ++i) ++p.pl[encp[i]];
			p.ref_len = nr;
			g.pend = sk.n_pend++;
		}
		else
		{	// EncodeWithEditScript (encoder.cpp:1315-1327) with GetEditScriptEntropyInput (:1299-1311)
			uint32_t nd = 0; while (nd < n_es && es[nd] == 'D') ++nd;
			const char* p = es; uint32_t n = n_es, extra = d_before;
			if (nd + d_before >= 10) { p += nd; n -= nd; extra = 0; }
			accept = entropy_es_dev(p, n, extra) * (double)(n + extra) * cfg.cost_mult < entropy_dna_dev(encp, ne) * (double)ne;
		}
		pool.release(mk);
		if (pending || accept)
		{
			g.state = pending ? 2u : 0u;
			sk.gaps[sk.n_gaps] = g; sk.item((IT_GAP << 60) | sk.n_gaps); ++sk.n_gaps;
			++F.i;
			continue;
		}
		// rejected long gap: EncodeWithAlternativeRead (encoder.cpp:1329-1346)
		bool use_alt = false;
		if (!(F.n_cands <= lv + 1 || ne < cfg.min_part_alt || lv >= cfg.max_rec) && fp + 1 < 10)
		{
			Frame& C = frames[fp + 1]; memset(&C, 0, sizeof(C));
			F.pool_mark = pool.mark();
			const uint32_t nc2 = F.n_cands;
			const uint32_t tot_a = F.a_span;
			uint32_t* f1 = (uint32_t*)pool.alloc((tot_a * 3ull + nc2 * 5ull + 16) * 4);
			if (pool.overflow) break;
			C.a_len = f1; C.a_pe = f1 + tot_a; C.a_pr = f1 + 2ull * tot_a;
			C.c_first = f1 + 3ull * tot_a; C.c_n = C.c_first + nc2; C.c_tot = C.c_n + nc2; C.c_ref = C.c_tot + nc2; C.c_rev = (uint8_t*)(C.c_ref + nc2);
			for (uint32_t j = 0; j < nc2; ++j) { C.c_first[j] = F.c_first[j]; C.c_n[j] = F.c_n[j]; C.c_tot[j] = F.c_tot[j]; C.c_ref[j] = F.c_ref[j]; C.c_rev[j] = F.c_rev[j]; }
			for (uint32_t t = 0; t < tot_a; ++t) { C.a_len[t] = F.a_len[t]; C.a_pe[t] = F.a_pe[t]; C.a_pr[t] = F.a_pr[t]; }
			for (uint32_t j = lv + 1; j < nc2; ++j)
				C.c_tot[j] = adjust_anchors(C.a_len + C.c_first[j], C.a_pe + C.c_first[j], C.a_pr + C.c_first[j], &C.c_n[j], F.cur_enc, end_enc, cfg.m);
			// stable insertion sort of candidates lv+1.. by total anchor length (libstdc++ std::sort on <= 16 elements)
			for (uint32_t i = lv + 2; i < nc2; ++i)
			{
				const uint32_t xf = C.c_first[i], xn = C.c_n[i], xt = C.c_tot[i], xr = C.c_ref[i]; const uint8_t xv = C.c_rev[i];
				uint32_t j = i;
				while (j > lv + 1 && xt > C.c_tot[j - 1]) { C.c_first[j] = C.c_first[j - 1]; C.c_n[j] = C.c_n[j - 1]; C.c_tot[j] = C.c_tot[j - 1]; C.c_ref[j] = C.c_ref[j - 1]; C.c_rev[j] = C.c_rev[j - 1]; --j; }
				C.c_first[j] = xf; C.c_n[j] = xn; C.c_tot[j] = xt; C.c_ref[j] = xr; C.c_rev[j] = xv;
			}
			use_alt = C.c_tot[lv + 1] != 0;
			if (use_alt)
			{
				sk.item((IT_STORE << 60) | ((uint64_t)lv << 56) | ((uint64_t)(rev ? 1 : 0) << 32) | ref_id);
				sk.item((IT_STORE2 << 60) | F.cur_ref);
				C.a_span = tot_a; C.level = lv + 1; C.enc_off = F.enc_off + F.cur_enc; C.enc_len = ne; C.n_cands = nc2;
				C.n_frag = C.c_n[lv + 1] * 2 + 1;
				C.pool_mark = pool.mark();
				sk.item((IT_BEGIN << 60) | ((uint64_t)(lv + 1) << 56) | ((uint64_t)C.c_rev[lv + 1] << 32) | C.c_ref[lv + 1]);
				F.after_child = last_frag ? 0 : end_ref - F.cur_ref;
				++F.i;
				++fp;
				continue;
			}
			pool.release(F.pool_mark);
		}
		// literal: the read part as insertions, then the reference part skipped (encoder.cpp:1497-1508)
		g.state = 1;
		sk.gaps[sk.n_gaps] = g; sk.item((IT_GAP << 60) | sk.n_gaps); ++sk.n_gaps;
		++F.i;
	}
	return !(pool.overflow || sk.overflow);
}

__global__ __launch_bounds__(64) void k_encode_expand(ArenaV A, ArenaV R, AnchorsV AV, EncCfg cfg, const uint8_t* __restrict__ has_n, uint32_t n_reads,
                                                     const uint32_t* __restrict__ todo, uint32_t n_todo, uint32_t* __restrict__ redo, unsigned int* __restrict__ n_redo,
                                                     uint8_t* __restrict__ scratch, uint64_t scratch_per_lane, unsigned int* __restrict__ next_read,
                                                     ReadOut* __restrict__ rout, uint64_t* __restrict__ g_items, uint64_t cap_items, GapRec* __restrict__ g_gaps, uint64_t cap_gaps,
                                                     PendRec* __restrict__ g_pend, uint64_t cap_pend, char* __restrict__ g_es, uint64_t cap_es,
                                                     unsigned long long* __restrict__ counters /* items, gaps, pend, es */, uint32_t* __restrict__ err)
{
	const uint32_t lane_id_g = blockIdx.x * blockDim.x + threadIdx.x;
	LanePool pool{ scratch + (uint64_t)lane_id_g * scratch_per_lane, scratch_per_lane, 0, false, 0 };
	for (;;)
	{
		const uint32_t slot = atomicAdd(next_read, 1u);
		if (slot >= n_todo) break;
		const uint32_t r = todo ? todo[slot] : slot;
		pool.top = 0; pool.overflow = false; pool.why = 0;
		ReadOut ro; Sink sk;
		const bool have = expand_read(pool, A, R, AV, cfg, has_n, r, ro, sk);
		if (!have && !pool.overflow && !sk.overflow) { rout[r] = ro; continue; }
		if (pool.overflow || sk.overflow) { redo[atomicAdd(n_redo, 1u)] = r; atomicOr(err, 1u | ((pool.why | sk.why) << 8)); ro.plain = 1; rout[r] = ro; continue; }
		// publish the read's chunks
		ro.n_items = sk.n_items; ro.n_gaps = sk.n_gaps; ro.n_pend = sk.n_pend; ro.es_len = sk.n_es;
		ro.item_off = atomicAdd(&counters[0], (unsigned long long)sk.n_items); ro.gap_off = atomicAdd(&counters[1], (unsigned long long)sk.n_gaps);
		ro.pend_off = atomicAdd(&counters[2], (unsigned long long)sk.n_pend); ro.es_off = atomicAdd(&counters[3], (unsigned long long)sk.n_es);
		if (ro.item_off + sk.n_items > cap_items || ro.gap_off + sk.n_gaps > cap_gaps || ro.pend_off + sk.n_pend > cap_pend || ro.es_off + sk.n_es > cap_es)
		{ atomicOr(err, 2u); ro.plain = 1; ro.n_items = ro.n_gaps = ro.n_pend = ro.es_len = 0; rout[r] = ro; continue; }
		for (uint32_t i = 0; i < sk.n_items; ++i) g_items[ro.item_off + i] = sk.items[i];
		for (uint32_t i = 0; i < sk.n_gaps; ++i) g_gaps[ro.gap_off + i] = sk.gaps[i];
		for (uint32_t i = 0; i < sk.n_pend; ++i) g_pend[ro.pend_off + i] = sk.pend[i];
		for (uint32_t i = 0; i < sk.n_es; ++i) g_es[ro.es_off + i] = sk.es[i];
		rout[r] = ro;
	}
}

// ---- pass 2: the adaptive estimator, one lane per reader pack (utils.h:877-1130) ---------------------------------
struct Estim { uint32_t dna[4], es[12], dec[2]; double dna_logs[4], es_logs[12], dec_logs[2]; uint32_t dna_sum, es_sum, dec_sum; };
CL_DEV inline void est_rescale(uint32_t* a, int n, uint32_t& sum, uint32_t mx) { while (sum > mx) { sum = 0; for (int i = 0; i < n; ++i) { a[i] = (a[i] + 1) / 2; sum += a[i]; } } }
CL_DEV inline void est_logs(const uint32_t* st, double* lg, int n, uint32_t sum)
{
	const double rec = 1.0 / sum;
	for (int i = 0; i < n; ++i) lg[i] = st[i] ? -log2((double)st[i] * rec) : 0.0;
}
CL_DEV void estimate_pack(const ReadOut* rout, const PendRec* pend, uint32_t r_begin, uint32_t r_end, const uint32_t* lens, uint8_t* decisions)
{
	Estim e;
	for (int i = 0; i < 4; ++i) e.dna[i] = 1;
	e.dna_sum = 4;
	for (int i = 0; i < 12; ++i) e.es[i] = 1;
	e.es_sum = 12;
	e.dec[0] = e.dec[1] = 1; e.dec_sum = 2;
	est_logs(e.dna, e.dna_logs, 4, e.dna_sum); est_logs(e.es, e.es_logs, 12, e.es_sum); est_logs(e.dec, e.dec_logs, 2, e.dec_sum);
	for (uint32_t r = r_begin; r < r_end; ++r)
	{
		const ReadOut ro = rout[r];
		if (ro.plain == 2) continue;                                               // reads with N never reach the estimator (encoder.cpp:1629-1633)
		for (int i = 0; i < 4; ++i) e.dna[i] += ro.dna[i];                           // LogRead (utils.h:946-955)
		e.dna_sum += lens[r];
		est_rescale(e.dna, 4, e.dna_sum, 1u << 20);
		est_logs(e.dna, e.dna_logs, 4, e.dna_sum);
		for (uint32_t k = 0; k < ro.n_pend; ++k)
		{	// EncodeWithEditScript (utils.h:1060-1130)
			const PendRec p = pend[ro.pend_off + k];
			uint32_t loc[12]; uint32_t loc_sum = e.es_sum;
			for (int i = 0; i < 12; ++i) { loc[i] = e.es[i] + p.rd[i]; loc_sum += p.rd[i]; }
			double es_cost = e.dec_logs[0], plain_cost = e.dec_logs[1];
			est_logs(loc, e.es_logs, 12, loc_sum);
			for (int i = 0; i < 12; ++i) es_cost += p.rd[i] * e.es_logs[i];
			es_cost += p.len_cost;
			for (int i = 0; i < 4; ++i) plain_cost += p.pl[i] * e.dna_logs[i];
			plain_cost += bitlen32(p.ref_len) + 1;
			const bool choose_plain = plain_cost < es_cost;
			if (choose_plain) { ++e.dec[1]; est_rescale(e.es, 12, e.es_sum, 1u << 20); }
			else { ++e.dec[0]; for (int i = 0; i < 12; ++i) e.es[i] = loc[i]; e.es_sum = loc_sum; est_rescale(e.es, 12, e.es_sum, 1u << 20); }
			++e.dec_sum;
			est_rescale(e.dec, 2, e.dec_sum, 1u << 20);
			est_logs(e.dec, e.dec_logs, 2, e.dec_sum);
			decisions[ro.pend_off + k] = choose_plain ? 0 : 1;
		}
	}
}

__global__ void k_estimator(const ReadOut* __restrict__ rout, const PendRec* __restrict__ pend, const uint32_t* __restrict__ pack_bounds, uint32_t n_packs,
                            const uint32_t* __restrict__ lens, uint8_t* __restrict__ decisions)
{
	const uint32_t pk = blockIdx.x * blockDim.x + threadIdx.x;
	if (pk < n_packs) estimate_pack(rout, pend, pack_bounds[pk], pack_bounds[pk + 1], lens, decisions);
}

// ---- pass 3: tuple emission (encoder.cpp:1348-1443) ---------------------------------------------------------------
struct TupleOut {
	uint8_t* p; uint64_t n; uint32_t n_tuples; bool write;
	CL_DEV inline void byte(uint8_t v) { if (write) p[n] = v; ++n; }
	CL_DEV inline void t1(uint32_t type, uint32_t val) { byte((uint8_t)((type << 4) + val)); ++n_tuples; }
	CL_DEV inline void t28(uint32_t type, uint32_t v) { byte((uint8_t)((type << 4) + (v >> 24))); byte((v >> 16) & 0xff); byte((v >> 8) & 0xff); byte(v & 0xff); ++n_tuples; }
	CL_DEV inline void tid(uint32_t type, uint32_t id, uint32_t rev) { byte((uint8_t)((type << 4) + rev)); byte(id >> 24); byte((id >> 16) & 0xff); byte((id >> 8) & 0xff); byte(id & 0xff); ++n_tuples; }
};
struct RunState {
	char sym; uint32_t rep; TupleOut* o;
	CL_DEV inline void flush()
	{
		if (!rep) return;
		if (sym == 'M') { if (rep >= 15) o->t28(4, rep); else for (uint32_t i = 0; i < rep; ++i) o->t1(2, 0); }
		else if (sym == 'D') { if (rep > 16) o->t28(5, rep); else for (uint32_t i = 0; i < rep; ++i) o->t1(1, 0); }
		else if (sym == 'X' || sym == 'Y' || sym == 'Z') { for (uint32_t i = 0; i < rep; ++i) o->t1(3, (uint32_t)(sym - 'X')); }
		else { const uint32_t code = sym == 'A' ? 0 : sym == 'C' ? 1 : sym == 'G' ? 2 : 3; for (uint32_t i = 0; i < rep; ++i) o->t1(0, code); }
		rep = 0;
	}
	CL_DEV inline void add(char s, uint32_t n) { if (!n) return; if (rep && s == sym) { rep += n; return; } flush(); sym = s; rep = n; }
};

template<bool WRITE>
CL_DEV void emit_read(const ArenaV& A, const uint32_t* inv, uint32_t r, const ReadOut* rout, const uint64_t* g_items, const GapRec* g_gaps, const char* g_es,
                      const uint8_t* decisions, uint32_t* sizes, uint32_t* ntuples, const uint64_t* es_off, uint8_t* out)
{
	const ReadOut ro = rout[r];
	const uint32_t len = A.lens[r]; const uint64_t wb = A.word_off[r];
	TupleOut o{ WRITE ? out + es_off[r] : nullptr, 0, 0, WRITE };
	if (ro.plain)
	{	// AddPlainRead / AddPlainReadWithN (encoder.cpp:663-681)
		o.t1(ro.plain == 2 ? 11 : 9, 0);
		for (uint32_t i = 0; i < len; ++i)
		{
			const bool isn = (inv[wb + (i >> 5)] >> (31 - (i & 31))) & 1u;
			o.t1(8, isn ? 4u : arena_base_at(A, wb, i));
		}
		if (!WRITE) { sizes[r] = (uint32_t)o.n; ntuples[r] = o.n_tuples; }
		return;
	}
	const uint64_t* items = g_items + ro.item_off;
	const GapRec* gaps = g_gaps + ro.gap_off;
	const char* es = g_es + ro.es_off;
	auto gap_accepted = [&](const GapRec& g) -> bool { return g.state == 0 || (g.state == 2 && decisions[ro.pend_off + g.pend] != 0); };
	auto gap_len = [&](const GapRec& g) -> uint64_t { return gap_accepted(g) ? (uint64_t)g.d_before + g.es_len : (uint64_t)g.ne + g.d_after; };
	uint32_t last_pos[10]; int depth = -1;
	bool first = true; uint32_t main_id = 0;
	RunState rs{ 'M', 0, &o };
	uint32_t i = 0;
	while (i < ro.n_items)
	{
		const uint64_t it = items[i]; const uint64_t kind = it >> 60;
		if (kind == IT_BEGIN)
		{
			++depth; last_pos[depth] = 0;
			if (depth == 0) { main_id = (uint32_t)it; o.tid(10, main_id, (uint32_t)(it >> 32) & 1u); }     // start_es (encoder.cpp:1523-1527)
			++i; continue;
		}
		if (kind == IT_END) { --depth; ++i; continue; }
		// a segment: items up to and including the next IT_STORE/IT_STORE2 pair
		uint32_t j = i; bool nonempty = false;
		while ((items[j] >> 60) != IT_STORE)
		{
			const uint64_t x = items[j];
			if ((x >> 60) == IT_RUN) nonempty |= (uint32_t)x != 0;
			else nonempty |= gap_len(gaps[(uint32_t)x]) != 0;
			++j;
		}
		const uint64_t st = items[j], st2 = items[j + 1];
		const uint32_t level = (uint32_t)(st >> 56) & 15u, rev = (uint32_t)(st >> 32) & 1u, ref_id = (uint32_t)st, cur_ref = (uint32_t)st2;
		if (nonempty)
		{	// StoreFrag (encoder.cpp:1414-1443)
			if (level == 0) { if (ref_id != main_id) o.tid(6, ref_id, rev); else if (!first) o.t1(7, 0); }
			else { if (ref_id != main_id) o.tid(6, ref_id, rev); else o.t1(7, 0); rs.add('D', last_pos[depth]); }
			for (uint32_t t = i; t < j; ++t)
			{
				const uint64_t x = items[t];
				if ((x >> 60) == IT_RUN) { rs.add((char)((x >> 32) & 0xff), (uint32_t)x); continue; }
				const GapRec g = gaps[(uint32_t)x];
				if (gap_accepted(g)) { rs.add('D', g.d_before); for (uint32_t q = 0; q < g.es_len; ++q) rs.add(es[g.es_off + q], 1); }
				else
				{
					for (uint32_t q = 0; q < g.ne; ++q) rs.add(base_letter(arena_base_at(A, wb, g.enc_start + q)), 1);
					rs.add('D', g.d_after);
				}
			}
			rs.flush();
			last_pos[depth] = cur_ref;
			first = false;
		}
		i = j + 2;
	}
	if (!WRITE) { sizes[r] = (uint32_t)o.n; ntuples[r] = o.n_tuples; }
}
template<bool WRITE>
__global__ __launch_bounds__(64) void k_emit_tuples(ArenaV A, const uint32_t* __restrict__ inv, uint32_t n_reads, const ReadOut* __restrict__ rout,
                                                   const uint64_t* __restrict__ g_items, const GapRec* __restrict__ g_gaps, const char* __restrict__ g_es,
                                                   const uint8_t* __restrict__ decisions, uint32_t* __restrict__ sizes, uint32_t* __restrict__ ntuples,
                                                   const uint64_t* __restrict__ es_off, uint8_t* __restrict__ out)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r < n_reads) emit_read<WRITE>(A, inv, r, rout, g_items, g_gaps, g_es, decisions, sizes, ntuples, es_off, out);
}
} // namespace

// CEncoder::Encode for all reads of the arena (encoder.cpp:1672-1691).  h_pack_bounds: the reader packs (the
// estimator is reset at every pack).  Output: tuple streams (es_t bytes) back to back.
extern "C" cl_status cl_encode_reads(cl_ctx* ctx, const cl_reads* reads, const cl_reads* refs, const cl_anchors* anchors, uint32_t c, uint32_t anchor_len,
                                     uint32_t min_part_alt, uint32_t max_rec, double cost_mult, const uint32_t* h_pack_bounds, uint32_t n_packs,
                                     uint8_t* d_es, uint64_t cap, uint64_t* d_es_off, uint32_t* d_es_ntuples, uint64_t* n_out)
{
	if (!ctx || !reads || !refs || !anchors || !h_pack_bounds || !d_es_off || !d_es_ntuples || !n_out) return cl_fail(ctx, CL_E_INVALID, "cl_encode_reads: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const uint32_t nr = reads->n_reads;
	if (n_packs && (h_pack_bounds[0] != 0 || h_pack_bounds[n_packs] != nr)) return cl_fail(ctx, CL_E_INVALID, "cl_encode_reads: packs must cover the arena");
	if (max_rec > 8 || c > 16) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_encode_reads: max_rec <= 8 and c <= 16");
	*n_out = 0;
	if (!nr) return CL_OK;
	ArenaV A{ reads->packed.p, reads->word_off.p, reads->lens.p }, R{ refs->packed.p, refs->word_off.p, refs->lens.p };
	AnchorsV AV{ cl_anchors_n_cands(anchors), cl_anchors_cands(anchors), cl_anchors_cand_offsets(anchors), cl_anchors_data(anchors) };
	uint32_t maxlen = 0, max_ref = 0;
	{
		std::vector<uint32_t> h(std::max(nr, refs->n_reads));
		HIP_TRY(ctx, hipMemcpy(h.data(), reads->lens.p, (uint64_t)nr * 4, hipMemcpyDeviceToHost));
		for (uint32_t i = 0; i < nr; ++i) maxlen = std::max(maxlen, h[i]);
		if (refs->n_reads) HIP_TRY(ctx, hipMemcpy(h.data(), refs->lens.p, (uint64_t)refs->n_reads * 4, hipMemcpyDeviceToHost));
		for (uint32_t i = 0; i < refs->n_reads; ++i) max_ref = std::max(max_ref, h[i]);
	}
	const uint64_t cap_items = 16ull * nr + reads->total_bases / 4 + 1024, cap_gaps = cap_items / 2, cap_pend = cap_gaps, cap_es = 3 * reads->total_bases + 4096ull * 64 + 2ull * max_ref;
	DevBuf<ReadOut> rout; DEV_ALLOC(ctx, rout, nr);
	DevBuf<uint64_t> items; DEV_ALLOC(ctx, items, cap_items);
	DevBuf<GapRec> gaps; DEV_ALLOC(ctx, gaps, cap_gaps);
	DevBuf<PendRec> pend; DEV_ALLOC(ctx, pend, cap_pend);
	DevBuf<char> esb; DEV_ALLOC(ctx, esb, cap_es);
	DevBuf<unsigned long long> counters; DEV_ALLOC(ctx, counters, 4);
	DevBuf<uint32_t> err; DEV_ALLOC(ctx, err, 1);
	DevBuf<unsigned int> cnt; DEV_ALLOC(ctx, cnt, 2);                    // next slot, number of reads to redo
	DevBuf<uint32_t> todo, redo; DEV_ALLOC(ctx, todo, nr); DEV_ALLOC(ctx, redo, nr);
	HIP_TRY(ctx, hipMemsetAsync(counters.p, 0, 32, ctx->stream));
	// Pass 1 in rounds: many lanes with small pools first; the reads whose lane ran out of pool (long gaps need up to
	// 1 MiB of traceback state, edlib's own threshold, plus Hirschberg columns) are redone by fewer lanes with larger pools.
	uint32_t n_todo = nr; bool have_list = false;
	uint64_t per_lane = (256ull << 10) + 24ull * std::min<uint32_t>(maxlen, 65536);
	uint32_t n_lanes = 32768, scale = 1, last_err = 0;
	for (int round = 0; n_todo; ++round)
	{
		if (round == 4) return cl_fail(ctx, CL_E_NOMEM, "cl_encode_reads: lane pool exhausted after 4 rounds (reasons " + std::to_string(last_err >> 8) + ", " + std::to_string(n_todo) + " reads)");
		const uint32_t lanes = (uint32_t)std::min<uint64_t>(((uint64_t)n_todo + 63) / 64 * 64, n_lanes);
		DevBuf<uint8_t> scratch; DEV_ALLOC(ctx, scratch, per_lane * lanes);
		EncCfg cfg{ c, anchor_len, min_part_alt, max_rec, cost_mult, scale, max_ref };
		HIP_TRY(ctx, hipMemsetAsync(cnt.p, 0, 8, ctx->stream));
		HIP_TRY(ctx, hipMemsetAsync(err.p, 0, 4, ctx->stream));
		LAUNCHB(ctx, reads->total_bases * 3.0, k_encode_expand, lanes / 64, 64, A, R, AV, cfg, (const uint8_t*)reads->has_n.p, nr,
			have_list ? (const uint32_t*)todo.p : (const uint32_t*)nullptr, n_todo, redo.p, cnt.p + 1, scratch.p, per_lane, cnt.p,
			rout.p, items.p, cap_items, gaps.p, cap_gaps, pend.p, cap_pend, esb.p, cap_es, counters.p, err.p);
		HIP_TRY(ctx, hipGetLastError());
		uint32_t herr = 0; unsigned int hcnt[2];
		HIP_TRY(ctx, hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(hcnt, cnt.p, 8, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		if (herr & 2) return cl_fail(ctx, CL_E_NOMEM, "cl_encode_reads: intermediate buffers too small");
		n_todo = hcnt[1]; last_err = herr;
		if (n_todo) { HIP_TRY(ctx, hipMemcpyAsync(todo.p, redo.p, (uint64_t)n_todo * 4, hipMemcpyDeviceToDevice, ctx->stream)); have_list = true; }
		per_lane = per_lane * 8 + 96ull * maxlen; n_lanes = std::max<uint32_t>(n_lanes / 8, 64); scale *= 4;
	}
	unsigned long long hc[4];
	HIP_TRY(ctx, hipMemcpy(hc, counters.p, 32, hipMemcpyDeviceToHost));
	// pass 2
	DevBuf<uint8_t> decisions; DEV_ALLOC(ctx, decisions, hc[2] + 1);
	DevBuf<uint32_t> d_pb; DEV_ALLOC(ctx, d_pb, (uint64_t)n_packs + 1);
	HIP_TRY(ctx, hipMemcpyAsync(d_pb.p, h_pack_bounds, ((uint64_t)n_packs + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
	LAUNCH(ctx, k_estimator, grid_for(n_packs, 64), 64, (const ReadOut*)rout.p, (const PendRec*)pend.p, (const uint32_t*)d_pb.p, n_packs, (const uint32_t*)reads->lens.p, decisions.p);
	// pass 3: sizes, offsets, bytes
	DevBuf<uint32_t> sizes; DEV_ALLOC(ctx, sizes, nr);
	LAUNCH(ctx, (k_emit_tuples<false>), grid_for(nr, 64), 64, A, (const uint32_t*)reads->inv.p, nr, (const ReadOut*)rout.p, (const uint64_t*)items.p,
		(const GapRec*)gaps.p, (const char*)esb.p, (const uint8_t*)decisions.p, sizes.p, d_es_ntuples, (const uint64_t*)nullptr, (uint8_t*)nullptr);
	HIP_TRY(ctx, hipGetLastError());
	uint64_t total = 0;
	CL_TRY(dev_exclusive_scan_u64(ctx, sizes.p, d_es_off, nr, &total));
	*n_out = total;
	if (total > cap || (total && !d_es)) return cl_fail(ctx, CL_E_CAPACITY, "cl_encode_reads: need " + std::to_string(total) + " bytes");
	LAUNCH(ctx, (k_emit_tuples<true>), grid_for(nr, 64), 64, A, (const uint32_t*)reads->inv.p, nr, (const ReadOut*)rout.p, (const uint64_t*)items.p,
		(const GapRec*)gaps.p, (const char*)esb.p, (const uint8_t*)decisions.p, (uint32_t*)nullptr, (uint32_t*)nullptr, (const uint64_t*)d_es_off, d_es);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	cl_timing_collect(ctx);
	return CL_OK;
}
