// encode_core.hpp — the edit-script encoder (a10 + a11 + a12) as data-parallel stages over FRAMES and GAPS.
//
// Reference: CEncoder::AddEncodedReadWithCandidates / EncodePart / GetEditDist / EncodeWithEditScript /
// EncodeWithAlternativeRead / AdjustAnchors / StoreFrag (src/colord/encoder.cpp:778-868,1255-1575),
// refactor_edit_script (edit_script.h:416-446,591-671), CEntropy / CEntropyEstimator (utils.h:700-1131).
//
// A FRAME is one activation of AddEncodedReadWithCandidates: a stretch of a read, a recursion level and a list of
// candidates whose anchors were clipped to the stretch (AdjustAnchors); the candidate at index `level` is the one the
// frame is coded against.  Its GAPS are the stretches between consecutive anchors (and the two flanks).  The only
// sequential dependences of the reference encoder are (1) a frame at level L+1 exists only where a long gap of a
// level-L frame was rejected by the static entropy test, and (2) the adaptive estimator that decides the short gaps,
// whose outcome never creates work.  Hence: for L = 0..maxRecurence process ALL gaps of ALL level-L frames as one
// batch (geometry -> alignment -> statistics/decision -> spawn children), then replay the estimator per reader pack,
// then emit tuples per read by walking its frame tree in the reference's order.
//
// Every function here is CL_DEV: device code in the library; the debugging build (tests/tools) also compiles it for
// the host so that a divergence can be bisected without a GPU.
#pragma once
#include "common.hpp"
#include "log2_glibc.hpp"
#include "align_dev.hpp"

namespace enc {

struct ArenaV { const uint64_t* packed; const uint64_t* word_off; const uint32_t* lens; };
CL_DEV inline uint32_t arena_base_at(const ArenaV& A, uint64_t wb, uint32_t p) { return (uint32_t)(A.packed[wb + (p >> 5)] >> (62 - 2 * (p & 31))) & 3u; }
// symbol `pos` of reference read (word base wb, length len) in the given orientation
CL_DEV inline uint32_t ref_sym(const ArenaV& R, uint64_t wb, uint32_t len, bool rev, uint32_t pos)
{
	const uint32_t b = arena_base_at(R, wb, rev ? len - 1 - pos : pos);
	return rev ? 3u - b : b;
}

struct EncCfg { uint32_t c, m, min_part_alt, max_rec; double cost_mult; };

// A candidate of a frame: a VIEW of the anchors the anchor stage produced (3 uint32 each: len, pos_enc, pos_ref).
// anchors 1..n-2 are data[off + t] with pos_enc - shift; the first and the last one are held explicitly because
// AdjustAnchors clips them.  n == 1: `f` is the anchor.
struct CandEnt { uint32_t ref_id, rev, tot, n; uint64_t off; uint32_t shift; uint32_t f[3], l[3]; uint32_t pad; };
CL_DEV inline void cand_anchor(const CandEnt& c, const uint32_t* data, uint32_t t, uint32_t& len, uint32_t& pe, uint32_t& pr)
{
	if (t == 0) { len = c.f[0]; pe = c.f[1]; pr = c.f[2]; }
	else if (t == c.n - 1) { len = c.l[0]; pe = c.l[1]; pr = c.l[2]; }
	else { const uint32_t* a = data + 3 * (c.off + t); len = a[0]; pe = a[1] - c.shift; pr = a[2]; }
}
// AdjustAnchors (encoder.cpp:778-868): the view of `in` clipped to [ns, ne) of the frame's stretch, rebased to ns.
CL_DEV inline void adjust_view(const CandEnt& in, const uint32_t* data, uint32_t ns, uint32_t ne, uint32_t m, CandEnt& out)
{
	out = in; out.n = 0; out.tot = 0;
	const uint32_t n = in.n, G = 0xffffffffu; uint32_t first = G, last = G, len, pe, pr;
	for (uint32_t i = 0; i < n; ++i) { cand_anchor(in, data, i, len, pe, pr); if (pe + len > ns) { first = i; break; } }
	if (first == G) return;
	cand_anchor(in, data, first, len, pe, pr);
	if (pe < ns && (pe + len) - ns < m) ++first;
	for (uint32_t i = n; i-- > 0;) { cand_anchor(in, data, i, len, pe, pr); if (pe < ne) { last = i; break; } }
	if (last == G) return;
	cand_anchor(in, data, last, len, pe, pr);
	if (pe + len > ne && ne - pe < m) { if (last == 0) return; --last; }
	if (first > last) return;
	const uint32_t n2 = last + 1 - first;
	uint32_t f[3], l[3];
	cand_anchor(in, data, first, f[0], f[1], f[2]);
	cand_anchor(in, data, last, l[0], l[1], l[2]);
	// clip the last anchor to the stretch end, then the first one to its start; everything is rebased to ns
	if (n2 == 1)
	{
		if (f[1] + f[0] > ne) f[0] -= (f[1] + f[0] - ne);
		if (f[1] < ns) { const uint32_t d = ns - f[1]; f[0] -= d; f[1] = 0; f[2] += d; } else f[1] -= ns;
		l[0] = f[0]; l[1] = f[1]; l[2] = f[2];
	}
	else
	{
		if (l[1] + l[0] > ne) l[0] -= (l[1] + l[0] - ne);
		l[1] -= ns;
		if (f[1] < ns) { const uint32_t d = ns - f[1]; f[0] -= d; f[1] = 0; f[2] += d; } else f[1] -= ns;
	}
	uint32_t tot = f[0] + (n2 > 1 ? l[0] : 0);
	for (uint32_t i = first + 1; i < last; ++i) { cand_anchor(in, data, i, len, pe, pr); tot += len; }
	out.n = n2; out.tot = tot; out.off = in.off + first; out.shift = in.shift + ns;
	for (int i = 0; i < 3; ++i) { out.f[i] = f[i]; out.l[i] = l[i]; }
}

struct FrameRec { uint32_t read, level, enc_off, enc_len, n_cands, first_gap, n_gaps, pad; uint64_t cand_base; };
// state: 0 edit script, 1 literal, 2 pending (estimator decides), 3 child frame (aux), 4 rejected (before spawn)
enum : uint32_t { GS_ES = 0, GS_LITERAL = 1, GS_PENDING = 2, GS_CHILD = 3, GS_REJECTED = 4 };
// kind: 0 trivial (one side empty), 1 inner (NW, rows = reference), 2 flank by SHW (rows = read), 3 flank tiny (NW, rows = reference)
enum : uint32_t { GK_TRIVIAL = 0, GK_INNER = 1, GK_FLANK = 2, GK_FLANK_TINY = 3 };
struct GapRec {
	uint32_t frame, g, cur_ref, enc_start;      // enc_start: absolute position in the read
	uint32_t nr, ne, use, d_after;               // use: reference symbols the alignment can touch
	uint32_t d_before, es_len, state, aux;       // script = d_before x 'D' + es; aux: child frame / pending index
	uint64_t es_off;
	uint32_t kind, left;                         // left: the frame's first gap (aligned on reversed sequences)
	uint32_t read, ref_rev;                      // ref_rev: reference id | rev << 31
};
// statistics of one short gap for the estimator.  lens: the (ilog2(run)+1) terms of the long D / M runs IN ORDER, 6 bits
// each, count in the top 4 bits (the reference adds them to a double one by one, utils.h:1103-1104); more than 10 runs:
// count = 15 and the low 32 bits hold their sum.
struct PendRec { uint32_t rd[12]; uint32_t pl[4]; uint32_t ref_len; uint32_t pad; uint64_t lens; };
CL_DEV inline void lens_push(uint64_t& lens, uint32_t v)
{
	const uint32_t cnt = (uint32_t)(lens >> 60);
	if (cnt < 10) lens = (lens & 0x0fffffffffffffffull) | ((uint64_t)v << (6 * cnt)) | ((uint64_t)(cnt + 1) << 60);
	else if (cnt == 10) { uint32_t sum = v; for (uint32_t i = 0; i < 10; ++i) sum += (uint32_t)(lens >> (6 * i)) & 63u; lens = (15ull << 60) | sum; }
	else lens = (15ull << 60) | (uint32_t)((uint32_t)lens + v);
}
CL_DEV inline double lens_add(double cost, uint64_t lens)
{
	const uint32_t cnt = (uint32_t)(lens >> 60);
	if (cnt == 15) return cost + (double)(uint32_t)lens;
	for (uint32_t i = 0; i < cnt; ++i) cost += (double)((uint32_t)(lens >> (6 * i)) & 63u);
	return cost;
}
// What the tuple emission needs of a gap's script when it only COUNTS (sizes, saved states): the first and the last run,
// which may merge with their neighbours, and the output of the runs in between, which is fixed.
struct GapSum { uint32_t first_len, last_len, mid_bytes, mid_tuples, syms; };     // syms: first | last << 8 | (one run only) << 16
constexpr uint32_t SUM_LIMIT = 1024;          // scripts up to this length are counted from their summary (longer ones symbol by
                                              // symbol, so that states can be saved inside them)
struct LevelV { FrameRec* frames; CandEnt* cands; GapRec* gaps; char* es; PendRec* pend; uint8_t* dec; uint32_t n_frames, n_gaps; const GapSum* sums = nullptr; };
struct TreeV { LevelV lv[10]; const uint32_t* frame_of_read; };     // frame_of_read: level-0 frame of a read or ~0

// geometry of gap g of frame F coded against candidate M (EncodePart, encoder.cpp:1445-1470)
CL_DEV inline bool gap_geometry(const FrameRec& F, const CandEnt& M, const uint32_t* data, uint32_t ref_len, uint32_t g, GapRec& o)
{
	uint32_t len, pe, pr, cur_ref = 0, cur_enc = 0;
	if (g > 0) { cand_anchor(M, data, g - 1, len, pe, pr); cur_ref = pr + len; cur_enc = pe + len; }
	const bool last = g == M.n;
	uint32_t end_enc = F.enc_len, end_ref = ref_len;
	if (!last) { cand_anchor(M, data, g, len, pe, pr); end_enc = pe; end_ref = pr; }
	if (end_enc < cur_enc || end_enc > F.enc_len || cur_ref > ref_len) return false;          // inconsistent anchors
	const uint32_t want = end_ref - cur_ref, avail = ref_len - cur_ref;                    // read_view::substr clamps (utils.h:52-56)
	o.g = g; o.cur_ref = cur_ref; o.enc_start = F.enc_off + cur_enc;
	o.nr = want < avail ? want : avail; o.ne = end_enc - cur_enc;
	o.d_after = last ? 0 : end_ref - cur_ref;
	o.left = g == 0 ? 1 : 0;
	const bool flank = g == 0 || last;
	o.use = flank ? (2 * o.ne < o.nr ? 2 * o.ne : o.nr) : o.nr;
	if (o.nr == 0 || o.ne == 0) { o.kind = GK_TRIVIAL; o.use = 0; }
	else if (!flank) o.kind = GK_INNER;
	else o.kind = (o.use < 2 || o.ne < 2) ? GK_FLANK_TINY : GK_FLANK;
	o.d_before = 0; o.es_len = 0; o.state = GS_ES; o.aux = 0; o.es_off = 0;
	return true;
}

CL_DEV inline char mismatch_sym(uint32_t ref, uint32_t nw) { return (char)('X' + (nw - (nw > ref ? 1u : 0u))); }    // utils.h:341-352
CL_DEV inline bool is_mismatch(char c) { return c == 'X' || c == 'Y' || c == 'Z'; }
CL_DEV inline uint64_t sel64(uint32_t mask, uint64_t a, uint64_t b)           // mask all ones: a, zero: b (v_bfi_b32 on both halves)
{
	const uint32_t lo = ((uint32_t)a & mask) | ((uint32_t)b & ~mask), hi = ((uint32_t)(a >> 32) & mask) | ((uint32_t)(b >> 32) & ~mask);
	return ((uint64_t)hi << 32) | lo;
}
CL_DEV inline char base_letter(uint32_t b) { return b == 0 ? 'A' : b == 1 ? 'C' : b == 2 ? 'G' : 'T'; }

// refactor_edit_script (edit_script.h:416-446,591-671) over accessors: es(k) read, es_set(k, c), ref(x), enc(x)
template<class ES>
CL_DEV inline void fix_in_range(ES& es, uint32_t start, uint32_t end)
{
	if (end < start + 2) return;
	--end;
	for (;;)
	{
		while (start < end && es.get(start) == 'M') ++start;
		while (start < end && es.get(end) != 'M') --end;
		if (start == end) break;
		const char t = es.get(start); es.set(start, es.get(end)); es.set(end, t);
	}
}
template<class ES, class RefAt, class EncAt>
CL_DEV inline void refactor_es(ES& es, uint32_t n, const RefAt& ref, const EncAt& enc)
{
	uint32_t st = 0, pos = 0, es_start = 0;
	for (uint32_t p = 0; p < n; ++p)
	{
		const char c = es.get(p);
		const bool mis = is_mismatch(c), ins = c == 'A' || c == 'C' || c == 'G' || c == 'T';
		if (ins || mis || ref(st) != ref(pos)) { fix_in_range(es, es_start, p); es_start = p; if (ins || mis) ++es_start; st = pos; }
		if (!ins) ++pos;
	}
	fix_in_range(es, es_start, n);
	st = 0; pos = 0; es_start = 0;
	for (uint32_t p = 0; p < n; ++p)
	{
		const char c = es.get(p);
		const bool mis = is_mismatch(c), del = c == 'D';
		if (del || mis || enc(st) != enc(pos)) { fix_in_range(es, es_start, p); es_start = p; if (del || mis) ++es_start; st = pos; }
		if (!del) ++pos;
	}
	fix_in_range(es, es_start, n);
}

// The same transformation in one forward stream per pass: a pass rewrites every maximal REGION — consecutive symbols that
// are not a break (pass 1: insertion / substitution; pass 2: deletion / substitution) and step onto the same sequence
// symbol as their predecessor — as its matches first, then its other symbols (pass 1: 'D'; pass 2: the inserted letter,
// one and the same throughout a region).  One read per symbol, one write per region symbol, no swaps.  Equivalent to
// refactor_es above (which the generic path keeps): tests/tools/encode_host_check.py runs both against the reference.
template<class ES, class SeqAt>
CL_DEV inline void refactor_stream_pass(ES& es, uint32_t n, const SeqAt& seq, bool pass1)
{
	uint32_t pos = 0, start = 0, a = 0, len = 0, prev = 0xffffffffu; bool in_region = false; char other = 'D';
	for (uint32_t p = 0; p <= n; ++p)
	{
		const char c = p < n ? es.get(p) : 'X';                                     // a final break closes the last region
		const bool mis = is_mismatch(c), ins = c == 'A' || c == 'C' || c == 'G' || c == 'T', del = c == 'D';
		const bool brk = pass1 ? (ins || mis) : (del || mis);
		uint32_t sym = 0xffffffffu;
		if (!brk) sym = seq(pos);
		if (in_region && (brk || sym != prev))
		{
			if (len > 1) for (uint32_t q = 0; q < len; ++q) es.set(start + q, q < a ? 'M' : other);
			in_region = false;
		}
		if (brk) { if (pass1 ? !ins : !del) ++pos; continue; }
		if (!in_region) { in_region = true; start = p; a = 0; len = 0; prev = sym; other = pass1 ? 'D' : base_letter(sym); }
		a += c == 'M'; ++len; ++pos;
	}
}
template<class ES, class RefAt, class EncAt>
CL_DEV inline void refactor_stream(ES& es, uint32_t n, const RefAt& ref, const EncAt& enc)
{
	refactor_stream_pass(es, n, ref, true);
	refactor_stream_pass(es, n, enc, false);
}

// ---- small gaps: rows <= 64*NB, columns <= 256; Myers' bit-vector recurrence held in registers ------------------
// MEM provides the lane's staging memory: q(i)/t(j) sequence bytes in ALIGNMENT orientation, es get/set, and the
// per-column history hist_put(j, b, P, Ph) / hist_get(j, b, P&, Ph&).  Traceback (edlib.cpp:1021-1147) prefers up
// (vertical delta +1), then left (horizontal delta +1), then the diagonal; it needs only those two bit-vectors.
// kind GK_INNER / GK_FLANK_TINY: global; rows = reference.  GK_FLANK: rows = read, prefix-free end (SHW) with end
// position -1 a candidate when rows % 64 != 0 (edlib.cpp:666-681).  `left`: sequences are reversed; the traceback
// then yields the script already in forward order.  Returns es length; *d_before as in GetEditDist (encoder.cpp:1263).
template<int NB, class MEM>
CL_DEV inline uint32_t align_small(MEM& mem, uint32_t n, uint32_t m, uint32_t kind, bool left, uint32_t nr, uint32_t use, uint32_t* d_before)
{
	uint64_t peq[4][NB];
#pragma unroll
	for (int b = 0; b < NB; ++b)
	{
		uint64_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
		const uint32_t lo = b * 64, hi = n < lo + 64 ? n : lo + 64;
		for (uint32_t i = lo; i < hi; ++i)
		{
			const uint32_t s = mem.q(i); const uint64_t bit = 1ull << (i - lo);
			e0 |= s == 0 ? bit : 0; e1 |= s == 1 ? bit : 0; e2 |= s == 2 ? bit : 0; e3 |= s == 3 ? bit : 0;
		}
		peq[0][b] = e0; peq[1][b] = e1; peq[2][b] = e2; peq[3][b] = e3;
	}
	uint64_t Pv[NB], Mv[NB];
#pragma unroll
	for (int b = 0; b < NB; ++b) { Pv[b] = ~0ull; Mv[b] = 0; }
	const bool shw = kind == GK_FLANK;
	const uint32_t lastbit = (n - 1) & 63;
	uint32_t score = n, best = 0xffffffffu; int32_t end = (int32_t)m - 1;
	if (shw && (n & 63)) { best = n; end = -1; }
	// (the column step as one straight line — round 5: the four-way choice of the match mask by bit selects, the horizontal delta as two
	// bits (1: +1, 2: -1); hipcc made branches of both, four times per column)
	const uint64_t lastmask = 1ull << lastbit;
	for (uint32_t j = 0; j < m; ++j)
	{
		const uint32_t c = mem.t(j);
		const uint32_t M0 = 0u - (c & 1), M1 = 0u - (c >> 1);
		uint32_t hin = 1;
#pragma unroll
		for (int b = 0; b < NB; ++b)
		{
			uint64_t Eq = sel64(M1, sel64(M0, peq[3][b], peq[2][b]), sel64(M0, peq[1][b], peq[0][b]));
			const uint64_t hneg = hin >> 1, hpos = hin & 1;
			const uint64_t Xv = Eq | Mv[b];
			Eq |= hneg;
			const uint64_t Xh = (((Eq & Pv[b]) + Pv[b]) ^ Pv[b]) | Eq;
			uint64_t Ph = Mv[b] | ~(Xh | Pv[b]);
			uint64_t Mh = Pv[b] & Xh;
			if (b == NB - 1) score += (uint32_t)((Ph & lastmask) != 0) - (uint32_t)((Mh & lastmask) != 0);
			const uint64_t ph_rows = Ph;
			hin = (uint32_t)(Ph >> 63) | ((uint32_t)(Mh >> 63) << 1);
			Ph = (Ph << 1) | hpos; Mh = (Mh << 1) | hneg;
			Pv[b] = Mh | ~(Xv | Ph);
			Mv[b] = Ph & Xv;
			mem.hist_put(j, b, Pv[b], ph_rows);
		}
		const bool better = shw && score < best;
		best = better ? score : best; end = better ? (int32_t)j : end;
	}
	// traceback from (n, jend)
	uint32_t i = n, j = shw ? (uint32_t)(end + 1) : m, k = 0;
	const uint32_t t_used = j;
	uint64_t P = 0, Ph = 0; uint32_t cj = 0xffffffffu, cb = 0xffffffffu;
	while (i > 0 && j > 0)
	{
		const uint32_t r = i - 1, b = r >> 6;
		if (cj != j || cb != b) { mem.hist_get(j - 1, b, P, Ph); cj = j; cb = b; }
		const uint64_t bit = 1ull << (r & 63);
		const uint32_t qs = mem.q(r), ts = mem.t(j - 1);
		char ch;
		if (P & bit) { ch = shw ? base_letter(qs) : 'D'; --i; }                        // up: consumes a row symbol
		else if (Ph & bit) { ch = shw ? 'D' : base_letter(ts); --j; }                  // left: consumes a column symbol
		else { ch = qs == ts ? 'M' : (shw ? mismatch_sym(ts, qs) : mismatch_sym(qs, ts)); --i; --j; }
		mem.es_set(k++, ch);
	}
	while (i > 0) { --i; mem.es_set(k++, shw ? base_letter(mem.q(i)) : 'D'); }
	while (j > 0) { --j; mem.es_set(k++, shw ? 'D' : base_letter(mem.t(j))); }
	if (!left) for (uint32_t a = 0, z = k; a + 1 < z; ++a) { --z; const char t = mem.es_get(a); mem.es_set(a, mem.es_get(z)); mem.es_set(z, t); }
	// canonical indel placement on the forward sequences
	*d_before = 0;
	struct ES { MEM& m; CL_DEV char get(uint32_t p) const { return m.es_get(p); } CL_DEV void set(uint32_t p, char c) { m.es_set(p, c); } } es{ mem };
	const bool rows_ref = !shw;
	if (left)
	{	// find_edit_dist_with_edlib_ex_odwr_reverse (edit_script.h:405-419): the reference part starts ref_offset symbols in
		const uint32_t ref_end = shw ? (uint32_t)end : use - 1;
		const uint32_t ref_offset = (nr - 1) - ref_end;                                  // uint32 wrap for end = -1, as the reference
		const uint32_t r_used = shw ? t_used : n, e_len = shw ? n : m;
		auto ref = [&](uint32_t x) -> uint32_t { const uint32_t idx = r_used - 1 - x; return rows_ref ? mem.q(idx) : mem.t(idx); };
		auto encf = [&](uint32_t x) -> uint32_t { const uint32_t idx = e_len - 1 - x; return rows_ref ? mem.t(idx) : mem.q(idx); };
		refactor_stream(es, k, ref, encf);
		*d_before = ref_offset;
	}
	else
	{
		auto ref = [&](uint32_t x) -> uint32_t { return rows_ref ? mem.q(x) : mem.t(x); };
		auto encf = [&](uint32_t x) -> uint32_t { return rows_ref ? mem.t(x) : mem.q(x); };
		refactor_stream(es, k, ref, encf);
	}
	return k;
}

// ---- mid-size gaps: the same recurrence with the state in the lane's memory (rows * columns / 64 <= MID_CELLS) --------
// MEM additionally provides peq / peq_set (symbol, block), pv / mv get + set (block).  Same observable behaviour as
// align_small (and as edlib below its 1 MiB traceback budget: 20 bytes * blocks * columns stays under it).
// No kernel of the library uses this form any more (measured on 1.27 M gaps of ~350 x 350: 250 ms against 155 ms for a wave per
// gap, whose sweep, symbol conversion and indel canonicalisation are wave-parallel); it stays as the plain sequential
// statement of the recurrence for the debugging host build (tests/tools/encode_host.hip).
constexpr uint32_t MID_ROWS = 16384, MID_COLS = 4096;
// NBR > 0: the vertical deltas of up to NBR blocks stay in registers (only the match masks and the history are memory).
template<int NBR, class MEM>
CL_DEV inline uint32_t align_mid(MEM& mem, uint32_t n, uint32_t m, uint32_t kind, bool left, uint32_t nr, uint32_t use, uint32_t* d_before)
{
	const uint32_t nb = (n + 63) / 64;
	uint64_t rPv[NBR > 0 ? NBR : 1], rMv[NBR > 0 ? NBR : 1];
	for (uint32_t b = 0; b < nb; ++b)
	{
		uint64_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
		const uint32_t lo = b * 64, hi = n < lo + 64 ? n : lo + 64;
		for (uint32_t i = lo; i < hi; ++i)
		{
			const uint32_t s = mem.q(i); const uint64_t bit = 1ull << (i - lo);
			e0 |= s == 0 ? bit : 0; e1 |= s == 1 ? bit : 0; e2 |= s == 2 ? bit : 0; e3 |= s == 3 ? bit : 0;
		}
		mem.peq_set(0, b, e0); mem.peq_set(1, b, e1); mem.peq_set(2, b, e2); mem.peq_set(3, b, e3);
		if (NBR == 0) { mem.pv_set(b, ~0ull); mem.mv_set(b, 0); }
	}
	if (NBR > 0)
	{
#pragma unroll
		for (int b = 0; b < (NBR > 0 ? NBR : 1); ++b) { rPv[b] = ~0ull; rMv[b] = 0; }
	}
	mem.lap(0);
	const bool shw = kind == GK_FLANK;
	const uint32_t lastbit = (n - 1) & 63;
	uint32_t score = n, best = 0xffffffffu; int32_t end = (int32_t)m - 1;
	if (shw && (n & 63)) { best = n; end = -1; }
	for (uint32_t j = 0; j < m; ++j)
	{
		const uint32_t c = mem.t(j);
		int hin = 1;
		if (NBR > 0)
		{
#pragma unroll
			for (int b = 0; b < (NBR > 0 ? NBR : 1); ++b)
			{
				if ((uint32_t)b >= nb) continue;
				uint64_t Eq = mem.peq(c, b); const uint64_t Pv = rPv[b], Mv = rMv[b];
				const uint64_t hneg = hin < 0 ? 1ull : 0ull;
				const uint64_t Xv = Eq | Mv;
				Eq |= hneg;
				const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
				uint64_t Ph = Mv | ~(Xh | Pv);
				uint64_t Mh = Pv & Xh;
				if ((uint32_t)b == nb - 1) score += (uint32_t)((Ph >> lastbit) & 1) - (uint32_t)((Mh >> lastbit) & 1);
				const uint64_t ph_rows = Ph;
				const int hout = (int)(Ph >> 63) - (int)(Mh >> 63);
				Ph <<= 1; Mh <<= 1;
				Mh |= hneg; Ph |= hin > 0 ? 1ull : 0ull;
				const uint64_t Pn = Mh | ~(Xv | Ph);
				rPv[b] = Pn; rMv[b] = Ph & Xv;
				mem.hist_put(j, b, Pn, ph_rows);
				hin = hout;
			}
		}
		else
		for (uint32_t b = 0; b < nb; ++b)
		{
			uint64_t Eq = mem.peq(c, b); const uint64_t Pv = mem.pv(b), Mv = mem.mv(b);
			const uint64_t hneg = hin < 0 ? 1ull : 0ull;
			const uint64_t Xv = Eq | Mv;
			Eq |= hneg;
			const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
			uint64_t Ph = Mv | ~(Xh | Pv);
			uint64_t Mh = Pv & Xh;
			if (b == nb - 1) score += (uint32_t)((Ph >> lastbit) & 1) - (uint32_t)((Mh >> lastbit) & 1);
			const uint64_t ph_rows = Ph;
			const int hout = (int)(Ph >> 63) - (int)(Mh >> 63);
			Ph <<= 1; Mh <<= 1;
			Mh |= hneg; Ph |= hin > 0 ? 1ull : 0ull;
			const uint64_t Pn = Mh | ~(Xv | Ph);
			mem.pv_set(b, Pn); mem.mv_set(b, Ph & Xv);
			mem.hist_put(j, b, Pn, ph_rows);
			hin = hout;
		}
		if (shw && score < best) { best = score; end = (int32_t)j; }
	}
	mem.lap(1);
	uint32_t i = n, j = shw ? (uint32_t)(end + 1) : m, k = 0;
	const uint32_t t_used = j;
	uint64_t P = 0, Ph = 0; uint32_t cj = 0xffffffffu, cb = 0xffffffffu;
	while (i > 0 && j > 0)
	{
		const uint32_t r = i - 1, b = r >> 6;
		if (cj != j || cb != b) { mem.hist_get(j - 1, b, P, Ph); cj = j; cb = b; }
		const uint64_t bit = 1ull << (r & 63);
		const uint32_t qs = mem.q(r), ts = mem.t(j - 1);
		char ch;
		if (P & bit) { ch = shw ? base_letter(qs) : 'D'; --i; }
		else if (Ph & bit) { ch = shw ? 'D' : base_letter(ts); --j; }
		else { ch = qs == ts ? 'M' : (shw ? mismatch_sym(ts, qs) : mismatch_sym(qs, ts)); --i; --j; }
		mem.es_set(k++, ch);
	}
	while (i > 0) { --i; mem.es_set(k++, shw ? base_letter(mem.q(i)) : 'D'); }
	while (j > 0) { --j; mem.es_set(k++, shw ? 'D' : base_letter(mem.t(j))); }
	if (!left) for (uint32_t a = 0, z = k; a + 1 < z; ++a) { --z; const char t = mem.es_get(a); mem.es_set(a, mem.es_get(z)); mem.es_set(z, t); }
	mem.lap(2);
	*d_before = 0;
	struct ES { MEM& m; CL_DEV char get(uint32_t p) const { return m.es_get(p); } CL_DEV void set(uint32_t p, char c) { m.es_set(p, c); } } es{ mem };
	const bool rows_ref = !shw;
	if (left)
	{
		const uint32_t ref_end = shw ? (uint32_t)end : use - 1;
		const uint32_t ref_offset = (nr - 1) - ref_end;
		const uint32_t r_used = shw ? t_used : n, e_len = shw ? n : m;
		auto ref = [&](uint32_t x) -> uint32_t { const uint32_t idx = r_used - 1 - x; return rows_ref ? mem.q(idx) : mem.t(idx); };
		auto encf = [&](uint32_t x) -> uint32_t { const uint32_t idx = e_len - 1 - x; return rows_ref ? mem.t(idx) : mem.q(idx); };
		refactor_stream(es, k, ref, encf);
		*d_before = ref_offset;
	}
	else
	{
		auto ref = [&](uint32_t x) -> uint32_t { return rows_ref ? mem.q(x) : mem.t(x); };
		auto encf = [&](uint32_t x) -> uint32_t { return rows_ref ? mem.t(x) : mem.q(x); };
		refactor_stream(es, k, ref, encf);
	}
	mem.lap(3);
	return k;
}

// ---- large gaps: generic path, all working memory from the lane's pool (align_dev.hpp), Hirschberg when edlib would ---
// rbuf: the reference symbols the alignment can touch (forward order): the whole part for an inner gap, its first `use`
// symbols for the right flank, its LAST `use` symbols for the left flank.  enc: the read part, forward.  dst: >= use + ne chars.
struct PtrES { char* p; CL_DEV char get(uint32_t i) const { return p[i]; } CL_DEV void set(uint32_t i, char c) { p[i] = c; } };
CL_DEV inline uint32_t align_large(LanePool& pool, const uint8_t* rbuf, uint32_t nr, uint32_t use, const uint8_t* enc, uint32_t ne, uint32_t kind, bool left, char* dst, uint32_t* d_before)
{
	uint32_t n = 0;
	*d_before = 0;
	const uint64_t mk = pool.mark();
	uint8_t* opsbuf = (uint8_t*)pool.alloc((uint64_t)use + ne + 16);
	uint8_t* r2 = (uint8_t*)pool.alloc(use + 16ull); uint8_t* e2 = (uint8_t*)pool.alloc(ne + 16ull);
	if (pool.overflow) { pool.release(mk); return 0; }
	OpsOut ops{ opsbuf, 0 };
	PtrES es{ dst };
	if (kind == GK_INNER)
	{	// find_edit_dist_with_edlib_ex (edit_script.h:272-336): query = ref, target = enc, global.  The small-input DP of the
		// reference (edit_script.h:156-239) has the same move preference in this orientation.
		const uint32_t best = nw_distance(pool, Seq{ rbuf, 1 }, nr, Seq{ enc, 1 }, ne);
		nw_path(pool, rbuf, nr, enc, ne, best, ops);
		uint32_t pr = 0, pe = 0;
		for (uint64_t i = 0; i < ops.n; ++i)
			switch (ops.p[i])
			{
			case 0: dst[n++] = 'M'; ++pr; ++pe; break;
			case 1: dst[n++] = 'D'; ++pr; break;
			case 2: dst[n++] = base_letter(enc[pe++]); break;
			default: dst[n++] = mismatch_sym(rbuf[pr], enc[pe]); ++pr; ++pe;
			}
		refactor_es(es, n, [&](uint32_t x) -> uint32_t { return rbuf[x]; }, [&](uint32_t x) -> uint32_t { return enc[x]; });
		pool.release(mk);
		return n;
	}
	if (left) { for (uint32_t i = 0; i < use; ++i) r2[i] = rbuf[use - 1 - i]; for (uint32_t i = 0; i < ne; ++i) e2[i] = enc[ne - 1 - i]; }
	else { for (uint32_t i = 0; i < use; ++i) r2[i] = rbuf[i]; for (uint32_t i = 0; i < ne; ++i) e2[i] = enc[i]; }
	uint32_t ref_end;
	if (kind == GK_FLANK_TINY)
	{	// find_edit_dist (edit_script.h:156-239): global, rows = ref
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
		const uint8_t* rf = rbuf + (ref_offset - (nr - use));                 // = reference part + ref_offset
		refactor_es(es, n, [&](uint32_t x) -> uint32_t { return rf[x]; }, [&](uint32_t x) -> uint32_t { return enc[x]; });
		*d_before = ref_offset;
	}
	else refactor_es(es, n, [&](uint32_t x) -> uint32_t { return rbuf[x]; }, [&](uint32_t x) -> uint32_t { return enc[x]; });
	pool.release(mk);
	return n;
}

// ---- statistics and the static decision of one gap (EncodePart, encoder.cpp:1471-1496) ---------------------------
CL_DEV inline uint32_t es_class(char c)       // order of CEntropy::es_sym = A C D G M T X Y Z
{
	switch (c) { case 'A': return 0; case 'C': return 1; case 'D': return 2; case 'G': return 3; case 'M': return 4; case 'T': return 5; case 'X': return 6; case 'Y': return 7; default: return 8; }
}
CL_DEV inline uint32_t bitlen32(uint64_t x) { return x ? 64u - (uint32_t)__builtin_clzll(x) : 0u; }
CL_DEV inline uint32_t est_code(char c)       // estimator alphabet (utils.h:914-930): A C G T D M X Y Z S R
{
	switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'D': return 4; case 'M': return 5; case 'X': return 6; case 'Y': return 7; case 'Z': return 8; default: return 11; }
}
// analyze_es (utils.h:819-874) reduced to what CEntropyEstimator::EncodeWithEditScript consumes
CL_DEV inline void analyze_es(const char* es, uint32_t n, uint32_t d_before, PendRec& p)
{
	for (int i = 0; i < 12; ++i) p.rd[i] = 0;
	p.lens = 0; p.pad = 0;
	char c = d_before ? 'D' : ' '; uint32_t len = d_before;
	for (uint32_t i = 0; i <= n; ++i)
	{
		const char x = i < n ? es[i] : ' ';
		if (x == c) { ++len; continue; }
		if (c == 'D') { if (len >= 10) { ++p.rd[9]; lens_push(p.lens, bitlen32(len) + 1); } else p.rd[4] += len; }
		else if (c == 'M') { if (len >= 15) { ++p.rd[10]; lens_push(p.lens, bitlen32(len) + 1); } else p.rd[5] += len; }
		else if (c != ' ') ++p.rd[est_code(c)];
		c = x; len = 1;
	}
}
CL_DEV inline double entropy_hist(const uint32_t* h, int k)       // CEntropy (utils.h:706-752)
{
	double sum = 0; for (int i = 0; i < k; ++i) sum += h[i];
	const double rec = 1.0 / sum; double e = 0;
	for (int c = 0; c < k; ++c) if (h[c]) { const double p = (double)h[c] * rec; e += glibc_log2::log2(p) * p; }
	return -e;
}
CL_DEV inline void gap_stats(GapRec& g, const char* es, const ArenaV& A, uint64_t read_wb, const EncCfg& cfg, PendRec* pend_slot)
{
	if (g.ne < cfg.min_part_alt)
	{	// the adaptive estimator decides later
		analyze_es(es, g.es_len, g.d_before, *pend_slot);
		pend_slot->pl[0] = pend_slot->pl[1] = pend_slot->pl[2] = pend_slot->pl[3] = 0;
		for (uint32_t i = 0; i < g.ne; ++i) ++pend_slot->pl[arena_base_at(A, read_wb, g.enc_start + i)];
		pend_slot->ref_len = g.nr;
		g.state = GS_PENDING;
		return;
	}
	// EncodeWithEditScript (encoder.cpp:1315-1327) with GetEditScriptEntropyInput (:1299-1311)
	uint32_t nd = 0; while (nd < g.es_len && es[nd] == 'D') ++nd;
	const char* p = es; uint32_t n = g.es_len, extra = g.d_before;
	if (nd + g.d_before >= 10) { p += nd; n -= nd; extra = 0; }
	uint32_t h[9] = { 0, 0, extra, 0, 0, 0, 0, 0, 0 };
	for (uint32_t i = 0; i < n; ++i) ++h[es_class(p[i])];
	uint32_t hd[4] = { 0, 0, 0, 0 };
	for (uint32_t i = 0; i < g.ne; ++i) ++hd[arena_base_at(A, read_wb, g.enc_start + i)];
	const bool accept = entropy_hist(h, 9) * (double)(n + extra) * cfg.cost_mult < entropy_hist(hd, 4) * (double)g.ne;
	g.state = accept ? GS_ES : GS_REJECTED;
}

// EncodeWithAlternativeRead (encoder.cpp:1329-1346): the candidates of the child frame for a rejected gap, or none.
// out: n_cands entries.  Returns true when the gap continues in a child frame.
CL_DEV inline bool spawn_cands(const FrameRec& F, const CandEnt* cands, const uint32_t* data, const GapRec& g, const EncCfg& cfg, CandEnt* out)
{
	const uint32_t lv = F.level, nc = F.n_cands;
	if (nc <= lv + 1 || g.ne < cfg.min_part_alt || lv >= cfg.max_rec) return false;
	const uint32_t ns = g.enc_start - F.enc_off, ne = ns + g.ne;
	for (uint32_t j = 0; j <= lv; ++j) out[j] = cands[j];
	for (uint32_t j = lv + 1; j < nc; ++j) adjust_view(cands[j], data, ns, ne, cfg.m, out[j]);
	// std::sort on <= 16 elements = insertion sort, i.e. stable (descending total anchor length)
	for (uint32_t i = lv + 2; i < nc; ++i)
	{
		const CandEnt x = out[i]; uint32_t j = i;
		while (j > lv + 1 && x.tot > out[j - 1].tot) { out[j] = out[j - 1]; --j; }
		out[j] = x;
	}
	return out[lv + 1].tot != 0;
}

// ---- the adaptive estimator over one reader pack (utils.h:877-1130; reset at encoder.cpp:1677) --------------------
struct Estim { uint32_t dna[4], es[12], dec[2]; double dna_logs[4], es_logs[12], dec_logs[2]; uint32_t dna_sum, es_sum, dec_sum; };
CL_DEV inline void est_rescale(uint32_t* a, int n, uint32_t& sum, uint32_t mx) { while (sum > mx) { sum = 0; for (int i = 0; i < n; ++i) { a[i] = (a[i] + 1) / 2; sum += a[i]; } } }
CL_DEV inline void est_logs(const uint32_t* st, double* lg, int n, uint32_t sum)
{
	const double rec = 1.0 / sum;
	for (int i = 0; i < n; ++i) lg[i] = st[i] ? -glibc_log2::log2((double)st[i] * rec) : 0.0;
}
CL_DEV inline void est_reset(Estim& e)
{
	for (int i = 0; i < 4; ++i) e.dna[i] = 1;
	e.dna_sum = 4;
	for (int i = 0; i < 12; ++i) e.es[i] = 1;
	e.es_sum = 12;
	e.dec[0] = e.dec[1] = 1; e.dec_sum = 2;
	est_logs(e.dna, e.dna_logs, 4, e.dna_sum); est_logs(e.es, e.es_logs, 12, e.es_sum); est_logs(e.dec, e.dec_logs, 2, e.dec_sum);
}
CL_DEV inline void est_log_read(Estim& e, const uint32_t* counts, uint32_t len)           // LogRead (utils.h:946-955)
{
	for (int i = 0; i < 4; ++i) e.dna[i] += counts[i];
	e.dna_sum += len;
	est_rescale(e.dna, 4, e.dna_sum, 1u << 20);
	est_logs(e.dna, e.dna_logs, 4, e.dna_sum);
}
CL_DEV inline bool est_decide(Estim& e, const PendRec& p)                                  // EncodeWithEditScript (utils.h:1060-1130); true = edit script
{
	uint32_t loc[12]; uint32_t loc_sum = e.es_sum;
	for (int i = 0; i < 12; ++i) { loc[i] = e.es[i] + p.rd[i]; loc_sum += p.rd[i]; }
	double es_cost = e.dec_logs[0], plain_cost = e.dec_logs[1];
	est_logs(loc, e.es_logs, 12, loc_sum);
	for (int i = 0; i < 12; ++i) es_cost += p.rd[i] * e.es_logs[i];
	es_cost = lens_add(es_cost, p.lens);
	for (int i = 0; i < 4; ++i) plain_cost += p.pl[i] * e.dna_logs[i];
	plain_cost += bitlen32(p.ref_len) + 1;
	const bool choose_plain = plain_cost < es_cost;
	if (choose_plain) { ++e.dec[1]; est_rescale(e.es, 12, e.es_sum, 1u << 20); }
	else { ++e.dec[0]; for (int i = 0; i < 12; ++i) e.es[i] = loc[i]; e.es_sum = loc_sum; est_rescale(e.es, 12, e.es_sum, 1u << 20); }
	++e.dec_sum;
	est_rescale(e.dec, 2, e.dec_sum, 1u << 20);
	est_logs(e.dec, e.dec_logs, 2, e.dec_sum);
	return !choose_plain;
}
// the pending gaps of one read in encoding order = depth-first over its frame tree
CL_DEV inline void est_read(Estim& e, const TreeV& T, uint32_t frame0)
{
	uint32_t sf[10], sg[10]; int sp = 0;
	sf[0] = frame0; sg[0] = 0;
	while (sp >= 0)
	{
		const LevelV& L = T.lv[sp];
		const FrameRec& F = L.frames[sf[sp]];
		if (sg[sp] == F.n_gaps) { --sp; continue; }
		const GapRec& g = L.gaps[F.first_gap + sg[sp]++];
		if (g.state == GS_PENDING) L.dec[g.aux] = est_decide(e, L.pend[g.aux]) ? 1 : 0;
		else if (g.state == GS_CHILD) { ++sp; sf[sp] = g.aux; sg[sp] = 0; }
	}
}

// the pending gaps of a read in encoding order as (level << 28 | pending index); out == nullptr: count only
CL_DEV inline uint32_t pend_walk(const TreeV& T, uint32_t frame0, uint32_t* out)
{
	uint32_t sf[10], sg[10], n = 0; int sp = 0;
	sf[0] = frame0; sg[0] = 0;
	while (sp >= 0)
	{
		const LevelV& L = T.lv[sp];
		const FrameRec& F = L.frames[sf[sp]];
		if (sg[sp] == F.n_gaps) { --sp; continue; }
		const GapRec& g = L.gaps[F.first_gap + sg[sp]++];
		if (g.state == GS_PENDING) { if (out) out[n] = ((uint32_t)sp << 28) | g.aux; ++n; }
		else if (g.state == GS_CHILD) { ++sp; sf[sp] = g.aux; sg[sp] = 0; }
	}
	return n;
}

// ---- tuple emission (encoder.cpp:1348-1443) -------------------------------------------------------------------------
struct TupleOut {
	uint8_t* p; uint64_t n; uint32_t n_tuples; bool write;
	// Bytes are assembled in the aligned 8-byte word that holds position n and leave with one store per word; `have` bytes
	// of that word are taken, the first `skip` of them by whoever wrote before this stream / chunk (first word only).
	uint64_t acc = 0; uint32_t have = 0, skip = 0;
	CL_DEV inline void start() { have = skip = (uint32_t)(((uint64_t)(size_t)p + n) & 7); acc = 0; }   // write mode: after p and n are set
	// nb <= 8 bytes, the first in the low byte of v
	CL_DEV inline void put(uint64_t v, uint32_t nb)
	{
		n += nb;
		if (!write) return;
		acc |= v << (8 * have);
		const uint32_t tot = have + nb;
		if (tot >= 8)
		{
			uint8_t* w = p + (n - tot);                                // the word's address
			if (skip) { for (uint32_t i = skip; i < 8; ++i) w[i] = (uint8_t)(acc >> (8 * i)); skip = 0; }
			else *(uint64_t*)w = acc;
			acc = have ? v >> (8 * (8 - have)) : 0;
			have = tot - 8;
		}
		else have = tot;
	}
	// `rep` one-byte tuples of the same value (runs of matches shorter than an anchor, of deletions, ...)
	CL_DEV inline void fill(uint8_t v, uint32_t rep)
	{
		n_tuples += rep;
		if (!write) { n += rep; return; }
		const uint64_t pat = 0x0101010101010101ull * v;
		while (rep)
		{
			const uint32_t k = rep < 8 ? rep : 8;
			put(k == 8 ? pat : pat & ((1ull << (8 * k)) - 1), k);
			rep -= k;
		}
	}
	CL_DEV inline void finish() { if (write) { uint8_t* w = p + (n - have); for (uint32_t i = skip; i < have; ++i) w[i] = (uint8_t)(acc >> (8 * i)); } have = skip = 0; acc = 0; }
	CL_DEV inline void t1(uint32_t type, uint32_t val) { put((type << 4) + val, 1); ++n_tuples; }
	CL_DEV inline void t28(uint32_t type, uint32_t v) { put(__builtin_bswap32((type << 28) + v), 4); ++n_tuples; }           // type nibble + 28 bits, big-endian
	CL_DEV inline void tid(uint32_t type, uint32_t id, uint32_t rev) { put(((type << 4) + rev) | ((uint64_t)__builtin_bswap32(id) << 8), 5); ++n_tuples; }
};
// One StoreFrag segment at a time (encoder.cpp:1414-1443): the header (alt_id / main_ref, and for level > 0 the
// last_pos_in_ref deletions) is emitted when the first symbol of the segment arrives, so empty segments leave no trace.
struct SegWriter {
	TupleOut* o; char sym; uint32_t rep; bool open, first; uint32_t main_id;
	uint32_t level, ref_id, rev, last_pos;
	CL_DEV inline void flush_run()
	{	// (one t28 and one t1 call site: their inlined bodies are large)
		if (!rep) return;
		uint32_t type, val = 0;
		if (sym == 'M') type = 2; else if (sym == 'D') type = 1;
		else if (sym == 'X' || sym == 'Y' || sym == 'Z') { type = 3; val = (uint32_t)(sym - 'X'); }
		else { type = 0; val = sym == 'A' ? 0 : sym == 'C' ? 1 : sym == 'G' ? 2 : 3; }
		if ((type == 2 && rep >= 15) || (type == 1 && rep > 16)) o->t28(type == 2 ? 4 : 5, rep);
		else o->fill((uint8_t)((type << 4) + val), rep);
		rep = 0;
	}
	CL_DEV inline void run(char s, uint32_t n) { if (rep && s == sym) { rep += n; return; } flush_run(); sym = s; rep = n; }
	CL_DEV inline void add(char s, uint32_t n)
	{
		if (!n) return;
		if (!open)
		{
			open = true;
			if (level == 0) { if (ref_id != main_id) o->tid(6, ref_id, rev); else if (!first) o->t1(7, 0); }
			else { if (ref_id != main_id) o->tid(6, ref_id, rev); else o->t1(7, 0); if (last_pos) run('D', last_pos); }
		}
		run(s, n);
	}
	// StoreFrag: closes the segment; returns true when it held anything (then last_pos_in_ref moves to cur_pos)
	CL_DEV inline bool store() { const bool had = open; if (open) { flush_run(); first = false; } open = false; return had; }
};

// bytes / tuples a finished run of `len` symbols `sym` puts out (singleEditScriptSymbolStore, as SegWriter::flush_run)
CL_DEV inline uint32_t run_bytes(char sym, uint32_t len) { return (sym == 'M' && len >= 15) || (sym == 'D' && len > 16) ? 4u : len; }
CL_DEV inline uint32_t run_tuples(char sym, uint32_t len) { return (sym == 'M' && len >= 15) || (sym == 'D' && len > 16) ? 1u : len; }
CL_DEV inline GapSum gap_summary(const char* es, uint32_t k)
{
	GapSum s{ 0, 0, 0, 0, 0 };
	if (!k) return s;
	const char f = es[0];
	uint32_t i = 1;
	while (i < k && es[i] == f) ++i;
	s.first_len = i;
	if (i == k) { s.last_len = i; s.syms = (uint32_t)(uint8_t)f | ((uint32_t)(uint8_t)f << 8) | (1u << 16); return s; }
	char c = es[i]; uint32_t len = 0;
	for (; i < k; ++i)
	{
		if (es[i] == c) { ++len; continue; }
		s.mid_bytes += run_bytes(c, len); s.mid_tuples += run_tuples(c, len);
		c = es[i]; len = 1;
	}
	s.last_len = len;
	s.syms = (uint32_t)(uint8_t)f | ((uint32_t)(uint8_t)c << 8);
	return s;
}
// A saved state of the walk below at the top of its loop (`it` fragments done, `n` bytes / `n_tuples` tuples out, the
// open run in (sym, rep)).  The count pass of the kernels saves one every few KB of output, so that the write pass can
// run one lane per CHUNK, not per read: the chunk that starts at the state ends at (stop_it, stop_q).  States are saved
// between fragments and, every 32 symbols, inside the scripts of gaps (a gap can be most of a read).
#ifdef CL_EMIT_DEBUG
__device__ unsigned long long g_emit_dbg[2];
#endif
struct EmitCk {
	uint32_t read, used; uint64_t start_it, stop_it, n; uint32_t mid_q, stop_q;     // mid_q / stop_q != 0: inside the script of fragment start_it / stop_it, at that symbol
	uint32_t sf[10], si[10], s_last[10], s_cur[10]; int32_t sp; uint32_t enter, n_tuples;
	uint32_t sym, rep, open, first, main_id, level, ref_id, rev, last_pos, pad;
};
// WRITE = false: sizes / tuple counts of read r; with cks != nullptr also the saved states, one per `chunk` output bytes, in
// its n_slots slots (slot 0 = the start of the read).  WRITE = true: the whole read (ck == nullptr) or the chunk of *ck.
template<bool WRITE>
CL_DEV inline void emit_read(const ArenaV& A, const uint32_t* inv, const uint8_t* has_n, const TreeV& T, uint32_t r, const uint32_t* anchors_data,
                             uint32_t* sizes, uint32_t* ntuples, const uint64_t* es_off, uint8_t* out,
                             EmitCk* cks = nullptr, uint32_t n_slots = 0, uint32_t chunk = 1, const EmitCk* ck = nullptr)
{
	const uint32_t len = A.lens[r]; const uint64_t wb = A.word_off[r];
	TupleOut o{ WRITE ? out + es_off[r] : nullptr, 0, 0, WRITE };
	if (WRITE) o.start();
	const uint32_t f0 = T.frame_of_read[r];
	if (f0 == 0xffffffffu)
	{	// AddPlainRead / AddPlainReadWithN (encoder.cpp:663-681)
		o.t1(has_n[r] ? 11 : 9, 0);
		for (uint32_t i0 = 0; i0 < len; i0 += 32)
		{	// one word of bases and one of N flags per 32 tuples
			const uint64_t pw = A.packed[wb + (i0 >> 5)]; const uint32_t iw = inv[wb + (i0 >> 5)];
			const uint32_t nb = len - i0 < 32 ? len - i0 : 32;
			for (uint32_t j = 0; j < nb; ++j)
			{
				const bool isn = (iw >> (31 - j)) & 1u;
				o.t1(8, isn ? 4u : (uint32_t)(pw >> (62 - 2 * j)) & 3u);
			}
		}
		o.finish();
		if (!WRITE) { sizes[r] = (uint32_t)o.n; ntuples[r] = o.n_tuples; }
		return;
	}
	uint32_t sf[10], si[10], s_last[10], s_cur[10]; int sp = 0;     // frame, next fragment, last_pos_in_ref, cur_pos_in_ref
	sf[0] = f0; si[0] = 0; s_last[0] = 0; s_cur[0] = 0;
	SegWriter w{ &o, 'M', 0, false, true, 0, 0, 0, 0, 0 };
	{
		const FrameRec& F = T.lv[0].frames[f0];
		const CandEnt& M = T.lv[0].cands[F.cand_base];
		w.main_id = M.ref_id;
		if (!(WRITE && ck && ck->start_it)) o.tid(10, M.ref_id, M.rev);  // start_es (encoder.cpp:1523-1527)
	}
	bool enter = true;
	uint64_t it = 0, stop = ~0ull; uint32_t last_k = 0, prev_slot = 0, stop_q = 0, mid_q = 0;
	if (WRITE && ck)
	{
		stop = ck->stop_it; stop_q = ck->stop_q;
		if (ck->start_it)
		{	// resume at the saved state
			for (int i = 0; i < 10; ++i) { sf[i] = ck->sf[i]; si[i] = ck->si[i]; s_last[i] = ck->s_last[i]; s_cur[i] = ck->s_cur[i]; }
			sp = ck->sp; enter = ck->enter != 0; it = ck->start_it; mid_q = ck->mid_q;
			o.n = ck->n; o.n_tuples = ck->n_tuples; o.start();
			w.sym = (char)ck->sym; w.rep = ck->rep; w.open = ck->open != 0; w.first = ck->first != 0; w.main_id = ck->main_id;
			w.level = ck->level; w.ref_id = ck->ref_id; w.rev = ck->rev; w.last_pos = ck->last_pos;
		}
	}
	if (!WRITE && cks) { cks[0].read = r; cks[0].used = 1; cks[0].start_it = 0; cks[0].stop_it = ~0ull; cks[0].mid_q = 0; cks[0].stop_q = 0; }
	auto save_state = [&](uint32_t q) {          // count pass: a state per `chunk` output bytes
		const uint32_t k = (uint32_t)(o.n / chunk);
		if (k <= last_k || k >= n_slots) return;
		EmitCk& c = cks[k];
		for (int i = 0; i < 10; ++i) { c.sf[i] = sf[i]; c.si[i] = si[i]; c.s_last[i] = s_last[i]; c.s_cur[i] = s_cur[i]; }
		c.sp = sp; c.enter = enter ? 1 : 0; c.n = o.n; c.n_tuples = o.n_tuples;
		c.sym = (uint32_t)(uint8_t)w.sym; c.rep = w.rep; c.open = w.open ? 1 : 0; c.first = w.first ? 1 : 0; c.main_id = w.main_id;
		c.level = w.level; c.ref_id = w.ref_id; c.rev = w.rev; c.last_pos = w.last_pos;
		c.read = r; c.used = 1; c.start_it = it; c.mid_q = q; c.stop_it = ~0ull; c.stop_q = 0;
		cks[prev_slot].stop_it = it; cks[prev_slot].stop_q = q; prev_slot = k; last_k = k;
	};
	for (;; ++it)
	{
		if (sp < 0 || (it == stop && stop_q == 0)) break;
		const bool mid = mid_q != 0;                                   // resuming inside the script of this fragment
		if (!WRITE && cks) save_state(0);
		const LevelV& L = T.lv[sp];
		const FrameRec& F = L.frames[sf[sp]];
		const CandEnt& M = L.cands[F.cand_base + F.level];
		if (enter) { w.level = F.level; w.ref_id = M.ref_id; w.rev = M.rev; w.last_pos = s_last[sp]; enter = false; }
		const uint32_t n_frag = 2 * M.n + 1;
		if (!mid && si[sp] == n_frag)
		{	// final StoreFrag of the frame (encoder.cpp:1574)
			if (w.store()) s_last[sp] = s_cur[sp];
			--sp;
			if (sp >= 0)
			{	// back in the parent: its segment restarts; the reference part the child replaced is skipped (encoder.cpp:1489)
				const LevelV& Lp = T.lv[sp]; const FrameRec& Fp = Lp.frames[sf[sp]]; const CandEnt& Mp = Lp.cands[Fp.cand_base + Fp.level];
				w.level = Fp.level; w.ref_id = Mp.ref_id; w.rev = Mp.rev; w.last_pos = s_last[sp];
				const GapRec& g = Lp.gaps[Fp.first_gap + (si[sp] - 1) / 2];
				w.add('D', g.d_after);
			}
			continue;
		}
		const uint32_t i = mid ? si[sp] - 1 : si[sp]++;
		if (i & 1)
		{
			uint32_t al, ape, apr; cand_anchor(M, anchors_data, i >> 1, al, ape, apr);
			w.add('M', al);
			s_cur[sp] = apr + al;
			continue;
		}
		const GapRec& g = L.gaps[F.first_gap + (i >> 1)];
		const bool as_es = g.state == GS_ES || (g.state == GS_PENDING && L.dec[g.aux] != 0);
		if (as_es)
		{
			if (!mid) w.add('D', g.d_before);
			if (!WRITE && L.sums && g.es_len && g.es_len <= SUM_LIMIT)
			{	// count pass: the script from its summary
				const GapSum sm = L.sums[F.first_gap + (i >> 1)];
				w.add((char)(sm.syms & 0xff), sm.first_len);
				if (!(sm.syms >> 16)) { w.flush_run(); o.n += sm.mid_bytes; o.n_tuples += sm.mid_tuples; w.sym = (char)((sm.syms >> 8) & 0xff); w.rep = sm.last_len; }
				continue;
			}
			const uint32_t* es4 = (const uint32_t*)(L.es + g.es_off);               // script slots are dword-aligned
			bool stopped = false;
			for (uint32_t q = mid_q; q < g.es_len; q += 32)
			{	// eight independent loads in flight, then 32 symbols from registers (the walk is latency-bound otherwise)
				if (q && q != mid_q)
				{
					if (!WRITE && cks) save_state(q);
					if (WRITE && it == stop && q == stop_q) { stopped = true; break; }
				}
				uint32_t wd[8];
#pragma unroll
				for (uint32_t t = 0; t < 8; ++t) wd[t] = q + 4 * t < g.es_len ? es4[(q >> 2) + t] : 0u;
				const uint32_t nb = g.es_len - q < 32 ? g.es_len - q : 32;
				// ONE call site of add() (its inlined body is large; 32 copies would not fit the instruction cache): the 32
				// bytes go through a 256-bit shift register
				uint64_t x0 = wd[0] | ((uint64_t)wd[1] << 32), x1 = wd[2] | ((uint64_t)wd[3] << 32), x2 = wd[4] | ((uint64_t)wd[5] << 32), x3 = wd[6] | ((uint64_t)wd[7] << 32);
#pragma nounroll
				for (uint32_t t = 0; t < nb;)
				{	// a run of matches (up to 8) at once, else one symbol
					const uint64_t nm = x0 ^ 0x4d4d4d4d4d4d4d4dull;
					uint32_t cnt = nm ? (uint32_t)__builtin_ctzll(nm) >> 3 : 8u;
					if (cnt > nb - t) cnt = nb - t;
					if (cnt == 0) cnt = 1;
					const char c = (char)(x0 & 0xff);
					if (cnt == 8) { x0 = x1; x1 = x2; x2 = x3; x3 = 0; }
					else { const uint32_t sh = 8 * cnt; x0 = (x0 >> sh) | (x1 << (64 - sh)); x1 = (x1 >> sh) | (x2 << (64 - sh)); x2 = (x2 >> sh) | (x3 << (64 - sh)); x3 >>= sh; }
					w.add(c, cnt);
					t += cnt;
				}
			}
			mid_q = 0;
			if (stopped) break;
		}
		else if (g.state == GS_CHILD)
		{	// StoreFrag of what the parent has so far, then the child frame (encoder.cpp:1483-1488)
			if (w.store()) s_last[sp] = s_cur[sp];
			++sp; sf[sp] = g.aux; si[sp] = 0; s_last[sp] = 0; s_cur[sp] = 0;
			enter = true;
		}
		else
		{	// literal: the read part as insertions, then the reference part skipped (encoder.cpp:1497-1508)
			for (uint32_t q = 0; q < g.ne; ++q) w.add(base_letter(arena_base_at(A, wb, g.enc_start + q)), 1);
			w.add('D', g.d_after);
		}
	}
	o.finish();
#ifdef CL_EMIT_DEBUG
	if (WRITE && ck) atomicAdd(&g_emit_dbg[0], (unsigned long long)(o.n - (ck->start_it ? ck->n : 0))), atomicAdd(&g_emit_dbg[1], 1ull);
#endif
	if (!WRITE) { sizes[r] = (uint32_t)o.n; ntuples[r] = o.n_tuples; }
}

// ---- stage helpers shared by the kernels and the debugging host build -------------------------------------------------
CL_DEV inline CandEnt cand_level0(const uint32_t* cand4, uint64_t off, const uint32_t* data)
{
	CandEnt e; e.ref_id = cand4[0]; e.rev = cand4[1]; e.tot = cand4[2]; e.n = cand4[3]; e.off = off; e.shift = 0; e.pad = 0;
	for (int k = 0; k < 3; ++k) { e.f[k] = e.n ? data[3 * off + k] : 0; e.l[k] = e.n ? data[3 * (off + e.n - 1) + k] : 0; }
	return e;
}
CL_DEV inline bool gap_init(const LevelV& L, uint32_t frame, uint32_t g_idx, const uint32_t* data, const ArenaV& R, GapRec& g)
{
	const FrameRec F = L.frames[frame];
	const CandEnt M = L.cands[F.cand_base + F.level];
	g.frame = frame; g.read = F.read; g.ref_rev = M.ref_id | (M.rev ? 0x80000000u : 0u);
	if (gap_geometry(F, M, data, R.lens[M.ref_id], g_idx, g)) return true;
	g.g = g_idx; g.cur_ref = 0; g.enc_start = F.enc_off; g.nr = g.ne = g.use = g.d_after = g.d_before = g.es_len = 0; g.state = GS_ES; g.aux = 0; g.es_off = 0; g.kind = GK_TRIVIAL; g.left = 0;
	return false;
}
CL_DEV inline uint32_t gap_es_capacity(const GapRec& g) { return ((g.kind == GK_TRIVIAL ? (g.nr == 0 ? g.ne : 0u) : g.use + g.ne) + 3u) & ~3u; }   // dword-aligned script slots
// size class of a gap: 0 trivial, 1..4 small with that many 64-row blocks, 5 up to 16 row blocks with a history of at most
// 512 KB (four of them share a wave, k_align_quad; edlib keeps the whole history of such a gap: 20 B * blocks * columns + 8 B *
// columns stays under its 1 MiB), 6 large (a wave each), 7 giant (a work-group each, align_team.hpp): more rows than one tile of 64
// blocks, at least 2^19 block-columns, and not so many more rows than columns that the sweeps saturate (wv::sat_rows)
constexpr uint32_t QUAD_ROWS = 1024, QUAD_CELLS = 32768, QUAD_SEQ = 2048;   // (QUAD_SEQ: what a 16-lane row keeps in LDS, align_rows.hpp)
constexpr uint32_t GIANT_ROWS = 4096, GIANT_MAX_ROWS = 64 * 4096, GIANT_WORK = 1u << 19;
// sort key = class << 17 | row blocks (9 bits) << 8 | columns / 16 (8 bits): lanes of a wave get gaps of like shape
constexpr uint32_t N_CLASSES = 8, KEY_BITS = 20;
CL_DEV inline uint32_t gap_class(const GapRec& g, uint32_t& rows, uint32_t& cols)
{
	if (g.kind == GK_TRIVIAL) { rows = cols = 0; return 0; }
	if (g.kind == GK_FLANK) { rows = g.ne; cols = g.use; } else { rows = g.use; cols = g.ne; }
	if (rows <= 256 && cols <= 256) return (rows + 63) / 64;
	if (rows <= QUAD_ROWS && rows + cols <= QUAD_SEQ && (uint64_t)((rows + 63) / 64) * (cols + 16) <= QUAD_CELLS) return 5;
	if (rows > GIANT_ROWS && rows <= GIANT_MAX_ROWS && (uint64_t)((rows + 63) / 64) * cols >= GIANT_WORK && rows / 8 < cols) return 7;
	return 6;
}
CL_DEV inline uint32_t gap_sort_key(const GapRec& g)
{
	uint32_t rows, cols; const uint32_t cls = gap_class(g, rows, cols);
	uint32_t a, b;
	if (cls >= 6) { const uint64_t w = (uint64_t)((rows + 63) / 64) * cols; a = 0; b = 0; uint64_t x = w >> 8; while (x) { ++b; x >>= 1; } a = b; b = 0; }   // log2 of the work
	else { a = (rows + 63) / 64; b = cols >> 4; }
	return (cls << 17) | ((a > 511 ? 511u : a) << 8) | (b > 255 ? 255u : b);
}
// sequences of a small gap into the lane's staging memory, in alignment orientation (reversed for the left flank)
template<class MEM>
CL_DEV inline void stage_small(MEM& mem, const GapRec& g, const ArenaV& A, const ArenaV& R, uint32_t& n, uint32_t& m)
{
	const uint32_t ref_id = g.ref_rev & 0x7fffffffu; const bool rev = g.ref_rev >> 31;
	const uint64_t rwb = R.word_off[ref_id], ewb = A.word_off[g.read]; const uint32_t rlen = R.lens[ref_id];
	const bool left = g.left != 0;
	const uint32_t rbase = g.cur_ref + (left ? g.nr - 1 : 0);
	if (g.kind == GK_FLANK)
	{
		n = g.ne; m = g.use;
		for (uint32_t i = 0; i < n; ++i) mem.q_set(i, arena_base_at(A, ewb, g.enc_start + (left ? g.ne - 1 - i : i)));
		for (uint32_t j = 0; j < m; ++j) mem.t_set(j, ref_sym(R, rwb, rlen, rev, left ? rbase - j : rbase + j));
	}
	else
	{
		n = g.use; m = g.ne;
		for (uint32_t i = 0; i < n; ++i) mem.q_set(i, ref_sym(R, rwb, rlen, rev, left ? rbase - i : rbase + i));
		for (uint32_t j = 0; j < m; ++j) mem.t_set(j, arena_base_at(A, ewb, g.enc_start + (left ? g.ne - 1 - j : j)));
	}
}
// a gap on the generic path; false when the lane's pool was too small (g untouched then)
CL_DEV inline bool align_large_gap(LanePool& pool, GapRec& g, const ArenaV& A, const ArenaV& R, char* dst)
{
	const uint32_t ref_id = g.ref_rev & 0x7fffffffu; const bool rev = g.ref_rev >> 31;
	const uint64_t rwb = R.word_off[ref_id], ewb = A.word_off[g.read]; const uint32_t rlen = R.lens[ref_id];
	uint8_t* rbuf = (uint8_t*)pool.alloc(g.use + 16ull); uint8_t* ebuf = (uint8_t*)pool.alloc(g.ne + 16ull);
	if (pool.overflow) return false;
	const uint32_t lo = g.left ? g.nr - g.use : 0;
	for (uint32_t i = 0; i < g.use; ++i) rbuf[i] = (uint8_t)ref_sym(R, rwb, rlen, rev, g.cur_ref + lo + i);
	for (uint32_t i = 0; i < g.ne; ++i) ebuf[i] = (uint8_t)arena_base_at(A, ewb, g.enc_start + i);
	uint32_t d_before = 0;
	const uint32_t k = align_large(pool, rbuf, g.nr, g.use, ebuf, g.ne, g.kind, g.left != 0, dst, &d_before);
	if (pool.overflow) return false;
	g.es_len = k; g.d_before = d_before;
	return true;
}
// after alignment: trivial scripts, statistics / static decision.  Returns true when the gap was rejected (spawn candidate).
CL_DEV inline bool gap_finish(const LevelV& L, uint32_t gi, const ArenaV& A, const EncCfg& cfg, uint32_t pend_idx)
{
	GapRec g = L.gaps[gi];
	char* es = L.es + g.es_off;
	const uint64_t ewb = A.word_off[g.read];
	if (g.kind == GK_TRIVIAL)
	{	// get_edit_dist_on_seq_empty (edit_script.h:250-267)
		if (g.nr == 0) { for (uint32_t i = 0; i < g.ne; ++i) es[i] = base_letter(arena_base_at(A, ewb, g.enc_start + i)); g.es_len = g.ne; }
		else g.d_before = g.nr;
	}
	gap_stats(g, es, A, ewb, cfg, L.pend + pend_idx);
	if (g.state == GS_PENDING) g.aux = pend_idx;
	L.gaps[gi] = g;
	return g.state == GS_REJECTED;
}
CL_DEV inline void spawn_child(const LevelV& L, uint32_t gi, const uint32_t* data, const EncCfg& cfg, uint32_t child, uint64_t cand_base, FrameRec* frames_next, CandEnt* cands_next)
{
	const GapRec g = L.gaps[gi];
	const FrameRec F = L.frames[g.frame];
	CandEnt out[16];
	spawn_cands(F, L.cands + F.cand_base, data, g, cfg, out);
	FrameRec C; C.read = F.read; C.level = F.level + 1; C.enc_off = g.enc_start; C.enc_len = g.ne; C.n_cands = F.n_cands; C.first_gap = 0; C.n_gaps = 0; C.pad = 0; C.cand_base = cand_base;
	frames_next[child] = C;
	for (uint32_t j = 0; j < F.n_cands; ++j) cands_next[cand_base + j] = out[j];
	L.gaps[gi].state = GS_CHILD; L.gaps[gi].aux = child;
}
CL_DEV inline void est_pack(const TreeV& T, uint32_t r_begin, uint32_t r_end, const uint32_t* lens, const uint8_t* has_n, const uint32_t* base_counts)
{
	Estim e; est_reset(e);
	for (uint32_t r = r_begin; r < r_end; ++r)
	{
		if (has_n[r]) continue;                                                  // reads with N never reach the estimator (encoder.cpp:1629-1633)
		est_log_read(e, base_counts + 4 * r, lens[r]);
		const uint32_t f0 = T.frame_of_read[r];
		if (f0 != 0xffffffffu) est_read(e, T, f0);
	}
}

} // namespace enc
