// align_rows.hpp — the quad class (rows <= 1024, rows + columns <= 2048): FOUR gaps per wave, each entirely in its 16-lane row (round 5).
// Same observable behaviour as align_wave.hpp / align_dev.hpp (edlib as the reference calls it, edit_script.h:272-413; edlib.cpp:141-296,
// 945-1147; refactor_edit_script, edit_script.h:416-446,591-671).  Rounds 2-4 ran only the SWEEP four gaps at a time; staging, traceback,
// operations -> script and both canonicalisation passes then took the gaps one after the other with the whole wave, through byte buffers
// and per-symbol temporaries in HBM: of ~500 us per quad the sweep was 95.  Here every phase runs in the rows side by side:
//   * the two sequences of a gap are staged ONCE, 2 bits per symbol, into LDS (orientation and strand are index arithmetic);
//   * the sweep writes the history (vertical +1 / horizontal +1 words, one 16-byte pair per block and column) to the wave's pool in HBM;
//   * the traceback of the four gaps runs in lock step: a lane holds the pairs of one column for the two row blocks around the path,
//     the window of the next 16 columns is fetched while the current one is walked; a RUN of cells per iteration (row ballots);
//   * operations, script and the per-step summaries of the canonicalisation live in LDS; the script leaves as whole words.
// All control flow is wave-uniform; what differs between rows is predicated.
#pragma once
#include "align_wave.hpp"

namespace qr {

constexpr uint32_t SEQ_MAX = 2048;                                            // rows + columns of a gap of the class (enc::gap_class)
constexpr uint32_t SEQ_WORDS = SEQ_MAX / 16 + 4;                              // two 2-bit arrays, each rounded up to a word
constexpr uint32_t ROW_BYTES = SEQ_WORDS * 4 + 2 * SEQ_MAX;                   // sequences | operations (then step summaries) | script
static_assert(ROW_BYTES % 16 == 0, "rows of the LDS layout stay 16-byte aligned");

struct Row {                                                                  // uniform within a 16-lane row
	uint32_t* seq; uint8_t* ops; uint8_t* es;
	uint32_t use, ne, e_off;                                                  // reference / read symbols staged; word offset of the read's
	uint32_t n, m;                                                            // rows, columns of the alignment (0 x 0: the row has no gap)
	bool rows_ref, rev_seq, left, shw;
};
__device__ inline uint32_t sym_at(const uint32_t* w, uint32_t i) { return (w[i >> 4] >> (2 * (i & 15))) & 3u; }
__device__ inline uint32_t ref_at(const Row& r, uint32_t i) { return sym_at(r.seq, i); }
__device__ inline uint32_t enc_at(const Row& r, uint32_t i) { return sym_at(r.seq + r.e_off, i); }
// the sequences in alignment orientation (a left flank is aligned on the reversed sequences)
__device__ inline uint32_t Qs(const Row& r, uint32_t i) { return r.rows_ref ? ref_at(r, r.rev_seq ? r.use - 1 - i : i) : enc_at(r, r.rev_seq ? r.ne - 1 - i : i); }
__device__ inline uint32_t Ts(const Row& r, uint32_t j) { return r.rows_ref ? enc_at(r, r.rev_seq ? r.ne - 1 - j : j) : ref_at(r, r.rev_seq ? r.use - 1 - j : j); }

__device__ inline uint32_t rb16(bool p, uint32_t row0) { return (uint32_t)(__ballot(p) >> row0) & 0xffffu; }
__device__ inline uint32_t max4(uint32_t v) { const uint32_t a = wv::bcast(v, 0u), b = wv::bcast(v, 16u), c = wv::bcast(v, 32u), d = wv::bcast(v, 48u); const uint32_t x = a > b ? a : b, y = c > d ? c : d; return x > y ? x : y; }
__device__ inline void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }

// bases p0 .. p0 + 15 of a stored read (word base wb, last word lw), first base in bits 31..30; p0 may be as low as -15 (bases < 0: zeros)
__device__ inline uint32_t take16(const uint64_t* __restrict__ packed, uint64_t wb, uint32_t lw, int64_t p0)
{
	uint32_t shr = 0;
	if (p0 < 0) { shr = (uint32_t)(-p0) * 2; p0 = 0; }
	if (shr >= 32) return 0;
	const uint32_t w = (uint32_t)p0 >> 5, sh = 2 * ((uint32_t)p0 & 31);
	const uint64_t a = packed[wb + (w < lw ? w : lw)], b = packed[wb + (w + 1 < lw ? w + 1 : lw)];
	const uint64_t hi = sh ? (a << sh) | (b >> (64 - sh)) : a;
	return (uint32_t)(hi >> 32) >> shr;
}
__device__ inline uint32_t lsb_first(uint32_t x) { x = __builtin_bitreverse32(x); return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); }

// the gap's reference stretch (g.use symbols from g.cur_ref + lo, in the candidate's orientation) and read stretch into the row's LDS
__device__ inline void row_stage(const Row& r, bool act, const enc::GapRec& g, const enc::ArenaV& A, const enc::ArenaV& R)
{
	const uint32_t bl = wv::lane_id() & 15;
	if (!act) return;
	const uint32_t ref_id = g.ref_rev & 0x7fffffffu; const bool rc = g.ref_rev >> 31;
	const uint64_t rwb = R.word_off[ref_id], ewb = A.word_off[g.read];
	const uint32_t rlen = R.lens[ref_id], elen = A.lens[g.read];
	const uint32_t rlw = rlen ? (rlen - 1) >> 5 : 0, elw = elen ? (elen - 1) >> 5 : 0;
	const uint32_t lo = g.left ? g.nr - g.use : 0;
	for (uint32_t w = bl; w * 16 < r.use; w += 16)
	{
		const uint32_t pos0 = g.cur_ref + lo + w * 16;
		r.seq[w] = rc ? ~take16(R.packed, rwb, rlw, (int64_t)rlen - 1 - pos0 - 15) : lsb_first(take16(R.packed, rwb, rlw, pos0));
	}
	for (uint32_t w = bl; w * 16 < r.ne; w += 16) r.seq[r.e_off + w] = lsb_first(take16(A.packed, ewb, elw, (int64_t)g.enc_start + w * 16));
}

// ---- the sweep (as wv::quad_sweep; sequences from LDS, the history as 16-byte pairs) -------------------------------------------------------
// The history is what this kernel moves: one pair { vertical +1 bits after the column, horizontal +1 bits of the column } per row block and
// column — measured (round 5) at 4 TB/s of writes with 2048 waves sweeping, i.e. the sweep ran at the speed of its stores, and the
// traceback's 16-byte reads with a stride of nb pairs fetched a whole sector each.  So:
//   * BLOCK-MAJOR: cell of block b, column j (0-based) = pair b * stride + j — the 16 columns of a traceback window are 256 contiguous bytes;
//   * a BAND: every cell (i, j) of an optimal path has D[i][j] <= d (the final distance) and D[i][j] >= |i - j|, so a block is stored for
//     column j only if it holds a row within `band` of j.  `band` is an assumption (a quarter of the gap's length: 99.3 % of the class, the
//     rest are unrelated sequences at d > len / 2); the caller checks d <= band after the sweep and hands the gap to the wave kernel
//     otherwise — the sweep itself is exact either way, only what it keeps is limited.
using enc::sel64;
__device__ inline uint32_t hist_stride(uint32_t m) { return (m + 15) & ~15u; }
__device__ inline wv::Sweep row_sweep(const Row& r, ulonglong2* __restrict__ hist, uint32_t band)
{
	const uint32_t lane = wv::lane_id(), bl = lane & 15, row0 = lane & 48;
	const uint32_t n = r.n, m = r.m, nb = (n + 63) / 64;
	const bool act = bl < nb;
	uint64_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
	if (act)
	{
		const uint32_t lo = bl * 64, hi = n < lo + 64 ? n : lo + 64;
		for (uint32_t i = lo; i < hi; ++i)
		{
			const uint32_t s = Qs(r, i); const uint64_t bit = 1ull << (i - lo);
			e0 |= s == 0 ? bit : 0; e1 |= s == 1 ? bit : 0; e2 |= s == 2 ? bit : 0; e3 |= s == 3 ? bit : 0;
		}
	}
	// The step is one straight line (round 5: hipcc turned the four-way choice of the match mask and the owner's score update into
	// branches — 110 instructions and six exec-mask switches per step): the mask is chosen by bit selects, the horizontal delta travels
	// as two bits (1: +1, 2: -1), the score is read through a mask that is zero outside the lane that owns the last row.
	uint64_t Pv = ~0ull, Mv = 0;
	const uint64_t lastmask = (act && bl == nb - 1) ? 1ull << ((n - 1) & 63) : 0ull;
	uint32_t sc = n, best = 0xffffffffu; int32_t end = (int32_t)m - 1;
	if (r.shw && (n & 63)) { best = n; end = -1; }
	const bool shw = r.shw;
	const uint32_t steps = nb ? m + nb - 1 : 0;
	const uint32_t steps_max = max4(steps);
	ulonglong2* const hrow = hist + (uint64_t)bl * hist_stride(m) - bl;        // (+ s: column s - bl of block bl)
	// 1-based columns [keep_lo, keep_hi] of this block are kept: steps s with s - s_lo <= s_span
	const uint32_t keep_lo = 64 * bl + 1 > band ? 64 * bl + 1 - band : 1, keep_hi = 64 * bl + 64 + band;
	const uint32_t s_lo = keep_lo - 1 + bl, s_span = keep_hi - keep_lo;
	const uint32_t m_act = act ? m : 0;
	uint32_t c = 0, tchunk = 0, hout = 0;
	for (uint32_t s = 0; s < steps_max; ++s)
	{
		if ((s & 15) == 0) { const uint32_t j0 = s + bl; tchunk = (nb && j0 < m) ? Ts(r, j0) : 0u; }
		else tchunk = wv::row_rol1(tchunk);
		const uint32_t c_up = wv::row_shr1(c), h_up = wv::row_shr1(hout);
		c = bl == 0 ? tchunk : c_up;
		const uint32_t hin = bl == 0 ? 1u : h_up;
		hout = 0;
		if (s - bl < m_act)
		{
			const uint32_t M0 = 0u - (c & 1), M1 = 0u - (c >> 1);               // (32-bit masks on both halves: v_bfi_b32)
			uint64_t Eq = sel64(M1, sel64(M0, e3, e2), sel64(M0, e1, e0));
			const uint64_t hneg = hin >> 1, hpos = hin & 1;
			const uint64_t Xv = Eq | Mv;
			Eq |= hneg;
			const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
			uint64_t Ph = Mv | ~(Xh | Pv);
			uint64_t Mh = Pv & Xh;
			const uint64_t ph_rows = Ph;
			sc += (uint32_t)((Ph & lastmask) != 0) - (uint32_t)((Mh & lastmask) != 0);
			const bool better = shw && sc < best;                                // (lanes that do not own the last row: sc stays n, never below best)
			best = better ? sc : best; end = better ? (int32_t)(s - bl) : end;
			hout = (uint32_t)(Ph >> 63) | ((uint32_t)(Mh >> 63) << 1);
			Ph = (Ph << 1) | hpos; Mh = (Mh << 1) | hneg;
			Pv = Mh | ~(Xv | Ph);
			Mv = Ph & Xv;
			if (s - s_lo <= s_span) hrow[s] = make_ulonglong2(Pv, ph_rows);
		}
	}
	wv::Sweep out;
	const int own = (int)(row0 + (nb ? nb - 1 : 0));
	out.score = (uint32_t)__shfl((int)sc, own); out.best = (uint32_t)__shfl((int)best, own); out.end = __shfl(end, own);
	return out;
}

// ---- the traceback of the four gaps in lock step (decisions of edlib.cpp:1021-1147: up if the vertical delta is +1, else left if the
// horizontal delta is +1, else diagonal) -----------------------------------------------------------------------------------------------------
// A WINDOW = the pairs of 64 consecutive columns (sub-window x of lane d: column j0 - 16 x - d, 1-based) for two row blocks, b0 and b0 - 1:
// 16 registers a lane.  A fetch from HBM is a microsecond and hipcc waits for every load it has issued before the next use of any of
// them (measured: a queue of four 16-column windows requested ahead changed nothing — `s_waitcnt vmcnt(0)` every iteration), so what counts
// is the NUMBER of fetches the wave waits for: whenever one row leaves its window ALL rows take a new one at their current cell (one wait
// for the four of them; ~20 per quad instead of 62 with 16 columns and a fetch per row).  The 16 lanes of a row look at the 16 columns
// j, j - 1, ... j - 15 wherever they fall in the window, so a run is never cut at a sub-window's edge.
// The operations go to r.ops, LAST operation first; the walk ends in cell (i_out, j_out), one of them 0.
struct WalkStats { uint32_t iters = 0, miss = 0, hit = 0; };                  // (per wave: iterations, iterations that waited for new windows; `hit` unused)
__device__ inline void row_walk(const Row& r, const ulonglong2* __restrict__ hist, uint32_t j_start, uint32_t& i_out, uint32_t& j_out, uint32_t& k_out, WalkStats* ws = nullptr)
{
	const uint32_t lane = wv::lane_id(), d = lane & 15, row0 = lane & 48;
	const uint32_t n = r.n, stride = hist_stride(r.m);
	uint32_t i = n, j = n ? j_start : 0, k = 0;
	uint64_t P0[4], H0[4], P1[4], H1[4];                                        // [x]: sub-window x; 0: block b0, 1: block b0 - 1
#pragma unroll
	for (int x = 0; x < 4; ++x) P0[x] = H0[x] = P1[x] = H1[x] = 0;
	uint32_t j0 = 0, b0 = 0; bool have = false;
	for (;;)
	{
		const bool go = i > 0 && j > 0;
		if (!__ballot(go)) break;
		const uint32_t rr = i - 1, b = rr >> 6, rb = rr & 63;
		const bool need = go && !(have && (b == b0 || b + 1 == b0) && j <= j0 && j0 - j < 64);
		if (ws) ++ws->iters;
		if (__ballot(need))
		{
			if (ws) ++ws->miss;
			if (go)
			{
				j0 = j; b0 = b; have = true;
#pragma unroll
				for (int x = 0; x < 4; ++x)
				{
					P0[x] = H0[x] = P1[x] = H1[x] = 0;
					if (j0 > 16 * x + d)
					{
						const uint32_t jj = j0 - 16 * x - d - 1;
						const ulonglong2 a = hist[(uint64_t)b0 * stride + jj];
						P0[x] = a.x; H0[x] = a.y;
						if (b0) { const ulonglong2 c = hist[(uint64_t)(b0 - 1) * stride + jj]; P1[x] = c.x; H1[x] = c.y; }
					}
				}
			}
		}
		const uint32_t o = go ? j0 - j : 0u, sub = o >> 4, src = o & 15;           // column j sits in sub-window `sub`, lane `src`
		const uint32_t dd = (d - src) & 15;                                     // this lane's column is j - dd ...
		const uint32_t x = sub + (d < src ? 1u : 0u);                           // ... in sub-window x
		const bool col_ok = go && dd < j && x < 4;
		const bool lower = b == b0;
		uint64_t Pb, Hb;
		{
			const uint64_t pa = lower ? P0[0] : P1[0], pb = lower ? P0[1] : P1[1], pc = lower ? P0[2] : P1[2], pd = lower ? P0[3] : P1[3];
			const uint64_t ha = lower ? H0[0] : H1[0], hb = lower ? H0[1] : H1[1], hc = lower ? H0[2] : H1[2], hd = lower ? H0[3] : H1[3];
			Pb = x == 0 ? pa : x == 1 ? pb : x == 2 ? pc : pd;
			Hb = x == 0 ? ha : x == 1 ? hb : x == 2 ? hc : hd;
		}
		const uint32_t pr = (uint32_t)(Pb >> rb) & 1, hr = (uint32_t)(Hb >> rb) & 1;
		auto rot = [&](uint32_t m16) { return ((m16 >> src) | (m16 << (16 - src))) & 0xffffu; };   // bit of lane d -> bit dd
		const uint32_t pm = rot(rb16(col_ok && pr, row0)), hm = rot(rb16(col_ok && hr, row0));
		const uint32_t p0 = pm & 1, h0 = hm & 1;
		// up: while the vertical +1 bits of column j continue (inside this block) — every lane for its own column, the row takes lane src's
		uint32_t myrun;
		{ const uint64_t y = ~Pb << (63 - rb); myrun = y ? (uint32_t)__builtin_clzll(y) : 64u; if (myrun > rb + 1) myrun = rb + 1; }
		const uint32_t uprun = (uint32_t)__shfl((int)myrun, (int)(row0 + src));
		const uint32_t left16 = rot(rb16(col_ok && !pr && hr, row0));
		const uint32_t bp = (rb - dd) & 63;
		const bool dg = col_ok && dd <= rb && !((Pb >> bp) & 1) && !((Hb >> bp) & 1);
		const uint32_t diag16 = rot(rb16(dg, row0));
		if (go)
		{
			uint32_t run;
			if (p0)
			{
				run = uprun;
				for (uint32_t y = d; y < run; y += 16) r.ops[k + y] = 1;
				i -= run;
			}
			else if (h0)
			{	// left while row rr has no vertical +1 and a horizontal +1
				const uint32_t stop = ~left16 & 0xffffu;
				run = stop ? (uint32_t)__builtin_ctz(stop) : 16u;
				if (dd < run) r.ops[k + dd] = 2;
				j -= run;
			}
			else
			{	// diagonal while neither bit is set at (rr - dd, j - dd); the symbols decide match / mismatch
				const uint32_t stop = ~diag16 & 0xffffu;
				run = stop ? (uint32_t)__builtin_ctz(stop) : 16u;
				if (dd < run) r.ops[k + dd] = Qs(r, rr - dd) == Ts(r, j - 1 - dd) ? 0 : 3;
				i -= run; j -= run;
			}
			k += run;
		}
	}
	i_out = i; j_out = j; k_out = k;
}

// ---- operations (0 match, 1 consume query, 2 consume target, 3 mismatch) -> script symbols, 16 per step and row ------------------------------
// forward operations = `pre` x pre_op, then the walk's in reverse; the script of a left flank is written reversed (edit_script.h:272-413)
__device__ inline void row_convert(const Row& r, uint32_t pre, uint8_t pre_op, uint32_t k)
{
	const uint32_t lane = wv::lane_id(), bl = lane & 15, row0 = lane & 48;
	const uint32_t K = r.n ? pre + k : 0, lt = (1u << bl) - 1;
	const uint32_t steps_max = max4((K + 15) / 16);
	uint32_t pq = 0, pt = 0;
	for (uint32_t ch = 0; ch < steps_max; ++ch)
	{
		const uint32_t x = ch * 16 + bl; const bool valid = x < K;
		const uint32_t op = valid ? (x < pre ? (uint32_t)pre_op : (uint32_t)r.ops[k - 1 - (x - pre)]) : 4u;
		const uint32_t cq = rb16(op != 2 && op != 4, row0), ct = rb16(op != 1 && op != 4, row0);
		const uint32_t myq = pq + (uint32_t)__popc(cq & lt), myt = pt + (uint32_t)__popc(ct & lt);
		if (valid)
		{
			const uint32_t qs = (op == 1 || op == 3) ? Qs(r, myq) : 0u, ts = (op == 2 || op == 3) ? Ts(r, myt) : 0u;
			char c;
			if (op == 0) c = 'M';
			else if (op == 1) c = r.rows_ref ? 'D' : enc::base_letter(qs);
			else if (op == 2) c = r.rows_ref ? enc::base_letter(ts) : 'D';
			else c = r.rows_ref ? enc::mismatch_sym(qs, ts) : enc::mismatch_sym(ts, qs);
			r.es[r.left ? K - 1 - x : x] = (uint8_t)c;
		}
		pq += (uint32_t)__popc(cq); pt += (uint32_t)__popc(ct);
	}
}

// ---- refactor_edit_script (edit_script.h:416-446,591-671), one pass: every maximal REGION — consecutive script symbols that are neither a
// break (pass 1: insertion / substitution, pass 2: deletion / substitution) nor step onto a different sequence symbol than their predecessor —
// becomes its matches first, then its other symbols (wv::wave_refactor_pass has the derivation).  Forward sweep: region heads, matches and the
// carries into every 16-symbol step, kept as a 16-byte summary per step (in r.ops, free by now); backward sweep: matches after, rewrite.
__device__ inline void row_refactor_pass(const Row& r, uint32_t k_row, uint32_t seq_off, int pass)
{
	const uint32_t lane = wv::lane_id(), bl = lane & 15, row0 = lane & 48;
	const uint32_t k = r.n ? k_row : 0;
	const uint32_t n_steps = (k + 15) / 16, steps_max = max4(n_steps);
	const uint32_t lt = (1u << bl) - 1, le = (2u << bl) - 1;
	uint16_t* const sm = (uint16_t*)r.ops;
	auto is_ins = [](uint32_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; };
	auto is_mis = [](uint32_t c) { return c == 'X' || c == 'Y' || c == 'Z'; };
	uint32_t pos_base = 0, c_sym = 0xff, c_start = 0, c_m = 0; bool c_reg = false;
	for (uint32_t ch = 0; ch < steps_max; ++ch)
	{
		const uint32_t x = ch * 16 + bl; const bool valid = x < k;
		const uint32_t c = valid ? (uint32_t)r.es[x] : (uint32_t)' ';
		const bool ins = is_ins(c), mis = is_mis(c), del = c == 'D';
		const bool cons = valid && (pass == 1 ? !ins : !del), reg = valid && (pass == 1 ? !(ins || mis) : !(del || mis));
		const uint32_t C16 = rb16(cons, row0);
		const uint32_t pos = pos_base + (uint32_t)__popc(C16 & lt);
		const uint32_t sym = reg ? (pass == 1 ? ref_at(r, seq_off + pos) : enc_at(r, pos)) : 0xffu;
		uint32_t p_sym = wv::row_shr1(sym); bool p_reg = wv::row_shr1((int)reg) != 0;
		if (bl == 0) { p_sym = c_sym; p_reg = c_reg; }
		const bool head = reg && (!p_reg || p_sym != sym);
		const uint32_t H16 = rb16(head, row0), R16 = rb16(reg, row0), M16 = rb16(reg && c == 'M', row0);
		const uint32_t S0 = rb16(reg && (sym & 1), row0), S1 = rb16(reg && (sym & 2), row0);
		if (bl == 0 && ch < n_steps)
		{
			uint4 v; v.x = H16 | (R16 << 16); v.y = M16 | (C16 << 16); v.z = c_start | (c_m << 16); v.w = pos_base;
			*(uint4*)(sm + 8 * ch) = v;
		}
		if (ch < n_steps)
		{	// the carries: the state of the step's last symbol
			if ((R16 >> 15) & 1)
			{
				if (H16) { const uint32_t h = 31 - (uint32_t)__builtin_clz(H16); c_start = ch * 16 + h; c_m = (uint32_t)__popc(M16 >> h); }
				else c_m += (uint32_t)__popc(M16);
				c_reg = true; c_sym = ((S0 >> 15) & 1) | (((S1 >> 15) & 1) << 1);
			}
			else { c_reg = false; c_sym = 0xff; c_start = 0; c_m = 0; }
			pos_base += (uint32_t)__popc(C16);
		}
	}
	lds_fence();
	uint32_t c_ma = 0;                                                          // matches from the start of the later steps up to their first boundary
	for (uint32_t ch = steps_max; ch-- > 0;)
	{
		const bool act = ch < n_steps;
		const uint32_t x = ch * 16 + bl;
		uint4 v = make_uint4(0, 0, 0, 0);
		if (act) v = *(const uint4*)(sm + 8 * ch);
		const uint32_t H16 = v.x & 0xffffu, R16 = v.x >> 16, M16 = v.y & 0xffffu, C16 = v.y >> 16, cs = v.z & 0xffffu, cm = v.z >> 16, pb = v.w;
		const uint32_t Bd = (H16 | ~R16) & 0xffffu;                               // boundary = region head or not a region symbol
		const uint32_t above = Bd & ~le & 0xffffu;
		uint32_t ma;
		if (above) { const uint32_t e = (uint32_t)__builtin_ctz(above); ma = (uint32_t)__popc(M16 & ~le & ((1u << e) - 1)); }
		else ma = (uint32_t)__popc(M16 & ~le & 0xffffu) + c_ma;
		if (act && x < k && ((R16 >> bl) & 1))
		{
			const uint32_t below = H16 & le;
			uint32_t start, m_incl;
			if (below) { const uint32_t h = 31 - (uint32_t)__builtin_clz(below); start = ch * 16 + h; m_incl = (uint32_t)__popc(M16 & le & ~((1u << h) - 1)); }
			else { start = cs; m_incl = cm + (uint32_t)__popc(M16 & le); }
			uint8_t other = (uint8_t)'D';
			if (pass == 2) other = (uint8_t)enc::base_letter(enc_at(r, pb + (uint32_t)__popc(C16 & lt)));
			r.es[x] = (x - start) < m_incl + ma ? (uint8_t)'M' : other;
		}
		if (act)
		{
			if (Bd) { const uint32_t e0 = (uint32_t)__builtin_ctz(Bd); c_ma = (uint32_t)__popc(M16 & ((1u << e0) - 1)); }
			else c_ma += (uint32_t)__popc(M16);
		}
	}
	lds_fence();
}

} // namespace qr
