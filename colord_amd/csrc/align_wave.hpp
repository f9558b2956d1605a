// align_wave.hpp — one WAVE aligns one large gap (device only).  Same observable behaviour as align_dev.hpp (edlib as
// the reference calls it, edit_script.h:272-413; edlib.cpp:141-296,547-700,945-1400), organised for a 64-wide wavefront:
//   * the 64-row blocks of Myers' recurrence are spread over the lanes and swept along the anti-diagonals: at step s
//     lane l advances block l of the current 64-block tile through column s - l; the horizontal delta leaving a block
//     reaches the lane below with one DPP shift, the column symbol travels the same way; tiles hand their last block's
//     deltas over through a byte array;
//   * the traceback (up, then left, then diagonal, edlib.cpp:1021-1147) is sequential by nature; the wave fetches the
//     history words of 64 consecutive columns at once and walks them with lane reads;
//   * Hirschberg (edlib.cpp:1230-1400) when the history would exceed edlib's 1 MiB budget: both half sweeps deliver their
//     last column, the split row is found with a ballot.
// All control flow is wave-uniform; memory comes from a per-wave bump pool in HBM.
#pragma once
#include "common.hpp"

namespace wv {

struct WavePool {
	uint8_t* base; uint64_t cap, top; bool overflow; volatile uint32_t* hb;       // hb: optional host-visible progress word (debugging)
	unsigned long long* prof = nullptr; uint64_t t_last = 0;                      // optional per-phase clock accumulation (debugging)
	uint64_t gp[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };                                  // the same per gap (reset by the kernel): phases of the slowest gap
	__device__ inline void lap(uint32_t phase) { if (prof) { const uint64_t now = wall_clock64(); gp[phase & 7] += now - t_last; if ((threadIdx.x & 63) == 0) atomicAdd(prof + phase, (unsigned long long)(now - t_last)); t_last = now; } }
	__device__ inline void beat(uint32_t code) { if (hb && (threadIdx.x & 63) == 0) *hb = code; }
	__device__ inline void* alloc(uint64_t bytes)
	{
		bytes = (bytes + 255) & ~255ull;
		if (top + bytes > cap) { overflow = true; return base; }
		void* p = base + top; top += bytes; return p;
	}
	__device__ inline uint64_t mark() const { return top; }
	__device__ inline void release(uint64_t m) { top = m; }
};

__device__ inline uint32_t lane_id() { return threadIdx.x & 63; }
// broadcast from a lane whose index is the same in every lane: v_readlane (a few cycles) instead of a permute through LDS
__device__ inline uint32_t bcast(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane(src)); }
__device__ inline int bcast(int v, uint32_t src) { return __builtin_amdgcn_readlane(v, (int)__builtin_amdgcn_readfirstlane(src)); }
__device__ inline uint64_t bcast(uint64_t v, uint32_t src)
{
	const int s = (int)__builtin_amdgcn_readfirstlane(src);
	return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), s) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, s);
}
__device__ inline uint32_t bcast_first(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// lane l receives the value of lane l - 1 (lane 0: 0): one DPP move (wave_shr:1) instead of a permute through LDS — the
// sweep is a dependent chain of such shifts
__device__ inline int shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }
__device__ inline uint32_t shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }

struct Hist { uint64_t* P; uint64_t* H; uint32_t W, m, rows; };  // cell (tile, s, lane) at ((tile * (m + 64)) + s) * W + lane; rows: the rows the sweep computed (sat_rows)
__device__ inline uint64_t hist_words(uint32_t nb, uint32_t m) { const uint32_t tiles = (nb + 63) / 64, W = nb < 64 ? nb : 64; return (uint64_t)tiles * (m + 64) * W; }

struct Sweep { uint32_t score; uint32_t best; int32_t end; };

// ---- saturation: the rows a sweep has to compute ---------------------------------------------------------------------------
// D[i][j] >= i - j (a column absorbs at most one row on a diagonal), with equality exactly when t[0..j) is a SUBSEQUENCE of
// q[0..i).  Let s_j be the row where the greedy embedding of t[0..j) in q ends (s_0 = 0 < s_1 < ...).  Then for every row
// i >= s_j: D[i][j] = i - j, so below row s_j the column is known without any recurrence — vertical delta +1, horizontal delta
// -1.  The typical gap of the wave class is a read flank of 10^3..10^5 symbols against the few hundred symbols that are left of
// the reference read (encoder.cpp:1262-1273: at most 2 x the flank, but the reference read ends): rows >> columns, and the rows
// below s_m (about 1.2 m for related sequences, about 4 m for unrelated ones) were 90..99 % of the swept cells.
// Returns the number of rows to compute: n itself, or a multiple of 64 with s_m <= rows < n (then the rest is closed form).
__device__ inline uint32_t sat_rows(const uint8_t* q, int qstep, uint32_t n, const uint8_t* t, int tstep, uint32_t m)
{
	if (n < m + 64) return n;
	const uint32_t lane = lane_id();
	uint32_t p = 0;                                                             // symbols of t embedded so far (wave-uniform)
	uint32_t tch = lane < m ? (uint32_t)(t[(int64_t)lane * tstep] & 3) : 0u;    // lanes hold t[(p & ~63) + lane]
	for (uint32_t i0 = 0; i0 < n; i0 += 64)
	{
		const uint32_t i = i0 + lane;
		const uint32_t s = i < n ? (uint32_t)(q[(int64_t)i * qstep] & 3) : 4u;
		const uint64_t e0 = __ballot(s == 0), e1 = __ballot(s == 1), e2 = __ballot(s == 2), e3 = __ballot(s == 3);
		uint32_t pos = 0;                                                       // rows of this chunk consumed
		while (p < m)
		{
			const uint32_t c = bcast(tch, p & 63);
			uint64_t mk = c == 0 ? e0 : c == 1 ? e1 : c == 2 ? e2 : e3;
			mk &= ~0ull << pos;
			if (!mk) break;
			pos = (uint32_t)__builtin_ctzll(mk) + 1; ++p;
			if ((p & 63) == 0) { const uint32_t j = p + lane; tch = j < m ? (uint32_t)(t[(int64_t)j * tstep] & 3) : 0u; }
			if (pos == 64) break;
		}
		if (p == m) return i0 + 64 < n ? i0 + 64 : n;
	}
	return n;
}

// One tile (64 row blocks, one per lane) of a sweep over all columns; the tiles of a sweep hand the horizontal deltas of their
// last block over through byte arrays.  `prog_in` / `prog_out` (team sweeps, align_team.hpp: the tiles of one sweep run on
// different waves of a work-group, each a little behind the one above it): LDS words counting the columns whose deltas
// are written — waited for before a 64-column chunk of `hin` is read, published every 64 steps.
struct TileState { uint32_t sc, best; int32_t end; };                         // meaningful in the lane that owns the last block
__device__ inline void sweep_tile(uint32_t tile, const uint8_t* q, int qstep, uint32_t n, uint32_t ne, const uint8_t* t, int tstep, uint32_t m, bool shw, bool sat,
                                  const Hist* hist, int32_t* lastcol, const int8_t* hin_arr, int8_t* hout_arr, volatile uint32_t* prog_in, volatile uint32_t* prog_out, TileState& st)
{
	const uint32_t lane = lane_id();
	const uint32_t nb = (ne + 63) / 64;
	const uint32_t lastbit = (n - 1) & 63;
	const uint32_t b = tile * 64 + lane; const bool act = b < nb;
	const uint32_t W = nb - tile * 64 < 64 ? nb - tile * 64 : 64;
	uint64_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
	for (uint32_t bb = 0; bb < W; ++bb)
	{	// the match masks of block tile * 64 + bb: one coalesced load of its 64 row symbols, four ballots; lane bb keeps them
		const uint32_t i = (tile * 64 + bb) * 64 + lane;
		const uint32_t s = i < ne ? (uint32_t)(q[(int64_t)i * qstep] & 3) : 4u;
		const uint64_t m0 = __ballot(s == 0), m1 = __ballot(s == 1), m2 = __ballot(s == 2), m3 = __ballot(s == 3);
		if (lane == bb) { e0 = m0; e1 = m1; e2 = m2; e3 = m3; }
	}
	uint64_t Pv = ~0ull, Mv = 0; int32_t S = (int32_t)((b + 1) * 64);
	const bool owner = !sat && act && b == nb - 1;
	uint32_t c = 0; int hout = 0; uint32_t tchunk = 0; int hchunk = 1;
	const uint32_t steps = m + W - 1;
	for (uint32_t s = 0; s < steps; ++s)
	{
		if ((s & 63) == 0)
		{
			const uint32_t j0 = s + lane;
			tchunk = j0 < m ? (uint32_t)(t[(int64_t)j0 * tstep] & 3) : 0u;
			if (prog_in && s < m)
			{	// the tile above has to be through columns s .. s + 63
				const uint32_t need = s + 64 < m ? s + 64 : m;
				while (*prog_in < need) __builtin_amdgcn_s_sleep(4);
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			}
			hchunk = hin_arr ? (j0 < m ? (int)hin_arr[j0] : 0) : 1;
		}
		const uint32_t c_new = bcast(tchunk, s & 63); const int h_new = bcast(hchunk, s & 63);
		const uint32_t c_up = shr1(c); const int h_up = shr1(hout);
		c = lane == 0 ? c_new : c_up;
		const int hin = lane == 0 ? h_new : h_up;
		const bool valid = act && s >= lane && s - lane < m;
		hout = 0;
		if (valid)
		{
			uint64_t Eq = c == 0 ? e0 : c == 1 ? e1 : c == 2 ? e2 : e3;
			const uint64_t hneg = hin < 0 ? 1ull : 0ull;
			const uint64_t Xv = Eq | Mv;
			Eq |= hneg;
			const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
			uint64_t Ph = Mv | ~(Xh | Pv);
			uint64_t Mh = Pv & Xh;
			const uint64_t ph_rows = Ph;
			if (owner)
			{
				st.sc += (uint32_t)((Ph >> lastbit) & 1) - (uint32_t)((Mh >> lastbit) & 1);
				if (shw && st.sc < st.best) { st.best = st.sc; st.end = (int32_t)(s - lane); }
			}
			hout = (int)(Ph >> 63) - (int)(Mh >> 63);
			Ph <<= 1; Mh <<= 1;
			Mh |= hneg; Ph |= hin > 0 ? 1ull : 0ull;
			Pv = Mh | ~(Xv | Ph);
			Mv = Ph & Xv;
			S += hout;
			if (hist) { const uint64_t idx = ((uint64_t)tile * (m + 64) + s) * hist->W + lane; hist->P[idx] = Pv; hist->H[idx] = ph_rows; }
			if (hout_arr && lane == W - 1) hout_arr[s - lane] = (int8_t)hout;
		}
		if (prog_out && (s & 63) == 63)
		{	// after step s the last block is through column s - (W - 1)
			__builtin_amdgcn_s_waitcnt(0);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			if (lane == 0) *prog_out = s + 2 > W ? s + 2 - W : 0u;
		}
	}
	if (lastcol && act)
	{	// column m of this block, bottom row upwards
		const uint32_t lo = b * 64; int32_t v = S;
		for (int r = 63; r >= 0; --r)
		{
			const uint32_t i = lo + (uint32_t)r + 1;
			if (i <= ne) lastcol[i] = v;
			v -= (int32_t)((Pv >> r) & 1); v += (int32_t)((Mv >> r) & 1);
		}
	}
	__builtin_amdgcn_s_waitcnt(0);          // the next tile reads what this one wrote
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	if (prog_out && lane == 0) *prog_out = m;
}

// Sweeps all columns.  q/t: byte sequences with steps +-1.  ne: the rows to compute (sat_rows: n, or a multiple of 64 below
// which every column is i - j).  hist: where to keep (vertical +1, horizontal +1) bit-vectors per cell of the ne rows, or
// null.  lastcol: ne + 1 values D[i][m], or null.  shw: track min over columns of D[n][j] (first minimum; end = -1 a
// candidate when n % 64 != 0).  score = D[n][m].
__device__ inline Sweep wave_sweep(WavePool& pool, const uint8_t* q, int qstep, uint32_t n, uint32_t ne, const uint8_t* t, int tstep, uint32_t m, bool shw, const Hist* hist, int32_t* lastcol)
{
	const uint32_t lane = lane_id();
	const bool sat = ne < n;
	Sweep out{ n, 0xffffffffu, (int32_t)m - 1 };
	if (sat)
	{	// row n is i - j in every column: D[n][j] = n - j falls strictly, the first minimum is the last column
		out.score = n - m; out.best = n - m; out.end = (int32_t)m - 1;
		if (!hist && !lastcol) return out;
	}
	const uint32_t nb = (ne + 63) / 64, tiles = (nb + 63) / 64;
	const uint64_t mk = pool.mark();
	int8_t* hb_a = (int8_t*)pool.alloc(m + 64ull); int8_t* hb_b = (int8_t*)pool.alloc(m + 64ull);
	if (pool.overflow) { pool.release(mk); return out; }
	if (!sat && shw && (n & 63)) { out.best = n; out.end = -1; }
	TileState st{ n, out.best, out.end };
	if (lastcol && lane == 0) lastcol[0] = (int32_t)m;
	for (uint32_t tile = 0; tile < tiles; ++tile)
	{
		sweep_tile(tile, q, qstep, n, ne, t, tstep, m, shw, sat, hist, lastcol, tile ? hb_a : nullptr, tile + 1 < tiles ? hb_b : nullptr, nullptr, nullptr, st);
		{ int8_t* x = hb_a; hb_a = hb_b; hb_b = x; }
	}
	if (!sat) { const uint32_t own_lane = (nb - 1) & 63; out.score = bcast(st.sc, own_lane); out.best = bcast(st.best, own_lane); out.end = bcast(st.end, own_lane); }
	pool.release(mk);
	return out;
}

struct Ops { uint8_t* p; uint64_t n; };                   // 0 match, 1 consume query, 2 consume target, 3 mismatch (forward order)

// walks the history of a sweep over q[0..n) x t[0..h.m) back from cell (n, j_start), j_start <= h.m; appends the ops
__device__ inline void wave_walk(WavePool& pool, const Hist& h, const uint8_t* q, uint32_t n, const uint8_t* t, uint32_t j_start, uint8_t* rev, Ops& out)
{
	const uint32_t lane = lane_id();
	const uint32_t m = h.m;
	uint32_t i = n, j = j_start; uint64_t k = 0;
	if (n > h.rows)
	{	// rows the sweep left out (sat_rows): vertical delta +1 in every column, the walk goes straight up through them
		const uint32_t run = n - h.rows;
		for (uint32_t x = lane; x < run; x += 64) rev[x] = 1;
		i = h.rows; k = run;
	}
	// window: lanes hold (P, H) of block wb for columns wj0 - lane (1-based column wj0 at lane 0, descending)
	uint32_t wb = 0xffffffffu, wj0 = 0; uint64_t wP = 0, wH = 0; uint32_t wt = 0;
	uint32_t wi0 = 0, wq = 0;                                                // lanes hold q[wi0 - 1 - lane]
	while (i > 0 && j > 0)
	{
		const uint32_t r = i - 1, b = r >> 6;
		if (b != wb || j > wj0 || wj0 - j >= 64)
		{
			wb = b; wj0 = j;
			const uint32_t jj = j > lane ? j - lane : 0;                      // this lane's column (1-based), 0 = none
			if (jj)
			{
				const uint32_t tile = b >> 6, bl = b & 63;
				const uint64_t idx = ((uint64_t)tile * (m + 64) + (jj - 1) + bl) * h.W + bl;
				wP = h.P[idx]; wH = h.H[idx]; wt = t[jj - 1];
			}
		}
		// The walk takes the same decisions as edlib's cell by cell (up if the vertical delta is +1, else left if the
		// horizontal delta is +1, else diagonal), but a whole RUN per iteration: the lanes look at the cells the run
		// would visit (same column going up, same row going left, the diagonal) and a ballot finds where it stops.
		const uint32_t src = wj0 - j, rb = r & 63;
		const int d = (int)lane - (int)src;                                  // this lane's column is j - d
		const bool col_ok = d >= 0 && lane < wj0;
		const uint32_t pr = (uint32_t)(wP >> rb) & 1, hr = (uint32_t)(wH >> rb) & 1;
		const uint32_t p0 = bcast(pr, src), h0 = bcast(hr, src);
		const uint64_t from_src = ~0ull << src;
		if (p0)
		{	// up while the vertical +1 bits of column j continue (inside this block)
			const uint64_t x = ~bcast(wP, src) << (63 - rb);
			uint32_t run = x ? (uint32_t)__builtin_clzll(x) : 64u;
			if (run > rb + 1) run = rb + 1;
			if (lane < run) rev[k + lane] = 1;
			i -= run; k += run;
		}
		else if (h0)
		{	// left while row r has no vertical +1 and a horizontal +1
			const uint64_t stop = ~__ballot(col_ok && !pr && hr) & from_src;
			const uint32_t run = (stop ? (uint32_t)__builtin_ctzll(stop) : 64u) - src;
			if (d >= 0 && (uint32_t)d < run) rev[k + (uint32_t)d] = 2;
			j -= run; k += run;
		}
		else
		{	// diagonal while neither bit is set at (r - d, j - d); the symbols decide match / mismatch
			const uint32_t bp = (rb - (uint32_t)d) & 63;
			const bool ok = col_ok && (uint32_t)d <= rb && !((wP >> bp) & 1) && !((wH >> bp) & 1);
			const uint64_t stop = ~__ballot(ok) & from_src;
			uint32_t run = (stop ? (uint32_t)__builtin_ctzll(stop) : 64u) - src;
			if (i > wi0 || wi0 - i >= 64) { wi0 = i; wq = i > lane ? q[i - 1 - lane] : 0u; }
			const uint32_t qoff = wi0 - i;
			if (run > 64 - qoff) run = 64 - qoff;
			const uint32_t qs = (uint32_t)__shfl((int)wq, (int)((qoff + (uint32_t)d) & 63));
			if (d >= 0 && (uint32_t)d < run) rev[k + (uint32_t)d] = qs == wt ? 0 : 3;
			i -= run; j -= run; k += run;
		}
	}
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	// out = [2 x j] [1 x i] reversed(rev)  in forward order: leading target consumption first?  No: the walk ended at (i, j)
	// with i == 0 or j == 0; the remaining prefix is consumed first in forward order.
	uint8_t* dst = out.p + out.n;
	const uint64_t pre = (uint64_t)i + j; const uint8_t pre_op = i ? 1 : 2;
	for (uint64_t x = lane; x < pre; x += 64) dst[x] = pre_op;
	for (uint64_t x = lane; x < k; x += 64) dst[pre + x] = rev[k - 1 - x];
	out.n += pre + k;
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}
// ---- 16-lane rows (align_rows.hpp: four gaps per wave, one per row) ----------------------------------------------------------------------
// DPP moves inside a row: lane l of a row receives lane l - 1's value (row_shr:1; the row's lane 0 keeps 0), or the value of the next lane
// round the row (row_ror:15).
__device__ inline uint32_t row_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); }
__device__ inline int row_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); }
__device__ inline uint32_t row_rol1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x12f, 0xf, 0xf, false); }     // row_ror:15

// edlib keeps the whole history when it fits 1 MiB (edlib.cpp:1176-1183), else it divides (Hirschberg)
__device__ inline bool wave_direct_fits(uint32_t n, uint32_t m)
{
	const long long blocks = (n + 63) / 64;
	return (2ll * 8 + 4) * blocks * m + 2ll * 4 * m < 1024 * 1024;
}
// ONE sweep that keeps the history, then the walk from (n, m) — or, shw, from (n, end + 1): the columns up to the end
// position are the same whether or not the sweep went on beyond it, so the reference's second pass over the truncated
// target (edlib.cpp:196-236) is not needed.  Appends the ops, returns the sweep's result.
__device__ inline Sweep wave_align_direct(WavePool& pool, const uint8_t* q, uint32_t n, const uint8_t* t, uint32_t m, bool shw, Ops& out)
{
	const uint64_t mk = pool.mark();
	const uint32_t ne = sat_rows(q, 1, n, t, 1, m);
	const uint32_t nb = (ne + 63) / 64;
	const uint64_t hw = hist_words(nb, m);
	Hist h{ (uint64_t*)pool.alloc(hw * 8), (uint64_t*)pool.alloc(hw * 8), nb < 64 ? nb : 64, m, ne };
	uint8_t* rev = (uint8_t*)pool.alloc((uint64_t)n + m + 64);
	Sweep sw{ n, 0xffffffffu, (int32_t)m - 1 };
	if (pool.overflow) { pool.release(mk); return sw; }
	sw = wave_sweep(pool, q, 1, n, ne, t, 1, m, shw, &h, nullptr);
	if (pool.overflow) { pool.release(mk); return sw; }
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	wave_walk(pool, h, q, n, t, shw ? (uint32_t)(sw.end + 1) : m, rev, out);
	pool.release(mk);
	return sw;
}
// traceback of q[0..n) x t[0..m) (forward byte sequences) on a fresh history; appends n..n+m ops
__device__ inline void wave_traceback(WavePool& pool, const uint8_t* q, uint32_t n, const uint8_t* t, uint32_t m, Ops& out) { wave_align_direct(pool, q, n, t, m, false, out); }

// Hirschberg's split row (edlib.cpp:1317-1356): the smallest i in 1..n-1 with left[i] + right[n - i] == best, else the empty
// prefix (0), else the whole query (n); -1: none.  left / right: the last columns of the two half sweeps, neL + 1 and neR + 1
// values; rows the sweeps left out (sat_rows) are closed form — left[i] = i - L, right[x] = x - R — and where both are, the sum
// is n - m for every i.  ls, rs: the two scores at the split.
__device__ inline int64_t hirschberg_split(const int32_t* left, uint32_t neL, uint32_t L, const int32_t* right, uint32_t neR, uint32_t R, uint32_t n, uint32_t best, uint32_t& ls, uint32_t& rs)
{
	const uint32_t lane = lane_id();
	auto Lv = [&](uint32_t i) -> uint32_t { return i <= neL ? (uint32_t)left[i] : i - L; };
	auto Rv = [&](uint32_t x) -> uint32_t { return x <= neR ? (uint32_t)right[x] : x - R; };
	int64_t found = -1;
	auto scan = [&](uint32_t a, uint32_t b) {                                        // i in [a, b], ascending
		for (uint32_t base = a; base <= b && found < 0; base += 256)
		{	// four 64-row steps per round trip (the eight loads are independent)
			bool hit[4];
#pragma unroll
			for (uint32_t u = 0; u < 4; ++u) { const uint32_t i = base + u * 64 + lane; hit[u] = i <= b && Lv(i) + Rv(n - i) == best; }
#pragma unroll
			for (uint32_t u = 0; u < 4; ++u)
			{
				const uint64_t bal = __ballot(hit[u]);
				if (bal && found < 0) { const uint32_t f = (uint32_t)__builtin_ctzll(bal); found = base + u * 64 + f; }
			}
		}
	};
	if (n >= 2)
	{
		const uint32_t z1 = neL < n - 1 ? neL : n - 1;                               // left explicit up to here
		scan(1, z1);
		const uint32_t z3 = (n > neR && n - neR > z1 + 1) ? n - neR : z1 + 1;        // right explicit from here
		if (found < 0 && z3 > z1 + 1 && n - (L + R) == best) found = z1 + 1;         // both closed form in (z1, z3)
		if (found < 0 && z3 <= n - 1) scan(z3, n - 1);
	}
	if (found >= 0) { ls = bcast_first(Lv((uint32_t)found)); rs = bcast_first(Rv(n - (uint32_t)found)); }
	if (found < 0 && L + bcast_first(Rv(n)) == best) { found = 0; ls = L; rs = bcast_first(Rv(n)); }
	if (found < 0 && bcast_first(Lv(n)) + R == best) { found = n; ls = bcast_first(Lv(n)); rs = R; }
	return found;
}

// obtainAlignment (edlib.cpp:1164-1215): optimal path of q (rows) against t (columns) given the optimal score
__device__ inline void wave_path(WavePool& pool, const uint8_t* q, uint32_t n, const uint8_t* t, uint32_t m, uint32_t best, Ops& out)
{
	struct Job { uint32_t qo, n, to, m, best; };
	const uint32_t lane = lane_id();
	const uint64_t mk0 = pool.mark();
	Job* stack = (Job*)pool.alloc(sizeof(Job) * 128);
	if (pool.overflow) return;
	uint32_t sp = 0;
	if (lane == 0) stack[0] = Job{ 0, n, 0, m, best };
	sp = 1;
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	while (sp && !pool.overflow)
	{
		--sp;
		Job jb;
		jb.qo = bcast_first(stack[sp].qo); jb.n = bcast_first(stack[sp].n); jb.to = bcast_first(stack[sp].to); jb.m = bcast_first(stack[sp].m); jb.best = bcast_first(stack[sp].best);
		if (jb.n == 0 || jb.m == 0)
		{
			const uint8_t op = jb.n == 0 ? 2 : 1;
			for (uint64_t x = lane; x < (uint64_t)jb.n + jb.m; x += 64) out.p[out.n + x] = op;
			out.n += (uint64_t)jb.n + jb.m;
			continue;
		}
		if (wave_direct_fits(jb.n, jb.m)) { wave_traceback(pool, q + jb.qo, jb.n, t + jb.to, jb.m, out); continue; }
		const uint32_t L = jb.m / 2, R = jb.m - L;
		const uint64_t mk = pool.mark();
		const uint32_t neL = sat_rows(q + jb.qo, 1, jb.n, t + jb.to, 1, L), neR = sat_rows(q + jb.qo + jb.n - 1, -1, jb.n, t + jb.to + jb.m - 1, -1, R);
		int32_t* left = (int32_t*)pool.alloc(((uint64_t)neL + 1) * 4);
		int32_t* right = (int32_t*)pool.alloc(((uint64_t)neR + 1) * 4);
		if (pool.overflow) break;
		wave_sweep(pool, q + jb.qo, 1, jb.n, neL, t + jb.to, 1, L, false, nullptr, left);
		wave_sweep(pool, q + jb.qo + jb.n - 1, -1, jb.n, neR, t + jb.to + jb.m - 1, -1, R, false, nullptr, right);
		if (pool.overflow) break;
		__builtin_amdgcn_s_waitcnt(0);
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		uint32_t ls = 0, rs = 0;
		const int64_t found = hirschberg_split(left, neL, L, right, neR, R, jb.n, jb.best, ls, rs);
		pool.release(mk);
		if (found < 0 || sp + 2 > 128) { pool.overflow = true; break; }
		if (lane == 0)
		{	// lower-right half first on the stack so that the upper-left half is emitted first
			stack[sp] = Job{ jb.qo + (uint32_t)found, jb.n - (uint32_t)found, jb.to + L, R, rs };
			stack[sp + 1] = Job{ jb.qo, (uint32_t)found, jb.to, L, ls };
		}
		sp += 2;
		__builtin_amdgcn_s_waitcnt(0);
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	}
	pool.release(mk0);
}

// refactor_edit_script (edit_script.h:416-446,591-671) by the whole wave.  Each of its two passes rewrites every maximal
// REGION — consecutive script symbols that are neither a break (pass 1: insertion / substitution, pass 2: deletion /
// substitution) nor step onto a different sequence symbol than their predecessor — as its matches first, then its
// other symbols (pass 1: 'D'; pass 2: the inserted letter, the same throughout a region).  A forward sweep gives every
// symbol its rank in the region and the matches up to it, a backward sweep the matches after it; 64 symbols per step,
// regions may span steps (carries).
__device__ inline bool wave_refactor_pass(WavePool& pool, char* es, uint32_t k, const uint8_t* seq, int pass)
{
	const uint32_t lane = lane_id();
	const uint64_t mk = pool.mark();
	const uint32_t n_chunks = (k + 63) / 64;
	uint32_t* rank = (uint32_t*)pool.alloc((uint64_t)n_chunks * 64 * 4); uint32_t* mi = (uint32_t*)pool.alloc((uint64_t)n_chunks * 64 * 4);
	uint8_t* oth = (uint8_t*)pool.alloc((uint64_t)n_chunks * 64);
	uint64_t* bits = (uint64_t*)pool.alloc((uint64_t)n_chunks * 16);              // per chunk: boundary mask, match mask
	if (pool.overflow) { pool.release(mk); return false; }
	const uint64_t le = lane == 63 ? ~0ull : ((2ull << lane) - 1);                  // lanes <= me
	// Both sweeps are chains of memory round trips when written naively (symbols of the step -> their positions -> the
	// sequence symbols there): on the slowest gaps of a launch — flanks of 10^5 symbols, which bound the launch — that was most
	// of the time.  So the loads run AHEAD of the carries: script symbols two steps ahead, sequence symbols one step ahead
	// (their positions need only the consumed-symbol counts of the steps before, not the region carries).
	auto is_ins = [](char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; };
	auto is_mis = [](char c) { return c == 'X' || c == 'Y' || c == 'Z'; };
	auto load_c = [&](uint32_t ch) -> char { const uint32_t x = ch * 64 + lane; return (ch < n_chunks && x < k) ? es[x] : ' '; };
	auto classify = [&](uint32_t ch, char c, bool& cons, bool& reg) {
		const bool valid = ch * 64 + lane < k && ch < n_chunks;
		const bool ins = is_ins(c), mis = is_mis(c), del = c == 'D';
		cons = valid && (pass == 1 ? !ins : !del); reg = valid && (pass == 1 ? !(ins || mis) : !(del || mis));
	};
	uint32_t pos_base = 0; bool c_reg = false; uint32_t c_sym = 0xff, c_start = 0, c_m = 0;
	char cA = load_c(0), cB = load_c(1);
	uint32_t symA; uint32_t pos_next;                                              // sequence symbols of the current step; consumed symbols before the next
	{
		bool cons, reg; classify(0, cA, cons, reg);
		const uint64_t cmask = __ballot(cons);
		const uint32_t pos = (uint32_t)__popcll(cmask & (le >> 1));
		symA = reg ? seq[pos] : 0xffu;
		pos_next = (uint32_t)__popcll(cmask);
	}
	for (uint32_t ch = 0; ch < n_chunks; ++ch)
	{
		const uint32_t x = ch * 64 + lane;
		const char c = cA;
		const char cC = load_c(ch + 2);
		// the next step's sequence symbols
		uint32_t symB; uint32_t pos_after;
		{
			bool consB, regB; classify(ch + 1, cB, consB, regB);
			const uint64_t cmaskB = __ballot(consB);
			const uint32_t posB = pos_next + (uint32_t)__popcll(cmaskB & (le >> 1));
			symB = regB ? seq[posB] : 0xffu;
			pos_after = pos_next + (uint32_t)__popcll(cmaskB);
		}
		bool cons, reg; classify(ch, c, cons, reg);
		const uint32_t sym = reg ? symA : 0xffu;
		uint32_t p_sym = shr1(sym); bool p_reg = shr1((int)reg) != 0;
		if (lane == 0) { p_sym = c_sym; p_reg = c_reg; }
		const bool head = reg && (!p_reg || p_sym != sym);
		const uint64_t H = __ballot(head), R = __ballot(reg), Mm = __ballot(reg && c == 'M');
		uint32_t start = 0, m_incl = 0;
		if (reg)
		{
			const uint64_t below = H & le;
			if (below) { const uint32_t h = 63 - (uint32_t)__builtin_clzll(below); start = ch * 64 + h; m_incl = (uint32_t)__popcll(Mm & le & ~((1ull << h) - 1)); }
			else { start = c_start; m_incl = c_m + (uint32_t)__popcll(Mm & le); }
			rank[x] = x - start; mi[x] = m_incl; oth[x] = pass == 1 ? (uint8_t)'D' : (uint8_t)(sym == 0 ? 'A' : sym == 1 ? 'C' : sym == 2 ? 'G' : 'T');
		}
		if (lane == 0) { bits[2 * ch] = H | ~R; bits[2 * ch + 1] = Mm; }
		c_reg = bcast((int)reg, 63) != 0; c_sym = bcast(sym, 63); c_start = bcast(start, 63); c_m = bcast(m_incl, 63);
		(void)pos_base;
		cA = cB; cB = cC; symA = symB; pos_next = pos_after;
	}
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	uint32_t c_ma = 0;                                                               // matches from the start of the later chunks up to their first boundary
	// backward: everything a step needs is loaded one step ahead
	struct Pre { uint64_t Bd, Mm; char c; uint32_t rk, mi_; uint8_t ot; };
	auto load_back = [&](uint32_t ch) -> Pre {
		Pre q{ 0, 0, ' ', 0, 0, 0 };
		if (ch >= n_chunks) return q;
		const uint32_t x = ch * 64 + lane;
		q.Bd = bits[2 * ch]; q.Mm = bits[2 * ch + 1];
		if (x < k) { q.c = es[x]; q.rk = rank[x]; q.mi_ = mi[x]; q.ot = oth[x]; }
		return q;
	};
	Pre cur = load_back(n_chunks - 1);
	for (uint32_t ch = n_chunks; ch-- > 0;)
	{
		const uint32_t x = ch * 64 + lane;
		const Pre nxt = ch ? load_back(ch - 1) : Pre{ 0, 0, ' ', 0, 0, 0 };
		const uint64_t Bd = cur.Bd, Mm = cur.Mm;                                      // boundary = region head or not a region symbol
		const uint64_t above = Bd & ~le;
		uint32_t ma;
		if (above) { const uint32_t e = (uint32_t)__builtin_ctzll(above); ma = (uint32_t)__popcll(Mm & ~le & ((1ull << e) - 1)); }
		else ma = (uint32_t)__popcll(Mm & ~le) + c_ma;
		if (x < k)
		{
			const char c = cur.c;
			const bool ins = is_ins(c), mis = is_mis(c), del = c == 'D';
			const bool rg = pass == 1 ? !(ins || mis) : !(del || mis);
			if (rg) es[x] = cur.rk < cur.mi_ + ma ? 'M' : (char)cur.ot;
		}
		if (Bd) { const uint32_t e0 = (uint32_t)__builtin_ctzll(Bd); c_ma = (uint32_t)__popcll(Mm & ((1ull << e0) - 1)); }
		else c_ma += (uint32_t)__popcll(Mm);
		cur = nxt;
	}
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	pool.release(mk);
	return true;
}
__device__ inline bool wave_refactor(WavePool& pool, char* es, uint32_t k, const uint8_t* ref, const uint8_t* enc)
{
	return wave_refactor_pass(pool, es, k, ref, 1) && wave_refactor_pass(pool, es, k, enc, 2);
}

} // namespace wv
