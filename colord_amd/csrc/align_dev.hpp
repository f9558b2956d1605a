// align_dev.hpp — device-side gap alignment with the observable behaviour of the reference's aligner
// (edlib as called from src/colord/edit_script.h:272-413; libs/edlib/edlib.cpp:141-296,547-700,945-1400):
//   * Myers' bit-vector recurrence (one 64-row block per word) for the scores — the same recurrence edlib uses,
//     here over all blocks of a column (no Ukkonen band: cells outside edlib's band never qualify in the
//     traceback equalities, so the paths are identical);
//   * NW distance D[n][m]; SHW distance min over end positions, first minimum, with end position -1 a candidate
//     when the query length is not a multiple of 64 (edlib.cpp:666-681);
//   * traceback preferring up (consume a query symbol), then left (consume a target symbol), then the diagonal
//     (edlib.cpp:1021-1147) on stored per-column P/M/score words when they fit edlib's 1 MiB budget;
//   * otherwise Hirschberg on the target's middle column with edlib's choice of the split row
//     (edlib.cpp:1230-1400), iteratively with an explicit stack.
// One LANE runs one alignment; all working memory comes from the lane's bump pool in HBM.
#pragma once
#include "common.hpp"

struct LanePool {
	uint8_t* base; uint64_t cap, top; bool overflow; uint32_t why;
	CL_DEV inline void* alloc(uint64_t bytes)
	{
		bytes = (bytes + 15) & ~15ull;
		if (top + bytes > cap) { overflow = true; why |= 1u; return base; }      // keeps running on garbage; the read is redone with a larger pool
		void* p = base + top; top += bytes; return p;
	}
	CL_DEV inline uint64_t mark() const { return top; }
	CL_DEV inline void release(uint64_t m) { top = m; }
};

// ---- Myers block step (Myers 1999 / Hyyro): vertical deltas Pv/Mv of the previous column -> this column --------
CL_DEV inline int myers_block(uint64_t Pv, uint64_t Mv, uint64_t Eq, int hin, uint64_t& PvOut, uint64_t& MvOut)
{
	const uint64_t hneg = hin < 0 ? 1ull : 0ull;
	const uint64_t Xv = Eq | Mv;
	Eq |= hneg;
	const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
	uint64_t Ph = Mv | ~(Xh | Pv);
	uint64_t Mh = Pv & Xh;
	int hout = (int)(Ph >> 63) - (int)(Mh >> 63);
	Ph <<= 1; Mh <<= 1;
	Mh |= hneg; Ph |= hin > 0 ? 1ull : 0ull;
	PvOut = Mh | ~(Xv | Ph);
	MvOut = Ph & Xv;
	return hout;
}

struct Seq { const uint8_t* p; int32_t step; };                       // element i = p[i * step] (step = -1: reversed view)
CL_DEV inline uint8_t seq_at(const Seq& s, uint32_t i) { return s.p[(int64_t)i * s.step]; }

// match masks of the query: peq[sym * nb + b]
CL_DEV inline uint64_t* build_peq(LanePool& pool, const Seq& q, uint32_t n, uint32_t nb)
{
	uint64_t* peq = (uint64_t*)pool.alloc((uint64_t)4 * nb * 8);
	if (pool.overflow) return peq;
	for (uint32_t i = 0; i < 4 * nb; ++i) peq[i] = 0;
	for (uint32_t i = 0; i < n; ++i) peq[(uint32_t)(seq_at(q, i) & 3) * nb + (i >> 6)] |= 1ull << (i & 63);
	return peq;
}

// Column state of the global (NW) matrix: per block vertical deltas and the score at the block's last row.
struct ColState { uint64_t* P; uint64_t* M; int32_t* S; };

CL_DEV inline void col_init(ColState& c, uint32_t nb)
{
	for (uint32_t b = 0; b < nb; ++b) { c.P[b] = ~0ull; c.M[b] = 0; c.S[b] = (int32_t)((b + 1) * 64); }   // D[i][0] = i
}
CL_DEV inline void col_step(ColState& c, const uint64_t* peq_sym, uint32_t nb)
{
	int h = 1;                                                    // D[0][j] - D[0][j-1] = +1
	for (uint32_t b = 0; b < nb; ++b) { h = myers_block(c.P[b], c.M[b], peq_sym[b], h, c.P[b], c.M[b]); c.S[b] += h; }
}
// D[i][j] for row i (1-based, 1..n) given the column's words; i = 0 -> boundary j
CL_DEV inline int32_t col_value(const uint64_t* P, const uint64_t* M, const int32_t* S, uint32_t i, uint32_t j)
{
	if (i == 0) return (int32_t)j;
	const uint32_t r = i - 1, b = r >> 6, l = r & 63;
	const uint64_t above = l == 63 ? 0ull : (~0ull << (l + 1));            // rows of the block below row r (bits l+1..63)
	return S[b] - (int32_t)__builtin_popcountll(P[b] & above) + (int32_t)__builtin_popcountll(M[b] & above);
}

// D[n][m] only
CL_DEV inline uint32_t nw_distance(LanePool& pool, const Seq& q, uint32_t n, const Seq& t, uint32_t m)
{
	const uint64_t mk = pool.mark();
	const uint32_t nb = (n + 63) / 64;
	uint64_t* peq = build_peq(pool, q, n, nb);
	ColState c{ (uint64_t*)pool.alloc(nb * 8ull), (uint64_t*)pool.alloc(nb * 8ull), (int32_t*)pool.alloc(nb * 4ull) };
	if (pool.overflow) { pool.release(mk); return 0; }
	col_init(c, nb);
	for (uint32_t j = 0; j < m; ++j) col_step(c, peq + (uint32_t)(seq_at(t, j) & 3) * nb, nb);
	const uint32_t d = (uint32_t)col_value(c.P, c.M, c.S, n, m);
	pool.release(mk);
	return d;
}
// last column of the NW matrix: out[i] = D[i][m], i = 0..n
CL_DEV inline void nw_last_column(LanePool& pool, const Seq& q, uint32_t n, const Seq& t, uint32_t m, int32_t* out)
{
	const uint64_t mk = pool.mark();
	const uint32_t nb = (n + 63) / 64;
	uint64_t* peq = build_peq(pool, q, n, nb);
	ColState c{ (uint64_t*)pool.alloc(nb * 8ull), (uint64_t*)pool.alloc(nb * 8ull), (int32_t*)pool.alloc(nb * 4ull) };
	if (pool.overflow) { pool.release(mk); return; }
	col_init(c, nb);
	for (uint32_t j = 0; j < m; ++j) col_step(c, peq + (uint32_t)(seq_at(t, j) & 3) * nb, nb);
	int32_t v = (int32_t)m; out[0] = v;
	for (uint32_t i = 1; i <= n; ++i)
	{
		const uint32_t r = i - 1, b = r >> 6; const uint64_t bit = 1ull << (r & 63);
		v += (c.P[b] & bit) ? 1 : 0; v -= (c.M[b] & bit) ? 1 : 0;
		out[i] = v;
	}
	pool.release(mk);
}
// SHW: best = min_j D[n][j], end = first such j-1, with end = -1 (score n) considered first when n % 64 != 0
CL_DEV inline void shw_distance(LanePool& pool, const Seq& q, uint32_t n, const Seq& t, uint32_t m, uint32_t* best_out, int64_t* end_out)
{
	const uint64_t mk = pool.mark();
	const uint32_t nb = (n + 63) / 64;
	uint64_t* peq = build_peq(pool, q, n, nb);
	ColState c{ (uint64_t*)pool.alloc(nb * 8ull), (uint64_t*)pool.alloc(nb * 8ull), (int32_t*)pool.alloc(nb * 4ull) };
	if (pool.overflow) { pool.release(mk); *best_out = 0; *end_out = -1; return; }
	col_init(c, nb);
	uint32_t best = 0xffffffffu; int64_t end = 0;
	if (n % 64 != 0) { best = n; end = -1; }
	for (uint32_t j = 0; j < m; ++j)
	{
		col_step(c, peq + (uint32_t)(seq_at(t, j) & 3) * nb, nb);
		const uint32_t v = (uint32_t)col_value(c.P, c.M, c.S, n, j + 1);
		if (v < best) { best = v; end = j; }
	}
	pool.release(mk);
	*best_out = best; *end_out = end;
}

struct OpsOut { uint8_t* p; uint64_t n; };                               // 0 match, 1 consume query, 2 consume target, 3 mismatch

// traceback on stored columns (forward sequences only)
CL_DEV inline void nw_traceback(LanePool& pool, const uint8_t* q, uint32_t n, const uint8_t* t, uint32_t m, OpsOut& out)
{
	const uint64_t mk = pool.mark();
	const uint32_t nb = (n + 63) / 64;
	Seq qs{ q, 1 };
	uint64_t* peq = build_peq(pool, qs, n, nb);
	uint64_t* P = (uint64_t*)pool.alloc((uint64_t)m * nb * 8), * M = (uint64_t*)pool.alloc((uint64_t)m * nb * 8);
	int32_t* S = (int32_t*)pool.alloc((uint64_t)m * nb * 4);
	ColState c{ (uint64_t*)pool.alloc(nb * 8ull), (uint64_t*)pool.alloc(nb * 8ull), (int32_t*)pool.alloc(nb * 4ull) };
	uint8_t* rev = (uint8_t*)pool.alloc((uint64_t)n + m + 16);
	if (pool.overflow) { pool.release(mk); return; }
	col_init(c, nb);
	for (uint32_t j = 0; j < m; ++j)
	{
		col_step(c, peq + (uint32_t)(t[j] & 3) * nb, nb);
		for (uint32_t b = 0; b < nb; ++b) { P[(uint64_t)j * nb + b] = c.P[b]; M[(uint64_t)j * nb + b] = c.M[b]; S[(uint64_t)j * nb + b] = c.S[b]; }
	}
	auto D = [&](uint32_t i, uint32_t j) -> int32_t {
		if (j == 0) return (int32_t)i;
		return col_value(P + (uint64_t)(j - 1) * nb, M + (uint64_t)(j - 1) * nb, S + (uint64_t)(j - 1) * nb, i, j);
	};
	uint64_t k = 0;
	uint32_t i = n, j = m;
	int32_t cur = D(n, m);
	while (i > 0 && j > 0)
	{
		const int32_t up = D(i - 1, j);
		if (up + 1 == cur) { rev[k++] = 1; --i; cur = up; continue; }
		const int32_t left = D(i, j - 1);
		if (left + 1 == cur) { rev[k++] = 2; --j; cur = left; continue; }
		const int32_t dg = D(i - 1, j - 1);
		rev[k++] = dg == cur ? 0 : 3; --i; --j; cur = dg;
	}
	while (i > 0) { rev[k++] = 1; --i; }
	while (j > 0) { rev[k++] = 2; --j; }
	while (k > 0) out.p[out.n++] = rev[--k];
	pool.release(mk);
}

// obtainAlignment: optimal path of q (rows) against t (columns) given the optimal score
CL_DEV inline void nw_path(LanePool& pool, const uint8_t* q, uint32_t n, const uint8_t* t, uint32_t m, uint32_t best, OpsOut& out)
{
	struct Job { uint32_t qo, n, to, m, best; };
	const uint64_t mk0 = pool.mark();
	Job* stack = (Job*)pool.alloc(sizeof(Job) * 96);
	if (pool.overflow) return;
	uint32_t sp = 0;
	stack[sp++] = Job{ 0, n, 0, m, best };
	while (sp && !pool.overflow)
	{
		const Job jb = stack[--sp];
		if (jb.n == 0 || jb.m == 0) { for (uint32_t i = 0; i < jb.n + jb.m; ++i) out.p[out.n++] = jb.n == 0 ? 2 : 1; continue; }
		const long long blocks = (jb.n + 63) / 64;
		const long long sz = (2ll * 8 + 4) * blocks * jb.m + 2ll * 4 * jb.m;
		if (sz < 1024 * 1024) { nw_traceback(pool, q + jb.qo, jb.n, t + jb.to, jb.m, out); continue; }
		const uint32_t L = jb.m / 2, R = jb.m - L;
		const uint64_t mk = pool.mark();
		int32_t* left = (int32_t*)pool.alloc(((uint64_t)jb.n + 1) * 4);
		int32_t* right = (int32_t*)pool.alloc(((uint64_t)jb.n + 1) * 4);
		if (pool.overflow) break;
		nw_last_column(pool, Seq{ q + jb.qo, 1 }, jb.n, Seq{ t + jb.to, 1 }, L, left);
		nw_last_column(pool, Seq{ q + jb.qo + jb.n - 1, -1 }, jb.n, Seq{ t + jb.to + jb.m - 1, -1 }, R, right);
		int64_t found = -1; uint32_t ls = 0, rs = 0;
		for (uint32_t i = 1; i + 1 <= jb.n; ++i)
			if ((uint32_t)(left[i] + right[jb.n - i]) == jb.best) { found = i; ls = (uint32_t)left[i]; rs = (uint32_t)right[jb.n - i]; break; }
		if (found < 0 && L + (uint32_t)right[jb.n] == jb.best) { found = 0; ls = L; rs = (uint32_t)right[jb.n]; }
		if (found < 0 && (uint32_t)left[jb.n] + R == jb.best) { found = jb.n; ls = (uint32_t)left[jb.n]; rs = R; }
		pool.release(mk);
		if (found < 0) { pool.overflow = true; pool.why |= 2u; found = 0; ls = L; rs = jb.best > L ? jb.best - L : 0; }      // cannot happen with a correct optimum
		if (sp + 2 > 96) { pool.overflow = true; pool.why |= 4u; continue; }
		// lower-right half first on the stack so that the upper-left half is emitted first
		stack[sp++] = Job{ jb.qo + (uint32_t)found, jb.n - (uint32_t)found, jb.to + L, R, rs };
		stack[sp++] = Job{ jb.qo, (uint32_t)found, jb.to, L, ls };
	}
	pool.release(mk0);
}
