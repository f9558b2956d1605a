// align_team.hpp — one WORK-GROUP ("team" of TW waves) aligns one giant gap (device only).  Same observable behaviour as
// align_wave.hpp (edlib as the reference calls it: edit_script.h:272-413; edlib.cpp:141-296,547-700,945-1400).
//
// Why: a launch of the wave-per-gap kernel lasts as long as its slowest gap, and the slowest gaps are the few whose rows AND
// columns run into the ten thousands (a 15 k x 22 k flank: 4 tiles x 22 k steps for the score, then ~7 Hirschberg levels whose
// sub-problems one wave works through one after the other — 60-70 ms with the machine idle; 10^5 x 10^5 of unrelated
// sequence: 0.8 s).  Two kinds of parallelism are free there and one wave cannot use them:
//   * the TILES of one sweep (64 row blocks each) only depend on the tile above, column by column: a team runs them on its waves
//     as a pipeline, each tile ~128 columns behind the one above (sweep_tile's progress words);
//   * the SUB-PROBLEMS of one Hirschberg level are independent: the team takes the level's jobs side by side — the left and the
//     right half sweep of a big job on the two halves of the team, small jobs one per wave.
// Leaves write their operations at position (query offset + target offset) of a sparse buffer: sub-problems tile the
// alignment in that order and a sub-problem of n x m symbols yields at most n + m operations, so the ranges are disjoint; one
// compaction at the end concatenates them.  No ordering bookkeeping between the levels.
#pragma once
#include "align_wave.hpp"

namespace wt {
using wv::WavePool; using wv::Sweep; using wv::Hist; using wv::Ops; using wv::lane_id; using wv::bcast_first;

constexpr uint32_t TW = 8;                       // waves per team
constexpr uint32_t MAX_TILES = 64;               // progress words per sweep: rows <= 64 * 4096
constexpr uint32_t BIG_ROWS = 4096;              // a job with more rows than one tile is swept by the team, not by one wave
constexpr uint32_t JOB_CAP = 1u << 15;

struct Job { uint32_t qo, n, to, m, best; };

struct TeamLds {
	uint32_t prog[2 * MAX_TILES];                // [0, 64): whole team or left half; [64, 128): right half
	uint32_t res[4];                             // score, best, end of a team sweep
	uint32_t n_cur, n_next, n_big, fail, gap;
	uint64_t total;
};

struct Team {
	uint32_t w;                                  // this wave's index in the team
	WavePool shared;                             // the SAME pool object in every wave: all waves make the same allocations (uniform control flow)
	WavePool own;                                // this wave's private scratch
	TeamLds* lds;
	__device__ inline void barrier() { __syncthreads(); }
};

// a sweep by waves [w0, w0 + cnt) of the team; tiles go round the waves.  hb: tiles x (m + 64) bytes of hand-over arrays, prog:
// one progress word per tile (zeroed, barrier, before the call).  Result: in T.lds->res once the waves are through (barrier).
__device__ inline void team_sweep(Team& T, uint32_t w0, uint32_t cnt, const uint8_t* q, int qstep, uint32_t n, uint32_t ne, const uint8_t* t, int tstep, uint32_t m, bool shw,
                                  const Hist* hist, int32_t* lastcol, int8_t* hb, volatile uint32_t* prog, bool want_result)
{
	if (T.w < w0 || T.w >= w0 + cnt) return;
	const uint32_t lane = lane_id();
	const bool sat = ne < n;
	const uint32_t nb = (ne + 63) / 64, tiles = (nb + 63) / 64;
	uint32_t best0 = 0xffffffffu; int32_t end0 = (int32_t)m - 1;
	if (!sat && shw && (n & 63)) { best0 = n; end0 = -1; }
	wv::TileState st{ n, best0, end0 };
	if (lastcol && lane == 0 && T.w == w0) lastcol[0] = (int32_t)m;
	for (uint32_t tile = T.w - w0; tile < tiles; tile += cnt)
	{
		wv::sweep_tile(tile, q, qstep, n, ne, t, tstep, m, shw, sat, hist, lastcol, tile ? hb + (uint64_t)(tile - 1) * (m + 64) : nullptr, tile + 1 < tiles ? hb + (uint64_t)tile * (m + 64) : nullptr,
		               tile ? prog + (tile - 1) : nullptr, tile + 1 < tiles ? prog + tile : nullptr, st);
		if (want_result && tile == tiles - 1)
		{
			uint32_t score = n - m, best = n - m; int32_t end = (int32_t)m - 1;
			if (!sat) { const uint32_t own_lane = (nb - 1) & 63; score = wv::bcast(st.sc, own_lane); best = wv::bcast(st.best, own_lane); end = wv::bcast(st.end, own_lane); }
			if (lane == 0) { T.lds->res[0] = score; T.lds->res[1] = best; T.lds->res[2] = (uint32_t)end; }
		}
	}
}

// obtainAlignment (edlib.cpp:1164-1215) of q (rows) against t (columns) by the team, given the optimal score: the operations
// (0 match, 1 consume query, 2 consume target, 3 mismatch) end up in out.p[0 .. out.n).  sparse: n + m + 64 bytes.
__device__ inline bool team_path(Team& T, const uint8_t* q, uint32_t n, const uint8_t* t, uint32_t m, uint32_t best, uint8_t* sparse, Ops& out)
{
	const uint32_t lane = lane_id();
	TeamLds& L = *T.lds;
	const uint64_t mk0 = T.shared.mark();
	Job* cur = (Job*)T.shared.alloc(sizeof(Job) * JOB_CAP); Job* nxt = (Job*)T.shared.alloc(sizeof(Job) * JOB_CAP);
	uint32_t* cbase = (uint32_t*)T.shared.alloc(4ull * JOB_CAP);             // first child slot of a job that is split, ~0 otherwise
	uint32_t* big = (uint32_t*)T.shared.alloc(4ull * JOB_CAP);               // the split jobs of more than one tile
	const uint64_t span = (uint64_t)n + m;
	const uint32_t n_chunks = (uint32_t)((span + 63) / 64);
	uint32_t* ccnt = (uint32_t*)T.shared.alloc(4ull * (n_chunks + 64));
	if (T.shared.overflow) return false;
	for (uint64_t x = (uint64_t)T.w * 64 + lane; x < span; x += 64 * TW) sparse[x] = 0xff;
	if (T.w == 0 && lane == 0) { cur[0] = Job{ 0, n, 0, m, best }; L.n_cur = 1; L.fail = 0; }
	T.barrier();
	// (shared words are read right after a barrier and written only after the next one)
	for (;;)
	{
		const uint32_t n_cur = L.n_cur; uint32_t fail = L.fail;
		T.barrier();
		if (!n_cur || fail) break;
		// which jobs are split (two children in the next list), which are leaves
		if (T.w == 0)
		{
			uint32_t base = 0, nbig = 0;
			for (uint32_t i0 = 0; i0 < n_cur; i0 += 64)
			{
				const uint32_t i = i0 + lane;
				bool split = false, isbig = false;
				if (i < n_cur) { const Job jb = cur[i]; split = jb.n && jb.m && !wv::wave_direct_fits(jb.n, jb.m); isbig = split && jb.n > BIG_ROWS; }
				const uint64_t bal = __ballot(split), bbal = __ballot(isbig);
				const uint64_t below = (1ull << lane) - 1;
				if (i < n_cur) cbase[i] = split ? base + 2 * (uint32_t)__popcll(bal & below) : 0xffffffffu;
				if (isbig && nbig + (uint32_t)__popcll(bbal & below) < JOB_CAP) big[nbig + (uint32_t)__popcll(bbal & below)] = i;
				base += 2 * (uint32_t)__popcll(bal); nbig += (uint32_t)__popcll(bbal);
			}
			if (lane == 0) { L.n_next = base; L.n_big = nbig; if (base > JOB_CAP) L.fail = 1; }
		}
		T.barrier();
		fail = L.fail; const uint32_t n_big = L.n_big;
		T.barrier();
		if (fail) break;
		// big jobs, one after the other: left half sweep on waves [0, TW/2), right half sweep on the others, as tile pipelines
		for (uint32_t bi = 0; bi < n_big; ++bi)
		{
			const uint32_t i = big[bi];
			const Job jb = cur[i];
			const uint32_t Lc = jb.m / 2, Rc = jb.m - Lc;
			const uint8_t* ql = q + jb.qo; const uint8_t* tl = t + jb.to;
			const uint8_t* qr = q + jb.qo + jb.n - 1; const uint8_t* tr = t + jb.to + jb.m - 1;
			const uint32_t neL = wv::sat_rows(ql, 1, jb.n, tl, 1, Lc), neR = wv::sat_rows(qr, -1, jb.n, tr, -1, Rc);
			const uint32_t tilesL = ((neL + 63) / 64 + 63) / 64, tilesR = ((neR + 63) / 64 + 63) / 64;
			const uint64_t mk = T.shared.mark();
			int32_t* left = (int32_t*)T.shared.alloc(((uint64_t)neL + 1) * 4); int32_t* right = (int32_t*)T.shared.alloc(((uint64_t)neR + 1) * 4);
			int8_t* hbL = (int8_t*)T.shared.alloc((uint64_t)tilesL * (Lc + 64)); int8_t* hbR = (int8_t*)T.shared.alloc((uint64_t)tilesR * (Rc + 64));
			if (T.shared.overflow || tilesL > MAX_TILES || tilesR > MAX_TILES) { fail = 4; break; }       // (the same in every wave)
			for (uint32_t x = T.w * 64 + lane; x < 2 * MAX_TILES; x += 64 * TW) L.prog[x] = 0;
			T.barrier();
			team_sweep(T, 0, TW / 2, ql, 1, jb.n, neL, tl, 1, Lc, false, nullptr, left, hbL, L.prog, false);
			team_sweep(T, TW / 2, TW / 2, qr, -1, jb.n, neR, tr, -1, Rc, false, nullptr, right, hbR, L.prog + MAX_TILES, false);
			T.barrier();
			if (T.w == 0)
			{
				uint32_t ls = 0, rs = 0;
				const int64_t found = wv::hirschberg_split(left, neL, Lc, right, neR, Rc, jb.n, jb.best, ls, rs);
				if (found < 0) { if (lane == 0) L.fail = 2; }
				else if (lane == 0)
				{
					nxt[cbase[i]] = Job{ jb.qo, (uint32_t)found, jb.to, Lc, ls };
					nxt[cbase[i] + 1] = Job{ jb.qo + (uint32_t)found, jb.n - (uint32_t)found, jb.to + Lc, Rc, rs };
				}
			}
			T.shared.release(mk);
			T.barrier();
			fail = L.fail;
			if (fail) break;
		}
		if (fail) { T.barrier(); if (T.w == 0 && lane == 0) L.fail = fail; T.barrier(); break; }
		// everything else, one job per wave at a time: leaves (direct traceback / trivial), small splits
		for (uint32_t i = T.w; i < n_cur; i += TW)
		{
			const Job jb = cur[i];
			uint8_t* dst = sparse + jb.qo + jb.to;
			if (jb.n == 0 || jb.m == 0)
			{
				const uint8_t op = jb.n == 0 ? 2 : 1;
				for (uint64_t x = lane; x < (uint64_t)jb.n + jb.m; x += 64) dst[x] = op;
				continue;
			}
			if (cbase[i] == 0xffffffffu)
			{
				T.own.top = 0; T.own.overflow = false;
				Ops o{ dst, 0 };
				wv::wave_traceback(T.own, q + jb.qo, jb.n, t + jb.to, jb.m, o);
				if (T.own.overflow) { if (lane == 0) L.fail = 3; }
				continue;
			}
			if (jb.n > BIG_ROWS) continue;                                       // (done above)
			T.own.top = 0; T.own.overflow = false;
			const uint32_t Lc = jb.m / 2, Rc = jb.m - Lc;
			const uint8_t* ql = q + jb.qo; const uint8_t* tl = t + jb.to;
			const uint8_t* qr = q + jb.qo + jb.n - 1; const uint8_t* tr = t + jb.to + jb.m - 1;
			const uint32_t neL = wv::sat_rows(ql, 1, jb.n, tl, 1, Lc), neR = wv::sat_rows(qr, -1, jb.n, tr, -1, Rc);
			int32_t* left = (int32_t*)T.own.alloc(((uint64_t)neL + 1) * 4); int32_t* right = (int32_t*)T.own.alloc(((uint64_t)neR + 1) * 4);
			if (T.own.overflow) { if (lane == 0) L.fail = 3; continue; }
			wv::wave_sweep(T.own, ql, 1, jb.n, neL, tl, 1, Lc, false, nullptr, left);
			wv::wave_sweep(T.own, qr, -1, jb.n, neR, tr, -1, Rc, false, nullptr, right);
			if (T.own.overflow) { if (lane == 0) L.fail = 3; continue; }
			__builtin_amdgcn_s_waitcnt(0);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			uint32_t ls = 0, rs = 0;
			const int64_t found = wv::hirschberg_split(left, neL, Lc, right, neR, Rc, jb.n, jb.best, ls, rs);
			if (found < 0) { if (lane == 0) L.fail = 2; continue; }
			if (lane == 0)
			{
				nxt[cbase[i]] = Job{ jb.qo, (uint32_t)found, jb.to, Lc, ls };
				nxt[cbase[i] + 1] = Job{ jb.qo + (uint32_t)found, jb.n - (uint32_t)found, jb.to + Lc, Rc, rs };
			}
		}
		T.barrier();
		{ Job* x = cur; cur = nxt; nxt = x; }
		if (T.w == 0 && lane == 0) L.n_cur = L.n_next;
		T.barrier();
	}
	T.barrier();
	const uint32_t fail = L.fail;
	T.barrier();
	if (!fail)
	{	// compaction of the sparse buffer: per 64-byte chunk the number of operations, their prefix sums, the copy
		for (uint32_t c = T.w; c < n_chunks; c += TW)
		{
			const uint64_t x = (uint64_t)c * 64 + lane;
			const uint64_t bal = __ballot(x < span && sparse[x] != 0xff);
			if (lane == 0) ccnt[c] = (uint32_t)__popcll(bal);
		}
		T.barrier();
		if (T.w == 0)
		{
			uint64_t carry = 0;
			for (uint32_t c0 = 0; c0 < n_chunks; c0 += 64)
			{
				const uint32_t c = c0 + lane;
				const uint32_t v = c < n_chunks ? ccnt[c] : 0u;
				uint32_t incl = v;
				for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o); if ((int)lane >= o) incl += u; }
				if (c < n_chunks) ccnt[c] = (uint32_t)(carry + incl - v);
				carry += wv::bcast(incl, 63u);
			}
			if (lane == 0) L.total = carry;
		}
		T.barrier();
		for (uint32_t c = T.w; c < n_chunks; c += TW)
		{
			const uint64_t x = (uint64_t)c * 64 + lane;
			const uint8_t v = x < span ? sparse[x] : (uint8_t)0xff;
			const uint64_t bal = __ballot(v != 0xff);
			if (v != 0xff) out.p[out.n + ccnt[c] + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = v;
		}
		T.barrier();
		out.n += L.total;
	}
	T.shared.release(mk0);
	T.barrier();
	return fail == 0;
}

} // namespace wt
