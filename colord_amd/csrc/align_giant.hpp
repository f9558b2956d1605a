// align_giant.hpp — the giant gaps of a level (class 7: more rows than one tile of 64 blocks, >= 2^19 block-columns, not saturating)
// aligned by MANY waves each, on as many CUs as are free (device only).  Same observable behaviour as align_wave.hpp (edlib as the
// reference calls it: edit_script.h:272-413; edlib.cpp:141-296,547-700,945-1400).
//
// Why: a 10^5 x 10^5 gap of unrelated sequence is 3 x 10^10 cell updates (score sweep + Hirschberg's 2 x area); one work-group (round
// 3's k_align_team) does ~1.6 x 10^11 per second, so such a gap held its level for 200 ms while 250 CUs had nothing of this lane to
// do.  The work has two free dimensions and a work-group can use neither fully:
//   * the TILES of one sweep (64 row blocks = 4096 rows each) depend only on the tile above, column by column: they run as a
//     pipeline of waves ANYWHERE on the device, each ~130 columns behind the one above.  The horizontal deltas of a tile's last
//     block cross to the next wave through memory: 2 bits per column, 64 columns per hand-over = two 8-byte write-through stores and
//     a progress word (agent-scope relaxed atomics: MI355X_MICROARCH.md, "inter-workgroup visibility", form R1);
//   * the sub-problems of one Hirschberg level are independent.
// Structure: PHASES, one launch each.  Phase 0 = the score sweeps of all giants; phase l + 1 = the half sweeps of Hirschberg level l
// of all giants.  A phase is a list of TILE JOBS; waves draw them with an atomic ticket IN LIST ORDER, and a sweep's tiles are
// consecutive in the list, top tile first — so whenever a wave waits for the tile above, that tile has been drawn by a wave that
// is running: no deadlock whatever the residency.  The wave that finishes the LAST tile of a node (both half sweeps; counted down
// with release / acquire fences) splits the node (edlib.cpp:1317-1356, wv::hirschberg_split) and appends its children to the next
// phase's lists: sub-problems whose history fits edlib's 1 MiB become LEAVES (traced back by a wave each in a last launch, into a
// sparse operations buffer at position query offset + target offset, as in round 3), the others new nodes.  Same split rows, same
// tie-breaking as edlib's sequential recursion: the split is a function of the two half sweeps only.
#pragma once
#include "align_wave.hpp"

namespace gt {
using wv::lane_id; using wv::bcast; using wv::bcast_first; using wv::shr1;

constexpr uint32_t MAX_PHASES = 28;               // score + up to 27 Hirschberg levels (columns halve per level)
constexpr uint32_t NODE_CAP = 1u << 16, JOB_CAP = 1u << 19, LEAF_CAP = 1u << 20, FLAG_CAP = 1u << 20;
constexpr uint32_t SPIN_LIMIT = 1u << 22;         // polls of a progress word before a tile gives up (the gap then goes to the wave kernel)

struct Sweep {
	const uint8_t* q; const uint8_t* t; int32_t qstep, tstep; uint32_t n, ne, m, shw;
	int32_t* lastcol;                             // ne + 1 values D[i][m], or null (score sweeps)
	unsigned long long* hand;                     // (tiles - 1) x chunks x {plus mask, minus mask}: horizontal deltas leaving a tile's last block, 64 columns per pair
	uint32_t* prog;                               // (tiles - 1) progress words: pairs written
	uint32_t tiles, node;                         // node: index of the owning node in this phase's node list
	uint32_t score, best; int32_t end; uint32_t pad;
};
struct Node { uint32_t giant, qo, n, to, m, best, L, R, neL, neR; int32_t* left; int32_t* right; uint32_t pending, sweep0; };   // phase 0: the giant's score sweep (sweep0 only)
struct Leaf { uint32_t giant, qo, n, to, m; };
struct Job { uint32_t sweep, tile; };
struct Giant {
	uint32_t gi, n, m, kind; uint32_t shw, rows_ref, left, fail; uint32_t ref_end_nw, pad;   // ref_end_nw: where a global alignment ends in the reference (use - 1 for a tiny flank, else 0)
	uint8_t* rbuf; uint8_t* ebuf; uint8_t* r2; uint8_t* e2; uint8_t* opsbuf; uint8_t* sparse;
	const uint8_t* Q; const uint8_t* T;
	uint32_t mp, ref_end;                         // columns of the path (m, or end + 1 for a flank), where the alignment ends in the reference
};
struct Ctl {
	unsigned long long top;                       // bump allocator over `heap`
	uint32_t n_flags, n_leaves, leaf_ticket, n_failed;
	uint32_t n_nodes[MAX_PHASES + 1], n_sweeps[MAX_PHASES + 1], n_jobs[MAX_PHASES + 1], ticket[MAX_PHASES + 1];
};
struct View {
	Ctl* ctl; uint8_t* heap; unsigned long long heap_bytes; uint32_t* flags;
	Node* nodes[2]; Sweep* sweeps[2]; Job* jobs[2]; Leaf* leaves; Giant* giants; uint32_t n_giants;
	uint32_t n_phases = MAX_PHASES;             // the phases the host launches (k_giant_level 0 .. n_phases - 1): a node of a later phase would never run
};

// ---- wave-uniform helpers (every lane calls; lane 0 does the atomic) --------------------------------------------------------
__device__ inline uint8_t* galloc(const View& V, unsigned long long bytes)
{
	bytes = (bytes + 255) & ~255ull;
	unsigned long long off = 0;
	if (lane_id() == 0) off = atomicAdd(&V.ctl->top, bytes);
	off = ((unsigned long long)bcast_first((uint32_t)(off >> 32)) << 32) | bcast_first((uint32_t)off);
	return off + bytes <= V.heap_bytes ? V.heap + off : nullptr;
}
__device__ inline uint32_t take(uint32_t* counter, uint32_t k)
{
	uint32_t v = 0;
	if (lane_id() == 0) v = atomicAdd(counter, k);
	return bcast_first(v);
}
__device__ inline void giant_fail(const View& V, uint32_t giant, uint32_t why) { if (lane_id() == 0) atomicMax(&V.giants[giant].fail, why); }

// One sub-problem q[qo .. qo + n) x t[to .. to + m) with optimal score `best` of giant `giant` (obtainAlignment's recursion,
// edlib.cpp:1164-1215,1230-1400): a leaf when edlib would trace it back directly or one side is empty, else a node of phase `ph`
// with its two half sweeps (target cut in the middle, edlib.cpp:1240-1250) and their tile jobs.
__device__ inline void emit_sub(const View& V, uint32_t giant, uint32_t qo, uint32_t n, uint32_t to, uint32_t m, uint32_t best, uint32_t ph)
{
	const uint32_t lane = lane_id();
	const Giant& G = V.giants[giant];
	if (n == 0 || m == 0 || wv::wave_direct_fits(n, m))
	{
		const uint32_t li = take(&V.ctl->n_leaves, 1);
		if (li >= LEAF_CAP) { giant_fail(V, giant, 10); return; }
		if (lane == 0) V.leaves[li] = Leaf{ giant, qo, n, to, m };
		return;
	}
	if (ph >= V.n_phases || ph > MAX_PHASES) { giant_fail(V, giant, 11); return; }   // (few columns but too many rows for a leaf at the last launched phase: the wave kernel takes the gap)
	const uint32_t L = m / 2, R = m - L;
	const uint8_t* ql = G.Q + qo; const uint8_t* tl = G.T + to;
	const uint8_t* qr = G.Q + qo + n - 1; const uint8_t* tr = G.T + to + m - 1;
	const uint32_t neL = wv::sat_rows(ql, 1, n, tl, 1, L), neR = wv::sat_rows(qr, -1, n, tr, -1, R);
	const uint32_t tL = ((neL + 63) / 64 + 63) / 64, tR = ((neR + 63) / 64 + 63) / 64;
	const uint32_t cL = (L + 63) / 64, cR = (R + 63) / 64;
	int32_t* left = (int32_t*)galloc(V, ((unsigned long long)neL + 1) * 4);
	int32_t* right = (int32_t*)galloc(V, ((unsigned long long)neR + 1) * 4);
	unsigned long long* hL = tL > 1 ? (unsigned long long*)galloc(V, (unsigned long long)(tL - 1) * cL * 16) : nullptr;
	unsigned long long* hR = tR > 1 ? (unsigned long long*)galloc(V, (unsigned long long)(tR - 1) * cR * 16) : nullptr;
	if (!left || !right || (tL > 1 && !hL) || (tR > 1 && !hR)) { giant_fail(V, giant, 12); return; }
	const uint32_t f0 = take(&V.ctl->n_flags, tL - 1 + tR - 1);
	const uint32_t ni = take(&V.ctl->n_nodes[ph], 1), si = take(&V.ctl->n_sweeps[ph], 2), ji = take(&V.ctl->n_jobs[ph], tL + tR);
	if (f0 + tL + tR - 2 > FLAG_CAP || ni >= NODE_CAP || si + 2 > 2 * NODE_CAP || ji + tL + tR > JOB_CAP)
	{	// (the job slots taken are drawn all the same: they must say "nothing to do")
		for (uint32_t x = lane; x < tL + tR; x += 64) if (ji + x < JOB_CAP) V.jobs[ph & 1][ji + x] = Job{ 0xffffffffu, 0 };
		giant_fail(V, giant, 13);
		return;
	}
	if (lane == 0)
	{
		V.nodes[ph & 1][ni] = Node{ giant, qo, n, to, m, best, L, R, neL, neR, left, right, tL + tR, si };
		V.sweeps[ph & 1][si] = Sweep{ ql, tl, 1, 1, n, neL, L, 0, left, hL, V.flags + f0, tL, ni, 0, 0, 0, 0 };
		V.sweeps[ph & 1][si + 1] = Sweep{ qr, tr, -1, -1, n, neR, R, 0, right, hR, V.flags + f0 + (tL - 1), tR, ni, 0, 0, 0, 0 };
	}
	Job* jb = V.jobs[ph & 1] + ji;
	for (uint32_t x = lane; x < tL + tR; x += 64) jb[x] = x < tL ? Job{ si, x } : Job{ si + 1, x - tL };
}

// One tile of a sweep (wv::sweep_tile with the hand-over through memory): block tile * 64 + lane on this lane, anti-diagonal
// steps.  Returns false when the tile above did not deliver (spin limit).
__device__ inline bool giant_tile(Sweep* Sp, uint32_t tile)
{
	const uint32_t lane = lane_id();
	const uint8_t* const q = Sp->q; const uint8_t* const t = Sp->t; const int qstep = Sp->qstep, tstep = Sp->tstep;
	const uint32_t n = Sp->n, ne = Sp->ne, m = Sp->m; const bool shw = Sp->shw != 0, sat = ne < n;
	const uint32_t tiles = Sp->tiles, chunks = (m + 63) / 64;
	const unsigned long long* const hin = tile ? Sp->hand + (unsigned long long)(tile - 1) * chunks * 2 : nullptr;
	unsigned long long* const hout_arr = tile + 1 < tiles ? Sp->hand + (unsigned long long)tile * chunks * 2 : nullptr;
	uint32_t* const prog_in = tile ? Sp->prog + (tile - 1) : nullptr; uint32_t* const prog_out = tile + 1 < tiles ? Sp->prog + tile : nullptr;
	int32_t* const lastcol = Sp->lastcol;
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
	uint32_t sc = n, best = 0xffffffffu; int32_t end = (int32_t)m - 1;
	if (!sat && shw && (n & 63)) { best = n; end = -1; }
	if (lastcol && tile == 0 && lane == 0) lastcol[0] = (int32_t)m;
	uint32_t c = 0; int hout = 0; uint32_t tchunk = 0;
	uint64_t hp = ~0ull, hm = 0;                                              // horizontal deltas entering this tile, 64 columns (tile 0: +1 everywhere)
	uint64_t op = 0, om = 0;                                                  // ... and leaving it (lane W - 1 collects)
	bool ok = true;
	const uint32_t steps = m + W - 1;
	for (uint32_t s = 0; s < steps; ++s)
	{
		if ((s & 63) == 0)
		{
			const uint32_t j0 = s + lane;
			tchunk = j0 < m ? (uint32_t)(t[(int64_t)j0 * tstep] & 3) : 0u;
			if (prog_in && s < m)
			{	// the tile above has to be through columns s .. s + 63: pair s / 64 written
				const uint32_t need = s / 64 + 1;
				for (uint32_t spins = 0;; )
				{	// (every lane reads the same word; the first lane's value decides for the wave)
					const uint32_t have = bcast_first(__hip_atomic_load(prog_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
					if (have >= need) break;
					__builtin_amdgcn_s_sleep(8);
					if (++spins > SPIN_LIMIT) { ok = false; break; }
				}
				if (!ok) break;
				__atomic_signal_fence(__ATOMIC_SEQ_CST);
				hp = __hip_atomic_load(hin + (unsigned long long)(s / 64) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				hm = __hip_atomic_load(hin + (unsigned long long)(s / 64) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				hp = ((uint64_t)bcast_first((uint32_t)(hp >> 32)) << 32) | bcast_first((uint32_t)hp);
				hm = ((uint64_t)bcast_first((uint32_t)(hm >> 32)) << 32) | bcast_first((uint32_t)hm);
			}
		}
		const uint32_t c_new = bcast(tchunk, s & 63);
		const int h_new = (int)((hp >> (s & 63)) & 1) - (int)((hm >> (s & 63)) & 1);
		const uint32_t c_up = shr1(c); const int h_up = shr1(hout);
		c = lane == 0 ? c_new : c_up;
		const int hin_v = lane == 0 ? h_new : h_up;
		const bool valid = act && s >= lane && s - lane < m;
		hout = 0;
		if (valid)
		{
			uint64_t Eq = c == 0 ? e0 : c == 1 ? e1 : c == 2 ? e2 : e3;
			const uint64_t hneg = hin_v < 0 ? 1ull : 0ull;
			const uint64_t Xv = Eq | Mv;
			Eq |= hneg;
			const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
			uint64_t Ph = Mv | ~(Xh | Pv);
			uint64_t Mh = Pv & Xh;
			if (owner)
			{
				sc += (uint32_t)((Ph >> lastbit) & 1) - (uint32_t)((Mh >> lastbit) & 1);
				if (shw && sc < best) { best = sc; end = (int32_t)(s - lane); }
			}
			hout = (int)(Ph >> 63) - (int)(Mh >> 63);
			Ph <<= 1; Mh <<= 1;
			Mh |= hneg; Ph |= hin_v > 0 ? 1ull : 0ull;
			Pv = Mh | ~(Xv | Ph);
			Mv = Ph & Xv;
			S += hout;
			if (hout_arr && lane == W - 1)
			{
				const uint32_t col = s - lane;
				op |= (uint64_t)(hout > 0) << (col & 63); om |= (uint64_t)(hout < 0) << (col & 63);
				if ((col & 63) == 63 || col == m - 1)
				{	// 64 columns through: the pair, write-through, then the progress word (one lane: its own store order)
					__hip_atomic_store(hout_arr + (unsigned long long)(col / 64) * 2, op, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(hout_arr + (unsigned long long)(col / 64) * 2 + 1, om, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					__hip_atomic_store(prog_out, col / 64 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					op = 0; om = 0;
				}
			}
		}
	}
	if (!ok)
	{	// (nobody below may wait for ever either)
		if (prog_out && lane == 0) __hip_atomic_store(prog_out, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return false;
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
	if (tile + 1 == tiles)
	{
		if (sat) { if (lane == 0) { Sp->score = n - m; Sp->best = n - m; Sp->end = (int32_t)m - 1; } }
		else if (owner) { Sp->score = sc; Sp->best = best; Sp->end = end; }
	}
	return true;
}

} // namespace gt
