// encode_es.hip — kernels and host driver of the edit-script encoder (a10 + a11 + a12); the algorithm, its data
// model (frames, gaps, levels) and the reference citations are in encode_core.hpp.
//
// Per recursion level L (host loop, <= maxRecurence + 1 iterations, each a handful of launches over ALL level-L gaps):
//   k_frame_gap_counts -> scan -> k_gap_geometry -> scans (script capacity, pending slots) -> sort gaps by size class
//   -> k_align_small<NB> (one lane per gap, rows <= 64*NB, columns <= 256: Myers' recurrence in registers, sequences and
//      the script in LDS, per-column history lane-interleaved in HBM) / k_align_large (the rest: lane pool, Hirschberg)
//   -> k_gap_stats (estimator statistics or the static entropy decision) -> k_spawn_mark -> scans -> k_spawn_fill.
// Then k_estimator (one lane per reader pack) and k_emit_tuples (one lane per read, count pass + write pass).
#include "objects.hpp"
#include "encode_core.hpp"
#include "align_wave.hpp"
#include "align_rows.hpp"
#include "align_giant.hpp"
#include "emit_wave.hpp"
#include <algorithm>
#include <memory>
#include <vector>
#include <thread>
#include <chrono>

struct cl_anchors;   // anchors.hip
extern "C" const uint32_t* cl_anchors_n_cands(const cl_anchors* a);
extern "C" const uint32_t* cl_anchors_cands(const cl_anchors* a);
extern "C" const uint64_t* cl_anchors_cand_offsets(const cl_anchors* a);
extern "C" const uint32_t* cl_anchors_data(const cl_anchors* a);

using namespace enc;

namespace {
struct AnchorsV { const uint32_t* n_cands; const uint32_t* cand; const uint64_t* cand_off; const uint32_t* data; };

// ---- level 0: one frame per read that has candidates ---------------------------------------------------------------
__global__ void k_read_flags(const uint32_t* __restrict__ n_cands, const uint8_t* __restrict__ has_n, uint32_t n, uint32_t* __restrict__ flag, uint32_t* __restrict__ ncand)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n) return;
	const uint32_t nc = has_n[r] ? 0u : n_cands[r];
	flag[r] = nc ? 1u : 0u; ncand[r] = nc;
}
__global__ void k_frames_level0(ArenaV A, AnchorsV AV, uint32_t c, const uint8_t* __restrict__ has_n, uint32_t n, const uint32_t* __restrict__ frame_idx, const uint64_t* __restrict__ cand_base,
                                FrameRec* __restrict__ frames, CandEnt* __restrict__ cands, uint32_t* __restrict__ frame_of_read)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n) return;
	const uint32_t nc = has_n[r] ? 0u : AV.n_cands[r];
	if (!nc) { frame_of_read[r] = 0xffffffffu; return; }
	const uint32_t f = frame_idx[r];
	frame_of_read[r] = f;
	FrameRec F; F.read = r; F.level = 0; F.enc_off = 0; F.enc_len = A.lens[r]; F.n_cands = nc; F.first_gap = 0; F.n_gaps = 0; F.pad = 0; F.cand_base = cand_base[r];
	frames[f] = F;
	for (uint32_t j = 0; j < nc; ++j) cands[F.cand_base + j] = cand_level0(AV.cand + ((uint64_t)r * c + j) * 4, AV.cand_off[(uint64_t)r * c + j], AV.data);
}

// ---- per level ---------------------------------------------------------------------------------------------------
__global__ void k_frame_gap_counts(const FrameRec* __restrict__ frames, const CandEnt* __restrict__ cands, uint32_t n, uint32_t* __restrict__ counts)
{
	const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f < n) counts[f] = cands[frames[f].cand_base + frames[f].level].n + 1;
}
__global__ void k_frame_first_gaps(FrameRec* __restrict__ frames, const uint32_t* __restrict__ first, uint32_t n, uint32_t n_gaps)
{
	const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= n) return;
	frames[f].first_gap = first[f];
	frames[f].n_gaps = (f + 1 < n ? first[f + 1] : n_gaps) - first[f];
}
__global__ void k_gap_geometry(LevelV L, ArenaV R, const uint32_t* __restrict__ data, uint32_t min_part_alt, const uint32_t* __restrict__ first_gap,
                               uint32_t* __restrict__ cap, uint32_t* __restrict__ pend_flag, uint32_t* __restrict__ keys, uint32_t* __restrict__ ids, uint32_t* __restrict__ err)
{
	const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
	if (gi >= L.n_gaps) return;
	uint32_t lo = 0, hi = L.n_frames;                                        // the frame that owns gap gi: last f with first_gap[f] <= gi
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (first_gap[mid] <= gi) lo = mid; else hi = mid; }
	GapRec g;
	if (!gap_init(L, lo, gi - first_gap[lo], data, R, g)) atomicOr(err, 4u);
	L.gaps[gi] = g;
	cap[gi] = gap_es_capacity(g);
	pend_flag[gi] = g.ne < min_part_alt ? 1u : 0u;
	keys[gi] = gap_sort_key(g);
	ids[gi] = gi;
}
__global__ void k_gap_offsets(GapRec* __restrict__ gaps, const uint64_t* __restrict__ es_off, uint32_t n)
{
	const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
	if (gi < n) gaps[gi].es_off = es_off[gi];
}
// bounds[N_CLASSES + 1], [N_CLASSES + 2]: over the giant gaps (class 7) the sum of their script capacities (reference + read symbols)
// in units of 256 and the largest one — what the host sizes their heap and the number of their phases by (zeroed by the caller)
__global__ void k_class_bounds(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ ids, const uint32_t* __restrict__ cap, uint32_t n, uint32_t* __restrict__ bounds /* N_CLASSES + 3 */)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n) return;
	if (i < n && (keys[i] >> 17) == 7) { const uint32_t c = cap[ids[i]]; atomicAdd(&bounds[N_CLASSES + 1], (c + 255) / 256); atomicMax(&bounds[N_CLASSES + 2], c); }
	const uint32_t a = i == 0 ? 0u : (keys[i - 1] >> 17) + 1, b = i == n ? N_CLASSES + 1 : (keys[i] >> 17) + 1;   // classes [a, b) start at i
	for (uint32_t c = a; c < b && c <= N_CLASSES; ++c) bounds[c] = i;
}

// algorithmic bytes of each size class: 2-bit symbols in, one script byte per symbol out — and its DP cells, rows x columns of every
// gap (what edlib's recurrence has to fill once; Hirschberg's second and third sweep over a large gap are not counted) — for the report
__global__ void k_class_bytes(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ ids, const uint32_t* __restrict__ cap, const GapRec* __restrict__ gaps, uint32_t n, unsigned long long* __restrict__ bytes /* 2 x N_CLASSES */)
{
	__shared__ unsigned long long s_b[2 * N_CLASSES];
	if (threadIdx.x < 2 * N_CLASSES) s_b[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		uint32_t rows, cols; const uint32_t cls = gap_class(gaps[ids[i]], rows, cols);
		atomicAdd(&s_b[cls], (unsigned long long)cap[ids[i]]);
		atomicAdd(&s_b[N_CLASSES + cls], (unsigned long long)rows * cols);
	}
	__syncthreads();
	if (threadIdx.x < 2 * N_CLASSES && s_b[threadIdx.x]) atomicAdd(&bytes[threadIdx.x], s_b[threadIdx.x]);
}

// lane-private staging memory in LDS (word w of a lane at lds[w * 64 + lane]) + column history in HBM (lane-interleaved).
// PACKED: the sequences at 2 bits per symbol, the script at 4 bits per symbol (its nine letters M D A C G T X Y Z as 0..8) — 60 to 96
// words per lane (15 - 25 KB per wave) instead of 160 - 256 (40 - 64 KB with a byte per symbol, rounds 1-3: four such waves took a
// CU's whole LDS, and every other kernel that needs LDS — the sorts, the table build, the match pass — waited for them).
template<int NB>
struct LdsMem {
	static constexpr uint32_t QW = 4 * NB, TW = 16, EW = 8 * NB + 32;         // words: rows (64 NB symbols), columns (256), script (64 NB + 256 symbols)
	uint32_t* base; uint64_t* hist; uint32_t lane;
	__device__ inline uint32_t& word(uint32_t w) const { return base[w * 64 + lane]; }
	__device__ inline uint32_t q(uint32_t i) const { return (word(i >> 4) >> (2 * (i & 15))) & 3u; }
	__device__ inline uint32_t t(uint32_t j) const { return (word(QW + (j >> 4)) >> (2 * (j & 15))) & 3u; }
	__device__ inline void q_set(uint32_t i, uint32_t v) { uint32_t& w = word(i >> 4); const uint32_t sh = 2 * (i & 15); w = (w & ~(3u << sh)) | ((v & 3u) << sh); }
	__device__ inline void t_set(uint32_t j, uint32_t v) { uint32_t& w = word(QW + (j >> 4)); const uint32_t sh = 2 * (j & 15); w = (w & ~(3u << sh)) | ((v & 3u) << sh); }
	static __device__ inline uint32_t code_of(char c) { return c == 'M' ? 0u : c == 'D' ? 1u : c == 'A' ? 2u : c == 'C' ? 3u : c == 'G' ? 4u : c == 'T' ? 5u : (uint32_t)(c - 'X') + 6u; }
	static __device__ inline char char_of(uint32_t v) { return v == 0 ? 'M' : v == 1 ? 'D' : v == 2 ? 'A' : v == 3 ? 'C' : v == 4 ? 'G' : v == 5 ? 'T' : (char)('X' + (v - 6)); }
	__device__ inline char es_get(uint32_t k) const { return char_of((word(QW + TW + (k >> 3)) >> (4 * (k & 7))) & 15u); }
	__device__ inline void es_set(uint32_t k, char c) { uint32_t& w = word(QW + TW + (k >> 3)); const uint32_t sh = 4 * (k & 7); w = (w & ~(15u << sh)) | (code_of(c) << sh); }
	// script symbols 4 w .. 4 w + 3 as the four bytes of an output word
	__device__ inline uint32_t es_word(uint32_t w) const
	{
		const uint32_t x = (word(QW + TW + (w >> 1)) >> (16 * (w & 1))) & 0xffffu;
		return (uint32_t)(uint8_t)char_of(x & 15u) | ((uint32_t)(uint8_t)char_of((x >> 4) & 15u) << 8) | ((uint32_t)(uint8_t)char_of((x >> 8) & 15u) << 16) | ((uint32_t)(uint8_t)char_of(x >> 12) << 24);
	}
	__device__ inline void lap(uint32_t) {}
	__device__ inline void hist_put(uint32_t j, uint32_t b, uint64_t P, uint64_t Ph) { uint64_t* h = hist + ((uint64_t)(j * NB + b) * 2) * 64 + lane; h[0] = P; h[64] = Ph; }
	__device__ inline void hist_get(uint32_t j, uint32_t b, uint64_t& P, uint64_t& Ph) const { const uint64_t* h = hist + ((uint64_t)(j * NB + b) * 2) * 64 + lane; P = h[0]; Ph = h[64]; }
};

template<int NB>
__global__ __launch_bounds__(64) void k_align_small(const uint32_t* __restrict__ list, uint32_t n_list, GapRec* __restrict__ gaps, char* __restrict__ es_pool,
                                                   ArenaV A, ArenaV R, uint64_t* __restrict__ hist_all)
{
	extern __shared__ uint32_t lds_raw[];
	LdsMem<NB> mem{ lds_raw, hist_all + (uint64_t)blockIdx.x * (256ull * NB * 2 * 64), threadIdx.x };
	for (uint32_t chunk = blockIdx.x; (uint64_t)chunk * 64 < n_list; chunk += gridDim.x)
	{
		const uint32_t idx = chunk * 64 + threadIdx.x;
		if (idx >= n_list) continue;
		const uint32_t gi = list[idx];
		const GapRec g = gaps[gi];
		uint32_t n, m;
		stage_small(mem, g, A, R, n, m);
		uint32_t d_before;
		const uint32_t k = align_small<NB>(mem, n, m, g.kind, g.left != 0, g.nr, g.use, &d_before);
		uint32_t* dst = (uint32_t*)(es_pool + g.es_off);
		for (uint32_t w = 0; w * 4 < k; ++w) dst[w] = mem.es_word(w);
		gaps[gi].es_len = k; gaps[gi].d_before = d_before;
	}
}

// large gaps: one WAVE per gap (align_wave.hpp), largest first.  Three steps: the sequences into byte buffers (stage), sweep + path -> operations, operations -> canonical script.
struct WaveGap {
	uint8_t* rbuf; uint8_t* ebuf; uint8_t* r2; uint8_t* e2; uint8_t* opsbuf;
	const uint8_t* Q; const uint8_t* T; uint32_t n, m; bool rows_ref, shw, left;     // m: columns offered to the sweep (g.use for a flank)
};
__device__ inline bool wave_gap_stage(wv::WavePool& pool, const GapRec& g, const ArenaV& A, const ArenaV& R, WaveGap& W)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t ref_id = g.ref_rev & 0x7fffffffu; const bool rev = g.ref_rev >> 31;
	const uint64_t rwb = R.word_off[ref_id], ewb = A.word_off[g.read]; const uint32_t rlen = R.lens[ref_id];
	const bool left = g.left != 0;
	W.left = left;
	W.rbuf = (uint8_t*)pool.alloc(g.use + 64ull); W.ebuf = (uint8_t*)pool.alloc(g.ne + 64ull);
	W.r2 = (uint8_t*)pool.alloc(g.use + 64ull); W.e2 = (uint8_t*)pool.alloc(g.ne + 64ull);
	W.opsbuf = (uint8_t*)pool.alloc((uint64_t)g.use + g.ne + 64);
	if (pool.overflow) return false;
	const uint32_t lo = left ? g.nr - g.use : 0;
	// eight symbols per lane and step: two packed words in, one 8-byte store per buffer out (a symbol per lane was use / 64 + ne / 64 load
	// latencies in a row and four byte stores per symbol pair).  A reverse-complemented reference is read backwards: eight ascending places
	// of the stored read, bytes swapped, complemented.
	auto sym8 = [](const ArenaV& X, uint64_t wb, uint32_t fp) -> uint64_t {      // bases fp .. fp + 7 of a stored read, one per byte
		const uint64_t wa = X.packed[wb + (fp >> 5)], wz = X.packed[wb + ((fp + 7) >> 5)];
		uint64_t out = 0;
#pragma unroll
		for (uint32_t j = 0; j < 8; ++j) { const uint32_t p = fp + j; const uint64_t w = (p >> 5) == (fp >> 5) ? wa : wz; out |= ((w >> (62 - 2 * (p & 31))) & 3ull) << (8 * j); }
		return out;
	};
	auto put8 = [&](uint8_t* fwd, uint8_t* other, uint32_t total, uint32_t i, uint64_t v) {     // places i .. i + 7 of fwd; `other` is fwd or its mirror image
		__builtin_memcpy(fwd + i, &v, 8);
		if (!left) __builtin_memcpy(other + i, &v, 8);
		else { const uint64_t m = __builtin_bswap64(v); __builtin_memcpy(other + (total - 8 - i), &m, 8); }
	};
	for (uint32_t i = lane * 8; i < g.use; i += 512)
	{
		const uint32_t pos0 = g.cur_ref + lo + i;
		if (i + 8 <= g.use) put8(W.rbuf, W.r2, g.use, i, rev ? 0x0303030303030303ull - __builtin_bswap64(sym8(R, rwb, rlen - 1 - (pos0 + 7))) : sym8(R, rwb, pos0));
		else for (uint32_t y = i; y < g.use; ++y) { const uint8_t v = (uint8_t)ref_sym(R, rwb, rlen, rev, g.cur_ref + lo + y); W.rbuf[y] = v; W.r2[left ? g.use - 1 - y : y] = v; }
	}
	for (uint32_t i = lane * 8; i < g.ne; i += 512)
	{
		if (i + 8 <= g.ne) put8(W.ebuf, W.e2, g.ne, i, sym8(A, ewb, g.enc_start + i));
		else for (uint32_t y = i; y < g.ne; ++y) { const uint8_t v = (uint8_t)arena_base_at(A, ewb, g.enc_start + y); W.ebuf[y] = v; W.e2[left ? g.ne - 1 - y : y] = v; }
	}
	if (g.kind == GK_INNER) { W.Q = W.rbuf; W.n = g.nr; W.T = W.ebuf; W.m = g.ne; W.rows_ref = true; W.shw = false; }
	else if (g.kind == GK_FLANK_TINY) { W.Q = W.r2; W.n = g.use; W.T = W.e2; W.m = g.ne; W.rows_ref = true; W.shw = false; }
	else { W.Q = W.e2; W.n = g.ne; W.T = W.r2; W.m = g.use; W.rows_ref = false; W.shw = true; }
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	return true;
}
__device__ inline bool wave_gap_finish(wv::WavePool& pool, GapRec& g, const WaveGap& W, const wv::Ops& ops, uint32_t ref_end, char* dst, uint32_t dbg_stage);
__device__ inline bool align_wave_gap(wv::WavePool& pool, GapRec& g, const ArenaV& A, const ArenaV& R, char* dst, uint32_t dbg_stage)
{
	g.es_len = 0; g.d_before = 0;
	WaveGap W;
	pool.beat(4);
	pool.lap(0);
	if (!wave_gap_stage(pool, g, A, R, W)) return false;
	uint8_t* const opsbuf = W.opsbuf; uint8_t* const rbuf = W.rbuf; uint8_t* const ebuf = W.ebuf; uint8_t* const r2 = W.r2; uint8_t* const e2 = W.e2;
	(void)rbuf; (void)ebuf; (void)r2; (void)e2;
	pool.beat(5);
	pool.lap(1);
	if (dbg_stage == 1) return true;
	wv::Ops ops{ opsbuf, 0 };
	const uint8_t* Q; const uint8_t* T; uint32_t n, m; uint32_t ref_end = 0;
	// when edlib would keep the whole history anyway (it decides on the truncated target, which is never longer), one
	// sweep delivers both the score / end position and the history; else score sweep first, then divide and conquer
	if (g.kind == GK_FLANK_TINY && g.use == 1 && g.ne >= 1)
	{	// ONE reference symbol against a flank of the read (seen: 1 x 121 957, 87 ms through the generic sweep + Hirschberg, the
		// slowest gap of its launch).  find_edit_dist (edit_script.h:156-239) on one row: c[1][j] = j - 1 once the symbol has
		// occurred among the first j read symbols, else j; its traceback (up, else left, else diagonal) walks left to the FIRST
		// occurrence and matches there — or, without any occurrence, substitutes the first read symbol.
		Q = r2; T = e2; n = 1; m = g.ne; ref_end = 0;
		const uint32_t lane = threadIdx.x & 63;
		const uint32_t qs = Q[0];
		uint32_t first = m;
		for (uint32_t base = 0; base < m && first == m; base += 256)
		{
			bool hit[4];
#pragma unroll
			for (uint32_t u = 0; u < 4; ++u) { const uint32_t j = base + u * 64 + lane; hit[u] = j < m && T[j] == qs; }
#pragma unroll
			for (uint32_t u = 0; u < 4; ++u) { const uint64_t bal = __ballot(hit[u]); if (bal && first == m) first = base + u * 64 + (uint32_t)__builtin_ctzll(bal); }
		}
		const uint32_t at = first < m ? first : 0; const uint8_t op_at = first < m ? 0 : 3;
		for (uint32_t j = lane; j < m; j += 64) opsbuf[j] = j == at ? op_at : 2;
		ops.n = m;
		pool.lap(2);
	}
	else if (g.kind == GK_INNER || g.kind == GK_FLANK_TINY)
	{
		if (g.kind == GK_INNER) { Q = rbuf; n = g.nr; T = ebuf; m = g.ne; }
		else { Q = r2; n = g.use; T = e2; m = g.ne; ref_end = g.use - 1; }
		if (n && m && wv::wave_direct_fits(n, m))
		{
			wv::wave_align_direct(pool, Q, n, T, m, false, ops);
			pool.lap(2);
		}
		else
		{
			const wv::Sweep sw = wv::wave_sweep(pool, Q, 1, n, wv::sat_rows(Q, 1, n, T, 1, m), T, 1, m, false, nullptr, nullptr);
			if (pool.overflow) return false;
			pool.lap(2);
			if (dbg_stage == 2) return true;
			wv::wave_path(pool, Q, n, T, m, sw.score, ops);
		}
	}
	else
	{
		Q = e2; n = g.ne; T = r2;
		if (n && g.use && wv::wave_direct_fits(n, g.use))
		{
			const wv::Sweep sw = wv::wave_align_direct(pool, Q, n, T, g.use, true, ops);
			ref_end = (uint32_t)sw.end; m = (uint32_t)(sw.end + 1);
			pool.lap(2);
		}
		else
		{
			const wv::Sweep sw = wv::wave_sweep(pool, Q, 1, n, wv::sat_rows(Q, 1, n, T, 1, g.use), T, 1, g.use, true, nullptr, nullptr);
			if (pool.overflow) return false;
			ref_end = (uint32_t)sw.end; m = (uint32_t)(sw.end + 1);
			pool.lap(2);
			if (dbg_stage == 2) return true;
			wv::wave_path(pool, Q, n, T, m, sw.best, ops);
		}
	}
	if (pool.overflow) return false;
	(void)m;
	return wave_gap_finish(pool, g, W, ops, ref_end, dst, dbg_stage);
}
// operations (0 match, 1 consume query, 2 consume target, 3 mismatch) -> the gap's canonical script
__device__ inline bool wave_gap_finish(wv::WavePool& pool, GapRec& g, const WaveGap& W, const wv::Ops& ops, uint32_t ref_end, char* dst, uint32_t dbg_stage)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint8_t* const Q = W.Q; const uint8_t* const T = W.T; const bool rows_ref = W.rows_ref, left = W.left;
	const uint8_t* const rbuf = W.rbuf; const uint8_t* const ebuf = W.ebuf;
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	pool.beat(6);
	pool.lap(3);
	if (dbg_stage == 3) return true;
	// operations -> script symbols, 64 at a time; for the left flank the script of the reversed sequences is written reversed
	const uint32_t k = (uint32_t)ops.n;
	// (loads run ahead of the position carries: operations two steps ahead, the sequence symbols they select one step ahead)
	{
		const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
		auto load_op = [&](uint32_t x0) -> uint32_t { const uint32_t x = x0 + lane; return x < k ? (uint32_t)ops.p[x] : 4u; };
		uint32_t opA = load_op(0), opB = load_op(64);
		uint32_t pq = 0, pt = 0, qA = 0, tA = 0;
		{
			const uint64_t cq = __ballot(opA != 2 && opA != 4), ct = __ballot(opA != 1 && opA != 4);
			const uint32_t myq = (uint32_t)__popcll(cq & lt), myt = (uint32_t)__popcll(ct & lt);
			if (opA == 1 || opA == 3) qA = Q[myq];
			if (opA == 2 || opA == 3) tA = T[myt];
			pq = (uint32_t)__popcll(cq); pt = (uint32_t)__popcll(ct);
		}
		for (uint32_t x0 = 0; x0 < k; x0 += 64)
		{
			const uint32_t x = x0 + lane;
			const uint32_t op = opA, qs = qA, ts = tA;
			const uint32_t opC = load_op(x0 + 128);
			uint32_t qB = 0, tB = 0;
			{	// the next step's symbols
				const uint64_t cq = __ballot(opB != 2 && opB != 4), ct = __ballot(opB != 1 && opB != 4);
				const uint32_t myq = pq + (uint32_t)__popcll(cq & lt), myt = pt + (uint32_t)__popcll(ct & lt);
				if (opB == 1 || opB == 3) qB = Q[myq];
				if (opB == 2 || opB == 3) tB = T[myt];
				pq += (uint32_t)__popcll(cq); pt += (uint32_t)__popcll(ct);
			}
			if (op != 4)
			{
				char ch;
				if (op == 0) ch = 'M';
				else if (op == 1) ch = rows_ref ? 'D' : base_letter(qs);
				else if (op == 2) ch = rows_ref ? base_letter(ts) : 'D';
				else ch = rows_ref ? mismatch_sym(qs, ts) : mismatch_sym(ts, qs);
				dst[left ? k - 1 - x : x] = ch;
			}
			opA = opB; opB = opC; qA = qB; tA = tB;
		}
	}
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	pool.beat(7);
	pool.lap(4);
	if (dbg_stage == 4) return true;
	// canonical indel placement (wave-parallel segmented form of refactor_edit_script, align_wave.hpp)
	uint32_t d_before = 0;
	const uint8_t* rf = rbuf;
	if (left && g.kind != GK_INNER)
	{
		const uint32_t ref_offset = (g.nr - 1) - ref_end;                      // uint32 wrap for end = -1, as in the reference
		rf = rbuf + (ref_offset - (g.nr - g.use));
		d_before = ref_offset;
	}
#ifdef CL_DEBUG_REFACTOR
	char* t0 = (char*)pool.alloc(k + 64ull); char* t1 = (char*)pool.alloc(k + 64ull);
	if (pool.overflow) return false;
	for (uint32_t i = lane; i < k; i += 64) { t0[i] = dst[i]; t1[i] = dst[i]; }
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	{
		struct WES { char* p; __device__ char get(uint32_t i) const { return p[i]; } __device__ void set(uint32_t i, char c) { p[i] = c; } } es{ t1 };
		refactor_es(es, k, [&](uint32_t i) -> uint32_t { return rf[i]; }, [&](uint32_t i) -> uint32_t { return ebuf[i]; });
	}
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#endif
	// A flank whose read side is much longer than what is left of the reference read ends (right flank) or starts (left flank:
	// the script of the reversed sequences is written reversed) in ONE run of inserted letters as long as the excess — the rows
	// the sweeps leave out (wv::sat_rows).  Both passes are the identity on such a run: pass 1 breaks at every insertion, and a
	// pass-2 region without a match is rewritten as itself.  A pass-2 region may reach into the run as far as the run's first
	// (last) letter repeats, so the passes look at the rest of the script plus that stretch — not at 10^5 letters.
	uint32_t ra = 0, rb = k;
	if (k >= 256)
	{
		auto is_ins = [](char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; };
		uint32_t b0 = k;                                                        // start of the maximal suffix of inserted letters
		while (b0 > 0)
		{
			const uint32_t lo = b0 >= 64 ? b0 - 64 : 0, x = lo + lane;
			const uint64_t non = __ballot(x < b0 && !is_ins(dst[x]));
			if (non) { b0 = lo + (63 - (uint32_t)__builtin_clzll(non)) + 1; break; }
			b0 = lo;
		}
		if (b0 < k)
		{	// ... plus the stretch over which its first letter repeats
			const char c0 = dst[b0];
			rb = k;
			for (uint32_t x0 = b0; x0 < k; x0 += 64)
			{
				const uint32_t x = x0 + lane;
				const uint64_t dif = __ballot(x < k && dst[x] != c0);
				if (dif) { rb = x0 + (uint32_t)__builtin_ctzll(dif); break; }
			}
		}
		uint32_t a0 = 0;                                                        // length of the maximal prefix of inserted letters
		for (; a0 < rb; )
		{
			const uint32_t x = a0 + lane;
			const uint64_t non = __ballot(x < rb && !is_ins(dst[x]));
			if (non) { a0 += (uint32_t)__builtin_ctzll(non); break; }
			a0 = a0 + 64 < rb ? a0 + 64 : rb;
		}
		if (a0 >= rb) ra = rb;                                                  // nothing but insertions: both passes are the identity
		else if (a0 > 0)
		{	// ... minus the stretch over which its last letter repeats
			const char c1 = dst[a0 - 1];
			ra = 0;
			for (uint32_t hi = a0; hi > 0; )
			{
				const uint32_t lo = hi >= 64 ? hi - 64 : 0, x = lo + lane;
				const uint64_t dif = __ballot(x < hi && dst[x] != c1);
				if (dif) { ra = lo + (63 - (uint32_t)__builtin_clzll(dif)) + 1; break; }
				hi = lo;
			}
		}
	}
	// (the skipped prefix consumes no reference symbol and `ra` read symbols)
	if (rb > ra && !wv::wave_refactor(pool, dst + ra, rb - ra, rf, ebuf + ra)) return false;
#ifdef CL_DEBUG_REFACTOR
	if (lane == 0)
	{
		uint32_t bad = k;
		for (uint32_t i = 0; i < k; ++i) if (dst[i] != t1[i]) { bad = i; break; }
		if (bad < k)
		{
			const uint32_t lo = bad > 70 ? bad - 70 : 0;
			printf("refactor mismatch k=%u at %u (lo %u)\n in : %.100s\n exp: %.100s\n got: %.100s\n", k, bad, lo, t0 + lo, t1 + lo, dst + lo);
		}
	}
#endif
	g.es_len = k; g.d_before = d_before;
	pool.lap(5);
	return true;
}
__global__ __launch_bounds__(64) void k_align_wave(const uint32_t* __restrict__ list, uint32_t n_list, GapRec* __restrict__ gaps, char* __restrict__ es_pool, ArenaV A, ArenaV R,
                                                  uint8_t* __restrict__ scratch, uint64_t per_wave, unsigned int* __restrict__ next, uint32_t* __restrict__ redo, unsigned int* __restrict__ n_redo, uint32_t dbg_stage, uint32_t* hbt, unsigned long long* prof)
{
	__builtin_amdgcn_s_setprio(3);                                          // a launch of this kernel lasts as long as its slowest chain: its waves go first on their SIMDs
	wv::WavePool pool{ scratch + (uint64_t)blockIdx.x * per_wave, per_wave, 0, false, hbt ? hbt + blockIdx.x : nullptr };
	pool.prof = prof; if (prof) pool.t_last = wall_clock64();
	pool.beat(1);
	const uint32_t lane = threadIdx.x;
	for (;;)
	{	// the list is ascending in size: the waves draw from its end, largest first, whenever they are free (a wave that got one
		// of the few huge gaps takes no further share of the rest)
		uint32_t slot = 0;
		if (lane == 0) slot = atomicAdd(next, 1u);
		slot = wv::bcast_first(slot);
		if (slot >= n_list) break;
		pool.beat(2);
		const uint32_t gi = list[n_list - 1 - slot];
		pool.top = 0; pool.overflow = false;
		GapRec g = gaps[gi];
		pool.beat(3);
		pool.lap(7);
		const uint64_t t_gap = prof ? wall_clock64() : 0;
		if (prof) for (int i = 0; i < 8; ++i) pool.gp[i] = 0;
		const bool ok_gap = align_wave_gap(pool, g, A, R, es_pool + g.es_off, dbg_stage);
		if (prof && lane == 0)
		{	// (diagnostic: the slowest gap, its rank, shape and phases)
			const unsigned long long mine = ((unsigned long long)(wall_clock64() - t_gap) << 24) | (slot & 0xffffffu);
			if (atomicMax(prof + 6, mine) < mine) { for (int i = 0; i < 6; ++i) prof[8 + i] = pool.gp[i]; prof[14] = ((unsigned long long)g.ne << 32) | g.use; prof[15] = g.kind; }
		}
		if (!ok_gap)
		{
			if (lane == 0) redo[atomicAdd(n_redo, 1u)] = gi;
			continue;
		}
		pool.beat(8);
		if (lane == 0) { gaps[gi].es_len = g.es_len; gaps[gi].d_before = g.d_before; }
	}
	pool.beat(9);
}

// giant gaps: MANY waves per gap (align_giant.hpp): phases of tile jobs over all giants of the level — score sweeps, then Hirschberg
// level by level —, then the leaves' tracebacks, then operations -> canonical script per gap.  What fails (capacities, heap, a
// hand-over that never came) goes to `redo` and the wave-per-gap kernel.
__global__ __launch_bounds__(256) void k_giant_stage(gt::View V, const uint32_t* __restrict__ list, const GapRec* __restrict__ gaps, ArenaV A, ArenaV R)
{
	__shared__ gt::Giant sg;
	const uint32_t gidx = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const GapRec g = gaps[list[V.n_giants - 1 - gidx]];                       // ascending by work: largest first
	const bool left = g.left != 0;
	if (w == 0)
	{
		uint8_t* rbuf = gt::galloc(V, g.use + 64ull); uint8_t* ebuf = gt::galloc(V, g.ne + 64ull);
		uint8_t* r2 = gt::galloc(V, g.use + 64ull); uint8_t* e2 = gt::galloc(V, g.ne + 64ull);
		uint8_t* opsbuf = gt::galloc(V, (uint64_t)g.use + g.ne + 64); uint8_t* sparse = gt::galloc(V, (uint64_t)g.use + g.ne + 64);
		if (lane == 0)
		{
			gt::Giant G; memset(&G, 0, sizeof(G));
			G.gi = list[V.n_giants - 1 - gidx]; G.kind = g.kind; G.left = left ? 1u : 0u;
			G.fail = (rbuf && ebuf && r2 && e2 && opsbuf && sparse) ? 0u : 12u;
			G.rbuf = rbuf; G.ebuf = ebuf; G.r2 = r2; G.e2 = e2; G.opsbuf = opsbuf; G.sparse = sparse;
			if (g.kind == GK_INNER) { G.Q = rbuf; G.n = g.nr; G.T = ebuf; G.m = g.ne; G.rows_ref = 1; G.shw = 0; }
			else if (g.kind == GK_FLANK_TINY) { G.Q = r2; G.n = g.use; G.T = e2; G.m = g.ne; G.rows_ref = 1; G.shw = 0; G.ref_end_nw = g.use - 1; }
			else { G.Q = e2; G.n = g.ne; G.T = r2; G.m = g.use; G.rows_ref = 0; G.shw = 1; }
			sg = G;
		}
	}
	__syncthreads();
	if (sg.fail) { if (tid == 0) V.giants[gidx] = sg; return; }
	{
		const uint32_t ref_id = g.ref_rev & 0x7fffffffu; const bool rev = g.ref_rev >> 31;
		const uint64_t rwb = R.word_off[ref_id], ewb = A.word_off[g.read]; const uint32_t rlen = R.lens[ref_id];
		const uint32_t lo = left ? g.nr - g.use : 0;
		for (uint32_t i = tid; i < g.use; i += 256) { const uint8_t v = (uint8_t)ref_sym(R, rwb, rlen, rev, g.cur_ref + lo + i); sg.rbuf[i] = v; sg.r2[left ? g.use - 1 - i : i] = v; }
		for (uint32_t i = tid; i < g.ne; i += 256) { const uint8_t v = (uint8_t)arena_base_at(A, ewb, g.enc_start + i); sg.ebuf[i] = v; sg.e2[left ? g.ne - 1 - i : i] = v; }
		for (uint64_t x = tid; x < (uint64_t)g.use + g.ne; x += 256) sg.sparse[x] = 0xff;
	}
	__syncthreads();
	if (w != 0) return;
	if (lane == 0) V.giants[gidx] = sg;
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	// the score (and, for a flank, where the alignment ends in the reference): a sweep of phase 0 — or closed form when the rows saturate
	const uint32_t n = sg.n, mc = sg.m;
	const uint32_t ne = wv::sat_rows(sg.Q, 1, n, sg.T, 1, mc);
	if (ne < n)
	{
		const uint32_t mp = mc, ref_end = sg.shw ? mc - 1 : sg.ref_end_nw;        // D[n][j] = n - j falls strictly: the first minimum is the last column
		if (lane == 0) { V.giants[gidx].mp = mp; V.giants[gidx].ref_end = ref_end; }
		gt::emit_sub(V, gidx, 0, n, 0, mp, n - mc, 1);
		return;
	}
	const uint32_t tiles = ((ne + 63) / 64 + 63) / 64, chunks = (mc + 63) / 64;
	unsigned long long* hand = tiles > 1 ? (unsigned long long*)gt::galloc(V, (uint64_t)(tiles - 1) * chunks * 16) : nullptr;
	const uint32_t f0 = gt::take(&V.ctl->n_flags, tiles - 1);
	const uint32_t ni = gt::take(&V.ctl->n_nodes[0], 1), si = gt::take(&V.ctl->n_sweeps[0], 1), ji = gt::take(&V.ctl->n_jobs[0], tiles);
	if ((tiles > 1 && !hand) || f0 + tiles - 1 > gt::FLAG_CAP || ni >= gt::NODE_CAP || si >= 2 * gt::NODE_CAP || ji + tiles > gt::JOB_CAP)
	{
		for (uint32_t x = lane; x < tiles; x += 64) if (ji + x < gt::JOB_CAP) V.jobs[0][ji + x] = gt::Job{ 0xffffffffu, 0 };
		gt::giant_fail(V, gidx, 13);
		return;
	}
	if (lane == 0)
	{
		V.nodes[0][ni] = gt::Node{ gidx, 0, n, 0, mc, 0, 0, 0, 0, 0, nullptr, nullptr, tiles, si };
		V.sweeps[0][si] = gt::Sweep{ sg.Q, sg.T, 1, 1, n, ne, mc, sg.shw, nullptr, hand, V.flags + f0, tiles, ni, 0, 0, 0, 0 };
	}
	for (uint32_t x = lane; x < tiles; x += 64) V.jobs[0][ji + x] = gt::Job{ si, x };
}
__global__ __launch_bounds__(64) void k_giant_level(gt::View V, uint32_t ph)
{
	__builtin_amdgcn_s_setprio(3);
	const uint32_t lane = threadIdx.x;
	const uint32_t n_jobs = V.ctl->n_jobs[ph] < gt::JOB_CAP ? V.ctl->n_jobs[ph] : gt::JOB_CAP;
	for (;;)
	{
		const uint32_t ticket = gt::take(&V.ctl->ticket[ph], 1);
		if (ticket >= n_jobs) break;
		const gt::Job jb = V.jobs[ph & 1][ticket];
		if (jb.sweep == 0xffffffffu) continue;
		gt::Sweep* S = &V.sweeps[ph & 1][jb.sweep];
		gt::Node* N = &V.nodes[ph & 1][S->node];
		const uint32_t giant = N->giant;
		if (!gt::giant_tile(S, jb.tile)) gt::giant_fail(V, giant, 14);
		// the node's last tile splits it: what the other tiles wrote (their parts of the last columns, the sweep's result) is released
		// before the count-down and acquired after it
		__threadfence();
		uint32_t left = 1;
		if (lane == 0) left = atomicSub(&N->pending, 1u) - 1;
		left = wv::bcast_first(left);
		if (left) continue;
		__threadfence();
		if (V.giants[giant].fail) continue;
		if (ph == 0)
		{
			const gt::Giant& G = V.giants[giant];
			const uint32_t score = wv::bcast_first(S->score), best = wv::bcast_first(S->best); const int32_t end = (int32_t)wv::bcast_first((uint32_t)S->end);
			const uint32_t mp = G.shw ? (uint32_t)(end + 1) : G.m, ref_end = G.shw ? (uint32_t)end : G.ref_end_nw;
			if (lane == 0) { V.giants[giant].mp = mp; V.giants[giant].ref_end = ref_end; }
			gt::emit_sub(V, giant, 0, G.n, 0, mp, G.shw ? best : score, 1);
		}
		else
		{
			const gt::Node nd = *N;
			uint32_t ls = 0, rs = 0;
			const int64_t found = wv::hirschberg_split(nd.left, nd.neL, nd.L, nd.right, nd.neR, nd.R, nd.n, nd.best, ls, rs);
			if (found < 0) { gt::giant_fail(V, giant, 15); continue; }
			gt::emit_sub(V, giant, nd.qo, (uint32_t)found, nd.to, nd.L, ls, ph + 1);
			gt::emit_sub(V, giant, nd.qo + (uint32_t)found, nd.n - (uint32_t)found, nd.to + nd.L, nd.R, rs, ph + 1);
		}
	}
}
__global__ __launch_bounds__(64) void k_giant_leaves(gt::View V, uint8_t* __restrict__ scratch, uint64_t per_wave)
{
	__builtin_amdgcn_s_setprio(3);
	wv::WavePool pool{ scratch + (uint64_t)blockIdx.x * per_wave, per_wave, 0, false, nullptr };
	const uint32_t lane = threadIdx.x;
	const uint32_t n_leaves = V.ctl->n_leaves < gt::LEAF_CAP ? V.ctl->n_leaves : gt::LEAF_CAP;
	for (;;)
	{
		const uint32_t ticket = gt::take(&V.ctl->leaf_ticket, 1);
		if (ticket >= n_leaves) break;
		const gt::Leaf lf = V.leaves[ticket];
		const gt::Giant& G = V.giants[lf.giant];
		if (G.fail) continue;
		uint8_t* dst = G.sparse + lf.qo + lf.to;                                // sub-problems tile the alignment in this order; n x m symbols give at most n + m operations
		if (lf.n == 0 || lf.m == 0)
		{
			const uint8_t op = lf.n == 0 ? 2 : 1;
			for (uint64_t x = lane; x < (uint64_t)lf.n + lf.m; x += 64) dst[x] = op;
			continue;
		}
		pool.top = 0; pool.overflow = false;
		wv::Ops o{ dst, 0 };
		wv::wave_traceback(pool, G.Q + lf.qo, lf.n, G.T + lf.to, lf.m, o);
		if (pool.overflow) gt::giant_fail(V, lf.giant, 16);
	}
}
__global__ __launch_bounds__(64) void k_giant_finish(gt::View V, GapRec* __restrict__ gaps, char* __restrict__ es_pool, uint8_t* __restrict__ scratch, uint64_t per_wave,
                                                    unsigned int* __restrict__ next, uint32_t* __restrict__ redo, unsigned int* __restrict__ n_redo)
{
	__builtin_amdgcn_s_setprio(3);
	wv::WavePool pool{ scratch + (uint64_t)blockIdx.x * per_wave, per_wave, 0, false, nullptr };
	const uint32_t lane = threadIdx.x;
	for (;;)
	{
		const uint32_t gidx = gt::take(next, 1);
		if (gidx >= V.n_giants) break;
		const gt::Giant G = V.giants[gidx];
		bool done = false;
		GapRec g = gaps[G.gi];
		if (!G.fail)
		{	// the sparse buffer -> the operations, in order (a chunk of 64 at a time, the offset carried along)
			const uint64_t span = (uint64_t)G.n + G.mp;
			uint64_t off = 0;
			for (uint64_t x0 = 0; x0 < span; x0 += 64)
			{
				const uint64_t x = x0 + lane;
				const uint8_t v = x < span ? G.sparse[x] : (uint8_t)0xff;
				const uint64_t bal = __ballot(v != 0xff);
				if (v != 0xff) G.opsbuf[off + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = v;
				off += (uint32_t)__popcll(bal);
			}
			__builtin_amdgcn_s_waitcnt(0);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			WaveGap W; W.rbuf = G.rbuf; W.ebuf = G.ebuf; W.r2 = G.r2; W.e2 = G.e2; W.opsbuf = G.opsbuf; W.Q = G.Q; W.T = G.T; W.n = G.n; W.m = G.m;
			W.rows_ref = G.rows_ref != 0; W.shw = G.shw != 0; W.left = G.left != 0;
			wv::Ops ops{ G.opsbuf, off };
			pool.top = 0; pool.overflow = false;
			g.es_len = 0; g.d_before = 0;
			done = wave_gap_finish(pool, g, W, ops, G.ref_end, es_pool + g.es_off, 0);
		}
		if (lane == 0)
		{
			if (done) { gaps[G.gi].es_len = g.es_len; gaps[G.gi].d_before = g.d_before; }
			else redo[atomicAdd(n_redo, 1u)] = G.gi;
		}
	}
}

// gaps of up to 16 row blocks (rows <= 1024, rows + columns <= 2048): FOUR per wave, every phase of a gap in its 16-lane row (align_rows.hpp;
// round 5): sequences, operations and script in LDS, the banded history in the wave's pool.  What does not fit (pool, LDS, a distance beyond the
// band) goes to `redo` (k_align_wave takes it).  Rounds 2-4 ran only the sweep four gaps at a time (k_align_quad_hist: 495 us per quad against
// 233 here; a form without history in HBM — checkpoints, windows recomputed into LDS — was 1.25 x slower still: numbers in DESIGN.md).
// PROF (COLORD_HIP_QUAD_PROFILE): 100-MHz clocks per phase and the traceback's window statistics, summed over the waves into prof[0..10]
template<bool PROF>
__global__ __launch_bounds__(64) void k_align_quad_rows(const uint32_t* __restrict__ list, uint32_t n_list, GapRec* __restrict__ gaps, char* __restrict__ es_pool, ArenaV A, ArenaV R,
                                                  uint8_t* __restrict__ scratch, uint64_t per_wave, unsigned int* __restrict__ next, uint32_t* __restrict__ redo, unsigned int* __restrict__ n_redo, unsigned long long* __restrict__ prof)
{
	uint64_t t_last = PROF ? wall_clock64() : 0;
	auto lap = [&](uint32_t phase) { if (PROF) { const uint64_t now = wall_clock64(); if (threadIdx.x == 0) atomicAdd(prof + phase, (unsigned long long)(now - t_last)); t_last = now; } };
	qr::WalkStats wstat;
	__shared__ __attribute__((aligned(16))) uint8_t s_rows[4 * qr::ROW_BYTES];
	const uint32_t lane = threadIdx.x, gq = lane >> 4, bl = lane & 15;
	ulonglong2* const pool = (ulonglong2*)(scratch + (uint64_t)blockIdx.x * per_wave);
	const uint64_t pool_pairs = per_wave / 16;
	const uint32_t n_quads = (n_list + 3) / 4;
	uint8_t* const mem = s_rows + gq * qr::ROW_BYTES;
	for (;;)
	{
		uint32_t slot = 0;
		if (lane == 0) slot = atomicAdd(next, 1u);
		slot = wv::bcast_first(slot);
		if (slot >= n_quads) break;
		const uint32_t hi = n_list - slot * 4, cnt = hi < 4 ? hi : 4;            // gaps list[hi - 1], list[hi - 2], ... (ascending list, taken from its end)
		const bool have = gq < cnt;
		const uint32_t gi = have ? list[hi - 1 - gq] : 0u;
		GapRec g; memset(&g, 0, sizeof(g));
		if (have) g = gaps[gi];
		qr::Row r;
		r.seq = (uint32_t*)mem; r.ops = mem + qr::SEQ_WORDS * 4; r.es = r.ops + qr::SEQ_MAX;
		r.use = g.use; r.ne = g.ne; r.e_off = (g.use + 15) / 16;
		r.left = g.left != 0; r.rev_seq = r.left && g.kind != GK_INNER;
		if (g.kind == GK_INNER) { r.n = g.nr; r.m = g.ne; r.rows_ref = true; r.shw = false; }
		else if (g.kind == GK_FLANK_TINY) { r.n = g.use; r.m = g.ne; r.rows_ref = true; r.shw = false; }
		else { r.n = g.ne; r.m = g.use; r.rows_ref = false; r.shw = true; }
		const uint32_t nb = (r.n + 63) / 64;
		const uint64_t pairs = (uint64_t)qr::hist_stride(r.m) * nb;
		// what the sweep keeps of its history: the blocks within `band` rows of a column (align_rows.hpp); a distance beyond it: not of this kernel
		const uint32_t band = (r.shw ? r.n : (r.n > r.m ? r.n : r.m)) / 4 + 16;
		bool ok = have && r.n && r.m && r.n <= 1024 && (g.kind != GK_INNER || g.nr == g.use) && g.use + g.ne <= qr::SEQ_MAX;
		// the rows' shares of the pool (a row that does not fit takes none)
		uint64_t off = 0;
		{
			const uint64_t mine = ok ? pairs : 0;
			const uint64_t p0 = wv::bcast(mine, 0u), p1 = wv::bcast(mine, 16u), p2 = wv::bcast(mine, 32u);
			off = gq == 0 ? 0 : gq == 1 ? p0 : gq == 2 ? p0 + p1 : p0 + p1 + p2;
			if (off + mine > pool_pairs) ok = false;
		}
		if (!ok) { r.n = 0; r.m = 0; }
		ulonglong2* const hist = pool + off;
		lap(5);
		qr::row_stage(r, ok, g, A, R);
		qr::lds_fence();
		lap(0);
		const wv::Sweep sw = qr::row_sweep(r, hist, band);
		__builtin_amdgcn_s_waitcnt(0);
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		if (ok && (r.shw ? sw.best : sw.score) > band) { ok = false; r.n = 0; r.m = 0; }
		lap(1);
		uint32_t wi, wj, wk;
		qr::row_walk(r, hist, r.shw ? (uint32_t)(sw.end + 1) : r.m, wi, wj, wk, PROF ? &wstat : nullptr);
		qr::lds_fence();
		lap(2);
		const uint32_t pre = wi + wj, K = pre + wk;
		qr::row_convert(r, pre, (uint8_t)(wi ? 1 : 2), wk);
		qr::lds_fence();
		lap(3);
		// canonical indel placement: pass 1 along the reference symbols the script consumes, pass 2 along the read's
		const uint32_t ref_end = r.shw ? (uint32_t)sw.end : g.kind == GK_FLANK_TINY ? g.use - 1 : 0u;
		uint32_t d_before = 0, seq_off = 0;
		if (r.left && g.kind != GK_INNER)
		{
			const uint32_t ref_offset = (g.nr - 1) - ref_end;                    // uint32 wrap for end = -1, as in the reference
			seq_off = ref_offset - (g.nr - g.use);
			d_before = ref_offset;
		}
		qr::row_refactor_pass(r, K, seq_off, 1);
		qr::row_refactor_pass(r, K, 0, 2);
		lap(4);
		if (ok)
		{
			uint32_t* dst = (uint32_t*)(es_pool + g.es_off); const uint32_t* src = (const uint32_t*)r.es;
			for (uint32_t w = bl; w * 4 < K; w += 16) dst[w] = src[w];
		}
		if (have && bl == 0)
		{
			if (ok) { gaps[gi].es_len = K; gaps[gi].d_before = d_before; }
			else redo[atomicAdd(n_redo, 1u)] = gi;
		}
		qr::lds_fence();
		if (PROF) { const uint32_t smax = qr::max4(r.n ? r.m + nb - 1 : 0); if (lane == 0) { atomicAdd(prof + 9, 1ull); atomicAdd(prof + 10, (unsigned long long)smax); } }
	}
	if (PROF && lane == 0) { atomicAdd(prof + 6, (unsigned long long)wstat.iters); atomicAdd(prof + 7, (unsigned long long)wstat.miss); atomicAdd(prof + 8, (unsigned long long)wstat.hit); }
}

// the rest: one lane per gap, lane pool in HBM; gaps whose lane ran out of pool are redone with larger pools
__global__ __launch_bounds__(64) void k_align_large(const uint32_t* __restrict__ list, uint32_t n_list, GapRec* __restrict__ gaps, char* __restrict__ es_pool, ArenaV A, ArenaV R,
                                                   uint8_t* __restrict__ scratch, uint64_t per_lane, unsigned int* __restrict__ next, uint32_t* __restrict__ redo, unsigned int* __restrict__ n_redo)
{
	LanePool pool{ scratch + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * per_lane, per_lane, 0, false, 0 };
	for (;;)
	{
		const uint32_t slot = atomicAdd(next, 1u);
		if (slot >= n_list) break;
		const uint32_t gi = list[n_list - 1 - slot];                            // the list is ascending by size: largest first
		pool.top = 0; pool.overflow = false; pool.why = 0;
		GapRec g = gaps[gi];
		if (!align_large_gap(pool, g, A, R, es_pool + g.es_off)) { redo[atomicAdd(n_redo, 1u)] = gi; continue; }
		gaps[gi].es_len = g.es_len; gaps[gi].d_before = g.d_before;
	}
}

// short gaps (the estimator decides them later): one lane per gap; long gaps are listed for k_gap_stats_long
__global__ void k_gap_stats(LevelV L, ArenaV A, EncCfg cfg, const uint32_t* __restrict__ pend_idx, uint32_t* __restrict__ spawn_flag, uint32_t* __restrict__ long_list)
{
	const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
	if (gi >= L.n_gaps) return;
	if (L.gaps[gi].ne >= cfg.min_part_alt) { long_list[gi - pend_idx[gi]] = gi; return; }      // pend_idx = number of short gaps before gi
	spawn_flag[gi] = gap_finish(L, gi, A, cfg, pend_idx[gi]) ? 1u : 0u;
}
// run summaries of the scripts for the count pass of the tuple emission: one lane per gap
__global__ void k_gap_sums(LevelV L, GapSum* __restrict__ sums, uint32_t* __restrict__ long_ids, unsigned int* __restrict__ n_long)
{
	const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
	if (gi >= L.n_gaps) return;
	const uint32_t k = L.gaps[gi].es_len;
	if (k && k <= SUM_LIMIT) sums[gi] = gap_summary(L.es + L.gaps[gi].es_off, k);
	else if (k) long_ids[atomicAdd(n_long, 1u)] = gi;                          // (k_gap_sums_long)
}
// ... and of the long scripts by a wave each (the wave-per-read emission works from summaries alone; a lane would take a
// millisecond for the 10^5 symbols of a long flank and hold its launch that long)
__global__ __launch_bounds__(256) void k_gap_sums_long(LevelV L, GapSum* __restrict__ sums, const uint32_t* __restrict__ long_ids, const unsigned int* __restrict__ n_long)
{
	const uint32_t n = *n_long, n_waves = gridDim.x * 4;
	for (uint32_t idx = blockIdx.x * 4 + (threadIdx.x >> 6); idx < n; idx += n_waves)
	{
		const uint32_t gi = long_ids[idx];
		const RunEl a = run_of_script_wave(L.es + L.gaps[gi].es_off, L.gaps[gi].es_len);
		if ((threadIdx.x & 63) == 0) sums[gi] = GapSum{ a.hl, a.tl, a.mb, a.mt, a.hs | (a.ts << 8) | (a.single ? 1u << 16 : 0u) };
	}
}
// long gaps: the static entropy test (EncodeWithEditScript, encoder.cpp:1315-1327; CEntropy, utils.h:706-752) by one wave
__global__ __launch_bounds__(256) void k_gap_stats_long(LevelV L, ArenaV A, EncCfg cfg, const uint32_t* __restrict__ long_list, uint32_t n_long, uint32_t* __restrict__ spawn_flag)
{
	const uint32_t wi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
	if (wi >= n_long) return;
	const uint32_t gi = long_list[wi];
	GapRec g = L.gaps[gi];
	char* es = L.es + g.es_off;
	const uint64_t ewb = A.word_off[g.read];
	if (g.kind == GK_TRIVIAL)
	{	// get_edit_dist_on_seq_empty (edit_script.h:250-267)
		if (g.nr == 0) { for (uint32_t i = lane; i < g.ne; i += 64) es[i] = base_letter(arena_base_at(A, ewb, g.enc_start + i)); g.es_len = g.ne; }
		else g.d_before = g.nr;
		__builtin_amdgcn_s_waitcnt(0);
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	}
	// leading deletions (GetEditScriptEntropyInput, encoder.cpp:1299-1311)
	uint32_t nd = 0;
	for (uint32_t i0 = 0; i0 < g.es_len; i0 += 64)
	{
		const uint32_t i = i0 + lane;
		const uint64_t notd = __ballot(i < g.es_len && es[i] != 'D');
		if (notd) { nd = i0 + (uint32_t)__builtin_ctzll(notd); break; }
		nd = i0 + 64 < g.es_len ? i0 + 64 : g.es_len;
	}
	uint32_t h[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, hd[4] = { 0, 0, 0, 0 };
	// the script eight symbols per lane and load, the read's bases a packed word (32 bases) per lane and load, counted by popcounts: a byte
	// and a base per lane were es_len / 64 + ne / 64 load latencies in a row for every gap
	for (uint32_t i = lane * 8; i < g.es_len; i += 512)
	{
		uint64_t w = 0; uint32_t nb = 8;
		if (i + 8 <= g.es_len) __builtin_memcpy(&w, es + i, 8);
		else { nb = g.es_len - i; for (uint32_t y = 0; y < nb; ++y) w |= (uint64_t)(uint8_t)es[i + y] << (8 * y); }
		for (uint32_t y = 0; y < nb; ++y)
		{
			const uint32_t c = es_class((char)((w >> (8 * y)) & 0xffu));
#pragma unroll
			for (int k = 0; k < 9; ++k) h[k] += c == (uint32_t)k;
		}
	}
	if (g.ne)
	{
		const uint32_t s = g.enc_start, last = g.enc_start + g.ne - 1, w0 = s >> 5, w1 = last >> 5;
		for (uint32_t wi2 = w0 + lane; wi2 <= w1; wi2 += 64)
		{
			const uint64_t w = A.packed[ewb + wi2];
			const uint32_t ja = wi2 == w0 ? (s & 31) : 0u, jb = wi2 == w1 ? (last & 31) : 31u;    // fields (bases) ja .. jb of the word count; base j sits at bits 63 - 2 j, 62 - 2 j
			const uint64_t from = ja == 0 ? ~0ull : ((1ull << (64 - 2 * ja)) - 1), upto = ~((1ull << (62 - 2 * jb)) - 1);
			const uint64_t v = from & upto & 0x5555555555555555ull, lo = w & 0x5555555555555555ull, hi = (w >> 1) & 0x5555555555555555ull;
			const uint32_t c3 = (uint32_t)__popcll(lo & hi & v), c2 = (uint32_t)__popcll(hi & ~lo & v), c1 = (uint32_t)__popcll(lo & ~hi & v);
			hd[3] += c3; hd[2] += c2; hd[1] += c1; hd[0] += (uint32_t)__popcll(v) - c1 - c2 - c3;
		}
	}
#pragma unroll
	for (int k = 0; k < 9; ++k) for (int o = 32; o; o >>= 1) h[k] += __shfl_xor(h[k], o);
#pragma unroll
	for (int k = 0; k < 4; ++k) for (int o = 32; o; o >>= 1) hd[k] += __shfl_xor(hd[k], o);
	uint32_t n = g.es_len, extra = g.d_before;
	if (nd + g.d_before >= 10) { h[2] -= nd; n -= nd; extra = 0; }                 // the script is scored without its leading deletions
	h[2] += extra;
	// the two entropies (entropy_hist, encode_core.hpp): their thirteen p log2 p terms one per lane in ONE evaluation of the logarithm (lanes
	// 0 .. 8 the script's classes, 16 .. 19 the bases) instead of thirteen in a row on every lane; the sums run over the terms in the
	// sequential order, so the doubles are the sequential ones
	double sum_h = 0, sum_d = 0;
#pragma unroll
	for (int i = 0; i < 9; ++i) sum_h += h[i];
#pragma unroll
	for (int i = 0; i < 4; ++i) sum_d += hd[i];
	const double rec_h = 1.0 / sum_h, rec_d = 1.0 / sum_d;
	uint32_t mine = 0;
#pragma unroll
	for (int i = 0; i < 9; ++i) if (lane == (uint32_t)i) mine = h[i];
#pragma unroll
	for (int i = 0; i < 4; ++i) if (lane == 16u + (uint32_t)i) mine = hd[i];
	double term = 0;
	if (mine) { const double p = (double)mine * (lane < 16 ? rec_h : rec_d); term = glibc_log2::log2(p) * p; }
	double e_h = 0, e_d = 0;
#pragma unroll
	for (int i = 0; i < 9; ++i) { const double t = __shfl(term, i, 64); if (h[i]) e_h += t; }
#pragma unroll
	for (int i = 0; i < 4; ++i) { const double t = __shfl(term, 16 + i, 64); if (hd[i]) e_d += t; }
	const bool accept = -e_h * (double)(n + extra) * cfg.cost_mult < -e_d * (double)g.ne;
	g.state = accept ? GS_ES : GS_REJECTED;
	if (lane == 0) { L.gaps[gi] = g; spawn_flag[gi] = accept ? 0u : 1u; }
}
// rejected long gaps: continue in a child frame when an alternative candidate still has anchors there, else literal
__global__ void k_spawn_mark(LevelV L, const uint32_t* __restrict__ data, EncCfg cfg, uint32_t* __restrict__ flag, uint32_t* __restrict__ ncand)
{
	const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
	if (gi >= L.n_gaps) return;
	ncand[gi] = 0;
	if (!flag[gi]) return;
	const GapRec g = L.gaps[gi];
	const FrameRec F = L.frames[g.frame];
	CandEnt out[16];
	const bool alt = spawn_cands(F, L.cands + F.cand_base, data, g, cfg, out);
	flag[gi] = alt ? 1u : 0u;
	ncand[gi] = alt ? F.n_cands : 0u;
	if (!alt) L.gaps[gi].state = GS_LITERAL;
}
__global__ void k_spawn_fill(LevelV L, const uint32_t* __restrict__ data, EncCfg cfg, const uint32_t* __restrict__ flag_scan, const uint64_t* __restrict__ cand_base,
                             FrameRec* __restrict__ frames_next, CandEnt* __restrict__ cands_next)
{
	const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
	if (gi >= L.n_gaps) return;
	if (cand_base[gi + 1] == cand_base[gi]) return;
	spawn_child(L, gi, data, cfg, flag_scan[gi], cand_base[gi], frames_next, cands_next);
}

// ---- estimator and emission -------------------------------------------------------------------------------------------
__global__ void k_read_base_counts(ArenaV A, const uint8_t* __restrict__ has_n, uint32_t n, uint32_t* __restrict__ counts /* 4 per read */)
{	// one wave per read
	const uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
	if (r >= n) return;
	uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
	if (!has_n[r])
	{
		const uint32_t len = A.lens[r]; const uint64_t wb = A.word_off[r];
		for (uint32_t w = lane; w * 32 < len; w += 64)
		{
			const uint64_t x = A.packed[wb + w]; const uint32_t cnt = len - w * 32 < 32 ? len - w * 32 : 32;
			const uint64_t lo = x & 0x5555555555555555ull, hi = (x >> 1) & 0x5555555555555555ull;
			const uint64_t valid = cnt == 32 ? 0x5555555555555555ull : (0x5555555555555555ull & ~((1ull << (64 - 2 * cnt)) - 1));
			c0 += __popcll(~hi & ~lo & valid); c1 += __popcll(~hi & lo & valid); c2 += __popcll(hi & ~lo & valid); c3 += __popcll(hi & lo & valid);
		}
	}
	for (int o = 32; o; o >>= 1) { c0 += __shfl_xor(c0, o); c1 += __shfl_xor(c1, o); c2 += __shfl_xor(c2, o); c3 += __shfl_xor(c3, o); }
	if (lane == 0) { counts[4 * r] = c0; counts[4 * r + 1] = c1; counts[4 * r + 2] = c2; counts[4 * r + 3] = c3; }
}
// events of the estimator: the pending gaps of every read in encoding order (depth-first over its frames)
__global__ void k_pend_count(TreeV T, uint32_t n_reads, uint32_t* __restrict__ cnt)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	const uint32_t f0 = T.frame_of_read[r];
	cnt[r] = f0 == 0xffffffffu ? 0u : pend_walk(T, f0, nullptr);
}
__global__ void k_pend_list(TreeV T, uint32_t n_reads, const uint64_t* __restrict__ off, uint32_t* __restrict__ events)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	const uint32_t f0 = T.frame_of_read[r];
	if (f0 != 0xffffffffu && off[r + 1] > off[r]) pend_walk(T, f0, events + off[r]);
}
// CEntropyEstimator over one reader pack by one wave (utils.h:877-1130): the events are inherently sequential, the work
// inside one event is not — lanes 0..11 hold the 12 edit-script counters and take the logarithms in parallel, lanes
// 12..15 precompute the decision logarithms of both outcomes; sums are then accumulated in the reference's order.
__global__ __launch_bounds__(64) void k_estimator(TreeV T, const uint32_t* __restrict__ pack_bounds, uint32_t n_packs, const uint32_t* __restrict__ lens, const uint8_t* __restrict__ has_n,
                                                 const uint32_t* __restrict__ base_counts, const uint64_t* __restrict__ ev_off, const uint32_t* __restrict__ events)
{
	__builtin_amdgcn_s_setprio(3);                                          // a launch of this kernel lasts as long as its slowest chain: its waves go first on their SIMDs
	__shared__ PendRec recs[64];
	__shared__ uint32_t ev_id[64];
	const uint32_t pk = blockIdx.x, lane = threadIdx.x;
	if (pk >= n_packs) return;
	const uint32_t MX = 1u << 20;
	uint32_t es_l = lane < 12 ? 1u : 0u, es_sum = 12;
	uint32_t dna0 = 1, dna1 = 1, dna2 = 1, dna3 = 1, dna_sum = 4, dec0 = 1, dec1 = 1, dec_sum = 2;
	double dl0, dl1, dl2, dl3, dcl0, dcl1;
	{ const double r4 = 1.0 / 4, r2 = 1.0 / 2; dl0 = dl1 = dl2 = dl3 = -glibc_log2::log2(1.0 * r4); dcl0 = dcl1 = -glibc_log2::log2(1.0 * r2); }
	for (uint32_t r = pack_bounds[pk]; r < pack_bounds[pk + 1]; ++r)
	{
		if (has_n[r]) continue;                                                  // reads with N never reach the estimator (encoder.cpp:1629-1633)
		{	// LogRead (utils.h:946-955)
			dna0 += base_counts[4 * r]; dna1 += base_counts[4 * r + 1]; dna2 += base_counts[4 * r + 2]; dna3 += base_counts[4 * r + 3];
			dna_sum += lens[r];
			while (dna_sum > MX) { dna0 = (dna0 + 1) / 2; dna1 = (dna1 + 1) / 2; dna2 = (dna2 + 1) / 2; dna3 = (dna3 + 1) / 2; dna_sum = dna0 + dna1 + dna2 + dna3; }
			const double rec = 1.0 / dna_sum;
			const uint32_t x = lane == 0 ? dna0 : lane == 1 ? dna1 : lane == 2 ? dna2 : dna3;
			const double lg = x ? -glibc_log2::log2((double)x * rec) : 0.0;
			dl0 = __shfl(lg, 0); dl1 = __shfl(lg, 1); dl2 = __shfl(lg, 2); dl3 = __shfl(lg, 3);
		}
		for (uint64_t e0 = ev_off[r]; e0 < ev_off[r + 1]; e0 += 64)
		{
			const uint32_t cnt = (uint32_t)(ev_off[r + 1] - e0 < 64 ? ev_off[r + 1] - e0 : 64);
			__syncthreads();
			if (lane < cnt)
			{
				const uint32_t ev = events[e0 + lane];
				ev_id[lane] = ev;
				recs[lane] = T.lv[ev >> 28].pend[ev & 0x0fffffffu];
			}
			__syncthreads();
			for (uint32_t k = 0; k < cnt; ++k)
			{	// EncodeWithEditScript (utils.h:1060-1130)
				const PendRec& p = recs[k];
				const uint32_t rd_l = lane < 12 ? p.rd[lane] : 0u;
				uint32_t rd_sum = rd_l;
				for (int o = 8; o; o >>= 1) rd_sum += __shfl_xor(rd_sum, o);              // lanes 0..15 (12..15 hold 0)
				rd_sum = __shfl(rd_sum, 0);
				const uint32_t loc_l = es_l + rd_l, loc_sum = es_sum + rd_sum;
				const bool dec_resc = dec_sum + 1 > MX;
				uint32_t x = loc_l; double rec = 1.0 / loc_sum;
				if (lane >= 12 && lane < 16)
				{	// decision logarithms after this event: lanes 12, 13 if the edit script wins, 14, 15 if the literal wins
					x = lane == 12 ? dec0 + 1 : lane == 13 ? dec1 : lane == 14 ? dec0 : dec1 + 1;
					rec = 1.0 / (dec_sum + 1);
				}
				const double lg = x ? -glibc_log2::log2((double)x * rec) : 0.0;
				const double term = (double)rd_l * lg;
				double es_cost = dcl0;
#pragma unroll
				for (int i = 0; i < 12; ++i) es_cost += __shfl(term, i);
				es_cost = lens_add(es_cost, p.lens);
				double plain_cost = dcl1;
				plain_cost += p.pl[0] * dl0; plain_cost += p.pl[1] * dl1; plain_cost += p.pl[2] * dl2; plain_cost += p.pl[3] * dl3;
				plain_cost += bitlen32(p.ref_len) + 1;
				const bool choose_plain = plain_cost < es_cost;
				if (choose_plain) ++dec1;
				else
				{
					++dec0; es_l = loc_l; es_sum = loc_sum;
					while (es_sum > MX)
					{
						es_l = (es_l + 1) / 2;
						uint32_t sm = lane < 12 ? es_l : 0u;
						for (int o = 8; o; o >>= 1) sm += __shfl_xor(sm, o);
						es_sum = __shfl(sm, 0);
					}
				}
				++dec_sum;
				if (!dec_resc) { dcl0 = __shfl(lg, choose_plain ? 14 : 12); dcl1 = __shfl(lg, choose_plain ? 15 : 13); }
				else
				{
					while (dec_sum > MX) { dec0 = (dec0 + 1) / 2; dec1 = (dec1 + 1) / 2; dec_sum = dec0 + dec1; }
					const double rc = 1.0 / dec_sum;
					dcl0 = dec0 ? -glibc_log2::log2((double)dec0 * rc) : 0.0; dcl1 = dec1 ? -glibc_log2::log2((double)dec1 * rc) : 0.0;
				}
				if (lane == 0) { const uint32_t ev = ev_id[k]; T.lv[ev >> 28].dec[ev & 0x0fffffffu] = choose_plain ? 0 : 1; }
			}
		}
	}
}
// Tuple emission: one wave per read over the run-length monoid of its fragments (emit_wave.hpp; k_emit_count_wave sizes and tuple counts,
// k_emit_write_wave the bytes), k_emit_plain for reads stored plain (one wave per read, one lane per base).  Rounds 1-2 walked a read's frame
// tree with one lane (emit_read, encode_core.hpp: 12.6 s per pass against 2.1; it stays as the sequential statement the host debugging
// build replays, tests/tools/encode_host.hip).
__global__ __launch_bounds__(256) void k_emit_count_wave(ArenaV A, TreeV T, uint32_t n_reads, const uint32_t* __restrict__ data, uint32_t* __restrict__ sizes, uint32_t* __restrict__ ntuples)
{
	const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n_reads) return;
	if (T.frame_of_read[r] == 0xffffffffu) { if ((threadIdx.x & 63) == 0) { sizes[r] = A.lens[r] + 1; ntuples[r] = A.lens[r] + 1; } return; }   // a start tuple and one tuple per base
	emit_read_wave<false>(A, T, r, data, sizes, ntuples, nullptr, nullptr);
}
__global__ __launch_bounds__(256) void k_emit_write_wave(ArenaV A, TreeV T, uint32_t n_reads, const uint32_t* __restrict__ data, const uint64_t* __restrict__ es_off, uint8_t* __restrict__ out)
{
	const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n_reads || T.frame_of_read[r] == 0xffffffffu) return;
	emit_read_wave<true>(A, T, r, data, nullptr, nullptr, es_off, out);
}
__global__ __launch_bounds__(256) void k_emit_plain(ArenaV A, const uint32_t* __restrict__ inv, const uint8_t* __restrict__ has_n, const uint32_t* __restrict__ frame_of_read, uint32_t n_reads,
                                                   const uint64_t* __restrict__ es_off, uint8_t* __restrict__ out)
{	// AddPlainRead / AddPlainReadWithN (encoder.cpp:663-681): a start tuple, then one tuple per base
	const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (r >= n_reads || frame_of_read[r] != 0xffffffffu) return;
	const uint32_t len = A.lens[r]; const uint64_t wb = A.word_off[r];
	uint8_t* o = out + es_off[r];
	if (lane == 0) o[0] = (uint8_t)((has_n[r] ? 11 : 9) << 4);
	for (uint32_t i = lane; i < len; i += 64)
	{
		const uint64_t pw = A.packed[wb + (i >> 5)]; const uint32_t iw = inv[wb + (i >> 5)];
		const bool isn = (iw >> (31 - (i & 31))) & 1u;
		o[1 + i] = (uint8_t)((8u << 4) + (isn ? 4u : (uint32_t)(pw >> (62 - 2 * (i & 31))) & 3u));
	}
}

struct LevelBufs {
	DevBuf<FrameRec> frames; DevBuf<CandEnt> cands; DevBuf<GapRec> gaps; DevBuf<char> es; DevBuf<PendRec> pend; DevBuf<uint8_t> dec; DevBuf<GapSum> sums;
	uint32_t n_frames = 0, n_gaps = 0; uint64_t n_cands = 0;
	LevelV view() { return LevelV{ frames.p, cands.p, gaps.p, es.p, pend.p, dec.p, n_frames, n_gaps, sums.p }; }
};
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
	const EncCfg cfg{ c, anchor_len, min_part_alt, max_rec, cost_mult };
	const uint8_t* has_n = reads->has_n.p;
	hipStream_t st = ctx->stream;

	std::vector<std::unique_ptr<LevelBufs>> levels;
	DevBuf<uint32_t> frame_of_read; DEV_ALLOC(ctx, frame_of_read, nr);
	DevBuf<uint32_t> err; DEV_ALLOC(ctx, err, 1);
	HIP_TRY(ctx, hipMemsetAsync(err.p, 0, 4, st));
	{	// level 0
		auto L = std::make_unique<LevelBufs>();
		DevBuf<uint32_t> flag, ncand; DEV_ALLOC(ctx, flag, (uint64_t)nr + 1); DEV_ALLOC(ctx, ncand, (uint64_t)nr + 1);
		DevBuf<uint64_t> cbase; DEV_ALLOC(ctx, cbase, (uint64_t)nr + 1);
		LAUNCH(ctx, k_read_flags, grid_for(nr, 256), 256, AV.n_cands, has_n, nr, flag.p, ncand.p);
		uint64_t nf = 0;
		CL_TRY(dev_exclusive_scan_u64(ctx, ncand.p, cbase.p, nr, &L->n_cands));
		CL_TRY(dev_exclusive_scan_u32(ctx, flag.p, nr, &nf));
		L->n_frames = (uint32_t)nf;
		DEV_ALLOC(ctx, L->frames, nf + 1); DEV_ALLOC(ctx, L->cands, L->n_cands + 1);
		LAUNCH(ctx, k_frames_level0, grid_for(nr, 256), 256, A, AV, c, has_n, nr, (const uint32_t*)flag.p, (const uint64_t*)cbase.p, L->frames.p, L->cands.p, frame_of_read.p);
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipStreamSynchronize(st));
		levels.push_back(std::move(L));
	}
	const uint32_t n_cu = (uint32_t)std::max(1, ctx->n_cu);                  // (cl_ctx_create: the device's multiProcessorCount)
	for (uint32_t lv = 0; lv < levels.size(); ++lv)
	{
		LevelBufs& L = *levels[lv];
		if (!L.n_frames) break;
		// gaps of the level
		DevBuf<uint32_t> first; DEV_ALLOC(ctx, first, (uint64_t)L.n_frames + 1);
		LAUNCH(ctx, k_frame_gap_counts, grid_for(L.n_frames, 256), 256, (const FrameRec*)L.frames.p, (const CandEnt*)L.cands.p, L.n_frames, first.p);
		uint64_t ng = 0;
		CL_TRY(dev_exclusive_scan_u32(ctx, first.p, L.n_frames, &ng));
		if (ng >= (1ull << 32)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_encode_reads: more than 2^32 gaps in one call");
		L.n_gaps = (uint32_t)ng;
		LAUNCH(ctx, k_frame_first_gaps, grid_for(L.n_frames, 256), 256, L.frames.p, (const uint32_t*)first.p, L.n_frames, L.n_gaps);
		DEV_ALLOC(ctx, L.gaps, ng + 1);
		DevBuf<uint32_t> capw, pflag, keys, ids; DEV_ALLOC(ctx, capw, ng + 1); DEV_ALLOC(ctx, pflag, ng + 1); DEV_ALLOC(ctx, keys, ng + 1); DEV_ALLOC(ctx, ids, ng + 1);
		LevelV V = L.view();
		LAUNCH(ctx, k_gap_geometry, grid_for(ng, 256), 256, V, R, AV.data, min_part_alt, (const uint32_t*)first.p, capw.p, pflag.p, keys.p, ids.p, err.p);
		DevBuf<uint64_t> es_off; DEV_ALLOC(ctx, es_off, ng + 1);
		uint64_t es_total = 0, n_pend = 0;
		CL_TRY(dev_exclusive_scan_u64(ctx, capw.p, es_off.p, ng, &es_total));
		CL_TRY(dev_exclusive_scan_u32(ctx, pflag.p, ng, &n_pend));
		LAUNCH(ctx, k_gap_offsets, grid_for(ng, 256), 256, L.gaps.p, (const uint64_t*)es_off.p, L.n_gaps);
		DEV_ALLOC(ctx, L.es, es_total + 16); DEV_ALLOC(ctx, L.pend, n_pend + 1); DEV_ALLOC(ctx, L.dec, n_pend + 1);
		V = L.view();
		// size classes
		CL_TRY(dev_sort_keys32_pairs(ctx, keys.p, ids.p, ng, 0, KEY_BITS));
		DevBuf<uint32_t> bounds; DEV_ALLOC(ctx, bounds, N_CLASSES + 3);
		HIP_TRY(ctx, hipMemsetAsync(bounds.p + N_CLASSES + 1, 0, 8, st));
		LAUNCH(ctx, k_class_bounds, grid_for(ng + 1, 256), 256, (const uint32_t*)keys.p, (const uint32_t*)ids.p, (const uint32_t*)capw.p, L.n_gaps, bounds.p);
		uint32_t hb[N_CLASSES + 3];
		unsigned long long h_cb[2 * N_CLASSES] = { 0 };                           // bytes, then DP cells, per class (timing runs only)
		HIP_TRY(ctx, hipMemcpyAsync(hb, bounds.p, 4 * (N_CLASSES + 3), hipMemcpyDeviceToHost, st));
		if (ctx->timing)
		{
			DevBuf<unsigned long long> cb; DEV_ALLOC(ctx, cb, 2 * N_CLASSES);
			HIP_TRY(ctx, hipMemsetAsync(cb.p, 0, 16 * N_CLASSES, st));
			LAUNCH(ctx, k_class_bytes, grid_for(ng, 256), 256, (const uint32_t*)keys.p, (const uint32_t*)ids.p, (const uint32_t*)capw.p, (const GapRec*)L.gaps.p, L.n_gaps, cb.p);
			HIP_TRY(ctx, hipMemcpyAsync(h_cb, cb.p, 16 * N_CLASSES, hipMemcpyDeviceToHost, st));
		}
		HIP_TRY(ctx, hipStreamSynchronize(st));
		// small gaps: on the side stream, next to the large ones on the main stream (one wave per SIMD with its state in
		// LDS here, five waves per SIMD on a bump pool in HBM there: they share the machine well)
		HIP_TRY(ctx, cl_side_stream(ctx, ctx->side));
		struct SideJoin { hipStream_t s; ~SideJoin() { (void)hipStreamSynchronize(s); } };
		DevBuf<uint64_t> hist;
		SideJoin side_join{ ctx->side };                                         // (after hist in destruction order: joins first)
		{
			uint32_t max_blocks = 1;
			static const uint32_t small_per_cu = getenv("COLORD_HIP_SMALL_PER_CU") ? (uint32_t)std::max(1, atoi(getenv("COLORD_HIP_SMALL_PER_CU"))) : 4u;   // (tuning knob: blocks of 15 - 25 KB of LDS each)
			for (int nb = 1; nb <= 4; ++nb) max_blocks = std::max(max_blocks, std::min<uint32_t>(grid_for(hb[nb + 1] - hb[nb], 64), n_cu * small_per_cu));
			DEV_ALLOC(ctx, hist, (uint64_t)max_blocks * 256 * 4 * 2 * 64);
			LaunchOn on(ctx, ctx->side);                                          // (launches + timing events on the side stream)
			hipError_t le = hipSuccess;
			for (int nb = 1; nb <= 4; ++nb)
			{
				const uint32_t n_list = hb[nb + 1] - hb[nb];
				if (!n_list) continue;
				const uint32_t blocks = std::min<uint32_t>(grid_for(n_list, 64), n_cu * small_per_cu);
				const uint32_t lds = (12 * nb + 48) * 64 * 4;                       // LdsMem<nb>: QW + TW + EW words per lane
				const double bytes = 1.25 * (double)h_cb[nb];
				ctx->next_cells = (double)h_cb[N_CLASSES + nb];
				const uint32_t* list = ids.p + hb[nb];
				switch (nb)
				{
				case 1: LAUNCHB_SHM(ctx, bytes, (k_align_small<1>), blocks, 64, lds, list, n_list, L.gaps.p, L.es.p, A, R, hist.p); break;
				case 2: LAUNCHB_SHM(ctx, bytes, (k_align_small<2>), blocks, 64, lds, list, n_list, L.gaps.p, L.es.p, A, R, hist.p); break;
				case 3: LAUNCHB_SHM(ctx, bytes, (k_align_small<3>), blocks, 64, lds, list, n_list, L.gaps.p, L.es.p, A, R, hist.p); break;
				default: LAUNCHB_SHM(ctx, bytes, (k_align_small<4>), blocks, 64, lds, list, n_list, L.gaps.p, L.es.p, A, R, hist.p); break;
				}
				if (le == hipSuccess) le = hipGetLastError();
			}
			HIP_TRY(ctx, le);
		}
		// gaps of up to 16 row blocks: four per wave, on a third stream — a throughput kernel next to the wave-per-gap kernel
		// below, whose launches last as long as their single slowest gap (measured: launch 61.5 ms, slowest gap 61.4 ms) and
		// leave the machine idle meanwhile.  The few gaps whose buffers do not fit a wave's pool come back in quad_redo.
		DevBuf<uint32_t> quad_redo; uint32_t n_quad_redo = 0;
		DevBuf<uint8_t> quad_scratch; DevBuf<unsigned int> qc;
		struct Side2Join { hipStream_t s = nullptr; ~Side2Join() { if (s) (void)hipStreamSynchronize(s); } } quad_join;   // (after the buffers in destruction order)
		if (hb[6] > hb[5])
		{
			const uint32_t n_list = hb[6] - hb[5];
			// 2.25 MB a wave: four banded histories of at most 512 KB; 2048 waves (4096 were no faster beside the other streams)
			const uint64_t per_wave = 9ull << 18;
			static const uint32_t quad_waves = getenv("COLORD_HIP_QUAD_WAVES") ? (uint32_t)std::max(64, atoi(getenv("COLORD_HIP_QUAD_WAVES"))) : 2048u;   // (tuning knob: waves of 18.5 KB of LDS each)
			const uint32_t waves = std::min<uint32_t>((n_list + 3) / 4, quad_waves);
			DEV_ALLOC(ctx, quad_scratch, per_wave * waves);
			DEV_ALLOC(ctx, qc, 2);
			DEV_ALLOC(ctx, quad_redo, (uint64_t)n_list + 1);
			HIP_TRY(ctx, cl_side_stream(ctx, ctx->side2));
			quad_join.s = ctx->side2;
			HIP_TRY(ctx, hipMemsetAsync(qc.p, 0, 8, ctx->side2));
			LaunchOn on(ctx, ctx->side2);                                         // (launch + timing events on the third stream)
			ctx->next_cells = (double)h_cb[N_CLASSES + 5];
			static const bool quad_prof = getenv("COLORD_HIP_QUAD_PROFILE") != nullptr;
			if (quad_prof)
			{	// diagnostic: the phases of the row kernel (waits for the launch)
				DevBuf<unsigned long long> qp; DEV_ALLOC(ctx, qp, 16);
				HIP_TRY(ctx, hipMemsetAsync(qp.p, 0, 128, ctx->side2));
				LAUNCHB(ctx, 1.25 * (double)h_cb[5], k_align_quad_rows<true>, waves, 64, (const uint32_t*)ids.p + hb[5], n_list, L.gaps.p, L.es.p, A, R, quad_scratch.p, per_wave, qc.p, quad_redo.p, qc.p + 1, qp.p);
				unsigned long long hp[16];
				HIP_TRY(ctx, hipStreamSynchronize(ctx->side2));
				HIP_TRY(ctx, hipMemcpy(hp, qp.p, 128, hipMemcpyDeviceToHost));
				const double q = (double)std::max<unsigned long long>(hp[9], 1);
				fprintf(stderr, "[quad rows, level %u] %u gaps in %llu quads on %u waves; us per quad: stage %.1f sweep %.1f walk %.1f convert %.1f refactor %.1f fetch+write %.1f; sweep steps per quad %.0f; walk iterations per quad %.0f, of which waited for new windows %.1f\n",
					lv, n_list, hp[9], waves, hp[0] / q / 100.0, hp[1] / q / 100.0, hp[2] / q / 100.0, hp[3] / q / 100.0, hp[4] / q / 100.0, hp[5] / q / 100.0, hp[10] / q, hp[6] / q, hp[7] / q);
			}
			else LAUNCHB(ctx, 1.25 * (double)h_cb[5], k_align_quad_rows<false>, waves, 64, (const uint32_t*)ids.p + hb[5], n_list, L.gaps.p, L.es.p, A, R, quad_scratch.p, per_wave, qc.p, quad_redo.p, qc.p + 1, (unsigned long long*)nullptr);
			HIP_TRY(ctx, hipGetLastError());
		}
		// giant gaps: many waves each (align_giant.hpp), on a stream of their own from the start of the level, next to everything else
		DevBuf<uint32_t> team_redo; uint32_t n_team_redo = 0;
		DevBuf<uint8_t> giant_mem, giant_heap, giant_scratch; DevBuf<unsigned int> tc;
		Side2Join team_join;
		if (hb[8] > hb[7] && !getenv("COLORD_HIP_NO_TEAM_ALIGN"))
		{
			const uint32_t n_list = hb[8] - hb[7];
			const uint64_t sum_cap = (uint64_t)hb[N_CLASSES + 1] * 256, max_cap = hb[N_CLASSES + 2];
			// sequences (4 copies), operations, sparse operations: 6 x (reference + read symbols) per gap; the last columns of a Hirschberg
			// level: 8 bytes per row and level, the hand-over pairs 2 bits per column and tile
			const uint64_t heap_bytes = std::min<uint64_t>(6ull << 30, (64ull << 20) + 100 * sum_cap + (uint64_t)n_list * 4096);
			uint32_t phases = 2; for (uint64_t c = max_cap; c > 8 && phases < gt::MAX_PHASES; c >>= 1) ++phases;     // the columns halve per level; below ~16 everything is a leaf
			auto al = [](uint64_t x) { return (x + 255) & ~255ull; };
			const uint64_t o_ctl = 0, o_flags = al(sizeof(gt::Ctl)), o_nodes = o_flags + al(4ull * gt::FLAG_CAP), o_sweeps = o_nodes + 2 * al(sizeof(gt::Node) * gt::NODE_CAP),
				o_jobs = o_sweeps + 2 * al(sizeof(gt::Sweep) * 2 * gt::NODE_CAP), o_leaves = o_jobs + 2 * al(sizeof(gt::Job) * gt::JOB_CAP), o_giants = o_leaves + al(sizeof(gt::Leaf) * gt::LEAF_CAP),
				o_end = o_giants + al(sizeof(gt::Giant) * n_list);
			DEV_ALLOC(ctx, giant_mem, o_end); DEV_ALLOC(ctx, giant_heap, heap_bytes);
			const uint32_t leaf_waves = 256; const uint64_t per_wave = 3ull << 20;     // a leaf's history is < 1 MiB twice over, + the reversed operations
			const uint32_t fin_waves = std::min<uint32_t>(n_list, 64); const uint64_t fin_per_wave = (per_wave * leaf_waves) / 64;   // 12 MB: the canonicalisation's 9 bytes per script symbol
			DEV_ALLOC(ctx, giant_scratch, per_wave * leaf_waves);
			DEV_ALLOC(ctx, tc, 2);
			DEV_ALLOC(ctx, team_redo, (uint64_t)n_list + 1);
			HIP_TRY(ctx, cl_side_stream(ctx, ctx->side3));
			team_join.s = ctx->side3;
			HIP_TRY(ctx, hipMemsetAsync(tc.p, 0, 8, ctx->side3));
			HIP_TRY(ctx, hipMemsetAsync(giant_mem.p, 0, o_nodes, ctx->side3));       // the control block and the progress words
			gt::View V; V.ctl = (gt::Ctl*)(giant_mem.p + o_ctl); V.heap = giant_heap.p; V.heap_bytes = heap_bytes; V.flags = (uint32_t*)(giant_mem.p + o_flags);
			for (int i = 0; i < 2; ++i) { V.nodes[i] = (gt::Node*)(giant_mem.p + o_nodes + i * al(sizeof(gt::Node) * gt::NODE_CAP)); V.sweeps[i] = (gt::Sweep*)(giant_mem.p + o_sweeps + i * al(sizeof(gt::Sweep) * 2 * gt::NODE_CAP));
				V.jobs[i] = (gt::Job*)(giant_mem.p + o_jobs + i * al(sizeof(gt::Job) * gt::JOB_CAP)); }
			V.leaves = (gt::Leaf*)(giant_mem.p + o_leaves); V.giants = (gt::Giant*)(giant_mem.p + o_giants); V.n_giants = n_list; V.n_phases = phases;
			LaunchOn on(ctx, ctx->side3);
			ctx->next_cells = (double)h_cb[N_CLASSES + 7];                         // (booked on the staging launch: the class as a whole is what the report adds up)
			LAUNCHB(ctx, 1.25 * (double)h_cb[7], k_giant_stage, n_list, 256, V, (const uint32_t*)ids.p + hb[7], (const GapRec*)L.gaps.p, A, R);
			for (uint32_t ph = 0; ph < phases; ++ph) LAUNCH(ctx, k_giant_level, 1024, 64, V, ph);
			LAUNCH(ctx, k_giant_leaves, leaf_waves, 64, V, giant_scratch.p, per_wave);
			LAUNCH(ctx, k_giant_finish, fin_waves, 64, V, L.gaps.p, L.es.p, giant_scratch.p, fin_per_wave, tc.p, team_redo.p, tc.p + 1);
			HIP_TRY(ctx, hipGetLastError());
		}
		// large gaps, in rounds of growing lane pools
		auto run_large = [&](const uint32_t* list, uint32_t n_list, double alg_bytes) -> cl_status
		{
			if (n_list && getenv("COLORD_HIP_GAP_DEBUG") && list == ids.p + hb[6])
			{	// diagnostic: shapes of the wave-class gaps of the level (the list is ascending in log2 of the work)
				HIP_TRY(ctx, hipStreamSynchronize(st));
				std::vector<uint32_t> h_ids(n_list);
				HIP_TRY(ctx, hipMemcpy(h_ids.data(), list, (uint64_t)n_list * 4, hipMemcpyDeviceToHost));
				std::vector<GapRec> all(L.n_gaps);
				HIP_TRY(ctx, hipMemcpy(all.data(), L.gaps.p, (uint64_t)L.n_gaps * sizeof(GapRec), hipMemcpyDeviceToHost));
				double work = 0, steps = 0; std::string txt; uint64_t hist[8] = { 0 }; double hwork[8] = { 0 };
				for (uint32_t i = 0; i < n_list; ++i)
				{
					const GapRec& t = all[h_ids[n_list - 1 - i]];
					const uint32_t rows = t.kind == GK_FLANK ? t.ne : t.use, cols = t.kind == GK_FLANK ? t.use : t.ne;
					const double w = (double)((rows + 63) / 64) * cols, st_ = (double)((rows + 4095) / 4096) * (cols + 63);
					work += w; steps += st_;
					const int b = rows <= 256 ? 0 : rows <= 1024 ? 1 : rows <= 4096 ? 2 : rows <= 16384 ? 3 : 4;
					hist[b]++; hwork[b] += st_;
					if (i < 8) txt += " " + std::string(t.kind == GK_FLANK ? "F" : "I") + std::to_string(rows) + "x" + std::to_string(cols);
				}
				fprintf(stderr, "[gaps] level %u: classes 1-4 %u, quad %u, wave %u; wave class: block-columns %.3g, sweep steps %.3g; by rows <=256 %llu (%.3g steps) <=1024 %llu (%.3g) <=4096 %llu (%.3g) <=16384 %llu (%.3g) more %llu (%.3g); largest:%s\n",
					lv, hb[5] - hb[1], hb[6] - hb[5], n_list, work, steps, (unsigned long long)hist[0], hwork[0], (unsigned long long)hist[1], hwork[1], (unsigned long long)hist[2], hwork[2],
					(unsigned long long)hist[3], hwork[3], (unsigned long long)hist[4], hwork[4], txt.c_str());
			}
			DevBuf<uint32_t> todo, redo; DEV_ALLOC(ctx, todo, (uint64_t)n_list + 1); DEV_ALLOC(ctx, redo, (uint64_t)n_list + 1);
			DevBuf<unsigned int> cnt; DEV_ALLOC(ctx, cnt, 2);
			// waves (k_align_wave) / lanes (k_align_large).  3 MB a wave: since the sweeps stop at the saturation row, sequences + operations
			// + history of nearly every gap fit (a read flank of 200 kb: 1.6 MB); the few others come back in the next round (x 8)
			uint64_t per_lane = 3ull << 20; uint32_t max_lanes = 2048;
			const bool use_wave = getenv("COLORD_HIP_NO_WAVE_ALIGN") == nullptr;
			DevBuf<unsigned long long> prof;
			if (getenv("COLORD_HIP_WAVE_PROFILE")) { DEV_ALLOC(ctx, prof, 16); HIP_TRY(ctx, hipMemsetAsync(prof.p, 0, 128, st)); }
			uint32_t* hbt_host = nullptr; uint32_t* hbt_dev = nullptr;
			if (use_wave && n_list && getenv("COLORD_HIP_WAVE_HEARTBEAT"))
			{	// debugging: host-visible progress words, dumped by a watchdog thread if the kernel is still running after a while
				HIP_TRY(ctx, hipHostMalloc((void**)&hbt_host, 4096 * 4, hipHostMallocMapped));
				memset(hbt_host, 0, 4096 * 4);
				HIP_TRY(ctx, hipHostGetDevicePointer((void**)&hbt_dev, hbt_host, 0));
				std::thread([hbt_host]() { for (int t = 0; t < 3; ++t) { std::this_thread::sleep_for(std::chrono::seconds(8)); fprintf(stderr, "[heartbeat]");
					for (int i = 0; i < 24; ++i) fprintf(stderr, " %u", hbt_host[i]); fprintf(stderr, "\n"); fflush(stderr); } }).detach();
			}
			for (int round = 0; n_list; ++round)
			{
				if (round == 5) return cl_fail(ctx, CL_E_NOMEM, "cl_encode_reads: lane pool exhausted (" + std::to_string(n_list) + " gaps left)");
				const uint32_t lanes = use_wave ? std::min<uint32_t>(n_list, max_lanes) : (uint32_t)std::min<uint64_t>(((uint64_t)n_list + 63) / 64 * 64, max_lanes);
				DevBuf<uint8_t> scratch; DEV_ALLOC(ctx, scratch, per_lane * lanes);
				HIP_TRY(ctx, hipMemsetAsync(cnt.p, 0, 8, st));
				if (use_wave)
				{
				const auto t_launch = std::chrono::steady_clock::now();
				LAUNCHB(ctx, round == 0 ? alg_bytes : 0.0, k_align_wave, lanes, 64, list, n_list, L.gaps.p, L.es.p, A, R, scratch.p, per_lane, cnt.p, redo.p, cnt.p + 1,
					(uint32_t)(getenv("COLORD_HIP_WAVE_DEBUG_STAGE") ? atoi(getenv("COLORD_HIP_WAVE_DEBUG_STAGE")) : 0), hbt_dev, prof.p);
				if (prof.p)
				{
					unsigned long long hp[16];
					HIP_TRY(ctx, hipStreamSynchronize(st));
					HIP_TRY(ctx, hipMemcpy(hp, prof.p, 128, hipMemcpyDeviceToHost));
					fprintf(stderr, "[slowest gap, level %u] kind %llu read symbols %llu reference symbols %llu: stage %.2f sweep %.2f path %.2f convert %.2f refactor %.2f ms\n", lv, hp[15], hp[14] >> 32, hp[14] & 0xffffffffull,
						hp[9] / 1e5, hp[10] / 1e5, hp[11] / 1e5, hp[12] / 1e5, hp[13] / 1e5);
					fprintf(stderr, "[wave phases, level %u, %u gaps, Mcycles of 100 MHz] alloc %llu stage %llu sweep %llu path %llu convert %llu refactor %llu fetch %llu; slowest gap %.2f ms (rank %llu from the largest); launch of %u waves %.1f ms\n", lv, n_list,
						hp[0] / 1000000, hp[1] / 1000000, hp[2] / 1000000, hp[3] / 1000000, hp[4] / 1000000, hp[5] / 1000000, hp[7] / 1000000, (double)(hp[6] >> 24) / 1e5, hp[6] & 0xffffffull, lanes, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_launch).count());
				}
				}
				else LAUNCH(ctx, k_align_large, lanes / 64, 64, list, n_list, L.gaps.p, L.es.p, A, R, scratch.p, per_lane, cnt.p, redo.p, cnt.p + 1);
				HIP_TRY(ctx, hipGetLastError());
				unsigned int hc[2];
				HIP_TRY(ctx, hipMemcpyAsync(hc, cnt.p, 8, hipMemcpyDeviceToHost, st));
				HIP_TRY(ctx, hipStreamSynchronize(st));
				n_list = hc[1];
				if (n_list)
				{
					HIP_TRY(ctx, hipMemcpyAsync(todo.p, redo.p, (uint64_t)n_list * 4, hipMemcpyDeviceToDevice, st));
					HIP_TRY(ctx, hipStreamSynchronize(st));
					list = todo.p;
				}
				per_lane *= 8; max_lanes = std::max<uint32_t>(max_lanes / 8, 64);
			}
			return CL_OK;
		};
		ctx->next_cells = (double)h_cb[N_CLASSES + 6];
		CL_TRY(run_large(ids.p + hb[6], hb[7] - hb[6], 1.25 * (double)h_cb[6]));
		if (quad_join.s)
		{
			unsigned int hc[2];
			HIP_TRY(ctx, hipMemcpyAsync(hc, qc.p, 8, hipMemcpyDeviceToHost, ctx->side2));
			HIP_TRY(ctx, hipStreamSynchronize(ctx->side2));
			quad_join.s = nullptr;
			n_quad_redo = hc[1];
		}
		if (n_quad_redo) CL_TRY(run_large(quad_redo.p, n_quad_redo, 0.0));
		if (team_join.s)
		{
			unsigned int hc[2];
			HIP_TRY(ctx, hipMemcpyAsync(hc, tc.p, 8, hipMemcpyDeviceToHost, ctx->side3));
			HIP_TRY(ctx, hipStreamSynchronize(ctx->side3));
			team_join.s = nullptr;
			n_team_redo = hc[1];
			if (getenv("COLORD_HIP_GAP_DEBUG")) fprintf(stderr, "[gaps] level %u: %u giant gaps by tile jobs (largest script %u symbols), %u back to the wave kernel\n", lv, hb[8] - hb[7], hb[N_CLASSES + 2], n_team_redo);
		}
		else if (hb[8] > hb[7]) CL_TRY(run_large(ids.p + hb[7], hb[8] - hb[7], 1.25 * (double)h_cb[7]));   // (COLORD_HIP_NO_TEAM_ALIGN)
		if (n_team_redo) CL_TRY(run_large(team_redo.p, n_team_redo, 0.0));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->side));                           // the small gaps are through
		if (getenv("COLORD_HIP_GAP_SHAPES"))
		{	// diagnostic: shapes and edit distances of the aligned gaps by class — what a band (edlib.cpp:192-212) would save, what fits LDS
			HIP_TRY(ctx, hipDeviceSynchronize());
			std::vector<GapRec> all(L.n_gaps); std::vector<char> hes(es_total + 16);
			HIP_TRY(ctx, hipMemcpy(all.data(), L.gaps.p, (uint64_t)L.n_gaps * sizeof(GapRec), hipMemcpyDeviceToHost));
			HIP_TRY(ctx, hipMemcpy(hes.data(), L.es.p, es_total, hipMemcpyDeviceToHost));
			struct Acc { uint64_t n = 0; double cells = 0, band = 0, band_ideal = 0; std::vector<uint32_t> nm, dpm; uint64_t ratio[11] = { 0 }; } acc[N_CLASSES + 1];
			for (const GapRec& g : all)
			{
				if (g.kind == GK_TRIVIAL) continue;
				const uint32_t rows = g.kind == GK_FLANK ? g.ne : g.use, cols = g.kind == GK_FLANK ? g.use : g.ne;      // (gap_class, restated for the host)
				const uint32_t cls = (rows <= 256 && cols <= 256) ? (rows + 63) / 64 : (rows <= QUAD_ROWS && rows + cols <= QUAD_SEQ && (uint64_t)((rows + 63) / 64) * (cols + 16) <= QUAD_CELLS) ? 5u
					: (rows > GIANT_ROWS && rows <= GIANT_MAX_ROWS && (uint64_t)((rows + 63) / 64) * cols >= GIANT_WORK && rows / 8 < cols) ? 7u : 6u;
				if (!cls || !rows || !cols) continue;
				uint32_t d = 0; for (uint32_t x = 0; x < g.es_len; ++x) d += hes[g.es_off + x] != 'M';
				Acc& a = acc[cls]; ++a.n;
				const double nb = (rows + 63) / 64; a.cells += nb * cols;
				// edlib's k-doubling (edlib.cpp:192-212: k = 64, 128, ... until the score fits) with Ukkonen's band of 2 k + |rows - cols| + 1 rows
				const double diff = rows > cols ? rows - cols : cols - rows;
				auto bandw = [&](double k) { return std::min(nb, std::ceil((2 * k + diff + 1) / 64.0) + 1); };
				double k = 64, cost = 0; while (k < d) { cost += bandw(k) * cols * 0.5; k *= 2; }   // (a failed round stops about half way)
				a.band += cost + bandw(k) * cols; a.band_ideal += bandw(d) * cols;
				a.nm.push_back(rows + cols); a.dpm.push_back((uint32_t)(1000.0 * d / std::max(rows, cols)));
				a.ratio[std::min<uint32_t>(10, (uint32_t)(20.0 * d / std::max(rows, cols)))]++;
			}
			for (uint32_t cls = 1; cls <= N_CLASSES; ++cls)
			{
				Acc& a = acc[cls]; if (!a.n) continue;
				std::sort(a.nm.begin(), a.nm.end()); std::sort(a.dpm.begin(), a.dpm.end());
				auto pc = [&](std::vector<uint32_t>& v, double p) { return v[std::min<size_t>(v.size() - 1, (size_t)(p * v.size()))]; };
				auto le = [&](uint32_t x) { return (double)(std::upper_bound(a.nm.begin(), a.nm.end(), x) - a.nm.begin()) / a.n; };
				fprintf(stderr, "[shapes] level %u class %u: %llu gaps, block-columns %.4g, k-doubling band %.4g (%.2f), band of the true distance %.4g (%.2f); rows+cols p50 %u p90 %u p99 %u p99.9 %u max %u; <=1536 %.4f <=2048 %.4f <=2730 %.4f <=4096 %.4f; d/len permille p10 %u p50 %u p90 %u p99 %u; by d/len in steps of 5%%:",
					lv, cls, (unsigned long long)a.n, a.cells, a.band, a.band / a.cells, a.band_ideal, a.band_ideal / a.cells, pc(a.nm, 0.5), pc(a.nm, 0.9), pc(a.nm, 0.99), pc(a.nm, 0.999), a.nm.back(),
					le(1536), le(2048), le(2730), le(4096), pc(a.dpm, 0.1), pc(a.dpm, 0.5), pc(a.dpm, 0.9), pc(a.dpm, 0.99));
				for (int i = 0; i < 11; ++i) fprintf(stderr, " %llu", (unsigned long long)a.ratio[i]);
				fprintf(stderr, "\n");
			}
		}
		// statistics / decisions, children
		DevBuf<uint32_t> sflag, sncand; DEV_ALLOC(ctx, sflag, ng + 1); DEV_ALLOC(ctx, sncand, ng + 1);
		const uint32_t n_long = (uint32_t)(ng - n_pend);
		DevBuf<uint32_t> long_list; DEV_ALLOC(ctx, long_list, (uint64_t)n_long + 1);
		LAUNCH(ctx, k_gap_stats, grid_for(ng, 64), 64, V, A, cfg, (const uint32_t*)pflag.p, sflag.p, long_list.p);
		if (n_long) LAUNCH(ctx, k_gap_stats_long, grid_for((uint64_t)n_long * 64, 256), 256, V, A, cfg, (const uint32_t*)long_list.p, n_long, sflag.p);
		DEV_ALLOC(ctx, L.sums, ng + 1);
		DevBuf<uint32_t> sum_long; DEV_ALLOC(ctx, sum_long, ng + 2);                // ids of the gaps with long scripts, their number in the last word
		HIP_TRY(ctx, hipMemsetAsync(sum_long.p + ng + 1, 0, 4, st));
		LAUNCH(ctx, k_gap_sums, grid_for(ng, 64), 64, V, L.sums.p, sum_long.p, (unsigned int*)(sum_long.p + ng + 1));
		LAUNCH(ctx, k_gap_sums_long, 1024, 256, V, L.sums.p, (const uint32_t*)sum_long.p, (const unsigned int*)(sum_long.p + ng + 1));
		LAUNCH(ctx, k_spawn_mark, grid_for(ng, 64), 64, V, AV.data, cfg, sflag.p, sncand.p);      // refuses beyond max_rec: those gaps become literals
		HIP_TRY(ctx, hipGetLastError());
		if (lv >= max_rec || lv + 1 >= 10) { HIP_TRY(ctx, hipStreamSynchronize(st)); break; }
		auto N = std::make_unique<LevelBufs>();
		DevBuf<uint64_t> cbase; DEV_ALLOC(ctx, cbase, ng + 1);
		uint64_t nf = 0;
		CL_TRY(dev_exclusive_scan_u64(ctx, sncand.p, cbase.p, ng, &N->n_cands));
		CL_TRY(dev_exclusive_scan_u32(ctx, sflag.p, ng, &nf));
		N->n_frames = (uint32_t)nf;
		if (!nf) break;
		DEV_ALLOC(ctx, N->frames, nf + 1); DEV_ALLOC(ctx, N->cands, N->n_cands + 1);
		LAUNCH(ctx, k_spawn_fill, grid_for(ng, 64), 64, V, AV.data, cfg, (const uint32_t*)sflag.p, (const uint64_t*)cbase.p, N->frames.p, N->cands.p);
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipStreamSynchronize(st));
		levels.push_back(std::move(N));
	}
	uint32_t herr = 0;
	HIP_TRY(ctx, hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipStreamSynchronize(st));
	if (herr) return cl_fail(ctx, CL_E_INVALID, "cl_encode_reads: inconsistent anchors");

	TreeV T; memset(&T, 0, sizeof(T));
	for (size_t i = 0; i < levels.size() && i < 10; ++i) T.lv[i] = levels[i]->view();
	T.frame_of_read = frame_of_read.p;
	// estimator (per reader pack), then tuples: sizes, offsets, bytes
	DevBuf<uint32_t> base_counts; DEV_ALLOC(ctx, base_counts, 4ull * nr);
	LAUNCHB(ctx, reads->total_bases / 4.0, k_read_base_counts, grid_for((uint64_t)nr * 64, 256), 256, A, has_n, nr, base_counts.p);
	DevBuf<uint32_t> d_pb; DEV_ALLOC(ctx, d_pb, (uint64_t)n_packs + 1);
	HIP_TRY(ctx, hipMemcpyAsync(d_pb.p, h_pack_bounds, ((uint64_t)n_packs + 1) * 4, hipMemcpyHostToDevice, st));
	DevBuf<uint32_t> ev_cnt; DEV_ALLOC(ctx, ev_cnt, (uint64_t)nr + 1);
	DevBuf<uint64_t> ev_off; DEV_ALLOC(ctx, ev_off, (uint64_t)nr + 1);
	LAUNCH(ctx, k_pend_count, grid_for(nr, 64), 64, T, nr, ev_cnt.p);
	uint64_t n_events = 0;
	CL_TRY(dev_exclusive_scan_u64(ctx, ev_cnt.p, ev_off.p, nr, &n_events));
	DevBuf<uint32_t> events; DEV_ALLOC(ctx, events, n_events + 1);
	LAUNCH(ctx, k_pend_list, grid_for(nr, 64), 64, T, nr, (const uint64_t*)ev_off.p, events.p);
	if (n_packs) LAUNCH(ctx, k_estimator, n_packs, 64, T, (const uint32_t*)d_pb.p, n_packs, (const uint32_t*)reads->lens.p, has_n, (const uint32_t*)base_counts.p, (const uint64_t*)ev_off.p, (const uint32_t*)events.p);
	DevBuf<uint32_t> sizes; DEV_ALLOC(ctx, sizes, nr);
	uint64_t total = 0;
	{	// one wave per read, the lanes over the fragments of a frame (emit_wave.hpp)
		LAUNCHB(ctx, reads->total_bases * 1.25, k_emit_count_wave, grid_for(nr, 4), 256, A, T, nr, AV.data, sizes.p, d_es_ntuples);
		HIP_TRY(ctx, hipGetLastError());
		CL_TRY(dev_exclusive_scan_u64(ctx, sizes.p, d_es_off, nr, &total));
		*n_out = total;
		if (total > cap || (total && !d_es)) return cl_fail(ctx, CL_E_CAPACITY, "cl_encode_reads: need " + std::to_string(total) + " bytes");
		LAUNCHB(ctx, reads->total_bases * 1.25 + (double)total, k_emit_write_wave, grid_for(nr, 4), 256, A, T, nr, AV.data, (const uint64_t*)d_es_off, d_es);
	}
	LAUNCH(ctx, k_emit_plain, grid_for(nr, 4), 256, A, (const uint32_t*)reads->inv.p, has_n, (const uint32_t*)frame_of_read.p, nr, (const uint64_t*)d_es_off, d_es);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(st));
	cl_timing_collect(ctx);
	return CL_OK;
}

// The decision logarithm on its own (calc_logs, utils.h:800-810), so that tests can pin it against the host's libm over
// the reachable (count, total) pairs: same translation unit, same flags, same log2 as k_estimator and gap_stats.
namespace {
__global__ void k_estimator_logs(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ tot, uint64_t n, double* __restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t x = cnt[i];
	const double rec = 1.0 / tot[i];
	out[i] = x ? -glibc_log2::log2((double)x * rec) : 0.0;
}
} // namespace
extern "C" cl_status cl_estimator_logs(cl_ctx* ctx, const uint32_t* d_count, const uint32_t* d_total, uint64_t n, double* d_out)
{
	if (!ctx || (n && (!d_count || !d_total || !d_out))) return cl_fail(ctx, CL_E_INVALID, "cl_estimator_logs: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (n) LAUNCH(ctx, k_estimator_logs, grid_for(n, 256), 256, d_count, d_total, n, d_out);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CL_OK;
}
