// qual.hip — quality stream coder (a13 + a15 + a16 framing) on the GPU, bit-identical to the reference's
// sequential adaptive coder.
//
// The reference codes one symbol per base through ONE adaptive model set that lives for the whole file
// (entr_qual.h:100-135).  Two facts make that parallel without changing a single output byte:
//   1. contexts depend only on INPUT data (previous quality values, neighbouring bases, edit-script
//      flags), never on coder state, so (context, symbol) of every position is computed independently;
//   2. a model is touched only by the symbols of its own context (basic_coder.h:116-137) and its update
//      is "+ADDER, halve-round-up when total >= MAX_TOTAL" (rc.h:233-244,347-358), so after a STABLE sort by
//      context each model is evolved by one wavefront over its own contiguous run, with 64 symbols
//      ranked per step by ballot + popcount prefixes.
// That yields for every coded symbol the exact triple (cumulative, frequency, total) the sequential coder
// would have used.  The interval arithmetic itself (sub_rc.h:83-100) is a dependent chain; it restarts
// per part (entr_qual.h:68-79), so one lane codes one part and all parts of a batch run concurrently.
#include "common.hpp"
#include <functional>
#include <deque>
#include <thread>
#include <chrono>
#include "objects.hpp"
#include "rc_dev.hpp"
#include <algorithm>

namespace {

enum { QM_ORIGINAL = 0, QM_QUINARY_AVG, QM_QUAD_AVG, QM_BINARY_AVG, QM_QUINARY_THR, QM_QUAD_THR, QM_BINARY_THR, QM_AVERAGE, QM_NONE };

struct QualCfg {
	int32_t mode, level;
	uint32_t bits_per_sym, n_ctx_sym, ctx_bits;   // previous-symbol history
	uint32_t base_bits;                            // neighbouring-base part of the context
	uint32_t n_sym, sym_bits;                      // alphabet of the per-base family
	uint32_t n_bins, navg;                         // *-avg: bins and coded bytes per read (2 per bin; avg: 2)
	uint32_t max_total, adder;
	uint32_t n_ctx;                                // dense context count of the per-base family
	uint32_t is_avg, is_thr;
	uint8_t map_fwd[96], quant[96];
};

} // namespace

// The model-independent half of cl_qual_encode for a batch of parts: per group of parts the interleaved triple layout, the
// (context, symbol) key and triple slot of every coded symbol, the stable sort by context and the context runs.  It depends on the
// input only (qualities, bases, flags), so the compressor makes it ahead, on a context and thread of its own, beside the model
// evolution and interval coding of the batch before (cl_qual_prepare_batch).
struct QualGroupPrep {
	uint32_t p0 = 0, p1 = 0, np = 0, ng = 0; uint64_t n_base = 0, n_syms = 0, n_byte = 0, trip_words = 0;
	std::vector<uint32_t> rank, plen_r;
	DevBuf<uint64_t> d_gbase; DevBuf<uint32_t> d_plen;
	DevBuf<uint32_t> key, sidx, bkey, bsidx, ss, se, bss, bse;
};
struct QualPrepared {
	const cl_reads* R = nullptr; const uint8_t* d_quals = nullptr; std::vector<uint32_t> part_bounds;
	std::vector<std::unique_ptr<QualGroupPrep>> groups;
};
struct cl_qual_coder {
	cl_ctx* ctx = nullptr;
	QualCfg cfg;
	DevBuf<QualCfg> d_cfg;
	DevBuf<uint32_t> state;       // per-base family: n_ctx * (n_sym + 1)  (counters..., total)
	DevBuf<uint32_t> bstate;      // byte family: 896 * 257
	uint64_t symbols_coded = 0;
	std::unique_ptr<QualPrepared> ahead;   // the next batch, prepared ahead
	std::deque<std::unique_ptr<struct QualEvolved>> evolved;   // the next batches, their models evolved and their interval coders running (cl_qual_evolve_ahead)
	std::function<cl_status()> before_tail;        // called by cl_qual_encode before it waits for the interval coders of its batch
	std::vector<hipStream_t> cstreams; uint32_t next_cstream = 0;
	~cl_qual_coder();
};
// a group whose interval coder is running (as PendingGroup of dna.hip)
struct QualPending {
	DevBuf<triple_t> trip; DevBuf<uint64_t> d_gbase, d_out_off, d_size, d_dst_off; DevBuf<uint32_t> d_plen; DevBuf<uint8_t> tmp;
	std::vector<uint32_t> rank; uint32_t p0 = 0, np = 0; uint64_t n_syms = 0;
	hipStream_t stream = nullptr;
	~QualPending() { if (stream) (void)hipStreamSynchronize(stream); }            // (nothing above is released while the coder runs)
};
struct QualEvolved {
	const cl_reads* R = nullptr; const uint8_t* d_quals = nullptr; std::vector<uint32_t> part_bounds;
	std::vector<std::unique_ptr<QualPending>> groups;
};
cl_qual_coder::~cl_qual_coder() { evolved.clear(); for (hipStream_t s : cstreams) (void)hipStreamDestroy(s); }

namespace {
constexpr uint32_t BYTE_CTX = 5 * 128 + 256;      // (bin, floor(prev avg)) and 0x100 + high byte (quality_coder_impl.cpp:821-834)

__device__ inline uint32_t arena_base(const uint64_t* __restrict__ packed, uint64_t wb, uint32_t p)
{
	return (uint32_t)(packed[wb + (p >> 5)] >> (62 - 2 * (p & 31))) & 3u;
}

// ---- Q1: (context, symbol) of every coded symbol, in stream order ---------------------------------
// one wave per read.  key32 = ctx << sym_bits | sym ; byte family: bkey = ctx << 8 | byte
__global__ __launch_bounds__(256) void k_qual_symbols(const QualCfg* __restrict__ cfgp, const uint64_t* __restrict__ packed,
                                                     const uint64_t* __restrict__ word_off, const uint8_t* __restrict__ quals,
                                                     const uint64_t* __restrict__ qoff, const uint8_t* __restrict__ flags,
                                                     uint32_t r0, uint32_t r1, uint64_t q0, TripLayoutDev lay,
                                                     uint32_t* __restrict__ key, uint32_t* __restrict__ sidx_out,
                                                     uint32_t* __restrict__ bkey, uint32_t* __restrict__ bsidx, uint32_t* __restrict__ bad)
{
	__shared__ QualCfg cfg;
	for (uint32_t i = threadIdx.x; i < sizeof(QualCfg) / 4; i += blockDim.x) ((uint32_t*)&cfg)[i] = ((const uint32_t*)cfgp)[i];
	__syncthreads();
	const uint32_t r = r0 + blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= r1) return;
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t qb = qoff[r]; const uint32_t len = (uint32_t)(qoff[r + 1] - qb);
	const uint64_t wb = word_off[r];
	const uint32_t navg = cfg.navg;
	// position of this read's first symbol in the batch's symbol stream, and of its first per-base key
	const bool per_base = !(cfg.mode == QM_AVERAGE || cfg.mode == QM_NONE);     // 'avg' codes two bytes per read and nothing per base
	const uint64_t s_read = (per_base ? (qb - q0) : 0) + (uint64_t)(r - r0) * navg;
	const uint64_t k_read = qb - q0;
	const uint32_t part = part_of_read(lay, r);
	// (every quality byte passes through here once: the range check — Phred+33 0..95, anything else would index past the 96-entry maps — rides
	// along; rounds 1-5 made a pass of its own over the qualities for it, k_qual_check: 1 byte per base of the 3.2 the whole path must move)
	bool out_of_range = false;

	if (navg)
	{	// per-read averages: integer sums are exact, so sum/cnt in double equals the reference's accumulation
		uint32_t sum[5] = { 0, 0, 0, 0, 0 }, cnt[5] = { 0, 0, 0, 0, 0 };
		for (uint32_t i = lane; i < len; i += 64)
		{
			uint32_t q = quals[qb + i] - 33u;
			out_of_range |= q > 95u;
			uint32_t b = cfg.mode == QM_AVERAGE ? 0u : cfg.map_fwd[q > 95u ? 0u : q];
#pragma unroll
			for (uint32_t t = 0; t < 5; ++t) if (b == t) { sum[t] += q; cnt[t] += 1; }
		}
#pragma unroll
		for (uint32_t t = 0; t < 5; ++t) { sum[t] = wave_sum(sum[t]); cnt[t] = wave_sum(cnt[t]); }
		if (lane == 0)
		{
			const uint64_t bo = (uint64_t)(r - r0) * navg;
			if (cfg.mode == QM_AVERAGE)
			{
				double avg = (double)sum[0] / (double)len;            // quality_coder_impl.cpp:438-450 (0/0 -> NaN -> cast 0 on x86: len==0 unsupported)
				uint32_t a = (uint32_t)(avg * 256), a1 = a >> 8, a2 = a & 0xff;
				bkey[bo] = (0u << 8) | a1; bsidx[bo] = trip_index(lay, part, s_read);
				bkey[bo + 1] = ((640u + a1) << 8) | a2; bsidx[bo + 1] = trip_index(lay, part, s_read + 1);
			}
			else
			{
				uint32_t ctx_p = 0;
				for (uint32_t t = 0; t < cfg.n_bins; ++t)
				{
					double avg = cnt[t] ? (double)sum[t] / (double)cnt[t] : 0.0;
					uint32_t a = (uint32_t)(avg * 256), a1 = a >> 8, a2 = a & 0xff;
					bkey[bo + 2 * t] = ((t * 128u + ctx_p) << 8) | a1; bsidx[bo + 2 * t] = trip_index(lay, part, s_read + 2 * t);
					bkey[bo + 2 * t + 1] = ((640u + a1) << 8) | a2; bsidx[bo + 2 * t + 1] = trip_index(lay, part, s_read + 2 * t + 1);
					ctx_p = (uint32_t)avg;
				}
			}
		}
	}
	if (cfg.mode == QM_AVERAGE || cfg.mode == QM_NONE) { if (__ballot(out_of_range) && lane == 0) atomicOr(bad, 1u); return; }

	const uint32_t sym_mask = (1u << cfg.bits_per_sym) - 1;
	for (uint32_t i = lane; i < len; i += 64)
	{
		uint32_t qv = quals[qb + i] - 33u;
		if (qv > 95u) { out_of_range = true; qv = 0; }
		uint32_t sym = cfg.map_fwd[qv];
		// history: context values of positions i-1 .. i-n (missing = all ones)
		uint32_t hist = 0;
		for (uint32_t t = 1; t <= cfg.n_ctx_sym; ++t)
		{
			uint32_t v = sym_mask;
			if (i >= t)
			{
				const uint32_t qp = quals[qb + i - t] - 33u;                          // (checked at its own position)
				uint32_t s = cfg.map_fwd[qp > 95u ? 0u : qp];
				v = cfg.mode == QM_ORIGINAL ? cfg.quant[s] : s;
			}
			hist |= (v & sym_mask) << ((t - 1) * cfg.bits_per_sym);
		}
		uint32_t b0 = arena_base(packed, wb, i);
		uint32_t bm1 = i > 0 ? arena_base(packed, wb, i - 1) : 0;
		uint32_t bm2 = i > 1 ? arena_base(packed, wb, i - 2) : 0;
		uint32_t bp1 = i + 1 < len ? arena_base(packed, wb, i + 1) : 0;
		uint32_t bctx;
		if (cfg.is_avg) bctx = (bm2 << 6) | (bm1 << 4) | (b0 << 2) | bp1;                       // :203-210
		else if (cfg.is_thr) bctx = b0 | (bm1 << 2) | (bm2 << 4) | (bp1 << 6);                  // :323-339
		else if (cfg.level == 3) bctx = b0 | (bm1 << 2) | (bm2 << 4) | (bp1 << 6);              // :88-108
		else bctx = b0 | (bm1 << 2) | ((uint32_t)(i > 1 && bm2 == bm1) << 4) | (bp1 << 5);
		uint32_t fl = 0;
		if (cfg.level > 1 && flags) { uint8_t c = flags[qb + i]; fl = (c == 'M' ? 1u : 0u) | (c == 'A' ? 2u : 0u); }
		uint32_t ctx = hist | (bctx << cfg.ctx_bits) | (fl << (cfg.ctx_bits + cfg.base_bits));
		key[k_read + i] = (ctx << cfg.sym_bits) | sym;
		sidx_out[k_read + i] = trip_index(lay, part, s_read + navg + i);
	}
	if (__ballot(out_of_range) && lane == 0) atomicOr(bad, 1u);
}

// ---- Q3: run bounds of every context in the sorted key array --------------------------------------
__global__ void k_seg_bounds(const uint32_t* __restrict__ skey, uint64_t n, uint32_t shift, uint32_t* __restrict__ seg_start, uint32_t* __restrict__ seg_end)
{
	uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	uint32_t c = skey[j] >> shift;
	if (j == 0 || (skey[j - 1] >> shift) != c) seg_start[c] = (uint32_t)j;
	if (j + 1 == n || (skey[j + 1] >> shift) != c) seg_end[c] = (uint32_t)j + 1;
}

// ---- Q4a: model evolution for alphabets of <= 5 symbols: one wave per context ----------------------
// 64 symbols per step: per-class ballots give every lane the number of earlier same-class symbols of the
// step; the rescale instant follows from the total alone.
template<uint32_t A>
__global__ __launch_bounds__(256) void k_evolve_small(const uint32_t* __restrict__ skey, const uint32_t* __restrict__ sval,
                                                     const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ seg_end,
                                                     uint32_t n_ctx, uint32_t sym_bits, uint32_t max_total, uint32_t adder,
                                                     uint32_t* __restrict__ state, triple_t* __restrict__ trip)
{
	const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (c >= n_ctx) return;
	const uint32_t s = seg_start[c], e = seg_end[c];
	if (e <= s) return;
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t lt = (1ULL << lane) - 1;
	uint32_t st[A]; uint32_t tot;
#pragma unroll
	for (uint32_t a = 0; a < A; ++a) st[a] = state[(uint64_t)c * (A + 1) + a];
	tot = state[(uint64_t)c * (A + 1) + A];
	const uint32_t smask = (1u << sym_bits) - 1;
	// (the step after this one is asked for before this one's triples go out: a wave's steps are a chain of load -> rank -> scatter, and
	// the scattered stores are what the memory system takes slowest)
	uint32_t nkey = s + lane < e ? skey[s + lane] : 0u, nval = s + lane < e ? sval[s + lane] : 0u;
	for (uint32_t j0 = s; j0 < e; j0 += 64)
	{
		const uint32_t j = j0 + lane;
		const bool valid = j < e;
		const uint32_t sym = valid ? (nkey & smask) : 0xffu;
		const uint32_t dst = valid ? nval : 0u;
		{ const uint32_t jn = j + 64; const bool vn = jn < e; nkey = vn ? skey[jn] : 0u; nval = vn ? sval[jn] : 0u; }
		uint64_t m[A];
#pragma unroll
		for (uint32_t a = 0; a < A; ++a) m[a] = __ballot(valid && sym == a);
		const uint32_t cnt = (e - j0) < 64 ? (e - j0) : 64;
		uint32_t start = 0;
		while (start < cnt)
		{
			// number of updates the current state absorbs before a rescale: the r-th update makes total >= max_total
			uint32_t r = (max_total - tot + adder - 1) / adder;
			uint32_t now = (cnt - start) < r ? (cnt - start) : r;
			uint64_t win = (now == 64 ? ~0ULL : ((1ULL << now) - 1)) << start;       // lanes [start, start+now)
			if (lane >= start && lane < start + now)
			{
				uint64_t before = lt & win;
				uint32_t cum = 0, freq = 0;
#pragma unroll
				for (uint32_t a = 0; a < A; ++a)
				{
					uint32_t v = st[a] + adder * (uint32_t)__popcll(m[a] & before);
					if (a < sym) cum += v;
					if (a == sym) freq = v;
				}
				trip[dst] = pack_triple(cum, freq, tot + adder * (lane - start));
			}
#pragma unroll
			for (uint32_t a = 0; a < A; ++a) st[a] += adder * (uint32_t)__popcll(m[a] & win);
			tot += adder * now;
			while (tot >= max_total)                                                   // rc.h:233-244
			{
				tot = 0;
#pragma unroll
				for (uint32_t a = 0; a < A; ++a) { st[a] = (st[a] + 1) / 2; tot += st[a]; }
			}
			start += now;
		}
	}
	if (lane == 0)
	{
#pragma unroll
		for (uint32_t a = 0; a < A; ++a) state[(uint64_t)c * (A + 1) + a] = st[a];
		state[(uint64_t)c * (A + 1) + A] = tot;
	}
}

// ---- Q4b: model evolution for large alphabets (96 / 256 symbols): one wave per context, counters and
// their exclusive prefix in LDS, 64 symbols per step.  Inside a step lane l needs, besides the table
// values, the number of earlier lanes of the step with a smaller / an equal symbol (uniform readlane loop).
__global__ __launch_bounds__(256) void k_evolve_large(const uint32_t* __restrict__ skey, const uint32_t* __restrict__ sval,
                                                     const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ seg_end,
                                                     uint32_t n_ctx, uint32_t n_sym, uint32_t sym_bits, uint32_t max_total, uint32_t adder,
                                                     uint32_t* __restrict__ state, triple_t* __restrict__ trip)
{
	__shared__ uint32_t s_cnt[4][256];
	__shared__ uint32_t s_pre[4][256];
	const uint32_t w = threadIdx.x >> 6;
	const uint32_t c = blockIdx.x * 4 + w;
	if (c >= n_ctx) return;
	const uint32_t s = seg_start[c], e = seg_end[c];
	if (e <= s) return;
	const uint32_t lane = threadIdx.x & 63;
	uint32_t* sp = state + (uint64_t)c * (n_sym + 1);
	uint32_t* cnt = s_cnt[w]; uint32_t* pre = s_pre[w];
#pragma unroll
	for (uint32_t t = 0; t < 4; ++t) { uint32_t a = lane * 4 + t; cnt[a] = a < n_sym ? sp[a] : 0u; }
	uint32_t tot = sp[n_sym];
	auto rebuild_prefix = [&]() {
		uint32_t v0 = cnt[lane * 4], v1 = cnt[lane * 4 + 1], v2 = cnt[lane * 4 + 2], v3 = cnt[lane * 4 + 3];
		uint32_t sum = v0 + v1 + v2 + v3;
		uint32_t ex = wave_incl_scan(sum) - sum;
		pre[lane * 4] = ex; pre[lane * 4 + 1] = ex + v0; pre[lane * 4 + 2] = ex + v0 + v1; pre[lane * 4 + 3] = ex + v0 + v1 + v2;
		__builtin_amdgcn_wave_barrier();
	};
	__builtin_amdgcn_wave_barrier();
	rebuild_prefix();
	const uint32_t smask = (1u << sym_bits) - 1;
	for (uint32_t j0 = s; j0 < e; j0 += 64)
	{
		const uint32_t j = j0 + lane;
		const bool valid = j < e;
		const uint32_t sym = valid ? (skey[j] & smask) : 0u;
		const uint32_t dst = valid ? sval[j] : 0u;
		const uint32_t n_here = (e - j0) < 64 ? (e - j0) : 64;
		uint32_t start = 0;
		while (start < n_here)
		{
			const uint32_t r = (max_total - tot + adder - 1) / adder;          // updates absorbed before the next rescale
			const uint32_t now = (n_here - start) < r ? (n_here - start) : r;
			uint32_t less = 0, eq = 0;
			for (uint32_t l = start; l < start + now; ++l)
			{
				const uint32_t o = __builtin_amdgcn_readlane(sym, l);
				if (l < lane) { less += (o < sym) ? 1u : 0u; eq += (o == sym) ? 1u : 0u; }
			}
			const bool mine = lane >= start && lane < start + now;
			if (mine) trip[dst] = pack_triple(pre[sym] + adder * less, cnt[sym] + adder * eq, tot + adder * (lane - start));
			__builtin_amdgcn_wave_barrier();
			if (mine) atomicAdd(&cnt[sym], adder);
			__builtin_amdgcn_wave_barrier();
			tot += adder * now;
			while (tot >= max_total)                                               // rc.h:233-244 / 661-676
			{
				uint32_t sum = 0;
#pragma unroll
				for (uint32_t t = 0; t < 4; ++t) { uint32_t a = lane * 4 + t; if (a < n_sym) { uint32_t v = (cnt[a] + 1) / 2; cnt[a] = v; sum += v; } }
				tot = wave_sum(sum);
			}
			__builtin_amdgcn_wave_barrier();
			rebuild_prefix();
			start += now;
		}
	}
#pragma unroll
	for (uint32_t t = 0; t < 4; ++t) { uint32_t a = lane * 4 + t; if (a < n_sym) sp[a] = cnt[a]; }
	if (lane == 0) sp[n_sym] = tot;
}

__global__ void k_gather_u64(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n, uint64_t* __restrict__ dst)
{ uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[idx[i]]; }
__global__ void k_fill_u32(uint32_t* v, uint64_t n, uint32_t x) { uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] = x; }
// state tables: counters 1, total = n_sym
__global__ void k_init_state(uint32_t* st, uint64_t n_ctx, uint32_t n_sym)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_ctx * (n_sym + 1)) return;
	st[i] = (i % (n_sym + 1)) == n_sym ? n_sym : 1u;
}
__global__ void k_gather_bytes(const uint8_t* __restrict__ src, const uint64_t* __restrict__ src_off, const uint64_t* __restrict__ dst_off,
                               const uint64_t* __restrict__ size, uint8_t* __restrict__ dst)
{
	const uint32_t p = blockIdx.x;
	const uint64_t n = size[p]; const uint8_t* s = src + src_off[p]; uint8_t* d = dst + dst_off[p];
	for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
}

void fill_range(uint8_t* a, int lo, int hi, uint8_t v) { for (int i = lo; i < hi; ++i) a[i] = v; }
} // namespace

// CQualityCoder::Init (quality_coder.cpp:26-247): mode tables; adjust_quality_map_* (:250-525)
extern "C" cl_status cl_qual_coder_create(cl_ctx* ctx, const cl_qual_params* prm, cl_qual_coder** out)
{
	if (!ctx || !prm || !out) return cl_fail(ctx, CL_E_INVALID, "cl_qual_coder_create: null argument");
	if (prm->mode < 0 || prm->mode > QM_NONE || prm->source < 0 || prm->source > 2 || prm->level < 1 || prm->level > 3)
		return cl_fail(ctx, CL_E_INVALID, "cl_qual_coder_create: mode 0..8, source 0..2, level 1..3");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	cl_qual_coder* Q = new cl_qual_coder(); Q->ctx = ctx;
	std::unique_ptr<cl_qual_coder> guard(Q);
	QualCfg& c = Q->cfg; memset(&c, 0, sizeof(c));
	c.mode = prm->mode; c.level = prm->level;
	c.max_total = 1u << 18; c.adder = 8;                                    // quality_coder.h:37-41
	auto bins = [&](uint32_t n) -> cl_status {
		if (prm->n_fwd != n - 1) return cl_fail(ctx, CL_E_INVALID, "cl_qual_coder_create: need " + std::to_string(n - 1) + " thresholds");
		for (uint32_t i = 0; i + 1 < n - 1; ++i) if (prm->fwd[i] > prm->fwd[i + 1]) return cl_fail(ctx, CL_E_INVALID, "thresholds must ascend");
		if (prm->fwd[n - 2] > 96) return cl_fail(ctx, CL_E_INVALID, "threshold > 96");
		fill_range(c.map_fwd, 0, (int)prm->fwd[0], 0);
		for (uint32_t b = 1; b + 1 < n; ++b) fill_range(c.map_fwd, (int)prm->fwd[b - 1], (int)prm->fwd[b], (uint8_t)b);
		fill_range(c.map_fwd, (int)prm->fwd[n - 2], 96, (uint8_t)(n - 1));
		c.n_bins = n; c.n_sym = n;
		return CL_OK;
	};
	switch (prm->mode)
	{
	case QM_ORIGINAL:
	{
		for (int i = 0; i < 96; ++i) c.map_fwd[i] = (uint8_t)i;
		static const int ont3[] = { 0, 1, 2, 4, 7, 11, 16, 22, 29, 37, 46, 56, 67, 79, 90, 96 };
		static const int ont12[] = { 0, 1, 2, 5, 10, 15, 20, 25, 35, 50, 70, 96 };
		static const int pb3[] = { 0, 1, 10, 20, 30, 39, 45, 51, 57, 63, 69, 75, 81, 87, 93, 94 };
		static const int pb12[] = { 0, 1, 15, 29, 41, 53, 63, 72, 80, 87, 93, 94 };
		const int* t; int n;
		if (prm->source == 0) { if (prm->level == 3) { t = ont3; n = 15; } else { t = ont12; n = 11; } }
		else { if (prm->level == 3) { t = pb3; n = 15; } else { t = pb12; n = 11; } }
		for (int b = 0; b < n; ++b) fill_range(c.quant, t[b], t[b + 1], (uint8_t)b);
		if (prm->source == 2) { for (int i = 0; i < 93; ++i) c.quant[i] += 1; c.quant[93] = 0; }
		c.bits_per_sym = 4; c.n_ctx_sym = 2; c.n_sym = 96; c.max_total = 1u << 20; c.adder = 32;   // quality_coder.h:36
		c.base_bits = prm->level == 3 ? 8 : 7;
		break;
	}
	case QM_QUINARY_AVG: CL_TRY(bins(5)); c.bits_per_sym = 3; c.n_ctx_sym = 3; c.is_avg = 1; c.navg = 10; c.base_bits = 8; break;
	case QM_QUAD_AVG: CL_TRY(bins(4)); c.bits_per_sym = 3; c.n_ctx_sym = 3; c.is_avg = 1; c.navg = 8; c.base_bits = 8; break;
	case QM_BINARY_AVG: CL_TRY(bins(2)); c.bits_per_sym = 2; c.n_ctx_sym = 6; c.is_avg = 1; c.navg = 4; c.base_bits = 8; break;
	case QM_QUINARY_THR: CL_TRY(bins(5)); c.bits_per_sym = 3; c.n_ctx_sym = 3; c.is_thr = 1; c.base_bits = 8; break;
	case QM_QUAD_THR: CL_TRY(bins(4)); c.bits_per_sym = 3; c.n_ctx_sym = 3; c.is_thr = 1; c.base_bits = 8; break;
	case QM_BINARY_THR: CL_TRY(bins(2)); c.bits_per_sym = 2; c.n_ctx_sym = 6; c.is_thr = 1; c.base_bits = 8; break;
	case QM_AVERAGE: c.navg = 2; c.n_sym = 2; break;
	case QM_NONE: c.n_sym = 2; break;
	}
	c.ctx_bits = c.bits_per_sym * c.n_ctx_sym;
	c.sym_bits = 1; while ((1u << c.sym_bits) < c.n_sym) ++c.sym_bits;
	uint32_t total_ctx_bits = c.ctx_bits + c.base_bits + (c.level > 1 ? 2 : 0);
	if (total_ctx_bits + c.sym_bits > 32) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_qual_coder_create: context does not fit 32-bit keys");
	c.n_ctx = (c.mode == QM_AVERAGE || c.mode == QM_NONE) ? 1u : (1u << total_ctx_bits);
	DEV_ALLOC(ctx, Q->d_cfg, 1);
	HIP_TRY(ctx, hipMemcpyAsync(Q->d_cfg.p, &c, sizeof(c), hipMemcpyHostToDevice, ctx->stream));
	DEV_ALLOC(ctx, Q->state, (uint64_t)c.n_ctx * (c.n_sym + 1));
	LAUNCH(ctx, k_init_state, grid_for((uint64_t)c.n_ctx * (c.n_sym + 1), 256), 256, Q->state.p, (uint64_t)c.n_ctx, c.n_sym);
	DEV_ALLOC(ctx, Q->bstate, (uint64_t)BYTE_CTX * 257);
	LAUNCH(ctx, k_init_state, grid_for((uint64_t)BYTE_CTX * 257, 256), 256, Q->bstate.p, (uint64_t)BYTE_CTX, 256u);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	*out = guard.release();
	return CL_OK;
}
extern "C" cl_ctx* cl_qual_coder_ctx(const cl_qual_coder* q) { return q ? q->ctx : nullptr; }
extern "C" void cl_qual_coder_free(cl_qual_coder* q) { delete q; }

// CEntrComprQuals::Compress for a batch of whole parts (entr_qual.h:100-135).  Models persist across calls.
namespace {
cl_status qual_prepare(cl_ctx* ctx, cl_qual_coder* Q, const cl_reads* R, const uint8_t* d_quals, const uint64_t* d_qual_off,
                       const uint8_t* d_flags, const uint32_t* h_part_bounds, uint32_t n_parts, QualPrepared& P)
{
	const QualCfg& c = Q->cfg;
	P.R = R; P.d_quals = d_quals; P.part_bounds.assign(h_part_bounds, h_part_bounds + n_parts + 1);
	// host copy of the quality offsets at part boundaries
	std::vector<uint64_t> qo(n_parts + 1);
	{
		DevBuf<uint32_t> d_pb; DEV_ALLOC(ctx, d_pb, (uint64_t)n_parts + 1);
		DevBuf<uint64_t> d_qo; DEV_ALLOC(ctx, d_qo, (uint64_t)n_parts + 1);
		HIP_TRY(ctx, hipMemcpyAsync(d_pb.p, h_part_bounds, ((uint64_t)n_parts + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
		LAUNCH(ctx, k_gather_u64, grid_for((uint64_t)n_parts + 1, 256), 256, d_qual_off, (const uint32_t*)d_pb.p, (uint64_t)n_parts + 1, d_qo.p);
		HIP_TRY(ctx, hipMemcpyAsync(qo.data(), d_qo.p, ((uint64_t)n_parts + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	// quality bytes index the 96-entry maps: anything outside Phred+33 0..95 is refused, not coded (the reference would read past its
	// tables).  k_qual_symbols looks at every byte anyway and reports here.  (Mode `none` codes nothing and never comes here: as in the
	// reference, whatever its quality bytes are is ignored.)
	DevBuf<uint32_t> bad; DEV_ALLOC(ctx, bad, 1);
	HIP_TRY(ctx, hipMemsetAsync(bad.p, 0, 4, ctx->stream));
	auto refuse_bad = [&]() -> cl_status {
		uint32_t h_bad = 0;
		HIP_TRY(ctx, hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		if (h_bad) return cl_fail(ctx, CL_E_INVALID, "cl_qual_encode: quality byte outside '!'..'~'+1 (Phred+33 values 0..95)");
		return CL_OK;
	};
	// process groups of parts so that one group's symbol stream stays below 2^31
	// One group of parts = one sort + one range-coding launch.  The per-part interval chain is latency bound (its
	// duration is set by the longest part, not by the number of parts), so groups are made as large as 32-bit
	// symbol indices allow: every extra launch would pay that latency again.
	const uint64_t GROUP_SYMS = (1ull << 31) - (1ull << 24);
	uint32_t p0 = 0;
	while (p0 < n_parts)
	{
		auto Gp = std::make_unique<QualGroupPrep>(); QualGroupPrep& G = *Gp;
		uint32_t p1 = p0 + 1;
		const bool per_base = !(c.mode == QM_AVERAGE);
		auto syms_of = [&](uint32_t a, uint32_t b) { return (per_base ? (qo[b] - qo[a]) : 0) + (uint64_t)(h_part_bounds[b] - h_part_bounds[a]) * c.navg; };
		while (p1 < n_parts && syms_of(p0, p1 + 1) <= GROUP_SYMS) ++p1;
		const uint32_t r0 = h_part_bounds[p0], r1 = h_part_bounds[p1];
		const uint64_t n_base = qo[p1] - qo[p0], n_syms = syms_of(p0, p1), n_byte = (uint64_t)(r1 - r0) * c.navg;
		if (n_syms >= (1ull << 32)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_qual_encode: a single part has >= 2^32 symbols");
		const uint32_t np = p1 - p0;
		// interleaved triple layout: groups of 64 parts, padded to the longest part of the group
		const uint32_t ng = (np + 63) / 64;
		G.p0 = p0; G.p1 = p1; G.np = np; G.ng = ng; G.n_base = per_base ? n_base : 0; G.n_syms = n_syms; G.n_byte = n_byte;
		std::vector<uint64_t> sym_start(np + 1), gbase(ng + 1);
		std::vector<uint32_t> plen(np), pfirst(np + 1);
		sym_start[0] = 0;
		for (uint32_t p = 0; p < np; ++p)
		{
			uint64_t sl = syms_of(p0 + p, p0 + p + 1);
			if (sl >= (1ull << 31)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_qual_encode: a part has >= 2^31 symbols");
			plen[p] = (uint32_t)sl; sym_start[p + 1] = sym_start[p] + sl; pfirst[p] = h_part_bounds[p0 + p];
		}
		pfirst[np] = h_part_bounds[p1];
		// places in the interleaved layout by descending length (as in cl_dna_encode): the 64 parts of a wave of the interval
		// coder are alike and no slots are wasted on the longest part of a group
		std::vector<uint32_t> order(np); std::vector<uint32_t>& rank = G.rank; std::vector<uint32_t>& plen_r = G.plen_r;
		rank.resize(np); plen_r.resize(np);
		for (uint32_t p = 0; p < np; ++p) order[p] = p;
		std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return plen[a] > plen[b]; });
		for (uint32_t i = 0; i < np; ++i) { rank[order[i]] = i; plen_r[i] = plen[order[i]]; }
		gbase[0] = 0;
		for (uint32_t g = 0; g < ng; ++g) gbase[g + 1] = gbase[g] + trip_group_words(plen_r[g * 64]);
		if (gbase[ng] >= (1ull << 32)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_qual_encode: group too large for 32-bit triple indices");
		G.trip_words = gbase[ng];
		DevBuf<uint64_t> d_sym_start; DevBuf<uint32_t> d_pfirst, d_rank;
		DEV_ALLOC(ctx, d_sym_start, np + 1); DEV_ALLOC(ctx, G.d_gbase, ng + 1); DEV_ALLOC(ctx, G.d_plen, np); DEV_ALLOC(ctx, d_pfirst, np + 1); DEV_ALLOC(ctx, d_rank, np);
		HIP_TRY(ctx, hipMemcpyAsync(d_rank.p, rank.data(), np * 4, hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(d_sym_start.p, sym_start.data(), (np + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(G.d_gbase.p, gbase.data(), (ng + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(G.d_plen.p, plen_r.data(), np * 4, hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(d_pfirst.p, pfirst.data(), (np + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
		TripLayoutDev lay{ d_pfirst.p, d_sym_start.p, G.d_gbase.p, np, d_rank.p };
		DevBuf<uint32_t>& key = G.key; DevBuf<uint32_t>& sidx = G.sidx; DevBuf<uint32_t>& bkey = G.bkey; DevBuf<uint32_t>& bsidx = G.bsidx;
		DEV_ALLOC(ctx, key, per_base ? n_base : 0); DEV_ALLOC(ctx, sidx, per_base ? n_base : 0);
		DEV_ALLOC(ctx, bkey, n_byte); DEV_ALLOC(ctx, bsidx, n_byte);
		if (r1 > r0)
			LAUNCHB(ctx, n_base * (1.0 + 0.25 + 8.0) + n_byte * 8.0, k_qual_symbols, grid_for(r1 - r0, 4), 256, (const QualCfg*)Q->d_cfg.p, (const uint64_t*)R->packed.p, (const uint64_t*)R->word_off.p,
				d_quals, d_qual_off, d_flags, r0, r1, qo[p0], lay, key.p, sidx.p, bkey.p, bsidx.p, bad.p);
		HIP_TRY(ctx, hipGetLastError());
		const uint32_t total_ctx_bits = c.ctx_bits + c.base_bits + (c.level > 1 ? 2 : 0);
		if (per_base && n_base)
		{
			CL_TRY(dev_sort_keys32_pairs(ctx, key.p, sidx.p, n_base, c.sym_bits, c.sym_bits + total_ctx_bits));
			DEV_ALLOC(ctx, G.ss, c.n_ctx); DEV_ALLOC(ctx, G.se, c.n_ctx);
			HIP_TRY(ctx, hipMemsetAsync(G.ss.p, 0, (uint64_t)c.n_ctx * 4, ctx->stream));
			HIP_TRY(ctx, hipMemsetAsync(G.se.p, 0, (uint64_t)c.n_ctx * 4, ctx->stream));
			LAUNCH(ctx, k_seg_bounds, grid_for(n_base, 256), 256, (const uint32_t*)key.p, n_base, c.sym_bits, G.ss.p, G.se.p);
			HIP_TRY(ctx, hipGetLastError());
		}
		if (n_byte)
		{
			CL_TRY(dev_sort_keys32_pairs(ctx, bkey.p, bsidx.p, n_byte, 8, 8 + 10));
			DEV_ALLOC(ctx, G.bss, BYTE_CTX); DEV_ALLOC(ctx, G.bse, BYTE_CTX);
			HIP_TRY(ctx, hipMemsetAsync(G.bss.p, 0, BYTE_CTX * 4, ctx->stream));
			HIP_TRY(ctx, hipMemsetAsync(G.bse.p, 0, BYTE_CTX * 4, ctx->stream));
			LAUNCH(ctx, k_seg_bounds, grid_for(n_byte, 256), 256, (const uint32_t*)bkey.p, n_byte, 8u, G.bss.p, G.bse.p);
			HIP_TRY(ctx, hipGetLastError());
		}
		CL_TRY(refuse_bad());                                                    // (waits for the stream: the uploads above read host vectors of this frame)
		P.groups.push_back(std::move(Gp));
		p0 = p1;
	}
	return CL_OK;
}
} // namespace
// Internal (stream.hip): the model-independent half of cl_qual_encode for a batch, on any context; cl_qual_set_ahead hands it to
// the coder, whose next cl_qual_encode uses it if it is the batch (arena, qualities, part bounds) it was made for.
cl_status cl_qual_prepare_batch(cl_ctx* ctx, cl_qual_coder* Q, const cl_reads* R, const uint8_t* d_quals, const uint64_t* d_qual_off,
                                const uint8_t* d_flags, const uint32_t* h_part_bounds, uint32_t n_parts, QualPrepared** out)
{
	if (!ctx || !Q || !R || !d_qual_off || !h_part_bounds || !n_parts || !out || Q->cfg.mode == QM_NONE) return CL_E_INVALID;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	for (uint32_t p = 0; p < n_parts; ++p) if (h_part_bounds[p] > h_part_bounds[p + 1]) return CL_E_INVALID;
	if (h_part_bounds[n_parts] > R->n_reads) return CL_E_INVALID;
	auto P = std::make_unique<QualPrepared>();
	CL_TRY(qual_prepare(ctx, Q, R, d_quals, d_qual_off, d_flags, h_part_bounds, n_parts, *P));
	*out = P.release();
	return CL_OK;
}
void cl_qual_prepared_free(QualPrepared* P) { delete P; }
void cl_qual_set_ahead(cl_qual_coder* Q, QualPrepared* P) { if (Q) Q->ahead.reset(P); else delete P; }

namespace {
// the model half of a batch: model evolution group by group, every group's interval coder started on a stream of the coder
cl_status qual_evolve_batch(cl_ctx* ctx, cl_qual_coder* Q, const cl_reads* R, const uint8_t* d_quals, const uint64_t* d_qual_off, const uint8_t* d_flags,
                            const uint32_t* h_part_bounds, uint32_t n_parts, std::unique_ptr<QualPrepared> Pp, QualEvolved& E)
{
	const QualCfg& c = Q->cfg;
	// the model-independent half: made ahead (cl_qual_prepare_batch) for exactly this batch, or here
	if (Pp && !(Pp->R == R && Pp->d_quals == d_quals && Pp->part_bounds.size() == (size_t)n_parts + 1 && memcmp(Pp->part_bounds.data(), h_part_bounds, ((size_t)n_parts + 1) * 4) == 0)) Pp.reset();
	if (!Pp) { Pp = std::make_unique<QualPrepared>(); CL_TRY(qual_prepare(ctx, Q, R, d_quals, d_qual_off, d_flags, h_part_bounds, n_parts, *Pp)); }
	E.R = R; E.d_quals = d_quals; E.part_bounds.assign(h_part_bounds, h_part_bounds + n_parts + 1);
	const uint32_t bits_max = c.max_total == (1u << 20) ? 20 : 18;
	const uint64_t* inv_tab = nullptr;
	CL_TRY(cl_inv_table(ctx, &inv_tab));
	for (auto& Gp : Pp->groups)
	{
		QualGroupPrep& G = *Gp;
		const uint32_t np = G.np, ng = G.ng;
		const uint64_t n_base = G.n_base, n_syms = G.n_syms, n_byte = G.n_byte;
		auto Pn = std::make_unique<QualPending>(); QualPending& PG = *Pn;
		PG.p0 = G.p0; PG.np = np; PG.rank = G.rank; PG.n_syms = n_syms;
		PG.d_gbase = std::move(G.d_gbase); PG.d_plen = std::move(G.d_plen);
		DevBuf<triple_t>& trip = PG.trip; DEV_ALLOC(ctx, trip, G.trip_words);
		if (n_base)
		{
			const uint32_t g = grid_for(c.n_ctx, 4);
			switch (c.n_sym)
			{
			case 2: LAUNCHB(ctx, n_base * 16.0, (k_evolve_small<2>), g, 256, (const uint32_t*)G.key.p, (const uint32_t*)G.sidx.p, (const uint32_t*)G.ss.p, (const uint32_t*)G.se.p, c.n_ctx, c.sym_bits, c.max_total, c.adder, Q->state.p, trip.p); break;
			case 4: LAUNCHB(ctx, n_base * 16.0, (k_evolve_small<4>), g, 256, (const uint32_t*)G.key.p, (const uint32_t*)G.sidx.p, (const uint32_t*)G.ss.p, (const uint32_t*)G.se.p, c.n_ctx, c.sym_bits, c.max_total, c.adder, Q->state.p, trip.p); break;
			case 5: LAUNCHB(ctx, n_base * 16.0, (k_evolve_small<5>), g, 256, (const uint32_t*)G.key.p, (const uint32_t*)G.sidx.p, (const uint32_t*)G.ss.p, (const uint32_t*)G.se.p, c.n_ctx, c.sym_bits, c.max_total, c.adder, Q->state.p, trip.p); break;
			default: LAUNCHB(ctx, n_base * 16.0, k_evolve_large, g, 256, (const uint32_t*)G.key.p, (const uint32_t*)G.sidx.p, (const uint32_t*)G.ss.p, (const uint32_t*)G.se.p, c.n_ctx, c.n_sym, c.sym_bits, c.max_total, c.adder, Q->state.p, trip.p); break;
			}
			HIP_TRY(ctx, hipGetLastError());
		}
		if (n_byte)
		{
			LAUNCH(ctx, k_evolve_large, grid_for(BYTE_CTX, 4), 256, (const uint32_t*)G.bkey.p, (const uint32_t*)G.bsidx.p, (const uint32_t*)G.bss.p, (const uint32_t*)G.bse.p,
				BYTE_CTX, 256u, 8u, 1u << 18, 8u, Q->bstate.p, trip.p);
			HIP_TRY(ctx, hipGetLastError());
		}
		// range coding: worst case bits_max bits per symbol + 8 flush bytes, rounded to 8-byte aligned regions
		std::vector<uint64_t> out_off(np + 1);
		out_off[0] = 0;
		for (uint32_t p = 0; p < np; ++p)
		{
			uint64_t s = G.plen_r[p];                                           // (by place)
			out_off[p + 1] = out_off[p] + ((s * bits_max + 7) / 8 + s / 16 + 64 + 7) / 8 * 8;
		}
		DEV_ALLOC(ctx, PG.tmp, out_off[np]);
		DEV_ALLOC(ctx, PG.d_out_off, np + 1); DEV_ALLOC(ctx, PG.d_size, np); DEV_ALLOC(ctx, PG.d_dst_off, np);
		HIP_TRY(ctx, hipMemcpyAsync(PG.d_out_off.p, out_off.data(), (np + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                         // the triples are complete (and out_off is read)
		G.key.release(); G.sidx.release(); G.bkey.release(); G.bsidx.release();   // (the models are through with them)
		// the interval arithmetic, one dependent chain per part, on a stream of the coder's own: the model half of the next batch
		// runs beside it (cl_qual_evolve_ahead)
		if (Q->cstreams.size() < 4) { hipStream_t ns = nullptr; HIP_TRY(ctx, cl_stream_create_role(CL_ROLE_CODER, cl_level_to_prio(cl_role_level(CL_ROLE_CODER, 0)), &ns)); Q->cstreams.push_back(ns); }
		PG.stream = Q->cstreams[Q->next_cstream++ % Q->cstreams.size()];
		{
			LaunchOn on(ctx, PG.stream);                                         // (launch + timing events on the coder's stream)
			LAUNCH_RANGE_CODE(ctx, n_syms * 8.0, ng, (const triple_t*)trip.p, (const uint64_t*)PG.d_gbase.p, (const uint32_t*)PG.d_plen.p, np, PG.tmp.p, (const uint64_t*)PG.d_out_off.p, PG.d_size.p, inv_tab);
			hipError_t e2 = hipGetLastError();
			HIP_TRY(ctx, e2);
		}
		E.groups.push_back(std::move(Pn));
	}
	return CL_OK;
}
} // namespace
// Internal (stream.hip): the model half of the batch that FOLLOWS the one being coded, from cl_qual_encode's before_tail hook
cl_status cl_qual_evolve_ahead(cl_ctx* ctx, cl_qual_coder* Q, const cl_reads* R, const uint8_t* d_quals, const uint64_t* d_qual_off, const uint32_t* h_part_bounds, uint32_t n_parts, QualPrepared* P)
{
	std::unique_ptr<QualPrepared> Pp(P);
	if (!ctx || !Q || !R || !h_part_bounds || !n_parts || Q->evolved.size() >= 4 || Q->cfg.mode == QM_NONE) return CL_E_INVALID;
	auto E = std::make_unique<QualEvolved>();
	CL_TRY(qual_evolve_batch(ctx, Q, R, d_quals, d_qual_off, nullptr, h_part_bounds, n_parts, std::move(Pp), *E));
	Q->evolved.push_back(std::move(E));
	return CL_OK;
}
void cl_qual_set_before_tail(cl_qual_coder* Q, std::function<cl_status()> fn) { if (Q) Q->before_tail = std::move(fn); }

extern "C" cl_status cl_qual_encode(cl_ctx* ctx, cl_qual_coder* Q, const cl_reads* R, const uint8_t* d_quals, const uint64_t* d_qual_off,
                                    const uint8_t* d_flags, const uint32_t* h_part_bounds, uint32_t n_parts,
                                    uint8_t* d_out, uint64_t cap, uint64_t* h_part_sizes, uint64_t* n_out)
{
	if (!ctx || !Q || !R || !d_qual_off || !h_part_bounds || !h_part_sizes || !n_out) return cl_fail(ctx, CL_E_INVALID, "cl_qual_encode: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const QualCfg& c = Q->cfg;
	for (uint32_t p = 0; p < n_parts; ++p) if (h_part_bounds[p] > h_part_bounds[p + 1]) return cl_fail(ctx, CL_E_INVALID, "cl_qual_encode: part bounds must ascend");
	if (n_parts && h_part_bounds[n_parts] > R->n_reads) return cl_fail(ctx, CL_E_INVALID, "cl_qual_encode: part bound beyond the arena");
	*n_out = 0;
	std::unique_ptr<QualPrepared> Pp = std::move(Q->ahead);
	std::unique_ptr<QualEvolved> Ep;
	if (!Q->evolved.empty()) { Ep = std::move(Q->evolved.front()); Q->evolved.pop_front(); }
	if (!n_parts) return Ep ? cl_fail(ctx, CL_E_INVALID, "cl_qual_encode: the batch evolved ahead is not the one encoded next") : CL_OK;
	if (c.mode == QM_NONE)
	{	// nothing is coded: every part is the 8 flush bytes of an untouched coder (zeros)
		if (cap < 8ull * n_parts) { *n_out = 8ull * n_parts; return cl_fail(ctx, CL_E_CAPACITY, "cl_qual_encode: output capacity"); }
		HIP_TRY(ctx, hipMemsetAsync(d_out, 0, 8ull * n_parts, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		for (uint32_t p = 0; p < n_parts; ++p) h_part_sizes[p] = 8;
		*n_out = 8ull * n_parts;
		return CL_OK;
	}
	if (Ep)
	{	// (the models are evolved with it already: it has to be this batch)
		if (!(Ep->R == R && Ep->d_quals == d_quals && Ep->part_bounds.size() == (size_t)n_parts + 1 && memcmp(Ep->part_bounds.data(), h_part_bounds, ((size_t)n_parts + 1) * 4) == 0))
			return cl_fail(ctx, CL_E_INVALID, "cl_qual_encode: the batch evolved ahead is not the one encoded next");
	}
	else { Ep = std::make_unique<QualEvolved>(); CL_TRY(qual_evolve_batch(ctx, Q, R, d_quals, d_qual_off, d_flags, h_part_bounds, n_parts, std::move(Pp), *Ep)); }
	// this batch's interval coders run; the caller's hook may take the next batch through its model half beside them
	if (Q->before_tail)
		for (;;)
		{	// (CL_HOOK_RETRY: the next batch is not prepared yet — asked again as long as this batch's coders run)
			const cl_status hs = Q->before_tail();
			if (hs == CL_OK) break;
			if (hs != CL_HOOK_RETRY) return hs;
			bool running = false;
			for (auto& g : Ep->groups) if (hipStreamQuery(g->stream) == hipErrorNotReady) running = true;
			(void)hipGetLastError();
			if (!running) break;
			std::this_thread::sleep_for(std::chrono::milliseconds(1));
		}
	uint64_t written = 0;
	for (auto& Pn : Ep->groups)
	{
		QualPending& PG = *Pn;
		const uint32_t p0 = PG.p0, np = PG.np;
		HIP_TRY(ctx, hipStreamSynchronize(PG.stream));
		std::vector<uint64_t> size_r(np);                                       // by place
		HIP_TRY(ctx, hipMemcpy(size_r.data(), PG.d_size.p, np * 8, hipMemcpyDeviceToHost));
		for (uint32_t p = 0; p < np; ++p)
		{
			h_part_sizes[p0 + p] = size_r[PG.rank[p]];
			if (h_part_sizes[p0 + p] == ~0ULL) return cl_fail(ctx, CL_E_CAPACITY, "cl_qual_encode: internal part buffer overflow (pathological interval clamping)");
		}
		std::vector<uint64_t> dst_off(np);                                      // by place, the bytes in part order
		uint64_t w = written;
		for (uint32_t p = 0; p < np; ++p) { dst_off[PG.rank[p]] = w; w += h_part_sizes[p0 + p]; }
		if (w > cap) { *n_out = w; return cl_fail(ctx, CL_E_CAPACITY, "cl_qual_encode: output capacity " + std::to_string(cap) + " too small"); }
		HIP_TRY(ctx, hipMemcpyAsync(PG.d_dst_off.p, dst_off.data(), np * 8, hipMemcpyHostToDevice, ctx->stream));
		LAUNCH(ctx, k_gather_bytes, np, 256, (const uint8_t*)PG.tmp.p, (const uint64_t*)PG.d_out_off.p, (const uint64_t*)PG.d_dst_off.p, (const uint64_t*)PG.d_size.p, d_out);
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		written = w;
		Q->symbols_coded += PG.n_syms;
	}
	cl_timing_collect(ctx);
	*n_out = written;
	return CL_OK;
}
