// rc_dev.hpp — device-side interval arithmetic of the 64-bit carry-less range coder (sub_rc.h:44-212) and the
// interleaved "triple" layout that feeds it.  Shared by the quality and DNA coders.
//
// A part (= one range-coder restart, entr_qual.h:68-79 / entr_read.h:69-77) is coded by ONE lane: the
// recurrence on (low, range) is a dependent chain.  64 consecutive parts form a group handled by one
// wavefront; the triples of a group are stored interleaved — index = group_base + pos * 64 + lane — so that
// every step of the wavefront is one coalesced 512-byte load.
#pragma once
#include "common.hpp"

// A coded symbol as the interval coder needs it: cum << 42 | freq << 21 | tot in ONE 64-bit word (totals stay below 2^21).
// The coder divides by tot through a multiplication with floor((2^64-1) / tot); that reciprocal used to travel with every
// symbol (16-byte triples: a 64-bit division per symbol in the model kernels, twice the memory and twice the coder's
// fetches).  It now comes from a table of the 2^21 possible totals (16 MB per context, filled once; the totals in use are a few
// thousand neighbouring values, L2-resident), looked up one step ahead of the dependent chain.
typedef uint64_t triple_t;
__device__ inline triple_t pack_triple(uint32_t cum, uint32_t freq, uint32_t tot)
{
	return ((uint64_t)cum << 42) | ((uint64_t)freq << 21) | tot;
}
constexpr uint32_t INV_TABLE_SIZE = 1u << 21;
static __global__ void k_fill_inv_table(uint64_t* __restrict__ tab)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < INV_TABLE_SIZE) tab[i] = i ? ~0ULL / (uint64_t)i : ~0ULL;
}
// the context's reciprocal table (made at first use)
static inline cl_status cl_inv_table(cl_ctx* ctx, const uint64_t** out)
{
	if (!ctx->inv_tab)
	{
		HIP_TRY(ctx, hipMalloc((void**)&ctx->inv_tab, (uint64_t)INV_TABLE_SIZE * 8));
		hipLaunchKernelGGL(k_fill_inv_table, dim3(INV_TABLE_SIZE / 256), dim3(256), 0, ctx->stream, ctx->inv_tab);
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	*out = ctx->inv_tab;
	return CL_OK;
}

struct TripLayoutDev {
	const uint32_t* part_first_read;   // np + 1 read indices (absolute)
	const uint64_t* part_sym_start;    // np + 1 stream positions (relative to the call's first symbol)
	const uint64_t* group_base;        // ceil(np / 64)
	uint32_t np;
	const uint32_t* rank = nullptr;    // optional: part -> place in the interleaved layout (places 64 g .. 64 g + 63 form group g);
	                                   // parts placed by descending length waste no slots on the longest part of their group
};
// part containing read r (binary search over the part bounds)
__device__ inline uint32_t part_of_read(const TripLayoutDev& L, uint32_t r)
{
	uint32_t lo = 0, hi = L.np;                    // invariant: bounds[lo] <= r < bounds[hi]
	while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (L.part_first_read[mid] <= r) lo = mid; else hi = mid; }
	return lo;
}
__device__ inline uint32_t trip_index(const TripLayoutDev& L, uint32_t part, uint64_t stream_pos)
{
	const uint32_t pl = L.rank ? L.rank[part] : part;
	return (uint32_t)(L.group_base[pl >> 6] + (stream_pos - L.part_sym_start[part]) * 64 + (pl & 63));
}

// floor(x / d) given inv = floor((2^64-1) / d): the high product is at most 2 below the quotient.
__device__ inline uint64_t div_by_inv(uint64_t x, uint32_t d, uint64_t inv)
{
	uint64_t q = __umul64hi(x, inv);
	uint64_t r = x - q * d;
	while (r >= d) { ++q; r -= d; }
	return q;
}

// Output bytes of one part.  The coder emits the top byte of `low` on every renormalisation step and only
// shifts `low` in between, so the n bytes of one symbol are simply the top n bytes of `low`: they are
// appended in one operation (big-endian accumulator, 8-byte aligned stores).
struct ByteSink {
	uint8_t* p; uint64_t n; uint64_t acc; uint32_t fill; uint64_t cap; bool overflow;
	// append the top `nb` bytes (0..8) of v
	__device__ inline void put_top(uint64_t v, uint32_t nb)
	{
		if (nb == 0) return;
		const uint64_t B = v >> (64 - 8 * nb);                      // nb >= 1
		const uint32_t total = fill + nb;
		if (total < 8) { acc = (acc << (8 * nb)) | B; fill = total; return; }
		const uint32_t k = 8 - fill;                                 // bytes that complete the word, 1..8
		const uint32_t rest = nb - k;                                // 0..7
		const uint64_t head = rest ? (B >> (8 * rest)) : B;
		const uint64_t word = (k == 8) ? head : ((acc << (8 * k)) | head);
		if (n + 8 <= cap) *(uint64_t*)(p + n) = __builtin_bswap64(word); else overflow = true;
		n += 8;
		acc = rest ? (B & ((1ULL << (8 * rest)) - 1)) : 0; fill = rest;
	}
	__device__ inline void flush()
	{
		if (n + fill <= cap) { for (uint32_t i = 0; i < fill; ++i) p[n + i] = (uint8_t)(acc >> (8 * (fill - 1 - i))); } else overflow = true;
		n += fill; fill = 0; acc = 0;
	}
};

// one lane per part, one wave per group of 64 parts (sub_rc.h:72-100,203-210)
static __global__ __launch_bounds__(64) void k_range_code(const triple_t* __restrict__ trip, const uint64_t* __restrict__ group_base,
                                                         const uint32_t* __restrict__ part_len, uint32_t n_parts,
                                                         uint8_t* __restrict__ out, const uint64_t* __restrict__ part_out_off, uint64_t* __restrict__ part_size,
                                                         const uint64_t* __restrict__ inv_tab)
{
	__builtin_amdgcn_s_setprio(3);                                          // a launch of this kernel lasts as long as its slowest chain: its waves go first on their SIMDs
	const uint32_t p = blockIdx.x * 64 + threadIdx.x;
	const bool live = p < n_parts;
	const uint64_t TOP = 0x00ffffffffffffULL, MASK = 0xff00000000000000ULL;
	uint64_t low = 0, range = MASK;
	ByteSink sink{ live ? out + part_out_off[p] : nullptr, 0, 0, 0, live ? part_out_off[p + 1] - part_out_off[p] : 0, false };
	const uint32_t len = live ? part_len[p] : 0;
	uint32_t lmax = len;
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) { uint32_t t = __shfl_xor(lmax, d, 64); lmax = t > lmax ? t : lmax; }
	if (lmax == 0) { if (live) { sink.put_top(0, 8); sink.flush(); part_size[p] = sink.n; } return; }
	const triple_t* src = trip + group_base[blockIdx.x] + threadIdx.x;
	constexpr uint32_t U = 8;
	// three stages ahead of the chain: symbols two rounds ahead, their reciprocals one round ahead (looked up from the symbols
	// fetched the round before), the round being coded.  The three register sets trade roles from round to round (no copies).
	triple_t A[U], B[U], C[U]; uint64_t iA[U], iB[U], iC[U];
	// a lane whose part is shorter than the group's longest keeps stepping with the neutral symbol
	// (cum 0, freq 1, total 1): range / 1 * 1 and low + 0 leave the coder untouched.
	const uint64_t NEUTRAL_X = (1ULL << 21) | 1ULL, NEUTRAL_Y = ~0ULL;
	const uint32_t last = lmax - 1;
#pragma unroll
	for (uint32_t u = 0; u < U; ++u) { A[u] = src[(uint64_t)(u < last ? u : last) * 64]; B[u] = src[(uint64_t)(U + u < last ? U + u : last) * 64]; }
#pragma unroll
	for (uint32_t u = 0; u < U; ++u) iA[u] = inv_tab[A[u] & 0x1fffff];
	// one round: fetch `far` (two rounds ahead), look up the reciprocals of `nxt`, code `cur` with `icur`
	auto round = [&](uint32_t pos, const triple_t (&cur)[U], const uint64_t (&icur)[U], const triple_t (&nxt)[U], uint64_t (&inxt)[U], triple_t (&far)[U])
	{
#pragma unroll
		for (uint32_t u = 0; u < U; ++u) { uint32_t q = pos + 2 * U + u; far[u] = src[(uint64_t)(q < last ? q : last) * 64]; }   // prefetch (index clamped, never a pointer select)
#pragma unroll
		for (uint32_t u = 0; u < U; ++u) inxt[u] = inv_tab[nxt[u] & 0x1fffff];
#pragma unroll
		for (uint32_t u = 0; u < U; ++u)
		{
			const bool act = pos + u < len;
			const uint64_t tx = act ? cur[u] : NEUTRAL_X, inv = act ? icur[u] : NEUTRAL_Y;
			const uint32_t tot = (uint32_t)(tx & 0x1fffff), freq = (uint32_t)((tx >> 21) & 0x1fffff), cum = (uint32_t)(tx >> 42);
			uint64_t q = __umul64hi(range, inv);
			uint64_t r = range - q * tot;
			if (r >= tot) { ++q; r -= tot; }
			if (r >= tot) { ++q; }
			range = q;
			low += range * cum;
			range *= freq;
			if (range == 0) { sink.overflow = true; range = MASK; }           // only with corrupt triples; keeps the loop finite
			const uint64_t low0 = low;
			uint32_t nb = 0;
			while (range <= TOP)
			{
				if ((low ^ (low + range)) & MASK) { uint64_t rr = low; range = (rr | TOP) - rr; }
				low <<= 8; range <<= 8; ++nb;
			}
			sink.put_top(low0, nb);
		}
	};
	for (uint32_t pos = 0; pos < lmax; pos += 3 * U)
	{
		round(pos, A, iA, B, iB, C);
		if (pos + U >= lmax) break;
		round(pos + U, B, iB, C, iC, A);
		if (pos + 2 * U >= lmax) break;
		round(pos + 2 * U, C, iC, A, iA, B);
	}
	if (!live) return;
	sink.put_top(low, 8);                                                    // End(): 8 bytes of low (sub_rc.h:203-210)
	sink.flush();
	part_size[p] = sink.overflow ? ~0ULL : sink.n;
}
