// rc_dev.hpp — device-side interval arithmetic of the 64-bit carry-less range coder (sub_rc.h:44-212) and the
// interleaved "triple" layout that feeds it.  Shared by the quality and DNA coders.
//
// A part (= one range-coder restart, entr_qual.h:68-79 / entr_read.h:69-77) is coded by ONE lane: the
// recurrence on (low, range) is a dependent chain.  64 consecutive parts form a group handled by one
// wavefront; the triples of a group are stored interleaved — index = group_base + pos * 64 + lane (trip_slot below) — so that
// every step of the wavefront is one coalesced 512-byte load.
#pragma once
#include "common.hpp"
#include <type_traits>

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
// (Computing the reciprocal instead — two-digit long division with a double-precision reciprocal, exact for all 2 097 151 totals — was
// measured in round 4: the coder 7.3 -> 8.7 s per pass, the pass unchanged; DESIGN.md.  The table stays.)
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

// Where a part's triples live.  A group = 64 parts coded by one wave, interleaved symbol by symbol: slot = group_base + pos * 64 + place —
// every step of the coder is one coalesced 512-byte load.  The price is paid by the model kernels, which write the triples in CONTEXT order:
// every 8-byte store hits a sector whose other slots belong to other parts (PMC: 32 GB written per launch of k_evolve_small for 8.4 GB of
// triples).  TRIP_RUN > 1 keeps RUNS of that many consecutive symbols of a part together instead (a 64-byte sector for 8), the 64 parts' runs
// side by side: slot = group_base + (pos / RUN) * 64 * RUN + place * RUN + pos % RUN.  Measured in round 5 with RUN = 8 (-DCL_TRIP_RUN=8, the
// two builds interleaved at 50 Gbases): the model kernels write exactly as many bytes as before (PMC 41.0 / 25.4 / 11.9 GB per launch of
// k_evolve_small / k_dna_evolve / k_long_apply: symbols of one context that are neighbours in the stream are too rare to share sectors), the
// coder takes 7 % longer, the pass is the same — so the default stays 1.
#ifndef CL_TRIP_RUN
#define CL_TRIP_RUN 1
#endif
constexpr uint32_t TRIP_RUN = CL_TRIP_RUN;
__host__ __device__ inline uint64_t trip_slot(uint64_t group_base, uint32_t place, uint64_t pos) { return group_base + (pos / TRIP_RUN) * (64ull * TRIP_RUN) + (uint64_t)place * TRIP_RUN + pos % TRIP_RUN; }
__host__ __device__ inline uint64_t trip_group_words(uint64_t longest_part) { return (longest_part + TRIP_RUN - 1) / TRIP_RUN * TRIP_RUN * 64; }   // triples of a group whose longest part has that many symbols
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
	return (uint32_t)trip_slot(L.group_base[pl >> 6], pl & 63, stream_pos - L.part_sym_start[part]);
}

// Output bytes of one part.  The coder emits the top byte of `low` on every renormalisation step and only shifts `low` in
// between, so the n bytes of one symbol are simply the top n bytes of `low` as it was before the first shift: the whole
// big-endian word is stored (unaligned) at the part's write position and the position advances by n — the bytes behind
// them are overwritten by the next symbol's store, and the last store of a part (End(): 8 bytes of low) is exact, so no store
// ever reaches beyond the part's final size.
struct __attribute__((packed)) unaligned_u64 { uint64_t v; };
__device__ inline void store_be64(uint8_t* p, uint64_t v) { ((unaligned_u64*)p)->v = __builtin_bswap64(v); }

// The bytes on their way out.  A lane that stores its own bytes issues one 8-byte write request per symbol, 64 scattered requests per
// step of the wave; alone that costs nothing, but next to kernels that fill the memory system's write queues with scattered writes
// (radix scatters, table inserts) the coder waits for its own stores — tools/rc_interference.py: 13 ms alone, 172 ms beside a random
// 8-byte scatter, 77 ms in the pipeline.  STAGED: every lane owns a 256-byte ring in LDS (RING_STRIDE apart: an odd
// number of words, so that the lanes' stores spread over the banks), written with the same whole-word trick, and whenever a lane has
// 128 bytes ready half a wave writes them out as ONE contiguous 128-byte store: two orders of magnitude fewer write requests.
// [8 bytes before | 256-byte ring | 8 bytes behind]: a word that runs over the ring's end is stored a second time 256 bytes lower.
constexpr uint32_t RING = 256, RING_UNIT = 128, RING_STRIDE = 8 + RING + 8 + 4;
typedef uint64_t __attribute__((may_alias, aligned(1))) u64_any;          // (the ring is written as words at any byte and read as words and bytes)
typedef uint32_t __attribute__((may_alias, aligned(1))) u32_any;
__device__ inline void lds_store_be64(uint8_t* p, uint64_t v) { *(u64_any*)p = __builtin_bswap64(v); }

// one lane per part, one wave per group of 64 parts (sub_rc.h:72-100,203-210)
static __global__ __launch_bounds__(64) void k_range_code(const triple_t* __restrict__ trip, const uint64_t* __restrict__ group_base,
                                                         const uint32_t* __restrict__ part_len, uint32_t n_parts,
                                                         uint8_t* __restrict__ out, const uint64_t* __restrict__ part_out_off, uint64_t* __restrict__ part_size,
                                                         const uint64_t* __restrict__ inv_tab)
{
	__builtin_amdgcn_s_setprio(3);                                          // a launch of this kernel lasts as long as its slowest chain: its waves go first on their SIMDs
	const uint32_t p = blockIdx.x * 64 + threadIdx.x;
	const bool live = p < n_parts;
	const uint64_t MASK = 0xff00000000000000ULL;                            // (TOP = 0x00ffffffffffff = 2^48 - 1 appears below as its 32-bit halves)
	uint64_t low = 0, range = MASK;
	const uint64_t out_off = live ? part_out_off[p] : 0;
	uint8_t* outp = out + out_off;
	const uint64_t cap = live ? part_out_off[p + 1] - part_out_off[p] : 0;
	uint32_t n_out = 0; bool overflow = cap < 8 || cap > 0xfffffff0ull;
	__shared__ __attribute__((aligned(16))) uint8_t s_ring[64 * RING_STRIDE];
	uint8_t* const ring = s_ring + threadIdx.x * RING_STRIDE + 8;
	const uint32_t room = overflow ? 0u : (uint32_t)cap;                      // bytes this lane may write
	uint32_t flushed = 0;                                                    // bytes of the ring already written out (a multiple of RING_UNIT)
	// write out every lane's complete 128-byte units (called once per round: a round adds at most 8 U = 64 bytes per lane, so a ring
	// holds at most 127 + 64 bytes and the 8-byte stores never reach back into bytes not yet written out)
	auto drain = [&]()
	{
		static_assert(RING_UNIT - 1 + 8 * 8 + 8 <= RING, "a round's bytes must fit behind an incomplete unit");
		uint64_t m;
		while ((m = __ballot(n_out - flushed >= RING_UNIT)) != 0)
		{
			const int L = __builtin_ctzll(m);
			const uint32_t fl = __builtin_amdgcn_readlane(flushed, L), rm = __builtin_amdgcn_readlane(room, L);
			const uint64_t o = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(out_off >> 32), L) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)out_off, L);   // (readlane returns int)
			if (fl + RING_UNIT <= rm && threadIdx.x < RING_UNIT / 4)
			{
				const uint32_t w = *(const u32_any*)(s_ring + L * RING_STRIDE + 8 + (fl & (RING - 1)) + threadIdx.x * 4);
				*(u32_any*)(out + o + fl + threadIdx.x * 4) = w;
			}
			if ((int)threadIdx.x == L) flushed += RING_UNIT;                      // (a unit that does not fit is dropped: the part is reported as overflowing at the end)
		}
	};
	const uint32_t len = live ? part_len[p] : 0;
	uint32_t lmax = len;
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) { uint32_t t = __shfl_xor(lmax, d, 64); lmax = t > lmax ? t : lmax; }
	if (lmax == 0) { if (live) { if (!overflow) store_be64(outp, 0); part_size[p] = overflow ? ~0ULL : 8; } return; }
	const triple_t* src = trip + group_base[blockIdx.x] + threadIdx.x * TRIP_RUN;   // this lane's runs: symbol q at src[at(q)]
	auto at = [](uint32_t q) -> uint64_t { return (uint64_t)(q / TRIP_RUN) * (64 * TRIP_RUN) + q % TRIP_RUN; };
	constexpr uint32_t U = 8;
	static_assert(U % TRIP_RUN == 0, "a round of the coder takes whole runs of a lane's triples");
	// three stages ahead of the chain: symbols two rounds ahead, their reciprocals one round ahead (looked up from the symbols
	// fetched the round before), the round being coded.  The three register sets trade roles from round to round (no copies).
	triple_t A[U], B[U], C[U]; uint64_t iA[U], iB[U], iC[U];
	// a lane whose part is shorter than the group's longest keeps stepping with the neutral symbol
	// (cum 0, freq 1, total 1): range / 1 * 1 and low + 0 leave the coder untouched.
	const uint64_t NEUTRAL_X = (1ULL << 21) | 1ULL, NEUTRAL_Y = ~0ULL;
	const uint32_t last = lmax - 1;
#pragma unroll
	for (uint32_t u = 0; u < U; ++u) { A[u] = src[at(u < last ? u : last)]; B[u] = src[at(U + u < last ? U + u : last)]; }
#pragma unroll
	for (uint32_t u = 0; u < U; ++u) iA[u] = inv_tab[A[u] & 0x1fffff];
	// one round: fetch `far` (two rounds ahead), look up the reciprocals of `nxt`, code `cur` with `icur`
	// (all_active: every lane of the wave still has symbols in this round — no neutral symbols to select; decided per round, wave-uniform)
	auto round = [&](auto all_active, uint32_t pos, const triple_t (&cur)[U], const uint64_t (&icur)[U], const triple_t (&nxt)[U], uint64_t (&inxt)[U], triple_t (&far)[U])
	{
		{	// prefetch: the lane's run two rounds ahead — whole (four 16-byte loads) while it lies inside the group's longest part, else symbol by
			// symbol with the index clamped (never a pointer select)
			const uint32_t q0 = pos + 2 * U;                                          // (a multiple of TRIP_RUN: rounds start at multiples of U)
			if (TRIP_RUN == U && q0 + U - 1 <= last)
			{
				const ulonglong2* v = (const ulonglong2*)(src + at(q0));
#pragma unroll
				for (uint32_t u = 0; u < U; u += 2) { const ulonglong2 w = v[u / 2]; far[u] = w.x; far[u + 1] = w.y; }
			}
			else
			{
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) { const uint32_t q = q0 + u; far[u] = src[at(q < last ? q : last)]; }
			}
		}
#pragma unroll
		for (uint32_t u = 0; u < U; ++u) inxt[u] = inv_tab[nxt[u] & 0x1fffff];
#pragma unroll
		for (uint32_t u = 0; u < U; ++u)
		{
			const bool act = decltype(all_active)::value || pos + u < len;
			const uint64_t tx = act ? cur[u] : NEUTRAL_X, inv = act ? icur[u] : NEUTRAL_Y;
			const uint32_t tot = (uint32_t)(tx & 0x1fffff), freq = (uint32_t)((tx >> 21) & 0x1fffff), cum = (uint32_t)(tx >> 42);
			// range / tot.  With inv = floor((2^64-1) / tot) = (2^64 - 1 - rho) / tot, 0 <= rho < tot:
			//   range inv / 2^64 = range / tot - (range / 2^64) (1 + rho) / tot > range / tot - 1,
			// so the high product is the quotient or one below it, the remainder it leaves is below 2 tot < 2^22 and 32-bit
			// arithmetic finds it exactly.
			uint64_t q = __umul64hi(range, inv);
			const uint32_t r = (uint32_t)range - (uint32_t)q * tot;
			q += (uint32_t)(r >= tot);
			low += q * cum;
			range = q * freq;
			// renormalisation (sub_rc.h:72-100): while range <= TOP, give out the top byte of low; a step whose interval straddles a
			// top-byte boundary first cuts the range at it.  Every lane takes the steps of the lane that needs most (predicated);
			// after 8 steps nothing of low is left: a range that is still empty then is 0 (corrupt triples) and stays 0 — seen at the end.
			const uint64_t low0 = low;
			uint32_t nb = 0;
			uint32_t rh = (uint32_t)(range >> 32), rl = (uint32_t)range, lh = (uint32_t)(low >> 32), ll = (uint32_t)low;
			asm volatile("" : "+v"(rh), "+v"(rl), "+v"(lh), "+v"(ll));                 // (keeps low + range out of the multiply-adds above)
			auto step = [&]()
			{	// (32-bit halves: 64-bit shifts and compares are slow instructions)
				const bool need = rh < 0x00010000u;                                      // range <= TOP = 2^48 - 1
				const uint32_t sh = lh + rh + (uint32_t)(rl > ~ll);                      // high half of low + range
				const bool straddle = (lh ^ sh) > 0x00ffffffu;                           // (low ^ (low + range)) & MASK
				const uint32_t fh = straddle ? (~lh & 0x0000ffffu) : rh, fl = straddle ? ~ll : rl;   // (low | TOP) - low
				rh = need ? __builtin_amdgcn_alignbit(fh, fl, 24) : rh; rl = need ? fl << 8 : rl;
				lh = need ? __builtin_amdgcn_alignbit(lh, ll, 24) : lh; ll = need ? ll << 8 : ll;
				nb += need ? 1u : 0u;
			};
			step();                                                                      // (nearly every symbol: some lane of the wave needs one)
			if (__any(rh < 0x00010000u))
			{
#pragma unroll 1
				for (uint32_t it = 1; it < 8; ++it) { step(); if (!__any(rh < 0x00010000u)) break; }
			}
			range = ((uint64_t)rh << 32) | rl; low = ((uint64_t)lh << 32) | ll;
			// the bytes: the whole word at the write position (held inside the part's room; a part that outgrows it is seen at the end)
			if (nb) { const uint32_t rp = n_out & (RING - 1); lds_store_be64(ring + rp, low0); if (rp > RING - 8) lds_store_be64(ring + rp - RING, low0); }
			n_out += nb;
		}
		drain();
	};
	uint32_t lmin = live ? len : 0u;
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) { uint32_t t = __shfl_xor(lmin, d, 64); lmin = t < lmin ? t : lmin; }
	const std::true_type ALL{}; const std::false_type SOME{};
	uint32_t pos = 0;
	for (; pos + 3 * U <= lmin; pos += 3 * U)                                  // every lane active for three whole rounds
	{
		round(ALL, pos, A, iA, B, iB, C);
		round(ALL, pos + U, B, iB, C, iC, A);
		round(ALL, pos + 2 * U, C, iC, A, iA, B);
	}
	for (; pos < lmax; pos += 3 * U)
	{
		round(SOME, pos, A, iA, B, iB, C);
		if (pos + U >= lmax) break;
		round(SOME, pos + U, B, iB, C, iC, A);
		if (pos + 2 * U >= lmax) break;
		round(SOME, pos + 2 * U, C, iC, A, iA, B);
	}
	{
		// End(): 8 bytes of low (sub_rc.h:203-210) go through the ring as well; then every lane's remainder, byte by byte, by the whole wave
		{ const uint32_t rp = n_out & (RING - 1); lds_store_be64(ring + rp, low); if (rp > RING - 8) lds_store_be64(ring + rp - RING, low); }
		n_out += 8;
		if (!live || n_out > room || (uint32_t)(range >> 32) < 0x00010000u) overflow = true;   // (an empty range: corrupt triples)
		drain();
		const uint32_t rem = overflow ? 0u : n_out - flushed;
		for (int L = 0; L < 64; ++L)
		{
			const uint32_t n = __builtin_amdgcn_readlane(rem, L), fl = __builtin_amdgcn_readlane(flushed, L);
			if (n == 0) continue;
			const uint64_t o = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(out_off >> 32), L) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)out_off, L);   // (readlane returns int)
			for (uint32_t b = threadIdx.x; b < n; b += 64) out[o + fl + b] = s_ring[L * RING_STRIDE + 8 + ((fl + b) & (RING - 1))];
		}
		if (live) part_size[p] = overflow ? ~0ULL : n_out;
	}
}

#define LAUNCH_RANGE_CODE(ctx, bytes, ng, ...) LAUNCHB_NAMED(ctx, "k_range_code", bytes, k_range_code, ng, 64, __VA_ARGS__)
