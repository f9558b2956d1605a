// sort.hip — in-tree stable LSD radix sort (8- to 10-bit digits) for uint32/uint64 keys with an optional
// uint32 payload.  Per pass: per-tile digit histogram -> device-wide exclusive scan (digit-major) ->
// stable scatter.  Ranking inside a wavefront uses 64-lane ballots ("match-any" over the 8 digit bits)
// so equal digits keep their input order without a second local sort.
//
// HBM roofline per pass: 2 key reads + 1 key write (+ payload read/write); n must be < 2^32.
#include "common.hpp"
#include <string>

namespace {
constexpr uint32_t ST = 256;            // threads per block
constexpr uint32_t SI = 16;             // keys per thread
constexpr uint32_t STILE = ST * SI;     // 4096 keys per tile

// DB = digit bits of the pass (8, 9 or 10): the passes of a sort share the key bits evenly, so 17 bits take two passes (9 + 8),
// not three, and 20 bits two of 10.
// G consecutive tiles per block: a digit's G counts are neighbours in the digit-major table and leave as one 16- or 64-byte run.  (One tile
// per block wrote every count into a sector of its own: 275 MB of counts cost 1.7 GB of HBM writes per pass of the DNA coder's sort by the
// round's first WRITE_SIZE pass; 0.39 -> 0.09 GB per launch on average since, profiles/r05_pmc_traffic_summary.txt.)  G = 1 for the small sorts, whose few tiles are wanted on as many CUs as there are tiles.
template<typename K, uint32_t DB, uint32_t G>
__global__ __launch_bounds__(ST) void k_sort_hist(const K* __restrict__ keys, uint64_t n, uint32_t shift,
                                                  uint32_t* __restrict__ hist, uint32_t nb)
{
	constexpr uint32_t ND = 1u << DB;
	constexpr uint32_t ROW = G > 1 ? ND + 1 : ND;        // (a tile's counters side by side as with one tile per block; the odd row length spreads the
	__shared__ uint32_t h[G * ROW];                      // transposed read-out below over the banks)  [tile of the block][digit]
	for (uint32_t i = threadIdx.x; i < G * ROW; i += ST) h[i] = 0;
	__syncthreads();
	const uint32_t tile0 = blockIdx.x * G;
	for (uint32_t g = 0; g < G && tile0 + g < nb; ++g)
	{
		const uint64_t base = (uint64_t)(tile0 + g) * STILE;
#pragma unroll
		for (uint32_t i = 0; i < SI; ++i)
		{
			uint64_t idx = base + (uint64_t)i * ST + threadIdx.x;
			if (idx < n) atomicAdd(&h[g * ROW + ((uint32_t)(keys[idx] >> shift) & (ND - 1))], 1u);
		}
	}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < G * ND; i += ST)
	{
		const uint32_t g = i % G, d = i / G;
		if (tile0 + g < nb) hist[(uint64_t)d * nb + tile0 + g] = h[g * ROW + d];
	}
}

template<typename K, bool HAS_V, uint32_t DB>
__global__ __launch_bounds__(ST) void k_sort_scatter(const K* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                     K* __restrict__ kout, uint32_t* __restrict__ vout,
                                                     uint64_t n, uint32_t shift, const uint32_t* __restrict__ offs, uint32_t nb)
{
	// The tile is first ordered by digit in LDS (stable: wave by wave, ballot ranks inside a wave), then written out
	// linearly: consecutive threads write consecutive elements of a bucket, so a bucket's share of the tile (16 keys on
	// average) leaves as whole lines instead of one 4/8-byte store per lane and bucket.
	constexpr uint32_t ND = 1u << DB, PER = ND / ST;   // digits; digits per thread in the scan (1, 2, 4)
	__shared__ uint32_t wh[4][ND];           // per wave and digit: count, then the wave's first place in the ordered tile
	__shared__ uint32_t lstart[ND + 1];      // first place of a digit in the ordered tile
	__shared__ uint32_t gdelta[ND];          // global position of a digit's first tile element minus lstart
	__shared__ uint32_t wtot[4];
	__shared__ K skey[STILE];
	__shared__ uint32_t sval[HAS_V ? STILE : 1];
	const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
	for (uint32_t i = threadIdx.x; i < 4 * ND; i += ST) (&wh[0][0])[i] = 0;
	__syncthreads();

	const uint64_t tbase = (uint64_t)blockIdx.x * STILE, wbase = tbase + (uint64_t)w * (64 * SI);
	K key[SI]; uint32_t rank[SI]; uint32_t val[HAS_V ? SI : 1];
	const uint64_t lt = (1ULL << lane) - 1;
	uint32_t goff[PER];                                                      // the tile's places in the output, asked for with the keys (they are one
#pragma unroll                                                                // scattered word per digit: a round trip of their own if read where they are used)
	for (uint32_t k = 0; k < PER; ++k) goff[k] = offs[(uint64_t)(threadIdx.x * PER + k) * nb + blockIdx.x];
#pragma unroll
	for (uint32_t r = 0; r < SI; ++r)
	{
		uint64_t idx = wbase + (uint64_t)r * 64 + lane;
		key[r] = idx < n ? kin[idx] : (K)0;
		if (HAS_V) val[r] = idx < n ? vin[idx] : 0u;                           // (with the keys: not a second round trip later)
	}
#pragma unroll
	for (uint32_t r = 0; r < SI; ++r)
	{
		uint64_t idx = wbase + (uint64_t)r * 64 + lane;
		bool valid = idx < n;
		uint32_t d = (uint32_t)(key[r] >> shift) & (ND - 1);
		uint64_t peers = __ballot(valid);
#pragma unroll
		for (uint32_t b = 0; b < DB; ++b)
		{
			bool bit = (d >> b) & 1u;
			uint64_t m = __ballot(valid && bit);
			peers &= bit ? m : ~m;
		}
		uint32_t prior = valid ? wh[w][d] : 0u;
		rank[r] = prior + (uint32_t)__popcll(peers & lt);
		__builtin_amdgcn_wave_barrier();
		if (valid && (peers & lt) == 0) wh[w][d] = prior + (uint32_t)__popcll(peers);   // lowest peer lane
		__builtin_amdgcn_wave_barrier();
	}
	__syncthreads();
	{	// thread t: digits t * PER .. t * PER + PER - 1: totals -> exclusive scan over the digits -> places of the waves' shares
		uint32_t c[PER][4], tot[PER], mine = 0;
#pragma unroll
		for (uint32_t k = 0; k < PER; ++k)
		{
			const uint32_t d = threadIdx.x * PER + k;
			c[k][0] = wh[0][d]; c[k][1] = wh[1][d]; c[k][2] = wh[2][d]; c[k][3] = wh[3][d];
			tot[k] = c[k][0] + c[k][1] + c[k][2] + c[k][3]; mine += tot[k];
		}
		uint32_t incl = mine;
#pragma unroll
		for (uint32_t o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
		if (lane == 63) wtot[w] = incl;
		__syncthreads();
		uint32_t ls = incl - mine;
		for (uint32_t i = 0; i < w; ++i) ls += wtot[i];
#pragma unroll
		for (uint32_t k = 0; k < PER; ++k)
		{
			const uint32_t d = threadIdx.x * PER + k;
			lstart[d] = ls;
			if (d == ND - 1) lstart[ND] = ls + tot[k];
			gdelta[d] = goff[k] - ls;
			wh[0][d] = ls; wh[1][d] = ls + c[k][0]; wh[2][d] = ls + c[k][0] + c[k][1]; wh[3][d] = ls + c[k][0] + c[k][1] + c[k][2];
			ls += tot[k];
		}
	}
	__syncthreads();
#pragma unroll
	for (uint32_t r = 0; r < SI; ++r)
	{
		uint64_t idx = wbase + (uint64_t)r * 64 + lane;
		if (idx < n)
		{
			uint32_t d = (uint32_t)(key[r] >> shift) & (ND - 1);
			uint32_t lp = wh[w][d] + rank[r];
			skey[lp] = key[r];
			if (HAS_V) sval[lp] = val[r];
		}
	}
	__syncthreads();
	const uint32_t tile_n = lstart[ND];
	for (uint32_t j = threadIdx.x; j < tile_n; j += ST)
	{
		const K k = skey[j];
		const uint32_t pos = gdelta[(uint32_t)(k >> shift) & (ND - 1)] + j;
		kout[pos] = k;
		if (HAS_V) vout[pos] = sval[j];
	}
}

template<typename K, uint32_t DB>
cl_status sort_pass(cl_ctx* ctx, const K* kin, const uint32_t* vin, K* kout, uint32_t* vout, uint64_t n, uint32_t shift, uint32_t* hist, uint32_t nb)
{
	// (names as rocprofv3 prints the instantiations)
	static const std::string kt = sizeof(K) == 8 ? "unsigned long" : "unsigned int", db = std::to_string(DB) + "u";
	// (group sizes measured inside the pipeline at 10 Gbases, summed histogram time per pass with G = 1 / 4 / 16: 8-bit digits 174 / 122 / 102 ms,
	// 9-bit 185 / 139 / 518 ms — 16 tiles of 512 counters are 32 KB of LDS and halve the blocks a CU holds)
	constexpr uint32_t G = DB == 8 ? 16 : 4; const bool grouped = nb >= 4096;           // (16 M keys and more)
	static const std::string n_hist1 = "k_sort_hist<" + kt + ", " + db + ", 1u>", n_histg = "k_sort_hist<" + kt + ", " + db + ", " + std::to_string(G) + "u>", n_sv = "k_sort_scatter<" + kt + ", true, " + db + ">", n_sk = "k_sort_scatter<" + kt + ", false, " + db + ">";
	if (grouped) LAUNCHB_NAMED(ctx, n_histg.c_str(), n * sizeof(K), (k_sort_hist<K, DB, G>), (nb + G - 1) / G, ST, kin, n, shift, hist, nb);
	else LAUNCHB_NAMED(ctx, n_hist1.c_str(), n * sizeof(K), (k_sort_hist<K, DB, 1>), nb, ST, kin, n, shift, hist, nb);
	HIP_TRY(ctx, hipGetLastError());
	CL_TRY(dev_exclusive_scan_u32(ctx, hist, (uint64_t)(1u << DB) * nb, nullptr));
	if (vin) LAUNCHB_NAMED(ctx, n_sv.c_str(), n * (2 * sizeof(K) + 8), (k_sort_scatter<K, true, DB>), nb, ST, kin, vin, kout, vout, n, shift, (const uint32_t*)hist, nb);
	else LAUNCHB_NAMED(ctx, n_sk.c_str(), n * 2 * sizeof(K), (k_sort_scatter<K, false, DB>), nb, ST, kin, (const uint32_t*)nullptr, kout, (uint32_t*)nullptr, n, shift, (const uint32_t*)hist, nb);
	HIP_TRY(ctx, hipGetLastError());
	return CL_OK;
}

// swap_k / swap_v (optional): the caller's arrays as buffers of exactly n elements — when an odd number of passes leaves the result in
// the temporaries, the buffers are SWAPPED with them instead of copying n elements back (the DNA coder's sort: 14 GB per 1-Gbase chunk
// read and written again for nothing, the largest share of the 3.5 s of copy kernels per pass)
template<typename K>
cl_status sort_impl(cl_ctx* ctx, K* d_keys, uint32_t* d_vals, uint64_t n, uint32_t begin_bit, uint32_t end_bit, DevBuf<K>* swap_k = nullptr, DevBuf<uint32_t>* swap_v = nullptr)
{
	if (n <= 1 || end_bit <= begin_bit) return CL_OK;
	if (n >= (1ULL << 32)) return cl_fail(ctx, CL_E_UNSUPPORTED, "radix sort: n must be < 2^32 per call");
	const uint32_t nb = grid_for(n, STILE);
	// passes of at most 10 bits each, the bits shared evenly (8-bit digits when that takes no more passes)
	const uint32_t bits = end_bit - begin_bit;
	const uint32_t n_pass_real = (bits + 7) / 8 == (bits + 9) / 10 ? (bits + 7) / 8 : (bits + 9) / 10;
	const uint32_t db_hi = (bits + n_pass_real - 1) / n_pass_real;               // digit bits of the first passes (8, 9 or 10; 1..8 when one pass does it)
	// ping-pong between the caller's arrays and a temporary; with an ODD number of passes a second temporary takes the
	// first pass, so that the last one lands in the caller's arrays without a copy back
	// (the second temporary only while it is small: on the largest sorts its footprint costs more than the copy)
	const bool third = (n_pass_real & 1) && n_pass_real > 1 && n * (sizeof(K) + (d_vals ? 4 : 0)) <= (2ull << 30);
	const uint32_t n_pass = third ? n_pass_real : (n_pass_real + 1) & ~1u;          // even: plain ping-pong (+ copy back if the real count is odd)
	DevBuf<K> ktmp, ktmp2; DEV_ALLOC(ctx, ktmp, n);
	DevBuf<uint32_t> vtmp, vtmp2; if (d_vals) DEV_ALLOC(ctx, vtmp, n);
	if (n_pass & 1) { DEV_ALLOC(ctx, ktmp2, n); if (d_vals) DEV_ALLOC(ctx, vtmp2, n); }
	DevBuf<uint32_t> hist; DEV_ALLOC(ctx, hist, (uint64_t)(1u << std::max(8u, db_hi)) * nb);
	K* kin = d_keys; uint32_t* vin = d_vals;
	uint32_t pass = 0;
	for (uint32_t shift = begin_bit; shift < end_bit; ++pass)
	{
		// destinations: even count: tmp, keys, tmp, keys ...; odd count: tmp2, tmp, keys, tmp, keys ...
		K* kout; uint32_t* vout;
		if (n_pass & 1) { kout = pass == 0 ? ktmp2.p : (pass & 1) ? ktmp.p : d_keys; vout = pass == 0 ? vtmp2.p : (pass & 1) ? vtmp.p : d_vals; }
		else { kout = (pass & 1) ? d_keys : ktmp.p; vout = (pass & 1) ? d_vals : vtmp.p; }
		const uint32_t left = end_bit - shift, passes_left = n_pass_real - pass;
		const uint32_t db = std::max(8u, (left + passes_left - 1) / passes_left); // (bits above end_bit in a digit are harmless only if they are equal: callers' keys are zero there or sorted on them anyway)
		const uint32_t use = std::min(db, 10u);
		if (use == 10) CL_TRY((sort_pass<K, 10>(ctx, kin, vin, kout, vout, n, shift, hist.p, nb)));
		else if (use == 9) CL_TRY((sort_pass<K, 9>(ctx, kin, vin, kout, vout, n, shift, hist.p, nb)));
		else CL_TRY((sort_pass<K, 8>(ctx, kin, vin, kout, vout, n, shift, hist.p, nb)));
		shift += use;
		kin = kout; vin = vout;
	}
	if (kin != d_keys && swap_k && kin == ktmp.p && (!d_vals || (swap_v && vin == vtmp.p)))
	{
		std::swap(*swap_k, ktmp);                                               // (same sizes: both were made for n elements)
		if (d_vals) std::swap(*swap_v, vtmp);
	}
	else if (kin != d_keys)
	{
		HIP_TRY(ctx, hipMemcpyAsync(d_keys, kin, n * sizeof(K), hipMemcpyDeviceToDevice, cl_launch_stream(ctx)));
		if (d_vals) HIP_TRY(ctx, hipMemcpyAsync(d_vals, vin, n * 4, hipMemcpyDeviceToDevice, cl_launch_stream(ctx)));
	}
	HIP_TRY(ctx, hipStreamSynchronize(cl_launch_stream(ctx)));                  // (the stream the passes ran on: the temporaries go back to the pool on return)
	return CL_OK;
}
} // namespace

cl_status dev_sort_pairs(cl_ctx* ctx, uint64_t* d_keys, uint32_t* d_vals, uint64_t n, uint32_t begin_bit, uint32_t end_bit)
{
	return sort_impl<uint64_t>(ctx, d_keys, d_vals, n, begin_bit, end_bit);
}
cl_status dev_sort_pairs_swap(cl_ctx* ctx, DevBuf<uint64_t>& keys, DevBuf<uint32_t>& vals, uint64_t n, uint32_t begin_bit, uint32_t end_bit)
{
	if (keys.n != n || vals.n != n) return sort_impl<uint64_t>(ctx, keys.p, vals.p, n, begin_bit, end_bit);
	return sort_impl<uint64_t>(ctx, keys.p, vals.p, n, begin_bit, end_bit, &keys, &vals);
}
cl_status dev_sort_keys32_pairs(cl_ctx* ctx, uint32_t* d_keys, uint32_t* d_vals, uint64_t n, uint32_t begin_bit, uint32_t end_bit)
{
	return sort_impl<uint32_t>(ctx, d_keys, d_vals, n, begin_bit, end_bit);
}
cl_status dev_sort_keys32_pairs_swap(cl_ctx* ctx, DevBuf<uint32_t>& keys, DevBuf<uint32_t>& vals, uint64_t n, uint32_t begin_bit, uint32_t end_bit)
{
	if (keys.n != n || vals.n != n) return sort_impl<uint32_t>(ctx, keys.p, vals.p, n, begin_bit, end_bit);
	return sort_impl<uint32_t>(ctx, keys.p, vals.p, n, begin_bit, end_bit, &keys, &vals);
}
