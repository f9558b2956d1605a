// scan.hip — device-wide exclusive prefix sums (tile reduce -> recursive scan of tile sums -> tile scan).
// HBM-bound helpers: 2 reads + 1 write of the array per call.
#include "common.hpp"

namespace {
constexpr uint32_t SCAN_THREADS = 256;
constexpr uint32_t SCAN_ITEMS = 8;
constexpr uint32_t SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

template<typename TOut> __device__ inline TOut wave_incl_scan_t(TOut v)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { TOut t = __shfl_up(v, d, 64); if ((int)(threadIdx.x & 63) >= d) v += t; }
	return v;
}

template<typename TIn, typename TOut>
__global__ __launch_bounds__(SCAN_THREADS) void k_tile_reduce(const TIn* __restrict__ in, uint64_t n, TOut* __restrict__ tile_sums)
{
	__shared__ TOut sh[4];
	uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
	TOut s = 0;
#pragma unroll
	for (uint32_t i = 0; i < SCAN_ITEMS; ++i)
	{
		uint64_t idx = base + (uint64_t)i * SCAN_THREADS + threadIdx.x;
		if (idx < n) s += (TOut)in[idx];
	}
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) tile_sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// out[idx] = tile_off[tile] + exclusive prefix inside the tile.  Thread t owns SCAN_ITEMS consecutive
// elements so that the order is the array order.  in may alias out when TIn == TOut.
template<typename TIn, typename TOut>
__global__ __launch_bounds__(SCAN_THREADS) void k_tile_scan(const TIn* in, uint64_t n, const TOut* __restrict__ tile_off, TOut* out)
{
	__shared__ TOut sh[4];
	uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
	TOut v[SCAN_ITEMS]; TOut s = 0;
#pragma unroll
	for (uint32_t i = 0; i < SCAN_ITEMS; ++i) { v[i] = (base + i < n) ? (TOut)in[base + i] : (TOut)0; s += v[i]; }
	TOut incl = wave_incl_scan_t<TOut>(s);
	uint32_t w = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 63) sh[w] = incl;
	__syncthreads();
	TOut pre = (tile_off ? tile_off[blockIdx.x] : (TOut)0) + incl - s;
	for (uint32_t i = 0; i < w; ++i) pre += sh[i];
#pragma unroll
	for (uint32_t i = 0; i < SCAN_ITEMS; ++i) { if (base + i < n) out[base + i] = pre; pre += v[i]; }
}

template<typename T> __global__ void k_write_total(const T* last_in_scanned, T last_value, T* dst) { *dst = *last_in_scanned + last_value; }

template<typename TIn, typename TOut>
cl_status scan_impl(cl_ctx* ctx, const TIn* d_in, TOut* d_out, uint64_t n)
{
	if (n == 0) return CL_OK;
	uint32_t tiles = grid_for(n, SCAN_TILE);
	if (tiles == 1)
	{
		LAUNCH(ctx, (k_tile_scan<TIn, TOut>), 1, SCAN_THREADS, d_in, n, (const TOut*)nullptr, d_out);
		HIP_TRY(ctx, hipGetLastError());
		return CL_OK;
	}
	DevBuf<TOut> sums; DEV_ALLOC(ctx, sums, tiles);
	LAUNCH(ctx, (k_tile_reduce<TIn, TOut>), tiles, SCAN_THREADS, d_in, n, sums.p);
	HIP_TRY(ctx, hipGetLastError());
	CL_TRY((scan_impl<TOut, TOut>(ctx, sums.p, sums.p, tiles)));
	LAUNCH(ctx, (k_tile_scan<TIn, TOut>), tiles, SCAN_THREADS, d_in, n, (const TOut*)sums.p, d_out);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // sums is freed on return
	return CL_OK;
}
} // namespace

// In-place exclusive scan of n uint32 (sums must fit 32 bits); *h_total (optional) = sum of all.
cl_status dev_exclusive_scan_u32(cl_ctx* ctx, uint32_t* d_data, uint64_t n, uint64_t* h_total)
{
	uint32_t last = 0, last_scanned = 0;
	if (h_total && n) HIP_TRY(ctx, hipMemcpyAsync(&last, d_data + n - 1, 4, hipMemcpyDeviceToHost, ctx->stream));
	CL_TRY((scan_impl<uint32_t, uint32_t>(ctx, d_data, d_data, n)));
	if (h_total)
	{
		if (n) HIP_TRY(ctx, hipMemcpyAsync(&last_scanned, d_data + n - 1, 4, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		*h_total = (uint64_t)last + last_scanned;
	}
	return CL_OK;
}

// d_out[0..n] = exclusive scan of d_in[0..n) widened to 64 bits, d_out[n] = total.
cl_status dev_exclusive_scan_u64(cl_ctx* ctx, const uint32_t* d_in, uint64_t* d_out, uint64_t n, uint64_t* h_total)
{
	uint64_t total = 0;
	if (n)
	{
		CL_TRY((scan_impl<uint32_t, uint64_t>(ctx, d_in, d_out, n)));
		uint32_t last = 0; uint64_t last_scanned = 0;
		HIP_TRY(ctx, hipMemcpyAsync(&last, d_in + n - 1, 4, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(&last_scanned, d_out + n - 1, 8, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		total = last_scanned + last;
	}
	HIP_TRY(ctx, hipMemcpyAsync(d_out + n, &total, 8, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	if (h_total) *h_total = total;
	return CL_OK;
}
