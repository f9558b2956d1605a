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

// ---- one pass: decoupled look-back --------------------------------------------------------------------------------------------
// A tile publishes its sum (AGGREGATE), looks back over the tiles before it until it meets one whose inclusive PREFIX is known,
// publishes its own PREFIX and writes its elements: one read and one write of the array, ONE launch (the three-level reduce / scan
// above: two reads, one write, five launches and more for large arrays — 11 000 of the 45 000 dispatches of a 50-Gbase pass).  Tiles are
// handed out by a ticket, so every tile a tile waits for has been started.  The status words carry their value WITH their flag
// (8-byte agent-scope relaxed atomics on both sides: MI355X_MICROARCH.md, "valid forms", data-is-the-flag granules).
constexpr uint32_t LB_ITEMS = 16, LB_TILE = SCAN_THREADS * LB_ITEMS;
constexpr unsigned long long LB_AGG = 1ull << 62, LB_PREFIX = 2ull << 62, LB_VAL = (1ull << 62) - 1;
// ctl[0]: ticket, ctl[1..]: status of tile 0, 1, ... (zeroed before the launch).  total_out (optional): receives the sum of all.
template<typename TIn, typename TOut>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_lookback(const TIn* in, uint64_t n, TOut* out, unsigned long long* __restrict__ ctl, TOut* total_out, unsigned long long* total64)
{
	__shared__ TOut sh[4];
	__shared__ unsigned long long s_excl;
	__shared__ uint32_t s_tile;
	if (threadIdx.x == 0) s_tile = (uint32_t)atomicAdd(ctl, 1ull);
	__syncthreads();
	const uint32_t tile = s_tile, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	unsigned long long* status = ctl + 1;
	const uint64_t base = (uint64_t)tile * LB_TILE + (uint64_t)threadIdx.x * LB_ITEMS;
	TOut v[LB_ITEMS]; TOut s = 0;
#pragma unroll
	for (uint32_t i = 0; i < LB_ITEMS; ++i) { v[i] = (base + i < n) ? (TOut)in[base + i] : (TOut)0; s += v[i]; }
	const TOut incl = wave_incl_scan_t<TOut>(s);
	if (lane == 63) sh[w] = incl;
	__syncthreads();
	TOut pre = incl - s, total = 0;
	for (uint32_t i = 0; i < 4; ++i) { if (i < w) pre += sh[i]; total += sh[i]; }
	if (w == 0)
	{
		unsigned long long excl = 0;
		if (tile == 0) { if (lane == 0) __hip_atomic_store(status, LB_PREFIX | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
		else
		{
			if (lane == 0) __hip_atomic_store(status + tile, LB_AGG | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			// lanes look at tiles hi - 1 - lane; the nearest PREFIX ends the walk, everything nearer is an AGGREGATE (or not there yet: read again)
			for (int64_t hi = tile; hi > 0; )
			{
				const int64_t j = hi - 1 - (int64_t)lane;
				unsigned long long x = j >= 0 ? __hip_atomic_load(status + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : LB_PREFIX;   // (before tile 0: an empty prefix)
				const uint64_t pref = __ballot((x >> 62) == 2);
				const uint32_t stop = pref ? (uint32_t)__builtin_ctzll(pref) : 64u;        // the nearest lane that holds a PREFIX
				const uint64_t need = stop >= 63 ? ~0ull : ((2ull << stop) - 1);
				const uint64_t ready = __ballot((x >> 62) != 0);
				if ((ready & need) != need) { __builtin_amdgcn_s_sleep(2); continue; }
				unsigned long long part = lane <= stop ? (x & LB_VAL) : 0ull;
				for (int o = 32; o; o >>= 1) part += __shfl_xor(part, o);
				excl += part;
				if (pref) break;
				hi -= 64;
			}
			if (lane == 0) __hip_atomic_store(status + tile, LB_PREFIX | (excl + (unsigned long long)total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (lane == 0) s_excl = excl;
	}
	__syncthreads();
	pre += (TOut)s_excl;
#pragma unroll
	for (uint32_t i = 0; i < LB_ITEMS; ++i) { if (base + i < n) out[base + i] = pre; pre += v[i]; }
	// (the last thread of the last tile has walked to the end; the 64-bit total comes from the look-back's own 62-bit sums, so a sum that
	// does not fit TOut is seen by the host instead of wrapping silently)
	if ((uint64_t)(tile + 1) * LB_TILE >= n && threadIdx.x == SCAN_THREADS - 1) { if (total_out) *total_out = pre; if (total64) *total64 = s_excl + (unsigned long long)total; }
}

template<typename TIn, typename TOut>
cl_status scan_lookback(cl_ctx* ctx, const TIn* d_in, TOut* d_out, uint64_t n, TOut* d_total, unsigned long long* d_total64 = nullptr)
{
	const uint32_t tiles = grid_for(n, LB_TILE);
	// The status words live in a buffer the context keeps PER STREAM (round 5).  Rounds 3-4 took them from the pool and gave them back on
	// return, with memset and kernel still queued: the pool hands a context its own blocks back at once, so the next allocation of the same
	// context — made for work on ANOTHER of its streams (a side stream, a LaunchOn scope) — could overwrite the flags under a running scan:
	// a wrong prefix or a look-back that never ends.  Scans on one stream are ordered, so one buffer per stream is safe; it grows rarely, and
	// only after the stream has drained.
	hipStream_t st = cl_launch_stream(ctx);
	cl_ctx::ScanCtl& C = ctx->scan_ctl[(void*)st];
	if (C.words < (uint64_t)tiles + 1)
	{
		if (C.p) { HIP_TRY(ctx, hipStreamSynchronize(st)); ctx->pool.put(C.p, C.got, ctx->pool_id); C.p = nullptr; C.words = 0; }
		const uint64_t want = std::max<uint64_t>((uint64_t)tiles + 1, 1u << 16);
		void* q = nullptr;
		if (ctx->pool.get(want * 8, &q, &C.got, ctx->pool_id) != hipSuccess) return cl_fail(ctx, CL_E_NOMEM, "scan_lookback: no memory for the status words");
		C.p = (unsigned long long*)q; C.words = want;
	}
	HIP_TRY(ctx, hipMemsetAsync(C.p, 0, ((uint64_t)tiles + 1) * 8, st));
	LAUNCH(ctx, (k_scan_lookback<TIn, TOut>), tiles, SCAN_THREADS, d_in, n, d_out, C.p, d_total, d_total64);
	HIP_TRY(ctx, hipGetLastError());
	return CL_OK;
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
	HIP_TRY(ctx, hipStreamSynchronize(cl_launch_stream(ctx)));   // sums is freed on return
	return CL_OK;
}
} // namespace

// In-place exclusive scan of n uint32 (sums must fit 32 bits); *h_total (optional) = sum of all.
cl_status dev_exclusive_scan_u32(cl_ctx* ctx, uint32_t* d_data, uint64_t n, uint64_t* h_total)
{
	if (h_total) *h_total = 0;
	if (!n) return CL_OK;
	hipStream_t st = cl_launch_stream(ctx);
	if (getenv("COLORD_HIP_OLD_SCAN"))
	{
		uint32_t last = 0, last_scanned = 0;
		if (h_total) HIP_TRY(ctx, hipMemcpyAsync(&last, d_data + n - 1, 4, hipMemcpyDeviceToHost, st));
		CL_TRY((scan_impl<uint32_t, uint32_t>(ctx, d_data, d_data, n)));
		if (h_total)
		{
			HIP_TRY(ctx, hipMemcpyAsync(&last_scanned, d_data + n - 1, 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(ctx, hipStreamSynchronize(st));
			*h_total = (uint64_t)last + last_scanned;
		}
		return CL_OK;
	}
	uint64_t* hs = nullptr; uint64_t* ds = nullptr;
	if (h_total) HIP_TRY(ctx, cl_slot(ctx, 1, &hs, &ds));                      // (the total goes straight to mapped host memory)
	CL_TRY((scan_lookback<uint32_t, uint32_t>(ctx, d_data, d_data, n, (uint32_t*)nullptr, (unsigned long long*)ds)));
	if (h_total)
	{
		HIP_TRY(ctx, hipStreamSynchronize(st));
		*h_total = *(volatile uint64_t*)hs;
		if (*h_total >= (1ull << 32)) return cl_fail(ctx, CL_E_UNSUPPORTED, "dev_exclusive_scan_u32: the sum " + std::to_string(*h_total) + " does not fit 32 bits");
	}
	return CL_OK;
}

// d_out[0..n] = exclusive scan of d_in[0..n) widened to 64 bits, d_out[n] = total.
cl_status dev_exclusive_scan_u64(cl_ctx* ctx, const uint32_t* d_in, uint64_t* d_out, uint64_t n, uint64_t* h_total)
{
	hipStream_t st = cl_launch_stream(ctx);
	uint64_t total = 0;
	if (n && getenv("COLORD_HIP_OLD_SCAN"))
	{
		CL_TRY((scan_impl<uint32_t, uint64_t>(ctx, d_in, d_out, n)));
		uint32_t last = 0; uint64_t last_scanned = 0;
		HIP_TRY(ctx, hipMemcpyAsync(&last, d_in + n - 1, 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(ctx, hipMemcpyAsync(&last_scanned, d_out + n - 1, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(ctx, hipStreamSynchronize(st));
		total = last_scanned + last;
		HIP_TRY(ctx, hipMemcpyAsync(d_out + n, &total, 8, hipMemcpyHostToDevice, st));
		HIP_TRY(ctx, hipStreamSynchronize(st));
		if (h_total) *h_total = total;
		return CL_OK;
	}
	if (n)
	{
		uint64_t* hs = nullptr; uint64_t* ds = nullptr;
		if (h_total) HIP_TRY(ctx, cl_slot(ctx, 1, &hs, &ds));
		CL_TRY((scan_lookback<uint32_t, uint64_t>(ctx, d_in, d_out, n, d_out + n, (unsigned long long*)ds)));   // (the total lands in d_out[n] — and in mapped host memory)
		if (h_total) { HIP_TRY(ctx, hipStreamSynchronize(st)); total = *(volatile uint64_t*)hs; }
	}
	else { HIP_TRY(ctx, hipMemsetAsync(d_out, 0, 8, st)); }
	if (h_total) *h_total = total;
	return CL_OK;
}
