// scan.hip — device-wide exclusive prefix sums in ONE launch (decoupled look-back): one read and one write of the array per call.
// (Rounds 1-3: tile reduce -> recursive scan of the tile sums -> tile scan: five launches and two reads.)
#include "common.hpp"

namespace {
constexpr uint32_t SCAN_THREADS = 256;

template<typename TOut> __device__ inline TOut wave_incl_scan_t(TOut v)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { TOut t = __shfl_up(v, d, 64); if ((int)(threadIdx.x & 63) >= d) v += t; }
	return v;
}

// ---- one pass: decoupled look-back --------------------------------------------------------------------------------------------
// A tile publishes its sum (AGGREGATE), looks back over the tiles before it until it meets one whose inclusive PREFIX is known,
// publishes its own PREFIX and writes its elements: one read and one write of the array, ONE launch (the three-level reduce / scan
// above: two reads, one write, five launches and more for large arrays — 11 000 of the 45 000 dispatches of a 50-Gbase pass).  Tiles are
// handed out by a ticket, so every tile a tile waits for has been started.  The status words carry their value WITH their flag
// (8-byte agent-scope relaxed atomics on both sides: MI355X_MICROARCH.md, "valid forms", data-is-the-flag granules).
constexpr uint32_t LB_ITEMS = 16, LB_TILE = SCAN_THREADS * LB_ITEMS;
constexpr unsigned long long LB_AGG = 1ull << 62, LB_PREFIX = 2ull << 62, LB_VAL = (1ull << 52) - 1;
// A status word = flag (2 bits) | generation of the scan that wrote it (10 bits) | value (52 bits).  The buffer of a stream is reused from
// scan to scan WITHOUT being zeroed in between (round 5: the memset before every scan was a quarter of a pass's dispatches — 77 per chunk on
// an encode lane's main queue): a word of another generation reads as "not there yet", the tile ticket counts on from scan to scan (the
// launch is told where it starts).  The buffer is zeroed when the generations wrap, every 1023 scans.  (Round 5: 18 + 44 bits, and a sum
// of 2^44 or more would have run into the generation — a look-back that never ends.  Every value is masked now, 2^52 is beyond what counts
// and byte offsets on this device reach, and the host refuses a total that large.)
constexpr uint32_t LB_GEN_BITS = 10, LB_GEN_SHIFT = 52;
// ctl[0]: ticket, ctl[1..]: status of tile 0, 1, ... (zeroed before the launch).  total_out (optional): receives the sum of all.
// What is scanned and where the prefixes go is the launch's OP: the plain scan reads an array and writes every element's prefix; the run
// scan (round 6) computes its input from the sorted keys — 1 where a new context begins — and stores, for those elements only, their
// position at the place their prefix names: the starts of the context runs in ONE pass over the keys (rounds 1-5: flags written, scanned
// in place, read again: 24 bytes per symbol of the DNA coder's preparation where 4 or 8 do).
template<typename TIn, typename TOut> struct PlainOp {
	const TIn* in; TOut* out;
	__device__ inline TOut load(uint64_t j) const { return (TOut)in[j]; }
	__device__ inline void store(uint64_t j, TOut pre, TOut) const { out[j] = pre; }
	__device__ inline void finish(TOut) const {}
};
template<typename K> struct RunsOp {
	const K* keys; uint32_t* seg; uint64_t n; uint32_t shift;
	__device__ inline uint32_t load(uint64_t j) const { return (j == 0 || (keys[j - 1] >> shift) != (keys[j] >> shift)) ? 1u : 0u; }
	__device__ inline void store(uint64_t j, uint32_t pre, uint32_t v) const { if (v) seg[pre] = (uint32_t)j; }
	__device__ inline void finish(uint32_t total) const { seg[total] = (uint32_t)n; }
};
template<typename TOut, typename Op>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_lookback(Op op, uint64_t n, unsigned long long* __restrict__ ctl, TOut* total_out, unsigned long long* total64, unsigned long long ticket_base, uint32_t gen)
{
	__shared__ TOut sh[4];
	__shared__ unsigned long long s_excl;
	__shared__ uint32_t s_tile;
	if (threadIdx.x == 0) s_tile = (uint32_t)(atomicAdd(ctl, 1ull) - ticket_base);
	const unsigned long long G = (unsigned long long)gen << LB_GEN_SHIFT;
	__syncthreads();
	const uint32_t tile = s_tile, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	unsigned long long* status = ctl + 1;
	const uint64_t base = (uint64_t)tile * LB_TILE + (uint64_t)threadIdx.x * LB_ITEMS;
	TOut v[LB_ITEMS]; TOut s = 0;
#pragma unroll
	for (uint32_t i = 0; i < LB_ITEMS; ++i) { v[i] = (base + i < n) ? op.load(base + i) : (TOut)0; s += v[i]; }
	const TOut incl = wave_incl_scan_t<TOut>(s);
	if (lane == 63) sh[w] = incl;
	__syncthreads();
	TOut pre = incl - s, total = 0;
	for (uint32_t i = 0; i < 4; ++i) { if (i < w) pre += sh[i]; total += sh[i]; }
	if (w == 0)
	{
		unsigned long long excl = 0;
		if (tile == 0) { if (lane == 0) __hip_atomic_store(status, LB_PREFIX | G | ((unsigned long long)total & LB_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
		else
		{
			if (lane == 0) __hip_atomic_store(status + tile, LB_AGG | G | ((unsigned long long)total & LB_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			// lanes look at tiles hi - 1 - lane; the nearest PREFIX ends the walk, everything nearer is an AGGREGATE (or not there yet: read again)
			for (int64_t hi = tile; hi > 0; )
			{
				const int64_t j = hi - 1 - (int64_t)lane;
				unsigned long long x = j >= 0 ? __hip_atomic_load(status + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (LB_PREFIX | G);   // (before tile 0: an empty prefix)
				if (((x >> LB_GEN_SHIFT) & ((1u << LB_GEN_BITS) - 1)) != gen) x = 0;       // written by an earlier scan: not there yet
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
			if (lane == 0) __hip_atomic_store(status + tile, LB_PREFIX | G | ((excl + (unsigned long long)total) & LB_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (lane == 0) s_excl = excl;
	}
	__syncthreads();
	pre += (TOut)s_excl;
#pragma unroll
	for (uint32_t i = 0; i < LB_ITEMS; ++i) { if (base + i < n) op.store(base + i, pre, v[i]); pre += v[i]; }
	// (the last thread of the last tile has walked to the end; the 64-bit total comes from the look-back's own 62-bit sums, so a sum that
	// does not fit TOut is seen by the host instead of wrapping silently)
	if ((uint64_t)(tile + 1) * LB_TILE >= n && threadIdx.x == SCAN_THREADS - 1) { op.finish(pre); if (total_out) *total_out = pre; if (total64) *total64 = s_excl + (unsigned long long)total; }
}

template<typename TOut, typename Op>
cl_status scan_lookback_op(cl_ctx* ctx, const char* name, const Op& op, uint64_t n, TOut* d_total, unsigned long long* d_total64)
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
		C.p = (unsigned long long*)q; C.words = want; C.gen = 0;
	}
	if (C.gen == 0 || C.gen + 1 >= (1u << LB_GEN_BITS))
	{	// a new buffer, or the generations wrap: every word (and the ticket) back to zero
		HIP_TRY(ctx, hipMemsetAsync(C.p, 0, C.words * 8, st));
		C.gen = 0; C.tickets = 0;
	}
	++C.gen;
	const unsigned long long ticket_base = C.tickets; C.tickets += tiles;
	LAUNCH_NAMED(ctx, name, (k_scan_lookback<TOut, Op>), tiles, SCAN_THREADS, op, n, C.p, d_total, d_total64, ticket_base, C.gen);
	HIP_TRY(ctx, hipGetLastError());
	return CL_OK;
}
template<typename TIn, typename TOut>
cl_status scan_lookback(cl_ctx* ctx, const TIn* d_in, TOut* d_out, uint64_t n, TOut* d_total, unsigned long long* d_total64 = nullptr)
{
	return scan_lookback_op<TOut>(ctx, "k_scan_lookback<TIn, TOut>", PlainOp<TIn, TOut>{ d_in, d_out }, n, d_total, d_total64);
}
template<typename K>
cl_status run_starts(cl_ctx* ctx, const K* d_keys, uint64_t n, uint32_t shift, uint32_t* d_seg, uint64_t seg_cap, uint64_t* h_n_runs)
{
	*h_n_runs = 0;
	if (!n) return CL_OK;
	uint64_t* hs = nullptr; uint64_t* ds = nullptr;
	HIP_TRY(ctx, cl_slot(ctx, 1, &hs, &ds));
	CL_TRY((scan_lookback_op<uint32_t>(ctx, "k_run_starts", RunsOp<K>{ d_keys, d_seg, n, shift }, n, (uint32_t*)nullptr, (unsigned long long*)ds)));
	HIP_TRY(ctx, hipStreamSynchronize(cl_launch_stream(ctx)));
	*h_n_runs = *(volatile uint64_t*)hs;
	if (*h_n_runs + 1 > seg_cap) return cl_fail(ctx, CL_E_CAPACITY, "dev_run_starts: more runs than the caller made room for");   // (the kernel has written past the buffer: the caller's bound was wrong)
	return CL_OK;
}

} // namespace

// In-place exclusive scan of n uint32 (sums must fit 32 bits); *h_total (optional) = sum of all.
cl_status dev_exclusive_scan_u32(cl_ctx* ctx, uint32_t* d_data, uint64_t n, uint64_t* h_total)
{
	if (h_total) *h_total = 0;
	if (!n) return CL_OK;
	hipStream_t st = cl_launch_stream(ctx);
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
	if (n)
	{
		uint64_t* hs = nullptr; uint64_t* ds = nullptr;
		if (h_total) HIP_TRY(ctx, cl_slot(ctx, 1, &hs, &ds));
		CL_TRY((scan_lookback<uint32_t, uint64_t>(ctx, d_in, d_out, n, d_out + n, (unsigned long long*)ds)));   // (the total lands in d_out[n] — and in mapped host memory)
		if (h_total) { HIP_TRY(ctx, hipStreamSynchronize(st)); total = *(volatile uint64_t*)hs; }
		if (total > LB_VAL) return cl_fail(ctx, CL_E_UNSUPPORTED, "dev_exclusive_scan_u64: the sum does not fit the scan's 52-bit status words");
	}
	else { HIP_TRY(ctx, hipMemsetAsync(d_out, 0, 8, st)); }
	if (h_total) *h_total = total;
	return CL_OK;
}

// d_seg[0..r] = positions where (key >> shift) changes in the sorted keys d_keys[0..n), d_seg[r] = n; *h_n_runs = r.  The caller bounds r
// (seg_cap > r): by n and by the number of distinct values of key >> shift.
cl_status dev_run_starts_u32(cl_ctx* ctx, const uint32_t* d_keys, uint64_t n, uint32_t shift, uint32_t* d_seg, uint64_t seg_cap, uint64_t* h_n_runs) { return run_starts<uint32_t>(ctx, d_keys, n, shift, d_seg, seg_cap, h_n_runs); }
cl_status dev_run_starts_u64(cl_ctx* ctx, const uint64_t* d_keys, uint64_t n, uint32_t shift, uint32_t* d_seg, uint64_t seg_cap, uint64_t* h_n_runs) { return run_starts<uint64_t>(ctx, d_keys, n, shift, d_seg, seg_cap, h_n_runs); }
