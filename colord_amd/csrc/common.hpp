// common.hpp — context, device memory helpers and wave64 device primitives shared by the kernels.
// gfx950 only: wavefront = 64 lanes, no dual paths.
#pragma once
// device functions that a debugging build (tests/tools, -DCL_HOST_DEBUG) also compiles for the host; the library never does
#ifdef CL_HOST_DEBUG
#define CL_DEV __host__ __device__
#else
#define CL_DEV __device__
#endif
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <mutex>
#include "../../include/colord_hip.h"

#define CL_WAVE 64

struct KernelTime { double ms = 0; uint32_t launches = 0; double bytes = 0; };   // bytes = algorithmic HBM bytes of the timed launches

// Grow-only caching device allocator: hipMalloc/hipFree cost ~0.1-1 ms each and synchronise the device,
// which dominated short calls.  Blocks are binned by rounded size and reused across calls.
struct DevPool {
	std::mutex mu;                                // a buffer made on one thread (an encode lane of cl_compressor) may be released on another
	std::multimap<uint64_t, void*> free_blocks;
	uint64_t cached_bytes = 0, live_bytes = 0, peak_live = 0, peak_total = 0;   // (statistics for COLORD_HIP_POOL_DEBUG)
	void account(uint64_t got) { live_bytes += got; if (live_bytes > peak_live) peak_live = live_bytes; if (live_bytes + cached_bytes > peak_total) peak_total = live_bytes + cached_bytes; }
	static uint64_t round_size(uint64_t bytes)
	{
		if (bytes < 256) return 256;
		if (bytes < (1ull << 21)) { uint64_t p = 256; while (p < bytes) p <<= 1; return p; }
		const uint64_t g = 1ull << 21; return (bytes + g - 1) / g * g;
	}
	hipError_t get(uint64_t bytes, void** out, uint64_t* got)
	{
		std::lock_guard<std::mutex> lock(mu);
		uint64_t r = round_size(bytes);
		// best fit: the smallest cached block that holds r without wasting more than half of it (hipMalloc of tens of GB
		// costs around a second, and a cache of exact sizes only would outgrow HBM over one pass of the pipeline)
		auto it = free_blocks.lower_bound(r);
		if (it != free_blocks.end() && it->first <= r + r / 2 + (64ull << 20))
		{ *out = it->second; *got = it->first; cached_bytes -= it->first; free_blocks.erase(it); account(*got); return hipSuccess; }
		// a large request rather borrows a larger cached block than grows the footprint (the stages of a pass run one after
		// the other; their big buffers are not needed at the same time)
		if (it != free_blocks.end() && r >= (1ull << 30))
		{ *out = it->second; *got = it->first; cached_bytes -= it->first; free_blocks.erase(it); account(*got); return hipSuccess; }
		hipError_t e = hipMalloc(out, r);
		static const bool dbg = getenv("COLORD_HIP_POOL_DEBUG") != nullptr;
		if (dbg) fprintf(stderr, "[pool] hipMalloc %.3f GB (%s), cached %.3f GB in %zu blocks\n", r / 1e9, e == hipSuccess ? "ok" : "failed", cached_bytes / 1e9, free_blocks.size());
		while (e != hipSuccess && !free_blocks.empty())
		{	// out of memory: give back cached blocks, largest first, until the request fits
			(void)hipGetLastError();
			auto last = std::prev(free_blocks.end());
			(void)hipFree(last->second); cached_bytes -= last->first; free_blocks.erase(last);
			e = hipMalloc(out, r);
		}
		*got = r;
		if (e == hipSuccess) account(r);
		return e;
	}
	void put(void* p, uint64_t r) { std::lock_guard<std::mutex> lock(mu); free_blocks.emplace(r, p); cached_bytes += r; live_bytes -= r; }
	void trim() { std::lock_guard<std::mutex> lock(mu); if (getenv("COLORD_HIP_POOL_DEBUG")) fprintf(stderr, "[pool] at trim: peak live %.1f GB, peak live + cached %.1f GB, cached %.1f GB\n", peak_live / 1e9, peak_total / 1e9, cached_bytes / 1e9);
		for (auto& b : free_blocks) (void)hipFree(b.second); free_blocks.clear(); cached_bytes = 0; }
};

static inline bool cl_pool_debug() { static const bool dbg = getenv("COLORD_HIP_POOL_DEBUG") != nullptr; return dbg; }

struct cl_ctx {
	DevPool pool;
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t side = nullptr;                  // second stream for chains that would leave the machine idle (created on first use)
	std::string err;
	bool timing = false;
	std::map<std::string, KernelTime> times;     // per-kernel accumulated HIP-event time of the last API call
	std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
	std::vector<double> pending_bytes;
	double next_bytes = 0;                       // algorithmic bytes of the next LAUNCH (set by LAUNCHB)
	std::vector<hipEvent_t> ev_pool;
	int n_cu = 256;
};

static inline cl_status cl_fail(cl_ctx* c, cl_status s, const std::string& msg)
{
	if (c) c->err = msg;
	return s;
}

#define HIP_TRY(ctx, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
	return cl_fail((ctx), CL_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + ":" + std::to_string(__LINE__)); } } while (0)
#define CL_TRY(expr) do { cl_status _s = (expr); if (_s != CL_OK) return _s; } while (0)

// ---- device buffers (RAII, freed with the owning object) -----------------------------------------
template<typename T> struct DevBuf {
	T* p = nullptr; uint64_t n = 0; uint64_t bytes = 0; DevPool* pool = nullptr;
	DevBuf() = default;
	DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
	DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n), bytes(o.bytes), pool(o.pool) { o.p = nullptr; o.n = 0; }
	DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; bytes = o.bytes; pool = o.pool; o.p = nullptr; o.n = 0; } return *this; }
	~DevBuf() { release(); }
	void release() { if (p) { if (pool) pool->put(p, bytes); else (void)hipFree(p); } p = nullptr; n = 0; }
	hipError_t alloc(cl_ctx* c, uint64_t count);
};
template<typename T> hipError_t DevBuf<T>::alloc(cl_ctx* c, uint64_t count)
{
	release(); n = count; if (!count) count = 1;
	pool = &c->pool;
	void* q = nullptr;
	hipError_t e = pool->get(count * sizeof(T), &q, &bytes);
	p = (T*)q;
	if (e != hipSuccess) { p = nullptr; n = 0; }
	return e;
}

#define DEV_ALLOC(ctx, buf, count) do { hipError_t _e = (buf).alloc((ctx), count); \
	if ((buf).bytes >= (1ull << 30) && cl_pool_debug()) fprintf(stderr, "[pool] %s:%d %s %.2f GB (live %.1f GB)\n", &__FILE__[sizeof(__FILE__) > 24 ? sizeof(__FILE__) - 24 : 0], __LINE__, #buf, (buf).bytes / 1e9, (ctx)->pool.live_bytes / 1e9); \
	if (_e != hipSuccess) \
	return cl_fail((ctx), CL_E_NOMEM, std::string("hipMalloc(" #buf ") of ") + std::to_string((uint64_t)(count)) + " elems: " + hipGetErrorString(_e)); } while (0)

// ---- per-kernel timing with HIP events on the context stream -------------------------------------
struct KernelTimer {
	cl_ctx* c; const char* name; hipEvent_t a = nullptr, b = nullptr;
	KernelTimer(cl_ctx* c_, const char* n) : c(c_), name(n)
	{
		if (!c->timing) return;
		auto get = [&]() { hipEvent_t e; if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
		a = get(); b = get();
		(void)hipEventRecord(a, c->stream);
	}
	~KernelTimer()
	{
		if (!c->timing) return;
		(void)hipEventRecord(b, c->stream);
		c->pending.push_back({ name, { a, b } });
		c->pending_bytes.push_back(c->next_bytes); c->next_bytes = 0;
	}
};
// every kernel launch goes through LAUNCH so that per-kernel HIP-event times are complete
#define LAUNCH(ctx, kernel, grid, block, ...) do { KernelTimer _kt((ctx), #kernel); \
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (ctx)->stream, __VA_ARGS__); } while (0)
// LAUNCH with the algorithmic HBM byte count of this launch (for achieved-GB/s reporting)
#define LAUNCHB(ctx, bytes, kernel, grid, block, ...) do { (ctx)->next_bytes = (double)(bytes); LAUNCH(ctx, kernel, grid, block, __VA_ARGS__); } while (0)
// the same with dynamic LDS
#define LAUNCHB_SHM(ctx, bytes, kernel, grid, block, shm, ...) do { (ctx)->next_bytes = (double)(bytes); KernelTimer _kt((ctx), #kernel); \
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (shm), (ctx)->stream, __VA_ARGS__); } while (0)
static inline void cl_timing_begin(cl_ctx*) {}   // times accumulate until cl_ctx_kernel_times reports them
static inline void cl_timing_collect(cl_ctx* c)
{
	if (!c->timing) return;
	size_t pi = 0;
	for (auto& p : c->pending)
	{
		(void)hipEventSynchronize(p.second.second);
		float ms = 0; (void)hipEventElapsedTime(&ms, p.second.first, p.second.second);
		std::string nm = p.first;
		if (!nm.empty() && nm.front() == '(' && nm.back() == ')') nm = nm.substr(1, nm.size() - 2);
		auto& t = c->times[nm]; t.ms += ms; t.launches += 1; t.bytes += c->pending_bytes[pi++];
		c->ev_pool.push_back(p.second.first); c->ev_pool.push_back(p.second.second);
	}
	c->pending.clear(); c->pending_bytes.clear();
}

// ---- device primitives ---------------------------------------------------------------------------
// MurmurHash3 fmix64 — the reference's filter hash (filtering-KMC/hash_filter.h:8-16)
__host__ __device__ static inline uint64_t hash_mm(uint64_t x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
	x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
	x ^= x >> 33;
	return x;
}

// Exact "h % f == 0" for a launch-constant f without a divide: f = 2^s * d (d odd);
// h divisible by f  <=>  low s bits of h are zero  and  (h >> s) * inv(d) mod 2^64 <= (2^64-1)/d.
struct ModTest { uint64_t low_mask, dinv, lim; uint32_t s; };
static inline ModTest make_modtest(uint32_t f)
{
	ModTest m; uint32_t s = 0; uint64_t d = f;
	while ((d & 1) == 0) { d >>= 1; ++s; }
	uint64_t inv = d;                       // Newton iteration for the inverse of odd d modulo 2^64
	for (int i = 0; i < 6; ++i) inv *= 2 - d * inv;
	m.s = s; m.low_mask = (s ? ((1ULL << s) - 1) : 0); m.dinv = inv; m.lim = ~0ULL / d;
	return m;
}
__device__ static inline bool mod_is_zero(uint64_t h, const ModTest& m)
{
	return ((h & m.low_mask) == 0) && (((h >> m.s) * m.dinv) <= m.lim);
}

__device__ static inline uint32_t lane_id() { return threadIdx.x & 63; }
__device__ static inline uint64_t lanemask_lt() { return (1ULL << lane_id()) - 1; }

// inclusive wave scan (sum) over 64 lanes
__device__ static inline uint32_t wave_incl_scan(uint32_t v)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(v, d, 64); if ((int)lane_id() >= d) v += t; }
	return v;
}
__device__ static inline uint32_t wave_sum(uint32_t v)
{
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
	return v;
}

// Block-wide exclusive scan for blockDim.x == 256 (4 waves).  sh must hold 4 uint32.  Returns the
// exclusive prefix of v; *total receives the block sum.
__device__ static inline uint32_t block_excl_scan_256(uint32_t v, uint32_t* sh, uint32_t* total)
{
	uint32_t incl = wave_incl_scan(v);
	uint32_t w = threadIdx.x >> 6;
	__syncthreads();                       // protect sh from a previous use
	if (lane_id() == 63) sh[w] = incl;
	__syncthreads();
	uint32_t base = 0, tot = 0;
#pragma unroll
	for (uint32_t i = 0; i < 4; ++i) { uint32_t s = sh[i]; if (i < w) base += s; tot += s; }
	*total = tot;
	return base + incl - v;
}

// device-wide primitives implemented in scan.hip / sort.hip
cl_status dev_exclusive_scan_u32(cl_ctx* ctx, uint32_t* d_data, uint64_t n, uint64_t* h_total);   // in place
cl_status dev_exclusive_scan_u64(cl_ctx* ctx, const uint32_t* d_in, uint64_t* d_out, uint64_t n, uint64_t* h_total); // d_out has n+1
cl_status dev_sort_pairs(cl_ctx* ctx, uint64_t* d_keys, uint32_t* d_vals, uint64_t n, uint32_t begin_bit, uint32_t end_bit);
cl_status dev_sort_keys32_pairs(cl_ctx* ctx, uint32_t* d_keys, uint32_t* d_vals, uint64_t n, uint32_t begin_bit, uint32_t end_bit);

static inline uint32_t grid_for(uint64_t n, uint32_t per_block) { return (uint32_t)((n + per_block - 1) / per_block); }
