// capi.hip — context management, host-side acceptor (a6) and thin exported wrappers.
#include "common.hpp"
#include "objects.hpp"
#include <cmath>
#include <random>

extern "C" cl_status cl_ctx_create(int device, cl_ctx** out)
{
	if (!out) return CL_E_INVALID;
	*out = nullptr;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) { fprintf(stderr, "colord_hip: no HIP device available (%s)\n", hipGetErrorString(e)); return CL_E_HIP; }
	if (device < 0 || device >= n) return CL_E_INVALID;
	if (hipSetDevice(device) != hipSuccess) return CL_E_HIP;
	cl_ctx* c = new cl_ctx(device);
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { cl_ctx_destroy(c); return CL_E_HIP; }   // (never reassigned while the context works)
	hipDeviceProp_t p;
	if (hipGetDeviceProperties(&p, device) == hipSuccess) c->n_cu = p.multiProcessorCount;
	*out = c;
	return CL_OK;
}
// Internal (stream.hip): the context's streams at a priority of the device's range: +1 = highest, -1 = lowest, 0 = default.  Queues of
// higher priority are served first when waves compete for the machine: the compressor raises its encode lanes (the chain that bounds a
// pass) above the coders and lowers the preparation threads (which have slack).  Call before the context has done any work.
void cl_ctx_set_priority(cl_ctx* c, int level, int role)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	int least = 0, greatest = 0;
	if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
	level = cl_role_level(role >= 0 ? role : c->role, level);
	const int prio = level > 0 ? greatest : level < 0 ? least : 0;
	const bool role_changes = role >= 0 && role < CL_N_ROLES && cl_cu_mask_cfg().on[role] && (role != c->role || !c->masked);
	if (role >= 0) c->role = role;
	if (prio == c->prio && !role_changes) return;
	c->prio = prio; c->masked = c->role < CL_N_ROLES && cl_cu_mask_cfg().on[c->role];
	// the scans' status words are kept per stream (scan.hip): those of a stream that goes away go back to the pool with it — a handle the
	// runtime hands out again must not inherit another stream's generation and ticket
	auto drop_scan_ctl = [&](hipStream_t old) {
		auto it = c->scan_ctl.find((void*)old);
		if (it == c->scan_ctl.end()) return;
		if (it->second.p) c->pool.put(it->second.p, it->second.got, c->pool_id);
		c->scan_ctl.erase(it);
	};
	if (c->stream)
	{
		(void)hipStreamSynchronize(c->stream); drop_scan_ctl(c->stream); (void)hipStreamDestroy(c->stream); c->stream = nullptr;
		if (cl_stream_create(c, &c->stream) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking); }
	}
	for (std::atomic<hipStream_t>* s : { &c->side, &c->side2, &c->side3 })
		if (hipStream_t old = s->load())
		{
			(void)hipStreamSynchronize(old); drop_scan_ctl(old); (void)hipStreamDestroy(old);
			hipStream_t ns = nullptr;
			if (cl_stream_create(c, &ns) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamCreateWithFlags(&ns, hipStreamNonBlocking); }
			s->store(ns);
		}
}
// COLORD_HIP_POOL_POISON (common.hpp)
namespace { __global__ void k_poison_check(const uint32_t* __restrict__ p, uint64_t n, unsigned long long* __restrict__ bad)
{
	uint64_t mine = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) mine += p[i] != 0xCDCDCDCDu;
	if (mine) atomicAdd(bad, (unsigned long long)mine);
} }
void cl_ctx_poison(cl_ctx* c, void* p, uint64_t bytes)
{
	if (!c || !c->stream) return;
	(void)hipSetDevice(c->device);
	if (hipMemsetAsync(p, 0xCD, bytes, c->stream) != hipSuccess) { (void)hipGetLastError(); return; }
	// (waited for: the pool hands the owner its own extent back at once, and the next use may be on ANOTHER of its streams — a pattern still
	// queued here would land on top of the new writes: false reports, or corrupted data, in the very mode that looks for races)
	(void)hipStreamSynchronize(c->stream);
}
uint64_t cl_pool_poison_check(void* p, uint64_t bytes)
{
	static unsigned long long* d_bad = nullptr;
	if (!d_bad && hipMalloc((void**)&d_bad, 8) != hipSuccess) { (void)hipGetLastError(); return 0; }
	(void)hipDeviceSynchronize();                                               // (the pattern's memset may still be queued on its owner's stream)
	(void)hipMemset(d_bad, 0, 8);
	hipLaunchKernelGGL(k_poison_check, dim3(1024), dim3(256), 0, nullptr, (const uint32_t*)p, bytes / 4, d_bad);
	unsigned long long h = 0;
	if (hipGetLastError() != hipSuccess || hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "colord_hip: pool poison: the check itself failed\n"); return 0; }
	return h;
}
// every stream of the context (the shared pool calls this before it hands memory the context released to another one)
void cl_ctx_drain(cl_ctx* c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	for (hipStream_t s : { c->stream, c->side.load(std::memory_order_acquire), c->side2.load(std::memory_order_acquire), c->side3.load(std::memory_order_acquire) }) if (s) (void)hipStreamSynchronize(s);
}
int cl_ctx_fence(cl_ctx* c, hipEvent_t* ev)
{
	if (!c) return 0;
	int n = 0;
	for (hipStream_t s : { c->stream, c->side.load(std::memory_order_acquire), c->side2.load(std::memory_order_acquire), c->side3.load(std::memory_order_acquire) }) if (s && n < 4) { if (hipEventRecord(ev[n], s) == hipSuccess) ++n; else (void)hipGetLastError(); }
	return n;
}
extern "C" void cl_ctx_destroy(cl_ctx* c)
{
	if (!c) return;
	for (cl_ctx* l : c->lanes) cl_ctx_destroy(l);
	c->lanes.clear();
	if (c->prep) { cl_ctx_destroy(c->prep); c->prep = nullptr; }
	if (c->qprep) { cl_ctx_destroy(c->qprep); c->qprep = nullptr; }
	(void)hipSetDevice(c->device);
	// out of the pool's owner table first (under its mutex; waits for a drain of this context another thread may be in): after
	// this no fence or drain can reach the streams destroyed below
	cl_ctx_drain(c);
	for (auto& kv : c->scan_ctl) if (kv.second.p) c->pool.put(kv.second.p, kv.second.got, c->pool_id);
	c->scan_ctl.clear();
	c->pool.drop_owner(c->pool_id);
	for (auto& p : c->pending) { (void)hipEventDestroy(p.second.first); (void)hipEventDestroy(p.second.second); }
	for (auto e : c->ev_pool) (void)hipEventDestroy(e);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	for (std::atomic<hipStream_t>* s : { &c->side, &c->side2, &c->side3 }) if (hipStream_t x = s->load()) (void)hipStreamDestroy(x);
	if (c->inv_tab) (void)hipFree(c->inv_tab);
	if (c->slots_h) (void)hipHostFree(c->slots_h);
	const int dev = c->device;
	delete c;
	cl_device_pool_release(dev);
}
extern "C" const char* cl_last_error(const cl_ctx* c) { return c ? c->err.c_str() : "null context"; }
extern "C" void* cl_ctx_stream(cl_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" void cl_ctx_set_timing(cl_ctx* c, int on) { if (c) c->timing = on != 0; }
extern "C" cl_status cl_ctx_last_kernel_ms(const cl_ctx* c, const char* kernel, double* ms, uint32_t* launches)
{
	if (!c || !kernel) return CL_E_INVALID;
	auto it = c->times.find(kernel);
	if (it == c->times.end()) { if (ms) *ms = 0; if (launches) *launches = 0; return CL_E_INVALID; }
	if (ms) *ms = it->second.ms;
	if (launches) *launches = it->second.launches;
	return CL_OK;
}

extern "C" cl_status cl_ctx_kernel_times(cl_ctx* c, char* buf, uint64_t cap, uint64_t* needed)
{
	if (!c) return CL_E_INVALID;
	cl_timing_collect(c);                                                      // (what has completed: never a wait — callers ask after every stage, and a coder of the next chunk may be running)
	std::string out;
	for (auto& kv : c->times) out += kv.first + "\t" + std::to_string(kv.second.ms) + "\t" + std::to_string(kv.second.launches) + "\t" + std::to_string(kv.second.bytes) + "\t" + std::to_string(kv.second.cells) + "\n";
	if (needed) *needed = out.size() + 1;
	if (!buf || cap < out.size() + 1) return CL_E_CAPACITY;
	memcpy(buf, out.c_str(), out.size() + 1);
	c->times.clear();
	return CL_OK;
}

// a6 — CRefReadsAccepter (ref_reads_accepter.h:23-58).  The decision stream is one sequential
// std::mt19937 (default seed) feeding std::uniform_real_distribution<double>(0,1): the archive format
// depends on libstdc++'s generate_canonical (two 32-bit draws per double), so the same library types are
// used here on the host; cost is negligible (one draw per read).
extern "C" cl_status cl_ref_accept(uint32_t n_reads, uint32_t n_pseudo, uint32_t range, double exponent, uint8_t* h_out)
{
	if (!h_out || range == 0) return CL_E_INVALID;
	std::mt19937 mt;
	std::uniform_real_distribution<double> dist(0.0, 1.0);
	for (uint64_t i = 0; i < (uint64_t)n_reads + n_pseudo; ++i)
	{
		if (i < n_pseudo) { h_out[i] = 1; continue; }
		uint32_t range_no = (uint32_t)((i - n_pseudo) / range);
		double p = std::pow(1.0 / (range_no + 1ul), exponent);
		h_out[i] = dist(mt) <= p;
	}
	return CL_OK;
}

extern "C" cl_status cl_sort_u64(cl_ctx* ctx, uint64_t* d_keys, uint64_t n, uint32_t begin_bit, uint32_t end_bit)
{
	if (!ctx || (!d_keys && n)) return cl_fail(ctx, CL_E_INVALID, "cl_sort_u64: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	cl_timing_begin(ctx);
	cl_status s = dev_sort_pairs(ctx, d_keys, nullptr, n, begin_bit, end_bit);
	cl_timing_collect(ctx);
	return s;
}
extern "C" cl_status cl_sort_u64_u32(cl_ctx* ctx, uint64_t* d_keys, uint32_t* d_vals, uint64_t n, uint32_t begin_bit, uint32_t end_bit)
{
	if (!ctx || ((!d_keys || !d_vals) && n)) return cl_fail(ctx, CL_E_INVALID, "cl_sort_u64_u32: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	cl_timing_begin(ctx);
	cl_status s = dev_sort_pairs(ctx, d_keys, d_vals, n, begin_bit, end_bit);
	cl_timing_collect(ctx);
	return s;
}
