// scatter_probe — how fast does MI355X take scattered 8-byte stores, as a function of the span they fall into?
// (measurement aid for DESIGN.md: the coders' model kernels write one 8-byte triple per symbol at a data-dependent place)
//   ./scatter_probe            prints one line per (pattern, span)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// every thread: `per` stores; store t of thread i goes to window w = (i*per+t) / per_window (windows taken one after the other), at a random 8-byte slot inside it
__global__ void k_scatter(uint64_t* __restrict__ dst, uint64_t n, uint64_t win_words, uint64_t per_window, uint64_t salt)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t w = i / per_window;
	dst[w * win_words + mix(i ^ salt) % win_words] = i;
}
__global__ void k_gather(const uint64_t* __restrict__ src, uint64_t n, uint64_t win_words, uint64_t per_window, uint64_t salt, uint64_t* __restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t w = i / per_window;
	const uint64_t v = src[w * win_words + mix(i ^ salt) % win_words];
	if (v == 0x1234567) out[0] = v;
}
// coalesced reference
__global__ void k_stream(uint64_t* __restrict__ dst, uint64_t n) { const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = i; }
int main()
{
	const uint64_t total_words = 1ull << 30;        // 8 GB target
	const uint64_t n = 1ull << 29;                  // 512 M stores
	uint64_t *d = nullptr, *d2 = nullptr, *o = nullptr;
	CK(hipMalloc(&d, total_words * 8)); CK(hipMalloc(&d2, total_words * 8)); CK(hipMalloc(&o, 64));
	CK(hipMemset(d, 0, total_words * 8)); CK(hipMemset(d2, 0, total_words * 8));
	hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	auto run = [&](const char* what, uint64_t win_bytes, int mode) -> int {
		const uint64_t win_words = win_bytes / 8, n_win = total_words / win_words, per_window = (n + n_win - 1) / n_win;
		float best = 1e30f;
		for (int rep = 0; rep < 3; ++rep)
		{
			CK(hipDeviceSynchronize());
			CK(hipEventRecord(a, s1));
			if (mode == 0) k_scatter<<<(uint32_t)(n / 256), 256, 0, s1>>>(d, n, win_words, per_window, rep);
			else if (mode == 1) k_gather<<<(uint32_t)(n / 256), 256, 0, s1>>>(d, n, win_words, per_window, rep, o);
			else if (mode == 2) { k_scatter<<<(uint32_t)(n / 256), 256, 0, s1>>>(d, n, win_words, per_window, rep); k_scatter<<<(uint32_t)(n / 256), 256, 0, s2>>>(d2, n, win_words, per_window, rep + 7); }
			else k_stream<<<(uint32_t)(n / 256), 256, 0, s1>>>(d, n);
			CK(hipGetLastError());
			CK(hipStreamSynchronize(s2));
			CK(hipEventRecord(b, s1)); CK(hipEventSynchronize(b));
			float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
		}
		const double stores = (mode == 2 ? 2.0 : 1.0) * n;
		printf("%-28s span %8.0f MB : %8.2f ms  %7.2f G accesses/s  %7.1f GB/s useful\n", what, win_bytes / 1048576.0, best, stores / best / 1e6, stores * 8 / best / 1e6);
		fflush(stdout);
		return 0;
	};
	run("coalesced 8-B stores", total_words * 8, 3);
	for (uint64_t wb : { 8ull << 30, 2ull << 30, 512ull << 20, 128ull << 20, 32ull << 20, 8ull << 20, 2ull << 20 }) run("scattered 8-B stores", wb, 0);
	for (uint64_t wb : { 8ull << 30, 512ull << 20, 128ull << 20, 32ull << 20 }) run("two kernels, two targets", wb, 2);
	for (uint64_t wb : { 8ull << 30, 512ull << 20, 128ull << 20, 32ull << 20, 2ull << 20 }) run("scattered 8-B loads", wb, 1);
	return 0;
}
