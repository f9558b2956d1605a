// pmc_calib.hip — known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md §HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ...
// other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel moves exactly BYTES bytes (printed) in one of the access patterns the library's kernels use:
//   calib_read_b128 / b64 / b32     coalesced streaming reads, 16 / 8 / 4 bytes per lane   (k_range_code triples; sort keys; ids)
//   calib_write_b128 / b64 / b32    coalesced streaming writes
//   calib_scatter_b128              16-byte records to pseudo-random slots (k_evolve_*: triples scattered to stream order)
//   calib_gather_b64x8              one random 64-byte bucket per lane (table probes)
// Run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- tools/pmc_calib/pmc_calib   (and again with WRITE_SIZE)
// tools/pmc_traffic.py turns the two passes into correction factors (bytes moved / counter bytes).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template<class T> __global__ __launch_bounds__(256) void calib_read(const T* __restrict__ in, uint64_t n, uint64_t* __restrict__ sink)
{
	uint64_t acc = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
	{
		const T v = in[i];
		const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
		for (unsigned j = 0; j < sizeof(T) / 4; ++j) acc += w[j];
	}
	if (acc == 0x123456789abcdefull) sink[0] = acc;          // never true for the zero-filled input; keeps the loads
}
template<class T> __global__ __launch_bounds__(256) void calib_write(T* __restrict__ out, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
	{
		T v; uint32_t* w = reinterpret_cast<uint32_t*>(&v);
		for (unsigned j = 0; j < sizeof(T) / 4; ++j) w[j] = (uint32_t)i + j;
		out[i] = v;
	}
}
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; return x ^ (x >> 33); }
// every slot written exactly once (i -> i * odd mod 2^k is a permutation), 16 bytes each, no locality
__global__ __launch_bounds__(256) void calib_scatter_b128(uint4* __restrict__ out, uint64_t n_pow2)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_pow2; i += (uint64_t)gridDim.x * 256)
	{
		const uint64_t s = (i * 0x9E3779B97F4A7C15ull) & (n_pow2 - 1);
		out[s] = make_uint4((uint32_t)i, 1, 2, 3);
	}
}
// one 64-byte bucket (8 x u64) per lane at a hashed position
__global__ __launch_bounds__(256) void calib_gather_b64x8(const uint64_t* __restrict__ in, uint64_t n_buckets_pow2, uint64_t n_probes, uint64_t* __restrict__ sink)
{
	uint64_t acc = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_probes; i += (uint64_t)gridDim.x * 256)
	{
		const uint64_t b = mix(i) & (n_buckets_pow2 - 1);
		const ulonglong2* p = reinterpret_cast<const ulonglong2*>(in + b * 8);
		const ulonglong2 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
		acc += a0.x + a0.y + a1.x + a1.y + a2.x + a2.y + a3.x + a3.y;
	}
	if (acc == 0x123456789abcdefull) sink[0] = acc;
}

int main(int argc, char** argv)
{
	const uint64_t BYTES = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 8ull) << 30;        // default 8 GiB: 32 x the 256-MiB Infinity Cache
	void* buf; uint64_t* sink;
	CK(hipMalloc(&buf, BYTES)); CK(hipMalloc(&sink, 64));
	CK(hipMemset(buf, 0, BYTES)); CK(hipMemset(sink, 0, 64));
	CK(hipDeviceSynchronize());
	const int grid = 256 * 16;
	printf("bytes_per_kernel %llu\n", (unsigned long long)BYTES);
	hipLaunchKernelGGL(calib_read<uint4>, grid, 256, 0, 0, (const uint4*)buf, BYTES / 16, sink);
	hipLaunchKernelGGL(calib_read<uint2>, grid, 256, 0, 0, (const uint2*)buf, BYTES / 8, sink);
	hipLaunchKernelGGL(calib_read<uint32_t>, grid, 256, 0, 0, (const uint32_t*)buf, BYTES / 4, sink);
	hipLaunchKernelGGL(calib_gather_b64x8, grid, 256, 0, 0, (const uint64_t*)buf, BYTES / 64, BYTES / 64, sink);
	hipLaunchKernelGGL(calib_write<uint4>, grid, 256, 0, 0, (uint4*)buf, BYTES / 16);
	hipLaunchKernelGGL(calib_write<uint2>, grid, 256, 0, 0, (uint2*)buf, BYTES / 8);
	hipLaunchKernelGGL(calib_write<uint32_t>, grid, 256, 0, 0, (uint32_t*)buf, BYTES / 4);
	hipLaunchKernelGGL(calib_scatter_b128, grid, 256, 0, 0, (uint4*)buf, BYTES / 16);
	CK(hipGetLastError());
	CK(hipDeviceSynchronize());
	printf("done\n");
	return 0;
}
