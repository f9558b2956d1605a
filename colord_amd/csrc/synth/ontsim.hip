// ontsim.hip — libontsim.so: the two forms of the synthetic ONT generator of ontsim_core.h.
//   host:   os_host_reads (arrays) and os_host_fastq (FASTQ text with the headers of SURVEY.md section 8d)
//   device: os_dev_lengths (output length of every read) and os_dev_fill (base codes 0..3 and quality bytes into HBM)
// Benchmark / test input tooling (bench.py, tests); the compressor library does not link it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "ontsim_core.h"

namespace {
// one wave per read: number of output bases
__global__ __launch_bounds__(256) void k_os_lengths(uint64_t seed, uint64_t gseed, const uint64_t* __restrict__ start, const uint32_t* __restrict__ len_src,
                                                    const uint8_t* __restrict__ strand, uint64_t r0, uint32_t n, uint32_t* __restrict__ out_len)
{
	const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (r >= n) return;
	const uint64_t rs = os_read_seed(seed, r0 + r);
	const uint32_t L = len_src[r], sd = strand[r]; const uint64_t st = start[r];
	uint32_t c = 0;
	for (uint32_t i = lane; i < L; i += 64) { uint32_t b0, b1; c += os_emit(rs, i, os_src_base(gseed, st, L, sd, i), &b0, &b1); }
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
	if (lane == 0) out_len[r] = c;
}
// one wave per read: bases and qualities at off[r]..
__global__ __launch_bounds__(256) void k_os_fill(uint64_t seed, uint64_t gseed, const uint64_t* __restrict__ start, const uint32_t* __restrict__ len_src,
                                                 const uint8_t* __restrict__ strand, uint64_t r0, uint32_t n, const uint64_t* __restrict__ off,
                                                 uint8_t* __restrict__ codes, uint8_t* __restrict__ quals)
{
	const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (r >= n) return;
	const uint64_t rs = os_read_seed(seed, r0 + r);
	const uint32_t L = len_src[r], sd = strand[r]; const uint64_t st = start[r];
	uint64_t o = off[r];
	for (uint32_t i0 = 0; i0 < L; i0 += 64)
	{
		const uint32_t i = i0 + lane;
		uint32_t b0 = 0, b1 = 0, e = 0;
		if (i < L) e = os_emit(rs, i, os_src_base(gseed, st, L, sd, i), &b0, &b1);
		const uint64_t m1 = __ballot(e >= 1), m2 = __ballot(e == 2), lt = (1ULL << lane) - 1;
		const uint64_t p = o + __popcll(m1 & lt) + __popcll(m2 & lt);
		if (e >= 1) codes[p] = (uint8_t)b0;
		if (e == 2) codes[p + 1] = (uint8_t)b1;
		o += __popcll(m1) + __popcll(m2);
	}
	if (quals)
	{
		const uint64_t a = off[r], b = off[r + 1];
		for (uint64_t j = a + lane; j < b; j += 64) quals[j] = os_qual(rs, (uint32_t)(j - a));
	}
}
} // namespace

extern "C" {
// device: d_out_len[n] = output lengths of reads r0 .. r0+n of the table slices d_start / d_len_src / d_strand
int os_dev_lengths(uint64_t seed, uint64_t gseed, const uint64_t* d_start, const uint32_t* d_len_src, const uint8_t* d_strand, uint64_t r0, uint32_t n, uint32_t* d_out_len, void* stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(k_os_lengths, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, seed, gseed, d_start, d_len_src, d_strand, r0, n, d_out_len);
	return (int)hipGetLastError();
}
// device: codes / quals (may be NULL) filled at d_off[r] (n + 1 offsets, exclusive scan of the lengths)
int os_dev_fill(uint64_t seed, uint64_t gseed, const uint64_t* d_start, const uint32_t* d_len_src, const uint8_t* d_strand, uint64_t r0, uint32_t n, const uint64_t* d_off,
                uint8_t* d_codes, uint8_t* d_quals, void* stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(k_os_fill, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, seed, gseed, d_start, d_len_src, d_strand, r0, n, d_off, d_codes, d_quals);
	return (int)hipGetLastError();
}
// host: one read into caller buffers (capacity 2 * len_src); returns its length
static uint32_t host_read(uint64_t seed, uint64_t gseed, uint64_t r, uint64_t start, uint32_t L, uint32_t sd, uint8_t* codes, uint8_t* quals)
{
	const uint64_t rs = os_read_seed(seed, r);
	uint32_t o = 0;
	for (uint32_t i = 0; i < L; ++i)
	{
		uint32_t b0 = 0, b1 = 0;
		const uint32_t e = os_emit(rs, i, os_src_base(gseed, start, L, sd, i), &b0, &b1);
		if (e >= 1) codes[o++] = (uint8_t)b0;
		if (e == 2) codes[o++] = (uint8_t)b1;
	}
	if (quals) for (uint32_t j = 0; j < o; ++j) quals[j] = os_qual(rs, j);
	return o;
}
// host: reads r0 .. r0+n back to back into h_codes / h_quals (capacity cap bytes each), h_off[n+1]; returns 0, or -1 if cap is too small
int os_host_reads(uint64_t seed, uint64_t gseed, const uint64_t* start, const uint32_t* len_src, const uint8_t* strand, uint64_t r0, uint32_t n,
                  uint8_t* h_codes, uint8_t* h_quals, uint64_t cap, uint64_t* h_off)
{
	uint64_t o = 0; h_off[0] = 0;
	for (uint32_t r = 0; r < n; ++r)
	{
		if (o + 2ull * len_src[r] > cap) return -1;
		o += host_read(seed, gseed, r0 + r, start[r], len_src[r], strand[r], h_codes + o, h_quals ? h_quals + o : nullptr);
		h_off[r + 1] = o;
	}
	return 0;
}
// host: FASTQ text of reads r0 .. r0+n appended to `path` ("@read_<n> ch=<n%512> start_time=2020-01-01T00:00:<n%60>Z", empty '+' line);
// *n_bases (optional) receives the number of bases written
int os_host_fastq(const char* path, int append, uint64_t seed, uint64_t gseed, const uint64_t* start, const uint32_t* len_src, const uint8_t* strand, uint64_t r0, uint32_t n, uint64_t* n_bases)
{
	FILE* f = fopen(path, append ? "ab" : "wb");
	if (!f) return -1;
	std::vector<char> obuf(1 << 22); setvbuf(f, obuf.data(), _IOFBF, obuf.size());
	std::vector<uint8_t> codes, quals; uint64_t tot = 0;
	for (uint32_t r = 0; r < n; ++r)
	{
		codes.resize(2ull * len_src[r] + 1); quals.resize(2ull * len_src[r] + 1);
		const uint32_t L = host_read(seed, gseed, r0 + r, start[r], len_src[r], strand[r], codes.data(), quals.data());
		for (uint32_t j = 0; j < L; ++j) codes[j] = (uint8_t)"ACGT"[codes[j]];
		const unsigned long long id = r0 + r;
		fprintf(f, "@read_%llu ch=%llu start_time=2020-01-01T00:00:%02lluZ\n", id, id % 512, id % 60);
		codes[L] = '\n'; quals[L] = '\n';
		if (fwrite(codes.data(), 1, L + 1, f) != L + 1 || fwrite("+\n", 1, 2, f) != 2 || fwrite(quals.data(), 1, L + 1, f) != L + 1) { fclose(f); return -2; }
		tot += L;
	}
	if (n_bases) *n_bases = tot;
	return fclose(f) == 0 ? 0 : -2;
}
}
