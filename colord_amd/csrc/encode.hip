// encode.hip — small tuple-stream helpers (a12 / a15).  The plain forms CEncoder::AddPlainRead / AddPlainReadWithN
// (src/colord/encoder.cpp:663-681): a read stored verbatim is `start_plain` (or `start_plain_with_Ns`) followed by one
// `plain` tuple per base (utils.h:56-273, one byte each); the edit-script forms are in encode_es.hip.  And the per-base
// classes the quality coder derives from a read's own tuple stream at levels 2 and 3 (quality_coder_impl.cpp:25-75).
#include "common.hpp"
#include "objects.hpp"

namespace {
__global__ void k_plain_sizes(const uint32_t* __restrict__ lens, uint32_t n, uint32_t* __restrict__ sizes, uint32_t* __restrict__ ntup)
{
	uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n) return;
	sizes[r] = lens[r] + 1; ntup[r] = lens[r] + 1;
}
// one wave per read, one lane per base
__global__ __launch_bounds__(256) void k_plain_es(const uint64_t* __restrict__ packed, const uint32_t* __restrict__ inv, const uint64_t* __restrict__ word_off,
                                                 const uint32_t* __restrict__ lens, const uint8_t* __restrict__ has_n, uint32_t n,
                                                 const uint64_t* __restrict__ es_off, uint8_t* __restrict__ es)
{
	const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n) return;
	const uint32_t lane = threadIdx.x & 63, len = lens[r];
	const uint64_t wb = word_off[r];
	uint8_t* o = es + es_off[r];
	if (lane == 0) o[0] = (uint8_t)((has_n[r] ? 11u : 9u) << 4);         // start_plain_with_Ns = 11, start_plain = 9
	for (uint32_t i = lane; i < len; i += 64)
	{
		const uint32_t w = i >> 5, j = i & 31;
		const uint32_t b = (uint32_t)(packed[wb + w] >> (62 - 2 * j)) & 3u;
		const bool isn = (inv[wb + w] >> (31 - j)) & 1u;
		o[1 + i] = (uint8_t)((8u << 4) | (isn ? 4u : b));                  // plain = 8, low nibble = base 0..4
	}
}
// analyze_es of the quality coder (quality_coder_impl.cpp:25-75): 'P' plain read, 'A' inside an anchor tuple, 'M' unit
// match, ' ' inserted / substituted base.  One lane per read, 16 reads per wave (divergent walk).
__global__ __launch_bounds__(64) void k_es_flags(const uint8_t* __restrict__ es, const uint64_t* __restrict__ es_off, const uint32_t* __restrict__ lens, uint32_t n,
                                                const uint64_t* __restrict__ base_off, uint8_t* __restrict__ flags)
{
	if (threadIdx.x >= 16) return;
	const uint32_t r = blockIdx.x * 16 + threadIdx.x;
	if (r >= n) return;
	const uint8_t* p = es + es_off[r]; const uint8_t* e = es + es_off[r + 1];
	uint8_t* f = flags + base_off[r]; const uint32_t len = lens[r];
	if (p >= e) return;
	const uint32_t t0 = p[0] >> 4;
	if (t0 == 9 || t0 == 11) { for (uint32_t i = 0; i < len; ++i) f[i] = 'P'; return; }
	p += t0 == 10 ? 5 : 1;                                                  // start_es carries a 32-bit id
	uint32_t o = 0;
	while (p < e && o <= len)
	{
		const uint32_t t = p[0] >> 4;
		switch (t)
		{
		case 4: { const uint32_t v = ((uint32_t)(p[0] & 0xf) << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
		          for (uint32_t i = 0; i < v && o + i < len; ++i) f[o + i] = 'A'; o += v; p += 4; break; }
		case 5: p += 4; break;
		case 2: if (o < len) f[o] = 'M'; ++o; p += 1; break;
		case 0: case 3: if (o < len) f[o] = ' '; ++o; p += 1; break;
		case 6: p += 5; break;
		default: p += 1;
		}
	}
}
} // namespace

extern "C" cl_status cl_es_flags(cl_ctx* ctx, const cl_reads* R, const uint8_t* d_es, const uint64_t* d_es_off, const uint64_t* d_base_off, uint8_t* d_flags)
{
	if (!ctx || !R || !d_es_off || !d_base_off || (R->total_bases && !d_flags)) return cl_fail(ctx, CL_E_INVALID, "cl_es_flags: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (R->n_reads) LAUNCHB(ctx, R->total_bases * 1.3, k_es_flags, grid_for(R->n_reads, 16), 64, d_es, d_es_off, (const uint32_t*)R->lens.p, R->n_reads, d_base_off, d_flags);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	cl_timing_collect(ctx);
	return CL_OK;
}

extern "C" cl_status cl_encode_plain(cl_ctx* ctx, const cl_reads* R, uint8_t* d_es, uint64_t cap, uint64_t* d_es_off, uint32_t* d_es_ntuples, uint64_t* n_out)
{
	if (!ctx || !R || !d_es_off || !d_es_ntuples || !n_out) return cl_fail(ctx, CL_E_INVALID, "cl_encode_plain: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const uint32_t n = R->n_reads;
	DevBuf<uint32_t> sizes; DEV_ALLOC(ctx, sizes, n);
	if (n) LAUNCH(ctx, k_plain_sizes, grid_for(n, 256), 256, (const uint32_t*)R->lens.p, n, sizes.p, d_es_ntuples);
	HIP_TRY(ctx, hipGetLastError());
	uint64_t total = 0;
	CL_TRY(dev_exclusive_scan_u64(ctx, sizes.p, d_es_off, n, &total));
	*n_out = total;
	if (total > cap || (total && !d_es)) return cl_fail(ctx, CL_E_CAPACITY, "cl_encode_plain: need " + std::to_string(total) + " bytes");
	if (n) LAUNCH(ctx, k_plain_es, grid_for(n, 4), 256, (const uint64_t*)R->packed.p, (const uint32_t*)R->inv.p, (const uint64_t*)R->word_off.p,
		(const uint32_t*)R->lens.p, (const uint8_t*)R->has_n.p, n, (const uint64_t*)d_es_off, d_es);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	cl_timing_collect(ctx);
	return CL_OK;
}
