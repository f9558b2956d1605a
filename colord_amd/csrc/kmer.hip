// kmer.hip — read arena, canonical k-mer scan (a1), exact count + filter (a2), membership table (a3),
// accepted k-mers per read (a4).  Integer/bit work on 2-bit packed bases: coalesced 8-byte loads of the
// arena, rolling forward/reverse-complement k-mers in registers, murmur fmix64 + divide-free modulo
// test, ballot/prefix compaction.  No MFMA anywhere on this path.
#include "common.hpp"
#include "objects.hpp"
static inline double cap_hint(uint64_t bases, uint32_t f) { return (double)bases / (f ? f : 1); }   // expected survivors, N/f
#include <algorithm>

// ======================================================================================================
// arena
// ======================================================================================================
namespace {

__global__ void k_read_geometry(const uint64_t* __restrict__ off, uint32_t n_reads, uint32_t* __restrict__ lens,
                                uint32_t* __restrict__ words, uint32_t* __restrict__ err)
{
	uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	uint64_t l = off[r + 1] - off[r];
	if (off[r + 1] < off[r] || l >= 0xfffffff0ull) { atomicOr(err, 1u); l = 0; }
	lens[r] = (uint32_t)l;
	words[r] = (uint32_t)(l / 32 + 1);          // always at least one pad base after the read
}

__device__ inline uint32_t base_code(uint8_t c, int ascii, uint32_t* bad)
{
	if (!ascii) { if (c > 4) *bad = 1; return c; }
	switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'N': return 4; }   // upper case only: SymbToBinMap (utils.h:466-480) has the lower-case entries commented out
	*bad = 1; return 4;
}

// one wave per read; lane = one 32-base word per iteration
__global__ __launch_bounds__(256) void k_pack(const uint8_t* __restrict__ codes, const uint64_t* __restrict__ off,
                                              const uint64_t* __restrict__ word_off, const uint32_t* __restrict__ lens,
                                              uint32_t n_reads, int ascii, uint64_t* __restrict__ packed,
                                              uint32_t* __restrict__ inv, uint8_t* __restrict__ has_n, uint32_t* __restrict__ err)
{
	uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n_reads) return;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t len = lens[r];
	const uint64_t src = off[r], wb = word_off[r];
	const uint32_t nw = len / 32 + 1;
	uint32_t bad = 0, anyn = 0;
	for (uint32_t w = lane; w < nw; w += 64)
	{
		uint64_t pw = 0; uint32_t iv = 0;
		uint32_t p0 = w * 32;
#pragma unroll 4
		for (uint32_t j = 0; j < 32; ++j)
		{
			uint32_t p = p0 + j;
			uint32_t c = 4; bool pad = p >= len;
			if (!pad) c = base_code(codes[src + p], ascii, &bad);
			if (c > 3) { iv |= 1u << (31 - j); if (!pad) anyn = 1; c = 0; }
			pw |= (uint64_t)c << (62 - 2 * j);
		}
		packed[wb + w] = pw; inv[wb + w] = iv;
	}
	uint64_t bn = __ballot(anyn != 0), bb = __ballot(bad != 0);
	if (lane == 0) { has_n[r] = bn != 0; if (bb) atomicOr(err, 2u); }
}

__global__ void k_arena_tail(uint64_t* packed, uint32_t* inv, uint64_t total_words)
{
	packed[total_words] = 0; inv[total_words] = 0xffffffffu;
}
} // namespace

extern "C" cl_status cl_reads_pack(cl_ctx* ctx, const uint8_t* d_codes, const uint64_t* d_offsets, uint32_t n_reads,
                                   int ascii, cl_reads** out)
{
	if (!ctx || !out || (!d_offsets)) return cl_fail(ctx, CL_E_INVALID, "cl_reads_pack: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	cl_reads* R = new cl_reads(); R->ctx = ctx; R->n_reads = n_reads;
	std::unique_ptr<cl_reads> guard(R);
	DEV_ALLOC(ctx, R->lens, n_reads); DEV_ALLOC(ctx, R->word_off, (uint64_t)n_reads + 1); DEV_ALLOC(ctx, R->has_n, n_reads);
	DevBuf<uint32_t> words; DEV_ALLOC(ctx, words, n_reads);
	DevBuf<uint32_t> err; DEV_ALLOC(ctx, err, 1);
	HIP_TRY(ctx, hipMemsetAsync(err.p, 0, 4, ctx->stream));
	if (n_reads)
	{
		LAUNCH(ctx, k_read_geometry, grid_for(n_reads, 256), 256, d_offsets, n_reads, R->lens.p, words.p, err.p);
		HIP_TRY(ctx, hipGetLastError());
	}
	CL_TRY(dev_exclusive_scan_u64(ctx, words.p, R->word_off.p, n_reads, &R->total_words));
	uint64_t first = 0, last = 0;
	if (n_reads)
	{
		HIP_TRY(ctx, hipMemcpyAsync(&first, d_offsets, 8, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(&last, d_offsets + n_reads, 8, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	R->total_bases = last - first;
	DEV_ALLOC(ctx, R->packed, R->total_words + 1); DEV_ALLOC(ctx, R->inv, R->total_words + 1);
	if (n_reads)
	{
		LAUNCH(ctx, k_pack, grid_for(n_reads, 4), 256, d_codes, d_offsets, (const uint64_t*)R->word_off.p,
			(const uint32_t*)R->lens.p, n_reads, ascii, R->packed.p, R->inv.p, R->has_n.p, err.p);
	}
	HIP_TRY(ctx, hipGetLastError());
	LAUNCH(ctx, k_arena_tail, 1, 1, R->packed.p, R->inv.p, R->total_words);
	uint32_t herr = 0;
	HIP_TRY(ctx, hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	cl_timing_collect(ctx);
	if (herr & 1) return cl_fail(ctx, CL_E_INVALID, "cl_reads_pack: offsets not monotone or read too long");
	if (herr & 2) return cl_fail(ctx, CL_E_INVALID, "Only ACGTN symbols supported inside a read");   // in_reads.cpp:31-35
	*out = guard.release();
	return CL_OK;
}
extern "C" void cl_reads_free(cl_reads* r) { delete r; }
extern "C" uint32_t cl_reads_count(const cl_reads* r) { return r->n_reads; }
extern "C" uint64_t cl_reads_total_bases(const cl_reads* r) { return r->total_bases; }
extern "C" uint64_t cl_reads_total_words(const cl_reads* r) { return r->total_words; }
extern "C" const uint64_t* cl_reads_packed(const cl_reads* r) { return r->packed.p; }
extern "C" const uint32_t* cl_reads_invalid(const cl_reads* r) { return r->inv.p; }
extern "C" const uint64_t* cl_reads_word_offsets(const cl_reads* r) { return r->word_off.p; }
extern "C" const uint32_t* cl_reads_lengths(const cl_reads* r) { return r->lens.p; }
extern "C" const uint8_t* cl_reads_has_n(const cl_reads* r) { return r->has_n.p; }

// CReferenceReads (reference_reads.h): the arena of the reads the acceptor kept, in reference-id order
namespace {
__global__ void k_select_geometry(const uint8_t* __restrict__ keep, const uint32_t* __restrict__ lens, uint32_t n, uint32_t* __restrict__ flag, uint32_t* __restrict__ words)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t kp = keep[i] ? 1u : 0u;
	flag[i] = kp; words[i] = kp ? lens[i] / 32 + 1 : 0u;       // arena layout: len/32 + 1 words per read (cl_reads_pack)
}
__global__ void k_select_copy(const uint8_t* __restrict__ keep, const uint32_t* __restrict__ rank, const uint64_t* __restrict__ new_off, const uint64_t* __restrict__ old_off,
                              const uint32_t* __restrict__ lens, const uint8_t* __restrict__ has_n, const uint64_t* __restrict__ packed, const uint32_t* __restrict__ inv, uint32_t n,
                              uint64_t* __restrict__ o_off, uint32_t* __restrict__ o_lens, uint8_t* __restrict__ o_has_n, uint64_t* __restrict__ o_packed, uint32_t* __restrict__ o_inv)
{	// one wave per read
	const uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
	if (r >= n || !keep[r]) return;
	const uint32_t j = rank[r]; const uint64_t so = old_off[r], d = new_off[r]; const uint32_t w = lens[r] / 32 + 1;
	if (lane == 0) { o_off[j] = d; o_lens[j] = lens[r]; o_has_n[j] = has_n[r]; }
	for (uint32_t i = lane; i < w; i += 64) { o_packed[d + i] = packed[so + i]; o_inv[d + i] = inv[so + i]; }
}
} // namespace
extern "C" cl_status cl_reads_select(cl_ctx* ctx, const cl_reads* src, const uint8_t* d_keep, cl_reads** out)
{
	if (!ctx || !src || !d_keep || !out) return cl_fail(ctx, CL_E_INVALID, "cl_reads_select: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const uint32_t n = src->n_reads;
	cl_reads* R = new cl_reads(); R->ctx = ctx;
	std::unique_ptr<cl_reads> guard(R);
	DevBuf<uint32_t> flag, words; DEV_ALLOC(ctx, flag, (uint64_t)n + 1); DEV_ALLOC(ctx, words, (uint64_t)n + 1);
	DevBuf<uint64_t> new_off; DEV_ALLOC(ctx, new_off, (uint64_t)n + 1);
	uint64_t kept = 0;
	if (n)
	{
		LAUNCH(ctx, k_select_geometry, grid_for(n, 256), 256, d_keep, (const uint32_t*)src->lens.p, n, flag.p, words.p);
		HIP_TRY(ctx, hipGetLastError());
	}
	CL_TRY(dev_exclusive_scan_u64(ctx, words.p, new_off.p, n, &R->total_words));
	CL_TRY(dev_exclusive_scan_u32(ctx, flag.p, n, &kept));
	R->n_reads = (uint32_t)kept;
	DEV_ALLOC(ctx, R->lens, kept); DEV_ALLOC(ctx, R->word_off, kept + 1); DEV_ALLOC(ctx, R->has_n, kept);
	DEV_ALLOC(ctx, R->packed, R->total_words + 1); DEV_ALLOC(ctx, R->inv, R->total_words + 1);
	if (n)
	{
		LAUNCHB(ctx, R->total_words * 24.0, k_select_copy, grid_for((uint64_t)n * 64, 256), 256, d_keep, (const uint32_t*)flag.p, (const uint64_t*)new_off.p, (const uint64_t*)src->word_off.p,
			(const uint32_t*)src->lens.p, (const uint8_t*)src->has_n.p, (const uint64_t*)src->packed.p, (const uint32_t*)src->inv.p, n,
			R->word_off.p, R->lens.p, R->has_n.p, R->packed.p, R->inv.p);
		HIP_TRY(ctx, hipGetLastError());
	}
	HIP_TRY(ctx, hipMemcpyAsync(R->word_off.p + kept, &R->total_words, 8, hipMemcpyHostToDevice, ctx->stream));
	LAUNCH(ctx, k_arena_tail, 1, 1, R->packed.p, R->inv.p, R->total_words);
	if (kept)
	{	// total bases = sum of the kept lengths
		std::vector<uint32_t> h(kept);
		HIP_TRY(ctx, hipMemcpyAsync(h.data(), R->lens.p, kept * 4, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		uint64_t tb = 0; for (uint32_t l : h) tb += l;
		R->total_bases = tb;
	}
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	cl_timing_collect(ctx);
	*out = guard.release();
	return CL_OK;
}

namespace {
__global__ void k_arena_geometry(const uint32_t* __restrict__ lens, uint32_t n, uint32_t* __restrict__ words)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) words[i] = lens[i] / 32 + 1;
}
__global__ void k_arena_has_n(const uint64_t* __restrict__ off, const uint32_t* __restrict__ lens, const uint32_t* __restrict__ inv, uint32_t n, uint8_t* __restrict__ has_n)
{	// one wave per read: any invalid bit inside the read's length
	const uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
	if (r >= n) return;
	const uint32_t len = lens[r], w = len / 32 + 1; const uint64_t o = off[r];
	bool any = false;
	for (uint32_t i = lane; i < w; i += 64)
	{
		uint32_t m = inv[o + i];
		if (i == w - 1) m &= (len & 31) ? ~0u << (32 - (len & 31)) : 0u;       // ignore the pad bits of the last word
		any |= m != 0;
	}
	const uint64_t b = __ballot(any);
	if (lane == 0) has_n[r] = b ? 1 : 0;
}
} // namespace
// An arena from packed words that already have the arena layout (word-aligned reads back to back), e.g. the
// concatenation of the reference-read arenas of several ranks.
extern "C" cl_status cl_reads_from_arena(cl_ctx* ctx, const uint64_t* d_packed, const uint32_t* d_inv, const uint32_t* d_lens, uint32_t n_reads, cl_reads** out)
{
	if (!ctx || !out || (n_reads && (!d_packed || !d_inv || !d_lens))) return cl_fail(ctx, CL_E_INVALID, "cl_reads_from_arena: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	cl_reads* R = new cl_reads(); R->ctx = ctx; R->n_reads = n_reads;
	std::unique_ptr<cl_reads> guard(R);
	DEV_ALLOC(ctx, R->lens, n_reads); DEV_ALLOC(ctx, R->word_off, (uint64_t)n_reads + 1); DEV_ALLOC(ctx, R->has_n, n_reads);
	DevBuf<uint32_t> words; DEV_ALLOC(ctx, words, (uint64_t)n_reads + 1);
	if (n_reads)
	{
		HIP_TRY(ctx, hipMemcpyAsync(R->lens.p, d_lens, (uint64_t)n_reads * 4, hipMemcpyDeviceToDevice, ctx->stream));
		LAUNCH(ctx, k_arena_geometry, grid_for(n_reads, 256), 256, d_lens, n_reads, words.p);
		HIP_TRY(ctx, hipGetLastError());
	}
	CL_TRY(dev_exclusive_scan_u64(ctx, words.p, R->word_off.p, n_reads, &R->total_words));
	DEV_ALLOC(ctx, R->packed, R->total_words + 1); DEV_ALLOC(ctx, R->inv, R->total_words + 1);
	if (R->total_words)
	{
		HIP_TRY(ctx, hipMemcpyAsync(R->packed.p, d_packed, R->total_words * 8, hipMemcpyDeviceToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(R->inv.p, d_inv, R->total_words * 4, hipMemcpyDeviceToDevice, ctx->stream));
	}
	LAUNCH(ctx, k_arena_tail, 1, 1, R->packed.p, R->inv.p, R->total_words);
	if (n_reads)
	{
		LAUNCH(ctx, k_arena_has_n, grid_for((uint64_t)n_reads * 64, 256), 256, (const uint64_t*)R->word_off.p, (const uint32_t*)R->lens.p, (const uint32_t*)R->inv.p, n_reads, R->has_n.p);
		HIP_TRY(ctx, hipGetLastError());
		std::vector<uint32_t> h(n_reads);
		HIP_TRY(ctx, hipMemcpyAsync(h.data(), R->lens.p, (uint64_t)n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		uint64_t tb = 0; for (uint32_t l : h) tb += l;
		R->total_bases = tb;
	}
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	cl_timing_collect(ctx);
	*out = guard.release();
	return CL_OK;
}

extern "C" cl_status cl_reads_compact(cl_ctx* ctx, const cl_reads* R, uint32_t read, uint8_t* h_out, uint64_t cap, uint64_t* n_out)
{
	if (!ctx || !R || read >= R->n_reads) return cl_fail(ctx, CL_E_INVALID, "cl_reads_compact: bad read index");
	uint32_t len = 0; uint64_t wo = 0;
	HIP_TRY(ctx, hipMemcpy(&len, R->lens.p + read, 4, hipMemcpyDeviceToHost));
	HIP_TRY(ctx, hipMemcpy(&wo, R->word_off.p + read, 8, hipMemcpyDeviceToHost));
	uint64_t nb = ((uint64_t)len + 3) / 4;
	if (n_out) *n_out = nb + 1;
	if (cap < nb + 1) return cl_fail(ctx, CL_E_CAPACITY, "cl_reads_compact: buffer too small");
	std::vector<uint64_t> w(len / 32 + 1);
	HIP_TRY(ctx, hipMemcpy(w.data(), R->packed.p + wo, w.size() * 8, hipMemcpyDeviceToHost));
	// arena words are MSB-first, i.e. the big-endian byte image of a word IS the reference's 4-bases/byte layout
	for (uint64_t b = 0; b < nb; ++b) h_out[b] = (uint8_t)(w[b / 8] >> (56 - 8 * (b % 8)));
	if (len % 4) h_out[nb - 1] &= (uint8_t)(0xff << (8 - 2 * (len % 4)));
	h_out[nb] = (uint8_t)(len % 4);
	return CL_OK;
}

// ======================================================================================================
// a1: canonical k-mer scan of one 32-base word (positions p = 0..31 start in word w, may extend into w+1)
// ======================================================================================================
namespace {

__device__ inline uint64_t revcomp_k(uint64_t x, uint32_t k)
{
	x = ~x;                                               // complement (A<->T, C<->G) of every 2-bit group
	x = __brevll(x);                                      // reverse all bits: groups reversed, bits inside a group swapped
	x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
	return x >> (64 - 2 * k);
}
// forward k-mer starting at base p of the 64-base window (hi:lo)
__device__ inline uint64_t fwd_at(uint64_t hi, uint64_t lo, uint32_t p, uint32_t k, uint64_t mask)
{
	uint32_t s = 128 - 2 * (p + k);                       // right shift of the 128-bit window
	uint64_t v = (s >= 64) ? (hi >> (s - 64)) : ((hi << (64 - s)) | (lo >> s));
	return v & mask;
}
__device__ inline bool window_valid(uint64_t inv64, uint32_t p, uint32_t k)
{
	return ((inv64 >> (64 - (p + k))) & ((1ULL << k) - 1)) == 0;
}

// Returns the 32-bit mask (bit p) of start positions whose canonical k-mer passes the modulo test.
__device__ inline uint32_t scan_word(uint64_t hi, uint64_t lo, uint64_t inv64, uint32_t k, const ModTest& mt)
{
	if ((inv64 >> 32) == 0xffffffffull) return 0;
	const uint64_t mask = (1ULL << (2 * k)) - 1;
	const uint32_t roff = 2 * (k - 1);
	uint64_t fwd = hi >> (64 - 2 * k);
	uint64_t rev = revcomp_k(fwd, k);
	uint32_t res = 0;
#pragma unroll 8
	for (uint32_t p = 0; p < 32; ++p)
	{
		if (window_valid(inv64, p, k))
		{
			uint64_t can = fwd < rev ? fwd : rev;
			if (mod_is_zero(hash_mm(can), mt)) res |= 1u << p;
		}
		uint32_t j = p + k;                               // next base index in the window, < 64
		uint64_t src = j < 32 ? hi : lo;
		uint64_t b = (src >> (62 - 2 * (j & 31))) & 3;
		fwd = ((fwd << 2) | b) & mask;
		rev = (rev >> 2) | ((3 - b) << roff);
	}
	return res;
}
__device__ inline uint64_t canonical_at(uint64_t hi, uint64_t lo, uint32_t p, uint32_t k)
{
	const uint64_t mask = (1ULL << (2 * k)) - 1;
	uint64_t f = fwd_at(hi, lo, p, k, mask);
	uint64_t r = revcomp_k(f, k);
	return f < r ? f : r;
}

__global__ __launch_bounds__(256) void k_kmer_scan(const uint64_t* __restrict__ packed, const uint32_t* __restrict__ inv,
                                                   uint64_t total_words, uint32_t k, ModTest mt,
                                                   uint64_t* __restrict__ out, uint64_t cap, unsigned long long* __restrict__ counter)
{
	__shared__ uint32_t sh[4];
	__shared__ unsigned long long sbase;
	uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	uint64_t hi = 0, lo = 0, inv64 = ~0ULL; uint32_t m = 0;
	if (w < total_words)
	{
		hi = packed[w]; lo = packed[w + 1];
		inv64 = ((uint64_t)inv[w] << 32) | inv[w + 1];
		m = scan_word(hi, lo, inv64, k, mt);
	}
	uint32_t cnt = __popc(m), total;
	uint32_t ex = block_excl_scan_256(cnt, sh, &total);
	if (threadIdx.x == 0) sbase = total ? atomicAdd(counter, (unsigned long long)total) : 0ULL;
	__syncthreads();
	uint64_t o = sbase + ex;
	while (m)
	{
		uint32_t p = __ffs(m) - 1; m &= m - 1;
		if (o < cap) out[o] = canonical_at(hi, lo, p, k);
		++o;
	}
}
} // namespace

extern "C" cl_status cl_kmer_scan(cl_ctx* ctx, const cl_reads* R, uint32_t k, uint32_t f, uint64_t* d_out, uint64_t cap, uint64_t* n_out)
{
	if (!ctx || !R || !n_out) return cl_fail(ctx, CL_E_INVALID, "cl_kmer_scan: null argument");
	if (k < 1 || k > 28 || f < 1) return cl_fail(ctx, CL_E_INVALID, "cl_kmer_scan: need 1 <= k <= 28 and f >= 1");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	cl_timing_begin(ctx);
	DevBuf<unsigned long long> counter; DEV_ALLOC(ctx, counter, 1);
	HIP_TRY(ctx, hipMemsetAsync(counter.p, 0, 8, ctx->stream));
	if (R->total_words)
	{
		LAUNCHB(ctx, R->total_bases / 4.0 + R->total_bases / 8.0 + 8.0 * cap_hint(R->total_bases, f), k_kmer_scan, grid_for(R->total_words, 256), 256,
			(const uint64_t*)R->packed.p, (const uint32_t*)R->inv.p, R->total_words, k, make_modtest(f), d_out, cap, counter.p);
	}
	HIP_TRY(ctx, hipGetLastError());
	unsigned long long n = 0;
	HIP_TRY(ctx, hipMemcpyAsync(&n, counter.p, 8, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	cl_timing_collect(ctx);
	*n_out = n;
	if (n > cap) return cl_fail(ctx, CL_E_CAPACITY, "cl_kmer_scan: output capacity " + std::to_string(cap) + " < " + std::to_string(n));
	return CL_OK;
}

// ======================================================================================================
// a2: exact counts by sort + run lengths; a3: bucketed open-addressing membership table
// ======================================================================================================
namespace {

__global__ void k_head_flags(const uint64_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ flags)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// flags already exclusive-scanned in place: position i is a run head iff it is the last element or scan[i+1] != scan[i]
__global__ void k_scatter_heads(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ scan, uint64_t n, uint64_t n_heads,
                                uint32_t* __restrict__ head_pos)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t s = scan[i];
	uint32_t nx = (i + 1 < n) ? scan[i + 1] : (uint32_t)n_heads;
	if (nx != s) head_pos[s] = (uint32_t)i;
}
__global__ void k_count_flags(const uint32_t* __restrict__ head_pos, uint64_t n_heads, uint64_t n, uint32_t ci,
                              uint32_t* __restrict__ flags)
{
	uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n_heads) return;
	uint64_t end = (j + 1 < n_heads) ? head_pos[j + 1] : n;
	flags[j] = (end - head_pos[j] >= ci) ? 1u : 0u;
}
__global__ void k_scatter_kept(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ head_pos, const uint32_t* __restrict__ scan,
                               uint64_t n_heads, uint64_t n_kept, uint64_t n, uint32_t cs,
                               uint64_t* __restrict__ kept_keys, uint32_t* __restrict__ kept_counts)
{
	uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t c = 0;
	if (j < n_heads)
	{
		uint32_t s = scan[j];
		uint32_t nx = (j + 1 < n_heads) ? scan[j + 1] : (uint32_t)n_kept;
		if (nx != s)
		{
			uint64_t end = (j + 1 < n_heads) ? head_pos[j + 1] : n;
			uint64_t cnt = end - head_pos[j];
			c = cnt > cs ? cs : (uint32_t)cnt;             // saturating counter ("-cs", kb_sorter.h:1000-1060)
			kept_keys[s] = keys[head_pos[j]]; kept_counts[s] = c;
		}
	}
}
// grid-stride sum of a uint32 array into one uint64 (one atomic per block)
__global__ __launch_bounds__(256) void k_sum_u32(const uint32_t* __restrict__ v, uint64_t n, unsigned long long* __restrict__ sum)
{
	__shared__ unsigned long long sh[4];
	unsigned long long s = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) s += v[i];
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) atomicAdd(sum, sh[0] + sh[1] + sh[2] + sh[3]);
}

// ---- membership table: buckets of 4 x {key, value} = 64 bytes, load <= 0.5, linear probing over buckets
struct Slot { uint64_t key, val; };
constexpr uint64_t SLOT_EMPTY = ~0ULL;
__device__ inline uint64_t bucket_of(uint64_t key, uint64_t bmask) { return (hash_mm(key) >> 20) & bmask; }

__global__ void k_table_build(const uint64_t* __restrict__ keys, uint64_t n, Slot* __restrict__ slots, uint64_t bmask)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint64_t key = keys[i];
	uint64_t b = bucket_of(key, bmask);
	for (;;)
	{
		for (uint32_t s = 0; s < 4; ++s)
		{
			unsigned long long* kp = (unsigned long long*)&slots[b * 4 + s].key;
			unsigned long long old = atomicCAS(kp, (unsigned long long)SLOT_EMPTY, (unsigned long long)key);
			if (old == SLOT_EMPTY) { slots[b * 4 + s].val = i; return; }
		}
		b = (b + 1) & bmask;
	}
}
} // namespace

// returns rank of key in the sorted kept array, or ~0u
__device__ uint32_t table_lookup(const void* slots_v, uint64_t bmask, uint64_t key)
{
	const Slot* slots = (const Slot*)slots_v;
	uint64_t b = bucket_of(key, bmask);
	for (;;)
	{
		const ulonglong2* bp = (const ulonglong2*)(slots + b * 4);
		ulonglong2 s0 = bp[0], s1 = bp[1], s2 = bp[2], s3 = bp[3];
		if (s0.x == key) return (uint32_t)s0.y;
		if (s1.x == key) return (uint32_t)s1.y;
		if (s2.x == key) return (uint32_t)s2.y;
		if (s3.x == key) return (uint32_t)s3.y;
		if (s0.x == SLOT_EMPTY || s1.x == SLOT_EMPTY || s2.x == SLOT_EMPTY || s3.x == SLOT_EMPTY) return ~0u;
		b = (b + 1) & bmask;
	}
}

namespace {
__global__ void k_table_check(const void* slots, uint64_t bmask, const uint64_t* __restrict__ kmers, uint64_t n, uint8_t* __restrict__ found)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) found[i] = table_lookup(slots, bmask, kmers[i]) != ~0u;
}
} // namespace

static cl_status build_table(cl_ctx* ctx, cl_kmer_set* S);

namespace {
// ---- key-range partitioning: inputs of 2^32 and more k-mers (C5: 4.2 G) are counted range by range, and the ranks of a
// multi-GPU run own one range each.  Ranges are unions of the 4096 bins of the keys' top 12 bits, so the kept keys of
// consecutive ranges are ascending as a whole (the rank of a key in that order is its id everywhere downstream).
constexpr uint32_t PART_BITS = 12, PART_BINS = 1u << PART_BITS;
__global__ __launch_bounds__(256) void k_key_hist(const uint64_t* __restrict__ keys, uint64_t n, uint32_t shift, unsigned long long* __restrict__ bins)
{
	__shared__ uint32_t sh[PART_BINS];
	for (uint32_t i = threadIdx.x; i < PART_BINS; i += 256) sh[i] = 0;
	__syncthreads();
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
		atomicAdd(&sh[(uint32_t)(keys[i] >> shift) & (PART_BINS - 1)], 1u);
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < PART_BINS; i += 256) if (sh[i]) atomicAdd(&bins[i], (unsigned long long)sh[i]);
}
// keys whose bin lies in [b0, b1) are appended to out (any order: they are sorted next).  A block takes 4096 keys and ONE place in the
// output (round 6: a block of 256 keys — 16 M returning atomics on one word per launch of the 50-Gbase pass, and one word takes ~88 per
// microsecond, MI355X_MICROARCH.md `dequeue`: 195 ms per launch where the 33 GB it reads are 10 ms; four launches per pass, all of them
// in pass 1, which nothing overlaps).
constexpr uint32_t GATHER_ITEMS = 16;
__global__ __launch_bounds__(256) void k_key_gather(const uint64_t* __restrict__ keys, uint64_t n, uint32_t shift, uint32_t b0, uint32_t b1,
                                                    uint64_t* __restrict__ out, unsigned long long* __restrict__ counter)
{
	__shared__ uint32_t sh[4];
	__shared__ unsigned long long sbase;
	const uint64_t base = (uint64_t)blockIdx.x * (256 * GATHER_ITEMS) + threadIdx.x;
	uint64_t key[GATHER_ITEMS]; uint32_t mask = 0, mine = 0;
#pragma unroll
	for (uint32_t j = 0; j < GATHER_ITEMS; ++j)
	{
		const uint64_t i = base + (uint64_t)j * 256;
		key[j] = i < n ? keys[i] : 0;
		const uint32_t b = (uint32_t)(key[j] >> shift) & (PART_BINS - 1);
		const uint32_t take = i < n && b >= b0 && b < b1;
		mask |= take << j; mine += take;
	}
	uint32_t total;
	const uint32_t ex = block_excl_scan_256(mine, sh, &total);
	if (threadIdx.x == 0) sbase = total ? atomicAdd(counter, (unsigned long long)total) : 0ULL;
	__syncthreads();
	uint64_t at = sbase + ex;
#pragma unroll
	for (uint32_t j = 0; j < GATHER_ITEMS; ++j) if ((mask >> j) & 1u) out[at++] = key[j];
}
struct CountPiece { DevBuf<uint64_t> keys; DevBuf<uint32_t> counts; uint64_t n = 0; };
} // namespace

uint32_t cl_part_shift(uint32_t k) { return 2 * k > PART_BITS ? 2 * k - PART_BITS : 0; }
// histogram of the keys over the 4096 top-bit bins (host copy)
cl_status cl_key_histogram(cl_ctx* ctx, const uint64_t* d_kmers, uint64_t n, uint32_t k, std::vector<uint64_t>& h_bins)
{
	DevBuf<unsigned long long> bins; DEV_ALLOC(ctx, bins, PART_BINS);
	HIP_TRY(ctx, hipMemsetAsync(bins.p, 0, PART_BINS * 8, ctx->stream));
	if (n) LAUNCHB(ctx, n * 8.0, k_key_hist, (uint32_t)std::min<uint64_t>(4096, grid_for(n, 256)), 256, d_kmers, n, cl_part_shift(k), bins.p);
	HIP_TRY(ctx, hipGetLastError());
	h_bins.resize(PART_BINS);
	HIP_TRY(ctx, hipMemcpyAsync(h_bins.data(), bins.p, PART_BINS * 8, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CL_OK;
}
// the keys of bins [b0, b1) copied to d_out (which holds the `expect` keys the histogram counted for them)
cl_status cl_key_gather(cl_ctx* ctx, const uint64_t* d_kmers, uint64_t n, uint32_t k, uint32_t b0, uint32_t b1, uint64_t* d_out, uint64_t expect)
{
	DevBuf<unsigned long long> counter; DEV_ALLOC(ctx, counter, 1);
	HIP_TRY(ctx, hipMemsetAsync(counter.p, 0, 8, ctx->stream));
	if (n) LAUNCHB(ctx, n * 8.0 + expect * 8.0, k_key_gather, grid_for(n, 256 * GATHER_ITEMS), 256, d_kmers, n, cl_part_shift(k), b0, b1, d_out, counter.p);
	HIP_TRY(ctx, hipGetLastError());
	unsigned long long got = 0;
	HIP_TRY(ctx, hipMemcpyAsync(&got, counter.p, 8, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	if (got != expect) return cl_fail(ctx, CL_E_INVALID, "cl_key_gather: histogram and gather disagree");
	return CL_OK;
}

// exact counts of one key range that fits a single sort: d_kmers (n < 2^32) is sorted in place
static cl_status count_range(cl_ctx* ctx, uint64_t* d_kmers, uint64_t n, uint32_t k, uint32_t ci, uint32_t cs, CountPiece& out, uint64_t* n_unique, uint64_t* filt_out)
{
	uint64_t n_heads = 0, n_kept = 0; unsigned long long filt = 0;
	if (n)
	{
		CL_TRY(dev_sort_pairs(ctx, d_kmers, nullptr, n, 0, 2 * k));
		DevBuf<uint32_t> flags; DEV_ALLOC(ctx, flags, n);
		{ LAUNCH(ctx, k_head_flags, grid_for(n, 256), 256, (const uint64_t*)d_kmers, n, flags.p); }
		HIP_TRY(ctx, hipGetLastError());
		CL_TRY(dev_exclusive_scan_u32(ctx, flags.p, n, &n_heads));
		DevBuf<uint32_t> head_pos; DEV_ALLOC(ctx, head_pos, n_heads);
		{ LAUNCH(ctx, k_scatter_heads, grid_for(n, 256), 256, (const uint64_t*)d_kmers, (const uint32_t*)flags.p, n, n_heads, head_pos.p); }
		HIP_TRY(ctx, hipGetLastError());
		flags.release();
		DevBuf<uint32_t> kflags; DEV_ALLOC(ctx, kflags, n_heads);
		LAUNCH(ctx, k_count_flags, grid_for(n_heads, 256), 256, (const uint32_t*)head_pos.p, n_heads, n, ci, kflags.p);
		HIP_TRY(ctx, hipGetLastError());
		CL_TRY(dev_exclusive_scan_u32(ctx, kflags.p, n_heads, &n_kept));
		DEV_ALLOC(ctx, out.keys, n_kept); DEV_ALLOC(ctx, out.counts, n_kept);
		DevBuf<unsigned long long> sum; DEV_ALLOC(ctx, sum, 1);
		HIP_TRY(ctx, hipMemsetAsync(sum.p, 0, 8, ctx->stream));
		LAUNCH(ctx, k_scatter_kept, grid_for(n_heads, 256), 256, (const uint64_t*)d_kmers, (const uint32_t*)head_pos.p,
			(const uint32_t*)kflags.p, n_heads, n_kept, n, cs, out.keys.p, out.counts.p);
		HIP_TRY(ctx, hipGetLastError());
		if (n_kept) LAUNCH(ctx, k_sum_u32, (uint32_t)std::min<uint64_t>(1024, grid_for(n_kept, 256)), 256, (const uint32_t*)out.counts.p, n_kept, sum.p);
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipMemcpyAsync(&filt, sum.p, 8, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	out.n = n_kept; *n_unique += n_heads; *filt_out += filt;
	return CL_OK;
}

extern "C" cl_status cl_kmer_count_filter(cl_ctx* ctx, uint64_t* d_kmers, uint64_t n, uint32_t k, uint32_t ci, uint32_t cs,
                                          cl_kmer_set** out, cl_kmer_stats* stats)
{
	if (!ctx || !out) return cl_fail(ctx, CL_E_INVALID, "cl_kmer_count_filter: null argument");
	if (k < 1 || k > 28) return cl_fail(ctx, CL_E_INVALID, "cl_kmer_count_filter: need 1 <= k <= 28");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	cl_timing_begin(ctx);
	cl_kmer_set* S = new cl_kmer_set(); S->ctx = ctx; S->k = k;
	std::unique_ptr<cl_kmer_set> guard(S);
	uint64_t n_unique = 0, filt = 0;
	// one sort handles < 2^32 keys and needs a second buffer of the same size: above the limit the input is counted
	// one key range after the other (COLORD_HIP_COUNT_LIMIT lowers it so that small test inputs take this path)
	uint64_t limit = 1ull << 30;
	if (const char* e = getenv("COLORD_HIP_COUNT_LIMIT")) { const uint64_t v = strtoull(e, nullptr, 10); if (v) limit = v; }
	if (n <= limit)
	{
		CountPiece pc;
		CL_TRY(count_range(ctx, d_kmers, n, k, ci, cs, pc, &n_unique, &filt));
		S->keys = std::move(pc.keys); S->counts = std::move(pc.counts); S->n = pc.n;
		if (!n) { DEV_ALLOC(ctx, S->keys, 0); DEV_ALLOC(ctx, S->counts, 0); }
	}
	else
	{
		std::vector<uint64_t> bins;
		CL_TRY(cl_key_histogram(ctx, d_kmers, n, k, bins));
		std::vector<CountPiece> pieces;
		uint64_t total_kept = 0;
		for (uint32_t b0 = 0; b0 < PART_BINS; )
		{
			uint32_t b1 = b0; uint64_t cnt = 0;
			while (b1 < PART_BINS && (b1 == b0 || cnt + bins[b1] <= limit)) cnt += bins[b1++];
			if (cnt >= (1ull << 32)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_kmer_count_filter: one of the 4096 key ranges holds >= 2^32 k-mers");
			if (cnt)
			{
				DevBuf<uint64_t> work; DEV_ALLOC(ctx, work, cnt);
				CL_TRY(cl_key_gather(ctx, d_kmers, n, k, b0, b1, work.p, cnt));
				pieces.emplace_back();
				CL_TRY(count_range(ctx, work.p, cnt, k, ci, cs, pieces.back(), &n_unique, &filt));
				total_kept += pieces.back().n;
			}
			b0 = b1;
		}
		if (total_kept >= (1ull << 32)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_kmer_count_filter: >= 2^32 kept k-mers (ids are 32-bit)");
		DEV_ALLOC(ctx, S->keys, total_kept); DEV_ALLOC(ctx, S->counts, total_kept);
		uint64_t o = 0;
		for (auto& pc : pieces)
		{
			if (!pc.n) continue;
			HIP_TRY(ctx, hipMemcpyAsync(S->keys.p + o, pc.keys.p, pc.n * 8, hipMemcpyDeviceToDevice, ctx->stream));
			HIP_TRY(ctx, hipMemcpyAsync(S->counts.p + o, pc.counts.p, pc.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
			o += pc.n;
		}
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		S->n = total_kept;
	}
	CL_TRY(build_table(ctx, S));          // a3: table with >= 2 slots per key
	cl_timing_collect(ctx);
	if (stats) { stats->tot_kmers = n; stats->n_unique = n_unique; stats->n_unique_counted = S->n; stats->total_count_filtered = filt; }
	*out = guard.release();
	return CL_OK;
}
// Builds a set object from already counted keys (ascending, distinct) — used to replicate the filtered set
// on every GPU after the all-gather of the per-rank partitions.
static cl_status build_table(cl_ctx* ctx, cl_kmer_set* S)
{
	uint64_t nbuckets = 16; while (nbuckets * 4 < 2 * S->n) nbuckets <<= 1;
	S->bmask = nbuckets - 1;
	DEV_ALLOC(ctx, S->slots, nbuckets * 4 * 2);
	HIP_TRY(ctx, hipMemsetAsync(S->slots.p, 0xff, nbuckets * 64, ctx->stream));
	if (S->n)
	{
		LAUNCH(ctx, k_table_build, grid_for(S->n, 256), 256, (const uint64_t*)S->keys.p, S->n, (Slot*)S->slots.p, S->bmask);
	}
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CL_OK;
}
extern "C" cl_status cl_kmer_set_create(cl_ctx* ctx, const uint64_t* d_keys, const uint32_t* d_counts, uint64_t n, uint32_t k, cl_kmer_set** out)
{
	if (!ctx || !out || (n && (!d_keys || !d_counts))) return cl_fail(ctx, CL_E_INVALID, "cl_kmer_set_create: null argument");
	if (k < 1 || k > 28 || n >= (1ULL << 32)) return cl_fail(ctx, CL_E_INVALID, "cl_kmer_set_create: need 1 <= k <= 28, n < 2^32");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	cl_timing_begin(ctx);
	cl_kmer_set* S = new cl_kmer_set(); S->ctx = ctx; S->k = k; S->n = n;
	std::unique_ptr<cl_kmer_set> guard(S);
	DEV_ALLOC(ctx, S->keys, n); DEV_ALLOC(ctx, S->counts, n);
	if (n)
	{
		HIP_TRY(ctx, hipMemcpyAsync(S->keys.p, d_keys, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(S->counts.p, d_counts, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
	}
	CL_TRY(build_table(ctx, S));
	cl_timing_collect(ctx);
	*out = guard.release();
	return CL_OK;
}
extern "C" void cl_kmer_set_free(cl_kmer_set* s) { delete s; }
extern "C" uint64_t cl_kmer_set_size(const cl_kmer_set* s) { return s->n; }
extern "C" const uint64_t* cl_kmer_set_keys(const cl_kmer_set* s) { return s->keys.p; }
extern "C" const uint32_t* cl_kmer_set_counts(const cl_kmer_set* s) { return s->counts.p; }
extern "C" cl_status cl_kmer_set_check(cl_ctx* ctx, const cl_kmer_set* S, const uint64_t* d_kmers, uint64_t n, uint8_t* d_found)
{
	if (!ctx || !S) return cl_fail(ctx, CL_E_INVALID, "cl_kmer_set_check: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (n) LAUNCH(ctx, k_table_check, grid_for(n, 256), 256, (const void*)S->slots.p, S->bmask, d_kmers, n, d_found);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CL_OK;
}

// ======================================================================================================
// a4: accepted k-mers per read
// ======================================================================================================
namespace {

// phase A: per word, mask of start positions whose canonical k-mer passes the modulo test AND is in the set
__global__ __launch_bounds__(256) void k_found_mask(const uint64_t* __restrict__ packed, const uint32_t* __restrict__ inv,
                                                    uint64_t total_words, uint32_t k, ModTest mt, const void* slots, uint64_t bmask,
                                                    uint32_t* __restrict__ fmask)
{
	uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (w >= total_words) return;
	uint64_t hi = packed[w], lo = packed[w + 1];
	uint64_t inv64 = ((uint64_t)inv[w] << 32) | inv[w + 1];
	uint32_t m = scan_word(hi, lo, inv64, k, mt), res = 0;
	while (m)
	{
		uint32_t p = __ffs(m) - 1; m &= m - 1;
		if (table_lookup(slots, bmask, canonical_at(hi, lo, p, k)) != ~0u) res |= 1u << p;
	}
	fmask[w] = res;
}
// per read: number of records (one wave per read)
__global__ __launch_bounds__(256) void k_read_counts(const uint32_t* __restrict__ fmask, const uint64_t* __restrict__ word_off,
                                                     const uint32_t* __restrict__ lens, const uint8_t* __restrict__ has_n,
                                                     uint32_t n_reads, uint32_t k, uint32_t* __restrict__ counts)
{
	uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n_reads) return;
	uint32_t lane = threadIdx.x & 63, c = 0;
	if (!has_n[r] && lens[r] >= k)                        // reads_sim_graph.cpp:142-146
	{
		uint64_t wb = word_off[r]; uint32_t nw = (uint32_t)(word_off[r + 1] - wb);
		for (uint32_t w = lane; w < nw; w += 64) c += __popc(fmask[wb + w]);
	}
	c = wave_sum(c);
	if (lane == 0) counts[r] = c;
}
// phase B: write records in (read, position) order
__global__ __launch_bounds__(256) void k_emit_records(const uint64_t* __restrict__ packed, const uint32_t* __restrict__ fmask,
                                                      const uint64_t* __restrict__ word_off, const uint64_t* __restrict__ rec_off,
                                                      uint32_t n_reads, uint32_t k, const void* slots, uint64_t bmask,
                                                      uint32_t* __restrict__ rec_id, uint32_t* __restrict__ rec_pos, uint32_t* __restrict__ rec_read)
{
	uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n_reads) return;
	uint64_t o = rec_off[r];
	if (rec_off[r + 1] == o) return;
	uint32_t lane = threadIdx.x & 63;
	uint64_t wb = word_off[r]; uint32_t nw = (uint32_t)(word_off[r + 1] - wb);
	for (uint32_t w0 = 0; w0 < nw; w0 += 64)
	{
		uint32_t w = w0 + lane;
		uint32_t m = (w < nw) ? fmask[wb + w] : 0u;
		uint32_t cnt = __popc(m);
		uint32_t incl = wave_incl_scan(cnt);
		uint32_t tot = __shfl(incl, 63, 64);
		uint64_t my = o + incl - cnt;
		if (m)
		{
			uint64_t hi = packed[wb + w], lo = packed[wb + w + 1];
			while (m)
			{
				uint32_t p = __ffs(m) - 1; m &= m - 1;
				rec_id[my] = table_lookup(slots, bmask, canonical_at(hi, lo, p, k));
				rec_pos[my] = w * 32 + p; rec_read[my] = r;
				++my;
			}
		}
		o += tot;
	}
}
__global__ void k_iota(uint32_t* v, uint64_t n) { uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] = (uint32_t)i; }
// sorted by id (stable): record j is a duplicate iff the previous record has the same id and the same read
__global__ void k_flag_first(const uint32_t* __restrict__ sid, const uint32_t* __restrict__ sidx, const uint32_t* __restrict__ rec_read,
                             uint64_t n, uint32_t* __restrict__ keep)
{
	uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	uint32_t me = sidx[j];
	bool dup = j > 0 && sid[j] == sid[j - 1] && rec_read[sidx[j - 1]] == rec_read[me];
	keep[me] = dup ? 0u : 1u;
}
__global__ void k_compact_records(const uint32_t* __restrict__ scan, uint64_t n, uint64_t n_keep, const uint32_t* __restrict__ rec_id,
                                  const uint32_t* __restrict__ rec_pos, const uint32_t* __restrict__ rec_read, const uint64_t* __restrict__ kept_keys,
                                  uint32_t* __restrict__ o_id, uint32_t* __restrict__ o_pos, uint32_t* __restrict__ o_read, uint64_t* __restrict__ o_kmer)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t s = scan[i];
	uint32_t nx = (i + 1 < n) ? scan[i + 1] : (uint32_t)n_keep;
	if (nx == s) return;
	uint32_t id = rec_id[i];
	o_id[s] = id; o_pos[s] = rec_pos[i]; o_read[s] = rec_read[i]; o_kmer[s] = kept_keys[id];
}
__global__ void k_remap_offsets(const uint64_t* __restrict__ rec_off, const uint32_t* __restrict__ scan, uint32_t n_reads, uint64_t n, uint64_t n_keep,
                                uint64_t* __restrict__ out_off)
{
	uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_reads) return;
	uint64_t o = rec_off[r];
	out_off[r] = (o < n) ? scan[o] : n_keep;
}
} // namespace

extern "C" cl_status cl_accepted_kmers(cl_ctx* ctx, const cl_kmer_set* S, const cl_reads* R, uint32_t k, uint32_t f, cl_kmer_lists** out)
{
	if (!ctx || !S || !R || !out) return cl_fail(ctx, CL_E_INVALID, "cl_accepted_kmers: null argument");
	if (k != S->k) return cl_fail(ctx, CL_E_INVALID, "cl_accepted_kmers: k differs from the set's k");
	if (f < 1) return cl_fail(ctx, CL_E_INVALID, "cl_accepted_kmers: f >= 1");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	cl_timing_begin(ctx);
	cl_kmer_lists* L = new cl_kmer_lists(); L->ctx = ctx; L->n_reads = R->n_reads;
	std::unique_ptr<cl_kmer_lists> guard(L);
	const uint32_t nr = R->n_reads;
	DevBuf<uint32_t> fmask; DEV_ALLOC(ctx, fmask, R->total_words);
	if (R->total_words)
	{
		LAUNCHB(ctx, R->total_bases / 4.0 + R->total_bases / 8.0 + 4.0 * (R->total_bases / 32.0) + 64.0 * R->total_bases / f, k_found_mask, grid_for(R->total_words, 256), 256, (const uint64_t*)R->packed.p, (const uint32_t*)R->inv.p,
			R->total_words, k, make_modtest(f), (const void*)S->slots.p, S->bmask, fmask.p);
	}
	HIP_TRY(ctx, hipGetLastError());
	DevBuf<uint32_t> counts; DEV_ALLOC(ctx, counts, nr);
	if (nr) LAUNCH(ctx, k_read_counts, grid_for(nr, 4), 256, (const uint32_t*)fmask.p, (const uint64_t*)R->word_off.p,
		(const uint32_t*)R->lens.p, (const uint8_t*)R->has_n.p, nr, k, counts.p);
	HIP_TRY(ctx, hipGetLastError());
	DevBuf<uint64_t> rec_off; DEV_ALLOC(ctx, rec_off, (uint64_t)nr + 1);
	uint64_t n_rec = 0;
	CL_TRY(dev_exclusive_scan_u64(ctx, counts.p, rec_off.p, nr, &n_rec));
	if (n_rec >= (1ULL << 32)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_accepted_kmers: more than 2^32 records in one arena; split the batch");
	DevBuf<uint32_t> rec_id, rec_pos, rec_read;
	DEV_ALLOC(ctx, rec_id, n_rec); DEV_ALLOC(ctx, rec_pos, n_rec); DEV_ALLOC(ctx, rec_read, n_rec);
	if (nr && n_rec)
	{
		LAUNCH(ctx, k_emit_records, grid_for(nr, 4), 256, (const uint64_t*)R->packed.p, (const uint32_t*)fmask.p,
			(const uint64_t*)R->word_off.p, (const uint64_t*)rec_off.p, nr, k, (const void*)S->slots.p, S->bmask, rec_id.p, rec_pos.p, rec_read.p);
	}
	HIP_TRY(ctx, hipGetLastError());
	fmask.release();
	// per-read dedup (first occurrence wins): stable sort of (id -> record index), flag repeats inside a read
	uint64_t n_keep = 0;
	DevBuf<uint32_t> keep; DEV_ALLOC(ctx, keep, n_rec);
	if (n_rec)
	{
		DevBuf<uint32_t> sid, sidx; DEV_ALLOC(ctx, sid, n_rec); DEV_ALLOC(ctx, sidx, n_rec);
		HIP_TRY(ctx, hipMemcpyAsync(sid.p, rec_id.p, n_rec * 4, hipMemcpyDeviceToDevice, ctx->stream));
		LAUNCH(ctx, k_iota, grid_for(n_rec, 256), 256, sidx.p, n_rec);
		uint32_t bits = 1; while (bits < 32 && (1ULL << bits) < S->n) ++bits;
		CL_TRY(dev_sort_keys32_pairs(ctx, sid.p, sidx.p, n_rec, 0, bits));
		LAUNCH(ctx, k_flag_first, grid_for(n_rec, 256), 256, (const uint32_t*)sid.p, (const uint32_t*)sidx.p,
			(const uint32_t*)rec_read.p, n_rec, keep.p);
		HIP_TRY(ctx, hipGetLastError());
		CL_TRY(dev_exclusive_scan_u32(ctx, keep.p, n_rec, &n_keep));
	}
	L->total = n_keep;
	DEV_ALLOC(ctx, L->ids, n_keep); DEV_ALLOC(ctx, L->pos, n_keep); DEV_ALLOC(ctx, L->read, n_keep); DEV_ALLOC(ctx, L->kmers, n_keep);
	DEV_ALLOC(ctx, L->off, (uint64_t)nr + 1);
	if (n_rec)
	{
		LAUNCH(ctx, k_compact_records, grid_for(n_rec, 256), 256, (const uint32_t*)keep.p, n_rec, n_keep,
			(const uint32_t*)rec_id.p, (const uint32_t*)rec_pos.p, (const uint32_t*)rec_read.p, (const uint64_t*)S->keys.p, L->ids.p, L->pos.p, L->read.p, L->kmers.p);
		HIP_TRY(ctx, hipGetLastError());
	}
	LAUNCH(ctx, k_remap_offsets, grid_for((uint64_t)nr + 1, 256), 256, (const uint64_t*)rec_off.p, (const uint32_t*)keep.p,
		nr, n_rec, n_keep, L->off.p);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	cl_timing_collect(ctx);
	*out = guard.release();
	return CL_OK;
}
extern "C" void cl_kmer_lists_free(cl_kmer_lists* l) { delete l; }
extern "C" uint32_t cl_kmer_lists_reads(const cl_kmer_lists* l) { return l->n_reads; }
extern "C" uint64_t cl_kmer_lists_total(const cl_kmer_lists* l) { return l->total; }
extern "C" const uint64_t* cl_kmer_lists_offsets(const cl_kmer_lists* l) { return l->off.p; }
extern "C" const uint64_t* cl_kmer_lists_kmers(const cl_kmer_lists* l) { return l->kmers.p; }
extern "C" const uint32_t* cl_kmer_lists_ids(const cl_kmer_lists* l) { return l->ids.p; }
extern "C" const uint32_t* cl_kmer_lists_pos(const cl_kmer_lists* l) { return l->pos.p; }
