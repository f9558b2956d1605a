// graph.hip — k-mer -> reference-reads index (a5 build) and candidate voting (a5 query) in the exact
// order-independent formulation: the list kept for a k-mer is "all pseudo-reads, then reference reads in
// id order while the list is shorter than maxKmerCount" (reads_sim_graph.cpp:306-317,381-393), and read i
// sees exactly the entries of earlier reference reads.  Both halves are sort + scan + gather work.
#include "common.hpp"
#include "objects.hpp"
#include <algorithm>

namespace {

__global__ void k_flags_from_bytes(const uint8_t* __restrict__ a, uint32_t n, uint32_t* __restrict__ f)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) f[i] = a[i] ? 1u : 0u;
}
// entry e of the lists belongs to an accepted read?
__global__ void k_entry_flags(const uint32_t* __restrict__ entry_read, const uint8_t* __restrict__ accept, uint64_t n, uint32_t* __restrict__ f)
{
	uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e < n) f[e] = accept[entry_read[e]] ? 1u : 0u;
}
__global__ void k_gather_ref_entries(const uint32_t* __restrict__ scan, uint64_t n, uint64_t n_sel, const uint32_t* __restrict__ ids,
                                     const uint32_t* __restrict__ entry_read, const uint32_t* __restrict__ ref_rank,
                                     uint32_t* __restrict__ o_id, uint32_t* __restrict__ o_ref)
{
	uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n) return;
	uint32_t s = scan[e];
	uint32_t nx = (e + 1 < n) ? scan[e + 1] : (uint32_t)n_sel;
	if (nx == s) return;
	o_id[s] = ids[e]; o_ref[s] = ref_rank[entry_read[e]];
}
__global__ void k_add_const(uint32_t* v, uint64_t n, uint32_t c) { uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] += c; }
__global__ void k_head_flags32(const uint32_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ flags)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// run_start[i] for sorted ids: position of the first element of i's run.  heads = inclusive count of heads
// up to i (from the exclusive scan + own flag); head_pos[h] = position of head h.
__global__ void k_scatter_head_pos(const uint32_t* __restrict__ scan, uint64_t n, uint64_t n_heads, uint32_t* __restrict__ head_pos)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t s = scan[i];
	uint32_t nx = (i + 1 < n) ? scan[i + 1] : (uint32_t)n_heads;
	if (nx != s) head_pos[s] = (uint32_t)i;
}
// keep entry i iff it is a pseudo-read entry or its rank inside the run is below the cap
__global__ void k_cap_flags(const uint32_t* __restrict__ sid, const uint32_t* __restrict__ sref, const uint32_t* __restrict__ scan,
                            const uint32_t* __restrict__ head_pos, uint64_t n, uint64_t n_heads, uint32_t n_pseudo, uint32_t cap,
                            uint32_t* __restrict__ keep)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t s = scan[i];
	uint32_t nx = (i + 1 < n) ? scan[i + 1] : (uint32_t)n_heads;
	uint32_t run = (nx != s) ? s : s - 1;                 // a head belongs to run s, a non-head to the previous head
	uint32_t rank = (uint32_t)i - head_pos[run];
	keep[i] = (sref[i] < n_pseudo || rank < cap) ? 1u : 0u;
}
__global__ void k_compact_index(const uint32_t* __restrict__ scan, uint64_t n, uint64_t n_keep, const uint32_t* __restrict__ sid,
                                const uint32_t* __restrict__ sref, uint32_t* __restrict__ o_ref, uint32_t* __restrict__ id_counts)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t s = scan[i];
	uint32_t nx = (i + 1 < n) ? scan[i + 1] : (uint32_t)n_keep;
	if (nx == s) return;
	o_ref[s] = sref[i];
	atomicAdd(&id_counts[sid[i]], 1u);
}

// number of list entries with ref id < bound (lists ascending)
__device__ inline uint32_t lower_bound_u32(const uint32_t* __restrict__ a, uint32_t n, uint32_t bound)
{
	uint32_t lo = 0, hi = n;
	while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (a[mid] < bound) lo = mid + 1; else hi = mid; }
	return lo;
}
__global__ void k_pair_counts(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ entry_read, uint64_t n,
                              const uint32_t* __restrict__ ref_rank, const uint64_t* __restrict__ ioff, const uint32_t* __restrict__ irefs,
                              uint32_t* __restrict__ cnt)
{
	uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n) return;
	uint32_t id = ids[e]; uint64_t a = ioff[id]; uint32_t len = (uint32_t)(ioff[id + 1] - a);
	cnt[e] = len ? lower_bound_u32(irefs + a, len, ref_rank[entry_read[e]]) : 0u;
}
// pairs of reads [r0, r1): key = (read - r0) << ref_bits | ref
__global__ void k_pair_fill(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ entry_read, uint64_t e0, uint64_t e1,
                            const uint32_t* __restrict__ cnt, const uint64_t* __restrict__ poff, uint64_t p0,
                            const uint64_t* __restrict__ ioff, const uint32_t* __restrict__ irefs, uint32_t r0, uint32_t ref_bits,
                            uint64_t* __restrict__ pairs)
{
	uint64_t e = e0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= e1) return;
	uint32_t c = cnt[e];
	if (!c) return;
	uint64_t a = ioff[ids[e]], o = poff[e] - p0;
	uint64_t hi = (uint64_t)(entry_read[e] - r0) << ref_bits;
	for (uint32_t j = 0; j < c; ++j) pairs[o + j] = hi | irefs[a + j];
}
__global__ void k_head_flags64(const uint64_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ flags)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// distinct (read, ref) with votes, in (read asc, ref asc) order
__global__ void k_pair_votes(const uint64_t* __restrict__ pairs, const uint32_t* __restrict__ head_pos, uint64_t n_heads, uint64_t n,
                             uint64_t* __restrict__ ukey, uint32_t* __restrict__ votes)
{
	uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n_heads) return;
	uint64_t end = (j + 1 < n_heads) ? head_pos[j + 1] : n;
	ukey[j] = pairs[head_pos[j]]; votes[j] = (uint32_t)(end - head_pos[j]);
}
__device__ inline uint64_t lower_bound_u64(const uint64_t* __restrict__ a, uint64_t n, uint64_t v)
{
	uint64_t lo = 0, hi = n;
	while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
	return lo;
}
// one wave per read: top-c by (votes desc, ref asc) (reads_sim_graph.cpp:411-418)
__global__ __launch_bounds__(256) void k_top_candidates(const uint64_t* __restrict__ ukey, const uint32_t* __restrict__ votes, uint64_t n_u,
                                                        uint32_t r0, uint32_t r1, uint32_t ref_bits, uint32_t c,
                                                        uint32_t* __restrict__ o_refs, uint32_t* __restrict__ o_votes, uint32_t* __restrict__ o_n)
{
	uint32_t r = r0 + blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= r1) return;
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t ref_mask = (1ULL << ref_bits) - 1;
	uint64_t a = lower_bound_u64(ukey, n_u, (uint64_t)(r - r0) << ref_bits);
	uint64_t b = lower_bound_u64(ukey, n_u, (uint64_t)(r - r0 + 1) << ref_bits);
	uint64_t prev = ~0ULL;                                 // packed (votes << 32 | ~ref) of the last pick; next must be smaller
	uint32_t got = 0;
	for (; got < c; ++got)
	{
		uint64_t best = 0; bool any = false;
		for (uint64_t i = a + lane; i < b; i += 64)
		{
			uint64_t p = ((uint64_t)votes[i] << 32) | (uint32_t)~(uint32_t)(ukey[i] & ref_mask);
			if (p < prev && (!any || p > best)) { best = p; any = true; }
		}
		// wave max over lanes that have a candidate (votes >= 1, so a valid p is never 0)
		uint64_t v = any ? best : 0;
#pragma unroll
		for (int d = 32; d > 0; d >>= 1) { uint64_t t = __shfl_xor(v, d, 64); v = t > v ? t : v; }
		if (v == 0) break;
		if (lane == 0) { o_refs[(uint64_t)r * c + got] = ~(uint32_t)v; o_votes[(uint64_t)r * c + got] = (uint32_t)(v >> 32); }
		prev = v;
	}
	if (lane == 0)
	{
		o_n[r] = got;
		for (uint32_t i = got; i < c; ++i) { o_refs[(uint64_t)r * c + i] = ~0u; o_votes[(uint64_t)r * c + i] = 0; }
	}
}
} // namespace

// Internal (stream.hip): d_bounds[i] (n + 1 entries) = ref_base + the number of accepted reads before read i — the reference reads a read of
// the chunk may be coded against —, *n_accepted = the accepted reads of the chunk.  (The first half of cl_index_entries_of on its own: the
// streaming compressor lists the k-mers of the ACCEPTED reads only, a tenth of a chunk in sparse mode, not of every read.)
cl_status cl_ref_bounds(cl_ctx* ctx, const uint8_t* d_accept, uint32_t n, uint32_t ref_base, uint32_t* d_bounds, uint32_t* n_accepted)
{
	if (!ctx || !d_accept || !d_bounds) return cl_fail(ctx, CL_E_INVALID, "cl_ref_bounds: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	uint64_t n_refs = 0;
	if (n) LAUNCH(ctx, k_flags_from_bytes, grid_for(n, 256), 256, d_accept, n, d_bounds);
	HIP_TRY(ctx, hipGetLastError());
	CL_TRY(dev_exclusive_scan_u32(ctx, d_bounds, n, &n_refs));
	{ const uint32_t t = (uint32_t)n_refs; HIP_TRY(ctx, hipMemcpyAsync(d_bounds + n, &t, 4, hipMemcpyHostToDevice, ctx->stream)); HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); }
	if (ref_base) LAUNCH(ctx, k_add_const, grid_for((uint64_t)n + 1, 256), 256, d_bounds, (uint64_t)n + 1, ref_base);
	HIP_TRY(ctx, hipGetLastError());
	if (n_accepted) *n_accepted = (uint32_t)n_refs;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CL_OK;
}

// (id, ref) pairs of the accepted reads of `lists`, in read order; ref = ref_base + rank among accepted
extern "C" cl_status cl_index_entries_of(cl_ctx* ctx, const cl_kmer_lists* L, const uint8_t* d_accept, uint32_t ref_base,
                                         uint32_t* d_ids, uint32_t* d_refs, uint64_t cap, uint64_t* n_out,
                                         uint32_t* d_bounds, uint32_t* n_accepted)
{
	if (!ctx || !L || !d_accept || !n_out) return cl_fail(ctx, CL_E_INVALID, "cl_index_entries_of: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const uint32_t nr = L->n_reads; const uint64_t ne = L->total;
	DevBuf<uint32_t> rank; DEV_ALLOC(ctx, rank, (uint64_t)nr + 1);
	uint64_t n_refs = 0;
	if (nr) LAUNCH(ctx, k_flags_from_bytes, grid_for(nr, 256), 256, d_accept, nr, rank.p);
	HIP_TRY(ctx, hipGetLastError());
	CL_TRY(dev_exclusive_scan_u32(ctx, rank.p, nr, &n_refs));
	{ uint32_t t = (uint32_t)n_refs; HIP_TRY(ctx, hipMemcpyAsync(rank.p + nr, &t, 4, hipMemcpyHostToDevice, ctx->stream)); }
	if (ref_base) LAUNCH(ctx, k_add_const, grid_for((uint64_t)nr + 1, 256), 256, rank.p, (uint64_t)nr + 1, ref_base);
	HIP_TRY(ctx, hipGetLastError());
	if (n_accepted) *n_accepted = (uint32_t)n_refs;
	if (d_bounds) HIP_TRY(ctx, hipMemcpyAsync(d_bounds, rank.p, ((uint64_t)nr + 1) * 4, hipMemcpyDeviceToDevice, ctx->stream));
	uint64_t n_sel = 0;
	if (ne)
	{
		DevBuf<uint32_t> ef; DEV_ALLOC(ctx, ef, ne);
		LAUNCH(ctx, k_entry_flags, grid_for(ne, 256), 256, (const uint32_t*)L->read.p, d_accept, ne, ef.p);
		HIP_TRY(ctx, hipGetLastError());
		CL_TRY(dev_exclusive_scan_u32(ctx, ef.p, ne, &n_sel));
		*n_out = n_sel;
		if (n_sel > cap || (n_sel && (!d_ids || !d_refs))) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); return cl_fail(ctx, CL_E_CAPACITY, "cl_index_entries_of: need " + std::to_string(n_sel) + " entries"); }
		LAUNCH(ctx, k_gather_ref_entries, grid_for(ne, 256), 256, (const uint32_t*)ef.p, ne, n_sel,
			(const uint32_t*)L->ids.p, (const uint32_t*)L->read.p, (const uint32_t*)rank.p, d_ids, d_refs);
		HIP_TRY(ctx, hipGetLastError());
	}
	*n_out = n_sel;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CL_OK;
}

// Index from explicit (id, ref) pairs listed in ascending ref order (e.g. the all-gathered entries of every
// rank).  d_bounds[i] (n_reads+1) = number of reference reads that precede local read i in the global order.
// d_ids / d_refs are sorted in place.
extern "C" cl_status cl_index_build_pairs(cl_ctx* ctx, const cl_kmer_set* S, uint32_t* d_ids, uint32_t* d_refs, uint64_t n_sel,
                                          const uint32_t* d_bounds, uint32_t n_reads, uint32_t n_refs_total,
                                          uint32_t n_pseudo, uint32_t max_kmer_count, cl_index** out)
{
	if (!ctx || !S || (!d_bounds && n_reads) || !out) return cl_fail(ctx, CL_E_INVALID, "cl_index_build_pairs: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	cl_timing_begin(ctx);
	cl_index* X = new cl_index(); X->ctx = ctx; X->n_reads = n_reads; X->n_pseudo = n_pseudo; X->n_keys = S->n; X->n_refs = n_refs_total;
	std::unique_ptr<cl_index> guard(X);
	DEV_ALLOC(ctx, X->ref_rank, (uint64_t)n_reads + 1);
	if (d_bounds) HIP_TRY(ctx, hipMemcpyAsync(X->ref_rank.p, d_bounds, ((uint64_t)n_reads + 1) * 4, hipMemcpyDeviceToDevice, ctx->stream));
	else HIP_TRY(ctx, hipMemsetAsync(X->ref_rank.p, 0, 4, ctx->stream));
	uint64_t n_keep = 0;
	DevBuf<uint32_t> id_counts; DEV_ALLOC(ctx, id_counts, S->n + 1);
	HIP_TRY(ctx, hipMemsetAsync(id_counts.p, 0, (S->n + 1) * 4, ctx->stream));
	if (n_sel)
	{
		uint32_t bits = 1; while (bits < 32 && (1ULL << bits) < S->n) ++bits;
		CL_TRY(dev_sort_keys32_pairs(ctx, d_ids, d_refs, n_sel, 0, bits));      // stable: refs stay ascending inside a k-mer
		DevBuf<uint32_t> hf; DEV_ALLOC(ctx, hf, n_sel);
		LAUNCH(ctx, k_head_flags32, grid_for(n_sel, 256), 256, (const uint32_t*)d_ids, n_sel, hf.p);
		uint64_t n_heads = 0;
		CL_TRY(dev_exclusive_scan_u32(ctx, hf.p, n_sel, &n_heads));
		DevBuf<uint32_t> head_pos; DEV_ALLOC(ctx, head_pos, n_heads);
		LAUNCH(ctx, k_scatter_head_pos, grid_for(n_sel, 256), 256, (const uint32_t*)hf.p, n_sel, n_heads, head_pos.p);
		DevBuf<uint32_t> keep; DEV_ALLOC(ctx, keep, n_sel);
		LAUNCH(ctx, k_cap_flags, grid_for(n_sel, 256), 256, (const uint32_t*)d_ids, (const uint32_t*)d_refs, (const uint32_t*)hf.p,
			(const uint32_t*)head_pos.p, n_sel, n_heads, n_pseudo, max_kmer_count, keep.p);
		HIP_TRY(ctx, hipGetLastError());
		CL_TRY(dev_exclusive_scan_u32(ctx, keep.p, n_sel, &n_keep));
		DEV_ALLOC(ctx, X->refs, n_keep);
		LAUNCH(ctx, k_compact_index, grid_for(n_sel, 256), 256, (const uint32_t*)keep.p, n_sel, n_keep,
			(const uint32_t*)d_ids, (const uint32_t*)d_refs, X->refs.p, id_counts.p);
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	else DEV_ALLOC(ctx, X->refs, 0);
	X->n_entries = n_keep;
	DEV_ALLOC(ctx, X->off, S->n + 1);
	uint64_t n_off = 0;
	CL_TRY(dev_exclusive_scan_u64(ctx, id_counts.p, X->off.p, S->n, &n_off));        // (with the total: the call returns with the offsets complete — other contexts read the index)
	cl_timing_collect(ctx);
	*out = guard.release();
	return CL_OK;
}

extern "C" cl_status cl_index_build(cl_ctx* ctx, const cl_kmer_set* S, const cl_kmer_lists* L, const uint8_t* d_accept,
                                    uint32_t n_pseudo, uint32_t max_kmer_count, cl_index** out)
{
	if (!ctx || !S || !L || !d_accept || !out) return cl_fail(ctx, CL_E_INVALID, "cl_index_build: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	uint64_t n_sel = 0; uint32_t n_refs = 0;
	DevBuf<uint32_t> bounds; DEV_ALLOC(ctx, bounds, (uint64_t)L->n_reads + 1);
	cl_status st = cl_index_entries_of(ctx, L, d_accept, 0, nullptr, nullptr, 0, &n_sel, bounds.p, &n_refs);
	if (st != CL_OK && st != CL_E_CAPACITY) return st;
	DevBuf<uint32_t> sid, sref; DEV_ALLOC(ctx, sid, n_sel); DEV_ALLOC(ctx, sref, n_sel);
	if (n_sel) CL_TRY(cl_index_entries_of(ctx, L, d_accept, 0, sid.p, sref.p, n_sel, &n_sel, nullptr, nullptr));
	return cl_index_build_pairs(ctx, S, sid.p, sref.p, n_sel, bounds.p, L->n_reads, n_refs, n_pseudo, max_kmer_count, out);
}
extern "C" void cl_index_free(cl_index* ix) { delete ix; }
extern "C" uint32_t cl_index_n_refs(const cl_index* ix) { return ix->n_refs; }
extern "C" uint64_t cl_index_entries(const cl_index* ix) { return ix->n_entries; }
extern "C" const uint32_t* cl_index_ref_rank(const cl_index* ix) { return ix->ref_rank.p; }

extern "C" cl_status cl_candidates(cl_ctx* ctx, const cl_index* X, const cl_kmer_lists* L, uint32_t c,
                                   uint32_t* d_refs, uint32_t* d_votes, uint32_t* d_n)
{
	if (!ctx || !X || !L) return cl_fail(ctx, CL_E_INVALID, "cl_candidates: null argument");
	if (X->n_reads != L->n_reads) return cl_fail(ctx, CL_E_INVALID, "cl_candidates: index/lists mismatch");
	return cl_candidates_at(ctx, X, L, X->ref_rank.p, c, d_refs, d_votes, d_n);
}
namespace { __global__ void k_gather_at(const uint64_t* __restrict__ src, const uint64_t* __restrict__ at, uint32_t n, uint64_t* __restrict__ out) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = src[at[i]]; } }
// The query of one CHUNK of reads against an index that covers the reference reads of the whole input: d_bounds[i] =
// number of reference reads that precede read i of `lists` in file order (read i sees exactly those, App. F1 of SURVEY.md).
extern "C" cl_status cl_candidates_at(cl_ctx* ctx, const cl_index* X, const cl_kmer_lists* L, const uint32_t* d_bounds, uint32_t c,
                                      uint32_t* d_refs, uint32_t* d_votes, uint32_t* d_n)
{
	if (!ctx || !X || !L || !d_bounds || !d_refs || !d_votes || !d_n) return cl_fail(ctx, CL_E_INVALID, "cl_candidates: null argument");
	if (c == 0) return cl_fail(ctx, CL_E_INVALID, "cl_candidates: max_candidates == 0");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	cl_timing_begin(ctx);
	const uint32_t nr = L->n_reads; const uint64_t ne = L->total;
	uint32_t ref_bits = 1; while (ref_bits < 32 && (1ULL << ref_bits) < (uint64_t)X->n_refs + 1) ++ref_bits;
	// vote pairs per entry
	DevBuf<uint32_t> cnt; DEV_ALLOC(ctx, cnt, ne);
	DevBuf<uint64_t> poff; DEV_ALLOC(ctx, poff, ne + 1);
	uint64_t n_pairs = 0;
	if (ne)
	{
		LAUNCH(ctx, k_pair_counts, grid_for(ne, 256), 256, (const uint32_t*)L->ids.p, (const uint32_t*)L->read.p, ne,
			d_bounds, (const uint64_t*)X->off.p, (const uint32_t*)X->refs.p, cnt.p);
	}
	HIP_TRY(ctx, hipGetLastError());
	CL_TRY(dev_exclusive_scan_u64(ctx, cnt.p, poff.p, ne, &n_pairs));
	// host copies of the (small) per-read entry offsets and of the pair offsets AT read boundaries to cut batches (gathered on the
	// device: the whole pair-offset array is 8 bytes per list entry — a quarter to half a gigabyte per 1-Gbase chunk, 30-60 ms of
	// copy to pageable memory with the lane's stream idle)
	std::vector<uint64_t> h_off((size_t)nr + 1), h_poff_at((size_t)nr + 1);
	{
		DevBuf<uint64_t> at; DEV_ALLOC(ctx, at, (uint64_t)nr + 1);
		LAUNCH(ctx, k_gather_at, grid_for((uint64_t)nr + 1, 256), 256, (const uint64_t*)poff.p, (const uint64_t*)L->off.p, nr + 1, at.p);
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipMemcpyAsync(h_off.data(), L->off.p, ((size_t)nr + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(h_poff_at.data(), at.p, ((size_t)nr + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	const uint64_t MAX_PAIRS = 1ull << 27;                  // pairs per batch (1 GiB of keys + sort scratch)
	const uint32_t MAX_READS = 1u << (40 - ref_bits > 31 ? 31 : 40 - ref_bits);   // keep keys within 40 bits => 5 radix passes
	uint32_t r0 = 0;
	DevBuf<uint64_t> pairs; DevBuf<uint32_t> hf, head_pos, votes; DevBuf<uint64_t> ukey;
	while (r0 < nr)
	{
		uint32_t r1 = r0 + 1;
		while (r1 < nr && r1 - r0 < MAX_READS && h_poff_at[r1 + 1] - h_poff_at[r0] <= MAX_PAIRS) ++r1;
		const uint64_t e0 = h_off[r0], e1 = h_off[r1], p0 = h_poff_at[r0], np = h_poff_at[r1] - p0;
		uint64_t n_u = 0;
		if (np)
		{
			if (np >= (1ULL << 32)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_candidates: a single read produces >= 2^32 vote pairs");
			if (pairs.n < np) { DEV_ALLOC(ctx, pairs, np); DEV_ALLOC(ctx, hf, np); }
			{ LAUNCH(ctx, k_pair_fill, grid_for(e1 - e0, 256), 256, (const uint32_t*)L->ids.p, (const uint32_t*)L->read.p, e0, e1,
				(const uint32_t*)cnt.p, (const uint64_t*)poff.p, p0, (const uint64_t*)X->off.p, (const uint32_t*)X->refs.p, r0, ref_bits, pairs.p); }
			HIP_TRY(ctx, hipGetLastError());
			uint32_t rb = 1; while ((1ULL << rb) < (uint64_t)(r1 - r0)) ++rb;
			CL_TRY(dev_sort_pairs(ctx, pairs.p, nullptr, np, 0, ref_bits + rb));
			LAUNCH(ctx, k_head_flags64, grid_for(np, 256), 256, (const uint64_t*)pairs.p, np, hf.p);
			CL_TRY(dev_exclusive_scan_u32(ctx, hf.p, np, &n_u));
			if (head_pos.n < n_u) { DEV_ALLOC(ctx, head_pos, n_u); DEV_ALLOC(ctx, votes, n_u); DEV_ALLOC(ctx, ukey, n_u); }
			LAUNCH(ctx, k_scatter_head_pos, grid_for(np, 256), 256, (const uint32_t*)hf.p, np, n_u, head_pos.p);
			LAUNCH(ctx, k_pair_votes, grid_for(n_u, 256), 256, (const uint64_t*)pairs.p, (const uint32_t*)head_pos.p, n_u, np, ukey.p, votes.p);
			HIP_TRY(ctx, hipGetLastError());
		}
		{ LAUNCH(ctx, k_top_candidates, grid_for(r1 - r0, 4), 256, (const uint64_t*)ukey.p, (const uint32_t*)votes.p, n_u,
			r0, r1, ref_bits, c, d_refs, d_votes, d_n); }
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		r0 = r1;
	}
	cl_timing_collect(ctx);
	return CL_OK;
}

// ---- HiFi: shared k-mers per chosen candidate, in read order (reads_sim_graph.cpp:473-486,518-522) ----
namespace {
__device__ inline bool list_has(const uint32_t* __restrict__ a, uint32_t n, uint32_t v)
{
	uint32_t p = lower_bound_u32(a, n, v);
	return p < n && a[p] == v;
}
// MODE 0: count per (read, slot); MODE 1: fill.  One wave per read.
template<int MODE>
__global__ __launch_bounds__(256) void k_common(const uint64_t* __restrict__ loff, const uint32_t* __restrict__ ids, const uint64_t* __restrict__ kmers,
                                                const uint64_t* __restrict__ ioff, const uint32_t* __restrict__ irefs,
                                                const uint32_t* __restrict__ refs, const uint32_t* __restrict__ nref, uint32_t n_reads, uint32_t c,
                                                uint32_t* __restrict__ counts, const uint64_t* __restrict__ coff, uint64_t* __restrict__ common, uint64_t cap)
{
	uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n_reads) return;
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t a = loff[r], b = loff[r + 1];
	const uint32_t n = nref[r];
	for (uint32_t s = 0; s < c; ++s)
	{
		uint32_t total = 0;
		if (s < n)
		{
			const uint32_t ref = refs[(uint64_t)r * c + s];
			uint64_t o = MODE ? coff[(uint64_t)r * c + s] : 0;
			for (uint64_t e0 = a; e0 < b; e0 += 64)
			{
				uint64_t e = e0 + lane; bool hit = false;
				if (e < b) { uint32_t id = ids[e]; uint64_t la = ioff[id]; hit = list_has(irefs + la, (uint32_t)(ioff[id + 1] - la), ref); }
				uint64_t m = __ballot(hit);
				if (MODE && hit) { uint64_t p = o + total + __popcll(m & ((1ULL << lane) - 1)); if (p < cap) common[p] = kmers[e]; }
				total += (uint32_t)__popcll(m);
			}
		}
		if (!MODE && lane == 0) counts[(uint64_t)r * c + s] = total;
	}
}
} // namespace

extern "C" cl_status cl_candidates_common(cl_ctx* ctx, const cl_index* X, const cl_kmer_lists* L, uint32_t c,
                                          const uint32_t* d_refs, const uint32_t* d_n,
                                          uint64_t* d_common_off, uint64_t* d_common, uint64_t cap, uint64_t* n_common)
{
	if (!ctx || !X || !L || !d_refs || !d_n || !d_common_off || !n_common) return cl_fail(ctx, CL_E_INVALID, "cl_candidates_common: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const uint32_t nr = L->n_reads; const uint64_t nslots = (uint64_t)nr * c;
	DevBuf<uint32_t> counts; DEV_ALLOC(ctx, counts, nslots);
	if (nr) LAUNCH(ctx, (k_common<0>), grid_for(nr, 4), 256, (const uint64_t*)L->off.p, (const uint32_t*)L->ids.p, (const uint64_t*)L->kmers.p,
		(const uint64_t*)X->off.p, (const uint32_t*)X->refs.p, d_refs, d_n, nr, c, counts.p, (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0);
	HIP_TRY(ctx, hipGetLastError());
	uint64_t total = 0;
	CL_TRY(dev_exclusive_scan_u64(ctx, counts.p, d_common_off, nslots, &total));
	*n_common = total;
	if (total > cap || (total && !d_common)) return cl_fail(ctx, CL_E_CAPACITY, "cl_candidates_common: need " + std::to_string(total) + " k-mers");
	if (nr && total) LAUNCH(ctx, (k_common<1>), grid_for(nr, 4), 256, (const uint64_t*)L->off.p, (const uint32_t*)L->ids.p, (const uint64_t*)L->kmers.p,
		(const uint64_t*)X->off.p, (const uint32_t*)X->refs.p, d_refs, d_n, nr, c, (uint32_t*)nullptr, (const uint64_t*)d_common_off, d_common, cap);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CL_OK;
}
