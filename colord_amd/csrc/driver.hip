// driver.hip — the compress data path of one shard as ONE native call: what runCompression wires together with threads
// and queues (src/colord/compression.cpp:432-689: CKmerCounter -> CKmerFilter -> CRefReadsAccepter ->
// CReadsSimilarityGraph -> CEncoder -> CEntrComprReads / CEntrComprQuals), here a straight sequence of the stage entry
// points of this library on one GPU stream.  Host code only; every byte of the result comes from the HIP stages.
#include "common.hpp"
#include "objects.hpp"
#include <vector>
#include <memory>
#include <thread>

namespace {
__global__ void k_accept_flags(const uint8_t* __restrict__ acc, const uint8_t* __restrict__ has_n, uint32_t n, uint8_t* __restrict__ out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = acc[i] && !has_n[i] ? 1 : 0;
}
template<class T, void (*F)(T*)> struct Handle {      // frees a stage object on scope exit
	T* p = nullptr; ~Handle() { if (p) F(p); } T** out() { return &p; } operator T*() const { return p; }
};
} // namespace

extern "C" cl_status cl_compress_shard(cl_ctx* ctx, const cl_compress_params* P, const cl_reads* reads, const uint8_t* d_quals, const uint64_t* d_base_off,
                                       const uint32_t* h_part_bounds, uint32_t n_parts, const uint32_t* h_pack_bounds, uint32_t n_packs,
                                       cl_dna_coder* dna, cl_qual_coder* qual,
                                       uint8_t* d_dna_out, uint64_t dna_cap, uint64_t* h_dna_part_sizes,
                                       uint8_t* d_qual_out, uint64_t qual_cap, uint64_t* h_qual_part_sizes, cl_compress_info* info)
{
	if (!ctx || !P || !reads || !h_part_bounds || !h_pack_bounds || !dna || !info) return cl_fail(ctx, CL_E_INVALID, "cl_compress_shard: null argument");
	if (qual && (!d_quals || !d_base_off || !h_qual_part_sizes)) return cl_fail(ctx, CL_E_INVALID, "cl_compress_shard: quality coder without qualities");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	memset(info, 0, sizeof(*info));
	const uint32_t n = reads->n_reads;
	info->n_reads = n; info->n_bases = reads->total_bases;
	if (!n) return CL_OK;
	hipStream_t st = ctx->stream;
	// The quality stream of level 1 does not depend on the edit scripts: when its coder lives on a second context of the
	// same GPU (own stream, own pool) it is coded concurrently with the whole DNA path — both are latency-bound chains
	// that leave most of the machine idle on their own.
	struct Joiner { std::thread t; ~Joiner() { if (t.joinable()) t.join(); } } qjob;
	cl_status qstatus = CL_OK;
	cl_ctx* qctx = qual ? cl_qual_coder_ctx(qual) : nullptr;
	const bool overlap = qual && P->level <= 1 && qctx && qctx != ctx;
	if (overlap)
		qjob.t = std::thread([&]() { qstatus = cl_qual_encode(qctx, qual, reads, d_quals, d_base_off, nullptr, h_part_bounds, n_parts, d_qual_out, qual_cap, h_qual_part_sizes, &info->qual_bytes); });
	// a1 + a2 + a3: k-mer scan, exact count / threshold, membership set (compression.cpp:432-464)
	Handle<cl_kmer_set, cl_kmer_set_free> kset; cl_kmer_stats ks{};
	{
		uint64_t cap = P->f > 1 ? (uint64_t)(reads->total_bases / P->f * 1.3) + 4096 : reads->total_bases + 64, got = 0;
		DevBuf<uint64_t> km;
		for (;;)
		{
			DEV_ALLOC(ctx, km, cap);
			const cl_status s = cl_kmer_scan(ctx, reads, P->k, P->f, km.p, cap, &got);
			if (s == CL_E_CAPACITY) { cap = got; continue; }
			CL_TRY(s);
			break;
		}
		CL_TRY(cl_kmer_count_filter(ctx, km.p, got, P->k, P->ci, P->cs, kset.out(), &ks));
	}
	info->tot_kmers = ks.tot_kmers; info->n_kept_kmers = ks.n_unique_counted;
	// a4: accepted k-mers per read
	Handle<cl_kmer_lists, cl_kmer_lists_free> lists;
	CL_TRY(cl_accepted_kmers(ctx, kset, reads, P->k, P->f, lists.out()));
	// a6: acceptor with the host scalars of compression.cpp:443,501-503
	const uint64_t mean_read_len = (uint64_t)((double)(ks.tot_kmers * P->f) / n + P->k - 1);
	DevBuf<uint8_t> accept; DEV_ALLOC(ctx, accept, n);
	{
		std::vector<uint8_t> h_acc(n, 1);
		if (P->sparse)
		{
			uint32_t range = (uint32_t)((P->sparse_g * (double)ks.n_unique_counted * P->f) / (double)(mean_read_len ? mean_read_len : 1));
			if (range < 1) range = 1;
			CL_TRY(cl_ref_accept(n, 0, range, P->sparse_exponent, h_acc.data()));
			info->sparse_range = range;
		}
		DevBuf<uint8_t> d_acc; DEV_ALLOC(ctx, d_acc, n);
		HIP_TRY(ctx, hipMemcpyAsync(d_acc.p, h_acc.data(), n, hipMemcpyHostToDevice, st));
		LAUNCH(ctx, k_accept_flags, grid_for(n, 256), 256, (const uint8_t*)d_acc.p, (const uint8_t*)reads->has_n.p, n, accept.p);
		HIP_TRY(ctx, hipStreamSynchronize(st));
	}
	// a5: index + candidates (+ shared k-mers for HiFi); a7: reference reads
	Handle<cl_index, cl_index_free> index;
	CL_TRY(cl_index_build(ctx, kset, lists, accept.p, 0, P->cs, index.out()));
	const uint32_t c = std::min<uint32_t>(P->c, 16);                          // (16 candidate views per frame, 8 recursion levels: see stage_a in stream.hip)
	DevBuf<uint32_t> crefs, votes, cnt; DEV_ALLOC(ctx, crefs, (uint64_t)n * c); DEV_ALLOC(ctx, votes, (uint64_t)n * c); DEV_ALLOC(ctx, cnt, n);
	CL_TRY(cl_candidates(ctx, index, lists, c, crefs.p, votes.p, cnt.p));
	DevBuf<uint64_t> common_off, common;
	const bool hifi = P->source == 2;
	if (hifi)
	{
		DEV_ALLOC(ctx, common_off, (uint64_t)n * c + 1);
		uint64_t need = 0;
		cl_status s = cl_candidates_common(ctx, index, lists, c, crefs.p, cnt.p, common_off.p, nullptr, 0, &need);
		if (s != CL_OK && s != CL_E_CAPACITY) return s;
		DEV_ALLOC(ctx, common, need + 1);
		CL_TRY(cl_candidates_common(ctx, index, lists, c, crefs.p, cnt.p, common_off.p, common.p, need, &need));
	}
	Handle<cl_reads, cl_reads_free> refs;
	CL_TRY(cl_reads_select(ctx, reads, accept.p, refs.out()));
	info->n_refs = refs.p->n_reads;
	// a8 / a9: anchors; a10-a12: tuple streams
	Handle<cl_anchors, cl_anchors_free> anc;
	CL_TRY(cl_anchor_candidates_hifi(ctx, reads, refs, crefs.p, cnt.p, c, P->anchor_len, P->frac_always, P->frac_min, P->max_matches_mult, P->min_anchors,
		P->k, P->f, hifi ? common_off.p : nullptr, hifi ? common.p : nullptr, anc.out()));
	info->n_anchors = cl_anchors_total(anc);
	DevBuf<uint8_t> es; DevBuf<uint64_t> es_off; DevBuf<uint32_t> es_nt;
	const uint64_t es_cap = reads->total_bases + 16ull * n + 4096;
	DEV_ALLOC(ctx, es, es_cap); DEV_ALLOC(ctx, es_off, (uint64_t)n + 1); DEV_ALLOC(ctx, es_nt, n);
	uint64_t es_bytes = 0;
	CL_TRY(cl_encode_reads(ctx, reads, refs, anc, c, P->anchor_len, P->min_part_alt, std::min<uint32_t>(P->max_rec, 8), P->cost_mult, h_pack_bounds, n_packs, es.p, es_cap, es_off.p, es_nt.p, &es_bytes));
	info->tuple_bytes = es_bytes;
	// a14 + a16: DNA stream; a13 + a15: quality stream (levels 2 and 3 take the per-base classes of the scripts)
	CL_TRY(cl_dna_encode(ctx, dna, refs, es.p, es_off.p, es_nt.p, n, h_part_bounds, n_parts, d_dna_out, dna_cap, h_dna_part_sizes, &info->dna_bytes));
	if (overlap)
	{
		qjob.t.join();
		if (qstatus != CL_OK) return cl_fail(ctx, qstatus, std::string("quality stream: ") + cl_last_error(qctx));
	}
	else if (qual)
	{
		DevBuf<uint8_t> flags;
		if (P->level > 1)
		{
			DEV_ALLOC(ctx, flags, reads->total_bases + 1);
			CL_TRY(cl_es_flags(ctx, reads, es.p, es_off.p, d_base_off, flags.p));
		}
		CL_TRY(cl_qual_encode(ctx, qual, reads, d_quals, d_base_off, P->level > 1 ? flags.p : nullptr, h_part_bounds, n_parts, d_qual_out, qual_cap, h_qual_part_sizes, &info->qual_bytes));
	}
	return CL_OK;
}
