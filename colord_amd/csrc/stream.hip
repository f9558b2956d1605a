// stream.hip — runCompression for inputs that do not fit one call, and for reads sharded over GPUs.
//
// The reference streams a file of any size through its two passes: pass 1 counts the k-mers of the WHOLE input
// (compression.cpp:432, CKmerCounter), pass 2 pushes reader packs of 4 Mi symbols through graph -> encoder -> coders
// (compression.cpp:547-561, in_reads.cpp:62-77), the similarity graph growing as reference reads pass by
// (reads_sim_graph.cpp:324-427).  cl_compressor is that state between calls; the caller hands it the input CHUNK by
// chunk (a chunk = whole reader packs, e.g. 1 Gbase), three times:
//   count_add*  -> count_finish      pass 1: surviving k-mers accumulate, then exact counts / the filtered set
//   refs_add*   -> refs_finish       which reads become reference reads (acceptor), their store (a7) and the k-mer ->
//                                    reference-reads index over the whole input; a read of any chunk sees exactly the
//                                    index entries of EARLIER reference reads (lists are prefixes in id order, SURVEY
//                                    App. F1), so the chunked result is byte-identical to one call over everything
//   encode*                          pass 2: candidates, anchors, edit scripts, `dna` / `qual` parts of a chunk; the
//                                    coders' adaptive models persist from chunk to chunk as in one CEntrCompr* thread
// With a cl_exchange (one process per GPU, reads sharded in file order) the two finish steps run the exchanges of SURVEY
// §8e: k-mers to the owner of their key range, kept keys all-gathered (replicated set); reference reads and index
// entries all-gathered (replicated store + index).  The collectives themselves are the caller's (torch.distributed over
// RCCL in colord_amd/parallel.py); this file only says what is exchanged.  Host code; all data work is in the stages.
#include "common.hpp"
#include "objects.hpp"
#include <set>
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <chrono>

namespace {
__global__ void k_accept_flags2(const uint8_t* __restrict__ acc, const uint8_t* __restrict__ has_n, uint32_t n, uint8_t* __restrict__ out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = acc[i] && !has_n[i] ? 1 : 0;
}
__global__ void k_add_u32(uint32_t* v, uint64_t n, uint32_t c) { const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] += c; }
// k-mers grouped by destination rank (the owner of the 4096-bin range a key falls into); cursor[r] = start of r's group
__global__ __launch_bounds__(256) void k_owner_scatter(const uint64_t* __restrict__ keys, uint64_t n, uint32_t shift, const uint32_t* __restrict__ owner_of_bin,
                                                       uint32_t world, unsigned long long* __restrict__ cursor, uint64_t* __restrict__ out)
{
	__shared__ uint32_t cnt[64];
	__shared__ unsigned long long base[64];
	if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
	__syncthreads();
	const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	uint64_t key = 0; uint32_t o = 0, my = 0;
	if (i < n) { key = keys[i]; o = owner_of_bin[(uint32_t)(key >> shift) & 4095u]; my = atomicAdd(&cnt[o], 1u); }
	__syncthreads();
	if (threadIdx.x < world && cnt[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
	__syncthreads();
	if (i < n) out[base[o] + my] = key;
}
template<class T> struct Grow {            // device array that grows geometrically (k-mers of pass 1, index entries of pass 2a)
	DevBuf<T> buf; uint64_t n = 0;
	cl_status reserve(cl_ctx* ctx, uint64_t need)
	{
		if (need <= buf.n) return CL_OK;
		uint64_t cap = std::max<uint64_t>(need, buf.n + buf.n / 2 + 1024);
		DevBuf<T> nb; DEV_ALLOC(ctx, nb, cap);
		if (n) { HIP_TRY(ctx, hipMemcpyAsync(nb.p, buf.p, n * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream)); HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); }
		buf = std::move(nb);
		return CL_OK;
	}
};
} // namespace

struct cl_compressor {
	cl_ctx* ctx = nullptr; cl_ctx* qctx = nullptr;
	cl_compress_params P{}; bool has_qual = false; cl_qual_params Q{};
	cl_exchange X{}; uint32_t rank = 0, world = 1;
	int phase = 0;                                  // 0 counting, 1 counted, 2 references listed, 3 encoding
	// pass 1
	Grow<uint64_t> kmers; uint64_t expected_bases = 0;
	std::vector<uint32_t> chunk_reads; uint64_t n_reads_local = 0, n_bases_local = 0;
	cl_kmer_set* kset = nullptr; cl_kmer_stats gstats{};
	uint64_t n_reads_total = 0, first_read = 0, mean_read_len = 0; uint32_t sparse_range = 0;
	uint64_t genome_seqs = 0, genome_len = 0; uint32_t n_pseudo = 0;      // reference-genome mode (compression.cpp:405-447)
	std::vector<uint8_t> h_accept;                  // acceptor decisions of this rank's reads
	// pass 2a
	size_t refs_chunk = 0; uint64_t refs_reads_seen = 0; uint32_t n_refs_local = 0;
	std::vector<cl_reads*> ref_pieces;
	Grow<uint32_t> pair_ids, pair_refs;
	std::vector<DevBuf<uint32_t>> bounds;           // per chunk: n_reads + 1, reference reads before each read
	cl_reads* refs = nullptr; cl_index* index = nullptr; uint32_t ref_base = 0, n_refs_total = 0;
	cl_dna_coder* dna = nullptr; cl_qual_coder* qual = nullptr;
	// pass 2b
	size_t enc_chunk = 0;
	// look-ahead (cl_compressor_prepare): stage A of announced chunks — candidates, anchors, alignments, tuple streams, none of
	// which depends on an earlier chunk's coders — runs on the compressor's own contexts ("encode lanes", one worker thread
	// each) while the caller's thread codes the chunks before them.
	struct Prepared {
		const cl_reads* reads = nullptr; std::vector<uint32_t> packs;
		DevBuf<uint8_t> es; DevBuf<uint64_t> es_off; DevBuf<uint32_t> es_nt;
		uint64_t es_bytes = 0, n_anchors = 0;
		cl_status status = CL_OK; std::string err;
		std::map<std::string, KernelTime> times;      // kernel times of the lane for this chunk (merged into the caller's context)
		bool done = false;
		// the model-independent half of the DNA coder for this chunk (tuple walks, and with part bounds the sort by context), made by
		// the compressor's preparation thread beside the coding of the chunk before (cl_dna_prepare_batch)
		std::vector<uint32_t> parts; DnaWalked* walked = nullptr; bool dna_done = false; std::map<std::string, KernelTime> dna_times;
		// ... and of the quality coder (symbols, sort by context), which needs the input only (level 1: no flags from the edit scripts)
		const uint8_t* d_quals = nullptr; const uint64_t* d_base_off = nullptr; QualPrepared* qprep = nullptr; bool q_done = false; std::map<std::string, KernelTime> q_times;
		~Prepared() { if (walked) cl_dna_walked_free(walked); if (qprep) cl_qual_prepared_free(qprep); }
	};
	std::mutex lane_mu; std::condition_variable lane_cv;
	std::deque<size_t> lane_queue;                   // announced chunk indices not yet started, ascending
	std::map<size_t, std::unique_ptr<Prepared>> prepared;
	std::vector<std::thread> lane_threads; std::vector<cl_ctx*> lane_ctx;
	size_t n_announced = 0; bool lane_stop = false;
	// DNA preparation thread: walks (and sorts) the chunks in order, one or two ahead of the coders, on a context of its own
	// DNA preparation: workers (contexts of their own) that CLAIM the chunks in order; the two scalars that chain from chunk to chunk — the
	// types of the last four reads, the read count — are advanced at the claim (cl_dna_batch_types: a few bytes of the tuple streams), so
	// the chunks themselves are prepared side by side
	std::vector<std::thread> prep_threads; std::vector<cl_ctx*> prep_ctxs; std::mutex prep_claim_mu;
	size_t prep_next = 0; bool prep_on = false, prep_broken = false; uint32_t prep_types = 0, prep_read_id = 0;
	std::thread qprep_thread; cl_ctx* qprep_ctx = nullptr; size_t qprep_next = 0; bool qprep_on = false;
	uint32_t n_dna_ahead = 0, n_qual_ahead = 0, n_dna_prep = 0, n_qual_prep = 0;      // (statistics: COLORD_HIP_STREAM_DEBUG)
	double w_lane_idle = 0, w_lane_work = 0, w_enc_lane = 0, w_enc_prep = 0, w_enc_qprep = 0, w_prep_idle = 0, w_prep_work = 0;   // seconds: who waited for whom
	// chunks whose model half is done ahead of their encode call (first not yet done), and how far ahead that may go: with the
	// reference's parts the interval coders of a chunk take 1.3 s, those of `depth` + 1 chunks run side by side
	size_t dna_evolved_upto = 0, qual_evolved_upto = 0; uint32_t evolve_depth = 0;
	void stop_lanes()
	{
		{ std::lock_guard<std::mutex> l(lane_mu); lane_stop = true; }
		lane_cv.notify_all();
		for (auto& t : lane_threads) if (t.joinable()) t.join();
		for (auto& t : prep_threads) if (t.joinable()) t.join();
		if (qprep_thread.joinable()) qprep_thread.join();
		lane_threads.clear();
		prepared.clear();                                // (buffers go back to the lanes' pools)
		lane_ctx.clear();                                // the contexts stay with ctx for the next compressor (their pools are warm)
	}
	~cl_compressor()
	{
		stop_lanes();
		if (getenv("COLORD_HIP_STREAM_DEBUG"))
		{
			fprintf(stderr, "[stream] %zu chunks: dna prepared ahead %u, evolved ahead %u; qual prepared ahead %u, evolved ahead %u; %zu lanes\n", enc_chunk, n_dna_prep, n_dna_ahead, n_qual_prep, n_qual_ahead, lane_ctx.size());
			fprintf(stderr, "[stream] lanes: %.2f s in stage A, %.2f s waiting for the window; dna preparation: %.2f s working, %.2f s waiting; encode calls waited %.2f s for a lane, %.2f s for the dna preparation, %.2f s for the quality preparation\n",
				w_lane_work, w_lane_idle, w_prep_work, w_prep_idle, w_enc_lane, w_enc_prep, w_enc_qprep);
		}
		for (auto* r : ref_pieces) cl_reads_free(r);
		if (refs) cl_reads_free(refs);
		if (index) cl_index_free(index);
		if (kset) cl_kmer_set_free(kset);
		if (dna) cl_dna_coder_free(dna);
		if (qual) cl_qual_coder_free(qual);
	}
};

extern "C" cl_status cl_compressor_create(cl_ctx* ctx, cl_ctx* qual_ctx, const cl_compress_params* params, const cl_qual_params* qparams,
                                          const cl_exchange* exchange, uint64_t expected_bases, cl_compressor** out)
{
	if (!ctx || !params || !out) return cl_fail(ctx, CL_E_INVALID, "cl_compressor_create: null argument");
	if (exchange && (exchange->world < 1 || exchange->world > 64 || exchange->rank >= exchange->world ||
	                 (exchange->world > 1 && (!exchange->all_gather_host || !exchange->all_to_all_v || !exchange->all_gather_v))))
		return cl_fail(ctx, CL_E_INVALID, "cl_compressor_create: exchange needs 1 <= world <= 64, rank < world and its three collectives");
	cl_compressor* c = new cl_compressor();
	c->ctx = ctx; c->qctx = qual_ctx ? qual_ctx : ctx; c->P = *params; c->expected_bases = expected_bases;
	if (qparams) { c->has_qual = true; c->Q = *qparams; }
	if (exchange && exchange->world > 1) { c->X = *exchange; c->rank = exchange->rank; c->world = exchange->world; }
	if (cl_cu_mask_cfg().any || getenv("COLORD_HIP_ROLE_PRIO"))
	{	// CU partitioning (COLORD_HIP_CU_MASK) / priorities by role: the caller's two contexts take their roles' CUs (their streams are made anew, idle as they are)
		cl_ctx_set_priority(ctx, 0, CL_ROLE_MAIN);
		if (qual_ctx && qual_ctx != ctx) cl_ctx_set_priority(qual_ctx, 0, CL_ROLE_QUAL);
	}
	*out = c;
	return CL_OK;
}
extern "C" void cl_compressor_free(cl_compressor* c) { delete c; }

// ---- pass 1 ---------------------------------------------------------------------------------------------------------
extern "C" cl_status cl_compressor_count_add(cl_compressor* c, const cl_reads* chunk)
{
	if (!c || !chunk) return CL_E_INVALID;
	cl_ctx* ctx = c->ctx;
	if (c->phase != 0) return cl_fail(ctx, CL_E_INVALID, "cl_compressor_count_add: pass 1 is already finished");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const uint32_t f = c->P.f;
	uint64_t want = f > 1 ? (uint64_t)(chunk->total_bases / f * 1.15) + 4096 : chunk->total_bases + 64;
	if (!c->kmers.buf.n && c->expected_bases > chunk->total_bases)          // one allocation for the whole input when its size is known
		CL_TRY(c->kmers.reserve(ctx, f > 1 ? (uint64_t)(c->expected_bases / f * 1.03) + 4096 : c->expected_bases + 64));
	if (c->kmers.buf.n - c->kmers.n >= chunk->total_bases / f) want = c->kmers.buf.n - c->kmers.n;     // what is left probably holds the chunk: no regrowth
	for (;;)
	{
		CL_TRY(c->kmers.reserve(ctx, c->kmers.n + want));
		uint64_t got = 0;
		const cl_status s = cl_kmer_scan(ctx, chunk, c->P.k, f, c->kmers.buf.p + c->kmers.n, c->kmers.buf.n - c->kmers.n, &got);
		if (s == CL_E_CAPACITY) { want = got + got / 64; continue; }
		CL_TRY(s);
		c->kmers.n += got;
		break;
	}
	c->chunk_reads.push_back(chunk->n_reads);
	c->n_reads_local += chunk->n_reads; c->n_bases_local += chunk->total_bases;
	return CL_OK;
}

// Reference-genome mode, pass 1 (compression.cpp:405-429): the genome's sequences are a second input of the k-mer counter.
extern "C" cl_status cl_compressor_genome_add(cl_compressor* c, const cl_reads* seqs)
{
	if (!c || !seqs) return CL_E_INVALID;
	cl_ctx* ctx = c->ctx;
	if (c->phase != 0) return cl_fail(ctx, CL_E_INVALID, "cl_compressor_genome_add: pass 1 is already finished");
	// Sharded reads: every rank is given the same sequences; rank 0 alone scans them (the k-mers travel to their owners with
	// the rest in count_finish), the others only note the numbers the statistics are corrected by.
	if (c->world > 1 && c->rank != 0) { c->genome_seqs += seqs->n_reads; c->genome_len += seqs->total_bases; return CL_OK; }
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const uint32_t f = c->P.f;
	uint64_t want = f > 1 ? (uint64_t)(seqs->total_bases / f * 1.15) + 4096 : seqs->total_bases + 64;
	for (;;)
	{
		CL_TRY(c->kmers.reserve(ctx, c->kmers.n + want));
		uint64_t got = 0;
		const cl_status s = cl_kmer_scan(ctx, seqs, c->P.k, f, c->kmers.buf.p + c->kmers.n, c->kmers.buf.n - c->kmers.n, &got);
		if (s == CL_E_CAPACITY) { want = got + got / 64; continue; }
		CL_TRY(s);
		c->kmers.n += got;
		break;
	}
	c->genome_seqs += seqs->n_reads; c->genome_len += seqs->total_bases;
	return CL_OK;
}

// k-mers to the rank that owns their key range; returns the received k-mers in c->kmers
static cl_status exchange_kmers(cl_compressor* c)
{
	cl_ctx* ctx = c->ctx; const uint32_t W = c->world, k = c->P.k;
	std::vector<uint64_t> bins;
	CL_TRY(cl_key_histogram(ctx, c->kmers.buf.p, c->kmers.n, k, bins));
	// global histogram -> W contiguous bin ranges of about equal weight (the same cut on every rank)
	std::vector<uint64_t> all((size_t)W * 4096);
	CL_TRY(c->X.all_gather_host(c->X.user, bins.data(), 4096, all.data()));
	std::vector<uint64_t> g(4096, 0); uint64_t total = 0;
	for (uint32_t r = 0; r < W; ++r) for (uint32_t b = 0; b < 4096; ++b) g[b] += all[(size_t)r * 4096 + b];
	for (uint64_t v : g) total += v;
	std::vector<uint32_t> owner(4096); uint64_t acc = 0; uint32_t r = 0;
	for (uint32_t b = 0; b < 4096; ++b)
	{
		while (r + 1 < W && acc >= (total * (r + 1) + W - 1) / W) ++r;
		owner[b] = r; acc += g[b];
	}
	DevBuf<uint32_t> d_owner; DEV_ALLOC(ctx, d_owner, 4096);
	HIP_TRY(ctx, hipMemcpyAsync(d_owner.p, owner.data(), 4096 * 4, hipMemcpyHostToDevice, ctx->stream));
	std::vector<uint64_t> send(W, 0);
	for (uint32_t b = 0; b < 4096; ++b) send[owner[b]] += bins[b];
	std::vector<unsigned long long> cur(W, 0);
	for (uint32_t i = 1; i < W; ++i) cur[i] = cur[i - 1] + send[i - 1];
	DevBuf<unsigned long long> d_cur; DEV_ALLOC(ctx, d_cur, 64);
	HIP_TRY(ctx, hipMemcpyAsync(d_cur.p, cur.data(), W * 8, hipMemcpyHostToDevice, ctx->stream));
	DevBuf<uint64_t> sorted; DEV_ALLOC(ctx, sorted, c->kmers.n);
	if (c->kmers.n)
		LAUNCHB(ctx, c->kmers.n * 16.0, k_owner_scatter, grid_for(c->kmers.n, 256), 256, (const uint64_t*)c->kmers.buf.p, c->kmers.n, cl_part_shift(k), (const uint32_t*)d_owner.p, W, d_cur.p, sorted.p);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	// counts each rank sends to me
	std::vector<uint64_t> mat((size_t)W * W);
	CL_TRY(c->X.all_gather_host(c->X.user, send.data(), W, mat.data()));
	std::vector<uint64_t> sb(W), rb(W); uint64_t n_recv = 0;
	for (uint32_t i = 0; i < W; ++i) { sb[i] = send[i] * 8; rb[i] = mat[(size_t)i * W + c->rank] * 8; n_recv += mat[(size_t)i * W + c->rank]; }
	c->kmers.buf.release(); c->kmers.n = 0;
	DevBuf<uint64_t> recv; DEV_ALLOC(ctx, recv, n_recv);
	CL_TRY(c->X.all_to_all_v(c->X.user, sorted.p, sb.data(), recv.p, rb.data()));
	c->kmers.buf = std::move(recv); c->kmers.n = n_recv;
	return CL_OK;
}

extern "C" cl_status cl_compressor_count_finish(cl_compressor* c, cl_kmer_stats* stats)
{
	if (!c) return CL_E_INVALID;
	cl_ctx* ctx = c->ctx;
	if (c->phase != 0) return cl_fail(ctx, CL_E_INVALID, "cl_compressor_count_finish: called twice");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const uint32_t W = c->world;
	cl_kmer_stats st{};
	if (W > 1) CL_TRY(exchange_kmers(c));
	CL_TRY(cl_kmer_count_filter(ctx, c->kmers.buf.p, c->kmers.n, c->P.k, c->P.ci, c->P.cs, &c->kset, &st));
	c->kmers.buf.release(); c->kmers.n = 0;
	c->first_read = 0; c->n_reads_total = c->n_reads_local;
	if (W > 1)
	{
		// replicate the filtered set: partitions are key ranges in rank order, so the concatenation is ascending
		uint64_t mine[6] = { st.tot_kmers, st.n_unique, st.n_unique_counted, st.total_count_filtered, c->n_reads_local, 0 };
		std::vector<uint64_t> all((size_t)W * 6);
		CL_TRY(c->X.all_gather_host(c->X.user, mine, 6, all.data()));
		std::vector<uint64_t> kb(W), cb(W); uint64_t n_all = 0;
		st = cl_kmer_stats{}; c->n_reads_total = 0;
		for (uint32_t r = 0; r < W; ++r)
		{
			const uint64_t* a = &all[(size_t)r * 6];
			st.tot_kmers += a[0]; st.n_unique += a[1]; st.n_unique_counted += a[2]; st.total_count_filtered += a[3];
			if (r < c->rank) c->first_read += a[4];
			c->n_reads_total += a[4];
			kb[r] = a[2] * 8; cb[r] = a[2] * 4; n_all += a[2];
		}
		if (n_all >= (1ull << 32)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_compressor_count_finish: >= 2^32 kept k-mers");
		DevBuf<uint64_t> keys; DevBuf<uint32_t> counts; DEV_ALLOC(ctx, keys, n_all); DEV_ALLOC(ctx, counts, n_all);
		CL_TRY(c->X.all_gather_v(c->X.user, cl_kmer_set_keys(c->kset), kb[c->rank], keys.p, kb.data()));
		CL_TRY(c->X.all_gather_v(c->X.user, cl_kmer_set_counts(c->kset), cb[c->rank], counts.p, cb.data()));
		cl_kmer_set_free(c->kset); c->kset = nullptr;
		CL_TRY(cl_kmer_set_create(ctx, keys.p, counts.p, n_all, c->P.k, &c->kset));
	}
	if (c->n_reads_total >= (1ull << 30)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_compressor: >= 2^30 reads (reference ids are 30-bit, hm_compact.h:545-552)");
	st.n_reads = c->n_reads_total;
	c->gstats = st;
	// host scalars of compression.cpp:443,501-503 and the acceptor's decisions (one stream over the whole input, a6)
	const uint64_t n = c->n_reads_total;
	c->mean_read_len = n ? (uint64_t)((double)(st.tot_kmers * c->P.f) / n + c->P.k - 1) : 0;
	if (c->genome_seqs && n)
	{	// the counter saw the genome's sequences as reads too; the statistics are corrected for them (compression.cpp:443-449)
		const uint64_t n_all = n + c->genome_seqs;
		const uint64_t m0 = (uint64_t)((double)(st.tot_kmers * c->P.f) / n_all + c->P.k - 1);
		c->mean_read_len = (uint64_t)((double)(m0 * n_all - c->genome_len) / (double)(n_all - c->genome_seqs));
	}
	c->h_accept.assign(c->n_reads_local, 1);
	if (c->P.sparse && n)
	{
		uint32_t range = (uint32_t)((c->P.sparse_g * (double)st.n_unique_counted * c->P.f) / (double)(c->mean_read_len ? c->mean_read_len : 1));
		if (range < 1) range = 1;
		c->sparse_range = range;
		std::vector<uint8_t> all(n);
		CL_TRY(cl_ref_accept((uint32_t)n, 0, range, c->P.sparse_exponent, all.data()));
		std::copy(all.begin() + c->first_read, all.begin() + c->first_read + c->n_reads_local, c->h_accept.begin());
	}
	c->phase = 1;
	if (stats) *stats = st;
	return CL_OK;
}

// ---- pass 2a --------------------------------------------------------------------------------------------------------
// Reference-genome mode (reference_genome.cpp:391-419, reads_sim_graph.cpp:295-322): the overlapping pieces of the genome are
// reference reads 0 .. n_pseudo-1 — always accepted, their k-mer lists not capped — ahead of the first read of the input.
extern "C" cl_status cl_compressor_pseudo_reads(cl_compressor* c, const cl_reads* pseudo)
{
	if (!c || !pseudo) return CL_E_INVALID;
	cl_ctx* ctx = c->ctx;
	if (c->phase != 1 || c->refs_chunk != 0 || c->n_pseudo) return cl_fail(ctx, CL_E_INVALID, "cl_compressor_pseudo_reads: once, after count_finish and before the first refs_add");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const uint32_t n = pseudo->n_reads;
	if (!n) return CL_OK;
	// Sharded reads: every rank is given the same pseudo reads; they are reference reads 0 .. n-1 of the replicated store, and rank 0
	// — whose references come first in the gather of pass 2a — is the one that contributes them and their index entries.  The
	// other ranks only note their number (reference ids, the coder's first read id and the acceptor's stream start behind them).
	if (c->world > 1 && c->rank != 0)
	{
		c->n_pseudo = n;
		if (c->P.sparse && c->n_reads_total)
		{
			std::vector<uint8_t> all((size_t)n + c->n_reads_total);
			CL_TRY(cl_ref_accept((uint32_t)c->n_reads_total, n, c->sparse_range, c->P.sparse_exponent, all.data()));
			std::copy(all.begin() + n + c->first_read, all.begin() + n + c->first_read + c->n_reads_local, c->h_accept.begin());
		}
		return CL_OK;
	}
	DevBuf<uint8_t> accept; DEV_ALLOC(ctx, accept, n);
	HIP_TRY(ctx, hipMemsetAsync(accept.p, 1, n, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	cl_kmer_lists* lists = nullptr;
	CL_TRY(cl_accepted_kmers(ctx, c->kset, pseudo, c->P.k, c->P.f, &lists));
	std::unique_ptr<cl_kmer_lists, void (*)(cl_kmer_lists*)> lg(lists, cl_kmer_lists_free);
	uint64_t n_sel = 0; uint32_t n_acc = 0;
	cl_status s = cl_index_entries_of(ctx, lists, accept.p, 0, nullptr, nullptr, 0, &n_sel, nullptr, &n_acc);
	if (s != CL_OK && s != CL_E_CAPACITY) return s;
	if (n_sel)
	{
		CL_TRY(c->pair_ids.reserve(ctx, n_sel)); CL_TRY(c->pair_refs.reserve(ctx, n_sel));
		CL_TRY(cl_index_entries_of(ctx, lists, accept.p, 0, c->pair_ids.buf.p, c->pair_refs.buf.p, n_sel, &n_sel, nullptr, nullptr));
		c->pair_ids.n = c->pair_refs.n = n_sel;
	}
	cl_reads* piece = nullptr;
	CL_TRY(cl_reads_select(ctx, pseudo, accept.p, &piece));
	c->ref_pieces.push_back(piece);
	c->n_refs_local = n; c->n_pseudo = n;
	// the acceptor's stream with the pseudo reads in front (ref_reads_accepter.h:41-58): decisions of the real reads follow them
	if (c->P.sparse && c->n_reads_total)
	{
		std::vector<uint8_t> all((size_t)n + c->n_reads_total);
		CL_TRY(cl_ref_accept((uint32_t)c->n_reads_total, n, c->sparse_range, c->P.sparse_exponent, all.data()));
		std::copy(all.begin() + n + c->first_read, all.begin() + n + c->first_read + c->n_reads_local, c->h_accept.begin());
	}
	return CL_OK;
}

extern "C" cl_status cl_compressor_refs_add(cl_compressor* c, const cl_reads* chunk)
{
	if (!c || !chunk) return CL_E_INVALID;
	cl_ctx* ctx = c->ctx;
	if (c->phase != 1) return cl_fail(ctx, CL_E_INVALID, "cl_compressor_refs_add: call after count_finish and before refs_finish");
	if (c->refs_chunk >= c->chunk_reads.size() || c->chunk_reads[c->refs_chunk] != chunk->n_reads)
		return cl_fail(ctx, CL_E_INVALID, "cl_compressor_refs_add: chunks must come in the order and sizes of pass 1");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const uint32_t n = chunk->n_reads;
	DevBuf<uint8_t> accept; DEV_ALLOC(ctx, accept, n);
	{
		DevBuf<uint8_t> d_acc; DEV_ALLOC(ctx, d_acc, n);
		if (n) HIP_TRY(ctx, hipMemcpyAsync(d_acc.p, c->h_accept.data() + c->refs_reads_seen, n, hipMemcpyHostToDevice, ctx->stream));
		if (n) LAUNCH(ctx, k_accept_flags2, grid_for(n, 256), 256, (const uint8_t*)d_acc.p, (const uint8_t*)chunk->has_n.p, n, accept.p);
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	// the chunk's reference reads first (a tenth of its reads with the sparse acceptor), THEIR accepted k-mers only: rounds 2-5 listed the
	// k-mers of every read of the chunk here and kept those of the accepted ones — 50 scans of a gigabase in a pass that nothing overlaps
	c->bounds.emplace_back();
	DEV_ALLOC(ctx, c->bounds.back(), (uint64_t)n + 1);
	uint32_t n_acc = 0;
	CL_TRY(cl_ref_bounds(ctx, accept.p, n, c->n_refs_local, c->bounds.back().p, &n_acc));
	cl_reads* piece = nullptr;
	CL_TRY(cl_reads_select(ctx, chunk, accept.p, &piece));
	c->ref_pieces.push_back(piece);
	if (n_acc)
	{
		cl_kmer_lists* lists = nullptr;
		CL_TRY(cl_accepted_kmers(ctx, c->kset, piece, c->P.k, c->P.f, &lists));
		std::unique_ptr<cl_kmer_lists, void (*)(cl_kmer_lists*)> lg(lists, cl_kmer_lists_free);
		DevBuf<uint8_t> all; DEV_ALLOC(ctx, all, n_acc);
		HIP_TRY(ctx, hipMemsetAsync(all.p, 1, n_acc, ctx->stream));            // (every read of the piece is a reference read: ref = base + its index)
		uint64_t n_sel = 0;
		cl_status s = cl_index_entries_of(ctx, lists, all.p, c->n_refs_local, nullptr, nullptr, 0, &n_sel, nullptr, nullptr);
		if (s != CL_OK && s != CL_E_CAPACITY) return s;
		if (n_sel)
		{
			CL_TRY(c->pair_ids.reserve(ctx, c->pair_ids.n + n_sel)); CL_TRY(c->pair_refs.reserve(ctx, c->pair_refs.n + n_sel));
			CL_TRY(cl_index_entries_of(ctx, lists, all.p, c->n_refs_local, c->pair_ids.buf.p + c->pair_ids.n, c->pair_refs.buf.p + c->pair_refs.n, n_sel, &n_sel, nullptr, nullptr));
			c->pair_ids.n += n_sel; c->pair_refs.n += n_sel;
		}
	}
	c->n_refs_local += n_acc;
	c->refs_reads_seen += n; ++c->refs_chunk;
	return CL_OK;
}

extern "C" cl_status cl_compressor_refs_finish(cl_compressor* c)
{
	if (!c) return CL_E_INVALID;
	cl_ctx* ctx = c->ctx;
	if (c->phase != 1 || c->refs_chunk != c->chunk_reads.size()) return cl_fail(ctx, CL_E_INVALID, "cl_compressor_refs_finish: every chunk of pass 1 must have been listed");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const uint32_t W = c->world;
	// this rank's reference reads as one arena
	uint64_t words = 0; uint32_t nr = 0;
	for (auto* p : c->ref_pieces) { words += p->total_words; nr += p->n_reads; }
	DevBuf<uint64_t> pk; DevBuf<uint32_t> iv, ln; DEV_ALLOC(ctx, pk, words + 1); DEV_ALLOC(ctx, iv, words + 1); DEV_ALLOC(ctx, ln, (uint64_t)nr + 1);
	{
		uint64_t wo = 0; uint32_t ro = 0;
		for (auto* p : c->ref_pieces)
		{
			if (p->total_words) { HIP_TRY(ctx, hipMemcpyAsync(pk.p + wo, p->packed.p, p->total_words * 8, hipMemcpyDeviceToDevice, ctx->stream)); HIP_TRY(ctx, hipMemcpyAsync(iv.p + wo, p->inv.p, p->total_words * 4, hipMemcpyDeviceToDevice, ctx->stream)); }
			if (p->n_reads) HIP_TRY(ctx, hipMemcpyAsync(ln.p + ro, p->lens.p, (uint64_t)p->n_reads * 4, hipMemcpyDeviceToDevice, ctx->stream));
			wo += p->total_words; ro += p->n_reads;
		}
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		for (auto* p : c->ref_pieces) cl_reads_free(p);
		c->ref_pieces.clear();
	}
	c->ref_base = 0; c->n_refs_total = nr;
	uint64_t n_pairs = c->pair_ids.n;
	if (W > 1)
	{
		uint64_t mine[4] = { nr, words, c->pair_ids.n, 0 };
		std::vector<uint64_t> all((size_t)W * 4);
		CL_TRY(c->X.all_gather_host(c->X.user, mine, 4, all.data()));
		std::vector<uint64_t> b_pk(W), b_iv(W), b_ln(W), b_pr(W); uint64_t t_words = 0, t_reads = 0, t_pairs = 0;
		for (uint32_t r = 0; r < W; ++r)
		{
			const uint64_t* a = &all[(size_t)r * 4];
			if (r < c->rank) c->ref_base += (uint32_t)a[0];
			t_reads += a[0]; t_words += a[1]; t_pairs += a[2];
			b_ln[r] = a[0] * 4; b_pk[r] = a[1] * 8; b_iv[r] = a[1] * 4; b_pr[r] = a[2] * 4;
		}
		if (t_reads >= (1ull << 30)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_compressor: >= 2^30 reference reads");
		c->n_refs_total = (uint32_t)t_reads;
		// global reference ids: this rank's references follow those of the lower ranks (file order)
		if (c->ref_base)
		{
			if (c->pair_refs.n) LAUNCH(ctx, k_add_u32, grid_for(c->pair_refs.n, 256), 256, c->pair_refs.buf.p, c->pair_refs.n, c->ref_base);
			for (size_t i = 0; i < c->bounds.size(); ++i) { const uint64_t m = (uint64_t)c->chunk_reads[i] + 1; LAUNCH(ctx, k_add_u32, grid_for(m, 256), 256, c->bounds[i].p, m, c->ref_base); }
			HIP_TRY(ctx, hipGetLastError());
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		}
		DevBuf<uint64_t> g_pk; DevBuf<uint32_t> g_iv, g_ln, g_id, g_rf;
		DEV_ALLOC(ctx, g_pk, t_words + 1); DEV_ALLOC(ctx, g_iv, t_words + 1); DEV_ALLOC(ctx, g_ln, t_reads + 1); DEV_ALLOC(ctx, g_id, t_pairs + 1); DEV_ALLOC(ctx, g_rf, t_pairs + 1);
		CL_TRY(c->X.all_gather_v(c->X.user, pk.p, b_pk[c->rank], g_pk.p, b_pk.data()));
		CL_TRY(c->X.all_gather_v(c->X.user, iv.p, b_iv[c->rank], g_iv.p, b_iv.data()));
		CL_TRY(c->X.all_gather_v(c->X.user, ln.p, b_ln[c->rank], g_ln.p, b_ln.data()));
		CL_TRY(c->X.all_gather_v(c->X.user, c->pair_ids.buf.p, b_pr[c->rank], g_id.p, b_pr.data()));
		CL_TRY(c->X.all_gather_v(c->X.user, c->pair_refs.buf.p, b_pr[c->rank], g_rf.p, b_pr.data()));
		pk = std::move(g_pk); iv = std::move(g_iv); ln = std::move(g_ln);
		c->pair_ids.buf = std::move(g_id); c->pair_refs.buf = std::move(g_rf); n_pairs = t_pairs;
		nr = (uint32_t)t_reads;
	}
	CL_TRY(cl_reads_from_arena(ctx, pk.p, iv.p, ln.p, nr, &c->refs));
	pk.release(); iv.release(); ln.release();
	CL_TRY(cl_index_build_pairs(ctx, c->kset, c->pair_ids.buf.p, c->pair_refs.buf.p, n_pairs, nullptr, 0, c->n_refs_total, c->n_pseudo, c->P.cs, &c->index));
	c->pair_ids.buf.release(); c->pair_refs.buf.release(); c->pair_ids.n = c->pair_refs.n = 0;
	// the coders of this rank's model domain; cur_read_id starts at the global index of the first read, so that the ids of
	// references from lower ranks fit the byte count the coder derives from it (dna_coder.cpp:26-63)
	CL_TRY(cl_dna_coder_create(ctx, c->P.c, c->P.level, (uint32_t)c->first_read + c->n_pseudo, &c->dna));     // (pseudo reads count as reads 0 .. n_pseudo-1, dna_coder.cpp:1242-1250)
	if (c->has_qual)
	{
		cl_ctx* qc = (c->P.level <= 1) ? c->qctx : ctx;        // levels 2 and 3 need the edit scripts: same stream as the DNA path
		const cl_status s = cl_qual_coder_create(qc, &c->Q, &c->qual);
		if (s != CL_OK) return cl_fail(ctx, s, std::string("quality coder: ") + cl_last_error(qc));
	}
	c->phase = 2;
	return CL_OK;
}

// ---- pass 2b --------------------------------------------------------------------------------------------------------
// Stage A of a chunk on context `ctx` (the caller's, or an encode lane's): a4 accepted k-mers, a5 candidates among the
// EARLIER reference reads (d_bounds), a8/a9 anchors, a10-a12 edit scripts -> tuple streams.  Reads only state that pass 2a
// completed (set, index, reference reads), so chunks are independent here.
static cl_status stage_a(cl_compressor* c, cl_ctx* ctx, size_t chunk_idx, const cl_reads* reads, const uint32_t* h_pack_bounds, uint32_t n_packs, cl_compressor::Prepared& out)
{
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const cl_compress_params* P = &c->P;
	const uint32_t n = reads->n_reads;
	const uint32_t* d_bounds = c->bounds[chunk_idx].p;
	cl_kmer_lists* lists = nullptr;
	CL_TRY(cl_accepted_kmers(ctx, c->kset, reads, P->k, P->f, &lists));
	std::unique_ptr<cl_kmer_lists, void (*)(cl_kmer_lists*)> lg(lists, cl_kmer_lists_free);
	// more than 16 candidates per read / more than 8 recursion levels (the reference takes any value; its presets stop at 12 and 6): this
	// build's frames hold 16 candidate views and 10 levels — the read is coded against its 16 best candidates, to depth 8.  Still a
	// valid archive (the alternative-id model keeps the alphabet of P->c symbols the `meta` stream announces), a little larger than
	// the reference's would be; never a failed call.
	const uint32_t cc = std::min<uint32_t>(P->c, 16), max_rec = std::min<uint32_t>(P->max_rec, 8);
	DevBuf<uint32_t> crefs, votes, cnt; DEV_ALLOC(ctx, crefs, (uint64_t)n * cc); DEV_ALLOC(ctx, votes, (uint64_t)n * cc); DEV_ALLOC(ctx, cnt, n);
	CL_TRY(cl_candidates_at(ctx, c->index, lists, d_bounds, cc, crefs.p, votes.p, cnt.p));
	votes.release();
	DevBuf<uint64_t> common_off, common;
	const bool hifi = P->source == 2;
	if (hifi)
	{
		DEV_ALLOC(ctx, common_off, (uint64_t)n * cc + 1);
		uint64_t need = 0;
		cl_status s = cl_candidates_common(ctx, c->index, lists, cc, crefs.p, cnt.p, common_off.p, nullptr, 0, &need);
		if (s != CL_OK && s != CL_E_CAPACITY) return s;
		DEV_ALLOC(ctx, common, need + 1);
		CL_TRY(cl_candidates_common(ctx, c->index, lists, cc, crefs.p, cnt.p, common_off.p, common.p, need, &need));
	}
	lg.reset();
	cl_anchors* anc = nullptr;
	CL_TRY(cl_anchor_candidates_hifi(ctx, reads, c->refs, crefs.p, cnt.p, cc, P->anchor_len, P->frac_always, P->frac_min, P->max_matches_mult, P->min_anchors,
		P->k, P->f, hifi ? common_off.p : nullptr, hifi ? common.p : nullptr, &anc));
	std::unique_ptr<cl_anchors, void (*)(cl_anchors*)> ag(anc, cl_anchors_free);
	out.n_anchors = cl_anchors_total(anc);
	crefs.release(); cnt.release(); common_off.release(); common.release();
	const uint64_t es_cap = reads->total_bases + 16ull * n + 4096;
	DEV_ALLOC(ctx, out.es, es_cap); DEV_ALLOC(ctx, out.es_off, (uint64_t)n + 1); DEV_ALLOC(ctx, out.es_nt, n);
	CL_TRY(cl_encode_reads(ctx, reads, c->refs, anc, cc, P->anchor_len, P->min_part_alt, max_rec, P->cost_mult, h_pack_bounds, n_packs, out.es.p, es_cap, out.es_off.p, out.es_nt.p, &out.es_bytes));
	return CL_OK;
}

static void lane_main(cl_compressor* c, cl_ctx* lane)
{
	for (;;)
	{
		size_t idx; cl_compressor::Prepared* job;
		{
			std::unique_lock<std::mutex> l(c->lane_mu);
			// the lanes run at most (lanes + 2) chunks ahead of the coders: what they finish (tuple streams, ~1.5 GB per Gbase) waits in
			// HBM until it is coded; the slack evens out chunks whose stage A or coders happen to be slow
			const auto tw = std::chrono::steady_clock::now();
			c->lane_cv.wait(l, [&]() { return c->lane_stop || (!c->lane_queue.empty() && c->lane_queue.front() <= c->enc_chunk + c->lane_ctx.size() + 1); });
			c->w_lane_idle += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
			if (c->lane_stop) return;
			idx = c->lane_queue.front(); c->lane_queue.pop_front();
			job = c->prepared[idx].get();
		}
		lane->timing = c->ctx->timing;
		const auto tw = std::chrono::steady_clock::now();
		const cl_status s = stage_a(c, lane, idx, job->reads, job->packs.data(), (uint32_t)job->packs.size() - 1, *job);
		cl_timing_collect(lane);
		{
			std::lock_guard<std::mutex> l(c->lane_mu);
			c->w_lane_work += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
			job->status = s; if (s != CL_OK) job->err = lane->err;
			job->times.swap(lane->times); lane->times.clear();
			job->done = true;
		}
		c->lane_cv.notify_all();
	}
}

// The DNA coder's chain per chunk was: tuple walks -> triple slots -> stable sort by (family, context) -> context runs -> model
// evolution -> interval coding, all on the caller's stream — after the aligner work of round 3 THE critical chain of a pass
// (22.5 of 25.7 s busy).  Everything before the model evolution depends on the tuple streams only (and on two scalars that chain
// from walk to walk), so this thread does it for the chunks ahead, in order, on a context of its own; the caller's stream keeps
// evolution and coding.
// what the device could still give this process: free memory + what the shared pool holds without using it
static uint64_t avail_bytes(cl_ctx* ctx)
{
	size_t fr = 0, tot = 0;
	if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return fr + (ctx->pool.reserved - std::min(ctx->pool.reserved, ctx->pool.live_bytes));
}
static void prep_main(cl_compressor* c, cl_ctx* ctx)
{
	for (;;)
	{
		size_t idx; cl_compressor::Prepared* job; uint32_t types_in = 0, read_id_in = 0, types_out = 0; cl_status s = CL_OK;
		{	// one claim at a time: the scalars of chunk idx + 1 follow from those of chunk idx
			std::lock_guard<std::mutex> claim(c->prep_claim_mu);
			{
				std::unique_lock<std::mutex> l(c->lane_mu);
				const auto tw = std::chrono::steady_clock::now();
				struct Lap { cl_compressor* c; std::chrono::steady_clock::time_point t; ~Lap() { c->w_prep_idle += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); } } lap_idle{ c, tw };
				c->lane_cv.wait(l, [&]() {
					if (c->lane_stop || c->prep_broken) return true;
					if (c->prep_next < c->enc_chunk) return true;                        // (a chunk was coded without these threads: the chain of walk scalars is lost)
					auto it = c->prepared.find(c->prep_next);
					return it != c->prepared.end() && it->second->done && c->prep_next <= c->enc_chunk + c->prep_ctxs.size() + c->evolve_depth;     // (far enough ahead for the chunks that may be evolved ahead)
				});
				if (c->lane_stop || c->prep_broken) return;
				if (c->prep_next < c->enc_chunk) { c->prep_broken = true; c->lane_cv.notify_all(); return; }
				idx = c->prep_next; job = c->prepared[idx].get();
				types_in = c->prep_types; read_id_in = c->prep_read_id;
			}
			const uint32_t n = job->reads->n_reads;
			types_out = types_in;
			if (job->status == CL_OK && n) s = cl_dna_batch_types(ctx, job->es.p, job->es_off.p, n, job->es_bytes, types_in, &types_out);
			{
				std::lock_guard<std::mutex> l(c->lane_mu);
				if (s == CL_OK && job->status == CL_OK) { c->prep_types = types_out; c->prep_read_id += n; }
				c->prep_next = idx + 1;
			}
			c->lane_cv.notify_all();
		}
		DnaWalked* W = nullptr; uint32_t walked_types = types_out;
		const uint32_t n = job->reads->n_reads;
		if (s == CL_OK && job->status == CL_OK && n)
		{
			ctx->timing = c->ctx->timing;
			s = cl_dna_prepare_batch(ctx, c->dna, c->refs, job->es.p, job->es_off.p, job->es_nt.p, n, types_in, read_id_in,
			                         job->parts.empty() ? nullptr : job->parts.data(), job->parts.empty() ? 0u : (uint32_t)job->parts.size() - 1, &W, &walked_types);
			cl_timing_collect(ctx);
			if (s == CL_OK && walked_types != types_out) s = cl_fail(ctx, CL_E_INVALID, "dna preparation: the read types of a chunk changed between the claim and the walk");
		}
		{
			std::lock_guard<std::mutex> l(c->lane_mu);
			if (s == CL_OK && job->status == CL_OK) job->walked = W;
			else { if (W) cl_dna_walked_free(W); if (job->status == CL_OK) c->prep_broken = true; }   // (the caller's thread walks this chunk itself and reports what fails)
			job->dna_times.swap(ctx->times); ctx->times.clear();
			job->dna_done = true;
		}
		c->lane_cv.notify_all();
	}
}

// The quality coder's chain per chunk — symbols -> stable sort by context -> context runs -> model evolution -> interval coding —
// became the critical one once the DNA coder's first half had moved to prep_main.  Its first three steps depend on the input only:
// this thread makes them for the chunks ahead (one or two), on a context of its own.
static void qprep_main(cl_compressor* c)
{
	cl_ctx* ctx = c->qprep_ctx;
	for (;;)
	{
		size_t idx; cl_compressor::Prepared* job;
		{
			std::unique_lock<std::mutex> l(c->lane_mu);
			c->lane_cv.wait(l, [&]() {
				if (c->lane_stop) return true;
				if (c->qprep_next < c->enc_chunk) return true;
				auto it = c->prepared.find(c->qprep_next);
				return it != c->prepared.end() && c->qprep_next <= c->enc_chunk + 1 + c->evolve_depth;
			});
			if (c->lane_stop) return;
			if (c->qprep_next < c->enc_chunk) { c->qprep_next = c->enc_chunk; continue; }   // (chunks coded without an announcement: nothing chains here, catch up)
			idx = c->qprep_next; job = c->prepared[idx].get();
		}
		QualPrepared* P = nullptr;
		if (job->d_quals && job->d_base_off && !job->parts.empty() && job->reads->n_reads)
		{
			ctx->timing = c->ctx->timing;
			const cl_status s = cl_qual_prepare_batch(ctx, c->qual, job->reads, job->d_quals, job->d_base_off, nullptr, job->parts.data(), (uint32_t)job->parts.size() - 1, &P);
			cl_timing_collect(ctx);
			if (s != CL_OK) P = nullptr;                                             // (the caller's thread prepares this chunk itself and reports what fails)
		}
		{
			std::lock_guard<std::mutex> l(c->lane_mu);
			job->qprep = P;
			job->q_times.swap(ctx->times); ctx->times.clear();
			job->q_done = true;
			c->qprep_next = idx + 1;
		}
		c->lane_cv.notify_all();
	}
}

extern "C" cl_status cl_compressor_prepare_parts(cl_compressor* c, const cl_reads* reads, const uint32_t* h_pack_bounds, uint32_t n_packs, const uint32_t* h_part_bounds, uint32_t n_parts,
                                                 const uint8_t* d_quals, const uint64_t* d_base_off);
extern "C" cl_status cl_compressor_prepare(cl_compressor* c, const cl_reads* reads, const uint32_t* h_pack_bounds, uint32_t n_packs)
{
	return cl_compressor_prepare_parts(c, reads, h_pack_bounds, n_packs, nullptr, 0, nullptr, nullptr);
}
extern "C" cl_status cl_compressor_prepare_parts(cl_compressor* c, const cl_reads* reads, const uint32_t* h_pack_bounds, uint32_t n_packs, const uint32_t* h_part_bounds, uint32_t n_parts,
                                                 const uint8_t* d_quals, const uint64_t* d_base_off)
{
	if (!c || !reads || !h_pack_bounds) return CL_E_INVALID;
	cl_ctx* ctx = c->ctx;
	if (c->phase != 2) return cl_fail(ctx, CL_E_INVALID, "cl_compressor_prepare: call after refs_finish");
	std::unique_lock<std::mutex> l(c->lane_mu);
	const size_t idx = std::max(c->n_announced, c->enc_chunk);
	if (idx < c->enc_chunk || idx >= c->chunk_reads.size() || c->chunk_reads[idx] != reads->n_reads)
		return cl_fail(ctx, CL_E_INVALID, "cl_compressor_prepare: chunks must be announced in the order and sizes of pass 1, before they are encoded");
	if (c->lane_ctx.empty())
	{
		// two lanes.  (A third one was measured at 50 Gbases: once 19.3 against 20.5 s per pass, then — same code but for the coders'
		// own streams — 23.5 against 20.3 s on one box, twice: the machine is shared by ~20 streams and whatever the lanes gain the
		// preparation threads lose.  COLORD_HIP_ENCODE_LANES overrides.)
		uint32_t lanes = 2;
		// Long coder parts (the reference's packs of 4 Mi symbols: the byte-identical mode) make the interval coders the bound of a
		// chunk — a dependent chain of 1.3 s per part: then the model halves of the next two chunks are done ahead so that the coders of
		// three chunks run side by side, and two lanes feed them easily.  With short parts (the bench's 64 Ki) the coders are no bound
		// and the memory serves a third lane better.
		const bool long_parts = h_part_bounds && n_parts && (reads->total_bases + reads->n_reads) / n_parts >= (1u << 19);
		c->evolve_depth = long_parts ? 2 : 0;
		if (long_parts) lanes = 2;
		if (const char* e = getenv("COLORD_HIP_ENCODE_LANES")) lanes = (uint32_t)std::min(4, std::max(1, atoi(e)));
		if (const char* e = getenv("COLORD_HIP_EVOLVE_DEPTH")) c->evolve_depth = (uint32_t)std::min(3, std::max(0, atoi(e)));
		while (ctx->lanes.size() < lanes)
		{
			cl_ctx* x = nullptr;
			const cl_status s = cl_ctx_create(ctx->device, &x);
			if (s != CL_OK) return cl_fail(ctx, s, "cl_compressor_prepare: no context for an encode lane");
			cl_ctx_set_priority(x, getenv("COLORD_HIP_NO_STREAM_PRIO") ? 0 : +1, CL_ROLE_LANE);      // the lanes bound a pass: their queues are served first
			ctx->lanes.push_back(x);
		}
		c->lane_ctx.assign(ctx->lanes.begin(), ctx->lanes.begin() + lanes);
		for (size_t li = 0; li < c->lane_ctx.size(); ++li) c->lane_threads.emplace_back(lane_main, c, c->lane_ctx[li]);
		// the DNA preparation thread: only from the first chunk on (its walk scalars chain from chunk to chunk)
		if (idx == 0 && c->enc_chunk == 0 && !getenv("COLORD_HIP_NO_DNA_PREP"))
		{
			// ONE worker.  (Two, claiming alternate chunks, were measured in round 5 when this chain was the busiest queue of a pass — 87 %: each
			// one's sort took twice as long beside the other's, 18.7 against 18.7 s per pass: the machine is the bound, not the chain.)
			if (!ctx->prep)
			{
				cl_ctx* x = nullptr;
				const cl_status s = cl_ctx_create(ctx->device, &x);
				if (s != CL_OK) return cl_fail(ctx, s, "cl_compressor_prepare: no context for the DNA preparation thread");
				cl_ctx_set_priority(x, getenv("COLORD_HIP_NO_STREAM_PRIO") ? 0 : -1, CL_ROLE_PREP);  // (works ahead: takes what the lanes and coders leave)
				ctx->prep = x;
			}
			c->prep_ctxs.assign(1, ctx->prep);
			c->prep_next = 0; c->prep_on = true;
			cl_dna_coder_state(c->dna, &c->prep_types, &c->prep_read_id);
			for (cl_ctx* pc : c->prep_ctxs) c->prep_threads.emplace_back(prep_main, c, pc);
		}
		// the quality preparation thread: level 1 only (above, the contexts take flags from the edit scripts), quality context of its own
		if (c->qual && c->P.level <= 1 && c->qctx && c->qctx != ctx && !getenv("COLORD_HIP_NO_QUAL_PREP"))
		{
			if (!ctx->qprep)
			{
				cl_ctx* x = nullptr;
				const cl_status s = cl_ctx_create(ctx->device, &x);
				if (s != CL_OK) return cl_fail(ctx, s, "cl_compressor_prepare: no context for the quality preparation thread");
				cl_ctx_set_priority(x, getenv("COLORD_HIP_NO_STREAM_PRIO") ? 0 : -1, CL_ROLE_PREP);
				ctx->qprep = x;
			}
			c->qprep_ctx = ctx->qprep; c->qprep_next = idx; c->qprep_on = true;
			c->qprep_thread = std::thread(qprep_main, c);
		}
	}
	auto job = std::make_unique<cl_compressor::Prepared>();
	job->reads = reads; job->packs.assign(h_pack_bounds, h_pack_bounds + n_packs + 1);
	if (h_part_bounds && n_parts) job->parts.assign(h_part_bounds, h_part_bounds + n_parts + 1);
	job->d_quals = d_quals; job->d_base_off = d_base_off;
	c->prepared[idx] = std::move(job);
	c->lane_queue.push_back(idx);
	c->n_announced = idx + 1;
	l.unlock();
	c->lane_cv.notify_all();
	return CL_OK;
}

extern "C" cl_status cl_compressor_encode(cl_compressor* c, const cl_reads* reads, const uint8_t* d_quals, const uint64_t* d_base_off,
                                          const uint32_t* h_part_bounds, uint32_t n_parts, const uint32_t* h_pack_bounds, uint32_t n_packs,
                                          uint8_t* d_dna_out, uint64_t dna_cap, uint64_t* h_dna_part_sizes,
                                          uint8_t* d_qual_out, uint64_t qual_cap, uint64_t* h_qual_part_sizes, cl_compress_info* info)
{
	if (!c || !reads || !h_part_bounds || !h_pack_bounds || !info) return CL_E_INVALID;
	cl_ctx* ctx = c->ctx;
	if (c->phase != 2) return cl_fail(ctx, CL_E_INVALID, "cl_compressor_encode: call after refs_finish");
	if (c->enc_chunk >= c->chunk_reads.size() || c->chunk_reads[c->enc_chunk] != reads->n_reads)
		return cl_fail(ctx, CL_E_INVALID, "cl_compressor_encode: chunks must come in the order and sizes of pass 1");
	if (c->has_qual && (!d_quals || !d_base_off || !h_qual_part_sizes || !d_qual_out)) return cl_fail(ctx, CL_E_INVALID, "cl_compressor_encode: quality stream without qualities");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const cl_compress_params* P = &c->P;
	memset(info, 0, sizeof(*info));
	const uint32_t n = reads->n_reads;
	const size_t idx = c->enc_chunk;
	info->n_reads = n; info->n_bases = reads->total_bases; info->tot_kmers = c->gstats.tot_kmers; info->n_kept_kmers = c->gstats.n_unique_counted;
	info->n_refs = c->n_refs_total; info->sparse_range = c->sparse_range;
	// the chunk leaves the look-ahead window whatever happens below (lanes may go on to the next announced chunk)
	struct Advance { cl_compressor* c; size_t idx; ~Advance() { { std::lock_guard<std::mutex> l(c->lane_mu); c->prepared.erase(idx); c->bounds[idx].release(); c->enc_chunk = idx + 1; } c->lane_cv.notify_all(); } };
	std::unique_ptr<cl_compressor::Prepared> own;          // stage A result: announced (taken from the lanes) or made here
	cl_compressor::Prepared* job = nullptr;
	{
		std::unique_lock<std::mutex> l(c->lane_mu);
		auto it = c->prepared.find(idx);
		if (it != c->prepared.end())
		{
			if (it->second->reads != reads) return cl_fail(ctx, CL_E_INVALID, "cl_compressor_encode: not the chunk that was announced for this position");
			auto tw = std::chrono::steady_clock::now();
			auto lap = [&](double& acc) { const auto t = std::chrono::steady_clock::now(); acc += std::chrono::duration<double>(t - tw).count(); tw = t; };
			c->lane_cv.wait(l, [&]() { return it->second->done; });
			lap(c->w_enc_lane);
			if (c->prep_on && !c->prep_broken) c->lane_cv.wait(l, [&]() { return it->second->dna_done || c->prep_broken; });
			lap(c->w_enc_prep);
			if (c->qprep_on && c->qprep_next <= idx) c->lane_cv.wait(l, [&]() { return it->second->q_done; });
			lap(c->w_enc_qprep);
			own = std::move(it->second); c->prepared.erase(it);
			job = own.get();
		}
	}
	Advance adv{ c, idx };
	if (!n) return CL_OK;
	struct Joiner { std::thread t; ~Joiner() { if (t.joinable()) t.join(); } } qjob;
	cl_status qstatus = CL_OK;
	cl_ctx* qctx = c->qual ? cl_qual_coder_ctx(c->qual) : nullptr;
	const bool overlap = c->qual && P->level <= 1 && qctx && qctx != ctx;
	if (overlap && job && job->qprep && job->d_quals == d_quals) { cl_qual_set_ahead(c->qual, job->qprep); job->qprep = nullptr; ++c->n_qual_prep; }
	if (job) for (auto& kv : job->q_times) { auto& t = qctx->times[kv.first]; t.ms += kv.second.ms; t.launches += kv.second.launches; t.bytes += kv.second.bytes; }
	struct QHookOff { cl_qual_coder* q; ~QHookOff() { if (q) cl_qual_set_before_tail(q, nullptr); } } qhook_off{ overlap ? c->qual : nullptr };
	if (overlap && c->qprep_on)
		cl_qual_set_before_tail(c->qual, [c, idx, qctx]() -> cl_status {
			if (getenv("COLORD_HIP_NO_EVOLVE_AHEAD")) return CL_OK;
			for (;;)
			{
				const size_t k = std::max(c->qual_evolved_upto, idx + 1);
				if (k > idx + c->evolve_depth) return CL_OK;
				if (avail_bytes(c->ctx) < (24ull << 30)) return CL_OK;
				cl_compressor::Prepared* nx = nullptr; QualPrepared* P = nullptr;
				{
					std::unique_lock<std::mutex> l(c->lane_mu);
					auto it = c->prepared.find(k);
					if (it == c->prepared.end() || it->second->parts.empty() || !it->second->d_quals) return CL_OK;
					if (!it->second->q_done) return c->lane_stop ? CL_OK : CL_HOOK_RETRY;
					if (!it->second->qprep) return CL_OK;
					nx = it->second.get(); P = nx->qprep; nx->qprep = nullptr;
				}
				const cl_status s = cl_qual_evolve_ahead(qctx, c->qual, nx->reads, nx->d_quals, nx->d_base_off, nx->parts.data(), (uint32_t)nx->parts.size() - 1, P);
				if (s != CL_OK) return s;
				++c->n_qual_ahead; c->qual_evolved_upto = k + 1;
			}
		});
	if (overlap)
		qjob.t = std::thread([&]() { qstatus = cl_qual_encode(qctx, c->qual, reads, d_quals, d_base_off, nullptr, h_part_bounds, n_parts, d_qual_out, qual_cap, h_qual_part_sizes, &info->qual_bytes); });
	if (job)
	{
		if (job->status != CL_OK) return cl_fail(ctx, job->status, "encode lane: " + job->err);
		for (auto& kv : job->times) { auto& t = ctx->times[kv.first]; t.ms += kv.second.ms; t.launches += kv.second.launches; t.bytes += kv.second.bytes; t.cells += kv.second.cells; }
		for (auto& kv : job->dna_times) { auto& t = ctx->times[kv.first]; t.ms += kv.second.ms; t.launches += kv.second.launches; t.bytes += kv.second.bytes; }
		if (job->walked) { cl_dna_set_ahead(c->dna, job->walked); job->walked = nullptr; ++c->n_dna_prep; }
	}
	else
	{
		own = std::make_unique<cl_compressor::Prepared>();
		job = own.get();
		CL_TRY(stage_a(c, ctx, idx, reads, h_pack_bounds, n_packs, *job));
	}
	info->n_anchors = job->n_anchors; info->tuple_bytes = job->es_bytes;
	// While this chunk's interval coders run (a dependent chain per part that nothing else on this stream can use: ~0.1 s with parts
	// of 64 Ki symbols, 1.3 s with the reference's 4 Mi), the NEXT chunk is taken through its model half — evolution of the models,
	// triples, its own interval coders started (cl_dna_evolve_ahead / cl_qual_evolve_ahead) — when the preparation threads have it
	// ready; without them only its tuple walk (cl_dna_walk_ahead).
	cl_dna_set_before_tail(c->dna, [c, idx]() -> cl_status {
		if (c->prep_on && !c->prep_broken)
		{
			if (getenv("COLORD_HIP_NO_EVOLVE_AHEAD")) return CL_OK;
			for (;;)
			{
				const size_t k = std::max(c->dna_evolved_upto, idx + 1);
				if (k > idx + c->evolve_depth) return CL_OK;
				if (avail_bytes(c->ctx) < (28ull << 30)) return CL_OK;                    // (a chunk ahead holds ~12 GB of triples and coder output)
				cl_compressor::Prepared* nx = nullptr; DnaWalked* W = nullptr;
				{
					std::unique_lock<std::mutex> l(c->lane_mu);
					auto it = c->prepared.find(k);
					if (it == c->prepared.end() || it->second->parts.empty()) return CL_OK;
					if (!it->second->dna_done) return (c->prep_broken || c->lane_stop) ? CL_OK : CL_HOOK_RETRY;     // (never a wait here: the chunk being coded is due)
					if (!it->second->walked || it->second->status != CL_OK) return CL_OK;
					nx = it->second.get(); W = nx->walked; nx->walked = nullptr;
				}
				const cl_status s = cl_dna_evolve_ahead(c->ctx, c->dna, c->refs, nx->es.p, nx->es_off.p, nx->es_nt.p, nx->reads->n_reads, nx->parts.data(), (uint32_t)nx->parts.size() - 1, W);
				if (s != CL_OK) return s;
				++c->n_dna_ahead; c->dna_evolved_upto = k + 1;
			}
		}
		cl_compressor::Prepared* nx = nullptr;
		{
			std::lock_guard<std::mutex> l(c->lane_mu);
			auto it = c->prepared.find(idx + 1);
			if (it != c->prepared.end() && it->second->done && it->second->status == CL_OK && it->second->reads->n_reads) nx = it->second.get();
		}
		if (!nx) return CL_OK;
		return cl_dna_walk_ahead(c->ctx, c->dna, c->refs, nx->es.p, nx->es_off.p, nx->es_nt.p, nx->reads->n_reads);
	});
	struct HookOff { cl_dna_coder* d; ~HookOff() { cl_dna_set_before_tail(d, nullptr); } } hook_off{ c->dna };
	CL_TRY(cl_dna_encode(ctx, c->dna, c->refs, job->es.p, job->es_off.p, job->es_nt.p, n, h_part_bounds, n_parts, d_dna_out, dna_cap, h_dna_part_sizes, &info->dna_bytes));
	if (overlap)
	{
		qjob.t.join();
		if (qstatus != CL_OK) return cl_fail(ctx, qstatus, std::string("quality stream: ") + cl_last_error(qctx));
	}
	else if (c->qual)
	{
		DevBuf<uint8_t> flags;
		if (P->level > 1)
		{
			DEV_ALLOC(ctx, flags, reads->total_bases + 1);
			CL_TRY(cl_es_flags(ctx, reads, job->es.p, job->es_off.p, d_base_off, flags.p));
		}
		CL_TRY(cl_qual_encode(ctx, c->qual, reads, d_quals, d_base_off, P->level > 1 ? flags.p : nullptr, h_part_bounds, n_parts, d_qual_out, qual_cap, h_qual_part_sizes, &info->qual_bytes));
	}
	return CL_OK;
}

extern "C" cl_status cl_compressor_info(const cl_compressor* c, cl_kmer_stats* stats, uint64_t* first_read, uint64_t* n_reads_total, uint64_t* mean_read_len,
                                        uint32_t* sparse_range, uint32_t* n_refs_total)
{
	if (!c || c->phase < 1) return CL_E_INVALID;
	if (stats) *stats = c->gstats;
	if (first_read) *first_read = c->first_read;
	if (n_reads_total) *n_reads_total = c->n_reads_total;
	if (mean_read_len) *mean_read_len = c->mean_read_len;
	if (sparse_range) *sparse_range = c->sparse_range;
	if (n_refs_total) *n_refs_total = c->n_refs_total;
	return CL_OK;
}
