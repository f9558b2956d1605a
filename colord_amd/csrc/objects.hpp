// objects.hpp — device-resident objects behind the opaque C-ABI handles.
#pragma once
#include "common.hpp"
#include <memory>

struct cl_reads {
	cl_ctx* ctx = nullptr;
	uint32_t n_reads = 0;
	uint64_t total_bases = 0, total_words = 0;
	DevBuf<uint64_t> packed;      // total_words + 1 (tail word), 32 bases per word, first base in bits 63..62
	DevBuf<uint32_t> inv;         // total_words + 1, bit 31-j = base j invalid (N or pad)
	DevBuf<uint64_t> word_off;    // n_reads + 1
	DevBuf<uint32_t> lens;        // n_reads
	DevBuf<uint8_t> has_n;        // n_reads
};

struct cl_kmer_set {
	cl_ctx* ctx = nullptr;
	uint32_t k = 0;
	uint64_t n = 0;
	DevBuf<uint64_t> keys;        // ascending
	DevBuf<uint32_t> counts;      // min(count, cs)
	DevBuf<uint64_t> slots;       // buckets of 4 x {key, rank}: 8 uint64 per bucket
	uint64_t bmask = 0;
};

struct cl_kmer_lists {
	cl_ctx* ctx = nullptr;
	uint32_t n_reads = 0;
	uint64_t total = 0;
	DevBuf<uint64_t> off;         // n_reads + 1
	DevBuf<uint64_t> kmers;       // k-mer values
	DevBuf<uint32_t> ids;         // rank in the set
	DevBuf<uint32_t> pos;         // start position in the read
	DevBuf<uint32_t> read;        // owning read of each entry
};

struct cl_index {
	cl_ctx* ctx = nullptr;
	uint32_t n_reads = 0, n_refs = 0, n_pseudo = 0;
	uint64_t n_keys = 0, n_entries = 0;
	DevBuf<uint32_t> ref_rank;    // n_reads + 1: number of reference reads before read i
	DevBuf<uint64_t> off;         // n_keys + 1 (CSR over set ranks)
	DevBuf<uint32_t> refs;        // reference ids, ascending inside a list
};

// key-range partitioning of k-mers (kmer.hip): histogram over the 4096 bins of the top 12 key bits, gather of a bin range
uint32_t cl_part_shift(uint32_t k);
cl_status cl_key_histogram(cl_ctx* ctx, const uint64_t* d_kmers, uint64_t n, uint32_t k, std::vector<uint64_t>& h_bins);
cl_status cl_key_gather(cl_ctx* ctx, const uint64_t* d_kmers, uint64_t n, uint32_t k, uint32_t b0, uint32_t b1, uint64_t* d_out, uint64_t expect);

// graph.hip: reference reads before each read of a chunk (the first half of cl_index_entries_of on its own)
cl_status cl_ref_bounds(cl_ctx* ctx, const uint8_t* d_accept, uint32_t n, uint32_t ref_base, uint32_t* d_bounds, uint32_t* n_accepted);

// the DNA coder's state-independent half ahead of time (dna.hip): stream.hip walks the NEXT chunk's tuples from the hook that
// cl_dna_encode calls before it waits for its last interval coding
#include <functional>
struct cl_dna_coder;
cl_status cl_dna_walk_ahead(cl_ctx* ctx, cl_dna_coder* D, const cl_reads* refs, const uint8_t* d_es, const uint64_t* d_es_off, const uint32_t* d_es_ntuples, uint32_t n_reads);
void cl_dna_set_before_tail(cl_dna_coder* D, std::function<cl_status()> fn);
// the model-independent half of a batch (walks; with part bounds also layout, sort and context runs) on any context — see dna.hip
struct DnaWalked;
cl_status cl_dna_prepare_batch(cl_ctx* ctx, cl_dna_coder* D, const cl_reads* refs, const uint8_t* d_es, const uint64_t* d_es_off, const uint32_t* d_es_ntuples, uint32_t n_reads,
                               uint32_t prev_types, uint32_t read_id, const uint32_t* h_part_bounds, uint32_t n_parts, DnaWalked** out, uint32_t* prev_types_out);
void cl_dna_walked_free(DnaWalked* W);
void cl_dna_set_ahead(cl_dna_coder* D, DnaWalked* W);               // takes W: the next cl_dna_encode uses it if it is the batch it was made for
void cl_dna_coder_state(const cl_dna_coder* D, uint32_t* prev_types, uint32_t* read_id);
// the read-type history a batch hands to the next one (types of its last four reads after `prev_types`): from the first tuple of those reads alone
cl_status cl_dna_batch_types(cl_ctx* ctx, const uint8_t* d_es, const uint64_t* d_es_off, uint32_t n_reads, uint64_t es_bytes, uint32_t prev_types, uint32_t* out);
// the same for the quality coder (qual.hip): symbols, sort by context, context runs of a batch, on any context
struct QualPrepared;
cl_status cl_qual_prepare_batch(cl_ctx* ctx, cl_qual_coder* Q, const cl_reads* R, const uint8_t* d_quals, const uint64_t* d_qual_off,
                                const uint8_t* d_flags, const uint32_t* h_part_bounds, uint32_t n_parts, QualPrepared** out);
void cl_qual_prepared_free(QualPrepared* P);
void cl_qual_set_ahead(cl_qual_coder* Q, QualPrepared* P);
// the model half of the NEXT batch (takes W / P) while the interval coders of the current one run: from the coders' before_tail hooks
cl_status cl_dna_evolve_ahead(cl_ctx* ctx, cl_dna_coder* D, const cl_reads* refs, const uint8_t* d_es, const uint64_t* d_es_off, const uint32_t* d_es_ntuples, uint32_t n_reads,
                              const uint32_t* h_part_bounds, uint32_t n_parts, DnaWalked* W);
cl_status cl_qual_evolve_ahead(cl_ctx* ctx, cl_qual_coder* Q, const cl_reads* R, const uint8_t* d_quals, const uint64_t* d_qual_off, const uint32_t* h_part_bounds, uint32_t n_parts, QualPrepared* P);
void cl_qual_set_before_tail(cl_qual_coder* Q, std::function<cl_status()> fn);
constexpr cl_status CL_HOOK_RETRY = (cl_status)1000;     // a before_tail hook: "nothing to do yet" — asked again while the batch's interval coders run
