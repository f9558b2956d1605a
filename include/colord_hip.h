/* colord_hip.h — C ABI of the MI355X-native CoLoRd hot path (libcolord_hip.so).
 *
 * The reference (refresh-bio/colord v1.2.1) has no FFI layer: its seams are the C++ stage objects that
 * `runCompression` (src/colord/compression.cpp:344-785) wires together.  Every entry point below names
 * the reference seam it replaces so that a host driver shaped like `runCompression` can call them 1:1.
 * See INTEGRATION.md for the binding a reference maintainer would add.
 *
 * Conventions
 *  - plain C, no C++/torch types; every function returns cl_status (0 = ok, <0 = error, text via
 *    cl_last_error); nothing throws, nothing calls exit().
 *  - pointers named d_* are DEVICE pointers (hipMalloc / torch CUDA tensor data_ptr), h_* are host.
 *  - one cl_ctx per GPU; calls on different contexts are thread-safe, calls on one context are
 *    serialised by the caller.  All kernels of a context run on the context's own HIP stream; a call
 *    returns after its results are complete (it synchronises that stream).
 *  - objects (cl_reads, cl_kmer_set, ...) are owned by the context and freed by their *_free or by
 *    cl_ctx_destroy.
 *  - bases are 2-bit codes A=0 C=1 G=2 T=3 (N=4 in the 1-byte input form), k-mers are uint64 with the
 *    first base in the most significant used bits (src/colord/in_reads.h:30-74), k <= 28
 *    (src/colord/arg_parse.cpp:471).
 */
#ifndef COLORD_HIP_H
#define COLORD_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t cl_status;
enum {
	CL_OK = 0,
	CL_E_INVALID = -1,      /* bad argument */
	CL_E_HIP = -2,          /* HIP runtime error (text in cl_last_error) */
	CL_E_CAPACITY = -3,     /* caller buffer too small; the needed element count was returned */
	CL_E_NOMEM = -4,
	CL_E_UNSUPPORTED = -5
};

typedef struct cl_ctx cl_ctx;
typedef struct cl_reads cl_reads;         /* packed read arena on device */
typedef struct cl_kmer_set cl_kmer_set;   /* filtered k-mer set (CKmerFilter) */
typedef struct cl_kmer_lists cl_kmer_lists; /* per-read accepted k-mers (getAcceptedKmers) */
typedef struct cl_index cl_index;         /* k-mer -> reference reads (CKmersToReads) */

/* ---- context ---------------------------------------------------------------------------------- */
cl_status cl_ctx_create(int device, cl_ctx** out);
void cl_ctx_destroy(cl_ctx* ctx);
const char* cl_last_error(const cl_ctx* ctx);
/* HIP stream the context launches on (hipStream_t as void*), for callers that time with HIP events. */
void* cl_ctx_stream(cl_ctx* ctx);
/* Wall-clock free device timing of the LAST call's dominant kernel, measured with HIP events on the
 * context stream: *ms = elapsed milliseconds, *launches = number of launches it covers. */
cl_status cl_ctx_last_kernel_ms(const cl_ctx* ctx, const char* kernel, double* ms, uint32_t* launches);
/* Kernel times recorded since the previous report, as text lines "name\tms\tlaunches\n" (needs timing on);
 * a successful report clears the record. */
cl_status cl_ctx_kernel_times(cl_ctx* ctx, char* buf, uint64_t cap, uint64_t* needed);
/* Enable/disable per-kernel HIP-event timing (off by default; on adds event records around kernels). */
void cl_ctx_set_timing(cl_ctx* ctx, int on);

/* ---- read arena: replaces read_t / read_pack_t (src/colord/utils.h:366-376, in_reads.cpp:24-42) -- */
/* d_codes: concatenated bases, 1 byte per base; either codes 0..4 (ascii=0) or ASCII ACGTN, upper
 * case only as the reference (ascii=1; anything else: CL_E_INVALID "Only ACGTN symbols supported inside a read").  d_offsets: n_reads+1 base offsets into d_codes.  Builds the 2-bit arena: every read starts
 * on a 32-base (uint64) word boundary and is followed by at least one pad base; a parallel bit mask
 * marks N and pad bases invalid. */
cl_status cl_reads_pack(cl_ctx* ctx, const uint8_t* d_codes, const uint64_t* d_offsets, uint32_t n_reads,
                        int ascii, cl_reads** out);
void cl_reads_free(cl_reads* r);
uint32_t cl_reads_count(const cl_reads* r);
uint64_t cl_reads_total_bases(const cl_reads* r);
uint64_t cl_reads_total_words(const cl_reads* r);
const uint64_t* cl_reads_packed(const cl_reads* r);     /* device: words, base j of a word in bits 63-2j..62-2j */
const uint32_t* cl_reads_invalid(const cl_reads* r);    /* device: per word, bit 31-j set = base j is N/pad */
const uint64_t* cl_reads_word_offsets(const cl_reads* r); /* device: n_reads+1 */
const uint32_t* cl_reads_lengths(const cl_reads* r);    /* device: n_reads */
const uint8_t* cl_reads_has_n(const cl_reads* r);       /* device: n_reads (1 = read contains N) */

/* ---- a1: CKmerWalker + CHashModuloFilter (in_reads.h:30-74, filtering-KMC/hash_filter.h:8-77) ---- */
/* Pass-1 scan: every N-free k-window of every read, canonical form, kept iff fmix64(kmer) % f == 0.
 * Survivors are written unordered to d_out (capacity cap, in k-mers); *n_out = number produced
 * (CL_E_CAPACITY if > cap; then only the first cap were written). */
cl_status cl_kmer_scan(cl_ctx* ctx, const cl_reads* reads, uint32_t k, uint32_t f,
                       uint64_t* d_out, uint64_t cap, uint64_t* n_out);

/* ---- a2 + a3: run_filtering_kmc + CKmerFilter (filtering_kmc.h:5-18, count_kmers.cpp:57-93,
 *      kmer_filter.h:30-167, filter_kmers.cpp:45-93) --------------------------------------------- */
typedef struct {
	uint64_t n_reads;              /* "#Total_reads" (caller-supplied read count is echoed) */
	uint64_t tot_kmers;            /* "#Total no. of k-mers" */
	uint64_t n_unique;             /* "#Unique_k-mers" */
	uint64_t n_unique_counted;     /* "#Unique_counted_k-mers" */
	uint64_t total_count_filtered; /* CKmerFilter::GetTotalKmers() */
} cl_kmer_stats;
/* d_kmers (n survivors of cl_kmer_scan, possibly from several arenas) is sorted in place; distinct
 * k-mers with multiplicity >= ci are kept with counter min(count, cs).  Builds the membership table.
 * Any n: above 2^30 k-mers the input is counted key range by key range (d_kmers is then left unsorted). */
cl_status cl_kmer_count_filter(cl_ctx* ctx, uint64_t* d_kmers, uint64_t n, uint32_t k, uint32_t ci, uint32_t cs,
                               cl_kmer_set** out, cl_kmer_stats* stats);
/* Set object from keys that are already counted/filtered (ascending, distinct) — used to replicate the
 * set on every GPU after the all-gather of the per-rank key partitions (SURVEY §8e exchange 1). */
cl_status cl_kmer_set_create(cl_ctx* ctx, const uint64_t* d_keys, const uint32_t* d_counts, uint64_t n, uint32_t k, cl_kmer_set** out);
void cl_kmer_set_free(cl_kmer_set* s);
uint64_t cl_kmer_set_size(const cl_kmer_set* s);
const uint64_t* cl_kmer_set_keys(const cl_kmer_set* s);   /* device, ascending */
const uint32_t* cl_kmer_set_counts(const cl_kmer_set* s); /* device, saturated counters */
/* CKmerFilter::Check for a batch of k-mers: d_found[i] = 1/0 */
cl_status cl_kmer_set_check(cl_ctx* ctx, const cl_kmer_set* s, const uint64_t* d_kmers, uint64_t n, uint8_t* d_found);

/* ---- a4: CReadsSimilarityGraph::getAcceptedKmers (reads_sim_graph.cpp:128-169, 221-228) ---------- */
/* Per read without N and with len >= k: the distinct canonical k-mers that pass the modulo test and
 * are in the set, in first-occurrence order. */
cl_status cl_accepted_kmers(cl_ctx* ctx, const cl_kmer_set* set, const cl_reads* reads, uint32_t k, uint32_t f,
                            cl_kmer_lists** out);
void cl_kmer_lists_free(cl_kmer_lists* l);
uint32_t cl_kmer_lists_reads(const cl_kmer_lists* l);
uint64_t cl_kmer_lists_total(const cl_kmer_lists* l);
const uint64_t* cl_kmer_lists_offsets(const cl_kmer_lists* l); /* device, n_reads+1 */
const uint64_t* cl_kmer_lists_kmers(const cl_kmer_lists* l);   /* device, k-mer values */
const uint32_t* cl_kmer_lists_ids(const cl_kmer_lists* l);     /* device, rank of each k-mer in the set */
const uint32_t* cl_kmer_lists_pos(const cl_kmer_lists* l);     /* device, start position in the read */

/* ---- a6: CRefReadsAccepter (ref_reads_accepter.h:23-58) ------------------------------------------ */
/* Host function (one sequential mt19937 stream, negligible cost).  h_out[i], i < n_pseudo + n_reads. */
cl_status cl_ref_accept(uint32_t n_reads, uint32_t n_pseudo, uint32_t range, double exponent, uint8_t* h_out);

/* ---- a5: CKmersToReads + processReadsPack[HiFi] (reads_sim_graph.h:45-119, .cpp:295-528) --------- */
/* d_accept[i] (n_reads bytes): read i becomes a reference read (acceptor decision AND no N).  The first
 * n_pseudo reads of `lists` are reference-genome pseudo reads (always accepted, list cap not applied).
 * Builds, for every k-mer of the set, the list of the first max_kmer_count reference ids containing it. */
cl_status cl_index_build(cl_ctx* ctx, const cl_kmer_set* set, const cl_kmer_lists* lists, const uint8_t* d_accept,
                         uint32_t n_pseudo, uint32_t max_kmer_count, cl_index** out);
/* Multi-GPU form of the same build (SURVEY §8e exchange 2).  cl_index_entries_of lists the (k-mer id,
 * reference id) pairs of this rank's accepted reads in read order, reference ids starting at ref_base;
 * d_bounds (n_reads+1, optional) receives ref_base + number of accepted reads before each local read.
 * After an all-gather in rank order the concatenated pairs (ascending reference id) go to
 * cl_index_build_pairs, which sorts d_ids/d_refs in place. */
cl_status cl_index_entries_of(cl_ctx* ctx, const cl_kmer_lists* lists, const uint8_t* d_accept, uint32_t ref_base,
                              uint32_t* d_ids, uint32_t* d_refs, uint64_t cap, uint64_t* n_out,
                              uint32_t* d_bounds, uint32_t* n_accepted);
cl_status cl_index_build_pairs(cl_ctx* ctx, const cl_kmer_set* set, uint32_t* d_ids, uint32_t* d_refs, uint64_t n,
                               const uint32_t* d_bounds, uint32_t n_reads, uint32_t n_refs_total,
                               uint32_t n_pseudo, uint32_t max_kmer_count, cl_index** out);
void cl_index_free(cl_index* ix);
uint32_t cl_index_n_refs(const cl_index* ix);
uint64_t cl_index_entries(const cl_index* ix);
const uint32_t* cl_index_ref_rank(const cl_index* ix);   /* device, n_reads+1: #references before read i */
/* For every read i: up to max_candidates earlier reference reads sharing most accepted k-mers, ordered
 * by (votes desc, ref id asc).  d_refs / d_votes: n_reads * max_candidates (row-major, unused = ~0u / 0),
 * d_n: n_reads. */
cl_status cl_candidates(cl_ctx* ctx, const cl_index* ix, const cl_kmer_lists* lists, uint32_t max_candidates,
                        uint32_t* d_refs, uint32_t* d_votes, uint32_t* d_n);
/* The same query for one CHUNK of reads against an index built over the reference reads of the WHOLE input
 * (cl_index_build_pairs with n_reads = 0): d_bounds[i] = number of reference reads that precede read i in file order. */
cl_status cl_candidates_at(cl_ctx* ctx, const cl_index* ix, const cl_kmer_lists* lists, const uint32_t* d_bounds, uint32_t max_candidates,
                           uint32_t* d_refs, uint32_t* d_votes, uint32_t* d_n);
/* HiFi variant (processReadsPackHiFi): additionally, per chosen candidate, the shared k-mers in read
 * order.  d_common_off: n_reads*max_candidates+1 offsets into d_common (capacity cap k-mers);
 * *n_common = needed size. */
cl_status cl_candidates_common(cl_ctx* ctx, const cl_index* ix, const cl_kmer_lists* lists, uint32_t max_candidates,
                               const uint32_t* d_refs, const uint32_t* d_n,
                               uint64_t* d_common_off, uint64_t* d_common, uint64_t cap, uint64_t* n_common);

/* ---- a13 + a15 + a16: CQualityCoder / CEntrComprQuals (quality_coder.{h,cpp}, quality_coder_impl.cpp,
 *      entr_qual.h:100-135; range coder sub_rc.h:44-212; models rc.h) ---------------------------------- */
typedef struct cl_qual_coder cl_qual_coder;
typedef struct {
	int32_t mode;        /* QualityComprMode (params.h:33-43): 0 org, 1 5-avg, 2 4-avg, 3 2-avg, 4 5-fix, 5 4-fix, 6 2-fix, 7 avg, 8 none */
	int32_t source;      /* DataSource: 0 ONT, 1 PBRaw, 2 PBHiFi */
	int32_t level;       /* compression level 1..3 */
	uint32_t n_fwd;      /* -T thresholds (bins - 1 values) */
	uint32_t fwd[8];
	uint32_t n_rev;      /* -D representatives (decoder side only; carried for completeness) */
	uint32_t rev[8];
} cl_qual_params;
/* CQualityCoder::Init(true, ...): one adaptive model set that persists across cl_qual_encode calls. */
cl_status cl_qual_coder_create(cl_ctx* ctx, const cl_qual_params* params, cl_qual_coder** out);
void cl_qual_coder_free(cl_qual_coder* q);
cl_ctx* cl_qual_coder_ctx(const cl_qual_coder* q);          /* the context the coder was created on */
/* CEntrComprQuals::Compress for a batch of whole parts.  Reads [h_part_bounds[p], h_part_bounds[p+1]) of the
 * arena form part p (the reference cuts parts where sum(len+1) >= 4 Mi, in_reads.cpp:62-77; any cut at read
 * boundaries yields a valid archive).  d_quals: ASCII quality bytes, d_qual_off: n_reads+1 offsets into it
 * (read i has the same length as in the arena).  d_flags (level > 1 only, else NULL): per-base class bytes
 * 'A' / 'M' / other (quality_coder_impl.cpp:25-75), same offsets.  Output: part payloads back to back in
 * d_out (capacity cap), h_part_sizes[n_parts]; *n_out = total bytes. */
cl_status cl_qual_encode(cl_ctx* ctx, cl_qual_coder* q, const cl_reads* reads, const uint8_t* d_quals, const uint64_t* d_qual_off,
                         const uint8_t* d_flags, const uint32_t* h_part_bounds, uint32_t n_parts,
                         uint8_t* d_out, uint64_t cap, uint64_t* h_part_sizes, uint64_t* n_out);

/* ---- a8: m-mer anchors (encoder.cpp:291-493,617-776,1016-1111,1149-1192,1577-1622) ------------------------ */
typedef struct cl_anchors cl_anchors;
/* prepareEncodeCandidates + fixOverlaping* for every read of `reads` (non-HiFi path): for each of its <= c candidate
 * reference ids (d_cand_refs row-major n_reads*c, d_cand_n per read — the output of cl_candidates; ids index `refs`,
 * the arena of reference reads) both orientations are analysed: shared m-mers (m = anchor_len, not canonical), "too
 * many matches" veto, LIS chain, merge into anchors; the better orientation is kept (reverse complement wins ties),
 * candidates are sorted by total anchor length (stable) and overlaps trimmed.  Reads with N, shorter than m or with
 * too few distinct m-mers get no candidates (they are stored plain). */
cl_status cl_anchor_candidates(cl_ctx* ctx, const cl_reads* reads, const cl_reads* refs, const uint32_t* d_cand_refs, const uint32_t* d_cand_n,
                               uint32_t c, uint32_t anchor_len, double frac_always, double frac_min, double max_matches_mult,
                               uint32_t min_anchors, cl_anchors** out);
/* a9, DataSource::PBHiFi (prepareEncodeCandidatesHiFi, KmerBasedAnchors, AnalyseRefReadWithKmers; encoder.cpp:870-1013,
 * 1113-1147,1194-1253): as above, but a candidate is first anchored on the k-mers it shares with the read
 * (d_common_off / d_common from cl_candidates_common; kmer_len, modulo = the graph's k and f): k-mers unique in both
 * reads seed anchors of length k, which must be colinear, are extended base by base and merged; the m-mer analysis
 * decides only the candidates without accepted k-mer anchors. */
cl_status cl_anchor_candidates_hifi(cl_ctx* ctx, const cl_reads* reads, const cl_reads* refs, const uint32_t* d_cand_refs, const uint32_t* d_cand_n,
                                    uint32_t c, uint32_t anchor_len, double frac_always, double frac_min, double max_matches_mult,
                                    uint32_t min_anchors, uint32_t kmer_len, uint32_t modulo, const uint64_t* d_common_off, const uint64_t* d_common,
                                    cl_anchors** out);
void cl_anchors_free(cl_anchors* a);
uint64_t cl_anchors_total(const cl_anchors* a);
const uint32_t* cl_anchors_n_cands(const cl_anchors* a);        /* device, n_reads */
const uint32_t* cl_anchors_cands(const cl_anchors* a);          /* device, n_reads*c*4: ref_id, rev, tot_anchor_len, n_anchors */
const uint64_t* cl_anchors_cand_offsets(const cl_anchors* a);   /* device, n_reads*c+1: first anchor of each slot */
const uint32_t* cl_anchors_data(const cl_anchors* a);           /* device, 3 per anchor: len, pos_in_enc, pos_in_ref */

/* ---- a12 (plain forms): CEncoder::AddPlainRead / AddPlainReadWithN (encoder.cpp:663-681) --------------- */
/* Tuple streams that store every read of the arena verbatim: `start_plain` (`start_plain_with_Ns` for reads
 * containing N) + one `plain` tuple per base.  d_es needs total_bases + n_reads bytes (cap), d_es_off
 * n_reads+1, d_es_ntuples n_reads.  This is what the reference emits for a read without usable candidates. */
cl_status cl_encode_plain(cl_ctx* ctx, const cl_reads* reads, uint8_t* d_es, uint64_t cap, uint64_t* d_es_off,
                          uint32_t* d_es_ntuples, uint64_t* n_out);

/* ---- a10 + a11 + a12: CEncoder::Encode (encoder.cpp:1672-1691) on anchored candidates ------------------- */
/* Every read of `reads` is encoded against its candidates from cl_anchor_candidates (same arena, same c):
 * gaps between anchors are aligned with edlib's observable behaviour (edit_script.h:156-419), indels canonicalised
 * (refactor_edit_script), each gap kept as edit script or replaced by literals / an alternative reference
 * (EncodePart, EncodeWithAlternativeRead up to max_rec levels), and the result written as es_t tuple bytes
 * (utils.h:56-273).  h_pack_bounds (n_packs+1 read indices, first 0, last n_reads) are the reader packs: the
 * adaptive cost estimator (CEntropyEstimator, utils.h:877-1130) is reset at each of them (encoder.cpp:1677).
 * Reads without candidates and reads with N are stored plain.  d_es (cap bytes) receives the streams back to back,
 * d_es_off n_reads+1 byte offsets, d_es_ntuples n_reads tuple counts; *n_out = bytes needed (CL_E_CAPACITY if > cap). */
cl_status cl_encode_reads(cl_ctx* ctx, const cl_reads* reads, const cl_reads* refs, const cl_anchors* anchors, uint32_t c, uint32_t anchor_len,
                          uint32_t min_part_alt, uint32_t max_rec, double cost_mult, const uint32_t* h_pack_bounds, uint32_t n_packs,
                          uint8_t* d_es, uint64_t cap, uint64_t* d_es_off, uint32_t* d_es_ntuples, uint64_t* n_out);

/* The logarithm every cost decision of the encoder is made of, evaluated ON THE DEVICE exactly as the decision kernels do:
 * d_out[i] = d_count[i] ? -log2((double)d_count[i] * (1.0 / d_total[i])) : 0.0  (calc_logs, utils.h:800-810; the entropy of
 * CEntropy, utils.h:706-757, takes log2 of the same products).  The sums built from these values are plain IEEE additions
 * and multiplications (the library is compiled without FMA contraction), so pinning this function against the host's libm
 * over the reachable (count, total) pairs pins the decisions (tests/test_gpu_floatpin.py). */
cl_status cl_estimator_logs(cl_ctx* ctx, const uint32_t* d_count, const uint32_t* d_total, uint64_t n, double* d_out);

/* The per-base classes the quality coder uses at levels 2 and 3 (analyze_es, quality_coder_impl.cpp:25-75), from the
 * reads' own tuple streams: 'P' plain read, 'A' base inside an anchor tuple, 'M' unit match, ' ' inserted or substituted.
 * d_base_off: n_reads+1 offsets of the reads' bases (the d_qual_off of cl_qual_encode); d_flags: total_bases bytes. */
cl_status cl_es_flags(cl_ctx* ctx, const cl_reads* reads, const uint8_t* d_es, const uint64_t* d_es_off, const uint64_t* d_base_off, uint8_t* d_flags);

/* ---- a14 + a16: CDNACoder / CEntrComprReads (dna_coder.{h,cpp}, entr_read.h:56-80) --------------------- */
typedef struct cl_dna_coder cl_dna_coder;
/* CDNACoder::Init(true, maxCandidates, level, ., n_ref_genome_pseudo_reads): one adaptive model set that
 * persists across cl_dna_encode calls; start_read_id seeds cur_read_id. */
cl_status cl_dna_coder_create(cl_ctx* ctx, uint32_t max_alt_refs, int32_t level, uint32_t start_read_id, cl_dna_coder** out);
void cl_dna_coder_free(cl_dna_coder* d);
/* CEntrComprReads::Compress for a batch of whole parts.  d_es: the reads' tuple streams (es_t byte layout,
 * utils.h:56-273) back to back, d_es_off: n_reads+1 byte offsets, d_es_ntuples: es_t::size() per read.
 * refs: arena whose read i is reference read i (CReferenceReads order).  Parts [h_part_bounds[p],
 * h_part_bounds[p+1]) must cover [0, n_reads).  Output as for cl_qual_encode; the archive metadata of
 * part p is its read count. */
cl_status cl_dna_encode(cl_ctx* ctx, cl_dna_coder* d, const cl_reads* refs, const uint8_t* d_es, const uint64_t* d_es_off,
                        const uint32_t* d_es_ntuples, uint32_t n_reads, const uint32_t* h_part_bounds, uint32_t n_parts,
                        uint8_t* d_out, uint64_t cap, uint64_t* h_part_sizes, uint64_t* n_out);

/* CReferenceReads (reference_reads.h:24-80): the arena of the reads with d_keep[i] != 0, in order — reference id r is
 * the r-th kept read.  The result is an independent arena (free with cl_reads_free). */
cl_status cl_reads_select(cl_ctx* ctx, const cl_reads* src, const uint8_t* d_keep, cl_reads** out);
/* An arena from words that already have the arena layout (word-aligned reads back to back; see cl_reads_packed /
 * cl_reads_invalid), e.g. the concatenation of the reference-read arenas of all ranks (the reference's
 * CReferenceReads is one process-wide store; with reads sharded over GPUs each rank replicates it). */
cl_status cl_reads_from_arena(cl_ctx* ctx, const uint64_t* d_packed, const uint32_t* d_inv, const uint32_t* d_lens, uint32_t n_reads, cl_reads** out);

/* ---- the `header` stream: CIDCoder + CEntrComprHeaders (id_coder.{h,cpp}, entr_header.cpp:23-45) — HOST functions --- */
typedef struct cl_id_coder cl_id_coder;
/* header_mode = HeaderComprMode (params.h:44): 0 Original (lossless), 1 Main, 2 None (the latter two code no id bytes,
 * as in the reference).  One coder per archive: models and the previous id persist from part to part. */
cl_status cl_id_coder_create(int32_t header_mode, cl_id_coder** out);
void cl_id_coder_free(cl_id_coder* c);
const char* cl_id_coder_error(const cl_id_coder* c);
/* One pack of headers -> one part (archive metadata = n).  h_ids: the ids back to back without the leading '@' / '>',
 * h_off: n+1 offsets, h_plus[i] != 0 when the '+' line of record i repeats the id.  The reference closes a pack when the
 * id bytes reach 4 Mi (in_reads.cpp:50-56,93-101).  cap >= 2 * id bytes + 64. */
cl_status cl_id_encode_part(cl_id_coder* c, const uint8_t* h_ids, const uint64_t* h_off, const uint8_t* h_plus, uint32_t n,
                            uint8_t* h_out, uint64_t cap, uint64_t* n_out);

/* ---- the whole compress data path of one shard: runCompression's stage wiring (compression.cpp:432-689) ---------- */
typedef struct {
	uint32_t k, f, ci, cs, c;                 /* kmerLen, filterHashModulo, minKmerCount, maxKmerCount, maxCandidates */
	uint32_t anchor_len, min_part_alt, max_rec, min_anchors;   /* anchorLen, minPartLenToConsiderAltRead, maxRecurence, minAnchors */
	int32_t level, source;                    /* compression level 1..3, DataSource (0 ONT, 1 PBRaw, 2 PBHiFi) */
	int32_t sparse;                           /* referenceReadsMode == Sparse (else every N-free read is a reference read) */
	double sparse_g, sparse_exponent;         /* sparseMode_range_symbols (in genome lengths), sparseMode_exponent */
	double cost_mult, frac_always, frac_min, max_matches_mult;   /* editScriptCostMultiplier, minFractionOfMmersInEncode*, maxMatchesMultiplier */
} cl_compress_params;
typedef struct {
	uint64_t n_reads, n_bases, tot_kmers, n_kept_kmers, n_refs, n_anchors, tuple_bytes, dna_bytes, qual_bytes;
	uint32_t sparse_range, pad;
} cl_compress_info;
/* reads (+ ASCII qualities d_quals with per-read offsets d_base_off, or NULLs with qual == NULL) -> `dna` and `qual` stream
 * parts.  h_part_bounds: the parts of both streams (read indices); h_pack_bounds: the reader packs (estimator reset).
 * dna / qual: the long-lived coders (model state persists across calls, as one CEntrCompr* thread).  Outputs as for
 * cl_dna_encode / cl_qual_encode.  If `qual` was created on a second context of the same GPU and level == 1 (no
 * dependence on the edit scripts) the quality stream is coded concurrently with the DNA path.  Single GPU; with reads sharded over GPUs the caller runs the stages itself around the
 * two exchanges (bench.py). */
cl_status cl_compress_shard(cl_ctx* ctx, const cl_compress_params* params, const cl_reads* reads, const uint8_t* d_quals, const uint64_t* d_base_off,
                            const uint32_t* h_part_bounds, uint32_t n_parts, const uint32_t* h_pack_bounds, uint32_t n_packs,
                            cl_dna_coder* dna, cl_qual_coder* qual,
                            uint8_t* d_dna_out, uint64_t dna_cap, uint64_t* h_dna_part_sizes,
                            uint8_t* d_qual_out, uint64_t qual_cap, uint64_t* h_qual_part_sizes, cl_compress_info* info);

/* ---- runCompression over an input of any size, chunk by chunk, on one GPU or on one GPU per rank --------------------
 * The reference streams the file twice: pass 1 counts the k-mers of the whole input (compression.cpp:432), pass 2 pushes
 * reader packs through graph -> encoder -> coders (compression.cpp:547-561, in_reads.cpp:62-77) while the similarity
 * graph grows (reads_sim_graph.cpp:324-427).  cl_compressor holds that state between calls.  A chunk is an arena of whole
 * reader packs (any size the GPU holds, e.g. 1 Gbase); the caller presents the same chunks, in file order, three times:
 *     cl_compressor_count_add (each chunk) -> cl_compressor_count_finish
 *     cl_compressor_refs_add  (each chunk) -> cl_compressor_refs_finish
 *     cl_compressor_encode    (each chunk)
 * The `dna` / `qual` parts are byte-identical to one cl_compress_shard call over the whole input.
 *
 * cl_exchange: with reads sharded over GPUs (one process per GPU, rank r holds the r-th contiguous range of the file)
 * the two *_finish steps exchange what SURVEY.md section 8e lists: k-mers go to the rank owning their key range and the
 * kept keys are all-gathered (replicated set); reference reads and index entries are all-gathered (replicated store and
 * index).  The caller supplies the collectives (colord_amd/parallel.py: torch.distributed, "nccl" = RCCL over xGMI);
 * every rank must call the compressor functions in the same order.  Each rank is its own model domain of the coders. */
typedef struct cl_compressor cl_compressor;
typedef struct {
	void* user;
	uint32_t rank, world;
	/* host: every rank contributes n uint64; h_out[world * n] receives them in rank order */
	cl_status (*all_gather_host)(void* user, const uint64_t* h_vals, uint32_t n, uint64_t* h_out);
	/* device: d_send holds the bytes for rank 0, 1, ... back to back (h_send_bytes[world]); d_recv receives the bytes
	 * from rank 0, 1, ... back to back (h_recv_bytes[world]) */
	cl_status (*all_to_all_v)(void* user, const void* d_send, const uint64_t* h_send_bytes, void* d_recv, const uint64_t* h_recv_bytes);
	/* device: d_recv = the send buffers of rank 0, 1, ... back to back; h_recv_bytes[world] (send_bytes == h_recv_bytes[rank]) */
	cl_status (*all_gather_v)(void* user, const void* d_send, uint64_t send_bytes, void* d_recv, const uint64_t* h_recv_bytes);
} cl_exchange;
/* qual_ctx: second context of the same GPU for the quality stream (coded concurrently at level 1), or NULL.  qparams NULL =
 * no quality stream.  exchange NULL = single GPU.  expected_bases: size hint for the k-mer buffer of pass 1 (0 = unknown). */
cl_status cl_compressor_create(cl_ctx* ctx, cl_ctx* qual_ctx, const cl_compress_params* params, const cl_qual_params* qparams,
                               const cl_exchange* exchange, uint64_t expected_bases, cl_compressor** out);
void cl_compressor_free(cl_compressor* c);
cl_status cl_compressor_count_add(cl_compressor* c, const cl_reads* chunk);
/* stats (optional): the statistics over the whole input (all ranks) */
cl_status cl_compressor_count_finish(cl_compressor* c, cl_kmer_stats* stats);
cl_status cl_compressor_refs_add(cl_compressor* c, const cl_reads* chunk);
cl_status cl_compressor_refs_finish(cl_compressor* c);
/* arguments as for cl_compress_shard, bounds relative to the chunk */
cl_status cl_compressor_encode(cl_compressor* c, const cl_reads* chunk, const uint8_t* d_quals, const uint64_t* d_base_off,
                               const uint32_t* h_part_bounds, uint32_t n_parts, const uint32_t* h_pack_bounds, uint32_t n_packs,
                               uint8_t* d_dna_out, uint64_t dna_cap, uint64_t* h_dna_part_sizes,
                               uint8_t* d_qual_out, uint64_t qual_cap, uint64_t* h_qual_part_sizes, cl_compress_info* info);
/* Reference-genome mode (`-G`; compression.cpp:405-447, reference_genome.cpp:372-429, reads_sim_graph.cpp:295-322).  With reads
 * sharded over GPUs every rank makes both calls with the SAME sequences / pseudo reads: rank 0 counts the genome's k-mers and
 * contributes the pseudo reads to the replicated reference store, the other ranks only take note of their number:
 *   cl_compressor_genome_add    before count_finish: the genome's sequences (arenas of ACGT codes, any number of calls) are a second
 *                               input of the k-mer counter; the statistics of count_finish are corrected for them as in the reference;
 *   cl_compressor_pseudo_reads  once, after count_finish and before the first refs_add: the overlapping pieces of the sequences
 *                               (length 20 x mean_read_len of cl_compressor_info, overlap 10 x (k - 1)) become reference reads
 *                               0 .. n-1 — always accepted, their k-mer lists uncapped; the acceptor and the DNA coder count them. */
cl_status cl_compressor_genome_add(cl_compressor* c, const cl_reads* sequences);
cl_status cl_compressor_pseudo_reads(cl_compressor* c, const cl_reads* pseudo_reads);
/* Optional look-ahead of pass 2.  Announces a chunk that a LATER cl_compressor_encode call will present (announce in file
 * order, before the chunk is encoded; the arena and the bounds' meaning stay as they are until that call returns).  What the
 * chunk needs before the coders — a4 accepted k-mers, a5 candidates, a8/a9 anchors, a10-a12 edit scripts and tuple streams —
 * depends on no earlier chunk's output (the reference's graph and encoder threads run ahead of its two coder threads in the
 * same way, compression.cpp:547-661), so it is computed in the background on a context of the compressor's own ("encode
 * lane", COLORD_HIP_ENCODE_LANES of them, default 2, each with its own HIP streams and memory pool, about 30 GB per 1-Gbase chunk) while the caller's
 * thread codes the chunks before it; only the adaptive models of the `dna` / `qual` coders chain chunk to chunk.  The lanes
 * run at most lanes + 2 chunks ahead of the encode calls.  Output bytes are the same with or without announcements. */
cl_status cl_compressor_prepare(cl_compressor* c, const cl_reads* chunk, const uint32_t* h_pack_bounds, uint32_t n_packs);
/* The same with the coder parts the chunk will be encoded with (h_part_bounds of the later cl_compressor_encode call): then the half
 * of the `dna` coder that needs no model state — tuple walks, triple slots, the stable sort by (family, context), context runs;
 * the reference's CEntrComprReads thread does all of that inline, entr_read.h:56-80 — is made ahead as well, on a preparation
 * thread and context of the compressor's own, one or two chunks ahead of the encode calls (~15 GB per 1-Gbase chunk ahead); the
 * caller's stream keeps model evolution and interval coding.  Announcements must start with the first chunk.  With d_quals /
 * d_base_off (those of the later encode call; null: not prepared) the same for the `qual` coder at level 1: symbols, sort by
 * context and context runs on a second preparation thread (~8 GB per 1-Gbase chunk ahead).  Bytes unchanged.  The announcement is
 * BINDING while the compressor evolves ahead (long parts, mean >= 2^19 symbols, or COLORD_HIP_EVOLVE_DEPTH > 0): the adaptive models
 * of the next chunks are then advanced from the announced bounds and quality buffer before their encode call and cannot be rolled
 * back — an encode call with other part bounds or buffers fails with CL_E_INVALID ("the batch evolved ahead is not the one encoded
 * next") and the compressor is unusable afterwards.  Without evolve-ahead a differing encode call is honoured (the preparation is
 * redone). */
cl_status cl_compressor_prepare_parts(cl_compressor* c, const cl_reads* chunk, const uint32_t* h_pack_bounds, uint32_t n_packs, const uint32_t* h_part_bounds, uint32_t n_parts,
                                      const uint8_t* d_quals, const uint64_t* d_base_off);
/* what the archive's `meta` stream needs (compression.cpp:704-779), valid after count_finish (n_refs_total after refs_finish):
 * first_read = global index of this rank's first read (start of its model domain) */
cl_status cl_compressor_info(const cl_compressor* c, cl_kmer_stats* stats, uint64_t* first_read, uint64_t* n_reads_total, uint64_t* mean_read_len,
                             uint32_t* sparse_range, uint32_t* n_refs_total);

/* ---- a17, the inverse path: CRangeDecoder (sub_rc.h:216-392), CDNACoder::Decode (dna_coder.cpp:234-437), CQualityCoder::Decode
 *      (quality_coder.cpp:605-657, quality_coder_impl.cpp:506-559,800-849), CIDCoder::Decode (id_coder.cpp:396-600); drivers
 *      CEntropyDecomprReads / CEntrDecomprQuals / CEntrDecomprHeaders (entr_read.h:146-191, entr_qual.h:136-260, entr_header.cpp:46-80).
 *      HOST functions (h_* pointers): a model domain decodes as one dependent chain; streams and domains are the parallelism. ---- */
/* The `ref-genome` stream of archives written with -G -s (CReferenceGenome::Store / its archive constructor,
 * reference_genome.cpp:235-279,325-370): every sequence a plain read under the DNA coder's "level 9" models, one part whose
 * archive metadata is the number of sequences.  h_codes: bases 0..3 back to back, h_off: n_seqs + 1 offsets.  HOST functions.
 * CL_E_CAPACITY with *n_out = bytes needed when the output does not fit. */
cl_status cl_genome_encode(const uint8_t* h_codes, const uint64_t* h_off, uint32_t n_seqs, uint8_t* h_out, uint64_t cap, uint64_t* n_out);
cl_status cl_genome_decode(const uint8_t* h_in, uint64_t n_in, uint32_t n_seqs, uint8_t* h_codes, uint64_t cap, uint64_t* h_off, uint64_t* n_out);
/* MD5 of the sequences in the reference's packed form — what the `meta` stream carries instead of the genome without -s
 * (reference_genome.cpp:29-67,205-213; compression.cpp:771-776). */
cl_status cl_genome_md5(const uint8_t* h_codes, const uint64_t* h_off, uint32_t n_seqs, uint8_t* h_md5_16);

typedef struct cl_dna_decoder cl_dna_decoder;
typedef struct cl_qual_decoder cl_qual_decoder;
typedef struct cl_id_decoder cl_id_decoder;
/* CDNACoder::Init(false, ...) + CReferenceReads + CRefReadsAccepter: max_alt_refs / level (1-3; 9 = the models of the stored reference genome) / sparse range + exponent come from the
 * archive's `meta` stream; accept_all = ReferenceReadsMode::All; start_read_id = n_pseudo for single-domain archives. */
cl_status cl_dna_decoder_create(uint32_t max_alt_refs, int32_t level, uint32_t start_read_id, uint32_t n_pseudo,
                                int32_t accept_all, uint32_t sparse_range, double sparse_exponent, cl_dna_decoder** out);
void cl_dna_decoder_free(cl_dna_decoder* d);
const char* cl_dna_decoder_error(const cl_dna_decoder* d);
/* a reference-genome pseudo read (codes 0..3), in order, before the first part (decompression_common.cpp:300-305) */
cl_status cl_dna_decoder_add_ref(cl_dna_decoder* d, const uint8_t* h_codes, uint32_t len);
/* Archives written by several GPUs hold one model domain per rank: fresh adaptive models from here on; the reference reads, the
 * read counter and the acceptor's random stream continue. */
cl_status cl_dna_decoder_new_domain(cl_dna_decoder* d);
/* One `dna` part of n_reads reads (the part's archive metadata) -> bases back to back: codes 0..3, 4 = N, at levels 2 and 3 with
 * the class flags 0x80 (anchor) / 0x40 (match) the quality decoder reads (basic_coder.h:34-35); h_off[n_reads+1].  If cap is too
 * small: CL_E_CAPACITY with *n_out = bytes needed; the decoded part stays inside the decoder and the same call with a buffer of
 * that size fetches it (nothing is decoded twice).  cl_id_decode_part behaves the same way. */
cl_status cl_dna_decode_part(cl_dna_decoder* d, const uint8_t* h_in, uint64_t n_in, uint32_t n_reads,
                             uint8_t* h_bases, uint64_t cap, uint64_t* h_off, uint64_t* n_out);
/* CQualityCoder::Init(false, ...): params as for the encoder; rev[] = the representatives stored in `meta` (-D). */
cl_status cl_qual_decoder_create(const cl_qual_params* params, cl_qual_decoder** out);
void cl_qual_decoder_free(cl_qual_decoder* q);
cl_status cl_qual_decoder_new_domain(cl_qual_decoder* q);
/* One `qual` part: h_bases / h_off are the output of cl_dna_decode_part for the same part; h_quals receives ASCII qualities at
 * the same offsets. */
cl_status cl_qual_decode_part(cl_qual_decoder* q, const uint8_t* h_in, uint64_t n_in, const uint8_t* h_bases, const uint64_t* h_off,
                              uint32_t n_reads, uint8_t* h_quals);
cl_status cl_id_decoder_create(int32_t header_mode, cl_id_decoder** out);
void cl_id_decoder_free(cl_id_decoder* c);
/* One `header` part of n ids -> the ids back to back (without '@' / '>'), h_off[n+1], h_plus[n] (1 = the '+' line repeats the id). */
cl_status cl_id_decode_part(cl_id_decoder* c, const uint8_t* h_in, uint64_t n_in, uint32_t n, uint8_t* h_ids, uint64_t cap,
                            uint64_t* h_off, uint8_t* h_plus, uint64_t* n_out);

/* ---- a7: CReferenceReads (reference_reads.h:27-259) ---------------------------------------------- */
/* Byte image of one stored reference read (4 bases/byte MSB first + trailing count byte) produced from
 * the arena; h_out needs (len+3)/4+1 bytes.  Used by the parity tests and by the host archive code. */
cl_status cl_reads_compact(cl_ctx* ctx, const cl_reads* reads, uint32_t read, uint8_t* h_out, uint64_t cap, uint64_t* n_out);

/* ---- utilities (device primitives exposed for tests/bench) -------------------------------------- */
cl_status cl_sort_u64(cl_ctx* ctx, uint64_t* d_keys, uint64_t n, uint32_t begin_bit, uint32_t end_bit);
cl_status cl_sort_u64_u32(cl_ctx* ctx, uint64_t* d_keys, uint32_t* d_vals, uint64_t n, uint32_t begin_bit, uint32_t end_bit);

#ifdef __cplusplus
}
#endif
#endif
