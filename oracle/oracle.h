/* oracle.h — CPU restatement of the CoLoRd v1.2.1 hot path (TEST INFRASTRUCTURE ONLY).
 *
 * Plain C99 restatement, written from the behaviour of the reference sources; every function cites the
 * reference file:line it follows (paths relative to /root/reference).  Nothing here is shipped or
 * measured as the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle.so.  The product path (colord_amd/) never calls into it.
 *
 * Parity status: PINNED.  Each stage is checked in tests/ against golden vectors dumped from the
 * unmodified reference objects by oracle/ref_harness/ref_dump.cpp (built by oracle/Makefile.ref into
 * oracle/_ref/) on the reference's own fixtures test/M.bovis, A.thaliana, D.melanogaster (+ genome).
 *
 * Conventions: bases are one byte each, A=0 C=1 G=2 T=3 N=4 (in_reads.cpp:24-42); k-mers are uint64,
 * first base in the most significant bits (in_reads.h:59-73).
 */
#ifndef COLORD_ORACLE_H
#define COLORD_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- a1: canonical k-mer walk + murmur-modulo filter ------------------------------------------- */
uint64_t orc_hash_mm(uint64_t x);                                   /* hash_filter.h:8-16 */
/* Pass-1 (KMC) semantics: every N-free window of k bases (splitter.cpp:560-660), canonical form,
 * kept iff hash_mm(kmer) % f == 0 (hash_filter.h:28-77).  Appends to out (capacity cap), returns count
 * (or the count that would be needed when out==NULL). */
size_t orc_kmer_scan(const uint8_t* bases, size_t len, uint32_t k, uint32_t f, uint64_t* out, size_t cap);

/* ---- a2: exact count, threshold, saturation (kb_sorter.h:1000-1060, kmc.h:1428-1485) ------------ */
typedef struct {
	uint64_t n_reads;          /* filled by the caller */
	uint64_t tot_kmers;        /* "#Total no. of k-mers": surviving instances */
	uint64_t n_unique;         /* distinct surviving k-mers */
	uint64_t n_unique_counted; /* "#Unique_counted_k-mers": count >= ci */
	uint64_t total_count_filtered; /* sum of min(count,cs) over kept keys (filter_kmers.cpp:77) */
} orc_kmer_stats;
/* kmers is sorted in place. keys/counts need capacity n. returns number of kept keys (ascending). */
size_t orc_count_filter(uint64_t* kmers, size_t n, uint32_t ci, uint32_t cs,
                        uint64_t* keys, uint32_t* counts, orc_kmer_stats* st);

/* ---- a4: accepted k-mers of one read (reads_sim_graph.cpp:128-169) ------------------------------ */
/* kept = ascending array of kept keys.  hasN reads and reads shorter than k yield nothing.
 * Output: distinct canonical k-mers passing the modulo test and present in kept, first-occurrence order. */
size_t orc_accepted_kmers(const uint8_t* bases, size_t len, uint32_t k, uint32_t f,
                          const uint64_t* kept, size_t n_kept, uint64_t* out, size_t cap);

/* ---- a6: reference-read acceptor (ref_reads_accepter.h:23-58) ----------------------------------- */
/* Replays std::mt19937 (default seed 5489) + std::uniform_real_distribution<double>(0,1) as libstdc++
 * implements them (two 32-bit draws per double).  out[i] for i in [0, n_pseudo + n_reads). */
void orc_ref_accept(uint32_t n_reads, uint32_t n_pseudo, uint32_t range, double exponent, uint8_t* out);

/* ---- a5: streaming k-mer -> reads multimap and candidate selection (reads_sim_graph.cpp:324-528) - */
typedef struct orc_graph orc_graph;
orc_graph* orc_graph_new(uint32_t max_candidates, uint32_t max_kmer_count);
void orc_graph_free(orc_graph*);
/* pseudo-read from the reference genome: always a reference, list cap not applied (:295-322) */
void orc_graph_add_pseudo(orc_graph*, const uint64_t* kmers, size_t n);
/* one real read in file order. accept = !hasN && acceptor decision. out_refs capacity max_candidates.
 * If out_common != NULL (HiFi, :429-528): out_common_off[i]..[i+1] delimit, in out_common, the k-mers
 * shared with out_refs[i] in read order (out_common capacity max_candidates * n). returns #candidates. */
uint32_t orc_graph_next_read(orc_graph*, const uint64_t* kmers, size_t n, int accept,
                             uint32_t* out_refs, uint32_t* out_votes,
                             uint64_t* out_common, uint32_t* out_common_off);
uint32_t orc_graph_n_refs(const orc_graph*);

/* ---- a7: 2-bit reference read store (reference_reads.h:35-72) ----------------------------------- */
size_t orc_refread_compact(const uint8_t* bases, size_t len, uint8_t* out);   /* returns (len+3)/4 + 1 */

/* ---- a13 + a15: range coder, models, quality coder (sub_rc.h, rc.h, quality_coder*.cpp) --------- */
typedef struct orc_qual orc_qual;
/* mode: QualityComprMode (params.h:33-43), source: DataSource (0 ONT, 1 PBRaw, 2 PBHiFi), level 1..3 */
orc_qual* orc_qual_new(int compress, int mode, int source, int level, const uint32_t* fwd_thr, int n_fwd, const uint32_t* rev_thr, int n_rev);
void orc_qual_free(orc_qual*);
void orc_qual_encode(orc_qual*, const uint8_t* bases, const uint8_t* qual, uint32_t len, const uint8_t* flags);
/* closes the current part (Finish/GetOutput/Restart, entr_qual.h:68-79); dst==NULL only returns the size */
size_t orc_qual_finish_part(orc_qual*, uint8_t* dst, size_t cap);
void orc_qual_set_input(orc_qual*, const uint8_t* data, size_t n);
void orc_qual_decode(orc_qual*, const uint8_t* bases, uint32_t len, const uint8_t* flags, uint8_t* qual_out);
/* per-base 'A'/'M'/' '/'P' classes from a read's tuple stream (quality_coder_impl.cpp:25-75) */
void orc_es_flags(const uint8_t* es, size_t n, uint32_t read_len, uint8_t* flags);

/* ---- a14: DNA coder (dna_coder.{h,cpp}, entr_read.h:56-80) ---------------------------------------- */
typedef struct orc_dna orc_dna;
orc_dna* orc_dna_new(int compress, int max_alt_refs, int level, uint32_t start_read_id);
void orc_dna_free(orc_dna*);
void orc_dna_add_ref(orc_dna*, const uint8_t* bases, uint32_t len);      /* CReferenceReads::Add, in reference-id order */
/* one read: its tuple stream in the App. A byte layout and the tuple count (es_t::size()) */
void orc_dna_encode(orc_dna*, const uint8_t* es, size_t n_bytes, uint32_t n_tuples);
size_t orc_dna_finish_part(orc_dna*, uint8_t* dst, size_t cap);

/* ---- a8-a12: CEncoder (encoder.{h,cpp}, edit_script.h, libs/edlib, utils.h:700-1131) ---------------- */
typedef struct orc_encoder orc_encoder;
orc_encoder* orc_encoder_new(uint32_t anchor_len, uint32_t kmer_len, uint32_t modulo, int source, double frac_always, double frac_min,
                             double max_matches_mult, double cost_mult, uint32_t min_part_alt, uint32_t max_rec, uint32_t min_anchors);
void orc_encoder_free(orc_encoder*);
void orc_encoder_add_ref(orc_encoder*, const uint8_t* bases, uint32_t len);     /* CReferenceReads::Add */
void orc_encoder_new_pack(orc_encoder*);                                          /* entropyEstimator.Reset() (encoder.cpp:1677) */
/* processComprElem for one read; common/common_off (HiFi) may be NULL.  Returns the byte size of the tuple stream
 * (written to out when cap suffices), *n_tuples = es_t::size(). */
/* a8/a9 only: the final candidate list of one read: out_cand[4*i..] = {ref_id, rev, tot_anchor_len, n_anchors},
 * anchors {len, pos_enc, pos_ref} appended to out_anchors (capacity in anchors). */
uint32_t orc_encoder_candidates(orc_encoder*, const uint8_t* read, uint32_t len, const uint32_t* neighbours, uint32_t n_nb,
                                const uint64_t* common, const uint32_t* common_off, uint32_t* out_cand, uint32_t* out_anchors, size_t cap_anchors, size_t* n_anchors);
size_t orc_encoder_encode(orc_encoder*, const uint8_t* read, uint32_t len, int has_n, const uint32_t* neighbours, uint32_t n_nb,
                          const uint64_t* common, const uint32_t* common_off, uint8_t* out, size_t cap, uint32_t* n_tuples);

/* ---- a11: the decision logarithm (utils.h:800-810), host libm — the pin of the device's log2 ------- */
void orc_estimator_logs(const uint32_t* count, const uint32_t* total, size_t n, double* out);

#ifdef __cplusplus
}
#endif
#endif
