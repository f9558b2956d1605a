// ref_dump — golden-vector tap around the UNMODIFIED reference classes (TEST INFRASTRUCTURE ONLY).
//
// Linked against oracle/_ref/libcolord_ref.a (objects compiled from /root/reference by
// oracle/Makefile.ref) together with the reference's own main.o / arg_parse.o, so the command line is
// exactly the reference's.  This translation unit supplies the `runCompression` symbol that
// arg_parse.cpp:630 calls; the archive member compression.o is then never pulled by the linker.
// The driver below instantiates the reference's stage objects (CKmerCounter, CKmerFilter,
// CRefReadsAccepter, CInputReads, CReadsSimilarityGraph, CEncoder, CEntrComprReads, CEntrComprQuals,
// CEntrComprHeaders) in the order compression.cpp:344-785 does and inserts two pass-through "tee"
// threads on the inter-stage queues to record what flows through them:
//
//   $COLORD_DUMP_DIR/params.txt   scalars: k a f ci cs n_reads tot_kmers n_unique mean_read_len ...
//   $COLORD_DUMP_DIR/kept.bin     filtered k-mer set as listed from the KMC db: {u64 kmer, u32 count}*
//   $COLORD_DUMP_DIR/accept.bin   one byte per read id (pseudo reads first): CRefReadsAccepter decision
//   $COLORD_DUMP_DIR/cands.bin    per read: u32 read_id, u8 hasN, u32 len, u32 n_ref, u32 ref[n_ref],
//                                 then per ref: u32 n_common, u64 kmer[n_common]   (HiFi only, else 0)
//   $COLORD_DUMP_DIR/es.bin       per read: u32 pack_id, u32 n_tuples, u32 n_bytes, bytes (App. A layout)
//
// The archive itself is written normally; tests compare its stream payloads with the plain `colord`
// binary's archive to prove that the tap changes nothing.
#include "defs.h"
#include "compression.h"
#include "count_kmers.h"
#include "in_reads.h"
#include "reads_sim_graph.h"
#include "encoder.h"
#include "parallel_queue.h"
#include "queues_data.h"
#include "reference_reads.h"
#include "entr_read.h"
#include "entr_qual.h"
#include "entr_header.h"
#include "archive.h"
#include "ref_reads_accepter.h"
#include "reference_genome.h"
#include "kmer_filter.h"
#include "kmc_file.h"
#include "info.h"
#include <thread>
#include <memory>
#include <filesystem>
#include <cstdio>
#include <cstdlib>

namespace fs = std::filesystem;

static std::string dump_dir()
{
	const char* d = getenv("COLORD_DUMP_DIR");
	if (!d) { fprintf(stderr, "ref_dump: COLORD_DUMP_DIR not set\n"); exit(2); }
	fs::create_directories(d);
	return d;
}
static FILE* dump_open(const std::string& name)
{
	FILE* f = fopen((fs::path(dump_dir()) / name).string().c_str(), "wb");
	if (!f) { perror("ref_dump"); exit(2); }
	return f;
}
template<class T> static void put(FILE* f, T v) { fwrite(&v, sizeof(T), 1, f); }

// same table as compression.cpp:42-94 (file-size heuristic); restated here because compression.o is not linked
static void pick_k_a(uint32_t& k, uint32_t& a, bool gz, bool fq, const std::string& path)
{
	if (k && a) return;
	double factor = gz ? (fq ? 2.08 : 3.98) : (fq ? 0.49 : 0.98);
	uint64_t est = static_cast<uint64_t>(factor * (uint64_t)fs::file_size(path));
	static const struct { uint64_t lim; uint32_t k, a; } tab[] = {
		{1000000000ull, 20, 16}, {4000000000ull, 21, 18}, {16000000000ull, 23, 21},
		{48000000000ull, 24, 22}, {128000000000ull, 25, 22}, {~0ull, 26, 23} };
	for (auto& t : tab) if (est < t.lim) { k = t.k; a = t.a; return; }
}

static void dump_es(FILE* f, uint32_t pack_id, es_t& es)
{
	std::vector<uint8_t> raw;
	tuple_types t; uint32_t v1 = 0, v2 = 0;
	es.restart_reading();
	while (es.load(t, v1, v2))
	{
		uint8_t hi = ((uint8_t)t) << 4;
		switch (t)
		{
		case tuple_types::insertion: case tuple_types::substitution: case tuple_types::plain:
			raw.push_back(hi | (uint8_t)v1); break;
		case tuple_types::anchor: case tuple_types::skip:
			raw.push_back(hi | (uint8_t)(v2 >> 24)); raw.push_back(v2 >> 16); raw.push_back(v2 >> 8); raw.push_back(v2); break;
		case tuple_types::alt_id: case tuple_types::start_es:
			raw.push_back(hi | (uint8_t)v2); raw.push_back(v1 >> 24); raw.push_back(v1 >> 16); raw.push_back(v1 >> 8); raw.push_back(v1); break;
		default:
			raw.push_back(hi);
		}
	}
	es.restart_reading();
	if (raw.size() != es.raw_size()) { fprintf(stderr, "ref_dump: es re-serialisation mismatch\n"); exit(2); }
	put<uint32_t>(f, pack_id); put<uint32_t>(f, es.size()); put<uint32_t>(f, (uint32_t)raw.size());
	fwrite(raw.data(), 1, raw.size(), f);
}

void runCompression(const CCompressorParams& params, CInfo& info)
{
	info.version_major = version_major; info.version_minor = version_minor; info.version_patch = version_patch;
	CArchive archive(false);
	if (!archive.Open(params.outputFilePath)) { std::cerr << "cannot open archive\n"; exit(1); }
	int s_meta = archive.RegisterStream("meta");

	const bool gz = izGzipFile(params.inputFilePath);
	const bool fq = isFastq(params.inputFilePath);

	// encoder thread count: compression.cpp:372-387
	int n_enc = (int)params.nThreads - (int)gz - (1 + (fq && params.qualityComprMode != QualityComprMode::None)) - 1;
	if (n_enc < 1) n_enc = 1;
	if (params.nThreads < 20) n_enc += fq ? 2 : 1;

	uint32_t k = params.kmerLen, a = params.anchorLen;
	pick_k_a(k, a, gz, fq, params.inputFilePath);

	auto tmp_dir = create_tmp_dir(fs::path(params.outputFilePath).remove_filename().string());
	std::string db = (fs::path(tmp_dir) / fs::path(params.inputFilePath).filename()).string() + "." + std::to_string(k) + "mers";

	const bool with_genome = !params.refGenomePath.empty();
	std::string kmc_in = params.inputFilePath;
	std::unique_ptr<CReferenceGenome> genome;
	uint32_t genome_overlap = (k - 1) * 10, genome_read_len = 0;
	if (with_genome)
	{
		genome = std::make_unique<CReferenceGenome>(params.refGenomePath, genome_overlap, !params.storeRefGenome, params.verbose);
		std::string gpath = (fs::path(tmp_dir) / (fq ? "refGen.fq" : "refGen.fa")).string();
		genome->Store(gpath, fq);
		if (params.storeRefGenome) genome->Store(archive);
		std::string lst = (fs::path(tmp_dir) / "kmc_file_list.txt").string();
		{ std::ofstream o(lst); o << kmc_in << "\n" << gpath << "\n"; }
		kmc_in = "@" + lst;
	}

	CKmerCounter counter(k, params.minKmerCount, params.maxKmerCount, params.nThreads, params.filterHashModulo, kmc_in, db, tmp_dir, fq, params.verbose);
	auto n_reads = counter.GetNReads();
	auto tot_kmers = counter.GetTotKmers();
	auto n_unique = counter.GetNUniqueCounted();
	std::cerr << "\n";
	uint64_t mean_len = static_cast<uint64_t>((double(tot_kmers * params.filterHashModulo) / n_reads + k - 1));   // compression.cpp:443
	if (with_genome)
	{
		mean_len = double(mean_len * n_reads - genome->GetTotSeqsLen()) / (n_reads - genome->GetTotNSeqs());      // :447
		n_reads -= genome->GetTotNSeqs();
		genome_read_len = 20 * mean_len;
		genome->SetReadLen(genome_read_len);
	}
	info.total_reads = n_reads;

	{	// tap 0: kept set straight from the KMC database, before the temp dir is removed
		FILE* f = dump_open("kept.bin");
		CKMCFile kf;
		if (!kf.OpenForListing(db)) { std::cerr << "cannot list kmc db\n"; exit(1); }
		CKmerAPI km(kf.KmerLength()); uint32_t cnt; std::vector<uint64> v;
		while (kf.ReadNextKmer(km, cnt)) { km.to_long(v); put<uint64_t>(f, v.back()); put<uint32_t>(f, cnt); }
		fclose(f);
	}

	CKmerFilter filter(db, params.filterHashModulo, k, n_unique, params.fillFactorFilteredKmers, params.verbose);
	std::error_code ec; fs::remove_all(tmp_dir, ec);

	const uint32_t q_es = 2 * n_enc;
	CQueueMonitor qm(std::cerr, false, true);
	CParallelQueue<read_pack_t> reads_q(reads_queue_size, 1, &qm, 0);
	CParallelQueue<qual_pack_t> quals_q(quals_queue_size, 1, &qm, 1);
	CParallelQueue<header_pack_t> headers_q(headers_queue_size, 1, &qm, 2);
	CParallelPriorityQueue<std::vector<es_t>> es_qual_q(q_es, n_enc, &qm, 3);
	CParallelQueuePopWaiting<CCompressPack> graph_out_q(compress_queue_size, &qm, 4);   // graph -> tee
	CParallelQueuePopWaiting<CCompressPack> enc_in_q(compress_queue_size, &qm, 4);      // tee -> encoders
	CParallelPriorityQueue<std::vector<es_t>> enc_out_q(q_es, n_enc, &qm, 5);            // encoders -> tee
	CParallelPriorityQueue<std::vector<es_t>> dna_in_q(q_es, 1, &qm, 5);                 // tee -> DNA coder

	uint32_t tot_ref = n_reads;
	uint32_t sparse_range = static_cast<uint32_t>((params.sparseMode_range_symbols * n_unique * params.filterHashModulo) / mean_len);   // :501
	if (!sparse_range) sparse_range = 1;
	double sparse_exp = params.sparseMode_exponent;
	uint32_t n_pseudo = with_genome ? genome->GetNPseudoReads() : 0;
	tot_ref += n_pseudo;
	CRefReadsAccepter accepter(sparse_range, sparse_exp, n_pseudo);
	const bool sparse = params.referenceReadsMode == ReferenceReadsMode::Sparse;
	if (sparse) tot_ref = accepter.GetNAccepted(n_reads);

	{	// tap: acceptance decisions from a fresh copy (identical RNG stream, ref_reads_accepter.h:42-49)
		FILE* f = dump_open("accept.bin");
		CRefReadsAccepter cp(sparse_range, sparse_exp, n_pseudo);
		for (uint32_t i = 0; i < n_reads + n_pseudo; ++i) put<uint8_t>(f, sparse ? cp.ShouldAddToReference(i) : 1);
		fclose(f);
	}
	{
		FILE* f = dump_open("params.txt");
		fprintf(f, "k %u\na %u\nf %u\nci %u\ncs %u\nc %u\nlevel %d\nsource %d\nsparse %d\nn_reads %u\ntot_kmers %llu\nn_unique %llu\n"
			"mean_read_len %llu\nsparse_range %u\nsparse_exp %.17g\ntot_ref_reads %u\nn_pseudo %u\ntotal_count_filtered %llu\nn_enc %d\nis_fastq %d\nqual_mode %d\n",
			k, a, params.filterHashModulo, params.minKmerCount, params.maxKmerCount, params.maxCandidates, params.compressionLevel,
			(int)params.dataSource, (int)sparse, n_reads, (unsigned long long)tot_kmers, (unsigned long long)n_unique,
			(unsigned long long)mean_len, sparse_range, sparse_exp, tot_ref, n_pseudo, (unsigned long long)filter.GetTotalKmers(), n_enc, (int)fq,
			(int)params.qualityComprMode);
		fclose(f);
	}

	CReferenceReads ref_reads(tot_ref);
	uint64_t hdr_symb = 0;

	std::thread t_reader([&] {
		CInputReads in(params.verbose, params.inputFilePath, reads_q, quals_q, headers_q);
		in.GetStats(info.total_bytes, info.total_bases, hdr_symb);
	});
	double fill_k2r = params.fillFactorKmersToReads;
	std::thread t_graph([&] {
		CReadsSimilarityGraph g(reads_q, graph_out_q, ref_reads, genome.get(), filter, k, params.maxCandidates, params.maxKmerCount,
			params.referenceReadsMode, accepter, (double)tot_ref / n_reads, n_enc, params.dataSource, fill_k2r, params.verbose);
	});
	std::thread t_tee_cands([&] {
		FILE* f = dump_open("cands.bin");
		CCompressPack p;
		while (graph_out_q.Pop(p))
		{
			for (auto& e : p.data)
			{
				put<uint32_t>(f, e.read_id); put<uint8_t>(f, e.hasN); put<uint32_t>(f, (uint32_t)read_len(e.read));
				put<uint32_t>(f, (uint32_t)e.ref_reads.size());
				for (auto r : e.ref_reads) put<uint32_t>(f, r);
				for (size_t i = 0; i < e.ref_reads.size(); ++i)
				{
					uint32_t n = i < e.common_kmers.size() ? (uint32_t)e.common_kmers[i].size() : 0;
					put<uint32_t>(f, n);
					for (uint32_t j = 0; j < n; ++j) put<uint64_t>(f, e.common_kmers[i][j]);
				}
			}
			enc_in_q.Push(std::move(p));
		}
		enc_in_q.MarkCompleted();
		fclose(f);
	});
	std::vector<std::thread> t_enc;
	for (int i = 0; i < n_enc; ++i)
		t_enc.emplace_back([&] {
			CEncoder enc(params.verbose, enc_in_q, ref_reads, enc_out_q, es_qual_q, a,
				params.minFractionOfMmersInEncodeToAlwaysEncode, params.minFractionOfMmersInEncode, params.maxMatchesMultiplier,
				params.editScriptCostMultiplier, params.minPartLenToConsiderAltRead, params.maxRecurence, params.minAnchors,
				fq, params.filterHashModulo, k, params.dataSource);
			enc.Encode();
		});
	std::thread t_tee_es([&] {
		FILE* f = dump_open("es.bin");
		std::vector<es_t> pack; uint32_t id = 0;
		while (enc_out_q.Pop(pack))
		{
			for (auto& es : pack) dump_es(f, id, es);
			dna_in_q.Push(id, std::move(pack));
			pack.clear();
			++id;
		}
		dna_in_q.MarkCompleted();
		fclose(f);
	});
	std::thread t_dna([&] {
		CEntrComprReads c{ dna_in_q, ref_reads, params.verbose, params.maxCandidates, params.compressionLevel, n_reads * mean_len, archive, n_reads, n_pseudo };
		c.Compress();
	});
	std::thread t_qual;
	if (fq)
		t_qual = std::thread([&] {
			CEntrComprQuals c{ quals_q, archive, params.qualityComprMode, params.qualityFwdThresholds, params.qualityRevThresholds, params.verbose,
				params.compressionLevel, n_reads * mean_len, es_qual_q, params.dataSource };
			c.Compress();
		});
	std::thread t_hdr([&] {
		CEntrComprHeaders c{ headers_q, archive, params.headerComprMode, params.compressionLevel, params.verbose };
		c.Compress();
	});

	if (fq) t_qual.join();
	t_hdr.join(); t_graph.join(); t_tee_cands.join(); t_reader.join();
	for (auto& t : t_enc) t.join();
	t_tee_es.join(); t_dna.join();

	// meta stream, field order of compression.cpp:705-779
	std::vector<uint8_t> m;
	StoreLittleEndian(m, tot_ref);
	StoreLittleEndian(m, params.maxCandidates);
	StoreLittleEndian(m, params.compressionLevel);
	StoreLittleEndian(m, static_cast<uint8_t>(params.dataSource));
	StoreLittleEndian(m, n_reads * mean_len);
	if (fq)
	{
		m.push_back(static_cast<uint8_t>(params.qualityComprMode));
		size_t nv = 0;
		switch (params.qualityComprMode)
		{
		case QualityComprMode::None: nv = 1; break;
		case QualityComprMode::BinaryThreshold: nv = 2; break;
		case QualityComprMode::QuadThreshold: nv = 4; break;
		case QualityComprMode::QuinaryThreshold: nv = 5; break;
		default: nv = 0;
		}
		for (size_t i = 0; i < nv; ++i) StoreLittleEndian(m, params.qualityRevThresholds[i]);
	}
	m.push_back(static_cast<uint8_t>(params.headerComprMode));
	m.push_back(static_cast<uint8_t>(params.referenceReadsMode));
	if (sparse) { StoreLittleEndian(m, sparse_range); StoreLittleEndian(m, sparse_exp); }
	m.push_back(static_cast<uint8_t>(with_genome));
	if (with_genome)
	{
		m.push_back(static_cast<uint8_t>(params.storeRefGenome));
		StoreLittleEndian(m, genome_read_len); StoreLittleEndian(m, genome_overlap); StoreLittleEndian(m, n_pseudo);
		if (!params.storeRefGenome) for (auto c : genome->GetChecksum()) StoreLittleEndian(m, c);
	}
	archive.AddPart(s_meta, m, 0);
	int s_info = archive.RegisterStream("info");
	auto inf = info.Serialize();
	archive.AddPart(s_info, inf, 0ull);
	archive.Close();
	std::cerr << "ref_dump: done, dumps in " << dump_dir() << "\n";
}
