// reader.hpp — a CoLoRd archive as a stream of records: the reference's CDecmpressionModule (src/colord/decompression_common.cpp:27-341)
// on top of the library's host decoders (cl_dna_decode_part / cl_qual_decode_part / cl_id_decode_part).  Shared by
// `colord_hip decompress` (cli/decompress.cpp, the FASTQ / FASTA writers of decompression.cpp:84-258) and by the public C++ API
// (include/colord_api.h, api/colord_api.cpp — the reference's src/API/colord_api.h).
//
// Three host threads decode the `dna`, `qual` and `header` streams part by part (the quality decoder consumes the bases the DNA
// decoder produced for the same part, entr_qual.h:136-260); next() hands out the records in file order.  Archives written by
// several GPUs carry a `hipdomains` stream: the first `dna` part of every model domain, where both coders start from fresh
// models.  No GPU is needed.  Errors are std::runtime_error (the command-line tool turns them into its exit message).
#pragma once
#include "colord_hip.h"
#include "archive.hpp"
#include "genome_io.hpp"
#include <condition_variable>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <thread>

#include <ctime>
// COLORD_HIP_DECODE_DEBUG: CPU seconds of a decoder thread on stderr when it ends
static inline void thread_report(const char* what)
{
	if (!getenv("COLORD_HIP_DECODE_DEBUG")) return;
	timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t);
	fprintf(stderr, "[decode] %s thread: %.2f s of CPU\n", what, (double)t.tv_sec + 1e-9 * (double)t.tv_nsec);
}

namespace colord_hip_reader {
template<class T> struct Queue {                                       // bounded hand-over between the stream threads
	std::mutex m; std::condition_variable cv; std::deque<T> q; bool done = false; size_t cap = 4;
	void push(T&& v) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return q.size() < cap || done; }); if (done) return; q.push_back(std::move(v)); cv.notify_all(); }
	bool pop(T& v) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty() || done; }); if (q.empty()) return false; v = std::move(q.front()); q.pop_front(); cv.notify_all(); return true; }
	void finish() { std::unique_lock<std::mutex> l(m); done = true; cv.notify_all(); }
	void abort() { std::unique_lock<std::mutex> l(m); done = true; q.clear(); cv.notify_all(); }          // the consumer goes away: producers must not block
};
struct ReadPart { std::vector<uint8_t> bases; std::vector<uint64_t> off; std::vector<uint8_t> quals; };
struct HeaderPart { std::vector<uint8_t> ids; std::vector<uint64_t> off; std::vector<uint8_t> plus; };

struct Meta {                                                          // the `meta` stream (compression.cpp:704-779)
	uint32_t tot_ref_reads = 0, max_candidates = 0; int32_t level = 1; uint8_t source = 0; uint64_t approx_size = 0;
	uint8_t qual_mode = 8; std::vector<uint32_t> rev; uint8_t header_mode = 0, ref_mode = 0; uint32_t sparse_range = 0; double sparse_exp = 0;
	bool genome = false, genome_in_archive = false; uint32_t genome_read_len = 0, genome_overlap = 0, n_pseudo = 0; uint8_t genome_md5[16] = { 0 };
};
struct ArchiveInfo {                                                   // the `info` stream (compression.cpp:42-95, info.cpp:24-53)
	uint32_t version_major = 0, version_minor = 0, version_patch = 0; uint64_t total_bytes = 0, total_bases = 0; uint32_t total_reads = 0; uint64_t time = 0;
	std::string command_line;
};
template<class T> T rd(const uint8_t*& p, const uint8_t* e) { if (p + sizeof(T) > e) throw std::runtime_error("truncated stream in the archive"); T v; memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
inline Meta parse_meta(const std::vector<uint8_t>& b, bool is_fastq)         // decompression_common.cpp:51-265
{
	Meta m; const uint8_t* p = b.data(); const uint8_t* e = p + b.size();
	m.tot_ref_reads = rd<uint32_t>(p, e); m.max_candidates = rd<uint32_t>(p, e); m.level = rd<int32_t>(p, e); m.source = rd<uint8_t>(p, e); m.approx_size = rd<uint64_t>(p, e);
	if (is_fastq)
	{
		m.qual_mode = rd<uint8_t>(p, e);
		const int n_rev = m.qual_mode == 8 ? 1 : m.qual_mode == 6 ? 2 : m.qual_mode == 5 ? 4 : m.qual_mode == 4 ? 5 : 0;     // None / 2-fix / 4-fix / 5-fix
		for (int i = 0; i < n_rev; ++i) m.rev.push_back(rd<uint32_t>(p, e));
	}
	m.header_mode = rd<uint8_t>(p, e); m.ref_mode = rd<uint8_t>(p, e);
	if (m.ref_mode == 1) { m.sparse_range = rd<uint32_t>(p, e); m.sparse_exp = rd<double>(p, e); }
	m.genome = rd<uint8_t>(p, e) != 0;
	if (m.genome)
	{	// decompression_common.cpp:231-260
		m.genome_in_archive = rd<uint8_t>(p, e) != 0;
		m.genome_read_len = rd<uint32_t>(p, e); m.genome_overlap = rd<uint32_t>(p, e); m.n_pseudo = rd<uint32_t>(p, e);
		if (!m.genome_in_archive) for (int i = 0; i < 16; ++i) m.genome_md5[i] = rd<uint8_t>(p, e);
	}
	return m;
}
inline ArchiveInfo parse_info(const std::vector<uint8_t>& b)
{
	ArchiveInfo I; const uint8_t* p = b.data(); const uint8_t* e = p + b.size();
	I.version_major = rd<uint32_t>(p, e); I.version_minor = rd<uint32_t>(p, e); I.version_patch = rd<uint32_t>(p, e);
	I.total_bytes = rd<uint64_t>(p, e); I.total_bases = rd<uint64_t>(p, e); I.total_reads = rd<uint32_t>(p, e); I.time = rd<uint64_t>(p, e);
	const uint32_t cl = rd<uint32_t>(p, e);
	I.command_line.assign((const char*)p, (const char*)p + std::min<size_t>(cl, (size_t)(e - p)));
	return I;
}

struct Record { const uint8_t* header; size_t header_len; const uint8_t* bases; size_t n_bases; const uint8_t* quals; bool plus_is_header; };   // views, valid until the next call

// every id of an archive, decoded once: what the domain-parallel decompressor's workers share (the `header` stream is one coder over
// the whole file, entr_header.cpp:46-80)
struct HeaderCache {
	std::vector<uint8_t> ids; std::vector<uint64_t> off{ 0 }; std::vector<uint8_t> plus; std::string err;
	void decode_all(const std::string& path)
	{
		ArchiveReader ar; if (!ar.open(path)) { err = "cannot open archive: " + path; return; }
		const int s_hdr = ar.id("header"), s_meta = ar.id("meta");
		std::vector<uint8_t> mb, in; uint64_t mm = 0, n = 0;
		cl_id_decoder* c = nullptr;
		try
		{
			if (s_hdr < 0 || s_meta < 0 || !ar.part(s_meta, 0, mb, mm)) throw std::runtime_error("header / meta stream missing");
			const Meta M = parse_meta(mb, ar.id("qual") >= 0);
			if (cl_id_decoder_create(M.header_mode, &c) != CL_OK) throw std::runtime_error("cl_id_decoder_create");
			for (size_t p = 0; p < ar.n_parts(s_hdr); ++p)
			{
				if (!ar.part(s_hdr, p, in, n) || n > 0xffffffffull) throw std::runtime_error("cannot read a `header` part");
				std::vector<uint64_t> o(n + 1); std::vector<uint8_t> pl(n), buf(std::max<uint64_t>(in.size() * 64, 1 << 20)); uint64_t got = 0;
				cl_status st = cl_id_decode_part(c, in.data(), in.size(), (uint32_t)n, buf.data(), buf.size(), o.data(), pl.data(), &got);
				if (st == CL_E_CAPACITY) { buf.resize(got); st = cl_id_decode_part(c, in.data(), in.size(), (uint32_t)n, buf.data(), got, o.data(), pl.data(), &got); }
				if (st != CL_OK) throw std::runtime_error("corrupt `header` part");
				const uint64_t base = ids.size();
				ids.insert(ids.end(), buf.begin(), buf.begin() + got); plus.insert(plus.end(), pl.begin(), pl.end());
				for (uint64_t i = 1; i <= n; ++i) off.push_back(base + o[i]);
			}
		}
		catch (const std::exception& e) { err = e.what(); }
		if (c) cl_id_decoder_free(c);
		ar.close();
	}
};

class RecordStream {
	ArchiveReader ar; Meta M; ArchiveInfo I; bool fastq = false, started = false, finished = false;
	int s_dna = -1, s_qual = -1, s_hdr = -1;
	std::vector<uint64_t> domain_first_part;                              // first `dna` part of every model domain after the first
	// INDEPENDENT domains (`colord_hip compress-* --domains K`: bit 31 of the count in `hipdomains`): a domain has its own reference reads,
	// so it is decoded by a DNA decoder of its own (with its own sparse range) — and K of them side by side (cli/decompress.cpp)
	bool independent = false; std::vector<uint64_t> dom_part{ 0 }, dom_read{ 0 }; std::vector<uint32_t> dom_sparse;
	int only_domain = -1; const HeaderCache* ext_hdr = nullptr; uint64_t read_index = 0;
	size_t part_begin() const { return only_domain >= 0 ? (size_t)dom_part[only_domain] : 0; }
	size_t part_end() const { return only_domain >= 0 && (size_t)only_domain + 1 < dom_part.size() ? (size_t)dom_part[only_domain + 1] : ar.n_parts(s_dna); }
	genome_io::Sequences pseudo;                                          // reference-genome mode: the pseudo reads that precede the first read
	Queue<ReadPart> q_bases_for_qual, q_reads, q_quals; Queue<HeaderPart> q_hdr;
	std::string err_dna, err_qual, err_hdr;
	std::thread t_dna, t_qual, t_hdr;
	ReadPart rp, qp; HeaderPart hp; size_t ri = 0, hi = 0; bool have_r = false, have_h = false;
	bool is_domain_start(size_t part) const { for (uint64_t f : domain_first_part) if (f == part) return true; return false; }
	// Part metadata is a record count read from the file: it sizes vectors, so it is checked before any allocation — against
	// 2^32 (the decoders count in 32 bits) and against the archive's own total (`info` stream) when there is one.
	bool count_ok(uint64_t n) const { return n <= 0xffffffffull && (I.total_reads == 0 || n <= I.total_reads); }
	void start();
	void join() { if (t_dna.joinable()) t_dna.join(); if (t_qual.joinable()) t_qual.join(); if (t_hdr.joinable()) t_hdr.join(); }
public:
	// only_domain >= 0 (archives with independent domains): the records of that domain only, their ids from `headers` (all ids of the archive)
	explicit RecordStream(const std::string& path, const std::string& genome_path = "", int only_domain = -1, const HeaderCache* headers = nullptr);
	bool independent_domains() const { return independent; }
	size_t n_domains() const { return dom_part.size(); }
	void prefetch() { if (!started) start(); }                           // starts the decoder threads before the first next()
	~RecordStream() { q_bases_for_qual.abort(); q_reads.abort(); q_quals.abort(); q_hdr.abort(); join(); ar.close(); }
	RecordStream(const RecordStream&) = delete; RecordStream& operator=(const RecordStream&) = delete;
	bool is_fastq() const { return fastq; }
	const Meta& meta() const { return M; }
	const ArchiveInfo& info() const { return I; }
	bool next(Record& r);                                                 // false at the end; throws on a corrupt archive
};

inline RecordStream::RecordStream(const std::string& path, const std::string& genome_path, int only_domain_, const HeaderCache* headers)
{
	only_domain = only_domain_; ext_hdr = headers;
	if (!ar.open(path)) throw std::runtime_error("cannot open archive: " + path);
	s_dna = ar.id("dna"); s_qual = ar.id("qual"); s_hdr = ar.id("header");
	const int s_meta = ar.id("meta"), s_dom = ar.id("hipdomains"), s_info = ar.id("info");
	if (s_dna < 0 || s_hdr < 0 || s_meta < 0) throw std::runtime_error("not a CoLoRd archive (dna / header / meta stream missing)");
	fastq = s_qual >= 0;
	std::vector<uint8_t> mb; uint64_t mm = 0;
	if (!ar.part(s_meta, 0, mb, mm)) throw std::runtime_error("cannot read the `meta` stream");
	M = parse_meta(mb, fastq);
	if (s_info >= 0 && ar.part(s_info, 0, mb, mm) && mb.size() >= 40) I = parse_info(mb);
	if (M.genome)
	{	// the genome from the archive (-s at compression) or from the caller's file, whose checksum must be the one of the
		// compression (decompression_common.cpp:262-305); its pseudo reads seed the decoder's reference reads
		genome_io::Sequences G;
		if (M.genome_in_archive)
		{
			const int s_gen = ar.id("ref-genome");
			std::vector<uint8_t> gb; uint64_t n_seqs = 0;
			if (s_gen < 0 || !ar.part(s_gen, 0, gb, n_seqs)) throw std::runtime_error("cannot read the `ref-genome` stream");
			G.off.assign(n_seqs + 1, 0);
			uint64_t got = 0;
			G.codes.resize(std::max<size_t>(gb.size() * 5, 1 << 20));
			cl_status st = cl_genome_decode(gb.data(), gb.size(), (uint32_t)n_seqs, G.codes.data(), G.codes.size(), G.off.data(), &got);
			if (st == CL_E_CAPACITY) { G.codes.resize(got); st = cl_genome_decode(gb.data(), gb.size(), (uint32_t)n_seqs, G.codes.data(), G.codes.size(), G.off.data(), &got); }
			if (st != CL_OK) throw std::runtime_error("corrupt `ref-genome` stream");
			G.codes.resize(got);
		}
		else
		{
			if (genome_path.empty()) throw std::runtime_error("compressed file was created without -s switch, reference genome is required for decompression");
			G = genome_io::read_fasta(genome_path);
			uint8_t md[16];
			if (cl_genome_md5(G.codes.data(), G.off.data(), (uint32_t)(G.off.size() - 1), md) != CL_OK || memcmp(md, M.genome_md5, 16) != 0)
				throw std::runtime_error("different reference genome was used during compression. Decompression impossible.");
		}
		pseudo = genome_io::pseudo_reads(G, M.genome_read_len, M.genome_overlap);
		if (pseudo.off.size() - 1 != M.n_pseudo) throw std::runtime_error("reference genome: the number of pseudo reads differs from the archive's");
	}
	if (s_dom >= 0)
	{
		std::vector<uint8_t> db; uint64_t dm = 0;
		if (!ar.part(s_dom, 0, db, dm)) throw std::runtime_error("cannot read the `hipdomains` stream");
		const uint8_t* p = db.data(); const uint8_t* e = p + db.size();
		const uint32_t nf = rd<uint32_t>(p, e), n = nf & 0x7fffffffu;
		independent = (nf >> 31) != 0;
		if (n > db.size() / 16) throw std::runtime_error("corrupt `hipdomains` stream");
		for (uint32_t i = 0; i < n; ++i)
		{
			const uint64_t fr = rd<uint64_t>(p, e), fp = rd<uint64_t>(p, e);
			if (fp > ar.n_parts(s_dna) || (i && fp < dom_part.back())) throw std::runtime_error("corrupt `hipdomains` stream");
			if (i) { domain_first_part.push_back(fp); dom_part.push_back(fp); dom_read.push_back(fr); }
		}
		if (independent) for (uint32_t i = 0; i < n; ++i) dom_sparse.push_back(rd<uint32_t>(p, e));
	}
	if (only_domain >= 0 && (!independent || (size_t)only_domain >= dom_part.size() || !ext_hdr)) throw std::runtime_error("a single domain can be read from an archive with independent domains only");
	if (only_domain >= 0) read_index = dom_read[only_domain];
	if (fastq && ar.n_parts(s_qual) != ar.n_parts(s_dna)) throw std::runtime_error("`dna` and `qual` streams have different numbers of parts");
}

inline void RecordStream::start()
{
	started = true;
	const size_t n_parts = part_end(), p_first = part_begin();
	t_dna = std::thread([this, n_parts, p_first]() {
		cl_dna_decoder* d = nullptr;
		try {
		const uint32_t sr0 = independent && !dom_sparse.empty() ? dom_sparse[only_domain >= 0 ? only_domain : 0] : M.sparse_range;
		if (cl_dna_decoder_create(M.max_candidates, M.level, M.n_pseudo, M.n_pseudo, M.ref_mode == 0, sr0, M.sparse_exp, &d) != CL_OK) { err_dna = "cl_dna_decoder_create"; }
		for (size_t i = 0; d && i + 1 < pseudo.off.size(); ++i)                    // decompression_common.cpp:287-292
			if (cl_dna_decoder_add_ref(d, pseudo.codes.data() + pseudo.off[i], (uint32_t)(pseudo.off[i + 1] - pseudo.off[i])) != CL_OK) { err_dna = "cl_dna_decoder_add_ref"; break; }
		if (!err_dna.empty() && d) { cl_dna_decoder_free(d); d = nullptr; }
		std::vector<uint8_t> in; uint64_t n_reads = 0;
		for (size_t p = p_first; d && p < n_parts; ++p)
		{
			if (!ar.part(s_dna, p, in, n_reads)) { err_dna = "cannot read a `dna` part"; break; }
			if (!count_ok(n_reads)) { err_dna = "a `dna` part claims more reads than the archive holds"; break; }
			if (p > p_first && is_domain_start(p))
			{
				if (!independent) cl_dna_decoder_new_domain(d);
				else
				{	// an independent domain: nothing of the domains before it is a reference read here
					size_t di = 0; while (di + 1 < dom_part.size() && dom_part[di + 1] <= p) ++di;
					cl_dna_decoder_free(d); d = nullptr;
					if (cl_dna_decoder_create(M.max_candidates, M.level, 0, 0, M.ref_mode == 0, di < dom_sparse.size() ? dom_sparse[di] : M.sparse_range, M.sparse_exp, &d) != CL_OK) { err_dna = "cl_dna_decoder_create"; break; }
				}
			}
			ReadPart x; x.off.resize(n_reads + 1);
			uint64_t cap = std::max<uint64_t>(in.size() * 8, 1 << 20), got = 0;
			x.bases.resize(cap);
			cl_status s = cl_dna_decode_part(d, in.data(), in.size(), (uint32_t)n_reads, x.bases.data(), cap, x.off.data(), &got);
			if (s == CL_E_CAPACITY) { x.bases.resize(got); s = cl_dna_decode_part(d, in.data(), in.size(), (uint32_t)n_reads, x.bases.data(), got, x.off.data(), &got); }   // the decoded part is kept inside
			if (s != CL_OK) { err_dna = cl_dna_decoder_error(d); break; }
			x.bases.resize(got);
			if (fastq) { ReadPart cp; cp.bases = x.bases; cp.off = x.off; q_bases_for_qual.push(std::move(cp)); }
			q_reads.push(std::move(x));
		}
		} catch (const std::exception& e) { err_dna = std::string("corrupt `dna` part (") + e.what() + ")"; }
		if (d) cl_dna_decoder_free(d);
		thread_report("dna");
		q_bases_for_qual.finish(); q_reads.finish();
	});
	t_qual = std::thread([this]() {
		if (!fastq) { q_quals.finish(); return; }
		cl_qual_params qpar{}; qpar.mode = M.qual_mode; qpar.source = M.source; qpar.level = M.level; qpar.n_rev = (uint32_t)M.rev.size();
		for (size_t i = 0; i < M.rev.size(); ++i) qpar.rev[i] = M.rev[i];
		cl_qual_decoder* q = nullptr;
		if (cl_qual_decoder_create(&qpar, &q) != CL_OK) { err_qual = "cl_qual_decoder_create"; }
		ReadPart x; std::vector<uint8_t> in; uint64_t meta = 0; size_t p = part_begin(); const size_t p_first = p;
		try {
		while (q && q_bases_for_qual.pop(x))
		{
			if (!ar.part(s_qual, p, in, meta)) { err_qual = "cannot read a `qual` part"; break; }
			if (p > p_first && is_domain_start(p)) cl_qual_decoder_new_domain(q);
			x.quals.resize(x.bases.size());
			if (cl_qual_decode_part(q, in.data(), in.size(), x.bases.data(), x.off.data(), (uint32_t)(x.off.size() - 1), x.quals.data()) != CL_OK) { err_qual = "corrupt `qual` part"; break; }
			x.bases.clear(); x.bases.shrink_to_fit();
			q_quals.push(std::move(x));
			++p;
		}
		} catch (const std::exception& e) { err_qual = std::string("corrupt `qual` part (") + e.what() + ")"; }
		while (q_bases_for_qual.pop(x)) {}                                    // drain after an error so that the producer can finish
		if (q) cl_qual_decoder_free(q);
		thread_report("qual");
		q_quals.finish();
	});
	if (ext_hdr) { q_hdr.finish(); return; }
	t_hdr = std::thread([this]() {
		cl_id_decoder* c = nullptr;
		if (cl_id_decoder_create(M.header_mode, &c) != CL_OK) { err_hdr = "cl_id_decoder_create"; }
		std::vector<uint8_t> in; uint64_t n = 0;
		try {
		for (size_t p = 0; c && p < ar.n_parts(s_hdr); ++p)
		{
			if (!ar.part(s_hdr, p, in, n)) { err_hdr = "cannot read a `header` part"; break; }
			if (!count_ok(n)) { err_hdr = "a `header` part claims more records than the archive holds"; break; }
			HeaderPart x; x.off.resize(n + 1); x.plus.resize(n);
			uint64_t cap = std::max<uint64_t>(in.size() * 64, 1 << 20), got = 0;
			x.ids.resize(cap);
			cl_status s = cl_id_decode_part(c, in.data(), in.size(), (uint32_t)n, x.ids.data(), cap, x.off.data(), x.plus.data(), &got);
			if (s == CL_E_CAPACITY) { x.ids.resize(got); s = cl_id_decode_part(c, in.data(), in.size(), (uint32_t)n, x.ids.data(), got, x.off.data(), x.plus.data(), &got); }
			if (s != CL_OK) { err_hdr = "corrupt `header` part"; break; }
			x.ids.resize(got);
			q_hdr.push(std::move(x));
		}
		} catch (const std::exception& e) { err_hdr = std::string("corrupt `header` part (") + e.what() + ")"; }
		if (c) cl_id_decoder_free(c);
		thread_report("header");
		q_hdr.finish();
	});
}

inline bool RecordStream::next(Record& r)
{
	if (finished) return false;
	if (!started) start();
	auto next_read = [&]() { while (!have_r || ri + 1 >= rp.off.size()) { if (!q_reads.pop(rp)) return false; if (fastq && !q_quals.pop(qp)) return false; ri = 0; have_r = true; } return true; };
	auto next_hdr = [&]() { if (ext_hdr) return read_index + 1 < ext_hdr->off.size(); while (!have_h || hi + 1 >= hp.off.size()) { if (!q_hdr.pop(hp)) return false; hi = 0; have_h = true; } return true; };
	const bool a = next_read(), b = ext_hdr ? (a ? next_hdr() : false) : next_hdr();
	if (!a || !b)
	{	// the end, or an error: let the producers run out, then report
		finished = true;
		{ ReadPart x; while (q_reads.pop(x)) {} while (q_quals.pop(x)) {} HeaderPart y; while (q_hdr.pop(y)) {} }
		join();
		if (!err_dna.empty()) throw std::runtime_error("dna stream: " + err_dna);
		if (!err_qual.empty()) throw std::runtime_error("qual stream: " + err_qual);
		if (!err_hdr.empty()) throw std::runtime_error("header stream: " + err_hdr);
		if (a != b) throw std::runtime_error("the streams hold different numbers of records");
		return false;
	}
	const uint64_t b0 = rp.off[ri], b1 = rp.off[ri + 1];
	if (ext_hdr)
	{
		const uint64_t h0 = ext_hdr->off[read_index], h1 = ext_hdr->off[read_index + 1];
		r.header = ext_hdr->ids.data() + h0; r.header_len = (size_t)(h1 - h0);
		r.bases = rp.bases.data() + b0; r.n_bases = (size_t)(b1 - b0);
		r.quals = fastq ? qp.quals.data() + b0 : nullptr;
		r.plus_is_header = fastq && ext_hdr->plus[read_index] != 0;
		++ri; ++read_index;
		return true;
	}
	const uint64_t h0 = hp.off[hi], h1 = hp.off[hi + 1];
	r.header = hp.ids.data() + h0; r.header_len = (size_t)(h1 - h0);
	r.bases = rp.bases.data() + b0; r.n_bases = (size_t)(b1 - b0);
	r.quals = fastq ? qp.quals.data() + b0 : nullptr;
	r.plus_is_header = fastq && hp.plus[hi] != 0;
	++ri; ++hi;
	return true;
}
} // namespace colord_hip_reader
