// decompress.cpp — `colord_hip decompress in.colord out.fastq|out.fasta` and `colord_hip info in.colord`: the reference's
// runDecompression / CDecmpressionModule::Run (src/colord/decompression.cpp:84-258, decompression_common.cpp:27-341) and runInfo
// (info.cpp:24-53) on top of the library's decoders (cl_dna_decode_part / cl_qual_decode_part / cl_id_decode_part).
//
// Three host threads decode the `dna`, `qual` and `header` streams part by part (the quality decoder consumes the bases the DNA
// decoder produced for the same part, entr_qual.h:136-260); the main thread writes FASTQ (or FASTA when the archive has no `qual`
// stream).  Archives written by several GPUs carry a `hipdomains` stream: the first `dna` part of every model domain, where both
// coders start from fresh models.  No GPU is needed to decompress.
#include "colord_hip.h"
#include "archive.hpp"
#include <condition_variable>
#include <ctime>
#include <deque>
#include <mutex>
#include <thread>

namespace {
template<class T> struct Queue {                                       // bounded hand-over between the stream threads
	std::mutex m; std::condition_variable cv; std::deque<T> q; bool done = false; size_t cap = 4;
	void push(T&& v) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return q.size() < cap; }); q.push_back(std::move(v)); cv.notify_all(); }
	bool pop(T& v) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty() || done; }); if (q.empty()) return false; v = std::move(q.front()); q.pop_front(); cv.notify_all(); return true; }
	void finish() { std::unique_lock<std::mutex> l(m); done = true; cv.notify_all(); }
};
struct ReadPart { std::vector<uint8_t> bases; std::vector<uint64_t> off; std::vector<uint8_t> quals; };
struct HeaderPart { std::vector<uint8_t> ids; std::vector<uint64_t> off; std::vector<uint8_t> plus; };

struct Meta {
	uint32_t tot_ref_reads = 0, max_candidates = 0; int32_t level = 1; uint8_t source = 0; uint64_t approx_size = 0;
	uint8_t qual_mode = 8; std::vector<uint32_t> rev; uint8_t header_mode = 0, ref_mode = 0; uint32_t sparse_range = 0; double sparse_exp = 0;
	bool genome = false;
};
template<class T> T rd(const uint8_t*& p, const uint8_t* e) { if (p + sizeof(T) > e) die("truncated `meta` stream"); T v; memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
Meta parse_meta(const std::vector<uint8_t>& b, bool is_fastq)         // decompression_common.cpp:51-265
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
	return m;
}
} // namespace

int run_info(int argc, char** argv)
{
	if (argc < 3) { fprintf(stderr, "usage: colord_hip info archive.colord\n"); return 1; }
	ArchiveReader ar;
	if (!ar.open(argv[2])) die(std::string("cannot open archive: ") + argv[2]);
	std::vector<uint8_t> b; uint64_t meta = 0;
	if (!ar.part(ar.id("info"), 0, b, meta) || b.size() < 40) die("archive without a readable `info` stream");
	const uint8_t* p = b.data(); const uint8_t* e = p + b.size();
	const uint32_t vmaj = rd<uint32_t>(p, e), vmin = rd<uint32_t>(p, e), vpat = rd<uint32_t>(p, e);
	const uint64_t bytes = rd<uint64_t>(p, e), bases = rd<uint64_t>(p, e); const uint32_t reads = rd<uint32_t>(p, e); const uint64_t tm = rd<uint64_t>(p, e);
	const uint32_t cl = rd<uint32_t>(p, e);
	std::string cmd((const char*)p, (const char*)p + std::min<size_t>(cl, (size_t)(e - p)));
	time_t t = (time_t)tm;
	fprintf(stderr, "version major: %u\nversion minor: %u\nversion patch: %u\ntotal bytes: %llu\ntotal bases: %llu\ntotal reads: %u\ntime: %s\ncommand: %s\n",
		vmaj, vmin, vpat, (unsigned long long)bytes, (unsigned long long)bases, reads, asctime(localtime(&t)), cmd.c_str());
	return 0;
}

int run_decompress(int argc, char** argv)
{
	std::vector<std::string> pos;
	for (int i = 2; i < argc; ++i)
	{
		const std::string a = argv[i];
		if ((a == "-G" || a == "--reference-genome") && i + 1 < argc) die("decompress -G: archives that need an external reference genome are not supported yet (compress with -s)");
		else if (a == "-v" || a == "--verbose") ;
		else pos.push_back(a);
	}
	if (pos.size() != 2) { fprintf(stderr, "usage: colord_hip decompress archive.colord output.fastq\n"); return 1; }
	ArchiveReader ar;
	if (!ar.open(pos[0])) die("cannot open archive: " + pos[0]);
	const int s_dna = ar.id("dna"), s_qual = ar.id("qual"), s_hdr = ar.id("header"), s_meta = ar.id("meta"), s_dom = ar.id("hipdomains");
	if (s_dna < 0 || s_hdr < 0 || s_meta < 0) die("not a CoLoRd archive (dna / header / meta stream missing)");
	const bool is_fastq = s_qual >= 0;
	std::vector<uint8_t> mb; uint64_t mm = 0;
	if (!ar.part(s_meta, 0, mb, mm)) die("cannot read the `meta` stream");
	const Meta M = parse_meta(mb, is_fastq);
	if (M.genome) die("archives compressed against a reference genome (-G) are not supported by this decompressor yet");
	std::vector<uint64_t> domain_first_part;                              // first `dna` part of every model domain after the first
	if (s_dom >= 0)
	{
		std::vector<uint8_t> db; uint64_t dm = 0;
		if (!ar.part(s_dom, 0, db, dm)) die("cannot read the `hipdomains` stream");
		const uint8_t* p = db.data(); const uint8_t* e = p + db.size();
		const uint32_t n = rd<uint32_t>(p, e);
		for (uint32_t i = 0; i < n; ++i) { (void)rd<uint64_t>(p, e); const uint64_t fp = rd<uint64_t>(p, e); if (i) domain_first_part.push_back(fp); }
	}
	FILE* out = fopen(pos[1].c_str(), "wb");
	if (!out) die("cannot open file: " + pos[1]);
	std::vector<char> obuf(1 << 24); setvbuf(out, obuf.data(), _IOFBF, obuf.size());

	Queue<ReadPart> q_bases_for_qual, q_reads; Queue<HeaderPart> q_hdr;
	std::string err_dna, err_qual, err_hdr;
	const size_t n_parts = ar.n_parts(s_dna);
	if (is_fastq && ar.n_parts(s_qual) != n_parts) die("`dna` and `qual` streams have different numbers of parts");
	auto is_domain_start = [&](size_t part) { for (uint64_t f : domain_first_part) if (f == part) return true; return false; };

	std::thread t_dna([&]() {
		cl_dna_decoder* d = nullptr;
		if (cl_dna_decoder_create(M.max_candidates, M.level, 0, 0, M.ref_mode == 0, M.sparse_range, M.sparse_exp, &d) != CL_OK) { err_dna = "cl_dna_decoder_create"; }
		std::vector<uint8_t> in; uint64_t n_reads = 0;
		for (size_t p = 0; d && p < n_parts; ++p)
		{
			if (!ar.part(s_dna, p, in, n_reads)) { err_dna = "cannot read a `dna` part"; break; }
			if (is_domain_start(p)) cl_dna_decoder_new_domain(d);
			ReadPart rp; rp.off.resize(n_reads + 1);
			uint64_t cap = std::max<uint64_t>(in.size() * 8, 1 << 20), got = 0;
			rp.bases.resize(cap);
			cl_status s = cl_dna_decode_part(d, in.data(), in.size(), (uint32_t)n_reads, rp.bases.data(), cap, rp.off.data(), &got);
			if (s == CL_E_CAPACITY) { rp.bases.resize(got); s = cl_dna_decode_part(d, in.data(), in.size(), (uint32_t)n_reads, rp.bases.data(), got, rp.off.data(), &got); }   // the decoded part is kept inside
			if (s != CL_OK) { err_dna = cl_dna_decoder_error(d); break; }
			rp.bases.resize(got);
			if (is_fastq) { ReadPart cp; cp.bases = rp.bases; cp.off = rp.off; q_bases_for_qual.push(std::move(cp)); }
			q_reads.push(std::move(rp));
		}
		if (d) cl_dna_decoder_free(d);
		q_bases_for_qual.finish(); q_reads.finish();
	});
	Queue<ReadPart> q_quals;
	std::thread t_qual([&]() {
		if (!is_fastq) { q_quals.finish(); return; }
		cl_qual_params qp{}; qp.mode = M.qual_mode; qp.source = M.source; qp.level = M.level; qp.n_rev = (uint32_t)M.rev.size();
		for (size_t i = 0; i < M.rev.size(); ++i) qp.rev[i] = M.rev[i];
		cl_qual_decoder* q = nullptr;
		if (cl_qual_decoder_create(&qp, &q) != CL_OK) { err_qual = "cl_qual_decoder_create"; }
		ReadPart rp; std::vector<uint8_t> in; uint64_t meta = 0; size_t p = 0;
		while (q && q_bases_for_qual.pop(rp))
		{
			if (!ar.part(s_qual, p, in, meta)) { err_qual = "cannot read a `qual` part"; break; }
			if (is_domain_start(p)) cl_qual_decoder_new_domain(q);
			rp.quals.resize(rp.bases.size());
			if (cl_qual_decode_part(q, in.data(), in.size(), rp.bases.data(), rp.off.data(), (uint32_t)(rp.off.size() - 1), rp.quals.data()) != CL_OK) { err_qual = "corrupt `qual` part"; break; }
			rp.bases.clear(); rp.bases.shrink_to_fit();
			q_quals.push(std::move(rp));
			++p;
		}
		while (q_bases_for_qual.pop(rp)) {}                                   // drain after an error so that the producer can finish
		if (q) cl_qual_decoder_free(q);
		q_quals.finish();
	});
	std::thread t_hdr([&]() {
		cl_id_decoder* c = nullptr;
		if (cl_id_decoder_create(M.header_mode, &c) != CL_OK) { err_hdr = "cl_id_decoder_create"; }
		std::vector<uint8_t> in; uint64_t n = 0;
		for (size_t p = 0; c && p < ar.n_parts(s_hdr); ++p)
		{
			if (!ar.part(s_hdr, p, in, n)) { err_hdr = "cannot read a `header` part"; break; }
			HeaderPart hp; hp.off.resize(n + 1); hp.plus.resize(n);
			uint64_t cap = std::max<uint64_t>(in.size() * 64, 1 << 20), got = 0;
			hp.ids.resize(cap);
			cl_status s = cl_id_decode_part(c, in.data(), in.size(), (uint32_t)n, hp.ids.data(), cap, hp.off.data(), hp.plus.data(), &got);
			if (s == CL_E_CAPACITY) { hp.ids.resize(got); s = cl_id_decode_part(c, in.data(), in.size(), (uint32_t)n, hp.ids.data(), got, hp.off.data(), hp.plus.data(), &got); }
			if (s != CL_OK) { err_hdr = "corrupt `header` part"; break; }
			hp.ids.resize(got);
			q_hdr.push(std::move(hp));
		}
		if (c) cl_id_decoder_free(c);
		q_hdr.finish();
	});

	// writer (decompression.cpp:84-258): records in file order; the three streams are packed independently
	ReadPart rp, qp; HeaderPart hp; size_t ri = 0, hi = 0; bool have_r = false, have_h = false;
	uint64_t n_rec = 0; std::vector<char> line;
	auto next_read = [&]() { while (!have_r || ri + 1 >= rp.off.size()) { if (!q_reads.pop(rp)) return false; if (is_fastq && !q_quals.pop(qp)) return false; ri = 0; have_r = true; } return true; };
	auto next_hdr = [&]() { while (!have_h || hi + 1 >= hp.off.size()) { if (!q_hdr.pop(hp)) return false; hi = 0; have_h = true; } return true; };
	bool write_ok = true;
	for (;;)
	{
		const bool r = next_read(), h = next_hdr();
		if (!r || !h) { if (r != h && err_dna.empty() && err_qual.empty() && err_hdr.empty()) err_dna = "the streams hold different numbers of records"; break; }
		const uint64_t b0 = rp.off[ri], b1 = rp.off[ri + 1], h0 = hp.off[hi], h1 = hp.off[hi + 1];
		line.clear();
		line.push_back(is_fastq ? '@' : '>');
		line.insert(line.end(), hp.ids.begin() + h0, hp.ids.begin() + h1); line.push_back('\n');
		for (uint64_t i = b0; i < b1; ++i) line.push_back("ACGTN"[(rp.bases[i] & 7) > 4 ? 4 : (rp.bases[i] & 7)]);
		line.push_back('\n');
		if (is_fastq)
		{
			line.push_back('+');
			if (hp.plus[hi]) line.insert(line.end(), hp.ids.begin() + h0, hp.ids.begin() + h1);
			line.push_back('\n');
			line.insert(line.end(), qp.quals.begin() + b0, qp.quals.begin() + b1); line.push_back('\n');
		}
		if (fwrite(line.data(), 1, line.size(), out) != line.size()) { write_ok = false; break; }
		++ri; ++hi; ++n_rec;
	}
	// let the producers run out (after an error too), then report
	{ ReadPart x; while (q_reads.pop(x)) {} while (q_quals.pop(x)) {} HeaderPart y; while (q_hdr.pop(y)) {} }
	t_dna.join(); t_qual.join(); t_hdr.join();
	if (fflush(out) != 0 || ferror(out)) write_ok = false;
	if (fclose(out) != 0) write_ok = false;
	ar.close();
	if (!err_dna.empty()) die("dna stream: " + err_dna);
	if (!err_qual.empty()) die("qual stream: " + err_qual);
	if (!err_hdr.empty()) die("header stream: " + err_hdr);
	if (!write_ok) die("cannot write " + pos[1] + " (disk full?)");
	fprintf(stderr, "colord_hip: %llu records decompressed\n", (unsigned long long)n_rec);
	return 0;
}
