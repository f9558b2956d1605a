// decompress.cpp — `colord_hip decompress in.colord out.fastq|out.fasta` and `colord_hip info in.colord`: the reference's
// runDecompression (src/colord/decompression.cpp:84-258: the FASTQ / FASTA writers) and runInfo (info.cpp:24-53) on top of the
// record stream of reader.hpp (the library's decoders behind the reference's decompression driver).  FASTA is written when the
// archive has no `qual` stream.  No GPU is needed to decompress.
#include "reader.hpp"
#include <ctime>
#include <memory>
#include <algorithm>
using namespace colord_hip_reader;

int run_info(int argc, char** argv)
{
	if (argc < 3) { fprintf(stderr, "usage: colord_hip info archive.colord\n"); return 1; }
	ArchiveReader ar;
	if (!ar.open(argv[2])) die(std::string("cannot open archive: ") + argv[2]);
	std::vector<uint8_t> b; uint64_t meta = 0;
	if (!ar.part(ar.id("info"), 0, b, meta) || b.size() < 40) die("archive without a readable `info` stream");
	ArchiveInfo I;
	try { I = parse_info(b); } catch (const std::exception& e) { die(e.what()); }
	time_t t = (time_t)I.time;
	fprintf(stderr, "version major: %u\nversion minor: %u\nversion patch: %u\ntotal bytes: %llu\ntotal bases: %llu\ntotal reads: %u\ntime: %s\ncommand: %s\n",
		I.version_major, I.version_minor, I.version_patch, (unsigned long long)I.total_bytes, (unsigned long long)I.total_bases, I.total_reads, asctime(localtime(&t)), I.command_line.c_str());
	return 0;
}

// one record as the reference's writers put it out (decompression.cpp:84-258)
static void format_record(std::vector<char>& line, const Record& r, bool is_fastq)
{
	line.clear();
	line.push_back(is_fastq ? '@' : '>');
	line.insert(line.end(), r.header, r.header + r.header_len); line.push_back('\n');
	for (size_t i = 0; i < r.n_bases; ++i) line.push_back("ACGTN"[(r.bases[i] & 7) > 4 ? 4 : (r.bases[i] & 7)]);
	line.push_back('\n');
	if (is_fastq)
	{
		line.push_back('+');
		if (r.plus_is_header) line.insert(line.end(), r.header, r.header + r.header_len);
		line.push_back('\n');
		line.insert(line.end(), r.quals, r.quals + r.n_bases); line.push_back('\n');
	}
}
// Archives with INDEPENDENT model domains (`colord_hip compress-* --domains K`): every domain is decoded by a worker of its own — three
// stream threads each, as for a whole archive — into a file of its own next to the output; the files are then joined in order.  The ids
// come from one pass over the `header` stream that all workers share.  Returns the number of records, or -1 if the archive is not of
// that kind (the caller then decodes it as one stream).
static long long decompress_domains(const std::string& arc, const std::string& genome, const std::string& out_path, int max_threads)
{
	size_t K = 0; bool is_fastq = true;
	{ RecordStream probe(arc, genome); if (!probe.independent_domains() || probe.n_domains() < 2) return -1; K = probe.n_domains(); is_fastq = probe.is_fastq(); }
	HeaderCache hc;
	std::thread ht([&]() { hc.decode_all(arc); });
	std::vector<std::string> errs(K), tmp(K); std::vector<uint64_t> n_rec(K, 0);
	for (size_t d = 0; d < K; ++d) tmp[d] = out_path + ".domain" + std::to_string(d) + ".tmp";
	// the dna / qual threads of the first `max_threads` domains start at once; the ids are needed from the first record on
	std::mutex mu; size_t next_dom = 0;
	auto worker = [&]() {
		for (;;)
		{
			size_t d; { std::lock_guard<std::mutex> l(mu); if (next_dom >= K) return; d = next_dom++; }
			try
			{
				RecordStream r(arc, genome, (int)d, &hc);
				r.prefetch();
				{ static std::mutex hm; std::lock_guard<std::mutex> l(hm); if (ht.joinable()) ht.join(); }
				if (!hc.err.empty()) throw std::runtime_error("header stream: " + hc.err);
				FILE* out = fopen(tmp[d].c_str(), "wb");
				if (!out) throw std::runtime_error("cannot open file: " + tmp[d]);
				std::vector<char> obuf(1 << 22); setvbuf(out, obuf.data(), _IOFBF, obuf.size());
				std::vector<char> line; Record rec; bool ok = true;
				while (r.next(rec)) { format_record(line, rec, is_fastq); if (fwrite(line.data(), 1, line.size(), out) != line.size()) { ok = false; break; } ++n_rec[d]; }
				if (fflush(out) != 0 || ferror(out)) ok = false;
				if (fclose(out) != 0) ok = false;
				if (!ok) throw std::runtime_error("cannot write " + tmp[d] + " (disk full?)");
			}
			catch (const std::exception& e) { errs[d] = e.what(); }
		}
	};
	const size_t T = std::min<size_t>(K, (size_t)std::max(1, max_threads));
	std::vector<std::thread> th; for (size_t i = 0; i < T; ++i) th.emplace_back(worker);
	for (auto& t : th) t.join();
	if (ht.joinable()) ht.join();
	for (size_t d = 0; d < K; ++d) if (!errs[d].empty()) { for (auto& t : tmp) remove(t.c_str()); die("domain " + std::to_string(d) + ": " + errs[d]); }
	FILE* out = fopen(out_path.c_str(), "wb");
	if (!out) die("cannot open file: " + out_path);
	std::vector<char> buf(1 << 24); uint64_t total = 0; bool ok = true;
	for (size_t d = 0; d < K && ok; ++d)
	{
		FILE* in = fopen(tmp[d].c_str(), "rb");
		if (!in) { ok = false; break; }
		for (size_t got; (got = fread(buf.data(), 1, buf.size(), in)) > 0; ) if (fwrite(buf.data(), 1, got, out) != got) { ok = false; break; }
		fclose(in); remove(tmp[d].c_str());
		total += n_rec[d];
	}
	if (fflush(out) != 0 || ferror(out)) ok = false;
	if (fclose(out) != 0) ok = false;
	if (!ok) die("cannot write " + out_path + " (disk full?)");
	fprintf(stderr, "colord_hip: %llu records decompressed (%zu independent domains, %zu at a time)\n", (unsigned long long)total, K, T);
	return (long long)total;
}

int run_decompress(int argc, char** argv)
{
	std::vector<std::string> pos; std::string genome; int dom_threads = (int)std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency() / 3));
	for (int i = 2; i < argc; ++i)
	{
		const std::string a = argv[i];
		if ((a == "-G" || a == "--reference-genome") && i + 1 < argc) genome = argv[++i];      // needed when the archive was written with -G but without -s
		else if (a == "-v" || a == "--verbose") ;
		else if ((a == "-t" || a == "--threads") && i + 1 < argc) dom_threads = atoi(argv[++i]);
		else pos.push_back(a);
	}
	if (pos.size() != 2) { fprintf(stderr, "usage: colord_hip decompress [-G reference_genome.fa] archive.colord output.fastq\n"); return 1; }
	uint64_t n_rec = 0; bool write_ok = true;
	try { if (decompress_domains(pos[0], genome, pos[1], dom_threads) >= 0) return 0; } catch (const std::exception& e) { die(e.what()); }
	try
	{
		RecordStream rs(pos[0], genome);
		FILE* out = fopen(pos[1].c_str(), "wb");
		if (!out) die("cannot open file: " + pos[1]);
		std::vector<char> obuf(1 << 24); setvbuf(out, obuf.data(), _IOFBF, obuf.size());
		const bool is_fastq = rs.is_fastq();
		// writer (decompression.cpp:84-258): records in file order; the three streams are packed independently
		std::vector<char> line; Record r;
		while (rs.next(r))
		{
			format_record(line, r, is_fastq);
			if (fwrite(line.data(), 1, line.size(), out) != line.size()) { write_ok = false; break; }
			++n_rec;
		}
		if (fflush(out) != 0 || ferror(out)) write_ok = false;
		if (fclose(out) != 0) write_ok = false;
	}
	catch (const std::exception& e) { die(e.what()); }
	if (!write_ok) die("cannot write " + pos[1] + " (disk full?)");
	fprintf(stderr, "colord_hip: %llu records decompressed\n", (unsigned long long)n_rec);
	return 0;
}
