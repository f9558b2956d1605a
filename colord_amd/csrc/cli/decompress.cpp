// decompress.cpp — `colord_hip decompress in.colord out.fastq|out.fasta` and `colord_hip info in.colord`: the reference's
// runDecompression (src/colord/decompression.cpp:84-258: the FASTQ / FASTA writers) and runInfo (info.cpp:24-53) on top of the
// record stream of reader.hpp (the library's decoders behind the reference's decompression driver).  FASTA is written when the
// archive has no `qual` stream.  No GPU is needed to decompress.
#include "reader.hpp"
#include <ctime>
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

int run_decompress(int argc, char** argv)
{
	std::vector<std::string> pos; std::string genome;
	for (int i = 2; i < argc; ++i)
	{
		const std::string a = argv[i];
		if ((a == "-G" || a == "--reference-genome") && i + 1 < argc) genome = argv[++i];      // needed when the archive was written with -G but without -s
		else if (a == "-v" || a == "--verbose") ;
		else pos.push_back(a);
	}
	if (pos.size() != 2) { fprintf(stderr, "usage: colord_hip decompress [-G reference_genome.fa] archive.colord output.fastq\n"); return 1; }
	uint64_t n_rec = 0; bool write_ok = true;
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
