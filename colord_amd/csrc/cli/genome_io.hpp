// genome_io.hpp — host I/O of the reference-genome mode shared by the compressor and the record stream: the multi-FASTA reader of
// CReferenceGenome (src/colord/reference_genome.cpp:106-196, reference_genome.h:50-55) and its pseudo reads (:391-419).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <stdexcept>
#include <zlib.h>

namespace genome_io {
struct Sequences { std::vector<uint8_t> codes; std::vector<uint64_t> off{ 0 }; };      // bases 0..3 back to back, n + 1 offsets

// plain or gzip multi-FASTA; header lines start a new sequence; only A C G T (either case) are kept, everything else is dropped
inline Sequences read_fasta(const std::string& path)
{
	gzFile g = gzopen(path.c_str(), "rb");
	if (!g) throw std::runtime_error("cannot open file: " + path);
	gzbuffer(g, 1 << 22);
	std::vector<uint8_t> buf(1 << 24);
	Sequences S; bool first = true, open = false;
	enum { HEADER, SEQ, EOL_HEADER, EOL_SEQ } st = HEADER;
	auto add = [&](char c) { const char u = (char)(c & ~0x20); const int v = u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : -1; if (v >= 0) S.codes.push_back((uint8_t)v); };
	auto end_seq = [&]() { if (open) S.off.push_back(S.codes.size()); open = false; };
	for (;;)
	{
		const int n = gzread(g, buf.data(), (unsigned)buf.size());
		if (n < 0) { gzclose(g); throw std::runtime_error("read error (zlib): " + path); }
		if (n == 0) break;
		if (first) { if (buf[0] != '>') { gzclose(g); throw std::runtime_error("wrong reference genome file format, multi fasta expected"); } first = false; open = true; }
		for (int i = 0; i < n; ++i)
		{
			const char c = (char)buf[i];
			const bool eol = c == '\n' || c == '\r';
			switch (st)
			{
			case HEADER: if (eol) st = EOL_HEADER; break;
			case SEQ: if (eol) st = EOL_SEQ; else add(c); break;
			case EOL_SEQ: if (eol) break; if (c == '>') { st = HEADER; end_seq(); open = true; } else { st = SEQ; add(c); } break;
			case EOL_HEADER: if (!eol) { st = SEQ; add(c); } break;
			}
		}
	}
	gzclose(g);
	if (first) throw std::runtime_error("file " + path + " is empty");
	end_seq();
	return S;
}

// pieces of read_len bases, consecutive pieces overlapping by `overlap` (reference_genome.cpp:391-419)
inline Sequences pseudo_reads(const Sequences& G, uint32_t read_len, uint32_t overlap)
{
	if (read_len <= overlap) throw std::runtime_error("reference genome: pseudo-read length does not exceed the overlap");
	Sequences P;
	for (size_t s = 0; s + 1 < G.off.size(); ++s)
	{
		const uint64_t b = G.off[s], len = G.off[s + 1] - b;
		for (uint64_t start = 0; start < len; start += read_len - overlap)
		{
			const uint64_t end = std::min<uint64_t>(start + read_len, len);
			P.codes.insert(P.codes.end(), G.codes.begin() + b + start, G.codes.begin() + b + end);
			P.off.push_back(P.codes.size());
		}
	}
	return P;
}
} // namespace genome_io
