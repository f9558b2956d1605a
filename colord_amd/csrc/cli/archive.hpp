// archive.hpp — the CoLoRd archive container (src/colord/archive.cpp:92-114,170-236,268-283), host I/O of the command-line tool.
//   part   = varint(metadata) + payload;   varint(x) = 1 byte n (number of significant bytes) + n bytes big-endian
//   file   = parts ..., footer, u64-LE footer size
//   footer = varint(n_streams), per stream: name\0, varint(n_parts), varint(raw_size), per part varint(offset), varint(size)
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

[[noreturn]] inline void die(const std::string& m) { fprintf(stderr, "colord_hip: %s\n", m.c_str()); exit(1); }

struct ArchiveWriter {
	struct Part { uint64_t off, size; };
	struct Stream { std::string name; std::vector<Part> parts; };
	FILE* f = nullptr; uint64_t off = 0; std::vector<Stream> streams;
	static void varint(std::vector<uint8_t>& v, uint64_t x) { int n = 0; for (uint64_t t = x; t; t >>= 8) ++n; v.push_back((uint8_t)n); for (int i = n - 1; i >= 0; --i) v.push_back((uint8_t)(x >> (8 * i))); }
	std::vector<char> iobuf;
	// (a large stream buffer: with the bench's 64-Ki parts an archive is half a million parts of 10-20 KB, each of which went to the kernel as a
	// write of its own through stdio's 4-KB buffer — the writer thread of `colord_hip` could not keep up with the coders at 20 Gbases)
	void open(const std::string& path) { f = fopen(path.c_str(), "wb"); if (!f) die("cannot open file: " + path); iobuf.resize(32u << 20); setvbuf(f, iobuf.data(), _IOFBF, iobuf.size()); }
	int reg(const std::string& n) { streams.push_back(Stream{ n, {} }); return (int)streams.size() - 1; }
	void add(int s, const uint8_t* p, uint64_t n, uint64_t meta)
	{
		uint8_t h[9]; size_t hn = 0;                                          // varint(meta)
		{ int nb = 0; for (uint64_t t = meta; t; t >>= 8) ++nb; h[hn++] = (uint8_t)nb; for (int i = nb - 1; i >= 0; --i) h[hn++] = (uint8_t)(meta >> (8 * i)); }
		streams[s].parts.push_back(Part{ off, n });
		if (fwrite(h, 1, hn, f) != hn || (n && fwrite(p, 1, n, f) != n)) die("cannot write the archive (disk full?)");
		off += hn + n;
	}
	void close()
	{
		std::vector<uint8_t> ft; varint(ft, streams.size());
		for (auto& s : streams)
		{
			ft.insert(ft.end(), s.name.begin(), s.name.end()); ft.push_back(0);
			varint(ft, s.parts.size()); varint(ft, 0);                       // raw size: unused by these streams
			for (auto& p : s.parts) { varint(ft, p.off); varint(ft, p.size); }
		}
		const uint64_t n = ft.size();
		uint8_t sz[8]; for (int i = 0; i < 8; ++i) sz[i] = (uint8_t)(n >> (8 * i));
		// an archive without its footer is unreadable: every write is checked, the file is flushed before it counts as written
		if (fwrite(ft.data(), 1, n, f) != n || fwrite(sz, 1, 8, f) != 8 || fflush(f) != 0 || ferror(f)) die("cannot write the archive footer (disk full?)");
		if (fclose(f) != 0) die("cannot close the archive (disk full?)");
		f = nullptr;
	}
};

struct ArchiveReader {
	struct Part { uint64_t off, size; };
	struct Stream { std::string name; uint64_t raw_size = 0; std::vector<Part> parts; size_t next = 0; };
	FILE* f = nullptr; std::vector<Stream> streams; uint64_t file_size = 0;
	bool open(const std::string& path)
	{
		f = fopen(path.c_str(), "rb");
		if (!f) return false;
		const bool ok_ = parse_footer();
		if (!ok_) close();                                                 // (no descriptor left behind by a file that is not an archive)
		return ok_;
	}
	bool parse_footer()
	{
		if (fseeko(f, 0, SEEK_END) != 0) return false;
		const off_t end = ftello(f);
		if (end < 8) return false;
		const uint64_t fs = file_size = (uint64_t)end;
		uint8_t sz[8];
		if (fseeko(f, (off_t)(fs - 8), SEEK_SET) != 0 || fread(sz, 1, 8, f) != 8) return false;
		uint64_t n = 0; for (int i = 0; i < 8; ++i) n |= (uint64_t)sz[i] << (8 * i);
		if (n > fs - 8) return false;
		std::vector<uint8_t> ft(n);
		if (fseeko(f, (off_t)(fs - 8 - n), SEEK_SET) != 0 || (n && fread(ft.data(), 1, n, f) != n)) return false;
		size_t p = 0; bool ok = true;
		auto vi = [&]() -> uint64_t { if (p >= ft.size()) { ok = false; return 0; } const int k = ft[p++]; uint64_t v = 0; for (int i = 0; i < k; ++i) { if (p >= ft.size()) { ok = false; return 0; } v = (v << 8) | ft[p++]; } return v; };
		const uint64_t ns = vi();
		for (uint64_t s = 0; s < ns && ok; ++s)
		{
			Stream st;
			while (p < ft.size() && ft[p]) st.name.push_back((char)ft[p++]);
			++p;
			const uint64_t np = vi(); st.raw_size = vi();
			if (np > ft.size()) ok = false;                                  // (a part costs at least two footer bytes)
			for (uint64_t i = 0; i < np && ok; ++i)
			{	// a part lies inside the file, before the footer: offsets and sizes of a crafted footer never reach resize() / pread()
				Part pt; pt.off = vi(); pt.size = vi();
				if (pt.off > fs || pt.size > fs - pt.off) ok = false;
				st.parts.push_back(pt);
			}
			streams.push_back(std::move(st));
		}
		return ok;
	}
	int id(const std::string& name) const { for (size_t i = 0; i < streams.size(); ++i) if (streams[i].name == name) return (int)i; return -1; }
	size_t n_parts(int s) const { return s < 0 ? 0 : streams[s].parts.size(); }
	// part `i` of stream s: payload + metadata (thread-safe through pread)
	bool part(int s, size_t i, std::vector<uint8_t>& data, uint64_t& meta) const;
	void close() { if (f) fclose(f); f = nullptr; }
};
#include <unistd.h>
inline bool ArchiveReader::part(int s, size_t i, std::vector<uint8_t>& data, uint64_t& meta) const
{
	if (s < 0 || i >= streams[s].parts.size()) return false;
	const Part& pt = streams[s].parts[i];
	uint8_t h[9];
	const ssize_t got = pread(fileno(f), h, 9, (off_t)pt.off);
	if (got < 1) return false;
	const int k = h[0]; if (k > 8 || got < 1 + k) return false;
	meta = 0; for (int j = 0; j < k; ++j) meta = (meta << 8) | h[1 + j];
	if (pt.off > file_size || 1ull + k > file_size - pt.off || pt.size > file_size - pt.off - 1 - k) return false;
	data.resize(pt.size);
	uint64_t done = 0;
	while (done < pt.size)
	{
		const ssize_t r = pread(fileno(f), data.data() + done, pt.size - done, (off_t)(pt.off + 1 + k + done));
		if (r <= 0) return false;
		done += (uint64_t)r;
	}
	return true;
}
