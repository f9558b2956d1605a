// compress.cpp — `colord_hip compress-ont | compress-pbhifi | compress-pbraw [options] input output`: the compress side of the
// reference CLI (src/colord/arg_parse.cpp:455-640 options and their checks, :89-408 presets) and the host side of runCompression
// (compression.cpp:344-785): input parsing with the reader's semantics (in_reads.cpp:62-226: FASTQ / FASTA / multi-line FASTA,
// plain or gzip, CR LF tolerated, blank lines skipped, '+' line empty or equal to the id), k / anchor length from the file size,
// reader packs, the `header`, `meta` and `info` streams and the archive container.  Everything between read bases / qualities
// and the `dna` / `qual` parts is the chunked compressor of the library (cl_compressor_*, csrc/stream.hip): the input is cut in
// chunks of whole reader packs (--chunk-bases, default 1 Gbase) that stay resident in HBM as 2-bit arenas + quality bytes for the
// three passes, so the file is parsed once and any size the GPU holds (~150 Gbases of FASTQ on 288 GB) is one run.
#include "colord_hip.h"
#include "archive.hpp"
#include "genome_io.hpp"
#include "transport.hpp"
#include <hip/hip_runtime_api.h>
#include <zlib.h>
#include <algorithm>
#include <chrono>
#include <ctime>
#include <thread>
#include <sys/mman.h>
#include <fcntl.h>
#include <unistd.h>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>

namespace {
struct Preset { int level; uint32_t ci, cs, f, c, max_rec, min_part_alt; int qual_mode; int sparse; double g; };
// arg_parse.cpp:89-408 — [source][priority]: ratio, balanced, memory (memory is the default priority)
const Preset PRESETS[3][3] = {
	{ { 3, 2, 120, 8, 10, 6, 48, 2, 0, 1 }, { 2, 3, 100, 9, 8, 5, 48, 2, 1, 2 }, { 1, 4, 80, 12, 5, 3, 64, 2, 1, 1 } },          // ONT, 4-avg qualities
	{ { 3, 2, 120, 8, 10, 6, 48, 8, 0, 1 }, { 2, 3, 100, 9, 8, 5, 48, 8, 1, 2 }, { 1, 4, 80, 12, 5, 3, 64, 8, 1, 1 } },          // PBRaw, qualities dropped
	{ { 3, 2, 150, 20, 12, 6, 48, 1, 0, 1 }, { 2, 3, 120, 30, 10, 5, 48, 1, 1, 6 }, { 2, 3, 100, 40, 8, 5, 48, 1, 1, 3 } },       // PBHiFi, 5-avg qualities
};
// default -T / -D values of the quality modes (arg_parse.cpp:32-84,410-450): mode -> forward thresholds, decoder representatives
struct QDef { std::vector<uint32_t> fwd, rev; };
QDef qual_defaults(int mode)
{
	switch (mode)
	{
	case 1: return { { 7, 14, 26, 93 }, {} };
	case 2: return { { 7, 14, 26 }, {} };
	case 3: return { { 7 }, {} };
	case 4: return { { 7, 14, 26, 93 }, { 3, 10, 18, 35, 93 } };
	case 5: return { { 7, 14, 26 }, { 3, 10, 18, 35 } };
	case 6: return { { 7 }, { 1, 13 } };
	case 8: return { {}, { 0 } };
	default: return { {}, {} };
	}
}
int qual_mode_of(const std::string& s)      // QualityComprMode (params.h:33-43)
{
	static const char* names[] = { "org", "5-avg", "4-avg", "2-avg", "5-fix", "4-fix", "2-fix", "avg", "none" };
	for (int i = 0; i < 9; ++i) if (s == names[i]) return i;
	return -1;
}
void hipck(hipError_t e, const char* what) { if (e != hipSuccess) die(std::string(what) + ": " + hipGetErrorString(e)); }
void ck(cl_ctx* ctx, cl_status s, const char* what) { if (s != CL_OK) die(std::string(what) + ": " + (ctx ? cl_last_error(ctx) : "error")); }
template<class T> void le(std::vector<uint8_t>& v, T x) { for (size_t i = 0; i < sizeof(T); ++i) v.push_back((uint8_t)((uint64_t)x >> (8 * i))); }
void le_double(std::vector<uint8_t>& v, double d) { uint64_t u; memcpy(&u, &d, 8); le(v, u); }
std::vector<uint32_t> list_u32(const std::string& s) { std::vector<uint32_t> v; size_t p = 0; while (p < s.size()) { size_t e = s.find_first_of(", ", p); if (e == std::string::npos) e = s.size(); if (e > p) v.push_back((uint32_t)strtoul(s.substr(p, e - p).c_str(), nullptr, 10)); p = e + 1; } return v; }

struct Options {
	int source = 0, prio = 2, gpu = 0; bool verbose = false;
	uint32_t k = 0, a = 0; std::string in, out, genome; bool store_genome = false;
	long ci = -1, cs = -1, f = -1, c = -1, max_rec = -1, min_to_alt = -1, min_anchors = 1;
	double cost_mult = 1.0, frac_min = 0.5, frac_always = 0.9, max_matches_mult = 10.0, g = -1, exponent = 1.0;
	int qual_mode = -1, header_mode = 0, ref_mode = -1;
	std::vector<uint32_t> T, D; bool has_T = false, has_D = false;
	double chunk_bases = 1.0e9; bool chunk_bases_set = false;
	uint64_t part_symbols = 2u << 21;               // --part-symbols: the coder parts close once their reads (+ 1 guard each) reach this; default = the reader packs (defs.h:45)
	int parse_threads = 0;                          // --parse-threads (0: as many as the host offers, at most 32)
	bool stream_input = false;                      // --stream-input: the input is read three times (k-mers, reference reads, coding) and only a window of chunks is resident in HBM
	int domains = 1;                                // --domains K: K INDEPENDENT model domains on one GPU (own k-mer set, references, index, models each): decoded side by side
	int gpus = 1; std::vector<int> gpu_list; std::string transport = "rccl";   // --gpus N [--gpu-list a,b,..] [--transport rccl|host]: reads sharded over N GPUs (run_compress_multi)
};

// ---- input: one sequential pass that finds lines (memchr) and assigns them their role; bases / qualities / ids are appended to the
// ---- chunk under construction.  A chunk closes at the first reader-pack boundary at or after chunk_bases.
struct Chunk {
	uint8_t* bases = nullptr; uint8_t* quals = nullptr; uint64_t cap = 0, n = 0;       // pinned staging (ASCII)
	bool pinned = true;                                                                // (false: plain host memory — `parse-check`, which needs no GPU)
	uint8_t* get(uint64_t bytes) { uint8_t* p = nullptr; if (pinned) hipck(hipHostMalloc((void**)&p, bytes, hipHostMallocDefault), "hipHostMalloc"); else { p = (uint8_t*)malloc(bytes); if (!p) die("out of memory"); } return p; }
	void give(uint8_t* p) { if (!p) return; if (pinned) (void)hipHostFree(p); else free(p); }
	std::vector<uint64_t> off{ 0 }; std::vector<uint32_t> packs{ 0 }; uint64_t pack_acc = 0;
	std::vector<uint32_t> parts{ 0 }; uint64_t part_acc = 0;                           // coder parts (--part-symbols); == packs by default
	void reserve(uint64_t need, bool with_quals)
	{
		if (need <= cap) return;
		// (chunks close at the first pack boundary at or after their target: the ones to come are a few MB larger or smaller than the first, and
		// pinning a gigabyte takes 0.1-0.3 s — exact first sizes meant a second, larger pair of buffers a few chunks later: 2.7 s of the
		// reader's 3.1 s at 20 Gbases)
		uint64_t nc = std::max<uint64_t>(need + need / 32 + (16ull << 20), cap + cap / 2 + (1ull << 24));
		uint8_t* nb = get(nc);
		if (n) memcpy(nb, bases, n);
		give(bases);
		bases = nb;
		if (with_quals) { uint8_t* nq = get(nc); if (n) memcpy(nq, quals, n); give(quals); quals = nq; }
		cap = nc;
	}
	// the range of the quality bytes, when whoever filled the chunk has looked (the indexed reader's copy threads do, while the bytes pass
	// through their caches: the check used to be one thread's loop over a gigabyte per chunk, on the thread that feeds the GPU)
	uint8_t qlo = 255, qhi = 0; bool q_range = false;
	void clear() { n = 0; off.assign(1, 0); packs.assign(1, 0); pack_acc = 0; parts.assign(1, 0); part_acc = 0; qlo = 255; qhi = 0; q_range = false; }
	// (Phred+33 0..95: anything else would index past the coder's tables) — false: the input is refused
	bool quals_in_range(int threads = 8)
	{
		if (!quals || !n) return true;
		if (!q_range)
		{
			const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)threads, n >> 22));
			std::vector<uint8_t> lo(T, 255), hi(T, 0); std::vector<std::thread> th;
			for (int i = 0; i < T; ++i) th.emplace_back([&, i]() {
				uint8_t a = 255, b = 0; const uint8_t* q = quals;
				for (uint64_t x = n * (uint64_t)i / T, e = n * (uint64_t)(i + 1) / T; x < e; ++x) { a = q[x] < a ? q[x] : a; b = q[x] > b ? q[x] : b; }
				lo[i] = a; hi[i] = b;
			});
			for (auto& t : th) t.join();
			for (int i = 0; i < T; ++i) { qlo = std::min(qlo, lo[i]); qhi = std::max(qhi, hi[i]); }
			q_range = true;
		}
		return qlo >= 33 && qhi <= 33 + 95;
	}
	void release() { give(bases); give(quals); bases = quals = nullptr; cap = 0; }
};
struct Reader {
	gzFile g = nullptr; bool gz = false, fastq = true; uint64_t file_bytes = 0, total_bytes = 0, header_symbols = 0;
	// plain FASTQ: the file is mapped and its lines go straight from the mapping into the pinned chunk buffers (one copy; the
	// generic path below copies every byte three times through zlib's buffer and a line string)
	const uint8_t* map = nullptr; const uint8_t* mp = nullptr; const uint8_t* me = nullptr;
	std::vector<uint8_t> buf; size_t pos = 0, len = 0; bool eof = false;
	std::string line[4]; int which = 0;                          // FASTQ record under construction
	std::string fa_header, fa_seq; int fa_state = 0;            // FASTA: 0 header, 1 EOLs after header, 2 read, 3 EOLs after / inside read
	std::vector<uint8_t> ids, plus; std::vector<uint64_t> id_off{ 0 };
	uint64_t n_reads = 0, n_bases = 0;
	uint64_t part_symbols = 2u << 21;
	bool replay = false;                                         // a later pass over the same input (--stream-input): ids, counters and checks are those of the first
	// back to the first record: the chunks come again exactly as in the first pass (the reference reads its input twice as well,
	// compression.cpp:432,547-561)
	void rewind()
	{
		replay = true;
		if (map) { mp = map; rec_pos = 0; return; }
		if (gzrewind(g) != 0) die("cannot rewind the input");
		pos = len = 0; eof = false; which = 0; for (auto& l : line) l.clear();
		fa_header.clear(); fa_seq.clear(); fa_state = 0;
	}
	// plain FASTQ, several threads: the mapping is cut into byte ranges at record starts, every range is indexed by a thread of its own
	// (line ends by memchr, the reader's checks), then the chunks are filled from the index by parallel copies (index_mapped below)
	struct Rec { const uint8_t* id; const uint8_t* seq; const uint8_t* qual; uint32_t id_len, len; uint8_t plus_eq; };
	std::vector<Rec> recs; size_t rec_pos = 0; bool indexed = false; int threads = 1;
	double t_book = 0, t_copy = 0;                               // (-v: bookkeeping on the reader's thread, parallel copies)
	void open(const std::string& path)
	{
		FILE* probe = fopen(path.c_str(), "rb");
		if (!probe) die("cannot open file: " + path);
		unsigned char mg[2] = { 0, 0 }; const size_t got = fread(mg, 1, 2, probe);
		fseeko(probe, 0, SEEK_END); file_bytes = (uint64_t)ftello(probe); fclose(probe);
		gz = got == 2 && mg[0] == 0x1f && mg[1] == 0x8b;
		g = gzopen(path.c_str(), "rb");
		if (!g) die("cannot open file: " + path);
		gzbuffer(g, 1 << 22);
		buf.resize(1 << 25);
		fill();
		if (!len) die("file " + path + " is empty");
		if (buf[0] != '@' && buf[0] != '>') die("unknown file format (the first character must be '@' or '>')");      // in_reads.cpp:256-262
		fastq = buf[0] == '@';
		if (!gz && fastq && file_bytes && !getenv("COLORD_HIP_NO_MMAP"))
		{
			const int fd = ::open(path.c_str(), O_RDONLY);
			if (fd >= 0)
			{
				void* m = mmap(nullptr, file_bytes, PROT_READ, MAP_PRIVATE, fd, 0);
				::close(fd);
				if (m != MAP_FAILED) { (void)madvise(m, file_bytes, MADV_SEQUENTIAL); map = mp = (const uint8_t*)m; me = map + file_bytes; total_bytes = file_bytes; }
			}
		}
	}
	// one line of the mapping: [a, b) without its end-of-line characters; lines end at '\n' or '\r', empty lines are skipped (in_reads.cpp:188-226)
	bool map_line(const uint8_t*& a, const uint8_t*& b, bool short_line)
	{
		while (mp < me && (*mp == '\n' || *mp == '\r')) ++mp;
		if (mp >= me) return false;
		a = mp;
		const uint8_t* q = (const uint8_t*)memchr(mp, '\n', (size_t)(me - mp));
		b = q ? q : me;
		mp = q ? q + 1 : me;
		if (b > a && b[-1] == '\r') --b;
		if (short_line) { const uint8_t* r = (const uint8_t*)memchr(a, '\r', (size_t)(b - a)); if (r) { mp = r + 1; b = r; } }   // (a lone '\r' ends a line too; in a sequence or quality line it is refused as a symbol / quality value)
		return true;
	}
	// One record at cursor `c` of the mapping, with the reader's rules (in_reads.cpp:79-92,188-226); "" = fine, else the reader's complaint.
	static const char* parse_record(const uint8_t*& c, const uint8_t* end, Rec& r, uint64_t& hdr_syms, bool& got)
	{
		auto line = [&](const uint8_t*& a, const uint8_t*& b, bool short_line) -> bool {
			while (c < end && (*c == '\n' || *c == '\r')) ++c;
			if (c >= end) return false;
			a = c;
			const uint8_t* q = (const uint8_t*)memchr(c, '\n', (size_t)(end - c));
			b = q ? q : end;
			c = q ? q + 1 : end;
			if (b > a && b[-1] == '\r') --b;
			if (short_line) { const uint8_t* rr = (const uint8_t*)memchr(a, '\r', (size_t)(b - a)); if (rr) { c = rr + 1; b = rr; } }
			return true;
		};
		const uint8_t *h0, *h1, *s0, *s1, *p0, *p1, *q0, *q1;
		got = false;
		if (!line(h0, h1, true)) return "";
		if (!line(s0, s1, false) || !line(p0, p1, true) || !line(q0, q1, false)) return "truncated FASTQ record at the end of the input";
		if (*h0 != '@') return "FASTQ record does not start with '@'";
		if (*p0 != '+') return "FASTQ record without '+' line";
		if (s1 - s0 != q1 - q0) return "sequence and quality lengths differ";
		const bool eq = p1 - p0 > 1;
		if (eq && ((p1 - p0) != (h1 - h0) || memcmp(p0 + 1, h0 + 1, (size_t)(h1 - h0 - 1)) != 0)) return "quality header not empty but different than read header";
		if ((uint64_t)(s1 - s0) >= (1ull << 32) || (uint64_t)(h1 - h0) >= (1ull << 32)) return "line longer than 4 Gi symbols";
		hdr_syms += (uint64_t)(h1 - h0) + (uint64_t)(p1 - p0);
		r = Rec{ h0 + 1, s0, q0, (uint32_t)(h1 - h0 - 1), (uint32_t)(s1 - s0), (uint8_t)(eq ? 1 : 0) };
		got = true;
		return "";
	}
	// Index of the whole mapping by `threads` threads.  A range starts at the first line at or after its byte offset that begins with
	// '@', is followed two lines later by a '+' line and whose sequence and quality lines are equally long.  That is a guess (a quality
	// line may begin with '@'), so it is VERIFIED: the thread before must end its last record exactly there.  Any complaint or
	// mismatch: the index is dropped and the sequential reader (which reports errors in file order) takes over.
	bool index_mapped()
	{
		const int T = threads;
		const char* mn = getenv("COLORD_HIP_INDEX_MIN_BYTES");                        // (tests index small files too)
		if (T < 2 || (uint64_t)(me - map) < (mn ? strtoull(mn, nullptr, 10) : (64ull << 20))) return false;
		std::vector<const uint8_t*> b((size_t)T + 1, me);
		b[0] = map;
		for (int i = 1; i < T; ++i)
		{
			const uint8_t* p = map + (uint64_t)(me - map) * i / T;
			const uint8_t* q = (const uint8_t*)memchr(p, '\n', (size_t)(me - p));
			const uint8_t* found = nullptr;
			for (int tries = 0; q && tries < 64 && !found; ++tries)
			{
				const uint8_t* c = q + 1;
				while (c < me && (*c == '\n' || *c == '\r')) ++c;
				if (c >= me) break;
				if (*c == '@')
				{
					const uint8_t* cc = c; Rec r; uint64_t hs = 0; bool got = false;
					if (parse_record(cc, me, r, hs, got)[0] == 0 && got) { const uint8_t* n2 = cc; while (n2 < me && (*n2 == '\n' || *n2 == '\r')) ++n2; if (n2 >= me || *n2 == '@') found = c; }
				}
				q = (const uint8_t*)memchr(c, '\n', (size_t)(me - c));
			}
			if (!found) return false;
			b[i] = found;
		}
		for (int i = 1; i <= T; ++i) if (b[i] < b[i - 1]) return false;
		std::vector<std::vector<Rec>> part((size_t)T); std::vector<uint64_t> hs((size_t)T, 0); std::vector<int> bad((size_t)T, 0);
		std::vector<std::thread> th;
		for (int i = 0; i < T; ++i) th.emplace_back([&, i]() {
			const uint8_t* c = b[i]; const uint8_t* const stop = b[i + 1];
			part[i].reserve((size_t)((stop - c) / 20000 + 1024));
			for (;;)
			{
				while (c < me && (*c == '\n' || *c == '\r')) ++c;                       // (blank lines between records belong to nobody)
				if (c >= stop) break;
				Rec r; bool got = false;
				if (parse_record(c, me, r, hs[i], got)[0] != 0) { bad[i] = 1; return; }
				if (!got) break;
				part[i].push_back(r);
			}
			while (c < me && (*c == '\n' || *c == '\r')) ++c;
			const uint8_t* want = stop; while (want < me && (*want == '\n' || *want == '\r')) ++want;
			if (c != want) bad[i] = 1;                                                  // the next range does not begin where this one's last record ends
		});
		for (auto& t : th) t.join();
		for (int i = 0; i < T; ++i) if (bad[i]) return false;
		size_t total = 0; for (auto& v : part) total += v.size();
		recs.reserve(total);
		for (int i = 0; i < T; ++i) { recs.insert(recs.end(), part[i].begin(), part[i].end()); header_symbols += hs[i]; std::vector<Rec>().swap(part[i]); }
		indexed = true;
		return true;
	}
	// a chunk from the index: the bookkeeping (offsets, packs, parts, ids) in file order on this thread, the bases and qualities by parallel copies
	bool next_chunk_indexed(Chunk& ch, uint64_t target)
	{
		ch.clear();
		const size_t first = rec_pos;
		const auto tb0 = std::chrono::steady_clock::now();
		auto chunk_full = [&]() { return ch.n >= target && ch.pack_acc == 0 && ch.off.size() > 1; };
		while (rec_pos < recs.size() && !chunk_full())
		{
			const Rec& r = recs[rec_pos++];
			if (!replay) { ids.insert(ids.end(), r.id, r.id + r.id_len); id_off.push_back(ids.size()); plus.push_back(r.plus_eq); ++n_reads; n_bases += r.len; }
			ch.n += r.len; ch.off.push_back(ch.n);
			close_bounds(ch, r.len);
		}
		finish_bounds(ch);
		if (ch.off.size() <= 1) return false;
		{ const uint64_t total = ch.n; ch.n = 0; ch.reserve(total + 1, true); ch.n = total; }     // (nothing to carry over: the buffers are filled below)
		const size_t cnt = rec_pos - first; const int T = (int)std::min<size_t>((size_t)threads, std::max<size_t>(1, cnt / 256));
		const auto tb1 = std::chrono::steady_clock::now(); t_book += std::chrono::duration<double>(tb1 - tb0).count();
		std::vector<std::thread> th; std::vector<uint8_t> qmin(T, 255), qmax(T, 0);
		for (int i = 0; i < T; ++i) th.emplace_back([&, i]() {
			// (equal shares of the chunk's bytes: the offsets are ascending)
			const uint64_t lo_b = ch.n * (uint64_t)i / T, hi_b = ch.n * (uint64_t)(i + 1) / T;
			size_t lo = (size_t)(std::lower_bound(ch.off.begin(), ch.off.end() - 1, lo_b) - ch.off.begin());
			size_t hi = i + 1 == T ? cnt : (size_t)(std::lower_bound(ch.off.begin(), ch.off.end() - 1, hi_b) - ch.off.begin());
			uint8_t a = 255, b = 0;
			for (size_t x = lo; x < hi; ++x)
			{
				const Rec& r = recs[first + x];
				memcpy(ch.bases + ch.off[x], r.seq, r.len); memcpy(ch.quals + ch.off[x], r.qual, r.len);      // (pread() instead of the mapping: 0.8 against 0.5 s per 20 Gbases, profiles/r06_m_*)
				const uint8_t* q = (const uint8_t*)r.qual;                           // (the range of the quality bytes while they are in this core's cache)
				for (uint32_t y = 0; y < r.len; ++y) { a = q[y] < a ? q[y] : a; b = q[y] > b ? q[y] : b; }
			}
			qmin[i] = a; qmax[i] = b;
		});
		for (auto& t : th) t.join();
		t_copy += std::chrono::duration<double>(std::chrono::steady_clock::now() - tb1).count();
		for (int i = 0; i < T; ++i) { ch.qlo = std::min(ch.qlo, qmin[i]); ch.qhi = std::max(ch.qhi, qmax[i]); }
		ch.q_range = true;
		return true;
	}
	// pack / part bookkeeping of one more read of `len` symbols: a pack closes once its reads (with one guard byte each) reach 4 Mi
	// symbols (in_reads.cpp:62-77); the coder parts likewise at --part-symbols
	void close_bounds(Chunk& ch, uint64_t len)
	{
		ch.pack_acc += len + 1;
		if (ch.pack_acc >= (2u << 21)) { ch.packs.push_back((uint32_t)(ch.off.size() - 1)); ch.pack_acc = 0; }
		ch.part_acc += len + 1;
		if (ch.part_acc >= part_symbols) { ch.parts.push_back((uint32_t)(ch.off.size() - 1)); ch.part_acc = 0; }
	}
	void finish_bounds(Chunk& ch)
	{
		if (ch.off.size() > 1 && ch.packs.back() != ch.off.size() - 1) { ch.packs.push_back((uint32_t)(ch.off.size() - 1)); ch.pack_acc = 0; }
		if (ch.off.size() > 1 && ch.parts.back() != ch.off.size() - 1) { ch.parts.push_back((uint32_t)(ch.off.size() - 1)); ch.part_acc = 0; }
		if (part_symbols == (2u << 21)) ch.parts = ch.packs;
	}
	bool next_chunk_mapped(Chunk& ch, uint64_t target)
	{
		if (indexed) return next_chunk_indexed(ch, target);
		ch.clear();
		auto chunk_full = [&]() { return ch.n >= target && ch.pack_acc == 0 && ch.off.size() > 1; };
		while (!chunk_full())
		{
			const uint8_t *h0, *h1, *s0, *s1, *p0, *p1, *q0, *q1;
			if (!map_line(h0, h1, true)) break;
			if (!map_line(s0, s1, false) || !map_line(p0, p1, true) || !map_line(q0, q1, false)) die("truncated FASTQ record at the end of the input");
			if (*h0 != '@') die("FASTQ record does not start with '@'");
			if (*p0 != '+') die("FASTQ record without '+' line");
			if (s1 - s0 != q1 - q0) die("sequence and quality lengths differ");
			if (!replay) header_symbols += (uint64_t)(h1 - h0) + (uint64_t)(p1 - p0);
			const bool eq = p1 - p0 > 1;
			if (eq && ((p1 - p0) != (h1 - h0) || memcmp(p0 + 1, h0 + 1, (size_t)(h1 - h0 - 1)) != 0)) die("quality header not empty but different than read header");   // in_reads.cpp:79-92
			add_record(ch, (const char*)h0 + 1, (size_t)(h1 - h0 - 1), (const char*)s0, (size_t)(s1 - s0), (const char*)q0, eq);
		}
		finish_bounds(ch);
		return ch.off.size() > 1;
	}
	void fill() { const int n = gzread(g, buf.data(), (unsigned)buf.size()); if (n < 0) die("read error (zlib)"); len = (size_t)n; pos = 0; if (!replay) total_bytes += len; if (!n) eof = true; }
	void add_record(Chunk& ch, const char* id, size_t id_len, const char* seq, size_t seq_len, const char* qual, bool plus_eq)
	{
		if (!replay) { ids.insert(ids.end(), id, id + id_len); id_off.push_back(ids.size()); plus.push_back(plus_eq ? 1 : 0); ++n_reads; n_bases += seq_len; }
		ch.reserve(ch.n + seq_len + 1, fastq);
		memcpy(ch.bases + ch.n, seq, seq_len);
		if (fastq) memcpy(ch.quals + ch.n, qual, seq_len);
		ch.n += seq_len; ch.off.push_back(ch.n);
		close_bounds(ch, seq_len);
	}
	void flush_fastq(Chunk& ch)
	{
		if (line[0].empty() || line[0][0] != '@') die("FASTQ record does not start with '@'");
		if (line[2].empty() || line[2][0] != '+') die("FASTQ record without '+' line");
		if (line[1].size() != line[3].size()) die("sequence and quality lengths differ");
		if (!replay) header_symbols += line[0].size() + line[2].size();
		const bool eq = line[2].size() > 1;
		if (eq && line[2].compare(1, std::string::npos, line[0], 1, std::string::npos) != 0) die("quality header not empty but different than read header");   // in_reads.cpp:79-92
		add_record(ch, line[0].data() + 1, line[0].size() - 1, line[1].data(), line[1].size(), line[3].data(), eq);
	}
	void flush_fasta(Chunk& ch)
	{
		if (!replay) header_symbols += fa_header.size();
		add_record(ch, fa_header.data() + 1, fa_header.size() - 1, fa_seq.data(), fa_seq.size(), nullptr, false);
		fa_header.clear(); fa_seq.clear();
	}
	// fills `ch` up to the first pack boundary at or after `target` bases; returns false when the input is exhausted and ch is empty
	bool next_chunk(Chunk& ch, uint64_t target)
	{
		if (map) return next_chunk_mapped(ch, target);
		ch.clear();
		auto chunk_full = [&]() { return ch.n >= target && ch.pack_acc == 0 && ch.off.size() > 1; };
		while (!eof && !chunk_full())
		{
			if (pos >= len) { fill(); if (eof) break; }
			if (fastq)
			{	// lines end at '\n' or '\r'; empty lines are skipped (in_reads.cpp:188-226)
				const uint8_t* p = buf.data() + pos; const uint8_t* e = buf.data() + len;
				const uint8_t* q = (const uint8_t*)memchr(p, '\n', (size_t)(e - p)); const uint8_t* lim = q ? q : e;
				const uint8_t* r = (const uint8_t*)memchr(p, '\r', (size_t)(lim - p)); const uint8_t* nl = r ? r : lim;
				line[which].append((const char*)p, (size_t)(nl - p));
				pos = (size_t)(nl - buf.data());
				if (nl < e)
				{
					++pos;
					if (!line[which].empty()) { if (++which == 4) { flush_fastq(ch); which = 0; for (auto& l : line) l.clear(); } }
				}
			}
			else
			{	// porcessFastaOrMultiFasta (in_reads.cpp:114-178)
				for (; pos < len && !chunk_full(); ++pos)
				{
					const uint8_t s = buf[pos]; const bool eol = s == '\n' || s == '\r';
					switch (fa_state)
					{
					case 0: if (eol) fa_state = 1; else fa_header.push_back((char)s); break;
					case 1: if (!eol) { fa_seq.push_back((char)s); fa_state = 2; } break;
					case 2: if (eol) fa_state = 3; else fa_seq.push_back((char)s); break;
					case 3: if (!eol) { if (s == '>') { flush_fasta(ch); fa_state = 0; fa_header.push_back((char)s); } else { fa_state = 2; fa_seq.push_back((char)s); } } break;
					}
				}
			}
		}
		if (eof)
		{
			if (fastq) { if (!line[which].empty()) { if (++which == 4) { flush_fastq(ch); which = 0; for (auto& l : line) l.clear(); } } if (which != 0) die("truncated FASTQ record at the end of the input"); }
			else if (!fa_header.empty()) flush_fasta(ch);
		}
		finish_bounds(ch);
		return ch.off.size() > 1;
	}
};
// the `meta` stream (compression.cpp:704-779) and the `info` stream (utils.cpp:326-342): one packing for the single- and the multi-GPU host
struct MetaIn { uint32_t n_reads, n_pseudo, tot_ref, c; int level, source; uint64_t mean_read_len; bool with_qual; int qual_mode; std::vector<uint32_t> qual_rev; int header_mode; bool sparse; uint32_t sparse_range; double exponent;
                bool with_genome, store_genome; uint32_t genome_read_len, genome_overlap; const uint8_t* genome_md5; };
std::vector<uint8_t> pack_meta(const MetaIn& M)
{
	std::vector<uint8_t> meta;
	le<uint32_t>(meta, M.tot_ref); le<uint32_t>(meta, M.c); le<int32_t>(meta, M.level); meta.push_back((uint8_t)M.source);
	le<uint64_t>(meta, (uint64_t)M.n_reads * M.mean_read_len);
	if (M.with_qual)
	{
		meta.push_back((uint8_t)M.qual_mode);
		if (M.qual_mode == 8 || (M.qual_mode >= 4 && M.qual_mode <= 6)) for (uint32_t v : M.qual_rev) le<uint32_t>(meta, v);
	}
	meta.push_back((uint8_t)M.header_mode);
	meta.push_back(M.sparse ? 1 : 0);                                    // ReferenceReadsMode: All = 0, Sparse = 1
	if (M.sparse) { le<uint32_t>(meta, M.sparse_range); le_double(meta, M.exponent); }
	meta.push_back(M.with_genome ? 1 : 0);                               // compression.cpp:764-777
	if (M.with_genome)
	{
		meta.push_back(M.store_genome ? 1 : 0);
		le<uint32_t>(meta, M.genome_read_len); le<uint32_t>(meta, M.genome_overlap); le<uint32_t>(meta, M.n_pseudo);
		if (!M.store_genome) meta.insert(meta.end(), M.genome_md5, M.genome_md5 + 16);       // the decompressor will ask for the same genome (md5 of its packed sequences)
	}
	return meta;
}
std::vector<uint8_t> pack_info(uint64_t file_bytes, uint64_t total_bases, uint32_t n_reads, int argc, char** argv)
{
	std::vector<uint8_t> inf;
	le<uint32_t>(inf, 1); le<uint32_t>(inf, 2); le<uint32_t>(inf, 1);                        // archive format of CoLoRd 1.2.1 (defs.h:24-26)
	le<uint64_t>(inf, file_bytes); le<uint64_t>(inf, total_bases); le<uint32_t>(inf, n_reads); le<uint64_t>(inf, (uint64_t)time(nullptr));
	std::string cmd; for (int i = 0; i < argc; ++i) { if (i) cmd += ' '; cmd += argv[i]; }
	le<uint32_t>(inf, (uint32_t)cmd.size()); inf.insert(inf.end(), cmd.begin(), cmd.end());
	return inf;
}
// the `header` stream of ids [0, n) of a reader (CEntrComprHeaders, entr_header.cpp:23-45): packs of >= 4 Mi id bytes (in_reads.cpp:50-56,93-101)
void code_headers(const Reader& R, uint32_t n, int header_mode, std::vector<std::vector<uint8_t>>& parts, std::vector<uint32_t>& counts, std::string& err)
{
	cl_id_coder* idc = nullptr;
	if (cl_id_coder_create(header_mode, &idc) != CL_OK) { err = "cl_id_coder_create"; return; }
	uint32_t i = 0;
	while (i < n)
	{
		uint32_t j = i; uint64_t acc = 0;
		while (j < n) { acc += R.id_off[j + 1] - R.id_off[j]; ++j; if (acc >= (2u << 21)) break; }
		std::vector<uint64_t> off(j - i + 1);
		for (uint32_t t = i; t <= j; ++t) off[t - i] = R.id_off[t] - R.id_off[i];
		std::vector<uint8_t> out(2 * (size_t)off.back() + 64); uint64_t got = 0;
		if (cl_id_encode_part(idc, R.ids.data() + R.id_off[i], off.data(), R.plus.data() + i, j - i, out.data(), out.size(), &got) != CL_OK) { err = cl_id_coder_error(idc); break; }
		out.resize(got); parts.push_back(std::move(out)); counts.push_back(j - i);
		i = j;
	}
	cl_id_coder_free(idc);
}
// Device buffers of chunks that come and go (--stream-input, one or several ranks): kept and handed out again instead of a hipMalloc + hipFree
// per chunk — hipFree waits for the WHOLE device, i.e. for the lanes and the preparation working ahead on the chunks after.
struct DevCache {
	std::vector<std::pair<void*, uint64_t>> idle; std::mutex mu;
	void* get(uint64_t bytes, uint64_t& cap)
	{
		std::lock_guard<std::mutex> l(mu);
		size_t best = idle.size();
		for (size_t i = 0; i < idle.size(); ++i) if (idle[i].second >= bytes && (best == idle.size() || idle[i].second < idle[best].second)) best = i;
		if (best != idle.size()) { void* p = idle[best].first; cap = idle[best].second; idle.erase(idle.begin() + (long)best); return p; }
		void* p = nullptr; cap = bytes + bytes / 8 + 4096;                       // (a little room: the chunks are alike, not equal)
		if (hipMalloc(&p, cap) != hipSuccess) { (void)hipGetLastError(); cap = bytes; if (hipMalloc(&p, cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; } }
		return p;
	}
	void put(void* p, uint64_t cap) { std::lock_guard<std::mutex> l(mu); if (p) idle.emplace_back(p, cap); }
	void clear() { std::lock_guard<std::mutex> l(mu); for (auto& x : idle) (void)hipFree(x.first); idle.clear(); }
};
struct DevChunk { cl_reads* reads = nullptr; uint8_t* d_quals = nullptr; uint64_t* d_off = nullptr; std::vector<uint32_t> packs, parts; uint64_t n_bases = 0; uint32_t n_reads = 0; uint64_t quals_cap = 0, off_cap = 0; };
} // namespace

static void usage()
{
	fprintf(stderr,
		"usage: colord_hip compress-ont|compress-pbhifi|compress-pbraw [options] input.fastq|fasta[.gz] output.colord\n"
		"       colord_hip decompress archive.colord output.fastq\n       colord_hip info archive.colord\n"
		"options (as the reference, arg_parse.cpp:455-640):\n"
		"  -p,--priority ratio|balanced|memory   -k,--kmer-len K with -a,--anchor-len A (both or none)\n"
		"  -q,--qual org|none|avg|2-fix|4-fix|5-fix|2-avg|4-avg|5-avg   -T,--qual-thresholds a,b,..   -D,--qual-values a,b,..\n"
		"  -i,--identifier org|main|none   -c,--max-candidates N   -L,--Lowest-count N   -H,--Highest-count N   -f,--filter-modulo N\n"
		"  -e,--edit-script-mult X   -r,--max-recurence-level N   --min-to-alt N   --min-mmer-frac X   --min-mmer-force-enc X\n"
		"  --max-matches-mult X   --min-anchors N   -R,--Ref-reads-mode all|sparse   -g,--sparse-range X   -x,--sparse-exponent X\n"
		"  -t,--threads N (accepted; the data path runs on the GPU)   -v,--verbose   --gpu N   --chunk-bases X\n"
		"  --part-symbols N   coder parts of N symbols instead of the reference's 4194304 (defs.h:45): same FASTQ back from either\n"
		"                     decompressor, 8 more bytes per part, far shorter interval-coder chains (65536: +0.04 %% size, 1.4x the speed)\n"
		"  --parse-threads N  threads that index a plain FASTQ (default: the host's, at most 32)\n"
		"  --stream-input     bounded device memory: the input is read three times (k-mers, reference reads, coding) and only a window of\n"
		"                     four chunks is resident at a time instead of the whole input (same archive)\n"
		"  --domains K        K independent model domains (equal shares of the reads, each compressed on its own): `colord_hip decompress`\n"
		"                     decodes them side by side; costs archive size (own k-mer statistics and reference reads per domain)\n"
		"  --gpus N [--gpu-list a,b,..] [--transport rccl|host]   reads sharded over N GPUs, one host thread and one model domain per GPU;\n"
		"                     the k-mer set, reference reads and index are replicated through RCCL (or host staging: several ranks per GPU)\n");
}

// `colord_hip parse-check [--parse-threads N] [--part-symbols N] [--chunk-bases X] input`: the reader alone (no GPU): per chunk a
// digest of what the compressor would be handed (bases, qualities, offsets, packs, parts), then of the ids — the test that the
// indexed, multi-threaded reader of a plain FASTQ returns exactly what the sequential one returns
int run_parse_check(int argc, char** argv)
{
	uint64_t part_symbols = 2u << 21; int threads = 1, passes = 1; double chunk_bases = 1.0e9; std::string in;
	for (int i = 2; i < argc; ++i)
	{
		const std::string a = argv[i];
		if (a == "--parse-threads" && i + 1 < argc) threads = atoi(argv[++i]);
		else if (a == "--part-symbols" && i + 1 < argc) part_symbols = strtoull(argv[++i], nullptr, 10);
		else if (a == "--chunk-bases" && i + 1 < argc) chunk_bases = atof(argv[++i]);
		else if (a == "--passes" && i + 1 < argc) passes = atoi(argv[++i]);          // the input again after Reader::rewind (--stream-input): same lines, same totals
		else in = a;
	}
	if (in.empty()) die("parse-check: expected an input path");
	Reader R; R.part_symbols = part_symbols; R.threads = threads; R.open(in);
	const bool idx = R.map && R.index_mapped();
	auto fnv = [](uint64_t h, const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; } return h; };
	Chunk ch; ch.pinned = false;
	for (int pass = 0; pass < passes; ++pass)
	{
		if (pass) { R.rewind(); printf("pass %d\n", pass + 1); }
		uint32_t ci = 0;
		while (R.next_chunk(ch, (uint64_t)chunk_bases))
		{
			uint64_t h = 0xcbf29ce484222325ull;
			h = fnv(h, ch.bases, ch.n); if (R.fastq) h = fnv(h, ch.quals, ch.n);
			h = fnv(h, ch.off.data(), ch.off.size() * 8); h = fnv(h, ch.packs.data(), ch.packs.size() * 4); h = fnv(h, ch.parts.data(), ch.parts.size() * 4);
			printf("chunk %u: %zu reads %llu bases %zu packs %zu parts %016llx\n", ci++, ch.off.size() - 1, (unsigned long long)ch.n, ch.packs.size() - 1, ch.parts.size() - 1, (unsigned long long)h);
		}
	}
	uint64_t h = 0xcbf29ce484222325ull;
	h = fnv(h, R.ids.data(), R.ids.size()); h = fnv(h, R.id_off.data(), R.id_off.size() * 8); h = fnv(h, R.plus.data(), R.plus.size());
	printf("ids %016llx reads %llu bases %llu header symbols %llu\n", (unsigned long long)h, (unsigned long long)R.n_reads, (unsigned long long)R.n_bases, (unsigned long long)R.header_symbols);
	fprintf(stderr, "parse-check: %s reader, %d thread(s)\n", idx ? "indexed" : "sequential", threads);
	ch.release();
	return 0;
}

// ---- reads sharded over several GPUs: one host thread per GPU (the reference's orchestrator is one process of threads too,
// ---- compression.cpp:547-689).  The input is read ONCE by the process (gzip and FASTA included): rank r takes the r-th contiguous range
// ---- of the reads (equal shares of the bases), cuts its own reader packs and chunks, and drives its own cl_compressor; the two exchanges
// ---- of the *_finish steps run through the Transport (RCCL, or host staging) bound to cl_exchange; every rank is one model domain of the
// ---- coders.  Each rank writes ITS parts into the archive file at the offsets an all-gather of the byte counts gives it (pwrite; no
// ---- part travels to another rank); rank 0's thread adds `meta`, `header`, `hipdomains`, `info` and the footer.
namespace {
struct Source {                                    // the whole input as records: slices of the mapping (indexed reader) or of one host chunk
	const Reader* R = nullptr; const Chunk* whole = nullptr; uint64_t n = 0;
	uint32_t len(uint64_t i) const { return R->indexed ? R->recs[i].len : (uint32_t)(whole->off[i + 1] - whole->off[i]); }
	const uint8_t* seq(uint64_t i) const { return R->indexed ? R->recs[i].seq : whole->bases + whole->off[i]; }
	const uint8_t* qual(uint64_t i) const { return R->indexed ? R->recs[i].qual : whole->quals + whole->off[i]; }
};
struct RankOut {
	std::vector<uint8_t> dna, qual; std::vector<uint64_t> dsz, qsz; std::vector<uint32_t> counts;      // this rank's parts, in order
	uint64_t n_reads = 0, n_bases = 0, mean_read_len = 0; uint32_t sparse_range = 0, n_refs = 0; cl_kmer_stats ks{}; size_t n_chunks = 0;
	uint64_t dna_base = 0, qual_base = 0, qual_framed = 0;     // where its framed `dna` / `qual` parts start in the file; bytes of the latter
	uint64_t moved = 0;
};
uint32_t varint_len(uint64_t x) { uint32_t n = 1; for (; x; x >>= 8) ++n; return n; }
}

static int run_compress_multi(const Options& O, const Preset& P, const QDef& qd, int argc, char** argv)
{
	const auto t0 = std::chrono::steady_clock::now();
	auto lap = [&](const char* what) { if (O.verbose) fprintf(stderr, "[%7.2f s] %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), what); };
	const bool independent = O.domains > 1;             // --domains K: the shares are compressed one after the other on one GPU, nothing is exchanged
	const uint32_t world = independent ? (uint32_t)O.domains : (uint32_t)O.gpus;
	std::vector<int> devs = O.gpu_list;
	if (independent) devs.assign(world, O.gpu);
	if (devs.empty()) for (int i = 0; i < O.gpus; ++i) devs.push_back(i);
	if (devs.size() != world) die("--gpu-list must name --gpus devices");
	int n_dev = 0; hipck(hipGetDeviceCount(&n_dev), "hipGetDeviceCount");
	for (int d : devs) if (d < 0 || d >= n_dev) die("--gpus / --gpu-list: no such device");
	// reference-genome mode (compression.cpp:405-447) with sharded reads: every rank is handed the genome and the pseudo reads, the library
	// lets rank 0 count the genome's k-mers and contribute the pseudo reads (reference reads 0 .. n_pseudo - 1 of the replicated store)
	const bool with_genome = !O.genome.empty();
	if (with_genome && independent) die("-G,--reference-genome is not available with --domains");
	genome_io::Sequences G, PR; std::mutex pr_mu; bool pr_made = false;
	uint32_t genome_read_len = 0, n_pseudo = 0;
	if (with_genome)
	{
		try { G = genome_io::read_fasta(O.genome); } catch (const std::exception& e) { die(e.what()); }
		if (G.off.size() - 1 >= (1ull << 32)) die("reference genome: too many sequences");
		if (O.verbose) fprintf(stderr, "total sequences in reference genome file: %zu (%zu bases)\n", G.off.size() - 1, G.codes.size());
	}
	const bool use_rccl = !independent && O.transport == "rccl";
	if (use_rccl) { std::vector<int> u = devs; std::sort(u.begin(), u.end()); if (std::adjacent_find(u.begin(), u.end()) != u.end()) die("--transport rccl needs distinct devices (several ranks on one GPU: --transport host)"); }

	// the input, once
	Reader R; R.part_symbols = O.part_symbols; R.open(O.in);
	R.threads = O.parse_threads ? O.parse_threads : (int)std::min<unsigned>(32, std::max<unsigned>(1, std::thread::hardware_concurrency()));
	Chunk whole; whole.pinned = false;
	Source S; S.R = &R;
	if (R.map && R.index_mapped())
	{
		S.n = R.recs.size();
		for (const auto& r : R.recs) { R.ids.insert(R.ids.end(), r.id, r.id + r.id_len); R.id_off.push_back(R.ids.size()); R.plus.push_back(r.plus_eq); R.n_bases += r.len; }
		R.n_reads = S.n;
	}
	else
	{
		R.indexed = false;
		if (!R.next_chunk(whole, ~0ull >> 1)) die("no reads in " + O.in);
		S.whole = &whole; S.n = whole.off.size() - 1;
	}
	lap("input read");
	const uint64_t n = S.n, total = R.n_bases;
	if (!n) die("no reads in " + O.in);
	if (n >= (1ull << 32)) die("more than 2^32 reads");
	const bool with_qual = R.fastq;
	uint32_t k = O.k, a = O.a;
	if (!k)
	{	// adjustKmerAndAnchorLen (compression.cpp:42-95) on the estimate from the file size
		const double fac = R.gz ? (R.fastq ? 2.08 : 3.98) : (R.fastq ? 0.49 : 0.98);
		const uint64_t est = (uint64_t)(fac * (double)R.file_bytes);
		if (est < 1000000000ull) { k = 20; a = 16; } else if (est < 4000000000ull) { k = 21; a = 18; } else if (est < 16000000000ull) { k = 23; a = 21; }
		else if (est < 48000000000ull) { k = 24; a = 22; } else if (est < 128000000000ull) { k = 25; a = 22; } else { k = 26; a = 23; }
	}
	cl_compress_params cp{};
	cp.k = k; cp.f = P.f; cp.ci = P.ci; cp.cs = P.cs; cp.c = P.c; cp.anchor_len = a; cp.min_part_alt = P.min_part_alt; cp.max_rec = P.max_rec; cp.min_anchors = (uint32_t)O.min_anchors;
	cp.level = P.level; cp.source = O.source; cp.sparse = P.sparse; cp.sparse_g = P.g; cp.sparse_exponent = O.exponent;
	cp.cost_mult = O.cost_mult; cp.frac_always = O.frac_always; cp.frac_min = O.frac_min; cp.max_matches_mult = O.max_matches_mult;
	cl_qual_params qp{}; qp.mode = P.qual_mode; qp.source = O.source; qp.level = P.level;
	qp.n_fwd = (uint32_t)qd.fwd.size(); std::copy(qd.fwd.begin(), qd.fwd.end(), qp.fwd);
	qp.n_rev = (uint32_t)qd.rev.size(); std::copy(qd.rev.begin(), qd.rev.end(), qp.rev);

	// shares: rank r starts at the first read whose cumulative base count reaches total * r / world (as colord_amd/mgpu.py)
	std::vector<uint64_t> first(world + 1, n);
	{
		first[0] = 0; uint64_t acc = 0; uint32_t r = 1;
		for (uint64_t i = 0; i < n && r < world; ++i)
		{
			acc += S.len(i);
			while (r < world && (double)acc >= (double)total * r / world) first[r++] = i;
		}
	}
	// the header stream on a host thread of its own, next to everything else
	std::vector<std::vector<uint8_t>> hdr_parts; std::vector<uint32_t> hdr_counts; std::string hdr_err;
	std::thread hdr([&]() { code_headers(R, (uint32_t)n, O.header_mode, hdr_parts, hdr_counts, hdr_err); });

	// transports
	std::vector<std::unique_ptr<Transport>> tp(world);
	RcclGroup rccl; rccl.comms.assign(world, nullptr);
	std::unique_ptr<HostHub> hub;
	if (independent) {}
	else if (use_rccl)
	{
		const ncclResult_t e = ncclCommInitAll(rccl.comms.data(), (int)world, devs.data());
		if (e != ncclSuccess) die(std::string("ncclCommInitAll: ") + ncclGetErrorString(e));
		for (uint32_t r = 0; r < world; ++r) { auto t = std::make_unique<RcclTransport>(); if (t->init(rccl.comms[r], devs[r], r, world, &rccl) != CL_OK) die(t->err); tp[r] = std::move(t); }
	}
	else
	{
		hub = std::make_unique<HostHub>(world);
		for (uint32_t r = 0; r < world; ++r) { auto t = std::make_unique<HostTransport>(); t->init(hub.get(), devs[r], r); tp[r] = std::move(t); }
	}
	const int fd = ::open(O.out.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
	if (fd < 0) die("cannot open file: " + O.out);
	std::vector<RankOut> out(world);
	auto rank_main = [&](uint32_t rank) {
		Transport* const T = tp[rank].get(); RankOut& RO = out[rank];
		hipck(hipSetDevice(devs[rank]), "hipSetDevice");
		cl_ctx* ctx = nullptr; cl_ctx* qctx = nullptr;
		ck(nullptr, cl_ctx_create(devs[rank], &ctx), "cl_ctx_create"); ck(nullptr, cl_ctx_create(devs[rank], &qctx), "cl_ctx_create");
		const uint64_t r0 = first[rank], r1 = first[rank + 1];
		uint64_t my_bases = 0; for (uint64_t i = r0; i < r1; ++i) my_bases += S.len(i);
		cl_exchange X; if (T) X = T->exchange();
		cl_compressor* cmp = nullptr;
		ck(ctx, cl_compressor_create(ctx, qctx, &cp, with_qual ? &qp : nullptr, T ? &X : nullptr, my_bases, &cmp), "cl_compressor_create");
		auto upload = [&](const genome_io::Sequences& Q) -> cl_reads* {
			uint8_t* d_codes = nullptr; uint64_t* d_off = nullptr; cl_reads* r = nullptr;
			hipck(hipMalloc((void**)&d_codes, Q.codes.size() + 1), "hipMalloc"); hipck(hipMalloc((void**)&d_off, Q.off.size() * 8), "hipMalloc");
			hipck(hipMemcpy(d_codes, Q.codes.data(), Q.codes.size(), hipMemcpyHostToDevice), "hipMemcpy");
			hipck(hipMemcpy(d_off, Q.off.data(), Q.off.size() * 8, hipMemcpyHostToDevice), "hipMemcpy");
			ck(ctx, cl_reads_pack(ctx, d_codes, d_off, (uint32_t)(Q.off.size() - 1), 0, &r), "reference genome");
			hipck(hipFree(d_codes), "hipFree"); hipck(hipFree(d_off), "hipFree");
			return r;
		};
		if (with_genome) { cl_reads* gr = upload(G); ck(ctx, cl_compressor_genome_add(cmp, gr), "reference genome k-mers"); cl_reads_free(gr); }
		// chunks of whole reader packs (the packs are cut from this rank's first read on: in_reads.cpp:62-77).  The chunk size follows the
		// rank's share unless --chunk-bases says otherwise: at least 12 chunks a rank, so that the look-ahead pipeline of the compressor (encode
		// lanes, preparation threads: three to five chunks deep) fills — 8 ranks on 5 Gbases would otherwise get one chunk each.
		const uint64_t rank_chunk = O.chunk_bases_set ? (uint64_t)O.chunk_bases : std::min<uint64_t>((uint64_t)O.chunk_bases, std::max<uint64_t>(my_bases / 12, 32ull << 20));
		std::vector<DevChunk> chunks; std::vector<uint64_t> cut;                 // cut[ci] .. cut[ci + 1]: the reads of chunk ci
		Chunk host;
		Reader B; B.part_symbols = O.part_symbols;                            // (its pack / part bookkeeping only)
		// the reads [c0, c1) into the host buffer (offsets, bases, qualities)
		auto fill = [&](uint64_t c0, uint64_t c1) {
			if (host.off.size() != c1 - c0 + 1) { host.clear(); for (uint64_t x = c0; x < c1; ++x) { host.n += S.len(x); host.off.push_back(host.n); } }
			{ const uint64_t tot = host.n; host.n = 0; host.reserve(tot + 1, with_qual); host.n = tot; }
			for (uint64_t x = c0; x < c1; ++x) { memcpy(host.bases + host.off[x - c0], S.seq(x), S.len(x)); if (with_qual) memcpy(host.quals + host.off[x - c0], S.qual(x), S.len(x)); }
		};
		// The rank's input buffers on the device.  Under --stream-input a chunk is uploaded three times (counting, reference pass, coding
		// pass) and freed after each: its buffers come from and go back to a small per-rank cache — round 5 made a hipMalloc and a hipFree per
		// buffer and chunk, and hipFree waits for the WHOLE device: every announce stalled the lanes and preparation threads of all ranks on
		// the GPU.  (Resident input: every buffer is asked for once.)
		DevCache dcache;
		uint8_t* d_bases_stage = nullptr; uint64_t bases_stage_cap = 0;          // (the 1-byte-per-base form cl_reads_pack reads: needed only during the call)
		auto upload_chunk = [&](DevChunk& dc) {
			if (host.n + 1 > bases_stage_cap) { if (d_bases_stage) hipck(hipFree(d_bases_stage), "hipFree"); d_bases_stage = (uint8_t*)dcache.get(host.n + 1, bases_stage_cap); if (!d_bases_stage) die("hipMalloc"); }
			dc.d_off = (uint64_t*)dcache.get(host.off.size() * 8, dc.off_cap); if (!dc.d_off) die("hipMalloc");
			hipck(hipMemcpy(d_bases_stage, host.bases, host.n, hipMemcpyHostToDevice), "hipMemcpy");
			hipck(hipMemcpy(dc.d_off, host.off.data(), host.off.size() * 8, hipMemcpyHostToDevice), "hipMemcpy");
			if (with_qual)
			{
				dc.d_quals = (uint8_t*)dcache.get(host.n + 1, dc.quals_cap);
				if (!dc.d_quals) die("hipMalloc (the input does not fit this GPU's memory: --stream-input keeps only a window of it resident)");
				hipck(hipMemcpy(dc.d_quals, host.quals, host.n, hipMemcpyHostToDevice), "hipMemcpy");
			}
			ck(ctx, cl_reads_pack(ctx, d_bases_stage, dc.d_off, dc.n_reads, 1, &dc.reads), "input");
		};
		auto free_chunk = [&](DevChunk& dc) {
			if (dc.reads) cl_reads_free(dc.reads);
			dcache.put(dc.d_quals, dc.quals_cap); dcache.put(dc.d_off, dc.off_cap);
			dc.reads = nullptr; dc.d_quals = nullptr; dc.d_off = nullptr;
		};
		for (uint64_t i = r0; i < r1; )
		{
			host.clear();
			auto full = [&]() { return host.n >= rank_chunk && host.pack_acc == 0 && host.off.size() > 1; };
			const uint64_t c0 = i;
			while (i < r1 && !full()) { const uint32_t L = S.len(i); host.n += L; host.off.push_back(host.n); B.close_bounds(host, L); ++i; }
			B.finish_bounds(host);
			fill(c0, i);
			DevChunk dc; dc.n_reads = (uint32_t)(host.off.size() - 1); dc.n_bases = host.n; dc.packs = host.packs; dc.parts = host.parts;
			if (with_qual && !host.quals_in_range()) die("quality values outside '!'..'~'+1 (Phred+33, 0..95) are not supported");
			upload_chunk(dc);
			ck(ctx, cl_compressor_count_add(cmp, dc.reads), "pass 1");
			if (O.stream_input) free_chunk(dc);                                 // (--stream-input: a chunk leaves HBM after each pass, as in the single-GPU path)
			cut.push_back(c0);
			chunks.push_back(std::move(dc));
		}
		cut.push_back(r1);
		// a chunk of an earlier pass again (--stream-input): the same reads, from the source this process holds
		auto reload = [&](size_t ci) { host.clear(); fill(cut[ci], cut[ci + 1]); upload_chunk(chunks[ci]); };
		if (!O.stream_input) host.release();
		ck(ctx, cl_compressor_count_finish(cmp, &RO.ks), "k-mer counting (exchange 1)");
		if (with_genome)
		{	// pseudo reads of 20 x the mean read length (of ALL reads: the same on every rank), made once
			uint64_t mrl = 0;
			ck(ctx, cl_compressor_info(cmp, nullptr, nullptr, nullptr, &mrl, nullptr, nullptr), "cl_compressor_info");
			{
				std::lock_guard<std::mutex> l(pr_mu);
				if (!pr_made)
				{
					if (20 * mrl >= (1ull << 32)) die("reference genome: pseudo reads too long");
					genome_read_len = (uint32_t)(20 * mrl);
					try { PR = genome_io::pseudo_reads(G, genome_read_len, (k - 1) * 10); } catch (const std::exception& e) { die(e.what()); }
					n_pseudo = (uint32_t)(PR.off.size() - 1);
					pr_made = true;
				}
				else if (genome_read_len != (uint32_t)(20 * mrl)) die("internal: the ranks disagree about the mean read length");
			}
			cl_reads* pr = upload(PR);
			ck(ctx, cl_compressor_pseudo_reads(cmp, pr), "reference genome pseudo reads");
			cl_reads_free(pr);
		}
		for (size_t ci = 0; ci < chunks.size(); ++ci)
		{
			if (O.stream_input) reload(ci);
			ck(ctx, cl_compressor_refs_add(cmp, chunks[ci].reads), "reference reads");
			if (O.stream_input) free_chunk(chunks[ci]);
		}
		ck(ctx, cl_compressor_refs_finish(cmp), "reference index (exchange 2)");
		ck(ctx, cl_compressor_info(cmp, nullptr, nullptr, nullptr, &RO.mean_read_len, &RO.sparse_range, &RO.n_refs), "cl_compressor_info");
		uint64_t max_bases = 0, max_parts = 0; for (auto& dc : chunks) { max_bases = std::max(max_bases, dc.n_bases); max_parts = std::max<uint64_t>(max_parts, dc.parts.size()); }
		const uint64_t dna_cap = max_bases + 64 * max_parts + 4096, qual_cap = (uint64_t)(max_bases * 1.35) + 64 * max_parts + 4096;
		uint8_t* d_dna = nullptr; uint8_t* d_qual = nullptr;
		hipck(hipMalloc((void**)&d_dna, dna_cap), "hipMalloc"); if (with_qual) hipck(hipMalloc((void**)&d_qual, qual_cap), "hipMalloc");
		// the chunks are announced a window ahead of the one being coded (look-ahead of the compressor: encode lanes, preparation threads);
		// --stream-input: they are uploaded again as they are announced and freed as they are coded, so at most window + 1 are resident
		size_t ann_window = 4;
		if (const char* e = getenv("COLORD_HIP_ANNOUNCE_WINDOW")) ann_window = (size_t)std::max(0, atoi(e));
		if (O.stream_input && !ann_window) ann_window = 4;
		size_t announced = 0;
		for (size_t ci = 0; ci < chunks.size(); ++ci)
		{
			DevChunk& dc = chunks[ci];
			const size_t have = ann_window ? std::min(chunks.size(), ci + 1 + ann_window) : chunks.size();
			for (; announced < have; ++announced)
			{
				DevChunk& x = chunks[announced];
				if (O.stream_input) reload(announced);
				ck(ctx, cl_compressor_prepare_parts(cmp, x.reads, x.packs.data(), (uint32_t)x.packs.size() - 1, x.parts.data(), (uint32_t)x.parts.size() - 1, x.d_quals, x.d_off), "look-ahead");
			}
			const uint32_t np = (uint32_t)dc.parts.size() - 1;
			std::vector<uint64_t> dsz(np), qsz(np); cl_compress_info info{};
			ck(ctx, cl_compressor_encode(cmp, dc.reads, dc.d_quals, dc.d_off, dc.parts.data(), np, dc.packs.data(), (uint32_t)dc.packs.size() - 1, d_dna, dna_cap, dsz.data(), d_qual, qual_cap, qsz.data(), &info), "pass 2");
			const size_t od = RO.dna.size(), oq = RO.qual.size();
			RO.dna.resize(od + info.dna_bytes); RO.qual.resize(oq + info.qual_bytes);
			if (info.dna_bytes) hipck(hipMemcpy(RO.dna.data() + od, d_dna, info.dna_bytes, hipMemcpyDeviceToHost), "hipMemcpy");
			if (info.qual_bytes) hipck(hipMemcpy(RO.qual.data() + oq, d_qual, info.qual_bytes, hipMemcpyDeviceToHost), "hipMemcpy");
			RO.dsz.insert(RO.dsz.end(), dsz.begin(), dsz.end()); if (with_qual) RO.qsz.insert(RO.qsz.end(), qsz.begin(), qsz.end());
			for (uint32_t p = 0; p < np; ++p) RO.counts.push_back(dc.parts[p + 1] - dc.parts[p]);
			RO.n_reads += dc.n_reads; RO.n_bases += dc.n_bases;
			free_chunk(dc);
		}
		host.release();
		RO.n_chunks = chunks.size();
		(void)hipFree(d_dna); if (d_qual) (void)hipFree(d_qual);
		if (d_bases_stage) (void)hipFree(d_bases_stage);
		dcache.clear();
		// where this rank's parts go: an all-gather of the framed byte counts, an exclusive sum, pwrite — `dna` of all ranks first, then `qual`
		uint64_t mine[2] = { 0, 0 };
		for (size_t p = 0; p < RO.dsz.size(); ++p) mine[0] += varint_len(RO.counts[p]) + RO.dsz[p];
		for (size_t p = 0; p < RO.qsz.size(); ++p) mine[1] += varint_len(0) + RO.qsz[p];
		if (T)
		{
			std::vector<uint64_t> all(2 * (size_t)world);
			ck(ctx, T->all_gather_host(mine, 2, all.data()), "all-gather of the stream sizes");
			uint64_t dna_all = 0; for (uint32_t r = 0; r < world; ++r) { if (r == rank) RO.dna_base = dna_all; dna_all += all[2 * r]; }
			uint64_t q = dna_all; for (uint32_t r = 0; r < world; ++r) { if (r == rank) RO.qual_base = q; q += all[2 * r + 1]; }
		}
		else
		{	// independent domains run one after the other: a domain's parts follow those of the domains before it
			uint64_t at = 0; for (uint32_t r = 0; r < rank; ++r) at = out[r].qual_base + out[r].qual_framed;
			RO.dna_base = at; RO.qual_base = at + mine[0];
		}
		RO.qual_framed = mine[1];
		auto write_parts = [&](uint64_t at, const std::vector<uint8_t>& data, const std::vector<uint64_t>& sz, bool counted) {
			std::vector<uint8_t> buf; uint64_t o = 0;
			for (size_t p = 0; p < sz.size(); ++p)
			{	// (parts are framed in memory in runs of ~64 MB, one pwrite per run)
				ArchiveWriter::varint(buf, counted ? RO.counts[p] : 0);
				buf.insert(buf.end(), data.begin() + o, data.begin() + o + sz[p]); o += sz[p];
				if (buf.size() >= (64u << 20) || p + 1 == sz.size())
				{
					size_t done = 0;
					while (done < buf.size()) { const ssize_t w = pwrite(fd, buf.data() + done, buf.size() - done, (off_t)(at + done)); if (w <= 0) die("cannot write the archive (disk full?)"); done += (size_t)w; }
					at += buf.size(); buf.clear();
				}
			}
		};
		write_parts(RO.dna_base, RO.dna, RO.dsz, true);
		if (with_qual) write_parts(RO.qual_base, RO.qual, RO.qsz, false);
		RO.moved = T ? T->bytes_moved : 0;
		cl_compressor_free(cmp);
		cl_ctx_destroy(qctx); cl_ctx_destroy(ctx);
	};
	if (independent) for (uint32_t r = 0; r < world; ++r) rank_main(r);
	else
	{
		std::vector<std::thread> th;
		for (uint32_t r = 0; r < world; ++r) th.emplace_back(rank_main, r);
		for (auto& t : th) t.join();
	}
	lap("all ranks through (parts written)");
	hdr.join();
	if (!hdr_err.empty()) die("header stream: " + hdr_err);
	// the rest of the archive behind the parts: meta, header, hipdomains, info, footer — by this thread
	uint64_t end = 0;
	for (auto& RO : out) { for (size_t p = 0; p < RO.dsz.size(); ++p) end += varint_len(RO.counts[p]) + RO.dsz[p]; for (size_t p = 0; p < RO.qsz.size(); ++p) end += 1 + RO.qsz[p]; }
	ArchiveWriter ar;
	ar.f = fdopen(fd, "r+b"); if (!ar.f) die("cannot open file: " + O.out);
	if (fseeko(ar.f, (off_t)end, SEEK_SET) != 0) die("cannot seek in the archive");
	ar.off = end;
	const int s_meta = ar.reg("meta"), s_genome = (with_genome && O.store_genome) ? ar.reg("ref-genome") : -1, s_header = ar.reg("header"), s_dna = ar.reg("dna"), s_qual = with_qual ? ar.reg("qual") : -1, s_dom = ar.reg("hipdomains");
	uint32_t tot_ref = (uint32_t)n + n_pseudo;
	const RankOut& R0 = out[0];
	if (P.sparse) { std::vector<uint8_t> acc((size_t)n + n_pseudo); ck(nullptr, cl_ref_accept((uint32_t)n, n_pseudo, R0.sparse_range, O.exponent, acc.data()), "cl_ref_accept"); tot_ref = 0; for (uint8_t x : acc) tot_ref += x; }
	uint8_t md[16] = { 0 };
	if (with_genome && !O.store_genome && cl_genome_md5(G.codes.data(), G.off.data(), (uint32_t)(G.off.size() - 1), md) != CL_OK) die("cannot checksum the reference genome");
	const std::vector<uint8_t> meta = pack_meta(MetaIn{ (uint32_t)n, n_pseudo, tot_ref, P.c, P.level, O.source, R0.mean_read_len, with_qual, P.qual_mode, qd.rev, O.header_mode, P.sparse != 0, R0.sparse_range, O.exponent,
	                                                    with_genome, O.store_genome, genome_read_len, (k - 1) * 10, md });
	ar.add(s_meta, meta.data(), meta.size(), 0);
	if (s_genome >= 0)
	{	// CReferenceGenome::Store(archive) (reference_genome.cpp:325-370): one part, metadata = number of sequences
		std::vector<uint8_t> gs(G.codes.size() / 3 + 4096); uint64_t got = 0;
		cl_status st = cl_genome_encode(G.codes.data(), G.off.data(), (uint32_t)(G.off.size() - 1), gs.data(), gs.size(), &got);
		if (st == CL_E_CAPACITY) { gs.resize(got); st = cl_genome_encode(G.codes.data(), G.off.data(), (uint32_t)(G.off.size() - 1), gs.data(), gs.size(), &got); }
		if (st != CL_OK) die("cannot code the reference genome");
		ar.add(s_genome, gs.data(), got, G.off.size() - 1);
	}
	for (size_t p = 0; p < hdr_parts.size(); ++p) ar.add(s_header, hdr_parts[p].data(), hdr_parts[p].size(), hdr_counts[p]);
	// part tables of the streams the ranks wrote, and the model domains (first read, first `dna` part of every rank)
	// (bit 31 of the count: INDEPENDENT domains — each has its own reference reads, so each decodes with a decoder of its own and its
	// own sparse range, appended below; cli/reader.hpp)
	std::vector<uint8_t> dom; le<uint32_t>(dom, world | (independent ? 0x80000000u : 0u));
	uint64_t first_read = 0, dna_total = 0, qual_total = 0;
	for (uint32_t r = 0; r < world; ++r)
	{
		const RankOut& RO = out[r];
		le<uint64_t>(dom, first_read); le<uint64_t>(dom, (uint64_t)ar.streams[s_dna].parts.size());
		uint64_t at = RO.dna_base;
		for (size_t p = 0; p < RO.dsz.size(); ++p) { ar.streams[s_dna].parts.push_back(ArchiveWriter::Part{ at, RO.dsz[p] }); at += varint_len(RO.counts[p]) + RO.dsz[p]; dna_total += RO.dsz[p]; }
		at = RO.qual_base;
		if (with_qual) for (size_t p = 0; p < RO.qsz.size(); ++p) { ar.streams[s_qual].parts.push_back(ArchiveWriter::Part{ at, RO.qsz[p] }); at += 1 + RO.qsz[p]; qual_total += RO.qsz[p]; }
		first_read += RO.n_reads;
	}
	if (first_read != n) die("internal: the ranks' reads do not add up");
	if (independent) for (uint32_t r = 0; r < world; ++r) le<uint32_t>(dom, out[r].sparse_range);
	ar.add(s_dom, dom.data(), dom.size(), 0);
	const int s_info = ar.reg("info");
	const std::vector<uint8_t> inf = pack_info(R.total_bytes, total, (uint32_t)n, argc, argv);
	ar.add(s_info, inf.data(), inf.size(), 0);
	ar.close();
	if (use_rccl) rccl.destroy_all();
	tp.clear();
	whole.release();
	if (R.g) gzclose(R.g);
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	fprintf(stderr, "colord_hip: %llu reads, %llu bases on %u GPU(s) [%s], k=%u a=%u; dna %llu B, qual %llu B, header %zu parts; %u reference reads; %llu B exchanged by rank 0; %.2f s\n", (unsigned long long)n, (unsigned long long)total,
		world, use_rccl ? "RCCL" : "host-staged", k, a, (unsigned long long)dna_total, (unsigned long long)qual_total, hdr_parts.size(), R0.n_refs, (unsigned long long)R0.moved, sec);
	return 0;
}

// `colord_hip rccl-selftest [--gpus N]`: the three collectives of RcclTransport on N devices (default 1: a communicator of one rank still runs
// the RCCL code path) with uneven and empty shares, checked against what they must deliver
int run_rccl_selftest(int argc, char** argv)
{
	int world = 1; for (int i = 2; i + 1 < argc; ++i) if (std::string(argv[i]) == "--gpus") world = atoi(argv[i + 1]);
	std::vector<int> devs; for (int i = 0; i < world; ++i) devs.push_back(i);
	std::vector<ncclComm_t> comms((size_t)world, nullptr);
	const ncclResult_t e = ncclCommInitAll(comms.data(), world, devs.data());
	if (e != ncclSuccess) die(std::string("ncclCommInitAll: ") + ncclGetErrorString(e));
	std::vector<std::string> errs((size_t)world);
	auto run = [&](int r) {
		RcclTransport T; if (T.init(comms[r], devs[r], (uint32_t)r, (uint32_t)world) != CL_OK) { errs[r] = T.err; return; }
		auto fillv = [&](uint32_t from, uint32_t to, uint64_t i) { return (uint8_t)(from * 31 + to * 7 + i * 13 + 5); };
		// all_gather_host
		uint64_t v[3] = { (uint64_t)r * 10 + 1, (uint64_t)r * 10 + 2, ~0ull - (uint64_t)r }; std::vector<uint64_t> o(3 * (size_t)world);
		if (T.all_gather_host(v, 3, o.data()) != CL_OK) { errs[r] = T.err; return; }
		for (int p = 0; p < world; ++p) if (o[3 * p] != (uint64_t)p * 10 + 1 || o[3 * p + 2] != ~0ull - (uint64_t)p) { errs[r] = "all_gather_host: wrong values"; return; }
		// all_to_all_v: rank a sends ((a + b) % 3 == 0 ? 0 : 1000 + 17 a + 5 b) bytes to rank b
		auto cnt = [&](int a_, int b_) -> uint64_t { return (a_ + b_) % 3 == 0 && a_ != b_ ? 0ull : 1000ull + 17 * a_ + 5 * b_; };
		std::vector<uint64_t> sb((size_t)world), rb((size_t)world); uint64_t st = 0, rt = 0;
		for (int p = 0; p < world; ++p) { sb[p] = cnt(r, p); rb[p] = cnt(p, r); st += sb[p]; rt += rb[p]; }
		std::vector<uint8_t> hs(st + 1), hr(rt + 1);
		{ uint64_t o2 = 0; for (int p = 0; p < world; ++p) for (uint64_t i = 0; i < sb[p]; ++i) hs[o2++] = fillv((uint32_t)r, (uint32_t)p, i); }
		uint8_t* ds = nullptr; uint8_t* dr = nullptr;
		if (hipMalloc((void**)&ds, st + 1) != hipSuccess || hipMalloc((void**)&dr, rt + 1) != hipSuccess) { errs[r] = "hipMalloc"; return; }
		(void)hipMemcpy(ds, hs.data(), st, hipMemcpyHostToDevice);
		if (T.all_to_all_v(ds, sb.data(), dr, rb.data()) != CL_OK) { errs[r] = T.err; return; }
		(void)hipMemcpy(hr.data(), dr, rt, hipMemcpyDeviceToHost);
		{ uint64_t o2 = 0; for (int p = 0; p < world; ++p) for (uint64_t i = 0; i < rb[p]; ++i) if (hr[o2++] != fillv((uint32_t)p, (uint32_t)r, i)) { errs[r] = "all_to_all_v: wrong bytes"; return; } }
		// all_gather_v: rank a contributes (a % 2 ? 0 : 777 + 3 a) bytes
		auto gc = [&](int a_) -> uint64_t { return a_ % 2 ? 0ull : 777ull + 3 * a_; };
		std::vector<uint64_t> gb((size_t)world); uint64_t gt = 0; for (int p = 0; p < world; ++p) { gb[p] = gc(p); gt += gb[p]; }
		std::vector<uint8_t> gs(gc(r) + 1), gr(gt + 1); for (uint64_t i = 0; i < gc(r); ++i) gs[i] = fillv((uint32_t)r, 99, i);
		uint8_t* dgs = nullptr; uint8_t* dgr = nullptr;
		if (hipMalloc((void**)&dgs, gc(r) + 1) != hipSuccess || hipMalloc((void**)&dgr, gt + 1) != hipSuccess) { errs[r] = "hipMalloc"; return; }
		(void)hipMemcpy(dgs, gs.data(), gc(r), hipMemcpyHostToDevice);
		if (T.all_gather_v(dgs, gc(r), dgr, gb.data()) != CL_OK) { errs[r] = T.err; return; }
		(void)hipMemcpy(gr.data(), dgr, gt, hipMemcpyDeviceToHost);
		{ uint64_t o2 = 0; for (int p = 0; p < world; ++p) for (uint64_t i = 0; i < gb[p]; ++i) if (gr[o2++] != fillv((uint32_t)p, 99, i)) { errs[r] = "all_gather_v: wrong bytes"; return; } }
		(void)hipFree(ds); (void)hipFree(dr); (void)hipFree(dgs); (void)hipFree(dgr);
	};
	std::vector<std::thread> th; for (int r = 0; r < world; ++r) th.emplace_back(run, r);
	for (auto& t : th) t.join();
	for (ncclComm_t c : comms) if (c) (void)ncclCommDestroy(c);
	for (int r = 0; r < world; ++r) if (!errs[r].empty()) die("rccl-selftest, rank " + std::to_string(r) + ": " + errs[r]);
	printf("rccl-selftest: all_gather_host, all_to_all_v, all_gather_v ok on %d rank(s)\n", world);
	return 0;
}

int run_compress(int argc, char** argv)
{
	Options O;
	const std::string mode = argv[1];
	O.source = mode == "compress-ont" ? 0 : mode == "compress-pbraw" ? 1 : mode == "compress-pbhifi" ? 2 : -1;
	if (O.source < 0) { usage(); die("unknown mode " + mode); }
	std::vector<std::string> pos;
	auto need = [&](int& i) -> std::string { if (i + 1 >= argc) die(std::string("option ") + argv[i] + " needs a value"); return argv[++i]; };
	for (int i = 2; i < argc; ++i)
	{
		const std::string a = argv[i];
		if (a == "-p" || a == "--priority") { const std::string v = need(i); O.prio = v == "ratio" ? 0 : v == "balanced" ? 1 : v == "memory" ? 2 : -1; if (O.prio < 0) die("unknown priority " + v); }
		else if (a == "-k" || a == "--kmer-len") { O.k = (uint32_t)atoi(need(i).c_str()); if (O.k < 15 || O.k > 28) die("-k,--kmer-len must be in [15, 28]"); }
		else if (a == "-a" || a == "--anchor-len") O.a = (uint32_t)atoi(need(i).c_str());
		else if (a == "-q" || a == "--qual") { const std::string v = need(i); O.qual_mode = qual_mode_of(v); if (O.qual_mode < 0) die("unknown quality mode " + v); }
		else if (a == "-T" || a == "--qual-thresholds") { O.T = list_u32(need(i)); O.has_T = true; while (i + 1 < argc && isdigit((unsigned char)argv[i + 1][0]) && pos.size() + (size_t)(argc - i - 1) > 2) O.T.push_back((uint32_t)atoi(argv[++i])); }
		else if (a == "-D" || a == "--qual-values") { O.D = list_u32(need(i)); O.has_D = true; while (i + 1 < argc && isdigit((unsigned char)argv[i + 1][0]) && pos.size() + (size_t)(argc - i - 1) > 2) O.D.push_back((uint32_t)atoi(argv[++i])); }
		else if (a == "-i" || a == "--identifier") { const std::string v = need(i); O.header_mode = v == "org" ? 0 : v == "main" ? 1 : v == "none" ? 2 : -1; if (O.header_mode < 0) die("unknown header mode " + v); }
		else if (a == "-c" || a == "--max-candidates") { O.c = atol(need(i).c_str()); if (O.c < 1) die("-c must be positive"); }
		else if (a == "-L" || a == "--Lowest-count") O.ci = atol(need(i).c_str());
		else if (a == "-H" || a == "--Highest-count") O.cs = atol(need(i).c_str());
		else if (a == "-f" || a == "--filter-modulo") { O.f = atol(need(i).c_str()); if (O.f < 1) die("-f must be positive"); }
		else if (a == "-e" || a == "--edit-script-mult") O.cost_mult = atof(need(i).c_str());
		else if (a == "-r" || a == "--max-recurence-level") O.max_rec = atol(need(i).c_str());
		else if (a == "--min-to-alt") O.min_to_alt = atol(need(i).c_str());
		else if (a == "--min-mmer-frac") O.frac_min = atof(need(i).c_str());
		else if (a == "--min-mmer-force-enc") O.frac_always = atof(need(i).c_str());
		else if (a == "--max-matches-mult") O.max_matches_mult = atof(need(i).c_str());
		else if (a == "--min-anchors") O.min_anchors = atol(need(i).c_str());
		else if (a == "-R" || a == "--Ref-reads-mode") { const std::string v = need(i); O.ref_mode = v == "all" ? 0 : v == "sparse" ? 1 : -1; if (O.ref_mode < 0) die("unknown reference reads mode " + v); }
		else if (a == "-g" || a == "--sparse-range") O.g = atof(need(i).c_str());
		else if (a == "-x" || a == "--sparse-exponent") O.exponent = atof(need(i).c_str());
		else if (a == "-t" || a == "--threads") (void)need(i);
		else if (a == "--fill-factor-filtered-kmers" || a == "--fill-factor-kmers-to-reads") (void)need(i);     // host hash-table tuning of the reference: no counterpart here
		else if (a == "-v" || a == "--verbose") O.verbose = true;
		else if (a == "-G" || a == "--reference-genome") O.genome = need(i);
		else if (a == "-s" || a == "--store-reference") O.store_genome = true;
		else if (a == "--gpu") O.gpu = atoi(need(i).c_str());
		else if (a == "--gpus") { O.gpus = atoi(need(i).c_str()); if (O.gpus < 1 || O.gpus > 64) die("--gpus must be in [1, 64]"); }
		else if (a == "--domains") { O.domains = atoi(need(i).c_str()); if (O.domains < 1 || O.domains > 1024) die("--domains must be in [1, 1024]"); }
		else if (a == "--gpu-list") { for (uint32_t v : list_u32(need(i))) O.gpu_list.push_back((int)v); }
		else if (a == "--transport") { O.transport = need(i); if (O.transport != "rccl" && O.transport != "host") die("--transport must be rccl or host"); }
		else if (a == "--chunk-bases") { O.chunk_bases = atof(need(i).c_str()); O.chunk_bases_set = true; }
		else if (a == "--part-symbols") { O.part_symbols = strtoull(need(i).c_str(), nullptr, 10); if (O.part_symbols < 1024 || O.part_symbols > (2u << 21)) die("--part-symbols must be in [1024, 4194304]"); }
		else if (a == "--stream-input") O.stream_input = true;
		else if (a == "--parse-threads") { O.parse_threads = atoi(need(i).c_str()); if (O.parse_threads < 1 || O.parse_threads > 256) die("--parse-threads must be in [1, 256]"); }
		else if (a == "-h" || a == "--help") { usage(); return 0; }
		else if (!a.empty() && a[0] == '-' && a.size() > 1) die("unknown option " + a);
		else pos.push_back(a);
	}
	if (pos.size() != 2) { usage(); die("expected input and output paths"); }
	O.in = pos[0]; O.out = pos[1];
	// the checks of arg_parse.cpp:604-625
	if (O.k && !O.a) die("if -k,--kmer-len is set -a,--anchor-len also must be set");
	if (!O.k && O.a) die("if -a,--anchor-len is set -k,--kmer-len also must be set");
	if (O.k && O.a > O.k) die("-a,--anchor-len must be less than or equal to -k,--kmer-len");
	const Preset P0 = PRESETS[O.source][O.prio];
	Preset P = P0;
	if (O.ci >= 0) P.ci = (uint32_t)O.ci;
	if (O.cs >= 0) P.cs = (uint32_t)O.cs;
	if (O.f >= 0) P.f = (uint32_t)O.f;
	if (O.c >= 0) P.c = (uint32_t)O.c;
	if (O.max_rec >= 0) P.max_rec = (uint32_t)O.max_rec;
	if (O.min_to_alt >= 0) P.min_part_alt = (uint32_t)O.min_to_alt;
	if (O.qual_mode >= 0) P.qual_mode = O.qual_mode;
	if (O.ref_mode >= 0) P.sparse = O.ref_mode;
	if (O.g >= 0) P.g = O.g;
	if (P.c > 64) die("-c,--max-candidates above 64 is not supported by the DNA coder of this build");
	// quality thresholds / representatives (adjust_quality_mode_and_thresholds, arg_parse.cpp:410-450)
	QDef qd = qual_defaults(P.qual_mode);
	static const char* qnames[] = { "org", "5-avg", "4-avg", "2-avg", "5-fix", "4-fix", "2-fix", "avg", "none" };
	if (O.has_T) { if (qd.fwd.empty()) die(std::string("-T,--qual-thresholds is not allowed for '") + qnames[P.qual_mode] + "' quality mode"); if (O.T.size() != qd.fwd.size()) die(std::string("for '") + qnames[P.qual_mode] + "' quality compression mode expected number of quality thresholds is " + std::to_string(qd.fwd.size()) + ", but " + std::to_string(O.T.size()) + " given."); qd.fwd = O.T; }
	if (O.has_D) { if (qd.rev.empty()) die(std::string("-D,--qual-values is not allowed for '") + qnames[P.qual_mode] + "' quality mode"); if (O.D.size() != qd.rev.size()) die(std::string("for '") + qnames[P.qual_mode] + "' quality compression mode expected number of quality values is " + std::to_string(qd.rev.size()) + ", but " + std::to_string(O.D.size()) + " given."); qd.rev = O.D; }
	for (size_t i = 0; i < qd.fwd.size(); ++i) if (qd.fwd[i] > 95 || (i && qd.fwd[i] < qd.fwd[i - 1])) die("quality thresholds must be ascending values in [0, 95]");
	if (!O.gpu_list.empty() && O.gpus == 1) O.gpus = (int)O.gpu_list.size();
	if (O.domains > 1 && O.gpus > 1) die("--domains and --gpus exclude each other (every GPU is a model domain already)");
	if (O.stream_input && O.domains > 1) die("--stream-input is not available with --domains");
	if (O.gpus > 1 || O.domains > 1) return run_compress_multi(O, P, qd, argc, argv);

	const auto t0 = std::chrono::steady_clock::now();
	auto lap = [&](const char* what) { if (O.verbose) fprintf(stderr, "[%7.2f s] %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), what); };
	hipck(hipSetDevice(O.gpu), "hipSetDevice");
	Reader R; R.part_symbols = O.part_symbols; R.open(O.in);
	R.threads = O.parse_threads ? O.parse_threads : (int)std::min<unsigned>(32, std::max<unsigned>(1, std::thread::hardware_concurrency()));
	if (const char* e = getenv("COLORD_HIP_PARSE_THREADS")) R.threads = std::max(1, atoi(e));
	// the two pinned staging buffers of the reader are made (and the HIP runtime started) beside the indexing of the input
	Chunk hostbuf[2]; std::thread prealloc[2];
	if (R.map && !getenv("COLORD_HIP_NO_PREALLOC"))
	{
		const uint64_t est = (uint64_t)(0.49 * (double)R.file_bytes), want = std::min<uint64_t>((uint64_t)O.chunk_bases, est);
		for (int i = 0; i < (est > want + want / 2 ? 2 : 1); ++i)
			prealloc[i] = std::thread([&, i, want]() { if (hipSetDevice(O.gpu) == hipSuccess) hostbuf[i].reserve(want + (8ull << 20), true); });
	}
	struct JoinPrealloc { std::thread* t; ~JoinPrealloc() { for (int i = 0; i < 2; ++i) if (t[i].joinable()) t[i].join(); } } join_prealloc{ prealloc };
	if (R.map && R.index_mapped()) lap("input indexed");
	// k-mer / anchor length from the estimated number of bases (adjustKmerAndAnchorLen, compression.cpp:42-95)
	uint32_t k = O.k, a = O.a;
	if (!k)
	{
		const double fac = R.gz ? (R.fastq ? 2.08 : 3.98) : (R.fastq ? 0.49 : 0.98);
		const uint64_t est = (uint64_t)(fac * (double)R.file_bytes);
		if (est < 1000000000ull) { k = 20; a = 16; } else if (est < 4000000000ull) { k = 21; a = 18; } else if (est < 16000000000ull) { k = 23; a = 21; }
		else if (est < 48000000000ull) { k = 24; a = 22; } else if (est < 128000000000ull) { k = 25; a = 22; } else { k = 26; a = 23; }
	}
	cl_ctx* ctx = nullptr; cl_ctx* qctx = nullptr;
	ck(nullptr, cl_ctx_create(O.gpu, &ctx), "cl_ctx_create");
	ck(nullptr, cl_ctx_create(O.gpu, &qctx), "cl_ctx_create");
	cl_compress_params cp{};
	cp.k = k; cp.f = P.f; cp.ci = P.ci; cp.cs = P.cs; cp.c = P.c; cp.anchor_len = a; cp.min_part_alt = P.min_part_alt; cp.max_rec = P.max_rec; cp.min_anchors = (uint32_t)O.min_anchors;
	cp.level = P.level; cp.source = O.source; cp.sparse = P.sparse; cp.sparse_g = P.g; cp.sparse_exponent = O.exponent;
	cp.cost_mult = O.cost_mult; cp.frac_always = O.frac_always; cp.frac_min = O.frac_min; cp.max_matches_mult = O.max_matches_mult;
	cl_qual_params qp{}; qp.mode = P.qual_mode; qp.source = O.source; qp.level = P.level;
	qp.n_fwd = (uint32_t)qd.fwd.size(); std::copy(qd.fwd.begin(), qd.fwd.end(), qp.fwd);
	qp.n_rev = (uint32_t)qd.rev.size(); std::copy(qd.rev.begin(), qd.rev.end(), qp.rev);
	const bool with_qual = R.fastq;
	const uint64_t est_bases = (uint64_t)((R.gz ? (R.fastq ? 2.08 : 3.98) : (R.fastq ? 0.49 : 0.98)) * (double)R.file_bytes);
	cl_compressor* cmp = nullptr;
	ck(ctx, cl_compressor_create(ctx, qctx, &cp, with_qual ? &qp : nullptr, nullptr, est_bases, &cmp), "cl_compressor_create");
	// reference-genome mode (compression.cpp:405-429): the genome's sequences are a second input of the k-mer counter
	genome_io::Sequences G; const bool with_genome = !O.genome.empty();
	auto upload = [&](const genome_io::Sequences& S) -> cl_reads* {
		uint8_t* d_codes = nullptr; uint64_t* d_off = nullptr; cl_reads* r = nullptr;
		hipck(hipMalloc((void**)&d_codes, S.codes.size() + 1), "hipMalloc"); hipck(hipMalloc((void**)&d_off, S.off.size() * 8), "hipMalloc");
		hipck(hipMemcpy(d_codes, S.codes.data(), S.codes.size(), hipMemcpyHostToDevice), "hipMemcpy");
		hipck(hipMemcpy(d_off, S.off.data(), S.off.size() * 8, hipMemcpyHostToDevice), "hipMemcpy");
		ck(ctx, cl_reads_pack(ctx, d_codes, d_off, (uint32_t)(S.off.size() - 1), 0, &r), "reference genome");
		hipck(hipFree(d_codes), "hipFree"); hipck(hipFree(d_off), "hipFree");
		return r;
	};
	if (with_genome)
	{
		try { G = genome_io::read_fasta(O.genome); } catch (const std::exception& e) { die(e.what()); }
		if (G.off.size() - 1 >= (1ull << 32)) die("reference genome: too many sequences");
		cl_reads* gr = upload(G);
		ck(ctx, cl_compressor_genome_add(cmp, gr), "reference genome k-mers");
		cl_reads_free(gr);
		if (O.verbose) fprintf(stderr, "total sequences in reference genome file: %zu (%zu bases)\n", G.off.size() - 1, G.codes.size());
	}

	// pass 1 while parsing: every chunk goes to HBM (2-bit arena + quality bytes) and stays there for the three passes
	// (the parser fills one pinned buffer on a thread of its own while this thread uploads and scans the other).
	// --stream-input: a chunk leaves HBM again after each pass and the input is read three times (k-mers; reference reads; coding,
	// where a loader thread keeps a window of chunks resident ahead of the coders) — the reference reads its file twice for the same
	// reason (compression.cpp:432,547-561).
	std::vector<DevChunk> chunks;
	for (int i = 0; i < 2; ++i) if (prealloc[i].joinable()) prealloc[i].join();
	double t_wait_parser = 0, t_check = 0, t_upload = 0, t_scan = 0;               // (-v: where this thread's time of a pass over the input went)
	auto for_each_chunk = [&](const std::function<void(Chunk&)>& fn) {
		std::mutex pmu; std::condition_variable pcv; int filled[2] = { 0, 0 };      // 0 free, 1 full, 2 end of input
		std::thread parser([&]() {
			for (int i = 0;; i ^= 1)
			{
				{ std::unique_lock<std::mutex> l(pmu); pcv.wait(l, [&]() { return filled[i] == 0; }); }
				const bool ok = R.next_chunk(hostbuf[i], (uint64_t)O.chunk_bases);
				{ std::lock_guard<std::mutex> l(pmu); filled[i] = ok ? 1 : 2; }
				pcv.notify_all();
				if (!ok) break;
			}
		});
		for (int hi = 0;; hi ^= 1)
		{
			const auto tw = std::chrono::steady_clock::now();
			{ std::unique_lock<std::mutex> l(pmu); pcv.wait(l, [&]() { return filled[hi] != 0; }); }
			t_wait_parser += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
			if (filled[hi] == 2) break;
			fn(hostbuf[hi]);
			{ std::lock_guard<std::mutex> l(pmu); filled[hi] = 0; }
			pcv.notify_all();
		}
		parser.join();
	};
	// (the 1-byte-per-base form cl_reads_pack reads is needed only during the call: ONE staging buffer per calling thread, kept — a
	// hipMalloc + hipFree per chunk were two device-wide synchronisations in front of every chunk's k-mer scan)
	struct BaseStage { uint8_t* p = nullptr; uint64_t cap = 0; hipStream_t s[2] = { nullptr, nullptr }; ~BaseStage() { if (p) (void)hipFree(p); for (hipStream_t x : s) if (x) (void)hipStreamDestroy(x); } };
	BaseStage stage_main, stage_loader; DevCache dcache;
	auto upload_chunk = [&](cl_ctx* uc, const Chunk& host, DevChunk& dc) {
		BaseStage& bs = uc == ctx ? stage_main : stage_loader;
		if (host.n + 1 > bs.cap) { if (bs.p) hipck(hipFree(bs.p), "hipFree"); bs.cap = host.n + host.n / 8 + 4096; hipck(hipMalloc((void**)&bs.p, bs.cap), "hipMalloc"); }
		if (O.stream_input)
		{	// (the window's buffers go round: DevCache)
			dc.d_off = (uint64_t*)dcache.get(host.off.size() * 8, dc.off_cap); if (with_qual) dc.d_quals = (uint8_t*)dcache.get(host.n + 1, dc.quals_cap);
			if (!dc.d_off || (with_qual && !dc.d_quals)) die("out of device memory for a chunk of the input");
		}
		else
		{
			hipck(hipMalloc((void**)&dc.d_off, host.off.size() * 8), "hipMalloc");
			if (with_qual) hipck(hipMalloc((void**)&dc.d_quals, host.n + 1), "hipMalloc (the input does not fit this GPU's memory: --stream-input keeps only a window of it resident)");
		}
		// bases and qualities on a stream each (two copy engines side by side; one after the other they took 91 ms per 2 GB)
		static const bool two_engines = !getenv("COLORD_HIP_UPLOAD_ONE_ENGINE");
		if (two_engines)
		{
			if (!bs.s[0]) for (int i = 0; i < 2; ++i) hipck(hipStreamCreateWithFlags(&bs.s[i], hipStreamNonBlocking), "hipStreamCreate");
			hipck(hipMemcpyAsync(bs.p, host.bases, host.n, hipMemcpyHostToDevice, bs.s[0]), "hipMemcpyAsync");
			if (with_qual) hipck(hipMemcpyAsync(dc.d_quals, host.quals, host.n, hipMemcpyHostToDevice, bs.s[1]), "hipMemcpyAsync");
			hipck(hipMemcpy(dc.d_off, host.off.data(), host.off.size() * 8, hipMemcpyHostToDevice), "hipMemcpy");
			hipck(hipStreamSynchronize(bs.s[0]), "hipMemcpyAsync"); hipck(hipStreamSynchronize(bs.s[1]), "hipMemcpyAsync");
		}
		else
		{
			hipck(hipMemcpy(bs.p, host.bases, host.n, hipMemcpyHostToDevice), "hipMemcpy");
			if (with_qual) hipck(hipMemcpy(dc.d_quals, host.quals, host.n, hipMemcpyHostToDevice), "hipMemcpy");
			hipck(hipMemcpy(dc.d_off, host.off.data(), host.off.size() * 8, hipMemcpyHostToDevice), "hipMemcpy");
		}
		ck(uc, cl_reads_pack(uc, bs.p, dc.d_off, dc.n_reads, 1, &dc.reads), "input");        // "Only ACGTN symbols supported inside a read"
	};
	auto free_chunk = [&](DevChunk& dc) {
		if (dc.reads) cl_reads_free(dc.reads);
		if (O.stream_input) { dcache.put(dc.d_quals, dc.quals_cap); dcache.put(dc.d_off, dc.off_cap); }
		else { if (dc.d_quals) (void)hipFree(dc.d_quals); if (dc.d_off) (void)hipFree(dc.d_off); }
		dc.reads = nullptr; dc.d_quals = nullptr; dc.d_off = nullptr;
	};
	// a later pass must see the chunks of the first
	auto same_chunk = [&](const Chunk& host, size_t ci) {
		if (ci >= chunks.size() || chunks[ci].n_reads != host.off.size() - 1 || chunks[ci].n_bases != host.n || chunks[ci].packs != host.packs || chunks[ci].parts != host.parts)
			die("the input changed between two passes over it (--stream-input)");
	};
	for_each_chunk([&](Chunk& host) {
		DevChunk dc; dc.n_reads = (uint32_t)(host.off.size() - 1); dc.n_bases = host.n; dc.packs = host.packs; dc.parts = host.parts;
		// quality bytes outside 33..128 would index past the coder's tables: the input is rejected, not coded (qualities are Phred+33)
		auto t0 = std::chrono::steady_clock::now();
		auto lapse = [&](double& acc) { const auto t = std::chrono::steady_clock::now(); acc += std::chrono::duration<double>(t - t0).count(); t0 = t; };
		if (with_qual && !host.quals_in_range()) die("quality values outside '!'..'~'+1 (Phred+33, 0..95) are not supported");
		lapse(t_check);
		upload_chunk(ctx, host, dc);
		lapse(t_upload);
		ck(ctx, cl_compressor_count_add(cmp, dc.reads), "pass 1");
		lapse(t_scan);
		if (O.stream_input) free_chunk(dc);
		chunks.push_back(std::move(dc));
	});
	lap("input parsed, uploaded and scanned (pass 1)");        // (the pinned staging of a resident input is used once more: pass 2 receives its parts in it)
	if (O.verbose) fprintf(stderr, "# pass 1, this thread: %.2f s waiting for the parser, %.2f s quality range, %.2f s upload + packing, %.2f s k-mer scan; the parser: %.2f s bookkeeping, %.2f s copies (%d threads)\n",
		t_wait_parser, t_check, t_upload, t_scan, R.t_book, R.t_copy, R.threads);
	const uint32_t n = (uint32_t)R.n_reads; const uint64_t total = R.n_bases;
	if (!n) die("no reads in " + O.in);
	// the header stream on a host thread, next to the GPU path (CEntrComprHeaders, entr_header.cpp:23-45)
	std::vector<std::vector<uint8_t>> hdr_parts; std::vector<uint32_t> hdr_counts; std::string hdr_err;
	std::thread hdr([&]() { code_headers(R, n, O.header_mode, hdr_parts, hdr_counts, hdr_err); });
	cl_kmer_stats ks{};
	ck(ctx, cl_compressor_count_finish(cmp, &ks), "k-mer counting");
	lap("k-mers counted");
	uint32_t genome_read_len = 0, n_pseudo = 0; const uint32_t genome_overlap = (k - 1) * 10;     // compression.cpp:407,447
	if (with_genome)
	{
		uint64_t mrl = 0;
		ck(ctx, cl_compressor_info(cmp, nullptr, nullptr, nullptr, &mrl, nullptr, nullptr), "cl_compressor_info");
		if (20 * mrl >= (1ull << 32)) die("reference genome: pseudo reads too long");
		genome_read_len = (uint32_t)(20 * mrl);
		genome_io::Sequences PR;
		try { PR = genome_io::pseudo_reads(G, genome_read_len, genome_overlap); } catch (const std::exception& e) { die(e.what()); }
		n_pseudo = (uint32_t)(PR.off.size() - 1);
		cl_reads* pr = upload(PR);
		ck(ctx, cl_compressor_pseudo_reads(cmp, pr), "reference genome pseudo reads");
		cl_reads_free(pr);
		if (O.verbose) fprintf(stderr, "# ref genome pseudo reads: %u (length %u, overlap %u)\n", n_pseudo, genome_read_len, genome_overlap);
	}
	if (!O.stream_input) for (auto& dc : chunks) ck(ctx, cl_compressor_refs_add(cmp, dc.reads), "reference reads");
	else
	{
		R.rewind();
		size_t ci = 0;
		for_each_chunk([&](Chunk& host) {
			same_chunk(host, ci);
			DevChunk dc; dc.n_reads = chunks[ci].n_reads;
			upload_chunk(ctx, host, dc);
			ck(ctx, cl_compressor_refs_add(cmp, dc.reads), "reference reads");
			free_chunk(dc);
			++ci;
		});
		if (ci != chunks.size()) die("the input changed between two passes over it (--stream-input)");
	}
	ck(ctx, cl_compressor_refs_finish(cmp), "reference index");
	lap("reference reads and index");
	uint64_t mean_read_len = 0; uint32_t sparse_range = 0, n_refs = 0;
	ck(ctx, cl_compressor_info(cmp, nullptr, nullptr, nullptr, &mean_read_len, &sparse_range, &n_refs), "cl_compressor_info");
	if (O.verbose) fprintf(stderr, "k=%u a=%u; %llu k-mers, %llu kept; %u reference reads; sparse range %u\n", k, a, (unsigned long long)ks.tot_kmers, (unsigned long long)ks.n_unique_counted, n_refs, sparse_range);

	ArchiveWriter ar; ar.open(O.out);
	const int s_meta = ar.reg("meta"), s_genome = (with_genome && O.store_genome) ? ar.reg("ref-genome") : -1, s_header = ar.reg("header"), s_dna = ar.reg("dna"), s_qual = with_qual ? ar.reg("qual") : -1;
	if (s_genome >= 0)
	{	// CReferenceGenome::Store(archive) (reference_genome.cpp:325-370): one part, metadata = number of sequences
		std::vector<uint8_t> gs(G.codes.size() / 3 + 4096); uint64_t got = 0;
		cl_status st = cl_genome_encode(G.codes.data(), G.off.data(), (uint32_t)(G.off.size() - 1), gs.data(), gs.size(), &got);
		if (st == CL_E_CAPACITY) { gs.resize(got); st = cl_genome_encode(G.codes.data(), G.off.data(), (uint32_t)(G.off.size() - 1), gs.data(), gs.size(), &got); }
		if (st != CL_OK) die("cannot code the reference genome");
		ar.add(s_genome, gs.data(), got, G.off.size() - 1);
	}
	uint64_t dna_total = 0, qual_total = 0; uint32_t n_parts_total = 0;
	{	// pass 2: chunk by chunk; the parts of a chunk go to the archive while the next chunk is coded
		uint64_t max_bases = 0, max_parts = 0; for (auto& dc : chunks) { max_bases = std::max(max_bases, dc.n_bases); max_parts = std::max<uint64_t>(max_parts, dc.parts.size()); }
		const uint64_t dna_cap = max_bases + 64 * max_parts + 4096, qual_cap = (uint64_t)(max_bases * 1.35) + 64 * max_parts + 4096;
		// The parts of a chunk leave through TWO sets of buffers (device and pinned host) and a writer thread: while chunk i + 1 is coded, chunk
		// i's parts are copied out on a stream of their own and added to the archive.  (Round 5: copied into pageable memory and written by the
		// coding thread itself they cost 0.2 s of the 0.55 s a chunk took at 20 Gbases.)  The pinned side is pass 1's staging where the input is
		// resident (it is free by now) and grows on demand otherwise: pinning two sets of the device side's worst-case sizes (4.7 GB at 1-Gbase
		// chunks, for the 0.8 GB the parts of two chunks take) cost about a second a run.
		uint8_t* d_dna2[2] = { nullptr, nullptr }; uint8_t* d_qual2[2] = { nullptr, nullptr }; uint8_t* h_dna2[2] = { nullptr, nullptr }; uint8_t* h_qual2[2] = { nullptr, nullptr };
		uint64_t h_dna_cap[2] = { 0, 0 }, h_qual_cap[2] = { 0, 0 };
		hipStream_t out_stream = nullptr; hipEvent_t out_ev[2] = { nullptr, nullptr };
		hipck(hipStreamCreateWithFlags(&out_stream, hipStreamNonBlocking), "hipStreamCreate");
		for (int b = 0; b < 2; ++b)
		{
			hipck(hipMalloc((void**)&d_dna2[b], dna_cap), "hipMalloc");
			if (with_qual) hipck(hipMalloc((void**)&d_qual2[b], qual_cap), "hipMalloc");
			hipck(hipEventCreateWithFlags(&out_ev[b], hipEventDisableTiming), "hipEventCreate");
			if (!O.stream_input)
			{
				h_dna2[b] = hostbuf[b].bases; h_dna_cap[b] = hostbuf[b].bases ? hostbuf[b].cap : 0;
				h_qual2[b] = hostbuf[b].quals; h_qual_cap[b] = hostbuf[b].quals ? hostbuf[b].cap : 0;
				hostbuf[b].bases = hostbuf[b].quals = nullptr; hostbuf[b].cap = 0;
			}
		}
		auto host_room = [&](uint8_t*& p, uint64_t& cap, uint64_t need) {
			if (need <= cap) return;
			if (p) (void)hipHostFree(p);
			cap = need + need / 4 + (1ull << 20); p = nullptr;
			hipck(hipHostMalloc((void**)&p, cap, hipHostMallocDefault), "hipHostMalloc");
		};
		struct OutJob { size_t ci; int b; std::vector<uint64_t> dsz, qsz; uint64_t dna_bytes, qual_bytes; };
		std::mutex omu; std::condition_variable ocv; std::deque<OutJob> ojobs; bool odone = false; bool obusy[2] = { false, false }; std::string oerr;
		double t_wait_writer = 0, t_encode = 0, t_writer = 0;                      // (-v: the coding thread waiting for the writer / inside the encode calls; the writer adding parts)
		std::thread writer([&]() {
			for (;;)
			{
				OutJob j;
				{ std::unique_lock<std::mutex> l(omu); ocv.wait(l, [&]() { return odone || !ojobs.empty(); }); if (ojobs.empty()) return; j = std::move(ojobs.front()); ojobs.pop_front(); }
				if (oerr.empty() && hipEventSynchronize(out_ev[j.b]) != hipSuccess) { (void)hipGetLastError(); oerr = "copy of the parts to the host failed"; }
				const auto tw0 = std::chrono::steady_clock::now();
				struct Busy { double& acc; std::chrono::steady_clock::time_point t; ~Busy() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); } } busy{ t_writer, tw0 };
				if (oerr.empty())
				{	// (after a failure nothing more goes into the archive: the jobs are only taken off the queue so that the coding thread is not left waiting)
					const DevChunk& dc = chunks[j.ci]; const uint32_t np = (uint32_t)j.dsz.size();
					uint64_t o = 0; for (uint32_t p = 0; p < np; ++p) { ar.add(s_dna, h_dna2[j.b] + o, j.dsz[p], dc.parts[p + 1] - dc.parts[p]); o += j.dsz[p]; }
					o = 0; if (with_qual) for (uint32_t p = 0; p < np; ++p) { ar.add(s_qual, h_qual2[j.b] + o, j.qsz[p], 0); o += j.qsz[p]; }
				}
				{ std::lock_guard<std::mutex> l(omu); obusy[j.b] = false; }
				ocv.notify_all();
			}
		});
		// every chunk is resident: announce them, so that candidates / anchors / edit scripts of the next chunks are computed on the
		// compressor's encode lanes while this thread codes and writes the parts of the chunks before them
		// (the coder parts are the reader packs: with them the `dna` coder's walks and sort of the next chunk are made ahead too)
		// --stream-input: a loader thread (a context of its own) parses and uploads the chunks again, at most WINDOW + 1 resident: the one
		// being coded and WINDOW announced ahead of it for the encode lanes and the preparation threads; it also frees what has been coded
		constexpr size_t WINDOW = 3;
		std::mutex lmu; std::condition_variable lcv; size_t n_loaded = 0, done_upto = 0; std::thread loader; cl_ctx* lctx = nullptr;
		if (O.stream_input)
		{
			ck(nullptr, cl_ctx_create(O.gpu, &lctx), "cl_ctx_create");
			R.rewind();
			loader = std::thread([&]() {
				hipck(hipSetDevice(O.gpu), "hipSetDevice");
				size_t freed = 0;
				auto free_done = [&](size_t upto) { for (; freed < upto; ++freed) free_chunk(chunks[freed]); };
				size_t ci = 0;
				for_each_chunk([&](Chunk& host) {
					same_chunk(host, ci);
					size_t upto;
					{ std::unique_lock<std::mutex> l(lmu); lcv.wait(l, [&]() { return ci - done_upto < WINDOW + 1; }); upto = done_upto; }
					free_done(upto);
					upload_chunk(lctx, host, chunks[ci]);
					++ci;
					{ std::lock_guard<std::mutex> l(lmu); n_loaded = ci; }
					lcv.notify_all();
				});
				if (ci != chunks.size()) die("the input changed between two passes over it (--stream-input)");
				{ std::unique_lock<std::mutex> l(lmu); lcv.wait(l, [&]() { return done_upto == chunks.size(); }); }
				free_done(chunks.size());
			});
		}
		// resident input: the chunks are announced a window ahead as well (COLORD_HIP_ANNOUNCE_WINDOW, 0 = all at once): lanes that run far
		// ahead of the coders only pile up edit scripts the pool has to grow for
		size_t ann_window = 4;
		if (const char* e = getenv("COLORD_HIP_ANNOUNCE_WINDOW")) ann_window = (size_t)std::max(0, atoi(e));
		size_t announced = 0;
		for (size_t ci = 0; ci < chunks.size(); ++ci)
		{
			DevChunk& dc = chunks[ci];
			{
				size_t have = ann_window ? std::min(chunks.size(), ci + 1 + ann_window) : chunks.size();
				if (O.stream_input) { std::unique_lock<std::mutex> l(lmu); lcv.wait(l, [&]() { return n_loaded > ci; }); have = n_loaded; }
				for (; announced < have; ++announced)
				{
					DevChunk& x = chunks[announced];
					ck(ctx, cl_compressor_prepare_parts(cmp, x.reads, x.packs.data(), (uint32_t)x.packs.size() - 1, x.parts.data(), (uint32_t)x.parts.size() - 1, x.d_quals, x.d_off), "look-ahead");
				}
			}
			if (ci == 0)
			{
				lap("pass 2 set up, chunks announced");
				size_t fr = 0, tot = 0;
				if (O.verbose && hipMemGetInfo(&fr, &tot) == hipSuccess) fprintf(stderr, "# device memory before the first chunk of pass 2: %.1f of %.1f GB free\n", fr / 1e9, tot / 1e9);
			}
			const uint32_t np = (uint32_t)dc.parts.size() - 1;
			std::vector<uint64_t> dsz(np), qsz(np); cl_compress_info info{};
			const int b = (int)(ci & 1);
			{
				const auto tw = std::chrono::steady_clock::now();
				std::unique_lock<std::mutex> l(omu); ocv.wait(l, [&]() { return !obusy[b]; }); obusy[b] = true;      // (the writer is through with this set: chunk ci - 2)
				t_wait_writer += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
			}
			const auto te = std::chrono::steady_clock::now();
			ck(ctx, cl_compressor_encode(cmp, dc.reads, dc.d_quals, dc.d_off, dc.parts.data(), np, dc.packs.data(), (uint32_t)dc.packs.size() - 1, d_dna2[b], dna_cap, dsz.data(), d_qual2[b], qual_cap, qsz.data(), &info), "pass 2");
			t_encode += std::chrono::duration<double>(std::chrono::steady_clock::now() - te).count();
			host_room(h_dna2[b], h_dna_cap[b], info.dna_bytes); host_room(h_qual2[b], h_qual_cap[b], info.qual_bytes);       // (set b is the writer's no more: awaited above)
			if (info.dna_bytes) hipck(hipMemcpyAsync(h_dna2[b], d_dna2[b], info.dna_bytes, hipMemcpyDeviceToHost, out_stream), "hipMemcpyAsync");
			if (info.qual_bytes) hipck(hipMemcpyAsync(h_qual2[b], d_qual2[b], info.qual_bytes, hipMemcpyDeviceToHost, out_stream), "hipMemcpyAsync");
			hipck(hipEventRecord(out_ev[b], out_stream), "hipEventRecord");
			{ std::lock_guard<std::mutex> l(omu); ojobs.push_back(OutJob{ ci, b, dsz, qsz, info.dna_bytes, info.qual_bytes }); }
			ocv.notify_all();
			dna_total += info.dna_bytes; qual_total += info.qual_bytes; n_parts_total += np;
			// (a resident chunk stays where it is until the pass is over: hipFree waits for the whole device — the lanes and the preparation
			// working ahead on the next chunks — and nobody needs the room)
			if (O.stream_input) { { std::lock_guard<std::mutex> l(lmu); done_upto = ci + 1; } lcv.notify_all(); }     // (the loader frees it)
			else { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr < (48ull << 30)) free_chunk(dc); }   // (... unless the device is nearly full: the pools of the chunks to come take what this one held)
		}
		{ std::lock_guard<std::mutex> l(omu); odone = true; }
		ocv.notify_all();
		const auto tj = std::chrono::steady_clock::now();
		writer.join();
		{ size_t fr = 0, tot = 0; if (O.verbose && hipMemGetInfo(&fr, &tot) == hipSuccess) fprintf(stderr, "# device memory after pass 2: %.1f of %.1f GB free\n", fr / 1e9, tot / 1e9); }
		if (O.verbose) fprintf(stderr, "# pass 2, this thread: %.2f s in the encode calls, %.2f s waiting for the writer to hand a buffer set back, %.2f s for its last parts; the writer: %.2f s adding parts to the archive\n",
			t_encode, t_wait_writer, std::chrono::duration<double>(std::chrono::steady_clock::now() - tj).count(), t_writer);
		if (!oerr.empty()) { (void)remove(O.out.c_str()); die(oerr + " (no archive was written)"); }     // (what is on disk is half a file: it goes with the error)
		if (O.stream_input) { loader.join(); cl_ctx_destroy(lctx); hostbuf[0].release(); hostbuf[1].release(); }
		else if (getenv("COLORD_HIP_FULL_TEARDOWN")) for (DevChunk& dc : chunks) free_chunk(dc);
		for (int b = 0; b < 2; ++b) { (void)hipFree(d_dna2[b]); if (h_dna2[b]) (void)hipHostFree(h_dna2[b]); if (d_qual2[b]) (void)hipFree(d_qual2[b]); if (h_qual2[b]) (void)hipHostFree(h_qual2[b]); (void)hipEventDestroy(out_ev[b]); }
		(void)hipStreamDestroy(out_stream);
	}
	lap("pass 2 (dna + qual parts written)");
	hdr.join();
	lap("header stream");
	if (!hdr_err.empty()) die("header stream: " + hdr_err);
	for (size_t p = 0; p < hdr_parts.size(); ++p) ar.add(s_header, hdr_parts[p].data(), hdr_parts[p].size(), hdr_counts[p]);

	// meta (compression.cpp:704-779), info (utils.cpp:326-342)
	uint32_t tot_ref = n + n_pseudo;
	if (P.sparse) { std::vector<uint8_t> acc((size_t)n + n_pseudo); ck(ctx, cl_ref_accept(n, n_pseudo, sparse_range, O.exponent, acc.data()), "cl_ref_accept"); tot_ref = 0; for (uint8_t x : acc) tot_ref += x; }
	uint8_t md[16] = { 0 };
	if (with_genome && !O.store_genome && cl_genome_md5(G.codes.data(), G.off.data(), (uint32_t)(G.off.size() - 1), md) != CL_OK) die("cannot checksum the reference genome");
	const std::vector<uint8_t> meta = pack_meta(MetaIn{ n, n_pseudo, tot_ref, P.c, P.level, O.source, mean_read_len, with_qual, P.qual_mode, qd.rev, O.header_mode, P.sparse != 0, sparse_range, O.exponent,
	                                                    with_genome, O.store_genome, genome_read_len, genome_overlap, md });
	ar.add(s_meta, meta.data(), meta.size(), 0);
	const int s_info = ar.reg("info");
	const std::vector<uint8_t> inf = pack_info(R.total_bytes, total, n, argc, argv);
	ar.add(s_info, inf.data(), inf.size(), 0);
	ar.close();
	gzclose(R.g);
	lap("archive closed");
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	fprintf(stderr, "colord_hip: %u reads, %llu bases, k=%u a=%u, %zu chunk(s); dna %llu B (%u parts), qual %llu B, header %zu parts; %u reference reads; %.2f s\n", n, (unsigned long long)total, k, a,
		chunks.size(), (unsigned long long)dna_total, n_parts_total, (unsigned long long)qual_total, hdr_parts.size(), n_refs, sec);
	// The archive is complete and closed.  What is left is handing back tens of GB of device memory, the pinned staging and the mapping of the
	// input allocation by allocation — 0.4-1.2 s at 20 Gbases for what the end of the process does at once.  COLORD_HIP_FULL_TEARDOWN=1 walks
	// through it (leak checks).
	if (!getenv("COLORD_HIP_FULL_TEARDOWN")) { fflush(nullptr); _exit(0); }
	cl_compressor_free(cmp);
	cl_ctx_destroy(qctx); cl_ctx_destroy(ctx);
	return 0;
}
