// colord_hip — command-line compressor on top of libcolord_hip.so: FASTQ in, CoLoRd archive out, decodable by the
// reference's `colord decompress`.  Mirrors the compress side of the reference CLI (src/colord/main.cpp, arg_parse.cpp:
// `compress-ont | compress-pbhifi | compress-pbraw [-p ratio|balanced|memory] in out`) and the host side of runCompression
// (compression.cpp:344-785): input parsing, k / anchor length from the file size, reader packs, the `header`, `meta` and
// `info` streams and the archive container are host code here; everything between read bases / qualities and the `dna` /
// `qual` stream parts is one call into the GPU library (cl_compress_shard).
//
// Limits of this first version: FASTQ with 4-line records (plain or gzip), one GPU, the whole file resident (288 GB of HBM
// hold ~100 Gbases), no reference genome (-G), quality / header modes = the preset's.
#include "colord_hip.h"
#include "archive.hpp"
#include <hip/hip_runtime_api.h>
#include <zlib.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <thread>
#include <vector>

namespace {
struct Preset { int level; uint32_t ci, cs, f, c, max_rec, min_part_alt; int qual_mode; int sparse; double g; };
// arg_parse.cpp:89-408 — [source][priority]: ratio, balanced, memory (memory is the default priority)
const Preset PRESETS[3][3] = {
	{ { 3, 2, 120, 8, 10, 6, 48, 2, 0, 1 }, { 2, 3, 100, 9, 8, 5, 48, 2, 1, 2 }, { 1, 4, 80, 12, 5, 3, 64, 2, 1, 1 } },          // ONT, 4-avg qualities
	{ { 3, 2, 120, 8, 10, 6, 48, 8, 0, 1 }, { 2, 3, 100, 9, 8, 5, 48, 8, 1, 2 }, { 1, 4, 80, 12, 5, 3, 64, 8, 1, 1 } },          // PBRaw, qualities dropped
	{ { 3, 2, 150, 20, 12, 6, 48, 1, 0, 1 }, { 2, 3, 120, 30, 10, 5, 48, 1, 1, 6 }, { 2, 3, 100, 40, 8, 5, 48, 1, 1, 3 } },       // PBHiFi, 5-avg qualities
};
// default -T / -D values of the quality modes (arg_parse.cpp:410-450): mode -> forward thresholds, decoder representatives
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
void hipck(hipError_t e, const char* what) { if (e != hipSuccess) die(std::string(what) + ": " + hipGetErrorString(e)); }
void ck(cl_ctx* ctx, cl_status s, const char* what) { if (s != CL_OK) die(std::string(what) + ": " + (ctx ? cl_last_error(ctx) : "error")); }

template<class T> void le(std::vector<uint8_t>& v, T x) { for (size_t i = 0; i < sizeof(T); ++i) v.push_back((uint8_t)((uint64_t)x >> (8 * i))); }
void le_double(std::vector<uint8_t>& v, double d) { uint64_t u; memcpy(&u, &d, 8); le(v, u); }

struct Input {
	std::vector<uint8_t> bases, quals, ids, plus; std::vector<uint64_t> off, id_off;      // off: per-read base offsets
	uint64_t file_bytes = 0, header_symbols = 0; bool gz = false;
};
void read_fastq(const std::string& path, Input& in)
{
	FILE* probe = fopen(path.c_str(), "rb");
	if (!probe) die("cannot open file: " + path);
	unsigned char mg[2] = { 0, 0 }; size_t got = fread(mg, 1, 2, probe);
	fseek(probe, 0, SEEK_END); in.file_bytes = (uint64_t)ftell(probe); fclose(probe);
	in.gz = got == 2 && mg[0] == 0x1f && mg[1] == 0x8b;
	gzFile g = gzopen(path.c_str(), "rb");
	if (!g) die("cannot open file: " + path);
	gzbuffer(g, 1 << 22);
	std::vector<char> buf(1 << 24);
	std::string line[4]; int which = 0; std::string cur;
	in.off.push_back(0); in.id_off.push_back(0);
	auto flush_record = [&]() {
		if (line[0].empty() || line[0][0] != '@') die("FASTQ record does not start with '@'");
		if (line[2].empty() || line[2][0] != '+') die("FASTQ record without '+' line");
		if (line[1].size() != line[3].size()) die("sequence and quality lengths differ");
		in.header_symbols += line[0].size() + line[2].size();
		in.ids.insert(in.ids.end(), line[0].begin() + 1, line[0].end()); in.id_off.push_back(in.ids.size());
		bool eq = line[2].size() > 1;
		if (eq && line[2].compare(1, std::string::npos, line[0], 1, std::string::npos) != 0) die("quality header not empty but different than read header");
		in.plus.push_back(eq ? 1 : 0);
		in.bases.insert(in.bases.end(), line[1].begin(), line[1].end());
		in.quals.insert(in.quals.end(), line[3].begin(), line[3].end());
		in.off.push_back(in.bases.size());
	};
	for (;;)
	{
		const int n = gzread(g, buf.data(), (unsigned)buf.size());
		if (n < 0) die("read error: " + path);
		if (n == 0) break;
		for (int i = 0; i < n; ++i)
		{
			const char ch = buf[i];
			if (ch == '\n') { if (!cur.empty() && cur.back() == '\r') cur.pop_back(); line[which] = cur; cur.clear(); if (++which == 4) { flush_record(); which = 0; } }
			else cur.push_back(ch);
		}
	}
	if (!cur.empty()) { line[which] = cur; if (++which == 4) { flush_record(); which = 0; } }
	if (which != 0) die("truncated FASTQ record at the end of " + path);
	gzclose(g);
}
template<class T> T* to_device(const std::vector<T>& v, size_t extra = 0)
{
	T* d = nullptr; hipck(hipMalloc((void**)&d, (v.size() + extra + 1) * sizeof(T)), "hipMalloc");
	if (!v.empty()) hipck(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy");
	return d;
}
} // namespace

int run_decompress(int argc, char** argv);      // decompress.cpp
int run_info(int argc, char** argv);

int main(int argc, char** argv)
{
	if (argc >= 2 && std::string(argv[1]) == "decompress") return run_decompress(argc, argv);
	if (argc >= 2 && std::string(argv[1]) == "info") return run_info(argc, argv);
	if (argc < 4)
	{
		fprintf(stderr, "usage: colord_hip compress-ont|compress-pbhifi|compress-pbraw [-p ratio|balanced|memory] [--gpu N] input.fastq[.gz] output.colord\n"
		                "       colord_hip decompress archive.colord output.fastq\n       colord_hip info archive.colord\n");
		return 1;
	}
	const std::string mode = argv[1];
	const int source = mode == "compress-ont" ? 0 : mode == "compress-pbraw" ? 1 : mode == "compress-pbhifi" ? 2 : -1;
	if (source < 0) die("unknown mode " + mode);
	int prio = 2, gpu = 0; std::vector<std::string> pos;
	for (int i = 2; i < argc; ++i)
	{
		const std::string a = argv[i];
		if ((a == "-p" || a == "--priority") && i + 1 < argc) { const std::string v = argv[++i]; prio = v == "ratio" ? 0 : v == "balanced" ? 1 : v == "memory" ? 2 : -1; if (prio < 0) die("unknown priority " + v); }
		else if (a == "--gpu" && i + 1 < argc) gpu = atoi(argv[++i]);
		else pos.push_back(a);
	}
	if (pos.size() != 2) die("expected input and output paths");
	const Preset P = PRESETS[source][prio];
	const auto t0 = std::chrono::steady_clock::now();
	Input in; read_fastq(pos[0], in);
	const uint32_t n = (uint32_t)in.plus.size();
	if (!n) die("no reads in " + pos[0]);
	// k-mer / anchor length from the estimated number of bases (adjustKmerAndAnchorLen, compression.cpp:42-95)
	uint64_t est = (uint64_t)((in.gz ? 2.08 : 0.49) * (double)in.file_bytes);
	uint32_t k, a;
	if (est < 1000000000ull) { k = 20; a = 16; } else if (est < 4000000000ull) { k = 21; a = 18; } else if (est < 16000000000ull) { k = 23; a = 21; }
	else if (est < 48000000000ull) { k = 24; a = 22; } else if (est < 128000000000ull) { k = 25; a = 22; } else { k = 26; a = 23; }

	hipck(hipSetDevice(gpu), "hipSetDevice");
	cl_ctx* ctx = nullptr; cl_ctx* qctx = nullptr;
	ck(nullptr, cl_ctx_create(gpu, &ctx), "cl_ctx_create");
	ck(nullptr, cl_ctx_create(gpu, &qctx), "cl_ctx_create");
	uint8_t* d_bases = to_device(in.bases); uint64_t* d_off = to_device(in.off); uint8_t* d_quals = to_device(in.quals);
	cl_reads* reads = nullptr;
	ck(ctx, cl_reads_pack(ctx, d_bases, d_off, n, 1, &reads), "cl_reads_pack");
	hipck(hipFree(d_bases), "hipFree");
	// reader packs: a pack closes once its reads (with one guard byte each) reach 4 Mi symbols (in_reads.cpp:62-77)
	std::vector<uint32_t> packs{ 0 };
	{
		uint64_t acc = 0;
		for (uint32_t i = 0; i < n; ++i) { acc += in.off[i + 1] - in.off[i] + 1; if (acc >= (2u << 21)) { packs.push_back(i + 1); acc = 0; } }
		if (packs.back() != n) packs.push_back(n);
	}
	const uint32_t n_parts = (uint32_t)packs.size() - 1;
	const bool with_qual = true;
	const QDef qd = qual_defaults(P.qual_mode);
	cl_dna_coder* dna = nullptr; cl_qual_coder* qual = nullptr;
	ck(ctx, cl_dna_coder_create(ctx, P.c, P.level, 0, &dna), "cl_dna_coder_create");
	{
		cl_qual_params qp{}; qp.mode = P.qual_mode; qp.source = source; qp.level = P.level;
		qp.n_fwd = (uint32_t)qd.fwd.size(); std::copy(qd.fwd.begin(), qd.fwd.end(), qp.fwd);
		qp.n_rev = (uint32_t)qd.rev.size(); std::copy(qd.rev.begin(), qd.rev.end(), qp.rev);
		ck(qctx, cl_qual_coder_create(P.level == 1 ? qctx : ctx, &qp, &qual), "cl_qual_coder_create");
	}
	// the header stream on a host thread, next to the GPU path
	std::vector<std::vector<uint8_t>> hdr_parts; std::vector<uint32_t> hdr_counts; std::string hdr_err;
	std::thread hdr([&]() {
		cl_id_coder* idc = nullptr;
		if (cl_id_coder_create(0, &idc) != CL_OK) { hdr_err = "cl_id_coder_create"; return; }
		uint32_t i = 0;
		while (i < n)
		{
			uint32_t j = i; uint64_t acc = 0;
			while (j < n) { acc += in.id_off[j + 1] - in.id_off[j]; ++j; if (acc >= (2u << 21)) break; }       // in_reads.cpp:93-101
			std::vector<uint64_t> off(j - i + 1);
			for (uint32_t t = i; t <= j; ++t) off[t - i] = in.id_off[t] - in.id_off[i];
			std::vector<uint8_t> out(2 * (size_t)off.back() + 64); uint64_t got = 0;
			if (cl_id_encode_part(idc, in.ids.data() + in.id_off[i], off.data(), in.plus.data() + i, j - i, out.data(), out.size(), &got) != CL_OK) { hdr_err = cl_id_coder_error(idc); break; }
			out.resize(got); hdr_parts.push_back(std::move(out)); hdr_counts.push_back(j - i);
			i = j;
		}
		cl_id_coder_free(idc);
	});
	cl_compress_params cp{};
	cp.k = k; cp.f = P.f; cp.ci = P.ci; cp.cs = P.cs; cp.c = P.c; cp.anchor_len = a; cp.min_part_alt = P.min_part_alt; cp.max_rec = P.max_rec; cp.min_anchors = 1;
	cp.level = P.level; cp.source = source; cp.sparse = P.sparse; cp.sparse_g = P.g; cp.sparse_exponent = 1.0;
	cp.cost_mult = 1.0; cp.frac_always = 0.9; cp.frac_min = 0.5; cp.max_matches_mult = 10.0;
	const uint64_t total = in.bases.size();
	const uint64_t dna_cap = total + 64ull * n_parts + 4096, qual_cap = (uint64_t)(total * 1.35) + 64ull * n_parts + 4096;
	uint8_t* d_dna = nullptr; uint8_t* d_qual = nullptr;
	hipck(hipMalloc((void**)&d_dna, dna_cap), "hipMalloc"); hipck(hipMalloc((void**)&d_qual, qual_cap), "hipMalloc");
	std::vector<uint64_t> dna_sz(n_parts), qual_sz(n_parts); cl_compress_info info{};
	ck(ctx, cl_compress_shard(ctx, &cp, reads, with_qual ? d_quals : nullptr, with_qual ? d_off : nullptr, packs.data(), n_parts, packs.data(), n_parts, dna, with_qual ? qual : nullptr,
		d_dna, dna_cap, dna_sz.data(), d_qual, qual_cap, qual_sz.data(), &info), "cl_compress_shard");
	std::vector<uint8_t> h_dna(info.dna_bytes), h_qual(info.qual_bytes);
	if (info.dna_bytes) hipck(hipMemcpy(h_dna.data(), d_dna, info.dna_bytes, hipMemcpyDeviceToHost), "hipMemcpy");
	if (info.qual_bytes) hipck(hipMemcpy(h_qual.data(), d_qual, info.qual_bytes, hipMemcpyDeviceToHost), "hipMemcpy");
	hdr.join();
	if (!hdr_err.empty()) die("header stream: " + hdr_err);

	// archive: meta (compression.cpp:704-779), info (utils.cpp:326-342), then the stream parts
	ArchiveWriter ar; ar.open(pos[1]);
	const int s_meta = ar.reg("meta"), s_header = ar.reg("header"), s_dna = ar.reg("dna"), s_qual = ar.reg("qual");
	uint64_t o = 0; for (uint32_t p = 0; p < n_parts; ++p) { ar.add(s_dna, h_dna.data() + o, dna_sz[p], packs[p + 1] - packs[p]); o += dna_sz[p]; }
	o = 0; for (uint32_t p = 0; p < n_parts; ++p) { ar.add(s_qual, h_qual.data() + o, qual_sz[p], 0); o += qual_sz[p]; }
	for (size_t p = 0; p < hdr_parts.size(); ++p) ar.add(s_header, hdr_parts[p].data(), hdr_parts[p].size(), hdr_counts[p]);
	const uint64_t mean_read_len = (uint64_t)((double)(info.tot_kmers * P.f) / n + k - 1);
	uint32_t tot_ref = n;
	if (P.sparse) { std::vector<uint8_t> acc(n); ck(ctx, cl_ref_accept(n, 0, info.sparse_range, 1.0, acc.data()), "cl_ref_accept"); tot_ref = 0; for (uint8_t x : acc) tot_ref += x; }
	std::vector<uint8_t> meta;
	le<uint32_t>(meta, tot_ref); le<uint32_t>(meta, P.c); le<int32_t>(meta, P.level); meta.push_back((uint8_t)source);
	le<uint64_t>(meta, (uint64_t)n * mean_read_len);
	meta.push_back((uint8_t)P.qual_mode);
	if (P.qual_mode == 8 || (P.qual_mode >= 4 && P.qual_mode <= 6)) for (uint32_t v : qd.rev) le<uint32_t>(meta, v);
	meta.push_back(0);                                                   // HeaderComprMode::Original
	meta.push_back(P.sparse ? 1 : 0);                                    // ReferenceReadsMode: All = 0, Sparse = 1
	if (P.sparse) { le<uint32_t>(meta, info.sparse_range); le_double(meta, 1.0); }
	meta.push_back(0);                                                   // no reference genome
	ar.add(s_meta, meta.data(), meta.size(), 0);
	const int s_info = ar.reg("info");
	std::vector<uint8_t> inf;
	le<uint32_t>(inf, 1); le<uint32_t>(inf, 2); le<uint32_t>(inf, 1);                        // archive format of CoLoRd 1.2.1 (defs.h:24-26)
	le<uint64_t>(inf, in.file_bytes); le<uint64_t>(inf, total); le<uint32_t>(inf, n); le<uint64_t>(inf, (uint64_t)time(nullptr));
	std::string cmd; for (int i = 0; i < argc; ++i) { if (i) cmd += ' '; cmd += argv[i]; }
	le<uint32_t>(inf, (uint32_t)cmd.size()); inf.insert(inf.end(), cmd.begin(), cmd.end());
	ar.add(s_info, inf.data(), inf.size(), 0);
	ar.close();
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	fprintf(stderr, "colord_hip: %u reads, %llu bases, k=%u a=%u; dna %llu B, qual %llu B, header %zu parts; %llu reference reads; %.2f s\n", n, (unsigned long long)total, k, a,
		(unsigned long long)info.dna_bytes, (unsigned long long)info.qual_bytes, hdr_parts.size(), (unsigned long long)info.n_refs, sec);
	cl_qual_coder_free(qual); cl_dna_coder_free(dna); cl_reads_free(reads);
	(void)hipFree(d_off); (void)hipFree(d_quals); (void)hipFree(d_dna); (void)hipFree(d_qual);
	cl_ctx_destroy(qctx); cl_ctx_destroy(ctx);
	return 0;
}
