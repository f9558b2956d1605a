// id_coder.hip — the `header` stream (read ids): CIDCoder + CEntrComprHeaders (src/colord/id_coder.{h,cpp},
// entr_header.cpp:23-45).  HOST code: the stream is one adaptive chain over a few bytes per read (0.5 % of the archive),
// it runs on a host thread next to the GPU path exactly as the reference runs it next to its other threads.
//
// What the reference coder does (id_coder.cpp:169-385): an id is split into tokens at every character that is not a
// letter, a digit or '@' (the separator is part of the token).  If the token structure (count, separators) equals the
// previous id's, a flag 1 is coded and every token is coded against its predecessor: same / same length (then per
// character "equal" = 0 or the character) / different length (characters + terminator).  Otherwise flag 0 and the id
// as plain characters + terminator.  (`a_numeric` is never set, id_coder.cpp:124-143, so the numeric-delta branch of
// the source is dead code and every token is a literal.)  Models: adaptive frequency counts per context, counters
// start at 1, +ADDER per coded symbol, halve-round-up at MAX_TOTAL (rc.h:225-480,487-764; the Fenwick-tree variant
// keeps the same counts).  Range coder: sub_rc.h:44-212.  The interval arithmetic restarts per part; models and the
// previous id persist (entr_header.cpp:35-38).
#include "common.hpp"
#include <string>
#include <unordered_map>
#include <vector>

namespace {
struct RangeEnc {                                                    // CRangeEncoder (sub_rc.h:44-212)
	static constexpr uint64_t TOP = 0x00ffffffffffffULL, MASK = 0xff00000000000000ULL;
	uint64_t low = 0, range = MASK; std::vector<uint8_t>* out = nullptr;
	void start() { low = 0; range = MASK; }
	void encode(uint64_t freq, uint64_t cum, uint64_t tot)
	{
		range /= tot; low += range * cum; range *= freq;
		while (range <= TOP)
		{
			if ((low ^ (low + range)) & MASK) { const uint64_t r = low; range = (r | TOP) - r; }
			out->push_back((uint8_t)(low >> 56));
			low <<= 8; range <<= 8;
		}
	}
	void end() { for (int i = 0; i < 8; ++i) { out->push_back((uint8_t)(low >> 56)); low <<= 8; } }
};
// one model family: context -> counters (find_rc_context: a new context is a copy of the all-ones template, basic_coder.h:116-137)
struct Family {
	uint32_t n_sym, max_total, adder;
	std::unordered_map<uint64_t, std::vector<uint32_t>> ctx;      // counters..., total
	void encode(RangeEnc& rc, uint64_t c, uint32_t sym)
	{
		auto it = ctx.find(c);
		if (it == ctx.end()) { it = ctx.emplace(c, std::vector<uint32_t>(n_sym + 1, 1u)).first; it->second[n_sym] = n_sym; }
		std::vector<uint32_t>& m = it->second;
		uint64_t cum = 0; for (uint32_t i = 0; i < sym; ++i) cum += m[i];
		rc.encode(m[sym], cum, m[n_sym]);
		m[sym] += adder; m[n_sym] += adder;
		while (m[n_sym] >= max_total) { uint32_t t = 0; for (uint32_t i = 0; i < n_sym; ++i) { m[i] = (m[i] + 1) / 2; t += m[i]; } m[n_sym] = t; }
	}
};
struct Token { uint8_t sep; uint32_t b, e; };                        // all tokens are literals (see the header comment)
} // namespace

struct cl_id_coder {
	int32_t mode = 0;                                                 // HeaderComprMode: 0 Original, 1 Main, 2 None (params.h:44)
	RangeEnc rc; std::vector<uint8_t> out;
	Family plus_id{ 2, 1u << 15, 1 }, flags{ 2, 1u << 15, 1 }, literal{ 256, 1u << 20, 64 }, same{ 2, 1u << 15, 1 }, same_len{ 2, 1u << 15, 1 }, plain{ 128, 1u << 19, 32 };
	std::vector<Token> prev, cur; std::string id_prev; uint64_t ctx_flags = 0;
	std::string err;
};

static inline bool id_is_literal(uint8_t c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '@'; }
static void id_tokenize(const uint8_t* id, uint32_t n, std::vector<Token>& v)       // id_coder.cpp:169-207
{
	v.clear();
	uint32_t start = 0;
	for (uint32_t i = 0; i < n; ++i) if (!id_is_literal(id[i])) { v.push_back(Token{ id[i], start, i }); start = i + 1; }
	v.push_back(Token{ 0, start, n });
}
static void id_encode_one(cl_id_coder* C, bool plus, const uint8_t* id, uint32_t n)   // compress_lossless (id_coder.cpp:210-385)
{
	id_tokenize(id, n, C->cur);
	C->plus_id.encode(C->rc, 0, plus ? 1u : 0u);
	bool same_types = C->cur.size() == C->prev.size();
	for (size_t i = 0; same_types && i < C->cur.size(); ++i) same_types = C->cur[i].sep == C->prev[i].sep;
	if (same_types)
	{
		C->flags.encode(C->rc, C->ctx_flags, 1);
		C->ctx_flags = ((C->ctx_flags << 1) + 1) & 0xff;
		const uint8_t* pid = (const uint8_t*)C->id_prev.data();
		for (uint32_t i = 0; i < C->cur.size(); ++i)
		{
			const Token& t = C->cur[i]; const Token& p = C->prev[i];
			const uint32_t len = t.e - t.b;
			const bool same_length = len == p.e - p.b;
			const bool same = same_length && std::equal(id + t.b, id + t.e, pid + p.b);
			if (same) { C->same.encode(C->rc, i, 1); continue; }
			C->same.encode(C->rc, i, 0);
			if (same_length)
			{
				C->same_len.encode(C->rc, i, 1);
				for (uint32_t j = 0; j < len; ++j)
				{	// prev_eq stays true in the source (its updates are commented out)
					const uint64_t c = C->ctx_flags + (1ull << 32) + j + ((uint64_t)i << 40) + (1ull << 60);
					C->literal.encode(C->rc, c, id[t.b + j] == pid[p.b + j] ? 0u : id[t.b + j]);
				}
			}
			else
			{
				C->same_len.encode(C->rc, i, 0);
				for (uint32_t j = 0; j < len; ++j) C->literal.encode(C->rc, C->ctx_flags + j + (1ull << 32) + ((uint64_t)i << 40), id[t.b + j]);
				C->literal.encode(C->rc, C->ctx_flags + len + (1ull << 32) + ((uint64_t)i << 40), 0);
			}
		}
	}
	else
	{
		C->flags.encode(C->rc, C->ctx_flags, 0);
		C->ctx_flags = (C->ctx_flags << 1) & 0xff;
		for (uint32_t i = 0; i < n; ++i) C->plain.encode(C->rc, i, id[i]);
		C->plain.encode(C->rc, n, 0);
	}
	C->prev.swap(C->cur);
	C->id_prev.assign((const char*)id, n);
}

extern "C" cl_status cl_id_coder_create(int32_t header_mode, cl_id_coder** out)
{
	if (!out || header_mode < 0 || header_mode > 2) return CL_E_INVALID;
	cl_id_coder* C = new cl_id_coder(); C->mode = header_mode; C->rc.out = &C->out; C->rc.start();
	*out = C;
	return CL_OK;
}
extern "C" void cl_id_coder_free(cl_id_coder* c) { delete c; }
extern "C" const char* cl_id_coder_error(const cl_id_coder* c) { return c ? c->err.c_str() : ""; }
// CEntrComprHeaders::Compress for one pack of headers (entr_header.cpp:30-44): ids back to back (without the leading '@' /
// '>'), h_off n+1 offsets, h_plus[i] != 0 when the '+' line repeats the id.  The part's archive metadata is n.
extern "C" cl_status cl_id_encode_part(cl_id_coder* C, const uint8_t* h_ids, const uint64_t* h_off, const uint8_t* h_plus, uint32_t n,
                                       uint8_t* h_out, uint64_t cap, uint64_t* n_out)
{
	if (!C || !n_out || (n && (!h_ids || !h_off))) return CL_E_INVALID;
	if (C->mode == 0)
		for (uint32_t i = 0; i < n; ++i)
		{
			const uint8_t* id = h_ids + h_off[i]; const uint64_t len = h_off[i + 1] - h_off[i];
			for (uint64_t j = 0; j < len; ++j) if (id[j] >= 128 || id[j] == 0) { C->err = "read id " + std::to_string(i) + " holds a byte outside 1..127"; return CL_E_UNSUPPORTED; }
			id_encode_one(C, h_plus && h_plus[i], id, (uint32_t)len);
		}
	// Main ("instrument") codes nothing in the reference either (compress_instrument is empty), None skips the ids
	C->rc.end();
	*n_out = C->out.size();
	if (C->out.size() > cap || !h_out) { C->err = "output capacity (2 bytes per id byte + 64 always suffice); the coder state is spent"; return CL_E_CAPACITY; }
	memcpy(h_out, C->out.data(), C->out.size());
	C->out.clear();
	C->rc.start(); C->ctx_flags = 0;                                  // Restart (id_coder.cpp:80-90)
	return CL_OK;
}
