// genome.hip — host pieces of the reference-genome mode (`-G`, config 4 of BASELINE.json): the `ref-genome` stream and the
// checksum that stands in for it (src/colord/reference_genome.cpp:29-104,205-213,235-279,325-370; compression.cpp:764-776).
//
//  * cl_genome_encode / cl_genome_decode: CReferenceGenome::Store(archive) codes every sequence as a PLAIN read (start_plain +
//    one plain tuple per base) with CDNACoder at "level 9" — not one of the coder's levels, so Init takes its last branch: one
//    tuple and ONE symbol of context (dna_coder.cpp:1253-1280) — in one part whose metadata is the number of sequences.  A plain
//    read is a read flag (encode_read_flag :443-463), its length (encode_read_len :1004-1057) and its symbols under the
//    previous symbol (encode_symbol_plain).  Host code: one dependent chain, a fraction of a second per 100 Mbases.  The decoder
//    is the library's DNA decoder at that level.
//  * cl_genome_md5: without -s the archive's `meta` stream carries the MD5 of the sequences in the reference's packed form
//    (packSeq: four bases a byte, the count of the last byte's symbols appended) and the decompressor refuses another genome.
//    MD5 itself is RFC 1321.
#include "common.hpp"
#include <unordered_map>

extern "C" cl_status cl_dna_decoder_create(uint32_t, int32_t, uint32_t, uint32_t, int32_t, uint32_t, double, cl_dna_decoder**);

namespace {
struct RangeEnc {                                                    // CRangeEncoder (sub_rc.h:44-212)
	static constexpr uint64_t TOP = 0x00ffffffffffffULL, MASK = 0xff00000000000000ULL;
	uint64_t low = 0, range = MASK; std::vector<uint8_t> out;
	void encode(uint64_t freq, uint64_t cum, uint64_t tot)
	{
		range /= tot; low += range * cum; range *= freq;
		while (range <= TOP)
		{
			if ((low ^ (low + range)) & MASK) { const uint64_t r = low; range = (r | TOP) - r; }
			out.push_back((uint8_t)(low >> 56));
			low <<= 8; range <<= 8;
		}
	}
	void end() { for (int i = 0; i < 8; ++i) { out.push_back((uint8_t)(low >> 56)); low <<= 8; } }
};
struct Model {                                                       // counters start at 1, +ADDER, halve-round-up at MAX_TOTAL (rc.h:233-244,347-358)
	uint32_t n_sym, max_total, adder;
	std::unordered_map<uint64_t, std::vector<uint32_t>> ctx;
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
inline uint32_t bit_length(uint32_t x) { uint32_t r = 0; for (; x; ++r) x >>= 1; return r; }      // ilog2 of basic_coder.h:39-47

// RFC 1321
struct Md5 {
	uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u; uint64_t n = 0; uint8_t buf[64]; uint32_t fill = 0;
	static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
	void block(const uint8_t* p)
	{
		static const uint32_t K[64] = {
			0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
			0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
			0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
			0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391 };
		static const int S[64] = { 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
			4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21 };
		uint32_t w[16]; for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
		uint32_t A = a, B = b, C = c, D = d;
		for (int i = 0; i < 64; ++i)
		{
			uint32_t F; int g;
			if (i < 16) { F = (B & C) | (~B & D); g = i; }
			else if (i < 32) { F = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
			else if (i < 48) { F = B ^ C ^ D; g = (3 * i + 5) & 15; }
			else { F = C ^ (B | ~D); g = (7 * i) & 15; }
			F += A + K[i] + w[g];
			A = D; D = C; C = B; B += rol(F, S[i]);
		}
		a += A; b += B; c += C; d += D;
	}
	void update(const uint8_t* p, uint64_t len)
	{
		n += len;
		while (len) { const uint32_t k = (uint32_t)std::min<uint64_t>(64 - fill, len); memcpy(buf + fill, p, k); fill += k; p += k; len -= k; if (fill == 64) { block(buf); fill = 0; } }
	}
	void finish(uint8_t out[16])
	{
		const uint64_t bits = n * 8; const uint8_t one = 0x80, zero = 0;
		update(&one, 1);
		while (fill != 56) update(&zero, 1);
		uint8_t l[8]; for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bits >> (8 * i));
		update(l, 8);
		const uint32_t v[4] = { a, b, c, d };
		for (int i = 0; i < 16; ++i) out[i] = (uint8_t)(v[i >> 2] >> (8 * (i & 3)));
	}
};
} // namespace

// h_codes: the sequences' bases (0..3) back to back, h_off: n_seqs + 1 offsets.  One part; its archive metadata is n_seqs.
extern "C" cl_status cl_genome_encode(const uint8_t* h_codes, const uint64_t* h_off, uint32_t n_seqs, uint8_t* h_out, uint64_t cap, uint64_t* n_out)
{
	if (!h_off || !n_out || (n_seqs && h_off[n_seqs] && !h_codes)) return CL_E_INVALID;
	RangeEnc rc;
	Model read_type{ 3, 1u << 15, 1, {} }, len_bits{ 32, 1u << 18, 8, {} }, len_data{ 256, 1u << 18, 8, {} }, symbols{ 4, 1u << 10, 1, {} };     // dna_coder.h:48-60
	uint64_t ctx_read_type = 0;
	for (uint32_t s = 0; s < n_seqs; ++s)
	{
		const uint64_t len64 = h_off[s + 1] - h_off[s];
		if (len64 >= (1ull << 32)) return CL_E_UNSUPPORTED;
		uint32_t len = (uint32_t)len64;
		read_type.encode(rc, ctx_read_type, 0);                                   // start_plain
		ctx_read_type = ((ctx_read_type << 2) + 0) & 0xff;
		int nb = (int)bit_length(len);
		len_bits.encode(rc, 0, (uint32_t)nb);
		if (nb >= 2)
		{
			uint64_t ctx = (uint64_t)nb << 3;
			len -= 1u << (nb - 1);
			uint32_t prefix = len, suffix = 0;
			if (nb > 9) { prefix = len >> (nb - 9); suffix = len - (prefix << (nb - 9)); }
			len_data.encode(rc, ctx, prefix);
			if (nb > 9)
			{
				nb -= 9; ctx += 1ull << 2;
				for (; nb > 0; nb -= 8) { len_data.encode(rc, ctx, suffix & 0xff); suffix >>= 8; ++ctx; }
			}
		}
		uint64_t ctx_symbol = 3;                                                   // ctx_mask_symbol of the last branch: one symbol
		for (uint64_t i = h_off[s]; i < h_off[s + 1]; ++i)
		{
			const uint32_t b = h_codes[i] & 3;
			symbols.encode(rc, ctx_symbol << 2, b);
			ctx_symbol = ((ctx_symbol << 2) + b) & 3;
		}
	}
	rc.end();
	*n_out = rc.out.size();
	if (rc.out.size() > cap || !h_out) return CL_E_CAPACITY;
	memcpy(h_out, rc.out.data(), rc.out.size());
	return CL_OK;
}

// the inverse: the part of the `ref-genome` stream -> n_seqs sequences (codes 0..3) back to back; CL_E_CAPACITY with the size needed
extern "C" cl_status cl_genome_decode(const uint8_t* h_in, uint64_t n_in, uint32_t n_seqs, uint8_t* h_codes, uint64_t cap, uint64_t* h_off, uint64_t* n_out)
{
	if (!h_in || !h_off || !n_out) return CL_E_INVALID;
	cl_dna_decoder* d = nullptr;
	CL_TRY(cl_dna_decoder_create(1, 9, 0, 0, 1, 1, 1.0, &d));
	const cl_status s = cl_dna_decode_part(d, h_in, n_in, n_seqs, h_codes, cap, h_off, n_out);
	cl_dna_decoder_free(d);
	return s;
}

// MD5 of the sequences in the reference's packed form (reference_genome.cpp:29-67,205-213): per sequence four bases a byte, first
// base in the high bits, a last partial byte left-aligned, then one byte holding the number of symbols in it (0 = none)
extern "C" cl_status cl_genome_md5(const uint8_t* h_codes, const uint64_t* h_off, uint32_t n_seqs, uint8_t* h_md5_16)
{
	if (!h_off || !h_md5_16) return CL_E_INVALID;
	Md5 md;
	std::vector<uint8_t> packed;
	for (uint32_t s = 0; s < n_seqs; ++s)
	{
		const uint8_t* p = h_codes + h_off[s]; const uint64_t n = h_off[s + 1] - h_off[s];
		packed.assign((n + 3) / 4 + 1, 0);
		const uint64_t full = n / 4;
		for (uint64_t i = 0; i < full; ++i) packed[i] = (uint8_t)((p[4 * i] << 6) + (p[4 * i + 1] << 4) + (p[4 * i + 2] << 2) + p[4 * i + 3]);
		const uint32_t last = (uint32_t)(n % 4);
		for (uint32_t j = 0; j < last; ++j) packed[packed.size() - 2] += (uint8_t)(p[4 * full + j] << (6 - 2 * j));
		packed.back() = (uint8_t)last;
		md.update(packed.data(), packed.size());
	}
	md.finish(h_md5_16);
	return CL_OK;
}
