// decode.hip — the inverse path: CDNACoder::Decode (src/colord/dna_coder.cpp:234-437 and the field decoders
// :466-1240), CQualityCoder::Decode (quality_coder.cpp:605-657, quality_coder_impl.cpp decode_* incl. :506-559, :800-849),
// CIDCoder::Decode (id_coder.cpp:396-600), over CRangeDecoder + the adaptive models (host_coder.hpp).
//
// HOST code by nature: in a decoder the symbol just decoded selects the next context, so a model domain is ONE dependent
// chain (SURVEY §7 hard part 1) — there is nothing for 256 CUs to do; parallelism across streams (dna / qual / header run on
// three host threads in colord_hip decompress, as in decompression_common.cpp:318-337) and across model domains (archives
// written by several GPUs hold one domain per rank).  Contexts are identities: only the partition of the symbols into
// models matters, and it is the reference's (the same partition the device encoders use; byte-identical streams prove it).
#include "common.hpp"
#include "host_coder.hpp"
#include <cmath>
#include <random>
#include <string>
#include <vector>

using hostrc::Family; using hostrc::RangeDec;

namespace {
enum { T_INS = 0, T_DEL, T_MATCH, T_SUBST, T_ANCHOR, T_SKIP, T_ALT_ID, T_MAIN_REF, T_PLAIN, T_START_PLAIN, T_START_ES, T_START_PLAIN_N, T_NONE };   // utils.h:56-59
inline uint64_t ilog2_(uint64_t x) { uint64_t r = 0; for (; x; ++r) x >>= 1; return r; }                  // basic_coder.h:39-47
inline uint64_t no_bytes_(uint64_t x) { uint64_t r = 1; x >>= 8; for (; x; ++r) x >>= 8; return r; }      // :50-60
constexpr uint8_t FLAG_ANCHOR = 0x80, FLAG_MATCH = 0x40;                                                  // basic_coder.h:34-35
}

// ======================================================================================================================
// dna
// ======================================================================================================================
struct cl_dna_decoder {
	int level = 1; uint32_t max_alt = 1, cur_read_id = 0, n_pseudo = 0;
	bool accept_all = true; uint32_t range = 1; double exponent = 1.0;
	std::mt19937 mt; std::uniform_real_distribution<double> dist{ 0.0, 1.0 };       // CRefReadsAccepter (ref_reads_accepter.h:23-58)
	int no_tuples_in_mask = 2, no_symbols_in_mask = 5; uint64_t mask_tuple = 0, mask_symbol = 0;
	uint64_t ctx_read_type = 0, ctx_rev_comp = 0, ctx_tuple_type = 0, ctx_symbol = 0; int cur_ref_delta = 0;
	std::vector<std::pair<int, int>> rev_cache;                                        // uo_rev_comp of the current read
	Family m_read_type, m_rev_comp, m_seen, m_len_bits, m_len_data, m_symbols, m_symbols_n, m_read_id, m_read_id_short, m_anchor_len, m_skip_local, m_skip_distant, m_tuple_type;
	std::vector<std::vector<uint8_t>> refs;                                            // CReferenceReads: accepted reads, codes 0..3
	RangeDec rc; std::string err;
	std::vector<uint8_t> pending; std::vector<uint64_t> pending_off; bool has_pending = false;   // a decoded part the caller's buffer could not hold
	void init_models()
	{	// dna_coder.h:48-60 <symbols, MAX_TOTAL, ADDER>
		// (last argument: the width of the family's regular contexts — indexed directly, host_coder.hpp; contexts built from the guard symbol
		// 255 behind a reference read's last base lie above it and are hashed)
		const uint32_t T3 = 3 * (uint32_t)no_tuples_in_mask, S2 = 2 * (uint32_t)no_symbols_in_mask;
		m_rev_comp.init(2, 1u << 15, 1, 4); m_read_type.init(3, 1u << 15, 1, 8); m_seen.init(2, 1u << 15, 1, 7); m_len_bits.init(32, 1u << 18, 8, 1);
		m_len_data.init(256, 1u << 18, 8, 9); m_symbols.init(4, 1u << 10, 1, level == 1 ? 22 : level == 2 ? 23 : level == 3 ? 24 : 14); m_symbols_n.init(5, 1u << 10, 1, S2 < 8 ? 8 : S2); m_read_id.init(256, 1u << 13, 1, 11);
		m_skip_distant.init(256, 1u << 15, 1, 8); m_tuple_type.init(8, 1u << 15, 1, T3 + 9); m_read_id_short.init(max_alt, 1u << 13, 1, 7);
		m_anchor_len.init(24, 1u << 15, 1, 8); m_skip_local.init(256, 1u << 15, 1, 8);
		ctx_read_type = 0;
	}
	bool should_add(uint32_t idx)
	{
		if (idx < n_pseudo) return true;
		const uint32_t range_no = (idx - n_pseudo) / range;
		return dist(mt) <= std::pow(1.0 / (range_no + 1ul), exponent);
	}
	uint32_t ref_at(int id, int rev, int64_t pos) const            // GetRefRead(id, rev)[pos], 255 = the guard after the last base
	{
		const std::vector<uint8_t>& r = refs[id];
		if (pos < 0 || pos >= (int64_t)r.size()) return 255;
		return rev ? 3u - r[r.size() - 1 - pos] : r[pos];
	}
	uint32_t dec_read_id()                                            // :554-569
	{
		const int n = (int)no_bytes_(cur_read_id); uint32_t id = 0;
		for (int i = n - 1; i >= 0; --i) { const uint64_t add = (i == n - 2) ? id : 0; id = (id << 8) + m_read_id.decode(rc, (uint64_t)i + (add << 3)); }
		return id;
	}
	int dec_rev_comp(int read_id)                                     // :512-531
	{
		for (auto& p : rev_cache) if (p.first == read_id) return p.second;
		const int f = (int)m_rev_comp.decode(rc, ctx_rev_comp);
		rev_cache.push_back({ read_id, f });
		ctx_rev_comp = ((ctx_rev_comp << 2) + (uint64_t)f) & 0xf;
		return f;
	}
	uint32_t dec_read_len()                                           // :1059-1099
	{
		int nb = (int)m_len_bits.decode(rc, 0);
		if (nb < 2) return (uint32_t)nb;
		uint64_t ctx = (uint64_t)nb << 3;
		uint32_t len = 1u << (nb - 1);
		const uint32_t prefix = m_len_data.decode(rc, ctx);
		if (nb <= 9) return len + prefix;
		len += prefix << (nb - 9);
		nb -= 9; ctx += 1ULL << 2;
		for (uint32_t shift = 0; nb > 0; nb -= 8, shift += 8, ++ctx) len += m_len_data.decode(rc, ctx) << shift;
		return len;
	}
	uint32_t dec_tuple_type(uint32_t ref_symbol, uint32_t last, bool first)   // :720-769
	{
		uint32_t shift = 3 * no_tuples_in_mask; uint64_t ctx = ctx_tuple_type;
		ctx += (ctx_symbol & 0xf) << shift; shift += 4;
		ctx += (uint64_t)ref_symbol << shift; shift += 2;
		if (cur_ref_delta < -10) ctx += 1ULL << shift; else if (cur_ref_delta < -1) ctx += 2ULL << shift;
		else if (cur_ref_delta > 10) ctx += 3ULL << shift; else if (cur_ref_delta > 1) ctx += 4ULL << shift;
		// what cannot follow the tuple before (dna_coder.cpp:735-757) as a table: a mask of the excluded types per previous type (no branch to mispredict)
		static const uint8_t EXCL[16] = { /* INS */ 0, /* DEL */ 1u << T_SKIP, /* MATCH */ 1u << T_ANCHOR, /* SUBST */ 0, /* ANCHOR */ (1u << T_ANCHOR) | (1u << T_MATCH), /* SKIP */ (1u << T_DEL) | (1u << T_SKIP),
			/* ALT_ID */ (1u << T_ALT_ID) | (1u << T_MAIN_REF), /* MAIN_REF */ (1u << T_ALT_ID) | (1u << T_MAIN_REF), 0, 0, 0, 0, 0, 0, 0, 0 };
		const uint32_t f = m_tuple_type.decode_masked8(rc, ctx, first ? 0u : EXCL[last & 15]);
		ctx_tuple_type = ((ctx_tuple_type << 3) + f) & mask_tuple;
		return f;
	}
	uint64_t ctx_insertion(uint32_t base) const                      // :850-876
	{
		uint32_t shift = 2; uint64_t ctx = 2;
		if (level == 1) { ctx += (ctx_symbol & 0xff) << shift; shift += 8; }
		else if (level == 2) { ctx += (ctx_symbol & 0x3ff) << shift; shift += 10; }
		else { ctx += (ctx_symbol & 0x3ff) << shift; shift += 10; ctx += (uint64_t)(((ctx_symbol >> 10) & 3) == ((ctx_symbol >> 8) & 3)) << shift; ++shift; }
		ctx += (uint64_t)base << shift; shift += 2;
		return ctx + ((ctx_tuple_type & 0777) << shift);
	}
	uint64_t ctx_substitution(uint32_t base) const                   // :925-945
	{
		uint32_t shift = 2; uint64_t ctx = 1;
		ctx += (ctx_symbol & 0x3f) << shift; shift += 6;
		if (level == 3) { ctx += (uint64_t)(((ctx_symbol >> 6) & 3) == ((ctx_symbol >> 4) & 3)) << shift; ++shift; }
		ctx += (uint64_t)base << shift; shift += 2;
		return ctx + ((ctx_tuple_type & 07777) << shift);
	}
	// A corrupt stream must end in an error, not in unbounded output: run lengths are 28-bit in the tuple format (utils.h:53), a read
	// is bounded, and the continuation loops of the length codes stop at those bounds.
	static constexpr uint32_t MAX_RUN = 1u << 28; static constexpr uint64_t MAX_READ_BASES = 1ull << 31;
	bool bad_len = false;
	uint32_t dec_anchor_len() { uint32_t len = 0; for (uint64_t part = 0;; ++part) { const uint32_t x = m_anchor_len.decode(rc, part); if (x < 23) return len + x; len += 22; if (len > MAX_RUN) { bad_len = true; return 0; } } }   // :981-1001
	uint32_t dec_skip_len(bool local)                                // :1141-1175
	{
		uint32_t len = 0;
		if (local) { for (uint64_t part = 0;; ++part) { const uint32_t x = m_skip_local.decode(rc, part); if (x < 255) return len + x; len += 254; if (len > MAX_RUN) { bad_len = true; return 0; } } }
		for (int i = 3; i >= 0; --i) len = (len << 8) + m_skip_distant.decode(rc, (uint64_t)i * 64 + ilog2_(len));
		return len;
	}
	// one read: appends its bases (codes 0..4, with FLAG_MATCH / FLAG_ANCHOR at level > 1) to `out`
	bool decode_read(std::vector<uint8_t>& out);
};

bool cl_dna_decoder::decode_read(std::vector<uint8_t>& out)           // CDNACoder::Decode (:234-437)
{
	ctx_tuple_type = mask_tuple; ctx_symbol = mask_symbol; ctx_rev_comp = 0xf; rev_cache.clear();
	const uint32_t flag = m_read_type.decode(rc, ctx_read_type);      // decode_read_flag (:466-486)
	ctx_read_type = ((ctx_read_type << 2) + flag) & 0xff;
	const uint32_t read_len = dec_read_len();
	if (read_len > MAX_READ_BASES) { err = "dna stream: implausible read length"; return false; }
	bool accept = flag != 1;
	if (!accept_all) accept &= should_add(cur_read_id);               // (the draw happens for every read, `&=` does not short-circuit)
	const size_t o0 = out.size();
	if (flag == 0 || flag == 1)
	{
		for (uint32_t i = 0; i < read_len; ++i)
			if (flag == 0) { const uint32_t s = m_symbols.decode(rc, ctx_symbol << 2); ctx_symbol = ((ctx_symbol << 2) + s) & mask_symbol; out.push_back((uint8_t)s); }
			else { const uint32_t s = m_symbols_n.decode(rc, ctx_symbol); ctx_symbol = ((ctx_symbol << 4) + s) & mask_symbol; out.push_back((uint8_t)s); }
		++cur_read_id;
		if (accept) refs.emplace_back(out.begin() + o0, out.end());
		return true;
	}
	const uint8_t f_match = level > 1 ? FLAG_MATCH : 0, f_anchor = level > 1 ? FLAG_ANCHOR : 0;
	bool is_main = true;
	const int ref_id = (int)dec_read_id();
	if ((size_t)ref_id >= refs.size()) { err = "dna stream: reference read id " + std::to_string(ref_id) + " of read " + std::to_string(cur_read_id) + " does not exist"; return false; }
	const int ref_rev = dec_rev_comp(ref_id);
	int alt_id = -1, alt_rev = 0, alt_slot = -1;
	std::vector<int> alt_ids, alt_pos_of, alt_rev_of;                 // v_alt_ids, m_alt_pos; the orientation of an alternative is decoded once per read
	uint32_t last = T_NONE;
	int64_t ref_pos = 0, alt_pos = 0;
	cur_ref_delta = 0;
	std::vector<uint8_t> plain;                                       // read_without_flags
	plain.reserve(std::min<uint32_t>(read_len, 1u << 24));                 // (a decoded value: a hint, never trusted with memory)
	// level 1 carries no flags: the bases are collected once (`plain`) and appended to the part when the read is complete
	const bool flagged = level > 1;
	auto emit = [&](uint32_t b, uint8_t fl) { if (flagged) out.push_back((uint8_t)(b | fl)); plain.push_back((uint8_t)b); };
	// the two reference reads in use as (bases, length): one indirection less per symbol than refs[id][pos] (they do not move while this
	// read is decoded: `refs` grows only at its end)
	const uint8_t* mp = refs[ref_id].data(); const int64_t ml = (int64_t)refs[ref_id].size();
	const uint8_t* ap = nullptr; int64_t al = 0;
	auto at = [](const uint8_t* p, int64_t l, int rev, int64_t pos) -> uint32_t { return pos < 0 || pos >= l ? 255u : rev ? 3u - p[l - 1 - pos] : p[pos]; };
	for (uint32_t t_i = 0; t_i < read_len; ++t_i)
	{
		const uint32_t ref_symbol = is_main ? at(mp, ml, ref_rev, ref_pos) : at(ap, al, alt_rev, alt_pos);
		const uint32_t t = dec_tuple_type(ref_symbol, last, t_i == 0);
		if (bad_len || plain.size() > MAX_READ_BASES) { err = "dna stream: run or read longer than the format allows"; return false; }
		switch (t)
		{
		case T_ALT_ID:
		{
			if (!is_main && alt_slot >= 0) alt_pos_of[alt_slot] = (int)alt_pos;
			int slot = -1;                                                // decode_alt_read_id (:618-647)
			if (alt_ids.empty()) { alt_ids.push_back((int)dec_read_id()); alt_pos_of.push_back(0); alt_rev_of.push_back(-1); slot = 0; }
			else
			{
				const uint64_t seen = alt_ids.size();
				if (!m_seen.decode(rc, seen)) { alt_ids.push_back((int)dec_read_id()); alt_pos_of.push_back(0); alt_rev_of.push_back(-1); slot = (int)alt_ids.size() - 1; }
				else { slot = (int)m_read_id_short.decode(rc, seen); if ((size_t)slot >= alt_ids.size()) { err = "dna stream: bad short alternative id"; return false; } }
			}
			alt_id = alt_ids[slot]; alt_slot = slot;
			if ((size_t)alt_id >= refs.size()) { err = "dna stream: alternative reference id out of range"; return false; }
			alt_rev = dec_rev_comp(alt_id);
			ap = refs[alt_id].data(); al = (int64_t)refs[alt_id].size();
			alt_pos = 0; is_main = false; cur_ref_delta = 0;
			break;
		}
		case T_ANCHOR:
		{
			const uint32_t len = dec_anchor_len();
			{	// the anchor's bases in one sweep (the stretch inside the reference read; what lies beyond its ends reads as the guard)
				const uint8_t* p = is_main ? mp : ap; const int64_t l = is_main ? ml : al, pos = is_main ? ref_pos : alt_pos; const int rev = is_main ? ref_rev : alt_rev;
				const size_t q = plain.size();
				plain.resize(q + len);
				uint8_t* pq = plain.data() + q;
				if (pos >= 0 && pos + (int64_t)len <= l)
				{
					if (!rev) memcpy(pq, p + pos, len);
					else { const uint8_t* src = p + (l - 1 - pos); for (uint32_t i = 0; i < len; ++i) pq[i] = (uint8_t)(3u - *(src - i)); }
				}
				else for (uint32_t i = 0; i < len; ++i) pq[i] = (uint8_t)(at(p, l, rev, pos + i) & 0xff);
				if (flagged) { const size_t o = out.size(); out.resize(o + len); uint8_t* po = out.data() + o; for (uint32_t i = 0; i < len; ++i) po[i] = (uint8_t)(pq[i] | f_anchor); }
			}
			if (is_main) ref_pos += len; else alt_pos += len;
			cur_ref_delta = 0;
			for (int i = no_symbols_in_mask; i > 0; --i) ctx_symbol = (ctx_symbol << 2) + ((int64_t)plain.size() >= i ? plain[plain.size() - i] : 0);
			ctx_symbol &= mask_symbol;
			break;
		}
		case T_MATCH:
			emit(ref_symbol & 0xff, f_match);
			ctx_symbol = ((ctx_symbol << 2) + ref_symbol) & mask_symbol;
			if (is_main) ++ref_pos; else ++alt_pos;
			break;
		case T_INS:
		{
			const uint32_t x = m_symbols.decode(rc, ctx_insertion(ref_symbol));
			emit(x, 0);
			ctx_symbol = ((ctx_symbol << 2) + x) & mask_symbol;
			++cur_ref_delta;
			break;
		}
		case T_DEL:
			if (is_main) ++ref_pos; else ++alt_pos;
			--cur_ref_delta;
			break;
		case T_SUBST:
		{
			const uint32_t s = m_symbols.decode(rc, ctx_substitution(ref_symbol), ref_symbol < 4 ? (int)ref_symbol : -1);
			emit(s, 0);
			ctx_symbol = ((ctx_symbol << 2) + s) & mask_symbol;
			if (is_main) ++ref_pos; else ++alt_pos;
			break;
		}
		case T_SKIP:
		{
			int skip_len;
			if (!is_main && last == T_ALT_ID)
			{
				int mod = (int)dec_skip_len(false);
				if (mod == 0) mod = -(int)dec_skip_len(false);
				skip_len = mod + alt_pos_of[alt_slot];
			}
			else skip_len = (int)dec_skip_len(last != T_ALT_ID && last != T_NONE);
			cur_ref_delta -= skip_len;
			if (is_main) ref_pos += skip_len; else alt_pos += skip_len;
			break;
		}
		case T_MAIN_REF:
			if (alt_slot >= 0) alt_pos_of[alt_slot] = (int)alt_pos;
			is_main = true; cur_ref_delta = 0;
			break;
		default:
			err = "dna stream: unknown tuple type"; return false;
		}
		last = t;
	}
	if (!flagged) out.insert(out.end(), plain.begin(), plain.end());
	if (accept) refs.push_back(std::move(plain));
	++cur_read_id;
	return true;
}

extern "C" cl_status cl_dna_decoder_create(uint32_t max_alt_refs, int32_t level, uint32_t start_read_id, uint32_t n_pseudo,
                                           int32_t accept_all, uint32_t sparse_range, double sparse_exponent, cl_dna_decoder** out)
{
	if (!out || level < 1 || (level > 3 && level != 9) || max_alt_refs < 1 || (!accept_all && sparse_range == 0)) return CL_E_INVALID;
	cl_dna_decoder* d = new cl_dna_decoder();
	d->level = level; d->max_alt = max_alt_refs; d->cur_read_id = start_read_id; d->n_pseudo = n_pseudo;
	d->accept_all = accept_all != 0; d->range = sparse_range ? sparse_range : 1; d->exponent = sparse_exponent;
	switch (level) { case 3: d->no_tuples_in_mask = 4; d->no_symbols_in_mask = 8; break; case 2: d->no_tuples_in_mask = 3; d->no_symbols_in_mask = 7; break; case 1: d->no_tuples_in_mask = 2; d->no_symbols_in_mask = 5; break;
	default: d->no_tuples_in_mask = 1; d->no_symbols_in_mask = 1; }   // dna_coder.cpp:1253-1280; any other level (9: the stored reference genome, reference_genome.cpp:262,340) takes the last branch
	d->mask_tuple = (1ULL << (3 * d->no_tuples_in_mask)) - 1; d->mask_symbol = (1ULL << (2 * d->no_symbols_in_mask)) - 1;
	d->init_models();
	*out = d;
	return CL_OK;
}
extern "C" void cl_dna_decoder_free(cl_dna_decoder* d) { delete d; }
extern "C" const char* cl_dna_decoder_error(const cl_dna_decoder* d) { return d ? d->err.c_str() : "null decoder"; }
extern "C" cl_status cl_dna_decoder_add_ref(cl_dna_decoder* d, const uint8_t* h_codes, uint32_t len)
{
	if (!d || (len && !h_codes)) return CL_E_INVALID;
	d->refs.emplace_back(h_codes, h_codes + len);
	return CL_OK;
}
extern "C" cl_status cl_dna_decoder_new_domain(cl_dna_decoder* d)
{
	if (!d) return CL_E_INVALID;
	d->init_models();                                                  // fresh adaptive models; reference reads, read counter and acceptor stream go on
	return CL_OK;
}
extern "C" cl_status cl_dna_decode_part(cl_dna_decoder* d, const uint8_t* h_in, uint64_t n_in, uint32_t n_reads,
                                        uint8_t* h_bases, uint64_t cap, uint64_t* h_off, uint64_t* n_out)
{
	if (!d || !h_off || !n_out || (n_in && !h_in)) return CL_E_INVALID;
	if (n_in < 8) { d->err = "dna part shorter than the coder's 8 flush bytes"; return CL_E_INVALID; }
	std::vector<uint8_t>& out = d->pending;
	if (!d->has_pending)
	{
		d->rc.start(h_in, n_in);                                       // SetInput + Restart (entr_read.h:146-191)
		out.clear(); d->pending_off.assign(1, 0);
		for (uint32_t i = 0; i < n_reads; ++i)
		{
			if (!d->decode_read(out)) return CL_E_INVALID;
			d->pending_off.push_back(out.size());
		}
		d->has_pending = true;
	}
	// a part that does not fit stays decoded inside the decoder: the same call with a buffer of *n_out bytes fetches it
	*n_out = out.size();
	if (out.size() > cap || (!h_bases && out.size())) return CL_E_CAPACITY;
	if (d->pending_off.size() != (size_t)n_reads + 1) { d->err = "cl_dna_decode_part: n_reads differs from the pending part"; return CL_E_INVALID; }
	if (!out.empty()) memcpy(h_bases, out.data(), out.size());
	memcpy(h_off, d->pending_off.data(), d->pending_off.size() * 8);
	d->has_pending = false; out.clear();
	return CL_OK;
}

// ======================================================================================================================
// qual
// ======================================================================================================================
struct cl_qual_decoder {
	int mode = 0, source = 0, level = 1;
	uint32_t map_fwd[96] = {}, map_rev[96] = {}, quant[96] = {};
	uint32_t n_ctx_sym = 0, bits_per_sym = 0, ctx_bits = 0, n_bins = 0; uint64_t ctx_mask = 0; uint32_t n_sym = 0;
	Family sym, bytes; RangeDec rc; std::string err;
	void init_models()
	{
		if (mode == 0) sym.init(96, 1u << 20, 32, ctx_bits + 11); else sym.init(n_sym, 1u << 18, 8, ctx_bits + 10);      // quality_coder.h:36-39 (contexts: history + bases + flags)
		bytes.init(256, 1u << 18, 8);                                                        // :41
	}
	static uint64_t vs(uint8_t x) { return (uint64_t)(x & 3); }                               // valid_sym: N aliases A
	uint64_t flag_bits(uint8_t b) const { return level <= 1 ? 0 : (uint64_t)((b & FLAG_MATCH) != 0) | ((uint64_t)((b & FLAG_ANCHOR) != 0) << 1); }
	double dec_avg(uint64_t ctx_base)                                                         // quality_coder_impl.cpp:837-849
	{
		const uint32_t a1 = bytes.decode(rc, ctx_base), a2 = bytes.decode(rc, a1 + 0x100ULL);
		return (double)((a1 << 8) + a2) / 256.0;
	}
	void decode_read(const uint8_t* b, uint32_t len, uint8_t* out);
};
static void q_fill(uint32_t* a, int lo, int hi, uint32_t v) { for (int i = lo; i < hi && i < 96; ++i) a[i] = v; }

void cl_qual_decoder::decode_read(const uint8_t* b, uint32_t len, uint8_t* out)
{
	if (mode == 8) { for (uint32_t i = 0; i < len; ++i) out[i] = (uint8_t)(33 + map_rev[0]); return; }      // quality_coder.cpp:611-617
	uint64_t hist = ctx_mask;
	auto B = [&](uint32_t i) { return vs(b[i]); };
	if (mode == 0)                                                                            // decode_original (:88-135 contexts)
	{
		for (uint32_t i = 0; i < len; ++i)
		{
			uint64_t c = hist; uint32_t sh = ctx_bits;
			c += B(i) << sh; sh += 2;
			if (i > 0) c += B(i - 1) << sh;
			sh += 2;
			if (level == 3) { if (i > 1) c += B(i - 2) << sh; sh += 2; }
			else { if (i > 1) c += (uint64_t)(B(i - 2) == B(i - 1)) << sh; sh += 1; }
			if (i + 1 < len) c += B(i + 1) << sh;
			sh += 2;
			c += flag_bits(b[i]) << sh;
			const uint32_t v = map_rev[sym.decode(rc, c)];
			out[i] = (uint8_t)(v + 33);
			hist = ((hist << bits_per_sym) + quant[v]) & ctx_mask;
		}
		return;
	}
	if (mode == 7)                                                                            // decode_average (:800-817)
	{
		const double avg = dec_avg(0ULL); double as = 0.0, qs = 0.0;
		for (uint32_t i = 0; i < len; ++i) { as += avg; const uint32_t v = (uint32_t)(as - qs); qs += v; out[i] = (uint8_t)(v + 33); }
		return;
	}
	if (mode >= 1 && mode <= 3)                                                               // *-avg: error diffusion in IEEE double (:506-559)
	{
		double avg[5], as[5] = { 0, 0, 0, 0, 0 }, qs[5] = { 0, 0, 0, 0, 0 };
		uint64_t ctx_p = 0;
		for (uint32_t i = 0; i < n_bins; ++i) { avg[i] = dec_avg((1ULL << 30) + ((uint64_t)i << 24) + (ctx_p << 16)); ctx_p = (uint64_t)avg[i]; }
		uint64_t dna = len ? B(0) : 3;
		for (uint32_t i = 0; i < len; ++i)
		{
			dna <<= 2; if (i + 1 < len) dna += B(i + 1);
			dna &= 0xff;
			const uint32_t d = sym.decode(rc, hist + (dna << ctx_bits) + (flag_bits(b[i]) << (ctx_bits + 8)));
			as[d] += avg[d];
			const uint32_t v = (uint32_t)(as[d] - qs[d]);
			qs[d] += v;
			out[i] = (uint8_t)(v + 33);
			hist = ((hist << bits_per_sym) + d) & ctx_mask;
		}
		return;
	}
	for (uint32_t i = 0; i < len; ++i)                                                        // *-fix (:313-435)
	{
		uint64_t c = hist; uint32_t sh = ctx_bits;
		c += B(i) << sh; sh += 2;
		if (i > 0) c += B(i - 1) << sh;
		sh += 2;
		if (i > 1) c += B(i - 2) << sh;
		sh += 2;
		if (i + 1 < len) c += B(i + 1) << sh;
		sh += 2;
		c += flag_bits(b[i]) << sh;
		const uint32_t d = sym.decode(rc, c);
		out[i] = (uint8_t)(map_rev[d] + 33);
		hist = ((hist << bits_per_sym) + d) & ctx_mask;
	}
}

extern "C" cl_status cl_qual_decoder_create(const cl_qual_params* P, cl_qual_decoder** out)
{
	if (!P || !out || P->mode < 0 || P->mode > 8 || P->level < 1 || P->level > 3 || P->n_fwd > 8 || P->n_rev > 8) return CL_E_INVALID;
	cl_qual_decoder* q = new cl_qual_decoder();
	q->mode = P->mode; q->source = P->source; q->level = P->level;
	auto bins = [&](uint32_t n) {                                      // adjust_quality_map_symbols (quality_coder.cpp:250-270)
		q->n_bins = n;
		if (P->n_fwd + 1 >= n && n >= 2)
		{
			q_fill(q->map_fwd, 0, (int)P->fwd[0], 0);
			for (uint32_t bin = 1; bin + 1 < n; ++bin) q_fill(q->map_fwd, (int)P->fwd[bin - 1], (int)P->fwd[bin], bin);
			q_fill(q->map_fwd, (int)P->fwd[n - 2], 96, n - 1);
		}
		for (uint32_t i = 0; i < n && i < P->n_rev; ++i) q->map_rev[i] = P->rev[i];
	};
	switch (P->mode)                                                   // quality_coder.cpp:58-240
	{
	case 0:
	{	// previous-quality classes of the Original mode (:276-504)
		for (int i = 0; i < 96; ++i) q->map_fwd[i] = q->map_rev[i] = (uint32_t)i;
		static const int ont3[] = { 0, 1, 2, 4, 7, 11, 16, 22, 29, 37, 46, 56, 67, 79, 90, 96 }, ont12[] = { 0, 1, 2, 5, 10, 15, 20, 25, 35, 50, 70, 96 };
		static const int pb3[] = { 0, 1, 10, 20, 30, 39, 45, 51, 57, 63, 69, 75, 81, 87, 93, 94 }, pb12[] = { 0, 1, 15, 29, 41, 53, 63, 72, 80, 87, 93, 94 };
		const int* t; int n;
		if (P->source == 0) { if (P->level == 3) { t = ont3; n = 15; } else { t = ont12; n = 11; } }
		else { if (P->level == 3) { t = pb3; n = 15; } else { t = pb12; n = 11; } }
		for (int b = 0; b < n; ++b) q_fill(q->quant, t[b], t[b + 1], (uint32_t)b);
		if (P->source == 2) { for (int i = 0; i < 93; ++i) q->quant[i] += 1; q->quant[93] = 0; }
		q->bits_per_sym = 4; q->n_ctx_sym = 2; q->n_sym = 96; break;
	}
	case 1: case 4: bins(5); q->bits_per_sym = 3; q->n_ctx_sym = 3; q->n_sym = 5; break;
	case 2: case 5: bins(4); q->bits_per_sym = 3; q->n_ctx_sym = 3; q->n_sym = 4; break;
	case 3: case 6: bins(2); q->bits_per_sym = 2; q->n_ctx_sym = 6; q->n_sym = 2; break;
	case 7: q->bits_per_sym = 8; q->n_ctx_sym = 2; q->n_sym = 2; break;
	case 8: if (P->n_rev > 0) q->map_rev[0] = P->rev[0]; q->n_sym = 2; break;
	}
	q->ctx_bits = q->bits_per_sym * q->n_ctx_sym; q->ctx_mask = (1ULL << q->ctx_bits) - 1;
	q->init_models();
	*out = q;
	return CL_OK;
}
extern "C" void cl_qual_decoder_free(cl_qual_decoder* q) { delete q; }
extern "C" cl_status cl_qual_decoder_new_domain(cl_qual_decoder* q) { if (!q) return CL_E_INVALID; q->init_models(); return CL_OK; }
// h_bases / h_off: the output of cl_dna_decode_part for the same part (flags included: levels 2 and 3 read them)
extern "C" cl_status cl_qual_decode_part(cl_qual_decoder* q, const uint8_t* h_in, uint64_t n_in, const uint8_t* h_bases, const uint64_t* h_off, uint32_t n_reads, uint8_t* h_quals)
{
	if (!q || !h_off || (n_reads && h_off[n_reads] && (!h_bases || !h_quals))) return CL_E_INVALID;
	if (q->mode != 8) { if (n_in < 8 || !h_in) return CL_E_INVALID; q->rc.start(h_in, n_in); }
	for (uint32_t i = 0; i < n_reads; ++i) q->decode_read(h_bases + h_off[i], (uint32_t)(h_off[i + 1] - h_off[i]), h_quals + h_off[i]);
	return CL_OK;
}

// ======================================================================================================================
// header (read ids)
// ======================================================================================================================
struct cl_id_decoder {
	int32_t mode = 0; RangeDec rc;
	Family plus_id, flags, literal, same, same_len, plain;
	struct Token { uint8_t sep; uint32_t b, e; };
	std::vector<Token> prev; std::string id_prev; uint64_t ctx_flags = 0; std::string err;
	std::string pending; std::vector<uint64_t> pending_off; std::vector<uint8_t> pending_plus; bool has_pending = false;
	static bool is_literal(uint8_t c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '@'; }
	static void tokenize(const std::string& id, std::vector<Token>& v)            // id_coder.cpp:169-207 (every token is a literal: a_numeric is never set, :124-143)
	{
		v.clear(); uint32_t start = 0;
		for (uint32_t i = 0; i < id.size(); ++i) if (!is_literal((uint8_t)id[i])) { v.push_back(Token{ (uint8_t)id[i], start, i }); start = i + 1; }
		v.push_back(Token{ 0, start, (uint32_t)id.size() });
	}
	void decode_one(bool& plus, std::string& id)                                    // decompress_lossless (:396-590)
	{
		id.clear();
		plus = plus_id.decode(rc, 0) != 0;
		if (flags.decode(rc, ctx_flags) == 1)
		{
			ctx_flags = ((ctx_flags << 1) + 1) & 0xff;
			for (uint32_t i = 0; i < prev.size(); ++i)
			{
				const Token& p = prev[i];
				if (same.decode(rc, i) == 1) id.append(id_prev, p.b, p.e - p.b);
				else if (same_len.decode(rc, i) == 1)
					for (uint32_t j = 0; j < p.e - p.b; ++j)
					{
						const uint32_t d = literal.decode(rc, ctx_flags + (1ull << 32) + j + ((uint64_t)i << 40) + (1ull << 60));
						id.push_back(d ? (char)d : id_prev[j + p.b]);
					}
				else
					for (uint32_t j = 0;; ++j) { const uint32_t d = literal.decode(rc, ctx_flags + j + (1ull << 32) + ((uint64_t)i << 40)); if (!d) break; id.push_back((char)d); }
				id.push_back((char)p.sep);
			}
			if (!id.empty() && id.back() == 0) id.pop_back();
		}
		else
		{
			ctx_flags = (ctx_flags << 1) & 0xff;
			for (uint32_t i = 0;; ++i) { const uint32_t d = plain.decode(rc, i); if (d == 0) break; id.push_back((char)d); if (d == 0xA) break; }
		}
		tokenize(id, prev);
		id_prev = id;
	}
};
extern "C" cl_status cl_id_decoder_create(int32_t header_mode, cl_id_decoder** out)
{
	if (!out || header_mode < 0 || header_mode > 2) return CL_E_INVALID;
	cl_id_decoder* c = new cl_id_decoder(); c->mode = header_mode;
	c->plus_id.init(2, 1u << 15, 1, 1); c->flags.init(2, 1u << 15, 1, 8); c->literal.init(256, 1u << 20, 64); c->same.init(2, 1u << 15, 1, 8); c->same_len.init(2, 1u << 15, 1, 8); c->plain.init(128, 1u << 19, 32, 12);
	*out = c;
	return CL_OK;
}
extern "C" void cl_id_decoder_free(cl_id_decoder* c) { delete c; }
// One part of the `header` stream -> n ids back to back in h_ids (capacity cap), h_off[n+1], h_plus[n] (1 = the '+' line repeats
// the id).  Modes Main / None carry no id bytes: the ids come back empty, as from the reference (id_coder.cpp:113-121, 388-394).
extern "C" cl_status cl_id_decode_part(cl_id_decoder* c, const uint8_t* h_in, uint64_t n_in, uint32_t n, uint8_t* h_ids, uint64_t cap, uint64_t* h_off, uint8_t* h_plus, uint64_t* n_out)
{
	if (!c || !h_off || !n_out) return CL_E_INVALID;
	std::string& all = c->pending; std::string id;
	if (!c->has_pending)
	{
		all.clear(); c->pending_off.assign(1, 0); c->pending_plus.clear();
		if (c->mode == 0)
		{
			if (n_in < 8 || !h_in) return CL_E_INVALID;
			c->rc.start(h_in, n_in); c->ctx_flags = 0;                  // Restart (id_coder.cpp:80-90)
			for (uint32_t i = 0; i < n; ++i) { bool plus = false; c->decode_one(plus, id); all += id; c->pending_off.push_back(all.size()); c->pending_plus.push_back(plus); }
		}
		else for (uint32_t i = 0; i < n; ++i)
		{	// None: decompress_none yields "@" (id_coder.cpp:388-391); Main: decompress_instrument yields nothing (:593-596)
			if (c->mode == 2) all.push_back('@');
			c->pending_off.push_back(all.size()); c->pending_plus.push_back(0);
		}
		c->has_pending = true;
	}
	// as cl_dna_decode_part: a part that does not fit stays decoded; call again with *n_out bytes
	*n_out = all.size();
	if (all.size() > cap || (!h_ids && !all.empty())) return CL_E_CAPACITY;
	if (c->pending_off.size() != (size_t)n + 1) return CL_E_INVALID;
	if (!all.empty()) memcpy(h_ids, all.data(), all.size());
	memcpy(h_off, c->pending_off.data(), c->pending_off.size() * 8);
	if (h_plus) for (uint32_t i = 0; i < n; ++i) h_plus[i] = c->pending_plus[i];
	c->has_pending = false; all.clear();
	return CL_OK;
}
