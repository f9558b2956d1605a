// host_coder.hpp — host side of the interval coder: CRangeDecoder (src/colord/sub_rc.h:216-392) and the adaptive frequency
// models behind every context (rc.h:34-220 CSimpleModel, :225-480 CSimpleModelFixedSize, :487-764 CFenwickTreeModelFixedSize —
// three containers of the same counts: start at 1, +ADDER per coded symbol, halve-round-up at MAX_TOTAL; basic_coder.h:116-137:
// a context seen for the first time gets a copy of the all-ones template).
//
// Decoding is one dependent chain per model domain (the symbol decides the next context): it runs on the host, one thread per
// stream, exactly where the reference runs it (decompression_common.cpp:318-337).  Used by decode.hip only.
//
// Round 6 — one family, three forms of the same counts, chosen per model as the reference chooses its containers:
//   * context -> model: a DENSE index where the family's regular contexts are a small integer range (context_hm.h:254-417 keeps
//     dense vectors for those), open addressing only for what lies outside it (contexts built from a reference read's end-of-read
//     guard, the byte models' sparse 2^30-range contexts);
//   * alphabets of up to 8 symbols: the cumulative search unrolled at compile time (rc.h:225-480, meta_switch.h);
//   * alphabets of more than 16: counts in blocks of 16 with the block sums beside them, so a symbol of 256 is found in at most
//     16 + 16 steps instead of 256 (the reference's answer is a Fenwick tree, rc.h:487-764: same counts, same bytes).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace hostrc {

struct RangeDec {
	static constexpr uint64_t TOP = 0x00ffffffffffffULL, MASK = 0xff00000000000000ULL;
	uint64_t low = 0, range = 0, buffer = 0; const uint8_t* in = nullptr; uint64_t n = 0, pos = 0;
	uint8_t byte() { return pos < n ? in[pos++] : 0; }
	void start(const uint8_t* p, uint64_t len)                         // SetInput + Start (sub_rc.h:249-262)
	{
		in = p; n = len; pos = 0; buffer = 0;
		for (int i = 1; i <= 8; ++i) buffer |= (uint64_t)byte() << (64 - 8 * i);
		low = 0; range = MASK;
	}
	// GetCumFreq (:264-268) is `buffer / (range /= tot)`.  The quotient is only ever COMPARED with cumulative counts, and for integers
	// t <= floor(buffer / range) <=> t * range <= buffer (no overflow: t <= tot, so t * range <= the range before the division) — the
	// models below compare products instead and the second division, a third of a symbol's dependent chain, is never made.
	void scale(uint64_t tot) { range /= tot; }
	bool below(uint64_t t) const { return t * range <= buffer; }          // t <= the target
	uint64_t cum_freq(uint64_t tot) { return buffer / (range /= tot); }
	void update(uint64_t freq, uint64_t cum)                            // UpdateFrequency (:270-287)
	{
		const uint64_t r = cum * range;
		buffer -= r; low += r; range *= freq;
		if (range > TOP) return;
		// The reference's loop gives out one byte per step while range <= TOP and first cuts the range where the interval straddles a
		// top-byte boundary.  TOP is 2^48 - 1: k steps are due when range < 2^(56 - 8 k); none of them cuts iff low and low + range agree in
		// their top k bytes — nearly always: then the k steps are one shift and one unaligned load, without a data-dependent loop to mispredict.
		if (range)
		{
			const unsigned k = (unsigned)(__builtin_clzll(range) >> 3) - 1;      // 1..6 (range <= TOP: at least 16 leading zeros)
			if ((((low ^ (low + range)) >> (64 - 8 * k)) == 0) && pos + 8 <= n)
			{
				uint64_t w; memcpy(&w, in + pos, 8); w = __builtin_bswap64(w);     // the next 8 stream bytes, first byte on top
				buffer = (buffer << (8 * k)) | (w >> (64 - 8 * k));
				low <<= 8 * k; range <<= 8 * k; pos += k;
				return;
			}
		}
		while (range <= TOP)
		{
			if ((low ^ (low + range)) & MASK) { const uint64_t q = low; range = (q | TOP) - q; }
			buffer = (buffer << 8) + byte();
			low <<= 8; range <<= 8;
		}
	}
};

// one model family: context value -> [counters (n_sym) | total | block sums (alphabets > 16)], all models in one pool
struct Family {
	static constexpr uint32_t BLK = 16;
	uint32_t n_sym = 0, max_total = 0, adder = 0, n_blk = 0, stride = 0;
	std::vector<uint32_t> dense; uint64_t dense_n = 0;                    // context < dense_n: model index + 1 (0: not seen yet)
	bool direct = false;                                                  // ... or, where all of them are a few MB, the models themselves in context order
	std::vector<uint64_t> keys; std::vector<uint32_t> vals; uint64_t mask = 0, used = 0;   // the others: open addressing
	std::vector<uint32_t> pool; uint32_t n_models = 0;
	// dense_bits: contexts below 2^dense_bits are indexed directly (0: none); any context is legal either way
	void init(uint32_t n, uint32_t mt, uint32_t ad, uint32_t dense_bits = 0)
	{
		n_sym = n; max_total = mt; adder = ad;
		n_blk = n > BLK ? (n + BLK - 1) / BLK : 0; stride = n + 1 + n_blk;
		dense_n = dense_bits ? 1ULL << (dense_bits > 25 ? 25 : dense_bits) : 0;        // (at most 128 MB of index; what lies above is hashed)
		reset();
	}
	void reset()
	{
		keys.assign(1024, ~0ULL); vals.assign(1024, 0); mask = 1023; used = 0; pool.clear(); n_models = 0;
		direct = dense_n && dense_n * stride * 4 <= (32ull << 20);
		if (direct) { dense.clear(); pool.reserve(dense_n * stride + 8192); for (uint64_t i = 0; i < dense_n; ++i) fresh(); }
		else dense.assign(dense_n, 0u);
	}
	static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
	uint32_t fresh()                                                      // a copy of the all-ones template (basic_coder.h:121-123)
	{
		const size_t at = pool.size();
		if (pool.capacity() < at + stride) pool.reserve(pool.capacity() < 4096 ? 8192 : pool.capacity() * 2);
		pool.resize(at + stride, 1u);
		uint32_t* m = pool.data() + at;
		m[n_sym] = n_sym;
		for (uint32_t b = 0; b < n_blk; ++b) m[n_sym + 1 + b] = (b + 1) * BLK <= n_sym ? BLK : n_sym - b * BLK;
		return n_models++;
	}
	uint32_t* model(uint64_t ctx)
	{
		if (ctx < dense_n)
		{
			if (direct) return pool.data() + ctx * stride;
			uint32_t& slot = dense[ctx];
			if (!slot) slot = fresh() + 1;
			return pool.data() + (uint64_t)(slot - 1) * stride;
		}
		uint64_t h = mix(ctx) & mask;
		while (keys[h] != ~0ULL && keys[h] != ctx) h = (h + 1) & mask;
		if (keys[h] == ~0ULL)
		{
			const uint32_t idx = fresh();
			keys[h] = ctx; vals[h] = idx;
			if (++used * 2 > mask + 1) grow();
			return pool.data() + (uint64_t)idx * stride;
		}
		return pool.data() + (uint64_t)vals[h] * stride;
	}
	void grow()
	{
		std::vector<uint64_t> ok; std::vector<uint32_t> ov; ok.swap(keys); ov.swap(vals);
		mask = 2 * (mask + 1) - 1; keys.assign(mask + 1, ~0ULL); vals.assign(mask + 1, 0);
		for (size_t i = 0; i < ok.size(); ++i) if (ok[i] != ~0ULL)
		{
			uint64_t h = mix(ok[i]) & mask;
			while (keys[h] != ~0ULL) h = (h + 1) & mask;
			keys[h] = ok[i]; vals[h] = ov[i];
		}
	}
	// +ADDER, halve-round-up at MAX_TOTAL (rc.h:233-244,347-358)
	inline void bump(uint32_t* m, uint32_t sym)
	{
		m[sym] += adder; m[n_sym] += adder;
		if (n_blk) m[n_sym + 1 + sym / BLK] += adder;
		if (m[n_sym] >= max_total) rescale(m);
	}
	void rescale(uint32_t* m)
	{
		while (m[n_sym] >= max_total) { uint32_t s = 0; for (uint32_t i = 0; i < n_sym; ++i) { m[i] = (m[i] + 1) / 2; s += m[i]; } m[n_sym] = s; }
		for (uint32_t b = 0; b < n_blk; ++b) { uint32_t s = 0; const uint32_t e = (b + 1) * BLK < n_sym ? (b + 1) * BLK : n_sym; for (uint32_t i = b * BLK; i < e; ++i) s += m[i]; m[n_sym + 1 + b] = s; }
	}
	template<uint32_t N> inline uint32_t decode_fixed(RangeDec& rc, uint32_t* m)
	{
		rc.scale(m[N]);
		uint32_t sym = 0, cum = 0, t = 0;                                      // the symbols wholly below the target are counted: no branch on the data
#pragma GCC unroll 8
		for (uint32_t i = 0; i + 1 < N; ++i) { t += m[i]; const bool past = rc.below(t); sym += past; cum = past ? t : cum; }
		rc.update(m[sym], cum);
		bump(m, sym);
		return sym;
	}
	// the same with up to two symbols left out: their counts read as 0
	template<uint32_t N> inline uint32_t decode_excl_fixed(RangeDec& rc, uint32_t* m, int exc1, int exc2)
	{
		uint32_t c[N], tot = 0;
#pragma GCC unroll 8
		for (uint32_t i = 0; i < N; ++i) { c[i] = ((int)i == exc1 || (int)i == exc2) ? 0u : m[i]; tot += c[i]; }
		const uint64_t target = rc.cum_freq(tot);
		uint32_t sym = N, cum = 0, t = 0;
#pragma GCC unroll 8
		for (uint32_t i = 0; i < N; ++i) { const uint32_t nt = t + c[i]; if (nt > target && sym == N) { sym = i; cum = t; } t = nt; }
		if (sym == N) { sym = N - 1; while (sym && ((int)sym == exc1 || (int)sym == exc2)) --sym; cum = 0; }   // (a target beyond the total: corrupt input)
		rc.update(m[sym], cum);
		bump(m, sym);
		return sym;
	}
	// ... the excluded symbols as a bit mask (8-symbol models: the DNA coder's tuple types)
	inline uint32_t decode_masked8(RangeDec& rc, uint64_t ctx, uint32_t excl)
	{
		uint32_t* m = model(ctx);
		uint32_t c[8], tot = 0;
#pragma GCC unroll 8
		for (uint32_t i = 0; i < 8; ++i) { c[i] = m[i] & (((excl >> i) & 1u) - 1u); tot += c[i]; }
		rc.scale(tot);
		uint32_t sym = 0, cum = 0, t = 0;
#pragma GCC unroll 8
		for (uint32_t i = 0; i < 7; ++i) { t += c[i]; const bool past = rc.below(t); sym += past; cum = past ? t : cum; }   // symbols wholly below the target
		if (sym == 7 && (excl >> 7 & 1u)) { sym = 6; while (sym && (excl >> sym & 1u)) --sym; cum = 0; }                      // (corrupt input only)
		rc.update(m[sym], cum);
		bump(m, sym);
		return sym;
	}
	// Decode (rc.h:850-878): no exclusions
	inline uint32_t decode(RangeDec& rc, uint64_t ctx)
	{
		uint32_t* m = model(ctx);
		switch (n_sym)
		{
		case 2: return decode_fixed<2>(rc, m);
		case 3: return decode_fixed<3>(rc, m);
		case 4: return decode_fixed<4>(rc, m);
		case 5: return decode_fixed<5>(rc, m);
		case 8: return decode_fixed<8>(rc, m);
		default: break;
		}
		const uint64_t target = rc.cum_freq(m[n_sym]);
		uint64_t t = 0; uint32_t i = 0;
		if (n_blk)
		{	// the block, then the symbol inside it
			const uint32_t* bs = m + n_sym + 1; uint32_t b = 0;
			while (b + 1 < n_blk && t + bs[b] <= target) { t += bs[b]; ++b; }
			i = b * BLK;
		}
		uint32_t sym = n_sym - 1, cum = 0;
		for (; i < n_sym; ++i) { const uint64_t nt = t + m[i]; if (nt > target) { sym = i; cum = (uint32_t)t; break; } t = nt; }
		if (i >= n_sym) cum = (uint32_t)(t - m[sym]);                        // (a target beyond the total: the last symbol, as the linear search of the reference ends)
		rc.update(m[sym], cum);
		bump(m, sym);
		return sym;
	}
	// DecodeExcluding (rc.h:926-1043): cumulative and total skip the excluded symbols (small alphabets only)
	uint32_t decode(RangeDec& rc, uint64_t ctx, int exc1, int exc2 = -1)
	{
		if (exc1 < 0 && exc2 < 0) return decode(rc, ctx);
		uint32_t* m = model(ctx);
		if (n_sym == 8) return decode_excl_fixed<8>(rc, m, exc1, exc2);
		if (n_sym == 4) return decode_excl_fixed<4>(rc, m, exc1, exc2);
		uint32_t tot = m[n_sym];
		if (exc1 >= 0) tot -= m[exc1];
		if (exc2 >= 0) tot -= m[exc2];
		const uint64_t target = rc.cum_freq(tot);
		uint64_t t = 0; uint32_t sym = n_sym - 1, cum = 0; bool found = false;
		for (uint32_t i = 0; i < n_sym; ++i)
		{
			if ((int)i == exc1 || (int)i == exc2) continue;
			t += m[i];
			if (t > target) { sym = i; cum = (uint32_t)(t - m[i]); found = true; break; }
		}
		(void)found;
		rc.update(m[sym], cum);
		bump(m, sym);
		return sym;
	}
};

} // namespace hostrc
