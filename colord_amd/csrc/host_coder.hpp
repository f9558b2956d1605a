// host_coder.hpp — host side of the interval coder: CRangeDecoder (src/colord/sub_rc.h:216-392) and the adaptive frequency
// models behind every context (rc.h:34-220 CSimpleModel, :225-480 CSimpleModelFixedSize, :487-764 CFenwickTreeModelFixedSize —
// three containers of the same counts: start at 1, +ADDER per coded symbol, halve-round-up at MAX_TOTAL; basic_coder.h:116-137:
// a context seen for the first time gets a copy of the all-ones template).
//
// Decoding is one dependent chain per model domain (the symbol decides the next context): it runs on the host, one thread per
// stream, exactly where the reference runs it (decompression_common.cpp:318-337).  Used by decode.hip only.
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
	uint64_t cum_freq(uint64_t tot) { return buffer / (range /= tot); }   // GetCumFreq (:264-268)
	void update(uint64_t freq, uint64_t cum)                            // UpdateFrequency (:270-287)
	{
		const uint64_t r = cum * range;
		buffer -= r; low += r; range *= freq;
		while (range <= TOP)
		{
			if ((low ^ (low + range)) & MASK) { const uint64_t q = low; range = (q | TOP) - q; }
			buffer = (buffer << 8) + byte();
			low <<= 8; range <<= 8;
		}
	}
};

// one model family: context value -> counters (n_sym) + total, in one pool; open addressing on the context
struct Family {
	uint32_t n_sym = 0, max_total = 0, adder = 0;
	std::vector<uint64_t> keys; std::vector<uint32_t> vals; uint64_t mask = 0, used = 0;
	std::vector<uint32_t> pool;                                          // model i at pool[i * (n_sym + 1)], total last
	void init(uint32_t n, uint32_t mt, uint32_t ad) { n_sym = n; max_total = mt; adder = ad; reset(); }
	void reset() { keys.assign(1024, ~0ULL); vals.assign(1024, 0); mask = 1023; used = 0; pool.clear(); }
	static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
	uint32_t* model(uint64_t ctx)
	{
		uint64_t h = mix(ctx) & mask;
		while (keys[h] != ~0ULL && keys[h] != ctx) h = (h + 1) & mask;
		if (keys[h] == ~0ULL)
		{
			const uint32_t idx = (uint32_t)(pool.size() / (n_sym + 1));
			pool.resize(pool.size() + n_sym + 1, 1u);
			pool.back() = n_sym;
			keys[h] = ctx; vals[h] = idx;
			if (++used * 2 > mask + 1) grow();
			return pool.data() + (uint64_t)idx * (n_sym + 1);
		}
		return pool.data() + (uint64_t)vals[h] * (n_sym + 1);
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
	// Decode / DecodeExcluding (rc.h:850-878, 926-1043): cumulative and total skip the excluded symbols
	uint32_t decode(RangeDec& rc, uint64_t ctx, int exc1 = -1, int exc2 = -1)
	{
		uint32_t* m = model(ctx);
		uint32_t tot = m[n_sym];
		if (exc1 >= 0) tot -= m[exc1];
		if (exc2 >= 0) tot -= m[exc2];
		const uint64_t target = rc.cum_freq(tot);
		uint64_t t = 0; uint32_t sym = n_sym - 1, cum = 0;
		for (uint32_t i = 0; i < n_sym; ++i)
		{
			if ((int)i == exc1 || (int)i == exc2) continue;
			t += m[i];
			if (t > target) { sym = i; cum = (uint32_t)(t - m[i]); break; }
		}
		rc.update(m[sym], cum);
		m[sym] += adder; m[n_sym] += adder;
		while (m[n_sym] >= max_total) { uint32_t s = 0; for (uint32_t i = 0; i < n_sym; ++i) { m[i] = (m[i] + 1) / 2; s += m[i]; } m[n_sym] = s; }
		return sym;
	}
};

} // namespace hostrc
