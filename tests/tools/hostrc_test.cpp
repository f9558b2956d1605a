// hostrc_test — the host decoder's interval arithmetic and models (colord_amd/csrc/host_coder.hpp) against the ORACLE's (oracle/rc.h,
// oracle/rc.c: the restatement of sub_rc.h / rc.h that the CPU suite pins to the reference's streams):
//   1. RangeDec::update — the loop-free renormalisation (k steps at once when low and low + range agree in their top k bytes) must leave
//      exactly the state of orc_rcd_update's byte-by-byte loop, on random states incl. ranges of every bit length and straddling intervals;
//   2. whole streams: symbols of alphabets 2, 4, 5, 8 (unrolled search), 24, 96, 256 (block sums), with and without exclusions, over dense
//      and hashed contexts, encoded by the oracle's encoder and decoded by BOTH decoders: same symbols, same final decoder state.
// Test infrastructure: g++ -O2 -std=c++17 tests/tools/hostrc_test.cpp oracle/rc.c -o hostrc_test && ./hostrc_test
#include "../../colord_amd/csrc/host_coder.hpp"
extern "C" {
#include "../../oracle/rc.h"
}
#include <cstdio>
#include <random>

int main()
{
	std::mt19937_64 rng(7);
	// ---- 1. update() on random states -----------------------------------------------------------------------------------------------
	std::vector<uint8_t> in(1 << 16); for (auto& b : in) b = (uint8_t)rng();
	uint64_t bad = 0;
	for (int it = 0; it < 4000000; ++it)
	{
		const uint64_t low = rng(); uint64_t range = rng() >> (rng() % 64); if (!range) range = 1;
		if (low + range < low) continue;                                   // (low + range never wraps in the coder)
		const uint64_t buf = rng(); const uint64_t pos = rng() % (in.size() + 4);
		hostrc::RangeDec a; a.low = low; a.range = range; a.buffer = buf; a.in = in.data(); a.n = in.size(); a.pos = pos < in.size() ? pos : in.size();
		orc_rcd o; o.low = low; o.range = range; o.buffer = buf; o.in = in.data(); o.n = in.size(); o.pos = a.pos;
		a.update(1, 0); orc_rcd_update(&o, 1, 0);
		if (a.low != o.low || a.range != o.range || a.buffer != o.buffer || a.pos != o.pos) { if (bad++ < 5) printf("update differs: low %016llx range %016llx\n", (unsigned long long)low, (unsigned long long)range); }
	}
	if (bad) { printf("FAILED: %llu states\n", (unsigned long long)bad); return 1; }
	// ---- 2. whole streams ---------------------------------------------------------------------------------------------------------------
	struct Cfg { uint32_t n_sym, max_total, adder, dense_bits; bool excl; };
	const Cfg cfgs[] = { { 2, 1u << 15, 1, 4, false }, { 4, 1u << 10, 1, 12, true }, { 5, 1u << 18, 8, 10, false }, { 8, 1u << 15, 1, 9, true },
	                     { 24, 1u << 15, 1, 8, false }, { 96, 1u << 20, 32, 0, false }, { 256, 1u << 13, 1, 11, false } };
	for (const Cfg& c : cfgs)
	{
		const uint32_t N = 300000;
		std::vector<uint64_t> ctx(N); std::vector<uint32_t> sym(N); std::vector<int> e1(N, -1), e2(N, -1);
		for (uint32_t i = 0; i < N; ++i)
		{
			ctx[i] = (rng() % 7 == 0) ? (1ull << 40) + rng() % 50 : rng() % 3000;        // dense range, beyond it, and far outside (hashed)
			if (c.excl && rng() % 3 == 0) { e1[i] = (int)(rng() % c.n_sym); if (c.n_sym > 4 && rng() % 2) { do e2[i] = (int)(rng() % c.n_sym); while (e2[i] == e1[i]); } }
			do sym[i] = (uint32_t)((rng() % 4) ? rng() % c.n_sym : (rng() % 3) % c.n_sym); while ((int)sym[i] == e1[i] || (int)sym[i] == e2[i]);   // skewed
		}
		orc_bytes ob{ nullptr, 0, 0 }; orc_rce enc; enc.out = &ob; orc_rce_start(&enc);
		orc_ctxmap em; orc_ctxmap_init(&em, c.n_sym, c.max_total, c.adder);
		for (uint32_t i = 0; i < N; ++i) orc_encode_sym(&enc, &em, ctx[i], sym[i], e1[i], e2[i]);
		orc_rce_end(&enc); orc_ctxmap_free(&em);
		hostrc::RangeDec rd; rd.start(ob.p, ob.n);
		hostrc::Family fam; fam.init(c.n_sym, c.max_total, c.adder, c.dense_bits);
		orc_rcd od; od.in = ob.p; od.n = ob.n; od.pos = 0; orc_rcd_start(&od);
		orc_ctxmap dm; orc_ctxmap_init(&dm, c.n_sym, c.max_total, c.adder);
		static const uint8_t EXCL_NONE = 0;
		(void)EXCL_NONE;
		for (uint32_t i = 0; i < N; ++i)
		{
			uint32_t got;
			if (c.n_sym == 8 && c.excl) { uint32_t m = 0; if (e1[i] >= 0) m |= 1u << e1[i]; if (e2[i] >= 0) m |= 1u << e2[i]; got = fam.decode_masked8(rd, ctx[i], m); }
			else got = fam.decode(rd, ctx[i], e1[i], e2[i]);
			const uint32_t want = orc_decode_sym(&od, &dm, ctx[i], e1[i], e2[i]);
			if (got != sym[i] || want != sym[i] || rd.low != od.low || rd.range != od.range || rd.buffer != od.buffer || rd.pos != od.pos)
			{ printf("FAILED: alphabet %u, symbol %u: host %u oracle %u coded %u\n", c.n_sym, i, got, want, sym[i]); return 1; }
		}
		orc_ctxmap_free(&dm); free(ob.p);
	}
	printf("ok: update() on 4e6 states and 7 alphabets x 300000 symbols equal the oracle's decoder\n");
	return 0;
}
