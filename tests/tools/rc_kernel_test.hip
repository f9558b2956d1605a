// Differential test of k_range_code (csrc/rc_dev.hpp) against the ORACLE's interval coder (oracle/rc.h: orc_rce_start / _encode / _end, the
// restatement of sub_rc.h:72-100,203-210 that the CPU suite pins to the reference's golden streams) on random triples: sizes and bytes of
// every part.  Test infrastructure: the oracle is the checker here, nothing of it is linked into the library.  Debugging aid:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude tests/tools/rc_kernel_test.hip -o /tmp/rc_test && /tmp/rc_test
#include "../../colord_amd/csrc/rc_dev.hpp"
extern "C" {
#include "../../oracle/rc.h"
}
#include <cstdio>
#include <random>
#include <vector>

static std::vector<uint8_t> host_code(const std::vector<triple_t>& sy, bool dbg = false)
{
	orc_bytes ob{ nullptr, 0, 0 }; orc_rce e; e.out = &ob;
	orc_rce_start(&e);
	for (const triple_t& t : sy)
	{
		const uint32_t tot = (uint32_t)(t & 0x1fffff), freq = (uint32_t)((t >> 21) & 0x1fffff), cum = (uint32_t)(t >> 42);
		const size_t before = ob.n;
		orc_rce_encode(&e, freq, cum, tot);
		if (dbg && (&t - sy.data()) < 40) printf("oracle %zu: tot %u freq %u cum %u nb %zu n_out %zu low %016llx range %016llx\n", (size_t)(&t - sy.data()), tot, freq, cum, ob.n - before, before, (unsigned long long)e.low, (unsigned long long)e.range);
	}
	orc_rce_end(&e);
	std::vector<uint8_t> out(ob.p, ob.p + ob.n);
	free(ob.p);
	return out;
}

int main()
{
	const uint32_t np = 150;                                  // three groups, the last one ragged
	std::mt19937_64 rng(5);
	std::vector<std::vector<triple_t>> parts(np);
	std::vector<uint32_t> plen(np);
	for (uint32_t p = 0; p < np; ++p)
	{
		const uint32_t n = p == 7 ? 0 : (uint32_t)(rng() % 5000) + (p % 3 == 0 ? 20000 : 1);
		for (uint32_t i = 0; i < n; ++i)
		{
			// totals over the whole range, the models' usual ones, and the extremes (1, 2, powers of two, 2^21 - 1) in parts of their own
			uint32_t tot = (rng() & 1) ? (uint32_t)(rng() % ((1u << 21) - 1)) + 1 : (uint32_t)(rng() % 60000) + 1000;
			if (p % 10 == 4) { const uint32_t ex[6] = { 1, 2, 3, 1u << (1 + rng() % 20), (1u << 21) - 1, (1u << 21) - 2 }; tot = ex[rng() % 6]; }
			const uint32_t freq = (rng() % 4 == 0) ? 1 : (uint32_t)(rng() % tot) + 1, cum = (uint32_t)(rng() % (tot - freq + 1));
			parts[p].push_back(((uint64_t)cum << 42) | ((uint64_t)freq << 21) | tot);
		}
		plen[p] = n;
	}
	const uint32_t ng = (np + 63) / 64;
	const uint64_t BIG = (1ull << 31) + 4096;                    // the parts' room begins beyond 2 GB: offsets whose low half has its top bit set
	std::vector<uint64_t> gbase(ng), out_off(np + 1, BIG);
	uint64_t total = 0;
	for (uint32_t g = 0; g < ng; ++g) { gbase[g] = total; uint32_t m = 0; for (uint32_t p = g * 64; p < np && p < g * 64 + 64; ++p) m = std::max(m, plen[p]); total += trip_group_words(m); }
	std::vector<triple_t> trip(total + 64, 0xdeadbeefdeadbeefULL);
	for (uint32_t p = 0; p < np; ++p) for (uint32_t i = 0; i < plen[p]; ++i) trip[trip_slot(gbase[p >> 6], p & 63, i)] = parts[p][i];
	for (uint32_t p = 0; p < np; ++p) out_off[p + 1] = out_off[p] + ((uint64_t)plen[p] * 8 + 64 + 7) / 8 * 8;
	triple_t* d_trip; uint64_t *d_gbase, *d_off, *d_size, *d_inv; uint32_t* d_plen; uint8_t* d_out;
	hipMalloc((void**)&d_trip, trip.size() * 8); hipMalloc((void**)&d_gbase, ng * 8); hipMalloc((void**)&d_off, (np + 1) * 8); hipMalloc((void**)&d_size, np * 8);
	hipMalloc((void**)&d_inv, (uint64_t)INV_TABLE_SIZE * 8); hipMalloc((void**)&d_plen, np * 4); hipMalloc((void**)&d_out, out_off[np]);
	hipMemcpy(d_trip, trip.data(), trip.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_gbase, gbase.data(), ng * 8, hipMemcpyHostToDevice);
	hipMemcpy(d_off, out_off.data(), (np + 1) * 8, hipMemcpyHostToDevice); hipMemcpy(d_plen, plen.data(), np * 4, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k_fill_inv_table, dim3(INV_TABLE_SIZE / 256), dim3(256), 0, 0, d_inv);
	int bad = 0;
	for (int form = 0; form < 1; ++form)
	{
		hipMemset(d_out + BIG, 0xAA, out_off[np] - BIG); hipMemset(d_size, 0, np * 8);
		hipLaunchKernelGGL(k_range_code, dim3(ng), dim3(64), 0, 0, (const triple_t*)d_trip, (const uint64_t*)d_gbase, (const uint32_t*)d_plen, np, d_out, (const uint64_t*)d_off, d_size, (const uint64_t*)d_inv);
		if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed (form %d)\n", form); return 2; }
		std::vector<uint64_t> size(np); std::vector<uint8_t> out_v(out_off[np] - BIG);
		hipMemcpy(size.data(), d_size, np * 8, hipMemcpyDeviceToHost); hipMemcpy(out_v.data(), d_out + BIG, out_v.size(), hipMemcpyDeviceToHost);
		const uint8_t* out = out_v.data() - BIG;
		if (form == 0 && getenv("RC_DEBUG")) host_code(parts[1], true);
		for (uint32_t p = 0; p < np; ++p)
		{
			const std::vector<uint8_t> e = host_code(parts[p]);
			if (size[p] != e.size()) { if (bad++ < 10) printf("form %d part %u (%u symbols): size %llu, expected %zu\n", form, p, plen[p], (unsigned long long)size[p], e.size()); continue; }
			for (size_t i = 0; i < e.size(); ++i) if (out[out_off[p] + i] != e[i]) { if (bad++ < 10) printf("form %d part %u: byte %zu of %zu differs (%02x, expected %02x)\n", form, p, i, e.size(), out[out_off[p] + i], e[i]); break; }
			for (uint64_t i = out_off[p] + e.size(); i < out_off[p + 1]; ++i) if (out[i] != 0xAA) { if (bad++ < 10) printf("form %d part %u: wrote beyond its size (offset %llu of size %zu)\n", form, p, (unsigned long long)(i - out_off[p]), e.size()); break; }
		}
	}
	printf(bad ? "FAILED: %d parts\n" : "ok: %u parts equal the oracle's coder (oracle/rc.h), nothing written beyond a part's size\n", bad ? bad : np);
	return bad ? 1 : 0;
}
