// Debugging aid, not part of the library and not used by any test verdict: compiles the per-read encoder logic of
// colord_amd/csrc/encode_es.hip for the HOST (-DCL_HOST_DEBUG turns its device functions into __host__ __device__)
// so that a divergence from the oracle can be bisected without a GPU.
//   hipcc -DCL_HOST_DEBUG --offload-arch=gfx950 -O1 -g -std=c++17 -shared -fPIC -I include -I colord_amd/csrc \
//         tests/tools/encode_host.hip -o /tmp/libenc_host.so -L colord_amd -lcolord_hip -Wl,-rpath,$PWD/colord_amd
#include "../../colord_amd/csrc/encode_es.hip"
#include <vector>
#include <cstdlib>

extern "C" int dbg_encode(const uint64_t* r_packed, const uint64_t* r_woff, const uint32_t* r_lens, const uint32_t* r_inv, const uint8_t* has_n, uint32_t n_reads,
                          const uint64_t* f_packed, const uint64_t* f_woff, const uint32_t* f_lens,
                          const uint32_t* n_cands, const uint32_t* cand, const uint64_t* cand_off, const uint32_t* data,
                          uint32_t c, uint32_t m, uint32_t min_part_alt, uint32_t max_rec, double cost_mult,
                          const uint32_t* pack_bounds, uint32_t n_packs, uint64_t pool_bytes, uint32_t scale,
                          uint8_t* out, uint64_t cap, uint64_t* off, uint32_t* nt, uint32_t* why)
{
	ArenaV A{ r_packed, r_woff, r_lens }, R{ f_packed, f_woff, f_lens };
	AnchorsV AV{ n_cands, cand, cand_off, data };
	EncCfg cfg{ c, m, min_part_alt, max_rec, cost_mult, scale, 0 };
	std::vector<uint8_t> mem(pool_bytes);
	std::vector<ReadOut> rout(n_reads);
	std::vector<uint64_t> items; std::vector<GapRec> gaps; std::vector<PendRec> pend; std::vector<char> es;
	int failed = 0;
	for (uint32_t r = 0; r < n_reads; ++r)
	{
		LanePool pool{ mem.data(), pool_bytes, 0, false, 0 };
		ReadOut ro; Sink sk;
		const bool have = expand_read(pool, A, R, AV, cfg, has_n, r, ro, sk);
		why[r] = 0;
		if (!have && (pool.overflow || sk.overflow)) { why[r] = pool.why | sk.why | 0x100; ro.plain = 1; ++failed; }
		else if (have)
		{
			ro.n_items = sk.n_items; ro.n_gaps = sk.n_gaps; ro.n_pend = sk.n_pend; ro.es_len = sk.n_es;
			ro.item_off = items.size(); ro.gap_off = gaps.size(); ro.pend_off = pend.size(); ro.es_off = es.size();
			items.insert(items.end(), sk.items, sk.items + sk.n_items); gaps.insert(gaps.end(), sk.gaps, sk.gaps + sk.n_gaps);
			pend.insert(pend.end(), sk.pend, sk.pend + sk.n_pend); es.insert(es.end(), sk.es, sk.es + sk.n_es);
		}
		rout[r] = ro;
	}
	std::vector<uint8_t> dec(pend.size() + 1);
	for (uint32_t p = 0; p < n_packs; ++p) estimate_pack(rout.data(), pend.data(), pack_bounds[p], pack_bounds[p + 1], r_lens, dec.data());
	std::vector<uint32_t> sizes(n_reads);
	items.push_back(0); gaps.push_back(GapRec{}); es.push_back(0);
	for (uint32_t r = 0; r < n_reads; ++r) emit_read<false>(A, r_inv, r, rout.data(), items.data(), gaps.data(), es.data(), dec.data(), sizes.data(), nt, nullptr, nullptr);
	off[0] = 0; for (uint32_t r = 0; r < n_reads; ++r) off[r + 1] = off[r] + sizes[r];
	if (off[n_reads] > cap) return -1;
	for (uint32_t r = 0; r < n_reads; ++r) emit_read<true>(A, r_inv, r, rout.data(), items.data(), gaps.data(), es.data(), dec.data(), nullptr, nullptr, off, out);
	return failed;
}
