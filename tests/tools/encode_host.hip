// Debugging aid, not part of the library and not used by any test verdict: compiles the encoder logic of
// colord_amd/csrc/encode_core.hpp for the HOST (-DCL_HOST_DEBUG turns its device functions into __host__ __device__)
// and replays the level-synchronous driver of encode_es.hip sequentially, so that a divergence from the reference can
// be bisected without a GPU.
//   hipcc -DCL_HOST_DEBUG --offload-arch=gfx950 -O1 -g -std=c++17 -shared -fPIC -I include -I colord_amd/csrc \
//         tests/tools/encode_host.hip -o /tmp/libenc_host.so
#include "../../colord_amd/csrc/encode_core.hpp"
#include <vector>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <numeric>
using namespace enc;

struct HostMem {
	uint8_t qb[MID_ROWS], tb[MID_COLS]; char esb[MID_ROWS + MID_COLS]; std::vector<uint64_t> hist; uint32_t nb;
	uint64_t peq_[4][256], pv_[256], mv_[256];
	void lap(uint32_t) {}
	uint64_t peq(uint32_t s, uint32_t b) const { return peq_[s][b]; } void peq_set(uint32_t s, uint32_t b, uint64_t v) { peq_[s][b] = v; }
	uint64_t pv(uint32_t b) const { return pv_[b]; } uint64_t mv(uint32_t b) const { return mv_[b]; } void pv_set(uint32_t b, uint64_t v) { pv_[b] = v; } void mv_set(uint32_t b, uint64_t v) { mv_[b] = v; }
	uint32_t q(uint32_t i) const { return qb[i]; } uint32_t t(uint32_t j) const { return tb[j]; }
	void q_set(uint32_t i, uint32_t v) { qb[i] = (uint8_t)v; } void t_set(uint32_t j, uint32_t v) { tb[j] = (uint8_t)v; }
	char es_get(uint32_t k) const { return esb[k]; } void es_set(uint32_t k, char c) { esb[k] = c; }
	void hist_put(uint32_t j, uint32_t b, uint64_t P, uint64_t Ph) { hist[(j * nb + b) * 2] = P; hist[(j * nb + b) * 2 + 1] = Ph; }
	void hist_get(uint32_t j, uint32_t b, uint64_t& P, uint64_t& Ph) const { P = hist[(j * nb + b) * 2]; Ph = hist[(j * nb + b) * 2 + 1]; }
};
struct Lvl { std::vector<FrameRec> frames; std::vector<CandEnt> cands; std::vector<GapRec> gaps; std::vector<char> es; std::vector<PendRec> pend; std::vector<uint8_t> dec;
	LevelV view() { return LevelV{ frames.data(), cands.data(), gaps.data(), es.data(), pend.data(), dec.data(), (uint32_t)frames.size(), (uint32_t)gaps.size() }; } };

extern "C" int dbg_encode(const uint64_t* r_packed, const uint64_t* r_woff, const uint32_t* r_lens, const uint32_t* r_inv, const uint8_t* has_n, uint32_t n_reads,
                          const uint64_t* f_packed, const uint64_t* f_woff, const uint32_t* f_lens,
                          const uint32_t* n_cands, const uint32_t* cand, const uint64_t* cand_off, const uint32_t* data,
                          uint32_t c, uint32_t m, uint32_t min_part_alt, uint32_t max_rec, double cost_mult,
                          const uint32_t* pack_bounds, uint32_t n_packs, uint64_t pool_bytes, uint32_t force_large,
                          uint8_t* out, uint64_t cap, uint64_t* off, uint32_t* nt, uint32_t* stats)
{
	ArenaV A{ r_packed, r_woff, r_lens }, R{ f_packed, f_woff, f_lens };
	EncCfg cfg{ c, m, min_part_alt, max_rec, cost_mult };
	std::vector<uint8_t> mem(pool_bytes);
	std::vector<Lvl> levels(1);
	std::vector<uint32_t> frame_of_read(n_reads, 0xffffffffu);
	for (uint32_t r = 0; r < n_reads; ++r)
	{
		const uint32_t nc = has_n[r] ? 0u : n_cands[r];
		if (!nc) continue;
		frame_of_read[r] = (uint32_t)levels[0].frames.size();
		FrameRec F; F.read = r; F.level = 0; F.enc_off = 0; F.enc_len = r_lens[r]; F.n_cands = nc; F.first_gap = 0; F.n_gaps = 0; F.pad = 0; F.cand_base = levels[0].cands.size();
		levels[0].frames.push_back(F);
		for (uint32_t j = 0; j < nc; ++j) levels[0].cands.push_back(cand_level0(cand + ((uint64_t)r * c + j) * 4, cand_off[(uint64_t)r * c + j], data));
	}
	int bad = 0;
	for (uint32_t lv = 0; lv < levels.size(); ++lv)
	{
		Lvl& L = levels[lv];
		if (L.frames.empty()) break;
		uint32_t ng = 0;
		for (auto& F : L.frames) { F.first_gap = ng; F.n_gaps = L.cands[F.cand_base + F.level].n + 1; ng += F.n_gaps; }
		L.gaps.resize(ng);
		LevelV V = L.view();
		uint64_t es_total = 0; uint32_t n_pend = 0;
		std::vector<uint32_t> pend_idx(ng);
		for (uint32_t f = 0; f < L.frames.size(); ++f)
			for (uint32_t g = 0; g < L.frames[f].n_gaps; ++g)
			{
				const uint32_t gi = L.frames[f].first_gap + g;
				if (!gap_init(V, f, g, data, R, L.gaps[gi])) ++bad;
				L.gaps[gi].es_off = es_total; es_total += gap_es_capacity(L.gaps[gi]);
				pend_idx[gi] = n_pend; n_pend += L.gaps[gi].ne < min_part_alt;
			}
		L.es.assign(es_total + 16, 0); L.pend.resize(n_pend + 1); L.dec.assign(n_pend + 1, 0);
		V = L.view();
		for (uint32_t gi = 0; gi < ng; ++gi)
		{
			GapRec& g = L.gaps[gi];
			uint32_t rows, cols; uint32_t cls = gap_class(g, rows, cols);
			if (cls == 0) continue;
			if (force_large == 1) cls = 6;
			if (force_large == 2 && cls <= 4) cls = 5;                           // everything alignable through the mid path
			if (cls == 5 && (rows > MID_ROWS || cols > MID_COLS)) cls = 6;
			stats[cls]++;
			if (cls == 5)
			{
				static HostMem hm;
				uint32_t n, mm, d_before;
				stage_small(hm, g, A, R, n, mm);
				hm.nb = (n + 63) / 64; hm.hist.assign(2ull * hm.nb * mm + 2, 0);
				const uint32_t k = hm.nb <= 8 ? align_mid<8>(hm, n, mm, g.kind, g.left != 0, g.nr, g.use, &d_before)
				                 : hm.nb <= 16 ? align_mid<16>(hm, n, mm, g.kind, g.left != 0, g.nr, g.use, &d_before)
				                 : align_mid<0>(hm, n, mm, g.kind, g.left != 0, g.nr, g.use, &d_before);
				memcpy(L.es.data() + g.es_off, hm.esb, k);
				g.es_len = k; g.d_before = d_before;
			}
			else if (cls <= 4)
			{
				static HostMem hm; hm.nb = cls; hm.hist.assign(256 * cls * 2, 0);
				uint32_t n, mm, d_before, k = 0;
				stage_small(hm, g, A, R, n, mm);
				switch (cls)
				{
				case 1: k = align_small<1>(hm, n, mm, g.kind, g.left != 0, g.nr, g.use, &d_before); break;
				case 2: k = align_small<2>(hm, n, mm, g.kind, g.left != 0, g.nr, g.use, &d_before); break;
				case 3: k = align_small<3>(hm, n, mm, g.kind, g.left != 0, g.nr, g.use, &d_before); break;
				default: k = align_small<4>(hm, n, mm, g.kind, g.left != 0, g.nr, g.use, &d_before); break;
				}
				memcpy(L.es.data() + g.es_off, hm.esb, k);
				g.es_len = k; g.d_before = d_before;
			}
			else
			{
				LanePool pool{ mem.data(), pool_bytes, 0, false, 0 };
				if (!align_large_gap(pool, g, A, R, L.es.data() + g.es_off)) { ++bad; fprintf(stderr, "pool too small for gap %u (why %u)\n", gi, pool.why); }
			}
		}
		std::vector<uint32_t> rejected;
		for (uint32_t gi = 0; gi < ng; ++gi) if (gap_finish(V, gi, A, cfg, pend_idx[gi])) rejected.push_back(gi);
		Lvl N;
		for (uint32_t gi : rejected)
		{
			const GapRec g = L.gaps[gi]; const FrameRec F = L.frames[g.frame];
			CandEnt o[16];
			if (!spawn_cands(F, L.cands.data() + F.cand_base, data, g, cfg, o)) { L.gaps[gi].state = GS_LITERAL; continue; }
			const uint32_t child = (uint32_t)N.frames.size(); const uint64_t cb = N.cands.size();
			N.frames.resize(child + 1); N.cands.resize(cb + F.n_cands);
			spawn_child(V, gi, data, cfg, child, cb, N.frames.data(), N.cands.data());
		}
		stats[8 + lv] = (uint32_t)L.frames.size();
		if (N.frames.empty() || lv + 1 >= 10) break;
		levels.push_back(std::move(N));
	}
	TreeV T; memset(&T, 0, sizeof(T));
	for (size_t i = 0; i < levels.size(); ++i) T.lv[i] = levels[i].view();
	T.frame_of_read = frame_of_read.data();
	std::vector<uint32_t> bc(4ull * n_reads, 0);
	for (uint32_t r = 0; r < n_reads; ++r) if (!has_n[r]) for (uint32_t i = 0; i < r_lens[r]; ++i) ++bc[4 * r + arena_base_at(A, r_woff[r], i)];
	for (uint32_t p = 0; p < n_packs; ++p) est_pack(T, pack_bounds[p], pack_bounds[p + 1], r_lens, has_n, bc.data());
	std::vector<uint32_t> sizes(n_reads);
	for (uint32_t r = 0; r < n_reads; ++r) emit_read<false>(A, r_inv, has_n, T, r, data, sizes.data(), nt, nullptr, nullptr);
	off[0] = 0; for (uint32_t r = 0; r < n_reads; ++r) off[r + 1] = off[r] + sizes[r];
	if (off[n_reads] > cap) return -1;
	for (uint32_t r = 0; r < n_reads; ++r) emit_read<true>(A, r_inv, has_n, T, r, data, nullptr, nullptr, off, out);
	return bad;
}
