// anchors.hip — m-mer anchors between a read and its candidate reference reads (a8):
// CMmers / AnalyseRefRead / get_aligned_mmers_LIS / MergeAnchors / MmerBasedAnchors / prepareEncodeCandidates /
// fixOverlaping* (src/colord/encoder.cpp:291-493,617-776,1016-1111,1149-1192,1577-1622).
//
// Only SETS matter up to the LIS, so the per-read hash map with its duplicate side vectors
// and the Bloom filter are replaced by:
//   1. one open-addressing table per read in HBM: m-mer -> chain of its positions (built by all positions in
//      parallel with CAS / exchange);
//   2. one task per (read, candidate, orientation): every reference position looks its m-mer up and emits the
//      (enc position, ref position) pairs — their number is exactly the reference's match count used for the
//      "too many matches" veto;
//   3. a device-wide radix sort of the pairs by (task, enc position asc, ref position desc) — the order in
//      which the reference feeds its LIS;
//   4. one lane per task replays the reference's patience LIS (same tie-breaking), its map-back scan and the
//      merge of consecutive hits into anchors;
//   5. per read: orientation choice (reverse complement wins ties), stable sort by total anchor length,
//      overlap trimming in reference then in read coordinates.
#include "common.hpp"
#include "objects.hpp"
#include <algorithm>

struct cl_anchors {
	cl_ctx* ctx = nullptr;
	uint32_t n_reads = 0, c = 0;
	uint64_t n_anchors = 0;
	DevBuf<uint32_t> n_cands;     // n_reads
	DevBuf<uint32_t> cand;        // n_reads * c * 4: ref_id, rev, tot_anchor_len, n_anchors
	DevBuf<uint64_t> cand_off;    // n_reads * c + 1: first anchor of each candidate slot
	DevBuf<uint32_t> anchors;     // n_anchors * 3: len, pos_enc, pos_ref
};

namespace {
// Match pairs are 64-bit sort keys  task | position in the read (pe bits) | ~position in the candidate (pr bits).  pe / pr are
// chosen per call / per batch from the longest candidate / read (TaskCfg), so ultra-long ONT reads only cost sort key bits where
// they occur, and the usual reads (< 2^18 bases) sort one radix pass less than a fixed 2 x 20 bits would.

struct Arena { const uint64_t* packed; const uint64_t* word_off; const uint32_t* lens; };

// m-mer (m <= 28) starting at base p of read r
__device__ inline uint64_t mmer_at(const Arena& A, uint64_t wb, uint32_t p, uint32_t m)
{
	const uint64_t* w = A.packed + wb + (p >> 5);
	const uint32_t j = p & 31;                     // first base inside word
	const uint64_t hi = w[0], lo = w[1];
	const uint32_t s = 128 - 2 * (j + m);
	const uint64_t v = (s >= 64) ? (hi >> (s - 64)) : ((hi << (64 - s)) | (lo >> s));
	return v & ((1ULL << (2 * m)) - 1);
}
// the same from the two words that hold it (loaded ahead of their use)
__device__ inline uint64_t mmer_of(uint64_t hi, uint64_t lo, uint32_t p, uint32_t m)
{
	const uint32_t s = 128 - 2 * ((p & 31) + m);
	const uint64_t v = (s >= 64) ? (hi >> (s - 64)) : ((hi << (64 - s)) | (lo >> s));
	return v & ((1ULL << (2 * m)) - 1);
}
__device__ inline uint64_t revcomp_m(uint64_t x, uint32_t m)
{
	x = ~x; x = __brevll(x);
	x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
	return x >> (64 - 2 * m);
}
// m-mer at position q of the oriented reference read (rev: reverse complement of the stored read)
__device__ inline uint64_t ref_mmer(const Arena& R, uint32_t id, bool rev, uint32_t q, uint32_t m)
{
	const uint32_t len = R.lens[id];
	const uint64_t wb = R.word_off[id];
	return rev ? revcomp_m(mmer_at(R, wb, len - m - q, m), m) : mmer_at(R, wb, q, m);
}

// Per read: open-addressing table of its m-mers under their CANONICAL form (the smaller of the m-mer and its reverse
// complement) with two position chains per key: head[0] the positions where the read has the canonical form
// itself, head[1] those where it has the other one.  One probe with a reference m-mer then serves both
// orientations of the reference (the reference's two anchor analyses, encoder.cpp:1046-1066, probe the same hash of the
// read with the m-mers of the reference and of its reverse complement).
// While a region of the table is built (in LDS) a slot holds the full key, ONE chain head and which strands the key has been seen
// on (the count of distinct m-mers per strand is exact).  What goes to HBM is 8 bytes per slot: a 32-bit tag of the key and the head.
// A chain link — head or `next` entry — is a position (30 bits) with the strand of THAT position in bit 31 (0: the read has the
// canonical form there, 1: the other one), so a walk knows both orientations' hits without reading the read again.  A tag can
// belong to another key (2^-32 per occupied slot compared, a handful per 10^10 probes): the look-up confirms a tag match on the
// read's own m-mer at the head's position before it believes it, and goes on probing otherwise.  (Rounds 1-3a: 16-byte slots with
// the full key and one head per strand: 15 GB written per table build of a 0.27-Gbase batch; now half.)
struct BuildSlot { uint64_t key; uint32_t head; uint32_t strands; };
typedef uint2 EncSlot;                                                     // x: tag, y: head link (NIL: empty)
constexpr uint32_t LINK_POS = 0x3fffffffu;
__device__ inline uint32_t tag_of(uint64_t hash) { return (uint32_t)((hash * 0xD6E8FEB86659FD93ULL) >> 32); }
// slot of a hash in a table of tsz slots: tsz is a power of two below one region, else a MULTIPLE of the region size (2 n slots
// rounded up: powers of two meant 2.5 n .. 5 n slots, 40 - 80 bytes per base written out by every table build), so the index
// is the high half of hash x tsz rather than a mask
__device__ inline uint32_t table_slot(uint64_t hash, uint32_t tsz) { return (uint32_t)(((uint64_t)(uint32_t)(hash >> 17) * tsz) >> 32); }
struct EncTable { EncSlot* slots; const uint64_t* toff; uint32_t* next; const uint64_t* noff; };
constexpr uint32_t REGION_SLOTS = 2048;                                  // = REGION of the table build below
constexpr uint64_t KEY_EMPTY = ~0ULL;
constexpr uint32_t NIL = 0xffffffffu;

// ---- A1: table sizes; one wave per read inserts every position --------------------------------------------
__global__ void k_table_sizes(const uint32_t* __restrict__ lens, const uint8_t* __restrict__ has_n, const uint32_t* __restrict__ ncand,
                              uint32_t r0, uint32_t r1, uint32_t m, uint32_t x4, uint32_t lds_max_n, uint32_t* __restrict__ tsize, uint32_t* __restrict__ nsize, uint32_t* __restrict__ err)
{
	uint32_t r = r0 + blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= r1) return;
	uint32_t len = lens[r];
	bool active = ncand[r] > 0 && !has_n[r] && len >= m;
	uint32_t n = active ? len - m + 1 : 0;
	if (n <= lds_max_n) n = 0;                                             // its table lives in LDS for the length of one block (k_match_lds): nothing of it in HBM
	uint32_t t = 0;
	if (n) { t = 16; while (t < 2 * n + n / 2 && t < REGION_SLOTS) t <<= 1; if (t >= REGION_SLOTS) t = (uint32_t)(((uint64_t)n * x4 / 4 + REGION_SLOTS - 1) / REGION_SLOTS * REGION_SLOTS); }
	tsize[r - r0] = t; nsize[r - r0] = n;
}
// Table build, one block of 16 waves per read.  Inserting straight into the table in HBM costs a random 128-byte line
// read and written per m-mer (measured 100 bytes written per insertion).  Instead the table is built REGION by region
// (REGION slots = 32 KB) in LDS and written out once, coalesced: the read's positions are first binned by the region of
// their slot (counting sort through a scratch list), then every region is filled with LDS atomics.  Probing wraps
// inside the region, here and in table_heads.
constexpr uint32_t REGION = REGION_SLOTS, INS_T = 256;
__global__ __launch_bounds__(INS_T) void k_table_insert(Arena A, uint32_t r0, uint32_t r1, uint32_t m, EncTable T, uint32_t* __restrict__ n_distinct,
                                                       uint2* __restrict__ bins /* per position: (position, slot), grouped by region */, uint32_t* __restrict__ err)
{
	__shared__ BuildSlot reg[REGION];
	__shared__ uint32_t cnt[512], start[513];                             // positions per region (a read of 2^20 bases has 2^21 * 2 / 2048 = ... see below)
	const uint32_t r = r0 + blockIdx.x;
	if (r >= r1) return;
	const uint64_t t0 = T.toff[r - r0]; const uint32_t tsz = (uint32_t)(T.toff[r - r0 + 1] - t0);
	if (!tsz) return;
	const uint64_t n0 = T.noff[r - r0]; const uint32_t n = (uint32_t)(T.noff[r - r0 + 1] - n0);
	const uint64_t wb = A.word_off[r];
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t rs = tsz < REGION ? tsz : REGION, n_reg = tsz / rs, rshift = 31 - (uint32_t)__clz((int)rs);
	uint2* bin = bins + n0;
	uint32_t fresh = 0;
	for (uint32_t rg0 = 0; rg0 < n_reg; rg0 += 512)
	{	// (512 regions = 2^20 slots per round: one round for reads up to ~400 k bases)
		const uint32_t nr = n_reg - rg0 < 512 ? n_reg - rg0 : 512;
		for (uint32_t i = threadIdx.x; i < 512; i += INS_T) cnt[i] = 0;
		__syncthreads();
		for (uint32_t p = threadIdx.x; p < n; p += INS_T)
		{
			const uint64_t xf = mmer_at(A, wb, p, m), xr = revcomp_m(xf, m), x = xf < xr ? xf : xr;
			const uint32_t h = table_slot(hash_mm(x), tsz), rg = (h >> rshift) - rg0;
			if (rg < nr) atomicAdd(&cnt[rg], 1u);
		}
		__syncthreads();
		if (threadIdx.x == 0) { uint32_t a = 0; for (uint32_t i = 0; i < nr; ++i) { start[i] = a; a += cnt[i]; cnt[i] = 0; } start[nr] = a; }
		__syncthreads();
		for (uint32_t p = threadIdx.x; p < n; p += INS_T)
		{
			const uint64_t xf = mmer_at(A, wb, p, m), xr = revcomp_m(xf, m), x = xf < xr ? xf : xr;
			const uint32_t h = table_slot(hash_mm(x), tsz), rg = (h >> rshift) - rg0;
			if (rg < nr) bin[start[rg] + atomicAdd(&cnt[rg], 1u)] = make_uint2(p, h);
		}
		__syncthreads();
		__threadfence_block();
		for (uint32_t rg = 0; rg < nr; ++rg)
		{
			for (uint32_t i = threadIdx.x; i < rs; i += INS_T) { reg[i].key = KEY_EMPTY; reg[i].head = NIL; reg[i].strands = 0; }
			__syncthreads();
			const uint32_t b = start[rg], e = start[rg + 1];
			for (uint32_t i = b + threadIdx.x; i < e; i += INS_T)
			{
				const uint2 ph = bin[i];
				const uint32_t pp = ph.x;
				const uint64_t xf = mmer_at(A, wb, pp, m), xr = revcomp_m(xf, m), x = xf < xr ? xf : xr;
				uint32_t off = ph.y & (rs - 1), tries = 0;
				for (;;)
				{
					unsigned long long old = atomicCAS((unsigned long long*)&reg[off].key, (unsigned long long)KEY_EMPTY, (unsigned long long)x);
					if (old == KEY_EMPTY || old == x) break;
					off = (off + 1) & (rs - 1);
					if (++tries > rs) { atomicOr(err, 2u); break; }                // a full region: cannot happen below load 1
				}
				const uint32_t strand = xf != x ? 1u : 0u;
				T.next[n0 + pp] = atomicExch(&reg[off].head, pp | (strand << 31));   // the link to the element before (with ITS strand), or NIL
				if (!(atomicOr(&reg[off].strands, 1u << strand) & (1u << strand))) ++fresh;   // first position with this m-mer on this strand: distinct m-mers of the read
			}
			__syncthreads();
			EncSlot* dst = T.slots + t0 + (uint64_t)(rg0 + rg) * rs;
			for (uint32_t i = threadIdx.x; i < rs; i += INS_T) dst[i] = make_uint2(reg[i].head == NIL ? 0u : tag_of(hash_mm(reg[i].key)), reg[i].head);
			__syncthreads();
		}
	}
	fresh = wave_sum(fresh);
	if (lane == 0 && fresh) atomicAdd(&n_distinct[r - r0], fresh);       // (zeroed by the caller)
}
// ---- A2 / A3: the match pairs of BOTH orientations of every candidate in one pass over the reference ----
// task id t -> read r0 + t / (2c), slot (t / 2) % c, orientation t & 1 (0 = reverse complement, analysed first).
// Pairs go to one array in any order (they are sorted by (task, read position, ~reference position) next): a wave
// reserves room for the hits of its 64 probes with one atomic add.  The total is counted past the capacity too, so the
// caller can repeat the pass with enough room.
struct TaskCfg { uint32_t r0, r1, c, m; uint32_t pe, pr; double frac_always, frac_min, max_mult; };

// One BLOCK per read.  Nearly all probes miss (a candidate shares a stretch with the read, not its whole length), and a
// miss in the read's table in HBM costs a random 64-byte line and, worse, its latency: the 64 probes of a wave step wait
// for the slowest.  So the block first builds a Bloom filter of the read's m-mers in LDS — 32 KB (five blocks per CU),
// blocked: one 64-bit word per m-mer, three bits in it, all from the m-mer's one hash — and only probes that pass it
// go to the table: for a read of 15 k bases ~0.6 % of the misses, i.e. two of three wave steps touch no table at all.
constexpr uint32_t FILT_WORDS = 2048, STAGE = 128;
__device__ inline uint32_t filt_word(uint64_t hash) { return (uint32_t)(hash >> 46) & (FILT_WORDS - 1); }
__device__ inline uint64_t filt_mask(uint64_t hash) { return (1ull << ((hash >> 40) & 63)) | (1ull << ((hash >> 34) & 63)) | (1ull << ((hash >> 28) & 63)); }
// A LONG read is matched by several blocks, each for a SEGMENT of its positions (round 6): the filter of a block holds the m-mers of its
// segment only — 131 072 bits take the 3 bits a position of ~16 k positions (0.6-2.4 % of the misses pass) but are 3/4 full for a read of
// 60 k, where 42 % passed and went to the table in HBM: reads above 24 576 m-mers, 39 % of the bases, took 65 % of this kernel's time — and
// the hits it emits are those whose read position lies in its segment (the table and its chains are the read's: a walk skips the others).
// Every pair comes out exactly once; the reference is hashed once per segment (ALU the kernel has to spare).
__device__ __host__ inline uint32_t match_segments(uint32_t n, uint32_t seg_n) { return (seg_n == 0 || n <= seg_n + seg_n / 2) ? 1u : (n + seg_n - 1) / seg_n; }
__global__ __launch_bounds__(256) void k_match(Arena A, Arena R, EncTable T, TaskCfg cfg, const uint32_t* __restrict__ cand_refs, const uint32_t* __restrict__ cand_n,
                                              const uint32_t* __restrict__ n_distinct, const uint2* __restrict__ work /* (read of the batch, segment) */, uint32_t n_work, uint32_t seg_n,
                                              unsigned long long* __restrict__ n_pairs, uint64_t cap, uint64_t* __restrict__ pairs)
{
	__shared__ unsigned long long filt[FILT_WORDS];
	__shared__ uint64_t stage_all[4][STAGE];                          // per wave: pairs on their way out
	if (blockIdx.x >= n_work) return;
	const uint32_t rl = work[blockIdx.x].x, seg = work[blockIdx.x].y;
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const uint32_t r = cfg.r0 + rl;
	const uint64_t t0 = T.toff[rl]; const uint32_t tsz = (uint32_t)(T.toff[rl + 1] - t0);
	const uint32_t n_slots = cand_n[r] < cfg.c ? cand_n[r] : cfg.c;
	if (tsz == 0 || n_slots == 0) return;
	const uint32_t elen = A.lens[r];
	// read-level decision (encoder.cpp:1069-1078): refuse when too few distinct m-mers
	if ((double)n_distinct[rl] < cfg.frac_min * (double)elen && !((double)n_distinct[rl] > cfg.frac_always * (double)elen)) return;
	for (uint32_t i = threadIdx.x; i < FILT_WORDS; i += 256) filt[i] = 0;
	__syncthreads();
	uint32_t p_lo, p_hi;
	{
		const uint64_t ewb = A.word_off[r]; const uint32_t n = elen - cfg.m + 1;       // tsz != 0: elen >= m
		const uint32_t nseg = match_segments(n, seg_n);
		if (seg >= nseg) return;                                                        // (never: the host lists the segments by the same rule)
		p_lo = (uint32_t)((uint64_t)n * seg / nseg); p_hi = (uint32_t)((uint64_t)n * (seg + 1) / nseg);
		for (uint32_t p = p_lo + threadIdx.x; p < p_hi; p += 256)
		{
			const uint64_t xf = mmer_at(A, ewb, p, cfg.m), xr = revcomp_m(xf, cfg.m);
			const uint64_t hash = hash_mm(xf < xr ? xf : xr);
			atomicOr(&filt[filt_word(hash)], (unsigned long long)filt_mask(hash));
		}
	}
	__syncthreads();
	const uint64_t n0 = T.noff[rl];
	const uint64_t ewb = A.word_off[r];                                    // (the read itself: a tag match is confirmed on its m-mer)
	// Pairs are staged per wave in LDS and leave in runs of up to STAGE with ONE atomic add on the global counter (an add
	// per wave step — ~15 M a pass, all on one address — serialises the whole grid in the L2).
	uint64_t* stage = stage_all[wv]; uint32_t fill = 0;
	auto flush = [&]() {
		if (!fill) return;
		unsigned long long base = 0;
		if (lane == 0) base = atomicAdd(n_pairs, (unsigned long long)fill);
		base = ((unsigned long long)__shfl((int)(base >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 0, 64);
		if (base + fill <= cap) for (uint32_t i = lane; i < fill; i += 64) pairs[base + i] = stage[i];
		fill = 0;
	};
	for (uint32_t slot = 0; slot < n_slots; ++slot)
	{
		const uint32_t id = cand_refs[(uint64_t)r * cfg.c + slot], rlen = R.lens[id];
		if (rlen < cfg.m) continue;
		const uint64_t rwb = R.word_off[id];
		const uint32_t nq = rlen - cfg.m + 1, sl = rl * cfg.c + slot;
		const uint32_t PR = cfg.pr; const uint32_t pr_mask = (1u << PR) - 1;
		const uint64_t key_rev = (uint64_t)(2 * sl) << (cfg.pe + PR), key_fwd = (uint64_t)(2 * sl + 1) << (cfg.pe + PR);
		if (threadIdx.x == 0 && seg == 0) atomicAdd(n_pairs + 1, (unsigned long long)nq);      // probes, for the achieved-bandwidth report
		// Two steps are in flight: the reference words of step i + 2 are being loaded and the table slot of step i + 1 is being
		// probed (first slot of its probe sequence: at load 0.5 nearly every look-up ends there) while the hits of step i walk
		// their chains — one memory latency per step instead of three in a row (words, slot, chain).
		const uint64_t* rw = R.packed + rwb;
		struct Probe { uint64_t x; uint2 slot; uint32_t h, tag; bool pass, fwd_other, rev_other; };
		auto probe = [&](uint32_t q, uint64_t hi, uint64_t lo) -> Probe
		{	// y: the m-mer at q of the reference as stored; z: the one at nq - 1 - q of its reverse complement
			Probe P{ 0, make_uint2(0u, NIL), 0, 0, false, false, false };
			if (q < nq)
			{
				const uint64_t y = mmer_of(hi, lo, q, cfg.m), z = revcomp_m(y, cfg.m), x = y < z ? y : z;
				const uint64_t hash = hash_mm(x);
				const uint64_t fm = filt_mask(hash);
				if ((filt[filt_word(hash)] & fm) == fm)
				{
					P.pass = true; P.x = x; P.fwd_other = y != x; P.rev_other = z != x;
					P.h = table_slot(hash, tsz); P.tag = tag_of(hash);
					P.slot = T.slots[t0 + P.h];
				}
			}
			return P;
		};
		uint64_t hi1 = 0, lo1 = 0, hi2 = 0, lo2 = 0;
		Probe cur;
		{
			const uint32_t q = wv * 64 + lane;
			uint64_t hi = 0, lo = 0;
			if (q < nq) { hi = rw[q >> 5]; lo = rw[(q >> 5) + 1]; }
			if (q + 256 < nq) { hi1 = rw[(q + 256) >> 5]; lo1 = rw[((q + 256) >> 5) + 1]; }
			cur = probe(q, hi, lo);
		}
		const uint32_t rmask = (tsz < REGION ? tsz : REGION) - 1;             // (probing wraps inside the region the table was built by: table_heads)
		for (uint32_t q0 = wv * 64; q0 < nq; q0 += 256)
		{
			const uint32_t q = q0 + lane;
			hi2 = 0; lo2 = 0;
			if (q + 512 < nq) { hi2 = rw[(q + 512) >> 5]; lo2 = rw[((q + 512) >> 5) + 1]; }
			const Probe nxt = probe(q + 256, hi1, lo1);
			// the chain of this lane's m-mer (NIL: none), and how many of its elements are hits of the forward / of the reverse analysis
			uint32_t head = NIL, cnt = 0;
			const uint32_t fs = cur.fwd_other ? 1u : 0u, rs_ = cur.rev_other ? 1u : 0u;   // strand of the read positions that match y / z
			if (cur.pass)
			{
				uint2 sl8 = cur.slot; uint32_t h = cur.h;
				while (sl8.y != NIL)
				{
					if (sl8.x == cur.tag)
					{	// a tag is not the key: the read's own m-mer at the head's position decides
						const uint64_t xf = mmer_at(A, ewb, sl8.y & LINK_POS, cfg.m), xr = revcomp_m(xf, cfg.m);
						if ((xf < xr ? xf : xr) == cur.x) { head = sl8.y; break; }
					}
					h = (h & ~rmask) | ((h + 1) & rmask);
					sl8 = T.slots[t0 + h];
				}
				for (uint32_t p = head; p != NIL; p = T.next[n0 + (p & LINK_POS)])
					if ((p & LINK_POS) >= p_lo && (p & LINK_POS) < p_hi) cnt += (uint32_t)((p >> 31) == fs) + (uint32_t)((p >> 31) == rs_);
			}
			cur = nxt; hi1 = hi2; lo1 = lo2;
			if (!__any(cnt != 0)) continue;
			const uint32_t incl = wave_incl_scan(cnt), tot = __shfl(incl, 63, 64);
			const uint64_t kf = key_fwd | (uint64_t)(~q & pr_mask), kr = key_rev | (uint64_t)(~(nq - 1 - q) & pr_mask);
			if (fill + tot > STAGE) flush();
			if (tot <= STAGE)
			{
				uint32_t o = fill + incl - cnt;
				for (uint32_t p = head; p != NIL; p = T.next[n0 + (p & LINK_POS)])
				{
					if ((p & LINK_POS) < p_lo || (p & LINK_POS) >= p_hi) continue;
					const uint64_t pos = (uint64_t)(p & LINK_POS) << PR;
					if ((p >> 31) == fs) stage[o++] = kf | pos;
					if ((p >> 31) == rs_) stage[o++] = kr | pos;
				}
				fill += tot;
				continue;
			}
			// more hits in one step than the stage holds (long chains of a repeated m-mer): straight to the array
			unsigned long long base = 0;
			if (lane == 0) base = atomicAdd(n_pairs, (unsigned long long)tot);
			base = ((unsigned long long)__shfl((int)(base >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 0, 64);
			if (base + tot > cap) continue;
			uint64_t o = base + incl - cnt;
			for (uint32_t p = head; p != NIL; p = T.next[n0 + (p & LINK_POS)])
			{
				if ((p & LINK_POS) < p_lo || (p & LINK_POS) >= p_hi) continue;
				const uint64_t pos = (uint64_t)(p & LINK_POS) << PR;
				if ((p >> 31) == fs) pairs[o++] = kf | pos;
				if ((p >> 31) == rs_) pairs[o++] = kr | pos;
			}
		}
	}
	flush();
}
// ---- A1-A3 for reads whose table fits in LDS: built, counted and probed by ONE block, nothing of it ever in HBM ----
// (north_star: "LDS-staged hash-bucket probes".  Built, parity-tested against the tables in HBM and the oracle, measured, and left OFF: see
// cl_anchor_candidates_hifi below.)  Since only the SET of (read position, reference position) pairs matters up to the sort,
// the table is a multiset: every position of the read takes a slot of its own in the probe sequence of its canonical m-mer (linear
// probing, load 2/3), a slot is the position alone — 16 bits — and whether it holds the probing m-mer is decided on the read itself, which
// sits in LDS beside it (2 bits a base).  No keys, no chains, no tags: 3 bytes of table + 1/4 byte of read per base, so a read of 24 k bases
// with a 32-KB blocked Bloom filter in front (the same one-word three-bit filter as k_match: a miss — nearly every probe — costs one LDS
// word) takes 112 KB.  The number of distinct m-mers (the read-level decision, encoder.cpp:1069-1078) = the positions that are the first
// with their m-mer: a second pass of look-ups.  Per base: 2 bits of the read and 2 bits per candidate from HBM, 8 bytes per pair out.
template<uint32_t MAXN> struct LdsGeom { static constexpr uint32_t SLOTS = ((MAXN + MAXN / 2) + 17) & ~1u, WORDS = (MAXN + 27 + 31) / 32 + 2; };
__device__ inline uint32_t lds_table_slots(uint32_t n) { const uint32_t s = (n + n / 2 + 1) & ~1u; return s < 16 ? 16 : s; }
template<uint32_t MAXN, uint32_t FW, uint32_t NT, uint32_t WAVES_PER_SIMD>
__global__ __launch_bounds__(NT, WAVES_PER_SIMD) void k_match_lds(Arena A, Arena R, TaskCfg cfg, uint32_t min_n, const uint8_t* __restrict__ has_n,
                                                 const uint32_t* __restrict__ cand_refs, const uint32_t* __restrict__ cand_n, uint32_t* __restrict__ n_distinct, uint32_t n_reads,
                                                 unsigned long long* __restrict__ n_pairs, uint64_t cap, uint64_t* __restrict__ pairs, uint32_t dbg)
{
	static_assert(MAXN <= 65535 && (FW & (FW - 1)) == 0 && NT % 64 == 0, "positions are 16-bit links; 0xffff is the empty slot");
	constexpr uint32_t NW = NT / 64, LSTAGE = 64;
	__shared__ unsigned long long filt[FW];
	__shared__ uint64_t rd[LdsGeom<MAXN>::WORDS];                        // the read, 32 bases a word as in the arena
	__shared__ uint32_t slots32[LdsGeom<MAXN>::SLOTS / 2];               // two 16-bit slots a word (LDS atomics are 32-bit)
	__shared__ uint64_t stage_all[NW][LSTAGE];
	__shared__ uint32_t s_distinct;
	const uint32_t rl = blockIdx.x;
	if (rl >= n_reads) return;
	const uint32_t r = cfg.r0 + rl, m = cfg.m;
	const uint32_t elen = A.lens[r];
	const uint32_t n_slots = cand_n[r] < cfg.c ? cand_n[r] : cfg.c;
	if (n_slots == 0 || has_n[r] || elen < m) return;
	const uint32_t n = elen - m + 1;
	if (n <= min_n || n > MAXN) return;                                    // (another class's, or k_table_insert + k_match's)
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const uint32_t S = lds_table_slots(n);
	const uint16_t* slot16 = (const uint16_t*)slots32;
	{
		const uint64_t* src = A.packed + A.word_off[r];
		const uint32_t nw = (elen + 31) / 32 + 1;                              // (+ the word after the last: mmer_of takes two)
		for (uint32_t i = threadIdx.x; i < nw; i += NT) rd[i] = src[i];
		for (uint32_t i = threadIdx.x; i < FW; i += NT) filt[i] = 0;
		for (uint32_t i = threadIdx.x; i < S / 2; i += NT) slots32[i] = 0xffffffffu;
		if (threadIdx.x == 0) s_distinct = 0;
	}
	__syncthreads();
	if (dbg == 1) return;
	auto own = [&](uint32_t p) -> uint64_t { return mmer_of(rd[p >> 5], rd[(p >> 5) + 1], p, m); };
	auto step = [&](uint32_t off) -> uint32_t { return off + 1 == S ? 0u : off + 1; };
	{	// Insertion.  Every position with the same m-mer sits in the run of occupied slots that starts at the m-mer's own slot, and a position
		// walks over every slot in front of the one it takes: of two equal m-mers exactly one — the one further along — sees the other.  So
		// the distinct m-mers of the read (encoder.cpp:1069-1078) = the positions that met no equal on their way, counted here
		uint32_t fresh = 0;
		for (uint32_t p = threadIdx.x; p < n; p += NT)
		{
			const uint64_t xf = own(p), xr = revcomp_m(xf, m), hash = hash_mm(xf < xr ? xf : xr);
			atomicOr(&filt[(uint32_t)(hash >> 46) & (FW - 1)], (unsigned long long)filt_mask(hash));
			bool dup = false;
			for (uint32_t off = table_slot(hash, S);;)
			{
				uint32_t* w = &slots32[off >> 1]; const uint32_t sh = (off & 1) * 16;
				const uint32_t old = *(volatile uint32_t*)w, q = (old >> sh) & 0xffffu;
				if (q != 0xffffu) { if (!dup && own(q) == xf) dup = true; off = step(off); continue; }
				if (atomicCAS(w, old, (old & ~(0xffffu << sh)) | (p << sh)) == old) break;   // (else: the word changed — maybe only its other half: look again)
			}
			fresh += dup ? 0u : 1u;
		}
		fresh = wave_sum(fresh);
		if (lane == 0 && fresh) atomicAdd(&s_distinct, fresh);
	}
	__syncthreads();
	const uint32_t nd = s_distinct;
	if (threadIdx.x == 0) n_distinct[rl] = nd;
	if (dbg == 2) return;
	if ((double)nd < cfg.frac_min * (double)elen && !((double)nd > cfg.frac_always * (double)elen)) return;
	uint64_t* stage = stage_all[wv]; uint32_t fill = 0;
	auto flush = [&]() {
		if (!fill) return;
		unsigned long long base = 0;
		if (lane == 0) base = atomicAdd(n_pairs, (unsigned long long)fill);
		base = ((unsigned long long)__shfl((int)(base >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 0, 64);
		if (base + fill <= cap) for (uint32_t i = lane; i < fill; i += 64) pairs[base + i] = stage[i];
		fill = 0;
	};
	// y: the m-mer at q of the reference as stored; z: the one at nq - 1 - q of its reverse complement.  A position of the read that has y is
	// a hit of the forward analysis, one that has z of the reverse one (a palindrome: of both).  The first two hits stay in registers
	struct Look { uint64_t y, z; uint32_t h0, cnt, nh, hp0, hp1, hf0, hf1; };
	auto look = [&](bool live, uint32_t q, uint64_t hi, uint64_t lo) -> Look {
		Look L{ 0, 0, 0, 0, 0, 0, 0, 0, 0 };
		if (!live) return L;
		L.y = mmer_of(hi, lo, q, m); L.z = revcomp_m(L.y, m);
		const uint64_t hash = hash_mm(L.y < L.z ? L.y : L.z), fm = filt_mask(hash);
		if ((filt[(uint32_t)(hash >> 46) & (FW - 1)] & fm) != fm || dbg == 3) return L;
		L.h0 = table_slot(hash, S);
		for (uint32_t off = L.h0;; off = step(off))
		{
			const uint32_t p = slot16[off];
			if (p == 0xffffu) break;
			const uint64_t xf = own(p);
			const uint32_t f = (xf == L.y ? 1u : 0u) | (xf == L.z ? 2u : 0u);
			if (!f) continue;
			if (L.nh == 0) { L.hp0 = p; L.hf0 = f; } else if (L.nh == 1) { L.hp1 = p; L.hf1 = f; }
			++L.nh; L.cnt += (f & 1) + (f >> 1);
		}
		return L;
	};
	for (uint32_t slot = 0; slot < n_slots; ++slot)
	{
		const uint32_t id = cand_refs[(uint64_t)r * cfg.c + slot], rlen = R.lens[id];
		if (rlen < m) continue;
		const uint64_t* rw = R.packed + R.word_off[id];
		const uint32_t nq = rlen - m + 1, sl = rl * cfg.c + slot;
		const uint32_t PR = cfg.pr; const uint32_t pr_mask = (1u << PR) - 1;
		const uint64_t key_rev = (uint64_t)(2 * sl) << (cfg.pe + PR), key_fwd = (uint64_t)(2 * sl + 1) << (cfg.pe + PR);
		if (threadIdx.x == 0) atomicAdd(n_pairs + 2, (unsigned long long)nq);
		// A wave takes 128 consecutive positions a step, a lane the positions q and q + 64 (two independent look-ups in flight: their LDS
		// reads overlap) out of four consecutive words of the reference, which are loaded a step ahead
		const uint32_t last_w = (rlen + 31) / 32;                              // (the word after the read's last exists in the arena)
		auto words = [&](uint32_t q, uint64_t (&w)[4]) {
			const uint32_t b = q >> 5;
#pragma unroll
			for (uint32_t i = 0; i < 4; ++i) w[i] = (q < nq && b + i <= last_w) ? rw[b + i] : 0;
		};
		uint64_t w[4], w1[4];
		words(wv * 128 + lane, w);
		for (uint32_t q0 = wv * 128; q0 < nq; q0 += 2 * NT)
		{
			const uint32_t qa = q0 + lane, qb = qa + 64;
			words(qa + 2 * NT, w1);
			const Look La = look(qa < nq, qa, w[0], w[1]), Lb = look(qb < nq, qb, w[2], w[3]);
#pragma unroll
			for (uint32_t i = 0; i < 4; ++i) w[i] = w1[i];
			const uint32_t cnt = La.cnt + Lb.cnt;
			if (!__any(cnt != 0)) continue;
			const uint32_t incl = wave_incl_scan(cnt), tot = __shfl(incl, 63, 64);
			auto put = [&](auto* dst, uint64_t o) {
				auto all_of = [&](const Look& L, uint32_t q) {
					if (!L.cnt) return;
					const uint64_t kf = key_fwd | (uint64_t)(~q & pr_mask), kr = key_rev | (uint64_t)(~(nq - 1 - q) & pr_mask);
					auto one = [&](uint32_t p, uint32_t f) { const uint64_t pos = (uint64_t)p << PR; if (f & 1) dst[o++] = kf | pos; if (f & 2) dst[o++] = kr | pos; };
					if (L.nh <= 2) { one(L.hp0, L.hf0); if (L.nh > 1) one(L.hp1, L.hf1); return; }
					for (uint32_t off = L.h0;; off = step(off))                   // (a repeated m-mer: walk its run again)
					{
						const uint32_t p = slot16[off];
						if (p == 0xffffu) break;
						const uint64_t xf = own(p);
						const uint32_t f = (xf == L.y ? 1u : 0u) | (xf == L.z ? 2u : 0u);
						if (f) one(p, f);
					}
				};
				all_of(La, qa); all_of(Lb, qb);
			};
			if (fill + tot > LSTAGE) flush();
			if (tot <= LSTAGE) { put(stage, (uint64_t)(fill + incl - cnt)); fill += tot; continue; }
			unsigned long long base = 0;                                       // more hits in one step than the stage holds: straight to the array
			if (lane == 0) base = atomicAdd(n_pairs, (unsigned long long)tot);
			base = ((unsigned long long)__shfl((int)(base >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 0, 64);
			if (base + tot > cap) continue;
			put(pairs, (uint64_t)(base + incl - cnt));
		}
	}
	flush();
}
// the pairs of every task in the sorted array; a task with "too many matches" keeps none unless its read is always
// encoded (encoder.cpp:1034-1042; enc_read.size() counts the guard)
__global__ void k_task_pairs(const uint64_t* __restrict__ pairs, uint64_t n_pairs, Arena A, TaskCfg cfg, const uint32_t* __restrict__ n_distinct, uint32_t n_tasks,
                             uint64_t* __restrict__ pair_off, uint32_t* __restrict__ pair_cnt)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t > n_tasks) return;
	auto lower = [&](uint64_t task) -> uint64_t {
		const uint64_t key = task << (cfg.pe + cfg.pr);
		uint64_t lo = 0, hi = n_pairs;
		while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (pairs[mid] < key) lo = mid + 1; else hi = mid; }
		return lo;
	};
	const uint64_t a = lower(t);
	pair_off[t] = a;
	if (t == n_tasks) return;
	uint32_t total = (uint32_t)(lower((uint64_t)t + 1) - a);
	const uint32_t rl = t / (2 * cfg.c); const uint32_t elen = A.lens[cfg.r0 + rl];
	const bool always = (double)n_distinct[rl] > cfg.frac_always * (double)elen;
	if (!always && (double)total > cfg.max_mult * (double)(elen + 1)) total = 0;
	pair_cnt[t] = total;
}

// ---- A5: one lane per task: LIS (utils.cpp:157-209), map-back (encoder.cpp:644-658), MergeAnchors (:731-776) ----
__device__ inline int lis_search(const int* __restrict__ tf, int size, int value)
{
	int low = 0;
	while (size > 0)
	{
		const int half = size / 2, other_half = size - half, probe = low + half, other_low = low + other_half;
		const int v = tf[probe];
		size = half;
		low = value > v ? other_low : low;
	}
	return low;
}
__global__ __launch_bounds__(64) void k_lis_anchors(Arena A, Arena R, TaskCfg cfg, const uint32_t* __restrict__ cand_refs, uint32_t n_tasks,
                                                   const uint64_t* __restrict__ pair_off, const uint32_t* __restrict__ pair_cnt, const uint64_t* __restrict__ pairs,
                                                   int* __restrict__ tf, int* __restrict__ ts, int* __restrict__ pred,
                                                   uint32_t* __restrict__ anch, uint32_t* __restrict__ t_nanch, uint32_t* __restrict__ t_tot, uint32_t dbg)
{
	__builtin_amdgcn_s_setprio(3);                                          // a launch of this kernel lasts as long as its slowest chain: its waves go first on their SIMDs
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_tasks) return;
	const uint64_t a = pair_off[t], b = a + pair_cnt[t];
	uint32_t n_anch = 0, tot = 0;
	if (b > a)
	{
		const uint32_t n = (uint32_t)(b - a);
		const uint64_t* P = pairs + a;
		int* F = tf + a; int* S = ts + a; int* PR = pred + a;
		const uint32_t pr_mask = (1u << cfg.pr) - 1, pe_mask = (1u << cfg.pe) - 1;
		auto ref_pos = [&](uint32_t i) -> int { return (int)(~(uint32_t)P[i] & pr_mask); };
		auto enc_pos = [&](uint32_t i) -> uint32_t { return (uint32_t)(P[i] >> cfg.pr) & pe_mask; };
		F[0] = ref_pos(0); S[0] = 0; PR[0] = -1;
		int out_len = 1;
		int last_f = F[0], last_s = 0;                                  // F / S at out_len - 1: the common step (the next match extends
		for (int i = 1; i < (int)n; ++i)                                // the chain) then needs no load but its own pair
		{
			const int x = ref_pos((uint32_t)i);
			if (last_f < x)
			{
				F[out_len] = x; S[out_len] = i; PR[i] = last_s;
				++out_len; last_f = x; last_s = i;
				continue;
			}
			const int pos = lis_search(F, out_len, x);
			F[pos] = x; S[pos] = i;
			PR[i] = pos > 0 ? S[pos - 1] : -1;
			if (pos == out_len - 1) { last_f = x; last_s = i; }
		}
		if (dbg == 1) { t_nanch[t] = 0; t_tot[t] = 0; return; }
		// chain in increasing order: walk the predecessor links backwards, storing the pair indices in S (reused)
		int cur = S[out_len - 1];
		for (int i = out_len - 1; i >= 0; --i) { F[i] = cur; cur = PR[cur]; }       // F[i] = pair index of chain element i
		if (dbg == 2) { t_nanch[t] = 0; t_tot[t] = 0; return; }
		// map-back + merge.  The reference re-derives the enc position by scanning the distinct enc positions forward for
		// the next one that carries the chain element's m-mer.
		const uint32_t rl = t / (2 * cfg.c), slot = (t / 2) % cfg.c; const bool rev = (t & 1) == 0;
		const uint32_t r = cfg.r0 + rl;
		const uint32_t id = cand_refs[(uint64_t)r * cfg.c + slot];
		const uint64_t ewb = A.word_off[r];
		const uint32_t rlen = R.lens[id]; const uint64_t rwb = R.word_off[id];
		uint32_t* out = anch + 3 * a;
		uint32_t ep = 0;                                   // index into the pairs, positioned on the first pair of a distinct enc position
		uint32_t pe_at_ep = enc_pos(0);                    // ... and that position (kept from the advance below: no load of its own)
		uint32_t pe_ahead = n > 1 ? enc_pos(1) : 0xffffffffu;   // ... and the one of pair ep + 1, fetched an advance ahead
		uint32_t run = 0, start_e = 0, start_r = 0, prev_e = 0, prev_r = 0;
		// This loop was 70 % of the kernel: per chain element a chain of four dependent loads (its pair, the reference words of its m-mer, the
		// pair under the scan, the read's words there), 2 us each on the longest list of a launch.  Everything of element i + 1 that does not
		// depend on the scan — its pair, its reference m-mer, the read's m-mer at ITS OWN position (where the scan nearly always hits) — is
		// fetched while element i is worked on; the scan's own pair comes from the advance of the step before.
		// THREE stages, one load level each, so that no load is waited for in the iteration that issues it: element i + 3's pair index,
		// element i + 2's pair, element i + 1's four sequence words (raw: the m-mers are cut out of them an iteration later).
		auto m_from = [&](uint64_t hi, uint64_t lo, uint32_t p) -> uint64_t {
			const uint32_t j = p & 31, sft = 128 - 2 * (j + cfg.m);
			const uint64_t v = (sft >= 64) ? (hi >> (sft - 64)) : ((hi << (64 - sft)) | (lo >> sft));
			return v & ((1ULL << (2 * cfg.m)) - 1);
		};
		auto idx_of = [&](int x) -> int { return x < out_len ? F[x] : F[out_len - 1]; };
		struct Raw { uint32_t pr, e, rp; uint64_t r0, r1, a0, a1; };
		auto words_of = [&](uint64_t pw) -> Raw {
			Raw w; w.pr = ~(uint32_t)pw & pr_mask; w.e = (uint32_t)(pw >> cfg.pr) & pe_mask;
			w.rp = rev ? rlen - cfg.m - w.pr : w.pr;
			const uint64_t* rw = R.packed + rwb + (w.rp >> 5); const uint64_t* aw = A.packed + ewb + (w.e >> 5);
			w.r0 = rw[0]; w.r1 = rw[1]; w.a0 = aw[0]; w.a1 = aw[1];
			return w;
		};
		int idx3 = idx_of(2);                                             // (filled to the pipeline's depth before the first element)
		uint64_t pw2 = P[idx_of(1)];
		Raw w1 = words_of(P[idx_of(0)]);
		for (int i = 0; i < out_len; ++i)
		{
			// this element: cut its m-mers out of the words fetched an iteration ago
			const uint32_t pr = w1.pr, e_own = w1.e;
			const uint64_t mraw = m_from(w1.r0, w1.r1, w1.rp), mm = rev ? revcomp_m(mraw, cfg.m) : mraw, am = m_from(w1.a0, w1.a1, w1.e);
			// the stages move up: words of element i + 1 from its pair, the pair of element i + 2 from its index, the index of element i + 3
			w1 = words_of(pw2);
			pw2 = P[idx3];
			idx3 = idx_of(i + 3);
			uint32_t pe;
			for (;;)
			{
				pe = pe_at_ep;
				const bool hit = (pe == e_own ? am : mmer_at(A, ewb, pe, cfg.m)) == mm;
				uint32_t nxt = pe;
				do { ++ep; nxt = pe_ahead; pe_ahead = ep + 1 < n ? enc_pos(ep + 1) : 0xffffffffu; } while (ep < n && nxt == pe);      // advance to the next distinct enc position
				pe_at_ep = nxt;
				if (hit || ep >= n) break;                              // (the scan always hits before the end; the bound only guards against a hang)
			}
			if (run && prev_e == pe - 1 && prev_r == pr - 1) ++run;
			else
			{
				if (run) { out[3 * n_anch] = run + cfg.m - 1; out[3 * n_anch + 1] = start_e; out[3 * n_anch + 2] = start_r; tot += run + cfg.m - 1; ++n_anch; }
				run = 1; start_e = pe; start_r = pr;
			}
			prev_e = pe; prev_r = pr;
		}
		if (run) { out[3 * n_anch] = run + cfg.m - 1; out[3 * n_anch + 1] = start_e; out[3 * n_anch + 2] = start_r; tot += run + cfg.m - 1; ++n_anch; }
	}
	t_nanch[t] = n_anch; t_tot[t] = tot;
}

// ---- A5': HiFi k-mer anchors (a9): AnalyseRefReadWithKmers / KmerBasedAnchors (encoder.cpp:870-1013,1113-1147) ------
// One lane per (read, candidate).  The shared k-mers of the pair (canonical, from the similarity graph) seed anchors of
// length k where the k-mer is unique in both reads on the same strand pairing; they must be colinear, overlapping ones
// are dropped, then every anchor is extended base by base and touching anchors merge.  Both orientations of the
// reference are analysed; a candidate with an accepted orientation keeps these anchors instead of the m-mer ones.
struct KmerArgs {
	const uint64_t* common; const uint64_t* common_off;      // per (read, slot): shared k-mers in read order (cl_candidates_common)
	uint64_t* sorted; uint32_t* tab;                          // scratch: per shared k-mer 1 u64 + 8 u32 (count / position per read and strand)
	uint32_t* kanch;                                          // per shared k-mer 2 x 3 u32: anchors of the rc / forward analysis
	uint32_t* k_n; uint32_t* k_tot; uint8_t* k_use;           // per task (2 per slot): anchors, total length; per slot: 0 rc, 1 fwd, 2 none
	uint64_t base;                                            // common_off value of the chunk's first slot
	uint32_t k; ModTest mt;
};
__device__ inline uint32_t oriented_sym(const Arena& R, uint64_t wb, uint32_t len, bool rev, uint32_t pos)
{
	const uint32_t p = rev ? len - 1 - pos : pos;
	const uint32_t b = (uint32_t)(R.packed[wb + (p >> 5)] >> (62 - 2 * (p & 31))) & 3u;
	return rev ? 3u - b : b;
}
// occurrences of the shared k-mers in one read: per k-mer and strand a count and the (last) start position
__device__ inline void kmer_occurrences(const Arena& A, uint64_t wb, uint32_t len, uint32_t k, const ModTest& mt, const uint64_t* sorted, uint32_t n, uint32_t* cntF, uint32_t* posF, uint32_t* cntR, uint32_t* posR)
{
	for (uint32_t i = 0; i < n; ++i) { cntF[i] = cntR[i] = 0; posF[i] = posR[i] = 0; }
	if (len < k) return;
	const uint64_t mask = (1ULL << (2 * k)) - 1; uint64_t f = 0, r = 0;
	for (uint32_t p = 0; p < len; ++p)
	{
		const uint64_t b = (A.packed[wb + (p >> 5)] >> (62 - 2 * (p & 31))) & 3u;
		f = ((f << 2) + b) & mask; r = (r >> 2) + ((3 - b) << (2 * (k - 1)));
		if (p + 1 < k) continue;
		const uint64_t can = f < r ? f : r;
		if (!mod_is_zero(hash_mm(can), mt)) continue;
		uint32_t lo = 0, hi = n;
		while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sorted[mid] < can) lo = mid + 1; else hi = mid; }
		if (lo >= n || sorted[lo] != can) continue;
		if (f == can) { ++cntF[lo]; posF[lo] = p + 1 - k; } else { ++cntR[lo]; posR[lo] = p + 1 - k; }
	}
}
// returns 0 no anchors, 1 incompatible, 2 accepted; anchors (len, pos_enc, pos_ref) x n_out
__device__ inline uint32_t kmer_anchor_chain(const Arena& A, uint64_t ewb, uint32_t elen, const Arena& R, uint64_t rwb, uint32_t rlen, bool rev, uint32_t k,
                                             uint32_t* a, uint32_t n, uint32_t* n_out, uint32_t* tot_out)
{
	*n_out = 0; *tot_out = 0;
	if (n == 0) return 0;
	auto L = [&](uint32_t i) -> uint32_t& { return a[3 * i]; }; auto E = [&](uint32_t i) -> uint32_t& { return a[3 * i + 1]; }; auto Q = [&](uint32_t i) -> uint32_t& { return a[3 * i + 2]; };
	auto erase = [&](uint32_t i) { for (uint32_t j = i; j + 1 < n; ++j) { a[3 * j] = a[3 * j + 3]; a[3 * j + 1] = a[3 * j + 4]; a[3 * j + 2] = a[3 * j + 5]; } --n; };
	auto enc = [&](uint32_t p) -> uint32_t { return (uint32_t)(A.packed[ewb + (p >> 5)] >> (62 - 2 * (p & 31))) & 3u; };
	auto ref = [&](uint32_t p) -> uint32_t { return oriented_sym(R, rwb, rlen, rev, p); };
	for (uint32_t i = 1; i < n; ++i)                                             // by position in the read (positions are distinct)
	{
		const uint32_t x0 = a[3 * i], x1 = a[3 * i + 1], x2 = a[3 * i + 2]; uint32_t j = i;
		while (j > 0 && a[3 * (j - 1) + 1] > x1) { a[3 * j] = a[3 * j - 3]; a[3 * j + 1] = a[3 * j - 2]; a[3 * j + 2] = a[3 * j - 1]; --j; }
		a[3 * j] = x0; a[3 * j + 1] = x1; a[3 * j + 2] = x2;
	}
	for (uint32_t i = 1; i < n; ++i) if (Q(i) < Q(i - 1)) return 1;
	for (uint64_t i = 0; i + 1 < n; ++i)                                         // drop overlapping k-mers (:912-920)
		if (E((uint32_t)i) + L((uint32_t)i) > E((uint32_t)i + 1) || Q((uint32_t)i) + L((uint32_t)i) > Q((uint32_t)i + 1)) { erase((uint32_t)i + 1); --i; }
	while (E(0) > 0 && Q(0) > 0 && enc(E(0) - 1) == ref(Q(0) - 1)) { --E(0); --Q(0); ++L(0); }
	for (uint64_t ii = 0; ii < n; ++ii)
	{
		uint32_t i = (uint32_t)ii;
		if (i > 0)
		{
			const uint32_t pe = E(i - 1) + L(i - 1), pr = Q(i - 1) + L(i - 1);
			for (;;)
			{
				const bool re = E(i) == pe, rr = Q(i) == pr;
				if (re && rr) { L(i) += L(i - 1); erase(i - 1); break; }               // as the reference: the merged anchor keeps its own positions (:944-948)
				if (re || rr) break;
				if (enc(E(i) - 1) != ref(Q(i) - 1)) break;
				++L(i); --E(i); --Q(i);
			}
		}
		if (i != n - 1)
		{
			const uint32_t ne_ = E(i + 1), nr_ = Q(i + 1);
			uint32_t pe = E(i) + L(i), pr = Q(i) + L(i);
			for (;;)
			{
				const bool re = pe == ne_, rr = pr == nr_;
				if (re && rr) { L(i) += L(i + 1); erase(i + 1); --ii; break; }
				else if (re || rr) break;
				if (enc(pe) != ref(pr)) break;
				++pe; ++pr; ++L(i);
			}
		}
	}
	{
		const uint32_t l = n - 1; uint32_t pe = E(l) + L(l), pr = Q(l) + L(l);
		while (pe < elen && pr < rlen && enc(pe) == ref(pr)) { ++pe; ++pr; ++L(l); }
	}
	uint32_t tot = 0; for (uint32_t i = 0; i < n; ++i) tot += L(i);
	*n_out = n; *tot_out = tot;
	return 2;
}
__global__ void k_kmer_anchors(Arena A, Arena R, TaskCfg cfg, const uint32_t* __restrict__ cand_refs, const uint32_t* __restrict__ cand_n, const uint8_t* __restrict__ has_n,
                               const uint32_t* __restrict__ n_distinct, KmerArgs ka)
{
	const uint64_t sl = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;              // slot inside the chunk
	const uint32_t c = cfg.c, rl = (uint32_t)(sl / c), j = (uint32_t)(sl % c);
	const uint32_t r = cfg.r0 + rl;
	if (r >= cfg.r1) return;
	ka.k_use[sl] = 2; ka.k_n[2 * sl] = ka.k_n[2 * sl + 1] = 0; ka.k_tot[2 * sl] = ka.k_tot[2 * sl + 1] = 0;
	if (has_n[r] || j >= cand_n[r]) return;
	{	// read-level decision (encoder.cpp:1069-1078), as in k_match
		const double el = (double)A.lens[r], nd = (double)n_distinct[rl];
		if (nd < cfg.frac_min * el && !(nd > cfg.frac_always * el)) return;
	}
	const uint64_t gs = (uint64_t)r * c + j;
	const uint64_t o = ka.common_off[gs] - ka.base; const uint32_t n = (uint32_t)(ka.common_off[gs + 1] - ka.common_off[gs]);
	if (n == 0) return;
	uint64_t* sorted = ka.sorted + o;
	for (uint32_t i = 0; i < n; ++i)                                              // ascending (the reference sorts the list, :1117)
	{
		const uint64_t x = ka.common[ka.common_off[gs] + i]; uint32_t q = i;
		while (q > 0 && sorted[q - 1] > x) { sorted[q] = sorted[q - 1]; --q; }
		sorted[q] = x;
	}
	uint32_t* tab = ka.tab + 8 * o;
	uint32_t* eF = tab, * ePF = tab + n, * eR = tab + 2 * n, * ePR = tab + 3 * n, * rF = tab + 4 * n, * rPF = tab + 5 * n, * rR = tab + 6 * n, * rPR = tab + 7 * n;
	const uint32_t id = cand_refs[gs];
	const uint64_t ewb = A.word_off[r], rwb = R.word_off[id]; const uint32_t elen = A.lens[r], rlen = R.lens[id], k = ka.k;
	kmer_occurrences(A, ewb, elen, k, ka.mt, sorted, n, eF, ePF, eR, ePR);
	kmer_occurrences(R, rwb, rlen, k, ka.mt, sorted, n, rF, rPF, rR, rPR);
	uint32_t st[2], na[2], tt[2];
	for (uint32_t orient = 0; orient < 2; ++orient)                                  // 0: reverse complement of the reference, 1: as stored
	{
		const bool rev = orient == 0;
		uint32_t* a = ka.kanch + 6 * o + (uint64_t)orient * 3 * n; uint32_t m = 0;
		for (uint32_t i = 0; i < n; ++i)
		{
			// forward text of the k-mer in the read and in the oriented reference, else its reverse complement in both
			// (in the reverse-complemented reference a forward occurrence is a reverse-strand occurrence of the stored read)
			const uint32_t cF = rev ? rR[i] : rF[i], cR = rev ? rF[i] : rR[i];
			const uint32_t pF = rev ? rlen - k - rPR[i] : rPF[i], pR = rev ? rlen - k - rPF[i] : rPR[i];
			if (eF[i] == 1 && cF == 1) { a[3 * m] = k; a[3 * m + 1] = ePF[i]; a[3 * m + 2] = pF; ++m; }
			else if (eR[i] == 1 && cR == 1) { a[3 * m] = k; a[3 * m + 1] = ePR[i]; a[3 * m + 2] = pR; ++m; }
		}
		st[orient] = kmer_anchor_chain(A, ewb, elen, R, rwb, rlen, rev, k, a, m, &na[orient], &tt[orient]);
		ka.k_n[2 * sl + orient] = st[orient] == 2 ? na[orient] : 0; ka.k_tot[2 * sl + orient] = st[orient] == 2 ? tt[orient] : 0;
	}
	uint8_t use = 2;
	if (st[0] == 2 && st[1] == 2) use = tt[1] > tt[0] ? 1 : 0;                       // reverse complement wins ties (:1127-1139)
	else if (st[0] == 2) use = 0;
	else if (st[1] == 2) use = 1;
	ka.k_use[sl] = use;
}

// ---- A6: per read: orientation choice, stable sort by total anchor length (encoder.cpp:1106-1108,1157-1191) --------
__global__ void k_select(TaskCfg cfg, const uint32_t* __restrict__ cand_refs, const uint32_t* __restrict__ cand_n, uint32_t min_anchors,
                         const uint32_t* __restrict__ t_nanch, const uint32_t* __restrict__ t_tot,
                         const uint8_t* __restrict__ k_use, const uint32_t* __restrict__ k_n, const uint32_t* __restrict__ k_tot,
                         uint32_t* __restrict__ o_ncand, uint32_t* __restrict__ o_cand, uint32_t* __restrict__ o_task, uint32_t* __restrict__ o_count)
{
	const uint32_t r = cfg.r0 + blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= cfg.r1) return;
	const uint32_t rl = r - cfg.r0, c = cfg.c;
	uint32_t n = 0;
	uint32_t sel_task[16], sel_tot[16], sel_n[16];                          // task | 0x80000000: the anchors are the k-mer ones
	const uint32_t nc = cand_n[r] < c ? cand_n[r] : c;
	for (uint32_t j = 0; j < nc && j < 16; ++j)
	{
		const uint32_t trc = (rl * c + j) * 2, tfw = trc + 1;
		const bool arc = t_nanch[trc] >= min_anchors && t_nanch[trc] > 0, afw = t_nanch[tfw] >= min_anchors && t_nanch[tfw] > 0;
		const uint32_t ku = k_use ? k_use[(uint64_t)rl * c + j] : 2u;
		if (ku != 2) { sel_task[n] = (trc + ku) | 0x80000000u; sel_tot[n] = k_tot[trc + ku]; sel_n[n] = k_n[trc + ku]; ++n; continue; }   // HiFi k-mer anchors (:1235-1239)
		int pick = -1;
		if (arc && afw) pick = t_tot[tfw] > t_tot[trc] ? 1 : 0;       // reverse complement wins ties
		else if (arc) pick = 0;
		else if (afw) pick = 1;
		if (pick >= 0) { sel_task[n] = trc + (uint32_t)pick; sel_tot[n] = t_tot[trc + pick]; sel_n[n] = t_nanch[trc + pick]; ++n; }
	}
	for (uint32_t i = 1; i < n; ++i)                                 // insertion sort = libstdc++ std::sort on <= 16 elements, stable
	{
		uint32_t xt = sel_task[i], xv = sel_tot[i], xn = sel_n[i]; uint32_t j = i;
		while (j > 0 && xv > sel_tot[j - 1]) { sel_task[j] = sel_task[j - 1]; sel_tot[j] = sel_tot[j - 1]; sel_n[j] = sel_n[j - 1]; --j; }
		sel_task[j] = xt; sel_tot[j] = xv; sel_n[j] = xn;
	}
	o_ncand[r] = n;
	for (uint32_t i = 0; i < c; ++i)
	{
		const uint64_t s = (uint64_t)r * c + i;
		if (i < n)
		{
			const uint32_t t = sel_task[i] & 0x7fffffffu, slot = (t / 2) % c;
			o_cand[4 * s] = cand_refs[(uint64_t)r * c + slot]; o_cand[4 * s + 1] = (t & 1) == 0 ? 1u : 0u;
			o_cand[4 * s + 2] = sel_tot[i]; o_cand[4 * s + 3] = sel_n[i];
			o_task[(uint64_t)rl * c + i] = sel_task[i]; o_count[(uint64_t)rl * c + i] = sel_n[i];
		}
		else { o_cand[4 * s] = ~0u; o_cand[4 * s + 1] = 0; o_cand[4 * s + 2] = 0; o_cand[4 * s + 3] = 0; o_task[(uint64_t)rl * c + i] = ~0u; o_count[(uint64_t)rl * c + i] = 0; }
	}
}
// ---- A7: copy the chosen anchors, trimming overlaps in reference then read coordinates (encoder.cpp:1577-1622) ------
__global__ void k_copy_fix(const uint32_t* __restrict__ o_task, const uint64_t* __restrict__ slot_off, uint64_t n_slots,
                           const uint64_t* __restrict__ pair_off, const uint32_t* __restrict__ anch, uint64_t dst_base, uint32_t* __restrict__ out,
                           TaskCfg cfg, const uint32_t* __restrict__ kanch, const uint64_t* __restrict__ common_off, uint64_t common_base)
{
	const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_slots) return;
	const uint32_t tm = o_task[s];
	if (tm == ~0u) return;
	const uint32_t t = tm & 0x7fffffffu;
	const uint32_t n = (uint32_t)(slot_off[s + 1] - slot_off[s]);
	const uint32_t* src = anch + 3 * pair_off[t];
	if (tm & 0x80000000u)
	{	// k-mer anchors of candidate slot t/2 of the chunk, orientation t&1
		const uint64_t gs = (uint64_t)cfg.r0 * cfg.c + t / 2;
		const uint64_t o = common_off[gs] - common_base; const uint32_t nk = (uint32_t)(common_off[gs + 1] - common_off[gs]);
		src = kanch + 6 * o + (uint64_t)(t & 1) * 3 * nk;
	}
	uint32_t* dst = out + 3 * (dst_base + slot_off[s]);
	for (uint32_t i = 0; i < 3 * n; ++i) dst[i] = src[i];
	for (uint32_t i = 0; i + 1 < n; ++i)
	{
		const uint32_t end = dst[3 * i + 2] + dst[3 * i];
		if (dst[3 * (i + 1) + 2] < end) { const uint32_t d = end - dst[3 * (i + 1) + 2]; dst[3 * (i + 1) + 2] += d; dst[3 * (i + 1)] -= d; dst[3 * (i + 1) + 1] += d; }
	}
	for (uint32_t i = 0; i + 1 < n; ++i)
	{
		const uint32_t end = dst[3 * i + 1] + dst[3 * i];
		if (dst[3 * (i + 1) + 1] < end) { const uint32_t d = end - dst[3 * (i + 1) + 1]; dst[3 * (i + 1) + 1] += d; dst[3 * (i + 1)] -= d; dst[3 * (i + 1) + 2] += d; }
	}
}
__global__ void k_add_u64(uint64_t* v, uint64_t n, uint64_t c) { uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] += c; }
} // namespace

extern "C" cl_status cl_anchor_candidates_hifi(cl_ctx* ctx, const cl_reads* reads, const cl_reads* refs, const uint32_t* d_cand_refs, const uint32_t* d_cand_n,
                                               uint32_t c, uint32_t anchor_len, double frac_always, double frac_min, double max_matches_mult,
                                               uint32_t min_anchors, uint32_t kmer_len, uint32_t modulo, const uint64_t* d_common_off, const uint64_t* d_common, cl_anchors** out);
extern "C" cl_status cl_anchor_candidates(cl_ctx* ctx, const cl_reads* reads, const cl_reads* refs, const uint32_t* d_cand_refs, const uint32_t* d_cand_n,
                                          uint32_t c, uint32_t anchor_len, double frac_always, double frac_min, double max_matches_mult,
                                          uint32_t min_anchors, cl_anchors** out)
{
	return cl_anchor_candidates_hifi(ctx, reads, refs, d_cand_refs, d_cand_n, c, anchor_len, frac_always, frac_min, max_matches_mult, min_anchors, 0, 0, nullptr, nullptr, out);
}
extern "C" cl_status cl_anchor_candidates_hifi(cl_ctx* ctx, const cl_reads* reads, const cl_reads* refs, const uint32_t* d_cand_refs, const uint32_t* d_cand_n,
                                               uint32_t c, uint32_t anchor_len, double frac_always, double frac_min, double max_matches_mult,
                                               uint32_t min_anchors, uint32_t kmer_len, uint32_t modulo, const uint64_t* d_common_off, const uint64_t* d_common, cl_anchors** out)
{
	if (!ctx || !reads || !refs || !d_cand_refs || !d_cand_n || !out) return cl_fail(ctx, CL_E_INVALID, "cl_anchor_candidates: null argument");
	const bool hifi = d_common_off != nullptr;
	if (hifi && (kmer_len < 2 || kmer_len > 28 || modulo == 0)) return cl_fail(ctx, CL_E_INVALID, "cl_anchor_candidates_hifi: 2 <= kmer_len <= 28, modulo >= 1");
	if (c == 0 || c > 16 || anchor_len < 2 || anchor_len > 28) return cl_fail(ctx, CL_E_INVALID, "cl_anchor_candidates: 1 <= c <= 16, 2 <= anchor_len <= 28");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const uint32_t nr = reads->n_reads, m = anchor_len;
	cl_anchors* X = new cl_anchors(); X->ctx = ctx; X->n_reads = nr; X->c = c;
	std::unique_ptr<cl_anchors> guard(X);
	DEV_ALLOC(ctx, X->n_cands, nr); DEV_ALLOC(ctx, X->cand, (uint64_t)nr * c * 4); DEV_ALLOC(ctx, X->cand_off, (uint64_t)nr * c + 1);
	Arena A{ reads->packed.p, reads->word_off.p, reads->lens.p }, R{ refs->packed.p, refs->word_off.p, refs->lens.p };
	std::vector<uint32_t> h_len(nr);
	if (nr) HIP_TRY(ctx, hipMemcpy(h_len.data(), reads->lens.p, (uint64_t)nr * 4, hipMemcpyDeviceToHost));
	auto bits_for = [](uint64_t v) -> uint32_t { uint32_t b = 1; while ((1ull << b) <= v) ++b; return b; };       // bits that hold 0..v
	uint32_t max_ref_len = 0;
	if (refs->n_reads)
	{
		std::vector<uint32_t> rl(refs->n_reads);
		HIP_TRY(ctx, hipMemcpy(rl.data(), refs->lens.p, (uint64_t)refs->n_reads * 4, hipMemcpyDeviceToHost));
		for (uint32_t l : rl) max_ref_len = std::max(max_ref_len, l);
	}
	const uint32_t pr_bits = bits_for(max_ref_len);
	// batches of reads: bounded by bases (tables) — the pair count is checked per batch
	const uint64_t BATCH_BASES = 1ull << 28;
	std::vector<DevBuf<uint32_t>> chunks; std::vector<uint64_t> chunk_n;
	uint64_t total_anchors = 0;
	double pairs_per_base = 0.5;                                     // match pairs per read base seen so far (room for the next batch)
	// The table build of a batch (random atomics, bound by the memory system) runs on the context's side stream while the
	// main stream matches, sorts and chains the batch before it (ALU / latency bound): batches are independent.
	struct TableBatch {
		uint32_t r0 = 0, r1 = 0, pe = 1; uint64_t acc = 0, nsum = 0;
		DevBuf<uint32_t> n_distinct, next, err; DevBuf<uint64_t> toff, noff; DevBuf<EncSlot> slots; DevBuf<uint2> bins;
		std::vector<uint2> h_work; DevBuf<uint2> work;                           // k_match's blocks: (read of the batch, segment of its positions)
		struct SideSync { hipStream_t s = nullptr; ~SideSync() { if (s) (void)hipStreamSynchronize(s); } } sync;   // destroyed first
	};
	HIP_TRY(ctx, cl_side_stream(ctx, ctx->side));
	// COLORD_HIP_ANCHORS_LDS=12288 | 24576: reads of up to that many m-mers keep their table in LDS (k_match_lds, two size classes).  OFF by
	// default — measured (round 6, profiles/r06_l_*): alone on the device the two classes take 14 + 25 ms per Gbase for the 61 % of the bases
	// whose tables k_table_insert + k_match make and probe in 6.7 + 15 ms (the probes that reach a table are few: the Bloom filter in LDS stops
	// 99 % of the misses in both forms), and in the pipeline a block that wants 63 / 118 KB of a CU's LDS waits for it: 104 / 214 ms per chunk,
	// 21.5 s per pass against 17.0
	constexpr uint32_t LDS_N_A = 12288, LDS_N_B = 24576, LDS_NT = 1024;
	uint32_t lds_max_n = 0;
	if (const char* e = getenv("COLORD_HIP_ANCHORS_LDS")) lds_max_n = atoi(e) <= 0 ? 0u : (uint32_t)atoi(e) <= LDS_N_A ? LDS_N_A : LDS_N_B;
	// reads above 1.5 x this many m-mers are matched segment by segment (k_match); COLORD_HIP_MATCH_SEG=0: one block per read whatever its length
	const uint32_t match_seg = getenv("COLORD_HIP_MATCH_SEG") ? (uint32_t)std::max(0, atoi(getenv("COLORD_HIP_MATCH_SEG"))) : 16384u;
	const uint32_t lds_dbg = getenv("COLORD_HIP_LDS_DBG") ? (uint32_t)atoi(getenv("COLORD_HIP_LDS_DBG")) : 0u;
	auto prepare = [&](uint32_t r0, std::unique_ptr<TableBatch>& out) -> cl_status {
		out = std::make_unique<TableBatch>();
		TableBatch& B = *out;
		uint32_t r1 = r0; uint64_t acc = 0;
		// the batch also ends where its sort keys would outgrow 64 bits: task bits + position bits of its longest read + those of the longest candidate
		uint32_t mx = 0;
		while (r1 < nr && (r1 == r0 || acc + h_len[r1] <= BATCH_BASES))
		{
			const uint32_t pe = bits_for(std::max(mx, h_len[r1]));
			if (r1 > r0 && pe + pr_bits + bits_for((r1 - r0 + 1) * 2 * c) > 64) break;
			mx = std::max(mx, h_len[r1]); acc += h_len[r1]; ++r1;
		}
		B.r0 = r0; B.r1 = r1; B.acc = acc; B.pe = bits_for(mx);
		for (uint32_t r = r0; r < r1; ++r)
			for (uint32_t sg = 0, ns = match_segments(h_len[r] >= m ? h_len[r] - m + 1 : 0, match_seg); sg < ns; ++sg) B.h_work.push_back(make_uint2(r - r0, sg));
		if (B.pe + pr_bits + bits_for((r1 - r0) * 2 * c) > 64) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_anchor_candidates: a read and its candidates are too long for 64-bit match keys");
		const uint32_t nb = r1 - r0;
		const uint32_t table_x4 = 8;                                          // table slots per m-mer, in quarters (8 = load 0.5; denser tables were slower: longer probe chains)
		// EVERYTHING of the batch on the side stream (round 6: sizes and offsets ran on the main stream, in front of the match pass of the batch
		// before — its two waits for scan totals and the allocation of the tables, 6 ms of host work, were a gap on the lane's main queue per
		// batch): the caller queues the match pass of the current batch first and prepares this one while it runs
		LaunchOn on(ctx, ctx->side);                                             // (launches, scans + timing events on the side stream)
		B.sync.s = ctx->side;
		DevBuf<uint32_t> tsize, nsize, err; DEV_ALLOC(ctx, tsize, nb); DEV_ALLOC(ctx, nsize, nb); DEV_ALLOC(ctx, err, 1); DEV_ALLOC(ctx, B.n_distinct, nb);
		HIP_TRY(ctx, hipMemsetAsync(err.p, 0, 4, ctx->side));
		LAUNCH(ctx, k_table_sizes, grid_for(nb, 256), 256, (const uint32_t*)reads->lens.p, (const uint8_t*)reads->has_n.p, d_cand_n, r0, r1, m, table_x4, lds_max_n, tsize.p, nsize.p, err.p);
		DEV_ALLOC(ctx, B.toff, (uint64_t)nb + 1); DEV_ALLOC(ctx, B.noff, (uint64_t)nb + 1);
		uint64_t tsum = 0;
		CL_TRY(dev_exclusive_scan_u64(ctx, tsize.p, B.toff.p, nb, &tsum));
		CL_TRY(dev_exclusive_scan_u64(ctx, nsize.p, B.noff.p, nb, &B.nsum));      // (waits for the side stream: sizes, offsets and err are complete)
		uint32_t herr = 0; HIP_TRY(ctx, hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, ctx->side)); HIP_TRY(ctx, hipStreamSynchronize(ctx->side));
		DEV_ALLOC(ctx, B.slots, tsum); DEV_ALLOC(ctx, B.next, B.nsum); DEV_ALLOC(ctx, B.bins, B.nsum); DEV_ALLOC(ctx, B.err, 1);
		DEV_ALLOC(ctx, B.work, B.h_work.size() + 1);
		if (!B.h_work.empty()) HIP_TRY(ctx, hipMemcpyAsync(B.work.p, B.h_work.data(), B.h_work.size() * sizeof(uint2), hipMemcpyHostToDevice, ctx->side));   // (h_work lives as long as the batch)
		HIP_TRY(ctx, hipMemsetAsync(B.n_distinct.p, 0, (uint64_t)nb * 4, ctx->side));
		HIP_TRY(ctx, hipMemsetAsync(B.err.p, 0, 4, ctx->side));                  // (the slots are written region by region, all of them)
		EncTable T{ B.slots.p, B.toff.p, B.next.p, B.noff.p };
		LAUNCHB(ctx, B.nsum * (0.25 + 16.0) /* 2 bits in, two 8-byte slots out per m-mer */, k_table_insert, nb, INS_T, A, r0, r1, m, T, B.n_distinct.p, B.bins.p, B.err.p);
		HIP_TRY(ctx, hipGetLastError());
		return CL_OK;                                                            // (the temporaries above were read by kernels the waits above saw end)
	};
	std::unique_ptr<TableBatch> cur, nxt;
	if (nr) CL_TRY(prepare(0, cur));
	while (cur)
	{
		const uint32_t r0 = cur->r0, r1 = cur->r1; const uint64_t acc = cur->acc;
		const uint32_t nb = r1 - r0, n_tasks = nb * 2 * c;
		TaskCfg cfg{ r0, r1, c, m, cur->pe, pr_bits, frac_always, frac_min, max_matches_mult };
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                         // the batch before is through (its buffers may be handed to the side stream)
		HIP_TRY(ctx, hipStreamSynchronize(ctx->side));                           // this batch's tables are built
		cur->sync.s = nullptr;
		{ uint32_t herr = 0; HIP_TRY(ctx, hipMemcpy(&herr, cur->err.p, 4, hipMemcpyDeviceToHost)); if (herr) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_anchor_candidates: a region of an m-mer table overflowed"); }
		DevBuf<uint32_t>& n_distinct = cur->n_distinct;
		EncTable T{ cur->slots.p, cur->toff.p, cur->next.p, cur->noff.p };
		DevBuf<uint32_t> pair_cnt; DEV_ALLOC(ctx, pair_cnt, (uint64_t)n_tasks + 1);
		DevBuf<uint64_t> pair_off; DEV_ALLOC(ctx, pair_off, (uint64_t)n_tasks + 1);
		DevBuf<unsigned long long> d_np; DEV_ALLOC(ctx, d_np, 4);           // match pairs, probes (tables in HBM), probes (tables in LDS)
		DevBuf<uint64_t> pairs;
		uint64_t n_pairs = 0;
		for (uint64_t cap = (uint64_t)(pairs_per_base * 1.25 * (double)acc) + (1u << 20);;)
		{	// one pass when the room guessed from the batches before suffices, else a second with the counted size
			DEV_ALLOC(ctx, pairs, cap);
			HIP_TRY(ctx, hipMemsetAsync(d_np.p, 0, 32, ctx->stream));
			if (lds_max_n)
			{
				LAUNCH_NAMED(ctx, "k_match_lds<12288>", (k_match_lds<LDS_N_A, 2048, LDS_NT, 8>), nb, LDS_NT, A, R, cfg, 0u, (const uint8_t*)reads->has_n.p, d_cand_refs, d_cand_n, n_distinct.p, nb, d_np.p, cap, pairs.p, lds_dbg);
				if (lds_max_n > LDS_N_A)
					LAUNCH_NAMED(ctx, "k_match_lds<24576>", (k_match_lds<LDS_N_B, 4096, LDS_NT, 4>), nb, LDS_NT, A, R, cfg, LDS_N_A, (const uint8_t*)reads->has_n.p, d_cand_refs, d_cand_n, n_distinct.p, nb, d_np.p, cap, pairs.p, lds_dbg);
			}
			LAUNCHB(ctx, 0.0, k_match, (uint32_t)cur->h_work.size(), 256, A, R, T, cfg, d_cand_refs, d_cand_n, (const uint32_t*)n_distinct.p, (const uint2*)cur->work.p, (uint32_t)cur->h_work.size(), match_seg, d_np.p, cap, pairs.p);
			HIP_TRY(ctx, hipGetLastError());
			unsigned long long h_np2[2] = { 0, 0 };
			HIP_TRY(ctx, hipMemcpyAsync(h_np2, d_np.p, 16, hipMemcpyDeviceToHost, ctx->stream));
			// while the match pass runs: the next batch's table sizes, offsets, allocations and its table build, all on the side stream
			if (r1 < nr && !nxt) CL_TRY(prepare(r1, nxt));
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
			const unsigned long long h_np = h_np2[0];
			n_pairs = h_np;
			// algorithmic bytes of the pass: 2 bits of reference and one 8-byte key slot per probe, 8 bytes per pair out
			if (ctx->timing && !ctx->pending_bytes.empty()) ctx->pending_bytes.back() = (double)h_np2[1] * 8.25 + (double)std::min<uint64_t>(h_np, cap) * 8.0;
			if (getenv("COLORD_HIP_ANCHOR_DEBUG")) fprintf(stderr, "[anchors] batch reads %u..%u bases %llu: pairs %llu cap %llu\n", r0, r1, (unsigned long long)acc, h_np, (unsigned long long)cap);
			if (n_pairs >= (1ull << 32)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_anchor_candidates: batch produces >= 2^32 match pairs");
			if (n_pairs <= cap) break;
			cap = n_pairs;
		}
		if (acc) pairs_per_base = std::max(pairs_per_base, (double)n_pairs / (double)acc);
		DevBuf<int> tf, ts, pred; DEV_ALLOC(ctx, tf, n_pairs); DEV_ALLOC(ctx, ts, n_pairs); DEV_ALLOC(ctx, pred, n_pairs);
		DevBuf<uint32_t> anch; DEV_ALLOC(ctx, anch, 3 * n_pairs);
		DevBuf<uint32_t> t_nanch, t_tot; DEV_ALLOC(ctx, t_nanch, n_tasks); DEV_ALLOC(ctx, t_tot, n_tasks);
		if (n_pairs)
		{
			uint32_t tb = 1; while ((1ull << tb) < n_tasks) ++tb;
			CL_TRY(dev_sort_pairs(ctx, pairs.p, nullptr, n_pairs, 0, cfg.pe + cfg.pr + tb));
		}
		LAUNCH(ctx, k_task_pairs, grid_for((uint64_t)n_tasks + 1, 256), 256, (const uint64_t*)pairs.p, n_pairs, A, cfg, (const uint32_t*)n_distinct.p, n_tasks, pair_off.p, pair_cnt.p);
		LAUNCHB(ctx, n_pairs * 32.0, k_lis_anchors, grid_for(n_tasks, 64), 64, A, R, cfg, d_cand_refs, n_tasks, (const uint64_t*)pair_off.p, (const uint32_t*)pair_cnt.p, (const uint64_t*)pairs.p,
			tf.p, ts.p, pred.p, anch.p, t_nanch.p, t_tot.p, (uint32_t)(getenv("COLORD_HIP_LIS_DBG") ? atoi(getenv("COLORD_HIP_LIS_DBG")) : 0));
		HIP_TRY(ctx, hipGetLastError());
		const uint64_t n_slots = (uint64_t)nb * c;
		// HiFi: k-mer anchors from the shared k-mers of every (read, candidate)
		DevBuf<uint64_t> k_sorted; DevBuf<uint32_t> k_tab, kanch, k_n, k_tot; DevBuf<uint8_t> k_use;
		uint64_t common_base = 0;
		if (hifi)
		{
			uint64_t hb[2] = { 0, 0 };
			HIP_TRY(ctx, hipMemcpy(&hb[0], d_common_off + (uint64_t)r0 * c, 8, hipMemcpyDeviceToHost));
			HIP_TRY(ctx, hipMemcpy(&hb[1], d_common_off + (uint64_t)r1 * c, 8, hipMemcpyDeviceToHost));
			common_base = hb[0];
			const uint64_t nk = hb[1] - hb[0];
			DEV_ALLOC(ctx, k_sorted, nk + 1); DEV_ALLOC(ctx, k_tab, 8 * nk + 8); DEV_ALLOC(ctx, kanch, 6 * nk + 6);
			DEV_ALLOC(ctx, k_n, 2 * n_slots); DEV_ALLOC(ctx, k_tot, 2 * n_slots); DEV_ALLOC(ctx, k_use, n_slots);
			KmerArgs ka{ d_common, d_common_off, k_sorted.p, k_tab.p, kanch.p, k_n.p, k_tot.p, k_use.p, common_base, kmer_len, make_modtest(modulo) };
			LAUNCH(ctx, k_kmer_anchors, grid_for(n_slots, 64), 64, A, R, cfg, d_cand_refs, d_cand_n, (const uint8_t*)reads->has_n.p, (const uint32_t*)n_distinct.p, ka);
			HIP_TRY(ctx, hipGetLastError());
		}
		DevBuf<uint32_t> o_task, o_count; DEV_ALLOC(ctx, o_task, n_slots); DEV_ALLOC(ctx, o_count, n_slots);
		LAUNCH(ctx, k_select, grid_for(nb, 128), 128, cfg, d_cand_refs, d_cand_n, min_anchors, (const uint32_t*)t_nanch.p, (const uint32_t*)t_tot.p,
			(const uint8_t*)(hifi ? k_use.p : nullptr), (const uint32_t*)k_n.p, (const uint32_t*)k_tot.p,
			X->n_cands.p, X->cand.p, o_task.p, o_count.p);
		DevBuf<uint64_t> slot_off; DEV_ALLOC(ctx, slot_off, n_slots + 1);
		uint64_t n_here = 0;
		CL_TRY(dev_exclusive_scan_u64(ctx, o_count.p, slot_off.p, n_slots, &n_here));
		DevBuf<uint32_t> chunk; DEV_ALLOC(ctx, chunk, 3 * n_here);
		LAUNCH(ctx, k_copy_fix, grid_for(n_slots, 128), 128, (const uint32_t*)o_task.p, (const uint64_t*)slot_off.p, n_slots, (const uint64_t*)pair_off.p,
			(const uint32_t*)anch.p, (uint64_t)0, chunk.p, cfg, (const uint32_t*)kanch.p, d_common_off, common_base);
		// global candidate offsets of this batch
		HIP_TRY(ctx, hipMemcpyAsync(X->cand_off.p + (uint64_t)r0 * c, slot_off.p, n_slots * 8, hipMemcpyDeviceToDevice, ctx->stream));
		if (total_anchors) LAUNCH(ctx, k_add_u64, grid_for(n_slots, 256), 256, X->cand_off.p + (uint64_t)r0 * c, n_slots, total_anchors);
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		chunks.push_back(std::move(chunk)); chunk_n.push_back(n_here);
		total_anchors += n_here;
		cur = std::move(nxt);
	}
	X->n_anchors = total_anchors;
	DEV_ALLOC(ctx, X->anchors, 3 * total_anchors);
	uint64_t o = 0;
	for (size_t i = 0; i < chunks.size(); ++i)
	{
		if (chunk_n[i]) HIP_TRY(ctx, hipMemcpyAsync(X->anchors.p + 3 * o, chunks[i].p, chunk_n[i] * 12, hipMemcpyDeviceToDevice, ctx->stream));
		o += chunk_n[i];
	}
	HIP_TRY(ctx, hipMemcpyAsync(X->cand_off.p + (uint64_t)nr * c, &total_anchors, 8, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	cl_timing_collect(ctx);
	*out = guard.release();
	return CL_OK;
}
extern "C" void cl_anchors_free(cl_anchors* a) { delete a; }
extern "C" uint64_t cl_anchors_total(const cl_anchors* a) { return a->n_anchors; }
extern "C" const uint32_t* cl_anchors_n_cands(const cl_anchors* a) { return a->n_cands.p; }
extern "C" const uint32_t* cl_anchors_cands(const cl_anchors* a) { return a->cand.p; }
extern "C" const uint64_t* cl_anchors_cand_offsets(const cl_anchors* a) { return a->cand_off.p; }
extern "C" const uint32_t* cl_anchors_data(const cl_anchors* a) { return a->anchors.p; }
