// dna.hip — DNA stream coder (a14 + a16 framing) on the GPU, bit-identical to CDNACoder
// (src/colord/dna_coder.{h,cpp}) driven by CEntrComprReads (entr_read.h:56-80).
//
// Same decomposition as the quality coder (qual.hip): every coded symbol's (model family, context, symbol,
// exclusions) is a function of INPUT data only — the read's tuple stream, the reference reads it points
// to and the types of the previous four reads — so symbols are generated independently per read (one
// lane walks one tuple stream; plain reads are expanded one lane per base), grouped by (family, context)
// with a stable sort, each model is evolved by one wavefront over its own run, and the interval
// arithmetic runs one lane per part (rc_dev.hpp).
#include "common.hpp"
#include <functional>
#include "objects.hpp"
#include "rc_dev.hpp"
#include <algorithm>
#include <deque>
#include <thread>
#include <chrono>

namespace {
enum { T_INS = 0, T_DEL, T_MATCH, T_SUBST, T_ANCHOR, T_SKIP, T_ALT_ID, T_MAIN_REF, T_PLAIN, T_START_PLAIN, T_START_ES, T_START_PLAIN_N, T_NONE };
enum { F_READ_TYPE = 0, F_REV_COMP, F_SEEN, F_LEN_BITS, F_LEN_DATA, F_SYMBOLS, F_SYMBOLS_N, F_READ_ID, F_READ_ID_SHORT,
       F_ANCHOR_LEN, F_SKIP_LOCAL, F_SKIP_DISTANT, F_TUPLE_TYPE, N_FAM };
constexpr uint32_t MAX_ALT = 64;

struct FamTab {
	uint32_t n_sym[N_FAM], max_total[N_FAM], adder[N_FAM];
	uint32_t ctx_base[N_FAM + 1];       // dense global context ids
	uint64_t state_base[N_FAM + 1];     // offsets into the state array (u32 units)
	uint32_t n_ctx[N_FAM];
	int32_t level, T, S;                // tuple / symbol history lengths (dna_coder.cpp:1253-1280)
	uint32_t sym_B;                     // bit width of the regular symbols-family contexts
	uint32_t max_alt;
	uint32_t long_run;                  // runs of a small-alphabet model at least this long take the k_long_* path
	uint32_t walk_chunk, walk_warm;     // tuples per chunk of the write pass; tuples before a saved state with the symbol history kept
};

struct RefStore { const uint64_t* packed; const uint64_t* word_off; const uint32_t* lens; uint32_t n; };
// GetRefRead(id, rev)[pos] including the trailing guard 255 (reference_reads.h:142-207)
__device__ inline uint32_t ref_at(const RefStore& R, uint32_t id, bool rev, int64_t pos)
{
	if (id >= R.n) return 255;
	const uint32_t len = R.lens[id];
	if (pos < 0 || pos >= (int64_t)len) return 255;
	const uint32_t p = rev ? (len - 1 - (uint32_t)pos) : (uint32_t)pos;
	const uint32_t b = (uint32_t)(R.packed[R.word_off[id] + (p >> 5)] >> (62 - 2 * (p & 31))) & 3u;
	return rev ? 3u - b : b;
}
// the same through a per-lane cursor: length / word offset of the current reference and the last packed word stay in
// registers, so a run of match tuples costs one load per 32 bases instead of three dependent loads per base
struct RefCur {
	uint32_t id = 0xffffffffu, len = 0, widx = 0xffffffffu; bool rev = false, ok = false; uint64_t wo = 0, word = 0;
	__device__ inline void set(const RefStore& R, uint32_t nid, bool nrev)
	{
		rev = nrev;
		if (nid == id) return;
		id = nid; widx = 0xffffffffu; ok = nid < R.n;
		if (ok) { len = R.lens[nid]; wo = R.word_off[nid]; }
	}
	__device__ inline uint32_t at(const RefStore& R, int64_t pos)
	{
		if (!ok || pos < 0 || pos >= (int64_t)len) return 255;
		const uint32_t p = rev ? (len - 1 - (uint32_t)pos) : (uint32_t)pos;
		if ((p >> 5) != widx) { widx = p >> 5; word = R.packed[wo + widx]; }
		const uint32_t b = (uint32_t)(word >> (62 - 2 * (p & 31))) & 3u;
		return rev ? 3u - b : b;
	}
};
__device__ inline uint32_t ilog2_(uint64_t x) { return x ? 64u - (uint32_t)__clzll((long long)x) : 0u; }     // bit length (basic_coder.h:39-47)
__device__ inline uint32_t no_bytes_(uint64_t x) { uint32_t r = 1; x >>= 8; for (; x; ++r) x >>= 8; return r; }

// A coded symbol's KEY, what the stable sort by context and the model kernels work on: dense context id << 8 | payload.  The payload is
// the symbol and, in the two families that code with exclusions (rc.h:316-341), WHICH symbols are excluded — as a small class, not as
// two 4-bit symbols (rounds 1-5: context << 16 | e1 << 12 | e2 << 8 | symbol, 40 bits, always a 64-bit word):
//   F_TUPLE_TYPE  x << 3 | type     x = what the tuple before forbids (dna_coder.cpp:735-757): 0 nothing, 1 after a match, 2 after a deletion,
//                                   3 after an anchor, 4 after a skip, 5 after an alternative-reference / main-reference tuple
//   F_SYMBOLS     x << 2 | base     x = 0, or 1 + the reference base a substitution excludes
//   the others    symbol (up to 256 of them)
// At level 1 the 8.7 M contexts take 24 bits: the key is ONE 32-bit word — the walk writes, the three sort passes move, and the model
// kernels read half the bytes (levels 2 and 3: 25 / 26 bits, 64-bit keys of the same form).
template<typename K> struct KeyFmt {
	static __host__ __device__ inline K make(uint32_t gctx, uint32_t payload) { return ((K)gctx << 8) | (K)payload; }
	static __host__ __device__ inline uint32_t ctx(K k) { return (uint32_t)(k >> 8); }
	static __host__ __device__ inline uint32_t payload(K k) { return (uint32_t)k & 0xffu; }
};
__device__ inline uint32_t key_payload(int fam, uint32_t sym, int e1, int e2)
{
	if (fam == F_TUPLE_TYPE)
	{
		const uint32_t x = e1 == 15 ? 0u : e2 == 15 ? (e1 == T_ANCHOR ? 1u : 2u) : e1 == T_ANCHOR ? 3u : e1 == T_DEL ? 4u : 5u;
		return (x << 3) | sym;
	}
	if (fam == F_SYMBOLS) return ((e1 == 15 ? 0u : (uint32_t)e1 + 1u) << 2) | sym;
	return sym;
}
// symbol and exclusions (15 = none) of a payload; fam is uniform over a context run
__device__ inline void key_decode(uint32_t fam, uint32_t p, uint32_t& sym, uint32_t& e1, uint32_t& e2)
{
	if (fam == F_TUPLE_TYPE)
	{
		const uint32_t x = p >> 3; sym = p & 7u;
		e1 = x == 0 ? 15u : (x == 1 || x == 3) ? (uint32_t)T_ANCHOR : x == 2 ? (uint32_t)T_SKIP : x == 4 ? (uint32_t)T_DEL : (uint32_t)T_ALT_ID;
		e2 = x == 3 ? (uint32_t)T_MATCH : x == 4 ? (uint32_t)T_SKIP : x == 5 ? (uint32_t)T_MAIN_REF : 15u;
	}
	else if (fam == F_SYMBOLS) { const uint32_t x = p >> 2; sym = p & 3u; e1 = x ? x - 1u : 15u; e2 = 15u; }
	else { sym = p; e1 = e2 = 15u; }
}
__device__ inline uint32_t key_sym_mask(uint32_t fam) { return fam == F_TUPLE_TYPE ? 7u : fam == F_SYMBOLS ? 3u : 0xffu; }

// The write pass's keys on their way out.  A lane emits its symbols one at a time to consecutive places, between long
// stretches of reading: stored one by one, the stores that fill a 64-byte sector arrive so far apart that the L2 writes the sector
// back in between (PMC: 37 GB written per launch for 8.8 GB of keys) — and every one of them is one more small scattered write for the
// memory system.  STAGED: the lane keeps the sector it is filling in LDS (slot = place mod the keys of a sector) and writes
// it as four 16-byte stores back to back when it is full; the first and the last sector of its range, shared with the neighbouring
// chunks' lanes (and, for plain reads, with k_dna_plain), go out key by key.
template<typename K>
struct Emitter {
	static constexpr uint32_t SLOTS = 64 / sizeof(K);                        // keys of a 64-byte sector
	bool write; K* key; uint64_t off; uint32_t count; const FamTab* ft;
	K* stage = nullptr; uint64_t first = 0; bool any = false;              // the lane's SLOTS slots in LDS (nullptr: direct stores); first place written
	__device__ inline void put(K k)
	{
		const uint64_t g = off + count;
		if (!stage) { key[g] = k; return; }
		if (!any) { first = g; any = true; }
		stage[g & (SLOTS - 1)] = k;
		if ((g & (SLOTS - 1)) != SLOTS - 1) return;
		const uint64_t base = g & ~(uint64_t)(SLOTS - 1);
		if (first <= base)
		{
			uint4* dst = (uint4*)(key + base); const uint32_t* sw = (const uint32_t*)stage;
#pragma unroll
			for (int q = 0; q < 4; ++q) dst[q] = make_uint4(sw[4 * q], sw[4 * q + 1], sw[4 * q + 2], sw[4 * q + 3]);
		}
		else for (uint64_t x = first; x <= g; ++x) key[x] = stage[x & (SLOTS - 1)];
	}
	// the keys of the sector the lane was filling when it stopped
	__device__ inline void finish()
	{
		if (!stage || !any) return;
		const uint64_t g = off + count, base = g & ~(uint64_t)(SLOTS - 1);
		for (uint64_t x = first > base ? first : base; x < g; ++x) key[x] = stage[x & (SLOTS - 1)];
	}
	__device__ inline void operator()(int fam, uint32_t ctx, uint32_t sym, int e1 = 15, int e2 = 15)
	{
		if (write)
		{
			put(KeyFmt<K>::make(ft->ctx_base[fam] + ctx, key_payload(fam, sym, e1, e2)));
			// the triple index of symbol i, trip_index(lay, part, i), is a function of i alone within a read: k_fill_sidx writes
			// them coalesced instead of one scattered 4-byte store per symbol here
		}
		++count;
	}
};

struct EsReader {
	const uint8_t* p; const uint8_t* e;
	uint64_t w0 = 0, w1 = 0; uint32_t have = 0;                  // up to 16 prefetched stream bytes (the walk is latency-bound on them)
	__device__ inline void refill()
	{	// 8-byte aligned loads of the words that hold stream bytes (never a word entirely past the end)
		const uint64_t a = (uint64_t)(size_t)p; const uint32_t sh = (uint32_t)(a & 7);
		const uint64_t* q = (const uint64_t*)(a & ~7ull);
		const uint64_t* lim = (const uint64_t*)(((uint64_t)(size_t)e + 7) & ~7ull);   // words holding at least one stream byte
		const uint64_t x0 = q[0], x1 = q + 1 < lim ? q[1] : 0ull, x2 = q + 2 < lim ? q[2] : 0ull;
		w0 = sh ? (x0 >> (8 * sh)) | (x1 << (64 - 8 * sh)) : x0;
		w1 = sh ? (x1 >> (8 * sh)) | (x2 << (64 - 8 * sh)) : x1;
		have = 16;
	}
	__device__ inline void drop(uint32_t n) { p += n; have -= n; w0 = (w0 >> (8 * n)) | (w1 << (64 - 8 * n)); w1 >>= 8 * n; }
	__device__ inline bool next(uint32_t& type, uint32_t& v1, uint32_t& v2)
	{
		if (p >= e) return false;
		if (have < 5) refill();
		const uint32_t b0 = (uint32_t)(w0 & 0xff), t = b0 >> 4; type = t;
		switch (t)
		{
		case T_INS: case T_SUBST: case T_PLAIN: v1 = b0 & 0xf; drop(1); break;
		case T_ANCHOR: case T_SKIP: v2 = ((b0 & 0xf) << 24) | ((uint32_t)((w0 >> 8) & 0xff) << 16) | ((uint32_t)((w0 >> 16) & 0xff) << 8) | (uint32_t)((w0 >> 24) & 0xff); drop(4); break;
		case T_ALT_ID: case T_START_ES: v2 = b0 & 0xf; v1 = ((uint32_t)((w0 >> 8) & 0xff) << 24) | ((uint32_t)((w0 >> 16) & 0xff) << 16) | ((uint32_t)((w0 >> 24) & 0xff) << 8) | (uint32_t)((w0 >> 32) & 0xff); drop(5); break;
		default: drop(1);
		}
		return true;
	}
};

template<typename K> __device__ inline void emit_read_len(Emitter<K>& em, uint32_t len)                  // dna_coder.cpp:1004-1056
{
	int nb = (int)ilog2_(len);
	em(F_LEN_BITS, 0, (uint32_t)nb);
	if (nb < 2) return;
	uint32_t ctx = (uint32_t)nb << 3;
	len -= 1u << (nb - 1);
	uint32_t prefix, suffix;
	if (nb <= 9) { prefix = len; suffix = 0; }
	else { prefix = len >> (nb - 9); suffix = len - (prefix << (nb - 9)); }
	em(F_LEN_DATA, ctx, prefix);
	if (nb <= 9) return;
	nb -= 9; ctx += 4;
	for (; nb > 0; nb -= 8) { em(F_LEN_DATA, ctx, suffix & 0xff); suffix >>= 8; ++ctx; }
}
template<typename K> __device__ inline void emit_read_id(Emitter<K>& em, uint32_t id, uint32_t cur_read_id)   // :535-551
{
	const int n = (int)no_bytes_(cur_read_id);
	for (int i = n - 1; i >= 0; --i)
	{
		uint32_t add = (i == n - 2) ? ((id >> (8 * (n - 1))) & 0xff) : 0;
		em(F_READ_ID, (uint32_t)i + (add << 3), (id >> (8 * i)) & 0xff);
	}
}
template<typename K> __device__ inline void emit_anchor_len(Emitter<K>& em, uint32_t len)                // :958-978
{
	for (uint32_t part = 0; len; ++part)
	{
		if (len < 23) { em(F_ANCHOR_LEN, part, len); break; }
		em(F_ANCHOR_LEN, part, 23);
		len -= 22;
	}
}
template<typename K> __device__ inline void emit_skip_len(Emitter<K>& em, uint32_t len, bool local)      // :1109-1137
{
	if (local)
	{
		for (uint32_t part = 0; len; ++part)
		{
			if (len < 255) { em(F_SKIP_LOCAL, part, len); break; }
			em(F_SKIP_LOCAL, part, 255);
			len -= 254;
		}
		return;
	}
	uint32_t encoded = 0;
	for (int i = 3; i >= 0; --i)
	{
		uint32_t x = (len >> (8 * i)) & 0xff;
		em(F_SKIP_DISTANT, (uint32_t)i * 64 + ilog2_(encoded), x);
		encoded = (encoded << 8) + x;
	}
}

// ---- D1: the walk over the tuple streams (CDNACoder::Encode, dna_coder.cpp:26-231) ----------
// A read's tuples are a chain (contexts, reference cursors, the tables of alternative references), and one lane walking
// a 170 k-tuple read takes ~0.1 s however idle the machine is.  So the walk is done twice, differently:
//   count pass  one lane per READ: counts the symbols and, every WALK_CHUNK tuples, saves the state of the walk;
//   write pass  one lane per CHUNK: resumes from the saved state and writes the keys of its WALK_CHUNK tuples.
// The count pass is the shorter chain: no stores, one-byte tuples eight at a time, and the symbol history (the only state
// that needs reference symbols) only over the WALK_WARM tuples before a saved state — enough unless fewer than S of
// them add a symbol, in which case the read is walked again keeping the history throughout.  The write pass is bounded
// by a chunk, not a read.
constexpr uint32_t WALK_LPW = 64, WALK_CHUNK = 4096, WALK_WARM = 128;   // defaults of FamTab::walk_chunk / walk_warm (COLORD_HIP_WALK_CHUNK / _WARM override them, for tests)
struct WalkCk {
	uint64_t byte_off, ctx_tuple, ctx_symbol; int64_t ref_pos, alt_pos;
	uint32_t read, sym, ctx_rev, n_rc, n_alt, alt_id; int32_t alt_slot, delta; uint32_t last_type, last_flag, flags, pad;   // flags: is_main | first << 1 | alt_rev << 2
	int32_t rc_ids[MAX_ALT + 1], alt_ids[MAX_ALT], alt_pos_of[MAX_ALT]; uint8_t alt_rev_of[MAX_ALT];
};
// chunks of every read: plain reads (their bases are left to k_dna_plain) and empty scripts have one, for the header symbols
__global__ void k_walk_chunks(const uint32_t* __restrict__ es_ntup, const uint8_t* __restrict__ read_flag, uint32_t n, uint32_t chunk, uint32_t* __restrict__ out)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n) return;
	const uint32_t t = es_ntup[r] ? es_ntup[r] - 1 : 0;
	out[r] = read_flag[r] != 2 || t == 0 ? 1u : (t + chunk - 1) / chunk;
}
template<bool WRITE, typename K>
__global__ __launch_bounds__(64) void k_dna_walk(const FamTab* __restrict__ ftp, RefStore R, const uint8_t* __restrict__ es, const uint64_t* __restrict__ es_off,
                                                const uint32_t* __restrict__ es_ntup, const uint8_t* __restrict__ read_flag,
                                                uint32_t n_reads, uint32_t prev_types, uint32_t cur_read_id0, const uint64_t* __restrict__ chunk_off, WalkCk* __restrict__ cks, uint32_t n_chunks,
                                                uint32_t* __restrict__ counts, uint32_t* __restrict__ hdr_counts, const uint64_t* __restrict__ sym_off,
                                                K* __restrict__ key, uint32_t* __restrict__ err, uint32_t staged)
{
	__builtin_amdgcn_s_setprio(3);                                          // a launch of this kernel lasts as long as its slowest chain: its waves go first on their SIMDs
	__shared__ FamTab ft;
	constexpr uint32_t STAGE_STRIDE = Emitter<K>::SLOTS + 1;                     // (a lane's slots, an odd number of keys apart: the lanes' stores spread over the banks)
	__shared__ K s_stage[WRITE ? 64 * STAGE_STRIDE : 1];
	for (uint32_t i = threadIdx.x; i < sizeof(FamTab) / 4; i += blockDim.x) ((uint32_t*)&ft)[i] = ((const uint32_t*)ftp)[i];
	__syncthreads();
	const uint32_t CH = ft.walk_chunk, WARM = ft.walk_warm;
	uint32_t r, c, j = 0;                                                        // read, chunk, chunk index within the read
	if (WRITE)
	{
		c = blockIdx.x * WALK_LPW + threadIdx.x;
		if (c >= n_chunks) return;
		r = cks[c].read; j = c - (uint32_t)chunk_off[r];
	}
	else
	{
		r = blockIdx.x * WALK_LPW + threadIdx.x;
		if (r >= n_reads) return;
		c = (uint32_t)chunk_off[r];
	}
	const uint32_t n_ch = (uint32_t)(chunk_off[r + 1] - chunk_off[r]);
	if (!WRITE) for (uint32_t x = 0; x < n_ch; ++x) cks[c + x].read = r;
	bool track_all = WRITE;                                                      // symbol history kept throughout
restart:
	Emitter<K> em{ WRITE, key, WRITE ? sym_off[r] : 0, 0, &ft };
	if (WRITE && staged) em.stage = s_stage + threadIdx.x * STAGE_STRIDE;
	EsReader rd{ es + es_off[r], es + es_off[r + 1] };
	uint32_t type = T_NONE, v1 = 0, v2 = 0;
	rd.next(type, v1, v2);
	const uint32_t ntup = es_ntup[r];
	const uint32_t cur_read_id = cur_read_id0 + r;
	const uint64_t mask_tuple = (1ULL << (3 * ft.T)) - 1, mask_symbol = (1ULL << (2 * ft.S)) - 1;
	uint64_t ctx_tuple = mask_tuple, ctx_symbol = mask_symbol; uint32_t ctx_rev = 0xf;
	int32_t rc_ids[MAX_ALT + 1]; uint32_t n_rc = 0;                               // uo_rev_comp keys (values are not needed to encode)
	int32_t alt_ids[MAX_ALT], alt_pos_of[MAX_ALT]; uint8_t alt_rev_of[MAX_ALT]; uint32_t n_alt = 0;
	const uint32_t ref_id = v1; const bool ref_rev = v2 != 0;
	uint32_t alt_id = 0; bool alt_rev = false; int32_t alt_slot = -1;
	int64_t ref_pos = 0, alt_pos = 0; int32_t delta = 0;
	uint32_t last_type = T_NONE, last_flag = T_NONE; bool is_main = true, first = true;
	auto rev_comp_flag = [&](uint32_t id, bool rc) {                                 // :489-509
		for (uint32_t i = 0; i < n_rc; ++i) if (rc_ids[i] == (int32_t)id) return;
		em(F_REV_COMP, ctx_rev, rc ? 1u : 0u);
		if (n_rc < MAX_ALT + 1) rc_ids[n_rc++] = (int32_t)id;
		ctx_rev = ((ctx_rev << 2) + (rc ? 1u : 0u)) & 0xf;
	};
	RefCur mainc, altc;
	uint32_t t_idx = 0;                                                          // tuples of the script done
	if (j == 0)
	{
		// read-type history: types of the four previous reads, also across calls (dna_coder.cpp:440-463)
		uint32_t ctx_rt = 0;
		for (uint32_t t = 1; t <= 4; ++t)
		{
			uint32_t f = (r >= t) ? read_flag[r - t] : ((prev_types >> (2 * (t - 1 - r))) & 3u);
			ctx_rt |= f << (2 * (t - 1));
		}
		const uint32_t flag = type == T_START_PLAIN ? 0u : type == T_START_PLAIN_N ? 1u : 2u;
		em(F_READ_TYPE, ctx_rt, flag);
		emit_read_len(em, ntup - 1);
		if (type == T_START_PLAIN || type == T_START_PLAIN_N)
		{
			if (!WRITE) { hdr_counts[r] = em.count; counts[r] = em.count + (ntup - 1); }
			em.finish();
			return;
		}
		emit_read_id(em, ref_id, cur_read_id);
		rev_comp_flag(ref_id, ref_rev);
	}
	else
	{	// resume where the count pass was after j * CH tuples
		const WalkCk& k = cks[c];
		rd = EsReader{ es + es_off[r] + k.byte_off, es + es_off[r + 1] };
		em.count = k.sym;
		ctx_tuple = k.ctx_tuple; ctx_symbol = k.ctx_symbol; ref_pos = k.ref_pos; alt_pos = k.alt_pos;
		ctx_rev = k.ctx_rev; n_rc = k.n_rc; n_alt = k.n_alt; alt_id = k.alt_id; alt_slot = k.alt_slot; delta = k.delta; last_type = k.last_type; last_flag = k.last_flag;
		is_main = k.flags & 1; first = (k.flags >> 1) & 1; alt_rev = (k.flags >> 2) & 1;
		for (uint32_t i = 0; i < n_rc; ++i) rc_ids[i] = k.rc_ids[i];
		for (uint32_t i = 0; i < n_alt; ++i) { alt_ids[i] = k.alt_ids[i]; alt_pos_of[i] = k.alt_pos_of[i]; alt_rev_of[i] = k.alt_rev_of[i]; }
		if (alt_slot >= 0) altc.set(R, alt_id, alt_rev);
		t_idx = j * CH;
	}
	const uint32_t stop = WRITE && j + 1 < n_ch ? (j + 1) * CH : 0xffffffffu;
	const uint32_t s3 = 3 * ft.T;
	mainc.set(R, ref_id, ref_rev);
	uint32_t known = 0;                                                          // symbols in ctx_symbol since the history is kept
	for (;;)
	{
		if (!WRITE && t_idx && t_idx % CH == 0 && t_idx / CH < n_ch && rd.p < rd.e)
		{
			if (!track_all && known < (uint32_t)ft.S) { track_all = true; goto restart; }
			known = 0;
			WalkCk& k = cks[c + t_idx / CH];
			k.byte_off = (uint64_t)(rd.p - (es + es_off[r])); k.sym = em.count;
			k.ctx_tuple = ctx_tuple; k.ctx_symbol = ctx_symbol; k.ref_pos = ref_pos; k.alt_pos = alt_pos;
			k.ctx_rev = ctx_rev; k.n_rc = n_rc; k.n_alt = n_alt; k.alt_id = alt_id; k.alt_slot = alt_slot; k.delta = delta; k.last_type = last_type; k.last_flag = last_flag;
			k.flags = (is_main ? 1u : 0u) | (first ? 2u : 0u) | (alt_rev ? 4u : 0u);
			for (uint32_t i = 0; i < n_rc; ++i) k.rc_ids[i] = rc_ids[i];
			for (uint32_t i = 0; i < n_alt; ++i) { k.alt_ids[i] = alt_ids[i]; k.alt_pos_of[i] = alt_pos_of[i]; k.alt_rev_of[i] = alt_rev_of[i]; }
		}
		if (t_idx == stop) break;
		bool track = true;
		if (!WRITE)
		{
			const uint32_t q = t_idx % CH;
			const bool ck_ahead = t_idx / CH + 1 < n_ch;
			track = track_all || (ck_ahead && q >= CH - WARM);
			if (!track) known = 0;
			if (!track && (!ck_ahead || q + 8 <= CH - WARM) && rd.e - rd.p >= 8)
			{	// eight one-byte tuples (insertion, deletion, match, substitution: types 0..3 in the high nibble) at once
				if (rd.have < 8) rd.refill();
				const uint64_t w = rd.w0;
				if (!(w & 0xC0C0C0C0C0C0C0C0ull))
				{
					const uint64_t b4 = (w >> 4) & 0x0101010101010101ull, b5 = (w >> 5) & 0x0101010101010101ull;
					const uint32_t n_sub = (uint32_t)__popcll(b4 & b5), n_del = (uint32_t)__popcll(b4 & ~b5), n_mat = (uint32_t)__popcll(b5 & ~b4), n_ins = 8 - n_sub - n_del - n_mat;
					em.count += 8 + n_ins + n_sub;                                      // a tuple-type symbol each, a base for insertions and substitutions
					delta += (int32_t)n_ins - (int32_t)n_del;
					if (is_main) ref_pos += n_del + n_mat + n_sub; else alt_pos += n_del + n_mat + n_sub;
#pragma unroll
					for (int x = 0; x < 8; ++x) ctx_tuple = ((ctx_tuple << 3) + ((w >> (8 * x + 4)) & 3)) & mask_tuple;
					first = false; last_flag = last_type = (uint32_t)(w >> 60) & 3;
					rd.p += 8; rd.have -= 8; rd.w0 = rd.w1; rd.w1 = 0;
					t_idx += 8;
					continue;
				}
			}
		}
		if (!rd.next(type, v1, v2)) break;
		++t_idx;
		const uint32_t ref_symbol = !track ? 0u : is_main ? mainc.at(R, ref_pos) : altc.at(R, alt_pos);
		{	// encode_tuple_type (:651-710) with the guard case moved to its own dense region
			uint32_t cls = delta < -10 ? 1u : delta < -1 ? 2u : delta > 10 ? 3u : delta > 1 ? 4u : 0u;
			uint32_t c = (uint32_t)ctx_tuple | ((uint32_t)(ctx_symbol & 0xf) << s3);
			if (ref_symbol <= 3) c |= (ref_symbol << (s3 + 4)) | (cls << (s3 + 6));
			else c = (1u << (s3 + 9)) + (c | (cls << (s3 + 4)));
			int e1 = 15, e2 = 15;
			if (!first)
				switch (last_flag)
				{
				case T_MATCH: e1 = T_ANCHOR; break;
				case T_DEL: e1 = T_SKIP; break;
				case T_ANCHOR: e1 = T_ANCHOR; e2 = T_MATCH; break;
				case T_SKIP: e1 = T_DEL; e2 = T_SKIP; break;
				case T_MAIN_REF: case T_ALT_ID: e1 = T_ALT_ID; e2 = T_MAIN_REF; break;
				default: break;
				}
			em(F_TUPLE_TYPE, c, type, e1, e2);
			ctx_tuple = ((ctx_tuple << 3) + type) & mask_tuple;
		}
		first = false; last_flag = type;
		switch (type)
		{
		case T_ALT_ID:
		{
			if (!is_main && alt_slot >= 0) alt_pos_of[alt_slot] = (int32_t)alt_pos;
			bool is_new = false; int32_t slot = -1;
			if (n_alt == 0) { emit_read_id(em, v1, cur_read_id); alt_ids[0] = (int32_t)v1; alt_pos_of[0] = 0; n_alt = 1; slot = 0; is_new = true; }
			else
			{
				const uint32_t seen = n_alt;
				for (uint32_t i = 0; i < n_alt; ++i) if (alt_ids[i] == (int32_t)v1) slot = (int32_t)i;
				const int32_t short_id = slot;
				if (slot < 0)
				{
					if (n_alt >= MAX_ALT) { if (err) atomicOr(err, 1u); em.finish(); return; }
					slot = (int32_t)n_alt; alt_ids[n_alt] = (int32_t)v1; alt_pos_of[n_alt] = 0; ++n_alt; is_new = true;
				}
				em(F_SEEN, seen, short_id >= 0 ? 1u : 0u);
				if (short_id < 0) emit_read_id(em, v1, cur_read_id);
				else em(F_READ_ID_SHORT, seen, (uint32_t)short_id);
			}
			if (is_new) alt_rev_of[slot] = (uint8_t)v2;
			rev_comp_flag(v1, v2 != 0);
			alt_id = v1; alt_slot = slot; alt_rev = alt_rev_of[slot] != 0;
			altc.set(R, alt_id, alt_rev);
			alt_pos = 0; is_main = false; delta = 0;
			break;
		}
		case T_ANCHOR:
			emit_anchor_len(em, v2);
			if (is_main) ref_pos += v2; else alt_pos += v2;
			if (track)
			{
				for (int i = ft.S; i > 0; --i)
					ctx_symbol = (ctx_symbol << 2) + (is_main ? mainc.at(R, ref_pos - i) : altc.at(R, alt_pos - i));
				ctx_symbol &= mask_symbol;
				known = (uint32_t)ft.S;
			}
			delta = 0;
			break;
		case T_MATCH:
			ctx_symbol = ((ctx_symbol << 2) + ref_symbol) & mask_symbol; ++known;
			if (is_main) ++ref_pos; else ++alt_pos;
			break;
		case T_INS:
		{	// encode_insertion (:772-811)
			uint32_t shift = 2, c = 2;
			if (ft.level == 1) { c += (uint32_t)(ctx_symbol & 0xff) << shift; shift += 8; }
			else if (ft.level == 2) { c += (uint32_t)(ctx_symbol & 0x3ff) << shift; shift += 10; }
			else { c += (uint32_t)(ctx_symbol & 0x3ff) << shift; shift += 10; c += (uint32_t)(((ctx_symbol >> 10) & 3) == ((ctx_symbol >> 8) & 3)) << shift; ++shift; }
			const bool guard = ref_symbol > 3;
			c += (guard ? 0u : ref_symbol) << shift; shift += 2;
			c += (uint32_t)(ctx_tuple & 0777) << shift;
			if (guard) c += 1u << ft.sym_B;
			em(F_SYMBOLS, c, v1);
			ctx_symbol = ((ctx_symbol << 2) + v1) & mask_symbol; ++known;
			++delta;
			break;
		}
		case T_DEL:
			if (is_main) ++ref_pos; else ++alt_pos;
			--delta;
			break;
		case T_SUBST:
		{	// encode_substitution (:889-922); the coded symbol is the new base, the reference base is excluded
			const uint32_t rs = ref_symbol & 3;
			const uint32_t sym = (rs == 0) ? (v1 == 0 ? 1u : v1 == 1 ? 2u : 3u) : (rs == 1) ? (v1 == 0 ? 0u : v1 == 1 ? 2u : 3u)
			                   : (rs == 2) ? (v1 == 0 ? 0u : v1 == 1 ? 1u : 3u) : (v1 == 0 ? 0u : v1 == 1 ? 1u : 2u);   // subst_to_code (dna_coder.h:37)
			uint32_t shift = 2, c = 1;
			c += (uint32_t)(ctx_symbol & 0x3f) << shift; shift += 6;
			if (ft.level == 3) { c += (uint32_t)(((ctx_symbol >> 6) & 3) == ((ctx_symbol >> 4) & 3)) << shift; ++shift; }
			c += rs << shift; shift += 2;
			c += (uint32_t)(ctx_tuple & 07777) << shift;
			em(F_SYMBOLS, c, sym, (int)rs);
			ctx_symbol = ((ctx_symbol << 2) + sym) & mask_symbol; ++known;
			if (is_main) ++ref_pos; else ++alt_pos;
			break;
		}
		case T_SKIP:
		{
			const int32_t skip_len = (int32_t)v2;
			delta -= skip_len;
			if (!is_main && last_type == T_ALT_ID)                               // :187-201
			{
				const int32_t mod = skip_len - (alt_slot >= 0 ? alt_pos_of[alt_slot] : 0);
				if (mod > 0) emit_skip_len(em, (uint32_t)mod, false);
				else { emit_skip_len(em, 0, false); emit_skip_len(em, (uint32_t)(-mod), false); }
			}
			else emit_skip_len(em, (uint32_t)skip_len, last_type != T_ALT_ID && last_type != T_NONE);
			if (is_main) ref_pos += v2; else alt_pos += v2;
			break;
		}
		case T_MAIN_REF:
			is_main = true;
			if (alt_slot >= 0) alt_pos_of[alt_slot] = (int32_t)alt_pos;
			delta = 0;
			break;
		default: break;
		}
		last_type = type;
	}
	em.finish();
	if (!WRITE)
	{
		hdr_counts[r] = em.count; counts[r] = em.count;
		if (t_idx != ntup - 1 && err) atomicOr(err, 2u);                            // the chunks rely on the tuple counts
	}
}

// first tuple of every read -> read-type flag (0 plain, 1 plain with N, 2 edit script)
__global__ void k_read_flags(const uint8_t* __restrict__ es, const uint64_t* __restrict__ es_off, uint32_t r0, uint32_t r1, uint8_t* __restrict__ flag)
{
	uint32_t r = r0 + blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= r1) return;
	uint32_t t = es[es_off[r]] >> 4;
	flag[r - r0] = t == T_START_PLAIN ? 0 : t == T_START_PLAIN_N ? 1 : 2;
}

// triple slot of every symbol: one wave per read (all symbols of a read are in one part)
// (sym_off: symbol offsets of ALL reads; s0 = that of read r0, the first of the group the layout describes)
__global__ __launch_bounds__(256) void k_fill_sidx(const uint64_t* __restrict__ sym_off, uint64_t s0, uint32_t r0, uint32_t r1, TripLayoutDev lay, uint32_t* __restrict__ sidx)
{
	const uint32_t r = r0 + blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= r1) return;
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t a = sym_off[r] - s0, b = sym_off[r + 1] - s0;
	if (a == b) return;
	const uint32_t part = part_of_read(lay, r), pl = lay.rank ? lay.rank[part] : part;
	const uint64_t gb = lay.group_base[pl >> 6], p0 = lay.part_sym_start[part];
	for (uint64_t i = a + lane; i < b; i += 64) sidx[i] = (uint32_t)trip_slot(gb, pl & 63, i - p0);
}
// ---- D1b: bases of plain reads, one wave per read, one lane per base (dna_coder.cpp:1178-1227) ----------
template<typename K>
__global__ __launch_bounds__(256) void k_dna_plain(const FamTab* __restrict__ ftp, const uint8_t* __restrict__ es, const uint64_t* __restrict__ es_off,
                                                  const uint32_t* __restrict__ es_ntup, const uint8_t* __restrict__ read_flag, const uint32_t* __restrict__ hdr_counts,
                                                  const uint64_t* __restrict__ sym_off, uint32_t r0, uint32_t r1, K* __restrict__ key)
{
	const uint32_t r = r0 + blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= r1) return;
	const uint32_t fl = read_flag[r - r0];
	if (fl == 2) return;
	const FamTab& ft = *ftp;
	const uint32_t lane = threadIdx.x & 63;
	const uint8_t* b = es + es_off[r] + 1;                    // one byte per base after the start tuple (low nibble = base)
	const uint32_t len = es_ntup[r] - 1;
	const uint64_t off = sym_off[r - r0] + hdr_counts[r - r0];
	const uint32_t S2 = 2 * ft.S; const uint32_t mask = (1u << S2) - 1;
	const uint32_t fam = fl == 0 ? F_SYMBOLS : F_SYMBOLS_N;
	const uint32_t cbase = ft.ctx_base[fam];
	for (uint32_t i = lane; i < len; i += 64)
	{
		uint32_t ctx = mask;
		if (fl == 0)
		{
			const uint32_t n = i < (uint32_t)ft.S ? i : (uint32_t)ft.S;
			for (uint32_t t = n; t >= 1; --t) ctx = ((ctx << 2) + (b[i - t] & 3u)) & mask;
			ctx <<= 2;                                                         // plain marker 0
		}
		else
		{
			const uint32_t n = i < 4 ? i : 4;
			for (uint32_t t = n; t >= 1; --t) ctx = ((ctx << 4) + (b[i - t] & 0xfu)) & mask;
		}
		key[off + i] = KeyFmt<K>::make(cbase + ctx, b[i] & 0xfu);                 // (no exclusions: the payload is the symbol in both families)
	}
}

// ---- D3: runs of equal context in the sorted keys: dev_run_starts (scan.hip), one pass over the keys ---------------------------------

// ---- D4': LONG runs of a small-alphabet model (e.g. the tuple-type context "match after match" holds a third of all
// symbols).  The counters of a model between two rescales are its state at the last rescale + ADDER x (occurrences
// since), and `total` grows by ADDER per symbol whatever the symbol: the rescale instants form a short sequential
// chain (one every (MAX_TOTAL - total) / ADDER symbols) once class counts of arbitrary prefixes are available.  So:
//   k_long_hist    per 64-symbol step: class histogram -> prefix inside its 64-step group, group totals      (parallel)
//   k_long_groups  exclusive scan of the group totals of each run                                            (1 wave/run)
//   k_long_epochs  walk the rescale chain: epoch records {start, total, counters, prefix counts at start}    (1 wave/run)
//   k_long_apply   every symbol: counters = epoch state + ADDER x (prefix(j) - prefix(epoch start))          (parallel)
// Bit-identical to the sequential model (rc.h:233-244,316-358): same counters, same rescale instants.
constexpr uint32_t LONG_RUN = 1u << 19;                  // default of FamTab::long_run (COLORD_HIP_LONG_RUN overrides it, for tests)
struct LongRun { uint32_t s, e, fam, gctx; uint64_t step0, group0, epoch0; uint32_t epoch_cap, pad; };
template<int NS> struct EpochRec { uint32_t start, tot; uint32_t st[NS]; uint32_t cnt0[NS]; };       // NS = 8 or 32 classes
__device__ inline uint32_t find_run(const LongRun* runs, uint32_t n_runs, uint64_t step)
{
	uint32_t lo = 0, hi = n_runs;
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (runs[mid].step0 <= step) lo = mid; else hi = mid; }
	return lo;
}
template<typename K>
__global__ void k_long_find(const uint32_t* __restrict__ seg_start, uint32_t n_seg, const K* __restrict__ skey, const FamTab* __restrict__ ftp, LongRun* __restrict__ runs /* 2 lists of cap */, uint32_t cap, uint32_t* __restrict__ n_runs /* 2 */)
{
	const uint32_t sg = blockIdx.x * blockDim.x + threadIdx.x;
	if (sg >= n_seg) return;
	const uint32_t s = seg_start[sg], e = seg_start[sg + 1];
	const FamTab& ft = *ftp;
	if (e - s < ft.long_run) return;
	const uint32_t gctx = KeyFmt<K>::ctx(skey[s]);
	uint32_t fam = 0;
	for (uint32_t f = 1; f < N_FAM; ++f) if (gctx >= ft.ctx_base[f]) fam = f;
	if (ft.n_sym[fam] > 32) return;
	const uint32_t which = ft.n_sym[fam] > 8 ? 1u : 0u;
	const uint32_t i = atomicAdd(n_runs + which, 1u);
	if (i < cap) { LongRun r; r.s = s; r.e = e; r.fam = fam; r.gctx = gctx; r.step0 = r.group0 = r.epoch0 = 0; r.epoch_cap = 0; r.pad = 0; runs[which * cap + i] = r; }
}
// one wave per group of 64 steps (4096 symbols)
template<int NS, typename K>
__global__ __launch_bounds__(256) void k_long_hist(const LongRun* __restrict__ runs, uint32_t n_runs, uint64_t n_groups, const K* __restrict__ skey,
                                                  uint32_t* __restrict__ step_pfx, uint32_t* __restrict__ group_tot)
{
	const uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; const uint32_t lane = threadIdx.x & 63;
	if (g >= n_groups) return;
	uint32_t lo = 0, hi = n_runs;
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (runs[mid].group0 <= g) lo = mid; else hi = mid; }
	const LongRun R = runs[lo];
	const uint64_t gl = g - R.group0;                                          // group inside the run
	const uint32_t smask = key_sym_mask(R.fam);
	uint32_t c[NS];
#pragma unroll
	for (int a = 0; a < NS; ++a) c[a] = 0;
	for (uint32_t k = 0; k < 64; ++k)
	{
		const uint64_t j = (uint64_t)R.s + (gl * 64 + k) * 64 + lane;
		const bool valid = j < R.e;
		const uint32_t sym = valid ? (uint32_t)skey[j] & smask : 0xffu;
#pragma unroll
		for (uint32_t a = 0; a < NS; ++a) { const uint32_t n = (uint32_t)__popcll(__ballot(sym == a)); if (lane == k) c[a] = n; }
	}
	const uint64_t n_steps = ((uint64_t)(R.e - R.s) + 63) / 64;
	const uint64_t step = R.step0 + gl * 64 + lane;
#pragma unroll
	for (uint32_t a = 0; a < NS; ++a)
	{
		const uint32_t incl = wave_incl_scan(c[a]);
		if (gl * 64 + lane < n_steps) step_pfx[step * NS + a] = incl - c[a];    // the steps of the next run follow immediately
		if (lane == 63) group_tot[g * NS + a] = incl;
	}
}
template<int NS>
__global__ __launch_bounds__(64) void k_long_groups(const LongRun* __restrict__ runs, uint32_t n_runs, uint32_t* __restrict__ group_tot /* in: totals, out: exclusive prefix */)
{
	const uint32_t r = blockIdx.x, lane = threadIdx.x;
	if (r >= n_runs) return;
	const LongRun R = runs[r];
	const uint64_t n_steps = ((uint64_t)(R.e - R.s) + 63) / 64, n_groups = (n_steps + 63) / 64;
	uint32_t carry[NS];
#pragma unroll
	for (int a = 0; a < NS; ++a) carry[a] = 0;
	for (uint64_t g0 = 0; g0 < n_groups; g0 += 64)
	{
		const uint64_t g = g0 + lane; const bool valid = g < n_groups;
#pragma unroll
		for (uint32_t a = 0; a < NS; ++a)
		{
			const uint32_t v = valid ? group_tot[(R.group0 + g) * NS + a] : 0u;
			const uint32_t incl = wave_incl_scan(v);
			if (valid) group_tot[(R.group0 + g) * NS + a] = carry[a] + incl - v;
			carry[a] += __shfl(incl, 63);
		}
	}
}
// occurrences of each class among the first x symbols of the run: lane a (< NS) returns class a
template<int NS, typename K>
__device__ inline uint32_t long_prefix(const LongRun& R, const K* skey, const uint32_t* step_pfx, const uint32_t* group_pfx, uint32_t x, uint32_t lane)
{
	const uint32_t smask = key_sym_mask(R.fam);
	const uint64_t step = x >> 6; const uint32_t off = x & 63;
	const uint32_t a = lane & (NS - 1);
	uint32_t v = group_pfx[(R.group0 + (step >> 6)) * NS + a] + (step * 64 < (uint64_t)(R.e - R.s) || off ? step_pfx[(R.step0 + step) * NS + a] : 0u);
	if (x == R.e - R.s && off == 0)
	{	// exactly at the end on a step boundary: the last step's prefix + its own histogram = prefix of a virtual next step
		const uint64_t last = step - 1;
		const uint64_t j = (uint64_t)R.s + last * 64 + lane; const uint32_t sym = j < R.e ? (uint32_t)skey[j] & smask : 0xffu;
		uint32_t mine = 0;
#pragma unroll
		for (uint32_t b = 0; b < NS; ++b) { const uint32_t n = (uint32_t)__popcll(__ballot(sym == b)); if (a == b) mine = n; }
		return group_pfx[(R.group0 + (last >> 6)) * NS + a] + step_pfx[(R.step0 + last) * NS + a] + mine;
	}
	if (off)
	{
		const uint64_t j = (uint64_t)R.s + step * 64 + lane; const uint32_t sym = j < R.e ? (uint32_t)skey[j] & smask : 0xffu;
		const uint64_t below = (1ULL << off) - 1;
		uint32_t mine = 0;
#pragma unroll
		for (uint32_t b = 0; b < NS; ++b) { const uint32_t n = (uint32_t)__popcll(__ballot(sym == b) & below); if (a == b) mine = n; }
		v += mine;
	}
	return v;
}
template<int NS, typename K>
__global__ __launch_bounds__(64) void k_long_epochs(const FamTab* __restrict__ ftp, const LongRun* __restrict__ runs, uint32_t n_runs, const K* __restrict__ skey,
                                                   const uint32_t* __restrict__ step_pfx, const uint32_t* __restrict__ group_pfx, uint32_t* __restrict__ state,
                                                   EpochRec<NS>* __restrict__ epochs, uint32_t* __restrict__ n_epochs, uint32_t* __restrict__ group_epoch, uint32_t* __restrict__ err)
{
	const uint32_t r = blockIdx.x, lane = threadIdx.x;
	if (r >= n_runs) return;
	const LongRun R = runs[r];
	const FamTab& ft = *ftp;
	const uint32_t n_sym = ft.n_sym[R.fam], max_total = ft.max_total[R.fam], adder = ft.adder[R.fam];
	uint32_t* sp = state + ft.state_base[R.fam] + (uint64_t)(R.gctx - ft.ctx_base[R.fam]) * (n_sym + 1);
	const uint32_t L = R.e - R.s;
	uint32_t st = lane < n_sym ? sp[lane] : 0u, tot = sp[n_sym];               // lane a: counter of class a
	uint32_t cnt0 = 0, p = 0, ne = 0;
	EpochRec<NS>* E = epochs + R.epoch0;
	for (;;)
	{
		if (ne >= R.epoch_cap) { if (lane == 0) atomicOr(err, 8u); break; }
		if (lane < NS) { E[ne].st[lane] = st; E[ne].cnt0[lane] = cnt0; }
		if (lane == 0) { E[ne].start = p; E[ne].tot = tot; }
		++ne;
		const uint32_t rr = (max_total - tot + adder - 1) / adder;              // symbols until total reaches MAX_TOTAL
		if ((uint64_t)p + rr > L) break;
		p += rr;
		const uint32_t c1 = long_prefix<NS, K>(R, skey, step_pfx, group_pfx, p, lane);
		st += adder * (c1 - cnt0); cnt0 = c1;
		tot += adder * rr;
		while (tot >= max_total)
		{
			st = (st + 1) / 2;
			uint32_t sm = lane < NS ? st : 0u;
			for (int o = NS / 2; o; o >>= 1) sm += __shfl_xor(sm, o);
			tot = __shfl(sm, 0);
		}
		if (p == L) { if (lane < NS && ne < R.epoch_cap) { E[ne].st[lane] = st; E[ne].cnt0[lane] = cnt0; } if (lane == 0 && ne < R.epoch_cap) { E[ne].start = p; E[ne].tot = tot; } ++ne; break; }
	}
	// final state of the model
	const uint32_t cl = long_prefix<NS, K>(R, skey, step_pfx, group_pfx, L, lane);
	if (p < L) { st += adder * (cl - cnt0); tot += adder * (L - p); }
	if (lane < n_sym) sp[lane] = st;
	if (lane == 0) { sp[n_sym] = tot; n_epochs[r] = ne; }
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	// epoch in force at the first symbol of every group
	const uint64_t n_groups = (((uint64_t)L + 63) / 64 + 63) / 64;
	for (uint64_t g0 = 0; g0 < n_groups; g0 += 64)
	{
		const uint64_t g = g0 + lane;
		if (g >= n_groups) continue;
		const uint32_t pos = (uint32_t)(g * 4096);
		uint32_t lo = 0, hi = ne;
		while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (E[mid].start <= pos) lo = mid; else hi = mid; }
		group_epoch[R.group0 + g] = lo;
	}
}
// one wave per step
template<int NS, typename K>
__global__ __launch_bounds__(256) void k_long_apply(const LongRun* __restrict__ runs, uint32_t n_runs, uint64_t n_steps, const K* __restrict__ skey, const uint32_t* __restrict__ sval,
                                                   const uint32_t* __restrict__ step_pfx, const uint32_t* __restrict__ group_pfx, const EpochRec<NS>* __restrict__ epochs,
                                                   const uint32_t* __restrict__ n_epochs, const uint32_t* __restrict__ group_epoch, const FamTab* __restrict__ ftp, triple_t* __restrict__ trip)
{
	const uint64_t step = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; const uint32_t lane = threadIdx.x & 63;
	if (step >= n_steps) return;
	const uint32_t ri = find_run(runs, n_runs, step);
	const LongRun R = runs[ri];
	const uint32_t adder = ftp->adder[R.fam];
	const uint64_t sl = step - R.step0;                                        // step inside the run
	const uint32_t x0 = (uint32_t)(sl * 64), x = x0 + lane;
	const uint64_t j = (uint64_t)R.s + x; const bool valid = j < R.e;
	uint32_t sym = 0xffu, e1 = 15u, e2 = 15u;
	if (valid) key_decode(R.fam, KeyFmt<K>::payload(skey[j]), sym, e1, e2);
	uint64_t m[NS];
#pragma unroll
	for (uint32_t a = 0; a < NS; ++a) m[a] = __ballot(sym == a);
	// epoch of this lane's symbol: the one in force at the step's first symbol or a later one starting inside the step
	const EpochRec<NS>* E = epochs + R.epoch0; const uint32_t ne = n_epochs[ri];
	uint32_t e = group_epoch[R.group0 + (sl >> 6)];
	while (e + 1 < ne && E[e + 1].start <= x0) ++e;
	uint32_t me = e;
	while (me + 1 < ne && E[me + 1].start <= x) ++me;
	if (!valid) return;
	const EpochRec<NS>& ep = E[me];
	const uint64_t lt = (1ULL << lane) - 1;
	uint32_t cum = 0, freq = 0, excl = 0;
#pragma unroll
	for (uint32_t a = 0; a < NS; ++a)
	{
		const uint32_t pre = group_pfx[(R.group0 + (sl >> 6)) * NS + a] + step_pfx[step * NS + a] + (uint32_t)__popcll(m[a] & lt);
		const uint32_t v = ep.st[a] + adder * (pre - ep.cnt0[a]);
		const bool ex = NS <= 8 && (a == e1 || a == e2);                         // exclusions exist only in the small models (15 = none)
		if (ex) excl += v;
		if (a < sym && !ex) cum += v;
		if (a == sym) freq = v;
	}
	trip[sval[j]] = pack_triple(cum, freq, ep.tot + adder * (x - ep.start) - excl);
}

// ---- D4: model evolution, one wave per non-empty (family, context) run ---------------------------------
// alphabets <= 8: per-class ballots (with the two optional exclusions of rc.h:316-341); larger: LDS counters.
template<typename K>
__global__ __launch_bounds__(256) void k_dna_evolve(const FamTab* __restrict__ ftp, const K* __restrict__ skey, const uint32_t* __restrict__ sval,
                                                   const uint32_t* __restrict__ seg_start, uint32_t n_seg, uint32_t* __restrict__ state, triple_t* __restrict__ trip)
{
	__shared__ uint32_t s_cnt[4][256];
	__shared__ uint32_t s_pre[4][256];
	const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t sg = blockIdx.x * 4 + w;
	if (sg >= n_seg) return;
	const uint32_t s = seg_start[sg], e = seg_start[sg + 1];
	const FamTab& ft = *ftp;
	const uint32_t gctx = KeyFmt<K>::ctx(skey[s]);
	uint32_t fam = 0;
	for (uint32_t f = 1; f < N_FAM; ++f) if (gctx >= ft.ctx_base[f]) fam = f;
	const uint32_t n_sym = ft.n_sym[fam], max_total = ft.max_total[fam], adder = ft.adder[fam];
	uint32_t* sp = state + ft.state_base[fam] + (uint64_t)(gctx - ft.ctx_base[fam]) * (n_sym + 1);
	const uint64_t lt = (1ULL << lane) - 1;
	if (n_sym <= 8)
	{
		if (e - s >= ft.long_run) return;                                      // k_long_* below
		uint32_t st[8]; uint32_t tot = sp[n_sym];
#pragma unroll
		for (uint32_t a = 0; a < 8; ++a) st[a] = a < n_sym ? sp[a] : 0u;
		for (uint32_t j0 = s; j0 < e; j0 += 64)
		{
			const uint32_t j = j0 + lane; const bool valid = j < e;
			uint32_t sym = 0xffu, e1 = 15u, e2 = 15u;
			if (valid) key_decode(fam, KeyFmt<K>::payload(skey[j]), sym, e1, e2);
			const uint32_t dst = valid ? sval[j] : 0u;
			uint64_t m[8];
#pragma unroll
			for (uint32_t a = 0; a < 8; ++a) m[a] = __ballot(valid && sym == a);
			const uint32_t cnt = (e - j0) < 64 ? (e - j0) : 64;
			uint32_t start = 0;
			while (start < cnt)
			{
				const uint32_t r = (max_total - tot + adder - 1) / adder;
				const uint32_t now = (cnt - start) < r ? (cnt - start) : r;
				const uint64_t win = (now == 64 ? ~0ULL : ((1ULL << now) - 1)) << start;
				if (lane >= start && lane < start + now)
				{
					const uint64_t before = lt & win;
					uint32_t cum = 0, freq = 0, excl = 0;
#pragma unroll
					for (uint32_t a = 0; a < 8; ++a)
					{
						const uint32_t v = st[a] + adder * (uint32_t)__popcll(m[a] & before);
						const bool ex = a == e1 || a == e2;
						if (ex) excl += v;
						if (a < sym && !ex) cum += v;
						if (a == sym) freq = v;
					}
					trip[dst] = pack_triple(cum, freq, tot + adder * (lane - start) - excl);
				}
#pragma unroll
				for (uint32_t a = 0; a < 8; ++a) st[a] += adder * (uint32_t)__popcll(m[a] & win);
				tot += adder * now;
				while (tot >= max_total)
				{
					tot = 0;
#pragma unroll
					for (uint32_t a = 0; a < 8; ++a) { st[a] = (st[a] + 1) / 2; tot += st[a]; }   // unused classes stay 0
				}
				start += now;
			}
		}
		if (lane == 0)
		{
			for (uint32_t a = 0; a < n_sym; ++a) sp[a] = st[a];
			sp[n_sym] = tot;
		}
		return;
	}
	if (n_sym <= 32 && e - s >= ft.long_run) return;                          // k_long_*<32>
	uint32_t* cnt = s_cnt[w]; uint32_t* pre = s_pre[w];
#pragma unroll
	for (uint32_t t = 0; t < 4; ++t) { uint32_t a = lane * 4 + t; cnt[a] = a < n_sym ? sp[a] : 0u; }
	uint32_t tot = sp[n_sym];
	auto rebuild_prefix = [&]() {
		uint32_t v0 = cnt[lane * 4], v1 = cnt[lane * 4 + 1], v2 = cnt[lane * 4 + 2], v3 = cnt[lane * 4 + 3];
		uint32_t sum = v0 + v1 + v2 + v3;
		uint32_t ex = wave_incl_scan(sum) - sum;
		pre[lane * 4] = ex; pre[lane * 4 + 1] = ex + v0; pre[lane * 4 + 2] = ex + v0 + v1; pre[lane * 4 + 3] = ex + v0 + v1 + v2;
		__builtin_amdgcn_wave_barrier();
	};
	__builtin_amdgcn_wave_barrier();
	rebuild_prefix();
	for (uint32_t j0 = s; j0 < e; j0 += 64)
	{
		const uint32_t j = j0 + lane; const bool valid = j < e;
		const uint32_t sym = valid ? KeyFmt<K>::payload(skey[j]) : 0u;                // (the large alphabets code without exclusions: the payload is the symbol)
		const uint32_t dst = valid ? sval[j] : 0u;
		const uint32_t n_here = (e - j0) < 64 ? (e - j0) : 64;
		uint32_t start = 0;
		while (start < n_here)
		{
			const uint32_t r = (max_total - tot + adder - 1) / adder;
			const uint32_t now = (n_here - start) < r ? (n_here - start) : r;
			uint32_t less = 0, eq = 0;
			for (uint32_t l = start; l < start + now; ++l)
			{
				const uint32_t o = __builtin_amdgcn_readlane(sym, l);
				if (l < lane) { less += (o < sym) ? 1u : 0u; eq += (o == sym) ? 1u : 0u; }
			}
			const bool mine = lane >= start && lane < start + now;
			if (mine) trip[dst] = pack_triple(pre[sym] + adder * less, cnt[sym] + adder * eq, tot + adder * (lane - start));
			__builtin_amdgcn_wave_barrier();
			if (mine) atomicAdd(&cnt[sym], adder);
			__builtin_amdgcn_wave_barrier();
			tot += adder * now;
			while (tot >= max_total)
			{
				uint32_t sum = 0;
#pragma unroll
				for (uint32_t t = 0; t < 4; ++t) { uint32_t a = lane * 4 + t; if (a < n_sym) { uint32_t v = (cnt[a] + 1) / 2; cnt[a] = v; sum += v; } }
				tot = wave_sum(sum);
			}
			__builtin_amdgcn_wave_barrier();
			rebuild_prefix();
			start += now;
		}
	}
#pragma unroll
	for (uint32_t t = 0; t < 4; ++t) { uint32_t a = lane * 4 + t; if (a < n_sym) sp[a] = cnt[a]; }
	if (lane == 0) sp[n_sym] = tot;
}

__global__ void k_init_fam_state(const FamTab* __restrict__ ftp, uint32_t fam, uint32_t* __restrict__ state)
{
	const FamTab& ft = *ftp;
	const uint32_t ns = ft.n_sym[fam];
	const uint64_t n = (uint64_t)ft.n_ctx[fam] * (ns + 1);
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
		state[ft.state_base[fam] + i] = (i % (ns + 1)) == ns ? ns : 1u;
}
__global__ void k_gather_bytes2(const uint8_t* __restrict__ src, const uint64_t* __restrict__ src_off, const uint64_t* __restrict__ dst_off,
                                const uint64_t* __restrict__ size, uint8_t* __restrict__ dst)
{
	const uint32_t p = blockIdx.x;
	const uint64_t n = size[p]; const uint8_t* s = src + src_off[p]; uint8_t* d = dst + dst_off[p];
	for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
}
__global__ void k_last_types(const uint8_t* __restrict__ flag, uint32_t n, uint32_t prev, uint32_t* __restrict__ out)
{
	// types of the last four reads, most recent in the low bits (as ctx_read_type keeps them)
	uint32_t v = prev;
	uint32_t start = n > 4 ? n - 4 : 0;
	for (uint32_t i = start; i < n; ++i) v = ((v << 2) + flag[i]) & 0xff;
	*out = v;
}
} // namespace

// The half of cl_dna_encode that needs no model state: the tuple walks of a batch of reads -> read flags, symbol offsets and the
// (family, context, symbol) key of every coded symbol.  It depends on the tuple streams, the reference reads and two scalars
// carried from the batch before (types of its last four reads, its read count) — not on the adaptive models — so it can be
// done for the NEXT batch while the interval coding of the current one, a dependent chain per part, drains (cl_dna_walk_ahead).
// ... and, when the part bounds are known ahead as well, the other model-independent half of a group of parts: its place in the
// interleaved triple layout, the triple slot of every symbol, the stable sort by (family, context) and the context runs.
struct DnaGroupPrep {
	uint32_t p0 = 0, p1 = 0, r0 = 0, r1 = 0; uint64_t s0 = 0, n_syms = 0, n_seg = 0, trip_words = 0;
	std::vector<uint32_t> rank, plen_r;                                           // part -> place (descending length); lengths by place
	DevBuf<uint64_t> d_gbase; DevBuf<uint32_t> d_plen;
	DevBuf<uint32_t> sidx, seg;
	// the model-independent half of the long context runs (round 6: made with the rest of the preparation — rounds 3-5 found the runs, sorted
	// them on the host and took their class histograms between the two model kernels of the coding thread, two host round trips and
	// k_long_hist on the chain every chunk waits for): per alphabet class (<= 8 / <= 32 symbols) the runs in key order with their places in the
	// step / group / epoch arrays, and the class prefixes per 64-symbol step and per 64-step group
	struct LongPrep { uint32_t n_runs = 0; uint64_t steps = 0, groups = 0, eps = 0; DevBuf<LongRun> runs; DevBuf<uint32_t> step_pfx, group_pfx; } lp[2];
	bool long_ready = false;
};
struct DnaWalked {
	const uint8_t* d_es = nullptr; uint32_t n_reads = 0;                       // identity of the batch
	uint32_t prev_types_in = 0, cur_read_id_in = 0, prev_types_out = 0;
	DevBuf<uint8_t> rflag; DevBuf<uint32_t> hdr; DevBuf<uint64_t> sym_off, key; DevBuf<uint32_t> key32; bool narrow = false;   // the keys: 32-bit words where context + payload fit (level 1), else 64-bit
	std::vector<uint64_t> h_sym_off;
	bool presorted = false; std::vector<uint32_t> part_bounds; std::vector<std::unique_ptr<DnaGroupPrep>> groups;   // (keys sorted group by group, in place)
};
struct cl_dna_coder {
	cl_ctx* ctx = nullptr;
	FamTab ft;
	DevBuf<FamTab> d_ft;
	DevBuf<uint32_t> state;
	uint32_t ctx_bits = 0; bool narrow = false;   // bits of the dense context ids; 32-bit keys (context << 8 | payload fits)
	uint32_t cur_read_id = 0;        // CDNACoder::cur_read_id
	uint32_t prev_types = 0;         // ctx_read_type (types of the last four reads)
	uint32_t next_read_id = 0, next_prev_types = 0; bool next_valid = false;   // the same after the batch being coded (known once it is walked)
	std::unique_ptr<DnaWalked> ahead;                                          // the next batch, walked ahead
	std::function<cl_status()> before_tail;                                    // called by cl_dna_encode before it waits for its last interval coding
	std::deque<std::unique_ptr<struct DnaEvolved>> evolved;                    // the next batches, their models evolved and their interval coders running (cl_dna_evolve_ahead)
	uint32_t ahead_prev_types = 0, ahead_read_id = 0;                          // the walk scalars after the last batch evolved ahead
	std::vector<hipStream_t> cstreams; uint32_t next_cstream = 0;              // streams of the interval coders: those of consecutive batches run side by side
	~cl_dna_coder();
};

// CDNACoder::Init(true, max_no_alt_refs, level, ., start_read_id) (dna_coder.cpp:1242-1340)
extern "C" cl_status cl_dna_coder_create(cl_ctx* ctx, uint32_t max_alt_refs, int32_t level, uint32_t start_read_id, cl_dna_coder** out)
{
	if (!ctx || !out) return cl_fail(ctx, CL_E_INVALID, "cl_dna_coder_create: null argument");
	if (level < 1 || level > 3 || max_alt_refs < 1 || max_alt_refs > MAX_ALT) return cl_fail(ctx, CL_E_INVALID, "cl_dna_coder_create: level 1..3, 1 <= max_alt_refs <= 64");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	cl_dna_coder* D = new cl_dna_coder(); D->ctx = ctx; D->cur_read_id = start_read_id;
	std::unique_ptr<cl_dna_coder> guard(D);
	FamTab& f = D->ft; memset(&f, 0, sizeof(f));
	f.level = level; f.max_alt = max_alt_refs;
	f.T = level == 3 ? 4 : level == 2 ? 3 : 2;
	f.S = level == 3 ? 8 : level == 2 ? 7 : 5;
	f.sym_B = level == 3 ? 24 : level == 2 ? 23 : 22;
	f.long_run = LONG_RUN;
	if (const char* lr = getenv("COLORD_HIP_LONG_RUN")) f.long_run = (uint32_t)std::max(64, atoi(lr));
	f.walk_chunk = WALK_CHUNK; f.walk_warm = WALK_WARM;
	if (const char* v = getenv("COLORD_HIP_WALK_CHUNK")) f.walk_chunk = (uint32_t)std::min(1 << 20, std::max(32, atoi(v)));
	if (const char* v = getenv("COLORD_HIP_WALK_WARM")) f.walk_warm = (uint32_t)std::max(0, atoi(v));
	f.walk_warm = std::min(f.walk_warm, f.walk_chunk / 2);
	auto set = [&](int i, uint32_t ns, uint32_t mt, uint32_t ad, uint32_t nc) { f.n_sym[i] = ns; f.max_total[i] = mt; f.adder[i] = ad; f.n_ctx[i] = nc; };
	set(F_READ_TYPE, 3, 1u << 15, 1, 256);                  // dna_coder.h:48-60
	set(F_REV_COMP, 2, 1u << 15, 1, 16);
	set(F_SEEN, 2, 1u << 15, 1, MAX_ALT + 1);
	set(F_LEN_BITS, 32, 1u << 18, 8, 1);
	set(F_LEN_DATA, 256, 1u << 18, 8, 512);
	set(F_SYMBOLS, 4, 1u << 10, 1, 1u << (f.sym_B + 1));
	set(F_SYMBOLS_N, 5, 1u << 10, 1, 1u << 16);
	set(F_READ_ID, 256, 1u << 13, 1, 2048);
	set(F_READ_ID_SHORT, max_alt_refs, 1u << 13, 1, MAX_ALT + 1);
	set(F_ANCHOR_LEN, 24, 1u << 15, 1, 1u << 17);
	set(F_SKIP_LOCAL, 256, 1u << 15, 1, 1u << 14);
	set(F_SKIP_DISTANT, 256, 1u << 15, 1, 256);
	set(F_TUPLE_TYPE, 8, 1u << 15, 1, 1u << (3 * f.T + 10));
	uint64_t cb = 0, sb = 0;
	for (int i = 0; i < N_FAM; ++i) { f.ctx_base[i] = (uint32_t)cb; f.state_base[i] = sb; cb += f.n_ctx[i]; sb += (uint64_t)f.n_ctx[i] * (f.n_sym[i] + 1); }
	f.ctx_base[N_FAM] = (uint32_t)cb; f.state_base[N_FAM] = sb;
	if (cb >= (1ull << 32)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_dna_coder_create: context space too large");
	D->ctx_bits = 1; while ((1ull << D->ctx_bits) < cb) ++D->ctx_bits;
	D->narrow = D->ctx_bits + 8 <= 32 && !getenv("COLORD_HIP_DNA_WIDE_KEYS");     // (the knob: 64-bit keys at level 1 too, for A/B runs and for the tests of that form)
	DEV_ALLOC(ctx, D->d_ft, 1);
	HIP_TRY(ctx, hipMemcpyAsync(D->d_ft.p, &f, sizeof(f), hipMemcpyHostToDevice, ctx->stream));
	DEV_ALLOC(ctx, D->state, sb);
	for (uint32_t i = 0; i < N_FAM; ++i) LAUNCH(ctx, k_init_fam_state, 2048, 256, (const FamTab*)D->d_ft.p, i, D->state.p);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	*out = guard.release();
	return CL_OK;
}
extern "C" void cl_dna_coder_free(cl_dna_coder* d) { delete d; }

// The interval coding of a group is a set of dependent chains (one lane per part, rc_dev.hpp) that leaves the machine
// idle: it runs on the context's side stream while the main stream sorts and evolves the models of the NEXT group.
namespace {
struct SideSync { hipStream_t s = nullptr; ~SideSync() { if (s) (void)hipStreamSynchronize(s); } };
struct PendingGroup {
	DevBuf<triple_t> trip; DevBuf<uint64_t> d_gbase, d_out_off, d_size, d_dst_off; DevBuf<uint32_t> d_plen; DevBuf<uint8_t> tmp;
	std::vector<uint64_t> out_off; std::vector<uint32_t> rank; uint32_t p0 = 0, np = 0;   // rank: part -> place (descending length)
	hipStream_t stream = nullptr;                   // where its interval coder runs
	SideSync sync;                                  // destroyed first: nothing above is released while the side stream runs
};
} // namespace
// A batch after the half of cl_dna_encode that touches the models: every group's triples are made and its interval coder is
// running on a stream of the coder.  The interval arithmetic changes no model, so the NEXT batch can be taken this far while the
// coders of this one — one dependent chain per part, 1.3 s for the reference's parts of 4 Mi symbols — are still at work.
struct DnaEvolved {
	const uint8_t* d_es = nullptr; uint32_t n_reads = 0, prev_types_in = 0, read_id_in = 0, prev_types_out = 0;
	std::vector<uint32_t> part_bounds;
	std::vector<std::unique_ptr<PendingGroup>> groups;
};
cl_dna_coder::~cl_dna_coder() { evolved.clear(); for (hipStream_t s : cstreams) (void)hipStreamDestroy(s); }

namespace {
// D1 for ALL reads of a batch at once: the walks are one lane per read and as long as the longest read's chain of tuples takes,
// whatever the number of reads — so one count pass and one write pass per batch, not per group.
cl_status dna_walk(cl_ctx* ctx, cl_dna_coder* D, const cl_reads* refs, const uint8_t* d_es, const uint64_t* d_es_off, const uint32_t* d_es_ntuples, uint32_t n_reads,
                   uint32_t prev_types, uint32_t cur_read_id, DnaWalked& W)
{
	const FamTab& f = D->ft;
	RefStore R{ refs->packed.p, refs->word_off.p, refs->lens.p, refs->n_reads };
	W.d_es = d_es; W.n_reads = n_reads; W.prev_types_in = prev_types; W.cur_read_id_in = cur_read_id;
	DevBuf<uint8_t>& rflag = W.rflag; DevBuf<uint32_t>& hdr = W.hdr; DevBuf<uint64_t>& sym_off = W.sym_off; std::vector<uint64_t>& h_sym_off = W.h_sym_off;
	W.narrow = D->narrow;
	DEV_ALLOC(ctx, rflag, (uint64_t)n_reads + 1); DEV_ALLOC(ctx, hdr, (uint64_t)n_reads + 1); DEV_ALLOC(ctx, sym_off, (uint64_t)n_reads + 1);
	DevBuf<uint32_t> err; DEV_ALLOC(ctx, err, 1);
	HIP_TRY(ctx, hipMemsetAsync(err.p, 0, 4, ctx->stream));
	h_sym_off.assign((size_t)n_reads + 1, 0);
	if (n_reads)
	{
		uint64_t total_syms = 0, h_eb[2] = { 0, 0 };
		HIP_TRY(ctx, hipMemcpyAsync(&h_eb[0], d_es_off, 8, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(&h_eb[1], d_es_off + n_reads, 8, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		DevBuf<uint64_t> chunk_off; DEV_ALLOC(ctx, chunk_off, (uint64_t)n_reads + 1);
		DevBuf<WalkCk> cks; uint64_t n_chunks = 0;
		{
			DevBuf<uint32_t> counts; DEV_ALLOC(ctx, counts, n_reads);
			LAUNCH(ctx, k_read_flags, grid_for(n_reads, 256), 256, d_es, d_es_off, 0u, n_reads, rflag.p);
			LAUNCH(ctx, k_walk_chunks, grid_for(n_reads, 256), 256, d_es_ntuples, (const uint8_t*)rflag.p, n_reads, f.walk_chunk, counts.p);
			CL_TRY(dev_exclusive_scan_u64(ctx, counts.p, chunk_off.p, n_reads, &n_chunks));
			if (n_chunks >= (1ull << 32)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_dna_encode: too many tuples in one call");
			DEV_ALLOC(ctx, cks, n_chunks);
			LAUNCHB_NAMED(ctx, "k_dna_walk<false>", (double)(h_eb[1] - h_eb[0]) + 8.0 * n_reads + (double)n_chunks * sizeof(WalkCk), (k_dna_walk<false, uint32_t>), grid_for(n_reads, WALK_LPW), 64, /* tuple bytes in, one count and the chunk states out */
				(const FamTab*)D->d_ft.p, R, d_es, d_es_off, d_es_ntuples, (const uint8_t*)rflag.p, n_reads,
				prev_types, cur_read_id, (const uint64_t*)chunk_off.p, cks.p, (uint32_t)n_chunks, counts.p, hdr.p, (const uint64_t*)nullptr, (uint32_t*)nullptr, err.p, 0u);
			HIP_TRY(ctx, hipGetLastError());
			CL_TRY(dev_exclusive_scan_u64(ctx, counts.p, sym_off.p, n_reads, &total_syms));
		}
		HIP_TRY(ctx, hipMemcpy(h_sym_off.data(), sym_off.p, ((uint64_t)n_reads + 1) * 8, hipMemcpyDeviceToHost));
		uint32_t herr0 = 0;
		HIP_TRY(ctx, hipMemcpy(&herr0, err.p, 4, hipMemcpyDeviceToHost));
		if (herr0 & 2) return cl_fail(ctx, CL_E_INVALID, "cl_dna_encode: the tuple count of a read does not match its stream");
		if (herr0) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_dna_encode: a read uses more than 64 alternative references");
		// tuple bytes in (<= 1 per symbol), one key per symbol out
		if (W.narrow)
		{
			DEV_ALLOC(ctx, W.key32, total_syms);
			LAUNCHB_NAMED(ctx, "k_dna_walk<true>", total_syms * 5.0, (k_dna_walk<true, uint32_t>), grid_for(n_chunks, WALK_LPW), 64,
				(const FamTab*)D->d_ft.p, R, d_es, d_es_off, d_es_ntuples, (const uint8_t*)rflag.p, n_reads,
				prev_types, cur_read_id, (const uint64_t*)chunk_off.p, cks.p, (uint32_t)n_chunks, (uint32_t*)nullptr, (uint32_t*)nullptr, (const uint64_t*)sym_off.p, W.key32.p, err.p, 1u);
			LAUNCH_NAMED(ctx, "k_dna_plain", (k_dna_plain<uint32_t>), grid_for(n_reads, 4), 256, (const FamTab*)D->d_ft.p, d_es, d_es_off, d_es_ntuples, (const uint8_t*)rflag.p, (const uint32_t*)hdr.p,
				(const uint64_t*)sym_off.p, 0u, n_reads, W.key32.p);
		}
		else
		{
			DEV_ALLOC(ctx, W.key, total_syms);
			LAUNCHB_NAMED(ctx, "k_dna_walk<true>", total_syms * 9.0, (k_dna_walk<true, uint64_t>), grid_for(n_chunks, WALK_LPW), 64,
				(const FamTab*)D->d_ft.p, R, d_es, d_es_off, d_es_ntuples, (const uint8_t*)rflag.p, n_reads,
				prev_types, cur_read_id, (const uint64_t*)chunk_off.p, cks.p, (uint32_t)n_chunks, (uint32_t*)nullptr, (uint32_t*)nullptr, (const uint64_t*)sym_off.p, W.key.p, err.p, 1u);
			LAUNCH_NAMED(ctx, "k_dna_plain", (k_dna_plain<uint64_t>), grid_for(n_reads, 4), 256, (const FamTab*)D->d_ft.p, d_es, d_es_off, d_es_ntuples, (const uint8_t*)rflag.p, (const uint32_t*)hdr.p,
				(const uint64_t*)sym_off.p, 0u, n_reads, W.key.p);
		}
		HIP_TRY(ctx, hipGetLastError());
	}
	{	// the types of the last four reads: what the batch after this one starts from
		DevBuf<uint32_t> lt; DEV_ALLOC(ctx, lt, 1);
		LAUNCH(ctx, k_last_types, 1, 1, (const uint8_t*)rflag.p, n_reads, prev_types, lt.p);
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipMemcpyAsync(&W.prev_types_out, lt.p, 4, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	return CL_OK;
}
} // namespace

namespace {
// The model-independent half of the long runs of a sorted group: which context runs are long (k_long_find), their order and places (host:
// a few hundred records), class histograms per step and group (k_long_hist, k_long_groups).  On the context of whoever prepares the group.
template<typename K>
cl_status dna_long_prepare(cl_ctx* ctx, cl_dna_coder* D, const K* gkey, uint64_t n_syms, DnaGroupPrep& G)
{
	const FamTab& f = D->ft;
	const uint32_t RUN_CAP = (uint32_t)std::max<uint64_t>(4096, n_syms / f.long_run + 2);   // (a long run holds at least long_run symbols: never more runs than this)
	DevBuf<LongRun> runs; DEV_ALLOC(ctx, runs, 2 * RUN_CAP);
	DevBuf<uint32_t> n_runs; DEV_ALLOC(ctx, n_runs, 2);
	HIP_TRY(ctx, hipMemsetAsync(n_runs.p, 0, 8, ctx->stream));
	LAUNCH_NAMED(ctx, "k_long_find", (k_long_find<K>), grid_for(G.n_seg, 256), 256, (const uint32_t*)G.seg.p, (uint32_t)G.n_seg, gkey, (const FamTab*)D->d_ft.p, runs.p, RUN_CAP, n_runs.p);
	uint32_t nlr2[2] = { 0, 0 };
	HIP_TRY(ctx, hipMemcpyAsync(nlr2, n_runs.p, 8, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	if (nlr2[0] > RUN_CAP || nlr2[1] > RUN_CAP) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_dna_encode: more than 4096 long context runs in one group");
	for (int which = 0; which < 2; ++which)
	{
		DnaGroupPrep::LongPrep& L = G.lp[which];
		const uint32_t nlr = nlr2[which], NS = which ? 32 : 8;
		L.n_runs = nlr; L.steps = L.groups = L.eps = 0;
		if (!nlr) continue;
		std::vector<LongRun> h(nlr);
		HIP_TRY(ctx, hipMemcpy(h.data(), runs.p + (uint64_t)which * RUN_CAP, nlr * sizeof(LongRun), hipMemcpyDeviceToHost));
		std::sort(h.begin(), h.end(), [](const LongRun& a, const LongRun& b) { return a.s < b.s; });
		for (auto& r : h)
		{
			const uint64_t Ln = r.e - r.s, ns = (Ln + 63) / 64, ngp = (ns + 63) / 64;
			const uint64_t half = f.max_total[r.fam] / 2 / f.adder[r.fam];
			r.step0 = L.steps; r.group0 = L.groups; r.epoch0 = L.eps; r.epoch_cap = (uint32_t)(Ln / (half > 80 ? half - 40 : 1) + 8);
			L.steps += ns; L.groups += ngp; L.eps += r.epoch_cap;
		}
		DEV_ALLOC(ctx, L.runs, nlr);
		HIP_TRY(ctx, hipMemcpyAsync(L.runs.p, h.data(), nlr * sizeof(LongRun), hipMemcpyHostToDevice, ctx->stream));
		DEV_ALLOC(ctx, L.step_pfx, (L.steps + 1) * NS); DEV_ALLOC(ctx, L.group_pfx, (L.groups + 1) * NS);
		const LongRun* cr = L.runs.p;
		if (which == 0)
		{
			LAUNCHB_NAMED(ctx, "k_long_hist<8>", L.steps * 64.0 * sizeof(K), (k_long_hist<8, K>), grid_for(L.groups * 64, 256), 256, cr, nlr, L.groups, gkey, L.step_pfx.p, L.group_pfx.p);
			LAUNCH(ctx, (k_long_groups<8>), nlr, 64, cr, nlr, L.group_pfx.p);
		}
		else
		{
			LAUNCHB_NAMED(ctx, "k_long_hist<32>", L.steps * 64.0 * sizeof(K), (k_long_hist<32, K>), grid_for(L.groups * 64, 256), 256, cr, nlr, L.groups, gkey, L.step_pfx.p, L.group_pfx.p);
			LAUNCH(ctx, (k_long_groups<32>), nlr, 64, cr, nlr, L.group_pfx.p);
		}
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                           // (the upload above reads `h`)
	}
	G.long_ready = true;
	return CL_OK;
}
} // namespace

namespace {
// The model-independent half of a group of parts starting at part p0 (as many parts as the 31-bit symbol indices and the memory
// for the triples allow): layout, triple slots, stable sort of (key, slot) by (family, context), context runs.
cl_status dna_group_prepare(cl_ctx* ctx, cl_dna_coder* D, DnaWalked& W, uint32_t p0, const uint32_t* h_part_bounds, uint32_t n_parts, DnaGroupPrep& G)
{
	const std::vector<uint64_t>& h_sym_off = W.h_sym_off;
	const uint64_t GROUP_SYMS = 5ull << 28;
	uint32_t p1 = p0; const uint32_t r0 = h_part_bounds[p0];
	while (p1 < n_parts)
	{
		const uint64_t pl = h_sym_off[h_part_bounds[p1 + 1]] - h_sym_off[h_part_bounds[p1]];
		if (pl >= (1ull << 31) - 64) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_dna_encode: one part has >= 2^31 symbols");
		if (p1 > p0 && h_sym_off[h_part_bounds[p1 + 1]] - h_sym_off[r0] >= GROUP_SYMS) break;
		++p1;
	}
	const uint32_t r1 = h_part_bounds[p1], nr = r1 - r0, np = p1 - p0, ng = (np + 63) / 64;
	const uint64_t s0 = h_sym_off[r0], n_syms = h_sym_off[r1] - s0;
	G.p0 = p0; G.p1 = p1; G.r0 = r0; G.r1 = r1; G.s0 = s0; G.n_syms = n_syms;
	// part geometry
	std::vector<uint64_t> sym_start(np + 1);
	std::vector<uint32_t> plen(np), pfirst(np + 1);
	for (uint32_t p = 0; p <= np; ++p) { pfirst[p] = h_part_bounds[p0 + p]; sym_start[p] = h_sym_off[pfirst[p]] - s0; }
	for (uint32_t p = 0; p < np; ++p) plen[p] = (uint32_t)(sym_start[p + 1] - sym_start[p]);
	// places in the interleaved layout by descending length: the 64 parts of a wave of the interval coder are alike, and no
	// slots are wasted on the longest part of a group (parts hold whole reads: 65 k to 265 k symbols in one group otherwise)
	std::vector<uint32_t> order(np); std::vector<uint32_t>& rank = G.rank; std::vector<uint32_t>& plen_r = G.plen_r;
	rank.resize(np); plen_r.resize(np);
	for (uint32_t p = 0; p < np; ++p) order[p] = p;
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return plen[a] > plen[b]; });
	for (uint32_t i = 0; i < np; ++i) { rank[order[i]] = i; plen_r[i] = plen[order[i]]; }
	std::vector<uint64_t> gbase(ng + 1, 0);
	for (uint32_t g = 0; g < ng; ++g) gbase[g + 1] = gbase[g] + trip_group_words(plen_r[g * 64]);
	if (gbase[ng] >= (1ull << 32)) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_dna_encode: group too large for 32-bit triple indices");
	G.trip_words = gbase[ng];
	DevBuf<uint64_t> d_sym_start; DevBuf<uint32_t> d_pfirst, d_rank;
	DEV_ALLOC(ctx, d_rank, np);
	HIP_TRY(ctx, hipMemcpyAsync(d_rank.p, rank.data(), np * 4, hipMemcpyHostToDevice, ctx->stream));
	DEV_ALLOC(ctx, d_sym_start, np + 1); DEV_ALLOC(ctx, G.d_gbase, ng + 1); DEV_ALLOC(ctx, G.d_plen, np); DEV_ALLOC(ctx, d_pfirst, np + 1);
	HIP_TRY(ctx, hipMemcpyAsync(d_sym_start.p, sym_start.data(), (np + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(G.d_gbase.p, gbase.data(), (ng + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(G.d_plen.p, plen_r.data(), np * 4, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(d_pfirst.p, pfirst.data(), (np + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
	TripLayoutDev lay{ d_pfirst.p, d_sym_start.p, G.d_gbase.p, np, d_rank.p };
	if (n_syms)
	{
		DEV_ALLOC(ctx, G.sidx, n_syms);
		LAUNCHB(ctx, n_syms * 4.0, k_fill_sidx, grid_for(nr, 4), 256, (const uint64_t*)W.sym_off.p, s0, r0, r1, lay, G.sidx.p);
		HIP_TRY(ctx, hipGetLastError());
		const uint32_t cbits = D->ctx_bits;
		// this group's keys (written by the walk) sorted in place — one group = the whole batch is the usual case: then the buffers are swapped
		// with the sort's —, then the starts of the context runs (at most one per context in use: the bound of the run list)
		const uint64_t seg_cap = std::min<uint64_t>(n_syms, D->ft.ctx_base[N_FAM]) + 1;
		DEV_ALLOC(ctx, G.seg, seg_cap);
		if (W.narrow)
		{
			if (s0 == 0 && W.key32.n == n_syms) CL_TRY(dev_sort_keys32_pairs_swap(ctx, W.key32, G.sidx, n_syms, 8, 8 + cbits));
			else CL_TRY(dev_sort_keys32_pairs(ctx, W.key32.p + s0, G.sidx.p, n_syms, 8, 8 + cbits));
			CL_TRY(dev_run_starts_u32(ctx, W.key32.p + s0, n_syms, 8, G.seg.p, seg_cap, &G.n_seg));
			CL_TRY(dna_long_prepare<uint32_t>(ctx, D, W.key32.p + s0, n_syms, G));
		}
		else
		{
			if (s0 == 0 && W.key.n == n_syms) CL_TRY(dev_sort_pairs_swap(ctx, W.key, G.sidx, n_syms, 8, 8 + cbits));
			else CL_TRY(dev_sort_pairs(ctx, W.key.p + s0, G.sidx.p, n_syms, 8, 8 + cbits));
			CL_TRY(dev_run_starts_u64(ctx, W.key.p + s0, n_syms, 8, G.seg.p, seg_cap, &G.n_seg));
			CL_TRY(dna_long_prepare<uint64_t>(ctx, D, W.key.p + s0, n_syms, G));
		}
	}
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                             // (the uploads above read host vectors of this frame)
	return CL_OK;
}
} // namespace

// Internal (stream.hip): everything of a batch that needs no model state — the walks and, with part bounds, the sorted groups — on
// ANY context (the compressor runs it on a context and thread of its own beside the coding of the batch before).  The two scalars
// the walk starts from chain batch to batch: the caller keeps them (cl_dna_coder_state for the first batch).
cl_status cl_dna_prepare_batch(cl_ctx* ctx, cl_dna_coder* D, const cl_reads* refs, const uint8_t* d_es, const uint64_t* d_es_off, const uint32_t* d_es_ntuples, uint32_t n_reads,
                               uint32_t prev_types, uint32_t read_id, const uint32_t* h_part_bounds, uint32_t n_parts, DnaWalked** out, uint32_t* prev_types_out)
{
	if (!ctx || !D || !refs || !d_es || !d_es_off || !d_es_ntuples || !n_reads || !out) return CL_E_INVALID;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	auto W = std::make_unique<DnaWalked>();
	CL_TRY(dna_walk(ctx, D, refs, d_es, d_es_off, d_es_ntuples, n_reads, prev_types, read_id, *W));
	if (h_part_bounds && n_parts && h_part_bounds[0] == 0 && h_part_bounds[n_parts] == n_reads)
	{
		W->part_bounds.assign(h_part_bounds, h_part_bounds + n_parts + 1);
		for (uint32_t p0 = 0; p0 < n_parts; )
		{
			auto G = std::make_unique<DnaGroupPrep>();
			CL_TRY(dna_group_prepare(ctx, D, *W, p0, h_part_bounds, n_parts, *G));
			p0 = G->p1;
			W->groups.push_back(std::move(G));
		}
		W->presorted = true;
	}
	if (prev_types_out) *prev_types_out = W->prev_types_out;
	*out = W.release();
	return CL_OK;
}
// What dna_walk's k_last_types computes, without the walk: the flags of the last four reads come from their first tuples.  ONE one-lane
// kernel on the context's own stream, its result in a mapped host word (round 5: five blocking copies on the legacy stream, four of them
// of one byte, under the compressor's claim lock) — and it refuses offsets that do not lie inside the tuple stream or a read without a tuple.
namespace {
__global__ void k_batch_types(const uint8_t* __restrict__ es, const uint64_t* __restrict__ es_off, uint32_t n_reads, uint64_t es_bytes, uint32_t prev, unsigned long long* __restrict__ out)
{
	uint32_t v = prev; bool ok = true;
	const uint32_t start = n_reads > 4 ? n_reads - 4 : 0;
	for (uint32_t r = start; r < n_reads; ++r)
	{
		const uint64_t a = es_off[r], b = es_off[r + 1];
		if (a >= b || b > es_bytes) { ok = false; break; }                        // a read without a tuple, or offsets outside the stream
		const uint32_t t = es[a] >> 4;
		v = ((v << 2) + (t == T_START_PLAIN ? 0u : t == T_START_PLAIN_N ? 1u : 2u)) & 0xff;
	}
	*out = ok ? (unsigned long long)v : ~0ull;
}
} // namespace
cl_status cl_dna_batch_types(cl_ctx* ctx, const uint8_t* d_es, const uint64_t* d_es_off, uint32_t n_reads, uint64_t es_bytes, uint32_t prev_types, uint32_t* out)
{
	if (!ctx || !d_es || !d_es_off || !out) return CL_E_INVALID;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	uint64_t* hs = nullptr; uint64_t* ds = nullptr;
	HIP_TRY(ctx, cl_slot(ctx, 1, &hs, &ds));
	LAUNCH(ctx, k_batch_types, 1, 1, d_es, d_es_off, n_reads, es_bytes, prev_types, (unsigned long long*)ds);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	const uint64_t v = *(volatile uint64_t*)hs;
	if (v == ~0ull) return cl_fail(ctx, CL_E_INVALID, "cl_dna_batch_types: a tuple stream offset of the last reads lies outside the stream");
	*out = (uint32_t)v;
	return CL_OK;
}
void cl_dna_walked_free(DnaWalked* W) { delete W; }
void cl_dna_set_ahead(cl_dna_coder* D, DnaWalked* W) { if (D) D->ahead.reset(W); else delete W; }
void cl_dna_coder_state(const cl_dna_coder* D, uint32_t* prev_types, uint32_t* read_id) { if (prev_types) *prev_types = D->prev_types; if (read_id) *read_id = D->cur_read_id; }

// Internal (stream.hip): the walk of the batch that FOLLOWS the one being coded, from inside cl_dna_encode's before_tail hook.
cl_status cl_dna_walk_ahead(cl_ctx* ctx, cl_dna_coder* D, const cl_reads* refs, const uint8_t* d_es, const uint64_t* d_es_off, const uint32_t* d_es_ntuples, uint32_t n_reads)
{
	if (!ctx || !D || !refs || !d_es || !d_es_off || !d_es_ntuples || !n_reads) return CL_E_INVALID;
	if (!D->next_valid) return cl_fail(ctx, CL_E_INVALID, "cl_dna_walk_ahead: only while a batch is being coded");
	auto W = std::make_unique<DnaWalked>();
	CL_TRY(dna_walk(ctx, D, refs, d_es, d_es_off, d_es_ntuples, n_reads, D->next_prev_types, D->next_read_id, *W));
	D->ahead = std::move(W);
	return CL_OK;
}
void cl_dna_set_before_tail(cl_dna_coder* D, std::function<cl_status()> fn) { if (D) D->before_tail = std::move(fn); }

namespace {
// The model half of a batch: (walk and group preparation unless made ahead,) model evolution group by group, every group's
// interval coder started on a stream of the coder.
cl_status dna_evolve_batch(cl_ctx* ctx, cl_dna_coder* D, const cl_reads* refs, const uint8_t* d_es, const uint64_t* d_es_off, const uint32_t* d_es_ntuples, uint32_t n_reads,
                           const uint32_t* h_part_bounds, uint32_t n_parts, std::unique_ptr<DnaWalked> Wp, uint32_t prev_types, uint32_t read_id, DnaEvolved& E)
{
	const uint64_t* inv_tab = nullptr;
	CL_TRY(cl_inv_table(ctx, &inv_tab));
	// (groups sorted ahead hold for the part bounds they were made for; the keys are sorted in place, so with other bounds the walk is redone)
	if (Wp && Wp->presorted && (Wp->part_bounds.size() != (size_t)n_parts + 1 || memcmp(Wp->part_bounds.data(), h_part_bounds, ((size_t)n_parts + 1) * 4) != 0)) Wp.reset();
	if (!Wp) { Wp = std::make_unique<DnaWalked>(); CL_TRY(dna_walk(ctx, D, refs, d_es, d_es_off, d_es_ntuples, n_reads, prev_types, read_id, *Wp)); }
	DnaWalked& W = *Wp;
	DevBuf<uint32_t> err; DEV_ALLOC(ctx, err, 1);
	HIP_TRY(ctx, hipMemsetAsync(err.p, 0, 4, ctx->stream));
	E.d_es = d_es; E.n_reads = n_reads; E.prev_types_in = prev_types; E.read_id_in = read_id; E.prev_types_out = W.prev_types_out;
	E.part_bounds.assign(h_part_bounds, h_part_bounds + n_parts + 1);
	uint32_t p0 = 0; size_t gi = 0;
	while (p0 < n_parts)
	{
		// the group's model-independent half: made ahead (cl_dna_prepare_batch) or here
		std::unique_ptr<DnaGroupPrep> GP;
		if (W.presorted) { if (gi >= W.groups.size() || !W.groups[gi] || W.groups[gi]->p0 != p0) return cl_fail(ctx, CL_E_INVALID, "cl_dna_encode: prepared groups do not match"); GP = std::move(W.groups[gi++]); }
		else { GP = std::make_unique<DnaGroupPrep>(); CL_TRY(dna_group_prepare(ctx, D, W, p0, h_part_bounds, n_parts, *GP)); }
		const uint32_t p1 = GP->p1, np = p1 - p0, ng = (np + 63) / 64;
		const uint64_t s0 = GP->s0, n_syms = GP->n_syms;
		const std::vector<uint32_t>& rank = GP->rank; const std::vector<uint32_t>& plen_r = GP->plen_r;
		auto G = std::make_unique<PendingGroup>(); G->p0 = p0; G->np = np; G->rank = rank;
		G->d_gbase = std::move(GP->d_gbase); G->d_plen = std::move(GP->d_plen);
		DevBuf<uint64_t>& d_gbase = G->d_gbase; DevBuf<uint32_t>& d_plen = G->d_plen; DevBuf<triple_t>& trip = G->trip;
		DEV_ALLOC(ctx, trip, GP->trip_words);
		if (n_syms)
		{
			DevBuf<uint32_t>& sidx = GP->sidx; DevBuf<uint32_t>& seg = GP->seg; const uint64_t n_seg = GP->n_seg;
			// this group's keys, sorted: the model kernels for the key width of the batch
			// the model half: short runs by a wave each, then the long ones — their rescale chains and every symbol's triple from the class
			// prefixes the preparation made (no host round trip in between: sizes and places are known)
			auto models = [&](auto* gkey) -> cl_status {
				typedef std::remove_cv_t<std::remove_pointer_t<decltype(gkey)>> K;
				LAUNCHB_NAMED(ctx, "k_dna_evolve", n_syms * (12.0 + sizeof(K)), (k_dna_evolve<K>), grid_for(n_seg, 4), 256, (const FamTab*)D->d_ft.p, (const K*)gkey, (const uint32_t*)sidx.p, (const uint32_t*)seg.p,
					(uint32_t)n_seg, D->state.p, trip.p);
				HIP_TRY(ctx, hipGetLastError());
				for (int which = 0; which < 2; ++which)
				{
					DnaGroupPrep::LongPrep& L = GP->lp[which];
					const uint32_t nlr = L.n_runs, NS = which ? 32 : 8;
					if (!nlr) continue;
					DevBuf<uint32_t> d_ne, group_epoch, epochs;
					DEV_ALLOC(ctx, d_ne, nlr); DEV_ALLOC(ctx, group_epoch, L.groups + 1); DEV_ALLOC(ctx, epochs, (L.eps + 1) * (2 + 2 * NS));
					const LongRun* cr = L.runs.p; const uint64_t steps = L.steps;
					if (which == 0)
					{
						LAUNCH_NAMED(ctx, "k_long_epochs<8>", (k_long_epochs<8, K>), nlr, 64, (const FamTab*)D->d_ft.p, cr, nlr, (const K*)gkey, (const uint32_t*)L.step_pfx.p, (const uint32_t*)L.group_pfx.p,
							D->state.p, (EpochRec<8>*)epochs.p, d_ne.p, group_epoch.p, err.p);
						LAUNCHB_NAMED(ctx, "k_long_apply<8>", steps * 64 * (12.0 + sizeof(K)), (k_long_apply<8, K>), grid_for(steps * 64, 256), 256, cr, nlr, steps, (const K*)gkey, (const uint32_t*)sidx.p,
							(const uint32_t*)L.step_pfx.p, (const uint32_t*)L.group_pfx.p, (const EpochRec<8>*)epochs.p, (const uint32_t*)d_ne.p, (const uint32_t*)group_epoch.p, (const FamTab*)D->d_ft.p, trip.p);
					}
					else
					{
						LAUNCH_NAMED(ctx, "k_long_epochs<32>", (k_long_epochs<32, K>), nlr, 64, (const FamTab*)D->d_ft.p, cr, nlr, (const K*)gkey, (const uint32_t*)L.step_pfx.p, (const uint32_t*)L.group_pfx.p,
							D->state.p, (EpochRec<32>*)epochs.p, d_ne.p, group_epoch.p, err.p);
						LAUNCHB_NAMED(ctx, "k_long_apply<32>", steps * 64 * (12.0 + sizeof(K)), (k_long_apply<32, K>), grid_for(steps * 64, 256), 256, cr, nlr, steps, (const K*)gkey, (const uint32_t*)sidx.p,
							(const uint32_t*)L.step_pfx.p, (const uint32_t*)L.group_pfx.p, (const EpochRec<32>*)epochs.p, (const uint32_t*)d_ne.p, (const uint32_t*)group_epoch.p, (const FamTab*)D->d_ft.p, trip.p);
					}
					HIP_TRY(ctx, hipGetLastError());
					HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                       // (the temporaries above go back to the pool)
				}
				return CL_OK;
			};
			if (W.narrow) CL_TRY(models((const uint32_t*)(W.key32.p + s0))); else CL_TRY(models((const uint64_t*)(W.key.p + s0)));
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		}
		uint32_t herr = 0;
		HIP_TRY(ctx, hipMemcpy(&herr, err.p, 4, hipMemcpyDeviceToHost));
		if (herr & 8) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_dna_encode: epoch table of a long context run too small");
		if (herr) return cl_fail(ctx, CL_E_UNSUPPORTED, "cl_dna_encode: a read uses more than 64 alternative references");
		// interval arithmetic per part: on a stream of the coder's own, beside whatever follows on this one
		if (D->cstreams.size() < 4) { hipStream_t ns = nullptr; HIP_TRY(ctx, cl_stream_create_role(CL_ROLE_CODER, cl_level_to_prio(cl_role_level(CL_ROLE_CODER, 0)), &ns)); D->cstreams.push_back(ns); }
		G->stream = D->cstreams[D->next_cstream++ % D->cstreams.size()];
		G->out_off.resize(np + 1);
		G->out_off[0] = 0;
		for (uint32_t p = 0; p < np; ++p) { uint64_t sy = plen_r[p]; G->out_off[p + 1] = G->out_off[p] + ((sy * 18 + 7) / 8 + sy / 16 + 64 + 7) / 8 * 8; }
		DEV_ALLOC(ctx, G->tmp, G->out_off[np]);
		DEV_ALLOC(ctx, G->d_out_off, np + 1); DEV_ALLOC(ctx, G->d_size, np); DEV_ALLOC(ctx, G->d_dst_off, np);
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                         // the triples are complete
		{
			LaunchOn on(ctx, G->stream);                                          // (launch + timing events on the coder's stream)
			G->sync.s = G->stream;
			hipError_t e1 = hipMemcpyAsync(G->d_out_off.p, G->out_off.data(), (np + 1) * 8, hipMemcpyHostToDevice, G->stream);
			LAUNCH_RANGE_CODE(ctx, n_syms * 8.0, ng, (const triple_t*)trip.p, (const uint64_t*)d_gbase.p, (const uint32_t*)d_plen.p, np, G->tmp.p, (const uint64_t*)G->d_out_off.p, G->d_size.p, inv_tab);
			hipError_t e2 = hipGetLastError();
			HIP_TRY(ctx, e1); HIP_TRY(ctx, e2);
		}
		E.groups.push_back(std::move(G));
		p0 = p1;
	}
	return CL_OK;
}
} // namespace

// Internal (stream.hip): the model half of the batch that FOLLOWS the one being coded, from inside cl_dna_encode's before_tail hook:
// W is that batch, walked (and sorted) ahead by cl_dna_prepare_batch; the next cl_dna_encode finds its coders running already.
cl_status cl_dna_evolve_ahead(cl_ctx* ctx, cl_dna_coder* D, const cl_reads* refs, const uint8_t* d_es, const uint64_t* d_es_off, const uint32_t* d_es_ntuples, uint32_t n_reads,
                              const uint32_t* h_part_bounds, uint32_t n_parts, DnaWalked* W)
{
	std::unique_ptr<DnaWalked> Wp(W);
	if (!ctx || !D || !refs || !d_es || !h_part_bounds || !n_parts || !W) return CL_E_INVALID;
	if (!D->next_valid || D->evolved.size() >= 4) return cl_fail(ctx, CL_E_INVALID, "cl_dna_evolve_ahead: only while a batch is being coded, at most four batches ahead");
	const uint32_t pt = D->evolved.empty() ? D->next_prev_types : D->ahead_prev_types, rid = D->evolved.empty() ? D->next_read_id : D->ahead_read_id;
	if (W->prev_types_in != pt || W->cur_read_id_in != rid || W->d_es != d_es || W->n_reads != n_reads) return cl_fail(ctx, CL_E_INVALID, "cl_dna_evolve_ahead: not the batch that follows");
	auto E = std::make_unique<DnaEvolved>();
	CL_TRY(dna_evolve_batch(ctx, D, refs, d_es, d_es_off, d_es_ntuples, n_reads, h_part_bounds, n_parts, std::move(Wp), pt, rid, *E));
	D->ahead_prev_types = E->prev_types_out; D->ahead_read_id = rid + n_reads;
	D->evolved.push_back(std::move(E));
	return CL_OK;
}

// CEntrComprReads::Compress for a batch of whole parts (entr_read.h:56-80)
extern "C" cl_status cl_dna_encode(cl_ctx* ctx, cl_dna_coder* D, const cl_reads* refs, const uint8_t* d_es, const uint64_t* d_es_off,
                                   const uint32_t* d_es_ntuples, uint32_t n_reads, const uint32_t* h_part_bounds, uint32_t n_parts,
                                   uint8_t* d_out, uint64_t cap, uint64_t* h_part_sizes, uint64_t* n_out)
{
	if (!ctx || !D || !refs || !d_es || !d_es_off || !d_es_ntuples || !h_part_bounds || !h_part_sizes || !n_out) return cl_fail(ctx, CL_E_INVALID, "cl_dna_encode: null argument");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	for (uint32_t p = 0; p < n_parts; ++p) if (h_part_bounds[p] > h_part_bounds[p + 1]) return cl_fail(ctx, CL_E_INVALID, "cl_dna_encode: part bounds must ascend");
	if (n_parts && (h_part_bounds[0] != 0 || h_part_bounds[n_parts] != n_reads)) return cl_fail(ctx, CL_E_INVALID, "cl_dna_encode: parts must cover reads [0, n_reads)");
	*n_out = 0;
	if (!n_parts) return CL_OK;
	uint64_t written = 0;
	// D1 for ALL reads at once: the walks are one lane per read and as long as the longest read's chain of tuples takes,
	// whatever the number of reads — so one count pass and one write pass per call, not per group.  Only what follows
	// (sort, models, interval arithmetic) is grouped, by the 32-bit symbol / triple indices.
	// The model half of this batch: done ahead (cl_dna_evolve_ahead, in the tail of the call before), or now — with the
	// state-independent half made ahead (cl_dna_prepare_batch / cl_dna_walk_ahead), or made now as well.
	std::unique_ptr<DnaEvolved> Ep;
	if (!D->evolved.empty()) { Ep = std::move(D->evolved.front()); D->evolved.pop_front(); }
	if (Ep)
	{	// (the models are evolved with it already: it has to be this batch)
		if (Ep->d_es != d_es || Ep->n_reads != n_reads || Ep->prev_types_in != D->prev_types || Ep->read_id_in != D->cur_read_id || Ep->part_bounds.size() != (size_t)n_parts + 1 ||
		    memcmp(Ep->part_bounds.data(), h_part_bounds, ((size_t)n_parts + 1) * 4) != 0)
			return cl_fail(ctx, CL_E_INVALID, "cl_dna_encode: the batch evolved ahead is not the one encoded next");
		D->ahead.reset();
	}
	else
	{
		std::unique_ptr<DnaWalked> Wp;
		if (D->ahead && D->ahead->d_es == d_es && D->ahead->n_reads == n_reads && D->ahead->prev_types_in == D->prev_types && D->ahead->cur_read_id_in == D->cur_read_id) Wp = std::move(D->ahead);
		D->ahead.reset();
		Ep = std::make_unique<DnaEvolved>();
		CL_TRY(dna_evolve_batch(ctx, D, refs, d_es, d_es_off, d_es_ntuples, n_reads, h_part_bounds, n_parts, std::move(Wp), D->prev_types, D->cur_read_id, *Ep));
	}
	DnaEvolved& E = *Ep;
	D->next_prev_types = E.prev_types_out; D->next_read_id = D->cur_read_id + n_reads; D->next_valid = true;
	struct NextOff { cl_dna_coder* D; ~NextOff() { D->next_valid = false; } } next_off{ D };
	auto finish_group = [&](PendingGroup& g) -> cl_status {
		HIP_TRY(ctx, hipStreamSynchronize(g.stream));
		g.sync.s = nullptr;
		std::vector<uint64_t> size_r(g.np);                                      // by place
		HIP_TRY(ctx, hipMemcpy(size_r.data(), g.d_size.p, g.np * 8, hipMemcpyDeviceToHost));   // (a copy to pageable memory queued behind the kernel would block the host there)
		for (uint32_t p = 0; p < g.np; ++p) { h_part_sizes[g.p0 + p] = size_r[g.rank[p]]; if (h_part_sizes[g.p0 + p] == ~0ULL) return cl_fail(ctx, CL_E_CAPACITY, "cl_dna_encode: internal part buffer overflow"); }
		std::vector<uint64_t> dst_off(g.np);                                     // by place, the bytes in part order
		uint64_t w = written;
		for (uint32_t p = 0; p < g.np; ++p) { dst_off[g.rank[p]] = w; w += h_part_sizes[g.p0 + p]; }
		if (w > cap) { *n_out = w; return cl_fail(ctx, CL_E_CAPACITY, "cl_dna_encode: output capacity " + std::to_string(cap) + " too small"); }
		HIP_TRY(ctx, hipMemcpyAsync(g.d_dst_off.p, dst_off.data(), g.np * 8, hipMemcpyHostToDevice, ctx->stream));
		LAUNCH(ctx, k_gather_bytes2, g.np, 256, (const uint8_t*)g.tmp.p, (const uint64_t*)g.d_out_off.p, (const uint64_t*)g.d_dst_off.p, (const uint64_t*)g.d_size.p, d_out);
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                         // (dst_off is read by the copy above)
		written = w;
		return CL_OK;
	};
	// this batch's interval coders are one dependent chain per part (~0.1 s for a 200-kb read, 1.3 s for a part of 4 Mi symbols): the
	// caller's hook runs now, beside them (cl_compressor takes the NEXT chunk through its model half here: cl_dna_evolve_ahead)
	if (D->before_tail)
		for (;;)
		{	// (the hook answers CL_HOOK_RETRY while the next batch is not prepared yet: asked again as long as this batch's coders run)
			const cl_status hs = D->before_tail();
			if (hs == CL_OK) break;
			if (hs != CL_HOOK_RETRY) return hs;
			bool running = false;
			for (auto& g : E.groups) if (hipStreamQuery(g->stream) == hipErrorNotReady) running = true;
			(void)hipGetLastError();
			if (!running) break;
			std::this_thread::sleep_for(std::chrono::milliseconds(1));
		}
	for (auto& g : E.groups) CL_TRY(finish_group(*g));
	// carry the coder state to the next call
	D->prev_types = E.prev_types_out;
	D->cur_read_id += n_reads;
	cl_timing_collect(ctx);
	*n_out = written;
	return CL_OK;
}
