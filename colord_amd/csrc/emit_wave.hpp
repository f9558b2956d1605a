// emit_wave.hpp — tuple emission (encoder.cpp:663-681,1348-1443; utils.h:56-273) by one WAVE per read (device only).
//
// A read's tuples are the run-length coding of ONE long edit script — its anchors (runs of matches) and the scripts of the gaps
// between them, in the order of a depth-first walk over its frame tree — cut into StoreFrag segments where a child frame
// (alternative reference) starts or ends: match runs of 15 and more and deletion runs of more than 16 become 4-byte tuples, every
// other symbol a 1-byte tuple, and runs merge across fragments.  emit_read (encode_core.hpp) walks that with one lane per read
// (count pass) or per 2 KB of output (write pass): every lane of a wave in another branch of add -> run -> flush -> byte, about
// 50 wave-instructions per output byte, a third of an encode lane's time.
//
// Here the lanes of a wave take 64 consecutive FRAGMENTS of a frame.  What a fragment contributes is summarised as
// (head run, bytes / tuples of the runs closed inside it, tail run); summaries combine associatively (the tail of one merges
// with the head of the next when the symbols agree), so ONE wave scan gives every fragment the run that is open when it starts
// and the number of bytes before it.  The count pass is that scan alone; in the write pass every lane then codes its own
// fragment at its own offset, starting from the open run it inherited and leaving its last run open for its successor.  The
// frame logic around it (segments, headers, child frames) is the sequential walk's, executed wave-uniformly.
#pragma once
#include "encode_core.hpp"

namespace enc {

// what a stretch of script symbols puts out: head / tail = first / last run (symbol, length; length 0: the stretch is empty),
// mb / mt = bytes / tuples of the runs closed strictly inside; single: one run only (head == tail, still open at both ends)
struct RunEl { uint32_t hs, hl, ts, tl, mb, mt, single; };
__device__ inline RunEl run_empty() { return RunEl{ 0, 0, 0, 0, 0, 0, 1 }; }
__device__ inline RunEl run_single(uint32_t sym, uint32_t len) { return RunEl{ sym, len, sym, len, 0, 0, 1 }; }
__device__ inline RunEl run_combine(const RunEl& A, const RunEl& B)
{
	if (A.hl == 0) return B;
	if (B.hl == 0) return A;
	RunEl R;
	R.hs = A.hs; R.hl = A.hl; R.single = 0;
	if (A.single)
	{
		if (B.single)
		{
			if (A.hs == B.hs) return run_single(A.hs, A.hl + B.hl);
			R.ts = B.hs; R.tl = B.hl; R.mb = 0; R.mt = 0;
		}
		else if (A.hs == B.hs) { R.hl = A.hl + B.hl; R.mb = B.mb; R.mt = B.mt; R.ts = B.ts; R.tl = B.tl; }
		else { R.mb = run_bytes((char)B.hs, B.hl) + B.mb; R.mt = run_tuples((char)B.hs, B.hl) + B.mt; R.ts = B.ts; R.tl = B.tl; }
	}
	else if (B.single)
	{
		if (A.ts == B.hs) { R.mb = A.mb; R.mt = A.mt; R.ts = A.ts; R.tl = A.tl + B.hl; }
		else { R.mb = A.mb + run_bytes((char)A.ts, A.tl); R.mt = A.mt + run_tuples((char)A.ts, A.tl); R.ts = B.hs; R.tl = B.hl; }
	}
	else
	{
		if (A.ts == B.hs) { R.mb = A.mb + run_bytes((char)A.ts, A.tl + B.hl) + B.mb; R.mt = A.mt + run_tuples((char)A.ts, A.tl + B.hl) + B.mt; }
		else { R.mb = A.mb + run_bytes((char)A.ts, A.tl) + run_bytes((char)B.hs, B.hl) + B.mb; R.mt = A.mt + run_tuples((char)A.ts, A.tl) + run_tuples((char)B.hs, B.hl) + B.mt; }
		R.ts = B.ts; R.tl = B.tl;
	}
	return R;
}
// bytes / tuples a prefix that starts where everything before is closed has put out, its open last run excluded
__device__ inline uint32_t run_closed_bytes(const RunEl& P) { return (P.hl == 0 || P.single) ? 0u : run_bytes((char)P.hs, P.hl) + P.mb; }
__device__ inline uint32_t run_closed_tuples(const RunEl& P) { return (P.hl == 0 || P.single) ? 0u : run_tuples((char)P.hs, P.hl) + P.mt; }
__device__ inline RunEl run_from_sum(const GapSum& s, uint32_t k)
{
	if (!k) return run_empty();
	if (s.syms >> 16) return run_single(s.syms & 0xff, s.first_len);
	return RunEl{ s.syms & 0xff, s.first_len, (s.syms >> 8) & 0xff, s.last_len, s.mid_bytes, s.mid_tuples, 0 };
}
__device__ inline RunEl run_shfl_up(const RunEl& e, int o)
{
	RunEl r;
	r.hs = (uint32_t)__shfl_up((int)e.hs, o); r.hl = (uint32_t)__shfl_up((int)e.hl, o); r.ts = (uint32_t)__shfl_up((int)e.ts, o); r.tl = (uint32_t)__shfl_up((int)e.tl, o);
	r.mb = (uint32_t)__shfl_up((int)e.mb, o); r.mt = (uint32_t)__shfl_up((int)e.mt, o); r.single = (uint32_t)__shfl_up((int)e.single, o);
	return r;
}
__device__ inline RunEl run_bcast(const RunEl& e, int src)
{
	RunEl r;
	r.hs = (uint32_t)__shfl((int)e.hs, src); r.hl = (uint32_t)__shfl((int)e.hl, src); r.ts = (uint32_t)__shfl((int)e.ts, src); r.tl = (uint32_t)__shfl((int)e.tl, src);
	r.mb = (uint32_t)__shfl((int)e.mb, src); r.mt = (uint32_t)__shfl((int)e.mt, src); r.single = (uint32_t)__shfl((int)e.single, src);
	return r;
}

// inclusive scan of one element per lane (log steps of shuffles); `front`: what precedes lane 0 (the open run), or empty
__device__ inline RunEl run_wave_scan(const RunEl& mine, const RunEl& front, RunEl& pre, RunEl& total)
{
	const uint32_t lane = threadIdx.x & 63;
	RunEl inc = lane == 0 ? run_combine(front, mine) : mine;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const RunEl up = run_shfl_up(inc, o); if ((int)lane >= o) inc = run_combine(up, inc); }
	pre = run_shfl_up(inc, 1);
	if (lane == 0) pre = front;
	total = run_bcast(inc, 63);
	return inc;
}
// summary of a script by the whole wave (gap_summary for scripts too long for one lane)
__device__ inline RunEl run_of_script_wave(const char* es, uint32_t k)
{
	const uint32_t lane = threadIdx.x & 63;
	RunEl acc = run_empty();
	// eight symbols per lane (one 8-byte load, combined in the lane) and one wave scan per 512 symbols: the scan — six rounds of seven
	// shuffles and a combine — was paid per 64 symbols, one byte load each
	for (uint32_t x0 = 0; x0 < k; x0 += 512)
	{
		const uint32_t x = x0 + lane * 8;
		RunEl e = run_empty();
		if (x + 8 <= k)
		{
			uint64_t w; __builtin_memcpy(&w, es + x, 8);
#pragma unroll
			for (int j = 0; j < 8; ++j) e = run_combine(e, run_single((uint32_t)(w >> (8 * j)) & 0xffu, 1));
		}
		else for (uint32_t y = x; y < k; ++y) e = run_combine(e, run_single((uint32_t)(uint8_t)es[y], 1));
		RunEl pre, total;
		run_wave_scan(e, acc, pre, total);
		acc = total;
	}
	return acc;
}
// one closed run at byte offset `at` of `out` (a contiguous stretch of its own)
__device__ inline void run_write_at(uint8_t* out, uint64_t at, uint32_t sym, uint32_t rep)
{
	TupleOut o{ out, at, 0, true };
	o.start();
	SegWriter w{ &o, (char)sym, rep, true, false, 0, 0, 0, 0, 0 };
	w.flush_run();
	o.finish();
}
// Write pass of ONE long script by the whole wave: `at` = where its first closed byte goes, (sym, rep) = the run that is open
// when it starts.  Every lane takes one symbol per step; the lane at which a run ends writes it.  The script's last run stays
// open (the successor writes it), exactly as in the per-lane form.
__device__ inline void script_write_wave(uint8_t* out, uint64_t at, uint32_t sym, uint32_t rep, uint32_t d_before, const char* es, uint32_t k)
{
	const uint32_t lane = threadIdx.x & 63;
	RunEl open = rep ? run_single(sym, rep) : run_empty();
	if (d_before)
	{
		const RunEl d = run_single((uint32_t)(uint8_t)'D', d_before);
		if (open.hl && open.hs != d.hs) { if (lane == 0) run_write_at(out, at, open.hs, open.hl); at += run_bytes((char)open.hs, open.hl); open = d; }
		else open = run_combine(open, d);
	}
	for (uint32_t x0 = 0; x0 < k; x0 += 64)
	{
		const uint32_t x = x0 + lane; const bool valid = x < k;
		const uint32_t c = valid ? (uint32_t)(uint8_t)es[x] : 0u;
		const RunEl e = valid ? run_single(c, 1) : run_empty();
		RunEl pre, total;
		run_wave_scan(e, open, pre, total);
		if (valid && pre.hl && pre.ts != c) run_write_at(out, at + run_closed_bytes(pre), pre.ts, pre.tl);   // the run before this symbol ends here
		at += run_closed_bytes(total);
		open = run_single(total.ts, total.tl);
	}
}
// ... and of one long literal: `ne` read bases as insertions, one byte each
__device__ inline void literal_write_wave(uint8_t* out, uint64_t at, uint32_t sym, uint32_t rep, const ArenaV& A, uint64_t wb, uint32_t enc_start, uint32_t ne, uint32_t d_after)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t l0 = (uint32_t)(uint8_t)base_letter(arena_base_at(A, wb, enc_start));
	// the first letter joins the open run when it is the same letter; that run closes when the second letter arrives
	if (rep && sym != l0) { if (lane == 0) run_write_at(out, at, sym, rep); at += run_bytes((char)sym, rep); rep = 0; }
	if (ne >= 2) { if (lane == 0) run_write_at(out, at, l0, rep + 1); at += rep + 1; }
	// letters 1 .. ne - 2: a byte each; the last letter stays open (or closes before the deletions, by whoever writes next)
	for (uint32_t q = 1 + lane; q + 1 < ne; q += 64) out[at + (q - 1)] = (uint8_t)arena_base_at(A, wb, enc_start + q);   // tuple type 0 (insertion): the byte is the base
	if (d_after && ne >= 2)
	{	// (the last letter closes here: the deletion run that follows is the open one)
		if (lane == 0) out[at + (ne - 2)] = (uint8_t)arena_base_at(A, wb, enc_start + ne - 1);
	}
	else if (d_after && ne == 1) { if (lane == 0) run_write_at(out, at, l0, rep + 1); }
}

// the wave-uniform part of SegWriter (encode_core.hpp): the open run, the segment flags, the output position
struct WaveSeg {
	uint8_t* out; uint64_t n; uint32_t n_tuples; bool write;             // out: the read's first byte (write pass); n: bytes CLOSED so far
	uint32_t sym, rep; bool open, first; uint32_t main_id, level, ref_id, rev, last_pos;
	// bytes written by lane 0 alone (headers, the flush of a run): a contiguous stretch of its own (TupleOut handles the shared words)
	__device__ inline void uput_begin(TupleOut& u) const { u.p = out; u.n = n; u.n_tuples = 0; u.write = write && (threadIdx.x & 63) == 0; u.acc = 0; u.have = u.skip = 0; if (u.write) u.start(); }
	__device__ inline void uput_end(TupleOut& u) { u.finish(); n = u.n; n_tuples += u.n_tuples; }
	__device__ inline void flush_run()
	{
		if (!rep) return;
		TupleOut u; uput_begin(u);
		SegWriter w{ &u, (char)sym, rep, true, false, 0, 0, 0, 0, 0 };
		w.flush_run();
		uput_end(u);
		rep = 0;
	}
	// SegWriter::add for a run the whole wave agrees on
	__device__ inline void add(char s, uint32_t cnt)
	{
		if (!cnt) return;
		open_segment();
		if (rep && (uint32_t)(uint8_t)s == sym) { rep += cnt; return; }
		flush_run(); sym = (uint32_t)(uint8_t)s; rep = cnt;
	}
	// the header of a segment, emitted when its first symbol arrives (encoder.cpp:1414-1443)
	__device__ inline void open_segment()
	{
		if (open) return;
		open = true;
		TupleOut u; uput_begin(u);
		bool d_run = false;
		if (level == 0) { if (ref_id != main_id) u.tid(6, ref_id, rev); else if (!first) u.t1(7, 0); }
		else { if (ref_id != main_id) u.tid(6, ref_id, rev); else u.t1(7, 0); d_run = last_pos != 0; }
		uput_end(u);
		if (d_run) { sym = (uint32_t)(uint8_t)'D'; rep = last_pos; }               // (the run of the skipped reference part may merge with what follows)
	}
	__device__ inline bool store() { const bool had = open; if (open) { flush_run(); first = false; } open = false; return had; }
};

constexpr uint32_t EMIT_LONG = 256;        // fragments of more symbols are written by the whole wave
// One read by one wave.  WRITE = false: sizes[r], ntuples[r]; WRITE = true: the bytes at out + es_off[r].
template<bool WRITE>
__device__ inline void emit_read_wave(const ArenaV& A, const TreeV& T, uint32_t r, const uint32_t* anchors_data, uint32_t* sizes, uint32_t* ntuples, const uint64_t* es_off, uint8_t* out)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t wb = A.word_off[r];
	const uint32_t f0 = T.frame_of_read[r];
	uint32_t sf[10], si[10], s_last[10], s_cur[10]; int sp = 0;     // frame, next fragment, last_pos_in_ref, cur_pos_in_ref
	sf[0] = f0; si[0] = 0; s_last[0] = 0; s_cur[0] = 0;
	WaveSeg w{ WRITE ? out + es_off[r] : nullptr, 0, 0, WRITE, (uint32_t)(uint8_t)'M', 0, false, true, 0, 0, 0, 0, 0 };
	{
		const FrameRec& F = T.lv[0].frames[f0];
		const CandEnt& M = T.lv[0].cands[F.cand_base];
		w.main_id = M.ref_id;
		TupleOut u; w.uput_begin(u); u.tid(10, M.ref_id, M.rev); w.uput_end(u);      // start_es (encoder.cpp:1523-1527)
	}
	bool enter = true;
	while (sp >= 0)
	{
		const LevelV& L = T.lv[sp];
		const FrameRec& F = L.frames[sf[sp]];
		const CandEnt& M = L.cands[F.cand_base + F.level];
		if (enter) { w.level = F.level; w.ref_id = M.ref_id; w.rev = M.rev; w.last_pos = s_last[sp]; enter = false; }
		const uint32_t n_frag = 2 * M.n + 1;
		if (si[sp] == n_frag)
		{	// final StoreFrag of the frame (encoder.cpp:1574)
			if (w.store()) s_last[sp] = s_cur[sp];
			--sp;
			if (sp >= 0)
			{	// back in the parent: its segment restarts; the reference part the child replaced is skipped (encoder.cpp:1489)
				const LevelV& Lp = T.lv[sp]; const FrameRec& Fp = Lp.frames[sf[sp]]; const CandEnt& Mp = Lp.cands[Fp.cand_base + Fp.level];
				w.level = Fp.level; w.ref_id = Mp.ref_id; w.rev = Mp.rev; w.last_pos = s_last[sp];
				const GapRec& g = Lp.gaps[Fp.first_gap + (si[sp] - 1) / 2];
				w.add('D', g.d_after);
			}
			continue;
		}
		// up to 64 fragments, one per lane, stopping before the first gap that continues in a child frame
		const uint32_t i0 = si[sp];
		uint32_t cnt = n_frag - i0 < 64 ? n_frag - i0 : 64;
		const uint32_t i = i0 + lane; const bool valid = lane < cnt;
		uint32_t al = 0, apr = 0, gi = 0; bool is_anchor = false, is_child = false, as_es = false;
		GapRec g; g.d_before = g.es_len = g.ne = g.d_after = g.enc_start = g.aux = 0; g.es_off = 0; g.state = GS_ES;
		if (valid)
		{
			if (i & 1) { uint32_t ape; cand_anchor(M, anchors_data, i >> 1, al, ape, apr); is_anchor = true; }
			else
			{
				gi = F.first_gap + (i >> 1);
				g = L.gaps[gi];
				is_child = g.state == GS_CHILD;
				as_es = g.state == GS_ES || (g.state == GS_PENDING && L.dec[g.aux] != 0);
			}
		}
		const uint64_t child_mask = __ballot(valid && is_child);
		if (child_mask & 1)
		{	// StoreFrag of what the parent has so far, then the child frame (encoder.cpp:1483-1488)
			const uint32_t child = (uint32_t)__shfl((int)g.aux, 0);
			++si[sp];
			if (w.store()) s_last[sp] = s_cur[sp];
			++sp; sf[sp] = child; si[sp] = 0; s_last[sp] = 0; s_cur[sp] = 0;
			enter = true;
			continue;
		}
		if (child_mask) cnt = (uint32_t)__builtin_ctzll(child_mask);
		const bool mine = lane < cnt;
		// this lane's fragment as a summary
		RunEl e = run_empty();
		uint32_t lit0 = 0, lit1 = 0;
		if (mine)
		{
			if (is_anchor) e = run_single((uint32_t)(uint8_t)'M', al);
			else if (as_es)
			{
				const RunEl d = g.d_before ? run_single((uint32_t)(uint8_t)'D', g.d_before) : run_empty();
				e = run_combine(d, run_from_sum(L.sums[gi], g.es_len));
			}
			else
			{	// literal: the read part as insertions (one byte each: where such runs are cut changes nothing), then the reference part skipped (encoder.cpp:1497-1508)
				if (g.ne) { lit0 = (uint32_t)(uint8_t)base_letter(arena_base_at(A, wb, g.enc_start)); lit1 = (uint32_t)(uint8_t)base_letter(arena_base_at(A, wb, g.enc_start + g.ne - 1)); }
				RunEl x = run_empty();
				if (g.ne == 1) x = run_single(lit0, 1);
				else if (g.ne >= 2) x = RunEl{ lit0, 1, lit1, 1, g.ne - 2, g.ne - 2, 0 };
				e = run_combine(x, g.d_after ? run_single((uint32_t)(uint8_t)'D', g.d_after) : run_empty());
			}
		}
		const bool any = __ballot(mine && e.hl != 0) != 0;
		if (any) w.open_segment();                                               // (empty segments leave no trace)
		// inclusive scan over the lanes, the wave's open run in front
		RunEl pre, total;
		run_wave_scan(e, w.rep ? run_single(w.sym, w.rep) : run_empty(), pre, total);
		if (WRITE)
		{
			// fragments too long for one lane (a script or a literal of 10^3 .. 10^5 symbols would hold the whole wave): by the wave, below
			const bool is_long = mine && e.hl != 0 && !is_anchor && (as_es ? g.es_len : g.ne) > EMIT_LONG;
			const uint64_t long_mask = __ballot(is_long);
			if (mine && e.hl != 0 && !is_long)
			{	// this lane's fragment at its own offset, from the open run it inherits; its last run stays open
				TupleOut o{ w.out, w.n + run_closed_bytes(pre), 0, true };
				o.start();
				SegWriter lw{ &o, (char)pre.ts, pre.hl ? pre.tl : 0u, true, false, 0, 0, 0, 0, 0 };
				if (is_anchor) lw.run('M', al);
				else if (as_es)
				{
					if (g.d_before) lw.run('D', g.d_before);
					const uint32_t* es4 = (const uint32_t*)(L.es + g.es_off);             // script slots are dword-aligned
					for (uint32_t q = 0; q < g.es_len; q += 32)
					{	// eight independent loads in flight, then 32 symbols from registers
						uint32_t wd[8];
#pragma unroll
						for (uint32_t t = 0; t < 8; ++t) wd[t] = q + 4 * t < g.es_len ? es4[(q >> 2) + t] : 0u;
						const uint32_t nb = g.es_len - q < 32 ? g.es_len - q : 32;
						uint64_t x0 = wd[0] | ((uint64_t)wd[1] << 32), x1 = wd[2] | ((uint64_t)wd[3] << 32), x2 = wd[4] | ((uint64_t)wd[5] << 32), x3 = wd[6] | ((uint64_t)wd[7] << 32);
#pragma nounroll
						for (uint32_t t = 0; t < nb;)
						{	// a run of matches (up to 8) at once, else one symbol
							const uint64_t nm = x0 ^ 0x4d4d4d4d4d4d4d4dull;
							uint32_t c8 = nm ? (uint32_t)__builtin_ctzll(nm) >> 3 : 8u;
							if (c8 > nb - t) c8 = nb - t;
							if (c8 == 0) c8 = 1;
							const char c = (char)(x0 & 0xff);
							if (c8 == 8) { x0 = x1; x1 = x2; x2 = x3; x3 = 0; }
							else { const uint32_t sh = 8 * c8; x0 = (x0 >> sh) | (x1 << (64 - sh)); x1 = (x1 >> sh) | (x2 << (64 - sh)); x2 = (x2 >> sh) | (x3 << (64 - sh)); x3 >>= sh; }
							lw.run(c, c8);
							t += c8;
						}
					}
				}
				else
				{
					// (the runs are cut exactly as the summary above cuts them — one letter each: whoever holds the open run writes it)
					for (uint32_t q = 0; q < g.ne; ++q) { if (q) lw.flush_run(); lw.run(base_letter(arena_base_at(A, wb, g.enc_start + q)), 1); }
					if (g.d_after) lw.run('D', g.d_after);
				}
				o.finish();                                                          // (lw's open run is the successor's to write)
			}
			for (uint64_t lm = long_mask; lm; lm &= lm - 1)
			{
				const int b = __builtin_ctzll(lm);
				const uint64_t at = w.n + (uint32_t)__shfl((int)run_closed_bytes(pre), b);
				const uint32_t isym = (uint32_t)__shfl((int)pre.ts, b), irep = (uint32_t)__shfl((int)(pre.hl ? pre.tl : 0u), b);
				const uint32_t b_es = (uint32_t)__shfl((int)(as_es ? 1u : 0u), b);
				const uint32_t b_len = (uint32_t)__shfl((int)(as_es ? g.es_len : g.ne), b);
				const uint32_t b_d = (uint32_t)__shfl((int)(as_es ? g.d_before : g.d_after), b);
				const uint64_t b_off = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(g.es_off >> 32), b) << 32) | (uint32_t)__shfl((int)(uint32_t)g.es_off, b);
				const uint32_t b_start = (uint32_t)__shfl((int)g.enc_start, b);
				if (b_es) script_write_wave(w.out, at, isym, irep, b_d, L.es + b_off, b_len);
				else literal_write_wave(w.out, at, isym, irep, A, wb, b_start, b_len, b_d);
			}
		}
		// the wave's state after the group
		if (total.hl != 0) { w.n += run_closed_bytes(total); w.n_tuples += run_closed_tuples(total); w.sym = total.ts; w.rep = total.tl; }
		{
			const uint64_t am = __ballot(mine && is_anchor);
			if (am) { const int src = 63 - __builtin_clzll(am); s_cur[sp] = (uint32_t)__shfl((int)(apr + al), src); }
		}
		si[sp] += cnt;
	}
	if (!WRITE && lane == 0) { sizes[r] = (uint32_t)w.n; ntuples[r] = w.n_tuples; }
}

} // namespace enc
