/* ontsim_core.h — the arithmetic of the synthetic ONT generator, shared VERBATIM by the host form
 * (writes the FASTQ the reference CPU path reads) and the device form (fills HBM directly, so that the GPU timer never sees
 * PCIe or a parser): both are functions of (seed, read, position) only — counter-based, no RNG state — hence bit-identical.
 * Benchmark / test INPUT tooling; not part of libcolord_hip.so.
 *
 * Recipe: genome = hash of the position (no genome array); a read copies len_src genome bases from `start` (reverse
 * complement when strand = 1) with per-base errors 2 % deletion, 3 % substitution (uniform other base), 2 % insertion
 * (uniform base, after a kept base); qualities i.i.d. from '%+5C' (Q4, Q10, Q20, Q34) with p = 0.1, 0.2, 0.4, 0.3.
 * Read lengths / starts / strands come from a table the caller draws once (colord_amd/ontsim.py, numpy lognormal). */
#ifndef ONTSIM_CORE_H
#define ONTSIM_CORE_H
#include <stdint.h>
#ifdef __HIPCC__
#define OS_FN __host__ __device__ static inline
#else
#define OS_FN static inline
#endif

OS_FN uint64_t os_mix(uint64_t x)            /* splitmix64 finaliser */
{
	x += 0x9e3779b97f4a7c15ULL;
	x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
	x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
	return x ^ (x >> 31);
}
/* genome base at position p: 32 bases per hashed word */
OS_FN uint32_t os_genome(uint64_t gseed, uint64_t p) { return (uint32_t)(os_mix(gseed ^ ((p >> 5) * 0xd1342543de82ef95ULL)) >> (2 * (p & 31))) & 3u; }
OS_FN uint64_t os_read_seed(uint64_t seed, uint64_t r) { return os_mix(seed ^ (r * 0x9e3779b97f4a7c15ULL) ^ 0x5bd1e995u); }
/* source base i of a read (after strand) */
OS_FN uint32_t os_src_base(uint64_t gseed, uint64_t start, uint32_t len_src, uint32_t strand, uint32_t i)
{
	return strand ? 3u - os_genome(gseed, start + (len_src - 1 - i)) : os_genome(gseed, start + i);
}
/* what source position i emits: returns the number of output bases (0, 1 or 2), *b0 / *b1 their codes */
OS_FN uint32_t os_emit(uint64_t rseed, uint32_t i, uint32_t g, uint32_t* b0, uint32_t* b1)
{
	const uint64_t u = os_mix(rseed ^ ((uint64_t)i * 0xa24baed4963ee407ULL));
	const uint32_t r1 = (uint32_t)u & 0xffffu;
	if (r1 < 1311u) return 0;                                              /* 2 % deletion */
	*b0 = r1 < 1311u + 1966u ? (g + 1u + (uint32_t)((u >> 16) & 0xffu) % 3u) & 3u : g;      /* 3 % substitution */
	if ((uint32_t)(u >> 24 & 0xffffu) < 1311u) { *b1 = (uint32_t)(u >> 40) & 3u; return 2; }   /* 2 % insertion */
	return 1;
}
/* quality character of output base j */
OS_FN uint8_t os_qual(uint64_t rseed, uint32_t j)
{
	const uint32_t v = (uint32_t)(os_mix(~rseed ^ ((uint64_t)(j >> 2) * 0x9fb21c651e98df25ULL)) >> (16 * (j & 3))) & 0xffffu;
	return v < 6554u ? '%' : v < 19661u ? '+' : v < 45875u ? '5' : 'C';
}
#endif
