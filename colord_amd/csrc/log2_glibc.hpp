// log2_glibc.hpp — log2 of a positive normal double, bit for bit as glibc 2.28 .. 2.35+ computes it on x86-64.
//
// Why: every cost decision of the reference's encoder compares double sums of -log2(count * (1/total)) terms
// (CEntropyEstimator::calc_logs utils.h:800-810, CEntropy utils.h:706-757), the reference calls libm's log2, and the device's
// own log2 (OCML) differs from it in the last bit on ~2.7 % of the reachable arguments (tests/test_gpu_floatpin.py) — enough
// to flip a decision on a near tie somewhere in 50 Gbases and silently leave the reference's byte stream.  libm is a
// dependency that is not vendored in /root/reference: glibc (2.35 in this image; oracle/_ref/colord links its libm.a).
// Its algorithm is published (sysdeps/ieee754/dbl-64/e_log2.c, from ARM's optimized-routines): table of N = 64 intervals,
// z = x / 2^k in [0x1.6p-1, 0x1.6p0), r = (z - c) * (1/c) with c the interval's centre held as chi + clo, log2(x) = k +
// log2(c) + r/ln2 + r^2 * A(r); arguments near 1 (|x - 1| < ~0.044) use one longer polynomial B instead.  x86-64 glibc has no
// FMA variant of the double log2 (only e_log2f-fma.o exists in libm-2.35.a), so the evaluation below — plain IEEE
// multiplications and additions in the order of the C source, the r/ln2 product split into 32-bit halves — is what runs on
// every x86-64 host; the library is built with -ffp-contract=off, so the device does the same.  The 274 constants come from
// the image's libm.a (tools/gen_glibc_log2_table.py).  tests/test_gpu_floatpin.py pins host restatement == libm == device.
// Only what the encoder needs: x > 0, finite, normal (counts / totals up to 2^20).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define CL_HD __host__ __device__
#else
#define CL_HD
#endif

namespace glibc_log2 {
#if defined(__HIP_DEVICE_COMPILE__)
__device__
#endif
static const double DATA[274] = {
#include "glibc_log2_table.inc"
};
CL_HD inline uint64_t to_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
CL_HD inline double from_bits(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

CL_HD inline double log2(double x)
{
	const double* const D = DATA;
	const double invln2hi = D[0], invln2lo = D[1];
	const double* const A = D + 2; const double* const B = D + 8; const double* const T = D + 18; const double* const T2 = D + 146;
	const uint64_t ix = to_bits(x);
	const uint64_t LO = 0x3feea4af00000000ull /* 1.0 - 0x1.5b51p-5 */, HI = 0x3ff0b55900000000ull /* 1.0 + 0x1.6ab2p-5 */;
	if (ix - LO < HI - LO)
	{	// close to 1: one polynomial in r = x - 1
		if (ix == 0x3ff0000000000000ull) return 0.0;
		const double r = x - 1.0;
		const double rhi = from_bits(to_bits(r) & 0xffffffff00000000ull), rlo = r - rhi;
		const double hi = rhi * invln2hi;
		double lo = rlo * invln2hi + r * invln2lo;
		const double r2 = r * r, r4 = r2 * r2;
		const double p = r2 * (B[0] + r * B[1]);
		double y = hi + p;
		lo += hi - y + p;
		lo += r4 * (B[2] + r * B[3] + r2 * (B[4] + r * B[5]) + r4 * (B[6] + r * B[7] + r2 * (B[8] + r * B[9])));
		y += lo;
		return y;
	}
	const uint64_t tmp = ix - 0x3fe6000000000000ull;
	const int i = (int)((tmp >> 46) & 63);
	const int64_t k = (int64_t)tmp >> 52;
	const uint64_t iz = ix - (tmp & (0xfffull << 52));
	const double invc = T[2 * i], logc = T[2 * i + 1];
	const double z = from_bits(iz), kd = (double)k;
	const double r = (z - T2[2 * i] - T2[2 * i + 1]) * invc;
	const double rhi = from_bits(to_bits(r) & 0xffffffff00000000ull), rlo = r - rhi;
	const double t1 = rhi * invln2hi;
	const double t2 = rlo * invln2hi + r * invln2lo;
	const double t3 = kd + logc;
	const double hi = t3 + t1;
	const double lo = t3 - hi + t1 + t2;
	const double r2 = r * r, r4 = r2 * r2;
	const double p = A[0] + r * A[1] + r2 * (A[2] + r * A[3]) + r4 * (A[4] + r * A[5]);
	return lo + r2 * p + hi;
}
} // namespace glibc_log2
