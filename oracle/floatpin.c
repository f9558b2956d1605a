/* floatpin.c — the host libm form of the logarithm every cost decision of the reference's encoder is made of
 * (TEST INFRASTRUCTURE ONLY).  calc_logs (utils.h:800-810): arr_log[i] = -log2((double)arr_stat[i] * sum_rec) with
 * sum_rec = 1.0 / sum, 0.0 for an empty counter; CEntropy (utils.h:706-757) takes log2 of the same products.
 * gcc on x86-64 without -mfma does not contract the product, glibc's log2 is the reference's log2 (oracle/_ref/colord is
 * linked against the same libm.a). */
#include "oracle.h"
#include <math.h>

void orc_estimator_logs(const uint32_t* count, const uint32_t* total, size_t n, double* out)
{
	for (size_t i = 0; i < n; ++i)
	{
		const double rec = 1.0 / total[i];
		out[i] = count[i] ? -log2((double)count[i] * rec) : 0.0;
	}
}
