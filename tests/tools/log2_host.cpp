// Host build of the product's glibc-exact log2 (colord_amd/csrc/log2_glibc.hpp) for the CPU suite:
// tests/test_gpu_floatpin.py compares it bit for bit with the libm of this machine.  Built by the test with
//   g++ -O2 -std=c++17 -ffp-contract=off -shared -fPIC tests/tools/log2_host.cpp -o <tmp>/liblog2_host.so
#include "../../colord_amd/csrc/log2_glibc.hpp"
#include <cstddef>
extern "C" void log2_restated(const double* x, size_t n, double* out) { for (size_t i = 0; i < n; ++i) out[i] = glibc_log2::log2(x[i]); }
// the encoder's expression (calc_logs, utils.h:800-810) on top of it
extern "C" void estimator_logs_restated(const uint32_t* count, const uint32_t* total, size_t n, double* out)
{
	for (size_t i = 0; i < n; ++i) { const double rec = 1.0 / total[i]; out[i] = count[i] ? -glibc_log2::log2((double)count[i] * rec) : 0.0; }
}
