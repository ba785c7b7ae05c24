/* TEST INFRASTRUCTURE (see oracle/__init__.py): exhaustive host check of the engine's fast
 * correctly-rounded -log1p (blackjax_amd/csrc/bjx_log1p.h, compiled here for the host).
 *
 * For every fp32 t in (-1, 0] (stride 1 = exhaustive, 1 065 353 217 values; a larger stride
 * samples) it checks that whenever the fast path claims to decide the rounding, its result equals
 * (float)(-log1p((double)t)) -- the contract of oracle/prng.py::erf_inv and of the device slow path --
 * and reports how often the fast path defers and the largest relative error of its fp64 value
 * against long-double log1pl.
 *
 * usage: check_log1p [stride]     exit status 0 iff no mismatch and max error < 2^-46
 * build: gcc -O2 -ffp-contract=off -fopenmp -DBJX_LOG1P_HOST check_log1p.c -lm
 */
#define BJX_LOG1P_HOST 1
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../blackjax_amd/csrc/bjx_log1p.h"

int main(int argc, char** argv) {
  const uint32_t stride = argc > 1 ? (uint32_t)strtoul(argv[1], 0, 10) : 1u;
  const uint32_t lo = 0x80000000u, hi = 0xBF800000u; /* -0.0 .. just below -1.0 (exclusive) */
  uint64_t n = 0, deferred = 0, mismatch = 0;
  double max_rel = 0.0;
#pragma omp parallel for reduction(+ : n, deferred, mismatch) reduction(max : max_rel) schedule(static)
  for (int64_t b = lo; b < (int64_t)hi + 1; b += stride) {
    uint32_t bits = b < (int64_t)hi ? (uint32_t)b : 0u; /* the last iteration checks +0.0 */
    float t, w;
    memcpy(&t, &bits, 4);
    ++n;
    const float want = (float)(-log1p((double)t));
    if (bjx_neg_log1p_fast(t, &w)) {
      if (!(w == want)) ++mismatch;
    } else {
      ++deferred;
    }
    if (t != 0.0f) {
      const long double ref = -log1pl((long double)t);
      const long double rel = fabsl(((long double)bjx_neg_log1p_core(t) - ref) / ref);
      if ((double)rel > max_rel) max_rel = (double)rel;
    }
  }
  printf("{\"checked\": %llu, \"stride\": %u, \"deferred\": %llu, \"mismatch\": %llu, \"max_rel_err_log2\": %.2f}\n",
         (unsigned long long)n, stride, (unsigned long long)deferred, (unsigned long long)mismatch,
         max_rel > 0 ? log2(max_rel) : -1074.0);
  return (mismatch == 0 && max_rel < 0x1p-46) ? 0 : 1;
}
