/* TEST INFRASTRUCTURE (see oracle/__init__.py): exhaustive host check of the engine's
 * correctly-rounded -log1p (blackjax_amd/csrc/bjx_log1p.h, compiled here for the host: the header uses exactly
 * rounded IEEE operations only, so this build computes what the device computes).
 *
 * For every fp32 t in (-1, 0] (stride 1 = exhaustive, 1 065 353 217 values; a larger stride samples) it checks
 *   - whenever the fast path claims to decide the rounding, its result equals (float)(-log1p((double)t)) -- the
 *     contract of oracle/prng.py::erf_inv;
 *   - whenever it defers, the input is in the slow table BJX_L1P_SLOW with exactly that value (bjx_neg_log1p);
 * and reports how often the fast path defers and the largest relative error of its fp64 value against long-double
 * log1pl.
 *
 * usage: check_log1p [stride] [--dump]   exit status 0 iff no mismatch, no deferred input missing from the table and
 *                                        max error < 2^-48.  --dump prints "SLOW <t bits> <w bits>" per deferred input
 *                                        (what oracle/c/gen_log1p_table.py builds the slow table from) and skips the
 *                                        table-membership check.
 * build: gcc -O2 -ffp-contract=off -fopenmp -DBJX_LOG1P_HOST check_log1p.c -lm
 */
#define BJX_LOG1P_HOST 1
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../blackjax_amd/csrc/bjx_log1p.h"

int main(int argc, char** argv) {
  uint32_t stride = 1u;
  bool dump = false;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--dump")) dump = true;
    else stride = (uint32_t)strtoul(argv[i], 0, 10);
  }
  const uint32_t lo = 0x80000000u, hi = 0xBF800000u; /* -0.0 .. just below -1.0 (exclusive) */
  uint64_t n = 0, deferred = 0, mismatch = 0, missing = 0;
  double max_rel = 0.0;
#pragma omp parallel for reduction(+ : n, deferred, mismatch, missing) reduction(max : max_rel) schedule(static)
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
      if (dump) {
        uint32_t wb;
        memcpy(&wb, &want, 4);
#pragma omp critical
        printf("SLOW %08x %08x\n", bits, wb);
      } else {
        /* the product's complete function: table lookup; a miss falls back to the fast guess and is counted */
        const float got = bjx_neg_log1p(t);
        bool in_table = false;
        for (int i = 0; i < BJX_L1P_N_SLOW; ++i) in_table |= BJX_L1P_SLOW[i][0] == bits;
        if (!in_table) ++missing;
        if (!(got == want)) ++mismatch;
      }
    }
    if (t != 0.0f) {
      const long double ref = -log1pl((long double)t);
      const long double rel = fabsl(((long double)bjx_neg_log1p_core(t) - ref) / ref);
      if ((double)rel > max_rel) max_rel = (double)rel;
    }
  }
  printf("{\"checked\": %llu, \"stride\": %u, \"deferred\": %llu, \"mismatch\": %llu, \"deferred_missing_from_table\": %llu, "
         "\"slow_table_entries\": %d, \"max_rel_err_log2\": %.2f}\n",
         (unsigned long long)n, stride, (unsigned long long)deferred, (unsigned long long)mismatch,
         (unsigned long long)missing, (int)BJX_L1P_N_SLOW, max_rel > 0 ? log2(max_rel) : -1074.0);
  return (mismatch == 0 && missing == 0 && max_rel < 0x1p-48) ? 0 : 1;
}
