// -log1p(t) for fp32 t in (-1, 0], correctly rounded to fp32 at a third of the cost of the
// library's fp64 log1p (which carries double-double precision nobody here needs).
//
// The momentum draw evaluates XLA's ErfInv32 on every element: w = -log1p(-x*x).  The engine's
// numerics contract is "fp64 log1p rounded once to fp32" (Numerics).  This header meets
// the same contract in ~30 fp64 operations:
//   fast path  y ~= log1p(t) with relative error < 2^-47:
//              u = 1 + t (exact for |t| >= 2^-29, otherwise the rounding error c = t - (u - 1) is
//              added back), u = m * 2^e with m in [sqrt(1/2), sqrt(2)) (split with integer operations on
//              the high word of u), s = (m - 1) / (m + 1),
//              log m = 2 s (1 + z/3 + z^2/5 + ... + z^8/17), z = s^2 <= 0.0295 (next term 2^-50),
//              y = e ln2 + log m + c.
//   Ziv test   when y lies within 2^-44 (relative) of a midpoint between two adjacent fp32 values
//              the fast result cannot decide the rounding (probability 2^-19 per element) and the
//              caller's slow path (library log1p) is used.
// Verified on the host against (float)(-log1p((double)t)) for EVERY fp32 t in (-1, 0]
// (oracle/c/check_log1p.c; 1 065 353 217 values).
//
// Compiles for the device (hipcc) and for the host (gcc, BJX_LOG1P_HOST) from the same source.
#pragma once

#ifndef __HIPCC_RTC__
#include <math.h>
#include <stdint.h>
#include <string.h>
#else  // hiprtc: no system headers; the fixed-width names live in __hip_internal there
#ifndef BJX_RTC_STDINT
#define BJX_RTC_STDINT
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef signed long long int64_t;
typedef unsigned long long uint64_t;
#endif
#endif

#ifdef BJX_LOG1P_HOST
#define BJX_L1P_FN static inline
BJX_L1P_FN double bjx_l1p_rcp(double d) { return 1.0 / d; }
BJX_L1P_FN double bjx_l1p_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
BJX_L1P_FN int64_t bjx_l1p_bits(double v) {
  int64_t b;
  memcpy(&b, &v, 8);
  return b;
}
BJX_L1P_FN double bjx_l1p_from_bits(int64_t b) {
  double v;
  memcpy(&v, &b, 8);
  return v;
}
#else
#define BJX_L1P_FN __device__ __forceinline__
BJX_L1P_FN double bjx_l1p_fma(double a, double b, double c) { return fma(a, b, c); }
BJX_L1P_FN double bjx_l1p_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);           // v_rcp_f64: ~2^-26 ... 2^-52 depending on the part
  r = bjx_l1p_fma(bjx_l1p_fma(-d, r, 1.0), r, r);  // two Newton steps: full fp64 accuracy
  r = bjx_l1p_fma(bjx_l1p_fma(-d, r, 1.0), r, r);
  return r;
}
BJX_L1P_FN int64_t bjx_l1p_bits(double v) { return __double_as_longlong(v); }
BJX_L1P_FN double bjx_l1p_from_bits(int64_t b) { return __longlong_as_double(b); }
#endif

// -log1p(t) in fp64 with relative error < 2^-47.  Requires -1 < t <= 0.
// Round 4: the exponent / mantissa split of u = 1 + t and the sqrt(1/2) threshold are integer operations on
// the HIGH WORD of u (u is a positive normal number: t > -1 means u >= 2^-24) instead of frexp + an fp64
// compare, multiply and select -- the threshold only picks which of two equally valid argument reductions
// is used (one hi-word quantum = 2^-20 relative off sqrt(1/2) moves z by 2^-21 of its bound), and the
// exhaustive host check below covers every fp32 input.
BJX_L1P_FN double bjx_neg_log1p_core(float t) {
  const double td = (double)t;
  const double u = 1.0 + td;
  const double c = td - (u - 1.0);  // exact; non-zero only for |t| < 2^-29 where u ~= 1
  const int64_t ub = bjx_l1p_bits(u);
  uint32_t hi = (uint32_t)((uint64_t)ub >> 32);
  int e = (int)(hi >> 20) - 1022;              // u = m * 2^e, m in [0.5, 1)   (frexp's convention)
  hi = (hi & 0x000FFFFFu) | 0x3FE00000u;
  if (hi < 0x3FE6A09Eu) {                      // m < sqrt(1/2), to the precision of the high word
    hi += 0x00100000u;                         // m *= 2
    e -= 1;
  }
  const double m = bjx_l1p_from_bits((int64_t)(((uint64_t)hi << 32) | ((uint64_t)ub & 0xFFFFFFFFull)));
  const double f = m - 1.0;  // exact
  const double s = f * bjx_l1p_rcp(m + 1.0);
  const double z = s * s;
  double p = 1.0 / 17.0;
  p = bjx_l1p_fma(p, z, 1.0 / 15.0);
  p = bjx_l1p_fma(p, z, 1.0 / 13.0);
  p = bjx_l1p_fma(p, z, 1.0 / 11.0);
  p = bjx_l1p_fma(p, z, 1.0 / 9.0);
  p = bjx_l1p_fma(p, z, 1.0 / 7.0);
  p = bjx_l1p_fma(p, z, 1.0 / 5.0);
  p = bjx_l1p_fma(p, z, 1.0 / 3.0);
  const double s2 = s + s;
  const double logm = bjx_l1p_fma(s2 * z, p, s2);
  const double ed = (double)e;
  // ln2 split: hi has 32 significant bits, so e*hi is exact for |e| < 2^20
  const double y = bjx_l1p_fma(ed, 6.93147180369123816490e-01, bjx_l1p_fma(ed, 1.90821492927058770002e-10, logm)) + c;
  return -y;
}

// Returns true and sets *w = (float)(-log1p((double)t)) when the fast path decides the rounding;
// returns false (ambiguous: the caller must use the library log1p) otherwise.
BJX_L1P_FN bool bjx_neg_log1p_fast(float t, float* w) {
  const double r = bjx_neg_log1p_core(t);
  // Ziv rounding test on the 29 mantissa bits fp32 discards (all of them in the low word)
  const uint32_t low = (uint32_t)bjx_l1p_bits(r) & 0x1FFFFFFFu;
  const uint32_t dist = low > 0x10000000u ? low - 0x10000000u : 0x10000000u - low;
  *w = (float)r;
  return dist > 512u;
}
