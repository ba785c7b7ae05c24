// Row-resident value-and-gradient of the built-in targets, shared by the stand-alone target kernels
// (bjx_targets.hip) and by the free-running NUTS tick kernels when the caller asks for the log-density to
// be evaluated INSIDE the tick (bjx_nuts_async_t.target_kind, round 3): one wave, the row in NI 16-byte
// pieces per lane (D % 4 == 0, D <= 256 NI).  Because both users run this one function, a fused tick
// produces bit for bit the (logp, grad) the separate launch produces.
#pragma once

#include "bjx_device.h"

namespace bjx {

// Neal's funnel (tests/fixtures.py:81-98 of the reference), y = q[0], v = q[1:].
// x: the row (already loaded) -> g: the gradient row, lp: the log-density (the same value in every lane).
// FULL: D == 256 NI, every lane holds a piece of every row.
template <int NI, bool FULL = false>
__device__ __forceinline__ void funnel_eval(int64_t D, const F4 (&x)[NI], F4 (&g)[NI], float& lp) {
  const int lane = threadIdx.x & 63;
  double S = 0.0;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int64_t j = ((int64_t)lane + 64 * k) * 4;
    if (FULL || j < D) {
      const double a = (double)x[k].x, b = (double)x[k].y, c = (double)x[k].z, d = (double)x[k].w;
      if (j != 0) S += a * a;  // element 0 is y
      S += b * b;
      S += c * c;
      S += d * d;
    }
  }
  S = wave_sum(S);
  const float y32 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x[0].x)));  // lane 0, k = 0
  const double y = (double)y32;
  const float ey32 = (float)exp((double)(-y32));
  const double ey = (double)ey32;
  const double dm1 = (double)(D - 1);
  const float g0 = (float)(-y / 9.0 + 0.5 * ey * S - 0.5 * dm1);
  const double t = y / 3.0;
  lp = (float)(-0.5 * (t * t) - 0.5 * ey * S - 0.5 * dm1 * y);
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int64_t j = ((int64_t)lane + 64 * k) * 4;
    if (FULL || j < D) {
      g[k] = F4{-(ey32 * x[k].x), -(ey32 * x[k].y), -(ey32 * x[k].z), -(ey32 * x[k].w)};
      if (j == 0) g[k].x = g0;
    }
  }
}

// Diagonal Gaussian: g = -(q * inv_var), logp = 0.5 * sum q * g (fp64 accumulate, pieces in ascending order
// per lane, then the DPP wave sum -- the order of k_diag_gaussian<4> for rows of at most 1 024 floats).
template <int NI, bool FULL = false>
__device__ __forceinline__ void diag_gaussian_eval(int64_t D, const F4 (&x)[NI], const float* __restrict__ iv,
                                                   F4 (&g)[NI], float& lp) {
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int64_t j = ((int64_t)lane + 64 * k) * 4;
    if (FULL || j < D) {
      const F4 vv = ld4(iv + j);
      g[k] = F4{-(x[k].x * vv.x), -(x[k].y * vv.y), -(x[k].z * vv.z), -(x[k].w * vv.w)};
      acc += (double)x[k].x * (double)g[k].x;
      acc += (double)x[k].y * (double)g[k].y;
      acc += (double)x[k].z * (double)g[k].z;
      acc += (double)x[k].w * (double)g[k].w;
    }
  }
  acc = wave_sum(acc);
  lp = (float)(0.5 * acc);
}

// gradient row + log-density (lane 0) to memory
template <int NI>
__device__ __forceinline__ void target_store(int64_t D, const F4 (&g)[NI], float lp, float* __restrict__ logp,
                                             float* __restrict__ g_row) {
  const int lane = threadIdx.x & 63;
  if (lane == 0) *logp = lp;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int64_t j = ((int64_t)lane + 64 * k) * 4;
    if (j < D) st4(g_row + j, g[k]);
  }
}

template <int NI>
__device__ __forceinline__ void funnel_row(int64_t D, const F4 (&x)[NI], float* __restrict__ logp,
                                           float* __restrict__ g_row) {
  F4 g[NI];
  float lp;
  funnel_eval<NI>(D, x, g, lp);
  target_store<NI>(D, g, lp, logp, g_row);
}

template <int NI>
__device__ __forceinline__ void diag_gaussian_row(int64_t D, const F4 (&x)[NI], const float* __restrict__ iv,
                                                  float* __restrict__ logp, float* __restrict__ g_row) {
  F4 g[NI];
  float lp;
  diag_gaussian_eval<NI>(D, x, iv, g, lp);
  target_store<NI>(D, g, lp, logp, g_row);
}

}  // namespace bjx
