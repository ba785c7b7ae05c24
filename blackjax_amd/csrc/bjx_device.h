// Device-side building blocks shared by every kernel of libbjxhip (gfx950 only).
//
// Floating-point contract (see NOTEBOOK.md "Numerics"): built with -ffp-contract=off;
// every fused multiply-add is an explicit fmaf(); reductions and scalar
// transcendentals are evaluated in fp64 and rounded once to fp32.
#pragma once

#ifndef __HIPCC_RTC__  // hiprtc (blackjax_amd/rtc.py) pre-includes the HIP device runtime and has no system headers
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "bjx_log1p.h"

#define BJX_WAVE 64

namespace bjx {

// ---------------------------------------------------------------------------------------
// threefry2x32 (Salmon et al. 2011), 20 rounds -- the block function under
// jax.random (jax/_src/prng.py).  Restated from the published algorithm.
struct Key {
  uint32_t k0, k1;
};

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) {
  return __builtin_rotateleft32(x, r);
}

__device__ __forceinline__ Key threefry2x32(Key key, uint32_t x0, uint32_t x1) {
  const uint32_t ks0 = key.k0, ks1 = key.k1, ks2 = key.k0 ^ key.k1 ^ 0x1BD11BDAu;
  x0 += ks0;
  x1 += ks1;
#define BJX_R(r) x0 += x1; x1 = rotl32(x1, r); x1 ^= x0;
  BJX_R(13) BJX_R(15) BJX_R(26) BJX_R(6)
  x0 += ks1; x1 += ks2 + 1u;
  BJX_R(17) BJX_R(29) BJX_R(16) BJX_R(24)
  x0 += ks2; x1 += ks0 + 2u;
  BJX_R(13) BJX_R(15) BJX_R(26) BJX_R(6)
  x0 += ks0; x1 += ks1 + 3u;
  BJX_R(17) BJX_R(29) BJX_R(16) BJX_R(24)
  x0 += ks1; x1 += ks2 + 4u;
  BJX_R(13) BJX_R(15) BJX_R(26) BJX_R(6)
  x0 += ks2; x1 += ks0 + 5u;
#undef BJX_R
  return Key{x0, x1};
}

// split(key, n)[i] == fold_in(key, i) == threefry(key, (hi32(i), lo32(i)))
__device__ __forceinline__ Key key_child(Key key, uint64_t i) {
  return threefry2x32(key, (uint32_t)(i >> 32), (uint32_t)i);
}

// per-chain key for global chain index `gidx` under the two layouts of include/bjx_hip.h
__device__ __forceinline__ Key chain_key(Key key, uint64_t gidx, int64_t step_fold) {
  Key kc = key_child(key, gidx);
  if (step_fold >= 0) kc = key_child(kc, (uint64_t)step_fold);
  return kc;
}

// random_bits(key, 32, shape)[i]
__device__ __forceinline__ uint32_t key_bits32(Key key, uint64_t i) {
  Key o = threefry2x32(key, (uint32_t)(i >> 32), (uint32_t)i);
  return o.k0 ^ o.k1;
}

// [0,1) float from the top 23 bits (jax/_src/random.py::_uniform)
__device__ __forceinline__ float unit_float(uint32_t bits) {
  return __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f;
}

// jax.random.uniform(key, (), float32)
__device__ __forceinline__ float key_uniform(Key key) {
  return fmaxf(0.0f, unit_float(key_bits32(key, 0)));
}

// XLA ErfInv32 (Giles' single-precision polynomial); log1p evaluated in fp64 and rounded once -- through the
// correctly-rounded table-driven path of bjx_log1p.h (no library call: the few hundred fp32 inputs whose rounding
// its fast path cannot decide are looked up).
// OPEN: the caller guarantees |x| < 1 (normal_from_bits: |u| <= 1 - 2^-24), so the |x| == 1 -> +-inf patch is dropped.
// The central polynomial (w < 5: 99.66 % of uniform draws) is evaluated for every lane; the tail polynomial sits
// behind a branch that a wave without tail lanes skips (4 of 5 waves at one element per lane).
template <bool OPEN = false>
__device__ __forceinline__ float erfinv_f32(float x) {
  const float t = -(x * x);
  float w = bjx_neg_log1p(t);
  float p;
  {
    const float wc = w - 2.5f;
    p = 2.81022636e-08f;
    p = fmaf(p, wc, 3.43273939e-07f);
    p = fmaf(p, wc, -3.5233877e-06f);
    p = fmaf(p, wc, -4.39150654e-06f);
    p = fmaf(p, wc, 0.00021858087f);
    p = fmaf(p, wc, -0.00125372503f);
    p = fmaf(p, wc, -0.00417768164f);
    p = fmaf(p, wc, 0.246640727f);
    p = fmaf(p, wc, 1.50140941f);
  }
  if (__builtin_expect(!(w < 5.0f), 0)) {
    const float wt = sqrtf(w) - 3.0f;
    float pt = -0.000200214257f;
    pt = fmaf(pt, wt, 0.000100950558f);
    pt = fmaf(pt, wt, 0.00134934322f);
    pt = fmaf(pt, wt, -0.00367342844f);
    pt = fmaf(pt, wt, 0.00573950773f);
    pt = fmaf(pt, wt, -0.0076224613f);
    pt = fmaf(pt, wt, 0.00943887047f);
    pt = fmaf(pt, wt, 1.00167406f);
    pt = fmaf(pt, wt, 2.83297682f);
    p = pt;
  }
  float r = p * x;
  if constexpr (!OPEN) {
    if (fabsf(x) == 1.0f) r = x * __builtin_inff();
  }
  return r;
}

// jax.random.normal element from its 32 random bits:
//   u = max(lo, f*(1-lo)+lo), lo = nextafter(-1,0) ; (1-lo) rounds to 2.0f ; z = sqrt(2)*erfinv(u)
// The max is the identity here (f >= 0 and rounding is monotone: fma(f, 2, lo) >= lo) and u <= 1 - 3 * 2^-24, so
// neither it nor erf_inv's |x| == 1 patch is evaluated; same values.
__device__ __forceinline__ float normal_from_bits(uint32_t bits) {
  const float lo = -0.99999994f;
  const float u = fmaf(unit_float(bits), 2.0f, lo);
  return 1.41421354f * erfinv_f32<true>(u);
}

// Four normals at once (the 16-byte piece a lane owns in the row kernels): the four threefry blocks, argument
// reductions and table fetches of bjx_log1p.h are independent straight-line code the scheduler interleaves, and the
// two rare paths -- an input the Ziv test defers, a tail-polynomial lane -- sit behind ONE branch each per piece
// instead of one per element.  Element for element the arithmetic of normal_from_bits: same values.
__device__ __forceinline__ void normal4_from_bits(const uint32_t (&bits)[4], float (&z)[4]) {
  const float lo = -0.99999994f;
  float x[4], w[4];
  bool defer = false, tail = false;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    x[e] = fmaf(unit_float(bits[e]), 2.0f, lo);
    defer |= !bjx_neg_log1p_fast(-(x[e] * x[e]), &w[e]);
  }
  if (__builtin_expect(defer, 0)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = bjx_neg_log1p(-(x[e] * x[e]));
  }
  float p[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float wc = w[e] - 2.5f;
    float q = 2.81022636e-08f;
    q = fmaf(q, wc, 3.43273939e-07f);
    q = fmaf(q, wc, -3.5233877e-06f);
    q = fmaf(q, wc, -4.39150654e-06f);
    q = fmaf(q, wc, 0.00021858087f);
    q = fmaf(q, wc, -0.00125372503f);
    q = fmaf(q, wc, -0.00417768164f);
    q = fmaf(q, wc, 0.246640727f);
    p[e] = fmaf(q, wc, 1.50140941f);
    tail |= !(w[e] < 5.0f);
  }
  if (__builtin_expect(tail, 0)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (!(w[e] < 5.0f)) {
        const float wt = sqrtf(w[e]) - 3.0f;
        float q = -0.000200214257f;
        q = fmaf(q, wt, 0.000100950558f);
        q = fmaf(q, wt, 0.00134934322f);
        q = fmaf(q, wt, -0.00367342844f);
        q = fmaf(q, wt, 0.00573950773f);
        q = fmaf(q, wt, -0.0076224613f);
        q = fmaf(q, wt, 0.00943887047f);
        q = fmaf(q, wt, 1.00167406f);
        p[e] = fmaf(q, wt, 2.83297682f);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) z[e] = 1.41421354f * (p[e] * x[e]);
}

// ---------------------------------------------------------------------------------------
// wave64 reductions (all lanes receive the result)
//
// DPP row shifts instead of the ds_bpermute butterfly a __shfl_xor loop compiles to: the permute goes
// through the LDS crossbar (~100 cycles of latency per step, twice per double) and a reduction sits
// on the critical path of every kernel that decides something per chain (kinetic energy, U-turn dot
// products: three per NUTS leaf).  Classic GCN sequence: inclusive prefix sums inside each row of 16
// lanes (row_shr 1, 2, 4, 8, zeros shifted in), lane 15 of rows 0 / 2 added into rows 1 / 3
// (row_bcast:15), lane 31 into rows 2 and 3 (row_bcast:31); lane 63 then holds the total, which
// readlane broadcasts.  The summation order differs from the butterfly's; in fp64 that moves the
// sum by ~1e-16 relative, far below the fp32 rounding every caller applies (NOTEBOOK.md section 3).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int tlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  const int thi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return v + __hiloint2double(thi, tlo);
}

__device__ __forceinline__ double wave_sum(double v) {
  v = dpp_add_f64<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add_f64<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add_f64<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add_f64<0x118, 0xf>(v);  // row_shr:8
  v = dpp_add_f64<0x142, 0xa>(v);  // row_bcast:15 -> rows 1, 3
  v = dpp_add_f64<0x143, 0xc>(v);  // row_bcast:31 -> rows 2, 3
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ float exp_cr(float x) { return (float)exp((double)x); }

// 16-byte vector helpers
struct alignas(16) F4 {
  float x, y, z, w;
};

__device__ __forceinline__ F4 ld4(const float* p) { return *reinterpret_cast<const F4*>(p); }
__device__ __forceinline__ void st4(float* p, F4 v) { *reinterpret_cast<F4*>(p) = v; }

// Nontemporal forms (global_load/store_dwordx4 ... nt): for data that streams through HBM once per
// launch and is not re-read before the caches have turned over.  Measured with tools/membw2.hip on
// MI355X (2 GiB arrays, one 16-byte piece per lane): 3-read / 2-write mix 6.17 -> 6.55 TB/s, read-only
// 6.81 -> 7.06, copy 6.32 -> 6.69 (profiles/r03/membw.json).  NOT for Infinity-Cache-sized blocks:
// there the point is that the lines stay.
typedef float bjx_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ F4 ld4_nt(const float* p) {
  const bjx_f4v v = __builtin_nontemporal_load(reinterpret_cast<const bjx_f4v*>(p));
  return F4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st4_nt(float* p, F4 v) {
  const bjx_f4v t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<bjx_f4v*>(p));
}
template <bool NT> __device__ __forceinline__ F4 ld4_t(const float* p) {
  if constexpr (NT) return ld4_nt(p);
  else return ld4(p);
}
template <bool NT> __device__ __forceinline__ void st4_t(float* p, F4 v) {
  if constexpr (NT) st4_nt(p, v);
  else st4(p, v);
}

// One wavefront sweeps a row of D floats (D % 4 == 0) 16 bytes per lane: for every span of
// U x 1 KB, `load(u, j)` is called for all its 16-byte pieces first and `body(u, j)` afterwards,
// both in ascending j (so fp64 accumulations keep their order).  Written this way because a loop
// with a run-time trip count -- or with stores that may alias later loads -- leaves only 2-4
// loads in flight per lane; here U x (loads per piece) are.
template <int U, typename Load, typename Body>
__device__ __forceinline__ void row_sweep4(int lane, int64_t D, Load&& load, Body&& body) {
  for (int64_t j0 = (int64_t)lane * 4; j0 < D; j0 += 256 * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = j0 + 256 * u;
      if (j < D) load(u, j);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = j0 + 256 * u;
      if (j < D) body(u, j);
    }
  }
}

}  // namespace bjx
