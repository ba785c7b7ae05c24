// Generalized HMC (persistent momentum, non-reversible slice accept) for a diagonal momentum metric
// (gfx950).  C ABI in include/bjx_ghmc.h; reference lines cited there.
//
// Same layout and mapping as bjx_hmc.hip: (N, D) row-major fp32, one wavefront owns one chain row at a
// time, lanes sweep the row in 16-byte pieces.  A transition is ONE leapfrog, so these kernels together
// move ~16 words per element and transition against the leapfrog's 5: memory-bound streams.
#include <math.h>

#include "../../include/bjx_ghmc.h"
#include "bjx_device.h"
#include "bjx_host.h"

using namespace bjx;

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / BJX_WAVE;

__device__ __forceinline__ int64_t wave_row0() {
  return (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
}
__device__ __forceinline__ int64_t wave_row_stride() { return (int64_t)gridDim.x * kWavesPerBlock; }

template <int VEC>
__device__ __forceinline__ void ldv(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const F4 t = ld4(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = p[0];
  }
}
template <int VEC>
__device__ __forceinline__ void stv(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) st4(p, F4{v[0], v[1], v[2], v[3]});
  else p[0] = v[0];
}

// ghmc.py:53-64
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_ghmc_init(Key key, int64_t off, int64_t N, int64_t D, float* __restrict__ p_out, float* __restrict__ slice_out) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    const Key kc = chain_key(key, (uint64_t)(r + off), -1);
    const Key km = key_child(kc, 0), ks = key_child(kc, 1);
    for (int64_t j = (int64_t)lane * VEC; j < D; j += 64 * VEC) {
      float z[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) z[e] = normal_from_bits(key_bits32(km, (uint64_t)(j + e)));
      stv<VEC>(p_out + r * D + j, z);
    }
    // uniform(ks, (), -1, 1) = max(minval, f * (maxval - minval) + minval)
    if (lane == 0) slice_out[r] = fmaxf(-1.0f, fmaf(unit_float(key_bits32(ks, 0)), 2.0f, -1.0f));
  }
}

// ghmc.py:168-176 (+ 203-223), metrics.py:260-270
// KICK: the opening half kick and the drift of the transition's one leapfrog ride along (the arithmetic of
// k_leapfrog_diag with n_kicks = 1: p_half = fma(eps/2, g0, p); q1 = fma(eps, imm * p_half, q0)) -- the
// refresh is bound by its RNG arithmetic, so the leapfrog's five words per element cost nothing here.
// HOIST (round 4, as k_momentum_diag): one shared inverse mass matrix, 16-byte rows of at most 1 024 floats --
// mass_sqrt = 1 / sqrt(imm) of a lane's <= 16 columns is computed once per wave, every wave sweeps several rows.
template <int VEC, bool KICK, bool HOIST = false>
__global__ void __launch_bounds__(kBlock)
k_ghmc_refresh(Key key, int64_t off, int64_t fold, int64_t N, int64_t D, const float* __restrict__ imm,
               int64_t imm_stride, float alpha_s, const float* __restrict__ alpha_pc, float delta_s,
               const float* __restrict__ delta_pc, const float* __restrict__ p_prev,
               const float* __restrict__ slice_prev, float* __restrict__ p_out,
               float* __restrict__ slice_out, float* __restrict__ ke_out, float eps_s,
               const float* __restrict__ eps_pc, const float* __restrict__ q0, const float* __restrict__ g0,
               float* __restrict__ q1_out, float* __restrict__ p_half_out) {
  const int lane = threadIdx.x & 63;
  float msh[HOIST ? 16 : 1];
  if constexpr (HOIST) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int64_t j = (int64_t)lane * 4 + 256 * it;
#pragma unroll
      for (int e = 0; e < 4; ++e) msh[4 * it + e] = j < D ? 1.0f / sqrtf(imm[j + e]) : 0.0f;
    }
  }
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    const Key kc = chain_key(key, (uint64_t)(r + off), fold);
    const Key km = key_child(kc, 0);  // key_momentum, key_noise = split(rng_key)
    const float alpha = alpha_pc ? alpha_pc[r] : alpha_s;
    const float s1 = sqrtf(1.0f - alpha), s2 = sqrtf(alpha);
    const float* im = imm + r * imm_stride;
    const int64_t base = r * D;
    const float eps = KICK ? (eps_pc ? eps_pc[r] : eps_s) : 0.0f;
    const float h = eps * 0.5f, ed = eps * 1.0f;
    double acc = 0.0;
    auto piece = [&](int64_t j, const float* msv) {  // msv: this piece's hoisted mass_sqrt values, or null
      float m[VEC], pp[VEC], pn[VEC];
      ldv<VEC>(im + j, m);
      ldv<VEC>(p_prev + base + j, pp);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float z = normal_from_bits(key_bits32(km, (uint64_t)(j + e)));
        const float fresh = (msv ? msv[e] : 1.0f / sqrtf(m[e])) * z;  // metrics.py:704-709 (two roundings)
        const float t1 = pp[e] * s1, t2 = s2 * fresh;  // two products, one sum (ghmc.py:216-221)
        pn[e] = t1 + t2;
        acc += (double)(m[e] * pn[e]) * (double)pn[e];
      }
      stv<VEC>(p_out + base + j, pn);
      if constexpr (KICK) {
        float gg[VEC], qq[VEC], ph[VEC], qn[VEC];
        ldv<VEC>(g0 + base + j, gg);
        ldv<VEC>(q0 + base + j, qq);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          ph[e] = fmaf(h, gg[e], pn[e]);
          qn[e] = fmaf(ed, m[e] * ph[e], qq[e]);
        }
        stv<VEC>(p_half_out + base + j, ph);
        stv<VEC>(q1_out + base + j, qn);
      }
    };
    if constexpr (HOIST) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int64_t j = (int64_t)lane * 4 + 256 * it;
        if (j < D) piece(j, msh + 4 * it);
      }
    } else {
      for (int64_t j = (int64_t)lane * VEC; j < D; j += 64 * VEC) piece(j, nullptr);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      ke_out[r] = 0.5f * (float)acc;
      const float delta = delta_pc ? delta_pc[r] : delta_s;
      const float t = ((slice_prev[r] + 1.0f) + delta) + 0.0f;
      float m = fmodf(t, 2.0f);  // jnp "%" : result takes the sign of the divisor
      if (m != 0.0f && m < 0.0f) m += 2.0f;
      slice_out[r] = m - 1.0f;
    }
  }
}

// hmc.py:153-176 (L = 1), proposal.py:243-264, ghmc.py:186-196
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_ghmc_finish(int64_t N, int64_t D, float eps_s, const float* __restrict__ eps_pc,
              const float* __restrict__ imm, int64_t imm_stride, float thr, const float* __restrict__ q0,
              const float* __restrict__ logp0, const float* __restrict__ g0, const float* __restrict__ ke0,
              const float* __restrict__ p, const float* __restrict__ sl, const float* __restrict__ p_prev,
              const float* __restrict__ sl_prev, const float* __restrict__ q1,
              const float* __restrict__ p_half, const float* __restrict__ logp1,
              const float* __restrict__ g1, int64_t skip_begin, int64_t skip_end, float* __restrict__ q_out,
              float* __restrict__ p_out, float* __restrict__ logp_out, float* __restrict__ g_out,
              float* __restrict__ slice_out, float* __restrict__ acc_rate_out,
              uint8_t* __restrict__ is_acc_out, uint8_t* __restrict__ is_div_out,
              float* __restrict__ energy_out, float* __restrict__ p_end_out) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    const float eps = eps_pc ? eps_pc[r] : eps_s;
    const float h = eps * 0.5f;
    const int64_t base = r * D;
    const float* im = imm + r * imm_stride;
    const bool skipped = r >= skip_begin && r < skip_end;
    // pass 1: closing half kick, kinetic energy of the end state (K(-p1) == K(p1) bit for bit)
    double acc = 0.0;
    for (int64_t j = (int64_t)lane * VEC; j < D; j += 64 * VEC) {
      float m[VEC], ph[VEC], gg[VEC], pe[VEC];
      ldv<VEC>(im + j, m);
      ldv<VEC>(p_half + base + j, ph);
      ldv<VEC>(g1 + base + j, gg);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float pn = fmaf(h, gg[e], ph[e]);
        acc += (double)(m[e] * pn) * (double)pn;
        pe[e] = -1.0f * pn;
      }
      if (p_end_out) stv<VEC>(p_end_out + base + j, pe);
    }
    acc = wave_sum(acc);
    const float ke1 = 0.5f * (float)acc;
    const float lp0 = logp0[r], lp1 = logp1[r];
    const float H0 = -lp0 + ke0[r];
    const float H1 = -lp1 + ke1;
    float dE = H0 - H1;
    if (dE != dE) dE = -__builtin_inff();  // proposal.py:45-48
    const bool is_div = (-dE) > thr;
    const float p_acc = fminf(exp_cr(dE), 1.0f);
    const float s = sl[r];
    const float log_abs = (float)log((double)fabsf(s));  // fp64, rounded once
    const bool accept = log_abs <= dE;
    const float accf = accept ? 1.0f : 0.0f;
    const float t1 = exp_cr(-dE) * accf, t2 = 1.0f - accf;  // as written in proposal.py:255 (inf * 0 = NaN)
    const float s_next = s * (t1 + t2);
    if (lane == 0) {
      acc_rate_out[r] = p_acc;
      is_acc_out[r] = accept ? 1 : 0;
      is_div_out[r] = is_div ? 1 : 0;
      energy_out[r] = H1;
      logp_out[r] = skipped ? lp0 : (accept ? lp1 : lp0);
      slice_out[r] = skipped ? sl_prev[r] : s_next;
    }
    // pass 2: the new state (wave-uniform sources)
    const bool take = accept && !skipped;
    const float* qs = take ? q1 : q0;
    const float* gs = take ? g1 : g0;
    for (int64_t j = (int64_t)lane * VEC; j < D; j += 64 * VEC) {
      float a[VEC], b[VEC], pm[VEC];
      ldv<VEC>(qs + base + j, a);
      ldv<VEC>(gs + base + j, b);
      if (skipped) {
        ldv<VEC>(p_prev + base + j, pm);
      } else if (accept) {
        float ph[VEC], gg[VEC];
        ldv<VEC>(p_half + base + j, ph);
        ldv<VEC>(g1 + base + j, gg);
#pragma unroll
        for (int e = 0; e < VEC; ++e) pm[e] = -1.0f * (-1.0f * fmaf(h, gg[e], ph[e]));
      } else {
        ldv<VEC>(p + base + j, pm);
#pragma unroll
        for (int e = 0; e < VEC; ++e) pm[e] = -1.0f * pm[e];
      }
      stv<VEC>(q_out + base + j, a);
      stv<VEC>(g_out + base + j, b);
      stv<VEC>(p_out + base + j, pm);
    }
  }
}

// The same for rows of at most 256 * NI floats (16-byte accesses): the end momentum and the new gradient
// stay in registers between the energy pass and the state select, so an accepted chain (the common
// case) re-reads nothing -- 7 words per element instead of 9.
template <int NI>
__global__ void __launch_bounds__(kBlock)
k_ghmc_finish_res(int64_t N, int64_t D, float eps_s, const float* __restrict__ eps_pc,
                  const float* __restrict__ imm, int64_t imm_stride, float thr, const float* __restrict__ q0,
                  const float* __restrict__ logp0, const float* __restrict__ g0, const float* __restrict__ ke0,
                  const float* __restrict__ p, const float* __restrict__ sl, const float* __restrict__ p_prev,
                  const float* __restrict__ sl_prev, const float* __restrict__ q1,
                  const float* __restrict__ p_half, const float* __restrict__ logp1,
                  const float* __restrict__ g1, int64_t skip_begin, int64_t skip_end, float* __restrict__ q_out,
                  float* __restrict__ p_out, float* __restrict__ logp_out, float* __restrict__ g_out,
                  float* __restrict__ slice_out, float* __restrict__ acc_rate_out,
                  uint8_t* __restrict__ is_acc_out, uint8_t* __restrict__ is_div_out,
                  float* __restrict__ energy_out, float* __restrict__ p_end_out) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    const float eps = eps_pc ? eps_pc[r] : eps_s;
    const float h = eps * 0.5f;
    const int64_t base = r * D;
    const float* im = imm + r * imm_stride;
    const bool skipped = r >= skip_begin && r < skip_end;
    F4 M[NI], P1[NI], G1[NI];
    bool ok[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int64_t j = ((int64_t)lane + 64 * k) * 4;
      ok[k] = j < D;
      if (ok[k]) {
        M[k] = ld4(im + j);
        P1[k] = ld4(p_half + base + j);
        G1[k] = ld4(g1 + base + j);
      }
    }
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (ok[k]) {
        const int64_t j = ((int64_t)lane + 64 * k) * 4;
        F4& pn = P1[k];
        pn.x = fmaf(h, G1[k].x, pn.x); pn.y = fmaf(h, G1[k].y, pn.y);
        pn.z = fmaf(h, G1[k].z, pn.z); pn.w = fmaf(h, G1[k].w, pn.w);
        acc += (double)(M[k].x * pn.x) * (double)pn.x;
        acc += (double)(M[k].y * pn.y) * (double)pn.y;
        acc += (double)(M[k].z * pn.z) * (double)pn.z;
        acc += (double)(M[k].w * pn.w) * (double)pn.w;
        if (p_end_out) st4(p_end_out + base + j, F4{-1.0f * pn.x, -1.0f * pn.y, -1.0f * pn.z, -1.0f * pn.w});
      }
    acc = wave_sum(acc);
    const float ke1 = 0.5f * (float)acc;
    const float lp0 = logp0[r], lp1 = logp1[r];
    const float H0 = -lp0 + ke0[r];
    const float H1 = -lp1 + ke1;
    float dE = H0 - H1;
    if (dE != dE) dE = -__builtin_inff();
    const bool is_div = (-dE) > thr;
    const float p_acc = fminf(exp_cr(dE), 1.0f);
    const float s = sl[r];
    const float log_abs = (float)log((double)fabsf(s));
    const bool accept = log_abs <= dE;
    const float accf = accept ? 1.0f : 0.0f;
    const float t1 = exp_cr(-dE) * accf, t2 = 1.0f - accf;
    const float s_next = s * (t1 + t2);
    if (lane == 0) {
      acc_rate_out[r] = p_acc;
      is_acc_out[r] = accept ? 1 : 0;
      is_div_out[r] = is_div ? 1 : 0;
      energy_out[r] = H1;
      logp_out[r] = skipped ? lp0 : (accept ? lp1 : lp0);
      slice_out[r] = skipped ? sl_prev[r] : s_next;
    }
    const bool take = accept && !skipped;
    if (take) {
#pragma unroll
      for (int k = 0; k < NI; ++k)
        if (ok[k]) {
          const int64_t j = ((int64_t)lane + 64 * k) * 4;
          st4(q_out + base + j, ld4(q1 + base + j));
          st4(g_out + base + j, G1[k]);
          st4(p_out + base + j, P1[k]);
        }
    } else {
      const float* ps = skipped ? p_prev : p;
      const float sgn = skipped ? 1.0f : -1.0f;
#pragma unroll
      for (int k = 0; k < NI; ++k)
        if (ok[k]) {
          const int64_t j = ((int64_t)lane + 64 * k) * 4;
          const F4 pm = ld4(ps + base + j);
          st4(q_out + base + j, ld4(q0 + base + j));
          st4(g_out + base + j, ld4(g0 + base + j));
          st4(p_out + base + j, skipped ? pm : F4{sgn * pm.x, sgn * pm.y, sgn * pm.z, sgn * pm.w});
        }
    }
  }
}

}  // namespace

extern "C" {

int bjx_ghmc_init(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset, int64_t N,
                  int64_t D, float* momentum_out, float* slice_out) {
  BJX_CHECK_ARG(N >= 0 && D > 0, "bjx_ghmc_init: bad sizes");
  if (N == 0) return 0;
  BJX_CHECK_ARG(momentum_out && slice_out, "bjx_ghmc_init: null pointer");
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  const Key key{key0, key1};
  if (bjx_vec4_ok(D, momentum_out))
    hipLaunchKernelGGL(k_ghmc_init<4>, grid, block, 0, (hipStream_t)stream, key, chain_offset, N, D,
                       momentum_out, slice_out);
  else
    hipLaunchKernelGGL(k_ghmc_init<1>, grid, block, 0, (hipStream_t)stream, key, chain_offset, N, D,
                       momentum_out, slice_out);
  return bjx_check_launch("bjx_ghmc_init");
}

int bjx_ghmc_refresh(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                     int64_t step_fold, int64_t N, int64_t D, const float* imm, int64_t imm_stride,
                     float alpha, const float* alpha_per_chain, float delta,
                     const float* delta_per_chain, const float* p_prev, const float* slice_prev,
                     float* p_out, float* slice_out, float* ke_out) {
  BJX_CHECK_ARG(N >= 0 && D > 0, "bjx_ghmc_refresh: bad sizes");
  if (N == 0) return 0;
  BJX_CHECK_ARG(imm && p_prev && slice_prev && p_out && slice_out && ke_out, "bjx_ghmc_refresh: null pointer");
  BJX_CHECK_ARG(imm_stride == 0 || imm_stride == D, "bjx_ghmc_refresh: imm_stride must be 0 or D");
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  const Key key{key0, key1};
#define BJX_REFRESH(V)                                                                                   \
  hipLaunchKernelGGL((k_ghmc_refresh<V, false>), grid, block, 0, (hipStream_t)stream, key, chain_offset,   \
                     step_fold, N, D, imm, imm_stride, alpha, alpha_per_chain, delta, delta_per_chain,     \
                     p_prev, slice_prev, p_out, slice_out, ke_out, 0.0f, nullptr, nullptr, nullptr,        \
                     nullptr, nullptr)
  if (bjx_vec4_ok(D, imm, p_prev, p_out) && imm_stride == 0 && D <= 1024 && N >= 4096)
    hipLaunchKernelGGL((k_ghmc_refresh<4, false, true>), dim3(bjx_row_grid((N + 3) / 4, kWavesPerBlock)), block, 0,
                       (hipStream_t)stream, key, chain_offset, step_fold, N, D, imm, imm_stride, alpha, alpha_per_chain,
                       delta, delta_per_chain, p_prev, slice_prev, p_out, slice_out, ke_out, 0.0f, nullptr, nullptr,
                       nullptr, nullptr, nullptr);
  else if (bjx_vec4_ok(D, imm, p_prev, p_out)) BJX_REFRESH(4);
  else BJX_REFRESH(1);
#undef BJX_REFRESH
  return bjx_check_launch("bjx_ghmc_refresh");
}

int bjx_ghmc_refresh_kick(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                          int64_t step_fold, int64_t N, int64_t D, const float* imm, int64_t imm_stride,
                          float alpha, const float* alpha_per_chain, float delta,
                          const float* delta_per_chain, float eps, const float* eps_per_chain,
                          const float* p_prev, const float* slice_prev, const float* q0, const float* g0,
                          float* p_out, float* slice_out, float* ke_out, float* q1_out, float* p_half_out) {
  BJX_CHECK_ARG(N >= 0 && D > 0, "bjx_ghmc_refresh_kick: bad sizes");
  if (N == 0) return 0;
  BJX_CHECK_ARG(imm && p_prev && slice_prev && q0 && g0 && p_out && slice_out && ke_out && q1_out && p_half_out,
                "bjx_ghmc_refresh_kick: null pointer");
  BJX_CHECK_ARG(imm_stride == 0 || imm_stride == D, "bjx_ghmc_refresh_kick: imm_stride must be 0 or D");
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  const Key key{key0, key1};
#define BJX_REFRESH_KICK(V)                                                                              \
  hipLaunchKernelGGL((k_ghmc_refresh<V, true>), grid, block, 0, (hipStream_t)stream, key, chain_offset,    \
                     step_fold, N, D, imm, imm_stride, alpha, alpha_per_chain, delta, delta_per_chain,     \
                     p_prev, slice_prev, p_out, slice_out, ke_out, eps, eps_per_chain, q0, g0, q1_out,     \
                     p_half_out)
  if (bjx_vec4_ok(D, imm, p_prev, p_out, q0, g0, q1_out, p_half_out) && imm_stride == 0 && D <= 1024 && N >= 4096)
    hipLaunchKernelGGL((k_ghmc_refresh<4, true, true>), dim3(bjx_row_grid((N + 3) / 4, kWavesPerBlock)), block, 0,
                       (hipStream_t)stream, key, chain_offset, step_fold, N, D, imm, imm_stride, alpha, alpha_per_chain,
                       delta, delta_per_chain, p_prev, slice_prev, p_out, slice_out, ke_out, eps, eps_per_chain, q0, g0,
                       q1_out, p_half_out);
  else if (bjx_vec4_ok(D, imm, p_prev, p_out, q0, g0, q1_out, p_half_out)) BJX_REFRESH_KICK(4);
  else BJX_REFRESH_KICK(1);
#undef BJX_REFRESH_KICK
  return bjx_check_launch("bjx_ghmc_refresh_kick");
}

int bjx_ghmc_finish(void* stream, int64_t N, int64_t D, float eps, const float* eps_per_chain,
                    const float* imm, int64_t imm_stride, float divergence_threshold,
                    const float* q0, const float* logp0, const float* g0, const float* ke0,
                    const float* p, const float* slice, const float* p_prev, const float* slice_prev,
                    const float* q1, const float* p_half, const float* logp1, const float* g1,
                    int64_t skip_begin, int64_t skip_end, float* q_out, float* p_out, float* logp_out,
                    float* g_out, float* slice_out, float* acceptance_rate_out, uint8_t* is_accepted_out,
                    uint8_t* is_divergent_out, float* energy_out, float* p_end_out) {
  BJX_CHECK_ARG(N >= 0 && D > 0, "bjx_ghmc_finish: bad sizes");
  if (N == 0) return 0;
  BJX_CHECK_ARG(imm && q0 && logp0 && g0 && ke0 && p && slice && p_prev && slice_prev && q1 && p_half &&
                    logp1 && g1 && q_out && p_out && logp_out && g_out && slice_out &&
                    acceptance_rate_out && is_accepted_out && is_divergent_out && energy_out,
                "bjx_ghmc_finish: null pointer");
  BJX_CHECK_ARG(imm_stride == 0 || imm_stride == D, "bjx_ghmc_finish: imm_stride must be 0 or D");
  BJX_CHECK_ARG(skip_begin <= skip_end, "bjx_ghmc_finish: skip_begin must not exceed skip_end");
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
#define BJX_FINISH(V)                                                                                     \
  hipLaunchKernelGGL(k_ghmc_finish<V>, grid, block, 0, (hipStream_t)stream, N, D, eps, eps_per_chain, imm,  \
                     imm_stride, divergence_threshold, q0, logp0, g0, ke0, p, slice, p_prev, slice_prev,    \
                     q1, p_half, logp1, g1, skip_begin, skip_end, q_out, p_out, logp_out, g_out, slice_out, \
                     acceptance_rate_out, is_accepted_out, is_divergent_out, energy_out, p_end_out)
#define BJX_FINISH_RES(NI_)                                                                                  \
  hipLaunchKernelGGL(k_ghmc_finish_res<NI_>, grid, block, 0, (hipStream_t)stream, N, D, eps, eps_per_chain,     \
                     imm, imm_stride, divergence_threshold, q0, logp0, g0, ke0, p, slice, p_prev, slice_prev,  \
                     q1, p_half, logp1, g1, skip_begin, skip_end, q_out, p_out, logp_out, g_out, slice_out,    \
                     acceptance_rate_out, is_accepted_out, is_divergent_out, energy_out, p_end_out)
  if (bjx_vec4_ok(D, imm, q0, g0, p, p_prev, q1, p_half, g1, q_out, p_out, g_out, p_end_out)) {
    if (D <= 256) BJX_FINISH_RES(1);
    else if (D <= 512) BJX_FINISH_RES(2);
    else if (D <= 1024) BJX_FINISH_RES(4);
    else BJX_FINISH(4);
  } else {
    BJX_FINISH(1);
  }
#undef BJX_FINISH_RES
#undef BJX_FINISH
  return bjx_check_launch("bjx_ghmc_finish");
}

}  // extern "C"
