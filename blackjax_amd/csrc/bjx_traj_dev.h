// Device side of bjx_hmc_trajectory_diag (csrc/bjx_traj.hip): a whole HMC transition of one chain per wave with
// the log-density evaluated in registers.  A header so that the SAME code is compiled twice: by hipcc into
// libbjxhip.so for the targets the library ships, and by hiprtc at run time around a user-written target
// (blackjax_amd/rtc.py, blackjax_amd.targets.DeviceTarget) -- HIP's run-time compiler in the place a tracing
// compiler has in the reference.
//
// A target is a type with
//   template <int NI> struct Ctx;                                  per-chain constants kept in registers (may be empty)
//   template <int NI> static __device__ void init(Ctx<NI>&, int64_t D, const float* params);
//   template <int NI> static __device__ void eval(const Ctx<NI>&, int64_t D, const float* params,
//                                                 const F4 (&x)[NI], bool need_logp, F4 (&g)[NI], float& lp);
// x / g: the chain's row in NI 16-byte pieces per lane -- piece k of lane l holds columns 4 (l + 64 k) .. + 3,
// columns >= D do not exist (guard with j < D); all 64 lanes of the wave call eval together, lp must come out the
// same in every lane (bjx::wave_sum does that); need_logp = false: only g is used.
#pragma once

#include "bjx_device.h"
#include "bjx_targets_dev.h"

namespace bjx {

struct TrajArgs {
  Key key;
  int64_t off, fold, N, D, L;
  float eps_s;
  const float* eps_pc;
  const float* imm;
  int64_t imm_stride;
  float thr;
  const float* params;  // the target's parameter pointer
  const float *q0, *logp0, *g0;
  float *p0_out, *q1_out, *p_end_out, *logp1_out, *g1_out;  // HMCInfo.momentum / .proposal: each may be NULL
  float *q_out, *logp_out, *g_out, *acc_rate_out, *energy_out;
  uint8_t *is_acc_out, *is_div_out;
};

static_assert(sizeof(TrajArgs) == 216, "blackjax_amd/rtc.py mirrors this struct field for field (TrajArgs)");

struct FunnelTarget {  // Neal's funnel (bjx_targets_dev.h)
  template <int NI> struct Ctx {};
  template <int NI> static __device__ __forceinline__ void init(Ctx<NI>&, int64_t, const float*) {}
  template <int NI>
  static __device__ __forceinline__ void eval(const Ctx<NI>&, int64_t D, const float*, const F4 (&x)[NI], bool,
                                              F4 (&g)[NI], float& lp) {
    funnel_eval<NI>(D, x, g, lp);
  }
};

struct DiagGaussianTarget {  // params = 1 / variance, kept in registers; the fp64 sum only when logp is needed
  template <int NI> struct Ctx { F4 iv[NI]; };
  template <int NI>
  static __device__ __forceinline__ void init(Ctx<NI>& c, int64_t D, const float* params) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int64_t j = ((int64_t)lane + 64 * k) * 4;
      if (j < D) c.iv[k] = ld4(params + j);
    }
  }
  template <int NI>
  static __device__ __forceinline__ void eval(const Ctx<NI>& c, int64_t D, const float*, const F4 (&x)[NI],
                                              bool need_logp, F4 (&g)[NI], float& lp) {
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int64_t j = ((int64_t)lane + 64 * k) * 4;
      if (j < D) {
        g[k] = F4{-(x[k].x * c.iv[k].x), -(x[k].y * c.iv[k].y), -(x[k].z * c.iv[k].z), -(x[k].w * c.iv[k].w)};
        if (need_logp) {  // diag_gaussian_eval's sum, term for term
          acc += (double)x[k].x * (double)g[k].x;
          acc += (double)x[k].y * (double)g[k].y;
          acc += (double)x[k].z * (double)g[k].z;
          acc += (double)x[k].w * (double)g[k].w;
        }
      }
    }
    if (need_logp) {
      acc = wave_sum(acc);
      lp = (float)(0.5 * acc);
    }
  }
};

// Arithmetic = the separate kernels', expression for expression (k_momentum_diag<4, true>,
// k_leapfrog_diag_flat<2>, k_hmc_finish_diag<4>).  One wave per chain: 128 < D <= 256 NI, D % 4 == 0.
template <int NI, class Target>
__device__ __forceinline__ void hmc_trajectory_rows(const TrajArgs& a) {
  const int lane = threadIdx.x & 63;
  const int64_t D = a.D;
  const int waves = blockDim.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * waves + (threadIdx.x >> 6); r < a.N; r += (int64_t)gridDim.x * waves) {
    const int64_t base = r * D;
    const Key kc = chain_key(a.key, (uint64_t)(r + a.off), a.fold);
    const Key km = key_child(kc, 0);  // split(kc, 2)[0]   hmc.py:299
    const float* im = a.imm + r * a.imm_stride;
    const float eps = a.eps_pc ? a.eps_pc[r] : a.eps_s;
    const float h = eps * 0.5f, ed = eps * 1.0f;  // (eps * coef) * g: integrators.py:200, 236
    F4 q[NI], p[NI], g[NI], m[NI];
    typename Target::template Ctx<NI> ctx;
    Target::template init<NI>(ctx, D, a.params);
    // momentum draw (k_momentum_diag): p0 = (1 / sqrt(imm)) * normal ; ke0 = 0.5 sum (imm p0) p0
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int64_t j = ((int64_t)lane + 64 * k) * 4;
      if (j < D) {
        m[k] = ld4(im + j);
        q[k] = ld4(a.q0 + base + j);
        g[k] = ld4(a.g0 + base + j);
        const float mm[4] = {m[k].x, m[k].y, m[k].z, m[k].w};
        float pv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = normal_from_bits(key_bits32(km, (uint64_t)(j + e)));
          const float ms = 1.0f / sqrtf(mm[e]);  // metrics.py:704-709 (two roundings)
          pv[e] = ms * z;
          const float v = mm[e] * pv[e];
          acc += (double)v * (double)pv[e];
        }
        p[k] = F4{pv[0], pv[1], pv[2], pv[3]};
        if (a.p0_out) st4(a.p0_out + base + j, p[k]);
      }
    }
    acc = wave_sum(acc);
    const float ke0 = 0.5f * (float)acc;
    float lp1 = a.logp0[r];
    // L leapfrogs (trajectory.py:136-167); kicks that meet between two steps stay two separately
    // rounded half kicks, as in k_leapfrog_diag_flat<2>
    for (int64_t i = 0; i < a.L; ++i) {
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int64_t j = ((int64_t)lane + 64 * k) * 4;
        if (j < D) {
          F4 pn;
          pn.x = fmaf(h, g[k].x, p[k].x); pn.y = fmaf(h, g[k].y, p[k].y);
          pn.z = fmaf(h, g[k].z, p[k].z); pn.w = fmaf(h, g[k].w, p[k].w);
          if (i > 0) {  // closing half of step i - 1 was the first kick, this is the opening half of step i
            pn.x = fmaf(h, g[k].x, pn.x); pn.y = fmaf(h, g[k].y, pn.y);
            pn.z = fmaf(h, g[k].z, pn.z); pn.w = fmaf(h, g[k].w, pn.w);
          }
          p[k] = pn;
          q[k].x = fmaf(ed, m[k].x * pn.x, q[k].x); q[k].y = fmaf(ed, m[k].y * pn.y, q[k].y);
          q[k].z = fmaf(ed, m[k].z * pn.z, q[k].z); q[k].w = fmaf(ed, m[k].w * pn.w, q[k].w);
        }
      }
      Target::template eval<NI>(ctx, D, a.params, q, i + 1 == a.L, g, lp1);
    }
    // finish (k_hmc_finish_diag): closing half kick, flipped momentum, energies, accept, select
    acc = 0.0;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int64_t j = ((int64_t)lane + 64 * k) * 4;
      if (j < D) {
        F4 pn = p[k];
        if (a.L > 0) {
          pn.x = fmaf(h, g[k].x, p[k].x); pn.y = fmaf(h, g[k].y, p[k].y);
          pn.z = fmaf(h, g[k].z, p[k].z); pn.w = fmaf(h, g[k].w, p[k].w);
        }
        acc += (double)(m[k].x * pn.x) * (double)pn.x;
        acc += (double)(m[k].y * pn.y) * (double)pn.y;
        acc += (double)(m[k].z * pn.z) * (double)pn.z;
        acc += (double)(m[k].w * pn.w) * (double)pn.w;
        if (a.p_end_out) st4(a.p_end_out + base + j, F4{-1.0f * pn.x, -1.0f * pn.y, -1.0f * pn.z, -1.0f * pn.w});
        if (a.q1_out) st4(a.q1_out + base + j, q[k]);
        if (a.g1_out) st4(a.g1_out + base + j, g[k]);
      }
    }
    acc = wave_sum(acc);
    const float ke1 = 0.5f * (float)acc;
    const float lp0 = a.logp0[r];
    const float H0 = -lp0 + ke0;
    const float H1 = -lp1 + ke1;
    float delta = H0 - H1;
    if (delta != delta) delta = -__builtin_inff();   // proposal.py:45-48
    const bool is_div = (-delta) > a.thr;             // hmc.py:162
    const float p_acc = fminf(exp_cr(delta), 1.0f);   // proposal.py:225
    const Key ki = key_child(kc, 1);                   // split(kc, 2)[1]
    const float u = key_uniform(ki);
    const bool accept = u < p_acc;                     // proposal.py:226
    if (lane == 0) {
      a.logp_out[r] = accept ? lp1 : lp0;
      a.acc_rate_out[r] = p_acc;
      a.is_acc_out[r] = accept ? 1 : 0;
      a.is_div_out[r] = is_div ? 1 : 0;
      a.energy_out[r] = H1;
      if (a.logp1_out) a.logp1_out[r] = lp1;
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int64_t j = ((int64_t)lane + 64 * k) * 4;
      if (j < D) {
        st4(a.q_out + base + j, accept ? q[k] : ld4(a.q0 + base + j));
        st4(a.g_out + base + j, accept ? g[k] : ld4(a.g0 + base + j));
      }
    }
  }
}

// The stand-alone form of a target (what the engine calls under the external-callable contract, and what
// init() uses for the first state): one wave per row, the same eval.
template <int NI, class Target>
__device__ __forceinline__ void target_rows(int64_t N, int64_t D, const float* params, const float* q,
                                            float* logp, float* grad) {
  const int lane = threadIdx.x & 63;
  const int waves = blockDim.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * waves + (threadIdx.x >> 6); r < N; r += (int64_t)gridDim.x * waves) {
    typename Target::template Ctx<NI> ctx;
    Target::template init<NI>(ctx, D, params);
    F4 x[NI], g[NI];
    float lp = 0.0f;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int64_t j = ((int64_t)lane + 64 * k) * 4;
      if (j < D) x[k] = ld4(q + r * D + j);
    }
    Target::template eval<NI>(ctx, D, params, x, true, g, lp);
    target_store<NI>(D, g, lp, logp + r, grad + r * D);
  }
}

}  // namespace bjx
