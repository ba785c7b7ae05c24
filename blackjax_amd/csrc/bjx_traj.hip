// A whole HMC transition of every chain in ONE launch, for log-densities the ENGINE can evaluate itself
// (bjx_hmc_trajectory_diag, include/bjx_hip.h): momentum draw, L velocity-Verlet leapfrogs with the
// log-density and its gradient evaluated in registers, energies, Metropolis accept, select.
//
// This is outside the external-callable contract of the engine (blackjax.hmc calls logdensity_fn between
// every two leapfrogs: hmc.py:279-312 through integrators.py:104-150) -- it exists to show what that
// contract costs: with the target resident the chain's q, p, g and inverse mass row never leave the
// registers of its wave, so a leapfrog moves NO bytes; the launch is bound by the momentum draw's
// arithmetic and the few FMAs per element.  Arithmetic = the separate kernels', expression for expression
// (k_momentum_diag<4, true>, k_leapfrog_diag_flat<2>, the stand-alone target kernels, k_hmc_finish_diag<4>):
// the results are bit for bit those of the default path (tests/test_hmc_traj_gpu.py).
// One wave per chain, the row in NI 16-byte pieces per lane: 128 < D <= 1024, D % 4 == 0, diagonal metric.
#include "../../include/bjx_hip.h"
#include "../../include/bjx_nuts.h"  // BJX_TARGET_*
#include "bjx_device.h"
#include "bjx_host.h"
#include "bjx_targets_dev.h"

using namespace bjx;

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / BJX_WAVE;

struct TrajArgs {
  Key key;
  int64_t off, fold, N, D, L;
  float eps_s;
  const float* eps_pc;
  const float* imm;
  int64_t imm_stride;
  float thr;
  int32_t target_kind;
  const float* target_vec;
  const float *q0, *logp0, *g0;
  float *p0_out, *q1_out, *p_end_out, *logp1_out, *g1_out;  // HMCInfo.momentum / .proposal: each may be NULL
  float *q_out, *logp_out, *g_out, *acc_rate_out, *energy_out;
  uint8_t *is_acc_out, *is_div_out;
};

// (logp, grad) of the position in registers.  `last`: the log-density itself is only used by the energy at the
// end of the trajectory, so the Gaussian's fp64 reduction is skipped on the other steps (its gradient is
// elementwise); `iv`: the Gaussian's 1 / variance row, loaded once per chain.
template <int NI>
__device__ __forceinline__ void target_eval(const TrajArgs& a, const F4 (&x)[NI], const F4 (&iv)[NI], bool last,
                                            F4 (&g)[NI], float& lp) {
  if (a.target_kind == BJX_TARGET_NEAL_FUNNEL) {
    funnel_eval<NI>(a.D, x, g, lp);
  } else {
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int64_t j = ((int64_t)lane + 64 * k) * 4;
      if (j < a.D) {
        g[k] = F4{-(x[k].x * iv[k].x), -(x[k].y * iv[k].y), -(x[k].z * iv[k].z), -(x[k].w * iv[k].w)};
        if (last) {  // diag_gaussian_eval's sum, term for term
          acc += (double)x[k].x * (double)g[k].x;
          acc += (double)x[k].y * (double)g[k].y;
          acc += (double)x[k].z * (double)g[k].z;
          acc += (double)x[k].w * (double)g[k].w;
        }
      }
    }
    if (last) {
      acc = wave_sum(acc);
      lp = (float)(0.5 * acc);
    }
  }
}

template <int NI>
__global__ void __launch_bounds__(kBlock) k_hmc_trajectory_diag(TrajArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t D = a.D;
  for (int64_t r = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); r < a.N;
       r += (int64_t)gridDim.x * kWavesPerBlock) {
    const int64_t base = r * D;
    const Key kc = chain_key(a.key, (uint64_t)(r + a.off), a.fold);
    const Key km = key_child(kc, 0);  // split(kc, 2)[0]   hmc.py:299
    const float* im = a.imm + r * a.imm_stride;
    const float eps = a.eps_pc ? a.eps_pc[r] : a.eps_s;
    const float h = eps * 0.5f, ed = eps * 1.0f;  // (eps * coef) * g: integrators.py:200, 236
    F4 q[NI], p[NI], g[NI], m[NI], iv[NI];
    if (a.target_kind == BJX_TARGET_DIAG_GAUSSIAN) {
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int64_t j = ((int64_t)lane + 64 * k) * 4;
        if (j < D) iv[k] = ld4(a.target_vec + j);
      }
    }
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
      target_eval<NI>(a, q, iv, i + 1 == a.L, g, lp1);
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

}  // namespace

extern "C" int bjx_hmc_trajectory_diag(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                                       int64_t step_fold, int64_t N, int64_t D, int64_t num_integration_steps,
                                       float eps, const float* eps_per_chain, const float* imm,
                                       int64_t imm_stride, float divergence_threshold, int32_t target_kind,
                                       const float* target_vec, const float* q0, const float* logp0,
                                       const float* g0, float* p0_out, float* q1_out, float* p_end_out,
                                       float* logp1_out, float* g1_out, float* q_out, float* logp_out,
                                       float* g_out, float* acceptance_rate_out, uint8_t* is_accepted_out,
                                       uint8_t* is_divergent_out, float* energy_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N > 0 && num_integration_steps >= 0 && imm && q0 && logp0 && g0 && q_out && logp_out && g_out &&
                    acceptance_rate_out && is_accepted_out && is_divergent_out && energy_out,
                "bjx_hmc_trajectory_diag: bad arguments");
  BJX_CHECK_ARG(imm_stride == 0 || imm_stride == D, "bjx_hmc_trajectory_diag: imm_stride must be 0 or D");
  BJX_CHECK_ARG(D > 128 && D <= 1024 && D % 4 == 0,
                "bjx_hmc_trajectory_diag: rows of 132 ... 1 024 floats, a multiple of 4 (shorter rows take another "
                "reduction order in the separate kernels this launch must reproduce)");
  BJX_CHECK_ARG(target_kind == BJX_TARGET_NEAL_FUNNEL || (target_kind == BJX_TARGET_DIAG_GAUSSIAN && target_vec),
                "bjx_hmc_trajectory_diag: target_kind must name an engine-resident target (the diagonal Gaussian "
                "with target_vec = 1 / variance)");
  BJX_CHECK_ARG(q_out != q0 && g_out != g0, "bjx_hmc_trajectory_diag: outputs must not alias the initial state");
  BJX_CHECK_ARG(bjx_vec4_ok(D, imm, target_vec, q0, g0, p0_out, q1_out, p_end_out, g1_out, q_out, g_out),
                "bjx_hmc_trajectory_diag: rows must be 16-byte aligned");
  TrajArgs a{Key{key0, key1}, chain_offset, step_fold, N, D, num_integration_steps, eps, eps_per_chain, imm,
             imm_stride, divergence_threshold, target_kind, target_vec, q0, logp0, g0, p0_out, q1_out, p_end_out,
             logp1_out, g1_out, q_out, logp_out, g_out, acceptance_rate_out, energy_out, is_accepted_out,
             is_divergent_out};
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
  if (D <= 256) hipLaunchKernelGGL(k_hmc_trajectory_diag<1>, grid, block, 0, s, a);
  else if (D <= 512) hipLaunchKernelGGL(k_hmc_trajectory_diag<2>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(k_hmc_trajectory_diag<4>, grid, block, 0, s, a);
  return bjx_check_launch("bjx_hmc_trajectory_diag");
}
