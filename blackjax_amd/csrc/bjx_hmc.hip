// HMC transition kernels for a diagonal metric (gfx950).  C ABI in include/bjx_hip.h.
//
// Layout: (N, D) row-major fp32, one wavefront owns one chain row at a time
// (grid-stride over rows), lanes sweep the row in 16-byte pieces so every wave
// instruction moves 1 KiB of contiguous HBM.  All kernels are HBM-bound streams.
#include "../../include/bjx_hip.h"
#include "bjx_device.h"
#include "bjx_host.h"

using namespace bjx;

namespace {

constexpr int kBlock = 256;                 // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / BJX_WAVE;

__device__ __forceinline__ int64_t wave_row0() {
  return (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
}
__device__ __forceinline__ int64_t wave_row_stride() { return (int64_t)gridDim.x * kWavesPerBlock; }

// ------------------------------------------------------------------------------ RNG probes
__global__ void __launch_bounds__(kBlock) k_rng_normal(Key key, int64_t off, int64_t N, int64_t D,
                                                        float* __restrict__ z) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    Key kc = key_child(key, (uint64_t)(r + off));
    for (int64_t j = lane; j < D; j += 64) z[r * D + j] = normal_from_bits(key_bits32(kc, j));
  }
}

__global__ void __launch_bounds__(kBlock) k_rng_uniform(Key key, int64_t off, int64_t N,
                                                         float* __restrict__ u) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) u[i] = key_uniform(key_child(key, (uint64_t)(i + off)));
}

// ------------------------------------------------------------------------------ momentum draw
// p0 = (1/sqrt(imm)) * normal(km, (D,)) ; ke0 = 0.5 * sum (imm*p0)*p0   (fp64 accumulate)
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_momentum_diag(Key key, int64_t off, int64_t fold, int64_t N, int64_t D, const float* __restrict__ imm,
                int64_t imm_stride, float* __restrict__ p_out, float* __restrict__ ke_out) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    const Key kc = chain_key(key, (uint64_t)(r + off), fold);
    const Key km = key_child(kc, 0);  // split(kc, 2)[0]
    const float* im = imm + r * imm_stride;
    float* pr = p_out + r * D;
    double acc = 0.0;
    for (int64_t j = (int64_t)lane * VEC; j < D; j += 64 * VEC) {
      float m[VEC], pv[VEC];
      if constexpr (VEC == 4) {
        F4 t = ld4(im + j);
        m[0] = t.x; m[1] = t.y; m[2] = t.z; m[3] = t.w;
      } else {
        m[0] = im[j];
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float z = normal_from_bits(key_bits32(km, (uint64_t)(j + e)));
        const float ms = 1.0f / sqrtf(m[e]);  // metrics.py:704-709 (two roundings)
        pv[e] = ms * z;
        const float v = m[e] * pv[e];
        acc += (double)v * (double)pv[e];
      }
      if constexpr (VEC == 4) st4(pr + j, F4{pv[0], pv[1], pv[2], pv[3]});
      else pr[j] = pv[0];
    }
    acc = wave_sum(acc);
    if (lane == 0) ke_out[r] = 0.5f * (float)acc;
  }
}

// ------------------------------------------------------------------------------ leapfrog
// p' = fma(h,g,p) [twice if KICKS==2] ; v = imm*p' ; q' = fma(eps, v, q)
template <int VEC, int KICKS>
__global__ void __launch_bounds__(kBlock)
k_leapfrog_diag(int64_t N, int64_t D, float eps_s, const float* __restrict__ eps_pc,
                const float* __restrict__ imm, int64_t imm_stride, const float* q_in,
                const float* p_in, const float* __restrict__ g, float* q_out, float* p_out) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    const float eps = eps_pc ? eps_pc[r] : eps_s;
    const float h = eps * 0.5f;
    const int64_t base = r * D;
    const float* im = imm + r * imm_stride;
    if constexpr (VEC == 4) {
#pragma unroll 4
      for (int64_t j = (int64_t)lane * 4; j < D; j += 256) {
        const F4 pp = ld4(p_in + base + j);
        const F4 gg = ld4(g + base + j);
        const F4 qq = ld4(q_in + base + j);
        const F4 mm = ld4(im + j);
        F4 pn, qn;
        pn.x = fmaf(h, gg.x, pp.x); pn.y = fmaf(h, gg.y, pp.y);
        pn.z = fmaf(h, gg.z, pp.z); pn.w = fmaf(h, gg.w, pp.w);
        if constexpr (KICKS == 2) {
          pn.x = fmaf(h, gg.x, pn.x); pn.y = fmaf(h, gg.y, pn.y);
          pn.z = fmaf(h, gg.z, pn.z); pn.w = fmaf(h, gg.w, pn.w);
        }
        qn.x = fmaf(eps, mm.x * pn.x, qq.x); qn.y = fmaf(eps, mm.y * pn.y, qq.y);
        qn.z = fmaf(eps, mm.z * pn.z, qq.z); qn.w = fmaf(eps, mm.w * pn.w, qq.w);
        st4(p_out + base + j, pn);
        st4(q_out + base + j, qn);
      }
    } else {
      for (int64_t j = lane; j < D; j += 64) {
        float pn = fmaf(h, g[base + j], p_in[base + j]);
        if constexpr (KICKS == 2) pn = fmaf(h, g[base + j], pn);
        const float qn = fmaf(eps, im[j] * pn, q_in[base + j]);
        p_out[base + j] = pn;
        q_out[base + j] = qn;
      }
    }
  }
}

// ------------------------------------------------------------------------------ finish
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_hmc_finish_diag(Key key, int64_t off, int64_t fold, int64_t N, int64_t D, float eps_s,
                  const float* __restrict__ eps_pc, const float* __restrict__ imm,
                  int64_t imm_stride, float thr, const float* __restrict__ q0,
                  const float* __restrict__ logp0, const float* __restrict__ g0,
                  const float* __restrict__ ke0, const float* __restrict__ q1,
                  const float* __restrict__ logp1, const float* __restrict__ g1, const float* p,
                  float* p_end, float* __restrict__ q_out, float* __restrict__ logp_out,
                  float* __restrict__ g_out, float* __restrict__ acc_rate_out,
                  uint8_t* __restrict__ is_acc_out, uint8_t* __restrict__ is_div_out,
                  float* __restrict__ energy_out) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    const float eps = eps_pc ? eps_pc[r] : eps_s;
    const float h = eps * 0.5f;
    const int64_t base = r * D;
    const float* im = imm + r * imm_stride;
    double acc = 0.0;
    // pass 1: closing half kick, flipped momentum out, kinetic energy
    for (int64_t j = (int64_t)lane * VEC; j < D; j += 64 * VEC) {
      if constexpr (VEC == 4) {
        const F4 pp = ld4(p + base + j);
        const F4 gg = ld4(g1 + base + j);
        const F4 mm = ld4(im + j);
        F4 pn;
        pn.x = fmaf(h, gg.x, pp.x); pn.y = fmaf(h, gg.y, pp.y);
        pn.z = fmaf(h, gg.z, pp.z); pn.w = fmaf(h, gg.w, pp.w);
        acc += (double)(mm.x * pn.x) * (double)pn.x;
        acc += (double)(mm.y * pn.y) * (double)pn.y;
        acc += (double)(mm.z * pn.z) * (double)pn.z;
        acc += (double)(mm.w * pn.w) * (double)pn.w;
        if (p_end) st4(p_end + base + j, F4{-1.0f * pn.x, -1.0f * pn.y, -1.0f * pn.z, -1.0f * pn.w});
      } else {
        const float pn = fmaf(h, g1[base + j], p[base + j]);
        acc += (double)(im[j] * pn) * (double)pn;
        if (p_end) p_end[base + j] = -1.0f * pn;
      }
    }
    acc = wave_sum(acc);
    const float ke1 = 0.5f * (float)acc;
    const float lp0 = logp0[r], lp1 = logp1[r];
    const float H0 = -lp0 + ke0[r];
    const float H1 = -lp1 + ke1;
    float delta = H0 - H1;
    if (delta != delta) delta = -__builtin_inff();  // proposal.py:45-48
    const bool is_div = (-delta) > thr;              // hmc.py:162
    const float p_acc = fminf(exp_cr(delta), 1.0f);  // proposal.py:225
    const Key kc = chain_key(key, (uint64_t)(r + off), fold);
    const Key ki = key_child(kc, 1);                  // split(kc, 2)[1]
    const float u = key_uniform(ki);
    const bool accept = u < p_acc;                    // proposal.py:226
    if (lane == 0) {
      logp_out[r] = accept ? lp1 : lp0;
      acc_rate_out[r] = p_acc;
      is_acc_out[r] = accept ? 1 : 0;
      is_div_out[r] = is_div ? 1 : 0;
      energy_out[r] = H1;
    }
    // pass 2: select the new state (wave-uniform source)
    const float* qs = accept ? q1 : q0;
    const float* gs = accept ? g1 : g0;
    for (int64_t j = (int64_t)lane * VEC; j < D; j += 64 * VEC) {
      if constexpr (VEC == 4) {
        const F4 a = ld4(qs + base + j);
        const F4 b = ld4(gs + base + j);
        st4(q_out + base + j, a);
        st4(g_out + base + j, b);
      } else {
        q_out[base + j] = qs[base + j];
        g_out[base + j] = gs[base + j];
      }
    }
  }
}

}  // namespace

// ======================================================================================
extern "C" {

int bjx_rng_normal(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset, int64_t N,
                   int64_t D, float* z_out) {
  BJX_CHECK_ARG(N >= 0 && D >= 0 && (N == 0 || D == 0 || z_out), "bjx_rng_normal: bad arguments");
  if (N == 0 || D == 0) return 0;
  hipLaunchKernelGGL(k_rng_normal, dim3(bjx_row_grid(N, kWavesPerBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, Key{key0, key1}, chain_offset, N, D, z_out);
  return bjx_check_launch("bjx_rng_normal");
}

int bjx_rng_uniform(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset, int64_t N,
                    float* u_out) {
  BJX_CHECK_ARG(N >= 0 && (N == 0 || u_out), "bjx_rng_uniform: bad arguments");
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_rng_uniform, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, Key{key0, key1}, chain_offset, N, u_out);
  return bjx_check_launch("bjx_rng_uniform");
}

int bjx_hmc_momentum_diag(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                          int64_t step_fold, int64_t N, int64_t D, const float* imm, int64_t imm_stride,
                          float* p_out, float* ke_out) {
  BJX_CHECK_ARG(N >= 0 && D > 0 && imm && p_out && ke_out, "bjx_hmc_momentum_diag: bad arguments");
  BJX_CHECK_ARG(imm_stride == 0 || imm_stride == D, "bjx_hmc_momentum_diag: imm_stride must be 0 or D");
  if (N == 0) return 0;
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  const Key key{key0, key1};
  if (bjx_vec4_ok(D, imm, p_out))
    hipLaunchKernelGGL(k_momentum_diag<4>, grid, block, 0, (hipStream_t)stream, key, chain_offset,
                       step_fold, N, D, imm, imm_stride, p_out, ke_out);
  else
    hipLaunchKernelGGL(k_momentum_diag<1>, grid, block, 0, (hipStream_t)stream, key, chain_offset,
                       step_fold, N, D, imm, imm_stride, p_out, ke_out);
  return bjx_check_launch("bjx_hmc_momentum_diag");
}

int bjx_leapfrog_diag(void* stream, int64_t N, int64_t D, int n_kicks, float eps,
                      const float* eps_per_chain, const float* imm, int64_t imm_stride,
                      const float* q_in, const float* p_in, const float* g, float* q_out,
                      float* p_out) {
  BJX_CHECK_ARG(N >= 0 && D > 0 && imm && q_in && p_in && g && q_out && p_out,
                "bjx_leapfrog_diag: bad arguments");
  BJX_CHECK_ARG(n_kicks == 1 || n_kicks == 2, "bjx_leapfrog_diag: n_kicks must be 1 or 2");
  BJX_CHECK_ARG(imm_stride == 0 || imm_stride == D, "bjx_leapfrog_diag: imm_stride must be 0 or D");
  if (N == 0) return 0;
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
#define BJX_LF(V, K)                                                                          \
  hipLaunchKernelGGL((k_leapfrog_diag<V, K>), grid, block, 0, s, N, D, eps, eps_per_chain, imm, \
                     imm_stride, q_in, p_in, g, q_out, p_out)
  if (bjx_vec4_ok(D, imm, q_in, p_in, g, q_out, p_out)) {
    if (n_kicks == 1) BJX_LF(4, 1); else BJX_LF(4, 2);
  } else {
    if (n_kicks == 1) BJX_LF(1, 1); else BJX_LF(1, 2);
  }
#undef BJX_LF
  return bjx_check_launch("bjx_leapfrog_diag");
}

int bjx_hmc_finish_diag(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                        int64_t step_fold, int64_t N, int64_t D, float eps, const float* eps_per_chain,
                        const float* imm, int64_t imm_stride, float divergence_threshold,
                        const float* q0, const float* logp0, const float* g0, const float* ke0,
                        const float* q1, const float* logp1, const float* g1, const float* p,
                        float* p_end_out, float* q_out, float* logp_out, float* g_out,
                        float* acceptance_rate_out, uint8_t* is_accepted_out,
                        uint8_t* is_divergent_out, float* energy_out) {
  BJX_CHECK_ARG(N >= 0 && D > 0 && imm && q0 && logp0 && g0 && ke0 && q1 && logp1 && g1 && p &&
                    q_out && logp_out && g_out && acceptance_rate_out && is_accepted_out &&
                    is_divergent_out && energy_out,
                "bjx_hmc_finish_diag: bad arguments");
  BJX_CHECK_ARG(imm_stride == 0 || imm_stride == D, "bjx_hmc_finish_diag: imm_stride must be 0 or D");
  if (N == 0) return 0;
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  const Key key{key0, key1};
  hipStream_t s = (hipStream_t)stream;
#define BJX_FIN(V)                                                                              \
  hipLaunchKernelGGL(k_hmc_finish_diag<V>, grid, block, 0, s, key, chain_offset, step_fold, N, D, \
                     eps,                                                                       \
                     eps_per_chain, imm, imm_stride, divergence_threshold, q0, logp0, g0, ke0,  \
                     q1, logp1, g1, p, p_end_out, q_out, logp_out, g_out, acceptance_rate_out,  \
                     is_accepted_out, is_divergent_out, energy_out)
  if (bjx_vec4_ok(D, imm, q0, g0, q1, g1, p, p_end_out, q_out, g_out)) BJX_FIN(4);
  else BJX_FIN(1);
#undef BJX_FIN
  return bjx_check_launch("bjx_hmc_finish_diag");
}

}  // extern "C"
