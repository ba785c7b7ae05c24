// Built-in synthetic targets: value + gradient in one pass over the row (gfx950).
// They play the role of the user's log-density callable in the bench and parity
// tests (the engine itself accepts any PyTorch callable).  logp is accumulated in
// fp64 and rounded once, the gradient is element-wise fp32.
#include "../../include/bjx_hip.h"
#include "bjx_device.h"
#include "bjx_host.h"
#include "bjx_targets_dev.h"

using namespace bjx;

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / BJX_WAVE;

__device__ __forceinline__ int64_t wave_row0() {
  return (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
}
__device__ __forceinline__ int64_t wave_row_stride() { return (int64_t)gridDim.x * kWavesPerBlock; }

// g = -(q*inv_var) ; logp = 0.5 * sum q*g
template <int VEC, bool NT = false>
__global__ void __launch_bounds__(kBlock)
k_diag_gaussian(int64_t N, int64_t D, const float* __restrict__ iv, const float* __restrict__ q,
                float* __restrict__ logp, float* __restrict__ g) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    const int64_t base = r * D;
    double acc = 0.0;
    if constexpr (VEC == 4) {
      // all loads of a 4 KB span first (a loop with a run-time trip count is not batched by the
      // compiler: it keeps 2 loads in flight per lane), then arithmetic and stores in the same order
      constexpr int U = 4;
      for (int64_t j0 = (int64_t)lane * 4; j0 < D; j0 += 256 * U) {
        F4 qq[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t j = j0 + 256 * u;
          if (j < D) {
            qq[u] = ld4_t<NT>(q + base + j);  // NT: the batch streams through HBM (see ld4_nt)
            vv[u] = ld4(iv + j);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t j = j0 + 256 * u;
          if (j < D) {
            F4 gg{-(qq[u].x * vv[u].x), -(qq[u].y * vv[u].y), -(qq[u].z * vv[u].z), -(qq[u].w * vv[u].w)};
            acc += (double)qq[u].x * (double)gg.x;
            acc += (double)qq[u].y * (double)gg.y;
            acc += (double)qq[u].z * (double)gg.z;
            acc += (double)qq[u].w * (double)gg.w;
            st4_t<NT>(g + base + j, gg);
          }
        }
      }
    } else {
      for (int64_t j = lane; j < D; j += 64) {
        const float qq = q[base + j];
        const float gg = -(qq * iv[j]);
        acc += (double)qq * (double)gg;
        g[base + j] = gg;
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) logp[r] = (float)(0.5 * acc);
  }
}

// short rows (D <= 128, D % 4 == 0): G lanes per row, 64 / G rows per wave (see k_momentum_diag_short)
template <int G>
__global__ void __launch_bounds__(kBlock)
k_diag_gaussian_short(int64_t N, int64_t D, const float* __restrict__ iv, const float* __restrict__ q,
                      float* __restrict__ logp, float* __restrict__ g) {
  constexpr int R = BJX_WAVE / G;
  const int lane = threadIdx.x & 63;
  const int sub = lane / G, gl = lane % G;
  for (int64_t r0 = wave_row0() * R; r0 < N; r0 += wave_row_stride() * R) {
    const int64_t r = r0 + sub;
    const bool valid = r < N;
    double acc = 0.0;
    if (valid) {
      const int64_t base = r * D;
      for (int64_t j = (int64_t)gl * 4; j < D; j += G * 4) {
        const F4 qq = ld4(q + base + j), vv = ld4(iv + j);
        const F4 gg{-(qq.x * vv.x), -(qq.y * vv.y), -(qq.z * vv.z), -(qq.w * vv.w)};
        acc += (double)qq.x * (double)gg.x;
        acc += (double)qq.y * (double)gg.y;
        acc += (double)qq.z * (double)gg.z;
        acc += (double)qq.w * (double)gg.w;
        st4(g + base + j, gg);
      }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, BJX_WAVE);
    if (valid && gl == 0) logp[r] = (float)(0.5 * acc);
  }
}

// Neal's funnel (tests/fixtures.py:81-98 of the reference), y = q[0], v = q[1:]
__global__ void __launch_bounds__(kBlock)
k_neal_funnel(int64_t N, int64_t D, const float* __restrict__ q, float* __restrict__ logp,
              float* __restrict__ g) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    const int64_t base = r * D;
    double S = 0.0;
    for (int64_t j = 1 + lane; j < D; j += 64) {
      const double v = (double)q[base + j];
      S += v * v;
    }
    S = wave_sum(S);
    const float y32 = q[base];
    const double y = (double)y32;
    const float ey32 = (float)exp((double)(-y32));
    const double ey = (double)ey32;
    const double dm1 = (double)(D - 1);
    if (lane == 0) {
      const double t = y / 3.0;
      logp[r] = (float)(-0.5 * (t * t) - 0.5 * ey * S - 0.5 * dm1 * y);
      g[base] = (float)(-y / 9.0 + 0.5 * ey * S - 0.5 * dm1);
    }
    for (int64_t j = 1 + lane; j < D; j += 64) g[base + j] = -(ey32 * q[base + j]);
  }
}

// The same target with 16-byte row accesses and the row kept in registers between the reduction and
// the gradient (D % 4 == 0, D <= 1024, 16-byte aligned rows): one load and one store instruction
// per 1 KiB of row instead of four 4-byte loads, a re-read and four 4-byte stores -- this callable
// sits between every two ticks of a NUTS run, where a tick is a few microseconds of dependent work.
// fp64 accumulation as above (the summation order differs, the fp64 sum rounds to the same fp32).
template <int NI>
__global__ void __launch_bounds__(64)
k_neal_funnel_v4(int64_t N, int64_t D, const float* __restrict__ q, float* __restrict__ logp,
                 float* __restrict__ g) {
  const int lane = threadIdx.x & 63;
  // one wave per WORKGROUP, row r in workgroup r: the same row -> XCD mapping as the NUTS tick kernels
  // (workgroup b runs on XCD b % 8), so the positions a tick wrote and the gradient this kernel
  // writes are found in the L2 of the XCD that uses them next
  for (int64_t r = blockIdx.x; r < N; r += gridDim.x) {
    const int64_t base = r * D;
    F4 x[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int64_t j = ((int64_t)lane + 64 * k) * 4;
      if (j < D) x[k] = ld4(q + base + j);
    }
    funnel_row<NI>(D, x, logp + r, g + base);  // bjx_targets_dev.h: shared with the fused NUTS tick
  }
}

// AR(1) Gaussian, tridiagonal precision: t = d_j q_j ; t = fma(off, q_{j-1}, t) ; t = fma(off, q_{j+1}, t)
__global__ void __launch_bounds__(kBlock)
k_ar1_gaussian(int64_t N, int64_t D, float d_edge, float d_mid, float off,
               const float* __restrict__ q, float* __restrict__ logp, float* __restrict__ g) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    const int64_t base = r * D;
    double acc = 0.0;
    for (int64_t j = lane; j < D; j += 64) {
      const float qj = q[base + j];
      const float ql = j > 0 ? q[base + j - 1] : 0.0f;
      const float qr = j + 1 < D ? q[base + j + 1] : 0.0f;
      const float d = (j == 0 || j == D - 1) ? d_edge : d_mid;
      float t = d * qj;
      t = fmaf(off, ql, t);
      t = fmaf(off, qr, t);
      const float gj = -t;
      acc += (double)qj * (double)gj;
      g[base + j] = gj;
    }
    acc = wave_sum(acc);
    if (lane == 0) logp[r] = (float)(0.5 * acc);
  }
}

}  // namespace

extern "C" {

int bjx_target_diag_gaussian(void* stream, int64_t N, int64_t D, const float* inv_var,
                             const float* q, float* logp_out, float* g_out) {
  BJX_CHECK_ARG(N >= 0 && D > 0, "bjx_target_diag_gaussian: bad arguments");
  if (N == 0) return 0;  // an empty batch has no buffers to check
  BJX_CHECK_ARG(inv_var && q && logp_out && g_out, "bjx_target_diag_gaussian: bad arguments");
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  if (bjx_vec4_ok(D, inv_var, q, g_out) && D <= 16)
    hipLaunchKernelGGL(k_diag_gaussian_short<4>, dim3(bjx_row_grid((N + 15) / 16, kWavesPerBlock)), block, 0,
                       (hipStream_t)stream, N, D, inv_var, q, logp_out, g_out);
  else if (bjx_vec4_ok(D, inv_var, q, g_out) && D <= 32)
    hipLaunchKernelGGL(k_diag_gaussian_short<8>, dim3(bjx_row_grid((N + 7) / 8, kWavesPerBlock)), block, 0,
                       (hipStream_t)stream, N, D, inv_var, q, logp_out, g_out);
  else if (bjx_vec4_ok(D, inv_var, q, g_out) && D <= 64)
    hipLaunchKernelGGL(k_diag_gaussian_short<16>, dim3(bjx_row_grid((N + 3) / 4, kWavesPerBlock)), block, 0,
                       (hipStream_t)stream, N, D, inv_var, q, logp_out, g_out);
  else if (bjx_vec4_ok(D, inv_var, q, g_out) && D <= 128)
    hipLaunchKernelGGL(k_diag_gaussian_short<32>, dim3(bjx_row_grid((N + 1) / 2, kWavesPerBlock)), block, 0,
                       (hipStream_t)stream, N, D, inv_var, q, logp_out, g_out);
  else if (bjx_vec4_ok(D, inv_var, q, g_out)) {
    // a batch larger than the Infinity Cache streams from HBM: nontemporal q loads / g stores
    // (BJX_LF_NT=0 / 1 forces plain / nontemporal, the switch of the leapfrog kernel)
    static const int nt_mode = [] { const char* e = getenv("BJX_LF_NT"); return e ? atoi(e) : -1; }();
    const bool nt = nt_mode < 0 ? N * D * 8 > ((int64_t)256 << 20) : nt_mode != 0;
    if (nt)
      hipLaunchKernelGGL((k_diag_gaussian<4, true>), grid, block, 0, (hipStream_t)stream, N, D, inv_var, q,
                         logp_out, g_out);
    else
      hipLaunchKernelGGL((k_diag_gaussian<4, false>), grid, block, 0, (hipStream_t)stream, N, D, inv_var, q,
                         logp_out, g_out);
  } else
    hipLaunchKernelGGL(k_diag_gaussian<1>, grid, block, 0, (hipStream_t)stream, N, D, inv_var, q,
                       logp_out, g_out);
  return bjx_check_launch("bjx_target_diag_gaussian");
}

int bjx_target_neal_funnel(void* stream, int64_t N, int64_t D, const float* q, float* logp_out,
                           float* g_out) {
  BJX_CHECK_ARG(N >= 0 && D >= 2, "bjx_target_neal_funnel: bad arguments");
  if (N == 0) return 0;
  BJX_CHECK_ARG(q && logp_out && g_out, "bjx_target_neal_funnel: bad arguments");
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock));
  hipStream_t st = (hipStream_t)stream;
  if (bjx_vec4_ok(D, q, g_out) && D <= 1024) {
    const dim3 wgrid((unsigned)(N < ((int64_t)1 << 20) ? N : ((int64_t)1 << 20)));
    if (D <= 256) hipLaunchKernelGGL(k_neal_funnel_v4<1>, wgrid, dim3(64), 0, st, N, D, q, logp_out, g_out);
    else if (D <= 512) hipLaunchKernelGGL(k_neal_funnel_v4<2>, wgrid, dim3(64), 0, st, N, D, q, logp_out, g_out);
    else hipLaunchKernelGGL(k_neal_funnel_v4<4>, wgrid, dim3(64), 0, st, N, D, q, logp_out, g_out);
  } else {
    hipLaunchKernelGGL(k_neal_funnel, grid, dim3(kBlock), 0, st, N, D, q, logp_out, g_out);
  }
  return bjx_check_launch("bjx_target_neal_funnel");
}

int bjx_target_ar1_gaussian(void* stream, int64_t N, int64_t D, float diag_edge, float diag_mid,
                            float off, const float* q, float* logp_out, float* g_out) {
  BJX_CHECK_ARG(N >= 0 && D >= 2, "bjx_target_ar1_gaussian: bad arguments");
  if (N == 0) return 0;
  BJX_CHECK_ARG(q && logp_out && g_out, "bjx_target_ar1_gaussian: bad arguments");
  hipLaunchKernelGGL(k_ar1_gaussian, dim3(bjx_row_grid(N, kWavesPerBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, N, D, diag_edge, diag_mid, off, q, logp_out, g_out);
  return bjx_check_launch("bjx_target_ar1_gaussian");
}

}  // extern "C"
