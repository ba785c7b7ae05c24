// Window-adaptation kernels (gfx950): per-chain dual averaging, Welford accumulation and
// the window-end inverse-mass-matrix blend.  C ABI in include/bjx_hip.h.
//
// Scalar convention (oracle/adaptation.py): the dual-averaging recursion is evaluated
// operation by operation in fp32 without fusing; pow/exp/log are fp64 rounded once.
#include <math.h>

#include "../../include/bjx_hip.h"
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

__device__ __forceinline__ float log_cr(float x) { return (float)log((double)x); }

// dual_averaging.py:87-99 ; from_log_avg: x = exp(log_x_avg) first (slow_final re-init,
// staged_adaptation.py:242-243)
__global__ void __launch_bounds__(kBlock)
k_da_init(int64_t N, int from_log_avg, const float* __restrict__ x_in, float* __restrict__ log_x,
          float* __restrict__ log_x_avg, float* __restrict__ avg_err, float* __restrict__ mu,
          float* __restrict__ step_size) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float x = x_in[i];
  if (from_log_avg) x = exp_cr(x);
  const float lx = log_cr(x);
  mu[i] = log_cr(10.0f * x);
  log_x[i] = lx;
  log_x_avg[i] = 0.0f;
  avg_err[i] = 0.0f;
  step_size[i] = exp_cr(lx);
}

// dual_averaging.py:101-123 with gradient = target - acceptance_rate (step_size.py:144)
__global__ void __launch_bounds__(kBlock)
k_da_update(int64_t N, float reg, float inv_reg, float eta, float coef, float target,
            const float* __restrict__ acc_rate, const float* log_x_in, const float* log_x_avg_in,
            const float* avg_err_in, const float* __restrict__ mu, float* log_x_out,
            float* log_x_avg_out, float* avg_err_out, float* __restrict__ step_size) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float g = target - acc_rate[i];
  const float ae = (1.0f - inv_reg) * avg_err_in[i] + g / reg;
  const float lx_prev = log_x_in[i];
  const float lx = mu[i] - coef * ae;
  const float lxa = eta * lx_prev + (1.0f - eta) * log_x_avg_in[i];
  avg_err_out[i] = ae;
  log_x_out[i] = lx;
  log_x_avg_out[i] = lxa;
  step_size[i] = exp_cr(lx);
}

__global__ void __launch_bounds__(kBlock)
k_exp(int64_t N, const float* __restrict__ x, float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) y[i] = exp_cr(x[i]);
}

// mass_matrix.py:410-435 (diagonal): delta = x - mean ; mean += delta/n ; m2 = fma(delta, x - mean, m2)
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_welford_update_diag(int64_t N, int64_t D, float n, const float* __restrict__ x,
                      const float* mean_in, const float* m2_in, float* mean_out, float* m2_out) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    const int64_t base = r * D;
    if constexpr (VEC == 4) {
      F4 xs[4], ms[4], ss[4];
      row_sweep4<4>(
          lane, D,
          [&](int u, int64_t j) {
            xs[u] = ld4(x + base + j);
            ms[u] = ld4(mean_in + base + j);
            ss[u] = ld4(m2_in + base + j);
          },
          [&](int u, int64_t j) {
            const F4 xv = xs[u], mv = ms[u], sv = ss[u];
            F4 mo, so;
#define BJX_W(c)                                   \
  {                                                \
    const float d = xv.c - mv.c;                   \
    mo.c = mv.c + d / n;                           \
    so.c = fmaf(d, xv.c - mo.c, sv.c);             \
  }
            BJX_W(x) BJX_W(y) BJX_W(z) BJX_W(w)
#undef BJX_W
            st4(mean_out + base + j, mo);
            st4(m2_out + base + j, so);
          });
    } else {
      for (int64_t j = lane; j < D; j += 64) {
        const float xv = x[base + j], mv = mean_in[base + j];
        const float d = xv - mv;
        const float mo = mv + d / n;
        mean_out[base + j] = mo;
        m2_out[base + j] = fmaf(d, xv - mo, m2_in[base + j]);
      }
    }
  }
}

// the same update over the flattened arrays (purely element-wise): used when a row does not fill
// whole waves, where one wave per row would idle most lanes
__global__ void __launch_bounds__(kBlock)
k_welford_update_flat(int64_t total4, float n, const float* __restrict__ x, const float* mean_in,
                      const float* m2_in, float* mean_out, float* m2_out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const F4 xv = ld4(x + 4 * i), mv = ld4(mean_in + 4 * i), sv = ld4(m2_in + 4 * i);
    F4 mo, so;
#define BJX_W(c)                                   \
  {                                                \
    const float d = xv.c - mv.c;                   \
    mo.c = mv.c + d / n;                           \
    so.c = fmaf(d, xv.c - mo.c, sv.c);             \
  }
    BJX_W(x) BJX_W(y) BJX_W(z) BJX_W(w)
#undef BJX_W
    st4(mean_out + 4 * i, mo);
    st4(m2_out + 4 * i, so);
  }
}

// mass_matrix.py:335-357 (diagonal): cov = m2/(n-1) ; imm = fma(beta_prev, prev, beta_data*cov) + reg
__global__ void __launch_bounds__(kBlock)
k_welford_final_diag(int64_t total, int64_t D, float nm1, float beta_data, float beta_prev,
                     float reg, const float* __restrict__ m2, const float* __restrict__ prev,
                     int64_t prev_stride, float* __restrict__ imm_out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float cov = m2[i] / nm1;
    const float pv = prev_stride ? prev[i] : prev[i % D];
    imm_out[i] = fmaf(beta_prev, pv, beta_data * cov) + reg;
  }
}

inline unsigned flat_grid(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

}  // namespace

extern "C" {

int bjx_da_init(void* stream, int64_t N, int from_log_avg, const float* x_in, float* log_x_out,
                float* log_x_avg_out, float* avg_error_out, float* mu_out, float* step_size_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && x_in && log_x_out && log_x_avg_out && avg_error_out && mu_out &&
                    step_size_out,
                "bjx_da_init: bad arguments");
  hipLaunchKernelGGL(k_da_init, dim3(flat_grid(N)), dim3(kBlock), 0, (hipStream_t)stream, N,
                     from_log_avg, x_in, log_x_out, log_x_avg_out, avg_error_out, mu_out,
                     step_size_out);
  return bjx_check_launch("bjx_da_init");
}

int bjx_da_update(void* stream, int64_t N, int64_t step, float target, float t0, float gamma,
                  float kappa, const float* acceptance_rate, const float* log_x_in,
                  const float* log_x_avg_in, const float* avg_error_in, const float* mu,
                  float* log_x_out, float* log_x_avg_out, float* avg_error_out,
                  float* step_size_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && step >= 1 && acceptance_rate && log_x_in && log_x_avg_in &&
                    avg_error_in && mu && log_x_out && log_x_avg_out && avg_error_out &&
                    step_size_out,
                "bjx_da_update: bad arguments");
  // wave-uniform scalars of dual_averaging.py:117-122, evaluated once on the host
  const float reg = (float)step + t0;
  const float inv_reg = 1.0f / reg;
  const float eta = (float)pow((double)step, -(double)kappa);
  const float coef = sqrtf((float)step) / gamma;
  hipLaunchKernelGGL(k_da_update, dim3(flat_grid(N)), dim3(kBlock), 0, (hipStream_t)stream, N,
                     reg, inv_reg, eta, coef, target, acceptance_rate, log_x_in, log_x_avg_in,
                     avg_error_in, mu, log_x_out, log_x_avg_out, avg_error_out, step_size_out);
  return bjx_check_launch("bjx_da_update");
}

int bjx_exp(void* stream, int64_t N, const float* x, float* y) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && x && y, "bjx_exp: bad arguments");
  hipLaunchKernelGGL(k_exp, dim3(flat_grid(N)), dim3(kBlock), 0, (hipStream_t)stream, N, x, y);
  return bjx_check_launch("bjx_exp");
}

int bjx_welford_update_diag(void* stream, int64_t N, int64_t D, int64_t sample_size_new,
                            const float* value, const float* mean_in, const float* m2_in,
                            float* mean_out, float* m2_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && sample_size_new >= 1 && value && mean_in && m2_in && mean_out &&
                    m2_out,
                "bjx_welford_update_diag: bad arguments");
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  const float n = (float)sample_size_new;
  if (bjx_vec4_ok(D, value, mean_in, m2_in, mean_out, m2_out) && (D / 4) % 64 != 0) {
    const int64_t total4 = N * (D / 4);
    int64_t blocks = (total4 + kBlock - 1) / kBlock;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(k_welford_update_flat, dim3((unsigned)blocks), block, 0, (hipStream_t)stream, total4,
                       n, value, mean_in, m2_in, mean_out, m2_out);
  } else if (bjx_vec4_ok(D, value, mean_in, m2_in, mean_out, m2_out))
    hipLaunchKernelGGL(k_welford_update_diag<4>, grid, block, 0, (hipStream_t)stream, N, D, n, value,
                       mean_in, m2_in, mean_out, m2_out);
  else
    hipLaunchKernelGGL(k_welford_update_diag<1>, grid, block, 0, (hipStream_t)stream, N, D, n, value,
                       mean_in, m2_in, mean_out, m2_out);
  return bjx_check_launch("bjx_welford_update_diag");
}

int bjx_welford_final_diag(void* stream, int64_t N, int64_t D, int64_t sample_size,
                           float imm_shrinkage_to_previous, const float* m2, const float* imm_prev,
                           int64_t imm_prev_stride, float* imm_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && sample_size >= 0 && m2 && imm_prev && imm_out,
                "bjx_welford_final_diag: bad arguments");
  BJX_CHECK_ARG(imm_prev_stride == 0 || imm_prev_stride == D,
                "bjx_welford_final_diag: imm_prev_stride must be 0 or D");
  // mass_matrix.py:339-343 scalars (fp32, as the reference evaluates them)
  const float denom = (float)(sample_size + 5) + imm_shrinkage_to_previous;
  const float beta_data = (float)sample_size / denom;
  const float beta_prev = imm_shrinkage_to_previous / denom;
  const float reg = (5.0f / denom) * 1e-3f;
  const int64_t total = N * D;
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(k_welford_final_diag, dim3((unsigned)blocks), dim3(kBlock), 0,
                     (hipStream_t)stream, total, D, (float)(sample_size - 1), beta_data, beta_prev,
                     reg, m2, imm_prev, imm_prev_stride, imm_out);
  return bjx_check_launch("bjx_welford_final_diag");
}

}  // extern "C"
