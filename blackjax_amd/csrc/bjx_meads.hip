// MEADS fold statistics (gfx950): the per-step, per-fold quantities of
// blackjax/adaptation/meads_adaptation.py:560-640, 790-817 as a handful of stream-ordered launches
// instead of ~70 small library operations (round 2 measured a MEADS step host-bound at 1.26 ms for
// 4 096 x 1 024).  C ABI in include/bjx_ghmc.h ("MEADS fold statistics").
//
//   chains are laid out fold-major: fold k owns rows [k n, (k + 1) n) of the (N, D) state arrays.
//   1. bjx_meads_fold_moments   per fold and column: mean and population std of the positions
//                               (fp64 sums of x - x_first and its square, K x R x column blocks
//                               workgroups, partials combined in a fixed order), then the fp64 mean of
//                               the WHITENED positions x / sd (what fold_damping centres with);
//   2. bjx_meads_fold_build     A = g * sd_k (preconditioned gradients), B = x / sd_k - mean_k(x / sd)
//                               (whitened, centred positions), both fp32 as the reference forms them,
//                               plus the fp64 row sums of squares (diag of A A^T, B B^T);
//   3. (caller) the two Gram matrices per fold with the library's batched fp32 GEMM -- a plain GEMM,
//      D x D or n x n whichever is smaller (|X X^T|_F = |X^T X|_F);
//   4. bjx_meads_frob           sum of squares of each Gram matrix, fp64;
//   5. bjx_meads_fold_params    maximum_eigenvalue (790-817), step size = min(mult / sqrt(lambda), 1)
//                               rolled to the right-hand neighbour fold (583-598), damping alpha / delta
//                               (604-614, fp64 exp rounded once), broadcast to per-chain arrays.
// Reductions accumulate in fp64 and are rounded once (the engine's numerics contract).
#include <math.h>

#include "../../include/bjx_ghmc.h"
#include "bjx_device.h"
#include "bjx_host.h"

using namespace bjx;

namespace {

constexpr int kSplits = 64;  // row splits of a fold per column block (16 left one wave per SIMD: 0.9 TB/s)

// partial[(k * kSplits + r) * 2 + s][c]: s = 0 sum, s = 1 sum of squares of (v - v_first) over the
// rows of split r of fold k, v = x (WHITEN = false) or x / sd_k (WHITEN = true)
template <bool WHITEN>
__global__ void __launch_bounds__(256)
k_meads_moments_partial(int64_t n, int64_t D, const float* __restrict__ x, const float* __restrict__ sd,
                        double* __restrict__ partial) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y, k = blockIdx.z;
  if (c >= D) return;
  const int64_t rows_per = (n + kSplits - 1) / kSplits;
  const int64_t r0 = (int64_t)r * rows_per, r1 = r0 + rows_per < n ? r0 + rows_per : n;
  const float* xf = x + (int64_t)k * n * D;
  float s = 1.0f;
  if constexpr (WHITEN) s = sd[(int64_t)k * D + c];
  const float first = WHITEN ? xf[c] / s : xf[c];  // shift: no cancellation whatever the fold's mean
  double s1 = 0.0, s2 = 0.0;
  for (int64_t i = r0; i < r1; ++i) {
    float v = xf[i * D + c];
    if constexpr (WHITEN) v = v / s;
    const double d = (double)v - (double)first;
    s1 += d;
    s2 += d * d;
  }
  double* out = partial + ((int64_t)(k * kSplits + r) * 2) * D;
  out[c] = s1;
  out[D + c] = s2;
}

// MODE 0: mean / population std of x per fold -> mean_out, sd_out;  MODE 1: mean of x / sd -> mean_out
template <int MODE>
__global__ void __launch_bounds__(256)
k_meads_moments_final(int64_t n, int64_t D, int K, const float* __restrict__ x, const float* __restrict__ sd_in,
                      const double* __restrict__ partial, float* __restrict__ mean_out,
                      float* __restrict__ sd_out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  if (c >= D || k >= K) return;
  double s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < kSplits; ++r) {
    const double* p = partial + ((int64_t)(k * kSplits + r) * 2) * D;
    s1 += p[c];
    s2 += p[D + c];
  }
  const float x0 = x[(int64_t)k * n * D + c];
  const double first = MODE == 1 ? (double)(x0 / sd_in[(int64_t)k * D + c]) : (double)x0;
  const double m = s1 / (double)n;
  mean_out[(int64_t)k * D + c] = (float)(first + m);
  if constexpr (MODE == 0) {
    double var = s2 / (double)n - m * m;  // jnp.std: ddof = 0
    if (var < 0.0) var = 0.0;
    sd_out[(int64_t)k * D + c] = (float)sqrt(var);
  }
}

// One wave per chain row: A = g * sd_k ; B = x / sd_k - mw_k ; fp64 row sums of squares.
__global__ void __launch_bounds__(256)
k_meads_build(int64_t N, int64_t n, int64_t D, const float* __restrict__ x, const float* __restrict__ g,
              const float* __restrict__ sd, const float* __restrict__ mw, float* __restrict__ A,
              float* __restrict__ B, double* __restrict__ rowsq) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const int64_t k = row / n, base = row * D;
  const float* s = sd + k * D;
  const float* m = mw + k * D;
  double qa = 0.0, qb = 0.0;
  for (int64_t j = lane; j < D; j += 64) {
    const float sj = s[j];
    const float a = g[base + j] * sj;
    const float b = x[base + j] / sj - m[j];
    A[base + j] = a;
    B[base + j] = b;
    qa += (double)a * (double)a;
    qb += (double)b * (double)b;
  }
  qa = wave_sum(qa);
  qb = wave_sum(qb);
  if (lane == 0) {
    rowsq[row] = qa;
    rowsq[N + row] = qb;
  }
}

// sum of squares of M floats per matrix, fp64: kFrobBlocks partials per matrix, combined in order
constexpr int kFrobBlocks = 256;
__global__ void __launch_bounds__(256)
k_meads_frob_partial(int64_t M, const float* __restrict__ G, double* __restrict__ partial) {
  __shared__ double sm[4];
  const int mtx = blockIdx.y;
  const float* gm = G + (int64_t)mtx * M;
  double s = 0.0;
  // a block sums a contiguous range so that the result does not depend on the launch geometry
  const int64_t per = (((M + kFrobBlocks - 1) / kFrobBlocks) + 3) & ~(int64_t)3;  // 16-byte pieces
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < M ? lo + per : M;
  const bool vec = ((M & 3) == 0) && ((reinterpret_cast<uintptr_t>(G) & 15) == 0);
  if (vec) {
    for (int64_t i = lo + (int64_t)threadIdx.x * 4; i < hi; i += 1024) {  // hi - lo is a multiple of 4
      const F4 v = ld4(gm + i);
      s += (double)v.x * (double)v.x;
      s += (double)v.y * (double)v.y;
      s += (double)v.z * (double)v.z;
      s += (double)v.w * (double)v.w;
    }
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
      const double v = (double)gm[i];
      s += v * v;
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[(int64_t)mtx * kFrobBlocks + blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// maximum_eigenvalue of one (matrix, fold) pair per workgroup (meads_adaptation.py:790-817): lam_out[mtx * K + k].
// frob_partial: (2, K, kFrobBlocks) [matrix mtx * K + k: mtx 0 = A, 1 = B of fold k].
__global__ void __launch_bounds__(256)
k_meads_lambda(int K, int64_t n, const double* __restrict__ frob_partial, const double* __restrict__ rowsq,
               float* __restrict__ lam_out) {
  __shared__ double red[3][4];
  const int k = blockIdx.x, mtx = blockIdx.y;
  const int64_t N = (int64_t)K * n;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // sum diag(S) and sum diag(S)^2 over the fold's rows, and the Frobenius partials (fixed order: thread
  // strided, wave tree, 4 waves)
  double d1 = 0.0, d2 = 0.0, fr = 0.0;
  const double* rs = rowsq + (int64_t)mtx * N + (int64_t)k * n;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const double v = rs[i];
    d1 += v;
    d2 += v * v;
  }
  const double* fp = frob_partial + ((int64_t)mtx * K + k) * kFrobBlocks;
  for (int b = threadIdx.x; b < kFrobBlocks; b += 256) fr += fp[b];
  d1 = wave_sum(d1);
  d2 = wave_sum(d2);
  fr = wave_sum(fr);
  if (lane == 0) {
    red[0][wv] = d1;
    red[1][wv] = d2;
    red[2][wv] = fr;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double sd1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double sd2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double frs = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    // lamda = sum diag / n ; lamda_sq = (sum S^2 - sum diag^2) / (n (n - 1))
    const double lam = sd1 / (double)n;
    const double lam_sq = (frs - sd2) / ((double)n * (double)(n - 1));
    lam_out[mtx * K + k] = (float)(lam_sq / lam);
  }
}

// One workgroup: the parameter table from the per-fold eigenvalue estimates and its per-chain broadcast.
__global__ void __launch_bounds__(256)
k_meads_params(int K, int64_t n, int64_t D, int64_t t, float multiplier, float slowdown,
               const float* __restrict__ lam_in, const float* __restrict__ sd, float* __restrict__ eps_fold,
               float* __restrict__ alpha_fold, float* __restrict__ delta_fold, float* __restrict__ sigma_fold,
               float* __restrict__ eps_pc, float* __restrict__ alpha_pc, float* __restrict__ delta_pc,
               float* __restrict__ scale_pc) {
  __shared__ float lam_max[2][64];  // [matrix][fold]
  __shared__ float eps_own[64], eps_r[64], al[64];
  const int64_t N = (int64_t)K * n;
  if (threadIdx.x < 2 * K) lam_max[threadIdx.x / K][threadIdx.x % K] = lam_in[threadIdx.x];
  __syncthreads();
  if (threadIdx.x < K) {
    const int k = threadIdx.x;
    const float e = multiplier / sqrtf(lam_max[0][k]);  // 583-588
    eps_own[k] = fminf(e, 1.0f);
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const int k = threadIdx.x;
    const float eps = eps_own[(k + K - 1) % K];  // fold k takes the step size of fold k - 1 (590-598)
    eps_r[k] = eps;
    const float g1 = 1.0f / sqrtf(lam_max[1][k]);
    const float g2 = slowdown / ((float)(t + 1) * eps);
    const float gamma = fmaxf(g1, g2);
    const float arg = (-2.0f * eps) * gamma;
    const float a = 1.0f - (float)exp((double)arg);
    al[k] = a;
    eps_fold[k] = eps;
    alpha_fold[k] = a;
    delta_fold[k] = a / 2.0f;
  }
  __syncthreads();
  // rolled scales (K, D) and the per-chain broadcasts (jnp.repeat, 636-640)
  for (int64_t i = threadIdx.x; i < (int64_t)K * D; i += 256) {
    const int64_t k = i / D, j = i - k * D;
    sigma_fold[i] = sd[((k + K - 1) % K) * D + j];
  }
  // the per-chain broadcasts of (eps, alpha, delta) ride along in k_meads_scale_rows (one wave per chain)
  (void)N;
  (void)eps_pc;
  (void)alpha_pc;
  (void)delta_pc;
  (void)scale_pc;
}

// scale_pc[row] = sd[(fold(row) - 1) mod K]: the per-chain inverse scale GHMC squares (one wave per row)
__global__ void __launch_bounds__(256)
k_meads_scale_rows(int64_t N, int64_t n, int64_t D, int K, const float* __restrict__ sd,
                   float* __restrict__ scale_pc, float* __restrict__ imm_pc, const float* __restrict__ eps_fold,
                   const float* __restrict__ alpha_fold, const float* __restrict__ delta_fold,
                   float* __restrict__ eps_pc, float* __restrict__ alpha_pc, float* __restrict__ delta_pc) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const int64_t k = row / n;
  if (lane == 0) {  // jnp.repeat of the fold table (meads_adaptation.py:636-640)
    eps_pc[row] = eps_fold[k];
    alpha_pc[row] = alpha_fold[k];
    delta_pc[row] = delta_fold[k];
  }
  const float* s = sd + ((k + K - 1) % K) * D;
  for (int64_t j = lane; j < D; j += 64) {
    const float v = s[j];
    if (scale_pc) scale_pc[row * D + j] = v;
    imm_pc[row * D + j] = v * v;  // ghmc.py:67-86 legacy branch: inverse mass = scale ** 2
  }
}

}  // namespace

extern "C" {

size_t bjx_meads_workspace_bytes(int64_t K, int64_t D) {
  const size_t moments = (size_t)K * kSplits * 2 * (size_t)D * sizeof(double);
  const size_t frob = (size_t)2 * K * kFrobBlocks * sizeof(double);
  return moments + frob + 2 * (size_t)K * sizeof(float) + 256;
}

int bjx_meads_fold_moments(void* stream, int64_t K, int64_t n, int64_t D, const float* x, void* workspace,
                           float* mean_out, float* sd_out, float* whitened_mean_out) {
  BJX_CHECK_ARG(K >= 1 && K <= 64 && n >= 2 && D >= 1, "bjx_meads_fold_moments: need 1 <= K <= 64, n >= 2, D >= 1");
  BJX_CHECK_ARG(x && workspace && mean_out && sd_out && whitened_mean_out, "bjx_meads_fold_moments: null pointer");
  hipStream_t s = (hipStream_t)stream;
  double* partial = (double*)workspace;
  const dim3 pg((unsigned)((D + 255) / 256), kSplits, (unsigned)K), fg((unsigned)((D + 255) / 256), (unsigned)K);
  hipLaunchKernelGGL(k_meads_moments_partial<false>, pg, dim3(256), 0, s, n, D, x, (const float*)nullptr, partial);
  hipLaunchKernelGGL(k_meads_moments_final<0>, fg, dim3(256), 0, s, n, D, (int)K, x, (const float*)nullptr,
                     partial, mean_out, sd_out);
  hipLaunchKernelGGL(k_meads_moments_partial<true>, pg, dim3(256), 0, s, n, D, x, (const float*)sd_out, partial);
  hipLaunchKernelGGL(k_meads_moments_final<1>, fg, dim3(256), 0, s, n, D, (int)K, x, (const float*)sd_out, partial,
                     whitened_mean_out, (float*)nullptr);
  return bjx_check_launch("bjx_meads_fold_moments");
}

int bjx_meads_fold_build(void* stream, int64_t K, int64_t n, int64_t D, const float* x, const float* g,
                         const float* sd, const float* whitened_mean, float* A, float* B, double* rowsq) {
  BJX_CHECK_ARG(K >= 1 && n >= 1 && D >= 1 && x && g && sd && whitened_mean && A && B && rowsq,
                "bjx_meads_fold_build: bad arguments");
  const int64_t N = K * n;
  hipLaunchKernelGGL(k_meads_build, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, N, n, D, x,
                     g, sd, whitened_mean, A, B, rowsq);
  return bjx_check_launch("bjx_meads_fold_build");
}

int bjx_meads_fold_params(void* stream, int64_t K, int64_t n, int64_t D, int64_t t, float step_size_multiplier,
                          float damping_slowdown, int64_t gram_elems, const float* gram, const double* rowsq,
                          const float* sd, void* workspace, float* eps_fold, float* alpha_fold, float* delta_fold,
                          float* sigma_fold, float* eps_pc, float* alpha_pc, float* delta_pc, float* imm_pc) {
  BJX_CHECK_ARG(K >= 1 && K <= 64 && n >= 2 && D >= 1 && t >= 0 && gram_elems >= 1,
                "bjx_meads_fold_params: bad sizes");
  BJX_CHECK_ARG(gram && rowsq && sd && workspace && eps_fold && alpha_fold && delta_fold && sigma_fold && eps_pc &&
                    alpha_pc && delta_pc && imm_pc,
                "bjx_meads_fold_params: null pointer");
  hipStream_t s = (hipStream_t)stream;
  double* frob = (double*)((char*)workspace + (size_t)K * kSplits * 2 * (size_t)D * sizeof(double));
  hipLaunchKernelGGL(k_meads_frob_partial, dim3(kFrobBlocks, (unsigned)(2 * K)), dim3(256), 0, s, gram_elems, gram,
                     frob);
  float* lam = (float*)(frob + (size_t)2 * K * kFrobBlocks);
  hipLaunchKernelGGL(k_meads_lambda, dim3((unsigned)K, 2), dim3(256), 0, s, (int)K, n, (const double*)frob, rowsq, lam);
  hipLaunchKernelGGL(k_meads_params, dim3(1), dim3(256), 0, s, (int)K, n, D, t, step_size_multiplier,
                     damping_slowdown, (const float*)lam, sd, eps_fold, alpha_fold, delta_fold, sigma_fold,
                     eps_pc, alpha_pc, delta_pc, (float*)nullptr);
  const int64_t N = K * n;
  hipLaunchKernelGGL(k_meads_scale_rows, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, N, n, D, (int)K, sd,
                     (float*)nullptr, imm_pc, (const float*)eps_fold, (const float*)alpha_fold,
                     (const float*)delta_fold, eps_pc, alpha_pc, delta_pc);
  return bjx_check_launch("bjx_meads_fold_params");
}

}  // extern "C"
