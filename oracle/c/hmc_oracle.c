/*
 * CPU port of the oracle's HMC transition for diagonal-Gaussian targets
 * (TEST INFRASTRUCTURE -- see oracle/__init__.py; never linked into the product).
 *
 * Same arithmetic as oracle/hmc.py::kernel with oracle/targets.py::diag_gaussian,
 * written in plain C + OpenMP so bench.py's `cpu_baseline` leg can time the
 * reference algorithm on all host cores ("kind": "port").  It is validated
 * bit-for-bit against the NumPy oracle in tests/test_oracle_c.py.
 *
 * Reference lines restated (via oracle/hmc.py): blackjax/mcmc/hmc.py:153-176,279-312;
 * integrators.py:104-150; metrics.py:260-270,704-709; proposal.py:45-48,214-235;
 * jax.random (threefry2x32, partitionable layout) as documented in oracle/prng.py.
 *
 * Numerics contract: compiled with -ffp-contract=off, explicit fmaf(), fp64-accumulated
 * reductions, fp64 transcendentals rounded once to fp32.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { uint32_t k0, k1; } key_t2;

static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

static inline key_t2 threefry2x32(key_t2 key, uint32_t x0, uint32_t x1) {
  static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  const uint32_t ks[3] = {key.k0, key.k1, key.k0 ^ key.k1 ^ 0x1BD11BDAu};
  x0 += ks[0];
  x1 += ks[1];
  for (int i = 0; i < 5; ++i) {
    for (int j = 0; j < 4; ++j) {
      x0 += x1;
      x1 = rotl32(x1, R[i & 1][j]);
      x1 ^= x0;
    }
    x0 += ks[(i + 1) % 3];
    x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
  }
  key_t2 o = {x0, x1};
  return o;
}

static inline key_t2 key_child(key_t2 k, uint64_t i) {
  return threefry2x32(k, (uint32_t)(i >> 32), (uint32_t)i);
}
static inline uint32_t key_bits32(key_t2 k, uint64_t i) {
  key_t2 o = key_child(k, i);
  return o.k0 ^ o.k1;
}
static inline float unit_float(uint32_t bits) {
  uint32_t fb = (bits >> 9) | 0x3F800000u;
  float f;
  memcpy(&f, &fb, 4);
  return f - 1.0f;
}

static inline float erfinv_f32(float x) {
  float t = -(x * x);
  float w = -(float)log1p((double)t);
  float p;
  if (w < 5.0f) {
    w = w - 2.5f;
    p = 2.81022636e-08f;
    p = fmaf(p, w, 3.43273939e-07f);
    p = fmaf(p, w, -3.5233877e-06f);
    p = fmaf(p, w, -4.39150654e-06f);
    p = fmaf(p, w, 0.00021858087f);
    p = fmaf(p, w, -0.00125372503f);
    p = fmaf(p, w, -0.00417768164f);
    p = fmaf(p, w, 0.246640727f);
    p = fmaf(p, w, 1.50140941f);
  } else {
    w = sqrtf(w) - 3.0f;
    p = -0.000200214257f;
    p = fmaf(p, w, 0.000100950558f);
    p = fmaf(p, w, 0.00134934322f);
    p = fmaf(p, w, -0.00367342844f);
    p = fmaf(p, w, 0.00573950773f);
    p = fmaf(p, w, -0.0076224613f);
    p = fmaf(p, w, 0.00943887047f);
    p = fmaf(p, w, 1.00167406f);
    p = fmaf(p, w, 2.83297682f);
  }
  float r = p * x;
  if (fabsf(x) == 1.0f) r = x * INFINITY;
  return r;
}

static inline float normal_from_bits(uint32_t bits) {
  const float lo = -0.99999994f;
  float u = fmaxf(lo, fmaf(unit_float(bits), 2.0f, lo));
  return 1.41421354f * erfinv_f32(u);
}

/* g = -(q*iv) ; logp = 0.5 * sum q*g (fp64) */
static inline float target_diag_gaussian(int64_t D, const float* iv, const float* q, float* g) {
  double acc = 0.0;
  for (int64_t j = 0; j < D; ++j) {
    const float gj = -(q[j] * iv[j]);
    g[j] = gj;
    acc += (double)q[j] * (double)gj;
  }
  return (float)(0.5 * acc);
}

/* One HMC transition of ONE chain (in place on q0/g0/logp): the body shared by the two entry
 * points below.  imm is that chain's (D,) inverse mass diagonal, eps its step size, kc its key. */
static inline void hmc_chain(key_t2 kc, int64_t D, int L, float eps, const float* imm,
                             const float* inv_var, float thr, float* q0, float* logp, float* g0,
                             float* acc_rate, uint8_t* is_acc, uint8_t* is_div, float* qw, float* pw,
                             float* gw) {
  const float h = eps * 0.5f;
  const key_t2 km = key_child(kc, 0), ki = key_child(kc, 1);
  double acc = 0.0;
  for (int64_t j = 0; j < D; ++j) {
    const float z = normal_from_bits(key_bits32(km, (uint64_t)j));
    const float ms = 1.0f / sqrtf(imm[j]);
    const float p = ms * z;
    pw[j] = p;
    acc += (double)(imm[j] * p) * (double)p;
    qw[j] = q0[j];
    gw[j] = g0[j];
  }
  const float ke0 = 0.5f * (float)acc;
  float lp = *logp;
  for (int l = 0; l < L; ++l) {
    for (int64_t j = 0; j < D; ++j) {
      const float pn = fmaf(h, gw[j], pw[j]);
      pw[j] = pn;
      qw[j] = fmaf(eps, imm[j] * pn, qw[j]);
    }
    lp = target_diag_gaussian(D, inv_var, qw, gw);
    for (int64_t j = 0; j < D; ++j) pw[j] = fmaf(h, gw[j], pw[j]);
  }
  acc = 0.0;
  for (int64_t j = 0; j < D; ++j) acc += (double)(imm[j] * pw[j]) * (double)pw[j];
  const float ke1 = 0.5f * (float)acc;
  const float H0 = -(*logp) + ke0;
  const float H1 = -lp + ke1;
  float delta = H0 - H1;
  if (delta != delta) delta = -INFINITY;
  const int div = (-delta) > thr;
  const float p_acc = fminf((float)exp((double)delta), 1.0f);
  const float u = fmaxf(0.0f, unit_float(key_bits32(ki, 0)));
  const int accept = u < p_acc;
  if (accept) {
    memcpy(q0, qw, sizeof(float) * D);
    memcpy(g0, gw, sizeof(float) * D);
    *logp = lp;
  }
  *acc_rate = p_acc;
  *is_acc = (uint8_t)accept;
  *is_div = (uint8_t)div;
}

/* One HMC transition for N chains (in place on q/logp/g), shared step size and inverse mass
 * diagonal, chain r keyed by split(key, .)[r + chain_offset].  Returns 0. */
int bjx_oracle_hmc_diag_gaussian(uint32_t key0, uint32_t key1, int64_t chain_offset, int64_t N,
                                 int64_t D, int L, float eps, const float* imm,
                                 const float* inv_var, float thr, float* q, float* logp, float* g,
                                 float* acc_rate, uint8_t* is_acc, uint8_t* is_div, int nthreads) {
  const key_t2 key = {key0, key1};
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    float* qw = (float*)malloc(sizeof(float) * D);
    float* pw = (float*)malloc(sizeof(float) * D);
    float* gw = (float*)malloc(sizeof(float) * D);
#pragma omp for schedule(static)
    for (int64_t r = 0; r < N; ++r) {
      const key_t2 kc = key_child(key, (uint64_t)(r + chain_offset));
      hmc_chain(kc, D, L, eps, imm, inv_var, thr, q + r * D, logp + r, g + r * D, acc_rate + r,
                is_acc + r, is_div + r, qw, pw, gw);
    }
    free(qw);
    free(pw);
    free(gw);
  }
  return 0;
}

/* The same transition with everything PER CHAIN: explicit chain keys (N, 2) -- any subset of the
 * global chain indices, either key layout --, step sizes (N,) and inverse mass diagonals with row
 * stride imm_stride (0 = one shared (D,) vector).  What a vmapped window_adaptation runs. */
int bjx_oracle_hmc_diag_gaussian_pc(const uint32_t* chain_keys, int64_t N, int64_t D, int L,
                                    const float* eps, const float* imm, int64_t imm_stride,
                                    const float* inv_var, float thr, float* q, float* logp, float* g,
                                    float* acc_rate, uint8_t* is_acc, uint8_t* is_div, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    float* qw = (float*)malloc(sizeof(float) * D);
    float* pw = (float*)malloc(sizeof(float) * D);
    float* gw = (float*)malloc(sizeof(float) * D);
#pragma omp for schedule(static)
    for (int64_t r = 0; r < N; ++r) {
      const key_t2 kc = {chain_keys[2 * r], chain_keys[2 * r + 1]};
      hmc_chain(kc, D, L, eps[r], imm + r * imm_stride, inv_var, thr, q + r * D, logp + r,
                g + r * D, acc_rate + r, is_acc + r, is_div + r, qw, pw, gw);
    }
    free(qw);
    free(pw);
    free(gw);
  }
  return 0;
}

/* fp32 GEMM as a k-ordered fmaf chain: C[m][n] = chain over k (in the order k_order[0..K-1]) of
 * acc = fmaf(A[m][k], B[k*sbk + n*sbn], acc), acc starting at +0.  This is what a sequence of
 * v_mfma_f32_32x32x2_f32 instructions computes bit for bit (one rounding per product-accumulate,
 * no wider internal accumulation) and what a "precision=highest" fp32 dot computes for that
 * summation order (blackjax/util.py:23-61).  Used by the oracle's dense-metric "f32 chain" mode. */
int bjx_oracle_gemm_f32chain(int64_t M, int64_t K, int64_t Nn, const float* A, const float* B,
                             int64_t sbk, int64_t sbn, const int32_t* k_order, float* C,
                             int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m) {
    const float* a = A + m * K;
    for (int64_t n = 0; n < Nn; ++n) {
      float acc = 0.0f;
      for (int64_t i = 0; i < K; ++i) {
        const int64_t k = k_order[i];
        acc = fmaf(a[k], B[k * sbk + n * sbn], acc);
      }
      C[m * Nn + n] = acc;
    }
  }
  return 0;
}

int bjx_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
