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

#ifndef BJX_REVERSE_ROWS
#define BJX_REVERSE_ROWS 1
#endif

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

// the key used AS IS (no per-chain child): z[j] = normal(key, (D,))[j], u = uniform(key, ()),
// children[i] = split(key, .)[i] -- the probe the doc-sourced jax.random pins are held against
__global__ void __launch_bounds__(kBlock) k_rng_key_probe(Key key, int64_t D, float* __restrict__ z,
                                                           float* __restrict__ u, int64_t n_children,
                                                           uint32_t* __restrict__ children) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D) z[i] = normal_from_bits(key_bits32(key, (uint64_t)i));
  if (i < n_children) {
    const Key c = key_child(key, (uint64_t)i);
    children[2 * i] = c.k0;
    children[2 * i + 1] = c.k1;
  }
  if (i == 0 && u) u[0] = key_uniform(key);
}

// Exhaustive device check of bjx_log1p.h: every fp32 t in (-1, 0] (bit patterns 0x80000000 .. 0xBF7FFFFF, and +0.0)
// through the product's bjx_neg_log1p against the device library's fp64 log1p rounded once.  counts[0] = inputs
// checked, [1] = results that differ, [2] = inputs the fast path deferred to the table, [3] = first differing bits.
__global__ void __launch_bounds__(kBlock) k_log1p_check(uint32_t first, uint64_t n, uint32_t stride,
                                                         unsigned long long* __restrict__ counts) {
  unsigned long long bad = 0, deferred = 0, seen = 0;
  uint32_t first_bad = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t b = (uint64_t)first + i * stride;
    const uint32_t bits = b < 0xBF800000ull ? (uint32_t)b : 0u;  // past the last negative input: +0.0
    const float t = __uint_as_float(bits);
    float wf;
    if (!bjx_neg_log1p_fast(t, &wf)) ++deferred;
    const float got = bjx_neg_log1p(t);
    const float want = (float)(-log1p((double)t));
    ++seen;
    if (!(got == want)) {
      if (!bad) first_bad = bits;
      ++bad;
    }
  }
  atomicAdd(&counts[0], seen);
  if (bad) {
    atomicAdd(&counts[1], bad);
    atomicCAS(&counts[3], 0ull, (unsigned long long)first_bad);
  }
  if (deferred) atomicAdd(&counts[2], deferred);
}

#ifndef BJX_NORMAL4
#define BJX_NORMAL4 1
#endif
// ------------------------------------------------------------------------------ momentum draw
// p0 = (1/sqrt(imm)) * normal(km, (D,)) ; ke0 = 0.5 * sum (imm*p0)*p0   (fp64 accumulate)
// KICK: the opening half kick and the drift of the trajectory's first leapfrog in the same launch (the
// arithmetic of k_leapfrog_diag with n_kicks = 1) -- this kernel is bound by its per-element RNG
// arithmetic, so the leapfrog's words ride along.
// HOIST (round 4): ONE inverse mass matrix for all chains (imm_stride == 0), 16-byte rows of at most 1 024
// floats -- a lane meets the same <= 16 columns in every row it sweeps, so mass_sqrt = 1 / sqrt(imm) (an IEEE
// square root and an IEEE division: ~25 instructions per element, a tenth of this issue-bound kernel) is
// computed once per wave and the launch gives every wave several rows.  Same values, same results.
template <int VEC, bool KICK, bool HOIST = false>
__global__ void __launch_bounds__(kBlock)
k_momentum_diag(Key key, int64_t off, int64_t fold, int64_t N, int64_t D, const float* __restrict__ imm,
                int64_t imm_stride, float* __restrict__ p_out, float* __restrict__ ke_out, float eps_s,
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
    const Key km = key_child(kc, 0);  // split(kc, 2)[0]
    const float* im = imm + r * imm_stride;
    float* pr = p_out + r * D;
    const float eps = KICK ? (eps_pc ? eps_pc[r] : eps_s) : 0.0f;
    const float h = eps * 0.5f, ed = eps * 1.0f;
    double acc = 0.0;
    auto piece = [&](int64_t j, const float* msv) {  // msv: this piece's four hoisted mass_sqrt values, or null
      float m[VEC], pv[VEC];
      if constexpr (VEC == 4) {
        F4 t = ld4(im + j);
        m[0] = t.x; m[1] = t.y; m[2] = t.z; m[3] = t.w;
      } else {
        m[0] = im[j];
      }
      // KICK: the leapfrog's operands are requested BEFORE the ~2 000 cycles of RNG arithmetic of this piece, so their
      // HBM round trip runs under it (round 6: requested after it, the kernel took the sum of its arithmetic and its
      // 20 bytes per element instead of the larger of the two)
      [[maybe_unused]] float gg[VEC], qq[VEC];
      if constexpr (KICK) {
        if constexpr (VEC == 4) {
          const F4 a = ld4(g0 + r * D + j), b = ld4(q0 + r * D + j);
          gg[0] = a.x; gg[1] = a.y; gg[2] = a.z; gg[3] = a.w;
          qq[0] = b.x; qq[1] = b.y; qq[2] = b.z; qq[3] = b.w;
        } else {
          gg[0] = g0[r * D + j];
          qq[0] = q0[r * D + j];
        }
      }
      float zs[VEC];
      if constexpr (VEC == 4 && BJX_NORMAL4) {
        uint32_t bits[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bits[e] = key_bits32(km, (uint64_t)(j + e));
        normal4_from_bits(bits, zs);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) zs[e] = normal_from_bits(key_bits32(km, (uint64_t)(j + e)));
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float ms = msv ? msv[e] : 1.0f / sqrtf(m[e]);  // metrics.py:704-709 (two roundings)
        pv[e] = ms * zs[e];
        const float v = m[e] * pv[e];
        acc += (double)v * (double)pv[e];
      }
      if constexpr (VEC == 4) st4(pr + j, F4{pv[0], pv[1], pv[2], pv[3]});
      else pr[j] = pv[0];
      if constexpr (KICK) {
        float ph[VEC], qn[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          ph[e] = fmaf(h, gg[e], pv[e]);
          qn[e] = fmaf(ed, m[e] * ph[e], qq[e]);
        }
        if constexpr (VEC == 4) {
          st4(p_half_out + r * D + j, F4{ph[0], ph[1], ph[2], ph[3]});
          st4(q1_out + r * D + j, F4{qn[0], qn[1], qn[2], qn[3]});
        } else {
          p_half_out[r * D + j] = ph[0];
          q1_out[r * D + j] = qn[0];
        }
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
    if (lane == 0) ke_out[r] = 0.5f * (float)acc;
  }
}

// Short rows (D <= 128): G = 4 ... 32 lanes per row, 64 / G rows per wave, so a row of 64 floats
// does not leave 48 of the 64 lanes idle in this VALU-bound kernel (threefry + erf_inv per element).
// Same per-element arithmetic and the same per-lane accumulation order as k_momentum_diag<4>; the
// fp64 partial sums are combined by a butterfly inside the lane group.
template <int G>
__global__ void __launch_bounds__(kBlock)
k_momentum_diag_short(Key key, int64_t off, int64_t fold, int64_t N, int64_t D, const float* __restrict__ imm,
                      int64_t imm_stride, float* __restrict__ p_out, float* __restrict__ ke_out) {
  constexpr int R = BJX_WAVE / G;
  const int lane = threadIdx.x & 63;
  const int sub = lane / G, gl = lane % G;
  for (int64_t r0 = wave_row0() * R; r0 < N; r0 += wave_row_stride() * R) {
    const int64_t r = r0 + sub;
    const bool valid = r < N;
    double acc = 0.0;
    if (valid) {
      const Key kc = chain_key(key, (uint64_t)(r + off), fold);
      const Key km = key_child(kc, 0);  // split(kc, 2)[0]
      const float* im = imm + r * imm_stride;
      float* pr = p_out + r * D;
      for (int64_t j = (int64_t)gl * 4; j < D; j += G * 4) {
        const F4 t = ld4(im + j);
        const float m[4] = {t.x, t.y, t.z, t.w};
        float pv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = normal_from_bits(key_bits32(km, (uint64_t)(j + e)));
          const float ms = 1.0f / sqrtf(m[e]);  // metrics.py:704-709 (two roundings)
          pv[e] = ms * z;
          const float v = m[e] * pv[e];
          acc += (double)v * (double)pv[e];
        }
        st4(pr + j, F4{pv[0], pv[1], pv[2], pv[3]});
      }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, BJX_WAVE);
    if (valid && gl == 0) ke_out[r] = 0.5f * (float)acc;
  }
}

// ------------------------------------------------------------------------------ leapfrog
// p' = fma(h,g,p) [twice if KICKS==2] ; v = imm*p' ; q' = fma(eps, v, q)
template <int VEC, int KICKS>
__global__ void __launch_bounds__(kBlock)
k_leapfrog_diag(int64_t N, int64_t D, float eps_s, const float* __restrict__ eps_pc,
                const float* __restrict__ imm, int64_t imm_stride, const float* q_in,
                const float* p_in, const float* __restrict__ g, float* q_out, float* p_out,
                const int32_t* __restrict__ n_steps, int32_t step_idx, float kick_a, float kick_b,
                float drift) {
  // General palindromic-integrator stage (integrators.py:104-150): kick coefficients kick_a
  // [, kick_b] and drift coefficient `drift` multiply the step size exactly as the reference's
  // `step_size * coef` (fp32 product); velocity Verlet is (0.5, 0.5, 1.0).
  const int lane = threadIdx.x & 63;
  // Rows are swept LAST-TO-FIRST (workgroup 0 takes the last rows).  The user's callable, which
  // runs between two leapfrog launches, sweeps first-to-last and leaves the tail of q / g in the
  // 256 MiB Infinity Cache; starting there turns that part of this kernel's g/q reads into cache
  // hits, and this kernel in turn finishes at row 0, which is where the next callable starts
  // reading q.  Pure traversal order: results are unchanged.
  for (int64_t rr = wave_row0(); rr < N; rr += wave_row_stride()) {
    const int64_t r = BJX_REVERSE_ROWS ? N - 1 - rr : rr;
    const int64_t base = r * D;
    if (n_steps && step_idx >= n_steps[r]) {
      // dynamic HMC: this chain's trajectory is already complete -- leave its state untouched
      // (copy it through when the launch is out of place)
      if (q_out != q_in)
        for (int64_t j = lane; j < D; j += 64) {
          q_out[base + j] = q_in[base + j];
          p_out[base + j] = p_in[base + j];
        }
      continue;
    }
    const float eps = eps_pc ? eps_pc[r] : eps_s;
    const float h = eps * kick_a;   // step_size * coef (integrators.py:236)
    const float h2 = eps * kick_b;
    const float ed = eps * drift;   // step_size * coef (integrators.py:200)
    const float* im = imm + r * imm_stride;
    if constexpr (VEC == 4) {
      // The launch is usually in place (q_out == q_in, p_out == p_in), so the compiler may not move
      // a load above an earlier store: all loads of a 4 KB span are issued first, in program
      // order, then the arithmetic and the stores -- 16 x 16 B in flight per lane instead of 2.
      constexpr int U = 4;
      for (int64_t j0 = (int64_t)lane * 4; j0 < D; j0 += 256 * U) {
        F4 pp[U], gg[U], qq[U], mm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t j = j0 + 256 * u;
          if (j < D) {
            pp[u] = ld4(p_in + base + j);
            gg[u] = ld4(g + base + j);
            qq[u] = ld4(q_in + base + j);
            mm[u] = ld4(im + j);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t j = j0 + 256 * u;
          if (j < D) {
            F4 pn, qn;
            pn.x = fmaf(h, gg[u].x, pp[u].x); pn.y = fmaf(h, gg[u].y, pp[u].y);
            pn.z = fmaf(h, gg[u].z, pp[u].z); pn.w = fmaf(h, gg[u].w, pp[u].w);
            if constexpr (KICKS == 2) {
              pn.x = fmaf(h2, gg[u].x, pn.x); pn.y = fmaf(h2, gg[u].y, pn.y);
              pn.z = fmaf(h2, gg[u].z, pn.z); pn.w = fmaf(h2, gg[u].w, pn.w);
            }
            qn.x = fmaf(ed, mm[u].x * pn.x, qq[u].x); qn.y = fmaf(ed, mm[u].y * pn.y, qq[u].y);
            qn.z = fmaf(ed, mm[u].z * pn.z, qq[u].z); qn.w = fmaf(ed, mm[u].w * pn.w, qq[u].w);
            st4(p_out + base + j, pn);
            st4(q_out + base + j, qn);
          }
        }
      }
    } else {
      for (int64_t j = lane; j < D; j += 64) {
        float pn = fmaf(h, g[base + j], p_in[base + j]);
        if constexpr (KICKS == 2) pn = fmaf(h2, g[base + j], pn);
        const float qn = fmaf(ed, im[j] * pn, q_in[base + j]);
        p_out[base + j] = pn;
        q_out[base + j] = qn;
      }
    }
  }
}

// Same stage, one 16-byte piece per lane and no loop: for rows that are a multiple of 1 KB floats
// (D % 1024 == 0) a workgroup owns one 4 KB span of one row, so the row index is workgroup-uniform.
// The leapfrog needs no row reduction, and many short waves fill the chip more evenly than one
// wave per row (measured on pseudo-random data, tools/lf_variants: 46.6 vs 48.3 us for 16 384 rows
// of 1 024, 233 vs 241 us for 65 536).  Workgroups are numbered from the END of the arrays for the
// same Infinity-Cache reason as above.
template <int KICKS, bool NT = false>
__global__ void __launch_bounds__(kBlock)
k_leapfrog_diag_flat(int64_t D, int bpr, float eps_s, const float* __restrict__ eps_pc,
                     const float* __restrict__ imm, int64_t imm_stride, const float* q_in,
                     const float* p_in, const float* __restrict__ g, float* q_out, float* p_out,
                     const int32_t* __restrict__ n_steps, int32_t step_idx, float kick_a, float kick_b,
                     float drift) {
  const unsigned b = BJX_REVERSE_ROWS ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const int64_t r = b / (unsigned)bpr;
  const int64_t j = (int64_t)(b - (unsigned)r * (unsigned)bpr) * 1024 + threadIdx.x * 4;
  const int64_t at = r * D + j;
  if (n_steps && step_idx >= n_steps[r]) {
    if (q_out != q_in) {
      st4(q_out + at, ld4(q_in + at));
      st4(p_out + at, ld4(p_in + at));
    }
    return;
  }
  const float eps = eps_pc ? eps_pc[r] : eps_s;
  const float h = eps * kick_a, h2 = eps * kick_b, ed = eps * drift;
  // NT: the launch streams more than the Infinity Cache holds -- nontemporal loads and stores of the
  // three state arrays (the shared inverse mass vector stays a plain, cached load)
  const F4 pp = ld4_t<NT>(p_in + at), gg = ld4_t<NT>(g + at), qq = ld4_t<NT>(q_in + at);
  const F4 mm = imm_stride ? ld4_t<NT>(imm + r * imm_stride + j) : ld4(imm + j);
  F4 pn, qn;
  pn.x = fmaf(h, gg.x, pp.x); pn.y = fmaf(h, gg.y, pp.y);
  pn.z = fmaf(h, gg.z, pp.z); pn.w = fmaf(h, gg.w, pp.w);
  if constexpr (KICKS == 2) {
    pn.x = fmaf(h2, gg.x, pn.x); pn.y = fmaf(h2, gg.y, pn.y);
    pn.z = fmaf(h2, gg.z, pn.z); pn.w = fmaf(h2, gg.w, pn.w);
  }
  qn.x = fmaf(ed, mm.x * pn.x, qq.x); qn.y = fmaf(ed, mm.y * pn.y, qq.y);
  qn.z = fmaf(ed, mm.z * pn.z, qq.z); qn.w = fmaf(ed, mm.w * pn.w, qq.w);
  st4_t<NT>(p_out + at, pn);
  st4_t<NT>(q_out + at, qn);
}

// The same one-piece-per-lane stage for ANY row length that is a multiple of 4 floats: lane i of
// the launch owns the i-th 16-byte piece of the (N, D) arrays, row = i / (D/4).  With one row per
// wave a row of D = 64 floats keeps 16 of 64 lanes busy (measured: 49 % of the HBM peak at
// 524 288 x 64, 84 % at D = 100); here every lane works whatever D is.  32-bit index arithmetic
// (the host checks N * D / 4 < 2^31).
template <int KICKS>
__global__ void __launch_bounds__(kBlock)
k_leapfrog_diag_flat_any(uint32_t total4, uint32_t D4, float eps_s, const float* __restrict__ eps_pc,
                         const float* __restrict__ imm, uint32_t imm_stride4, const float* q_in,
                         const float* p_in, const float* __restrict__ g, float* q_out, float* p_out,
                         const int32_t* __restrict__ n_steps, int32_t step_idx, float kick_a,
                         float kick_b, float drift) {
  const uint32_t b = BJX_REVERSE_ROWS ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const uint32_t i = b * kBlock + threadIdx.x;
  if (i >= total4) return;
  const uint32_t r = i / D4;
  const uint32_t j4 = i - r * D4;
  const int64_t at = (int64_t)i * 4;
  if (n_steps && step_idx >= n_steps[r]) {
    if (q_out != q_in) {
      st4(q_out + at, ld4(q_in + at));
      st4(p_out + at, ld4(p_in + at));
    }
    return;
  }
  const float eps = eps_pc ? eps_pc[r] : eps_s;
  const float h = eps * kick_a, h2 = eps * kick_b, ed = eps * drift;
  const F4 pp = ld4(p_in + at), gg = ld4(g + at), qq = ld4(q_in + at);
  const F4 mm = ld4(imm + ((int64_t)r * imm_stride4 + j4) * 4);
  F4 pn, qn;
  pn.x = fmaf(h, gg.x, pp.x); pn.y = fmaf(h, gg.y, pp.y);
  pn.z = fmaf(h, gg.z, pp.z); pn.w = fmaf(h, gg.w, pp.w);
  if constexpr (KICKS == 2) {
    pn.x = fmaf(h2, gg.x, pn.x); pn.y = fmaf(h2, gg.y, pn.y);
    pn.z = fmaf(h2, gg.z, pn.z); pn.w = fmaf(h2, gg.w, pn.w);
  }
  qn.x = fmaf(ed, mm.x * pn.x, qq.x); qn.y = fmaf(ed, mm.y * pn.y, qq.y);
  qn.z = fmaf(ed, mm.z * pn.z, qq.z); qn.w = fmaf(ed, mm.w * pn.w, qq.w);
  st4(p_out + at, pn);
  st4(q_out + at, qn);
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
                  float* __restrict__ energy_out, float kick_coef) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    const float eps = eps_pc ? eps_pc[r] : eps_s;
    const float h = eps * kick_coef;  // last coefficient of the palindromic integrator (0.5 for VV)
    const int64_t base = r * D;
    const float* im = imm + r * imm_stride;
    double acc = 0.0;
    // pass 1: closing half kick, flipped momentum out, kinetic energy
    if constexpr (VEC == 4) {
      F4 pp[4], gg[4], mm[4];
      row_sweep4<4>(
          lane, D,
          [&](int u, int64_t j) {
            pp[u] = ld4(p + base + j);
            gg[u] = ld4(g1 + base + j);
            mm[u] = ld4(im + j);
          },
          [&](int u, int64_t j) {
            F4 pn;
            pn.x = fmaf(h, gg[u].x, pp[u].x); pn.y = fmaf(h, gg[u].y, pp[u].y);
            pn.z = fmaf(h, gg[u].z, pp[u].z); pn.w = fmaf(h, gg[u].w, pp[u].w);
            acc += (double)(mm[u].x * pn.x) * (double)pn.x;
            acc += (double)(mm[u].y * pn.y) * (double)pn.y;
            acc += (double)(mm[u].z * pn.z) * (double)pn.z;
            acc += (double)(mm[u].w * pn.w) * (double)pn.w;
            if (p_end) st4(p_end + base + j, F4{-1.0f * pn.x, -1.0f * pn.y, -1.0f * pn.z, -1.0f * pn.w});
          });
    } else {
      for (int64_t j = lane; j < D; j += 64) {
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
    if constexpr (VEC == 4) {
      F4 a[4], b[4];
      row_sweep4<4>(
          lane, D,
          [&](int u, int64_t j) {
            a[u] = ld4(qs + base + j);
            b[u] = ld4(gs + base + j);
          },
          [&](int u, int64_t j) {
            st4(q_out + base + j, a[u]);
            st4(g_out + base + j, b[u]);
          });
    } else {
      for (int64_t j = lane; j < D; j += 64) {
        q_out[base + j] = qs[base + j];
        g_out[base + j] = gs[base + j];
      }
    }
  }
}

// short rows (D <= 128, D % 4 == 0): G lanes per row, 64 / G rows per wave -- the per-row scalar chain
// (three threefry blocks, one exp) is then paid once per 2 or 4 rows of vector issue instead of once per
// row.  Same arithmetic and per-lane accumulation order as k_hmc_finish_diag<4>.
template <int G>
__global__ void __launch_bounds__(kBlock)
k_hmc_finish_diag_short(Key key, int64_t off, int64_t fold, int64_t N, int64_t D, float eps_s,
                        const float* __restrict__ eps_pc, const float* __restrict__ imm,
                        int64_t imm_stride, float thr, const float* __restrict__ q0,
                        const float* __restrict__ logp0, const float* __restrict__ g0,
                        const float* __restrict__ ke0, const float* __restrict__ q1,
                        const float* __restrict__ logp1, const float* __restrict__ g1, const float* p,
                        float* p_end, float* __restrict__ q_out, float* __restrict__ logp_out,
                        float* __restrict__ g_out, float* __restrict__ acc_rate_out,
                        uint8_t* __restrict__ is_acc_out, uint8_t* __restrict__ is_div_out,
                        float* __restrict__ energy_out, float kick_coef) {
  constexpr int R = BJX_WAVE / G;
  const int lane = threadIdx.x & 63;
  const int sub = lane / G, gl = lane % G;
  for (int64_t r0 = wave_row0() * R; r0 < N; r0 += wave_row_stride() * R) {
    const int64_t r = r0 + sub;
    const bool valid = r < N;
    const int64_t rr = valid ? r : N - 1;  // idle groups shadow the last row and write nothing
    const float eps = eps_pc ? eps_pc[rr] : eps_s;
    const float h = eps * kick_coef;
    const int64_t base = rr * D;
    const float* im = imm + rr * imm_stride;
    double acc = 0.0;
    for (int64_t j = (int64_t)gl * 4; j < D; j += G * 4) {
      const F4 pp = ld4(p + base + j), gg = ld4(g1 + base + j), mm = ld4(im + j);
      F4 pn;
      pn.x = fmaf(h, gg.x, pp.x); pn.y = fmaf(h, gg.y, pp.y);
      pn.z = fmaf(h, gg.z, pp.z); pn.w = fmaf(h, gg.w, pp.w);
      acc += (double)(mm.x * pn.x) * (double)pn.x;
      acc += (double)(mm.y * pn.y) * (double)pn.y;
      acc += (double)(mm.z * pn.z) * (double)pn.z;
      acc += (double)(mm.w * pn.w) * (double)pn.w;
      if (p_end && valid) st4(p_end + base + j, F4{-1.0f * pn.x, -1.0f * pn.y, -1.0f * pn.z, -1.0f * pn.w});
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, BJX_WAVE);
    const float ke1 = 0.5f * (float)acc;
    const float lp0 = logp0[rr], lp1 = logp1[rr];
    const float H0 = -lp0 + ke0[rr];
    const float H1 = -lp1 + ke1;
    float delta = H0 - H1;
    if (delta != delta) delta = -__builtin_inff();
    const bool is_div = (-delta) > thr;
    const float p_acc = fminf(exp_cr(delta), 1.0f);
    const Key kc = chain_key(key, (uint64_t)(rr + off), fold);
    const Key ki = key_child(kc, 1);
    const float u = key_uniform(ki);
    const bool accept = u < p_acc;
    if (!valid) continue;
    if (gl == 0) {
      logp_out[r] = accept ? lp1 : lp0;
      acc_rate_out[r] = p_acc;
      is_acc_out[r] = accept ? 1 : 0;
      is_div_out[r] = is_div ? 1 : 0;
      energy_out[r] = H1;
    }
    const float* qs = accept ? q1 : q0;
    const float* gs = accept ? g1 : g0;
    for (int64_t j = (int64_t)gl * 4; j < D; j += G * 4) {
      const F4 a = ld4(qs + base + j), b = ld4(gs + base + j);
      st4(q_out + base + j, a);
      st4(g_out + base + j, b);
    }
  }
}

// ------------------------------------------------------------------------------ multinomial HMC
// One step of static_progressive_integration (trajectory.py:214-225) fused with the opening half of
// the next leapfrog.  Input: p = momentum after the OPENING half kick of step i, g = gradient at the
// new position q.  Pass 1: closing half kick + kinetic energy -> proposal weight, divergence,
// progressive uniform sampling with key fold_in(key_integrator, i) (proposal.py:118-143).
// Pass 2: reservoir update on accept; if do_next, opening half kick + drift of step i+1 in place.
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_mhmc_step_diag(Key key, int64_t off, int64_t fold, int64_t N, int64_t D, int64_t step, int do_next_arg,
                 float eps_s, const float* __restrict__ eps_pc, const float* __restrict__ imm,
                 int64_t imm_stride, float thr, const float* __restrict__ logp0,
                 const float* __restrict__ ke0, float* q, float* p, const float* __restrict__ g,
                 const float* __restrict__ logp_new, float* __restrict__ W, float* __restrict__ S,
                 uint8_t* __restrict__ any_div, uint8_t* __restrict__ ever, float* __restrict__ Rq,
                 float* __restrict__ Rp, float* __restrict__ Rg, float* __restrict__ Rlogp,
                 float* __restrict__ Renergy, const int32_t* __restrict__ n_steps, float kick_c,
                 float drift_c) {
  // kick_c / drift_c: coefficients b_1 / a_1 of a palindromic integrator (integrators.py:104-150): the
  // closing kick of step i and the opening kick of step i + 1 are both (eps * b_1) g, the drift that
  // follows is (eps * a_1) M^{-1} p.  Velocity Verlet = (0.5, 1.0): eps * 1.0f == eps, the former bits.
  const int lane = threadIdx.x & 63;
  const int do_next_all = do_next_arg;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    // per-chain trajectory lengths (blackjax.dmhmc): a chain whose trajectory is complete is left
    // alone, and the last step of a chain does not open another leapfrog
    int do_next = do_next_all;
    if (n_steps) {
      const int ns = n_steps[r];
      if (step >= ns) continue;
      do_next = do_next_all && step + 1 < ns;
    }
    const float eps = eps_pc ? eps_pc[r] : eps_s;
    const float h = eps * kick_c;
    const float ed = eps * drift_c;
    const int64_t base = r * D;
    const float* im = imm + r * imm_stride;
    double acc = 0.0;
    if constexpr (VEC == 4) {
      F4 pp[4], gg[4], mm[4];
      row_sweep4<4>(
          lane, D,
          [&](int u, int64_t j) {
            pp[u] = ld4(p + base + j);
            gg[u] = ld4(g + base + j);
            mm[u] = ld4(im + j);
          },
          [&](int u, int64_t) {
            const float a = fmaf(h, gg[u].x, pp[u].x), b = fmaf(h, gg[u].y, pp[u].y);
            const float c = fmaf(h, gg[u].z, pp[u].z), d = fmaf(h, gg[u].w, pp[u].w);
            acc += (double)(mm[u].x * a) * (double)a + (double)(mm[u].y * b) * (double)b;
            acc += (double)(mm[u].z * c) * (double)c + (double)(mm[u].w * d) * (double)d;
          });
    } else {
      for (int64_t j = lane; j < D; j += 64) {
        const float a = fmaf(h, g[base + j], p[base + j]);
        acc += (double)(im[j] * a) * (double)a;
      }
    }
    acc = wave_sum(acc);
    const float ke = 0.5f * (float)acc;
    const float lp = logp_new[r];
    const float H0 = -logp0[r] + ke0[r];
    const float e_new = -lp + ke;
    float w = H0 - e_new;
    if (w != w) w = -__builtin_inff();
    const float s_new = fminf(w, 0.0f);
    const bool is_div = (-w) > thr;
    const float Wc = W[r];
    const Key ki = key_child(chain_key(key, (uint64_t)(r + off), fold), 1);
    const float u = key_uniform(key_child(ki, (uint64_t)step));
    const float pa = (float)(1.0 / (1.0 + exp(-(double)(w - Wc))));  // expit
    const bool take = u < pa;
    // logaddexp in fp64, rounded once (same formula as np.logaddexp)
    auto lae = [](float a, float b) {
      const double x = (double)a, y = (double)b;
      if (x == y) return (float)(x + 0.6931471805599453);
      const double t = x - y;
      if (t > 0) return (float)(x + log1p(exp(-t)));
      if (t <= 0) return (float)(y + log1p(exp(t)));
      return (float)t;
    };
    const float Wn = lae(Wc, w), Sn = lae(S[r], s_new);
    if (lane == 0) {
      W[r] = Wn;
      S[r] = Sn;
      if (is_div) any_div[r] = 1;
      if (take) {
        ever[r] = 1;
        Rlogp[r] = lp;
        Renergy[r] = e_new;
      }
    }
    if constexpr (VEC == 4) {
      F4 pp[4], gg[4], qq[4], mm[4];
      row_sweep4<4>(
          lane, D,
          [&](int u, int64_t j) {
            pp[u] = ld4(p + base + j);
            gg[u] = ld4(g + base + j);
            qq[u] = ld4(q + base + j);
            if (do_next) mm[u] = ld4(im + j);
          },
          [&](int u, int64_t j) {
            const F4 G = gg[u], Q = qq[u];
            F4 pf{fmaf(h, G.x, pp[u].x), fmaf(h, G.y, pp[u].y), fmaf(h, G.z, pp[u].z), fmaf(h, G.w, pp[u].w)};
            if (take) {
              st4(Rq + base + j, Q);
              st4(Rp + base + j, pf);
              st4(Rg + base + j, G);
            }
            if (do_next) {
              const F4 M = mm[u];
              F4 pn{fmaf(h, G.x, pf.x), fmaf(h, G.y, pf.y), fmaf(h, G.z, pf.z), fmaf(h, G.w, pf.w)};
              F4 qn{fmaf(ed, M.x * pn.x, Q.x), fmaf(ed, M.y * pn.y, Q.y),
                    fmaf(ed, M.z * pn.z, Q.z), fmaf(ed, M.w * pn.w, Q.w)};
              st4(p + base + j, pn);
              st4(q + base + j, qn);
            }
          });
    } else {
      for (int64_t j = lane; j < D; j += 64) {
        const float gg = g[base + j], qq = q[base + j];
        const float pf = fmaf(h, gg, p[base + j]);
        if (take) {
          Rq[base + j] = qq;
          Rp[base + j] = pf;
          Rg[base + j] = gg;
        }
        if (do_next) {
          const float pn = fmaf(h, gg, pf);
          p[base + j] = pn;
          q[base + j] = fmaf(ed, im[j] * pn, qq);
        }
      }
    }
  }
}

// Chains that never replaced the initial proposal keep z0 (trajectory.py:212); acceptance_rate =
// exp(sum_log_p_accept) / L (hmc.py:234).
__global__ void __launch_bounds__(kBlock)
k_mhmc_finish(int64_t N, int64_t D, float n_steps, const float* __restrict__ q0,
              const float* __restrict__ p0, const float* __restrict__ g0,
              const float* __restrict__ logp0, const float* __restrict__ ke0,
              const uint8_t* __restrict__ ever, const float* __restrict__ S, float* __restrict__ Rq,
              float* __restrict__ Rp, float* __restrict__ Rg, float* __restrict__ Rlogp,
              float* __restrict__ Renergy, float* __restrict__ acc_rate,
              const int32_t* __restrict__ n_steps_pc) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    if (lane == 0) acc_rate[r] = exp_cr(S[r]) / (n_steps_pc ? (float)n_steps_pc[r] : n_steps);
    if (ever[r]) continue;
    const int64_t base = r * D;
    for (int64_t j = lane; j < D; j += 64) {
      Rq[base + j] = q0[base + j];
      Rp[base + j] = p0[base + j];
      Rg[base + j] = g0[base + j];
    }
    if (lane == 0) {
      Rlogp[r] = logp0[r];
      Renergy[r] = -logp0[r] + ke0[r];
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
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && (N == 0 || u_out), "bjx_rng_uniform: bad arguments");
  hipLaunchKernelGGL(k_rng_uniform, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, Key{key0, key1}, chain_offset, N, u_out);
  return bjx_check_launch("bjx_rng_uniform");
}

int bjx_rng_key_probe(void* stream, uint32_t key0, uint32_t key1, int64_t D, float* z_out, float* u_out,
                      int64_t n_children, uint32_t* children_out) {
  BJX_CHECK_ARG(D >= 0 && n_children >= 0 && (D == 0 || z_out) && (n_children == 0 || children_out),
                "bjx_rng_key_probe: bad arguments");
  const int64_t n = D > n_children ? D : n_children;
  hipLaunchKernelGGL(k_rng_key_probe, dim3((unsigned)((n > 0 ? n : 1) + kBlock - 1) / kBlock), dim3(kBlock), 0,
                     (hipStream_t)stream, Key{key0, key1}, D, z_out, u_out, n_children, children_out);
  return bjx_check_launch("bjx_rng_key_probe");
}

int bjx_log1p_device_check(void* stream, uint32_t stride, unsigned long long* counts_out) {
  BJX_CHECK_ARG(stride >= 1 && counts_out, "bjx_log1p_device_check: bad arguments");
  const uint32_t first = 0x80000000u;
  const uint64_t span = 0xBF800000ull - first;       // negative inputs in (-1, -0.0]
  const uint64_t n = (span + stride - 1) / stride + 1;  // + one iteration for +0.0
  if (hipMemsetAsync(counts_out, 0, 4 * sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess) {
    bjx_set_error("%s", "bjx_log1p_device_check: hipMemsetAsync failed");
    return 2;
  }
  hipLaunchKernelGGL(k_log1p_check, dim3(256 * 16), dim3(kBlock), 0, (hipStream_t)stream, first, n, stride, counts_out);
  return bjx_check_launch("bjx_log1p_device_check");
}

int bjx_hmc_momentum_diag(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                          int64_t step_fold, int64_t N, int64_t D, const float* imm, int64_t imm_stride,
                          float* p_out, float* ke_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && imm && p_out && ke_out, "bjx_hmc_momentum_diag: bad arguments");
  BJX_CHECK_ARG(imm_stride == 0 || imm_stride == D, "bjx_hmc_momentum_diag: imm_stride must be 0 or D");
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  const Key key{key0, key1};
  if (bjx_vec4_ok(D, imm, p_out) && D <= 128) {
    // G lanes per row = the smallest power of two with G * 4 >= D (at least 4): 64 / G rows per wave
#define BJX_MOM_SHORT(G_)                                                                             \
  hipLaunchKernelGGL(k_momentum_diag_short<G_>, dim3(bjx_row_grid((N * G_ + 63) / 64, kWavesPerBlock)), \
                     block, 0, (hipStream_t)stream, key, chain_offset, step_fold, N, D, imm, imm_stride, \
                     p_out, ke_out)
    if (D <= 16) BJX_MOM_SHORT(4);
    else if (D <= 32) BJX_MOM_SHORT(8);
    else if (D <= 64) BJX_MOM_SHORT(16);
    else BJX_MOM_SHORT(32);
#undef BJX_MOM_SHORT
  } else if (bjx_vec4_ok(D, imm, p_out) && imm_stride == 0 && D <= 1024 && N >= 4096)
    // one shared metric: mass_sqrt hoisted out of the row loop, four rows per wave (k_momentum_diag, HOIST)
    hipLaunchKernelGGL((k_momentum_diag<4, false, true>), dim3(bjx_row_grid((N + 3) / 4, kWavesPerBlock)), block, 0,
                       (hipStream_t)stream, key, chain_offset, step_fold, N, D, imm, imm_stride, p_out, ke_out, 0.0f,
                       nullptr, nullptr, nullptr, nullptr, nullptr);
  else if (bjx_vec4_ok(D, imm, p_out))
    hipLaunchKernelGGL((k_momentum_diag<4, false>), grid, block, 0, (hipStream_t)stream, key, chain_offset,
                       step_fold, N, D, imm, imm_stride, p_out, ke_out, 0.0f, nullptr, nullptr, nullptr, nullptr,
                       nullptr);
  else
    hipLaunchKernelGGL((k_momentum_diag<1, false>), grid, block, 0, (hipStream_t)stream, key, chain_offset,
                       step_fold, N, D, imm, imm_stride, p_out, ke_out, 0.0f, nullptr, nullptr, nullptr, nullptr,
                       nullptr);
  return bjx_check_launch("bjx_hmc_momentum_diag");
}

int bjx_hmc_momentum_kick_diag(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                               int64_t step_fold, int64_t N, int64_t D, const float* imm, int64_t imm_stride,
                               float eps, const float* eps_per_chain, const float* q0, const float* g0,
                               float* p_out, float* ke_out, float* q1_out, float* p_half_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && imm && q0 && g0 && p_out && ke_out && q1_out && p_half_out,
                "bjx_hmc_momentum_kick_diag: bad arguments");
  BJX_CHECK_ARG(imm_stride == 0 || imm_stride == D, "bjx_hmc_momentum_kick_diag: imm_stride must be 0 or D");
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  const Key key{key0, key1};
  if (bjx_vec4_ok(D, imm, q0, g0, p_out, q1_out, p_half_out) && imm_stride == 0 && D <= 1024 && N >= 4096)
    hipLaunchKernelGGL((k_momentum_diag<4, true, true>), dim3(bjx_row_grid((N + 3) / 4, kWavesPerBlock)), block, 0,
                       (hipStream_t)stream, key, chain_offset, step_fold, N, D, imm, imm_stride, p_out, ke_out, eps,
                       eps_per_chain, q0, g0, q1_out, p_half_out);
  else if (bjx_vec4_ok(D, imm, q0, g0, p_out, q1_out, p_half_out))
    hipLaunchKernelGGL((k_momentum_diag<4, true>), grid, block, 0, (hipStream_t)stream, key, chain_offset,
                       step_fold, N, D, imm, imm_stride, p_out, ke_out, eps, eps_per_chain, q0, g0, q1_out,
                       p_half_out);
  else
    hipLaunchKernelGGL((k_momentum_diag<1, true>), grid, block, 0, (hipStream_t)stream, key, chain_offset,
                       step_fold, N, D, imm, imm_stride, p_out, ke_out, eps, eps_per_chain, q0, g0, q1_out,
                       p_half_out);
  return bjx_check_launch("bjx_hmc_momentum_kick_diag");
}

int bjx_leapfrog_diag_coef(void* stream, int64_t N, int64_t D, int n_kicks, float kick_a,
                           float kick_b, float drift, float eps, const float* eps_per_chain,
                           const float* imm, int64_t imm_stride, const float* q_in,
                           const float* p_in, const float* g, float* q_out, float* p_out,
                           const int32_t* n_steps, int32_t step_idx) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && imm && q_in && p_in && g && q_out && p_out,
                "bjx_leapfrog_diag: bad arguments");
  BJX_CHECK_ARG(n_kicks == 1 || n_kicks == 2, "bjx_leapfrog_diag: n_kicks must be 1 or 2");
  BJX_CHECK_ARG(imm_stride == 0 || imm_stride == D, "bjx_leapfrog_diag: imm_stride must be 0 or D");
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
#define BJX_LF(V, K)                                                                          \
  hipLaunchKernelGGL((k_leapfrog_diag<V, K>), grid, block, 0, s, N, D, eps, eps_per_chain, imm, \
                     imm_stride, q_in, p_in, g, q_out, p_out, n_steps, step_idx, kick_a, kick_b,  \
                     drift)
  static const bool flat_ok = [] {
    const char* e = getenv("BJX_LF_FLAT");
    return e ? atoi(e) != 0 : true;
  }();
  if (flat_ok && D % 1024 == 0 && N * (D / 1024) < ((int64_t)1 << 31) &&
      bjx_vec4_ok(D, imm, q_in, p_in, g, q_out, p_out)) {
    const int bpr = (int)(D / 1024);
    const dim3 fgrid((unsigned)(N * bpr));
    // nontemporal accesses when ONE launch moves more than the 256 MiB Infinity Cache holds (q, p, g
    // [, per-chain imm] of all its rows): such a launch streams from HBM whatever the caches do.
    // BJX_LF_NT=0 / 1 forces plain / nontemporal (A/B: tools/README.md).
    static const int nt_mode = [] { const char* e = getenv("BJX_LF_NT"); return e ? atoi(e) : -1; }();
    const int64_t launch_bytes = N * D * 4 * (3 + (imm_stride ? 1 : 0));
    const bool nt = nt_mode < 0 ? launch_bytes > ((int64_t)256 << 20) : nt_mode != 0;
#define BJX_LFF(K, T)                                                                                    \
  hipLaunchKernelGGL((k_leapfrog_diag_flat<K, T>), fgrid, block, 0, s, D, bpr, eps, eps_per_chain, imm, \
                     imm_stride, q_in, p_in, g, q_out, p_out, n_steps, step_idx, kick_a, kick_b, drift)
    if (n_kicks == 1) { if (nt) BJX_LFF(1, true); else BJX_LFF(1, false); }
    else { if (nt) BJX_LFF(2, true); else BJX_LFF(2, false); }
#undef BJX_LFF
  } else if (flat_ok && (D / 4) % 64 != 0 && N * (D / 4) < ((int64_t)1 << 31) &&
             bjx_vec4_ok(D, imm, q_in, p_in, g, q_out, p_out)) {
    // rows that do not fill whole waves: one 16-byte piece per lane over the flattened arrays
    const uint32_t D4 = (uint32_t)(D / 4), total4 = (uint32_t)(N * (D / 4));
    const dim3 fgrid((total4 + kBlock - 1) / kBlock);
    const uint32_t is4 = imm_stride ? D4 : 0u;
    if (n_kicks == 1)
      hipLaunchKernelGGL(k_leapfrog_diag_flat_any<1>, fgrid, block, 0, s, total4, D4, eps, eps_per_chain,
                         imm, is4, q_in, p_in, g, q_out, p_out, n_steps, step_idx, kick_a, kick_b, drift);
    else
      hipLaunchKernelGGL(k_leapfrog_diag_flat_any<2>, fgrid, block, 0, s, total4, D4, eps, eps_per_chain,
                         imm, is4, q_in, p_in, g, q_out, p_out, n_steps, step_idx, kick_a, kick_b, drift);
  } else if (bjx_vec4_ok(D, imm, q_in, p_in, g, q_out, p_out)) {
    if (n_kicks == 1) BJX_LF(4, 1); else BJX_LF(4, 2);
  } else {
    if (n_kicks == 1) BJX_LF(1, 1); else BJX_LF(1, 2);
  }
#undef BJX_LF
  return bjx_check_launch("bjx_leapfrog_diag");
}

// velocity Verlet: coefficients [0.5, 1.0, 0.5] (integrators.py:321-322)
int bjx_leapfrog_diag_masked(void* stream, int64_t N, int64_t D, int n_kicks, float eps,
                             const float* eps_per_chain, const float* imm, int64_t imm_stride,
                             const float* q_in, const float* p_in, const float* g, float* q_out,
                             float* p_out, const int32_t* n_steps, int32_t step_idx) {
  return bjx_leapfrog_diag_coef(stream, N, D, n_kicks, 0.5f, 0.5f, 1.0f, eps, eps_per_chain, imm,
                                imm_stride, q_in, p_in, g, q_out, p_out, n_steps, step_idx);
}

int bjx_leapfrog_diag(void* stream, int64_t N, int64_t D, int n_kicks, float eps,
                      const float* eps_per_chain, const float* imm, int64_t imm_stride,
                      const float* q_in, const float* p_in, const float* g, float* q_out,
                      float* p_out) {
  return bjx_leapfrog_diag_coef(stream, N, D, n_kicks, 0.5f, 0.5f, 1.0f, eps, eps_per_chain, imm,
                                imm_stride, q_in, p_in, g, q_out, p_out, nullptr, 0);
}

namespace {
// split(key, 2)[child] per chain: keys_out[i] = threefry(keys_in[i], (0, child))
__global__ void k_keys_child(int64_t N, const uint32_t* __restrict__ kin, uint32_t child,
                             uint32_t* __restrict__ kout) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const Key k = key_child(Key{kin[2 * i], kin[2 * i + 1]}, child);
  kout[2 * i] = k.k0;
  kout[2 * i + 1] = k.k1;
}

// jax.random.randint(key, (), minval, maxval, int32) per chain (jax/_src/random.py::_randint)
__global__ void k_keys_randint(int64_t N, const uint32_t* __restrict__ kin, int32_t minval,
                               int32_t maxval, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const Key key{kin[2 * i], kin[2 * i + 1]};
  const uint32_t hi = key_bits32(key_child(key, 0), 0);  // random_bits(split(key)[0], 32, ())
  const uint32_t lo = key_bits32(key_child(key, 1), 0);
  uint32_t span = (uint32_t)(maxval - minval);
  if (maxval <= minval) span = 1u;
  uint32_t mult = (1u << 16) % span;  // 2^32 % span computed as ((2^16 % span)^2) % span
  mult = (mult * mult) % span;
  uint32_t off = (hi % span) * mult + (lo % span);
  off %= span;
  out[i] = minval + (int32_t)off;
}
}  // namespace

int bjx_keys_child(void* stream, int64_t N, const uint32_t* keys_in, uint32_t child,
                   uint32_t* keys_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && keys_in && keys_out, "bjx_keys_child: bad arguments");
  hipLaunchKernelGGL(k_keys_child, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, N, keys_in, child, keys_out);
  return bjx_check_launch("bjx_keys_child");
}

int bjx_keys_randint(void* stream, int64_t N, const uint32_t* keys, int32_t minval, int32_t maxval,
                     int32_t* out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && keys && out, "bjx_keys_randint: bad arguments");
  hipLaunchKernelGGL(k_keys_randint, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, N, keys, minval, maxval, out);
  return bjx_check_launch("bjx_keys_randint");
}

int bjx_hmc_finish_diag_coef(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                             int64_t step_fold, int64_t N, int64_t D, float kick_coef, float eps,
                             const float* eps_per_chain, const float* imm, int64_t imm_stride,
                             float divergence_threshold,
                        const float* q0, const float* logp0, const float* g0, const float* ke0,
                        const float* q1, const float* logp1, const float* g1, const float* p,
                        float* p_end_out, float* q_out, float* logp_out, float* g_out,
                        float* acceptance_rate_out, uint8_t* is_accepted_out,
                        uint8_t* is_divergent_out, float* energy_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && imm && q0 && logp0 && g0 && ke0 && q1 && logp1 && g1 && p &&
                    q_out && logp_out && g_out && acceptance_rate_out && is_accepted_out &&
                    is_divergent_out && energy_out,
                "bjx_hmc_finish_diag: bad arguments");
  BJX_CHECK_ARG(imm_stride == 0 || imm_stride == D, "bjx_hmc_finish_diag: imm_stride must be 0 or D");
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  const Key key{key0, key1};
  hipStream_t s = (hipStream_t)stream;
#define BJX_FIN(V)                                                                              \
  hipLaunchKernelGGL(k_hmc_finish_diag<V>, grid, block, 0, s, key, chain_offset, step_fold, N, D, \
                     eps,                                                                       \
                     eps_per_chain, imm, imm_stride, divergence_threshold, q0, logp0, g0, ke0,  \
                     q1, logp1, g1, p, p_end_out, q_out, logp_out, g_out, acceptance_rate_out,  \
                     is_accepted_out, is_divergent_out, energy_out, kick_coef)
  if (bjx_vec4_ok(D, imm, q0, g0, q1, g1, p, p_end_out, q_out, g_out) && D <= 128) {
#define BJX_FIN_SHORT(G_, GRID)                                                                       \
  hipLaunchKernelGGL(k_hmc_finish_diag_short<G_>, GRID, block, 0, s, key, chain_offset, step_fold, N, \
                     D, eps, eps_per_chain, imm, imm_stride, divergence_threshold, q0, logp0, g0, ke0, \
                     q1, logp1, g1, p, p_end_out, q_out, logp_out, g_out, acceptance_rate_out,        \
                     is_accepted_out, is_divergent_out, energy_out, kick_coef)
    if (D <= 16) BJX_FIN_SHORT(4, dim3(bjx_row_grid((N + 15) / 16, kWavesPerBlock)));
    else if (D <= 32) BJX_FIN_SHORT(8, dim3(bjx_row_grid((N + 7) / 8, kWavesPerBlock)));
    else if (D <= 64) BJX_FIN_SHORT(16, dim3(bjx_row_grid((N + 3) / 4, kWavesPerBlock)));
    else BJX_FIN_SHORT(32, dim3(bjx_row_grid((N + 1) / 2, kWavesPerBlock)));
#undef BJX_FIN_SHORT
  } else if (bjx_vec4_ok(D, imm, q0, g0, q1, g1, p, p_end_out, q_out, g_out)) BJX_FIN(4);
  else BJX_FIN(1);
#undef BJX_FIN
  return bjx_check_launch("bjx_hmc_finish_diag");
}

int bjx_hmc_finish_diag(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                        int64_t step_fold, int64_t N, int64_t D, float eps, const float* eps_per_chain,
                        const float* imm, int64_t imm_stride, float divergence_threshold,
                        const float* q0, const float* logp0, const float* g0, const float* ke0,
                        const float* q1, const float* logp1, const float* g1, const float* p,
                        float* p_end_out, float* q_out, float* logp_out, float* g_out,
                        float* acceptance_rate_out, uint8_t* is_accepted_out,
                        uint8_t* is_divergent_out, float* energy_out) {
  return bjx_hmc_finish_diag_coef(stream, key0, key1, chain_offset, step_fold, N, D, 0.5f, eps,
                                  eps_per_chain, imm, imm_stride, divergence_threshold, q0, logp0, g0,
                                  ke0, q1, logp1, g1, p, p_end_out, q_out, logp_out, g_out,
                                  acceptance_rate_out, is_accepted_out, is_divergent_out, energy_out);
}

static int mhmc_step_diag(const char* what, void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                          int64_t step_fold, int64_t N, int64_t D, int64_t step, int do_next, float eps,
                          const float* eps_per_chain, const float* imm, int64_t imm_stride,
                          float divergence_threshold, const float* logp0, const float* ke0, float* q,
                          float* p, const float* g, const float* logp_new, float* weight,
                          float* sum_log_p_accept, uint8_t* any_divergent, uint8_t* ever_accepted,
                          float* prop_q, float* prop_p, float* prop_g, float* prop_logp,
                          float* prop_energy, const int32_t* n_steps, float kick_c = 0.5f,
                          float drift_c = 1.0f) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && step >= 0 && imm && logp0 && ke0 && q && p && g && logp_new &&
                    weight && sum_log_p_accept && any_divergent && ever_accepted && prop_q && prop_p &&
                    prop_g && prop_logp && prop_energy,
                "bjx_mhmc_step_diag: bad arguments");
  BJX_CHECK_ARG(imm_stride == 0 || imm_stride == D, "bjx_mhmc_step_diag: imm_stride must be 0 or D");
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  const Key key{key0, key1};
  hipStream_t s = (hipStream_t)stream;
#define BJX_MH(V)                                                                               \
  hipLaunchKernelGGL(k_mhmc_step_diag<V>, grid, block, 0, s, key, chain_offset, step_fold, N, D, \
                     step, do_next, eps, eps_per_chain, imm, imm_stride, divergence_threshold,  \
                     logp0, ke0, q, p, g, logp_new, weight, sum_log_p_accept, any_divergent,    \
                     ever_accepted, prop_q, prop_p, prop_g, prop_logp, prop_energy, n_steps, kick_c, drift_c)
  if (bjx_vec4_ok(D, imm, q, p, g, prop_q, prop_p, prop_g)) BJX_MH(4);
  else BJX_MH(1);
#undef BJX_MH
  return bjx_check_launch(what);
}

int bjx_mhmc_step_diag(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                       int64_t step_fold, int64_t N, int64_t D, int64_t step, int do_next, float eps,
                       const float* eps_per_chain, const float* imm, int64_t imm_stride,
                       float divergence_threshold, const float* logp0, const float* ke0, float* q,
                       float* p, const float* g, const float* logp_new, float* weight,
                       float* sum_log_p_accept, uint8_t* any_divergent, uint8_t* ever_accepted,
                       float* prop_q, float* prop_p, float* prop_g, float* prop_logp,
                       float* prop_energy) {
  return mhmc_step_diag("bjx_mhmc_step_diag", stream, key0, key1, chain_offset, step_fold, N, D, step, do_next,
                        eps, eps_per_chain, imm, imm_stride, divergence_threshold, logp0, ke0, q, p, g,
                        logp_new, weight, sum_log_p_accept, any_divergent, ever_accepted, prop_q, prop_p,
                        prop_g, prop_logp, prop_energy, nullptr);
}

int bjx_mhmc_step_diag_masked(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                              int64_t step_fold, int64_t N, int64_t D, int64_t step, int do_next, float eps,
                              const float* eps_per_chain, const float* imm, int64_t imm_stride,
                              float divergence_threshold, const float* logp0, const float* ke0, float* q,
                              float* p, const float* g, const float* logp_new, float* weight,
                              float* sum_log_p_accept, uint8_t* any_divergent, uint8_t* ever_accepted,
                              float* prop_q, float* prop_p, float* prop_g, float* prop_logp,
                              float* prop_energy, const int32_t* n_steps) {
  BJX_CHECK_ARG(N == 0 || n_steps, "bjx_mhmc_step_diag_masked: n_steps is NULL");
  return mhmc_step_diag("bjx_mhmc_step_diag_masked", stream, key0, key1, chain_offset, step_fold, N, D, step,
                        do_next, eps, eps_per_chain, imm, imm_stride, divergence_threshold, logp0, ke0, q, p,
                        g, logp_new, weight, sum_log_p_accept, any_divergent, ever_accepted, prop_q, prop_p,
                        prop_g, prop_logp, prop_energy, n_steps);
}

int bjx_mhmc_step_diag_coef(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                            int64_t step_fold, int64_t N, int64_t D, int64_t step, int do_next, float eps,
                            const float* eps_per_chain, const float* imm, int64_t imm_stride,
                            float divergence_threshold, const float* logp0, const float* ke0, float* q,
                            float* p, const float* g, const float* logp_new, float* weight,
                            float* sum_log_p_accept, uint8_t* any_divergent, uint8_t* ever_accepted,
                            float* prop_q, float* prop_p, float* prop_g, float* prop_logp,
                            float* prop_energy, const int32_t* n_steps, float kick_coef, float drift_coef) {
  return mhmc_step_diag("bjx_mhmc_step_diag_coef", stream, key0, key1, chain_offset, step_fold, N, D, step,
                        do_next, eps, eps_per_chain, imm, imm_stride, divergence_threshold, logp0, ke0, q, p,
                        g, logp_new, weight, sum_log_p_accept, any_divergent, ever_accepted, prop_q, prop_p,
                        prop_g, prop_logp, prop_energy, n_steps, kick_coef, drift_coef);
}

static int mhmc_finish(const char* what, void* stream, int64_t N, int64_t D, int64_t num_integration_steps,
                       const int32_t* n_steps, const float* q0, const float* p0, const float* g0,
                       const float* logp0, const float* ke0, const uint8_t* ever_accepted,
                       const float* sum_log_p_accept, float* prop_q, float* prop_p, float* prop_g,
                       float* prop_logp, float* prop_energy, float* acceptance_rate_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && q0 && p0 && g0 && logp0 && ke0 && ever_accepted &&
                    sum_log_p_accept && prop_q && prop_p && prop_g && prop_logp && prop_energy &&
                    acceptance_rate_out,
                "bjx_mhmc_finish: bad arguments");
  hipLaunchKernelGGL(k_mhmc_finish, dim3(bjx_row_grid(N, kWavesPerBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, N, D, (float)num_integration_steps, q0, p0, g0, logp0, ke0,
                     ever_accepted, sum_log_p_accept, prop_q, prop_p, prop_g, prop_logp, prop_energy,
                     acceptance_rate_out, n_steps);
  return bjx_check_launch(what);
}

int bjx_mhmc_finish(void* stream, int64_t N, int64_t D, int64_t num_integration_steps,
                    const float* q0, const float* p0, const float* g0, const float* logp0,
                    const float* ke0, const uint8_t* ever_accepted, const float* sum_log_p_accept,
                    float* prop_q, float* prop_p, float* prop_g, float* prop_logp, float* prop_energy,
                    float* acceptance_rate_out) {
  return mhmc_finish("bjx_mhmc_finish", stream, N, D, num_integration_steps, nullptr, q0, p0, g0, logp0, ke0,
                     ever_accepted, sum_log_p_accept, prop_q, prop_p, prop_g, prop_logp, prop_energy,
                     acceptance_rate_out);
}

int bjx_mhmc_finish_masked(void* stream, int64_t N, int64_t D, const int32_t* n_steps, const float* q0,
                           const float* p0, const float* g0, const float* logp0, const float* ke0,
                           const uint8_t* ever_accepted, const float* sum_log_p_accept, float* prop_q,
                           float* prop_p, float* prop_g, float* prop_logp, float* prop_energy,
                           float* acceptance_rate_out) {
  BJX_CHECK_ARG(N == 0 || n_steps, "bjx_mhmc_finish_masked: n_steps is NULL");
  return mhmc_finish("bjx_mhmc_finish_masked", stream, N, D, 1, n_steps, q0, p0, g0, logp0, ke0, ever_accepted,
                     sum_log_p_accept, prop_q, prop_p, prop_g, prop_logp, prop_energy, acceptance_rate_out);
}

}  // extern "C"
