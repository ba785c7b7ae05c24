// Pooled (cross-chain) statistics for many-chain warmup: ChEES-HMC criterion, ensemble moment
// blocks, Halton trajectory jitter.  C ABI in include/bjx_pool.h; reference lines cited there.
//
// Two reduction shapes over an (N, D) batch:
//   * over chains (columns survive): thread (ty, tx) owns VEC consecutive columns and walks the rows
//     of its slab in fp64 registers; a row segment of tpr*VEC floats is one coalesced burst; the
//     slab partials are combined in a fixed order (LDS across ty, then a second kernel across
//     slabs), so results are reproducible run to run.  HBM-bound: one read of each input.
//   * over dimensions (rows survive): one wavefront per chain, 16 B per lane, fp64 wave reduction
//     (same shape as the leapfrog kernels).
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/bjx_pool.h"
#include "bjx_device.h"
#include "bjx_host.h"

using namespace bjx;

namespace {

// ------------------------------------------------------------------------------ column reductions
struct ColGeom {
  int tpr_log2;  // threads per row segment = 1 << tpr_log2 (<= 256)
  int64_t ncb;   // column blocks (grid.y)
  int64_t nslab; // row slabs (grid.x)
};

constexpr int kColU = 8;  // rows a thread has in flight = rows per thread and chunk

ColGeom col_geom(int64_t N, int64_t D, int vec) {
  ColGeom g;
  const int64_t groups = (D + vec - 1) / vec;
  g.tpr_log2 = 0;
  while ((1 << g.tpr_log2) < groups && g.tpr_log2 < 8) ++g.tpr_log2;
  const int64_t tpr = 1 << g.tpr_log2, rp = 256 >> g.tpr_log2;
  g.ncb = (groups + tpr - 1) / tpr;
  if (g.ncb < 1) g.ncb = 1;
  // 512 slabs = 2 workgroups per CU on a large batch: more slabs buy occupancy but every slab writes
  // (and k_colfinal re-reads) K * D doubles; 256 / 512 / 1024 slabs measured 131 / 136 / 143 us for the
  // fused ChEES pass at 65 536 x 1 024 (NOTEBOOK.md section 11)
  int64_t want = 512 / g.ncb;
  if (want < 1) want = 1;
  // rows are dealt to the slabs in chunks of kColU * rp rows, round robin (slab b takes chunks b,
  // b + nslab, ...), so the workgroups of a launch read neighbouring chunks at any moment
  const int64_t chunk = (int64_t)kColU * rp;
  int64_t nchunks = (N + chunk - 1) / chunk;
  if (nchunks < 1) nchunks = 1;
  g.nslab = want < nchunks ? want : nchunks;
  return g;
}

template <int VEC>
__device__ __forceinline__ void ld_vec(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    F4 t = ld4(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = p[0];
  }
}

struct OpChees {  // chees_adaptation.py:241-246, 384-386
  static constexpr int K = 4;
  const float* qp;
  const float* w;
  const float* qi;
  template <int VEC>
  struct Regs {
    float x[VEC], y[VEC], w;
  };
  template <int VEC>
  __device__ __forceinline__ void load(int64_t r, int64_t off, Regs<VEC>& g) const {
    g.w = w[r];
    ld_vec<VEC>(qp + off, g.x);
    ld_vec<VEC>(qi + off, g.y);
  }
  template <int VEC>
  __device__ __forceinline__ void acc(const Regs<VEC>& g, double (&a)[K][VEC], const float*) const {
    const double wr = (double)g.w;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float xs = isfinite(g.x[v]) ? g.x[v] : 0.0f;
      a[0][v] += wr * (double)xs;
      const bool ok = !(g.y[v] != g.y[v]);
      a[1][v] += ok ? (double)g.y[v] : 0.0;
      a[2][v] += ok ? 1.0 : 0.0;
    }
    a[3][0] += wr;  // one accumulator per thread; broadcast over its columns when stored
  }
  static constexpr bool kBroadcastLast = true;
};

struct OpSum {  // metric_buffers.py:429
  static constexpr int K = 1;
  const float* x;
  template <int VEC>
  struct Regs {
    float t[VEC];
  };
  template <int VEC>
  __device__ __forceinline__ void load(int64_t, int64_t off, Regs<VEC>& g) const {
    ld_vec<VEC>(x + off, g.t);
  }
  template <int VEC>
  __device__ __forceinline__ void acc(const Regs<VEC>& g, double (&a)[K][VEC], const float*) const {
#pragma unroll
    for (int v = 0; v < VEC; ++v) a[0][v] += (double)g.t[v];
  }
  static constexpr bool kBroadcastLast = false;
};

struct OpCenteredSq {  // metric_buffers.py:430-433
  static constexpr int K = 1;
  const float* x;
  const float* center;
  template <int VEC>
  struct Regs {
    float t[VEC];
  };
  template <int VEC>
  __device__ __forceinline__ void load(int64_t, int64_t off, Regs<VEC>& g) const {
    ld_vec<VEC>(x + off, g.t);
  }
  template <int VEC>
  __device__ __forceinline__ void acc(const Regs<VEC>& g, double (&a)[K][VEC], const float* c) const {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float d = g.t[v] - c[v];
      a[0][v] += (double)d * (double)d;
    }
  }
  static constexpr bool kBroadcastLast = false;
};

template <int VEC, class Op>
__global__ __launch_bounds__(256) void k_colreduce(int64_t N, int64_t D, int tpr_log2, Op op,
                                                   double* partial) {
  constexpr int K = Op::K;
  __shared__ double sm[256 * K * VEC];
  const int tpr = 1 << tpr_log2, rp = 256 >> tpr_log2;
  const int tx = threadIdx.x & (tpr - 1), ty = threadIdx.x >> tpr_log2;
  const int64_t c0 = ((int64_t)blockIdx.y * tpr + tx) * VEC;
  double a[K][VEC];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int v = 0; v < VEC; ++v) a[k][v] = 0.0;
  if (c0 < D) {
    float c[VEC] = {};
    if constexpr (std::is_same<Op, OpCenteredSq>::value) ld_vec<VEC>(op.center + c0, c);
    // U rows are loaded before any is accumulated (memory-level parallelism: U independent 16-byte
    // loads per input in flight per lane); a thread accumulates its rows in ascending order.
    constexpr int U = kColU;
    using R = typename Op::template Regs<VEC>;
    const int64_t chunk = (int64_t)U * rp;
    for (int64_t rb = (int64_t)blockIdx.x * chunk; rb < N; rb += (int64_t)gridDim.x * chunk) {
      if (rb + chunk <= N) {
        R g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t r = rb + ty + (int64_t)u * rp;
          op.template load<VEC>(r, r * D + c0, g[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) op.template acc<VEC>(g[u], a, c);
      } else {
        for (int64_t r = rb + ty; r < N; r += rp) {
          R g;
          op.template load<VEC>(r, r * D + c0, g);
          op.template acc<VEC>(g, a, c);
        }
      }
    }
  }
  if constexpr (Op::kBroadcastLast) {
#pragma unroll
    for (int v = 1; v < VEC; ++v) a[K - 1][v] = a[K - 1][0];
  }
  double* mine = sm + (size_t)threadIdx.x * K * VEC;
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int v = 0; v < VEC; ++v) mine[k * VEC + v] = a[k][v];
  __syncthreads();
  if (ty == 0 && c0 < D) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        double s = 0.0;
        for (int j = 0; j < rp; ++j) s += sm[((size_t)(j * tpr + tx)) * K * VEC + k * VEC + v];
        partial[((int64_t)blockIdx.x * K + k) * D + c0 + v] = s;
      }
  }
}

// Second stage: lanes over 64 consecutive outputs (coalesced), the 16 waves of a block split the
// slabs (wave w takes slabs w, w+16, ...; 8 loads in flight), LDS combine in wave order.
__global__ __launch_bounds__(1024) void k_colfinal(int64_t nslab, int64_t KD, const double* __restrict__ partial,
                                                   double* __restrict__ out) {
  __shared__ double sm[16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 64 + lane;
  double s = 0.0;
  if (j < KD) {
    constexpr int U = 8;
    int64_t t = wv;
    for (; t + (int64_t)(U - 1) * 16 < nslab; t += (int64_t)U * 16) {
      double v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = partial[(t + (int64_t)u * 16) * KD + j];
#pragma unroll
      for (int u = 0; u < U; ++u) s += v[u];
    }
    for (; t < nslab; t += 16) s += partial[t * KD + j];
  }
  sm[wv][lane] = s;
  __syncthreads();
  if (wv == 0 && j < KD) {
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += sm[w][lane];
    out[j] = tot;
  }
}

template <class Op>
int run_colreduce(hipStream_t stream, int64_t N, int64_t D, bool vec4, Op op, void* workspace,
                  double* out, const char* what) {
  constexpr int K = Op::K;
  if (D == 0) return 0;
  if (N == 0) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double) * K * D, stream);
    if (e != hipSuccess) {
      bjx_set_error("%s: memset failed: %s", what, hipGetErrorString(e));
      return 2;
    }
    return 0;
  }
  const ColGeom g = col_geom(N, D, vec4 ? 4 : 1);
  dim3 grid((unsigned)g.nslab, (unsigned)g.ncb);
  double* partial = (double*)workspace;
  if (vec4)
    hipLaunchKernelGGL((k_colreduce<4, Op>), grid, dim3(256), 0, stream, N, D, g.tpr_log2, op, partial);
  else
    hipLaunchKernelGGL((k_colreduce<1, Op>), grid, dim3(256), 0, stream, N, D, g.tpr_log2, op, partial);
  const int64_t KD = (int64_t)K * D;
  hipLaunchKernelGGL(k_colfinal, dim3((unsigned)((KD + 63) / 64)), dim3(1024), 0, stream, g.nslab, KD,
                     partial, out);
  return bjx_check_launch(what);
}

// ChEES weights fused into the column statistics (chees_adaptation.py:376 + 241-246, 384-386): the
// tpr threads that hold one row of q' between them also decide whether that row has a non-finite
// entry (wave ballot; LDS across the waves of a row when tpr > 64), so q' is read once instead of
// twice.  Needs the whole row in one workgroup (ncb == 1, i.e. D <= 1024 at 16 B per lane); the
// accumulation order is k_colreduce<OpChees>'s, so the two paths agree bit for bit.
template <int VEC, int U>
struct WcolGeom {
  int tpr_log2, tpr, rp, tx, ty, lane, wv;
  int64_t c0, cl, r_hi, D;
  bool col_in;
};

// U rows of q', q and their per-chain scalars as held by one thread.  No branches around the loads:
// rows past the slab shadow its last row and columns past D shadow column 0 (neither is accumulated),
// so the 2 U row loads of a batch issue back to back.
template <int VEC, int U>
__device__ __forceinline__ void wcol_load(const WcolGeom<VEC, U>& g, int64_t rb, const float* __restrict__ qp,
                                          const float* __restrict__ qi, const float* __restrict__ acc,
                                          const uint8_t* __restrict__ is_div, float (&x)[U][VEC],
                                          float (&y)[U][VEC], float (&an)[U], int (&dn)[U]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int64_t r = rb + g.ty + (int64_t)u * g.rp;
    r = r < g.r_hi ? r : g.r_hi - 1;
    an[u] = acc[r];
    dn[u] = is_div[r];
    ld_vec<VEC>(qp + r * g.D + g.cl, x[u]);
    ld_vec<VEC>(qi + r * g.D + g.cl, y[u]);
  }
}

template <int VEC, int U>
__device__ __forceinline__ void wcol_process(const WcolGeom<VEC, U>& g, int64_t rb, unsigned* sbad_par,
                                             const float (&x)[U][VEC], const float (&y)[U][VEC],
                                             const float (&an)[U], const int (&dn)[U],
                                             float* __restrict__ w_out, double (&a)[4][VEC]) {
  unsigned bad = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    bool nf = false;
#pragma unroll
    for (int v = 0; v < VEC; ++v) nf |= !isfinite(x[u][v]);
    nf &= g.col_in;
    const unsigned long long m = __ballot(nf);
    bool any;
    if (g.tpr >= 64) any = m != 0ull;
    else any = ((m >> ((g.lane >> g.tpr_log2) << g.tpr_log2)) & ((1ull << g.tpr) - 1ull)) != 0ull;
    bad |= any ? (1u << u) : 0u;
  }
  if (g.tpr > 64) {  // a row spans tpr / 64 waves
    if (g.lane == 0) sbad_par[g.wv] = bad;
    __syncthreads();
    const int wpr = g.tpr >> 6, w0 = (g.wv / wpr) * wpr;
    for (int j = 0; j < wpr; ++j) bad |= sbad_par[w0 + j];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t r = rb + g.ty + (int64_t)u * g.rp;
    const bool in = r < g.r_hi;
    const float wf = (dn[u] || ((bad >> u) & 1u)) ? 0.0f : an[u];
    if (in && g.tx == 0) w_out[r] = wf;
    if (in) {  // uniform over the threads of a row; a skipped row adds nothing (not even +0.0)
      const double wr = (double)wf;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float xs = isfinite(x[u][v]) ? x[u][v] : 0.0f;
        a[0][v] += wr * (double)xs;
        const bool ok = !(y[u][v] != y[u][v]);
        a[1][v] += ok ? (double)y[u][v] : 0.0;
        a[2][v] += ok ? 1.0 : 0.0;
      }
      a[3][0] += wr;
    }
  }
}

template <int VEC, int U>
__global__ __launch_bounds__(256) void k_chees_wcol(int64_t N, int64_t D, int tpr_log2,
                                                    const float* __restrict__ qp, const float* __restrict__ qi,
                                                    const float* __restrict__ acc,
                                                    const uint8_t* __restrict__ is_div, float* __restrict__ w_out,
                                                    double* __restrict__ partial) {
  constexpr int K = 4;
  __shared__ double sm[256 * K * VEC];
  __shared__ unsigned sbad[2][4];
  WcolGeom<VEC, U> g;
  g.tpr_log2 = tpr_log2;
  g.tpr = 1 << tpr_log2;
  g.rp = 256 >> tpr_log2;
  g.tx = threadIdx.x & (g.tpr - 1);
  g.ty = threadIdx.x >> tpr_log2;
  g.lane = threadIdx.x & 63;
  g.wv = threadIdx.x >> 6;
  g.c0 = (int64_t)g.tx * VEC;
  g.col_in = g.c0 < D;
  g.cl = g.col_in ? g.c0 : 0;
  g.D = D;
  g.r_hi = N;
  const int64_t chunk = (int64_t)U * g.rp, step = (int64_t)gridDim.x * chunk;
  double a[K][VEC];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int v = 0; v < VEC; ++v) a[k][v] = 0.0;
  int64_t rb = (int64_t)blockIdx.x * chunk;  // chunks blockIdx.x, + gridDim.x, ... (col_geom)
  // uniform trip count over the workgroup (wcol_process has a barrier when a row spans waves).  A
  // second batch requested ahead of the reduction (twice the registers) measured the same.
  int par = 0;
  for (; rb < g.r_hi; rb += step, par ^= 1) {
    float x[U][VEC], y[U][VEC], an[U];
    int dn[U];
    wcol_load<VEC, U>(g, rb, qp, qi, acc, is_div, x, y, an, dn);
    wcol_process<VEC, U>(g, rb, sbad[par], x, y, an, dn, w_out, a);
  }
#pragma unroll
  for (int v = 1; v < VEC; ++v) a[K - 1][v] = a[K - 1][0];
  double* mine = sm + (size_t)threadIdx.x * K * VEC;
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int v = 0; v < VEC; ++v) mine[k * VEC + v] = a[k][v];
  __syncthreads();
  if (g.ty == 0 && g.col_in) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        double s = 0.0;
        for (int j = 0; j < g.rp; ++j) s += sm[((size_t)(j * g.tpr + g.tx)) * K * VEC + k * VEC + v];
        partial[((int64_t)blockIdx.x * K + k) * D + g.c0 + v] = s;
      }
  }
}

// The same pass for rows of 129 ... 1 024 floats with whole rows per WAVE (lane l owns the 16-byte pieces
// l, l + 64, ...): the non-finite test of a row is one wave ballot, its scalars are wave-uniform
// (scalar loads), and nothing synchronises inside the row loop -- the four waves of a workgroup only
// meet at the end, where they add their column sums into LDS one after the other (fixed order).
// Measured at 65 536 x 1 024 against k_chees_wcol (a row spread over four waves, one barrier per eight
// rows): see NOTEBOOK.md section 11.
// FULL: D == NI * 256, every lane owns NI valid pieces -- no exec-masked branches around the loads, so
// the compiler counts the outstanding loads exactly (with them it falls back to vmcnt(0) and the
// pipeline degenerates).
template <int NI, bool FULL, bool NT>
__global__ __launch_bounds__(256) void k_chees_wrow(int64_t N, int64_t D, const float* __restrict__ qp,
                                                    const float* __restrict__ qi, const float* __restrict__ acc,
                                                    const uint8_t* __restrict__ is_div, float* __restrict__ w_out,
                                                    double* __restrict__ partial) {
  __shared__ double sm[3][NI * 256];
  __shared__ double sm_w;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n_waves = (int)gridDim.x * 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wv);
  const int n = (int)N;
  bool ok[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) ok[k] = FULL || ((int64_t)lane + 64 * k) * 4 < D;
  double a0[NI][4], a1[NI][4], aw = 0.0;
  int a2[NI][4];  // counts of finite initial entries: exact integers (converted once at the end)
#pragma unroll
  for (int k = 0; k < NI; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a0[k][e] = a1[k][e] = 0.0;
      a2[k][e] = 0;
    }
  // Round 3: an explicit two-stage pipeline (two register sets, A and B, no copies between them): the
  // loads of row r + n_waves are in flight while row r is accumulated, so a wave always has 8 KB
  // (D = 1 024) outstanding instead of alternating between a load phase and a ~1 500-cycle arithmetic
  // phase -- with two waves per SIMD (the accumulators need ~190 VGPRs) nothing else hides that.
  // Nontemporal loads: both arrays are read once (tools/membw2.hip: 6.45 -> 6.79 TB/s for this mix).
  // A wave still takes rows wave, wave + n_waves, ... in ascending order: the sums are those of the
  // round-2 kernel bit for bit.
  struct Stage {
    F4 x[NI], y[NI];
    float an;
    int dn;
  };
  auto issue = [&](Stage& st, int r) {
    st.an = acc[r];
    st.dn = is_div[r];
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (FULL || ok[k]) {
        const int64_t at = (int64_t)r * D + ((int64_t)lane + 64 * k) * 4;
        st.x[k] = ld4_t<NT>(qp + at);
        st.y[k] = ld4_t<NT>(qi + at);
      }
  };
  auto consume = [&](const Stage& st, int r) {
    bool nf = false;
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (FULL || ok[k]) nf |= !(isfinite(st.x[k].x) && isfinite(st.x[k].y) && isfinite(st.x[k].z) && isfinite(st.x[k].w));
    const bool bad = __any(nf);
    const float wf = (st.dn || bad) ? 0.0f : st.an;
    if (lane == 0) w_out[r] = wf;
    const double wr = (double)wf;
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (FULL || ok[k]) {
        const float xv[4] = {st.x[k].x, st.x[k].y, st.x[k].z, st.x[k].w};
        const float yv[4] = {st.y[k].x, st.y[k].y, st.y[k].z, st.y[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xs = isfinite(xv[e]) ? xv[e] : 0.0f;
          a0[k][e] += wr * (double)xs;
          const bool fin = !(yv[e] != yv[e]);
          a1[k][e] += fin ? (double)yv[e] : 0.0;
          a2[k][e] += fin ? 1 : 0;
        }
      }
    aw += wr;
  };
  Stage A, B;
  // Every iteration issues its loads UNCONDITIONALLY (past the end it re-requests the last row and
  // drops it): a conditional issue gives the two paths different outstanding-load counts, the compiler
  // then waits for the smaller one and the row just requested is waited for before the previous one is
  // consumed.
  const int last = n - 1;
  int r = wave;
  if (r < n) {
    issue(A, r);
    for (;;) {
      const int r1 = r + n_waves;
      issue(B, r1 < n ? r1 : last);
      consume(A, r);
      if (r1 >= n) break;
      const int r2 = r1 + n_waves;
      issue(A, r2 < n ? r2 : last);
      consume(B, r1);
      if (r2 >= n) break;
      r = r2;
    }
  }
  for (int w = 0; w < 4; ++w) {  // the four waves add their sums in wave order
    if (wv == w) {
#pragma unroll
      for (int k = 0; k < NI; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = (lane + 64 * k) * 4 + e;
          if (w == 0) {
            sm[0][c] = a0[k][e]; sm[1][c] = a1[k][e]; sm[2][c] = (double)a2[k][e];
          } else {
            sm[0][c] += a0[k][e]; sm[1][c] += a1[k][e]; sm[2][c] += (double)a2[k][e];
          }
        }
      if (lane == 0) sm_w = (w == 0) ? aw : sm_w + aw;
    }
    __syncthreads();
  }
  double* out = partial + (int64_t)blockIdx.x * 4 * D;
  for (int64_t c = threadIdx.x; c < D; c += 256) {
    out[c] = sm[0][c];
    out[D + c] = sm[1][c];
    out[2 * D + c] = sm[2][c];
    out[3 * D + c] = sm_w;
  }
}

// ------------------------------------------------------------------------------ row kernels
template <int VEC>
__global__ __launch_bounds__(256) void k_chees_weights(int64_t N, int64_t D, const float* __restrict__ qp,
                                                       const float* __restrict__ acc,
                                                       const uint8_t* __restrict__ is_div,
                                                       float* __restrict__ w) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // RW rows per wave and iteration, all their loads issued before any is tested: a 4 KB row alone
  // is too little work per wave to keep HBM busy
  constexpr int RW = 4;
  for (int64_t n0 = wave * RW; n0 < N; n0 += nwaves * RW) {
    bool bad[RW];
    float a_n[RW];
    uint8_t d_n[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {  // per-chain scalars requested up front, not after the row sweep
      bad[r] = false;
      const bool in = n0 + r < N;
      a_n[r] = in ? acc[n0 + r] : 0.0f;
      d_n[r] = in ? is_div[n0 + r] : (uint8_t)1;
    }
    for (int64_t c = (int64_t)lane * VEC; c < D; c += 64 * VEC) {
      float x[RW][VEC];
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        if (n0 + r < N) {
          ld_vec<VEC>(qp + (n0 + r) * D + c, x[r]);
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) x[r][v] = 0.0f;
        }
      }
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int v = 0; v < VEC; ++v) bad[r] |= !isfinite(x[r][v]);
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const bool any_bad = __any(bad[r]);
      if (lane == 0 && n0 + r < N) w[n0 + r] = (d_n[r] || any_bad) ? 0.0f : a_n[r];
    }
  }
}

// weights for short rows (D <= 128, D % 4 == 0): G lanes per row, 64 / G rows per wave
template <int G>
__global__ __launch_bounds__(256) void k_chees_weights_short(int64_t N, int64_t D, const float* __restrict__ qp,
                                                             const float* __restrict__ acc,
                                                             const uint8_t* __restrict__ is_div,
                                                             float* __restrict__ w) {
  constexpr int R = 64 / G;
  const int lane = threadIdx.x & 63;
  const int sub = lane / G, gl = lane % G;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t n0 = wave * R; n0 < N; n0 += nwaves * R) {
    const int64_t n = n0 + sub;
    const bool valid = n < N;
    int bad = 0;
    if (valid)
      for (int64_t c = (int64_t)gl * 4; c < D; c += G * 4) {
        float x[4];
        ld_vec<4>(qp + n * D + c, x);
        bad |= !(isfinite(x[0]) && isfinite(x[1]) && isfinite(x[2]) && isfinite(x[3]));
      }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) bad |= __shfl_xor(bad, o, 64);
    if (valid && gl == 0) w[n] = (is_div[n] || bad) ? 0.0f : acc[n];
  }
}

// The same criterion for rows of exactly NI KiB-pieces per lane (D == 256 NI, 16-byte aligned), one
// wave per row: EVERY load of the row -- 3 NI nontemporal 16-byte row pieces and the shared vectors --
// is issued before the first use (the general kernel's run-time loop keeps 2 in flight), and the rows
// stream past the caches (round 3; tools/membw2.hip read-only ceilings).  Per-lane accumulation order
// and the butterfly are those of k_chees_criterion<4, W, 64>: identical results.
template <int NI, bool WHITEN, bool NT>
__global__ __launch_bounds__(256) void k_chees_criterion_rows(
    int64_t N, int64_t D, const float* __restrict__ qp, const float* __restrict__ pp,
    const float* __restrict__ qi, const float* __restrict__ pm, const float* __restrict__ im,
    const float* __restrict__ imm, const float* __restrict__ isq, float* __restrict__ crit) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int64_t base = n * D;
  F4 a[NI], b[NI], m[NI], ma[NI], mb[NI], sg[NI], sq[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int64_t c = ((int64_t)lane + 64 * k) * 4;
    a[k] = ld4_t<NT>(qp + base + c);
    b[k] = ld4_t<NT>(qi + base + c);
    m[k] = ld4_t<NT>(pp + base + c);
    ma[k] = ld4(pm + c);
    mb[k] = ld4(im + c);
    if constexpr (WHITEN) {
      sg[k] = ld4(imm + c);
      sq[k] = ld4(isq + c);
    }
  }
  double s_pp = 0.0, s_ii = 0.0, s_pv = 0.0;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const float av[4] = {a[k].x, a[k].y, a[k].z, a[k].w}, bv[4] = {b[k].x, b[k].y, b[k].z, b[k].w};
    const float mv[4] = {m[k].x, m[k].y, m[k].z, m[k].w};
    const float mav[4] = {ma[k].x, ma[k].y, ma[k].z, ma[k].w}, mbv[4] = {mb[k].x, mb[k].y, mb[k].z, mb[k].w};
    float sgv[4] = {1, 1, 1, 1}, sqv[4] = {1, 1, 1, 1};
    if constexpr (WHITEN) {
      sgv[0] = sg[k].x; sgv[1] = sg[k].y; sgv[2] = sg[k].z; sgv[3] = sg[k].w;
      sqv[0] = sq[k].x; sqv[1] = sq[k].y; sqv[2] = sq[k].z; sqv[3] = sq[k].w;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float pc = av[v] - mav[v];
      float ic = bv[v] - mbv[v];
      float vel = mv[v];
      if constexpr (WHITEN) {
        pc = pc * sqv[v];
        ic = ic * sqv[v];
        vel = (vel * sgv[v]) * sqv[v];
      }
      s_pp += (double)pc * (double)pc;
      s_ii += (double)ic * (double)ic;
      s_pv += (double)pc * (double)vel;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s_pp += __shfl_xor(s_pp, o, 64);
    s_ii += __shfl_xor(s_ii, o, 64);
    s_pv += __shfl_xor(s_pv, o, 64);
  }
  if (lane == 0) {
    const float diff = (float)s_pp - (float)s_ii;
    crit[n] = diff * (float)s_pv;
  }
}

// G lanes per chain row (64 = one wave per row; 4 ... 32 for rows of at most 16 ... 128 floats, VEC == 4)
template <int VEC, bool WHITEN, int G = 64>
__global__ __launch_bounds__(256) void k_chees_criterion(
    int64_t N, int64_t D, const float* __restrict__ qp, const float* __restrict__ pp,
    const float* __restrict__ qi, const float* __restrict__ pm, const float* __restrict__ im,
    const float* __restrict__ imm, const float* __restrict__ isq, float* __restrict__ crit) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  constexpr int R = 64 / G;
  const int sub = lane / G, gl = lane % G;
  for (int64_t n0 = wave * R; n0 < N; n0 += nwaves * R) {
    const int64_t n = n0 + sub;
    const bool valid = n < N;
    const int64_t base = (valid ? n : N - 1) * D;  // idle groups shadow the last row, write nothing
    double s_pp = 0.0, s_ii = 0.0, s_pv = 0.0;
    for (int64_t c = (int64_t)gl * VEC; c < D; c += G * VEC) {
      float a[VEC], b[VEC], m[VEC], ma[VEC], mb[VEC];
      ld_vec<VEC>(qp + base + c, a);
      ld_vec<VEC>(qi + base + c, b);
      ld_vec<VEC>(pp + base + c, m);
      ld_vec<VEC>(pm + c, ma);
      ld_vec<VEC>(im + c, mb);
      float sg[VEC], sq[VEC];
      if constexpr (WHITEN) {
        ld_vec<VEC>(imm + c, sg);
        ld_vec<VEC>(isq + c, sq);
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float pc = a[v] - ma[v];
        float ic = b[v] - mb[v];
        float vel = m[v];
        if constexpr (WHITEN) {
          pc = pc * sq[v];
          ic = ic * sq[v];
          vel = (vel * sg[v]) * sq[v];
        }
        s_pp += (double)pc * (double)pc;
        s_ii += (double)ic * (double)ic;
        s_pv += (double)pc * (double)vel;
      }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
      s_pp += __shfl_xor(s_pp, o, 64);
      s_ii += __shfl_xor(s_ii, o, 64);
      s_pv += __shfl_xor(s_pv, o, 64);
    }
    if (valid && gl == 0) {
      const float diff = (float)s_pp - (float)s_ii;
      crit[n] = diff * (float)s_pv;
    }
  }
}

__global__ __launch_bounds__(1024) void k_chees_scalars(int64_t N, const float* __restrict__ acc,
                                                        const uint8_t* __restrict__ is_div,
                                                        const float* __restrict__ crit, float scale,
                                                        double* __restrict__ out) {
  __shared__ double sm[4][16];
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  // one block (the result is 4 numbers); U chains per thread are loaded before any is used so the
  // single CU keeps U x 1024 loads in flight
  constexpr int U = 8;
  for (int64_t n0 = threadIdx.x; n0 < N; n0 += (int64_t)1024 * U) {
    float a[U], c[U];
    uint8_t dv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t n = n0 + (int64_t)u * 1024;
      const bool in = n < N;
      dv[u] = in ? is_div[n] : (uint8_t)1;
      a[u] = in ? acc[n] : 0.0f;
      c[u] = (in && crit != nullptr) ? crit[n] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (dv[u]) continue;
      s[0] += (double)(1.0f / a[u]);
      s[1] += 1.0;
      if (crit != nullptr) {
        const float tg = scale * c[u];
        s[2] += (double)a[u] * (double)tg;
      }
      s[3] += (double)(a[u] + 1e-20f);
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double t = wave_sum(s[k]);
    if (lane == 0) sm[k][wv] = t;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0.0;
    for (int j = 0; j < 16; ++j) t += sm[threadIdx.x][j];
    out[threadIdx.x] = t;
  }
}

// ------------------------------------------------------------------------------ D-sized glue
__global__ void k_chees_means(int64_t D, const double* __restrict__ stats, const float* __restrict__ imm,
                              float* __restrict__ pm, float* __restrict__ im, float* __restrict__ isq) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  pm[d] = (float)stats[d] / ((float)stats[3 * D + d] + 1e-20f);
  im[d] = (float)stats[D + d] / (float)stats[2 * D + d];
  if (imm != nullptr && isq != nullptr) isq[d] = 1.0f / sqrtf(imm[d]);
}

__global__ void k_pool_mean(int64_t D, const double* __restrict__ sum, double count, float* __restrict__ mean) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d < D) mean[d] = (float)(sum[d] / count);
}

__global__ void k_pool_merge_diag(int64_t D, float n_a, float n_b, const float* __restrict__ mean_b,
                                  const double* __restrict__ m2_b_sum, float* __restrict__ mean,
                                  float* __restrict__ m2) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  const float n_ab = n_a + n_b;
  const float delta = mean_b[d] - mean[d];
  const float mean_ab = mean[d] + delta * (n_b / n_ab);
  const float coef = (n_a * n_b) / n_ab;
  const float cross = (delta * delta) * coef;
  m2[d] = (m2[d] + (float)m2_b_sum[d]) + cross;
  mean[d] = mean_ab;
}

__global__ void k_pool_final_diag(int64_t D, float count, const float* __restrict__ m2, float* __restrict__ imm) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  const float v = m2[d] / (count - 1.0f);
  imm[d] = (v != v) ? v : fmaxf(v, 1e-20f);
}

template <int VEC>
__global__ __launch_bounds__(256) void k_pool_center(int64_t N, int64_t D, const float* __restrict__ x,
                                                     const float* __restrict__ center,
                                                     float* __restrict__ out) {
  const int64_t per_row = D / VEC;
  const int64_t total = N * per_row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / per_row, c = (i - r * per_row) * VEC;
    float t[VEC], m[VEC];
    ld_vec<VEC>(x + r * D + c, t);
    ld_vec<VEC>(center + c, m);
    if constexpr (VEC == 4) {
      st4(out + r * D + c, F4{t[0] - m[0], t[1] - m[1], t[2] - m[2], t[3] - m[3]});
    } else {
      out[r * D + c] = t[0] - m[0];
    }
  }
}

__global__ void k_halton_steps(int64_t N, const int32_t* __restrict__ arg, int max_bits, float ja, float jb,
                               float num_leapfrog, int32_t* __restrict__ steps) {
  // grid-stride: flat_grid caps the launch at 65 536 workgroups (16.7 M chains per sweep)
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N;
       n += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t i = (uint32_t)arg[n] + 1u;
    float h = 0.0f;
    for (int k = 0; k < max_bits; ++k)
      if ((i >> k) & 1u) h += ldexpf(0.5f, -k);  // exact: distinct powers of two, max_bits <= 24
    const float jitter = h * ja + jb;
    steps[n] = (int32_t)ceilf(jitter * num_leapfrog);
  }
}

inline unsigned flat_grid(int64_t n, int block) {
  int64_t b = (n + block - 1) / block;
  if (b < 1) b = 1;
  if (b > 65536) b = 65536;
  return (unsigned)b;
}

}  // namespace

extern "C" {

int64_t bjx_pool_workspace_bytes(int64_t N, int64_t D) {
  if (N <= 0 || D <= 0) return 0;
  const ColGeom g4 = col_geom(N, D, 4), g1 = col_geom(N, D, 1);
  const int64_t slabs = g4.nslab > g1.nslab ? g4.nslab : g1.nslab;
  return slabs * 4 * D * (int64_t)sizeof(double);
}

int bjx_chees_weights(hipStream_t stream, int64_t N, int64_t D, const float* q_prop, const float* acc,
                      const uint8_t* is_divergent, float* w) {
  BJX_CHECK_ARG(N >= 0 && D >= 0, "bjx_chees_weights: negative size");
  if (N == 0) return 0;
  BJX_CHECK_ARG((q_prop || D == 0) && acc && is_divergent && w, "bjx_chees_weights: null pointer");
  const unsigned grid = bjx_row_grid((N + 3) / 4, 4);  // 4 rows per wave, 4 waves per block
  if (bjx_vec4_ok(D, q_prop) && D > 0 && D <= 128) {
#define BJX_W_SHORT(G_)                                                                                \
  hipLaunchKernelGGL(k_chees_weights_short<G_>, dim3(bjx_row_grid((N * G_ + 63) / 64, 4)), dim3(256), 0, \
                     stream, N, D, q_prop, acc, is_divergent, w)
    if (D <= 16) BJX_W_SHORT(4);
    else if (D <= 32) BJX_W_SHORT(8);
    else if (D <= 64) BJX_W_SHORT(16);
    else BJX_W_SHORT(32);
#undef BJX_W_SHORT
  } else if (bjx_vec4_ok(D, q_prop))
    hipLaunchKernelGGL(k_chees_weights<4>, dim3(grid), dim3(256), 0, stream, N, D, q_prop, acc,
                       is_divergent, w);
  else
    hipLaunchKernelGGL(k_chees_weights<1>, dim3(grid), dim3(256), 0, stream, N, D, q_prop, acc,
                       is_divergent, w);
  return bjx_check_launch("bjx_chees_weights");
}

int bjx_chees_colstats(hipStream_t stream, int64_t N, int64_t D, const float* q_prop, const float* w,
                       const float* q_init, void* workspace, double* stats) {
  BJX_CHECK_ARG(N >= 0 && D >= 0, "bjx_chees_colstats: negative size");
  BJX_CHECK_ARG(stats || D == 0, "bjx_chees_colstats: null stats");
  BJX_CHECK_ARG(N == 0 || D == 0 || (q_prop && w && q_init && workspace), "bjx_chees_colstats: null pointer");
  OpChees op{q_prop, w, q_init};
  return run_colreduce(stream, N, D, bjx_vec4_ok(D, q_prop, q_init), op, workspace, stats,
                       "bjx_chees_colstats");
}

int bjx_chees_weights_colstats(hipStream_t stream, int64_t N, int64_t D, const float* q_prop,
                               const float* acc, const uint8_t* is_divergent, const float* q_init,
                               float* w, void* workspace, double* stats) {
  BJX_CHECK_ARG(N >= 0 && D >= 0, "bjx_chees_weights_colstats: negative size");
  BJX_CHECK_ARG(stats || D == 0, "bjx_chees_weights_colstats: null stats");
  BJX_CHECK_ARG(N == 0 || (acc && is_divergent && w), "bjx_chees_weights_colstats: null pointer");
  BJX_CHECK_ARG(N == 0 || D == 0 || (q_prop && q_init && workspace), "bjx_chees_weights_colstats: null pointer");
  const bool vec4 = bjx_vec4_ok(D, q_prop, q_init);
  const ColGeom g = (N > 0 && D > 0) ? col_geom(N, D, vec4 ? 4 : 1) : ColGeom{0, 2, 1};
  static const bool unfused = getenv("BJX_CHEES_UNFUSED") && atoi(getenv("BJX_CHEES_UNFUSED")) != 0;
  if (N == 0 || D == 0 || g.ncb != 1 || unfused) {  // a row does not fit one workgroup: two passes
    int rc = bjx_chees_weights(stream, N, D, q_prop, acc, is_divergent, w);
    if (rc) return rc;
    return bjx_chees_colstats(stream, N, D, q_prop, w, q_init, workspace, stats);
  }
  double* partial = (double*)workspace;
  int64_t nslab = g.nslab;
  static const bool by_rows = !(getenv("BJX_CHEES_WCOL") && atoi(getenv("BJX_CHEES_WCOL")) != 0);
  if (vec4 && D > 128 && D <= 1024 && N < ((int64_t)1 << 31) && by_rows) {  // whole rows per wave
    // nontemporal loads when the two arrays exceed the Infinity Cache (tools/membw2.hip: 6.45 -> 6.79
    // TB/s for this mix).  They leave nothing of q' / q in the cache for the criterion kernel that
    // follows, which therefore streams with nontemporal loads too (k_chees_criterion_rows); with plain
    // loads here that kernel found half of its input cached (round 3 A/B: 136 + 128 us plain / plain,
    // 96 + 163 us nontemporal / plain).  BJX_CHEES_NT=0 / 1 forces plain / nontemporal in both.
    static const int chees_nt_mode = [] { const char* e = getenv("BJX_CHEES_NT"); return e ? atoi(e) : -1; }();
    const bool chees_nt = chees_nt_mode < 0 ? N * D * 8 > ((int64_t)256 << 20) : chees_nt_mode != 0;
    const int u = D > 512 ? 2 : 4;
    int64_t wgs = (N + 4 * u - 1) / (4 * u);
    nslab = wgs < 512 ? wgs : 512;  // <= the slab count bjx_pool_workspace_bytes sizes the partials for
#define BJX_WROW(NI_)                                                                                   \
  do {                                                                                                  \
    if (D == (NI_) * 256 && chees_nt)                                                                   \
      hipLaunchKernelGGL((k_chees_wrow<NI_, true, true>), dim3((unsigned)nslab), dim3(256), 0, stream, N, D, \
                         q_prop, q_init, acc, is_divergent, w, partial);                                \
    else if (D == (NI_) * 256)                                                                          \
      hipLaunchKernelGGL((k_chees_wrow<NI_, true, false>), dim3((unsigned)nslab), dim3(256), 0, stream, N, D, \
                         q_prop, q_init, acc, is_divergent, w, partial);                                \
    else                                                                                                \
      hipLaunchKernelGGL((k_chees_wrow<NI_, false, false>), dim3((unsigned)nslab), dim3(256), 0, stream, N, D, \
                         q_prop, q_init, acc, is_divergent, w, partial);                                \
  } while (0)
    if (D <= 256) BJX_WROW(1);
    else if (D <= 512) BJX_WROW(2);
    else BJX_WROW(4);
#undef BJX_WROW
  } else {
#define BJX_WCOL(V)                                                                                         \
  hipLaunchKernelGGL((k_chees_wcol<V, kColU>), dim3((unsigned)g.nslab), dim3(256), 0, stream, N, D, g.tpr_log2, \
                     q_prop, q_init, acc, is_divergent, w, partial)
    if (vec4) BJX_WCOL(4);
    else BJX_WCOL(1);
#undef BJX_WCOL
  }
  const int64_t KD = 4 * D;
  hipLaunchKernelGGL(k_colfinal, dim3((unsigned)((KD + 63) / 64)), dim3(1024), 0, stream, nslab, KD,
                     partial, stats);
  return bjx_check_launch("bjx_chees_weights_colstats");
}

int bjx_chees_means(hipStream_t stream, int64_t D, const double* stats, const float* imm,
                    float* proposals_mean, float* initials_mean, float* inv_sqrt_imm) {
  BJX_CHECK_ARG(D >= 0, "bjx_chees_means: negative size");
  if (D == 0) return 0;
  BJX_CHECK_ARG(stats && proposals_mean && initials_mean, "bjx_chees_means: null pointer");
  hipLaunchKernelGGL(k_chees_means, dim3(flat_grid(D, 256)), dim3(256), 0, stream, D, stats, imm,
                     proposals_mean, initials_mean, inv_sqrt_imm);
  return bjx_check_launch("bjx_chees_means");
}

int bjx_chees_criterion(hipStream_t stream, int64_t N, int64_t D, const float* q_prop,
                        const float* p_prop, const float* q_init, const float* proposals_mean,
                        const float* initials_mean, const float* imm, const float* inv_sqrt_imm,
                        float* crit) {
  BJX_CHECK_ARG(N >= 0 && D >= 0, "bjx_chees_criterion: negative size");
  if (N == 0) return 0;
  BJX_CHECK_ARG(crit && (D == 0 || (q_prop && p_prop && q_init && proposals_mean && initials_mean)),
                "bjx_chees_criterion: null pointer");
  BJX_CHECK_ARG((imm == nullptr) == (inv_sqrt_imm == nullptr),
                "bjx_chees_criterion: imm and inv_sqrt_imm must both be given or both be NULL");
  const unsigned grid = bjx_row_grid(N, 4);
  const bool v4 = bjx_vec4_ok(D, q_prop, p_prop, q_init, proposals_mean, initials_mean, imm, inv_sqrt_imm);
#define BJX_LAUNCH_CRIT(V, W)                                                                         \
  hipLaunchKernelGGL((k_chees_criterion<V, W>), dim3(grid), dim3(256), 0, stream, N, D, q_prop,       \
                     p_prop, q_init, proposals_mean, initials_mean, imm, inv_sqrt_imm, crit)
#define BJX_LAUNCH_CRIT_G(W, G_)                                                                          \
  hipLaunchKernelGGL((k_chees_criterion<4, W, G_>), dim3(bjx_row_grid((N * G_ + 63) / 64, 4)), dim3(256), 0, \
                     stream, N, D, q_prop, p_prop, q_init, proposals_mean, initials_mean, imm, inv_sqrt_imm, crit)
#define BJX_LAUNCH_CRIT_SHORT(W)              \
  do {                                        \
    if (D <= 16) BJX_LAUNCH_CRIT_G(W, 4);     \
    else if (D <= 32) BJX_LAUNCH_CRIT_G(W, 8); \
    else if (D <= 64) BJX_LAUNCH_CRIT_G(W, 16); \
    else BJX_LAUNCH_CRIT_G(W, 32);            \
  } while (0)
  // rows of 1 / 2 / 4 KiB-pieces per lane: every load up front; nontemporal when the three arrays
  // exceed the Infinity Cache (BJX_CHEES_NT=0 / 1 forces plain / nontemporal)
  static const int nt_mode = [] { const char* e = getenv("BJX_CHEES_NT"); return e ? atoi(e) : -1; }();
  static const bool rows_ok = !(getenv("BJX_CHEES_CRIT_ROWS") && atoi(getenv("BJX_CHEES_CRIT_ROWS")) == 0);
  if (rows_ok && v4 && (D == 256 || D == 512 || D == 1024)) {
    const bool nt = nt_mode < 0 ? N * D * 12 > ((int64_t)256 << 20) : nt_mode != 0;
    const dim3 rg((unsigned)((N + 3) / 4));
#define BJX_CRIT_ROWS(NI_, W_, T_)                                                                      \
  hipLaunchKernelGGL((k_chees_criterion_rows<NI_, W_, T_>), rg, dim3(256), 0, stream, N, D, q_prop, p_prop, \
                     q_init, proposals_mean, initials_mean, imm, inv_sqrt_imm, crit)
#define BJX_CRIT_ROWS_W(NI_)                                                          \
  do {                                                                                \
    if (imm != nullptr) { if (nt) BJX_CRIT_ROWS(NI_, true, true); else BJX_CRIT_ROWS(NI_, true, false); } \
    else { if (nt) BJX_CRIT_ROWS(NI_, false, true); else BJX_CRIT_ROWS(NI_, false, false); }               \
  } while (0)
    if (D == 256) BJX_CRIT_ROWS_W(1);
    else if (D == 512) BJX_CRIT_ROWS_W(2);
    else BJX_CRIT_ROWS_W(4);
#undef BJX_CRIT_ROWS_W
#undef BJX_CRIT_ROWS
  } else if (v4 && D > 0 && D <= 128) {  // short rows: several chains per wave
    if (imm != nullptr) BJX_LAUNCH_CRIT_SHORT(true); else BJX_LAUNCH_CRIT_SHORT(false);
  } else if (imm != nullptr) {
    if (v4) BJX_LAUNCH_CRIT(4, true); else BJX_LAUNCH_CRIT(1, true);
  } else {
    if (v4) BJX_LAUNCH_CRIT(4, false); else BJX_LAUNCH_CRIT(1, false);
  }
#undef BJX_LAUNCH_CRIT_SHORT
#undef BJX_LAUNCH_CRIT_G
#undef BJX_LAUNCH_CRIT
  return bjx_check_launch("bjx_chees_criterion");
}

int bjx_chees_scalars(hipStream_t stream, int64_t N, const float* acc, const uint8_t* is_divergent,
                      const float* crit, float scale, double* out) {
  BJX_CHECK_ARG(N >= 0 && out, "bjx_chees_scalars: bad arguments");
  BJX_CHECK_ARG(N == 0 || (acc && is_divergent), "bjx_chees_scalars: null pointer");
  hipLaunchKernelGGL(k_chees_scalars, dim3(1), dim3(1024), 0, stream, N, acc, is_divergent, crit, scale,
                     out);
  return bjx_check_launch("bjx_chees_scalars");
}

int bjx_pool_colsum(hipStream_t stream, int64_t N, int64_t D, const float* x, const float* center,
                    void* workspace, double* out) {
  BJX_CHECK_ARG(N >= 0 && D >= 0, "bjx_pool_colsum: negative size");
  BJX_CHECK_ARG(out || D == 0, "bjx_pool_colsum: null out");
  BJX_CHECK_ARG(N == 0 || D == 0 || (x && workspace), "bjx_pool_colsum: null pointer");
  if (center != nullptr) {
    OpCenteredSq op{x, center};
    return run_colreduce(stream, N, D, bjx_vec4_ok(D, x, center), op, workspace, out, "bjx_pool_colsum");
  }
  OpSum op{x};
  return run_colreduce(stream, N, D, bjx_vec4_ok(D, x), op, workspace, out, "bjx_pool_colsum");
}

int bjx_pool_mean(hipStream_t stream, int64_t D, const double* sum, double count, float* mean) {
  BJX_CHECK_ARG(D >= 0, "bjx_pool_mean: negative size");
  if (D == 0) return 0;
  BJX_CHECK_ARG(sum && mean, "bjx_pool_mean: null pointer");
  hipLaunchKernelGGL(k_pool_mean, dim3(flat_grid(D, 256)), dim3(256), 0, stream, D, sum, count, mean);
  return bjx_check_launch("bjx_pool_mean");
}

int bjx_pool_merge_diag(hipStream_t stream, int64_t D, float n_a, float n_b, const float* mean_b,
                        const double* m2_b_sum, float* mean, float* m2) {
  BJX_CHECK_ARG(D >= 0, "bjx_pool_merge_diag: negative size");
  if (D == 0) return 0;
  BJX_CHECK_ARG(mean_b && m2_b_sum && mean && m2, "bjx_pool_merge_diag: null pointer");
  BJX_CHECK_ARG(n_a + n_b > 0.0f, "bjx_pool_merge_diag: empty merge");
  hipLaunchKernelGGL(k_pool_merge_diag, dim3(flat_grid(D, 256)), dim3(256), 0, stream, D, n_a, n_b,
                     mean_b, m2_b_sum, mean, m2);
  return bjx_check_launch("bjx_pool_merge_diag");
}

int bjx_pool_final_diag(hipStream_t stream, int64_t D, float count, const float* m2, float* imm) {
  BJX_CHECK_ARG(D >= 0, "bjx_pool_final_diag: negative size");
  if (D == 0) return 0;
  BJX_CHECK_ARG(m2 && imm, "bjx_pool_final_diag: null pointer");
  hipLaunchKernelGGL(k_pool_final_diag, dim3(flat_grid(D, 256)), dim3(256), 0, stream, D, count, m2, imm);
  return bjx_check_launch("bjx_pool_final_diag");
}

int bjx_pool_center(hipStream_t stream, int64_t N, int64_t D, const float* x, const float* center,
                    float* centered) {
  BJX_CHECK_ARG(N >= 0 && D >= 0, "bjx_pool_center: negative size");
  if (N == 0 || D == 0) return 0;
  BJX_CHECK_ARG(x && center && centered, "bjx_pool_center: null pointer");
  if (bjx_vec4_ok(D, x, center, centered))
    hipLaunchKernelGGL(k_pool_center<4>, dim3(flat_grid(N * (D / 4), 256)), dim3(256), 0, stream, N, D,
                       x, center, centered);
  else
    hipLaunchKernelGGL(k_pool_center<1>, dim3(flat_grid(N * D, 256)), dim3(256), 0, stream, N, D, x,
                       center, centered);
  return bjx_check_launch("bjx_pool_center");
}

int bjx_halton_steps(hipStream_t stream, int64_t N, const int32_t* arg, int32_t max_bits,
                     float jitter_amount, float jitter_offset, float num_leapfrog_steps,
                     int32_t* steps) {
  BJX_CHECK_ARG(N >= 0, "bjx_halton_steps: negative size");
  BJX_CHECK_ARG(max_bits >= 0 && max_bits < 32,
                "bjx_halton_steps: max_bits must be less than bit width of dtype int32 (32)");
  if (N == 0) return 0;
  BJX_CHECK_ARG(arg && steps, "bjx_halton_steps: null pointer");
  hipLaunchKernelGGL(k_halton_steps, dim3(flat_grid(N, 256)), dim3(256), 0, stream, N, arg, (int)max_bits,
                     jitter_amount, jitter_offset, num_leapfrog_steps, steps);
  return bjx_check_launch("bjx_halton_steps");
}

}  // extern "C"
