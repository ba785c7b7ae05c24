// Dense Gaussian-Euclidean metric (gfx950): fp32 MFMA GEMMs for p = L^{-T} z and v = M^{-1} p,
// fused with the leapfrog kick (A-operand prologue) and drift (epilogue).  C ABI in bjx_hip.h.
//
// The reference's dense products are `lax.dot(..., precision="highest")` (blackjax/util.py:23-61):
// full fp32.  CDNA4 has native fp32-input MFMA (v_mfma_f32_32x32x2_f32, exact fp32 fma chain at
// the fp32 vector rate, no TF32), so nothing is down-cast.
//
// GEMM: C[M x Nn] = A'[M x K] * B[K x Nn], A' = optional kick prologue of (P, G); M = chains,
// K = Nn = D, B row-major (D, D) shared by all chains (stays in L2).
// Tiling: 128 x 128 block tile, K-tile 16, 256 threads = 2 x 2 waves, each wave a 64 x 64 tile =
// 2 x 2 MFMA 32x32 accumulators (64 acc VGPRs); operands staged k-major in LDS, double buffered.
#include "../../include/bjx_hip.h"
#include "bjx_device.h"
#include "bjx_host.h"

using namespace bjx;

namespace {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int EPT = BK / 2;  // staged elements per thread and operand: BM*BK/256 = BK*BN/256
constexpr int LDA = BM + 1;  // padded k-major A tile: As[k][m]
constexpr int LDB = BN;      // Bs[k][n]
constexpr int kThreads = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_STORE = 0, EPI_DRIFT = 1 };

struct GemmArgs {
  int64_t M, D;          // rows (chains), K = Nn = D
  const float* A;        // (M, D)  P (or Z)
  const float* G;        // (M, D)  gradient for the kick prologue, or nullptr
  int n_kicks;           // 0, 1, 2
  float eps;             // scalar step size
  const float* eps_pc;   // per-chain step size or nullptr
  float* A_out;          // (M, D) kicked A' (written by column-block 0) or nullptr
  const float* B;        // (D, D) row-major
  float* C;              // EPI_STORE: (M, D) output
  const float* Q_in;     // EPI_DRIFT: q_out = fma(eps, C, Q_in)
  float* Q_out;
  bool b_symmetric = false;  // B is the inverse mass matrix: B[k][n] and B[n][k] are interchangeable,
                             // so complete aligned tiles may take the k-contiguous "TN" kernel
  // General palindromic integrators (integrators.py:104-150): the kicks are p += (eps*kick_a) g
  // [; p += (eps*kick_b) g] and the drift q += (eps*drift) v, `eps*coef` an fp32 product as in the
  // reference.  Velocity Verlet: (0.5, 0.5, 1.0) -- eps*0.5f and eps*1.0f are the values used before
  // the coefficients existed, bit for bit.
  float kick_a = 0.5f, kick_b = 0.5f, drift = 1.0f;
  // Per-chain trajectory length (dynamic HMC): row r advances only while step_idx < n_steps[r];
  // otherwise its momentum and position are copied through untouched.  NULL = every row advances.
  const int32_t* n_steps = nullptr;
  int32_t step_idx = 0;
#ifdef BJX_DENSE_PROBE
  // PROBE builds only (tools/dense_timeline.py): four wall-clock stamps per workgroup -- entry, first K-tile staged,
  // end of the main loop, end of the epilogue.  The shipped library has no such field.
  unsigned long long* probe = nullptr;
#endif
};

__device__ __forceinline__ bool gemm_row_active(const GemmArgs& a, int64_t row) {
  return !a.n_steps || a.step_idx < a.n_steps[row];
}

// FULL: M % BM == 0 and D % BN == 0 (hence D % BK == 0): every tile is complete, no bounds checks.
template <int EPI, bool ALIGNED, bool FULL>
__global__ void __launch_bounds__(kThreads) k_dense_gemm(GemmArgs a) {
  // one LDS block: [2 x A tile][2 x B tile]; re-used as the epilogue's transpose staging area
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * LDA + 2 * BK * LDB];
  float (*As)[BK * LDA] = reinterpret_cast<float (*)[BK * LDA]>(smem);
  float (*Bs)[BK * LDB] = reinterpret_cast<float (*)[BK * LDB]>(smem + 2 * BK * LDA);
  static_assert(2 * BK * LDA + 2 * BK * LDB >= 4 * 32 * 64, "epilogue staging needs 32 KiB");
  static_assert((2 * BK * LDA) % 4 == 0, "B tiles must stay 16-byte aligned");
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (speed only): workgroup b is observed to run on XCD b % 8 and each XCD
  // has its own L2.  All column blocks of one row panel are mapped to the same XCD at adjacent
  // dispatch slots so the panel's P/G tiles are fetched from HBM once and shared through that L2
  // instead of being re-read by every column block from a different XCD.
  const int64_t n_col = (a.D + BN - 1) / BN, n_row = (a.M + BM - 1) / BM;
  const int64_t lin = blockIdx.x;
  int64_t row_blk, col_blk;
  {
    const int64_t full = (n_row / 8) * 8 * n_col;  // blocks covered by complete groups of 8 panels
    if (lin < full) {
      const int64_t xcd = lin % 8, slot = lin / 8;
      col_blk = slot % n_col;
      row_blk = (slot / n_col) * 8 + xcd;
    } else {  // ragged tail: plain row-major order
      const int64_t r = lin - full;
      row_blk = (n_row / 8) * 8 + r / n_col;
      col_blk = r % n_col;
    }
  }
  const int64_t row0 = row_blk * BM;
  const int64_t col0 = col_blk * BN;
  const int64_t D = a.D;

  // global->register staging assignments
  const int a_row = tid >> 1, a_k = (tid & 1) * EPT;              // A tile: 128 rows x BK, EPT k per thread
  const int b_k = tid / (BN / EPT), b_n = (tid % (BN / EPT)) * EPT;  // B tile: BK x 128, EPT n per thread
  const int64_t g_row = row0 + a_row;
  const bool row_ok = FULL || g_row < a.M;
  float ha = 0.0f, hb = 0.0f;
  bool kick_row = false;
  if (a.n_kicks > 0 && row_ok) {
    const float e = a.eps_pc ? a.eps_pc[g_row] : a.eps;
    ha = e * a.kick_a;
    hb = e * a.kick_b;
    kick_row = gemm_row_active(a, g_row);
  }

  // Staging is split (issue early / consume late): load_tiles only ISSUES the global loads of the
  // next K-tile; the kick fma, the A_out store and the LDS writes happen in store_tiles AFTER the
  // MFMA phase of the current tile, so HBM/L2 latency hides under the matrix math.
  float ra[EPT], rg[EPT], rb[EPT];
  int64_t cur_kk = 0;
  auto load_tiles = [&](int64_t k0) {
    cur_kk = k0 + a_k;
#pragma unroll
    for (int e = 0; e < EPT; ++e) { ra[e] = 0.0f; rg[e] = 0.0f; rb[e] = 0.0f; }
    if (row_ok) {
      const int64_t kk = cur_kk;
      const float* ap = a.A + g_row * D + kk;
      if (ALIGNED && (FULL || kk + EPT <= D)) {
#pragma unroll
        for (int v = 0; v < EPT / 4; ++v) {
          const F4 x = ld4(ap + 4 * v);
          ra[4 * v] = x.x; ra[4 * v + 1] = x.y; ra[4 * v + 2] = x.z; ra[4 * v + 3] = x.w;
        }
        if (a.n_kicks > 0) {
          const float* gp = a.G + g_row * D + kk;
#pragma unroll
          for (int v = 0; v < EPT / 4; ++v) {
            const F4 x = ld4(gp + 4 * v);
            rg[4 * v] = x.x; rg[4 * v + 1] = x.y; rg[4 * v + 2] = x.z; rg[4 * v + 3] = x.w;
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          if (kk + e < D) {
            ra[e] = ap[e];
            if (a.n_kicks > 0) rg[e] = a.G[g_row * D + kk + e];
          }
        }
      }
    }
    const int64_t bk = k0 + b_k, bn = col0 + b_n;
    if (FULL || bk < D) {
      const float* bp = a.B + bk * D + bn;
      if (ALIGNED && (FULL || bn + EPT <= D)) {
#pragma unroll
        for (int v = 0; v < EPT / 4; ++v) {
          const F4 x = ld4(bp + 4 * v);
          rb[4 * v] = x.x; rb[4 * v + 1] = x.y; rb[4 * v + 2] = x.z; rb[4 * v + 3] = x.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e)
          if (bn + e < D) rb[e] = bp[e];
      }
    }
  };
  auto store_tiles = [&](int buf) {
    if (a.n_kicks > 0 && row_ok) {  // kick prologue on the staged A values
      if (kick_row) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          ra[e] = fmaf(ha, rg[e], ra[e]);
          if (a.n_kicks == 2) ra[e] = fmaf(hb, rg[e], ra[e]);
        }
      }
      if (a.A_out && col_blk == 0) {
        const int64_t kk = cur_kk;
        float* op = a.A_out + g_row * D + kk;
        if (ALIGNED && (FULL || kk + EPT <= D)) {
#pragma unroll
          for (int v = 0; v < EPT / 4; ++v)
            st4(op + 4 * v, F4{ra[4 * v], ra[4 * v + 1], ra[4 * v + 2], ra[4 * v + 3]});
        } else {
#pragma unroll
          for (int e = 0; e < EPT; ++e)
            if (kk + e < D) op[e] = ra[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) As[buf][(a_k + e) * LDA + a_row] = ra[e];
#pragma unroll
    for (int e = 0; e < EPT; ++e) Bs[buf][b_k * LDB + b_n + e] = rb[e];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int64_t n_tiles = (D + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  const int lm = lane & 31, lk = lane >> 5;
  for (int64_t t = 0; t < n_tiles; ++t) {
    const int buf = (int)(t & 1);
    if (t + 1 < n_tiles) load_tiles((t + 1) * BK);
    const float* as = As[buf];
    const float* bs = Bs[buf];
    // LDS operand reads for the whole K-tile are issued up front (32 VGPRs) so the MFMAs below
    // run back to back instead of waiting on an LDS round trip before every pair.
    float fa0[BK / 2], fa1[BK / 2], fb0[BK / 2], fb1[BK / 2];
    // MFMA step u pairs k = u (lanes 0-31) with k = 8 + u (lanes 32-63): the SAME summation order as
    // k_dense_gemm_tn below, so every shared-matrix product of the engine is one fp32 fmaf chain
    // per output element in the order 0,8,1,9,...,7,15 within each K-tile, whatever kernel a shape
    // dispatches to (oracle/fp.py::mfma_k_order restates it; parity tests are bit-exact).
#pragma unroll
    for (int u = 0; u < BK / 2; ++u) {
      const int kk = u + lk * (BK / 2);
      fa0[u] = as[kk * LDA + wm * 64 + lm];
      fa1[u] = as[kk * LDA + wm * 64 + 32 + lm];
      fb0[u] = bs[kk * LDB + wn * 64 + lm];
      fb1[u] = bs[kk * LDB + wn * 64 + 32 + lm];
    }
#pragma unroll
    for (int u = 0; u < BK / 2; ++u) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[u], fb0[u], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[u], fb1[u], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[u], fb0[u], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[u], fb1[u], acc[1][1], 0, 0, 0);
    }
    // Pin the interleave (LLVM otherwise sinks every LDS read next to its MFMA and waits on it):
    // two K-steps of operand reads up front, then 4 MFMAs per 2 further (paired) LDS reads.
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int u = 0; u < BK / 2; ++u) {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    if (t + 1 < n_tiles) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // epilogue: acc[i][j][4*bq + r] <-> row = 8*bq + 4*(lane/32) + r, col = lane%32 of the 32x32 tile
  if constexpr (ALIGNED) {
    // Transpose through LDS so that global accesses are 16-byte row segments (4 rows x 256 B per
    // wave instruction) instead of 4-byte scalars in 128-B pieces.  Each wave stages 32 rows x 64
    // columns (8 KiB) of its 64 x 64 tile at a time in its private slice of the (now idle) tile
    // buffers; the loop above ended with a workgroup barrier, so nobody still reads them.
    float* stage = smem + wave * (32 * 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int bq = 0; bq < 4; ++bq)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            stage[(bq * 8 + lk * 4 + r) * 64 + j * 32 + lm] = acc[i][j][bq * 4 + r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int c4 = (lane & 15) * 4;
      const int64_t col = col0 + wn * 64 + c4;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int rl = (lane >> 4) + 4 * it;
        const int64_t row = row0 + wm * 64 + i * 32 + rl;
        const F4 c = *reinterpret_cast<const F4*>(stage + rl * 64 + c4);
        if (FULL || (row < a.M && col < D)) {
          if constexpr (EPI == EPI_STORE) {
            st4(a.C + row * D + col, c);
          } else {
            const float e = (a.eps_pc ? a.eps_pc[row] : a.eps) * a.drift;
            const F4 q = ld4(a.Q_in + row * D + col);
            if (gemm_row_active(a, row))
              st4(a.Q_out + row * D + col,
                  F4{fmaf(e, c.x, q.x), fmaf(e, c.y, q.y), fmaf(e, c.z, q.z), fmaf(e, c.w, q.w)});
            else
              st4(a.Q_out + row * D + col, q);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = col0 + wn * 64 + j * 32 + lm;
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + wm * 64 + i * 32 + bq * 8 + lk * 4 + r;
          if (row < a.M && col < D) {
            const float c = acc[i][j][bq * 4 + r];
            if constexpr (EPI == EPI_STORE) {
              a.C[row * D + col] = c;
            } else {
              const float e = (a.eps_pc ? a.eps_pc[row] : a.eps) * a.drift;
              const float q = a.Q_in[row * D + col];
              a.Q_out[row * D + col] = gemm_row_active(a, row) ? fmaf(e, c, q) : q;
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// "TN" variant for complete, 16-byte-aligned tiles: C[m][n] = sum_k A'[m][k] * Bt[n][k] with Bt the
// (Nn x K) row-major matrix -- for v = M^{-1} p that is the inverse mass matrix exactly as the
// reference stores it (linear_map(imm, p) = imm @ p, util.py:58-61), no symmetry assumed.
//
// Both operands have k contiguous in memory, so both tiles are staged ROW-major in LDS
// ([rows][BK], row stride LDK = 20 floats) with 16-byte stores, and an MFMA step u pairs
// k = u (lanes 0-31) with k = 8 + u (lanes 32-63): the 8 operands a lane needs for a whole K-tile
// are 8 consecutive floats of one LDS row = two ds_read_b128 per 32-row operand half (conflict
// free at stride 20) instead of eight ds_read_b32.  Per K-tile and wave: 8 LDS reads feed 32 MFMAs.
constexpr int LDK = BK + 4;

// ------------------------------------------------------------------------------------------
// The "TN" kernel: 128 x 128 block tile, K-tile 16, row-major LDS tiles of stride 20, computed by EIGHT waves
// (512 threads, 2 x 4 waves of 64 x 32).  A CU holds two workgroups = 16 waves, four per SIMD, so the matrix pipe
// finds an MFMA to issue while other waves wait on LDS reads, on the staging stores or at the K-tile barrier.
// Per wave and K-tile: 6 ds_read_b128 feed 16 MFMAs on two independent accumulators.  (Rounds 1-4 also carried a
// four-wave form of this tile -- 64 x 64 per wave, 172 VGPRs, two waves per SIMD, 2 % slower -- behind BJX_DENSE_TN8=0;
// removed in round 5.)
constexpr int kThreads8 = 512;

template <int EPI, int KICKS>
__global__ void __launch_bounds__(kThreads8) k_dense_gemm_tn8(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDK + 2 * BN * LDK];
  static_assert(2 * BM * LDK + 2 * BN * LDK >= 8 * 32 * 32, "epilogue staging needs 32 KiB");
  static_assert(BK == 16, "the k pairing below assumes two 8-wide halves");
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int64_t n_col = a.D / BN, n_row = a.M / BM;
  const int64_t lin = blockIdx.x;
  int64_t row_blk, col_blk;
  {  // XCD-aware tile order, as in k_dense_gemm
    const int64_t full = (n_row / 8) * 8 * n_col;
    if (lin < full) {
      const int64_t xcd = lin % 8, slot = lin / 8;
      col_blk = slot % n_col;
      row_blk = (slot / n_col) * 8 + xcd;
    } else {
      const int64_t r = lin - full;
      row_blk = (n_row / 8) * 8 + r / n_col;
      col_blk = r % n_col;
    }
  }
  const int64_t row0 = row_blk * BM, col0 = col_blk * BN, D = a.D;
  float* As0 = smem;
  float* Bs0 = smem + 2 * BM * LDK;
#ifdef BJX_DENSE_PROBE
  if (a.probe && tid == 0) {
    a.probe[blockIdx.x * 8 + 0] = wall_clock64();
    // HW_REG_LDS_ALLOC (hwreg 6): LDS_BASE in its low bits -- non-zero = this workgroup's LDS block sits above
    // another one's, i.e. it was placed second on its CU
    a.probe[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 6);
    a.probe[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID
  }
#endif

  // staging: thread -> (tile row tid/4, 4 consecutive k starting at (tid&3)*4) of A and of Bt
  const int s_row = tid >> 2, s_k = (tid & 3) * 4;
  const float* a_src = a.A + (row0 + s_row) * D + s_k;
  const float* g_src = KICKS > 0 ? a.G + (row0 + s_row) * D + s_k : nullptr;
  const float* b_src = a.B + (col0 + s_row) * D + s_k;
  float* a_out = (KICKS > 0 && a.A_out) ? a.A_out + (row0 + s_row) * D + s_k : nullptr;
  float ha = 0.0f, hb = 0.0f;
  bool kick_row = true;
  if (KICKS > 0) {
    const float e = a.eps_pc ? a.eps_pc[row0 + s_row] : a.eps;
    ha = e * a.kick_a;
    hb = e * a.kick_b;
    kick_row = gemm_row_active(a, row0 + s_row);
  }
  struct Regs {
    F4 a, g, b;
  };
  Regs R0, R1;
  auto load_tiles = [&](Regs& r, int64_t k0) {
    r.a = ld4(a_src + k0);
    if constexpr (KICKS > 0) r.g = ld4(g_src + k0);
    r.b = ld4(b_src + k0);
  };
  auto kick = [&](F4& x, const F4& g, float h) {
    x.x = fmaf(h, g.x, x.x); x.y = fmaf(h, g.y, x.y); x.z = fmaf(h, g.z, x.z); x.w = fmaf(h, g.w, x.w);
  };
  auto store_tiles = [&](Regs& r, int buf, int64_t k0) {
    if constexpr (KICKS > 0) {
      if (kick_row) {
        kick(r.a, r.g, ha);
        if constexpr (KICKS == 2) kick(r.a, r.g, hb);
      }
    }
    st4(As0 + buf * BM * LDK + s_row * LDK + s_k, r.a);
    st4(Bs0 + buf * BN * LDK + s_row * LDK + s_k, r.b);
    // The store of the kicked momentum goes LAST (round 5): vmcnt counts loads and stores in issue order, so with
    // the store ahead of the B-tile's LDS write the `s_waitcnt vmcnt(0)` for r.b also waited for the store's
    // acknowledgement -- a full memory round trip in front of the K-tile barrier in 8 of 32 K-tiles.  Issued here,
    // nothing waits for it before the next K-tile's staging (whose loads were requested a K-tile earlier).
    if constexpr (KICKS > 0) {
      if (a_out && k0 / BN == col_blk) st4(a_out + k0, r.a);
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  const int64_t n_tiles = D / BK;  // even: D is a multiple of 128
  const int lm = lane & 31, lk = lane >> 5;
  const int a_off = (wm * 64 + lm) * LDK + lk * 8;
  const int b_off = (wn * 32 + lm) * LDK + lk * 8;
  load_tiles(R0, 0);
  load_tiles(R1, BK);
  store_tiles(R0, 0, 0);
  __syncthreads();
#ifdef BJX_DENSE_PROBE
  if (a.probe && tid == 0) a.probe[blockIdx.x * 8 + 1] = wall_clock64();
#endif
  auto tile = [&](int64_t t, Regs& stage, Regs& refill) {
    const int buf = (int)(t & 1);
    const float* as = As0 + buf * BM * LDK + a_off;
    const float* bs = Bs0 + buf * BN * LDK + b_off;
    float fa0[8], fa1[8], fb[8];
    // the operands of the first four k-steps first (measured: no difference to any other order, with or
    // without a scheduling barrier between the halves -- round 3, call 24)
    *reinterpret_cast<F4*>(fa0) = ld4(as);
    *reinterpret_cast<F4*>(fb) = ld4(bs);
    *reinterpret_cast<F4*>(fa1) = ld4(as + 32 * LDK);
    *reinterpret_cast<F4*>(fa0 + 4) = ld4(as + 4);
    *reinterpret_cast<F4*>(fb + 4) = ld4(bs + 4);
    *reinterpret_cast<F4*>(fa1 + 4) = ld4(as + 32 * LDK + 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[u], fb[u], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[u], fb[u], acc[1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the staging work here, behind 8 queued MFMAs
    if (t + 1 < n_tiles) store_tiles(stage, buf ^ 1, (t + 1) * BK);
    if (t + 2 < n_tiles) load_tiles(refill, (t + 2) * BK);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 4; u < 8; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[u], fb[u], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[u], fb[u], acc[1], 0, 0, 0);
    }
    __syncthreads();
  };
  for (int64_t t = 0; t < n_tiles; t += 2) {
    tile(t, R1, R0);
    tile(t + 1, R0, R1);
  }
#ifdef BJX_DENSE_PROBE
  if (a.probe && tid == 0) a.probe[blockIdx.x * 8 + 2] = wall_clock64();
#endif

  // epilogue: each wave transposes its two 32 x 32 accumulator tiles through its private 4 KiB slice
  // of the (now idle) tile buffers so that global accesses are 16-byte row segments.
  // Round 5: (1) all eight q segments of a lane are requested up front, in one batch -- before, every segment was
  // load -> s_waitcnt vmcnt(0) -> fma -> store, and since vmcnt counts stores too each wait also covered the
  // PREVIOUS segment's store: eight dependent memory round trips per wave at the end of the launch, when every
  // workgroup of the one-round grid is in its epilogue; (2) per-row step sizes / trajectory masks are read once per
  // row, ahead of the transposes; (3) a masked row is a select on the result, not a branch, so every segment is one
  // 16-byte store.  Arithmetic unchanged: q' = fma(eps * drift, v, q).
  float* stage = smem + wave * (32 * 32);
  const int c4 = (lane & 7) * 4;
  const int64_t col = col0 + wn * 32 + c4;
  [[maybe_unused]] F4 qv[2][4];
  [[maybe_unused]] float ev[2][4];
  [[maybe_unused]] int32_t nsv[2][4];  // the row's trajectory length (dynamic HMC), INT32_MAX = every row advances
  if constexpr (EPI == EPI_DRIFT) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int64_t row = row0 + wm * 64 + i * 32 + (lane >> 3) + 8 * it;
        qv[i][it] = ld4(a.Q_in + row * D + col);
        ev[i][it] = a.eps_pc ? a.eps_pc[row] : a.eps;
        nsv[i][it] = a.n_steps ? a.n_steps[row] : INT32_MAX;
      }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int bq = 0; bq < 4; ++bq)
#pragma unroll
      for (int r = 0; r < 4; ++r) stage[(bq * 8 + lk * 4 + r) * 32 + lm] = acc[i][bq * 4 + r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = (lane >> 3) + 8 * it;
      const int64_t row = row0 + wm * 64 + i * 32 + rl;
      const F4 c = *reinterpret_cast<const F4*>(stage + rl * 32 + c4);
      if constexpr (EPI == EPI_STORE) {
        st4(a.C + row * D + col, c);
      } else {
        const float e = ev[i][it] * a.drift;
        const F4 q = qv[i][it];
        const bool act = a.step_idx < nsv[i][it];  // gemm_row_active
        const F4 r{fmaf(e, c.x, q.x), fmaf(e, c.y, q.y), fmaf(e, c.z, q.z), fmaf(e, c.w, q.w)};
        st4(a.Q_out + row * D + col, F4{act ? r.x : q.x, act ? r.y : q.y, act ? r.z : q.z, act ? r.w : q.w});
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#ifdef BJX_DENSE_PROBE
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): the stamp follows the last store's acknowledgement
  __syncthreads();
  if (a.probe && tid == 0) a.probe[blockIdx.x * 8 + 3] = wall_clock64();
#endif
}

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / BJX_WAVE;
__device__ __forceinline__ int64_t wave_row0() {
  return (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
}
__device__ __forceinline__ int64_t wave_row_stride() { return (int64_t)gridDim.x * kWavesPerBlock; }

// z = normal(km, (D,)), km = split(chain key, 2)[0]   (util.py:89-90)
__global__ void __launch_bounds__(kBlock)
k_dense_z(Key key, int64_t off, int64_t fold, int64_t N, int64_t D, float* __restrict__ z) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    const Key km = key_child(chain_key(key, (uint64_t)(r + off), fold), 0);
    for (int64_t j = lane; j < D; j += 64) z[r * D + j] = normal_from_bits(key_bits32(km, (uint64_t)j));
  }
}

// ke = 0.5 * sum_j v_j p_j  (metrics.py:263-270), fp64 accumulate
__global__ void __launch_bounds__(kBlock)
k_rowdot_half(int64_t N, int64_t D, const float* __restrict__ v, const float* __restrict__ p,
              float* __restrict__ ke) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    double acc = 0.0;
    for (int64_t j = lane; j < D; j += 64) acc += (double)v[r * D + j] * (double)p[r * D + j];
    acc = wave_sum(acc);
    if (lane == 0) ke[r] = 0.5f * (float)acc;
  }
}

// Metropolis tail for the dense metric: p1 (already fully kicked) and v1 = imm @ p1 are inputs.
__global__ void __launch_bounds__(kBlock)
k_hmc_finish_dense(Key key, int64_t off, int64_t fold, int64_t N, int64_t D, float thr,
                   const float* __restrict__ q0, const float* __restrict__ logp0,
                   const float* __restrict__ g0, const float* __restrict__ ke0,
                   const float* __restrict__ q1, const float* __restrict__ logp1,
                   const float* __restrict__ g1, const float* __restrict__ p1,
                   const float* __restrict__ v1, float* __restrict__ p_end, float* __restrict__ q_out,
                   float* __restrict__ logp_out, float* __restrict__ g_out,
                   float* __restrict__ acc_rate_out, uint8_t* __restrict__ is_acc_out,
                   uint8_t* __restrict__ is_div_out, float* __restrict__ energy_out) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    const int64_t base = r * D;
    double acc = 0.0;
    for (int64_t j = lane; j < D; j += 64) {
      const float p = p1[base + j];
      acc += (double)v1[base + j] * (double)p;
      if (p_end) p_end[base + j] = -1.0f * p;  // flip_momentum (hmc.py:95-112)
    }
    acc = wave_sum(acc);
    const float ke1 = 0.5f * (float)acc;
    const float lp0 = logp0[r], lp1 = logp1[r];
    const float H0 = -lp0 + ke0[r];
    const float H1 = -lp1 + ke1;
    float delta = H0 - H1;
    if (delta != delta) delta = -__builtin_inff();
    const bool is_div = (-delta) > thr;
    const float p_acc = fminf(exp_cr(delta), 1.0f);
    const Key ki = key_child(chain_key(key, (uint64_t)(r + off), fold), 1);
    const bool accept = key_uniform(ki) < p_acc;
    if (lane == 0) {
      logp_out[r] = accept ? lp1 : lp0;
      acc_rate_out[r] = p_acc;
      is_acc_out[r] = accept ? 1 : 0;
      is_div_out[r] = is_div ? 1 : 0;
      energy_out[r] = H1;
    }
    const float* qs = accept ? q1 : q0;
    const float* gs = accept ? g1 : g0;
    for (int64_t j = lane; j < D; j += 64) {
      q_out[base + j] = qs[base + j];
      g_out[base + j] = gs[base + j];
    }
  }
}

// Multinomial HMC with a dense metric (hmc.py:181-248, trajectory.py:170-232): the part of
// k_mhmc_step_diag after the closing kick.  p1 (fully kicked momentum of the new state) and
// v1 = imm @ p1 come from the GEMM; energy, weight, divergence, progressive uniform sampling with
// key fold_in(integrator_key, step) (proposal.py:118-143) and the reservoir copy happen here.
__global__ void __launch_bounds__(kBlock)
k_mhmc_step_dense(Key key, int64_t off, int64_t fold, int64_t N, int64_t D, int64_t step, float thr,
                  const float* __restrict__ logp0, const float* __restrict__ ke0,
                  const float* __restrict__ q, const float* __restrict__ p1,
                  const float* __restrict__ v1, const float* __restrict__ g,
                  const float* __restrict__ logp_new, float* __restrict__ W, float* __restrict__ S,
                  uint8_t* __restrict__ any_div, uint8_t* __restrict__ ever, float* __restrict__ Rq,
                  float* __restrict__ Rp, float* __restrict__ Rg, float* __restrict__ Rlogp,
                  float* __restrict__ Renergy, const int32_t* __restrict__ n_steps) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = wave_row0(); r < N; r += wave_row_stride()) {
    if (n_steps && step >= (int64_t)n_steps[r]) continue;  // this chain's trajectory is complete (dmhmc)
    const int64_t base = r * D;
    double acc = 0.0;
    for (int64_t j = lane; j < D; j += 64) acc += (double)v1[base + j] * (double)p1[base + j];
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
    auto lae = [](float a, float b) {  // np.logaddexp in fp64, rounded once
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
    if (take)
      for (int64_t j = lane; j < D; j += 64) {
        Rq[base + j] = q[base + j];
        Rp[base + j] = p1[base + j];
        Rg[base + j] = g[base + j];
      }
  }
}

// ---------------------------------------------------------------------------------------------
// PER-CHAIN dense metric (one (D, D) matrix per chain: what a vmapped dense window_adaptation
// produces).  Each chain has its own matrix, so this is a batched matrix-vector product, bound by
// reading D^2 words per chain, not by MFMA.  One wave per chain: x (after the optional kick) is
// staged in LDS, lane i accumulates y_i = sum_j M[j][i] x_j in fp64 (row j is read coalesced; for
// the symmetric inverse mass matrix M^T = M, for the momentum draw M = L^{-1} gives L^{-T} z).
// The fp64 accumulation makes this path bit-compatible with the oracle's reductions.
struct PcArgs {
  int64_t N, D;
  const float* M;       // (N, D, D), or one shared (D, D) matrix when m_stride == 0
  int64_t m_stride;     // D*D or 0
  const float* X;       // (N, D)
  const float* G;       // gradient for the kick prologue or nullptr
  int n_kicks;
  float eps;
  const float* eps_pc;
  float* X_out;         // kicked x or nullptr (may alias X)
  float* Y;             // EPI_STORE
  const float* Q_in;    // EPI_DRIFT
  float* Q_out;
  float kick_a = 0.5f, kick_b = 0.5f, drift = 1.0f;  // as in GemmArgs
  const int32_t* n_steps = nullptr;
  int32_t step_idx = 0;
};

template <int EPI>
__global__ void __launch_bounds__(kBlock) k_pc_gemv(PcArgs a) {
  extern __shared__ float pc_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* xs = pc_lds + (int64_t)wave * a.D;
  // uniform trip count across the workgroup so that __syncthreads() is legal
  for (int64_t base = (int64_t)blockIdx.x * kWavesPerBlock; base < a.N;
       base += (int64_t)gridDim.x * kWavesPerBlock) {
    const int64_t c = base + wave;
    const bool ok = c < a.N;
    float e = 0.0f;
    bool active = true;
    if (ok) {
      e = a.eps_pc ? a.eps_pc[c] : a.eps;
      const float ha = e * a.kick_a, hb = e * a.kick_b;
      active = !a.n_steps || a.step_idx < a.n_steps[c];
      for (int64_t j = lane; j < a.D; j += 64) {
        float x = a.X[c * a.D + j];
        if (a.n_kicks > 0) {
          if (active) {
            const float g = a.G[c * a.D + j];
            x = fmaf(ha, g, x);
            if (a.n_kicks == 2) x = fmaf(hb, g, x);
          }
          if (a.X_out) a.X_out[c * a.D + j] = x;
        }
        xs[j] = x;
      }
      e = e * a.drift;
    }
    __syncthreads();
    if (ok) {
      const float* m = a.M + c * a.m_stride;
      for (int64_t i = lane; i < a.D; i += 64) {
        double acc = 0.0;
        for (int64_t j = 0; j < a.D; ++j) acc += (double)m[j * a.D + i] * (double)xs[j];
        const float y = (float)acc;
        if constexpr (EPI == EPI_STORE) a.Y[c * a.D + i] = y;
        else a.Q_out[c * a.D + i] = active ? fmaf(e, y, a.Q_in[c * a.D + i]) : a.Q_in[c * a.D + i];
      }
    }
    __syncthreads();
  }
}

// Welford with a dense M2 (mass_matrix.py:424-435): m2 += outer(updated_delta, delta)
__global__ void __launch_bounds__(kBlock)
k_welford_update_dense(int64_t N, int64_t D, float n, const float* __restrict__ x,
                       const float* mean_in, const float* m2_in, float* mean_out, float* m2_out) {
  extern __shared__ float pc_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* delta = pc_lds + (int64_t)wave * 2 * D;
  float* upd = delta + D;
  for (int64_t base = (int64_t)blockIdx.x * kWavesPerBlock; base < N;
       base += (int64_t)gridDim.x * kWavesPerBlock) {
    const int64_t c = base + wave;
    const bool ok = c < N;
    if (ok) {
      for (int64_t j = lane; j < D; j += 64) {
        const float xv = x[c * D + j], mv = mean_in[c * D + j];
        const float d = xv - mv;
        const float mo = mv + d / n;
        mean_out[c * D + j] = mo;
        delta[j] = d;
        upd[j] = xv - mo;
      }
    }
    __syncthreads();
    if (ok) {
      const float* mi = m2_in + c * D * D;
      float* mo = m2_out + c * D * D;
      for (int64_t i = 0; i < D; ++i) {
        const float u = upd[i];
        for (int64_t j = lane; j < D; j += 64) mo[i * D + j] = fmaf(u, delta[j], mi[i * D + j]);
      }
    }
    __syncthreads();
  }
}

// mass_matrix.py:335-357 (dense): imm = beta_data*cov + beta_prev*prev + beta_ident*1e-3*I
__global__ void __launch_bounds__(kBlock)
k_welford_final_dense(int64_t total, int64_t D, float nm1, float beta_data, float beta_prev,
                      float reg, const float* __restrict__ m2, const float* __restrict__ prev,
                      int64_t prev_per_chain, float* __restrict__ imm_out) {
  const int64_t dd = D * D;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t % dd;
    const float cov = m2[t] / nm1;
    const float pv = prev_per_chain ? prev[t] : prev[r];
    const float diag = (r / D == r % D) ? reg : 0.0f;
    imm_out[t] = fmaf(beta_prev, pv, beta_data * cov) + diag;
  }
}

// Skinny product for FEW rows (the deep doublings of a NUTS transition, the tail of a free-running run):
// C[r][i] = the fp32 fma chain over k of A[r][k] * Bkn[k][i] in the k order of the MFMA kernels (tiles of
// 16, inside a tile k = u, 8 + u for u = 0 .. 7 -- oracle/fp.py::mfma_k_order), starting from +0: bit for
// bit the MFMA kernels' result.  Those kernels walk K in dependent LDS-staged steps of one workgroup per
// 128 x 128 tile -- 18-28 us however few rows there are; here a wave owns 64 columns x kSkinnyRows rows, reads
// the matrix rows coalesced (Bkn[k][i .. i + 63]) with the next tile's 16 rows in flight while it works on
// the current one, and the A values as wave-uniform (scalar) loads: 7 / 11 / 21 us at D = 128 / 256 / 512 for up to
// ~1 000 rows against 11 / 19 / 34 us.  (A variant with both operands staged through LDS, all loads of a 256-wide K
// chunk in flight at once, measured the same within 1 us -- the chain of D dependent fmas and the launch are what is
// left -- and was not kept.)
constexpr int kSkinnyRows = 4;
__global__ void __launch_bounds__(256)
k_dense_skinny(int64_t M, int64_t D, const float* __restrict__ A, const float* __restrict__ Bkn, float* __restrict__ C) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t r0 = ((int64_t)blockIdx.y * 4 + wave) * kSkinnyRows;
  if (r0 >= M) return;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const int64_t ic = i < D ? i : D - 1;  // out-of-range lanes compute column D - 1 again and store nothing
  const float* a[kSkinnyRows];
#pragma unroll
  for (int r = 0; r < kSkinnyRows; ++r) a[r] = A + (r0 + r < M ? r0 + r : M - 1) * D;
  float acc[kSkinnyRows];
#pragma unroll
  for (int r = 0; r < kSkinnyRows; ++r) acc[r] = 0.0f;
  const int64_t n_full = D / 16;
  float b[16], bn[16];
  if (n_full > 0) {
#pragma unroll
    for (int u = 0; u < 16; ++u) b[u] = Bkn[(int64_t)u * D + ic];
  }
  for (int64_t t = 0; t < n_full; ++t) {
    const int64_t k0 = t * 16;
    if (t + 1 < n_full) {
#pragma unroll
      for (int u = 0; u < 16; ++u) bn[u] = Bkn[(k0 + 16 + u) * D + ic];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int r = 0; r < kSkinnyRows; ++r) {
        acc[r] = fmaf(a[r][k0 + u], b[u], acc[r]);
        acc[r] = fmaf(a[r][k0 + 8 + u], b[8 + u], acc[r]);
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) b[u] = bn[u];
  }
  const int64_t k0 = n_full * 16;  // the zero-padded last tile: indices >= D are dropped (fma(0, b, acc) == acc)
  if (k0 < D) {
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t k = k0 + 8 * h + u;
        if (k < D) {
          const float bk = Bkn[k * D + ic];
#pragma unroll
          for (int r = 0; r < kSkinnyRows; ++r) acc[r] = fmaf(a[r][k], bk, acc[r]);
        }
      }
    }
  }
  if (i < D) {
#pragma unroll
    for (int r = 0; r < kSkinnyRows; ++r)
      if (r0 + r < M) C[(r0 + r) * D + i] = acc[r];
  }
}

// rows x D up to which the skinny kernel is used (BJX_DENSE_SKINNY_MAX; 0 = never)
static int64_t skinny_max() {
  static const int64_t v = [] {
    const char* e = getenv("BJX_DENSE_SKINNY_MAX");
    // measured on MI355X (tools/bench_skinny.py, profiles/r04/dense_skinny.json): the crossover with the MFMA kernels
    // sits at 4 096 / 2 048 / 1 024 / 512 rows for D = 128 / 256 / 512 / 1 024
    return e ? atoll(e) : (int64_t)1 << 19;
  }();
  return v;
}
static bool use_skinny(int64_t M, int64_t D) { return M * D <= skinny_max(); }

int launch_skinny(hipStream_t s, int64_t M, int64_t D, const float* A, const float* Bkn, float* C) {
  const dim3 grid((unsigned)((D + 63) / 64), (unsigned)((M + 4 * kSkinnyRows - 1) / (4 * kSkinnyRows)));
  hipLaunchKernelGGL(k_dense_skinny, grid, dim3(256), 0, s, M, D, A, Bkn, C);
  return bjx_check_launch("bjx_dense skinny");
}

int launch_pc(hipStream_t s, int epi, const PcArgs& pa) {
  const dim3 grid(bjx_row_grid(pa.N, kWavesPerBlock)), block(kBlock);
  const size_t lds = (size_t)kWavesPerBlock * pa.D * sizeof(float);
  if (lds > 64 * 1024) {
    bjx_set_error("per-chain dense metric: D = %lld does not fit the LDS staging buffer", (long long)pa.D);
    return 1;
  }
  if (epi == EPI_STORE) hipLaunchKernelGGL(k_pc_gemv<EPI_STORE>, grid, block, lds, s, pa);
  else hipLaunchKernelGGL(k_pc_gemv<EPI_DRIFT>, grid, block, lds, s, pa);
  return bjx_check_launch("bjx_dense_pc gemv");
}

#ifdef BJX_DENSE_PROBE
static unsigned long long* g_probe_buf = nullptr;
#endif

int launch_gemm(hipStream_t s, int epi, const GemmArgs& ga_in) {
  GemmArgs ga = ga_in;
  const dim3 grid((unsigned)(((ga.D + BN - 1) / BN) * ((ga.M + BM - 1) / BM)));
  const bool aligned = bjx_vec4_ok(ga.D, ga.A, ga.G, ga.A_out, ga.B, ga.C, ga.Q_in, ga.Q_out);
  const bool full = (ga.M % BM == 0) && (ga.D % BN == 0);
  if (ga.b_symmetric && full && !aligned) {
    // The host side (blackjax_amd/dense.py::_imm_ptr) hands over the matrix AS STORED for whole tiles and its
    // transposed copy for ragged shapes; whole tiles on the general kernel would read `imm` transposed, which is
    // wrong for a matrix that is symmetric only up to rounding.  Refuse instead of silently transposing.
    bjx_set_error("dense metric: whole %dx%d tiles need 16-byte aligned buffers (N=%lld, D=%lld)", BM, BN,
                  (long long)ga.M, (long long)ga.D);
    return 1;
  }
  if (ga.b_symmetric && aligned && full) {
#ifdef BJX_DENSE_PROBE
    ga.probe = g_probe_buf;
#endif
#define BJX_LAUNCH_TN(E, K) hipLaunchKernelGGL((k_dense_gemm_tn8<E, K>), grid, dim3(kThreads8), 0, s, ga)
    int kicks = ga.G ? ga.n_kicks : 0;
#ifdef BJX_DENSE_PROBE
    // BJX_DENSE_ABLATE (measurement aid, RESULTS INVALID -- compiled only into PROBE builds, `make
    // CXXFLAGS+=-DBJX_DENSE_PROBE`; the shipped library ignores the variable): bit 0 no store of the kicked
    // momentum; bit 1 store-only epilogue (no read of q, no drift); bit 2 no kick prologue (no gradient loads).
    static const int ablate = [] { const char* e = getenv("BJX_DENSE_ABLATE"); return e ? atoi(e) : 0; }();
    if (ablate) {
      if (ablate & 1) ga.A_out = nullptr;
      if ((ablate & 2) && epi == EPI_DRIFT) {
        epi = EPI_STORE;
        ga.C = ga.Q_out;
      }
      if (ablate & 4) kicks = 0;
    }
#endif
    if (epi == EPI_STORE) {
      if (kicks == 0) BJX_LAUNCH_TN(EPI_STORE, 0); else if (kicks == 1) BJX_LAUNCH_TN(EPI_STORE, 1); else BJX_LAUNCH_TN(EPI_STORE, 2);
    } else {
      if (kicks == 0) BJX_LAUNCH_TN(EPI_DRIFT, 0); else if (kicks == 1) BJX_LAUNCH_TN(EPI_DRIFT, 1); else BJX_LAUNCH_TN(EPI_DRIFT, 2);
    }
#undef BJX_LAUNCH_TN
    return bjx_check_launch("bjx_dense gemm (tn)");
  }
  if (epi == EPI_STORE) {
    if (aligned && full) hipLaunchKernelGGL((k_dense_gemm<EPI_STORE, true, true>), grid, dim3(kThreads), 0, s, ga);
    else if (aligned) hipLaunchKernelGGL((k_dense_gemm<EPI_STORE, true, false>), grid, dim3(kThreads), 0, s, ga);
    else hipLaunchKernelGGL((k_dense_gemm<EPI_STORE, false, false>), grid, dim3(kThreads), 0, s, ga);
  } else {
    if (aligned && full) hipLaunchKernelGGL((k_dense_gemm<EPI_DRIFT, true, true>), grid, dim3(kThreads), 0, s, ga);
    else if (aligned) hipLaunchKernelGGL((k_dense_gemm<EPI_DRIFT, true, false>), grid, dim3(kThreads), 0, s, ga);
    else hipLaunchKernelGGL((k_dense_gemm<EPI_DRIFT, false, false>), grid, dim3(kThreads), 0, s, ga);
  }
  return bjx_check_launch("bjx_dense gemm");
}

}  // namespace

extern "C" {

int bjx_dense_matmul(void* stream, int64_t N, int64_t D, const float* A, const float* B, float* C) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && A && B && C, "bjx_dense_matmul: bad arguments");
  GemmArgs ga{N, D, A, nullptr, 0, 0.0f, nullptr, nullptr, B, C, nullptr, nullptr};
  return launch_gemm((hipStream_t)stream, EPI_STORE, ga);
}

int bjx_dense_matmul_bt(void* stream, int64_t N, int64_t D, const float* A, const float* B, const float* Bt,
                        float* C) {
  if (N == 0) return 0;
  BJX_CHECK_ARG(N >= 0 && D > 0 && A && B && Bt && C, "bjx_dense_matmul_bt: bad arguments");
  if (use_skinny(N, D)) return launch_skinny((hipStream_t)stream, N, D, A, B, C);
  GemmArgs ga{N, D, A, nullptr, 0, 0.0f, nullptr, nullptr, B, C, nullptr, nullptr};
  // the kernel that reads its matrix as stored (rows of Bt = columns of B) exists for whole, 16-byte aligned tiles
  if ((N % BM == 0) && (D % BN == 0) && bjx_vec4_ok(D, A, Bt, C)) {
    ga.B = Bt;
    ga.b_symmetric = true;
  }
  return launch_gemm((hipStream_t)stream, EPI_STORE, ga);
}

int bjx_dense_apply_imm(void* stream, int64_t N, int64_t D, const float* P, const float* imm, float* V) {
  if (N == 0) return 0;
  BJX_CHECK_ARG(N >= 0 && D > 0 && P && imm && V, "bjx_dense_apply_imm: bad arguments");
  GemmArgs ga{N, D, P, nullptr, 0, 0.0f, nullptr, nullptr, imm, V, nullptr, nullptr};
  ga.b_symmetric = true;  // B is an inverse mass matrix: the TN kernel reads it as stored
  return launch_gemm((hipStream_t)stream, EPI_STORE, ga);
}

int bjx_dense_apply_imm_t(void* stream, int64_t N, int64_t D, const float* P, const float* imm, const float* imm_t,
                          float* V) {
  if (N == 0) return 0;
  BJX_CHECK_ARG(N >= 0 && D > 0 && P && imm && imm_t && V, "bjx_dense_apply_imm_t: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (use_skinny(N, D)) return launch_skinny(s, N, D, P, imm_t, V);
  GemmArgs ga{N, D, P, nullptr, 0, 0.0f, nullptr, nullptr, imm, V, nullptr, nullptr};
  if ((N % BM == 0) && (D % BN == 0) && bjx_vec4_ok(D, P, imm, V)) ga.b_symmetric = true;  // reads imm as stored
  else ga.B = imm_t;  // the general kernel walks B[k][n]: the transposed copy is what "as stored" means there
  return launch_gemm(s, EPI_STORE, ga);
}

int bjx_hmc_momentum_dense(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                           int64_t step_fold, int64_t N, int64_t D, const float* mass_sqrt_t,
                           const float* imm, float* z_work, float* v_work, float* p_out,
                           float* ke_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && mass_sqrt_t && imm && z_work && v_work && p_out && ke_out,
                "bjx_hmc_momentum_dense: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const dim3 rgrid(bjx_row_grid(N, kWavesPerBlock)), rblock(kBlock);
  hipLaunchKernelGGL(k_dense_z, rgrid, rblock, 0, s, Key{key0, key1}, chain_offset, step_fold, N, D,
                     z_work);
  if (int rc = bjx_check_launch("bjx_hmc_momentum_dense(z)")) return rc;
  GemmArgs g1{N, D, z_work, nullptr, 0, 0.0f, nullptr, nullptr, mass_sqrt_t, p_out, nullptr, nullptr};
  if (int rc = launch_gemm(s, EPI_STORE, g1)) return rc;  // p = L^{-T} z   (metrics.py:260-261)
  GemmArgs g2{N, D, p_out, nullptr, 0, 0.0f, nullptr, nullptr, imm, v_work, nullptr, nullptr};
  g2.b_symmetric = true;
  if (int rc = launch_gemm(s, EPI_STORE, g2)) return rc;  // v = imm p
  hipLaunchKernelGGL(k_rowdot_half, rgrid, rblock, 0, s, N, D, v_work, p_out, ke_out);
  return bjx_check_launch("bjx_hmc_momentum_dense(ke)");
}

int bjx_leapfrog_dense(void* stream, int64_t N, int64_t D, int n_kicks, float eps,
                       const float* eps_per_chain, const float* imm, const float* q_in,
                       const float* p_in, const float* g, float* q_out, float* p_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && imm && q_in && p_in && g && q_out && p_out,
                "bjx_leapfrog_dense: bad arguments");
  BJX_CHECK_ARG(n_kicks == 1 || n_kicks == 2, "bjx_leapfrog_dense: n_kicks must be 1 or 2");
  BJX_CHECK_ARG(p_out != p_in, "bjx_leapfrog_dense: p_out must not alias p_in");
  GemmArgs ga{N, D, p_in, g, n_kicks, eps, eps_per_chain, p_out, imm, nullptr, q_in, q_out};
  ga.b_symmetric = true;
  return launch_gemm((hipStream_t)stream, EPI_DRIFT, ga);
}

int bjx_hmc_finish_dense(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                         int64_t step_fold, int64_t N, int64_t D, float eps,
                         const float* eps_per_chain, const float* imm, float divergence_threshold,
                         const float* q0, const float* logp0, const float* g0, const float* ke0,
                         const float* q1, const float* logp1, const float* g1, const float* p,
                         float* p1_work, float* v_work, float* p_end_out, float* q_out,
                         float* logp_out, float* g_out, float* acceptance_rate_out,
                         uint8_t* is_accepted_out, uint8_t* is_divergent_out, float* energy_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && imm && q0 && logp0 && g0 && ke0 && q1 && logp1 && g1 && p &&
                    p1_work && v_work && q_out && logp_out && g_out && acceptance_rate_out &&
                    is_accepted_out && is_divergent_out && energy_out,
                "bjx_hmc_finish_dense: bad arguments");
  BJX_CHECK_ARG(p1_work != p, "bjx_hmc_finish_dense: p1_work must not alias p");
  hipStream_t s = (hipStream_t)stream;
  // closing half kick fused into the GEMM prologue: p1 = p + (eps/2) g1 ; v1 = imm p1
  GemmArgs ga{N, D, p, g1, 1, eps, eps_per_chain, p1_work, imm, v_work, nullptr, nullptr};
  ga.b_symmetric = true;
  if (int rc = launch_gemm(s, EPI_STORE, ga)) return rc;
  hipLaunchKernelGGL(k_hmc_finish_dense, dim3(bjx_row_grid(N, kWavesPerBlock)), dim3(kBlock), 0, s,
                     Key{key0, key1}, chain_offset, step_fold, N, D, divergence_threshold, q0, logp0,
                     g0, ke0, q1, logp1, g1, p1_work, v_work, p_end_out, q_out, logp_out, g_out,
                     acceptance_rate_out, is_accepted_out, is_divergent_out, energy_out);
  return bjx_check_launch("bjx_hmc_finish_dense");
}

// ------------------------------------------------------------------ per-chain dense metric
int bjx_pc_matvec_t(void* stream, int64_t N, int64_t D, const float* M, int64_t matrix_stride,
                    const float* x, float* y) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && M && x && y && (matrix_stride == 0 || matrix_stride == D * D),
                "bjx_pc_matvec_t: bad arguments");
  PcArgs pa{N, D, M, matrix_stride, x, nullptr, 0, 0.0f, nullptr, nullptr, y, nullptr, nullptr};
  return launch_pc((hipStream_t)stream, EPI_STORE, pa);
}

int bjx_hmc_momentum_dense_pc(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                              int64_t step_fold, int64_t N, int64_t D, const float* mass_sqrt_t,
                              const float* imm, int64_t matrix_stride, float* z_work, float* v_work,
                              float* p_out, float* ke_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && mass_sqrt_t && imm && z_work && v_work && p_out && ke_out &&
                    (matrix_stride == 0 || matrix_stride == D * D),
                "bjx_hmc_momentum_dense_pc: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const dim3 rgrid(bjx_row_grid(N, kWavesPerBlock)), rblock(kBlock);
  hipLaunchKernelGGL(k_dense_z, rgrid, rblock, 0, s, Key{key0, key1}, chain_offset, step_fold, N, D,
                     z_work);
  if (int rc = bjx_check_launch("bjx_hmc_momentum_dense_pc(z)")) return rc;
  PcArgs p1{N, D, mass_sqrt_t, matrix_stride, z_work, nullptr, 0, 0.0f, nullptr, nullptr, p_out, nullptr, nullptr};
  if (int rc = launch_pc(s, EPI_STORE, p1)) return rc;  // p = L^{-T} z
  PcArgs p2{N, D, imm, matrix_stride, p_out, nullptr, 0, 0.0f, nullptr, nullptr, v_work, nullptr, nullptr};
  if (int rc = launch_pc(s, EPI_STORE, p2)) return rc;  // v = imm p
  hipLaunchKernelGGL(k_rowdot_half, rgrid, rblock, 0, s, N, D, v_work, p_out, ke_out);
  return bjx_check_launch("bjx_hmc_momentum_dense_pc(ke)");
}

int bjx_leapfrog_dense_pc(void* stream, int64_t N, int64_t D, int n_kicks, float eps,
                          const float* eps_per_chain, const float* imm, int64_t matrix_stride,
                          const float* q_in, const float* p_in, const float* g, float* q_out,
                          float* p_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && imm && q_in && p_in && g && q_out && p_out &&
                    (matrix_stride == 0 || matrix_stride == D * D),
                "bjx_leapfrog_dense_pc: bad arguments");
  BJX_CHECK_ARG(n_kicks == 1 || n_kicks == 2, "bjx_leapfrog_dense_pc: n_kicks must be 1 or 2");
  PcArgs pa{N, D, imm, matrix_stride, p_in, g, n_kicks, eps, eps_per_chain, p_out, nullptr, q_in, q_out};
  return launch_pc((hipStream_t)stream, EPI_DRIFT, pa);
}

int bjx_hmc_finish_dense_pc(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                            int64_t step_fold, int64_t N, int64_t D, float eps,
                            const float* eps_per_chain, const float* imm, int64_t matrix_stride,
                            float divergence_threshold, const float* q0, const float* logp0,
                            const float* g0, const float* ke0, const float* q1, const float* logp1,
                            const float* g1, const float* p, float* p1_work, float* v_work,
                            float* p_end_out, float* q_out, float* logp_out, float* g_out,
                            float* acceptance_rate_out, uint8_t* is_accepted_out,
                            uint8_t* is_divergent_out, float* energy_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && imm && q0 && logp0 && g0 && ke0 && q1 && logp1 && g1 && p &&
                    p1_work && v_work && q_out && logp_out && g_out && acceptance_rate_out &&
                    is_accepted_out && is_divergent_out && energy_out &&
                    (matrix_stride == 0 || matrix_stride == D * D),
                "bjx_hmc_finish_dense_pc: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  PcArgs pa{N, D, imm, matrix_stride, p, g1, 1, eps, eps_per_chain, p1_work, v_work, nullptr, nullptr};
  if (int rc = launch_pc(s, EPI_STORE, pa)) return rc;  // p1 = p + (eps/2) g1 ; v1 = imm p1
  hipLaunchKernelGGL(k_hmc_finish_dense, dim3(bjx_row_grid(N, kWavesPerBlock)), dim3(kBlock), 0, s,
                     Key{key0, key1}, chain_offset, step_fold, N, D, divergence_threshold, q0, logp0,
                     g0, ke0, q1, logp1, g1, p1_work, v_work, p_end_out, q_out, logp_out, g_out,
                     acceptance_rate_out, is_accepted_out, is_divergent_out, energy_out);
  return bjx_check_launch("bjx_hmc_finish_dense_pc");
}

int bjx_welford_update_dense(void* stream, int64_t N, int64_t D, int64_t sample_size_new,
                             const float* value, const float* mean_in, const float* m2_in,
                             float* mean_out, float* m2_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && sample_size_new >= 1 && value && mean_in && m2_in && mean_out &&
                    m2_out,
                "bjx_welford_update_dense: bad arguments");
  const size_t lds = (size_t)kWavesPerBlock * 2 * D * sizeof(float);
  BJX_CHECK_ARG(lds <= 64 * 1024, "bjx_welford_update_dense: D too large for the LDS staging buffer");
  hipLaunchKernelGGL(k_welford_update_dense, dim3(bjx_row_grid(N, kWavesPerBlock)), dim3(kBlock), lds,
                     (hipStream_t)stream, N, D, (float)sample_size_new, value, mean_in, m2_in,
                     mean_out, m2_out);
  return bjx_check_launch("bjx_welford_update_dense");
}

int bjx_welford_final_dense(void* stream, int64_t N, int64_t D, int64_t sample_size,
                            float imm_shrinkage_to_previous, const float* m2, const float* imm_prev,
                            int imm_prev_per_chain, float* imm_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && sample_size >= 0 && m2 && imm_prev && imm_out,
                "bjx_welford_final_dense: bad arguments");
  const float denom = (float)(sample_size + 5) + imm_shrinkage_to_previous;
  const float beta_data = (float)sample_size / denom;
  const float beta_prev = imm_shrinkage_to_previous / denom;
  const float reg = (5.0f / denom) * 1e-3f;
  const int64_t total = N * D * D;
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(k_welford_final_dense, dim3((unsigned)blocks), dim3(kBlock), 0,
                     (hipStream_t)stream, total, D, (float)(sample_size - 1), beta_data, beta_prev,
                     reg, m2, imm_prev, (int64_t)imm_prev_per_chain, imm_out);
  return bjx_check_launch("bjx_welford_final_dense");
}

// ------------------------------------------------------------------ general coefficients / masks
// matrix_stride < 0: ONE shared matrix on the MFMA GEMM path; 0 or D*D: the fp64-accumulated
// matrix-vector path (shared or per-chain matrices), as in the *_pc entry points.
int bjx_leapfrog_dense_coef(void* stream, int64_t N, int64_t D, int n_kicks, float kick_a, float kick_b,
                            float drift, float eps, const float* eps_per_chain, const float* imm,
                            int64_t matrix_stride, const float* q_in, const float* p_in, const float* g,
                            float* q_out, float* p_out, const int32_t* n_steps, int32_t step_idx) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && imm && q_in && p_in && g && q_out && p_out &&
                    (matrix_stride < 0 || matrix_stride == 0 || matrix_stride == D * D),
                "bjx_leapfrog_dense_coef: bad arguments");
  BJX_CHECK_ARG(n_kicks == 1 || n_kicks == 2, "bjx_leapfrog_dense_coef: n_kicks must be 1 or 2");
  if (matrix_stride < 0) {
    BJX_CHECK_ARG(p_out != p_in, "bjx_leapfrog_dense_coef: p_out must not alias p_in on the GEMM path");
    GemmArgs ga{N, D, p_in, g, n_kicks, eps, eps_per_chain, p_out, imm, nullptr, q_in, q_out};
    ga.b_symmetric = true;
    ga.kick_a = kick_a; ga.kick_b = kick_b; ga.drift = drift;
    ga.n_steps = n_steps; ga.step_idx = step_idx;
    return launch_gemm((hipStream_t)stream, EPI_DRIFT, ga);
  }
  PcArgs pa{N, D, imm, matrix_stride, p_in, g, n_kicks, eps, eps_per_chain, p_out, nullptr, q_in, q_out};
  pa.kick_a = kick_a; pa.kick_b = kick_b; pa.drift = drift;
  pa.n_steps = n_steps; pa.step_idx = step_idx;
  return launch_pc((hipStream_t)stream, EPI_DRIFT, pa);
}

int bjx_hmc_finish_dense_coef(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                              int64_t step_fold, int64_t N, int64_t D, float kick_coef, float eps,
                              const float* eps_per_chain, const float* imm, int64_t matrix_stride,
                              float divergence_threshold, const float* q0, const float* logp0,
                              const float* g0, const float* ke0, const float* q1, const float* logp1,
                              const float* g1, const float* p, float* p1_work, float* v_work,
                              float* p_end_out, float* q_out, float* logp_out, float* g_out,
                              float* acceptance_rate_out, uint8_t* is_accepted_out,
                              uint8_t* is_divergent_out, float* energy_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N >= 0 && D > 0 && imm && q0 && logp0 && g0 && ke0 && q1 && logp1 && g1 && p &&
                    p1_work && v_work && q_out && logp_out && g_out && acceptance_rate_out &&
                    is_accepted_out && is_divergent_out && energy_out && p1_work != p &&
                    (matrix_stride < 0 || matrix_stride == 0 || matrix_stride == D * D),
                "bjx_hmc_finish_dense_coef: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  // closing kick fused into the product's prologue: p1 = p + (eps*kick_coef) g1 ; v1 = imm p1
  if (matrix_stride < 0) {
    GemmArgs ga{N, D, p, g1, 1, eps, eps_per_chain, p1_work, imm, v_work, nullptr, nullptr};
    ga.b_symmetric = true;
    ga.kick_a = kick_coef;
    if (int rc = launch_gemm(s, EPI_STORE, ga)) return rc;
  } else {
    PcArgs pa{N, D, imm, matrix_stride, p, g1, 1, eps, eps_per_chain, p1_work, v_work, nullptr, nullptr};
    pa.kick_a = kick_coef;
    if (int rc = launch_pc(s, EPI_STORE, pa)) return rc;
  }
  hipLaunchKernelGGL(k_hmc_finish_dense, dim3(bjx_row_grid(N, kWavesPerBlock)), dim3(kBlock), 0, s,
                     Key{key0, key1}, chain_offset, step_fold, N, D, divergence_threshold, q0, logp0,
                     g0, ke0, q1, logp1, g1, p1_work, v_work, p_end_out, q_out, logp_out, g_out,
                     acceptance_rate_out, is_accepted_out, is_divergent_out, energy_out);
  return bjx_check_launch("bjx_hmc_finish_dense_coef");
}

static int mhmc_step_dense(const char* who, void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                           int64_t step_fold, int64_t N, int64_t D, int64_t step, float eps,
                           const float* eps_per_chain, const float* imm, int64_t matrix_stride,
                           float divergence_threshold, const float* logp0, const float* ke0, const float* q,
                           const float* p, const float* g, const float* logp_new, float* p1_work,
                           float* v_work, float* weight, float* sum_log_p_accept, uint8_t* any_divergent,
                           uint8_t* ever_accepted, float* prop_q, float* prop_p, float* prop_g,
                           float* prop_logp, float* prop_energy, const int32_t* n_steps,
                           float kick_coef = 0.5f) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  if (!(N >= 0 && D > 0 && step >= 0 && step < ((int64_t)1 << 31) && imm && logp0 && ke0 && q && p && g &&
        logp_new && p1_work && v_work && weight && sum_log_p_accept && any_divergent && ever_accepted && prop_q &&
        prop_p && prop_g && prop_logp && prop_energy && p1_work != p &&
        (matrix_stride < 0 || matrix_stride == 0 || matrix_stride == D * D))) {
    bjx_set_error("%s: bad arguments", who);
    return 1;
  }
  hipStream_t s = (hipStream_t)stream;
  // closing kick p1 = p + (eps kick_coef) g (kick_coef = 1/2 for velocity Verlet) ; v1 = imm p1.  With n_steps,
  // rows whose trajectory is complete keep their momentum (the prologue copies it through) and are skipped by
  // the reservoir step below.
  if (matrix_stride < 0) {
    GemmArgs ga{N, D, p, g, 1, eps, eps_per_chain, p1_work, imm, v_work, nullptr, nullptr};
    ga.kick_a = kick_coef;
    ga.b_symmetric = true;
    ga.n_steps = n_steps;
    ga.step_idx = (int32_t)step;
    if (int rc = launch_gemm(s, EPI_STORE, ga)) return rc;
  } else {
    PcArgs pa{N, D, imm, matrix_stride, p, g, 1, eps, eps_per_chain, p1_work, v_work, nullptr, nullptr};
    pa.kick_a = kick_coef;
    pa.n_steps = n_steps;
    pa.step_idx = (int32_t)step;
    if (int rc = launch_pc(s, EPI_STORE, pa)) return rc;
  }
  hipLaunchKernelGGL(k_mhmc_step_dense, dim3(bjx_row_grid(N, kWavesPerBlock)), dim3(kBlock), 0, s,
                     Key{key0, key1}, chain_offset, step_fold, N, D, step, divergence_threshold, logp0,
                     ke0, q, p1_work, v_work, g, logp_new, weight, sum_log_p_accept, any_divergent,
                     ever_accepted, prop_q, prop_p, prop_g, prop_logp, prop_energy, n_steps);
  return bjx_check_launch(who);
}

int bjx_mhmc_step_dense(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                        int64_t step_fold, int64_t N, int64_t D, int64_t step, float eps,
                        const float* eps_per_chain, const float* imm, int64_t matrix_stride,
                        float divergence_threshold, const float* logp0, const float* ke0, const float* q,
                        const float* p, const float* g, const float* logp_new, float* p1_work,
                        float* v_work, float* weight, float* sum_log_p_accept, uint8_t* any_divergent,
                        uint8_t* ever_accepted, float* prop_q, float* prop_p, float* prop_g,
                        float* prop_logp, float* prop_energy) {
  return mhmc_step_dense("bjx_mhmc_step_dense", stream, key0, key1, chain_offset, step_fold, N, D, step, eps,
                         eps_per_chain, imm, matrix_stride, divergence_threshold, logp0, ke0, q, p, g, logp_new,
                         p1_work, v_work, weight, sum_log_p_accept, any_divergent, ever_accepted, prop_q, prop_p,
                         prop_g, prop_logp, prop_energy, nullptr);
}

int bjx_mhmc_step_dense_masked(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                               int64_t step_fold, int64_t N, int64_t D, int64_t step, float eps,
                               const float* eps_per_chain, const float* imm, int64_t matrix_stride,
                               float divergence_threshold, const float* logp0, const float* ke0, const float* q,
                               const float* p, const float* g, const float* logp_new, float* p1_work,
                               float* v_work, float* weight, float* sum_log_p_accept, uint8_t* any_divergent,
                               uint8_t* ever_accepted, float* prop_q, float* prop_p, float* prop_g,
                               float* prop_logp, float* prop_energy, const int32_t* n_steps) {
  BJX_CHECK_ARG(N == 0 || n_steps, "bjx_mhmc_step_dense_masked: n_steps is NULL");
  return mhmc_step_dense("bjx_mhmc_step_dense_masked", stream, key0, key1, chain_offset, step_fold, N, D, step,
                         eps, eps_per_chain, imm, matrix_stride, divergence_threshold, logp0, ke0, q, p, g,
                         logp_new, p1_work, v_work, weight, sum_log_p_accept, any_divergent, ever_accepted, prop_q,
                         prop_p, prop_g, prop_logp, prop_energy, n_steps);
}

int bjx_mhmc_step_dense_coef(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                             int64_t step_fold, int64_t N, int64_t D, int64_t step, float kick_coef, float eps,
                             const float* eps_per_chain, const float* imm, int64_t matrix_stride,
                             float divergence_threshold, const float* logp0, const float* ke0, const float* q,
                             const float* p, const float* g, const float* logp_new, float* p1_work,
                             float* v_work, float* weight, float* sum_log_p_accept, uint8_t* any_divergent,
                             uint8_t* ever_accepted, float* prop_q, float* prop_p, float* prop_g,
                             float* prop_logp, float* prop_energy, const int32_t* n_steps) {
  return mhmc_step_dense("bjx_mhmc_step_dense_coef", stream, key0, key1, chain_offset, step_fold, N, D, step,
                         eps, eps_per_chain, imm, matrix_stride, divergence_threshold, logp0, ke0, q, p, g,
                         logp_new, p1_work, v_work, weight, sum_log_p_accept, any_divergent, ever_accepted, prop_q,
                         prop_p, prop_g, prop_logp, prop_energy, n_steps, kick_coef);
}

#ifdef BJX_DENSE_PROBE
/* PROBE builds only: device buffer of 4 x (number of workgroups) 64-bit stamps, or NULL. */
int bjx_dense_probe_set(void* buf) {
  g_probe_buf = (unsigned long long*)buf;
  return 0;
}
#endif

}  // extern "C"
