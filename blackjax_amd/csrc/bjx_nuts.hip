// NUTS kernels (gfx950): iterative tree doubling for N chains -- lockstep launches and free-running
// (asynchronous) chains share the per-chain device functions below.  C ABI and the algorithm map are
// in include/bjx_nuts.h.
//
// One wavefront per chain row.  Per-chain control state lives in the fs / is slot tables so that
// every decision (direction, progressive sampling, divergence, U-turn) is wave-uniform.
// Row sweeps move 16 bytes per lane (VEC = 4) whenever D % 4 == 0 and the buffers are 16-byte
// aligned; the dense-metric paths keep the 4-byte mapping (VEC = 1) their shuffle-based
// matrix-vector product needs.  All sweeps of one instantiation use the same lane <-> element
// mapping, so a lane only ever re-reads elements it wrote itself within a kernel.
// Numerics contract as in bjx_device.h: explicit fmaf, fp64-accumulated reductions, fp64 scalar
// transcendentals rounded once.
#ifndef __HIPCC_RTC__
#include "../../include/bjx_hip.h"
#include "bjx_device.h"
#include "bjx_host.h"
#else  // compiled at run time around a user-written target (blackjax_amd/rtc.py): device code only
#include "bjx_device.h"
#include "../../include/bjx_nuts.h"
#endif
#include "bjx_targets_dev.h"

using namespace bjx;

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / BJX_WAVE;

// The wave's index is the same in all 64 lanes: readfirstlane tells the compiler so, which puts
// everything derived from it (chain index, slot-table addresses, the threefry key arithmetic of the
// chain's RNG stream) on the scalar unit instead of repeating it in 64 vector lanes.
__device__ __forceinline__ int64_t wave_row0() {
  return (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}
__device__ __forceinline__ int64_t wave_row_stride() { return (int64_t)gridDim.x * kWavesPerBlock; }

#define FS(slot, c) nt.fs[(int64_t)(slot)*nt.N + (c)]
#define IS(slot, c) nt.is[(int64_t)(slot)*nt.N + (c)]
// this lane's pieces of a row: VEC consecutive elements starting at j0
#define BJX_ROW_SWEEP(j0) for (int64_t j0 = (int64_t)(threadIdx.x & 63) * VEC; j0 < nt.D; j0 += 64 * VEC)

template <int VEC>
struct Row {
  float v[VEC];
};
template <int VEC>
__device__ __forceinline__ Row<VEC> ldr(const float* p) {
  Row<VEC> r;
  if constexpr (VEC == 4) {
    const F4 t = ld4(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
    r.v[0] = p[0];
  }
  return r;
}
template <int VEC>
__device__ __forceinline__ void str(float* p, const Row<VEC>& r) {
  if constexpr (VEC == 4) st4(p, F4{r.v[0], r.v[1], r.v[2], r.v[3]});
  else p[0] = r.v[0];
}

// np.logaddexp / jnp.logaddexp in fp64, rounded once
__device__ __forceinline__ float logaddexp_cr(float a, float b) {
  const double x = (double)a, y = (double)b;
  double r;
  if (x == y) {
    r = x + 0.6931471805599453;
  } else {
    const double t = x - y;
    if (t > 0) r = x + log1p(exp(-t));
    else if (t <= 0) r = y + log1p(exp(t));
    else r = t;  // NaN
  }
  return (float)r;
}

// jax.scipy.special.expit in fp64, rounded once
__device__ __forceinline__ float expit_cr(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }

// The scalar transcendentals of one tree step, evaluated in ONE pass of the fp64 routines instead of
// three: the values are wave-uniform, so three lanes get three different operands --
//   lane 0: e0 = exp(arg0), r0 = 1 / (1 + e0)          lanes 1, 2: logaddexp(a1, b1), logaddexp(a2, b2)
// (logaddexp = max + log1p(exp(-|a - b|)), the same expression tree as logaddexp_cr) -- and the
// results are broadcast with readlane.  Bit-identical to the one-at-a-time helpers above.
struct Scalars3 {
  float e0, r0, lae1, lae2;
};
__device__ __forceinline__ float bcast_lane(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ Scalars3 scalars3(double arg0, float a1, float b1, float a2, float b2) {
  const int lane = threadIdx.x & 63;
  const double xa = lane == 1 ? (double)a1 : (double)a2;
  const double xb = lane == 1 ? (double)b1 : (double)b2;
  const double t = xa - xb;
  const double e = exp(lane == 0 ? arg0 : -fabs(t));
  const double l1p = log1p(e);
  double r;
  if (lane == 0) r = 1.0 / (1.0 + e);
  else if (xa == xb) r = xa + 0.6931471805599453;
  else if (t > 0) r = xa + l1p;
  else if (t <= 0) r = xb + l1p;
  else r = t;  // NaN
  const float rf = (float)r, ef = (float)e;
  return Scalars3{bcast_lane(ef, 0), bcast_lane(rf, 0), bcast_lane(rf, 1), bcast_lane(rf, 2)};
}

// jnp.minimum(x, 1): NaN propagates
__device__ __forceinline__ float min1_nan(float x) { return (x < 1.0f || x != x) ? x : 1.0f; }

// Launch-time parameters either come from the kernel arguments (eager launches) or, for HIP-graph
// replays, from a device control block ctl = {depth, s_base, n_rows, key0, key1, step_fold,
// chain_offset} so that one captured graph serves every chunk of every transition.
struct StepCtx {
  int32_t depth, s;
  int64_t n_rows;
  Key key;
  int64_t off, fold;
};

__device__ __forceinline__ StepCtx make_ctx(const bjx_nuts_t& nt, int32_t depth, int32_t s,
                                            int64_t n_rows, const int64_t* __restrict__ ctl) {
  if (ctl) {
    StepCtx c;
    c.depth = (int32_t)ctl[0];
    c.s = (int32_t)ctl[1] + s;  // s is the offset inside the captured chunk
    c.n_rows = ctl[2] < n_rows ? ctl[2] : n_rows;
    c.key = Key{(uint32_t)ctl[3], (uint32_t)ctl[4]};
    c.fold = ctl[5];
    c.off = ctl[6];
    return c;
  }
  return StepCtx{depth, s, n_rows, Key{nt.key0, nt.key1}, nt.chain_offset, nt.step_fold};
}

__device__ __forceinline__ Key integrator_key(const StepCtx& cx, int64_t c) {
  const Key kc = chain_key(cx.key, (uint64_t)(c + cx.off), cx.fold);
  return key_child(kc, 1);  // split(kc, 2)[1]   (nuts.py:133)
}

// coefficients b_1 / a_1 of the palindromic integrator (both 0 in the descriptor = velocity Verlet)
__device__ __forceinline__ float int_kick(const bjx_nuts_t& nt) { return nt.int_kick != 0.0f ? nt.int_kick : 0.5f; }
__device__ __forceinline__ float int_drift(const bjx_nuts_t& nt) { return nt.int_drift != 0.0f ? nt.int_drift : 1.0f; }

__device__ __forceinline__ float chain_eps(const bjx_nuts_t& nt, int64_t c) {
  return nt.eps_per_chain ? nt.eps_per_chain[c] : nt.eps;
}

// Dense metric: y_i = sum_j M[j][i] x_j for THIS lane's output index i (M symmetric, so this is
// (M x)_i with coalesced row reads), fp64 accumulate.  x is supplied lane-wise by xf(j) and
// broadcast with wave shuffles, so the vector never round-trips memory.  Must be called by all
// 64 lanes (uniform loops); lanes with i >= D just take part in the shuffles.
template <class XF>
__device__ __forceinline__ double matvec_t_lane(const float* __restrict__ M, int64_t D, int64_t i,
                                                XF xf) {
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (int64_t jc = 0; jc < D; jc += 64) {
    const int64_t jl = jc + lane;
    const float xr = jl < D ? xf(jl) : 0.0f;
    const int lim = (int)((D - jc) < 64 ? (D - jc) : 64);
    for (int t = 0; t < lim; ++t) {
      const float xj = __shfl(xr, t, BJX_WAVE);
      if (i < D) acc += (double)M[(jc + t) * D + i] * (double)xj;
    }
  }
  return acc;
}

// Opening half of a leapfrog on the trajectory end `dir` of chain c (integrators.py:104-150 with
// step dir*eps): p += h g ; q += deps * (M^{-1} p) ; new position also to the compact row qo.
// gsrc = gradient at the current end state.
template <int VEC, bool DENSE>
__device__ __forceinline__ void nuts_open_half(const bjx_nuts_t& nt, int64_t c, int dir, float deps,
                                               float h, const float* gsrc, float* qo,
                                               const float* vpre = nullptr) {
  const int64_t base = c * nt.D;
  float* fq = (dir > 0 ? nt.Rq : nt.Lq) + base;
  float* fp = (dir > 0 ? nt.Rp : nt.Lp) + base;
  if constexpr (DENSE) {
    // VEC == 4 instantiations exist for the v_pre path only (the launcher picks them when v_pre is set and
    // the buffers are 16-byte aligned): the mat-vec below uses the 4-byte lane mapping, and all sweeps of
    // one kernel must share one mapping
    const int lane = threadIdx.x & 63;
    if (vpre) {  // velocity of the kicked momentum from the caller's GEMM (bjx_nuts_t.v_pre)
      BJX_ROW_SWEEP(j0) {
        const Row<VEC> g = ldr<VEC>(gsrc + j0), v = ldr<VEC>(vpre + j0);
        Row<VEC> p = ldr<VEC>(fp + j0), q = ldr<VEC>(fq + j0);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          q.v[e] = fmaf(deps, v.v[e], q.v[e]);
          p.v[e] = fmaf(h, g.v[e], p.v[e]);
        }
        str<VEC>(fq + j0, q);
        str<VEC>(qo + j0, q);
        str<VEC>(fp + j0, p);
      }
      return;
    }
    const float* M = nt.Mdense + c * nt.Mdense_stride;
    for (int64_t ic = 0; ic < nt.D; ic += 64) {
      const int64_t i = ic + lane;
      const double acc = matvec_t_lane(M, nt.D, i, [&](int64_t j) { return fmaf(h, gsrc[j], fp[j]); });
      if (i < nt.D) {
        const float qn = fmaf(deps, (float)acc, fq[i]);
        fq[i] = qn;
        qo[i] = qn;
      }
    }
    for (int64_t j = lane; j < nt.D; j += 64) fp[j] = fmaf(h, gsrc[j], fp[j]);
  } else {
    const float* im = nt.imm + c * nt.imm_stride;
    BJX_ROW_SWEEP(j0) {
      const Row<VEC> g = ldr<VEC>(gsrc + j0), m = ldr<VEC>(im + j0);
      Row<VEC> p = ldr<VEC>(fp + j0), q = ldr<VEC>(fq + j0);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        p.v[e] = fmaf(h, g.v[e], p.v[e]);
        q.v[e] = fmaf(deps, m.v[e] * p.v[e], q.v[e]);
      }
      str<VEC>(fp + j0, p);
      str<VEC>(fq + j0, q);
      str<VEC>(qo + j0, q);
    }
  }
}

// ------------------------------------------------------------------------------------ init
// Tree state of chain c at the start of a transition (nuts.py:278-291): both ends and the proposal
// are the current state, momentum_sum = p0, num_states = 0.  lp = logdensity, ke = K(p0).
template <int VEC, bool DENSE>
__device__ __forceinline__ void nuts_init_chain(const bjx_nuts_t& nt, int64_t c, float lp, float ke) {
  const int lane = threadIdx.x & 63;
  const int64_t base = c * nt.D;
  BJX_ROW_SWEEP(j0) {
    const Row<VEC> q = ldr<VEC>(nt.q0 + base + j0), p = ldr<VEC>(nt.p0 + base + j0),
                   g = ldr<VEC>(nt.g0 + base + j0);
    str<VEC>(nt.Lq + base + j0, q); str<VEC>(nt.Rq + base + j0, q); str<VEC>(nt.Pq + base + j0, q);
    str<VEC>(nt.Lp + base + j0, p); str<VEC>(nt.Rp + base + j0, p); str<VEC>(nt.msum + base + j0, p);
    str<VEC>(nt.Lg + base + j0, g); str<VEC>(nt.Rg + base + j0, g); str<VEC>(nt.Pg + base + j0, g);
    if constexpr (DENSE) {
      const Row<VEC> v = ldr<VEC>(nt.v0 + base + j0);
      str<VEC>(nt.Lv + base + j0, v); str<VEC>(nt.Rv + base + j0, v);
    }
  }
  if (lane == 0) {
    const float H0 = -lp + ke;
    FS(BJX_NUTS_F_H0, c) = H0;
    FS(BJX_NUTS_F_LLOGP, c) = lp;
    FS(BJX_NUTS_F_RLOGP, c) = lp;
    FS(BJX_NUTS_F_PLOGP, c) = lp;
    FS(BJX_NUTS_F_PENERGY, c) = H0;
    FS(BJX_NUTS_F_PW, c) = 0.0f;
    FS(BJX_NUTS_F_PSLPA, c) = -__builtin_inff();
    FS(BJX_NUTS_F_SLOGP, c) = lp;
    FS(BJX_NUTS_F_SENERGY, c) = H0;
    FS(BJX_NUTS_F_SW, c) = 0.0f;
    FS(BJX_NUTS_F_SSLPA, c) = -__builtin_inff();
    FS(BJX_NUTS_F_ACC, c) = __builtin_nanf("");
    IS(BJX_NUTS_I_ACTIVE, c) = nt.max_depth > 0 ? 1 : 0;
    IS(BJX_NUTS_I_SUB_ACTIVE, c) = 0;
    IS(BJX_NUTS_I_DIR, c) = 1;
    IS(BJX_NUTS_I_NSTATES, c) = 0;
    IS(BJX_NUTS_I_SUBN, c) = 0;
    IS(BJX_NUTS_I_SDIV, c) = 0;
    IS(BJX_NUTS_I_STURN, c) = 0;
    IS(BJX_NUTS_I_DIV, c) = 0;
    IS(BJX_NUTS_I_TURN, c) = 0;
    IS(BJX_NUTS_I_DEPTH, c) = 0;
  }
}

template <int VEC, bool DENSE>
__global__ void __launch_bounds__(kBlock)
k_nuts_init(bjx_nuts_t nt, const float* __restrict__ logp0, const float* __restrict__ ke0) {
  for (int64_t c = wave_row0(); c < nt.N; c += wave_row_stride())
    nuts_init_chain<VEC, DENSE>(nt, c, logp0[c], ke0[c]);
}

// Start of doubling `depth` for chain c: draw the direction and reset the subtree flags
// (trajectory.py:645-650).  Returns the direction (+1 / -1).
__device__ __forceinline__ int nuts_begin_doubling(const bjx_nuts_t& nt, const StepCtx& cx, int64_t c,
                                                   int32_t depth) {
  const int lane = threadIdx.x & 63;
  // The transition's integrator key (nuts.py:133) is derived once, at doubling 0, and kept in the
  // slot table; so are the doubling's leaf-sampling and proposal keys below.  A leaf or a merge then
  // runs the threefry blocks that depend on its own index only.
  Key ik;
  if (depth == 0) {
    ik = integrator_key(cx, c);
    if (lane == 0) {
      IS(BJX_NUTS_I_IK, c) = (int32_t)ik.k0;
      IS(BJX_NUTS_I_IKB, c) = (int32_t)ik.k1;
    }
  } else {
    ik = Key{(uint32_t)__builtin_amdgcn_readfirstlane(IS(BJX_NUTS_I_IK, c)),
             (uint32_t)__builtin_amdgcn_readfirstlane(IS(BJX_NUTS_I_IKB, c))};
  }
  const Key subkey = key_child(ik, (uint64_t)depth);                      // trajectory.py:645
  const Key kd = key_child(subkey, 0);                                    // split(subkey,3)[0]
  const int dir = key_uniform(kd) < 0.5f ? 1 : -1;                        // trajectory.py:650
  // split(subkey,3)[1]: every leaf of this doubling folds its index into this key (trajectory.py:
  // 329-339); kept in the slot table so a leaf derives ONE key instead of the whole chain of five
  const Key kt = key_child(subkey, 1);
  const Key kp = key_child(subkey, 2);  // split(subkey,3)[2]: progressive_biased_sampling at the merge
  if (lane == 0) {
    IS(BJX_NUTS_I_KT, c) = (int32_t)kt.k0;
    IS(BJX_NUTS_I_KTB, c) = (int32_t)kt.k1;
    IS(BJX_NUTS_I_KP, c) = (int32_t)kp.k0;
    IS(BJX_NUTS_I_KPB, c) = (int32_t)kp.k1;
    IS(BJX_NUTS_I_DIR, c) = dir;
    IS(BJX_NUTS_I_SUB_ACTIVE, c) = 1;
    IS(BJX_NUTS_I_SDIV, c) = 0;
    IS(BJX_NUTS_I_STURN, c) = 0;
    IS(BJX_NUTS_I_SUBN, c) = 0;
  }
  return dir;
}

// ------------------------------------------------------------------------------------ pre
template <int VEC, bool DENSE>
__global__ void __launch_bounds__(kBlock)
k_nuts_pre(bjx_nuts_t nt, int32_t depth_arg, int32_t s_arg, int64_t n_rows_arg,
           const int32_t* __restrict__ idx, const int64_t* __restrict__ ctl, float* __restrict__ qf) {
  const StepCtx cx = make_ctx(nt, depth_arg, s_arg, n_rows_arg, ctl);
  const int32_t depth = cx.depth, s = cx.s;
  const int64_t n_rows = cx.n_rows;
  for (int64_t b = wave_row0(); b < n_rows; b += wave_row_stride()) {
    const int64_t c = idx ? (int64_t)idx[b] : b;
    if (!IS(BJX_NUTS_I_ACTIVE, c)) continue;
    int dir;
    if (s == 0) {
      dir = nuts_begin_doubling(nt, cx, c, depth);
    } else {
      if (!IS(BJX_NUTS_I_SUB_ACTIVE, c)) continue;
      dir = IS(BJX_NUTS_I_DIR, c);
    }
    const float deps = (float)dir * chain_eps(nt, c);  // direction * step_size (trajectory.py:323)
    const float h = deps * int_kick(nt);               // step_size * coef (integrators.py:236)
    const float* fg = (dir > 0 ? nt.Rg : nt.Lg) + c * nt.D;
    nuts_open_half<VEC, DENSE>(nt, c, dir, deps * int_drift(nt), h, fg, qf + b * nt.D,
                               nt.v_pre ? nt.v_pre + b * nt.D : nullptr);
  }
}

// Stage 2 .. K of a multi-stage palindromic integrator (integrators.py:128-146) on the integrating end
// of every chain whose subtree is still running: the arithmetic of nuts_open_half with the stage's
// coefficients and the callable's latest gradient.
template <int VEC, bool DENSE>
__global__ void __launch_bounds__(kBlock)
k_nuts_mid(bjx_nuts_t nt, int64_t n_rows_arg, const int32_t* __restrict__ idx,
           const int64_t* __restrict__ ctl, float* __restrict__ qf, const float* __restrict__ gf, float kick,
           float drift) {
  const int64_t n_rows = ctl ? (ctl[2] < n_rows_arg ? ctl[2] : n_rows_arg) : n_rows_arg;
  for (int64_t b = wave_row0(); b < n_rows; b += wave_row_stride()) {
    const int64_t c = idx ? (int64_t)idx[b] : b;
    if (!IS(BJX_NUTS_I_ACTIVE, c) || !IS(BJX_NUTS_I_SUB_ACTIVE, c)) continue;
    const int dir = IS(BJX_NUTS_I_DIR, c);
    const float deps = (float)dir * chain_eps(nt, c);
    nuts_open_half<VEC, DENSE>(nt, c, dir, deps * drift, deps * kick, gf + b * nt.D, qf + b * nt.D,
                               nt.v_pre ? nt.v_pre + b * nt.D : nullptr);
  }
}

// Compact rows of kicked momenta for the GEMM that applies a shared dense inverse mass matrix
// (bjx_nuts_t.v_pre): pc[b] = p_end + (dir eps kick) g, g = gf[b] or the end's stored gradient.
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_nuts_dense_kick(bjx_nuts_t nt, int32_t depth_arg, int32_t s_arg, int64_t n_rows_arg,
                  const int32_t* __restrict__ idx, const int64_t* __restrict__ ctl,
                  const float* __restrict__ gf, float kick, float* __restrict__ pc) {
  const StepCtx cx = make_ctx(nt, depth_arg, s_arg, n_rows_arg, ctl);
  for (int64_t b = wave_row0(); b < cx.n_rows; b += wave_row_stride()) {
    const int64_t c = idx ? (int64_t)idx[b] : b;
    float* out = pc + b * nt.D;
    bool act = IS(BJX_NUTS_I_ACTIVE, c) != 0;
    int dir = 1;
    if (act) {
      if (cx.s == 0 && !gf) {
        dir = nuts_begin_doubling(nt, cx, c, cx.depth);
      } else {
        act = IS(BJX_NUTS_I_SUB_ACTIVE, c) != 0;
        dir = IS(BJX_NUTS_I_DIR, c);
      }
    }
    if (!act) {
      Row<VEC> z;
#pragma unroll
      for (int e = 0; e < VEC; ++e) z.v[e] = 0.0f;
      BJX_ROW_SWEEP(j0) str<VEC>(out + j0, z);
      continue;
    }
    const float h = ((float)dir * chain_eps(nt, c)) * kick;
    const int64_t base = c * nt.D;
    const float* p = (dir > 0 ? nt.Rp : nt.Lp) + base;
    const float* g = gf ? gf + b * nt.D : (dir > 0 ? nt.Rg : nt.Lg) + base;
    BJX_ROW_SWEEP(j0) {
      const Row<VEC> gg = ldr<VEC>(g + j0);
      Row<VEC> pp = ldr<VEC>(p + j0);
#pragma unroll
      for (int e = 0; e < VEC; ++e) pp.v[e] = fmaf(h, gg.v[e], pp.v[e]);
      str<VEC>(out + j0, pp);
    }
  }
}

// ------------------------------------------------------------------------------------ post
// Register-resident leaf (diagonal metric, D <= 64 * VEC * NI): the same arithmetic as the general
// nuts_post_chain below, but every row a leaf needs -- new gradient, metric, end momentum and
// position, subtree momentum sum, new position -- is requested ONCE, up front, and the three passes
// plus the fused opening half of the next leaf run out of registers.  A leaf is then a chain of
// ~4 dependent memory round trips (scalars; rows; checkpoint rows per U-turn level; nothing) instead
// of ~8, and moves ~12 rows instead of ~23: the tick kernels are latency-bound on exactly that chain.
template <int VEC, int NI>
__device__ __forceinline__ bool nuts_post_chain_resident(const bjx_nuts_t& nt, const StepCtx& cx,
                                                         int64_t c, int64_t b, int32_t depth, int32_t s,
                                                         float* qf, const float* __restrict__ logp_f,
                                                         const float* __restrict__ gf, bool fuse_next) {
  const int lane = threadIdx.x & 63;
  const int dir = IS(BJX_NUTS_I_DIR, c);
  const float deps = (float)dir * chain_eps(nt, c);
  const float h = deps * int_kick(nt);    // closing kick b_K = b_1 (and the next leaf's opening kick)
  const float dd = deps * int_drift(nt);  // first drift a_1 of the next leaf
  const int64_t base = c * nt.D;
  float* fq = (dir > 0 ? nt.Rq : nt.Lq) + base;
  float* fp = (dir > 0 ? nt.Rp : nt.Lp) + base;
  float* fg = (dir > 0 ? nt.Rg : nt.Lg) + base;
  const float* im = nt.imm + c * nt.imm_stride;
  const float* gn = gf + b * nt.D;
  float* qn = qf + b * nt.D;
  float* sm = nt.Smsum + base;
  // scalars the decisions need, requested together with the rows
  const float lp = logp_f[b];
  const float H0 = FS(BJX_NUTS_F_H0, c);
  const float sw = FS(BJX_NUTS_F_SW, c);
  const float sslpa = FS(BJX_NUTS_F_SSLPA, c);

  int64_t j0[NI];
  bool ok[NI];
  Row<VEC> G[NI], M[NI], P[NI], S[NI], Q[NI], X[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    j0[k] = ((int64_t)lane + 64 * k) * VEC;
    ok[k] = j0[k] < nt.D;
    if (ok[k]) {
      G[k] = ldr<VEC>(gn + j0[k]);
      M[k] = ldr<VEC>(im + j0[k]);
      P[k] = ldr<VEC>(fp + j0[k]);
      Q[k] = ldr<VEC>(fq + j0[k]);
      X[k] = ldr<VEC>(qn + j0[k]);
      if (s != 0) S[k] = ldr<VEC>(sm + j0[k]);
    }
  }
  // pass 1: closing half kick, kinetic energy
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < NI; ++k)
    if (ok[k]) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        P[k].v[e] = fmaf(h, G[k].v[e], P[k].v[e]);
        acc += (double)(M[k].v[e] * P[k].v[e]) * (double)P[k].v[e];
      }
    }
  acc = wave_sum(acc);
  const float ke = 0.5f * (float)acc;
  const float e_new = -lp + ke;  // hmc_energy (trajectory.py:745-748)
  float w = H0 - e_new;          // proposal.py:91-95
  if (w != w) w = -__builtin_inff();
  const float slpa_new = fminf(w, 0.0f);
  const bool sdiv = (-w) > nt.divergence_threshold;  // trajectory.py:325
  bool take;
  float Wn, Sn;
  if (s == 0) {
    take = true;
    Wn = w;
    Sn = slpa_new;
  } else {  // progressive uniform sampling (trajectory.py:329-339, proposal.py:118-143)
    const Key kt{(uint32_t)__builtin_amdgcn_readfirstlane(IS(BJX_NUTS_I_KT, c)),
                 (uint32_t)__builtin_amdgcn_readfirstlane(IS(BJX_NUTS_I_KTB, c))};  // nuts_begin_doubling
    const float u = key_uniform(key_child(kt, (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(s)));  // fold_in(kt, s)
    const Scalars3 sc = scalars3(-(double)(w - sw), sw, w, sslpa, slpa_new);
    take = u < sc.r0;
    Wn = sc.lae1;
    Sn = sc.lae2;
  }
  const uint32_t us = (uint32_t)s;  // checkpoint indices (termination.py:75-84)
  const int idx_max = __popc(us >> 1);
  const int nsub = __popc((~us & (us + 1u)) - 1u);
  const int idx_min = idx_max - nsub + 1;
  const bool even = (us & 1u) == 0u;

  // pass 2: momentum-sum append, checkpoint store, subtree-proposal state copy
  float* ckr = nt.ckpt_r + (c * nt.max_depth + idx_max) * nt.D;
  float* ckrs = nt.ckpt_rs + (c * nt.max_depth + idx_max) * nt.D;
#pragma unroll
  for (int k = 0; k < NI; ++k)
    if (ok[k]) {
      if (s != 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) S[k].v[e] = S[k].v[e] + P[k].v[e];
      } else {
        S[k] = P[k];
      }
      str<VEC>(sm + j0[k], S[k]);
      str<VEC>(fg + j0[k], G[k]);
      if (even) {
        str<VEC>(ckr + j0[k], P[k]);
        str<VEC>(ckrs + j0[k], S[k]);
      }
      if (take) {
        str<VEC>(nt.Sq + base + j0[k], X[k]);
        str<VEC>(nt.Sg + base + j0[k], G[k]);
      }
    }

  // pass 3: iterative U-turn over the checkpoints idx_max .. idx_min (odd leaves only, so none of
  // them was written by this leaf)
  bool turning = false;
  for (int i = idx_max; i >= idx_min && !turning; --i) {
    const float* r_ck = nt.ckpt_r + (c * nt.max_depth + i) * nt.D;
    const float* rs_ck = nt.ckpt_rs + (c * nt.max_depth + i) * nt.D;
    double a_left = 0.0, a_right = 0.0;
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (ok[k]) {
        const Row<VEC> rl = ldr<VEC>(r_ck + j0[k]), rs = ldr<VEC>(rs_ck + j0[k]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float ssum = (S[k].v[e] - rs.v[e]) + rl.v[e];
          const float rho = ssum - (P[k].v[e] + rl.v[e]) * 0.5f;  // metrics.py:300
          a_left += (double)(M[k].v[e] * rl.v[e]) * (double)rho;
          a_right += (double)(M[k].v[e] * P[k].v[e]) * (double)rho;
        }
      }
    a_left = wave_sum(a_left);
    a_right = wave_sum(a_right);
    turning = ((float)a_left <= 0.0f) || ((float)a_right <= 0.0f);
  }
  const bool stop = sdiv || turning;
  if (lane == 0) {
    FS(dir > 0 ? BJX_NUTS_F_RLOGP : BJX_NUTS_F_LLOGP, c) = lp;
    FS(BJX_NUTS_F_SW, c) = Wn;
    FS(BJX_NUTS_F_SSLPA, c) = Sn;
    if (take) {
      FS(BJX_NUTS_F_SLOGP, c) = lp;
      FS(BJX_NUTS_F_SENERGY, c) = e_new;
    }
    IS(BJX_NUTS_I_SUBN, c) = s + 1;
    IS(BJX_NUTS_I_SDIV, c) = sdiv ? 1 : 0;
    IS(BJX_NUTS_I_STURN, c) = turning ? 1 : 0;
    if (stop) IS(BJX_NUTS_I_SUB_ACTIVE, c) = 0;
  }
  // end momentum: as kicked by this leaf, or already carrying the opening half of the next one
  const bool open_next = fuse_next && !stop;
#pragma unroll
  for (int k = 0; k < NI; ++k)
    if (ok[k]) {
      if (open_next) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          P[k].v[e] = fmaf(h, G[k].v[e], P[k].v[e]);
          Q[k].v[e] = fmaf(dd, M[k].v[e] * P[k].v[e], Q[k].v[e]);
        }
        str<VEC>(fq + j0[k], Q[k]);
        str<VEC>(qn + j0[k], Q[k]);
      }
      str<VEC>(fp + j0[k], P[k]);
    }
  return stop;
}

// Second half of leaf s of doubling `depth` for chain c, whose new position / log-density /
// gradient sit in row b of (qf, logp_f, gf): closing kick, energy, progressive sampling,
// momentum-sum append, checkpoint store, iterative U-turn (trajectory.py:242-395,
// termination.py:31-106).  With fuse_next the opening half of leaf s+1 follows when the subtree
// keeps integrating.  Returns true when the subtree stops (divergence or U-turn).
template <int VEC, bool DENSE>
__device__ __forceinline__ bool nuts_post_chain(const bjx_nuts_t& nt, const StepCtx& cx, int64_t c,
                                                int64_t b, int32_t depth, int32_t s, float* qf,
                                                const float* __restrict__ logp_f,
                                                const float* __restrict__ gf, bool fuse_next) {
  const int lane = threadIdx.x & 63;
  const int dir = IS(BJX_NUTS_I_DIR, c);
  const float deps = (float)dir * chain_eps(nt, c);
  const float h = deps * int_kick(nt);
  const int64_t base = c * nt.D;
  float* fp = (dir > 0 ? nt.Rp : nt.Lp) + base;
  float* fg = (dir > 0 ? nt.Rg : nt.Lg) + base;
  const float* im = nt.imm + c * nt.imm_stride;
  const float* gn = gf + b * nt.D;
  float* qn = qf + b * nt.D;

  // pass 1: closing half kick, store the new end state, kinetic energy
  double acc = 0.0;
  float* fv = nullptr;  // dense metric: velocity M^{-1} p of the new end state
  if constexpr (DENSE) {
    const float* M = nt.Mdense + c * nt.Mdense_stride;
    fv = (dir > 0 ? nt.Rv : nt.Lv) + base;
    if (nt.v_pre) {  // velocity of the closing-kicked momentum from the caller's GEMM
      const float* vp = nt.v_pre + b * nt.D;
      BJX_ROW_SWEEP(j0) {
        const Row<VEC> g = ldr<VEC>(gn + j0), v = ldr<VEC>(vp + j0);
        Row<VEC> p = ldr<VEC>(fp + j0);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          p.v[e] = fmaf(h, g.v[e], p.v[e]);
          acc += (double)v.v[e] * (double)p.v[e];
        }
        str<VEC>(fv + j0, v);
        str<VEC>(fp + j0, p);
        str<VEC>(fg + j0, g);
      }
    } else {
      for (int64_t ic = 0; ic < nt.D; ic += 64) {
        const int64_t i = ic + lane;
        const double av = matvec_t_lane(M, nt.D, i, [&](int64_t j) { return fmaf(h, gn[j], fp[j]); });
        if (i < nt.D) {
          const float p = fmaf(h, gn[i], fp[i]);
          const float v = (float)av;
          fv[i] = v;
          acc += (double)v * (double)p;
        }
      }
    }
    if (!nt.v_pre)
      for (int64_t j = lane; j < nt.D; j += 64) {
        fp[j] = fmaf(h, gn[j], fp[j]);
        fg[j] = gn[j];
      }
  } else {
    BJX_ROW_SWEEP(j0) {
      const Row<VEC> g = ldr<VEC>(gn + j0), m = ldr<VEC>(im + j0);
      Row<VEC> p = ldr<VEC>(fp + j0);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        p.v[e] = fmaf(h, g.v[e], p.v[e]);
        acc += (double)(m.v[e] * p.v[e]) * (double)p.v[e];
      }
      str<VEC>(fp + j0, p);
      str<VEC>(fg + j0, g);
    }
  }
  acc = wave_sum(acc);
  const float ke = 0.5f * (float)acc;
  const float lp = logp_f[b];
  const float e_new = -lp + ke;                       // hmc_energy (trajectory.py:745-748)
  float w = FS(BJX_NUTS_F_H0, c) - e_new;             // proposal.py:91-95
  if (w != w) w = -__builtin_inff();
  const float slpa_new = fminf(w, 0.0f);
  const bool sdiv = (-w) > nt.divergence_threshold;   // trajectory.py:325

  // progressive uniform sampling (trajectory.py:329-339, proposal.py:118-143)
  bool take;
  float Wn, Sn;
  if (s == 0) {
    take = true;
    Wn = w;
    Sn = slpa_new;
  } else {
    const float sw = FS(BJX_NUTS_F_SW, c);
    const Key kt{(uint32_t)__builtin_amdgcn_readfirstlane(IS(BJX_NUTS_I_KT, c)),
                 (uint32_t)__builtin_amdgcn_readfirstlane(IS(BJX_NUTS_I_KTB, c))};  // nuts_begin_doubling
    const float u = key_uniform(key_child(kt, (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(s)));  // fold_in(kt, s)
    // pa = expit(w - sw), Wn = logaddexp(sw, w), Sn = logaddexp(sum_log_p_accept, min(w, 0))
    const Scalars3 sc = scalars3(-(double)(w - sw), sw, w, FS(BJX_NUTS_F_SSLPA, c), slpa_new);
    take = u < sc.r0;
    Wn = sc.lae1;
    Sn = sc.lae2;
  }
  // checkpoint indices (termination.py:75-84)
  const uint32_t us = (uint32_t)s;
  const int idx_max = __popc(us >> 1);
  const int nsub = __popc((~us & (us + 1u)) - 1u);
  const int idx_min = idx_max - nsub + 1;
  const bool even = (us & 1u) == 0u;

  // pass 2: momentum-sum append, checkpoint store, subtree-proposal state copy
  float* sm = nt.Smsum + base;
  float* ckr = nt.ckpt_r + (c * nt.max_depth + idx_max) * nt.D;
  float* ckrs = nt.ckpt_rs + (c * nt.max_depth + idx_max) * nt.D;
  float* sq = nt.Sq + base;
  float* sg = nt.Sg + base;
  BJX_ROW_SWEEP(j0) {
    const Row<VEC> p = ldr<VEC>(fp + j0);
    Row<VEC> m = p;
    if (s != 0) {  // append_to_trajectory (trajectory.py:62-67)
      const Row<VEC> old = ldr<VEC>(sm + j0);
#pragma unroll
      for (int e = 0; e < VEC; ++e) m.v[e] = old.v[e] + p.v[e];
    }
    str<VEC>(sm + j0, m);
    if (even) {
      str<VEC>(ckr + j0, p);
      str<VEC>(ckrs + j0, m);
      if constexpr (DENSE) str<VEC>(nt.ckpt_v + (c * nt.max_depth + idx_max) * nt.D + j0, ldr<VEC>(fv + j0));
    }
    if (take) {
      str<VEC>(sq + j0, ldr<VEC>(qn + j0));
      str<VEC>(sg + j0, ldr<VEC>(gn + j0));
    }
  }

  // pass 3: iterative U-turn over the checkpoints idx_max .. idx_min (termination.py:86-104)
  bool turning = false;
  for (int i = idx_max; i >= idx_min && !turning; --i) {
    const float* r_ck = nt.ckpt_r + (c * nt.max_depth + i) * nt.D;
    const float* rs_ck = nt.ckpt_rs + (c * nt.max_depth + i) * nt.D;
    double a_left = 0.0, a_right = 0.0;
    BJX_ROW_SWEEP(j0) {
      const Row<VEC> p = ldr<VEC>(fp + j0), rl = ldr<VEC>(r_ck + j0), msum = ldr<VEC>(sm + j0),
                     rs = ldr<VEC>(rs_ck + j0);
      Row<VEC> vl, vr;
      if constexpr (DENSE) {
        vl = ldr<VEC>(nt.ckpt_v + (c * nt.max_depth + i) * nt.D + j0);
        vr = ldr<VEC>(fv + j0);
      } else {
        const Row<VEC> m = ldr<VEC>(im + j0);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          vl.v[e] = m.v[e] * rl.v[e];  // velocity_left / velocity_right
          vr.v[e] = m.v[e] * p.v[e];
        }
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float ssum = (msum.v[e] - rs.v[e]) + rl.v[e];
        const float rho = ssum - (p.v[e] + rl.v[e]) * 0.5f;  // metrics.py:300
        a_left += (double)vl.v[e] * (double)rho;
        a_right += (double)vr.v[e] * (double)rho;
      }
    }
    a_left = wave_sum(a_left);
    a_right = wave_sum(a_right);
    turning = ((float)a_left <= 0.0f) || ((float)a_right <= 0.0f);
  }

  if (lane == 0) {
    FS(dir > 0 ? BJX_NUTS_F_RLOGP : BJX_NUTS_F_LLOGP, c) = lp;
    FS(BJX_NUTS_F_SW, c) = Wn;
    FS(BJX_NUTS_F_SSLPA, c) = Sn;
    if (take) {
      FS(BJX_NUTS_F_SLOGP, c) = lp;
      FS(BJX_NUTS_F_SENERGY, c) = e_new;
    }
    IS(BJX_NUTS_I_SUBN, c) = s + 1;
    IS(BJX_NUTS_I_SDIV, c) = sdiv ? 1 : 0;
    IS(BJX_NUTS_I_STURN, c) = turning ? 1 : 0;
    if (sdiv || turning) IS(BJX_NUTS_I_SUB_ACTIVE, c) = 0;
  }

  // Fused opening half of the NEXT leapfrog (same arithmetic as k_nuts_pre at s + 1): saves a
  // launch and the re-read of p, g, q.  Only when the subtree keeps integrating.
  if (fuse_next && !(sdiv || turning)) nuts_open_half<VEC, DENSE>(nt, c, dir, deps * int_drift(nt), h, gn, qn);
  return sdiv || turning;
}

template <int VEC, bool DENSE>
__global__ void __launch_bounds__(kBlock)
k_nuts_post(bjx_nuts_t nt, int32_t depth_arg, int32_t s_arg, int64_t n_rows_arg,
            const int32_t* __restrict__ idx, const int64_t* __restrict__ ctl, float* qf,
            const float* __restrict__ logp_f, const float* __restrict__ gf, int fuse_next) {
  const StepCtx cx = make_ctx(nt, depth_arg, s_arg, n_rows_arg, ctl);
  const int32_t depth = cx.depth, s = cx.s;
  const int64_t n_rows = cx.n_rows;
  for (int64_t b = wave_row0(); b < n_rows; b += wave_row_stride()) {
    const int64_t c = idx ? (int64_t)idx[b] : b;
    if (!IS(BJX_NUTS_I_ACTIVE, c) || !IS(BJX_NUTS_I_SUB_ACTIVE, c)) continue;
    nuts_post_chain<VEC, DENSE>(nt, cx, c, b, depth, s, qf, logp_f, gf, fuse_next != 0);
  }
}

// same launch contract, register-resident leaf (diagonal metric, 16-byte rows, D <= 256 * NI)
template <int NI>
__global__ void __launch_bounds__(kBlock)
k_nuts_post_res(bjx_nuts_t nt, int32_t depth_arg, int32_t s_arg, int64_t n_rows_arg,
                const int32_t* __restrict__ idx, const int64_t* __restrict__ ctl, float* qf,
                const float* __restrict__ logp_f, const float* __restrict__ gf, int fuse_next) {
  const StepCtx cx = make_ctx(nt, depth_arg, s_arg, n_rows_arg, ctl);
  const int32_t depth = cx.depth, s = cx.s;
  const int64_t n_rows = cx.n_rows;
  for (int64_t b = wave_row0(); b < n_rows; b += wave_row_stride()) {
    const int64_t c = idx ? (int64_t)idx[b] : b;
    if (!IS(BJX_NUTS_I_ACTIVE, c) || !IS(BJX_NUTS_I_SUB_ACTIVE, c)) continue;
    nuts_post_chain_resident<4, NI>(nt, cx, c, b, depth, s, qf, logp_f, gf, fuse_next != 0);
  }
}

// ------------------------------------------------------------------------------------ merge
// End of doubling `depth` for chain c: biased progressive sampling of the new subtree's proposal,
// momentum-sum merge, U-turn of the whole trajectory, stop flags (trajectory.py:680-727,
// proposal.py:146-176, nuts.py:303-305).  Returns true when the tree keeps growing.
template <int VEC, bool DENSE>
__device__ __forceinline__ bool nuts_merge_chain(const bjx_nuts_t& nt, const StepCtx& kcx, int64_t c,
                                                 int32_t depth) {
  const int lane = threadIdx.x & 63;
  const bool sdiv = IS(BJX_NUTS_I_SDIV, c) != 0, sturn = IS(BJX_NUTS_I_STURN, c) != 0;
  const int64_t base = c * nt.D;
  const float pw = FS(BJX_NUTS_F_PW, c), sw = FS(BJX_NUTS_F_SW, c);
  const float pslpa = FS(BJX_NUTS_F_PSLPA, c), sslpa = FS(BJX_NUTS_F_SSLPA, c);
  bool take = false;
  float new_pw = pw;
  // exp(sw - pw), logaddexp(pslpa, sslpa), logaddexp(pw, sw) in one pass
  const Scalars3 sc = scalars3((double)(sw - pw), pslpa, sslpa, pw, sw);
  const float new_pslpa = sc.lae1;
  if (!(sdiv || sturn)) {  // progressive_biased_sampling (proposal.py:146-176)
    const Key kp{(uint32_t)__builtin_amdgcn_readfirstlane(IS(BJX_NUTS_I_KP, c)),
                 (uint32_t)__builtin_amdgcn_readfirstlane(IS(BJX_NUTS_I_KPB, c))};  // nuts_begin_doubling
    const float pa = min1_nan(sc.e0);
    take = key_uniform(kp) < pa;
    new_pw = sc.lae2;
  }
  // merged trajectory: momentum sum + U-turn of the whole trajectory (trajectory.py:696-710)
  const float* im = nt.imm + c * nt.imm_stride;
  double a_left = 0.0, a_right = 0.0;
  BJX_ROW_SWEEP(j0) {
    Row<VEC> m = ldr<VEC>(nt.msum + base + j0);
    const Row<VEC> sm = ldr<VEC>(nt.Smsum + base + j0), pl = ldr<VEC>(nt.Lp + base + j0),
                   pr = ldr<VEC>(nt.Rp + base + j0);
    Row<VEC> vl, vr;
    if constexpr (DENSE) {
      vl = ldr<VEC>(nt.Lv + base + j0);
      vr = ldr<VEC>(nt.Rv + base + j0);
    } else {
      const Row<VEC> mm = ldr<VEC>(im + j0);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        vl.v[e] = mm.v[e] * pl.v[e];
        vr.v[e] = mm.v[e] * pr.v[e];
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      m.v[e] = m.v[e] + sm.v[e];
      const float rho = m.v[e] - (pr.v[e] + pl.v[e]) * 0.5f;
      a_left += (double)vl.v[e] * (double)rho;
      a_right += (double)vr.v[e] * (double)rho;
    }
    str<VEC>(nt.msum + base + j0, m);
    if (take) {
      str<VEC>(nt.Pq + base + j0, ldr<VEC>(nt.Sq + base + j0));
      str<VEC>(nt.Pg + base + j0, ldr<VEC>(nt.Sg + base + j0));
    }
  }
  a_left = wave_sum(a_left);
  a_right = wave_sum(a_right);
  const bool turn = sturn || ((float)a_left <= 0.0f) || ((float)a_right <= 0.0f);
  const bool grow = !sdiv && !turn && depth + 1 < nt.max_depth;
  if (lane == 0) {
    const int n = IS(BJX_NUTS_I_NSTATES, c) + IS(BJX_NUTS_I_SUBN, c);
    FS(BJX_NUTS_F_PW, c) = new_pw;
    FS(BJX_NUTS_F_PSLPA, c) = new_pslpa;
    if (take) {
      FS(BJX_NUTS_F_PLOGP, c) = FS(BJX_NUTS_F_SLOGP, c);
      FS(BJX_NUTS_F_PENERGY, c) = FS(BJX_NUTS_F_SENERGY, c);
    }
    FS(BJX_NUTS_F_ACC, c) = exp_cr(new_pslpa) / (float)n;  // nuts.py:303-305
    IS(BJX_NUTS_I_NSTATES, c) = n;
    IS(BJX_NUTS_I_DIV, c) = sdiv ? 1 : 0;
    IS(BJX_NUTS_I_TURN, c) = turn ? 1 : 0;
    IS(BJX_NUTS_I_DEPTH, c) = depth + 1;
    IS(BJX_NUTS_I_SUB_ACTIVE, c) = 0;
    IS(BJX_NUTS_I_ACTIVE, c) = grow ? 1 : 0;
  }
  return grow;
}

template <int VEC, bool DENSE>
__global__ void __launch_bounds__(kBlock)
k_nuts_merge(bjx_nuts_t nt, int32_t depth, int64_t n_rows, const int32_t* __restrict__ idx) {
  const StepCtx kcx{depth, 0, n_rows, Key{nt.key0, nt.key1}, nt.chain_offset, nt.step_fold};
  for (int64_t b = wave_row0(); b < n_rows; b += wave_row_stride()) {
    const int64_t c = idx ? (int64_t)idx[b] : b;
    if (!IS(BJX_NUTS_I_ACTIVE, c)) continue;
    nuts_merge_chain<VEC, DENSE>(nt, kcx, c, depth);
  }
}

// ------------------------------------------------------------------------------------ free-running chains
// One tick of the asynchronous schedule (include/bjx_nuts.h): every chain that is not finished
// ends the tick with the opening half of a leapfrog done and its new position in qf[c].
//   phase 1: post(leaf) [-> fused pre(next leaf)]  | subtree complete -> phase 3       (k_nuts_async_leaf)
//   phase 3: merge [-> begin next doubling + pre -> phase 1]
//            | record the transition, accept its proposal -> phase 0                   (k_nuts_async_boundary)
//   phase 0: momentum draw, tree init, begin doubling 0, pre -> phase 1
// Two kernels because the leaf path runs for every chain in every tick and must stay light in
// registers (occupancy hides its dependent memory round trips); the boundary path is heavy and rare.
// All per-chain decisions are wave-uniform; scalars written by lane 0 and read by the whole wave
// later in the same kernel are separated by a workgroup-scope fence (one CU, one L1).
__device__ __forceinline__ StepCtx async_ctx(const bjx_nuts_t& nt, const bjx_nuts_async_t& ax, int32_t t) {
  StepCtx cx;
  cx.depth = 0;
  cx.s = 0;
  cx.n_rows = nt.N;
  cx.off = nt.chain_offset;
  if (ax.step_keys) {
    cx.key = Key{ax.step_keys[2 * (int64_t)t], ax.step_keys[2 * (int64_t)t + 1]};
    cx.fold = -1;
  } else {
    cx.key = Key{nt.key0, nt.key1};
    cx.fold = (int64_t)ax.t_first + t;
  }
  return cx;
}

// Work distribution of the general free-running kernels: one wave per compact row; a wave whose chain has no work
// in this kernel exits after two loads.  (A grouped form -- one wave working off eight consecutive rows -- serialised
// the live chains of a group in the tail of a run and was dropped: C3 87 -> 101 M/s, NOTEBOOK.md section 7.)
// occupancy hint of the fused tick kernel: 3 waves per SIMD (168 VGPRs); 4 forces 140 B of spills
// per lane and measured slower (C3: 101 vs 108 M/s)
#ifndef BJX_FUSED_WAVES
#define BJX_FUSED_WAVES 3
#endif

// rows to process: the host's count, or -- so that one recorded launch sequence serves every batch
// size of the tail -- the smaller device-side count the last compaction wrote
__device__ __forceinline__ int64_t async_n_rows(const bjx_nuts_async_t& ax) {
  if (!ax.n_rows_dev) return ax.n_rows;
  const int64_t n = (int64_t)__builtin_amdgcn_readfirstlane(*ax.n_rows_dev);
  return n < ax.n_rows ? n : ax.n_rows;
}

// f(chain, compact row, phase) for every chain of the compact rows whose phase is want_a or want_b
template <class F>
__device__ __forceinline__ void async_for_each_chain(const bjx_nuts_async_t& ax, int want_a, int want_b, F f) {
  const int64_t n_rows = async_n_rows(ax);
  for (int64_t b = wave_row0(); b < n_rows; b += wave_row_stride()) {
    const int chain = ax.rows ? ax.rows[b] : (int)b;
    const int ph = ax.phase[chain];
    if (ph == want_a || ph == want_b)
      f((int64_t)__builtin_amdgcn_readfirstlane(chain), b, __builtin_amdgcn_readfirstlane(ph));
  }
}

// Tick, part 1 (every chain with a leaf in flight, phase 1): the second half of the leaf and, when
// the subtree keeps integrating, the fused opening half of the next leaf.  A chain whose subtree is
// complete moves to phase 3 and is finished by part 2.
// NI = 0: general sweeps; NI > 0: register-resident leaf (VEC == 4, D <= 256 * NI)
// Returns true when the chain's subtree is complete (phase 3 written).
template <int VEC, int NI, bool DENSE = false>
__device__ __forceinline__ bool async_leaf_chain(const bjx_nuts_t& nt, const bjx_nuts_async_t& ax,
                                                 float* qf, const float* __restrict__ logp_f,
                                                 const float* __restrict__ gf, int64_t c, int64_t b) {
  static_assert(!DENSE || NI == 0, "the register-resident leaf is for the diagonal metric");
  const int lane = threadIdx.x & 63;
  if (ax.int_stages > 1) {
    // multi-stage palindromic integrator (integrators.py:128-146) on the general kernel (round 6: rows beyond 1 024
    // floats, 4-byte rows, per-chain dense metrics; the lean tick kernel keeps its own counter in the record): a leaf
    // lasts int_stages ticks.  The first int_stages - 1 gradients drive a middle stage on the integrating end --
    // k_nuts_mid's arithmetic: p += (dir eps b_i) g ; q += (dir eps a_i) M^-1 p -- and only the last one closes the leaf.
    const int st = __builtin_amdgcn_readfirstlane(IS(BJX_NUTS_I_STAGE, c));
    if (st < ax.int_stages - 1) {
      const int dir = IS(BJX_NUTS_I_DIR, c);
      const float deps = (float)dir * chain_eps(nt, c);
      nuts_open_half<VEC, DENSE>(nt, c, dir, deps * ax.int_mid_drift[st], deps * ax.int_mid_kick[st], gf + b * nt.D,
                                 qf + b * nt.D);
      if (lane == 0) IS(BJX_NUTS_I_STAGE, c) = st + 1;
      return false;
    }
    if (lane == 0) IS(BJX_NUTS_I_STAGE, c) = 0;
  }
  const StepCtx cx = async_ctx(nt, ax, ax.t[c]);
  const int32_t depth = IS(BJX_NUTS_I_DEPTH, c);
  const int32_t s = IS(BJX_NUTS_I_SUBN, c);  // states already in the subtree = index of this leaf
  const bool last = (s + 1) >= (1 << depth);
  bool stop;
  if constexpr (NI > 0) stop = nuts_post_chain_resident<VEC, NI>(nt, cx, c, b, depth, s, qf, logp_f, gf, !last);
  else stop = nuts_post_chain<VEC, DENSE>(nt, cx, c, b, depth, s, qf, logp_f, gf, !last);
  if ((stop || last) && lane == 0) ax.phase[c] = 3;
  return stop || last;
}

// Per-chain window adaptation at the end of transition t (include/bjx_nuts.h, adapt_* fields): the
// arithmetic of k_welford_update_diag, k_da_update, k_welford_final_diag and k_da_init
// (bjx_adapt.hip), expression for expression, applied to one chain by its own wave.
template <int VEC>
__device__ __forceinline__ void async_adapt_chain(const bjx_nuts_t& nt, const bjx_nuts_async_t& ax,
                                                  int64_t c, int32_t t) {
  const int lane = threadIdx.x & 63;
  const float* tab = ax.adapt_tab + (int64_t)t * BJX_NUTS_ADAPT_COLS;
  const int flags = (int)tab[BJX_NUTS_AT_FLAGS];
  const int64_t base = c * nt.D;
  if (flags & 1) {  // slow window: Welford update with the chain's new position (mass_matrix.py:410-435)
    const float n = tab[BJX_NUTS_AT_WEL_N];
    BJX_ROW_SWEEP(j0) {
      const Row<VEC> x = ldr<VEC>(ax.q + base + j0);
      Row<VEC> m = ldr<VEC>(ax.adapt_mean + base + j0), s2 = ldr<VEC>(ax.adapt_m2 + base + j0);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float d = x.v[e] - m.v[e];
        const float mo = m.v[e] + d / n;
        s2.v[e] = fmaf(d, x.v[e] - mo, s2.v[e]);
        m.v[e] = mo;
      }
      str<VEC>(ax.adapt_mean + base + j0, m);
      str<VEC>(ax.adapt_m2 + base + j0, s2);
    }
  }
  // dual averaging with gradient = target - acceptance_rate (dual_averaging.py:101-123)
  const float reg = tab[BJX_NUTS_AT_DA_REG], inv_reg = tab[BJX_NUTS_AT_DA_INV_REG];
  const float eta = tab[BJX_NUTS_AT_DA_ETA], coef = tab[BJX_NUTS_AT_DA_COEF];
  const float g = ax.adapt_target - FS(BJX_NUTS_F_ACC, c);
  float ae = (1.0f - inv_reg) * ax.adapt_avg_err[c] + g / reg;
  const float lx_prev = ax.adapt_log_x[c];
  float mu = ax.adapt_mu[c];
  float lx = mu - coef * ae;
  float lxa = eta * lx_prev + (1.0f - eta) * ax.adapt_log_x_avg[c];
  float step = exp_cr(lx);
  if (flags & 2) {  // window end: metric update + Welford reset, dual averaging restarts (staged_adaptation.py:233-249)
    const float nm1 = tab[BJX_NUTS_AT_FIN_NM1], beta_data = tab[BJX_NUTS_AT_FIN_BETA_DATA];
    const float beta_prev = tab[BJX_NUTS_AT_FIN_BETA_PREV], freg = tab[BJX_NUTS_AT_FIN_REG];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // m2 just written by this wave's lanes
    BJX_ROW_SWEEP(j0) {
      const Row<VEC> s2 = ldr<VEC>(ax.adapt_m2 + base + j0), pv = ldr<VEC>(ax.adapt_imm + base + j0);
      Row<VEC> im, zero;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float cov = s2.v[e] / nm1;
        im.v[e] = fmaf(beta_prev, pv.v[e], beta_data * cov) + freg;
        zero.v[e] = 0.0f;
      }
      str<VEC>(ax.adapt_imm + base + j0, im);
      str<VEC>(ax.adapt_mean + base + j0, zero);
      str<VEC>(ax.adapt_m2 + base + j0, zero);
    }
    const float x = exp_cr(lxa);
    lx = (float)log((double)x);
    mu = (float)log((double)(10.0f * x));
    lxa = 0.0f;
    ae = 0.0f;
    step = exp_cr(lx);
  }
  if (lane == 0) {
    ax.adapt_avg_err[c] = ae;
    ax.adapt_log_x[c] = lx;
    ax.adapt_log_x_avg[c] = lxa;
    ax.adapt_mu[c] = mu;
    ax.adapt_step_size[c] = step;
    if (ax.out_step_size) ax.out_step_size[(int64_t)t * nt.N + c] = step;
  }
}

// Tick, part 2 (phase 3: subtree complete; phase 0: start a transition): merge, then either the
// next doubling, or record the finished transition, accept its proposal and start the next one.
template <int VEC, bool DENSE = false>
__device__ __forceinline__ void async_boundary_chain(const bjx_nuts_t& nt, const bjx_nuts_async_t& ax,
                                                     float* qf, int64_t c, int64_t b, int phase) {
  const int lane = threadIdx.x & 63;
  {
    int32_t t = ax.t[c];
    StepCtx cx = async_ctx(nt, ax, t);
    const int64_t base = c * nt.D;
    float* qrow = qf + b * nt.D;  // this chain's row of the callable's batch
    if (phase == 3) {
      const int32_t depth = IS(BJX_NUTS_I_DEPTH, c);
      const bool grow = nuts_merge_chain<VEC, DENSE>(nt, cx, c, depth);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      if (grow) {
        const int dir = nuts_begin_doubling(nt, cx, c, depth + 1);
        const float deps = (float)dir * chain_eps(nt, c);
        nuts_open_half<VEC, DENSE>(nt, c, dir, deps * int_drift(nt), deps * int_kick(nt),
                                   (dir > 0 ? nt.Rg : nt.Lg) + base, qrow);  // (deps, deps / 2 for velocity Verlet)
        if (lane == 0) ax.phase[c] = 1;
        return;
      }
      // transition t is complete: record it and make the proposal the chain's state
      const int64_t row = (int64_t)t * nt.N + c;
      BJX_ROW_SWEEP(j0) {
        const Row<VEC> q = ldr<VEC>(nt.Pq + base + j0);
        str<VEC>(ax.q + base + j0, q);
        str<VEC>(ax.g + base + j0, ldr<VEC>(nt.Pg + base + j0));
        if (ax.out_position) str<VEC>(ax.out_position + row * nt.D + j0, q);
      }
      if (lane == 0) {
        const float lp = FS(BJX_NUTS_F_PLOGP, c);
        ax.logp[c] = lp;
        if (ax.out_logdensity) ax.out_logdensity[row] = lp;
        if (ax.out_acceptance_rate) ax.out_acceptance_rate[row] = FS(BJX_NUTS_F_ACC, c);
        if (ax.out_energy) ax.out_energy[row] = FS(BJX_NUTS_F_PENERGY, c);
        if (ax.out_num_integration_steps) ax.out_num_integration_steps[row] = IS(BJX_NUTS_I_NSTATES, c);
        if (ax.out_num_trajectory_expansions) ax.out_num_trajectory_expansions[row] = IS(BJX_NUTS_I_DEPTH, c);
        if (ax.out_is_divergent) ax.out_is_divergent[row] = (uint8_t)(IS(BJX_NUTS_I_DIV, c) != 0);
        if (ax.out_is_turning) ax.out_is_turning[row] = (uint8_t)(IS(BJX_NUTS_I_TURN, c) != 0);
      }
      if (ax.adapt_tab) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // ax.q: written above, read by the Welford update
        async_adapt_chain<VEC>(nt, ax, c, t);
      }
      t += 1;
      if (lane == 0) ax.t[c] = t;
      if (t >= ax.n_steps) {
        if (lane == 0) {
          ax.phase[c] = 2;
          atomicAdd(ax.n_done, 1);
        }
        return;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      cx = async_ctx(nt, ax, t);
    }
    // start transition t -- momentum draw (hmc.py:299-302, metrics.py:260-270) with the lane <->
    // element mapping of the other sweeps, then the tree of nuts.py:278-294
    const Key kc = chain_key(cx.key, (uint64_t)(c + cx.off), cx.fold);
    const Key km = key_child(kc, 0);  // split(kc, 2)[0]
    double acc = 0.0;
    if constexpr (DENSE) {
      // the arithmetic of bjx_hmc_momentum_dense_pc (metrics.py:260-270, util.py:23-91), which the
      // lockstep step uses for NUTS: z = normal(km); p = L^{-T} z; v = M^{-1} p, fp64 accumulated in
      // ascending j; K = v.p / 2 summed lane-strided then across the wave
      const float* Mt = ax.mass_sqrt_t + c * nt.Mdense_stride;
      const float* Mi = nt.Mdense + c * nt.Mdense_stride;
      float* pz = ax.p + base;
      float* v0 = ax.v0 + base;
      for (int64_t ic = 0; ic < nt.D; ic += 64) {
        const int64_t i = ic + lane;
        const double a = matvec_t_lane(Mt, nt.D, i, [&](int64_t j) {
          return normal_from_bits(key_bits32(km, (uint64_t)j));
        });
        if (i < nt.D) pz[i] = (float)a;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // p: written lane-wise, read by every lane
      for (int64_t ic = 0; ic < nt.D; ic += 64) {
        const int64_t i = ic + lane;
        const double a = matvec_t_lane(Mi, nt.D, i, [&](int64_t j) { return pz[j]; });
        if (i < nt.D) {
          const float v = (float)a;
          v0[i] = v;
          acc += (double)v * (double)pz[i];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // v0 is read by nuts_init_chain below
    } else {
      const float* im = nt.imm + c * nt.imm_stride;
      BJX_ROW_SWEEP(j0) {
        const Row<VEC> m = ldr<VEC>(im + j0);
        Row<VEC> pv;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float z = normal_from_bits(key_bits32(km, (uint64_t)(j0 + e)));
          const float ms = 1.0f / sqrtf(m.v[e]);
          pv.v[e] = ms * z;
          acc += (double)(m.v[e] * pv.v[e]) * (double)pv.v[e];
        }
        str<VEC>(ax.p + base + j0, pv);
      }
    }
    acc = wave_sum(acc);
    nuts_init_chain<VEC, DENSE>(nt, c, ax.logp[c], 0.5f * (float)acc);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const int dir = nuts_begin_doubling(nt, cx, c, 0);
    const float deps = (float)dir * chain_eps(nt, c);
    nuts_open_half<VEC, DENSE>(nt, c, dir, deps * int_drift(nt), deps * int_kick(nt),
                               (dir > 0 ? nt.Rg : nt.Lg) + base, qrow);
    if (lane == 0) ax.phase[c] = 1;
  }
}

// The GENERAL free-running tick (every metric, every row width, 4-byte sweeps when D % 4 != 0 or a buffer is not
// 16-byte aligned): leaf, then -- same wave, after a fence -- the transition end the leaf may have produced, in ONE
// launch.  Diagonal metrics with 16-byte rows of at most 1 024 floats take k_nuts_async_tick3 instead.  (Round 5:
// the round-1 two-launch form of this tick -- k_nuts_async_leaf + k_nuts_async_boundary -- is gone.)
template <int VEC, int NI, bool DENSE = false>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(BJX_FUSED_WAVES)))
k_nuts_async_fused(bjx_nuts_t nt, bjx_nuts_async_t ax, float* qf, const float* __restrict__ logp_f,
                   const float* __restrict__ gf) {
  async_for_each_chain(ax, 1, 0, [&](int64_t c, int64_t b, int phase) {
    if (phase == 1) {
      if (!async_leaf_chain<VEC, NI, DENSE>(nt, ax, qf, logp_f, gf, c, b)) return;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      phase = 3;
    }
    async_boundary_chain<VEC, DENSE>(nt, ax, qf, c, b, phase);
  });
}

// ------------------------------------------------------------------------------------ free-running chains, v2
// The same schedule with the data movement and the dependent round trips cut to what the algorithm
// needs (diagonal metric, 16-byte rows, D <= 256 * NI; needs bjx_nuts_async_t.rec / front_p).
// A tick kernel's duration is (chains / resident waves) x (lifetime of a wave), and a wave's
// lifetime is its chain of DEPENDENT memory round trips; the kernels above need 4-6 per leaf
// (row list -> phase -> slot tables -> rows -> checkpoint rows -> merge rows ...).  Here:
//  * every per-chain scalar a leaf needs sits in ONE 128-byte record rec[c] (the slot tables spread
//    them over 25 cache lines), loaded by one wave instruction and broadcast with readlane;
//  * while a subtree integrates, the moving end's position lives in the callable's row qf[b], its
//    momentum in front_p[c] and its gradient in registers only -- all direction independent, so
//    the record, the phase and ALL rows of a leaf are requested in the first round trip; the second
//    one (checkpoint rows of an odd leaf, the two merge rows of a subtree's last leaf) is issued as
//    soon as the record has arrived and overlaps the leaf's arithmetic;
//  * the end arrays (Lq/Lp/Lg, Rq/Rp/Rg) are written only when a doubling LEAVES that end: a leaf no
//    longer writes the end position and gradient (nor reads the position twice -- fq and qf[b] held
//    the same values), and continuing a trajectory in the same direction moves no extra row;
//  * a leaf that completes a subtree merges it at once and opens the next doubling (no second
//    kernel visit with five row loads);
//  * a transition starts lazily: "this end / the proposal / the momentum sum is still the initial
//    state" is a bit of the record and those rows are read from q0 / p0 / g0 while it is set,
//    instead of nine row copies per transition (nuts.py:278-291 builds the tree from z0).
// Arithmetic, keys and decisions are those of the functions above, expression for expression; the
// per-transition records are identical (tests/test_nuts_free_gpu.py, test_full_shape_gpu.py).
// The trajectory-end states are NOT kept after a transition ends (run_free does not expose them).
enum { LZ_L = 1, LZ_R = 2, LZ_P = 4, LZ_M = 8 };
// words of rec[c] (BJX_NUTS_REC_WORDS = 32 per chain)
// words 0 .. 15: what a leaf that keeps integrating reads and writes; 16 .. 27: touched only when a subtree
// is merged or a transition starts (the v3 leaf loads them there)
enum {
  RW_H0 = 0, RW_SW, RW_SSLPA, RW_SLOGP, RW_SENERGY, RW_DEPTH, RW_SUBN, RW_DIR, RW_LAZY, RW_KT, RW_KTB, RW_EPS,
  RW_U0,  // .. RW_U0 + 3: the progressive-sampling uniforms of leaves (s & ~3) .. (s | 3) of the current subtree
  RW_PW = RW_U0 + 4, RW_PSLPA, RW_PLOGP, RW_PENERGY, RW_ACC, RW_NSTATES, RW_KP, RW_KPB, RW_IK, RW_IKB,
  RW_DIV, RW_TURN,
  RW_END,
  RW_STAGE = RW_END,  // multi-stage integrators (bjx_nuts_async_t.int_stages > 1): gradients of the leaf in flight already used
  RW_LLOGP = RW_END + 1, RW_RLOGP  // bjx_nuts_async_t.keep_ends: log-density of the leftmost / rightmost trajectory state
};
static_assert(RW_U0 == 12 && RW_PW == 16 && RW_END == 28 && RW_STAGE < BJX_NUTS_REC_WORDS && RW_RLOGP < BJX_NUTS_REC_WORDS,
              "record layout");

__device__ __forceinline__ int rec_i(int w, int k) { return __builtin_amdgcn_readlane(w, k); }
__device__ __forceinline__ float rec_f(int w, int k) { return __int_as_float(__builtin_amdgcn_readlane(w, k)); }
// lane k of the wave-wide record register takes the (wave-uniform) value v
__device__ __forceinline__ void rec_set_i(int& w, int k, int v) { if ((int)(threadIdx.x & 63) == k) w = v; }
__device__ __forceinline__ void rec_set_f(int& w, int k, float v) { rec_set_i(w, k, __float_as_int(v)); }

// Direction and keys of doubling `depth` (trajectory.py:645-650) into the record register.
__device__ __forceinline__ int begin_doubling_rec(int& w, Key ik, int32_t depth) {
  const Key subkey = key_child(ik, (uint64_t)depth);
  // split(subkey, 3): the three children in lanes 0 .. 2 of ONE block instead of three blocks
  const int lane_ = threadIdx.x & 63;
  const Key ch = key_child(subkey, (uint64_t)(lane_ < 3 ? lane_ : 0));
  const Key kd{(uint32_t)__builtin_amdgcn_readlane((int)ch.k0, 0), (uint32_t)__builtin_amdgcn_readlane((int)ch.k1, 0)};
  const Key kt{(uint32_t)__builtin_amdgcn_readlane((int)ch.k0, 1), (uint32_t)__builtin_amdgcn_readlane((int)ch.k1, 1)};
  const Key kp{(uint32_t)__builtin_amdgcn_readlane((int)ch.k0, 2), (uint32_t)__builtin_amdgcn_readlane((int)ch.k1, 2)};
  const int dir = key_uniform(kd) < 0.5f ? 1 : -1;
  rec_set_i(w, RW_KT, (int)kt.k0);
  rec_set_i(w, RW_KTB, (int)kt.k1);
  rec_set_i(w, RW_KP, (int)kp.k0);
  rec_set_i(w, RW_KPB, (int)kp.k1);
  rec_set_i(w, RW_DIR, dir);
  rec_set_i(w, RW_SUBN, 0);
  return dir;
}

template <int NI>
struct LeafRows {  // requested before the chain's phase and record are known (direction independent)
  Row<4> G[NI], M[NI], P[NI], S[NI], X[NI];
};
// Several ticks of one chain in one launch (engine-resident target, async_multi_tick_row): while a subtree
// keeps integrating, everything leaf s + 1 reads is what leaf s just computed -- the rows stay in registers
// (`hot`) and so does the checkpoint an even leaf stores for the odd leaf after it (PK).  A hot leaf has no
// top-of-tick fence, so the loads it still makes (deeper checkpoint levels, merge rows: stored by this same
// wave, earlier in the launch) are preceded by a workgroup-scope fence of their own.
#ifdef BJX_TICK_PROBE
// Build-time instrumentation (make PROBE=1; never in the product build): s_memtime stamps between the
// stages of a multi-tick leaf, summed by the wave of compact row 0 into bjx_tick_probe[] (100 MHz ticks).
__device__ unsigned long long bjx_tick_probe[16];
#define BJX_PROBE(hs_, k_)                                            \
  do {                                                                \
    const unsigned long long t_ = __builtin_readcyclecounter();       \
    (hs_)->acc[k_] += t_ - (hs_)->last;                               \
    (hs_)->last = t_;                                                 \
  } while (0)
#else
#define BJX_PROBE(hs_, k_) do { } while (0)
#endif
template <int NI>
struct HotState {
#ifdef BJX_TICK_PROBE
  unsigned long long acc[12], last;
#endif
  bool hot, pk_valid;
  bool merged;  // the last leaf completed a subtree and opened the next doubling
  Row<4> PK[NI];
};

// One leaf of chain c (phase 1), record register `w` and rows already requested.  Returns 0 = a
// leaf is in flight again (phase stays 1), 1 = the transition is complete (phase 3 written;
// async_end2_chain finishes it).
template <int NI, bool LOOP = false, bool FULL = false>
__device__ __forceinline__ int async_leaf2_chain(const bjx_nuts_t& nt, const bjx_nuts_async_t& ax, float* qf,
                                                 float lp, int64_t c, int64_t b, int& w, LeafRows<NI>& R,
                                                 HotState<NI>* hs = nullptr) {
  constexpr int VEC = 4;
  const int lane = threadIdx.x & 63;
  const int32_t depth = rec_i(w, RW_DEPTH);
  const int32_t s = rec_i(w, RW_SUBN);  // states already in the subtree = index of this leaf
  const int dir = rec_i(w, RW_DIR);
  int lazy = rec_i(w, RW_LAZY);
  const float eps = rec_f(w, RW_EPS);
  const float deps = (float)dir * eps;
  const float h = deps * int_kick(nt);    // closing kick b_K = b_1, and the next leaf's opening kick (0.5 for velocity Verlet)
  const float dd = deps * int_drift(nt);  // first drift a_1 of the next leaf (deps * 1.0f == deps for velocity Verlet)
  const int64_t base = c * nt.D;
  float* fpp = ax.front_p + base;
  float* qn = qf + b * nt.D;
  float* sm = nt.Smsum + base;
  const float H0 = rec_f(w, RW_H0), sw = rec_f(w, RW_SW), sslpa = rec_f(w, RW_SSLPA);
  const bool last = (s + 1) >= (1 << depth);
  const uint32_t us = (uint32_t)s;  // checkpoint indices (termination.py:75-84)
  const int idx_max = __popc(us >> 1);
  const int nsub = __popc((~us & (us + 1u)) - 1u);
  const int idx_min = idx_max - nsub + 1;
  const bool even = (us & 1u) == 0u;

  uint32_t j0[NI];  // 32-bit element offsets: rows are addressed as (uniform base) + (32-bit lane offset)
  bool ok[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    j0[k] = ((uint32_t)lane + 64u * k) * VEC;
    ok[k] = j0[k] < (uint32_t)nt.D;
  }
  // second round trip, issued now: first checkpoint level of an odd leaf, merge rows of a last leaf
  const int other_bit = dir > 0 ? LZ_L : LZ_R;
  const float* op = ((lazy & other_bit) ? nt.p0 : (dir > 0 ? nt.Lp : nt.Rp)) + base;
  const float* ms_src = ((lazy & LZ_M) ? nt.p0 : nt.msum) + base;
  // The first checkpoint level of an odd leaf s was stored by leaf s - 1: its momentum row is loaded,
  // its momentum-SUM row is the subtree sum before this leaf, i.e. the S row already in registers
  // (Smsum after leaf s - 1), so it is neither loaded here nor -- when no later leaf reads it, s - 1
  // not a multiple of 4 -- stored by leaf s - 1.
  Row<VEC> C0[NI], C1[NI], MS[NI], OP[NI];
  bool hot = false;
  if constexpr (LOOP) hot = hs->hot;
  if (nsub > 0) {
    bool from_regs = false;
    if constexpr (LOOP) from_regs = hs->pk_valid;  // leaf s - 1 ran in this launch: its checkpoint is in registers
    const float* r_ck = nt.ckpt_r + (c * nt.max_depth + idx_max) * nt.D;
    if (LOOP && hot && !from_regs) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (not reached)
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (FULL || ok[k]) {
        if constexpr (LOOP) C0[k] = from_regs ? hs->PK[k] : ldr<VEC>(r_ck + j0[k]);
        else C0[k] = ldr<VEC>(r_ck + j0[k]);
        C1[k] = R.S[k];
      }
  }
  if (last) {
    if (LOOP && hot) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (FULL || ok[k]) {
        MS[k] = ldr<VEC>(ms_src + j0[k]);
        OP[k] = ldr<VEC>(op + j0[k]);
      }
  }

  // the uniform of the progressive sampling step below needs the record only (fold_in(kt, s): two
  // threefry blocks, ~250 dependent instructions): drawn here, it runs under the row loads of this leaf
  // instead of after its energy reduction
  // FOUR leaves' uniforms per draw: the two blocks cost the same whether one lane or four use them, so lanes
  // RW_U0 .. RW_U0 + 3 of the record take uniform(fold_in(kt, s + 0 .. 3)) at every fourth leaf (the first leaf
  // of a subtree has s = 0) and the three leaves after it read theirs from the record -- 43 instead of 170
  // vector instructions per leaf on average (SQ counters: 1 068 per leapfrog before)
  if ((s & 3) == 0) {
    const Key kt{(uint32_t)rec_i(w, RW_KT), (uint32_t)rec_i(w, RW_KTB)};
    const uint32_t sl = (uint32_t)s + ((uint32_t)(lane - RW_U0) & 3u);
    const float ul = key_uniform(key_child(kt, (uint64_t)sl));
    if (lane >= RW_U0 && lane < RW_U0 + 4) w = __float_as_int(ul);
  }
  const float u = rec_f(w, RW_U0 + (s & 3));
  if constexpr (LOOP) BJX_PROBE(hs, 0);  // uniform draw

  // pass 1: closing half kick, kinetic energy
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < NI; ++k)
    if (FULL || ok[k]) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        R.P[k].v[e] = fmaf(h, R.G[k].v[e], R.P[k].v[e]);
        acc += (double)(R.M[k].v[e] * R.P[k].v[e]) * (double)R.P[k].v[e];
      }
    }
  acc = wave_sum(acc);
  const float ke = 0.5f * (float)acc;
  if constexpr (LOOP) BJX_PROBE(hs, 1);  // pass 1
  const float e_new = -lp + ke;  // hmc_energy (trajectory.py:745-748)
  float wgt = H0 - e_new;        // proposal.py:91-95
  if (wgt != wgt) wgt = -__builtin_inff();
  const float slpa_new = fminf(wgt, 0.0f);
  const bool sdiv = (-wgt) > nt.divergence_threshold;  // trajectory.py:325
  bool take;
  float Wn, Sn;
  if (s == 0) {
    take = true;
    Wn = wgt;
    Sn = slpa_new;
  } else {  // progressive uniform sampling (trajectory.py:329-339, proposal.py:118-143)
    const Scalars3 sc = scalars3(-(double)(wgt - sw), sw, wgt, sslpa, slpa_new);
    take = u < sc.r0;
    Wn = sc.lae1;
    Sn = sc.lae2;
  }

  if constexpr (LOOP) BJX_PROBE(hs, 2);  // scalars3
  // pass 2: momentum-sum append, checkpoint store, subtree-proposal state copy
#pragma unroll
  for (int k = 0; k < NI; ++k)
    if (FULL || ok[k]) {
      if (s != 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) R.S[k].v[e] = R.S[k].v[e] + R.P[k].v[e];
      } else {
        R.S[k] = R.P[k];
      }
      // checkpoints are read by later leaves of the same subtree only: none follow the last leaf
      // or a divergence (even leaves run no U-turn check, so `sdiv` is all that can stop them)
      if (even && !last && !sdiv) {
        if constexpr (LOOP) hs->PK[k] = R.P[k];
        str<VEC>(nt.ckpt_r + (c * nt.max_depth + idx_max) * nt.D + j0[k], R.P[k]);
        // read from memory only as a SECOND or deeper level, i.e. by leaves s + 3, s + 7, ...: s % 4 == 0
        if ((us & 3u) == 0u) str<VEC>(nt.ckpt_rs + (c * nt.max_depth + idx_max) * nt.D + j0[k], R.S[k]);
      }
      if (take) {
        str<VEC>(nt.Sq + base + j0[k], R.X[k]);
        str<VEC>(nt.Sg + base + j0[k], R.G[k]);
      }
    }

  if constexpr (LOOP) hs->pk_valid = even && !last && !sdiv;
  if constexpr (LOOP) BJX_PROBE(hs, 3);  // pass 2
  // pass 3: iterative U-turn over the checkpoints idx_max .. idx_min (termination.py:86-104); the
  // rows of level i - 1 are requested before the reduction of level i
  bool turning = false;
  for (int i = idx_max; i >= idx_min && !turning; --i) {
    Row<VEC> N0[NI], N1[NI];
    if (i > idx_min) {
      if (LOOP && hot && i == idx_max) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      const float* r_ck = nt.ckpt_r + (c * nt.max_depth + i - 1) * nt.D;
      const float* rs_ck = nt.ckpt_rs + (c * nt.max_depth + i - 1) * nt.D;
#pragma unroll
      for (int k = 0; k < NI; ++k)
        if (FULL || ok[k]) {
          N0[k] = ldr<VEC>(r_ck + j0[k]);
          N1[k] = ldr<VEC>(rs_ck + j0[k]);
        }
    }
    double a_left = 0.0, a_right = 0.0;
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (FULL || ok[k]) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float rl = C0[k].v[e];
          const float ssum = (R.S[k].v[e] - C1[k].v[e]) + rl;
          const float rho = ssum - (R.P[k].v[e] + rl) * 0.5f;  // metrics.py:300
          a_left += (double)(R.M[k].v[e] * rl) * (double)rho;
          a_right += (double)(R.M[k].v[e] * R.P[k].v[e]) * (double)rho;
        }
      }
    a_left = wave_sum(a_left);
    a_right = wave_sum(a_right);
    turning = ((float)a_left <= 0.0f) || ((float)a_right <= 0.0f);
    if (i > idx_min) {
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        C0[k] = N0[k];
        C1[k] = N1[k];
      }
    }
  }
  const bool stop = sdiv || turning;
  if constexpr (LOOP) BJX_PROBE(hs, 4);  // pass 3
  if (!(stop || last)) {  // the subtree keeps integrating: opening half of leaf s + 1
    rec_set_f(w, RW_SW, Wn);
    rec_set_f(w, RW_SSLPA, Sn);
    if (take) {
      rec_set_f(w, RW_SLOGP, lp);
      rec_set_f(w, RW_SENERGY, e_new);
    }
    rec_set_i(w, RW_SUBN, s + 1);
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (FULL || ok[k]) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          R.P[k].v[e] = fmaf(h, R.G[k].v[e], R.P[k].v[e]);
          R.X[k].v[e] = fmaf(dd, R.M[k].v[e] * R.P[k].v[e], R.X[k].v[e]);
        }
        str<VEC>(fpp + j0[k], R.P[k]);
        str<VEC>(qn + j0[k], R.X[k]);
        str<VEC>(sm + j0[k], R.S[k]);  // the subtree's momentum sum is only stored while it keeps growing
      }
    if constexpr (LOOP) {
      hs->merged = false;
      BJX_PROBE(hs, 5);
    }
    return 0;
  }

  // ---- the subtree is complete: merge it (trajectory.py:680-727, proposal.py:146-176)
  if (!last) {  // stopped early: the merge rows were not requested above
    if (LOOP && hot) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (FULL || ok[k]) {
        MS[k] = ldr<VEC>(ms_src + j0[k]);
        OP[k] = ldr<VEC>(op + j0[k]);
      }
  }
  const float pw = rec_f(w, RW_PW), pslpa = rec_f(w, RW_PSLPA);
  bool take_m = false;
  float new_pw = pw;
  const Scalars3 scm = scalars3((double)(Wn - pw), pslpa, Sn, pw, Wn);
  const float new_pslpa = scm.lae1;
  if (!stop) {  // progressive_biased_sampling
    const Key kp{(uint32_t)rec_i(w, RW_KP), (uint32_t)rec_i(w, RW_KPB)};
    take_m = key_uniform(kp) < min1_nan(scm.e0);
    new_pw = scm.lae2;
  }
  // merged momentum sum + U-turn of the whole trajectory
  double a_left = 0.0, a_right = 0.0;
#pragma unroll
  for (int k = 0; k < NI; ++k)
    if (FULL || ok[k]) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float pl = dir > 0 ? OP[k].v[e] : R.P[k].v[e];
        const float pr = dir > 0 ? R.P[k].v[e] : OP[k].v[e];
        MS[k].v[e] = MS[k].v[e] + R.S[k].v[e];
        const float rho = MS[k].v[e] - (pr + pl) * 0.5f;
        a_left += (double)(R.M[k].v[e] * pl) * (double)rho;
        a_right += (double)(R.M[k].v[e] * pr) * (double)rho;
      }
      str<VEC>(nt.msum + base + j0[k], MS[k]);
      if (take_m) {
        // the subtree's proposal: this leaf's state when the leaf itself was taken, else rows an
        // earlier leaf of the subtree stored
        str<VEC>(nt.Pq + base + j0[k], take ? R.X[k] : ldr<VEC>(nt.Sq + base + j0[k]));
        str<VEC>(nt.Pg + base + j0[k], take ? R.G[k] : ldr<VEC>(nt.Sg + base + j0[k]));
      }
    }
  lazy &= ~LZ_M;
  if (take_m) lazy &= ~LZ_P;
  a_left = wave_sum(a_left);
  a_right = wave_sum(a_right);
  const bool turn = turning || ((float)a_left <= 0.0f) || ((float)a_right <= 0.0f);
  const bool grow = !sdiv && !turn && depth + 1 < nt.max_depth;
  const int n = rec_i(w, RW_NSTATES) + s + 1;
  rec_set_f(w, RW_PW, new_pw);
  rec_set_f(w, RW_PSLPA, new_pslpa);
  if (take_m) {
    rec_set_f(w, RW_PLOGP, take ? lp : rec_f(w, RW_SLOGP));
    rec_set_f(w, RW_PENERGY, take ? e_new : rec_f(w, RW_SENERGY));
  }
  rec_set_f(w, RW_ACC, exp_cr(new_pslpa) / (float)n);  // nuts.py:303-305
  rec_set_i(w, RW_NSTATES, n);
  rec_set_i(w, RW_DIV, sdiv ? 1 : 0);
  rec_set_i(w, RW_TURN, turn ? 1 : 0);
  rec_set_i(w, RW_DEPTH, depth + 1);
  if (!grow) {  // the transition is complete
    rec_set_i(w, RW_LAZY, lazy);
    if (lane == 0) ax.phase[c] = 3;
    return 1;
  }
  // ---- next doubling (trajectory.py:645-670)
  const Key ik{(uint32_t)rec_i(w, RW_IK), (uint32_t)rec_i(w, RW_IKB)};
  const int dir2 = begin_doubling_rec(w, ik, depth + 1);
  const float deps2 = (float)dir2 * eps;
  const float h2 = deps2 * int_kick(nt);
  const float dd2 = deps2 * int_drift(nt);
  if (dir2 == dir) {  // the end just reached keeps moving: its state is in registers
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (FULL || ok[k]) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          R.P[k].v[e] = fmaf(h2, R.G[k].v[e], R.P[k].v[e]);
          R.X[k].v[e] = fmaf(dd2, R.M[k].v[e] * R.P[k].v[e], R.X[k].v[e]);
        }
        str<VEC>(fpp + j0[k], R.P[k]);
        str<VEC>(qn + j0[k], R.X[k]);
      }
  } else {  // park this end in its arrays, continue from the other one
    float* eq = (dir > 0 ? nt.Rq : nt.Lq) + base;
    float* eg = (dir > 0 ? nt.Rg : nt.Lg) + base;
    float* ep = (dir > 0 ? nt.Rp : nt.Lp) + base;
    const bool z0 = (lazy & other_bit) != 0;
    const float* oq = (z0 ? nt.q0 : (dir2 > 0 ? nt.Rq : nt.Lq)) + base;
    const float* og = (z0 ? nt.g0 : (dir2 > 0 ? nt.Rg : nt.Lg)) + base;
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (FULL || ok[k]) {
        str<VEC>(eq + j0[k], R.X[k]);
        str<VEC>(eg + j0[k], R.G[k]);
        str<VEC>(ep + j0[k], R.P[k]);
        const Row<VEC> g2 = ldr<VEC>(og + j0[k]);
        Row<VEC> q2 = ldr<VEC>(oq + j0[k]);
        Row<VEC> p2 = OP[k];  // the other end's momentum was loaded for the merge
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          p2.v[e] = fmaf(h2, g2.v[e], p2.v[e]);
          q2.v[e] = fmaf(dd2, R.M[k].v[e] * p2.v[e], q2.v[e]);
        }
        str<VEC>(fpp + j0[k], p2);
        str<VEC>(qn + j0[k], q2);
        if constexpr (LOOP) {  // the rows of the next leaf, as it would load them
          R.P[k] = p2;
          R.X[k] = q2;
        }
      }
    lazy &= ~other_bit;
  }
  rec_set_i(w, RW_LAZY, lazy);
  if constexpr (LOOP) hs->merged = true;
  return 0;
}

// End of a transition (phase 3) and start of the next one (phase 0 / after phase 3): record, accept,
// adapt, momentum draw, lazy tree start, doubling 0, opening half of its first leaf.
// Returns true when the chain leaves with a new pending position in qf[b] (false: it has completed its
// last transition).
template <int NI>
__device__ __forceinline__ bool async_end2_chain(const bjx_nuts_t& nt, const bjx_nuts_async_t& ax, float* qf,
                                                 int64_t c, int64_t b, int phase, int& w) {
  constexpr int VEC = 4;
  const int lane = threadIdx.x & 63;
  int32_t t = ax.t[c];
  const int64_t base = c * nt.D;
  float* qrow = qf + b * nt.D;
  uint32_t j0[NI];  // 32-bit element offsets: rows are addressed as (uniform base) + (32-bit lane offset)
  bool ok[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    j0[k] = ((uint32_t)lane + 64u * k) * VEC;
    ok[k] = j0[k] < (uint32_t)nt.D;
  }
  Row<VEC> Q[NI], G[NI];
  float lp;
  if (phase == 3) {
    // transition t is complete: the proposal becomes the chain's state unless it still IS the state
    const bool same = (rec_i(w, RW_LAZY) & LZ_P) != 0;
    const int64_t row = (int64_t)t * nt.N + c;
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (ok[k]) {
        Q[k] = ldr<VEC>((same ? ax.q : nt.Pq) + base + j0[k]);
        G[k] = ldr<VEC>((same ? ax.g : nt.Pg) + base + j0[k]);
        if (!same) {
          str<VEC>(ax.q + base + j0[k], Q[k]);
          str<VEC>(ax.g + base + j0[k], G[k]);
        }
        if (ax.out_position) str<VEC>(ax.out_position + row * nt.D + j0[k], Q[k]);
      }
    lp = rec_f(w, RW_PLOGP);
    const float acc_rate = rec_f(w, RW_ACC);
    if (lane == 0) {
      ax.logp[c] = lp;
      if (ax.out_logdensity) ax.out_logdensity[row] = lp;
      if (ax.out_acceptance_rate) ax.out_acceptance_rate[row] = acc_rate;
      if (ax.out_energy) ax.out_energy[row] = rec_f(w, RW_PENERGY);
      if (ax.out_num_integration_steps) ax.out_num_integration_steps[row] = rec_i(w, RW_NSTATES);
      if (ax.out_num_trajectory_expansions) ax.out_num_trajectory_expansions[row] = rec_i(w, RW_DEPTH);
      if (ax.out_is_divergent) ax.out_is_divergent[row] = (uint8_t)(rec_i(w, RW_DIV) != 0);
      if (ax.out_is_turning) ax.out_is_turning[row] = (uint8_t)(rec_i(w, RW_TURN) != 0);
    }
    if (ax.adapt_tab) {
      if (lane == 0) FS(BJX_NUTS_F_ACC, c) = acc_rate;  // async_adapt_chain reads it from the slot table
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // ax.q: written above, read by the Welford update
      async_adapt_chain<VEC>(nt, ax, c, t);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // step size / metric: read again below
    }
    t += 1;
    if (lane == 0) ax.t[c] = t;
    if (t >= ax.n_steps) {
      if (lane == 0) {
        ax.phase[c] = 2;
        atomicAdd(ax.n_done, 1);
      }
      return false;
    }
  } else {
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (ok[k]) {
        Q[k] = ldr<VEC>(ax.q + base + j0[k]);
        G[k] = ldr<VEC>(ax.g + base + j0[k]);
      }
    lp = ax.logp[c];
  }
  // start transition t: momentum draw (hmc.py:299-302, metrics.py:260-270), tree of nuts.py:278-294
  const StepCtx cx = async_ctx(nt, ax, t);
  const Key kc = chain_key(cx.key, (uint64_t)(c + cx.off), cx.fold);
  const Key km = key_child(kc, 0);  // split(kc, 2)[0]
  const Key ik = key_child(kc, 1);  // split(kc, 2)[1]   (nuts.py:133)
  const float* im = nt.imm + c * nt.imm_stride;
  const float eps = chain_eps(nt, c);
  Row<VEC> M[NI], P[NI];
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < NI; ++k)
    if (ok[k]) {
      M[k] = ldr<VEC>(im + j0[k]);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float z = normal_from_bits(key_bits32(km, (uint64_t)(j0[k] + e)));
        const float ms = 1.0f / sqrtf(M[k].v[e]);
        P[k].v[e] = ms * z;
        acc += (double)(M[k].v[e] * P[k].v[e]) * (double)P[k].v[e];
      }
      str<VEC>(ax.p + base + j0[k], P[k]);
    }
  acc = wave_sum(acc);
  const float ke = 0.5f * (float)acc;
  const float H0 = -lp + ke;
  rec_set_f(w, RW_H0, H0);
  rec_set_f(w, RW_PLOGP, lp);
  rec_set_f(w, RW_PENERGY, H0);
  rec_set_f(w, RW_PW, 0.0f);
  rec_set_f(w, RW_PSLPA, -__builtin_inff());
  rec_set_f(w, RW_SW, 0.0f);
  rec_set_f(w, RW_SSLPA, -__builtin_inff());
  rec_set_f(w, RW_ACC, __builtin_nanf(""));
  rec_set_i(w, RW_NSTATES, 0);
  rec_set_i(w, RW_DIV, 0);
  rec_set_i(w, RW_TURN, 0);
  rec_set_i(w, RW_DEPTH, 0);
  rec_set_i(w, RW_IK, (int)ik.k0);
  rec_set_i(w, RW_IKB, (int)ik.k1);
  rec_set_f(w, RW_EPS, eps);
  const int dir = begin_doubling_rec(w, ik, 0);
  const float deps = (float)dir * eps;
  const float h = deps * int_kick(nt);
  const float dd = deps * int_drift(nt);
  rec_set_i(w, RW_STAGE, 0);
#pragma unroll
  for (int k = 0; k < NI; ++k)
    if (ok[k]) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        P[k].v[e] = fmaf(h, G[k].v[e], P[k].v[e]);
        Q[k].v[e] = fmaf(dd, M[k].v[e] * P[k].v[e], Q[k].v[e]);
      }
      str<VEC>(ax.front_p + base + j0[k], P[k]);
      str<VEC>(qrow + j0[k], Q[k]);
    }
  rec_set_i(w, RW_LAZY, (LZ_L | LZ_R | LZ_P | LZ_M) & ~(dir > 0 ? LZ_R : LZ_L));
  if (lane == 0) ax.phase[c] = 1;
  return true;
}

// Engine-resident log-density of the row this wave just wrote to qf[b] (bjx_nuts_async_t.target_kind):
// the position is re-read (same wave, after a fence: L1 / L2 resident) and (logp, grad) of the stand-alone
// target kernels written to logp_f[b] / gf[b] for the next tick.
template <int NI>
__device__ __forceinline__ void async_target_row(const bjx_nuts_t& nt, const bjx_nuts_async_t& ax,
                                                 const float* qf, float* logp_f, float* gf, int64_t b) {
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // qf[b]: written above by this wave
  const float* qrow = qf + b * nt.D;
  F4 x[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int64_t j = ((int64_t)lane + 64 * k) * 4;
    if (j < nt.D) x[k] = ld4(qrow + j);
  }
#ifdef BJX_RTC_USER_TARGET
  if (ax.target_kind == BJX_TARGET_USER) {  // user-written device target (csrc/bjx_traj_dev.h interface)
    typename BJX_RTC_USER_TARGET::template Ctx<NI> ctx;
    BJX_RTC_USER_TARGET::template init<NI>(ctx, nt.D, ax.target_vec);
    F4 g[NI];
    float lp = 0.0f;
    BJX_RTC_USER_TARGET::template eval<NI>(ctx, nt.D, ax.target_vec, x, true, g, lp);
    target_store<NI>(nt.D, g, lp, logp_f + b, gf + b * nt.D);
    return;
  }
#endif
  if (ax.target_kind == BJX_TARGET_NEAL_FUNNEL) funnel_row<NI>(nt.D, x, logp_f + b, gf + b * nt.D);
  else diag_gaussian_row<NI>(nt.D, x, ax.target_vec, logp_f + b, gf + b * nt.D);
}

// `k_ticks` ticks of one compact row in one launch (engine-resident target).  Only this wave touches the
// chain during the launch.  A tick whose inputs are not in registers (the first one, the one after a
// transition end) starts with a workgroup-scope fence and loads everything;
// after a leaf that leaves a new leaf in flight, the next tick's rows are the registers this one holds and
// its gradient / log-density come straight from the target's registers (the same values are still stored:
// memory is what the next launch, or the host, sees).
template <int NI, bool FULL>
__device__ __forceinline__ void async_multi_tick_row(const bjx_nuts_t& nt, const bjx_nuts_async_t& ax, float* qf,
                                                     float* logp_f, float* gf, int64_t b, int k_ticks) {
  constexpr int VEC = 4;
  const int lane = threadIdx.x & 63;
  const int chain = ax.rows ? ax.rows[b] : (int)b;
  const int64_t c = (int64_t)__builtin_amdgcn_readfirstlane(chain);
  const int64_t base = c * nt.D;
  int* recp = ax.rec + c * BJX_NUTS_REC_WORDS;
  const float* im = nt.imm + c * nt.imm_stride;
  HotState<NI> hs;
  hs.hot = false;
  hs.pk_valid = false;
  hs.merged = false;
#ifdef BJX_RTC_USER_TARGET
  typename BJX_RTC_USER_TARGET::template Ctx<NI> user_ctx;
  if (ax.target_kind == BJX_TARGET_USER) BJX_RTC_USER_TARGET::template init<NI>(user_ctx, nt.D, ax.target_vec);
#endif
#ifdef BJX_TICK_PROBE
  for (int k = 0; k < 12; ++k) hs.acc[k] = 0;
  hs.last = __builtin_readcyclecounter();
#endif
  LeafRows<NI> R;
  int w = 0, phase = 0;
  float lp = 0.0f;
  for (int it = 0; it < k_ticks; ++it) {
    if (!hs.hot) {
      if (it) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      phase = ax.phase[c];
      w = recp[lane & (BJX_NUTS_REC_WORDS - 1)];
      lp = logp_f[b];
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const uint32_t j = ((uint32_t)lane + 64u * k) * VEC;
        if (j < (uint32_t)nt.D) {
          R.G[k] = ldr<VEC>(gf + b * nt.D + j);
          R.M[k] = ldr<VEC>(im + j);
          R.P[k] = ldr<VEC>(ax.front_p + base + j);
          R.X[k] = ldr<VEC>(qf + b * nt.D + j);
          R.S[k] = ldr<VEC>(nt.Smsum + base + j);
        }
      }
      phase = __builtin_amdgcn_readfirstlane(phase);
      hs.pk_valid = false;
    }
    const int w_in = w;
    bool pending, in_regs = false;
    BJX_PROBE(&hs, 6);  // loop top (cold: fence + loads)
    if (phase == 1) {
      const int done = async_leaf2_chain<NI, true, FULL>(nt, ax, qf, lp, c, b, w, R, &hs);
      if (done) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        pending = async_end2_chain<NI>(nt, ax, qf, c, b, 3, w);
        BJX_PROBE(&hs, 7);  // merge path + transition end
      } else {
        pending = in_regs = true;
        if (hs.merged) BJX_PROBE(&hs, 8);  // merge path + next doubling
      }
    } else if (phase == 3 || phase == 0) {
      pending = async_end2_chain<NI>(nt, ax, qf, c, b, phase, w);
    } else {
      break;  // the chain has completed all its transitions
    }
    if (lane < BJX_NUTS_REC_WORDS && w != w_in) recp[lane] = w;
    if (!pending) break;
    if (in_regs) {  // the position this tick wrote to qf[b] is R.X
      F4 x[NI], g[NI];
#pragma unroll
      for (int k = 0; k < NI; ++k) x[k] = F4{R.X[k].v[0], R.X[k].v[1], R.X[k].v[2], R.X[k].v[3]};
#ifdef BJX_RTC_USER_TARGET
      if (ax.target_kind == BJX_TARGET_USER)
        BJX_RTC_USER_TARGET::template eval<NI>(user_ctx, nt.D, ax.target_vec, x, true, g, lp);
      else
#endif
      if (ax.target_kind == BJX_TARGET_NEAL_FUNNEL) funnel_eval<NI, FULL>(nt.D, x, g, lp);
      else diag_gaussian_eval<NI, FULL>(nt.D, x, ax.target_vec, g, lp);
      target_store<NI>(nt.D, g, lp, logp_f + b, gf + b * nt.D);
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        R.G[k].v[0] = g[k].x; R.G[k].v[1] = g[k].y; R.G[k].v[2] = g[k].z; R.G[k].v[3] = g[k].w;
      }
      hs.hot = true;
      phase = 1;
    } else {
      async_target_row<NI>(nt, ax, qf, logp_f, gf, b);
      hs.hot = false;
    }
    BJX_PROBE(&hs, 9);  // record + target stores
#ifdef BJX_TICK_PROBE
    hs.acc[10] += 1;
#endif
  }
#ifdef BJX_TICK_PROBE
  if (b == 0 && lane == 0)
    for (int k = 0; k < 12; ++k) atomicAdd(&bjx_tick_probe[k], hs.acc[k]);
#endif
}

// WAVES = occupancy hint (waves per SIMD): 4 caps the kernel at 128 VGPRs, 3 at 168.
// ONE WAVE PER WORKGROUP: the waves of a workgroup are placed together and a new workgroup needs
// all its wave slots at once, so with four chains per workgroup a CU slot group lives as long as
// the slowest of four leaves (a merge + direction change takes several times a plain leaf); these
// kernels use neither LDS nor barriers, so nothing is lost by launching 64-thread workgroups.
// Engine-resident target, bjx_nuts_async_t.ticks_per_launch > 1: one wave per row, that many ticks each.
// FULL: D == 256 NI, every lane holds a piece of every row (no per-piece guards: straight-line code).
template <int NI, int WAVES, bool FULL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES)))
k_nuts_async_multi(bjx_nuts_t nt, bjx_nuts_async_t ax, float* qf, float* logp_f, float* gf) {
  const int64_t n_rows = async_n_rows(ax);
  for (int64_t b = blockIdx.x; b < n_rows; b += gridDim.x)
    async_multi_tick_row<NI, FULL>(nt, ax, qf, logp_f, gf, b, ax.ticks_per_launch);
}

// ------------------------------------------------------------------------------------ free-running chains, v3
// (round 4) FOUR CHAINS PER WAVE for the busy phase of rows of at most 256 floats: one DPP row of 16 lanes
// per chain, up to four 16-byte pieces per lane.  The v2 leaf above spends one wave on one 1 KB row: four
// floats per lane of row arithmetic against ~600 wave-uniform "scalar" vector instructions (threefry blocks,
// fp64 exp / log1p of the sampling step, checkpoint index arithmetic, decisions) that all 64 lanes repeat --
// 673 vector instructions per row and leaf (SQ counters, profiles/r03), i.e. half of a launch is instruction
// issue.  Here a lane of the scalar chain serves one of FOUR chains (the four rows of a wave take the same
// instruction stream with their own operands; branches are uniform within a DPP row), so the scalar chain is
// paid once per four leaves, and a wave keeps four chains' rows in flight.
// Same data movement, same keys, same arithmetic, expression for expression, as async_leaf2_chain.  The
// reductions reproduce wave_sum's summation TREE for a v2 row (lane l of v2 = piece l / 16, lane l % 16 here):
// per piece a balanced adjacent-pair tree over the row's 16 lanes (the xor butterfly below builds the same
// tree as the row_shr scan; every node adds the same two operands, and IEEE addition commutes), then
// (r3 + r2) + (r1 + r0) as row_bcast:15 / row_bcast:31 combine them -- so every sum, hence every decision
// and every record, is bit-identical to the v2 kernels (tests/test_nuts_free_gpu.py).
// MODE 0 only (leaf work; transition ends go on the work list of k_nuts_async_end_list, unchanged).
#ifndef __HIPCC_RTC__
constexpr int kRecHot = 16;   // words 0 .. 15: loaded with the rows
constexpr int kRecCold = 12;  // words 16 .. 27: loaded by a leaf that merges its subtree
static_assert(RW_PW == kRecHot && RW_END == kRecHot + kRecCold, "record layout");

// GL = lanes per chain: 16 (one DPP row: four chains per wave) or 64 (the whole wave: one chain per wave,
// chain-uniform values are wave-uniform and live in SGPRs)
template <int GL, int N_>
__device__ __forceinline__ int row_bcast_i(int v) {  // lane N_ of every chain's lane group to the whole group
  if constexpr (GL == 64) return __builtin_amdgcn_readlane(v, N_);
  else return __builtin_amdgcn_update_dpp(0, v, 0x150 + N_, 0xf, 0xf, false);  // row_newbcast:N_
}
template <int GL, int N_>
__device__ __forceinline__ float row_bcast_f(float v) { return __int_as_float(row_bcast_i<GL, N_>(__float_as_int(v))); }
template <int GL>
__device__ __forceinline__ int chain_uniform(int v) {  // tells the compiler a value is the same in all lanes of a chain
  if constexpr (GL == 64) return __builtin_amdgcn_readfirstlane(v);
  else return v;
}

// sum over the 16 lanes of a DPP row, in every lane of the row (see the header comment for the tree)
__device__ __forceinline__ double row_sum16(double v) {
  v = dpp_add_f64<0xB1, 0xf>(v);   // quad_perm:[1,0,3,2]
  v = dpp_add_f64<0x4E, 0xf>(v);   // quad_perm:[2,3,0,1]
  v = dpp_add_f64<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add_f64<0x140, 0xf>(v);  // row_mirror
  return v;
}
// total of a chain's row from its per-piece lane partials: wave_sum's tree for the v2 layout
template <int GL, int NI>
__device__ __forceinline__ double chain_sum(const double (&a)[NI]) {
  if constexpr (GL == 64) return wave_sum(a[0]);  // one chain per wave: the partials of all pieces are in a[0] (ACC below)
  double r[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < NI; ++k) r[k] = row_sum16(a[k]);
  return (r[3] + r[2]) + (r[1] + r[0]);
}

// scalars3 for a DPP row: lanes 0, 1, 2 of the row take the three operands (same expressions)
template <int GL>
__device__ __forceinline__ Scalars3 scalars3_row(double arg0, float a1, float b1, float a2, float b2) {
  const int g = threadIdx.x & (GL - 1);
  const double xa = g == 1 ? (double)a1 : (double)a2;
  const double xb = g == 1 ? (double)b1 : (double)b2;
  const double t = xa - xb;
  const double e = exp(g == 0 ? arg0 : -fabs(t));
  const double l1p = log1p(e);
  double r;
  if (g == 0) r = 1.0 / (1.0 + e);
  else if (xa == xb) r = xa + 0.6931471805599453;
  else if (t > 0) r = xa + l1p;
  else if (t <= 0) r = xb + l1p;
  else r = t;  // NaN
  const float rf = (float)r, ef = (float)e;
  return Scalars3{row_bcast_f<GL, 0>(ef), row_bcast_f<GL, 0>(rf), row_bcast_f<GL, 1>(rf), row_bcast_f<GL, 2>(rf)};
}

#define RF(k_) __int_as_float(rw[k_])
#define RSETF(k_, v_) rw[k_] = __float_as_int(v_)
#define CF(k_) __int_as_float(rc[(k_) - kRecHot])
#define CI(k_) rc[(k_) - kRecHot]
#define CSETF(k_, v_) rc[(k_) - kRecHot] = __float_as_int(v_)

// One leaf of chain c (phase 1) by the 16 lanes of its DPP row.  Returns true when the transition is
// complete (phase 3 written).  Transcription of async_leaf2_chain<1> (LOOP = false) with three changes in
// what is held where (none in what is computed): the subtree's momentum sum after this leaf, S + P, is
// recomputed where it is used instead of kept beside S (the same single rounding each time); the words of
// the record only a merge touches are loaded, and stored, by a merging leaf; the two merge rows of a
// subtree's last leaf are requested into L2 up front (one word per cache line) instead of into registers.
template <int GL, int NI>
__device__ __forceinline__ bool async_leaf3_row(const bjx_nuts_t& nt, const bjx_nuts_async_t& ax, float* qf, float lp,
                                                int64_t c, int64_t b, int* recp, int (&rw)[kRecHot],
                                                LeafRows<NI>& R) {
  constexpr int VEC = 4;
  const int g = threadIdx.x & (GL - 1);
  const int32_t depth = rw[RW_DEPTH];
  const int32_t s = rw[RW_SUBN];
  const int dir = rw[RW_DIR];
  int lazy = rw[RW_LAZY];
  const float eps = RF(RW_EPS);
  const float deps = (float)dir * eps;
  const float h = deps * int_kick(nt);
  const float dd = deps * int_drift(nt);
  const int64_t base = c * nt.D;
  float* fpp = ax.front_p + base;
  float* qn = qf + b * nt.D;
  float* sm = nt.Smsum + base;
  const float H0 = RF(RW_H0), sw = RF(RW_SW), sslpa = RF(RW_SSLPA);
  const bool last = (s + 1) >= (1 << depth);
  const uint32_t us = (uint32_t)s;  // checkpoint indices (termination.py:75-84)
  const int idx_max = __popc(us >> 1);
  const int nsub = __popc((~us & (us + 1u)) - 1u);
  const int idx_min = idx_max - nsub + 1;
  const bool even = (us & 1u) == 0u;
  uint32_t j0[NI];
  bool ok[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    j0[k] = ((uint32_t)g + (uint32_t)GL * k) * VEC;
    ok[k] = j0[k] < (uint32_t)nt.D;
  }
  const int other_bit = dir > 0 ? LZ_L : LZ_R;
  const float* op = ((lazy & other_bit) ? nt.p0 : (dir > 0 ? nt.Lp : nt.Rp)) + base;
  const float* ms_src = ((lazy & LZ_M) ? nt.p0 : nt.msum) + base;
  // second round trip, issued now: first checkpoint level of an odd leaf (its momentum-SUM row is the S row
  // already in registers, see async_leaf2_chain); the merge rows and the cold record words of a last leaf
  // are pulled into L2 (one word per 64-byte line; volatile: the values are not used)
  Row<VEC> C0[NI];
  if (nsub > 0) {
    const float* r_ck = nt.ckpt_r + (c * nt.max_depth + idx_max) * nt.D;
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (ok[k]) C0[k] = ldr<VEC>(r_ck + j0[k]);
  }
  if (last && (uint32_t)g * 16u < (uint32_t)nt.D) {
    (void)*(const volatile float*)(ms_src + g * 16);
    (void)*(const volatile float*)(op + g * 16);
  }
  // four leaves' uniforms per draw (lanes g & 3 of the row), as in the v2 leaf
  if ((s & 3) == 0) {
    const Key kt{(uint32_t)rw[RW_KT], (uint32_t)rw[RW_KTB]};
    const uint32_t sl = (uint32_t)s + ((uint32_t)g & 3u);
    const float ul = key_uniform(key_child(kt, (uint64_t)sl));
    RSETF(RW_U0 + 0, (row_bcast_f<GL, 0>(ul)));
    RSETF(RW_U0 + 1, (row_bcast_f<GL, 1>(ul)));
    RSETF(RW_U0 + 2, (row_bcast_f<GL, 2>(ul)));
    RSETF(RW_U0 + 3, (row_bcast_f<GL, 3>(ul)));
  }
  const int sq = s & 3;
  const float u = sq == 0 ? RF(RW_U0) : (sq == 1 ? RF(RW_U0 + 1) : (sq == 2 ? RF(RW_U0 + 2) : RF(RW_U0 + 3)));

  // pass 1: closing half kick, kinetic energy
  // ACC: with one chain per wave (GL = 64) a lane adds the pieces of a row into ONE accumulator, piece after
  // piece -- the order of the round-2 kernels for rows of more than 256 floats; with 16 lanes per chain every
  // piece keeps its own partial (chain_sum combines them in wave_sum's order)
#define ACC(k_) (GL == 64 ? 0 : (k_))
  double a1[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) a1[k] = 0.0;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    if (ok[k]) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        R.P[k].v[e] = fmaf(h, R.G[k].v[e], R.P[k].v[e]);
        a1[ACC(k)] += (double)(R.M[k].v[e] * R.P[k].v[e]) * (double)R.P[k].v[e];
      }
    }
  }
  const float ke = 0.5f * (float)chain_sum<GL, NI>(a1);
  const float e_new = -lp + ke;  // hmc_energy (trajectory.py:745-748)
  float wgt = H0 - e_new;        // proposal.py:91-95
  if (wgt != wgt) wgt = -__builtin_inff();
  const float slpa_new = fminf(wgt, 0.0f);
  const bool sdiv = (-wgt) > nt.divergence_threshold;  // trajectory.py:325
  bool take;
  float Wn, Sn;
  if (s == 0) {
    take = true;
    Wn = wgt;
    Sn = slpa_new;
  } else {  // progressive uniform sampling (trajectory.py:329-339, proposal.py:118-143)
    const Scalars3 sc = scalars3_row<GL>(-(double)(wgt - sw), sw, wgt, sslpa, slpa_new);
    take = u < sc.r0;
    Wn = sc.lae1;
    Sn = sc.lae2;
  }
  // the subtree's momentum sum after this leaf (R.S keeps the sum BEFORE it: the U-turn check needs both)
#define S_AFTER(k_, e_) (s != 0 ? R.S[k_].v[e_] + R.P[k_].v[e_] : R.P[k_].v[e_])

  // pass 2: checkpoint store, subtree-proposal state copy
#pragma unroll
  for (int k = 0; k < NI; ++k)
    if (ok[k]) {
      if (even && !last && !sdiv) {
        str<VEC>(nt.ckpt_r + (c * nt.max_depth + idx_max) * nt.D + j0[k], R.P[k]);
        if ((us & 3u) == 0u) {
          Row<VEC> sn;
#pragma unroll
          for (int e = 0; e < VEC; ++e) sn.v[e] = S_AFTER(k, e);
          str<VEC>(nt.ckpt_rs + (c * nt.max_depth + idx_max) * nt.D + j0[k], sn);
        }
      }
      if (take) {
        str<VEC>(nt.Sq + base + j0[k], R.X[k]);
        str<VEC>(nt.Sg + base + j0[k], R.G[k]);
      }
    }

  // pass 3: iterative U-turn over the checkpoints idx_max .. idx_min (termination.py:86-104); the first
  // level's momentum-sum checkpoint is the S row in registers, deeper levels load both rows
  bool turning = false;
#define BJX_UTURN_LEVEL(C1_)                                                      \
  do {                                                                            \
    double al[NI], ar[NI];                                                        \
    _Pragma("unroll") for (int k = 0; k < NI; ++k) {                              \
      al[k] = 0.0;                                                                \
      ar[k] = 0.0;                                                                \
    }                                                                             \
    _Pragma("unroll") for (int k = 0; k < NI; ++k) {                              \
      if (ok[k]) {                                                                \
        _Pragma("unroll") for (int e = 0; e < VEC; ++e) {                         \
          const float rl = C0[k].v[e];                                            \
          const float ssum = (S_AFTER(k, e) - C1_[k].v[e]) + rl;                  \
          const float rho = ssum - (R.P[k].v[e] + rl) * 0.5f; /* metrics.py:300 */ \
          al[ACC(k)] += (double)(R.M[k].v[e] * rl) * (double)rho;                 \
          ar[ACC(k)] += (double)(R.M[k].v[e] * R.P[k].v[e]) * (double)rho;        \
        }                                                                         \
      }                                                                           \
    }                                                                             \
    const double a_left = chain_sum<GL, NI>(al);                                      \
    const double a_right = chain_sum<GL, NI>(ar);                                     \
    turning = ((float)a_left <= 0.0f) || ((float)a_right <= 0.0f);                \
  } while (0)
  if (nsub > 0) BJX_UTURN_LEVEL(R.S);
  for (int i = idx_max - 1; i >= idx_min && !turning; --i) {
    Row<VEC> C1[NI];
    const float* r_ck = nt.ckpt_r + (c * nt.max_depth + i) * nt.D;
    const float* rs_ck = nt.ckpt_rs + (c * nt.max_depth + i) * nt.D;
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (ok[k]) {
        C0[k] = ldr<VEC>(r_ck + j0[k]);
        C1[k] = ldr<VEC>(rs_ck + j0[k]);
      }
    BJX_UTURN_LEVEL(C1);
  }
#undef BJX_UTURN_LEVEL
  const bool stop = sdiv || turning;
  if (!(stop || last)) {  // the subtree keeps integrating: opening half of leaf s + 1
    RSETF(RW_SW, Wn);
    RSETF(RW_SSLPA, Sn);
    if (take) {
      RSETF(RW_SLOGP, lp);
      RSETF(RW_SENERGY, e_new);
    }
    rw[RW_SUBN] = s + 1;
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (ok[k]) {
        Row<VEC> sn;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          sn.v[e] = S_AFTER(k, e);
          R.P[k].v[e] = fmaf(h, R.G[k].v[e], R.P[k].v[e]);
          R.X[k].v[e] = fmaf(dd, R.M[k].v[e] * R.P[k].v[e], R.X[k].v[e]);
        }
        str<VEC>(fpp + j0[k], R.P[k]);
        str<VEC>(qn + j0[k], R.X[k]);
        str<VEC>(sm + j0[k], sn);  // the subtree's momentum sum is only stored while it keeps growing
      }
    return false;
  }

  // ---- the subtree is complete: merge it (trajectory.py:680-727, proposal.py:146-176)
  int rc[kRecCold];
#pragma unroll
  for (int k = 0; k < kRecCold / 4; ++k) {
    const int4 t = *reinterpret_cast<const int4*>(recp + kRecHot + 4 * k);
    rc[4 * k] = chain_uniform<GL>(t.x); rc[4 * k + 1] = chain_uniform<GL>(t.y);
    rc[4 * k + 2] = chain_uniform<GL>(t.z); rc[4 * k + 3] = chain_uniform<GL>(t.w);
  }
  Row<VEC> MS[NI], OP[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k)
    if (ok[k]) {
      MS[k] = ldr<VEC>(ms_src + j0[k]);
      OP[k] = ldr<VEC>(op + j0[k]);
    }
  const float pw = CF(RW_PW), pslpa = CF(RW_PSLPA);
  bool take_m = false;
  float new_pw = pw;
  const Scalars3 scm = scalars3_row<GL>((double)(Wn - pw), pslpa, Sn, pw, Wn);
  const float new_pslpa = scm.lae1;
  if (!stop) {  // progressive_biased_sampling
    const Key kp{(uint32_t)CI(RW_KP), (uint32_t)CI(RW_KPB)};
    take_m = key_uniform(kp) < min1_nan(scm.e0);
    new_pw = scm.lae2;
  }
  double al[NI], ar[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    al[k] = 0.0;
    ar[k] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    if (ok[k]) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float pl = dir > 0 ? OP[k].v[e] : R.P[k].v[e];
        const float pr = dir > 0 ? R.P[k].v[e] : OP[k].v[e];
        MS[k].v[e] = MS[k].v[e] + S_AFTER(k, e);
        const float rho = MS[k].v[e] - (pr + pl) * 0.5f;
        al[ACC(k)] += (double)(R.M[k].v[e] * pl) * (double)rho;
        ar[ACC(k)] += (double)(R.M[k].v[e] * pr) * (double)rho;
      }
      str<VEC>(nt.msum + base + j0[k], MS[k]);
      if (take_m) {
        // the subtree's proposal: this leaf's state when the leaf itself was taken, else rows an
        // earlier leaf of the subtree stored
        str<VEC>(nt.Pq + base + j0[k], take ? R.X[k] : ldr<VEC>(nt.Sq + base + j0[k]));
        str<VEC>(nt.Pg + base + j0[k], take ? R.G[k] : ldr<VEC>(nt.Sg + base + j0[k]));
      }
    }
  }
  lazy &= ~LZ_M;
  if (take_m) lazy &= ~LZ_P;
  const double a_left = chain_sum<GL, NI>(al);
  const double a_right = chain_sum<GL, NI>(ar);
  const bool turn = turning || ((float)a_left <= 0.0f) || ((float)a_right <= 0.0f);
  const bool grow = !sdiv && !turn && depth + 1 < nt.max_depth;
  const int n = CI(RW_NSTATES) + s + 1;
  CSETF(RW_PW, new_pw);
  CSETF(RW_PSLPA, new_pslpa);
  if (take_m) {
    CSETF(RW_PLOGP, take ? lp : RF(RW_SLOGP));
    CSETF(RW_PENERGY, take ? e_new : RF(RW_SENERGY));
  }
  CSETF(RW_ACC, exp_cr(new_pslpa) / (float)n);  // nuts.py:303-305
  CI(RW_NSTATES) = n;
  CI(RW_DIV) = sdiv ? 1 : 0;
  CI(RW_TURN) = turn ? 1 : 0;
  rw[RW_DEPTH] = depth + 1;
  bool done = false;
  if (!grow) {  // the transition is complete
    rw[RW_LAZY] = lazy;
    if (ax.keep_ends) {
      // NUTSInfo.trajectory_leftmost_state / rightmost_state (nuts.py:66-70): the end that was moving is in registers,
      // the other one is parked in its arrays -- or still the transition's initial state (lazy), copied now
      // because the chain state is about to be replaced by the accepted proposal
      float* mq = (dir > 0 ? nt.Rq : nt.Lq) + base;
      float* mg = (dir > 0 ? nt.Rg : nt.Lg) + base;
      float* mp = (dir > 0 ? nt.Rp : nt.Lp) + base;
      const bool z0 = (lazy & other_bit) != 0;
      float* oq = (dir > 0 ? nt.Lq : nt.Rq) + base;
      float* og = (dir > 0 ? nt.Lg : nt.Rg) + base;
      float* opw = (dir > 0 ? nt.Lp : nt.Rp) + base;
#pragma unroll
      for (int k = 0; k < NI; ++k)
        if (ok[k]) {
          str<VEC>(mq + j0[k], R.X[k]);
          str<VEC>(mg + j0[k], R.G[k]);
          str<VEC>(mp + j0[k], R.P[k]);
          if (z0) {
            str<VEC>(oq + j0[k], ldr<VEC>(nt.q0 + base + j0[k]));
            str<VEC>(og + j0[k], ldr<VEC>(nt.g0 + base + j0[k]));
            str<VEC>(opw + j0[k], OP[k]);  // (loaded from p0 above)
          }
        }
      if (g == 0) {
        recp[dir > 0 ? RW_RLOGP : RW_LLOGP] = __float_as_int(lp);
        if (z0) recp[dir > 0 ? RW_LLOGP : RW_RLOGP] = __float_as_int(ax.logp[c]);
      }
    }
    if (g == 0) ax.phase[c] = 3;
    done = true;
  } else {
    // ---- next doubling (trajectory.py:645-670): direction and keys (begin_doubling_rec)
    const Key ik{(uint32_t)CI(RW_IK), (uint32_t)CI(RW_IKB)};
    const Key subkey = key_child(ik, (uint64_t)(depth + 1));
    const Key ch = key_child(subkey, (uint64_t)(g < 3 ? g : 0));  // split(subkey, 3) in lanes 0 .. 2 of the row
    const Key kd{(uint32_t)row_bcast_i<GL, 0>((int)ch.k0), (uint32_t)row_bcast_i<GL, 0>((int)ch.k1)};
    const int dir2 = key_uniform(kd) < 0.5f ? 1 : -1;
    rw[RW_KT] = row_bcast_i<GL, 1>((int)ch.k0);
    rw[RW_KTB] = row_bcast_i<GL, 1>((int)ch.k1);
    CI(RW_KP) = row_bcast_i<GL, 2>((int)ch.k0);
    CI(RW_KPB) = row_bcast_i<GL, 2>((int)ch.k1);
    rw[RW_DIR] = dir2;
    rw[RW_SUBN] = 0;
    const float deps2 = (float)dir2 * eps;
    const float h2 = deps2 * int_kick(nt);
    const float dd2 = deps2 * int_drift(nt);
    if (dir2 == dir) {  // the end just reached keeps moving: its state is in registers
#pragma unroll
      for (int k = 0; k < NI; ++k)
        if (ok[k]) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            R.P[k].v[e] = fmaf(h2, R.G[k].v[e], R.P[k].v[e]);
            R.X[k].v[e] = fmaf(dd2, R.M[k].v[e] * R.P[k].v[e], R.X[k].v[e]);
          }
          str<VEC>(fpp + j0[k], R.P[k]);
          str<VEC>(qn + j0[k], R.X[k]);
        }
    } else {  // park this end in its arrays, continue from the other one
      float* eq = (dir > 0 ? nt.Rq : nt.Lq) + base;
      float* eg = (dir > 0 ? nt.Rg : nt.Lg) + base;
      float* ep = (dir > 0 ? nt.Rp : nt.Lp) + base;
      const bool z0 = (lazy & other_bit) != 0;
      if (ax.keep_ends && g == 0) recp[dir > 0 ? RW_RLOGP : RW_LLOGP] = __float_as_int(lp);
      const float* oq = (z0 ? nt.q0 : (dir2 > 0 ? nt.Rq : nt.Lq)) + base;
      const float* og = (z0 ? nt.g0 : (dir2 > 0 ? nt.Rg : nt.Lg)) + base;
#pragma unroll
      for (int k = 0; k < NI; ++k)
        if (ok[k]) {
          str<VEC>(eq + j0[k], R.X[k]);
          str<VEC>(eg + j0[k], R.G[k]);
          str<VEC>(ep + j0[k], R.P[k]);
          const Row<VEC> g2 = ldr<VEC>(og + j0[k]);
          Row<VEC> q2 = ldr<VEC>(oq + j0[k]);
          Row<VEC> p2 = OP[k];  // the other end's momentum was loaded for the merge
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            p2.v[e] = fmaf(h2, g2.v[e], p2.v[e]);
            q2.v[e] = fmaf(dd2, R.M[k].v[e] * p2.v[e], q2.v[e]);
          }
          str<VEC>(fpp + j0[k], p2);
          str<VEC>(qn + j0[k], q2);
        }
      lazy &= ~other_bit;
    }
    rw[RW_LAZY] = lazy;
  }
  if (g == 0) {
#pragma unroll
    for (int k = 0; k < kRecCold / 4; ++k)
      *reinterpret_cast<int4*>(recp + kRecHot + 4 * k) = make_int4(rc[4 * k], rc[4 * k + 1], rc[4 * k + 2], rc[4 * k + 3]);
  }
  return done;
}
#undef ACC
#undef S_AFTER
#undef RF
#undef RSETF
#undef CF
#undef CI
#undef CSETF

// THE free-running tick of the contract path (diagonal metric, 16-byte rows of at most 256 NI floats): one wave =
// one 64-thread workgroup per compact row.  A chain with a leaf in flight (phase 1) does its leaf work
// (async_leaf3_row; a middle stage of a multi-stage integrator is a kick + drift only); a chain whose transition
// ended in tick k (phase 3), or that has not started (phase 0), is finished and restarted by ITS wave of tick
// k + 1's launch (async_end2_chain: record, accept, adapt, momentum draw, tree start, first opening half) -- the
// "deferred transition end": no second kernel and no work list, an ending chain spends one extra tick per
// transition.  Rounds 2-4 also carried a four-chains-per-wave leaf (GL = 16), a work-list kernel for the ends and
// the v2 leaf: measured slower (NOTEBOOK.md section 15) and removed in round 5; the lane-group parameter GL of the
// device functions below is now always 64.
template <int NI, int WAVES>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES)))
k_nuts_async_tick3(bjx_nuts_t nt, bjx_nuts_async_t ax, float* qf, const float* __restrict__ logp_f,
                   const float* __restrict__ gf) {
  constexpr int VEC = 4;
  constexpr int GL = 64;
  const int64_t n_rows = async_n_rows(ax);
  const int g = threadIdx.x;
  const int64_t b = (int64_t)blockIdx.x;
  if (b >= n_rows) return;
  int64_t c = ax.rows ? (int64_t)ax.rows[b] : b;
  c = (int64_t)__builtin_amdgcn_readfirstlane((int)c);
  // first round trip: the phase, the record and every row of a leaf, all at once
  int phase = ax.phase[c];
  int* recp = ax.rec + c * BJX_NUTS_REC_WORDS;
  int rw[kRecHot];
#pragma unroll
  for (int k = 0; k < kRecHot / 4; ++k) {
    const int4 t = *reinterpret_cast<const int4*>(recp + 4 * k);
    rw[4 * k] = t.x; rw[4 * k + 1] = t.y; rw[4 * k + 2] = t.z; rw[4 * k + 3] = t.w;
  }
  int stage = 0;
  if (ax.int_stages > 1) stage = recp[RW_STAGE];
  LeafRows<NI> R;
  const int64_t base = c * nt.D;
  const float* im = nt.imm + c * nt.imm_stride;
  float lp = logp_f[b];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const uint32_t j = ((uint32_t)g + (uint32_t)GL * k) * VEC;
    if (j < (uint32_t)nt.D) {
      R.G[k] = ldr<VEC>(gf + b * nt.D + j);
      R.M[k] = ldr<VEC>(im + j);
      R.P[k] = ldr<VEC>(ax.front_p + base + j);
      R.X[k] = ldr<VEC>(qf + b * nt.D + j);
      R.S[k] = ldr<VEC>(nt.Smsum + base + j);
    }
  }
  // wave-uniform: into SGPRs (after every load has been issued)
  phase = __builtin_amdgcn_readfirstlane(phase);
  lp = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(lp)));
#pragma unroll
  for (int k = 0; k < kRecHot; ++k) rw[k] = __builtin_amdgcn_readfirstlane(rw[k]);
  stage = chain_uniform<GL>(stage);
  if (phase == 1 && ax.int_stages > 1 && stage < ax.int_stages - 1) {
    // a middle stage of a multi-stage integrator: kick b_i with the gradient just evaluated, drift a_i
    const float deps = (float)rw[RW_DIR] * __int_as_float(rw[RW_EPS]);
    const float hk = deps * ax.int_mid_kick[stage], dk = deps * ax.int_mid_drift[stage];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const uint32_t j = ((uint32_t)g + (uint32_t)GL * k) * VEC;
      if (j < (uint32_t)nt.D) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          R.P[k].v[e] = fmaf(hk, R.G[k].v[e], R.P[k].v[e]);
          R.X[k].v[e] = fmaf(dk, R.M[k].v[e] * R.P[k].v[e], R.X[k].v[e]);
        }
        str<VEC>(ax.front_p + base + j, R.P[k]);
        str<VEC>(qf + b * nt.D + j, R.X[k]);
      }
    }
    if (g == 0) recp[RW_STAGE] = stage + 1;
  } else if (phase == 1) {
    if (ax.int_stages > 1 && g == 0) recp[RW_STAGE] = 0;
    async_leaf3_row<GL, NI>(nt, ax, qf, lp, c, b, recp, rw, R);
    if (g == 0) {
#pragma unroll
      for (int k = 0; k < kRecHot / 4; ++k)
        *reinterpret_cast<int4*>(recp + 4 * k) = make_int4(rw[4 * k], rw[4 * k + 1], rw[4 * k + 2], rw[4 * k + 3]);
    }
  } else if (phase == 3 || phase == 0) {  // record, accept, adapt, momentum draw, tree start, first opening half
    int w = recp[threadIdx.x & (BJX_NUTS_REC_WORDS - 1)];
    const int w_in = w;
    async_end2_chain<NI>(nt, ax, qf, c, b, phase, w);
    if ((int)threadIdx.x < BJX_NUTS_REC_WORDS && w != w_in) recp[threadIdx.x] = w;
  }
}

// ------------------------------------------------------------------------------------ free-running chains: speculative tail
// (round 5; include/bjx_nuts.h "Speculative tail")  Stream A: k_nuts_spec_integrate, the serial part of a leaf
// (closing kick, next position along the key's direction schedule) + a ring push of the callable's outputs.
// Stream B: k_nuts_spec_book, the UNCHANGED tick arithmetic (async_leaf3_row / async_end2_chain) replayed over the
// ring on the bookkeeper's own replica of the pending position -- every decision and record is the one-stream tick's.
enum { SW_EP = 0, SW_DEPTH, SW_S, SW_DIRS, SW_EPS, SW_CNT, SW_STATE };  // iw: stream A's words of a row
enum { BWD_EP = 0, BWD_CNT, BWD_IK0, BWD_IK1, BWD_EPS };                // bw: stream B -> A
enum { TG_EP = 0, TG_DEPTH, TG_S, TG_LP, TG_X0 };                       // ring_tag: identity of a pushed leaf
enum { SPD_MISMATCH = 0, SPD_STALL, SPD_RESTART, SPD_STALE, SPD_TIMEOUT, SPD_ORDER };
enum { SPS_RUN = 0, SPS_WAIT = 1, SPS_DONE = 2 };  // integrator state: integrating / tree exhausted or end pending / chain finished
static_assert(SW_STATE < BJX_NUTS_SPEC_IW && BWD_EPS < BJX_NUTS_SPEC_IW && TG_X0 + 4 <= BJX_NUTS_SPEC_TAG, "spec layout");

__device__ __forceinline__ int ld_agent(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ long long ld64_agent(const long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st64_agent(long long* p, long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int64_t spec_n_rows(const bjx_nuts_spec_t& sp) {
  if (!sp.n_rows_dev) return sp.n_rows;
  const int64_t n = (int64_t)__builtin_amdgcn_readfirstlane(*sp.n_rows_dev);
  return n < sp.n_rows ? n : sp.n_rows;
}
// bit d = 1: doubling d goes forward (begin_doubling_rec's draw for every depth at once, one lane per depth)
__device__ __forceinline__ int spec_dirs(Key ik, int max_depth) {
  const int lane = threadIdx.x & 63;
  const bool in = lane < max_depth;
  const Key kd = key_child(key_child(ik, (uint64_t)(in ? lane : 0)), 0);
  const bool fwd = key_uniform(kd) < 0.5f;
  return (int)(uint32_t)__ballot(in && fwd);
}

template <int NI>
__device__ __forceinline__ void spec_integrate_row(const bjx_nuts_t& nt, const bjx_nuts_async_t& ax,
                                                   const bjx_nuts_spec_t& sp, const float* __restrict__ logp_f,
                                                   const float* __restrict__ gf, int64_t b) {
  constexpr int VEC = 4;
  const int g = threadIdx.x;
  const int64_t c = (int64_t)__builtin_amdgcn_readfirstlane(sp.rows[b]);
  int* iwp = sp.iw + b * BJX_NUTS_SPEC_IW;
  int* bwp = sp.bw + b * BJX_NUTS_SPEC_IW;
  // one round trip: the row's words, the bookkeeper's epoch / consumed count, the callable's outputs, the rows
  const int4 i0 = *reinterpret_cast<const int4*>(iwp);
  const int4 i1 = *reinterpret_cast<const int4*>(iwp + 4);
  // (agent-scope loads; plain loads -- a launch starts with an acquire, so they would be at most one launch stale --
  // measured 2.5 % slower over the tail of the C3 T = 400 run)
  int ep_b = ld_agent(bwp + BWD_EP);
  int cnt_b = ld_agent(bwp + BWD_CNT);
  float lp = logp_f[b];
  const int64_t base = c * nt.D, rbase = b * nt.D;
  const float* im = nt.imm + c * nt.imm_stride;
  uint32_t j0[NI];
  bool ok[NI];
  Row<VEC> G[NI], X[NI], P[NI], M[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    j0[k] = ((uint32_t)g + 64u * k) * VEC;
    ok[k] = j0[k] < (uint32_t)nt.D;
    if (ok[k]) {
      G[k] = ldr<VEC>(gf + rbase + j0[k]);
      X[k] = ldr<VEC>(sp.qf + rbase + j0[k]);
      P[k] = ldr<VEC>(sp.fp + rbase + j0[k]);
      M[k] = ldr<VEC>(im + j0[k]);
    }
  }
  const int ep = __builtin_amdgcn_readfirstlane(i0.x);
  int depth = __builtin_amdgcn_readfirstlane(i0.y);
  int s = __builtin_amdgcn_readfirstlane(i0.z);
  const int dirs = __builtin_amdgcn_readfirstlane(i0.w);
  const float eps = __int_as_float(__builtin_amdgcn_readfirstlane(i1.x));
  int cnt = __builtin_amdgcn_readfirstlane(i1.y);
  int state = __builtin_amdgcn_readfirstlane(i1.z);
  ep_b = __builtin_amdgcn_readfirstlane(ep_b);
  cnt_b = __builtin_amdgcn_readfirstlane(cnt_b);
  lp = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(lp)));
  if (state == SPS_DONE) return;
  // every record pushed by EARLIER launches is complete in memory (those launches have ended): announce them
  if (g == 0) st_agent(sp.avail + b, cnt);
  if (ep_b == -1) {  // the chain has completed its last transition
    if (g == 0) iwp[SW_STATE] = SPS_DONE;
    return;
  }
  if (ep_b != ep) {
    // ---- the bookkeeper has started a new transition: restart from its first pending position
    __threadfence();  // acquire: what stream B wrote before it published the epoch
    const Key ik{(uint32_t)ld_agent(bwp + BWD_IK0), (uint32_t)ld_agent(bwp + BWD_IK1)};
    const int eps_bits = ld_agent(bwp + BWD_EPS);
    const int dirs2 = spec_dirs(ik, nt.max_depth);
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (ok[k]) {
        const Row<VEC> q0 = ldr<VEC>(ax.q + base + j0[k]);
        const Row<VEC> p0 = ldr<VEC>(ax.p + base + j0[k]);
        const Row<VEC> g0 = ldr<VEC>(ax.g + base + j0[k]);
        const Row<VEC> x1 = ldr<VEC>(sp.qf_book + rbase + j0[k]);
        const Row<VEC> p1 = ldr<VEC>(ax.front_p + base + j0[k]);
        str<VEC>(sp.eLq + rbase + j0[k], q0); str<VEC>(sp.eLp + rbase + j0[k], p0); str<VEC>(sp.eLg + rbase + j0[k], g0);
        str<VEC>(sp.eRq + rbase + j0[k], q0); str<VEC>(sp.eRp + rbase + j0[k], p0); str<VEC>(sp.eRg + rbase + j0[k], g0);
        str<VEC>(sp.qf + rbase + j0[k], x1);
        str<VEC>(sp.fp + rbase + j0[k], p1);
      }
    if (g == 0) {
      *reinterpret_cast<int4*>(iwp) = make_int4(ep_b, 0, 0, dirs2);
      *reinterpret_cast<int4*>(iwp + 4) = make_int4(eps_bits, cnt, SPS_RUN, 0);
      // acknowledge: the first record of the new epoch will be number cnt (everything before it is stale)
      st64_agent(reinterpret_cast<long long*>(sp.ack) + b, ((long long)ep_b << 32) | (long long)(uint32_t)cnt);
      atomicAdd(sp.dbg + SPD_RESTART, 1);
    }
    return;
  }
  if (state != SPS_RUN) return;  // tree exhausted (max_depth doublings speculated) or a transition end pending
  // Do not run further ahead of the bookkeeper than `lead` records (at most a ring): everything pushed past the
  // end of a transition is wasted, and the bookkeeper -- a full leaf of dependent arithmetic per record -- is not
  // much faster than this stream.  A waiting row pushes nothing; the callable re-evaluates the same qf.
  const int lead = sp.lead > 0 && sp.lead < sp.ring - 1 ? sp.lead : sp.ring - 1;
  if (cnt - cnt_b >= lead) {
    if (g == 0) atomicAdd(sp.dbg + SPD_STALL, 1);
    return;
  }
  const int dir = ((dirs >> depth) & 1) ? 1 : -1;
  const float deps = (float)dir * eps;
  const float h = deps * int_kick(nt);
  const float dd = deps * int_drift(nt);
  // ---- push (logp, gradient) of the leaf whose position the callable just evaluated
  const int slot = cnt & (sp.ring - 1);
  float* rg = sp.ring_g + ((int64_t)b * sp.ring + slot) * nt.D;
#pragma unroll
  for (int k = 0; k < NI; ++k)
    if (ok[k]) str<VEC>(rg + j0[k], G[k]);
  if (g == 0) {
    int* tp = sp.ring_tag + ((int64_t)b * sp.ring + slot) * BJX_NUTS_SPEC_TAG;
    *reinterpret_cast<int4*>(tp) = make_int4(ep, depth, s, __float_as_int(lp));
    *reinterpret_cast<int4*>(tp + 4) = make_int4(__float_as_int(X[0].v[0]), __float_as_int(X[0].v[1]),
                                                 __float_as_int(X[0].v[2]), __float_as_int(X[0].v[3]));
  }
  cnt += 1;
  // ---- closing kick of this leaf, then the opening half of the next one (async_leaf3_row's expressions)
#pragma unroll
  for (int k = 0; k < NI; ++k)
    if (ok[k]) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) P[k].v[e] = fmaf(h, G[k].v[e], P[k].v[e]);
    }
  const bool last = (s + 1) >= (1 << depth);
  if (!last) {
    s += 1;
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (ok[k]) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          P[k].v[e] = fmaf(h, G[k].v[e], P[k].v[e]);
          X[k].v[e] = fmaf(dd, M[k].v[e] * P[k].v[e], X[k].v[e]);
        }
        str<VEC>(sp.fp + rbase + j0[k], P[k]);
        str<VEC>(sp.qf + rbase + j0[k], X[k]);
      }
  } else if (depth + 1 >= nt.max_depth) {
    state = SPS_WAIT;  // nothing left to speculate: this transition ends here at the latest
  } else {
    const int dir2 = ((dirs >> (depth + 1)) & 1) ? 1 : -1;
    const float deps2 = (float)dir2 * eps;
    const float h2 = deps2 * int_kick(nt);
    const float dd2 = deps2 * int_drift(nt);
    if (dir2 == dir) {  // the end just reached keeps moving
#pragma unroll
      for (int k = 0; k < NI; ++k)
        if (ok[k]) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            P[k].v[e] = fmaf(h2, G[k].v[e], P[k].v[e]);
            X[k].v[e] = fmaf(dd2, M[k].v[e] * P[k].v[e], X[k].v[e]);
          }
          str<VEC>(sp.fp + rbase + j0[k], P[k]);
          str<VEC>(sp.qf + rbase + j0[k], X[k]);
        }
    } else {  // park this end, continue from the other one
      float* eq = (dir > 0 ? sp.eRq : sp.eLq) + rbase;
      float* eg = (dir > 0 ? sp.eRg : sp.eLg) + rbase;
      float* epp = (dir > 0 ? sp.eRp : sp.eLp) + rbase;
      const float* oq = (dir2 > 0 ? sp.eRq : sp.eLq) + rbase;
      const float* og = (dir2 > 0 ? sp.eRg : sp.eLg) + rbase;
      const float* op = (dir2 > 0 ? sp.eRp : sp.eLp) + rbase;
#pragma unroll
      for (int k = 0; k < NI; ++k)
        if (ok[k]) {
          Row<VEC> q2 = ldr<VEC>(oq + j0[k]);
          const Row<VEC> g2 = ldr<VEC>(og + j0[k]);
          Row<VEC> p2 = ldr<VEC>(op + j0[k]);
          str<VEC>(eq + j0[k], X[k]);
          str<VEC>(eg + j0[k], G[k]);
          str<VEC>(epp + j0[k], P[k]);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            p2.v[e] = fmaf(h2, g2.v[e], p2.v[e]);
            q2.v[e] = fmaf(dd2, M[k].v[e] * p2.v[e], q2.v[e]);
          }
          str<VEC>(sp.fp + rbase + j0[k], p2);
          str<VEC>(sp.qf + rbase + j0[k], q2);
        }
    }
    depth += 1;
    s = 0;
  }
  if (g == 0) {
    *reinterpret_cast<int4*>(iwp) = make_int4(ep, depth, s, dirs);
    *reinterpret_cast<int4*>(iwp + 4) = make_int4(__float_as_int(eps), cnt, state, 0);
  }
}

template <int NI>
__global__ void __launch_bounds__(64)
k_nuts_spec_integrate(bjx_nuts_t nt, bjx_nuts_async_t ax, bjx_nuts_spec_t sp, const float* __restrict__ logp_f,
                      const float* __restrict__ gf, int bump) {
  const int64_t b = (int64_t)blockIdx.x;
  if (b < spec_n_rows(sp)) spec_integrate_row<NI>(nt, ax, sp, logp_f, gf, b);
  if (bump && b == 0 && threadIdx.x == 0) atomicAdd(sp.a_seq, 1);  // the last launch of a sequence of stream A
}

// Stream B: one wave per row, alive for one whole sequence of stream A (`target` = the value *a_seq reaches when that
// sequence has completed).  A row's wave: transition end pending -> async_end2_chain, publish the new epoch; leaf in
// flight -> wait for its record, then the tick kernel's leaf (k_nuts_async_tick3's phase-1 branch, cold path).
// Reading the ring.  A record is announced one integrate launch after the launch that wrote it, i.e. once that launch
// has COMPLETED and its stores are performed at agent scope; the wave only requests the record after the
// announcement has arrived (control dependence), and it reads the ring with agent-scope (sc1) loads, which are
// serviced at the coherence point instead of a possibly stale line of this XCD's L2 (a slot's previous lap).  So the
// per-record path needs no cache invalidation: the wave's own state (record words, momentum, pending position,
// momentum sums, checkpoints) stays in its L1 / L2.  (First form of this kernel: an agent-scope fence per record --
// L2 write-back + invalidate -- and three dependent round trips: 5.3 us per record against stream A's 6.5 us per
// leaf; the bookkeeper fell a ring behind and a quarter of the pushed leaves were wasted.)
template <int VEC>
__device__ __forceinline__ Row<VEC> ldr_agent(const float* p) {
  Row<VEC> r;
#pragma unroll
  for (int e = 0; e < VEC; ++e) r.v[e] = __int_as_float(ld_agent(reinterpret_cast<const int*>(p) + e));
  return r;
}

template <int NI>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2)))  // a wave per row, rows << CUs: registers are free
k_nuts_spec_book(bjx_nuts_t nt, bjx_nuts_async_t ax, bjx_nuts_spec_t sp, int target, long long timeout_ticks) {
  constexpr int VEC = 4;
  constexpr int GL = 64;
  const int g = threadIdx.x;
  const int64_t b = (int64_t)blockIdx.x;
  if (b >= spec_n_rows(sp)) return;
  const int64_t c = (int64_t)__builtin_amdgcn_readfirstlane(sp.rows[b]);
  int* recp = ax.rec + c * BJX_NUTS_REC_WORDS;
  int* bwp = sp.bw + b * BJX_NUTS_SPEC_IW;
  int my_ep = __builtin_amdgcn_readfirstlane(bwp[BWD_EP]);
  int cnt = __builtin_amdgcn_readfirstlane(bwp[BWD_CNT]);
  if (my_ep == -1) return;
  const int64_t base = c * nt.D, rbase = b * nt.D;
  const float* im = nt.imm + c * nt.imm_stride;
  const long long t0 = wall_clock64();
  long long busy = 0;
  int n_rec = 0, n_stale = 0, n_order = 0;
  int phase = __builtin_amdgcn_readfirstlane(ax.phase[c]);
  int n_avail = cnt;
  // has the integrator acknowledged this epoch?  (its acknowledgement names the first record of the epoch, so the
  // leaves it speculated past the end of the previous transition are skipped by count, not read one by one)
  const long long* ackp = reinterpret_cast<const long long*>(sp.ack) + b;
  bool need_ack = (int)(__builtin_amdgcn_readfirstlane((int)(ld64_agent(ackp) >> 32))) != my_ep;
  // the NEXT record's identity and gradient, requested while this one is worked on (an agent-scope load is ~2 us)
  int pf_cnt = -1;
  int pf_tg[BJX_NUTS_SPEC_TAG];
  Row<VEC> pf_G[NI];
  for (int it = 0;; ++it) {
    if (phase == 2) {
      if (g == 0) st_agent(bwp + BWD_EP, -1);
      break;
    }
    if (phase == 3 || phase == 0) {  // record, accept, momentum draw, tree start, first opening half -> qf_book / front_p
      if (it) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // this wave's own stores of the iteration before
      int w = recp[g & (BJX_NUTS_REC_WORDS - 1)];
      const int w_in = w;
      const bool pending = async_end2_chain<NI>(nt, ax, sp.qf_book, c, b, phase, w);
      if (g < BJX_NUTS_REC_WORDS && w != w_in) recp[g] = w;
      if (!pending) {
        if (g == 0) st_agent(bwp + BWD_EP, -1);
        break;
      }
      my_ep += 1;
      const int ik0 = rec_i(w, RW_IK), ik1 = rec_i(w, RW_IKB), epsb = rec_i(w, RW_EPS);
      if (g == 0) {
        st_agent(bwp + BWD_IK0, ik0);
        st_agent(bwp + BWD_IK1, ik1);
        st_agent(bwp + BWD_EPS, epsb);
      }
      __threadfence();  // release: chain state, first pending position / momentum, key and step size before the epoch
      if (g == 0) st_agent(bwp + BWD_EP, my_ep);
      phase = 1;
      need_ack = true;
      continue;
    }
    // ---- a leaf is in flight (phase 1): wait for a record (the announcement is only re-read when the records
    // known so far are used up: an agent-scope load is a trip to the coherence point, ~2 us)
    bool give_up = false;
    if (need_ack) {
      long long a = ld64_agent(ackp);
      while ((int)(a >> 32) != my_ep) {
        if (ld_agent(sp.a_seq) - target >= 0 || wall_clock64() - t0 > timeout_ticks) { give_up = true; break; }
        __builtin_amdgcn_s_sleep(8);
        a = ld64_agent(ackp);
      }
      if (give_up) break;
      const int first = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffll));
      n_stale += first - cnt;
      cnt = first;
      if (g == 0) st_agent(bwp + BWD_CNT, cnt);
      need_ack = false;
    }
    if (n_avail - cnt <= 0) n_avail = __builtin_amdgcn_readfirstlane(ld_agent(sp.avail + b));
    while (n_avail - cnt <= 0) {
      if (ld_agent(sp.a_seq) - target >= 0) { give_up = true; break; }  // stream A's sequence is over: the next launch carries on
      if (wall_clock64() - t0 > timeout_ticks) {
        if (g == 0) atomicAdd(sp.dbg + SPD_TIMEOUT, 1);
        give_up = true;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
      n_avail = __builtin_amdgcn_readfirstlane(ld_agent(sp.avail + b));
    }
    if (give_up) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler ordering only: no ring load before the announcement
    const long long tb = wall_clock64();
    if (it) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // this wave's own stores of the iteration before
    // one round trip: the record's identity and gradient (agent-scope loads) + this chain's own state
    const int slot = cnt & (sp.ring - 1);
    const int* tp = sp.ring_tag + ((int64_t)b * sp.ring + slot) * BJX_NUTS_SPEC_TAG;
    const float* rg = sp.ring_g + ((int64_t)b * sp.ring + slot) * nt.D;
    const bool pf_hit = pf_cnt == cnt;
    int tg[BJX_NUTS_SPEC_TAG];
#pragma unroll
    for (int k = 0; k < BJX_NUTS_SPEC_TAG; ++k) tg[k] = pf_hit ? pf_tg[k] : ld_agent(tp + k);
    int rw[kRecHot];
#pragma unroll
    for (int k = 0; k < kRecHot / 4; ++k) {
      const int4 t = *reinterpret_cast<const int4*>(recp + 4 * k);
      rw[4 * k] = t.x; rw[4 * k + 1] = t.y; rw[4 * k + 2] = t.z; rw[4 * k + 3] = t.w;
    }
    LeafRows<NI> R;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const uint32_t j = ((uint32_t)g + (uint32_t)GL * k) * VEC;
      if (j < (uint32_t)nt.D) {
        R.G[k] = pf_hit ? pf_G[k] : ldr_agent<VEC>(rg + j);
        R.M[k] = ldr<VEC>(im + j);
        R.P[k] = ldr<VEC>(ax.front_p + base + j);
        R.X[k] = ldr<VEC>(sp.qf_book + rbase + j);
        R.S[k] = ldr<VEC>(nt.Smsum + base + j);
      }
    }
#pragma unroll
    for (int k = 0; k < kRecHot; ++k) rw[k] = __builtin_amdgcn_readfirstlane(rw[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) tg[k] = __builtin_amdgcn_readfirstlane(tg[k]);
    if (!(tg[TG_EP] == my_ep && tg[TG_DEPTH] == rw[RW_DEPTH] && tg[TG_S] == rw[RW_SUBN])) {
      // not the next leaf of this transition: a leaf speculated past the end of an earlier one.  Skip those
      // (identity words only) up to the first record of this epoch or the end of what is announced.
      if (tg[TG_EP] == my_ep) ++n_order; else ++n_stale;
      ++cnt;
      while (cnt < n_avail) {
        const int* tq = sp.ring_tag + ((int64_t)b * sp.ring + (cnt & (sp.ring - 1))) * BJX_NUTS_SPEC_TAG;
        const int e = __builtin_amdgcn_readfirstlane(ld_agent(tq + TG_EP));
        if (e == my_ep) break;
        ++n_stale;
        ++cnt;
      }
      if (g == 0) st_agent(bwp + BWD_CNT, cnt);
      busy += wall_clock64() - tb;
      continue;
    }
    {  // the integrator evaluated the gradient at ITS position: it must be this replica's, bit for bit
      const bool bad = g == 0 && (tg[TG_X0] != __float_as_int(R.X[0].v[0]) || tg[TG_X0 + 1] != __float_as_int(R.X[0].v[1]) ||
                                  tg[TG_X0 + 2] != __float_as_int(R.X[0].v[2]) || tg[TG_X0 + 3] != __float_as_int(R.X[0].v[3]));
      if (bad) atomicAdd(sp.dbg + SPD_MISMATCH, 1);
    }
    const float lp = __int_as_float(tg[TG_LP]);
    const bool done = async_leaf3_row<GL, NI>(nt, ax, sp.qf_book, lp, c, b, recp, rw, R);
    if (g == 0) {
#pragma unroll
      for (int k = 0; k < kRecHot / 4; ++k)
        *reinterpret_cast<int4*>(recp + 4 * k) = make_int4(rw[4 * k], rw[4 * k + 1], rw[4 * k + 2], rw[4 * k + 3]);
    }
    if (done) phase = 3;  // (async_leaf3_row has written ax.phase[c] = 3 as well)
    cnt += 1;
    if (!done && n_avail - cnt > 0) {  // request the next record now: it arrives behind this leaf's stores and the fence
      const int slot2 = cnt & (sp.ring - 1);
      const int* tp2 = sp.ring_tag + ((int64_t)b * sp.ring + slot2) * BJX_NUTS_SPEC_TAG;
      const float* rg2 = sp.ring_g + ((int64_t)b * sp.ring + slot2) * nt.D;
#pragma unroll
      for (int k = 0; k < BJX_NUTS_SPEC_TAG; ++k) pf_tg[k] = ld_agent(tp2 + k);
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const uint32_t j = ((uint32_t)g + (uint32_t)GL * k) * VEC;
        if (j < (uint32_t)nt.D) pf_G[k] = ldr_agent<VEC>(rg2 + j);
      }
      pf_cnt = cnt;
    }
    if (g == 0) st_agent(bwp + BWD_CNT, cnt);  // (the record's loads have returned: the leaf used them)
    busy += wall_clock64() - tb;
    n_rec += 1;
  }
  if (g == 0) {
    st_agent(bwp + BWD_CNT, cnt);
    if (n_stale) atomicAdd(sp.dbg + SPD_STALE, n_stale);
    if (n_order) atomicAdd(sp.dbg + SPD_ORDER, n_order);
    if (n_rec) {  // statistics: time spent on records (100 MHz ticks) and records consumed
      atomicAdd(sp.dbg + 6, (int)busy);
      atomicAdd(sp.dbg + 7, n_rec);
    }
  }
}


// Do two streams really run concurrently?  (Streams share a few hardware queues; two streams on one queue execute
// in order, and a bookkeeper queued in front of its sequence would then wait for its time-out.)  The wait kernel,
// launched FIRST, spins until the set kernel on the other stream has run, or gives up: flag2[1] = 1 / 2.
__global__ void k_stream_probe_wait(int* flag2, long long timeout_ticks) {
  const long long t0 = wall_clock64();
  int seen = 0;
  while (!(seen = ld_agent(flag2)) && wall_clock64() - t0 < timeout_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) flag2[1] = seen ? 1 : 2;
}
__global__ void k_stream_probe_set(int* flag2) {
  if (threadIdx.x == 0) st_agent(flag2, 1);
}

// Hand-over from the one-stream tail: stream A's words, ends and momentum from the chain's record and arrays.
template <int NI>
__global__ void __launch_bounds__(64)
k_nuts_spec_enter(bjx_nuts_t nt, bjx_nuts_async_t ax, bjx_nuts_spec_t sp) {
  constexpr int VEC = 4;
  const int g = threadIdx.x;
  const int64_t b = (int64_t)blockIdx.x;
  if (b >= spec_n_rows(sp)) return;
  const int64_t c = (int64_t)__builtin_amdgcn_readfirstlane(sp.rows[b]);
  const int phase = __builtin_amdgcn_readfirstlane(ax.phase[c]);
  const int w = ax.rec[c * BJX_NUTS_REC_WORDS + (g & (BJX_NUTS_REC_WORDS - 1))];
  const int depth = rec_i(w, RW_DEPTH), s = rec_i(w, RW_SUBN), lazy = rec_i(w, RW_LAZY), dir = rec_i(w, RW_DIR);
  const int eps_bits = rec_i(w, RW_EPS), ik0 = rec_i(w, RW_IK), ik1 = rec_i(w, RW_IKB);
  const int dirs = spec_dirs(Key{(uint32_t)ik0, (uint32_t)ik1}, nt.max_depth);
  const int64_t base = c * nt.D, rbase = b * nt.D;
  const bool zl = (lazy & LZ_L) != 0, zr = (lazy & LZ_R) != 0;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const uint32_t j = ((uint32_t)g + 64u * k) * VEC;
    if (j < (uint32_t)nt.D) {
      str<VEC>(sp.qf_book + rbase + j, ldr<VEC>(sp.qf + rbase + j));
      str<VEC>(sp.fp + rbase + j, ldr<VEC>(ax.front_p + base + j));
      str<VEC>(sp.eLq + rbase + j, ldr<VEC>((zl ? nt.q0 : nt.Lq) + base + j));
      str<VEC>(sp.eLp + rbase + j, ldr<VEC>((zl ? nt.p0 : nt.Lp) + base + j));
      str<VEC>(sp.eLg + rbase + j, ldr<VEC>((zl ? nt.g0 : nt.Lg) + base + j));
      str<VEC>(sp.eRq + rbase + j, ldr<VEC>((zr ? nt.q0 : nt.Rq) + base + j));
      str<VEC>(sp.eRp + rbase + j, ldr<VEC>((zr ? nt.p0 : nt.Rp) + base + j));
      str<VEC>(sp.eRg + rbase + j, ldr<VEC>((zr ? nt.g0 : nt.Rg) + base + j));
    }
  }
  if (g == 0) {
    const bool run = phase == 1;
    int* iwp = sp.iw + b * BJX_NUTS_SPEC_IW;
    int* bwp = sp.bw + b * BJX_NUTS_SPEC_IW;
    *reinterpret_cast<int4*>(iwp) = make_int4(0, run ? depth : 0, run ? s : 0, dirs);
    *reinterpret_cast<int4*>(iwp + 4) = make_int4(eps_bits, 0, run ? SPS_RUN : SPS_WAIT, 0);
    *reinterpret_cast<int4*>(bwp) = make_int4(phase == 2 ? -1 : 0, 0, ik0, ik1);
    *reinterpret_cast<int4*>(bwp + 4) = make_int4(eps_bits, 0, 0, 0);
    sp.avail[b] = 0;
    reinterpret_cast<long long*>(sp.ack)[b] = 0;  // epoch 0 starts at record 0
    if (run && (((dirs >> depth) & 1) != (dir > 0 ? 1 : 0))) atomicAdd(sp.dbg + SPD_ORDER, 1);
  }
}

// ------------------------------------------------------------------------------------ free-running chains, shared dense metric on the GEMM
// (round 4)  One dense inverse mass matrix shared by all chains: every product v = M^{-1} p
// a tick needs is ONE fp32 MFMA GEMM over the compact rows (bjx_dense_apply_imm), exactly the arithmetic the
// lockstep `step` uses for this metric (nuts.py: dense_gemm; the oracle's "f32chain" mode) -- instead of D^2
// fp64-accumulated words per chain and product in k_nuts_async_fused<.., true>.  A tick is then a fixed
// sequence of launches on one stream (bjx_nuts_async_tick, GEMM mode), built from the SAME per-chain device
// functions as the lockstep kernels (nuts_post_chain / nuts_merge_chain / nuts_init_chain / nuts_open_half with
// bjx_nuts_t.v_pre), with the lane <-> element mapping each of them has there:
//   kick(1)   pc[b] = p_end + (dir eps b1) gf[b]            chains with a leaf in flight (phase 1)
//   GEMM      vc = pc M^{-1}
//   leaf      closing kick, energy, sampling, U-turn (-> phase 4); subtree complete: merge -> next doubling (phase 4)
//             | transition complete: record, accept; then (also phase 0) z = normal(km) into a slot of the
//             momentum list (phase 5) -- at most `cap` chains per tick, the others stay in phase 0 and
//             try again in the next tick (chains are independent: a chain's results do not depend on when
//             it runs)
//   GEMM x 2  pm = z L^{-1} ; vm = pm M^{-1}                 (metrics.py:260-270 as bjx_hmc_momentum_dense)
//   start     p0, v0, K(p0) -> tree init, doubling 0 (phase 4)
//             (leaf and start also write pc[b] = p_end + (dir eps b1) g_end for the chains that open a leaf)
//   GEMM      vc = pc M^{-1}
//   pre       q += (dir eps a1) vc[b], p += (dir eps b1) g_end -> qf[b]  (phase 1)
// Velocity Verlet / one-gradient integrators only (multi-stage integrators use lockstep steps for this metric).
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_nuts_gemm_kick(bjx_nuts_t nt, bjx_nuts_async_t ax, const float* __restrict__ gf, int want_phase) {
  if (want_phase == 1 && blockIdx.x == 0 && threadIdx.x == 0) *ax.end_count = 0;  // this tick's momentum list
  async_for_each_chain(ax, want_phase, want_phase, [&](int64_t c, int64_t b, int) {
    const int dir = IS(BJX_NUTS_I_DIR, c);
    const float h = ((float)dir * chain_eps(nt, c)) * int_kick(nt);
    const int64_t base = c * nt.D;
    const float* p = (dir > 0 ? nt.Rp : nt.Lp) + base;
    const float* g = want_phase == 1 ? gf + b * nt.D : (dir > 0 ? nt.Rg : nt.Lg) + base;
    float* out = ax.gemm_pc + b * nt.D;
    BJX_ROW_SWEEP(j0) {
      const Row<VEC> gg = ldr<VEC>(g + j0);
      Row<VEC> pp = ldr<VEC>(p + j0);
#pragma unroll
      for (int e = 0; e < VEC; ++e) pp.v[e] = fmaf(h, gg.v[e], pp.v[e]);
      str<VEC>(out + j0, pp);
    }
  });
}

// pc[b] = p_end + (dir eps b1) g_end for a chain that opens a leaf on its end `dir` (the expression of k_nuts_gemm_kick)
template <int VEC>
__device__ __forceinline__ void gemm_open_kick(const bjx_nuts_t& nt, int64_t c, int dir, float* __restrict__ out) {
  const float h = ((float)dir * chain_eps(nt, c)) * int_kick(nt);
  const float* p = (dir > 0 ? nt.Rp : nt.Lp) + c * nt.D;
  const float* g = (dir > 0 ? nt.Rg : nt.Lg) + c * nt.D;
  BJX_ROW_SWEEP(j0) {
    const Row<VEC> gg = ldr<VEC>(g + j0);
    Row<VEC> pp = ldr<VEC>(p + j0);
#pragma unroll
    for (int e = 0; e < VEC; ++e) pp.v[e] = fmaf(h, gg.v[e], pp.v[e]);
    str<VEC>(out + j0, pp);
  }
}

template <int VEC>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4)))  // 124 VGPRs
k_nuts_gemm_leaf(bjx_nuts_t nt, bjx_nuts_async_t ax, float* qf, const float* __restrict__ logp_f,
                 const float* __restrict__ gf, int32_t cap) {
  async_for_each_chain(ax, 1, 0, [&](int64_t c, int64_t b, int phase) {
    const int lane = threadIdx.x & 63;
    int32_t t = ax.t[c];
    if (phase == 1) {
      const StepCtx cx = async_ctx(nt, ax, t);
      const int32_t depth = IS(BJX_NUTS_I_DEPTH, c);
      const int32_t s = IS(BJX_NUTS_I_SUBN, c);
      const bool last = (s + 1) >= (1 << depth);
      const bool stop = nuts_post_chain<VEC, true>(nt, cx, c, b, depth, s, qf, logp_f, gf, false);
      if (!(stop || last)) {  // the subtree keeps integrating: the next leaf opens after this tick's second GEMM
        gemm_open_kick<VEC>(nt, c, IS(BJX_NUTS_I_DIR, c), ax.gemm_pc + b * nt.D);  // rows this wave's lanes just wrote
        if (lane == 0) ax.phase[c] = 4;
        return;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      const bool grow = nuts_merge_chain<1, true>(nt, cx, c, depth);  // <1, true>: the lockstep merge kernel's mapping
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      if (grow) {
        const int dir = nuts_begin_doubling(nt, cx, c, depth + 1);
        gemm_open_kick<VEC>(nt, c, dir, ax.gemm_pc + b * nt.D);
        if (lane == 0) ax.phase[c] = 4;
        return;
      }
      // transition t is complete: record it and make the proposal the chain's state (as async_boundary_chain)
      const int64_t base = c * nt.D;
      const int64_t row = (int64_t)t * nt.N + c;
      for (int64_t j = lane; j < nt.D; j += 64) {
        const float q = nt.Pq[base + j];
        ax.q[base + j] = q;
        ax.g[base + j] = nt.Pg[base + j];
        if (ax.out_position) ax.out_position[row * nt.D + j] = q;
      }
      if (lane == 0) {
        const float lp = FS(BJX_NUTS_F_PLOGP, c);
        ax.logp[c] = lp;
        if (ax.out_logdensity) ax.out_logdensity[row] = lp;
        if (ax.out_acceptance_rate) ax.out_acceptance_rate[row] = FS(BJX_NUTS_F_ACC, c);
        if (ax.out_energy) ax.out_energy[row] = FS(BJX_NUTS_F_PENERGY, c);
        if (ax.out_num_integration_steps) ax.out_num_integration_steps[row] = IS(BJX_NUTS_I_NSTATES, c);
        if (ax.out_num_trajectory_expansions) ax.out_num_trajectory_expansions[row] = IS(BJX_NUTS_I_DEPTH, c);
        if (ax.out_is_divergent) ax.out_is_divergent[row] = (uint8_t)(IS(BJX_NUTS_I_DIV, c) != 0);
        if (ax.out_is_turning) ax.out_is_turning[row] = (uint8_t)(IS(BJX_NUTS_I_TURN, c) != 0);
      }
      t += 1;
      if (lane == 0) ax.t[c] = t;
      if (t >= ax.n_steps) {
        if (lane == 0) {
          ax.phase[c] = 2;
          atomicAdd(ax.n_done, 1);
        }
        return;
      }
    }
    // start transition t: a slot of this tick's momentum list, or wait for the next tick
    int e = 0;
    if (lane == 0) e = atomicAdd(ax.end_count, 1);
    e = __builtin_amdgcn_readfirstlane(e);
    if (e >= cap) {
      if (lane == 0) ax.phase[c] = 0;
      return;
    }
    const StepCtx cx = async_ctx(nt, ax, t);
    const Key km = key_child(chain_key(cx.key, (uint64_t)(c + cx.off), cx.fold), 0);  // split(kc, 2)[0]
    float* z = ax.gemm_z + (int64_t)e * nt.D;
    for (int64_t j = lane; j < nt.D; j += 64) z[j] = normal_from_bits(key_bits32(km, (uint64_t)j));
    if (lane == 0) {
      ax.end_list[e] = (int32_t)c;
      ax.end_list[nt.N + e] = (int32_t)b;  // its compact row: k_nuts_gemm_start writes the opening kick there
      ax.phase[c] = 5;
    }
  });
}

// momentum list -> tree start: p0 = pm[e], v0 = vm[e], K = v0.p0 / 2 with the accumulation of k_rowdot_half
__global__ void __launch_bounds__(kBlock)
k_nuts_gemm_start(bjx_nuts_t nt, bjx_nuts_async_t ax, int32_t cap) {
  const int lane = threadIdx.x & 63;
  int64_t n = (int64_t)__builtin_amdgcn_readfirstlane(*ax.end_count);
  if (n > cap) n = cap;
  for (int64_t e = wave_row0(); e < n; e += wave_row_stride()) {
    const int64_t c = (int64_t)__builtin_amdgcn_readfirstlane(ax.end_list[e]);
    const int64_t base = c * nt.D;
    const float* pm = ax.gemm_pm + e * nt.D;
    const float* vm = ax.gemm_vm + e * nt.D;
    double acc = 0.0;
    for (int64_t j = lane; j < nt.D; j += 64) {
      const float p = pm[j], v = vm[j];
      ax.p[base + j] = p;
      ax.v0[base + j] = v;
      acc += (double)v * (double)p;
    }
    acc = wave_sum(acc);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // p0 / v0: read back by nuts_init_chain
    nuts_init_chain<1, true>(nt, c, ax.logp[c], 0.5f * (float)acc);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const StepCtx cx = async_ctx(nt, ax, ax.t[c]);
    const int dir = nuts_begin_doubling(nt, cx, c, 0);
    const int64_t b = (int64_t)__builtin_amdgcn_readfirstlane(ax.end_list[nt.N + e]);
    gemm_open_kick<1>(nt, c, dir, ax.gemm_pc + b * nt.D);  // rows nuts_init_chain<1> just wrote, same lanes
    if (lane == 0) ax.phase[c] = 4;
  }
}

template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_nuts_gemm_pre(bjx_nuts_t nt, bjx_nuts_async_t ax, float* __restrict__ qf) {
  async_for_each_chain(ax, 4, 4, [&](int64_t c, int64_t b, int) {
    const int dir = IS(BJX_NUTS_I_DIR, c);
    const float deps = (float)dir * chain_eps(nt, c);
    const float h = deps * int_kick(nt);
    const float* fg = (dir > 0 ? nt.Rg : nt.Lg) + c * nt.D;
    nuts_open_half<VEC, true>(nt, c, dir, deps * int_drift(nt), h, fg, qf + b * nt.D, nt.v_pre + b * nt.D);
    if ((threadIdx.x & 63) == 0) ax.phase[c] = 1;
  });
}
#endif  // !__HIPCC_RTC__

// Compaction of the free-running rows: keep, in order, the rows whose chain is not finished.
// One 1024-thread workgroup (same ballot + LDS scan as k_nuts_compact); src[b'] remembers the old
// row so the pending positions can be gathered by k_nuts_async_gather.
__global__ void __launch_bounds__(1024)
k_nuts_async_compact(bjx_nuts_async_t ax, int32_t* __restrict__ rows_out, int32_t* __restrict__ src,
                     int32_t* __restrict__ n_out) {
  __shared__ int wave_counts[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  const int64_t n_rows = async_n_rows(ax);
  for (int64_t start = 0; start < n_rows; start += 1024) {
    const int64_t i = start + tid;
    int32_t c = -1;
    if (i < n_rows) c = ax.rows ? ax.rows[i] : (int32_t)i;
    const bool keep = c >= 0 && ax.phase[c] != 2;
    const unsigned long long ballot = __ballot(keep);
    const int lane_prefix = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) wave_counts[wave] = __popcll(ballot);
    __syncthreads();
    int wave_off = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
      const int cnt = wave_counts[w];
      if (w < wave) wave_off += cnt;
      total += cnt;
    }
    const int b0 = base;
    if (keep) {
      rows_out[b0 + wave_off + lane_prefix] = c;
      src[b0 + wave_off + lane_prefix] = (int32_t)i;
    }
    __syncthreads();
    if (tid == 0) base = b0 + total;
    __syncthreads();
  }
  if (tid == 0) *n_out = base;
}

__global__ void __launch_bounds__(kBlock)
k_nuts_async_gather(int64_t D, const int32_t* __restrict__ n_rows, const int32_t* __restrict__ src,
                    const float* __restrict__ qf_in, float* __restrict__ qf_out) {
  const int lane = threadIdx.x & 63;
  const int64_t n = *n_rows;
  for (int64_t b = wave_row0(); b < n; b += wave_row_stride()) {
    const float* s = qf_in + (int64_t)src[b] * D;
    float* d = qf_out + b * D;
    for (int64_t j = lane; j < D; j += 64) d[j] = s[j];
  }
}

__global__ void k_nuts_set_ctl(int64_t* ctl, int64_t depth, int64_t s_base, int64_t n_rows,
                               int64_t key0, int64_t key1, int64_t fold, int64_t off) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    ctl[0] = depth; ctl[1] = s_base; ctl[3] = key0;
    ctl[4] = key1; ctl[5] = fold; ctl[6] = off; ctl[7] = 0;
    if (n_rows >= 0) ctl[2] = n_rows;  // negative: keep the count written by k_nuts_compact
  }
}

// Device-side active-chain compaction (no host round trip): keeps, in order, the chains of
// idx_in[0..n_in) (identity list if idx_in == NULL) whose flag slot is set, writes them to
// idx_out (may alias idx_in) and the count to ctl[2].  One 1024-thread workgroup; wave ballots
// + a 16-entry LDS scan per 1024-entry slice.
constexpr int kCompactThreads = 1024;
__global__ void __launch_bounds__(kCompactThreads)
k_nuts_compact(bjx_nuts_t nt, int flag_slot, int64_t n_in_arg, const int32_t* idx_in,
               int32_t* idx_out, int64_t* ctl) {
  __shared__ int wave_counts[kCompactThreads / 64];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n_in = n_in_arg >= 0 ? n_in_arg : ctl[2];
  if (tid == 0) base = 0;
  __syncthreads();
  for (int64_t start = 0; start < n_in; start += kCompactThreads) {
    const int64_t i = start + tid;
    int32_t c = -1;
    if (i < n_in) c = idx_in ? idx_in[i] : (int32_t)i;
    const bool keep = c >= 0 && IS(flag_slot, c) != 0 && IS(BJX_NUTS_I_ACTIVE, c) != 0;
    const unsigned long long ballot = __ballot(keep);
    const int lane_prefix = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) wave_counts[wave] = __popcll(ballot);
    __syncthreads();  // every read of this slice is done before any write below
    int wave_off = 0, total = 0;
    for (int w = 0; w < kCompactThreads / 64; ++w) {
      const int cnt = wave_counts[w];
      if (w < wave) wave_off += cnt;
      total += cnt;
    }
    const int b0 = base;
    if (keep) idx_out[b0 + wave_off + lane_prefix] = c;
    __syncthreads();
    if (tid == 0) base = b0 + total;
    __syncthreads();
  }
  if (tid == 0) ctl[2] = base;
}

#ifndef __HIPCC_RTC__  // ---- host side from here on
int check_nuts(const bjx_nuts_t* nt, const char* what) {
  if (!nt) { bjx_set_error("%s: null descriptor", what); return 1; }
  if (nt->N == 0 && nt->D > 0 && nt->max_depth >= 0 && nt->max_depth <= 30) return 0;  // callers return next
  const bool dense_ok =
      !nt->Mdense || ((nt->Mdense_stride == 0 || nt->Mdense_stride == nt->D * nt->D) && nt->v0 &&
                      nt->Lv && nt->Rv && (nt->max_depth == 0 || nt->ckpt_v));
  const bool ok = nt->N >= 0 && nt->D > 0 && nt->max_depth >= 0 && nt->max_depth <= 30 &&
                  (nt->Mdense || nt->imm) && (nt->imm_stride == 0 || nt->imm_stride == nt->D) &&
                  nt->q0 && nt->g0 && nt->p0 && nt->Lq && nt->Lp && nt->Lg && nt->Rq && nt->Rp &&
                  nt->Rg && nt->msum && nt->Smsum && nt->Pq && nt->Pg && nt->Sq && nt->Sg && nt->fs &&
                  nt->is && (nt->max_depth == 0 || (nt->ckpt_r && nt->ckpt_rs)) && dense_ok;
  if (!ok) { bjx_set_error("%s: bad descriptor", what); return 1; }
  return 0;
}

// 16-byte row accesses are legal when the metric is diagonal, D % 4 == 0 and every (N, D) buffer
// the kernels touch is 16-byte aligned (rows then are, too).
template <typename... P>
bool nuts_vec4(const bjx_nuts_t* nt, P... extra) {
  return !nt->Mdense &&
         bjx_vec4_ok(nt->D, nt->imm, nt->q0, nt->g0, nt->p0, nt->Lq, nt->Lp, nt->Lg, nt->Rq, nt->Rp,
                     nt->Rg, nt->msum, nt->Smsum, nt->Pq, nt->Pg, nt->Sq, nt->Sg, nt->ckpt_r,
                     nt->ckpt_rs, extra...);
}

// rows per lane of the register-resident leaf (0 = use the general sweeps)
template <typename... P>
int nuts_resident_ni(const bjx_nuts_t* nt, P... extra) {
  if (!nuts_vec4(nt, extra...)) return 0;
  return nt->D <= 256 ? 1 : (nt->D <= 512 ? 2 : 0);
}

// 16-byte sweeps for a dense metric whose velocities come from the caller's GEMM (v_pre): every row array
// the pre / mid / post kernels touch is aligned and D % 4 == 0
template <typename... P>
bool nuts_vec4_dense(const bjx_nuts_t* nt, P... extra) {
  return nt->Mdense && nt->v_pre &&
         bjx_vec4_ok(nt->D, nt->v_pre, nt->Lq, nt->Lp, nt->Lg, nt->Rq, nt->Rp, nt->Rg, nt->Lv, nt->Rv, nt->msum,
                     nt->Smsum, nt->Sq, nt->Sg, nt->ckpt_r, nt->ckpt_rs, nt->ckpt_v, extra...);
}
#define BJX_NUTS_LAUNCH_V(KERNEL, grid, stream, vec4, dense, vec4_dense, ...)                          \
  do {                                                                                                 \
    if (vec4_dense) hipLaunchKernelGGL((KERNEL<4, true>), grid, dim3(kBlock), 0, stream, __VA_ARGS__); \
    else BJX_NUTS_LAUNCH(KERNEL, grid, stream, vec4, dense, __VA_ARGS__);                              \
  } while (0)

// pick the <VEC, DENSE> instantiation of a kernel template
#define BJX_NUTS_LAUNCH(KERNEL, grid, stream, vec4, dense, ...)                                        \
  do {                                                                                                 \
    if (dense) hipLaunchKernelGGL((KERNEL<1, true>), grid, dim3(kBlock), 0, stream, __VA_ARGS__);      \
    else if (vec4) hipLaunchKernelGGL((KERNEL<4, false>), grid, dim3(kBlock), 0, stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<1, false>), grid, dim3(kBlock), 0, stream, __VA_ARGS__);           \
  } while (0)

#endif  // !__HIPCC_RTC__
}  // namespace

#ifndef __HIPCC_RTC__
extern "C" {

int bjx_nuts_init(void* stream, const bjx_nuts_t* nuts, const float* logp0, const float* ke0) {
  if (check_nuts(nuts, "bjx_nuts_init")) return 1;
  if (nuts->N == 0) return 0;  // an empty ensemble has no buffers to check
  BJX_CHECK_ARG(logp0 && ke0, "bjx_nuts_init: bad arguments");
  if (nuts->N == 0) return 0;
  const dim3 grid(bjx_row_grid(nuts->N, kWavesPerBlock));
  BJX_NUTS_LAUNCH(k_nuts_init, grid, (hipStream_t)stream, nuts_vec4(nuts), nuts->Mdense != nullptr, *nuts,
                  logp0, ke0);
  return bjx_check_launch("bjx_nuts_init");
}

int bjx_nuts_pre(void* stream, const bjx_nuts_t* nuts, int32_t depth, int64_t s, int64_t n_rows,
                 const int32_t* idx, float* qf) {
  if (check_nuts(nuts, "bjx_nuts_pre")) return 1;
  if (nuts->N == 0) return 0;  // an empty ensemble has no buffers to check
  BJX_CHECK_ARG(depth >= 0 && depth < nuts->max_depth && s >= 0 && s < ((int64_t)1 << depth) &&
                    n_rows >= 0 && n_rows <= nuts->N && qf,
                "bjx_nuts_pre: bad arguments");
  if (n_rows == 0) return 0;
  const dim3 grid(bjx_row_grid(n_rows, kWavesPerBlock));
  BJX_NUTS_LAUNCH_V(k_nuts_pre, grid, (hipStream_t)stream, nuts_vec4(nuts, qf), nuts->Mdense != nullptr,
                    nuts_vec4_dense(nuts, qf),
                  *nuts, depth, (int32_t)s, n_rows, idx, (const int64_t*)nullptr, qf);
  return bjx_check_launch("bjx_nuts_pre");
}

int bjx_nuts_pre_ctl(void* stream, const bjx_nuts_t* nuts, int32_t s_off, int64_t n_cap,
                     const int32_t* idx, const int64_t* ctl, float* qf) {
  if (check_nuts(nuts, "bjx_nuts_pre_ctl")) return 1;
  if (nuts->N == 0) return 0;  // an empty ensemble has no buffers to check
  BJX_CHECK_ARG(s_off >= 0 && n_cap >= 0 && n_cap <= nuts->N && idx && ctl && qf,
                "bjx_nuts_pre_ctl: bad arguments");
  if (n_cap == 0) return 0;
  const dim3 grid(bjx_row_grid(n_cap, kWavesPerBlock));
  BJX_NUTS_LAUNCH_V(k_nuts_pre, grid, (hipStream_t)stream, nuts_vec4(nuts, qf), nuts->Mdense != nullptr,
                    nuts_vec4_dense(nuts, qf),
                  *nuts, 0, s_off, n_cap, idx, ctl, qf);
  return bjx_check_launch("bjx_nuts_pre_ctl");
}

int bjx_nuts_dense_kick(void* stream, const bjx_nuts_t* nuts, int32_t depth, int64_t s, int64_t n_rows,
                        const int32_t* idx, const int64_t* ctl, const float* gf, float kick, float* pc_out) {
  if (check_nuts(nuts, "bjx_nuts_dense_kick")) return 1;
  if (nuts->N == 0) return 0;  // an empty ensemble has no buffers to check
  BJX_CHECK_ARG(nuts->Mdense && nuts->Mdense_stride == 0, "bjx_nuts_dense_kick: needs a shared dense metric");
  BJX_CHECK_ARG(depth >= 0 && s >= 0 && n_rows >= 0 && n_rows <= nuts->N && pc_out && (idx || !ctl),
                "bjx_nuts_dense_kick: bad arguments");
  if (n_rows == 0) return 0;
  const dim3 grid(bjx_row_grid(n_rows, kWavesPerBlock));
  if (bjx_vec4_ok(nuts->D, nuts->Lp, nuts->Rp, nuts->Lg, nuts->Rg, gf, pc_out))
    hipLaunchKernelGGL(k_nuts_dense_kick<4>, grid, dim3(kBlock), 0, (hipStream_t)stream, *nuts, depth, (int32_t)s,
                       n_rows, idx, ctl, gf, kick, pc_out);
  else
    hipLaunchKernelGGL(k_nuts_dense_kick<1>, grid, dim3(kBlock), 0, (hipStream_t)stream, *nuts, depth, (int32_t)s,
                       n_rows, idx, ctl, gf, kick, pc_out);
  return bjx_check_launch("bjx_nuts_dense_kick");
}

int bjx_nuts_mid(void* stream, const bjx_nuts_t* nuts, int64_t n_rows, const int32_t* idx,
                 const int64_t* ctl, float* qf, const float* gf, float kick, float drift) {
  if (check_nuts(nuts, "bjx_nuts_mid")) return 1;
  if (nuts->N == 0) return 0;  // an empty ensemble has no buffers to check
  BJX_CHECK_ARG(n_rows >= 0 && n_rows <= nuts->N && qf && gf && (idx || !ctl), "bjx_nuts_mid: bad arguments");
  if (n_rows == 0) return 0;
  const dim3 grid(bjx_row_grid(n_rows, kWavesPerBlock));
  BJX_NUTS_LAUNCH_V(k_nuts_mid, grid, (hipStream_t)stream, nuts_vec4(nuts, qf, gf), nuts->Mdense != nullptr,
                    nuts_vec4_dense(nuts, qf, gf),
                  *nuts, n_rows, idx, ctl, qf, gf, kick, drift);
  return bjx_check_launch("bjx_nuts_mid");
}

int bjx_nuts_post(void* stream, const bjx_nuts_t* nuts, int32_t depth, int64_t s, int64_t n_rows,
                  const int32_t* idx, float* qf, const float* logp_f, const float* gf,
                  int32_t fuse_next) {
  if (check_nuts(nuts, "bjx_nuts_post")) return 1;
  if (nuts->N == 0) return 0;  // an empty ensemble has no buffers to check
  BJX_CHECK_ARG(depth >= 0 && depth < nuts->max_depth && s >= 0 && s < ((int64_t)1 << depth) &&
                    n_rows >= 0 && n_rows <= nuts->N && qf && logp_f && gf,
                "bjx_nuts_post: bad arguments");
  if (n_rows == 0) return 0;
  const dim3 grid(bjx_row_grid(n_rows, kWavesPerBlock));
  const int fuse = (int)(fuse_next && s + 1 < ((int64_t)1 << depth));
  const int ni = nuts_resident_ni(nuts, qf, gf);
  hipStream_t st = (hipStream_t)stream;
  if (ni == 1)
    hipLaunchKernelGGL(k_nuts_post_res<1>, grid, dim3(kBlock), 0, st, *nuts, depth, (int32_t)s, n_rows, idx,
                       (const int64_t*)nullptr, qf, logp_f, gf, fuse);
  else if (ni == 2)
    hipLaunchKernelGGL(k_nuts_post_res<2>, grid, dim3(kBlock), 0, st, *nuts, depth, (int32_t)s, n_rows, idx,
                       (const int64_t*)nullptr, qf, logp_f, gf, fuse);
  else
    BJX_NUTS_LAUNCH_V(k_nuts_post, grid, st, nuts_vec4(nuts, qf, gf), nuts->Mdense != nullptr,
                      nuts_vec4_dense(nuts, qf, gf), *nuts, depth,
                    (int32_t)s, n_rows, idx, (const int64_t*)nullptr, qf, logp_f, gf, fuse);
  return bjx_check_launch("bjx_nuts_post");
}

int bjx_nuts_post_ctl(void* stream, const bjx_nuts_t* nuts, int32_t s_off, int64_t n_cap,
                      const int32_t* idx, const int64_t* ctl, float* qf, const float* logp_f,
                      const float* gf, int32_t fuse_next) {
  if (check_nuts(nuts, "bjx_nuts_post_ctl")) return 1;
  if (nuts->N == 0) return 0;  // an empty ensemble has no buffers to check
  BJX_CHECK_ARG(s_off >= 0 && n_cap >= 0 && n_cap <= nuts->N && idx && ctl && qf && logp_f && gf,
                "bjx_nuts_post_ctl: bad arguments");
  if (n_cap == 0) return 0;
  const dim3 grid(bjx_row_grid(n_cap, kWavesPerBlock));
  const int ni = nuts_resident_ni(nuts, qf, gf);
  hipStream_t st = (hipStream_t)stream;
  if (ni == 1)
    hipLaunchKernelGGL(k_nuts_post_res<1>, grid, dim3(kBlock), 0, st, *nuts, 0, s_off, n_cap, idx, ctl, qf,
                       logp_f, gf, (int)fuse_next);
  else if (ni == 2)
    hipLaunchKernelGGL(k_nuts_post_res<2>, grid, dim3(kBlock), 0, st, *nuts, 0, s_off, n_cap, idx, ctl, qf,
                       logp_f, gf, (int)fuse_next);
  else
    BJX_NUTS_LAUNCH_V(k_nuts_post, grid, st, nuts_vec4(nuts, qf, gf), nuts->Mdense != nullptr,
                      nuts_vec4_dense(nuts, qf, gf), *nuts, 0, s_off,
                    n_cap, idx, ctl, qf, logp_f, gf, (int)fuse_next);
  return bjx_check_launch("bjx_nuts_post_ctl");
}

int bjx_nuts_compact(void* stream, const bjx_nuts_t* nuts, int32_t flag_slot, int64_t n_in,
                     const int32_t* idx_in, int32_t* idx_out, int64_t* ctl) {
  if (check_nuts(nuts, "bjx_nuts_compact")) return 1;
  if (nuts->N == 0) return 0;  // an empty ensemble has no buffers to check
  BJX_CHECK_ARG((flag_slot == BJX_NUTS_I_ACTIVE || flag_slot == BJX_NUTS_I_SUB_ACTIVE) && idx_out &&
                    ctl && n_in <= nuts->N,
                "bjx_nuts_compact: bad arguments");
  hipLaunchKernelGGL(k_nuts_compact, dim3(1), dim3(kCompactThreads), 0, (hipStream_t)stream, *nuts,
                     (int)flag_slot, n_in, idx_in, idx_out, ctl);
  return bjx_check_launch("bjx_nuts_compact");
}

int bjx_nuts_set_ctl(void* stream, int64_t* ctl, int32_t depth, int64_t s_base, int64_t n_rows,
                     uint32_t key0, uint32_t key1, int64_t step_fold, int64_t chain_offset) {
  BJX_CHECK_ARG(ctl && depth >= 0 && s_base >= 0, "bjx_nuts_set_ctl: bad arguments");
  hipLaunchKernelGGL(k_nuts_set_ctl, dim3(1), dim3(64), 0, (hipStream_t)stream, ctl, (int64_t)depth,
                     s_base, n_rows, (int64_t)key0, (int64_t)key1, step_fold, chain_offset);
  return bjx_check_launch("bjx_nuts_set_ctl");
}

int bjx_nuts_merge(void* stream, const bjx_nuts_t* nuts, int32_t depth, int64_t n_rows,
                   const int32_t* idx) {
  if (check_nuts(nuts, "bjx_nuts_merge")) return 1;
  if (nuts->N == 0) return 0;  // an empty ensemble has no buffers to check
  BJX_CHECK_ARG(depth >= 0 && depth < nuts->max_depth && n_rows >= 0 && n_rows <= nuts->N,
                "bjx_nuts_merge: bad arguments");
  if (n_rows == 0) return 0;
  const dim3 grid(bjx_row_grid(n_rows, kWavesPerBlock));
  BJX_NUTS_LAUNCH(k_nuts_merge, grid, (hipStream_t)stream, nuts_vec4(nuts), nuts->Mdense != nullptr,
                  *nuts, depth, n_rows, idx);
  return bjx_check_launch("bjx_nuts_merge");
}

int bjx_nuts_async_tick(void* stream, const bjx_nuts_t* nuts, const bjx_nuts_async_t* run, float* qf,
                        const float* logp_f, const float* gf) {
  if (check_nuts(nuts, "bjx_nuts_async_tick")) return 1;
  if (nuts->N == 0) return 0;  // an empty ensemble has no buffers to check
  BJX_CHECK_ARG(run && qf && logp_f && gf, "bjx_nuts_async_tick: null argument");
  BJX_CHECK_ARG(!nuts->Mdense || (run->mass_sqrt_t && run->v0 && run->v0 == nuts->v0 && !run->adapt_tab),
                "bjx_nuts_async_tick: a dense metric needs run->mass_sqrt_t, run->v0 == nuts->v0 and no "
                "per-chain adaptation (adapt_tab adapts a diagonal metric)");
  BJX_CHECK_ARG(nuts->max_depth >= 1, "bjx_nuts_async_tick: max_depth must be >= 1");
  BJX_CHECK_ARG(run->n_steps >= 0 && run->t_first >= 0 && run->q && run->g && run->logp && run->p &&
                    run->t && run->phase && run->n_done,
                "bjx_nuts_async_tick: bad run descriptor");
  BJX_CHECK_ARG(run->q == nuts->q0 && run->g == nuts->g0 && run->p == nuts->p0,
                "bjx_nuts_async_tick: run->q / g / p must alias nuts->q0 / g0 / p0");
  BJX_CHECK_ARG(run->n_rows >= 0 && run->n_rows <= nuts->N && (run->rows || run->n_rows == nuts->N),
                "bjx_nuts_async_tick: n_rows must be N when rows is NULL and never exceed N");
  BJX_CHECK_ARG(!run->adapt_tab ||
                    (run->adapt_log_x && run->adapt_log_x_avg && run->adapt_avg_err && run->adapt_mu &&
                     run->adapt_step_size && run->adapt_mean && run->adapt_m2 && run->adapt_imm &&
                     nuts->eps_per_chain == run->adapt_step_size && nuts->imm == run->adapt_imm &&
                     nuts->imm_stride == nuts->D),
                "bjx_nuts_async_tick: adaptation needs every adapt_* buffer, nuts->eps_per_chain == "
                "adapt_step_size and nuts->imm == adapt_imm with imm_stride == D");
  BJX_CHECK_ARG(run->target_kind == BJX_TARGET_NONE ||
                    ((run->target_kind == BJX_TARGET_NEAL_FUNNEL ||
                      (run->target_kind == BJX_TARGET_DIAG_GAUSSIAN && run->target_vec && nuts->D > 128)) &&
                     !nuts->Mdense && run->rec && run->front_p),
                "bjx_nuts_async_tick: target_kind needs the low-traffic tick kernels (diagonal metric, rec / "
                "front_p) and a supported target (funnel; diagonal Gaussian with target_vec and D > 128)");
  BJX_CHECK_ARG(run->ticks_per_launch <= 1 || run->target_kind != BJX_TARGET_NONE,
                "bjx_nuts_async_tick: ticks_per_launch > 1 needs an engine-resident target (target_kind)");
  if (run->n_rows == 0 || run->n_steps == 0) return 0;
  if (nuts->Mdense && run->gemm_pc) {
    // ONE shared dense matrix, products on the MFMA GEMM (see "free-running chains, shared dense metric on the GEMM")
    BJX_CHECK_ARG(nuts->Mdense_stride == 0 && nuts->v_pre && nuts->v_pre == run->gemm_vc && run->gemm_z &&
                      run->gemm_pm && run->gemm_vm && run->gemm_cap >= 1 && run->end_list && run->end_count &&
                      run->int_stages <= 1 && run->target_kind == BJX_TARGET_NONE,
                  "bjx_nuts_async_tick: GEMM mode needs one shared dense matrix (Mdense_stride == 0), nuts->v_pre == "
                  "run->gemm_vc, gemm_z / gemm_pm / gemm_vm with gemm_cap >= 1, end_list / end_count, a one-gradient "
                  "integrator and no engine-resident target");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = run->n_rows, D = nuts->D;
    const int32_t cap = (int32_t)(run->gemm_cap < n ? run->gemm_cap : n);
    const dim3 rgrid(bjx_row_grid(n, kWavesPerBlock)), cgrid(bjx_row_grid(cap, kWavesPerBlock)), blk(kBlock);
    const bool v4 = nuts_vec4_dense(nuts, qf, gf, run->gemm_pc);
#define BJX_GEMM_K(KERNEL, ...)                                                          \
  do {                                                                                   \
    if (v4) hipLaunchKernelGGL(KERNEL<4>, rgrid, blk, 0, st, *nuts, *run, __VA_ARGS__);  \
    else hipLaunchKernelGGL(KERNEL<1>, rgrid, blk, 0, st, *nuts, *run, __VA_ARGS__);     \
    if (int rc = bjx_check_launch("bjx_nuts_async_tick(gemm)")) return rc;               \
  } while (0)
    auto apply_imm = [&](int64_t rows, const float* p_in, float* v_out) {
      return run->gemm_imm_t ? bjx_dense_apply_imm_t(stream, rows, D, p_in, nuts->Mdense, run->gemm_imm_t, v_out)
                             : bjx_dense_apply_imm(stream, rows, D, p_in, nuts->Mdense, v_out);
    };
    BJX_GEMM_K(k_nuts_gemm_kick, gf, 1);
    if (int rc = apply_imm(n, run->gemm_pc, run->gemm_vc)) return rc;
    BJX_GEMM_K(k_nuts_gemm_leaf, qf, logp_f, gf, cap);
    if (run->gemm_mass_sqrt) {  // p = L^{-T} z with the matrix read as stored where that kernel applies (faster)
      if (int rc = bjx_dense_matmul_bt(stream, cap, D, run->gemm_z, run->mass_sqrt_t, run->gemm_mass_sqrt, run->gemm_pm))
        return rc;
    } else if (int rc = bjx_dense_matmul(stream, cap, D, run->gemm_z, run->mass_sqrt_t, run->gemm_pm)) return rc;
    if (int rc = apply_imm(cap, run->gemm_pm, run->gemm_vm)) return rc;
    hipLaunchKernelGGL(k_nuts_gemm_start, cgrid, blk, 0, st, *nuts, *run, cap);
    if (int rc = bjx_check_launch("bjx_nuts_async_tick(gemm start)")) return rc;
    // (the opening kicks pc[b] = p_end + (dir eps b1) g_end were written by the leaf / start kernels)
    if (int rc = apply_imm(n, run->gemm_pc, run->gemm_vc)) return rc;
    BJX_GEMM_K(k_nuts_gemm_pre, qf);
#undef BJX_GEMM_K
    return 0;
  }
  if (nuts->Mdense) {
    // dense metric: every leaf is a D x D matrix-vector product per chain (fp64 accumulated, the
    // arithmetic of the lockstep kernels), so one launch per tick whatever the row count
    hipLaunchKernelGGL((k_nuts_async_fused<1, 0, true>), dim3(bjx_row_grid(run->n_rows, kWavesPerBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, *nuts, *run, qf, logp_f, gf);
    return bjx_check_launch("bjx_nuts_async_tick");
  }
  hipStream_t s = (hipStream_t)stream;
  // Which kernel ticks a batch (round 5: ONE choice per shape, no environment switches -- the measured losers of
  // rounds 1-4 are recorded in NOTEBOOK.md sections 7, 15 and no longer compiled in):
  //   diagonal metric, 16-byte rows of at most 1 024 floats, external callable
  //       -> k_nuts_async_tick3<NI, W>: lean leaf + the transition ends deferred from the tick before, one launch
  //          per tick at every batch size; multi-stage integrators included
  //   the same shapes with an engine-resident target (fuse_target, outside the callable contract, D <= 512)
  //       -> k_nuts_async_multi: ticks_per_launch ticks of every row per launch
  //   everything else (4-byte sweeps, rows beyond 1 024 floats; per-chain dense metrics are handled above)
  //       -> k_nuts_async_fused: the general one-launch tick
  const bool all_vec4 = nuts_vec4(nuts, qf, gf, run->out_position, run->adapt_mean, run->adapt_m2, run->adapt_imm);
  const int ni2 = all_vec4 ? nuts_resident_ni(nuts, qf, gf) : 0;  // 1 / 2: rows of at most 256 / 512 floats
  const bool lean = all_vec4 && nuts->D <= 1024 && run->rec && run->front_p;
  if (run->target_kind != BJX_TARGET_NONE) {
    BJX_CHECK_ARG(lean && ni2 > 0,
                  "bjx_nuts_async_tick: target_kind is served by the low-traffic kernels only (D % 4 == 0, D <= 512, "
                  "16-byte aligned buffers)");
    bjx_nuts_async_t r = *run;
    if (r.ticks_per_launch < 1) r.ticks_per_launch = 1;
    const dim3 wgrid((unsigned)(r.n_rows < (int64_t)1 << 20 ? r.n_rows : (int64_t)1 << 20));  // one wave per workgroup
#define BJX_MULTI(NI_)                                                                                          \
  do {                                                                                                          \
    if (nuts->D == 256 * NI_)                                                                                   \
      hipLaunchKernelGGL((k_nuts_async_multi<NI_, 2, true>), wgrid, dim3(64), 0, s, *nuts, r, qf,               \
                         const_cast<float*>(logp_f), const_cast<float*>(gf));                                   \
    else                                                                                                        \
      hipLaunchKernelGGL((k_nuts_async_multi<NI_, 2, false>), wgrid, dim3(64), 0, s, *nuts, r, qf,              \
                         const_cast<float*>(logp_f), const_cast<float*>(gf));                                   \
  } while (0)
    if (ni2 == 1) BJX_MULTI(1);
    else BJX_MULTI(2);
#undef BJX_MULTI
    return bjx_check_launch("bjx_nuts_async_tick");
  }
  if (lean) {
    const dim3 g1((unsigned)run->n_rows);  // one wave (= one workgroup) per compact row
    if (ni2 == 1) hipLaunchKernelGGL((k_nuts_async_tick3<1, 4>), g1, dim3(64), 0, s, *nuts, *run, qf, logp_f, gf);
    else if (ni2 == 2) hipLaunchKernelGGL((k_nuts_async_tick3<2, 4>), g1, dim3(64), 0, s, *nuts, *run, qf, logp_f, gf);
    else if (nuts->D <= 768) hipLaunchKernelGGL((k_nuts_async_tick3<3, 2>), g1, dim3(64), 0, s, *nuts, *run, qf, logp_f, gf);
    else hipLaunchKernelGGL((k_nuts_async_tick3<4, 2>), g1, dim3(64), 0, s, *nuts, *run, qf, logp_f, gf);
    return bjx_check_launch("bjx_nuts_async_tick");
  }
  const dim3 fgrid(bjx_row_grid(run->n_rows, kWavesPerBlock));  // one wave per row
  if (all_vec4) {
    const int ni = nuts_resident_ni(nuts, qf, gf);
    if (ni == 1) hipLaunchKernelGGL((k_nuts_async_fused<4, 1>), fgrid, dim3(kBlock), 0, s, *nuts, *run, qf, logp_f, gf);
    else if (ni == 2) hipLaunchKernelGGL((k_nuts_async_fused<4, 2>), fgrid, dim3(kBlock), 0, s, *nuts, *run, qf, logp_f, gf);
    else hipLaunchKernelGGL((k_nuts_async_fused<4, 0>), fgrid, dim3(kBlock), 0, s, *nuts, *run, qf, logp_f, gf);
  } else {
    hipLaunchKernelGGL((k_nuts_async_fused<1, 0>), fgrid, dim3(kBlock), 0, s, *nuts, *run, qf, logp_f, gf);
  }
  return bjx_check_launch("bjx_nuts_async_tick");
}

int bjx_nuts_async_compact(void* stream, const bjx_nuts_t* nuts, const bjx_nuts_async_t* run,
                           const float* qf_in, int32_t* rows_out, float* qf_out, int32_t* src_work,
                           int32_t* n_out) {
  if (check_nuts(nuts, "bjx_nuts_async_compact")) return 1;
  if (nuts->N == 0) return 0;  // an empty ensemble has no buffers to check
  BJX_CHECK_ARG(run && run->phase && qf_in && rows_out && qf_out && src_work && n_out,
                "bjx_nuts_async_compact: null argument");
  BJX_CHECK_ARG(rows_out != run->rows && qf_out != qf_in, "bjx_nuts_async_compact: outputs must not alias inputs");
  BJX_CHECK_ARG(run->n_rows >= 0 && run->n_rows <= nuts->N && (run->rows || run->n_rows == nuts->N),
                "bjx_nuts_async_compact: bad row count");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_nuts_async_compact, dim3(1), dim3(1024), 0, s, *run, rows_out, src_work, n_out);
  if (run->n_rows > 0)
    hipLaunchKernelGGL(k_nuts_async_gather, dim3(bjx_row_grid(run->n_rows, kWavesPerBlock)), dim3(kBlock), 0,
                       s, nuts->D, n_out, src_work, qf_in, qf_out);
  return bjx_check_launch("bjx_nuts_async_compact");
}

int bjx_stream_probe(void* stream_wait, void* stream_set, int32_t* flag2, int32_t timeout_us) {
  BJX_CHECK_ARG(flag2 && timeout_us > 0 && stream_wait != stream_set, "bjx_stream_probe: two streams, flag2 and a time-out needed");
  hipLaunchKernelGGL(k_stream_probe_wait, dim3(1), dim3(64), 0, (hipStream_t)stream_wait, flag2, (long long)timeout_us * 100);
  hipLaunchKernelGGL(k_stream_probe_set, dim3(1), dim3(64), 0, (hipStream_t)stream_set, flag2);
  return bjx_check_launch("bjx_stream_probe");
}

namespace {
int check_spec(const bjx_nuts_t* nuts, const bjx_nuts_async_t* run, const bjx_nuts_spec_t* sp, const char* what) {
  if (check_nuts(nuts, what)) return 1;
  const bool ok = run && sp && run->q && run->g && run->logp && run->p && run->t && run->phase && run->n_done &&
                  run->q == nuts->q0 && run->g == nuts->g0 && run->p == nuts->p0 && run->rec && run->front_p &&
                  !nuts->Mdense && !run->adapt_tab && !run->gemm_pc && run->target_kind == BJX_TARGET_NONE &&
                  run->int_stages <= 1 && nuts->max_depth >= 1 && nuts->max_depth <= 30 && nuts->D <= 1024 &&
                  sp->n_rows >= 1 && sp->n_rows <= nuts->N && sp->rows && sp->ring >= 8 &&
                  (sp->ring & (sp->ring - 1)) == 0 && sp->qf && sp->fp && sp->eLq && sp->eLp && sp->eLg && sp->eRq &&
                  sp->eRp && sp->eRg && sp->iw && sp->ring_g && sp->ring_tag && sp->avail && sp->ack && sp->qf_book && sp->bw &&
                  sp->a_seq && sp->dbg &&
                  nuts_vec4(nuts, sp->qf, sp->fp, sp->eLq, sp->eLp, sp->eLg, sp->eRq, sp->eRp, sp->eRg, sp->ring_g,
                            sp->qf_book, run->front_p, run->out_position) &&
                  ((uintptr_t)sp->iw & 15) == 0 && ((uintptr_t)sp->bw & 15) == 0 && ((uintptr_t)sp->ack & 7) == 0 && ((uintptr_t)sp->ring_tag & 15) == 0 &&
                  ((uintptr_t)run->rec & 15) == 0;
  if (!ok) {
    bjx_set_error("%s: the speculative tail serves a diagonal metric, D %% 4 == 0, D <= 1024, 16-byte aligned "
                  "buffers, one-gradient integrators, an external callable and no per-chain adaptation; every "
                  "bjx_nuts_spec_t buffer must be given and ring a power of two >= 8", what);
    return 1;
  }
  return 0;
}
#define BJX_SPEC_LAUNCH(KERNEL, stream, ...)                                                        \
  do {                                                                                              \
    const dim3 g1((unsigned)spec->n_rows);                                                          \
    hipStream_t s_ = (hipStream_t)stream;                                                           \
    if (nuts->D <= 256) hipLaunchKernelGGL((KERNEL<1>), g1, dim3(64), 0, s_, __VA_ARGS__);          \
    else if (nuts->D <= 512) hipLaunchKernelGGL((KERNEL<2>), g1, dim3(64), 0, s_, __VA_ARGS__);     \
    else if (nuts->D <= 768) hipLaunchKernelGGL((KERNEL<3>), g1, dim3(64), 0, s_, __VA_ARGS__);     \
    else hipLaunchKernelGGL((KERNEL<4>), g1, dim3(64), 0, s_, __VA_ARGS__);                         \
  } while (0)
}  // namespace

int bjx_nuts_spec_enter(void* stream, const bjx_nuts_t* nuts, const bjx_nuts_async_t* run, const bjx_nuts_spec_t* spec) {
  if (check_spec(nuts, run, spec, "bjx_nuts_spec_enter")) return 1;
  BJX_SPEC_LAUNCH(k_nuts_spec_enter, stream, *nuts, *run, *spec);
  return bjx_check_launch("bjx_nuts_spec_enter");
}

int bjx_nuts_spec_integrate(void* stream, const bjx_nuts_t* nuts, const bjx_nuts_async_t* run,
                            const bjx_nuts_spec_t* spec, const float* logp_f, const float* gf, int32_t bump) {
  if (check_spec(nuts, run, spec, "bjx_nuts_spec_integrate")) return 1;
  BJX_CHECK_ARG(logp_f && gf && bjx_vec4_ok(nuts->D, gf), "bjx_nuts_spec_integrate: logp_f / gf (16-byte aligned) needed");
  BJX_SPEC_LAUNCH(k_nuts_spec_integrate, stream, *nuts, *run, *spec, logp_f, gf, (int)bump);
  return bjx_check_launch("bjx_nuts_spec_integrate");
}

int bjx_nuts_spec_book(void* stream, const bjx_nuts_t* nuts, const bjx_nuts_async_t* run,
                       const bjx_nuts_spec_t* spec, int32_t target, int32_t timeout_us) {
  if (check_spec(nuts, run, spec, "bjx_nuts_spec_book")) return 1;
  BJX_CHECK_ARG(timeout_us >= 0, "bjx_nuts_spec_book: timeout_us must be >= 0");
  const long long ticks = (long long)timeout_us * 100;  // the wall clock counts at 100 MHz
  BJX_SPEC_LAUNCH(k_nuts_spec_book, stream, *nuts, *run, *spec, (int)target, ticks);
  return bjx_check_launch("bjx_nuts_spec_book");
}

}  // extern "C"

#ifdef BJX_TICK_PROBE
extern "C" int bjx_debug_tick_probe(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(bjx_tick_probe), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(bjx_tick_probe), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
#endif  // !__HIPCC_RTC__
