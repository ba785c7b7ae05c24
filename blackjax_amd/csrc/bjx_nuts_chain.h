// NUTS (gfx950): the per-chain DEVICE functions every NUTS translation unit shares -- row helpers, scalar
// transcendentals, the opening half of a leapfrog, tree initialisation, the leaf ("post"), the merge.  Kernels and
// C-ABI entry points: bjx_nuts_lockstep.hip (lockstep tree driver), bjx_nuts_tick.hip (free-running ticks),
// bjx_nuts_spec.hip (two-stream speculative tail); free-running device functions: bjx_nuts_tick_dev.h.
// Split out of the former bjx_nuts.hip in round 6 (no behaviour change).
//
// One wavefront per chain row.  Per-chain control state lives in the fs / is slot tables so that
// every decision (direction, progressive sampling, divergence, U-turn) is wave-uniform.
// Row sweeps move 16 bytes per lane (VEC = 4) whenever D % 4 == 0 and the buffers are 16-byte
// aligned; the dense-metric paths keep the 4-byte mapping (VEC = 1) their shuffle-based
// matrix-vector product needs.  All sweeps of one instantiation use the same lane <-> element
// mapping, so a lane only ever re-reads elements it wrote itself within a kernel.
// Numerics contract as in bjx_device.h: explicit fmaf, fp64-accumulated reductions, fp64 scalar
// transcendentals rounded once.
#pragma once
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

}  // namespace
