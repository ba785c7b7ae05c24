// NUTS free-running chains (gfx950): the DEVICE functions of the tick kernels -- the general one-launch tick
// (async_leaf_chain / async_boundary_chain), the record-based lean leaf and transition end (async_leaf2_chain,
// async_end2_chain, async_multi_tick_row for engine-resident targets) and the v3 leaf (async_leaf3_row).  Shared by
// bjx_nuts_tick.hip, bjx_nuts_spec.hip and the hiprtc unit of blackjax_amd/rtc.py (device code only there).
#pragma once
#include "bjx_nuts_chain.h"

namespace {

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

#endif  // !__HIPCC_RTC__
}  // namespace
