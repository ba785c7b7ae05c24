// NUTS free-running ticks (gfx950): the tick kernels (general one-launch tick, engine-resident multi-tick, the
// lean v3 tick of the contract path, the shared-dense GEMM mode), row compaction, and bjx_nuts_async_tick /
// bjx_nuts_async_compact.  Device functions: bjx_nuts_tick_dev.h; C ABI in include/bjx_nuts.h.
#include "bjx_nuts_tick_dev.h"
#include "bjx_nuts_host.h"

namespace {

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

}  // namespace

extern "C" {

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
