// NUTS lockstep tree driver (gfx950): init / pre / mid / post / merge kernels over the live rows, device-side
// compaction and the control block; C ABI in include/bjx_nuts.h.  Per-chain device functions: bjx_nuts_chain.h.
#include "bjx_nuts_chain.h"
#include "bjx_nuts_host.h"

namespace {

template <int VEC, bool DENSE>
__global__ void __launch_bounds__(kBlock)
k_nuts_init(bjx_nuts_t nt, const float* __restrict__ logp0, const float* __restrict__ ke0) {
  for (int64_t c = wave_row0(); c < nt.N; c += wave_row_stride())
    nuts_init_chain<VEC, DENSE>(nt, c, logp0[c], ke0[c]);
}

// Start of doubling `depth` for chain c: draw the direction and reset the subtree flags
// (trajectory.py:645-650).  Returns the direction (+1 / -1).

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

}  // namespace

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

}  // extern "C"
