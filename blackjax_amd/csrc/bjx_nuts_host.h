// Host-side helpers shared by the NUTS translation units: descriptor checks, the 16-byte-sweep predicates and the
// <VEC, DENSE> launch macros.
#pragma once
#ifndef __HIPCC_RTC__
namespace {
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

}  // namespace
#endif
