// A whole HMC transition of every chain in ONE launch, for log-densities the ENGINE can evaluate itself
// (bjx_hmc_trajectory_diag, include/bjx_hip.h): momentum draw, L velocity-Verlet leapfrogs with the
// log-density and its gradient evaluated in registers, energies, Metropolis accept, select.
//
// This is outside the external-callable contract of the engine (blackjax.hmc calls logdensity_fn between
// every two leapfrogs: hmc.py:279-312 through integrators.py:104-150) -- it exists to show what that
// contract costs: with the target resident the chain's q, p, g and inverse mass row never leave the
// registers of its wave, so a leapfrog moves NO bytes; the launch is bound by the momentum draw's
// arithmetic and the few FMAs per element.  Arithmetic = the separate kernels', expression for expression
// (k_momentum_diag<4, true>, k_leapfrog_diag_flat<2>, the stand-alone target kernels, k_hmc_finish_diag<4>):
// the results are bit for bit those of the default path (tests/test_hmc_traj_gpu.py).
// One wave per chain, the row in NI 16-byte pieces per lane: 128 < D <= 1024, D % 4 == 0, diagonal metric.
#include "../../include/bjx_hip.h"
#include "../../include/bjx_nuts.h"  // BJX_TARGET_*
#include "bjx_device.h"
#include "bjx_host.h"
#include "bjx_traj_dev.h"

using namespace bjx;

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / BJX_WAVE;

template <int NI, class Target>
__global__ void __launch_bounds__(kBlock) k_hmc_trajectory_diag(TrajArgs a) {
  hmc_trajectory_rows<NI, Target>(a);
}

}  // namespace

extern "C" int bjx_hmc_trajectory_diag(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                                       int64_t step_fold, int64_t N, int64_t D, int64_t num_integration_steps,
                                       float eps, const float* eps_per_chain, const float* imm,
                                       int64_t imm_stride, float divergence_threshold, int32_t target_kind,
                                       const float* target_vec, const float* q0, const float* logp0,
                                       const float* g0, float* p0_out, float* q1_out, float* p_end_out,
                                       float* logp1_out, float* g1_out, float* q_out, float* logp_out,
                                       float* g_out, float* acceptance_rate_out, uint8_t* is_accepted_out,
                                       uint8_t* is_divergent_out, float* energy_out) {
  if (N == 0) return 0;  // empty batch: no buffers to check, nothing to do
  BJX_CHECK_ARG(N > 0 && num_integration_steps >= 0 && imm && q0 && logp0 && g0 && q_out && logp_out && g_out &&
                    acceptance_rate_out && is_accepted_out && is_divergent_out && energy_out,
                "bjx_hmc_trajectory_diag: bad arguments");
  BJX_CHECK_ARG(imm_stride == 0 || imm_stride == D, "bjx_hmc_trajectory_diag: imm_stride must be 0 or D");
  BJX_CHECK_ARG(D > 128 && D <= 1024 && D % 4 == 0,
                "bjx_hmc_trajectory_diag: rows of 132 ... 1 024 floats, a multiple of 4 (shorter rows take another "
                "reduction order in the separate kernels this launch must reproduce)");
  BJX_CHECK_ARG(target_kind == BJX_TARGET_NEAL_FUNNEL || (target_kind == BJX_TARGET_DIAG_GAUSSIAN && target_vec),
                "bjx_hmc_trajectory_diag: target_kind must name an engine-resident target (the diagonal Gaussian "
                "with target_vec = 1 / variance)");
  BJX_CHECK_ARG(q_out != q0 && g_out != g0, "bjx_hmc_trajectory_diag: outputs must not alias the initial state");
  BJX_CHECK_ARG(bjx_vec4_ok(D, imm, target_vec, q0, g0, p0_out, q1_out, p_end_out, g1_out, q_out, g_out),
                "bjx_hmc_trajectory_diag: rows must be 16-byte aligned");
  TrajArgs a{Key{key0, key1}, chain_offset, step_fold, N, D, num_integration_steps, eps, eps_per_chain, imm,
             imm_stride, divergence_threshold, target_vec, q0, logp0, g0, p0_out, q1_out, p_end_out,
             logp1_out, g1_out, q_out, logp_out, g_out, acceptance_rate_out, energy_out, is_accepted_out,
             is_divergent_out};
  const dim3 grid(bjx_row_grid(N, kWavesPerBlock)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
#define BJX_TRAJ(T_)                                                                        \
  do {                                                                                      \
    if (D <= 256) hipLaunchKernelGGL((k_hmc_trajectory_diag<1, T_>), grid, block, 0, s, a);   \
    else if (D <= 512) hipLaunchKernelGGL((k_hmc_trajectory_diag<2, T_>), grid, block, 0, s, a); \
    else hipLaunchKernelGGL((k_hmc_trajectory_diag<4, T_>), grid, block, 0, s, a);            \
  } while (0)
  if (target_kind == BJX_TARGET_NEAL_FUNNEL) BJX_TRAJ(FunnelTarget);
  else BJX_TRAJ(DiagGaussianTarget);
#undef BJX_TRAJ
  return bjx_check_launch("bjx_hmc_trajectory_diag");
}
