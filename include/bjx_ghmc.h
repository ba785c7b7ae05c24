/* libbjxhip C ABI, part 4: Generalized HMC (persistent momentum + non-reversible slice accept), the
 * sampler the reference's MEADS warm-up drives (blackjax/adaptation/meads_adaptation.py; the
 * reference recommends ChEES / MEADS for thousands of chains, howto_sample_multiple_chains.md:246).
 *
 * A transition is ONE velocity-Verlet step, so per chain and transition the engine runs
 *   bjx_ghmc_refresh_kick (= bjx_ghmc_refresh + bjx_leapfrog_diag with n_kicks = 1, bjx_hip.h)
 *   -> user callable -> bjx_ghmc_finish.
 * Every parameter may be per chain (MEADS hands every fold its own step size, scale, alpha, delta).
 * Only the per-dimension "inverse scale" form of ghmc's momentum metric is built (ghmc.py:67-86
 * legacy branch: inverse mass matrix = scale ** 2, squared by the caller).
 *
 * Conventions as in bjx_hip.h: device pointers, row-major (N, D) fp32, explicit stream, int status.
 * Chain i uses the key split(key, .)[chain_offset + i] (step_fold as in bjx_hmc_momentum_diag).
 */
#ifndef BJX_GHMC_H
#define BJX_GHMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ghmc.init (blackjax/mcmc/ghmc.py:53-64):  km, ks = split(k_i);
 *   momentum[i] = normal(km, (D,))  (generate_gaussian_noise with mu = 0, sigma = 1, util.py:66-91)
 *   slice[i]    = uniform(ks, (), minval = -1, maxval = 1)                                      */
int bjx_ghmc_init(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset, int64_t N,
                  int64_t D, float* momentum_out, float* slice_out);

/* First half of ghmc.build_kernel.kernel (ghmc.py:168-176) + the kinetic energy of the refreshed
 * state:  km, _ = split(k_i)
 *   p[i]     = p_prev[i] * sqrt(1 - alpha_i) + sqrt(alpha_i) * ((1/sqrt(imm_i)) * normal(km, (D,)))
 *              (update_momentum 203-223 over metric.sample_momentum, metrics.py:260-261)
 *   slice[i] = ((slice_prev[i] + 1 + delta_i + 0) % 2) - 1          (noise_fn = 0)
 *   ke[i]    = 0.5 * dot(imm_i * p[i], p[i])                        (fp64 accumulate)
 * alpha / delta: per-chain arrays or NULL (then the scalars are used).  Out of place.            */
int bjx_ghmc_refresh(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                     int64_t step_fold, int64_t N, int64_t D, const float* imm, int64_t imm_stride,
                     float alpha, const float* alpha_per_chain, float delta,
                     const float* delta_per_chain, const float* p_prev, const float* slice_prev,
                     float* p_out, float* slice_out, float* ke_out);

/* bjx_ghmc_refresh followed by bjx_leapfrog_diag(n_kicks = 1) in one launch (same arithmetic, same
 * results): additionally  p_half = p + (eps_i / 2) g0 ;  q1 = q0 + eps_i * (imm_i * p_half)
 * (integrators.py:104-150, first half of velocity Verlet).  The refresh is bound by its per-element
 * RNG arithmetic, so the kick + drift traffic rides along.                                        */
int bjx_ghmc_refresh_kick(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                          int64_t step_fold, int64_t N, int64_t D, const float* imm, int64_t imm_stride,
                          float alpha, const float* alpha_per_chain, float delta,
                          const float* delta_per_chain, float eps, const float* eps_per_chain,
                          const float* p_prev, const float* slice_prev, const float* q0, const float* g0,
                          float* p_out, float* slice_out, float* ke_out, float* q1_out, float* p_half_out);

/* Second half (hmc.py:153-176 with L = 1, proposal.py:243-264, ghmc.py:186-196), after
 * bjx_leapfrog_diag(n_kicks = 1) produced (q1, p_half) and the callable (logp1, g1):
 *   p1 = p_half + (eps_i / 2) g1 ;  H0 = -logp0 + ke0 ;  H1 = -logp1 + 0.5 dot(imm p1, p1)
 *   dE = H0 - H1 (NaN -> -inf) ; is_divergent = -dE > threshold ; acceptance_rate = min(exp(dE), 1)
 *   accept = log|slice| <= dE ; slice' = slice * (exp(-dE) * accept + (1 - accept))
 *   state' = accept ? (q1, +p1, logp1, g1) : (q0, -p, logp0, g0)     (the two momentum flips of
 *   hmc.py:157 and ghmc.py:188 combined)
 * p = refreshed momentum, slice = refreshed slice (bjx_ghmc_refresh).  Chains in
 * [skip_begin, skip_end) do not move (MEADS freezes the fold t mod K, meads_adaptation.py:664-677):
 * their outputs are (q0, p_prev, logp0, g0, slice_prev) and their info entries are still those of
 * the computed proposal, as in the reference.  p_end_out (optional): momentum of the proposal's end
 * state, -p1 (HMCInfo.proposal).                                                                 */
int bjx_ghmc_finish(void* stream, int64_t N, int64_t D, float eps, const float* eps_per_chain,
                    const float* imm, int64_t imm_stride, float divergence_threshold,
                    const float* q0, const float* logp0, const float* g0, const float* ke0,
                    const float* p, const float* slice, const float* p_prev, const float* slice_prev,
                    const float* q1, const float* p_half, const float* logp1, const float* g1,
                    int64_t skip_begin, int64_t skip_end, float* q_out, float* p_out, float* logp_out,
                    float* g_out, float* slice_out, float* acceptance_rate_out, uint8_t* is_accepted_out,
                    uint8_t* is_divergent_out, float* energy_out, float* p_end_out);

/* ---- MEADS fold statistics (blackjax/adaptation/meads_adaptation.py:560-640, 790-817) ------------
 * The per-step quantities of the K-fold cross-chain adaptation as stream-ordered launches (no host
 * synchronisation): chains are fold-major, fold k = rows [k n, (k + 1) n) of the (N = K n, D) arrays.
 *   bjx_meads_fold_moments : mean_out, sd_out (K, D) = per-fold mean / population std of x (jnp.std,
 *                            ddof 0; fp64 sums of x - x_first), whitened_mean_out (K, D) = fp64 mean of
 *                            x / sd_k (the centring of fold_damping, 604-606)
 *   bjx_meads_fold_build   : A = g * sd_k (583-588), B = x / sd_k - whitened_mean_k (616-622), both
 *                            (N, D) fp32; rowsq (2, N) doubles = row sums of squares of A and B
 *   -- the caller forms the Gram matrices of A_k and B_k with a plain library GEMM (fp32), laid out
 *      gram[mtx][k] with gram_elems floats each (D x D or n x n, |X X^T|_F = |X^T X|_F) --
 *   bjx_meads_fold_params  : maximum_eigenvalue of every matrix (812-817: (sum S^2 - sum diag S^2) /
 *                            (n (n - 1)) / (sum diag S / n)), step size min(multiplier / sqrt(lambda_A), 1)
 *                            of fold k - 1 -> fold k, gamma = max(1 / sqrt(lambda_B), slowdown /
 *                            ((t + 1) eps)), alpha = 1 - exp(-2 eps gamma) (fp64 exp, rounded once),
 *                            delta = alpha / 2, sigma_fold (K, D) = sd rolled by one fold; per-chain
 *                            broadcasts eps_pc / alpha_pc / delta_pc (N,) and imm_pc (N, D) = the SQUARED
 *                            rolled scale (ghmc.py:67-86: inverse mass = scale ** 2).
 * workspace: bjx_meads_workspace_bytes(K, D) bytes, shared by the three calls of one step. */
size_t bjx_meads_workspace_bytes(int64_t K, int64_t D);
int bjx_meads_fold_moments(void* stream, int64_t K, int64_t n, int64_t D, const float* x, void* workspace,
                           float* mean_out, float* sd_out, float* whitened_mean_out);
int bjx_meads_fold_build(void* stream, int64_t K, int64_t n, int64_t D, const float* x, const float* g,
                         const float* sd, const float* whitened_mean, float* A, float* B, double* rowsq);
int bjx_meads_fold_params(void* stream, int64_t K, int64_t n, int64_t D, int64_t t, float step_size_multiplier,
                          float damping_slowdown, int64_t gram_elems, const float* gram, const double* rowsq,
                          const float* sd, void* workspace, float* eps_fold, float* alpha_fold, float* delta_fold,
                          float* sigma_fold, float* eps_pc, float* alpha_pc, float* delta_pc, float* imm_pc);

#ifdef __cplusplus
}
#endif
#endif
