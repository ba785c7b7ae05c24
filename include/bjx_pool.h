/* libbjxhip C ABI, part 3: pooled (cross-chain) statistics for many-chain warmup.
 *
 * ChEES-HMC adaptation (blackjax/adaptation/chees_adaptation.py) tunes
 * ONE step size and ONE trajectory length from statistics of the whole ensemble.  Everything that
 * touches an (N, D) array runs here; the scalar recursions (dual averaging, Adam, moving averages)
 * are host code in blackjax_amd/chees.py.  This is the only part of the engine with a real
 * exchange step under chain sharding: the fp64 statistics written by bjx_chees_colstats,
 * bjx_chees_scalars and bjx_pool_colsum are SUMS over chains, so ranks all-reduce(sum) them
 * (RCCL) between the calls below; every other argument is rank-local.
 *
 * Conventions as in bjx_hip.h: device pointers, row-major (N, D) fp32, explicit stream, int status,
 * no allocation (workspace sized by bjx_pool_workspace_bytes), fp64 accumulation of every sum.
 */
#ifndef BJX_POOL_H
#define BJX_POOL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bytes of scratch the column reductions below need for an (N, D) batch */
int64_t bjx_pool_workspace_bytes(int64_t N, int64_t D);

/* chees_adaptation.py:376 + weighted_empirical_mean 241-242:
 *   w[n] = (is_divergent[n] || any non-finite in q_prop[n, :]) ? 0 : acc[n]                    */
int bjx_chees_weights(hipStream_t stream, int64_t N, int64_t D, const float* q_prop, const float* acc,
                      const uint8_t* is_divergent, float* w);

/* Column statistics of one ensemble step (chees_adaptation.py:377-386), stats = 4*D doubles:
 *   stats[0*D+d] = sum_n w[n] * finite0(q_prop[n,d])        (numerator of the weighted mean, 245)
 *   stats[1*D+d] = sum_n nan0(q_init[n,d])                  (jnp.nanmean numerator, 384-386)
 *   stats[2*D+d] = #{n : q_init[n,d] is not NaN}
 *   stats[3*D+d] = sum_n w[n]                               (same value for every d)
 * Sums over chains: all-reduce(sum) across ranks before bjx_chees_means.                          */
int bjx_chees_colstats(hipStream_t stream, int64_t N, int64_t D, const float* q_prop, const float* w,
                       const float* q_init, void* workspace, double* stats);

/* bjx_chees_weights followed by bjx_chees_colstats in one pass over q_prop (the threads holding a row
 * between them decide its weight before accumulating it); same results bit for bit.  Falls back to
 * the two launches when a row does not fit one workgroup (D > 1024).  w is still written.        */
int bjx_chees_weights_colstats(hipStream_t stream, int64_t N, int64_t D, const float* q_prop,
                               const float* acc, const uint8_t* is_divergent, const float* q_init,
                               float* w, void* workspace, double* stats);

/* proposals_mean[d] = f32(stats0)/(f32(stats3) + 1e-20) (246-247); initials_mean[d] = f32(stats1)/f32(stats2);
 * inv_sqrt_imm[d] = 1/sqrt(imm[d]) (450) when imm != NULL (inv_sqrt_imm may be NULL otherwise).    */
int bjx_chees_means(hipStream_t stream, int64_t D, const double* stats, const float* imm,
                    float* proposals_mean, float* initials_mean, float* inv_sqrt_imm);

/* Per-chain ChEES factor (chees_adaptation.py:387-466):
 *   crit[n] = (|dx'|^2 - |dx|^2) * <dx', v'>,  dx' = q_prop - proposals_mean, dx = q_init - initials_mean,
 * whitened by the diagonal metric when imm != NULL: dx*inv_sqrt_imm, v' = (p_prop*imm)*inv_sqrt_imm
 * (450-454); raw (v' = p_prop) when imm == NULL (identical to imm = ones, 440-442).               */
int bjx_chees_criterion(hipStream_t stream, int64_t N, int64_t D, const float* q_prop,
                        const float* p_prop, const float* q_init, const float* proposals_mean,
                        const float* initials_mean, const float* imm, const float* inv_sqrt_imm,
                        float* crit);

/* Ensemble scalars over the non-divergent chains (chees_adaptation.py:358-360, 468-471), out = 4 doubles:
 *   out[0] = sum 1/acc[n]   out[1] = #{non-divergent}
 *   out[2] = sum acc[n] * f32(scale * crit[n])   out[3] = sum f32(acc[n] + 1e-20)
 * (crit may be NULL: out[2] = 0).  Sums over chains: all-reduce(sum) across ranks.               */
int bjx_chees_scalars(hipStream_t stream, int64_t N, const float* acc, const uint8_t* is_divergent,
                      const float* crit, float scale, double* out);

/* Column sums of a batch of draws (cgl_update_batch, blackjax/adaptation/metric_buffers.py:428-433), out = D doubles:
 *   center == NULL: out[d] = sum_n x[n,d]
 *   center != NULL: out[d] = sum_n f32(x[n,d] - center[d])^2
 * Sums over chains: all-reduce(sum) across ranks.                                                */
int bjx_pool_colsum(hipStream_t stream, int64_t N, int64_t D, const float* x, const float* center,
                    void* workspace, double* out);

/* mean[d] = f32(sum[d] / count)  (metric_buffers.py:429) */
int bjx_pool_mean(hipStream_t stream, int64_t D, const double* sum, double count, float* mean);

/* In-place CGL merge of a diagonal moment block (count n_a, mean, m2) with a batch
 * (count n_b, mean_b, m2_b = f32(m2_b_sum)) -- metric_buffers.py:437-451.                          */
int bjx_pool_merge_diag(hipStream_t stream, int64_t D, float n_a, float n_b, const float* mean_b,
                        const double* m2_b_sum, float* mean, float* m2);

/* imm[d] = max(m2[d] / (count - 1), 1e-20)  (chees_adaptation.py:83-90; mass_matrix.py:437-442) */
int bjx_pool_final_diag(hipStream_t stream, int64_t D, float count, const float* m2, float* imm);

/* centered[n,d] = x[n,d] - center[d] (metric_buffers.py:430), input of the dense m2_b = centered^T centered */
int bjx_pool_center(hipStream_t stream, int64_t N, int64_t D, const float* x, const float* center,
                    float* centered);

/* Halton trajectory jitter (blackjax/mcmc/dynamic_hmc.py:205-215 + chees_adaptation.py:762-771):
 *   steps[n] = ceil((halton(arg[n], max_bits) * jitter_amount + jitter_offset) * num_leapfrog_steps)
 * with jitter_offset = f32(1 - jitter_amount) formed by the caller in double.                      */
int bjx_halton_steps(hipStream_t stream, int64_t N, const int32_t* arg, int32_t max_bits,
                     float jitter_amount, float jitter_offset, float num_leapfrog_steps,
                     int32_t* steps);

#ifdef __cplusplus
}
#endif
#endif /* BJX_POOL_H */
