/*
 * bjx_hip.h -- C ABI of libbjxhip.so, the MI355X (gfx950) engine behind
 * blackjax_amd.hmc / .nuts / .window_adaptation.
 *
 * The reference (blackjax-devs/blackjax) has no FFI: its hot path is a Python
 * protocol (blackjax/base.py:88-113, SamplingAlgorithm(init, step)).  This header
 * is the boundary a maintainer would bind (ctypes stub in INTEGRATION.md); every
 * entry point cites the reference function(s) whose arithmetic it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every function returns 0 on success, non-zero on failure;
 *     bjx_last_error() returns a thread-local message for the last failure.
 *   - all array pointers are DEVICE pointers owned by the caller; nothing is
 *     allocated inside a call; calls are asynchronous on `stream` (a hipStream_t
 *     passed as void*), re-entrant, no global mutable state.
 *   - chain-major row layout: an (N, D) array is N rows of D contiguous fp32.
 *   - `key` arguments are threefry keys passed BY VALUE as two uint32 words
 *     (jax.random key data); per-chain keys are derived in-kernel so sharded runs need
 *     no exchange.  Two layouts, selected by `step_fold`:
 *       step_fold <  0  "step-major":  k_i = split(key, .)[chain_offset+i]
 *                       (key = this step's key; vmap inside the step)
 *       step_fold >= 0  "chain-major": k_i = split(split(key, .)[chain_offset+i], .)[step_fold]
 *                       (key = the run key, step_fold = t; vmap over whole per-chain loops,
 *                       e.g. a vmapped window_adaptation(...).run)
 *   - `eps` (step size): if `eps_per_chain` != NULL it is a device (N,) array,
 *     otherwise the scalar `eps` is used for every chain.
 *   - `imm` (diagonal inverse mass matrix): row stride `imm_stride` = 0 for one
 *     shared (D,) vector, = D for a per-chain (N, D) array.
 */
#ifndef BJX_HIP_H
#define BJX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BJX_ABI_VERSION 7 /* 2: bjx_nuts_t gained int_kick / int_drift (round 3); 3: bjx_nuts_async_t gained int_stages /
                             int_mid_kick / int_mid_drift, bjx_rng_key_probe added (round 4); 4: bjx_nuts_async_t gained
                             gemm_pc .. gemm_cap (round 4); 5 (round 5): same entry points and layouts -- bjx_nuts_async_tick now
                             has ONE kernel per shape and reads no environment switch (an engine-resident target is ticked
                             ticks_per_launch >= 1 times per launch whatever the batch size; run->end_list / end_count are
                             only used in GEMM mode), multi-stage integrators tick free for rows of up to 1 024 floats, and the
                             shared-dense entry points REFUSE whole 128 x 128 tiles on buffers that are not 16-byte aligned
                             instead of reading imm transposed; 6 (round 5): bjx_nuts_spec_t and bjx_nuts_spec_enter / _integrate /
                             _book added (two-stream speculative tail of a free-running run), nothing else changed; 7 (round 6):
                             bjx_log1p_device_check added; the NUTS `is` table gained the slot BJX_NUTS_I_STAGE
                             (BJX_NUTS_NI 17 -> 18: multi-stage integrators on the general free-running tick kernel) */

const char* bjx_last_error(void);
int bjx_abi_version(void);

/* jax.random.split(key, n)[offset : offset+n] on the host (no device work).
 * Replaces: jax.random.split at blackjax/util.py:203, adaptation/staged_adaptation.py:868.
 * out: host uint32[n][2]. */
int bjx_keys_split(uint32_t key0, uint32_t key1, int64_t n, int64_t offset, uint32_t* out);

/* Debug/parity probes of the in-kernel RNG (device): z[i][j] = jax.random.normal(
 * split(key, .)[chain_offset+i] , (D,))[j]  and  u[i] = jax.random.uniform(same key, ()).
 * Replaces: jax.random.normal at blackjax/util.py:90; jax.random.uniform inside bernoulli
 * at blackjax/mcmc/proposal.py:226. */
int bjx_rng_normal(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                   int64_t N, int64_t D, float* z_out);
int bjx_rng_uniform(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                    int64_t N, float* u_out);
/* The same device functions with the key used AS IS (no per-chain child key):
 *   z_out[j] = jax.random.normal(key, (D,))[j] ; u_out[0] = jax.random.uniform(key, ()) (may be NULL) ;
 *   children_out[i][0..1] = jax.random.split(key, n_children)[i]  (device uint32[n_children][2]).
 * The probe the jax.random values printed in JAX's documentation are held against (tests/golden/
 * reference_kats.json "jax_docs_streams").  Replaces: jax.random.normal / uniform / split as used at
 * blackjax/util.py:90, mcmc/proposal.py:226, mcmc/hmc.py:299. */
int bjx_rng_key_probe(void* stream, uint32_t key0, uint32_t key1, int64_t D, float* z_out, float* u_out,
                      int64_t n_children, uint32_t* children_out);

/* Device self-check of the correctly-rounded fp32 -log1p inside jax.random.normal's erf_inv (csrc/bjx_log1p.h): every
 * stride-th fp32 t in (-1, 0] through the product's function against the device library's fp64 log1p rounded once.
 * counts_out: device uint64[4] = {inputs checked, results that differ, inputs resolved by the slow table, bit
 * pattern of the first differing input}.  stride 1 = exhaustive (1 065 353 217 inputs, ~0.1 s).
 * Checks the arithmetic behind: jax.random.normal as used at blackjax/util.py:88-91. */
int bjx_log1p_device_check(void* stream, uint32_t stride, unsigned long long* counts_out);

/* Momentum draw for a diagonal metric + initial kinetic energy.
 *   k_i = split(key, .)[chain_offset+i]; km = split(k_i, 2)[0]
 *   p0[i] = (1/sqrt(imm)) * normal(km, (D,)) ;  ke0[i] = 0.5 * dot(imm*p0[i], p0[i])
 * Replaces: blackjax/mcmc/hmc.py:299,302 ; metrics.py:260-261,263-270,704-709 ; util.py:66-91. */
int bjx_hmc_momentum_diag(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                          int64_t step_fold, int64_t N, int64_t D, const float* imm, int64_t imm_stride,
                          float* p_out, float* ke_out);

/* bjx_hmc_momentum_diag followed by bjx_leapfrog_diag(n_kicks = 1) in ONE launch (same arithmetic,
 * same results): additionally  p_half = p0 + (eps_i / 2) g0 ;  q1 = q0 + eps_i * (imm_i * p_half)
 * (integrators.py:104-150, first half of the trajectory's first velocity-Verlet step).  The momentum
 * draw is bound by its per-element RNG arithmetic, so the first kick + drift ride along.          */
int bjx_hmc_momentum_kick_diag(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                               int64_t step_fold, int64_t N, int64_t D, const float* imm, int64_t imm_stride,
                               float eps, const float* eps_per_chain, const float* q0, const float* g0,
                               float* p_out, float* ke_out, float* q1_out, float* p_half_out);

/* Fused velocity-Verlet "kick(s) + drift" for a diagonal metric:
 *   n_kicks = 1:  p = p + (eps/2) g                       (first step of a trajectory)
 *   n_kicks = 2:  p = (p + (eps/2) g) + (eps/2) g         (closing half kick of the previous
 *                                                          step + opening half kick of this one;
 *                                                          two separately rounded fmas)
 *   v = imm * p ;  q = q + eps * v
 * reads p_in, g, q_in ; writes p_out, q_out (may alias the inputs).
 * Replaces: blackjax/mcmc/integrators.py:104-150 (one_step), 191-205, 226-243 with
 * coefficients [0.5, 1.0, 0.5] (321-322), driven by trajectory.py:155-165. */
int bjx_leapfrog_diag(void* stream, int64_t N, int64_t D, int n_kicks, float eps,
                      const float* eps_per_chain, const float* imm, int64_t imm_stride,
                      const float* q_in, const float* p_in, const float* g, float* q_out,
                      float* p_out);

/* Same as bjx_leapfrog_diag with a per-chain trajectory length (dynamic HMC, SURVEY.md section 8f
 * row 2): chain i is advanced only while step_idx < n_steps[i] (n_steps: device (N,) int32);
 * otherwise its (q, p) is left untouched (copied through when the launch is out of place).
 * Replaces: the per-chain `num_integration_steps` of blackjax/mcmc/dynamic_hmc.py:85-118 under vmap. */
int bjx_leapfrog_diag_masked(void* stream, int64_t N, int64_t D, int n_kicks, float eps,
                             const float* eps_per_chain, const float* imm, int64_t imm_stride,
                             const float* q_in, const float* p_in, const float* g, float* q_out,
                             float* p_out, const int32_t* n_steps, int32_t step_idx);

/* General palindromic integrators: coefficients
 * [b1, a1, b2, a2, ..., b1] of generalized_two_stage_integrator (blackjax/mcmc/integrators.py:
 * 104-150; mclachlan / yoshida / omelyan 335-369).  One launch per position update:
 *   p = p + (eps*kick_a) g  [ ; p = p + (eps*kick_b) g  if n_kicks == 2 ]
 *   q = q + (eps*drift) * (imm * p)
 * `eps*coef` is an fp32 product like the reference's `step_size * coef`.  Velocity Verlet is
 * (kick_a, kick_b, drift) = (0.5, 0.5, 1.0), i.e. bjx_leapfrog_diag.  n_steps/step_idx as in
 * bjx_leapfrog_diag_masked (NULL = every chain advances). */
int bjx_leapfrog_diag_coef(void* stream, int64_t N, int64_t D, int n_kicks, float kick_a,
                           float kick_b, float drift, float eps, const float* eps_per_chain,
                           const float* imm, int64_t imm_stride, const float* q_in,
                           const float* p_in, const float* g, float* q_out, float* p_out,
                           const int32_t* n_steps, int32_t step_idx);

/* bjx_hmc_finish_diag with the closing kick p1 = p + (eps*kick_coef) g1, kick_coef = last
 * coefficient of the palindromic integrator (0.5 for velocity Verlet). */
int bjx_hmc_finish_diag_coef(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                             int64_t step_fold, int64_t N, int64_t D, float kick_coef, float eps,
                             const float* eps_per_chain, const float* imm, int64_t imm_stride,
                             float divergence_threshold, const float* q0, const float* logp0,
                             const float* g0, const float* ke0, const float* q1, const float* logp1,
                             const float* g1, const float* p, float* p_end_out, float* q_out,
                             float* logp_out, float* g_out, float* acceptance_rate_out,
                             uint8_t* is_accepted_out, uint8_t* is_divergent_out, float* energy_out);

/* Per-chain key utilities on the device (keys: (N, 2) uint32 jax.random key data).
 *   bjx_keys_child:   keys_out[i] = split(keys_in[i], .)[child]   (default next_random_arg_fn of
 *                     dynamic_hmc.py:69: `lambda key: jax.random.split(key)[1]`)
 *   bjx_keys_randint: out[i] = jax.random.randint(keys[i], (), minval, maxval) as int32 (default
 *                     integration_steps_fn of dynamic_hmc.py:70: randint(key, (), 1, 10)).
 * jax.random.randint is restated from jax/_src/random.py::_randint (two 32-bit draws from
 * split(key, 2), offset = ((hi % span) * (2^32 % span) + lo % span) % span). */
int bjx_keys_child(void* stream, int64_t N, const uint32_t* keys_in, uint32_t child,
                   uint32_t* keys_out);
int bjx_keys_randint(void* stream, int64_t N, const uint32_t* keys, int32_t minval, int32_t maxval,
                     int32_t* out);

/* Closing half kick + flip + energies + Metropolis accept + state select (diag metric).
 *   p1 = p + (eps/2) g1 ; p_end = -p1 ; ke1 = 0.5 dot(imm*p1, p1)
 *   H0 = -logp0 + ke0 ; H1 = -logp1 + ke1 ; delta = H0 - H1 (NaN -> -inf)
 *   is_divergent = -delta > divergence_threshold ; p_acc = min(1, exp(delta))
 *   ki = split(split(key,.)[chain_offset+i], 2)[1] ; accept = uniform(ki) < p_acc
 *   (q,logp,g)_out = accept ? (q1,logp1,g1) : (q0,logp0,g0)
 * p_end_out may be NULL (HMCInfo.proposal.momentum not wanted) or alias p.
 * Replaces: blackjax/mcmc/hmc.py:95-112,153-176 ; trajectory.py:730-750 ;
 * proposal.py:45-48,214-235. */
int bjx_hmc_finish_diag(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                        int64_t step_fold, int64_t N, int64_t D, float eps, const float* eps_per_chain,
                        const float* imm, int64_t imm_stride, float divergence_threshold,
                        const float* q0, const float* logp0, const float* g0, const float* ke0,
                        const float* q1, const float* logp1, const float* g1, const float* p,
                        float* p_end_out, float* q_out, float* logp_out, float* g_out,
                        float* acceptance_rate_out, uint8_t* is_accepted_out,
                        uint8_t* is_divergent_out, float* energy_out);

/* ---- multinomial HMC (blackjax.mhmc) ----------------------------
 * One step of static_progressive_integration fused with the opening half of the next leapfrog.
 * On entry p holds the momentum after the OPENING half kick of leapfrog `step` and (q, g,
 * logp_new) the new position and the callable's outputs there.  The kernel applies the closing
 * half kick, forms the proposal weight w = H0 - H (NaN -> -inf), any_divergent |= -w > threshold,
 * draws u = uniform(fold_in(key_integrator, step)), accepts the new state into the reservoir
 * (prop_*) with probability expit(w - weight), updates weight and sum_log_p_accept by logaddexp,
 * and, if do_next, performs the opening half kick + drift of leapfrog step+1 in place on (q, p).
 * weight / sum_log_p_accept: (N,) initialised to 0 / -inf; ever_accepted / any_divergent: (N,)
 * uint8 initialised to 0.  H0 = -logp0 + ke0.
 * Replaces: blackjax/mcmc/hmc.py:181-248 ; trajectory.py:170-232 ; proposal.py:51-105,118-143. */
int bjx_mhmc_step_diag(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                       int64_t step_fold, int64_t N, int64_t D, int64_t step, int do_next, float eps,
                       const float* eps_per_chain, const float* imm, int64_t imm_stride,
                       float divergence_threshold, const float* logp0, const float* ke0, float* q,
                       float* p, const float* g, const float* logp_new, float* weight,
                       float* sum_log_p_accept, uint8_t* any_divergent, uint8_t* ever_accepted,
                       float* prop_q, float* prop_p, float* prop_g, float* prop_logp,
                       float* prop_energy);

/* End of a multinomial-HMC transition: chains whose reservoir never replaced the initial state get
 * (q0, p0, g0, logp0, H0) copied into prop_*; acceptance_rate = exp(sum_log_p_accept) / L
 * (hmc.py:234).  prop_(q, logp, g) is the new chain state. */
int bjx_mhmc_finish(void* stream, int64_t N, int64_t D, int64_t num_integration_steps,
                    const float* q0, const float* p0, const float* g0, const float* logp0,
                    const float* ke0, const uint8_t* ever_accepted, const float* sum_log_p_accept,
                    float* prop_q, float* prop_p, float* prop_g, float* prop_logp, float* prop_energy,
                    float* acceptance_rate_out);

/* The same two entry points with one trajectory length PER CHAIN (blackjax.dmhmc =
 * dynamic_hmc.build_kernel(build_proposal=multinomial_hmc_proposal), blackjax/__init__.py:155-163;
 * dynamic_hmc.py:85-118 under vmap): chain i takes part in step `step` only while step < n_steps[i]
 * (device (N,) int32), its last step opens no further leapfrog, and its acceptance rate is
 * exp(sum_log_p_accept[i]) / n_steps[i]. */
int bjx_mhmc_step_diag_masked(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                              int64_t step_fold, int64_t N, int64_t D, int64_t step, int do_next, float eps,
                              const float* eps_per_chain, const float* imm, int64_t imm_stride,
                              float divergence_threshold, const float* logp0, const float* ke0, float* q,
                              float* p, const float* g, const float* logp_new, float* weight,
                              float* sum_log_p_accept, uint8_t* any_divergent, uint8_t* ever_accepted,
                              float* prop_q, float* prop_p, float* prop_g, float* prop_logp,
                              float* prop_energy, const int32_t* n_steps);
/* bjx_mhmc_step_diag[_masked] for any palindromic integrator [b_1, a_1, ..., b_1] (blackjax.mhmc /
 * dmhmc with integrator=mclachlan / yoshida / omelyan; blackjax/mcmc/integrators.py:335-369 through
 * trajectory.py:170-232): the closing kick of the step and -- with do_next -- the opening kick of the
 * next one are (eps * kick_coef) g, the drift that follows is (eps * drift_coef) imm p; the stages in
 * between (b_2, a_2 ...) are bjx_leapfrog_diag_coef launches with n_kicks = 1.  n_steps may be NULL
 * (every chain advances).  (0.5, 1.0) reproduces bjx_mhmc_step_diag bit for bit. */
int bjx_mhmc_step_diag_coef(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                            int64_t step_fold, int64_t N, int64_t D, int64_t step, int do_next, float eps,
                            const float* eps_per_chain, const float* imm, int64_t imm_stride,
                            float divergence_threshold, const float* logp0, const float* ke0, float* q,
                            float* p, const float* g, const float* logp_new, float* weight,
                            float* sum_log_p_accept, uint8_t* any_divergent, uint8_t* ever_accepted,
                            float* prop_q, float* prop_p, float* prop_g, float* prop_logp,
                            float* prop_energy, const int32_t* n_steps, float kick_coef, float drift_coef);

int bjx_mhmc_finish_masked(void* stream, int64_t N, int64_t D, const int32_t* n_steps, const float* q0,
                           const float* p0, const float* g0, const float* logp0, const float* ke0,
                           const uint8_t* ever_accepted, const float* sum_log_p_accept, float* prop_q,
                           float* prop_p, float* prop_g, float* prop_logp, float* prop_energy,
                           float* acceptance_rate_out);

/* ---- dense Gaussian-Euclidean metric (one (D, D) inverse mass matrix shared by all chains) ----
 * fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32, exact fp32 fma chains: "precision=highest",
 * blackjax/util.py:23-61).  All matrices row-major.
 *
 * C = A @ B with A (N, D), B (D, D): the building block, exported for parity tests. */
int bjx_dense_matmul(void* stream, int64_t N, int64_t D, const float* A, const float* B, float* C);

/* V = P imm^T -- v_r = imm p_r for every row, the matrix read AS STORED (v[i] = sum_k imm[i][k] p[k],
 * blackjax/mcmc/metrics.py:263-304 `linear_map`) -- given imm AND its transpose imm_t (both (D, D) row-major).
 * bjx_dense_apply_imm with three differences: few rows (N D <= BJX_DENSE_SKINNY_MAX, default 2^19) run on a
 * latency-oriented kernel (a few microseconds instead of 18-28), ragged shapes use imm_t on the general kernel --
 * so a matrix that is symmetric only up to rounding (a Welford covariance) gives the same product at every batch
 * size -- and whole aligned tiles run on the kernel bjx_dense_apply_imm uses.  One ascending-k fp32 fma chain per
 * output element in the k order of the MFMA kernels everywhere: identical results.  (round 4) */
int bjx_dense_apply_imm_t(void* stream, int64_t N, int64_t D, const float* P, const float* imm, const float* imm_t,
                          float* V);

/* C = A B like bjx_dense_matmul, given B AND its transpose Bt (both (D, D) row-major): whole 128-row / 128-column
 * tiles of 16-byte aligned buffers run on the kernel that reads its matrix as stored (the one bjx_dense_apply_imm uses
 * for a symmetric matrix, faster), few rows (N D <= BJX_DENSE_SKINNY_MAX) on the latency-oriented kernel of
 * bjx_dense_apply_imm_t, anything else on bjx_dense_matmul's kernel.  Same ascending-k fp32 fma chain per
 * output element either way: identical results.  (round 4) */
int bjx_dense_matmul_bt(void* stream, int64_t N, int64_t D, const float* A, const float* B, const float* Bt, float* C);

/* V = P @ imm^T for N rows: v_i = imm p_i (linear_map(inverse_mass_matrix, p), util.py:58-61;
 * metrics.py:263-304) on the fp32 MFMA GEMM, imm read as the reference stores it (row n = output n).
 * Complete 128 x 128 tiles take the k-contiguous "TN" kernel, which reads imm[i][k]; ragged shapes run on the general
 * kernel, which walks imm[k][i] -- the same product for an EXACTLY symmetric matrix only.  For a matrix that is
 * symmetric up to rounding (a dense Welford estimate) use bjx_dense_apply_imm_t, which takes the transposed copy too
 * and reads the matrix as stored at every shape (what dense-metric NUTS and the Python layer do since round 4; the
 * fused entry points -- bjx_leapfrog_dense, bjx_hmc_finish_dense, ... -- follow the same rule, so their callers pass
 * the transposed copy as `imm` for ragged shapes: blackjax_amd/dense.py::_imm_ptr). */
int bjx_dense_apply_imm(void* stream, int64_t N, int64_t D, const float* P, const float* imm, float* V);

/* Momentum draw: z = normal(km, (D,)) ; p = L^{-T} z = z @ mass_sqrt_t with mass_sqrt_t = L^{-1}
 * (L = cholesky(imm, lower)) ; ke = 0.5 dot(imm @ p, p).  z_work, v_work: (N, D) scratch.
 * Replaces: blackjax/mcmc/hmc.py:299,302 ; metrics.py:260-261,263-270,711-715 ; util.py:58-61,89-91. */
int bjx_hmc_momentum_dense(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                           int64_t step_fold, int64_t N, int64_t D, const float* mass_sqrt_t,
                           const float* imm, float* z_work, float* v_work, float* p_out,
                           float* ke_out);

/* Fused kick(s) + drift with a dense metric: p' = kick(p, g) [GEMM A-operand prologue, written to
 * p_out] ; v = imm @ p' [MFMA] ; q_out = q_in + eps v [epilogue].  p_out must not alias p_in;
 * q_out may alias q_in.  Replaces: blackjax/mcmc/integrators.py:104-150 with util.py:58-61. */
int bjx_leapfrog_dense(void* stream, int64_t N, int64_t D, int n_kicks, float eps,
                       const float* eps_per_chain, const float* imm, const float* q_in,
                       const float* p_in, const float* g, float* q_out, float* p_out);

/* Dense-metric counterparts of bjx_leapfrog_diag_coef / bjx_leapfrog_diag_masked,
 * bjx_hmc_finish_diag_coef and bjx_mhmc_step_diag (the reference's
 * multinomial_hmc_proposal, dynamic_hmc and palindromic integrators work with any metric,
 * blackjax/mcmc/hmc.py:181-248, dynamic_hmc.py:65-126, integrators.py:335-369).
 * matrix_stride < 0: ONE (D, D) matrix shared by all chains, applied on the fp32 MFMA GEMM
 * (p_out must not alias p_in); matrix_stride = 0 or D*D: the fp64-accumulated matrix-vector kernels
 * (shared matrix / one matrix per chain).  Kicks p += (eps*kick) g, drift q += (eps*drift) (imm p);
 * n_steps / step_idx as in bjx_leapfrog_diag_masked (NULL = every chain advances). */
int bjx_leapfrog_dense_coef(void* stream, int64_t N, int64_t D, int n_kicks, float kick_a, float kick_b,
                            float drift, float eps, const float* eps_per_chain, const float* imm,
                            int64_t matrix_stride, const float* q_in, const float* p_in, const float* g,
                            float* q_out, float* p_out, const int32_t* n_steps, int32_t step_idx);

int bjx_hmc_finish_dense_coef(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                              int64_t step_fold, int64_t N, int64_t D, float kick_coef, float eps,
                              const float* eps_per_chain, const float* imm, int64_t matrix_stride,
                              float divergence_threshold, const float* q0, const float* logp0,
                              const float* g0, const float* ke0, const float* q1, const float* logp1,
                              const float* g1, const float* p, float* p1_work, float* v_work,
                              float* p_end_out, float* q_out, float* logp_out, float* g_out,
                              float* acceptance_rate_out, uint8_t* is_accepted_out,
                              uint8_t* is_divergent_out, float* energy_out);

/* A WHOLE HMC transition in one launch for a log-density the engine evaluates itself (target_kind /
 * target_vec: the BJX_TARGET_* values of bjx_nuts.h) -- momentum draw, num_integration_steps velocity-Verlet
 * leapfrogs with (logp, grad) computed in registers, energies, Metropolis accept, select.  NOT the reference's
 * contract (blackjax.hmc calls logdensity_fn between two leapfrogs, hmc.py:279-312): an opt-in path that shows
 * what the contract costs; bit for bit the results of bjx_hmc_momentum_kick_diag + the target kernel +
 * bjx_leapfrog_diag + bjx_hmc_finish_diag.  Diagonal metric, 128 < D <= 1024, D % 4 == 0.
 * p0_out (HMCInfo.momentum) and q1_out / p_end_out / logp1_out / g1_out (HMCInfo.proposal) may each be NULL. */
int bjx_hmc_trajectory_diag(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                            int64_t step_fold, int64_t N, int64_t D, int64_t num_integration_steps,
                            float eps, const float* eps_per_chain, const float* imm, int64_t imm_stride,
                            float divergence_threshold, int32_t target_kind, const float* target_vec,
                            const float* q0, const float* logp0, const float* g0, float* p0_out,
                            float* q1_out, float* p_end_out, float* logp1_out, float* g1_out, float* q_out,
                            float* logp_out, float* g_out, float* acceptance_rate_out,
                            uint8_t* is_accepted_out, uint8_t* is_divergent_out, float* energy_out);

/* One step of multinomial HMC after the callable, dense metric: closing half kick p1 = p + (eps/2) g
 * [-> p1_work], v1 = imm p1 [-> v_work], then energy, weight, divergence flag, progressive uniform
 * sampling with key fold_in(integrator_key, step) and the reservoir copy of (q, p1, g).  The caller
 * continues the trajectory from p1_work with bjx_leapfrog_dense(_coef) and n_kicks = 1. */
int bjx_mhmc_step_dense(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                        int64_t step_fold, int64_t N, int64_t D, int64_t step, float eps,
                        const float* eps_per_chain, const float* imm, int64_t matrix_stride,
                        float divergence_threshold, const float* logp0, const float* ke0, const float* q,
                        const float* p, const float* g, const float* logp_new, float* p1_work,
                        float* v_work, float* weight, float* sum_log_p_accept, uint8_t* any_divergent,
                        uint8_t* ever_accepted, float* prop_q, float* prop_p, float* prop_g,
                        float* prop_logp, float* prop_energy);

/* bjx_mhmc_step_dense with per-chain trajectory lengths (blackjax.dmhmc with a dense metric: blackjax/__init__.py
 * 155-163 over mcmc/dynamic_hmc.py:85-118): chain r takes part in step `step` only while step < n_steps[r];
 * otherwise its momentum is copied to p1_work unchanged and its reservoir is left alone.  Continue with
 * bjx_leapfrog_dense_coef(..., n_steps, step + 1). */
int bjx_mhmc_step_dense_masked(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                               int64_t step_fold, int64_t N, int64_t D, int64_t step, float eps,
                               const float* eps_per_chain, const float* imm, int64_t matrix_stride,
                               float divergence_threshold, const float* logp0, const float* ke0, const float* q,
                               const float* p, const float* g, const float* logp_new, float* p1_work,
                               float* v_work, float* weight, float* sum_log_p_accept, uint8_t* any_divergent,
                               uint8_t* ever_accepted, float* prop_q, float* prop_p, float* prop_g,
                               float* prop_logp, float* prop_energy, const int32_t* n_steps);

/* bjx_mhmc_step_dense[_masked] for any palindromic integrator [b1, a1, ..., b1] (integrators.py:104-150, 335-369):
 * the closing kick of a leapfrog is p1 = p + (eps kick_coef) g with kick_coef = b1 (1/2 = velocity Verlet);
 * n_steps may be NULL (all chains take part: blackjax.mhmc) or per-chain trajectory lengths (blackjax.dmhmc).
 * The opening stage and the stages in between are bjx_leapfrog_dense_coef launches (round 4). */
int bjx_mhmc_step_dense_coef(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                             int64_t step_fold, int64_t N, int64_t D, int64_t step, float kick_coef, float eps,
                             const float* eps_per_chain, const float* imm, int64_t matrix_stride,
                             float divergence_threshold, const float* logp0, const float* ke0, const float* q,
                             const float* p, const float* g, const float* logp_new, float* p1_work,
                             float* v_work, float* weight, float* sum_log_p_accept, uint8_t* any_divergent,
                             uint8_t* ever_accepted, float* prop_q, float* prop_p, float* prop_g,
                             float* prop_logp, float* prop_energy, const int32_t* n_steps);

/* Closing half kick + energies + Metropolis accept + select with a dense metric (same contract as
 * bjx_hmc_finish_diag).  p1_work, v_work: (N, D) scratch (p1_work must not alias p). */
int bjx_hmc_finish_dense(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                         int64_t step_fold, int64_t N, int64_t D, float eps,
                         const float* eps_per_chain, const float* imm, float divergence_threshold,
                         const float* q0, const float* logp0, const float* g0, const float* ke0,
                         const float* q1, const float* logp1, const float* g1, const float* p,
                         float* p1_work, float* v_work, float* p_end_out, float* q_out,
                         float* logp_out, float* g_out, float* acceptance_rate_out,
                         uint8_t* is_accepted_out, uint8_t* is_divergent_out, float* energy_out);

/* ---- PER-CHAIN dense metric: one (D, D) inverse mass matrix per chain, (N, D, D) row-major ----
 * This is what a vmapped `window_adaptation(..., is_mass_matrix_diagonal=False).run` produces in
 * the reference.  Batched matrix-vector products with fp64 accumulation (bound by reading D^2
 * words per chain, not by MFMA).  Same contracts as the shared-matrix entry points above.
 *
 * y[c][i] = sum_j M[c][j][i] * x[c][j]   (M^T x per chain; the building block). */
int bjx_pc_matvec_t(void* stream, int64_t N, int64_t D, const float* M, int64_t matrix_stride,
                    const float* x, float* y);
/* matrix_stride = D*D for (N, D, D) per-chain matrices, 0 to apply ONE (D, D) matrix to every chain
 * through the same fp64-accumulated kernels (used by NUTS with a shared dense metric, where
 * bit-compatibility with the oracle matters more than MFMA throughput). */
int bjx_hmc_momentum_dense_pc(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                              int64_t step_fold, int64_t N, int64_t D, const float* mass_sqrt_t,
                              const float* imm, int64_t matrix_stride, float* z_work, float* v_work,
                              float* p_out, float* ke_out);
/* p_out may alias p_in here (each element is read and written by the same lane). */
int bjx_leapfrog_dense_pc(void* stream, int64_t N, int64_t D, int n_kicks, float eps,
                          const float* eps_per_chain, const float* imm, int64_t matrix_stride,
                          const float* q_in, const float* p_in, const float* g, float* q_out,
                          float* p_out);
int bjx_hmc_finish_dense_pc(void* stream, uint32_t key0, uint32_t key1, int64_t chain_offset,
                            int64_t step_fold, int64_t N, int64_t D, float eps,
                            const float* eps_per_chain, const float* imm, int64_t matrix_stride,
                            float divergence_threshold, const float* q0, const float* logp0,
                            const float* g0, const float* ke0, const float* q1, const float* logp1,
                            const float* g1, const float* p, float* p1_work, float* v_work,
                            float* p_end_out, float* q_out, float* logp_out, float* g_out,
                            float* acceptance_rate_out, uint8_t* is_accepted_out,
                            uint8_t* is_divergent_out, float* energy_out);

/* Welford with a dense second-moment matrix per chain: m2 (N, D, D) += outer(value - mean_new,
 * value - mean_old).  Replaces: blackjax/adaptation/mass_matrix.py:424-435 (dense branch). */
int bjx_welford_update_dense(void* stream, int64_t N, int64_t D, int64_t sample_size_new,
                             const float* value, const float* mean_in, const float* m2_in,
                             float* mean_out, float* m2_out);
/* Window end (dense): imm = (n/(n+5+k)) m2/(n-1) + (k/(n+5+k)) imm_prev + (5/(n+5+k)) 1e-3 I.
 * imm_prev is (D, D) [imm_prev_per_chain = 0] or (N, D, D) [= 1].  Replaces: mass_matrix.py:335-357. */
int bjx_welford_final_dense(void* stream, int64_t N, int64_t D, int64_t sample_size,
                            float imm_shrinkage_to_previous, const float* m2, const float* imm_prev,
                            int imm_prev_per_chain, float* imm_out);

/* ---- window adaptation (per-chain state; every array is (N,) unless noted) ------------------
 *
 * Dual averaging init (from_log_avg = 0: x = x_in) or window-end re-init (from_log_avg = 1:
 * x = exp(x_in) with x_in = log_step_size_avg):
 *   mu = log(10 x) ; log_x = log(x) ; log_x_avg = 0 ; avg_error = 0 ; step_size = exp(log_x)
 * Replaces: blackjax/optimizers/dual_averaging.py:87-99 ; adaptation/staged_adaptation.py:242-243. */
int bjx_da_init(void* stream, int64_t N, int from_log_avg, const float* x_in, float* log_x_out,
                float* log_x_avg_out, float* avg_error_out, float* mu_out, float* step_size_out);

/* One dual-averaging update with gradient = target - acceptance_rate (`step` >= 1 is the
 * reference's DualAveragingState.step, identical for all chains); outputs may alias inputs.
 *   reg = step + t0 ; eta = step^-kappa ; avg_error = (1 - 1/reg) avg_error + g/reg
 *   log_x = mu - (sqrt(step)/gamma) avg_error ; log_x_avg = eta log_x_prev + (1-eta) log_x_avg
 *   step_size = exp(log_x)
 * Replaces: dual_averaging.py:101-123 ; adaptation/step_size.py:129-145 ;
 * staged_adaptation.py:186-198. */
int bjx_da_update(void* stream, int64_t N, int64_t step, float target, float t0, float gamma,
                  float kappa, const float* acceptance_rate, const float* log_x_in,
                  const float* log_x_avg_in, const float* avg_error_in, const float* mu,
                  float* log_x_out, float* log_x_avg_out, float* avg_error_out,
                  float* step_size_out);

/* y = exp(x) element-wise, fp64 rounded once (final step size exp(log_step_size_avg),
 * staged_adaptation.py:301-305). */
int bjx_exp(void* stream, int64_t N, const float* x, float* y);

/* Welford accumulation of one new sample per chain, diagonal: value/mean/m2 are (N, D);
 * sample_size_new is the count AFTER adding this sample.  Outputs may alias inputs.
 * Replaces: blackjax/adaptation/mass_matrix.py:288-291,410-435. */
int bjx_welford_update_diag(void* stream, int64_t N, int64_t D, int64_t sample_size_new,
                            const float* value, const float* mean_in, const float* m2_in,
                            float* mean_out, float* m2_out);

/* Window end: imm = (n/(n+5+k)) m2/(n-1) + (k/(n+5+k)) imm_prev + (5/(n+5+k)) 1e-3, k =
 * imm_shrinkage_to_previous.  imm_prev is (D,) [stride 0] or (N, D) [stride D]; imm_out (N, D).
 * Replaces: mass_matrix.py:335-357,437-442. */
int bjx_welford_final_diag(void* stream, int64_t N, int64_t D, int64_t sample_size,
                           float imm_shrinkage_to_previous, const float* m2, const float* imm_prev,
                           int64_t imm_prev_stride, float* imm_out);

/* Built-in synthetic targets (value + gradient in one pass, fp64-accumulated logp) used
 * as the "user callable" by the bench and parity tests.
 *   diag gaussian:  g = -(q*inv_var) ; logp = 0.5 * sum q*g     (tests/fixtures.py:60-78)
 *   neal funnel:    tests/fixtures.py:81-98
 *   ar1 gaussian:   Sigma_ij = rho^|i-j| (tridiagonal precision)                         */
int bjx_target_diag_gaussian(void* stream, int64_t N, int64_t D, const float* inv_var,
                             const float* q, float* logp_out, float* g_out);
int bjx_target_neal_funnel(void* stream, int64_t N, int64_t D, const float* q,
                           float* logp_out, float* g_out);
int bjx_target_ar1_gaussian(void* stream, int64_t N, int64_t D, float diag_edge,
                            float diag_mid, float off, const float* q, float* logp_out,
                            float* g_out);

#ifdef __cplusplus
}
#endif

#include "bjx_nuts.h" /* NUTS entry points */

#endif /* BJX_HIP_H */
