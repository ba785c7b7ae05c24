/*
 * bjx_nuts.h -- C ABI of the NUTS part of libbjxhip.so (included by bjx_hip.h).
 *
 * Replaces blackjax/mcmc/nuts.py:113-145,223-321 (kernel, iterative_nuts_proposal),
 * blackjax/mcmc/trajectory.py:242-395,580-727 (dynamic_progressive_integration,
 * dynamic_multiplicative_expansion), blackjax/mcmc/proposal.py:51-105,118-176 and
 * blackjax/mcmc/termination.py:31-106 for N chains advanced in lockstep.
 *
 * Lockstep structure.  All chains start a transition together and every doubling `depth` has
 * a fixed length 2^depth unless the chain stops, so every chain that is still running is at the
 * same (depth, s).  The host loops over (depth, s); per leapfrog it launches
 *     bjx_nuts_pre   -> user log-density callable on the front positions -> bjx_nuts_post
 * and per doubling bjx_nuts_merge.  `idx` lists the chains that were running when the doubling
 * started (active-chain compaction: the callable only sees those rows); a chain whose subtree
 * has diverged / turned is skipped by the kernels through its SUB_ACTIVE flag.
 *
 * All per-chain scalars live in two caller-owned device tables, one row per slot:
 *     fs : float   [BJX_NUTS_NF][N]      is : int32_t [BJX_NUTS_NI][N]
 */
#ifndef BJX_NUTS_H
#define BJX_NUTS_H

#ifndef __HIPCC_RTC__ /* hiprtc has no system headers; blackjax_amd/rtc.py supplies the fixed-width names */
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* float slots (fs) */
enum {
  BJX_NUTS_F_H0 = 0,      /* initial energy of the transition (nuts.py:282) */
  BJX_NUTS_F_LLOGP = 1,   /* logdensity of the leftmost / rightmost trajectory state */
  BJX_NUTS_F_RLOGP = 2,
  BJX_NUTS_F_PLOGP = 3,   /* main proposal: logdensity, energy, weight, sum_log_p_accept */
  BJX_NUTS_F_PENERGY = 4,
  BJX_NUTS_F_PW = 5,
  BJX_NUTS_F_PSLPA = 6,
  BJX_NUTS_F_SLOGP = 7,   /* subtree proposal */
  BJX_NUTS_F_SENERGY = 8,
  BJX_NUTS_F_SW = 9,
  BJX_NUTS_F_SSLPA = 10,
  BJX_NUTS_F_ACC = 11,    /* acceptance_rate = exp(sum_log_p_accept) / num_states (nuts.py:303-305) */
  BJX_NUTS_NF = 12
};

/* int slots (is) */
enum {
  BJX_NUTS_I_ACTIVE = 0,     /* 1 while the chain keeps doubling */
  BJX_NUTS_I_SUB_ACTIVE = 1, /* 1 while the current subtree keeps integrating */
  BJX_NUTS_I_DIR = 2,        /* +1 / -1 direction of the current doubling */
  BJX_NUTS_I_NSTATES = 3,    /* trajectory.num_states */
  BJX_NUTS_I_SUBN = 4,       /* states in the current subtree */
  BJX_NUTS_I_SDIV = 5,       /* subtree diverged / turned */
  BJX_NUTS_I_STURN = 6,
  BJX_NUTS_I_DIV = 7,        /* NUTSInfo.is_divergent / is_turning */
  BJX_NUTS_I_TURN = 8,
  BJX_NUTS_I_DEPTH = 9,      /* NUTSInfo.num_trajectory_expansions */
  BJX_NUTS_I_KT = 10,        /* the two words of the current doubling's leaf-sampling key: */
  BJX_NUTS_I_KTB = 11,       /*   split(fold_in(integrator_key, depth), 3)[1], written once per doubling */
  BJX_NUTS_I_KP = 12,        /* likewise the doubling's proposal key split(...)[2] (merge step) */
  BJX_NUTS_I_KPB = 13,
  BJX_NUTS_I_IK = 14,        /* the transition's integrator key split(chain_key, 2)[1], written at doubling 0 */
  BJX_NUTS_I_IKB = 15,
  BJX_NUTS_I_LAZY = 16,      /* free-running chains: bit 1/2 = the left/right trajectory end, 4 = the
                                proposal, 8 = the momentum sum is still the transition's initial state
                                (its rows are then read from q0 / p0 / g0 instead of being copied) */
  BJX_NUTS_I_STAGE = 17,     /* free-running chains on the GENERAL tick kernel with a multi-stage integrator
                                (bjx_nuts_async_t.int_stages > 1): gradients of the leaf in flight already used */
  BJX_NUTS_NI = 18
};

typedef struct {
  int64_t N, D;
  int32_t max_depth;          /* max_num_doublings */
  int32_t reserved;
  const float* imm;           /* diagonal inverse mass matrix, (D,) or (N, D) */
  int64_t imm_stride;         /* 0 or D */
  const float* eps_per_chain; /* (N,) or NULL */
  float eps;
  float divergence_threshold;
  uint32_t key0, key1;        /* key layout as in bjx_hip.h */
  int64_t chain_offset, step_fold;
  /* (N, D) arrays */
  const float *q0, *g0, *p0;  /* initial position / gradient, drawn momentum */
  float *Lq, *Lp, *Lg;        /* leftmost trajectory state */
  float *Rq, *Rp, *Rg;        /* rightmost trajectory state */
  float *msum, *Smsum;        /* momentum sums: trajectory, current subtree */
  float *Pq, *Pg;             /* main proposal state (becomes the new chain state) */
  float *Sq, *Sg;             /* subtree proposal state */
  float *ckpt_r, *ckpt_rs;    /* (N, max_depth, D) U-turn checkpoints (termination.py:46-54) */
  float* fs;                  /* (BJX_NUTS_NF, N) */
  int32_t* is;                /* (BJX_NUTS_NI, N) */
  /* Dense metric (Mdense != NULL; `imm` is then ignored): velocities M^{-1} p are fp64-accumulated
   * matrix-vector products (metrics.py:263-304 with util.py:58-61); the velocities of the two
   * trajectory ends and of every checkpointed momentum are kept so U-turn checks need no extra
   * products. */
  const float* Mdense;        /* (D, D) shared [stride 0] or (N, D, D) per chain [stride D*D] */
  int64_t Mdense_stride;
  const float* v0;            /* (N, D) M^{-1} p0 from the momentum draw */
  float *Lv, *Rv;             /* (N, D) velocities of the leftmost / rightmost state */
  float* ckpt_v;              /* (N, max_depth, D) velocities of the checkpointed momenta */
  /* Palindromic integrator [b_1, a_1, b_2, ..., b_1] (blackjax/mcmc/nuts.py:150-158 `integrator=`,
   * integrators.py:104-150, 335-369): int_kick = b_1 (the opening AND the closing kick of a leaf are
   * (dir * eps * b_1) g), int_drift = a_1 (the first drift, (dir * eps * a_1) M^{-1} p).  The stages
   * in between (b_2, a_2, ...), each followed by a callable evaluation, are bjx_nuts_mid launches.
   * Both 0: velocity Verlet (0.5, 1.0).  Lockstep entry points only (bjx_nuts_pre / post / _ctl);
   * the free-running tick kernels integrate with velocity Verlet. */
  float int_kick, int_drift;
  /* Shared dense metric on the MFMA GEMM (Mdense != NULL, Mdense_stride == 0): v_pre != NULL is a
   * compact (n_rows, D) array holding, for row b, the velocity M^{-1} p' of the kicked momentum the
   * next kernel is about to form -- computed by the caller with bjx_nuts_dense_kick (p' rows) and
   * bjx_dense_apply_imm (one GEMM for all live rows).  bjx_nuts_pre / _mid / _post (and their _ctl
   * forms) then read their velocities from it instead of running one fp64 mat-vec per chain
   * (D^2 words per chain and leapfrog).  The products are fp32 fmaf chains in the engine's stated k
   * order (NOTEBOOK.md section 3 item 6), i.e. the oracle's "f32chain" mode.  post must then be called
   * with fuse_next = 0 (the next leaf's opening velocity needs its own product).  NULL: as before. */
  const float* v_pre;
} bjx_nuts_t;

/* Start of a transition: trajectory = (z0, z0, momentum_sum = p0, num_states = 0), proposal =
 * (z0, H0, 0, -inf) with H0 = -logp0 + ke0 (nuts.py:278-294).  logp0, ke0: (N,). */
int bjx_nuts_init(void* stream, const bjx_nuts_t* nuts, const float* logp0, const float* ke0);

/* Opening half of leapfrog `s` of doubling `depth` for rows idx[0..n_rows) (idx NULL = all chains
 * in order): at s == 0 draws the direction (trajectory.py:645-650); p += (dir*eps/2) g ;
 * q += dir*eps * imm*p on the trajectory end in that direction; writes the new position also to
 * the compact row qf[b] for the user callable (integrators.py:104-150, trajectory.py:323). */
int bjx_nuts_pre(void* stream, const bjx_nuts_t* nuts, int32_t depth, int64_t s, int64_t n_rows,
                 const int32_t* idx, float* qf);

/* Closing half: p += (dir*eps/2) g_new; proposal weight, divergence, progressive uniform sampling
 * (key fold_in(trajectory_key, s)), momentum-sum append, checkpoint update and iterative U-turn
 * (trajectory.py:321-346, proposal.py:68-103,118-143, termination.py:56-104).
 * logp_f (n_rows,), gf (n_rows, D): callable outputs for the compact rows.
 * fuse_next != 0: for chains whose subtree keeps integrating, also perform the opening half of
 * leapfrog s + 1 (exactly bjx_nuts_pre's arithmetic; the new position overwrites qf[b]) so the
 * caller skips the bjx_nuts_pre launch of the next leaf.  Ignored at the last leaf of a doubling.
 * The caller must not fuse across a re-compaction of idx (row order of qf would change). */
int bjx_nuts_post(void* stream, const bjx_nuts_t* nuts, int32_t depth, int64_t s, int64_t n_rows,
                  const int32_t* idx, float* qf, const float* logp_f, const float* gf,
                  int32_t fuse_next);

/* Intermediate stage of a multi-stage integrator on the trajectory end that is integrating, for the
 * rows idx[0..n_rows) whose subtree is still running: p += (dir * eps * kick) gf[b] ;
 * q += (dir * eps * drift) M^{-1} p ; the new position also goes to qf[b] for the next callable
 * evaluation (integrators.py:128-146, stages 2 .. K of generalized_two_stage_integrator).  gf = the
 * callable's gradient at the previous stage's position.  ctl != NULL: replayable form, row count from
 * ctl[2] (at most n_rows) as in bjx_nuts_pre_ctl.  Diagonal and dense metrics. */
int bjx_nuts_mid(void* stream, const bjx_nuts_t* nuts, int64_t n_rows, const int32_t* idx,
                 const int64_t* ctl, float* qf, const float* gf, float kick, float drift);

/* Shared dense metric on the GEMM: compact rows of KICKED momenta for the next product.  Row b gets
 * p_end + (dir * eps * kick) g with p_end the momentum of the trajectory end that is integrating and
 * g = gf[b] (the callable's latest gradient; closing kick or a stage kick) or, when gf == NULL, the
 * end's own stored gradient (opening kick of leaf s; at s == 0 the direction of the doubling is drawn
 * first, exactly as bjx_nuts_pre does -- that kernel re-derives the same values).  Rows whose chain
 * is not integrating are zero-filled.  Nothing but pc_out is written.  ctl != NULL: replayable form
 * (depth, leaf base and row count from the control block, s = ctl[1] + s_off). */
int bjx_nuts_dense_kick(void* stream, const bjx_nuts_t* nuts, int32_t depth, int64_t s, int64_t n_rows,
                        const int32_t* idx, const int64_t* ctl, const float* gf, float kick, float* pc_out);

/* HIP-graph-replayable variants of bjx_nuts_pre / bjx_nuts_post.  The per-launch parameters that
 * change between replays are read from a DEVICE control block
 *     ctl : int64_t[8] = { depth, s_base, n_rows, key0, key1, step_fold, chain_offset, 0 }
 * so that one captured chunk of k leapfrogs ( pre_ctl(s_off=0), then [callable, post_ctl(s_off=i,
 * fuse_next = i < k-1)] for i = 0..k-1 ) serves every chunk of every transition: leaf index s = s_base + s_off, rows
 * idx[0 .. min(n_rows, n_cap)) are processed, the key fields of `nuts` are ignored.  idx must be
 * non-NULL (a caller-owned device buffer whose CONTENTS may change between replays). */
int bjx_nuts_pre_ctl(void* stream, const bjx_nuts_t* nuts, int32_t s_off, int64_t n_cap,
                     const int32_t* idx, const int64_t* ctl, float* qf);
int bjx_nuts_post_ctl(void* stream, const bjx_nuts_t* nuts, int32_t s_off, int64_t n_cap,
                      const int32_t* idx, const int64_t* ctl, float* qf, const float* logp_f,
                      const float* gf, int32_t fuse_next);

/* Device-side active-chain compaction: idx_out <- the chains of idx_in[0..n_in) (identity list if
 * idx_in is NULL; n_in < 0 means "read the count from ctl[2]") whose `flag_slot`
 * (BJX_NUTS_I_ACTIVE or BJX_NUTS_I_SUB_ACTIVE) is set, order preserved; the new count is written
 * to ctl[2].  idx_out may alias idx_in.  No host synchronisation. */
int bjx_nuts_compact(void* stream, const bjx_nuts_t* nuts, int32_t flag_slot, int64_t n_in,
                     const int32_t* idx_in, int32_t* idx_out, int64_t* ctl);

/* Stream-ordered update of the control block (a one-thread kernel; no host staging buffer).
 * n_rows < 0 keeps the count last written by bjx_nuts_compact. */
int bjx_nuts_set_ctl(void* stream, int64_t* ctl, int32_t depth, int64_t s_base, int64_t n_rows,
                     uint32_t key0, uint32_t key1, int64_t step_fold, int64_t chain_offset);

/* End of doubling `depth`: biased progressive sampling or sum_log_p_accept update, trajectory
 * merge, U-turn check on the merged trajectory, flags and counters, acceptance rate
 * (trajectory.py:672-715, proposal.py:146-176, metrics.py:272-304, nuts.py:303-305). */
int bjx_nuts_merge(void* stream, const bjx_nuts_t* nuts, int32_t depth, int64_t n_rows,
                   const int32_t* idx);

/* ---- Free-running chains: many transitions per chain without lockstep ---------------------------
 *
 * A run of n_steps transitions (blackjax/util.py:150-213 run_inference_algorithm over
 * nuts.step) in which every chain advances through ITS OWN sequence of trees: one "tick" =
 * one leapfrog of whatever tree each chain is building; a chain that finishes a transition
 * records it and starts the next one in the same tick.  Every tick therefore carries N useful
 * leapfrogs (until chains run out of transitions) instead of the lockstep scheme's handful of
 * stragglers in deep doublings; the number of ticks is max_c sum_t leaves(c, t) instead of
 * sum_t max_c leaves(c, t).  Per-chain results are IDENTICAL to n_steps lockstep transitions:
 * chain c at transition t uses the same key either way.
 *
 * The host loops   bjx_nuts_async_tick -> user callable on qf (all N rows) -> bjx_nuts_async_tick ...
 * until *n_done == N.  max_depth >= 1.  Diagonal or dense metric (per-chain adaptation: diagonal). */
#define BJX_NUTS_MAX_MID 6
typedef struct {
  const uint32_t* step_keys;  /* (n_steps, 2) device table of the run's per-transition keys
                                 (step-major layout: chain key = split(step_keys[t], N)[c]); NULL:
                                 chain-major layout, transition t folds t_first + t into the chain key
                                 derived from nuts->key0/key1 */
  int32_t t_first, n_steps;
  float *q, *g, *logp;        /* (N,D), (N,D), (N,): current chain state, in/out; q, g must be the
                                 buffers nuts->q0 / nuts->g0 point to */
  float* p;                   /* (N,D) momentum buffer; must be the buffer nuts->p0 points to */
  int32_t* t;                 /* (N,) transitions completed per chain, in/out (start at 0) */
  int32_t* phase;             /* (N,) 0 = start a transition, 1 = a leaf awaits its gradient, 2 = done,
                                 3 = subtree complete (transient, between the two kernels of a tick) */
  int32_t* n_done;            /* device counter of chains that completed n_steps transitions */
  const int32_t* rows;        /* compact row b of (qf, logp_f, gf) belongs to chain rows[b]; NULL = identity.
                                 Lets the caller drop finished chains from the callable's batch
                                 (bjx_nuts_async_compact). */
  int64_t n_rows;             /* number of compact rows (N when rows == NULL) */
  /* per-(transition, chain) records, row-major (n_steps, N[, D]); any of them may be NULL */
  float* out_position;
  float* out_logdensity;
  float* out_acceptance_rate;
  float* out_energy;
  int32_t* out_num_integration_steps;
  int32_t* out_num_trajectory_expansions;
  uint8_t* out_is_divergent;
  uint8_t* out_is_turning;
  /* Optional per-chain window adaptation (blackjax/adaptation/staged_adaptation.py:186-249 as a
   * vmapped warm-up runs it: every chain adapts its own step size and diagonal inverse mass matrix).
   * adapt_tab == NULL: none.  Otherwise, when chain c completes transition t it runs, in this order,
   * the Welford update with its new position if the transition lies in a slow window
   * (mass_matrix.py:410-435), the dual-averaging update with the transition's acceptance rate
   * (dual_averaging.py:101-123, step_size.py:144) and, at a window end, the inverse-mass-matrix
   * update + Welford reset (mass_matrix.py:335-357) and the dual-averaging restart from the averaged
   * log step size (staged_adaptation.py:233-249) -- then starts transition t + 1 with the new
   * step size and metric.  The schedule and every scalar that depends only on t are tabulated by the
   * host: adapt_tab is (n_steps, BJX_NUTS_ADAPT_COLS) floats, columns BJX_NUTS_AT_*.
   * Requires nuts->eps_per_chain == adapt_step_size and nuts->imm == adapt_imm with imm_stride == D. */
  const float* adapt_tab;
  float adapt_target;         /* target acceptance rate */
  float adapt_reserved;
  float *adapt_log_x, *adapt_log_x_avg, *adapt_avg_err, *adapt_mu; /* (N,) dual-averaging state, in/out */
  float* adapt_step_size;     /* (N,) in/out */
  float *adapt_mean, *adapt_m2; /* (N, D) Welford state, in/out */
  float* adapt_imm;           /* (N, D) per-chain diagonal inverse mass matrix, in/out */
  float* out_step_size;       /* optional (n_steps, N): the step size after the update of transition t */
  /* Optional work buffers of the lean tick kernel (k_nuts_async_tick3: diagonal metric, D % 4 == 0, D <= 1 024,
   * 16-byte aligned buffers; engine-resident targets: D <= 512): rec = (N, BJX_NUTS_REC_WORDS) packed per-chain
   * scalars (every scalar a leaf needs in one 128-byte line; the kernels own the layout, the caller only zero-fills
   * it before the first tick), front_p = (N, D) momentum of the trajectory end that is integrating.
   * Both NULL: the general one-launch tick over the fs / is slot tables is used (k_nuts_async_fused). */
  int32_t* rec;
  float* front_p;
  /* Work lists.  ABI 5 (round 5): the two-kernel ticks that used them are gone -- a transition end is now served
   * by the chain's own wave in the NEXT tick's launch -- so for the diagonal-metric ticks end_list / end_count /
   * tick are ignored (callers may pass NULL / 0).  GEMM mode (gemm_pc != NULL) still needs them: end_list = (2, N)
   * chain and compact row of every slot of a tick's momentum list, end_count = int32[1]. */
  int32_t* end_list;
  int32_t* end_count;
  int32_t tick;
  int32_t keep_ends;          /* ABI 6, lean tick kernel only (rec / front_p given): != 0 makes a chain that completes a
                                 transition leave both trajectory ends in nuts->Lq/Lp/Lg, Rq/Rp/Rg and their log-densities in
                                 words 29 / 30 of its record (as float bits) -- NUTSInfo.trajectory_leftmost_state /
                                 rightmost_state (nuts.py:66-70) of the chain's LAST transition; meant for n_steps == 1 */
  /* Optional device-side row count: when non-NULL the kernels process min(n_rows, *n_rows_dev) rows
   * (launch geometry still follows n_rows).  Lets ONE recorded sequence of ticks serve every batch
   * size of a run's tail: bjx_nuts_async_compact writes the new count to its n_out argument, which
   * the next ticks then read here. */
  const int32_t* n_rows_dev;
  /* Dense metric (nuts->Mdense != NULL): the transition-start momentum draw of chain c needs
   * mass_sqrt_t = L^{-1} ((D, D) shared or (N, D, D), the layout of nuts->Mdense; metrics.py:701-729,
   * util.py:23-61) and writes the velocity M^{-1} p0 to v0, which must be the buffer nuts->v0 points
   * to.  Both NULL for a diagonal metric. */
  const float* mass_sqrt_t;
  float* v0;
  /* Optional engine-resident log-density (round 3; low-traffic tick kernels only: diagonal metric,
   * rec / front_p given).  target_kind != 0: every row that leaves the tick with a new pending position
   * also gets its log-density and gradient evaluated IN the tick, by the device function the stand-alone
   * target kernel runs (csrc/bjx_targets_dev.h: identical results), written to logp_f[b] / gf[b] -- the
   * arrays passed to bjx_nuts_async_tick, which are then in/out -- so the caller launches NO callable
   * between ticks: one launch per leapfrog instead of two.  This is outside the external-callable
   * contract of the engine (NOTEBOOK.md section 7): it exists for the targets the library itself ships.
   * BJX_TARGET_NEAL_FUNNEL: no parameters; BJX_TARGET_DIAG_GAUSSIAN: target_vec = inv_var (D,), D > 128. */
  int32_t target_kind;
  int32_t ticks_per_launch;   /* target_kind != 0: every wave advances ITS chain by this many ticks inside one
                                 launch of the one-kernel tick (chains are independent and the log-density is
                                 evaluated in place, so nothing separates two leapfrogs of a chain but the
                                 wave's own stores); 0 or 1: one tick per launch */
  const float* target_vec;
  /* Multi-stage palindromic integrators [b1, a1, b2, a2, ..., b1] (blackjax/mcmc/integrators.py:270-369; round 4:
   * low-traffic tick kernels, stage counter in the record; round 6: the general tick kernel too -- any row length,
   * per-chain dense metrics --, stage counter in the slot BJX_NUTS_I_STAGE, which the caller zeroes before the first
   * tick; NOT the shared-dense GEMM mode): int_stages = gradients per leapfrog
   * (0 or 1: velocity Verlet / any one-gradient integrator; b1, a1 are bjx_nuts_t.int_kick / int_drift).  A leaf
   * then lasts int_stages ticks: after the opening (b1, a1) each of the first int_stages - 1 gradients drives a
   * middle stage p += (dir eps int_mid_kick[i]) g ; q += (dir eps int_mid_drift[i]) M^-1 p, i = 0 .. int_stages - 2
   * (the coefficients b_2, a_2, ...), and the last one the closing kick b1 and the leaf's bookkeeping.
   * At most BJX_NUTS_MAX_MID middle stages. */
  int32_t int_stages;
  int32_t reserved3;
  float int_mid_kick[BJX_NUTS_MAX_MID];
  float int_mid_drift[BJX_NUTS_MAX_MID];
  /* ONE shared dense inverse mass matrix on the MFMA GEMM (round 4; nuts->Mdense with Mdense_stride == 0,
   * velocity Verlet / one-gradient integrators, no per-chain adaptation, no engine-resident target): gemm_pc != NULL
   * makes every product v = M^{-1} p of a tick one GEMM over the compact rows (bjx_dense_apply_imm) -- the arithmetic
   * of the lockstep step for this metric (bjx_nuts_dense_kick -> bjx_dense_apply_imm -> kernels reading
   * nuts->v_pre; blackjax/mcmc/metrics.py:263-304 with util.py:58-61) -- instead of one fp64-accumulated
   * matrix-vector product per chain.  A tick is then a fixed sequence of launches on `stream` (kick, GEMM, leaf,
   * two GEMMs for the momenta of the chains that start a transition, tree start, kick, GEMM, opening half).
   * gemm_pc, gemm_vc: (n_rows capacity, D) kicked momenta of the compact rows and their velocities; gemm_vc must
   * be nuts->v_pre.  gemm_z, gemm_pm, gemm_vm: (gemm_cap, D) normal draws, momenta p = L^{-T} z and velocities of
   * the chains that start a transition in this tick: at most gemm_cap per tick, the others wait a tick (which
   * changes no result: chains are independent).  Needs end_list ((2, N): chain and compact row of every list slot)
   * and end_count (int32[1]). */
  float* gemm_pc;
  float* gemm_vc;
  float* gemm_z;
  float* gemm_pm;
  float* gemm_vm;
  int64_t gemm_cap;
  const float* gemm_mass_sqrt; /* optional (D, D): L^{-T} = transpose of mass_sqrt_t, row-major.  Given, p = L^{-T} z runs on
                                  the GEMM kernel that reads its matrix as stored (the one bjx_dense_apply_imm uses); NULL:
                                  bjx_dense_matmul with mass_sqrt_t.  Same k order, same results. */
  const float* gemm_imm_t;     /* optional (D, D): transpose of nuts->Mdense, row-major.  Given, the products run through
                                  bjx_dense_apply_imm_t (few live rows -- the tail of a run -- on its latency-oriented
                                  kernel); NULL: bjx_dense_apply_imm.  Same results. */
} bjx_nuts_async_t;

enum {
  BJX_TARGET_NONE = 0,
  BJX_TARGET_NEAL_FUNNEL = 1,
  BJX_TARGET_DIAG_GAUSSIAN = 2,
  BJX_TARGET_USER = 3 /* only in kernels compiled at run time around a user-written device target
                         (blackjax_amd/rtc.py; target_vec = the user's parameter pointer); libbjxhip refuses it */
};

#define BJX_NUTS_REC_WORDS 32

/* columns of adapt_tab */
enum {
  BJX_NUTS_AT_FLAGS = 0,      /* 1 = slow window (Welford update), 2 = window end; as a float */
  BJX_NUTS_AT_DA_REG = 1,     /* dual averaging: step + t0 */
  BJX_NUTS_AT_DA_INV_REG = 2, /* 1 / (step + t0) */
  BJX_NUTS_AT_DA_ETA = 3,     /* step^-kappa */
  BJX_NUTS_AT_DA_COEF = 4,    /* sqrt(step) / gamma */
  BJX_NUTS_AT_WEL_N = 5,      /* Welford sample size AFTER this transition's update */
  BJX_NUTS_AT_FIN_NM1 = 6,    /* window end: sample_size - 1 */
  BJX_NUTS_AT_FIN_BETA_DATA = 7,
  BJX_NUTS_AT_FIN_BETA_PREV = 8,
  BJX_NUTS_AT_FIN_REG = 9,
  BJX_NUTS_ADAPT_COLS = 12
};

/* One tick for the chains of the compact rows.  logp_f (n_rows,), gf (n_rows, D): callable outputs at
 * qf from the previous tick (ignored by chains in phase 0); qf (n_rows, D): positions whose
 * log-density / gradient the next tick needs (rows of finished chains are left untouched). */
int bjx_nuts_async_tick(void* stream, const bjx_nuts_t* nuts, const bjx_nuts_async_t* run, float* qf,
                        const float* logp_f, const float* gf);

/* Drop the finished chains (phase 2) from the compact rows, order preserved: rows_out[b'] = chain,
 * qf_out[b'] = the pending position of that chain (row gathered from qf_in), *n_out = new count.
 * rows_out / qf_out must not alias run->rows / qf_in.  No host synchronisation. */
int bjx_nuts_async_compact(void* stream, const bjx_nuts_t* nuts, const bjx_nuts_async_t* run,
                           const float* qf_in, int32_t* rows_out, float* qf_out, int32_t* src_work,
                           int32_t* n_out);

/* ---- Speculative tail of a free-running run (round 5, ABI 6) -------------------------------------
 *
 * With a handful of live chains a tick is pure latency: two DEPENDENT launches per leapfrog (tick kernel, user
 * callable), each a launch boundary plus a couple of memory round trips.  Only two things are truly serial in a NUTS
 * leaf, though: the position update that feeds the next gradient evaluation and that evaluation.  Everything else
 * (energy, progressive sampling, checkpoints, U-turn tests, subtree merges, transition ends) merely decides WHEN THE
 * TRANSITION STOPS -- within a transition the sequence of integration steps is fixed by the key alone (the direction
 * of doubling d is bernoulli(split(fold_in(integrator_key, d), 3)[0]), trajectory.py:645-650, and any stop ends the
 * transition, trajectory.py:672-715).  So the tail runs as two host-driven streams:
 *
 *   stream A (latency-critical):  [ user callable on qf  ->  bjx_nuts_spec_integrate ] per leapfrog.  The integrator
 *       closes the leaf (kick), pushes the callable's (logp, gradient) with the leaf's identity (epoch, depth, s)
 *       into a per-row ring, and opens the next leaf ALONG THE KEY'S DIRECTION SCHEDULE, across doubling boundaries
 *       (it keeps its own copies of the two trajectory ends) -- one memory round trip, a few fmaf per element.
 *   stream B (off the critical path):  bjx_nuts_spec_book, one long-lived launch per recorded sequence of stream A.
 *       Each row's wave replays the UNCHANGED tick arithmetic (the lean leaf + deferred transition end of
 *       bjx_nuts_async_tick) over the ring records in order, on its own copy of the pending position, so every
 *       decision, record and state it produces is bit for bit what the one-stream tick produces.  When a
 *       transition ends it publishes the next transition's start (epoch + 1, integrator key, step size; the first
 *       pending position / momentum are its own buffers) and the integrator, which has meanwhile speculated a few
 *       leaves past the end, restarts from there; ring records of an older epoch are skipped.
 *
 * Cost: the leaves speculated past each transition end (the bookkeeper's lag: 2-3 leapfrogs) -- wasted callable
 * evaluations at positions of a trajectory the reference would not have continued (finite or not: a diverged
 * trajectory may overflow; the callable sees what it would see one leaf before a divergence is detected).
 * Synchronisation is device memory only (no events between the streams): counters are agent-scope atomics, ring
 * data is published one kernel late (a record is announced by the NEXT integrate launch, i.e. after the launch
 * that wrote it has completed), the book side fences before reading, and the integrator fences before reading a
 * restart.  Nothing ever spins unboundedly: the integrator returns when its ring is full or its tree is exhausted,
 * the bookkeeper returns when stream A's sequence counter reaches `target` (or after timeout_us).
 * Every record carries the first four floats of the integrator's position; the bookkeeper compares them with its
 * own replica and counts mismatches in dbg[0] -- the caller must treat a non-zero count as a failed run.
 *
 * Diagonal metric, D % 4 == 0, D <= 1024, one-gradient integrators, external callable, no per-chain adaptation. */
#define BJX_NUTS_SPEC_IW 8    /* int32 words per row of bjx_nuts_spec_t.iw / .bw */
#define BJX_NUTS_SPEC_TAG 8   /* int32 words per ring slot of ring_tag */
typedef struct {
  int64_t n_rows;             /* capacity: rows of every buffer below and launch geometry */
  const int32_t* n_rows_dev;  /* optional device-side live count (<= n_rows), as bjx_nuts_async_t.n_rows_dev */
  const int32_t* rows;        /* (n_rows,) chain of spec row b (bjx_nuts_async_compact output) */
  int32_t ring;               /* slots per row, a power of two >= 8 */
  int32_t lead;               /* records stream A may be ahead of stream B's consumed count (<= 0: ring - 1) */
  float* qf;                  /* (n_rows, D) stream A: the callable's input, in/out */
  float* fp;                  /* (n_rows, D) stream A: momentum after the opening kick */
  float *eLq, *eLp, *eLg;     /* (n_rows, D) stream A: its copy of the leftmost trajectory state */
  float *eRq, *eRp, *eRg;     /* ... and of the rightmost */
  int32_t* iw;                /* (n_rows, BJX_NUTS_SPEC_IW) stream A: epoch, depth, s, directions, eps, pushed, state */
  float* ring_g;              /* (n_rows, ring, D) gradients pushed by A */
  int32_t* ring_tag;          /* (n_rows, ring, BJX_NUTS_SPEC_TAG): epoch, depth, s, logp bits, 4 position words */
  int32_t* avail;             /* (n_rows,) records announced by A */
  int64_t* ack;               /* (n_rows,) A -> B: (epoch << 32) | number of the first record of that epoch */
  float* qf_book;             /* (n_rows, D) stream B: its replica of the pending position */
  int32_t* bw;                /* (n_rows, BJX_NUTS_SPEC_IW) stream B -> A: epoch (-1 = chain finished), consumed, key, eps */
  int32_t* a_seq;             /* int32[1]: sequences completed by stream A (bumped by integrate launches with bump != 0) */
  int32_t* dbg;               /* int32[8]: 0 replica mismatches (must stay 0), 1 ring-full stalls, 2 restarts, 6 / 7 book time (100 MHz ticks) / records,
                                 3 stale records skipped, 4 book time-outs, 5 out-of-order records (must stay 0) */
} bjx_nuts_spec_t;

/* Hand the rows over to the speculative tail: spec->rows / spec->qf (and *n_rows_dev) hold the output of
 * bjx_nuts_async_compact for the live chains, each between two ticks (a leaf awaiting its gradient, or a
 * transition end pending).  Fills every other buffer of `spec`. */
int bjx_nuts_spec_enter(void* stream, const bjx_nuts_t* nuts, const bjx_nuts_async_t* run, const bjx_nuts_spec_t* spec);

/* Stream A, once per leapfrog after the callable: logp_f (n_rows,), gf (n_rows, D) = callable outputs at spec->qf.
 * bump != 0: the launch also counts a completed sequence in *a_seq (last launch of a recorded sequence). */
int bjx_nuts_spec_integrate(void* stream, const bjx_nuts_t* nuts, const bjx_nuts_async_t* run,
                            const bjx_nuts_spec_t* spec, const float* logp_f, const float* gf, int32_t bump);

/* Stream B: consume ring records until *a_seq >= target (and the ring is drained) or timeout_us have passed. */
int bjx_nuts_spec_book(void* stream, const bjx_nuts_t* nuts, const bjx_nuts_async_t* run,
                       const bjx_nuts_spec_t* spec, int32_t target, int32_t timeout_us);

/* Do launches on stream_wait and stream_set overlap?  (HIP streams share a small number of hardware queues; two
 * streams that landed on the same queue run in order, and the speculative tail must not be used with such a pair.)
 * Launches a one-wave kernel on stream_wait that spins until a kernel launched AFTER it on stream_set has set
 * flag2[0], or timeout_us have passed; flag2 (device int32[2], zero on entry) reads {1, 1} afterwards when the two
 * launches overlapped, {., 2} when they did not.  The caller synchronises and reads flag2. */
int bjx_stream_probe(void* stream_wait, void* stream_set, int32_t* flag2, int32_t timeout_us);

#ifdef __cplusplus
}
#endif
#endif /* BJX_NUTS_H */
