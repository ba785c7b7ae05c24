"""ctypes binding of ``libbjxhip.so`` (C ABI declared in ``include/bjx_hip.h``).

There is NO fallback: if the shared library is missing or a call fails the
product path raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C blackjax_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbjxhip.so")

_f32p = c_void_p  # device pointers travel as integers (tensor.data_ptr())
_u8p = c_void_p

# name -> argtypes ; every function returns int except bjx_last_error
SIGNATURES = {
    "bjx_abi_version": [],
    "bjx_keys_split": [c_uint32, c_uint32, c_int64, c_int64, POINTER(c_uint32)],
    "bjx_rng_normal": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, _f32p],
    "bjx_rng_uniform": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, _f32p],
    "bjx_rng_key_probe": [c_void_p, c_uint32, c_uint32, c_int64, _f32p, _f32p, c_int64, c_void_p],
    "bjx_log1p_device_check": [c_void_p, c_uint32, c_void_p],
    "bjx_hmc_momentum_diag": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, _f32p,
                              c_int64, _f32p, _f32p],
    "bjx_hmc_momentum_kick_diag": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, _f32p,
                                   c_int64, c_float, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p],
    "bjx_leapfrog_diag": [c_void_p, c_int64, c_int64, c_int, c_float, _f32p, _f32p, c_int64,
                          _f32p, _f32p, _f32p, _f32p, _f32p],
    "bjx_hmc_finish_diag": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, c_float,
                            _f32p, _f32p, c_int64, c_float,
                            _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                            _f32p, _f32p, _f32p, _f32p, _f32p, _u8p, _u8p, _f32p],
    "bjx_leapfrog_diag_masked": [c_void_p, c_int64, c_int64, c_int, c_float, _f32p, _f32p, c_int64,
                                 _f32p, _f32p, _f32p, _f32p, _f32p, c_void_p, ctypes.c_int32],
    "bjx_leapfrog_diag_coef": [c_void_p, c_int64, c_int64, c_int, c_float, c_float, c_float, c_float,
                               _f32p, _f32p, c_int64, _f32p, _f32p, _f32p, _f32p, _f32p, c_void_p,
                               ctypes.c_int32],
    "bjx_hmc_finish_diag_coef": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64,
                                 c_float, c_float, _f32p, _f32p, c_int64, c_float,
                                 _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                 _f32p, _f32p, _f32p, _f32p, _f32p, _u8p, _u8p, _f32p],
    "bjx_keys_child": [c_void_p, c_int64, c_void_p, c_uint32, c_void_p],
    "bjx_keys_randint": [c_void_p, c_int64, c_void_p, ctypes.c_int32, ctypes.c_int32, c_void_p],
    "bjx_mhmc_step_diag": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, c_int64,
                           c_int, c_float, _f32p, _f32p, c_int64, c_float, _f32p, _f32p, _f32p, _f32p,
                           _f32p, _f32p, _f32p, _f32p, _u8p, _u8p, _f32p, _f32p, _f32p, _f32p, _f32p],
    "bjx_mhmc_finish": [c_void_p, c_int64, c_int64, c_int64, _f32p, _f32p, _f32p, _f32p, _f32p, _u8p,
                        _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p],
    "bjx_mhmc_step_diag_masked": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, c_int64,
                                  c_int, c_float, _f32p, _f32p, c_int64, c_float, _f32p, _f32p, _f32p, _f32p,
                                  _f32p, _f32p, _f32p, _f32p, _u8p, _u8p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                  c_void_p],
    "bjx_mhmc_step_diag_coef": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, c_int64,
                                c_int, c_float, _f32p, _f32p, c_int64, c_float, _f32p, _f32p, _f32p, _f32p,
                                _f32p, _f32p, _f32p, _f32p, _u8p, _u8p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                c_void_p, c_float, c_float],
    "bjx_mhmc_finish_masked": [c_void_p, c_int64, c_int64, c_void_p, _f32p, _f32p, _f32p, _f32p, _f32p, _u8p,
                               _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p],
    "bjx_dense_matmul": [c_void_p, c_int64, c_int64, _f32p, _f32p, _f32p],
    "bjx_dense_apply_imm": [c_void_p, c_int64, c_int64, _f32p, _f32p, _f32p],
    "bjx_dense_matmul_bt": [c_void_p, c_int64, c_int64, _f32p, _f32p, _f32p, _f32p],
    "bjx_dense_apply_imm_t": [c_void_p, c_int64, c_int64, _f32p, _f32p, _f32p, _f32p],
    "bjx_hmc_momentum_dense": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64,
                               _f32p, _f32p, _f32p, _f32p, _f32p, _f32p],
    "bjx_leapfrog_dense": [c_void_p, c_int64, c_int64, c_int, c_float, _f32p, _f32p, _f32p, _f32p,
                           _f32p, _f32p, _f32p],
    "bjx_hmc_finish_dense": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, c_float,
                             _f32p, _f32p, c_float,
                             _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                             _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _u8p, _u8p, _f32p],
    "bjx_leapfrog_dense_coef": [c_void_p, c_int64, c_int64, c_int, c_float, c_float, c_float, c_float, _f32p,
                                _f32p, c_int64, _f32p, _f32p, _f32p, _f32p, _f32p, c_void_p, ctypes.c_int32],
    "bjx_hmc_finish_dense_coef": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, c_float,
                                  c_float, _f32p, _f32p, c_int64, c_float,
                                  _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                  _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _u8p, _u8p, _f32p],
    "bjx_mhmc_step_dense": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, c_int64,
                            c_float, _f32p, _f32p, c_int64, c_float, _f32p, _f32p, _f32p, _f32p, _f32p,
                            _f32p, _f32p, _f32p, _f32p, _f32p, _u8p, _u8p, _f32p, _f32p, _f32p, _f32p,
                            _f32p],
    "bjx_hmc_trajectory_diag": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, c_int64,
                                c_float, _f32p, _f32p, c_int64, c_float, ctypes.c_int32, _f32p,
                                _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                _f32p, _u8p, _u8p, _f32p],
    "bjx_mhmc_step_dense_masked": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, c_int64,
                                   c_float, _f32p, _f32p, c_int64, c_float, _f32p, _f32p, _f32p, _f32p, _f32p,
                                   _f32p, _f32p, _f32p, _f32p, _f32p, _u8p, _u8p, _f32p, _f32p, _f32p, _f32p,
                                   _f32p, c_void_p],
    "bjx_mhmc_step_dense_coef": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, c_int64,
                                 c_float, c_float, _f32p, _f32p, c_int64, c_float, _f32p, _f32p, _f32p, _f32p, _f32p,
                                 _f32p, _f32p, _f32p, _f32p, _f32p, _u8p, _u8p, _f32p, _f32p, _f32p, _f32p,
                                 _f32p, c_void_p],
    "bjx_pc_matvec_t": [c_void_p, c_int64, c_int64, _f32p, c_int64, _f32p, _f32p],
    "bjx_hmc_momentum_dense_pc": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64,
                                  _f32p, _f32p, c_int64, _f32p, _f32p, _f32p, _f32p],
    "bjx_leapfrog_dense_pc": [c_void_p, c_int64, c_int64, c_int, c_float, _f32p, _f32p, c_int64, _f32p,
                              _f32p, _f32p, _f32p, _f32p],
    "bjx_hmc_finish_dense_pc": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64,
                                c_float, _f32p, _f32p, c_int64, c_float,
                                _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _u8p, _u8p, _f32p],
    "bjx_welford_update_dense": [c_void_p, c_int64, c_int64, c_int64, _f32p, _f32p, _f32p, _f32p,
                                 _f32p],
    "bjx_welford_final_dense": [c_void_p, c_int64, c_int64, c_int64, c_float, _f32p, _f32p, c_int,
                                _f32p],
    "bjx_da_init": [c_void_p, c_int64, c_int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p],
    "bjx_da_update": [c_void_p, c_int64, c_int64, c_float, c_float, c_float, c_float, _f32p, _f32p,
                      _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p],
    "bjx_exp": [c_void_p, c_int64, _f32p, _f32p],
    "bjx_welford_update_diag": [c_void_p, c_int64, c_int64, c_int64, _f32p, _f32p, _f32p, _f32p, _f32p],
    "bjx_welford_final_diag": [c_void_p, c_int64, c_int64, c_int64, c_float, _f32p, _f32p, c_int64,
                               _f32p],
    "bjx_target_diag_gaussian": [c_void_p, c_int64, c_int64, _f32p, _f32p, _f32p, _f32p],
    "bjx_target_neal_funnel": [c_void_p, c_int64, c_int64, _f32p, _f32p, _f32p],
    "bjx_target_ar1_gaussian": [c_void_p, c_int64, c_int64, c_float, c_float, c_float, _f32p,
                                _f32p, _f32p],
}



class NutsDesc(ctypes.Structure):
    """ctypes mirror of ``bjx_nuts_t`` (include/bjx_nuts.h)."""

    _fields_ = [
        ("N", c_int64), ("D", c_int64), ("max_depth", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("imm", c_void_p), ("imm_stride", c_int64), ("eps_per_chain", c_void_p),
        ("eps", c_float), ("divergence_threshold", c_float),
        ("key0", c_uint32), ("key1", c_uint32), ("chain_offset", c_int64), ("step_fold", c_int64),
        ("q0", c_void_p), ("g0", c_void_p), ("p0", c_void_p),
        ("Lq", c_void_p), ("Lp", c_void_p), ("Lg", c_void_p),
        ("Rq", c_void_p), ("Rp", c_void_p), ("Rg", c_void_p),
        ("msum", c_void_p), ("Smsum", c_void_p),
        ("Pq", c_void_p), ("Pg", c_void_p), ("Sq", c_void_p), ("Sg", c_void_p),
        ("ckpt_r", c_void_p), ("ckpt_rs", c_void_p),
        ("fs", c_void_p), ("is_", c_void_p),
        ("Mdense", c_void_p), ("Mdense_stride", c_int64), ("v0", c_void_p),
        ("Lv", c_void_p), ("Rv", c_void_p), ("ckpt_v", c_void_p),
        ("int_kick", c_float), ("int_drift", c_float), ("v_pre", c_void_p),
    ]


# slot indices of the fs / is tables (include/bjx_nuts.h enums; checked by tests/test_abi.py)
NUTS_F = {"H0": 0, "LLOGP": 1, "RLOGP": 2, "PLOGP": 3, "PENERGY": 4, "PW": 5, "PSLPA": 6,
          "SLOGP": 7, "SENERGY": 8, "SW": 9, "SSLPA": 10, "ACC": 11}
NUTS_NF = 12
NUTS_I = {"ACTIVE": 0, "SUB_ACTIVE": 1, "DIR": 2, "NSTATES": 3, "SUBN": 4, "SDIV": 5, "STURN": 6,
          "DIV": 7, "TURN": 8, "DEPTH": 9, "KT": 10, "KTB": 11, "KP": 12, "KPB": 13, "IK": 14, "IKB": 15,
          "LAZY": 16, "STAGE": 17}
NUTS_NI = 18

SIGNATURES.update({
    "bjx_nuts_init": [c_void_p, POINTER(NutsDesc), _f32p, _f32p],
    "bjx_nuts_pre": [c_void_p, POINTER(NutsDesc), ctypes.c_int32, c_int64, c_int64, c_void_p, _f32p],
    "bjx_nuts_post": [c_void_p, POINTER(NutsDesc), ctypes.c_int32, c_int64, c_int64, c_void_p, _f32p,
                      _f32p, _f32p, ctypes.c_int32],
    "bjx_nuts_dense_kick": [c_void_p, POINTER(NutsDesc), ctypes.c_int32, c_int64, c_int64, c_void_p, c_void_p,
                            _f32p, c_float, _f32p],
    "bjx_nuts_mid": [c_void_p, POINTER(NutsDesc), c_int64, c_void_p, c_void_p, _f32p, _f32p, c_float, c_float],
    "bjx_nuts_pre_ctl": [c_void_p, POINTER(NutsDesc), ctypes.c_int32, c_int64, c_void_p, c_void_p, _f32p],
    "bjx_nuts_post_ctl": [c_void_p, POINTER(NutsDesc), ctypes.c_int32, c_int64, c_void_p, c_void_p, _f32p,
                          _f32p, _f32p, ctypes.c_int32],
    "bjx_nuts_compact": [c_void_p, POINTER(NutsDesc), ctypes.c_int32, c_int64, c_void_p, c_void_p,
                         c_void_p],
    "bjx_nuts_set_ctl": [c_void_p, c_void_p, ctypes.c_int32, c_int64, c_int64, c_uint32, c_uint32,
                         c_int64, c_int64],
    "bjx_nuts_merge": [c_void_p, POINTER(NutsDesc), ctypes.c_int32, c_int64, c_void_p],
})


class NutsAsync(ctypes.Structure):
    """ctypes mirror of ``bjx_nuts_async_t`` (include/bjx_nuts.h): free-running chains."""

    _fields_ = [
        ("step_keys", c_void_p), ("t_first", ctypes.c_int32), ("n_steps", ctypes.c_int32),
        ("q", c_void_p), ("g", c_void_p), ("logp", c_void_p), ("p", c_void_p),
        ("t", c_void_p), ("phase", c_void_p), ("n_done", c_void_p),
        ("rows", c_void_p), ("n_rows", c_int64),
        ("out_position", c_void_p), ("out_logdensity", c_void_p), ("out_acceptance_rate", c_void_p),
        ("out_energy", c_void_p), ("out_num_integration_steps", c_void_p),
        ("out_num_trajectory_expansions", c_void_p), ("out_is_divergent", c_void_p),
        ("out_is_turning", c_void_p),
        ("adapt_tab", c_void_p), ("adapt_target", c_float), ("adapt_reserved", c_float),
        ("adapt_log_x", c_void_p), ("adapt_log_x_avg", c_void_p), ("adapt_avg_err", c_void_p),
        ("adapt_mu", c_void_p), ("adapt_step_size", c_void_p), ("adapt_mean", c_void_p),
        ("adapt_m2", c_void_p), ("adapt_imm", c_void_p), ("out_step_size", c_void_p),
        ("rec", c_void_p), ("front_p", c_void_p), ("end_list", c_void_p), ("end_count", c_void_p),
        ("tick", ctypes.c_int32), ("keep_ends", ctypes.c_int32), ("n_rows_dev", c_void_p),
        ("mass_sqrt_t", c_void_p), ("v0", c_void_p),
        ("target_kind", ctypes.c_int32), ("ticks_per_launch", ctypes.c_int32), ("target_vec", c_void_p),
        ("int_stages", ctypes.c_int32), ("reserved3", ctypes.c_int32),
        ("int_mid_kick", c_float * 6), ("int_mid_drift", c_float * 6),  # BJX_NUTS_MAX_MID
        ("gemm_pc", c_void_p), ("gemm_vc", c_void_p), ("gemm_z", c_void_p), ("gemm_pm", c_void_p),
        ("gemm_vm", c_void_p), ("gemm_cap", ctypes.c_int64), ("gemm_mass_sqrt", c_void_p),
        ("gemm_imm_t", c_void_p),
    ]


# columns of bjx_nuts_async_t.adapt_tab (include/bjx_nuts.h BJX_NUTS_AT_*; checked by tests/test_abi.py)
NUTS_AT = {"FLAGS": 0, "DA_REG": 1, "DA_INV_REG": 2, "DA_ETA": 3, "DA_COEF": 4, "WEL_N": 5,
           "FIN_NM1": 6, "FIN_BETA_DATA": 7, "FIN_BETA_PREV": 8, "FIN_REG": 9}
NUTS_ADAPT_COLS = 12
NUTS_MAX_MID = 6  # BJX_NUTS_MAX_MID: middle stages of a multi-stage integrator in the free-running tick kernels
NUTS_REC_WORDS = 32  # BJX_NUTS_REC_WORDS: packed per-chain record of the low-traffic tick kernels
NUTS_TARGET_USER = 3  # BJX_TARGET_USER (include/bjx_nuts.h): kernels compiled at run time around a user target


SIGNATURES.update({
    "bjx_nuts_async_tick": [c_void_p, POINTER(NutsDesc), POINTER(NutsAsync), _f32p, _f32p, _f32p],
    "bjx_nuts_async_compact": [c_void_p, POINTER(NutsDesc), POINTER(NutsAsync), _f32p, c_void_p, _f32p,
                               c_void_p, c_void_p],
})

class NutsSpec(ctypes.Structure):
    """ctypes mirror of ``bjx_nuts_spec_t`` (include/bjx_nuts.h): the two-stream speculative tail of a run."""

    _fields_ = [
        ("n_rows", c_int64), ("n_rows_dev", c_void_p), ("rows", c_void_p),
        ("ring", ctypes.c_int32), ("lead", ctypes.c_int32),
        ("qf", c_void_p), ("fp", c_void_p),
        ("eLq", c_void_p), ("eLp", c_void_p), ("eLg", c_void_p),
        ("eRq", c_void_p), ("eRp", c_void_p), ("eRg", c_void_p),
        ("iw", c_void_p), ("ring_g", c_void_p), ("ring_tag", c_void_p), ("avail", c_void_p), ("ack", c_void_p),
        ("qf_book", c_void_p), ("bw", c_void_p), ("a_seq", c_void_p), ("dbg", c_void_p),
    ]


NUTS_SPEC_IW = 8   # BJX_NUTS_SPEC_IW
NUTS_SPEC_TAG = 8  # BJX_NUTS_SPEC_TAG

SIGNATURES.update({
    "bjx_nuts_spec_enter": [c_void_p, POINTER(NutsDesc), POINTER(NutsAsync), POINTER(NutsSpec)],
    "bjx_nuts_spec_integrate": [c_void_p, POINTER(NutsDesc), POINTER(NutsAsync), POINTER(NutsSpec), _f32p, _f32p,
                                ctypes.c_int32],
    "bjx_nuts_spec_book": [c_void_p, POINTER(NutsDesc), POINTER(NutsAsync), POINTER(NutsSpec), ctypes.c_int32,
                           ctypes.c_int32],
    "bjx_stream_probe": [c_void_p, c_void_p, c_void_p, ctypes.c_int32],
})

# include/bjx_pool.h (pooled cross-chain statistics; bjx_pool_workspace_bytes returns int64, see load())
_f64p = c_void_p
SIGNATURES.update({
    "bjx_chees_weights": [c_void_p, c_int64, c_int64, _f32p, _f32p, _u8p, _f32p],
    "bjx_chees_colstats": [c_void_p, c_int64, c_int64, _f32p, _f32p, _f32p, c_void_p, _f64p],
    "bjx_chees_weights_colstats": [c_void_p, c_int64, c_int64, _f32p, _f32p, _u8p, _f32p, _f32p, c_void_p, _f64p],
    "bjx_chees_means": [c_void_p, c_int64, _f64p, _f32p, _f32p, _f32p, _f32p],
    "bjx_chees_criterion": [c_void_p, c_int64, c_int64, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                            _f32p],
    "bjx_chees_scalars": [c_void_p, c_int64, _f32p, _u8p, _f32p, c_float, _f64p],
    "bjx_pool_colsum": [c_void_p, c_int64, c_int64, _f32p, _f32p, c_void_p, _f64p],
    "bjx_pool_mean": [c_void_p, c_int64, _f64p, ctypes.c_double, _f32p],
    "bjx_pool_merge_diag": [c_void_p, c_int64, c_float, c_float, _f32p, _f64p, _f32p, _f32p],
    "bjx_pool_final_diag": [c_void_p, c_int64, c_float, _f32p, _f32p],
    "bjx_pool_center": [c_void_p, c_int64, c_int64, _f32p, _f32p, _f32p],
    "bjx_halton_steps": [c_void_p, c_int64, c_void_p, ctypes.c_int32, c_float, c_float, c_float,
                         c_void_p],
})
# include/bjx_ghmc.h (Generalized HMC: persistent momentum, slice accept)
SIGNATURES.update({
    "bjx_ghmc_init": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, _f32p, _f32p],
    "bjx_ghmc_refresh": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, _f32p, c_int64,
                         c_float, _f32p, c_float, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p],
    "bjx_ghmc_refresh_kick": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, _f32p, c_int64,
                              c_float, _f32p, c_float, _f32p, c_float, _f32p] + [_f32p] * 9,
    "bjx_ghmc_finish": [c_void_p, c_int64, c_int64, c_float, _f32p, _f32p, c_int64, c_float]
                       + [_f32p] * 12 + [c_int64, c_int64] + [_f32p] * 6 + [_u8p, _u8p, _f32p, _f32p],
})
SIGNATURES.update({
    "bjx_meads_fold_moments": [c_void_p, c_int64, c_int64, c_int64, _f32p, c_void_p, _f32p, _f32p, _f32p],
    "bjx_meads_fold_build": [c_void_p, c_int64, c_int64, c_int64, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                             c_void_p],
    "bjx_meads_fold_params": [c_void_p, c_int64, c_int64, c_int64, c_int64, c_float, c_float, c_int64, _f32p,
                              c_void_p, _f32p, c_void_p] + [_f32p] * 8,
})
INT64_FUNCTIONS = {"bjx_pool_workspace_bytes": [c_int64, c_int64],
                   "bjx_meads_workspace_bytes": [c_int64, c_int64]}

_lib = None


class BjxError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libbjxhip.so (once) and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BjxError(
            f"{LIB_PATH} not found: the HIP engine is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C blackjax_amd/csrc`). "
            "blackjax_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.bjx_last_error.restype = c_char_p
    lib.bjx_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = c_int
        fn.argtypes = argtypes
    for name, argtypes in INT64_FUNCTIONS.items():
        fn = getattr(lib, name)
        fn.restype = c_int64
        fn.argtypes = argtypes
    _lib = lib
    return lib


class LaunchTimer:
    """Brackets selected entry points with HIP events on the current stream so a caller
    (bench.py) can read per-launch kernel durations of the timed region afterwards."""

    def __init__(self, names, every: int = 1, capacity: int = 0):
        """``every``: bracket only every k-th launch of each name -- an int, or ``{name: k}`` (two event records cost host
        time, which matters once launches are ~50 us).  ``capacity``: number of brackets whose
        events are created up front, outside the timed region (creating an event costs more than
        recording it); brackets beyond the pool create theirs on the fly."""
        import torch

        self.names = set(names)
        self.events = {n: [] for n in self.names}
        # one sampling rate for all names, or {name: rate} (names not listed: every launch)
        self.every = ({n: max(1, int(every.get(n, 1))) for n in self.names} if isinstance(every, dict)
                      else {n: max(1, int(every)) for n in self.names})
        self.seen = {n: 0 for n in self.names}
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * max(0, int(capacity)))]
        for ev in self.pool:  # torch creates the HIP event lazily, at the first record()
            ev.record()

    def take_event(self):
        import torch

        return self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)

    def durations_ms(self, name):
        import torch

        torch.cuda.synchronize()
        return [s.elapsed_time(e) for s, e in self.events[name]]


_timer = None


def set_timer(timer) -> None:
    global _timer
    _timer = timer


def call(name: str, *args) -> None:
    lib = load()
    timed = _timer is not None and name in _timer.names
    if timed:
        _timer.seen[name] += 1
        timed = _timer.seen[name] % _timer.every[name] == 0
    if timed:
        s = _timer.take_event()
        e = _timer.take_event()
        s.record()
        rc = getattr(lib, name)(*args)
        e.record()
        _timer.events[name].append((s, e))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise BjxError(f"{name} failed (rc={rc}): {lib.bjx_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = None  # torch._C._cuda_getCurrentRawStream when this torch has it


def current_stream() -> int:
    """``hipStream_t`` of torch's current stream on the current device, as an int.  Goes through
    torch's raw-stream getter (sub-microsecond) when available: ``torch.cuda.current_stream()``
    builds a Stream object and costs ~8 us per call, a third of the host time of a C2 transition."""
    global _raw_stream
    import torch

    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
