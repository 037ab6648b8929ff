"""ctypes mirror of ``include/gjx.h`` (struct layouts, enums, prototypes).

The same declarations serve the HIP library (``genjax_amd/csrc/libgjx_hip.so``) and, in the
test-suite only, the CPU checker under ``oracle/`` — they share the *ABI*, never code.
"""
from __future__ import annotations

import ctypes as C

ABI_VERSION = 10

# status
OK, EINVAL, EUNSUPPORTED, EHIP, EWORKSPACE = 0, -1, -2, -3, -4

# distribution kinds
NORMAL, FLIP, BERNOULLI_LOGITS, BETA, CATEGORICAL_LOGITS, CATEGORICAL_PROBS = 1, 2, 3, 4, 5, 6
UNIFORM, MVNORMAL_DIAG, EXPONENTIAL, HALF_NORMAL, LAPLACE, LOG_NORMAL, CAUCHY, GAMMA = 7, 8, 9, 10, 11, 12, 13, 14
STUDENT_T, TRUNCATED_NORMAL, POISSON, GEOMETRIC, DIRICHLET, GUMBEL, HALF_CAUCHY, INVERSE_GAMMA = 15, 16, 17, 18, 19, 20, 21, 22
WEIBULL, LOGIT_NORMAL, CHI2 = 23, 24, 25
CHI, EXP_GAMMA, EXP_INVERSE_GAMMA, HALF_STUDENT_T, KUMARASWAMY, MOYAL, TRUNCATED_CAUCHY, DOUBLESIDED_MAXWELL, INVERSE_GAUSSIAN = 26, 27, 28, 29, 30, 31, 32, 33, 34
NEGATIVE_BINOMIAL, VON_MISES = 35, 36
KIND_MAX = 37
KIND_NAMES = {
    NORMAL: "normal", FLIP: "flip", BERNOULLI_LOGITS: "bernoulli", BETA: "beta",
    CATEGORICAL_LOGITS: "categorical", CATEGORICAL_PROBS: "categorical(probs)", UNIFORM: "uniform",
    MVNORMAL_DIAG: "mv_normal_diag", EXPONENTIAL: "exponential", HALF_NORMAL: "half_normal",
    LAPLACE: "laplace", LOG_NORMAL: "log_normal", CAUCHY: "cauchy", GAMMA: "gamma",
    STUDENT_T: "student_t", TRUNCATED_NORMAL: "truncated_normal", POISSON: "poisson", GEOMETRIC: "geometric",
    DIRICHLET: "dirichlet", GUMBEL: "gumbel", HALF_CAUCHY: "half_cauchy", INVERSE_GAMMA: "inverse_gamma",
    WEIBULL: "weibull", LOGIT_NORMAL: "logit_normal", CHI2: "chi2",
    CHI: "chi", EXP_GAMMA: "exp_gamma", EXP_INVERSE_GAMMA: "exp_inverse_gamma", HALF_STUDENT_T: "half_student_t", KUMARASWAMY: "kumaraswamy",
    MOYAL: "moyal", TRUNCATED_CAUCHY: "truncated_cauchy", DOUBLESIDED_MAXWELL: "double_sided_maxwell", INVERSE_GAUSSIAN: "inverse_gaussian",
    NEGATIVE_BINOMIAL: "negative_binomial", VON_MISES: "von_mises",
}
DISCRETE_KINDS = (FLIP, BERNOULLI_LOGITS, CATEGORICAL_LOGITS, CATEGORICAL_PROBS, POISSON, GEOMETRIC, NEGATIVE_BINOMIAL)
NO_GRADIENT_KINDS = DISCRETE_KINDS + (DIRICHLET,)      # values HMC cannot move (integers; simplex-constrained)

# param forms / transforms / modes / flags / rng
P_CONST, P_VALUE, P_GATHER, P_AFFINE, P_VGATHER, P_EXPR = 0, 1, 2, 3, 4, 5
# nodes of a GJX_P_EXPR block (include/gjx.h GJX_E_*)
(E_CONST, E_VALUE, E_ADD, E_SUB, E_MUL, E_DIV, E_NEG, E_EXP, E_LOG, E_SQRT, E_SQUARE, E_TANH, E_SIGMOID, E_SOFTPLUS, E_ABS, E_SIN, E_COS,
 E_LOG1P, E_RECIP, E_MAX, E_MIN, E_GT, E_WHERE, E_LINV, E_LINN) = range(25)
EXPR_MAX_NODES = 96
EXPR_NODE_FLOATS = 6        # {op, a, b, c, da, db}
XF_NONE, XF_EXP, XF_SOFTPLUS, XF_SIGMOID = 0, 1, 2, 3
MODE_SAMPLE, MODE_OBS_TAB, MODE_OBS_SLOT, MODE_OBS_MASK, MODE_INPUT = 0, 1, 2, 3, 4
MODE_OBS_PROPOSED = 5
RUN_LEAVE_TILES, RUN_TIME_DISPATCH, RUN_STORE_INPUTS = 1, 2, 4
SITE_HMC_SELECTED = 1
SITE_PROPOSAL = 2
SITE_CARRIED = 4
RNG_FLAT, RNG_JAX32 = 0, 1
FLAT_SITE_SHIFT, FLAT_MAX_SITES = 22, 1023
OP_RUN, OP_LSE, OP_PICK, OP_RESAMPLE, OP_HMC, OP_SSM = 1, 2, 3, 4, 5, 6
WEIGHTS_GLOBAL_MAX, WEIGHTS_TILE_SCALED = 0, 1      # gjx.h: fixed-point weight schemes of the filter's resampler
WEIGHTS_PLAIN_LAUNCHES = 256                        # OR-ed into the scheme: plain launches only (the repeat after a poll time-out)
MAX_PARAMS = 4

i32, i64, u32, u64, f32, f64 = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_double
vp = C.c_void_p


class GjxParam(C.Structure):
    _fields_ = [("op", i32), ("xf", i32), ("off", i32), ("len", i32), ("slot", i32), ("n", i32),
                ("moff", i32), ("d_off", i32), ("d_slot", i32), ("d_moff", i32), ("pad_", i32 * 2)]


class GjxSite(C.Structure):
    _fields_ = [("kind", i32), ("dim", i32), ("slot", i32), ("mode", i32), ("obs_off", i32),
                ("ncat", i32), ("flags", i32), ("scan", i32), ("plate", i32), ("plate_n", i32), ("d_obs", i32), ("pad_", i32),
                ("p", GjxParam * MAX_PARAMS)]


class GjxProgram(C.Structure):
    _fields_ = [("n_sites", i32), ("n_slots", i32), ("n_tab", i32), ("rng_mode", i32),
                ("sites", vp), ("sites_dev", vp), ("tab", vp), ("tab_dev", vp), ("aux_dev", vp), ("n_aux", i32),
                ("uid", i32)]


class GjxRunResample(C.Structure):
    _fields_ = [("logw", vp), ("tile_S", vp), ("tile_E", vp), ("lse_partials", vp), ("n_partials", i32), ("pad_", i32), ("lse_out", vp),
                ("u", C.c_double), ("ancestors_out", vp), ("status_ws", vp)]


class GjxRunOpts(C.Structure):
    _fields_ = [("flags", i32), ("pad_", i32), ("start_event", vp), ("stop_event", vp), ("in_rows", vp), ("in_stride", i64),
                ("in_ancestors", vp), ("resample", C.POINTER(GjxRunResample))]


class GjxRunInfo(C.Structure):
    _fields_ = [("n_partials", i32), ("engine", i32), ("tiles_offset", i64)]


# gjx_scan_filter: flags, forms, options, record (include/gjx.h)
FILTER_NO_WIDE, FILTER_NO_STEPS, FILTER_NO_ONE_LAUNCH, FILTER_TWO_LAUNCH, FILTER_MULTINOMIAL, FILTER_ABSOLUTE_INPUTS = 1, 2, 3, 4, 8, 16
FILTER_FORM_TWO_LAUNCH, FILTER_FORM_PER_STEP, FILTER_FORM_STEPS, FILTER_FORM_WIDE = 0, 1, 2, 3
FILTER_FORM_NAMES = {0: "two launches per step", 1: "one launch per step", 2: "steps kernel (256 threads x 4 particles)",
                     3: "filter kernel on the shared skeleton (16 waves per tile)"}


PEER_VERIFY_ON, PEER_DATA_FINE, PEER_VERIFY_FAULTY = 1, 2, 4


class GjxFilterOpts(C.Structure):
    _fields_ = [("flags", i32), ("coresident_blocks", i32), ("timeline", vp), ("timeline_bytes", i64), ("n_moves", i32), ("move_scale", f32),
                ("accepted_total", vp), ("hmc_targets", vp), ("hmc_eps", f32), ("hmc_L", i32), ("hmc_rows", vp), ("hmc_out", vp),
                ("hmc_workspace", vp), ("hmc_workspace_bytes", C.c_size_t)]


class GjxFilterInfo(C.Structure):
    _fields_ = [("form", i32), ("launches", i32), ("grid", i32), ("tiles_per_block", i32)]


class GjxSsm(C.Structure):
    _fields_ = [("dx", i32), ("dy", i32), ("A_dev", vp), ("H_dev", vp), ("q", f32), ("r", f32),
                ("q0", f32)]


MAX_RANKS = 64


class GjxShardPlan(C.Structure):
    _fields_ = [("base", C.c_uint64), ("total", C.c_uint64), ("slot0", i64), ("n_valid", i64), ("own_lo", i64),
                ("own_n", i64), ("keep_lo", i64), ("keep_hi", i64), ("n_ranks", i64), ("status", i64),
                ("seq", i64), ("reserved", i64), ("bounds", i64 * (MAX_RANKS + 1))]


assert C.sizeof(GjxParam) == 48 and C.sizeof(GjxSite) == 240 and C.sizeof(GjxShardPlan) == 8 * (12 + MAX_RANKS + 1)

PP = C.POINTER(GjxProgram)

# name -> (restype, argtypes) for every symbol include/gjx.h declares
PROTOTYPES = {
    "gjx_version": (C.c_int, []),
    "gjx_host_threefry2x32": (C.c_uint64, [u32, u32, u32, u32]),
    "gjx_last_error": (C.c_char_p, []),
    "gjx_program_engine": (C.c_int, [PP]),
    "gjx_program_source": (i64, [PP, i32, C.c_char_p, i64]),
    "gjx_program_precompile": (C.c_int, [PP, i32]),
    "gjx_program_hmc_source": (i64, [PP, C.c_char_p, i64]),
    "gjx_program_hmc_precompile": (C.c_int, [PP]),
    "gjx_jit_stats": (C.c_int, [vp]),
    "gjx_program_aux_floats": (C.c_int, [PP]),
    "gjx_program_prepare": (C.c_int, [PP, vp, i32, vp]),
    "gjx_threefry2x32": (C.c_int, [u32, u32, u32, u32, i64, vp, vp]),
    "gjx_run_program": (C.c_int, [PP, u32, u32, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp,
                                  C.c_size_t, vp]),
    "gjx_run_program_ex": (C.c_int, [PP, u32, u32, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, C.c_size_t, vp,
                                     C.POINTER(GjxRunOpts), C.POINTER(GjxRunInfo)]),
    "gjx_importance_step": (C.c_int, [PP, u32, u32, i64, i64, vp, vp, vp, vp, f64, vp, vp, vp, C.c_size_t, vp]),
    "gjx_importance_step_ex": (C.c_int, [PP, u32, u32, i64, i64, vp, vp, vp, vp, f64, vp, vp, vp, C.c_size_t, vp, vp, vp]),
    "gjx_workspace_bytes": (C.c_size_t, [C.c_int, i64]),
    "gjx_workspace_status": (C.c_int, [vp, C.POINTER(i32), vp]),
    "gjx_debug_timeline": (C.c_int, [vp, C.c_size_t]),
    "gjx_logsumexp": (C.c_int, [vp, i64, i64, vp, vp, C.c_size_t, vp]),
    "gjx_lse_combine": (C.c_int, [vp, C.c_int, i64, vp, vp]),
    "gjx_categorical_pick": (C.c_int, [vp, i64, i64, vp, u32, u32, i32, vp, vp, C.c_size_t, vp]),
    "gjx_trials_lse_pick": (C.c_int, [vp, i64, i64, i64, u32, u32, i32, vp, vp, vp]),
    "gjx_weight_cumsum": (C.c_int, [vp, i64, i32, vp, i32, vp, vp, vp, i64, vp, C.c_size_t, vp]),
    "gjx_event_create": (C.c_int, [C.POINTER(vp)]),
    "gjx_event_destroy": (C.c_int, [vp]),
    "gjx_event_elapsed_us": (C.c_int, [vp, vp, C.POINTER(f32)]),
    "gjx_run_partials_count": (C.c_int, [PP, i64, i64]),
    "gjx_resample_gather_tiled": (C.c_int, [vp, i64, vp, vp, i32, vp, i32, f64, vp, i64, i32, vp, i64, vp, vp, i64, vp, C.c_size_t, vp]),
    "gjx_resample_systematic": (C.c_int, [vp, i64, vp, f64, i64, i64, i64, vp, vp]),
    "gjx_resample_indices": (C.c_int, [vp, i64, i32, vp, i32, f64, i64, vp, vp, vp, vp, i64, vp, C.c_size_t, vp]),
    "gjx_resample_gather": (C.c_int, [vp, i64, i32, vp, i32, f64, vp, i64, i32, vp, i64, vp, vp, i64, vp, C.c_size_t, vp]),
    "gjx_resample_gather_systematic": (C.c_int, [vp, i64, vp, f64, i64, i64, i64, vp, i64, i32, vp, i64, vp, vp]),
    "gjx_resample_multinomial": (C.c_int, [vp, i64, vp, u32, u32, i64, i64, i64, vp, vp]),
    "gjx_gather_rows": (C.c_int, [vp, i64, vp, i64, i32, vp, i64, vp]),
    "gjx_shard_plan_build": (C.c_int, [vp, i32, i32, f64, i64, i64, vp, vp, vp]),
    "gjx_rccl_unique_id": (C.c_int, [C.c_char_p, vp]),
    "gjx_shard_ctx_create": (C.c_int, [C.c_char_p, vp, i32, i32, i64, i32, i64, C.POINTER(vp)]),
    "gjx_shard_ctx_destroy": (C.c_int, [vp]),
    "gjx_shard_resample_step": (C.c_int, [vp, vp, vp, vp, i64, vp, i64, f64, vp, vp, vp]),
    "gjx_shard_message_counts": (C.c_int, [vp, i32, i64, vp, vp, vp]),
    "gjx_shard_pack": (C.c_int, [vp, i64, i32, vp, i64, i64, i64, vp, vp]),
    "gjx_shard_unpack": (C.c_int, [vp, i64, i64, i32, vp, i64, i64, vp]),
    "gjx_shard_resample": (C.c_int, [vp, i64, vp, f64, i64, vp, i64, vp, i64, i32, vp, i64, i64, vp]),
    "gjx_gather_rows_strided": (C.c_int, [vp, i64, i64, vp, i64, i32, vp, i64, i64, vp]),
    "gjx_ssm_step": (C.c_int, [C.POINTER(GjxSsm), u32, u32, i32, i32, i64, i64, vp, i64, vp, vp, vp,
                               vp, vp, i64, vp, C.c_size_t, vp]),
    "gjx_ssm_step_move": (C.c_int, [C.POINTER(GjxSsm), u32, u32, i32, i32, i64, i64, vp, vp, i64, vp, vp, vp, i32, f32, vp, vp,
                                    vp, vp, vp, vp, i64, vp, C.c_size_t, vp]),
    "gjx_ssm_filter": (C.c_int, [C.POINTER(GjxSsm), u32, u32, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "gjx_ssm_filter_scheme": (C.c_int, [C.POINTER(GjxSsm), u32, u32, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, i32, vp, C.c_size_t, vp]),
    "gjx_ssm_filter_move": (C.c_int, [C.POINTER(GjxSsm), u32, u32, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, vp, vp,
                                      C.c_size_t, vp]),
    "gjx_scan_filter": (C.c_int, [vp, i32, u32, u32, i64, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp, vp, vp]),
    "gjx_scan_filter_history": (C.c_int, [vp, i32, u32, u32, i64, vp, i32, vp, vp, vp, vp, C.c_size_t, vp, vp, vp]),
    "gjx_program_filter_source": (i64, [PP, i32, C.c_char_p, i64]),
    "gjx_program_filter_precompile": (C.c_int, [PP, i32]),
    "gjx_resample_indices_tiled": (C.c_int, [vp, i64, f64, i64, vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "gjx_mh_accept": (C.c_int, [vp, i64, u32, u32, vp, vp, i64, i32, vp, vp, vp]),
    "gjx_resample_sorted_multinomial_tiled": (C.c_int, [vp, i64, u32, u32, i64, vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "gjx_ssm_filter_sharded": (C.c_int, [C.POINTER(GjxSsm), u32, u32, i32, i32, vp, i64, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "gjx_shard_ctx_shape": (C.c_int, [vp, vp]),
    "gjx_shard_ctx_stats": (C.c_int, [vp, vp]),
    "gjx_shard_resample_multinomial_step": (C.c_int, [vp, vp, vp, vp, i64, vp, i64, u32, u32, vp, vp, vp]),
    "gjx_shard_global_lse": (C.c_int, [vp, vp, vp, vp]),
    "gjx_scan_filter_peer": (C.c_int, [vp, vp, i32, u32, u32, vp, vp, vp, C.c_size_t, vp, vp]),
    "gjx_scan_filter_peer_prepare": (C.c_int, [vp, vp, i32, vp]),
    "gjx_scan_filter_peer_opts": (C.c_int, [vp, vp, i32, u32, u32, vp, vp, vp, C.c_size_t, vp, vp, vp]),
    "gjx_scan_filter_peer_prepare_opts": (C.c_int, [vp, vp, i32, vp, vp]),
    "gjx_peer_ctx_create": (C.c_int, [i32, i32, i64, i32, i32, C.POINTER(vp)]),
    "gjx_peer_ctx_create_ex": (C.c_int, [i32, i32, i64, i32, i32, i32, C.POINTER(vp)]),
    "gjx_peer_ctx_export": (C.c_int, [vp, vp]),
    "gjx_peer_ctx_connect": (C.c_int, [vp, vp]),
    "gjx_peer_ctx_buffers": (C.c_int, [vp, vp]),
    "gjx_peer_ctx_status": (C.c_int, [vp, C.POINTER(i32), vp]),
    "gjx_peer_ctx_destroy": (C.c_int, [vp]),
    "gjx_ssm_filter_peer": (C.c_int, [C.POINTER(GjxSsm), u32, u32, i32, i32, vp, vp, vp, vp, vp]),
    "gjx_ssm_filter_peer_move": (C.c_int, [C.POINTER(GjxSsm), u32, u32, i32, i32, vp, vp, vp, vp, i32, f32, vp, vp]),
    "gjx_peer_resample_gather": (C.c_int, [vp, i32, vp, i32, f64, vp, i64, vp, vp, vp]),
    "gjx_hmc_workspace_bytes": (C.c_size_t, [PP, i64]),
    "gjx_hmc_engine": (C.c_int, [PP]),
    "gjx_hmc": (C.c_int, [PP, u32, u32, i64, i64, f32, i32, i32, i32, vp, vp, vp, vp, vp,
                          C.c_size_t, vp]),
    "gjx_score_grad": (C.c_int, [PP, i64, vp, vp, vp, vp]),
}


def bind(lib: C.CDLL, prototypes=PROTOTYPES) -> C.CDLL:
    """Attach restype/argtypes; raises AttributeError if a declared symbol is missing."""
    for name, (res, args) in prototypes.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
