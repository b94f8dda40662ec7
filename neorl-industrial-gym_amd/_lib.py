"""ctypes binding of include/nig.h (libnig.so).  No torch types cross this boundary:
device pointers are plain integers (tensor.data_ptr()), streams are hipStream_t handles.

The extension is mandatory: importing this module without a built libnig.so raises.
"""
import ctypes as C
import os

# torch ships its own HIP runtime (libamdhip64).  It must be in the process BEFORE libnig.so is
# loaded so that libnig.so binds to that same runtime: with the opposite order the loader pulls the
# system ROCm copy for libnig.so, two runtimes coexist, and the second one to initialise reports
# "no ROCm-capable device is detected".
import torch  # noqa: F401,E402

from . import _build

NIG_OK = 0
F_AUTORESET, F_TALLY = 0x1, 0x2
FLAG_TERMINATED, FLAG_TRUNCATED = 0x1, 0x2
FLAG_VIOL_SHIFT, FLAG_NVIOL_SHIFT, FLAG_NCRIT_SHIFT = 2, 5, 7
FLAG_SHUTDOWN, FLAG_DID_RESET, FLAG_INACTIVE, FLAG_STEP_SHIFT = 0x200, 0x400, 0x800, 16
FLAG_VIOL3, FLAG_NVIOL_HI = 0x1000, 0x2000
CTR_STEP_MASK, CTR_DONE, CTR_VIOL_SHIFT = 0x7FFF, 0x8000, 16
MAX_EPISODE_STEPS = 21845
(T_EPISODES, T_RET_SUM, T_RET_SQ, T_RET_MIN, T_RET_MAX, T_LEN_SUM, T_LEN_SQ, T_VIOL, T_CRIT,
 T_SHUTDOWN, T_SUCCESS, T_SATISFIED, T_CONSTRAINTS, T_ROWS) = range(14)

TUNE_SPLIT_BLOCKS = 0
TUNE_WIDE_MIN_BLOCKS = 1
TUNE_DIAG_RING_FAULT = 2

SYMBOLS = [
    "nig_version", "nig_last_error", "nig_env_id", "nig_env_name", "nig_env_spec_get", "nig_layout_query",
    "nig_create", "nig_destroy", "nig_get_layout", "nig_workspace", "nig_get_counter", "nig_set_counter",
    "nig_set_constraint_mask", "nig_reset", "nig_step", "nig_fill_actions", "nig_set_state", "nig_get_state",
    "nig_get_safety_metrics", "nig_reduce_tally", "nig_plan_create", "nig_plan_launch", "nig_plan_destroy", "nig_rollout", "nig_rollout_noise", "nig_bind_state", "nig_set_policy", "nig_rollout_policy", "nig_set_mlp_policy", "nig_rollout_mlp", "nig_reset_host", "nig_step_host",
    "nig_step64", "nig_step_host64", "nig_reduce_metrics",
    "nig_create_mixed", "nig_mixed_destroy", "nig_mixed_get_info", "nig_mixed_state", "nig_mixed_segment", "nig_mixed_reset",
    "nig_mixed_fill_actions", "nig_mixed_rollout", "nig_rollout_mixed", "nig_mixed_step", "nig_rollout_mixed_obs", "nig_mixed_rollout_obs",
    "nig_tune", "nig_tune_get", "nig_handle_tune_get", "nig_clock_stamp",
]


class EnvSpec(C.Structure):
    _fields_ = [("state_dim", C.c_int32), ("action_dim", C.c_int32), ("n_constraints", C.c_int32),
                ("max_episode_steps", C.c_int32), ("k_step", C.c_int32), ("k_reset", C.c_int32),
                ("dt", C.c_double), ("penalty", C.c_double * 3), ("critical", C.c_int32 * 3),
                ("reward_is_f32", C.c_int32)]


class Layout(C.Structure):
    _fields_ = [("batch", C.c_int64), ("ld", C.c_int64), ("bytes", C.c_int64), ("off_state", C.c_int64),
                ("off_ctr", C.c_int64), ("off_life_viol", C.c_int64), ("off_ep_return", C.c_int64),
                ("off_tally", C.c_int64)]


class Policy(C.Structure):
    """nig_policy (include/nig.h, "nig-policy-v1")."""
    _fields_ = [("kind", C.c_int32), ("colmask", C.c_uint32), ("Wt", (C.c_float * 10) * 32), ("b", C.c_float * 10),
                ("sigma", C.c_float * 10), ("half_range", C.c_float * 10), ("p_uniform", C.c_float),
                ("uniform_range", C.c_float), ("clip_lo", C.c_float), ("clip_hi", C.c_float),
                ("kp", C.c_float), ("ki", C.c_float), ("kd", C.c_float), ("setpoint", C.c_float * 10)]


POLICY_AFFINE, POLICY_PID = 1, 2


MIXED_MAX_SEGMENTS = 12


class MixedInfo(C.Structure):
    """nig_mixed_info (include/nig.h)."""
    _fields_ = [("n_segments", C.c_int32), ("state_dim_max", C.c_int32), ("action_dim_max", C.c_int32),
                ("reserved", C.c_int32), ("lanes", C.c_int64), ("ld", C.c_int64),
                ("env", C.c_int32 * MIXED_MAX_SEGMENTS), ("offset", C.c_int64 * MIXED_MAX_SEGMENTS),
                ("count", C.c_int64 * MIXED_MAX_SEGMENTS)]


class NigError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libnig.so.  A stale or missing library is rebuilt first (content hash of csrc/), except
    inside a profiled process or with NIG_NO_AUTOBUILD set, where that is an ImportError (_build.ensure)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    override = os.environ.get("NIG_LIB_PATH")
    if override:
        # an experiment's library variant (profiles/mkvariant.sh) for same-box A/B runs: loaded in place of libnig.so,
        # which is never overwritten (an interrupted A/B script used to leave a variant behind as "current")
        path = override
    else:
        _build.ensure()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
    L = C.CDLL(path)
    vp, i64, u32, u64, i32 = C.c_void_p, C.c_int64, C.c_uint32, C.c_uint64, C.c_int32
    L.nig_version.restype = C.c_char_p
    L.nig_last_error.restype = C.c_char_p
    L.nig_tune.argtypes = [C.c_int32, C.c_int64]
    L.nig_tune_get.argtypes = [C.c_int32]
    L.nig_tune_get.restype = C.c_int64
    L.nig_handle_tune_get.argtypes = [C.c_void_p, C.c_int32]
    L.nig_handle_tune_get.restype = C.c_int64
    L.nig_clock_stamp.argtypes = [C.c_void_p, C.c_void_p]
    L.nig_env_id.argtypes = [C.c_char_p]
    L.nig_env_name.restype = C.c_char_p
    L.nig_env_name.argtypes = [C.c_int]
    L.nig_env_spec_get.argtypes = [C.c_int, C.POINTER(EnvSpec)]
    L.nig_layout_query.argtypes = [C.c_int, i64, u32, C.POINTER(Layout)]
    L.nig_create.argtypes = [C.c_int, i64, C.c_int, u64, u64, i32, C.c_double, u32, vp, C.POINTER(vp)]
    L.nig_destroy.argtypes = [vp]
    L.nig_get_layout.argtypes = [vp, C.POINTER(Layout)]
    L.nig_workspace.restype = vp
    L.nig_workspace.argtypes = [vp]
    L.nig_get_counter.argtypes = [vp, C.POINTER(u32)]
    L.nig_set_counter.argtypes = [vp, u32]
    L.nig_set_constraint_mask.argtypes = [vp, u32]
    L.nig_reset.argtypes = [vp, vp, vp, i64, vp]
    L.nig_step.argtypes = [vp, vp, i64, vp, vp, i64, vp, vp, vp, vp, i64, vp]
    L.nig_fill_actions.argtypes = [vp, u32, vp, i64, vp]
    L.nig_set_state.argtypes = [vp, vp, i64, vp, vp]
    L.nig_get_state.argtypes = [vp, vp, i64, vp, vp]
    L.nig_get_safety_metrics.argtypes = [vp, vp, vp, i64, vp]
    L.nig_reduce_tally.argtypes = [vp, vp, vp]
    L.nig_plan_create.argtypes = [vp, i32, vp, i64, i64, i32, vp, vp, i64, C.POINTER(vp)]
    L.nig_plan_launch.argtypes = [vp, vp]
    L.nig_plan_destroy.argtypes = [vp]
    L.nig_bind_state.argtypes = [vp, vp, i64]
    L.nig_set_policy.argtypes = [vp, C.POINTER(Policy), vp]
    L.nig_rollout_policy.argtypes = [vp, i32, vp, vp, i64, vp, i64, vp, i64, i64, vp]
    L.nig_set_mlp_policy.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.nig_rollout_mlp.argtypes = [vp, i32, vp, vp, i64, vp, i64, vp, i64, i64, vp]
    L.nig_reset_host.argtypes = [vp, vp, vp, vp]
    L.nig_step_host.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.nig_step_host64.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.nig_reduce_metrics.argtypes = [C.POINTER(vp), i32, vp, vp, i64, vp, vp]
    L.nig_step64.argtypes = [vp, vp, i64, vp, vp, i64, vp, vp, vp, vp, i64, vp]
    L.nig_rollout.argtypes = [vp, i32, vp, i64, i64, i32, vp, vp, i64, vp, i64, i64, vp]
    L.nig_rollout_noise.argtypes = [vp, i32, vp, i64, i64, i32, vp, i64, vp, i64, i64, vp, vp, i64, vp, i64, vp]
    L.nig_create_mixed.argtypes = [i32, C.POINTER(i32), C.POINTER(i64), C.c_int, u64, u64, u32, C.POINTER(vp)]
    L.nig_mixed_destroy.argtypes = [vp]
    L.nig_mixed_get_info.argtypes = [vp, C.POINTER(MixedInfo)]
    L.nig_mixed_state.restype = vp
    L.nig_mixed_state.argtypes = [vp]
    L.nig_mixed_segment.restype = vp
    L.nig_mixed_segment.argtypes = [vp, i32]
    L.nig_mixed_reset.argtypes = [vp, vp]
    L.nig_mixed_fill_actions.argtypes = [vp, u32, vp, vp]
    L.nig_mixed_rollout.argtypes = [vp, i32, vp, i64, i32, vp, vp, i64, vp]
    L.nig_mixed_step.argtypes = [vp, vp, vp, vp, vp]
    L.nig_rollout_mixed.argtypes = [C.POINTER(vp), C.POINTER(i64), i32, i32, vp, i64, i64, i32, vp, vp, i64, vp]
    L.nig_rollout_mixed_obs.argtypes = [C.POINTER(vp), C.POINTER(i64), i32, i32, vp, i64, i64, i32, vp, vp, i64, vp, i64, i64, vp]
    L.nig_mixed_rollout_obs.argtypes = [vp, i32, vp, i64, i32, vp, vp, i64, vp, i64, vp]
    _lib = L
    return L


def check(status):
    if status != NIG_OK:
        raise NigError(f"libnig error {status}: {lib().nig_last_error().decode()}")


def env_spec(env_id: int) -> EnvSpec:
    s = EnvSpec()
    check(lib().nig_env_spec_get(env_id, C.byref(s)))
    return s


def layout_query(env_id: int, batch: int, flags: int) -> Layout:
    lay = Layout()
    check(lib().nig_layout_query(env_id, batch, flags, C.byref(lay)))
    return lay
