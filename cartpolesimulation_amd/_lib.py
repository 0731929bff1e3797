"""ctypes binding of libcpmppi.so (include/cpmppi.h).  There is NO fallback: a missing library is an ImportError."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CPMPPI_LIB") or os.path.join(_HERE, "libcpmppi.so")   # CPMPPI_LIB: development builds (tools/)

ABI_VERSION = 5
COST_QBGM, COST_DEFAULT, COST_LEGACY, COST_QBG, COST_QB, COST_QB_NONCONVEX = 0, 1, 2, 3, 4, 5
REDUCE_SUM, REDUCE_MEAN = 0, 1
CONTROL_CLIP, CONTROL_PENALISE = 0, 1
SHIFT_REPEAT_LAST, SHIFT_APPEND_ZERO, SHIFT_NONE = 0, 1, 2
CORRECTION_U_RUN, CORRECTION_U_NOM = 0, 1
MATH_PRECISE, MATH_FAST = 0, 1
ODE_V0, ODE_CROMER = 0, 1
NOISE_DELTA_U, NOISE_KNOTS, NOISE_PHILOX, NOISE_DELTA_U_TILED = 0, 1, 2, 3

EXPORTS = ("cpmppi_create", "cpmppi_destroy", "cpmppi_last_error", "cpmppi_get_config", "cpmppi_set_cost_weights", "cpmppi_set_pole_mass",
           "cpmppi_sample", "cpmppi_interpolate", "cpmppi_predict", "cpmppi_trajectory_cost", "cpmppi_step",
           "cpmppi_reward_weighted_average", "cpmppi_plant_advance", "cpmppi_plant_advance_record", "cpmppi_step_host", "cpmppi_set_profiling", "cpmppi_get_profile",
           "cpmppi_set_gru", "cpmppi_gru_predict", "cpmppi_rollout_cost", "cpmppi_cem_sample", "cpmppi_cem_update",
           "cpmppi_rollout_cost_grad", "cpmppi_adam_step", "cpmppi_sgd_step", "cpmppi_version", "cpmppi_tiled_floats",
           "cpmppi_sample_tiled", "cpmppi_tile_delta_u", "cpmppi_cem_gmm_sample", "cpmppi_comm_unique_id",
           "cpmppi_comm_init", "cpmppi_comm_gather", "cpmppi_comm_wait", "cpmppi_comm_sync", "cpmppi_comm_destroy",
           "cpmppi_step_gather", "cpmppi_last_launch", "cpmppi_comm_set_timeout", "cpmppi_write_recordings", "cpmppi_plant_step",
           "cpmppi_abi_version", "cpmppi_stream_create", "cpmppi_stream_destroy", "cpmppi_comm_get_info",
           "cpmppi_groups_create", "cpmppi_groups_destroy", "cpmppi_groups_count", "cpmppi_groups_slice", "cpmppi_groups_handle",
           "cpmppi_groups_stream", "cpmppi_groups_fork", "cpmppi_groups_join", "cpmppi_groups_run", "cpmppi_groups_last_error",
           "cpmppi_comm_set_stamped", "cpmppi_groups_comm_init", "cpmppi_groups_run_gather")
COMM_ID_BYTES, COMM_SLOTS, GATHER_STAMP_FLOATS = 128, 4, 4


class cpmppi_config(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("E", C.c_uint32), ("N", C.c_uint32), ("H", C.c_uint32),
                ("S", C.c_uint32), ("dt", C.c_float),
                ("k", C.c_float), ("m_cart", C.c_float), ("m_pole", C.c_float), ("g", C.c_float),
                ("J_fric", C.c_float), ("M_fric", C.c_float), ("u_max", C.c_float), ("track_half_length", C.c_float),
                ("L_default", C.c_float),
                ("cost_id", C.c_uint32), ("cost_w", C.c_float * 24),
                ("R", C.c_float), ("LBD", C.c_float), ("NU", C.c_float), ("cc_weight", C.c_float),
                ("sigma", C.c_float), ("period", C.c_uint32), ("action_low", C.c_float), ("action_high", C.c_float),
                ("horizon_reduce", C.c_uint32), ("control_mode", C.c_uint32), ("shift_mode", C.c_uint32),
                ("correction_u", C.c_uint32), ("math_mode", C.c_uint32), ("rollouts_per_lane", C.c_uint32),
                ("ode_predictor", C.c_uint32)]


class cpmppi_step_args(C.Structure):
    _fields_ = [("E", C.c_uint32), ("s0", C.c_void_p), ("u_nom", C.c_void_p), ("u_prev", C.c_void_p),
                ("target_position", C.c_void_p), ("target_equilibrium", C.c_void_p), ("L", C.c_void_p),
                ("noise_kind", C.c_uint32), ("noise", C.c_void_p), ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("env_offset", C.c_uint32), ("Q_out", C.c_void_p), ("S_out", C.c_void_p),
                ("predictor", C.c_uint32), ("h0", C.c_void_p), ("previous_input", C.c_void_p), ("offset_dev", C.c_void_p),
                ("u_nom_out", C.c_void_p)]


class cpmppi_gru_model(C.Structure):
    _FP = C.POINTER(C.c_float)
    _fields_ = [("inputs", C.c_uint32), ("hidden", C.c_uint32), ("layers", C.c_uint32), ("outputs", C.c_uint32),
                ("w_ih", _FP * 2), ("w_hh", _FP * 2), ("b_ih", _FP * 2), ("b_hh", _FP * 2), ("w_out", _FP), ("b_out", _FP),
                ("in_scale", _FP), ("in_shift", _FP), ("out_scale", _FP), ("out_shift", _FP)]


class cpmppi_launch_info(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("cost_id", "math_mode", "noise_kind", "rollouts_per_lane", "build_variant",
                                          "ode_predictor", "blocks", "cost_plugin")]


class cpmppi_plant_args(C.Structure):
    _fields_ = [("E", C.c_uint32), ("s", C.c_void_p), ("Q", C.c_void_p), ("L", C.c_void_p),
                ("n_substeps", C.c_uint32), ("period_steps", C.c_uint32), ("dt_sim", C.c_float),
                ("period", C.c_uint64), ("period_dev", C.c_void_p),
                ("states_log", C.c_void_p), ("dd_log", C.c_void_p), ("save_rows", C.c_uint64), ("save_every", C.c_uint32),
                ("Q_log", C.c_void_p), ("ctrl_rows", C.c_uint64),
                ("target_position_table", C.c_void_p), ("target_equilibrium_table", C.c_void_p), ("L_table", C.c_void_p),
                ("sched_rows", C.c_uint64), ("sched_stride", C.c_uint32),
                ("target_position_out", C.c_void_p), ("target_equilibrium_out", C.c_void_p), ("L_out", C.c_void_p),
                ("row_envs", C.c_uint32),
                ("m_pole", C.c_void_p), ("m_pole_table", C.c_void_p), ("L_controller_table", C.c_void_p),
                ("Q_disturbance_table", C.c_void_p), ("Q_bias", C.c_float), ("Q_applied_out", C.c_void_p),
                ("s_measured", C.c_void_p), ("state_history", C.c_void_p), ("history_len", C.c_uint32), ("latency_steps", C.c_uint32),
                ("latency_frac", C.c_double), ("measurement_noise_table", C.c_void_p), ("angle_offset_table", C.c_void_p),
                ("informed_table", C.c_void_p)]


class cpmppi_recording(C.Structure):
    _fields_ = [("E", C.c_uint32), ("rows", C.c_uint32), ("time", C.c_void_p), ("states", C.c_void_p), ("dd", C.c_void_p),
                ("Q", C.c_void_p), ("Q_ccrc", C.c_void_p), ("target_position", C.c_void_p), ("target_equilibrium", C.c_void_p),
                ("L", C.c_void_p), ("m_pole", C.c_double), ("u_max", C.c_float), ("first_update_row", C.c_uint32),
                ("q_update_time", C.c_double), ("m_pole_rows", C.c_void_p), ("informed", C.c_void_p), ("Q_applied", C.c_void_p), ("angle_offset", C.c_void_p)]


class cpmppi_comm_info(C.Structure):
    _fields_ = [("world", C.c_uint32), ("rank", C.c_uint32), ("rccl_ranks", C.c_int32), ("rccl_rank", C.c_int32),
                ("rccl_version", C.c_int32), ("stream_memory_ops", C.c_uint32), ("gathers_enqueued", C.c_uint32),
                ("stamped", C.c_uint32)]


PREDICTOR_ODE_V0, PREDICTOR_GRU = 0, 1


class CpmppiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libcpmppi error {code}: {message}")
        self.code = code


_lib = None


def load():
    """Load libcpmppi.so (built by __graft_entry__.build()).  Raises ImportError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  cartpolesimulation_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7) but its libraries ask for it by file name, so
    # if the system runtime were loaded first the process would end up with two HIP runtimes.  Importing torch first
    # makes libcpmppi.so (NEEDED libamdhip64.so.7) bind to the runtime torch already loaded: one runtime, shared
    # streams and device pointers.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, f = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
    lib.cpmppi_create.argtypes = [C.POINTER(cpmppi_config), C.c_int, C.POINTER(vp)]
    lib.cpmppi_destroy.argtypes = [vp]
    lib.cpmppi_destroy.restype = None
    lib.cpmppi_last_error.argtypes = [vp]
    lib.cpmppi_last_error.restype = C.c_char_p
    lib.cpmppi_get_config.argtypes = [vp, C.POINTER(cpmppi_config)]
    lib.cpmppi_set_cost_weights.argtypes = [vp, u32, C.POINTER(f), u32]
    lib.cpmppi_set_pole_mass.argtypes = [vp, f]
    lib.cpmppi_sample.argtypes = [vp, u32, u64, u64, u32, vp, vp, vp]
    lib.cpmppi_interpolate.argtypes = [vp, u32, vp, vp, vp]
    lib.cpmppi_predict.argtypes = [vp, u32, u32, vp, vp, vp, vp, vp]
    lib.cpmppi_trajectory_cost.argtypes = [vp, u32, u32, vp, vp, f, f, vp, vp, vp, vp, vp, vp]
    lib.cpmppi_step.argtypes = [vp, C.POINTER(cpmppi_step_args), vp]
    lib.cpmppi_reward_weighted_average.argtypes = [vp, u32, vp, vp, vp, vp]
    lib.cpmppi_plant_advance.argtypes = [vp, u32, vp, vp, vp, u32, f, vp]
    lib.cpmppi_plant_advance_record.argtypes = [vp, u32, vp, vp, vp, u32, f, vp, vp, u64, u64, vp, vp]
    lib.cpmppi_step_host.argtypes = [vp, u32, vp, vp, vp, vp, vp, u64, u64, u32, vp, vp]
    lib.cpmppi_set_profiling.argtypes = [vp, C.c_int]
    lib.cpmppi_get_profile.argtypes = [vp, C.POINTER(f), C.POINTER(f), u32, C.POINTER(u32)]
    lib.cpmppi_set_gru.argtypes = [vp, C.POINTER(cpmppi_gru_model)]
    lib.cpmppi_gru_predict.argtypes = [vp, u32, u32, vp, vp, vp, vp, vp, vp]
    lib.cpmppi_rollout_cost.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp, vp]
    lib.cpmppi_rollout_cost_grad.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cpmppi_adam_step.argtypes = [vp, u32, vp, vp, vp, vp, u32, f, f, f, f, f, vp]
    lib.cpmppi_sgd_step.argtypes = [vp, u32, vp, vp, f, f, vp]
    lib.cpmppi_cem_sample.argtypes = [vp, u32, vp, vp, u64, u64, u32, vp, vp]
    lib.cpmppi_cem_update.argtypes = [vp, u32, vp, vp, u32, f, vp, vp, vp, vp]
    lib.cpmppi_tiled_floats.argtypes = [vp, u32]
    lib.cpmppi_tiled_floats.restype = C.c_size_t
    lib.cpmppi_sample_tiled.argtypes = [vp, u32, u64, u64, u32, vp, vp, vp]
    lib.cpmppi_tile_delta_u.argtypes = [vp, u32, vp, vp, vp]
    lib.cpmppi_cem_gmm_sample.argtypes = [vp, u32, vp, u32, vp, u64, u64, u32, vp, vp, vp]
    lib.cpmppi_comm_unique_id.argtypes = [vp, C.c_char_p]
    lib.cpmppi_comm_init.argtypes = [vp, vp, C.c_int, C.c_int, C.c_char_p]
    lib.cpmppi_comm_gather.argtypes = [vp, u32, vp, vp, C.c_size_t, vp]
    lib.cpmppi_comm_wait.argtypes = [vp, u32, vp]
    lib.cpmppi_comm_sync.argtypes = [vp]
    lib.cpmppi_comm_set_timeout.argtypes = [vp, C.c_double]
    lib.cpmppi_comm_destroy.argtypes = [vp]
    lib.cpmppi_step_gather.argtypes = [vp, C.POINTER(cpmppi_step_args), vp, vp]
    lib.cpmppi_write_recordings.argtypes = [C.POINTER(C.c_char_p), C.c_char_p, C.c_size_t, C.POINTER(cpmppi_recording), C.c_int]
    lib.cpmppi_plant_step.argtypes = [vp, C.POINTER(cpmppi_plant_args), vp]
    lib.cpmppi_abi_version.restype = u32
    lib.cpmppi_stream_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.cpmppi_stream_destroy.argtypes = [vp]
    lib.cpmppi_comm_get_info.argtypes = [vp, C.POINTER(cpmppi_comm_info)]
    lib.cpmppi_groups_create.argtypes = [C.POINTER(cpmppi_config), C.c_int, u32, u32, C.POINTER(vp)]
    lib.cpmppi_groups_destroy.argtypes = [vp]
    lib.cpmppi_groups_destroy.restype = None
    lib.cpmppi_groups_count.argtypes = [vp]
    lib.cpmppi_groups_count.restype = u32
    lib.cpmppi_groups_slice.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(u32)]
    lib.cpmppi_groups_handle.argtypes = [vp, u32]
    lib.cpmppi_groups_handle.restype = vp
    lib.cpmppi_groups_stream.argtypes = [vp, u32]
    lib.cpmppi_groups_stream.restype = vp
    lib.cpmppi_groups_fork.argtypes = [vp, vp]
    lib.cpmppi_groups_join.argtypes = [vp, vp]
    lib.cpmppi_groups_run.argtypes = [vp, C.POINTER(cpmppi_step_args), C.POINTER(cpmppi_plant_args), u32]
    lib.cpmppi_groups_comm_init.argtypes = [vp, vp, C.c_int, C.c_int, C.c_char_p]
    lib.cpmppi_groups_run_gather.argtypes = [vp, C.POINTER(cpmppi_step_args), C.POINTER(cpmppi_plant_args), u32, vp]
    lib.cpmppi_comm_set_stamped.argtypes = [vp, C.c_int]
    lib.cpmppi_groups_last_error.argtypes = [vp]
    lib.cpmppi_groups_last_error.restype = C.c_char_p
    lib.cpmppi_last_launch.argtypes = [vp, C.POINTER(cpmppi_launch_info)]
    lib.cpmppi_version.restype = C.c_char_p
    for name in EXPORTS:
        getattr(lib, name)          # AttributeError here = the .so does not export what include/cpmppi.h declares
    if lib.cpmppi_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} was built as ABI {lib.cpmppi_abi_version()}, this binding is ABI {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib
