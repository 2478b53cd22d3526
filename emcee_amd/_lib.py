"""ctypes binding of libemx.so (include/emx.h).  No CPU fallback: a missing library raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libemx.so")

TARGET_HOST, TARGET_ISO, TARGET_DIAG, TARGET_DENSE, TARGET_ROSENBROCK, TARGET_BOX, TARGET_CALLBACK = range(7)
MOVE_STRETCH, MOVE_DE, MOVE_SNOOKER, MOVE_GAUSS = range(4)
GAUSS_VECTOR, GAUSS_RANDOM, GAUSS_SEQUENTIAL = range(3)
RNG_INPUTS, RNG_MT19937, RNG_PHILOX = range(3)
EXCHANGE_ALLGATHER, EXCHANGE_PULL, EXCHANGE_DIRECT, EXCHANGE_LOGPROB, EXCHANGE_REPLAY = range(5)


class MoveDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("nsplits", C.c_int32), ("randomize_split", C.c_int32), ("reserved", C.c_int32),
                ("a", C.c_double), ("sigma", C.c_double), ("g0", C.c_double), ("gammas", C.c_double)]


class EmxError(RuntimeError):
    pass


_lib = None

# emx_device_log_prob_fn (include/emx.h): (user, coords_dev, n, ndim, log_prob_dev, hip_stream) -> int
DEVICE_LOG_PROB_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p)

_P = C.c_void_p
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")

# name -> (restype, argtypes); every symbol include/emx.h declares
SIGNATURES = {
    "emx_version": (C.c_char_p, []),
    "emx_last_error": (C.c_char_p, [_P]),
    "emx_device_count": (C.c_int, [C.POINTER(C.c_int32)]),
    "emx_create": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.POINTER(_P)]),
    "emx_destroy": (C.c_int, [_P]),
    "emx_set_stream": (C.c_int, [_P, _P]),
    "emx_sync": (C.c_int, [_P]),
    "emx_status": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "emx_set_tuning": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "emx_set_state": (C.c_int, [_P, _P, _P]),
    "emx_get_state": (C.c_int, [_P, _P, _P]),
    "emx_get_accepted": (C.c_int, [_P, _u8p]),
    "emx_snapshot_save": (C.c_int, [_P, C.c_int32]),
    "emx_snapshot_read": (C.c_int, [_P, C.c_int32, _P, _P]),
    "emx_snapshot_restore": (C.c_int, [_P, C.c_int32]),
    "emx_snapshot_free": (C.c_int, [_P, C.c_int32]),
    "emx_set_target": (C.c_int, [_P, C.c_int32, _P, _P, C.c_double]),
    "emx_set_target_callback": (C.c_int, [_P, DEVICE_LOG_PROB_FN, _P]),
    "emx_eval_state_log_prob": (C.c_int, [_P]),
    "emx_eval_log_prob": (C.c_int, [_P, _dp, C.c_int64, _dp]),
    "emx_set_moves": (C.c_int, [_P, C.c_int32, C.POINTER(MoveDesc), _dp]),
    "emx_set_rng_mode": (C.c_int, [_P, C.c_int32]),
    "emx_rng_set_mt19937": (C.c_int, [_P, _u32p, C.c_int32, C.c_int32, C.c_double]),
    "emx_rng_get_mt19937": (C.c_int, [_P, _u32p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "emx_rng_set_philox": (C.c_int, [_P, C.c_uint64, C.c_uint64]),
    "emx_rng_get_philox": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "emx_chain_config": (C.c_int, [_P, C.c_int64]),
    "emx_chain_reset": (C.c_int, [_P]),
    "emx_run": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32]),
    "emx_iteration": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "emx_graph_state": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "emx_chain_read": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int64, C.c_int64, _dp]),
    "emx_accepted_counts": (C.c_int, [_P, _dp]),
    "emx_step_begin": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "emx_step_begin_with": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "emx_accept_proposals": (C.c_int, [_P, C.c_int32, _dp, _dp, _dp]),
    "emx_halfstep": (C.c_int, [_P, C.c_int32]),
    "emx_propose": (C.c_int, [_P, C.c_int32, _P, _P, C.POINTER(C.c_int64)]),
    "emx_accept": (C.c_int, [_P, C.c_int32, _dp]),
    "emx_step_end": (C.c_int, [_P]),
    "emx_plan_set": (C.c_int, [_P, C.c_int32, _ip, _ip, _ip, _ip, _ip, _dp, _dp]),
    "emx_plan_get": (C.c_int, [_P, _ip, _ip, _ip, _ip, _ip, _dp, _dp]),
    "emx_set_shard": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "emx_set_shard_buffers": (C.c_int, [_P, _P, _P, C.c_int64]),
    "emx_device_ptr": (C.c_int, [_P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "emx_shard_slots": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "emx_scatter_gathered": (C.c_int, [_P, C.c_int32]),
    "emx_set_move_scale": (C.c_int, [_P, C.c_int32, C.c_void_p, C.c_int32]),
    "emx_get_move": (C.c_int, [_P, C.c_int32, C.POINTER(MoveDesc)]),
    "emx_plan_set_noise": (C.c_int, [_P, _dp, C.c_double]),
    "emx_set_exchange": (C.c_int, [_P, C.c_int32]),
    "emx_exchange_layout": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "emx_set_exchange_buffers": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64]),
    "emx_own_walkers": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "emx_pull_prepare": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int64)]),
    "emx_pull_apply": (C.c_int, [_P, C.c_int32]),
    "emx_logprob_begin": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int64)]),
    "emx_logprob_finish": (C.c_int, [_P, C.c_int32]),
    "emx_replay_begin": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int64)]),
    "emx_replay_finish": (C.c_int, [_P, C.c_int32]),
    "emx_replay_exchange": (C.c_int, [_P, C.c_int32]),
    "emx_replica_pack": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "emx_replica_unpack": (C.c_int, [_P]),
    "emx_direct_export": (C.c_int, [_P, _u8p]),
    "emx_direct_import": (C.c_int, [_P, _u8p]),
    "emx_direct_attach": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "emx_direct_halfstep": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "emx_fft_load": (C.c_int, [C.c_char_p]),
    "emx_autocorr": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_double, _dp, _ip, C.POINTER(C.c_int64)]),
    "emx_walkers_independent": (C.c_int, [C.c_int32, _dp, C.c_int64, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "emx_walkers_independent_resident": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "emx_host_pull_capacity": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "emx_comm_load": (C.c_int, [C.c_char_p]),
    "emx_comm_get_unique_id": (C.c_int, [_u8p]),
    "emx_comm_init": (C.c_int, [_P, C.c_int32, C.c_int32, _u8p]),
    "emx_comm_destroy": (C.c_int, [_P]),
    "emx_comm_count": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "emx_pipeline_stats": (C.c_int, [_P, _dp, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "emx_pipeline_handovers": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "emx_persist_info": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "emx_mtdev_info": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "emx_mtdev_debug": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_void_p, C.c_int64]),
    "emx_mtdev_tok_stats": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "emx_persist_local_launches": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "emx_host_mt_jump": (C.c_int, [_u32p, C.c_uint64, C.c_int32, _u32p]),
    "emx_host_persist_shape": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "emx_timer_start": (C.c_int, [_P]),
    "emx_timer_stop": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "emx_profile_enable": (C.c_int, [_P, C.c_int32]),
    "emx_profile_read": (C.c_int, [_P, _f32p, C.POINTER(C.c_int32)]),
    "emx_mt_create": (_P, [_u32p, C.c_int32, C.c_int32, C.c_double]),
    "emx_mt_destroy": (None, [_P]),
    "emx_mt_get_state": (None, [_P, _u32p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "emx_mt_random_sample": (None, [_P, C.c_int64, _dp]),
    "emx_mt_randint": (None, [_P, C.c_uint64, C.c_int64, _i64p]),
    "emx_mt_randn": (None, [_P, C.c_int64, _dp]),
    "emx_mt_shuffle_labels": (None, [_P, C.c_int64, C.c_int32, _ip]),
    "emx_mt_choice_cdf": (C.c_int32, [_P, _dp, C.c_int32]),
    "emx_host_plan_mt": (C.c_int, [_P, C.c_int64, C.c_int32, C.POINTER(MoveDesc), _ip, _ip, _ip, _ip, _ip, _dp, _dp]),
    "emx_host_plan_mt_stream": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.POINTER(MoveDesc), _dp, C.c_int64, C.c_int32, C.c_int32,
                                          _P, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_double)]),
    "emx_host_split_draws": (C.c_int, [_P, C.c_int64, C.POINTER(MoveDesc), _ip, _ip, C.c_int32, _ip, _ip, _ip, _dp]),
    "emx_host_plan_philox": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int64, C.POINTER(MoveDesc), _ip, _ip, _ip, _ip, _ip, _dp, _dp]),
    "emx_host_move_choice_philox": (C.c_int32, [C.c_uint64, C.c_uint64, _dp, C.c_int32]),
}


def load():
    """Load libemx.so and type every entry point.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime (libamdhip64) with the same SONAME as /opt/rocm's.
    # Whichever loads first serves the whole process, and torch cannot initialise on the system
    # runtime ("No HIP GPUs are available").  Import torch first when it is installed so that
    # libemx, torch and RCCL share ONE runtime; libemx itself needs nothing from torch.
    if not os.environ.get("EMX_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001
            pass
    path = os.environ.get("EMX_LIB") or LIB_PATH          # EMX_LIB: an alternative build (A/B measurements)
    if not os.path.exists(path):
        raise EmxError("libemx.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`. "
                       "There is no CPU fallback." % path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(lib, ctx, rc):
    if rc != 0:
        msg = lib.emx_last_error(ctx)
        raise EmxError((msg or b"unknown error").decode() + " (code %d)" % rc)


def device_count():
    lib = load()
    n = C.c_int32(0)
    lib.emx_device_count(C.byref(n))
    return n.value
