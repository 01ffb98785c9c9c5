"""ctypes binding of libmjmpc_amd.so (the C ABI in include/mjmpc_amd.h).

There is NO CPU fallback: if the library is missing or no GPU is visible, the product path raises.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MJMPC_AMD_LIB", os.path.join(_HERE, "libmjmpc_amd.so"))

F32, F64 = 0, 1
ABI_VERSION = 4      # include/mjmpc_amd.h MJMPC_ABI_VERSION this binding was written for

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_dbl = ctypes.c_double
_dp = ctypes.POINTER(ctypes.c_double)

# name -> (restype, argtypes); doubles as the list of symbols include/mjmpc_amd.h declares
SIGNATURES = {
    "mjmpc_abi_version": (_int, []),
    "mjmpc_last_error": (ctypes.c_char_p, []),
    "mjmpc_device_count": (_int, []),
    "mjmpc_arm_create": (_int, [_dp, _int, _int, ctypes.POINTER(_vp)]),
    "mjmpc_arm_set_shard_models": (_int, [_vp, _dp, _int]),
    "mjmpc_arm_set_shard_states": (_int, [_vp, _dp, _int, _vp]),
    "mjmpc_arm_destroy": (_int, [_vp]),
    "mjmpc_arm_dims": (_int, [_vp, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "mjmpc_arm_set_state": (_int, [_vp, _dp, _dp, _dp, _vp]),
    "mjmpc_arm_state_ptr": (_vp, [_vp]),
    "mjmpc_arm_rollout": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mjmpc_arm_rollout_cl": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mjmpc_arm_rollout_fused": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mjmpc_tree_rollout_fused": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mjmpc_arm_step_state": (_int, [_vp, _int, _vp, _vp, _vp, _vp]),
    "mjmpc_arm_mppi_combine": (_int, [_vp, _int, _vp, _int, _int, _vp, _vp, _vp, ctypes.c_double, _int, _vp, _vp, _int, _vp, _vp, _vp]),
    "mjmpc_arm_rollout_sampled": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _int, ctypes.c_uint64, ctypes.c_uint64, _i64, _vp,
                                         _vp, _vp, _vp, _vp]),
    "mjmpc_arm_mppi_step": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_uint64, _i64, _vp,
                                   _dbl, _dbl, _int, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mjmpc_tree_create": (_int, [_dp, _int, _int, ctypes.POINTER(_vp)]),
    "mjmpc_tree_destroy": (_int, [_vp]),
    "mjmpc_tree_dims": (_int, [_vp, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "mjmpc_tree_nq": (_int, [_vp]),
    "mjmpc_tree_set_shard_models": (_int, [_vp, _dp, _int]),
    "mjmpc_tree_set_state": (_int, [_vp, _dp, _dp, _dp, _vp]),
    "mjmpc_tree_set_shard_states": (_int, [_vp, _dp, _int, _vp]),
    "mjmpc_tree_rollout": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mjmpc_tree_rollout_cl": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mjmpc_tree_step_state": (_int, [_vp, _int, _vp, _vp, _vp, _vp]),
    "mjmpc_tree_get_state": (_int, [_vp, _dp, _dp, _vp]),
    "mjmpc_tree_solver_failures": (_int, [_vp, ctypes.POINTER(ctypes.c_uint32)]),
    "mjmpc_tree_diverged": (_int, [_vp, ctypes.POINTER(ctypes.c_uint32)]),
    "mjmpc_analytic_rollout": (_int, [_int, _vp, _int, _int, _vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _int,
                                      _vp]),
    "mjmpc_arm_solver_failures": (_int, [_vp, ctypes.POINTER(ctypes.c_uint32)]),
    "mjmpc_arm_diverged": (_int, [_vp, ctypes.POINTER(ctypes.c_uint32)]),
    "mjmpc_tree_env_resets": (_int, [_vp, ctypes.POINTER(ctypes.c_uint32)]),
    "mjmpc_arm_env_resets": (_int, [_vp, ctypes.POINTER(ctypes.c_uint32)]),
    "mjmpc_tree_set_reset_returns": (_int, [_vp, _int]),
    "mjmpc_arm_set_reset_returns": (_int, [_vp, _int]),
    "mjmpc_update_workspace_bytes": (_i64, [_i64, _int, _int]),
    "mjmpc_softmax_record_len": (_int, [_int, _int, _int]),
    "mjmpc_traj_cost": (_int, [_int, _i64, _int, _int, _vp, _vp, _int, _vp, _vp]),
    "mjmpc_workspace_q0": (_vp, [_vp, _i64, _int, _int]),
    "mjmpc_td_lambda_returns": (_int, [_int, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _int, _dbl, _int, _dbl, _dbl,
                                       _vp, _vp, _vp]),
    "mjmpc_softmax_stats": (_int, [_int, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _int, _dbl, _int, _int, _int,
                                    _vp, _vp, _vp]),
    "mjmpc_softmax_combine": (_int, [_vp, _int, _int, _int, _int, _dbl, _dbl, _int, _dbl, _vp, _vp, _vp, _vp, _vp]),
    "mjmpc_softmax_weights": (_int, [_i64, _int, _int, _vp, _vp, _vp, _vp]),
    "mjmpc_cem_elite_sums": (_int, [_int, _i64, _int, _int, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "mjmpc_cem_elite_cov": (_int, [_int, _i64, _int, _int, _vp, _vp, _vp, _int, _vp, _vp, _vp]),
    "mjmpc_cem_final": (_int, [_vp, _int, _i64, _int, _int, _dbl, _int, _dbl, _vp, _vp, _vp, _vp]),
    "mjmpc_cem_combine": (_int, [_vp, _int, _int, _int, _dbl, _int, _dbl, _vp, _vp, _vp]),
    "mjmpc_cem_fused_supported": (_int, [_i64, _i64, _i64, _int, _int]),
    "mjmpc_cem_select_moments": (_int, [_int, _i64, _int, _int, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "mjmpc_cem_record": (_int, [_i64, _int, _int, _i64, _vp, _vp, _vp, _vp]),
    "mjmpc_cem_finish": (_int, [_int, _i64, _int, _int, _i64, _vp, _int, _dbl, _int, _dbl, _int, _vp, _vp, _vp, _vp, _vp, _dbl,
                                _vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_uint64, _i64, _vp, _vp]),
    "mjmpc_rs_best": (_int, [_int, _i64, _int, _int, _vp, _i64, _vp, _vp, _vp]),
    "mjmpc_rs_combine": (_int, [_vp, _int, _int, _int, _dbl, _vp, _vp]),
    "mjmpc_mppi_fused_update": (_int, [_int, _i64, _int, _int, _vp, _vp, _dbl, _dbl, _int, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _vp, _vp]),
    "mjmpc_mppi_fused_combine": (_int, [_vp, _int, _dbl, _int, _int, _dbl, _dbl, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mjmpc_mppi_fused_update_draw_next": (_int, [_int, _i64, _int, _int, _vp, _vp, _dbl, _dbl, _int, _vp, _vp, _vp, _vp,
                                                   _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_uint64, _i64, _vp,
                                                   _int, _vp]),
    "mjmpc_q0_sum": (_int, [_i64, _int, _int, _vp, _vp, _vp]),
    "mjmpc_shift_mean": (_int, [_vp, _int, _int, _int, _vp, _vp]),
    "mjmpc_step_tail": (_int, [_vp, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp]),
    "mjmpc_cholesky_lower": (_int, [_vp, _int, _vp, _vp, _vp]),
    "mjmpc_cov_add_diag": (_int, [_vp, _int, _vp, _dbl, _vp]),
    "mjmpc_color_noise": (_int, [_int, _vp, _i64, _int, _vp, _vp]),
    "mjmpc_filter_noise": (_int, [_int, _vp, _i64, _int, _int, _vp, _vp]),
    "mjmpc_mt19937_workspace_bytes": (_i64, [_i64]),
    "mjmpc_sample_noise_mt19937": (_int, [_int, _vp, _i64, _dbl, ctypes.c_uint64, _vp, _vp, _vp, _vp]),
    "mjmpc_mt19937_stream_words": (_i64, [_i64]),
    "mjmpc_sample_noise_mt19937_jump": (_int, [_int, _vp, _i64, _dbl, ctypes.c_uint64, _vp, _vp, _vp, _i64, _i64, _int,
                                                 _i64, _vp, _vp, _vp]),
    "mjmpc_graph_kernel_nodes": (_int, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    "mjmpc_graph_signature": (_int, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "mjmpc_comm_unique_id": (_int, [_vp]),
    "mjmpc_comm_create": (_int, [_vp, _int, _int, _int, ctypes.POINTER(_vp)]),
    "mjmpc_comm_all_gather_f64": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "mjmpc_comm_destroy": (_int, [_vp]),
    "mjmpc_sample_noise": (_int, [_int, _vp, _i64, _int, _int, _vp, _vp, ctypes.c_uint64, ctypes.c_uint64, _i64, _vp, _int, _vp]),
}

_LIB = None


class MjmpcError(RuntimeError):
    pass


def load():
    """Load the shared library (no GPU needed just to load and inspect symbols)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise MjmpcError("%s not built - run `python -m mjmpc_amd.build` (needs hipcc); "
                             "mjmpc_amd has no CPU fallback" % LIB_PATH)
        # torch bundles its own libamdhip64.so.7; it has to be the copy this process binds, so
        # that device pointers and streams are shared - import torch BEFORE dlopen'ing our library
        # (the other order leaves torch with "No HIP GPUs are available").
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        # the version FIRST: a stale library lacks the newer entry points, and binding them would end in a bare
        # AttributeError instead of the instruction to rebuild
        lib.mjmpc_abi_version.restype, lib.mjmpc_abi_version.argtypes = SIGNATURES["mjmpc_abi_version"]
        got = lib.mjmpc_abi_version()
        if got != ABI_VERSION:
            raise MjmpcError("%s speaks ABI version %d, this binding expects %d (blob / state layouts or entry points differ): "
                             "rebuild it with `python -m mjmpc_amd.build --force`" % (LIB_PATH, got, ABI_VERSION))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _LIB = lib
    return _LIB


class recording:
    """``with recording(tape):`` every call into the library made inside the block is also appended to ``tape`` as
    (function, args) - the launch sequence of one control iteration, which ``Controller`` replays call by call where
    that is cheaper than a hipGraph replay (control/controller.py ``_LaunchTape``)."""

    _lock = threading.Lock()        # the wrappers sit on the process-wide CDLL: one recording at a time

    def __init__(self, tape):
        self.tape, self.saved = tape, {}
        self.owner = None

    def __enter__(self):
        lib = load()
        if not recording._lock.acquire(timeout=60.0):
            raise MjmpcError("another thread has been recording a launch tape for a minute")
        self.owner = threading.get_ident()
        try:
            for name in SIGNATURES:
                fn = getattr(lib, name)
                self.saved[name] = fn

                def wrapper(*args, _fn=fn, _tape=self.tape, _me=self.owner):
                    if threading.get_ident() == _me:    # (another thread's calls - a second controller - are not this tape's)
                        _tape.append((_fn, args))
                    return _fn(*args)
                setattr(lib, name, wrapper)
        except BaseException:           # (whatever was patched goes back, and the lock with it)
            for name, fn in self.saved.items():
                setattr(lib, name, fn)
            recording._lock.release()
            raise
        return self

    def __exit__(self, *exc):
        lib = load()
        for name, fn in self.saved.items():
            setattr(lib, name, fn)
        recording._lock.release()
        return False


def check(rc):
    if rc != 0:
        raise MjmpcError("mjmpc_amd call failed (%d): %s" % (rc, load().mjmpc_last_error().decode()))


def require_gpu():
    lib = load()
    if lib.mjmpc_device_count() < 1:
        raise MjmpcError("no HIP device visible: the mjmpc_amd product path runs only on the GPU "
                         "(the CPU oracle under oracle/ is test infrastructure, not a fallback)")
    return lib
