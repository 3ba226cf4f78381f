"""ctypes binding of libtoricenv.so (include/toricenv.h).

There is no CPU fallback: if the library is missing or no HIP device is usable the
functions raise -- a GPU box must never pass silently on some other path.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtoricenv.so")
CSRC = os.path.join(_HERE, "csrc")

TQ_F32, TQ_F16, TQ_BF16, TQ_U8 = 0, 1, 2, 3
TQ_PERR_FIXED, TQ_PERR_LINEAR, TQ_PERR_RANDOM = 0, 1, 2
TQ_E_INVALID, TQ_E_HIP, TQ_E_CAPACITY, TQ_E_ACTION, TQ_E_INDEX, TQ_E_RESET = -1, -2, -3, -4, -5, -6

# every symbol include/toricenv.h declares: (name, restype, argtypes)
_vp, _i, _i64, _u64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
SYMBOLS = [
    ("tq_version", _i, []),
    ("tq_last_error", C.c_char_p, []),
    ("tq_create", _i, [C.POINTER(_vp), _i, _i, _i, _u64, _i64]),
    ("tq_destroy", _i, [_vp]),
    ("tq_set_params", _i, [_vp, _d, _d, _i]),
    ("tq_set_min_qubit_errors", _i, [_vp, _i]),
    ("tq_set_perror_schedule", _i, [_vp, _i, _d, _d, _d]),
    ("tq_stack_alloc", _i, [_i, _u64, C.POINTER(_vp)]),
    ("tq_stack_free", _i, [_vp]),
    ("tq_set_xcd_bias", _i, [_i]),
    ("tq_get_xcd_bias", _i, []),
    ("tq_env_set_xcd_bias", _i, [_vp, _i]),
    ("tq_env_get_xcd_bias", _i, [_vp]),
    ("tq_num_envs", _i, [_vp]),
    ("tq_size", _i, [_vp]),
    ("tq_reset_all", _i, [_vp, _vp, _vp]),
    ("tq_reset_idx", _i, [_vp, _vp, _i, _vp, _vp]),
    ("tq_step", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("tq_get_state", _i, [_vp, _vp, _vp]),
    ("tq_get_state_idx", _i, [_vp, _vp, _i, _vp, _vp]),
    ("tq_get_qubits", _i, [_vp, _vp, _vp]),
    ("tq_set_qubits", _i, [_vp, _vp, _vp]),
    ("tq_get_counters", _i, [_vp, _vp, _vp, _vp]),
    ("tq_eval_ground_state", _i, [_vp, _vp, _vp]),
    ("tq_is_terminal", _i, [_vp, _vp, _vp]),
    ("tq_persp_count", _i, [_vp, _vp, _vp, _vp]),
    ("tq_persp_write", _i, [_vp, _vp, _vp, _vp, _i64, _i, _vp]),
    ("tq_persp_write_range", _i, [_vp, _vp, _i, _i, _vp, _vp, _i64, _i, _vp]),
    ("tq_states_reserve", _i, [_i, _i]),
    ("tq_states_persp_count", _i, [_i, _i, _vp, _vp, _vp, _vp]),
    ("tq_states_persp_write", _i, [_i, _i, _vp, _vp, _vp, _vp, _i64, _i, _vp]),
    ("tq_select_action", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("tq_states_select_action", _i, [_i, _vp, _vp, _vp, _vp, _u64, _u64, _i64, _vp, _vp, _vp]),
    ("tq_states_check", _i, [_vp]),
    ("tq_segment_max", _i, [_vp, _vp, _i, _vp, _vp, _vp]),
    ("tq_transition_write", _i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    ("tq_states_transition", _i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("tq_transition_block_bytes", _i64, [_i, _i64]),
    ("tq_transition_unpack", _i, [_i, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("tq_block_priorities", _i, [_i, _vp, _i64, _i, _i, _vp, _d, _vp]),
    ("tq_actor_step", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    ("tq_check", _i, [_vp, _vp]),
]

_lib = None


class ToricEnvError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile libtoricenv.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in ("toricenv.hip", "kernels.hpp", "stream_write.hpp", "lattice.hpp", "Makefile")]
    srcs.append(os.path.join(_HERE, "..", "include", "toricenv.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        cmd = ["make", "-C", CSRC] + (["-B"] if force else [])
        out = None if verbose else subprocess.DEVNULL
        subprocess.check_call(cmd, stdout=out)
    return LIB_PATH


def load():
    """dlopen the library and bind every symbol of the header.  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ToricEnvError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)           # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        msg = load().tq_last_error().decode("utf-8", "replace")
        if rc in (TQ_E_INVALID, TQ_E_ACTION, TQ_E_INDEX):
            raise ValueError(f"libtoricenv: {msg}")
        raise ToricEnvError(f"libtoricenv error {rc}: {msg}")
