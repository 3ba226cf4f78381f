"""ctypes view of oracle/libtoricenv_host_twin.so -- TEST INFRASTRUCTURE ONLY.

The host twin (oracle/host_twin.cpp) exports the hot path of include/toricenv.h for ``device = -1`` on host memory.
Same import restrictions as the rest of oracle/: tests/, smoke() and bench.py's cpu_baseline leg only; the product
(toric-rl-decoder_amd/) never imports this module and has no CPU path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libtoricenv_host_twin.so")
_lib = None
_vp, _i, _i64, _u64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double

TQ_F32, TQ_U8 = 0, 3
STRATEGY = {"fixed": 0, "linear": 1, "random": 2}


def build(force=False):
    srcs = [os.path.join(_HERE, "host_twin.cpp"), os.path.join(_HERE, "..", "toric-rl-decoder_amd", "csrc", "lattice.hpp"),
            os.path.join(_HERE, "..", "include", "toricenv.h")]
    if force or not os.path.exists(_PATH) or os.path.getmtime(_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtoricenv_host_twin.so"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    return _PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_PATH)
        L.tq_last_error.restype = C.c_char_p
        L.tq_create.argtypes = [C.POINTER(_vp), _i, _i, _i, _u64, _i64]
        L.tq_destroy.argtypes = [_vp]
        L.tq_set_params.argtypes = [_vp, _d, _d, _i]
        L.tq_set_min_qubit_errors.argtypes = [_vp, _i]
        L.tq_set_perror_schedule.argtypes = [_vp, _i, _d, _d, _d]
        L.tq_reset_all.argtypes = [_vp, _vp, _vp]
        L.tq_get_state.argtypes = [_vp, _vp, _vp]
        L.tq_get_qubits.argtypes = [_vp, _vp, _vp]
        L.tq_get_counters.argtypes = [_vp, _vp, _vp, _vp]
        L.tq_persp_count.argtypes = [_vp, _vp, _vp, _vp]
        L.tq_persp_write.argtypes = [_vp, _vp, _vp, _vp, _i64, _i, _vp]
        L.tq_transition_block_bytes.argtypes = [_i, _i64]
        L.tq_transition_block_bytes.restype = _i64
        L.tq_actor_step.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]
        L.tq_check.argtypes = [_vp, _vp]
        _lib = L
    return _lib


class TwinError(RuntimeError):
    pass


def _ptr(a):
    return None if a is None else a.ctypes.data


class HostEnvSet:
    """The hot path of the C-ABI on host memory (numpy arrays), call for call like the product's EnvSet uses it."""

    def __init__(self, size, no_envs, p_error=0.1, seed=0, first_env_id=0, terminal_reward=100.0, max_steps_per_episode=75,
                 min_qubit_errors=0):
        self.L = lib()
        self.size, self.no_envs = int(size), int(no_envs)
        self.h = _vp(None)
        self._ck(self.L.tq_create(C.byref(self.h), self.no_envs, self.size, -1, int(seed), int(first_env_id)))
        if min_qubit_errors:
            self._ck(self.L.tq_set_min_qubit_errors(self.h, int(min_qubit_errors)))
        self._ck(self.L.tq_set_params(self.h, float(p_error), float(terminal_reward), int(max_steps_per_episode)))

    def _ck(self, rc):
        if rc != 0:
            raise TwinError("host twin error %d: %s" % (rc, self.L.tq_last_error().decode()))

    def close(self):
        if self.h:
            self.L.tq_destroy(self.h)
            self.h = _vp(None)

    __del__ = close

    def set_perror_schedule(self, strategy, p_start, p_final, p_delta):
        self._ck(self.L.tq_set_perror_schedule(self.h, STRATEGY[strategy], p_start, p_final, p_delta))

    def check(self):
        self._ck(self.L.tq_check(self.h, None))

    def reset_all(self, p_errors=None):
        p = None if p_errors is None else np.ascontiguousarray(p_errors, np.float64)
        self._ck(self.L.tq_reset_all(self.h, _ptr(p), None))
        return self.states()

    def states(self):
        out = np.empty((self.no_envs, 2, self.size, self.size), np.uint8)
        self._ck(self.L.tq_get_state(self.h, out.ctypes.data, None))
        return out

    def qubits(self):
        out = np.empty((self.no_envs, 2, self.size, self.size), np.uint8)
        self._ck(self.L.tq_get_qubits(self.h, out.ctypes.data, None))
        return out

    def counters(self):
        ep, st = np.empty(self.no_envs, np.uint32), np.empty(self.no_envs, np.uint32)
        self._ck(self.L.tq_get_counters(self.h, ep.ctypes.data, st.ctypes.data, None))
        return ep, st

    def counts(self):
        cnt, off = np.empty(self.no_envs, np.int32), np.empty(self.no_envs + 1, np.int64)
        self._ck(self.L.tq_persp_count(self.h, cnt.ctypes.data, off.ctypes.data, None))
        return cnt, off

    def perspectives(self, dtype=np.float32, capacity=None, out=None, positions=None):
        cnt, off = self.counts()
        P = int(off[-1])
        cap = P if capacity is None else int(capacity)
        if out is None:
            out = np.empty((cap, 2, self.size, self.size), dtype)
        if positions is None:
            positions = np.empty((cap, 3), np.int32)
        self._ck(self.L.tq_persp_write(self.h, off.ctypes.data, out.ctypes.data, positions.ctypes.data, cap,
                                       TQ_F32 if out.dtype == np.float32 else TQ_U8, None))
        return out, positions, cnt, off

    def new_block(self, steps=1):
        cap = self.no_envs * int(steps)
        return np.zeros(self.L.tq_transition_block_bytes(self.size, cap), np.uint8), cap

    def actor_step(self, actions=None, block=None, block_cap=0, slot=0, want_actions=True):
        a = None if actions is None else np.ascontiguousarray(actions, np.int32)
        act = np.empty((self.no_envs, 4), np.int32) if want_actions else None
        rew, term = np.empty(self.no_envs, np.float32), np.empty(self.no_envs, np.uint8)
        self._ck(self.L.tq_actor_step(self.h, _ptr(a), _ptr(act), rew.ctypes.data, term.ctypes.data, _ptr(block), int(block_cap),
                                      int(slot) * self.no_envs, None))
        return act, rew, term
