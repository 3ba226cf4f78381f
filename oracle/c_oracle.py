"""ctypes view of oracle/libtoric_oracle.so -- TEST INFRASTRUCTURE ONLY.

Same import restrictions as oracle/toric_oracle.py: tests/, smoke() and bench.py's
cpu_baseline leg only.  ``build()`` compiles the library with ``make -C oracle``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libtoric_oracle.so")
_lib = None

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    if force or not os.path.exists(_PATH) or \
            os.path.getmtime(_PATH) < os.path.getmtime(os.path.join(_HERE, "toric_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtoric_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        L = C.CDLL(_PATH)
        L.tor_philox4x32.argtypes = [_u32p, _u32p, _u32p]
        L.tor_syndrome.argtypes = [_u8p, _u8p, C.c_int]
        L.tor_syndrome.restype = C.c_int
        L.tor_eval_ground_state.argtypes = [_u8p, C.c_int]
        L.tor_eval_ground_state.restype = C.c_int
        L.tor_reset_batch.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_double, _u8p, _u8p, _u32p, _u32p]
        L.tor_step_batch.argtypes = [C.c_int, C.c_int, _i32p, C.c_double, _u8p, _u8p, _f32p, _u8p, _u32p]
        L.tor_persp_count.argtypes = [C.c_int, C.c_int, _u8p, _i32p, _i64p]
        L.tor_persp_write.argtypes = [C.c_int, C.c_int, _u8p, _i64p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.tor_rotate_state.argtypes = [_u8p, _u8p, C.c_int]
        L.tor_transition.argtypes = [C.c_int, C.c_int, _i32p, _u8p, _u8p, _u8p, _i32p, _u8p]
        L.tor_select_action.argtypes = [C.c_int, _f32p, _i64p, _i32p, _f64p, C.c_uint64, C.c_int64,
                                        _u32p, _u32p, _i32p, _f32p]
        L.tor_perror_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, C.c_double]
        L.tor_perror_draw.restype = C.c_double
        L.tor_actor_steps.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double,
                                      C.c_double, C.c_double, C.c_int, _u8p, _u8p, _u32p, _u32p,
                                      C.POINTER(C.c_double)]
        L.tor_actor_steps.restype = C.c_int64
        L.tor_num_threads.restype = C.c_int
        L.tor_set_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


class CEnvBatch:
    """A batch of lattices driven by the C oracle (arrays owned here, numpy)."""

    def __init__(self, size, no_envs, p_error=0.1, seed=0, first_env_id=0, terminal_reward=100.0):
        self.L = lib()
        self.size, self.no_envs = int(size), int(no_envs)
        self.p_error, self.seed, self.first = float(p_error), int(seed), int(first_env_id)
        self.terminal_reward = float(terminal_reward)
        d = self.size
        self.qubits = np.zeros((no_envs, 2, d, d), np.uint8)
        self.states = np.zeros((no_envs, 2, d, d), np.uint8)
        self.episodes = np.zeros(no_envs, np.uint32)
        self.steps = np.zeros(no_envs, np.uint32)

    def reset(self, idx=None, p_errors=None):
        ip = pp = None
        n_idx = 0
        if idx is not None:
            idx = np.ascontiguousarray(idx, np.int32)
            ip, n_idx = idx.ctypes.data, idx.shape[0]
        if p_errors is not None:
            p_errors = np.ascontiguousarray(p_errors, np.float64)
            pp = p_errors.ctypes.data
        self.L.tor_reset_batch(self.seed, self.first, self.no_envs, self.size, ip, n_idx, pp,
                               self.p_error, self.qubits, self.states, self.episodes, self.steps)
        return self.states

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.int32).reshape(self.no_envs, 4)
        rew = np.empty(self.no_envs, np.float32)
        term = np.empty(self.no_envs, np.uint8)
        self.L.tor_step_batch(self.no_envs, self.size, a, self.terminal_reward, self.qubits,
                              self.states, rew, term, self.steps)
        return self.states, rew, term.astype(bool)

    def perspectives(self, states=None, dtype=np.uint8):
        s = self.states if states is None else np.ascontiguousarray(states, np.uint8)
        n, d = s.shape[0], self.size
        counts = np.empty(n, np.int32)
        offsets = np.empty(n + 1, np.int64)
        self.L.tor_persp_count(n, d, s, counts, offsets)
        P = int(offsets[-1])
        out = np.empty((P, 2, d, d), dtype)
        pos = np.empty((P, 3), np.int32)
        u8 = out.ctypes.data if dtype == np.uint8 else None
        f32 = out.ctypes.data if dtype == np.float32 else None
        self.L.tor_persp_write(n, d, s, offsets, u8, f32, pos.ctypes.data)
        return out, pos, counts, offsets

    def positions(self, states=None):
        """(positions (P,3), counts, offsets) without materialising the stack (full-size shards)."""
        s = self.states if states is None else np.ascontiguousarray(states, np.uint8)
        n, d = s.shape[0], self.size
        counts = np.empty(n, np.int32)
        offsets = np.empty(n + 1, np.int64)
        self.L.tor_persp_count(n, d, s, counts, offsets)
        pos = np.empty((int(offsets[-1]), 3), np.int32)
        self.L.tor_persp_write(n, d, s, offsets, None, None, pos.ctypes.data)
        return pos, counts, offsets

    def transition(self, actions, states, next_states):
        n, d = self.no_envs, self.size
        a = np.ascontiguousarray(actions, np.int32).reshape(n, 4)
        per = np.empty((n, 2, d, d), np.uint8)
        nper = np.empty((n, 2, d, d), np.uint8)
        act = np.empty((n, 4), np.int32)
        self.L.tor_transition(n, d, a, np.ascontiguousarray(states, np.uint8),
                              np.ascontiguousarray(next_states, np.uint8), per, act, nper)
        return per, act, nper

    def select(self, q_table, offsets, positions, eps):
        n = self.no_envs
        actions = np.empty((n, 4), np.int32)
        qv = np.empty((n, 3), np.float32)
        eps = np.ascontiguousarray(np.broadcast_to(np.asarray(eps, np.float64), (n,)))
        self.L.tor_select_action(n, np.ascontiguousarray(q_table, np.float32).reshape(-1),
                                 np.ascontiguousarray(offsets, np.int64),
                                 np.ascontiguousarray(positions, np.int32).reshape(-1), eps, self.seed,
                                 self.first, self.episodes, self.steps, actions, qv)
        return actions, qv

    def actor_steps(self, n_steps, eps=1.0, p_reset=None, max_steps_per_episode=75):
        cs = C.c_double(0.0)
        P = self.L.tor_actor_steps(self.seed, self.first, self.no_envs, self.size, int(n_steps),
                                   float(eps), self.p_error if p_reset is None else float(p_reset),
                                   self.terminal_reward, int(max_steps_per_episode), self.qubits,
                                   self.states, self.episodes, self.steps, C.byref(cs))
        return int(P), cs.value
