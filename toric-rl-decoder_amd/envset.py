"""Host-side mirror of the reference's env surface over libtoricenv (HIP).

Replaces, with the same names / argument order / return shapes:
  * ``gym.make('toric-code-v0', config=...)``  -> :func:`make` / :class:`ToricEnv`
    (gym_ToricCode is an absent submodule upstream; API census in SURVEY.md 8(b))
  * ``src/EnvSet.py:4-51``                      -> :class:`EnvSet`
  * ``generatePerspectiveBatch`` + concatenate (``src/numba/util_actor.py:33-39,56-67``)
                                                -> :meth:`EnvSet.generatePerspective`
  * ``_selectActionBatch_prime`` (``src/numba/util_actor.py:69-107``) -> :meth:`EnvSet.selectAction`
  * ``generateTransitionParallel`` (``src/util_actor.py:223-264``)    -> :meth:`EnvSet.generateTransition`
  * the body of the actor loop after the policy (``src/Actor_mp.py:116-183``) -> :meth:`EnvSet.actorStep`

PyTorch-ROCm is only the device-memory container and stream provider; all lattice work
happens in the HIP kernels behind the C-ABI.  With ``numpy_io=True`` (default) the methods
take/return numpy arrays with the reference's dtypes, so ``Actor_mp``-style loops run
unchanged; with ``numpy_io=False`` they take/return device tensors and never synchronise.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, wire
from ._lib import TQ_BF16, TQ_F16, TQ_F32, TQ_U8, check

_DTYPES = {torch.float32: TQ_F32, torch.float16: TQ_F16, torch.bfloat16: TQ_BF16, torch.uint8: TQ_U8}
_STRATEGY = {None: 0, "fixed": 0, "linear": 1, "random": 2}
SUPPORTED_SIZES = (3, 5, 7, 9, 11, 13, 15, 17, 19, 21)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise _lib.ToricEnvError("no HIP device visible to PyTorch-ROCm: the toric env has no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise ValueError(f"device must be a cuda (ROCm) device, got {dev}")
    return torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())


class _ActionSpace:
    """``env.action_space.high[-1] == 3`` (Actor_mp.py:58)."""

    def __init__(self, d):
        self.low = np.array([0, 0, 0, 1])
        self.high = np.array([1, d - 1, d - 1, 3])


class ToricEnv:
    """Single-lattice facade with the attributes the reference touches on a gym env:
    system_size, action_space, reset, step, qubit_matrix, state, createSyndromOpt,
    isTerminalState, evalGroundState.  Backed by an EnvSet of one lattice on the GPU,
    created on first use."""

    def __init__(self, config=None, device=None, seed=0):
        config = dict(config or {})
        self.config = config
        self.system_size = int(config.get("size", 3))
        if self.system_size not in SUPPORTED_SIZES:
            raise ValueError(f"size must be odd in {SUPPORTED_SIZES}, got {self.system_size}")
        self.min_qubit_errors = int(config.get("min_qubit_errors", 0))
        if not 0 <= self.min_qubit_errors <= 2 * self.system_size ** 2:
            raise ValueError("min_qubit_errors must be in [0, 2*size*size]")
        self.p_error = float(config.get("p_error", 0.1))
        self.terminal_reward = float(config.get("terminal_reward", 100.0))
        self.action_space = _ActionSpace(self.system_size)
        self.device = device
        self.seed = int(seed)
        self._set = None
        self._scratch = None

    def _envs(self):
        if self._set is None:
            self._set = EnvSet(self, 1, device=self.device, seed=self.seed)
        return self._set

    def reset(self, p_error=None):
        return self._envs().resetAll(None if p_error is None else [p_error])[0]

    def step(self, action):
        s, r, t, info = self._envs().step(np.asarray(action).reshape(1, 4))
        return s[0], float(r[0]), bool(t[0]), info

    @property
    def state(self):
        return self._envs().getStates()[0]

    @property
    def qubit_matrix(self):
        return self._envs().getQubits()[0]

    @qubit_matrix.setter
    def qubit_matrix(self, q):
        self._envs().setQubits(np.asarray(q).reshape(1, 2, self.system_size, self.system_size))

    def createSyndromOpt(self, qubit_matrix):
        d = self.system_size
        if self._scratch is None:                              # one scratch lattice, created once (11 hipMallocs)
            self._scratch = EnvSet(self, 1, device=self.device, seed=self.seed)
        self._scratch.setQubits(np.asarray(qubit_matrix).reshape(1, 2, d, d))
        return self._scratch.getStates()[0]

    @staticmethod
    def isTerminalState(state):
        return bool(np.all(np.asarray(state) == 0))

    def evalGroundState(self):
        return bool(self._envs().evalGroundState()[0])


def make(env_id, config=None, device=None, seed=0):
    """Stand-in for ``gym.make('toric-code-v0', config=...)`` (Distributed_mp.py:72-76)."""
    if env_id != "toric-code-v0":
        raise ValueError(f"unknown env id {env_id!r} (only 'toric-code-v0')")
    return ToricEnv(config, device=device, seed=seed)


class _ChunkedBuffer:
    """Device memory from tq_stack_alloc, exposed through __cuda_array_interface__ and freed with the last tensor
    that views it."""

    def __init__(self, nbytes, device):
        self.ptr = C.c_void_p(None)
        self.nbytes = int(nbytes)
        self._L = _lib.load()
        check(self._L.tq_stack_alloc(device.index, self.nbytes, C.byref(self.ptr)))
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr.value, False), "version": 2}

    def __del__(self):
        try:
            if self.ptr.value:
                self._L.tq_stack_free(self.ptr)
                self.ptr = C.c_void_p(None)
        except Exception:
            pass


def alloc_chunked(shape, dtype=torch.float32, device=None):
    """A device tensor of ``shape`` / ``dtype`` in tq_stack_alloc memory: 2 MiB physical chunks behind one virtual
    range, zero-filled, every page verified to be reached through its own address (include/toricenv.h).  The kind of
    allocation the stack write ran fastest on in most processes of round 3 (6.5-6.8 TB/s against 5.1-5.5 into
    torch.empty buffers; on some boxes no kind is faster than another).  The memory is released when the returned
    tensor (and every view of it) is gone."""
    dev = _require_gpu(device)
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
    nbytes = int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dtype).element_size()
    holder = _ChunkedBuffer(max(nbytes, 16), dev)
    with torch.cuda.device(dev):
        flat = torch.as_tensor(holder, device=dev)            # zero-copy view; keeps `holder` alive
    return flat[:nbytes].view(dtype).view(shape)


def alloc_stack(capacity, size, dtype=torch.float32, device=None):
    """A stack buffer (capacity, 2, d, d) of ``dtype`` from alloc_chunked (tq_stack_alloc)."""
    return alloc_chunked((int(capacity), 2, int(size), int(size)), dtype, device)


class TransitionBlock:
    """Packed transition block on the device (layout: include/toricenv.h)."""

    def __init__(self, d, capacity, device):
        self.d, self.capacity = int(d), int(capacity)
        nbytes = _lib.load().tq_transition_block_bytes(self.d, self.capacity)
        if nbytes < 0:
            raise ValueError("bad transition block shape")
        self.buf = torch.zeros(max(int(nbytes), 8), dtype=torch.uint8, device=device)

    @property
    def nbytes(self):
        return int(self.buf.numel())

    def unpack(self, first=0, count=None, buf=None):
        """-> dict of device tensors (perspective u8, next_perspective u8, action i32[n,4],
        reward f32, terminal u8, priority f32) for slots [first, first+count).  Slots without a
        transition have action op 0 (include/toricenv.h)."""
        buf = self.buf if buf is None else buf
        count = self.capacity - first if count is None else int(count)
        d, dev = self.d, buf.device
        out = dict(perspective=torch.empty((count, 2, d, d), dtype=torch.uint8, device=dev),
                   next_perspective=torch.empty((count, 2, d, d), dtype=torch.uint8, device=dev),
                   action=torch.empty((count, 4), dtype=torch.int32, device=dev),
                   reward=torch.empty(count, dtype=torch.float32, device=dev),
                   terminal=torch.empty(count, dtype=torch.uint8, device=dev),
                   priority=torch.empty(count, dtype=torch.float32, device=dev))
        with torch.cuda.device(dev):
            check(_lib.load().tq_transition_unpack(d, _ptr(buf), self.capacity, int(first), count,
                                                   _ptr(out["perspective"]), _ptr(out["next_perspective"]),
                                                   _ptr(out["action"]), _ptr(out["reward"]),
                                                   _ptr(out["terminal"]), _ptr(out["priority"]), _stream()))
        return out

    def computePriorities(self, no_envs, steps, q_values=None, discount=0.95):
        """computePrioritiesParallel (util_actor.py:268-287) into the block's priority section for
        the ``steps`` steps of ``no_envs`` lattices it holds (slot t*no_envs + e).  ``q_values``:
        device f32 (steps+1, no_envs, 3) -- the q_values of every step plus the step after -- or
        None for all-zero Q (pure exploration).  No synchronisation."""
        if q_values is not None:
            if (q_values.dtype != torch.float32 or not q_values.is_contiguous()
                    or q_values.numel() != (int(steps) + 1) * int(no_envs) * 3 or q_values.device != self.buf.device):
                raise ValueError("q_values must be a contiguous float32 device tensor of shape (steps+1, no_envs, 3)")
        with torch.cuda.device(self.buf.device):
            check(_lib.load().tq_block_priorities(self.d, _ptr(self.buf), self.capacity, int(no_envs), int(steps),
                                                  _ptr(q_values), float(discount), _stream()))


transition_dtype = wire.transition_type     # the reference's replay record (Actor_mp.py:52-56, util.py:10)


def to_structured(unpacked, size):
    """dict from TransitionBlock.unpack / generateTransition -> numpy array of transition_dtype."""
    n = unpacked["perspective"].shape[0]
    rec = np.empty(n, dtype=transition_dtype(size))
    get = lambda k: unpacked[k].cpu().numpy() if torch.is_tensor(unpacked[k]) else np.asarray(unpacked[k])
    a = get("action")
    rec['perspective'] = get("perspective")
    rec['next_perspective'] = get("next_perspective")
    rec['action']['position'] = a[:, :3]
    rec['action']['op'] = a[:, 3]
    rec['reward'] = get("reward")
    rec['terminal'] = get("terminal").astype(bool)
    return rec


def configured_xcd_bias():
    """The process-wide workgroup-share setting (toricenv.h: tq_set_xcd_bias; the library's default or TORICENV_XCD_BIAS).
    pickStackBuffer's check decides per EnvSet (tq_env_set_xcd_bias) and leaves this alone."""
    return int(_lib.load().tq_get_xcd_bias())


def set_xcd_bias(bias):
    """tq_set_xcd_bias for this process."""
    check(_lib.load().tq_set_xcd_bias(int(bias)))


class EnvSet:
    """Batch of N toric-code lattices resident on one MI355X (reference: src/EnvSet.py:4-51).

    ``env`` is a :class:`ToricEnv` (or anything with ``system_size`` and optionally
    ``p_error`` / ``terminal_reward``).  ``first_env_id`` is the shard offset of this handle's
    lattices in the global env numbering (RNG is keyed by global id, so any partition over
    GPUs yields identical lattices).
    """

    def __init__(self, env, no_envs, device=None, seed=None, first_env_id=0, numpy_io=True,
                 max_steps_per_episode=75):
        self._h = C.c_void_p(None)
        self.size = int(env.system_size)
        self.no_envs = int(no_envs)
        self.numpy_io = bool(numpy_io)
        self.device = _require_gpu(device if device is not None else getattr(env, "device", None))
        self.seed = int(getattr(env, "seed", 0) if seed is None else seed)
        self.first_env_id = int(first_env_id)
        self.p_error = float(getattr(env, "p_error", 0.1))
        self.terminal_reward = float(getattr(env, "terminal_reward", 100.0))
        self.max_steps_per_episode = int(max_steps_per_episode)
        self._L = _lib.load()
        with torch.cuda.device(self.device):
            check(self._L.tq_create(C.byref(self._h), self.no_envs, self.size, self.device.index,
                                    C.c_uint64(self.seed & 0xFFFFFFFFFFFFFFFF), self.first_env_id))
        # gym config "min_qubit_errors" (always 0 in the reference's own configs): n > 0 = fixed-n sampler,
        # which does not use p_error -- set first, so that {min_qubit_errors: n, p_error: 0} is a valid config
        self.min_qubit_errors = int(getattr(env, "min_qubit_errors", 0))
        if self.min_qubit_errors > 0:
            check(self._L.tq_set_min_qubit_errors(self._h, self.min_qubit_errors))
        check(self._L.tq_set_params(self._h, self.p_error, self.terminal_reward, self.max_steps_per_episode))
        n, d, dev = self.no_envs, self.size, self.device
        self._state_u8 = torch.zeros((n, 2, d, d), dtype=torch.uint8, device=dev)
        self._rewards = torch.zeros(n, dtype=torch.float32, device=dev)
        self._terminals = torch.zeros(n, dtype=torch.uint8, device=dev)
        self._actions = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        self._qv = torch.zeros((n, 3), dtype=torch.float32, device=dev)
        self._counts = torch.zeros(n, dtype=torch.int32, device=dev)
        self._offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        self._positions = None
        # attributes of the reference class (EnvSet.py:9-11)
        self.states = np.zeros((n, 2, d, d), dtype=np.int64)
        self.rewards = np.zeros(n)
        self.terminals = np.zeros(n, dtype=bool)

    # ------------------------------------------------------------------ plumbing
    def close(self):
        self._parked = []
        self.__dict__.pop("_stack_cache", None)
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.tq_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, fn, *args):
        with torch.cuda.device(self.device):
            check(fn(self._h, *args, _stream()))

    def _dev(self, x, dtype):
        if x is None:
            return None
        if torch.is_tensor(x):
            return x.to(device=self.device, dtype=dtype).contiguous()
        x = np.array(x, copy=True) if isinstance(x, np.ndarray) and not x.flags.writeable else np.ascontiguousarray(x)
        return torch.as_tensor(x, device=self.device).to(dtype).contiguous()

    def check(self):
        """Raise if a kernel latched an error (bad action / capacity).  Synchronises."""
        self._call(self._L.tq_check)

    def set_perror_schedule(self, strategy, p_start, p_final, p_delta):
        """Reset policy of the actor (Actor_mp.py:41-46,176-180) used by actorStep."""
        with torch.cuda.device(self.device):
            check(self._L.tq_set_perror_schedule(self._h, _STRATEGY[strategy], float(p_start), float(p_final),
                                                 float(p_delta)))

    # ------------------------------------------------------------------ reference surface
    def resetAll(self, p_errors=None):
        """EnvSet.py:29-36 -> states (N,2,d,d) (int64 numpy / uint8 tensor)."""
        p = self._dev(p_errors, torch.float64)
        if p is not None and p.numel() != self.no_envs:
            raise ValueError("p_errors must have one entry per env")
        self._call(self._L.tq_reset_all, _ptr(p))
        return self._return_states()

    def resetTerminalEnvs(self, idx, p_errors=None):
        """EnvSet.py:19-27 -> states of the reset lattices (len(idx),2,d,d) (float64 numpy)."""
        idx_t = self._dev(idx, torch.int32)
        k = int(idx_t.numel())
        p = self._dev(p_errors, torch.float64)
        if p is not None and p.numel() != k:
            raise ValueError("p_errors must have one entry per idx")
        if self.numpy_io and k:
            idx_np = idx_t.cpu().numpy()
            if idx_np.min() < 0 or idx_np.max() >= self.no_envs or np.unique(idx_np).size != k:
                raise ValueError("idx must be distinct env indices in range")
        out = torch.empty((k, 2, self.size, self.size), dtype=torch.uint8, device=self.device)
        if k:
            # the device checks idx as well (range, duplicates) and latches TQ_E_INDEX for check()
            self._call(self._L.tq_reset_idx, _ptr(idx_t), k, _ptr(p))
            self._call(self._L.tq_get_state_idx, _ptr(idx_t), k, _ptr(out))
        return out.cpu().numpy().astype(np.float64) if self.numpy_io else out

    def step(self, actions):
        """EnvSet.py:38-47 -> (states, rewards, terminals, info)."""
        a = self._dev(actions, torch.int32)
        if a.numel() != 4 * self.no_envs:
            raise ValueError("actions must be (no_envs, 4)")
        self._actions.copy_(a.reshape(self.no_envs, 4))
        self._call(self._L.tq_step, _ptr(self._actions), _ptr(self._rewards), _ptr(self._terminals))
        states = self._return_states()
        if self.numpy_io:
            self.check()
            self.rewards = self._rewards.cpu().numpy().astype(np.float64)
            self.terminals = self._terminals.cpu().numpy().astype(bool)
            return states, self.rewards, self.terminals, {}
        return states, self._rewards, self._terminals, {}

    def _return_states(self):
        self._call(self._L.tq_get_state, _ptr(self._state_u8))
        if self.numpy_io:
            self.states = self._state_u8.cpu().numpy().astype(np.int64)
            return self.states
        return self._state_u8

    def getStates(self):
        return self._return_states()

    def getQubits(self):
        q = torch.empty((self.no_envs, 2, self.size, self.size), dtype=torch.uint8, device=self.device)
        self._call(self._L.tq_get_qubits, _ptr(q))
        return q.cpu().numpy().astype(np.int64) if self.numpy_io else q

    def setQubits(self, qubits):
        q = self._dev(qubits, torch.uint8)
        if q.numel() != self.no_envs * 2 * self.size * self.size:
            raise ValueError("qubits must be (no_envs, 2, d, d)")
        self._call(self._L.tq_set_qubits, _ptr(q))

    def getCounters(self):
        ep = torch.empty(self.no_envs, dtype=torch.int32, device=self.device)
        st = torch.empty(self.no_envs, dtype=torch.int32, device=self.device)
        self._call(self._L.tq_get_counters, _ptr(ep), _ptr(st))
        return (ep.cpu().numpy(), st.cpu().numpy()) if self.numpy_io else (ep, st)

    def evalGroundState(self):
        out = torch.empty(self.no_envs, dtype=torch.uint8, device=self.device)
        self._call(self._L.tq_eval_ground_state, _ptr(out))
        return out.cpu().numpy().astype(bool) if self.numpy_io else out

    def isTerminal(self):
        out = torch.empty(self.no_envs, dtype=torch.uint8, device=self.device)
        self._call(self._L.tq_is_terminal, _ptr(out))
        return out.cpu().numpy().astype(bool) if self.numpy_io else out

    # ------------------------------------------------------------------ perspectives
    def perspectiveCounts(self, offsets=None):
        """-> (counts i32[N], offsets i64[N+1]) device tensors; no synchronisation.  ``offsets``:
        optional caller-owned int64[N+1] tensor to receive the scan instead of the internal one."""
        if offsets is not None:
            if offsets.dtype != torch.int64 or offsets.numel() != self.no_envs + 1 or not offsets.is_contiguous():
                raise ValueError("offsets must be a contiguous int64 tensor of no_envs + 1 elements")
            self._offsets = offsets
        self._call(self._L.tq_persp_count, _ptr(self._counts), _ptr(self._offsets))
        return self._counts, self._offsets

    def writePerspectives(self, out, positions=None, offsets=None, first=0, count=None):
        """Write the stack for ``offsets`` (default: the last perspectiveCounts) into the
        caller's tensor ``out`` (capacity = out.shape[0] perspectives).  No synchronisation.
        ``first`` / ``count``: only the lattices [first, first+count), their first perspective at
        out[0] -- for consumers that walk the batch in chunks."""
        if out.dtype not in _DTYPES or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32/float16/bfloat16/uint8 tensor")
        nq = 2 * self.size * self.size
        cap = out.numel() // nq
        if positions is not None and (positions.dtype != torch.int32 or positions.numel() < 3 * cap):
            raise ValueError("positions must be int32 with at least 3*capacity elements")
        off = self._offsets if offsets is None else offsets
        if first == 0 and count is None:
            self._call(self._L.tq_persp_write, _ptr(off), _ptr(out), _ptr(positions), cap, _DTYPES[out.dtype])
        else:
            count = self.no_envs - int(first) if count is None else int(count)
            self._call(self._L.tq_persp_write_range, _ptr(off), int(first), count, _ptr(out), _ptr(positions), cap,
                       _DTYPES[out.dtype])
        self._positions = positions

    def pickStackBuffer(self, candidates=4, dtype=torch.float32, capacity=None, positions=None, launches=10,
                        kinds=("torch", "chunked"), park=False, first=0, count=None, timer=None, passes=2, among=None,
                        check_shares=True, extend_if_uniform=True):
        """Set-up helper: allocate ``candidates`` stack buffers (``capacity`` perspectives each, default the worst
        case no_envs * 2*d*d), time the stack write on each of them and keep the fastest.  On MI355X the rate of a
        write stream into a buffer depends on the buffer AND on the stream's shape (5.2-6.9 TB/s for this kernel,
        from allocation to allocation; a buffer that is fast for the f32 stack can be slow for the bf16 stack of the
        same lattices: profiles/r04_stream_tune_d7_all.txt), and a caller writes the same buffer every step, so the
        choice is worth a few dozen launches at set-up -- per dtype.
        ALL candidates are allocated first (and stay allocated while the timing runs), then every candidate is timed
        ``passes`` times in turn, ``launches`` writes in all, and the MEDIAN decides.  ``timer(stack, k)`` -> list of
        k write times in ms: what is timed -- default: scan + write of the current lattices, back to back;
        ExploreLoop.time_writes times the write inside the caller's loop, the env kernels beside it.
        ``kinds``: where the candidates come from -- kinds[0] for candidate 0, the rest cyclically for the others:
        "torch" = torch.empty (candidate 0 by default: what a caller has without this helper), "chunked" = alloc_stack
        (2 MiB physical chunks).  ``among``: time these tensors instead of allocating (a re-probe of parked candidates).
        -> (stack tensor (capacity,2,d,d), report dict: median / min ms and kind of every candidate, which was kept).
        ``park``: keep the rejected candidates allocated until releaseParked() / close() instead of freeing them here
        -- the driver wipes freed device memory in the background, tens of GB of it take HBM bandwidth away from
        whatever runs in the next tens of milliseconds (a benchmark's timed region, say).
        ``first`` / ``count``: the default timer writes that lattice range only (a consumer that walks the batch in ranges
        with a small buffer: ``capacity`` is then the small buffer's).  ``check_shares``: also time the kept buffer with
        equal shares per workgroup against the library's unequal ones (toricenv.h: tq_set_xcd_bias) and keep the faster
        setting (report["xcd_bias"]).  ``extend_if_uniform``: when no candidate writes 7 % faster than candidate 0, allocate
        and time as many candidates again (the first ones stay allocated; bounded by half of the free memory) -- once.
        Synchronises; never call it in the step loop."""
        d, nq = self.size, 2 * self.size * self.size
        cap = self.no_envs * nq if capacity is None else int(capacity)
        if positions is None:
            positions = torch.empty((cap, 3), dtype=torch.int32, device=self.device)
        keep, used = [], []
        if among is not None:
            keep = list(among)
            used = ["re-probed"] * len(keep)
        else:
            # all candidates exist at once: never more of them than half of the free device memory holds
            nbytes = cap * nq * torch.empty((), dtype=dtype).element_size()
            with torch.cuda.device(self.device):
                free = torch.cuda.mem_get_info()[0]
            wanted = max(1, int(candidates))
            room = int(0.5 * free // max(nbytes, 1))              # candidates that may exist at once
            fit = max(1, min(wanted, room))

            def allocate(count):
                """``count`` more candidates (fewer when the device runs out); -> how many there are now."""
                for _ in range(count):
                    k = len(keep)
                    kind = kinds[0] if k == 0 or len(kinds) == 1 else kinds[1 + (k - 1) % (len(kinds) - 1)]
                    c = None
                    try:
                        if kind == "chunked":
                            try:
                                c = alloc_stack(cap, d, dtype, self.device)
                            except _lib.ToricEnvError:            # no virtual-memory API on this driver (or no memory): plain allocation
                                kind = "torch"
                        if c is None:
                            c = torch.empty((cap, 2, d, d), dtype=dtype, device=self.device)
                    except torch.OutOfMemoryError:
                        if not keep:
                            raise
                        break                                     # the candidates that exist will do
                    used.append("torch.empty" if kind == "torch" else "alloc_stack (2 MiB chunks)")
                    keep.append(c)
                    c = None
                return len(keep)
            allocate(fit)
        if timer is None:
            off = self.perspectiveCounts()[1].clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def timer(stack, k):
                out = []
                for r in range(k + 1):
                    e0.record()
                    self.writePerspectives(stack, positions, off, first=first, count=count)
                    e1.record()
                    e1.synchronize()
                    out.append(e0.elapsed_time(e1))
                return out[1:]
        torch.cuda.synchronize(self.device)
        per_pass = max(1, int(launches) // max(1, int(passes)))
        samples = [[] for _ in keep]
        for _ in range(max(1, int(passes))):
            for i, c in enumerate(keep):
                samples[i] += list(timer(c, per_pass))
        # No candidate stands out (every one within 7 % of candidate 0)?  On some boxes the first tens of GB a process
        # allocates ALL write at the slow rate and faster buffers only turn up behind them (profiles/README.md, round 4:
        # 24 x 2.5 GB at 0.358-0.368 ms, then, for the next leg, candidates 12 and 16-20 of 24 x 5 GB at 0.57 against 0.71 ms).
        # So the search goes on once, with as many candidates again, WHILE the first ones stay allocated.
        extended = 0
        if among is None and extend_if_uniform and len(keep) >= 3 and len(keep) < room:
            ms0 = [float(np.median(x)) for x in samples]
            if min(ms0) > 0.93 * ms0[0] and ms0[0] >= 0.1:         # (a write of under 0.1 ms is not about bandwidth)
                before = len(keep)
                extended = allocate(min(before, room - before)) - before
                samples += [[] for _ in range(extended)]
                torch.cuda.synchronize(self.device)
                for _ in range(max(1, int(passes))):
                    for i in range(before, len(keep)):
                        samples[i] += list(timer(keep[i], per_pass))
        self.check()
        ms = [float(np.median(x)) for x in samples]
        chosen = int(np.argmin(ms))
        best = keep[chosen]
        rejected = [x for x in keep if x is not best]
        if park:
            self._parked = [x for x in getattr(self, "_parked", []) if all(x is not y for y in keep)] + rejected
        report = {"candidates": len(ms), "candidates_asked": int(candidates) if among is None else len(ms), "candidates_added_because_uniform": extended, "write_ms": ms, "write_ms_min": [float(min(x)) for x in samples], "chosen": chosen,
                  "probe_ms_chosen": ms[chosen], "writes_per_candidate": len(samples[0]), "kinds": used,
                  "addresses": [hex(x.data_ptr()) for x in keep]}
        # The shares of the write's workgroups (tq_set_xcd_bias: the even XCDs' workgroups take more of the stack) against
        # equal shares, on the buffer that was kept: the setting rests on a measured asymmetry of MI355X, so it is checked
        # where it is used.  The outcome is set on this EnvSet's handle only; d <= 5 and u8 stacks never use unequal shares.
        if check_shares and d >= 7 and dtype != torch.uint8:
            L = self._L
            b0 = int(L.tq_get_xcd_bias())                       # the process-wide setting; the decision is this HANDLE's own
            if b0 > 0:
                check(L.tq_env_set_xcd_bias(self._h, 0))
                eq = float(np.median(list(timer(best, per_pass)) + list(timer(best, per_pass))))
                check(L.tq_env_set_xcd_bias(self._h, b0))
                un = float(np.median(list(timer(best, per_pass)) + list(timer(best, per_pass))))
                keep_bias = un <= eq
                check(L.tq_env_set_xcd_bias(self._h, -1 if keep_bias else 0))
                report["xcd_bias"] = {"bias": b0 if keep_bias else 0, "write_ms_biased": un, "write_ms_equal_shares": eq}
                report["probe_ms_chosen"] = min(un, eq)
            else:
                report["xcd_bias"] = {"bias": 0}
        del keep, rejected
        if not park:
            torch.cuda.empty_cache()
        if len(ms) > 2 and min(ms) > 0.9 * ms[0]:
            report["uniform"] = ("no candidate writes more than 10 % faster than candidate 0: on some boxes every buffer -- and "
                                 "every write stream, hipMemset included -- runs at one rate (profiles/r03_stack_write_ab.txt)")
        return best, report

    def releaseParked(self):
        """Free the candidates pickStackBuffer(park=True) kept."""
        self._parked = []
        torch.cuda.empty_cache()

    REUSED_PROBE_CANDIDATES = 4     # candidates of the one-off probe behind generatePerspectiveReused (0 / 1 = no probe)
    REUSED_HEADROOM = 1.5           # capacity of the re-used buffer = observed perspectives x this (grown when exceeded)

    def generatePerspectiveReused(self, dtype=torch.float32, capacity=None):
        """generatePerspective for the current states into ONE stack buffer this EnvSet keeps and re-uses, for loops that
        consume the stack at once (the policy's forward pass: numba/util_actor.py:39-46) -- no allocation per step, and
        the buffer comes from the same probe bench.py uses: at first use (and again when the dtype changes or the
        buffer has to grow) ``REUSED_PROBE_CANDIDATES`` candidates (one torch.empty, the rest tq_stack_alloc) are timed
        with the stack write of the current lattices and the fastest is kept (pickStackBuffer; ~0.1-1 s, once).
        ``capacity`` (perspectives; default: the observed count x REUSED_HEADROOM, at most the worst case
        no_envs * 2*d*d): 1.0 GB instead of 2.5 GB at 65 536 lattices of d=7 in f32.  When a later step has more
        perspectives than the buffer holds it is re-allocated larger (the count is read back before the write anyway).
        -> (perspectives (P,2,d,d), positions (P,3), counts (N,)) as VIEWS of that buffer: valid until the next call."""
        d, nq = self.size, 2 * self.size * self.size
        counts, offsets = self.perspectiveCounts()
        P = int(offsets[-1].item())
        worst = self.no_envs * nq
        cache = self.__dict__.get("_stack_cache")
        if cache is None or cache["dtype"] != dtype or cache["capacity"] < P or (capacity is not None and cache["capacity"] < min(int(capacity), worst)):
            want = min(worst, max(int(capacity) if capacity is not None else int(P * self.REUSED_HEADROOM) + 1024, P, 1024))
            self.__dict__.pop("_stack_cache", None)
            cache = None
            torch.cuda.empty_cache()
            try:
                pos = alloc_chunked((want, 3), torch.int32, self.device)
            except _lib.ToricEnvError:                            # no virtual-memory API on this driver
                pos = torch.empty((want, 3), dtype=torch.int32, device=self.device)
            k = int(self.REUSED_PROBE_CANDIDATES)
            if k > 1 and P > 0:
                buf, report = self.pickStackBuffer(k, dtype=dtype, capacity=want, positions=pos, launches=6, passes=2)
            else:
                try:
                    buf = alloc_stack(want, d, dtype, self.device)
                except _lib.ToricEnvError:
                    buf = torch.empty((want, 2, d, d), dtype=dtype, device=self.device)
                report = None
            cache = self._stack_cache = {"dtype": dtype, "capacity": want, "buf": buf, "pos": pos, "probe": report}
        buf, pos = cache["buf"], cache["pos"]
        if P:
            self.writePerspectives(buf, pos, offsets)
        self._positions = pos[:P]
        return buf[:P], pos[:P], counts

    def reusedStackBacking(self):
        """The whole buffer behind the last generatePerspectiveReused result (rows past P are slack)."""
        c = self.__dict__.get("_stack_cache")
        return None if c is None else c["buf"]

    def generatePerspective(self, states=None, dtype=torch.float32):
        """generatePerspectiveBatch + concatenate (numba/util_actor.py:33-39,56-67) for the current
        states, or for an explicit ``states`` array (n,2,d,d) like the reference's function
        -> (perspectives (P,2,d,d), positions (P,3), counts (N,)).
        Reads P back from the device (one 8-byte copy), like the reference's data-dependent shape."""
        if states is not None:
            out, pos, counts = generatePerspectiveBatch(self.size // 2, self.size, states, dtype=dtype, device=self.device)[:3]
            if self.numpy_io:
                return out.cpu().numpy(), pos.cpu().numpy().astype(np.int64), counts.cpu().numpy().astype(np.int64)
            return out, pos, counts
        counts, offsets = self.perspectiveCounts()
        P = int(offsets[-1].item())
        d = self.size
        out = torch.empty((P, 2, d, d), dtype=dtype, device=self.device)
        pos = torch.empty((P, 3), dtype=torch.int32, device=self.device)
        if P:
            self.writePerspectives(out, pos, offsets)
        self._positions = pos
        if self.numpy_io:
            return out.cpu().numpy(), pos.cpu().numpy().astype(np.int64), counts.cpu().numpy().astype(np.int64)
        return out, pos, counts

    # ------------------------------------------------------------------ policy glue
    def selectAction(self, q_table, eps, positions=None, offsets=None):
        """_selectActionBatch_prime on the device -> (actions (N,4), q_values (N,3)).
        q_table None = pure exploration (every eps must be 1)."""
        pos = self._positions if positions is None else positions
        if pos is None:
            raise ValueError("call generatePerspective / writePerspectives (with positions) first")
        pos = self._dev(pos, torch.int32)
        off = self._offsets if offsets is None else self._dev(offsets, torch.int64)
        q = self._dev(q_table, torch.float32)
        e = None
        if q is not None:
            e = self._dev(np.broadcast_to(np.asarray(eps, np.float64), (self.no_envs,)) if not torch.is_tensor(eps) else eps,
                          torch.float64)
        self._call(self._L.tq_select_action, _ptr(q), _ptr(off), _ptr(pos), _ptr(e), _ptr(self._actions), _ptr(self._qv))
        if self.numpy_io:
            return self._actions.cpu().numpy().astype(np.int64), self._qv.cpu().numpy()
        return self._actions, self._qv

    def generateTransition(self, actions):
        """generateTransitionParallel for the last step() (util_actor.py:223-264)
        -> dict(perspective u8, next_perspective u8, action i32[N,4]) of device tensors
        (numpy arrays with numpy_io)."""
        a = self._dev(actions, torch.int32)
        n, d, dev = self.no_envs, self.size, self.device
        out = dict(perspective=torch.empty((n, 2, d, d), dtype=torch.uint8, device=dev),
                   next_perspective=torch.empty((n, 2, d, d), dtype=torch.uint8, device=dev),
                   action=torch.empty((n, 4), dtype=torch.int32, device=dev))
        self._call(self._L.tq_transition_write, _ptr(a), _ptr(out["perspective"]), _ptr(out["next_perspective"]),
                   _ptr(out["action"]))
        if self.numpy_io:
            self.check()
            return {k: v.cpu().numpy() for k, v in out.items()}
        return out

    def newTransitionBlock(self, steps=1):
        return TransitionBlock(self.size, self.no_envs * int(steps), self.device)

    def actorStep(self, actions=None, block=None, slot=0, want_actions=True):
        """Fused step -> transition -> auto-reset -> counts (Actor_mp.py:116-183).
        actions None = pure exploration drawn in-kernel.  ``block``/``slot``: lattice e writes
        transition slot ``slot*no_envs + e`` of the TransitionBlock.
        -> (actions_taken i32[N,4], rewards f32[N], terminals u8[N]) device tensors."""
        a = self._dev(actions, torch.int32)
        blk_ptr, cap, base = C.c_void_p(0), 0, 0
        if block is not None:
            blk_ptr, cap, base = _ptr(block.buf), block.capacity, int(slot) * self.no_envs
        self._call(self._L.tq_actor_step, _ptr(a), _ptr(self._actions) if want_actions else C.c_void_p(0),
                   _ptr(self._rewards), _ptr(self._terminals), blk_ptr, cap, base)
        if self.numpy_io:
            self.check()
            return (self._actions.cpu().numpy().astype(np.int64), self._rewards.cpu().numpy().astype(np.float64),
                    self._terminals.cpu().numpy().astype(bool))
        return self._actions, self._rewards, self._terminals


_reserved = {}      # (device index, d) -> states the device's stateless scratch is sized for


def _reserve_states(dev, d, n):
    """tq_states_reserve is a set-up call (allocates, synchronises): issue it only when a larger
    batch than ever before arrives on this device; the tq_states_* calls themselves never allocate."""
    key = (dev.index, d)
    if _reserved.get(key, 0) < n:
        with torch.cuda.device(dev):
            check(_lib.load().tq_states_reserve(d, n))
        _reserved[key] = n                                # the scratch only ever grows


def generatePerspectiveBatch(grid_shift, toric_size, states, dtype=torch.float32, device=None, return_offsets=False):
    """numba/util_actor.py:56-67 for syndromes that do not live in an EnvSet (e.g. the learner's
    next_state batch, util_learner.py:48-111).  states: (n,2,d,d) numpy / tensor.
    -> (perspectives (P,2,d,d) tensor, positions (P,3) i32 tensor, counts (n,) i32 tensor)
    [+ offsets (n+1,) i64 tensor, the exclusive scan the kernels produced, with ``return_offsets``].
    One 8-byte read-back of P (the output shape is data dependent, as upstream)."""
    dev = _require_gpu(device)
    if int(grid_shift) != int(toric_size) // 2:
        raise ValueError("grid_shift must be int(toric_size/2) (Actor_mp.py:59)")
    L = _lib.load()
    st = states if torch.is_tensor(states) else torch.as_tensor(np.ascontiguousarray(states))
    st = st.to(device=dev, dtype=torch.uint8).contiguous()
    n, d = int(st.shape[0]), int(toric_size)
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    offsets = torch.empty(n + 1, dtype=torch.int64, device=dev)
    _reserve_states(dev, d, n)
    with torch.cuda.device(dev):
        rc = L.tq_states_persp_count(d, n, _ptr(st), _ptr(counts), _ptr(offsets), _stream())
        if rc == _lib.TQ_E_CAPACITY:                           # the python-side record of the scratch size was stale
            _reserved.pop((dev.index, d), None)
            _reserve_states(dev, d, n)
            rc = L.tq_states_persp_count(d, n, _ptr(st), _ptr(counts), _ptr(offsets), _stream())
        check(rc)
        P = int(offsets[-1].item())
        out = torch.empty((P, 2, d, d), dtype=dtype, device=dev)
        pos = torch.empty((P, 3), dtype=torch.int32, device=dev)
        if P:
            check(L.tq_states_persp_write(d, n, _ptr(st), _ptr(offsets), _ptr(out), _ptr(pos), P, _DTYPES[dtype],
                                          _stream()))
    return (out, pos, counts, offsets) if return_offsets else (out, pos, counts)


def generateTransitionParallel(action, reward, state, next_state, terminal_state, grid_shift, trans_type=None,
                               device=None):
    """Drop-in for src/util_actor.py:223-264 on the GPU: same arguments, returns a numpy record
    array of ``trans_type`` (default: transition_dtype(size), Actor_mp.py:52-56)."""
    dev = _require_gpu(device)
    L = _lib.load()
    st = torch.as_tensor(np.ascontiguousarray(state)).to(device=dev, dtype=torch.uint8).contiguous()
    nst = torch.as_tensor(np.ascontiguousarray(next_state)).to(device=dev, dtype=torch.uint8).contiguous()
    act = torch.as_tensor(np.ascontiguousarray(action)).to(device=dev, dtype=torch.int32).contiguous()
    n, d = int(nst.shape[0]), int(nst.shape[-1])
    if int(grid_shift) != d // 2:
        raise ValueError("grid_shift must be int(toric_size/2) (Actor_mp.py:59)")
    out = dict(perspective=torch.empty((n, 2, d, d), dtype=torch.uint8, device=dev),
               next_perspective=torch.empty((n, 2, d, d), dtype=torch.uint8, device=dev),
               action=torch.empty((n, 4), dtype=torch.int32, device=dev))
    with torch.cuda.device(dev):
        check(L.tq_states_transition(d, n, _ptr(st), _ptr(nst), _ptr(act), _ptr(out["perspective"]),
                                     _ptr(out["next_perspective"]), _ptr(out["action"]), _stream()))
        check(L.tq_states_check(_stream()))                       # bad action -> ValueError
    out["reward"] = np.asarray(reward, np.float64)
    out["terminal"] = np.asarray(terminal_state, bool)
    rec = to_structured(out, d)
    return rec if trans_type is None else rec.astype(trans_type)
