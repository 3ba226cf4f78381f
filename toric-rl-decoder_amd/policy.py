"""Policy-side glue around the env kernels (callers of the hot path).

The Q-network itself is stock torch convolution work and out of scope; NN_11 is restated here
(30 lines, same parameter names as the reference so its state_dicts load unchanged) only so that
BASELINE configs[2] ("generatePerspective feeding NN_11 policy for selectAction") and the
evaluation loop can run without the reference's Python.

  selectActionBatch    src/numba/util_actor.py:11-53
  predictMaxOptimized  src/util_learner.py:48-111
  evaluate             src/evaluation.py:10-124
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import check
from .envset import EnvSet, ToricEnv, _ptr, _stream, generatePerspectiveBatch


class NN_11(nn.Module):
    """11 conv3x3 + ReLU then one linear layer (src/nn/torch/NN.py:10-45); the first and the last
    convolution are unpadded after a circular pad of 1 (src/nn/torch/util.py:21-26)."""

    CHANNELS = (2, 128, 128, 120, 111, 104, 103, 90, 80, 73, 71, 64)

    def __init__(self, system_size, number_of_actions=3, device=None):
        super().__init__()
        ch = self.CHANNELS
        for i in range(11):
            pad = 0 if i in (0, 10) else 1
            setattr(self, f"conv{i + 1}", nn.Conv2d(ch[i], ch[i + 1], kernel_size=3, stride=1, padding=pad))
        self.linear1 = nn.Linear(64 * (system_size - 2) ** 2, number_of_actions)
        self.device = device

    def forward(self, x):
        x = F.pad(x, (1, 1, 1, 1), mode="circular")
        for i in range(11):
            x = F.relu(getattr(self, f"conv{i + 1}")(x))
        return self.linear1(x.flatten(1))


def _forward_chunked(model, persp, chunk, pad_to=1024, backing=None, out=None):
    """model(persp) in chunks of `chunk` rows -> (rows, 3) float32.  The number of perspectives changes every step, and
    every new batch size is a new problem for MIOpen's solver search, so the last, ragged chunk is run at a row count
    rounded up to a multiple of `pad_to` (its surplus Q rows are cut off again): a handful of shapes instead of
    thousands.  The surplus rows come from the slack of ``backing`` -- the re-used stack buffer ``persp`` is a view
    of (EnvSet.reusedStackBacking(): rows past P hold earlier perspectives or zeros) -- when it has enough of it, from a
    zero-filled copy otherwise.  Batch rows do not influence each other, but a different row count may select a
    different convolution kernel and summation order: Q-values agree with the unpadded forward to ~1e-6, bit-identity
    needs ``pad_to=1`` (tests/test_gpu_policy.py pins both, and that the greedy choice is the same on the golden
    weights).  ``out``: optional (>= rows, 3) float32 tensor to receive the Q-values (no torch.cat)."""
    n = persp.shape[0]
    if out is None:
        out = torch.empty((n, 3), dtype=torch.float32, device=persp.device)
    with torch.no_grad():
        for i in range(0, n, chunk):
            x = persp[i:i + chunk]
            r = x.shape[0]
            if r < chunk and pad_to > 1 and r % pad_to:
                rows = min(chunk, (r + pad_to - 1) // pad_to * pad_to)
                if backing is not None and backing.data_ptr() == persp.data_ptr() and backing.shape[0] >= i + rows:
                    xp = backing[i:i + rows]
                else:
                    xp = torch.zeros((rows,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
                    xp[:r] = x
                out[i:i + r] = model(xp)[:r]
            else:
                out[i:i + r] = model(x)
    return out[:n]


def selectActionEnvSet(envs, model, epsilon, dtype=torch.float32, chunk=1 << 16):
    """The reference's selectActionBatch for the lattices that live in ``envs`` (an EnvSet), everything
    on the device: perspectives -> model forward (in chunks) -> epsilon-greedy selection keyed by each
    lattice's own (episode, step) counters.
    -> (actions (N,4), q_values (N,3)); numpy with envs.numpy_io, device tensors otherwise."""
    model.eval()
    io = envs.numpy_io
    envs.numpy_io = False
    try:
        persp, pos, _ = envs.generatePerspectiveReused(dtype=dtype)       # consumed at once by the forward pass
        q = _forward_chunked(model, persp, chunk, backing=envs.reusedStackBacking())
        act, qv = envs.selectAction(q, epsilon, positions=pos)
    finally:
        envs.numpy_io = io
    if io:
        return act.cpu().numpy().astype(np.int64), qv.cpu().numpy()
    return act, qv


# The reference's selectActionBatch draws from numpy's global RNG, which nothing ever seeds
# (numba/util_actor.py:49,97-98).  Here the draws are Philox keyed (seed, state index, call number):
# seed_select() fixes the stream; every selectActionBatch call advances the call number.
_select_rng = {"seed": 0, "calls": 0}


def seed_select(seed, calls=0):
    _select_rng["seed"] = int(seed) & 0xFFFFFFFFFFFFFFFF
    _select_rng["calls"] = int(calls)


def selectActionBatch(number_of_actions, epsilon, grid_shift, toric_size, state, model, device, chunk=1 << 16):
    """Drop-in for src/numba/util_actor.py:11-53 -- same argument names, order and return types:
    ``state`` (N,2,d,d) numpy array (or tensor), ``epsilon`` scalar or (N,) array, ``device`` the
    ROCm device the model lives on -> (actions (N,4) int64 numpy, q_values (N,3) float64 numpy).
    Perspective generation, the model forward and the epsilon-greedy selection (greedy iff
    (1-eps) > U; first maximum in row-major order, :93-95; otherwise a uniform perspective and op,
    :97-98) all run on the device; only the (N,4) / (N,3) results come back.  A state without defects
    (the reference would raise on it, :93) gets action op 0 and zero q_values."""
    if int(number_of_actions) != 3:
        raise ValueError("number_of_actions must be 3 (env.action_space.high[-1], Actor_mp.py:58)")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise ValueError("device must be a cuda (ROCm) device: the toric env has no CPU path")
    model.eval()
    persp, pos, counts, offsets = generatePerspectiveBatch(grid_shift, toric_size, state, device=dev, return_offsets=True)
    dev = persp.device
    n = int(counts.numel())
    q = _forward_chunked(model, persp, chunk)
    eps = torch.as_tensor(np.array(np.broadcast_to(np.asarray(epsilon, np.float64), (n,))), device=dev)   # writable copy
    actions = torch.empty((n, 4), dtype=torch.int32, device=dev)
    qv = torch.empty((n, 3), dtype=torch.float32, device=dev)
    call = _select_rng["calls"]
    _select_rng["calls"] = call + 1
    with torch.cuda.device(dev):
        check(_lib.load().tq_states_select_action(n, _ptr(q), _ptr(offsets), _ptr(pos), _ptr(eps),
                                                  _select_rng["seed"], call, 0, _ptr(actions), _ptr(qv), _stream()))
    return actions.cpu().numpy().astype(np.int64), qv.cpu().numpy().astype(np.float64)


def _selectActionBatch_prime(q_values_table, splice_idx, positions, greedy, device=None):
    """Drop-in for src/numba/util_actor.py:69-107 on the device -- same arguments: the (P,3) Q-table of all
    perspectives, ``splice_idx`` = cumsum of the perspectives per state (:31), ``positions`` (P,3), ``greedy`` bool
    (N,) -> (actions (N,4) float64, q_values (N,3) float64) like the reference's np.empty defaults.  greedy[i]: the
    first (perspective, op) attaining the maximum of the state's slice in row-major order (:93-95); otherwise a
    uniform perspective and op (:97-98) from the Philox stream of seed_select() (upstream: numpy's unseeded global
    RNG).  A state with an empty slice (the reference raises on it) gets action op 0 and zero q_values."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise ValueError("device must be a cuda (ROCm) device: the toric env has no CPU path")
    q = torch.as_tensor(np.ascontiguousarray(q_values_table) if not torch.is_tensor(q_values_table) else q_values_table)
    q = q.to(device=dev, dtype=torch.float32).contiguous()
    sp = torch.as_tensor(np.asarray(splice_idx, np.int64) if not torch.is_tensor(splice_idx) else splice_idx).to(dev, torch.int64)
    n = int(sp.numel())
    offsets = torch.zeros(n + 2, dtype=torch.int64, device=dev)[:n + 1]       # even length behind it: 16-byte aligned rows
    offsets[1:] = sp
    pos = torch.as_tensor(np.ascontiguousarray(positions) if not torch.is_tensor(positions) else positions)
    pos = pos.to(device=dev, dtype=torch.int32).contiguous()
    if q.shape[0] != pos.shape[0] or (n and int(sp[-1].item()) != q.shape[0]):
        raise ValueError("q_values_table / positions / splice_idx do not describe the same perspectives")
    g = torch.as_tensor(np.asarray(greedy, bool) if not torch.is_tensor(greedy) else greedy).to(dev)
    if g.numel() != n:
        raise ValueError("greedy must have one entry per state")
    eps = (~g.bool()).to(torch.float64).contiguous()          # greedy iff (1 - eps) > U with U in [0,1): eps 0 -> always, 1 -> never
    actions = torch.empty((n, 4), dtype=torch.int32, device=dev)
    qv = torch.empty((n, 3), dtype=torch.float32, device=dev)
    call = _select_rng["calls"]
    _select_rng["calls"] = call + 1
    if n:
        with torch.cuda.device(dev):
            check(_lib.load().tq_states_select_action(n, _ptr(q), _ptr(offsets), _ptr(pos), _ptr(eps),
                                                      _select_rng["seed"], call, 0, _ptr(actions), _ptr(qv), _stream()))
    return actions.cpu().numpy().astype(np.float64), qv.cpu().numpy().astype(np.float64)


def segment_max(q_table, offsets, largest=None):
    """out[i] = max of q_table[offsets[i]:offsets[i+1]] (0 for an empty slice); with ``largest``
    (device int32[1]) the reference's zero padding is reproduced."""
    n = int(offsets.numel()) - 1
    out = torch.empty(n, dtype=torch.float32, device=q_table.device)
    with torch.cuda.device(q_table.device):
        check(_lib.load().tq_segment_max(_ptr(q_table.contiguous()), _ptr(offsets), n, _ptr(largest), _ptr(out),
                                         _stream()))
    return out


def predictMaxOptimized(model, batch_state, grid_shift, system_size, device, chunk=1 << 16):
    """util_learner.py:48-111: max Q-value of every state of the batch (0 for terminal states),
    perspectives generated and reduced on the device.  -> float32 tensor (n,) on ``device``."""
    model.eval()
    persp, pos, counts, offsets = generatePerspectiveBatch(grid_shift, system_size, batch_state, device=device,
                                                           return_offsets=True)
    q = _forward_chunked(model, persp, chunk)
    # the reference gives a terminal state one dummy perspective (:74-76), which counts for the padding
    largest = torch.clamp(counts.max(), min=1).to(torch.int32).reshape(1)
    return segment_max(q, offsets, largest)


def _greedy_episodes(envs, model, epsilon, num_of_steps, chunk):
    """The inner loop of evaluation.py:78-110 / results/small_p_error_test.py:131-151 for all lattices of ``envs`` side
    by side: select (policy forward + device selection) -> step, until every lattice is solved or ``num_of_steps``
    steps were taken.  -> dict of device tensors: done (syndrome cleared), steps per episode, sum / count of the
    chosen action's Q-value over all live steps."""
    n = envs.no_envs
    done = torch.zeros(n, dtype=torch.bool, device=envs.device)
    n_steps = torch.zeros(n, dtype=torch.int64, device=envs.device)
    q_sum = torch.zeros((), dtype=torch.float64, device=envs.device)
    q_cnt = torch.zeros((), dtype=torch.int64, device=envs.device)
    for _ in range(int(num_of_steps)):
        act, qv = selectActionEnvSet(envs, model, epsilon, chunk=chunk)    # op 0 for solved lattices
        live = ~done
        chosen = torch.gather(qv, 1, (act[:, 3].long() - 1).clamp(min=0).unsqueeze(1)).squeeze(1)
        q_sum += (chosen.double() * live).sum()
        q_cnt += live.sum()
        n_steps += live
        _, _, term, _ = envs.step(act)
        done = done | term.bool()
        if bool(done.all()):
            break
    return dict(done=done, n_steps=n_steps, q_sum=q_sum, q_cnt=q_cnt)


def evaluate(model, env, env_config, grid_shift, device, prediction_list_p_error, num_of_episodes=1,
             num_actions=3, epsilon=0.0, num_of_steps=50, plot_one_episode=False, minimum_nbr_of_qubit_errors=0,
             seed=0, chunk=1 << 16, round_like_reference=True):
    """evaluation.py:10-124 with the episodes of one p_error run side by side as one EnvSet.
    -> (error_corrected_list, ground_state_list, average_number_of_steps_list, mean_q_list, failed_syndroms).
    ``round_like_reference=False`` leaves steps (1 decimal upstream) and mean Q (3 decimals) unrounded."""
    # like upstream, `minimum_nbr_of_qubit_errors` is accepted and unused: the sampler is chosen by
    # env_config["min_qubit_errors"] (evaluation.py:51)
    model.to(device)
    model.eval()
    size = int(env_config["size"])
    if int(grid_shift) != size // 2:
        raise ValueError("grid_shift must be int(size/2)")
    k = len(prediction_list_p_error)
    corrected, ground, steps_avg, mean_q = np.zeros(k), np.zeros(k), np.zeros(k), np.zeros(k)
    failed = []
    for i, p in enumerate(prediction_list_p_error):
        cfg = {"size": size, "min_qubit_errors": int(env_config.get("min_qubit_errors", 0)), "p_error": float(p)}
        envs = EnvSet(ToricEnv(cfg, device=device, seed=seed + i), num_of_episodes, device=device, numpy_io=False)
        envs.resetAll()
        init_q = envs.getQubits().clone()
        r = _greedy_episodes(envs, model, epsilon, num_of_steps, chunk)
        done = r["done"]
        gs = envs.evalGroundState().bool()
        corrected[i] = float(done.double().mean())
        ground[i] = float(gs.double().mean())
        steps_avg[i] = float(r["n_steps"].double().mean())
        mean_q[i] = float(r["q_sum"] / r["q_cnt"].clamp(min=1))
        if round_like_reference:
            steps_avg[i], mean_q[i] = np.round(steps_avg[i], 1), np.round(mean_q[i], 3)
        bad = (~done) | (~gs)
        if bool(bad.any()):
            fq = envs.getQubits()[bad].cpu().numpy()
            for a, b in zip(init_q[bad].cpu().numpy(), fq):
                failed.append(a)
                failed.append(b)
        envs.check()
        envs.close()
    return corrected, ground, steps_avg, mean_q, failed


# ---- the forced-errors sampler of results/small_p_error_test.py (the reference's second recorded accuracy target)
def generateRandomError(matrix, p_error, rng=np.random):
    """results/small_p_error_test.py:22-31: depolarizing errors -- u ~ U(0,1) per qubit, error iff u < p_error,
    Pauli = randint(3) + 1.  ``matrix``: (..., 2, d, d), only its shape is used (as upstream)."""
    u = rng.uniform(0, 1, size=matrix.shape)
    pauli = rng.integers(3, size=matrix.shape) + 1 if hasattr(rng, "integers") else rng.randint(3, size=matrix.shape) + 1
    return ((u < p_error) * pauli).astype(np.int64)


def generateNRandomErrors(matrix, n, rng=np.random):
    """:34-40: exactly n errors on uniformly chosen distinct qubits, Pauli uniform; batched over leading axes."""
    shape = matrix.shape
    nq = 2 * shape[-1] * shape[-1]
    flat = np.zeros((int(np.prod(shape[:-3], dtype=np.int64)), nq), np.int64)
    pauli = rng.integers(3, size=(flat.shape[0], n)) + 1 if hasattr(rng, "integers") else rng.randint(3, size=(flat.shape[0], n)) + 1
    flat[:, :n] = pauli
    # a uniform shuffle of each row: argsort of iid uniforms
    order = np.argsort(rng.uniform(size=flat.shape), axis=1)
    flat = np.take_along_axis(flat, order, axis=1)
    return flat.reshape(shape)


def generateNPlusQRandomErrors(q, p_error, qubit_matrix, rng=np.random):
    """:43-52: q forced errors plus depolarizing noise at p_error on the OTHER qubits.  Batched: qubit_matrix
    (n, 2, d, d) (or (2, d, d)); only its shape is used."""
    forced = generateNRandomErrors(np.zeros(qubit_matrix.shape, np.int64), q, rng)
    noise = generateRandomError(np.zeros(qubit_matrix.shape), p_error, rng)
    noise[forced != 0] = 0
    return forced + noise


def prediction_smart(model, env, env_config, grid_shift, device, prediction_list_p_error, num_of_episodes=1, epsilon=0.0,
                     num_of_steps=50, plot_one_episode=False, show_network=False, show_plot=False, nbr_of_qubit_errors=0,
                     print_Q_values=False, checkpoint=10000, seed=0, chunk=1 << 16, round_like_reference=True):
    """results/small_p_error_test.py:55-196 with the episodes of one p_error side by side on the GPU: every episode
    starts from ``nbr_of_qubit_errors`` forced errors plus depolarizing noise (generateNPlusQRandomErrors), redrawn
    until the syndrome is not empty (:109-120; a stabilizer-shaped draw has none), written into the lattices with
    ``env.qubit_matrix = ...`` (tq_set_qubits), then the greedy loop and evalGroundState.
    -> (error_corrected_list, ground_state_list, average_number_of_steps_list, mean_q_list,
        number_of_failed_syndroms_list, N_fail, P_l_list, failed_syndromes) as upstream."""
    from math import comb
    model.to(device)
    model.eval()
    size = int(env_config["size"])
    if int(grid_shift) != size // 2:
        raise ValueError("grid_shift must be int(size/2)")
    k = len(prediction_list_p_error)
    n_ep, q = int(num_of_episodes), int(nbr_of_qubit_errors)
    max_err = size * size
    ground, corrected, steps_avg, mean_q, P_l_list = (np.zeros(k) for _ in range(5))
    failed = []
    table = np.zeros((3, max_err))
    table[0] = np.arange(max_err)
    N_fail = 0.0
    rng = np.random.default_rng(seed)
    for i, p in enumerate(prediction_list_p_error):
        cfg = {"size": size, "min_qubit_errors": int(env_config.get("min_qubit_errors", 0)), "p_error": float(p)}
        envs = EnvSet(ToricEnv(cfg, device=device, seed=seed + i), n_ep, device=device, numpy_io=False)
        qm = generateNPlusQRandomErrors(q, float(p), np.zeros((n_ep, 2, size, size), np.int64), rng)
        for _ in range(1000):                                   # "while terminal_state": redraw the lattices without a defect
            envs.setQubits(qm.astype(np.uint8))
            empty = envs.isTerminal().bool().cpu().numpy()
            if not empty.any():
                break
            qm[empty] = generateNPlusQRandomErrors(q, float(p), np.zeros((int(empty.sum()), 2, size, size), np.int64), rng)
        else:
            raise RuntimeError("the sampler kept producing empty syndromes")
        flips = (qm != 0).reshape(n_ep, -1).sum(1)
        r = _greedy_episodes(envs, model, epsilon, num_of_steps, chunk)
        done = r["done"].cpu().numpy()
        gs = envs.evalGroundState().bool().cpu().numpy()
        inside = flips < max_err
        np.add.at(table[2], flips[inside & ~gs], 1)
        np.add.at(table[1], flips[inside & gs], 1)
        failed.extend(qm[~gs])
        n_fail = np.array([0.0 if j < q else table[2, j] / comb(j, q) for j in range(max_err)])
        N_fail = float(n_fail.sum())
        nq = 2 * size * size
        P_l_list[i] = comb(nq, q) * float(p) ** q * (1 - float(p)) ** (nq - q) * N_fail / n_ep
        corrected[i] = float(done.mean())
        ground[i] = float(gs.mean())
        steps_avg[i] = float(r["n_steps"].double().mean())
        mean_q[i] = float(r["q_sum"] / r["q_cnt"].clamp(min=1))
        if round_like_reference:
            steps_avg[i], mean_q[i] = np.round(steps_avg[i], 1), np.round(mean_q[i], 3)
        envs.check()
        envs.close()
    return corrected, ground, steps_avg, mean_q, table, N_fail, P_l_list, failed
