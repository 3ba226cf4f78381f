"""Device-resident counterpart of the reference's actor loop (src/Actor_mp.py:104-185).

Same call order, nothing leaves the GPU inside the loop:
    selectActionBatch -> envs.step -> generateTransitionParallel -> local buffers (T, A, Q, R)
    -> every `size_local_memory_buffer + 1` steps: computePrioritiesParallel, send (T, priority)
    -> reset of terminal / timed-out lattices with the p_error schedule.
The last three env steps are one fused kernel (EnvSet.actorStep); the send is ONE packed transition
block that carries the f32 priority of every transition (handed to `sink`, e.g.
gather.TransitionGather on N>1) -- the (transition, priority) pairs of Actor_mp.py:152.
"""
import numpy as np
import torch

from .policy import selectActionEnvSet


def computePrioritiesParallel(A, R, Q, Qns, discount):
    """Drop-in for util_actor.py:268-287: |R + discount * max_a Qns - Q[a]|.
    A (N,T,4) actions, R (N,T) rewards, Q / Qns (N,T,3) q-values of the state / the next state.
    numpy arrays in -> float64 numpy array out, exactly the reference's arithmetic (the actor's
    host-side buffers, Actor_mp.py:146-150); device tensors in -> a tensor of R's dtype on the device."""
    if not torch.is_tensor(Q):
        A, R, Q, Qns = np.asarray(A), np.asarray(R), np.asarray(Q), np.asarray(Qns)
        q_taken = np.take_along_axis(Q, (A[:, :, -1].astype(np.int64) - 1)[:, :, None], axis=2)[:, :, 0]
        return np.absolute(R + discount * np.amax(Qns, axis=2) - q_taken)
    q_taken = torch.gather(Q, 2, (A[..., 3].long() - 1).clamp(min=0).unsqueeze(-1)).squeeze(-1)
    return (R + discount * Qns.max(dim=2).values - q_taken).abs()


def run_actor(envs, model, n_flushes, size_local_memory_buffer, epsilon, discount_factor=0.95, sink=None,
              chunk=1 << 16, weight_sync=None):
    """Runs `n_flushes` buffer flushes of the actor loop on `envs` (an EnvSet with numpy_io=False,
    already reset).  Yields one TransitionBlock per flush: no_envs * size_local_memory_buffer
    transitions in slot order t * no_envs + e with their priorities in the block's priority section
    (tq_block_priorities = computePrioritiesParallel on the device, f64 arithmetic stored as f32).
    As upstream, the local buffer has one extra column whose transition is dropped at the flush
    (local_buffer_T[:, :-1], Actor_mp.py:67,146-152): that step is taken, its q_values close the
    priorities of the column before it, and its transition is not recorded.
    ``weight_sync``: called with the model when the buffer is full, before the priorities and the send --
    where upstream loads the learner's new weights (Actor_mp.py:133-144); on N>1 pass
    ``lambda m: gather.broadcast_weights(m, src=learner_rank)``."""
    assert not envs.numpy_io, "run_actor needs an EnvSet with numpy_io=False"
    n, dev = envs.no_envs, envs.device
    T = int(size_local_memory_buffer)
    blocks = [envs.newTransitionBlock(steps=T) for _ in range(2)]
    Q = torch.zeros((T + 1, n, 3), dtype=torch.float32, device=dev)       # step-major: Q[t] = q_values of step t
    for f in range(n_flushes):
        blk = blocks[f & 1]
        for t in range(T + 1):
            act, qv = selectActionEnvSet(envs, model, epsilon, chunk=chunk)
            Q[t] = qv
            envs.actorStep(act, block=blk if t < T else None, slot=t)
        if weight_sync is not None:
            weight_sync(model)
        blk.computePriorities(n, T, Q, discount_factor)
        if sink is not None:
            sink(blk)
        yield blk
