"""Device-resident counterpart of the reference's actor loop (src/Actor_mp.py:104-185).

Same call order, nothing leaves the GPU inside the loop:
    selectActionBatch -> envs.step -> generateTransitionParallel -> local buffers (T, A, Q, R)
    -> every `size_local_memory_buffer + 1` steps: computePrioritiesParallel, send (T, priority)
    -> reset of terminal / timed-out lattices with the p_error schedule.
The last three env steps are one fused kernel (EnvSet.actorStep); the send is a packed transition
block plus an f32 priority vector (handed to `sink`, e.g. gather.TransitionGather on N>1).
"""
import torch

from .policy import selectActionBatch


def computePrioritiesParallel(A, R, Q, Qns, discount):
    """util_actor.py:268-287 on device tensors: |R + discount * max_a Qns - Q[a]|.
    A (N,T,4) actions, R (N,T) rewards, Q / Qns (N,T,3) q-values of the state / the next state."""
    q_taken = torch.gather(Q, 2, (A[..., 3].long() - 1).clamp(min=0).unsqueeze(-1)).squeeze(-1)
    return (R + discount * Qns.max(dim=2).values - q_taken).abs()


def run_actor(envs, model, n_flushes, size_local_memory_buffer, epsilon, discount_factor=0.95, sink=None,
              chunk=1 << 16):
    """Runs `n_flushes` buffer flushes of the actor loop on `envs` (an EnvSet with numpy_io=False,
    already reset).  Yields (block, priorities) per flush: block = TransitionBlock holding
    no_envs * size_local_memory_buffer transitions in slot order t * no_envs + e, priorities f32
    (no_envs, size_local_memory_buffer) -- the (transition, priority) pairs of Actor_mp.py:152.
    As upstream, the buffer has one extra column whose transition is dropped at the flush
    (local_buffer_T[:, :-1], Actor_mp.py:67,146-152)."""
    assert not envs.numpy_io, "run_actor needs an EnvSet with numpy_io=False"
    n, dev = envs.no_envs, envs.device
    T = int(size_local_memory_buffer) + 1
    blocks = [envs.newTransitionBlock(steps=T) for _ in range(2)]
    A = torch.zeros((n, T, 4), dtype=torch.int32, device=dev)
    Q = torch.zeros((n, T, 3), dtype=torch.float32, device=dev)
    R = torch.zeros((n, T), dtype=torch.float32, device=dev)
    for f in range(n_flushes):
        blk = blocks[f & 1]
        for t in range(T):
            act, qv = selectActionBatch(envs, model, epsilon, chunk=chunk)
            A[:, t] = act
            Q[:, t] = qv
            _, rew, _ = envs.actorStep(act, block=blk, slot=t)
            R[:, t] = rew
        prio = computePrioritiesParallel(A[:, :-1], R[:, :-1], Q[:, :-1], torch.roll(Q, -1, dims=1)[:, :-1],
                                         discount_factor)
        if sink is not None:
            sink(blk, prio)
        yield blk, prio
