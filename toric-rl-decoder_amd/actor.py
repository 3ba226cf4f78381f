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


class ExploreLoop:
    """The actor loop body (Actor_mp.py:104-183) at epsilon = 1 -- where upstream starts (Actor_mp.py:37) and where the
    selection never looks at a Q-value -- with the env kernels BESIDE the stack write instead of behind it:

        stream A (the caller's current stream):  ... write(t) ............................ write(t+1) ...
        stream B (a side stream):                      step(t) -> [priorities] -> scan(t+1)

    write(t) = tq_persp_write of the pre-step lattices into the caller's stack / positions buffers (what the policy
    network would read); step(t) = tq_actor_step with in-kernel selection (step, transition record into ``blocks``,
    auto-reset, counts); scan(t+1) = tq_persp_count into the other of two offsets rows.  The handle keeps two plane
    buffers and two cut-point tables in turn (include/toricenv.h, tq_actor_step), so the only ordering left to the
    caller is: write(t) behind scan(t); step(t+1) behind write(t) -- two events per step.  With a Q-table in the loop
    step(t) depends on the stack and the path is serial (run_actor).  ``overlap=False``: everything on stream A, in
    the reference's order.

    ``offsets``: (R >= 2, >= no_envs + 1 rounded up to even) int64 tensor; step t scans into row t % R (P of step t stays
    readable at offsets[t % R, no_envs] until the row is re-used).  ``chunks`` > 1: the stack is written in that many
    lattice ranges, one after the other, into a buffer of 1/chunks the size (tq_persp_write_range).
    ``on_flush(block)``: called (on stream B) when a block of ``flush`` steps is complete and its priorities are in
    -- e.g. gather.TransitionGather.gather.
    ``pace``: how write(t) is ordered behind scan(t): "host" (default) -- the host waits for scan(t)'s event before it
    enqueues write(t), so the host runs at most one step ahead of the GPU and stream A carries no barrier; "device" -- a
    stream-side wait (hipStreamWaitEvent), the host runs ahead freely, 6-8 us of stream time per step."""

    def __init__(self, envs, stack, positions, offsets, blocks=None, flush=8, chunks=1, overlap=True, on_flush=None, pace="host"):
        assert not envs.numpy_io, "ExploreLoop needs an EnvSet with numpy_io=False"
        n = envs.no_envs
        if offsets.dtype != torch.int64 or offsets.dim() != 2 or offsets.shape[0] < 2 or offsets.shape[1] < n + 1 or offsets.shape[1] % 2:
            raise ValueError("offsets must be an int64 tensor (rows >= 2, even row length >= no_envs + 1)")
        if n % int(chunks):
            raise ValueError("no_envs must be divisible by chunks")
        self.envs, self.stack, self.positions, self.offsets = envs, stack, positions, offsets
        self.blocks, self.flush, self.chunks, self.on_flush = blocks, int(flush), int(chunks), on_flush
        if pace not in ("host", "device"):
            raise ValueError("pace must be 'host' or 'device'")
        self.pace = pace
        self.dev = envs.device
        self.A = torch.cuda.current_stream(self.dev)
        self.B = torch.cuda.Stream(device=self.dev) if overlap else self.A
        self.overlap = self.B is not self.A
        self.scanned = [torch.cuda.Event() for _ in range(2)]
        self.written = [torch.cuda.Event() for _ in range(2)]
        self.t = 0
        if self.overlap:
            self.B.wait_stream(self.A)                           # whatever set the lattices up
        with torch.cuda.stream(self.B):
            envs.perspectiveCounts(self._row(0))
            if self.overlap:
                self.scanned[0].record(self.B)

    def _row(self, t):
        return self.offsets[t % self.offsets.shape[0]][:self.envs.no_envs + 1]

    def step(self, bracket=None):
        """One pass.  ``bracket``: a pair of timing events recorded on stream A right before and after the stack write."""
        envs, t, k = self.envs, self.t, self.t & 1
        off = self._row(t)
        if self.overlap:
            # write(t) behind scan(t).  scan(t) finished long ago -- it ran beside write(t-1) -- but a device-side wait
            # (a barrier packet in front of the write) still costs stream A 6-8 us per step on MI355X
            # (profiles/r04_overlap_cost.txt); the HOST waiting for the event costs stream A nothing: write(t-1) is
            # still running for another ~0.25 ms when the event completes, ample time to enqueue write(t) behind it.
            if self.pace == "host":
                self.scanned[k].synchronize()
            else:
                self.A.wait_event(self.scanned[k])
        if bracket is not None:
            bracket[0].record(self.A)
        if self.chunks == 1:
            envs.writePerspectives(self.stack, self.positions, off)
        else:
            per = envs.no_envs // self.chunks
            for c in range(self.chunks):                          # a consumer would read the buffer between two ranges
                envs.writePerspectives(self.stack, self.positions, off, first=c * per, count=per)
        if bracket is not None:
            bracket[1].record(self.A)
        if self.overlap:
            self.written[k].record(self.A)
        with torch.cuda.stream(self.B):
            if self.overlap and t > 0:
                self.B.wait_event(self.written[k ^ 1])            # write(t-1) read the plane buffer this step writes
            blk = self.blocks[(t // self.flush) % len(self.blocks)] if self.blocks else None
            envs.actorStep(None, block=blk, slot=t % self.flush, want_actions=True)
            if blk is not None and (t + 1) % self.flush == 0:
                blk.computePriorities(envs.no_envs, self.flush, None, 0.95)     # eps = 1: no Q-values, priority = |reward|
                if self.on_flush is not None:
                    self.on_flush(blk)
            envs.perspectiveCounts(self._row(t + 1))
            if self.overlap:
                self.scanned[k ^ 1].record(self.B)
        self.t = t + 1

    def time_writes(self, stack, steps, skip=1):
        """Run ``skip`` + ``steps`` passes with the stack written into ``stack`` and -> the milliseconds of each of the
        last ``steps`` stack writes (HIP events on stream A around the write, the env kernels running beside it as in
        the loop).  Synchronises; for the set-up probe (EnvSet.pickStackBuffer(timer=...))."""
        keep = self.stack
        self.stack = stack
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(int(steps))]
        for _ in range(int(skip)):
            self.step()
        for e in evs:
            self.step(e)
        self.drain()
        torch.cuda.current_stream(self.dev).synchronize()
        self.stack = keep
        return [a.elapsed_time(b) for a, b in evs]

    def drain(self):
        """Order stream A behind everything enqueued on stream B so far (no host synchronisation)."""
        if self.overlap:
            self.A.wait_stream(self.B)


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
