"""Transition exchange between env shards: one process per GPU, RCCL gather over xGMI.

The reference's actors push pickled lists of (transition, priority) through an mp.Queue to the
replay process (Actor_mp.py:152-169, IO_mp.py:60-66; MPI variant: object gather,
mpi/Actor_mpi.py:134-150).  Here every rank owns a contiguous block of env ids and contributes
one fixed-size packed transition block per flush; rank 0 receives all of them into a replay
ring that stays in its HBM (288 GB: ~7e9 transitions at d=7), so nothing crosses PCIe per step.
Lattices never interact, so this gather is the ONLY collective of the path.

The other direction is the weights: the learner publishes the flattened parameter vector
(Learner_mp.py:124-130, ~0.9 M f32 = 3.6 MB for NN_11) and every actor loads it at its next flush
(Actor_mp.py:133-144; MPI variant: comm.bcast, mpi/Actor_mpi.py:84) -> :func:`broadcast_weights`,
ONE broadcast per flush.

Backend "nccl" is RCCL on ROCm (device buffers, xGMI point-to-point fan-in to the root);
"gloo" (CPU tests, world_size 2) stages the same bytes through host tensors.
"""
import torch
import torch.distributed as dist
from torch.nn.utils import parameters_to_vector, vector_to_parameters


def broadcast_weights(model, src=0, group=None, buffer=None):
    """Learner -> actors, once per flush: rank ``src`` flattens its parameters (parameters_to_vector,
    Learner_mp.py:124), ONE ``dist.broadcast`` of the f32 vector (RCCL over xGMI; gloo on the CPU), every other
    rank loads it with vector_to_parameters (Actor_mp.py:143).  The reference's shared array is float64 and the
    actor casts back to FloatTensor: f32 -> f64 -> f32 is the identity, so sending f32 is bit-identical.
    ``buffer``: optional reusable f32 staging vector (returned; allocated on first use).  Enqueues on the
    current stream for nccl (no host synchronisation); parameters only, like upstream (no module buffers)."""
    params = list(model.parameters())
    if not params:
        return buffer
    backend = dist.get_backend(group)
    dev = params[0].device
    stage = dev if backend == "nccl" else torch.device("cpu")
    n = sum(p.numel() for p in params)
    if buffer is None or buffer.numel() != n or buffer.device != stage or buffer.dtype != torch.float32:
        buffer = torch.empty(n, dtype=torch.float32, device=stage)
    with torch.no_grad():
        if dist.get_rank(group) == src:
            buffer.copy_(parameters_to_vector(params).detach().to(torch.float32))
        dist.broadcast(buffer, src=src, group=group)
        if dist.get_rank(group) != src:
            vector_to_parameters(buffer.to(device=dev, dtype=params[0].dtype), params)
    return buffer


def shard_range(total_envs, world_size, rank):
    """Contiguous block of global env ids of `rank`: [first, first+count)."""
    base, rem = divmod(int(total_envs), int(world_size))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


class TransitionGather:
    """Gathers equal-size byte blocks from every rank into rings of `ring_slots` slots on the root rank(s).

    ``gather(buf)`` enqueues the exchange of this flush's block (async on the collective's own
    stream for nccl, so it overlaps the next env steps) and returns the ring slot it lands in;
    ``wait()`` blocks until every outstanding exchange has completed.

    The caller alternates between ``source_buffers`` blocks (default 2).
    ``roots`` > 1: flush i is gathered to rank ``i % roots`` -- every root keeps its own ring and, with
    ``host_drain``, drains it to its own pinned host ring over its own PCIe link (one Gen5 x16 link
    carries ~55 GB/s; eight ranks of BASELINE configs[4] produce ~63 GB/s of packed records).  The
    host replay buffer is then the union of the roots' pinned rings.  ``last_root`` names the root of
    the most recent gather.
    """

    def __init__(self, block_nbytes, device, ring_slots=2, group=None, host_drain=False, roots=1, source_buffers=2):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.nbytes = int(block_nbytes)
        self.device = torch.device(device)
        self.stage_device = self.device if self.backend == "nccl" else torch.device("cpu")
        self.ring_slots = int(ring_slots)
        self.roots = max(1, min(int(roots), self.world))
        self.is_root = self.rank < self.roots
        self.ring = None
        if self.is_root:
            self.ring = torch.zeros((self.ring_slots, self.world, self.nbytes), dtype=torch.uint8,
                                    device=self.stage_device)
        # optional drain to the host replay process: every gathered slot is copied to a pinned host
        # ring on a dedicated copy stream (D2H overlaps the env kernels; a slot is only re-used once
        # its copy has finished, so PCIe back-pressures the actors only if it cannot keep up)
        self.host_ring = None
        self._copy_stream = None
        self._drained = []
        # The collective is issued from a stream of its own (nccl): it waits for the env stream (the block is
        # complete) and for the ring slot's previous D2H copy, so a slow PCIe drain holds back the next gather into
        # that slot -- never the env kernels on the compute stream.
        self._coll_stream = torch.cuda.Stream(device=self.device) if self.stage_device.type == "cuda" else None
        self._ready = torch.cuda.Event() if self._coll_stream is not None else None
        if host_drain and self.is_root and self.stage_device.type == "cuda":
            self.host_ring = torch.empty((self.ring_slots, self.world, self.nbytes), dtype=torch.uint8).pin_memory()
            self._copy_stream = torch.cuda.Stream(device=self.device)
            self._drained = [torch.cuda.Event() for _ in range(self.ring_slots)]
        self._next = 0
        self._pending = []
        self.last_root = 0
        # exchanges that may be in flight after gather() returns: the caller rotates `source_buffers` blocks
        # (one is being filled while the others are being sent), and a ring slot of a root is re-used
        # ring_slots * roots flushes later
        self._max_pending = max(0, min(int(source_buffers) - 1, self.ring_slots * self.roots - 1))

    def gather(self, buf):
        if buf.dtype != torch.uint8 or buf.numel() != self.nbytes:
            raise ValueError("block must be a uint8 tensor of block_nbytes elements")
        root = self._next % self.roots
        use = self._next // self.roots                      # how many flushes this root has received before
        slot = use % self.ring_slots
        self._next += 1
        self.last_root = root
        mine = self.rank == root
        src = buf if buf.device == self.stage_device else buf.to(self.stage_device)
        outs = [self.ring[slot, r] for r in range(self.world)] if mine else None
        if self._coll_stream is not None:
            self._ready.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._coll_stream):
                self._coll_stream.wait_event(self._ready)
                if mine and self.host_ring is not None and use >= self.ring_slots:
                    self._coll_stream.wait_event(self._drained[slot])             # slot still being copied out?
                work = dist.gather(src, gather_list=outs, dst=root, group=self.group, async_op=True)
        else:
            work = dist.gather(src, gather_list=outs, dst=root, group=self.group, async_op=True)
        self._pending.append((work, src))
        if mine and self.host_ring is not None:
            with torch.cuda.stream(self._copy_stream):
                work.wait()                                     # copy stream waits for the collective, not the host
                self.host_ring[slot].copy_(self.ring[slot], non_blocking=True)
                self._drained[slot].record(self._copy_stream)
        self._drain(self._max_pending)                      # never overwrite a slot or a source block still in flight
        return slot

    def _drain(self, keep):
        while len(self._pending) > keep:
            work, _ = self._pending.pop(0)
            work.wait()

    def wait(self):
        self._drain(0)
        if self._coll_stream is not None:
            self._coll_stream.synchronize()
        if self._copy_stream is not None:
            self._copy_stream.synchronize()
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    def slot_view(self, slot, rank, host=False):
        """Root only: the bytes rank `rank` contributed to ring slot `slot` (device ring, or the pinned
        host copy when the gather was created with host_drain=True)."""
        return (self.host_ring if host else self.ring)[slot, rank]
