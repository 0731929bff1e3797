"""Multi-GPU: independent MPPI problem instances (envs) shard across ranks; ONE collective gathers the chosen controls.

The reference has no counterpart (its only fan-out is SLURM job arrays, others/EulerClusterScripts/
ParallelDataGeneration.sh:2-17).  Envs are fully independent, so the data path needs no exchange: each rank (one
process per GPU, torch.distributed backend "nccl" = RCCL over xGMI) owns a contiguous block of envs, and the updated
nominal control sequences u_nom[E_local,H] (C4: 64 x 50 floats = 12.8 KB per rank) are all-gathered once per step.
The per-env Philox streams are keyed by the GLOBAL env index, so results do not depend on the number of ranks.

Two implementations of that one collective:
  * :class:`NativeGather` — the production path: ``cpmppi_comm_gather`` (csrc/cpmppi_comm.hip) enqueues ``ncclAllGather``
    from C on a high-priority side stream under the NEXT step's rollout kernel, straight from one of the two nominal-
    sequence buffers of ``cpmppi_step_args.u_nom_out`` (no snapshot copy, no torch ``Work`` object per step);
  * :func:`gather_controls` — ``torch.distributed`` (gloo on CPU: what the world_size-2 tests run; also the fallback
    when RCCL cannot be bound).
"""
import ctypes as C

import torch
import torch.distributed as dist


def env_shard(E_total, world_size, rank):
    """Contiguous block split of the env axis -> (start, count); the first E_total % world_size ranks get one more."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(int(E_total), int(world_size))
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def gather_controls(local, E_total, group=None):
    """All-gather per-rank blocks [E_local, ...] (block split of env_shard) into [E_total, ...] on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [env_shard(E_total, world, r)[1] for r in range(world)]
    if local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} envs, expected {counts[rank]}")
    cmax = max(counts)
    tail = tuple(local.shape[1:])
    if min(counts) == cmax:
        out = torch.empty((world * cmax,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros((cmax,) + tail, dtype=local.dtype, device=local.device)
    pad[:counts[rank]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


class ShardedMPPI:
    """Steps this rank's block of envs with ``step_fn`` and gathers the controls of all envs.

    ``step_fn(s_local[E_local,6], env_offset) -> (u_nom_local[E_local,H], Q_local[E_local])`` — in production the fused
    HIP step of an MPPIEngine (see :func:`hip_step_fn`); tests inject a checker-backed function to exercise the
    sharding and the collective on CPU (gloo).
    """

    def __init__(self, E_total, step_fn, group=None):
        self.E_total = int(E_total)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.start, self.count = env_shard(self.E_total, self.world, self.rank)
        self.step_fn = step_fn

    def local(self, x):
        """The rows of a global per-env array this rank owns."""
        return x[self.start:self.start + self.count]

    def step(self, s_local):
        """-> (u_nom[E_total,H], Q[E_total]) on every rank.  ONE collective: the control to apply is the first element of
        the gathered nominal sequence (Q = u_nom[:,0], controller_mppi_cartpole.py:563), so only u_nom travels."""
        u_local, _ = self.step_fn(s_local, self.start)
        u_all = gather_controls(u_local, self.E_total, self.group)
        return u_all, u_all[:, 0].clone()


def hip_step_fn(engine, u_nom, target_position, target_equilibrium, L, seed):
    """step_fn for ShardedMPPI on the HIP path: in-kernel Philox noise keyed by the global env index."""
    state = {"offset": 0}

    def fn(s_local, env_offset):
        Q, _ = engine.step(s_local, u_nom, target_position, target_equilibrium, L=L, seed=seed,
                           offset=state["offset"], env_offset=env_offset)
        state["offset"] += 1
        return u_nom, Q

    return fn


def exchange_unique_id(lib, rank, key="cpmppi_comm_id", rccl_path=None):
    """Rank 0 draws the RCCL unique id (cpmppi_comm_unique_id), every rank receives it through the process group's
    key-value store (no collective, no device tensors: works under any torch.distributed backend)."""
    from . import _lib as _L
    store = dist.distributed_c10d._get_default_store()
    if rank == 0:
        buf = C.create_string_buffer(_L.COMM_ID_BYTES)
        rc = lib.cpmppi_comm_unique_id(buf, rccl_path)
        if rc != 0:
            store.set(key, b"!" + lib.cpmppi_last_error(None))
            raise _L.CpmppiError(rc, lib.cpmppi_last_error(None).decode())
        store.set(key, b"+" + buf.raw)
        return buf.raw
    got = store.get(key)
    if got[:1] != b"+":
        raise RuntimeError("rank 0 could not create the RCCL id: " + got[1:].decode(errors="replace"))
    return got[1:]


def block_stamps(gathered, count):
    """[world] int64: the stamp behind every rank's block of a stamped gather ``gathered[world, count + GATHER_STAMP_FLOATS]``
    (cpmppi_comm_set_stamped, include/cpmppi.h): the number of the step-gather that produced the block."""
    return gathered[:, count:count + 1].contiguous().view(torch.int32).to(torch.int64).view(-1)


def accepted_blocks(gathered, count, number):
    """[world] bool: which blocks of the ``number``-th step-gather (1, 2, ...) a receiver may use - those stamped with exactly that
    number.  A rank that dropped the step (its device-side wait for an earlier all-gather timed out) sent its buffer as it was,
    old stamp included: that block is stale and the receiver keeps what it had for that rank."""
    return block_stamps(gathered, count) == int(number)


def merge_accepted(previous, gathered, count, number):
    """The receiver's rule in one place: -> (sequences[world, count], accepted[world]) where rejected ranks keep ``previous``'s rows."""
    ok = accepted_blocks(gathered, count, number)
    return torch.where(ok[:, None], gathered[:, :count], previous), ok


class NativeGather:
    """The per-step all-gather of u_nom[E_local,H] through libcpmppi's own RCCL communicator.

    ``u[0]``, ``u[1]``: the two nominal-sequence buffers (step i reads ``u[i & 1]`` and writes ``u[(i + 1) & 1]`` via
    ``u_nom_out``); ``gathered[b]`` [world, E_local*H] receives the gather of ``u[b]``.  Per step:
    ``before_step(i)`` (device-side wait for the gather that still reads the buffer this step overwrites), the step,
    ``after_step(i)`` (enqueue the gather of the buffer just written).  Nothing here blocks the host."""

    def __init__(self, engine, unique_id, world, rank, rccl_path=None, stamped=False):
        """``stamped``: every gathered block carries its step number behind the sequences (cpmppi_comm_set_stamped, cpmppi.h): a
        receiver can tell a block its sender DROPPED (a device-side wait that timed out leaves the old stamp) from a fresh one -
        :meth:`accepted`.  ``rccl_path``: a collective library to bind instead of the process's RCCL (bytes or str)."""
        from . import _lib as _L
        self.engine, self.world, self.rank, self.stamped = engine, int(world), int(rank), bool(stamped)
        e = engine
        if isinstance(rccl_path, str):
            rccl_path = rccl_path.encode()
        e._check(e.lib.cpmppi_comm_init(e._h, unique_id, self.world, self.rank, rccl_path))
        n = e.E * e.H
        pad = _L.GATHER_STAMP_FLOATS if self.stamped else 0
        if self.stamped:
            e._check(e.lib.cpmppi_comm_set_stamped(e._h, 1))
        self.count = n
        self._flat = [torch.zeros(n + pad, dtype=torch.float32, device=e.device) for _ in range(2)]     # (stamp words zeroed once)
        self.u = [f[:n].view(e.E, e.H) for f in self._flat]
        self.gathered = [torch.zeros(self.world, n + pad, dtype=torch.float32, device=e.device) for _ in range(2)]

    def blocks(self, i):
        """[world, E_local*H]: the sequences of every rank as gathered after step i (without the stamp words)."""
        return self.recv(i)[:, :self.count]

    def stamps(self, i):
        """[world] int64: the stamp every rank's block of step i's gather carries (stamped communicators)."""
        if not self.stamped:
            raise ValueError("this communicator is not stamped")
        return block_stamps(self.recv(i), self.count)

    def accepted(self, i, gather_number=None):
        """[world] bool: block r of step i's gather was written by its sender's step number ``gather_number`` (default i + 1: one
        cpmppi_step_gather per step since the communicator was made) and is complete; a False entry = a stale block to be ignored."""
        return self.stamps(i) == int(i + 1 if gather_number is None else gather_number)

    def before_step(self, i):
        e = self.engine
        e._check(e.lib.cpmppi_comm_wait(e._h, (i + 1) & 1, e._stream()))       # gather i-2 read the buffer step i writes

    def after_step(self, i):
        e = self.engine
        b = (i + 1) & 1
        e._check(e.lib.cpmppi_comm_gather(e._h, b, self.u[b].data_ptr(), self.gathered[b].data_ptr(), e.E * e.H, e._stream()))

    def recv(self, i):
        """The buffer the gather of step i's result goes to."""
        return self.gathered[(i + 1) & 1]

    def u_in(self, i):
        return self.u[i & 1]

    def u_out(self, i):
        return self.u[(i + 1) & 1]

    def sync(self):
        e = self.engine
        e._check(e.lib.cpmppi_comm_sync(e._h))

    def info(self):
        """cpmppi_comm_get_info: what RCCL itself says the communicator is (ranks, this rank, version) next to what it was told."""
        import ctypes as C
        from . import _lib as _L
        e = self.engine
        out = _L.cpmppi_comm_info()
        e._check(e.lib.cpmppi_comm_get_info(e._h, C.byref(out)))
        return {n: int(getattr(out, n)) for n, _ in out._fields_}

    def close(self):
        e = self.engine
        if getattr(e, "_h", None) and e._h.value:
            e.lib.cpmppi_comm_destroy(e._h)
