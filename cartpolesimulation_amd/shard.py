"""Multi-GPU: independent MPPI problem instances (envs) shard across ranks; ONE collective gathers the chosen controls.

The reference has no counterpart (its only fan-out is SLURM job arrays, others/EulerClusterScripts/
ParallelDataGeneration.sh:2-17).  Envs are fully independent, so the data path needs no exchange: each rank (one
process per GPU, torch.distributed backend "nccl" = RCCL over xGMI) owns a contiguous block of envs, and the updated
nominal control sequences u_nom[E_local,H] (C4: 64 x 50 floats = 12.8 KB per rank) are all-gathered once per step.
The per-env Philox streams are keyed by the GLOBAL env index, so results do not depend on the number of ranks.
"""
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
        u_local, q_local = self.step_fn(s_local, self.start)
        return gather_controls(u_local, self.E_total, self.group), gather_controls(q_local, self.E_total, self.group)


def hip_step_fn(engine, u_nom, target_position, target_equilibrium, L, seed):
    """step_fn for ShardedMPPI on the HIP path: in-kernel Philox noise keyed by the global env index."""
    state = {"offset": 0}

    def fn(s_local, env_offset):
        Q, _ = engine.step(s_local, u_nom, target_position, target_equilibrium, L=L, seed=seed,
                           offset=state["offset"], env_offset=env_offset)
        state["offset"] += 1
        return u_nom, Q

    return fn
