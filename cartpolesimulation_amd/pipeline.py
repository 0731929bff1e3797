"""Env groups on their own streams: independent MPPI problem instances need not march in step.

A launch of a FEW envs (BASELINE configs[2] / configs[3]: 64 envs per GPU = one or two waves per SIMD) ends with its slowest wave:
the waves of a launch differ by 20-40 % in run time (rollouts that meet the track edge or spin fast take the eventful path;
profiles/r5/placement.txt: C3 median 160 us, slowest 225 us, every SIMD evenly loaded), and in a chain step -> plant -> step ... the
next launch cannot start before the last wave of the previous one has left, although only the envs of THAT wave's block wait for
it.  Envs are independent (the reference's only fan-out is share-nothing job arrays, others/EulerClusterScripts/
ParallelDataGeneration.sh:2-17), so here the E envs of a device are split into G contiguous groups, each with its own handle
(a handle serves one stream) and its own HIP stream, and every group runs its OWN chain: group A's step i + 1 fills the SIMDs group
B's slow waves leave idle.  No group ever waits for another; Philox keys use the GLOBAL env index (`env_offset`), so results do not
depend on the split (bit-identical to the unsplit launch when both pick the same lane mapping; within the parity band otherwise).
Measured (profiles/r5/multistream_*.txt, wall time per step of all envs): C4 73.8 -> 62.4 us with 2 groups, C3 230.7 -> 175.4 us,
256 envs x 2048 x 50 173.3 -> 138.7 us with 4.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as _L
from .configs import MPPIConfig, PhysicalParameters
from .engine import MPPIEngine


def split_envs(E, groups):
    """Contiguous, as even as possible: [(start, stop)] * groups (the first E % groups groups get one env more)."""
    groups = max(1, min(int(groups), int(E)))
    base, extra = divmod(int(E), groups)
    out, e0 = [], 0
    for g in range(groups):
        e1 = e0 + base + (1 if g < extra else 0)
        out.append((e0, e1))
        e0 = e1
    return out


class EnvGroups:
    """G engines + G streams over one device's E envs.  Tensors handed to `prepare_step` are the FULL [E, ...] arrays; every group
    works on its contiguous slice of them in place."""

    def __init__(self, E, mppi: MPPIConfig = None, groups=2, phys: PhysicalParameters = None, device=0, env_offset=0):
        self.E, self.env_offset = int(E), int(env_offset)
        self.slices = split_envs(E, groups)
        self.engines, self.streams = [], []
        self._raw = []
        for (e0, e1) in self.slices:
            eng = MPPIEngine(e1 - e0, mppi, phys, device)
            # a stream with a hardware queue of its own (cpmppi_stream_create): pooled streams may share a queue and then serialise
            raw = C.c_void_p()
            rc = eng.lib.cpmppi_stream_create(eng.device.index, C.byref(raw))
            if rc != 0:
                raise _L.CpmppiError(rc, eng.lib.cpmppi_last_error(None).decode())
            self._raw.append(raw)
            st = torch.cuda.ExternalStream(raw.value, device=eng.device)
            eng.use_stream(st)
            self.engines.append(eng)
            self.streams.append(st)
        self.device = self.engines[0].device
        self.H, self.N = self.engines[0].H, self.engines[0].N

    def __len__(self):
        return len(self.engines)

    def fork(self):
        """Every group stream waits for what has been enqueued on the caller's current stream (uploads, allocations)."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            st.wait_stream(cur)

    def join(self):
        """The caller's current stream waits for every group (before results are read or freed on it)."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)

    def prepare_step(self, s0, u_nom, target_position, target_equilibrium, L=None, seed=0, Q_out=None, S_out=None, **kw):
        """-> one PreparedStep per group over the slices of the given [E, ...] device tensors (in-kernel Philox noise keyed by the
        global env index)."""
        E = self.E
        for name, t in (("s0", s0), ("u_nom", u_nom), ("target_position", target_position), ("target_equilibrium", target_equilibrium)):
            if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape[0] == E):
                raise ValueError(f"{name} must be a contiguous float32 ROCm tensor with {E} rows")
        if Q_out is None:
            Q_out = torch.empty(E, dtype=torch.float32, device=self.device)
        preps = []
        for eng, (e0, e1) in zip(self.engines, self.slices):
            preps.append(eng.prepare_step(s0[e0:e1], u_nom[e0:e1], target_position[e0:e1], target_equilibrium[e0:e1],
                                          L=None if L is None else L[e0:e1], seed=seed, offset=0, env_offset=self.env_offset + e0,
                                          Q_out=Q_out[e0:e1], S_out=None if S_out is None else S_out[e0:e1], **kw))
        return preps

    def overlap(self, preps, steps=20):
        """Do the groups really run side by side?  `steps` steps of every group one group after the other, then all together:
        -> sum of the groups' times alone / time together (1.0 = serialised on one hardware queue, ~G = perfect overlap)."""
        import time
        torch.cuda.synchronize(self.device)

        def timed(which):
            t0 = time.perf_counter()
            for i in range(steps):
                for p in which:
                    p.run(offset=10_000 + i)
            torch.cuda.synchronize(self.device)
            return time.perf_counter() - t0

        timed(preps)                                                   # warm
        alone = sum(timed([p]) for p in preps)
        return alone / timed(preps)

    def close(self):
        for eng in self.engines:
            eng.close()
        if self.engines:
            torch.cuda.synchronize(self.device)
            lib = _L.load()
            for raw in self._raw:
                lib.cpmppi_stream_destroy(raw)
        self.engines, self._raw, self.streams = [], [], []


def run_schedule_groups(groups: EnvGroups, batch, seed, graph=False, steps_per_graph=10):
    """harness.BatchedCartPoleExperiment.run_schedule with the experiments split over the env groups: every group runs its own
    chain of (controller step, plant step) launches on its own stream, the host interleaves the enqueueing.  -> the same result
    dict (device tensors concatenated along the env axis)."""
    import dataclasses
    from .harness import ScheduleRun
    runs = []
    for eng, (e0, e1) in zip(groups.engines, groups.slices):
        sub = dataclasses.replace(batch, s0=batch.s0[e0:e1], target_position=np.ascontiguousarray(batch.target_position[:, e0:e1]),
                                  target_equilibrium=np.ascontiguousarray(batch.target_equilibrium[:, e0:e1]),
                                  interpolation_type=batch.interpolation_type[e0:e1], L=None if batch.L is None else batch.L[e0:e1],
                                  L_table=None if batch.L_table is None else np.ascontiguousarray(batch.L_table[:, e0:e1]))
        runs.append(ScheduleRun(eng, sub, seed, env_offset=groups.env_offset + e0))
    groups.fork()                                              # (the runs' buffers were set up on the caller's stream)
    if graph:
        for r in runs:
            if r.T > 0:
                r.capture(steps_per_graph)
    while any(r.periods_left for r in runs):
        for r in runs:
            if r.periods_left:
                r.enqueue_next()
    outs = [r.finish() for r in runs]
    groups.join()
    cat = lambda k, dim: torch.cat([o[k] for o in outs], dim=dim)   # noqa: E731
    return dict(states=cat("states", 1), dd=cat("dd", 1), Q=cat("Q", 1), final_state=cat("final_state", 0), u_nom=cat("u_nom", 0),
                batch=batch)
