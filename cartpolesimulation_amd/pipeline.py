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
    """The library's env groups (cpmppi_groups_*: G handles + G dedicated-queue streams over one device's E envs) with an
    MPPIEngine view of every group.  Tensors handed to `prepare` / `prepare_step` are the FULL [E, ...] arrays; every group works on
    its contiguous slice of them in place."""

    def __init__(self, E, mppi: MPPIConfig = None, groups=2, phys: PhysicalParameters = None, device=0, env_offset=0):
        from .configs import build_c_config
        self.lib = _L.load()
        if not torch.cuda.is_available():
            raise RuntimeError("cartpolesimulation_amd needs an MI355X (gfx950) visible to PyTorch-ROCm; there is no CPU fallback.")
        self.E, self.env_offset = int(E), int(env_offset)
        self.mppi, self.phys = mppi or MPPIConfig(), phys or PhysicalParameters()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        cfg = build_c_config(self.E, self.mppi, self.phys)
        self._g = C.c_void_p()
        rc = self.lib.cpmppi_groups_create(C.byref(cfg), self.device.index, int(groups), self.env_offset, C.byref(self._g))
        if rc != 0:
            raise _L.CpmppiError(rc, self.lib.cpmppi_groups_last_error(None).decode())
        self.slices, self.engines, self.streams = [], [], []
        for i in range(self.lib.cpmppi_groups_count(self._g)):
            first, n = C.c_uint32(), C.c_uint32()
            self.lib.cpmppi_groups_slice(self._g, i, C.byref(first), C.byref(n))
            self.slices.append((first.value, first.value + n.value))
            eng = MPPIEngine.from_handle(self.lib.cpmppi_groups_handle(self._g, i), n.value, self.mppi, self.phys, self.device.index)
            st = torch.cuda.ExternalStream(self.lib.cpmppi_groups_stream(self._g, i), device=self.device)
            eng.use_stream(st)
            self.engines.append(eng)
            self.streams.append(st)
        # builds and validates argument blocks over ALL envs (it never launches: the handle behind it is group 0's)
        self.args_engine = MPPIEngine.from_handle(self.lib.cpmppi_groups_handle(self._g, 0), self.E, self.mppi, self.phys, self.device.index)
        self.H, self.N = self.args_engine.H, self.args_engine.N

    def __len__(self):
        return len(self.engines)

    def _check(self, rc):
        if rc != 0:
            raise _L.CpmppiError(rc, self.lib.cpmppi_groups_last_error(self._g).decode())

    def _cur(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def fork(self):
        """Every group stream waits for what has been enqueued on the caller's current stream (uploads, allocations)."""
        self._check(self.lib.cpmppi_groups_fork(self._g, self._cur()))

    def join(self):
        """The caller's current stream waits for every group (before results are read or freed on it)."""
        self._check(self.lib.cpmppi_groups_join(self._g, self._cur()))

    def prepare(self, s0, u_nom, target_position, target_equilibrium, L=None, seed=0, Q_out=None, S_out=None, **kw):
        """-> ONE PreparedStep over the full [E, ...] device tensors (in-kernel Philox noise keyed by the global env index), to be
        handed to `run`."""
        if Q_out is None:
            Q_out = torch.empty(self.E, dtype=torch.float32, device=self.device)
        return self.args_engine.prepare_step(s0, u_nom, target_position, target_equilibrium, L=L, seed=seed, offset=0, env_offset=0,
                                             Q_out=Q_out, S_out=S_out, **kw)

    def comm_init(self, unique_id, world, rank, rccl_path=None, stamped=False, timeout_s=None):
        """cpmppi_groups_comm_init: ONE communicator and side stream for all groups of this device (collective over all ranks);
        `run(..., gather_into=recv_all)` then all-gathers the device's whole u_nom[E, H] once per period."""
        if isinstance(rccl_path, str):
            rccl_path = rccl_path.encode()
        self._check(self.lib.cpmppi_groups_comm_init(self._g, unique_id, int(world), int(rank), rccl_path))
        self.world, self.rank, self.stamped = int(world), int(rank), bool(stamped)
        h0 = self.lib.cpmppi_groups_handle(self._g, 0)
        if stamped:
            self.engines[0]._check(self.lib.cpmppi_comm_set_stamped(h0, 1))
        if timeout_s is not None:
            self.engines[0]._check(self.lib.cpmppi_comm_set_timeout(h0, float(timeout_s)))

    def comm_sync(self):
        """Host wait for every all-gather enqueued so far (after `join` + a stream sync of the caller's); raises on a timeout."""
        self.engines[0]._check(self.lib.cpmppi_comm_sync(self.lib.cpmppi_groups_handle(self._g, 0)))

    def comm_info(self):
        out = _L.cpmppi_comm_info()
        self.engines[0]._check(self.lib.cpmppi_comm_get_info(self.lib.cpmppi_groups_handle(self._g, 0), C.byref(out)))
        return {n: int(getattr(out, n)) for n, _ in out._fields_}

    def run(self, step=None, plant=None, periods=1, offset=0, period=0, n_substeps=None, gather_into=None):
        """cpmppi_groups_run: `periods` control periods of every group enqueued from C, round robin - step (Philox step counter
        offset + k) and, if given, plant step (period + k).  `step` / `plant`: PreparedStep / PreparedPlantStep over the full arrays.
        `gather_into` [world, E*H (+ stamp words)]: cpmppi_groups_run_gather - one all-gather of u_nom[E, H] per period too."""
        sa = pa = None
        if step is not None:
            step.args.offset = int(offset)
            sa = C.byref(step.args)
        if plant is not None:
            plant.args.period = int(period)
            if n_substeps is not None:
                plant.args.n_substeps = int(n_substeps)
            pa = C.byref(plant.args)
        if gather_into is not None:
            self._check(self.lib.cpmppi_groups_run_gather(self._g, sa, pa, int(periods), gather_into.data_ptr()))
            return
        self._check(self.lib.cpmppi_groups_run(self._g, sa, pa, int(periods)))

    def prepare_step(self, s0, u_nom, target_position, target_equilibrium, L=None, seed=0, Q_out=None, S_out=None, **kw):
        """-> one PreparedStep per GROUP over the slices of the given [E, ...] device tensors (for callers that pace the groups
        themselves; `prepare` + `run` is the one-call form)."""
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

    def overlap(self, step, steps=20):
        """Do the groups really run side by side?  `steps` steps of every group one group after the other, then all together:
        -> sum of the groups' times alone / time together (1.0 = serialised on one hardware queue, ~G = perfect overlap).
        `step`: a PreparedStep over the full arrays (`prepare`)."""
        import time
        preps = []
        a = step.args
        keep = step._keep
        for eng, (e0, e1) in zip(self.engines, self.slices):          # the same launches, one group at a time
            b = type(a)()
            C.memmove(C.byref(b), C.byref(a), C.sizeof(a))
            n = e1 - e0
            f = 4                                                      # bytes per float
            b.E = n
            b.s0 = a.s0 + e0 * 6 * f; b.u_nom = a.u_nom + e0 * self.H * f
            b.u_nom_out = (a.u_nom_out + e0 * self.H * f) if a.u_nom_out else None
            b.target_position = a.target_position + e0 * f; b.target_equilibrium = a.target_equilibrium + e0 * f
            b.L = (a.L + e0 * f) if a.L else None
            b.Q_out = a.Q_out + e0 * f
            b.S_out = (a.S_out + e0 * self.N * f) if a.S_out else None
            b.env_offset = self.env_offset + e0
            preps.append((eng, b))
        torch.cuda.synchronize(self.device)

        def timed(which):
            t0 = time.perf_counter()
            for i in range(steps):
                for eng, b in which:
                    b.offset = 10_000 + i
                    eng._check(eng.lib.cpmppi_step(eng._h, C.byref(b), eng._stream()))
            torch.cuda.synchronize(self.device)
            return time.perf_counter() - t0

        timed(preps)                                                   # warm
        alone = sum(timed([p]) for p in preps)
        together = timed(preps)
        del keep
        return alone / together

    def close(self):
        if getattr(self, "_g", None) is not None and self._g.value:
            torch.cuda.synchronize(self.device)
            for eng in self.engines:
                eng.close()
            self.args_engine.close()
            self.lib.cpmppi_groups_destroy(self._g)
            self._g = C.c_void_p()
        self.engines, self.streams = [], []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_schedule_groups(groups: EnvGroups, batch, seed):
    """harness.BatchedCartPoleExperiment.run_schedule with the experiments split over the env groups: every group runs its own
    chain of (controller step, plant step) launches on its own stream; ALL periods of ALL groups are enqueued by one library call
    (cpmppi_groups_run), the groups working in place on their slices of the batch's buffers.  -> the same result dict."""
    from .harness import ScheduleRun
    eng = groups.args_engine
    run = ScheduleRun(eng, batch, seed)                        # the batch's buffers and logs, set up on the caller's stream
    step = groups.prepare(run.s_ctrl, run.u_nom, run.cur_tp, run.cur_te, L=run.cur_L, seed=seed, Q_out=run.Q, **run._prev)
    plant = eng.prepare_plant_step(run.s, run.Q, batch.n_ctrl, period=0, **run.plant)
    groups.fork()
    if run.m_ctrl is not None and len(np.unique(run.m_ctrl)) > 1:
        # the controller's pole mass follows the plant's (predictor_ODE): a handle parameter read when a launch is enqueued - one
        # library call per period instead of one for the whole run
        for c in range(run.T):
            run.set_controller_mass(c, groups.engines)
            groups.run(step, plant, periods=1, offset=c, period=c)
        run.set_controller_mass(run.T, groups.engines)
    else:
        run.set_controller_mass(0, groups.engines)
        groups.run(step, plant, periods=run.T, offset=0, period=0)
    groups.run(step, plant, periods=1, offset=run.T, period=run.T, n_substeps=run.tail)   # the run's last controller call (+ trailing steps)
    groups.join()
    return dict(states=run.states, dd=run.dd, Q=run.Qs, final_state=run.s, u_nom=run.u_nom, batch=batch)
