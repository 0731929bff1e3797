"""MPPIEngine — thin host wrapper around one libcpmppi handle.  PyTorch-ROCm is used only for device buffers and streams.

All array arguments are float32 ROCm tensors (numpy arrays are uploaded); every call is enqueued on torch's current
stream of the engine's device, so ``torch.cuda.Event`` brackets the kernels correctly.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as _L
from .configs import MPPIConfig, PhysicalParameters, build_c_config, cost_vector


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class MPPIEngine:
    def __init__(self, E, mppi: MPPIConfig = None, phys: PhysicalParameters = None, device=0):
        self.lib = _L.load()
        if not torch.cuda.is_available():
            raise RuntimeError("cartpolesimulation_amd needs an MI355X (gfx950) visible to PyTorch-ROCm; "
                               "there is no CPU fallback.")
        self.mppi = mppi or MPPIConfig()
        self.phys = phys or PhysicalParameters()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        self.E, self.N, self.H = int(E), int(self.mppi.num_rollouts), int(self.mppi.mpc_horizon)
        self.P = self.mppi.num_knots
        self._cfg = build_c_config(self.E, self.mppi, self.phys)
        self._m_pole = float(np.float32(self.phys.m_pole))
        self._h = C.c_void_p()
        rc = self.lib.cpmppi_create(C.byref(self._cfg), self.device.index, C.byref(self._h))
        if rc != 0:
            raise _L.CpmppiError(rc, self.lib.cpmppi_last_error(None).decode())

    @classmethod
    def from_handle(cls, handle, E, mppi: MPPIConfig = None, phys: PhysicalParameters = None, device=0):
        """An engine over a handle somebody else owns (an env group's: pipeline.EnvGroups / cpmppi_groups_create); `close()` then
        only forgets it."""
        self = cls.__new__(cls)
        self.lib = _L.load()
        self.mppi = mppi or MPPIConfig()
        self.phys = phys or PhysicalParameters()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        self.E, self.N, self.H = int(E), int(self.mppi.num_rollouts), int(self.mppi.mpc_horizon)
        self.P = self.mppi.num_knots
        self._cfg = build_c_config(self.E, self.mppi, self.phys)
        self._m_pole = float(np.float32(self.phys.m_pole))
        self._h = C.c_void_p(handle)
        self._borrowed = True
        return self

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            if not getattr(self, "_borrowed", False):
                self.lib.cpmppi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise _L.CpmppiError(rc, self.lib.cpmppi_last_error(self._h).decode())

    def use_stream(self, stream):
        """Enqueue every later call of this engine on `stream` (a torch.cuda.Stream; None = torch's current stream again) -
        env groups that run side by side each keep their own stream (pipeline.py) without a stream context per call."""
        self._fixed_stream = None if stream is None else C.c_void_p(stream.cuda_stream)
        self._fixed_stream_obj = stream                              # (keeps it alive)

    def _stream(self):
        fixed = getattr(self, "_fixed_stream", None)
        if fixed is not None:
            return fixed
        # the caller's current stream as a raw handle (torch.cuda.current_stream() builds a Stream object: ~5 us per call,
        # three calls per control step at the host seam)
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if raw is not None:
            return C.c_void_p(raw(self.device.index))
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def tensor(self, x, shape=None):
        """float32 contiguous tensor on the engine's device (uploads numpy / python data)."""
        if x is None:
            return None
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))
        elif x.dtype == torch.float32 and x.device == self.device and x.is_contiguous():
            return x if shape is None else x.reshape(shape)      # (the common case inside optimizer loops: nothing to convert)
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        if shape is not None:
            x = x.reshape(shape)
        return x

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ seams
    def predict(self, s0, Q, L=None, horizon=None):
        """predict_core: s0[B,6], Q[B,H] -> traj[B,H+1,6]."""
        Q = self.tensor(Q)
        if Q.dim() == 3:
            Q = Q[:, :, 0].contiguous()
        B, H = Q.shape
        s0 = self.tensor(s0)
        if s0.dim() == 1:
            s0 = s0.unsqueeze(0).expand(B, 6).contiguous()
        if s0.shape != (B, 6):
            raise ValueError(f"s0 must be [{B},6] (or [6]), got {tuple(s0.shape)}")
        if L is not None:
            L = self.tensor(L).reshape(-1)
            if L.numel() == 1:
                L = L.expand(B).contiguous()
        traj = self.empty(B, H + 1, 6)
        self._check(self.lib.cpmppi_predict(self._h, B, H, _ptr(s0), _ptr(Q), _ptr(L), _ptr(traj), self._stream()))
        return traj

    def trajectory_cost(self, traj, inputs, target_position, target_equilibrium, u_nom=None, u_prev=None,
                        want=("stage", "terminal", "total")):
        traj = self.tensor(traj)
        inputs = self.tensor(inputs)
        if inputs.dim() == 3:
            inputs = inputs[:, :, 0].contiguous()
        B, H = inputs.shape
        if traj.shape != (B, H + 1, 6):
            raise ValueError(f"traj must be [{B},{H + 1},6], got {tuple(traj.shape)}")
        stage = self.empty(B, H) if "stage" in want else None
        term = self.empty(B) if "terminal" in want else None
        total = self.empty(B) if "total" in want else None
        u_nom, u_prev = self.tensor(u_nom), self.tensor(u_prev)
        self._check(self.lib.cpmppi_trajectory_cost(self._h, B, H, _ptr(traj), _ptr(inputs), float(target_position),
                                                    float(target_equilibrium), _ptr(u_nom), _ptr(u_prev), _ptr(stage),
                                                    _ptr(term), _ptr(total), self._stream()))
        return stage, term, total

    def set_cost(self, name, overrides=None):
        cost_id, w = cost_vector(name, overrides)
        arr = (C.c_float * len(w))(*w)
        self._check(self.lib.cpmppi_set_cost_weights(self._h, cost_id, arr, len(w)))

    def apply_pole_mass_of(self, variable_parameters):
        """predictor_ODE takes the pole's mass from variable_parameters at every step (predictors_customization.py:55-58;
        the simulator sends 'm_pole' with every controller.step, CartPole/__init__.py:516); predictor_ODE_v0 does not."""
        if self.mppi.predictor_type != "ODE":
            return
        m = getattr(variable_parameters, "m_pole", None)
        if m is None:
            return
        if isinstance(m, (float, int, np.floating)):            # immutable: the same object as last time means the same value
            if m is getattr(self, "_m_pole_obj", None):
                return
            self._m_pole_obj = m                                # (arrays / tensors may be assigned in place: converted every time)
        a = np.asarray(m.cpu() if hasattr(m, "cpu") else m, dtype=np.float32).reshape(-1)
        if a.size == 0 or not np.all(a == a[0]):
            raise NotImplementedError(f"m_pole must be the same for every env of a handle (got {a[:4]}...)")
        self.set_pole_mass(float(a[0]))

    def set_pole_mass(self, m_pole):
        """The pole mass every later call computes with (predictor_ODE: variable_parameters.m_pole).  No-op when unchanged."""
        m = float(np.float32(m_pole))
        if m != self._m_pole:
            self._check(self.lib.cpmppi_set_pole_mass(self._h, m))
            self._m_pole = m

    def sample(self, seed, offset=0, env_offset=0, E=None, knots=True, delta_u=False):
        E = self.E if E is None else int(E)
        kn = self.empty(E, self.N, self.P) if knots else None
        du = self.empty(E, self.N, self.H) if delta_u else None
        self._check(self.lib.cpmppi_sample(self._h, E, int(seed), int(offset), int(env_offset), _ptr(kn), _ptr(du),
                                           self._stream()))
        return kn, du

    def tiled_empty(self, E=None):
        """An uninitialised perturbation buffer in the library's tiled layout (include/cpmppi.h) for E envs."""
        E = self.E if E is None else int(E)
        return torch.empty(int(self.lib.cpmppi_tiled_floats(self._h, E)), dtype=torch.float32, device=self.device)

    def sample_tiled(self, seed=0, offset=0, env_offset=0, E=None, knots=None, out=None):
        """a17 straight into the tiled layout: Philox knots, or caller ``knots`` [E,N,P] interpolated."""
        kn = None if knots is None else self.tensor(knots)
        if kn is not None and kn.dim() == 2:
            kn = kn.unsqueeze(0)
        E = (self.E if E is None else int(E)) if kn is None else kn.shape[0]
        out = self.tiled_empty(E) if out is None else out
        self._check(self.lib.cpmppi_sample_tiled(self._h, E, int(seed), int(offset), int(env_offset), _ptr(kn), _ptr(out),
                                                 self._stream()))
        return out

    def tile_delta_u(self, delta_u, out=None):
        """delta_u [E,N,H] (reference layout) -> the tiled layout."""
        du = self.tensor(delta_u)
        if du.dim() == 2:
            du = du.unsqueeze(0)
        E = du.shape[0]
        if du.shape != (E, self.N, self.H):
            raise ValueError(f"delta_u must be [E,{self.N},{self.H}], got {tuple(du.shape)}")
        out = self.tiled_empty(E) if out is None else out
        self._check(self.lib.cpmppi_tile_delta_u(self._h, E, _ptr(du), _ptr(out), self._stream()))
        return out

    def untile(self, tiled, E=None):
        """The tiled buffer viewed back as delta_u [E,N,H] (host-side index arithmetic; tests and debugging)."""
        E = self.E if E is None else int(E)
        G, Hq = (self.N + 63) // 64, (self.H + 3) // 4
        t = tiled.reshape(E, G, Hq, 64, 4).permute(0, 1, 3, 2, 4).reshape(E, G * 64, Hq * 4)
        return t[:, :self.N, :self.H].contiguous()

    def interpolate(self, knots):
        knots = self.tensor(knots)
        if knots.dim() == 2:
            knots = knots.unsqueeze(0)
        E = knots.shape[0]
        if knots.shape != (E, self.N, self.P):
            raise ValueError(f"knots must be [E,{self.N},{self.P}], got {tuple(knots.shape)}")
        du = self.empty(E, self.N, self.H)
        self._check(self.lib.cpmppi_interpolate(self._h, E, _ptr(knots), _ptr(du), self._stream()))
        return du

    def reward_weighted_average(self, S, delta_u):
        S, delta_u = self.tensor(S), self.tensor(delta_u)
        if S.dim() == 1:
            S, delta_u = S.unsqueeze(0), delta_u.unsqueeze(0)
        E = S.shape[0]
        if S.shape != (E, self.N) or delta_u.shape != (E, self.N, self.H):
            raise ValueError("S must be [E,N] and delta_u [E,N,H] for this engine's N, H")
        out = self.empty(E, self.H)
        self._check(self.lib.cpmppi_reward_weighted_average(self._h, E, _ptr(S), _ptr(delta_u), _ptr(out),
                                                            self._stream()))
        return out

    def plant_advance(self, s, Q, L=None, n_substeps=10, dt_sim=0.002, states_log=None, Q_log=None, row=0, row_dev=None):
        """In-place plant update of s[E,6] under held controls Q[E].  With ``states_log`` [T+1,E,6] / ``Q_log`` [T,E] the
        same launch records the control period: Q_log[row] = Q, states_log[row+1] = the advanced state; ``row_dev`` (an
        int64 device tensor: the step counter of ``step(offset_dev=...)``, already advanced) replaces ``row`` by its
        value - 1."""
        if not (torch.is_tensor(s) and s.is_cuda and s.dtype == torch.float32 and s.is_contiguous()):
            raise ValueError("s must be a contiguous float32 ROCm tensor (it is updated in place)")
        E = s.shape[0]
        Q = self.tensor(Q).reshape(E)
        L = self.tensor(L).reshape(E) if L is not None else None
        if states_log is None and Q_log is None:
            self._check(self.lib.cpmppi_plant_advance(self._h, E, _ptr(s), _ptr(Q), _ptr(L), int(n_substeps),
                                                      float(dt_sim), self._stream()))
            return s
        for name, t, tail in (("states_log", states_log, (E, 6)), ("Q_log", Q_log, (E,))):
            if t is not None and not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                                      and tuple(t.shape[1:]) == tail):
                raise ValueError(f"{name} must be a contiguous float32 ROCm tensor [T, {', '.join(map(str, tail))}]")
        # control periods the logs can take; the kernel itself refuses to write outside them (device counter case)
        log_rows = min(([states_log.shape[0] - 1] if states_log is not None else []) + ([Q_log.shape[0]] if Q_log is not None else []))
        if row_dev is None:
            if not 0 <= row < log_rows:
                raise IndexError(f"row {row} outside the logs")
        elif not (torch.is_tensor(row_dev) and row_dev.is_cuda and row_dev.dtype == torch.int64):
            raise ValueError("row_dev must be an int64 ROCm tensor")
        self._check(self.lib.cpmppi_plant_advance_record(self._h, E, _ptr(s), _ptr(Q), _ptr(L), int(n_substeps), float(dt_sim),
                                                         _ptr(states_log), _ptr(Q_log), int(max(log_rows, 0)), int(row),
                                                         _ptr(row_dev), self._stream()))
        return s

    def plant_step(self, s, Q, n_substeps, dt_sim=0.002, period=0, period_dev=None, period_steps=None, L=None, states_log=None,
                   dd_log=None, save_every=None, Q_log=None, target_position_table=None, target_equilibrium_table=None,
                   L_table=None, sched_stride=1, target_position_out=None, target_equilibrium_out=None, L_out=None, m_pole=None,
                   m_pole_table=None, L_controller_table=None, Q_disturbance_table=None, Q_bias=0.0, Q_applied_out=None,
                   s_measured=None, state_history=None, latency=0.0, measurement_noise_table=None, angle_offset_table=None,
                   informed_table=None, _prepare=False):
        """cpmppi_plant_step: one control period of the simulated cartpoles with the experiment schedule and the recording in the
        same launch (include/cpmppi.h).  ``s`` [E,6] in place under held ``Q`` [E]; logs ``states_log`` [rows,E,6], ``dd_log``
        [rows,E,2], ``Q_log`` [periods,E]; schedule tables [sched_rows,E] sampled every ``sched_stride`` simulation steps; ``*_out``
        [E] receive the row the NEXT controller call reads.  ``n_substeps`` = 0 records only.  ``m_pole`` [E] / ``m_pole_table``
        [sched_rows,E]: the PLANT's pole mass (default: the config's); ``L_controller_table``: what ``L_out`` publishes instead of
        ``L_table`` (the pole length the controller is told, CartPole/controller_informer.py).  ``Q_disturbance_table`` [periods,E]
        + ``Q_bias``: the plant is driven by (Q + table[period]) + Q_bias (the simulator's additive control disturbance);
        ``Q_applied_out`` [E] receives that control (the next controller call's ``previous_input``).
        The measurement chain (CartPole.add_noise_and_latency): ``s_measured`` [E,6] receives what the next controller call sees -
        the state ``latency`` seconds back (``state_history`` [>= latency / dt_sim + 2, E, 6], zeros with cos = 1 before the run),
        plus ``measurement_noise_table`` [calls,E,4], plus ``angle_offset_table`` [sched_rows,E] float64 on the angle (taken out again
        where ``informed_table`` [sched_rows,E] uint8 says so; None = everywhere)."""
        if not (torch.is_tensor(s) and s.is_cuda and s.dtype == torch.float32 and s.is_contiguous()):
            raise ValueError("s must be a contiguous float32 ROCm tensor (it is updated in place)")
        E = s.shape[0]

        def dev(name, t, tail, dtype=torch.float32):
            if t is None:
                return None
            if not (torch.is_tensor(t) and t.is_cuda and t.dtype == dtype and t.is_contiguous() and tuple(t.shape[1:]) == tail):
                raise ValueError(f"{name} must be a contiguous {dtype} ROCm tensor [rows{''.join(', ' + str(x) for x in tail)}]")
            return t

        a = _L.cpmppi_plant_args()
        Q = self.tensor(Q).reshape(E)
        Lt = self.tensor(L).reshape(E) if L is not None else None
        a.E, a.s, a.Q, a.L = E, s.data_ptr(), Q.data_ptr(), (Lt.data_ptr() if Lt is not None else None)
        a.n_substeps, a.period_steps, a.dt_sim = int(n_substeps), int(period_steps if period_steps is not None else n_substeps), float(dt_sim)
        a.period = int(period)
        if period_dev is not None:
            if not (torch.is_tensor(period_dev) and period_dev.is_cuda and period_dev.dtype == torch.int64):
                raise ValueError("period_dev must be an int64 ROCm tensor")
            a.period_dev = period_dev.data_ptr()
        states_log, dd_log, Q_log = dev("states_log", states_log, (E, 6)), dev("dd_log", dd_log, (E, 2)), dev("Q_log", Q_log, (E,))
        rows = [t.shape[0] for t in (states_log, dd_log) if t is not None]
        a.states_log = states_log.data_ptr() if states_log is not None else None
        a.dd_log = dd_log.data_ptr() if dd_log is not None else None
        a.save_rows = min(rows) if rows else 0
        a.save_every = int(save_every) if save_every else 0
        if Q_log is not None:
            a.Q_log, a.ctrl_rows = Q_log.data_ptr(), Q_log.shape[0]
            if period_dev is None and not 0 <= int(period) < Q_log.shape[0]:
                raise IndexError(f"period {period} outside Q_log")
        tabs = [dev(n, t, (E,)) for n, t in (("target_position_table", target_position_table),
                                             ("target_equilibrium_table", target_equilibrium_table), ("L_table", L_table),
                                             ("m_pole_table", m_pole_table), ("L_controller_table", L_controller_table))]
        sched = [t.shape[0] for t in tabs if t is not None]
        if sched and min(sched) != max(sched):
            raise ValueError("the schedule tables must have the same number of rows")
        (a.target_position_table, a.target_equilibrium_table, a.L_table, a.m_pole_table,
         a.L_controller_table) = [t.data_ptr() if t is not None else None for t in tabs]
        qd = dev("Q_disturbance_table", Q_disturbance_table, (E,))
        if qd is not None:
            if Q_log is not None and qd.shape[0] != Q_log.shape[0]:
                raise ValueError("Q_disturbance_table and Q_log must have the same number of rows (one per controller call)")
            a.Q_disturbance_table, a.ctrl_rows, a.Q_bias = qd.data_ptr(), qd.shape[0], float(Q_bias)
        sm = dev("s_measured", s_measured, (6,))
        if sm is not None and sm.shape[0] != E:
            raise ValueError("s_measured must be [E,6]")
        hist = dev("state_history", state_history, (E, 6))
        nz = dev("measurement_noise_table", measurement_noise_table, (E, 4))
        off = dev("angle_offset_table", angle_offset_table, (E,), torch.float64)
        inf = dev("informed_table", informed_table, (E,), torch.uint8)
        if sm is not None:
            steps = float(latency) / float(dt_sim)
            a.latency_steps = int(steps)
            a.latency_frac = steps - int(steps)
            a.s_measured = sm.data_ptr()
            if hist is not None:
                a.state_history, a.history_len = hist.data_ptr(), hist.shape[0]
            if nz is not None:
                if Q_log is not None and nz.shape[0] != Q_log.shape[0]:
                    raise ValueError("measurement_noise_table and Q_log must have the same number of rows (one per controller call)")
                a.measurement_noise_table, a.ctrl_rows = nz.data_ptr(), nz.shape[0]
            for t in (off, inf):
                if t is not None:
                    if sched and t.shape[0] != sched[0]:
                        raise ValueError("the schedule tables must have the same number of rows")
                    sched.append(t.shape[0])
            a.angle_offset_table = off.data_ptr() if off is not None else None
            a.informed_table = inf.data_ptr() if inf is not None else None
        qa = dev("Q_applied_out", Q_applied_out.reshape(E, 1) if Q_applied_out is not None else None, (1,))
        a.Q_applied_out = qa.data_ptr() if qa is not None else None
        mt = self.tensor(m_pole).reshape(E) if m_pole is not None else None
        a.m_pole = mt.data_ptr() if mt is not None else None
        a.sched_rows, a.sched_stride = (sched[0] if sched else 0), int(sched_stride)
        outs = [dev(n, t.reshape(E, 1) if t is not None else None, (1,)) for n, t in (
            ("target_position_out", target_position_out), ("target_equilibrium_out", target_equilibrium_out), ("L_out", L_out))]
        a.target_position_out, a.target_equilibrium_out, a.L_out = [t.data_ptr() if t is not None else None for t in outs]
        keep = (s, Q, Lt, mt, qd, qa, sm, hist, nz, off, inf, period_dev, states_log, dd_log, Q_log, tabs, outs)
        if _prepare:
            return PreparedPlantStep(self, a, keep)
        self._check(self.lib.cpmppi_plant_step(self._h, C.byref(a), self._stream()))
        del keep
        return s

    def prepare_plant_step(self, *args, **kwargs):
        """The argument block of ``plant_step(...)`` built and validated ONCE (a closed loop calls it every control period with the
        same buffers): ``.run(period=..., n_substeps=...)`` only updates those two fields and enqueues the launch."""
        return self.plant_step(*args, _prepare=True, **kwargs)

    # ------------------------------------------------------------------ GRU predictor (BASELINE configs[4])
    GRU_KEYS = ("w_ih0", "w_hh0", "b_ih0", "b_hh0", "w_ih1", "w_hh1", "b_ih1", "b_hh1", "w_out", "b_out")

    def set_gru(self, model):
        """Attach a GRU-6IN-32H1-32H2-5OUT model: dict of float32 arrays in the torch.nn.GRU layout
        (w_ih0[96,6], w_hh0[96,32], b_ih0[96], b_hh0[96], w_ih1[96,32], w_hh1, b_ih1, b_hh1, w_out[5,32], b_out[5]) and
        optional in_scale/in_shift[6], out_scale/out_shift[5]."""
        shapes = dict(w_ih0=(96, 6), w_hh0=(96, 32), b_ih0=(96,), b_hh0=(96,), w_ih1=(96, 32), w_hh1=(96, 32),
                      b_ih1=(96,), b_hh1=(96,), w_out=(5, 32), b_out=(5,), in_scale=(6,), in_shift=(6,),
                      out_scale=(5,), out_shift=(5,))
        keep = {}
        for k, shp in shapes.items():
            if k in model and model[k] is not None:
                a = np.ascontiguousarray(np.asarray(model[k].detach().cpu() if torch.is_tensor(model[k]) else model[k],
                                                    dtype=np.float32))
                if a.shape != shp:
                    raise ValueError(f"GRU weight {k} must have shape {shp}, got {a.shape}")
                keep[k] = a
            elif k in self.GRU_KEYS:
                raise ValueError(f"GRU weight {k} missing")
        fp = lambda k: keep[k].ctypes.data_as(C.POINTER(C.c_float)) if k in keep else None
        m = _L.cpmppi_gru_model()
        m.inputs, m.hidden, m.layers, m.outputs = 6, 32, 2, 5
        for l in range(2):
            m.w_ih[l], m.w_hh[l], m.b_ih[l], m.b_hh[l] = fp(f"w_ih{l}"), fp(f"w_hh{l}"), fp(f"b_ih{l}"), fp(f"b_hh{l}")
        m.w_out, m.b_out = fp("w_out"), fp("b_out")
        m.in_scale, m.in_shift, m.out_scale, m.out_shift = fp("in_scale"), fp("in_shift"), fp("out_scale"), fp("out_shift")
        self._check(self.lib.cpmppi_set_gru(self._h, C.byref(m)))
        self.has_gru = True

    def gru_predict(self, s0, Q, h0=None, return_hidden=False):
        """Neural predictor seam: s0[B,6] | [6], Q[B,H], h0[2,B,32] -> traj[B,H+1,6] (and final hidden [2,B,32])."""
        Q = self.tensor(Q)
        if Q.dim() == 3:
            Q = Q[:, :, 0].contiguous()
        B, H = Q.shape
        s0 = self.tensor(s0)
        if s0.dim() == 1:
            s0 = s0.unsqueeze(0).expand(B, 6).contiguous()
        h0 = self.tensor(h0, (2, B, 32)) if h0 is not None else None
        traj = self.empty(B, H + 1, 6)
        h_out = self.empty(2, B, 32) if return_hidden else None
        self._check(self.lib.cpmppi_gru_predict(self._h, B, H, _ptr(s0), _ptr(Q), _ptr(h0), _ptr(traj), _ptr(h_out),
                                                self._stream()))
        return (traj, h_out) if return_hidden else traj

    # ------------------------------------------------------------------ cost-only launch and CEM
    def _per_env(self, x, E):
        x = self.tensor(x).reshape(-1)
        return x.expand(E).contiguous() if x.numel() == 1 else x

    def rollout_cost(self, s0, inputs, target_position, target_equilibrium, L=None):
        """inputs[E,N,H] -> S[E,N]: trajectory cost of given control sequences (no update)."""
        inputs = self.tensor(inputs)
        E = inputs.shape[0]
        if inputs.shape != (E, self.N, self.H):
            raise ValueError(f"inputs must be [E,{self.N},{self.H}]")
        s0 = self.tensor(s0, (E, 6))
        tp, te = self._per_env(target_position, E), self._per_env(target_equilibrium, E)
        Lt = self._per_env(L, E) if L is not None else None
        S = self.empty(E, self.N)
        self._check(self.lib.cpmppi_rollout_cost(self._h, E, _ptr(s0), _ptr(inputs), _ptr(tp), _ptr(te), _ptr(Lt), _ptr(S),
                                                 self._stream()))
        return S

    def rollout_cost_grad(self, s0, inputs, target_position, target_equilibrium, L=None, previous_input=None):
        """inputs[E,N,H] -> (S[E,N], grad[E,N,H]): trajectory cost and its derivative w.r.t. the inputs."""
        inputs = self.tensor(inputs)
        E = inputs.shape[0]
        if inputs.shape != (E, self.N, self.H):
            raise ValueError(f"inputs must be [E,{self.N},{self.H}]")
        s0 = self.tensor(s0, (E, 6))
        tp, te = self._per_env(target_position, E), self._per_env(target_equilibrium, E)
        Lt = self._per_env(L, E) if L is not None else None
        pi = self._per_env(previous_input, E) if previous_input is not None else None
        S, grad = self.empty(E, self.N), self.empty(E, self.N, self.H)
        self._check(self.lib.cpmppi_rollout_cost_grad(self._h, E, _ptr(s0), _ptr(inputs), _ptr(tp), _ptr(te), _ptr(Lt),
                                                      _ptr(pi), _ptr(S), _ptr(grad), self._stream()))
        return S, grad

    def adam_step(self, Q, grad, m, v, iteration, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8, gradmax_clip=0.0):
        """One Adam iteration on Q[E,N,H] in place (m, v: caller-owned moments, zero before iteration 1)."""
        for x in (Q, grad, m, v):
            if not (isinstance(x, torch.Tensor) and x.is_contiguous() and x.dtype == torch.float32 and x.shape == Q.shape):
                raise ValueError("Q, grad, m, v must be contiguous float32 device tensors of one shape [E,N,H]")
        self._check(self.lib.cpmppi_adam_step(self._h, Q.shape[0], _ptr(Q), _ptr(grad), _ptr(m), _ptr(v), int(iteration),
                                              float(learning_rate), float(beta1), float(beta2), float(epsilon),
                                              float(gradmax_clip), self._stream()))
        return Q

    def sgd_step(self, Q, grad, learning_rate, gradmax_clip=0.0):
        """Q <- clip(Q - lr * clip_by_norm(grad)) in place."""
        for x in (Q, grad):
            if not (isinstance(x, torch.Tensor) and x.is_contiguous() and x.dtype == torch.float32 and x.shape == Q.shape):
                raise ValueError("Q, grad must be contiguous float32 device tensors of one shape [E,N,H]")
        self._check(self.lib.cpmppi_sgd_step(self._h, Q.shape[0], _ptr(Q), _ptr(grad), float(learning_rate),
                                             float(gradmax_clip), self._stream()))
        return Q

    def cem_sample(self, mean, stdev, seed, offset=0, env_offset=0):
        mean, stdev = self.tensor(mean), self.tensor(stdev)
        E = mean.shape[0]
        Q = self.empty(E, self.N, self.H)
        self._check(self.lib.cpmppi_cem_sample(self._h, E, _ptr(mean), _ptr(stdev), int(seed), int(offset), int(env_offset),
                                               _ptr(Q), self._stream()))
        return Q

    def cem_gmm_sample(self, centres, stdev, seed, offset=0, env_offset=0, return_components=False):
        """centres [E,K,H], stdev [E,H] -> Q [E,N,H]: each rollout drawn around one (uniformly chosen) centre."""
        centres, stdev = self.tensor(centres), self.tensor(stdev)
        E, K = centres.shape[0], centres.shape[1]
        if centres.shape != (E, K, self.H) or stdev.shape != (E, self.H):
            raise ValueError(f"centres must be [E,K,{self.H}] and stdev [E,{self.H}]")
        Q = self.empty(E, self.N, self.H)
        comp = torch.empty(E, self.N, dtype=torch.int32, device=self.device) if return_components else None
        self._check(self.lib.cpmppi_cem_gmm_sample(self._h, E, _ptr(centres), K, _ptr(stdev), int(seed), int(offset),
                                                   int(env_offset), _ptr(Q), _ptr(comp), self._stream()))
        return (Q, comp) if return_components else Q

    def cem_update(self, S, Q, best_k, stdev_min, return_elites=False):
        S, Q = self.tensor(S), self.tensor(Q)
        E = S.shape[0]
        mean, stdev = self.empty(E, self.H), self.empty(E, self.H)
        elites = torch.empty(E, best_k, dtype=torch.int32, device=self.device) if return_elites else None
        self._check(self.lib.cpmppi_cem_update(self._h, E, _ptr(S), _ptr(Q), int(best_k), float(stdev_min), _ptr(mean),
                                               _ptr(stdev), _ptr(elites), self._stream()))
        return (mean, stdev, elites) if return_elites else (mean, stdev)

    def last_launch(self):
        """The rollout-kernel instantiation the most recent step launched (cpmppi_last_launch): dict of the template
        arguments plus `kernel`, its name as rocprofv3 prints it."""
        info = _L.cpmppi_launch_info()
        self._check(self.lib.cpmppi_last_launch(self._h, C.byref(info)))
        d = {n: int(getattr(info, n)) for n, _ in info._fields_}
        d["kernel"] = ("rollout_cost_kernel<%d, %s, %d, %d, %d%s>" % (
            d["cost_id"], "true" if d["math_mode"] == _L.MATH_FAST else "false", (0, 1, 2, 3)[d["noise_kind"]],
            d["rollouts_per_lane"], d["build_variant"], ", PREDICTOR_ODE" if d["ode_predictor"] else ""))
        return d

    def set_profiling(self, enable=True, group=1):
        """HIP-event timing on the launch stream.  ``group`` = 1: every rollout kernel is bracketed; ``group`` = n > 1: one
        bracket around every n consecutive steps, reported as the average per step (an event costs ~5 us on the stream;
        group when small launches are timed by wall clock at the same time)."""
        self._check(self.lib.cpmppi_set_profiling(self._h, int(group) if enable else 0))

    def get_profile(self, max_steps=4096):
        """-> (rollout_ms[n], finalize_ms[n]) of the steps since the last call (synchronises on their events)."""
        a = (C.c_float * max_steps)()
        b = (C.c_float * max_steps)()
        n = C.c_uint32(0)
        self._check(self.lib.cpmppi_get_profile(self._h, a, b, max_steps, C.byref(n)))
        k = min(n.value, max_steps)
        return np.array(a[:k], dtype=np.float64), np.array(b[:k], dtype=np.float64)

    # ------------------------------------------------------------------ the fused hot path
    def step(self, s0, u_nom, target_position, target_equilibrium, L=None, delta_u=None, knots=None, seed=None,
             offset=0, env_offset=0, u_prev=None, Q_out=None, S_out=None, predictor="ODE_v0", h0=None,
             previous_input=None, offset_dev=None, delta_u_tiled=None, u_nom_out=None, gather_into=None, _prepare=False):
        """One MPPI optimizer step for E envs.  ``u_nom`` [E,H] is updated IN PLACE, or — with ``u_nom_out`` [E,H] — only
        read, the updated sequence going to ``u_nom_out`` (two buffers used alternately: shard.NativeGather).

        Exactly one noise source: ``delta_u`` [E,N,H], ``knots`` [E,N,P], ``seed`` (in-kernel Philox) or
        ``delta_u_tiled`` (a buffer from sample_tiled / tile_delta_u).
        ``gather_into`` [world, E*H] (needs a communicator, shard.NativeGather): cpmppi_step_gather — the step plus the
        all-gather of the sequences it writes, ordered on the device, nothing but the kernel on the launch stream.
        Returns (Q_out[E], S_out or None).
        """
        if not (torch.is_tensor(u_nom) and u_nom.is_cuda and u_nom.dtype == torch.float32 and u_nom.is_contiguous()):
            raise ValueError("u_nom must be a contiguous float32 ROCm tensor (it is updated in place)")
        E = u_nom.shape[0]
        if u_nom.shape != (E, self.H) or E > self.E:
            raise ValueError(f"u_nom must be [E<={self.E},{self.H}], got {tuple(u_nom.shape)}")
        given = [x is not None for x in (delta_u, knots, seed, delta_u_tiled)]
        if sum(given) != 1:
            raise ValueError("give exactly one of delta_u, knots, seed, delta_u_tiled")
        a = _L.cpmppi_step_args()
        s0 = self.tensor(s0, (E, 6))
        tp = self.tensor(target_position).reshape(-1)
        te = self.tensor(target_equilibrium).reshape(-1)
        tp = tp.expand(E).contiguous() if tp.numel() == 1 else tp
        te = te.expand(E).contiguous() if te.numel() == 1 else te
        Lt = None
        if L is not None:
            Lt = self.tensor(L).reshape(-1)
            Lt = Lt.expand(E).contiguous() if Lt.numel() == 1 else Lt
        noise = None
        if delta_u_tiled is not None:
            noise = delta_u_tiled
            if not (torch.is_tensor(noise) and noise.is_cuda and noise.dtype == torch.float32 and noise.is_contiguous()
                    and noise.numel() >= int(self.lib.cpmppi_tiled_floats(self._h, E))):
                raise ValueError("delta_u_tiled must be a contiguous float32 ROCm tensor of cpmppi_tiled_floats(E) elements")
            a.noise_kind = _L.NOISE_DELTA_U_TILED
        elif delta_u is not None:
            noise = self.tensor(delta_u, (E, self.N, self.H))
            a.noise_kind = L_NOISE[0]
        elif knots is not None:
            noise = self.tensor(knots, (E, self.N, self.P))
            a.noise_kind = L_NOISE[1]
        else:
            a.noise_kind = L_NOISE[2]
            a.seed, a.offset, a.env_offset = int(seed), int(offset), int(env_offset)
        u_prev = self.tensor(u_prev, (E, self.H)) if u_prev is not None else None
        if Q_out is None:
            Q_out = self.empty(E)
        a.E = E
        a.s0, a.u_nom, a.u_prev = s0.data_ptr(), u_nom.data_ptr(), (u_prev.data_ptr() if u_prev is not None else None)
        a.target_position, a.target_equilibrium = tp.data_ptr(), te.data_ptr()
        a.L = Lt.data_ptr() if Lt is not None else None
        a.noise = noise.data_ptr() if noise is not None else None
        a.Q_out = Q_out.data_ptr()
        a.S_out = S_out.data_ptr() if S_out is not None else None
        if predictor not in ("ODE_v0", "GRU"):
            raise ValueError("predictor must be 'ODE_v0' or 'GRU'")
        a.predictor = _L.PREDICTOR_GRU if predictor == "GRU" else _L.PREDICTOR_ODE_V0
        h0 = self.tensor(h0, (E, 2, 32)) if h0 is not None else None
        a.h0 = h0.data_ptr() if h0 is not None else None
        previous_input = self._per_env(previous_input, E) if previous_input is not None else None
        a.previous_input = previous_input.data_ptr() if previous_input is not None else None
        if u_nom_out is not None:
            if not (torch.is_tensor(u_nom_out) and u_nom_out.is_cuda and u_nom_out.dtype == torch.float32
                    and u_nom_out.is_contiguous() and u_nom_out.shape == u_nom.shape):
                raise ValueError("u_nom_out must be a contiguous float32 ROCm tensor of u_nom's shape")
            a.u_nom_out = u_nom_out.data_ptr()
        if offset_dev is not None:           # int64 device scalar: Philox step counter kept on the device (graph replay)
            if not (torch.is_tensor(offset_dev) and offset_dev.is_cuda and offset_dev.dtype == torch.int64 and offset_dev.numel() == 1):
                raise ValueError("offset_dev must be a one-element int64 ROCm tensor")
            a.offset_dev = offset_dev.data_ptr()
        if _prepare:
            # every tensor the argument block points into is kept alive by the returned object
            return PreparedStep(self, a, (s0, u_nom, tp, te, Lt, noise, u_prev, Q_out, S_out, h0, previous_input, offset_dev, u_nom_out),
                                Q_out, S_out)
        if gather_into is not None:
            self._check(self.lib.cpmppi_step_gather(self._h, C.byref(a), gather_into.data_ptr(), self._stream()))
        else:
            self._check(self.lib.cpmppi_step(self._h, C.byref(a), self._stream()))
        # keep the temporaries alive until the launch is enqueued (stream-ordered frees are safe in torch's allocator)
        return Q_out, S_out

    def step_host(self, s0, u_nom, target_position, target_equilibrium, L, seed, offset, Q_host, env_offset=0):
        """cpmppi_step_host: float32 numpy arrays on the HOST for the state [E,6], the per-env vectors [E] and the result
        ``Q_host`` [E] (written in place); ``u_nom`` [E,H] stays on the device.  One library call per control step:
        staging, copy up, launch (in-kernel Philox noise), copy down, wait.  Synchronous."""
        E = u_nom.shape[0]
        for name, x, shape in (("s0", s0, (E, 6)), ("target_position", target_position, (E,)),
                               ("target_equilibrium", target_equilibrium, (E,)), ("Q_host", Q_host, (E,))) + \
                (() if L is None else (("L", L, (E,)),)):
            if not (isinstance(x, np.ndarray) and x.dtype == np.float32 and x.flags.c_contiguous and x.shape == shape):
                raise ValueError(f"{name} must be a C-contiguous float32 numpy array of shape {shape}")
        self._check(self.lib.cpmppi_step_host(self._h, E, s0.ctypes.data, target_position.ctypes.data, target_equilibrium.ctypes.data,
                                              None if L is None else L.ctypes.data, u_nom.data_ptr(), int(seed), int(offset),
                                              int(env_offset), Q_host.ctypes.data, self._stream()))
        return Q_host

    def prepare_step(self, *args, **kwargs):
        """The argument block of ``step(...)`` built and validated ONCE, for callers whose buffers persist from step to step
        (the host seam's staging block, a closed loop): ``.run(offset=...)`` only updates the Philox step counter and
        enqueues the launch - the per-call argument handling of ``step`` is ~10 us of a 55 us control step."""
        return self.step(*args, _prepare=True, **kwargs)


class PreparedPlantStep:
    """A validated cpmppi_plant_step argument block plus the tensors it points into (MPPIEngine.prepare_plant_step)."""

    def __init__(self, engine, args, keep):
        self.engine, self.args, self._keep = engine, args, keep

    def run(self, period=None, n_substeps=None):
        if period is not None:
            self.args.period = int(period)
        if n_substeps is not None:
            self.args.n_substeps = int(n_substeps)
        e = self.engine
        e._check(e.lib.cpmppi_plant_step(e._h, C.byref(self.args), e._stream()))


class PreparedStep:
    """A validated cpmppi_step argument block plus the tensors it points into (MPPIEngine.prepare_step)."""

    def __init__(self, engine, args, keep, Q_out, S_out):
        self.engine, self.args, self._keep, self.Q_out, self.S_out = engine, args, keep, Q_out, S_out

    def run(self, offset=None, gather_into=None):
        if offset is not None:
            self.args.offset = int(offset)
        e = self.engine
        if gather_into is not None:
            e._check(e.lib.cpmppi_step_gather(e._h, C.byref(self.args), gather_into.data_ptr(), e._stream()))
        else:
            e._check(e.lib.cpmppi_step(e._h, C.byref(self.args), e._stream()))
        return self.Q_out, self.S_out


L_NOISE = (_L.NOISE_DELTA_U, _L.NOISE_KNOTS, _L.NOISE_PHILOX)
