"""Closed-loop harness on the device: E simulated cartpoles, each driven by its own MPPI problem instance.

This is the build's counterpart of the reference's experiment loop (run_data_generator.py:9-10 ->
CartPole/data_generator.py:259-367 -> CartPole.run_cartpole_random_experiment, CartPole/__init__.py:659-735, whose
inner update_state loop is :283-324): every control period the controller sees the state and holds Q for
dt_control / dt_simulation plant steps.  Plant (cpmppi_plant_advance) and controller (cpmppi_step) both run on the
GPU and no value crosses PCIe inside the loop; the simulator's measurement chain (latency, measurement noise, vertical angle offset:
cartpole_physical_parameters.yml:18-28, all off as shipped), its additive control disturbance (:13-15, amplitude 0 as shipped) and its
parameter updaters / controller informer (:29-52) come as tables of the batch (schedule.apply_parameter_schedule).

`run` holds target position, target equilibrium and pole length constant per env; `run_schedule` runs a batch of the data
generator's random experiments (schedule.ExperimentBatch): the target position follows each experiment's random trace and the
target equilibrium flips on its dwell times, tabulated on the host and read by the device loop at its own step counter
(cpmppi_plant_step), with rows saved every dt_save whatever the control period.
"""
import numpy as np
import torch

from .state_utilities import ANGLE_IDX, ANGLED_IDX, ANGLE_COS_IDX, ANGLE_SIN_IDX, POSITION_IDX, POSITIOND_IDX


def generate_random_initial_states(E, rng, track_half_length=0.198, init_limits=None):
    """CartPole/data_generator.py:221-256 with the limits of config_data_gen.yml:14-18, for E envs at once."""
    lim = dict(angle=(0.0, 180.0), angleD=1200.0, position=0.8, positionD=0.5)
    lim.update(init_limits or {})
    s = np.zeros((E, 6), dtype=np.float32)
    s[:, POSITION_IDX] = rng.uniform(-1.0, 1.0, E) * track_half_length * lim["position"]
    s[:, POSITIOND_IDX] = rng.uniform(-1.0, 1.0, E) * track_half_length * lim["positionD"]
    side = np.where(rng.uniform(size=E) > 0.5, 1.0, -1.0)
    angle = side * rng.uniform(lim["angle"][0], lim["angle"][1], E) * (np.pi / 180.0)
    s[:, ANGLE_IDX] = angle
    s[:, ANGLED_IDX] = rng.uniform(-1.0, 1.0, E) * lim["angleD"] * (np.pi / 180.0)
    s[:, ANGLE_COS_IDX], s[:, ANGLE_SIN_IDX] = np.cos(angle), np.sin(angle)
    return s


class BatchedCartPoleExperiment:
    def __init__(self, engine, dt_simulation=0.002, dt_control=0.02, seed=0):
        self.engine = engine
        self.dt_simulation = float(dt_simulation)
        self.n_sub = int(round(dt_control / dt_simulation))
        self.seed = int(seed)

    def run(self, s0, n_control_steps, target_position=0.0, target_equilibrium=1.0, L=None, record=True,
            env_offset=0, graph=False, steps_per_graph=10):
        """-> dict(states[T+1,E,6], Q[T,E]) as device tensors (only if ``record``), final state, final u_nom."""
        eng = self.engine
        s = eng.tensor(s0).clone()
        E = s.shape[0]
        tp = eng.tensor(np.broadcast_to(np.asarray(target_position, dtype=np.float32), (E,)).copy())
        te = eng.tensor(np.broadcast_to(np.asarray(target_equilibrium, dtype=np.float32), (E,)).copy())
        Lt = None if L is None else eng.tensor(np.broadcast_to(np.asarray(L, dtype=np.float32), (E,)).copy())
        u_nom = eng.zeros(E, eng.H)
        Q = eng.empty(E)
        states = eng.empty(n_control_steps + 1, E, 6) if record else None
        Qs = eng.empty(n_control_steps, E) if record else None
        if record:
            states[0] = s
        if graph:
            return self._run_graph(s, u_nom, Q, tp, te, Lt, states, Qs, n_control_steps, env_offset, int(steps_per_graph))
        for t in range(n_control_steps):
            eng.step(s, u_nom, tp, te, L=Lt, seed=self.seed, offset=t, env_offset=env_offset, Q_out=Q)
            # plant + this period's row of the recording in ONE launch (two kernels per control step in all)
            eng.plant_advance(s, Q, L=Lt, n_substeps=self.n_sub, dt_sim=self.dt_simulation, states_log=states, Q_log=Qs, row=t)
        return dict(states=states, Q=Qs, final_state=s, u_nom=u_nom)

    def _run_graph(self, s, u_nom, Q, tp, te, Lt, states, Qs, n_control_steps, env_offset, per_graph):
        """The same loop as ONE captured HIP graph of `per_graph` control steps, replayed: controller step (Philox counter in device
        memory, `offset_dev`), plant + recording at the row the same counter names — no launch argument changes between steps, the
        host only enqueues replays (the launch-bound case: few envs, ~70 us of GPU work per control step)."""
        eng = self.engine
        counter = torch.zeros(1, dtype=torch.int64, device=s.device)          # Philox step counter = control step index
                                                                              # = recording row (after the step: + 1)

        def one_step():
            eng.step(s, u_nom, tp, te, L=Lt, seed=self.seed, offset_dev=counter, env_offset=env_offset, Q_out=Q)
            eng.plant_advance(s, Q, L=Lt, n_substeps=self.n_sub, dt_sim=self.dt_simulation, states_log=states, Q_log=Qs,
                              row_dev=counter)

        side = torch.cuda.Stream(device=s.device)
        side.wait_stream(torch.cuda.current_stream(s.device))
        g = torch.cuda.CUDAGraph()
        per_graph = max(1, min(per_graph, n_control_steps))
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for _ in range(per_graph):
                    one_step()
        torch.cuda.current_stream(s.device).wait_stream(side)
        for _ in range(n_control_steps // per_graph):
            g.replay()
        for _ in range(n_control_steps % per_graph):          # the remainder, launched directly with the same device counter
            one_step()
        return dict(states=states, Q=Qs, final_state=s, u_nom=u_nom)

    # ------------------------------------------------------------------ the data generator's experiments (moving targets)
    def run_schedule(self, batch, env_offset=0, graph=False, steps_per_graph=10, knots_fn=None, u_nom0=None, optimizer=None):
        """Run the E experiments of a schedule.ExperimentBatch to their end: CartPole.run_cartpole_random_experiment
        (CartPole/__init__.py:659-735) for all of them at once.  Per control period two launches - the fused MPPI step, reading
        the period's target position / equilibrium (/ pole length) from three [E] vectors, and cpmppi_plant_step, which advances
        the plants, records the rows that fall into the period and refills those vectors from the schedule tables for the next
        controller call.  The run ends, as the reference's does, with a controller call on the final state (its control completes
        the last row).  -> dict of device tensors: states [R,E,6], dd [R,E,2] (angleDD, positionDD), Q [periods + 1, E], final
        state and nominal sequences; rows are the simulation steps 0, n_save, 2 n_save, ...
        ``u_nom0`` [E,H]: nominal sequences to start from (a controller that has been stepped before; default zeros).
        ``knots_fn(c)`` (tests): perturbation knots [E,N,P] for controller call c instead of the in-kernel Philox draw.
        ``optimizer``: any of the package's optimizers configured for the batch's E envs (`controller_mpc(..., num_envs=E)
        .configure(...).optimizer`: cem, rpgd, gradient, ...) computes the controls instead of the fused MPPI step - host-paced, one
        `optimizer.step` per control period on device tensors, the plant / schedule / recording launch unchanged."""
        run = ScheduleRun(self.engine, batch, self.seed, env_offset=env_offset, knots_fn=knots_fn, u_nom0=u_nom0, optimizer=optimizer)
        if batch.dt_simulation != self.dt_simulation or batch.n_ctrl != self.n_sub:
            raise ValueError("the batch was drawn for other time scales than this experiment runner's")
        if graph and optimizer is not None:
            raise ValueError("an optimizer object is paced by the host: run this batch launched (graph=False)")
        if graph and knots_fn is None and run.T > 0:
            run.capture(steps_per_graph)
            while run.periods_left:
                run.enqueue_next()
        else:
            while run.periods_left:
                run.enqueue_next()
        return run.finish()


class ScheduleRun:
    """The device loop of BatchedCartPoleExperiment.run_schedule as an object that enqueues one piece at a time, so that several
    runs - env groups on their own streams, pipeline.run_schedule_groups - can be interleaved by one host thread."""

    def __init__(self, engine, batch, seed, env_offset=0, knots_fn=None, u_nom0=None, optimizer=None):
        self.eng, self.b, self.seed, self.env_offset, self.knots_fn = engine, batch, int(seed), int(env_offset), knots_fn
        self.optimizer = optimizer
        if optimizer is not None:
            if getattr(optimizer, "num_envs", batch.E) != batch.E:
                raise ValueError(f"the optimizer is configured for {optimizer.num_envs} envs, the batch has {batch.E}")
            if getattr(optimizer, "variable_parameters", None) is None:
                from types import SimpleNamespace
                optimizer.variable_parameters = SimpleNamespace()
        eng, b = engine, batch
        E, T = b.E, b.n_periods
        self.T, self.c = T, 0
        R = b.n_sim // b.n_save + 1
        self.s = s = eng.tensor(b.s0).clone()
        tp_tab = eng.tensor(b.target_position.astype(np.float32))                 # the controller computes in float32
        te_tab = eng.tensor(b.target_equilibrium.astype(np.float32))
        L_tab = eng.tensor(b.L_table) if b.L_table is not None else None
        m_tab = eng.tensor(b.m_pole_table) if b.m_pole_table is not None else None
        for tab in (L_tab, m_tab):
            if tab is not None and (b.stride != 1 or tab.shape[0] != b.n_sim + 1):
                raise ValueError("a pole-length / pole-mass table is per simulation step: draw the batch with stride 1 (dt_save = dt_simulation)")
        Lc_tab = None
        if L_tab is not None and b.informed is not None:                            # what the controller is TOLD: the true length or the initial one
            told = np.asarray(b.informed, bool)
            Lc_tab = eng.tensor(np.where(told if told.ndim == 2 else told[:, None], np.asarray(b.L_table, np.float32),
                                         np.asarray(b.L_table, np.float32)[0]))
        self.cur_tp, self.cur_te = tp_tab[0].clone(), te_tab[0].clone()
        self.cur_L = (Lc_tab if Lc_tab is not None else L_tab)[0].clone() if L_tab is not None else (eng.tensor(b.L) if b.L is not None else None)
        self.u_nom = eng.zeros(E, eng.H) if u_nom0 is None else eng.tensor(u_nom0, (E, eng.H)).clone()
        self.Q = eng.empty(E)
        self.states, self.dd, self.Qs = eng.zeros(R, E, 6), eng.zeros(R, E, 2), eng.zeros(T + 1, E)
        self.states[0] = s
        self.tail = b.n_sim - T * b.n_ctrl                                        # simulation steps after the last controller call
        self.plant = dict(dt_sim=b.dt_simulation, period_steps=b.n_ctrl, L=None if L_tab is not None else self.cur_L,
                          states_log=self.states, dd_log=self.dd, save_every=b.n_save, Q_log=self.Qs, target_position_table=tp_tab,
                          target_equilibrium_table=te_tab, L_table=L_tab, sched_stride=b.stride, target_position_out=self.cur_tp,
                          target_equilibrium_out=self.cur_te, L_out=self.cur_L if L_tab is not None else None, m_pole_table=m_tab,
                          L_controller_table=Lc_tab)
        if b.Q_disturbance is not None:                                           # the simulator's additive control disturbance
            qd = np.ascontiguousarray(b.Q_disturbance, np.float32)
            if qd.shape != (T + 1, E):
                raise ValueError(f"Q_disturbance is {qd.shape}, expected one row per controller call {(T + 1, E)}")
            self.plant.update(Q_disturbance_table=eng.tensor(qd), Q_bias=float(b.Q_bias))
        # the measurement chain (CartPole.add_noise_and_latency): the controller reads s_meas, which every plant step refills; the
        # t = 0 call sees the true state (:869-870)
        self.s_ctrl = self.s
        if b.latency or b.measurement_noise is not None or b.angle_offset is not None:
            self.s_ctrl = self.s.clone()
            self.plant.update(s_measured=self.s_ctrl, latency=float(b.latency))
            n_back = float(b.latency) / b.dt_simulation
            if n_back > 0:
                hist = np.zeros((int(n_back) + 2, E, 6), np.float32)
                hist[:, :, 2] = 1.0                                               # (CartPole/latency_adder.py:25-26)
                self.plant.update(state_history=eng.tensor(hist))
            if b.measurement_noise is not None:
                nz = np.ascontiguousarray(b.measurement_noise, np.float32)
                if nz.shape != (T + 1, E, 4):
                    raise ValueError(f"measurement_noise is {nz.shape}, expected one row per controller call {(T + 1, E, 4)}")
                self.plant.update(measurement_noise_table=eng.tensor(nz))
            if b.angle_offset is not None:
                if b.stride != 1 or b.angle_offset.shape[0] != b.n_sim + 1:
                    raise ValueError("the angle-offset table is per simulation step: draw the batch with stride 1")
                self.plant.update(angle_offset_table=torch.as_tensor(np.ascontiguousarray(b.angle_offset, np.float64), device=self.s.device))
                if b.informed is not None:
                    told = np.asarray(b.informed, bool)
                    told = told if told.ndim == 2 else np.broadcast_to(told[:, None], (len(told), E))
                    self.plant.update(informed_table=torch.as_tensor(np.ascontiguousarray(told, np.uint8), device=self.s.device))
        # the control applied in the last period = the next call's Q_ccrc / "Q_applied_-1" (CartPole/__init__.py:489, 517-518), for the
        # cost plugins that read a previous input (0 before the first update, :838)
        from .optimizer_mppi import PREVIOUS_INPUT_COSTS
        self.prev_Q = None
        if eng.mppi.cost_function_specification in PREVIOUS_INPUT_COSTS:
            self.prev_Q = eng.zeros(E)
            self.plant.update(Q_applied_out=self.prev_Q)
        # the pole MASS the controller computes with (predictor_ODE takes it from the simulator's 'm_pole' attribute at every call,
        # predictors_customization.py:55-58; predictor_ODE_v0 does not): one value per handle, so it can follow the plant's mass only
        # when every experiment of the batch has the same schedule (the deterministic updater modes) - told or not as the informer says
        self.m_ctrl = None
        if b.m_pole_table is not None and getattr(eng.mppi, "predictor_type", "ODE_v0") == "ODE":
            mt = np.asarray(b.m_pole_table, np.float32)
            if (mt == mt[:, :1]).all():
                calls = np.minimum(np.arange(T + 1) * b.n_ctrl, mt.shape[0] - 1)
                told = np.ones(T + 1, bool)
                shared = True
                if b.informed is not None:
                    inf = np.asarray(b.informed, bool)
                    if inf.ndim == 2:
                        shared = bool((inf == inf[:, :1]).all())
                        inf = inf[:, 0]
                    told = inf[np.minimum(calls, len(inf) - 1)]
                if shared:
                    self.m_ctrl = np.where(told, mt[calls, 0], mt[0, 0]).astype(np.float32)
                else:
                    import warnings
                    warnings.warn("the informer differs between experiments ('switching_random'): the controller's one pole mass cannot follow "
                                  "it and stays the handle's; the plants follow their own tables")
        self.counter = self.graph = None
        self.per = 0
        self._prep = self._prep_plant = None                       # argument blocks built once (the launched Philox loop)

    @property
    def _prev(self):
        return {} if self.prev_Q is None else {"previous_input": self.prev_Q}

    @property
    def periods_left(self):
        return self.c < self.T

    def set_controller_mass(self, c, engines=None):
        """Before controller call c: the pole mass the simulator would hand it (the launch reads it from the handle when enqueued)."""
        if self.m_ctrl is not None:
            for e in (engines or [self.eng]):
                e.set_pole_mass(float(self.m_ctrl[c]))

    def _control(self, c):
        eng = self.eng
        if self.optimizer is not None:
            # what the simulator hands controller.step (CartPole/__init__.py:509-520), as attributes of the optimizer's
            # variable_parameters; the optimizer returns the controls as a device tensor
            vp = self.optimizer.variable_parameters
            vp.target_position, vp.target_equilibrium = self.cur_tp, self.cur_te
            if self.cur_L is not None:
                vp.L = self.cur_L
            if self.m_ctrl is not None:
                vp.m_pole = float(self.m_ctrl[c])
            if self.prev_Q is not None:
                vp.Q_ccrc = self.prev_Q
                setattr(vp, "Q_applied_-1", self.prev_Q)
            t = float(self.b.times[min(c * self.b.n_ctrl, len(self.b.times) - 1)])
            q = self.optimizer.step(self.s_ctrl, t, as_tensor=True)
            self.Q.copy_(q.reshape(-1))
            return
        if c is not None:
            self.set_controller_mass(c)
        if self.knots_fn is not None:
            eng.step(self.s_ctrl, self.u_nom, self.cur_tp, self.cur_te, L=self.cur_L, knots=self.knots_fn(c), Q_out=self.Q, **self._prev)
        elif self.counter is not None:
            eng.step(self.s_ctrl, self.u_nom, self.cur_tp, self.cur_te, L=self.cur_L, seed=self.seed, offset_dev=self.counter,
                     env_offset=self.env_offset, Q_out=self.Q, **self._prev)
        else:
            eng.step(self.s_ctrl, self.u_nom, self.cur_tp, self.cur_te, L=self.cur_L, seed=self.seed, offset=c, env_offset=self.env_offset,
                     Q_out=self.Q, **self._prev)

    def _period(self, c):
        if self.counter is None and self.knots_fn is None and self.optimizer is None:
            # the launched loop: two library calls per period on argument blocks built once (the Python-side argument handling of
            # step + plant_step is ~30 us per period - more than the GPU needs for a few dozen envs)
            self.set_controller_mass(c)
            if self._prep is None:
                self._prep = self.eng.prepare_step(self.s_ctrl, self.u_nom, self.cur_tp, self.cur_te, L=self.cur_L, seed=self.seed, offset=0,
                                                   env_offset=self.env_offset, Q_out=self.Q, **self._prev)
                self._prep_plant = self.eng.prepare_plant_step(self.s, self.Q, self.b.n_ctrl, period=0, **self.plant)
            self._prep.run(offset=c)
            self._prep_plant.run(period=c)
            return
        self._control(c)
        if self.counter is not None:
            self.eng.plant_step(self.s, self.Q, self.b.n_ctrl, period_dev=self.counter, **self.plant)
        else:
            self.eng.plant_step(self.s, self.Q, self.b.n_ctrl, period=c, **self.plant)

    def capture(self, steps_per_graph=10):
        """Capture `steps_per_graph` control periods as ONE HIP graph (device step counter: Philox offset = schedule row =
        recording row, no launch argument changes between periods); enqueue_next then replays it."""
        if self.optimizer is not None:
            raise ValueError("an optimizer object is paced by the host: run this batch launched (graph=False)")
        if self.m_ctrl is not None and len(np.unique(self.m_ctrl)) > 1:
            raise ValueError("a captured graph replays ONE pole mass for the controller: run this batch launched (graph=False)")
        dev = self.s.device
        # the graph replays the handle's controller mass as it is when CAPTURED: the run's (constant) one, which may differ from
        # the mass the handle was created with (an uninformed controller; advisor, round 5)
        self.set_controller_mass(0)
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)              # controller calls made
        fixed = getattr(self.eng, "_fixed_stream_obj", None)
        cap = torch.cuda.Stream(device=dev)
        launch = fixed if fixed is not None else torch.cuda.current_stream(dev)
        cap.wait_stream(launch)
        cap.wait_stream(torch.cuda.current_stream(dev))
        self.graph = torch.cuda.CUDAGraph()
        self.per = max(1, min(int(steps_per_graph), self.T))
        self.eng.use_stream(cap)
        try:
            with torch.cuda.stream(cap):
                with torch.cuda.graph(self.graph, stream=cap):
                    for _ in range(self.per):
                        self._period(None)
        finally:
            self.eng.use_stream(fixed)
        launch.wait_stream(cap)

    def enqueue_next(self):
        """The next control period(s) of the run: one period launched, or one replay of the captured graph."""
        if self.graph is not None and self.T - self.c >= self.per:
            fixed = getattr(self.eng, "_fixed_stream_obj", None)
            if fixed is not None:
                with torch.cuda.stream(fixed):
                    self.graph.replay()
            else:
                self.graph.replay()
            self.c += self.per
        else:
            self._period(self.c)
            self.c += 1

    def finish(self):
        """The run's last controller call (+ the trailing simulation steps of a length that is not a whole number of periods)."""
        assert not self.periods_left
        self._control(self.T)
        if self.counter is not None:
            self.eng.plant_step(self.s, self.Q, self.tail, period_dev=self.counter, **self.plant)
        else:
            self.eng.plant_step(self.s, self.Q, self.tail, period=self.T, **self.plant)
        return dict(states=self.states, dd=self.dd, Q=self.Qs, final_state=self.s, u_nom=self.u_nom, batch=self.b)
