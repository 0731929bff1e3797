"""Closed-loop harness on the device: E simulated cartpoles, each driven by its own MPPI problem instance.

This is the build's counterpart of the reference's experiment loop (run_data_generator.py:9-10 ->
CartPole/data_generator.py:259-367 -> CartPole.run_cartpole_random_experiment, CartPole/__init__.py:659-735, whose
inner update_state loop is :283-324): every control period the controller sees the state and holds Q for
dt_control / dt_simulation plant steps.  Plant (cpmppi_plant_advance) and controller (cpmppi_step) both run on the
GPU and no value crosses PCIe inside the loop; measurement noise, latency and actuator disturbance are OFF as in the
shipped YAML (cartpole_physical_parameters.yml:13-14,18,20).

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
    def run_schedule(self, batch, env_offset=0, graph=False, steps_per_graph=10, knots_fn=None, u_nom0=None):
        """Run the E experiments of a schedule.ExperimentBatch to their end: CartPole.run_cartpole_random_experiment
        (CartPole/__init__.py:659-735) for all of them at once.  Per control period two launches - the fused MPPI step, reading
        the period's target position / equilibrium (/ pole length) from three [E] vectors, and cpmppi_plant_step, which advances
        the plants, records the rows that fall into the period and refills those vectors from the schedule tables for the next
        controller call.  The run ends, as the reference's does, with a controller call on the final state (its control completes
        the last row).  -> dict of device tensors: states [R,E,6], dd [R,E,2] (angleDD, positionDD), Q [periods + 1, E], final
        state and nominal sequences; rows are the simulation steps 0, n_save, 2 n_save, ...
        ``u_nom0`` [E,H]: nominal sequences to start from (a controller that has been stepped before; default zeros).
        ``knots_fn(c)`` (tests): perturbation knots [E,N,P] for controller call c instead of the in-kernel Philox draw."""
        eng, b = self.engine, batch
        E, T = b.E, b.n_periods
        if b.dt_simulation != self.dt_simulation or b.n_ctrl != self.n_sub:
            raise ValueError("the batch was drawn for other time scales than this experiment runner's")
        R = b.n_sim // b.n_save + 1
        s = eng.tensor(b.s0).clone()
        tp_tab = eng.tensor(b.target_position.astype(np.float32))                 # the controller computes in float32
        te_tab = eng.tensor(b.target_equilibrium.astype(np.float32))
        L_tab = eng.tensor(b.L_table) if b.L_table is not None else None
        if L_tab is not None and (b.stride != 1 or L_tab.shape[0] != b.n_sim + 1):
            raise ValueError("a pole-length table is per simulation step: draw the batch with stride 1 (dt_save = dt_simulation)")
        cur_tp, cur_te = tp_tab[0].clone(), te_tab[0].clone()
        cur_L = L_tab[0].clone() if L_tab is not None else (eng.tensor(b.L) if b.L is not None else None)
        u_nom, Q = (eng.zeros(E, eng.H) if u_nom0 is None else eng.tensor(u_nom0, (E, eng.H)).clone()), eng.empty(E)
        states, dd, Qs = eng.zeros(R, E, 6), eng.zeros(R, E, 2), eng.zeros(T + 1, E)
        states[0] = s
        tail = b.n_sim - T * b.n_ctrl                                             # simulation steps after the last controller call
        plant = dict(dt_sim=b.dt_simulation, period_steps=b.n_ctrl, L=None if L_tab is not None else cur_L, states_log=states,
                     dd_log=dd, save_every=b.n_save, Q_log=Qs, target_position_table=tp_tab, target_equilibrium_table=te_tab,
                     L_table=L_tab, sched_stride=b.stride, target_position_out=cur_tp, target_equilibrium_out=cur_te,
                     L_out=cur_L if L_tab is not None else None)

        def control(c, counter=None):
            if knots_fn is not None:
                eng.step(s, u_nom, cur_tp, cur_te, L=cur_L, knots=knots_fn(c), Q_out=Q)
            elif counter is not None:
                eng.step(s, u_nom, cur_tp, cur_te, L=cur_L, seed=self.seed, offset_dev=counter, env_offset=env_offset, Q_out=Q)
            else:
                eng.step(s, u_nom, cur_tp, cur_te, L=cur_L, seed=self.seed, offset=c, env_offset=env_offset, Q_out=Q)

        if graph and knots_fn is None and T > 0:
            counter = torch.zeros(1, dtype=torch.int64, device=s.device)          # controller calls made = Philox step counter

            def one_period():
                control(None, counter)
                eng.plant_step(s, Q, b.n_ctrl, period_dev=counter, **plant)

            side = torch.cuda.Stream(device=s.device)
            side.wait_stream(torch.cuda.current_stream(s.device))
            g = torch.cuda.CUDAGraph()
            per = max(1, min(int(steps_per_graph), T))
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    for _ in range(per):
                        one_period()
            torch.cuda.current_stream(s.device).wait_stream(side)
            for _ in range(T // per):
                g.replay()
            for _ in range(T % per):
                one_period()
            control(None, counter)                                               # the run's last controller call
            eng.plant_step(s, Q, tail, period_dev=counter, **plant)
        else:
            for c in range(T):
                control(c)
                eng.plant_step(s, Q, b.n_ctrl, period=c, **plant)
            control(T)
            eng.plant_step(s, Q, tail, period=T, **plant)
        return dict(states=states, dd=dd, Q=Qs, final_state=s, u_nom=u_nom, batch=b)
