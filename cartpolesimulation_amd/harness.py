"""Closed-loop harness on the device: E simulated cartpoles, each driven by its own MPPI problem instance.

This is the build's counterpart of the reference's experiment loop (run_data_generator.py:9-10 ->
CartPole/data_generator.py:259-367 -> CartPole.run_cartpole_random_experiment, CartPole/__init__.py:659-735, whose
inner update_state loop is :283-324): every control period the controller sees the state and holds Q for
dt_control / dt_simulation plant steps.  Plant (cpmppi_plant_advance) and controller (cpmppi_step) both run on the
GPU and no value crosses PCIe inside the loop; measurement noise, latency and actuator disturbance are OFF as in the
shipped YAML (cartpole_physical_parameters.yml:13-14,18,20).
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
