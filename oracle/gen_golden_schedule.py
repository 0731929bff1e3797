"""Golden vectors for the EXPERIMENT SCHEDULE of the data generator, made by EXECUTING THE REFERENCE'S
OWN CODE - including, here, the simulator class itself.

TEST INFRASTRUCTURE.  Build container only (needs /root/reference).  Usage:
    cd /root/reference && python -B /root/repo/oracle/gen_golden_schedule.py
-> tests/golden/schedule.npz

Reference code objects whose outputs are stored (the stand-ins of oracle/ref_shims.py only make them importable):
  * CartPole/random_target_generator.py::Generate_Random_Trace_Function                      ("trace/*")
  * CartPole/data_generator.py::random_experiment_setter.set, generate_random_initial_state   ("setter*/*")
  * CartPole/__init__.py::CartPole - setup_cartpole_random_experiment, run_cartpole_random_experiment, update_state
    (step_time, update_target_position, update_target_equilibrium, cartpole_integration, edge_bounce, wrap, Update_Q,
    cartpole_ode, save_csv_routine) with the in-tree controller Control_Toolkit_ASF/Controllers/controller_mppi_cartpole.py
    in the loop, and CartPole/csv_logger.py writing the recording                                ("exp*/*")
What is configuration here, not code: the dict random_experiment_setter reads through load_config (config_data_gen.yml with
a seed, a short experiment and fast-moving targets), the controller's problem size and seed, the rng of the CartPole instance
(the shipped YAML seeds it from the clock).  Q_update_time (wall-clock seconds of the controller call) is not stored.
"""
import copy
import io
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
os.chdir(ref_shims.REFERENCE_ROOT)

import yaml  # noqa: E402
from numpy.random import SFC64, Generator  # noqa: E402
import Control_Toolkit_ASF.Controllers.controller_mppi_cartpole as LEG  # noqa: E402

APP = ref_shims.install_app({"mppi-cartpole": LEG.controller_mppi_cartpole})
import CartPole.data_generator as DG  # noqa: E402
from CartPole.random_target_generator import Generate_Random_Trace_Function  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
f32 = np.float32


def accumulated_times(n, dt=0.002):
    """time after g calls of CartPole.step_time (:326-327): g additions of dt in float64."""
    t, out = 0.0, [0.0]
    for _ in range(n):
        t = t + dt
        out.append(t)
    return np.array(out)


# ------------------------------------------------------------------------------------------------ trace function KATs
TRACE_CASES = [
    # name, length, complexity, interpolation, turning_points, period, start, end, used fraction, rng seed
    ("previous_regular", 10.0, 1, "previous", None, "regular", 0.05, -0.03, 1.0, 11),
    ("linear_regular", 10.0, 1, "linear", None, "regular", 0.05, -0.03, 1.0, 12),
    ("smooth_regular", 10.0, 1, "0-derivative-smooth", None, "regular", 0.05, -0.03, 1.0, 13),
    ("previous_random", 7.0, 2, "previous", None, "random", None, 0.1, 0.8, 14),
    ("linear_random", 7.0, 2, "linear", None, "random", -0.12, None, 0.8, 15),
    ("smooth_random", 7.0, 2, "0-derivative-smooth", None, "random", None, None, 0.8, 16),
    ("none_smooth", 0.5, 1, "0-derivative-smooth", None, "regular", 0.04, 0.02, 1.0, 17),          # 0 turning points
    ("none_linear", 0.5, 1, "linear", [], "regular", 0.04, 0.02, 1.0, 18),
    ("one_start", 1.5, 1, "0-derivative-smooth", None, "regular", 0.06, -0.06, 1.0, 19),             # 1 turning point
    ("one_end", 1.5, 1, "linear", None, "regular", None, -0.06, 1.0, 20),
    ("one_free", 1.5, 1, "previous", None, "regular", None, None, 1.0, 21),
    ("given_four", 4.0, 1, "0-derivative-smooth", [0.0, 0.1, -0.1, 0.0], "regular", 0.5, 0.5, 1.0, 22),
    ("given_four_linear", 4.0, 1, "linear", [0.0, 0.1, -0.1, 0.0], "random", None, None, 1.0, 23),
    ("given_one", 3.0, 1, "previous", [0.07], "regular", None, None, 1.0, 24),
    ("clipped", 6.0, 1.5, "0-derivative-smooth", None, "regular", 0.15, -0.19, 0.5, 25),              # start / end beyond the usable track
    ("shipped_like", 12.0, 1, "0-derivative-smooth", None, "regular", 0.0312, None, 1.0, 26),
]


def gen_traces(out):
    names = []
    for (name, length, cx, interp, tps, period, start, end, frac, seed) in TRACE_CASES:
        rng = Generator(SFC64(seed))
        f = Generate_Random_Trace_Function(length_of_experiment=length, rtf_rng=rng, track_relative_complexity=cx,
                                           interpolation_type=interp, turning_points=tps, turning_points_period=period,
                                           start_random_target_position_at=start, end_random_target_position_at=end,
                                           used_track_fraction=frac)
        n = int(np.ceil(length / 0.002))
        t = accumulated_times(n)                                   # what update_target_position evaluates (:370-372)
        extra = Generator(SFC64(1000 + seed)).uniform(0.0, length, 64)
        t_eval = np.concatenate([t[t < length], extra])
        if interp == "0-derivative-smooth":                        # BPoly: periodic beyond the experiment
            t_eval = np.concatenate([t_eval, length + np.array([0.0, 0.25, 1.0, length + 0.3])])
        y = np.array([float(f(x)) for x in t_eval])
        out[f"trace/{name}/t"] = t_eval
        out[f"trace/{name}/y"] = y
        out[f"trace/{name}/y_vec"] = np.asarray(f(t_eval), dtype=np.float64)      # the same through one array call
        names.append(name)
        print(f"trace {name}: {len(t_eval)} points, range [{y.min():.4f}, {y.max():.4f}]")
    out["trace/cases"] = np.array(json.dumps([dict(name=c[0], length=c[1], complexity=c[2], interpolation=c[3],
                                                   turning_points=c[4], period=c[5], start=c[6], end=c[7], used_fraction=c[8],
                                                   seed=c[9]) for c in TRACE_CASES]))


# ------------------------------------------------------------------------------------------------ the experiment setter
def data_gen_config(**over):
    cfg = yaml.safe_load(open("config_data_gen.yml"))
    cfg["controller"] = "mppi-cartpole"
    for k, v in over.items():
        if isinstance(v, dict):
            cfg[k].update(v)
        else:
            cfg[k] = v
    return cfg


def set_legacy_size(N, H, seed):
    LEG.num_rollouts, LEG.mpc_horizon = N, H
    LEG.predictor.configure(batch_size=N, horizon=H, dt=0.02)
    LEG.config_mppi_cartpole["seed"] = seed


def gen_setter(out, tag, cfg, K, cartpole_seed0):
    """K consecutive RES.set(CartPole()) calls: what each experiment starts from and the target trace it follows."""
    DG.load_config = lambda name: copy.deepcopy(cfg)               # (config data; random_experiment_setter.__init__ reads it, :95-96)
    set_legacy_size(32, 5, 1)
    RES = DG.random_experiment_setter()
    n = int(np.ceil(cfg["length_of_experiment"] / cfg["dt"]["simulation"]))
    t = accumulated_times(n, cfg["dt"]["simulation"])
    s0, te, interp, tp = [], [], [], []
    for i in range(K):
        inst = APP.CartPole()
        inst.rng_CartPole = Generator(SFC64(cartpole_seed0 + i))
        inst = RES.set(inst)
        s0.append(np.array(inst.s, dtype=f32)); te.append(int(inst.target_equilibrium)); interp.append(inst.interpolation_type)
        tp.append(np.array([float(inst.random_track_f(x)) for x in t[t < cfg["length_of_experiment"]]]))
    out[f"{tag}/config"] = np.array(json.dumps(cfg))
    out[f"{tag}/cartpole_seed0"] = np.int64(cartpole_seed0)
    out[f"{tag}/s0"] = np.array(s0)
    out[f"{tag}/target_equilibrium"] = np.array(te, dtype=np.int64)
    out[f"{tag}/interpolation_type"] = np.array(interp)
    out[f"{tag}/target_position"] = np.array(tp)                   # [K, sim steps with time < length]
    print(f"{tag}: {K} experiments, interpolation {interp}, te {te}")


# ------------------------------------------------------------------------------------------------ whole experiments
class CallLog:
    """Records every controller.step the simulator makes (instance attribute in front of the bound method: no reference code
    is changed) - state, time, the attributes it was handed, the returned Q and the controller's updated sequence."""

    def __init__(self, ctrl):
        self.ctrl, self.inner, self.calls = ctrl, ctrl.step, []
        ctrl.step = self

    def __call__(self, s, time=None, updated_attributes={}):
        s_in, s_64 = np.array(s, dtype=f32), np.array(s, dtype=np.float64)
        Q = self.inner(s, time, updated_attributes)
        a = updated_attributes
        self.calls.append(dict(s=s_in, s64=s_64, time=float(time), tp=float(a["target_position"]), te=float(a["target_equilibrium"]),
                               L=float(a["L"]), m_pole=float(a.get("m_pole", np.nan)), Q=f32(Q), u=self.ctrl.u_prev.copy(),
                               minS=float(np.min(self.ctrl.S_tilde_k))))
        return Q


def gen_experiments(out, tag, cfg, K, cartpole_seed0, N=256, H=20, ctrl_seed=1234, p_Q=None, L_updater=None, m_pole_updater=None,
                    informer=None, disturbance=None, sensor=None):
    DG.load_config = lambda name: copy.deepcopy(cfg)
    if p_Q is not None:                                            # (cartpole_physical_parameters.yml `actuator_noise`, read at :83)
        LEG.p_Q = p_Q
    shipped_L = copy.deepcopy(APP.config["cartpole"]["L"])
    if L_updater is not None:                                      # (cartpole_physical_parameters.yml `L:` block, read by CartPole.__init__ :120)
        APP.config["cartpole"]["L"] = dict(L_updater)
        out[f"{tag}/L_updater"] = np.array(json.dumps(L_updater))
    shipped_m, shipped_inf = copy.deepcopy(APP.config["cartpole"]["m_pole"]), copy.deepcopy(
        APP.config["cartpole"]["inform_controller_about_parameters_change"])
    if m_pole_updater is not None:                                 # (the `m_pole:` block, :123)
        APP.config["cartpole"]["m_pole"] = dict(m_pole_updater)
        out[f"{tag}/m_pole_updater"] = np.array(json.dumps(m_pole_updater))
    if informer is not None:                                       # (`inform_controller_about_parameters_change`, :128)
        APP.config["cartpole"]["inform_controller_about_parameters_change"] = dict(informer)
        out[f"{tag}/informer"] = np.array(json.dumps(informer))
    if disturbance is not None:
        # controlDisturbance / controlBias (cartpole_physical_parameters.yml; mode 'additive' as shipped, amplitude 0 as shipped): module-
        # level 0-d float32 arrays read at every controller update (:521-524), and the module-level generator `rng` (:75) they draw from
        APP.controlDisturbance[...], APP.controlBias[...] = disturbance["controlDisturbance"], disturbance["controlBias"]
        APP.rng = Generator(SFC64(disturbance["seed"]))
        out[f"{tag}/disturbance"] = np.array(json.dumps(disturbance))
    shipped_sensor = (APP.config["cartpole"]["latency"], copy.deepcopy(APP.config["cartpole"]["vertical_angle_offset"]))
    if sensor is not None:
        # the measurement chain between plant and controller (add_noise_and_latency, :336-340): `latency` and the vertical_angle_offset
        # block are read by CartPole.__init__ (:136-143); the noise amplitudes and mode are module-level values of CartPole/noise_adder.py
        # (read from the YAML at import, :42-49) and its generator an attribute of the instance's NoiseAdder (:56)
        import CartPole.noise_adder as NA
        APP.config["cartpole"]["latency"] = sensor["latency"]
        APP.config["cartpole"]["vertical_angle_offset"] = dict(sensor["vertical_angle_offset"])
        shipped_noise = (NA.NOISE_MODE, NA.sigma_angle, NA.sigma_position, NA.sigma_angleD, NA.sigma_positionD)
        NA.NOISE_MODE = "ON"
        NA.sigma_angle, NA.sigma_position = sensor["noise"]["sigma_angle"], sensor["noise"]["sigma_position"]
        NA.sigma_angleD, NA.sigma_positionD = sensor["noise"]["sigma_angleD"], sensor["noise"]["sigma_positionD"]
        out[f"{tag}/sensor"] = np.array(json.dumps(sensor))
    RES = DG.random_experiment_setter()
    out[f"{tag}/config"] = np.array(json.dumps(cfg))
    out[f"{tag}/N"], out[f"{tag}/H"], out[f"{tag}/ctrl_seed"] = np.int64(N), np.int64(H), np.int64(ctrl_seed)
    out[f"{tag}/cartpole_seed0"] = np.int64(cartpole_seed0)
    out[f"{tag}/stdev"], out[f"{tag}/p_Q"] = np.float64(LEG.SQRTRHODTINV), np.float64(LEG.p_Q)
    for i in range(K):
        set_legacy_size(N, H, ctrl_seed + i)
        inst = APP.CartPole()
        inst.rng_CartPole = Generator(SFC64(cartpole_seed0 + i))
        if sensor is not None:
            inst.NoiseAdderInstance.rng_noise_adder = Generator(SFC64(sensor["noise"]["seed"]))
        # the controller is created inside RES.set (set_controller, :759-779): hook the class's configure so that the log
        # sits in front of step from the very first call (set_cartpole_state_at_t0 steps the controller twice before the
        # experiment: once on the placeholder state when the controller is set, once on the initial state)
        logs = []
        orig_configure = LEG.controller_mppi_cartpole.configure

        def configure(self, *a, **k):
            r = orig_configure(self, *a, **k)
            logs.append(CallLog(self))
            return r

        LEG.controller_mppi_cartpole.configure = configure
        try:
            inst = RES.set(inst)
        finally:
            LEG.controller_mppi_cartpole.configure = orig_configure
        d = tempfile.mkdtemp()
        data = inst_run(inst, d)
        calls = logs[-1].calls
        key = f"{tag}/{i}"
        for col in data.columns:
            if col == "Q_update_time":
                continue
            v = data[col].to_numpy()
            out[f"{key}/col/{col}"] = np.array([str(x) for x in v]) if v.dtype == object else v
        rows = open(os.path.join(d, "Experiment.csv"), newline="").read().split("\r\n")
        k0 = next(j for j, r in enumerate(rows) if r.startswith("time,"))
        body = [r.rsplit(",", 1)[0] for r in rows[k0:] if r]         # without the last column (Q_update_time)
        out[f"{key}/csv_rows"] = np.array("\r\n".join(body))
        out[f"{key}/csv_preamble"] = np.array("\r\n".join(rows[3:k0]))   # header block below the title / revision lines
        for name in ("s", "time", "tp", "te", "L", "m_pole", "Q", "u", "minS") + (("s64",) if sensor is not None else ()):
            out[f"{key}/call/{name}"] = np.array([c[name] for c in calls])
        out[f"{key}/interpolation_type"] = np.array(inst_interp[-1])
        if L_updater is not None:
            out[f"{key}/L_steps"] = np.array(L_log[-1], dtype=np.float64)   # float(L) after every update_parameters call (one per simulation step)
        if m_pole_updater is not None:
            out[f"{key}/m_pole_steps"] = np.array(M_log[-1], dtype=np.float64)
        print(f"{key}: {len(data)} rows, {len(calls)} controller calls, interpolation {inst_interp[-1]}, "
              f"te flips at rows {np.flatnonzero(np.diff(data['target_equilibrium'].to_numpy()) != 0) + 1}, "
              f"target range [{data['target_position'].min():.4f}, {data['target_position'].max():.4f}]")
    APP.config["cartpole"]["L"] = shipped_L
    APP.config["cartpole"]["m_pole"] = shipped_m
    APP.config["cartpole"]["inform_controller_about_parameters_change"] = shipped_inf
    APP.L[...], APP.m_pole[...] = shipped_L["init_value"], shipped_m["init_value"]      # (module-level arrays the updaters write into)
    if disturbance is not None:
        APP.controlDisturbance[...], APP.controlBias[...] = 0.0, 0.0
    if sensor is not None:
        APP.config["cartpole"]["latency"], APP.config["cartpole"]["vertical_angle_offset"] = shipped_sensor
        NA.NOISE_MODE, NA.sigma_angle, NA.sigma_position, NA.sigma_angleD, NA.sigma_positionD = shipped_noise


inst_interp, L_log, M_log = [], [], []


def inst_run(inst, d):
    inst_interp.append(inst.interpolation_type)
    # the pole length the simulator holds after each update_parameters call (CartPole/__init__.py:529-537): an instance attribute in
    # front of the bound method records it, the method itself is the reference's
    steps, steps_m, inner = [], [], inst.update_parameters

    def update_parameters():
        inner()
        steps.append(float(APP.L))
        steps_m.append(float(APP.m_pole))

    inst.update_parameters = update_parameters
    L_log.append(steps)
    M_log.append(steps_m)
    stderr = sys.stderr
    sys.stderr = io.StringIO()                                     # (tqdm's progress bar)
    try:
        return inst.run_cartpole_random_experiment(csv="Experiment", path_to_experiment_recordings=d, save_mode="offline",
                                                   show_summary_plots=False)
    finally:
        sys.stderr = stderr


if __name__ == "__main__":
    out = {}
    gen_traces(out)
    # the shipped config_data_gen.yml, seeded, 6 s: start_at_target, random end, alternating interpolation, 'up'
    gen_setter(out, "setter_shipped", data_gen_config(seed=101, length_of_experiment=6.0), 6, 500)
    # the other branches of random_experiment_setter.set: random start (no start_at_target), fixed end, one interpolation type,
    # 'down', a narrower usable track, some of the initial state given
    gen_setter(out, "setter_alt", data_gen_config(
        seed=102, length_of_experiment=4.0, start_at_target=False, target_position_end=0.05, initial_target_equilibrium="down",
        track_fraction_usable_for_target_position=0.7,
        random_initial_state=dict(position=0.01, angleD=0.0),
        turning_points=dict(interpolation_type="linear", track_relative_complexity=2)), 4, 600)
    # whole experiments, legacy controller in the loop.  Fast targets (10 turning points per second) and short dwell times so that
    # the first ten control steps already see the target move and the equilibrium flip; dt_save != dt_control both ways
    fast = dict(seed=77, length_of_experiment=1.0, keep_target_equilibrium_x_seconds_up=0.09,
                keep_target_equilibrium_x_seconds_down=0.05, turning_points=dict(track_relative_complexity=10),
                random_initial_state=dict(init_limits=dict(angle=[0.0, 30.0], angleD=100.0, position=0.5, positionD=0.3)))
    gen_experiments(out, "exp_fine", data_gen_config(dt=dict(saving=0.01), **fast), 2, 700)          # previous, then smooth
    gen_experiments(out, "exp_coarse", data_gen_config(dt=dict(saving=0.04), **dict(
        fast, seed=78, initial_target_equilibrium="down", length_of_experiment=0.6,
        turning_points=dict(track_relative_complexity=10, interpolation_type="linear"))), 1, 800)
    # the same kind of experiment with the legacy controller's multiplicative output noise (actuator_noise) switched off: nothing
    # but the controller's own update stands between its nominal sequence and the plant - what a device-resident loop computes
    gen_experiments(out, "exp_device", data_gen_config(dt=dict(saving=0.004), **dict(fast, seed=79, length_of_experiment=0.5)), 2, 900,
                    ctrl_seed=4321, p_Q=0.0)
    # a pole length that changes DURING the experiment (every 7 simulation steps, inside control periods): the plant's order of events
    # with update_parameters, and the recording's L column
    gen_experiments(out, "exp_varL", data_gen_config(dt=dict(saving=0.004), **dict(fast, seed=80, length_of_experiment=0.4)), 1, 950,
                    ctrl_seed=777, p_Q=0.0,
                    L_updater=dict(init_value=0.395, change_every_x_seconds=0.014, mode="bounce", range_random=[0.2, 0.5], range_clip=[0.36, 0.43],
                                   increment=0.01, reset_every_x_seconds="inf"))
    # an experiment whose length is not a whole number of control periods (25 simulation steps: controller calls at 0, 10, 20, five
    # trailing steps) and shorter than the first turning point (no turning points at all: the target is 0 whatever the start)
    gen_experiments(out, "exp_tail", data_gen_config(**dict(fast, seed=81, length_of_experiment=0.05,
                                                            turning_points=dict(track_relative_complexity=1))), 1, 960,
                    ctrl_seed=555, p_Q=0.0)
    # pole length AND pole mass changing during the experiment, the controller told about it only part of the time
    # (inform_controller_about_parameters_change 'switching_regular': the value handed to the controller alternates between the true
    # one and the initial one; the recording's L_for_controller / m_pole_for_controller columns say which)
    gen_experiments(out, "exp_varM", data_gen_config(dt=dict(saving=0.004), **dict(fast, seed=82, length_of_experiment=0.4)), 1, 970,
                    ctrl_seed=888, p_Q=0.0,
                    L_updater=dict(init_value=0.395, change_every_x_seconds=0.014, mode="bounce", range_random=[0.2, 0.5], range_clip=[0.36, 0.43],
                                   increment=0.01, reset_every_x_seconds="inf"),
                    m_pole_updater=dict(init_value=0.087, change_every_x_seconds=0.022, mode="bounce", range_random=[0.015, 0.15],
                                        range_clip=[0.07, 0.1], increment=0.004, reset_every_x_seconds="inf"),
                    informer=dict(mode="switching_regular", change_to_on_after_x_seconds_off=0.05, change_to_off_after_x_seconds_on=0.07))
    # the simulator's CONTROL DISTURBANCE switched on (shipped: mode 'additive' with amplitude 0; its author's note: 0.2-0.5 for data
    # collection): Q_applied = Q_calculated + controlDisturbance * N(0, 1) + controlBias at every controller update, two experiments
    # in a row drawing from ONE generator
    gen_experiments(out, "exp_dist", data_gen_config(dt=dict(saving=0.004), **dict(fast, seed=83, length_of_experiment=0.4)), 2, 980,
                    ctrl_seed=999, p_Q=0.0, disturbance=dict(controlDisturbance=0.3, controlBias=0.05, seed=4242))
    # the MEASUREMENT CHAIN switched on: 5 ms of latency (2.5 simulation steps: interpolated), measurement noise, a vertical-angle
    # offset that moves ('bounce', every 0.03 s), the informer switching (informed: the controller gets the offset taken out again)
    gen_experiments(out, "exp_sensor", data_gen_config(dt=dict(saving=0.004), **dict(fast, seed=84, length_of_experiment=0.4)), 1, 990,
                    ctrl_seed=1111, p_Q=0.0,
                    informer=dict(mode="switching_regular", change_to_on_after_x_seconds_off=0.05, change_to_off_after_x_seconds_on=0.07),
                    sensor=dict(latency=0.005,
                                noise=dict(sigma_angle=0.001, sigma_position=0.0005, sigma_angleD=0.075, sigma_positionD=0.005, seed=777),
                                vertical_angle_offset=dict(init_value=2.0, change_every_x_seconds=0.03, mode="bounce",
                                                           range_random=[-3.141592, 3.141592], range_clip=[-0.05, 0.1], increment=0.01,
                                                           reset_every_x_seconds="inf")))
    np.savez_compressed(os.path.join(OUT, "schedule.npz"), **out)
    print("wrote", os.path.join(OUT, "schedule.npz"), len(out), "arrays")
