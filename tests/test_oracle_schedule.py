"""CPU: the oracle's restatement of the reference's EXPERIMENT SCHEDULE (oracle/schedule_np.py) against tests/golden/schedule.npz,
which oracle/gen_golden_schedule.py made by running the reference's own code objects - Generate_Random_Trace_Function,
random_experiment_setter.set and the simulator class CartPole itself (update_state with its target-position / target-equilibrium
updates, controller calls and save routine) with the in-tree legacy MPPI controller in the loop."""
import json

import numpy as np
import pytest
from numpy.random import SFC64, Generator

from oracle import oracle_np as O
from oracle import schedule_np as S

f32 = np.float32


@pytest.fixture(scope="module")
def g(golden_dir):
    import os
    return np.load(os.path.join(golden_dir, "schedule.npz"))


def test_random_trace_is_bit_equal_for_every_interpolation_type(g):
    """'previous' / 'linear' (scipy interp1d) and '0-derivative-smooth' (BPoly.from_derivatives, periodic), regular and random
    turning-point times, 0 / 1 / n / given turning points, clipping to the usable track: the doubles the reference returns."""
    cases = json.loads(g["trace/cases"].item())
    assert {c["interpolation"] for c in cases} == {"previous", "linear", "0-derivative-smooth"}
    for c in cases:
        f = S.random_trace(c["length"], Generator(SFC64(c["seed"])), c["complexity"], c["interpolation"], c["turning_points"],
                           c["period"], c["start"], c["end"], c["used_fraction"])
        t, y = g[f"trace/{c['name']}/t"], g[f"trace/{c['name']}/y"]
        assert np.array_equal(f(t), y), c["name"]
        assert np.array_equal(np.array([f(float(x)) for x in t[::37]]), y[::37]), c["name"]     # scalar calls, as the simulator makes them
        hi = float(f32(c["used_fraction"]) * S.THL32)
        assert np.abs(y).max() <= hi
    y = g["trace/clipped/y"]
    assert (np.abs(y) == float(f32(0.5) * S.THL32)).any()            # the clip is exercised


@pytest.mark.parametrize("tag", ["setter_shipped", "setter_alt"])
def test_experiment_setter_draws_what_the_reference_draws(g, tag):
    """random_experiment_setter.set for consecutive experiments: initial state (float32, bit for bit), alternating interpolation
    type, initial target equilibrium, and the whole target trace each experiment then follows."""
    cfg = json.loads(g[f"{tag}/config"].item())
    es = S.ExperimentSetter(cfg)
    K = g[f"{tag}/s0"].shape[0]
    n = int(np.ceil(cfg["length_of_experiment"] / cfg["dt"]["simulation"]))
    t = S.accumulated_times(n, cfg["dt"]["simulation"])
    t = t[t < cfg["length_of_experiment"]]
    for i in range(K):
        st = es.set(Generator(SFC64(int(g[f"{tag}/cartpole_seed0"]) + i)))
        assert np.array_equal(st["s0"], g[f"{tag}/s0"][i])
        assert st["interpolation_type"] == g[f"{tag}/interpolation_type"][i]
        assert st["target_equilibrium"] == g[f"{tag}/target_equilibrium"][i]
        assert np.array_equal(st["trace"](t), g[f"{tag}/target_position"][i])
    if tag == "setter_shipped":
        assert list(g[f"{tag}/interpolation_type"][:4]) == ["previous", "0-derivative-smooth"] * 2      # config_data_gen.yml:33
        assert np.array_equal(g[f"{tag}/target_position"][:, 0], g[f"{tag}/s0"][:, O.POSITION_IDX].astype(np.float64))   # start_at_target


@pytest.mark.parametrize("key", ["exp_fine/0", "exp_fine/1", "exp_coarse/0", "exp_device/0", "exp_device/1"])
def test_experiment_loop_against_the_simulator_class(g, key):
    """Whole experiments of the REAL CartPole class (moving target, target-equilibrium flips, dt_save != dt_control): the
    oracle's loop reproduces the schedule the controller saw and the recording's time / target columns bit for bit, and the
    closed-loop dynamics (state, Q, second derivatives) to 1e-4 while trajectories have not diverged (first 12 controller calls)."""
    tag, i = key.split("/")
    i = int(i)
    cfg = json.loads(g[f"{tag}/config"].item())
    es = S.ExperimentSetter(cfg)
    for j in range(i + 1):                                            # (the setter's rng and alternation carry over)
        st = es.set(Generator(SFC64(int(g[f"{tag}/cartpole_seed0"]) + j)))
    N, H = int(g[f"{tag}/N"]), int(g[f"{tag}/H"])
    ctrl = O.LegacyMPPIController(int(g[f"{tag}/ctrl_seed"]) + i, N, H, SQRTRHOINV=0.02, p_Q=float(g[f"{tag}/p_Q"]))
    cs, ctp = g[f"{key}/call/s"], g[f"{key}/call/tp"]
    # the simulator steps a freshly set controller once on its placeholder state (set_controller -> set_cartpole_state_at_t0,
    # CartPole/__init__.py:759-794) before the experiment's own t = 0 call: same rng stream, same warm nominal sequence
    assert g[f"{key}/call/time"][0] == 0.0 and g[f"{key}/call/time"][1] == 0.0
    ctrl.step(cs[0], f32(ctp[0]), L=O.DEFAULT_PARAMS.L)
    out = S.run_experiment(st, cfg, lambda s, t, tp, te, L: ctrl.step(s, f32(tp), L=L))
    rows, calls = out["rows"], out["calls"]
    col = lambda name: g[f"{key}/col/{name}"]                         # noqa: E731
    # --- the schedule: bit-exact
    assert np.array_equal(rows["time"], col("time"))
    assert np.array_equal(rows["target_position"], col("target_position"))
    assert np.array_equal(rows["target_equilibrium"], col("target_equilibrium"))
    assert np.array_equal(np.array([c["time"] for c in calls]), g[f"{key}/call/time"][1:])
    assert np.array_equal(np.array([c["tp"] for c in calls]), ctp[1:])
    assert np.array_equal(np.array([c["te"] for c in calls]), g[f"{key}/call/te"][1:])
    flips = np.flatnonzero(np.diff(col("target_equilibrium")) != 0)
    assert len(flips) >= 4 and np.ptp(col("target_position")) > 0.02   # the fixture does exercise a moving target and flips
    assert len(calls) == len(cs) - 1
    # --- the dynamics
    K = 12
    Qo, Qf = np.array([c["Q"] for c in calls]), g[f"{key}/call/Q"][1:]
    so, sf = np.array([c["s"] for c in calls]), cs[1:]
    np.testing.assert_allclose(Qo[:K], Qf[:K], atol=1e-4)
    np.testing.assert_allclose(so[:K], sf[:K], atol=1e-4, rtol=1e-4)
    r = (K - 1) * out["n_ctrl"] // out["n_save"]                      # rows recorded before controller call K
    np.testing.assert_allclose(rows["s"][:r, 0], col("angle")[:r], atol=1e-4)
    np.testing.assert_allclose(rows["s"][:r, 4], col("position")[:r], atol=1e-4)
    np.testing.assert_allclose(rows["Q"][:r], col("Q_calculated")[:r], atol=1e-4)
    np.testing.assert_allclose(rows["Q_ccrc"][:r], col("Q_ccrc")[:r].astype(np.float64), atol=1e-4)
    np.testing.assert_allclose(rows["u"][:r], col("u")[:r], atol=2e-4)
    np.testing.assert_allclose(rows["angleDD"][:r], col("angleDD")[:r], atol=1e-3, rtol=1e-4)
    np.testing.assert_allclose(rows["positionDD"][:r], col("positionDD")[:r], atol=1e-3, rtol=1e-4)
    # constant columns of the recording as the reference writes them (shipped physical-parameters YAML)
    assert set(col("L_for_controller")) == {"true"} and set(col("m_pole_for_controller")) == {"true"}
    assert np.all(col("L") == float(f32(0.395))) and np.all(col("m_pole") == float(f32(0.087)))
    assert np.all(col("vertical_angle_offset") == 0.0) and np.all(col("vertical_angle_offset_cos") == 1.0)


def test_row_and_call_counts_follow_the_time_scales(g):
    """dt_save = dt_control / 2 -> two rows per controller call; dt_save = 2 dt_control -> one row per two calls; the last simulation
    step both calls the controller and saves (n = ceil(length / dt_sim) steps)."""
    for key, per_call in (("exp_fine/0", 2.0), ("exp_coarse/0", 0.5)):
        cfg = json.loads(g[f"{key.split('/')[0]}/config"].item())
        n = int(np.ceil(cfg["length_of_experiment"] / cfg["dt"]["simulation"]))
        n_ctrl, n_save = int(np.rint(cfg["dt"]["control"] / cfg["dt"]["simulation"])), int(np.rint(cfg["dt"]["saving"] / cfg["dt"]["simulation"]))
        assert len(g[f"{key}/col/time"]) == n // n_save + 1
        assert len(g[f"{key}/call/time"]) == 2 + n // n_ctrl
        assert n_ctrl / n_save == per_call


def test_experiment_with_a_pole_length_that_changes_during_control_periods(g):
    """exp_varL: the reference's simulator with its L updater in 'bounce' mode, a change every 7 simulation steps (inside control
    periods).  The oracle's loop with the recorded per-step pole length reproduces the plant (update_parameters is the first thing a
    simulation step does), the recording's L column exactly, and the closed loop to 1e-4 over the first controller calls."""
    tag, key = "exp_varL", "exp_varL/0"
    cfg = json.loads(g[f"{tag}/config"].item())
    st = S.ExperimentSetter(cfg).set(Generator(SFC64(int(g[f"{tag}/cartpole_seed0"]))))
    N, H = int(g[f"{tag}/N"]), int(g[f"{tag}/H"])
    ctrl = O.LegacyMPPIController(int(g[f"{tag}/ctrl_seed"]), N, H, SQRTRHOINV=0.02, p_Q=float(g[f"{tag}/p_Q"]))
    cs = g[f"{key}/call/s"]
    ctrl.step(cs[0], f32(g[f"{key}/call/tp"][0]), L=O.DEFAULT_PARAMS.L)
    L_steps = np.concatenate([[float(f32(0.395))], g[f"{key}/L_steps"]])
    assert len(np.unique(L_steps)) > 10
    # (the legacy controller predicts with the default pole length whatever the simulator tells it: its module-level predictor is
    # configured without variable_parameters, controller_mppi_cartpole.py:51-52)
    out = S.run_experiment(st, cfg, lambda s, t, tp, te, L: ctrl.step(s, f32(tp), L=O.DEFAULT_PARAMS.L), L_steps=L_steps)
    rows, calls = out["rows"], out["calls"]
    assert np.array_equal(rows["L"], g[f"{key}/col/L"]) and np.array_equal(np.array([c["L"] for c in calls]), g[f"{key}/call/L"][1:])
    assert np.array_equal(rows["time"], g[f"{key}/col/time"]) and np.array_equal(rows["target_position"], g[f"{key}/col/target_position"])
    K = 12
    np.testing.assert_allclose(np.array([c["Q"] for c in calls])[:K], g[f"{key}/call/Q"][1:K + 1], atol=1e-4)
    np.testing.assert_allclose(np.array([c["s"] for c in calls])[:K], cs[1:K + 1], atol=1e-4, rtol=1e-4)
    r = (K - 1) * out["n_ctrl"] // out["n_save"]
    np.testing.assert_allclose(rows["angleDD"][:r], g[f"{key}/col/angleDD"][:r], atol=1e-3, rtol=1e-4)
    # ... and with the pole length held at its initial value the plant does NOT follow the reference (the test has teeth)
    ctrl2 = O.LegacyMPPIController(int(g[f"{tag}/ctrl_seed"]), N, H, SQRTRHOINV=0.02, p_Q=float(g[f"{tag}/p_Q"]))
    ctrl2.step(cs[0], f32(g[f"{key}/call/tp"][0]), L=O.DEFAULT_PARAMS.L)
    st2 = S.ExperimentSetter(cfg).set(Generator(SFC64(int(g[f"{tag}/cartpole_seed0"]))))
    out2 = S.run_experiment(st2, cfg, lambda s, t, tp, te, L: ctrl2.step(s, f32(tp), L=O.DEFAULT_PARAMS.L))
    assert np.abs(out2["rows"]["angleDD"][:r] - g[f"{key}/col/angleDD"][:r]).max() > 0.05


def test_experiment_with_pole_mass_and_length_changing_and_a_switching_informer(g):
    """exp_varM: the reference's simulator with BOTH parameter updaters in 'bounce' mode (L every 7, m_pole every 11 simulation steps)
    and `inform_controller_about_parameters_change` 'switching_regular' (the controller is handed the true values only part of the
    time).  The oracle's loop reproduces the plant under the changing pole mass, the recording's L / m_pole / *_for_controller columns
    and what every controller call was told (L, m_pole) exactly; the informer is the oracle's restatement, not a recorded input."""
    tag, key = "exp_varM", "exp_varM/0"
    cfg = json.loads(g[f"{tag}/config"].item())
    st = S.ExperimentSetter(cfg).set(Generator(SFC64(int(g[f"{tag}/cartpole_seed0"]))))
    N, H = int(g[f"{tag}/N"]), int(g[f"{tag}/H"])
    new_ctrl = lambda: O.LegacyMPPIController(int(g[f"{tag}/ctrl_seed"]), N, H, SQRTRHOINV=0.02, p_Q=float(g[f"{tag}/p_Q"]))   # noqa: E731
    cs = g[f"{key}/call/s"]
    L_steps = np.concatenate([[float(f32(0.395))], g[f"{key}/L_steps"]])
    m_steps = np.concatenate([[float(f32(0.087))], g[f"{key}/m_pole_steps"]])
    assert len(np.unique(L_steps)) > 5 and len(np.unique(m_steps)) > 5
    inf_cfg = json.loads(g[f"{tag}/informer"].item())

    def run(**kw):
        ctrl = new_ctrl()
        ctrl.step(cs[0], f32(g[f"{key}/call/tp"][0]), L=O.DEFAULT_PARAMS.L)
        st_ = S.ExperimentSetter(cfg).set(Generator(SFC64(int(g[f"{tag}/cartpole_seed0"]))))
        return S.run_experiment(st_, cfg, lambda s, t, tp, te, L: ctrl.step(s, f32(tp), L=O.DEFAULT_PARAMS.L), **kw)

    out = run(L_steps=L_steps, m_pole_steps=m_steps, informer=S.controller_informer(inf_cfg))
    rows, calls = out["rows"], out["calls"]
    assert np.array_equal(rows["L"], g[f"{key}/col/L"]) and np.array_equal(rows["m_pole"], g[f"{key}/col/m_pole"])
    told = np.where(rows["informed"], "true", "default")
    assert np.array_equal(told, g[f"{key}/col/L_for_controller"]) and np.array_equal(told, g[f"{key}/col/m_pole_for_controller"])
    assert 0.2 < rows["informed"].mean() < 0.8 and np.count_nonzero(np.diff(rows["informed"].astype(int))) >= 4
    # what the controller calls were told: the initial values are handed over as the YAML's doubles, the true ones as the simulator's
    # float32 values - compared in float32, the precision a controller computes in (the t = 0 call is not told the mass, :866-878)
    assert np.array_equal(f32([c["L"] for c in calls]), f32(g[f"{key}/call/L"][1:]))
    assert np.array_equal(f32([c["m_pole"] for c in calls])[1:], f32(g[f"{key}/call/m_pole"][2:]))
    assert np.array_equal(rows["time"], g[f"{key}/col/time"]) and np.array_equal(rows["target_position"], g[f"{key}/col/target_position"])
    K = 12
    np.testing.assert_allclose(np.array([c["Q"] for c in calls])[:K], g[f"{key}/call/Q"][1:K + 1], atol=1e-4)
    np.testing.assert_allclose(np.array([c["s"] for c in calls])[:K], cs[1:K + 1], atol=1e-4, rtol=1e-4)
    r = (K - 1) * out["n_ctrl"] // out["n_save"]
    np.testing.assert_allclose(rows["angleDD"][:r], g[f"{key}/col/angleDD"][:r], atol=1e-3, rtol=1e-4)
    np.testing.assert_allclose(rows["positionDD"][:r], g[f"{key}/col/positionDD"][:r], atol=1e-3, rtol=1e-4)
    # ... with the pole MASS held at its initial value the plant does not follow the reference (the test has teeth)
    out2 = run(L_steps=L_steps)
    assert np.abs(out2["rows"]["angleDD"][:r] - g[f"{key}/col/angleDD"][:r]).max() > 0.02


@pytest.mark.parametrize("i", [0, 1])
def test_experiment_with_the_control_disturbance_switched_on(g, i):
    """exp_dist: the reference's simulator with controlDisturbance 0.3 and controlBias 0.05 (mode 'additive', as shipped), two
    experiments in a row on ONE module-level generator: per experiment two draws outside the run, then one per controller update.  The oracle's loop reproduces Q_applied from the recorded Q_calculated bit for bit (float32 arithmetic) and
    the closed loop under the disturbed control to 1e-4."""
    tag, key = "exp_dist", f"exp_dist/{i}"
    cfg = json.loads(g[f"{tag}/config"].item())
    d = json.loads(g[f"{tag}/disturbance"].item())
    es = S.ExperimentSetter(cfg)
    for j in range(i + 1):
        st = es.set(Generator(SFC64(int(g[f"{tag}/cartpole_seed0"]) + j)))
    n_upd = len(g[f"{key}/call/time"]) - 1                            # the t = 0 call + the updates inside the loop
    stream = Generator(SFC64(d["seed"]))
    z = None
    for j in range(i + 1):                # every experiment: 2 draws outside the run (set_cartpole_state_at_t0 when the controller is
        z = stream.standard_normal(size=2 + n_upd, dtype=f32)[2:]     # set, :792, and when the simulator is reset, :733), then its updates
    col = lambda name: g[f"{key}/col/{name}"]                         # noqa: E731
    # the arithmetic alone, from the recorded controls: Q_applied = (Q_calculated + 0.3 z) + 0.05 in float32
    per_call = col("Q_calculated")[::5]                               # (rows every 2 simulation steps, a controller update every 10)
    want = f32(f32(f32(per_call) + f32(0.3) * z) + f32(0.05))
    assert np.array_equal(want, col("Q_applied")[::5].astype(f32)) and np.abs(want - f32(per_call)).max() > 0.3
    assert np.array_equal(col("u").astype(f32), f32(1.77) * col("Q_applied").astype(f32))
    assert np.array_equal(col("Q_ccrc")[5:], col("Q_applied")[:-5]) and (col("Q_ccrc")[:5] == 0).all()
    # the loop
    N, H = int(g[f"{tag}/N"]), int(g[f"{tag}/H"])
    ctrl = O.LegacyMPPIController(int(g[f"{tag}/ctrl_seed"]) + i, N, H, SQRTRHOINV=0.02, p_Q=float(g[f"{tag}/p_Q"]))
    cs = g[f"{key}/call/s"]
    ctrl.step(cs[0], f32(g[f"{key}/call/tp"][0]), L=O.DEFAULT_PARAMS.L)
    out = S.run_experiment(st, cfg, lambda s, t, tp, te, L: ctrl.step(s, f32(tp), L=L), disturbance=(z, d["controlDisturbance"], d["controlBias"]))
    rows, calls = out["rows"], out["calls"]
    assert np.array_equal(rows["time"], col("time")) and np.array_equal(rows["target_position"], col("target_position"))
    K = 10
    np.testing.assert_allclose(np.array([c["Q"] for c in calls])[:K], g[f"{key}/call/Q"][1:K + 1], atol=1e-4)
    np.testing.assert_allclose(np.array([c["s"] for c in calls])[:K], cs[1:K + 1], atol=1e-4, rtol=1e-4)
    r = (K - 1) * out["n_ctrl"] // out["n_save"]
    np.testing.assert_allclose(rows["Q"][:r], col("Q_applied")[:r], atol=1e-4)
    np.testing.assert_allclose(rows["Q_calculated"][:r], col("Q_calculated")[:r], atol=1e-4)
    np.testing.assert_allclose(rows["Q_ccrc"][:r], col("Q_ccrc")[:r], atol=1e-4)
    np.testing.assert_allclose(rows["angleDD"][:r], col("angleDD")[:r], atol=2e-3, rtol=1e-4)
    # without the disturbance the loop does not follow (the test has teeth)
    ctrl2 = O.LegacyMPPIController(int(g[f"{tag}/ctrl_seed"]) + i, N, H, SQRTRHOINV=0.02, p_Q=float(g[f"{tag}/p_Q"]))
    ctrl2.step(cs[0], f32(g[f"{key}/call/tp"][0]), L=O.DEFAULT_PARAMS.L)
    es2 = S.ExperimentSetter(cfg)
    for j in range(i + 1):
        st2 = es2.set(Generator(SFC64(int(g[f"{tag}/cartpole_seed0"]) + j)))
    out2 = S.run_experiment(st2, cfg, lambda s, t, tp, te, L: ctrl2.step(s, f32(tp), L=L))
    assert np.abs(out2["rows"]["angleDD"][:r] - col("angleDD")[:r]).max() > 0.5


def _sensor_chain(g, tag="exp_sensor"):
    sen = json.loads(g[f"{tag}/sensor"].item())
    n = sen["noise"]
    return S.MeasurementChain(sen["latency"], 0.002,
                              noise=(Generator(SFC64(n["seed"])), n["sigma_angle"], n["sigma_position"], n["sigma_angleD"], n["sigma_positionD"]),
                              offset_updater=S.parameter_updater(sen["vertical_angle_offset"]),
                              offset_init_deg=sen["vertical_angle_offset"]["init_value"])


def test_experiment_with_the_measurement_chain_switched_on(g):
    """exp_sensor: the reference's simulator with 5 ms of latency (2.5 simulation steps: the delayed state is interpolated),
    measurement noise from its seeded generator, a vertical-angle offset in 'bounce' mode and a switching informer (informed, the
    controller gets the offset taken out again).  The oracle's MeasurementChain reproduces the float64 state every controller call
    was handed to 1e-6 over the first calls (float32 against float64 controller arithmetic in the loop: 9e-8 measured), the
    recording's vertical_angle_offset columns exactly; without the chain the loop is somewhere else entirely."""
    tag, key = "exp_sensor", "exp_sensor/0"
    cfg = json.loads(g[f"{tag}/config"].item())
    inf = json.loads(g[f"{tag}/informer"].item())
    N, H = int(g[f"{tag}/N"]), int(g[f"{tag}/H"])
    cs = g[f"{key}/call/s"]

    def run(**kw):
        st = S.ExperimentSetter(cfg).set(Generator(SFC64(int(g[f"{tag}/cartpole_seed0"]))))
        ctrl = O.LegacyMPPIController(int(g[f"{tag}/ctrl_seed"]), N, H, SQRTRHOINV=0.02, p_Q=float(g[f"{tag}/p_Q"]))
        ctrl.step(cs[0], f32(g[f"{key}/call/tp"][0]), L=O.DEFAULT_PARAMS.L)
        return S.run_experiment(st, cfg, lambda s, t, tp, te, L: ctrl.step(np.asarray(s, f32), f32(tp), L=O.DEFAULT_PARAMS.L),
                                informer=S.controller_informer(inf), **kw)

    out = run(sensor=_sensor_chain(g))
    rows, calls = out["rows"], out["calls"]
    col = lambda name: g[f"{key}/col/{name}"]                         # noqa: E731
    off = col("vertical_angle_offset")
    assert np.array_equal(rows["vertical_angle_offset"], off) and len(np.unique(off)) > 8 and off[0] == np.deg2rad(2.0)
    assert np.array_equal(np.cos(off), col("vertical_angle_offset_cos")) and np.array_equal(np.sin(off), col("vertical_angle_offset_sin"))
    assert np.array_equal(np.where(rows["informed"], "true", "default"), col("L_for_controller"))
    K = 12
    so, sf = np.array([c["s"] for c in calls]), g[f"{key}/call/s64"][1:]
    assert so.dtype == np.float64 and np.abs(so[:K] - sf[:K]).max() < 1e-6
    # the chain matters: the handed-over state is far from the true one (latency on a moving pole, noise on the velocities, 2-6 degrees
    # of offset while the controller is not informed)
    true_at_calls = np.stack([col(n)[::5] for n in ("angle", "angleD", "angle_cos", "angle_sin", "position", "positionD")], axis=-1)
    assert np.abs(sf[1:K] - true_at_calls[1:K]).max() > 0.03
    np.testing.assert_allclose(np.array([c["Q"] for c in calls])[:K], g[f"{key}/call/Q"][1:K + 1], atol=1e-5)
    r = (K - 1) * out["n_ctrl"] // out["n_save"]
    np.testing.assert_allclose(rows["s"][:r, 0], col("angle")[:r], atol=1e-5)
    out0 = run()
    assert np.abs(np.array([c["Q"] for c in out0["calls"]])[:K] - g[f"{key}/call/Q"][1:K + 1]).max() > 0.01


def test_experiment_that_ends_inside_a_control_period(g):
    """exp_tail: 25 simulation steps = two control periods and five trailing steps, no turning points (length x complexity < 1: the
    target is 0 whatever the start, random_target_generator.py:31-33): controller calls at t = 0, 0.02, 0.04, three saved rows."""
    key = "exp_tail/0"
    cfg = json.loads(g["exp_tail/config"].item())
    st = S.ExperimentSetter(cfg).set(Generator(SFC64(int(g["exp_tail/cartpole_seed0"]))))
    ctrl = O.LegacyMPPIController(int(g["exp_tail/ctrl_seed"]), int(g["exp_tail/N"]), int(g["exp_tail/H"]), SQRTRHOINV=0.02, p_Q=0.0)
    cs = g[f"{key}/call/s"]
    ctrl.step(cs[0], f32(g[f"{key}/call/tp"][0]), L=O.DEFAULT_PARAMS.L)
    out = S.run_experiment(st, cfg, lambda s, t, tp, te, L: ctrl.step(s, f32(tp), L=L))
    assert len(out["times"]) == 26 and len(out["calls"]) == 3 and len(out["rows"]["time"]) == 3
    assert np.array_equal(out["rows"]["time"], g[f"{key}/col/time"]) and np.array_equal(out["rows"]["target_position"], np.zeros(3))
    assert st["s0"][O.POSITION_IDX] != 0.0 and np.array_equal(st["s0"], cs[1])
    np.testing.assert_allclose(np.array([c["Q"] for c in out["calls"]]), g[f"{key}/call/Q"][1:], atol=1e-4)
    np.testing.assert_allclose(out["rows"]["s"], np.stack([g[f"{key}/col/{n}"] for n in ("angle", "angleD", "angle_cos", "angle_sin", "position", "positionD")], axis=-1),
                               atol=1e-4, rtol=1e-4)
