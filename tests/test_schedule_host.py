"""CPU: the product's host-side experiment schedule (cartpolesimulation_amd/schedule.py: all experiments of a batch tabulated at
once) against the fixture made by the reference's own code (tests/golden/schedule.npz) and against the oracle's per-experiment
restatement (oracle/schedule_np.py)."""
import json
import os

import numpy as np
import pytest
from numpy.random import SFC64, Generator

from cartpolesimulation_amd import schedule as SC
from oracle import schedule_np as S


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "schedule.npz"))


def test_trace_evaluation_matches_the_reference_for_every_case(g):
    for c in json.loads(g["trace/cases"].item()):
        tk, yk = SC.turning_points(c["length"], Generator(SFC64(c["seed"])), c["complexity"], c["turning_points"], c["period"], c["start"],
                                   c["end"], c["used_fraction"], S.THL32)
        hi = np.float64(np.float32(c["used_fraction"]) * S.THL32)
        y = np.clip(SC.evaluate_trace(tk, yk, c["interpolation"], g[f"trace/{c['name']}/t"]), -hi, hi)
        assert np.array_equal(y, g[f"trace/{c['name']}/y"]), c["name"]


@pytest.mark.parametrize("tag", ["setter_shipped", "setter_alt"])
def test_batch_draw_equals_consecutive_reference_experiments(g, tag):
    """RandomExperimentSetter.draw(K) = K consecutive random_experiment_setter.set calls of the reference: initial states, the
    alternation of interpolation types, and every experiment's target position at every simulation step - the same doubles."""
    cfg = json.loads(g[f"{tag}/config"].item())
    cfg["dt"]["saving"] = cfg["dt"]["simulation"]                     # stride 1: every simulation step is a table row
    b = SC.RandomExperimentSetter(cfg).draw(g[f"{tag}/s0"].shape[0], int(g[f"{tag}/cartpole_seed0"]))
    n = g[f"{tag}/target_position"].shape[1]
    assert b.stride == 1 and b.target_position.shape[0] == b.n_sim + 1
    assert np.array_equal(b.s0, g[f"{tag}/s0"])
    assert np.array_equal(b.target_position[:n].T, g[f"{tag}/target_position"])
    assert np.array_equal(b.target_position[n:], np.broadcast_to(b.target_position[n - 1], b.target_position[n:].shape))   # held at the end
    assert b.interpolation_type == list(g[f"{tag}/interpolation_type"])
    assert np.array_equal(b.target_equilibrium[0], g[f"{tag}/target_equilibrium"])


@pytest.mark.parametrize("tag,K", [("exp_fine", 2), ("exp_coarse", 1), ("exp_device", 2), ("exp_tail", 1)])
def test_tables_equal_what_the_simulator_handed_its_controller_and_logged(g, tag, K):
    """The rows the device loop will read - at the controller's instants and at the saved rows - against the REAL simulator class:
    target position and equilibrium handed to every controller call, the recording's time / target columns, the initial state."""
    cfg = json.loads(g[f"{tag}/config"].item())
    b = SC.RandomExperimentSetter(cfg).draw(K, int(g[f"{tag}/cartpole_seed0"]))
    from math import gcd
    assert b.stride == gcd(b.n_ctrl, b.n_save) and b.n_periods == b.n_sim // b.n_ctrl
    ctrl_steps, save_steps = np.arange(0, b.n_sim + 1, b.n_ctrl), np.arange(0, b.n_sim + 1, b.n_save)
    for i in range(K):
        key = f"{tag}/{i}"
        assert np.array_equal(b.s0[i], g[f"{key}/call/s"][1])
        assert np.array_equal(b.target_position[b.rows_at(ctrl_steps), i], g[f"{key}/call/tp"][1:])
        assert np.array_equal(b.target_equilibrium[b.rows_at(ctrl_steps), i], g[f"{key}/call/te"][1:])
        assert np.array_equal(b.times[ctrl_steps], g[f"{key}/call/time"][1:])
        assert np.array_equal(b.target_position[b.rows_at(save_steps), i], g[f"{key}/col/target_position"])
        assert np.array_equal(b.target_equilibrium[b.rows_at(save_steps), i], g[f"{key}/col/target_equilibrium"])
        assert np.array_equal(b.times[save_steps], g[f"{key}/col/time"])
        assert b.interpolation_type[i] == str(g[f"{key}/interpolation_type"])


def test_equilibrium_table_and_oracle_agree_on_the_shipped_dwell_times():
    """keep_target_equilibrium_x_seconds_up = 10, _down = 2.5 (config_data_gen.yml:23-24), 30 s, both initial sides, against the
    oracle's step-by-step rule (CartPole/__init__.py:380-388); 'inf' never flips."""
    cfg = SC.merged_config(dict(seed=5, length_of_experiment=30.0, initial_target_equilibrium="down"))
    b = SC.RandomExperimentSetter(cfg).draw(3, 11)
    times, _, te = S.schedule_tables(lambda t: 0.0, -1, 30.0, 0.002, 10, 2.5)
    assert np.array_equal(times, b.times)
    assert np.array_equal(b.target_equilibrium[:, 0], te[::b.stride]) and len(np.flatnonzero(np.diff(te))) == 5   # 2.5 s, 12.5 s, 15 s, 25 s, 27.5 s
    b2 = SC.RandomExperimentSetter(SC.merged_config(dict(seed=5, length_of_experiment=3.0, keep_target_equilibrium_x_seconds_up="inf"))).draw(2, 1)
    assert (b2.target_equilibrium == 1).all()
    with pytest.raises(ValueError):
        SC.RandomExperimentSetter(dict(length_of_experiment=1.0))     # no seed: the reference would seed from the clock


def test_batch_is_reproducible_and_experiments_differ():
    cfg = dict(seed=9, length_of_experiment=4.0)
    a = SC.RandomExperimentSetter(cfg).draw(8, 100)
    b = SC.RandomExperimentSetter(cfg).draw(8, 100)
    assert np.array_equal(a.s0, b.s0) and np.array_equal(a.target_position, b.target_position)
    assert len({tuple(r) for r in a.s0}) == 8 and a.interpolation_type == ["previous", "0-derivative-smooth"] * 4
    assert np.abs(a.target_position).max() <= float(S.THL32)
    # a run drawn in two halves continues the same random streams (the setter is ONE object for all experiments of a run)
    rs = SC.RandomExperimentSetter(cfg)
    h1, h2 = rs.draw(4, 100), rs.draw(4, 104)
    assert np.array_equal(np.concatenate([h1.s0, h2.s0]), a.s0)
    assert np.array_equal(np.concatenate([h1.target_position, h2.target_position], axis=1), a.target_position)


def test_parameter_table_reproduces_the_references_updater(g):
    """schedule.parameter_table against the pole length the reference's simulator held after every update_parameters call
    (ParameterUpdater in 'bounce' mode, float32 arithmetic, clip), and the other modes' basic behaviour."""
    u = json.loads(g["exp_varL/L_updater"].item())
    times = SC.accumulated_times(200, 0.002)
    tab = SC.parameter_table(u, times)
    assert tab.dtype == np.float32 and np.array_equal(tab[1:].astype(np.float64), g["exp_varL/0/L_steps"]) and tab[0] == np.float32(0.395)
    const = SC.parameter_table(dict(u, mode="constant"), times)
    assert (const == np.float32(0.395)).all()
    inc = SC.parameter_table(dict(u, mode="increase", increment=0.001, range_clip=[0.2, 0.5]), times)
    assert (np.diff(inc) >= 0).all() and inc[-1] > inc[0] and len(np.unique(inc)) == 200 // 7 + 1
    import random
    rw = SC.parameter_table(dict(u, mode="random walk", change_every_x_seconds=None, increment=0.002, range_clip=[0.2, 0.5]), times,
                            py_random=random.Random(3))
    assert {float(x) for x in np.round(np.abs(np.diff(rw)).astype(np.float64), 4)} == {0.002}                    # empty change_every: a step on every simulation step
    rnd = SC.parameter_table(dict(u, mode="random"), times, np_random=np.random.RandomState(4))
    assert rnd.min() >= 0.2 and rnd.max() <= 0.5 and len(np.unique(rnd)) == 200 // 7 + 1
    rst = SC.parameter_table(dict(u, mode="increase", increment=0.001, range_clip=[0.2, 0.5], reset_every_x_seconds=0.1), times)
    assert (rst == np.float32(0.395)).sum() > 8                                 # back to the initial value every 0.1 s


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_run_is_the_single_process_run(world):
    """One process per GPU, each drawing its block of the run's experiments (schedule.draw_shard): the union over the ranks equals
    the single-process run, experiment for experiment - uneven blocks included (7 experiments over 2 / 3 ranks)."""
    cfg = dict(seed=12, length_of_experiment=3.0, start_at_target=False, target_position_end=None)
    whole = SC.RandomExperimentSetter(cfg).draw(7, 300, L=np.linspace(0.3, 0.45, 7))
    parts = [SC.draw_shard(cfg, 7, 300, rank=r, world=world, L=np.linspace(0.3, 0.45, 7)) for r in range(world)]
    assert [first for _, first in parts] == [sum(p.E for p, _ in parts[:r]) for r in range(world)] and sum(p.E for p, _ in parts) == 7
    assert np.array_equal(np.concatenate([p.s0 for p, _ in parts]), whole.s0)
    assert np.array_equal(np.concatenate([p.target_position for p, _ in parts], axis=1), whole.target_position)
    assert np.array_equal(np.concatenate([p.target_equilibrium for p, _ in parts], axis=1), whole.target_equilibrium)
    assert sum((p.interpolation_type for p, _ in parts), []) == whole.interpolation_type
    assert np.array_equal(np.concatenate([p.L for p, _ in parts]), whole.L)


def test_pole_mass_table_and_informer_table_reproduce_the_references(g):
    """exp_varM: the simulator's `m_pole:` updater ('bounce', a change every 11 simulation steps) and its controller informer in
    'switching_regular' mode.  schedule.parameter_table gives the pole mass the simulator held after every update_parameters call,
    schedule.informer_table the answer its informer gave - the recording's *_for_controller columns and what every controller call
    was handed (the true pole length or the initial one)."""
    from oracle import schedule_np as SN
    cfg = json.loads(g["exp_varM/config"].item())
    n = int(np.ceil(cfg["length_of_experiment"] / cfg["dt"]["simulation"]))
    n_ctrl, n_save = int(np.rint(cfg["dt"]["control"] / cfg["dt"]["simulation"])), int(np.rint(cfg["dt"]["saving"] / cfg["dt"]["simulation"]))
    times = SC.accumulated_times(n, cfg["dt"]["simulation"])
    m = SC.parameter_table(json.loads(g["exp_varM/m_pole_updater"].item()), times)
    assert m.dtype == np.float32 and np.array_equal(m[1:].astype(np.float64), g["exp_varM/0/m_pole_steps"]) and m[0] == np.float32(0.087)
    L = SC.parameter_table(json.loads(g["exp_varM/L_updater"].item()), times)
    assert np.array_equal(L[1:].astype(np.float64), g["exp_varM/0/L_steps"])
    inf = json.loads(g["exp_varM/informer"].item())
    told = SC.informer_table(inf, times, n_ctrl)
    assert told.shape == (n + 1,) and not told[0]
    assert np.array_equal(np.where(told[::n_save], "true", "default"), g["exp_varM/0/col/L_for_controller"])
    calls = np.arange(0, n + 1, n_ctrl)
    L_told = np.where(told[calls], L[calls], L[0])                                     # the float32 the controller computes with
    assert np.array_equal(L_told, g["exp_varM/0/call/L"][1:].astype(np.float32))
    assert np.array_equal(np.where(told[calls], m[calls], m[0])[1:], g["exp_varM/0/call/m_pole"][2:].astype(np.float32))
    # the other modes; 'switching_random' against the checker's restatement of the class on the same random stream
    assert SC.informer_table(dict(inf, mode="ON"), times, n_ctrl).all() and not SC.informer_table(dict(inf, mode="OFF"), times, n_ctrl).any()
    rnd = dict(inf, mode="switching_random", change_to_on_after_x_seconds_off=0.09, change_to_off_after_x_seconds_on=0.12)
    got = SC.informer_table(rnd, times, n_ctrl, np_random=np.random.RandomState(5))
    ref = SN.controller_informer(rnd, np_random=np.random.RandomState(5))
    want, cur = np.empty(n + 1, bool), False
    for k in range(n + 1):
        if k % n_ctrl == 0:
            cur = ref(times[k])
        want[k] = cur
    assert np.array_equal(got, want) and 2 <= np.count_nonzero(np.diff(got.astype(int)))
    with pytest.raises(ValueError):
        SC.informer_table(dict(inf, mode="sometimes"), times, n_ctrl)


def test_a_batch_drawn_at_stride_one_carries_the_parameter_schedule():
    """draw(stride=1) tabulates the same experiments per simulation step (the default stride's rows are every gcd-th of them);
    apply_parameter_schedule attaches the simulator's parameter updaters and informer, one column per experiment, and the random
    modes do not depend on how the run is split over processes."""
    cfg = dict(seed=9, length_of_experiment=0.5, dt=dict(saving=0.008))
    a = SC.RandomExperimentSetter(cfg).draw(3, 40)
    b = SC.RandomExperimentSetter(cfg).draw(3, 40, stride=1)
    assert a.stride == 2 and b.stride == 1 and np.array_equal(a.s0, b.s0)
    assert np.array_equal(a.target_position, b.target_position[::2]) and np.array_equal(a.target_equilibrium, b.target_equilibrium[::2])
    with pytest.raises(ValueError):
        SC.RandomExperimentSetter(cfg).draw(3, 40, stride=4)
    prm = dict(L=dict(init_value=0.395, change_every_x_seconds=0.01, mode="increase", range_random=[0.2, 0.5], range_clip=[0.2, 0.5],
                      increment=0.001, reset_every_x_seconds=0.2),
               m_pole=dict(init_value="random", change_every_x_seconds=0.05, mode="random", range_random=[0.015, 0.15], range_clip=None,
                           increment=0.002, reset_every_x_seconds="inf"),
               inform_controller_about_parameters_change=dict(mode="switching_random", change_to_on_after_x_seconds_off=0.1,
                                                              change_to_off_after_x_seconds_on=0.1))
    with pytest.raises(ValueError):
        SC.apply_parameter_schedule(a, prm)
    full = SC.apply_parameter_schedule(b, prm, seed=5)
    n = b.n_sim + 1
    assert full.L_table.shape == full.m_pole_table.shape == full.informed.shape == (n, 3)
    assert np.array_equal(full.L_table[:, 0], full.L_table[:, 2]) and np.array_equal(full.L_table[:, 0], SC.parameter_table(prm["L"], b.times))
    assert full.L_table[:, 0].max() > 0.4 and np.count_nonzero(np.diff(full.L_table[:, 0]) < 0) == 2        # grows, reset every 0.2 s
    assert not np.array_equal(full.m_pole_table[:, 0], full.m_pole_table[:, 1]) and not np.array_equal(full.informed[:, 0], full.informed[:, 1])
    assert full.m_pole_table.min() >= np.float32(0.015) and full.m_pole_table.max() <= np.float32(0.15)
    tail, first = SC.draw_shard(cfg, 3, 40, rank=1, world=2, stride=1)
    part = SC.apply_parameter_schedule(tail, prm, seed=5, first=first)
    assert first == 2 and np.array_equal(part.m_pole_table[:, 0], full.m_pole_table[:, 2]) and np.array_equal(part.informed[:, 0], full.informed[:, 2])


def test_control_disturbance_draws_are_the_references(g):
    """exp_dist: two experiments of the reference's simulator in a row with controlDisturbance 0.3, controlBias 0.05 on ONE seeded
    module-level generator.  schedule.control_disturbance gives every experiment's draws - also to a process that owns only the second
    experiment - and the float32 arithmetic of apply_parameter_schedule + recording_block's form of it reproduces the recorded
    Q_applied from the recorded Q_calculated bit for bit."""
    d = json.loads(g["exp_dist/disturbance"].item())
    cfg = json.loads(g["exp_dist/config"].item())
    n_calls = len(g["exp_dist/0/call/time"]) - 1
    z = SC.control_disturbance(2, n_calls, d["seed"])
    assert z.shape == (n_calls, 2) and z.dtype == np.float32
    assert np.array_equal(SC.control_disturbance(1, n_calls, d["seed"], first=1)[:, 0], z[:, 1])
    b = SC.RandomExperimentSetter(cfg).draw(2, int(g["exp_dist/cartpole_seed0"]))
    b = SC.apply_parameter_schedule(b, dict(controlDisturbance=d["controlDisturbance"], controlBias=d["controlBias"], seed=d["seed"]))
    assert b.Q_disturbance.shape == (b.n_periods + 1, 2) and b.Q_bias == float(np.float32(0.05)) and b.L_table is None
    for i in range(2):
        qc = g[f"exp_dist/{i}/col/Q_calculated"][::5].astype(np.float32)
        qa = ((qc + b.Q_disturbance[:, i]).astype(np.float32) + np.float32(b.Q_bias)).astype(np.float32)
        assert np.array_equal(qa, g[f"exp_dist/{i}/col/Q_applied"][::5].astype(np.float32))
    with pytest.raises(ValueError):
        SC.apply_parameter_schedule(b, dict(controlDisturbance=0.2))                    # no seed: the reference would use the clock
    with pytest.raises(NotImplementedError):
        SC.apply_parameter_schedule(b, dict(controlDisturbance=0.2, seed=1, controlDisturbance_mode="truncnorm"))
    off = SC.apply_parameter_schedule(b, dict(controlDisturbance=0.2, seed=1, controlDisturbance_mode="OFF"))
    assert off.Q_disturbance is b.Q_disturbance                                          # (nothing added)


def test_active_parameters_of_a_physical_parameters_file():
    """schedule.active_parameters: which blocks of cartpole_physical_parameters.yml's `cartpole:` section change a run.  The shipped
    file (read from the reference checkout when it is mounted) has none: updaters 'constant', informer 'ON', disturbance amplitude 0."""
    import yaml
    ref = "/root/reference/cartpole_physical_parameters.yml"
    if os.path.isfile(ref):
        with open(ref) as fh:
            assert SC.active_parameters(yaml.safe_load(fh)["cartpole"]) is None
    sec = dict(seed=7, controlDisturbance_mode="additive", controlDisturbance=0.0, controlBias=0.0,
               L=dict(init_value=0.395, mode="constant"), m_pole=dict(init_value=0.087, mode="constant"),
               inform_controller_about_parameters_change=dict(mode="ON"))
    assert SC.active_parameters(sec) is None
    got = SC.active_parameters(dict(sec, controlDisturbance=0.3, L=dict(init_value="random", mode="constant"),
                                    m_pole=dict(init_value=0.087, mode="bounce"),
                                    inform_controller_about_parameters_change=dict(mode="OFF")))
    assert set(got) == {"L", "m_pole", "inform_controller_about_parameters_change", "controlDisturbance_mode", "controlDisturbance",
                        "controlBias", "seed"} and got["seed"] == 7 and got["controlDisturbance"] == 0.3
    assert SC.active_parameters(dict(sec, controlDisturbance=0.3, controlDisturbance_mode="OFF")) is None


def test_measurement_chain_tables_follow_the_references(g):
    """exp_sensor: the vertical-angle-offset updater (a float64, updated with the time AFTER the step, started from deg2rad(init)) and
    the measurement-noise draws (four float32 normals per simulation step from SFC64(seed), scaled in float32; the rows of the steps
    that end a control period) as apply_parameter_schedule tabulates them."""
    sen = json.loads(g["exp_sensor/sensor"].item())
    cfg = json.loads(g["exp_sensor/config"].item())
    inf = json.loads(g["exp_sensor/informer"].item())
    b = SC.RandomExperimentSetter(cfg).draw(2, int(g["exp_sensor/cartpole_seed0"]), stride=1)
    prm = dict(latency=sen["latency"], noise=dict(sen["noise"], noise_mode="ON"), vertical_angle_offset=sen["vertical_angle_offset"],
               inform_controller_about_parameters_change=inf, seed=sen["noise"]["seed"])
    full = SC.apply_parameter_schedule(b, prm)
    assert full.latency == 0.005 and full.angle_offset.dtype == np.float64 and full.angle_offset.shape == (b.n_sim + 1, 2)
    assert np.array_equal(full.angle_offset[::2, 0], g["exp_sensor/0/col/vertical_angle_offset"])
    assert np.array_equal(full.angle_offset[:, 0], full.angle_offset[:, 1])
    nz = full.measurement_noise
    assert nz.shape == (b.n_periods + 1, 2, 4) and nz.dtype == np.float32 and not nz[0].any() and np.array_equal(nz[:, 0], nz[:, 1])
    z = np.random.Generator(np.random.SFC64(sen["noise"]["seed"])).standard_normal(size=(b.n_sim, 4), dtype=np.float32)
    n = sen["noise"]
    want = z[9::10] * np.array([n["sigma_angle"], n["sigma_position"], n["sigma_angleD"], n["sigma_positionD"]], np.float32)
    assert np.array_equal(nz[1:, 0], want)
    # what the file's section would switch on
    sec = dict(seed=5, latency=0.005, noise=dict(noise_mode="ON", sigma_angle=0.0, sigma_position=0.0005, sigma_angleD=0.075, sigma_positionD=0.005),
               vertical_angle_offset=dict(init_value=2.0, mode="constant"), controlDisturbance_mode="additive", controlDisturbance=0.0, controlBias=0.0)
    assert set(SC.active_parameters(sec)) == {"latency", "noise", "seed", "vertical_angle_offset"}
    with pytest.raises(ValueError):
        SC.apply_parameter_schedule(b, dict(latency=0.5))                               # more than the reference's buffer holds
    with pytest.raises(ValueError):
        SC.apply_parameter_schedule(b, dict(noise=dict(sen["noise"], noise_mode="ON")))  # no seed


def test_parameter_tables_for_many_experiments_equal_the_single_experiment_form():
    """schedule.parameter_tables (every experiment's column at once: the change schedule is a function of time only) against
    parameter_table column by column - all six modes, 'random' and given initial values, a change at every step and every few steps,
    resets, the float32 form (L, m_pole) and the float64 form updated after the step (the vertical angle offset)."""
    import random
    times = SC.accumulated_times(900, 0.002)
    base = dict(init_value=0.1, change_every_x_seconds=0.014, range_random=[0.05, 0.3], range_clip=[0.06, 0.25], increment=0.01,
                reset_every_x_seconds=0.5)
    seeds = [5, 6, 9]
    for mode in ("constant", "increase", "bounce", "random walk", "random", "random_gaussian"):
        for init in (0.1, "random"):
            for every in (0.014, None):
                for kw in (dict(dtype=np.float32), dict(dtype=np.float64, init=0.02, time_after_step=True)):
                    u = dict(base, mode=mode, init_value=init, change_every_x_seconds=every)
                    tabs = SC.parameter_tables(u, times, seeds, **kw)
                    assert tabs.shape == (len(times), 3) and tabs.dtype == kw["dtype"]
                    for e, sd in enumerate(seeds):
                        col = SC.parameter_table(u, times, random.Random(sd), np.random.RandomState(sd), **kw)
                        assert np.array_equal(col, tabs[:, e]), (mode, init, every, kw, e)
    with pytest.raises(ValueError):
        SC.parameter_tables(dict(base, mode="sideways"), times, seeds)
