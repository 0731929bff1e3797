"""GPU: the data generator's experiment schedule in the device loop: cpmppi_plant_step reading the schedule
tables and recording at the saving period, harness.run_schedule, against the oracle's experiment loop and against whole experiments
the reference's own simulator class ran (tests/golden/schedule.npz: moving target, target-equilibrium flips, dt_save != dt_control)."""
import dataclasses
import json
import os

import numpy as np
import pytest
from numpy.random import SFC64, Generator

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle_np as O  # noqa: E402
from oracle import schedule_np as S  # noqa: E402

f32 = np.float32


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "schedule.npz"))


def test_plant_step_follows_the_oracle_loop_row_by_row():
    """cpmppi_plant_step under a GIVEN control sequence (no controller): states and second derivatives at every saved row, the
    controls' log, the values published for the next controller call and a pole length that changes DURING a control period, against
    the oracle's update_state loop (oracle/schedule_np.py) - for dt_save below, equal to and above dt_control."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    E, T, n_ctrl = 6, 7, 10
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=64, mpc_horizon=10))
    rng = Generator(SFC64(21))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-3, 3), rng.uniform(-4, 4), rng.uniform(-0.19, 0.19), rng.uniform(-0.6, 0.6))
                   for _ in range(E)])
    s0[:, 2], s0[:, 3] = np.cos(s0[:, 0]), np.sin(s0[:, 0])
    Qs = rng.uniform(-1, 1, (T + 1, E)).astype(f32)
    n_sim = T * n_ctrl
    for n_save in (4, 10, 20, 1):
        from math import gcd
        stride = 1                                                     # (a pole-length table is per simulation step)
        rows_sched = n_sim + 1
        tp = rng.uniform(-0.15, 0.15, (rows_sched, E)).astype(f32)
        te = rng.choice([-1.0, 1.0], (rows_sched, E)).astype(f32)
        Ltab = np.repeat(rng.uniform(0.2, 0.5, (n_sim // 7 + 1, E)).astype(f32), 7, axis=0)[:rows_sched]   # changes every 7 steps
        mtab = np.repeat(rng.uniform(0.03, 0.15, (n_sim // 11 + 1, E)).astype(f32), 11, axis=0)[:rows_sched]  # the pole MASS: every 11
        Lctab = rng.uniform(0.2, 0.5, (rows_sched, E)).astype(f32)    # what the controller is told instead of the true length
        qd, qb = rng.normal(0, 0.3, (T + 1, E)).astype(f32), f32(0.05)  # the simulator's additive control disturbance
        R = n_sim // n_save + 1
        s = eng.tensor(s0.copy())
        states, dd, Qlog = eng.zeros(R, E, 6), eng.zeros(R, E, 2), eng.zeros(T + 1, E)
        states[0] = s
        tp_d, te_d, L_d = eng.tensor(tp), eng.tensor(te), eng.tensor(Ltab)
        cur_tp, cur_te, cur_L = eng.zeros(E), eng.zeros(E), eng.zeros(E)
        told = n_save in (4, 1)                                        # with and without a separate controller-side table
        kw = dict(dt_sim=0.002, period_steps=n_ctrl, states_log=states, dd_log=dd, save_every=n_save, Q_log=Qlog,
                  target_position_table=tp_d, target_equilibrium_table=te_d, L_table=L_d, sched_stride=stride,
                  target_position_out=cur_tp, target_equilibrium_out=cur_te, L_out=cur_L, m_pole_table=eng.tensor(mtab),
                  L_controller_table=eng.tensor(Lctab) if told else None, Q_disturbance_table=eng.tensor(qd), Q_bias=float(qb))
        Qa = ((Qs + qd).astype(f32) + qb).astype(f32)                  # what drives the plant; the log keeps the calculated control
        for c in range(T):
            eng.plant_step(s, Qs[c], n_ctrl, period=c, **kw)
            g1 = (c + 1) * n_ctrl
            assert np.array_equal(cur_tp.cpu().numpy(), tp[g1]) and np.array_equal(cur_te.cpu().numpy(), te[g1])
            assert np.array_equal(cur_L.cpu().numpy(), (Lctab if told else Ltab)[g1])
        eng.plant_step(s, Qs[T], 0, period=T, **kw)                    # the run's last controller call: record only
        st_h, dd_h = states.cpu().numpy(), dd.cpu().numpy()
        assert np.array_equal(Qlog.cpu().numpy(), Qs)
        assert np.array_equal(st_h[-1], s.cpu().numpy()) if n_sim % n_save == 0 else True
        for e in range(E):
            r = s0[e].copy()
            k = 0
            Q = Qa[0, e]
            pm = lambda gs: dataclasses.replace(O.DEFAULT_PARAMS, m_pole=mtab[gs, e])   # noqa: E731
            add, pdd = O.plant_ode(r, Q, Ltab[0, e], pm(0))
            rows = [(r.copy(), add, pdd)]
            for gstep in range(1, n_sim + 1):                          # update_state: L and m_pole first, integrate, (controller), ode, save
                L = Ltab[gstep, e]
                r = O.plant_substep(r, add, pdd, 0.002, L, pm(gstep))
                if gstep % n_ctrl == 0:
                    Q = Qa[gstep // n_ctrl, e]
                add, pdd = O.plant_ode(r, Q, L, pm(gstep))
                if gstep % n_save == 0:
                    rows.append((r.copy(), add, pdd))
            assert len(rows) == R
            for i, (rs, a_, p_) in enumerate(rows):
                assert np.all(np.abs(st_h[i, e] - rs) <= 3e-5 + 3e-5 * np.abs(rs)), (n_save, e, i, st_h[i, e], rs)
                assert abs(dd_h[i, e, 0] - a_) <= 2e-3 + 1e-4 * abs(a_) and abs(dd_h[i, e, 1] - p_) <= 5e-4 + 1e-4 * abs(p_), (n_save, e, i)
    eng.close()


@pytest.mark.parametrize("n_ctrl,n_save,stride", [(10, 4, 2), (4, 6, 2), (1, 1, 1), (25, 10, 5), (10, 20, 10), (7, 3, 1)])
def test_plant_step_tables_at_any_stride_and_time_scale(n_ctrl, n_save, stride):
    """cpmppi_plant_step with schedule tables sampled every `stride` simulation steps (the gcd of control and saving periods, as
    schedule.py builds them), control periods of 1 ... 25 simulation steps, saving periods that divide them or not: the published
    next-call values, the rows that are saved (and only those), the plant - against the oracle loop."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    E, T = 4, 5
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=64, mpc_horizon=10))
    rng = Generator(SFC64(100 * n_ctrl + n_save))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-3, 3), rng.uniform(-4, 4), rng.uniform(-0.19, 0.19), rng.uniform(-0.6, 0.6))
                   for _ in range(E)])
    Qs = rng.uniform(-1, 1, (T + 1, E)).astype(f32)
    Lv = rng.uniform(0.25, 0.45, E).astype(f32)
    n_sim = T * n_ctrl
    srows = n_sim // stride + 1
    tp = rng.uniform(-0.15, 0.15, (srows, E)).astype(f32)
    te = rng.choice([-1.0, 1.0], (srows, E)).astype(f32)
    R = n_sim // n_save + 1
    s = eng.tensor(s0.copy())
    guard = 3.0
    states, dd, Qlog = eng.zeros(R + 2, E, 6) + guard, eng.zeros(R + 2, E, 2) + guard, eng.zeros(T + 1, E)
    states[0] = s
    cur_tp, cur_te = eng.zeros(E), eng.zeros(E)
    kw = dict(dt_sim=0.002, period_steps=n_ctrl, L=Lv, states_log=states[:R], dd_log=dd[:R], save_every=n_save, Q_log=Qlog,
              target_position_table=eng.tensor(tp), target_equilibrium_table=eng.tensor(te), sched_stride=stride,
              target_position_out=cur_tp, target_equilibrium_out=cur_te)
    for c in range(T):
        eng.plant_step(s, Qs[c], n_ctrl, period=c, **kw)
        row = min((c + 1) * n_ctrl // stride, srows - 1)
        assert np.array_equal(cur_tp.cpu().numpy(), tp[row]) and np.array_equal(cur_te.cpu().numpy(), te[row])
    eng.plant_step(s, Qs[T], 0, period=T, **kw)
    st_h, dd_h = states.cpu().numpy(), dd.cpu().numpy()
    assert (st_h[R:] == guard).all() and (dd_h[R:] == guard).all()                  # nothing beyond the logs' rows
    for e in range(E):
        r, Q = s0[e].copy(), Qs[0, e]
        add, pdd = O.plant_ode(r, Q, Lv[e])
        rows = [(r.copy(), add, pdd)]
        for gstep in range(1, n_sim + 1):
            r = O.plant_substep(r, add, pdd, 0.002, Lv[e])
            if gstep % n_ctrl == 0:
                Q = Qs[gstep // n_ctrl, e]
            add, pdd = O.plant_ode(r, Q, Lv[e])
            if gstep % n_save == 0:
                rows.append((r.copy(), add, pdd))
        assert len(rows) == R
        for i, (rs, a_, p_) in enumerate(rows):
            assert np.all(np.abs(st_h[i, e] - rs) <= 3e-5 + 3e-5 * np.abs(rs)), (e, i)
            assert abs(dd_h[i, e, 0] - a_) <= 2e-3 + 1e-4 * abs(a_) and abs(dd_h[i, e, 1] - p_) <= 5e-4 + 1e-4 * abs(p_), (e, i)
    eng.close()


@pytest.mark.parametrize("n_ctrl,n_save,tail", [(10, 5, 5), (10, 2, 4), (8, 4, 7), (10, 5, 3)])
def test_a_trailing_partial_period_completes_its_saved_rows(n_ctrl, n_save, tail):
    """A run that ends INSIDE a control period (cpmppi_plant_step with n_substeps < period_steps) with dt_save < dt_control: no
    controller call follows the trailing steps, so a row saved on the run's LAST step gets its second derivatives from that very
    launch, under the held control - as the reference's save does (CartPole/__init__.py:403-433).  (Advisor, round 5: the kernel
    deferred the derivatives of a launch's last step to a next call that never comes; 25 steps at 10 / 5 left zeros in the
    recording's final row.)  Cases: the tail ends on a saved step; it ends between two saved steps."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    E, T = 3, 2
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=64, mpc_horizon=10))
    rng = Generator(SFC64(1000 * n_ctrl + 10 * n_save + tail))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-3, 3), rng.uniform(-4, 4), rng.uniform(-0.15, 0.15), rng.uniform(-0.5, 0.5))
                   for _ in range(E)])
    Qs = rng.uniform(-1, 1, (T + 1, E)).astype(f32)
    Lv = rng.uniform(0.25, 0.45, E).astype(f32)
    n_sim = T * n_ctrl + tail
    R = n_sim // n_save + 1
    s = eng.tensor(s0.copy())
    guard = 7.0
    states, dd, Qlog = eng.zeros(R, E, 6) + guard, eng.zeros(R, E, 2) + guard, eng.zeros(T + 1, E)
    states[0] = s
    kw = dict(dt_sim=0.002, period_steps=n_ctrl, L=Lv, states_log=states, dd_log=dd, save_every=n_save, Q_log=Qlog)
    for c in range(T):
        eng.plant_step(s, Qs[c], n_ctrl, period=c, **kw)
    eng.plant_step(s, Qs[T], tail, period=T, **kw)                           # the run's last controller call + the trailing steps
    st_h, dd_h = states.cpu().numpy(), dd.cpu().numpy()
    assert (st_h != guard).all() and (dd_h != guard).all()                   # every row of both logs was written
    for e in range(E):
        r, Q = s0[e].copy(), Qs[0, e]
        add, pdd = O.plant_ode(r, Q, Lv[e])
        rows = [(r.copy(), add, pdd)]
        for gstep in range(1, n_sim + 1):
            r = O.plant_substep(r, add, pdd, 0.002, Lv[e])
            if gstep % n_ctrl == 0:
                Q = Qs[gstep // n_ctrl, e]
            add, pdd = O.plant_ode(r, Q, Lv[e])
            if gstep % n_save == 0:
                rows.append((r.copy(), add, pdd))
        assert len(rows) == R
        for i, (rs, a_, p_) in enumerate(rows):
            assert np.all(np.abs(st_h[i, e] - rs) <= 3e-5 + 3e-5 * np.abs(rs)), (e, i)
            assert abs(dd_h[i, e, 0] - a_) <= 2e-3 + 1e-4 * abs(a_) and abs(dd_h[i, e, 1] - p_) <= 5e-4 + 1e-4 * abs(p_), (e, i)
    assert np.array_equal(Qlog.cpu().numpy(), Qs)
    eng.close()


def _replay(g, tag, i, math_mode, graph=False):
    """One fixture experiment on the device loop with the reference's own perturbations (SFC64 knots from the host)."""
    from cartpolesimulation_amd import schedule as SC
    from cartpolesimulation_amd.configs import legacy_mppi_config
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.harness import BatchedCartPoleExperiment
    key = f"{tag}/{i}"
    cfg = json.loads(g[f"{tag}/config"].item())
    N, H = int(g[f"{tag}/N"]), int(g[f"{tag}/H"])
    batch_all = SC.RandomExperimentSetter(cfg).draw(i + 1, int(g[f"{tag}/cartpole_seed0"]))
    import dataclasses
    b = dataclasses.replace(batch_all, s0=batch_all.s0[i:i + 1], target_position=batch_all.target_position[:, i:i + 1],
                            target_equilibrium=batch_all.target_equilibrium[:, i:i + 1], interpolation_type=batch_all.interpolation_type[i:])
    eng = MPPIEngine(1, legacy_mppi_config(num_rollouts=N, mpc_horizon=H, math_mode=math_mode))
    rng = Generator(SFC64(int(g[f"{tag}/ctrl_seed"]) + i))
    for _ in range(5):
        rng.uniform(-1.0, 1.0)                                         # configure()'s draws (controller_mppi_cartpole.py:355-359)
    stdev = np.float64(g[f"{tag}/stdev"])

    def knots(_c):
        kn = O.sample_knots(rng, N, H, stdev)
        rng.uniform(-1.0, 1.0)                                         # the output-noise draw (:553); its amplitude is 0 in this fixture
        return kn[None]

    # the simulator steps a freshly set controller once on its placeholder state before the experiment (CartPole/__init__.py:759-794)
    un = eng.zeros(1, H)
    eng.step(g[f"{key}/call/s"][0][None], un, float(g[f"{key}/call/tp"][0]), 1.0, knots=knots(0))
    res = BatchedCartPoleExperiment(eng, seed=0).run_schedule(b, knots_fn=knots, u_nom0=un)
    return res, b, eng


@pytest.mark.parametrize("math_mode", ["precise", "fast"])
@pytest.mark.parametrize("i", [0, 1])
def test_reference_experiment_with_moving_target_and_flips_replayed_on_the_device(g, i, math_mode):
    """A WHOLE experiment of the reference's simulator class - legacy MPPI controller in the loop, target position moving along
    its random trace ('previous' / '0-derivative-smooth'), target equilibrium flipping, rows saved every 4 ms (dt_control = 20 ms)
    - replayed by harness.run_schedule: controller steps, plants, schedule look-ups and recording all on the GPU, the host only
    feeds the reference's perturbations.  To 1e-4 while the trajectories have not diverged (first 10 control steps)."""
    res, b, eng = _replay(g, "exp_device", i, math_mode)
    key = f"exp_device/{i}"
    col = lambda n: g[f"{key}/col/{n}"]                                # noqa: E731
    from cartpolesimulation_amd import recording as R
    blk = R.recording_block(res, eng.phys)
    assert blk["states"].shape[0] == len(col("time")) == b.n_sim // b.n_save + 1
    # the schedule columns: exact
    assert np.array_equal(blk["time"], col("time")) and np.array_equal(blk["target_position"][:, 0], col("target_position"))
    assert np.array_equal(blk["target_equilibrium"][:, 0], col("target_equilibrium"))
    flips = np.flatnonzero(np.diff(col("target_equilibrium")) != 0)
    assert len(flips) >= 4
    K = 10                                                             # control steps compared
    r = K * b.n_ctrl // b.n_save + 1
    assert flips[0] < r and np.ptp(col("target_position")[:r]) > 0     # both events fall inside the compared window
    Qc = res["Q"].cpu().numpy()[:, 0]
    np.testing.assert_allclose(Qc[:K + 1], g[f"{key}/call/Q"][1:K + 2], atol=1e-4)
    st = blk["states"][:r, 0]
    for j, n in enumerate(("angle", "angleD", "angle_cos", "angle_sin", "position", "positionD")):
        np.testing.assert_allclose(st[:, j], col(n)[:r], atol=2e-4, rtol=1e-4, err_msg=n)
    np.testing.assert_allclose(blk["Q"][:r, 0], col("Q_calculated")[:r], atol=1e-4)
    np.testing.assert_allclose(blk["Q_ccrc"][:r, 0], col("Q_ccrc")[:r].astype(np.float64), atol=1e-4)
    np.testing.assert_allclose(blk["dd"][:r, 0, 0], col("angleDD")[:r], atol=5e-3, rtol=1e-3)
    np.testing.assert_allclose(blk["dd"][:r, 0, 1], col("positionDD")[:r], atol=2e-3, rtol=1e-3)
    # the whole run stays on the track and its controls in range
    assert np.abs(blk["states"][:, 0, 4]).max() <= 0.198 + 1e-6 and np.abs(Qc).max() <= 1.0
    eng.close()


def test_reference_experiment_that_ends_inside_a_control_period(g):
    """exp_tail on the device loop: 25 simulation steps = two control periods + five trailing steps after the last controller call
    (cpmppi_plant_step with n_substeps < period_steps), three saved rows, no turning points."""
    res, b, eng = _replay(g, "exp_tail", 0, "fast")
    key = "exp_tail/0"
    assert (b.n_sim, b.n_periods, b.n_ctrl) == (25, 2, 10) and res["states"].shape[0] == 3 and res["Q"].shape[0] == 3
    np.testing.assert_allclose(res["Q"].cpu().numpy()[:, 0], g[f"{key}/call/Q"][1:], atol=1e-4)
    st = res["states"].cpu().numpy()[:, 0]
    for j, n in enumerate(("angle", "angleD", "angle_cos", "angle_sin", "position", "positionD")):
        np.testing.assert_allclose(st[:, j], g[f"{key}/col/{n}"], atol=2e-4, rtol=1e-4, err_msg=n)
    np.testing.assert_allclose(res["dd"].cpu().numpy()[:, 0, 0], g[f"{key}/col/angleDD"], atol=5e-3, rtol=1e-3)
    # the five trailing steps did advance the plant beyond the last saved row
    assert not np.array_equal(res["final_state"].cpu().numpy()[0], st[-1])
    assert (b.target_position == 0.0).all()
    eng.close()


@pytest.mark.parametrize("tag,i", [("exp_fine", 0), ("exp_fine", 1), ("exp_coarse", 0)])
def test_reference_experiments_host_paced(g, tag, i):
    """The experiments whose legacy controller multiplies its output by (1 + actuator_noise * uniform) on the host
    (controller_mppi_cartpole.py:553): controller step and plant step on the GPU, that one multiplication in between on the host.
    Target position / equilibrium come from the schedule vectors cpmppi_plant_step refills; dt_save = dt_control / 2 and 2 x."""
    from cartpolesimulation_amd import schedule as SC
    from cartpolesimulation_amd.configs import legacy_mppi_config
    from cartpolesimulation_amd.engine import MPPIEngine
    key = f"{tag}/{i}"
    cfg = json.loads(g[f"{tag}/config"].item())
    N, H, p_Q = int(g[f"{tag}/N"]), int(g[f"{tag}/H"]), float(g[f"{tag}/p_Q"])
    assert p_Q == 0.1
    b = SC.RandomExperimentSetter(cfg).draw(i + 1, int(g[f"{tag}/cartpole_seed0"]))
    eng = MPPIEngine(1, legacy_mppi_config(num_rollouts=N, mpc_horizon=H))
    rng = Generator(SFC64(int(g[f"{tag}/ctrl_seed"]) + i))
    for _ in range(5):
        rng.uniform(-1.0, 1.0)
    stdev = np.float64(g[f"{tag}/stdev"])
    tp_d = eng.tensor(b.target_position[:, i:i + 1].astype(f32))
    te_d = eng.tensor(b.target_equilibrium[:, i:i + 1].astype(f32))
    cur_tp, cur_te = tp_d[0].clone(), te_d[0].clone()
    R = b.n_sim // b.n_save + 1
    s = eng.tensor(b.s0[i:i + 1].copy())
    states, dd, Qlog = eng.zeros(R, 1, 6), eng.zeros(R, 1, 2), eng.zeros(b.n_periods + 1, 1)
    states[0] = s
    un = eng.zeros(1, H)

    def control(state, tp, te):
        Qd, _ = eng.step(state, un, tp, te, knots=O.sample_knots(rng, N, H, stdev)[None])
        Q = f32(Qd.cpu().numpy()[0] * (1 + p_Q * rng.uniform(-1.0, 1.0)))
        return np.clip(Q, f32(-1), f32(1))

    control(g[f"{key}/call/s"][0][None], float(g[f"{key}/call/tp"][0]), 1.0)          # the placeholder call
    kw = dict(dt_sim=b.dt_simulation, period_steps=b.n_ctrl, states_log=states, dd_log=dd, save_every=b.n_save, Q_log=Qlog,
              target_position_table=tp_d, target_equilibrium_table=te_d, sched_stride=b.stride, target_position_out=cur_tp,
              target_equilibrium_out=cur_te)
    K = 12
    for c in range(b.n_periods + 1):
        if c <= K:
            np.testing.assert_allclose(s.cpu().numpy()[0], g[f"{key}/call/s"][c + 1], atol=2e-4, rtol=1e-4)
            assert float(cur_tp[0]) == f32(g[f"{key}/call/tp"][c + 1]) and float(cur_te[0]) == g[f"{key}/call/te"][c + 1]
        Q = control(s, cur_tp, cur_te)
        if c <= K:
            np.testing.assert_allclose(Q, g[f"{key}/call/Q"][c + 1], atol=1e-4)
        eng.plant_step(s, np.array([Q], f32), b.n_ctrl if c < b.n_periods else 0, period=c, **kw)
    col = lambda n: g[f"{key}/col/{n}"]                                # noqa: E731
    r = K * b.n_ctrl // b.n_save
    st = states.cpu().numpy()[:, 0]
    np.testing.assert_allclose(st[:r, 0], col("angle")[:r], atol=2e-4)
    np.testing.assert_allclose(st[:r, 4], col("position")[:r], atol=2e-4)
    np.testing.assert_allclose(dd.cpu().numpy()[:r, 0, 0], col("angleDD")[:r], atol=5e-3, rtol=1e-3)
    eng.close()


def test_reference_experiment_with_a_changing_pole_length_on_the_device_loop(g):
    """exp_varL: the reference's simulator with its pole-length updater switched on ('bounce', a change every 7 simulation steps,
    i.e. INSIDE control periods).  schedule.parameter_table gives the per-step table, harness.run_schedule runs the experiment with
    it - plant, schedule, recording on the GPU; the controller keeps predicting with the default length, as the reference's legacy
    controller does.  To 1e-4 over the first ten control steps; the recording's L column exactly."""
    import dataclasses
    from cartpolesimulation_amd import recording as R
    from cartpolesimulation_amd import schedule as SC
    from cartpolesimulation_amd.configs import legacy_mppi_config
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.harness import ScheduleRun
    tag, key = "exp_varL", "exp_varL/0"
    cfg = json.loads(g[f"{tag}/config"].item())
    cfg["dt"]["saving"] = cfg["dt"]["simulation"]                      # (a pole-length table is per simulation step: stride 1)
    N, H = int(g[f"{tag}/N"]), int(g[f"{tag}/H"])
    b = SC.RandomExperimentSetter(cfg).draw(1, int(g[f"{tag}/cartpole_seed0"]))
    Ltab = SC.parameter_table(json.loads(g[f"{tag}/L_updater"].item()), b.times)
    b = dataclasses.replace(b, L_table=Ltab[:, None].copy())
    eng = MPPIEngine(1, legacy_mppi_config(num_rollouts=N, mpc_horizon=H))
    rng = Generator(SFC64(int(g[f"{tag}/ctrl_seed"])))
    for _ in range(5):
        rng.uniform(-1.0, 1.0)
    stdev = np.float64(g[f"{tag}/stdev"])

    def knots(_c):
        kn = O.sample_knots(rng, N, H, stdev)
        rng.uniform(-1.0, 1.0)
        return kn[None]

    un = eng.zeros(1, H)
    eng.step(g[f"{key}/call/s"][0][None], un, float(g[f"{key}/call/tp"][0]), 1.0, knots=knots(0))
    run = ScheduleRun(eng, b, 0, knots_fn=knots, u_nom0=un)
    # the legacy controller ignores the pole length it is told (its predictor is configured without variable_parameters,
    # controller_mppi_cartpole.py:51-52): the controller's L vector is pinned to the default, the PLANT follows the table
    run.plant["L_out"] = None
    run.cur_L = None
    while run.periods_left:
        run.enqueue_next()
    res = run.finish()
    blk = R.recording_block(res, eng.phys)
    n_save_ref = 2                                                     # the fixture saved every 4 ms = every second simulation step
    st = blk["states"][::n_save_ref, 0]
    col = lambda n: g[f"{key}/col/{n}"]                                # noqa: E731
    assert np.array_equal(blk["L"][::n_save_ref, 0].astype(np.float64), col("L")) and len(np.unique(col("L"))) > 8
    K = 10
    r = K * b.n_ctrl // n_save_ref + 1
    Qc = res["Q"].cpu().numpy()[:, 0]
    np.testing.assert_allclose(Qc[:K + 1], g[f"{key}/call/Q"][1:K + 2], atol=1e-4)
    for j, n in enumerate(("angle", "angleD", "angle_cos", "angle_sin", "position", "positionD")):
        np.testing.assert_allclose(st[:r, j], col(n)[:r], atol=2e-4, rtol=1e-4, err_msg=n)
    np.testing.assert_allclose(blk["dd"][::n_save_ref, 0, 0][:r], col("angleDD")[:r], atol=5e-3, rtol=1e-3)
    # teeth: with the pole length held constant the plant's angular acceleration is off by far more than the tolerance
    b0 = dataclasses.replace(b, L_table=None)
    rng = Generator(SFC64(int(g[f"{tag}/ctrl_seed"])))
    for _ in range(5):
        rng.uniform(-1.0, 1.0)
    un0 = eng.zeros(1, H)
    eng.step(g[f"{key}/call/s"][0][None], un0, float(g[f"{key}/call/tp"][0]), 1.0, knots=knots(0))
    run0 = ScheduleRun(eng, b0, 0, knots_fn=knots, u_nom0=un0)
    while run0.periods_left:
        run0.enqueue_next()
    dd0 = run0.finish()["dd"].cpu().numpy()[::n_save_ref, 0, 0]
    assert np.abs(dd0[:r] - col("angleDD")[:r]).max() > 0.05
    eng.close()


def test_reference_experiment_with_changing_pole_mass_and_a_switching_informer_on_the_device_loop(g):
    """exp_varM: the reference's simulator with BOTH parameter updaters on (pole length every 7, pole MASS every 11 simulation steps)
    and its controller informer in 'switching_regular' mode.  schedule.parameter_table / informer_table give the per-step tables,
    harness.ScheduleRun runs the experiment with them: the plant follows the changing mass (1e-4 over the first ten control steps),
    the vector the next controller call reads its pole length from holds what the reference's controller was TOLD (the true length or
    the initial one), and the recording carries the reference's L / m_pole / *_for_controller columns."""
    from cartpolesimulation_amd import recording as R
    from cartpolesimulation_amd import schedule as SC
    from cartpolesimulation_amd.configs import legacy_mppi_config
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.harness import ScheduleRun
    tag, key = "exp_varM", "exp_varM/0"
    cfg = json.loads(g[f"{tag}/config"].item())
    cfg["dt"]["saving"] = cfg["dt"]["simulation"]                      # (parameter tables are per simulation step: stride 1)
    N, H = int(g[f"{tag}/N"]), int(g[f"{tag}/H"])
    b = SC.RandomExperimentSetter(cfg).draw(1, int(g[f"{tag}/cartpole_seed0"]))
    Ltab = SC.parameter_table(json.loads(g[f"{tag}/L_updater"].item()), b.times)
    mtab = SC.parameter_table(json.loads(g[f"{tag}/m_pole_updater"].item()), b.times)
    told = SC.informer_table(json.loads(g[f"{tag}/informer"].item()), b.times, b.n_ctrl)
    b = dataclasses.replace(b, L_table=Ltab[:, None].copy(), m_pole_table=mtab[:, None].copy(), informed=told)
    eng = MPPIEngine(1, legacy_mppi_config(num_rollouts=N, mpc_horizon=H))
    stdev = np.float64(g[f"{tag}/stdev"])

    def run_with(batch):
        rng = Generator(SFC64(int(g[f"{tag}/ctrl_seed"])))
        for _ in range(5):
            rng.uniform(-1.0, 1.0)

        def knots(_c):
            kn = O.sample_knots(rng, N, H, stdev)
            rng.uniform(-1.0, 1.0)
            return kn[None]

        un = eng.zeros(1, H)
        eng.step(g[f"{key}/call/s"][0][None], un, float(g[f"{key}/call/tp"][0]), 1.0, knots=knots(0))
        run = ScheduleRun(eng, batch, 0, knots_fn=knots, u_nom0=un)
        # the legacy controller ignores the pole length it is told (controller_mppi_cartpole.py:51-52): its L stays the default;
        # the published vector is kept aside and compared with what the reference's calls were handed
        handed = run.cur_L
        run.cur_L = None
        told_L = [float(handed.cpu().numpy()[0])] if handed is not None else []
        while run.periods_left:
            run.enqueue_next()
            if handed is not None:
                told_L.append(float(handed.cpu().numpy()[0]))
        return run.finish(), np.array(told_L, f32)

    res, told_L = run_with(b)
    assert np.array_equal(told_L, g[f"{key}/call/L"][1:].astype(f32)) and len(np.unique(told_L)) > 4
    blk = R.recording_block(res, eng.phys)
    n_save_ref = 2                                                     # the fixture saved every 4 ms = every second simulation step
    col = lambda n: g[f"{key}/col/{n}"]                                # noqa: E731
    assert np.array_equal(blk["L"][::n_save_ref, 0].astype(np.float64), col("L"))
    assert np.array_equal(blk["m_pole"][::n_save_ref, 0].astype(np.float64), col("m_pole")) and len(np.unique(col("m_pole"))) > 5
    assert np.array_equal(np.where(blk["informed"][::n_save_ref, 0], "true", "default"), col("m_pole_for_controller"))
    K = 10
    r = K * b.n_ctrl // n_save_ref + 1
    st = blk["states"][::n_save_ref, 0]
    np.testing.assert_allclose(res["Q"].cpu().numpy()[:K + 1, 0], g[f"{key}/call/Q"][1:K + 2], atol=1e-4)
    for j, n in enumerate(("angle", "angleD", "angle_cos", "angle_sin", "position", "positionD")):
        np.testing.assert_allclose(st[:r, j], col(n)[:r], atol=2e-4, rtol=1e-4, err_msg=n)
    np.testing.assert_allclose(blk["dd"][::n_save_ref, 0, 0][:r], col("angleDD")[:r], atol=5e-3, rtol=1e-3)
    np.testing.assert_allclose(blk["dd"][::n_save_ref, 0, 1][:r], col("positionDD")[:r], atol=2e-3, rtol=1e-3)
    # teeth: with the pole MASS held at the handle's value the plant's angular acceleration is off by more than the tolerance
    res0, _ = run_with(dataclasses.replace(b, m_pole_table=None))
    assert np.abs(res0["dd"].cpu().numpy()[::n_save_ref, 0, 0][:r] - col("angleDD")[:r]).max() > 0.02
    # the recording of this run, native writer == Python writer, with the reference's per-row columns
    import tempfile
    d = tempfile.mkdtemp()
    header = R.create_csv_header(cfg["length_of_experiment"], 0.002, 0.02, 0.002, "mppi-cartpole", "", eng.phys)
    R.write_recordings_native([os.path.join(d, "n.csv")], blk, eng.phys, header)
    R.write_recording(os.path.join(d, "p.csv"), R.typed_columns(blk, 0, eng.phys), header=header)
    assert open(os.path.join(d, "n.csv"), "rb").read() == open(os.path.join(d, "p.csv"), "rb").read()
    body = open(os.path.join(d, "n.csv"), newline="").read().split("\r\n")
    k0 = next(i for i, x in enumerate(body) if x.startswith("time,"))
    ref_rows = g[f"{key}/csv_rows"].item().split("\r\n")
    names = ref_rows[0].split(",")
    for n in ("L", "L_for_controller", "m_pole", "m_pole_for_controller", "time", "target_position", "target_equilibrium"):
        j = names.index(n)
        assert [x.split(",")[j] for x in body[k0 + 1:] if x][::n_save_ref] == [x.split(",")[j] for x in ref_rows[1:]], n
    eng.close()


@pytest.mark.parametrize("i", [0, 1])
def test_reference_experiment_with_control_disturbance_on_the_device_loop(g, i):
    """exp_dist: the reference's simulator with its additive control disturbance on (controlDisturbance 0.3, controlBias 0.05), two
    experiments in a row on one generator.  schedule.apply_parameter_schedule draws both experiments' disturbances, the device loop
    drives the plants with Q_applied = (Q_calculated + disturbance) + bias: states to 1e-4 over the first ten control steps, and
    the recording's Q_calculated / Q_applied / Q_ccrc / u columns."""
    from cartpolesimulation_amd import recording as R
    from cartpolesimulation_amd import schedule as SC
    from cartpolesimulation_amd.configs import legacy_mppi_config
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.harness import BatchedCartPoleExperiment
    tag, key = "exp_dist", f"exp_dist/{i}"
    cfg = json.loads(g[f"{tag}/config"].item())
    d = json.loads(g[f"{tag}/disturbance"].item())
    N, H = int(g[f"{tag}/N"]), int(g[f"{tag}/H"])
    both = SC.RandomExperimentSetter(cfg).draw(2, int(g[f"{tag}/cartpole_seed0"]))
    both = SC.apply_parameter_schedule(both, dict(controlDisturbance=d["controlDisturbance"], controlBias=d["controlBias"], seed=d["seed"]))
    b = dataclasses.replace(both, s0=both.s0[i:i + 1], target_position=both.target_position[:, i:i + 1],
                            target_equilibrium=both.target_equilibrium[:, i:i + 1], interpolation_type=both.interpolation_type[i:],
                            Q_disturbance=both.Q_disturbance[:, i:i + 1].copy())
    eng = MPPIEngine(1, legacy_mppi_config(num_rollouts=N, mpc_horizon=H))
    rng = Generator(SFC64(int(g[f"{tag}/ctrl_seed"]) + i))
    for _ in range(5):
        rng.uniform(-1.0, 1.0)
    stdev = np.float64(g[f"{tag}/stdev"])

    def knots(_c):
        kn = O.sample_knots(rng, N, H, stdev)
        rng.uniform(-1.0, 1.0)
        return kn[None]

    un = eng.zeros(1, H)
    eng.step(g[f"{key}/call/s"][0][None], un, float(g[f"{key}/call/tp"][0]), 1.0, knots=knots(0))
    res = BatchedCartPoleExperiment(eng, seed=0).run_schedule(b, knots_fn=knots, u_nom0=un)
    blk = R.recording_block(res, eng.phys)
    col = lambda n: g[f"{key}/col/{n}"]                                # noqa: E731
    K = 10
    r = K * b.n_ctrl // b.n_save + 1
    np.testing.assert_allclose(res["Q"].cpu().numpy()[:K + 1, 0], g[f"{key}/call/Q"][1:K + 2], atol=1e-4)
    st = blk["states"][:, 0]
    for j, n in enumerate(("angle", "angleD", "angle_cos", "angle_sin", "position", "positionD")):
        np.testing.assert_allclose(st[:r, j], col(n)[:r], atol=2e-4, rtol=1e-4, err_msg=n)
    np.testing.assert_allclose(blk["Q"][:r, 0], col("Q_calculated")[:r], atol=1e-4)
    np.testing.assert_allclose(blk["Q_applied"][:r, 0], col("Q_applied")[:r], atol=1e-4)
    np.testing.assert_allclose(blk["Q_ccrc"][:r, 0], col("Q_ccrc")[:r], atol=1e-4)
    assert np.abs(blk["Q_applied"][:, 0] - blk["Q"][:, 0]).max() > 0.3
    # the disturbance itself is exact: applied - calculated of OUR run is the float32 sum the reference's arithmetic gives for OUR controls
    qc = res["Q"].cpu().numpy()[:, 0]
    assert np.array_equal(blk["Q_applied"][::5, 0], ((qc + b.Q_disturbance[:, 0]).astype(f32) + f32(b.Q_bias)).astype(f32))
    np.testing.assert_allclose(blk["dd"][:r, 0, 0], col("angleDD")[:r], atol=5e-3, rtol=1e-3)
    cols = R.typed_columns(blk, 0, eng.phys)
    assert np.array_equal(f32(cols["u"]), f32(1.77) * blk["Q_applied"][:, 0]) and cols["Q_calculated"] == [float(x) for x in blk["Q"][:, 0]]
    # teeth: without the disturbance the states leave the reference's within the first control steps
    rng = Generator(SFC64(int(g[f"{tag}/ctrl_seed"]) + i))
    for _ in range(5):
        rng.uniform(-1.0, 1.0)
    un0 = eng.zeros(1, H)
    eng.step(g[f"{key}/call/s"][0][None], un0, float(g[f"{key}/call/tp"][0]), 1.0, knots=knots(0))
    res0 = BatchedCartPoleExperiment(eng, seed=0).run_schedule(dataclasses.replace(b, Q_disturbance=None), knots_fn=knots, u_nom0=un0)
    assert np.abs(res0["states"].cpu().numpy()[:r, 0, 1] - col("angleD")[:r]).max() > 0.1
    eng.close()


@pytest.mark.parametrize("E,N,H,cost,dt_save", [(5, 512, 20, "default", 0.004), (3, 256, 15, "quadratic_boundary_grad_minimal", 0.04),
                                                 (1600, 1024, 20, "quadratic_boundary_grad_minimal", 0.02), (4, 512, 20, "quadratic_boundary", 0.01)])
def test_graph_replayed_schedule_equals_the_launched_loop(E, N, H, cost, dt_save):
    """run_schedule captured as a HIP graph of control periods (device step counter: Philox offset = schedule row = recording row)
    and replayed gives the launched loop's recording bit for bit - moving targets, flips, partial last graph included."""
    from cartpolesimulation_amd import schedule as SC
    from cartpolesimulation_amd.configs import MPPIConfig
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.harness import BatchedCartPoleExperiment
    cfg = dict(seed=31, length_of_experiment=0.46, dt=dict(saving=dt_save), keep_target_equilibrium_x_seconds_up=0.1,
               keep_target_equilibrium_x_seconds_down=0.06, turning_points=dict(track_relative_complexity=12),
               random_initial_state=dict(init_limits=dict(angle=[0.0, 20.0], angleD=40.0, position=0.4, positionD=0.2)))
    outs = []
    for graph in (False, True):
        b = SC.RandomExperimentSetter(cfg).draw(E, 77, L=np.linspace(0.3, 0.45, E).astype(f32))
        eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, cost_function_specification=cost))
        res = BatchedCartPoleExperiment(eng, seed=7).run_schedule(b, graph=graph, steps_per_graph=8)
        outs.append({k: res[k].cpu().numpy() for k in ("states", "dd", "Q", "final_state", "u_nom")})
        if E >= 1600:
            assert eng.last_launch()["build_variant"] == 1
        eng.close()
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert b.n_periods == 23 and outs[0]["Q"].shape == (24, E) and np.abs(outs[0]["Q"]).max() > 0.01
    assert np.ptp(b.target_position, axis=0).min() > 0 and (b.target_equilibrium == -1).any()
    # the controller did follow the flips: with target_equilibrium = -1 the plugin costs reward the hanging pole
    assert outs[0]["states"].shape[0] == b.n_sim // b.n_save + 1


@pytest.mark.parametrize("groups", [1, 2])
def test_device_loop_hands_the_previous_control_to_the_costs_that_read_it(groups):
    """The simulator hands every controller call the control applied before it (Q_ccrc / "Q_applied_-1", CartPole/__init__.py:489,
    517-518); quadratic_boundary (and _nonconvex, _grad) charge the change from it (quadratic_boundary.py:83-85).  The device loop:
    cpmppi_plant_step publishes the applied control (Q_applied_out), the next cpmppi_step reads it as previous_input - bit-equal to a
    host-paced loop that passes it by hand, different from a loop that does not."""
    from cartpolesimulation_amd import schedule as SC
    from cartpolesimulation_amd.configs import MPPIConfig
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.harness import BatchedCartPoleExperiment
    from cartpolesimulation_amd.pipeline import EnvGroups, run_schedule_groups
    E, N, H = 4, 512, 20
    cfg = dict(seed=33, length_of_experiment=0.3, keep_target_equilibrium_x_seconds_up=0.1, turning_points=dict(track_relative_complexity=12),
               random_initial_state=dict(init_limits=dict(angle=[0.0, 20.0], angleD=40.0, position=0.4, positionD=0.2)))
    b = SC.RandomExperimentSetter(cfg).draw(E, 78)
    mppi = MPPIConfig(num_rollouts=N, mpc_horizon=H, cost_function_specification="quadratic_boundary")
    eng = MPPIEngine(E, mppi)
    if groups == 1:
        res = BatchedCartPoleExperiment(eng, seed=7).run_schedule(b)
    else:
        eg = EnvGroups(E, mppi, groups)
        res = run_schedule_groups(eg, b, 7)
        torch.cuda.synchronize()
    Q_loop = res["Q"].cpu().numpy()

    def by_hand(with_previous):
        s, u, prev, out = eng.tensor(b.s0).clone(), eng.zeros(E, H), eng.zeros(E), []
        for c in range(b.n_periods + 1):
            row = int(b.rows_at(c * b.n_ctrl))
            tp, te = b.target_position[row].astype(f32), b.target_equilibrium[row].astype(f32)
            Q, _ = eng.step(s, u, tp, te, seed=7, offset=c, **({"previous_input": prev} if with_previous else {}))
            out.append(Q.cpu().numpy().copy())
            prev = Q.clone()
            if c < b.n_periods:
                eng.plant_step(s, Q, b.n_ctrl, dt_sim=b.dt_simulation, period=c)
        return np.stack(out)

    assert np.array_equal(Q_loop, by_hand(True))
    assert np.abs(Q_loop - by_hand(False)).max() > 5e-5                                # (a small term at the shipped weight 1.0: 1.7e-4 here)
    if groups > 1:
        eg.close()
    eng.close()


def test_reference_experiment_with_the_measurement_chain_on_the_device_loop(g):
    """exp_sensor: the reference's simulator with 5 ms latency (2.5 simulation steps), measurement noise, a moving vertical-angle
    offset and a switching informer.  schedule.apply_parameter_schedule tabulates noise draws and offset, cpmppi_plant_step keeps the
    latency ring buffer and ends every period with the measured state (float64 arithmetic on the device): what each controller call
    is handed agrees with what the reference's controller was handed to 2e-6 over the first ten calls (the closed loops drift apart
    at the float32 level), the plant follows to 1e-4, the recording carries the offset columns."""
    from cartpolesimulation_amd import recording as R
    from cartpolesimulation_amd import schedule as SC
    from cartpolesimulation_amd.configs import legacy_mppi_config
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.harness import ScheduleRun
    tag, key = "exp_sensor", "exp_sensor/0"
    cfg = json.loads(g[f"{tag}/config"].item())
    cfg["dt"]["saving"] = cfg["dt"]["simulation"]
    sen, inf = json.loads(g[f"{tag}/sensor"].item()), json.loads(g[f"{tag}/informer"].item())
    N, H = int(g[f"{tag}/N"]), int(g[f"{tag}/H"])
    b = SC.RandomExperimentSetter(cfg).draw(1, int(g[f"{tag}/cartpole_seed0"]))
    prm = dict(latency=sen["latency"], noise=dict(sen["noise"], noise_mode="ON"), vertical_angle_offset=sen["vertical_angle_offset"],
               inform_controller_about_parameters_change=inf, seed=sen["noise"]["seed"])
    b = SC.apply_parameter_schedule(b, prm)
    eng = MPPIEngine(1, legacy_mppi_config(num_rollouts=N, mpc_horizon=H))
    stdev = np.float64(g[f"{tag}/stdev"])

    def run_with(batch):
        rng = Generator(SFC64(int(g[f"{tag}/ctrl_seed"])))
        for _ in range(5):
            rng.uniform(-1.0, 1.0)

        def knots(_c):
            kn = O.sample_knots(rng, N, H, stdev)
            rng.uniform(-1.0, 1.0)
            return kn[None]

        un = eng.zeros(1, H)
        eng.step(g[f"{key}/call/s"][0][None], un, float(g[f"{key}/call/tp"][0]), 1.0, knots=knots(0))
        run = ScheduleRun(eng, batch, 0, knots_fn=knots, u_nom0=un)
        seen = [run.s_ctrl.cpu().numpy()[0].copy()]
        while run.periods_left:
            run.enqueue_next()
            seen.append(run.s_ctrl.cpu().numpy()[0].copy())
        return run.finish(), np.array(seen)

    res, seen = run_with(b)
    K = 10
    want = g[f"{key}/call/s64"][1:]
    assert np.array_equal(seen[0], g[f"{key}/call/s"][1])               # the t = 0 call sees the true state
    assert np.abs(seen[:K + 1] - want[:K + 1]).max() < 2e-6
    blk = R.recording_block(res, eng.phys)
    true_states = blk["states"][::b.n_ctrl, 0]
    assert np.abs(seen[1:K + 1] - true_states[1:K + 1]).max() > 0.03       # ... which is NOT the true state once the chain is on
    n_save_ref = 2
    col = lambda n: g[f"{key}/col/{n}"]                                # noqa: E731
    np.testing.assert_allclose(res["Q"].cpu().numpy()[:K + 1, 0], g[f"{key}/call/Q"][1:K + 2], atol=1e-4)
    r = K * b.n_ctrl // n_save_ref + 1
    st = blk["states"][::n_save_ref, 0]
    for j, n in enumerate(("angle", "angleD", "angle_cos", "angle_sin", "position", "positionD")):
        np.testing.assert_allclose(st[:r, j], col(n)[:r], atol=2e-4, rtol=1e-4, err_msg=n)
    assert np.array_equal(blk["angle_offset"][::n_save_ref, 0, 0], col("vertical_angle_offset"))
    assert np.array_equal(blk["angle_offset"][::n_save_ref, 0, 1], col("vertical_angle_offset_cos"))
    assert np.array_equal(np.where(blk["informed"][::n_save_ref, 0], "true", "default"), col("L_for_controller"))
    # each part of the chain on its own moves what the controller sees (and with it the loop)
    import dataclasses
    for drop in (dict(latency=0.0), dict(measurement_noise=None), dict(angle_offset=None)):
        _, other = run_with(dataclasses.replace(b, **drop))
        assert np.abs(other[1:4] - want[1:4]).max() > 1e-3, drop
    eng.close()


def test_everything_on_launched_graph_and_groups_agree():
    """Every table of the schedule at once - pole length and mass updaters, a switching informer, control disturbance, latency,
    measurement noise, a moving angle offset, a cost that reads the previous control - run launched, as a replayed HIP graph (device
    step counter) and as two env groups enqueued from C: the same recording bit for bit."""
    from cartpolesimulation_amd import schedule as SC
    from cartpolesimulation_amd.configs import MPPIConfig
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.harness import BatchedCartPoleExperiment
    from cartpolesimulation_amd.pipeline import EnvGroups, run_schedule_groups
    E, N, H = 6, 512, 20
    cfg = dict(seed=41, length_of_experiment=0.5, dt=dict(saving=0.004), keep_target_equilibrium_x_seconds_up=0.1,
               turning_points=dict(track_relative_complexity=12),
               random_initial_state=dict(init_limits=dict(angle=[0.0, 20.0], angleD=40.0, position=0.4, positionD=0.2)))
    upd = lambda init, every, inc, lo, hi, mode="bounce": dict(init_value=init, change_every_x_seconds=every, mode=mode,   # noqa: E731
                                                                range_random=[lo, hi], range_clip=[lo, hi], increment=inc, reset_every_x_seconds="inf")
    prm = dict(L=upd(0.395, 0.014, 0.01, 0.3, 0.45), m_pole=upd(0.087, 0.03, 0.004, 0.05, 0.12, "random walk"),
               inform_controller_about_parameters_change=dict(mode="switching_random", change_to_on_after_x_seconds_off=0.08,
                                                              change_to_off_after_x_seconds_on=0.1),
               controlDisturbance=0.2, controlBias=0.01, seed=13, latency=0.0085,
               noise=dict(noise_mode="ON", sigma_angle=0.002, sigma_position=0.0005, sigma_angleD=0.075, sigma_positionD=0.005),
               vertical_angle_offset=upd(1.5, 0.02, 0.005, -0.03, 0.06))
    b = SC.apply_parameter_schedule(SC.RandomExperimentSetter(cfg).draw(E, 91, stride=1), prm, seed=92)
    assert all(x is not None for x in (b.L_table, b.m_pole_table, b.informed, b.Q_disturbance, b.measurement_noise, b.angle_offset))
    assert 0 < b.informed.mean() < 1 and b.latency == 0.0085
    mppi = MPPIConfig(num_rollouts=N, mpc_horizon=H, cost_function_specification="quadratic_boundary")
    outs = []
    for form in ("launched", "graph", "groups"):
        eng = MPPIEngine(E, mppi)
        if form == "groups":
            eg = EnvGroups(E, mppi, 2)
            res = run_schedule_groups(eg, b, 7)
            torch.cuda.synchronize()
        else:
            res = BatchedCartPoleExperiment(eng, seed=7).run_schedule(b, graph=form == "graph", steps_per_graph=6)
        outs.append({k: res[k].cpu().numpy() for k in ("states", "dd", "Q", "final_state", "u_nom")})
        if form == "groups":
            eg.close()
        eng.close()
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), ("graph", k)
        assert np.array_equal(outs[0][k], outs[2][k]), ("groups", k)
    assert np.isfinite(outs[0]["states"]).all() and np.abs(outs[0]["Q"]).max() > 0.05


@pytest.mark.parametrize("latency", [0.0, 0.001, 0.004, 0.0332])
def test_measured_state_follows_the_oracle_chain_at_any_latency(latency):
    """cpmppi_plant_step's measurement chain alone, under given controls: the state handed to the next controller call against the
    oracle's MeasurementChain fed with the DEVICE's own states of every simulation step (so that only the chain is compared) - no
    latency, half a step, exactly two steps, more than a control period (the ring buffer spans periods; before the first step it
    holds zeros with cos = 1), with noise rows, a per-step angle offset and a random informer."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    E, T, n_ctrl, dt = 5, 6, 10, 0.002
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=64, mpc_horizon=10))
    rng = Generator(SFC64(31))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-3, 3), rng.uniform(-6, 6), rng.uniform(-0.15, 0.15), rng.uniform(-0.5, 0.5)) for _ in range(E)])
    Qs = rng.uniform(-1, 1, (T, E)).astype(f32)
    n_sim = T * n_ctrl
    nz = rng.normal(0, 0.05, (T + 1, E, 4)).astype(f32)
    off = np.cumsum(rng.normal(0, 0.02, (n_sim + 1, E)), axis=0)              # float64, moves every step
    told = rng.uniform(size=(n_sim + 1, E)) < 0.5
    tp = np.zeros((n_sim + 1, E), f32)
    s = eng.tensor(s0.copy())
    states = eng.zeros(n_sim + 1, E, 6)
    states[0] = s
    n_back = latency / dt
    hist = np.zeros((int(n_back) + 2, E, 6), f32)
    hist[:, :, 2] = 1.0
    s_meas = eng.zeros(E, 6)
    kw = dict(dt_sim=dt, period_steps=n_ctrl, states_log=states, save_every=1, target_position_table=eng.tensor(tp), s_measured=s_meas,
              latency=latency, state_history=eng.tensor(hist) if n_back > 0 else None, measurement_noise_table=eng.tensor(nz),
              angle_offset_table=torch.as_tensor(off, device=s.device), informed_table=torch.as_tensor(told.astype(np.uint8), device=s.device),
              Q_log=eng.zeros(T + 1, E))
    seen = []
    for c in range(T):
        eng.plant_step(s, Qs[c], n_ctrl, period=c, **kw)
        seen.append(s_meas.cpu().numpy().copy())
    st = states.cpu().numpy()
    for e in range(E):
        chain = S.MeasurementChain(latency, dt)
        for gstep in range(1, n_sim + 1):
            chain.hist.append(st[gstep, e].astype(np.float64))
            if gstep % n_ctrl:
                continue
            k = gstep // n_ctrl
            s1, s2 = chain._past(chain.li), chain._past(chain.li + 1)
            m = s1 + chain.frac * (s2 - s1)
            m[0] = S.wrap_angle_rad(m[0] + float(nz[k, e, 0]))
            m[2], m[3] = np.cos(m[0]), np.sin(m[0])
            m[4] += nz[k, e, 1]; m[1] += nz[k, e, 2]; m[5] += nz[k, e, 3]
            m[0] = S.wrap_angle_rad(m[0] + off[gstep, e])
            if told[gstep, e]:
                m[0] = S.wrap_angle_rad(m[0] - off[gstep, e])
            m[2], m[3] = np.cos(m[0]), np.sin(m[0])
            got = seen[k - 1][e]
            assert np.all(np.abs(got - m) <= 2e-7 + 2e-7 * np.abs(m)), (latency, e, k, got, m)
    if latency > 0.02:                                               # the first call reaches back before the first step: the buffer's zeros
        assert abs(seen[0][0][4]) < 1e-2 + abs(float(nz[1, 0, 1])) and not np.allclose(seen[0][0][4], st[n_ctrl, 0, 4], atol=1e-4)
    eng.close()


@pytest.mark.parametrize("groups", [1, 2])
def test_controller_pole_mass_follows_a_uniform_schedule_with_predictor_ODE(groups):
    """predictor_ODE takes the pole mass from the simulator's 'm_pole' attribute at every call (predictors_customization.py:55-58): with
    a deterministic `m_pole:` updater every experiment has the same schedule, and the device loop sets the handle's mass before each
    controller call - the true mass while the informer says so, the initial one otherwise.  Equal to a loop paced by hand; different
    from a controller that keeps the initial mass; predictor_ODE_v0 ignores the attribute (as the reference's does)."""
    from cartpolesimulation_amd import schedule as SC
    from cartpolesimulation_amd.configs import MPPIConfig
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.harness import BatchedCartPoleExperiment, ScheduleRun
    from cartpolesimulation_amd.pipeline import EnvGroups, run_schedule_groups
    E, N, H = 4, 512, 20
    cfg = dict(seed=35, length_of_experiment=0.3, keep_target_equilibrium_x_seconds_up=0.1, turning_points=dict(track_relative_complexity=12),
               random_initial_state=dict(init_limits=dict(angle=[0.0, 20.0], angleD=40.0, position=0.4, positionD=0.2)))
    prm = dict(m_pole=dict(init_value=0.087, change_every_x_seconds=0.02, mode="increase", range_random=[0.015, 0.3], range_clip=[0.015, 0.3],
                           increment=0.02, reset_every_x_seconds="inf"),
               inform_controller_about_parameters_change=dict(mode="switching_regular", change_to_on_after_x_seconds_off=0.04,
                                                              change_to_off_after_x_seconds_on=0.1))
    b = SC.apply_parameter_schedule(SC.RandomExperimentSetter(cfg).draw(E, 79, stride=1), prm)
    mppi = MPPIConfig(num_rollouts=N, mpc_horizon=H, predictor_type="ODE")
    eng = MPPIEngine(E, mppi)
    run = ScheduleRun(eng, b, 7)
    calls = np.arange(b.n_periods + 1) * b.n_ctrl
    want = np.where(b.informed[calls, 0], b.m_pole_table[calls, 0], np.float32(0.087)).astype(f32)
    assert np.array_equal(run.m_ctrl, want) and len(np.unique(want)) > 4 and (want == np.float32(0.087)).sum() >= 3
    if groups == 1:
        res = BatchedCartPoleExperiment(eng, seed=7).run_schedule(b)
    else:
        eg = EnvGroups(E, mppi, groups)
        res = run_schedule_groups(eg, b, 7)
        torch.cuda.synchronize()
    Q_loop = res["Q"].cpu().numpy()

    def by_hand(masses):
        s, u, out = eng.tensor(b.s0).clone(), eng.zeros(E, H), []
        m_tab = eng.tensor(b.m_pole_table)
        for c in range(b.n_periods + 1):
            row = int(b.rows_at(c * b.n_ctrl))
            eng.set_pole_mass(float(masses[c]))
            Q, _ = eng.step(s, u, b.target_position[row].astype(f32), b.target_equilibrium[row].astype(f32), seed=7, offset=c)
            out.append(Q.cpu().numpy().copy())
            if c < b.n_periods:
                eng.plant_step(s, Q, b.n_ctrl, dt_sim=b.dt_simulation, period=c, period_steps=b.n_ctrl, m_pole_table=m_tab)
        return np.stack(out)

    assert np.array_equal(Q_loop, by_hand(want))
    assert np.abs(Q_loop - by_hand(np.full(len(want), 0.087, f32))).max() > 1e-3
    with pytest.raises(ValueError):
        BatchedCartPoleExperiment(eng, seed=7).run_schedule(b, graph=True)              # a graph replays one mass
    eng.set_pole_mass(0.087)
    v0 = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H))
    assert ScheduleRun(v0, b, 7).m_ctrl is None                                          # predictor_ODE_v0 never reads the attribute
    v0.close()
    if groups > 1:
        eg.close()
    eng.close()


def test_plant_step_and_groups_refuse_what_they_cannot_run():
    """Argument validation of the round's C entry points straight through ctypes: a refused call returns its error code with a text in
    cpmppi_last_error and launches nothing (the state is untouched)."""
    import ctypes as C
    from cartpolesimulation_amd import _lib as L
    from cartpolesimulation_amd.configs import MPPIConfig, build_c_config
    from cartpolesimulation_amd.engine import MPPIEngine
    E = 4
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=64, mpc_horizon=10))
    lib = eng.lib
    s, Q = eng.tensor(np.tile(O.create_cartpole_state(0.1, 0.0, 0.0, 0.0), (E, 1))), eng.zeros(E)
    before = s.clone()
    tab, tab64 = eng.zeros(11, E), torch.zeros(11, E, dtype=torch.float64, device=s.device)
    meas, hist = eng.zeros(E, 6), eng.zeros(3, E, 6)

    def call(**kw):
        a = L.cpmppi_plant_args()
        a.E, a.s, a.Q, a.n_substeps, a.period_steps, a.dt_sim = E, s.data_ptr(), Q.data_ptr(), 10, 10, 0.002
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if torch.is_tensor(v) else v)
        rc = lib.cpmppi_plant_step(eng._h, C.byref(a), None)
        return rc, lib.cpmppi_last_error(eng._h).decode()

    bad = [
        (dict(n_substeps=11), -1, "n_substeps"),
        (dict(dt_sim=0.0), -1, "bad argument"),
        (dict(L_table=tab), -1, "sched_rows"),                                             # a table without its row count
        (dict(L_controller_table=tab, sched_rows=11), -1, "L_controller_table"),           # ... stands in for L_table: give both
        (dict(Q_disturbance_table=tab), -1, "ctrl_rows"),
        (dict(row_envs=2), -1, "row_envs"),
        (dict(s_measured=meas, latency_steps=2, state_history=hist, history_len=3), -1, "history_len"),
        (dict(s_measured=meas, latency_steps=1), -1, "state_history"),
        (dict(s_measured=meas, latency_frac=1.5, state_history=hist, history_len=3), -1, "latency_frac"),
        (dict(s_measured=meas, measurement_noise_table=eng.zeros(3, E, 4)), -1, "ctrl_rows"),
        (dict(s_measured=meas, angle_offset_table=tab64), -1, "sched_rows"),
        (dict(s_measured=meas, angle_offset_table=tab64.data_ptr() + 4, sched_rows=11), -5, "misaligned"),
        (dict(Q_log=tab, ctrl_rows=11, period=11), -1, "period"),
    ]
    for kw, code, text in bad:
        rc, msg = call(**kw)
        assert rc == code and text in msg, (kw.keys(), rc, msg)
    torch.cuda.synchronize()
    assert torch.equal(s, before)                                      # nothing was launched
    assert call()[0] == 0                                              # ... and the plain call still runs
    torch.cuda.synchronize()
    assert not torch.equal(s, before)
    # env groups
    cfg = build_c_config(E, eng.mppi, eng.phys)
    g = C.c_void_p()
    assert lib.cpmppi_groups_create(C.byref(cfg), 0, 0, 0, C.byref(g)) == -1 and b"groups" in lib.cpmppi_groups_last_error(None)
    assert lib.cpmppi_groups_create(C.byref(cfg), 0, 9, 0, C.byref(g)) == 0 and lib.cpmppi_groups_count(g) == E    # (more groups than envs: one env each)
    assert lib.cpmppi_groups_run(g, None, None, 1) == -1 and b"neither" in lib.cpmppi_groups_last_error(g)
    a = L.cpmppi_plant_args()
    a.E, a.s, a.Q, a.n_substeps, a.period_steps, a.dt_sim = E - 1, s.data_ptr(), Q.data_ptr(), 10, 10, 0.002
    assert lib.cpmppi_groups_run(g, None, C.byref(a), 1) == -1 and b"total env count" in lib.cpmppi_groups_last_error(g)
    a.E = E
    counter = torch.zeros(1, dtype=torch.int64, device=s.device)
    a.period_dev = counter.data_ptr()
    assert lib.cpmppi_groups_run(g, None, C.byref(a), 1) == -1 and b"period_dev" in lib.cpmppi_groups_last_error(g)
    assert lib.cpmppi_groups_slice(g, E, None, None) == -1 and lib.cpmppi_groups_handle(g, E) is None
    lib.cpmppi_groups_destroy(g)
    eng.close()


@pytest.mark.parametrize("name", ["rpgd", "cem"])
def test_device_loop_with_another_optimizer_of_the_package(name):
    """The shipped config_controllers.yml names `optimizer: rpgd`: run_schedule(optimizer=...) lets any of the package's optimizers
    (built by controller_mpc for the batch's E envs) compute the controls - one optimizer.step per control period on device tensors,
    fed the schedule's target position / equilibrium like the simulator feeds controller.step - while plant, schedule and recording
    stay the one plant launch.  Equal to a loop paced by hand with an identically seeded optimizer; the plants are steered
    (the targets are followed better than with no control at all)."""
    from cartpolesimulation_amd import schedule as SC
    from cartpolesimulation_amd.controller_mpc import controller_mpc
    from cartpolesimulation_amd.harness import BatchedCartPoleExperiment
    E = 3
    cfg = dict(seed=37, length_of_experiment=0.24, keep_target_equilibrium_x_seconds_up=0.1, turning_points=dict(track_relative_complexity=12),
               random_initial_state=dict(init_limits=dict(angle=[0.0, 10.0], angleD=20.0, position=0.3, positionD=0.1)))
    b = SC.RandomExperimentSetter(cfg).draw(E, 81)
    over = dict(seed=9, mpc_horizon=15, num_rollouts=32) if name == "rpgd" else dict(seed=9, mpc_horizon=15, num_rollouts=128)

    def build():
        c = controller_mpc("CartPole", {}, control_limits=([-1.0], [1.0]), config=over, num_envs=E)
        c.configure(name)
        return c.optimizer

    opt = build()
    assert opt.optimizer_name == name
    res = BatchedCartPoleExperiment(opt.engine, seed=0).run_schedule(b, optimizer=opt)
    Q_loop, states = res["Q"].cpu().numpy(), res["states"].cpu().numpy()
    assert Q_loop.shape == (b.n_periods + 1, E) and np.isfinite(states).all() and np.abs(Q_loop).max() > 0.05
    with pytest.raises(ValueError):
        BatchedCartPoleExperiment(opt.engine, seed=0).run_schedule(b, optimizer=opt, graph=True)
    # by hand, a fresh optimizer with the same seed
    opt2 = build()
    eng = opt2.engine
    s, out = eng.tensor(b.s0).clone(), []
    for c in range(b.n_periods + 1):
        row = int(b.rows_at(c * b.n_ctrl))
        vp = opt2.variable_parameters
        vp.target_position, vp.target_equilibrium = b.target_position[row].astype(f32), b.target_equilibrium[row].astype(f32)
        Q = opt2.step(s, float(b.times[c * b.n_ctrl]), as_tensor=True).reshape(-1).clone()
        out.append(Q.cpu().numpy().copy())
        if c < b.n_periods:
            eng.plant_step(s, Q, b.n_ctrl, dt_sim=b.dt_simulation, period=c)
    assert np.array_equal(Q_loop, np.stack(out))
    assert np.allclose(states[-1], s.cpu().numpy(), atol=0, rtol=0)
    opt.engine.close()
    eng.close()
