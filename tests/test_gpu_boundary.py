"""GPU tests of the drop-in boundary: the reference's controller / optimizer / predictor / cost-function call signatures
 on the HIP path, checked against the oracle.  They read like the reference's own usage:
others/Tests/test_controller_mppi_tf.py:7-48, controller_mppi_cartpole.py:51-52,191, cost_function_gym.py:18-21."""
from types import SimpleNamespace

import numpy as np
import pytest
from numpy.random import SFC64, Generator

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle_np as O
import parity_util as PU  # noqa: E402

f32 = np.float32


class MockSpace:                                    # others/globals_and_utils.py:236-240
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high = np.atleast_1d(low).astype(dtype), np.atleast_1d(high).astype(dtype)
        self.dtype, self.shape = dtype, shape


def test_controller_mpc_reference_usage():
    """The reference's timing script, with assertions: construct, configure("mppi"), step(s0) repeatedly."""
    from cartpolesimulation_amd.controller_mpc import controller_mpc
    from cartpolesimulation_amd.state_utilities import create_cartpole_state, ANGLE_IDX
    ctrl = controller_mpc(environment_name="CartPole",
                          initial_environment_attributes={"target_position": 0.0, "target_equilibrium": 1.0},
                          action_space=MockSpace(-1.0, 1.0, (1,)), observation_space=MockSpace(-np.inf, np.inf, (6,)),
                          config=dict(num_rollouts=512, mpc_horizon=20, seed=3))
    ctrl.configure(optimizer_name="mppi")
    assert ctrl.has_optimizer and ctrl.optimizer.optimizer_name == "mppi"
    assert ctrl.optimizer.num_rollouts == 512 and ctrl.optimizer.mpc_horizon == 20
    s0 = create_cartpole_state({"angle": 0.2, "angleD": 0.0, "position": 0.02, "positionD": 0.0})
    u = ctrl.step(s0)
    assert np.asarray(u).shape == (1,) and -1.0 <= float(u[0]) <= 1.0
    # pole leaning to positive angle: the cart must be pushed to get under it -> consistent sign over repeated solves
    us = [float(ctrl.step(s0, time=0.02 * i, updated_attributes={"target_position": 0.0, "target_equilibrium": 1.0,
                                                                  "L": 0.395, "m_pole": 0.087, "Q_ccrc": 0.0,
                                                                  "Q_applied_-1": 0.0})[0]) for i in range(5)]
    assert all(abs(x) <= 1.0 for x in us) and np.sign(us[-1]) == np.sign(us[0]) != 0
    ctrl.controller_reset()
    assert float(ctrl.optimizer.u_nom.abs().max()) == 0.0
    with pytest.raises(NotImplementedError):
        ctrl.configure(optimizer_name="mppi-var-tf")          # an optimizer section the reference does not ship
    assert s0[ANGLE_IDX] == f32(0.2)


def test_optimizer_mppi_sfc64_matches_oracle_over_steps():
    """Identical noise seeds: three consecutive optimizer steps (shift, clip, cost, correction, update) vs the oracle."""
    from cartpolesimulation_amd.optimizer_mppi import optimizer_mppi
    N, H = 768, 25
    vp = SimpleNamespace(target_position=f32(0.03), target_equilibrium=f32(1.0), L=f32(0.31))
    opt = optimizer_mppi(control_limits=(np.array([-1.0]), np.array([1.0])), seed=5, num_rollouts=N, mpc_horizon=H,
                         noise="sfc64", variable_parameters=vp, cost_function_specification="quadratic_boundary_grad_minimal")
    opt.configure(dt=0.02)
    rng = Generator(SFC64(5))
    cfg = O.MPPIConfig(N=N, H=H)
    u_ref = np.zeros(H, dtype=f32)
    s = O.create_cartpole_state(0.25, -0.5, 0.01, 0.05)
    for it in range(3):
        u = opt.step(s, time=0.02 * it)
        du = O.sample_delta_u(rng, N, H, np.float64(cfg.stdev))
        ref = O.mppi_step(s, u_ref, du, vp.target_position, vp.target_equilibrium, cfg, L=vp.L)
        u_ref = ref["u_new"]
        np.testing.assert_allclose(opt.u_nom.cpu().numpy()[0], u_ref, atol=1e-4)
        np.testing.assert_allclose(u, [ref["Q"]], atol=1e-4)
        s = O.ode_v0_step(s[None], np.array([ref["Q"]], f32), L=vp.L)[0]
    opt.optimizer_reset()
    assert opt.step_counter == 0


def test_optimizer_batched_envs_equal_single_env_runs():
    """E envs in one launch == E separate single-env optimizers (same per-env Philox streams via env_offset)."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    E, N, H = 5, 512, 20
    cfg = MPPIConfig(num_rollouts=N, mpc_horizon=H)
    rng = Generator(SFC64(11))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-1, 1), rng.uniform(-2, 2), rng.uniform(-0.1, 0.1), 0.0)
                   for _ in range(E)])
    tp = rng.uniform(-0.05, 0.05, E).astype(f32)
    Lv = rng.uniform(0.25, 0.45, E).astype(f32)
    big = MPPIEngine(E, cfg)
    un = big.zeros(E, H)
    Q, _ = big.step(s0, un, tp, np.ones(E, f32), L=Lv, seed=77, offset=2, env_offset=0)
    small = MPPIEngine(1, cfg)
    for e in range(E):
        u1 = small.zeros(1, H)
        q1, _ = small.step(s0[e:e + 1], u1, tp[e:e + 1], np.ones(1, f32), L=Lv[e:e + 1], seed=77, offset=2, env_offset=e)
        assert np.array_equal(u1.cpu().numpy()[0], un.cpu().numpy()[e])
        assert float(q1[0]) == float(Q[e])


def test_predictor_seam():
    from cartpolesimulation_amd.predictors import PredictorWrapper, next_state_predictor_ODE_v0, predictor_ODE_v0
    rng = Generator(SFC64(2))
    B, H = 96, 12
    s = np.stack([O.create_cartpole_state(rng.uniform(-3, 3), rng.uniform(-8, 8), rng.uniform(-0.19, 0.19),
                                          rng.uniform(-0.6, 0.6)) for _ in range(B)])
    Q = rng.uniform(-1, 1, (B, H, 1)).astype(f32)
    vp = SimpleNamespace(L=f32(0.27))
    hook = next_state_predictor_ODE_v0(0.02, 10, B, variable_parameters=vp)
    assert abs(hook.t_step - 0.002) < 1e-12
    nxt = hook.step(s, Q[:, 0, :])
    ref = O.ode_v0_step(s, Q[:, 0, 0], L=vp.L)
    ref_b = O.ode_v0_step(s, Q[:, 0, 0], L=vp.L, mode="f64sub")
    # one control step: a tenth of the band (1e-5) around the reference's [float32, float64-substep] interval for every
    # state clear of the edge and of +-pi (flags from the oracle's own two samples), the rest capped (parity_util)
    PU.assert_states(nxt, ref, ref_b, PU.flag_discontinuities(np.stack([s, ref], axis=1)), "one ODE_v0 step", scale=0.1)
    with pytest.raises(AssertionError):
        hook.step(s, Q[:, 0, 0])                               # Q must be 2-D, as the reference asserts (:43-45)
    pw = PredictorWrapper()
    pw.configure(batch_size=B, horizon=H, dt=0.02, predictor_specification="ODE_v0", variable_parameters=vp)
    assert pw.predictor_config["predictor_type"] == "ODE_v0" and pw.predictor_type == "ODE_v0" and pw.horizon == H
    traj = pw.predict(s, Q)
    assert traj.shape == (B, H + 1, 6) and traj.dtype == np.float32 and np.array_equal(traj[:, 0], s)
    ref_traj = O.predict_core(s, Q, L=vp.L)
    ref_traj_b = O.predict_core(s, Q, L=vp.L, mode="f64sub")
    PU.assert_states(traj, ref_traj, ref_traj_b, PU.flag_discontinuities(ref_traj), "predict over 12 steps")
    # one state broadcast over the batch, [H,1] controls for a single rollout, update() is a no-op
    t1 = predictor_ODE_v0(H, 0.02).predict(s[0], Q[0])
    assert t1.shape == (1, H + 1, 6)
    assert pw.update(Q[:, :1], s) is None
    pw.horizon = 6                                             # settable, as the legacy controller does (:473)
    assert pw.predict_core(s, Q[:, :6]).shape == (B, 7, 6)
    with pytest.raises(NotImplementedError):
        pw.configure(batch_size=B, horizon=H, dt=0.02, predictor_specification="GRU-6IN-32H1-32H2-5OUT-0")


def test_cost_function_seam():
    """SURVEY.md Appendix D2 known answers + the plugin interface shapes."""
    from cartpolesimulation_amd.cost_functions import CostFunctionWrapper, quadratic_boundary_grad_minimal, default
    states = np.array([[[-0.320988894, -0.381744802, 0.948923886, -0.315505087, -0.095265023, 0.180272102]],
                       [[2.861763477, -1.590178132, -0.961102605, 0.276191473, 0.190087095, -0.315414101]]], dtype=f32)
    inputs = np.array([[[-0.24]], [[0.8]]], dtype=f32)
    vp = SimpleNamespace(target_position=f32(0.05), target_equilibrium=f32(1.0))
    q = quadratic_boundary_grad_minimal(vp, None)
    stage = q.get_stage_cost(states, inputs, None)
    assert stage.shape == (2, 1)
    np.testing.assert_allclose(stage[:, 0], [1.883728743, 5542.100097656], rtol=1e-5)
    assert q.get_terminal_cost(states[:, 0]).shape == (2, 1) and not q.get_terminal_cost(states[:, 0]).any()
    d = default(vp, None)
    np.testing.assert_allclose(d.get_stage_cost(states, inputs, None)[:, 0], [93.84038544, 6.000019456e+09], rtol=1e-5)
    np.testing.assert_allclose(d.get_terminal_cost(states[:, 0])[:, 0], [10000.0, 10000.0])
    # trajectory cost = sum of stage costs on state_horizon[:, :-1] + terminal on state_horizon[:, -1]
    rng = Generator(SFC64(8))
    traj = O.predict_core(O.create_cartpole_state(0.4, 1.0, 0.05, 0.1), rng.uniform(-1, 1, (64, 15)).astype(f32))
    u = rng.uniform(-1, 1, (64, 15, 1)).astype(f32)
    w = CostFunctionWrapper()
    w.configure(batch_size=64, horizon=15, variable_parameters=vp, environment_name="CartPole",
                cost_function_specification="default")
    total = w.get_trajectory_cost(traj, u, None)
    ref = O.trajectory_cost(O.COST_DEFAULT, traj, u[:, :, 0], vp.target_position, vp.target_equilibrium)
    np.testing.assert_allclose(total, ref, rtol=1e-4)
    np.testing.assert_allclose(w.get_summed_stage_cost(traj[:, :-1], u, None),
                               O.default_stage_cost(traj[:, :-1], u[:, :, 0], vp.target_position, vp.target_equilibrium).sum(1),
                               rtol=1e-4)
    vp.target_position = f32(-0.02)                           # read at call time, like the plugins do
    total2 = w.get_trajectory_cost(traj, u, None)
    assert not np.allclose(total, total2)
    with pytest.raises(ValueError):
        w.configure(cost_function_specification="no_such_plugin")


@pytest.mark.parametrize("case", ["up_centre", "up_edge", "down_edge", "up_no_previous"])
def test_quadratic_boundary_seam_and_fused(golden_dir, case):
    """The in-tree plugin quadratic_boundary (cost_id 4: default.py's kernels with a sub-mode): the cost seam against the
    reference's own outputs (stage under the class's stale `_get_stage_cost` name too, terminal indicator), then the fused
    step against the oracle in both math modes and lane mappings, with and without a previous input; the nonconvex sibling
    (cost_id 5; the reference imports it only with the one missing configuration key supplied: "nc/..." in the fixture) against
    those outputs of the reference's class and against the oracle's restatement; what is not built for them is refused."""
    import os
    from cartpolesimulation_amd import _lib as L
    from cartpolesimulation_amd.cost_functions import quadratic_boundary, quadratic_boundary_nonconvex
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    g = np.load(os.path.join(golden_dir, "qb_costs.npz"))
    vp = SimpleNamespace(target_position=g[f"{case}/target_position"], target_equilibrium=g[f"{case}/target_equilibrium"])
    traj, Qin = g[f"{case}/traj"], g[f"{case}/Q"]
    prev = g[f"{case}/previous_input"]
    prev = None if np.isnan(prev) else f32(prev)
    c = quadratic_boundary(vp, None)
    stage = c._get_stage_cost(traj[:, :-1], Qin[..., None], prev)
    np.testing.assert_allclose(stage, g[f"{case}/stage"], rtol=1e-4, atol=1e-2)      # (costs from 1e2 to 6e11; cos of the stored angle)
    np.testing.assert_array_equal(c.get_terminal_cost(traj[:, -1]), g[f"{case}/terminal"])
    total = c.get_trajectory_cost(traj, Qin[..., None], prev)
    np.testing.assert_allclose(total, g[f"{case}/stage"].astype(np.float64).sum(1) + g[f"{case}/terminal"][:, 0], rtol=1e-4)
    nc = quadratic_boundary_nonconvex(vp, None).get_stage_cost(traj[:, :-1], Qin[..., None], prev)
    np.testing.assert_allclose(nc, O.qb_stage_cost(traj[:, :-1], Qin, prev, vp.target_position, vp.target_equilibrium, nonconvex=True),
                               rtol=1e-4, atol=2e-2)
    np.testing.assert_allclose(nc, g[f"nc/{case}/stage"], rtol=1e-4, atol=2e-2)        # the reference's own class (augmented config)
    np.testing.assert_array_equal(quadratic_boundary_nonconvex(vp, None).get_terminal_cost(traj[:, -1]), g[f"nc/{case}/terminal"])
    # fused step
    N, H = Qin.shape
    for name, cid in (("quadratic_boundary", O.COST_QB), ("quadratic_boundary_nonconvex", O.COST_QB_NONCONVEX)):
        for kw in (dict(rollouts_per_lane=1), dict(rollouts_per_lane=2), dict(math_mode="precise")):
            eng = MPPIEngine(1, MPPIConfig(num_rollouts=N, mpc_horizon=H, cost_function_specification=name, **kw))
            rng = Generator(SFC64(3))
            u0 = (0.2 * rng.standard_normal(H)).astype(f32)
            du = O.sample_delta_u(rng, N, H, np.float64(eng.mppi.sigma))
            un, S = eng.tensor(u0[None].copy()), eng.empty(1, N)
            eng.step(g[f"{case}/s0"][None], un, vp.target_position, vp.target_equilibrium, delta_u=du[None], S_out=S, previous_input=prev)
            cfg = O.MPPIConfig(N=N, H=H, cost_id=cid)
            cfg.cost.qb_previous_input = prev
            ref = O.mppi_step(g[f"{case}/s0"], u0, du, vp.target_position, vp.target_equilibrium, cfg)
            ref_b = O.mppi_step(g[f"{case}/s0"], u0, du, vp.target_position, vp.target_equilibrium, cfg, mode="f64sub")
            Sd = S.cpu().numpy()[0]
            fl = PU.flag_discontinuities(ref["traj"]) | (np.abs(np.abs(ref["traj"][:, :-1, O.POSITION_IDX]) - 0.95 * PU.THL) < 2e-4).any(axis=1)
            if cid == O.COST_QB_NONCONVEX:          # the ripple's slope: 0.15 dd_weight 8 pi / (2 THL) per metre of position error
                fl |= np.abs(ref["S"]) < 1e4
            PU.assert_costs(Sd, ref["S"], ref_b["S"], fl, f"{case} {name} {kw} costs")
            u_shift = np.concatenate([u0[1:], u0[-1:]])
            u_chk = np.clip(u_shift + O.reward_weighted_average(Sd, du), -1, 1)
            np.testing.assert_allclose(un.cpu().numpy()[0], u_chk, atol=2e-5)       # the reduction on the device's own costs
            assert eng.last_launch()["cost_id"] == 1                                # default.py's kernels
            eng.close()
    # not built for these plugins: the adjoint, the GRU predictor, predictor_ODE
    eng = MPPIEngine(1, MPPIConfig(num_rollouts=N, mpc_horizon=H, cost_function_specification="quadratic_boundary"))
    with pytest.raises(L.CpmppiError):
        eng.rollout_cost_grad(g[f"{case}/s0"][None], eng.tensor(Qin[None]), 0.0, 1.0)
    with pytest.raises(L.CpmppiError):
        MPPIEngine(1, MPPIConfig(num_rollouts=N, mpc_horizon=H, cost_function_specification="quadratic_boundary", predictor_type="ODE"))
    eng.close()


@pytest.mark.parametrize("case", ["up_shipped", "down_shipped", "up_all_terms", "down_all_terms"])
def test_quadratic_boundary_grad_seam_and_fused(golden_dir, case):
    """The in-tree plugin quadratic_boundary_grad: cost seam against the reference's own outputs, then the fused step
    (rollout + this cost + update) against the oracle, both weight sets (target_equilibrium = +-1)."""
    import os
    from cartpolesimulation_amd.cost_functions import quadratic_boundary_grad
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    g = np.load(os.path.join(golden_dir, "qbg_costs.npz"))
    w = dict(zip(g[f"{case}/weight_names"], g[f"{case}/weight_values"]))
    w["cos_admissible_angle"] = float(np.cos(f32(w.pop("admissible_angle"))))          # the fixture holds radians
    vp = SimpleNamespace(target_position=g[f"{case}/target_position"], target_equilibrium=g[f"{case}/target_equilibrium"])
    c = quadratic_boundary_grad(vp, None, weights=w)
    traj, Qin, prev = g[f"{case}/traj"], g[f"{case}/Q"], g[f"{case}/previous_input"]
    stage = c.get_stage_cost(traj[:, :-1], Qin[..., None], prev)
    np.testing.assert_allclose(stage, g[f"{case}/stage"], rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(c.get_trajectory_cost(traj, Qin[..., None], prev), g[f"{case}/total"], rtol=1e-4)
    assert not c.get_terminal_cost(traj[:, -1]).any()
    # fused step
    N, H = Qin.shape
    for rpl in (1, 2):
        eng = MPPIEngine(1, MPPIConfig(num_rollouts=N, mpc_horizon=H, cost_function_specification="quadratic_boundary_grad",
                                       cost_weights=w, rollouts_per_lane=rpl))
        rng = Generator(SFC64(3))
        u0 = (0.2 * rng.standard_normal(H)).astype(f32)
        du = O.sample_delta_u(rng, N, H, np.float64(eng.mppi.sigma))
        un = eng.tensor(u0[None].copy())
        S = eng.empty(1, N)
        eng.step(g[f"{case}/s0"][None], un, vp.target_position, vp.target_equilibrium, delta_u=du[None], S_out=S,
                 previous_input=prev)
        cfg = O.MPPIConfig(N=N, H=H, cost_id=O.COST_QBG)
        cfg.cost.qbg_weights = {k: v for k, v in zip(g[f"{case}/weight_names"], g[f"{case}/weight_values"])
                                if k in O.QBG_DEFAULT_WEIGHTS}
        cfg.cost.qbg_previous_input = prev
        ref = O.mppi_step(g[f"{case}/s0"], u0, du, vp.target_position, vp.target_equilibrium, cfg)
        Sd = S.cpu().numpy()[0]
        ref_b = O.mppi_step(g[f"{case}/s0"], u0, du, vp.target_position, vp.target_equilibrium, cfg, mode="f64sub")
        PU.assert_costs(Sd, ref["S"], ref_b["S"], PU.flag_discontinuities(ref["traj"]), f"{case} rpl={rpl} costs")
        # the update: exactness of the reduction on the device's own costs, and end to end at the north_star's 1e-4
        u_shift = np.concatenate([u0[1:], u0[-1:]])
        u_chk = np.clip(u_shift + O.reward_weighted_average(Sd, du), -1, 1)
        np.testing.assert_allclose(un.cpu().numpy()[0], u_chk, atol=2e-5)
        PU.assert_controls(un.cpu().numpy()[0], ref["u_new"], ref_b["u_new"], f"{case} rpl={rpl} u_new",
                           allowance=PU.softmin_allowance(ref["S"], ref_b["S"], du))


def test_previous_input_reaches_the_cost_through_updated_attributes():
    """CartPole.Update_Q hands the control applied last as "Q_applied_-1" / "Q_ccrc" (CartPole/__init__.py:517-518); with
    quadratic_boundary_grad's control-change-rate weight switched on it must change the optimizer's choice."""
    from cartpolesimulation_amd.controller_mpc import controller_mpc
    s = O.create_cartpole_state(0.15, 0.0, 0.0, 0.0)
    J = []
    for prev in (None, 0.9, -0.9):
        ctrl = controller_mpc("CartPole", {"target_position": 0.0, "target_equilibrium": 1.0, "L": 0.395},
                              control_limits=([-1.0], [1.0]),
                              config=dict(seed=5, num_rollouts=1024, mpc_horizon=35, cost_function_specification="quadratic_boundary_grad",
                                          cost_weights=dict(ccrc_weight_up=50.0)))
        ctrl.configure("mppi", controller_logging=True)
        upd = {} if prev is None else {"Q_applied_-1": prev, "Q_ccrc": prev}
        ctrl.step(s, 0.0, upd)
        J.append(np.asarray(ctrl.controller_data_for_csv["J_logged"], dtype=np.float64).reshape(-1))
    # stage 0 adds w (u_0 - prev)^2 instead of w u_0^2: the same rollouts, costs shifted by w (prev^2 - 2 prev u_0)
    d_pos, d_neg = J[1] - J[0], J[2] - J[0]
    np.testing.assert_allclose(0.5 * (d_pos + d_neg), 50.0 * 0.81, rtol=2e-3)          # w prev^2
    u0 = (d_neg - d_pos) / (4.0 * 50.0 * 0.9)                                            # the rollouts' first controls
    assert np.abs(u0).max() <= 1.0 + 1e-3 and u0.std() > 0.05


@pytest.mark.parametrize("mode", ["random_walk", "uniform", "repeated", "iid"])
def test_sampling_types_of_the_legacy_sampler(mode):
    """config_controllers.yml:28 SAMPLING_TYPE: the optimizer seam with each of the sampler's other modes on the SFC64
    stream against the oracle stepping on the oracle's restatement of the same mode (itself pinned to the reference's
    outputs, tests/test_oracle_golden.py)."""
    from types import SimpleNamespace
    from cartpolesimulation_amd.optimizer_mppi import optimizer_mppi
    N, H = 512, 20
    vp = SimpleNamespace(target_position=f32(0.02), target_equilibrium=f32(1.0))
    opt = optimizer_mppi(control_limits=(np.array([-1.0]), np.array([1.0])), seed=9, num_rollouts=N, mpc_horizon=H,
                         noise="sfc64", SAMPLING_TYPE=mode, variable_parameters=vp, optimizer_logging=True)
    opt.configure(dt=0.02, predictor_specification="ODE_v0")
    rng = Generator(SFC64(9))
    cfg = O.MPPIConfig(N=N, H=H)
    u_ref = np.zeros(H, f32)
    s = O.create_cartpole_state(0.2, -0.3, 0.01, 0.05)
    for it in range(2):
        u = opt.step(s)
        du = O.sample_delta_u_mode(rng, N, H, np.float64(cfg.stdev), mode).astype(f32)
        ref, ref_b = PU.oracle_step_both_modes(s, u_ref, du, vp.target_position, vp.target_equilibrium, cfg)
        PU.assert_costs(opt.logging_values["J_logged"][0], ref["S"], ref_b["S"], PU.flag_discontinuities(ref["traj"]), f"{mode} step {it}")
        PU.assert_controls(opt.u_nom.cpu().numpy()[0], ref["u_new"], ref_b["u_new"], f"{mode} step {it} u_nom",
                           allowance=PU.softmin_allowance(ref["S"], ref_b["S"], du))
        np.testing.assert_allclose(u, [ref["Q"]], atol=1e-4)
        u_ref = ref["u_new"]
        s = O.ode_v0_step(s[None], np.array([ref["Q"]], f32))[0]
    with pytest.raises(ValueError):
        optimizer_mppi(num_rollouts=8, mpc_horizon=4, SAMPLING_TYPE="uniform")           # device RNG: interpolated only


def test_profiling_brackets_single_launches_or_groups():
    """cpmppi_set_profiling: 1 = an event pair around every rollout kernel, n > 1 = one pair around every n consecutive
    steps, reported per completed group as bracket / n (an unfinished group is dropped); get_profile resets."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    E, N, H = 4, 512, 20
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H))
    s0 = np.stack([O.create_cartpole_state(0.1 * e, 0.0, 0.0, 0.0) for e in range(E)])
    un = eng.zeros(E, H)
    tp, te = np.zeros(E, f32), np.ones(E, f32)

    def run(k):
        for i in range(k):
            eng.step(s0, un, tp, te, seed=1, offset=i)

    run(3)
    eng.set_profiling(True)
    run(10)
    single, fin = eng.get_profile()
    assert len(single) == 10 and all(0.001 < t < 5.0 for t in single) and all(f == 0.0 for f in fin)   # fused finalize: no third event
    assert len(eng.get_profile()[0]) == 0                     # the recorder was reset
    eng.set_profiling(True, group=4)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(10)                                                   # two complete groups, two steps of a third
    torch.cuda.synchronize()
    wall_ms = 1e3 * (time.perf_counter() - t0)
    grouped, _ = eng.get_profile()
    assert len(grouped) == 2
    # a group's per-step average is a launch CADENCE (the host paces these 15 us kernels): at least about a kernel's duration per
    # step, and the two brackets of four steps each lie inside the wall time of the ten steps.  (Round 6: the upper bound used to
    # be 3 x the slowest singly bracketed kernel - a statement about how evenly the host enqueues, which failed once in some thirty
    # runs of the suite when the box hiccuped.)
    assert all(g > 0.2 * min(single) for g in grouped) and 4.0 * sum(grouped) <= 1.05 * wall_ms, (grouped, single, wall_ms)
    eng.set_profiling(False)
    run(2)
    assert len(eng.get_profile()[0]) == 0
    eng.close()


@pytest.mark.parametrize("math_mode", ["fast", "precise"])
def test_predict_large_launch_equals_small_launches(math_mode):
    """cpmppi_predict stages its stores through LDS above 65536 rollouts (whole 192-byte row segments per store): the
    trajectories are, bit for bit, what the direct-store kernel of smaller launches writes — ragged B (not a multiple of
    the block), H not a multiple of the 8-step staging chunk, per-rollout pole lengths."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    B, H = 65536 + 4391, 13
    eng = MPPIEngine(1, MPPIConfig(num_rollouts=64, mpc_horizon=H, math_mode=math_mode))
    rng = Generator(SFC64(31))
    ang = rng.uniform(-np.pi, np.pi, B)
    s = np.zeros((B, 6), f32)
    s[:, 0], s[:, 1], s[:, 2], s[:, 3] = ang, rng.uniform(-8, 8, B), np.cos(ang), np.sin(ang)
    s[:, 4], s[:, 5] = rng.uniform(-0.19, 0.19, B), rng.uniform(-0.6, 0.6, B)
    Q = rng.uniform(-1, 1, (B, H)).astype(f32)
    Lv = rng.uniform(0.25, 0.45, B).astype(f32)
    big = eng.predict(s, Q, L=Lv).cpu().numpy()
    half = B // 2
    small = np.concatenate([eng.predict(s[:half], Q[:half], L=Lv[:half]).cpu().numpy(),
                            eng.predict(s[half:], Q[half:], L=Lv[half:]).cpu().numpy()])
    assert big.shape == (B, H + 1, 6) and np.array_equal(big, small)
    assert np.array_equal(big[:, 0], s)
    eng.close()


@pytest.mark.parametrize("noise", ["philox", "knots"])
@pytest.mark.parametrize("rpl,small_E,big_E,H", [(2, 200, 1600, 70), (2, 100, 1600, 70), (1, 32, 100, 70), (1, 32, 100, 150)])
def test_builds_of_the_kernel_agree_bit_for_bit(rpl, small_E, big_E, H, noise):
    """The rollout kernel exists in three builds chosen by launch size (latency / mid-size / throughput: different
    scheduling strategies, constants in scalar or vector registers, triples with rollback or not, the nominal sequence
    read from memory per control step or held in lanes and fetched with v_readlane).  An env's result must not depend on
    which build integrated it: the first envs of a large launch (throughput build: > 1.5 M rollouts with two per lane,
    > 65 536 with one) equal, bit for bit, the same envs in a launch small enough for the mid-size (two rollouts per lane:
    200 envs; 100 envs = at most one wave per SIMD, the build with the quiet control step unrolled) or latency (one per
    lane) build.  The nominal sequence is nonzero and the horizon longer than 64 steps (the lanes of
    one register; 150: three register loads), with in-kernel noise and with knots from memory (the latency build holds
    the sequence in lanes only then)."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    N = 1024
    cfg = MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=rpl)
    rng = Generator(SFC64(17))
    ang = rng.uniform(-np.pi, np.pi, big_E)
    s0 = np.zeros((big_E, 6), f32)
    s0[:, 0], s0[:, 1], s0[:, 2], s0[:, 3] = ang, rng.uniform(-6, 6, big_E), np.cos(ang), np.sin(ang)
    s0[:, 4], s0[:, 5] = rng.uniform(-0.18, 0.18, big_E), rng.uniform(-0.5, 0.5, big_E)      # some near the edge: rare events too
    tp = rng.uniform(-0.1, 0.1, big_E).astype(f32)
    Lv = rng.uniform(0.25, 0.45, big_E).astype(f32)
    u0 = rng.uniform(-0.6, 0.6, (big_E, H)).astype(f32)
    outs = []
    for E in (big_E, small_E):
        eng = MPPIEngine(E, cfg)
        un, S = eng.tensor(u0[:E].copy()), eng.empty(E, N)
        if noise == "knots":
            kn, _ = eng.sample(seed=5, offset=3, env_offset=0)
            Q, _ = eng.step(s0[:E], un, tp[:E], np.ones(E, f32), L=Lv[:E], knots=kn, S_out=S)
        else:
            Q, _ = eng.step(s0[:E], un, tp[:E], np.ones(E, f32), L=Lv[:E], seed=5, offset=3, env_offset=0, S_out=S)
        outs.append((Q.cpu().numpy()[:small_E], un.cpu().numpy()[:small_E], S.cpu().numpy()[:small_E]))
        eng.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert np.abs(outs[0][1] - u0[:small_E]).max() > 1e-3                   # the step did update the sequence


def test_host_seam_staging_equals_device_inputs():
    """optimizer_mppi.step with the state and attributes on the HOST (one pinned block, one asynchronous copy, Q back
    through a pinned buffer) gives exactly what the same call with a device-resident state gives, for one env (the
    simulator's call) and for several, over consecutive steps with changing attributes."""
    from cartpolesimulation_amd.controller_mpc import controller_mpc
    for E in (1, 5):
        rng = Generator(SFC64(3 + E))
        ctrls = []
        for _ in range(2):
            c = controller_mpc("CartPole", {"target_position": 0.0, "target_equilibrium": 1.0, "L": 0.395},
                               control_limits=([-1.0], [1.0]), num_envs=E, config=dict(seed=11, num_rollouts=256, mpc_horizon=15))
            c.configure("mppi")
            ctrls.append(c)
        for it in range(4):
            s = np.stack([O.create_cartpole_state(rng.uniform(-1, 1), rng.uniform(-2, 2), rng.uniform(-0.1, 0.1), rng.uniform(-0.2, 0.2))
                          for _ in range(E)])
            attrs = {"target_position": (0.02 * (it + 1) * np.ones(E)).astype(f32), "L": rng.uniform(0.3, 0.45, E).astype(f32)}
            s_in = s[0] if E == 1 else s
            q_host = ctrls[0].step(s_in, 0.02 * it, dict(attrs))
            s_dev = ctrls[1].optimizer.engine.tensor(s_in)
            q_dev = ctrls[1].step(s_dev, 0.02 * it, dict(attrs))
            assert np.array_equal(np.asarray(q_host), np.asarray(q_dev))
            assert q_host.shape == ((1,) if E == 1 else (E, 1))
        assert torch.equal(ctrls[0].optimizer.u_nom, ctrls[1].optimizer.u_nom)


def test_host_seam_reads_attribute_arrays_by_value():
    """The single-env host path keeps its staging arrays between calls and skips the conversion of an attribute that is the
    SAME OBJECT as last time - which is only safe for immutable scalars: an array (or tensor) attribute updated in place
    between two controller steps must reach the kernel with its new value."""
    from types import SimpleNamespace
    from cartpolesimulation_amd.optimizer_mppi import optimizer_mppi
    N, H = 256, 12
    s = O.create_cartpole_state(0.3, -0.5, 0.02, 0.1)
    tp = np.array([0.03], f32)
    vp = SimpleNamespace(target_position=tp, target_equilibrium=f32(1.0), L=np.array(0.3, f32))
    a = optimizer_mppi(seed=8, num_rollouts=N, mpc_horizon=H, variable_parameters=vp)
    a.configure()
    a.step(s)
    tp[0] = -0.06                                   # in place: the same object, a new value
    vp.L[...] = 0.45
    q_a = a.step(s)
    vp2 = SimpleNamespace(target_position=f32(0.03), target_equilibrium=f32(1.0), L=f32(0.3))
    b = optimizer_mppi(seed=8, num_rollouts=N, mpc_horizon=H, variable_parameters=vp2)
    b.configure()
    b.step(s)
    vp2.target_position, vp2.L = f32(-0.06), f32(0.45)
    q_b = b.step(s)
    assert np.array_equal(q_a, q_b) and torch.equal(a.u_nom, b.u_nom)


def test_step_host_equals_step():
    """cpmppi_step_host (host state / attributes in, host Q out, one call) == cpmppi_step on device copies of the same
    inputs with the same Philox (seed, offset, env_offset), with and without per-env pole lengths; bad arrays raise."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    E, N, H = 3, 320, 12
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H))
    rng = Generator(SFC64(23))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-1, 1), rng.uniform(-2, 2), rng.uniform(-0.1, 0.1), 0.1) for _ in range(E)])
    tp, te = rng.uniform(-0.05, 0.05, E).astype(f32), np.ones(E, f32)
    for Lv in (None, rng.uniform(0.3, 0.45, E).astype(f32)):
        ua, ub = eng.zeros(E, H), eng.zeros(E, H)
        for it in range(3):
            q = np.empty(E, f32)
            eng.step_host(s0, ua, tp, te, Lv, 9, it, q, env_offset=4)
            Qd, _ = eng.step(s0, ub, tp, te, L=Lv, seed=9, offset=it, env_offset=4)
            assert np.array_equal(q, Qd.cpu().numpy()) and torch.equal(ua, ub)
    with pytest.raises(ValueError):
        eng.step_host(s0.astype(np.float64), eng.zeros(E, H), tp, te, None, 1, 0, np.empty(E, f32))
    with pytest.raises(ValueError):
        eng.step_host(s0, eng.zeros(E, H), tp[:2], te, None, 1, 0, np.empty(E, f32))
    eng.close()


def test_step_host_zero_copy_and_copy_paths_agree():
    """cpmppi_step_host has two routes: up to 64 envs the kernel reads the pinned block directly and the caller spins on
    a system-scope ticket (no copies, no stream wait); above, one copy each way and a stream wait.  Same controls, bit
    for bit, and the same as the device-pointer entry point; many calls in a row (the ticket is a running counter)."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    N, H = 256, 10
    rng = Generator(SFC64(31))
    for E in (1, 64, 65, 130):
        eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H))
        s0 = np.stack([O.create_cartpole_state(rng.uniform(-1, 1), rng.uniform(-2, 2), rng.uniform(-0.1, 0.1), 0.1) for _ in range(E)])
        tp, te = rng.uniform(-0.05, 0.05, E).astype(f32), np.ones(E, f32)
        Lv = rng.uniform(0.3, 0.45, E).astype(f32)
        ua, ub = eng.zeros(E, H), eng.zeros(E, H)
        for it in range(40):
            q = np.full(E, np.nan, f32)
            eng.step_host(s0, ua, tp, te, Lv, 5, it, q)
            Qd, _ = eng.step(s0, ub, tp, te, L=Lv, seed=5, offset=it)
            assert np.array_equal(q, Qd.cpu().numpy()), (E, it)
            s0[:, 4] += 0.001 * q                                   # the next call sees a different host state
        assert torch.equal(ua, ub)
        eng.close()


def test_u_nom_out_leaves_the_input_untouched():
    """cpmppi_step_args.u_nom_out: the updated sequence goes to the second buffer, u_nom is only read; same values as the
    in-place step, for every noise source and for the GRU path's separate finalize kernel."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    E, N, H = 3, 512, 20
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H))
    rng = Generator(SFC64(41))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-1, 1), rng.uniform(-2, 2), rng.uniform(-0.1, 0.1), 0.1) for _ in range(E)])
    tp, te = rng.uniform(-0.05, 0.05, E).astype(f32), np.ones(E, f32)
    u0 = (0.2 * rng.standard_normal((E, H))).astype(f32)
    kn, du = eng.sample(seed=3, offset=1, delta_u=True)
    for kw in (dict(seed=3, offset=1), dict(knots=kn), dict(delta_u=du), dict(delta_u_tiled=eng.tile_delta_u(du))):
        a, b_in, b_out = eng.tensor(u0.copy()), eng.tensor(u0.copy()), eng.zeros(E, H)
        Qa, _ = eng.step(s0, a, tp, te, **kw)
        Qb, _ = eng.step(s0, b_in, tp, te, u_nom_out=b_out, **kw)
        assert torch.equal(a, b_out) and torch.equal(Qa, Qb) and np.array_equal(b_in.cpu().numpy(), u0), list(kw)
    with pytest.raises(ValueError):
        eng.step(s0, eng.tensor(u0.copy()), tp, te, seed=1, u_nom_out=eng.zeros(E, H + 1))
    eng.close()


def test_native_gather_one_rank():
    """cpmppi_comm_*: the library's own RCCL communicator with world = 1 (what a 1-GPU box can run): unique id,
    communicator, the per-step all-gather from the two alternating u_nom buffers on the side stream — same controls as
    the in-place loop, bit for bit, and every gathered block equals the buffer it was taken from."""
    import ctypes as C
    from cartpolesimulation_amd import _lib as L
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    from cartpolesimulation_amd.shard import NativeGather
    E, N, H = 6, 512, 16
    eng, ref = (MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H)) for _ in range(2))
    uid = C.create_string_buffer(L.COMM_ID_BYTES)
    assert eng.lib.cpmppi_comm_unique_id(uid, None) == 0, eng.lib.cpmppi_last_error(None)
    g = NativeGather(eng, uid.raw, 1, 0)
    assert eng.lib.cpmppi_comm_init(eng._h, uid.raw, 1, 0, None) == -1          # one communicator per handle
    rng = Generator(SFC64(51))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-1, 1), rng.uniform(-2, 2), rng.uniform(-0.1, 0.1), 0.1) for _ in range(E)])
    tp, te = rng.uniform(-0.05, 0.05, E).astype(f32), np.ones(E, f32)
    u_ref = ref.zeros(E, H)
    for i in range(9):
        g.before_step(i)
        eng.step(s0, g.u_in(i), tp, te, seed=2, offset=i, u_nom_out=g.u_out(i))
        g.after_step(i)
        ref.step(s0, u_ref, tp, te, seed=2, offset=i)
        g.sync()
        torch.cuda.synchronize()
        assert torch.equal(g.u_out(i), u_ref), i
        assert torch.equal(g.gathered[(i + 1) & 1].view(E, H), u_ref), i
    # the production form: cpmppi_step_gather (one call; step and gather ordered through device memory), host running
    # far ahead of the device; alternating buffers, then in place
    for i in range(9, 60):
        eng.step(s0, g.u_in(i), tp, te, seed=2, offset=i, u_nom_out=g.u_out(i), gather_into=g.recv(i))
        ref.step(s0, u_ref, tp, te, seed=2, offset=i)
    g.sync()
    torch.cuda.synchronize()
    assert torch.equal(g.u_out(59), u_ref) and torch.equal(g.recv(59).view(E, H), u_ref)
    inplace, rec = g.u_out(59).clone(), torch.zeros(1, E * H, device=u_ref.device)
    for i in range(60, 75):
        eng.step(s0, inplace, tp, te, seed=2, offset=i, gather_into=rec)
        ref.step(s0, u_ref, tp, te, seed=2, offset=i)
    g.sync()
    assert torch.equal(inplace, u_ref) and torch.equal(rec.view(E, H), u_ref)
    assert eng.lib.cpmppi_comm_gather(eng._h, 9, g.u[0].data_ptr(), g.gathered[0].data_ptr(), E * H, None) == -1   # slot out of range
    g.close()
    assert eng.lib.cpmppi_comm_gather(eng._h, 0, g.u[0].data_ptr(), g.gathered[0].data_ptr(), E * H, None) == -1   # no communicator
    eng.close(); ref.close()


def _gather_setup(E=6, N=512, H=16, seed=61):
    import ctypes as C
    from cartpolesimulation_amd import _lib as L
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    from cartpolesimulation_amd.shard import NativeGather
    eng, ref = (MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H)) for _ in range(2))
    uid = C.create_string_buffer(L.COMM_ID_BYTES)
    assert eng.lib.cpmppi_comm_unique_id(uid, None) == 0, eng.lib.cpmppi_last_error(None)
    g = NativeGather(eng, uid.raw, 1, 0)
    rng = Generator(SFC64(seed))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-1, 1), rng.uniform(-2, 2), rng.uniform(-0.1, 0.1), 0.1) for _ in range(E)])
    tp, te = rng.uniform(-0.05, 0.05, E).astype(f32), np.ones(E, f32)
    return eng, ref, g, s0, tp, te


@pytest.mark.parametrize("waiter", ["stream-ops", "kernel"])
def test_step_gather_protocol_with_a_slow_collective(waiter, monkeypatch):
    """The device-side ordering of cpmppi_step_gather against a gather that joins LATE (a spin on the side stream in front
    of every all-gather: a slow peer), the host enqueueing far ahead, with one receive buffer per step so that every gather
    can be checked afterwards against the loop without any collective, bit for bit:
      (a) in place: step i + 1 must not overwrite the sequence gather i still has to read;
      (b) alternating buffers: step i + 2 must not, and gather i never sees a half-written buffer;
    in both forms of the side-stream waiter (hipStreamWaitValue32 / the one-lane kernel of devices without it)."""
    monkeypatch.setenv("CPMPPI_COMM_WAITER", waiter)                 # (the default is the kernel form since the end of round 6)
    eng, ref, g, s0, tp, te = _gather_setup()
    E, H = eng.E, eng.H
    assert eng.lib.cpmppi_debug_comm_mode(eng._h) == (1 if waiter == "stream-ops" else 0)
    K = 12
    eng.lib.cpmppi_debug_comm_delay(eng._h, 400)                     # 400 us per gather; a step of this size takes ~30 us
    u_ref = ref.zeros(E, H)
    want = []
    # (b) alternating buffers
    recv = [torch.zeros(1, E * H, device=u_ref.device) for _ in range(K)]
    for i in range(K):
        eng.step(s0, g.u_in(i), tp, te, seed=4, offset=i, u_nom_out=g.u_out(i), gather_into=recv[i])
        ref.step(s0, u_ref, tp, te, seed=4, offset=i)
        want.append(u_ref.clone())
    g.sync()
    torch.cuda.synchronize()
    for i in range(K):
        assert torch.equal(recv[i].view(E, H), want[i]), f"alternating buffers: gather {i} is not step {i}'s result"
    assert torch.equal(g.u_out(K - 1), u_ref)
    # (a) in place
    inplace = g.u_out(K - 1).clone()
    recv = [torch.zeros(1, E * H, device=u_ref.device) for _ in range(K)]
    want = []
    for i in range(K, 2 * K):
        eng.step(s0, inplace, tp, te, seed=4, offset=i, gather_into=recv[i - K])
        ref.step(s0, u_ref, tp, te, seed=4, offset=i)
        want.append(u_ref.clone())
    g.sync()
    torch.cuda.synchronize()
    for i in range(K):
        assert torch.equal(recv[i].view(E, H), want[i]), f"in place: gather {i} read a sequence step {i + 1} had already overwritten"
    assert torch.equal(inplace, u_ref)
    g.close(); eng.close(); ref.close()


@pytest.mark.parametrize("waiter", ["stream-ops", "kernel"])
def test_step_gather_timeout_drops_the_step_and_reaches_the_host(waiter, monkeypatch):
    """(c) a device-side wait that outlasts cpmppi_comm_set_timeout does NOT proceed: the buffer the late gather still reads
    is left alone (its gather delivers the right sequence), the NEXT cpmppi_step_gather returns CPMPPI_ERR_COMM without a
    cpmppi_comm_sync in between, cpmppi_comm_sync reports it once and clears it, and the handle works again afterwards."""
    from cartpolesimulation_amd import _lib as L
    monkeypatch.setenv("CPMPPI_COMM_WAITER", waiter)
    eng, ref, g, s0, tp, te = _gather_setup(seed=62)
    E, H = eng.E, eng.H
    assert eng.lib.cpmppi_comm_set_timeout(eng._h, 0.002) == 0       # 2 ms
    inplace, u_ref = eng.zeros(E, H), ref.zeros(E, H)
    r0, r1 = (torch.zeros(1, E * H, device=u_ref.device) for _ in range(2))
    eng.lib.cpmppi_debug_comm_delay(eng._h, 60000)                   # the gather joins 60 ms late
    eng.step(s0, inplace, tp, te, seed=5, offset=0, gather_into=r0)  # step 0: fine; its gather is the late one
    ref.step(s0, u_ref, tp, te, seed=5, offset=0)
    eng.lib.cpmppi_debug_comm_delay(eng._h, 0)
    eng.step(s0, inplace, tp, te, seed=5, offset=1, gather_into=r1)  # step 1 (in place): waits for gather 0, gives up after 2 ms
    torch.cuda.synchronize()                                         # (the launch stream: the rollout kernels are done)
    with pytest.raises(L.CpmppiError) as ei:                         # no sync in between: the next call already says so
        eng.step(s0, inplace, tp, te, seed=5, offset=2, gather_into=r1)
    assert ei.value.code == -6
    with pytest.raises(L.CpmppiError) as ei:
        g.sync()
    assert ei.value.code == -6
    assert torch.equal(r0.view(E, H), u_ref), "the late gather did not deliver step 0's sequence"
    assert torch.equal(inplace, u_ref), "the step whose wait timed out overwrote the buffer all the same"
    g.sync()                                                         # reported once, cleared
    eng.lib.cpmppi_comm_set_timeout(eng._h, 10.0)
    for i in (1, 2, 3):                                              # the dropped step 1 is simply taken again
        eng.step(s0, inplace, tp, te, seed=5, offset=i, gather_into=r1)
        ref.step(s0, u_ref, tp, te, seed=5, offset=i)
    g.sync()
    assert torch.equal(inplace, u_ref) and torch.equal(r1.view(E, H), u_ref)
    assert eng.lib.cpmppi_comm_set_timeout(ref._h, 1.0) == -1        # no communicator on that handle
    g.close(); eng.close(); ref.close()


def test_the_default_side_stream_waiter_times_out_by_itself():
    """The default form of the side stream's ordering (one folded kernel per step, since the end of round 6) has a timeout OF ITS OWN
    (verdict r5, weak #4: hipStreamWaitValue32 has none): a step that never publishes - an orphan wait enqueued by a test hook - makes
    the waiter give up after the handle's timeout and raise the error by itself; cpmppi_comm_sync returns CPMPPI_ERR_COMM once, clears,
    and the handle and its communicator work again - and the published / completed bookkeeping is still consistent."""
    import time
    from cartpolesimulation_amd import _lib as L
    eng, ref, g, s0, tp, te = _gather_setup(seed=64)
    E, H = eng.E, eng.H
    assert eng.lib.cpmppi_debug_comm_mode(eng._h) == 0 and g.info()["stream_memory_ops"] == 0        # the kernel form is the default
    u, u_ref = eng.zeros(E, H), ref.zeros(E, H)
    recv = torch.zeros(1, E * H, device=u.device)
    eng.step(s0, u, tp, te, seed=5, offset=0, gather_into=recv)
    ref.step(s0, u_ref, tp, te, seed=5, offset=0)
    g.sync()
    assert torch.equal(recv.view(E, H), u_ref)
    assert eng.lib.cpmppi_comm_set_timeout(eng._h, 0.05) == 0
    assert eng.lib.cpmppi_debug_comm_orphan_wait(eng._h) == 0          # the side stream now waits for a step nobody launches
    t0 = time.perf_counter()
    with pytest.raises(L.CpmppiError) as ei:
        g.sync()
    dt = time.perf_counter() - t0
    assert ei.value.code == -6 and 0.04 < dt < 5.0, dt                 # the waiter's own timeout
    g.sync()                                                           # reported once, cleared
    eng.lib.cpmppi_comm_set_timeout(eng._h, 10.0)
    for i in (1, 2, 3):                                                # the handle and its communicator work again (the orphan took a step number)
        eng.step(s0, u, tp, te, seed=5, offset=i, gather_into=recv)
        ref.step(s0, u_ref, tp, te, seed=5, offset=i)
    g.sync()
    assert torch.equal(u, u_ref) and torch.equal(recv.view(E, H), u_ref)
    g.close(); eng.close(); ref.close()


def test_comm_sync_and_destroy_escape_a_wait_that_nothing_will_satisfy(monkeypatch):
    """advisor r4: in stream-memory-operation mode (CPMPPI_COMM_WAITER=stream-ops; the default until round 6, where the folded waiter
    kernel with its own timeout took over) the side stream's wait for a published step has no timeout of its own.  A step
    that never publishes (here: an orphan wait enqueued by a test hook) must not wedge cpmppi_comm_sync / cpmppi_comm_destroy:
    they poll for the handle's timeout, release the wait from the host, and report CPMPPI_ERR_COMM; the handle works afterwards.
    And cpmppi_step_gather refuses a stream that is being captured (a graph would re-publish a baked step number)."""
    import time
    from cartpolesimulation_amd import _lib as L
    monkeypatch.setenv("CPMPPI_COMM_WAITER", "stream-ops")
    eng, ref, g, s0, tp, te = _gather_setup(seed=63)
    E, H = eng.E, eng.H
    if eng.lib.cpmppi_debug_comm_mode(eng._h) != 1:
        pytest.skip("no stream memory operations on this device")
    u, u_ref = eng.zeros(E, H), ref.zeros(E, H)
    recv = torch.zeros(1, E * H, device=u.device)
    eng.step(s0, u, tp, te, seed=5, offset=0, gather_into=recv)
    ref.step(s0, u_ref, tp, te, seed=5, offset=0)
    g.sync()
    assert torch.equal(recv.view(E, H), u_ref)
    assert eng.lib.cpmppi_comm_set_timeout(eng._h, 0.05) == 0
    assert eng.lib.cpmppi_debug_comm_orphan_wait(eng._h) == 0          # the side stream now waits for a step nobody launches
    t0 = time.perf_counter()
    with pytest.raises(L.CpmppiError) as ei:
        g.sync()
    dt = time.perf_counter() - t0
    assert ei.value.code == -6 and 0.04 < dt < 5.0, dt                 # bounded by the timeout, not for ever
    g.sync()                                                           # reported once, cleared
    eng.lib.cpmppi_comm_set_timeout(eng._h, 10.0)
    for i in (1, 2):                                                   # the handle and its communicator work again
        eng.step(s0, u, tp, te, seed=5, offset=i, gather_into=recv)
        ref.step(s0, u_ref, tp, te, seed=5, offset=i)
    g.sync()
    assert torch.equal(u, u_ref) and torch.equal(recv.view(E, H), u_ref)
    info = g.info()
    assert info["world"] == 1 and info["rccl_ranks"] == 1 and info["rccl_rank"] == 0 and info["stream_memory_ops"] == 1
    assert info["gathers_enqueued"] == 4 and info["rccl_version"] > 20000
    # a captured launch stream is refused
    s0, tp, te = eng.tensor(s0), eng.tensor(tp), eng.tensor(te)     # (no upload inside the capture)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with pytest.raises(L.CpmppiError) as ei:
            with torch.cuda.graph(graph, stream=side):
                eng.step(s0, u, tp, te, seed=5, offset=3, gather_into=recv)
    assert ei.value.code == -1 and "captured" in str(ei.value)
    torch.cuda.current_stream().wait_stream(side)
    # destroy with an orphan wait pending: returns (bounded), does not hang
    assert eng.lib.cpmppi_comm_set_timeout(eng._h, 0.05) == 0
    assert eng.lib.cpmppi_debug_comm_orphan_wait(eng._h) == 0
    t0 = time.perf_counter()
    g.close()
    assert time.perf_counter() - t0 < 5.0
    eng.close(); ref.close()


def test_failed_step_leaves_the_event_recorder_intact():
    """A step that fails validation while profiling is on (GRU requested without a model) must not leave a half-recorded
    bracket: the steps before and after it are still reported."""
    from cartpolesimulation_amd import _lib as L
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    E, N, H = 2, 256, 10
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H))
    s0 = np.stack([O.create_cartpole_state(0.1, 0.0, 0.0, 0.0)] * E)
    u = eng.zeros(E, H)
    for group in (1, 2):
        eng.set_profiling(True, group=group)
        eng.step(s0, u, 0.0, 1.0, seed=1, offset=0)
        eng.step(s0, u, 0.0, 1.0, seed=1, offset=1)
        with pytest.raises(L.CpmppiError):
            eng.step(s0, u, 0.0, 1.0, seed=1, offset=2, predictor="GRU")
        eng.step(s0, u, 0.0, 1.0, seed=1, offset=3)
        eng.step(s0, u, 0.0, 1.0, seed=1, offset=4)
        r, _ = eng.get_profile()
        assert len(r) == 4 // group and (r > 0).all(), (group, r)
    eng.close()


def test_a_late_gather_does_not_starve_a_launch_of_many_envs():
    """Round 6: every env's finalizing block waits INSIDE the rollout kernel for the all-gather that still reads the buffer the step
    overwrites.  With more envs than the device has workgroup slots, a gather that is late by more than a step finds every slot held
    by a spinning block.  Two ranks sharing ONE device deadlocked on that until the timeout (the other rank's rollout blocks could
    not be dispatched: bench.py --gpus 2 on one device, first run); launches of many envs therefore wait in front of the kernel, with
    one lane (gather_guard_kernel).  Here, one process: 4096 envs of one block each, gather 0 joins 30 ms late, 1 s timeout.  WITH
    the guard (the default) nothing is dropped, the run takes the 30 ms, every gathered block equals the loop without a collective
    and carries its stamp.  WITHOUT it the outcome is printed, not asserted beyond "either it gets through or the timeout is
    reported": measured on MI355X it gets through too (the side stream's high-priority dispatches pass the spinning blocks)."""
    import ctypes as C
    import time
    from cartpolesimulation_amd import _lib as L
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    from cartpolesimulation_amd.shard import NativeGather
    E, N, H, K = 4096, 256, 10, 6
    rng = Generator(SFC64(71))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-1, 1), rng.uniform(-2, 2), rng.uniform(-0.1, 0.1), 0.1) for _ in range(E)])
    tp, te = rng.uniform(-0.05, 0.05, E).astype(f32), np.ones(E, f32)
    cfg = MPPIConfig(num_rollouts=N, mpc_horizon=H)
    ref = MPPIEngine(E, cfg)
    u_ref, want = ref.zeros(E, H), []
    for i in range(K):
        ref.step(s0, u_ref, tp, te, seed=6, offset=i)
        want.append(u_ref.clone())
    for guard in (False, True):
        eng = MPPIEngine(E, cfg)
        uid = C.create_string_buffer(L.COMM_ID_BYTES)
        assert eng.lib.cpmppi_comm_unique_id(uid, None) == 0
        g = NativeGather(eng, uid.raw, 1, 0, stamped=True)
        assert eng.lib.cpmppi_comm_set_timeout(eng._h, 1.0) == 0
        if not guard:
            assert eng.lib.cpmppi_debug_comm_guard_min_envs(eng._h, 0xFFFFFFFF) == 0       # the round-5 behaviour
        recv = [torch.zeros(1, E * H + L.GATHER_STAMP_FLOATS, device=u_ref.device) for _ in range(K)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        err = None
        try:
            for i in range(K):
                eng.lib.cpmppi_debug_comm_delay(eng._h, 30000 if i == 0 else 0)          # gather 0 joins 30 ms late
                eng.step(s0, g.u_in(i), tp, te, seed=6, offset=i, u_nom_out=g.u_out(i), gather_into=recv[i])
            torch.cuda.synchronize()
            g.sync()
        except L.CpmppiError as e:
            err = e
            try:
                g.sync()
            except L.CpmppiError:
                pass
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if guard:
            assert err is None and dt < 5.0, (err, dt)       # (nothing dropped; the 30 ms the gather was late, not the timeout)
            for i in range(K):
                assert torch.equal(recv[i][0, :E * H].view(E, H), want[i]), f"gather {i}"
                assert int(recv[i][0, E * H:E * H + 1].view(torch.int32)) == i + 1
        else:
            print(f"[without the guard] {dt:.2f} s, error: {err}")
            assert err is None or (err.code == -6 and dt >= 0.9)
        g.close(); eng.close()
    ref.close()
