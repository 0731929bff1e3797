"""GPU: the reference's IN-TREE MPPI controller (Control_Toolkit_ASF/Controllers/controller_mppi_cartpole.py:337-580) as a
class over the HIP engine — driven exactly as the simulator drives it and compared with traces of the reference's own class
(tests/golden/legacy_step_*.npz: three consecutive `step` calls at the C1 and C2 sizes; closed_loop_c1.npz: BASELINE config
C1, 50 control steps in closed loop with the plant)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle_np as O  # noqa: E402
import parity_util as PU  # noqa: E402

f32 = np.float32


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def make(N, H, seed, p_Q, math_mode, **cfg):
    from cartpolesimulation_amd.controller_mppi_cartpole import controller_mppi_cartpole
    ctrl = controller_mppi_cartpole("CartPole", {"target_position": f32(0.0), "target_equilibrium": f32(1.0)},
                                    (np.array([-1.0], f32), np.array([1.0], f32)),
                                    config=dict(seed=int(seed), num_rollouts=N, mpc_horizon=H, **cfg),
                                    actuator_noise=p_Q, math_mode=math_mode)
    ctrl.configure()
    return ctrl


@pytest.mark.parametrize("math_mode", ["precise", "fast"])
@pytest.mark.parametrize("shape", ["256x20", "1024x50"])
def test_step_traces_of_the_reference_class(golden_dir, shape, math_mode):
    g = load(golden_dir, f"legacy_step_{shape}.npz")
    N, H = int(g["N"]), int(g["H"])
    ctrl = make(N, H, g["seed"], float(g["p_Q"]), math_mode)
    assert np.isclose(ctrl.SQRTRHODTINV, g["stdev"], rtol=0, atol=0)
    cfg = O.MPPIConfig(N=N, H=H, SQRTRHOINV=0.02, cost_id=O.COST_LEGACY, control_mode="penalise", shift_mode="none",
                       correction_u="u_nom")
    # the knots the controller will draw, regenerated on an identical stream (5 configure() draws, then per step the
    # knots and ONE uniform for the output noise)
    rng = np.random.Generator(np.random.SFC64(int(g["seed"])))
    for _ in range(5):
        rng.uniform(-1.0, 1.0)
    kn_all = []
    for it in range(g["s_seq"].shape[0]):
        kn_all.append(O.sample_knots(rng, N, H, np.float64(g["stdev"])))
        rng.uniform(-1.0, 1.0)
    for it in range(g["s_seq"].shape[0]):
        u_before, u_prev_before = ctrl.u.copy(), ctrl.u_prev.copy()
        Q = ctrl.step(g["s_seq"][it], 0.02 * it, {"target_position": g["target"]})
        assert Q.dtype == np.float32 and ctrl.iteration == it
        np.testing.assert_allclose(Q, g["Q"][it], atol=1e-4)
        np.testing.assert_allclose(ctrl.u_prev, g["u_updated"][it], atol=1e-4)          # u after the update, before the shift
        np.testing.assert_array_equal(ctrl.u[:-1], ctrl.u_prev[1:])                     # :561-562
        assert ctrl.u[-1] == 0.0
        # per-rollout costs against the reference's own: every rollout the oracle does not flag within 1e-4 + its A/B gap
        du = O.interpolate_knots(kn_all[it], H)
        S_b, _ = O.legacy_rollout_costs(g["s_seq"][it], u_before, du, u_prev_before, f32(g["target"]), cfg, mode="f64sub")
        _, traj = O.legacy_rollout_costs(g["s_seq"][it], u_before, du, u_prev_before, f32(g["target"]), cfg)
        fl = PU.flag_discontinuities(traj) | PU.flag_indicators(traj, "legacy", float(g["target"]))
        PU.assert_costs(ctrl.S_tilde_k, g["S"][it], S_b, fl, f"{shape} step {it} S_tilde_k", strict=True)


@pytest.mark.parametrize("math_mode", ["precise", "fast"])
def test_closed_loop_c1_through_the_controller_class(golden_dir, math_mode):
    """BASELINE config C1: the controller class in the loop with the device plant, on the reference's noise seed."""
    g = load(golden_dir, "closed_loop_c1.npz")
    N, H = int(g["N"]), int(g["H"])
    ctrl = make(N, H, g["seed"], float(g["p_Q"]), math_mode)
    eng = ctrl.engine
    s = eng.tensor(g["s"][0][None].copy())
    for c in range(g["s"].shape[0]):
        s_host = s.cpu().numpy()[0]
        if c < 10:
            np.testing.assert_allclose(s_host, g["s"][c], atol=2e-4, rtol=1e-4)
        Q = ctrl.step(s_host, 0.02 * c, {"target_position": g["target"]})
        if c < 10:
            np.testing.assert_allclose(Q, g["Q"][c], atol=1e-4)
            np.testing.assert_allclose(ctrl.u_prev, g["u_updated"][c], atol=1e-4)
        eng.plant_advance(s, np.array([Q], dtype=f32), n_substeps=10, dt_sim=0.002)
    s_host = s.cpu().numpy()[0]
    assert abs(s_host[O.ANGLE_IDX]) < 0.2 and abs(s_host[O.POSITION_IDX]) < 0.198


def test_update_every_sampling_types_and_horizon_change():
    N, H = 256, 20
    # update_every = 2: the optimisation (and its RNG draws) happens on even iterations only; the shift happens every step
    ctrl = make(N, H, 5, 0.0, "fast", update_every=2)
    s = O.create_cartpole_state(0.1, 0.0, 0.0, 0.0)
    q0 = ctrl.step(s, 0.0)
    u_after_first = ctrl.u.copy()
    q1 = ctrl.step(s, 0.02)                                   # no optimisation: Q is the shifted sequence's head
    np.testing.assert_array_equal(q1, np.clip(np.float32(u_after_first[0]), -1, 1))
    np.testing.assert_array_equal(ctrl.u[:-1], u_after_first[1:])
    # every SAMPLING_TYPE runs through the class and matches the oracle controller stepping on the same stream
    for mode in ("random_walk", "uniform", "repeated", "iid"):
        c2 = make(N, H, 7, 0.1, "fast", SAMPLING_TYPE=mode)
        rng = np.random.Generator(np.random.SFC64(7))
        for _ in range(5):
            rng.uniform(-1.0, 1.0)
        cfg = O.MPPIConfig(N=N, H=H, SQRTRHOINV=0.02, cost_id=O.COST_LEGACY, control_mode="penalise", shift_mode="none")
        du = O.sample_delta_u_mode(rng, N, H, c2.SQRTRHODTINV, mode).astype(f32)
        S_ref, u_ref, _ = O.legacy_mppi_update(s, np.zeros(H, f32), du, np.zeros(H, f32), f32(0.0), cfg)
        Q = c2.step(s, 0.0)
        np.testing.assert_allclose(c2.u_prev, u_ref, atol=1e-4)
        np.testing.assert_allclose(Q, np.clip(f32(u_ref[0] * (1 + 0.1 * rng.uniform(-1.0, 1.0))), -1, 1), atol=1e-4)
    # the GUI changes the horizon on a live controller (:472-475): the leading part of u is kept
    c3 = make(N, H, 9, 0.0, "fast")
    c3.step(s, 0.0)
    head = c3.u[:10].copy()
    c3.mpc_horizon = 30
    c3.step(s, 0.02)
    assert c3.u.shape == (30,) and c3.engine.H == 30
    c4 = make(N, H, 9, 0.0, "fast")
    c4.step(s, 0.0)
    c4.mpc_horizon = 10
    c4.update_control_vector()
    np.testing.assert_array_equal(c4.u, head)
    with pytest.raises(NotImplementedError):
        make(N, H, 1, 0.0, "fast", predictor_specification="SGP_10")
    # the shipped YAML's "ODE" (config_controllers.yml:14) = next_state_predictor_ODE: its own kernels, other controls
    c5, c6 = make(N, H, 9, 0.0, "fast", predictor_specification="ODE"), make(N, H, 9, 0.0, "fast")
    q5, q6 = c5.step(s, 0.0), c6.step(s, 0.0)
    assert c5.engine.mppi.predictor_type == "ODE" and np.isfinite(q5) and q5 != q6
