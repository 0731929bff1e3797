"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors and the numpy oracle.

Tolerances (BASELINE.json north_star "within 1e-4"; the rules live in tests/parity_util.py):
  * states            |d| <= 1e-4 + 1e-4*|x| + |A - B| (the reference's own float32-vs-float64-substep ambiguity, from the
                      oracle) for EVERY rollout clear of the discontinuities (edge bounce, +-pi wrap); rollouts the oracle
                      flags near one are counted and capped (<= 2 % of them, <= 0.5 % of all)
  * controls (u_new, Q, soft-min weights)   1e-4 absolute, fixed
  * costs             1e-4 relative + the same oracle A/B gap, every unflagged rollout
"""
import os

import numpy as np
import pytest
from numpy.random import SFC64, Generator

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import oracle_np as O  # noqa: E402  (the checker)
import parity_util as PU  # noqa: E402

f32 = np.float32
MATH_MODES = ["precise", "fast"]


def engine(E, N, H, **kw):
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    return MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, **kw))


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def state_ok(out, ref, scale=1.0):
    return np.abs(out - ref) <= scale * (1e-4 + 1e-4 * np.abs(ref))


def regen_delta_u(seed, N, H, stdev):
    rng = Generator(SFC64(int(seed)))
    for _ in range(5):
        rng.uniform(-1.0, 1.0)
    kn = O.sample_knots(rng, N, H, np.float64(stdev))
    return kn, O.interpolate_knots(kn, H)


@pytest.mark.parametrize("math_mode", MATH_MODES)
def test_predict_single_step_kats(golden_dir, math_mode):
    g = load(golden_dir, "kat_step.npz")
    eng = engine(1, 256, 1, math_mode=math_mode)
    s, Q, L = g["s_in"], g["Q_in"], g["L_in"]
    out1 = eng.predict(s, Q[:, None], L=L)[:, 1].cpu().numpy()
    # rows the reference itself takes through a bounce or to the +-pi seam within these two control steps are flagged
    both = np.stack([s, g["step1_A"], g["step2_A"]], axis=1)
    flagged = PU.flag_discontinuities(both)
    PU.assert_states(out1, g["step1_A"], g["step1_B"], flagged, "one control step", scale=0.1, strict=True)     # a tenth of the band
    eng2 = engine(1, 256, 2, math_mode=math_mode)
    out2 = eng2.predict(s, np.stack([Q, Q], 1), L=L)[:, 2].cpu().numpy()
    PU.assert_states(out2, g["step2_A"], g["step2_B"], flagged, "two control steps", scale=0.2, strict=True)
    # one substep: run with dt = 0.002, S = 1 through a dedicated engine
    eng3 = engine(1, 256, 1, math_mode=math_mode, mpc_timestep=0.002, intermediate_steps=1)
    sub = eng3.predict(s, Q[:, None], L=L)[:, 1].cpu().numpy()
    assert np.abs(sub - g["sub1_A"]).max() < 2e-5          # includes the bounce and wrap rows


@pytest.mark.parametrize("math_mode", MATH_MODES)
@pytest.mark.parametrize("name", ["upright", "hanging", "near_edge", "fast", "random0", "random1", "random2", "random3"])
def test_c2_rollouts_costs_update(golden_dir, name, math_mode):
    g = load(golden_dir, "rollouts_c2.npz")
    N, H = int(g["N"]), int(g["H"])
    kn, du = regen_delta_u(g[f"{name}/seed"], N, H, g["stdev"])
    s0, u_nom, u_prev, target = g[f"{name}/s0"], g[f"{name}/u_nom"], g[f"{name}/u_prev"], float(g[f"{name}/target"])
    THL = float(O.DEFAULT_PARAMS.TrackHalfLength)

    # ---- predictor seam: trajectories
    eng = engine(1, N, H, math_mode=math_mode, shift_mode="none", control_mode="penalise", correction_u="u_nom")
    u_run = (u_nom + du).astype(f32)
    traj = eng.predict(s0, u_run).cpu().numpy()
    head, final, final_B = g[f"{name}/raw/traj_head"], g[f"{name}/raw/final"], g[f"{name}/raw/final_B"]
    ref_traj = O.predict_core(s0, u_run)                                  # oracle, full tensor (mode A; pinned to `final`)
    ref_traj_B = O.predict_core(s0, u_run, mode="f64sub")                 # mode B (pinned to `final_B`)
    assert np.array_equal(ref_traj[:, -1], final) or np.abs(ref_traj[:, -1] - final).max() < 2e-6
    flagged = PU.flag_discontinuities(ref_traj)
    # final states after 50 control steps: every clear rollout inside the band around the reference's [A, B] interval
    PU.assert_states(traj[:, -1], final, final_B, flagged, f"{name} final states", strict=True)
    # ... and the whole first half of the horizon for the rollouts the golden holds in full
    nh = head.shape[0]
    PU.assert_states(traj[:nh, :H // 2], head[:, :H // 2], ref_traj_B[:nh, :H // 2], flagged[:nh], f"{name} trajectory heads", strict=True)

    # ---- cost seam on the ORACLE's trajectories (isolates the cost arithmetic from integration differences)
    for cost_name, key in (("quadratic_boundary_grad_minimal", "S_qbgm"), ("default", "S_default")):
        eng.set_cost(cost_name)
        _, _, total = eng.trajectory_cost(ref_traj, u_run, target, 1.0)
        np.testing.assert_allclose(total.cpu().numpy(), g[f"{name}/raw/{key}"], rtol=1e-4)
    eng.set_cost("quadratic_boundary_grad_minimal")
    stage, _, _ = eng.trajectory_cost(ref_traj[:32], u_run[:32], target, 1.0, want=("stage",))
    np.testing.assert_allclose(stage.cpu().numpy(), g[f"{name}/raw/stage_qbgm_head"], rtol=1e-4, atol=1e-5)

    # ---- fused step, plugin cost, both input conventions; compare S and the soft-min update
    for tag, control_mode in (("raw", "penalise"), ("clip", "clip")):
        e2 = engine(1, N, H, math_mode=math_mode, shift_mode="none", control_mode=control_mode, correction_u="u_nom",
                    cc_weight=0.0, rollouts_per_lane=(2 if (math_mode == "fast" and tag == "clip") else 1))       # cc_weight 0: S is the plugin trajectory cost alone (what the golden holds)
        un = e2.tensor(u_nom[None].copy())
        S = e2.empty(1, N)
        e2.step(s0[None], un, target, 1.0, delta_u=du[None], S_out=S)
        S = S.cpu().numpy()[0]
        S_ref = g[f"{name}/{tag}/S_qbgm"]
        uc = np.clip(u_run, -1, 1).astype(f32) if control_mode == "clip" else u_run
        tr_a, tr_b = O.predict_core(s0, uc), O.predict_core(s0, uc, mode="f64sub")
        S_b = O.trajectory_cost(O.COST_QBGM, tr_b, uc, f32(target), f32(1.0))
        PU.assert_costs(S, S_ref, S_b, PU.flag_discontinuities(tr_a), f"{name}/{tag} S_qbgm", strict=True)
        u_new_ref = u_nom + O.reward_weighted_average(S_ref, du)
        if control_mode == "clip":
            u_new_ref = np.clip(u_new_ref, -1, 1)
        np.testing.assert_allclose(un.cpu().numpy()[0], u_new_ref, atol=1e-4)

    # ---- fused step, legacy cost (q + phi + 1e5 penalty + ccrc) against the reference's own S and u
    from cartpolesimulation_amd.configs import legacy_mppi_config
    from cartpolesimulation_amd.engine import MPPIEngine
    e3 = MPPIEngine(1, legacy_mppi_config(num_rollouts=N, mpc_horizon=H, shift_mode="none", math_mode=math_mode,
                                          rollouts_per_lane=2 if math_mode == "fast" else 1))
    un = e3.tensor(u_nom[None].copy())
    S = e3.empty(1, N)
    e3.step(s0[None], un, target, 1.0, delta_u=du[None], u_prev=u_prev[None], S_out=S)
    S, S_ref = S.cpu().numpy()[0], g[f"{name}/S_legacy"]
    cfg_l = O.MPPIConfig(N=N, H=H, cost_id=O.COST_LEGACY, shift_mode="none")
    S_b = O.legacy_rollout_costs(s0, u_nom, du, u_prev, f32(target), cfg_l, mode="f64sub")
    S_b = S_b[0] if isinstance(S_b, tuple) else S_b
    fl = PU.flag_discontinuities(ref_traj) | PU.flag_indicators(ref_traj, "legacy", target)
    PU.assert_costs(S, S_ref, S_b, fl, f"{name} S_legacy", strict=True)
    np.testing.assert_allclose(un.cpu().numpy()[0], g[f"{name}/u_new_legacy"], atol=1e-4)


@pytest.mark.parametrize("math_mode", MATH_MODES)
@pytest.mark.parametrize("case,e", [("c3", 0), ("c3", 1), ("c4", 0), ("c4", 1)])
def test_c3_c4_shape_rollouts_costs_update(golden_dir, case, e, math_mode):
    """The HIP path against the REFERENCE at the shape of BASELINE configs[2] / configs[3] (tests/golden/rollouts_c3c4.npz: the
    reference's own next_state_predictor_ODE_v0 with the env's pole length as `variable_parameters.L`, H = 100 / 50 from 11 / 6 SFC64
    knots, quadratic_boundary_grad_minimal, reward_weighted_average): predictor seam and fused step, both math modes, both lane
    mappings, rule ODE_V0, strict."""
    g = load(golden_dir, "rollouts_c3c4.npz")
    key = f"{case}/{e}"
    N, H = int(g["N"]), int(g[f"{case}/H"])
    kn, du = regen_delta_u(g[f"{key}/seed"], N, H, g["stdev"])
    assert np.array_equal(du[:4], g[f"{key}/delta_u_head"])
    s0, target, Lv = g[f"{key}/s0"], float(g[f"{key}/target"]), g[f"{key}/L"]
    # ---- predictor seam
    eng = engine(1, N, H, math_mode=math_mode, shift_mode="none", control_mode="penalise", correction_u="u_nom")
    u_run = du.astype(f32)
    traj = eng.predict(s0, u_run, L=Lv).cpu().numpy()
    ref_traj = O.predict_core(s0, u_run, L=Lv)                             # oracle, mode A (pinned to `final` by tests/test_oracle_c3c4.py)
    ref_traj_B = O.predict_core(s0, u_run, L=Lv, mode="f64sub")
    flagged = PU.flag_discontinuities(ref_traj)
    PU.assert_states(traj[:, -1], g[f"{key}/raw/final"], g[f"{key}/raw/final_B"], flagged, f"{key} final states", strict=True)
    head = g[f"{key}/raw/traj_head"]
    nh = head.shape[0]
    PU.assert_states(traj[:nh, :H // 2], head[:, :H // 2], ref_traj_B[:nh, :H // 2], flagged[:nh], f"{key} trajectory heads", strict=True)
    # ---- cost seam on the ORACLE's trajectories, both plugins (isolates the cost arithmetic from integration differences)
    for cost_name, ckey in (("quadratic_boundary_grad_minimal", "S_qbgm"), ("default", "S_default")):
        eng.set_cost(cost_name)
        _, _, total = eng.trajectory_cost(ref_traj, u_run, target, 1.0)
        np.testing.assert_allclose(total.cpu().numpy(), g[f"{key}/raw/{ckey}"], rtol=1e-4)
    # ---- fused step with the `default` plugin on the device's own rollouts (its indicator terms flag the rollouts that graze a threshold)
    e_d = engine(1, N, H, math_mode=math_mode, shift_mode="none", control_mode="penalise", correction_u="u_nom", cc_weight=0.0,
                 cost_function_specification="default", rollouts_per_lane=1)
    un_d, S_dv = e_d.zeros(1, H), e_d.empty(1, N)
    e_d.step(s0[None], un_d, target, 1.0, L=np.array([Lv], f32), delta_u=du[None], S_out=S_dv)
    S_db = O.trajectory_cost(O.COST_DEFAULT, ref_traj_B, u_run, f32(target), f32(1.0))
    fl_d = flagged | PU.flag_indicators(ref_traj, "default", target)
    PU.assert_costs(S_dv.cpu().numpy()[0], g[f"{key}/raw/S_default"], S_db, fl_d, f"{key} S_default", strict=True)
    e_d.close()
    # ---- fused step on the device's own rollouts: S and the soft-min update against the reference's
    for tag, control_mode in (("raw", "penalise"), ("clip", "clip")):
        for rpl in ((1, 2) if math_mode == "fast" else (1,)):
            e2 = engine(1, N, H, math_mode=math_mode, shift_mode="none", control_mode=control_mode, correction_u="u_nom", cc_weight=0.0,
                        rollouts_per_lane=rpl)
            un, S = e2.zeros(1, H), e2.empty(1, N)
            e2.step(s0[None], un, target, 1.0, L=np.array([Lv], f32), delta_u=du[None], S_out=S)
            S = S.cpu().numpy()[0]
            uc = np.clip(u_run, -1, 1).astype(f32) if control_mode == "clip" else u_run
            tr_a, tr_b = O.predict_core(s0, uc, L=Lv), O.predict_core(s0, uc, L=Lv, mode="f64sub")
            S_b = O.trajectory_cost(O.COST_QBGM, tr_b, uc, f32(target), f32(1.0))
            PU.assert_costs(S, g[f"{key}/{tag}/S_qbgm"], S_b, PU.flag_discontinuities(tr_a), f"{key}/{tag} S_qbgm (rpl {rpl})", strict=True)
            np.testing.assert_allclose(un.cpu().numpy()[0], g[f"{key}/{tag}/u_new"], atol=1e-4)
            # ... and from the KNOTS (in-kernel interpolation: what the Philox path does with its own draws)
            un2 = e2.zeros(1, H)
            e2.step(s0[None], un2, target, 1.0, L=np.array([Lv], f32), knots=kn[None])
            np.testing.assert_allclose(un2.cpu().numpy()[0], g[f"{key}/{tag}/u_new"], atol=1e-4)
            e2.close()
    eng.close()


LANE_MODES = [("precise", 1), ("fast", 1), ("fast", 2)]      # (math_mode, rollouts_per_lane)


@pytest.mark.parametrize("math_mode,rpl", LANE_MODES)
def test_noise_sources_agree(math_mode, rpl):
    """delta_u buffer == in-kernel interpolation of the same knots == in-kernel Philox of the same (seed, offset) == the
    tiled delta_u buffer (N = 700 and H = 35 are multiples of neither the 64-row group nor the 4-step quad)."""
    E, N, H = 3, 700, 35
    eng = engine(E, N, H, math_mode=math_mode, rollouts_per_lane=rpl)
    rng = Generator(SFC64(7))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-3, 3), rng.uniform(-5, 5), rng.uniform(-0.15, 0.15),
                                           rng.uniform(-0.3, 0.3)) for _ in range(E)])
    tp = rng.uniform(-0.1, 0.1, E).astype(f32)
    te = np.ones(E, dtype=f32)
    Lv = rng.uniform(0.2, 0.5, E).astype(f32)
    u0 = (0.2 * rng.standard_normal((E, H))).astype(f32)
    kn, du = eng.sample(seed=1234, offset=5, env_offset=11, knots=True, delta_u=True)
    # interpolation of CALLER knots == oracle interpolation (bit-exact: float64 slope, float32 store, as scipy does)
    du64 = eng.interpolate(kn)
    assert np.array_equal(du64.cpu().numpy().reshape(E * N, H), O.interpolate_knots(kn.cpu().numpy().reshape(E * N, -1), H))
    # the sampler's own delta_u: PRECISE uses the same float64 form; FAST uses one float32 FMA (<= 1 ulp away)
    if math_mode == "precise":
        assert np.array_equal(du64.cpu().numpy(), du.cpu().numpy())
    else:
        np.testing.assert_allclose(du.cpu().numpy(), du64.cpu().numpy(), rtol=0, atol=1.2e-7)
    # the tiled layout holds the same values: device sampler straight into it == re-tiled reference-layout buffer
    tiled = eng.sample_tiled(seed=1234, offset=5, env_offset=11)
    assert np.array_equal(eng.untile(tiled).cpu().numpy(), du.cpu().numpy())
    assert np.array_equal(eng.tile_delta_u(du).cpu().numpy(), tiled.cpu().numpy())            # padding included (zeros)
    assert np.array_equal(eng.untile(eng.sample_tiled(knots=kn)).cpu().numpy(), du64.cpu().numpy())
    outs = []
    for kw in (dict(delta_u=du), dict(knots=kn), dict(seed=1234, offset=5, env_offset=11), dict(delta_u_tiled=tiled)):
        un = eng.tensor(u0.copy())
        S = eng.empty(E, N)
        Q, _ = eng.step(s0, un, tp, te, L=Lv, S_out=S, **kw)
        outs.append((un.cpu().numpy(), S.cpu().numpy(), Q.cpu().numpy()))
    # identical perturbations -> identical costs: sampler buffer == in-kernel Philox always; caller knots (float64
    # interpolation) == both in PRECISE, and within the 1-ulp interpolation difference in FAST
    assert np.array_equal(outs[2][1], outs[0][1])
    if math_mode == "precise":
        assert np.array_equal(outs[1][1], outs[0][1])
    else:
        np.testing.assert_allclose(outs[1][1], outs[0][1], rtol=2e-5)
    assert np.array_equal(outs[3][1], outs[0][1])                  # tiled buffer: the very same perturbations
    for o in outs[1:]:
        np.testing.assert_allclose(o[0], outs[0][0], atol=5e-6)    # knot-space vs delta_u-space reduction order
        np.testing.assert_allclose(o[2], outs[0][2], atol=5e-6)


def test_philox_sampler_statistics():
    """The device sampler (a17 with a counter-based generator): knots ~ sigma * N(0,1), independent across
    rollouts / envs / knots / offsets."""
    E, N, H = 8, 4096, 50
    eng = engine(E, N, H)
    kn, _ = eng.sample(seed=99, offset=3, env_offset=5)
    z = (kn.cpu().numpy().astype(np.float64) / eng.mppi.sigma)
    n = z.size                                               # 196608 samples
    assert abs(z.mean()) < 4 / np.sqrt(n) and abs(z.std() - 1) < 4 / np.sqrt(2 * n)
    assert abs((z ** 3).mean()) < 4 * np.sqrt(15 / n) and abs((z ** 4).mean() - 3) < 4 * np.sqrt(96 / n)
    assert np.abs(z).max() < 5.8                             # 24-bit uniforms: tails end at sqrt(2*24*ln 2)
    flat = z.reshape(E * N, -1)
    c = np.corrcoef(flat.T)                                  # knots of one rollout are uncorrelated
    assert np.abs(c - np.eye(c.shape[0])).max() < 5 / np.sqrt(E * N)
    assert abs(np.corrcoef(flat[:-1, 0], flat[1:, 0])[0, 1]) < 5 / np.sqrt(E * N)   # neighbouring rollouts
    kn2, _ = eng.sample(seed=99, offset=4, env_offset=5)
    assert not np.array_equal(kn2.cpu().numpy(), kn.cpu().numpy())
    kn3, _ = eng.sample(seed=99, offset=3, env_offset=6)     # env_offset shifts the env axis of the counter
    assert np.array_equal(kn3.cpu().numpy()[:-1], kn.cpu().numpy()[1:])


@pytest.mark.parametrize("math_mode,rpl", LANE_MODES)
@pytest.mark.parametrize("flags", [
    dict(),                                                          # Control_Toolkit-flavoured defaults
    dict(horizon_reduce="mean"),
    dict(shift_mode="append_zero", correction_u="u_nom"),
    dict(cost_function_specification="default", control_mode="penalise"),
])
def test_fused_step_vs_oracle_multi_env(math_mode, rpl, flags):
    """Full optimizer step (shift, clip, cost, correction, soft-min update) for several envs with per-env L / targets."""
    E, N, H = 4, 1000, 30            # N not a multiple of the 256-thread block: ragged last block
    eng = engine(E, N, H, math_mode=math_mode, rollouts_per_lane=rpl, **flags)
    rng = Generator(SFC64(21))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-0.6, 0.6), rng.uniform(-2, 2), rng.uniform(-0.1, 0.1),
                                           rng.uniform(-0.2, 0.2)) for _ in range(E)])
    tp = rng.uniform(-0.08, 0.08, E).astype(f32)
    te = np.ones(E, dtype=f32)
    Lv = rng.uniform(0.25, 0.45, E).astype(f32)
    u0 = (0.3 * rng.standard_normal((E, H))).astype(f32)
    du = np.stack([O.sample_delta_u(rng, N, H, np.float64(eng.mppi.sigma)) for _ in range(E)])
    un = eng.tensor(u0.copy())
    S = eng.empty(E, N)
    Q, _ = eng.step(s0, un, tp, te, L=Lv, delta_u=du, S_out=S)
    un, S, Q = un.cpu().numpy(), S.cpu().numpy(), Q.cpu().numpy()
    m = eng.mppi
    cid = {"quadratic_boundary_grad_minimal": O.COST_QBGM, "default": O.COST_DEFAULT}[m.cost_function_specification]
    cfg = O.MPPIConfig(N=N, H=H, cc_weight=m.cc_weight, R=m.R, LBD=m.LBD, NU=m.NU, cost_id=cid,
                       horizon_reduce=m.horizon_reduce, control_mode=m.control_mode, shift_mode=m.shift_mode,
                       correction_u=m.correction_u)
    for e in range(E):
        ref, ref_b = PU.oracle_step_both_modes(s0[e], u0[e], du[e], tp[e], te[e], cfg, L=Lv[e])
        fl = PU.flag_discontinuities(ref["traj"]) | PU.flag_indicators(ref["traj"], m.cost_function_specification, tp[e])
        PU.assert_costs(S[e], ref["S"], ref_b["S"], fl, f"env {e} costs")
        np.testing.assert_allclose(un[e], ref["u_new"], atol=1e-4)
        np.testing.assert_allclose(Q[e], ref["Q"], atol=1e-4)


def test_reward_weighted_average_seam(golden_dir):
    h = load(golden_dir, "sampler_rwa.npz")
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    for lbd, key in ((100.0, "rwa_lbd100"), (1.0, "rwa_lbd1")):
        eng = MPPIEngine(1, MPPIConfig(num_rollouts=4, mpc_horizon=3, LBD=lbd))
        out = eng.reward_weighted_average(h["S_kat"], h["du_kat"]).cpu().numpy()[0]
        np.testing.assert_allclose(out, h[key], atol=1e-6)


def test_edge_cases():
    """N=1, H=1, H not a multiple of the period or the LDS tile, N below one wave."""
    for (N, H, rpl) in ((1, 1, 1), (5, 7, 2), (63, 17, 1), (65, 9, 2), (257, 33, 1), (513, 21, 2)):
        eng = engine(2, N, H, rollouts_per_lane=rpl)
        rng = Generator(SFC64(N * 100 + H))
        du = (0.2 * rng.standard_normal((2, N, H))).astype(f32)
        s0 = np.stack([O.create_cartpole_state(0.1, 0.2, 0.01, 0.0), O.create_cartpole_state(-2.5, 1.0, -0.1, 0.1)])
        un = eng.zeros(2, H)
        S = eng.empty(2, N)
        Q, _ = eng.step(s0, un, 0.0, 1.0, delta_u=du, S_out=S)
        cfg = O.MPPIConfig(N=N, H=H)
        for e in range(2):
            ref, ref_b = PU.oracle_step_both_modes(s0[e], np.zeros(H, f32), du[e], f32(0), f32(1), cfg)
            PU.assert_costs(S.cpu().numpy()[e], ref["S"], ref_b["S"], PU.flag_discontinuities(ref["traj"]), f"N={N} H={H} env {e}")
            np.testing.assert_allclose(un.cpu().numpy()[e], ref["u_new"], atol=1e-4)


def test_error_behaviour():
    from cartpolesimulation_amd import _lib as L
    eng = engine(2, 64, 10)
    un = eng.zeros(3, 10)                                   # E larger than the handle's capacity
    with pytest.raises(ValueError):
        eng.step(np.zeros((3, 6), f32), un, 0.0, 1.0, seed=1)
    with pytest.raises(ValueError):
        eng.step(np.zeros((2, 6), f32), eng.zeros(2, 10), 0.0, 1.0)          # no noise source
    import ctypes as C
    a = L.cpmppi_step_args()
    a.E = 5
    rc = eng.lib.cpmppi_step(eng._h, C.byref(a), None)
    assert rc == -1 and b"E out of range" in eng.lib.cpmppi_last_error(eng._h)
    with pytest.raises(ValueError):
        engine(1, 8, 8, horizon_reduce="median")


@pytest.mark.parametrize("math_mode", MATH_MODES)
def test_closed_loop_c1_plumbing(golden_dir, math_mode):
    """BASELINE config C1 (256 x 20, legacy MPPI, closed loop with the plant) on the HIP path: SFC64 knots from the
    host (identical noise seeds), rollouts/cost/update and the plant on the GPU.  Both arithmetic modes (FAST is the
    bench default) must track the reference's own closed-loop trace."""
    from cartpolesimulation_amd.configs import legacy_mppi_config
    from cartpolesimulation_amd.engine import MPPIEngine
    g = load(golden_dir, "closed_loop_c1.npz")
    N, H = int(g["N"]), int(g["H"])
    cfg = legacy_mppi_config(num_rollouts=N, mpc_horizon=H, math_mode=math_mode)
    eng = MPPIEngine(1, cfg)
    rng = Generator(SFC64(int(g["seed"])))
    for _ in range(5):
        rng.uniform(-1.0, 1.0)
    s = eng.tensor(g["s"][0][None].copy())
    un = eng.zeros(1, H)
    for c in range(g["s"].shape[0]):
        s_host = s.cpu().numpy()[0]
        if c < 10:
            np.testing.assert_allclose(s_host, g["s"][c], atol=2e-4, rtol=1e-4)
        kn = O.sample_knots(rng, N, H, np.float64(g["stdev"]))
        Qd, _ = eng.step(s, un, float(g["target"]), 1.0, knots=kn[None])
        Q = f32(Qd.cpu().numpy()[0] * (1 + float(g["p_Q"]) * rng.uniform(-1.0, 1.0)))     # :553
        Q = np.clip(Q, f32(-1), f32(1))
        if c < 10:
            np.testing.assert_allclose(Q, g["Q"][c], atol=1e-4)
            np.testing.assert_allclose(un.cpu().numpy()[0], g["u_updated"][c], atol=1e-4)
        eng.plant_advance(s, np.array([Q], dtype=f32), n_substeps=10, dt_sim=0.002)
    s_host = s.cpu().numpy()[0]
    assert abs(s_host[O.ANGLE_IDX]) < 0.2 and abs(s_host[O.POSITION_IDX]) < 0.198


@pytest.mark.parametrize("math_mode,rpl", LANE_MODES)
@pytest.mark.parametrize("shape", ["256x20", "1024x50"])
def test_legacy_controller_step_traces(golden_dir, shape, math_mode, rpl):
    """Full `controller_mppi_cartpole.step` traces of the reference itself at exactly the C1 / C2 size (seed -> delta_u
    -> S -> u -> Q, three consecutive steps): the same SFC64 knots go to the GPU, the reference's own per-rollout costs
    S, updated sequence u and applied control Q come back."""
    from cartpolesimulation_amd.configs import legacy_mppi_config
    from cartpolesimulation_amd.engine import MPPIEngine
    g = load(golden_dir, f"legacy_step_{shape}.npz")
    N, H, target = int(g["N"]), int(g["H"]), float(g["target"])
    eng = MPPIEngine(1, legacy_mppi_config(num_rollouts=N, mpc_horizon=H, math_mode=math_mode, rollouts_per_lane=rpl))
    rng = Generator(SFC64(int(g["seed"])))
    for _ in range(5):
        rng.uniform(-1.0, 1.0)                                                # configure()'s draws (:355-359)
    un, u_prev = eng.zeros(1, H), eng.zeros(1, H)
    cfg = O.MPPIConfig(N=N, H=H, SQRTRHOINV=0.02, cost_id=O.COST_LEGACY, control_mode="penalise", shift_mode="append_zero",
                       correction_u="u_nom")
    u_host = np.zeros(H, f32)                                                 # the controller's u as the oracle carries it
    for it in range(g["s_seq"].shape[0]):
        s = g["s_seq"][it]
        kn = O.sample_knots(rng, N, H, np.float64(g["stdev"]))
        du = O.interpolate_knots(kn, H)
        assert abs(du.astype(np.float64).sum() - g["delta_u_sum64"][it]) < 1e-9      # identical noise, by construction
        S = eng.empty(1, N)
        Qd, _ = eng.step(s[None], un, target, 1.0, knots=kn[None], u_prev=u_prev, S_out=S)
        S_b, _ = O.legacy_rollout_costs(s, u_host, du, u_prev.cpu().numpy()[0], f32(target), cfg, mode="f64sub")
        _, traj = O.legacy_rollout_costs(s, u_host, du, u_prev.cpu().numpy()[0], f32(target), cfg)
        fl = PU.flag_discontinuities(traj) | PU.flag_indicators(traj, "legacy", target)
        PU.assert_costs(S.cpu().numpy()[0], g["S"][it], S_b, fl, f"{shape} step {it} S", strict=True)
        np.testing.assert_allclose(un.cpu().numpy()[0], g["u_updated"][it], atol=1e-4)
        Q = f32(Qd.cpu().numpy()[0] * (1 + float(g["p_Q"]) * rng.uniform(-1.0, 1.0)))      # :553
        np.testing.assert_allclose(np.clip(Q, f32(-1), f32(1)), g["Q"][it], atol=1e-4)
        u_prev.copy_(un)                                                      # :558
        u_host = np.concatenate([g["u_updated"][it][1:], np.zeros(1, f32)])   # :561-562 (the GPU shifts at its next step)


def test_limits_and_bad_arguments():
    """Maximum horizon, many envs, unsupported knot counts, misaligned pointers."""
    import ctypes as C
    from cartpolesimulation_amd import _lib as L
    # H = CPMPPI_MAX_HORIZON (1024) with one knot period per 32 steps; a single block per env
    H = 1024
    eng = engine(1, 64, H, period_interpolation_inducing_points=32, intermediate_steps=1, mpc_timestep=0.002)
    rng = Generator(SFC64(1))
    du = (0.1 * rng.standard_normal((1, 64, H))).astype(f32)
    s0 = O.create_cartpole_state(0.1, 0.0, 0.0, 0.0)[None]
    un = eng.zeros(1, H)
    S = eng.empty(1, 64)
    eng.step(s0, un, 0.0, 1.0, delta_u=du, S_out=S)
    cfg = O.MPPIConfig(N=64, H=H, S=1, dt=0.002, period=32)
    ref, ref_b = PU.oracle_step_both_modes(s0[0], np.zeros(H, f32), du[0], f32(0), f32(1), cfg)
    PU.assert_costs(S.cpu().numpy()[0], ref["S"], ref_b["S"], PU.flag_discontinuities(ref["traj"], dt=0.002), "H = 1024")
    np.testing.assert_allclose(un.cpu().numpy()[0], ref["u_new"], atol=1e-4)
    with pytest.raises(L.CpmppiError):
        engine(1, 8, 1025)                                                    # beyond the maximum horizon
    # many small envs in one launch; every env gets its own noise stream and a finite answer
    E = 5000
    e2 = engine(E, 64, 10)
    u2 = e2.zeros(E, 10)
    Q, _ = e2.step(np.tile(s0, (E, 1)), u2, 0.0, 1.0, seed=3)
    q = Q.cpu().numpy()
    assert np.isfinite(q).all() and len(np.unique(np.round(q, 7))) > E * 0.9
    # the device sampler stages [256][P+1] knots in LDS: 101 knots (H = 100, one per step) fit the 160 KB of gfx950 and
    # match the oracle's interpolation; more than 154 per rollout -> clean error, not a crash
    e3 = engine(1, 8, 100, period_interpolation_inducing_points=1)
    kn3, du3 = e3.sample(seed=1, knots=True, delta_u=True)
    np.testing.assert_allclose(du3.cpu().numpy()[0], O.interpolate_knots(kn3.cpu().numpy()[0], 100, period=1), atol=2e-7)
    e4 = engine(1, 8, 200, period_interpolation_inducing_points=1)
    with pytest.raises(L.CpmppiError):
        e4.sample(seed=1)
    # a pointer that is not 4-byte aligned is refused
    a = L.cpmppi_step_args()
    buf = e2.zeros(64)
    a.E, a.s0, a.u_nom = 1, buf.data_ptr() + 1, buf.data_ptr()
    a.target_position = a.target_equilibrium = buf.data_ptr()
    a.noise_kind = L.NOISE_PHILOX
    assert e2.lib.cpmppi_step(e2._h, C.byref(a), None) == -5
