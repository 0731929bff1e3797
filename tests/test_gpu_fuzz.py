"""GPU: randomised differential parity of the fused step against the C oracle (a small cut of tools/fuzz_parity.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle_np as O  # noqa: E402
from oracle import oracle_c as OC  # noqa: E402
import parity_util as PU  # noqa: E402


@pytest.mark.parametrize("math,rpl", [("fast", 1), ("fast", 2), ("precise", 1)])
def test_random_instances_against_the_c_oracle(math, rpl):
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    E, N, H, THL = 32, 1024, 50, 0.198
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, math_mode=math, rollouts_per_lane=rpl))
    ocfg = O.MPPIConfig(N=N, H=H)
    rng = np.random.Generator(np.random.SFC64(2024))
    for b in range(2):
        ang = rng.uniform(-np.pi, np.pi, E)
        s0 = np.zeros((E, 6), np.float32)
        s0[:, 0], s0[:, 1] = ang, rng.uniform(-12, 12, E)
        s0[:, 2], s0[:, 3] = np.cos(ang), np.sin(ang)
        s0[:, 4], s0[:, 5] = rng.uniform(-0.9, 0.9, E) * THL, rng.uniform(-0.6, 0.6, E)
        tp = (rng.uniform(-0.8, 0.8, E) * THL).astype(np.float32)
        te = np.where(rng.uniform(size=E) < 0.8, 1.0, -1.0).astype(np.float32)
        Lv = rng.uniform(0.2, 0.5, E).astype(np.float32)
        u0 = np.clip(0.3 * rng.standard_normal((E, H)), -1, 1).astype(np.float32)
        _, du = eng.sample(seed=50 + b, offset=b, knots=False, delta_u=True)
        un = eng.tensor(u0.copy())
        S = eng.empty(E, N)
        eng.step(s0, un, tp, te, L=Lv, delta_u=du, S_out=S)
        duh = du.cpu().numpy()
        ref = PU.c_oracle_step_with_flags(ocfg, s0, u0, duh, tp, te, L=Lv)
        Sh, uh = S.cpu().numpy(), un.cpu().numpy()
        for e in range(E):
            PU.assert_costs(Sh[e], ref["S_a"][e], ref["S_b"][e], ref["flags"][e], f"batch {b} env {e} costs")
            PU.assert_controls(uh[e], ref["u_a"][e], ref["u_b"][e], f"batch {b} env {e} u_nom",
                               allowance=PU.softmin_allowance(ref["S_a"][e], ref["S_b"][e], duh[e]))


@pytest.mark.parametrize("noise", ["delta_u", "philox"])
def test_maximum_horizon(noise):
    """H = 1024 (the ABI's maximum), 104 knots: launches, and with a given buffer matches the C oracle."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    E, N, H = 2, 96, 1024
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H))
    rng = np.random.Generator(np.random.SFC64(6))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-0.5, 0.5), rng.uniform(-1, 1), rng.uniform(-0.1, 0.1), 0.0) for _ in range(E)])
    tp, te = np.zeros(E, np.float32), np.ones(E, np.float32)
    un = eng.zeros(E, H)
    S = eng.empty(E, N)
    if noise == "philox":
        Q, _ = eng.step(s0, un, tp, te, S_out=S, seed=4, offset=0)
        assert torch.isfinite(S).all() and torch.isfinite(un).all() and float(un.abs().max()) <= 1.0
        return
    _, du = eng.sample(seed=4, offset=0, knots=False, delta_u=True)
    eng.step(s0, un, tp, te, S_out=S, delta_u=du)
    # 10 240 substeps of a chaotic system: the reference's own two arithmetic modes diverge on many rollouts, and the
    # allowance (band + their gap, from the oracle) follows; unflagged rollouts must all be inside it
    ref = PU.c_oracle_step_with_flags(O.MPPIConfig(N=N, H=H), s0, np.zeros((E, H), np.float32), du.cpu().numpy(), tp, te)
    rel = np.abs(S.cpu().numpy() - ref["S_a"]) / np.abs(ref["S_a"])
    assert np.median(rel) < 1e-3
    for e in range(E):
        PU.assert_costs(S.cpu().numpy()[e], ref["S_a"][e], ref["S_b"][e], ref["flags"][e], f"H=1024 env {e} costs", rtol=1e-3)


@pytest.mark.parametrize("S,dt,period,H", [(1, 0.02, 10, 30), (3, 0.01, 4, 23), (7, 0.03, 50, 20), (16, 0.02, 1, 12)])
@pytest.mark.parametrize("rpl", [1, 2])
def test_other_discretisations(S, dt, period, H, rpl):
    """Substep counts, control periods and knot spacings other than the shipped 10 / 0.02 s / 10 (incl. a knot period
    longer than the horizon and one knot per step): the fused step against the C oracle on the sampler's perturbations."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    E, N = 3, 257
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, intermediate_steps=S, mpc_timestep=dt,
                                   period_interpolation_inducing_points=period, rollouts_per_lane=rpl))
    rng = np.random.Generator(np.random.SFC64(S * 100 + H))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-1, 1), rng.uniform(-3, 3), rng.uniform(-0.15, 0.15), rng.uniform(-0.3, 0.3))
                   for _ in range(E)])
    tp = rng.uniform(-0.05, 0.05, E).astype(np.float32)
    te = np.ones(E, np.float32)
    u0 = (0.2 * rng.standard_normal((E, H))).astype(np.float32)
    kn, du = eng.sample(seed=8, offset=1, knots=True, delta_u=True)
    assert kn.shape == (E, N, (H + period - 1) // period + 1)
    outs = []
    for kw in (dict(delta_u=du), dict(knots=kn), dict(seed=8, offset=1)):
        un = eng.tensor(u0.copy())
        S_out = eng.empty(E, N)
        eng.step(s0, un, tp, te, S_out=S_out, **kw)
        outs.append((S_out.cpu().numpy(), un.cpu().numpy()))
    ocfg = O.MPPIConfig(N=N, H=H, S=S, dt=dt, period=period)
    duh = du.cpu().numpy()
    ref = PU.c_oracle_step_with_flags(ocfg, s0, u0, duh, tp, te)
    for e in range(E):
        PU.assert_costs(outs[0][0][e], ref["S_a"][e], ref["S_b"][e], ref["flags"][e], f"S={S} dt={dt} env {e} costs")
        PU.assert_controls(outs[0][1][e], ref["u_a"][e], ref["u_b"][e], f"S={S} dt={dt} env {e} u_nom",
                           allowance=PU.softmin_allowance(ref["S_a"][e], ref["S_b"][e], duh[e]))
    for o in outs[1:]:                                     # the three noise sources describe the same perturbations
        np.testing.assert_allclose(o[0], outs[0][0], rtol=3e-5)
        np.testing.assert_allclose(o[1], outs[0][1], atol=1e-5)


@pytest.mark.parametrize("rpl", [1, 2])
def test_other_physical_parameters(rpl):
    """A different cart (heavier pole, lighter friction, stronger motor, longer track, another inertia factor): the
    folded per-env constants of the FAST path must follow (cartpole_physical_parameters.yml values are not baked in)."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig, PhysicalParameters
    f = lambda v: float(np.float32(v))  # noqa: E731
    phys = PhysicalParameters(k=f(0.4), m_cart=f(0.31), m_pole=f(0.15), g=f(9.81), J_fric=f(2.0e-4), M_fric=f(1.1), L=f(0.5),
                              u_max=f(2.5), TrackHalfLength=f(0.25))
    p = O.CartPoleParams(k=np.float32(0.4), m_cart=np.float32(0.31), m_pole=np.float32(0.15), g=np.float32(9.81),
                         J_fric=np.float32(2.0e-4), M_fric=np.float32(1.1), L=np.float32(0.5), u_max=np.float32(2.5),
                         TrackHalfLength=np.float32(0.25))
    E, N, H = 3, 512, 40
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=rpl), phys)
    rng = np.random.Generator(np.random.SFC64(12))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-1.5, 1.5), rng.uniform(-4, 4), rng.uniform(-0.2, 0.2), rng.uniform(-0.4, 0.4))
                   for _ in range(E)])
    tp, te = rng.uniform(-0.1, 0.1, E).astype(np.float32), np.ones(E, np.float32)
    Lv = np.asarray([0.5, 0.35, 0.6], np.float32)
    u0 = (0.2 * rng.standard_normal((E, H))).astype(np.float32)
    _, du = eng.sample(seed=2, offset=0, knots=False, delta_u=True)
    un = eng.tensor(u0.copy())
    S = eng.empty(E, N)
    eng.step(s0, un, tp, te, L=Lv, S_out=S, delta_u=du)
    duh = du.cpu().numpy()
    ref = PU.c_oracle_step_with_flags(O.MPPIConfig(N=N, H=H), s0, u0, duh, tp, te, L=Lv, params=p)
    for e in range(E):
        PU.assert_costs(S.cpu().numpy()[e], ref["S_a"][e], ref["S_b"][e], ref["flags"][e], f"env {e} costs")
        PU.assert_controls(un.cpu().numpy()[e], ref["u_a"][e], ref["u_b"][e], f"env {e} u_nom",
                           allowance=PU.softmin_allowance(ref["S_a"][e], ref["S_b"][e], duh[e]))


@pytest.mark.parametrize("predictor_type,extra", [("ODE_v0", []), ("ODE", ["--probes"])])
def test_random_configurations_regression(predictor_type, extra):
    """A fixed cut of tools/dev/shape_fuzz.py (60 random shape / substep / period / cost / glue / noise-source / lane-
    mapping / math-mode configurations, seed 33) must stay inside the parity rules: ragged rollout counts around wave and
    block sizes, horizons that are not multiples of the knot period or of the tile quads, every glue flag - for both in-tree
    ODE predictors (predictor_ODE with the allowance of the full-size tests: the envelope of the reference's realisations)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "dev", "shape_fuzz.py"), "--n", "60", "--seed", "33",
                        "--predictor-type", predictor_type] + extra, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    summary = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert summary == {"configurations": 60, "passed": 60, "failed": 0, "seed": 33, "predictor_type": predictor_type}, r.stdout[-3000:]
