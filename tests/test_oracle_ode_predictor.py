"""Pin the oracle's restatement of predictor_type "ODE" (next_state_predictor_ODE: Euler-Cromer substeps, no edge bounce,
angle = atan2(sin, cos); oracle_np.ode_step / fine_integration_cromer and the C oracle's integrator = 1) to the outputs of
the reference's own class (tests/golden/ode_predictor.npz, oracle/gen_golden_ode.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import oracle_np as O
from oracle import oracle_c as OC

f32 = np.float32
REGIMES = ["upright", "hanging", "edge", "spin", "fastspin"]


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "ode_predictor.npz"))


@pytest.mark.parametrize("key,kw", [("s_next", {}), ("s_next_L030", dict(L=0.30)), ("s_next_mpole", dict(m_pole=0.12)),
                                    ("s_next_dt04_S4", dict(dt=0.04, S=4))])
def test_single_control_step(g, key, kw):
    """numpy against numpy: bit for bit on the machine that made the fixture, a few float32 ulps elsewhere (libm's cos / sin)."""
    out = O.ode_step(g["kat/s"], g["kat/Q"], **kw)
    np.testing.assert_allclose(out, g[f"kat/{key}"], rtol=2e-6, atol=2e-6)
    assert (np.abs(g["kat/s"][:, O.POSITION_IDX]) > O.DEFAULT_PARAMS.TrackHalfLength).sum() > 20      # states beyond the edge are in
    # ... and they do NOT bounce: the cart's velocity keeps its sign where the control does not reverse it
    beyond = np.abs(g["kat/s"][:, O.POSITION_IDX]) > O.DEFAULT_PARAMS.TrackHalfLength
    x0, x1 = g["kat/s"][beyond, O.POSITION_IDX], g[f"kat/{key}"][beyond, O.POSITION_IDX]
    assert np.abs(x1 - x0).max() < 0.05


@pytest.mark.parametrize("name", REGIMES)
def test_rollouts(g, name):
    traj = O.predict_core(g[f"{name}/s0"], g[f"{name}/Q"], integrator="ODE")
    ref = g[f"{name}/traj"]
    # 50 control steps of a chaotic system: rounding-level differences of libm grow; on the generating machine this is exact
    np.testing.assert_allclose(traj[:, :6], ref[:, :6], rtol=5e-6, atol=5e-6)
    angle_err = np.abs(np.angle(np.exp(1j * (traj[..., O.ANGLE_IDX].astype(np.float64) - ref[..., O.ANGLE_IDX]))))
    assert angle_err.max() < (5e-2 if name in ("spin", "fastspin") else 2e-3)
    np.testing.assert_allclose(traj[..., O.POSITION_IDX], ref[..., O.POSITION_IDX], atol=2e-4)
    # angle = atan2(sin, cos): always inside (-pi, pi]
    assert np.abs(ref[:, 1:, O.ANGLE_IDX]).max() <= np.pi + 1e-6


@pytest.mark.parametrize("name", REGIMES)
def test_c_oracle_matches(g, name):
    """The plain-C restatement (what the GPU parity tests at size use) against the same reference rollouts."""
    Q = g[f"{name}/Q"]
    N, H = Q.shape
    cfg = OC.make_config(O.MPPIConfig(N=N, H=H, integrator="ODE"))
    traj = OC.predict(cfg, np.tile(g[f"{name}/s0"], (N, 1)), Q)
    ref = g[f"{name}/traj"]
    np.testing.assert_allclose(traj[:, :6], ref[:, :6], rtol=5e-6, atol=5e-6)
    np.testing.assert_allclose(traj[..., O.POSITION_IDX], ref[..., O.POSITION_IDX], atol=2e-4)
    angle_err = np.abs(np.angle(np.exp(1j * (traj[..., O.ANGLE_IDX].astype(np.float64) - ref[..., O.ANGLE_IDX]))))
    assert angle_err.max() < (5e-2 if name in ("spin", "fastspin") else 2e-3)


def test_differs_from_ode_v0(g):
    """The two in-tree ODE predictors are different integrators: ~1e-3 apart after one control step."""
    a = O.ode_step(g["kat/s"], g["kat/Q"])
    inside = np.abs(a[:, O.POSITION_IDX]) < 0.15
    b = O.ode_v0_step(g["kat/s"], g["kat/Q"])
    d = np.abs(a - b)[inside][:, [O.ANGLED_IDX, O.POSITION_IDX, O.POSITIOND_IDX]]
    assert 1e-4 < d.max() < 0.2


def test_mppi_step_with_the_ode_predictor():
    """mppi_step / the C oracle's step with integrator = ODE agree with each other (cost + update on Euler-Cromer rollouts)."""
    rng = np.random.Generator(np.random.SFC64(5))
    N, H = 200, 20
    cfg = O.MPPIConfig(N=N, H=H, integrator="ODE")
    s0 = O.create_cartpole_state(0.3, -0.5, 0.05, 0.1)
    u0 = (0.2 * rng.standard_normal(H)).astype(f32)
    du = O.sample_delta_u(rng, N, H, np.float64(cfg.stdev))
    ref = O.mppi_step(s0, u0, du, 0.02, 1.0, cfg)
    u_c, Q_c, S_c = OC.step(OC.make_config(cfg), s0[None], u0[None], du[None], np.array([0.02], f32), np.array([1.0], f32))
    np.testing.assert_allclose(S_c[0], ref["S"], rtol=2e-5)
    np.testing.assert_allclose(u_c[0], ref["u_new"], atol=2e-5)
    other = O.mppi_step(s0, u0, du, 0.02, 1.0, O.MPPIConfig(N=N, H=H))
    assert np.abs(other["S"] - ref["S"]).max() > 1e-3 * np.abs(ref["S"]).max()
