"""Pin the plain-C oracle (oracle/cpmppi_oracle.c) to the golden vectors and to the numpy oracle.  CPU-only."""
import os

import numpy as np
import pytest
from numpy.random import SFC64, Generator

from oracle import oracle_np as O
from oracle import oracle_c as OC

f32 = np.float32


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def regen_delta_u(seed, N, H, stdev):
    rng = Generator(SFC64(int(seed)))
    for _ in range(5):
        rng.uniform(-1.0, 1.0)
    return O.sample_delta_u(rng, N, H, np.float64(stdev))


@pytest.mark.parametrize("mode,key", [("f32", "step1_A"), ("f64sub", "step1_B")])
def test_c_single_step_kats(golden_dir, mode, key):
    g = load(golden_dir, "kat_step.npz")
    c = OC.make_config(O.MPPIConfig(N=1, H=1), mode=mode)
    out = OC.predict(c, g["s_in"], g["Q_in"][:, None], L=g["L_in"])[:, 1]
    # (float)cos((double)x) vs numpy's float32 SIMD kernels: <= 1 ulp apart on a few inputs, amplified by 10 substeps
    np.testing.assert_allclose(out, g[key], rtol=3e-6, atol=3e-6)


@pytest.mark.parametrize("name", ["upright", "near_edge", "fast", "random1"])
def test_c_rollouts_and_costs(golden_dir, name):
    g = load(golden_dir, "rollouts_c2.npz")
    N, H = int(g["N"]), int(g["H"])
    du = regen_delta_u(g[f"{name}/seed"], N, H, g["stdev"])
    s0, u_nom, u_prev, target = g[f"{name}/s0"], g[f"{name}/u_nom"], g[f"{name}/u_prev"], g[f"{name}/target"]
    base = dict(N=N, H=H, shift_mode="none", correction_u="u_nom")
    c = OC.make_config(O.MPPIConfig(**base))
    traj = OC.predict(c, s0, (u_nom + du).astype(f32))
    final = g[f"{name}/raw/final"]
    ok = (np.abs(traj[:, -1] - final) <= 1e-4 + 1e-4 * np.abs(final)).all(axis=1)
    assert ok.mean() >= 0.97
    for tag, cm in (("raw", "penalise"), ("clip", "clip")):
        for cid, key in ((O.COST_QBGM, "S_qbgm"), (O.COST_DEFAULT, "S_default")):
            cfg = O.MPPIConfig(cost_id=cid, control_mode=cm, cc_weight=0.0, **base)
            u_new, Q, S = OC.step(OC.make_config(cfg), s0[None], u_nom[None], du[None], target, 1.0)
            rel = np.abs(S[0] - g[f"{name}/{tag}/{key}"]) / np.abs(g[f"{name}/{tag}/{key}"])
            assert np.median(rel) < 2e-5 and (rel < 2e-3).mean() >= 0.97
    cfg = O.MPPIConfig(cost_id=O.COST_LEGACY, control_mode="penalise", **base)
    u_new, Q, S = OC.step(OC.make_config(cfg), s0[None], u_nom[None], du[None], target, 1.0, u_prev=u_prev[None])
    rel = np.abs(S[0] - g[f"{name}/S_legacy"]) / np.abs(g[f"{name}/S_legacy"])
    assert np.median(rel) < 2e-5 and (rel < 2e-3).mean() >= 0.95
    np.testing.assert_allclose(u_new[0], g[f"{name}/u_new_legacy"], atol=1e-4)


@pytest.mark.parametrize("flags", [dict(), dict(horizon_reduce="mean"), dict(shift_mode="append_zero"),
                                   dict(cost_id=O.COST_DEFAULT, correction_u="u_nom")])
def test_c_matches_numpy_oracle(flags):
    E, N, H = 3, 300, 25
    rng = Generator(SFC64(5))
    cfg = O.MPPIConfig(N=N, H=H, **flags)
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-1, 1), rng.uniform(-3, 3), rng.uniform(-0.1, 0.1),
                                           rng.uniform(-0.2, 0.2)) for _ in range(E)])
    u0 = (0.3 * rng.standard_normal((E, H))).astype(f32)
    du = np.stack([O.sample_delta_u(rng, N, H, np.float64(cfg.stdev)) for _ in range(E)])
    tp, te, Lv = rng.uniform(-0.05, 0.05, E).astype(f32), np.ones(E, f32), rng.uniform(0.2, 0.5, E).astype(f32)
    u_new, Q, S = OC.step(OC.make_config(cfg), s0, u0, du, tp, te, L=Lv)
    for e in range(E):
        ref = O.mppi_step(s0[e], u0[e], du[e], tp[e], te[e], cfg, L=Lv[e])
        rel = np.abs(S[e] - ref["S"]) / np.abs(ref["S"])
        assert np.median(rel) < 2e-5
        np.testing.assert_allclose(u_new[e], ref["u_new"], atol=2e-5)
        np.testing.assert_allclose(Q[e], ref["Q"], atol=2e-5)


def test_c_threads_are_deterministic():
    N, H = 256, 10
    rng = Generator(SFC64(3))
    cfg = OC.make_config(O.MPPIConfig(N=N, H=H))
    du = O.sample_delta_u(rng, N, H, 0.2)[None]
    s0 = O.create_cartpole_state(0.2, 0.0, 0.0, 0.0)[None]
    a = OC.step(cfg, s0, np.zeros((1, H), f32), du, 0.0, 1.0, n_threads=1)
    b = OC.step(cfg, s0, np.zeros((1, H), f32), du, 0.0, 1.0, n_threads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
