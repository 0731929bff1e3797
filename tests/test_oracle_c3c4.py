"""CPU: the oracles pinned to the reference at the SHAPE of BASELINE configs[2] / configs[3] (tests/golden/rollouts_c3c4.npz, made by
oracle/gen_golden_c3c4.py from the reference's own next_state_predictor_ODE_v0 + quadratic_boundary_grad_minimal +
reward_weighted_average): a 100-step horizon (11 interpolation knots) and a pole length per env handed over as
`variable_parameters.L` (SI_Toolkit_ASF/ToolkitCustomization/predictors_customization_v0.py:47-54) - what the C oracle that grades
the full-size C3 / C4 runs (bench.py `verified`, tests/test_gpu_configs.py) had so far only met at H <= 50 with the default length."""
import os

import numpy as np
import pytest
from numpy.random import SFC64, Generator

from oracle import oracle_c as OC
from oracle import oracle_np as O

f32 = np.float32
CASES = [("c3", 0), ("c3", 1), ("c4", 0), ("c4", 1)]


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "rollouts_c3c4.npz"))


def regen_delta_u(seed, N, H, stdev):
    rng = Generator(SFC64(int(seed)))
    for _ in range(5):
        rng.uniform(-1.0, 1.0)                                  # configure()'s draws (controller_mppi_cartpole.py:355-359)
    return O.sample_delta_u(rng, N, H, np.float64(stdev))


def state_tol(ref):
    return 1e-4 + 1e-4 * np.abs(ref)


def inputs(g, case, e):
    key = f"{case}/{e}"
    N, H = int(g["N"]), int(g[f"{case}/H"])
    du = regen_delta_u(g[f"{key}/seed"], N, H, g["stdev"])
    return key, N, H, du, g[f"{key}/s0"], g[f"{key}/target"], g[f"{key}/L"]


def test_fixture_shape(g):
    assert list(g["cases"]) == ["c3", "c4"] and int(g["N"]) == 512 and int(g["c3/H"]) == 100 and int(g["c4/H"]) == 50
    Ls = [float(g[f"{c}/{e}/L"]) for c, e in CASES]
    assert len(set(Ls)) == 4 and all(0.2 <= x <= 0.5 and abs(x - 0.395) > 0.03 for x in Ls)       # none of them the default length
    # the reference's own two arithmetic modes differ - at H = 100 too
    assert max(np.abs(g[f"{c}/{e}/raw/final"] - g[f"{c}/{e}/raw/final_B"]).max() for c, e in CASES) > 1e-5


@pytest.mark.parametrize("case,e", CASES)
def test_numpy_oracle_at_c3_c4_shape(g, case, e):
    key, N, H, du, s0, target, Lv = inputs(g, case, e)
    assert np.array_equal(du[:4], g[f"{key}/delta_u_head"])
    assert abs(du.astype(np.float64).sum() - g[f"{key}/delta_u_sum64"]) < 1e-9
    for tag in ("raw", "clip"):
        u_run = du.astype(f32) if tag == "raw" else np.clip(du, f32(-1), f32(1)).astype(f32)
        traj = O.predict_core(s0, u_run, L=Lv)
        head = g[f"{key}/{tag}/traj_head"]
        assert traj.shape == (N, H + 1, 6)
        # bit-exact on the generating machine; elsewhere allow trig ulps amplified over the horizon
        assert np.all(np.abs(traj[:head.shape[0]] - head) <= 0.2 * state_tol(head))
        assert np.all(np.abs(traj[:, -1] - g[f"{key}/{tag}/final"]) <= 0.2 * state_tol(g[f"{key}/{tag}/final"]))
        S = O.trajectory_cost(O.COST_QBGM, traj, u_run, target, f32(1.0))
        np.testing.assert_allclose(S, g[f"{key}/{tag}/S_qbgm"], rtol=2e-5)
        S_d = O.trajectory_cost(O.COST_DEFAULT, traj, u_run, target, f32(1.0))          # the `default` plugin: stage sum + terminal indicator
        np.testing.assert_allclose(S_d, g[f"{key}/{tag}/S_default"], rtol=2e-5)
        u_new = O.reward_weighted_average(g[f"{key}/{tag}/S_qbgm"], du)
        if tag == "clip":
            u_new = np.clip(u_new, -1, 1)
        np.testing.assert_allclose(u_new, g[f"{key}/{tag}/u_new"], atol=5e-7)
        trajB = O.predict_core(s0, u_run, L=Lv, mode="f64sub")
        assert np.all(np.abs(trajB[:, -1] - g[f"{key}/{tag}/final_B"]) <= 0.2 * state_tol(g[f"{key}/{tag}/final_B"]))
    # the pole length matters: the default length gives other trajectories
    wrong = O.predict_core(s0, du.astype(f32)[:64], L=f32(0.395))
    assert np.abs(wrong[:, -1] - g[f"{key}/raw/final"][:64]).max() > 1e-2


@pytest.mark.parametrize("case,e", CASES)
def test_c_oracle_at_c3_c4_shape(g, case, e):
    """oracle/cpmppi_oracle.c - the checker of the full-size runs - on the same vectors: the predictor with a per-env length, and
    the whole step (rollouts, quadratic_boundary_grad_minimal, soft-min update) in both of the reference's arithmetic modes."""
    key, N, H, du, s0, target, Lv = inputs(g, case, e)
    base = dict(N=N, H=H, shift_mode="none", correction_u="u_nom", cc_weight=0.0)
    Ls = np.array([Lv], dtype=f32)
    for mode, fkey in (("f32", "final"), ("f64sub", "final_B")):
        c = OC.make_config(O.MPPIConfig(**base), mode=mode)
        traj = OC.predict(c, s0, du.astype(f32), L=np.full(N, Lv, f32))
        final = g[f"{key}/raw/{fkey}"]
        ok = (np.abs(traj[:, -1] - final) <= state_tol(final)).all(axis=1)
        # (float)cos((double)x) vs numpy's float32 kernels: an ulp now and then, amplified over 500 / 1000 substeps by the rollouts
        # that pass a bounce or the +-pi seam; the bulk is far inside the band
        assert ok.mean() >= 0.97, (mode, ok.mean())
        assert np.median(np.abs(traj[:, -1] - final)) < 2e-6
    for tag, cm in (("raw", "penalise"), ("clip", "clip")):
        cfg = O.MPPIConfig(cost_id=O.COST_QBGM, control_mode=cm, **base)
        u_new, Q, S = OC.step(OC.make_config(cfg), s0[None], np.zeros((1, H), f32), du[None], target, 1.0, L=Ls)
        ref = g[f"{key}/{tag}/S_qbgm"]
        rel = np.abs(S[0] - ref) / np.abs(ref)
        assert np.median(rel) < 2e-5 and (rel < 2e-3).mean() >= 0.97, (tag, np.median(rel), (rel < 2e-3).mean())
        np.testing.assert_allclose(u_new[0], g[f"{key}/{tag}/u_new"], atol=1e-4)
        assert abs(float(Q[0]) - float(g[f"{key}/{tag}/u_new"][0])) <= 1e-4
        cfg_d = O.MPPIConfig(cost_id=O.COST_DEFAULT, control_mode=cm, **base)
        _, _, S_d = OC.step(OC.make_config(cfg_d), s0[None], np.zeros((1, H), f32), du[None], target, 1.0, L=Ls)
        ref_d = g[f"{key}/{tag}/S_default"]
        rel_d = np.abs(S_d[0] - ref_d) / np.abs(ref_d)
        # (the default plugin's 1e7 / 1e4 indicator terms flip on rollouts that graze a threshold: those are the tail)
        assert np.median(rel_d) < 2e-5 and (rel_d < 2e-3).mean() >= 0.95, (tag, np.median(rel_d), (rel_d < 2e-3).mean())
