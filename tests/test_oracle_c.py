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


def test_verify_envs_says_yes_no_and_asks_for_more_realisations(monkeypatch):
    """oracle/parity.py verify_envs, the rule behind bench.py's `verified` objects, on the CPU: a chaotic env (18 rad/s, 100 control
    steps: the reference's own realisations disagree on the update by more than the 1e-4 band) and a calm one.
      * another float32 sin / cos implementation of the reference (what a GPU is against the host's libm) is accepted;
      * a different pole mass is not;
      * on the chaotic env (the oracle's first seven realisations already more than the band apart) the checker ALWAYS draws six more
        - whether the update handed in would pass without them or not: the rule does not look at the device's result before it
        fixes its envelope (round 5 drew them after a failure only) -, counts them, and still says no to an update they do not reach;
      * a bucket nothing fell into reports None, not 0."""
    from oracle import parity as PR
    N, H = 1024, 100
    ocfg = O.MPPIConfig(N=N, H=H)
    rng = np.random.Generator(np.random.SFC64(5))
    states = [(-0.2, 0.4, 0.01, 0.02)]
    for trial in range(2):                                         # (the second draw of this stream: measured envelope 6e-4)
        chaotic = (rng.uniform(-3, 3), rng.uniform(15, 21) * rng.choice([-1, 1]), rng.uniform(-0.15, 0.15), rng.uniform(-0.3, 0.3))
        u0c = (0.1 * rng.standard_normal((1, H))).astype(np.float32)
        knc = (0.2121 * rng.standard_normal((1, N, 11))).astype(np.float32)
    states.append(chaotic)
    s0 = np.array([[a, ad, np.cos(a), np.sin(a), x, xd] for a, ad, x, xd in states], np.float32)
    u0 = np.concatenate([np.zeros((1, H), np.float32), u0c])
    kn = np.concatenate([(0.2121 * rng.standard_normal((1, N, 11))).astype(np.float32), knc])
    tp, te, L = np.full(2, 0.05, np.float32), np.ones(2, np.float32), np.full(2, 0.395, np.float32)
    du = np.stack([O.interpolate_knots(kn[e], H) for e in range(2)])
    (u_j, S_j), = PR.trig_jitter_realisations(ocfg, s0, u0, du, tp, te, L=L, seeds=(21,))
    rep = PR.verify_envs(ocfg, s0, u0, kn, tp, te, L, S_j, u_j)
    assert rep["ok"] and rep["clear"] > N // 2 and rep["clear_off"] == 0 and rep["worst_cost_rel"] is not None
    # the second look is taken for the chaotic env although the update PASSES (non-adaptive), never for the calm one
    assert rep["second_stage_envs"] == 1 and [x["env"] for x in rep["second_stage"]] == [1] and rep["second_stage"][0]["realisations"] == 6
    ref = PR.c_oracle_step_with_flags(ocfg, s0, u0, du, tp, te, L=L, probes=True)
    gap = float(PR.envelope(ref["u_a"][1], ref["u_b"][1], *[a[1] for a in ref["u_alt"]]).max())
    assert gap > 1e-4                                              # the chaotic env is one
    import dataclasses
    heavy = dataclasses.replace(O.DEFAULT_PARAMS, m_pole=np.float32(0.09))
    assert not PR.verify_envs(ocfg, s0, u0, kn, tp, te, L, S_j, u_j, params=heavy)["ok"]
    # outside the envelope of the first seven on the chaotic env
    u_far = u_j.copy()
    u_far[1, 3] = np.float32(ref["u_a"][1][3] + 2.5 * (gap + 1e-4))
    calls = []
    real = PR.trig_jitter_realisations

    def counted(*a, **k):
        calls.append(1)
        return real(*a, **k)

    monkeypatch.setattr(PR, "trig_jitter_realisations", counted)
    rep = PR.verify_envs(ocfg, s0, u0, kn, tp, te, L, S_j, u_far)
    assert calls == [1] and rep["second_stage_envs"] == 1 and rep["second_stage"][0]["realisations"] == 6
    assert not rep["ok"] and rep["u_off_envs"] == 1 and rep["second_stage"][0]["envelope_after"] < 2.5 * gap
    # ... and accepted when the further realisations do scatter that far (here: one that is handed in)
    monkeypatch.setattr(PR, "trig_jitter_realisations", lambda *a, **k: [(u_far[1:2], S_j[1:2])])
    assert PR.verify_envs(ocfg, s0, u0, kn, tp, te, L, S_j, u_far)["ok"]
    # on the CALM env no second look is taken: outside is outside (the one call is the chaotic env's, as always)
    monkeypatch.setattr(PR, "trig_jitter_realisations", counted)
    u_bad = u_j.copy()
    u_bad[0, 0] += np.float32(5e-4)
    del calls[:]
    rep = PR.verify_envs(ocfg, s0, u0, kn, tp, te, L, S_j, u_bad)
    assert not rep["ok"] and calls == [1] and rep["second_stage_envs"] == 1 and rep["u_off_envs"] == 1
    # a report without any such env says so: the key is always there
    calm = PR.verify_envs(ocfg, s0[:1], u0[:1], kn[:1], tp[:1], te[:1], L[:1], S_j[:1], u_j[:1])
    assert calm["second_stage_envs"] == 0 and calm["second_stage"] == []
    # an empty bucket: nothing compared is not a perfect match
    one = PR.verify_envs(ocfg, s0[:1], u0[:1], kn[:1], tp[:1], te[:1], L[:1], S_j[:1], u_j[:1])
    if one["flagged"] == 0:
        assert one["worst_flagged_excess"] is None and one["worst_flagged_cost_rel"] is None
