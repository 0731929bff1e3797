"""GPU: the BASELINE.json config shapes at full size.

  C2  1024 x 50,  E = 1            covered fixture-by-fixture in test_gpu_parity.py
  C3  4096 x 100, E = 64           one launch; parity of ALL 64 envs against the plain-C oracle (both of its arithmetic
                                   modes) in both math modes of the kernel + properties
  C4  2048 x 50,  E = 64 per GPU   (512 envs sharded 64/GPU): the per-GPU launch; the same
Size-independent properties checked on the FULL launch: batching invariance (an env's result does not depend on which
other envs share the launch), rollout-permutation invariance of the update, the soft-min update is a convex combination
of the perturbations, and determinism."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle_np as O  # noqa: E402
from oracle import oracle_c as OC  # noqa: E402
import parity_util as PU  # noqa: E402

f32 = np.float32


def make(E, N, H, **kw):
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    return MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, **kw))


def _record(name, tally):
    """Evidence for profiles/: the launch-wide tallies of test_config_full_size (written next to gpurun's other outputs)."""
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_records")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"config_full_size_{name}.json"), "w") as f:
            json.dump({"config": name, "rule": PU.ODE_V0.name, "modes": tally}, f, indent=1)
    except OSError:
        pass
    print(f"[{name}] " + json.dumps(tally))


def inputs(E, H, seed):
    rng = np.random.Generator(np.random.SFC64(seed))
    THL = 0.198
    ang = np.where(rng.uniform(size=E) > 0.5, 1.0, -1.0) * rng.uniform(0, 180, E) * np.pi / 180
    s0 = np.zeros((E, 6), f32)
    s0[:, 0], s0[:, 1] = ang, rng.uniform(-1, 1, E) * 1200 * np.pi / 180
    s0[:, 2], s0[:, 3] = np.cos(ang), np.sin(ang)
    s0[:, 4], s0[:, 5] = rng.uniform(-1, 1, E) * THL * 0.8, rng.uniform(-1, 1, E) * THL * 0.5
    return s0, (rng.uniform(-0.8, 0.8, E) * THL).astype(f32), np.ones(E, f32), rng.uniform(0.2, 0.5, E).astype(f32)


@pytest.mark.parametrize("name,E,N,H,seed", [("C3", 64, 4096, 100, 2), ("C4", 64, 2048, 50, 3),
                                              # the input seed (of four tried, tools/c3_spread.py) with the most adverse env: a start
                                              # on which the reference's own one-ulp probes scatter 13 x its two-mode gap
                                              ("C3", 64, 4096, 100, 21)])
def test_config_full_size(name, E, N, H, seed):
    eng = make(E, N, H)
    s0, tp, te, Lv = inputs(E, H, seed=seed)
    rng = np.random.Generator(np.random.SFC64(9))
    u0 = (0.1 * rng.standard_normal((E, H))).astype(f32)
    kn, _ = eng.sample(seed=2 if seed in (2, 3) else seed, offset=0)   # device RNG knots for the whole launch
    un = eng.tensor(u0.copy())
    S = eng.empty(E, N)
    Q, _ = eng.step(s0, un, tp, te, L=Lv, knots=kn, S_out=S)
    un_h, S_h, Q_h = un.cpu().numpy(), S.cpu().numpy(), Q.cpu().numpy()
    assert np.isfinite(un_h).all() and np.isfinite(S_h).all() and np.abs(un_h).max() <= 1.0
    assert np.array_equal(Q_h, un_h[:, 0])

    # ---- parity of EVERY env of the launch against the C oracle in both reference arithmetic modes (same knots,
    # interpolated by the oracle; H2 flags from the oracle's own trajectories), for both math modes of the kernel
    ocfg = O.MPPIConfig(N=N, H=H)
    kn_h = kn.cpu().numpy()
    prec = make(E, N, H, math_mode="precise")
    un_p, S_p = prec.tensor(u0.copy()), prec.empty(E, N)
    Q_p, _ = prec.step(s0, un_p, tp, te, L=Lv, knots=kn, S_out=S_p)
    outs = {"fast": (un_h, S_h, Q_h), "precise": (un_p.cpu().numpy(), S_p.cpu().numpy(), Q_p.cpu().numpy())}
    CH = 8                                                             # envs per oracle call (bounds the trajectory buffer)
    tally = {m: dict(clear=0, flagged=0, flagged_off=0, worst_clear_excess=0.0, worst_u_abs=0.0, worst_spread_ratio=0.0,
                     worst_spread_env=-1, worst_two_mode_ratio=0.0) for m in outs}
    for e0 in range(0, E, CH):
        sl = slice(e0, e0 + CH)
        du = np.stack([O.interpolate_knots(kn_h[e], H) for e in range(e0, e0 + CH)])
        # probes: besides modes A and B, five more realisations of the REFERENCE one rounding away from mode A (FMA float32
        # build, initial state / perturbations / pole length one ulp up).  A rollout on which those disagree among themselves
        # by more than a quarter of the band is rounding-sensitive (a pole swinging at 20 rad/s for 1000 substeps amplifies
        # 1e-7 to 1e-3) and joins the flagged bucket; every other rollout must sit inside band + that envelope (100 %).
        ref = PU.c_oracle_step_with_flags(ocfg, s0[sl], u0[sl], du, tp[sl], te[sl], L=Lv[sl], probes=True)
        for mode, (u_m, S_m, Q_m) in outs.items():
            T = tally[mode]
            for i, e in enumerate(range(e0, e0 + CH)):
                b = PU.assert_costs(S_m[e], ref["S_a"][i], ref["S_b"][i], ref["flags"][i], f"{name} {mode} env {e} costs",
                                    S_alt=[a[i] for a in ref["S_alt"]], flag_sensitive=True, rule=PU.ODE_V0)
                clear = ~b["flagged"]
                T["clear"] += int(clear.sum()); T["flagged"] += int(b["flagged"].sum())
                T["flagged_off"] += int((b["off"] & b["flagged"]).sum())
                T["worst_clear_excess"] = max(T["worst_clear_excess"], float(b["excess"][clear].max()) if clear.any() else 0.0)
                # 1e-4 + the spread of the oracle's own realisations of the reference on this env: modes A / B and - as for the costs -
                # the probes one rounding away from mode A.  (Round 5, input seed 21, env 8: the probes scatter 1.5e-3 around mode A,
                # 13 x the A/B gap; the PRECISE kernel, the reference's own operand order with the device's libm, sits 6.6e-4 away.)
                u_alt = [a[i] for a in ref["u_alt"]]
                PU.assert_controls(u_m[e], ref["u_a"][i], ref["u_b"][i], f"{name} {mode} env {e} u_nom", u_alt=u_alt,
                                   allowance=PU.softmin_allowance(ref["S_a"][i], ref["S_b"][i], du[i]))
                PU.assert_controls(Q_m[e], ref["u_a"][i][0], ref["u_b"][i][0], f"{name} {mode} env {e} Q", u_alt=[a[0] for a in u_alt],
                                   allowance=PU.softmin_allowance(ref["S_a"][i], ref["S_b"][i], du[i])[0])
                # how far the update sits from the reference's float32 result in units of the reference's OWN spread on this env
                # (never below the 1e-4 band) - the envelope of modes A / B and the one-rounding probes: the kernel must not
                # scatter more than three times what the reference's realisations do among themselves.  (The two-mode gap alone,
                # round 4's yardstick, is recorded: 2.48 / 0.38 on seed 2, 2.83 / 5.53 on seed 21 - see reference_spread_ratio.)
                ratio = PU.reference_spread_ratio(u_m[e], ref["u_a"][i], ref["u_b"][i], u_alt=[a[i] for a in ref["u_alt"]])
                T["worst_two_mode_ratio"] = max(T["worst_two_mode_ratio"], PU.reference_spread_ratio(u_m[e], ref["u_a"][i], ref["u_b"][i]))
                T["worst_u_abs"] = max(T["worst_u_abs"], float(np.abs(u_m[e] - ref["u_a"][i]).max()))
                if ratio > T["worst_spread_ratio"]:
                    T["worst_spread_ratio"], T["worst_spread_env"] = ratio, e
    _record(name if seed in (2, 3) else f"{name}_seed{seed}", tally)
    for mode, T in tally.items():
        total = T["clear"] + T["flagged"]
        assert total == E * N
        # the rule is not vacuous: the bulk of the launch is compared at band + envelope, and the flagged bucket - capped per
        # env at 2 % above - holds next to nothing outside over the whole launch
        # (ONE floor for every input seed, fixed by what it is to mean - the MAJORITY of the launch is compared at full strength - and
        # not by the numbers seen: measured clear fractions over five seeds in profiles/r6/c3_spread.txt, 66.9 % on the most
        # adverse one (seed 21: the oracle itself flags a third of that launch) to 76 %)
        assert T["clear"] >= 0.50 * total, f"{name} {mode}: only {T['clear']} of {total} rollouts are clear of every flag"
        assert T["flagged_off"] <= 0.005 * T["flagged"], f"{name} {mode}: {T['flagged_off']} of {T['flagged']} flagged rollouts outside"
        assert T["worst_spread_ratio"] <= 3.0, (f"{name} {mode}: env {T['worst_spread_env']}: |u - u_A| is {T['worst_spread_ratio']:.2f} x "
                                                f"max(1e-4, the spread of the oracle's realisations)")
    prec.close()
    del prec

    # ---- batching invariance + determinism: envs stepped alone / again give bit-identical results
    un2 = eng.tensor(u0.copy())
    eng.step(s0, un2, tp, te, L=Lv, knots=kn)
    assert np.array_equal(un2.cpu().numpy(), un_h)
    # stepped alone with the SAME lane mapping: bit-identical; with the other mapping (its intermediate substeps carry
    # the rotation differently): the same update to within the parity tolerance
    rpl_batch = 2 if E * N >= 131072 else 1                  # the library's lane-mapping rule (cpmppi.hip PACKED_MIN_ROLLOUTS)
    for rpl, exact in ((rpl_batch, True), (3 - rpl_batch, False)):
        small = make(1, N, H, rollouts_per_lane=rpl)
        for e in (1, E - 2):
            u1 = small.tensor(u0[e:e + 1].copy())
            small.step(s0[e:e + 1], u1, tp[e:e + 1], te[e:e + 1], L=Lv[e:e + 1], knots=kn[e:e + 1].contiguous())
            if exact:
                assert np.array_equal(u1.cpu().numpy()[0], un_h[e])
                continue
            # the OTHER lane mapping is another realisation of the same arithmetic (its intermediate substeps carry the rotation
            # differently): held to the same rule as the launch itself - band + the envelope of the oracle's realisations on this
            # env - for EVERY input seed (round 5 skipped this check on seed 21); where the env is benign, additionally to the
            # launch's own result within the band
            j = slice(e, e + 1)
            du1 = O.interpolate_knots(kn_h[e], H)[None]
            r1 = PU.c_oracle_step_with_flags(ocfg, s0[j], u0[j], du1, tp[j], te[j], L=Lv[j], probes=True)
            alt = [a[0] for a in r1["u_alt"]]
            PU.assert_controls(u1.cpu().numpy()[0], r1["u_a"][0], r1["u_b"][0], f"{name} env {e}, lane mapping {rpl}", u_alt=alt,
                               allowance=PU.softmin_allowance(r1["S_a"][0], r1["S_b"][0], du1[0]))
            if float(PU.envelope(r1["u_a"][0], r1["u_b"][0], *alt).max()) <= 1e-4:
                np.testing.assert_allclose(u1.cpu().numpy()[0], un_h[e], atol=2e-4)
        small.close()

    # ---- the same perturbations through the reference-layout buffer and through its re-tiled form: identical costs
    du_full = eng.interpolate(kn)
    S4, u4 = eng.empty(E, N), eng.tensor(u0.copy())
    eng.step(s0, u4, tp, te, L=Lv, delta_u=du_full, S_out=S4)
    S5, u5 = eng.empty(E, N), eng.tensor(u0.copy())
    eng.step(s0, u5, tp, te, L=Lv, delta_u_tiled=eng.tile_delta_u(du_full), S_out=S5)
    assert np.array_equal(S4.cpu().numpy(), S5.cpu().numpy())
    np.testing.assert_allclose(u4.cpu().numpy(), u5.cpu().numpy(), atol=5e-6)     # row-major vs quad-major summation order
    np.testing.assert_allclose(S4.cpu().numpy(), S_h, rtol=2e-5)                  # caller knots: float64 interpolation either way
    np.testing.assert_allclose(u4.cpu().numpy(), un_h, atol=2e-5)
    del du_full

    # ---- permutation invariance: shuffling an env's rollouts leaves its update unchanged (to summation order)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(5)).to(kn.device)
    u3 = eng.tensor(u0.copy())
    eng.step(s0, u3, tp, te, L=Lv, knots=kn[:, perm].contiguous())
    np.testing.assert_allclose(u3.cpu().numpy(), un_h, atol=2e-5)      # float32 weighted sums over 4096 rollouts

    # ---- the update is a convex combination of the perturbations (before clipping): min <= u_new - u_shift <= max
    du0 = O.interpolate_knots(kn_h[0], H)
    u_shift = np.concatenate([u0[0, 1:], u0[0, -1:]])
    inc = un_h[0] - u_shift
    free = np.abs(un_h[0]) < 1.0
    assert np.all(inc[free] <= du0.max(0)[free] + 1e-6) and np.all(inc[free] >= du0.min(0)[free] - 1e-6)


def test_headline_instantiation_vs_oracle():
    """The kernel the default bench line times - rollout_cost_kernel<quadratic_boundary_grad_minimal, FAST, in-kernel Philox,
    two rollouts per lane, throughput build>, selected above 1.5 M rollouts per launch - compared with the C oracle DIRECTLY
    (not through bit-identity with a smaller build): 1664 envs x 1024 x 50 in one launch, 16 envs spread over it re-computed by
    the oracle from their regenerated knots (cpmppi_sample with the launch's seed, step counter and global env index), two
    consecutive steps (cold and warm nominal sequence), rule ODE_V0 at full-width strength."""
    E, N, H = 1664, 1024, 50
    eng = make(E, N, H)
    s0, tp, te, Lv = inputs(E, H, seed=12)
    te[1::3] = -1.0            # (pole length, targets and the initial angle differ per env: the throughput build reads what it
                               # derives from them out of fold_env_kernel's per-env block)
    rng = np.random.Generator(np.random.SFC64(13))
    u_h = (0.1 * rng.standard_normal((E, H))).astype(f32)
    un, S = eng.tensor(u_h.copy()), eng.empty(E, N)
    envs = sorted({int(round(x)) for x in np.linspace(0, E - 1, 16)})
    seed, env_offset = 1234, 4096
    ocfg = O.MPPIConfig(N=N, H=H)
    for step in (7, 8):
        before = un.cpu().numpy()
        eng.step(s0, un, tp, te, L=Lv, seed=seed, offset=step, env_offset=env_offset, S_out=S)
        info = eng.last_launch()
        assert (info["cost_id"], info["math_mode"], info["noise_kind"], info["rollouts_per_lane"], info["build_variant"],
                info["ode_predictor"]) == (0, 1, 2, 2, 1, 0), info
        assert info["kernel"] == "rollout_cost_kernel<0, true, 2, 2, 1>"
        kn = np.concatenate([eng.sample(seed, offset=step, env_offset=env_offset + e, E=1)[0].cpu().numpy() for e in envs])
        rep = PU.verify_envs(ocfg, s0[envs], before[envs], kn, tp[envs], te[envs], Lv[envs], S.cpu().numpy()[envs],
                             un.cpu().numpy()[envs])
        print(f"[headline step {step}] {rep}")
        assert rep["ok"], rep
        assert rep["clear"] >= 0.70 * rep["rollouts"], rep
        assert rep["flagged_off"] <= max(1, 0.005 * rep["flagged"]), rep
        assert rep["worst_u_vs_reference_spread"] <= 3.0, rep
    eng.close()
