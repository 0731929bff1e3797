"""GPU: gradient of rollout + plugin cost w.r.t. the inputs (cpmppi_rollout_cost_grad) against torch.autograd of the
float64 oracle, the Adam step against numpy, and the gradient optimizers on top."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle_np as O  # noqa: E402
from oracle import oracle_torch as OT  # noqa: E402
import parity_util as PU  # noqa: E402

f32 = np.float32
QBG_W = dict(ccrc_weight_up=3.0, ccrc_weight_down=3.0, dd_linear_weight_up=2.0, dd_linear_weight_down=2.0)


def make(E, N, H, **kw):
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    return MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, shift_mode="none", **kw))


def envs(E, seed, edge=False):
    rng = np.random.Generator(np.random.SFC64(seed))
    if edge:
        s0 = np.stack([O.create_cartpole_state(0.4, 1.0, sgn * 0.185, sgn * 0.55) for sgn in np.resize([1.0, -1.0], E)])
    else:
        s0 = np.stack([O.create_cartpole_state(rng.uniform(-0.8, 0.8), rng.uniform(-2, 2), rng.uniform(-0.1, 0.1),
                                               rng.uniform(-0.3, 0.3)) for _ in range(E)])
    tp = rng.uniform(-0.05, 0.05, E).astype(f32)
    Lv = rng.uniform(0.3, 0.45, E).astype(f32)
    return s0, tp, Lv, rng


CASES = [("quadratic_boundary_grad_minimal", O.COST_QBGM, 1.0, "sum", False),
         ("quadratic_boundary_grad_minimal", O.COST_QBGM, -1.0, "mean", False),
         ("quadratic_boundary_grad_minimal", O.COST_QBGM, 1.0, "sum", True),
         ("default", O.COST_DEFAULT, 1.0, "sum", False),
         ("quadratic_boundary_grad", 3, 1.0, "sum", False),
         ("quadratic_boundary_grad", 3, -1.0, "sum", True)]


@pytest.mark.parametrize("predictor_type", ["ODE_v0", "ODE"])
@pytest.mark.parametrize("name,cost_id,te,reduce,edge", CASES)
def test_gradient_vs_autograd(name, cost_id, te, reduce, edge, predictor_type):
    """Both in-tree ODE predictors: predictor_ODE_v0 (bounce branch in the adjoint) and predictor_ODE (Euler-Cromer, no bounce -
    what the shipped config_controllers.yml:2-3 pairs with `optimizer: rpgd`)."""
    E, N, H = 3, 40, 35                                          # gradient-tf sizes (config_optimizers.yml:50,60)
    eng = make(E, N, H, cost_function_specification=name, horizon_reduce=reduce,
               cost_weights=QBG_W if cost_id == 3 else None, predictor_type=predictor_type)
    s0, tp, Lv, rng = envs(E, 21, edge)
    Q = (0.5 * rng.standard_normal((E, N, H))).astype(f32)
    Q[:, :4] *= 3.0                                              # some controls beyond the limits: clipped, zero gradient
    prev = np.asarray([0.2, -0.1, 0.0], dtype=f32)
    S, G = eng.rollout_cost_grad(s0, Q, tp, np.full(E, te, f32), L=Lv, previous_input=prev)
    S, G = S.cpu().numpy(), G.cpu().numpy()
    # the forward value is the cost-only launch's
    S2 = eng.rollout_cost(s0, Q, tp, np.full(E, te, f32), L=Lv).cpu().numpy() if cost_id != 3 else None
    if S2 is not None:
        np.testing.assert_allclose(S, S2, rtol=2e-4)
    bounced, n_flagged, n_flagged_off = 0, 0, 0
    for e in range(E):
        J, g = OT.cost_and_grad(cost_id, s0[e], Q[e], tp[e], te, L=Lv[e], horizon_reduce=reduce, previous_input=prev[e],
                                qbg_weights=QBG_W, integrator=predictor_type)
        traj = O.predict_core(s0[e], np.clip(Q[e], -1, 1), L=Lv[e], integrator=predictor_type)
        bounced += int((np.abs(traj[:, :, O.POSITION_IDX]).max(axis=1) >= 0.197).sum())
        np.testing.assert_allclose(S[e], J, rtol=5e-4)
        assert np.all(G[e][np.abs(Q[e]) > 1.0] == 0.0) and np.all(g[np.abs(Q[e]) > 1.0] == 0.0)
        # float32 adjoint through 350 substeps vs float64 autograd, relative to each rollout's gradient scale.  Buckets as
        # in parity_util, from the ORACLE's trajectory: a rollout that comes within float32 reach of a branch
        # (edge bounce, +-pi wrap, a cost indicator threshold, the control limit) may legitimately take the other branch
        # in one of the two evaluations - those are flagged and capped; every other rollout must be inside the bound.
        scale = np.abs(g).max(axis=1, keepdims=True) + 1e-6
        err = (np.abs(G[e] - g) / scale).max(axis=1)
        # (predictor_ODE has no bounce: its only state discontinuity is the +-pi seam of atan2, derivative 1 through it)
        flagged = (PU.flag_discontinuities(traj) if predictor_type == "ODE_v0" else np.zeros(N, bool)) | PU.flag_indicators(traj, {O.COST_QBGM: "qbgm", O.COST_DEFAULT: "default"}.get(cost_id, "qbg"), tp[e])
        flagged |= (np.abs(np.abs(Q[e]) - 1.0) < 1e-3).any(axis=1)
        clear_off = int(((err >= 5e-4) & ~flagged).sum())
        assert clear_off == 0, f"env {e}: {clear_off} of {int((~flagged).sum())} rollouts clear of every branch differ by more than 5e-4 (worst {err[~flagged].max():.2e})"
        assert np.median(err) < 1e-4
        n_flagged += int(flagged.sum()); n_flagged_off += int(((err >= 2e-3) & flagged).sum())
    assert n_flagged_off <= int(np.ceil(0.05 * n_flagged)), f"{n_flagged_off} of {n_flagged} flagged rollouts outside 2e-3"
    if edge:
        assert bounced > 0                                       # the bounce branch of the adjoint was exercised (ODE: rollouts beyond the edge)


def test_adam_step_vs_numpy():
    E, N, H = 2, 40, 35
    eng = make(E, N, H)
    rng = np.random.Generator(np.random.SFC64(3))
    Q = rng.uniform(-0.9, 0.9, (E, N, H)).astype(f32)
    m, v = np.zeros_like(Q), np.zeros_like(Q)
    Qd, md, vd = eng.tensor(Q.copy()), eng.tensor(m.copy()), eng.tensor(v.copy())
    lr, b1, b2, eps, clipn = 0.05, 0.9, 0.999, 1e-8, 5.0
    for it in range(1, 4):
        g = (rng.standard_normal((E, N, H)) * rng.choice([0.1, 1.0, 30.0], (E, N, 1))).astype(f32)
        eng.adam_step(Qd, eng.tensor(g), md, vd, it, lr, b1, b2, eps, clipn)
        nrm = np.sqrt((g.astype(np.float64) ** 2).sum(-1, keepdims=True))
        gc = g * np.minimum(1.0, clipn / np.maximum(nrm, 1e-30))
        m = b1 * m + (1 - b1) * gc
        v = b2 * v + (1 - b2) * gc * gc
        lr_t = lr * np.sqrt(1 - b2 ** it) / (1 - b1 ** it)
        Q = np.clip(Q - lr_t * m / (np.sqrt(v) + eps), -1, 1)
        np.testing.assert_allclose(Qd.cpu().numpy(), Q, atol=2e-6)
    np.testing.assert_allclose(md.cpu().numpy(), m, rtol=1e-5, atol=1e-7)
    with pytest.raises(RuntimeError):
        eng.adam_step(Qd, eng.tensor(g), md, vd, 0, lr)          # iterations count from 1


def test_gradient_descent_lowers_the_costs():
    E, N, H = 4, 32, 35
    eng = make(E, N, H)
    s0, tp, Lv, rng = envs(E, 8)
    te = np.ones(E, f32)
    Q = eng.tensor((0.3 * rng.standard_normal((E, N, H))).astype(f32))
    m, v = torch.zeros_like(Q), torch.zeros_like(Q)
    S0, _ = eng.rollout_cost_grad(s0, Q, tp, te, L=Lv)
    S0 = S0.clone()
    for it in range(1, 31):
        _, G = eng.rollout_cost_grad(s0, Q, tp, te, L=Lv)
        eng.adam_step(Q, G, m, v, it, 0.02, gradmax_clip=5.0)
    S1, _ = eng.rollout_cost_grad(s0, Q, tp, te, L=Lv)
    assert (S1 < S0).float().mean().item() > 0.8 and S1.mean().item() < 0.9 * S0.mean().item()


@pytest.mark.parametrize("spec", ["ODE_v0", "ODE"])
@pytest.mark.parametrize("name", ["gradient", "rpgd"])
def test_gradient_optimizers_through_the_controller_seam(name, spec):
    """controller_mpc.configure('gradient-tf' | 'rpgd'): shipped hyper-parameters (config_optimizers.yml:49-86), the plan
    improves the cost of the best candidate, the loop on the batched plant keeps mildly perturbed poles upright - on
    predictor_ODE_v0 and on the shipped pairing, `optimizer: rpgd` + `predictor_specification: "ODE"` (config_controllers.yml:2-3)."""
    from cartpolesimulation_amd.controller_mpc import controller_mpc
    E = 8
    ctrl = controller_mpc("CartPole", {"target_position": 0.0, "target_equilibrium": 1.0, "L": 0.395},
                          control_limits=([-1.0], [1.0]), num_envs=E, config=dict(seed=3))
    ctrl.configure(name, predictor_specification=spec)
    opt = ctrl.optimizer
    assert opt.cfg.predictor_type == spec
    assert opt.optimizer_name == name and opt.num_rollouts == (40 if name == "gradient" else 16) and opt.mpc_horizon == 35
    eng = opt.engine
    rng = np.random.Generator(np.random.SFC64(2))
    s = eng.tensor(np.stack([O.create_cartpole_state(rng.uniform(-0.25, 0.25), rng.uniform(-0.5, 0.5),
                                                     rng.uniform(-0.05, 0.05), 0.0) for _ in range(E)]))
    tp, te, Lv = np.zeros(E, f32), np.ones(E, f32), np.full(E, 0.395, f32)
    S_before = eng.rollout_cost(s, opt.Q, tp, te, L=Lv).min(dim=1).values.clone()
    Q0 = ctrl.step(s, 0.0, {})
    assert Q0.shape == (E, 1) and np.abs(Q0).max() <= 1.0
    # (plans were shifted after the step: compare the best cost reached on the un-shifted problem via the log)
    ctrl2 = controller_mpc("CartPole", {"target_position": 0.0, "target_equilibrium": 1.0, "L": 0.395},
                           control_limits=([-1.0], [1.0]), num_envs=E, config=dict(seed=3))
    ctrl2.configure(name, controller_logging=True, predictor_specification=spec)
    ctrl2.step(s, 0.0, {})
    S_after = torch.as_tensor(ctrl2.controller_data_for_csv["J_logged"]).min(dim=1).values
    assert (S_after < S_before.cpu()).all()
    # closed loop: 60 control steps on the device plant
    for k in range(60):
        Q = ctrl.optimizer.step(s, as_tensor=True)
        eng.plant_advance(s, Q, L=Lv, n_substeps=10)
    sh = s.cpu().numpy()
    assert (np.abs(sh[:, O.ANGLE_IDX]) < 0.35).mean() >= 0.75 and np.abs(sh[:, O.POSITION_IDX]).max() < 0.198


def test_gradient_with_more_substeps_than_the_default_lds_budget():
    """intermediate_steps = 20 needs 120 KB of LDS for the sub-states: launches (160 KB opt-in) and matches autograd."""
    E, N, H = 1, 8, 6
    eng = make(E, N, H, intermediate_steps=20)
    s0, tp, Lv, rng = envs(E, 5)
    Q = (0.4 * rng.standard_normal((E, N, H))).astype(f32)
    S, G = eng.rollout_cost_grad(s0, Q, tp, np.ones(E, f32), L=Lv)
    J, g = OT.cost_and_grad(O.COST_QBGM, s0[0], Q[0], tp[0], 1.0, L=Lv[0], S=20)
    np.testing.assert_allclose(S.cpu().numpy()[0], J, rtol=5e-4)
    scale = np.abs(g).max(axis=1, keepdims=True) + 1e-6
    assert (np.abs(G.cpu().numpy()[0] - g) / scale).max() < 2e-3
