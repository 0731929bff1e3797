"""CPU: the host-side bookkeeping of the sampling / gradient optimizers (shift, elite refit, survivor gather,
re-sampling, logging) exercised against a CHECKER-backed stand-in for the device engine.

The product classes build an MPPIEngine in configure(); here that class is replaced by FakeEngine, whose rollout, cost
and gradient come from the numpy / torch oracles (test infrastructure) on CPU tensors.  Nothing in the product imports
this; the real engine is covered by the -m gpu tests."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import oracle_np as O  # noqa: E402
from oracle import oracle_torch as OT  # noqa: E402


class FakeEngine:
    def __init__(self, E, cfg, phys=None, device=0):
        self.E, self.N, self.H, self.cfg = int(E), int(cfg.num_rollouts), int(cfg.mpc_horizon), cfg
        self.device = torch.device("cpu")
        self.lo, self.hi = float(cfg.action_low), float(cfg.action_high)
        self.calls = {"grad": 0, "cost": 0, "adam": 0, "sgd": 0, "cem_sample": 0, "sample": 0}
        self._g = torch.Generator().manual_seed(0)

    # -- plumbing
    def apply_pole_mass_of(self, variable_parameters):
        pass

    def set_gru(self, model):
        self.gru = model

    def tensor(self, x, shape=None):
        t = x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x, dtype=np.float32))
        return t.to(torch.float32).reshape(shape).contiguous() if shape is not None else t.to(torch.float32).contiguous()

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32)

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32)

    # -- samplers (any deterministic normal stream will do for the bookkeeping)
    def sample(self, seed, offset=0, env_offset=0, E=None, knots=True, delta_u=False):
        self.calls["sample"] += 1
        g = torch.Generator().manual_seed(int(seed) * 1000003 + int(offset))
        return None, self.cfg.sigma * torch.randn(self.E, self.N, self.H, generator=g)

    def cem_sample(self, mean, stdev, seed, offset=0, env_offset=0):
        self.calls["cem_sample"] += 1
        g = torch.Generator().manual_seed(int(seed) * 1000003 + int(offset))
        z = torch.randn(self.E, self.N, self.H, generator=g)
        return (mean[:, None, :] + stdev[:, None, :] * z).clamp(self.lo, self.hi).contiguous()

    # -- oracle-backed evaluation
    def _each(self, s0, Q, tp, te, L, fn):
        s0, Q = np.asarray(s0, dtype=np.float32), Q.detach().numpy()
        return [fn(s0[e], Q[e], float(np.asarray(tp).reshape(-1)[e]), float(np.asarray(te).reshape(-1)[e]),
                   None if L is None else float(np.asarray(L).reshape(-1)[e])) for e in range(Q.shape[0])]

    def rollout_cost(self, s0, inputs, tp, te, L=None):
        self.calls["cost"] += 1
        out = self._each(s0, inputs, tp, te, L, lambda s, q, a, b, l: OT.cost_and_grad(O.COST_QBGM, s, q, a, b, L=l)[0])
        return torch.as_tensor(np.stack(out), dtype=torch.float32)

    def rollout_cost_grad(self, s0, inputs, tp, te, L=None, previous_input=None):
        self.calls["grad"] += 1
        out = self._each(s0, inputs, tp, te, L, lambda s, q, a, b, l: OT.cost_and_grad(O.COST_QBGM, s, q, a, b, L=l))
        return (torch.as_tensor(np.stack([o[0] for o in out]), dtype=torch.float32),
                torch.as_tensor(np.stack([o[1] for o in out]), dtype=torch.float32))

    def adam_step(self, Q, grad, m, v, it, lr, b1=0.9, b2=0.999, eps=1e-8, clip=0.0):
        self.calls["adam"] += 1
        n = grad.norm(dim=2, keepdim=True)
        g = grad * torch.clamp(clip / n.clamp_min(1e-30), max=1.0) if clip > 0 else grad
        m.mul_(b1).add_((1 - b1) * g)
        v.mul_(b2).add_((1 - b2) * g * g)
        lr_t = lr * np.sqrt(1 - b2 ** it) / (1 - b1 ** it)
        Q.sub_(lr_t * m / (v.sqrt() + eps)).clamp_(self.lo, self.hi)
        return Q

    def sgd_step(self, Q, grad, lr, clip=0.0):
        self.calls["sgd"] += 1
        n = grad.norm(dim=2, keepdim=True)
        g = grad * torch.clamp(clip / n.clamp_min(1e-30), max=1.0) if clip > 0 else grad
        Q.sub_(lr * g).clamp_(self.lo, self.hi)
        return Q

    def cem_update(self, S, Q, best_k, stdev_min, return_elites=False):
        idx = torch.argsort(S, dim=1, stable=True)[:, :best_k]
        el = torch.gather(Q, 1, idx[:, :, None].expand(-1, -1, Q.shape[2]))
        mean, std = el.mean(dim=1), el.std(dim=1, unbiased=False).clamp_min(stdev_min)
        return (mean, std, idx.to(torch.int32)) if return_elites else (mean, std)


@pytest.fixture()
def fake_engine(monkeypatch):
    import cartpolesimulation_amd.engine as EN
    monkeypatch.setattr(EN, "MPPIEngine", FakeEngine)
    return FakeEngine


def _states(E):
    rng = np.random.Generator(np.random.SFC64(3))
    return np.stack([O.create_cartpole_state(rng.uniform(-0.3, 0.3), rng.uniform(-0.5, 0.5), rng.uniform(-0.05, 0.05), 0.0)
                     for _ in range(E)])


def test_gradient_and_rpgd_bookkeeping(fake_engine):
    from cartpolesimulation_amd.optimizer_gradient import optimizer_gradient, optimizer_rpgd
    E, H = 2, 8
    s = _states(E)
    # gradient: plans improve, best first control is applied, everything shifts by one
    g = optimizer_gradient(seed=1, mpc_horizon=H, num_rollouts=6, gradient_steps=3, num_envs=E, optimizer_logging=True)
    g.configure()
    Q0 = g.Q.clone()
    J0 = g.engine.rollout_cost(s, Q0, np.zeros(E), np.ones(E)).min(dim=1).values
    u = g.step(s)
    assert u.shape == (E, 1) and g.engine.calls["grad"] == 3 and g.engine.calls["adam"] == 3 and g.adam_it == 3
    J1 = torch.as_tensor(g.logging_values["J_logged"]).min(dim=1).values
    assert (J1 < J0).all()
    best = np.argmin(g.logging_values["J_logged"], axis=1)
    np.testing.assert_allclose(u[:, 0], g.logging_values["u_logged"][:, 0], atol=0)
    # after the step every plan was shifted left by one, the last input repeated, the moments' tail zeroed
    assert torch.equal(g.Q[:, :, -1], g.Q[:, :, -2]) and not g.m[:, :, -1].any() and not g.v[:, :, -1].any()
    np.testing.assert_allclose(g.Q[np.arange(E), best, :-1].numpy(), g.logging_values["u_logged"][:, 1:], atol=0)
    assert g._previous_input is not None and np.allclose(g._previous_input.numpy(), u[:, 0])

    # rpgd: every resamp_per steps the worst plans are re-drawn, survivors keep their order by cost and their moments
    r = optimizer_rpgd(seed=2, mpc_horizon=H, num_rollouts=8, outer_its=2, resamp_per=2, opt_keep_k_ratio=0.5,
                       shift_previous=1, num_envs=E, period_interpolation_inducing_points=4, sample_stdev=0.3)
    r.configure()
    assert r.opt_keep_k == 4 and r.engine.calls["sample"] == 1
    r.step(s)                                   # count = 1: no resampling
    assert r.engine.calls["sample"] == 1
    m_before = r.m.clone()
    r.step(s)                                   # count = 2: resample
    assert r.engine.calls["sample"] == 2
    assert not r.m[:, 4:].any() and not r.v[:, 4:].any()          # fresh plans start with zero moments
    assert r.m[:, :4].abs().sum() > 0 and m_before.abs().sum() > 0
    assert float(r.Q.abs().max()) <= 1.0
    with pytest.raises(ValueError):
        optimizer_rpgd(SAMPLING_DISTRIBUTION="cauchy")
    # uniform sampling maps the normal draw through its CDF into the requested interval
    ru = optimizer_rpgd(seed=2, mpc_horizon=H, num_rollouts=8, num_envs=E, SAMPLING_DISTRIBUTION="uniform",
                        uniform_dist_min=-0.4, uniform_dist_max=0.2)
    ru.configure()
    assert float(ru.Q.min()) >= -0.4 and float(ru.Q.max()) <= 0.2 and float(ru.Q.std()) > 0.1


def test_cem_family_bookkeeping(fake_engine):
    from cartpolesimulation_amd.optimizer_cem import (optimizer_cem, optimizer_cem_grad_bharadhwaj, optimizer_cem_naive_grad,
                                                     optimizer_random_action)
    E, H = 2, 8
    s = _states(E)
    c = optimizer_cem(seed=4, mpc_horizon=H, num_rollouts=12, cem_best_k=4, cem_outer_it=2, num_envs=E, optimizer_logging=True)
    c.configure()
    u = c.step(s)
    assert u.shape == (E, 1) and c.engine.calls["cem_sample"] == 2 and c.step_counter == 2
    # the applied control is the refitted mean's first element; mean and stdev were shifted (mid-point / sqrt(0.5) appended)
    np.testing.assert_allclose(u[:, 0], c.logging_values["u_logged"][:, 0], atol=0)
    np.testing.assert_allclose(c.dist_mue[:, :-1].numpy(), c.logging_values["u_logged"][:, 1:], atol=0)
    assert np.allclose(c.dist_mue[:, -1].numpy(), 0.0) and np.allclose(c.stdev[:, -1].numpy(), np.sqrt(0.5))
    n = optimizer_cem_naive_grad(seed=4, mpc_horizon=H, num_rollouts=12, cem_best_k=4, num_envs=E)
    n.configure()
    n.step(s)
    assert n.engine.calls["grad"] == 1 and n.engine.calls["sgd"] == 1 and n.cem_outer_it == 1
    b = optimizer_cem_grad_bharadhwaj(seed=4, mpc_horizon=H, num_rollouts=8, cem_best_k=2, num_envs=E)
    b.configure()
    b.step(s)
    assert b.engine.calls["grad"] == 2 and b.engine.calls["adam"] == 2 and b._it == 2
    b.step(s)
    assert b._it == 2                            # moments and iteration count restart every control step
    ra = optimizer_random_action(seed=4, mpc_horizon=H, num_rollouts=16, num_envs=E, optimizer_logging=True)
    ra.configure()
    u = ra.step(s)
    J = ra.logging_values["J_logged"]
    assert u.shape == (E, 1) and J.shape == (E, 16) and np.abs(u).max() <= 1.0
    with pytest.raises(ValueError):
        ra.step(_states(E + 1))


def _fake_step(self, s0, u_nom, tp, te, L=None, delta_u=None, knots=None, seed=None, offset=0, env_offset=0, u_prev=None,
               Q_out=None, S_out=None, predictor="ODE_v0", h0=None, previous_input=None, offset_dev=None):
    """MPPIEngine.step on the numpy oracle (default flags of MPPIConfig), in-place on u_nom like the real one."""
    E = u_nom.shape[0]
    s0 = self.tensor(s0).reshape(E, 6).numpy()
    tp = np.broadcast_to(np.asarray(tp, dtype=np.float32).reshape(-1), (E,)) if np.size(tp) in (1, E) else tp
    te = np.broadcast_to(np.asarray(te, dtype=np.float32).reshape(-1), (E,))
    cfg = O.MPPIConfig(N=self.N, H=self.H)
    if Q_out is None:
        Q_out = self.empty(E)
    for e in range(E):
        rng = np.random.Generator(np.random.SFC64([int(seed), int(offset), int(env_offset) + e]))
        du = O.sample_delta_u(rng, self.N, self.H, np.float64(cfg.stdev))
        out = O.mppi_step(s0[e], u_nom[e].numpy(), du, float(tp[e]), float(te[e]), cfg,
                          L=None if L is None else float(np.asarray(L).reshape(-1)[e]))
        u_nom[e] = torch.as_tensor(out["u_new"])
        Q_out[e] = float(out["Q"])
        if S_out is not None:
            S_out[e] = torch.as_tensor(out["S"])
    self.calls["step"] = self.calls.get("step", 0) + 1
    self.last_kwargs = dict(previous_input=previous_input, L=L, tp=np.array(tp), te=np.array(te))
    return Q_out, S_out


def test_controller_mpc_and_optimizer_mppi_host_logic(fake_engine, monkeypatch):
    """The controller seam end to end on CPU: attribute updates reach the optimizer, the nominal sequence is carried and
    shifted by the step, logging mirrors the reference's keys, reset clears the plan."""
    monkeypatch.setattr(FakeEngine, "step", _fake_step, raising=False)
    from cartpolesimulation_amd.controller_mpc import controller_mpc
    ctrl = controller_mpc("CartPole", {"target_position": 0.0, "target_equilibrium": 1.0, "L": 0.395},
                          control_limits=(np.array([-1.0], np.float32), np.array([1.0], np.float32)),
                          config=dict(seed=9, num_rollouts=64, mpc_horizon=10))
    ctrl.configure("mppi", controller_logging=True)
    opt = ctrl.optimizer
    assert ctrl.has_optimizer and opt.optimizer_name == "mppi" and opt.num_rollouts == 64 and opt.mpc_horizon == 10
    s = O.create_cartpole_state(0.2, 0.0, 0.0, 0.0)
    q1 = ctrl.step(s, 0.0, {"target_position": 0.05, "L": 0.3})
    assert np.asarray(q1).shape == (1,) and abs(float(q1[0])) <= 1.0
    assert np.allclose(opt.engine.last_kwargs["tp"], 0.05) and np.allclose(np.asarray(opt.engine.last_kwargs["L"]), 0.3)
    assert set(ctrl.controller_data_for_csv) >= {"Q_logged", "J_logged", "u_logged"}
    assert ctrl.controller_data_for_csv["J_logged"].shape == (1, 64)
    u_after_1 = opt.u_nom.clone()
    q2 = ctrl.step(s, 0.02, {})
    assert opt.engine.calls["step"] == 2 and opt.step_counter == 2
    assert float(q2[0]) == float(opt.u_nom[0, 0]) and not torch.equal(opt.u_nom, u_after_1)
    assert np.allclose(opt.engine.last_kwargs["tp"], 0.05)            # attributes persist until updated again
    ctrl.controller_reset()
    assert float(opt.u_nom.abs().max()) == 0.0 and opt.step_counter == 0
    with pytest.raises(ValueError):
        opt.step(np.zeros((3, 6), np.float32))                        # configured for one env


def test_ode_is_not_served_as_ode_v0(fake_engine):
    """`ODE` / `ODE_default` name next_state_predictor_ODE (predictors_customization.py:25-69: Euler-Cromer, atan2, no
    bounce), 1.6e-3 from ODE_v0 after ONE control step: every seam selects that integrator's kernels
    (cpmppi_config.ode_predictor), never the ODE_v0 ones - the adjoint-based optimizers too (the shipped configuration is
    `optimizer: rpgd` on `predictor_specification: "ODE"`, config_controllers.yml:2-3)."""
    from cartpolesimulation_amd import _lib as L
    from cartpolesimulation_amd.configs import build_c_config
    from cartpolesimulation_amd.optimizer_cem import optimizer_cem, optimizer_cem_naive_grad
    from cartpolesimulation_amd.optimizer_gradient import optimizer_gradient, optimizer_rpgd
    from cartpolesimulation_amd.optimizer_mppi import optimizer_mppi
    from cartpolesimulation_amd.predictors import PredictorWrapper
    for spec, kind in (("ODE", L.ODE_CROMER), ("ODE_default", L.ODE_CROMER), ("ODE_v0", L.ODE_V0), (None, L.ODE_V0)):
        for cls in (optimizer_mppi, optimizer_cem, optimizer_cem_naive_grad, optimizer_gradient, optimizer_rpgd):
            opt = cls(num_rollouts=8, mpc_horizon=4, seed=7)
            opt.configure(predictor_specification=spec)
            assert build_c_config(1, opt.cfg).ode_predictor == kind
        w = PredictorWrapper()
        w.update_predictor_config_from_specification(spec)
        assert w.predictor_type == ("ODE" if kind == L.ODE_CROMER else "ODE_v0")
    with pytest.raises(NotImplementedError):
        PredictorWrapper().update_predictor_config_from_specification("SGP_10")
    for cls in (optimizer_cem, optimizer_gradient):
        with pytest.raises(NotImplementedError):
            cls(num_rollouts=8, mpc_horizon=4, seed=7).configure(predictor_specification="SGP_10")
    # a model that the chosen specification would silently ignore is an error too
    with pytest.raises(ValueError):
        optimizer_mppi(num_rollouts=8, mpc_horizon=4, gru_model={"w_ih0": None}).configure(predictor_specification="ODE_v0")


@pytest.mark.skipif(not os.path.isdir("/root/reference/Control_Toolkit_ASF"), reason="reference checkout not mounted")
def test_a_checkout_is_configured_as_shipped(fake_engine):
    """controller_mpc(config_root=<checkout>).configure() with no arguments = what the reference's controller_mpc does with its
    own YAML files: `mpc: optimizer: rpgd`, `predictor_specification: "ODE"`, cost quadratic_boundary_grad_minimal
    (config_controllers.yml:1-4), the rpgd section's hyper-parameters (config_optimizers.yml:63-86) - not the mppi section's."""
    from cartpolesimulation_amd.controller_mpc import controller_mpc
    from cartpolesimulation_amd.optimizer_gradient import optimizer_rpgd
    from cartpolesimulation_amd.optimizer_mppi import optimizer_mppi
    c = controller_mpc("CartPole", {"target_position": 0.0}, config_root="/root/reference", config=dict(seed=5))
    c.configure()
    opt = c.optimizer
    assert isinstance(opt, optimizer_rpgd) and opt.num_rollouts == 16 and opt.mpc_horizon == 35
    assert opt.cfg.predictor_type == "ODE" and opt.cfg.cost_function_specification == "quadratic_boundary_grad_minimal"
    assert opt.cfg.cost_weights["db_weight_up"] == 10000
    # the caller's overrides still win, and naming the optimizer still selects it (with ITS section)
    c2 = controller_mpc("CartPole", {"target_position": 0.0}, config_root="/root/reference", config=dict(seed=5, num_rollouts=64))
    c2.configure("mppi")
    assert isinstance(c2.optimizer, optimizer_mppi) and c2.optimizer.num_rollouts == 64 and c2.optimizer.cfg.predictor_type == "ODE"
    c3 = controller_mpc("CartPole", {"target_position": 0.0}, config_root="/root/reference", config=dict(seed=5))
    c3.configure("cem-tf")
    assert c3.optimizer.num_rollouts == 200 and c3.optimizer.cfg.predictor_type == "ODE"


@pytest.mark.skipif(not os.path.isdir("/root/reference/Control_Toolkit_ASF"), reason="reference checkout not mounted")
def test_a_checkout_with_an_explicit_gru_model_runs_the_gru(fake_engine, monkeypatch):
    """config_root reads the checkout's `predictor_specification: "ODE"` as the DEFAULT predictor only: a gru_model handed over
    explicitly wins (it did before config_root learnt to read that line), and naming an ODE specification next to a model is
    still an error."""
    import cartpolesimulation_amd.optimizer_mppi as OM
    from cartpolesimulation_amd.controller_mpc import controller_mpc
    monkeypatch.setattr(OM, "MPPIEngine", fake_engine, raising=False)
    model = {"w_ih0": np.zeros((96, 6), np.float32)}
    c = controller_mpc("CartPole", {"target_position": 0.0}, config_root="/root/reference", config=dict(seed=5, gru_model=model))
    c.configure("mppi")
    assert c.predictor is None and c.optimizer.gru_model is model and c.optimizer.h is not None
    assert c.optimizer.engine.gru is model
    with pytest.raises(ValueError):
        c.configure("mppi", predictor_specification="ODE")


def test_a_named_predictor_entry_keeps_its_own_substeps():
    """config_controllers.yml:3 may name an ENTRY of config_predictors.yml (e.g. `I_love_control_too`): its own
    intermediate_steps apply, "<type>_default" only when the entry has none."""
    from cartpolesimulation_amd.configs import mppi_config_from_yaml
    opt = dict(seed=1, mpc_horizon=10, mpc_timestep=0.02, num_rollouts=32, cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0,
               SQRTRHOINV=0.03, period_interpolation_inducing_points=10)
    base = dict(optimizers={"mppi": opt}, cost={"cost_function_name_default": "default", "CartPole": {"default": {}}})
    preds = {"ODE_v0_default": {"predictor_type": "ODE_v0", "intermediate_steps": 10},
             "ODE_default": {"predictor_type": "ODE", "intermediate_steps": 10},
             "custom": {"predictor_type": "ODE", "intermediate_steps": 2}, "bare": {"predictor_type": "ODE"}}
    for spec, (ptype, steps) in {"custom": ("ODE", 2), "bare": ("ODE", 10), "ODE": ("ODE", 10), "ODE_v0": ("ODE_v0", 10)}.items():
        cfgs = dict(base, controllers={"mpc": {"predictor_specification": spec}}, predictors={"predictors": preds})
        cfg = mppi_config_from_yaml(cfgs)
        assert (cfg.predictor_type, cfg.intermediate_steps) == (ptype, steps), spec
