"""Differentiable float64 restatement (torch, CPU) of predictor_ODE_v0 + the plugin costs — the checker of the
gradient kernel.

TEST INFRASTRUCTURE ONLY (see oracle/oracle_np.py's header): imported by tests/ only.  The gradient-based optimizers of
the absent Control_Toolkit obtain d cost / d inputs from TensorFlow's automatic differentiation of the very code
oracle_np.py restates; here the same arithmetic is written with torch ops so that torch.autograd plays that role:
  physics  — CartPole/cartpole_equations.py:44-105,130-131,341-347,356-364; cartpole_numba.py:55-78;
             _CartPole_mathematical_helpers.py:24-29; predictors_customization_v0.py:41-55   (= oracle_np a3-a11)
  costs    — quadratic_boundary_grad_minimal.py:64-126, default.py:23-88, quadratic_boundary_grad.py:64-232
Pinned by tests/test_oracle_torch.py: its forward values equal oracle_np's (which the golden vectors pin), and its
autograd gradient equals central finite differences of itself.
"""
import numpy as np
import torch

from . import oracle_np as O

f64 = torch.float64


def _wrap(angle):
    two_pi, pi = 2 * np.pi, np.pi
    m = torch.fmod(angle, two_pi)
    return torch.where(m < -pi, m + two_pi, torch.where(m > pi, m - two_pi, m))


def _ode(ca, sa, angleD, positionD, u, L, p):
    k, m_cart, m_pole, g, J_fric, M_fric = (float(p.k), float(p.m_cart), float(p.m_pole), float(p.g), float(p.J_fric),
                                             float(p.M_fric))
    A = (k + 1) * (m_cart + m_pole) - m_pole * (ca * ca)
    F_fric = -M_fric * positionD
    T_fric = -J_fric * angleD
    L_half = L / 2.0
    positionDD = (m_pole * g * sa * ca + ((T_fric * ca) / L_half)
                  + (k + 1) * (-(m_pole * L_half * (angleD * angleD) * sa) + F_fric + u)) / A
    angleDD = (g * sa + positionDD * ca + T_fric / (m_pole * L_half)) / ((k + 1) * L_half)
    return angleDD, positionDD


def predict_core(s0, Q, dt=0.02, S=10, L=None, p=O.DEFAULT_PARAMS, integrator="ODE_v0"):
    """s0[6] or [N,6], Q[N,H] (torch float64, Q may require grad) -> list of H+1 states, each a 6-tuple of [N] tensors
    (angle, angleD, cos, sin, position, positionD).  integrator "ODE": predictor_ODE (oracle_np.fine_integration_cromer:
    Euler-Cromer, no edge bounce, angle = atan2(sin, cos) - cartpole_equations.py:229-249,293-308)."""
    N, H = Q.shape
    s0 = torch.as_tensor(np.asarray(s0, dtype=np.float64)).reshape(-1, 6).expand(N, 6)
    L = float(p.L) if L is None else float(L)
    THL = float(p.TrackHalfLength)
    t = float(dt / float(S))
    st = tuple(s0[:, i] for i in range(6))
    out = [st]
    for k in range(H):
        a, ad, ca, sa, x, xd = st
        u = float(p.u_max) * Q[:, k]
        for _ in range(S):
            aDD, xDD = _ode(ca, sa, ad, xd, u, L, p)
            if integrator == "ODE":
                ad, xd = ad + aDD * t, xd + xDD * t
                a, x = a + ad * t, x + xd * t
                ca, sa = torch.cos(a), torch.sin(a)
                a = torch.atan2(sa, ca)
                continue
            a, ad, x, xd = a + ad * t, ad + aDD * t, x + xd * t, xd + xDD * t
            cb = torch.cos(a)
            hit = (x >= THL) | (-x >= THL)
            ad_b = ad - 2 * (xd * cb) / (0.5 * L)
            a_b = a + ad_b * t
            xd_b = -xd
            x_b = x + xd_b * t
            a, ad = torch.where(hit, a_b, a), torch.where(hit, ad_b, ad)
            x, xd = torch.where(hit, x_b, x), torch.where(hit, xd_b, xd)
            a = _wrap(a)
            ca, sa = torch.cos(a), torch.sin(a)
        st = (a, ad, ca, sa, x, xd)
        out.append(st)
    return out


def _stack(traj, idx, lo, hi):
    return torch.stack([traj[k][idx] for k in range(lo, hi)], dim=1)


def trajectory_cost(cost_id, traj, inputs, target_position, target_equilibrium, horizon_reduce="sum", previous_input=0.0,
                    qbg_weights=None, p=O.DEFAULT_PARAMS, c=O.DEFAULT_COST):
    """The get_trajectory_cost of oracle_np.trajectory_cost on torch tensors: traj from predict_core, inputs[N,H]."""
    H = inputs.shape[1]
    THL, te, x_t = float(p.TrackHalfLength), float(target_equilibrium), float(target_position)
    ang, angD, x = _stack(traj, 0, 0, H), _stack(traj, 1, 0, H), _stack(traj, 4, 0, H)
    cosang = torch.cos(ang)
    term = torch.zeros(inputs.shape[0], dtype=f64)
    if cost_id == O.COST_QBGM:
        ptf = float(np.float32(c.qbgm_permissible_track_fraction))
        dd = c.qbgm_dd_quadratic_weight * ((x - x_t) / (2 * THL)) ** 2
        near = (x.abs() > ptf * THL).to(f64)
        db = c.qbgm_db_weight * (near * ((x.abs() - ptf * THL) / ((1 - ptf) * THL)) ** 2)
        ep = c.qbgm_ep_weight * (1.0 - te * cosang) ** 2
        ekp = c.qbgm_ekp_weight * angD ** 2
        cc = c.qbgm_cc_weight * (c.qbgm_R * inputs ** 2)
        stage = dd + db + ep + ekp + cc
    elif cost_id == O.COST_DEFAULT:
        dd = c.def_dd_weight * (((x - x_t) / (2.0 * THL)) ** 2 + (x.abs() > 0.90 * THL).to(f64) * 1.0e7)
        ep = c.def_ep_weight * (te * 0.25 * (1.0 - cosang) ** 2)
        cc = c.def_cc_weight * (c.def_R * inputs ** 2)
        stage = dd + ep + cc
        aT, xT = traj[H][0], traj[H][4]
        term = 10000.0 * ((aT.abs() > 0.2) | ((xT - x_t).abs() > 0.1 * THL)).to(f64)
    elif cost_id == 3:
        w = dict(O.QBG_DEFAULT_WEIGHTS, **(qbg_weights or {}))
        sfx = "_up" if te == 1.0 else "_down"
        g = lambda k: float(np.float32(w[k + sfx]))  # noqa: E731
        d = (x - x_t) / (2 * THL)
        ptf = float(np.float32(w["permissible_track_fraction"]))
        near = (x.abs() > ptf * THL).to(f64)
        db = g("db_weight") * (near * ((x.abs() - ptf * THL) / ((1 - ptf) * THL)) ** 2)
        ep = g("ep_weight") * (((2.0 - te * cosang) ** 2) - 1.0)
        tas_max = abs(120.0 * (1.0 + te) / 2.0 + g("target_angular_speed_sqr_max_correction"))
        basic = (1.0 - te * cosang) / 2
        cond = te * (cosang - float(np.cos(np.float32(w["admissible_angle"])))) > 0
        tas = tas_max * torch.where(cond, torch.zeros_like(basic), basic)
        ekp = g("ekp_weight") * (angD ** 2 - tas).abs()
        cc = g("cc_weight") * (float(np.float32(w["R"])) * inputs ** 2)
        u_before = torch.cat([torch.full((inputs.shape[0], 1), float(previous_input), dtype=f64), inputs[:, :-1]], dim=1)
        ccrc = g("ccrc_weight") * (inputs - u_before) ** 2
        stage = g("dd_linear_weight") * d.abs() + g("dd_quadratic_weight") * d ** 2 + db + ep + ekp + cc + ccrc
    else:
        raise ValueError(cost_id)
    if horizon_reduce == "sum":
        return stage.sum(dim=1) + term
    return torch.cat([stage, term[:, None]], dim=1).mean(dim=1)


def cost_and_grad(cost_id, s0, Q, target_position, target_equilibrium, L=None, dt=0.02, S=10, horizon_reduce="sum",
                  previous_input=0.0, qbg_weights=None, clip=(-1.0, 1.0), p=O.DEFAULT_PARAMS, c=O.DEFAULT_COST,
                  integrator="ODE_v0"):
    """numpy in / numpy out: (cost[N], grad[N,H]) of the trajectory cost w.r.t. the inputs Q[N,H], float64.
    ``clip``: the optimizer-style clip of the applied control (gradient zero where clipped); None = no clip."""
    Qt = torch.tensor(np.asarray(Q, dtype=np.float64), requires_grad=True)
    Qa = Qt.clamp(clip[0], clip[1]) if clip is not None else Qt
    traj = predict_core(s0, Qa, dt, S, L, p, integrator)
    J = trajectory_cost(cost_id, traj, Qa, target_position, target_equilibrium, horizon_reduce, previous_input, qbg_weights,
                        p, c)
    (g,) = torch.autograd.grad(J.sum(), Qt)
    return J.detach().numpy(), g.numpy()
