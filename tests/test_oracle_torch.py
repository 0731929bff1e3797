"""The differentiable float64 oracle (oracle/oracle_torch.py) pinned to the numpy oracle and to finite differences."""
import numpy as np
import pytest

from oracle import oracle_np as O
from oracle import oracle_torch as OT

f32 = np.float32


def _case(seed, N=6, H=12, regime="mild"):
    rng = np.random.Generator(np.random.SFC64(seed))
    if regime == "mild":
        s0 = O.create_cartpole_state(rng.uniform(-0.6, 0.6), rng.uniform(-1.5, 1.5), rng.uniform(-0.1, 0.1), rng.uniform(-0.3, 0.3))
    else:       # heading into the track edge: several rollouts bounce
        s0 = O.create_cartpole_state(0.4, 1.0, 0.185, 0.55)
    Q = np.clip(0.6 * rng.standard_normal((N, H)), -1.3, 1.3).astype(f32)
    return s0, Q


@pytest.mark.parametrize("integrator", ["ODE_v0", "ODE"])
@pytest.mark.parametrize("cost_id,te", [(O.COST_QBGM, 1.0), (O.COST_QBGM, -1.0), (O.COST_DEFAULT, 1.0), (3, 1.0), (3, -1.0)])
def test_forward_equals_numpy_oracle(cost_id, te, integrator):
    import torch
    s0, Q = _case(3)
    Qc = np.clip(Q, -1, 1)
    traj_np = O.predict_core(s0, Qc, mode="f64sub", integrator=integrator)
    J_np = O.trajectory_cost(cost_id, traj_np, Qc, f32(0.05), f32(te))
    traj_t = OT.predict_core(s0, torch.tensor(Qc, dtype=torch.float64), integrator=integrator)
    last = np.stack([c.numpy() for c in traj_t[-1]], axis=1)
    np.testing.assert_allclose(last, traj_np[:, -1], rtol=1e-4, atol=1e-4)
    J_t, _ = OT.cost_and_grad(cost_id, s0, Q, 0.05, te, integrator=integrator)
    np.testing.assert_allclose(J_t, J_np, rtol=2e-4)


@pytest.mark.parametrize("integrator", ["ODE_v0", "ODE"])
@pytest.mark.parametrize("regime", ["mild", "edge"])
@pytest.mark.parametrize("cost_id,te,reduce", [(O.COST_QBGM, 1.0, "sum"), (O.COST_DEFAULT, 1.0, "mean"), (3, -1.0, "sum")])
def test_autograd_equals_finite_differences(cost_id, te, reduce, regime, integrator):
    s0, Q = _case(11, N=4, H=8, regime=regime)
    Q = Q.astype(np.float64)
    kw = dict(horizon_reduce=reduce, previous_input=0.2, qbg_weights=dict(ccrc_weight_down=3.0, dd_linear_weight_down=2.0),
              integrator=integrator)
    J, g = OT.cost_and_grad(cost_id, s0, Q, 0.03, te, **kw)
    if regime == "edge" and integrator == "ODE_v0":
        traj = O.predict_core(s0, np.clip(Q, -1, 1).astype(f32))
        assert (np.abs(traj[:, :, O.POSITION_IDX]).max(axis=1) > 0.19).any()      # the bounce branch is exercised
    eps = 1e-6
    rng = np.random.Generator(np.random.SFC64(5))
    for _ in range(10):
        n, k = int(rng.integers(Q.shape[0])), int(rng.integers(Q.shape[1]))
        if abs(Q[n, k]) > 1.0 - 1e-3:
            assert g[n, k] == 0.0                                                   # clipped control: zero derivative
            continue
        Qp, Qm = Q.copy(), Q.copy()
        Qp[n, k] += eps
        Qm[n, k] -= eps
        fd = (OT.cost_and_grad(cost_id, s0, Qp, 0.03, te, **kw)[0][n] - OT.cost_and_grad(cost_id, s0, Qm, 0.03, te, **kw)[0][n]) / (2 * eps)
        # (a perturbation that moves a rollout across a bounce / indicator boundary would break this; eps is tiny)
        # + the cancellation floor of the difference quotient (the 1e7 edge indicator makes J ~ 1e10 in the edge regime)
        assert abs(fd - g[n, k]) <= 1e-4 * max(1.0, abs(fd)) + 4e-16 * abs(J[n]) / eps, (n, k, fd, g[n, k])
