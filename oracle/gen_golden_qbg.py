"""Golden vectors for the in-tree cost plugin quadratic_boundary_grad, produced by the reference's own
code (Control_Toolkit_ASF/Cost_Functions/CartPole/quadratic_boundary_grad.py) under the import stand-ins.
TEST INFRASTRUCTURE; usage:  cd /root/reference && python -B /root/repo/oracle/gen_golden_qbg.py"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
os.chdir(ref_shims.REFERENCE_ROOT)
from Control_Toolkit_ASF.Cost_Functions.CartPole.quadratic_boundary_grad import quadratic_boundary_grad  # noqa: E402
import Control_Toolkit_ASF.Controllers.controller_mppi_cartpole as LEG  # noqa: E402
from CartPole.state_utilities import create_cartpole_state  # noqa: E402

f32 = np.float32
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
N, H = 96, 30
LEG.num_rollouts, LEG.mpc_horizon = N, H
LEG.predictor.configure(batch_size=N, horizon=H, dt=0.02)
rng = np.random.Generator(np.random.SFC64(77))
out = {}
lib = ref_shims.NumpyLibrary()
cases = {"up_shipped": (1.0, {}), "down_shipped": (-1.0, {}),
         "up_all_terms": (1.0, dict(dd_linear_weight_up=7.0, ccrc_weight_up=3.0, target_angular_speed_sqr_max_correction_up=5.0,
                                    admissible_angle=0.35)),
         "down_all_terms": (-1.0, dict(dd_linear_weight_down=4.0, ccrc_weight_down=2.0, admissible_angle=0.35))}
for name, (te, overrides) in cases.items():
    s0 = create_cartpole_state(dict(angle=rng.uniform(-3, 3), angleD=rng.uniform(-6, 6), position=rng.uniform(-0.15, 0.15),
                                    positionD=rng.uniform(-0.3, 0.3)))
    Q = np.clip(0.6 * rng.standard_normal((N, H)), -1, 1).astype(f32)
    traj = LEG.predictor.predict(np.tile(s0, (N, 1)), Q[..., None])
    vp = SimpleNamespace(target_position=f32(rng.uniform(-0.1, 0.1)), target_equilibrium=f32(te))
    c = quadratic_boundary_grad(vp, lib)
    for k, v in overrides.items():
        setattr(c, k, np.array(v, dtype=f32))
    prev = f32(rng.uniform(-0.5, 0.5))
    stage = c.get_stage_cost(traj[:, :-1], Q[..., None], prev)
    total = c.get_trajectory_cost(traj, Q[..., None], prev)
    weights = {k: float(getattr(c, k)) for k in c.config}
    out[f"{name}/s0"], out[f"{name}/Q"], out[f"{name}/traj"] = s0, Q, traj
    out[f"{name}/stage"], out[f"{name}/total"] = np.asarray(stage, f32), np.asarray(total, f32)
    out[f"{name}/target_position"], out[f"{name}/target_equilibrium"], out[f"{name}/previous_input"] = vp.target_position, vp.target_equilibrium, prev
    out[f"{name}/weight_names"] = np.array(list(weights))
    out[f"{name}/weight_values"] = np.array(list(weights.values()), dtype=np.float64)
    print(name, "stage range", float(np.min(stage)), float(np.max(stage)))
out["names"] = np.array(list(cases))
np.savez_compressed(os.path.join(OUT, "qbg_costs.npz"), **out)
