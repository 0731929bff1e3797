"""Golden vectors for the in-tree cost plugin quadratic_boundary (SURVEY 8f N4 "the remaining cost plugins"), produced by the
reference's own class (Control_Toolkit_ASF/Cost_Functions/CartPole/quadratic_boundary.py) under the import stand-ins.  The
class still implements the stale `_get_stage_cost` name, which is called directly (:79-87); `get_terminal_cost` :43-67.
Its sibling quadratic_boundary_nonconvex cannot be imported at all in the reference (KeyError 'cem_ccrc_weight' against the
shipped config_cost_function.yml:47-52): the script records that fact.  It then imports the module a second time with ONE key added
to what yaml.safe_load returns for that section - cem_ccrc_weight := the section's own ccrc_weight (the key the module asks for was
renamed in the YAML) - and records the outputs of the reference's class under that augmented configuration ("nc/..." arrays): the
module's code is the reference's, unmodified; the augmentation is stated in the fixture.
TEST INFRASTRUCTURE; usage:  cd /root/reference && python -B /root/repo/oracle/gen_golden_qb.py"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
os.chdir(ref_shims.REFERENCE_ROOT)
from Control_Toolkit_ASF.Cost_Functions.CartPole.quadratic_boundary import quadratic_boundary  # noqa: E402
import Control_Toolkit_ASF.Cost_Functions.CartPole.quadratic_boundary as QB  # noqa: E402
import Control_Toolkit_ASF.Controllers.controller_mppi_cartpole as LEG  # noqa: E402
from CartPole.state_utilities import create_cartpole_state  # noqa: E402

try:
    import Control_Toolkit_ASF.Cost_Functions.CartPole.quadratic_boundary_nonconvex  # noqa: F401
    nonconvex_import = "ok"
except Exception as e:                                      # KeyError('cem_ccrc_weight') with the shipped YAML
    nonconvex_import = repr(e)

f32 = np.float32
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
N, H = 96, 30
LEG.num_rollouts, LEG.mpc_horizon = N, H
LEG.predictor.configure(batch_size=N, horizon=H, dt=0.02)
rng = np.random.Generator(np.random.SFC64(78))
out = {"nonconvex_import": np.array(nonconvex_import)}
lib = ref_shims.NumpyLibrary()
cases = {"up_centre": (1.0, 0.0, 0.1), "up_edge": (1.0, 0.17, 0.5), "down_edge": (-1.0, -0.18, -0.4), "up_no_previous": (1.0, 0.1, None)}
for name, (te, x0, prev) in cases.items():
    s0 = create_cartpole_state(dict(angle=rng.uniform(-3, 3), angleD=rng.uniform(-6, 6), position=x0, positionD=rng.uniform(-0.3, 0.3)))
    Q = np.clip(0.6 * rng.standard_normal((N, H)), -1, 1).astype(f32)
    traj = LEG.predictor.predict(np.tile(s0, (N, 1)), Q[..., None])
    vp = SimpleNamespace(target_position=f32(rng.uniform(-0.1, 0.1)), target_equilibrium=f32(te))
    c = quadratic_boundary(vp, lib)
    prev_in = None if prev is None else f32(prev)
    stage = c._get_stage_cost(traj[:, :-1], Q[..., None], prev_in)
    term = c.get_terminal_cost(traj[:, -1])
    out[f"{name}/s0"], out[f"{name}/Q"], out[f"{name}/traj"] = s0, Q, traj
    out[f"{name}/stage"], out[f"{name}/terminal"] = np.asarray(stage), np.asarray(term)
    out[f"{name}/target_position"], out[f"{name}/target_equilibrium"] = vp.target_position, vp.target_equilibrium
    out[f"{name}/previous_input"] = np.array(np.nan if prev is None else prev, dtype=f32)
    print(name, "stage range", float(np.min(stage)), float(np.max(stage)), "dtype", np.asarray(stage).dtype, "beyond 0.95 THL:",
          int((np.abs(traj[:, :-1, 4]) > 0.95 * 0.198).sum()))
# ---- quadratic_boundary_nonconvex under the one-key augmentation of the configuration
import yaml  # noqa: E402

_safe_load = yaml.safe_load


def _augmented(stream):
    cfg = _safe_load(stream)
    sec = cfg.get("CartPole", {}).get("quadratic_boundary_nonconvex") if isinstance(cfg, dict) else None
    if isinstance(sec, dict) and "cem_ccrc_weight" not in sec:
        sec["cem_ccrc_weight"] = sec["ccrc_weight"]
    return cfg


yaml.safe_load = _augmented
sys.modules.pop("Control_Toolkit_ASF.Cost_Functions.CartPole.quadratic_boundary_nonconvex", None)
import Control_Toolkit_ASF.Cost_Functions.CartPole.quadratic_boundary_nonconvex as NC  # noqa: E402
yaml.safe_load = _safe_load
out["nc/augmentation"] = np.array("CartPole.quadratic_boundary_nonconvex.cem_ccrc_weight := ccrc_weight (%r)" % NC.ccrc_weight)
for name in cases:
    traj, Q = out[f"{name}/traj"], out[f"{name}/Q"]
    vp = SimpleNamespace(target_position=out[f"{name}/target_position"], target_equilibrium=out[f"{name}/target_equilibrium"])
    c = NC.quadratic_boundary_nonconvex(vp, lib)
    prev = out[f"{name}/previous_input"]
    prev_in = None if np.isnan(prev) else f32(prev)
    out[f"nc/{name}/stage"] = np.asarray(c._get_stage_cost(traj[:, :-1], Q[..., None], prev_in))
    out[f"nc/{name}/terminal"] = np.asarray(c.get_terminal_cost(traj[:, -1]))
    print("nonconvex", name, "stage range", float(out[f"nc/{name}/stage"].min()), float(out[f"nc/{name}/stage"].max()))
out["nc/weights"] = np.array([NC.dd_weight, NC.ep_weight, NC.cc_weight, NC.R, NC.ccrc_weight], dtype=np.float64)
out["weights"] = np.array([QB.dd_weight, QB.ep_weight, QB.cc_weight, QB.R, QB.ccrc_weight], dtype=np.float64)
out["names"] = np.array(list(cases))
np.savez_compressed(os.path.join(OUT, "qb_costs.npz"), **out)
print("nonconvex import:", nonconvex_import)
