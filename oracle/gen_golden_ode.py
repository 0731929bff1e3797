"""Golden vectors for the in-tree predictor_type "ODE" (SI_Toolkit_ASF/ToolkitCustomization/predictors_customization.py:25-69,
next_state_predictor_ODE -> CartPole/cartpole_equations.py:181-259,293-308: Euler-Cromer substeps, no edge bounce, angle =
atan2(sin, cos)), produced by the reference's own class under the import stand-ins of oracle/ref_shims.py (NumpyLibrary
stand-in: numpy float32 throughout; its `loop` follows the contract visible at cartpole_equations.py:251-258).
TEST INFRASTRUCTURE; usage:  cd /root/reference && python -B /root/repo/oracle/gen_golden_ode.py"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
os.chdir(ref_shims.REFERENCE_ROOT)
from SI_Toolkit_ASF.ToolkitCustomization.predictors_customization import next_state_predictor_ODE  # noqa: E402
from CartPole.state_utilities import create_cartpole_state as _ccs  # noqa: E402


def create_cartpole_state(a, ad, x, xd):
    return _ccs({"angle": a, "angleD": ad, "position": x, "positionD": xd})


OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
f32 = np.float32
DT, S_SUB = 0.02, 10
lib = ref_shims.NumpyLibrary()
rng = np.random.Generator(np.random.SFC64(77))
out = {}


def stepper(dt=DT, S=S_SUB, **vp):
    return next_state_predictor_ODE(dt, S, lib, batch_size=1, variable_parameters=SimpleNamespace(**vp) if vp else None,
                                    disable_individual_compilation=True)


def rollout(ns, s0, Q):
    """predict_core as ODE_module.py:46-50 drives it: out[:, 0] = s0, out[:, k+1] = step(out[:, k], Q[:, k, newaxis])."""
    N, H = Q.shape
    traj = np.zeros((N, H + 1, 6), dtype=f32)
    traj[:, 0] = s0
    for k in range(H):
        traj[:, k + 1] = ns.step(traj[:, k], Q[:, k, np.newaxis])
    return traj


# single control steps from random states (incl. beyond the track edge: this predictor does not bounce)
N = 256
s = np.stack([create_cartpole_state(a, ad, x, xd) for a, ad, x, xd in
              zip(rng.uniform(-np.pi, np.pi, N), rng.uniform(-12, 12, N), rng.uniform(-0.25, 0.25, N), rng.uniform(-1.5, 1.5, N))]).astype(f32)
Q = rng.uniform(-1, 1, (N, 1)).astype(f32)
out["kat/s"], out["kat/Q"] = s, Q[:, 0]
out["kat/s_next"] = stepper().step(s, Q)
out["kat/s_next_L030"] = stepper(L=f32(0.30)).step(s, Q)
out["kat/s_next_mpole"] = stepper(m_pole=f32(0.12)).step(s, Q)
out["kat/s_next_dt04_S4"] = stepper(dt=0.04, S=4).step(s, Q)

# rollouts in five regimes: upright, hanging, leaving the track, spinning, spinning beyond 125 rad/s
H = 50
for name, (a, ad, x, xd), amp in (("upright", (0.05, 0.0, 0.0, 0.0), 0.4), ("hanging", (np.pi, 0.0, 0.05, 0.0), 0.8),
                                  ("edge", (0.3, 0.0, 0.17, 0.6), 0.9), ("spin", (1.0, 45.0, 0.0, 0.0), 0.5),
                                  ("fastspin", (-2.0, 150.0, 0.0, 0.0), 0.5)):       # |w t| > 0.25 rad per substep
    n = 32
    s0 = np.tile(create_cartpole_state(a, ad, x, xd).astype(f32), (n, 1))
    Qr = np.clip(amp * rng.standard_normal((n, H)), -1, 1).astype(f32)
    out[f"{name}/s0"], out[f"{name}/Q"] = s0[0], Qr
    out[f"{name}/traj"] = rollout(stepper(), s0, Qr)
    print(name, "max |x|", float(np.abs(out[f"{name}/traj"][:, :, 4]).max()), "max |w|", float(np.abs(out[f"{name}/traj"][:, :, 1]).max()))
np.savez_compressed(os.path.join(OUT, "ode_predictor.npz"), **out)
print("wrote", os.path.join(OUT, "ode_predictor.npz"), os.path.getsize(os.path.join(OUT, "ode_predictor.npz")), "bytes")
