"""Golden vectors for the legacy sampler's remaining modes (controller_mppi_cartpole.py:414-450: random_walk, uniform,
repeated, iid), produced by the reference's own `initialize_perturbations` on Generator(SFC64(seed)) under the import
stand-ins.  TEST INFRASTRUCTURE; usage:  cd /root/reference && python -B /root/repo/oracle/gen_golden_sampler_modes.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (installs the stand-ins, imports the reference controller module as G.LEG)

f32 = np.float32
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
N, H, SEED = 96, 23, 4321
stdev = 0.02 / np.sqrt(G.DT)
out = {"N": np.int64(N), "H": np.int64(H), "seed": np.int64(SEED), "stdev": np.float64(stdev)}
for mode in ("random_walk", "uniform", "repeated", "iid", "interpolated"):
    ctrl = G.make_legacy_controller(SEED, N, H, 0.0)             # fresh stream: 5 configure() draws, then the sampler
    du = ctrl.initialize_perturbations(stdev=stdev, sampling_type=mode)
    assert du.shape == (N, H), (mode, du.shape)            # (dtype: float32 where the reference fills a float32 array, float64 where it returns stdev * normal)
    out[mode] = du
    print(mode, du[0, :4], float(du.astype(np.float64).sum()))
np.savez_compressed(os.path.join(OUT, "sampler_modes.npz"), **out)
print("written sampler_modes.npz")
