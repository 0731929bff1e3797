"""Development tool (GPU): the data generator at scale with EVERY table of the schedule on - 1024 experiments x 60 s (30 000 simulation
steps each), shipped MPPI size 3500 x 35, pole length / mass updaters, switching informer, control disturbance, latency, measurement
noise, moving angle offset - end to end into CSV files; checks that every recording is finite and complete.
python tools/dev/gen_soak.py [E] [length]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cartpolesimulation_amd import recording as R  # noqa: E402
from cartpolesimulation_amd.configs import legacy_mppi_config  # noqa: E402
from cartpolesimulation_amd.engine import MPPIEngine  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
length = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
upd = lambda init, every, inc, lo, hi, mode: dict(init_value=init, change_every_x_seconds=every, mode=mode, range_random=[lo, hi],   # noqa: E731
                                                  range_clip=[lo, hi], increment=inc, reset_every_x_seconds="inf")
prm = dict(L=upd(0.395, 7, 0.02, 0.2, 0.5, "random walk"), m_pole=upd(0.087, 2, 0.002, 0.015, 0.15, "random walk"),
           inform_controller_about_parameters_change=dict(mode="switching_random", change_to_on_after_x_seconds_off=1.5,
                                                          change_to_off_after_x_seconds_on=4),
           controlDisturbance=0.2, controlBias=0.0, seed=17, latency=0.005,
           noise=dict(noise_mode="ON", sigma_angle=0.0, sigma_position=0.0005, sigma_angleD=0.075, sigma_positionD=0.005),
           vertical_angle_offset=upd(0.0, 1.0, 0.02, -0.2, 0.2, "random walk"))
eng = MPPIEngine(E, legacy_mppi_config(num_rollouts=3500, mpc_horizon=35))
out = tempfile.mkdtemp()
t0 = time.perf_counter()
paths = R.generate_dataset(eng, E, out, config=dict(length_of_experiment=length), seed=3, parameters=prm)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
size = sum(os.path.getsize(p) for p in paths)
rows = int(round(length / 0.02)) + 1
print(f"{E} experiments x {length} s, every schedule table on: {dt:.2f} s end to end, {len(paths)} files, {size / 1e6:.0f} MB "
      f"({E * (rows - 1) / dt:.3g} control steps/s incl. tabulation and files)")
bad = 0
for p in paths[:: max(1, E // 16)]:
    d = pd.read_csv(p, comment="#")
    ok = len(d) == rows and np.isfinite(d.drop(columns=["Q_update_time", "L_for_controller", "m_pole_for_controller"]).to_numpy(dtype=float)).all()
    ok = ok and d["position"].abs().max() <= 0.1985 and len(np.unique(d["L"])) > 1 and len(np.unique(d["m_pole"])) > 3
    ok = ok and set(d["L_for_controller"]) == {"true", "default"} and (d["Q_applied"] - d["Q_calculated"]).abs().max() > 0.1
    bad += not ok
print(f"checked {len(paths[:: max(1, E // 16)])} recordings: {bad} bad; upright fraction of the last 10 s: "
      f"{np.mean([ (pd.read_csv(p, comment='#')['angle_cos'].to_numpy()[-500:] > 0.9).mean() for p in paths[:: max(1, E // 16)]]):.2f}")
shutil.rmtree(out)
sys.exit(1 if bad else 0)
