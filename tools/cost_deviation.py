#!/usr/bin/env python3
"""Deviation of the FUSED rollout kernel's per-rollout costs from the golden reference costs (development tool, GPU).

tools/deviation.py looks at final states through the predictor seam, which always runs one rollout per lane; this tool
drives cpmppi_step itself — both lane mappings, i.e. also the packed two-rollouts-per-lane path with its hoisted test
and carried rotation — and compares S[N] with the reference's S_qbgm on all 8 x 1024 golden rollouts.
Usage: python tools/cost_deviation.py [lib.so ...]
"""
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 or (len(sys.argv) == 2 and not os.environ.get("_COSTDEV_CHILD")):
    keep = os.path.join(ROOT, "cartpolesimulation_amd", "libcpmppi.so")
    backup = keep + ".bak"
    shutil.copy(keep, backup)
    try:
        for lib in sys.argv[1:]:
            shutil.copy(lib, keep)
            subprocess.run([sys.executable, __file__, lib], env=dict(os.environ, _COSTDEV_CHILD="1"), check=True)
    finally:
        shutil.move(backup, keep)
    sys.exit(0)

from cartpolesimulation_amd.engine import MPPIEngine  # noqa: E402
from cartpolesimulation_amd.configs import MPPIConfig  # noqa: E402
from tests.test_gpu_parity import regen_delta_u  # noqa: E402

tag = os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "in-tree"
g = np.load(os.path.join(ROOT, "tests", "golden", "rollouts_c2.npz"))
N, H = int(g["N"]), int(g["H"])
for rpl in (1, 2):
    rels = []
    eng = MPPIEngine(1, MPPIConfig(num_rollouts=N, mpc_horizon=H, math_mode="fast", shift_mode="none", control_mode="clip",
                                   correction_u="u_nom", cc_weight=0.0, rollouts_per_lane=rpl))
    for name in g["names"]:
        _, du = regen_delta_u(g[f"{name}/seed"], N, H, g["stdev"])
        un = eng.tensor(g[f"{name}/u_nom"][None].astype(np.float32).copy())
        S = eng.empty(1, N)
        eng.step(g[f"{name}/s0"][None], un, float(g[f"{name}/target"]), 1.0,
                 delta_u=du[None], S_out=S)
        ref = g[f"{name}/clip/S_qbgm"]
        rels.append(np.abs(S.cpu().numpy()[0] - ref) / np.abs(ref))
    r = np.concatenate(rels)
    print(f"{tag:22s} rollouts/lane {rpl}: relative cost deviation  median {np.median(r):.2e}  p90 {np.percentile(r, 90):.2e}  "
          f"p99 {np.percentile(r, 99):.2e}  max {r.max():.2e}  (<1e-4: {np.mean(r < 1e-4):.4f})")
