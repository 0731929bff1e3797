#!/usr/bin/env python3
"""Development tool (GPU, run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES`): three steps of the headline shape with inputs on
which NO rollout can reach the track edge (carts at the centre, at rest, perturbations 1e-3 of the default) or with bench.py's
inputs - the difference of SQ_INSTS_VALU per wave between the two is what edge events cost.

  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d out -- python3 tools/dev/valu_split.py quiet|bench"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synthetic_inputs  # noqa: E402
from cartpolesimulation_amd.engine import MPPIEngine  # noqa: E402
from cartpolesimulation_amd.configs import MPPIConfig  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bench"
E, N, H = (int(sys.argv[2]) if len(sys.argv) > 2 else 8192), 1024, 50
dev = torch.device("cuda", 0)
kw = dict(SQRTRHOINV=0.03e-3) if mode == "quiet" else {}
eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=(2 if E * N >= 131072 else 0), **kw), device=0)
s0, tp, te, Lt = synthetic_inputs(E, H, 2, dev)
if mode == "quiet":
    s0[:, 4] = 0.0
    s0[:, 5] = 0.0
u_nom = eng.zeros(E, H)
for i in range(3 if E > 64 else 30):
    eng.step(s0, u_nom, tp, te, L=Lt, seed=1234, offset=i)
torch.cuda.synchronize()
print(mode, eng.last_launch())
