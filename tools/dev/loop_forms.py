"""Development tool (GPU): the data generator's device loop launched or as a replayed graph, for a kernel trace of either form.
python tools/dev/loop_forms.py launched|graph [E]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cartpolesimulation_amd import schedule as SC  # noqa: E402
from cartpolesimulation_amd.configs import legacy_mppi_config  # noqa: E402
from cartpolesimulation_amd.engine import MPPIEngine  # noqa: E402
from cartpolesimulation_amd.harness import BatchedCartPoleExperiment  # noqa: E402

form, E = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 64
eng = MPPIEngine(E, legacy_mppi_config(num_rollouts=3500, mpc_horizon=35))
batch = SC.RandomExperimentSetter(dict(seed=1, length_of_experiment=4.0)).draw(E, 2)
exp = BatchedCartPoleExperiment(eng, seed=1)
for _ in range(2):
    t0 = time.perf_counter()
    exp.run_schedule(batch, graph=form == "graph")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"{form}: {dt / (batch.n_periods + 1) * 1e6:.1f} us per period")
