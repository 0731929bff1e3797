#!/usr/bin/env python3
"""Development tool (GPU): cProfile of the reference's shipped controller pairing (optimizer rpgd on predictor "ODE") through
controller_mpc.step, one env: ~0.87 ms per control step = ~0.48 ms of device work (four adjoint launches of 16 plans x 350 substeps,
each a partially filled lone wave) + ~0.38 ms of host-side tensor bookkeeping."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cartpolesimulation_amd.controller_mpc import controller_mpc
from oracle import oracle_np as O
ctrl = controller_mpc("CartPole", {"target_position": 0.0, "target_equilibrium": 1.0, "L": 0.395}, control_limits=([-1.0], [1.0]), config=dict(seed=1))
ctrl.configure("rpgd", predictor_specification="ODE")
opt = ctrl.optimizer
s = O.create_cartpole_state(0.1, 0.2, 0.01, 0.0)
for _ in range(20): ctrl.step(s, 0.0, {"target_position": 0.0, "m_pole": 0.087})
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(200): ctrl.step(s, 0.0, {"target_position": 0.0, "m_pole": 0.087})
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
pr.disable()
print("ms per step", dt * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
