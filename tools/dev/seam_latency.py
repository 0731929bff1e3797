#!/usr/bin/env python3
"""Development tool (GPU): what a CartPoleSimulation caller sees per control step through the controller seam —
controller_mpc.step(s: float32[6] on the host, time, updated_attributes) -> Q on the host — against the rollout kernel's
own duration, for the reference's single-env shapes.   python tools/dev/seam_latency.py"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cartpolesimulation_amd.controller_mpc import controller_mpc  # noqa: E402
from cartpolesimulation_amd.controller_mppi_cartpole import controller_mppi_cartpole  # noqa: E402
from oracle import oracle_np as O  # noqa: E402

s = O.create_cartpole_state(0.1, 0.0, 0.0, 0.0)
for N, H in ((1024, 50), (3500, 35), (256, 20)):
    ctrl = controller_mpc("CartPole", {"target_position": 0.0, "target_equilibrium": 1.0, "L": 0.395},
                          control_limits=([-1.0], [1.0]), config=dict(num_rollouts=N, mpc_horizon=H, seed=1))
    ctrl.configure("mppi")
    attrs = {"target_position": np.float32(0.02), "target_equilibrium": np.float32(1.0), "L": np.float32(0.395)}
    for _ in range(20):
        ctrl.step(s, 0.0, attrs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 500
    for i in range(K):
        ctrl.step(s, 0.02 * i, attrs)
    dt = (time.perf_counter() - t0) / K
    eng = ctrl.optimizer.engine
    import ctypes as C
    ht = (C.c_double * 4)()
    eng.lib.cpmppi_debug_host_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    eng.lib.cpmppi_debug_host_times(eng._h, ht)              # (resets)
    t0 = time.perf_counter()
    for i in range(K):
        ctrl.step(s, 0.02 * i, attrs)
    dt2 = (time.perf_counter() - t0) / K
    eng.lib.cpmppi_debug_host_times(eng._h, ht)
    breakdown = {"calls": int(ht[3]), "stage_us": round(ht[0] * 1e6, 2), "launch_call_us": round(ht[1] * 1e6, 2),
                 "spin_us": round(ht[2] * 1e6, 2), "python_us": round((dt2 - ht[0] - ht[1] - ht[2]) * 1e6, 2)}
    eng.set_profiling(True, group=10)
    for i in range(50):
        ctrl.step(s, 0.0, attrs)
    k_ms, _ = eng.get_profile()
    eng.set_profiling(False)
    print(json.dumps({"seam": "controller_mpc('mppi').step", "rollouts": N, "horizon": H, "us_per_call": round(dt * 1e6, 1),
                      "kernel_us": round(float(np.mean(k_ms)) * 1e3, 1), "host_path": breakdown}), flush=True)
leg = controller_mppi_cartpole("CartPole", {"target_position": 0.0}, control_limits=([-1.0], [1.0]),
                               config=dict(seed=1, num_rollouts=3500, mpc_horizon=35, predictor_specification="ODE_v0"))
leg.configure()
for _ in range(10):
    leg.step(s, 0.0, {"target_position": 0.0})
t0 = time.perf_counter()
for i in range(200):
    leg.step(s, 0.0, {"target_position": 0.0})
print(json.dumps({"seam": "controller_mppi_cartpole.step (SFC64 knots on the host)", "rollouts": 3500, "horizon": 35,
                  "us_per_call": round((time.perf_counter() - t0) / 200 * 1e6, 1)}))
