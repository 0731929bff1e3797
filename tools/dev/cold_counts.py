#!/usr/bin/env python3
"""Development tool (GPU): per launch of the rollout kernel at a small config — duration, rare-event counters and the
distribution of wave lifetimes — from a -DCPMPPI_DEBUG_COUNTERS build.

  python __graft_entry__.py --variant dbg -DCPMPPI_DEBUG_COUNTERS=1
  CPMPPI_LIB=build_variants/dbg.so python tools/dev/cold_counts.py --config C4 --rpl 2 --steps 24
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synthetic_inputs  # noqa: E402
from cartpolesimulation_amd.engine import MPPIEngine  # noqa: E402
from cartpolesimulation_amd.configs import MPPIConfig  # noqa: E402
from cartpolesimulation_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C4")
ap.add_argument("--rpl", type=int, default=2)
ap.add_argument("--steps", type=int, default=24)
ap.add_argument("--frozen", action="store_true")
args = ap.parse_args()
E, N, H = {"C2": (8192, 1024, 50), "C3": (64, 4096, 100), "C4": (64, 2048, 50)}[args.config]
dev = torch.device("cuda", 0)
eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=args.rpl), device=0)
lib = L.load()
lib.cpmppi_debug_read.argtypes = [C.POINTER(C.c_uint), C.POINTER(C.c_ulonglong), C.c_void_p, C.c_uint, C.c_int]
s0, tp, te, Lt = synthetic_inputs(E, H, 2, dev)
u_nom = eng.zeros(E, H)
rpl = args.rpl or 2
n_waves = min(16384, E * ((N + 256 * rpl - 1) // (256 * rpl)) * 4)
cnt = (C.c_uint * n_waves)()
wc = (C.c_ulonglong * n_waves)()
lib.cpmppi_debug_read(cnt, wc, None, n_waves, 1)
rows = []
for i in range(args.steps):
    if args.frozen:
        u_nom.zero_()
    eng.set_profiling(True)
    eng.step(s0, u_nom, tp, te, L=Lt, seed=1234, offset=i)
    torch.cuda.synchronize()
    r, _ = eng.get_profile()
    eng.set_profiling(False)
    assert lib.cpmppi_debug_read(cnt, wc, None, n_waves, 1) == 0
    w = np.frombuffer(wc, dtype=np.uint64).astype(np.float64)
    c = np.frombuffer(cnt, dtype=np.uint32).astype(np.float64)
    ok = w > 0
    w, c = w[ok], c[ok]
    base = float(np.median(w[c == 0])) if np.any(c == 0) else float(w.min())
    big = c >= 20
    per_event = float(np.median((w[big] - base) / c[big])) if np.any(big) else None
    top = np.argsort(-w)[:4]
    rows.append({"step": i, "kernel_us": round(float(r[0]) * 1e3, 1), "cold_entries_total": int(c.sum()),
                 "waves_with_20plus": int(big.sum()), "cycles_per_cold_entry": None if per_event is None else round(per_event),
                 "base_kcycles": round(base / 1e3, 1),
                 "slowest_waves": [(round(float(w[j]) / 1e3, 1), int(c[j])) for j in top]})
for r in rows:
    print(json.dumps(r))
