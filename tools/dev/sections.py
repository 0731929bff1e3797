#!/usr/bin/env python3
"""Development tool (GPU): where the horizon loop of the mid-size build's rollout kernel spends a wave's time — s_memtime
at the section boundaries of every control step, summed per wave (-DCPMPPI_DEBUG_COUNTERS -DCPMPPI_SECTION_STAMPS build).

  python __graft_entry__.py --variant sec -DCPMPPI_DEBUG_COUNTERS=1 -DCPMPPI_SECTION_STAMPS=1
  CPMPPI_LIB=build_variants/sec.so python tools/dev/sections.py --config C4

Sections (shader cycles per control step, median wave and slowest wave; the stamp's own cost — section 0, two adjacent
stamps — is subtracted from every other section once per stamp):
  1 nominal control + clamp + stage cost + correction      2 rotation seed + spin test
  3 intermediate substeps (3 triples / event loop)         4 last substep (wrap + sincos + near test)
  5 between control steps (noise knots, interpolation, loop bookkeeping)
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
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--noise", default="philox")
ap.add_argument("--reader", default=None, help="cpmppi_debug_sections_{latency,mid,throughput} (default: by size)")
args = ap.parse_args()
E, N, H = {"C3": (64, 4096, 100), "C4": (64, 2048, 50), "E256": (256, 1024, 50), "C2": (1, 1024, 50), "C1": (1, 256, 20)}[args.config]
latency = E * N < 131072                                   # one rollout per lane: the latency build
dev = torch.device("cuda", 0)
eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=1 if latency else 2), device=0)
lib = L.load()
fn = getattr(lib, args.reader) if args.reader else (lib.cpmppi_debug_sections_latency if latency else lib.cpmppi_debug_sections_mid)
fn.argtypes = [C.c_void_p, C.c_uint]
n_waves = min(16384, E * ((N + (255 if latency else 511)) // (256 if latency else 512)) * 4)
s0, tp, te, Lt = synthetic_inputs(E, H, 2, dev)
u_nom = eng.zeros(E, H)
sec = np.zeros((n_waves, 8), np.uint32)
names = ["stamp", "pre (nominal, clamp, cost)", "seed + spin test", "intermediate substeps", "last substep", "between steps"]
for i in range(args.steps):
    eng.set_profiling(True)
    eng.step(s0, u_nom, tp, te, L=Lt, seed=1234, offset=i)
    torch.cuda.synchronize()
    r, _ = eng.get_profile()
    eng.set_profiling(False)
    assert fn(sec.ctypes.data, n_waves) == 0
    c = sec[:, :6].astype(np.float64) / H                      # cycles per control step
    stamp = c[:, 0]
    body = c[:, 1:6] - stamp[:, None]                         # every section ends with one stamp
    total = body.sum(axis=1)
    slow = int(np.argmax(total))
    out = {"step": i, "event_us": round(float(r[0]) * 1e3, 1), "stamp_cycles": round(float(np.median(stamp)), 1),
           "total_cycles_per_step_p50": round(float(np.median(total)), 1), "slowest": round(float(total[slow]), 1)}
    for k in range(5):
        out[names[k + 1]] = [round(float(np.median(body[:, k])), 1), round(float(body[slow, k]), 1)]
    print(json.dumps(out))
