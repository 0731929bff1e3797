#!/usr/bin/env python3
"""Development tool (GPU): WHERE do the waves of a small rollout launch run?  Per wave: HW_REG_HW_ID / HW_REG_XCC_ID at entry plus the
entry / exit time stamps of the diagnostic build -> waves per SIMD and per CU, and the lifetime of a wave against the number of waves
it shares its SIMD with.

  python __graft_entry__.py --variant dbg -DCPMPPI_DEBUG_COUNTERS=1 [-DCPMPPI_DEV_KNOBS   for --lds-pad]
  CPMPPI_LIB=build_variants/dbg.so python tools/dev/placement.py --config C4 [--rpl 0|1|2] [--lds-pad BYTES]
"""
import argparse
import collections
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
ap.add_argument("--rpl", type=int, default=0)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--lds-pad", type=int, default=0)
args = ap.parse_args()
if args.lds_pad:
    os.environ["CPMPPI_LDS_PAD"] = str(args.lds_pad)
E, N, H = {"C3": (64, 4096, 100), "C4": (64, 2048, 50), "C4half": (32, 2048, 50), "E256": (256, 1024, 50)}[args.config]
dev = torch.device("cuda", 0)
eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=args.rpl), device=0)
lib = L.load()
s0, tp, te, Lt = synthetic_inputs(E, H, 3, dev)
u_nom = eng.zeros(E, H)
for i in range(args.steps):
    eng.set_profiling(True)
    eng.step(s0, u_nom, tp, te, L=Lt, seed=1234, offset=i)
    torch.cuda.synchronize()
    r, _ = eng.get_profile()
    eng.set_profiling(False)
    info = eng.last_launch()
    unit = {0: "latency", 1: "throughput", 2: "mid", 3: "mid"}[info["build_variant"]]
    n_waves = min(16384, info["blocks"] * 4)
    rd = getattr(lib, {"latency": "cpmppi_debug_read_latency", "mid": "cpmppi_debug_read_mid", "throughput": "cpmppi_debug_read"}[unit])
    rd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_int]
    hwf = getattr(lib, f"cpmppi_debug_hw_{unit}")
    hwf.argtypes = [C.c_void_p, C.c_uint]
    stamps = np.zeros((n_waves, 4), np.uint64)
    hw = np.zeros((n_waves, 2), np.uint32)
    assert rd(None, None, stamps.ctypes.data, n_waves, 0) == 0 and hwf(hw.ctypes.data, n_waves) == 0
    t = (stamps.astype(np.float64) - stamps[:, 0].min()) / 100.0
    life = t[:, 1] - t[:, 0]
    simd, cu, sh, se, xcc = (hw[:, 0] >> 4) & 3, (hw[:, 0] >> 8) & 15, (hw[:, 0] >> 12) & 1, (hw[:, 0] >> 13) & 7, hw[:, 1] & 15
    cu_key = [(int(x), int(a), int(b), int(c)) for x, a, b, c in zip(xcc, se, sh, cu)]
    simd_key = [k + (int(s),) for k, s in zip(cu_key, simd)]
    per_simd, per_cu = collections.Counter(simd_key), collections.Counter(cu_key)
    share = np.array([per_simd[k] for k in simd_key])
    by_share = {int(k): (int((share == k).sum()), round(float(np.median(life[share == k])), 1), round(float(life[share == k].max()), 1))
                for k in sorted(set(share))}
    print(json.dumps({"step": i, "kernel": info["kernel"], "blocks": info["blocks"], "waves": n_waves, "event_us": round(float(r[0]) * 1e3, 1),
                      "cus_used": len(per_cu), "simds_used": len(per_simd), "xccs": sorted(set(int(x) for x in xcc)),
                      "waves_per_simd_hist": dict(sorted(collections.Counter(per_simd.values()).items())),
                      "waves_per_cu_hist": dict(sorted(collections.Counter(per_cu.values()).items())),
                      "lifetime_us_by_waves_on_the_same_simd {n: (waves, median, max)}": by_share,
                      "entry_last_us": round(float(t[:, 0].max()), 1), "loop_end_max_us": round(float(t[:, 1].max()), 1),
                      "exit_max_us": round(float(t[:, 3].max()), 1)}))
