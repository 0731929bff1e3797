#!/usr/bin/env python3
"""Runs on the GPU box: the numbers of BASELINE.md §4's table — for each BASELINE config the CPU port (one thread, all
threads; the fastest of the three builds of oracle/cpmppi_oracle.c, see bench.cpu_baseline) and the HIP path on one
MI355X (kernel time from HIP events over a few hundred steps).  -> JSON on stdout."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONFIGS = {"C1": (1, 256, 20), "C2": (1, 1024, 50), "C2x8192": (8192, 1024, 50), "C3": (64, 4096, 100), "C4": (64, 2048, 50)}
out = {"cpu": bench.cpu_model()}
ctx = {"world": 1, "rank": 0, "local_rank": 0, "device": torch.device("cuda", 0), "collective": False, "backend": "nccl"}
torch.cuda.set_device(0)
for name, (E, N, H) in CONFIGS.items():
    steps = 20 if E >= 4096 else 400
    w = bench.Workload(ctx, E, N, H)
    r = w.run(steps, 10)
    w.close()
    rec = {"E": E, "N": N, "H": H, "gpu_rollouts_per_s": r["value"], "kernel_ms": r["kernel_ms"], "kernel_ms_min": r["kernel_ms_min"],
           "hbm_fraction": r["alg_gbs"] / bench.HBM_PEAK_GBS, "valu_fraction": r["valu_tflops"] / bench.FP32_VALU_PEAK_TFLOPS}
    if name != "C2x8192":
        c = bench.cpu_baseline(N, H, budget_s=4.0)
        best = max((v for v in c["builds"].values() if isinstance(v, dict)), key=lambda v: v["all_cores"]["value"])
        rec.update(cpu_all_cores=best["all_cores"]["value"], cpu_cores=c["cores"], cpu_one_core=best["one_core"]["value"], cpu_build=best["flags"])
    out[name] = rec
print(json.dumps(out))
