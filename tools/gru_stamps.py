#!/usr/bin/env python3
"""Development tool (GPU): per-phase cycle breakdown of the GRU rollout kernel from a -DCPMPPI_GRU_STAMPS build.
Usage: python tools/gru_stamps.py build_variants/gru_stamps.so [envs]"""
import ctypes as C
import shutil
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_path = sys.argv[1]
E = int(sys.argv[2]) if len(sys.argv) > 2 else 256
os.environ["CPMPPI_LIB"] = os.path.abspath(lib_path)          # (the stamped build is loaded instead of the product library)
import numpy as np
import torch
from cartpolesimulation_amd.engine import MPPIEngine
from cartpolesimulation_amd.configs import MPPIConfig
from cartpolesimulation_amd import _lib as L

eng = MPPIEngine(E, MPPIConfig(num_rollouts=1024, mpc_horizon=50))
rng = np.random.Generator(np.random.SFC64(5))
u = lambda *s: rng.uniform(-1, 1, s).astype(np.float32) / np.sqrt(32.0, dtype=np.float32)
eng.set_gru(dict(w_ih0=u(96, 6), w_hh0=u(96, 32), b_ih0=u(96), b_hh0=u(96), w_ih1=u(96, 32), w_hh1=u(96, 32),
                 b_ih1=u(96), b_hh1=u(96), w_out=u(5, 32), b_out=u(5)))
s0 = np.tile(np.array([0.1, 0, np.cos(0.1), np.sin(0.1), 0, 0], np.float32), (E, 1))
un = eng.zeros(E, 50)
tp, te = np.zeros(E, np.float32), np.ones(E, np.float32)
lib = L.load()
out = (C.c_ulonglong * 8)()
for it in range(3):
    eng.step(s0, un, tp, te, seed=1, offset=it, predictor="GRU")
    torch.cuda.synchronize()
    lib.cpmppi_debug_gru_stamps(out, 1)
names = ["misc (cost, noise, x tile) between steps", "L1 x-products (+ x split in the f16 path)", "W_hh2 h2 products + gates 1 (+ h1 split)",
         "L2 x-products", "W_hh1 h1 products (next step) + gates 2 (+ h2 split)", "head"]
waves = out[6]
tot = sum(out[i] for i in range(6))
print(f"waves {waves}, cycles per wave per step (s_memtime ticks = shader cycles):")
for i in range(6):
    print(f"  {names[i]:45s} {out[i] / waves / 50:9.0f}  ({100.0 * out[i] / tot:4.1f} %)")
print(f"  total {tot / waves / 50:9.0f}")
