#!/usr/bin/env python3
"""A/B timing of libcpmppi builds in ONE process (interleaved rounds, HIP events on the launch stream).

Usage: python tools/kbench.py lib_a.so lib_b.so ... [--envs 2048] [--rounds 6] [--noise philox buffer knots]
Development tool (not part of the product): every variant is loaded with ctypes under its own path.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cartpolesimulation_amd import _lib as L  # noqa: E402
from cartpolesimulation_amd.configs import MPPIConfig, build_c_config  # noqa: E402
sys.path.insert(0, ROOT)
from bench import synthetic_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--envs", type=int, default=2048)
    ap.add_argument("--rollouts", type=int, default=1024)
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--noise", nargs="+", default=["philox", "buffer"])
    ap.add_argument("--math", nargs="+", default=["fast"])
    ap.add_argument("--rpl", type=int, default=0, help="rollouts per lane: 0 auto, 1, 2")
    args = ap.parse_args()
    E, N, H = args.envs, args.rollouts, args.horizon
    dev = torch.device("cuda", 0)
    s0, tp, te, Lt = synthetic_inputs(E, H, 2, dev)
    u_nom = torch.zeros(E, H, device=dev)
    Q = torch.empty(E, device=dev)
    du = torch.empty(E, N, H, device=dev)
    P = MPPIConfig(num_rollouts=N, mpc_horizon=H).num_knots
    kn = torch.empty(E, N, P, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    variants = []
    vp, u32, u64, f = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
    for path in args.libs:
        lib = C.CDLL(os.path.abspath(path))
        lib.cpmppi_create.argtypes = [C.POINTER(L.cpmppi_config), C.c_int, C.POINTER(vp)]
        lib.cpmppi_step.argtypes = [vp, C.POINTER(L.cpmppi_step_args), vp]
        lib.cpmppi_sample.argtypes = [vp, u32, u64, u64, u32, vp, vp, vp]
        lib.cpmppi_set_profiling.argtypes = [vp, C.c_int]
        lib.cpmppi_get_profile.argtypes = [vp, C.POINTER(f), C.POINTER(f), u32, C.POINTER(u32)]
        lib.cpmppi_last_error.restype = C.c_char_p
        lib.cpmppi_last_error.argtypes = [vp]
        for math in args.math:
            cfg = build_c_config(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, math_mode=math, rollouts_per_lane=args.rpl))
            h = vp()
            rc = lib.cpmppi_create(C.byref(cfg), 0, C.byref(h))
            assert rc == 0, lib.cpmppi_last_error(None)
            variants.append((os.path.basename(path), math, lib, h))
    # one shared perturbation buffer (sampled by the first variant)
    lib0, h0 = variants[0][2], variants[0][3]
    lib0.cpmppi_sample(h0, E, 1234, 0, 0, kn.data_ptr(), du.data_ptr(), stream)
    tiled = None
    if "tiled" in args.noise:            # (needs a build with the tiled layout: the newest library goes last on the command line)
        libT, hT = variants[-1][2], variants[-1][3]
        libT.cpmppi_tiled_floats.argtypes = [vp, u32]
        libT.cpmppi_tiled_floats.restype = C.c_size_t
        libT.cpmppi_sample_tiled.argtypes = [vp, u32, u64, u64, u32, vp, vp, vp]
        tiled = torch.empty(int(libT.cpmppi_tiled_floats(hT, E)), device=dev)
        assert libT.cpmppi_sample_tiled(hT, E, 1234, 0, 0, None, tiled.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    res = {}
    for rnd in range(args.rounds):
        for name, math, lib, h in variants:
            for noise in args.noise:
                a = L.cpmppi_step_args()
                a.E = E
                a.s0, a.u_nom, a.target_position, a.target_equilibrium = s0.data_ptr(), u_nom.data_ptr(), tp.data_ptr(), te.data_ptr()
                a.L, a.Q_out = Lt.data_ptr(), Q.data_ptr()
                if noise == "tiled" and not hasattr(lib, "cpmppi_sample_tiled"):
                    continue
                a.noise_kind = {"buffer": 0, "knots": 1, "philox": 2, "tiled": 3}[noise]
                a.noise = {"buffer": du.data_ptr(), "knots": kn.data_ptr(), "philox": None,
                           "tiled": tiled.data_ptr() if tiled is not None else None}[noise]
                a.seed, a.offset = 1234, 0
                u_nom.zero_()
                lib.cpmppi_set_profiling(h, 1)
                for it in range(args.steps):
                    a.offset = it                 # (a different noise draw per step: the same ones for every variant)
                    rc = lib.cpmppi_step(h, C.byref(a), stream)
                    assert rc == 0, lib.cpmppi_last_error(h)
                ra = (f * 64)(); rb = (f * 64)(); n = u32(0)
                lib.cpmppi_get_profile(h, ra, rb, 64, C.byref(n))
                lib.cpmppi_set_profiling(h, 0)
                if rnd > 0:
                    res.setdefault((name, math, noise), []).extend(list(ra[:n.value]))
                res.setdefault(("chk", name, math, noise), float(u_nom.abs().sum()))
    print(f"E={E} N={N} H={H}: rollout_cost_kernel ms per launch (median / min / mean / p90), rollouts/s at median")
    for k, v in res.items():
        if k[0] == "chk":
            continue
        med, mn = float(np.median(v)), float(np.min(v))
        print(f"  {k[0]:28s} {k[1]:8s} {k[2]:7s} median {med:8.4f}  min {mn:8.4f}  mean {float(np.mean(v)):8.4f}  p90 {float(np.percentile(v, 90)):8.4f}  {E * N / med * 1e3:.3e} rollouts/s   "
              f"checksum {res[('chk',) + k]:.6f}")


if __name__ == "__main__":
    main()
