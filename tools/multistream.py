#!/usr/bin/env python3
"""Small launches: do independent env GROUPS on their own streams fill the SIMDs one launch leaves idle behind its slowest wave?

Development tool (task 2a).  The E envs of a small configuration (C4: 64 x 2048 x 50, C3: 64 x 4096 x 100) are split
into G contiguous groups, each with its own handle (a handle serves one stream) and its own stream; every group runs its chain of
K steps back to back (step i + 1 of a group depends on step i of that group only; `env_offset` keeps the Philox keys of the
unsplit launch, so the results are bit-identical - checked).  Reported: wall time per step of ALL envs (host timer around
enqueue + synchronize, K steps), for G = 1, 2, 4, 8 and for forced lane mappings.

Usage: python tools/multistream.py [--envs 64 --rollouts 2048 --horizon 50] [--groups 1 2 4] [--rpl 0 1 2] [--steps 200] [--rounds 7]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_inputs  # noqa: E402
from cartpolesimulation_amd.configs import MPPIConfig  # noqa: E402
from cartpolesimulation_amd.engine import MPPIEngine  # noqa: E402


class Groups:
    """pipeline.EnvGroups (dedicated-queue streams, prepared argument blocks) over the synthetic inputs."""

    def __init__(self, E, N, H, G, rpl, s0, tp, te, Lt, seed):
        from cartpolesimulation_amd.pipeline import EnvGroups
        self.g = EnvGroups(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=rpl), G)
        self.u = torch.zeros(E, H, device=s0.device)
        self.Q = torch.empty(E, device=s0.device)
        self.preps = self.g.prepare_step(s0, self.u, tp, te, L=Lt, seed=seed, Q_out=self.Q)
        self.g.fork()
        self.eng = self.g.engines[0]


def run(groups, K, offset0=0):
    """K steps of every group, enqueued round-robin; returns the wall time per step of all envs (seconds)."""
    groups.u.zero_()
    torch.cuda.synchronize()
    groups.g.fork()
    t0 = time.perf_counter()
    for i in range(K):
        for p in groups.preps:
            p.run(offset=offset0 + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--rollouts", type=int, default=2048)
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--groups", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--rpl", type=int, nargs="+", default=[0, 1, 2])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    E, N, H = args.envs, args.rollouts, args.horizon
    dev = torch.device("cuda", 0)
    s0, tp, te, Lt = synthetic_inputs(E, H, 3, dev)
    res, sums = {}, {}
    for G in args.groups:                                      # (one set of groups alive at a time: every stream owns a hardware queue)
        for rpl in args.rpl:
            if G > E:
                continue
            groups = Groups(E, N, H, G, rpl, s0, tp, te, Lt, 1234)
            run(groups, 3)                                     # checksum of the nominal sequences after 3 steps: the split changes nothing
            info = groups.eng.last_launch()
            sums[(G, rpl)] = (groups.u.double().abs().sum().item(), info["kernel"], info["blocks"])
            res[(G, rpl)] = [run(groups, args.steps) * 1e6 for _ in range(args.rounds)][1:]
            groups.g.close()
    print(f"E={E} N={N} H={H}: wall time per step of all {E} envs, us (median / min over {args.rounds - 1} rounds of {args.steps} steps)")
    out = []
    for (G, rpl), v in res.items():
        chk, kern, blocks = sums[(G, rpl)]
        print(f"  groups {G}  rpl {rpl}  {kern:48s} blocks/launch {blocks:5d}  median {np.median(v):8.2f}  min {np.min(v):8.2f}   checksum {chk:.9f}")
        out.append(dict(groups=G, rpl=rpl, kernel=kern, blocks=blocks, median_us=float(np.median(v)), min_us=float(np.min(v)), checksum=chk))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(dict(E=E, N=N, H=H, steps=args.steps, rounds=args.rounds, results=out), f, indent=1)


if __name__ == "__main__":
    main()
