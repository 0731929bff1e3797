#!/usr/bin/env python3
"""Development tool (GPU, one rank on real RCCL): what does the per-step all-gather cost the GROUPED form of a small configuration?
Wall time per step of cpmppi_groups_run (no collective) against cpmppi_groups_run_gather with / without stamped blocks, two
alternating buffers / in place, several repetitions each in one process.

  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29640 python tools/dev/groups_gather_cost.py [--config C4] [--groups 2]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import PRESETS, synthetic_inputs  # noqa: E402
from cartpolesimulation_amd import _lib as L  # noqa: E402
from cartpolesimulation_amd.configs import MPPIConfig  # noqa: E402
from cartpolesimulation_amd.pipeline import EnvGroups  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C4")
ap.add_argument("--groups", type=int, default=2)
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--rpl", type=int, default=0, help="rollouts per lane (0 = the library's size rule)")
ap.add_argument("--overlap", action="store_true", help="also print EnvGroups.overlap (sum of the groups' times alone / time together), as bench.py's stream_overlap")
ap.add_argument("--only", default=None, choices=[None, "none", "gather", "gather-stamped", "inplace", "inplace-stamped"], help="one case only (for a kernel trace)")
args = ap.parse_args()
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29640")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
E, N, H = PRESETS[args.config]
s0, tp, te, Lt = synthetic_inputs(E, H, 2, dev)
cfg = MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=args.rpl)
n, pad = E * H, L.GATHER_STAMP_FLOATS


ENQ = []


def timed(g, prep, recv, K):
    g.fork()
    for _ in range(2):                      # warm, then timed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if recv is None:
            g.run(prep[0], None, periods=K, offset=0)
        else:
            g.run(prep[0], None, periods=K, offset=0, gather_into=recv)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        if recv is not None:
            g.comm_sync()
        dt = time.perf_counter() - t0
    ENQ.append(1e6 * t_enq / K)
    return 1e6 * dt / K


def case(name, collective, stamped=False, alternate=True):
    ts = []
    for _ in range(args.reps):
        g = EnvGroups(E, cfg, args.groups)
        flat = [torch.zeros(n + pad, device=dev) for _ in range(2)]
        u = [f[:n].view(E, H) for f in flat]
        recv = None
        if collective:
            uid = C.create_string_buffer(L.COMM_ID_BYTES)
            assert g.lib.cpmppi_comm_unique_id(uid, None) == 0
            g.comm_init(uid.raw, 1, 0, stamped=stamped)
            recv = torch.zeros(1, n + pad, device=dev)
        kw = dict(u_nom_out=u[1]) if (collective and alternate) else {}
        prep = [g.prepare(s0, u[0], tp, te, L=Lt, seed=1234, **kw)]
        ts.append(timed(g, prep, recv, args.steps))
        if args.overlap and not collective:
            g.fork()
            print(f"stream_overlap {g.overlap(prep[0], steps=50):.3f}  (kernel {g.engines[0].last_launch()['kernel']})", flush=True)
        g.close()
    print(f"{name:48s} {np.median(ts):7.2f} us/step  (min {np.min(ts):.2f}, max {np.max(ts):.2f}); the host's enqueueing alone "
          f"{np.median(ENQ[-args.reps:]):6.2f} us/step", flush=True)
    return float(np.median(ts))


print(f"{args.config}: {E} envs x {N} x {H} as {args.groups} groups, {args.steps} steps x {args.reps} repetitions")
CASES = {"gather": ("gather, alternating buffers", dict()), "gather-stamped": ("gather, alternating buffers, stamped", dict(stamped=True)),
         "inplace": ("gather, in place", dict(alternate=False)), "inplace-stamped": ("gather, in place, stamped", dict(stamped=True, alternate=False))}
if args.only == "none":
    case("no collective (cpmppi_groups_run)", False)
elif args.only:
    case(CASES[args.only][0], True, **CASES[args.only][1])
else:
    base = case("no collective (cpmppi_groups_run)", False)
    for nm, kw in CASES.values():
        t = case(nm, True, **kw)
        print(f"{'':48s} = {t / base:.3f} x without")
    case("no collective again", False)
dist.destroy_process_group()
