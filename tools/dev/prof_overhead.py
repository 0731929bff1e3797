"""Development tool (GPU): what do the per-launch HIP events of `set_profiling` cost the small configurations' wall time, and does a
preceding heavy load (the headline workload, as in a default bench run) change the small launches' time?
python tools/dev/prof_overhead.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B  # noqa: E402

ctx = dict(rank=0, world=1, local_rank=0, device=torch.device("cuda", 0), backend="nccl", collective=False)


def timed(gw, steps, warm, prof):
    gw.u_all.zero_()
    torch.cuda.synchronize()
    gw.groups.fork()
    gw.groups.run(gw.step_all, None, periods=warm, offset=0)
    torch.cuda.synchronize()
    for w in gw.parts:
        w.eng.set_profiling(prof, group=8)
    t0 = time.perf_counter()
    gw.groups.run(gw.step_all, None, periods=steps, offset=warm)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for w in gw.parts:
        w.eng.set_profiling(False)
    return 1e6 * dt / steps


def heavy(seconds):
    w = B.Workload(ctx, 8192, 1024, 50)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        w.run(10, 0)
    w.close()


for name, (E, N, H), steps in (("C4", (64, 2048, 50), 200), ("C3", (64, 4096, 100), 100)):
    for G in (1, 2):
        gw = B.GroupedWorkload(ctx, E, N, H, G)
        gw.groups.fork()
        res = {}
        for label, pre in (("cold", 0.0), ("after 3 s of the headline workload", 3.0)):
            if pre:
                heavy(pre)
            for prof in (False, True):
                res[(label, prof)] = [timed(gw, steps, 20, prof) for _ in range(5)]
        for (label, prof), v in res.items():
            print(f"{name} groups {G} {label:36s} events {'on ' if prof else 'off'}: median {np.median(v):7.2f} us  min {np.min(v):7.2f}  max {np.max(v):7.2f}")
        gw.close()
