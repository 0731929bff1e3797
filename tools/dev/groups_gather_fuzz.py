#!/usr/bin/env python3
"""Development tool (GPU, one rank on real RCCL): randomised differential test of cpmppi_groups_run_gather against the plain loop of ONE
handle without any collective - random env counts and group counts (uneven splits, more groups than envs), rollouts, horizons, two
alternating buffers or in place, stamped or not, both forms of the side stream, a random partition of the steps into library calls
(so that the buffer parity is carried across calls), a random slow collective.  After every call the gathered block must be bit for
bit the reference's nominal sequences at that step, and the stamp the number of step-gathers made.

  python tools/dev/groups_gather_fuzz.py --n 40 --seed 1
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
from cartpolesimulation_amd import _lib as L  # noqa: E402
from cartpolesimulation_amd.configs import MPPIConfig  # noqa: E402
from cartpolesimulation_amd.engine import MPPIEngine  # noqa: E402
from cartpolesimulation_amd.pipeline import EnvGroups  # noqa: E402
from cartpolesimulation_amd.shard import block_stamps  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rng = np.random.Generator(np.random.SFC64(args.seed))
dev = torch.device("cuda", 0)
lib = L.load()
failed = []
for case in range(args.n):
    E, G = int(rng.integers(1, 13)), int(rng.integers(1, 6))
    N, H = int(rng.choice([256, 512, 768])), int(rng.integers(5, 31))
    alternate, stamped = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    waiter = str(rng.choice(["kernel", "stream-ops"]))
    delay = int(rng.choice([0, 0, 50, 300]))
    calls = [int(x) for x in rng.integers(1, 6, size=int(rng.integers(2, 7)))]
    K = sum(calls)
    desc = dict(case=case, E=E, groups=G, N=N, H=H, alternate=alternate, stamped=stamped, waiter=waiter, delay_us=delay, calls=calls)
    os.environ["CPMPPI_COMM_WAITER"] = waiter
    s0, tp, te, Lt = synthetic_inputs(E, H, 100 + case, dev)
    cfg = MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=1)
    ref = MPPIEngine(E, cfg)
    u_ref, want = ref.zeros(E, H), []
    for i in range(K):
        ref.step(s0, u_ref, tp, te, L=Lt, seed=5, offset=i, env_offset=3)
        want.append(u_ref.clone())
    ref.close()
    g = EnvGroups(E, cfg, G, env_offset=3)
    uid = C.create_string_buffer(L.COMM_ID_BYTES)                 # (a fresh id per communicator: RCCL's bootstrap root serves one)
    assert lib.cpmppi_comm_unique_id(uid, None) == 0
    g.comm_init(uid.raw, 1, 0, stamped=stamped)
    h0 = C.c_void_p(lib.cpmppi_groups_handle(g._g, 0))
    lib.cpmppi_debug_comm_delay(h0, delay)
    n, pad = E * H, L.GATHER_STAMP_FLOATS
    flat = [torch.zeros(n + pad, device=dev) for _ in range(2)]
    u = [f[:n].view(E, H) for f in flat]
    prep = ([g.prepare(s0, u[b], tp, te, L=Lt, seed=5, u_nom_out=u[1 - b]) for b in range(2)] if alternate
            else [g.prepare(s0, u[0], tp, te, L=Lt, seed=5)] * 2)
    recv = torch.zeros(len(calls), 1, n + pad, device=dev)
    g.fork()
    done = 0
    for c, k in enumerate(calls):
        g.run(prep[done & 1], None, periods=k, offset=done, gather_into=recv[c])
        done += k
    g.join()
    torch.cuda.synchronize()
    g.comm_sync()
    ok, done = True, 0
    for c, k in enumerate(calls):
        done += k
        ok &= bool(torch.equal(recv[c, 0, :n].view(E, H), want[done - 1]))
        if stamped:
            ok &= block_stamps(recv[c], n).tolist() == [done]
    final = u[K & 1] if alternate else u[0]
    ok &= bool(torch.equal(final, want[-1])) and g.comm_info()["gathers_enqueued"] == K and len(g) == min(G, E)
    g.close()
    if not ok:
        failed.append(desc)
        print("FAILED", json.dumps(desc), flush=True)
os.environ.pop("CPMPPI_COMM_WAITER", None)
print(json.dumps({"configurations": args.n, "passed": args.n - len(failed), "failed": len(failed), "seed": args.seed}))
sys.exit(1 if failed else 0)
