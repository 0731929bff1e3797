"""TEST INFRASTRUCTURE: ONE RANK of the two-process tests of the per-step all-gather (tests/test_gpu_two_rank_gather.py).

Every rank is a fresh interpreter on device 0 with its own block of envs; libcpmppi's communicator binds tests/fake_rccl/
libfake_rccl.so through the `rccl_path` argument (the product code path is the production one - cpmppi_comm_init,
cpmppi_step_gather / cpmppi_groups_run_gather, the side stream, the device-memory ordering - only the collective library behind it is
the stand-in).  The 128-byte id travels through a file the test names.  Results go to an .npz the test reads.

  mode steps   K cpmppi_step_gather calls (two alternating buffers, stamped blocks), no host sync in between
  mode stall   the same, the launch stream paced by the host; rank `--stall-rank` joins gather `--stall-gather` 60 ms late while the
               other ranks run with a 2 ms timeout: a timed-out rank drops steps, gets CPMPPI_ERR_COMM from the NEXT call, clears it
               with cpmppi_comm_sync and goes on (re-issuing the refused call, so that the ranks' collectives stay matched)
  mode groups  env groups under one communicator: K1 periods one library call each, then K2 periods in ONE cpmppi_groups_run_gather
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
FAKE = os.path.join(HERE, "libfake_rccl.so")


def inputs(E_total, seed):
    """Per-env inputs of ALL envs (every rank slices its block): the bench's synthetic distribution."""
    rng = np.random.Generator(np.random.SFC64(seed))
    THL = 0.198
    angle = np.where(rng.uniform(size=E_total) > 0.5, 1.0, -1.0) * rng.uniform(0.0, 180.0, E_total) * np.pi / 180.0
    s0 = np.zeros((E_total, 6), dtype=np.float32)
    s0[:, 0] = angle
    s0[:, 1] = rng.uniform(-1, 1, E_total) * 300.0 * np.pi / 180.0
    s0[:, 2], s0[:, 3] = np.cos(angle), np.sin(angle)
    s0[:, 4] = rng.uniform(-1, 1, E_total) * THL * 0.8
    s0[:, 5] = rng.uniform(-1, 1, E_total) * THL * 0.5
    tp = (rng.uniform(-0.8, 0.8, E_total) * THL).astype(np.float32)
    te = np.ones(E_total, dtype=np.float32)
    L = rng.uniform(0.2, 0.5, E_total).astype(np.float32)
    return s0, tp, te, L


def get_id(lib, path, rank):
    """Rank 0 draws the id (through the library, from the stand-in) and publishes it in a file; the others poll for it."""
    from cartpolesimulation_amd import _lib as _L
    if rank == 0:
        buf = C.create_string_buffer(_L.COMM_ID_BYTES)
        rc = lib.cpmppi_comm_unique_id(buf, FAKE.encode())
        assert rc == 0, lib.cpmppi_last_error(None)
        with open(path + ".tmp", "wb") as f:
            f.write(buf.raw)
        os.rename(path + ".tmp", path)
        return buf.raw
    t0 = time.time()
    while not os.path.exists(path):
        assert time.time() - t0 < 120, "rank 0 never published the id"
        time.sleep(0.005)
    return open(path, "rb").read()


def file_barrier(base, rank, world, tag):
    open(f"{base}.{tag}.{rank}", "w").close()
    t0 = time.time()
    while not all(os.path.exists(f"{base}.{tag}.{r}") for r in range(world)):
        assert time.time() - t0 < 120, f"barrier {tag}: a rank is missing"
        time.sleep(0.002)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True, choices=["steps", "stall", "groups"])
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--envs", type=int, default=4, help="envs per rank")
    ap.add_argument("--rollouts", type=int, default=512)
    ap.add_argument("--horizon", type=int, default=30)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=20, help="groups: periods of the single cpmppi_groups_run_gather call that follows")
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--seed", type=int, default=77)
    ap.add_argument("--stall-rank", type=int, default=1)
    ap.add_argument("--stall-gather", type=int, default=20)
    ap.add_argument("--timeout-ms", type=float, default=2.0)
    ap.add_argument("--slow-collective-us", type=int, default=0, help="cpmppi_debug_comm_delay in front of every all-gather")
    ap.add_argument("--jitter", default="", help="stress: '<per mille>:<max us>' - a random subset of this rank's gathers joins late by a random time "
                                                 "(inside the stand-in collective), and the host naps at random between calls")
    ap.add_argument("--base", required=True, help="path prefix for the id file, the barrier files and the result")
    a = ap.parse_args()
    if a.mode == "stall" and a.rank == a.stall_rank:
        os.environ["FAKE_RCCL_DELAY_US"] = f"{a.stall_gather}:60000"
    naps = None
    if a.jitter:
        pm, us = (int(x) for x in a.jitter.split(":"))
        os.environ["FAKE_RCCL_DELAY_US"] = f"rand:{pm}:{us}:{a.seed + 17 * a.rank}"
        naps = np.random.Generator(np.random.SFC64(a.seed + 1000 * a.rank))

    import torch
    from cartpolesimulation_amd import _lib as _L
    from cartpolesimulation_amd.configs import MPPIConfig
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.pipeline import EnvGroups
    from cartpolesimulation_amd.shard import NativeGather
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib = _L.load()
    E, N, H, K, W, rank = a.envs, a.rollouts, a.horizon, a.steps, a.world, a.rank
    n, pad = E * H, _L.GATHER_STAMP_FLOATS
    s0_all, tp_all, te_all, L_all = inputs(E * W, a.seed)
    sl = slice(rank * E, (rank + 1) * E)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=dev)          # noqa: E731
    s0, tp, te, Lv = t(s0_all[sl]), t(tp_all[sl]), t(te_all[sl]), t(L_all[sl])
    mppi = MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=1)
    uid = get_id(lib, a.base + ".id", rank)
    out = dict(rank=rank, errors=[], info=None)

    if a.mode in ("steps", "stall"):
        eng = MPPIEngine(E, mppi, device=0)
        g = NativeGather(eng, uid, W, rank, rccl_path=FAKE, stamped=True)
        out["info"] = g.info()
        if a.mode == "stall" and rank != a.stall_rank:
            eng._check(lib.cpmppi_comm_set_timeout(eng._h, a.timeout_ms * 1e-3))
        if a.slow_collective_us:
            eng._check(lib.cpmppi_debug_comm_delay(eng._h, a.slow_collective_us))
        log = torch.zeros(K, W, n + pad, dtype=torch.float32, device=dev)      # one receive buffer per step: nothing is overwritten
        Q = eng.empty(E)
        prep = [eng.prepare_step(s0, g.u[b], tp, te, L=Lv, seed=a.seed, offset=0, env_offset=rank * E, Q_out=Q, u_nom_out=g.u[1 - b])
                for b in range(2)]
        file_barrier(a.base, rank, W, "ready")
        i, refused = 0, []
        t0 = time.perf_counter()
        while i < K:
            try:
                prep[i & 1].run(offset=i, gather_into=log[i])
            except _L.CpmppiError as e:
                # the call was refused BEFORE anything was enqueued: report, clear (cpmppi_comm_sync reports once more and clears),
                # and re-issue it - the peers are inside the same sequence of collectives
                assert e.code == -6 and "timed out" in str(e), str(e)
                refused.append(i)
                try:
                    g.sync()
                    out["errors"].append("sync returned OK after a refused call")
                except _L.CpmppiError as e2:
                    assert "timed out" in str(e2), str(e2)
                eng._check(lib.cpmppi_comm_set_timeout(eng._h, 10.0))           # (the stall is over: the rest of the run with the default)
                continue
            if a.mode == "stall":
                torch.cuda.current_stream(dev).synchronize()                    # the launch stream only: the host keeps pace with the steps
            if naps is not None and naps.uniform() < 0.05:
                time.sleep(float(naps.uniform(0.0, 2e-3)))                      # the ranks drift apart and catch up again
            i += 1
        torch.cuda.synchronize()
        try:
            g.sync()
            final_sync = "ok"
        except _L.CpmppiError as e:
            final_sync = str(e)
            try:
                g.sync()
            except _L.CpmppiError:
                pass
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        file_barrier(a.base, rank, W, "done")                                   # nobody tears its mailbox down while a peer still reads it
        np.savez(a.base + f".rank{rank}.npz", log=log.cpu().numpy(), refused=np.array(refused, dtype=np.int64),
                 u_final=g.u[K & 1].cpu().numpy(), us_per_step=1e6 * dt / K)
        out.update(refused=refused, final_sync=final_sync, us_per_step=round(1e6 * dt / K, 1))
        g.close()
        eng.close()
    else:
        grp = EnvGroups(E, mppi, groups=a.groups, device=0, env_offset=rank * E)
        grp.comm_init(uid, W, rank, rccl_path=FAKE, stamped=True)
        out["info"] = grp.comm_info()
        if a.slow_collective_us:
            h0 = C.c_void_p(lib.cpmppi_groups_handle(grp._g, 0))
            assert lib.cpmppi_debug_comm_delay(h0, a.slow_collective_us) == 0
        flat = [torch.zeros(n + pad, dtype=torch.float32, device=dev) for _ in range(2)]
        u = [f[:n].view(E, H) for f in flat]
        K1, K2 = K, a.batch
        log = torch.zeros(K1 + 1, W, n + pad, dtype=torch.float32, device=dev)
        Q = torch.empty(E, dtype=torch.float32, device=dev)
        # one argument block per parity for the one-period calls; the batch call alternates inside the library
        prep = [grp.prepare(s0, u[b], tp, te, L=Lv, seed=a.seed, Q_out=Q, u_nom_out=u[1 - b]) for b in range(2)]
        file_barrier(a.base, rank, W, "ready")
        grp.fork()
        t0 = time.perf_counter()
        for i in range(K1):
            grp.run(prep[i & 1], periods=1, offset=i, gather_into=log[i])
            if naps is not None and naps.uniform() < 0.05:
                time.sleep(float(naps.uniform(0.0, 2e-3)))
        grp.run(prep[K1 & 1], periods=K2, offset=K1, gather_into=log[K1])
        grp.join()
        torch.cuda.synchronize()
        grp.comm_sync()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        file_barrier(a.base, rank, W, "done")
        np.savez(a.base + f".rank{rank}.npz", log=log.cpu().numpy(), u_final=u[(K1 + K2) & 1].cpu().numpy(), us_per_step=1e6 * dt / (K1 + K2))
        out.update(us_per_step=round(1e6 * dt / (K1 + K2), 1), slices=grp.slices)
        grp.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
