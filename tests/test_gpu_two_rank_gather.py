"""GPU: the path's ONE collective meets a PEER.  Two processes on ONE device, each a rank with its own block of
envs, run the production protocol of csrc/cpmppi_comm.hip - cpmppi_comm_init, cpmppi_step_gather / cpmppi_groups_run_gather, the
side stream, the ordering through device memory - against each other.  RCCL refuses two ranks on one device and the pool hands out
one GPU per box, so the collective library behind the communicator is the test double tests/fake_rccl/libfake_rccl.so (an intra-node
all-gather over IPC-mapped device memory with device-side flags, handed to the library through the existing `rccl_path` argument);
everything in front of ncclAllGather is the shipped code.

The bar (the reference's fan-out is share-nothing job arrays, others/EulerClusterScripts/ParallelDataGeneration.sh:2-17):
sharding changes no number - every gathered block equals what a SINGLE process computes for all envs (global-index Philox keys) -
and a stalled peer costs the waiting rank a dropped step that everybody can see, never a corrupted or silently stale block."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests", "fake_rccl")
sys.path.insert(0, HERE)


def _ranks(tmp_path, mode, world=2, timeout=420, **kw):
    base = str(tmp_path / "run")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["FAKE_RCCL_DIR"] = str(tmp_path)
    args = [x for k, v in kw.items() for x in ("--" + k.replace("_", "-"), str(v))]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "worker.py"), "--mode", mode, "--rank", str(r), "--world", str(world),
                               "--base", base] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = []
    for r, p in enumerate(procs):
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            so, se = p.communicate()
            pytest.fail(f"rank {r} timed out\n{se[-2000:]}")
        assert p.returncode == 0, f"rank {r}: rc {p.returncode}\n{se[-3000:]}"
        line = [l for l in so.splitlines() if l.startswith("{")][-1]
        outs.append(json.loads(line))
    data = [np.load(base + f".rank{r}.npz") for r in range(world)]
    return outs, data


def _single_process(E_total, N, H, K, seed):
    """The same envs in ONE process, one handle, no collective: u_nom[E_total, H] after every step."""
    import torch
    from worker import inputs
    from cartpolesimulation_amd.configs import MPPIConfig
    from cartpolesimulation_amd.engine import MPPIEngine
    dev = torch.device("cuda", 0)
    s0, tp, te, L = (torch.as_tensor(x, device=dev) for x in inputs(E_total, seed))
    eng = MPPIEngine(E_total, MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=1), device=0)
    u, Q = eng.zeros(E_total, H), eng.empty(E_total)
    prep = eng.prepare_step(s0, u, tp, te, L=L, seed=seed, offset=0, env_offset=0, Q_out=Q)
    hist = []
    for i in range(K):
        prep.run(offset=i)
        hist.append(u.clone())
    torch.cuda.synchronize()
    out = torch.stack(hist).cpu().numpy()
    eng.close()
    return out


def _stamps(log, n):
    return log[:, :, n].view(np.uint32).astype(np.int64)


def test_three_ranks_block_order_and_contents(tmp_path):
    """Three processes on one device: block r of every gather is rank r's (the all-gather's block order), every block bit-equal to
    the single-process run of all 3 * E_local envs, on every rank."""
    E, N, H, K, seed, W = 3, 512, 20, 24, 81, 3
    outs, data = _ranks(tmp_path, "steps", world=W, envs=E, rollouts=N, horizon=H, steps=K, seed=seed)
    ref = _single_process(W * E, N, H, K, seed)
    n = E * H
    for r in range(W):
        assert outs[r]["info"]["rccl_ranks"] == W and outs[r]["info"]["rccl_rank"] == r and outs[r]["refused"] == []
        log = data[r]["log"]
        assert log.shape == (K, W, n + 4)
        for p in range(W):
            assert np.array_equal(log[:, p, :n].reshape(K, E, H), ref[:, p * E:(p + 1) * E]), (r, p)
        assert np.array_equal(_stamps(log, n), np.repeat(np.arange(1, K + 1)[:, None], W, axis=1))
    assert np.array_equal(data[0]["log"], data[1]["log"]) and np.array_equal(data[0]["log"], data[2]["log"])


@pytest.mark.parametrize("slow_us", [0, 400])
def test_two_ranks_gather_every_step_equal_to_the_single_process_run(tmp_path, slow_us):
    """50 cpmppi_step_gather steps per rank, no host sync in between, alternating buffers: every block of every gather, on both
    ranks, is bit for bit what ONE process computes for all 2 * E_local envs; every block carries its step's stamp.  Second case: a
    400 us spin in front of every all-gather (the collective finishes deep inside the NEXT step's kernel)."""
    E, N, H, K, seed = 4, 512, 30, 50, 77
    outs, data = _ranks(tmp_path, "steps", envs=E, rollouts=N, horizon=H, steps=K, seed=seed, slow_collective_us=slow_us)
    ref = _single_process(2 * E, N, H, K, seed)                                  # [K, 2E, H]
    n = E * H
    for r in range(2):
        info = outs[r]["info"]
        assert info["rccl_ranks"] == 2 and info["rccl_rank"] == r and info["rccl_version"] == 99 and info["stamped"] == 1
        assert outs[r]["refused"] == [] and outs[r]["final_sync"] == "ok" and outs[r]["errors"] == []
        log = data[r]["log"]                                                     # [K, 2, n + 4]
        assert log.shape == (K, 2, n + 4)
        for p in range(2):
            got = log[:, p, :n].reshape(K, E, H)
            assert np.array_equal(got, ref[:, p * E:(p + 1) * E]), f"rank {r}: block of rank {p} differs from the single-process run"
        assert np.array_equal(_stamps(log, n), np.repeat(np.arange(1, K + 1)[:, None], 2, axis=1))
        assert np.all(log[:, :, n + 1:] == 0)                                    # reserved words
        assert np.array_equal(data[r]["u_final"], ref[-1, r * E:(r + 1) * E])
    assert np.array_equal(data[0]["log"], data[1]["log"])                       # both ranks received the same thing
    assert np.abs(ref).max() > 1e-2 and not np.array_equal(ref[:, :E], ref[:, E:])


def test_a_stalled_peer_drops_steps_visibly_and_everybody_recovers(tmp_path):
    """Rank 1 joins gather 20 sixty milliseconds late; rank 0 runs with a 2 ms timeout.  Rank 0's finalize, about to overwrite the
    buffer the late gather still reads, gives up: the step is DROPPED (buffer intact), the next cpmppi_step_gather returns
    CPMPPI_ERR_COMM, cpmppi_comm_sync reports and clears it, the run goes on.  The PEER sees it too: the blocks rank 0 sent while it
    was dropping steps carry an OLD stamp (cpmppi_comm_set_stamped) - on both ranks the same set of gathers; every block whose
    content shows a dropped step (the buffer left intact) is among them; rank 1 itself never errs."""
    E, N, H, K, seed, stall = 4, 512, 30, 60, 78, 20
    outs, data = _ranks(tmp_path, "stall", envs=E, rollouts=N, horizon=H, steps=K, seed=seed, stall_rank=1, stall_gather=stall,
                        timeout_ms=2.0)
    n = E * H
    assert outs[1]["refused"] == [] and outs[1]["final_sync"] == "ok"            # the late rank has the default timeout: no drop
    assert len(outs[0]["refused"]) >= 1 and outs[0]["errors"] == [], outs[0]     # refused -> sync reported the timeout -> cleared
    assert outs[0]["final_sync"] == "ok", outs[0]
    first_refused = outs[0]["refused"][0]
    assert stall <= first_refused <= stall + 8, outs[0]
    st0, st1 = _stamps(data[0]["log"], n), _stamps(data[1]["log"], n)
    assert np.array_equal(st0, st1)                                              # what rank 1 can know = what rank 0 knows
    want = np.arange(1, K + 1)
    assert np.array_equal(st0[:, 1], want)                                       # rank 1's blocks: all fresh
    stale = np.nonzero(st0[:, 0] != want)[0]
    assert len(stale) >= 1 and stale.min() >= stall - 1 and stale.max() < first_refused + 2, (stale, outs[0])
    assert np.all(st0[stale, 0] < want[stale])                                   # an OLD stamp, never a future one
    log = data[1]["log"]                                                         # the PEER's view
    # the steps rank 0 DROPPED: its block is bit for bit the block of two steps earlier (the buffer was left intact; a step that
    # ran changes every element).  Every one of them must be among the stale stamps - the peer rejects all of them (it may reject
    # a good block gathered while the error was up as well: the stamp errs on the safe side)
    dropped = np.array([i for i in range(2, K) if np.array_equal(log[i, 0, :n], log[i - 2, 0, :n])])
    assert len(dropped) >= 1 and dropped.min() == stall + 1, (dropped, stale)    # gather 20 = step 19 is late: step 21 cannot store
    assert set(dropped.tolist()) <= set(stale.tolist()), (dropped, stale)
    assert len(stale) <= len(dropped) + 2, (dropped, stale)
    for i in stale:                                                              # a stale stamp = the number of the last stamped step of that buffer
        j = i - 2
        while j in stale:
            j -= 2
        assert j >= 0 and st1[i, 0] == j + 1
    fresh = np.setdiff1d(np.arange(K), stale)
    assert len(fresh) >= K - 10 and fresh.max() == K - 1                         # recovered: the tail of the run is fresh again
    # rank 1's own sequences are untouched by rank 0's trouble: equal to the single-process run of its envs
    ref = _single_process(2 * E, N, H, K, seed)
    assert np.array_equal(log[:, 1, :n].reshape(K, E, H), ref[:, E:])
    # rank 0 before the stall = the single-process run too; after the drop it lags (it skipped updates) but stays finite
    assert np.array_equal(log[:dropped.min(), 0, :n].reshape(-1, E, H), ref[:dropped.min(), :E])
    assert np.isfinite(log).all()


@pytest.mark.parametrize("slow_us", [0, 400])
def test_env_groups_under_one_communicator_two_ranks(tmp_path, slow_us):
    """cpmppi_groups_run_gather: every rank's envs as TWO env groups on their own streams, ONE communicator and side stream per
    device, one all-gather of the device's whole u_nom per period - 30 periods one call each, then 20 periods in one call.  Every
    gathered block equals the single-process, single-handle run of all envs (the split and the sharding change no number)."""
    E, N, H, K1, K2, seed = 6, 512, 30, 30, 20, 79
    outs, data = _ranks(tmp_path, "groups", envs=E, rollouts=N, horizon=H, steps=K1, batch=K2, groups=2, seed=seed,
                        slow_collective_us=slow_us)
    ref = _single_process(2 * E, N, H, K1 + K2, seed)
    n = E * H
    for r in range(2):
        assert outs[r]["info"]["rccl_ranks"] == 2 and outs[r]["info"]["stamped"] == 1 and outs[r]["slices"] == [[0, 3], [3, 6]]
        log = data[r]["log"]                                                     # [K1 + 1, 2, n + 4]; the last row = after the batch call
        for p in range(2):
            assert np.array_equal(log[:K1, p, :n].reshape(K1, E, H), ref[:K1, p * E:(p + 1) * E])
            assert np.array_equal(log[K1, p, :n].reshape(E, H), ref[-1, p * E:(p + 1) * E])
        st = _stamps(log, n)
        assert np.array_equal(st[:K1], np.repeat(np.arange(1, K1 + 1)[:, None], 2, axis=1)) and np.all(st[K1] == K1 + K2)
        assert np.array_equal(data[r]["u_final"], ref[-1, r * E:(r + 1) * E])
    assert np.array_equal(data[0]["log"], data[1]["log"])


@pytest.mark.parametrize("mode", ["steps", "groups"])
def test_two_ranks_with_random_stalls_on_both_sides(tmp_path, mode):
    """Stress: 400 steps per rank with a fifth of the all-gathers joining late by up to 300 us on EITHER rank (inside the stand-in
    collective) and host naps of up to 2 ms now and then, default timeout (nothing may be dropped): the ranks drift apart by several
    steps and catch up again, the finalizes wait for real - and still every gathered block on both ranks is bit for bit the
    single-process run, every block carries its step's stamp, nobody errs.  One handle per rank, and env groups under one communicator."""
    E, N, H = 4, 256, 16
    # (a soak run: CPMPPI_TWO_RANK_STRESS="<steps>:<per mille>:<max us>:<seed>", e.g. 6000:300:1500:7 - tools/dev/r6_soak_two_rank.sh)
    steps, pm, us, seed = (int(x) for x in os.environ.get("CPMPPI_TWO_RANK_STRESS", "400:200:300:83").split(":"))
    K, KB = (steps, 0) if mode == "steps" else (steps - steps // 4, steps // 4)
    kw = dict(envs=E, rollouts=N, horizon=H, steps=K, seed=seed, jitter=f"{pm}:{us}")
    if mode == "groups":
        kw.update(batch=KB, groups=2)
    outs, data = _ranks(tmp_path, mode, **kw)
    ref = _single_process(2 * E, N, H, K + KB, seed)
    n = E * H
    for r in range(2):
        assert outs[r]["errors"] == [] and outs[r].get("refused", []) == [] and outs[r].get("final_sync", "ok") == "ok"
        log = data[r]["log"]
        for p in range(2):
            assert np.array_equal(log[:K, p, :n].reshape(K, E, H), ref[:K, p * E:(p + 1) * E]), (r, p)
        st = _stamps(log, n)
        assert np.array_equal(st[:K], np.repeat(np.arange(1, K + 1)[:, None], 2, axis=1))
        if mode == "groups":
            assert np.all(st[K] == K + KB) and np.array_equal(log[K, :, :n].reshape(2 * E, H), ref[-1])
        assert np.array_equal(data[r]["u_final"], ref[-1, r * E:(r + 1) * E])
    assert np.array_equal(data[0]["log"], data[1]["log"])
