"""world_size-2 gloo test (CPU) of the env sharding and the single gather of chosen controls (§8e).

The per-rank compute is injected (a checker-backed step using the numpy oracle): what is tested here is the
partitioning, the rank-independent keying of per-env inputs, and the collective — the HIP step itself is covered by the
-m gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cartpolesimulation_amd.shard import ShardedMPPI, env_shard, gather_controls
from oracle import oracle_np as O

f32 = np.float32
E_TOTAL, N, H = 5, 64, 8            # 5 envs over 2 ranks: uneven shards (3 + 2)


def _inputs():
    rng = np.random.Generator(np.random.SFC64(3))
    s0 = np.stack([O.create_cartpole_state(rng.uniform(-1, 1), rng.uniform(-2, 2), rng.uniform(-0.1, 0.1), 0.0)
                   for _ in range(E_TOTAL)])
    tp = rng.uniform(-0.05, 0.05, E_TOTAL).astype(f32)
    return s0, tp


def _oracle_step_fn(tp_all):
    cfg = O.MPPIConfig(N=N, H=H)

    def fn(s_local, env_offset):
        u, q = [], []
        for i in range(s_local.shape[0]):
            e = env_offset + i                                   # noise keyed by the GLOBAL env index
            du = O.sample_delta_u(np.random.Generator(np.random.SFC64(1000 + e)), N, H, np.float64(cfg.stdev))
            r = O.mppi_step(s_local[i].numpy(), np.zeros(H, f32), du, tp_all[e], f32(1.0), cfg)
            u.append(r["u_new"]); q.append(r["Q"])
        return torch.tensor(np.array(u)), torch.tensor(np.array(q, dtype=f32))
    return fn


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, tp = _inputs()
        sh = ShardedMPPI(E_TOTAL, _oracle_step_fn(tp))
        u_all, q_all = sh.step(sh.local(torch.tensor(s0)))
        ret[rank] = (sh.start, sh.count, u_all.numpy(), q_all.numpy())
        # even shards take the all_gather_into_tensor path
        even = gather_controls(torch.full((2, 3), float(rank)), 4)
        assert even.shape == (4, 3) and even[:2].eq(0).all() and even[2:].eq(1).all()
    finally:
        dist.destroy_process_group()


def test_env_shard_partition():
    for E, W in ((512, 8), (5, 2), (7, 3), (2, 4)):
        blocks = [env_shard(E, W, r) for r in range(W)]
        assert sum(c for _, c in blocks) == E
        assert all(blocks[i][0] + blocks[i][1] == blocks[i + 1][0] for i in range(W - 1))
    assert env_shard(512, 8, 3) == (192, 64)                     # BASELINE config C4: 64 envs per GPU
    with pytest.raises(ValueError):
        env_shard(4, 2, 2)


@pytest.mark.timeout(300)
def test_two_rank_gather_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    s0, tp = _inputs()
    u_ref, q_ref = _oracle_step_fn(tp)(torch.tensor(s0), 0)      # all envs in one process
    assert ret[0][:2] == (0, 3) and ret[1][:2] == (3, 2)
    for r in (0, 1):
        assert np.array_equal(ret[r][2], u_ref.numpy()) and np.array_equal(ret[r][3], q_ref.numpy())


def _id_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cartpolesimulation_amd import _lib
        from cartpolesimulation_amd.shard import exchange_unique_id
        ret[rank] = bytes(exchange_unique_id(_lib.load(), rank, key="test_comm_id"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_unique_id_reaches_every_rank_through_the_store():
    """The bootstrap of the library's own RCCL communicator (cpmppi_comm_unique_id on rank 0, the 128 bytes handed to the
    other ranks through the process group's key-value store): every rank ends up with the same id.  (Creating an id
    needs RCCL but no GPU; the communicator itself is covered on the GPU box with one rank.)"""
    import ctypes as C
    from cartpolesimulation_amd import _lib
    probe = C.create_string_buffer(_lib.COMM_ID_BYTES)
    if _lib.load().cpmppi_comm_unique_id(probe, None) != 0:
        pytest.skip("RCCL not available on this host: " + _lib.load().cpmppi_last_error(None).decode())
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_id_worker, args=(3, port, ret), nprocs=3, join=True)
    assert len(ret[0]) == _lib.COMM_ID_BYTES and ret[0] == ret[1] == ret[2] and ret[0] != probe.raw


def test_stamped_blocks_receivers_rule():
    """shard.block_stamps / accepted_blocks / merge_accepted on CPU tensors: a block is used iff the word behind its sequences is
    the number of the gather it arrived with (cpmppi_comm_set_stamped, include/cpmppi.h); a rejected rank keeps its previous rows."""
    import torch
    from cartpolesimulation_amd import _lib as L
    from cartpolesimulation_amd.shard import accepted_blocks, block_stamps, merge_accepted
    W, n, pad = 3, 10, L.GATHER_STAMP_FLOATS
    prev = torch.arange(W * n, dtype=torch.float32).view(W, n)
    got = torch.zeros(W, n + pad)
    got[:, :n] = 100.0 + prev
    stamps = torch.tensor([7, 5, 7], dtype=torch.int32)                       # rank 1 dropped steps 6 and 7: its block is stale
    got[:, n] = stamps.view(torch.float32)
    assert block_stamps(got, n).tolist() == [7, 5, 7]
    assert accepted_blocks(got, n, 7).tolist() == [True, False, True]
    merged, ok = merge_accepted(prev, got, n, 7)
    assert ok.tolist() == [True, False, True]
    assert torch.equal(merged[0], got[0, :n]) and torch.equal(merged[2], got[2, :n]) and torch.equal(merged[1], prev[1])
    # a stamp is an integer bit pattern: 2**31 - 1 survives the float32 container, NaN patterns included
    got[0, n] = torch.tensor([0x7FC00001], dtype=torch.int32).view(torch.float32)
    assert int(block_stamps(got, n)[0]) == 0x7FC00001
