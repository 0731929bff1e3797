"""GPU: env groups on their own streams (cartpolesimulation_amd/pipeline.py) - the split never changes a result that the lane
mapping does not change, and the data generator's device loop gives the same recording whether it runs as one chain or as groups."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

f32 = np.float32


def _inputs(E, H, seed=3):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synthetic_inputs
    return synthetic_inputs(E, H, seed, torch.device("cuda", 0))


def test_split_envs_is_contiguous_and_even():
    from cartpolesimulation_amd.pipeline import split_envs
    assert split_envs(64, 2) == [(0, 32), (32, 64)] and split_envs(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert split_envs(3, 8) == [(0, 1), (1, 2), (2, 3)] and split_envs(5, 1) == [(0, 5)]


@pytest.mark.parametrize("E,N,H,groups,rpl", [(64, 2048, 50, 2, 2), (64, 2048, 50, 4, 1), (24, 512, 20, 3, 0), (10, 256, 15, 4, 0)])
def test_grouped_steps_equal_the_unsplit_launch(E, N, H, groups, rpl):
    """K steps of every env through G groups (own handles, own streams, round-robin enqueueing, no waits in between) = K steps of
    one launch over all envs: bit for bit when the lane mapping is the same (forced, or the same by the size rule), because the
    Philox keys are global env indices and an env's arithmetic never depends on its launch partners."""
    from cartpolesimulation_amd.configs import MPPIConfig
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.pipeline import EnvGroups
    s0, tp, te, Lt = _inputs(E, H)
    cfg = MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=rpl)
    one = MPPIEngine(E, cfg)
    u1, Q1 = one.zeros(E, H), one.empty(E)
    for i in range(5):
        one.step(s0, u1, tp, te, L=Lt, seed=77, offset=i, env_offset=1000, Q_out=Q1)
    g = EnvGroups(E, cfg, groups, env_offset=1000)
    u2, Q2 = one.zeros(E, H), one.empty(E)
    step = g.prepare(s0, u2, tp, te, L=Lt, seed=77, Q_out=Q2)
    g.fork()
    g.run(step, None, periods=3, offset=0)                             # cpmppi_groups_run: every group's launches enqueued from C
    preps = g.prepare_step(s0, u2, tp, te, L=Lt, seed=77, Q_out=Q2)    # ... and the caller-paced form, one prepared step per group
    for i in (3, 4):
        for p in preps:
            p.run(offset=i)
    g.join()
    torch.cuda.synchronize()
    same_mapping = rpl != 0 or all(((e1 - e0) * N >= 131072) == (E * N >= 131072) for e0, e1 in g.slices)
    if same_mapping:
        assert torch.equal(u1, u2) and torch.equal(Q1, Q2)
    else:
        np.testing.assert_allclose(u2.cpu().numpy(), u1.cpu().numpy(), atol=1e-4)
    assert float(u2.abs().max()) > 0.01
    assert g.overlap(step, steps=5) > 0.5                              # (a ratio near 1 would mean the groups ran one after the other)
    g.close(); one.close()


def test_grouped_data_generator_loop_equals_the_single_chain():
    """harness.run_schedule (one chain) vs pipeline.run_schedule_groups (3 groups working in place on their slices of the batch's
    buffers, all periods enqueued by one cpmppi_groups_run call): identical recordings - states, second derivatives, controls -
    for experiments with moving targets and equilibrium flips."""
    from cartpolesimulation_amd import schedule as SC
    from cartpolesimulation_amd.configs import MPPIConfig
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.harness import BatchedCartPoleExperiment
    from cartpolesimulation_amd.pipeline import EnvGroups, run_schedule_groups
    E, N, H = 7, 512, 20
    cfg = dict(seed=41, length_of_experiment=0.5, dt=dict(saving=0.004), keep_target_equilibrium_x_seconds_up=0.1,
               keep_target_equilibrium_x_seconds_down=0.06, turning_points=dict(track_relative_complexity=12),
               random_initial_state=dict(init_limits=dict(angle=[0.0, 20.0], angleD=40.0, position=0.4, positionD=0.2)))
    mppi = MPPIConfig(num_rollouts=N, mpc_horizon=H, cost_function_specification="default")
    b = SC.RandomExperimentSetter(cfg).draw(E, 5, L=np.linspace(0.3, 0.45, E).astype(f32))
    eng = MPPIEngine(E, mppi)
    a = BatchedCartPoleExperiment(eng, seed=9).run_schedule(b)
    g = EnvGroups(E, mppi, 3)
    c = run_schedule_groups(g, b, seed=9)
    torch.cuda.synchronize()
    for k in ("states", "dd", "Q", "final_state", "u_nom"):
        assert torch.equal(a[k], c[k]), k
    assert a["Q"].shape == (b.n_periods + 1, E) and float(a["Q"].abs().max()) > 0.01
    # a shared device counter cannot serve several groups
    from cartpolesimulation_amd import _lib as L
    cnt = torch.zeros(1, dtype=torch.int64, device=a["Q"].device)
    bad = g.args_engine.prepare_step(a["final_state"], a["u_nom"], a["Q"][0], a["Q"][0], seed=1, offset_dev=cnt)
    with pytest.raises(L.CpmppiError):
        g.run(bad, None, periods=1)
    g.close(); eng.close()


def _real_rccl_id(lib):
    import ctypes as C
    from cartpolesimulation_amd import _lib as L
    uid = C.create_string_buffer(L.COMM_ID_BYTES)
    assert lib.cpmppi_comm_unique_id(uid, None) == 0, lib.cpmppi_last_error(None)
    return uid.raw


@pytest.mark.parametrize("waiter", ["stream-ops", "kernel"])
@pytest.mark.parametrize("E,groups", [(8, 2), (9, 3)])
def test_groups_under_one_communicator_equal_the_groups_without_a_collective(E, groups, waiter, monkeypatch):
    """cpmppi_groups_run_gather: the env groups of a device under ONE communicator and side stream - one real RCCL
    rank here; two ranks in test_gpu_two_rank_gather.py -, one all-gather of the whole u_nom[E, H] per period, with a 400 us spin in
    front of every all-gather (a slow peer: the gather ends deep inside the NEXT period's kernels, the groups drift apart), two
    alternating buffers, then in place.  Every period's gathered block is bit for bit what the same groups compute WITHOUT any
    collective, every block carries its step's stamp - in both forms of the side-stream waiter."""
    import ctypes as C
    from cartpolesimulation_amd import _lib as L
    from cartpolesimulation_amd.configs import MPPIConfig
    from cartpolesimulation_amd.pipeline import EnvGroups
    from cartpolesimulation_amd.shard import block_stamps
    monkeypatch.setenv("CPMPPI_COMM_WAITER", waiter)        # (env groups default to the one-kernel form: fewer side-stream dispatches)
    N, H, K, KB = 512, 20, 14, 9
    s0, tp, te, Lt = _inputs(E, H, seed=5)
    cfg = MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=1)
    dev = s0.device
    # the yardstick: the same groups, no communicator
    plain = EnvGroups(E, cfg, groups, env_offset=40)
    up = torch.zeros(E, H, device=dev)
    sp = plain.prepare(s0, up, tp, te, L=Lt, seed=91)
    plain.fork()
    want = []
    for i in range(K + KB + K):
        plain.run(sp, None, periods=1, offset=i)
        plain.join()
        want.append(up.clone())
    torch.cuda.synchronize()
    plain.close()
    g = EnvGroups(E, cfg, groups, env_offset=40)
    g.comm_init(_real_rccl_id(g.lib), 1, 0, stamped=True)
    h0 = C.c_void_p(g.lib.cpmppi_groups_handle(g._g, 0))
    assert g.lib.cpmppi_debug_comm_mode(h0) == (1 if waiter == "stream-ops" else 0)      # (env groups default to the kernel form)
    info = g.comm_info()
    assert info["rccl_ranks"] == 1 and info["stamped"] == 1 and info["world"] == 1
    assert g.lib.cpmppi_debug_comm_delay(h0, 400) == 0
    n, pad = E * H, L.GATHER_STAMP_FLOATS
    flat = [torch.zeros(n + pad, device=dev) for _ in range(2)]
    u = [f[:n].view(E, H) for f in flat]
    prep = [g.prepare(s0, u[b], tp, te, L=Lt, seed=91, u_nom_out=u[1 - b]) for b in range(2)]
    recv = torch.zeros(K + 1, 1, n + pad, device=dev)
    g.fork()
    for i in range(K):                                                         # one period per call, far ahead of the device
        g.run(prep[i & 1], None, periods=1, offset=i, gather_into=recv[i])
    g.run(prep[K & 1], None, periods=KB, offset=K, gather_into=recv[K])        # KB periods in ONE call (buffers alternate inside)
    g.join()
    torch.cuda.synchronize()
    g.comm_sync()
    for i in range(K):
        assert torch.equal(recv[i, 0, :n].view(E, H), want[i]), f"gather {i} is not period {i}'s result"
    assert torch.equal(recv[K, 0, :n].view(E, H), want[K + KB - 1]) and torch.equal(u[(K + KB) & 1], want[K + KB - 1])
    st = torch.stack([block_stamps(recv[i], n) for i in range(K + 1)]).view(-1).tolist()
    assert st == list(range(1, K + 1)) + [K + KB]
    # in place: period i + 1's finalize must not overwrite what gather i still reads
    ip = torch.zeros(n + pad, device=dev)
    ip[:n] = u[(K + KB) & 1].reshape(-1)
    pin = g.prepare(s0, ip[:n].view(E, H), tp, te, L=Lt, seed=91)
    recv2 = torch.zeros(K, 1, n + pad, device=dev)
    g.fork()
    for i in range(K):
        g.run(pin, None, periods=1, offset=K + KB + i, gather_into=recv2[i])
    g.join()
    torch.cuda.synchronize()
    g.comm_sync()
    for i in range(K):
        assert torch.equal(recv2[i, 0, :n].view(E, H), want[K + KB + i]), f"in place: gather {i} read a sequence a later period had overwritten"
    assert g.comm_info()["gathers_enqueued"] == 2 * K + KB
    # without a communicator / without a receive buffer: refused
    other = EnvGroups(E, cfg, groups)
    with pytest.raises(L.CpmppiError):
        other.run(other.prepare(s0, up, tp, te, L=Lt, seed=1), None, periods=1, gather_into=recv[0])
    other.close()
    g.close()


def test_groups_gather_timeout_drops_the_period_and_reaches_the_host():
    """A gather that joins 60 ms late under a 2 ms timeout, env groups: the finalizes of ALL groups that would overwrite the buffer
    it still reads give up (the buffer stays intact: the late gather delivers the right block), the next cpmppi_groups_run_gather
    returns CPMPPI_ERR_COMM, cpmppi_comm_sync reports once and clears, and the groups work again; while the error was up no block
    was stamped."""
    import ctypes as C
    from cartpolesimulation_amd import _lib as L
    from cartpolesimulation_amd.configs import MPPIConfig
    from cartpolesimulation_amd.pipeline import EnvGroups
    from cartpolesimulation_amd.shard import block_stamps
    E, N, H, groups = 8, 512, 20, 2
    s0, tp, te, Lt = _inputs(E, H, seed=6)
    cfg = MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=1)
    dev = s0.device
    g = EnvGroups(E, cfg, groups)
    g.comm_init(_real_rccl_id(g.lib), 1, 0, stamped=True, timeout_s=0.002)
    h0 = C.c_void_p(g.lib.cpmppi_groups_handle(g._g, 0))
    n, pad = E * H, L.GATHER_STAMP_FLOATS
    ip = torch.zeros(n + pad, device=dev)
    step = g.prepare(s0, ip[:n].view(E, H), tp, te, L=Lt, seed=92)
    r = torch.zeros(3, 1, n + pad, device=dev)
    g.fork()
    g.lib.cpmppi_debug_comm_delay(h0, 60000)
    g.run(step, None, periods=1, offset=0, gather_into=r[0])                  # period 0: fine; its gather is the late one
    g.lib.cpmppi_debug_comm_delay(h0, 0)
    g.join(); torch.cuda.current_stream().synchronize()
    after0 = ip[:n].clone()
    g.run(step, None, periods=1, offset=1, gather_into=r[1])                  # period 1 (in place): waits for gather 0, gives up
    g.join(); torch.cuda.current_stream().synchronize()
    assert torch.equal(ip[:n], after0)                                        # dropped by BOTH groups: nothing overwritten
    with pytest.raises(L.CpmppiError) as ei:
        g.run(step, None, periods=1, offset=2, gather_into=r[2])
    assert ei.value.code == -6 and "timed out" in str(ei.value)
    with pytest.raises(L.CpmppiError):
        g.comm_sync()                                                         # reports it once more and clears it
    g.comm_sync()
    torch.cuda.synchronize()
    assert torch.equal(r[0, 0, :n], after0) and torch.equal(r[1, 0, :n], after0)      # the late gather delivered period 0's block
    assert block_stamps(r[0], n).tolist() == [1] and block_stamps(r[1], n).tolist() == [1]   # gather 2 carries the OLD stamp: stale
    g.lib.cpmppi_comm_set_timeout(h0, 10.0)
    g.fork()
    g.run(step, None, periods=1, offset=2, gather_into=r[2])
    g.join(); torch.cuda.synchronize()
    g.comm_sync()
    assert not torch.equal(ip[:n], after0) and torch.equal(r[2, 0, :n], ip[:n]) and block_stamps(r[2], n).tolist() == [3]
    g.close()
