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
