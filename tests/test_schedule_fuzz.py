"""CPU: randomised differential test of the product's batched schedule tabulation (cartpolesimulation_amd/schedule.py) against the
oracle's per-experiment, per-simulation-step restatement of the reference (oracle/schedule_np.py) - two independent implementations
of CartPole/data_generator.py:93-256, random_target_generator.py:9-89 and CartPole/__init__.py:360-388 - over random configurations:
time scales, lengths (incl. ones that end inside a control period), interpolation types and their alternation, regular / random /
given turning points, start / end options, usable track fraction, dwell times."""
import numpy as np
import pytest
from numpy.random import SFC64, Generator

from cartpolesimulation_amd import schedule as SC
from oracle import schedule_np as S


def random_config(rng):
    dt_sim = 0.002
    n_ctrl = int(rng.choice([1, 2, 5, 10, 25]))
    n_save = int(rng.choice([1, 2, 4, 5, 10, 15, 20, 50]))
    length = float(np.round(rng.uniform(0.05, 3.0), int(rng.choice([1, 2, 3]))))
    interp = [["previous", "0-derivative-smooth"], "linear", "previous", "0-derivative-smooth", ["linear", "previous", "linear"]][int(rng.integers(5))]
    given = [None, None, None, [0.0, 0.1, -0.1, 0.0], [0.07], []][int(rng.integers(6))]
    return dict(seed=int(rng.integers(1, 10 ** 6)), length_of_experiment=length,
                dt=dict(simulation=dt_sim, control=dt_sim * n_ctrl, saving=dt_sim * n_save),
                start_at_target=bool(rng.integers(2)), target_position_end=[None, 0.05, -0.12][int(rng.integers(3))],
                track_fraction_usable_for_target_position=float(rng.choice([1.0, 0.8, 0.5, 0.33])),
                initial_target_equilibrium=["up", "down", 1, -1][int(rng.integers(4))],
                keep_target_equilibrium_x_seconds_up=[10, 0.3, 0.05, "inf"][int(rng.integers(4))],
                keep_target_equilibrium_x_seconds_down=[2.5, 0.11, 0.02, "inf"][int(rng.integers(4))],
                random_initial_state=dict(position=[None, 0.01][int(rng.integers(2))], positionD=None, angle=[None, 0.3][int(rng.integers(2))],
                                          angleD=[None, 0.0][int(rng.integers(2))], target_position=[None, 0.03][int(rng.integers(2))],
                                          init_limits=dict(angle=[0.0, float(rng.choice([10.0, 180.0]))], angleD=float(rng.choice([30.0, 1200.0])),
                                                           position=0.8, positionD=0.5)),
                turning_points=dict(track_relative_complexity=float(rng.choice([0.4, 1, 3, 10])), interpolation_type=interp,
                                    turning_points=given, turning_points_period=["regular", "random"][int(rng.integers(2))]))


@pytest.mark.parametrize("seed", range(40))
def test_batched_tables_equal_the_per_experiment_oracle(seed):
    rng = Generator(SFC64(1000 + seed))
    cfg = SC.merged_config(random_config(rng))
    E, cseed = int(rng.integers(1, 5)), int(rng.integers(1, 10 ** 6))
    b = SC.RandomExperimentSetter(cfg).draw(E, cseed)
    es = S.ExperimentSetter(cfg)
    n_sim = int(np.ceil(cfg["length_of_experiment"] / cfg["dt"]["simulation"]))
    assert b.n_sim == n_sim and np.array_equal(b.times, S.accumulated_times(n_sim, cfg["dt"]["simulation"]))
    steps = np.arange(0, n_sim + 1, b.stride)
    inf = lambda v: np.inf if v == "inf" else v                       # noqa: E731
    for e in range(E):
        st = es.set(Generator(SFC64(cseed + e)))
        assert np.array_equal(b.s0[e], st["s0"]), (seed, e)
        assert b.interpolation_type[e] == st["interpolation_type"]
        _, tp, te = S.schedule_tables(st["trace"], st["target_equilibrium"], cfg["length_of_experiment"], cfg["dt"]["simulation"],
                                      inf(cfg["keep_target_equilibrium_x_seconds_up"]), inf(cfg["keep_target_equilibrium_x_seconds_down"]))
        assert np.array_equal(b.target_position[:, e], tp[steps]), (seed, e, cfg["turning_points"], np.abs(b.target_position[:, e] - tp[steps]).max())
        assert np.array_equal(b.target_equilibrium[:, e], te[steps]), (seed, e)
    # what the device loop relies on: every controller instant and every saved row is a table row
    assert b.n_ctrl % b.stride == 0 and b.n_save % b.stride == 0
    assert b.rows_at(np.arange(0, n_sim + 1, b.n_ctrl)).max() < b.target_position.shape[0]
