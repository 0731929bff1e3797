"""The recording writer: file layout as CartPole/csv_logger.py writes it, column set as
CartPole/__init__.py:221-259 logs it, every field in the form the reference's own files have it - pinned to a recording the
reference's CartPole class wrote (tests/golden/schedule.npz, "csv_rows"); readable the way the reference's loaders read it."""
import ctypes as C
import dataclasses
import json
import os

import numpy as np
import pandas as pd
import pytest

from cartpolesimulation_amd import recording as R
from cartpolesimulation_amd.configs import PhysicalParameters

REFERENCE_COLUMNS = ["time", "angle", "angleD", "angleDD", "angle_cos", "angle_sin", "position", "positionD",
                     "positionDD", "Q_calculated", "Q_applied", "Q_ccrc", "u", "target_position", "target_equilibrium",
                     "L", "L_for_controller", "m_pole", "m_pole_for_controller", "vertical_angle_offset",
                     "vertical_angle_offset_cos", "vertical_angle_offset_sin", "Q_update_time"]
f32 = np.float32


def test_columns_match_reference_order():
    assert R.COLUMNS == REFERENCE_COLUMNS


def random_block(rng, T, E, first_update_row=2):
    s = (rng.standard_normal((T, E, 6)) * 10.0 ** rng.integers(-7, 3, (T, E, 6))).astype(f32)
    s[3, 1, 0], s[4, 1, 1], s[5, E - 1, 4], s[6, E - 1, 5] = 0.0, -0.0, 1.0, 123456.0
    Q = rng.uniform(-1, 1, (T, E)).astype(f32)
    return dict(time=np.cumsum(np.full(T, 0.002 * 5)) - 0.01, states=s, Q=Q, Q_ccrc=np.roll(Q, 1, axis=0),
                dd=np.stack([(1e3 * rng.standard_normal((T, E))), (1e-6 * rng.standard_normal((T, E)))], axis=-1).astype(f32),
                target_position=rng.uniform(-0.198, 0.198, (T, E)), target_equilibrium=rng.choice([-1, 1], (T, E)).astype(np.int32),
                L=np.broadcast_to(rng.uniform(0.2, 0.5, E).astype(f32), (T, E)).copy(), first_update_row=first_update_row)


def test_write_and_read_back(tmp_path):
    rng = np.random.Generator(np.random.SFC64(3))
    T = 7
    phys = PhysicalParameters()
    block = random_block(rng, T, 2)
    cols = R.typed_columns(block, 1, phys, q_update_time=0.25)
    header = R.create_csv_header(0.14, 0.002, 0.02, 0.02, "mpc", "mppi", phys)
    path = R.write_recording(R._unique_path(str(tmp_path), "CPS_test"), cols, header=header)
    lines = open(path).read().splitlines()
    assert lines[0].startswith("# This is CartPole simulation from ") and lines[1].startswith("# Done with git-revision: ")
    assert lines[2] == "#" and "# Length of experiment: 0.14 s" in lines and "# MPC Optimizer: mppi" in lines
    assert "# Saving: 0.02 s" in lines and "# Data:" in lines
    first_data = lines.index("# Data:") + 1
    assert lines[first_data].split(",") == REFERENCE_COLUMNS
    df = pd.read_csv(path, comment="#", float_precision="round_trip")
    assert list(df.columns) == REFERENCE_COLUMNS and len(df) == T
    np.testing.assert_array_equal(df["angleD"].to_numpy().astype(f32), block["states"][:, 1, 1])       # shortest float32 digits round-trip
    np.testing.assert_array_equal(df["target_position"].to_numpy(), block["target_position"][:, 1])
    assert list(df["L_for_controller"]) == [True] * T and df["Q_update_time"].isna().sum() == 2       # pandas reads 'true' as a bool
    # never overwrite an existing recording (csv_logger.py:76-88)
    p2 = R._unique_path(str(tmp_path), "CPS_test")
    assert p2 != path and p2.endswith("CPS_test-1.csv")
    with pytest.raises(FileExistsError):
        R.write_recording(path, cols, header=header)
    assert R.create_csv_file_name("mpc", "mppi", with_date=False) == "CPS_mpc_mppi.csv"


def test_float_formats_equal_pythons_and_numpys():
    """repr(float) and str(numpy.float32) as the native writer formats them, on random bit patterns of every magnitude."""
    from cartpolesimulation_amd import _lib
    lib = _lib.load()
    lib.cpmppi_debug_py_repr.argtypes = [C.c_double, C.c_char_p]
    lib.cpmppi_debug_np_str_f32.argtypes = [C.c_float, C.c_char_p]
    rng = np.random.Generator(np.random.SFC64(5))
    buf = C.create_string_buffer(64)
    f = rng.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32).view(f32)
    f = np.concatenate([f[np.isfinite(f)], np.array([0.0, -0.0, 1.0, 1e-4, 9.999e-5, 1e16, 9.9999e15, 123456.0, 0.1, 1e-45, 3.4e38], f32),
                        (rng.standard_normal(20000) * 10.0 ** rng.integers(-8, 8, 20000)).astype(f32)])
    for x in f:
        n = lib.cpmppi_debug_np_str_f32(float(x), buf)
        assert buf.raw[:n].decode() == str(x), (x.view(np.uint32), buf.raw[:n], str(x))
    d = rng.integers(0, 2 ** 63, 100000, dtype=np.uint64).view(np.float64)
    d = np.concatenate([d[np.isfinite(d)], -d[np.isfinite(d)][:1000], rng.standard_normal(20000) * 10.0 ** rng.integers(-20, 20, 20000),
                        np.array([0.0, -0.0, 0.020000000000000004, 1e16, 1e-5, 5e-324])])
    for x in d:
        n = lib.cpmppi_debug_py_repr(float(x), buf)
        assert buf.raw[:n].decode() == repr(float(x)), (x, buf.raw[:n])


def test_native_writer_is_byte_identical_to_the_csv_module(tmp_path):
    """cpmppi_write_recordings (host code of libcpmppi.so, no GPU involved) against write_recording (Python's csv module fed the
    reference's value TYPES) on the same block of E experiments: equal byte for byte - repr of Python floats, str of numpy
    float32 scalars (incl. exponent notation, integers, negative zero), ints, 'true', empty fields, "\\r\\n" rows, comment block."""
    rng = np.random.Generator(np.random.SFC64(7))
    T, E = 57, 5
    block = random_block(rng, T, E)
    phys = PhysicalParameters()
    header = R.create_csv_header(1.14, 0.002, 0.02, 0.01, "mpc", "mppi", phys)
    title = "This is CartPole simulation from 01.01.2026 at time 00:00:00"
    a = [str(tmp_path / f"py_{e}.csv") for e in range(E)]
    b = [str(tmp_path / f"native_{e}.csv") for e in range(E)]
    for e in range(E):
        R.write_recording(a[e], R.typed_columns(block, e, phys, 0.0123), title=title, header=header)
    for threads in (1, 3):
        for p in b:
            if os.path.exists(p):
                os.remove(p)
        R.write_recordings_native(b, block, phys, header, title=title, q_update_time=0.0123, n_threads=threads)
        for e in range(E):
            assert open(a[e], "rb").read() == open(b[e], "rb").read(), (e, threads)
    df = pd.read_csv(b[1], comment="#")
    assert list(df.columns) == REFERENCE_COLUMNS and len(df) == T


def test_native_writer_never_appends_and_cleans_up_after_a_failure(tmp_path):
    """advisor r4: a failing call must not leave a half-written dataset behind that a retry then appends to."""
    from cartpolesimulation_amd import _lib
    rng = np.random.Generator(np.random.SFC64(8))
    E = 4
    block = random_block(rng, 9, E)
    phys = PhysicalParameters()
    header = R.create_csv_header(1.0, 0.002, 0.02, 0.02, "mpc", "mppi", phys)
    good = [str(tmp_path / f"ok_{e}.csv") for e in range(E)]
    bad = list(good)
    bad[2] = str(tmp_path / "no_such_dir" / "x.csv")
    with pytest.raises(_lib.CpmppiError) as ei:
        R.write_recordings_native(bad, block, phys, header, n_threads=2)
    assert ei.value.code == -7 and "No such file or directory" in str(ei.value)                  # CPMPPI_ERR_IO with the errno text
    assert os.listdir(tmp_path) == []                                                            # nothing left: the retry starts clean
    R.write_recordings_native(good, block, phys, header, n_threads=2)
    sizes = [os.path.getsize(p) for p in good]
    with pytest.raises(_lib.CpmppiError) as ei:                                                  # existing names are refused, not appended to
        R.write_recordings_native(good, block, phys, header)
    assert ei.value.code == -7 and "File exists" in str(ei.value)
    assert [os.path.getsize(p) for p in good] == sizes and sorted(os.listdir(tmp_path)) == sorted(os.path.basename(p) for p in good)
    twice = [str(tmp_path / f"new_{e}.csv") for e in (0, 1, 2, 1)]                                # one name twice in a call: refused up front
    with pytest.raises(_lib.CpmppiError) as ei:
        R.write_recordings_native(twice, block, phys, header, n_threads=4)
    assert ei.value.code == -1 and "new_1.csv" in str(ei.value)
    assert sorted(os.listdir(tmp_path)) == sorted(os.path.basename(p) for p in good)


@pytest.mark.parametrize("key", ["exp_fine/0", "exp_fine/1", "exp_coarse/0", "exp_varM/0", "exp_dist/1", "exp_sensor/0"])
def test_writer_reproduces_the_references_own_recording(golden_dir, tmp_path, key):
    """The data rows of a recording written by the REFERENCE (its CartPole class + csv_logger, tests/golden/schedule.npz) from the
    values it logged: fed the same values, both writers here give the same bytes (all columns but the wall-clock Q_update_time)."""
    g = np.load(os.path.join(golden_dir, "schedule.npz"))
    col = lambda n: g[f"{key}/col/{n}"]                              # noqa: E731
    rows = len(col("time"))
    cfg = json.loads(g[f"{key.split('/')[0]}/config"].item())
    n_ctrl = int(np.rint(cfg["dt"]["control"] / cfg["dt"]["simulation"]))
    n_save = int(np.rint(cfg["dt"]["saving"] / cfg["dt"]["simulation"]))
    st = np.stack([col(n) for n in ("angle", "angleD", "angle_cos", "angle_sin", "position", "positionD")], axis=-1).astype(f32)
    block = dict(time=col("time"), states=st[:, None], dd=np.stack([col("angleDD"), col("positionDD")], axis=-1).astype(f32)[:, None],
                 Q=col("Q_calculated").astype(f32)[:, None], Q_ccrc=col("Q_ccrc").astype(f32)[:, None],
                 target_position=col("target_position")[:, None], target_equilibrium=col("target_equilibrium").astype(np.int32)[:, None],
                 L=col("L").astype(f32)[:, None], first_update_row=-(-n_ctrl // n_save))
    if key.startswith("exp_dist"):                                   # the control disturbance: Q_applied differs from Q_calculated
        block["Q_applied"] = col("Q_applied").astype(f32)[:, None]
        assert np.abs(block["Q_applied"] - block["Q"]).max() > 0.3
    if key.startswith("exp_sensor"):                                 # a vertical angle offset that moves: three columns of Python floats
        off = col("vertical_angle_offset")
        block["angle_offset"] = np.stack([off, np.cos(off), np.sin(off)], axis=-1)[:, None]
        block["informed"] = (col("L_for_controller") == "true").astype(np.uint8)[:, None]
        assert len(np.unique(off)) > 8
    if key.startswith("exp_varM"):                                   # a pole mass that changes, a controller informer that switches
        block["m_pole"] = col("m_pole").astype(f32)[:, None]
        block["informed"] = (col("L_for_controller") == "true").astype(np.uint8)[:, None]
        assert len(np.unique(block["m_pole"])) > 5 and 0 < block["informed"].mean() < 1
        assert np.array_equal(col("L_for_controller"), col("m_pole_for_controller"))
    # (the fixture's columns are what the reference held: float32 values widened by pandas; narrowing them back is exact)
    assert np.array_equal(block["states"][:, 0, 0].astype(np.float64), col("angle")) and np.array_equal(block["Q"][:, 0].astype(np.float64), col("Q_calculated"))
    assert np.array_equal(col("u").astype(f32), f32(1.77) * block.get("Q_applied", block["Q"])[:, 0])   # u = u_max * Q_applied in float32
    phys = PhysicalParameters()
    want = g[f"{key}/csv_rows"].item().split("\r\n")
    assert want[0].split(",") == REFERENCE_COLUMNS[:-1] and len(want) == rows + 1
    header = R.create_csv_header(cfg["length_of_experiment"], 0.002, 0.02, cfg["dt"]["saving"], "mppi-cartpole", "", phys)
    for native in (True, False):
        path = str(tmp_path / f"{'native' if native else 'python'}.csv")
        if native:
            R.write_recordings_native([path], block, phys, header, q_update_time=0.5)
        else:
            R.write_recording(path, R.typed_columns(block, 0, phys, 0.5), header=header)
        got = open(path, newline="").read().split("\r\n")
        k0 = next(i for i, r in enumerate(got) if r.startswith("time,"))
        body = [r for r in got[k0:] if r]
        assert [r.rsplit(",", 1)[0] for r in body] == want, native
        tails = [r.rsplit(",", 1)[1] for r in body[1:]]
        assert tails[:block["first_update_row"]] == [""] * block["first_update_row"] and set(tails[block["first_update_row"]:]) == {"0.5"}
    # the header block (below the title and revision lines) is the reference's too, parameter for parameter, up to its dict-valued
    # entries (noise, updaters of L / m_pole / vertical_angle_offset, informer), which this build has no counterpart of
    ref_head = g[f"{key}/csv_preamble"].item().split("\r\n")
    ours = ["# " + h for h in header]
    for line in ("# Length of experiment: %s s" % cfg["length_of_experiment"], "# Time intervals dt:", "# Simulation: 0.002 s",
                 "# Controller update: 0.02 s", "# Saving: %s s" % cfg["dt"]["saving"], "# Controller: mppi-cartpole", "# Data:"):
        assert line in ref_head and line in ours, line


def test_dataset_paths_follow_the_reference_naming(tmp_path):
    """run_data_generator names every file "Experiment" and csv_logger appends -1, -2, ... (CartPole/data_generator.py:290-322,
    csv_logger.py:61-91); ML_Pipeline_mode sorts them into Train / Validate / Test by position (split 0.8 / 0.1)."""
    p = R.dataset_paths(4, str(tmp_path))
    assert [os.path.basename(x) for x in p] == ["Experiment.csv", "Experiment-1.csv", "Experiment-2.csv", "Experiment-3.csv"]
    open(p[0], "w").close()                                          # a later run continues the numbering past existing files
    assert os.path.basename(R.dataset_paths(1, str(tmp_path))[0]) == "Experiment-1.csv"
    p = R.dataset_paths(2, str(tmp_path / "idx"), secondary_experiment_index=7)
    assert [os.path.basename(x) for x in p] == ["Experiment-007.csv", "Experiment-007-1.csv"]
    p = R.dataset_paths(10, str(tmp_path / "ml"), ml_pipeline=True, split=(0.8, 0.1))
    assert [os.path.basename(os.path.dirname(x)) for x in p] == ["Train"] * 8 + ["Validate"] + ["Test"]
    assert [os.path.basename(x) for x in p[:3]] == ["Experiment.csv", "Experiment-1.csv", "Experiment-2.csv"] and os.path.basename(p[8]) == "Experiment.csv"
    f1 = R.experiment_folder(str(tmp_path / "exps"))
    assert f1.endswith(os.path.join("Experiment-1", "Recordings"))
    os.makedirs(f1)
    assert R.experiment_folder(str(tmp_path / "exps")).endswith(os.path.join("Experiment-2", "Recordings"))
    assert R.experiment_folder(str(tmp_path / "exps"), 3).endswith(os.path.join("Experiment-003-1", "Recordings"))


@pytest.mark.gpu
def test_generate_dataset_on_device(tmp_path):
    torch = pytest.importorskip("torch")
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import legacy_mppi_config
    E = 4
    eng = MPPIEngine(E, legacy_mppi_config(num_rollouts=512, mpc_horizon=20))
    cfg = dict(length_of_experiment=0.4, dt=dict(saving=0.01), keep_target_equilibrium_x_seconds_up=0.1,
               turning_points=dict(track_relative_complexity=10),
               random_initial_state=dict(init_limits=dict(angle=[0.0, 10.0], angleD=20.0, position=0.3, positionD=0.1)))
    paths = R.generate_dataset(eng, E, str(tmp_path), config=cfg, seed=3)
    assert [os.path.basename(p) for p in paths] == ["Experiment.csv", "Experiment-1.csv", "Experiment-2.csv", "Experiment-3.csv"]
    df = pd.read_csv(paths[1], comment="#")
    assert list(df.columns) == REFERENCE_COLUMNS and len(df) == 41                    # t = 0, 0.01, ..., 0.4
    np.testing.assert_allclose(np.diff(df["time"]), 0.01, atol=1e-9)
    np.testing.assert_allclose(df["angle_cos"], np.cos(df["angle"]), atol=1e-5)
    np.testing.assert_allclose(df["u"], 1.77 * df["Q_applied"], rtol=1e-6)
    assert np.abs(df["Q_applied"]).max() <= 1.0 and np.isfinite(df.drop(columns=["Q_update_time"]).to_numpy(dtype=float)).all()
    assert np.ptp(df["target_position"]) > 0 and set(df["target_equilibrium"]) == {1, -1}    # moving target, flipping equilibrium
    q = df["Q_calculated"].to_numpy()
    assert np.array_equal(q[0:40:2], q[1:40:2])                                       # the control is held for dt_control = 2 rows
    # (Q_ccrc is written with float32 digits, Q_calculated as the double: the same float32 values)
    assert np.array_equal(df["Q_ccrc"].to_numpy()[2:].astype(np.float32), q[:-2].astype(np.float32)) and (df["Q_ccrc"][:2] == 0).all()
    # second derivatives are those of the logged state under the logged control (cartpole_equations.py:44-105)
    from oracle import oracle_np as O
    for i in (0, 5, 40):
        r = df.iloc[i]
        add, pdd = O.cartpole_ode(np.float32(r.angle_cos), np.float32(r.angle_sin), np.float32(r.angleD),
                                  np.float32(r.positionD), np.float32(1.77 * r.Q_applied), np.float32(r.L))
        assert abs(add - r.angleDD) < 1e-3 * max(1, abs(add)) and abs(pdd - r.positionDD) < 1e-3 * max(1, abs(pdd)), i
    # ML pipeline mode: the folder structure SI_Toolkit expects
    paths = R.generate_dataset(eng, E, str(tmp_path / "ml"), config=dict(cfg, ML_Pipeline_mode=True, split=[0.5, 0.25]), seed=4)
    assert [os.path.relpath(p, str(tmp_path / "ml")) for p in paths] == [
        os.path.join("Experiment-1", "Recordings", d, n) for d, n in (("Train", "Experiment.csv"), ("Train", "Experiment-1.csv"),
                                                                       ("Validate", "Experiment.csv"), ("Test", "Experiment.csv"))]
    # one process per GPU: two ranks' blocks of the same run = the single-process run, row for row (global Philox keys, the same
    # random streams); every rank names its files like a job of the reference's array (-i rank)
    whole = R.generate_dataset(eng, E, str(tmp_path / "whole"), config=cfg, seed=6)
    parts = []
    for r in range(2):
        parts += R.generate_dataset(eng, E, str(tmp_path / "ranks"), config=cfg, seed=6, rank=r, world=2)
    assert [os.path.basename(p) for p in parts] == ["Experiment-000.csv", "Experiment-000-1.csv", "Experiment-001.csv", "Experiment-001-1.csv"]
    for a, b in zip(whole, parts):
        da, db = (pd.read_csv(p, comment="#", float_precision="round_trip").drop(columns=["Q_update_time"]) for p in (a, b))
        assert da.equals(db), (a, b)
    # the simulator's parameter updaters and controller informer (cartpole_physical_parameters.yml): a pole length and a pole mass that
    # change during the run, a controller that is told the true length only part of the time - the columns follow the tables, the
    # second derivatives are those of the row's OWN length and mass, and groups / ranks change nothing
    from cartpolesimulation_amd import schedule as SC
    prm = dict(L=dict(init_value=0.395, change_every_x_seconds=0.014, mode="bounce", range_random=[0.2, 0.5], range_clip=[0.3, 0.45],
                      increment=0.01, reset_every_x_seconds="inf"),
               m_pole=dict(init_value=0.087, change_every_x_seconds=0.03, mode="random walk", range_random=[0.015, 0.15],
                           range_clip=[0.05, 0.12], increment=0.004, reset_every_x_seconds="inf"),
               inform_controller_about_parameters_change=dict(mode="switching_regular", change_to_on_after_x_seconds_off=0.05,
                                                              change_to_off_after_x_seconds_on=0.07),
               controlDisturbance=0.2, controlBias=-0.02, seed=11, latency=0.007,
               noise=dict(noise_mode="ON", sigma_angle=0.0, sigma_position=0.0005, sigma_angleD=0.075, sigma_positionD=0.005),
               vertical_angle_offset=dict(init_value=1.0, change_every_x_seconds=0.05, mode="increase", range_random=[-3.14, 3.14],
                                          range_clip=None, increment=0.004, reset_every_x_seconds="inf"))
    pp = R.generate_dataset(eng, E, str(tmp_path / "prm"), config=cfg, seed=7, parameters=prm)
    times = SC.accumulated_times(200, 0.002)
    Ltab = SC.parameter_table(prm["L"], times)
    told = SC.informer_table(prm["inform_controller_about_parameters_change"], times, 10)
    masses = []
    for e, pth in enumerate(pp):
        d = pd.read_csv(pth, comment="#", float_precision="round_trip")
        assert np.array_equal(d["L"].to_numpy().astype(np.float32), Ltab[::5]) and len(np.unique(d["L"])) > 8
        assert list(d["L_for_controller"]) == ["true" if x else "default" for x in told[::5]] == list(d["m_pole_for_controller"])
        m = d["m_pole"].to_numpy().astype(np.float32)
        masses.append(m)
        zq = SC.control_disturbance(E, 21, 11)[:, e]                                    # one draw per controller call: rows 0, 2, 4, ...
        qc_, qa_ = d["Q_calculated"].to_numpy().astype(np.float32), d["Q_applied"].to_numpy().astype(np.float32)
        assert np.array_equal(qa_[::2], ((qc_[::2] + np.float32(0.2) * zq).astype(np.float32) + np.float32(-0.02)).astype(np.float32))
        assert np.array_equal(d["u"].to_numpy().astype(np.float32), np.float32(1.77) * qa_) and np.abs(qa_ - qc_).max() > 0.2
        vo = d["vertical_angle_offset"].to_numpy()
        assert vo[0] == np.deg2rad(1.0) and (np.diff(vo) >= 0).all() and 5 <= len(np.unique(vo)) <= 9
        assert np.array_equal(d["vertical_angle_offset_cos"].to_numpy(), np.cos(vo)) and np.array_equal(d["vertical_angle_offset_sin"].to_numpy(), np.sin(vo))
        assert len(np.unique(m)) > 3 and m.min() >= np.float32(0.05) and m.max() <= np.float32(0.12) and m[0] == np.float32(0.087)
        for i in (0, 7, 23, 40):
            r = d.iloc[i]
            par = dataclasses.replace(O.DEFAULT_PARAMS, m_pole=np.float32(r.m_pole))
            add, pdd = O.cartpole_ode(np.float32(r.angle_cos), np.float32(r.angle_sin), np.float32(r.angleD), np.float32(r.positionD),
                                      np.float32(1.77 * r.Q_applied), np.float32(r.L), par)
            assert abs(add - r.angleDD) < 1e-3 * max(1, abs(add)) and abs(pdd - r.positionDD) < 1e-3 * max(1, abs(pdd)), (e, i)
    assert not np.array_equal(masses[0], masses[1])                                    # a random walk per experiment
    again = R.generate_dataset(eng, E, str(tmp_path / "prm_groups"), config=cfg, seed=7, parameters=prm, groups=2)
    ranks = []
    for r in range(2):
        ranks += R.generate_dataset(eng, E, str(tmp_path / "prm_ranks"), config=cfg, seed=7, parameters=prm, rank=r, world=2)
    for a, b2, c2 in zip(pp, again, ranks):
        da, db, dc = (pd.read_csv(p, comment="#", float_precision="round_trip").drop(columns=["Q_update_time"]) for p in (a, b2, c2))
        assert da.equals(db) and da.equals(dc), (a, b2, c2)


@pytest.mark.gpu
def test_generator_cli_reads_a_checkouts_yaml_files(tmp_path, capsys):
    """python -m cartpolesimulation_amd.recording --config-root <checkout>: config_data_gen.yml decides the experiments, the physical-
    parameters file the plant and its parameter schedule (here: a pole length in 'bounce' mode, an informer that is OFF, a control
    disturbance), config_optimizers.yml's mppi section the controller - what run_data_generator.py reads from its working directory."""
    pytest.importorskip("torch")
    import yaml
    from cartpolesimulation_amd import schedule as SC
    root = tmp_path / "checkout"
    (root / "Control_Toolkit_ASF").mkdir(parents=True)
    (root / "SI_Toolkit_ASF").mkdir()
    dg = SC.default_data_gen_config()
    dg.update(seed=5, length_of_experiment=0.2, number_of_experiments=3, PATH_TO_EXPERIMENT_RECORDINGS_DEFAULT=str(tmp_path / "rec") + "/")
    dg["dt"]["saving"] = 0.01
    dg["turning_points"]["track_relative_complexity"] = 10
    dump = lambda rel, obj: (root / rel).write_text(yaml.safe_dump(obj))     # noqa: E731
    dump("config_data_gen.yml", dg)
    dump("cartpole_physical_parameters.yml", dict(cartpole=dict(
        seed=3, m_cart=0.23, u_max=1.77, M_fric=3.22, J_fric=5.0e-5, v_max=0.8, cart_length=4.4e-2, track_length=44.0e-2, g=9.81, k="1.0/3.0",
        controlDisturbance_mode="additive", controlDisturbance=0.1, controlBias=0.0,
        L=dict(init_value=0.395, change_every_x_seconds=0.014, mode="bounce", range_random=[0.2, 0.5], range_clip=[0.3, 0.45], increment=0.01,
               reset_every_x_seconds="inf"),
        m_pole=dict(init_value=0.087, change_every_x_seconds=2, mode="constant", range_random=[0.015, 0.15], range_clip=[0.015, 0.15],
                    increment=0.002, reset_every_x_seconds="inf"),
        inform_controller_about_parameters_change=dict(mode="OFF", change_to_on_after_x_seconds_off=1.5, change_to_off_after_x_seconds_on=4))))
    dump("Control_Toolkit_ASF/config_optimizers.yml", dict(mppi=dict(seed=None, mpc_horizon=20, mpc_timestep=0.02, num_rollouts=256, cc_weight=1.0,
                                                                     R=1.0, LBD=100.0, NU=1000.0, SQRTRHOINV=0.03,
                                                                     period_interpolation_inducing_points=10)))
    dump("Control_Toolkit_ASF/config_controllers.yml", dict(mpc=dict(optimizer="mppi", predictor_specification="ODE_v0",
                                                                     cost_function_specification="quadratic_boundary_grad_minimal")))
    dump("Control_Toolkit_ASF/config_cost_function.yml", dict(cost_function_name_default="quadratic_boundary_grad_minimal",
                                                              CartPole=dict(quadratic_boundary_grad_minimal={})))
    dump("SI_Toolkit_ASF/config_predictors.yml", dict(predictors=dict(ODE_v0_default=dict(predictor_type="ODE_v0", intermediate_steps=10))))
    R.main(["--config-root", str(root)])                               # (this checkout's config_controllers.yml names mppi)
    assert "wrote 3 recordings" in capsys.readouterr().out
    files = sorted(os.listdir(tmp_path / "rec"))
    assert files == ["Experiment-1.csv", "Experiment-2.csv", "Experiment.csv"]
    d = pd.read_csv(tmp_path / "rec" / "Experiment-1.csv", comment="#", float_precision="round_trip")
    assert len(d) == 21 and len(np.unique(d["L"])) > 4 and set(d["L_for_controller"]) == {"default"} and set(d["m_pole"]) == {float(f32(0.087))}
    qc, qa = d["Q_calculated"].to_numpy().astype(f32), d["Q_applied"].to_numpy().astype(f32)
    z = SC.control_disturbance(3, 11, 3)[:, 1]                                          # experiment 1 of the run, seed of the YAML
    assert np.array_equal(qa[::2], ((qc[::2] + f32(0.1) * z).astype(f32) + f32(0.0)).astype(f32))
    # another optimizer of the package: the checkout's own section of config_optimizers.yml, the same plant, schedule and recording
    import yaml as _y
    opts = _y.safe_load((root / "Control_Toolkit_ASF" / "config_optimizers.yml").read_text())
    opts["rpgd"] = dict(seed=None, mpc_horizon=12, mpc_timestep=0.02, SAMPLING_DISTRIBUTION="uniform", period_interpolation_inducing_points=4,
                        learning_rate=0.05, adam_beta_1=0.9, adam_beta_2=0.999, adam_epsilon=1.0e-8, gradmax_clip=5, rtol=1.0e-3,
                        num_rollouts=16, opt_keep_k_ratio=0.75, outer_its=2, resamp_per=10, sample_stdev=0.5, sample_mean=0.0,
                        sample_whole_control_space=True, uniform_dist_max=0.8, uniform_dist_min=-0.8, shift_previous=1, warmup=False,
                        warmup_iterations=250)
    dump("Control_Toolkit_ASF/config_optimizers.yml", opts)
    R.main(["--config-root", str(root), "--optimizer", "rpgd", "--experiments", "2", "--out", str(tmp_path / "rec_rpgd")])
    d2 = pd.read_csv(tmp_path / "rec_rpgd" / "Experiment-1.csv", comment="#", float_precision="round_trip")
    assert len(d2) == 21 and np.isfinite(d2["angle"]).all() and np.abs(d2["Q_calculated"]).max() > 0.01
    assert np.array_equal(d2["L"].to_numpy(), d["L"].to_numpy()) and list(d2["L_for_controller"]) == list(d["L_for_controller"])
    head = open(tmp_path / "rec_rpgd" / "Experiment-1.csv").read(2000)
    assert "rpgd" in head
    # flags override the files
    R.main(["--config-root", str(root), "--experiments", "2", "--length", "0.1", "--out", str(tmp_path / "rec2")])
    assert sorted(os.listdir(tmp_path / "rec2")) == ["Experiment-1.csv", "Experiment.csv"]
    assert len(pd.read_csv(tmp_path / "rec2" / "Experiment.csv", comment="#")) == 11


@pytest.mark.skipif(not os.path.isdir("/root/reference/Control_Toolkit_ASF"), reason="reference checkout not mounted")
def test_generator_cli_on_the_reference_checkout(monkeypatch):
    """--config-root on the reference's own tree (CPU: engine and run captured): its config_data_gen.yml as shipped, the mppi
    section of config_optimizers.yml, the mpc cost and predictor of config_controllers.yml, the shipped physical parameters (nothing
    active in them: no parameter schedule is built)."""
    import cartpolesimulation_amd.engine as ENG
    seen = {}

    class Engine:
        def __init__(self, E, cfg, phys=None, device=0):
            seen.update(E=E, cfg=cfg, phys=phys)

    def gen(engine, n, out, **kw):
        seen.update(n=n, out=out, **kw)
        return [os.path.join(out, "Experiment.csv")]

    monkeypatch.setattr(ENG, "MPPIEngine", Engine)
    monkeypatch.setattr(R, "generate_dataset", gen)
    R.main(["--config-root", "/root/reference", "--seed", "4", "--experiments", "8", "--optimizer", "mppi"])
    import yaml
    dg = yaml.safe_load(open("/root/reference/config_data_gen.yml"))
    opt = yaml.safe_load(open("/root/reference/Control_Toolkit_ASF/config_optimizers.yml"))["mppi"]
    assert seen["E"] == 8 and seen["n"] == 8 and seen["seed"] == 4 and seen["parameters"] is None
    assert seen["config"]["length_of_experiment"] == dg["length_of_experiment"] and seen["config"]["dt"] == dg["dt"]
    assert seen["config"]["turning_points"] == dg["turning_points"] and seen["out"] == dg["PATH_TO_EXPERIMENT_RECORDINGS_DEFAULT"]
    assert seen["cfg"].num_rollouts == opt["num_rollouts"] and seen["cfg"].mpc_horizon == opt["mpc_horizon"]
    assert seen["cfg"].cost_function_specification == "quadratic_boundary_grad_minimal" and seen["cfg"].predictor_type == "ODE"
    assert abs(seen["phys"].m_pole - 0.087) < 1e-6 and abs(seen["phys"].TrackHalfLength - 0.198) < 1e-6
