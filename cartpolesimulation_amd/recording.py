"""Experiment recordings in the reference's CSV format and the batched data generator (N1 + N2).

Format (CartPole/csv_logger.py:10-58,125-159): comment block of ``# `` lines — title, git revision, ``#``, header with the time
intervals, controller, optimizer and physical parameters, ``# Data:`` — then one row with the column names and one row per saved
time step (every dt_save).  Column set and order = ``CartPole.variables_to_log`` (CartPole/__init__.py:221-259).  Downstream
tooling reads these files with ``pandas.read_csv(path, comment='#')``.

What a field looks like in the reference's files depends on the TYPE its simulator holds the value in (csv.writer: repr() for a
Python float, str() for everything else): Python floats for time, Q_calculated, target_position, L, m_pole; numpy float32 scalars
for the state, the second derivatives, Q_applied, Q_ccrc, u; an int for target_equilibrium; the controller informer's string
'true' for L_for_controller / m_pole_for_controller; None (an empty field) for Q_update_time until the first controller update
inside the loop.  ``typed_columns`` reproduces those types, so that ``write_recording`` (Python's csv module) and the native
``cpmppi_write_recordings`` both give the reference's bytes - pinned to a file the reference's own CartPole class wrote
(tests/golden/schedule.npz "csv_rows", tests/test_recording.py).

``generate_dataset`` is the counterpart of ``run_data_generator.py`` -> ``CartPole/data_generator.py:259-367``: instead of
``number_of_experiments`` sequential single-env runs it runs them all at once on the GPU (schedule.py tabulates every experiment's
random target trace and equilibrium flips, harness.run_schedule runs plant + MPPI in one device loop) and writes one recording per
experiment under the reference's file names, with its Train / Validate / Test split in ML_Pipeline_mode.
"""
import csv
import os
from datetime import datetime

import numpy as np

COLUMNS = ["time", "angle", "angleD", "angleDD", "angle_cos", "angle_sin", "position", "positionD", "positionDD",
           "Q_calculated", "Q_applied", "Q_ccrc", "u", "target_position", "target_equilibrium", "L", "L_for_controller",
           "m_pole", "m_pole_for_controller", "vertical_angle_offset", "vertical_angle_offset_cos",
           "vertical_angle_offset_sin", "Q_update_time"]
f32 = np.float32


def create_csv_file_name(controller_name="mpc", optimizer_name="mppi", prefix="CPS", with_date=True, title=""):
    """CartPole/csv_logger.py:96-116."""
    date = datetime.now().strftime("_%Y-%m-%d_%H-%M-%S") if with_date else ""
    name_controller = "" if controller_name == "" else "_" + controller_name + ("_" + optimizer_name if optimizer_name else "")
    return prefix + ("_" + title if title else "") + name_controller + date + ".csv"


def _unique_path(folder, csv_name, taken=(), start=None):
    """CartPole/csv_logger.py:61-91: never overwrite, append -1, -2, ... (`taken`: names already given out in this batch;
    `start`: {base: first index worth trying} - a batch of n files would otherwise probe n^2 / 2 names)."""
    os.makedirs(folder, exist_ok=True)
    if not csv_name.endswith(".csv"):
        csv_name += ".csv"
    path = os.path.join(folder, csv_name)
    base = path[:-4]
    idx = 0 if start is None else start.get(base, 0)       # 0: the plain name has not been tried yet
    if idx > 0:
        path = f"{base}-{idx}.csv"
    idx += 1
    while os.path.isfile(path) or path in taken:
        path = f"{base}-{idx}.csv"
        idx += 1
    if start is not None:
        start[base] = idx
    return path


def create_csv_header(length_of_experiment, dt_simulation, dt_controller, dt_save, controller_name, optimizer_name, phys):
    """CartPole/csv_logger.py:125-159."""
    header = [f"Length of experiment: {length_of_experiment} s", "", "Time intervals dt:",
              f"Simulation: {dt_simulation} s", f"Controller update: {dt_controller} s", f"Saving: {dt_save} s", "",
              f"Controller: {controller_name}"]
    if optimizer_name:
        header.append(f"MPC Optimizer: {optimizer_name}")
    header.append("Parameters:")
    for k, v in vars(phys).items():
        header.append(f"{k}: {v}")
    header += ["", "Data:"]
    return header


def _title(title):
    return title or (f"This is CartPole simulation from {datetime.now().strftime('%d.%m.%Y')}"
                     f" at time {datetime.now().strftime('%H:%M:%S')}")


def write_recording(path, columns, title=None, header=(), revision="cartpolesimulation_amd"):
    """columns: dict name -> sequence (all the same length), written in dict order with Python's csv module exactly as
    CartPole/csv_logger.py:10-58 does (a Python float is written with repr(), anything else with str(), None as '')."""
    if os.path.exists(path):
        raise FileExistsError(path)                             # (the reference never reuses a name: csv_logger.py:76-88)
    with open(path, "x", newline="") as f:
        w = csv.writer(f)
        w.writerow(["# " + _title(title)])
        w.writerow(["# Done with git-revision: {}".format(revision)])
        w.writerow(["#"])
        for line in header:
            w.writerow(["# " + line])
        w.writerow(list(columns.keys()))
        w.writerows(zip(*[v.tolist() if isinstance(v, np.ndarray) and v.dtype == np.float64 else list(v) for v in columns.values()]))
    return path


def preamble_bytes(header, columns=None, title=None, revision="cartpolesimulation_amd"):
    """The comment block and the column-name row exactly as write_recording's csv.writer emits them, as bytes (what
    cpmppi_write_recordings puts at the top of every file)."""
    import io
    f = io.StringIO(newline="")
    w = csv.writer(f)
    w.writerow(["# " + _title(title)])
    w.writerow(["# Done with git-revision: {}".format(revision)])
    w.writerow(["#"])
    for line in header:
        w.writerow(["# " + line])
    w.writerow(list(columns or COLUMNS))
    return f.getvalue().encode()


# ------------------------------------------------------------------------------------------------ rows of a run
def recording_block(result, phys, L_default=None):
    """A harness.run_schedule result -> the per-row host arrays of ALL its experiments (one copy per log):
    time [R] float64; states [R,E,6], dd [R,E,2], Q [R,E], Q_ccrc [R,E], L [R,E] float32; target_position [R,E] float64;
    target_equilibrium [R,E] int32; first_update_row.  Row r is simulation step r * n_save: its control is the one of the last
    controller update at or before that step (Update_Q precedes save_csv_routine, CartPole/__init__.py:316-324), Q_ccrc the
    control before that one (:489), 0 before the first update inside the loop."""
    b = result["batch"]
    c = lambda t: t.detach().cpu().numpy()                          # noqa: E731
    states, dd, Qc = c(result["states"]), c(result["dd"]), c(result["Q"])
    R, E = states.shape[0], states.shape[1]
    steps = np.arange(R) * b.n_save
    k = np.minimum(steps // b.n_ctrl, Qc.shape[0] - 1)               # controller call whose control is in force at the row
    Q = Qc[k]
    Qa = Qc
    if b.Q_disturbance is not None:                                  # Q_applied = (Q_calculated + disturbance) + bias, float32 (plant_kernel)
        Qa = ((Qc + np.asarray(b.Q_disturbance, f32)).astype(f32) + f32(b.Q_bias)).astype(f32)
    Q_ccrc = np.where((k > 0)[:, None], Qa[np.maximum(k - 1, 0)], f32(0.0)).astype(f32)
    rows = b.rows_at(steps)
    if b.L_table is not None:
        L = np.asarray(b.L_table, f32)[np.minimum(steps, b.L_table.shape[0] - 1)]
    else:
        L = np.broadcast_to(np.asarray(b.L if b.L is not None else (phys.L if L_default is None else L_default), f32), (R, E))
    out = dict(time=np.ascontiguousarray(b.times[steps]), states=states, dd=dd, Q=np.ascontiguousarray(Q), Q_ccrc=Q_ccrc,
               target_position=np.ascontiguousarray(b.target_position[rows]),
               target_equilibrium=np.ascontiguousarray(b.target_equilibrium[rows].astype(np.int32)),
               L=np.ascontiguousarray(L, dtype=f32), first_update_row=int(-(-b.n_ctrl // b.n_save)))
    if Qa is not Qc:
        out["Q_applied"] = np.ascontiguousarray(Qa[k])
    if b.m_pole_table is not None:                                   # the `m_pole:` updater's values, row by row
        out["m_pole"] = np.ascontiguousarray(np.asarray(b.m_pole_table, f32)[np.minimum(steps, b.m_pole_table.shape[0] - 1)])
    if b.angle_offset is not None:                                   # vertical_angle_offset and its cos / sin (numpy's, as the reference logs them)
        off = np.asarray(b.angle_offset, np.float64)[np.minimum(steps, b.angle_offset.shape[0] - 1)]
        out["angle_offset"] = np.ascontiguousarray(np.stack([off, np.cos(off), np.sin(off)], axis=-1))
    if b.informed is not None:                                       # the controller informer's answer in force at the row
        told = np.asarray(b.informed, bool)[np.minimum(steps, len(b.informed) - 1)]
        out["informed"] = np.ascontiguousarray(np.broadcast_to(told if told.ndim == 2 else told[:, None], (R, E)), dtype=np.uint8)
    return out


def typed_columns(block, env, phys, q_update_time=0.0):
    """One experiment of a recording block as the reference's column dict, every value in the TYPE the reference logs it in."""
    s, dd, Q = block["states"][:, env], block["dd"][:, env], block["Q"][:, env]
    Qa = block["Q_applied"][:, env] if block.get("Q_applied") is not None else Q
    R = s.shape[0]
    py = lambda a: [float(x) for x in a]                            # noqa: E731  (Python floats: written with repr)
    u_max = f32(phys.u_max)
    m_pole = py(block["m_pole"][:, env]) if block.get("m_pole") is not None else [float(f32(phys.m_pole))] * R
    told = ["true" if x else "default" for x in block["informed"][:, env]] if block.get("informed") is not None else ["true"] * R
    ao = block["angle_offset"][:, env] if block.get("angle_offset") is not None else None
    cols = {"time": py(block["time"]), "angle": list(s[:, 0]), "angleD": list(s[:, 1]), "angleDD": list(dd[:, 0]),
            "angle_cos": list(s[:, 2]), "angle_sin": list(s[:, 3]), "position": list(s[:, 4]), "positionD": list(s[:, 5]),
            "positionDD": list(dd[:, 1]), "Q_calculated": py(Q), "Q_applied": list(Qa), "Q_ccrc": list(block["Q_ccrc"][:, env]),
            "u": list(u_max * Qa), "target_position": py(block["target_position"][:, env]),
            "target_equilibrium": [int(x) for x in block["target_equilibrium"][:, env]], "L": py(block["L"][:, env]),
            "L_for_controller": told, "m_pole": m_pole, "m_pole_for_controller": told,
            "vertical_angle_offset": [0.0] * R if ao is None else py(ao[:, 0]),
            "vertical_angle_offset_cos": [1.0] * R if ao is None else py(ao[:, 1]),
            "vertical_angle_offset_sin": [0.0] * R if ao is None else py(ao[:, 2]),
            "Q_update_time": [None if r < block["first_update_row"] else float(q_update_time) for r in range(R)]}
    assert list(cols) == COLUMNS
    return cols


def write_recordings_native(paths, block, phys, header, title=None, q_update_time=0.0, n_threads=0):
    """All recordings of a batched run through libcpmppi's native writer (cpmppi_write_recordings: one thread per file, the
    reference's field formats reproduced byte for byte - tests/test_recording.py)."""
    import ctypes as C
    from . import _lib as _L
    lib = _L.load()
    R, E = block["states"].shape[:2]
    assert len(paths) == E
    want = dict(time=((R,), np.float64), states=((R, E, 6), f32), dd=((R, E, 2), f32), Q=((R, E), f32), Q_ccrc=((R, E), f32),
                target_position=((R, E), np.float64), target_equilibrium=((R, E), np.int32), L=((R, E), f32))
    arrs = {}
    for k, (shape, dt) in want.items():
        arrs[k] = a = np.ascontiguousarray(block[k], dtype=dt)
        if a.shape != shape:
            raise ValueError(f"recording block: {k} is {a.shape}, expected {shape}")
    rec = _L.cpmppi_recording()
    rec.E, rec.rows = E, R
    for k, a in arrs.items():
        setattr(rec, k, a.ctypes.data)
    rec.m_pole, rec.u_max = float(f32(phys.m_pole)), float(f32(phys.u_max))
    for k, field, dt, tail in (("m_pole", "m_pole_rows", f32, ()), ("informed", "informed", np.uint8, ()), ("Q_applied", "Q_applied", f32, ()),
                               ("angle_offset", "angle_offset", np.float64, (3,))):
        if block.get(k) is not None:
            arrs[k] = a = np.ascontiguousarray(block[k], dtype=dt)
            if a.shape != (R, E) + tail:
                raise ValueError(f"recording block: {k} is {a.shape}, expected {(R, E) + tail}")
            setattr(rec, field, a.ctypes.data)
    rec.first_update_row, rec.q_update_time = int(block["first_update_row"]), float(q_update_time)
    pre = preamble_bytes(header, title=title)
    cpaths = (C.c_char_p * E)(*[os.fsencode(p) for p in paths])
    rc = lib.cpmppi_write_recordings(cpaths, pre, len(pre), C.byref(rec), int(n_threads))
    if rc != 0:
        raise _L.CpmppiError(rc, lib.cpmppi_last_error(None).decode())
    return list(paths)


# ------------------------------------------------------------------------------------------------ the data generator
def dataset_paths(n, out_dir, ml_pipeline=False, split=(0.8, 0.1), secondary_experiment_index=None, digits=3):
    """The files run_data_generator gives its experiments (CartPole/data_generator.py:290-322 + csv_logger.py:61-91): all are
    called "Experiment" (or "Experiment-007" with a secondary index) and the logger makes the names unique by appending -1, -2, ...;
    in ML_Pipeline_mode they go to Train / Validate / Test by their position in the run."""
    paths, taken, start = [], set(), {}
    for i in range(n):
        if ml_pipeline:
            sub = "Train" if i < int(split[0] * n) else ("Validate" if i < int((split[0] + split[1]) * n) else "Test")
            folder, name = os.path.join(out_dir, sub), "Experiment"
        else:
            folder = out_dir
            name = "Experiment" if secondary_experiment_index is None else f"Experiment-{secondary_experiment_index:0{digits}d}"
        p = _unique_path(folder, name, taken, start)
        taken.add(p)
        paths.append(p)
    return paths


def experiment_folder(root, secondary_experiment_index=None, digits=3):
    """get_record_path (CartPole/data_generator.py:33-51): the first unused <root>/Experiment-[idx-]k, + "/Recordings"."""
    base = "Experiment-" + (f"{secondary_experiment_index:0{digits}d}-" if secondary_experiment_index is not None else "")
    k = 1
    while os.path.exists(os.path.join(root, base + str(k))):
        k += 1
    return os.path.join(root, base + str(k), "Recordings")


def generate_dataset(engine, num_experiments=None, out_dir=None, config=None, seed=None, cartpole_seed=None, L=None, native=True,
                     graph=False, secondary_experiment_index=None, controller_name="mpc", optimizer_name="mppi", title=None, groups=1,
                     rank=0, world=1, parameters=None, optimizer=None):
    """Batched run_data_generator: ``config`` = config_data_gen.yml as a dict (or overrides of the shipped file, see
    schedule.merged_config) - length_of_experiment, the three dt, the random initial state, the target trace's turning points
    and interpolation types, the target-equilibrium dwell times, number_of_experiments, ML_Pipeline_mode / split.  All
    experiments run at once on ``engine``'s GPU; one CSV each.  -> list of paths.
    ``seed`` overrides config['seed'] (the shipped file leaves it empty = clock); ``cartpole_seed`` seeds the per-experiment
    generators of the turning points (default: seed + 1).  ``groups`` > 1: the experiments run as that many independent env groups,
    each on its own stream (pipeline.py: 15-25 % more experiments per second for a few dozen envs; `engine` then only provides the
    problem definition).  ``rank`` / ``world``: one process per GPU, each generating its contiguous block of the run's experiments
    (schedule.draw_shard: the same experiments as the single-process run) and writing them like a job of the reference's array
    would - `secondary_experiment_index` defaults to the rank (run_data_generator.py -i, others/EulerClusterScripts/
    ParallelDataGeneration.sh:17), so no two ranks ever ask for the same file name; no collective is involved.
    ``parameters``: the `L` / `m_pole` / `inform_controller_about_parameters_change` blocks of cartpole_physical_parameters.yml's
    `cartpole:` section - a pole length and a pole mass that change DURING the experiments (CartPole/parameter_updater.py) and a
    controller that is told the true length only part of the time - its `controlDisturbance` / `controlBias` / `seed`: the additive
    control disturbance the reference's author collects training data with - and its measurement chain: `latency`, `noise`
    (noise_mode + the four sigmas), `vertical_angle_offset` (schedule.apply_parameter_schedule).
    ``optimizer``: one of the package's optimizer objects configured for the run's experiments (`controller_mpc(config_root=...,
    num_envs=n).configure().optimizer` - the shipped config_controllers.yml names rpgd) controls the plants instead of the fused MPPI
    step; `engine` may then be None (the optimizer's own engine runs the plant)."""
    import time
    from .harness import BatchedCartPoleExperiment
    from .schedule import RandomExperimentSetter, merged_config
    cfg = merged_config(config)
    if seed is not None:
        cfg["seed"] = int(seed)
    n_total = int(num_experiments if num_experiments is not None else cfg["number_of_experiments"])
    if cfg.get("seed") is None:
        raise ValueError("config['seed'] is empty: the reference then seeds from the clock; give a seed for a reproducible batch")
    cseed = cartpole_seed if cartpole_seed is not None else cfg["seed"] + 1
    if optimizer is not None:
        if int(groups) > 1 or graph:
            raise ValueError("an optimizer object is paced by the host: groups=1, graph=False")
        if getattr(optimizer, "engine", None) is None:
            from .shard import env_shard
            optimizer.configure(num_envs=env_shard(n_total, int(world), int(rank))[1])
        engine = optimizer.engine
        optimizer_name = getattr(optimizer, "optimizer_name", optimizer_name)
    _first = 0                                                     # global index of this process's first experiment (Philox keys)
    _stride = 1 if parameters and any(parameters.get(k) is not None for k in ("L", "m_pole", "inform_controller_about_parameters_change",
                                                                                "vertical_angle_offset")) else None
    if int(world) > 1:
        from .schedule import draw_shard
        batch, _first = draw_shard(cfg, n_total, cseed, rank, world, L=L, stride=_stride)
        if secondary_experiment_index is None:
            secondary_experiment_index = int(rank)
    else:
        batch = RandomExperimentSetter(cfg, track_half_length=engine.phys.TrackHalfLength).draw(n_total, cseed, L=L, stride=_stride)
    if parameters:
        from .schedule import apply_parameter_schedule
        batch = apply_parameter_schedule(batch, parameters, seed=cseed, first=_first)
    n = batch.E
    if n > engine.E:
        raise ValueError(f"{n} experiments on an engine created for {engine.E} envs")
    import torch
    t0 = time.perf_counter()
    if int(groups) > 1:
        from .pipeline import EnvGroups, run_schedule_groups
        eg = EnvGroups(n, engine.mppi, int(groups), engine.phys, engine.device.index, env_offset=_first)
        try:
            res = run_schedule_groups(eg, batch, cfg["seed"])
            torch.cuda.synchronize()
        finally:
            eg.close()
    else:
        exp = BatchedCartPoleExperiment(engine, batch.dt_simulation, batch.dt_control, seed=cfg["seed"])
        res = exp.run_schedule(batch, graph=graph, env_offset=_first, optimizer=optimizer)
        torch.cuda.synchronize()
    per_call = (time.perf_counter() - t0) / (batch.n_periods + 1)   # what Q_update_time can honestly say: wall time per controller update of the batch
    phys = engine.phys
    header = create_csv_header(cfg["length_of_experiment"], batch.dt_simulation, batch.dt_control, batch.dt_save, controller_name,
                               optimizer_name, phys)
    block = recording_block(res, phys)
    root = out_dir if out_dir is not None else cfg["PATH_TO_EXPERIMENT_RECORDINGS_DEFAULT"]
    if cfg.get("ML_Pipeline_mode"):
        root = experiment_folder(root, secondary_experiment_index)
    paths = dataset_paths(n, root, bool(cfg.get("ML_Pipeline_mode")), cfg["split"], secondary_experiment_index)
    if native:
        return write_recordings_native(paths, block, phys, header, title=title, q_update_time=per_call)
    for e in range(n):
        write_recording(paths[e], typed_columns(block, e, phys, per_call), title=title, header=header)
    return paths


def main(argv=None):
    """python -m cartpolesimulation_amd.recording --experiments 64 --length 10 --out ./Experiment_Recordings/ --seed 1
    python -m cartpolesimulation_amd.recording --config-root <CartPoleSimulation checkout> --seed 1     (its YAML files decide)"""
    import argparse
    from .configs import MPPIConfig, legacy_mppi_config, load_reference_yaml, mppi_config_from_yaml
    from .engine import MPPIEngine
    ap = argparse.ArgumentParser(description="Batched CartPole data generator on MI355X (reference: run_data_generator.py)")
    ap.add_argument("--config-root", default=None,
                    help="a CartPoleSimulation checkout: config_data_gen.yml (experiments, lengths, time scales, target traces, split), "
                         "cartpole_physical_parameters.yml (the plant's constants, parameter updaters, informer, control disturbance) and the "
                         "mppi section of Control_Toolkit_ASF/config_optimizers.yml + the mpc cost / predictor are read from it, as "
                         "run_data_generator.py reads them from its working directory; the flags below override them when given")
    ap.add_argument("--experiments", "--envs", type=int, default=None, dest="experiments")
    ap.add_argument("--length", type=float, default=None, help="length of each experiment in seconds")
    ap.add_argument("--out", default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--rollouts", type=int, default=None)
    ap.add_argument("--horizon", type=int, default=None)
    ap.add_argument("--dt-save", type=float, default=None)
    ap.add_argument("--ml-pipeline", action="store_true", default=None,
                    help="Train / Validate / Test folders (config_data_gen.yml: ML_Pipeline_mode)")
    ap.add_argument("-i", "--secondary_experiment_index", type=int, default=-1)
    ap.add_argument("--groups", type=int, default=1, help="independent env groups, each on its own stream (pipeline.py)")
    ap.add_argument("--optimizer", default=None,
                    help="mppi (the fused hot path; default without --config-root) or any other optimizer of the package - cem, cem-gmm, "
                         "rpgd, gradient, ... - paced by the host; with --config-root the default is the checkout's own "
                         "config_controllers.yml `mpc: optimizer` (shipped: rpgd)")
    ap.add_argument("--cost", default=None,
                    choices=["legacy_mppi_cartpole", "default", "quadratic_boundary_grad_minimal", "quadratic_boundary_grad"])
    args = ap.parse_args(argv)
    phys, parameters, dg, optimizer = None, None, {}, None
    opt_name = args.optimizer
    if args.config_root and opt_name is None:
        import yaml as _yaml
        with open(os.path.join(args.config_root, "Control_Toolkit_ASF", "config_controllers.yml")) as fh:
            opt_name = _yaml.safe_load(fh)["mpc"].get("optimizer") or "mppi"
    opt_name = opt_name or "mppi"
    if args.config_root:
        from .schedule import active_parameters
        import yaml
        phys, cfgs = load_reference_yaml(args.config_root)
        dg = dict(cfgs["data_gen"])
        if dg.get("controller", "mpc") != "mpc":
            raise SystemExit(f"config_data_gen.yml names controller {dg['controller']!r}: the batched generator runs the MPPI optimizer of `mpc`")
        over = {k: v for k, v in (("num_rollouts", args.rollouts), ("mpc_horizon", args.horizon), ("cost_function_specification", args.cost))
                if v is not None and v != "legacy_mppi_cartpole"}
        cfg = legacy_mppi_config(**{k: v for k, v in over.items() if k != "cost_function_specification"}) if args.cost == "legacy_mppi_cartpole" \
            else mppi_config_from_yaml(cfgs, **over)
        with open(os.path.join(args.config_root, "cartpole_physical_parameters.yml")) as fh:
            parameters = active_parameters(yaml.safe_load(fh)["cartpole"])
        if parameters and parameters.get("seed") is None and ("controlDisturbance" in parameters or "noise" in parameters):
            parameters["seed"] = args.seed                             # (the file's own `seed:` is empty = clock)
    else:
        n, h, cost = args.rollouts or 3500, args.horizon or 35, args.cost or "legacy_mppi_cartpole"
        cfg = legacy_mppi_config(num_rollouts=n, mpc_horizon=h) if cost == "legacy_mppi_cartpole" \
            else MPPIConfig(num_rollouts=n, mpc_horizon=h, cost_function_specification=cost)
        dg = dict(length_of_experiment=10.0, dt=dict(saving=0.02), number_of_experiments=64)
    if args.length is not None:
        dg["length_of_experiment"] = args.length
    if args.dt_save is not None:
        dg["dt"] = dict(dg.get("dt") or {}, saving=args.dt_save)
    if args.ml_pipeline is not None:
        dg["ML_Pipeline_mode"] = bool(args.ml_pipeline)
    seed = args.seed if args.seed is not None else (dg.get("seed") if dg.get("seed") is not None else 0)
    n_exp = args.experiments if args.experiments is not None else int(dg.get("number_of_experiments", 64))
    out = args.out if args.out is not None else dg.get("PATH_TO_EXPERIMENT_RECORDINGS_DEFAULT", "./Experiment_Recordings/")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))     # one process per GPU under torchrun
    from .shard import env_shard
    n_local, device = env_shard(n_exp, world, rank)[1], int(os.environ.get("LOCAL_RANK", "0"))
    eng = None
    if opt_name == "mppi":
        eng = MPPIEngine(n_local, cfg, phys=phys, device=device)
    else:                                                          # another optimizer of the package, built as controller_mpc builds it
        from .controller_mpc import controller_mpc
        over = {k: v for k, v in (("num_rollouts", args.rollouts), ("mpc_horizon", args.horizon), ("seed", seed)) if v is not None}
        if args.cost not in (None, "legacy_mppi_cartpole"):
            over["cost_function_specification"] = args.cost
        ctrl = controller_mpc("CartPole", {}, control_limits=([-1.0], [1.0]), config=over, phys=phys, device=device, num_envs=n_local,
                              config_root=args.config_root)
        ctrl.configure(opt_name)
        optimizer = ctrl.optimizer
    paths = generate_dataset(eng, n_exp, out, seed=seed, rank=rank, world=world, config=dg, groups=args.groups, parameters=parameters,
                             optimizer=optimizer,
                             secondary_experiment_index=None if args.secondary_experiment_index < 0 else args.secondary_experiment_index)
    print(f"wrote {len(paths)} recordings under {os.path.dirname(paths[0])}")


if __name__ == "__main__":
    main()
