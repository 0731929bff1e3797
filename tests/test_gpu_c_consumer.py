"""GPU: the drop-in boundary used from plain C.  tests/c_abi/consumer.c (gcc, C99; the HIP runtime's C API for device
buffers, include/cpmppi.h, nothing else) creates a handle from a config blob, runs fused MPPI steps and prints Q; the
same configuration through the Python binding must give the same numbers bit for bit — the library needs neither torch
nor C++ on the caller's side."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def test_c_consumer_matches_the_python_binding(tmp_path):
    torch = pytest.importorskip("torch")
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists(os.path.join(ROCM, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("needs gcc and the HIP runtime headers")
    from cartpolesimulation_amd.configs import MPPIConfig, build_c_config
    from cartpolesimulation_amd.engine import MPPIEngine
    pkg = os.path.join(ROOT, "cartpolesimulation_amd")
    exe = tmp_path / "consumer"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROCM, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "consumer.c"), "-o", str(exe), "-L", pkg, "-l:libcpmppi.so",
                    "-L", os.path.join(ROCM, "lib"), "-lamdhip64", f"-Wl,-rpath,{pkg}", f"-Wl,-rpath,{os.path.join(ROCM, 'lib')}"],
                   check=True)
    E, N, H, seed, steps = 3, 640, 25, 77, 4
    mppi = MPPIConfig(num_rollouts=N, mpc_horizon=H)
    blob = tmp_path / "config.bin"
    blob.write_bytes(bytes(build_c_config(E, mppi)))
    r = subprocess.run([str(exe), str(blob), str(E), str(seed), str(steps)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("cpmppi 1 gfx950")
    rows = lambda tag: [np.array([np.float32(x) for x in l.split()[1:]], dtype=np.float32)  # noqa: E731
                        for l in lines if l.split(" ", 1)[0] == tag]       # (RCCL prints its banner on stdout too)
    Q_c, Qh_c, Qg_c = (np.stack(rows(t)) for t in ("Q", "Qh", "Qg"))
    assert Q_c.shape == Qh_c.shape == Qg_c.shape == (steps, E)
    assert np.array_equal(Qh_c, Q_c)                                       # host-pointer entry point
    (u_c,), (g_c,) = rows("u"), rows("g")
    # the multi-GPU entry points from plain C, one rank: the same controls again, and the gathered copy of the result
    assert np.array_equal(Qg_c, Q_c) and np.array_equal(g_c, u_c)
    # ... with stamped blocks: the last gathered block carries the number of the last step-gather
    (st,) = [l.split() for l in lines if l.startswith("st ")]
    assert int(st[1]) == steps
    # the same through the Python binding
    eng = MPPIEngine(E, mppi)
    f32 = np.float32
    s0 = np.zeros((E, 6), f32)
    for e in range(E):
        th = f32(0.05) + f32(0.1) * f32(e)
        s0[e] = [th, f32(-0.2) * f32(e), f32(1.0) - f32(0.5) * th * th, th - th * th * th / f32(6.0), f32(0.01) * f32(e), 0.0]
    un = eng.zeros(E, H)
    Q_py = []
    for it in range(steps):
        Q, _ = eng.step(s0, un, np.full(E, 0.02, f32), np.ones(E, f32), L=np.full(E, 0.395, f32), seed=seed, offset=it)
        Q_py.append(Q.cpu().numpy().copy())
    assert np.array_equal(np.stack(Q_py), Q_c), (np.stack(Q_py), Q_c)
    assert np.array_equal(un.cpu().numpy()[0], u_c)
    assert np.abs(Q_c).max() > 1e-3
    # ---- the data generator's loop from C through the env-group entry points == the Python harness as ONE chain; the groups run
    # under ONE communicator with an all-gather of u_nom per period (cpmppi_groups_comm_init / cpmppi_groups_run_gather, one rank):
    # "gg <last gathered block == u_nom> <gathers enqueued> <ranks as RCCL reports them>"
    (gg,) = [l.split() for l in lines if l.startswith("gg ")]
    assert gg[1:] == ["1", str(steps + 3 + 1), "1"], gg
    (R_c,), (Qc_c,), (D_c,) = rows("R"), rows("Qc"), rows("D")
    T, n_ctrl, n_save, stride = steps + 3, 10, 5, 5
    srows = T * n_ctrl // stride + 1
    r = np.arange(srows, dtype=np.uint32)[:, None]
    e = np.arange(E, dtype=np.uint32)[None, :]
    tp_tab = (f32(0.002) * ((r * 7 + e * 3) % 40).astype(f32) - f32(0.04)).astype(f32)
    te_tab = np.where(((r // 6 + e) % 2) != 0, f32(-1.0), f32(1.0)).astype(f32)
    s = eng.tensor(s0.copy())
    un2, Q2 = eng.zeros(E, H), eng.empty(E)
    tp_d, te_d = eng.tensor(tp_tab), eng.tensor(te_tab)
    cur_tp, cur_te = tp_d[0].clone(), te_d[0].clone()
    Lv = eng.tensor(np.full(E, 0.395, f32))
    R = T * n_ctrl // n_save + 1
    states, dd, Qlog = eng.zeros(R, E, 6), eng.zeros(R, E, 2), eng.zeros(T + 1, E)
    states[0] = s
    kw = dict(dt_sim=0.002, period_steps=n_ctrl, L=Lv, states_log=states, dd_log=dd, save_every=n_save, Q_log=Qlog, target_position_table=tp_d,
              target_equilibrium_table=te_d, sched_stride=stride, target_position_out=cur_tp, target_equilibrium_out=cur_te)
    for c in range(T + 1):
        eng.step(s, un2, cur_tp, cur_te, L=Lv, seed=seed, offset=c, Q_out=Q2)
        eng.plant_step(s, Q2, n_ctrl if c < T else 0, period=c, **kw)
    torch.cuda.synchronize()
    assert np.array_equal(states[-1].cpu().numpy().reshape(-1), R_c) and np.array_equal(Qlog[T].cpu().numpy(), Qc_c)
    assert np.array_equal(dd[-1].cpu().numpy().reshape(-1), D_c) and np.abs(Qc_c).max() > 1e-3
