"""Generate the golden vectors under tests/golden/ by EXECUTING THE REFERENCE'S OWN IN-TREE CODE.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference); the GPU box never runs it and
never sees the reference.  Usage:   cd /root/reference && python -B /root/repo/oracle/gen_golden.py

Every *expected output* stored in a fixture is produced by a reference code object:
  * CartPole/cartpole_numba.py::cartpole_fine_integration_numba_interface / cartpole_fine_integration_numba
  * SI_Toolkit_ASF/ToolkitCustomization/predictors_customization_v0.py::next_state_predictor_ODE_v0.step
  * Control_Toolkit_ASF/Cost_Functions/CartPole/{quadratic_boundary_grad_minimal,default}.py
  * Control_Toolkit_ASF/Controllers/controller_mppi_cartpole.py (q, phi, trajectory_rollouts,
    reward_weighted_average, update_inputs, initialize_perturbations, controller_mppi_cartpole.step)
  * CartPole/cartpole_equations.py (plant: euler-cromer integration, edge_bounce, ode) for the closed-loop trace
The stand-ins in oracle/ref_shims.py only make those modules importable.
Inputs are either stored, or regenerable from a numpy SFC64 seed (numpy is available on the GPU box).
"""
import hashlib
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
os.chdir(ref_shims.REFERENCE_ROOT)           # the reference opens its YAML files relative to cwd

from numpy.random import SFC64, Generator  # noqa: E402
from CartPole.cartpole_numba import cartpole_fine_integration_numba, cartpole_fine_integration_numba_interface  # noqa: E402
from CartPole.cartpole_equations import (CartPoleEquations, cartpole_integration_euler_cromer_numba,  # noqa: E402
                                         edge_bounce_numba, _cartpole_ode_numba)
from CartPole._CartPole_mathematical_helpers import wrap_angle_rad as ref_wrap_angle_rad  # noqa: E402
from CartPole.state_utilities import (ANGLE_IDX, ANGLED_IDX, ANGLE_COS_IDX, ANGLE_SIN_IDX, POSITION_IDX,  # noqa: E402
                                      POSITIOND_IDX, STATE_VARIABLES, create_cartpole_state)
from SI_Toolkit_ASF.ToolkitCustomization.predictors_customization_v0 import next_state_predictor_ODE_v0  # noqa: E402
from Control_Toolkit_ASF.Cost_Functions.CartPole.quadratic_boundary_grad_minimal import quadratic_boundary_grad_minimal  # noqa: E402
from Control_Toolkit_ASF.Cost_Functions.CartPole.default import default as default_cost  # noqa: E402
import Control_Toolkit_ASF.Controllers.controller_mppi_cartpole as LEG  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
f32 = np.float32
DT, S_SUB = 0.02, 10


def provenance():
    """sha256 of the reference files whose code produced the fixtures."""
    files = ["CartPole/cartpole_numba.py", "CartPole/cartpole_equations.py", "CartPole/_CartPole_mathematical_helpers.py",
             "CartPole/state_utilities.py", "CartPole/cartpole_parameters.py", "cartpole_physical_parameters.yml",
             "SI_Toolkit_ASF/ToolkitCustomization/predictors_customization_v0.py",
             "Control_Toolkit_ASF/Cost_Functions/CartPole/quadratic_boundary_grad_minimal.py",
             "Control_Toolkit_ASF/Cost_Functions/CartPole/default.py",
             "Control_Toolkit_ASF/Controllers/controller_mppi_cartpole.py",
             "Control_Toolkit_ASF/config_cost_function.yml", "Control_Toolkit_ASF/config_controllers.yml",
             "Control_Toolkit_ASF/config_optimizers.yml", "SI_Toolkit_ASF/config_predictors.yml"]
    lines = []
    for f in files:
        with open(f, "rb") as fh:
            lines.append(f"{hashlib.sha256(fh.read()).hexdigest()}  {f}")
    return "\n".join(lines)


def fill_trig(s):
    s[:, ANGLE_COS_IDX] = np.cos(s[:, ANGLE_IDX])
    s[:, ANGLE_SIN_IDX] = np.sin(s[:, ANGLE_IDX])
    return s


def kat_states(rng, n):
    """Diverse single-step inputs: benign, fast-spinning, near +-pi (wrap), near the track edge (bounce)."""
    THL = 0.198
    s = np.zeros((n, 6), dtype=f32)
    s[:, ANGLE_IDX] = rng.uniform(-np.pi, np.pi, n)
    s[:, ANGLED_IDX] = rng.uniform(-1, 1, n) * 20.0
    s[:, POSITION_IDX] = rng.uniform(-1, 1, n) * THL * 0.8
    s[:, POSITIOND_IDX] = rng.uniform(-1, 1, n) * 0.5
    q = n // 8
    # wrap cases: angle within 0.03 rad of +-pi moving outward
    s[:q, ANGLE_IDX] = np.sign(rng.uniform(-1, 1, q)) * (np.pi - rng.uniform(0, 0.03, q))
    s[:q, ANGLED_IDX] = np.sign(s[:q, ANGLE_IDX]) * rng.uniform(2, 25, q)
    # bounce cases: within 1 mm of the edge moving outward
    s[q:2 * q, POSITION_IDX] = np.sign(rng.uniform(-1, 1, q)) * (THL - rng.uniform(0, 1e-3, q))
    s[q:2 * q, POSITIOND_IDX] = np.sign(s[q:2 * q, POSITION_IDX]) * rng.uniform(0.1, 1.0, q)
    # already outside the track (bounces every substep until back inside)
    s[2 * q:2 * q + 8, POSITION_IDX] = np.sign(rng.uniform(-1, 1, 8)) * (THL + rng.uniform(0, 5e-3, 8))
    return fill_trig(s)


def ref_step_mode_B(s, u, L, params, n_steps=1):
    """Mode B: the reference's own substep loop fed float64 arrays, float32 store per control step."""
    s = s.copy()
    for _ in range(n_steps):
        a, ad, x, xd, ca, sa = cartpole_fine_integration_numba(
            angle=s[:, ANGLE_IDX].astype(np.float64), angleD=s[:, ANGLED_IDX].astype(np.float64),
            angle_cos=s[:, ANGLE_COS_IDX].astype(np.float64), angle_sin=s[:, ANGLE_SIN_IDX].astype(np.float64),
            position=s[:, POSITION_IDX].astype(np.float64), positionD=s[:, POSITIOND_IDX].astype(np.float64),
            u=u.astype(np.float64), t_step=DT / S_SUB, intermediate_steps=S_SUB, L=L,
            k=params.k, m_cart=params.m_cart, m_pole=params.m_pole, g=params.g, J_fric=params.J_fric,
            M_fric=params.M_fric)
        nxt = np.zeros_like(s)
        nxt[:, ANGLE_IDX], nxt[:, ANGLED_IDX], nxt[:, POSITION_IDX], nxt[:, POSITIOND_IDX] = a, ad, x, xd
        nxt[:, ANGLE_COS_IDX], nxt[:, ANGLE_SIN_IDX] = ca, sa
        s = nxt
    return s


def gen_kat():
    rng = Generator(SFC64(20240711))
    n = 256
    s = kat_states(rng, n)
    Q = rng.uniform(-1, 1, n).astype(f32)
    Ls = np.array([0.395, 0.2, 0.5, 0.31], dtype=f32)
    L_per_row = np.repeat(Ls, n // 4)
    cpe = CartPoleEquations()
    sub1 = np.zeros_like(s); step1 = np.zeros_like(s); step2 = np.zeros_like(s)
    step1_B = np.zeros_like(s); step2_B = np.zeros_like(s)
    for i, Lv in enumerate(Ls):
        sl = slice(i * (n // 4), (i + 1) * (n // 4))
        vp = SimpleNamespace(L=np.asarray(Lv, dtype=f32))
        pred = next_state_predictor_ODE_v0(DT, S_SUB, n // 4, variable_parameters=vp)
        u = pred.cpe.Q2u(Q[sl])
        sub1[sl] = cartpole_fine_integration_numba_interface(s[sl], u, pred.t_step, 1, pred.cpe.params, L=vp.L)
        step1[sl] = pred.step(s[sl], Q[sl, None])
        step2[sl] = pred.step(step1[sl], Q[sl, None])
        step1_B[sl] = ref_step_mode_B(s[sl], u, vp.L, pred.cpe.params, 1)
        step2_B[sl] = ref_step_mode_B(s[sl], u, vp.L, pred.cpe.params, 2)
    P = cpe.params
    np.savez_compressed(
        os.path.join(OUT, "kat_step.npz"), s_in=s, Q_in=Q, L_in=L_per_row, sub1_A=sub1, step1_A=step1, step2_A=step2,
        step1_B=step1_B, step2_B=step2_B,
        params=np.array([P.k, P.m_cart, P.m_pole, P.g, P.J_fric, P.M_fric, P.L, P.u_max, P.TrackHalfLength], dtype=f32),
        state_variables=np.array(list(STATE_VARIABLES)))
    moved = np.abs(sub1[:, POSITIOND_IDX] + s[:, POSITIOND_IDX]) < np.abs(s[:, POSITIOND_IDX]) * 0.5
    print(f"kat_step: n={n}, bounced in first substep ~{int(moved.sum())}, max|A-B| step2 = {np.abs(step2 - step2_B).max():.3e}")


REGIMES = {  # SURVEY.md §8(d) C2 regimes: (angle, angleD, position, positionD), target_position
    "upright": ((0.05, 0.0, 0.0, 0.0), 0.0),
    "hanging": ((3.0, 0.0, 0.1, 0.0), 0.0),
    "near_edge": ((0.5, 2.0, 0.18, 0.4), 0.05),
    "fast": ((1.5, 15.0, -0.1, -0.3), 0.05),
}


def configure_legacy(N, H):
    LEG.num_rollouts = N
    LEG.mpc_horizon = H
    LEG.predictor.configure(batch_size=N, horizon=H, dt=DT)


def make_legacy_controller(seed, N, H, target_position):
    configure_legacy(N, H)
    LEG.config_mppi_cartpole["seed"] = seed
    # cost-weight noise is 0.0 in the shipped YAML (config_controllers.yml:22) so the module-level weights stay put
    ctrl = LEG.controller_mppi_cartpole("CartPole", {"target_position": f32(target_position),
                                                     "target_equilibrium": f32(1.0)},
                                        (np.array([-1.0], dtype=f32), np.array([1.0], dtype=f32)))
    ctrl.configure()
    return ctrl


def gen_rollouts(N=1024, H=50, keep=32):
    lib = ref_shims.NumpyLibrary()
    out = {}
    rng_extra = Generator(SFC64(1))
    regimes = dict(REGIMES)
    for j in range(4):                      # random initial states as data_generator.py:221-256 / config_data_gen.yml:14-18
        THL = 0.198
        st = (float(np.sign(rng_extra.uniform(-1, 1)) * rng_extra.uniform(0, 180) * np.pi / 180),
              float(rng_extra.uniform(-1, 1) * 1200 * np.pi / 180), float(rng_extra.uniform(-1, 1) * THL * 0.8),
              float(rng_extra.uniform(-1, 1) * THL * 0.5))
        regimes[f"random{j}"] = (st, float(rng_extra.uniform(-0.8, 0.8) * THL))
    names = list(regimes)
    for r, name in enumerate(names):
        (angle, angleD, position, positionD), target = regimes[name]
        s0 = create_cartpole_state(dict(angle=angle, angleD=angleD, position=position, positionD=positionD))
        seed = 1234 + r
        ctrl = make_legacy_controller(seed, N, H, target)
        delta_u = ctrl.initialize_perturbations(stdev=0.03 / np.sqrt(DT), sampling_type="interpolated")
        # warm-started nominal for half the regimes, zeros for the others
        u_nom = (0.3 * np.sin(np.arange(H) / 7.0)).astype(f32) if r % 2 else np.zeros(H, dtype=f32)
        u_prev = np.roll(u_nom, 1).astype(f32)
        vp = SimpleNamespace(target_position=f32(target), target_equilibrium=f32(1.0))
        qbgm = quadratic_boundary_grad_minimal(vp, lib)
        dflt = default_cost(vp, lib)
        for tag, u_run in (("raw", (u_nom + delta_u).astype(f32)),
                           ("clip", np.clip(u_nom + delta_u, f32(-1), f32(1)).astype(f32))):
            traj = LEG.predictor.predict(np.tile(s0, (N, 1)), u_run[..., np.newaxis])
            S_qbgm = qbgm.get_trajectory_cost(traj, u_run[..., np.newaxis], None)
            stage_d = dflt._get_stage_cost(traj[:, :-1], u_run[..., np.newaxis], None)
            S_default = np.sum(stage_d, 1) + dflt.get_terminal_cost(traj[:, -1])[:, 0]
            out[f"{name}/{tag}/traj_head"] = traj[:keep]
            out[f"{name}/{tag}/final"] = traj[:, -1]
            out[f"{name}/{tag}/S_qbgm"] = np.asarray(S_qbgm, dtype=f32)
            out[f"{name}/{tag}/S_default"] = np.asarray(S_default, dtype=f32)
            out[f"{name}/{tag}/stage_qbgm_head"] = qbgm.get_stage_cost(traj[:keep, :-1], u_run[:keep, :, None], None)
        # mode B trajectories (raw inputs)
        u_phys = CartPoleEquations().Q2u((u_nom + delta_u).astype(f32))
        sB = np.tile(s0, (N, 1))
        P = CartPoleEquations().params
        for kk in range(H):
            sB = ref_step_mode_B(sB, u_phys[:, kk], P.L, P, 1)
        out[f"{name}/raw/final_B"] = sB
        # legacy: rollouts + q + phi + softmin update, all reference code
        ctrl.variable_parameters.target_position = f32(target)
        S_leg = LEG.trajectory_rollouts(s0, np.zeros(N, dtype=f32), u_nom, delta_u, u_prev, f32(target))
        u_leg = u_nom.copy()
        LEG.update_inputs(u_leg, S_leg, delta_u)
        out[f"{name}/S_legacy"] = np.asarray(S_leg)
        out[f"{name}/u_new_legacy"] = u_leg
        # a16 on the plugin cost + correction (a15 algebra of :261-263 evaluated with the reference's own numpy ops)
        out[f"{name}/rwa_qbgm_raw"] = LEG.reward_weighted_average(out[f"{name}/raw/S_qbgm"], delta_u)
        out[f"{name}/s0"] = s0
        out[f"{name}/target"] = f32(target)
        out[f"{name}/seed"] = np.int64(seed)
        out[f"{name}/u_nom"] = u_nom
        out[f"{name}/u_prev"] = u_prev
        out[f"{name}/delta_u_head"] = delta_u[:4]
        out[f"{name}/delta_u_sum64"] = np.float64(delta_u.astype(np.float64).sum())
        print(f"rollouts[{name}]: S_qbgm[min,max]=({out[f'{name}/raw/S_qbgm'].min():.3f},{out[f'{name}/raw/S_qbgm'].max():.3f}) "
              f"S_legacy min={S_leg.min():.3f} max|final A-B|={np.abs(out[f'{name}/raw/final'] - sB).max():.2e}")
    out["names"] = np.array(names)
    out["N"], out["H"], out["stdev"] = np.int64(N), np.int64(H), np.float64(0.03 / np.sqrt(DT))
    np.savez_compressed(os.path.join(OUT, "rollouts_c2.npz"), **out)


def gen_legacy_steps():
    """Complete controller_mppi_cartpole.step traces: (seed, s sequence) -> per call (Q, updated u, S stats)."""
    for (N, H) in ((256, 20), (1024, 50)):
        ctrl = make_legacy_controller(1234, N, H, 0.05)
        rng = Generator(SFC64(99))
        s_seq, Qs, us, S_all, du_sums = [], [], [], [], []
        s = create_cartpole_state(dict(angle=0.3, angleD=-1.0, position=0.02, positionD=0.1))
        for it in range(3):
            s_seq.append(s.copy())
            Q = ctrl.step(s.copy(), time=it * DT, updated_attributes={"target_position": f32(0.05),
                                                                      "target_equilibrium": f32(1.0)})
            Qs.append(Q); us.append(ctrl.u_prev.copy()); S_all.append(np.asarray(ctrl.S_tilde_k, dtype=np.float64).copy())
            du_sums.append(ctrl.delta_u.astype(np.float64).sum())
            s = create_cartpole_state(dict(angle=float(s[ANGLE_IDX] + rng.uniform(-0.05, 0.05)),
                                           angleD=float(s[ANGLED_IDX] + rng.uniform(-0.5, 0.5)),
                                           position=float(s[POSITION_IDX] + rng.uniform(-0.005, 0.005)),
                                           positionD=float(s[POSITIOND_IDX] + rng.uniform(-0.05, 0.05))))
        np.savez_compressed(os.path.join(OUT, f"legacy_step_{N}x{H}.npz"), seed=np.int64(1234), N=np.int64(N),
                            H=np.int64(H), s_seq=np.array(s_seq), Q=np.array(Qs, dtype=f32), u_updated=np.array(us),
                            S=np.array(S_all), delta_u_sum64=np.array(du_sums), target=f32(0.05),
                            stdev=np.float64(LEG.SQRTRHODTINV), p_Q=np.float64(LEG.p_Q))
        print(f"legacy_step {N}x{H}: Q={np.array(Qs)}")


def gen_closed_loop(N=256, H=20, n_control=50):
    """C1 plumbing trace (SURVEY.md §8b harness row): legacy controller + in-tree plant functions, noise OFF."""
    rng0 = Generator(SFC64(0))                         # s0 per §8(d) C1
    THL = 0.198
    x0 = rng0.uniform(-1, 1) * THL * 0.8
    v0 = rng0.uniform(-1, 1) * THL * 0.5
    th0 = np.sign(rng0.uniform(-1, 1)) * rng0.uniform(0, 180) * np.pi / 180
    w0 = rng0.uniform(-1, 1) * 1200 * np.pi / 180
    s = create_cartpole_state(dict(angle=th0, angleD=w0, position=x0, positionD=v0))
    ctrl = make_legacy_controller(1234, N, H, 0.0)
    cpe = CartPoleEquations(numba_compiled=True)
    P = cpe.params
    Lf = float(P.L)
    attrs = {"target_position": f32(0.0), "target_equilibrium": f32(1.0), "L": Lf}
    dt_sim = 0.002
    s_log, Q_log, minS_log, u_log = [], [], [], []
    t = 0.0
    for c in range(n_control):
        s_log.append(s.copy())
        Q = ctrl.step(s.copy(), t, attrs)
        Q_log.append(Q); minS_log.append(float(np.min(ctrl.S_tilde_k))); u_log.append(ctrl.u_prev.copy())
        u = cpe.Q2u(Q)
        angleDD, positionDD = cpe.cartpole_ode_interface(s, u, L=Lf)
        for _ in range(10):                            # CartPole/__init__.py:283-324 with dt_control/dt_sim = 10
            t += dt_sim
            (s[ANGLE_IDX], s[ANGLED_IDX], s[POSITION_IDX], s[POSITIOND_IDX]) = cartpole_integration_euler_cromer_numba(
                s[ANGLE_IDX], s[ANGLED_IDX], angleDD, s[POSITION_IDX], s[POSITIOND_IDX], positionDD, dt_sim)
            (s[ANGLE_IDX], s[ANGLED_IDX], s[POSITION_IDX], s[POSITIOND_IDX]) = edge_bounce_numba(
                s[ANGLE_IDX], np.cos(s[ANGLE_IDX]), s[ANGLED_IDX], s[POSITION_IDX], s[POSITIOND_IDX], dt_sim, L=Lf)
            s[ANGLE_COS_IDX] = np.cos(s[ANGLE_IDX]); s[ANGLE_SIN_IDX] = np.sin(s[ANGLE_IDX])
            s[ANGLE_IDX] = ref_wrap_angle_rad(s[ANGLE_IDX])
            angleDD, positionDD = cpe.cartpole_ode_interface(s, u, L=Lf)
    np.savez_compressed(os.path.join(OUT, "closed_loop_c1.npz"), s=np.array(s_log), Q=np.array(Q_log, dtype=f32),
                        minS=np.array(minS_log), u_updated=np.array(u_log), N=np.int64(N), H=np.int64(H),
                        seed=np.int64(1234), target=f32(0.0), stdev=np.float64(LEG.SQRTRHODTINV),
                        p_Q=np.float64(LEG.p_Q))
    print(f"closed_loop: final |angle|={abs(s[ANGLE_IDX]):.3f} position={s[POSITION_IDX]:.4f}; Q[:5]={np.array(Q_log[:5])}")


def gen_sampler_and_rwa():
    ctrl = make_legacy_controller(1234, 3500, 35, 0.0)
    du = ctrl.initialize_perturbations(stdev=0.02 / np.sqrt(DT), sampling_type="interpolated")
    old = LEG.LBD
    S_kat = np.array([10, 12, 9, 30], dtype=f32)
    du_kat = np.array([[0.1, -0.2, 0.3], [0, 0.5, -0.5], [-0.3, 0.1, 0.2], [1, 1, 1]], dtype=f32)
    r100 = LEG.reward_weighted_average(S_kat, du_kat)
    LEG.LBD = 1.0
    r1 = LEG.reward_weighted_average(S_kat, du_kat)
    LEG.LBD = old
    np.savez_compressed(os.path.join(OUT, "sampler_rwa.npz"), du_row0=du[0], du_row3499=du[-1],
                        du_sum64=np.float64(du.astype(np.float64).sum()), N=np.int64(3500), H=np.int64(35),
                        seed=np.int64(1234), stdev=np.float64(0.02 / np.sqrt(DT)), S_kat=S_kat, du_kat=du_kat,
                        rwa_lbd100=r100, rwa_lbd1=r1)
    print("sampler: du[0,:4] =", du[0, :4], " rwa100 =", r100)


if __name__ == "__main__":
    gen_kat()
    gen_rollouts()
    gen_legacy_steps()
    gen_closed_loop()
    gen_sampler_and_rwa()
    with open(os.path.join(OUT, "PROVENANCE.txt"), "w") as fh:
        fh.write("Golden vectors generated by oracle/gen_golden.py executing the reference's in-tree code.\n"
                 f"numpy {np.__version__}; arithmetic mode A = strict float32 (numpy>=2 NEP-50 weak scalars).\n"
                 "sha256 of the reference files whose code objects produced the expected outputs:\n" + provenance() + "\n")
    print("golden vectors written to", OUT)
