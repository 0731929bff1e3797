"""Golden vectors at the SHAPE of BASELINE configs[2] / configs[3] ("SFC64 subset (E = 2) for parity"),
produced by the reference's own in-tree code under the import stand-ins - VERDICT r5, missing #3: the oracle that grades C3
(H = 100) and C4 (a pole length per env) was pinned to the reference only at H <= 50 with the default length.

TEST INFRASTRUCTURE; usage:  cd /root/reference && python -B /root/repo/oracle/gen_golden_c3c4.py

Per case (c3: H = 100, envs 0, 1 of the bench's synthetic inputs for seed 2; c4: H = 50, envs 0, 1 for seed 3; N = 512):
  * SI_Toolkit_ASF/ToolkitCustomization/predictors_customization_v0.py::next_state_predictor_ODE_v0.step, H times, with the env's
    pole length handed over the way the simulator does it - `variable_parameters.L` (:47-54) - -> trajectories [N, H + 1, 6] (mode A:
    the float32 arithmetic of the shipped code under numpy >= 2); mode B = the reference's own substep function fed float64
    (SURVEY.md H1), float32 store per control step
  * Control_Toolkit_ASF/Cost_Functions/CartPole/quadratic_boundary_grad_minimal.py::get_trajectory_cost on those trajectories, and
    the `default` plugin's stage sum + terminal cost (default.py:23-88)
  * Control_Toolkit_ASF/Controllers/controller_mppi_cartpole.py::initialize_perturbations ("interpolated", SFC64 knots: P = 11 at
    H = 100) and ::reward_weighted_average -> the soft-min update
for the perturbed inputs as sampled ("raw") and clipped to [-1, 1] ("clip", what the product's default control_mode does).
"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (installs the stand-ins, imports the reference modules; its __main__ does not run)

f32 = np.float32
OUT = G.OUT
N = 512
CASES = {"c3": dict(H=100, seed=2), "c4": dict(H=50, seed=3)}


def synthetic_inputs_np(E, seed):
    """bench.py::synthetic_inputs without torch: SURVEY.md 8(d) - s0 as data_generator.py:221-256 / config_data_gen.yml:14-18, a
    target and a pole length per env (cartpole_physical_parameters.yml:37)."""
    rng = np.random.Generator(np.random.SFC64(seed))
    THL = 0.198
    angle = np.where(rng.uniform(size=E) > 0.5, 1.0, -1.0) * rng.uniform(0.0, 180.0, E) * np.pi / 180.0
    s0 = np.zeros((E, 6), dtype=f32)
    s0[:, 0] = angle
    s0[:, 1] = rng.uniform(-1, 1, E) * 1200.0 * np.pi / 180.0
    s0[:, 2], s0[:, 3] = np.cos(angle), np.sin(angle)
    s0[:, 4] = rng.uniform(-1, 1, E) * THL * 0.8
    s0[:, 5] = rng.uniform(-1, 1, E) * THL * 0.5
    tp = (rng.uniform(-0.8, 0.8, E) * THL).astype(f32)
    L = rng.uniform(0.2, 0.5, E).astype(f32)
    return s0, tp, L


def main():
    lib = G.ref_shims.NumpyLibrary()
    out = {"N": np.int64(N), "stdev": np.float64(0.03 / np.sqrt(G.DT)), "cases": np.array(list(CASES))}
    for ci, (case, spec) in enumerate(CASES.items()):
        H = spec["H"]
        s0_all, tp_all, L_all = synthetic_inputs_np(64, spec["seed"])
        out[f"{case}/H"], out[f"{case}/input_seed"] = np.int64(H), np.int64(spec["seed"])
        for e in range(2):
            key = f"{case}/{e}"
            s0, target, Lv = s0_all[e], tp_all[e], L_all[e]
            seed = 4000 + 10 * ci + e
            ctrl = G.make_legacy_controller(seed, N, H, target)
            delta_u = ctrl.initialize_perturbations(stdev=0.03 / np.sqrt(G.DT), sampling_type="interpolated")
            u_nom = np.zeros(H, dtype=f32)
            vp_pred = SimpleNamespace(L=np.asarray(Lv, dtype=f32))
            pred = G.next_state_predictor_ODE_v0(G.DT, G.S_SUB, N, variable_parameters=vp_pred)
            vp_cost = SimpleNamespace(target_position=f32(target), target_equilibrium=f32(1.0))
            qbgm = G.quadratic_boundary_grad_minimal(vp_cost, lib)
            dflt = G.default_cost(vp_cost, lib)
            P = pred.cpe.params
            for tag, u_run in (("raw", (u_nom + delta_u).astype(f32)), ("clip", np.clip(u_nom + delta_u, f32(-1), f32(1)).astype(f32))):
                traj = np.zeros((N, H + 1, 6), dtype=f32)                   # predict_core: out[:, k + 1] = step(out[:, k], Q[:, k])
                traj[:, 0] = np.tile(s0, (N, 1))
                for k in range(H):
                    traj[:, k + 1] = pred.step(traj[:, k], u_run[:, k, None])
                S = np.asarray(qbgm.get_trajectory_cost(traj, u_run[..., None], None), dtype=f32)
                stage_d = dflt._get_stage_cost(traj[:, :-1], u_run[..., None], None)          # (as gen_golden.gen_rollouts: stage sum + terminal)
                S_d = np.asarray(np.sum(stage_d, 1) + dflt.get_terminal_cost(traj[:, -1])[:, 0], dtype=f32)
                sB = np.tile(s0, (N, 1))
                u_phys = pred.cpe.Q2u(u_run)
                for k in range(H):
                    sB = G.ref_step_mode_B(sB, u_phys[:, k], vp_pred.L, P, 1)
                rwa = G.LEG.reward_weighted_average(S, delta_u)
                u_new = (u_nom + rwa).astype(f32)
                if tag == "clip":
                    u_new = np.clip(u_new, f32(-1), f32(1))
                out[f"{key}/{tag}/final"], out[f"{key}/{tag}/final_B"] = traj[:, -1], sB
                out[f"{key}/{tag}/traj_head"] = traj[:8]
                out[f"{key}/{tag}/S_qbgm"], out[f"{key}/{tag}/u_new"] = S, u_new
                out[f"{key}/{tag}/S_default"] = S_d
                print(f"{key}/{tag}: L={float(Lv):.4f} S[min,max]=({S.min():.3f},{S.max():.3f}) max|final A-B|={np.abs(traj[:, -1] - sB).max():.2e} "
                      f"|u_new|max={np.abs(u_new).max():.4f}")
            out[f"{key}/s0"], out[f"{key}/target"], out[f"{key}/L"], out[f"{key}/seed"] = s0, f32(target), f32(Lv), np.int64(seed)
            out[f"{key}/delta_u_head"] = delta_u[:4]
            out[f"{key}/delta_u_sum64"] = np.float64(delta_u.astype(np.float64).sum())
    np.savez_compressed(os.path.join(OUT, "rollouts_c3c4.npz"), **out)
    print("written", os.path.join(OUT, "rollouts_c3c4.npz"))


if __name__ == "__main__":
    main()
