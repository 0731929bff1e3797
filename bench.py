#!/usr/bin/env python3
"""bench.py — MPPI rollouts/s of the fused HIP hot path on MI355X (BASELINE.json metric).

A "step" = one full MPPI optimizer step for a batch of E independent problem instances (envs) of the C2 shape
(1024 samples x 50-step horizon x 10 Euler substeps, BASELINE.json configs[1]): perturbation sampling (a17) +
rollout (a3-a11) + cost (a12, a15) + importance-weighted update (a16) + shift/clip (a18).  One rollout = one sampled
control sequence integrated over the horizon + its cost + its share of the update.  Inputs are synthetic and resident
in HBM before the timed region.  With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank owns
E envs (weak scaling, no data-path collective) and the chosen control sequences are gathered with ONE RCCL all-gather
per step.

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F16_MFMA_PEAK_TFLOPS = 2500.0      # dense f16/bf16 MFMA, MI355X_MICROARCH.md
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector


def algorithmic_bytes_per_rollout(N, H):
    """delta_u read once (4H), S written once (4), per-env vectors amortised over N."""
    return 4.0 * H + 4.0 + (24.0 + 8.0 * H + 12.0) / N


def algorithmic_flops_per_rollout(H, S=10):
    """38 algebraic flops per substep + ~30 per control step for cost/correction/update."""
    return H * (S * 38.0 + 30.0) + 2.0 * H


def synthetic_inputs(E, H, seed, device):
    """s0 as data_generator.py:221-256 / config_data_gen.yml:14-18; targets and L per env."""
    import torch
    rng = np.random.Generator(np.random.SFC64(seed))
    THL = 0.198
    angle = np.where(rng.uniform(size=E) > 0.5, 1.0, -1.0) * rng.uniform(0.0, 180.0, E) * np.pi / 180.0
    s0 = np.zeros((E, 6), dtype=np.float32)
    s0[:, 0] = angle
    s0[:, 1] = rng.uniform(-1, 1, E) * 1200.0 * np.pi / 180.0
    s0[:, 2], s0[:, 3] = np.cos(angle), np.sin(angle)
    s0[:, 4] = rng.uniform(-1, 1, E) * THL * 0.8
    s0[:, 5] = rng.uniform(-1, 1, E) * THL * 0.5
    tp = (rng.uniform(-0.8, 0.8, E) * THL).astype(np.float32)
    te = np.ones(E, dtype=np.float32)
    L = rng.uniform(0.2, 0.5, E).astype(np.float32)
    t = lambda a: torch.as_tensor(a, device=device)
    return t(s0), t(tp), t(te), t(L)


def cpu_baseline(N, H, budget_s=12.0):
    """The plain-C oracle (validated against the golden vectors) timed on this host's cores: same step, same shape."""
    from oracle import oracle_np as O
    from oracle import oracle_c as OC
    cfg = O.MPPIConfig(N=N, H=H)
    c = OC.make_config(cfg)
    threads = OC.max_threads()
    rng = np.random.Generator(np.random.SFC64(4))

    def run(E):
        s0 = np.stack([O.create_cartpole_state(rng.uniform(-3, 3), rng.uniform(-5, 5), rng.uniform(-0.15, 0.15),
                                               rng.uniform(-0.3, 0.3)) for _ in range(E)])
        du = (cfg.stdev * rng.standard_normal((E, N, H))).astype(np.float32)
        t0 = time.perf_counter()
        OC.step(c, s0, np.zeros((E, H), np.float32), du, 0.0, 1.0, n_threads=threads, want_S=False)
        return time.perf_counter() - t0

    t1 = run(threads)                      # one env per core: calibrates the sample size
    reps = int(max(1, min(64, budget_s / max(t1, 1e-3))))
    E = threads * reps
    t = run(E)
    all_threads = threads
    threads = 1                            # the reference's own situation: one process, one core
    k1 = int(max(2, min(64, 2 * 1.5 / max(run(2), 1e-3))))      # ~1.5 s of one core
    ts = run(k1)
    return {"value": E * N / t, "unit": "rollouts/s", "cores": all_threads, "kind": "port",
            "sample": f"{E} envs x {N} rollouts x {H} steps x 10 substeps in {t:.2f} s; oracle/cpmppi_oracle.c "
                      f"(gcc -O2, no fast-math, OpenMP over envs x rollouts)",
            "single_thread": {"value": k1 * N / ts, "unit": "rollouts/s", "cores": 1,
                              "sample": f"{k1} envs x {N} rollouts x {H} steps in {ts:.2f} s"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=8192, help="independent MPPI problem instances per GPU")
    ap.add_argument("--rollouts", type=int, default=1024)
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--noise", choices=["buffer", "philox"], default="philox",
                    help="philox: perturbation knots generated in-kernel from a counter-based RNG (no buffer); "
                         "buffer: device sampler writes delta_u[E,N,H] to HBM, rollout kernel reads it back "
                         "(the reference's tensor layout at the optimizer/predictor seam)")
    ap.add_argument("--math", choices=["fast", "precise"], default="fast")
    ap.add_argument("--predictor", choices=["ode", "gru"], default="ode",
                    help="ode: predictor_ODE_v0 (the headline path); gru: GRU-6IN-32H1-32H2-5OUT on the f32 matrix "
                         "cores inside the same MPPI loop (BASELINE configs[4], synthetic weights)")
    ap.add_argument("--config", choices=["C2", "C3", "C4"], default=None,
                    help="BASELINE.json config presets: C2 = 1024x50 (the metric's shape, default: 2048 envs per GPU); "
                         "C3 = 64 envs x 4096 x 100 in one launch; C4 = 64 envs per GPU x 2048 x 50 (512 envs over 8 GPUs)")
    ap.add_argument("--rpl", type=int, default=0, help="rollouts per lane: 0 auto, 1, 2 (tuning knob)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-env", action="store_true")
    args = ap.parse_args()
    if args.config == "C3":
        args.envs, args.rollouts, args.horizon = 64, 4096, 100
    elif args.config == "C4":
        args.envs, args.rollouts, args.horizon = 64, 2048, 50

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ       # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # (development aid: CPMPPI_BENCH_BACKEND=gloo CPMPPI_BENCH_ONE_DEVICE=1 runs the N>1 code path on a 1-GPU box)
    backend = os.environ.get("CPMPPI_BENCH_BACKEND", "nccl")
    if os.environ.get("CPMPPI_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    if not os.path.exists(os.path.join(ROOT, "cartpolesimulation_amd", "libcpmppi.so")):     # (git-ignored artefact)
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if distributed:
            dist.barrier()
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig

    E, N, H = args.envs, args.rollouts, args.horizon
    cfg = MPPIConfig(num_rollouts=N, mpc_horizon=H, math_mode=args.math, rollouts_per_lane=args.rpl)
    eng = MPPIEngine(E, cfg, device=local_rank)
    s0, tp, te, L = synthetic_inputs(E, H, seed=2 + rank, device=device)
    u_nom = eng.zeros(E, H)
    Q_out = eng.empty(E)
    du = eng.empty(E, N, H) if args.noise == "buffer" else None
    # the one collective of the path: all-gather of the updated nominal sequences.  Envs are independent,
    # so step i+1 does not need step i's gathered result: the gather of a snapshot runs asynchronously on RCCL's stream
    # while the next step's kernel computes (two snapshot/result buffers, each waited on before it is reused).
    gathered = [torch.empty(world * E * H, dtype=torch.float32, device=device) for _ in range(2)] if distributed else None
    snapshot = [torch.empty(E * H, dtype=torch.float32, device=device) for _ in range(2)] if distributed else None
    pending = [None, None]
    seed = 1234
    pred_kw = {}
    if args.predictor == "gru":
        rng = np.random.Generator(np.random.SFC64(5))
        u = lambda *s: rng.uniform(-1, 1, s).astype(np.float32) / np.sqrt(32.0, dtype=np.float32)
        eng.set_gru(dict(w_ih0=u(96, 6), w_hh0=u(96, 32), b_ih0=u(96), b_hh0=u(96), w_ih1=u(96, 32), w_hh1=u(96, 32),
                         b_ih1=u(96), b_hh1=u(96), w_out=u(5, 32), b_out=u(5)))
        pred_kw = dict(predictor="GRU")

    def step(i):
        if args.noise == "buffer":
            eng._check(eng.lib.cpmppi_sample(eng._h, E, seed, i, rank * E, None, du.data_ptr(), eng._stream()))
            eng.step(s0, u_nom, tp, te, L=L, delta_u=du, Q_out=Q_out, **pred_kw)
        else:
            eng.step(s0, u_nom, tp, te, L=L, seed=seed, offset=i, env_offset=rank * E, Q_out=Q_out, **pred_kw)
        if distributed:
            b = i & 1
            if pending[b] is not None:
                pending[b].wait()                                   # stream-level wait: the buffers are free again
            snapshot[b].copy_(u_nom.view(-1))
            pending[b] = dist.all_gather_into_tensor(gathered[b], snapshot[b], async_op=True)

    def barrier():
        if distributed:
            for b in range(2):
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    eng.set_profiling(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    rollout_ms, finalize_ms = eng.get_profile()
    eng.set_profiling(False)
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(u_nom).all(), "non-finite nominal controls"
    if distributed:      # the last gather delivered this rank's block (and finite blocks from every other rank)
        last = gathered[(args.warmup + args.steps - 1) & 1].view(world, E * H)
        assert torch.equal(last[rank], u_nom.view(-1)) and torch.isfinite(last).all(), "all-gather of the controls is wrong"

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * E * N * args.steps / elapsed
        k_ms = float(np.mean(rollout_ms))
        bytes_launch = algorithmic_bytes_per_rollout(N, H) * E * N
        flops_launch = algorithmic_flops_per_rollout(H) * E * N
        achieved_gbs = bytes_launch / (k_ms * 1e-3) / 1e9
        traffic = None          # HBM bytes per launch of the dominant kernel from separate rocprofv3 --pmc passes
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # written by tools/summarize_profile.py
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(args.noise, {})
                if (rec.get("E"), rec.get("N"), rec.get("H")) == (E, N, H):
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        if args.predictor == "gru":
            # useful flops of the GRU-6IN-32H1-32H2-5OUT forward per rollout-step: 2 x (3*32*(6+32) + 3*32*(32+32) + 5*32)
            gru_flops = 2.0 * (3 * 32 * 38 + 3 * 32 * 64 + 5 * 32) * H * E * N
            if args.math == "fast":
                # FAST: float32-equivalent products as 3 f16 MFMAs (hi*hi + hi*lo + lo*hi), dense f16 peak ~2.5 PFLOP/s
                peak, issued = F16_MFMA_PEAK_TFLOPS, 69 * 2.0 * 32 * 32 * 16 / 32.0 * H * E * N
                note = ("split-f16 products on v_mfma_f32_32x32x16_f16 (69 MFMAs per 32-rollout tile step, 3.6x the algorithmic "
                        "flops); the matrix pipe is ~40 % busy and overlaps with the gate math (exp, rcp), which bounds the "
                        "kernel; --math precise runs the exact-f32 MFMA kernel")
            else:
                peak, issued = FP32_VALU_PEAK_TFLOPS, 156 * 2.0 * 32 * 32 * 2 / 32.0 * H * E * N
                note = ("f32-input MFMA (v_mfma_f32_32x32x2_f32): dense peak = the fp32 vector peak, 157.3 TFLOP/s; this MFMA "
                        "does not co-execute with vector instructions, so gate math adds to it")
            roof = {"bound": "mfma", "kernel": "gru_rollout_cost_kernel", "achieved": gru_flops / (k_ms * 1e-3) / 1e12,
                    "peak": peak, "unit": "TFLOP/s", "frac": gru_flops / (k_ms * 1e-3) / 1e12 / peak, "traffic": traffic,
                    "issued_mfma_tflops": issued / (k_ms * 1e-3) / 1e12,
                    "kernel_ms": k_ms, "finalize_kernel_ms": float(np.mean(finalize_ms)), "note": note}
        else:
            roof = None
        out = {
            "metric": f"MPPI rollouts/sec ({N} samples x {H}-step horizon)", "value": value, "unit": "rollouts/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config or 'C2'}-shape MPPI problems: {N} samples x {H}-step horizon x 10 Euler "
                                   f"substeps, {E} independent envs per GPU batched in one launch"
                                   + ("" if args.config in ("C3", "C4") else " (BASELINE configs[1] shape)"),
                       "envs_per_gpu": E, "rollouts": N, "horizon": H, "substeps": 10,
                       "cost": cfg.cost_function_specification, "noise": args.noise, "math": args.math,
                       "predictor": "predictor_ODE_v0" if args.predictor == "ode" else "GRU-6IN-32H1-32H2-5OUT (synthetic weights)",
                       "parallelism": f"env-sharded x{world}, one RCCL all-gather of u_nom per step" if world > 1
                       else "single GPU"},
            "roofline": roof or {"bound": "hbm", "kernel": "rollout_cost_kernel", "achieved": achieved_gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": k_ms, "finalize_kernel_ms": float(np.mean(finalize_ms)),
                         "algorithmic_bytes_per_rollout": algorithmic_bytes_per_rollout(N, H),
                         "note": "the path is fp32-VALU bound, not HBM bound (SURVEY.md F8): see roofline_valu"},
            "roofline_valu": {"bound": "fp32-valu", "achieved": flops_launch / (k_ms * 1e-3) / 1e12,
                              "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": flops_launch / (k_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS,
                              "algorithmic_flops_per_rollout": algorithmic_flops_per_rollout(H),
                              # the same peak with every substep's sincos costed at ~30 flop-equivalents
                              # (the algebraic count above leaves the 2 transcendental evaluations + fmod per substep out)
                              "survey_ceiling_rollouts_per_s": FP32_VALU_PEAK_TFLOPS * 1e12 /
                              (algorithmic_flops_per_rollout(H) + 2.0 * 10 * H * 30.0),
                              "kernel_rollouts_per_s": E * N / (k_ms * 1e-3)},
        }
        if not args.no_single_env and world == 1:
            # latency of ONE problem instance (BASELINE configs[1] literally: single env), same kernels
            e1 = MPPIEngine(1, cfg, device=local_rank)
            if args.predictor == "gru":
                e1.lib.cpmppi_set_gru  # same synthetic model
                e1.set_gru({k: v for k, v in zip(("w_ih0", "w_hh0", "b_ih0", "b_hh0", "w_ih1", "w_hh1", "b_ih1", "b_hh1",
                                                  "w_out", "b_out"),
                                                 (u(96, 6), u(96, 32), u(96), u(96), u(96, 32), u(96, 32), u(96), u(96),
                                                  u(5, 32), u(5)))})
            u1 = e1.zeros(1, H)
            for i in range(5):
                e1.step(s0[:1], u1, tp[:1], te[:1], L=L[:1], seed=seed, offset=i, **pred_kw)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 50
            for i in range(reps):
                e1.step(s0[:1], u1, tp[:1], te[:1], L=L[:1], seed=seed, offset=100 + i, **pred_kw)
            torch.cuda.synchronize()
            dt1 = (time.perf_counter() - t1) / reps
            e1.set_profiling(True)
            for i in range(20):
                e1.step(s0[:1], u1, tp[:1], te[:1], L=L[:1], seed=seed, offset=200 + i, **pred_kw)
            r1, f1 = e1.get_profile()
            out["single_env"] = {"us_per_step": dt1 * 1e6, "rollouts_per_s": N / dt1, "noise": "philox",
                                 "rollout_kernel_us": float(np.median(r1)) * 1e3,
                                 "finalize_kernel_us": float(np.median(f1)) * 1e3,
                                 "note": "host-paced python loop, one launch per step (finalize fused into the rollout kernel); kernel time from HIP events"}
        if not args.no_cpu_baseline and world == 1:          # reported at N = 1 only (bench contract)
            out["cpu_baseline"] = cpu_baseline(N, H)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
