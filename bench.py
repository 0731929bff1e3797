#!/usr/bin/env python3
"""bench.py — MPPI rollouts/s of the fused HIP hot path on MI355X (BASELINE.json metric).

A "step" = one full MPPI optimizer step for a batch of E independent problem instances (envs) of the C2 shape
(1024 samples x 50-step horizon x 10 Euler substeps, BASELINE.json configs[1]): perturbation sampling (a17) +
rollout (a3-a11) + cost (a12, a15) + importance-weighted update (a16) + shift/clip (a18).  One rollout = one sampled
control sequence integrated over the horizon + its cost + its share of the update.  Inputs are synthetic and resident
in HBM before the timed region.

`python bench.py --gpus N` with N > 1 starts N ranks itself (one fresh `torch.distributed.run` child, before this
process has touched torch or the GPU, never an exec) and forwards rank 0's line; launched by `torch.distributed.run`
directly (RANK / WORLD_SIZE in the environment) it is one of those ranks.  Every rank owns E envs (weak scaling, no
data-path collective) and the chosen control sequences are gathered with ONE RCCL all-gather per step.

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` and `cpu_baseline` objects, plus
side objects for BASELINE's other configurations (`configs.C3`, `configs.C4`, `configs.C5_gru`, `single_env`).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA, MI355X_MICROARCH.md
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector

PRESETS = {"C2": (8192, 1024, 50), "C3": (64, 4096, 100), "C4": (64, 2048, 50)}


def algorithmic_bytes_per_rollout(N, H):
    """delta_u read once (4H), S written once (4), per-env vectors amortised over N."""
    return 4.0 * H + 4.0 + (24.0 + 8.0 * H + 12.0) / N


def algorithmic_flops_per_rollout(H, S=10):
    """38 algebraic flops per substep + ~30 per control step for cost/correction/update."""
    return H * (S * 38.0 + 30.0) + 2.0 * H


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; --config C3: 100, C4: 200)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default 5; C3: 110, C4: 220 - a full pass: the first one after idle runs slow)")
    ap.add_argument("--envs", type=int, default=8192, help="independent MPPI problem instances per GPU")
    ap.add_argument("--rollouts", type=int, default=1024)
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--noise", choices=["buffer", "buffer-ref", "philox"], default="philox",
                    help="philox: perturbation knots generated in-kernel from a counter-based RNG (no buffer); "
                         "buffer: the device sampler writes delta_u to HBM in the library's tiled layout and the rollout "
                         "kernel reads it back with fully used, coalesced accesses; buffer-ref: the sampler writes the "
                         "reference's rollout-major delta_u[E,N,H] (the tensor at the optimizer/predictor seam), read in "
                         "32-byte row segments")
    ap.add_argument("--math", choices=["fast", "precise"], default="fast")
    ap.add_argument("--predictor", choices=["ode", "gru"], default="ode",
                    help="ode: predictor_ODE_v0 (the headline path); gru: GRU-6IN-32H1-32H2-5OUT on the matrix "
                         "cores inside the same MPPI loop (BASELINE configs[4], synthetic weights)")
    ap.add_argument("--predictor-type", choices=["ODE_v0", "ODE"], default="ODE_v0",
                    help="which in-tree ODE predictor integrates the rollouts: ODE_v0 (the headline path: predictor_ODE_v0) or ODE "
                         "(predictor_ODE - the shipped config_controllers.yml's predictor_specification: Euler-Cromer, no edge bounce)")
    ap.add_argument("--config", choices=["C2", "C3", "C4"], default=None,
                    help="BASELINE.json config presets: C2 = 1024x50 (the metric's shape, 8192 envs per GPU, the default); "
                         "C3 = 64 envs x 4096 x 100 in one launch; C4 = 64 envs per GPU x 2048 x 50 (512 envs over 8 GPUs)")
    ap.add_argument("--rpl", type=int, default=0, help="rollouts per lane: 0 auto, 1, 2 (tuning knob)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=["quick", "full"], default="quick",
                    help="quick (default): one ~1 s all-core pass + one ~0.3 s one-core pass per build of the C oracle; full: the longer "
                         "samples of rounds 1-5 (~2.7 s / ~1 s per build).  Same fields")
    ap.add_argument("--no-single-env", action="store_true")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the oracle check of the timed configurations (the `verified` objects; it runs after the timed "
                         "regions, on rank 0)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the C3 / C4 / GRU side measurements")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / collective plumbing only (CPU, gloo): no kernel runs and `value` is null; what the "
                         "CPU test of the N > 1 entry point uses")
    args = ap.parse_args(argv)
    if args.predictor == "gru" and args.noise == "buffer":
        ap.error("--predictor gru takes --noise philox or buffer-ref (the tiled buffer layout is read by the ODE rollout "
                 "kernel only; cpmppi_step refuses the combination)")
    if args.config:
        args.envs, args.rollouts, args.horizon = PRESETS[args.config]
    # the small configurations take ~0.1-0.3 ms per step and start from u_nom = 0 (the first steps meet more rare events
    # than the settled loop): their default run is as long as the side measurements of the default line
    d_steps, d_warm = {"C3": (100, 110), "C4": (200, 220)}.get(args.config, (20, 5))           # (default line: the driver's --steps 20 --warmup 5)
    args.steps = d_steps if args.steps is None else args.steps
    args.warmup = d_warm if args.warmup is None else args.warmup
    return args


def launch_ranks(args, argv):
    """--gpus N > 1 and not yet inside torch.distributed.run: start N ranks as a CHILD process (this process has not
    imported torch nor touched the GPU), forward its output, return its exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def synthetic_inputs(E, H, seed, device):
    """s0 as data_generator.py:221-256 / config_data_gen.yml:14-18; targets and L per env."""
    import numpy as np
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(N, H, budget_s=8.0, integrator="ODE_v0", full=False):
    """The plain-C oracle (validated against the golden vectors) timed on this host's cores: same step, same shape.
    Three builds of the same source (oracle/Makefile): the checker itself (-O2, strict float32, sin/cos through double)
    and two timing-only builds compiled here for this host (-O3 -march=native with libm float trig, without and with
    -ffast-math: the reference's numba kernels are fastmath=True), each on all cores and on one.
    Default: ONE timed all-core pass per build of ~1 s (after a one-env-per-core calibration pass) and a one-core pass of ~0.3 s -
    about 130 + CPU-seconds on a 128-thread host, the whole leg a few seconds of wall time; `full` (--cpu-baseline full): the
    round-5 sample (a third of `budget_s` per all-core pass, ~1 s per one-core pass).  Same fields either way."""
    import numpy as np
    from oracle import oracle_np as O
    from oracle import oracle_c as OC
    cfg = O.MPPIConfig(N=N, H=H, integrator=integrator)
    c = OC.make_config(cfg)
    threads = OC.max_threads()
    rng = np.random.Generator(np.random.SFC64(4))
    variants = {"checker_O2": (None, "-O2 -ffp-contract=off, float32 sin/cos evaluated in double (the validated checker)")}
    try:
        variants.update(OC.build_bench_variants())
    except Exception as e:                                   # no compiler on this host: report the checker only
        variants["_build_error"] = (None, repr(e))

    pool = {"s0": np.zeros((0, 6), np.float32), "du": np.zeros((0, N, H), np.float32)}

    def run(E, n_threads, lib):
        # (the synthetic inputs are drawn once and grown as needed: every build and every pass works on the same sample, and the
        # generator - 65 M normals for a 1 s all-core pass - stays out of the wall time the driver sees)
        if pool["s0"].shape[0] < E:
            more = E - pool["s0"].shape[0]
            pool["s0"] = np.concatenate([pool["s0"], np.stack([O.create_cartpole_state(rng.uniform(-3, 3), rng.uniform(-5, 5), rng.uniform(-0.15, 0.15),
                                                                                     rng.uniform(-0.3, 0.3)) for _ in range(more)])])
            extra = rng.standard_normal((more, N, H), dtype=np.float32)
            extra *= np.float32(cfg.stdev)
            pool["du"] = np.concatenate([pool["du"], extra])
        s0, du = pool["s0"][:E], pool["du"][:E]
        t0 = time.perf_counter()
        OC.step(c, s0, np.zeros((E, H), np.float32), du, 0.0, 1.0, n_threads=n_threads, want_S=False, use_lib=lib)
        return time.perf_counter() - t0

    table = {}
    for name, (lib, flags) in variants.items():
        if name.startswith("_"):
            table[name] = flags
            continue
        t1 = run(threads, threads, lib)                     # one env per core: calibrates the sample size ...
        if not full:                                        # ... and four per core: a sample that no longer fits the caches runs slower per env
            t1 = max(t1, run(4 * threads, threads, lib) / 4.0)
        reps = int(max(1, min(64, (budget_s / 3.0 if full else 1.0) / max(t1, 1e-3))))
        E = threads * reps
        t = run(E, threads, lib)
        k1 = int(max(2, min(64, 2 * (1.0 if full else 0.3) / max(run(2, 1, lib), 1e-3))))      # ~1 s (0.3 s) of one core
        ts = run(k1, 1, lib)
        table[name] = {"flags": flags, "all_cores": {"value": E * N / t, "cores": threads,
                                                     "sample": f"{E} envs x {N} x {H} x 10 substeps in {t:.2f} s"},
                       "one_core": {"value": k1 * N / ts, "cores": 1, "sample": f"{k1} envs x {N} x {H} in {ts:.2f} s"}}
    best = max((v for v in table.values() if isinstance(v, dict)), key=lambda v: v["all_cores"]["value"])
    return {"value": best["all_cores"]["value"], "unit": "rollouts/s", "cores": threads, "kind": "port",
            "sample": best["all_cores"]["sample"] + f"; oracle/cpmppi_oracle.c ({best['flags']}), OpenMP over envs x "
                                                    f"rollouts; the fastest of the builds in `builds`",
            "cpu": cpu_model(), "builds": table}


class Workload:
    """One engine + its synthetic inputs; `run(steps, warmup)` times K steps between barriers (max over ranks)."""

    def __init__(self, ctx, E, N, H, noise="philox", math="fast", predictor="ode", rpl=0, predictor_type="ODE_v0", engine=None,
                 inputs=None, env_base=None):
        import numpy as np
        import torch
        from cartpolesimulation_amd.engine import MPPIEngine
        from cartpolesimulation_amd.configs import MPPIConfig
        self.ctx, self.E, self.N, self.H, self.noise, self.predictor = ctx, E, N, H, noise, predictor
        self.cfg = MPPIConfig(num_rollouts=N, mpc_horizon=H, math_mode=math, rollouts_per_lane=rpl, predictor_type=predictor_type)
        # (an env group of a GroupedWorkload brings its engine - on its own stream -, its slice of the inputs and the global index
        # of its first env; everything else owns the device's E envs: global index = rank * E + env)
        self.eng = engine if engine is not None else MPPIEngine(E, self.cfg, device=ctx["local_rank"])
        self.env_base = ctx["rank"] * E if env_base is None else int(env_base)
        dev = ctx["device"]
        self.s0, self.tp, self.te, self.L = inputs if inputs is not None else synthetic_inputs(E, H, seed=2 + ctx["rank"], device=dev)
        self.u_nom = self.eng.zeros(E, H)
        self.Q_out = self.eng.empty(E)
        self.du = self.eng.empty(E, N, H) if noise == "buffer-ref" else (self.eng.tiled_empty(E) if noise == "buffer" else None)
        self.seed = 1234
        self.pred_kw = {}
        if predictor == "gru":
            rng = np.random.Generator(np.random.SFC64(5))
            u = lambda *s: rng.uniform(-1, 1, s).astype(np.float32) / np.sqrt(32.0, dtype=np.float32)
            self.gru_model = dict(w_ih0=u(96, 6), w_hh0=u(96, 32), b_ih0=u(96), b_hh0=u(96), w_ih1=u(96, 32),
                                  w_hh1=u(96, 32), b_ih1=u(96), b_hh1=u(96), w_out=u(5, 32), b_out=u(5))
            self.eng.set_gru(self.gru_model)
            self.pred_kw = dict(predictor="GRU")
        # the one collective of the path: all-gather of the updated nominal sequences.  Envs are
        # independent, so step i+1 does not need step i's gathered result: the gather of step i runs on a side stream under
        # step i+1's rollout kernel.  Production path = the library's own RCCL communicator (shard.NativeGather:
        # cpmppi_comm_gather, two alternating u_nom buffers, everything enqueued from C); if RCCL cannot be bound on ANY rank,
        # every rank falls back to torch.distributed (snapshot copy + async all_gather_into_tensor) and the line says so.
        W = ctx["world"]
        coll = ctx["collective"]
        self.native, self.collective_impl = None, None
        self.collective_events = os.environ.get("CPMPPI_BENCH_COLLECTIVE") == "native-events"
        self.gathered = self.snapshot = None
        self.pending = [None, None]
        self.prepared = None
        if coll:
            self._init_collective(W, dev)

    _serial = 0

    def _init_collective(self, W, dev):
        import torch
        import torch.distributed as dist
        from cartpolesimulation_amd.shard import NativeGather, exchange_unique_id
        E, H, rank = self.E, self.H, self.ctx["rank"]
        Workload._serial += 1
        ok, why = 1, ""
        want = os.environ.get("CPMPPI_BENCH_COLLECTIVE")          # unset: the library's own RCCL communicator under the nccl backend
        if want not in (None, "native", "native-events") or (want is None and self.ctx["backend"] != "nccl"):
            ok, why = 0, "torch.distributed requested (CPMPPI_BENCH_COLLECTIVE / non-RCCL backend)"
        else:
            try:
                # (development aid: CPMPPI_BENCH_RCCL_PATH names the collective library the communicator binds - with
                # tests/fake_rccl/libfake_rccl.so two ranks can share ONE device, which RCCL refuses)
                lib_path = os.environ.get("CPMPPI_BENCH_RCCL_PATH") or None
                uid = exchange_unique_id(self.eng.lib, rank, key=f"cpmppi_comm_id_{Workload._serial}", rccl_path=lib_path and lib_path.encode())
                self.native = NativeGather(self.eng, uid, W, rank, rccl_path=lib_path, stamped=not self.collective_events)
            except Exception as e:                       # RCCL missing / init failed on this rank
                ok, why = 0, repr(e)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev if self.ctx["backend"] == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # every rank takes the same path
        if int(flag.item()) == 1:
            self.collective_impl = ("cpmppi_comm_gather: ncclAllGather on the library's side stream, ordered with HIP events"
                                    if self.collective_events else
                                    "cpmppi_step_gather: one ncclAllGather per step on the library's side stream, ordered with "
                                    "the rollout kernel through device memory")
            self.u_nom = None                            # (the two buffers live in self.native.u)
            return
        if self.native is not None:
            self.native.close()
            self.native = None
        self.collective_impl = "torch.distributed all_gather_into_tensor (async_op) of a snapshot" + (f" [{why}]" if why else "")
        self.gathered = [torch.empty(W * E * H, dtype=torch.float32, device=dev) for _ in range(2)]
        self.snapshot = [torch.empty(E * H, dtype=torch.float32, device=dev) for _ in range(2)]

    def final_u_nom(self, last_step):
        """The nominal sequences as step `last_step` left them."""
        return self.native.u_out(last_step) if self.native else self.u_nom

    def step(self, i):
        import torch.distributed as dist
        e, rank = self.eng, self.ctx["rank"]
        if self.native is not None:
            g = self.native
            uin, uout, recv = g.u_in(i), g.u_out(i), g.recv(i)
            ev = self.collective_events                  # development aid: the HIP-event form of the same ordering
            if ev:
                g.before_step(i)
                recv = None
            if self.noise == "buffer-ref":
                e._check(e.lib.cpmppi_sample(e._h, self.E, self.seed, i, self.env_base, None, self.du.data_ptr(), e._stream()))
                e.step(self.s0, uin, self.tp, self.te, L=self.L, delta_u=self.du, Q_out=self.Q_out, u_nom_out=uout,
                       gather_into=recv, **self.pred_kw)
            elif self.noise == "buffer":
                e.sample_tiled(self.seed, i, self.env_base, E=self.E, out=self.du)
                e.step(self.s0, uin, self.tp, self.te, L=self.L, delta_u_tiled=self.du, Q_out=self.Q_out, u_nom_out=uout,
                       gather_into=recv)
            else:
                if self.prepared is None:                # argument blocks built once: the pointers only alternate
                    self.prepared = [e.prepare_step(self.s0, g.u[b], self.tp, self.te, L=self.L, seed=self.seed, offset=0,
                                                    env_offset=self.env_base, Q_out=self.Q_out, u_nom_out=g.u[1 - b],
                                                    **self.pred_kw) for b in range(2)]
                self.prepared[i & 1].run(offset=i, gather_into=recv)
            if ev:
                g.after_step(i)
            return
        if self.noise == "buffer-ref":
            e._check(e.lib.cpmppi_sample(e._h, self.E, self.seed, i, self.env_base, None, self.du.data_ptr(), e._stream()))
            e.step(self.s0, self.u_nom, self.tp, self.te, L=self.L, delta_u=self.du, Q_out=self.Q_out, **self.pred_kw)
        elif self.noise == "buffer":
            e.sample_tiled(self.seed, i, self.env_base, E=self.E, out=self.du)
            e.step(self.s0, self.u_nom, self.tp, self.te, L=self.L, delta_u_tiled=self.du, Q_out=self.Q_out)
        else:
            e.step(self.s0, self.u_nom, self.tp, self.te, L=self.L, seed=self.seed, offset=i, env_offset=self.env_base,
                   Q_out=self.Q_out, **self.pred_kw)
        if self.ctx["collective"]:
            b = i & 1
            if self.pending[b] is not None:
                self.pending[b].wait()                              # stream-level wait: the buffers are free again
            self.snapshot[b].copy_(self.u_nom.view(-1))
            self.pending[b] = dist.all_gather_into_tensor(self.gathered[b], self.snapshot[b], async_op=True)

    def barrier(self):
        import torch
        import torch.distributed as dist
        if self.ctx["collective"]:
            if self.native is not None:
                self.native.sync()
            for b in range(2):
                if self.pending[b] is not None:
                    self.pending[b].wait()
                    self.pending[b] = None
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    def run(self, steps, warmup):
        import numpy as np
        import torch
        import torch.distributed as dist
        for i in range(warmup):
            self.step(i)
        self.barrier()
        # kernel duration from HIP events on the launch stream, live inside the timed region.  An event costs ~5 us on the
        # stream (two per bracketed launch = 10 % of a 100 us launch, and the wall clock of these K steps is what `value`
        # is made of), so launches shorter than ~1 ms are bracketed in GROUPS of 8: kernel_ms is then the average
        # duration of a launch back to back, inter-launch gaps included (an upper bound of the kernel's own duration).
        small = self.predictor == "ode" and self.E * self.N * self.H < 100_000_000      # (< ~0.5 ms per launch)
        self.profile_group = 8 if (small and steps >= 16) else 1
        self.eng.set_profiling(True, group=self.profile_group)
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(warmup + i)
        self.barrier()
        elapsed = time.perf_counter() - t0
        rollout_ms, finalize_ms = self.eng.get_profile()
        self.eng.set_profiling(False)
        self.timed_kernel = self.eng.last_launch()["kernel"] if self.predictor == "ode" else "gru_rollout_cost_kernel"
        self.next_step = warmup + steps
        W, rank = self.ctx["world"], self.ctx["rank"]
        if self.ctx["collective"]:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=self.ctx["device"])
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        i_last = warmup + steps - 1
        u_final = self.final_u_nom(i_last)
        assert torch.isfinite(u_final).all(), "non-finite nominal controls"
        self.collective_report = None
        if self.ctx["collective"]:        # the last gather delivered this rank's block (and finite blocks from every other rank)
            last = (self.native.gathered[(i_last + 1) & 1][:, :self.E * self.H] if self.native else self.gathered[i_last & 1].view(W, self.E * self.H))
            assert torch.equal(last[rank], u_final.view(-1)) and torch.isfinite(last).all(), "all-gather of the controls is wrong"
            # did the collective really span W ranks?  Every rank contributes the checksum of what IT computed (one all-gather of a
            # double through torch.distributed, outside the timed region); the W blocks this rank received through the library's
            # communicator must carry exactly those checksums and - the envs differ per rank - be pairwise distinct.  Plus what
            # RCCL itself reports about the communicator (ncclCommCount / ncclCommUserRank).
            # (an EXACT checksum: the float32 bit patterns summed as integers - independent of any reduction order)
            bits = lambda t: t.contiguous().view(torch.int32).to(torch.int64)            # noqa: E731
            mine = bits(u_final.view(-1)).sum().reshape(1)
            sums = torch.empty(W, dtype=torch.int64, device=mine.device)
            dist.all_gather_into_tensor(sums, mine)
            got = bits(last).sum(dim=1)
            blocks_match = bool(torch.equal(got, sums))
            distinct = bool(W == 1 or len({int(x) for x in got.tolist()}) == W)
            # rccl_*: what ncclCommCount / ncclCommUserRank / ncclGetVersion answer for libcpmppi's own communicator
            # (cpmppi_comm_get_info); null on the torch.distributed form, whose communicator is not ours to query - there
            # backend_ranks is the process group's size
            info = self.native.info() if self.native else {}
            stamps_ok = None
            if self.native and not self.collective_events:
                # stamped blocks (cpmppi_comm_set_stamped): every rank's block of the last gather carries the number of the last step-gather
                stamps_ok = bool((self.native.stamps(i_last) == info["gathers_enqueued"]).all())
            self.collective_report = {"impl": self.collective_impl, "ranks_requested": W, "rccl_ranks": info.get("rccl_ranks"),
                                      "rccl_rank_of_rank0": info.get("rccl_rank"), "rccl_version": info.get("rccl_version"),
                                      "stream_memory_ops": info.get("stream_memory_ops"), "stamped": info.get("stamped"),
                                      "every_block_stamped_with_the_last_step": stamps_ok,
                                      "backend": dist.get_backend(), "backend_ranks": dist.get_world_size(),
                                      "rank_blocks_match_every_ranks_own_checksum": blocks_match, "rank_blocks_distinct": distinct}
            spans = info["rccl_ranks"] == W if self.native else dist.get_world_size() == W
            assert blocks_match and distinct and spans and stamps_ok is not False, f"the collective did not span {W} ranks: {self.collective_report}"
        k_ms = float(np.mean(rollout_ms))
        E, N, H = self.E, self.N, self.H
        return {"elapsed": elapsed, "ms_per_step": 1e3 * elapsed / steps, "value": W * E * N * steps / elapsed,
                "kernel_ms": k_ms, "kernel_ms_min": float(np.min(rollout_ms)), "finalize_kernel_ms": float(np.mean(finalize_ms)),
                "kernel_launches_timed": int(len(rollout_ms)) * self.profile_group, "kernel_event_group": self.profile_group,
                "valu_tflops": algorithmic_flops_per_rollout(H) * E * N / (k_ms * 1e-3) / 1e12,
                "alg_gbs": algorithmic_bytes_per_rollout(N, H) * E * N / (k_ms * 1e-3) / 1e9}

    # the four fixed regimes of tests/golden/rollouts_c2.npz - (angle, angleD, position, positionD), target
    C2_REGIMES = {"upright": ((0.05, 0.0, 0.0, 0.0), 0.0), "hanging": ((3.0, 0.0, 0.1, 0.0), 0.0),
                  "near_edge": ((0.5, 2.0, 0.18, 0.4), 0.05), "fast": ((1.5, 15.0, -0.1, -0.3), 0.05)}

    def verify_regimes(self):
        """The single-env configuration (BASELINE configs[1] literally) verified on MORE than the one state it was timed on: the
        launch's own state plus the four fixed regimes of SURVEY.md 8(d) - a lone chaotic start (21 rad/s) puts every rollout into
        the oracle's flagged bucket and would leave the clear-bucket rule with nothing to compare (round 4: clear 0 of 1024).  Each
        state: one more step of the SAME kernel from a cold nominal sequence, checked like `verify`.  -> the merged report; `ok`
        needs every state's own check to pass AND rollouts in the clear bucket."""
        import numpy as np
        import torch
        assert self.E == 1
        keep = (self.s0.clone(), self.tp.clone(), self.te.clone())
        reports = {"timed_state": self.verify()}
        for name, ((a, ad, x, xd), tgt) in self.C2_REGIMES.items():
            self.s0.copy_(torch.tensor([[a, ad, np.cos(a), np.sin(a), x, xd]], dtype=torch.float32))
            self.tp.fill_(tgt)
            self.te.fill_(1.0)
            self.next_step += 1
            reports[name] = self.verify(cold=True)
        for t, k in zip((self.s0, self.tp, self.te), keep):
            t.copy_(k)
        rep = dict(reports["timed_state"])
        mx = lambda key: (lambda v: max(v) if v else None)([r[key] for r in reports.values() if r.get(key) is not None])   # noqa: E731
        for key in ("envs", "rollouts", "clear", "flagged", "clear_off", "flagged_off", "u_off_envs", "flagged_cap"):
            rep[key] = int(sum(r[key] for r in reports.values()))
        for key in ("worst_clear_excess", "worst_cost_rel", "worst_flagged_excess", "worst_flagged_cost_rel", "worst_u_abs",
                    "worst_u_vs_reference_spread", "noise_device_vs_oracle_max"):
            rep[key] = mx(key)
        rep["states"] = {k: dict(clear=r["clear"], flagged=r["flagged"], clear_off=r["clear_off"], flagged_off=r["flagged_off"],
                                 worst_cost_rel=r["worst_cost_rel"], worst_u_abs=r["worst_u_abs"], ok=r["ok"]) for k, r in reports.items()}
        rep["ok"] = bool(all(r["ok"] for r in reports.values()) and rep["clear"] > 0)
        return rep

    def verify(self, n_envs=8, cold=False):
        """The timed configuration checked against the oracle - OUTSIDE the timed region, the checker only: one more step of
        the SAME launch (same engine, same shape, same noise source, hence the same kernel instantiation: asserted), from the
        nominal sequences the timed steps left, with the per-rollout costs written out; `n_envs` envs spread over the launch
        are then re-computed by the plain-C oracle (modes A and B + the rounding probes; oracle/parity.py rule ODE_V0, or
        PREDICTOR_ODE for that predictor) from their regenerated perturbations (cpmppi_sample with the launch's seed, step
        counter and global env index), the GRU side configuration by the numpy GRU oracle.  -> the `verified` object."""
        import numpy as np
        from oracle import oracle_np as O
        from oracle import parity as PR
        e, E, N, H, rank = self.eng, self.E, self.N, self.H, self.ctx["rank"]
        i = self.next_step
        u_before = self.final_u_nom(i - 1).clone()
        if cold:
            u_before.zero_()
        u_work, S, Q = u_before.clone(), e.empty(E, N), e.empty(E)
        envs = sorted({int(round(x)) for x in np.linspace(0, E - 1, min(n_envs, E))})
        if self.predictor == "gru":
            envs = envs[:2]                                         # (the numpy GRU oracle: ~1 s per env at 1024 x 50)
        du_envs = kn_envs = noise_diff = None
        if self.noise == "buffer-ref":
            e._check(e.lib.cpmppi_sample(e._h, E, self.seed, i, self.env_base, None, self.du.data_ptr(), e._stream()))
            e.step(self.s0, u_work, self.tp, self.te, L=self.L, delta_u=self.du, Q_out=Q, S_out=S, **self.pred_kw)
            du_envs = self.du[envs].cpu().numpy()
        elif self.noise == "buffer":
            e.sample_tiled(self.seed, i, self.env_base, E=E, out=self.du)
            e.step(self.s0, u_work, self.tp, self.te, L=self.L, delta_u_tiled=self.du, Q_out=Q, S_out=S)
            du_envs = e.untile(self.du, E)[envs].cpu().numpy()
        else:
            e.step(self.s0, u_work, self.tp, self.te, L=self.L, seed=self.seed, offset=i, env_offset=self.env_base, Q_out=Q, S_out=S,
                   **self.pred_kw)
            # the perturbations of the checked envs come from the ORACLE's restatement of the generator (oracle/philox_np.py:
            # Philox4x32-10 pinned to Random123's known-answer vectors + the sampler's keying), not from the library's own
            # sampler; what the device would have drawn is read back only to report how far its hardware ln / sin / cos sit
            from oracle import philox_np as PH
            kn_envs = np.concatenate([PH.knots(self.seed, i, self.env_base + x, 1, N, e.P, e.mppi.sigma) for x in envs])
            kn_dev = np.concatenate([e.sample(self.seed, offset=i, env_offset=self.env_base + x, E=1)[0].cpu().numpy() for x in envs])
            noise_diff = float(np.abs(kn_dev.astype(np.float64) - kn_envs).max())
        kernel = e.last_launch()["kernel"] if self.predictor == "ode" else "gru_rollout_cost_kernel"
        host = lambda t: t[envs].cpu().numpy()
        s0, tp, te, L = host(self.s0), host(self.tp), host(self.te), host(self.L)
        ub, ua, Sg = host(u_before), host(u_work), host(S)
        ptype = self.cfg.predictor_type
        if self.predictor == "gru":
            cfg = O.MPPIConfig(N=N, H=H)
            rep = dict(envs=len(envs), rollouts=len(envs) * N, clear=0, flagged=0, clear_off=0, flagged_off=0, worst_cost_rel=0.0,
                       worst_u_abs=0.0, u_off_envs=0, flagged_cap=0, rule="1e-4 band; flagged = rollouts the numpy GRU oracle cannot pin to a quarter band in "
                                             "float32 (float32 vs float64 evaluation)", oracle="oracle_np.gru_mppi_step")
            ident = dict(in_scale=np.ones(6, np.float32), in_shift=np.zeros(6, np.float32), out_scale=np.ones(5, np.float32),
                         out_shift=np.zeros(5, np.float32))
            model = dict(ident, **self.gru_model)                  # (the bench's synthetic model has no normalisation vectors)
            for j in range(len(envs)):
                du = O.interpolate_knots(kn_envs[j], H)
                r32 = O.gru_mppi_step(model, s0[j], ub[j], du, tp[j], te[j], cfg)
                r64 = O.gru_mppi_step(model, s0[j], ub[j], du, tp[j], te[j], cfg, dtype=np.float64)
                b = PR.cost_buckets(Sg[j], r32["S"], r64["S"], PR.flag_rounding_sensitive(r32["S"], r64["S"]))
                clear = ~b["flagged"]
                rep["clear"] += int(clear.sum()); rep["flagged"] += int(b["flagged"].sum())
                rep["clear_off"] += int((b["off"] & clear).sum()); rep["flagged_off"] += int((b["off"] & b["flagged"]).sum())
                rep["worst_cost_rel"] = max(rep["worst_cost_rel"], float(b["rel"][clear].max()))
                rep["worst_u_abs"] = max(rep["worst_u_abs"], float(np.abs(ua[j] - r32["u_new"]).max()))
            rep["ok"] = bool(rep["clear_off"] == 0 and rep["flagged_off"] <= int(np.ceil(PR.FLAGGED_CAP * rep["flagged"]))
                             and rep["worst_u_abs"] <= 1e-4)
        else:
            ocfg = O.MPPIConfig(N=N, H=H, integrator=ptype)
            params = None
            if os.environ.get("CPMPPI_BENCH_TEST_PERTURB_ORACLE") == "1":      # test hook: the checker must be able to say no
                import dataclasses
                params = dataclasses.replace(O.DEFAULT_PARAMS, m_pole=np.float32(0.09))
            rep = PR.verify_envs(ocfg, s0, ub, kn_envs, tp, te, L, Sg, ua, rule=PR.ODE_V0 if ptype == "ODE_v0" else PR.PREDICTOR_ODE,
                                 delta_u=du_envs, params=params)
            rep["oracle"] = "oracle/cpmppi_oracle.c: modes A (float32) and B (float64 substeps) + rounding probes"
        rep.update(env_indices=envs, step=int(i), kernel=kernel, same_kernel_as_timed=bool(kernel == self.timed_kernel),
                   math=self.cfg.math_mode,
                   noise_source=("oracle philox (oracle/philox_np.py: Philox4x32-10 + the sampler's keying, regenerated on the host)"
                                 if kn_envs is not None else "the launch's own perturbation buffer, read back"),
                   noise_device_vs_oracle_max=noise_diff)
        rep["ok"] = bool(rep["ok"] and rep["same_kernel_as_timed"] and np.isfinite(Sg).all())
        return rep

    def close(self):
        if self.native is not None:
            self.native.close()
        self.eng.close()


class CollectiveUnavailable(RuntimeError):
    """Raised by EVERY rank together (agreed by an all-reduce): a side configuration's communicator could not be made."""


class GroupedWorkload:
    """A small configuration with its envs split into G groups, each on its own handle + stream and running its own chain of
    steps (cartpolesimulation_amd/pipeline.py): the `*_pipelined` side configurations.  Same synthetic inputs and Philox keys as
    the unsplit configuration (global env indices); no group ever waits for another inside the timed region."""

    _serial = 0

    def __init__(self, ctx, E, N, H, groups, math="fast"):
        import torch
        from cartpolesimulation_amd import _lib as _L
        from cartpolesimulation_amd.configs import MPPIConfig
        from cartpolesimulation_amd.pipeline import EnvGroups
        self.ctx, self.E, self.N, self.H = ctx, E, N, H
        base = ctx["rank"] * E
        self.groups = EnvGroups(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, math_mode=math), groups, device=ctx["local_rank"], env_offset=base)
        full = synthetic_inputs(E, H, seed=2 + ctx["rank"], device=ctx["device"])
        sub = dict(ctx, collective=False)                   # (the parts only verify: the collective belongs to the groups as a whole)
        self.parts = [Workload(sub, e1 - e0, N, H, math=math, engine=eng, inputs=tuple(t[e0:e1].contiguous() for t in full), env_base=base + e0)
                      for eng, (e0, e1) in zip(self.groups.engines, self.groups.slices)]
        self.full_inputs = full
        self.Q_all = torch.empty(E, device=ctx["device"])
        self.collective_impl, self.collective_report, self.recv, self.done = None, None, None, 0
        n, dev = E * H, ctx["device"]
        if ctx["collective"]:
            # env groups + the per-step all-gather: ONE communicator and side stream for the device, one all-gather
            # of the whole u_nom[E, H] per step, two alternating buffers, stamped blocks (cpmppi_groups_run_gather)
            lib_path = os.environ.get("CPMPPI_BENCH_RCCL_PATH") or None
            if ctx["backend"] != "nccl" and not lib_path:
                raise RuntimeError("the grouped configurations gather through the library's own RCCL communicator (backend nccl)")
            import torch.distributed as dist
            from cartpolesimulation_amd.shard import exchange_unique_id
            GroupedWorkload._serial += 1
            ok, why = 1, ""
            try:
                uid = exchange_unique_id(self.groups.lib, ctx["rank"], key=f"cpmppi_groups_comm_id_{GroupedWorkload._serial}",
                                         rccl_path=lib_path and lib_path.encode())
                self.groups.comm_init(uid, ctx["world"], ctx["rank"], rccl_path=lib_path, stamped=True)
            except Exception as e:  # noqa: BLE001
                ok, why = 0, repr(e)
            # every rank takes the same way out: a communicator that could not be made on ANY rank makes this configuration
            # unavailable on ALL of them, before anybody has entered a collective the others would wait in
            flag = torch.tensor([ok], dtype=torch.int32, device=dev if ctx["backend"] == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                self.groups.close()
                raise CollectiveUnavailable(f"the env groups' communicator could not be created on every rank" + (f" (this rank: {why})" if why else ""))
            pad = _L.GATHER_STAMP_FLOATS
            self._flat = [torch.zeros(n + pad, device=dev) for _ in range(2)]
            self.u = [f[:n].view(E, H) for f in self._flat]
            self.recv = torch.zeros(ctx["world"], n + pad, device=dev)
            self.step_ab = [self.groups.prepare(full[0], self.u[b], full[1], full[2], L=full[3], seed=self.parts[0].seed, Q_out=self.Q_all,
                                                u_nom_out=self.u[1 - b]) for b in range(2)]
            self.collective_impl = ("cpmppi_groups_run_gather: the env groups of the device under ONE communicator and side stream, one "
                                    "ncclAllGather of u_nom[E, H] per step (stamped blocks), ordered with the groups' rollout kernels "
                                    "through device memory")
        else:
            # one argument block over all envs; cpmppi_groups_run enqueues every group's launch of a step from C
            self.u = [torch.zeros(E, H, device=dev)]
            self.step_ab = [self.groups.prepare(full[0], self.u[0], full[1], full[2], L=full[3], seed=self.parts[0].seed, Q_out=self.Q_all)]
        torch.cuda.synchronize()

    @property
    def u_all(self):
        """The nominal sequences as the steps run so far left them."""
        return self.u[self.done & 1] if self.recv is not None else self.u[0]

    def _run(self, periods, offset):
        if self.recv is not None:
            self.groups.run(self.step_ab[self.done & 1], None, periods=periods, offset=offset, gather_into=self.recv)
        else:
            self.groups.run(self.step_ab[0], None, periods=periods, offset=offset)
        self.done += periods

    def _settle(self):
        import torch
        torch.cuda.synchronize()
        if self.recv is not None:
            import torch.distributed as dist
            self.groups.comm_sync()
            dist.barrier()
            torch.cuda.synchronize()

    def run(self, steps, warmup, overlap=True):
        import numpy as np
        import torch
        self.groups.fork()
        # (outside the timed region) 1.0 = the groups were serialised
        self.stream_overlap = self.groups.overlap(self.step_ab[0]) if overlap else None
        for u in self.u:
            u.zero_()
        torch.cuda.synchronize()
        self.groups.fork()
        self._run(warmup, 0)
        self._settle()
        t0 = time.perf_counter()
        self._run(steps, warmup)                 # K steps of every group (+ K all-gathers): ONE library call, nothing else
        self._settle()
        elapsed = time.perf_counter() - t0
        W, rank = self.ctx["world"], self.ctx["rank"]
        if self.recv is not None:
            import torch.distributed as dist
            from cartpolesimulation_amd.shard import block_stamps
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=self.ctx["device"])
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            # the last gather: this rank's block = what it computed, every rank's block = that rank's own checksum, every stamp = the
            # number of step-gathers made (see Workload.run for the form of the check)
            n = self.E * self.H
            bits = lambda t: t.contiguous().view(torch.int32).to(torch.int64)            # noqa: E731
            mine = bits(self.u_all.reshape(-1)).sum().reshape(1)
            sums = torch.empty(W, dtype=torch.int64, device=mine.device)
            dist.all_gather_into_tensor(sums, mine)
            got = bits(self.recv[:, :n]).sum(dim=1)
            info = self.groups.comm_info()
            stamps = block_stamps(self.recv, n)
            self.collective_report = {"impl": self.collective_impl, "ranks_requested": W, "rccl_ranks": info.get("rccl_ranks"),
                                      "rccl_version": info.get("rccl_version"), "stream_memory_ops": info.get("stream_memory_ops"),
                                      "stamped": info.get("stamped"), "gathers": info.get("gathers_enqueued"),
                                      "rank_blocks_match_every_ranks_own_checksum": bool(torch.equal(got, sums)),
                                      "rank_blocks_distinct": bool(W == 1 or len({int(x) for x in got.tolist()}) == W),
                                      "every_block_stamped_with_the_last_step": bool((stamps == self.done).all())}
            rep = self.collective_report
            assert torch.equal(self.recv[rank, :n], self.u_all.reshape(-1)), "all-gather of the controls is wrong"
            assert (rep["rank_blocks_match_every_ranks_own_checksum"] and rep["rank_blocks_distinct"] and rep["rccl_ranks"] == W
                    and rep["every_block_stamped_with_the_last_step"] and rep["gathers"] == self.done), f"the collective is wrong: {rep}"
        # the groups' own kernel durations (they overlap: informative only) from a pass of their own - the library's HIP events cost
        # these small launches 0.9 us per step (profiles/r5/prof_overhead.txt) and the timed pass above is wall time only
        for w in self.parts:
            w.eng.set_profiling(True, group=8 if steps >= 16 else 1)
        self._run(steps, warmup + steps)
        self._settle()
        for w, (e0, e1) in zip(self.parts, self.groups.slices):         # (the parts verify from the nominal sequences these steps left)
            w.u_nom.copy_(self.u_all[e0:e1])
        k = []
        for w in self.parts:
            r, _ = w.eng.get_profile()
            w.eng.set_profiling(False)
            k.append(float(np.mean(r)))
            w.timed_kernel, w.next_step = w.eng.last_launch()["kernel"], warmup + 2 * steps
            assert torch.isfinite(w.u_nom).all()
        E, N = self.E, self.N
        return {"elapsed": elapsed, "ms_per_step": 1e3 * elapsed / steps, "value": W * E * N * steps / elapsed,
                "group_kernel_ms": k, "kernels": sorted({w.timed_kernel for w in self.parts}), "stream_overlap": self.stream_overlap}

    def verify(self):
        """Every group checked like a workload of its own (4 envs each, global Philox keys), on torch's current stream."""
        import torch
        torch.cuda.synchronize()
        reps = []
        for w in self.parts:
            w.eng.use_stream(None)
            reps.append(w.verify(n_envs=4))
        rep = dict(reps[0])
        for key in ("envs", "rollouts", "clear", "flagged", "clear_off", "flagged_off", "u_off_envs", "flagged_cap", "second_stage_envs"):
            rep[key] = int(sum(r[key] for r in reps))
        rep["second_stage"] = [dict(x, group=gi) for gi, r in enumerate(reps) for x in r["second_stage"]]
        for key in ("worst_clear_excess", "worst_cost_rel", "worst_flagged_excess", "worst_flagged_cost_rel", "worst_u_abs",
                    "worst_u_vs_reference_spread", "noise_device_vs_oracle_max"):
            v = [r[key] for r in reps if r.get(key) is not None]
            rep[key] = max(v) if v else None
        rep["env_indices"] = [w.env_base - self.parts[0].env_base + x for w, r in zip(self.parts, reps) for x in r["env_indices"]]
        rep["kernel"] = sorted({r["kernel"] for r in reps})
        rep["same_kernel_as_timed"] = bool(all(r["same_kernel_as_timed"] for r in reps))
        rep["ok"] = bool(all(r["ok"] for r in reps))
        return rep

    def close(self):
        self.groups.close()


def roofline_valu(r, E, N, H):
    return {"bound": "fp32-valu", "achieved": r["valu_tflops"], "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": r["valu_tflops"] / FP32_VALU_PEAK_TFLOPS, "algorithmic_flops_per_rollout": algorithmic_flops_per_rollout(H),
            "kernel_rollouts_per_s": E * N / (r["kernel_ms"] * 1e-3)}


def profiled_traffic(noise, E, N, H):
    """HBM bytes per launch of the rollout kernel from an EARLIER rocprofv3 --pmc run of this command (FETCH_SIZE and
    WRITE_SIZE in separate passes, tools/profile.sh -> tools/summarize_profile.py -> profiles/<round>/pmc_traffic.json,
    committed).  PMC counters cannot be collected from inside the run; the figure is labelled with its source."""
    import glob
    import re
    rounds = glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json"))
    key = lambda p: int((re.search(r"profiles[/\\]r(\d+)", p) or [0, -1])[1])      # newest ROUND first (r10 after r9)
    for path in sorted(rounds, key=key, reverse=True):
        try:
            rec = json.load(open(path)).get(noise, {})
        except Exception:
            continue
        if (rec.get("E"), rec.get("N"), rec.get("H")) == (E, N, H):
            return rec.get("hbm_bytes_per_launch"), os.path.relpath(path, ROOT), rec.get("collected")
    return None, None, None


def main():
    t_main = time.perf_counter()
    phases = {}

    def mark(name):                                        # wall seconds since the process entered main(): where a run's time goes
        phases[name] = round(time.perf_counter() - t_main, 2)

    args = parse_args()
    in_rank = "RANK" in os.environ and "WORLD_SIZE" in os.environ       # started by torch.distributed.run
    if args.gpus > 1 and not in_rank:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist
    mark("imports_done")
    world = int(os.environ.get("WORLD_SIZE", "1")) if in_rank else 1
    rank = int(os.environ.get("RANK", "0")) if in_rank else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if in_rank else 0
    if in_rank and world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but torch.distributed.run started {world} ranks")
    # the data-path collective runs whenever there is more than one rank; CPMPPI_BENCH_FORCE_COLLECTIVE=1 runs it with a
    # single rank too (development aid: exercises RCCL init, the async all-gather and the max-reduce on a 1-GPU box)
    collective = world > 1 or (in_rank and os.environ.get("CPMPPI_BENCH_FORCE_COLLECTIVE") == "1")
    if in_rank:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # (development aid: CPMPPI_BENCH_BACKEND=gloo CPMPPI_BENCH_ONE_DEVICE=1 runs the N>1 code path on a 1-GPU box)
    backend = os.environ.get("CPMPPI_BENCH_BACKEND", "nccl")
    if os.environ.get("CPMPPI_BENCH_ONE_DEVICE") == "1":
        local_rank = 0

    if args.dry_run:
        # launcher + rendezvous + the all-gather of the control sequences on CPU tensors; nothing is computed or timed
        if in_rank:
            dist.init_process_group("gloo")
            mine = torch.full((4,), float(rank))
            out = torch.empty(world * 4)
            dist.all_gather_into_tensor(out, mine)
            assert torch.equal(out.view(world, 4)[:, 0], torch.arange(world, dtype=torch.float32))
            dist.barrier()
            blocks = out.view(world, 4)
            # the receiver's side of the STAMPED blocks (cpmppi_comm_set_stamped; what C4_pipelined's gather carries under --gpus N):
            # three gathers of [n + stamp words] per rank; the last rank re-sends its block of gather 1 as gather 2 - a rank that
            # dropped a step - and every rank must reject exactly that block and keep what it had
            from cartpolesimulation_amd._lib import GATHER_STAMP_FLOATS
            from cartpolesimulation_amd.shard import merge_accepted
            n, seqs, rejected = 20, torch.zeros(world, 20), []
            for number in (1, 2, 3):
                sent = number - 1 if (number == 2 and rank == world - 1 and world > 1) else number
                blk = torch.zeros(n + GATHER_STAMP_FLOATS)
                blk[:n] = float(rank) + sent / 100.0
                blk[n:n + 1] = torch.tensor([sent], dtype=torch.int32).view(torch.float32)
                got = torch.empty(world, n + GATHER_STAMP_FLOATS)
                dist.all_gather_into_tensor(got.view(-1), blk)
                seqs, ok = merge_accepted(seqs, got, n, number)
                rejected += [[number, r] for r in range(world) if not bool(ok[r])]
            want = torch.arange(world, dtype=torch.float32) + 0.03
            coll = {"impl": "torch.distributed all_gather_into_tensor over gloo (dry run: CPU tensors, no RCCL)", "ranks_requested": world,
                    "rccl_ranks": None, "backend_ranks": dist.get_world_size(),
                    "rank_blocks_distinct": bool(world == 1 or len({float(x) for x in blocks[:, 0].tolist()}) == world),
                    "stamped_blocks": {"gathers": 3, "rejected": rejected, "final_rows_are_gather_3": bool(torch.allclose(seqs[:, 0], want))},
                    "pipelined_under_collective": "C4_pipelined: env groups under one communicator, cpmppi_groups_run_gather "
                                                  "(a real run reports it in configs.C4_pipelined.collective)"}
        else:
            coll = None
        if rank == 0:
            print(json.dumps({"metric": f"MPPI rollouts/sec ({args.rollouts} samples x {args.horizon}-step horizon)",
                              "value": None, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "dry_run": True, "config": {"collective": coll},
                              "note": "launcher / collective plumbing check on CPU (gloo); no kernel ran"}), flush=True)
        if in_rank:
            dist.destroy_process_group()
        return

    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if in_rank:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    if not os.path.exists(os.path.join(ROOT, "cartpolesimulation_amd", "libcpmppi.so")):     # (git-ignored artefact)
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if in_rank:
            dist.barrier()

    ctx = {"world": world, "rank": rank, "local_rank": local_rank, "device": device, "collective": collective,
           "backend": backend}
    E, N, H = args.envs, args.rollouts, args.horizon
    main_wl = Workload(ctx, E, N, H, noise=args.noise, math=args.math, predictor=args.predictor, rpl=args.rpl,
                       predictor_type=args.predictor_type)
    r = main_wl.run(args.steps, args.warmup)
    mark("headline_timed")
    cfg = main_wl.cfg
    # Verification against the oracle happens AFTER every timed region of the run (rank 0 only): the workloads are kept until
    # then.  (Not only a matter of principle: the C oracle's OpenMP pool, once started in this process, slows the host-paced
    # single-env loop below by an order of magnitude.)
    verified, to_verify = {}, [("main", main_wl)]

    # BASELINE's other configurations, measured by EVERY rank (the collective is part of them), reported by rank 0
    extras, side_failed = {}, []
    if (not args.no_extra_configs and args.config is None and args.predictor == "ode" and args.predictor_type == "ODE_v0"
            and args.noise == "philox" and args.math == "fast"):
        # (the small configurations are warmed for a full pass before their timed pass: the first ~15 ms of such launches after the
        # GPU has sat idle behind host work run 5-9 % slower - profiles/r5/prof_overhead.txt, first round against the later ones)
        side = [("C4", PRESETS["C4"], "ode", 200, 220)]
        if world == 1:
            # ... and the headline shape on the reference's OTHER in-tree ODE predictor, the one its shipped config_controllers.yml
            # names (predictor_specification "ODE": Euler-Cromer substeps, no edge bounce)
            side = [("C3", PRESETS["C3"], "ode", 100, 110)] + side + [("C5_gru", (256, 1024, 50), "gru", 20, 3),
                                                                      ("C2_predictor_ODE", (E, N, H), "ode:ODE", 20, 3)]
        for name, (e_, n_, h_), pred, steps_, warm_ in side:
            pred, ptype = (pred.split(":") + ["ODE_v0"])[:2]
            try:
                w = Workload(ctx, e_, n_, h_, predictor=pred, predictor_type=ptype)
                rr = w.run(steps_, warm_)
            except Exception as ex:  # noqa: BLE001
                if world > 1:
                    raise                                  # (the other ranks are inside the same collective sequence)
                # a side configuration must not cost the run its headline line: say what failed, go on
                extras[name] = {"error": f"{type(ex).__name__}: {ex}"}
                side_failed.append(name)
                continue
            impl = w.collective_impl
            if rank == 0 and not args.no_verify and world == 1:
                to_verify.append((name, w))                # (closed after its verification)
            else:
                # several ranks: every rank tears its communicator down at the same point of the run (rank 0 checks its side
                # configuration first; no host-paced single-env loop follows in that case)
                try:
                    if rank == 0 and not args.no_verify:
                        verified[name] = w.verify()
                except Exception as ex:  # noqa: BLE001
                    verified[name] = {"ok": False, "error": f"{type(ex).__name__}: {ex}"}
                finally:
                    w.close()
            obj = {"workload": f"{e_} envs per GPU x {n_} samples x {h_}-step horizon, {steps_} steps after {warm_}"
                               + ("" if ptype == "ODE_v0" else ", predictor_ODE (Euler-Cromer, no edge bounce)"),
                   "value": rr["value"], "unit": "rollouts/s", "n_gpus": world, "ms_per_step": rr["ms_per_step"],
                   "kernel_ms": rr["kernel_ms"], "kernel_ms_min": rr["kernel_ms_min"],
                   "kernel_launches_timed": rr["kernel_launches_timed"], "kernel_event_group": rr["kernel_event_group"]}
            if impl:
                obj["collective"] = impl
            if pred == "ode":
                obj["roofline_valu"] = roofline_valu(rr, e_, n_, h_)
            else:
                gru_flops = 2.0 * (3 * 32 * 38 + 3 * 32 * 64 + 5 * 32) * h_ * e_ * n_
                obj["roofline"] = {"bound": "mfma", "achieved": gru_flops / (rr["kernel_ms"] * 1e-3) / 1e12,
                                   "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": gru_flops / (rr["kernel_ms"] * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS,
                                   "note": "useful GRU flops against the dense f16 MFMA peak (split-f16 products issue 3.6x these)"}
            extras[name] = obj
        if world == 1 or backend == "nccl" or os.environ.get("CPMPPI_BENCH_RCCL_PATH"):
            # the same small configurations with their envs in independent groups, each on its own stream (pipeline.py): what the
            # share-nothing structure of the problem allows and one launch per step cannot use.  Under --gpus N (or
            # CPMPPI_BENCH_FORCE_COLLECTIVE=1) the groups run under ONE communicator with the per-step all-gather
            # (cpmppi_groups_run_gather): the 8-GPU shape of BASELINE configs[3] is this form
            pipelined = (("C4_pipelined", "C4", 2, 200, 220), ("C3_pipelined", "C3", 2, 100, 110)) if world == 1 else (("C4_pipelined", "C4", 2, 200, 220),)
            for name, base, groups, steps_, warm_ in pipelined:
                e_, n_, h_ = PRESETS[base]
                try:
                    # the yardstick first: the SAME code path with ONE group = one launch per step, enqueued from C, wall time only
                    one = GroupedWorkload(ctx, e_, n_, h_, 1)
                    one_ms = one.run(steps_, warm_, overlap=False)["ms_per_step"]
                    one.close()
                    plain_ms = None
                    if collective:
                        # ... and the grouped form WITHOUT the collective, same process, same box: what the gather costs
                        plain = GroupedWorkload(dict(ctx, collective=False), e_, n_, h_, groups)
                        plain_ms = plain.run(steps_, warm_, overlap=False)["ms_per_step"]
                        plain.close()
                    gw = GroupedWorkload(ctx, e_, n_, h_, groups)
                    rr = gw.run(steps_, warm_)
                except Exception as ex:  # noqa: BLE001
                    if world > 1 and not isinstance(ex, CollectiveUnavailable):
                        raise                              # (the other ranks are inside the same collective sequence)
                    extras[name] = {"error": f"{type(ex).__name__}: {ex}"}
                    side_failed.append(name)
                    continue
                if not args.no_verify and rank == 0 and world == 1:
                    to_verify.append((name, gw))
                else:
                    try:
                        if rank == 0 and not args.no_verify:
                            verified[name] = gw.verify()
                    except Exception as ex:  # noqa: BLE001
                        verified[name] = {"ok": False, "error": f"{type(ex).__name__}: {ex}"}
                    finally:
                        gw.close()
                extras[name] = {"workload": f"{e_} envs per GPU x {n_} samples x {h_}-step horizon as {groups} independent env groups of "
                                            f"{e_ // groups}, each its own handle, stream and chain of steps; {steps_} steps after {warm_}",
                                "value": rr["value"], "unit": "rollouts/s", "n_gpus": world, "ms_per_step": rr["ms_per_step"],
                                "groups": groups, "group_kernel_ms": rr["group_kernel_ms"], "kernels": rr["kernels"],
                                "stream_overlap": rr["stream_overlap"],
                                **({"collective": gw.collective_report or {"impl": gw.collective_impl},
                                    "without_collective_ms_per_step": plain_ms, "collective_cost": rr["ms_per_step"] / plain_ms}
                                   if gw.collective_impl else {}),
                                "one_group_ms_per_step": one_ms, "vs_one_launch_per_step": one_ms / rr["ms_per_step"],
                                "roofline_valu": {"bound": "fp32-valu", "unit": "TFLOP/s", "peak": FP32_VALU_PEAK_TFLOPS,
                                                  "achieved": algorithmic_flops_per_rollout(h_) * e_ * n_ / (rr["ms_per_step"] * 1e-3) / 1e12,
                                                  "frac": algorithmic_flops_per_rollout(h_) * e_ * n_ / (rr["ms_per_step"] * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS,
                                                  "note": "from the WALL time per step of all groups (their kernels overlap: a kernel's own "
                                                          "duration, `group_kernel_ms`, says nothing about throughput here)"}}

    mark("side_configurations_timed")
    if rank == 0:
        k_ms = r["kernel_ms"]
        traffic, traffic_src, traffic_date = profiled_traffic(args.noise, E, N, H)
        roof = None
        if args.predictor == "gru":
            # useful flops of the GRU-6IN-32H1-32H2-5OUT forward per rollout-step: 2 x (3*32*(6+32) + 3*32*(32+32) + 5*32)
            gru_flops = 2.0 * (3 * 32 * 38 + 3 * 32 * 64 + 5 * 32) * H * E * N
            if args.math == "fast":
                # FAST: float32-equivalent products as 3 f16 MFMAs (hi*hi + hi*lo + lo*hi), dense f16 peak ~2.5 PFLOP/s
                peak, issued = F16_MFMA_PEAK_TFLOPS, 69 * 2.0 * 32 * 32 * 16 / 32.0 * H * E * N
                note = ("split-f16 products on v_mfma_f32_32x32x16_f16 (69 MFMAs per 32-rollout tile step, 3.6x the algorithmic "
                        "flops); the matrix pipe overlaps with the gate math (exp, rcp), which bounds the "
                        "kernel; --math precise runs the exact-f32 MFMA kernel")
            else:
                peak, issued = FP32_VALU_PEAK_TFLOPS, 156 * 2.0 * 32 * 32 * 2 / 32.0 * H * E * N
                note = ("f32-input MFMA (v_mfma_f32_32x32x2_f32): dense peak = the fp32 vector peak, 157.3 TFLOP/s; this MFMA "
                        "does not co-execute with vector instructions, so gate math adds to it")
            roof = {"bound": "mfma", "kernel": "gru_rollout_cost_kernel", "achieved": gru_flops / (k_ms * 1e-3) / 1e12,
                    "peak": peak, "unit": "TFLOP/s", "frac": gru_flops / (k_ms * 1e-3) / 1e12 / peak, "traffic": traffic,
                    "issued_mfma_tflops": issued / (k_ms * 1e-3) / 1e12,
                    "kernel_ms": k_ms, "finalize_kernel_ms": r["finalize_kernel_ms"],
                    "kernel_launches_timed": r["kernel_launches_timed"], "note": note}
        out = {
            "metric": f"MPPI rollouts/sec ({N} samples x {H}-step horizon)", "value": r["value"], "unit": "rollouts/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config or 'C2'}-shape MPPI problems: {N} samples x {H}-step horizon x 10 Euler "
                                   f"substeps, {E} independent envs per GPU batched in one launch"
                                   + ("" if args.config in ("C3", "C4") else " (BASELINE configs[1] shape)"),
                       "envs_per_gpu": E, "rollouts": N, "horizon": H, "substeps": 10,
                       "cost": cfg.cost_function_specification, "noise": args.noise, "math": args.math,
                       "predictor": ("predictor_ODE_v0" if args.predictor_type == "ODE_v0" else
                                     "predictor_ODE (Euler-Cromer, no edge bounce: config_controllers.yml:3)") if args.predictor == "ode"
                       else "GRU-6IN-32H1-32H2-5OUT (synthetic weights)",
                       "parallelism": f"env-sharded x{world}, one RCCL all-gather of u_nom per step" if world > 1
                       else "single GPU",
                       **({"collective": main_wl.collective_report or {"impl": main_wl.collective_impl}} if main_wl.collective_impl else {})},
            "roofline": roof or {"bound": "hbm", "kernel": "rollout_cost_kernel", "achieved": r["alg_gbs"],
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["alg_gbs"] / HBM_PEAK_GBS,
                                 "traffic": traffic, "traffic_source": (f"profiled earlier, not in this run: {traffic_src}"
                                                                        + (f" ({traffic_date})" if traffic_date else ""))
                                 if traffic_src else None,
                                 "kernel_ms": k_ms, "finalize_kernel_ms": r["finalize_kernel_ms"],
                                 "kernel_launches_timed": r["kernel_launches_timed"],
                                 "algorithmic_bytes_per_rollout": algorithmic_bytes_per_rollout(N, H),
                                 "binding_roof": "roofline_valu",
                                 "note": "achieved = ALGORITHMIC bytes (SURVEY.md 8d) / kernel time, as the contract defines it; the "
                                         "path is fp32-VALU bound, not HBM bound (SURVEY.md F8: ~100 flop/B vs a machine balance of "
                                         "~20), so the binding roof is `roofline_valu`"
                                         + ("; with in-kernel Philox noise no perturbation buffer moves at all" if args.noise == "philox" else "")},
        }
        if args.predictor == "ode":
            rv = roofline_valu(r, E, N, H)
            # the same peak with every substep's sincos costed at ~30 flop-equivalents
            rv["survey_ceiling_rollouts_per_s"] = FP32_VALU_PEAK_TFLOPS * 1e12 / (algorithmic_flops_per_rollout(H) + 2.0 * 10 * H * 30.0)
            out["roofline_valu"] = rv
        if extras:
            out["configs"] = extras
        if not args.no_single_env and world == 1:
            # latency of ONE problem instance (BASELINE configs[1] literally: single env), same kernels
            w1 = Workload(ctx, 1, N, H, noise="philox", math=args.math, predictor=args.predictor, predictor_type=args.predictor_type)
            e1 = w1.eng
            for i in range(5):
                w1.step(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 200
            for i in range(reps):
                w1.step(100 + i)
            torch.cuda.synchronize()
            dt1 = (time.perf_counter() - t1) / reps
            e1.set_profiling(True, group=10)             # (an event pair per launch would add ~10 us to each 60 us step)
            for i in range(50):
                w1.step(400 + i)
            r1, f1 = e1.get_profile()
            e1.set_profiling(False)
            w1.timed_kernel = e1.last_launch()["kernel"] if args.predictor == "ode" else "gru_rollout_cost_kernel"
            w1.next_step = 450
            k1 = float(np.median(r1))
            out["single_env"] = {"us_per_step": dt1 * 1e6, "rollouts_per_s": N / dt1, "noise": "philox",
                                 "rollout_kernel_us": k1 * 1e3, "finalize_kernel_us": float(np.median(f1)) * 1e3,
                                 "note": "host-paced python loop, one launch per step (finalize fused into the rollout kernel); kernel time = HIP events around groups of 10 launches / 10"}
            if args.predictor == "ode" and args.math == "fast":
                # the simulator's own call: state and attributes on the HOST, the control back on the host
                # (CartPole/__init__.py:509-520) through cpmppi_step_host - PCIe-inclusive, never `value`
                s_h = w1.s0.cpu().numpy().copy()
                tp_h, te_h, L_h = (x.cpu().numpy().copy() for x in (w1.tp, w1.te, w1.L))
                q_h = np.zeros(1, np.float32)
                u_h = e1.zeros(1, H)                     # (its own plan buffer: with the collective the workload's live elsewhere)
                for i in range(20):
                    e1.step_host(s_h, u_h, tp_h, te_h, L_h, w1.seed, 1000 + i, q_h)
                t2 = time.perf_counter()
                for i in range(reps):
                    e1.step_host(s_h, u_h, tp_h, te_h, L_h, w1.seed, 2000 + i, q_h)
                dt2 = (time.perf_counter() - t2) / reps
                out["single_env"]["host_seam"] = {
                    "us_per_call": dt2 * 1e6, "rollouts_per_s": N / dt2,
                    "note": "cpmppi_step_host: state / attributes read from pinned host memory by the kernel, the control "
                            "delivered into it with a system-scope ticket the caller spins on (no copies, no stream wait)"}
            if args.predictor == "ode":
                out["single_env"]["roofline_valu"] = {
                    "bound": "fp32-valu", "achieved": algorithmic_flops_per_rollout(H) * N / (k1 * 1e-3) / 1e12,
                    "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": algorithmic_flops_per_rollout(H) * N / (k1 * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS,
                    "note": f"{N} rollouts = {N // 64} waves on 1024 SIMDs: latency of one wave's dependency chain, not throughput"}
            if not args.no_verify:
                to_verify.append(("single_env", w1))
            else:
                w1.close()
        # ---- every timed region is over: the checker's turn
        mark("single_env_timed")
        prebuild = None
        if not args.no_cpu_baseline and world == 1:
            # the CPU baseline's timing builds (gcc, ~2.5 s) compile in the background while the checker works
            import threading
            from oracle import oracle_c as _OC

            def _compile():
                try:
                    _OC.compile_bench_variants()
                except Exception:                            # noqa: BLE001  (cpu_baseline reports a build error itself)
                    pass
            prebuild = threading.Thread(target=_compile, daemon=True)
            prebuild.start()
        if not args.no_verify:
            for name, w in to_verify:
                # a checker that cannot run (no C compiler for the oracle, an exception on its side) must not cost the bench line:
                # it is recorded as a failed verification and the process still exits 3 (advisor, round 4)
                try:
                    verified[name] = w.verify_regimes() if name == "single_env" and w.predictor == "ode" else w.verify()
                except Exception as ex:  # noqa: BLE001
                    verified[name] = {"ok": False, "error": f"{type(ex).__name__}: {ex}"}
                finally:
                    if w is not main_wl:
                        w.close()
                mark("verified_" + name)
            out["verified"] = verified["main"]
            for name, v in verified.items():
                if name == "single_env":
                    out["single_env"]["verified"] = v
                elif name != "main" and name in out.get("configs", {}):
                    out["configs"][name]["verified"] = v
        mark("verified")
        if not args.no_cpu_baseline and world == 1:          # reported at N = 1 only (bench contract)
            if prebuild is not None:
                prebuild.join()
            out["cpu_baseline"] = cpu_baseline(N, H, integrator=args.predictor_type, full=args.cpu_baseline == "full")
        mark("cpu_baseline_done")
        out["wall_s"] = phases                               # cumulative wall seconds at the end of each phase of this process
        print(json.dumps(out), flush=True)
    if in_rank:
        dist.barrier()
        dist.destroy_process_group()
    bad = [k for k, v in verified.items() if not v["ok"]]
    if bad:                                               # a timed configuration whose results the oracle does not confirm
        print(f"bench.py: verification FAILED for {bad}: " + json.dumps({k: verified[k] for k in bad}), file=sys.stderr, flush=True)
        sys.exit(3)
    if side_failed:                                       # the line was printed (its `configs` entries carry the error); not a clean run
        print(f"bench.py: side configurations FAILED to run: " + json.dumps({k: extras[k] for k in side_failed}), file=sys.stderr, flush=True)
        sys.exit(4)


if __name__ == "__main__":
    main()
