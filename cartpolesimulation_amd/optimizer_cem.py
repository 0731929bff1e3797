"""optimizer_cem — cross-entropy-method optimizer over the same fused rollout + cost kernel.

Constructor keywords = the keys of ``Control_Toolkit_ASF/config_optimizers.yml:1-11`` (section ``cem-tf``).  The class
itself lives in the absent Control_Toolkit submodule, so the update is the one those keys name: ``cem_outer_it`` times
{sample ``num_rollouts`` sequences from N(mean, stdev) clipped to the control limits, roll out + cost, refit mean and
stdev to the ``cem_best_k`` cheapest, floor stdev at ``cem_stdev_min``}; apply ``mean[0]``; shift mean (append the
mid-point of the limits) and stdev (append sqrt(0.5)).  Sampling, rollout, cost and the top-k refit all run on the GPU
(cpmppi_cem_sample, cpmppi_rollout_cost, cpmppi_cem_update); ``num_envs`` problem instances advance in one launch.
"""
import time as _time

import numpy as np
import torch

from .configs import MPPIConfig, PhysicalParameters
from .optimizer_mppi import _vec


class optimizer_cem:
    optimizer_name = "cem"

    def __init__(self, predictor=None, cost_function=None, control_limits=None, computation_library=None, seed=None,
                 mpc_horizon=35, mpc_timestep=0.02, cem_outer_it=3, cem_initial_action_stdev=0.5, num_rollouts=200,
                 cem_stdev_min=0.01, cem_best_k=40, warmup=False, warmup_iterations=250, optimizer_logging=False,
                 calculate_optimal_trajectory=False, num_envs=1, cost_function_specification=None, cost_weights=None,
                 math_mode="fast", intermediate_steps=10, phys=None, device=0, variable_parameters=None, **kwargs):
        low, high = (-1.0, 1.0) if control_limits is None else (float(np.asarray(control_limits[0]).reshape(-1)[0]),
                                                                  float(np.asarray(control_limits[1]).reshape(-1)[0]))
        self.action_low, self.action_high = low, high
        if seed is None:
            import os
            seed = (_time.time_ns() ^ os.getpid()) & 0x7FFFFFFFFFFFFFFF
        self.seed = int(seed)
        self.num_envs = int(num_envs)
        self.cem_outer_it, self.cem_best_k = int(cem_outer_it), int(cem_best_k)
        self.cem_initial_action_stdev, self.cem_stdev_min = float(cem_initial_action_stdev), float(cem_stdev_min)
        self.warmup, self.warmup_iterations = bool(warmup), int(warmup_iterations)
        if cost_function is not None and cost_function_specification is None:
            cost_function_specification = getattr(cost_function, "cost_name", None)
            cost_weights = cost_weights or getattr(cost_function, "weights", None)
        self.variable_parameters = variable_parameters if variable_parameters is not None else \
            getattr(cost_function, "variable_parameters", None)
        self.cfg = MPPIConfig(seed=self.seed, mpc_horizon=int(mpc_horizon), mpc_timestep=float(mpc_timestep),
                              num_rollouts=int(num_rollouts), intermediate_steps=int(intermediate_steps),
                              cost_function_specification=cost_function_specification or "quadratic_boundary_grad_minimal",
                              cost_weights=dict(cost_weights or {}), control_mode="clip", shift_mode="none",
                              math_mode=math_mode, action_low=low, action_high=high)
        self.phys = phys or PhysicalParameters()
        self.device = device
        self.num_rollouts, self.mpc_horizon = self.cfg.num_rollouts, self.cfg.mpc_horizon
        self.optimizer_logging = optimizer_logging
        self.logging_values = {}
        self.engine = None
        self.step_counter = 0

    def configure(self, dt=None, predictor_specification=None, num_envs=None, **kwargs):
        from .engine import MPPIEngine
        if dt is not None:
            self.cfg.mpc_timestep = float(dt)
        if num_envs is not None:
            self.num_envs = int(num_envs)
        if predictor_specification not in (None, "ODE_v0", "ODE_v0_default", "ODE", "ODE_default"):
            raise NotImplementedError("only the ODE_v0 predictor is built into the fused kernel on this tier")
        self.engine = MPPIEngine(self.num_envs, self.cfg, self.phys, device=self.device)
        self.optimizer_reset()

    def optimizer_reset(self):
        E, H = self.num_envs, self.mpc_horizon
        self.dist_mue = self.engine.zeros(E, H) + 0.5 * (self.action_low + self.action_high)
        self.stdev = self.engine.zeros(E, H) + self.cem_initial_action_stdev
        self.step_counter = 0
        self._first = True

    def step(self, s, time=None, as_tensor=False):
        if self.engine is None:
            self.configure()
        eng = self.engine
        s_t = eng.tensor(s)
        single = s_t.dim() == 1
        s_t = s_t.reshape(-1, 6)
        E = s_t.shape[0]
        if E != self.num_envs:
            raise ValueError(f"optimizer configured for {self.num_envs} envs, got {E} states")
        vp = self.variable_parameters
        tp = _vec(getattr(vp, "target_position", None), E, 0.0)
        te = _vec(getattr(vp, "target_equilibrium", None), E, 1.0)
        L = _vec(getattr(vp, "L", None), E, self.phys.L)
        iters = self.warmup_iterations if (self.warmup and self._first) else self.cem_outer_it
        self._first = False
        for _ in range(iters):
            Q = eng.cem_sample(self.dist_mue, self.stdev, self.seed, offset=self.step_counter)
            S = eng.rollout_cost(s_t, Q, tp, te, L=L)
            self.dist_mue, self.stdev = eng.cem_update(S, Q, self.cem_best_k, self.cem_stdev_min)
            self.step_counter += 1
        u = self.dist_mue[:, 0].clone()
        if self.optimizer_logging:
            self.logging_values = {"Q_logged": u.cpu().numpy(), "J_logged": S.cpu().numpy(),
                                   "u_logged": self.dist_mue.cpu().numpy()}
        mid = 0.5 * (self.action_low + self.action_high)
        self.dist_mue = torch.cat([self.dist_mue[:, 1:], torch.full_like(self.dist_mue[:, :1], mid)], dim=1).contiguous()
        self.stdev = torch.cat([self.stdev[:, 1:], torch.full_like(self.stdev[:, :1], float(np.sqrt(0.5)))], dim=1).contiguous()
        if as_tensor:
            return u
        q = u.cpu().numpy()
        return q[:1].copy() if single else q.reshape(E, 1).copy()
