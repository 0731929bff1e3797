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
        spec = None if predictor_specification is None else str(predictor_specification).split(":")[0]
        if spec in ("ODE", "ODE_default"):        # next_state_predictor_ODE: Euler-Cromer, no bounce (config_controllers.yml:3)
            self.cfg.predictor_type = "ODE"
        elif spec in ("ODE_v0", "ODE_v0_default"):
            self.cfg.predictor_type = "ODE_v0"
        elif spec is not None:
            raise NotImplementedError("the sampling optimizers run on the ODE_v0 and ODE predictors")
        self.engine = MPPIEngine(self.num_envs, self.cfg, self.phys, device=self.device)
        self.optimizer_reset()

    def _refine(self, Q, s_t, tp, te, L):
        """Hook of the CEM + gradient hybrids: improve the samples before they are ranked."""
        return Q

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
        eng.apply_pole_mass_of(self.variable_parameters)
        s_t = eng.tensor(s)
        single = s_t.dim() == 1
        s_t = s_t.reshape(-1, 6)
        E = s_t.shape[0]
        if E != self.num_envs:
            raise ValueError(f"optimizer configured for {self.num_envs} envs, got {E} states")
        vp = self.variable_parameters
        # (uploaded ONCE per control step: every sampler / cost / gradient launch of the iterations below reuses the tensors)
        tp = eng.tensor(_vec(getattr(vp, "target_position", None), E, 0.0))
        te = eng.tensor(_vec(getattr(vp, "target_equilibrium", None), E, 1.0))
        L = eng.tensor(_vec(getattr(vp, "L", None), E, self.phys.L))
        iters = self.warmup_iterations if (self.warmup and self._first) else self.cem_outer_it
        self._first = False
        for _ in range(iters):
            Q = eng.cem_sample(self.dist_mue, self.stdev, self.seed, offset=self.step_counter)
            Q = self._refine(Q, s_t, tp, te, L)
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


class optimizer_cem_gmm(optimizer_cem):
    """``cem-gmm-tf`` (config_optimizers.yml:12-20): CEM whose sampling distribution is a Gaussian MIXTURE — one
    component per elite of the previous iteration, centred on that elite sequence, equal weights, all sharing the
    per-time-step standard deviation refitted to the elites (floored at ``cem_stdev_min``).  The first iteration after a
    reset samples around the mid-point sequence with ``cem_initial_action_stdev`` (a single component).  Applied control:
    the first input of the best sequence found; the elites are shifted by one step for the next control step.
    [recalled semantics: the class lives in the absent Control_Toolkit submodule; unpinned, stated in DESIGN.md]"""
    optimizer_name = "cem-gmm"

    def optimizer_reset(self):
        super().optimizer_reset()
        self.centres = self.dist_mue[:, None, :].contiguous()           # [E, 1, H]: one component until elites exist

    def step(self, s, time=None, as_tensor=False):
        if self.engine is None:
            self.configure()
        eng = self.engine
        eng.apply_pole_mass_of(self.variable_parameters)
        s_t = eng.tensor(s)
        single = s_t.dim() == 1
        s_t = s_t.reshape(-1, 6)
        E = s_t.shape[0]
        if E != self.num_envs:
            raise ValueError(f"optimizer configured for {self.num_envs} envs, got {E} states")
        vp = self.variable_parameters
        # (uploaded ONCE per control step: every sampler / cost / gradient launch of the iterations below reuses the tensors)
        tp = eng.tensor(_vec(getattr(vp, "target_position", None), E, 0.0))
        te = eng.tensor(_vec(getattr(vp, "target_equilibrium", None), E, 1.0))
        L = eng.tensor(_vec(getattr(vp, "L", None), E, self.phys.L))
        ar = torch.arange(E, device=self.dist_mue.device)[:, None]
        for _ in range(self.cem_outer_it):
            Q = eng.cem_gmm_sample(self.centres, self.stdev, self.seed, offset=self.step_counter)
            S = eng.rollout_cost(s_t, Q, tp, te, L=L)
            self.dist_mue, self.stdev, el = eng.cem_update(S, Q, self.cem_best_k, self.cem_stdev_min, return_elites=True)
            self.centres = Q[ar, el.long()].contiguous()                # [E, K, H], cheapest first (stable order)
            self.step_counter += 1
        u = self.centres[:, 0, 0].clone()                               # first input of the best sequence
        if self.optimizer_logging:
            self.logging_values = {"Q_logged": u.cpu().numpy(), "J_logged": S.cpu().numpy(),
                                   "u_logged": self.centres[:, 0].cpu().numpy()}
        mid = 0.5 * (self.action_low + self.action_high)
        self.centres = torch.cat([self.centres[:, :, 1:], torch.full_like(self.centres[:, :, :1], mid)], dim=2).contiguous()
        self.dist_mue = torch.cat([self.dist_mue[:, 1:], torch.full_like(self.dist_mue[:, :1], mid)], dim=1).contiguous()
        self.stdev = torch.cat([self.stdev[:, 1:], torch.full_like(self.stdev[:, :1], float(np.sqrt(0.5)))], dim=1).contiguous()
        if as_tensor:
            return u
        q = u.cpu().numpy()
        return q[:1].copy() if single else q.reshape(E, 1).copy()


class optimizer_cem_naive_grad(optimizer_cem):
    """``cem-naive-grad-tf`` (config_optimizers.yml:21-31): CEM whose samples take ONE plain gradient step
    ``Q <- clip(Q - learning_rate * clip_by_norm(dJ/dQ, gradmax_clip))`` (cpmppi_rollout_cost_grad + cpmppi_sgd_step)
    before the elites are chosen.  [recalled semantics, class absent from the tree]"""
    optimizer_name = "cem-naive-grad"

    def __init__(self, *args, learning_rate=0.1, gradmax_clip=10, cem_outer_it=1, cem_stdev_min=0.1, **kwargs):
        super().__init__(*args, cem_outer_it=cem_outer_it, cem_stdev_min=cem_stdev_min, **kwargs)
        self.learning_rate, self.gradmax_clip = float(learning_rate), float(gradmax_clip)

    def _refine(self, Q, s_t, tp, te, L):
        _, G = self.engine.rollout_cost_grad(s_t, Q, tp, te, L=L)
        return self.engine.sgd_step(Q, G, self.learning_rate, self.gradmax_clip)


class optimizer_cem_grad_bharadhwaj(optimizer_cem):
    """``cem-grad-bharadhwaj-tf`` (config_optimizers.yml:32-48; Bharadhwaj et al. 2020, "Model-predictive control via
    cross-entropy and gradient-based optimization"): every CEM sample takes an Adam step on dJ/dQ before the elites
    are chosen; the Adam moments belong to the sample slots and persist over the outer iterations of a control step.
    [recalled semantics, class absent from the tree]"""
    optimizer_name = "cem-grad-bharadhwaj"

    def __init__(self, *args, learning_rate=0.05, adam_beta_1=0.9, adam_beta_2=0.999, adam_epsilon=1.0e-8, num_rollouts=32,
                 cem_best_k=8, cem_outer_it=2, cem_initial_action_stdev=2, cem_stdev_min=1.0e-6, gradmax_clip=5, **kwargs):
        super().__init__(*args, num_rollouts=num_rollouts, cem_best_k=cem_best_k, cem_outer_it=cem_outer_it,
                         cem_initial_action_stdev=cem_initial_action_stdev, cem_stdev_min=cem_stdev_min, **kwargs)
        self.learning_rate, self.adam_beta_1, self.adam_beta_2 = float(learning_rate), float(adam_beta_1), float(adam_beta_2)
        self.adam_epsilon, self.gradmax_clip = float(adam_epsilon), float(gradmax_clip)

    def step(self, s, time=None, as_tensor=False):
        self._m = self._v = None                    # fresh moments every control step
        self._it = 0
        return super().step(s, time, as_tensor)

    def _refine(self, Q, s_t, tp, te, L):
        if self._m is None:
            self._m, self._v = torch.zeros_like(Q), torch.zeros_like(Q)
        _, G = self.engine.rollout_cost_grad(s_t, Q, tp, te, L=L)
        self._it += 1
        return self.engine.adam_step(Q, G, self._m, self._v, self._it, self.learning_rate, self.adam_beta_1, self.adam_beta_2,
                                     self.adam_epsilon, self.gradmax_clip)


class optimizer_random_action(optimizer_cem):
    """``random-action-tf`` (config_optimizers.yml:98-102): ``num_rollouts`` input plans drawn uniformly from the control
    limits every control step, the first input of the cheapest one is applied; nothing is carried over.
    [recalled semantics, class absent from the tree]"""
    optimizer_name = "random-action"

    def __init__(self, *args, num_rollouts=640, **kwargs):
        kwargs.pop("cem_outer_it", None)
        super().__init__(*args, num_rollouts=num_rollouts, cem_outer_it=1, cem_best_k=1, **kwargs)

    def step(self, s, time=None, as_tensor=False):
        import math
        if self.engine is None:
            self.configure()
        eng = self.engine
        eng.apply_pole_mass_of(self.variable_parameters)
        s_t = eng.tensor(s)
        single = s_t.dim() == 1
        s_t = s_t.reshape(-1, 6)
        E = s_t.shape[0]
        if E != self.num_envs:
            raise ValueError(f"optimizer configured for {self.num_envs} envs, got {E} states")
        vp = self.variable_parameters
        # (uploaded ONCE per control step: every sampler / cost / gradient launch of the iterations below reuses the tensors)
        tp = eng.tensor(_vec(getattr(vp, "target_position", None), E, 0.0))
        te = eng.tensor(_vec(getattr(vp, "target_equilibrium", None), E, 1.0))
        L = eng.tensor(_vec(getattr(vp, "L", None), E, self.phys.L))
        # N(0,1) from the device Philox sampler (clip limits far away), mapped to U(low, high) through the normal CDF
        wide = eng.zeros(E, self.mpc_horizon)
        z = self._normal(wide, self.step_counter)
        lo, hi = self.action_low, self.action_high
        Q = (lo + (hi - lo) * 0.5 * (1.0 + torch.erf(z * (1.0 / math.sqrt(2.0))))).clamp_(lo, hi).contiguous()
        self.step_counter += 1
        S = eng.rollout_cost(s_t, Q, tp, te, L=L)
        best = torch.argmin(S, dim=1)
        u = Q[torch.arange(E, device=S.device), best, 0].clone()
        if self.optimizer_logging:
            self.logging_values = {"Q_logged": u.cpu().numpy(), "J_logged": S.cpu().numpy()}
        if as_tensor:
            return u
        q = u.cpu().numpy()
        return q[:1].copy() if single else q.reshape(E, 1).copy()

    def _normal(self, zeros_EH, offset):
        # the sampler clips to the control limits, so draw with a small stdev and rescale: z = (x / 0.01), |x| <= 1 keeps
        # |z| <= 100, i.e. no clipping of a standard normal
        x = self.engine.cem_sample(zeros_EH + 0.5 * (self.action_low + self.action_high),
                                   zeros_EH + 0.01 * 0.5 * (self.action_high - self.action_low), self.seed, offset=offset)
        return (x - 0.5 * (self.action_low + self.action_high)) / (0.01 * 0.5 * (self.action_high - self.action_low))
