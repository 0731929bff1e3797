"""optimizer_gradient / optimizer_rpgd — the gradient-based optimizers over the adjoint of the rollout + cost kernel
(SURVEY.md §8f N4).

Constructor keywords = the keys of ``Control_Toolkit_ASF/config_optimizers.yml:49-62`` (section ``gradient-tf``) and
``:63-86`` (section ``rpgd``).  The classes themselves live in the absent Control_Toolkit submodule (TensorFlow
GradientTape through predictor + cost function, Keras Adam); their behaviour here is the one those keys name and is NOT
pinned by anything in-tree ([recalled] in SURVEY.md Appendix B terms):

* both keep ``num_rollouts`` candidate input plans per env and improve ALL of them in parallel with Adam on
  d cost / d inputs (cpmppi_rollout_cost_grad + cpmppi_adam_step: per-plan gradient-norm clipping to ``gradmax_clip``,
  clip to the control limits), then apply the first input of the cheapest plan and shift every plan;
* ``gradient``: ``gradient_steps`` iterations per control step, plans initialised from N(0, ``initial_action_stdev``);
* ``rpgd`` (resampling parallel gradient descent): ``outer_its`` iterations per control step; every ``resamp_per``
  control steps only the best ``opt_keep_k_ratio`` of the plans survive (with their Adam moments), the others are
  re-drawn from the sampling distribution (``SAMPLING_DISTRIBUTION`` normal(``sample_mean``, ``sample_stdev``) or
  uniform, one random point every ``period_interpolation_inducing_points`` steps, linear in between);
  plans are shifted by ``shift_previous``.

Everything that scales with rollouts x horizon runs in HIP kernels (sampling, forward + reverse sweep, Adam, top-k);
torch is used for the per-plan bookkeeping (gather of survivors, shift).  ``num_envs`` problem instances advance in one
launch.
"""
import math
import time as _time

import numpy as np
import torch

from .configs import MPPIConfig, PhysicalParameters
from .optimizer_mppi import _vec


class _GradientBase:
    optimizer_name = "gradient"

    def _setup(self, cost_function, control_limits, seed, mpc_horizon, mpc_timestep, num_rollouts, sample_stdev, period,
               num_envs, cost_function_specification, cost_weights, intermediate_steps, phys, device,
               variable_parameters, optimizer_logging, horizon_reduce):
        low, high = (-1.0, 1.0) if control_limits is None else (float(np.asarray(control_limits[0]).reshape(-1)[0]),
                                                                  float(np.asarray(control_limits[1]).reshape(-1)[0]))
        self.action_low, self.action_high = low, high
        if seed is None:
            import os
            seed = (_time.time_ns() ^ os.getpid()) & 0x7FFFFFFFFFFFFFFF
        self.seed = int(seed)
        self.num_envs = int(num_envs)
        if cost_function is not None and cost_function_specification is None:
            cost_function_specification = getattr(cost_function, "cost_name", None)
            cost_weights = cost_weights or getattr(cost_function, "weights", None)
        self.variable_parameters = variable_parameters if variable_parameters is not None else \
            getattr(cost_function, "variable_parameters", None)
        dt = float(mpc_timestep)
        # the handle's sampler draws knots ~ N(0, SQRTRHOINV / sqrt(dt)): set it to the requested stdev
        self.cfg = MPPIConfig(seed=self.seed, mpc_horizon=int(mpc_horizon), mpc_timestep=dt,
                              num_rollouts=int(num_rollouts), intermediate_steps=int(intermediate_steps),
                              cost_function_specification=cost_function_specification or "quadratic_boundary_grad_minimal",
                              cost_weights=dict(cost_weights or {}), control_mode="clip", shift_mode="none",
                              math_mode="fast", action_low=low, action_high=high, horizon_reduce=horizon_reduce,
                              SQRTRHOINV=float(sample_stdev) * math.sqrt(dt),
                              period_interpolation_inducing_points=int(period))
        self.phys = phys or PhysicalParameters()
        self.device = device
        self.num_rollouts, self.mpc_horizon = self.cfg.num_rollouts, self.cfg.mpc_horizon
        self.optimizer_logging = optimizer_logging
        self.logging_values = {}
        self.engine = None
        self.count = 0               # control steps taken
        self.draws = 0               # sampler launches (the Philox offset)

    def configure(self, dt=None, predictor_specification=None, num_envs=None, **kwargs):
        from .engine import MPPIEngine
        if dt is not None and float(dt) != self.cfg.mpc_timestep:
            s = self.cfg.SQRTRHOINV / math.sqrt(self.cfg.mpc_timestep)
            self.cfg.mpc_timestep = float(dt)
            self.cfg.SQRTRHOINV = s * math.sqrt(float(dt))
        if num_envs is not None:
            self.num_envs = int(num_envs)
        spec = None if predictor_specification is None else str(predictor_specification).split(":")[0]
        if spec in ("ODE", "ODE_default"):      # next_state_predictor_ODE (Euler-Cromer, no bounce): the shipped config_controllers.yml:2-3
            self.cfg.predictor_type = "ODE"     # pairs it with `optimizer: rpgd`; the adjoint kernel has that substep's reverse too
        elif spec in ("ODE_v0", "ODE_v0_default"):
            self.cfg.predictor_type = "ODE_v0"
        elif spec is not None:
            raise NotImplementedError("the adjoint kernel differentiates the ODE_v0 and ODE predictors")
        self.engine = MPPIEngine(self.num_envs, self.cfg, self.phys, device=self.device)
        self.optimizer_reset()

    # -- sampling ------------------------------------------------------------------------------------------------
    def _draw(self):
        """[E,N,H] plans from the sampling distribution (device)."""
        _, z = self.engine.sample(self.seed, offset=self.draws, knots=False, delta_u=True)
        self.draws += 1
        return self._shape_samples(z).clamp_(self.action_low, self.action_high).contiguous()

    def _shape_samples(self, z):
        return z

    def optimizer_reset(self):
        self.Q = self._draw()
        self.m, self.v = torch.zeros_like(self.Q), torch.zeros_like(self.Q)
        self.adam_it = 0
        self.count = 0
        self._first = True

    # -- one control step ----------------------------------------------------------------------------------------
    def _targets(self, E):
        vp = self.variable_parameters
        self.engine.apply_pole_mass_of(vp)
        t = self.engine.tensor      # (uploaded once per control step; every gradient / cost launch below reuses the tensors)
        return (t(_vec(getattr(vp, "target_position", None), E, 0.0)), t(_vec(getattr(vp, "target_equilibrium", None), E, 1.0)),
                t(_vec(getattr(vp, "L", None), E, self.phys.L)))

    def _descend(self, s_t, tp, te, L, iterations):
        eng = self.engine
        for _ in range(iterations):
            _, G = eng.rollout_cost_grad(s_t, self.Q, tp, te, L=L, previous_input=self._previous_input)
            self.adam_it += 1
            eng.adam_step(self.Q, G, self.m, self.v, self.adam_it, self.learning_rate, self.adam_beta_1, self.adam_beta_2,
                          self.adam_epsilon, self.gradmax_clip)
        return eng.rollout_cost(s_t, self.Q, tp, te, L=L) if self.cfg.cost_function_specification != "quadratic_boundary_grad" \
            else eng.rollout_cost_grad(s_t, self.Q, tp, te, L=L, previous_input=self._previous_input)[0]

    _previous_input = None

    def _shift(self, by):
        if by <= 0:
            return
        for name in ("Q", "m", "v"):
            x = getattr(self, name)
            tail = x[:, :, -1:].expand(-1, -1, by) if name == "Q" else torch.zeros_like(x[:, :, :by])
            setattr(self, name, torch.cat([x[:, :, by:], tail], dim=2).contiguous())

    def _finish(self, S, single, as_tensor):
        E = S.shape[0]
        best = torch.argmin(S, dim=1)
        rows = torch.arange(E, device=S.device)
        u = self.Q[rows, best, 0].clone()
        self._previous_input = u.clone()
        if self.optimizer_logging:
            self.logging_values = {"Q_logged": u.cpu().numpy(), "J_logged": S.cpu().numpy(),
                                   "u_logged": self.Q[rows, best].cpu().numpy()}
        self.count += 1
        if as_tensor:
            return u
        q = u.cpu().numpy()
        return q[:1].copy() if single else q.reshape(E, 1).copy()

    def _state(self, s):
        if self.engine is None:
            self.configure()
        s_t = self.engine.tensor(s)
        single = s_t.dim() == 1
        s_t = s_t.reshape(-1, 6)
        if s_t.shape[0] != self.num_envs:
            raise ValueError(f"optimizer configured for {self.num_envs} envs, got {s_t.shape[0]} states")
        return s_t, single


class optimizer_gradient(_GradientBase):
    """config_optimizers.yml:49-62 (gradient-tf)."""
    optimizer_name = "gradient"

    def __init__(self, predictor=None, cost_function=None, control_limits=None, computation_library=None, seed=None,
                 mpc_horizon=35, mpc_timestep=0.02, learning_rate=0.05, adam_beta_1=0.9, adam_beta_2=0.999,
                 adam_epsilon=1.0e-7, rtol=1.0e-3, gradient_steps=5, num_rollouts=40, initial_action_stdev=0.5,
                 gradmax_clip=5, warmup=False, warmup_iterations=250, optimizer_logging=False,
                 calculate_optimal_trajectory=False, num_envs=1, cost_function_specification=None, cost_weights=None,
                 intermediate_steps=10, horizon_reduce="sum", phys=None, device=0, variable_parameters=None, **kwargs):
        self.learning_rate, self.adam_beta_1, self.adam_beta_2 = float(learning_rate), float(adam_beta_1), float(adam_beta_2)
        self.adam_epsilon, self.gradmax_clip, self.rtol = float(adam_epsilon), float(gradmax_clip), float(rtol)
        self.gradient_steps, self.warmup, self.warmup_iterations = int(gradient_steps), bool(warmup), int(warmup_iterations)
        self.initial_action_stdev = float(initial_action_stdev)
        self._setup(cost_function, control_limits, seed, mpc_horizon, mpc_timestep, num_rollouts, initial_action_stdev, 10,
                    num_envs, cost_function_specification, cost_weights, intermediate_steps, phys, device,
                    variable_parameters, optimizer_logging, horizon_reduce)

    def _draw(self):
        """Independent N(0, initial_action_stdev) per time-step, clipped (cpmppi_cem_sample)."""
        eng = self.engine
        mid = 0.5 * (self.action_low + self.action_high)
        Q = eng.cem_sample(eng.zeros(self.num_envs, self.mpc_horizon) + mid,
                           eng.zeros(self.num_envs, self.mpc_horizon) + self.initial_action_stdev, self.seed, offset=self.draws)
        self.draws += 1
        return Q

    def step(self, s, time=None, as_tensor=False):
        s_t, single = self._state(s)
        tp, te, L = self._targets(s_t.shape[0])
        iters = self.warmup_iterations if (self.warmup and self._first) else self.gradient_steps
        self._first = False
        S = self._descend(s_t, tp, te, L, iters)
        out = self._finish(S, single, as_tensor)
        self._shift(1)
        return out


class optimizer_rpgd(_GradientBase):
    """config_optimizers.yml:63-86 (rpgd)."""
    optimizer_name = "rpgd"

    def __init__(self, predictor=None, cost_function=None, control_limits=None, computation_library=None, seed=None,
                 mpc_horizon=35, mpc_timestep=0.02, SAMPLING_DISTRIBUTION="normal", period_interpolation_inducing_points=4,
                 learning_rate=0.05, adam_beta_1=0.9, adam_beta_2=0.999, adam_epsilon=1.0e-8, gradmax_clip=5, rtol=1.0e-3,
                 num_rollouts=16, opt_keep_k_ratio=0.75, outer_its=4, resamp_per=10, sample_stdev=0.5, sample_mean=0.0,
                 sample_whole_control_space=False, uniform_dist_max=0.8, uniform_dist_min=-0.8, shift_previous=1,
                 warmup=False, warmup_iterations=250, optimizer_logging=False, calculate_optimal_trajectory=False,
                 num_envs=1, cost_function_specification=None, cost_weights=None, intermediate_steps=10,
                 horizon_reduce="sum", phys=None, device=0, variable_parameters=None, **kwargs):
        if SAMPLING_DISTRIBUTION not in ("normal", "uniform"):
            raise ValueError(f"SAMPLING_DISTRIBUTION={SAMPLING_DISTRIBUTION!r}; expected 'normal' or 'uniform'")
        self.learning_rate, self.adam_beta_1, self.adam_beta_2 = float(learning_rate), float(adam_beta_1), float(adam_beta_2)
        self.adam_epsilon, self.gradmax_clip, self.rtol = float(adam_epsilon), float(gradmax_clip), float(rtol)
        self.outer_its, self.resamp_per, self.shift_previous = int(outer_its), int(resamp_per), int(shift_previous)
        self.warmup, self.warmup_iterations = bool(warmup), int(warmup_iterations)
        self.distribution, self.sample_mean, self.sample_stdev = SAMPLING_DISTRIBUTION, float(sample_mean), float(sample_stdev)
        self.sample_whole_control_space = bool(sample_whole_control_space)
        self.uniform_dist_min, self.uniform_dist_max = float(uniform_dist_min), float(uniform_dist_max)
        self.opt_keep_k = max(1, int(float(opt_keep_k_ratio) * int(num_rollouts)))
        stdev = self.sample_stdev if SAMPLING_DISTRIBUTION == "normal" else 1.0
        self._setup(cost_function, control_limits, seed, mpc_horizon, mpc_timestep, num_rollouts, stdev,
                    period_interpolation_inducing_points, num_envs, cost_function_specification, cost_weights,
                    intermediate_steps, phys, device, variable_parameters, optimizer_logging, horizon_reduce)

    def _shape_samples(self, z):
        if self.distribution == "normal":
            return z + self.sample_mean if self.sample_mean != 0.0 else z
        lo, hi = (self.action_low, self.action_high) if self.sample_whole_control_space else \
            (self.uniform_dist_min, self.uniform_dist_max)
        return lo + (hi - lo) * 0.5 * (1.0 + torch.erf(z * (1.0 / math.sqrt(2.0))))     # N(0,1) -> U(lo, hi)

    def step(self, s, time=None, as_tensor=False):
        s_t, single = self._state(s)
        E = s_t.shape[0]
        tp, te, L = self._targets(E)
        iters = self.warmup_iterations if (self.warmup and self._first) else self.outer_its
        self._first = False
        S = self._descend(s_t, tp, te, L, iters)
        out = self._finish(S, single, as_tensor)
        if self.resamp_per > 0 and self.count % self.resamp_per == 0 and self.opt_keep_k < self.num_rollouts:
            # survivors: the opt_keep_k cheapest plans (stable top-k on the device), moments kept; the rest re-drawn
            _, _, elite = self.engine.cem_update(S, self.Q, self.opt_keep_k, 0.0, return_elites=True)
            idx = elite.long().unsqueeze(-1).expand(-1, -1, self.mpc_horizon)
            fresh = self._draw()
            k = self.opt_keep_k
            for name in ("Q", "m", "v"):
                x = getattr(self, name)
                rest = fresh[:, k:] if name == "Q" else torch.zeros_like(x[:, k:])
                setattr(self, name, torch.cat([torch.gather(x, 1, idx), rest], dim=1).contiguous())
        self._shift(self.shift_previous)
        return out
