"""controller_mpc — the controller seam of the reference, driving optimizer_mppi on the HIP path.

Call-site contract (CartPole/__init__.py:755-779,509-520,805,419; others/Tests/test_controller_mppi_tf.py:
10-16,34): ``Controller(environment_name, initial_environment_attributes, control_limits)``; ``configure(optimizer_name)``;
``step(s, time, updated_attributes) -> value castable by float()``; ``controller_reset()``; attributes
``has_optimizer``, ``optimizer.optimizer_name``, ``controller_data_for_csv``.
"""
from types import SimpleNamespace

import numpy as np

from .configs import MPPIConfig, PhysicalParameters
from .cost_functions import CostFunctionWrapper
from .optimizer_cem import (optimizer_cem, optimizer_cem_gmm, optimizer_cem_grad_bharadhwaj, optimizer_cem_naive_grad,
                            optimizer_random_action)
from .optimizer_gradient import optimizer_gradient, optimizer_rpgd
from .optimizer_mppi import optimizer_mppi
from .predictors import PredictorWrapper


class template_controller:
    def __init__(self, environment_name="CartPole", initial_environment_attributes=None, control_limits=None, **kwargs):
        self.environment_name = environment_name
        self.variable_parameters = SimpleNamespace(**(initial_environment_attributes or {}))
        if control_limits is None:
            control_limits = (np.array([-1.0], dtype=np.float32), np.array([1.0], dtype=np.float32))
        self.control_limits = control_limits
        self.action_low, self.action_high = control_limits
        self.controller_data_for_csv = {}
        self.has_optimizer = False

    def update_attributes(self, updated_attributes):
        for k, v in (updated_attributes or {}).items():
            setattr(self.variable_parameters, k, v)

    def controller_reset(self):
        raise NotImplementedError

    def controller_report(self):
        return None


class controller_mpc(template_controller):
    """``configure(optimizer_name="mppi")`` builds CostFunctionWrapper + PredictorWrapper-equivalent + optimizer;
    ``step`` = update_attributes -> optimizer.step -> logging.  Only the MPPI optimizer is built on this tier."""

    def __init__(self, environment_name="CartPole", initial_environment_attributes=None, control_limits=None,
                 action_space=None, observation_space=None, config=None, phys=None, device=0, num_envs=1,
                 config_root=None, **kwargs):
        if control_limits is None and action_space is not None:          # the gym-style ctor of others/Tests/*.py
            control_limits = (np.asarray(action_space.low, dtype=np.float32), np.asarray(action_space.high, dtype=np.float32))
        super().__init__(environment_name, initial_environment_attributes, control_limits)
        if environment_name != "CartPole":
            raise ValueError("only the CartPole environment is built")
        self.config_optimizer = dict(config or {})     # overrides of config_optimizers.yml:87-97 + glue flags
        self._user_config = dict(config or {})
        self._yaml = None                              # a checkout's YAML files (config_root): sections of the other optimizers
        if config_root is not None:                    # read a CartPoleSimulation checkout's YAML files
            from .configs import as_dict, load_reference_yaml, mppi_config_from_yaml
            yaml_phys, cfgs = load_reference_yaml(config_root)
            self._yaml = cfgs
            base = as_dict(mppi_config_from_yaml(cfgs))
            base.update(self.config_optimizer)
            self.config_optimizer = base
            phys = phys or yaml_phys
        self.phys = phys or PhysicalParameters()
        self.device, self.num_envs = device, num_envs
        self.has_optimizer = True
        self.optimizer = None
        self.cost_function_wrapper = None
        self.predictor = None
        self.controller_logging = False

    def configure(self, optimizer_name=None, predictor_specification=None, cost_function_specification=None,
                  controller_logging=False, **kwargs):
        # the reference's default is the checkout's config_controllers.yml `mpc: optimizer:` (shipped: rpgd); without a checkout
        # this package's is the north-star path
        if optimizer_name is None and self._yaml is not None:
            optimizer_name = self._yaml["controllers"]["mpc"].get("optimizer")
        optimizer_name = optimizer_name or "mppi"
        others = {"cem": optimizer_cem, "cem-tf": optimizer_cem, "cem-gmm": optimizer_cem_gmm, "cem-gmm-tf": optimizer_cem_gmm,
                  "gradient": optimizer_gradient,
                  "gradient-tf": optimizer_gradient, "rpgd": optimizer_rpgd, "rpgd-tf": optimizer_rpgd,
                  "cem-naive-grad": optimizer_cem_naive_grad, "cem-naive-grad-tf": optimizer_cem_naive_grad,
                  "cem-grad-bharadhwaj": optimizer_cem_grad_bharadhwaj, "cem-grad-bharadhwaj-tf": optimizer_cem_grad_bharadhwaj,
                  "random-action": optimizer_random_action, "random-action-tf": optimizer_random_action}
        if optimizer_name in others:
            return self._configure_other(others[optimizer_name], predictor_specification, cost_function_specification,
                                         controller_logging, **kwargs)
        if optimizer_name != "mppi":
            raise NotImplementedError(f"optimizer {optimizer_name!r}: built are 'mppi' (the hot path) and "
                                      f"{sorted(k for k in others if not k.endswith('-tf'))}")
        cfg = dict(self.config_optimizer)
        cfg.update(kwargs)
        cost_name = cost_function_specification or cfg.pop("cost_function_specification", None) or \
            "quadratic_boundary_grad_minimal"
        self.controller_logging = controller_logging
        opt_probe = MPPIConfig(**{k: v for k, v in cfg.items() if k in MPPIConfig.__dataclass_fields__})
        self.cost_function_wrapper = CostFunctionWrapper()
        self.cost_function_wrapper.configure(batch_size=opt_probe.num_rollouts, horizon=opt_probe.mpc_horizon,
                                             variable_parameters=self.variable_parameters,
                                             environment_name=self.environment_name,
                                             cost_function_specification=cost_name,
                                             weights=cfg.get("cost_weights"), phys=self.phys, device=self.device)
        spec = predictor_specification or cfg.pop("predictor_specification", None)
        # a checkout's config_controllers.yml:3 read by config_root names the ODE predictor - but only as the DEFAULT: a
        # gru_model handed over explicitly (config={"gru_model": ...}) wins over the YAML's predictor, as it did before
        # config_root learnt to read that line
        if spec is None and cfg.get("predictor_type") == "ODE" and cfg.get("gru_model") is None:
            spec = "ODE"
        neural = spec is not None and str(spec).startswith("GRU-6IN-32H1-32H2-5OUT")
        if cfg.get("gru_model") is not None and spec is not None and not neural:
            raise ValueError(f"gru_model was given but predictor_specification={spec!r} selects the ODE predictor")
        if neural or (spec is None and cfg.get("gru_model") is not None):
            self.predictor = None            # the network runs inside the fused kernel (optimizer_mppi.gru_model); no ODE seam object
        else:
            self.predictor = PredictorWrapper(self.phys, device=self.device)
            self.predictor.configure(batch_size=opt_probe.num_rollouts, horizon=opt_probe.mpc_horizon,
                                     dt=opt_probe.mpc_timestep, predictor_specification=spec or "ODE_v0",
                                     variable_parameters=self.variable_parameters)
        self.optimizer = optimizer_mppi(predictor=self.predictor, cost_function=self.cost_function_wrapper.cost_function,
                                        control_limits=self.control_limits, optimizer_logging=controller_logging,
                                        phys=self.phys, device=self.device, num_envs=self.num_envs,
                                        variable_parameters=self.variable_parameters, **cfg)
        self.optimizer.configure(dt=opt_probe.mpc_timestep, predictor_specification=spec)

    def _configure_other(self, cls, predictor_specification, cost_function_specification, controller_logging, **kwargs):
        if self._yaml is not None:
            # with a checkout: THIS optimizer's section of config_optimizers.yml (the class's constructor keywords are its keys),
            # the controller-level keys of config_controllers.yml `mpc:`, then the caller's overrides
            sections = self._yaml["optimizers"]
            name = cls.optimizer_name
            cfg = dict(sections.get(name) or sections.get(name + "-tf") or {})
            ctrl = self._yaml["controllers"]["mpc"]
            cost_name = ctrl.get("cost_function_specification") or self._yaml["cost"]["cost_function_name_default"]
            cfg.update(cost_function_specification=cost_name, cost_weights=dict(self._yaml["cost"]["CartPole"].get(cost_name, {})),
                       predictor_type=self.config_optimizer.get("predictor_type", "ODE_v0"),
                       intermediate_steps=self.config_optimizer.get("intermediate_steps", 10))
            cfg.update(self._user_config)
        else:
            cfg = dict(self.config_optimizer)
        cfg.update(kwargs)
        cost_name = cost_function_specification or cfg.pop("cost_function_specification", None) or \
            "quadratic_boundary_grad_minimal"
        self.controller_logging = controller_logging
        spec = predictor_specification or cfg.pop("predictor_specification", None)
        if spec is None and cfg.get("predictor_type") == "ODE":      # (a checkout's config_controllers.yml:3 read by config_root)
            spec = "ODE"
        self.cost_function_wrapper = CostFunctionWrapper()
        self.cost_function_wrapper.configure(variable_parameters=self.variable_parameters,
                                             environment_name=self.environment_name,
                                             cost_function_specification=cost_name, weights=cfg.get("cost_weights"),
                                             phys=self.phys, device=self.device)
        self.optimizer = cls(cost_function=self.cost_function_wrapper.cost_function,
                                       control_limits=self.control_limits, optimizer_logging=controller_logging,
                                       phys=self.phys, device=self.device, num_envs=self.num_envs,
                                       variable_parameters=self.variable_parameters, **cfg)
        self.optimizer.configure(predictor_specification=spec)

    def step(self, s, time=None, updated_attributes=None):
        self.update_attributes(updated_attributes)
        u = self.optimizer.step(s, time)          # (host arrays and device tensors alike: the optimizer converts what it needs)
        if self.controller_logging:
            self.controller_data_for_csv = dict(self.optimizer.logging_values)
        return u

    def controller_reset(self):
        if self.optimizer is not None:
            self.optimizer.optimizer_reset()
