"""Configuration of the hot path: same keys and default values as the reference's YAML files.

  physics      cartpole_physical_parameters.yml:6-17,34,42   (rounded to float32 as CartPole/cartpole_parameters.py:27-31)
  optimizer    Control_Toolkit_ASF/config_optimizers.yml:87-97  (section ``mppi``)
  controller   Control_Toolkit_ASF/config_controllers.yml:1-8   (section ``mpc``), :9-30 (``mppi-cartpole``, legacy)
  cost         Control_Toolkit_ASF/config_cost_function.yml:6-11,37-45
  predictor    SI_Toolkit_ASF/config_predictors.yml:18-21       (``ODE_v0_default``: intermediate_steps 10)
  timing       config_data_gen.yml:25-28                        (dt control 0.02, simulation 0.002)

``load_reference_yaml(root)`` reads those files from a CartPoleSimulation checkout when one is available, so an
existing project keeps its tuned values; nothing here needs the checkout at run time.
"""
import os
from dataclasses import dataclass, field, asdict

import numpy as np
import yaml

from . import _lib as L

f32 = np.float32


@dataclass
class PhysicalParameters:
    k: float = float(f32(1.0 / 3.0))
    m_cart: float = float(f32(0.230))
    m_pole: float = float(f32(0.087))
    g: float = float(f32(9.81))
    J_fric: float = float(f32(5.0e-5))
    M_fric: float = float(f32(3.22))
    L: float = float(f32(0.395))
    u_max: float = float(f32(1.77))
    TrackHalfLength: float = float(f32((44.0e-2 - 4.4e-2) / 2.0))
    v_max: float = float(f32(0.8))


COST_WEIGHTS = {
    # name -> (cost_id, ordered key list, defaults) ; order = cost_w layout documented in include/cpmppi.h
    "quadratic_boundary_grad_minimal": (L.COST_QBGM,
                                        ["dd_quadratic_weight_up", "db_weight_up", "ep_weight_up", "ekp_weight_up",
                                         "cc_weight_up", "R", "permissible_track_fraction"],
                                        dict(dd_quadratic_weight_up=10.0, db_weight_up=10000.0, ep_weight_up=40.0,
                                             ekp_weight_up=1.0, cc_weight_up=5.0, R=1.0,
                                             permissible_track_fraction=0.85)),
    "default": (L.COST_DEFAULT, ["dd_weight", "ep_weight", "cc_weight", "R"],
                dict(dd_weight=600.0, ep_weight=20000.0, cc_weight=1.0, R=1.0, ccrc_weight=1.0)),
    # config_cost_function.yml:12-36; "cos_admissible_angle" is derived from admissible_angle (degrees in the YAML)
    "quadratic_boundary_grad": (L.COST_QBG,
                                ["dd_quadratic_weight_up", "dd_linear_weight_up", "db_weight_up", "ep_weight_up",
                                 "ekp_weight_up", "cc_weight_up", "ccrc_weight_up",
                                 "dd_quadratic_weight_down", "dd_linear_weight_down", "db_weight_down", "ep_weight_down",
                                 "ekp_weight_down", "cc_weight_down", "ccrc_weight_down",
                                 "target_angular_speed_sqr_max_correction_up",
                                 "target_angular_speed_sqr_max_correction_down", "permissible_track_fraction",
                                 "cos_admissible_angle", "R"],
                                dict(dd_quadratic_weight_up=500.0, dd_linear_weight_up=0.0, ep_weight_up=6000.0,
                                     target_angular_speed_sqr_max_correction_up=0.0, ekp_weight_up=30.0,
                                     db_weight_up=10000.0, cc_weight_up=5.0, ccrc_weight_up=0.0,
                                     dd_quadratic_weight_down=500.0, dd_linear_weight_down=0.0, ep_weight_down=6000.0,
                                     target_angular_speed_sqr_max_correction_down=100.0, ekp_weight_down=30.0,
                                     db_weight_down=10000.0, cc_weight_down=5.0, ccrc_weight_down=0.0,
                                     permissible_track_fraction=0.85, admissible_angle=0.0, R=1.0)),
    # config_cost_function.yml:53-58 / :47-52 (the nonconvex module reads `cem_ccrc_weight`, which the shipped YAML lacks: here
    # the key is ccrc_weight for both)
    "quadratic_boundary": (L.COST_QB, ["dd_weight", "ep_weight", "cc_weight", "R", "ccrc_weight"],
                           dict(dd_weight=600.0, ep_weight=20000.0, cc_weight=1.0, R=1.0, ccrc_weight=1.0)),
    "quadratic_boundary_nonconvex": (L.COST_QB_NONCONVEX, ["dd_weight", "ep_weight", "cc_weight", "R", "ccrc_weight"],
                                     dict(dd_weight=600.0, ep_weight=20000.0, cc_weight=1.0, R=1.0, ccrc_weight=1.0)),
    "legacy_mppi_cartpole": (L.COST_LEGACY,
                             ["dd_weight", "ep_weight", "ekp_weight", "ekc_weight", "cc_weight", "ccrc_weight"],
                             dict(dd_weight=120.0, ep_weight=50000.0, ekp_weight=0.01, ekc_weight=5.0, cc_weight=1.0,
                                  ccrc_weight=1.0)),
}


@dataclass
class MPPIConfig:
    """config_optimizers.yml:87-97 plus the glue choices the absent Control_Toolkit leaves unpinned."""
    seed: int = None
    mpc_horizon: int = 35
    mpc_timestep: float = 0.02
    num_rollouts: int = 3500
    cc_weight: float = 1.0
    R: float = 1.0
    LBD: float = 100.0
    NU: float = 1000.0
    SQRTRHOINV: float = 0.03
    period_interpolation_inducing_points: int = 10
    intermediate_steps: int = 10
    cost_function_specification: str = "quadratic_boundary_grad_minimal"
    cost_weights: dict = field(default_factory=dict)       # overrides of COST_WEIGHTS defaults
    horizon_reduce: str = "sum"          # "sum" | "mean"
    control_mode: str = "clip"           # "clip" | "penalise"
    shift_mode: str = "repeat_last"      # "repeat_last" | "append_zero" | "none"
    correction_u: str = "u_run"          # "u_run" | "u_nom"
    math_mode: str = "fast"              # "fast" | "precise"
    rollouts_per_lane: int = 0           # 0 auto | 1 (latency mapping) | 2 (packed float2 throughput mapping)
    predictor_type: str = "ODE_v0"       # "ODE_v0" | "ODE" (SI_Toolkit_ASF/config_predictors.yml:18-26; "ODE" = Euler-Cromer, no bounce)
    action_low: float = -1.0
    action_high: float = 1.0

    @property
    def sigma(self):
        return float(np.float64(self.SQRTRHOINV) * (1 / np.sqrt(self.mpc_timestep)))

    @property
    def num_knots(self):
        return int(np.ceil(self.mpc_horizon / self.period_interpolation_inducing_points)) + 1


_ENUMS = {
    "horizon_reduce": {"sum": L.REDUCE_SUM, "mean": L.REDUCE_MEAN},
    "control_mode": {"clip": L.CONTROL_CLIP, "penalise": L.CONTROL_PENALISE},
    "shift_mode": {"repeat_last": L.SHIFT_REPEAT_LAST, "append_zero": L.SHIFT_APPEND_ZERO, "none": L.SHIFT_NONE},
    "correction_u": {"u_run": L.CORRECTION_U_RUN, "u_nom": L.CORRECTION_U_NOM},
    "math_mode": {"precise": L.MATH_PRECISE, "fast": L.MATH_FAST},
}
ODE_PREDICTORS = {"ODE_v0": L.ODE_V0, "ODE": L.ODE_CROMER}


def cost_vector(name, overrides=None):
    if name not in COST_WEIGHTS:
        raise ValueError(f"unknown cost_function_specification {name!r}; available: {sorted(COST_WEIGHTS)}")
    cost_id, keys, defaults = COST_WEIGHTS[name]
    vals = dict(defaults)
    vals.update({k: v for k, v in (overrides or {}).items() if k in defaults or k in keys})
    if "admissible_angle" in vals and "cos_admissible_angle" not in (overrides or {}):
        rad = f32(np.float32(np.pi) * vals["admissible_angle"] / 180.0)      # quadratic_boundary_grad.py:33 (float32)
        vals["cos_admissible_angle"] = float(np.cos(rad))
    return cost_id, [float(vals[k]) for k in keys]


def legacy_mppi_config(**kw):
    """The in-tree legacy controller's behaviour (config_controllers.yml:9-30, controller_mppi_cartpole.py)."""
    base = dict(cost_function_specification="legacy_mppi_cartpole", SQRTRHOINV=0.02, control_mode="penalise",
                shift_mode="append_zero", correction_u="u_nom", horizon_reduce="sum")
    base.update(kw)
    return MPPIConfig(**base)


def build_c_config(E, mppi: MPPIConfig, phys: PhysicalParameters = None):
    phys = phys or PhysicalParameters()
    c = L.cpmppi_config()
    c.abi_version = L.ABI_VERSION
    c.E, c.N, c.H, c.S = int(E), int(mppi.num_rollouts), int(mppi.mpc_horizon), int(mppi.intermediate_steps)
    c.dt = mppi.mpc_timestep
    c.k, c.m_cart, c.m_pole, c.g = phys.k, phys.m_cart, phys.m_pole, phys.g
    c.J_fric, c.M_fric, c.u_max, c.track_half_length = phys.J_fric, phys.M_fric, phys.u_max, phys.TrackHalfLength
    c.L_default = phys.L
    cost_id, w = cost_vector(mppi.cost_function_specification, mppi.cost_weights)
    c.cost_id = cost_id
    for i, v in enumerate(w):
        c.cost_w[i] = v
    c.R, c.LBD, c.NU, c.cc_weight = mppi.R, mppi.LBD, mppi.NU, mppi.cc_weight
    c.sigma = mppi.sigma
    c.period = int(mppi.period_interpolation_inducing_points)
    c.action_low, c.action_high = mppi.action_low, mppi.action_high
    for key, table in _ENUMS.items():
        val = getattr(mppi, key)
        if val not in table:
            raise ValueError(f"{key}={val!r}; expected one of {sorted(table)}")
        setattr(c, key, table[val])
    c.rollouts_per_lane = int(mppi.rollouts_per_lane)
    if mppi.predictor_type not in ODE_PREDICTORS:
        raise ValueError(f"predictor_type={mppi.predictor_type!r}; expected one of {sorted(ODE_PREDICTORS)}")
    c.ode_predictor = ODE_PREDICTORS[mppi.predictor_type]
    return c


def load_reference_yaml(root):
    """Read a CartPoleSimulation checkout's YAML files -> (PhysicalParameters, dict of optimizer/controller/cost cfg)."""
    def rd(*parts):
        with open(os.path.join(root, *parts)) as fh:
            return yaml.safe_load(fh)
    ph = rd("cartpole_physical_parameters.yml")["cartpole"]
    k = ph["k"]
    k = float(k.split("/")[0]) / float(k.split("/")[1]) if isinstance(k, str) else float(k)
    r32 = lambda v: float(f32(v))
    phys = PhysicalParameters(k=r32(k), m_cart=r32(ph["m_cart"]), m_pole=r32(ph["m_pole"]["init_value"]),
                              g=r32(ph["g"]), J_fric=r32(ph["J_fric"]), M_fric=r32(ph["M_fric"]),
                              L=r32(ph["L"]["init_value"]), u_max=r32(ph["u_max"]), v_max=r32(ph["v_max"]),
                              TrackHalfLength=r32((ph["track_length"] - ph["cart_length"]) / 2.0))
    return phys, dict(optimizers=rd("Control_Toolkit_ASF", "config_optimizers.yml"),
                      controllers=rd("Control_Toolkit_ASF", "config_controllers.yml"),
                      cost=rd("Control_Toolkit_ASF", "config_cost_function.yml"),
                      predictors=rd("SI_Toolkit_ASF", "config_predictors.yml"),
                      data_gen=rd("config_data_gen.yml"))


def mppi_config_from_yaml(cfgs, **overrides):
    opt = dict(cfgs["optimizers"]["mppi"])
    ctrl = cfgs["controllers"]["mpc"]
    name = ctrl.get("cost_function_specification") or cfgs["cost"]["cost_function_name_default"]
    weights = dict(cfgs["cost"]["CartPole"].get(name, {}))
    kw = dict(seed=opt.get("seed"), mpc_horizon=opt["mpc_horizon"], mpc_timestep=opt["mpc_timestep"],
              num_rollouts=opt["num_rollouts"], cc_weight=opt["cc_weight"], R=opt["R"], LBD=opt["LBD"], NU=opt["NU"],
              SQRTRHOINV=opt["SQRTRHOINV"],
              period_interpolation_inducing_points=opt["period_interpolation_inducing_points"],
              intermediate_steps=cfgs["predictors"]["predictors"]["ODE_v0_default"]["intermediate_steps"],
              cost_function_specification=name, cost_weights=weights)
    # config_controllers.yml:3 predictor_specification: a predictor_type, or the name of an entry of config_predictors.yml
    spec = str(ctrl.get("predictor_specification") or "ODE_v0").split(":")[0]
    entries = cfgs["predictors"]["predictors"]
    entry = entries.get(spec) or {}
    ptype = entry.get("predictor_type", spec)
    if ptype in ODE_PREDICTORS:                      # (a neural / GP specification is the caller's to resolve: gru_model=...)
        kw["predictor_type"] = ptype
        # the named entry's own substep count first (a custom entry such as `I_love_control_too: {predictor_type: ODE,
        # intermediate_steps: 2}`), "<type>_default" only when the entry has none
        steps = entry.get("intermediate_steps")
        if steps is None:
            steps = (entries.get(f"{ptype}_default") or {}).get("intermediate_steps", kw["intermediate_steps"])
        kw["intermediate_steps"] = steps
    kw.update(overrides)
    return MPPIConfig(**kw)


def as_dict(cfg):
    return asdict(cfg)
