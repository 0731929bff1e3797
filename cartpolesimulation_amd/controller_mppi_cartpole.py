"""controller_mppi_cartpole — the reference's IN-TREE MPPI controller on the fused HIP path.

Mirror of ``Control_Toolkit_ASF/Controllers/controller_mppi_cartpole.py:337-580`` (class ``controller_mppi_cartpole``,
the only MPPI implementation whose source is in the reference tree; it is what BASELINE config C1 runs): same
constructor (``template_controller``), ``configure()``, ``step(s, time, updated_attributes)``, ``controller_reset()``,
``update_control_vector()``, ``initialize_perturbations(stdev, sampling_type)`` and the attributes the simulator and
the GUI read (``u``, ``u_prev``, ``delta_u``, ``S_tilde_k``, ``rng_mppi``, ``iteration``).  What the reference computes
in ``trajectory_rollouts`` + ``update_inputs`` (:164-224, :324-335) — predict 10-substep Euler rollouts, the legacy cost
``q`` + ``phi`` (:227-303), the soft-min update (:306-321) — is ONE launch of ``cpmppi_step`` here.

RNG discipline is the reference's (numpy ``Generator(SFC64(seed))``): five uniforms for the cost-weight noise in
``configure`` (:355-359), the perturbations of every optimisation step (:479-483, every ``SAMPLING_TYPE``), one uniform
for the multiplicative output noise ``Q (1 + p_Q U(-1, 1))`` (:553) — so for a given seed the controller reproduces the
reference's own traces (tests/golden/legacy_step_*.npz, closed_loop_c1.npz) to the float32 tolerance of the kernels.

Configuration: the keys of ``config_controllers.yml:9-30`` (section ``mppi-cartpole``) as ``config=dict(...)``, or
``config_root=<CartPoleSimulation checkout>`` to read them (and ``dt.control`` of ``config_data_gen.yml:27``,
``actuator_noise`` of ``cartpole_physical_parameters.yml:2``) from the YAML files the reference reads at import (:38-41).
"""
import time as _time

import numpy as np
import torch

from .configs import PhysicalParameters, legacy_mppi_config
from .controller_mpc import template_controller
from .sampling import SAMPLING_TYPES, sample_delta_u_sfc64, sample_knots_sfc64

# config_controllers.yml:9-30
DEFAULTS = dict(seed=None, mpc_horizon=35, num_rollouts=3500, update_every=1, predictor_specification="ODE_v0",
                dd_weight=120.0, ep_weight=50000.0, ekp_weight=0.01, ekc_weight=5.0, cc_weight=1.0, ccrc_weight=1.0,
                cost_noise=0.0, R=1.0, LBD=100.0, NU=1000.0, SQRTRHOINV=0.02, SAMPLING_TYPE="interpolated",
                controller_logging=False, WASH_OUT_LEN=100)


class controller_mppi_cartpole(template_controller):
    def __init__(self, environment_name="CartPole", initial_environment_attributes=None, control_limits=None,
                 config=None, config_root=None, dt=0.02, actuator_noise=0.1, phys=None, device=0, math_mode="fast", **kwargs):
        super().__init__(environment_name, initial_environment_attributes, control_limits)
        cfg = dict(DEFAULTS)
        if config_root is not None:
            from .configs import load_reference_yaml
            import os
            import yaml
            yaml_phys, cfgs = load_reference_yaml(config_root)
            cfg.update(cfgs["controllers"]["mppi-cartpole"])
            dt = cfgs["data_gen"]["dt"]["control"]
            with open(os.path.join(config_root, "cartpole_physical_parameters.yml")) as fh:
                actuator_noise = yaml.safe_load(fh)["cartpole"]["actuator_noise"]
            phys = phys or yaml_phys
        cfg.update(config or {})
        unknown = set(cfg) - set(DEFAULTS) - {"control_noise", "cost_function_specification"}
        if unknown:
            raise ValueError(f"unknown mppi-cartpole keys: {sorted(unknown)}")
        if cfg["SAMPLING_TYPE"] not in SAMPLING_TYPES:
            raise ValueError(f"SAMPLING_TYPE must be one of {SAMPLING_TYPES}")
        spec = cfg["predictor_specification"]
        if spec in ("ODE", "ODE_default"):
            # the shipped YAML (config_controllers.yml:14) says "ODE": next_state_predictor_ODE (Euler-Cromer, no bounce)
            self.predictor_type = "ODE"
        elif spec in ("ODE_v0", "ODE_v0_default"):
            self.predictor_type = "ODE_v0"
        else:
            raise NotImplementedError(f"predictor_specification {spec!r}: this controller runs on the ODE_v0 and ODE kernels")
        self.config = cfg
        self.dt, self.p_Q = float(dt), float(actuator_noise)
        self.phys = phys or PhysicalParameters()
        self.device, self.math_mode = device, math_mode
        self.mpc_horizon, self.num_rollouts = int(cfg["mpc_horizon"]), int(cfg["num_rollouts"])
        self.update_every = int(cfg["update_every"])
        self.SQRTRHODTINV = np.float64(cfg["SQRTRHOINV"]) * (1 / np.sqrt(self.dt))         # :91
        self.LOGS = {"cost_to_go": [], "states": [], "trajectory": [], "target_trajectory": [], "inputs": [],
                     "nominal_rollouts": []}
        self.engine = None
        self._staging = {}

    # ------------------------------------------------------------------ :345-390
    def configure(self):
        from .engine import MPPIEngine
        cfg = self.config
        seed = cfg["seed"]
        if seed is None:
            seed = int(_time.time() * 1000.0)                                               # :348-349 (fully random)
        self.rng_mppi = np.random.Generator(np.random.SFC64(int(seed)))
        self.rng_mppi_rnn = np.random.Generator(np.random.SFC64(int(seed) * 2))
        w = {}
        for k in ("dd_weight", "ep_weight", "ekp_weight", "ekc_weight", "cc_weight"):       # :355-359, in this order
            w[k] = cfg[k] * (1 + cfg["cost_noise"] * self.rng_mppi.uniform(-1.0, 1.0))
        w["ccrc_weight"] = cfg["ccrc_weight"]
        self.cost_weights = w
        self.iteration = -1
        self.control_enabled = True
        self._build_engine()
        self.delta_u = np.zeros((self.num_rollouts, self.mpc_horizon), dtype=np.float32)

    def _build_engine(self):
        from .engine import MPPIEngine
        cfg, w = self.config, self.cost_weights
        mcfg = legacy_mppi_config(num_rollouts=self.num_rollouts, mpc_horizon=self.mpc_horizon, mpc_timestep=self.dt,
                                  R=cfg["R"], LBD=cfg["LBD"], NU=cfg["NU"], SQRTRHOINV=cfg["SQRTRHOINV"],
                                  shift_mode="none",           # the shift is this class's own last step, as in the reference
                                  math_mode=self.math_mode, predictor_type=self.predictor_type,
                                  cost_weights=dict(dd_weight=w["dd_weight"], ep_weight=w["ep_weight"], ekp_weight=w["ekp_weight"],
                                                    ekc_weight=w["ekc_weight"], cc_weight=w["cc_weight"], ccrc_weight=w["ccrc_weight"]))
        old_u = getattr(self, "_u", None)
        self.engine = MPPIEngine(1, mcfg, self.phys, device=self.device)
        self._u = self.engine.zeros(1, self.mpc_horizon)
        self._u_prev = self.engine.zeros(1, self.mpc_horizon)
        if old_u is not None:                                                               # update_control_vector (:574-583)
            n = min(self.mpc_horizon, old_u.shape[1])
            self._u[:, :n] = old_u[:, :n]
            self._u_prev.copy_(self._u)
        self._S = self.engine.empty(1, self.num_rollouts)
        self._Q = self.engine.empty(1)

    # the reference's attributes, as host arrays
    @property
    def u(self):
        return self._u[0].cpu().numpy()

    @property
    def u_prev(self):
        return self._u_prev[0].cpu().numpy()

    @property
    def S_tilde_k(self):
        return self._S[0].cpu().numpy()

    # ------------------------------------------------------------------ :392-450
    def initialize_perturbations(self, stdev=1.0, sampling_type=None):
        """-> delta_u [num_rollouts, mpc_horizon] float32 from ``self.rng_mppi``, every mode of the reference."""
        N, H = self.num_rollouts, self.mpc_horizon
        if sampling_type == "interpolated":
            kn = self._knots(stdev)
            return self.engine.interpolate(kn).cpu().numpy()[0]
        if sampling_type in ("random_walk", "uniform", "repeated"):
            return sample_delta_u_sfc64(self.rng_mppi, 1, N, H, stdev, sampling_type)[0]
        return sample_delta_u_sfc64(self.rng_mppi, 1, N, H, stdev, "iid")[0]                # :447-450: anything else is iid

    def _knots(self, stdev):
        P = self.engine.P
        z = self.rng_mppi.standard_normal(size=(self.num_rollouts, P), dtype=np.float32)
        return (np.float64(stdev) * z.astype(np.float64)).astype(np.float32)[None]

    # ------------------------------------------------------------------ :454-569
    def step(self, s, time=None, updated_attributes=None):
        if self.engine is None:
            self.configure()
        self.update_attributes(updated_attributes)
        self.s = np.asarray(s, dtype=np.float32)
        self.iteration += 1
        if self.mpc_horizon != self._u.shape[1]:                                            # :472-475 (horizon changed in the GUI)
            self.update_control_vector()
        vp = self.variable_parameters
        target = float(np.asarray(getattr(vp, "target_position", 0.0)).reshape(-1)[0])
        if self.iteration % self.update_every == 0:
            L = getattr(vp, "L", None)
            # host -> device through pinned staging buffers and asynchronous copies (state / target / pole length as one
            # block, the perturbations as another): pageable uploads cost ~80 us of a 300 us control step
            small = self._staged("small", (9,))
            small[0][:6], small[0][6], small[0][7] = self.s, target, 1.0
            small[0][8] = self.phys.L if L is None else float(np.asarray(L, np.float32).reshape(-1)[0])
            d = self._upload("small")
            kw = dict(L=d[8:9])
            if self.config["SAMPLING_TYPE"] == "interpolated":
                kn = self._staged("knots", (1, self.num_rollouts, self.engine.P))
                kn[0][0] = self._knots(self.SQRTRHODTINV)[0]                                # interpolated on the device
                kw["knots"] = self._upload("knots")
                self.delta_u = None
            else:
                self.delta_u = self.initialize_perturbations(self.SQRTRHODTINV, self.config["SAMPLING_TYPE"])
                du = self._staged("delta_u", (1, self.num_rollouts, self.mpc_horizon))
                du[0][0] = self.delta_u
                kw["delta_u"] = self._upload("delta_u")
            self.engine.step(d[:6].view(1, 6), self._u, d[6:7], d[7:8], u_prev=self._u_prev, S_out=self._S, Q_out=self._Q, **kw)
            if self.config["controller_logging"]:
                self.LOGS["cost_to_go"].append(self.S_tilde_k.copy())
                self.LOGS["inputs"].append(self.u.copy())
                traj = self.engine.predict(self.s[None], self._u, L=kw["L"]).cpu().numpy()[0]
                self.LOGS["nominal_rollouts"].append(traj[:-1])
        if self.config["controller_logging"]:
            self.LOGS["trajectory"].append(self.s.copy())
            self.LOGS["target_trajectory"].append(np.float32(target))
        Q = self._u[0, 0].item()                                                             # :536
        Q = np.float32(Q * (1 + self.p_Q * self.rng_mppi.uniform(-1.0, 1.0)))               # :553
        Q = np.clip(Q, np.float32(-1.0), np.float32(1.0))                                    # :555
        self._u_prev.copy_(self._u)                                                          # :558
        self._u[:, :-1] = self._u[:, 1:].clone()                                             # :561-562
        self._u[:, -1] = 0.0
        return Q

    def _staged(self, name, shape):
        """-> (numpy view of a pinned host buffer, the buffer, its device twin) for `name`, (re)allocated on a shape change."""
        ent = self._staging.get(name)
        if ent is None or tuple(ent[1].shape) != tuple(shape):
            host = torch.empty(*shape, dtype=torch.float32).pin_memory()
            ent = (host.numpy(), host, torch.empty(*shape, dtype=torch.float32, device=self._u.device), torch.cuda.Event())
            self._staging[name] = ent
        else:
            ent[3].synchronize()                              # the previous upload has left the pinned buffer
        return ent

    def _upload(self, name):
        _, host, dev, ev = self._staging[name]
        dev.copy_(host, non_blocking=True)
        ev.record()
        return dev

    def update_control_vector(self):
        """:574-583 — the horizon was changed: keep the leading part of the best-guess sequence, zero the rest."""
        self._build_engine()

    def controller_reset(self):
        self.configure()

    def controller_report(self):
        return self.LOGS if self.config["controller_logging"] else None
