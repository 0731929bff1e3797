"""Predictor seam — stand-ins with the reference's call signatures, backed by cpmppi_predict (HIP).

Mirrors (paths in the reference checkout):
  * next_state_predictor_ODE_v0   SI_Toolkit_ASF/ToolkitCustomization/predictors_customization_v0.py:22-55
  * predictor_ODE_v0 / PredictorWrapper (absent SI_Toolkit submodule) as their in-tree callers use them:
    Control_Toolkit_ASF/Controllers/controller_mppi_cartpole.py:51-52,63-65,191,473,566-567 and
    SI_Toolkit_ASF/ToolkitCustomization/Modules/ODE_module.py:29-31,46-50.
Inputs may be numpy arrays or ROCm tensors; outputs are numpy by default (what the reference returns) or device
tensors with ``as_tensor=True``.  There is no CPU path: construction fails without the HIP library / an MI355X.
"""
from types import SimpleNamespace

import numpy as np

from .configs import MPPIConfig, PhysicalParameters
from .state_utilities import STATE_INDICES, STATE_VARIABLES, CONTROL_INPUTS, create_cartpole_state  # noqa: F401


def _engine(horizon, dt, intermediate_steps, phys, math_mode, device, predictor_type="ODE_v0"):
    from .engine import MPPIEngine
    cfg = MPPIConfig(num_rollouts=1, mpc_horizon=max(1, int(horizon)), mpc_timestep=float(dt),
                     intermediate_steps=int(intermediate_steps), math_mode=math_mode, predictor_type=predictor_type)
    return MPPIEngine(1, cfg, phys, device=device)


def _apply_pole_mass(eng, variable_parameters):
    """predictor_ODE reads variable_parameters.m_pole at every call (predictors_customization.py:55-58)."""
    m = getattr(variable_parameters, "m_pole", None) if variable_parameters is not None else None
    if m is not None:                      # (by value: the attribute may be mutated in place between calls)
        eng.set_pole_mass(float(np.asarray(m.cpu() if hasattr(m, "cpu") else m, dtype=np.float32).reshape(-1)[0]))


def _pole_length(variable_parameters, phys):
    if variable_parameters is not None and hasattr(variable_parameters, "L"):
        return float(np.asarray(variable_parameters.L).reshape(-1)[0])
    return phys.L


class next_state_predictor_ODE_v0:
    """Per-step hook: ``step(s[B,6], Q[B,1]) -> s_next[B,6]`` (one control step = ``intermediate_steps`` Euler
    substeps with edge bounce and angle wrap).  Honours ``variable_parameters.L`` only, like the reference (:47-50)."""
    predictor_type = "ODE_v0"

    def __init__(self, dt, intermediate_steps, batch_size=1, variable_parameters=None, phys=None, math_mode="precise",
                 device=0, **kwargs):
        self.phys = phys or PhysicalParameters()
        self.params = self.phys
        self.variable_parameters = variable_parameters
        self.intermediate_steps = int(intermediate_steps)
        self.t_step = float(dt / float(self.intermediate_steps))
        self.s = create_cartpole_state()
        self._eng = _engine(1, dt, intermediate_steps, self.phys, math_mode, device, self.predictor_type)

    def step(self, s, Q, as_tensor=False):
        assert Q.shape[0] == s.shape[0]
        assert Q.ndim == 2
        assert s.ndim == 2
        L = _pole_length(self.variable_parameters, self.phys)
        if self.predictor_type == "ODE":
            _apply_pole_mass(self._eng, self.variable_parameters)
        out = self._eng.predict(s, Q[:, :1], L=L)[:, 1]
        return out if as_tensor else out.cpu().numpy()


class next_state_predictor_ODE(next_state_predictor_ODE_v0):
    """``SI_Toolkit_ASF/ToolkitCustomization/predictors_customization.py:25-69``: the per-step hook of predictor_type
    "ODE" - ``intermediate_steps`` Euler-Cromer substeps (cartpole_equations.py:293-304), no edge bounce, angle =
    atan2(sin, cos).  ``variable_parameters.L`` and ``variable_parameters.m_pole`` (:45-58) are read at every call, as
    in the reference."""
    predictor_type = "ODE"

    def __init__(self, dt, intermediate_steps, lib=None, batch_size=1, variable_parameters=None,
                 disable_individual_compilation=False, **kwargs):
        self.lib = lib
        super().__init__(dt, intermediate_steps, batch_size, variable_parameters, **kwargs)


class predictor_ODE_v0:
    """``predict / predict_core(s0[B,6] | [6], Q[B,H,1] | [H,1]) -> [B,H+1,6]`` with ``out[:,0] = s0``."""
    predictor_type = "ODE_v0"

    def __init__(self, horizon, dt, intermediate_steps=10, batch_size=1, variable_parameters=None, phys=None,
                 math_mode="precise", device=0, **kwargs):
        self.horizon = int(horizon)
        self.dt = float(dt)
        self.intermediate_steps = int(intermediate_steps)
        self.batch_size = int(batch_size)
        self.variable_parameters = variable_parameters
        self.phys = phys or PhysicalParameters()
        self.params = self.phys
        self._math_mode, self._device = math_mode, device
        self._eng = _engine(self.horizon, dt, intermediate_steps, self.phys, math_mode, device, self.predictor_type)
        self.next_step_predictor = SimpleNamespace(params=self.phys)

    def predict_core(self, initial_state, Q, as_tensor=False):
        eng = self._eng
        Q = eng.tensor(Q)
        if Q.dim() == 2:                       # [H,1] -> one rollout
            Q = Q.unsqueeze(0)
        if Q.dim() == 3:
            Q = Q[:, :, 0]
        s0 = eng.tensor(initial_state)
        if s0.dim() == 2 and s0.shape[0] == 1 and Q.shape[0] != 1:
            s0 = s0[0]
        if self.predictor_type == "ODE":
            _apply_pole_mass(eng, self.variable_parameters)
        out = eng.predict(s0, Q.contiguous(), L=_pole_length(self.variable_parameters, self.phys))
        return out if as_tensor else out.cpu().numpy()

    predict = predict_core

    def update(self, Q0=None, s=None):
        """No internal state for an ODE predictor (controller_mppi_cartpole.py:566-567 calls it regardless)."""
        return None


class predictor_ODE(predictor_ODE_v0):
    """predictor_type "ODE" (SI_Toolkit_ASF/config_predictors.yml:22-26; config_controllers.yml:3,14): the same seam on
    next_state_predictor_ODE's integrator (Euler-Cromer, no edge bounce, atan2 angle)."""
    predictor_type = "ODE"


class PredictorWrapper:
    """configure / predict / predict_core / update with the attributes the in-tree callers read."""

    def __init__(self, phys=None, math_mode="precise", device=0):
        self.predictor = None
        self.predictor_config = {"predictor_type": "ODE_v0", "model_name": None, "intermediate_steps": 10}
        self.predictor_type = "ODE_v0"
        self.model_name = None
        self.batch_size = None
        self._horizon = None
        self.dt = None
        self.variable_parameters = None
        self.phys, self._math_mode, self._device = phys, math_mode, device

    # the reference lets the GUI change the horizon on a live predictor (controller_mppi_cartpole.py:473)
    @property
    def horizon(self):
        return self._horizon

    @horizon.setter
    def horizon(self, value):
        if value is not None and self._horizon is not None and int(value) != self._horizon and self.predictor is not None:
            self._horizon = int(value)
            self._build()
        else:
            self._horizon = None if value is None else int(value)

    def update_predictor_config_from_specification(self, predictor_specification=None, **kwargs):
        spec = predictor_specification or "ODE_v0"
        name = str(spec).split(":")[0]
        if name in ("ODE", "ODE_default"):
            # predictors_customization.py:25-69: Euler-Cromer, atan2 angle, no edge bounce - a different integrator from
            # ODE_v0 (1.6e-3 apart after one control step), served by the kernels' predictor_ODE form
            ptype = "ODE"
        elif name in ("ODE_v0", "ODE_v0_default"):
            ptype = "ODE_v0"
        else:
            raise NotImplementedError(f"predictor_specification {spec!r}: the predictor seam serves ODE_v0 and ODE; the GRU "
                                      "predictor runs inside the fused kernel (optimizer_mppi(gru_model=...))")
        self.predictor_config = {"predictor_type": ptype, "model_name": None,
                                 "intermediate_steps": self.predictor_config["intermediate_steps"]}
        self.predictor_type = ptype

    def _build(self):
        cls = predictor_ODE if self.predictor_type == "ODE" else predictor_ODE_v0
        self.predictor = cls(self._horizon, self.dt, self.predictor_config["intermediate_steps"], self.batch_size,
                             self.variable_parameters, phys=self.phys, math_mode=self._math_mode, device=self._device)

    def configure(self, batch_size, horizon, dt, predictor_specification=None, variable_parameters=None, **kwargs):
        self.update_predictor_config_from_specification(predictor_specification)
        self.configure_with_compilation(batch_size, horizon, dt, variable_parameters=variable_parameters)

    def configure_with_compilation(self, batch_size, horizon, dt, variable_parameters=None, **kwargs):
        self.batch_size, self._horizon, self.dt = int(batch_size), int(horizon), float(dt)
        self.variable_parameters = variable_parameters
        self._build()

    def predict(self, s, Q, **kw):
        return self.predictor.predict(s, Q, **kw)

    def predict_core(self, s, Q, **kw):
        return self.predictor.predict_core(s, Q, **kw)

    def update(self, Q0=None, s=None):
        return self.predictor.update(Q0, s)


class predictor_output_augmentation:
    """``SI_Toolkit_ASF/ToolkitCustomization/predictors_customization.py:72-139``: the features a neural predictor's
    output lacks and the state vector needs.  All three legs of the reference:
      * outputs hold ``angle_sin`` and ``angle_cos`` but no ``angle``  -> append ``angle = atan2(angle_sin, angle_cos)``
      * outputs hold ``angle`` but no ``angle_sin``                    -> append ``angle_sin = sin(angle)``
      * outputs hold ``angle`` but no ``angle_cos``                    -> append ``angle_cos = cos(angle)``
    in that order (:88-96, :121-137).  ``net_info`` needs ``.outputs`` (list of names; a differential network's ``D_*``
    names are stripped as in :79-83).  ``augment`` takes ``[batch, time, features]`` as a torch tensor (any device) or a
    numpy array and returns the same kind with the new features concatenated last.  (The fused GRU kernel applies the
    first leg in registers; this class is the seam for networks evaluated outside it.)"""

    def __init__(self, net_info, lib=None, disable_individual_compilation=False, differential_network=False):
        from .state_utilities import ANGLE_COS_IDX, ANGLE_IDX, ANGLE_SIN_IDX
        self.lib = lib
        self.differential_network = differential_network
        outputs = [x[2:] for x in net_info.outputs] if differential_network else list(net_info.outputs)
        self.net_output_indices = {key: value for value, key in enumerate(outputs)}
        self.indices_augmentation, self.features_augmentation = [], []
        if "angle" not in outputs and "angle_sin" in outputs and "angle_cos" in outputs:
            self.indices_augmentation.append(ANGLE_IDX)
            self.features_augmentation.append("angle")
        if "angle_sin" not in outputs and "angle" in outputs:
            self.indices_augmentation.append(ANGLE_SIN_IDX)
            self.features_augmentation.append("angle_sin")
        if "angle_cos" not in outputs and "angle" in outputs:
            self.indices_augmentation.append(ANGLE_COS_IDX)
            self.features_augmentation.append("angle_cos")
        self.augmentation_len = len(self.indices_augmentation)
        self.index_angle = self.net_output_indices.get("angle")
        self.index_angle_sin = self.net_output_indices.get("angle_sin")
        self.index_angle_cos = self.net_output_indices.get("angle_cos")
        self.augment = self._augment

    def get_indices_augmentation(self):
        return self.indices_augmentation

    def get_features_augmentation(self):
        return self.features_augmentation

    def _augment(self, net_output):
        import torch
        is_t = torch.is_tensor(net_output)
        xp_atan2 = torch.atan2 if is_t else np.arctan2
        xp_sin, xp_cos = (torch.sin, torch.cos) if is_t else (np.sin, np.cos)
        cat = (lambda xs: torch.cat(xs, dim=-1)) if is_t else (lambda xs: np.concatenate(xs, axis=-1))
        output = net_output
        if "angle" in self.features_augmentation:
            angle = xp_atan2(net_output[..., self.index_angle_sin], net_output[..., self.index_angle_cos])[:, :, None]
            output = cat([output, angle])
        if "angle_sin" in self.features_augmentation:
            output = cat([output, xp_sin(net_output[..., self.index_angle])[:, :, None]])
        if "angle_cos" in self.features_augmentation:
            output = cat([output, xp_cos(net_output[..., self.index_angle])[:, :, None]])
        return output
