"""cartpolesimulation_amd — MI355X-native (gfx950) MPPI rollout hot path for SensorsINI/CartPoleSimulation.

Hand-written HIP kernels behind a C ABI (include/cpmppi.h, libcpmppi.so) with Python stand-ins that keep the
reference's controller / optimizer / predictor / cost-function call signatures.  No CPU fallback: importing the
package loads libcpmppi.so and raises ImportError if it has not been built.
"""
from . import _lib

_lib.load()          # fail loudly, at import time, if the HIP library is missing

from .state_utilities import (STATE_VARIABLES, STATE_INDICES, CONTROL_INPUTS, ANGLE_IDX, ANGLED_IDX, ANGLE_COS_IDX,  # noqa: E402,F401
                              ANGLE_SIN_IDX, POSITION_IDX, POSITIOND_IDX, create_cartpole_state)
from .configs import MPPIConfig, PhysicalParameters, legacy_mppi_config  # noqa: E402,F401

__all__ = ["MPPIConfig", "PhysicalParameters", "legacy_mppi_config", "create_cartpole_state", "STATE_VARIABLES",
           "STATE_INDICES"]


def __getattr__(name):
    # torch-dependent layers are imported lazily so that `import cartpolesimulation_amd` stays cheap
    import importlib
    lazy = {"MPPIEngine": "engine", "optimizer_mppi": "optimizer_mppi", "controller_mpc": "controller_mpc",
            "PredictorWrapper": "predictors", "predictor_ODE_v0": "predictors", "predictor_ODE": "predictors",
            "next_state_predictor_ODE_v0": "predictors", "next_state_predictor_ODE": "predictors", "CostFunctionWrapper": "cost_functions"}
    if name in lazy:
        return getattr(importlib.import_module(f".{lazy[name]}", __name__), name)
    raise AttributeError(name)
