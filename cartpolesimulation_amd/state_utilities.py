"""State layout of the cartpole (mirror of the reference's CartPole/state_utilities.py:5-53 interface)."""
import numpy as np

STATE_VARIABLES = np.array(sorted(["angle", "angleD", "angle_cos", "angle_sin", "position", "positionD"]))
STATE_INDICES = {x: int(np.where(STATE_VARIABLES == x)[0][0]) for x in STATE_VARIABLES}
CONTROL_INPUTS = np.array(["Q"])
CONTROL_INDICES = {"Q": 0}

ANGLE_IDX = STATE_INDICES["angle"]
ANGLED_IDX = STATE_INDICES["angleD"]
POSITION_IDX = STATE_INDICES["position"]
POSITIOND_IDX = STATE_INDICES["positionD"]
ANGLE_COS_IDX = STATE_INDICES["angle_cos"]
ANGLE_SIN_IDX = STATE_INDICES["angle_sin"]


def create_cartpole_state(state=None, dtype=np.float32):
    """float32[6] in STATE_VARIABLES order; unset entries are 0; angle_cos/angle_sin are filled from ``angle``."""
    state = dict(state or {})
    angle = state.get("angle", 0.0)
    state["angle_cos"], state["angle_sin"] = np.cos(angle), np.sin(angle)
    s = np.zeros(len(STATE_VARIABLES), dtype=np.float32)
    for i, v in enumerate(STATE_VARIABLES):
        if v in state:
            s[i] = state[v]
    return s.astype(dtype, copy=False)
