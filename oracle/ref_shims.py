"""Stand-in modules that let the reference's IN-TREE hot-path code import in this container.

TEST INFRASTRUCTURE ONLY.  Used by ``oracle/gen_golden.py`` (run in the build container, where
``/root/reference`` is mounted) to produce the golden vectors under ``tests/golden/``.  Nothing in the
product package imports this file and it never runs on the GPU box.

Why it exists: ``numba``, ``tensorflow``, ``ruamel.yaml``,
``engineering_notation`` and the two git submodules ``SI_Toolkit`` / ``Control_Toolkit`` are absent from the
reference mount, so the in-tree physics (``CartPole/cartpole_numba.py``, ``CartPole/cartpole_equations.py``), the
per-step predictor hook (``SI_Toolkit_ASF/ToolkitCustomization/predictors_customization_v0.py``), the cost plugins
(``Control_Toolkit_ASF/Cost_Functions/CartPole/*.py``) and the legacy MPPI controller
(``Control_Toolkit_ASF/Controllers/controller_mppi_cartpole.py``) cannot be imported as they stand.  The stand-ins
below carry NO reference code: they are the thinnest possible glue (``jit`` = identity, a numpy "computation
library" with the handful of methods the plugins call, a predictor wrapper that only loops over the *in-tree*
``next_state_predictor_ODE_v0.step``).  All arithmetic that ends up in a fixture is executed by the reference's own
code objects.
"""
import sys
import types
from types import SimpleNamespace

import numpy as np
import yaml

REFERENCE_ROOT = "/root/reference"


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


# --------------------------------------------------------------------------- numba: jit = identity
def _jit(*args, **kwargs):
    if len(args) >= 1 and callable(args[0]):
        return args[0]

    def deco(fn):
        return fn

    return deco


# --------------------------------------------------------------------------- SI_Toolkit.computation_library
class NumpyLibrary:
    """The subset of SI_Toolkit's NumpyLibrary interface the in-tree plugins call."""
    lib = "Numpy"
    float32 = np.float32
    float64 = np.float64
    int32 = np.int32
    int64 = np.int64
    bool = np.bool_
    pi = np.array(np.pi).astype(np.float32)
    newaxis = np.newaxis

    @staticmethod
    def to_tensor(x, dtype=None):
        return np.asarray(x, dtype=dtype)

    @staticmethod
    def to_variable(x, dtype=None):
        return np.array(x, dtype=dtype)

    @staticmethod
    def to_numpy(x):
        return np.asarray(x)

    @staticmethod
    def assign(v, x):
        v[...] = x

    @staticmethod
    def cast(x, dtype):
        return np.asarray(x).astype(dtype)

    abs = staticmethod(np.abs)
    cos = staticmethod(np.cos)
    sin = staticmethod(np.sin)
    exp = staticmethod(np.exp)
    sqrt = staticmethod(np.sqrt)
    sign = staticmethod(np.sign)
    atan2 = staticmethod(np.arctan2)
    zeros_like = staticmethod(np.zeros_like)
    ones_like = staticmethod(np.ones_like)
    reshape = staticmethod(np.reshape)
    stack = staticmethod(np.stack)
    tile = staticmethod(np.tile)
    clip = staticmethod(np.clip)
    where = staticmethod(np.where)
    floormod = staticmethod(np.mod)

    @staticmethod
    def sum(x, axis=None):
        return np.sum(x, axis=axis)

    @staticmethod
    def mean(x, axis=None):
        return np.mean(x, axis=axis)

    @staticmethod
    def reduce_min(x, axis=None):
        return np.min(x, axis=axis)

    @staticmethod
    def concat(xs, axis):
        return np.concatenate(xs, axis=axis)

    @staticmethod
    def ones(shape, dtype=np.float32):
        return np.ones(shape, dtype=dtype)

    @staticmethod
    def zeros(shape, dtype=np.float32):
        return np.zeros(shape, dtype=dtype)

    @staticmethod
    def cond(pred, true_fn=None, false_fn=None):
        return true_fn() if pred else false_fn()

    @staticmethod
    def equal(a, b):
        return np.equal(a, b)

    @staticmethod
    def stop_gradient(x):
        return x

    @staticmethod
    def min(a, b):
        return np.minimum(a, b)

    @staticmethod
    def norm(x, axis=None):
        return np.linalg.norm(x, axis=axis)

    @staticmethod
    def create_rng(seed):
        return np.random.Generator(np.random.SFC64(seed))

    @staticmethod
    def uniform(gen, shape, low, high, dtype=np.float32):
        return gen.uniform(low, high, size=shape).astype(dtype)

    @staticmethod
    def loop(body_fn, state, steps, counter=0):
        """The contract CartPole/cartpole_equations.py:251-258 shows at its call site (the class itself is in the absent
        SI_Toolkit submodule): body_fn(counter, *state) -> (counter, *state), `steps` times; returns (counter, *state)."""
        for _ in range(int(steps)):
            counter, *state = body_fn(counter, *state)
        return (counter, *state)


class _Unavailable:
    def __init__(self, *a, **k):
        pass


def _compile_adaptive(arg):
    """CompileAdaptive used both as ``@CompileAdaptive`` and ``CompileAdaptive(lib)(fn)``."""
    if callable(arg) and not isinstance(arg, NumpyLibrary) and not isinstance(arg, type):
        return arg
    return lambda fn: fn


# --------------------------------------------------------------------------- Control_Toolkit stand-ins
class cost_function_base:
    """Shape of Control_Toolkit.Cost_Functions.cost_function_base as the in-tree plugins use it.

    ``get_trajectory_cost`` follows the two in-tree witnesses (SUM over the horizon):
    Control_Toolkit_ASF/Cost_Functions/GymlikeCartPole/cost_function_gym.py:18-21 and
    Control_Toolkit_ASF/Controllers/controller_mppi_cartpole.py:194-199.
    """

    def __init__(self, variable_parameters, lib):
        self.variable_parameters = variable_parameters
        self.lib = lib
        self.logged_attributes = {}

    def set_logged_attributes(self, d):
        self.logged_attributes = d

    def get_trajectory_cost(self, state_horizon, inputs, previous_input=None):
        stage = self.get_stage_cost(state_horizon[:, :-1, :], inputs, previous_input)
        return self.lib.sum(stage, 1) + self.lib.reshape(self.get_terminal_cost(state_horizon[:, -1, :]), (-1,))


class template_controller:
    def __init__(self, environment_name=None, initial_environment_attributes=None, control_limits=None):
        self.variable_parameters = SimpleNamespace(**(initial_environment_attributes or {}))
        self.control_limits = control_limits
        self.lib = NumpyLibrary()

    def update_attributes(self, updated_attributes):
        for k, v in updated_attributes.items():
            setattr(self.variable_parameters, k, v)


# --------------------------------------------------------------------------- SI_Toolkit.Predictors stand-ins
class predictor_ODE_v0:
    """Only a loop over the IN-TREE per-step hook: out[:,0]=s0; out[:,k+1]=hook.step(out[:,k], Q[:,k])."""
    predictor_type = "ODE_v0"

    def __init__(self, horizon, dt, intermediate_steps=10, batch_size=1, variable_parameters=None, **kwargs):
        from SI_Toolkit_ASF.ToolkitCustomization.predictors_customization_v0 import next_state_predictor_ODE_v0
        self.horizon = horizon
        self.batch_size = batch_size
        self.next_step_predictor = next_state_predictor_ODE_v0(dt, intermediate_steps, batch_size,
                                                               variable_parameters=variable_parameters)
        self.params = self.next_step_predictor.cpe.params

    def predict(self, initial_state, Q):
        initial_state = np.asarray(initial_state, dtype=np.float32)
        Q = np.asarray(Q, dtype=np.float32)
        if initial_state.ndim == 1:
            initial_state = initial_state[np.newaxis, :]
        if Q.ndim == 2:
            Q = Q[np.newaxis, :, :]
        if initial_state.shape[0] == 1 and Q.shape[0] != 1:
            initial_state = np.tile(initial_state, (Q.shape[0], 1))
        H = Q.shape[1]
        out = np.zeros((Q.shape[0], H + 1, initial_state.shape[1]), dtype=np.float32)
        out[:, 0, :] = initial_state
        for k in range(H):
            out[:, k + 1, :] = self.next_step_predictor.step(out[:, k, :], Q[:, k, :])
        return out

    predict_core = predict

    def update(self, Q0, s):
        pass


class PredictorWrapper:
    def __init__(self):
        self.predictor = None
        self.predictor_config = {"predictor_type": "ODE_v0", "model_name": None, "intermediate_steps": 10}
        self.predictor_type = "ODE_v0"
        self.horizon = None
        self.variable_parameters = None

    def configure(self, batch_size, horizon, dt, predictor_specification=None, variable_parameters=None, **kw):
        self.horizon = horizon
        self.predictor = predictor_ODE_v0(horizon=horizon, dt=dt,
                                          intermediate_steps=self.predictor_config["intermediate_steps"],
                                          batch_size=batch_size, variable_parameters=variable_parameters)

    def predict(self, s, Q):
        return self.predictor.predict(s, Q)

    def predict_core(self, s, Q):
        return self.predictor.predict_core(s, Q)

    def update(self, Q0, s):
        pass


def _load_yaml(path, mode="r", return_path=False):
    with open(path, mode) as f:
        cfg = yaml.safe_load(f)
    return (cfg, path) if return_path else cfg


def install():
    """Register every stand-in; afterwards the in-tree hot-path modules import from REFERENCE_ROOT unmodified."""
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    _module("numba", jit=_jit, njit=_jit)
    # package stubs: skip the heavy CartPole/__init__.py (imports the whole app)
    for name in ("CartPole", "others"):
        m = _module(name)
        m.__path__ = [f"{REFERENCE_ROOT}/{name}"]
    _module("tensorflow")
    _module("engineering_notation", EngNumber=_Unavailable)
    ruamel = _module("ruamel")
    ruamel.__path__ = []
    _module("ruamel.yaml", YAML=_Unavailable)

    si = _module("SI_Toolkit")
    si.__path__ = []
    _module("SI_Toolkit.load_and_normalize", load_yaml=_load_yaml)
    _module("SI_Toolkit.computation_library", NumpyLibrary=NumpyLibrary, PyTorchLibrary=_Unavailable,
            TensorFlowLibrary=_Unavailable, TensorType=np.ndarray, ComputationLibrary=NumpyLibrary)
    _module("SI_Toolkit.Compile", CompileAdaptive=_compile_adaptive)
    preds = _module("SI_Toolkit.Predictors")
    preds.__path__ = []
    _module("SI_Toolkit.Predictors.predictor_ODE_v0", predictor_ODE_v0=predictor_ODE_v0)
    _module("SI_Toolkit.Predictors.predictor_wrapper", PredictorWrapper=PredictorWrapper)

    ct = _module("Control_Toolkit")
    ct.__path__ = []
    _module("Control_Toolkit.Cost_Functions", cost_function_base=cost_function_base)
    _module("Control_Toolkit.Controllers", template_controller=template_controller)


# --------------------------------------------------------------------------- the simulator class itself (CartPole/__init__.py)
class EnvironmentBatched:
    """Base class of CartPole.CartPole in the absent Control_Toolkit (only inherited from, CartPole/__init__.py:76)."""


class FunctionalDict(dict):
    """SI_Toolkit.Functions.FunctionalDict as CartPole/__init__.py:221-259, 413-433 uses it: a dict of zero-argument
    callables whose VALUES are what item access / items() / values() yield."""

    def __getitem__(self, k):
        return dict.__getitem__(self, k)()

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


class HistoryClass(dict):
    """... and its HistoryClass: name -> list of logged values (add_keys, update_history; read as a plain dict of lists)."""

    def add_keys(self, keys):
        for k in keys:
            self.setdefault(k, [])

    def update_history(self, d):
        for k, v in d.items():
            self[k].append(v)


class _Repo:
    """git.Repo(search_parent_directories=True).head.object.hexsha (CartPole/csv_logger.py:21-22)."""

    def __init__(self, *a, **k):
        self.head = SimpleNamespace(object=SimpleNamespace(hexsha="ref_shims"))


def install_app(controllers):
    """After install(): make the simulator class itself importable - the experiment loop of CartPole/__init__.py
    (update_state :283-324, update_target_position :360-378, update_target_equilibrium :380-388, Update_Q :475-527,
    save_csv_routine :403-433, setup / run_cartpole_random_experiment :570-735) and CartPole/data_generator.py's
    random_experiment_setter run as the reference's OWN code objects.  `controllers`: name -> class, what the absent
    Control_Toolkit's import_controller_by_name would find.  Returns the module object of CartPole/__init__.py."""
    import importlib.util
    import matplotlib
    matplotlib.use("Agg")
    template_controller.has_optimizer = False          # (class attribute of the absent template_controller; the legacy controller has none)
    o = _module("Control_Toolkit.others")
    o.__path__ = []
    _module("Control_Toolkit.others.environment", EnvironmentBatched=EnvironmentBatched)
    _module("Control_Toolkit.others.globals_and_utils",
            get_available_controller_names=lambda: ["manual-stabilization"] + list(controllers),
            get_available_optimizer_names=lambda: [],
            get_controller_name=lambda controller_name=None, controller_idx=None: (controller_name, 0),
            get_optimizer_name=lambda optimizer_name=None, optimizer_idx=None: (optimizer_name, 0),
            import_controller_by_name=lambda name: controllers[name])
    fn = _module("SI_Toolkit.Functions")
    fn.__path__ = []
    _module("SI_Toolkit.Functions.FunctionalDict", FunctionalDict=FunctionalDict, HistoryClass=HistoryClass)
    _module("git", Repo=_Repo)
    spec = importlib.util.spec_from_file_location("CartPole_app", f"{REFERENCE_ROOT}/CartPole/__init__.py")
    app = importlib.util.module_from_spec(spec)
    sys.modules["CartPole_app"] = app
    spec.loader.exec_module(app)
    sys.modules["CartPole"].CartPole = app.CartPole    # `from CartPole import CartPole` (CartPole/data_generator.py:8)
    return app
