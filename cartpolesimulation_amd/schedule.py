"""The data generator's EXPERIMENT SCHEDULE for a whole batch of experiments.

In the reference every experiment is one CartPole instance that, on EVERY simulation step, moves its target position along a random
trace and flips its target equilibrium after a dwell time (CartPole/__init__.py:283-293, :360-388), and random_experiment_setter
(CartPole/data_generator.py:93-218) draws the initial state and the trace's parameters per experiment.  All of it is a function of
time alone, so here it is tabulated ONCE per batch on the host - `ExperimentBatch`: initial states [E,6] and tables [rows, E] -
and the device loop (harness.py, cpmppi_plant_step) indexes the tables with its own step counter.

Random streams: the setter's own generator (SFC64(config seed), one for the run, consumed experiment after experiment in the
reference's order) and one generator per experiment for the turning points (the reference: CartPole.rng_CartPole, clock-seeded by the
shipped YAML; here SFC64(cartpole_seed + experiment index)).  With the same seeds the tables equal the reference's doubles bit for
bit (tests/test_schedule_host.py against tests/golden/schedule.npz).
"""
from dataclasses import dataclass, field
from math import gcd

import numpy as np

f32 = np.float32
INTERPOLATION_TYPES = ("previous", "linear", "0-derivative-smooth")


def default_data_gen_config():
    """config_data_gen.yml as shipped (the keys the experiment setter and the file naming read)."""
    return dict(controller="mpc", ML_Pipeline_mode=False, split=[0.8, 0.1],
                PATH_TO_EXPERIMENT_RECORDINGS_DEFAULT="./Experiment_Recordings/", seed=None, length_of_experiment=360,
                random_initial_state=dict(position=None, positionD=None, angle=None, angleD=None, target_position=None,
                                          init_limits=dict(angle=[0.0, 180.0], angleD=1200.0, position=0.8, positionD=0.5)),
                start_at_target=True, track_fraction_usable_for_target_position=1.0, target_position_end=None,
                initial_target_equilibrium="up", keep_target_equilibrium_x_seconds_up=10,
                keep_target_equilibrium_x_seconds_down=2.5, dt=dict(simulation=0.002, control=0.02, saving=0.02),
                turning_points=dict(track_relative_complexity=1, interpolation_type=["previous", "0-derivative-smooth"],
                                    turning_points=None, turning_points_period="regular"),
                save_mode="online", number_of_experiments=1)


def merged_config(overrides=None, base=None):
    """The shipped config with `overrides` applied (nested dicts are merged key by key)."""
    import copy
    cfg = copy.deepcopy(base if base is not None else default_data_gen_config())
    for k, v in (overrides or {}).items():
        if isinstance(v, dict) and isinstance(cfg.get(k), dict):
            cfg[k].update(v)
        else:
            cfg[k] = v
    return cfg


def _inf(v):
    return np.inf if isinstance(v, str) and v == "inf" else v


@dataclass
class ExperimentBatch:
    """E experiments' worth of schedule.  Simulation step g = 0 is the initial state; table row j holds the values the simulator
    has AFTER step j * stride (stride = gcd of the control and saving periods: every step at which anything reads them)."""
    s0: np.ndarray                      # [E,6] float32
    times: np.ndarray                   # [n_sim + 1] float64: time after g calls of step_time (accumulated, not g * dt)
    target_position: np.ndarray         # [rows, E] float64
    target_equilibrium: np.ndarray      # [rows, E] int32 (+1 / -1)
    stride: int
    n_sim: int                          # simulation steps of the experiment: ceil(length / dt_simulation)
    n_ctrl: int                         # simulation steps per controller update
    n_save: int                         # simulation steps per saved row
    dt_simulation: float
    dt_control: float
    dt_save: float
    length_of_experiment: float
    interpolation_type: list = field(default_factory=list)
    L: np.ndarray = None                # [E] float32 constant pole length per experiment, or None (the handle's default)
    L_table: np.ndarray = None          # [rows_L, E] float32 with its own stride 1 (per simulation step), or None
    m_pole_table: np.ndarray = None     # [n_sim + 1, E] float32 per simulation step: the PLANT's pole mass, or None (the handle's)
    informed: np.ndarray = None         # [n_sim + 1, E] bool per simulation step: is the controller handed the true pole length /
                                        # mass (informer_table) or the initial ones; None = always (mode 'ON', as shipped)
    Q_disturbance: np.ndarray = None    # [n_periods + 1, E] float32 = controlDisturbance * N(0, 1) per controller call, or None:
    Q_bias: float = 0.0                 # the plant is driven by Q_applied = (Q_calculated + Q_disturbance) + Q_bias (float32)
    # the measurement chain between plant and controller (CartPole.add_noise_and_latency): all off / None as shipped
    latency: float = 0.0                # seconds; the controller is handed the state latency / dt_simulation steps back (interpolated)
    measurement_noise: np.ndarray = None  # [n_periods + 1, E, 4] float32 = sigma * N(0, 1) for (angle, position, angleD, positionD) of the
                                        # state handed to controller call k (row 0, the t = 0 call, is unused: it sees the true state)
    angle_offset: np.ndarray = None     # [n_sim + 1, E] float64 per simulation step: the vertical angle offset added to the measured angle
                                        # (taken out again for a controller that is informed)

    @property
    def E(self):
        return self.s0.shape[0]

    @property
    def n_periods(self):
        """control periods the plant is advanced by = controller calls - 1 (the last call only completes the last row)."""
        return self.n_sim // self.n_ctrl

    def rows_at(self, steps):
        return np.minimum(np.asarray(steps) // self.stride, self.target_position.shape[0] - 1)


class RandomExperimentSetter:
    """random_experiment_setter (CartPole/data_generator.py:93-218) for any number of consecutive experiments."""

    def __init__(self, config=None, track_half_length=None):
        self.config = c = merged_config(config)
        if c["seed"] is None:
            raise ValueError("config['seed'] is empty: the reference then seeds from the clock; give a seed for a reproducible batch")
        self.rng = np.random.Generator(np.random.SFC64(c["seed"]))                  # others/globals_and_utils.py:198-214
        self.thl = f32((44.0e-2 - 4.4e-2) / 2.0) if track_half_length is None else f32(track_half_length)
        self._interp_idx = 0

    # -- CartPole/data_generator.py:221-256, in the reference's draw order and its float32 / float64 mix
    def _initial_state(self):
        ris = self.config["random_initial_state"]
        lim = ris["init_limits"]
        rng, thl = self.rng, self.thl
        s = np.zeros(6, dtype=f32)
        s[4] = f32(rng.uniform(low=-1.0, high=1.0)) * thl * f32(lim["position"]) if ris["position"] is None else ris["position"]
        s[5] = f32(rng.uniform(low=-1.0, high=1.0)) * thl * f32(lim["positionD"]) if ris["positionD"] is None else ris["positionD"]
        if ris["angle"] is None:
            lo, hi = lim["angle"]
            s[0] = (rng.uniform(low=lo, high=hi) if rng.uniform() > 0.5 else rng.uniform(low=-hi, high=-lo)) * (np.pi / 180.0)
        else:
            s[0] = ris["angle"]
        s[1] = rng.uniform(low=-1.0, high=1.0) * lim["angleD"] * (np.pi / 180.0) if ris["angleD"] is None else ris["angleD"]
        s[2], s[3] = np.cos(s[0]), np.sin(s[0])
        return s

    def _one(self, cartpole_rng):
        """What one `set` call fixes: initial state, turning points (times, values), interpolation type, initial equilibrium."""
        c = self.config
        s0 = self._initial_state()                                                   # :155
        frac = c["track_fraction_usable_for_target_position"]
        if c["start_at_target"]:
            start = s0[4]                                                            # :157-158
        elif c["random_initial_state"]["target_position"] is None:
            start = f32(frac) * self.thl * f32(self.rng.uniform(-1.0, 1.0))          # :160-162
        else:
            start = c["random_initial_state"]["target_position"]
        end = f32(frac) * self.thl * f32(self.rng.uniform(-1.0, 1.0)) if c["target_position_end"] is None else c["target_position_end"]
        ite = c["initial_target_equilibrium"]
        if ite in ("up", 1):
            te = 1
        elif ite in ("down", -1):
            te = -1
        elif ite == "random":                                                        # (the reference: numpy's GLOBAL generator, :177)
            te = int(2 * (self.rng.uniform() > 0.5) - 1)
        else:
            raise ValueError(f"{ite!r} is not a valid specification for target equilibrium")
        it = c["turning_points"]["interpolation_type"]
        if isinstance(it, (list, tuple)):                                            # :181-185
            interp = it[self._interp_idx]
            self._interp_idx = (self._interp_idx + 1) % len(it)
        else:
            interp = it
        if interp not in INTERPOLATION_TYPES:
            raise ValueError("Unknown interpolation type.")
        t_knots, y_knots = turning_points(c["length_of_experiment"], cartpole_rng, c["turning_points"]["track_relative_complexity"],
                                          c["turning_points"]["turning_points"], c["turning_points"]["turning_points_period"],
                                          start, end, frac, self.thl)
        return s0, t_knots, y_knots, interp, te

    def skip(self, k):
        """Advance the run by k experiments without tabulating them (a rank that owns a later block of the run's experiments)."""
        for _ in range(int(k)):
            self._one(np.random.Generator(np.random.SFC64(0)))
        return self

    def draw(self, E, cartpole_seed, L=None, stride=None):
        """The next E experiments of the run -> ExperimentBatch.  ``stride``: simulation steps per table row (default: the gcd of
        the control and saving periods; 1 when per-step parameter tables are to be attached, apply_parameter_schedule)."""
        c = self.config
        dt_sim, dt_ctrl, dt_save = c["dt"]["simulation"], c["dt"]["control"], c["dt"]["saving"]
        n_ctrl = max(1, int(np.rint(dt_ctrl / dt_sim)))                              # CartPole/__init__.py:909-916
        n_save = max(1, int(np.rint(dt_save / dt_sim)))                              # :925-933
        length = c["length_of_experiment"]
        n_sim = int(np.ceil(length / dt_sim))                                        # :648
        times = accumulated_times(n_sim, dt_sim)
        if stride is None:
            stride = gcd(n_ctrl, n_save)
        elif stride < 1 or gcd(n_ctrl, n_save) % int(stride):
            raise ValueError("stride must divide the control and the saving period")
        stride = int(stride)
        steps = np.arange(0, n_sim + 1, stride)
        frac = c["track_fraction_usable_for_target_position"]
        hi = np.float64(f32(frac) * self.thl)                                        # random_target_generator.py:85 (float32 bounds)
        s0 = np.empty((E, 6), f32)
        tp = np.empty((len(steps), E))
        te0 = np.empty(E, np.int32)
        interps = []
        # update_target_position (:360-378) stops updating once time >= length: the value of the last simulation step before that
        # is kept (the run's final step, whose accumulated time may or may not have reached the length)
        g_last = int(np.flatnonzero(times < length)[-1])
        t_eval = times[np.minimum(steps, g_last)]
        for e in range(E):
            s0[e], tk, yk, interp, te0[e] = self._one(np.random.Generator(np.random.SFC64(int(cartpole_seed) + e)))
            interps.append(interp)
            tp[:, e] = np.clip(evaluate_trace(tk, yk, interp, t_eval), -hi, hi)
        te = equilibrium_table(times, steps, te0, _inf(c["keep_target_equilibrium_x_seconds_up"]),
                               _inf(c["keep_target_equilibrium_x_seconds_down"]))
        Lv = None if L is None else np.broadcast_to(np.asarray(L, f32), (E,)).copy()
        return ExperimentBatch(s0=s0, times=times, target_position=tp, target_equilibrium=te, stride=stride, n_sim=n_sim, n_ctrl=n_ctrl,
                               n_save=n_save, dt_simulation=dt_sim, dt_control=dt_ctrl, dt_save=dt_save, length_of_experiment=length,
                               interpolation_type=interps, L=Lv)


def accumulated_times(n, dt):
    """time after g = 0..n calls of CartPole.step_time (CartPole/__init__.py:326-327: time = time + dt, in float64)."""
    return np.concatenate([[0.0], np.cumsum(np.full(n, float(dt)))]) if n else np.zeros(1)


def turning_points(length, rng, complexity, given, period, start, end, used_fraction, thl):
    """The knots of Generate_Random_Trace_Function (CartPole/random_target_generator.py:24-68) -> (times[n], values[n])."""
    if given is None or len(given) == 0:
        n = int(np.floor(length * complexity))
        y = rng.uniform(-1.0, 1.0, n) * used_fraction * np.float64(thl)
        if n == 0:
            y = np.zeros(2)
        elif n == 1:
            if start is not None:
                y[0] = start
            elif end is not None:
                y[0] = end
            y = np.array([y[0], y[0]])
        else:
            if start is not None:
                y[0] = start
            if end is not None:
                y[-1] = end
    else:
        n = len(given)
        y = np.array([given[0], given[0]] if n == 1 else given, dtype=np.float64)
    inner = max(n - 2, 0)
    if period == "random":
        t = np.concatenate([[0.0], np.sort(rng.uniform(0.0, 1.0, inner)), [1.0]])
    elif period == "regular":
        t = np.linspace(0, 1.0, num=inner + 2, endpoint=True)
    else:
        raise NotImplementedError("There is no mode corresponding to this value of turning_points_period variable")
    return t * length, y


def evaluate_trace(tk, yk, interpolation_type, t):
    """The interpolant the reference builds over the knots (random_target_generator.py:70-79), evaluated at the times `t`:
    scipy's interp1d 'previous' / 'linear' with extrapolation, BPoly.from_derivatives with zero slopes and periodic extension -
    here in closed form (no scipy), the same doubles."""
    t = np.asarray(t, dtype=np.float64)
    if interpolation_type == "previous":
        # the knot at or before t (scipy shifts the knots one ulp down so that t == knot selects that knot)
        idx = np.searchsorted(np.nextafter(tk, -np.inf), t, side="left")
        return yk[np.clip(idx, 1, len(tk)) - 1]
    if interpolation_type == "linear":
        hi = np.clip(np.searchsorted(tk, t), 1, len(tk) - 1)
        lo = hi - 1
        return (yk[hi] - yk[lo]) / (tk[hi] - tk[lo]) * (t - tk[lo]) + yk[lo]
    if interpolation_type == "0-derivative-smooth":
        # cubic Hermite with zero end slopes per interval = Bernstein coefficients [a, a, b, b]; periodic in the experiment's length
        t = tk[0] + (t - tk[0]) % (tk[-1] - tk[0])
        lo = np.clip(np.searchsorted(tk, t, side="right") - 1, 0, len(tk) - 2)
        s = (t - tk[lo]) / (tk[lo + 1] - tk[lo])
        r = 1.0 - s
        a, b = yk[lo], yk[lo + 1]
        return a * r * r * r + a * 3.0 * r * r * s + b * 3.0 * r * s * s + b * s * s * s
    raise ValueError("Unknown interpolation type.")


def equilibrium_table(times, steps, te0, keep_up, keep_down):
    """update_target_equilibrium (CartPole/__init__.py:380-388) over all simulation steps, sampled at `steps`: the first update
    only starts the dwell clock; afterwards the equilibrium flips when the time since the last change EXCEEDS the dwell time of
    the side it is on.  It depends on the initial side only: computed once per side, shared by the experiments."""
    out = np.empty((len(steps), len(te0)), np.int32)
    want = set(int(g) for g in steps)
    for side in (1, -1):
        cols = np.flatnonzero(te0 == side)
        if not len(cols):
            continue
        cur, last = side, None
        seq = np.empty(len(steps), np.int32)
        k = 0
        if 0 in want:
            seq[k] = cur
            k += 1
        for g in range(1, len(times)):
            t = times[g]
            if last is None:
                last = t
            elif (t - last) > (keep_down if cur == -1 else keep_up):
                last, cur = t, -cur
            if g in want:
                seq[k] = cur
                k += 1
        out[:, cols] = seq[:, None]
    return out


def parameter_table(updater, times, py_random=None, np_random=None, dtype=f32, init=None, time_after_step=False):
    """A physical parameter that changes in time (cartpole_physical_parameters.yml `L:` / `m_pole:` blocks; CartPole/parameter_updater.py
    ParameterUpdater, called by CartPole.update_parameters at the START of every simulation step with the time BEFORE the step,
    CartPole/__init__.py:529-537) -> float32 [len(times)]: entry g is the value the simulator holds DURING step g (entry 0: the
    initial value).  Modes as in the reference: 'constant', 'increase', 'bounce', 'random walk', 'random', 'random_gaussian';
    `change_every_x_seconds` (empty = every step), `reset_every_x_seconds`, `range_clip`.  The value is carried in float32 (the
    reference keeps it in a 0-d float32 array: `current + increment` and the clip are float32 operations).  Random modes draw from
    `py_random` (random.Random: the reference uses the module-level random()) / `np_random` (numpy RandomState: np.random.uniform /
    normal); seeded like the reference's globals they give the reference's own sequence.
    The vertical angle offset is the same updater held in a float64 (``dtype``), started from ``init`` = deg2rad(init_value) and
    called AFTER the step's time update (``time_after_step``; CartPole/__init__.py:142-143, 348-352)."""
    import random as _random
    u = dict(updater)
    inf = lambda v: np.inf if isinstance(v, str) and v == "inf" else v                      # noqa: E731
    change_every, reset_every = inf(u["change_every_x_seconds"]), inf(u["reset_every_x_seconds"])
    mode, increment, clip = u["mode"], u["increment"], u["range_clip"]
    py_random = py_random or _random.Random(0)
    np_random = np_random or np.random.RandomState(0)
    f = dtype
    reset_to = u["init_value"]
    if reset_to == "random":
        reset_to = np_random.uniform(*u["range_random"])
    cur = f(reset_to if init is None else init)
    last_change = last_reset = 0.0
    direction = 1
    out = np.empty(len(times), f)
    out[0] = cur
    init = reset_to
    for g in range(1, len(times)):
        t = times[g] if time_after_step else times[g - 1]
        if change_every and t - last_change < change_every:
            pass
        elif reset_every and mode != "constant" and t - last_reset >= reset_every:
            last_reset = t
            cur = f(init)
        else:
            last_change = t
            if mode in ("random", "random_gaussian"):
                cur = f(np_random.uniform(*u["range_random"]) if mode == "random" else np_random.normal(init, increment))
                out[g] = cur
                continue
            if mode == "constant":
                inc = 0.0
            elif mode == "random walk":
                inc = (1.0 if py_random.random() < 0.5 else -1.0) * increment
            elif mode == "increase":
                inc = increment
            elif mode == "bounce":
                inc = direction * increment
                nxt = cur + f(inc)
                if nxt >= clip[1] or nxt <= clip[0]:
                    direction = -direction
            else:
                raise ValueError("mode with value {} not valid".format(mode))
            cur = cur + f(inc)
            if clip:
                cur = f(np.clip(cur, f(clip[0]), f(clip[1])))
        out[g] = cur
    return out


def parameter_tables(updater, times, seeds, dtype=f32, init=None, time_after_step=False):
    """parameter_table for MANY experiments at once -> [len(times), len(seeds)]: column e is parameter_table(updater, times,
    random.Random(seeds[e]), numpy.random.RandomState(seeds[e]), ...).  WHEN the value changes is the same for every experiment (a
    function of time only); only what it changes to differs, so the per-step state machine runs once and every change is applied to
    all columns together, each column drawing from its own generators in its own order."""
    import random as _random
    u = dict(updater)
    inf = lambda v: np.inf if isinstance(v, str) and v == "inf" else v                      # noqa: E731
    change_every, reset_every = inf(u["change_every_x_seconds"]), inf(u["reset_every_x_seconds"])
    mode, increment, clip = u["mode"], u["increment"], u["range_clip"]
    if mode not in ("constant", "random walk", "increase", "random", "random_gaussian", "bounce"):
        raise ValueError("mode with value {} not valid".format(mode))
    f, E, n = dtype, len(seeds), len(times)
    rs = [np.random.RandomState(int(sd)) for sd in seeds]
    # the schedule of events: 0 = keep, 1 = reset, 2 = change
    events = np.zeros(n, np.int8)
    last_change = last_reset = 0.0
    for g in range(1, n):
        t = times[g] if time_after_step else times[g - 1]
        if change_every and t - last_change < change_every:
            continue
        if reset_every and mode != "constant" and t - last_reset >= reset_every:
            last_reset = t
            events[g] = 1
        else:
            last_change = t
            events[g] = 2
    n_changes = int((events == 2).sum())
    reset_to = np.empty(E, np.float64)
    if u["init_value"] == "random":                               # the first draw of every experiment's numpy generator
        for e in range(E):
            reset_to[e] = rs[e].uniform(*u["range_random"])
    else:
        reset_to[:] = u["init_value"]
    cur = reset_to.astype(f) if init is None else np.broadcast_to(np.asarray(init, np.float64), (E,)).astype(f)
    draws = None
    if mode == "random":
        draws = np.stack([r.uniform(*u["range_random"], size=n_changes) for r in rs], axis=1) if n_changes else np.empty((0, E))
    elif mode == "random_gaussian":
        draws = np.stack([r.normal(reset_to[e], increment, size=n_changes) for e, r in enumerate(rs)], axis=1) if n_changes else np.empty((0, E))
    elif mode == "random walk":
        draws = np.empty((n_changes, E))
        for e, sd in enumerate(seeds):
            pr = _random.Random(int(sd))
            draws[:, e] = [1.0 if pr.random() < 0.5 else -1.0 for _ in range(n_changes)]
    direction = np.ones(E)
    out = np.empty((n, E), f)
    out[0] = cur
    k = 0
    for g in range(1, n):
        ev = events[g]
        if ev == 1:
            cur = reset_to.astype(f)
        elif ev == 2:
            if mode in ("random", "random_gaussian"):
                cur = draws[k].astype(f)
            else:
                if mode == "constant":
                    inc = np.zeros(E)
                elif mode == "random walk":
                    inc = draws[k] * increment
                elif mode == "increase":
                    inc = np.full(E, increment)
                else:                                              # bounce
                    inc = direction * increment
                    nxt = cur + inc.astype(f)
                    direction = np.where((nxt >= clip[1]) | (nxt <= clip[0]), -direction, direction)
                cur = cur + inc.astype(f)
                if clip:
                    cur = np.clip(cur, f(clip[0]), f(clip[1])).astype(f)
            k += 1
        out[g] = cur
    return out


def informer_table(informer, times, n_ctrl, np_random=None):
    """`inform_controller_about_parameters_change` (cartpole_physical_parameters.yml; CartPole/controller_informer.py:5-50): the
    simulator asks its ControllerInformer at every controller update - simulation steps 0, n_ctrl, 2 n_ctrl, ... with the time AFTER
    the step (CartPole/__init__.py:495-500) - whether to hand the controller the TRUE pole length / mass or the initial ones, and
    logs the answer in the L_for_controller / m_pole_for_controller columns.  -> bool [len(times)]: the answer in force after step g.
    Modes 'ON', 'OFF', 'switching_regular' (on after x seconds off, off after y seconds on; starts off), 'switching_random' (the
    dwell times drawn from U(0, x) / U(0, y) of `np_random` - a numpy RandomState; the reference draws from numpy's global one)."""
    mode = informer["mode"]
    on_after, off_after = informer["change_to_on_after_x_seconds_off"], informer["change_to_off_after_x_seconds_on"]
    out = np.empty(len(times), bool)
    if mode in ("ON", "OFF"):
        out[:] = mode == "ON"
        return out
    if mode not in ("switching_regular", "switching_random"):
        raise ValueError(f"unknown informer mode {mode!r}")
    rnd = mode == "switching_random"
    if rnd:
        np_random = np_random or np.random.RandomState(0)
        on_after_now, off_after_now = np_random.uniform(0, on_after), np_random.uniform(0, off_after)
    else:
        on_after_now, off_after_now = on_after, off_after
    told, since_on, since_off = False, 0.0, 0.0
    for g in range(0, len(times), n_ctrl):                       # asked at the controller updates only; the answer holds in between
        t = times[g]
        if not told and t - since_off >= on_after_now:
            told, since_on = True, t
            if rnd:
                on_after_now = np_random.uniform(0, on_after)
        elif told and t - since_on >= off_after_now:
            told, since_off = False, t
            if rnd:
                off_after_now = np_random.uniform(0, off_after)
        out[g:g + n_ctrl] = told
    return out


def control_disturbance(E, n_calls, seed, first=0):
    """The standard normal draws behind the simulator's additive control disturbance (CartPole/noise_control_signal.py:14-16;
    `controlDisturbance_mode: additive`, the shipped mode - with amplitude 0): ONE generator for all experiments of a run (the
    module-level `rng`, CartPole/__init__.py:75, SFC64(cartpole seed)), float32 draws; every experiment takes two draws outside its run
    (set_cartpole_state_at_t0 when its controller is set, :792, and when the simulator is reset, :733) and then one per controller
    call (:881-882 at t = 0, :523-524 inside the loop).  -> float32 [n_calls, E] for the experiments first .. first + E - 1 of the
    run: the reference's own numbers for a seeded run, whatever the split over processes."""
    per = int(n_calls) + 2
    gen = np.random.Generator(np.random.SFC64(int(seed)))
    skip = int(first) * per                                        # the draws of the experiments before ours (SFC64 cannot jump ahead,
    while skip > 0:                                                # and a normal takes a variable number of words: drawn and dropped)
        n = min(skip, 1 << 22)
        gen.standard_normal(size=n, dtype=f32)
        skip -= n
    z = gen.standard_normal(size=int(E) * per, dtype=f32)
    return np.ascontiguousarray(z.reshape(-1, per)[:, 2:].T)


def measurement_noise(E, n_sim, n_ctrl, seed, sigmas):
    """The simulator's measurement noise (CartPole/noise_adder.py:71-82) as the controller calls see it: every simulation step draws
    four float32 standard normals - angle, position, angleD, positionD, in this order - from the instance's generator
    (SFC64(cartpole seed), one NoiseAdder per experiment: every experiment of a seeded run sees the SAME sequence) and scales each by
    its sigma in float32; only the draws of the steps that end a control period reach a controller.
    -> float32 [n_sim // n_ctrl + 1, E, 4]; row k belongs to controller call k (row 0: the t = 0 call sees the true state)."""
    z = np.random.Generator(np.random.SFC64(int(seed))).standard_normal(size=(int(n_sim), 4), dtype=f32)
    calls = n_sim // n_ctrl
    rows = np.zeros((calls + 1, 4), f32)
    rows[1:] = z[np.arange(1, calls + 1) * n_ctrl - 1]                           # simulation step g draws row g - 1
    rows *= np.array([f32(x) for x in sigmas], f32)                              # (python float * np.float32 -> float32 product)
    return np.array(np.broadcast_to(rows[:, None, :], (calls + 1, int(E), 4)))


def apply_parameter_schedule(batch, parameters, seed=0, first=0):
    """The simulator's time-varying physical parameters for a batch drawn with stride 1: `parameters` holds any of the blocks `L`,
    `m_pole` (ParameterUpdater configs) and `inform_controller_about_parameters_change` of cartpole_physical_parameters.yml's
    `cartpole:` section, and its control disturbance: `controlDisturbance` (+ `controlBias`, `controlDisturbance_mode` 'additive' /
    'OFF', `seed` = the section's own seed, which seeds the disturbance's generator: control_disturbance).
    -> the batch with L_table / m_pole_table / informed / Q_disturbance filled in (one column per experiment).  Deterministic
    modes give every experiment the same column, as in the reference; the random modes draw per experiment from generators seeded
    with seed + first + e (the reference draws from the process-global, clock-seeded generators: nothing to reproduce there)."""
    import dataclasses
    import random as _random
    per_step = [k for k in ("L", "m_pole", "inform_controller_about_parameters_change", "vertical_angle_offset") if parameters.get(k) is not None]
    if per_step and batch.stride != 1:
        raise ValueError("parameter tables are per simulation step: draw the batch with stride=1")
    E, out = batch.E, {}
    mode = parameters.get("controlDisturbance_mode", "additive")
    if mode not in ("additive", "OFF"):
        raise NotImplementedError(f"controlDisturbance_mode {mode!r}: 'additive' (the shipped mode) and 'OFF' are supported")
    amp, bias = float(parameters.get("controlDisturbance") or 0.0), float(parameters.get("controlBias") or 0.0)
    if mode == "additive" and (amp != 0.0 or bias != 0.0):
        if parameters.get("seed") is None:
            raise ValueError("parameters['seed'] is empty: the reference then seeds the disturbance from the clock; give a seed")
        z = control_disturbance(E, batch.n_periods + 1, parameters["seed"], first)
        out["Q_disturbance"], out["Q_bias"] = f32(amp) * z, float(f32(bias))
    for name, field_ in (("L", "L_table"), ("m_pole", "m_pole_table")):
        blk = parameters.get(name)
        if blk is None:
            continue
        if blk.get("mode") in ("constant", "increase", "bounce") and blk.get("init_value") != "random":
            col = parameter_table(blk, batch.times)               # the same for every experiment: tabulated once
            out[field_] = np.array(np.broadcast_to(col[:, None], (len(col), E)))
        else:
            out[field_] = parameter_tables(blk, batch.times, [int(seed) + first + e for e in range(E)])
    inf = parameters.get("inform_controller_about_parameters_change")
    if inf is not None:
        if inf.get("mode") != "switching_random":
            col = informer_table(inf, batch.times, batch.n_ctrl)
            out["informed"] = np.array(np.broadcast_to(col[:, None], (len(col), E)))
        else:
            out["informed"] = np.stack([informer_table(inf, batch.times, batch.n_ctrl, np.random.RandomState(int(seed) + first + e + 1))
                                        for e in range(E)], axis=1)
    # the measurement chain (add_noise_and_latency, CartPole/__init__.py:336-356)
    if parameters.get("latency"):
        lat = float(parameters["latency"])
        if lat < 0 or lat / batch.dt_simulation > 200:                               # latency_adder.py:8, 77-78
            raise ValueError("Not possible to add so much latency!")
        out["latency"] = lat
    noise = parameters.get("noise")
    if isinstance(noise, dict) and noise.get("noise_mode", "OFF") != "OFF":
        if parameters.get("seed") is None:
            raise ValueError("parameters['seed'] is empty: the reference then seeds the measurement noise from the clock; give a seed")
        out["measurement_noise"] = measurement_noise(E, batch.n_sim, batch.n_ctrl, parameters["seed"],
                                                     [noise["sigma_angle"], noise["sigma_position"], noise["sigma_angleD"], noise["sigma_positionD"]])
    vao = parameters.get("vertical_angle_offset")
    if vao is not None:
        init = np.deg2rad(vao["init_value"]) if vao["init_value"] != "random" else None
        kw = dict(dtype=np.float64, time_after_step=True)
        if vao.get("mode") in ("constant", "increase", "bounce") and init is not None:
            col = parameter_table(vao, batch.times, init=init, **kw)
            out["angle_offset"] = np.array(np.broadcast_to(col[:, None], (len(col), E)))
        else:
            seeds = [int(seed) + first + e + 2 for e in range(E)]
            if init is None:                                       # 'random' start: drawn in DEGREES like the YAML's number, per experiment
                i0 = np.deg2rad([np.random.RandomState(sd + 7).uniform(*np.rad2deg(vao["range_random"])) for sd in seeds])
                out["angle_offset"] = parameter_tables(dict(vao, init_value=0.0), batch.times, seeds, init=i0, **kw)
            else:
                out["angle_offset"] = parameter_tables(vao, batch.times, seeds, init=init, **kw)
    return dataclasses.replace(batch, **out)


def active_parameters(cartpole_section):
    """The blocks of cartpole_physical_parameters.yml's `cartpole:` section that make a run differ from one with constant parameters,
    an always-informed controller and no control disturbance -> the `parameters` dict of apply_parameter_schedule /
    recording.generate_dataset, or None when there is none (the shipped file: updaters 'constant', informer 'ON', disturbance 0)."""
    sec, out = cartpole_section, {}
    for name in ("L", "m_pole"):
        blk = sec.get(name)
        if isinstance(blk, dict) and (blk.get("mode", "constant") != "constant" or blk.get("init_value") == "random"):
            out[name] = dict(blk)
    inf = sec.get("inform_controller_about_parameters_change")
    if isinstance(inf, dict) and inf.get("mode", "ON") != "ON":
        out["inform_controller_about_parameters_change"] = dict(inf)
    if float(sec.get("latency") or 0.0) != 0.0:
        out["latency"] = float(sec["latency"])
    noise = sec.get("noise")
    if isinstance(noise, dict) and noise.get("noise_mode", "OFF") != "OFF":
        out["noise"] = dict(noise)
        out["seed"] = sec.get("seed")
    vao = sec.get("vertical_angle_offset")
    if isinstance(vao, dict) and (vao.get("mode", "constant") != "constant" or vao.get("init_value") not in (0, 0.0)):
        out["vertical_angle_offset"] = dict(vao)
    mode = sec.get("controlDisturbance_mode", "OFF")
    if mode != "OFF" and (float(sec.get("controlDisturbance") or 0.0) != 0.0 or float(sec.get("controlBias") or 0.0) != 0.0):
        out.update(controlDisturbance_mode=mode, controlDisturbance=sec.get("controlDisturbance"), controlBias=sec.get("controlBias"),
                   seed=sec.get("seed"))
    return out or None


def draw_shard(config, n_total, cartpole_seed, rank=0, world=1, L=None, stride=None):
    """Rank `rank` of `world` processes' share of a run of `n_total` experiments: the contiguous block shard.env_shard gives it, drawn
    from the SAME random streams as the single-process run (the union over the ranks is that run, experiment for experiment) - the
    share-nothing fan-out of others/EulerClusterScripts/ParallelDataGeneration.sh:2-17 with reproducible content.
    -> (ExperimentBatch, first experiment index)."""
    from .shard import env_shard
    start, count = env_shard(n_total, world, rank)
    setter = RandomExperimentSetter(config).skip(start)
    Lv = None if L is None else np.broadcast_to(np.asarray(L, f32), (n_total,))[start:start + count]
    return setter.draw(count, int(cartpole_seed) + start, L=Lv, stride=stride), start
