"""CPU restatement of the reference's EXPERIMENT SCHEDULE - what the data generator's simulator does around the controller
(SURVEY.md §8f N1 / N2): the random target-position trace, the target-equilibrium flips, the experiment setter and the order of
events inside one simulation step, down to the rows of the recording.

TEST INFRASTRUCTURE (like everything under oracle/): only tests/ may import it; the product's counterpart is
cartpolesimulation_amd/schedule.py + harness.py + the plant kernel.  Plain numpy, one experiment at a time, line by line after:
  * CartPole/random_target_generator.py:9-89          Generate_Random_Trace_Function      -> random_trace
  * CartPole/data_generator.py:95-218, 221-256         random_experiment_setter, generate_random_initial_state
  * CartPole/__init__.py:283-324 (update_state), :360-378 (update_target_position), :380-388 (update_target_equilibrium),
    :403-433 (save_csv_routine), :475-527 (Update_Q), :570-657 / :659-735 (setup / run_cartpole_random_experiment),
    :796-880 (set_cartpole_state_at_t0)                                                   -> run_experiment
Pinned by tests/golden/schedule.npz, which oracle/gen_golden_schedule.py produced by running those very code objects - the
simulator class itself with the in-tree legacy MPPI controller in the loop (tests/test_oracle_schedule.py).
scipy is NOT used: interp1d (kinds 'linear' and 'previous', fill_value='extrapolate') and BPoly.from_derivatives with zero
slopes are restated from their published algorithms (scipy 1.15: interpolate/_interpolate.py interp1d._call_linear /
_call_previousnext; _ppoly.pyx evaluate_bpoly1; BPoly._construct_from_derivatives gives c = [ya, ya, yb, yb] exactly when both
slopes are zero) and reproduce scipy's doubles bit for bit on the fixture.
"""
from dataclasses import replace

import numpy as np

from . import oracle_np as O

f32 = np.float32
THL32 = f32((44.0e-2 - 4.4e-2) / 2.0)             # CartPole/cartpole_parameters.py:31 (a 0-d float32 array there)


def _weak_times_thl(x):
    """python-float * TrackHalfLength: the 0-d float32 array wins (NEP 50), the product is formed in float32."""
    return f32(x) * THL32


# ------------------------------------------------------------------------------------------------ random_target_generator.py
def random_trace(length_of_experiment, rtf_rng, track_relative_complexity, interpolation_type, turning_points,
                 turning_points_period, start_random_target_position_at, end_random_target_position_at, used_track_fraction):
    """-> f(time) (scalar or array, float64), CartPole/random_target_generator.py:9-89."""
    if (turning_points is None) or (len(turning_points) == 0):                                # :24
        n = int(np.floor(length_of_experiment * track_relative_complexity))                   # :26
        y = rtf_rng.uniform(-1.0, 1.0, n)                                                     # :28
        y = y * used_track_fraction * np.float64(THL32)                                       # :29 (float64 array x float32 0-d)
        if n == 0:
            y = np.append(np.append(y, 0.0), 0.0)                                             # :31-33
        elif n == 1:
            if start_random_target_position_at is not None:                                   # :35-40
                y[0] = start_random_target_position_at
            elif end_random_target_position_at is not None:
                y[0] = end_random_target_position_at
            y = np.append(y, y[0])                                                            # :41
        else:
            if start_random_target_position_at is not None:                                   # :43-46
                y[0] = start_random_target_position_at
            if end_random_target_position_at is not None:
                y[-1] = end_random_target_position_at
    else:
        n = len(turning_points)                                                               # :49
        y = np.array([turning_points[0], turning_points[0]]) if n == 1 else np.array(turning_points)   # :52-55
        y = y.astype(np.float64)
    random_samples = n - 2 if n - 2 >= 0 else 0                                               # :57
    if turning_points_period == "random":
        t_init = np.sort(rtf_rng.uniform(0.0, 1.0, random_samples))                           # :60-62
        t_init = np.append(np.insert(t_init, 0, 0.0), 1.0)
    elif turning_points_period == "regular":
        t_init = np.linspace(0, 1.0, num=random_samples + 2, endpoint=True)                   # :64
    else:
        raise NotImplementedError("There is no mode corresponding to this value of turning_points_period variable")
    t_init = t_init * length_of_experiment                                                    # :68
    if interpolation_type == "0-derivative-smooth":
        inner = lambda t: _bpoly_zero_slopes(t_init, y, t)                                    # noqa: E731  (:71-73)
    elif interpolation_type == "linear":
        inner = lambda t: _interp_linear(t_init, y, t)                                        # noqa: E731  (:75)
    elif interpolation_type == "previous":
        inner = lambda t: _interp_previous(t_init, y, t)                                      # noqa: E731  (:77)
    else:
        raise ValueError("Unknown interpolation type.")
    # :82-87: the bounds are python-float x float32 0-d array -> float32, compared as doubles
    hi = np.float64(f32(used_track_fraction) * THL32)

    def truncated(time):
        scalar = np.ndim(time) == 0
        v = np.clip(inner(np.atleast_1d(np.asarray(time, dtype=np.float64))), -hi, hi)
        return float(v[0]) if scalar else v

    return truncated


def _interp_linear(x, y, t):
    """scipy interp1d(kind='linear', fill_value='extrapolate')._call_linear."""
    i = np.searchsorted(x, t).clip(1, len(x) - 1)
    lo, hi = i - 1, i
    slope = (y[hi] - y[lo]) / (x[hi] - x[lo])
    return slope * (t - x[lo]) + y[lo]


def _interp_previous(x, y, t):
    """scipy interp1d(kind='previous', fill_value='extrapolate')._call_previousnext: _ind = 0, side 'left',
    _x_shift = nextafter(x, -inf); below x[0] scipy fills nan (never evaluated there: time >= 0 = x[0])."""
    i = np.searchsorted(np.nextafter(x, -np.inf), t, side="left").clip(1, len(x))
    return y[i - 1]                                           # (_y[i + _ind - 1])


def _bpoly_zero_slopes(x, y, t):
    """BPoly.from_derivatives(x, [[y_i, 0]], extrapolate='periodic')(t): per interval the cubic with Bernstein coefficients
    [y_i, y_i, y_i+1, y_i+1]; periodic mapping t -> x0 + (t - x0) % (xN - x0) for every t (_PPolyBase.__call__)."""
    t = x[0] + (t - x[0]) % (x[-1] - x[0])
    i = (np.searchsorted(x, t, side="right") - 1).clip(0, len(x) - 2)          # x[i] <= t < x[i+1]; the last interval is closed
    s = (t - x[i]) / (x[i + 1] - x[i])
    s1 = 1.0 - s
    c0, c3 = y[i], y[i + 1]
    return c0 * s1 * s1 * s1 + c0 * 3.0 * s1 * s1 * s + c3 * 3.0 * s1 * s * s + c3 * s * s * s   # evaluate_bpoly1, k == 3


# ------------------------------------------------------------------------------------------------ data_generator.py
def generate_random_initial_state(stub, init_limits, rng):
    """CartPole/data_generator.py:221-256; `stub`: (position, positionD, angle, angleD) with None = draw it."""
    position_lim, positionD_lim, angle_lim, angleD_lim = init_limits
    position, positionD, angle, angleD = stub
    s = np.zeros(6, dtype=f32)
    s[O.POSITION_IDX] = _weak_times_thl(rng.uniform(low=-1.0, high=1.0)) * f32(position_lim) if position is None else position
    s[O.POSITIOND_IDX] = _weak_times_thl(rng.uniform(low=-1.0, high=1.0)) * f32(positionD_lim) if positionD is None else positionD
    if angle is None:
        if rng.uniform() > 0.5:
            s[O.ANGLE_IDX] = rng.uniform(low=angle_lim[0], high=angle_lim[1]) * (np.pi / 180.0)
        else:
            s[O.ANGLE_IDX] = rng.uniform(low=-angle_lim[1], high=-angle_lim[0]) * (np.pi / 180.0)
    else:
        s[O.ANGLE_IDX] = angle
    s[O.ANGLED_IDX] = rng.uniform(low=-1.0, high=1.0) * angleD_lim * (np.pi / 180.0) if angleD is None else angleD
    s[O.ANGLE_COS_IDX], s[O.ANGLE_SIN_IDX] = np.cos(s[O.ANGLE_IDX]), np.sin(s[O.ANGLE_IDX])     # float32 cos / sin of the stored angle
    return s


class ExperimentSetter:
    """CartPole/data_generator.py:93-218: one instance serves all experiments of a run (its rng and the alternation of the
    interpolation types carry over from one experiment to the next)."""

    def __init__(self, config):
        c = config
        self.c = c
        ris = c["random_initial_state"]
        self.stub = (ris["position"], ris["positionD"], ris["angle"], ris["angleD"])
        lim = ris["init_limits"]
        self.init_limits = [lim["position"], lim["positionD"], lim["angle"], lim["angleD"]]
        self.interpolation_type_idx = 0
        inf = lambda v: np.inf if isinstance(v, str) and v == "inf" else v                     # noqa: E731
        self.keep_up, self.keep_down = inf(c["keep_target_equilibrium_x_seconds_up"]), inf(c["keep_target_equilibrium_x_seconds_down"])
        self.rng = np.random.Generator(np.random.SFC64(c["seed"]))                              # create_rng, others/globals_and_utils.py:198-214

    def set(self, cartpole_rng):
        """-> dict(s0, trace f(t), target_equilibrium, interpolation_type, start, end); `cartpole_rng` = the CartPole
        instance's own generator, which draws the turning points (CartPole/__init__.py:631-646)."""
        c = self.c
        s0 = generate_random_initial_state(self.stub, self.init_limits, self.rng)               # :155
        frac = c["track_fraction_usable_for_target_position"]
        if c["start_at_target"]:
            start = s0[O.POSITION_IDX]                                                          # :157-158
        elif c["random_initial_state"]["target_position"] is None:
            start = f32(frac) * THL32 * f32(self.rng.uniform(-1.0, 1.0))                        # :160-162 (float32 products)
        else:
            start = c["random_initial_state"]["target_position"]
        if c["target_position_end"] is None:
            end = f32(frac) * THL32 * f32(self.rng.uniform(-1.0, 1.0))                          # :166-168
        else:
            end = c["target_position_end"]
        ite = c["initial_target_equilibrium"]
        if ite == "up" or ite == 1:
            te = 1
        elif ite == "down" or ite == -1:
            te = -1
        else:
            raise NotImplementedError("initial_target_equilibrium 'random' draws from numpy's global generator (:177)")
        it = c["turning_points"]["interpolation_type"]
        if isinstance(it, list):                                                                # :181-185
            interpolation_type = it[self.interpolation_type_idx]
            self.interpolation_type_idx = (self.interpolation_type_idx + 1) % len(it)
        else:
            interpolation_type = it
        f = random_trace(c["length_of_experiment"], cartpole_rng, c["turning_points"]["track_relative_complexity"],
                         interpolation_type, c["turning_points"]["turning_points"], c["turning_points"]["turning_points_period"],
                         start, end, frac)
        return dict(s0=s0, trace=f, target_equilibrium=te, interpolation_type=interpolation_type, start=start, end=end)


# ------------------------------------------------------------------------------------------------ the experiment loop
def accumulated_times(n, dt):
    """time after g calls of step_time (CartPole/__init__.py:326-327), g = 0..n."""
    t, out = 0.0, [0.0]
    for _ in range(n):
        t = t + dt
        out.append(t)
    return np.array(out)


def schedule_tables(trace, te0, length, dt_sim, keep_up, keep_down):
    """target_position and target_equilibrium as the simulator holds them AFTER each simulation step g = 0..n
    (update_target_position :360-378 - not updated once time >= length -, update_target_equilibrium :380-388)."""
    n = int(np.ceil(length / dt_sim))                                                           # :648
    times = accumulated_times(n, dt_sim)
    tp = np.empty(n + 1)
    te = np.empty(n + 1, dtype=np.int64)
    tp[0], te[0] = trace(0.0), te0                                                              # :651 (time 0)
    last = None
    for g in range(1, n + 1):
        t = times[g]
        tp[g] = trace(t) if not (t >= length) else tp[g - 1]
        cur = te[g - 1]
        if last is None:
            last = t
        elif cur == -1 and (t - last) > keep_down:
            last, cur = t, -cur
        elif cur == 1 and (t - last) > keep_up:
            last, cur = t, -cur
        te[g] = cur
    return times, tp, te


def controller_informer(cfg, np_random=None):
    """ControllerInformer (CartPole/controller_informer.py:5-50; `inform_controller_about_parameters_change`): is the controller
    handed the TRUE pole length / pole mass or the initial ones?  -> get(time_now) -> bool, to be called where the simulator calls
    get_parameters (at every controller update, CartPole/__init__.py:495-500, and - time 0 - when the experiment is set up).
    'switching_random' draws from `np_random` (the reference: numpy's global generator, two draws at construction)."""
    mode = cfg["mode"]
    on_after, off_after = cfg["change_to_on_after_x_seconds_off"], cfg["change_to_off_after_x_seconds_on"]
    st = dict(true=False, t_on=0.0, t_off=0.0)
    if mode == "switching_random":
        st["on_r"], st["off_r"] = np_random.uniform(0, on_after), np_random.uniform(0, off_after)     # :11-12

    def get(t):
        if mode == "OFF":
            st["true"] = False
        elif mode == "ON":
            st["true"] = True
        elif mode == "switching_regular":                                                       # :25-33
            if not st["true"]:
                if t - st["t_off"] >= on_after:
                    st["true"], st["t_on"] = True, t
            elif t - st["t_on"] >= off_after:
                st["true"], st["t_off"] = False, t
        elif mode == "switching_random":                                                        # :34-44
            if not st["true"]:
                if t - st["t_off"] >= st["on_r"]:
                    st["true"], st["t_on"] = True, t
                    st["on_r"] = np_random.uniform(0, on_after)
            elif t - st["t_on"] >= st["off_r"]:
                st["true"], st["t_off"] = False, t
                st["off_r"] = np_random.uniform(0, off_after)
        else:
            raise ValueError(mode)
        return st["true"]

    return get


def parameter_updater(cfg, py_random=None, np_random=None):
    """ParameterUpdater (CartPole/parameter_updater.py:6-76) -> update(current_value, time_now) -> new value.  The arithmetic is
    whatever `current_value` brings (the simulator keeps L / m_pole in 0-d float32 arrays, the vertical angle offset in a float64)."""
    inf = lambda v: np.inf if isinstance(v, str) and v == "inf" else v                          # noqa: E731
    st = dict(change=inf(cfg["change_every_x_seconds"]), reset=inf(cfg["reset_every_x_seconds"]), last_change=0.0, last_reset=0.0,
              direction=1, increment=cfg["increment"])
    mode, clip, init = cfg["mode"], cfg["range_clip"], cfg["init_value"]

    def update(cur, t):
        if st["change"] and t - st["last_change"] < st["change"]:                               # :32-33
            return cur
        if st["reset"] and mode != "constant" and t - st["last_reset"] >= st["reset"]:          # :34-36
            st["last_reset"] = t
            return init
        st["last_change"] = t
        if mode == "constant":
            inc = 0.0
        elif mode == "random walk":
            inc = (1.0 if py_random.random() < 0.5 else -1.0) * st["increment"]
        elif mode == "increase":
            inc = st["increment"]
        elif mode == "random":
            return np_random.uniform(*cfg["range_random"])
        elif mode == "random_gaussian":
            return np_random.normal(init, cfg["increment"])
        elif mode == "bounce":
            inc = st["direction"] * st["increment"]
            if cur + inc >= clip[1] or cur + inc <= clip[0]:
                st["direction"] = -st["direction"]
        else:
            raise ValueError(mode)
        new = cur + inc
        if clip:
            new = np.clip(new, *clip)
        return new

    return update


class MeasurementChain:
    """add_noise_and_latency (CartPole/__init__.py:336-356): what the simulator hands its controller instead of the true state.
    Every simulation step: the state joins a latency buffer (float64, zero-initialised with cos = 1: CartPole/latency_adder.py:24-26);
    the delayed state is interpolated between the entries latency / dt and one more steps back (:63-67); measurement noise
    (CartPole/noise_adder.py:71-82: angle, then cos / sin, position, angleD, positionD - four float32 draws per step from the
    instance's generator, each times its sigma in float32, added in float64); the vertical angle offset is updated with the time
    after the step and added (:348-356).  `for_controller(informed)`: informed, the controller gets the offset taken out again
    (:501-505)."""

    def __init__(self, latency, dt_sampling, noise=None, offset_updater=None, offset_init_deg=0.0):
        self.len = latency / dt_sampling                                                        # latency_adder.py:74-80
        self.li, self.frac = int(self.len), self.len - int(self.len)
        self.hist = []                                                                          # states appended so far (step 1, 2, ...)
        self.noise = noise                                                                      # (generator, sigma_angle, sigma_position, sigma_angleD, sigma_positionD) or None
        self.update_offset = offset_updater
        self.offset = np.deg2rad(offset_init_deg)                                               # :142-143
        self.s = None

    def _past(self, k):
        """The state k appends before the latest; before anything was appended: the buffer's initial content."""
        i = len(self.hist) - 1 - k
        if i < 0:
            z = np.zeros(6)
            z[O.ANGLE_COS_IDX] = 1.0
            return z
        return self.hist[i]

    def step(self, s, time_now):
        self.hist.append(np.array(s, dtype=np.float64))
        s1, s2 = self._past(self.li), self._past(self.li + 1)
        m = s1 + self.frac * (s2 - s1)
        if self.noise is not None:
            gen, sa, sp, sad, spd = self.noise
            m[O.ANGLE_IDX] += sa * gen.standard_normal(dtype=f32)
            m[O.ANGLE_IDX] = wrap_angle_rad(m[O.ANGLE_IDX])
            m[O.ANGLE_COS_IDX], m[O.ANGLE_SIN_IDX] = np.cos(m[O.ANGLE_IDX]), np.sin(m[O.ANGLE_IDX])
            m[O.POSITION_IDX] += sp * gen.standard_normal(dtype=f32)
            m[O.ANGLED_IDX] += sad * gen.standard_normal(dtype=f32)
            m[O.POSITIOND_IDX] += spd * gen.standard_normal(dtype=f32)
        if self.update_offset is not None:
            self.offset = self.update_offset(self.offset, time_now)
        m[O.ANGLE_IDX] = wrap_angle_rad(m[O.ANGLE_IDX] + self.offset)
        m[O.ANGLE_COS_IDX], m[O.ANGLE_SIN_IDX] = np.cos(m[O.ANGLE_IDX]), np.sin(m[O.ANGLE_IDX])
        self.s = m

    def for_controller(self, informed):
        if not informed:
            return self.s
        c = self.s.copy()
        c[O.ANGLE_IDX] = wrap_angle_rad(c[O.ANGLE_IDX] - self.offset)
        c[O.ANGLE_COS_IDX], c[O.ANGLE_SIN_IDX] = np.cos(c[O.ANGLE_IDX]), np.sin(c[O.ANGLE_IDX])
        return c


def wrap_angle_rad(angle):
    """CartPole/_CartPole_mathematical_helpers.py:13-21 (math.fmod on a float64)."""
    import math
    m = math.fmod(angle, 2 * np.pi)
    if m < -np.pi:
        return m + 2 * np.pi
    if m > np.pi:
        return m - 2 * np.pi
    return m


def run_experiment(setup, config, controller_step, L=None, p=O.DEFAULT_PARAMS, L_steps=None, m_pole_steps=None, informer=None,
                   disturbance=None, sensor=None):
    """One experiment as CartPole.run_cartpole_random_experiment runs it (noise, latency, disturbance OFF as shipped).
    controller_step(s, time, target_position, target_equilibrium, L) -> Q.  Returns dict(rows: column -> list, calls).
    ``L_steps`` [n + 1]: a pole length that changes in time - entry g is what the simulator holds DURING simulation step g
    (update_parameters is the first thing update_state does, CartPole/__init__.py:285, 529-537); entry 0 the initial value.
    ``m_pole_steps``: the same for the pole mass (the plant's; the controller is TOLD, 'm_pole' of updated_attributes - `calls`
    records it).  ``informer``: controller_informer(...) or None = 'ON'.
    ``disturbance`` = (z, controlDisturbance, controlBias): the simulator's additive control disturbance (CartPole/
    noise_control_signal.py:14-16 at every controller update, CartPole/__init__.py:523-524, 881-882): Q_applied = Q_calculated +
    controlDisturbance * z[call] + controlBias in float32 (0-d float32 arrays and a float32 draw: the Python float Q_calculated is the
    weak operand); the plant, `u` and the next call's Q_ccrc use Q_applied.
    ``sensor``: a MeasurementChain - the in-loop controller calls then see the measured state (the t = 0 call sees the true one,
    :869-870); `calls` records what was handed over (float64), the rows gain `vertical_angle_offset`."""
    c = config
    dt_sim = c["dt"]["simulation"]
    n_ctrl = max(1, int(np.rint(c["dt"]["control"] / dt_sim)))                                  # :909-916
    n_save = max(1, int(np.rint(c["dt"]["saving"] / dt_sim)))                                   # :925-933
    Lf = float(p.L if L is None else L) if L_steps is None else float(L_steps[0])
    inf = lambda v: np.inf if isinstance(v, str) and v == "inf" else v                          # noqa: E731
    times, tp_g, te_g = schedule_tables(setup["trace"], setup["target_equilibrium"], c["length_of_experiment"], dt_sim,
                                        inf(c["keep_target_equilibrium_x_seconds_up"]), inf(c["keep_target_equilibrium_x_seconds_down"]))
    s = np.array(setup["s0"], dtype=f32)
    calls = []
    L_init = Lf
    m_init = float(p.m_pole) if m_pole_steps is None else float(m_pole_steps[0])
    mf = m_init
    if m_pole_steps is not None:
        p = replace(p, m_pole=f32(mf))
    informed = [True]

    def control(g):
        informed[0] = True if informer is None else bool(informer(times[g]))                    # :495-500 (get_parameters, twice: idempotent)
        L_c, m_c = (Lf, mf) if informed[0] else (L_init, m_init)
        s_c = s.copy() if (sensor is None or g == 0) else np.array(sensor.for_controller(informed[0]))   # :501-507
        Q = controller_step(s_c.copy(), times[g], tp_g[g], te_g[g], L_c)
        calls.append(dict(s=s_c.copy(), time=times[g], tp=tp_g[g], te=te_g[g], Q=f32(Q), L=L_c, m_pole=m_c))
        return f32(Q)

    rows = {k: [] for k in ("time", "s", "angleDD", "positionDD", "Q", "Q_ccrc", "u", "target_position", "target_equilibrium", "L",
                            "m_pole", "informed", "Q_calculated", "vertical_angle_offset")}
    n_call = [0]

    def applied(Qc):
        k = n_call[0]
        n_call[0] += 1
        if disturbance is None:
            return Qc
        z, mult, bias = disturbance
        return f32(f32(f32(Qc) + f32(f32(mult) * f32(z[k]))) + f32(bias))

    Q_ccrc = f32(0.0)                                                                           # :838
    Q_calc = control(0)                                                                         # set_cartpole_state_at_t0 :842-852
    Q = applied(Q_calc)                                                                         # :881-882
    aDD, xDD = O.plant_ode(s, Q, Lf, p)                                                         # :859-860

    def save(g):
        rows["time"].append(times[g]); rows["s"].append(s.copy()); rows["angleDD"].append(aDD); rows["positionDD"].append(xDD)
        rows["Q"].append(Q); rows["Q_ccrc"].append(Q_ccrc); rows["u"].append(O.Q2u(Q, p))
        rows["target_position"].append(tp_g[g]); rows["target_equilibrium"].append(te_g[g]); rows["L"].append(Lf)
        rows["m_pole"].append(mf); rows["informed"].append(informed[0]); rows["Q_calculated"].append(Q_calc)
        rows["vertical_angle_offset"].append(0.0 if sensor is None else float(sensor.offset))

    save(0)                                                                                     # :875 (the t = 0 row)
    ctrl_counter = save_counter = 0
    for g in range(1, len(times)):                                                              # update_state, :283-324
        if L_steps is not None:
            Lf = float(L_steps[g])                                                              # update_parameters
        if m_pole_steps is not None and float(m_pole_steps[g]) != mf:
            mf = float(m_pole_steps[g])
            p = replace(p, m_pole=f32(mf))
        s = O.plant_substep(s, aDD, xDD, dt_sim, Lf, p)                                         # integration, bounce, cos/sin, wrap
        if sensor is not None:
            sensor.step(s, times[g])                                                            # add_noise_and_latency :310
        ctrl_counter += 1
        if ctrl_counter == n_ctrl:                                                              # Update_Q :475-527
            Q_ccrc = Q
            Q_calc = control(g)
            Q = applied(Q_calc)                                                                 # :523-524
            ctrl_counter = 0
        aDD, xDD = O.plant_ode(s, Q, Lf, p)                                                     # :319-320
        save_counter += 1
        if save_counter == n_save:                                                              # save_csv_routine :403-433
            save(g)
            save_counter = 0
    return dict(rows={k: np.array(v) for k, v in rows.items()}, calls=calls, n_ctrl=n_ctrl, n_save=n_save, times=times,
                target_position=tp_g, target_equilibrium=te_g)
