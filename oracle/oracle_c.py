"""ctypes wrapper of oracle/liboracle.so (the plain-C CPU oracle).  TEST INFRASTRUCTURE — see cpmppi_oracle.c."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
f32 = np.float32


class oracle_config(C.Structure):
    _fields_ = [("N", C.c_uint32), ("H", C.c_uint32), ("S", C.c_uint32), ("period", C.c_uint32), ("dt", C.c_float),
                ("k", C.c_float), ("m_cart", C.c_float), ("m_pole", C.c_float), ("g", C.c_float), ("J_fric", C.c_float),
                ("M_fric", C.c_float), ("u_max", C.c_float), ("THL", C.c_float), ("cost_id", C.c_uint32),
                ("w", C.c_float * 16), ("R", C.c_float), ("LBD", C.c_float), ("NU", C.c_float), ("cc_weight", C.c_float),
                ("lo", C.c_float), ("hi", C.c_float), ("horizon_reduce", C.c_uint32), ("control_mode", C.c_uint32),
                ("shift_mode", C.c_uint32), ("correction_u", C.c_uint32), ("f64_substeps", C.c_uint32),
                ("integrator", C.c_uint32)]


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "cpmppi_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_max_threads.restype = C.c_int
        _lib.oracle_max_threads()            # (caches the thread count the process started with)
    return _lib


_W = {0: lambda c: [c.qbgm_dd_quadratic_weight, c.qbgm_db_weight, c.qbgm_ep_weight, c.qbgm_ekp_weight, c.qbgm_cc_weight,
                    c.qbgm_R, c.qbgm_permissible_track_fraction],
      1: lambda c: [c.def_dd_weight, c.def_ep_weight, c.def_cc_weight, c.def_R],
      2: lambda c: [c.leg_dd_weight, c.leg_ep_weight, c.leg_ekp_weight, c.leg_ekc_weight, c.leg_cc_weight,
                    c.leg_ccrc_weight]}


def make_config(cfg, p=None, mode="f32"):
    """cfg: oracle_np.MPPIConfig, p: oracle_np.CartPoleParams."""
    from . import oracle_np as O
    p = p or O.DEFAULT_PARAMS
    c = oracle_config()
    c.N, c.H, c.S, c.period, c.dt = cfg.N, cfg.H, cfg.S, cfg.period, cfg.dt
    c.k, c.m_cart, c.m_pole, c.g, c.J_fric, c.M_fric = p.k, p.m_cart, p.m_pole, p.g, p.J_fric, p.M_fric
    c.u_max, c.THL = p.u_max, p.TrackHalfLength
    c.cost_id = cfg.cost_id
    for i, v in enumerate(_W[cfg.cost_id](cfg.cost)):
        c.w[i] = v
    c.R, c.LBD, c.NU, c.cc_weight, c.lo, c.hi = cfg.R, cfg.LBD, cfg.NU, cfg.cc_weight, -1.0, 1.0
    c.horizon_reduce = {"sum": 0, "mean": 1}[cfg.horizon_reduce]
    c.control_mode = {"clip": 0, "penalise": 1}[cfg.control_mode]
    c.shift_mode = {"repeat_last": 0, "append_zero": 1, "none": 2}[cfg.shift_mode]
    c.correction_u = {"u_run": 0, "u_nom": 1}[cfg.correction_u]
    c.f64_substeps = {"f32": 0, "f64sub": 1}[mode]
    c.integrator = {"ODE_v0": 0, "ODE": 1}[getattr(cfg, "integrator", "ODE_v0")]
    return c


_fma = None


def fma_lib():
    """Mode C of the checker (oracle/Makefile liboracle_fma.so): float32 substeps with FMA contraction and libm float
    trig.  None when the host has no FMA or no compiler (callers then use modes A and B only)."""
    global _fma
    if _fma is None:
        path = os.path.join(HERE, "liboracle_fma.so")
        try:
            if " fma " not in open("/proc/cpuinfo").read():
                raise OSError("host CPU without FMA")
            if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(HERE, "cpmppi_oracle.c")):
                subprocess.check_call(["make", "-s", "-C", HERE, "liboracle_fma.so"])
            _fma = C.CDLL(path)
            _fma.oracle_max_threads.restype = C.c_int
            _fma.oracle_max_threads()
        except Exception:
            _fma = False
    return _fma or None


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def predict(cfg_c, s0, Q, L=None, L_default=0.395, n_threads=0, use_lib=None):
    Q = np.ascontiguousarray(Q, dtype=f32)
    B, H = Q.shape
    s0 = np.ascontiguousarray(np.broadcast_to(np.asarray(s0, dtype=f32), (B, 6)))
    L = None if L is None else np.ascontiguousarray(np.broadcast_to(np.asarray(L, dtype=f32), (B,)))
    traj = np.empty((B, H + 1, 6), dtype=f32)
    (use_lib or lib()).oracle_predict(C.byref(cfg_c), C.c_uint32(B), C.c_uint32(H), _p(s0), _p(Q), _p(L), C.c_float(L_default),
                         _p(traj), C.c_int(n_threads))
    return traj


def _bench_key():
    """What the timing builds depend on: the source, the Makefile (flags), the compiler and - they are -march=native - the host's CPU."""
    import hashlib
    h = hashlib.sha256()
    for f in ("cpmppi_oracle.c", "Makefile"):
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(fh.read())
    try:
        h.update(subprocess.check_output([os.environ.get("CC", "gcc"), "--version"]))
        with open("/proc/cpuinfo") as fh:
            h.update("".join(l for l in fh if l.startswith(("model name", "flags"))).encode())
    except (OSError, subprocess.CalledProcessError):
        pass
    return h.hexdigest()


def compile_bench_variants():
    """The compile step of build_bench_variants alone (bench.py starts it in a thread behind its timed regions, so that it runs
    under the verification leg instead of in front of the baseline)."""
    key, stamp = _bench_key(), os.path.join(HERE, "_bench", "KEY")
    libs = [os.path.join(HERE, "_bench", f"liboracle_{n}.so") for n in ("native", "native_fastmath")]
    fresh = os.path.exists(stamp) and open(stamp).read().strip() == key and all(os.path.exists(x) for x in libs)
    if not fresh:
        subprocess.check_call(["make", "-s", "-B", "-j2", "-C", HERE, "bench"])
        with open(stamp, "w") as f:
            f.write(key + "\n")


def build_bench_variants():
    """cpu_baseline timing builds (oracle/Makefile `bench`), compiled for the host this runs on (-march=native) - once per (source,
    flags, compiler, CPU): oracle/_bench/KEY remembers what the cached libraries were built from (the directory is git- and
    gpurun-ignored, so a fresh box always compiles).  Returns {name: (ctypes lib, compiler flags)}.  Timing only - never a checker."""
    compile_bench_variants()
    out = {}
    for name, flags in (("native", "-O3 -march=native -ffp-contract=off, libm float trig"),
                        ("native_fastmath", "-O3 -march=native -ffast-math, libm float trig")):
        v = C.CDLL(os.path.join(HERE, "_bench", f"liboracle_{name}.so"))
        v.oracle_max_threads.restype = C.c_int
        v.oracle_max_threads()
        out[name] = (v, flags)
    return out


def step(cfg_c, s0, u_nom, delta_u, x_t, te, L=None, u_prev=None, L_default=0.395, n_threads=0, want_S=True, use_lib=None):
    """E envs.  Returns (u_new[E,H], Q[E], S[E,N] or None); u_nom is not modified."""
    delta_u = np.ascontiguousarray(delta_u, dtype=f32)
    E, N, H = delta_u.shape
    assert (N, H) == (cfg_c.N, cfg_c.H)
    s0 = np.ascontiguousarray(s0, dtype=f32).reshape(E, 6)
    u = np.array(u_nom, dtype=f32).reshape(E, H).copy()
    x_t = np.ascontiguousarray(np.broadcast_to(np.asarray(x_t, dtype=f32), (E,)))
    te = np.ascontiguousarray(np.broadcast_to(np.asarray(te, dtype=f32), (E,)))
    L = None if L is None else np.ascontiguousarray(np.broadcast_to(np.asarray(L, dtype=f32), (E,)))
    u_prev = None if u_prev is None else np.ascontiguousarray(u_prev, dtype=f32).reshape(E, H)
    Q = np.empty(E, dtype=f32)
    S = np.empty((E, N), dtype=f32) if want_S else None
    (use_lib or lib()).oracle_step(C.byref(cfg_c), C.c_uint32(E), _p(s0), _p(u), _p(delta_u), _p(u_prev), _p(x_t), _p(te), _p(L),
                      C.c_float(L_default), _p(Q), _p(S), C.c_int(n_threads))
    return u, Q, S


def set_trig_jitter(seed, use_lib=None):
    """Probe switch of the C oracle (see cpmppi_oracle.c): sin / cos results moved to a neighbouring float32 at random; 0 = off."""
    (use_lib or lib()).oracle_set_trig_jitter(C.c_uint32(int(seed)))


def max_threads():
    return lib().oracle_max_threads()
