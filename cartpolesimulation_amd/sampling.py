"""Host-side perturbation knots from numpy's SFC64 stream — for runs that must use the reference's noise seeds.

Follows the "interpolated" sampler of Control_Toolkit_ASF/Controllers/controller_mppi_cartpole.py:434-446: knots every
``period`` steps, ``stdev * standard_normal(float32)`` with the product formed in float64 (``stdev`` is a numpy
float64 there, :91) and stored as float32.  Interpolation to delta_u happens on the device (cpmppi_interpolate / the
rollout kernel), bit-identically to scipy's interp1d as the reference calls it.
"""
import numpy as np


def sample_knots_sfc64(rng, E, N, cfg):
    """-> float32 [E, N, P].  Draw order: env-major, then the reference's (N, P) block per env."""
    P = cfg.num_knots
    stdev = np.float64(cfg.SQRTRHOINV) * (1 / np.sqrt(cfg.mpc_timestep))
    out = np.empty((E, N, P), dtype=np.float32)
    for e in range(E):
        z = rng.standard_normal(size=(N, P), dtype=np.float32)
        out[e] = (stdev * z.astype(np.float64)).astype(np.float32)
    return out
