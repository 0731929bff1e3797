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


SAMPLING_TYPES = ("interpolated", "random_walk", "uniform", "repeated", "iid")


def sample_delta_u_sfc64(rng, E, N, H, stdev, sampling_type):
    """The legacy sampler's other modes (controller_mppi_cartpole.py:414-433,447-450) on numpy's stream, for runs on the
    reference's noise seeds: -> float32 [E, N, H] in the reference's rollout-major layout (cpmppi_step(noise_kind =
    CPMPPI_NOISE_DELTA_U)).  Draw order per env exactly as the reference's; where the reference hands on a float64
    product (repeated, iid) it is rounded to float32 here, the predictor's input type.
      random_walk  cumulative sum of Gaussian steps, accumulated in float32 (each step's product formed in float64)
      uniform      U(-1, 1) per horizon step (stdev is not used)
      repeated     one Gaussian perturbation per rollout, held over the horizon
      iid          independent Gaussians
    ("interpolated" is sample_knots_sfc64 + the device interpolation.)"""
    if sampling_type not in SAMPLING_TYPES or sampling_type == "interpolated":
        raise ValueError(f"sampling_type must be one of {SAMPLING_TYPES[1:]} here (got {sampling_type!r})")
    stdev = np.float64(stdev)
    out = np.empty((E, N, H), dtype=np.float32)
    for e in range(E):
        if sampling_type == "random_walk":
            out[e, :, 0] = stdev * rng.standard_normal(size=(N,), dtype=np.float32)
            for i in range(1, H):
                out[e, :, i] = out[e, :, i - 1] + stdev * rng.standard_normal(size=(N,), dtype=np.float32)
        elif sampling_type == "uniform":
            for i in range(H):
                out[e, :, i] = rng.uniform(low=-1.0, high=1.0, size=(N,)).astype(np.float32)
        elif sampling_type == "repeated":
            out[e] = np.tile(stdev * rng.standard_normal(size=(N, 1), dtype=np.float32), (1, H))
        else:
            out[e] = stdev * rng.standard_normal(size=(N, H), dtype=np.float32)
    return out
