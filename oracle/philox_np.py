"""Philox4x32-10 and the device sampler's keying, restated in numpy - the pin of the library's own noise stream.

TEST INFRASTRUCTURE (only tests/, bench.py's verify leg and __graft_entry__.smoke may import it).  The reference draws its
perturbations from numpy's SFC64 on the host (Control_Toolkit_ASF/Controllers/controller_mppi_cartpole.py:351, :434-446); parity with
the reference is defined on IDENTICAL perturbations and tested with SFC64 knots fed from the host.  The in-kernel
generator exists for throughput only - but a bench line that says "Philox4x32-10" should be able to prove it: this file restates
  * the Philox4x32-10 bijection of Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC 2011), Random123
    1.x `philox4x32_R(10, ctr, key)`: ten rounds of (hi, lo) = M * c with M0 = 0xD2511F53 on c0 and M1 = 0xCD9E8D57 on c2, output
    (hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0), key bumped by (0x9E3779B9, 0xBB67AE85) between rounds - checked against the known-answer
    vectors Random123 ships (tests/test_oracle_philox.py);
  * the sampler's counter / key layout and its Box-Muller (cartpolesimulation_amd/csrc/cpmppi_device.hpp, philox_normal_quad /
    philox_normal_pair): counter = (rollout, GLOBAL env index, knot quad, low word of the step counter), key = (seed_lo ^ 0x51ed270b,
    seed_hi ^ high word of the step counter); words -> uniforms ua, uc in (0, 1] and ub, ud in [0, 1) with 24 bits each; knots
    4q .. 4q+3 = sigma * (ra cos 2 pi ub, ra sin 2 pi ub, rc cos 2 pi ud, rc sin 2 pi ud), r = sqrt(-2 ln u).
The device evaluates ln / sin / cos with the hardware's v_log_f32 / v_sin_f32 / v_cos_f32 (absolute error ~1e-6); here they are
evaluated in float64 and rounded once: the test bound on the normals is that hardware error, the uniforms behind them are exact.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
QUAD_KEY_XOR = 0x51ED270B
f32 = np.float32


def philox4x32_10(counter, key):
    """counter [..., 4], key [..., 2] (uint32, broadcastable) -> [..., 4] uint32."""
    c = np.asarray(counter, dtype=np.uint64) & MASK
    k = np.asarray(key, dtype=np.uint64) & MASK
    c0, c1, c2, c3 = (c[..., i] for i in range(4))
    k0, k1 = k[..., 0], k[..., 1]
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2                       # 32 x 32 -> 64 bit products (no overflow in uint64)
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + np.uint64(W0)) & MASK, (k1 + np.uint64(W1)) & MASK
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def _uniforms(words):
    """Four words -> (ua, ub, uc, ud): 24 bits each; a, c in (0, 1], b, d in [0, 1).  Exact in float32."""
    w = words.astype(np.uint64)
    s = 5.9604644775390625e-8                             # 2^-24
    return ((w[..., 0] >> np.uint64(8)) + np.uint64(1)) * s, (w[..., 1] >> np.uint64(8)) * s, \
           ((w[..., 2] >> np.uint64(8)) + np.uint64(1)) * s, (w[..., 3] >> np.uint64(8)) * s


def quad_words(seed, offset, env, rollout, quad):
    """The Philox block behind knots 4 quad .. 4 quad + 3 of (global env, rollout) at step `offset` (philox_normal_quad)."""
    seed, offset = int(seed), int(offset)
    env, rollout, quad = np.broadcast_arrays(np.asarray(env, np.uint64), np.asarray(rollout, np.uint64), np.asarray(quad, np.uint64))
    ctr = np.stack([rollout, env, quad, np.full(env.shape, offset & 0xFFFFFFFF, np.uint64)], axis=-1)
    key = np.array([(seed & 0xFFFFFFFF) ^ QUAD_KEY_XOR, ((seed >> 32) ^ (offset >> 32)) & 0xFFFFFFFF], np.uint64)
    return philox4x32_10(ctr, key)


def quad_uniforms(seed, offset, env, rollout, quad):
    return _uniforms(quad_words(seed, offset, env, rollout, quad))


def standard_normal_quads(seed, offset, env, rollout, quad):
    """-> [..., 4] float64: the four standard normals of a block, before any rounding."""
    ua, ub, uc, ud = quad_uniforms(seed, offset, env, rollout, quad)
    ra, rc = np.sqrt(-2.0 * np.log(ua)), np.sqrt(-2.0 * np.log(uc))
    return np.stack([ra * np.cos(2 * np.pi * ub), ra * np.sin(2 * np.pi * ub), rc * np.cos(2 * np.pi * ud), rc * np.sin(2 * np.pi * ud)], axis=-1)


def knots(seed, offset, env_offset, E, N, P, sigma):
    """What cpmppi_sample(seed, offset, env_offset) writes: knots[E, N, P] float32 = float32(sigma) * float32(z)."""
    nq = (P + 3) // 4
    e = (np.arange(E, dtype=np.uint64) + np.uint64(env_offset))[:, None, None]
    n = np.arange(N, dtype=np.uint64)[None, :, None]
    q = np.arange(nq, dtype=np.uint64)[None, None, :]
    z = standard_normal_quads(seed, offset, e, n, q).reshape(E, N, nq * 4)[:, :, :P]
    return (f32(sigma) * z.astype(f32)).astype(f32)
