// cpmppi_device.hpp — device-side building blocks of the MPPI rollout path (gfx950 only).
//
// Arithmetic spec: SURVEY.md Appendix A, restating (reference checkout paths)
//   CartPole/cartpole_equations.py:44-105   _cartpole_ode
//   CartPole/cartpole_equations.py:356-364  cartpole_integration_numba (simultaneous forward Euler)
//   CartPole/cartpole_equations.py:341-347  edge_bounce
//   CartPole/_CartPole_mathematical_helpers.py:24-29  wrap_angle_rad_inplace
//   CartPole/cartpole_numba.py:55-78        cartpole_fine_integration_numba (the substep loop)
// Everything is float32.  One lane integrates one rollout; all per-env quantities are wave-uniform and end up in
// SGPRs (they are derived from kernel arguments and blockIdx only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cpmppi {

constexpr float PI_F = 3.14159274101257324f;       // float32(np.pi)
constexpr float TWO_PI_F = 6.28318548202514648f;   // float32(2*np.pi)

enum : int { COST_QBGM = 0, COST_DEFAULT = 1, COST_LEGACY = 2 };
enum : int { NOISE_DELTA_U = 0, NOISE_KNOTS = 1, NOISE_PHILOX = 2 };

// Kernel-argument block (passed by value -> kernarg segment -> SGPRs).
struct Params {
  uint32_t E, N, H, S, P;            // P = number of knots = ceil(H/period)+1
  uint32_t period;
  float t_step;                      // dt / S
  float k, m_cart, m_pole, g, J_fric, M_fric, u_max, THL, L_default;
  uint32_t cost_id;
  float w[16];
  float R, LBD, NU, cc_weight, sigma;
  float lo, hi;
  uint32_t horizon_reduce, control_mode, shift_mode, correction_u;
};

// Per-env constants.  PRECISE keeps the reference's operands; FAST folds them (all wave-uniform).
struct EnvConst {
  float L, Lh;
  float kp1, kp1_mt;                 // (k+1), (k+1)*(m_cart+m_pole)
  float mg, JinvLh, kmLh, kM, g_i, cT_i, inv_kLh, inv_halfL;
};

__device__ __forceinline__ EnvConst make_env_const(const Params& p, float L) {
  EnvConst c;
  c.L = L;
  c.Lh = L / 2.0f;
  c.kp1 = p.k + 1.0f;
  c.kp1_mt = c.kp1 * (p.m_cart + p.m_pole);
  // folded constants are formed in double and rounded once
  const double Lh = (double)c.Lh, kp1 = (double)c.kp1;
  c.mg = (float)((double)p.m_pole * (double)p.g);
  c.JinvLh = (float)((double)p.J_fric / Lh);
  c.kmLh = (float)(kp1 * (double)p.m_pole * Lh);
  c.kM = (float)(kp1 * (double)p.M_fric);
  const double inv_kLh = 1.0 / (kp1 * Lh);
  c.inv_kLh = (float)inv_kLh;
  c.g_i = (float)((double)p.g * inv_kLh);
  c.cT_i = (float)((double)p.J_fric / ((double)p.m_pole * Lh) * inv_kLh);
  c.inv_halfL = (float)(1.0 / (0.5 * (double)L));
  return c;
}

// ------------------------------------------------------------------------------------------------------------------
// sincos on [-pi_f32, pi_f32] (the angle is wrapped every substep, so the argument never leaves this range).
// Cody-Waite reduction to |r| <= pi/4 with q in {-2..2} (q*PIO2_HI is exact), then the classic single-precision
// minimax polynomials; <= 1.5 ulp for both outputs over the range.
__device__ __forceinline__ void sincos_pi(float x, float& sn, float& cs) {
  constexpr float TWO_OVER_PI = 0.636619746685028076f;
  constexpr float PIO2_HI = 1.57079637050628662f;
  constexpr float PIO2_LO = -4.37113900018624283e-8f;
  const float q = __builtin_rintf(x * TWO_OVER_PI);
  float r = __builtin_fmaf(-q, PIO2_HI, x);
  r = __builtin_fmaf(-q, PIO2_LO, r);
  const float r2 = r * r;
  float ps = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, r2, -1.6666654611e-1f);
  const float S = __builtin_fmaf(ps * r2, r, r);
  float pc = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, r2, 4.166664568298827e-2f);
  const float C = __builtin_fmaf(pc * r2, r2, __builtin_fmaf(-0.5f, r2, 1.0f));
  const int n = (int)q;
  const bool swap = (n & 1) != 0;
  const float s0 = swap ? C : S;
  const float c0 = swap ? S : C;
  const uint32_t sflip = ((uint32_t)n & 2u) << 30;
  const uint32_t cflip = ((uint32_t)(n + 1) & 2u) << 30;
  sn = __uint_as_float(__float_as_uint(s0) ^ sflip);
  cs = __uint_as_float(__float_as_uint(c0) ^ cflip);
}

struct State {
  float th, w, c, s, x, v;   // angle, angleD, angle_cos, angle_sin, position, positionD
};

// One Euler substep.  FAST=false follows the reference's operand grouping with IEEE divides, libm sincos and no FMA
// contraction; FAST=true evaluates the same float32 formulas with folded per-env constants, FMA, a
// reciprocal+Newton divide and sincos_pi.
template <bool FAST>
__device__ __forceinline__ void substep(State& st, float u, float uK, float t, const Params& p, const EnvConst& e);

// _cartpole_ode with the reference's operand grouping (cartpole_equations.py:71-99), IEEE divides, no contraction.
__device__ __forceinline__ void ode_precise(float c, float s, float w, float v, float u, const Params& p,
                                            const EnvConst& e, float& aDD, float& xDD) {
#pragma clang fp contract(off)
  const float A = e.kp1_mt - p.m_pole * (c * c);
  const float F = -p.M_fric * v;
  const float T = -p.J_fric * w;
  const float Lh = e.Lh;
  xDD = (p.m_pole * p.g * s * c + ((T * c) / Lh) + e.kp1 * (-(p.m_pole * Lh * (w * w) * s) + F + u)) / A;
  aDD = (p.g * s + xDD * c + T / (p.m_pole * Lh)) / (e.kp1 * Lh);
}

template <>
__device__ __forceinline__ void substep<false>(State& st, float u, float /*uK*/, float t, const Params& p,
                                               const EnvConst& e) {
#pragma clang fp contract(off)
  const float w = st.w, v = st.v;
  float aDD, xDD;
  ode_precise(st.c, st.s, w, v, u, p, e, aDD, xDD);
  float th1 = st.th + w * t;
  float w1 = w + aDD * t;
  float x1 = st.x + v * t;
  float v1 = v + xDD * t;
  if (x1 >= p.THL || -x1 >= p.THL) {
    const float cb = cosf(th1);
    w1 = w1 - 2.0f * (v1 * cb) / (0.5f * e.L);
    th1 = th1 + w1 * t;
    v1 = -v1;
    x1 = x1 + v1 * t;
  }
  const float m = fmodf(th1, TWO_PI_F);
  th1 = (m < -PI_F) ? (m + TWO_PI_F) : ((m > PI_F) ? (m - TWO_PI_F) : m);
  st.th = th1; st.w = w1; st.x = x1; st.v = v1;
  st.c = cosf(th1);
  st.s = sinf(th1);
}

// One simulation step of the PLANT (the caller side of the boundary): Euler-Cromer (cartpole_equations.py:367-378),
// edge bounce with cos of the integrated angle (CartPole/__init__.py:462-470), cos/sin (:329-331), wrap (:333-334).
__device__ __forceinline__ void plant_substep(State& st, float aDD, float xDD, float t, const Params& p,
                                              const EnvConst& e) {
#pragma clang fp contract(off)
  float w1 = st.w + aDD * t;
  float v1 = st.v + xDD * t;
  float th1 = st.th + w1 * t;
  float x1 = st.x + v1 * t;
  if (x1 >= p.THL || -x1 >= p.THL) {
    const float cb = cosf(th1);
    w1 = w1 - 2.0f * (v1 * cb) / (0.5f * e.L);
    th1 = th1 + w1 * t;
    v1 = -v1;
    x1 = x1 + v1 * t;
  }
  st.c = cosf(th1);
  st.s = sinf(th1);
  const float m = fmodf(th1, TWO_PI_F);
  st.th = (m < -PI_F) ? (m + TWO_PI_F) : ((m > PI_F) ? (m - TWO_PI_F) : m);
  st.w = w1; st.x = x1; st.v = v1;
}

#ifndef CPMPPI_RARE_MODE
#define CPMPPI_RARE_MODE 1     // 0: per-lane divergent branches (compiler exec masking)  1: wave-uniform test, inline cold path
#endif

template <>
__device__ __forceinline__ void substep<true>(State& st, float /*u*/, float uK, float t, const Params& p,
                                              const EnvConst& e) {
  const float c = st.c, s = st.s, w = st.w, v = st.v;
  const float A = __builtin_fmaf(-(p.m_pole * c), c, e.kp1_mt);
  const float t1 = __builtin_fmaf(e.mg, s, -(e.JinvLh * w));
  float num = __builtin_fmaf(c, t1, uK);
  num = __builtin_fmaf(-(e.kmLh * (w * w)), s, num);
  num = __builtin_fmaf(-e.kM, v, num);
  const float r = __builtin_amdgcn_rcpf(A);          // A in [0.33, 0.43]: no scaling needed
  const float q0 = num * r;
  const float xDD = __builtin_fmaf(__builtin_fmaf(-A, q0, num), r, q0);
  const float aDD = __builtin_fmaf(e.g_i, s, __builtin_fmaf(xDD * c, e.inv_kLh, -(e.cT_i * w)));
  float th1 = __builtin_fmaf(w, t, st.th);
  float w1 = __builtin_fmaf(aDD, t, w);
  float x1 = __builtin_fmaf(v, t, st.x);
  float v1 = __builtin_fmaf(xDD, t, v);
#if CPMPPI_RARE_MODE == 0
  if (__builtin_expect(__builtin_fabsf(x1) >= p.THL, 0)) {
    const float cb = cosf(th1);
    w1 = __builtin_fmaf(-(2.0f * (v1 * cb)), e.inv_halfL, w1);
    th1 = __builtin_fmaf(w1, t, th1);
    v1 = -v1;
    x1 = __builtin_fmaf(v1, t, x1);
  }
  if (__builtin_expect(__builtin_fabsf(th1) >= TWO_PI_F, 0)) th1 = fmodf(th1, TWO_PI_F);
#else
  // One wave-uniform test for both rare events (edge bounce, cartpole_equations.py:341-347; an angle beyond one
  // 2*pi wrap, |angleD| > 1500 rad/s): the hot instruction stream carries no exec-mask bookkeeping.
  const bool rare = (__builtin_fabsf(x1) >= p.THL) | (__builtin_fabsf(th1) >= TWO_PI_F);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare) != 0, 0)) {
    if (__builtin_fabsf(x1) >= p.THL) {
      const float cb = cosf(th1);
      w1 = __builtin_fmaf(-(2.0f * (v1 * cb)), e.inv_halfL, w1);
      th1 = __builtin_fmaf(w1, t, th1);
      v1 = -v1;
      x1 = __builtin_fmaf(v1, t, x1);
    }
    if (__builtin_fabsf(th1) >= TWO_PI_F) th1 = fmodf(th1, TWO_PI_F);
  }
#endif
  // fmod(theta, 2pi) is the identity for |theta| < 2pi; then the reference's two comparisons
  const float off = (th1 > PI_F) ? -TWO_PI_F : ((th1 < -PI_F) ? TWO_PI_F : 0.0f);
  th1 += off;
  st.th = th1; st.w = w1; st.x = x1; st.v = v1;
  sincos_pi(th1, st.s, st.c);
}

// ------------------------------------------------------------------------------------------------------------------
// Stage / terminal costs.  `x_t` target position, `te` target equilibrium, `u` the control applied at this stage.
// quadratic_boundary_grad_minimal.py:64-126; w = {dd, db, ep, ekp, cc, R, permissible_track_fraction}
__device__ __forceinline__ float stage_qbgm(const Params& p, float x, float cosang, float w_ang, float u, float x_t,
                                            float te) {
  const float THL = p.THL;
  const float d = (x - x_t) / (2.0f * THL);
  const float dd = p.w[0] * (d * d);
  const float ptf = p.w[6];
  const float ax = __builtin_fabsf(x);
  const float near = (ax > ptf * THL) ? 1.0f : 0.0f;
  const float b = (ax - ptf * THL) / ((1.0f - ptf) * THL);
  const float db = p.w[1] * (near * (b * b));
  const float e1 = 1.0f - te * cosang;
  const float ep = p.w[2] * (e1 * e1);
  const float ekp = p.w[3] * (w_ang * w_ang);
  const float cc = p.w[4] * (p.w[5] * (u * u));
  return dd + db + ep + ekp + cc;
}

// default.py:23-88; w = {dd, ep, cc, R}
__device__ __forceinline__ float stage_default(const Params& p, float x, float cosang, float u, float x_t, float te) {
  const float THL = p.THL;
  const float d = (x - x_t) / (2.0f * THL);
  const float ind = (__builtin_fabsf(x) > 0.90f * THL) ? 1.0e7f : 0.0f;
  const float dd = p.w[0] * (d * d + ind);
  const float e1 = 1.0f - cosang;
  const float ep = p.w[1] * (te * 0.25f * (e1 * e1));
  const float cc = p.w[2] * (p.w[3] * (u * u));
  return dd + ep + cc;
}

// default.py:55-63 and controller_mppi_cartpole.py:278-303 (phi): 10000 * 1[|angle| > 0.2 or |x - x*| > 0.1*THL]
__device__ __forceinline__ float terminal_indicator(const Params& p, float angle, float x, float x_t) {
  const bool bad = (__builtin_fabsf(angle) > 0.2f) || (__builtin_fabsf(x - x_t) > 0.1f * p.THL);
  return bad ? 10000.0f : 0.0f;
}

// MPPI correction term, algebra of controller_mppi_cartpole.py:261-263
__device__ __forceinline__ float mppi_correction(const Params& p, float u, float du) {
  return p.cc_weight * (0.5f * (1.0f - 1.0f / p.NU) * p.R * (du * du) + p.R * u * du + 0.5f * p.R * (u * u));
}

// controller_mppi_cartpole.py:227-275 (q); w = {dd, ep, ekp, ekc, cc, ccrc}
__device__ __forceinline__ float stage_legacy(const Params& p, float x, float cosang, float w_ang, float v, float u,
                                              float du, float u_prev, float x_t) {
  const float THL = p.THL;
  const float d = (x - x_t) / (2.0f * THL);
  const float ind = (__builtin_fabsf(x) > 0.95f * THL) ? 1.0e6f : 0.0f;
  const float dd = p.w[0] * (d * d + ind);
  const float e1 = 1.0f - cosang;
  const float ep = p.w[1] * (0.25f * (e1 * e1));
  const float ekp = p.w[2] * (w_ang * w_ang);
  const float ekc = p.w[3] * (v * v);
  float cc = p.w[4] * (0.5f * (1.0f - 1.0f / p.NU) * p.R * (du * du) + p.R * u * du + 0.5f * p.R * (u * u));
  const float ur = u + du;
  if (__builtin_fabsf(ur) > 1.0f) cc = 1.0e5f;
  const float dcr = ur - u_prev;
  const float ccrc = p.w[5] * (dcr * dcr);
  return dd + ep + ekp + ekc + cc + ccrc;
}

// ------------------------------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al. 2011).  Counter = (rollout, env, knot pair, offset lo),
// key = seed.  One call yields the two standard normals of one Box-Muller pair (knots 2j and 2j+1).
__device__ __forceinline__ void philox4x32_10(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                              uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__device__ __forceinline__ void philox_normal_pair(uint64_t seed, uint64_t offset, uint32_t env, uint32_t rollout,
                                                   uint32_t pair, float& z0, float& z1) {
  uint32_t c0 = rollout, c1 = env, c2 = pair, c3 = (uint32_t)offset;
  philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32));
  const float u1 = (float)((c0 >> 8) + 1u) * 5.9604644775390625e-8f;   // (0, 1]
  const float u2 = (float)(c1 >> 8) * 5.9604644775390625e-8f;          // [0, 1)
  const float r = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(TWO_PI_F * u2, &sn, &cs);
  z0 = r * cs;
  z1 = r * sn;
}

// Knot j of (env, rollout) scaled by sigma: the float32 product of the float64 values, as the reference forms
// stdev * z in float64 before the float32 store (controller_mppi_cartpole.py:441-443).
__device__ __forceinline__ float philox_knot(uint64_t seed, uint64_t offset, uint32_t env, uint32_t rollout,
                                             uint32_t j, float sigma) {
  float z0, z1;
  philox_normal_pair(seed, offset, env, rollout, j >> 1, z0, z1);
  return sigma * ((j & 1u) ? z1 : z0);
}

// Linear interpolation between knots as scipy interp1d does it at controller_mppi_cartpole.py:444-445:
// slope = float64(float32(hi - lo)) / period ; y = slope * i + float64(lo) ; stored as float32.
__device__ __forceinline__ float interp_knots(float z_lo, float z_hi, uint32_t i, uint32_t period) {
#pragma clang fp contract(off)
  if (i == 0) return z_lo;
  const float diff = z_hi - z_lo;                         // float32 difference, as numpy forms it
  const double slope = (double)diff / (double)period;
  const double prod = slope * (double)i;                  // two roundings (no FMA), as numpy
  return (float)(prod + (double)z_lo);
}

// ------------------------------------------------------------------------------------------------------------------
// wave64 reductions
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace cpmppi
