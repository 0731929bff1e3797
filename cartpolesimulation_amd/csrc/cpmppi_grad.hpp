// Reverse-mode derivative of the rollout + plugin cost with respect to the input sequence (device code).
//
// What it serves: the gradient-based optimizers of the absent Control_Toolkit that share MPPI's predictor and
// cost-function seams (Control_Toolkit_ASF/config_optimizers.yml:49-86: gradient-tf, rpgd; :21-48 the CEM+gradient
// hybrids) — there TensorFlow differentiates get_trajectory_cost(predict_core(s, Q), Q) w.r.t. Q; here the adjoint of
// the same arithmetic is written out by hand:
//   * forward = the FAST substep with the reference's per-substep wrap + sin/cos (substep_fast<float>; formulas of
//     cartpole_equations.py:71-99,130-131,341-347, cartpole_numba.py:55-78), states check-pointed per control step;
//   * backward = per control step, recompute the S substeps (sub-states parked in LDS), then sweep them in reverse.
// Non-smooth pieces are differentiated the way automatic differentiation of the reference's code does it: the branch
// taken (edge bounce), derivative 1 through fmod / the wrap comparisons, 0 through indicator functions and through a
// clipped control.
#pragma once
#include "cpmppi_device.hpp"

namespace cpmppi {

struct Adjoint {
  float th, w, x, v;          // dJ/d(angle, angleD, position, positionD); angle_cos / angle_sin are functions of angle
};

// Forward substep of predictor_ODE (Euler-Cromer, no bounce) with the exact wrap + polynomial sincos on EVERY substep - the
// plain form the adjoint differentiates (the rollout kernel's control_step_cromer_fast advances (cos, sin) by rotation inside a
// control step: the same function of the state to float32 rounding).
__device__ __forceinline__ void substep_cromer_plain(State<float>& st, float uK, float t, const Params& p, const EnvConst& e) {
  float th1, w1, x1, v1, aDD;
  ode_cromer_fast<float>(st, uK, t, p, e, th1, w1, x1, v1, aDD);
  th1 = wrap_rint<float>(th1);
  st.th = th1; st.w = w1; st.x = x1; st.v = v1;
  sincos_pi_half<float>(th1, st.s, st.c);
}

// Reverse of one FAST substep.  `st` = state at the START of the substep, `uK` = (k+1) u; `lam` enters as the adjoint of
// the substep's END state and leaves as the adjoint of its START state; `guK` accumulates dJ/d(uK).
// CROMER: predictor_ODE's substep - w1 = w + aDD t, v1 = v + xDD t, th1 = th + w1 t, x1 = x + v1 t (cartpole_equations.py:
// 293-304), no bounce, derivative 1 through atan2(sin, cos): the adjoints of w1 / v1 first collect t times those of th1 / x1,
// and the angle and the position no longer feed the OLD velocities forward.
template <bool CROMER = false>
__device__ __forceinline__ void substep_reverse(const State<float>& st, float uK, float t, const Params& p,
                                                const EnvConst& e, Adjoint& lam, float& guK) {
  const float c = st.c, s = st.s, w = st.w, v = st.v;
  // ---- recompute the forward quantities the derivatives need (same grouping as ode_euler_fast)
  const float A = __builtin_fmaf(-(c * p.m_pole), c, e.kp1_mt);
  const float r = 1.0f / A;
  const float t1 = __builtin_fmaf(e.mg, s, -(w * e.JinvLh));
  float num = __builtin_fmaf(c, t1, uK);
  num = __builtin_fmaf(-((w * w) * e.kmLh), s, num);
  num = __builtin_fmaf(-e.kM, v, num);
  const float xDD = num * r;
  const float aDD = __builtin_fmaf(e.g_i, s, __builtin_fmaf(xDD * c, e.inv_kLh, -(w * e.cT_i)));
  const float th1 = __builtin_fmaf(w, t, st.th);
  const float w1 = __builtin_fmaf(aDD, t, w);
  const float x1 = __builtin_fmaf(v, t, st.x);
  const float v1 = __builtin_fmaf(xDD, t, v);
  (void)w1;

  float lth = lam.th, lw = lam.w, lx = lam.x, lv = lam.v;
  if constexpr (CROMER) {
    lw = __builtin_fmaf(lth, t, lw);
    lv = __builtin_fmaf(lx, t, lv);
  }
  // ---- edge bounce (taken branch): w2 = w1 - 2 v1 cos(th1) / (L/2); th2 = th1 + w2 t; v2 = -v1; x2 = x1 + v2 t
  if (!CROMER && __builtin_fabsf(x1) >= p.THL) {
    float sb, cb;
    sincosf(th1, &sb, &cb);
    const float lw2 = __builtin_fmaf(lth, t, lw);
    const float lv2 = __builtin_fmaf(lx, t, lv);
    lth = __builtin_fmaf(lw2, 2.0f * v1 * sb * e.inv_halfL, lth);
    lv = -lv2 - lw2 * (2.0f * cb * e.inv_halfL);
    lw = lw2;
  }
  // ---- simultaneous forward Euler + ODE
  const float dA_dth = 2.0f * p.m_pole * c * s;                       // A = K - m c^2
  const float dnum_dth = __builtin_fmaf(-s, t1, c * (e.mg * c)) - e.kmLh * (w * w) * c;
  const float dx_dth = (dnum_dth - xDD * dA_dth) * r;
  const float dx_dw = -(__builtin_fmaf(c, e.JinvLh, 2.0f * e.kmLh * w * s)) * r;
  const float dx_dv = -e.kM * r;
  const float dx_du = r;
  const float ic = e.inv_kLh * c;
  const float da_dth = __builtin_fmaf(e.g_i, c, e.inv_kLh * __builtin_fmaf(dx_dth, c, -(xDD * s)));
  const float da_dw = __builtin_fmaf(ic, dx_dw, -e.cT_i);
  const float da_dv = ic * dx_dv;
  const float da_du = ic * dx_du;
  lam.th = __builtin_fmaf(t, __builtin_fmaf(lw, da_dth, lv * dx_dth), lth);
  lam.w = __builtin_fmaf(t, (CROMER ? 0.0f : lth) + __builtin_fmaf(lw, da_dw, lv * dx_dw), lw);
  lam.x = lx;
  lam.v = __builtin_fmaf(t, (CROMER ? 0.0f : lx) + __builtin_fmaf(lw, da_dv, lv * dx_dv), lv);
  guK = __builtin_fmaf(t, __builtin_fmaf(lw, da_du, lv * dx_du), guK);
}

// Partial derivatives of one stage cost: d/dx, d/dcos(angle), d/d(angleD), d/du, d/d(u_before).
struct StageGrad {
  float x, cosang, w, u, u_before;
};

__device__ __forceinline__ float sign_(float a) { return (a > 0.0f) ? 1.0f : ((a < 0.0f) ? -1.0f : 0.0f); }

// quadratic_boundary_grad_minimal.py:64-126 (weights as in stage_qbgm)
__device__ __forceinline__ StageGrad stage_qbgm_grad(const Params& p, float x, float cosang, float w_ang, float u,
                                                     float x_t, float te) {
  const float THL = p.THL, ptf = p.w[6];
  StageGrad g{};
  const float d = (x - x_t) / (2.0f * THL);
  g.x = 2.0f * p.w[0] * d / (2.0f * THL);
  const float ax = __builtin_fabsf(x);
  if (ax > ptf * THL) {
    const float den = (1.0f - ptf) * THL;
    g.x += 2.0f * p.w[1] * ((ax - ptf * THL) / den) * sign_(x) / den;
  }
  g.cosang = -2.0f * p.w[2] * (1.0f - cosang * te) * te;
  g.w = 2.0f * p.w[3] * w_ang;
  g.u = 2.0f * p.w[5] * p.w[4] * u;
  return g;
}

// default.py:23-88 (weights as in stage_default); the two indicator terms have zero derivative
__device__ __forceinline__ StageGrad stage_default_grad(const Params& p, float x, float cosang, float u, float x_t,
                                                        float te) {
  const float THL = p.THL;
  StageGrad g{};
  const float d = (x - x_t) / (2.0f * THL);
  g.x = 2.0f * p.w[0] * d / (2.0f * THL);
  g.cosang = -2.0f * (1.0f - cosang) * 0.25f * te * p.w[1];
  g.u = 2.0f * p.w[3] * p.w[2] * u;
  return g;
}

// quadratic_boundary_grad.py:64-232 (weights as in stage_qbg)
__device__ __forceinline__ StageGrad stage_qbg_grad(const Params& p, float x, float cosang, float w_ang, float u,
                                                    float u_before, float x_t, float te) {
  const bool up = (te == 1.0f);
  const float* w = p.w + (up ? 0 : 7);
  const float corr = up ? p.w[14] : p.w[15];
  const float ptf = p.w[16], cos_adm = p.w[17], R = p.w[18], THL = p.THL;
  StageGrad g{};
  const float d = (x - x_t) / (2.0f * THL);
  g.x = (2.0f * w[0] * d + w[1] * sign_(d)) / (2.0f * THL);
  const float ax = __builtin_fabsf(x);
  if (ax > ptf * THL) {
    const float den = (1.0f - ptf) * THL;
    g.x += 2.0f * w[2] * ((ax - ptf * THL) / den) * sign_(x) / den;
  }
  const float tc = cosang * te;
  g.cosang = -2.0f * (2.0f - tc) * te * w[3];
  const float tas_max = __builtin_fabsf(120.0f * (1.0f + te) / 2.0f + corr);
  const bool inside = te * (cosang - cos_adm) > 0.0f;
  const float scaling = inside ? 0.0f : (1.0f - tc) * 0.5f;
  const float sg = sign_(w_ang * w_ang - scaling * tas_max);
  g.w = sg * 2.0f * w_ang * w[4];
  if (!inside) g.cosang += sg * tas_max * (0.5f * te) * w[4];
  g.u = 2.0f * R * w[5] * u;
  const float dc = u - u_before;
  g.u += 2.0f * w[6] * dc;
  g.u_before = -2.0f * w[6] * dc;
  return g;
}

}  // namespace cpmppi
