/*
 * cpmppi_oracle.c — plain-C CPU restatement of the MPPI rollout hot path of SensorsINI/CartPoleSimulation.
 *
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py
 * load this library (oracle/liboracle.so); the product (libcpmppi.so) never links, calls or falls back to it.
 *
 * It is pinned two ways (tests/test_oracle_c.py): against the golden vectors under tests/golden/ (produced by the
 * reference's own in-tree code, oracle/gen_golden.py) and against the numpy restatement oracle/oracle_np.py.
 * float32 sin/cos are evaluated as (float)cos((double)x): correctly rounded for all practical purposes and
 * independent of the host's libm float kernels (numpy's SIMD float32 sin/cos differ from this by <= 1 ulp on a few
 * percent of the arguments, which is why rollout comparisons carry the H1 tolerance rather than bit equality).
 *
 * Reference citations (paths relative to the reference checkout):
 *   ode()            CartPole/cartpole_equations.py:44-105   (_cartpole_ode)
 *   substep()        CartPole/cartpole_numba.py:55-78        (cartpole_fine_integration_numba body)
 *                    CartPole/cartpole_equations.py:356-364  (simultaneous forward Euler)
 *                    CartPole/cartpole_equations.py:341-347  (edge_bounce), cartpole_numba.py:47-52
 *                    CartPole/_CartPole_mathematical_helpers.py:24-29 (wrap)
 *   control_step()   SI_Toolkit_ASF/ToolkitCustomization/predictors_customization_v0.py:41-55, cartpole_numba.py:10-41
 *   stage costs      Control_Toolkit_ASF/Cost_Functions/CartPole/quadratic_boundary_grad_minimal.py:64-126,
 *                    .../default.py:23-88, Control_Toolkit_ASF/Controllers/controller_mppi_cartpole.py:119-161,227-303
 *   soft-min update  controller_mppi_cartpole.py:306-321
 * Compile: see oracle/Makefile (-O2 -ffp-contract=off, no -ffast-math, OpenMP over rollouts).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  uint32_t N, H, S, period;
  float dt;
  float k, m_cart, m_pole, g, J_fric, M_fric, u_max, THL;
  uint32_t cost_id;            /* 0 qbgm, 1 default, 2 legacy */
  float w[16];                 /* same layout as cpmppi_config.cost_w */
  float R, LBD, NU, cc_weight;
  float lo, hi;
  uint32_t horizon_reduce;     /* 0 sum, 1 mean */
  uint32_t control_mode;       /* 0 clip, 1 penalise */
  uint32_t shift_mode;         /* 0 repeat last, 1 append zero, 2 none */
  uint32_t correction_u;       /* 0 u_run, 1 u_nom */
  uint32_t f64_substeps;       /* 0 = mode A (strict float32), 1 = mode B (float64 substeps, float32 store) */
  uint32_t integrator;         /* 0 = predictor_ODE_v0 (simultaneous Euler, edge bounce, fmod wrap), 1 = predictor_ODE
                                  (predictors_customization.py:25-69 -> cartpole_equations.py:229-249,293-308:
                                  Euler-Cromer, no bounce, angle = atan2(sin, cos)) */
} oracle_config;

#define PI_F 3.14159274101257324f
#define TWO_PI_F 6.28318548202514648f

/* A probe, not a reference arithmetic (tests/parity_util.py): with a nonzero seed every sin / cos result is moved to a
 * neighbouring float32 (up, down or not at all, decided by a hash of the argument and the seed) - "another float32 sin / cos
 * implementation", which is what a GPU's is against the host's.  Set before a call, read-only inside the parallel loops. */
static uint32_t g_trig_jitter = 0;
void oracle_set_trig_jitter(uint32_t seed) { g_trig_jitter = seed; }
static inline float jitter_(float r, float x, uint32_t salt) {
  if (!g_trig_jitter) return r;
  union { float f; uint32_t u; } b; b.f = x;
  uint32_t h = (b.u ^ (g_trig_jitter * 0x9E3779B9u) ^ salt) * 0x85EBCA6Bu;
  h ^= h >> 15; h *= 0xC2B2AE35u; h ^= h >> 13;
  switch (h & 3u) {
    case 0: return nextafterf(r, INFINITY);
    case 1: return nextafterf(r, -INFINITY);
    default: return r;
  }
}
#ifdef ORACLE_F32_TRIG      /* cpu_baseline timing builds only (oracle/Makefile: *_native*): libm's float kernels, as numba */
static inline float cos32(float x) { return jitter_(cosf(x), x, 0x11u); }
static inline float sin32(float x) { return jitter_(sinf(x), x, 0x22u); }
#else                       /* the checker */
static inline float cos32(float x) { return jitter_((float)cos((double)x), x, 0x11u); }
static inline float sin32(float x) { return jitter_((float)sin((double)x), x, 0x22u); }
#endif

/* ---- mode A: strict float32 ------------------------------------------------------------------------------------ */
static inline void ode_f32(const oracle_config* p, float L, float ca, float sa, float w, float v, float u, float* aDD,
                           float* xDD) {
  const float kp1 = p->k + 1.0f;
  const float A = kp1 * (p->m_cart + p->m_pole) - p->m_pole * (ca * ca);
  const float F = -p->M_fric * v;
  const float T = -p->J_fric * w;
  const float Lh = L / 2.0f;
  *xDD = (p->m_pole * p->g * sa * ca + ((T * ca) / Lh) + kp1 * (-(p->m_pole * Lh * (w * w) * sa) + F + u)) / A;
  *aDD = (p->g * sa + *xDD * ca + T / (p->m_pole * Lh)) / (kp1 * Lh);
}

static inline void substep_f32(const oracle_config* p, float L, float t, float u, float s[6]) {
  float aDD, xDD;
  ode_f32(p, L, s[2], s[3], s[1], s[5], u, &aDD, &xDD);
  if (p->integrator == 1) {                      /* Euler-Cromer: velocities first, positions with the NEW velocities */
    const float w1 = s[1] + aDD * t, v1 = s[5] + xDD * t;
    const float th1 = s[0] + w1 * t, x1 = s[4] + v1 * t;
    const float c1 = cos32(th1), s1 = sin32(th1);
    s[0] = atan2f(s1, c1); s[1] = w1; s[2] = c1; s[3] = s1; s[4] = x1; s[5] = v1;
    return;
  }
  float th = s[0] + s[1] * t, w = s[1] + aDD * t, x = s[4] + s[5] * t, v = s[5] + xDD * t;
  const float cb = cos32(th);
  if (x >= p->THL || -x >= p->THL) {
    w = w - 2.0f * (v * cb) / (0.5f * L);
    th = th + w * t;
    v = -v;
    x = x + v * t;
  }
  const float m = fmodf(th, TWO_PI_F);
  th = (m < -PI_F) ? (m + TWO_PI_F) : ((m > PI_F) ? (m - TWO_PI_F) : m);
  s[0] = th; s[1] = w; s[2] = cos32(th); s[3] = sin32(th); s[4] = x; s[5] = v;
}

/* ---- mode B: float64 substeps (numba typing emulation) ---------------------------------------------- */
static inline void substep_f64(const oracle_config* p, double L, double t, double u, double s[6]) {
  const double kp1 = (double)p->k + 1.0;
  const double A = kp1 * ((double)p->m_cart + (double)p->m_pole) - (double)p->m_pole * (s[2] * s[2]);
  const double F = -(double)p->M_fric * s[5];
  const double T = -(double)p->J_fric * s[1];
  const double Lh = L / 2.0;
  const double xDD = ((double)p->m_pole * (double)p->g * s[3] * s[2] + ((T * s[2]) / Lh) +
                      kp1 * (-((double)p->m_pole * Lh * (s[1] * s[1]) * s[3]) + F + u)) / A;
  const double aDD = ((double)p->g * s[3] + xDD * s[2] + T / ((double)p->m_pole * Lh)) / (kp1 * Lh);
  if (p->integrator == 1) {                      /* (a rounding-sensitivity probe only: the ODE predictor is float32 throughout) */
    const double w1 = s[1] + aDD * t, v1 = s[5] + xDD * t;
    const double th1 = s[0] + w1 * t, x1 = s[4] + v1 * t;
    const double c1 = cos(th1), s1 = sin(th1);
    s[0] = atan2(s1, c1); s[1] = w1; s[2] = c1; s[3] = s1; s[4] = x1; s[5] = v1;
    return;
  }
  double th = s[0] + s[1] * t, w = s[1] + aDD * t, x = s[4] + s[5] * t, v = s[5] + xDD * t;
  const double cb = cos(th);
  if (x >= (double)p->THL || -x >= (double)p->THL) {
    w = w - 2.0 * (v * cb) / (0.5 * L);
    th = th + w * t;
    v = -v;
    x = x + v * t;
  }
  const double two_pi = 2.0 * M_PI;
  const double m = fmod(th, two_pi);
  th = (m < -M_PI) ? (m + two_pi) : ((m > M_PI) ? (m - two_pi) : m);
  s[0] = th; s[1] = w; s[2] = cos(th); s[3] = sin(th); s[4] = x; s[5] = v;
}

/* One control step of predictor_ODE_v0: Q -> u = u_max*Q, S substeps of dt/S, float32 store. */
static inline void control_step(const oracle_config* p, float L, float Q, float s[6]) {
  const double t_step = (double)p->dt / (double)p->S;            /* python float */
  const float u = p->u_max * Q;
  if (!p->f64_substeps) {
    const float t = (float)t_step;                               /* weak python scalar -> float32 */
    for (uint32_t i = 0; i < p->S; ++i) substep_f32(p, L, t, u, s);
  } else {
    double d[6];
    for (int i = 0; i < 6; ++i) d[i] = (double)s[i];
    for (uint32_t i = 0; i < p->S; ++i) substep_f64(p, (double)L, t_step, (double)u, d);
    for (int i = 0; i < 6; ++i) s[i] = (float)d[i];
  }
}

/* ---- stage / terminal costs (float32) ---------------------------------------------------------------------------- */
static inline float stage_qbgm(const oracle_config* p, const float s[6], float u, float x_t, float te) {
  const float THL = p->THL, x = s[4];
  const float d = (x - x_t) / (2.0f * THL);
  const float dd = p->w[0] * (d * d);
  const float ptf = p->w[6];
  const float ax = fabsf(x);
  const float near = (ax > ptf * THL) ? 1.0f : 0.0f;
  const float b = (ax - ptf * THL) / ((1.0f - ptf) * THL);
  const float db = p->w[1] * (near * (b * b));
  const float e1 = 1.0f - te * cos32(s[0]);
  const float ep = p->w[2] * (e1 * e1);
  const float ekp = p->w[3] * (s[1] * s[1]);
  const float cc = p->w[4] * (p->w[5] * (u * u));
  return dd + db + ep + ekp + cc;
}

static inline float stage_default(const oracle_config* p, const float s[6], float u, float x_t, float te) {
  const float THL = p->THL, x = s[4];
  const float d = (x - x_t) / (2.0f * THL);
  const float ind = (fabsf(x) > 0.90f * THL) ? 1.0e7f : 0.0f;
  const float dd = p->w[0] * (d * d + ind);
  const float e1 = 1.0f - cos32(s[0]);
  const float ep = p->w[1] * (te * 0.25f * (e1 * e1));
  const float cc = p->w[2] * (p->w[3] * (u * u));
  return dd + ep + cc;
}

static inline float terminal_indicator(const oracle_config* p, const float s[6], float x_t) {
  return (fabsf(s[0]) > 0.2f || fabsf(s[4] - x_t) > 0.1f * p->THL) ? 10000.0f : 0.0f;
}

static inline float correction(const oracle_config* p, float u, float du) {
  return p->cc_weight * (0.5f * (1.0f - 1.0f / p->NU) * p->R * (du * du) + p->R * u * du + 0.5f * p->R * (u * u));
}

static inline float stage_legacy(const oracle_config* p, const float s[6], float u, float du, float u_prev, float x_t) {
  const float THL = p->THL, x = s[4];
  const float d = (x - x_t) / (2.0f * THL);
  const float ind = (fabsf(x) > 0.95f * THL) ? 1.0e6f : 0.0f;
  const float dd = p->w[0] * (d * d + ind);
  const float e1 = 1.0f - cos32(s[0]);
  const float ep = p->w[1] * (0.25f * (e1 * e1));
  const float ekp = p->w[2] * (s[1] * s[1]);
  const float ekc = p->w[3] * (s[5] * s[5]);
  float cc = p->w[4] * (0.5f * (1.0f - 1.0f / p->NU) * p->R * (du * du) + p->R * u * du + 0.5f * p->R * (u * u));
  const float ur = u + du;
  if (fabsf(ur) > 1.0f) cc = 1.0e5f;
  const float dcr = ur - u_prev;
  return dd + ep + ekp + ekc + cc + p->w[5] * (dcr * dcr);
}

static inline float shifted(const oracle_config* p, const float* un, uint32_t k) {
  if (p->shift_mode == 2) return un[k];
  if (k + 1 < p->H) return un[k + 1];
  return p->shift_mode == 0 ? un[p->H - 1] : 0.0f;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* predictor seam: s0[B,6], Q[B,H], L[B] or NULL(L_default) -> traj[B,H+1,6] */
void oracle_predict(const oracle_config* p, uint32_t B, uint32_t H, const float* s0, const float* Q, const float* L,
                    float L_default, float* traj, int n_threads) {
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < (int64_t)B; ++b) {
    float s[6];
    memcpy(s, s0 + b * 6, sizeof(s));
    float* o = traj + b * (size_t)(H + 1) * 6;
    memcpy(o, s, sizeof(s));
    const float Lb = L ? L[b] : L_default;
    for (uint32_t k = 0; k < H; ++k) {
      control_step(p, Lb, Q[b * H + k], s);
      memcpy(o + (size_t)(k + 1) * 6, s, sizeof(s));
    }
  }
}

/* Fused MPPI step for E envs (same semantics as cpmppi_step with CPMPPI_NOISE_DELTA_U).
 * u_nom[E,H] in/out, delta_u[E,N,H], u_prev[E,H] or NULL, S_out[E,N] or NULL, Q_out[E] or NULL. */
void oracle_step(const oracle_config* p, uint32_t E, const float* s0, float* u_nom, const float* delta_u,
                 const float* u_prev, const float* x_t, const float* te, const float* L, float L_default, float* Q_out,
                 float* S_out, int n_threads) {
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  const uint32_t N = p->N, H = p->H;
  float* S = (float*)malloc((size_t)E * N * sizeof(float));
#pragma omp parallel for schedule(static) collapse(2)
  for (int64_t e = 0; e < (int64_t)E; ++e) {
    for (int64_t n = 0; n < (int64_t)N; ++n) {
      const float* un = u_nom + e * H;
      const float* up = (u_prev ? u_prev : u_nom) + e * H;
      const float* du = delta_u + ((size_t)e * N + n) * H;
      const float Le = L ? L[e] : L_default;
      float s[6];
      memcpy(s, s0 + e * 6, sizeof(s));
      float cost = 0.0f, corr = 0.0f;
      for (uint32_t k = 0; k < H; ++k) {
        const float uk = shifted(p, un, k);
        float ur = uk + du[k];
        if (p->control_mode == 0) ur = fminf(fmaxf(ur, p->lo), p->hi);
        if (p->cost_id == 0) {
          cost += stage_qbgm(p, s, ur, x_t[e], te[e]);
          corr += correction(p, p->correction_u == 0 ? ur : uk, du[k]);
        } else if (p->cost_id == 1) {
          cost += stage_default(p, s, ur, x_t[e], te[e]);
          corr += correction(p, p->correction_u == 0 ? ur : uk, du[k]);
        } else {
          cost += stage_legacy(p, s, uk, du[k], up[k], x_t[e]);
        }
        control_step(p, Le, ur, s);
      }
      float total;
      if (p->cost_id == 2) {
        total = cost + terminal_indicator(p, s, x_t[e]);
      } else {
        const float term = (p->cost_id == 1) ? terminal_indicator(p, s, x_t[e]) : 0.0f;
        total = (p->horizon_reduce == 0) ? (cost + term) : (cost + term) / (float)(H + 1);
        total += corr;
      }
      S[(size_t)e * N + n] = total;
    }
  }
  /* soft-min weighted update per env (controller_mppi_cartpole.py:306-321) */
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < (int64_t)E; ++e) {
    const float* Se = S + (size_t)e * N;
    float rho = Se[0];
    for (uint32_t n = 1; n < N; ++n) rho = fminf(rho, Se[n]);
    float* ex = (float*)malloc(N * sizeof(float));
    double a = 0.0;
    for (uint32_t n = 0; n < N; ++n) { ex[n] = expf((-1.0f / p->LBD) * (Se[n] - rho)); a += ex[n]; }
    float* un = u_nom + e * H;
    float* tmp = (float*)malloc(H * sizeof(float));
    for (uint32_t k = 0; k < H; ++k) {
      double b = 0.0;
      for (uint32_t n = 0; n < N; ++n) b += (double)ex[n] * (double)delta_u[((size_t)e * N + n) * H + k];
      float v = shifted(p, un, k) + (float)(b / a);
      if (p->control_mode == 0) v = fminf(fmaxf(v, p->lo), p->hi);
      tmp[k] = v;
    }
    memcpy(un, tmp, H * sizeof(float));
    if (Q_out) Q_out[e] = un[0];
    free(tmp);
    free(ex);
  }
  if (S_out) memcpy(S_out, S, (size_t)E * N * sizeof(float));
  free(S);
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  static int cached = 0;             /* the value at first call (the binding calls it when it loads the library): a later */
  if (!cached) cached = omp_get_max_threads();   /* n_threads = 1 run must not stick, and affinity limits are respected */
  return cached;
#else
  return 1;
#endif
}
