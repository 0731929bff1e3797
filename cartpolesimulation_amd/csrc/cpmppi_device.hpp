// cpmppi_device.hpp — device-side building blocks of the MPPI rollout path (gfx950 only).
//
// Arithmetic spec: SURVEY.md Appendix A, restating (reference checkout paths)
//   CartPole/cartpole_equations.py:44-105   _cartpole_ode
//   CartPole/cartpole_equations.py:356-364  cartpole_integration_numba (simultaneous forward Euler)
//   CartPole/cartpole_equations.py:341-347  edge_bounce
//   CartPole/_CartPole_mathematical_helpers.py:24-29  wrap_angle_rad_inplace
//   CartPole/cartpole_numba.py:55-78        cartpole_fine_integration_numba (the substep loop)
// Everything is float32.  All per-env quantities are wave-uniform and end up in SGPRs (they are derived from kernel
// arguments and blockIdx only).
//
// Lane mapping.  A rollout is one serial dependency chain of ~30 000 VALU instructions.  Measured on MI355X
// (tools/valu_peak.hip): a dependent chain inside ONE wave issues at ~1.85 ns per wave64 instruction however many
// waves share the SIMD, two independent chains inside one wave at ~1.0 ns, and v_pk_fma_f32 performs two FMAs for the
// price of one instruction.  The FAST path is therefore written generically over F = float (one rollout per lane: the
// latency-optimal mapping when there are few rollouts) and F = float2 (two rollouts per lane: every arithmetic
// instruction is a packed v_pk_* op or one of two independent scalar ops; the throughput mapping).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cpmppi {

// Diagnostic build only (-DCPMPPI_DEBUG_COUNTERS, tools/dev/cold_counts.py): per-translation-unit event counters and
// per-wave lifetimes of the rollout kernel.  No counter exists in the product build.
#ifdef CPMPPI_DEBUG_COUNTERS
static __device__ unsigned long long g_dbg[8];        // unused slots kept for ad-hoc counters
static __device__ unsigned long long g_wave_cycles[16384];
static __device__ unsigned int g_wave_cold[16384];    // cold-branch entries of each wave (all substep flavours)
static __device__ unsigned long long g_wave_t[16384][4];   // s_memrealtime (100 MHz, chip-wide): entry, loop end, partials written, exit
// -DCPMPPI_SECTION_STAMPS on top (tools/dev/sections.py): s_memtime at the section boundaries of a control step, summed per
// wave.  `sec[0..6]` accumulate shader cycles per section, `sec[7]` holds the previous stamp.
static __device__ unsigned int g_wave_sec[16384][8];
// where a wave runs (tools/dev/placement.py): HW_REG_HW_ID (wave / SIMD / CU / shader array / shader engine) and HW_REG_XCC_ID
static __device__ unsigned int g_wave_hw[16384][2];
#define CPMPPI_DBG_STAMP(slot)                                                                       \
  do {                                                                                               \
    const unsigned wv_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                        \
    if ((threadIdx.x & 63u) == 0u && wv_ < 16384u) {                                                 \
      cpmppi::g_wave_t[wv_][slot] = __builtin_amdgcn_s_memrealtime();                                \
      if ((slot) == 0) {                                                                             \
        unsigned hw_, xcc_;                                                                          \
        asm volatile("s_getreg_b32 %0, hwreg(4)" : "=s"(hw_));                                       \
        asm volatile("s_getreg_b32 %0, hwreg(20)" : "=s"(xcc_));                                     \
        cpmppi::g_wave_hw[wv_][0] = hw_; cpmppi::g_wave_hw[wv_][1] = xcc_;                           \
      }                                                                                              \
    }                                                                                                \
  } while (0)
#define CPMPPI_HW_READER(NAME)                                                                                          \
  extern "C" int NAME(unsigned int* hw, unsigned n_waves) {                                                            \
    return hipMemcpyFromSymbol(hw, HIP_SYMBOL(cpmppi::g_wave_hw), (size_t)n_waves * 8) == hipSuccess ? 0 : -1;         \
  }
// one reader per translation unit (the arrays are per unit): extern "C" int NAME(cold, cycles, stamps, n_waves, reset)
#define CPMPPI_DEBUG_READER(NAME)                                                                                      \
  extern "C" int NAME(unsigned int* wave_cold, unsigned long long* wave_cycles, unsigned long long* stamps,            \
                      unsigned n_waves, int reset) {                                                                   \
    if (wave_cold && hipMemcpyFromSymbol(wave_cold, HIP_SYMBOL(cpmppi::g_wave_cold), (size_t)n_waves * 4) != hipSuccess) return -1; \
    if (wave_cycles && hipMemcpyFromSymbol(wave_cycles, HIP_SYMBOL(cpmppi::g_wave_cycles), (size_t)n_waves * 8) != hipSuccess) return -1; \
    if (stamps && hipMemcpyFromSymbol(stamps, HIP_SYMBOL(cpmppi::g_wave_t), (size_t)n_waves * 32) != hipSuccess) return -1; \
    if (reset) {                                                                                                       \
      static unsigned int z[16384];                                                                                    \
      if (hipMemcpyToSymbol(HIP_SYMBOL(cpmppi::g_wave_cold), z, sizeof(z)) != hipSuccess) return -1;                   \
    }                                                                                                                  \
    return 0;                                                                                                          \
  }
#define CPMPPI_SECTION_READER(NAME)                                                                                     \
  extern "C" int NAME(unsigned int* sections, unsigned n_waves) {                                                      \
    return hipMemcpyFromSymbol(sections, HIP_SYMBOL(cpmppi::g_wave_sec), (size_t)n_waves * 32) == hipSuccess ? 0 : -1; \
  }
#define CPMPPI_DBG(i, n)                                                                             \
  do {                                                                                               \
    const unsigned wv_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                        \
    if ((i) != 1 && (i) != 2 && (threadIdx.x & 63u) == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)) && wv_ < 16384u) \
      atomicAdd(&cpmppi::g_wave_cold[wv_], 1u);                                                      \
  } while (0)
#else
#define CPMPPI_DBG(i, n) ((void)0)
#define CPMPPI_DBG_STAMP(slot) ((void)0)
#endif

constexpr float PI_F = 3.14159274101257324f;       // float32(np.pi)
constexpr float TWO_PI_F = 6.28318548202514648f;   // float32(2*np.pi)

enum : int { COST_QBGM = 0, COST_DEFAULT = 1, COST_LEGACY = 2, COST_QBG = 3 };
enum : int { NOISE_DELTA_U = 0, NOISE_KNOTS = 1, NOISE_PHILOX = 2, NOISE_TILED = 3 };

// Kernel-argument block (passed by value -> kernarg segment -> SGPRs).
struct Params {
  uint32_t E, N, H, S, P;            // P = number of knots = ceil(H/period)+1
  uint32_t period;
  float t_step;                      // dt / S
  float k, m_cart, m_pole, g, J_fric, M_fric, u_max, THL, L_default;
  uint32_t cost_id;
  float w[24];
  float R, LBD, NU, cc_weight, sigma;
  float lo, hi;
  float run_lo, run_hi;              // limits of the control a rollout applies: (lo, hi) when control_mode clips, (-inf, inf)
                                     // otherwise - the clamp is then unconditional (no select on the mode per control step)
  uint32_t horizon_reduce, control_mode, shift_mode, correction_u;
  uint32_t interp_f32;               // FAST + device-generated knots: interpolate with one float32 FMA (<= 1 ulp of the
                                     // float64 scipy form, which stays in force for caller-provided knots)
  uint32_t qb_mode;                  // cost_id == COST_DEFAULT only: 0 = default.py, 1 = quadratic_boundary.py, 2 = its
                                     // _nonconvex sibling (the public ids CPMPPI_COST_QB / _QB_NONCONVEX; same kernels)
};

// Per-env constants.  PRECISE keeps the reference's operands; FAST folds them (all wave-uniform).
constexpr float ROT_LIMIT_LO = 0.125f;
constexpr float ROT_LIMIT = 0.25f;
#ifndef CPMPPI_SEED_LO
#define CPMPPI_SEED_LO 1        // packed path: the carried pair is seeded from the degree-5/4 polynomials (|w t| <= 0.125)
#endif
constexpr float ROT_LIMIT_SEED = CPMPPI_SEED_LO ? ROT_LIMIT_LO : ROT_LIMIT;     // (see rot_pair_lo / rot_pair below)

struct EnvConst {
  float L, Lh;
  float kp1, kp1_mt;                 // (k+1), (k+1)*(m_cart+m_pole)
  float mg, JinvLh, kmLh, kM, g_i, cT_i, inv_kLh, inv_halfL;
  float uK_scale;                    // (k+1) u_max: FAST forms (k+1) u = (k+1) u_max Q with one product
  float t1_i;                        // inv_kLh / m_pole: g_i s - cT_i w = t1_i (m_p g s - J/Lh w), the bracket xDD's numerator forms anyway
  float tg_i, tcT_i, tinv_kLh;       // the same three angleDD coefficients times the substep length t
  float wlim;                        // ROT_LIMIT_SEED / t: |w| beyond it leaves the carried rotation pair's seed range (packed path)
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
  c.t1_i = (float)(inv_kLh / (double)p.m_pole);
  c.uK_scale = (float)(kp1 * (double)p.u_max);
  const double t = (double)p.t_step;
  c.tg_i = (float)(t * (double)p.g * inv_kLh);
  c.tcT_i = (float)(t * ((double)p.J_fric / ((double)p.m_pole * Lh) * inv_kLh));
  c.tinv_kLh = (float)(t * inv_kLh);
  c.wlim = ROT_LIMIT_SEED / p.t_step;
  return c;
}

// For kernels where the env (hence L) is the same for the whole wave: pin every field to an SGPR.  Field by field — the
// first version walked the struct through a float pointer, which kept it in a 28-byte private (scratch) slot per lane:
// 117 MB of scratch stores per 8192-env launch.
__device__ __forceinline__ float uniform_(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ EnvConst make_env_const_uniform(const Params& p, float L) {
  const EnvConst c = make_env_const(p, L);
  EnvConst u;
  u.L = uniform_(c.L); u.Lh = uniform_(c.Lh); u.kp1 = uniform_(c.kp1); u.kp1_mt = uniform_(c.kp1_mt);
  u.mg = uniform_(c.mg); u.JinvLh = uniform_(c.JinvLh); u.kmLh = uniform_(c.kmLh); u.kM = uniform_(c.kM);
  u.g_i = uniform_(c.g_i); u.cT_i = uniform_(c.cT_i); u.inv_kLh = uniform_(c.inv_kLh); u.inv_halfL = uniform_(c.inv_halfL);
  u.tg_i = uniform_(c.tg_i); u.tcT_i = uniform_(c.tcT_i); u.tinv_kLh = uniform_(c.tinv_kLh); u.t1_i = uniform_(c.t1_i);
  u.uK_scale = uniform_(c.uK_scale); u.wlim = uniform_(c.wlim);
  return u;
}

// ------------------------------------------------------------------------------------------------------------------
// float / float2 helpers
typedef float f2 __attribute__((ext_vector_type(2)));

template <int R> struct Lanes;
template <> struct Lanes<1> { using F = float; };
template <> struct Lanes<2> { using F = f2; };

template <class F> struct Width;
template <> struct Width<float> { static constexpr int value = 1; };
template <> struct Width<f2> { static constexpr int value = 2; };

__device__ __forceinline__ float get(float v, int) { return v; }
__device__ __forceinline__ float get(f2 v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ void put(float& v, int, float x) { v = x; }
__device__ __forceinline__ void put(f2& v, int i, float x) { if (i == 0) v.x = x; else v.y = x; }
template <class F> __device__ __forceinline__ F splat(float x);
template <> __device__ __forceinline__ float splat<float>(float x) { return x; }
template <> __device__ __forceinline__ f2 splat<f2>(float x) { return f2{x, x}; }

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f2 fma_(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float rcp_(float a) { return __builtin_amdgcn_rcpf(a); }
#ifndef CPMPPI_RCP_SHARED
#define CPMPPI_RCP_SHARED 0     // float2: one v_rcp_f32 (a quarter-rate instruction) for both lanes, 1/(ab) * (b, a)
#endif
__device__ __forceinline__ f2 rcp_(f2 a) {
#if CPMPPI_RCP_SHARED
  const float r = __builtin_amdgcn_rcpf(a.x * a.y);
  return f2{r, r} * a.yx;
#else
  return f2{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)};
#endif
}
__device__ __forceinline__ float rint_(float a) { return __builtin_rintf(a); }
__device__ __forceinline__ f2 rint_(f2 a) { return f2{__builtin_rintf(a.x), __builtin_rintf(a.y)}; }
__device__ __forceinline__ float abs_(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ f2 abs_(f2 a) { return f2{__builtin_fabsf(a.x), __builtin_fabsf(a.y)}; }
__device__ __forceinline__ float clamp_(float a, float lo, float hi) { return __builtin_amdgcn_fmed3f(a, lo, hi); }
__device__ __forceinline__ f2 clamp_(f2 a, float lo, float hi) { return f2{clamp_(a.x, lo, hi), clamp_(a.y, lo, hi)}; }
// max(|a|, b) / max(|a|, |b|) as ONE v_max_f32 with source modifiers (fmaxf(fabsf(a), ...) comes out as three instructions: in
// IEEE mode the compiler canonicalises each operand of a maximum with a v_max x, x of its own; the operands here are results of
// arithmetic, never signalling NaNs)
__device__ __forceinline__ float max_abs_(float a, float b) {
  float r;
  asm("v_max_f32_e64 %0, |%1|, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float max_abs2_(float a, float b) {
  float r;
  asm("v_max_f32_e64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// max(m, |a|, |b|) as ONE v_max3_f32 (m >= 0; a NaN operand is ignored, as by an ordered compare)
__device__ __forceinline__ float max3_abs2_(float m, float a, float b) {
  float r;
  asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float cos_(float a) { return cosf(a); }
__device__ __forceinline__ f2 cos_(f2 a) { return f2{cosf(a.x), cosf(a.y)}; }

// ------------------------------------------------------------------------------------------------------------------
// sincos on [-pi_f32, pi_f32] (the angle is wrapped every substep, so the argument never leaves this range).
// Cody-Waite reduction to |r| <= pi/4 with q in {-2..2} (q*PIO2_HI is exact), then the classic single-precision
// minimax polynomials; <= 1.5 ulp for both outputs over the range.
template <class F>
__device__ __forceinline__ void sincos_pi(F x, F& sn, F& cs) {
  constexpr float TWO_OVER_PI = 0.636619746685028076f;
  constexpr float PIO2_HI = 1.57079637050628662f;
  constexpr float PIO2_LO = -4.37113900018624283e-8f;
  const F q = rint_(x * splat<F>(TWO_OVER_PI));
  F r = fma_(-q, splat<F>(PIO2_HI), x);
  r = fma_(-q, splat<F>(PIO2_LO), r);
  const F r2 = r * r;
  F ps = fma_(r2, splat<F>(-1.9515295891e-4f), splat<F>(8.3321608736e-3f));
  ps = fma_(ps, r2, splat<F>(-1.6666654611e-1f));
  const F S = fma_(ps * r2, r, r);
  F pc = fma_(r2, splat<F>(2.443315711809948e-5f), splat<F>(-1.388731625493765e-3f));
  pc = fma_(pc, r2, splat<F>(4.166664568298827e-2f));
  const F C = fma_(pc * r2, r2, fma_(splat<F>(-0.5f), r2, splat<F>(1.0f)));
#pragma unroll
  for (int i = 0; i < Width<F>::value; ++i) {
    const int n = (int)get(q, i);
    const bool swap = (n & 1) != 0;
    const float s0 = swap ? get(C, i) : get(S, i);
    const float c0 = swap ? get(S, i) : get(C, i);
    const uint32_t sflip = ((uint32_t)n & 2u) << 30;
    const uint32_t cflip = ((uint32_t)(n + 1) & 2u) << 30;
    put(sn, i, __uint_as_float(__float_as_uint(s0) ^ sflip));
    put(cs, i, __uint_as_float(__float_as_uint(c0) ^ cflip));
  }
}

// Variant without per-lane quadrant selects: reduce by multiples of pi to |r| <= pi/2 (q in {-1,0,1}, q*PI_HI exact),
// sin x = (-1)^q sin r, cos x = (-1)^q cos r; the sign is a multiply, so every instruction has a packed float2 form.
// Polynomials: sin r = r + r z P3(z), cos r = 1 + z Q4(z), z = r^2, fitted on [-pi/2, pi/2] (Lawson-weighted
// least squares, coefficients rounded to float32): relative error 1.2e-8 (sin), absolute 4.9e-9 (cos) before rounding.
template <class F>
__device__ __forceinline__ void sincos_pi_half(F x, F& sn, F& cs) {
  constexpr float INV_PI = 0.318309886183790672f;
  constexpr float PI_HI = 3.14159274101257324f;
  constexpr float PI_LO = -8.74227765734758577e-8f;
  const F q = rint_(x * splat<F>(INV_PI));
  F r = fma_(-q, splat<F>(PI_HI), x);
  r = fma_(-q, splat<F>(PI_LO), r);
  const F z = r * r;
  // (-1)^q for q in {-1, 0, 1}: 1 - 2|q|; packed float2 has no abs modifier, there q*q is one instruction instead of two
  F sg;
  if constexpr (Width<F>::value == 2) sg = fma_(q * q, splat<F>(-2.0f), splat<F>(1.0f));
  else sg = fma_(abs_(q), splat<F>(-2.0f), splat<F>(1.0f));
  F P = fma_(z, splat<F>(2.6056311526190257e-06f), splat<F>(-0.00019809538207482547f));
  P = fma_(P, z, splat<F>(0.008333065547049046f));
  P = fma_(P, z, splat<F>(-0.16666659712791443f));
  const F rs = r * sg;
  sn = fma_(rs * z, P, rs);
  F Q = fma_(z, splat<F>(-2.6075662162838853e-07f), splat<F>(2.476180816302076e-05f));
  Q = fma_(Q, z, splat<F>(-0.0013888402609154582f));
  Q = fma_(Q, z, splat<F>(0.04166664183139801f));
  Q = fma_(Q, z, splat<F>(-0.5f));
  cs = fma_(z * sg, Q, sg);
}

#ifndef CPMPPI_SINCOS_MODE
#define CPMPPI_SINCOS_MODE 1    // 0: pi/4 reduction + quadrant selects   1: pi/2 reduction + sign multiply
#endif
#ifndef CPMPPI_WRAP_MODE
#if defined(CPMPPI_DEBUG_COUNTERS) && defined(CPMPPI_SECTION_STAMPS)
// the state is pinned in front of the stamp (an opaque asm it passes through), so a section's arithmetic cannot drift
// across its boundary
#define CPMPPI_SEC(sec, i, ST)                                                                                          \
  do {                                                                                                                 \
    asm volatile("" : "+v"((ST).th), "+v"((ST).w), "+v"((ST).c), "+v"((ST).s), "+v"((ST).x), "+v"((ST).v));           \
    const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime();                                                      \
    (sec)[i] += now_ - (sec)[7];                                                                                       \
    (sec)[7] = now_;                                                                                                   \
  } while (0)
#else
#define CPMPPI_SEC(sec, i, ST) ((void)0)
#endif
#define CPMPPI_WRAP_MODE 1      // 0: the reference's two comparisons     1: theta - 2pi*rint(theta/2pi)
#endif

template <class F>
struct State {
  F th, w, c, s, x, v;   // angle, angleD, angle_cos, angle_sin, position, positionD
};

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

// One Euler substep, PRECISE: the reference's operand grouping with IEEE divides, libm sincos and no FMA contraction.
__device__ __forceinline__ void substep_precise(State<float>& st, float u, float t, const Params& p,
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
__device__ __forceinline__ void plant_substep(State<float>& st, float aDD, float xDD, float t, const Params& p,
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

#ifndef CPMPPI_NEWTON
#define CPMPPI_NEWTON 0         // 0: num * v_rcp_f32(A) (<= 1.5 ulp; measured deviation identical)   1: + one Newton correction
#endif
#ifndef CPMPPI_FOLD_T
#define CPMPPI_FOLD_T 0         // 1: fold t into the angleDD coefficients (-1 instruction; measured: 2.4x the deviation, so off)
#endif
#ifndef CPMPPI_T1_REUSE
#define CPMPPI_T1_REUSE 1       // angleDD from the numerator's own bracket (-1 instruction per substep)
#endif
#ifndef CPMPPI_LATENCY_NEAR
#define CPMPPI_LATENCY_NEAR 0   // one rollout per lane: the last substep tests the edge directly (no coarser "near" limit)
#endif
#ifndef CPMPPI_QBGM_FOLD
#define CPMPPI_QBGM_FOLD 1      // FAST: folded factors in quadratic_boundary_grad_minimal's stage cost
#endif
#ifndef CPMPPI_HOIST_SPIN
#define CPMPPI_HOIST_SPIN 1     // test |w t| once per control step (<= 0.1) instead of every substep (<= 0.125)
#endif
#ifndef CPMPPI_ROTATE
#define CPMPPI_ROTATE 1         // 1: rotate (cos, sin) on intermediate substeps, full wrap + sincos at the control step's end
#endif

// ODE + simultaneous forward Euler of one substep with folded constants (FAST).  Outputs the un-wrapped new state.
//   A    = (k+1)(m_c+m_p) - m_p c^2
//   xDD  = [ c (m_p g s - J/Lh w) - (k+1) m_p Lh w^2 s - (k+1) M_fric v + (k+1) u ] / A
//   aDD  = g_i s + xDD c /((k+1)Lh) - cT w                                      (cartpole_equations.py:95-99)
template <class F>
__device__ __forceinline__ void ode_euler_fast(const State<F>& st, F uK, float t, const Params& p, const EnvConst& e,
                                               F& th1, F& w1, F& x1, F& v1, F* aDD_out = nullptr) {
  const F c = st.c, s = st.s, w = st.w, v = st.v;
  const F A = fma_(-(c * splat<F>(p.m_pole)), c, splat<F>(e.kp1_mt));
  const F t1 = fma_(splat<F>(e.mg), s, -(w * splat<F>(e.JinvLh)));
  F num = fma_(c, t1, uK);
  num = fma_(-((w * w) * splat<F>(e.kmLh)), s, num);
  num = fma_(splat<F>(-e.kM), v, num);
  const F r = rcp_(A);                                  // A in [0.33, 0.43]: no scaling needed
#if CPMPPI_NEWTON
  const F q0 = num * r;
  const F xDD = fma_(fma_(-A, q0, num), r, q0);
#else
  const F xDD = num * r;
#endif
  const F tt = splat<F>(t);
  th1 = fma_(w, tt, st.th);
  // (do NOT fold "1 - cT*t" into one constant: its rounding error would bias w the same way every substep)
#if CPMPPI_FOLD_T
  // w + t*aDD with t folded into the three coefficients (each product still rounds relative to its own size)
  w1 = fma_(splat<F>(e.tg_i), s, fma_(xDD * c, splat<F>(e.tinv_kLh), fma_(w, splat<F>(-e.tcT_i), w)));
#elif CPMPPI_T1_REUSE
  // g s + T/(m_p Lh) is the bracket t1 = m_p g s - (J/Lh) w of xDD's numerator divided by m_p: one product instead of a
  // product and an FMA per substep (same formula, one rounding placed differently)
  const F aDD = fma_(xDD * c, splat<F>(e.inv_kLh), t1 * splat<F>(e.t1_i));
  w1 = fma_(aDD, tt, w);
  if (aDD_out) *aDD_out = aDD;
#else
  const F aDD = fma_(splat<F>(e.g_i), s, fma_(xDD * c, splat<F>(e.inv_kLh), -(w * splat<F>(e.cT_i))));
  w1 = fma_(aDD, tt, w);
  if (aDD_out) *aDD_out = aDD;
#endif
  x1 = fma_(v, tt, st.x);
  v1 = fma_(xDD, tt, v);
}

// Edge bounce of one lane (cartpole_equations.py:341-347), rare.  `cb` = cos of the integrated (un-wrapped) angle.
__device__ __forceinline__ void bounce_lane(float& thi, float& wi, float& xi, float& vi, float cb, float t, float inv_halfL) {
  wi = __builtin_fmaf(-(2.0f * (vi * cb)), inv_halfL, wi);
  thi = __builtin_fmaf(wi, t, thi);
  vi = -vi;
  xi = __builtin_fmaf(vi, t, xi);
}

// theta - 2pi*rint(theta/2pi): k*2pi_f32 is subtracted exactly for |k| <= 2 (Sterbenz), i.e. this IS fmod followed by the
// reference's two comparisons except when theta is within one ulp of +-pi, where both values are the same point of
// the circle; for absurd |theta| (> 4pi in one substep) it differs from the exact fmod by <= 0.5 ulp.
template <class F>
__device__ __forceinline__ F wrap_rint(F th) {
  return fma_(-rint_(th * splat<F>(0.159154943091895336f)), splat<F>(TWO_PI_F), th);
}

// ------------------------------------------------------------------------------------------------------------------
// Rotation by a small angle d: (cos d, sin d) from their Taylor polynomials.
//   rot_pair_lo: degree 5 / 4, |d| <= 0.125 (truncation < 1e-9): evaluated every substep by the one-rollout-per-lane path
//   rot_pair   : degree 7 / 6, |d| <= ROT_LIMIT = 0.25 (truncation 1e-11 / 4e-10): seeds the carried pair of the packed
//                path once per control step and serves every rare event (bounce, fast-spinning lane) of both paths
// |d| = |w t| > 0.25 means an angular velocity beyond 125 rad/s at t = 2 ms; such a lane is evaluated with the exact
// wrap + sincos on every substep (it practically never happens: a pole released from rest tops out near 20 rad/s).
// (ROT_LIMIT_LO = 0.125, ROT_LIMIT = 0.25: defined ahead of EnvConst, which carries the seed's limit as an angular velocity)
// Packed path: the range of |w t| within which a control step runs on the carried rotation pair, ROT_LIMIT_SEED.  Seeded from
// rot_pair_lo (CPMPPI_SEED_LO; two instructions fewer per control step than the degree-7/6 pair; truncation below 1e-9 up to
// 0.125 rad per substep = 62 rad/s at t = 2 ms - a pole released from rest tops out near 20), lanes beyond take the exact sincos
// on every substep.

template <class F>
__device__ __forceinline__ void rot_pair_lo(F d, F& cd, F& sd) {
  const F d2 = d * d;
  sd = fma_(d * d2, fma_(d2, splat<F>(8.3333333e-3f), splat<F>(-1.6666667e-1f)), d);
  cd = fma_(d2, fma_(d2, splat<F>(4.1666667e-2f), splat<F>(-0.5f)), splat<F>(1.0f));
}

template <class F>
__device__ __forceinline__ void rot_pair(F d, F& cd, F& sd) {
  const F d2 = d * d;
  F ps = fma_(d2, splat<F>(-1.9841270e-4f), splat<F>(8.3333333e-3f));
  ps = fma_(ps, d2, splat<F>(-1.6666667e-1f));
  sd = fma_(d * d2, ps, d);
  F pc = fma_(d2, splat<F>(-1.3888889e-3f), splat<F>(4.1666667e-2f));
  pc = fma_(pc, d2, splat<F>(-0.5f));
  cd = fma_(d2, pc, splat<F>(1.0f));
}

// (cos, sin)(a + d) from (cos, sin)(a) and |d| <= ROT_LIMIT.
__device__ __forceinline__ void rotate_lane(float& c, float& s, float d) {
  float cd, sd;
  rot_pair<float>(d, cd, sd);
  const float c1 = __builtin_fmaf(c, cd, -(s * sd)), s1 = __builtin_fmaf(s, cd, c * sd);
  c = c1; s = s1;
}

// Rare event of ONE lane, shared by every FAST substep flavour.  On entry (thi, wi, xi, vi) is the Euler-advanced state
// (angle un-wrapped) and (ci, si) = (cos, sin) of thi if `have_cs`, else they are evaluated here: by rotating the
// previous substep's pair (c0, s0) through d0 = w_old t when |d0| <= ROT_LIMIT, exactly (wrap + polynomial sincos,
// the angle is stored wrapped) otherwise.  A lane beyond the track edge bounces (cartpole_equations.py:341-347: cos of
// the integrated angle, then the angle advances by the NEW angular velocity); (ci, si) follow by one more rotation.
// Returns false if the lane's angular velocity is beyond the rotation range (its caller then treats it exactly on
// every substep).  No libm call on this path: a wave that hits it pays ~40 instructions, about one substep.
__device__ __forceinline__ bool rare_lane(float& thi, float& wi, float& xi, float& vi, float& ci, float& si, bool have_cs,
                                          float c0, float s0, float d0, float t, float THL, float inv_halfL) {
  bool in_range = true;
  if (!have_cs) {
    if (__builtin_fabsf(d0) <= ROT_LIMIT) {
      ci = c0; si = s0;
      rotate_lane(ci, si, d0);
    } else {
      thi = fma_(-rint_(thi * 0.159154943091895336f), TWO_PI_F, thi);
      sincos_pi_half<float>(thi, si, ci);
      in_range = false;
    }
  }
  if (__builtin_fabsf(xi) >= THL) {
    bounce_lane(thi, wi, xi, vi, ci, t, inv_halfL);
    const float dl = wi * t;
    if (__builtin_fabsf(dl) <= ROT_LIMIT) {
      rotate_lane(ci, si, dl);
    } else {
      thi = fma_(-rint_(thi * 0.159154943091895336f), TWO_PI_F, thi);
      sincos_pi_half<float>(thi, si, ci);
      in_range = false;
    }
  }
  return in_range;
}

// The bounce of cartpole_equations.py:341-347 applied to ALL lanes of the wave under a 0/1 mask `m` (packed where F is
// float2): with one wave per SIMD the slowest wave sets a small launch's time, and a rollout that is caught beyond the
// edge bounces on EVERY substep (v flips sign whichever way it points; the oracle shows rollouts with > 100 consecutive
// bounces), so this event has to cost about as much as a substep, not the ~1000 cycles of per-lane divergent code.
// `c1` = cos of the integrated (un-wrapped) angle.  Returns dl = m * w_new * t, the extra angle a bounced lane advances by.
template <class F>
__device__ __forceinline__ F bounce_masked(F m, F c1, F& th1, F& w1, F& x1, F& v1, float t, float inv_halfL) {
  w1 = fma_(-(v1 * c1), m * splat<F>(2.0f * inv_halfL), w1);
  const F mt = m * splat<F>(t);
  th1 = fma_(mt, w1, th1);
  v1 = v1 * fma_(m, splat<F>(-2.0f), splat<F>(1.0f));
  x1 = fma_(mt, v1, x1);
  return mt * w1;
}

template <class F>
__device__ __forceinline__ void rotate_pair(F& c, F& s, F cd, F sd) {
  const F cn = fma_(c, cd, -(s * sd));
  s = fma_(s, cd, c * sd);
  c = cn;
}

// One Euler substep, FAST, with the reference's per-substep wrap and sin/cos evaluation (the control step's LAST
// substep of the rotating flavours, every substep of the plain one).
// CHECK = false: no edge test at all (callers that have excluded the event; none at present).
// `nearlim` <= THL (wave-uniform): the ONE pair of compares of the common path tests |x| against it instead of THL, and
// the return value says whether any lane of the wave ends the substep at or beyond it - the next stage's boundary cost
// (nonzero only for |x| > permissible_track_fraction * THL = nearlim) is evaluated only then.  The edge itself is tested
// behind that branch, so the common path costs what it did with the plain edge test.
// NEAR = false: `nearlim` is not used, the common path tests the edge itself (the latency build).
// INLINE_EVENTS: the bounce arithmetic is evaluated on every call under its mask instead of behind the wave-uniform branch
// (the eventful loop of the phased build: a wave that is there bounces on nearly every control step, and behind the branch
// the block costs ~500 cycles against ~150 inline).
template <class F, bool CHECK = true, bool NEAR = true, bool INLINE_EVENTS = false>
__device__ __forceinline__ bool substep_fast(State<F>& st, F uK, float t, const Params& p, const EnvConst& e, float nearlim,
                                             bool check = true, bool* at_edge = nullptr) {
  constexpr int W = Width<F>::value;
  F th1, w1, x1, v1;
  ode_euler_fast<F>(st, uK, t, p, e, th1, w1, x1, v1);
  uint64_t near = 0, rare = 0;
  if constexpr (NEAR) {
    if (CHECK && check) {
#pragma unroll
      for (int i = 0; i < W; ++i) near |= __builtin_amdgcn_fcmpf(__builtin_fabsf(get(x1, i)), nearlim, 11);   // 11 = unordered or >=
    }
    if (CHECK && __builtin_expect(near != 0, 0)) {
#pragma unroll
      for (int i = 0; i < W; ++i) rare |= __builtin_amdgcn_fcmpf(__builtin_fabsf(get(x1, i)), p.THL, 3);
    }
  } else if (CHECK && check) {
#pragma unroll
    for (int i = 0; i < W; ++i) rare |= __builtin_amdgcn_fcmpf(__builtin_fabsf(get(x1, i)), p.THL, 3);
    near = ~0ull;
  }
  if (at_edge) *at_edge = rare != 0;
  if (CHECK && (INLINE_EVENTS || __builtin_expect(rare != 0, 0))) {     // wave-uniform: no exec bookkeeping when cold
    CPMPPI_DBG(3, 1);
    // cos of the integrated angle by rotating the previous pair through d = w t; lanes beyond the rotation range (deep)
    const F d = st.w * splat<F>(t);
    F m;
    bool deep = false;
#pragma unroll
    for (int i = 0; i < W; ++i) {
      const bool hit = __builtin_fabsf(get(x1, i)) >= p.THL, far = __builtin_fabsf(get(d, i)) > ROT_LIMIT;
      put(m, i, (hit && !far) ? 1.0f : 0.0f);
      deep |= hit && far;
    }
    const F th0 = th1, w0 = w1, x0 = x1, v0 = v1;
    F cd, sd;
    rot_pair<F>(d, cd, sd);
    const F cb = fma_(st.c, cd, -(st.s * sd));
    bounce_masked<F>(m, cb, th1, w1, x1, v1, t, e.inv_halfL);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(deep) != 0, 0)) {
#pragma unroll
      for (int i = 0; i < W; ++i) {
        if (__builtin_fabsf(get(x0, i)) >= p.THL && __builtin_fabsf(get(d, i)) > ROT_LIMIT) {
          float thi = get(th0, i), wi = get(w0, i), xi = get(x0, i), vi = get(v0, i), ci, si;
          rare_lane(thi, wi, xi, vi, ci, si, false, get(st.c, i), get(st.s, i), get(d, i), t, p.THL, e.inv_halfL);
          put(th1, i, thi); put(w1, i, wi); put(x1, i, xi); put(v1, i, vi);
        }
      }
    }
  }
  th1 = wrap_rint<F>(th1);
  st.th = th1; st.w = w1; st.x = x1; st.v = v1;
  sincos_pi_half<F>(th1, st.s, st.c);
  return near != 0;
}

// Intermediate substep, FAST + ROTATE: the angle is left un-wrapped and (cos, sin) are advanced by the rotation
// through d = w*t, the exact increment of the angle, with sin d, cos d from their Taylor polynomials (|d| <= 0.125:
// truncation < 1e-9).  Rounding adds ~1 ulp per rotation; the control step's LAST substep re-synchronises with the full
// wrap + sincos (substep_fast), so the states observed at control-step granularity carry at most S-1 rotations of drift
// (~2e-7).  Lanes that bounce, or spin faster than 0.125 rad per substep, are re-evaluated in the cold branch
// (rare_lane: a higher-degree rotation up to 0.25 rad per substep, the exact wrap + sincos beyond).
// EVENTS: 1 = rare events behind a wave-uniform branch, 0 = none handled — only the wave mask of lanes that WOULD need it
// is returned (the caller rolls the substeps back), 2 = the event arithmetic inline on every call.
template <class F, bool CHECK_SPIN = true, int EVENTS = 1>
__device__ __forceinline__ uint64_t substep_fast_rot(State<F>& st, F uK, float t, const Params& p, const EnvConst& e) {
  constexpr int W = Width<F>::value;
  F th1, w1, x1, v1;
  const F d = st.w * splat<F>(t);
  ode_euler_fast<F>(st, uK, t, p, e, th1, w1, x1, v1);
  F cd, sd;
  rot_pair_lo<F>(d, cd, sd);
  F c1 = fma_(st.c, cd, -(st.s * sd));
  F s1 = fma_(st.s, cd, st.c * sd);
  uint64_t fired = 0;
#pragma unroll
  for (int i = 0; i < W; ++i) {
    fired |= __builtin_amdgcn_fcmpf(__builtin_fabsf(get(x1, i)), p.THL, 3);                        // 3 = ordered >=
    if constexpr (CHECK_SPIN) fired |= __builtin_amdgcn_fcmpf(__builtin_fabsf(get(d, i)), ROT_LIMIT_LO, 2);   // 2 = ordered >
  }
  if (EVENTS == 2 || (EVENTS == 1 && __builtin_expect(fired != 0, 0))) {
    CPMPPI_DBG(4, 1);
    // plain bounces of lanes inside the rotation range: masked, all lanes at once; everything else per lane (deep)
    F m;
    bool deep = false;
#pragma unroll
    for (int i = 0; i < W; ++i) {
      const bool hit = __builtin_fabsf(get(x1, i)) >= p.THL;
      const bool spin = CHECK_SPIN && __builtin_fabsf(get(d, i)) > ROT_LIMIT_LO;
      put(m, i, (hit && !spin) ? 1.0f : 0.0f);
      deep |= spin;
    }
    const F th0 = th1, w0 = w1, x0 = x1, v0 = v1, c0 = c1, s0 = s1;
    const F dl = bounce_masked<F>(m, c1, th1, w1, x1, v1, t, e.inv_halfL);
    F cdn, sdn;
    rot_pair<F>(dl, cdn, sdn);
    rotate_pair<F>(c1, s1, cdn, sdn);
#pragma unroll
    for (int i = 0; i < W; ++i) deep |= __builtin_fabsf(get(dl, i)) > ROT_LIMIT;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(deep) != 0, 0)) {
#pragma unroll
      for (int i = 0; i < W; ++i) {
        const bool spin = CHECK_SPIN && __builtin_fabsf(get(d, i)) > ROT_LIMIT_LO;
        if (spin || __builtin_fabsf(get(dl, i)) > ROT_LIMIT) {
          float thi = get(th0, i), wi = get(w0, i), xi = get(x0, i), vi = get(v0, i), ci = get(c0, i), si = get(s0, i);
          rare_lane(thi, wi, xi, vi, ci, si, !spin, get(st.c, i), get(st.s, i), get(d, i), t, p.THL, e.inv_halfL);
          put(s1, i, si); put(c1, i, ci);
          put(th1, i, thi); put(w1, i, wi); put(x1, i, xi); put(v1, i, vi);
        }
      }
    }
  }
  st.th = th1; st.w = w1; st.x = x1; st.v = v1; st.c = c1; st.s = s1;
  return fired;
}

#ifndef CPMPPI_EVENTFUL_LAST_INLINE
#define CPMPPI_EVENTFUL_LAST_INLINE 1
#endif
#ifndef CPMPPI_LATENCY_UNROLL
#define CPMPPI_LATENCY_UNROLL 1
#endif
#ifndef CPMPPI_ROLLBACK
#define CPMPPI_ROLLBACK 1       // packed builds, quadratic_boundary_grad_minimal (control_step_fast): 0 = an edge test per substep,
#endif                          // 1 = one per three substeps, the triple redone on an event (A/B switch)
#ifndef CPMPPI_INCR_ROT
#define CPMPPI_INCR_ROT 1       // packed path: (cos d, sin d) of d = w t advanced by d' - d = angleDD t^2 instead of re-evaluated
#endif

// Intermediate substep of the packed path with the rotation's (cos d, sin d) carried along: d = w t changes by
// eps = angleDD t^2 (up to ~1e-3 for a fast-spinning pole) per substep, so
//     (cd, sd) <- (cd - eps sd - eps^2/2, sd + eps cd)
// replaces the two Taylor polynomials (4 instead of 7 instructions; truncation O(eps^3) in cd, eps^2 sd / 2 in sd:
// <= 2e-7 at the range limit |sd| = 0.25, <= 1e-8 for ordinary angular velocities).  The pair is seeded from the
// degree-7/6 polynomials at every control step, and the control step's last substep re-synchronises (cos, sin) exactly
// as before.  Rare lanes are handled PER LANE in the wave-uniform cold branch (rare_lane) — a lane that hits the track
// edge bounces with the cosine it already carries, is rotated on by its new angular velocity and re-seeds its pair; a
// lane whose |w t| exceeded ROT_LIMIT at the start of the control step or after a bounce (`beyond`: its edge limit
// `xlim` is set to -1, so the one comparison per lane covers both events) gets the exact sincos on every substep — so a
// rollout's arithmetic never depends on what its wave partners do, and a wave that meets a rare lane pays about one
// extra substep for it (no libm call anywhere on the FAST path: with ONE wave per SIMD the slowest wave sets the
// kernel's time, and the first version's libm cosf + exact sincos per event made launches of the C3 / C4 size take
// between 80 and 180 us depending on the noise drawn).  The one-rollout-per-lane path keeps the per-substep
// polynomials: it is bound by the LATENCY of a single wave's dependency chain, and carrying the pair makes the next
// rotation wait for this substep's angleDD (measured: 79 us instead of 67 us per single-env step).
// BOUNCY = false: the event block sits behind a wave-uniform branch (cold).  BOUNCY = true: the same arithmetic inline,
// evaluated every substep under its mask — the loop a wave switches to for the rest of a control step once one of its
// lanes has bounced: a rollout caught beyond the edge bounces on EVERY substep, and behind the branch each of those
// costs ~500 cycles (exec-mask and SGPR shuffling around 45 instructions) against ~300 for the whole substep; inline
// and scheduled with the rest it costs ~100.  Returns whether an event occurred (wave-uniform).
// `check` (wave-uniform): false = the two compares and everything behind them are skipped with one scalar branch (no caller
// passes false at present: proving a control step clear of the edge beforehand was measured slower).
template <class F, bool BOUNCY, bool MASK_ONLY = false>
__device__ __forceinline__ uint64_t substep_fast_rot_carried(State<F>& st, F uK, float t, const Params& p, const EnvConst& e,
                                                             F& cd, F& sd, F& xlim, bool check = true, float* xmax = nullptr) {
  constexpr int W = Width<F>::value;
  F th1, w1, x1, v1, aDD;
  ode_euler_fast<F>(st, uK, t, p, e, th1, w1, x1, v1, &aDD);
  F c1 = fma_(st.c, cd, -(st.s * sd));
  F s1 = fma_(st.s, cd, st.c * sd);
  const F eps = aDD * splat<F>(t * t);
  F cd1 = fma_(-eps, fma_(eps, splat<F>(0.5f), sd), cd);            // cd - eps sd - eps^2/2  (cd ~ 1: the eps^2 term matters)
  F sd1 = fma_(cd, eps, sd);                                         // sd + eps cd
  uint64_t fired = 0;                                               // wave mask (a scalar register pair; the loops' exit tests read it)
  if (xmax != nullptr) {
    // (MASK_ONLY callers that test once per control step: the lane's largest |x| so far, one v_max3 instead of two compares)
    if constexpr (W == 2) *xmax = max3_abs2_(*xmax, get(x1, 0), get(x1, 1));
  } else if (check) {
#pragma unroll
    for (int i = 0; i < W; ++i)                                     // edge, or a lane flagged `beyond`  (3 = ordered >=)
      fired |= __builtin_amdgcn_fcmpf(__builtin_fabsf(get(x1, i)), get(xlim, i), 3);
  }
  if (!MASK_ONLY && (BOUNCY || __builtin_expect(fired != 0, 0))) {
    if (!BOUNCY) CPMPPI_DBG(0, 1);
    // plain bounces (lanes inside the rotation range): masked, both rollouts of all lanes at once — about one substep's
    // worth of packed instructions; lanes flagged `beyond`, or thrown beyond the range by this bounce, per lane (deep)
    F m;
    bool deep = false;
#pragma unroll
    for (int i = 0; i < W; ++i) {
      const bool hit = __builtin_fabsf(get(x1, i)) >= p.THL, beyond = get(xlim, i) < 0.0f;
      put(m, i, (hit && !beyond) ? 1.0f : 0.0f);
      deep |= beyond;
    }
    const F th0 = th1, w0 = w1, x0 = x1, v0 = v1, c0 = c1, s0 = s1;
    const F dl = bounce_masked<F>(m, c1, th1, w1, x1, v1, t, e.inv_halfL);
    F cdn, sdn;
    rot_pair<F>(dl, cdn, sdn);                       // lanes that did not bounce: dl = 0, the identity
    rotate_pair<F>(c1, s1, cdn, sdn);
#pragma unroll
    for (int i = 0; i < W; ++i) {
      if (get(m, i) != 0.0f) { put(cd1, i, get(cdn, i)); put(sd1, i, get(sdn, i)); }   // the pair of the new angular velocity
      deep |= __builtin_fabsf(get(dl, i)) > ROT_LIMIT;
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(deep) != 0, 0)) {
#pragma unroll
      for (int i = 0; i < W; ++i) {
        const bool beyond = get(xlim, i) < 0.0f;
        if (beyond || __builtin_fabsf(get(dl, i)) > ROT_LIMIT) {
          float thi = get(th0, i), wi = get(w0, i), xi = get(x0, i), vi = get(v0, i), ci = get(c0, i), si = get(s0, i);
          rare_lane(thi, wi, xi, vi, ci, si, !beyond, get(st.c, i), get(st.s, i), ROT_LIMIT + 1.0f, t, p.THL, e.inv_halfL);
          put(s1, i, si); put(c1, i, ci);
          put(th1, i, thi); put(w1, i, wi); put(x1, i, xi); put(v1, i, vi);
          put(xlim, i, -1.0f);                        // exact on every substep until the next control step re-tests
        }
      }
    }
  }
  cd = cd1; sd = sd1;
  st.th = th1; st.w = w1; st.x = x1; st.v = v1; st.c = c1; st.s = s1;
  return fired;
}

// One control step of S substeps under a held control (FAST).
// Returns whether any lane of the wave ends the step with |x| >= nearlim (wave-uniform; always true where it is
// not tracked): the caller's next stage evaluates the boundary cost only then.
// SPIN_BRANCH (packed mapping, launches with several waves per SIMD): the once-per-control-step test "does a lane spin beyond
// the rotation range?" is ONE v_max of the lane's two |w| and one compare behind which a practically never taken wave-uniform
// branch flags the lanes, instead of a compare + select per rollout (two vector instructions per control step fewer).  Not for
// launches of one wave per SIMD: there the compare -> scalar-branch hand-over sits on the lone wave's critical path.
// ROLLBACK (packed builds, the reference's intermediate_steps = 10): the nine intermediate substeps run WITHOUT an event test of
// their own - each folds its two |x| into the lane's running maximum with one v_max3 instead of two compares, an s_or and a branch
// - as three straight-line triples, each on a fresh copy of the state (SSA renaming: no register moves), and ONE compare after
// substeps 3, 6 and 9 decides: no rollout of the wave reached the edge -> the copy is the state (the common case: 9 x 27 + 3
// vector instructions instead of 9 x 29); otherwise the triple is discarded and the loop with the per-substep test and the event
// arithmetic integrates from that triple's entry state to the end of the control step.  A rollout's arithmetic is the same on
// both routes (bit-identical results with the switch off).  Measured first with one test per control step: an event then wastes
// the whole step and SQ_INSTS_VALU did not move (profiles/HISTORY.md).
// `at_edge` (in/out; required with SPIN_BRANCH): a wave one of whose rollouts ENDED the previous control step at or beyond the edge
// does not speculate - a rollout caught there bounces on every substep, for tens of control steps (see bounce_masked).
template <class F, bool QUIET_UNROLL = false, bool SPIN_BRANCH = false, bool ROLLBACK = false>
__device__ __forceinline__ bool control_step_fast(State<F>& st, F uK, uint32_t S, float t, const Params& p,
                                                  const EnvConst& e, float nearlim, unsigned* sec = nullptr,
                                                  bool* at_edge = nullptr) {
  // `at_edge` (out, wave-uniform; the phased mid-size build passes it): does a rollout of this wave END the control step
  // at or beyond the track edge, or spin beyond the rotation range?  It bounces on the very next substep - and, caught
  // beyond the edge, on every one after it (cartpole_equations.py:341-347 flips v whichever way it points) - so the caller
  // integrates the next control step with the event arithmetic inline (control_step_fast_eventful).
#if CPMPPI_ROTATE && CPMPPI_HOIST_SPIN
  if constexpr (Width<F>::value == 1) {
    // one rollout per lane is the small-launch (latency-bound) mapping: there the per-substep test, which overlaps with
    // the arithmetic, is faster than the shorter loop with its test at the head (single env: 70 us vs 80 us)
    // (three substeps per loop iteration without event handling, under a rollback, as in the packed mid-size build, was
    // measured here twice: with three compare pairs or-ed (round 2: single env 63 -> 72 us) and with ONE test on v_max3 of
    // the positions and of the rotation angles (round 3: 56.6 -> 58.3 us; with section stamps: 1730 -> 1850 cycles for the
    // nine intermediate substeps) - this mapping gains nothing from it.  Nor from instruction order: the substep written
    // as ONE asm block with the links of the loop-carried chain spaced apart (tools/dev/lone_wave.hip, bit-identical) runs
    // at 61 ns against the compiler's 64 - a lone wave issues these three-operand instructions at ~4.5 cycles each
    // whatever their order, so only fewer instructions would help)
    if (sec) CPMPPI_SEC(sec, 2, st);
#if CPMPPI_LATENCY_UNROLL
    if (S == 10u) {
      // the reference's intermediate_steps = 10 as straight-line code: a wave that has its SIMD to itself pays ~50 cycles
      // for every TAKEN branch (the instruction buffer refills from the cache; how many depends on where the target
      // falls, which is why this build's times moved by 10-25 % with unrelated changes of code layout), and the substep
      // loop's back edge is one per 190-cycle substep.  Unrolled, the nine substeps fall through their untaken event
      // branches: single env 56.3 -> 50.0 us, 256 x 20 27.9 -> 25.4 us, 3500 x 35 44.7 -> 39.1 us, 64 envs 62 -> 56 us.
      // (In this straight-line form too, one test per three substeps - rollback as in the mid-size build - is slower:
      // 49.9 -> 51.2 us.  The untaken event branch costs less than the rollback's copies.  Round 4 repeated it in the copy-free
      // form the packed builds now use - fresh State per triple, two running maxima instead of the two compares, one compare pair
      // and branch per triple: 48.9 -> 51.8 us, 256 x 20 24.7 -> 25.9, 64 envs 52.6 -> 56.5.  Not for one rollout per lane.)
#pragma unroll
      for (int sub = 0; sub < 9; ++sub) substep_fast_rot<F>(st, uK, t, p, e);
    } else
#endif
    {
      for (uint32_t sub = 0; sub + 1 < S; ++sub) substep_fast_rot<F>(st, uK, t, p, e);
    }
    if (sec) CPMPPI_SEC(sec, 3, st);
    const bool near_one = substep_fast<F, true, (CPMPPI_LATENCY_NEAR != 0)>(st, uK, t, p, e, nearlim);
    if (sec) CPMPPI_SEC(sec, 4, st);
    return near_one;
  }
  // The seed needs |w t| <= ROT_LIMIT_SEED.  Tested once per control step: without a bounce w cannot leave the range within
  // one control step by more than the polynomials' margin, and a lane that bounces is re-tested.  Lanes beyond the
  // range are flagged and take the exact sincos on every substep.
#if CPMPPI_INCR_ROT
  bool check = true;
  F xlim = splat<F>(p.THL);
  float xmax = 0.0f;                            // ROLLBACK: the lane's largest |x| over the substeps not yet tested
  const float wlim = e.wlim;                    // = ROT_LIMIT_SEED / t: |w t| > the seed's range as one compare with a free abs modifier per lane
  uint64_t spinning = 0;      // wave mask of lanes beyond the rotation range: the same compare as the select's (one v_cmp)
  if constexpr (SPIN_BRANCH && Width<F>::value == 2) {
    const float wmax = max_abs2_(get(st.w, 0), get(st.w, 1));
    spinning = __builtin_amdgcn_fcmpf(wmax, wlim, 2);              // 2 = ordered >  (a NaN counts as within, as in the select form)
    // (ROLLBACK: the lanes are flagged BEHIND the triples - only the loop with the per-substep test reads xlim, and a wave with a
    // spinning lane never enters the triples, so st.w there is still the entry value: the common path no longer carries the
    // v_mov_b64 that formed xlim ahead of the branch)
    if (!ROLLBACK && __builtin_expect(spinning != 0, 0)) {
#pragma unroll
      for (int i = 0; i < Width<F>::value; ++i) put(xlim, i, !(__builtin_fabsf(get(st.w, i)) > wlim) ? p.THL : -1.0f);
    }
  } else if (check) {
#pragma unroll
    for (int i = 0; i < Width<F>::value; ++i) {
      const bool within = !(__builtin_fabsf(get(st.w, i)) > wlim);
      put(xlim, i, within ? p.THL : -1.0f);
      // (the select's own compare, in the sense the compiler emits it - v_cmp_ngt - so that no second compare is needed)
      if (at_edge != nullptr) spinning |= ~__builtin_amdgcn_ballot_w64(within) & __builtin_amdgcn_ballot_w64(true);
      if constexpr (ROLLBACK) xmax = within ? xmax : INFINITY;     // (no branch on the spin test here: such a lane fails the first edge test)
    }
  }
  F cd, sd;
  if constexpr (CPMPPI_SEED_LO != 0) rot_pair_lo<F>(st.w * splat<F>(t), cd, sd);
  else rot_pair<F>(st.w * splat<F>(t), cd, sd);
  uint32_t left = S - 1u;                         // intermediate substeps the loop below still has to integrate
  if constexpr (ROLLBACK && Width<F>::value == 2) {
    // three tests per control step (after substeps 3, 6, 9; the reference's intermediate_steps = 10 only): an event discards at
    // most one triple, and the loop below takes over from that triple's entry state
    // (the phased mid-size build's quiet loop - no SPIN_BRANCH - enters with *at_edge false; a lane beyond the rotation range
    // carries xmax = inf into the first test)
    if (__builtin_expect(S == 10u && (!SPIN_BRANCH || (spinning == 0 && !*at_edge)), 1)) {
      State<F> a = st;
      F cda = cd, sda = sd;
#pragma unroll
      for (int sub = 0; sub < 3; ++sub) substep_fast_rot_carried<F, false, true>(a, uK, t, p, e, cda, sda, xlim, true, &xmax);
      if (__builtin_expect(__builtin_amdgcn_fcmpf(xmax, p.THL, 3) == 0, 1)) {         // 3 = ordered >=
        State<F> b = a;
        F cdb = cda, sdb = sda;
#pragma unroll
        for (int sub = 0; sub < 3; ++sub) substep_fast_rot_carried<F, false, true>(b, uK, t, p, e, cdb, sdb, xlim, true, &xmax);
        if (__builtin_expect(__builtin_amdgcn_fcmpf(xmax, p.THL, 3) == 0, 1)) {
          State<F> c = b;
          F cdc = cdb, sdc = sdb;
#pragma unroll
          for (int sub = 0; sub < 3; ++sub) substep_fast_rot_carried<F, false, true>(c, uK, t, p, e, cdc, sdc, xlim, true, &xmax);
          if (__builtin_expect(__builtin_amdgcn_fcmpf(xmax, p.THL, 3) == 0, 1)) {
            st = c;
            return substep_fast<F>(st, uK, t, p, e, nearlim, true, at_edge);
          }
          st = b; cd = cdb; sd = sdb; left = 3u;
        } else {
          st = a; cd = cda; sd = sda; left = 6u;
        }
      }
      asm volatile("; rollback: a rollout of this wave reached the track edge within the last three substeps");
    }
    if constexpr (SPIN_BRANCH) {
      if (__builtin_expect(spinning != 0, 0)) {
#pragma unroll
        for (int i = 0; i < Width<F>::value; ++i) put(xlim, i, !(__builtin_fabsf(get(st.w, i)) > wlim) ? p.THL : -1.0f);
      }
    }
  }
  // (Rounds 2 and 3 ran the packed mid-size build on three substeps at a time without event handling, under a rollback -
  // one v_max3 test per triple, the discarded triple redone substep by substep with the event arithmetic inline - until
  // section stamps showed that build's median wave 17 % slower per control step than this plain loop, in every section:
  // DESIGN.md §4.  The phased horizon loop replaced it; the code is in the history.)
  {
    if (sec) { asm volatile("" : "+v"(cd), "+v"(sd), "+v"(xlim)); CPMPPI_SEC(sec, 2, st); }
    if (QUIET_UNROLL && !ROLLBACK && S == 10u) {
      // the quiet control step of the phased mid-size build in a launch of one wave per SIMD, the reference's
      // intermediate_steps = 10: the nine substeps as straight-line code - a lone wave pays ~50 cycles per taken branch
      // (see the one-rollout-per-lane mapping above), and the loop's back edge is one per substep
#pragma unroll
      for (int sub = 0; sub < 9; ++sub) substep_fast_rot_carried<F, false>(st, uK, t, p, e, cd, sd, xlim, check);
    } else if (QUIET_UNROLL && ROLLBACK && S == 10u) {
      // (the lone-wave build after a discarded triple: 9, 6 or 3 substeps left, whole triples as straight-line code)
      for (; left != 0u; left -= 3u) {
#pragma unroll
        for (int sub = 0; sub < 3; ++sub) substep_fast_rot_carried<F, false>(st, uK, t, p, e, cd, sd, xlim, check);
      }
    } else {
      for (; left != 0u; --left) substep_fast_rot_carried<F, false>(st, uK, t, p, e, cd, sd, xlim, check);
    }
    if (sec) CPMPPI_SEC(sec, 3, st);
  }
  const bool near_end = substep_fast<F>(st, uK, t, p, e, nearlim, check, at_edge) && check;
  // phased horizon loop: a wave with a lane beyond the rotation range (it takes the event path on every substep, ~500
  // cycles each behind the branch) goes to the loop with the event arithmetic inline like one with a rollout at the edge
  if (at_edge != nullptr) *at_edge = *at_edge || spinning != 0;
  if (sec) CPMPPI_SEC(sec, 4, st);
  return near_end;
#else
  bool spin = false;
#pragma unroll
  for (int i = 0; i < Width<F>::value; ++i) spin |= __builtin_fabsf(get(st.w, i)) * t > 0.1f;
  bool exact = __builtin_amdgcn_ballot_w64(spin) != 0;
  uint32_t sub = 0;
  if (__builtin_expect(!exact, 1)) {
    while (sub + 1 < S) {
      ++sub;
      if (__builtin_expect(substep_fast_rot<F, false>(st, uK, t, p, e), 0)) break;
    }
  }
  for (; sub < S; ++sub) substep_fast<F>(st, uK, t, p, e, p.THL);       // the last substep always; all remaining after a bounce
  return true;
#endif
#elif CPMPPI_ROTATE
  for (uint32_t sub = 0; sub + 1 < S; ++sub) substep_fast_rot<F>(st, uK, t, p, e);
  return substep_fast<F>(st, uK, t, p, e, nearlim);
#else
  for (uint32_t sub = 0; sub < S; ++sub) substep_fast<F>(st, uK, t, p, e, p.THL);
  return true;
#endif
}
// A control step of a wave one of whose rollouts ENDED the previous step at or beyond the track edge (`at_edge`): the
// event arithmetic inline on every intermediate substep, no test, no speculation (see control_step_fast).  Used by the
// phased horizon loop (cpmppi_rollout.hpp), which keeps this code out of the loop the quiet control steps run in.
template <class F, bool UNROLL = false>
__device__ __forceinline__ bool control_step_fast_eventful(State<F>& st, F uK, uint32_t S, float t, const Params& p,
                                                           const EnvConst& e, float nearlim, bool* at_edge) {
  F xlim = splat<F>(p.THL);
  const float wlim = e.wlim;
  uint64_t spinning = 0;
#pragma unroll
  for (int i = 0; i < Width<F>::value; ++i) {
    const bool within = !(__builtin_fabsf(get(st.w, i)) > wlim);
    put(xlim, i, within ? p.THL : -1.0f);
    spinning |= ~__builtin_amdgcn_ballot_w64(within) & __builtin_amdgcn_ballot_w64(true);
  }
  F cd, sd;
  if constexpr (CPMPPI_SEED_LO != 0) rot_pair_lo<F>(st.w * splat<F>(t), cd, sd);
  else rot_pair<F>(st.w * splat<F>(t), cd, sd);
  if (UNROLL && S == 10u) {                      // (launches of one wave per SIMD: no taken branch between the substeps)
#pragma unroll
    for (int sub = 0; sub < 9; ++sub) substep_fast_rot_carried<F, true>(st, uK, t, p, e, cd, sd, xlim);
  } else {
    uint32_t left = S - 1u;
    while (left >= 3u) {
      substep_fast_rot_carried<F, true>(st, uK, t, p, e, cd, sd, xlim);
      substep_fast_rot_carried<F, true>(st, uK, t, p, e, cd, sd, xlim);
      substep_fast_rot_carried<F, true>(st, uK, t, p, e, cd, sd, xlim);
      left -= 3u;
    }
    for (; left != 0u; --left) substep_fast_rot_carried<F, true>(st, uK, t, p, e, cd, sd, xlim);
  }
  const bool near_end = substep_fast<F, true, true, (CPMPPI_EVENTFUL_LAST_INLINE != 0)>(st, uK, t, p, e, nearlim, true, at_edge);
  *at_edge = *at_edge || spinning != 0;          // (stays in this loop while the pole keeps spinning)
  return near_end;
}

// ------------------------------------------------------------------------------------------------------------------
// The OTHER in-tree ODE predictor: predictor_type "ODE" (SI_Toolkit_ASF/config_predictors.yml:22-26 - what the shipped
// config_controllers.yml:3,14 name as the mpc controllers' predictor_specification).  next_state_predictor_ODE
// (SI_Toolkit_ASF/ToolkitCustomization/predictors_customization.py:25-69) -> CartPoleEquations.cartpole_fine_integration
// (CartPole/cartpole_equations.py:181-259): per substep the same _cartpole_ode (:232), then EULER-CROMER (:293-304:
// velocities first, angle and position advance by the NEW velocities), NO edge bounce (:241-243 is commented out in the
// reference), cos / sin of the integrated angle (:245-246) and angle = atan2(sin, cos) (:248, 307-308).
enum : int { PREDICTOR_ODE_V0 = 0, PREDICTOR_ODE = 1 };

// PRECISE: the reference's operand grouping with IEEE divides, libm cos / sin, no FMA contraction.  The angle this predictor
// re-derives on every substep, atan2(sin, cos), is evaluated in double and rounded once: the device's atan2f is a ~2 ulp
// function, and that noise - re-entering the state 500 times per rollout - moved a few rollouts of a full-width C4 launch
// (|w| = 11 rad/s) 2e-4 from the oracle where the oracle's own float32 realisations scatter by 2e-5; FAST, which never goes
// through atan2, was inside the band all along.
__device__ __forceinline__ void substep_precise_cromer(State<float>& st, float u, float t, const Params& p,
                                                       const EnvConst& e) {
#pragma clang fp contract(off)
  float aDD, xDD;
  ode_precise(st.c, st.s, st.w, st.v, u, p, e, aDD, xDD);
  const float w1 = st.w + aDD * t;
  const float v1 = st.v + xDD * t;
  const float th1 = st.th + w1 * t;
  const float x1 = st.x + v1 * t;
  const float c1 = cosf(th1), s1 = sinf(th1);
  st.th = (float)atan2((double)s1, (double)c1);
  st.w = w1; st.c = c1; st.s = s1; st.x = x1; st.v = v1;
}

// FAST: _cartpole_ode with the folded constants of ode_euler_fast (same formulas, same instruction sequence) followed by
// the Euler-Cromer update.  Outputs the un-wrapped new state and angleDD.
template <class F>
__device__ __forceinline__ void ode_cromer_fast(const State<F>& st, F uK, float t, const Params& p, const EnvConst& e,
                                                F& th1, F& w1, F& x1, F& v1, F& aDD) {
  const F c = st.c, s = st.s, w = st.w, v = st.v;
  const F A = fma_(-(c * splat<F>(p.m_pole)), c, splat<F>(e.kp1_mt));
  const F t1 = fma_(splat<F>(e.mg), s, -(w * splat<F>(e.JinvLh)));
  F num = fma_(c, t1, uK);
  num = fma_(-((w * w) * splat<F>(e.kmLh)), s, num);
  num = fma_(splat<F>(-e.kM), v, num);
  const F xDD = num * rcp_(A);
  aDD = fma_(xDD * c, splat<F>(e.inv_kLh), t1 * splat<F>(e.t1_i));
  const F tt = splat<F>(t);
  w1 = fma_(aDD, tt, w);
  v1 = fma_(xDD, tt, v);
  th1 = fma_(w1, tt, st.th);
  x1 = fma_(v1, tt, st.x);
}

// One control step of S Euler-Cromer substeps under a held control (FAST).  As in control_step_fast the angle is left
// un-wrapped inside the control step and (cos, sin) advance by ROTATION: a substep moves the angle by d' = w' t with the
// NEW angular velocity w' = w + angleDD t, i.e. by the previous substep's rotation angle plus eps = angleDD t^2, so the
// carried pair (cos d, sin d) - seeded from the degree-7/6 polynomials of d = w t once per control step - is advanced by
// eps FIRST and (cos, sin) are then rotated by it (substep_fast_rot_carried rotates first: simultaneous Euler moves the
// angle by the OLD velocity).  The control step's last substep re-synchronises with the exact wrap + polynomial sincos;
// theta - 2 pi rint(theta / 2 pi) is the value atan2(sin, cos) returns up to rounding (both in [-pi, pi]).  No edge
// test anywhere: this predictor does not bounce.  The carried pair is second order in eps: its truncation, eps^2 |sin d| / 2
// per substep, stays below 1e-7 while |d| = |w t| <= ~0.1 (the centrifugal term makes eps ~ 0.11 d^2 at worst: 0.006 d^5);
// a lane that starts a control step beyond CROMER_CARRY_LIMIT (45 rad/s at t = 2 ms - a pole released from rest tops out
// near 20) takes the exact wrap + sincos on every substep instead - per lane, so a rollout's arithmetic does not depend on
// its wave partners; the branch is wave-uniform and practically never taken.  (Found by the reference-generated "fastspin"
// fixture, 150 rad/s decaying through 100: with the limit at the polynomials' 0.25 every rollout left the band.)
constexpr float CROMER_CARRY_LIMIT = 0.09f;
template <class F, bool UNROLL = false>
__device__ __forceinline__ void control_step_cromer_fast(State<F>& st, F uK, uint32_t S, float t, const Params& p,
                                                         const EnvConst& e) {
  constexpr int W = Width<F>::value;
  const float wlim = CROMER_CARRY_LIMIT / t;
  bool beyond[W];
  bool any = false;
#pragma unroll
  for (int i = 0; i < W; ++i) { beyond[i] = __builtin_fabsf(get(st.w, i)) > wlim; any |= beyond[i]; }
  const bool any_beyond = __builtin_amdgcn_ballot_w64(any) != 0;
  F cd, sd;
  rot_pair<F>(st.w * splat<F>(t), cd, sd);
  const F tt2 = splat<F>(t * t);
  auto substep = [&]() __attribute__((always_inline)) {
    F th1, w1, x1, v1, aDD;
    ode_cromer_fast<F>(st, uK, t, p, e, th1, w1, x1, v1, aDD);
    const F eps = aDD * tt2;
    const F cd1 = fma_(-eps, fma_(eps, splat<F>(0.5f), sd), cd);     // cd - eps sd - eps^2/2
    const F sd1 = fma_(cd, eps, sd);                                  // sd + eps cd
    F c1 = fma_(st.c, cd1, -(st.s * sd1));
    F s1 = fma_(st.s, cd1, st.c * sd1);
    if (__builtin_expect(any_beyond, 0)) {
      const F thw = wrap_rint<F>(th1);
      F se, ce;
      sincos_pi_half<F>(thw, se, ce);
#pragma unroll
      for (int i = 0; i < W; ++i)
        if (beyond[i]) { put(th1, i, get(thw, i)); put(c1, i, get(ce, i)); put(s1, i, get(se, i)); }
    }
    cd = cd1; sd = sd1;
    st.th = th1; st.w = w1; st.x = x1; st.v = v1; st.c = c1; st.s = s1;
  };
  if (UNROLL && S == 10u) {                     // (a lone wave pays ~50 cycles per taken branch: control_step_fast)
#pragma unroll
    for (int sub = 0; sub < 9; ++sub) substep();
  } else {
    for (uint32_t sub = 0; sub + 1 < S; ++sub) substep();
  }
  F th1, w1, x1, v1, aDD;
  ode_cromer_fast<F>(st, uK, t, p, e, th1, w1, x1, v1, aDD);
  th1 = wrap_rint<F>(th1);
  st.th = th1; st.w = w1; st.x = x1; st.v = v1;
  sincos_pi_half<F>(th1, st.s, st.c);
}

// ------------------------------------------------------------------------------------------------------------------
// Stage / terminal costs (generic over float / float2).  `x_t` target position, `te` target equilibrium, `u` the
// control applied at this stage.
// quadratic_boundary_grad_minimal.py:64-126; w = {dd, db, ep, ekp, cc, R, permissible_track_fraction}
// x / c for a wave-uniform c: the IEEE divide (PRECISE) or a multiply by the float32-rounded reciprocal (FAST; differs
// from the divide by <= 1 ulp).
template <bool FAST, class F>
__device__ __forceinline__ F div_uniform(F x, float c) {
  if constexpr (FAST) return x * splat<F>(1.0f / c);
  else return x / splat<F>(c);
}

// near = false: the caller knows |x| < permissible_track_fraction * THL for every lane, i.e. the boundary term is exactly
// zero and dd + 0 == dd: it is left out (wave-uniform branch), bit-identical.
// FAST folds the wave-uniform factors of three terms (QbgmFolded, formed once per kernel in double): dd = (x - x*)^2 *
// [w_dd / (2 THL)^2], cc = u^2 * [R w_cc], 1 - cos * te as one FMA - three instructions fewer per stage, each term within
// 2 ulp of the reference's grouping (PRECISE keeps that grouping operation for operation).
struct QbgmFolded {
  float c_dd, c_cc, neg_te;
  // stage_qbgm_acc (FAST, the rollout kernel): every term's weight with the horizon aggregation's scale (1 for sum, 1/(H+1)
  // for mean) folded in, and the MPPI correction's coefficients (controller_mppi_cartpole.py:261-263: cc_weight (0.5 (1 - 1/NU) R du^2 + R u du + 0.5 R u^2))
  float a_dd, a_ep, a_ekp, a_db, db_lim;   // db: b^2 w_db = (|x| - lim)^2 a_db for |x| > lim = permissible_track_fraction THL
  float a_u2;                              // u_run^2: w_cc R scale, + 0.5 cc_weight R when the correction takes u_run
  float k_a, k_b_run, k_b_nom;             // du (k_a du + k_b_run u_run + k_b_nom u_nom): one of the two k_b is zero
  float k_c_nom;                           // 0.5 cc_weight R when the correction takes u_nom: the (rollout-independent) term u_nom^2
};
// (`_lane`: every lane its own `te` - the per-env fold kernel; the rollout kernel's in-kernel form pins the fields to SGPRs)
__device__ __forceinline__ QbgmFolded make_qbgm_folded_lane(const Params& p, float te) {
  QbgmFolded f;
  const double two_thl = 2.0 * (double)p.THL;
  f.c_dd = (float)((double)p.w[0] / (two_thl * two_thl));
  f.c_cc = (float)((double)p.w[5] * (double)p.w[4]);
  f.neg_te = -te;
  const double scale = (p.horizon_reduce == 0u) ? 1.0 : 1.0 / (double)(p.H + 1u);      // CPMPPI_REDUCE_SUM = 0
  const double ptf = (double)p.w[6], span = (1.0 - ptf) * (double)p.THL;
  f.a_dd = (float)(scale * (double)p.w[0] / (two_thl * two_thl));
  f.a_ep = (float)(scale * (double)p.w[2]);
  f.a_ekp = (float)(scale * (double)p.w[3]);
  f.a_db = (float)(scale * (double)p.w[1] / (span * span));
  f.db_lim = (float)(ptf * (double)p.THL);
  const bool run = p.correction_u == 0u;                                                // CPMPPI_CORRECTION_U_RUN = 0
  const double half_r = (double)p.cc_weight * 0.5 * (double)p.R;
  f.a_u2 = (float)(scale * (double)p.w[5] * (double)p.w[4] + (run ? half_r : 0.0));
  f.k_a = (float)((double)p.cc_weight * 0.5 * (1.0 - 1.0 / (double)p.NU) * (double)p.R);
  f.k_b_run = run ? (float)((double)p.cc_weight * (double)p.R) : 0.0f;
  f.k_b_nom = run ? 0.0f : (float)((double)p.cc_weight * (double)p.R);
  f.k_c_nom = run ? 0.0f : (float)half_r;
  return f;
}
__device__ __forceinline__ QbgmFolded make_qbgm_folded(const Params& p, float te) {
  const QbgmFolded c = make_qbgm_folded_lane(p, te);
  QbgmFolded f;
  f.c_dd = uniform_(c.c_dd); f.c_cc = uniform_(c.c_cc); f.neg_te = uniform_(c.neg_te);
  f.a_dd = uniform_(c.a_dd); f.a_ep = uniform_(c.a_ep); f.a_ekp = uniform_(c.a_ekp); f.a_db = uniform_(c.a_db);
  f.db_lim = uniform_(c.db_lim); f.a_u2 = uniform_(c.a_u2); f.k_a = uniform_(c.k_a); f.k_b_run = uniform_(c.k_b_run);
  f.k_b_nom = uniform_(c.k_b_nom); f.k_c_nom = uniform_(c.k_c_nom);
  return f;
}

// Everything the throughput build's rollout kernel derives from (Params, L[env], target_equilibrium[env], s0[env]) alone, formed
// ONCE per env by fold_env_kernel (cpmppi.hip) instead of once per WAVE in the rollout kernel's prologue: ~300 vector instructions
// (double-precision folds, two IEEE divides, libm cosf) per wave of 128 rollouts = 1.5 % of the launch, read back with scalar loads.
// Same device functions, same values.  136 bytes per env.
struct EnvFold {
  EnvConst ec;          // 18 words
  QbgmFolded qf;        // 13 words
  float cos0;           // cosf(s0[0]): the cost plugins take cos(angle), not the stored angle_cos, at stage 0
  float inv_period;     // 1 / period_interpolation_inducing_points (the in-kernel interpolation's slope factor)
  float nearlim;        // min(permissible_track_fraction, 1) * THL: below it quadratic_boundary_grad_minimal's boundary term is zero
};
static_assert(sizeof(EnvFold) == 136, "EnvFold: 34 words per env");

// quadratic_boundary_grad_minimal's stage cost AND the MPPI correction term of one stage added to two running sums (FAST path of
// the rollout kernel).  Same terms as stage_qbgm + mppi_correction; what differs is the grouping: every term is accumulated
// with one FMA (weight folded into the constant) instead of formed, weighted, summed into the stage cost and added to the
// horizon's sum - 13 vector instructions instead of 20 per stage, each term within 2 ulp of the reference's grouping (PRECISE
// keeps that grouping operation for operation).  Two sums (`acc_a`: position and angle terms, `acc_b`: speed, control and
// correction terms) keep the dependent chain of a stage at two FMAs.  `b_nom` = k_b_nom * u_nom of this stage (wave-uniform; zero unless the correction takes
// u_nom); the rollout-independent term k_c_nom u_nom^2 of that mode is added by the caller once per rollout.
// near = false: |x| < permissible_track_fraction * THL for every lane, the boundary term is exactly zero and left out.
template <class F>
__device__ __forceinline__ void stage_qbgm_acc(const QbgmFolded& f, F x, F cosang, F w, F ur, F du, bool nom_mode, float b_nom,
                                               float x_t, bool near, F& acc_a, F& acc_b) {
  const F dx = x - splat<F>(x_t);
  acc_a = fma_(dx * dx, splat<F>(f.a_dd), acc_a);
  const F e1 = fma_(cosang, splat<F>(f.neg_te), splat<F>(1.0f));
  acc_a = fma_(e1 * e1, splat<F>(f.a_ep), acc_a);
  acc_b = fma_(w * w, splat<F>(f.a_ekp), acc_b);
  acc_b = fma_(ur * ur, splat<F>(f.a_u2), acc_b);
  const F lin = fma_(du, splat<F>(f.k_a), ur * splat<F>(f.k_b_run));
  acc_b = fma_(du, lin, acc_b);
  if (__builtin_expect(nom_mode, 0)) {
    asm volatile("; correction term on u_nom");    // (keeps this a branch: the common path carries no operand for it)
    acc_b = fma_(du, splat<F>(b_nom), acc_b);
  }
  if (__builtin_expect(near, 0)) {
    // (the asm statement keeps this a BRANCH: if-converted, the term's eight instructions would run on every stage)
    asm volatile("; quadratic_boundary_grad_minimal: boundary term");
    F over;
#pragma unroll
    for (int i = 0; i < Width<F>::value; ++i) put(over, i, max_abs_(get(x, i), f.db_lim));
    over = over - splat<F>(f.db_lim);
    acc_a = fma_(over * over, splat<F>(f.a_db), acc_a);
  }
}

template <class F, bool FAST = false>
__device__ __forceinline__ F stage_qbgm(const Params& p, F x, F cosang, F w_ang, F u, float x_t, float te, bool near = true,
                                        const QbgmFolded* fold = nullptr) {
#pragma clang fp contract(off)     // every product and sum rounds once, wherever the function is inlined
  const float THL = p.THL;
  F dd, ep, cc;
  if (FAST && fold != nullptr) {
    const F dx = x - splat<F>(x_t);
    dd = (dx * dx) * splat<F>(fold->c_dd);
    const F e1 = fma_(cosang, splat<F>(fold->neg_te), splat<F>(1.0f));
    ep = (e1 * e1) * splat<F>(p.w[2]);
    cc = (u * u) * splat<F>(fold->c_cc);
  } else {
    const F d = div_uniform<FAST, F>(x - splat<F>(x_t), 2.0f * THL);
    dd = (d * d) * splat<F>(p.w[0]);
    const F e1 = splat<F>(1.0f) - cosang * splat<F>(te);
    ep = (e1 * e1) * splat<F>(p.w[2]);
    cc = ((u * u) * splat<F>(p.w[5])) * splat<F>(p.w[4]);
  }
  const F ekp = (w_ang * w_ang) * splat<F>(p.w[3]);
  if (!near) return dd + ep + ekp + cc;
  const float ptf = p.w[6];
  const F ax = abs_(x);
  // indicator(|x| > ptf THL) * b^2 == (max(|x| - ptf THL, 0) / ...)^2: the same value without compare + select
  F over = ax - splat<F>(ptf * THL);
#pragma unroll
  for (int i = 0; i < Width<F>::value; ++i) put(over, i, __builtin_fmaxf(get(over, i), 0.0f));
  const F b = div_uniform<FAST, F>(over, (1.0f - ptf) * THL);
  const F db = (b * b) * splat<F>(p.w[1]);
  return dd + db + ep + ekp + cc;
}

// quadratic_boundary_grad.py:64-232; w = {up: ddq, ddl, db, ep, ekp, cc, ccrc | down: the same 7 | tas_corr_up,
// tas_corr_down, permissible_track_fraction, cos(admissible_angle), R}; u_before = the control of the previous stage
// (previous_input at stage 0); terminal cost zero.
template <class F, bool FAST = false>
__device__ __forceinline__ F stage_qbg(const Params& p, F x, F cosang, F w_ang, F u, F u_before, float x_t, float te) {
#pragma clang fp contract(off)     // every product and sum rounds once, wherever the function is inlined
  const bool up = (te == 1.0f);
  const float* w = p.w + (up ? 0 : 7);
  const float corr = up ? p.w[14] : p.w[15];
  const float ptf = p.w[16], cos_adm = p.w[17], R = p.w[18], THL = p.THL;
  const F d = div_uniform<FAST, F>(x - splat<F>(x_t), 2.0f * THL);
  const F dd_quadratic = (d * d) * splat<F>(w[0]);
  const F dd_linear = abs_(d) * splat<F>(w[1]);
  const F ax = abs_(x);
  F over = ax - splat<F>(ptf * THL);
#pragma unroll
  for (int i = 0; i < Width<F>::value; ++i) put(over, i, __builtin_fmaxf(get(over, i), 0.0f));
  const F b = div_uniform<FAST, F>(over, (1.0f - ptf) * THL);
  const F db = (b * b) * splat<F>(w[2]);
  const F tc = cosang * splat<F>(te);
  const F e2 = splat<F>(2.0f) - tc;
  const F ep = (e2 * e2 - splat<F>(1.0f)) * splat<F>(w[3]);
  const float tas_max = __builtin_fabsf(120.0f * (1.0f + te) / 2.0f + corr);
  const F basic = (splat<F>(1.0f) - tc) * splat<F>(0.5f);
  F scaling;
#pragma unroll
  for (int i = 0; i < Width<F>::value; ++i)
    put(scaling, i, (te * (get(cosang, i) - cos_adm) > 0.0f) ? 0.0f : get(basic, i));
  const F ekp = abs_(w_ang * w_ang - scaling * splat<F>(tas_max)) * splat<F>(w[4]);
  const F cc = ((u * u) * splat<F>(R)) * splat<F>(w[5]);
  const F dc = u - u_before;
  const F ccrc = (dc * dc) * splat<F>(w[6]);
  return dd_linear + dd_quadratic + db + ep + ekp + cc + ccrc;
}

// default.py:23-88; w = {dd, ep, cc, R}.
// The same function serves the plugin's two siblings (p.qb_mode, wave-uniform; w = {dd, ep, cc, R, ccrc}):
//   1  quadratic_boundary.py:26-87 - the track-edge term is 1[|x| > 0.95 THL] 1e9 ((|x| - 0.95 THL) / (0.05 THL))^2 instead of
//      the 1e7 indicator at 0.9 THL, and a control-change-rate term ccrc_weight (u - u_before)^2 joins when the caller hands a
//      previous input over (`with_ccrc`; the reference adds the term only `if previous_input is not None`, :83-85);
//   2  quadratic_boundary_nonconvex.py:27-105 - the same plus the ripple -0.15 (cos(4 2 pi (x - x*) / (2 THL)) - 1) on the
//      position term (restated from the source text: the reference cannot import that module, oracle_np.qb_stage_cost).
// QB = false: compiled without the siblings (the predictor_ODE kernels: the plugins are built for predictor_ODE_v0 only).
template <class F, bool FAST = false, bool QB = true>
__device__ __forceinline__ F stage_default(const Params& p, F x, F cosang, F u, float x_t, float te, F u_before = splat<F>(0.0f),
                                           bool with_ccrc = false) {
#pragma clang fp contract(off)     // every product and sum rounds once, wherever the function is inlined
  const float THL = p.THL;
  const F d = div_uniform<FAST, F>(x - splat<F>(x_t), 2.0f * THL);
  F pos = d * d, edge;
  if (!QB || p.qb_mode == 0u) {
#pragma unroll
    for (int i = 0; i < Width<F>::value; ++i) put(edge, i, (__builtin_fabsf(get(x, i)) > 0.90f * THL) ? 1.0e7f : 0.0f);
  } else {
    const F b = div_uniform<FAST, F>(abs_(x) - splat<F>(0.95f * THL), 0.05f * THL);
    F ind;
#pragma unroll
    for (int i = 0; i < Width<F>::value; ++i) put(ind, i, (__builtin_fabsf(get(x, i)) > 0.95f * THL) ? 1.0e9f : 0.0f);
    edge = ind * (b * b);
    if (p.qb_mode == 2u) {
      const F arg = div_uniform<FAST, F>(splat<F>(25.132741228718345f) * (x - splat<F>(x_t)), 2.0f * THL);
      pos = pos - splat<F>(0.15f) * (cos_(arg) - splat<F>(1.0f));
    }
  }
  const F dd = (pos + edge) * splat<F>(p.w[0]);
  const F e1 = splat<F>(1.0f) - cosang;
  const F ep = ((e1 * e1) * splat<F>(0.25f) * splat<F>(te)) * splat<F>(p.w[1]);
  const F cc = ((u * u) * splat<F>(p.w[3])) * splat<F>(p.w[2]);
  if (QB && with_ccrc) {
    const F dc = u - u_before;
    return dd + ep + cc + (dc * dc) * splat<F>(p.w[4]);
  }
  return dd + ep + cc;
}

// default.py:55-63 and controller_mppi_cartpole.py:278-303 (phi): 10000 * 1[|angle| > 0.2 or |x - x*| > 0.1*THL]
template <class F>
__device__ __forceinline__ F terminal_indicator(const Params& p, F angle, F x, float x_t) {
  F out;
#pragma unroll
  for (int i = 0; i < Width<F>::value; ++i) {
    const bool bad = (__builtin_fabsf(get(angle, i)) > 0.2f) || (__builtin_fabsf(get(x, i) - x_t) > 0.1f * p.THL);
    put(out, i, bad ? 10000.0f : 0.0f);
  }
  return out;
}

// MPPI correction term, algebra of controller_mppi_cartpole.py:261-263
template <class F>
__device__ __forceinline__ F mppi_correction(const Params& p, F u, F du) {
  // cc_weight (0.5 (1 - 1/NU) R du^2 + R u du + 0.5 R u^2)  =  du (a du + b u) + c u^2   with the weight folded in
  const float a = p.cc_weight * (0.5f * (1.0f - 1.0f / p.NU) * p.R), b = p.cc_weight * p.R, c = p.cc_weight * (0.5f * p.R);
  return fma_(u * splat<F>(c), u, du * fma_(du, splat<F>(a), u * splat<F>(b)));
}

// controller_mppi_cartpole.py:227-275 (q); w = {dd, ep, ekp, ekc, cc, ccrc}; u = nominal control of the stage
template <class F, bool FAST = false>
__device__ __forceinline__ F stage_legacy(const Params& p, F x, F cosang, F w_ang, F v, float u, F du, float u_prev,
                                          float x_t) {
#pragma clang fp contract(off)     // every product and sum rounds once, wherever the function is inlined
  const float THL = p.THL;
  const F d = div_uniform<FAST, F>(x - splat<F>(x_t), 2.0f * THL);
  F ind;
#pragma unroll
  for (int i = 0; i < Width<F>::value; ++i) put(ind, i, (__builtin_fabsf(get(x, i)) > 0.95f * THL) ? 1.0e6f : 0.0f);
  const F dd = (d * d + ind) * splat<F>(p.w[0]);
  const F e1 = splat<F>(1.0f) - cosang;
  const F ep = ((e1 * e1) * splat<F>(0.25f)) * splat<F>(p.w[1]);
  const F ekp = (w_ang * w_ang) * splat<F>(p.w[2]);
  const F ekc = (v * v) * splat<F>(p.w[3]);
  const float half_nu = 0.5f * (1.0f - 1.0f / p.NU) * p.R;
  F cc = ((du * du) * splat<F>(half_nu) + du * splat<F>(p.R * u) + splat<F>(0.5f * p.R * (u * u))) * splat<F>(p.w[4]);
  const F ur = splat<F>(u) + du;
#pragma unroll
  for (int i = 0; i < Width<F>::value; ++i)
    if (__builtin_fabsf(get(ur, i)) > 1.0f) put(cc, i, 1.0e5f);
  const F dcr = ur - splat<F>(u_prev);
  const F ccrc = (dcr * dcr) * splat<F>(p.w[5]);
  return dd + ep + ekp + ekc + cc + ccrc;
}

// ------------------------------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al. 2011).  Counter = (rollout, env, knot pair, offset lo),
// key = seed.  One call yields the two standard normals of one Box-Muller pair (knots 2j and 2j+1).
__device__ __forceinline__ void philox4x32_10(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                              uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // (the full 64-bit products: one v_mad_u64_u32 each instead of a v_mul_hi_u32 + v_mul_lo_u32 pair)
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// Two standard normals per Philox block (Box-Muller) on the hardware transcendentals: ln u1 = ln2 * v_log_f32(u1),
// v_sin_f32 / v_cos_f32 take their argument in revolutions, so sin(2 pi u2) is v_sin_f32(u2) (abs. error ~1e-6: this
// is the library's OWN noise stream, the same function feeds the sampler kernel and the in-kernel generation).
__device__ __forceinline__ void philox_normal_pair(uint64_t seed, uint64_t offset, uint32_t env, uint32_t rollout,
                                                   uint32_t pair, float& z0, float& z1) {
  uint32_t c0 = rollout, c1 = env, c2 = pair, c3 = (uint32_t)offset;
  philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32));
  const float u1 = (float)((c0 >> 8) + 1u) * 5.9604644775390625e-8f;   // (0, 1]
  const float u2 = (float)(c1 >> 8) * 5.9604644775390625e-8f;          // [0, 1)
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
  z0 = r * __builtin_amdgcn_cosf(u2);
  z1 = r * __builtin_amdgcn_sinf(u2);
}

// All four words of a Philox block: two Box-Muller pairs = the four consecutive knots 4q .. 4q+3 of (env, rollout).
__device__ __forceinline__ void philox_normal_quad(uint64_t seed, uint64_t offset, uint32_t env, uint32_t rollout,
                                                   uint32_t quad, float z[4]) {
  uint32_t c0 = rollout, c1 = env, c2 = quad, c3 = (uint32_t)offset;
  philox4x32_10(c0, c1, c2, c3, (uint32_t)seed ^ 0x51ed270bu, (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32));
  const float ua = (float)((c0 >> 8) + 1u) * 5.9604644775390625e-8f, ub = (float)(c1 >> 8) * 5.9604644775390625e-8f;
  const float uc = (float)((c2 >> 8) + 1u) * 5.9604644775390625e-8f, ud = (float)(c3 >> 8) * 5.9604644775390625e-8f;
  const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(ua));
  const float rc = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(uc));
  z[0] = ra * __builtin_amdgcn_cosf(ub);
  z[1] = ra * __builtin_amdgcn_sinf(ub);
  z[2] = rc * __builtin_amdgcn_cosf(ud);
  z[3] = rc * __builtin_amdgcn_sinf(ud);
}

// Knot j of (env, rollout) scaled by sigma (random access; sequential consumers keep the other three of the block).
__device__ __forceinline__ float philox_knot(uint64_t seed, uint64_t offset, uint32_t env, uint32_t rollout,
                                             uint32_t j, float sigma) {
  float z[4];
  philox_normal_quad(seed, offset, env, rollout, j >> 2, z);
  const uint32_t s = j & 3u;
  return sigma * (s == 0u ? z[0] : (s == 1u ? z[1] : (s == 2u ? z[2] : z[3])));
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

// The same value with the slope hoisted: slope changes only when the knots change (once per `period` steps).
__device__ __forceinline__ double knot_slope(float z_lo, float z_hi, uint32_t period) {
#pragma clang fp contract(off)      // (else sigma*z of philox_knot is fused into this subtraction and skips a rounding)
  const float diff = z_hi - z_lo;
  return (double)diff / (double)period;
}
// float32 form for the FAST path's own Philox noise: slope rounded to float, then ONE fma per control step.
__device__ __forceinline__ float knot_slope32(float z_lo, float z_hi, float inv_period) {
#pragma clang fp contract(off)      // same reason: the knots must be the rounded float32 values the sampler stores
  const float diff = z_hi - z_lo;
  return diff * inv_period;
}
__device__ __forceinline__ float interp_from_slope32(float slope, float z_lo, uint32_t i) {
  return __builtin_fmaf(slope, (float)i, z_lo);
}
__device__ __forceinline__ float interp_from_slope(double slope, float z_lo, uint32_t i) {
#pragma clang fp contract(off)
  if (i == 0) return z_lo;
  const double prod = slope * (double)i;
  return (float)(prod + (double)z_lo);
}

// ------------------------------------------------------------------------------------------------------------------
// wave64 reductions
// Wave-wide reductions on the VALU's data-parallel-primitive lanes (no LDS round trips: a __shfl_xor butterfly is six
// DEPENDENT ds_bpermute_b32: single env 61.0 -> 60.1 us, C4 93.6 -> 91.5 us, neutral at 8192 envs).  Four DPP steps
// leave every lane of a 16-lane row with its row's result (quad swaps, half-row mirror, row mirror), the four row
// results are combined through scalar registers.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_lane_(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, dpp_<0xB1>(v));      // quad_perm [1,0,3,2]
  v = fminf(v, dpp_<0x4E>(v));      // quad_perm [2,3,0,1]
  v = fminf(v, dpp_<0x141>(v));     // row_half_mirror
  v = fminf(v, dpp_<0x140>(v));     // row_mirror
  return fminf(fminf(row_lane_(v, 0), row_lane_(v, 16)), fminf(row_lane_(v, 32), row_lane_(v, 48)));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_<0xB1>(v);
  v += dpp_<0x4E>(v);
  v += dpp_<0x141>(v);
  v += dpp_<0x140>(v);
  return (row_lane_(v, 0) + row_lane_(v, 16)) + (row_lane_(v, 32) + row_lane_(v, 48));
}

}  // namespace cpmppi
