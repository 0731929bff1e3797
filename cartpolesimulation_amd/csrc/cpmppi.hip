// cpmppi.hip — HIP kernels (gfx950) and the C ABI of libcpmppi.so.   See include/cpmppi.h for the contract.
//
// Kernel inventory
//   rollout_cost_kernel<COST,FAST,NOISE>  the hot path: one lane = one rollout, 6-float state + held control + running
//                                         cost in VGPRs; per-env data wave-uniform (SGPR); perturbation tile staged
//                                         through LDS with coalesced HBM reads; block-level soft-min partials
//                                         {min S, sum e, sum e*du[.]} via wave shuffles + LDS.
//   finalize_kernel<KNOT_SPACE>           merges the per-block partials of one env (rescaled to the env-wide minimum),
//                                         applies shift / update / clip, writes u_nom and Q.
//   sample_kernel / interpolate_kernel    a17 (Philox knots, scipy-interp1d-compatible interpolation).
//   predict_kernel<FAST, STAGED>          predictor seam: trajectories [B,H+1,6] (stores staged through LDS for large launches).
//   trajectory_cost_kernel                cost seam on materialised trajectories.
//   rwa_kernel                            a16 on given (S, delta_u).
#include <hip/hip_runtime.h>
#include <chrono>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <exception>
#include <vector>

#include "cpmppi.h"
#include "cpmppi_internal.hpp"
#include "cpmppi_device.hpp"
#include "cpmppi_gru.hpp"
#include "cpmppi_gru16.hpp"
#include "cpmppi_grad.hpp"

using namespace cpmppi;

#include "cpmppi_rollout.hpp"

using namespace cpmppi_k;

// rollout_cost_kernel is instantiated in cpmppi_rollout_latency.hip (VARIANT 0), cpmppi_rollout_throughput.hip (1) and
// cpmppi_rollout_mid.hip (2), each with its own compiler flags; nothing of it may be instantiated here
namespace cpmppi_k {
CPMPPI_LATENCY_INSTANCES(CPMPPI_DECLARE_ROLLOUT)
CPMPPI_LATENCY_BUFFER_INSTANCES(CPMPPI_DECLARE_ROLLOUT)
CPMPPI_MID_INSTANCES(CPMPPI_DECLARE_ROLLOUT)
CPMPPI_MID_BUFFER_INSTANCES(CPMPPI_DECLARE_ROLLOUT)
CPMPPI_THROUGHPUT_INSTANCES(CPMPPI_DECLARE_ROLLOUT)
CPMPPI_ODE_LATENCY_INSTANCES(CPMPPI_DECLARE_ROLLOUT_ODE)
CPMPPI_ODE_THROUGHPUT_INSTANCES(CPMPPI_DECLARE_ROLLOUT_ODE)
CPMPPI_ODE_LONE_INSTANCES(CPMPPI_DECLARE_ROLLOUT_ODE)
}  // namespace cpmppi_k

namespace {

template <bool KNOT_SPACE>
__global__ __launch_bounds__(BLOCK) void finalize_kernel(const Params p, const float* __restrict__ partial,
                                                         uint32_t nb, uint32_t W, const float* u_nom, float* u_nom_out,
                                                         float* __restrict__ Q_out, const GatherSync gs) {
  finalize_env<KNOT_SPACE, false>(p, partial, nb, W, u_nom, u_nom_out, Q_out, blockIdx.x, nullptr, gs);
}

// a17: knots[E,N,P] and/or delta_u[E,N,H].  One lane draws (or loads) the knots of one rollout into LDS; the wave then
// writes its 64 delta_u rows with lane = time-step, i.e. whole 256-byte row segments per store instruction.
__global__ __launch_bounds__(BLOCK) void sample_kernel(const Params p, uint32_t E, uint64_t seed, uint64_t offset,
                                                       uint32_t env_offset, const float* __restrict__ knots_in,
                                                       float* __restrict__ knots_out, float* __restrict__ du_out) {
  extern __shared__ float kn_lds[];                               // [WAVES][64][P+1]
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const size_t total = (size_t)E * p.N;
  const size_t wave_row0 = ((size_t)blockIdx.x * WAVES + wave) * 64;
  const size_t r = wave_row0 + lane;                              // flat (env, rollout)
  const uint32_t stride = p.P + 1;
  float* __restrict__ mine = kn_lds + (wave * 64 + lane) * stride;
  if (r < total) {
    const uint32_t env = (uint32_t)(r / p.N), n = (uint32_t)(r % p.N);
    for (uint32_t j0 = 0; j0 < p.P; j0 += 4) {
      float zq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (!knots_in) philox_normal_quad(seed, offset, env_offset + env, n, j0 >> 2, zq);
#pragma unroll
      for (uint32_t s = 0; s < 4; ++s) {
        const uint32_t j = j0 + s;
        if (j < p.P) {
          const float z = knots_in ? knots_in[r * p.P + j] : p.sigma * zq[s];
          mine[j] = z;
          if (knots_out) knots_out[r * p.P + j] = z;
        }
      }
    }
  }
  if (!du_out) return;
  __syncthreads();
  const float* __restrict__ wk = kn_lds + wave * 64 * stride;
  for (uint32_t row = 0; row < 64 && wave_row0 + row < total; ++row) {
    for (uint32_t k = lane; k < p.H; k += 64) {
      const uint32_t j = k / p.period, i = k % p.period;
      const float zl = wk[row * stride + j], zh = wk[row * stride + j + 1];
      du_out[(wave_row0 + row) * p.H + k] = (p.interp_f32 && !knots_in)
          ? interp_from_slope32(knot_slope32(zl, zh, 1.0f / (float)p.period), zl, i)      // what the FAST Philox kernel forms
          : interp_knots(zl, zh, i, p.period);
    }
  }
}

// predictor seam.  traj[B,H+1,6] is the reference's tensor (row-major per rollout, 24 bytes per state): a lane integrates
// one rollout, the states of PRED_KS control steps are parked in LDS (odd row stride: conflict-free) and then written by
// the whole wave with consecutive lanes on consecutive floats of a row's 192-byte segment — whole sectors per store
// instead of 64 scattered 4-byte pieces 1224 bytes apart (0.9 TB/s at 262144 rollouts before).
constexpr int PRED_KS = 8;
constexpr int PRED_ROW = PRED_KS * 6 + 1;
// STAGED = false: every lane stores its own states directly — the shorter path for launches that do not fill the chip
// (1024 rollouts: 59 us against 82 us staged; 262144 rollouts: 416 us against 280 us staged).
// INTEG: the in-tree ODE predictor (cpmppi_device.hpp: PREDICTOR_ODE_V0 | PREDICTOR_ODE).
template <bool FAST, bool STAGED, int INTEG = PREDICTOR_ODE_V0>
__device__ __forceinline__ void predict_control_step(State<float>& st, float Qk, const Params& p, const EnvConst& ec) {
  if constexpr (INTEG == PREDICTOR_ODE) {
    if constexpr (FAST) control_step_cromer_fast<float>(st, ec.uK_scale * Qk, p.S, p.t_step, p, ec);
    else for (uint32_t sub = 0; sub < p.S; ++sub) substep_precise_cromer(st, p.u_max * Qk, p.t_step, p, ec);
  } else {
    if constexpr (FAST) control_step_fast<float>(st, ec.uK_scale * Qk, p.S, p.t_step, p, ec, p.THL);
    else for (uint32_t sub = 0; sub < p.S; ++sub) substep_precise(st, p.u_max * Qk, p.t_step, p, ec);
  }
}

template <bool FAST, bool STAGED, int INTEG = PREDICTOR_ODE_V0>
__global__ __launch_bounds__(BLOCK) void predict_kernel(const Params p, uint32_t B, uint32_t H,
                                                        const float* __restrict__ s0, const float* __restrict__ Q,
                                                        const float* __restrict__ Lp, float* __restrict__ traj) {
  if constexpr (!STAGED) {
    const size_t b = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (b >= B) return;
    const EnvConst ec = make_env_const(p, Lp ? Lp[b] : p.L_default);
    const float* s = s0 + b * 6;
    State<float> st{s[0], s[1], s[2], s[3], s[4], s[5]};
    float* o = traj + b * (size_t)(H + 1) * 6;
    o[0] = st.th; o[1] = st.w; o[2] = st.c; o[3] = st.s; o[4] = st.x; o[5] = st.v;
    for (uint32_t k = 0; k < H; ++k) {
      predict_control_step<FAST, STAGED, INTEG>(st, Q[b * H + k], p, ec);
      o += 6;
      o[0] = st.th; o[1] = st.w; o[2] = st.c; o[3] = st.s; o[4] = st.x; o[5] = st.v;
    }
    return;
  }
  __shared__ float park[STAGED ? WAVES : 1][STAGED ? 64 * PRED_ROW : 1];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const size_t b_raw = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const bool valid = b_raw < B;
  const size_t b = valid ? b_raw : (size_t)B - 1;                 // (idle lanes of the last block shadow the last rollout)
  const size_t wave_b0 = (size_t)blockIdx.x * BLOCK + (size_t)wave * 64;
  const EnvConst ec = make_env_const(p, Lp ? Lp[b] : p.L_default);
  const float* s = s0 + b * 6;
  State<float> st{s[0], s[1], s[2], s[3], s[4], s[5]};
  if (valid) {
    float* o = traj + b * (size_t)(H + 1) * 6;
    o[0] = st.th; o[1] = st.w; o[2] = st.c; o[3] = st.s; o[4] = st.x; o[5] = st.v;
  }
  float* __restrict__ mine = park[wave] + lane * PRED_ROW;
  for (uint32_t k0 = 0; k0 < H; k0 += PRED_KS) {
    const uint32_t kn = (H - k0 < (uint32_t)PRED_KS) ? H - k0 : (uint32_t)PRED_KS;
    for (uint32_t kk = 0; kk < kn; ++kk) {
      predict_control_step<FAST, STAGED, INTEG>(st, Q[b * H + k0 + kk], p, ec);
      float* m = mine + kk * 6;
      m[0] = st.th; m[1] = st.w; m[2] = st.c; m[3] = st.s; m[4] = st.x; m[5] = st.v;
    }
    __syncthreads();
    const uint32_t seg = kn * 6;                                  // floats per row in this chunk
    for (uint32_t idx = lane; idx < 64u * seg; idx += 64u) {
      const uint32_t row = idx / seg, col = idx - row * seg;
      if (wave_b0 + row < B)
        traj[((wave_b0 + row) * (size_t)(H + 1) + k0 + 1) * 6 + col] = park[wave][row * PRED_ROW + col];
    }
    __syncthreads();
  }
}

// cost seam on materialised trajectories
// Cost seam.  traj[B,H+1,6] and inputs[B,H] are the reference's tensors (row-major per rollout): lane = time-step, a
// wave walks its rows — a row's 24(H+1) bytes are read by consecutive lanes (coalesced) instead of 64 rows 1224 bytes
// apart per load as in the first version (0.59 TB/s at 262144 rows) — and sums a row's stage costs by wave reduction.
__global__ __launch_bounds__(BLOCK) void trajectory_cost_kernel(const Params p, uint32_t B, uint32_t H, uint32_t rows_per_wave,
                                                                const float* __restrict__ traj,
                                                                const float* __restrict__ inputs, float x_t, float te,
                                                                const float* __restrict__ u_nom,
                                                                const float* __restrict__ u_prev,
                                                                float* __restrict__ stage_out,
                                                                float* __restrict__ terminal_out,
                                                                float* __restrict__ total_out) {
  const uint32_t lane = threadIdx.x & 63u;
  const size_t wave = ((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const size_t b0 = wave * rows_per_wave;
  for (size_t b = b0; b < b0 + rows_per_wave && b < B; ++b) {
    const float* __restrict__ row = traj + b * (size_t)(H + 1) * 6;
    float sum = 0.0f;
    for (uint32_t k0 = 0; k0 < H; k0 += 64) {
      const uint32_t k = k0 + lane;
      float c = 0.0f;
      if (k < H) {
        const float* __restrict__ t = row + (size_t)k * 6;
        const float in = inputs[b * H + k];
        const float cosang = cosf(t[0]);
        if (p.cost_id == CPMPPI_COST_QBGM) c = stage_qbgm<float>(p, t[4], cosang, t[1], in, x_t, te);
        else if (p.cost_id == CPMPPI_COST_DEFAULT)      // (default.py and, by p.qb_mode, quadratic_boundary / _nonconvex: ccrc only with a previous input)
          c = stage_default<float>(p, t[4], cosang, in, x_t, te, k == 0 ? (u_prev ? u_prev[0] : 0.0f) : inputs[b * H + k - 1],
                                   p.qb_mode != 0u && u_prev != nullptr);
        else if (p.cost_id == CPMPPI_COST_QBG)
          c = stage_qbg<float>(p, t[4], cosang, t[1], in, k == 0 ? (u_prev ? u_prev[0] : 0.0f) : inputs[b * H + k - 1], x_t, te);
        else c = stage_legacy<float>(p, t[4], cosang, t[1], t[5], u_nom[k], in, u_prev ? u_prev[k] : 0.0f, x_t);
        if (stage_out) stage_out[b * H + k] = c;
      }
      sum += wave_sum(c);
    }
    if (lane == 0) {
      const float* __restrict__ tl = row + (size_t)H * 6;
      const float term = (p.cost_id == CPMPPI_COST_QBGM || p.cost_id == CPMPPI_COST_QBG) ? 0.0f : terminal_indicator<float>(p, tl[0], tl[4], x_t);
      if (terminal_out) terminal_out[b] = term;
      if (total_out)
        total_out[b] = (p.cost_id == CPMPPI_COST_LEGACY || p.horizon_reduce == CPMPPI_REDUCE_SUM)
                           ? (sum + term) : (sum + term) / (float)(H + 1);
    }
  }
}

// a16 on given (S, delta_u): one block per env
__global__ __launch_bounds__(BLOCK) void rwa_kernel(const Params p, const float* __restrict__ S,
                                                    const float* __restrict__ du, float* __restrict__ out) {
  __shared__ float red[WAVES];
  __shared__ float sh_m, sh_a;
  const uint32_t env = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const float* Se = S + (size_t)env * p.N;
  const float* de = du + (size_t)env * p.N * p.H;
  float m = INFINITY;
  for (uint32_t n = tid; n < p.N; n += BLOCK) m = fminf(m, Se[n]);
  m = wave_min(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (tid == 0) { float v = red[0]; for (int w = 1; w < WAVES; ++w) v = fminf(v, red[w]); sh_m = v; }
  __syncthreads();
  m = sh_m;
  float a = 0.0f;
  for (uint32_t n = tid; n < p.N; n += BLOCK) a += expf((-1.0f / p.LBD) * (Se[n] - m));
  a = wave_sum(a);
  __syncthreads();
  if (lane == 0) red[wave] = a;
  __syncthreads();
  if (tid == 0) { float v = red[0]; for (int w = 1; w < WAVES; ++w) v += red[w]; sh_a = v; }
  __syncthreads();
  a = sh_a;
  // lane = column, one wave = every WAVES-th row: rows are read coalesced and a row's weight is formed once per wave (the
  // first version evaluated expf N x H times from H threads); the waves' sums meet in LDS
  __shared__ float part[WAVES][64];
  for (uint32_t k0 = 0; k0 < p.H; k0 += 64) {
    const uint32_t k = k0 + lane;
    float acc = 0.0f;
    const uint32_t kk = k < p.H ? k : 0u;
    uint32_t n = wave;
    for (; n + 7 * WAVES < p.N; n += 8 * WAVES) {           // eight rows in flight: the pass is bound by load latency
      float x[8], e[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { x[u] = de[(size_t)(n + u * WAVES) * p.H + kk]; e[u] = Se[n + u * WAVES]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_fmaf(expf((-1.0f / p.LBD) * (e[u] - m)), x[u], acc);
    }
    for (; n < p.N; n += WAVES) acc = __builtin_fmaf(expf((-1.0f / p.LBD) * (Se[n] - m)), de[(size_t)n * p.H + kk], acc);
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && k < p.H) {
      float v = part[0][lane];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) v += part[w][lane];
      out[(size_t)env * p.H + k] = v / a;
    }
    __syncthreads();
  }
}

// Plant (caller side): one control period of E simulated cartpoles under held controls, with the reference's
// experiment schedule and the recording in the same launch (include/cpmppi.h, cpmppi_plant_args, lists the order of events; one
// env per lane).  The pole length may change from one simulation step to the next (CartPole/__init__.py:529-537): its folded
// constants are re-formed only on a change.
struct PlantDev {
  uint32_t E, row_envs, n_sub, period_steps, save_every, sched_stride;
  float dt_sim;
  uint64_t period, save_rows, ctrl_rows, sched_rows;
  const unsigned long long* period_dev;
  float* s;
  const float* Q;
  const float* L;
  float *states_log, *dd_log, *Q_log;
  const float *tp_table, *te_table, *L_table;
  float *tp_out, *te_out, *L_out;
  const float *m_pole, *m_table, *Lc_table;
  const float* Qd_table;
  float Q_bias;
  float* Qa_out;
  // measurement chain
  float *s_meas, *hist;
  uint32_t hist_len, lat_steps;
  double lat_frac;
  const float* noise_table;
  const double* off_table;
  const uint8_t* informed_table;
};

// wrap_angle_rad on a float64 (CartPole/_CartPole_mathematical_helpers.py:13-21)
__device__ __forceinline__ double wrap_angle_f64(double a) {
  constexpr double PI = 3.141592653589793, TWO_PI = 6.283185307179586;
  const double m = fmod(a, TWO_PI);
  return m < -PI ? m + TWO_PI : (m > PI ? m - TWO_PI : m);
}

__global__ __launch_bounds__(BLOCK) void plant_kernel(const Params p0, const PlantDev a) {
  const uint32_t env = blockIdx.x * BLOCK + threadIdx.x;
  if (env >= a.E) return;
  const uint32_t E = a.row_envs;                                       // envs per ROW of the logs and tables (>= a.E: an env group's slice)
  // a period the device counter cannot name (still 0) is advanced from the schedule's first row and neither recorded nor published
  uint64_t c = a.period;
  bool known = true;
  if (a.period_dev) {
    const uint64_t cnt = (uint64_t)*a.period_dev;
    known = cnt != 0u;
    c = known ? cnt - 1u : 0u;
  }
  const uint64_t g0 = c * a.period_steps;                              // simulation step at which this period's control was computed
  auto sched_row = [&](uint64_t g) -> size_t {
    const uint64_t r = g / a.sched_stride;
    return (size_t)(r < a.sched_rows ? r : a.sched_rows - 1u) * E + env;
  };
  Params p = p0;                                                       // (this env's copy: its pole mass may differ and change)
  float Lcur = a.L_table ? a.L_table[sched_row(g0)] : (a.L ? a.L[env] : p.L_default);
  p.m_pole = a.m_table ? a.m_table[sched_row(g0)] : (a.m_pole ? a.m_pole[env] : p0.m_pole);
  EnvConst ec = make_env_const(p, Lcur);
  float* se = a.s + (size_t)env * 6;
  State<float> st{se[0], se[1], se[2], se[3], se[4], se[5]};
  float q = a.Q[env];
  if (a.Q_log && known && c < a.ctrl_rows) a.Q_log[(size_t)c * E + env] = q;
  if (a.Qd_table && known && c < a.ctrl_rows)                          // add_control_noise (:523-524): two float32 additions
    q = __fadd_rn(__fadd_rn(q, a.Qd_table[(size_t)c * E + env]), a.Q_bias);
  if (a.Qa_out) a.Qa_out[env] = q;                                     // the next call's Q_ccrc (:489)
  const float u = p.u_max * q;
  float aDD, xDD;
  ode_precise(st.c, st.s, st.w, st.v, u, p, ec, aDD, xDD);             // CartPole/__init__.py:316-320 (Update_Q, Q2u, cartpole_ode)
  auto log_dd = [&](uint64_t g) {
    if (!a.dd_log || !known || g % a.save_every) return;
    const uint64_t r = g / a.save_every;
    if (r < a.save_rows) { float* d = a.dd_log + ((size_t)r * E + env) * 2u; d[0] = aDD; d[1] = xDD; }
  };
  log_dd(g0);
  for (uint32_t i = 0; i < a.n_sub; ++i) {
    const uint64_t g = g0 + i + 1u;
    if (a.L_table || a.m_table) {                                      // update_parameters (:529-537) comes first in update_state
      const size_t r = sched_row(g);
      const float Ln = a.L_table ? a.L_table[r] : Lcur;
      const float mn = a.m_table ? a.m_table[r] : p.m_pole;
      if (Ln != Lcur || mn != p.m_pole) { Lcur = Ln; p.m_pole = mn; ec = make_env_const(p, Lcur); }
    }
    plant_substep(st, aDD, xDD, a.dt_sim, p, ec);
    if (a.hist && known) {                                             // the latency buffer (CartPole/latency_adder.py:36-47)
      float* hs = a.hist + ((size_t)(g % a.hist_len) * E + env) * 6u;
      hs[0] = st.th; hs[1] = st.w; hs[2] = st.c; hs[3] = st.s; hs[4] = st.x; hs[5] = st.v;
    }
    ode_precise(st.c, st.s, st.w, st.v, u, p, ec, aDD, xDD);
    if (known && g % a.save_every == 0u) {
      const uint64_t r = g / a.save_every;
      if (a.states_log && r < a.save_rows) {
        float* lg = a.states_log + ((size_t)r * E + env) * 6u;
        lg[0] = st.th; lg[1] = st.w; lg[2] = st.c; lg[3] = st.s; lg[4] = st.x; lg[5] = st.v;
      }
      // a FULL period's last step gets its control (hence its derivatives) from the next controller call; the steps of a
      // trailing partial period (n_sub < period_steps: the run ends inside a period) are followed by no call - their rows are
      // completed here under the held control, as the reference's save does (advisor, round 5)
      if (i + 1u < a.period_steps) log_dd(g);
    }
  }
  se[0] = st.th; se[1] = st.w; se[2] = st.c; se[3] = st.s; se[4] = st.x; se[5] = st.v;
  if (a.n_sub && known) {                                              // what the next controller call is handed (:509-520)
    const size_t r = sched_row(g0 + a.n_sub);
    if (a.tp_table && a.tp_out) a.tp_out[env] = a.tp_table[r];
    if (a.te_table && a.te_out) a.te_out[env] = a.te_table[r];
    if (a.L_table && a.L_out) a.L_out[env] = (a.Lc_table ? a.Lc_table : a.L_table)[r];
    if (a.s_meas && a.n_sub == a.period_steps) {                       // what the NEXT controller call sees (add_noise_and_latency, :336-356)
      const uint64_t g1 = g0 + a.n_sub;
      double m_th = st.th, m_w = st.w, m_c = st.c, m_s = st.s, m_x = st.x, m_v = st.v;
      if (a.hist) {
        // the state k steps back; before step 1: the buffer's initial content (zeros, cos = 1)
        const bool h1 = g1 >= (uint64_t)a.lat_steps + 1u, h2 = g1 >= (uint64_t)a.lat_steps + 2u;
        const float* p1 = a.hist + ((size_t)((g1 - (h1 ? a.lat_steps : 0u)) % a.hist_len) * E + env) * 6u;
        const float* p2 = a.hist + ((size_t)((g1 - (h2 ? a.lat_steps + 1u : 0u)) % a.hist_len) * E + env) * 6u;
        const double a_th = h1 ? (double)p1[0] : 0.0, a_w = h1 ? (double)p1[1] : 0.0, a_c = h1 ? (double)p1[2] : 1.0,
                     a_s = h1 ? (double)p1[3] : 0.0, a_x = h1 ? (double)p1[4] : 0.0, a_v = h1 ? (double)p1[5] : 0.0;
        const double b_th = h2 ? (double)p2[0] : 0.0, b_w = h2 ? (double)p2[1] : 0.0, b_c = h2 ? (double)p2[2] : 1.0,
                     b_s = h2 ? (double)p2[3] : 0.0, b_x = h2 ? (double)p2[4] : 0.0, b_v = h2 ? (double)p2[5] : 0.0;
        const double f = a.lat_frac;
        m_th = a_th + f * (b_th - a_th); m_w = a_w + f * (b_w - a_w); m_c = a_c + f * (b_c - a_c);
        m_s = a_s + f * (b_s - a_s); m_x = a_x + f * (b_x - a_x); m_v = a_v + f * (b_v - a_v);
      }
      if (a.noise_table && c + 1u < a.ctrl_rows) {                     // noise_adder.py:71-82
        const float* nz = a.noise_table + ((size_t)(c + 1u) * E + env) * 4u;
        m_th = wrap_angle_f64(m_th + (double)nz[0]);
        m_c = cos(m_th); m_s = sin(m_th);
        m_x += (double)nz[1]; m_w += (double)nz[2]; m_v += (double)nz[3];
      }
      const double off = a.off_table ? a.off_table[r] : 0.0;          // :348-356 (always re-forms cos / sin from the float64 angle)
      m_th = wrap_angle_f64(m_th + off);
      if (!a.informed_table || a.informed_table[r]) m_th = wrap_angle_f64(m_th - off);   // :501-505 (cos / sin formed again: only the last pair survives)
      m_c = cos(m_th); m_s = sin(m_th);
      float* sm = a.s_meas + (size_t)env * 6u;
      sm[0] = (float)m_th; sm[1] = (float)m_w; sm[2] = (float)m_c; sm[3] = (float)m_s; sm[4] = (float)m_x; sm[5] = (float)m_v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// GRU predictor (BASELINE configs[4]): see cpmppi_gru.hpp.  256 threads = 4 waves x 32 rollouts.
constexpr int GRU_ROLLOUTS_PER_BLOCK = 32 * WAVES;

__device__ __forceinline__ void gru_load_image(float* __restrict__ lds, const float* __restrict__ image) {
  for (int i = threadIdx.x; i < GRU_IMAGE_FLOATS; i += BLOCK) lds[i] = image[i];
  __syncthreads();
}

// predictor seam with the neural predictor: s0[B,6], Q[B,H], h0[2,B,32] or NULL -> traj[B,H+1,6], h_out[2,B,32] or NULL
// (one wave per SIMD: the exact-f32 MFMA chain of gru_step with all its fragment loads in flight wants more than 256 registers -
// compiled for two waves per SIMD it spilled 38 of them to a 156-byte scratch slot; the seam is bound by the matrix pipe either way)
__global__ __launch_bounds__(BLOCK, 1) void gru_predict_kernel(const GruNorm nm, const float* __restrict__ image, uint32_t B,
                                                            uint32_t H, const float* __restrict__ s0,
                                                            const float* __restrict__ Q, const float* __restrict__ h0,
                                                            float* __restrict__ traj, float* __restrict__ h_out) {
  extern __shared__ float lds[];
  gru_load_image(lds, image);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, c = lane & 31u;
  const size_t b = (size_t)blockIdx.x * GRU_ROLLOUTS_PER_BLOCK + wave * 32 + c;
  const bool valid = b < B;
  const size_t bb = valid ? b : 0;
  const float* s = s0 + bb * 6;
  f16v h1 = gru_load_hidden(h0 ? h0 + bb * 32 : nullptr, lane);
  f16v h2 = gru_load_hidden(h0 ? h0 + ((size_t)B + bb) * 32 : nullptr, lane);
  f16v x = gru_input_tile(nm, s, Q[bb * H], lane);
  float* o = traj + bb * (size_t)(H + 1) * 6;
  if (valid && lane < 32) for (int i = 0; i < 6; ++i) o[i] = s[i];
  for (uint32_t k = 0; k < H; ++k) {
    const f16v out = gru_step(lds, x, h1, h2, lane);
    float st[6];
    gru_output_state(nm, out, lane, st);
    if (valid && lane < 32) {
      o += 6;
      for (int i = 0; i < 6; ++i) o[i] = st[i];
    }
    x = out;                                               // normalised outputs are fed back unchanged
    if (lane >= 32 && k + 1 < H) x[1] = __builtin_fmaf(Q[bb * H + k + 1], nm.in_scale[0], nm.in_shift[0]);
  }
  if (h_out && valid) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      h_out[bb * 32 + gru_tile_row(v, lane >> 5)] = h1[v];
      h_out[((size_t)B + bb) * 32 + gru_tile_row(v, lane >> 5)] = h2[v];
    }
  }
}

// Fused MPPI step with the GRU predictor: same contract as rollout_cost_kernel (plugin costs), h0[E,2,32] or NULL.
template <int COST, int NOISE, bool F16>
__global__ __launch_bounds__(BLOCK, CPMPPI_GRU_MIN_WAVES) void gru_rollout_cost_kernel(const Params p, const StepPtrs a, const GruNorm nm,
                                                                 const float* __restrict__ image,
                                                                 const float* __restrict__ h0) {
  extern __shared__ float lds[];                           // GRU image, then [WAVES][W] weighted sums
  __shared__ float red[2 * WAVES];
  constexpr int IMAGE_FLOATS = F16 ? G16_IMAGE_BYTES / 4 : GRU_IMAGE_FLOATS;
  for (int i = threadIdx.x; i < IMAGE_FLOATS; i += BLOCK) lds[i] = image[i];
  __syncthreads();
  float* bsum = lds + IMAGE_FLOATS;
  const uint32_t env = blockIdx.x / a.nb, blk = blockIdx.x % a.nb;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, c = lane & 31u;
  const uint32_t row0 = blk * GRU_ROLLOUTS_PER_BLOCK + wave * 32;
  const uint32_t n = row0 + c;
  const bool owner = lane < 32 && n < p.N;                 // the lane that accounts for rollout n
  const uint32_t nn = n < p.N ? n : 0;
  const uint32_t H = p.H;
  const uint64_t step_offset = a.offset_dev ? (uint64_t)*a.offset_dev : a.offset;
  const float x_t = a.x_t[env], te = a.te[env];
  const float* __restrict__ s0 = a.s0 + (size_t)env * 6;
  const float* __restrict__ un = a.u_nom + (size_t)env * H;
  f16v h1 = gru_load_hidden(h0 ? h0 + (size_t)env * 64 : nullptr, lane);
  f16v h2 = gru_load_hidden(h0 ? h0 + (size_t)env * 64 + 32 : nullptr, lane);

  auto knot = [&](uint32_t j) __attribute__((always_inline)) -> float {
    if constexpr (NOISE == NOISE_KNOTS) return a.noise[((size_t)env * p.N + nn) * p.P + j];
    else return philox_knot(a.seed, step_offset, a.env_offset + env, nn, j, p.sigma);
  };
  float z_lo = 0.0f, z_hi = 0.0f;
  if constexpr (NOISE != NOISE_DELTA_U) { z_lo = knot(0); z_hi = knot(1); }
  uint32_t ii = 0, j = 0;

  float st[6] = {s0[0], s0[1], s0[2], s0[3], s0[4], s0[5]};
  float cost = 0.0f, corr = 0.0f;
  float cosang = cosf(s0[0]);               // the plugins take cos(angle) of the given state at stage 0
  f16v x;
  GruCarry carry;
  Gru16Carry carry16;
  Gru16State gs;
  const char* __restrict__ ldsb = reinterpret_cast<const char*>(lds);
  if constexpr (F16) {
    gs.h1 = h1; gs.h2 = h2;
    gru16_carry_init(ldsb, gs, lane, carry16);
  } else {
    gru_carry_init(lds, h1, lane, carry);
  }
  const bool half1 = lane >= 32;
#ifdef CPMPPI_GRU_STAMPS
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#endif
  for (uint32_t k = 0; k < H; ++k) {
    float du;
    if constexpr (NOISE == NOISE_DELTA_U) du = a.noise[((size_t)env * p.N + nn) * H + k];
    else if constexpr (F16 && NOISE == NOISE_PHILOX)      // FAST + own noise: one float32 FMA, as the ODE kernel and the sampler
      du = interp_from_slope32(knot_slope32(z_lo, z_hi, 1.0f / (float)p.period), z_lo, ii);
    else du = interp_knots(z_lo, z_hi, ii, p.period);
    const float uk = shifted_nominal(p, un, k);
    float ur = uk + du;
    if (p.control_mode == CPMPPI_CONTROL_CLIP) ur = clamp_(ur, p.lo, p.hi);
    if constexpr (COST == COST_QBGM) cost += stage_qbgm<float, F16>(p, st[4], cosang, st[1], ur, x_t, te);
    else cost += stage_default<float, F16>(p, st[4], cosang, ur, x_t, te);
    corr += mppi_correction<float>(p, p.correction_u == CPMPPI_CORRECTION_U_RUN ? ur : uk, du);
    if (k == 0) x = gru_input_tile(nm, s0, ur, lane);
    else if (half1) x[1] = __builtin_fmaf(ur, nm.in_scale[0], nm.in_shift[0]);
    float out[5];
    if constexpr (F16) {
#ifdef CPMPPI_GRU_STAMPS
      const f16v o = gru16_step(ldsb, x, gs, carry16, lane, stamp_acc, stamp_prev);
#else
      const f16v o = gru16_step(ldsb, x, gs, carry16, lane);
#endif
      out[0] = o[0]; out[1] = o[1]; out[2] = o[2]; out[3] = o[3];
      out[4] = __shfl(o[0], (int)(c + 32u), 64);           // positionD (row 4) lives on the partner lane-half
      if (half1) out[4] = o[0];
    } else {
#ifdef CPMPPI_GRU_STAMPS
      gru_step_pipelined(lds, x, h1, h2, carry, lane, out, stamp_acc, stamp_prev);
#else
      gru_step_pipelined(lds, x, h1, h2, carry, lane, out);
#endif
    }
    gru_output_state_fast(nm, out, st, cosang);
    // normalised outputs are fed back unchanged: rows 0..3 on lane-half 0, row 4 (and Q, row 5) on lane-half 1
    x[0] = half1 ? out[4] : out[0];
    x[1] = half1 ? 0.0f : out[1];
    x[2] = half1 ? 0.0f : out[2];
    x[3] = half1 ? 0.0f : out[3];
    if constexpr (NOISE != NOISE_DELTA_U) {
      if (++ii == p.period) {
        ii = 0; ++j;
        z_lo = z_hi;
        if (j + 1 < p.P) z_hi = knot(j + 1);
      }
    }
  }
  st[0] = atan2f(st[3], st[2]);             // predictors_customization.py:121-127, needed for the terminal cost only
#ifdef CPMPPI_GRU_STAMPS
  if (lane == 0)
    for (int i = 0; i < 6; ++i) atomicAdd(&g_gru_stamp_sum[i], stamp_acc[i]);
  if (lane == 0) atomicAdd(&g_gru_stamp_sum[6], 1ull);
#endif
  const float term = (COST == COST_DEFAULT) ? terminal_indicator<float>(p, st[0], st[4], x_t) : 0.0f;
  float S_total = (p.horizon_reduce == CPMPPI_REDUCE_SUM) ? (cost + term) : (cost + term) / (float)(H + 1);
  S_total += corr;
  if (a.S_out && owner) a.S_out[(size_t)env * p.N + n] = S_total;

  const float m_w = wave_min(owner ? S_total : INFINITY);
  if (lane == 0) red[wave] = m_w;
  __syncthreads();
  float m_b = red[0];
#pragma unroll
  for (int w = 1; w < WAVES; ++w) m_b = fminf(m_b, red[w]);
  const float e = owner ? expf((-1.0f / p.LBD) * (S_total - m_b)) : 0.0f;
  const float a_w = wave_sum(e);
  if (lane == 0) red[WAVES + wave] = a_w;
  const uint32_t W = a.W;
  float* __restrict__ my_bsum = bsum + wave * W;
  if constexpr (NOISE == NOISE_PHILOX) {
    for (uint32_t jj = 0; jj < W; ++jj) {
      const float v = wave_sum(e * philox_knot(a.seed, step_offset, a.env_offset + env, nn, jj, p.sigma));
      if (lane == 0) my_bsum[jj] = v;
    }
  } else {
    const float* __restrict__ src = a.noise + ((size_t)env * p.N + row0) * W;
    const uint32_t rows = (row0 < p.N) ? ((p.N - row0 < 32u) ? p.N - row0 : 32u) : 0u;
    for (uint32_t c0 = 0; c0 < W; c0 += 64) {
      const uint32_t col = c0 + lane;
      float acc = 0.0f;
      const float* __restrict__ colp = src + (col < W ? col : 0u);
      uint32_t r = 0;
      for (; r + 8 <= rows; r += 8) {                      // eight independent row loads in flight (latency-bound pass)
        float xr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xr[u] = colp[(size_t)(r + u) * W];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          acc = __builtin_fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), r + u)), xr[u], acc);
      }
      for (; r < rows; ++r)
        acc = __builtin_fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), r)), colp[(size_t)r * W], acc);
      if (col < W) my_bsum[col] = acc;
    }
  }
  __syncthreads();
  float* __restrict__ outp = a.partial + ((size_t)env * a.nb + blk) * (2 + W);
  if (tid == 0) {
    float a_b = red[WAVES];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) a_b += red[WAVES + w];
    outp[0] = m_b;
    outp[1] = a_b;
  }
  for (uint32_t cc = tid; cc < W; cc += BLOCK) {
    float v = bsum[cc];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) v += bsum[w * W + cc];
    outp[2 + cc] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// CEM (hyper-parameters Control_Toolkit_ASF/config_optimizers.yml:1-11 "cem-tf"): the same rollout +
// cost kernel, a different sampler and a top-k reduction instead of the soft-min.
// Q[e,n,k] = clip(mean[e,k] + stdev[e,k] * z), z ~ N(0,1) from Philox (rollout, env, step pair, offset).
// ---- the TILED perturbation layout --------------------------------------------------------------------------------
// delta_u_tiled[E][G = ceil(N/64)][Hq = ceil(H/4)][64 rows][4 steps]: element (env, n, k) lives at
//   ((((env * G + n / 64) * Hq + k / 4) * 64 + n % 64) * 4 + k % 4;   rows >= N and steps >= H are zero.
// A wave of the rollout kernel reads it with one 16-byte load per lane per four control steps: 1 KB of contiguous memory
// per wave-instruction, every byte used once per pass (the rollout-major reference layout delta_u[E,N,H] gives 200-byte
// rows, of which a time tile touches 32 bytes: 5.9x the algorithmic traffic).

// a17 straight into the tiled layout: one wave per (env, row group); lane = row; knots staged per lane in LDS.
__global__ __launch_bounds__(BLOCK) void sample_tiled_kernel(const Params p, uint32_t E, uint64_t seed, uint64_t offset,
                                                             uint32_t env_offset, const float* __restrict__ knots_in,
                                                             float* __restrict__ tiled_out) {
  extern __shared__ float kn_lds[];                               // [WAVES][64][P+1]
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t G = (p.N + 63u) >> 6, Hq = (p.H + 3u) >> 2;
  const size_t grp = (size_t)blockIdx.x * WAVES + wave;           // flat (env, group)
  if (grp >= (size_t)E * G) return;
  const uint32_t env = (uint32_t)(grp / G), n = (uint32_t)(grp % G) * 64u + lane;
  const uint32_t stride = p.P + 1;
  float* __restrict__ mine = kn_lds + (wave * 64 + lane) * stride;
  const bool valid = n < p.N;
  for (uint32_t j0 = 0; j0 < p.P; j0 += 4) {
    float zq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (valid && !knots_in) philox_normal_quad(seed, offset, env_offset + env, n, j0 >> 2, zq);
#pragma unroll
    for (uint32_t s = 0; s < 4; ++s) {
      const uint32_t j = j0 + s;
      if (j < p.P) mine[j] = !valid ? 0.0f : (knots_in ? knots_in[((size_t)env * p.N + n) * p.P + j] : p.sigma * zq[s]);
    }
  }
  float4* __restrict__ out = reinterpret_cast<float4*>(tiled_out) + grp * Hq * 64u + lane;
  const float inv_period = 1.0f / (float)p.period;
  for (uint32_t q = 0; q < Hq; ++q) {
    float v[4];
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) {
      const uint32_t k = 4u * q + c;
      if (k < p.H && valid) {
        const uint32_t j = k / p.period, i = k % p.period;
        const float zl = mine[j], zh = mine[j + 1];
        v[c] = (p.interp_f32 && !knots_in) ? interp_from_slope32(knot_slope32(zl, zh, inv_period), zl, i)
                                           : interp_knots(zl, zh, i, p.period);
      } else {
        v[c] = 0.0f;
      }
    }
    out[(size_t)q * 64u] = float4{v[0], v[1], v[2], v[3]};
  }
}

// delta_u[E,N,H] (reference layout) -> tiled: one wave per (env, row group); 64 x 64 sub-blocks through LDS (rows of the
// sub-block are 256 contiguous bytes of the source; the destination quads are written 1 KB per wave-instruction).
__global__ __launch_bounds__(BLOCK) void tile_kernel(const Params p, uint32_t E, const float* __restrict__ du,
                                                     float* __restrict__ tiled_out) {
  __shared__ float blk[WAVES][64][65];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t G = (p.N + 63u) >> 6, Hq = (p.H + 3u) >> 2;
  const size_t grp = (size_t)blockIdx.x * WAVES + wave;
  if (grp >= (size_t)E * G) return;
  const uint32_t env = (uint32_t)(grp / G), n0 = (uint32_t)(grp % G) * 64u;
  const float* __restrict__ src = du + ((size_t)env * p.N + n0) * p.H;
  float4* __restrict__ out = reinterpret_cast<float4*>(tiled_out) + grp * Hq * 64u + lane;
  for (uint32_t k0 = 0; k0 < p.H; k0 += 64) {
    for (uint32_t r0 = 0; r0 < 64; r0 += 16) {                             // row r: steps k0 .. k0+63, lane = step
      float v[16];                                                         // sixteen row segments in flight
#pragma unroll
      for (uint32_t u = 0; u < 16; ++u)
        v[u] = (n0 + r0 + u < p.N && k0 + lane < p.H) ? src[(size_t)(r0 + u) * p.H + k0 + lane] : 0.0f;
#pragma unroll
      for (uint32_t u = 0; u < 16; ++u) blk[wave][r0 + u][lane] = v[u];
    }
    // (one wave owns blk[wave]: no block barrier; the wave's own LDS writes are ordered before its reads)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (uint32_t c = 0; c < 16 && k0 + 4u * c < p.H; ++c)
      out[(size_t)((k0 >> 2) + c) * 64u] = float4{blk[wave][lane][4 * c], blk[wave][lane][4 * c + 1],
                                                 blk[wave][lane][4 * c + 2], blk[wave][lane][4 * c + 3]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
}

__global__ __launch_bounds__(BLOCK) void cem_sample_kernel(const Params p, uint32_t E, const float* __restrict__ mean,
                                                           const float* __restrict__ stdev, uint64_t seed, uint64_t offset,
                                                           uint32_t env_offset, float* __restrict__ Q) {
  const size_t r = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (r >= (size_t)E * p.N) return;
  const uint32_t env = (uint32_t)(r / p.N), n = (uint32_t)(r % p.N);
  const float* m = mean + (size_t)env * p.H;
  const float* sd = stdev + (size_t)env * p.H;
  float* q = Q + r * p.H;
  for (uint32_t k = 0; k < p.H; k += 2) {
    float z0, z1;
    philox_normal_pair(seed, offset, env_offset + env, n, k >> 1, z0, z1);
    q[k] = fminf(fmaxf(__builtin_fmaf(sd[k], z0, m[k]), p.lo), p.hi);
    if (k + 1 < p.H) q[k + 1] = fminf(fmaxf(__builtin_fmaf(sd[k + 1], z1, m[k + 1]), p.lo), p.hi);
  }
}

// cem-gmm: samples from a mixture of K Gaussians with equal weights — component c of env e is centred on the elite
// sequence centres[e, c, :] and shares the per-time-step stdev[e, :] — clipped to the control limits.  The component of
// a rollout comes from the same Philox stream as its normals (counter word `pair` = 0x80000000: never a real pair index).
__global__ __launch_bounds__(BLOCK) void cem_gmm_sample_kernel(const Params p, uint32_t E, const float* __restrict__ centres,
                                                               uint32_t K, const float* __restrict__ stdev, uint64_t seed,
                                                               uint64_t offset, uint32_t env_offset, float* __restrict__ Q,
                                                               uint32_t* __restrict__ comp_out) {
  const size_t r = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (r >= (size_t)E * p.N) return;
  const uint32_t env = (uint32_t)(r / p.N), n = (uint32_t)(r % p.N);
  uint32_t c0 = n, c1 = env_offset + env, c2 = 0x80000000u, c3 = (uint32_t)offset;
  philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32));
  const uint32_t comp = (uint32_t)(((uint64_t)c0 * K) >> 32);              // uniform over 0 .. K-1
  if (comp_out) comp_out[r] = comp;
  const float* m = centres + ((size_t)env * K + comp) * p.H;
  const float* sd = stdev + (size_t)env * p.H;
  float* q = Q + r * p.H;
  for (uint32_t k = 0; k < p.H; k += 2) {
    float z0, z1;
    philox_normal_pair(seed, offset, env_offset + env, n, k >> 1, z0, z1);
    q[k] = fminf(fmaxf(__builtin_fmaf(sd[k], z0, m[k]), p.lo), p.hi);
    if (k + 1 < p.H) q[k + 1] = fminf(fmaxf(__builtin_fmaf(sd[k + 1], z1, m[k + 1]), p.lo), p.hi);
  }
}

// One block per env: sort (S, index) ascending with a bitonic network in LDS (ties by index = stable argsort), then
// mean and population standard deviation of the best_k input sequences per time-step, stdev floored at stdev_min.
__global__ __launch_bounds__(BLOCK) void cem_update_kernel(const Params p, const float* __restrict__ S,
                                                           const float* __restrict__ Q, uint32_t best_k, float stdev_min,
                                                           uint32_t Np, float* __restrict__ mean_out,
                                                           float* __restrict__ stdev_out, uint32_t* __restrict__ elite_out) {
  extern __shared__ float cem_lds[];                     // keys[Np], idx[Np]
  float* key = cem_lds;
  uint32_t* idx = reinterpret_cast<uint32_t*>(cem_lds + Np);
  const uint32_t env = blockIdx.x, tid = threadIdx.x;
  for (uint32_t i = tid; i < Np; i += BLOCK) {
    key[i] = i < p.N ? S[(size_t)env * p.N + i] : INFINITY;
    idx[i] = i;
  }
  __syncthreads();
  for (uint32_t k = 2; k <= Np; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < Np; i += BLOCK) {
        const uint32_t l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const float ki = key[i], kl = key[l];
          const uint32_t ii = idx[i], il = idx[l];
          const bool gt = (ki > kl) || (ki == kl && ii > il);
          if (gt == up) { key[i] = kl; key[l] = ki; idx[i] = il; idx[l] = ii; }
        }
      }
      __syncthreads();
    }
  }
  if (elite_out) for (uint32_t i = tid; i < best_k; i += BLOCK) elite_out[(size_t)env * best_k + i] = idx[i];
  const float* Qe = Q + (size_t)env * p.N * p.H;
  for (uint32_t k = tid; k < p.H; k += BLOCK) {
    float m = 0.0f;
    for (uint32_t i = 0; i < best_k; ++i) m += Qe[(size_t)idx[i] * p.H + k];
    m /= (float)best_k;
    float v = 0.0f;
    for (uint32_t i = 0; i < best_k; ++i) { const float d = Qe[(size_t)idx[i] * p.H + k] - m; v = __builtin_fmaf(d, d, v); }
    mean_out[(size_t)env * p.H + k] = m;
    stdev_out[(size_t)env * p.H + k] = fmaxf(sqrtf(v / (float)best_k), stdev_min);
  }
}

thread_local std::string g_create_error;


// ------------------------------------------------------------------------------------------------------------------
// Rollout + plugin cost + its gradient w.r.t. the inputs (cpmppi_grad.hpp).  One lane = one (env, rollout) of the
// flattened [E*N] axis (the gradient optimizers run 16-40 rollouts per env, config_optimizers.yml:60,75: a block per
// env would idle), so per-env quantities are per-lane here.  Check-points ckpt[H][6][E*N] (lane-contiguous) hold the
// state at every control step; sub[S][6][BLOCK] in LDS holds the sub-states of the control step being reversed.
struct GradPtrs {
  const float* s0; const float* Q; const float* x_t; const float* te; const float* L; const float* prev_in;
  float* ckpt; float* S_out; float* grad; uint32_t E;
};

template <int COST, int INTEG = PREDICTOR_ODE_V0>
__global__ __launch_bounds__(BLOCK) void rollout_grad_kernel(const Params p, const GradPtrs a) {
  auto forward_substep = [&](State<float>& s, float uK, const EnvConst& e) __attribute__((always_inline)) {
    if constexpr (INTEG == PREDICTOR_ODE) substep_cromer_plain(s, uK, p.t_step, p, e);
    else substep_fast<float>(s, uK, p.t_step, p, e, p.THL);
  };
  extern __shared__ float sub_states[];            // [S][6][BLOCK]
  const uint32_t tid = threadIdx.x;
  const size_t B = (size_t)a.E * p.N;
  const size_t g = (size_t)blockIdx.x * BLOCK + tid;
  if (g >= B) return;                               // (no block-level barrier below)
  const uint32_t env = (uint32_t)(g / p.N);
  const uint32_t H = p.H, S = p.S;
  const float t = p.t_step;
  const EnvConst ec = make_env_const(p, a.L ? a.L[env] : p.L_default);
  const float x_t = a.x_t[env], te = a.te[env];
  const float* __restrict__ s0 = a.s0 + (size_t)env * 6;
  const float* __restrict__ Q = a.Q + g * H;
  const float cos0 = cosf(s0[0]), sin0 = sinf(s0[0]);   // the plugins take cos(angle) of the given state at stage 0
  const float ub0 = a.prev_in ? a.prev_in[env] : 0.0f;
  const bool clip = p.control_mode == CPMPPI_CONTROL_CLIP;
  const float scale = (p.horizon_reduce == CPMPPI_REDUCE_SUM) ? 1.0f : 1.0f / (float)(H + 1);

  // ---- forward, check-pointing every control step -------------------------------------------------------------
  State<float> st{s0[0], s0[1], s0[2], s0[3], s0[4], s0[5]};
  float cost = 0.0f, cosang = cos0, u_before = ub0;
  for (uint32_t k = 0; k < H; ++k) {
    float* __restrict__ ck = a.ckpt + ((size_t)k * 6) * B + g;
    ck[0] = st.th; ck[B] = st.w; ck[2 * B] = st.c; ck[3 * B] = st.s; ck[4 * B] = st.x; ck[5 * B] = st.v;
    float ur = Q[k];
    if (clip) ur = clamp_(ur, p.lo, p.hi);
    if constexpr (COST == COST_QBGM) cost += stage_qbgm<float, true>(p, st.x, cosang, st.w, ur, x_t, te);
    else if constexpr (COST == COST_DEFAULT) cost += stage_default<float, true>(p, st.x, cosang, ur, x_t, te);
    else cost += stage_qbg<float, true>(p, st.x, cosang, st.w, ur, u_before, x_t, te);
    u_before = ur;
    const float uK = ur * ec.uK_scale;
    for (uint32_t s = 0; s < S; ++s) forward_substep(st, uK, ec);
    cosang = st.c;
  }
  const float term = (COST == COST_DEFAULT) ? terminal_indicator<float>(p, st.th, st.x, x_t) : 0.0f;
  if (a.S_out) a.S_out[g] = (cost + term) * scale;

  // ---- backward --------------------------------------------------------------------------------------------------
  Adjoint lam{0.0f, 0.0f, 0.0f, 0.0f};              // the terminal indicator has zero derivative
  float carry = 0.0f;                               // d stage_{k+1} / d u_k through u_before (quadratic_boundary_grad)
  float* __restrict__ my = sub_states + tid;
  for (uint32_t k = H; k-- > 0;) {
    const float* __restrict__ ck = a.ckpt + ((size_t)k * 6) * B + g;
    const State<float> st0{ck[0], ck[B], ck[2 * B], ck[3 * B], ck[4 * B], ck[5 * B]};
    const float q = Q[k];
    const bool clipped = clip && (q < p.lo || q > p.hi);
    const float ur = clip ? clamp_(q, p.lo, p.hi) : q;
    const float uK = ur * ec.uK_scale;
    State<float> s = st0;
    for (uint32_t i = 0; i < S; ++i) {
      float* __restrict__ d = my + (size_t)i * 6 * BLOCK;
      d[0] = s.th; d[BLOCK] = s.w; d[2 * BLOCK] = s.c; d[3 * BLOCK] = s.s; d[4 * BLOCK] = s.x; d[5 * BLOCK] = s.v;
      forward_substep(s, uK, ec);
    }
    float guK = 0.0f;
    for (uint32_t i = S; i-- > 0;) {
      const float* __restrict__ d = my + (size_t)i * 6 * BLOCK;
      const State<float> si{d[0], d[BLOCK], d[2 * BLOCK], d[3 * BLOCK], d[4 * BLOCK], d[5 * BLOCK]};
      substep_reverse<(INTEG == PREDICTOR_ODE)>(si, uK, t, p, ec, lam, guK);
    }
    // stage k: its own state and control
    const float ca = (k == 0) ? cos0 : st0.c, sa = (k == 0) ? sin0 : st0.s;
    float ub = ub0;
    if (COST == COST_QBG && k > 0) { ub = Q[k - 1]; if (clip) ub = clamp_(ub, p.lo, p.hi); }
    StageGrad sg;
    if constexpr (COST == COST_QBGM) sg = stage_qbgm_grad(p, st0.x, ca, st0.w, ur, x_t, te);
    else if constexpr (COST == COST_DEFAULT) sg = stage_default_grad(p, st0.x, ca, ur, x_t, te);
    else sg = stage_qbg_grad(p, st0.x, ca, st0.w, ur, ub, x_t, te);
    lam.x = __builtin_fmaf(scale, sg.x, lam.x);
    lam.w = __builtin_fmaf(scale, sg.w, lam.w);
    lam.th = __builtin_fmaf(scale * sg.cosang, -sa, lam.th);
    const float gk = __builtin_fmaf(guK, ec.uK_scale, scale * sg.u + carry);
    carry = scale * sg.u_before;
    a.grad[g * H + k] = clipped ? 0.0f : gk;
  }
}

// Adam on input sequences with per-rollout gradient-norm clipping (tf.clip_by_norm over the horizon) and the final
// clip to the action limits; hyper-parameters config_optimizers.yml:52-58,69-73.  One lane = one (env, rollout) row.
__global__ __launch_bounds__(BLOCK) void adam_step_kernel(size_t rows, uint32_t H, float* __restrict__ Q,
                                                          const float* __restrict__ grad, float* __restrict__ m,
                                                          float* __restrict__ v, float lr_t, float beta1, float beta2,
                                                          float eps, float gradmax_clip, float lo, float hi) {
  const size_t r = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (r >= rows) return;
  const float* __restrict__ gr = grad + r * H;
  float ss = 0.0f;
  for (uint32_t k = 0; k < H; ++k) ss = __builtin_fmaf(gr[k], gr[k], ss);
  const float nrm = sqrtf(ss);
  const float sc = (gradmax_clip > 0.0f && nrm > gradmax_clip) ? gradmax_clip / nrm : 1.0f;
  for (uint32_t k = 0; k < H; ++k) {
    const size_t i = r * H + k;
    const float gk = gr[k] * sc;
    const float mk = beta1 * m[i] + (1.0f - beta1) * gk;
    const float vk = beta2 * v[i] + (1.0f - beta2) * gk * gk;
    m[i] = mk; v[i] = vk;
    Q[i] = clamp_(Q[i] - lr_t * mk / (sqrtf(vk) + eps), lo, hi);
  }
}

// advances a device-resident Philox step counter after a step that used it (stream-ordered; graph-replayable)
__global__ void bump_counter_kernel(unsigned long long* c) { *c += 1ull; }

// Plain gradient step with the same per-rollout norm clipping and limit clip (cem-naive-grad-tf, config_optimizers.yml:21-31).
__global__ __launch_bounds__(BLOCK) void sgd_step_kernel(size_t rows, uint32_t H, float* __restrict__ Q,
                                                         const float* __restrict__ grad, float lr, float gradmax_clip,
                                                         float lo, float hi) {
  const size_t r = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (r >= rows) return;
  const float* __restrict__ gr = grad + r * H;
  float ss = 0.0f;
  for (uint32_t k = 0; k < H; ++k) ss = __builtin_fmaf(gr[k], gr[k], ss);
  const float nrm = sqrtf(ss);
  const float sc = (gradmax_clip > 0.0f && nrm > gradmax_clip) ? gradmax_clip / nrm : 1.0f;
  for (uint32_t k = 0; k < H; ++k) Q[r * H + k] = clamp_(Q[r * H + k] - lr * (gr[k] * sc), lo, hi);
}

}  // namespace

struct cpmppi_handle {
  cpmppi_config cfg;
  Params prm;
  int device;
  float* workspace;
  size_t workspace_floats;
  uint32_t nb;
  std::string err;
  // optional per-kernel timing with HIP events recorded on the launch stream (cpmppi_set_profiling)
  uint32_t* counters = nullptr;        // [cfg.E] block-arrival tickets of the fused finalize
  EnvFold* env_fold = nullptr;         // [cfg.E] per-env constants of the throughput build (fold_env_kernel, rewritten before every such launch)
  float* zeros_H = nullptr;            // [cfg.E, cfg.H] zeros: the nominal sequence of a cost-only launch
  float* host_stage = nullptr;         // pinned [cfg.E * 10 + 16]: staging of cpmppi_step_host (state 6, target, equilibrium, L | Q | ticket)
  uint32_t host_ticket_value = 0;      // what the ticket in that block reads once every launch so far has delivered
  double host_t[4] = {0, 0, 0, 0};     // development aid (cpmppi_debug_host_times): sums of staging / launch / wait seconds, calls
  uint32_t host_zero_copy_max = 64;    // up to this many envs cpmppi_step_host runs without copies and stream waits (CPMPPI_HOST_ZERO_COPY_MAX)
  float* dev_stage = nullptr;          // device [cfg.E * 10], allocated on first use
  float* gru_image = nullptr;          // device copy of the LDS fragment image (cpmppi_set_gru)
  void* gru16_image = nullptr;         // device copy of the f16 split image (cpmppi_gru16.hpp)
  float* grad_ckpt = nullptr;          // [H][6][E*N] check-points of cpmppi_rollout_cost_grad (allocated on first use)
  size_t grad_ckpt_floats = 0;
  GruNorm gru_norm;
  bool fuse_finalize = true;           // ODE path: the env's last block finalizes in-kernel (CPMPPI_FUSE_FINALIZE=0 disables)
  uint32_t profile_every = 0;          // 0 = off, 1 = every rollout kernel bracketed, n > 1 = one bracket around n steps
  uint32_t profile_count = 0;
  bool group_open = false;             // n > 1: the current group's closing event is still to come
  std::vector<hipEvent_t> ev;          // triples per sampled step: before rollout, after it, after the trailing kernels
  std::vector<uint8_t> ev_tail;        // per triple: was the third event recorded (a separate finalize / counter kernel ran)
  size_t ev_used = 0;
  cpmppi_comm::CommState* comm = nullptr;   // RCCL communicator + side stream of cpmppi_comm_* (cpmppi_comm.hip)
  float plant_m_pole = 0.0f;           // the pole mass of the simulated PLANT (cfg.m_pole at creation; cpmppi_set_pole_mass does not touch it)
  cpmppi_launch_info last_launch = {0, 0, 0, 0, 0, 0, 0};   // cpmppi_last_launch: the instantiation the last rollout launch used
};

cpmppi_comm::CommState*& cpmppi_internal_comm(cpmppi_handle* h) { return h->comm; }
int cpmppi_internal_device(const cpmppi_handle* h) { return h->device; }

namespace {

int fail(cpmppi_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}

}  // namespace
int cpmppi_internal_fail(cpmppi_handle* h, int code, const std::string& msg) { return fail(h, code, msg); }
namespace {

#define CPMPPI_HIP(h, call)                                                                        \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail((h), CPMPPI_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));         \
  } while (0)

uint32_t knot_count(uint32_t H, uint32_t period) { return (H + period - 1) / period + 1; }

void fill_params(const cpmppi_config& c, Params& p) {
  memset(&p, 0, sizeof(p));
  p.E = c.E; p.N = c.N; p.H = c.H; p.S = c.S; p.period = c.period;
  p.P = knot_count(c.H, c.period);
  p.t_step = (float)((double)c.dt / (double)c.S);      // predictors_customization_v0.py:39
  p.k = c.k; p.m_cart = c.m_cart; p.m_pole = c.m_pole; p.g = c.g; p.J_fric = c.J_fric; p.M_fric = c.M_fric;
  p.u_max = c.u_max; p.THL = c.track_half_length; p.L_default = c.L_default;
  p.cost_id = c.cost_id;
  p.qb_mode = 0;
  if (c.cost_id == CPMPPI_COST_QB || c.cost_id == CPMPPI_COST_QB_NONCONVEX) {     // default.py's kernels, sub-mode in qb_mode
    p.qb_mode = (c.cost_id == CPMPPI_COST_QB) ? 1u : 2u;
    p.cost_id = CPMPPI_COST_DEFAULT;
  }
  memcpy(p.w, c.cost_w, sizeof(p.w));
  p.R = c.R; p.LBD = c.LBD; p.NU = c.NU; p.cc_weight = c.cc_weight; p.sigma = c.sigma;
  p.lo = c.action_low; p.hi = c.action_high;
  const bool clip_run = c.control_mode == CPMPPI_CONTROL_CLIP;
  p.run_lo = clip_run ? c.action_low : -INFINITY; p.run_hi = clip_run ? c.action_high : INFINITY;
  p.horizon_reduce = c.horizon_reduce; p.control_mode = c.control_mode; p.shift_mode = c.shift_mode;
  p.correction_u = c.correction_u;
  p.interp_f32 = (c.math_mode == CPMPPI_MATH_FAST) ? 1u : 0u;
}

// Every entry point runs on the handle's device and leaves the CALLER's current device as it found it (a process that
// shares the HIP runtime with torch must not have its later raw HIP calls retargeted).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      switched = (err == hipSuccess);
    }
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define CPMPPI_ON_DEVICE(h)                                                                        \
  DeviceGuard device_guard_((h)->device);                                                          \
  if (device_guard_.err != hipSuccess)                                                             \
    return fail((h), CPMPPI_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(device_guard_.err))

bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) != 0; }

// Per-env constants of the throughput build (EnvFold, cpmppi_device.hpp): one lane per env, the very device functions the
// other builds call in their prologues.  Runs in front of every throughput-build launch on the same stream (L, the targets and
// s0 are the caller's device arrays and may change between any two steps): ~2 us against launches of a millisecond and more.
__global__ __launch_bounds__(BLOCK) void fold_env_kernel(const Params p, const float* __restrict__ L, const float* __restrict__ te,
                                                         const float* __restrict__ s0, EnvFold* __restrict__ out, uint32_t envs) {
  const uint32_t env = blockIdx.x * BLOCK + threadIdx.x;
  if (env >= envs) return;
  EnvFold f;
  f.ec = make_env_const(p, L ? L[env] : p.L_default);
  f.qf = make_qbgm_folded_lane(p, te[env]);
  f.cos0 = cosf(s0[(size_t)env * 6]);
  f.inv_period = 1.0f / (float)p.period;
  f.nearlim = __builtin_fminf(p.w[6], 1.0f) * p.THL;
  out[env] = f;
}

// development aid (tools/variant_sweep.py, tools/dev/placement.py): the two size limits below which the straight-line builds are
// launched and an LDS pad, overridable from the environment - in a -DCPMPPI_DEV_KNOBS build ONLY (build_variant "devknobs"), and
// read when a handle is created, not per launch.  The shipped library has the constants: no getenv on the launch path, and no stray
// environment variable can change which kernel production launches (advisor, round 5).
struct DevKnobs { uint64_t lone_form_max_waves = 1024ull, latency_max_rollouts = 131071ull, lds_pad = 0; };
static DevKnobs g_knobs;
static void refresh_dev_knobs() {
#ifdef CPMPPI_DEV_KNOBS
  auto env_u64 = [](const char* name, uint64_t dflt) {
    const char* v = getenv(name);
    return (v && *v) ? (uint64_t)strtoull(v, nullptr, 10) : dflt;
  };
  const DevKnobs d;
  g_knobs.lone_form_max_waves = env_u64("CPMPPI_LONE_FORM_MAX_WAVES", d.lone_form_max_waves);
  g_knobs.latency_max_rollouts = env_u64("CPMPPI_LATENCY_MAX_ROLLOUTS", d.latency_max_rollouts);
  g_knobs.lds_pad = env_u64("CPMPPI_LDS_PAD", 0);
#endif
}
template <int COST, bool FAST, int R, int V, int INTEG = PREDICTOR_ODE_V0>
hipError_t launch_rollout_noise(uint32_t noise, dim3 grid, size_t lds, hipStream_t s, const Params& p,
                                const StepPtrs& a) {
  switch (noise) {
    case CPMPPI_NOISE_DELTA_U:
      hipLaunchKernelGGL((rollout_cost_kernel<COST, FAST, NOISE_DELTA_U, R, V, INTEG>), grid, dim3(BLOCK), lds, s, p, a); break;
    case CPMPPI_NOISE_KNOTS:
      hipLaunchKernelGGL((rollout_cost_kernel<COST, FAST, NOISE_KNOTS, R, V, INTEG>), grid, dim3(BLOCK), lds, s, p, a); break;
    case CPMPPI_NOISE_DELTA_U_TILED:
      hipLaunchKernelGGL((rollout_cost_kernel<COST, FAST, NOISE_TILED, R, V, INTEG>), grid, dim3(BLOCK), lds, s, p, a); break;
    default: {
      // development aid (tools/dev/placement.py): CPMPPI_LDS_PAD=<bytes> of extra dynamic LDS per workgroup caps how many
      // workgroups the dispatcher can put on one CU (160 KB each)
      const size_t pad = (size_t)g_knobs.lds_pad;
      if (pad) {
        static bool raised = false;
        if (!raised) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_cost_kernel<COST, FAST, NOISE_PHILOX, R, V, INTEG>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + pad));
          raised = true;
        }
      }
      hipLaunchKernelGGL((rollout_cost_kernel<COST, FAST, NOISE_PHILOX, R, V, INTEG>), grid, dim3(BLOCK), lds + pad, s, p, a); break;
    }
  }
  return hipGetLastError();
}

// Which build of the kernel a launch gets (measured on MI355X, tools/kbench.py / tools/dev/r3_cross.sh):
//   one rollout per lane : latency build up to one wave per SIMD (1024 SIMDs x 64 lanes), throughput build above
//   two rollouts per lane: mid-size build (phased horizon loop: quiet control steps and eventful ones in separate loops) up
//                          to 1572864 rollouts - variant 3 (quiet step unrolled) while the launch has at most one wave
//                          per SIMD, variant 2 above - and the throughput build beyond.  Phased mid-size vs throughput build,
//                          envs x 1024 x 50: 256 envs 126 vs 133 us, 1024 envs 350 vs 355, 1536 envs 482 vs 495, 2048 envs
//                          630 vs 636, 3072 envs 904 vs 890, 8192 envs 2.34 vs 2.29 ms
#ifndef CPMPPI_MID_SIZE_MAX
#define CPMPPI_MID_SIZE_MAX 1572864ull
#endif
constexpr uint64_t MID_SIZE_MAX_ROLLOUTS = CPMPPI_MID_SIZE_MAX;   // (a -D override exists for A/B builds only)
constexpr uint64_t PACKED_MIN_ROLLOUTS = 131072ull;
static uint64_t lone_form_max_waves() { return g_knobs.lone_form_max_waves; }
// (round 5, tools/variant_sweep.py: between 65536 and 131072 rollouts - where the size rule still picks one rollout per lane - the
// straight-line latency build beats the throughput build's loop: 48 x 2048 x 50 71.3 vs 79.8 us, 96 x 1024 x 50 74.9 vs 83.3)
static uint64_t latency_max_rollouts() { return g_knobs.latency_max_rollouts; }
template <int COST>
hipError_t launch_rollout_math(uint32_t math, uint32_t ode, uint32_t rpl, uint32_t noise, dim3 grid, size_t lds, hipStream_t s,
                               const Params& p, const StepPtrs& a, uint32_t* variant_out) {
  // (the build VARIANT of the instantiation launched: 0 latency, 1 throughput, 2 mid-size phased, 3 its lone-wave form)
#define CPMPPI_LAUNCH_V(FAST, R, V, ...) (*variant_out = (V), launch_rollout_noise<COST, FAST, R, V, ##__VA_ARGS__>(noise, grid, lds, s, p, a))
  if (ode == CPMPPI_ODE_CROMER) {
    // predictor_ODE has no events, hence no mid-size (phased) build: latency build up to one wave per SIMD with one rollout
    // per lane, the throughput build otherwise
    if (math != CPMPPI_MATH_FAST) return CPMPPI_LAUNCH_V(false, 1, 1, PREDICTOR_ODE);
    if (rpl == 2)
      return ((uint64_t)grid.x * WAVES <= 1024ull) ? CPMPPI_LAUNCH_V(true, 2, 3, PREDICTOR_ODE) : CPMPPI_LAUNCH_V(true, 2, 1, PREDICTOR_ODE);
    return ((uint64_t)grid.x * BLOCK <= 65536ull) ? CPMPPI_LAUNCH_V(true, 1, 0, PREDICTOR_ODE) : CPMPPI_LAUNCH_V(true, 1, 1, PREDICTOR_ODE);
  }
  if (math == CPMPPI_MATH_FAST) {
    // the throughput build reads its per-env constants from a.env_fold: written here, on the same stream, first
    auto fold_first = [&]() -> hipError_t {
#if CPMPPI_ENV_FOLD
      const uint32_t envs = grid.x / a.nb;
      hipLaunchKernelGGL(fold_env_kernel, dim3((envs + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, p, a.L, a.te, a.s0,
                         const_cast<EnvFold*>(a.env_fold), envs);
      return hipGetLastError();
#else
      return hipSuccess;
#endif
    };
    if (rpl == 2) {
      const bool mid = (uint64_t)grid.x * BLOCK * 2 <= MID_SIZE_MAX_ROLLOUTS;
      // at most one wave per SIMD (256 CUs x 4): the phased build with the quiet control step unrolled
      if ((uint64_t)grid.x * WAVES <= lone_form_max_waves()) return CPMPPI_LAUNCH_V(true, 2, 3);
      if (mid) return CPMPPI_LAUNCH_V(true, 2, 2);
      if (hipError_t fe = fold_first(); fe != hipSuccess) return fe;
      return CPMPPI_LAUNCH_V(true, 2, 1);
    }
    const bool small = (uint64_t)grid.x * BLOCK <= latency_max_rollouts();
    if (small) return CPMPPI_LAUNCH_V(true, 1, 0);
    if (hipError_t fe = fold_first(); fe != hipSuccess) return fe;
    return CPMPPI_LAUNCH_V(true, 1, 1);
  }
  return CPMPPI_LAUNCH_V(false, 1, 1);
#undef CPMPPI_LAUNCH_V
}

// `prm`: the kernel-argument block of THIS launch (the handle's, or a modified copy: cost-only launches)
hipError_t launch_rollout(cpmppi_handle* h, const Params& prm, uint32_t rpl, uint32_t noise, dim3 grid, size_t lds,
                          hipStream_t s, const StepPtrs& a_in) {
  uint32_t variant = 0;
  hipError_t e;
  StepPtrs a = a_in;
  a.env_fold = h->env_fold;
  switch (prm.cost_id) {
    case CPMPPI_COST_QBGM: e = launch_rollout_math<COST_QBGM>(h->cfg.math_mode, h->cfg.ode_predictor, rpl, noise, grid, lds, s, prm, a, &variant); break;
    case CPMPPI_COST_DEFAULT: e = launch_rollout_math<COST_DEFAULT>(h->cfg.math_mode, h->cfg.ode_predictor, rpl, noise, grid, lds, s, prm, a, &variant); break;
    case CPMPPI_COST_QBG: e = launch_rollout_math<COST_QBG>(h->cfg.math_mode, h->cfg.ode_predictor, rpl, noise, grid, lds, s, prm, a, &variant); break;
    default: e = launch_rollout_math<COST_LEGACY>(h->cfg.math_mode, h->cfg.ode_predictor, rpl, noise, grid, lds, s, prm, a, &variant); break;
  }
  h->last_launch = cpmppi_launch_info{prm.cost_id, h->cfg.math_mode, noise, rpl, variant, h->cfg.ode_predictor, grid.x,
                                      prm.qb_mode == 1u ? (uint32_t)CPMPPI_COST_QB : (prm.qb_mode == 2u ? (uint32_t)CPMPPI_COST_QB_NONCONVEX : prm.cost_id)};
  return e;
}

}  // namespace

extern "C" {

const char* cpmppi_version(void) { return "cpmppi 1 gfx950 hip"; }

const char* cpmppi_last_error(const cpmppi_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int cpmppi_last_launch(const cpmppi_handle* h, cpmppi_launch_info* out) {
  if (!h || !out) return CPMPPI_ERR_BAD_ARG;
  *out = h->last_launch;
  return CPMPPI_OK;
}

int cpmppi_create(const cpmppi_config* cfg, int device, cpmppi_handle** out) try {
  if (!cfg || !out) return fail(nullptr, CPMPPI_ERR_BAD_ARG, "cpmppi_create: null argument");
  *out = nullptr;
  if (cfg->abi_version != CPMPPI_ABI_VERSION)
    return fail(nullptr, CPMPPI_ERR_ABI, "cpmppi_create: abi_version mismatch");
  if (cfg->E == 0 || cfg->N == 0 || cfg->H == 0 || cfg->S == 0 || cfg->period == 0 || cfg->H > CPMPPI_MAX_HORIZON)
    return fail(nullptr, CPMPPI_ERR_BAD_ARG, "cpmppi_create: E, N, H, S, period must be > 0 and H <= 1024");
  if (!(cfg->dt > 0.0f) || !(cfg->LBD > 0.0f) || !(cfg->NU > 0.0f) || !(cfg->L_default > 0.0f))
    return fail(nullptr, CPMPPI_ERR_BAD_ARG, "cpmppi_create: dt, LBD, NU, L_default must be > 0");
  if (cfg->cost_id > CPMPPI_COST_QB_NONCONVEX || cfg->horizon_reduce > 1 || cfg->control_mode > 1 || cfg->shift_mode > 2 ||
      cfg->correction_u > 1 || cfg->math_mode > 1 || cfg->rollouts_per_lane > 2 || cfg->ode_predictor > CPMPPI_ODE_CROMER)
    return fail(nullptr, CPMPPI_ERR_BAD_ARG, "cpmppi_create: unknown enum value");
  if ((cfg->cost_id == CPMPPI_COST_QB || cfg->cost_id == CPMPPI_COST_QB_NONCONVEX) && cfg->ode_predictor != CPMPPI_ODE_V0)
    return fail(nullptr, CPMPPI_ERR_BAD_ARG, "cpmppi_create: quadratic_boundary / _nonconvex are built for predictor_ODE_v0");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
    return fail(nullptr, CPMPPI_ERR_NO_DEVICE, "cpmppi_create: no HIP device (this library has no CPU fallback)");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess)
    return fail(nullptr, CPMPPI_ERR_NO_DEVICE, "cpmppi_create: hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, CPMPPI_ERR_NO_DEVICE,
                std::string("cpmppi_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  cpmppi_handle* h = new cpmppi_handle();
  h->cfg = *cfg;
  h->device = device;
  fill_params(*cfg, h->prm);
  h->plant_m_pole = cfg->m_pole;
  h->nb = (cfg->N + GRU_ROLLOUTS_PER_BLOCK - 1) / GRU_ROLLOUTS_PER_BLOCK;     // the finest block split in use
  refresh_dev_knobs();
  if (const char* ev = getenv("CPMPPI_FUSE_FINALIZE")) h->fuse_finalize = ev[0] != '0';
  if (const char* ev = getenv("CPMPPI_HOST_ZERO_COPY_MAX")) h->host_zero_copy_max = (uint32_t)strtoul(ev, nullptr, 10);
  DeviceGuard guard(device);                     // the caller's current device is restored on every exit path
  if (guard.err != hipSuccess) {
    delete h;
    return fail(nullptr, CPMPPI_ERR_HIP, std::string("cpmppi_create: hipSetDevice: ") + hipGetErrorString(guard.err));
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)SAMPLER_LDS_MAX);
  // the adjoint kernel parks S x 6 x 256 sub-states in LDS (61 KB at S = 10; more substeps need the opt-in as well)
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_grad_kernel<COST_QBGM>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)SAMPLER_LDS_MAX);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_grad_kernel<COST_DEFAULT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)SAMPLER_LDS_MAX);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_grad_kernel<COST_QBG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)SAMPLER_LDS_MAX);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_grad_kernel<COST_QBGM, PREDICTOR_ODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)SAMPLER_LDS_MAX);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_grad_kernel<COST_DEFAULT, PREDICTOR_ODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)SAMPLER_LDS_MAX);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_grad_kernel<COST_QBG, PREDICTOR_ODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)SAMPLER_LDS_MAX);
  // the CEM top-k sorts N (padded to a power of two) 8-byte records in LDS: 128 KB at N = 16384
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cem_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)SAMPLER_LDS_MAX);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sample_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)SAMPLER_LDS_MAX);
  const uint32_t Wmax = cfg->H > h->prm.P ? cfg->H : h->prm.P;
  h->workspace_floats = (size_t)cfg->E * h->nb * (2 + Wmax);
  h->workspace = nullptr;
  hipError_t e = hipMalloc(&h->workspace, h->workspace_floats * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&h->counters, (size_t)cfg->E * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(h->counters, 0, (size_t)cfg->E * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&h->env_fold, (size_t)cfg->E * sizeof(EnvFold));
  if (e == hipSuccess) e = hipMalloc(&h->zeros_H, (size_t)cfg->E * cfg->H * sizeof(float));
  if (e == hipSuccess) e = hipMemset(h->zeros_H, 0, (size_t)cfg->E * cfg->H * sizeof(float));
  if (e != hipSuccess) {
    std::string msg = std::string("cpmppi_create: hipMalloc workspace: ") + hipGetErrorString(e);
    delete h;
    return fail(nullptr, CPMPPI_ERR_HIP, msg);
  }
  *out = h;
  return CPMPPI_OK;
} catch (const std::exception&) { return CPMPPI_ERR_NOMEM; }   // (no C++ exception leaves the C ABI)

void cpmppi_destroy(cpmppi_handle* h) {
  if (!h) return;
  DeviceGuard guard(h->device);                    // (frees, the side stream and the communicator belong to the handle's device)
  if (h->comm) { cpmppi_comm::destroy(h->comm); h->comm = nullptr; }
  if (h->workspace) (void)hipFree(h->workspace);
  if (h->gru_image) (void)hipFree(h->gru_image);
  if (h->gru16_image) (void)hipFree(h->gru16_image);
  if (h->grad_ckpt) (void)hipFree(h->grad_ckpt);
  if (h->counters) (void)hipFree(h->counters);
  if (h->env_fold) (void)hipFree(h->env_fold);
  if (h->zeros_H) (void)hipFree(h->zeros_H);
  if (h->host_stage) (void)hipHostFree(h->host_stage);
  if (h->dev_stage) (void)hipFree(h->dev_stage);
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  delete h;
}

int cpmppi_get_config(const cpmppi_handle* h, cpmppi_config* out) {
  if (!h || !out) return CPMPPI_ERR_BAD_ARG;
  *out = h->cfg;
  return CPMPPI_OK;
}

int cpmppi_set_cost_weights(cpmppi_handle* h, uint32_t cost_id, const float* cost_w, uint32_t n) {
  if (!h || !cost_w || n > 24 || cost_id > CPMPPI_COST_QB_NONCONVEX)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_set_cost_weights: bad argument");
  const bool qb = cost_id == CPMPPI_COST_QB || cost_id == CPMPPI_COST_QB_NONCONVEX;   // default.py's kernels + sub-mode (fill_params)
  if (qb && h->cfg.ode_predictor != CPMPPI_ODE_V0)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_set_cost_weights: quadratic_boundary / _nonconvex are built for predictor_ODE_v0");
  h->cfg.cost_id = cost_id;
  h->prm.cost_id = qb ? (uint32_t)CPMPPI_COST_DEFAULT : cost_id;
  h->prm.qb_mode = qb ? (cost_id == CPMPPI_COST_QB ? 1u : 2u) : 0u;
  for (uint32_t i = 0; i < n; ++i) h->cfg.cost_w[i] = h->prm.w[i] = cost_w[i];
  return CPMPPI_OK;
}

int cpmppi_set_pole_mass(cpmppi_handle* h, float m_pole) {
  if (!h || !(m_pole > 0.0f) || !(m_pole < INFINITY)) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_set_pole_mass: m_pole must be a positive number");
  // the CONTROLLER's belief (CartPole/__init__.py:516 sends m_pole_for_controller): the predictor / cost kernels compute with it.
  // The PLANT of cpmppi_plant_advance* is the simulated system itself and keeps the mass the handle was created with
  // (plant_m_pole): a handle that serves as both must not change the plant by updating the controller's attribute.
  h->cfg.m_pole = m_pole;
  h->prm.m_pole = m_pole;
  return CPMPPI_OK;
}

int cpmppi_sample(cpmppi_handle* h, uint32_t E, uint64_t seed, uint64_t offset, uint32_t env_offset, float* knots_out,
                  float* delta_u_out, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || E > h->cfg.E || (!knots_out && !delta_u_out))
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_sample: E out of range or no output buffer");
  if (misaligned(knots_out) || misaligned(delta_u_out)) return fail(h, CPMPPI_ERR_ALIGN, "cpmppi_sample: misaligned");
  if ((size_t)BLOCK * (h->prm.P + 1) * sizeof(float) > SAMPLER_LDS_MAX)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_sample: more than 154 knots per rollout are not supported");
  CPMPPI_ON_DEVICE(h);
  const size_t rows = (size_t)E * h->cfg.N;
  hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((rows + BLOCK - 1) / BLOCK)), dim3(BLOCK),
                     (size_t)BLOCK * (h->prm.P + 1) * sizeof(float), (hipStream_t)stream, h->prm, E, seed, offset,
                     env_offset, (const float*)nullptr, knots_out, delta_u_out);
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

int cpmppi_interpolate(cpmppi_handle* h, uint32_t E, const float* knots, float* delta_u_out, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || E > h->cfg.E || !knots || !delta_u_out)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_interpolate: bad argument");
  if (misaligned(knots) || misaligned(delta_u_out)) return fail(h, CPMPPI_ERR_ALIGN, "cpmppi_interpolate: misaligned");
  if ((size_t)BLOCK * (h->prm.P + 1) * sizeof(float) > SAMPLER_LDS_MAX)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_interpolate: more than 154 knots per rollout are not supported");
  CPMPPI_ON_DEVICE(h);
  const size_t rows = (size_t)E * h->cfg.N;
  hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((rows + BLOCK - 1) / BLOCK)), dim3(BLOCK),
                     (size_t)BLOCK * (h->prm.P + 1) * sizeof(float), (hipStream_t)stream, h->prm, E, (uint64_t)0,
                     (uint64_t)0, 0u, knots, (float*)nullptr, delta_u_out);
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

size_t cpmppi_tiled_floats(const cpmppi_handle* h, uint32_t E) {
  if (!h) return 0;
  return (size_t)E * ((h->cfg.N + 63u) / 64u) * ((h->cfg.H + 3u) / 4u) * 256u;
}

int cpmppi_sample_tiled(cpmppi_handle* h, uint32_t E, uint64_t seed, uint64_t offset, uint32_t env_offset,
                        const float* knots_in, float* tiled_out, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || E > h->cfg.E || !tiled_out) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_sample_tiled: bad argument");
  if ((reinterpret_cast<uintptr_t>(tiled_out) & 15u) != 0 || misaligned(knots_in))
    return fail(h, CPMPPI_ERR_ALIGN, "cpmppi_sample_tiled: tiled_out must be 16-byte aligned");
  if ((size_t)BLOCK * (h->prm.P + 1) * sizeof(float) > SAMPLER_LDS_MAX)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_sample_tiled: more than 154 knots per rollout are not supported");
  CPMPPI_ON_DEVICE(h);
  const size_t groups = (size_t)E * ((h->cfg.N + 63u) / 64u);
  hipLaunchKernelGGL(sample_tiled_kernel, dim3((unsigned)((groups + WAVES - 1) / WAVES)), dim3(BLOCK),
                     (size_t)BLOCK * (h->prm.P + 1) * sizeof(float), (hipStream_t)stream, h->prm, E, seed, offset,
                     env_offset, knots_in, tiled_out);
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

int cpmppi_tile_delta_u(cpmppi_handle* h, uint32_t E, const float* delta_u, float* tiled_out, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || E > h->cfg.E || !delta_u || !tiled_out) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_tile_delta_u: bad argument");
  if ((reinterpret_cast<uintptr_t>(tiled_out) & 15u) != 0 || misaligned(delta_u))
    return fail(h, CPMPPI_ERR_ALIGN, "cpmppi_tile_delta_u: tiled_out must be 16-byte aligned");
  CPMPPI_ON_DEVICE(h);
  const size_t groups = (size_t)E * ((h->cfg.N + 63u) / 64u);
  hipLaunchKernelGGL(tile_kernel, dim3((unsigned)((groups + WAVES - 1) / WAVES)), dim3(BLOCK), 0, (hipStream_t)stream,
                     h->prm, E, delta_u, tiled_out);
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

int cpmppi_predict(cpmppi_handle* h, uint32_t B, uint32_t horizon, const float* s0, const float* Q, const float* L,
                   float* traj_out, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (horizon == 0) horizon = h->cfg.H;
  if (B == 0 || !s0 || !Q || !traj_out) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_predict: bad argument");
  if (misaligned(s0) || misaligned(Q) || misaligned(L) || misaligned(traj_out))
    return fail(h, CPMPPI_ERR_ALIGN, "cpmppi_predict: misaligned pointer");
  CPMPPI_ON_DEVICE(h);
  const dim3 grid((B + BLOCK - 1) / BLOCK);
  // stores staged through LDS once the launch puts more than one wave on every SIMD (below that the direct stores'
  // shorter path wins: measured 59 vs 82 us at 1024 rollouts, 416 vs 280 us at 262144)
  const bool staged = (uint64_t)B > 65536ull;
  const hipStream_t st = (hipStream_t)stream;
#define CPMPPI_PREDICT_LAUNCH(FAST, STAGED, INTEG) \
  hipLaunchKernelGGL((predict_kernel<FAST, STAGED, INTEG>), grid, dim3(BLOCK), 0, st, h->prm, B, horizon, s0, Q, L, traj_out)
  const bool fast = h->cfg.math_mode == CPMPPI_MATH_FAST;
  if (h->cfg.ode_predictor == CPMPPI_ODE_CROMER) {
    if (fast) { if (staged) CPMPPI_PREDICT_LAUNCH(true, true, PREDICTOR_ODE); else CPMPPI_PREDICT_LAUNCH(true, false, PREDICTOR_ODE); }
    else { if (staged) CPMPPI_PREDICT_LAUNCH(false, true, PREDICTOR_ODE); else CPMPPI_PREDICT_LAUNCH(false, false, PREDICTOR_ODE); }
  } else {
    if (fast) { if (staged) CPMPPI_PREDICT_LAUNCH(true, true, PREDICTOR_ODE_V0); else CPMPPI_PREDICT_LAUNCH(true, false, PREDICTOR_ODE_V0); }
    else { if (staged) CPMPPI_PREDICT_LAUNCH(false, true, PREDICTOR_ODE_V0); else CPMPPI_PREDICT_LAUNCH(false, false, PREDICTOR_ODE_V0); }
  }
#undef CPMPPI_PREDICT_LAUNCH
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

int cpmppi_trajectory_cost(cpmppi_handle* h, uint32_t B, uint32_t horizon, const float* traj, const float* inputs,
                           float target_position, float target_equilibrium, const float* u_nom, const float* u_prev,
                           float* stage_out, float* terminal_out, float* total_out, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (horizon == 0) horizon = h->cfg.H;
  if (B == 0 || !traj || !inputs || (!stage_out && !terminal_out && !total_out))
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_trajectory_cost: bad argument");
  if (h->prm.cost_id == CPMPPI_COST_LEGACY && !u_nom)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_trajectory_cost: legacy cost needs u_nom");
  CPMPPI_ON_DEVICE(h);
  // rows per wave: one for the reference's call shape (a few thousand rollouts: spread them over the chip), up to 64
  // once there are more rows than ~8 waves per SIMD can take one each
  uint32_t rpw = (uint32_t)(((uint64_t)B + 8191) / 8192);
  rpw = rpw < 1 ? 1 : (rpw > 64 ? 64 : rpw);
  const uint32_t waves = (B + rpw - 1) / rpw;
  hipLaunchKernelGGL(trajectory_cost_kernel, dim3((waves + WAVES - 1) / WAVES), dim3(BLOCK), 0, (hipStream_t)stream, h->prm,
                     B, horizon, rpw, traj, inputs, target_position, target_equilibrium, u_nom, u_prev, stage_out,
                     terminal_out, total_out);
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

static int step_impl(cpmppi_handle* h, const cpmppi_step_args* a, void* stream, uint32_t* host_ticket,
                     const cpmppi_comm::GatherTicket* gather = nullptr);

int cpmppi_step(cpmppi_handle* h, const cpmppi_step_args* a, void* stream) { return step_impl(h, a, stream, nullptr); }

// The step and the all-gather of its result in ONE call (contract in cpmppi.h; mechanism in cpmppi_comm.hip): the launch
// stream gets the rollout kernel and nothing else; the side stream gets waiter -> ncclAllGather -> post.
int cpmppi_step_gather(cpmppi_handle* h, const cpmppi_step_args* a, float* recv_all, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (!a || !recv_all) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_step_gather: null argument");
  if (!h->comm) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_step_gather: no communicator (cpmppi_comm_init)");
  if (a->E == 0 || a->E > h->cfg.E || !a->u_nom) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_step_gather: bad step arguments");
  // a wait on the device gave up (a peer rank stalled beyond cpmppi_comm_set_timeout): the steps since then have dropped their
  // results and the gathered blocks are not to be used - say so NOW, not at a cpmppi_comm_sync the caller may never make
  if (cpmppi_comm::comm_error_pending(h))
    return fail(h, CPMPPI_ERR_COMM, "cpmppi_step_gather: an earlier step's device-side wait for an all-gather timed out; "
                                    "cpmppi_comm_sync reports and clears the condition");
  // a stream being captured would bake THIS step's number into the graph: every replay would re-publish it and the side
  // stream's wait for the next step could never be satisfied (advisor, round 4)
  {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
      return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_step_gather: the launch stream is being captured; a step with its all-gather "
                                         "carries a step number and cannot be part of a graph");
    (void)hipGetLastError();
  }
  const bool in_place = !a->u_nom_out || a->u_nom_out == a->u_nom;
  cpmppi_comm::GatherTicket t;
  cpmppi_comm::begin_step_gather(h->comm, in_place ? a->u_nom : a->u_nom_out, &t);
  int rc = cpmppi_comm::enqueue_guard(h, t, a->E, stream);
  if (rc == CPMPPI_OK) rc = step_impl(h, a, stream, nullptr, &t);
  if (rc != CPMPPI_OK) {
    cpmppi_comm::abort_step_gather(h->comm);
    return rc;
  }
  return cpmppi_comm::enqueue_gather(h, in_place ? a->u_nom : a->u_nom_out, recv_all, (size_t)a->E * h->cfg.H);
}

static int step_impl(cpmppi_handle* h, const cpmppi_step_args* a, void* stream, uint32_t* host_ticket,
                     const cpmppi_comm::GatherTicket* gather) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (!a) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_step: null args");
  if (a->E == 0 || a->E > h->cfg.E) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_step: E out of range");
  if (!a->s0 || !a->u_nom || !a->target_position || !a->target_equilibrium)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_step: s0, u_nom, target_position, target_equilibrium are required");
  if (a->noise_kind > CPMPPI_NOISE_DELTA_U_TILED) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_step: unknown noise_kind");
  if (a->noise_kind == CPMPPI_NOISE_DELTA_U_TILED && (reinterpret_cast<uintptr_t>(a->noise) & 15u) != 0)
    return fail(h, CPMPPI_ERR_ALIGN, "cpmppi_step: the tiled perturbation buffer must be 16-byte aligned");
  if (a->noise_kind == CPMPPI_NOISE_DELTA_U_TILED && a->predictor == CPMPPI_PREDICTOR_GRU)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_step: the GRU predictor takes delta_u, knots or Philox noise");
  if (a->noise_kind != CPMPPI_NOISE_PHILOX && !a->noise)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_step: noise buffer required for this noise_kind");
  if (misaligned(a->s0) || misaligned(a->u_nom) || misaligned(a->noise) || misaligned(a->S_out) ||
      misaligned(a->Q_out) || misaligned(a->u_nom_out))
    return fail(h, CPMPPI_ERR_ALIGN, "cpmppi_step: misaligned pointer");
  // (every check that can fail comes BEFORE the event recorder is touched: a failed step must not leave a half-recorded
  // bracket behind for cpmppi_get_profile)
  if (a->predictor > CPMPPI_PREDICTOR_GRU) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_step: unknown predictor");
  if (a->predictor == CPMPPI_PREDICTOR_GRU) {
    if (!h->gru_image) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_step: predictor GRU requested but no model set (cpmppi_set_gru)");
    if ((h->prm.cost_id != CPMPPI_COST_QBGM && h->prm.cost_id != CPMPPI_COST_DEFAULT) || h->prm.qb_mode != 0u)
      return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_step: the GRU predictor supports quadratic_boundary_grad_minimal and default");
  }
  CPMPPI_ON_DEVICE(h);
  StepPtrs p{};
  p.s0 = a->s0; p.u_nom = a->u_nom; p.u_prev = a->u_prev; p.x_t = a->target_position; p.te = a->target_equilibrium;
  p.L = a->L; p.noise = a->noise; p.prev_in = a->previous_input; p.seed = a->seed; p.offset = a->offset; p.env_offset = a->env_offset;
  p.offset_dev = (a->noise_kind == CPMPPI_NOISE_PHILOX) ? (const unsigned long long*)a->offset_dev : nullptr;
  // lane mapping: two rollouts per lane (packed float2) once the launch fills every SIMD with at least one such wave
  // (1024 SIMDs x 128 rollouts); one rollout per lane (shortest critical path) below.  Measured at 128 envs x 1024 x 50:
  // 76 us packed vs 90 us one per lane; at 64 envs the packed mapping would leave half the SIMDs empty.
  uint32_t rpl = h->cfg.rollouts_per_lane;
  if (h->cfg.math_mode != CPMPPI_MATH_FAST) rpl = 1;
  else if (rpl == 0) rpl = ((uint64_t)a->E * h->cfg.N >= PACKED_MIN_ROLLOUTS) ? 2 : 1;
  p.nb = (h->cfg.N + BLOCK * rpl - 1) / (BLOCK * rpl);
  uint32_t noise_kind = a->noise_kind;
  const hipStream_t s = (hipStream_t)stream;
  const bool du_space = (noise_kind == CPMPPI_NOISE_DELTA_U || noise_kind == CPMPPI_NOISE_DELTA_U_TILED);
  p.W = du_space ? h->cfg.H : h->prm.P;
  p.S_out = a->S_out; p.partial = h->workspace;
  p.counter = nullptr; p.u_nom_out = a->u_nom_out ? a->u_nom_out : a->u_nom; p.Q_out = a->Q_out;
  p.host_ticket = host_ticket;
  p.gs = gather ? GatherSync{gather->flags, gather->publish, gather->need, gather->envs ? gather->envs : a->E} : GatherSync{nullptr, 0u, 0u, 0u};
  hipEvent_t* ev = nullptr;
  const uint32_t group_pos = h->profile_every ? h->profile_count++ % h->profile_every : 0;
  const bool grouped = h->profile_every > 1;
  if (h->profile_every && group_pos == 0) {
    if (h->ev_used + 3 > h->ev.size()) {
      for (int i = 0; i < 3; ++i) {
        hipEvent_t e;
        CPMPPI_HIP(h, hipEventCreate(&e));
        h->ev.push_back(e);
      }
    }
    ev = &h->ev[h->ev_used];
    if (h->ev_tail.size() < h->ev.size() / 3) h->ev_tail.resize(h->ev.size() / 3, 0);
    h->ev_used += 3;
    h->group_open = grouped;
    CPMPPI_HIP(h, hipEventRecord(ev[0], s));
  }
  if (grouped) ev = nullptr;                       // (no events inside a group)
  if (a->predictor == CPMPPI_PREDICTOR_GRU) {
    p.nb = (h->cfg.N + GRU_ROLLOUTS_PER_BLOCK - 1) / GRU_ROLLOUTS_PER_BLOCK;
    // FAST: float32-equivalent split products on the f16 matrix cores (cpmppi_gru16.hpp); PRECISE: exact f32 MFMA chains
    const bool f16 = h->cfg.math_mode == CPMPPI_MATH_FAST && h->gru16_image != nullptr;
    const size_t lds = ((f16 ? (size_t)G16_IMAGE_BYTES / 4 : (size_t)GRU_IMAGE_FLOATS) + (size_t)WAVES * p.W) * sizeof(float);
    const dim3 grid(a->E * p.nb);
#define CPMPPI_GRU_LAUNCH(COST, NOISE)                                                                                  \
    do {                                                                                                                \
      if (lds > 64 * 1024) {   /* long horizons with a perturbation buffer: weights image + [WAVES][H] sums */          \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_rollout_cost_kernel<COST, NOISE, true>),          \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)SAMPLER_LDS_MAX);                    \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_rollout_cost_kernel<COST, NOISE, false>),         \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)SAMPLER_LDS_MAX);                    \
      }                                                                                                                 \
      if (f16) hipLaunchKernelGGL((gru_rollout_cost_kernel<COST, NOISE, true>), grid, dim3(BLOCK), lds, s, h->prm, p,  \
                                  h->gru_norm, (const float*)h->gru16_image, a->h0);                                    \
      else hipLaunchKernelGGL((gru_rollout_cost_kernel<COST, NOISE, false>), grid, dim3(BLOCK), lds, s, h->prm, p,     \
                              h->gru_norm, (const float*)h->gru_image, a->h0);                                          \
    } while (0)
    const bool q = h->prm.cost_id == CPMPPI_COST_QBGM;
    if (a->noise_kind == CPMPPI_NOISE_DELTA_U) { if (q) CPMPPI_GRU_LAUNCH(COST_QBGM, NOISE_DELTA_U); else CPMPPI_GRU_LAUNCH(COST_DEFAULT, NOISE_DELTA_U); }
    else if (a->noise_kind == CPMPPI_NOISE_KNOTS) { if (q) CPMPPI_GRU_LAUNCH(COST_QBGM, NOISE_KNOTS); else CPMPPI_GRU_LAUNCH(COST_DEFAULT, NOISE_KNOTS); }
    else { if (q) CPMPPI_GRU_LAUNCH(COST_QBGM, NOISE_PHILOX); else CPMPPI_GRU_LAUNCH(COST_DEFAULT, NOISE_PHILOX); }
#undef CPMPPI_GRU_LAUNCH
    CPMPPI_HIP(h, hipGetLastError());
  } else {
    p.counter = h->fuse_finalize ? h->counters : nullptr;
    size_t lds = (size_t)WAVES * p.W * sizeof(float);
    p.stash = 0;
    if (a->noise_kind == CPMPPI_NOISE_PHILOX) {                 // park the generated knots in LDS when they fit
      const size_t park = (size_t)p.W * rpl * BLOCK * sizeof(float);
      if (lds + park <= 32 * 1024) { p.stash = 1; lds += park; }
    }
    CPMPPI_HIP(h, launch_rollout(h, h->prm, rpl, noise_kind, dim3(a->E * p.nb), lds, s, p));
  }
  const bool separate_finalize = (p.counter == nullptr);
  if (ev) CPMPPI_HIP(h, hipEventRecord(ev[1], s));
  if (separate_finalize) {
    if (du_space)
      hipLaunchKernelGGL(finalize_kernel<false>, dim3(a->E), dim3(BLOCK), 0, s, h->prm, (const float*)h->workspace,
                         p.nb, p.W, (const float*)a->u_nom, p.u_nom_out, a->Q_out, p.gs);
    else
      hipLaunchKernelGGL(finalize_kernel<true>, dim3(a->E), dim3(BLOCK), 0, s, h->prm, (const float*)h->workspace,
                         p.nb, p.W, (const float*)a->u_nom, p.u_nom_out, a->Q_out, p.gs);
  }
  CPMPPI_HIP(h, hipGetLastError());
  if (p.offset_dev) {
    hipLaunchKernelGGL(bump_counter_kernel, dim3(1), dim3(1), 0, s, (unsigned long long*)a->offset_dev);
    CPMPPI_HIP(h, hipGetLastError());
  }
  if (ev) {
    // an event costs ~5 us on the stream: the third one only if something ran after the rollout kernel
    const bool tail = separate_finalize || p.offset_dev;
    h->ev_tail[(size_t)(ev - h->ev.data()) / 3] = tail ? 1 : 0;
    if (tail) CPMPPI_HIP(h, hipEventRecord(ev[2], s));
  }
  if (grouped && h->group_open && group_pos == h->profile_every - 1) {     // the group's last step: close the bracket
    hipEvent_t* g = &h->ev[h->ev_used - 3];
    h->ev_tail[(h->ev_used - 3) / 3] = 0;
    CPMPPI_HIP(h, hipEventRecord(g[1], s));
    h->group_open = false;
  }
  return CPMPPI_OK;
}

int cpmppi_step_host(cpmppi_handle* h, uint32_t E, const float* s0, const float* target_position,
                     const float* target_equilibrium, const float* L, float* u_nom, uint64_t seed, uint64_t offset,
                     uint32_t env_offset, float* Q, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || E > h->cfg.E || !s0 || !target_position || !target_equilibrium || !u_nom || !Q)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_step_host: bad argument");
  const auto tp0 = std::chrono::steady_clock::now();
  CPMPPI_ON_DEVICE(h);
  // pinned, fine-grained, device-mapped block: [cfg.E * 10 floats: state 6, target, equilibrium, L | Q][ticket]
  const size_t cap = (size_t)h->cfg.E * 10;
  if (!h->host_stage) {
    CPMPPI_HIP(h, hipHostMalloc((void**)&h->host_stage, (cap + 16) * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
    memset(h->host_stage, 0, (cap + 16) * sizeof(float));
  }
  const hipStream_t st = (hipStream_t)stream;
  float* hs = h->host_stage;
  memcpy(hs, s0, (size_t)E * 6 * sizeof(float));
  memcpy(hs + 6 * E, target_position, (size_t)E * sizeof(float));
  memcpy(hs + 7 * E, target_equilibrium, (size_t)E * sizeof(float));
  for (uint32_t e = 0; e < E; ++e) hs[8 * E + e] = L ? L[e] : h->cfg.L_default;
  cpmppi_step_args a{};
  a.E = E; a.u_nom = u_nom;
  a.noise_kind = CPMPPI_NOISE_PHILOX; a.seed = seed; a.offset = offset; a.env_offset = env_offset;
  if (h->fuse_finalize && E <= h->host_zero_copy_max) {
    // Few envs (the simulator's own call: one): no copy packets and no stream wait at all.  The kernel reads the 9 floats
    // per env straight from the pinned block over the host link, the env's finalizing block stores Q into the same
    // block and bumps a system-scope ticket; this thread spins on the ticket (a stream wait is an interrupt + a thread
    // wake-up: ~15 us of an 80 us control step).  The stream is polled now and then so that a failed launch cannot hang the caller.
    uint32_t* ticket = reinterpret_cast<uint32_t*>(hs + cap);
    const auto tp1 = std::chrono::steady_clock::now();
    a.s0 = hs; a.target_position = hs + 6 * E; a.target_equilibrium = hs + 7 * E; a.L = hs + 8 * E; a.Q_out = hs + 9 * E;
    const uint32_t target = (h->host_ticket_value += E);
    const int rc = step_impl(h, &a, stream, ticket);
    if (rc != CPMPPI_OK) { h->host_ticket_value -= E; return rc; }
    const auto tp2 = std::chrono::steady_clock::now();
    for (uint32_t spins = 1;; ++spins) {
      if (__atomic_load_n(ticket, __ATOMIC_ACQUIRE) == target) break;
      __builtin_ia32_pause();
      if ((spins & 0xFFFu) == 0u) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipErrorNotReady) continue;
        if (q != hipSuccess) return fail(h, CPMPPI_ERR_HIP, std::string("cpmppi_step_host: ") + hipGetErrorString(q));
        if (__atomic_load_n(ticket, __ATOMIC_ACQUIRE) == target) break;
        // the stream drained without the ticket: resynchronise the counter and report
        h->host_ticket_value = __atomic_load_n(ticket, __ATOMIC_ACQUIRE);
        return fail(h, CPMPPI_ERR_HIP, "cpmppi_step_host: the launch completed without delivering its controls");
      }
    }
    memcpy(Q, hs + 9 * E, (size_t)E * sizeof(float));
    const auto tp3 = std::chrono::steady_clock::now();
    h->host_t[0] += std::chrono::duration<double>(tp1 - tp0).count();
    h->host_t[1] += std::chrono::duration<double>(tp2 - tp1).count();
    h->host_t[2] += std::chrono::duration<double>(tp3 - tp2).count();
    h->host_t[3] += 1.0;
    return CPMPPI_OK;
  }
  if (!h->dev_stage) CPMPPI_HIP(h, hipMalloc((void**)&h->dev_stage, cap * sizeof(float)));
  float* d = h->dev_stage;
  CPMPPI_HIP(h, hipMemcpyAsync(d, hs, (size_t)E * 9 * sizeof(float), hipMemcpyHostToDevice, st));
  a.s0 = d; a.target_position = d + 6 * E; a.target_equilibrium = d + 7 * E; a.L = d + 8 * E; a.Q_out = d + 9 * E;
  const int rc = cpmppi_step(h, &a, stream);
  if (rc != CPMPPI_OK) return rc;
  CPMPPI_HIP(h, hipMemcpyAsync(hs + 9 * E, d + 9 * E, (size_t)E * sizeof(float), hipMemcpyDeviceToHost, st));
  CPMPPI_HIP(h, hipStreamSynchronize(st));
  memcpy(Q, hs + 9 * E, (size_t)E * sizeof(float));
  return CPMPPI_OK;
}

// development aid (tools/dev/seam_latency.py; not part of the contract): mean seconds per zero-copy cpmppi_step_host call
// spent staging the inputs, inside the launch call, and spinning on the ticket; the number of calls; resets the sums
int cpmppi_debug_host_times(cpmppi_handle* h, double out[4]) {
  if (!h || !out) return CPMPPI_ERR_BAD_ARG;
  const double n = h->host_t[3] > 0 ? h->host_t[3] : 1.0;
  for (int i = 0; i < 3; ++i) out[i] = h->host_t[i] / n;
  out[3] = h->host_t[3];
  for (double& v : h->host_t) v = 0.0;
  return CPMPPI_OK;
}

int cpmppi_set_profiling(cpmppi_handle* h, int enable) {
  if (!h || enable < 0) return CPMPPI_ERR_BAD_ARG;
  h->profile_every = (uint32_t)enable;
  h->profile_count = 0;
  h->ev_used = 0;
  h->group_open = false;
  return CPMPPI_OK;
}

int cpmppi_get_profile(cpmppi_handle* h, float* rollout_ms, float* finalize_ms, uint32_t max_steps, uint32_t* n_steps) {
  if (!h || !n_steps) return CPMPPI_ERR_BAD_ARG;
  if (h->group_open) { h->ev_used -= 3; h->group_open = false; }       // an unfinished group has no closing event
  const uint32_t n = (uint32_t)(h->ev_used / 3);
  const float per = h->profile_every > 1 ? 1.0f / (float)h->profile_every : 1.0f;
  *n_steps = n;
  for (uint32_t i = 0; i < n && i < max_steps; ++i) {
    hipEvent_t* ev = &h->ev[(size_t)i * 3];
    const bool tail = h->ev_tail[i] != 0;
    CPMPPI_HIP(h, hipEventSynchronize(ev[tail ? 2 : 1]));
    float a = 0.f, b = 0.f;
    CPMPPI_HIP(h, hipEventElapsedTime(&a, ev[0], ev[1]));
    if (tail) CPMPPI_HIP(h, hipEventElapsedTime(&b, ev[1], ev[2]));
    if (rollout_ms) rollout_ms[i] = a * per;
    if (finalize_ms) finalize_ms[i] = b;
  }
  h->ev_used = 0;
  h->profile_count = 0;
  return CPMPPI_OK;
}

int cpmppi_set_gru(cpmppi_handle* h, const cpmppi_gru_model* m) try {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (!m || m->hidden != 32 || m->layers != 2 || m->inputs != 6 || m->outputs != 5)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_set_gru: only GRU-6IN-32H1-32H2-5OUT is built");
  for (int l = 0; l < 2; ++l)
    if (!m->w_ih[l] || !m->w_hh[l] || !m->b_ih[l] || !m->b_hh[l]) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_set_gru: null weights");
  if (!m->w_out || !m->b_out) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_set_gru: null head weights");
  std::vector<float> img((size_t)GRU_IMAGE_FLOATS, 0.0f);
  auto frag = [&](int f) { return img.data() + (size_t)f * 64; };
  // x-tile row r -> network input column: rows 0..4 = the 5 state features (inputs 1..5), row 5 = Q (input 0)
  auto xcol = [](int r) { return r < 5 ? r + 1 : (r == 5 ? 0 : -1); };
  for (int g = 0; g < 3; ++g) {
    for (int s = 0; s < 4; ++s)
      for (int l = 0; l < 64; ++l) {
        const int col = xcol(gru_tile_row(s, l >> 5));
        frag(GF_L1X + g * 4 + s)[l] = col < 0 ? 0.0f : m->w_ih[0][(size_t)(g * 32 + (l & 31)) * 6 + col];
      }
    for (int s = 0; s < 16; ++s)
      for (int l = 0; l < 64; ++l) {
        const int k = gru_tile_row(s, l >> 5);
        const size_t row = (size_t)(g * 32 + (l & 31));
        frag(GF_L1H + g * 16 + s)[l] = m->w_hh[0][row * 32 + k];
        frag(GF_L2X + g * 16 + s)[l] = m->w_ih[1][row * 32 + k];
        frag(GF_L2H + g * 16 + s)[l] = m->w_hh[1][row * 32 + k];
      }
  }
  for (int layer = 0; layer < 2; ++layer) {
    const int fb = layer == 0 ? GF_L1B : GF_L2B;
    for (int l = 0; l < 32; ++l) {                           // lane-half 0 carries the bias (k = 0), half 1 zeros
      frag(fb + 0)[l] = m->b_ih[layer][l] + m->b_hh[layer][l];
      frag(fb + 1)[l] = m->b_ih[layer][32 + l] + m->b_hh[layer][32 + l];
      frag(fb + 2)[l] = m->b_ih[layer][64 + l];
      frag(fb + 3)[l] = m->b_hh[layer][64 + l];
    }
  }
  for (int s = 0; s < 16; ++s)
    for (int l = 0; l < 64; ++l)
      frag(GF_DW + s)[l] = (l & 31) < 5 ? m->w_out[(size_t)(l & 31) * 32 + gru_tile_row(s, l >> 5)] : 0.0f;
  for (int l = 0; l < 5; ++l) frag(GF_DB)[l] = m->b_out[l];
  // plain vectors of the fused rollout kernel (biases as accumulator tiles, dense head on the VALU)
  for (int layer = 0; layer < 2; ++layer)
    for (int hf = 0; hf < 2; ++hf)
      for (int v = 0; v < 16; ++v) {
        const int r = gru_tile_row(v, hf);
        float* b = img.data() + GV_BIAS + (size_t)layer * 4 * 32 + hf * 16 + v;
        b[0 * 32] = m->b_ih[layer][r] + m->b_hh[layer][r];
        b[1 * 32] = m->b_ih[layer][32 + r] + m->b_hh[layer][32 + r];
        b[2 * 32] = m->b_ih[layer][64 + r];
        b[3 * 32] = m->b_hh[layer][64 + r];
      }
  for (int hf = 0; hf < 2; ++hf)
    for (int o = 0; o < 5; ++o)
      for (int v = 0; v < 16; ++v)
        img[GV_HEAD + (size_t)hf * 80 + o * 16 + v] = m->w_out[(size_t)o * 32 + gru_tile_row(v, hf)];
  for (int o = 0; o < 5; ++o) img[GV_HEADB + o] = m->b_out[o];
  for (int i = 0; i < 6; ++i) {
    h->gru_norm.in_scale[i] = m->in_scale ? m->in_scale[i] : 1.0f;
    h->gru_norm.in_shift[i] = m->in_shift ? m->in_shift[i] : 0.0f;
  }
  for (int i = 0; i < 5; ++i) {
    h->gru_norm.out_scale[i] = m->out_scale ? m->out_scale[i] : 1.0f;
    h->gru_norm.out_shift[i] = m->out_shift ? m->out_shift[i] : 0.0f;
  }
  // ---- f16 split image (cpmppi_gru16.hpp): fragment f holds, for lane l and t = 0..7, the weight of output row l%32
  // against k-slot (block b, lane half l/32, t) = tile register v = 8b + t of that half
  std::vector<unsigned char> img16((size_t)G16_IMAGE_BYTES, 0);
  bool in_range = true;
  auto put16 = [&](int f, int lane, int t, float w) {
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)(w - (float)hi);
    if (!(fabsf(w) < 60000.0f)) in_range = false;
    reinterpret_cast<_Float16*>(img16.data() + (size_t)f * G16_FRAG_BYTES + lane * 16)[t] = hi;
    reinterpret_cast<_Float16*>(img16.data() + (size_t)(f + 1) * G16_FRAG_BYTES + lane * 16)[t] = lo;
  };
  // gate rows pre-scaled so that the gates need no multiply before v_exp_f32 (gru16_gates): r, z by -log2(e), n by 2 log2(e)
  const double LOG2E = 1.4426950408889634;
  const double gate_scale[3] = {-LOG2E, -LOG2E, 2.0 * LOG2E};
  auto sc = [&](int g, float w) { return (float)(gate_scale[g] * (double)w); };
  for (int g = 0; g < 3; ++g)
    for (int l = 0; l < 64; ++l)
      for (int tt = 0; tt < 8; ++tt) {
        const size_t row = (size_t)(g * 32 + (l & 31));
        const int col = xcol(gru_tile_row(tt, l >> 5));                       // x tile registers 0..7
        put16(HF_L1X + g * 2, l, tt, (col < 0 || tt >= 4) ? 0.0f : sc(g, m->w_ih[0][row * 6 + col]));
        for (int b = 0; b < 2; ++b) {
          const int k = gru_tile_row(8 * b + tt, l >> 5);
          put16(HF_L1H + (g * 2 + b) * 2, l, tt, sc(g, m->w_hh[0][row * 32 + k]));
          put16(HF_L2X + (g * 2 + b) * 2, l, tt, sc(g, m->w_ih[1][row * 32 + k]));
          put16(HF_L2H + (g * 2 + b) * 2, l, tt, sc(g, m->w_hh[1][row * 32 + k]));
        }
      }
  for (int b = 0; b < 2; ++b)
    for (int l = 0; l < 64; ++l)
      for (int tt = 0; tt < 8; ++tt)
        put16(HF_HEAD + b * 2, l, tt, (l & 31) < 5 ? m->w_out[(size_t)(l & 31) * 32 + gru_tile_row(8 * b + tt, l >> 5)] : 0.0f);
  if (!in_range) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_set_gru: weights beyond the f16 range");
  {
    float* bv = reinterpret_cast<float*>(img16.data() + G16_BIAS_OFF);
    for (int i = 0; i < 8 * 32; ++i) {                                        // the same 8 gate-bias tiles, scaled alike
      const int kind = (i / 32) % 4;                                          // r, z, n_x, n_h
      bv[i] = (float)(gate_scale[kind < 2 ? kind : 2] * (double)img[GV_BIAS + i]);
    }
    for (int hf = 0; hf < 2; ++hf)
      for (int v = 0; v < 16; ++v) {
        const int r = gru_tile_row(v, hf);
        bv[8 * 32 + hf * 16 + v] = r < 5 ? m->b_out[r] : 0.0f;
      }
  }
  CPMPPI_ON_DEVICE(h);
  if (!h->gru_image) CPMPPI_HIP(h, hipMalloc(&h->gru_image, img.size() * sizeof(float)));
  CPMPPI_HIP(h, hipMemcpy(h->gru_image, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
  if (!h->gru16_image) CPMPPI_HIP(h, hipMalloc(&h->gru16_image, img16.size()));
  CPMPPI_HIP(h, hipMemcpy(h->gru16_image, img16.data(), img16.size(), hipMemcpyHostToDevice));
  return CPMPPI_OK;
} catch (const std::exception&) { return CPMPPI_ERR_NOMEM; }   // (no C++ exception leaves the C ABI)

int cpmppi_gru_predict(cpmppi_handle* h, uint32_t B, uint32_t horizon, const float* s0, const float* Q, const float* h0,
                       float* traj_out, float* h_out, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (!h->gru_image) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_gru_predict: no model set (cpmppi_set_gru)");
  if (horizon == 0) horizon = h->cfg.H;
  if (B == 0 || !s0 || !Q || !traj_out) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_gru_predict: bad argument");
  CPMPPI_ON_DEVICE(h);
  hipLaunchKernelGGL(gru_predict_kernel, dim3((B + GRU_ROLLOUTS_PER_BLOCK - 1) / GRU_ROLLOUTS_PER_BLOCK), dim3(BLOCK),
                     (size_t)GRU_IMAGE_FLOATS * sizeof(float), (hipStream_t)stream, h->gru_norm,
                     (const float*)h->gru_image, B, horizon, s0, Q, h0, traj_out, h_out);
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

int cpmppi_rollout_cost(cpmppi_handle* h, uint32_t E, const float* s0, const float* inputs, const float* target_position,
                        const float* target_equilibrium, const float* L, float* S_out, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || E > h->cfg.E || !s0 || !inputs || !target_position || !target_equilibrium || !S_out)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_rollout_cost: bad argument");
  if (h->prm.cost_id == CPMPPI_COST_LEGACY)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_rollout_cost: plugin costs only");
  CPMPPI_ON_DEVICE(h);
  Params prm = h->prm;
  prm.shift_mode = CPMPPI_SHIFT_NONE;
  prm.cc_weight = 0.0f;
  StepPtrs p{};
  p.s0 = s0; p.u_nom = h->zeros_H; p.u_prev = nullptr; p.x_t = target_position; p.te = target_equilibrium; p.L = L;
  p.noise = inputs; p.prev_in = nullptr; p.seed = 0; p.offset = 0; p.offset_dev = nullptr; p.env_offset = 0; p.stash = 0;
  uint32_t rpl = h->cfg.rollouts_per_lane;
  if (h->cfg.math_mode != CPMPPI_MATH_FAST) rpl = 1;
  else if (rpl == 0) rpl = ((uint64_t)E * h->cfg.N >= PACKED_MIN_ROLLOUTS) ? 2 : 1;
  p.nb = (h->cfg.N + BLOCK * rpl - 1) / (BLOCK * rpl);
  p.W = h->cfg.H;
  p.S_out = S_out; p.partial = h->workspace; p.counter = nullptr; p.u_nom_out = nullptr; p.Q_out = nullptr;
  hipError_t e = launch_rollout(h, prm, rpl, CPMPPI_NOISE_DELTA_U, dim3(E * p.nb), (size_t)WAVES * p.W * sizeof(float),
                                (hipStream_t)stream, p);
  CPMPPI_HIP(h, e);
  return CPMPPI_OK;
}

#ifdef CPMPPI_GRU_STAMPS
extern "C" int cpmppi_debug_gru_stamps(unsigned long long out[8], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gru_stamp_sum), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_gru_stamp_sum), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

int cpmppi_rollout_cost_grad(cpmppi_handle* h, uint32_t E, const float* s0, const float* inputs,
                             const float* target_position, const float* target_equilibrium, const float* L,
                             const float* previous_input, float* S_out, float* grad_out, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || E > h->cfg.E || !s0 || !inputs || !target_position || !target_equilibrium || !grad_out)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_rollout_cost_grad: bad argument");
  if (h->prm.cost_id == CPMPPI_COST_LEGACY)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_rollout_cost_grad: plugin costs only");
  if (h->prm.qb_mode != 0u)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_rollout_cost_grad: no adjoint for quadratic_boundary / quadratic_boundary_nonconvex "
                                       "(built: quadratic_boundary_grad_minimal, default, quadratic_boundary_grad)");
  if (h->cfg.math_mode != CPMPPI_MATH_FAST)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_rollout_cost_grad: the adjoint is written for the FAST arithmetic");
  const size_t lds = (size_t)h->cfg.S * 6 * BLOCK * sizeof(float);
  if (lds > 150 * 1024) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_rollout_cost_grad: S too large for the LDS sub-state buffer (<= 25)");
  CPMPPI_ON_DEVICE(h);
  const size_t B = (size_t)E * h->cfg.N;
  const size_t need = (size_t)h->cfg.H * 6 * (size_t)h->cfg.E * h->cfg.N;
  if (h->grad_ckpt_floats < need) {
    if (h->grad_ckpt) (void)hipFree(h->grad_ckpt);
    h->grad_ckpt = nullptr; h->grad_ckpt_floats = 0;
    CPMPPI_HIP(h, hipMalloc(&h->grad_ckpt, need * sizeof(float)));
    h->grad_ckpt_floats = need;
  }
  GradPtrs a{s0, inputs, target_position, target_equilibrium, L, previous_input, h->grad_ckpt, S_out, grad_out, E};
  const dim3 grid((unsigned)((B + BLOCK - 1) / BLOCK));
  hipStream_t st = (hipStream_t)stream;
  if (h->cfg.ode_predictor == CPMPPI_ODE_CROMER) {
    switch (h->prm.cost_id) {
      case CPMPPI_COST_QBGM: hipLaunchKernelGGL((rollout_grad_kernel<COST_QBGM, PREDICTOR_ODE>), grid, dim3(BLOCK), lds, st, h->prm, a); break;
      case CPMPPI_COST_DEFAULT: hipLaunchKernelGGL((rollout_grad_kernel<COST_DEFAULT, PREDICTOR_ODE>), grid, dim3(BLOCK), lds, st, h->prm, a); break;
      default: hipLaunchKernelGGL((rollout_grad_kernel<COST_QBG, PREDICTOR_ODE>), grid, dim3(BLOCK), lds, st, h->prm, a); break;
    }
  } else {
    switch (h->prm.cost_id) {
      case CPMPPI_COST_QBGM: hipLaunchKernelGGL(rollout_grad_kernel<COST_QBGM>, grid, dim3(BLOCK), lds, st, h->prm, a); break;
      case CPMPPI_COST_DEFAULT: hipLaunchKernelGGL(rollout_grad_kernel<COST_DEFAULT>, grid, dim3(BLOCK), lds, st, h->prm, a); break;
      default: hipLaunchKernelGGL(rollout_grad_kernel<COST_QBG>, grid, dim3(BLOCK), lds, st, h->prm, a); break;
    }
  }
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

int cpmppi_adam_step(cpmppi_handle* h, uint32_t E, float* Q, const float* grad, float* m, float* v, uint32_t iteration,
                     float learning_rate, float beta1, float beta2, float epsilon, float gradmax_clip, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || E > h->cfg.E || !Q || !grad || !m || !v || iteration == 0)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_adam_step: bad argument (iteration counts from 1)");
  CPMPPI_ON_DEVICE(h);
  const size_t rows = (size_t)E * h->cfg.N;
  // Keras Adam: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t), epsilon outside the square root
  const double lr_t = (double)learning_rate * sqrt(1.0 - pow((double)beta2, (double)iteration)) /
                      (1.0 - pow((double)beta1, (double)iteration));
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)((rows + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     rows, h->cfg.H, Q, grad, m, v, (float)lr_t, beta1, beta2, epsilon, gradmax_clip, h->prm.lo, h->prm.hi);
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

int cpmppi_sgd_step(cpmppi_handle* h, uint32_t E, float* Q, const float* grad, float learning_rate, float gradmax_clip,
                    void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || E > h->cfg.E || !Q || !grad) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_sgd_step: bad argument");
  CPMPPI_ON_DEVICE(h);
  const size_t rows = (size_t)E * h->cfg.N;
  hipLaunchKernelGGL(sgd_step_kernel, dim3((unsigned)((rows + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     rows, h->cfg.H, Q, grad, learning_rate, gradmax_clip, h->prm.lo, h->prm.hi);
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

int cpmppi_cem_sample(cpmppi_handle* h, uint32_t E, const float* mean, const float* stdev, uint64_t seed, uint64_t offset,
                      uint32_t env_offset, float* Q_out, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || E > h->cfg.E || !mean || !stdev || !Q_out) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_cem_sample: bad argument");
  CPMPPI_ON_DEVICE(h);
  const size_t rows = (size_t)E * h->cfg.N;
  hipLaunchKernelGGL(cem_sample_kernel, dim3((unsigned)((rows + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     h->prm, E, mean, stdev, seed, offset, env_offset, Q_out);
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

int cpmppi_cem_gmm_sample(cpmppi_handle* h, uint32_t E, const float* centres, uint32_t K, const float* stdev, uint64_t seed,
                          uint64_t offset, uint32_t env_offset, float* Q_out, uint32_t* component_out, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || E > h->cfg.E || !centres || K == 0 || !stdev || !Q_out)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_cem_gmm_sample: bad argument");
  CPMPPI_ON_DEVICE(h);
  const size_t rows = (size_t)E * h->cfg.N;
  hipLaunchKernelGGL(cem_gmm_sample_kernel, dim3((unsigned)((rows + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     h->prm, E, centres, K, stdev, seed, offset, env_offset, Q_out, component_out);
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

int cpmppi_cem_update(cpmppi_handle* h, uint32_t E, const float* S, const float* Q, uint32_t best_k, float stdev_min,
                      float* mean_out, float* stdev_out, uint32_t* elite_idx_out, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || !S || !Q || !mean_out || !stdev_out || best_k == 0 || best_k > h->cfg.N)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_cem_update: bad argument (0 < best_k <= N)");
  uint32_t Np = 1;
  while (Np < h->cfg.N) Np <<= 1;
  if ((size_t)Np * 8 > 160 * 1024 - 1024) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_cem_update: N too large for the LDS sort (<= 16384)");
  CPMPPI_ON_DEVICE(h);
  hipLaunchKernelGGL(cem_update_kernel, dim3(E), dim3(BLOCK), (size_t)Np * 8, (hipStream_t)stream, h->prm, S, Q, best_k,
                     stdev_min, Np, mean_out, stdev_out, elite_idx_out);
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

int cpmppi_reward_weighted_average(cpmppi_handle* h, uint32_t E, const float* S, const float* delta_u, float* out,
                                   void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || !S || !delta_u || !out) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_reward_weighted_average: bad argument");
  CPMPPI_ON_DEVICE(h);
  hipLaunchKernelGGL(rwa_kernel, dim3(E), dim3(BLOCK), 0, (hipStream_t)stream, h->prm, S, delta_u, out);
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

int cpmppi_plant_step(cpmppi_handle* h, const cpmppi_plant_args* a, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (!a || a->E == 0 || !a->s || !a->Q || !(a->dt_sim > 0.0f)) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_step: bad argument");
  const uint32_t period_steps = a->period_steps ? a->period_steps : a->n_substeps;
  if (a->n_substeps > period_steps)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_step: n_substeps must not exceed period_steps");
  const uint32_t save_every = a->save_every ? a->save_every : period_steps;
  if ((a->states_log || a->dd_log) && save_every == 0)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_step: save_every / period_steps missing");
  const bool tables = a->target_position_table || a->target_equilibrium_table || a->L_table || a->m_pole_table || a->L_controller_table;
  if (a->Q_disturbance_table && a->ctrl_rows == 0)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_step: Q_disturbance_table needs ctrl_rows > 0");
  // (the kernel writes the ring whenever it is given, measurement chain or not: advisor, round 5)
  if (a->state_history && a->history_len == 0u)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_step: state_history needs history_len > 0");
  if (misaligned(a->state_history)) return fail(h, CPMPPI_ERR_ALIGN, "cpmppi_plant_step: misaligned");
  if (a->s_measured) {
    const bool delayed = a->latency_steps != 0u || a->latency_frac != 0.0;
    if (delayed && (!a->state_history || a->history_len < a->latency_steps + 2u))
      return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_step: a latency needs state_history with history_len >= latency_steps + 2");
    if (!(a->latency_frac >= 0.0 && a->latency_frac < 1.0))
      return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_step: latency_frac must lie in [0, 1)");
    if ((a->angle_offset_table || a->informed_table) && a->sched_rows == 0)
      return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_step: schedule tables need sched_rows > 0");
    if (a->measurement_noise_table && a->ctrl_rows == 0)
      return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_step: measurement_noise_table needs ctrl_rows > 0");
    if (misaligned(a->s_measured) || misaligned(a->state_history) || misaligned(a->measurement_noise_table) ||
        ((uintptr_t)a->angle_offset_table & 7u))
      return fail(h, CPMPPI_ERR_ALIGN, "cpmppi_plant_step: misaligned");
  }
  if (a->L_controller_table && !a->L_table)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_step: L_controller_table stands in for L_table in L_out: give both");
  if (tables && a->sched_rows == 0) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_step: schedule tables need sched_rows > 0");
  if (a->row_envs != 0 && a->row_envs < a->E) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_step: row_envs < E");
  // a host-named period must lie inside the control log it is to be written to (rows of the state logs that fall outside are
  // skipped by the kernel, as for a device counter)
  if (!a->period_dev && a->Q_log && a->period >= a->ctrl_rows)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_step: period outside Q_log (period >= ctrl_rows)");
  if (misaligned(a->s) || misaligned(a->Q) || misaligned(a->L) || misaligned(a->states_log) || misaligned(a->dd_log) ||
      misaligned(a->Q_log) || misaligned(a->target_position_table) || misaligned(a->target_equilibrium_table) ||
      misaligned(a->L_table) || misaligned(a->target_position_out) || misaligned(a->target_equilibrium_out) || misaligned(a->L_out) ||
      misaligned(a->m_pole) || misaligned(a->m_pole_table) || misaligned(a->L_controller_table) || misaligned(a->Q_disturbance_table) ||
      misaligned(a->Q_applied_out) ||
      (a->period_dev && ((uintptr_t)a->period_dev & 7u)))
    return fail(h, CPMPPI_ERR_ALIGN, "cpmppi_plant_step: misaligned");
  CPMPPI_ON_DEVICE(h);
  Params plant = h->prm;                  // the simulated system's own pole mass (see cpmppi_set_pole_mass)
  plant.m_pole = h->plant_m_pole;
  PlantDev d{};
  d.E = a->E; d.row_envs = a->row_envs ? a->row_envs : a->E; d.n_sub = a->n_substeps; d.period_steps = period_steps; d.save_every = save_every ? save_every : 1u;
  d.sched_stride = a->sched_stride ? a->sched_stride : 1u;
  d.dt_sim = a->dt_sim;
  d.period = a->period; d.save_rows = a->save_rows; d.ctrl_rows = a->ctrl_rows; d.sched_rows = a->sched_rows ? a->sched_rows : 1u;
  d.period_dev = (const unsigned long long*)a->period_dev;
  d.s = a->s; d.Q = a->Q; d.L = a->L;
  d.states_log = a->states_log; d.dd_log = a->dd_log; d.Q_log = a->Q_log;
  d.tp_table = a->target_position_table; d.te_table = a->target_equilibrium_table; d.L_table = a->L_table;
  d.tp_out = a->target_position_out; d.te_out = a->target_equilibrium_out; d.L_out = a->L_out;
  d.m_pole = a->m_pole; d.m_table = a->m_pole_table; d.Lc_table = a->L_controller_table;
  d.Qd_table = a->Q_disturbance_table; d.Q_bias = a->Q_bias; d.Qa_out = a->Q_applied_out;
  d.s_meas = a->s_measured; d.hist = a->state_history; d.hist_len = a->history_len; d.lat_steps = a->latency_steps;
  d.lat_frac = a->latency_frac; d.noise_table = a->measurement_noise_table; d.off_table = a->angle_offset_table;
  d.informed_table = a->informed_table;
  hipLaunchKernelGGL(plant_kernel, dim3((a->E + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream, plant, d);
  CPMPPI_HIP(h, hipGetLastError());
  return CPMPPI_OK;
}

int cpmppi_plant_advance(cpmppi_handle* h, uint32_t E, float* s, const float* Q, const float* L, uint32_t n_substeps,
                         float dt_sim, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || !s || !Q || !(dt_sim > 0.0f)) return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_advance: bad argument");
  cpmppi_plant_args a{};
  a.E = E; a.s = s; a.Q = Q; a.L = L; a.n_substeps = n_substeps; a.period_steps = n_substeps; a.dt_sim = dt_sim;
  return cpmppi_plant_step(h, &a, stream);
}

// (ABI 2's form of the recording plant: states_log[row + 1] = the advanced state, Q_log[row] = Q - cpmppi_plant_step with one saved
// row per control period)
int cpmppi_plant_advance_record(cpmppi_handle* h, uint32_t E, float* s, const float* Q, const float* L, uint32_t n_substeps,
                                float dt_sim, float* states_log, float* Q_log, uint64_t log_rows, uint64_t row,
                                const void* row_dev, void* stream) {
  if (!h) return CPMPPI_ERR_BAD_ARG;
  if (E == 0 || !s || !Q || !(dt_sim > 0.0f) || n_substeps == 0)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_advance_record: bad argument");
  if ((states_log || Q_log) && !row_dev && row >= log_rows)
    return fail(h, CPMPPI_ERR_BAD_ARG, "cpmppi_plant_advance_record: row outside the logs (row >= log_rows)");
  cpmppi_plant_args a{};
  a.E = E; a.s = s; a.Q = Q; a.L = L; a.n_substeps = n_substeps; a.period_steps = n_substeps; a.dt_sim = dt_sim;
  a.period = row; a.period_dev = row_dev;
  a.states_log = states_log; a.save_rows = log_rows + 1u; a.save_every = n_substeps;
  a.Q_log = Q_log; a.ctrl_rows = log_rows;
  return cpmppi_plant_step(h, &a, stream);
}

uint32_t cpmppi_abi_version(void) { return CPMPPI_ABI_VERSION; }

int cpmppi_stream_create(int device, void** stream_out) try {
  if (!stream_out) return fail(nullptr, CPMPPI_ERR_BAD_ARG, "cpmppi_stream_create: null argument");
  *stream_out = nullptr;
  DeviceGuard guard(device);
  if (guard.err != hipSuccess) return fail(nullptr, CPMPPI_ERR_HIP, std::string("cpmppi_stream_create: hipSetDevice: ") + hipGetErrorString(guard.err));
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return fail(nullptr, CPMPPI_ERR_HIP, std::string("cpmppi_stream_create: ") + hipGetErrorString(e));
  // one bit per CU, all set: the mask only serves to make the runtime give this stream a queue of its own
  const uint32_t words = ((uint32_t)prop.multiProcessorCount + 31u) / 32u;
  std::vector<uint32_t> mask(words, 0xFFFFFFFFu);
  if (prop.multiProcessorCount % 32) mask[words - 1] = (1u << (prop.multiProcessorCount % 32)) - 1u;
  hipStream_t st = nullptr;
  e = hipExtStreamCreateWithCUMask(&st, words, mask.data());
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);          // (a runtime without CU masks: an ordinary stream)
  }
  if (e != hipSuccess) return fail(nullptr, CPMPPI_ERR_HIP, std::string("cpmppi_stream_create: ") + hipGetErrorString(e));
  *stream_out = st;
  return CPMPPI_OK;
} catch (const std::exception&) { return CPMPPI_ERR_NOMEM; }   // (no C++ exception leaves the C ABI)

int cpmppi_stream_destroy(void* stream) {
  if (!stream) return CPMPPI_OK;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  return e == hipSuccess ? CPMPPI_OK : fail(nullptr, CPMPPI_ERR_HIP, std::string("cpmppi_stream_destroy: ") + hipGetErrorString(e));
}

}  // extern "C"

int cpmppi_internal_step_ticket(cpmppi_handle* h, const cpmppi_step_args* a, void* stream, const cpmppi_comm::GatherTicket* ticket) {
  return step_impl(h, a, stream, nullptr, ticket);
}
