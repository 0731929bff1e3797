/*
 * cpmppi.h — C ABI of libcpmppi.so: the MI355X-native (gfx950) MPPI rollout hot path for CartPoleSimulation.
 *
 * One fused HIP path replaces three nested Python plugin seams of the reference (paths relative to the
 * reference checkout):
 *
 *   predictor seam   PredictorWrapper.predict_core(s[N,6], Q[N,H,1]) -> [N,H+1,6]
 *                    call sites: Control_Toolkit_ASF/Controllers/controller_mppi_cartpole.py:191,
 *                    SI_Toolkit_ASF/ToolkitCustomization/Modules/ODE_module.py:46-50; per-step hook
 *                    SI_Toolkit_ASF/ToolkitCustomization/predictors_customization_v0.py:41-55
 *                    -> cpmppi_predict()
 *   cost seam        cost_function.get_stage_cost / get_terminal_cost / get_trajectory_cost
 *                    Control_Toolkit_ASF/Cost_Functions/CartPole/quadratic_boundary_grad_minimal.py:95-126,
 *                    .../default.py:41-88, Cost_Functions/GymlikeCartPole/cost_function_gym.py:12-21
 *                    -> cpmppi_trajectory_cost()
 *   optimizer seam   optimizer_mppi.step(s, time) (Control_Toolkit submodule, absent from the reference mount;
 *                    in-tree statement of the same algorithm: controller_mppi_cartpole.py:454-569 with
 *                    trajectory_rollouts :164-224, q :227-275, phi :278-303, reward_weighted_average :306-321,
 *                    initialize_perturbations :392-452)
 *                    -> cpmppi_sample() + cpmppi_step()
 *
 * Conventions
 *   - every array argument is a caller-owned DEVICE pointer to contiguous float32 (e.g. torch.Tensor.data_ptr() of a
 *     ROCm tensor); the library owns only the handle and its small reduction workspace;
 *   - all work is enqueued on the hipStream_t passed as `stream` (NULL = default stream); no hidden synchronisation,
 *     no allocation inside the launch functions (they are hipGraph-capturable);
 *   - return value 0 = success, negative = cpmppi_status; the message of the last failure of a handle is
 *     available from cpmppi_last_error();
 *   - a handle is not thread-safe: one handle per (device, stream);
 *   - there is NO CPU fallback: without a gfx950 device cpmppi_create() fails with CPMPPI_ERR_NO_DEVICE.
 *
 * State layout (CartPole/state_utilities.py:5-23): [angle, angleD, angle_cos, angle_sin, position, positionD].
 */
#ifndef CPMPPI_H
#define CPMPPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPMPPI_ABI_VERSION 5u   /* 2: cpmppi_step_args.u_nom_out, cpmppi_plant_advance_record(log_rows), cpmppi_comm_*;
                                   3: cpmppi_config.ode_predictor (+ cost ids 4 / 5, cpmppi_last_launch, cpmppi_comm_set_timeout);
                                   4: cpmppi_plant_step / cpmppi_plant_args (the experiment schedule), cpmppi_write_recordings takes
                                      per-row columns (cpmppi_recording), cpmppi_launch_info.cost_plugin, CPMPPI_ERR_IO,
                                      cpmppi_comm_info, cpmppi_groups_*.  cpmppi_abi_version() reports what a loaded library was built as;
                                   5: cpmppi_comm_set_stamped (+ cpmppi_comm_info.stamped), cpmppi_groups_comm_init / cpmppi_groups_run_gather;
                                      a caller-given rccl_path now wins over an RCCL the process has already loaded. */
#define CPMPPI_STATE_DIM 6u
#define CPMPPI_MAX_HORIZON 1024u

typedef enum {
  CPMPPI_OK = 0,
  CPMPPI_ERR_BAD_ARG = -1,      /* null pointer, bad shape, unknown enum value */
  CPMPPI_ERR_ABI = -2,          /* abi_version mismatch */
  CPMPPI_ERR_NO_DEVICE = -3,    /* no HIP device / not a gfx950 part */
  CPMPPI_ERR_HIP = -4,          /* a HIP runtime call failed (text in cpmppi_last_error) */
  CPMPPI_ERR_ALIGN = -5,        /* pointer not 4-byte aligned */
  CPMPPI_ERR_COMM = -6,         /* RCCL missing or an RCCL call failed (text in cpmppi_last_error) */
  CPMPPI_ERR_IO = -7,           /* a file could not be created or written (cpmppi_write_recordings; errno text in cpmppi_last_error) */
  CPMPPI_ERR_NOMEM = -8         /* host memory exhausted (no C++ exception ever leaves the library) */
} cpmppi_status;

/* cost_id: which in-tree cost formulation the rollout kernel evaluates. */
enum {
  CPMPPI_COST_QBGM = 0,    /* quadratic_boundary_grad_minimal.py:64-126 ; cost_w = {dd_quadratic_weight, db_weight,
                              ep_weight, ekp_weight, cc_weight, R, permissible_track_fraction} */
  CPMPPI_COST_DEFAULT = 1, /* default.py:23-88 ; cost_w = {dd_weight, ep_weight, cc_weight, R} */
  CPMPPI_COST_LEGACY = 2,  /* controller_mppi_cartpole.py:119-161,227-303 (q + phi); cost_w = {dd_weight, ep_weight,
                              ekp_weight, ekc_weight, cc_weight, ccrc_weight}; the MPPI correction term is part of q */
  CPMPPI_COST_QBG = 3,     /* quadratic_boundary_grad.py:64-232; cost_w = {up: dd_quadratic, dd_linear, db, ep, ekp, cc,
                              ccrc | down: the same seven | target_angular_speed_sqr_max_correction up, down |
                              permissible_track_fraction | cos(admissible_angle) | R}; the set is chosen per env by
                              target_equilibrium == 1 */
  CPMPPI_COST_QB = 4,      /* quadratic_boundary.py:26-87 ; cost_w = {dd_weight, ep_weight, cc_weight, R, ccrc_weight}: default.py
                              with a quadratic track-edge term beyond 0.95 THL and a control-change-rate term against
                              previous_input (added only when a previous input is given, :83-85); default.py's terminal
                              cost.  Runs on default.py's kernels (one more wave-uniform switch); no adjoint, no GRU */
  CPMPPI_COST_QB_NONCONVEX = 5 /* quadratic_boundary_nonconvex.py:27-105: the same plus the cosine ripple on the position term.
                              The reference cannot import this module as shipped (it reads `cem_ccrc_weight`, absent from
                              config_cost_function.yml:47-52); pinned to the outputs of its own class with that one key supplied
                              (:= the section's ccrc_weight; tests/golden/qb_costs.npz "nc/...") */
};

enum { CPMPPI_REDUCE_SUM = 0, CPMPPI_REDUCE_MEAN = 1 };            /* horizon aggregation of the plugin costs */
enum { CPMPPI_CONTROL_CLIP = 0, CPMPPI_CONTROL_PENALISE = 1 };     /* clip u_run & u_nom | legacy 1e5 penalty */
enum { CPMPPI_SHIFT_REPEAT_LAST = 0, CPMPPI_SHIFT_APPEND_ZERO = 1, CPMPPI_SHIFT_NONE = 2 };
enum { CPMPPI_CORRECTION_U_RUN = 0, CPMPPI_CORRECTION_U_NOM = 1 }; /* which u enters the MPPI correction term */
enum { CPMPPI_MATH_PRECISE = 0,  /* IEEE divide, libm-grade sincos, no FMA contraction: closest to numpy float32 */
       CPMPPI_MATH_FAST = 1 };   /* same float32 formulas with folded constants and FMA, v_rcp_f32 divide (<=1.5 ulp),
                                    polynomial sincos on the wrapped angle; inside a control step (cos,sin) advance
                                    by rotation through w*t and are re-synchronised by the exact wrap + sincos at the
                                    step's last substep, so states at control-step granularity carry rounding noise
                                    only (tools/deviation.py: median 1e-6, p99 1e-5 of the reference's own mode A) */
enum { CPMPPI_ODE_V0 = 0,        /* predictor_ODE_v0 (predictors_customization_v0.py:22-57 -> cartpole_numba.py:55-78):
                                    simultaneous forward Euler, elastic edge bounce, fmod angle wrap - the default */
       CPMPPI_ODE_CROMER = 1 };  /* predictor_ODE, predictor_specification "ODE" - what the shipped config_controllers.yml:3,14
                                    names (predictors_customization.py:25-69 -> cartpole_equations.py:181-259,293-308):
                                    Euler-Cromer, NO edge bounce, angle = atan2(sin, cos).  Serves every entry point
                                    that integrates the ODE: cpmppi_step*, cpmppi_predict, cpmppi_rollout_cost (the CEM
                                    family) and cpmppi_rollout_cost_grad (the adjoint of this substep) */
enum { CPMPPI_NOISE_DELTA_U = 0, /* noise = delta_u[E,N,H]  (reference layout, rollout-major)              */
       CPMPPI_NOISE_KNOTS = 1,   /* noise = knots[E,N,P], P = ceil(H/period)+1; interpolated in-kernel       */
       CPMPPI_NOISE_PHILOX = 2,  /* knots generated in-kernel from (seed, offset): no perturbation buffer   */
       CPMPPI_NOISE_DELTA_U_TILED = 3 }; /* noise = delta_u in the library's tiled layout (cpmppi_sample_tiled /
                                    cpmppi_tile_delta_u): [E][ceil(N/64)][ceil(H/4)][64][4], 16-byte aligned          */

typedef struct {
  uint32_t abi_version;          /* CPMPPI_ABI_VERSION */
  uint32_t E;                    /* capacity: independent MPPI problem instances (envs) per call */
  uint32_t N;                    /* rollouts (samples) per env          config_optimizers.yml:91 num_rollouts */
  uint32_t H;                    /* horizon in control steps            config_optimizers.yml:89 mpc_horizon  */
  uint32_t S;                    /* Euler substeps per control step     config_predictors.yml:21 intermediate_steps */
  float dt;                      /* control period; t_step = dt / S     config_optimizers.yml:90 mpc_timestep */
  /* physics — cartpole_physical_parameters.yml:6-17,34,42 rounded to float32 (cartpole_parameters.py:27-31) */
  float k, m_cart, m_pole, g, J_fric, M_fric, u_max, track_half_length;
  float L_default;               /* used where the per-env L pointer is NULL */
  /* cost */
  uint32_t cost_id;
  float cost_w[24];
  /* MPPI — config_optimizers.yml:92-97 */
  float R, LBD, NU, cc_weight;
  float sigma;                   /* knot std-dev = SQRTRHOINV / sqrt(dt) */
  uint32_t period;               /* period_interpolation_inducing_points */
  float action_low, action_high; /* control_limits */
  uint32_t horizon_reduce;       /* CPMPPI_REDUCE_*      (unpinned upstream choice) */
  uint32_t control_mode;         /* CPMPPI_CONTROL_*     (u3) */
  uint32_t shift_mode;           /* CPMPPI_SHIFT_*       (u2) */
  uint32_t correction_u;         /* CPMPPI_CORRECTION_*  */
  uint32_t math_mode;            /* CPMPPI_MATH_* */
  uint32_t rollouts_per_lane;    /* lane mapping of the FAST rollout kernel: 0 = automatic (2 for launches of
                                    >= 131072 rollouts = one packed wave on every SIMD, else 1; measured crossover),
                                    1 = one rollout per lane (lowest latency), 2 = two rollouts per lane as packed
                                    float2 (highest throughput) */
  uint32_t ode_predictor;        /* CPMPPI_ODE_*: which in-tree ODE predictor integrates the rollouts (predictor_type of
                                    SI_Toolkit_ASF/config_predictors.yml:18-26) */
} cpmppi_config;

typedef struct cpmppi_handle cpmppi_handle;

enum { CPMPPI_PREDICTOR_ODE_V0 = 0,   /* the handle's ODE predictor (cpmppi_config.ode_predictor; the name is ABI 1's) */
       CPMPPI_PREDICTOR_GRU = 1 };

/* Neural predictor of BASELINE configs[4] (model naming SI_Toolkit_ASF/config_predictors.yml:8-13): two GRU layers of
 * 32 units and a dense head, torch.nn.GRU convention (gate rows r, z, n; b_ih and b_hh).  HOST pointers; inputs are
 * ordered (Q, angleD, angle_cos, angle_sin, position, positionD), outputs (angleD, angle_cos, angle_sin, position,
 * positionD) — SI_Toolkit's alphabetical feature order; normalised = x*scale + shift (NULL = identity). */
typedef struct {
  uint32_t inputs, hidden, layers, outputs;   /* 6, 32, 2, 5 */
  const float* w_ih[2];                       /* [96,6], [96,32] */
  const float* w_hh[2];                       /* [96,32] */
  const float* b_ih[2];                       /* [96] */
  const float* b_hh[2];                       /* [96] */
  const float* w_out;                         /* [5,32] */
  const float* b_out;                         /* [5] */
  const float* in_scale;  const float* in_shift;    /* [6] */
  const float* out_scale; const float* out_shift;   /* [5] */
} cpmppi_gru_model;

/* One optimizer step for `E` envs (E <= config.E).  All pointers are device pointers. */
typedef struct {
  uint32_t E;                       /* active envs in this call */
  const float* s0;                  /* [E,6]  current state of each env */
  float* u_nom;                     /* [E,H]  in: nominal sequence as left by the previous step (the kernel applies
                                              config.shift_mode itself); out: updated nominal sequence */
  const float* u_prev;              /* [E,H]  legacy control-change-rate term only; NULL = use u_nom as found on entry
                                              (controller_mppi_cartpole.py:558) */
  const float* target_position;     /* [E] */
  const float* target_equilibrium;  /* [E] */
  const float* L;                   /* [E] pole length per env, or NULL = config.L_default */
  uint32_t noise_kind;              /* CPMPPI_NOISE_* */
  const float* noise;               /* delta_u[E,N,H], knots[E,N,P] or the tiled delta_u; ignored for CPMPPI_NOISE_PHILOX */
  uint64_t seed;                    /* CPMPPI_NOISE_PHILOX: key */
  uint64_t offset;                  /* CPMPPI_NOISE_PHILOX: step counter (fresh noise per step) */
  uint32_t env_offset;              /* CPMPPI_NOISE_PHILOX: global index of env 0 (rank * E_local when sharded) */
  float* Q_out;                     /* [E]    first element of the updated nominal sequence */
  float* S_out;                     /* [E,N]  per-rollout total cost, or NULL */
  uint32_t predictor;               /* CPMPPI_PREDICTOR_ODE_V0 (default, 0) or CPMPPI_PREDICTOR_GRU */
  const float* h0;                  /* GRU only: hidden state per env [E,2,32] shared by the env's rollouts, or NULL=0 */
  const float* previous_input;      /* [E] control applied before this step (Q_ccrc of CartPole/__init__.py:517): the
                                       control-change-rate term of quadratic_boundary_grad at stage 0; NULL = 0 */
  uint64_t* offset_dev;             /* CPMPPI_NOISE_PHILOX, optional: the step counter in DEVICE memory — read by the kernel
                                       instead of `offset` and incremented by one after the step, stream-ordered.  With it a
                                       captured HIP graph of (step, plant, ...) can be replayed: no launch argument changes
                                       between control steps.  NULL = use `offset`. */
  float* u_nom_out;                 /* [E,H] optional (ODE predictor): where the updated nominal sequence is written; `u_nom`
                                       is then only read.  Two buffers used alternately let a consumer of step i's result
                                       (the all-gather of cpmppi_comm_gather) overlap step i+1 without a snapshot copy.
                                       NULL = in place (u_nom). */
} cpmppi_step_args;

int cpmppi_create(const cpmppi_config* cfg, int device, cpmppi_handle** out);
void cpmppi_destroy(cpmppi_handle* h);
const char* cpmppi_last_error(const cpmppi_handle* h);  /* h may be NULL: error of the last failed cpmppi_create */
int cpmppi_get_config(const cpmppi_handle* h, cpmppi_config* out);

/* Which instantiation of the rollout kernel the handle's most recent cpmppi_step / cpmppi_step_host / cpmppi_step_gather /
 * cpmppi_rollout_cost launch used (all zero before the first one) - so that a caller that verifies or times a launch can
 * name the kernel it ran: rollout_cost_kernel<cost_id, math_mode == FAST, noise_kind, rollouts_per_lane, build_variant>.
 * build_variant: 0 = latency build (one rollout per lane, at most one wave per SIMD), 1 = throughput build, 2 = mid-size
 * build (phased horizon loop), 3 = its form for launches of at most one wave per SIMD. */
typedef struct {
  uint32_t cost_id, math_mode, noise_kind, rollouts_per_lane, build_variant, ode_predictor, blocks;
  uint32_t cost_plugin;   /* the PUBLIC cost id of the handle (CPMPPI_COST_*): quadratic_boundary (4) and _nonconvex (5) run on
                             default.py's kernels, so `cost_id` - the kernel's template argument - reads 1 for both */
} cpmppi_launch_info;
int cpmppi_last_launch(const cpmppi_handle* h, cpmppi_launch_info* out);

/* Mutable per-call knobs (GUI sliders / attribute updates in the reference mutate these between steps). */
int cpmppi_set_cost_weights(cpmppi_handle* h, uint32_t cost_id, const float* cost_w, uint32_t n);
/* The pole mass every later call of this handle computes with (config.m_pole until then): predictor_ODE takes it from
 * variable_parameters.m_pole at every step (predictors_customization.py:55-58; the simulator sends 'm_pole' with every
 * controller.step, CartPole/__init__.py:509-520).  Handle-wide (one value for all envs; the pole LENGTH is the per-env
 * attribute); launches already enqueued - and captured graphs - keep the value they were enqueued with.  This is the
 * CONTROLLER's belief (the simulator sends m_pole_for_controller): the plant of cpmppi_plant_advance* keeps config.m_pole. */
int cpmppi_set_pole_mass(cpmppi_handle* h, float m_pole);

/* a17 — device sampler: knots ~ sigma * N(0,1) from Philox4x32-10 keyed by (seed), counter (rollout, env, knot pair,
 * offset); writes knots[E,N,P] and/or the interpolated delta_u[E,N,H] (either pointer may be NULL).
 * Interpolation follows controller_mppi_cartpole.py:434-446. */
int cpmppi_sample(cpmppi_handle* h, uint32_t E, uint64_t seed, uint64_t offset, uint32_t env_offset,
                  float* knots_out, float* delta_u_out, void* stream);

/* Interpolate caller-provided knots[E,N,P] (e.g. drawn with numpy SFC64 for bit-identical parity runs) to
 * delta_u[E,N,H]. */
int cpmppi_interpolate(cpmppi_handle* h, uint32_t E, const float* knots, float* delta_u_out, void* stream);

/* The TILED perturbation layout — delta_u[E,N,H] (controller_mppi_cartpole.py:434-446 produces it rollout-major) stored as
 * [E][G = ceil(N/64)][Hq = ceil(H/4)][64 rows][4 steps]: element (env, n, k) at
 * ((((env*G + n/64)*Hq + k/4)*64 + n%64)*4 + k%4, padding zero.  The rollout kernel reads it with fully used, contiguous
 * 1 KB wave accesses (the rollout-major layout costs 5.9x the algorithmic traffic at H = 50).
 *   cpmppi_tiled_floats   number of floats of the tiled buffer for E envs (allocate 16-byte aligned)
 *   cpmppi_sample_tiled   a17 straight into it: Philox knots (knots_in NULL) or caller knots[E,N,P], interpolated
 *   cpmppi_tile_delta_u   re-tile a reference-layout delta_u[E,N,H] (one coalesced pass: worth it when the buffer is
 *                         reused over several steps — for a single step the extra pass costs more than the rollout-major
 *                         kernel's over-fetch: 0.7 ms vs 0.36 ms at 8192 envs x 1024 x 50) */
size_t cpmppi_tiled_floats(const cpmppi_handle* h, uint32_t E);
int cpmppi_sample_tiled(cpmppi_handle* h, uint32_t E, uint64_t seed, uint64_t offset, uint32_t env_offset,
                        const float* knots_in, float* tiled_out, void* stream);
int cpmppi_tile_delta_u(cpmppi_handle* h, uint32_t E, const float* delta_u, float* tiled_out, void* stream);

/* Predictor seam (a9-a11): B independent rollouts.  s0[B,6], Q[B,H] (dimensionless control in [-1,1]),
 * L[B] or NULL -> traj[B,H+1,6] with traj[:,0]=s0.  horizon may be < config.H (0 = config.H). */
int cpmppi_predict(cpmppi_handle* h, uint32_t B, uint32_t horizon, const float* s0, const float* Q, const float* L,
                   float* traj_out, void* stream);

/* Cost seam (a12-a14): trajectories traj[B,H+1,6], inputs[B,H] -> stage_out[B,H] (may be NULL), terminal_out[B]
 * (may be NULL), total_out[B] (may be NULL; sum or mean per config.horizon_reduce, plugin costs only).
 * target_position / target_equilibrium are host scalars here (the plugin reads them from variable_parameters).
 * Legacy cost additionally needs u_nom[H], u_prev[H] (device) and interprets `inputs` as delta_u;
 * quadratic_boundary_grad reads its previous_input from u_prev[0] (NULL = 0). */
int cpmppi_trajectory_cost(cpmppi_handle* h, uint32_t B, uint32_t horizon, const float* traj, const float* inputs,
                           float target_position, float target_equilibrium, const float* u_nom, const float* u_prev,
                           float* stage_out, float* terminal_out, float* total_out, void* stream);

/* Fused hot path: rollout (a3-a11) + cost (a12-a15) + importance-weighted update (a16) + shift/clip (a18). */
int cpmppi_step(cpmppi_handle* h, const cpmppi_step_args* args, void* stream);

/* The fused step for a caller whose state lives on the HOST - the simulator's own call, controller.step(s, time,
 * updated_attributes) -> Q (CartPole/__init__.py:509-520): s0[E,6], target_position[E], target_equilibrium[E], L[E] (or
 * NULL) and Q[E] are HOST pointers, u_nom[E,H] stays a device buffer (the plan persists between control steps); in-kernel
 * Philox noise (seed, offset, env_offset as in cpmppi_step_args).  Synchronous by nature.
 *   up to 64 envs (CPMPPI_HOST_ZERO_COPY_MAX; the simulator's call is one): NO copy and NO stream wait - the inputs are
 *     staged into the handle's pinned, device-mapped block, which the kernel reads directly; the env's finalizing block
 *     stores Q into the same block and bumps a system-scope ticket the calling thread spins on (the stream is polled
 *     every few thousand spins, so a failed launch returns CPMPPI_ERR_HIP instead of hanging);
 *   more envs: one asynchronous copy up, the launch, one copy down, a wait on the stream. */
int cpmppi_step_host(cpmppi_handle* h, uint32_t E, const float* s0, const float* target_position,
                     const float* target_equilibrium, const float* L, float* u_nom, uint64_t seed, uint64_t offset,
                     uint32_t env_offset, float* Q, void* stream);

/* Kernel timing with HIP events recorded on the launch stream (off by default; enable = 0 switches it off).
 * enable = 1: every cpmppi_step is bracketed by an event before the rollout kernel, one after it and - only when a
 *   separate finalize or counter kernel follows - a third after those; rollout_ms / finalize_ms are per step.
 * enable = n > 1: ONE bracket around every n consecutive steps (an event costs ~5 us on the stream - two per launch
 *   are 10 % of a 100 us launch and would show in any wall-clock figure taken at the same time); rollout_ms then holds,
 *   per completed group, the bracket divided by n: the average duration of a step's kernels back to back, inter-launch
 *   gaps included; finalize_ms is 0.
 * cpmppi_get_profile synchronises on the events, returns the entries recorded since the last call (at most max_steps
 * are written; *n_steps receives their number) and resets the recorder. */
int cpmppi_set_profiling(cpmppi_handle* h, int enable);
int cpmppi_get_profile(cpmppi_handle* h, float* rollout_ms, float* finalize_ms, uint32_t max_steps, uint32_t* n_steps);

/* Attach / replace the GRU model of a handle (synchronous upload; not for the launch path). */
int cpmppi_set_gru(cpmppi_handle* h, const cpmppi_gru_model* model);

/* Predictor seam with the neural predictor (predictor_autoregressive_neural; augmentation angle = atan2(sin, cos),
 * SI_Toolkit_ASF/ToolkitCustomization/predictors_customization.py:121-127): s0[B,6], Q[B,H], h0[2,B,32] or NULL ->
 * traj[B,H+1,6], h_out[2,B,32] or NULL (device pointers). */
int cpmppi_gru_predict(cpmppi_handle* h, uint32_t B, uint32_t horizon, const float* s0, const float* Q, const float* h0,
                       float* traj_out, float* h_out, void* stream);

/* Rollout + cost only (no update): per-rollout trajectory cost S[E,N] of GIVEN input sequences inputs[E,N,H] under the
 * handle's plugin cost — cost_function.get_trajectory_cost(predictor.predict_core(s, Q), Q) fused; the building block of
 * the sampling optimizers that are not MPPI. */
int cpmppi_rollout_cost(cpmppi_handle* h, uint32_t E, const float* s0, const float* inputs, const float* target_position,
                        const float* target_equilibrium, const float* L, float* S_out, void* stream);

/* Rollout + plugin cost + its gradient with respect to the inputs: S[E,N] (may be NULL) and
 * grad[E,N,H] = d get_trajectory_cost(predict_core(s, Q), Q, previous_input) / d Q — what TensorFlow's GradientTape
 * hands to the gradient-based optimizers of the absent Control_Toolkit (Control_Toolkit_ASF/config_optimizers.yml:49-86,
 * sections gradient-tf and rpgd; :21-48 the CEM+gradient hybrids).  FAST arithmetic, plugin costs only; the edge bounce
 * is differentiated along the branch taken, indicators and a clipped control contribute zero.  previous_input[E] may be
 * NULL (0).  Allocates H*6*E*N floats of check-points on first use. */
int cpmppi_rollout_cost_grad(cpmppi_handle* h, uint32_t E, const float* s0, const float* inputs,
                             const float* target_position, const float* target_equilibrium, const float* L,
                             const float* previous_input, float* S_out, float* grad_out, void* stream);

/* One Adam iteration on the input sequences Q[E,N,H] (in place; m, v are the caller-owned moment buffers, zero before
 * iteration 1): per-rollout gradient-norm clipping to gradmax_clip (<= 0: off), Keras-style bias correction, then the
 * clip to [action_low, action_high] (config_optimizers.yml:52-58,69-73). */
int cpmppi_adam_step(cpmppi_handle* h, uint32_t E, float* Q, const float* grad, float* m, float* v, uint32_t iteration,
                     float learning_rate, float beta1, float beta2, float epsilon, float gradmax_clip, void* stream);

/* Plain gradient step Q <- clip(Q - learning_rate * clip_by_norm(grad, gradmax_clip)) on Q[E,N,H] in place
 * (cem-naive-grad-tf, config_optimizers.yml:21-31). */
int cpmppi_sgd_step(cpmppi_handle* h, uint32_t E, float* Q, const float* grad, float learning_rate, float gradmax_clip,
                    void* stream);

/* CEM (hyper-parameters: Control_Toolkit_ASF/config_optimizers.yml:1-11, section cem-tf).
 * cpmppi_cem_sample: Q[E,N,H] = clip(mean[E,H] + stdev[E,H] * z), z ~ N(0,1) from Philox(seed, offset, env, rollout).
 * cpmppi_cem_update: per env, the best_k sequences by cost (stable ascending order) -> their mean and population
 * standard deviation per time-step, the latter floored at stdev_min; elite_idx_out[E,best_k] (may be NULL). */
int cpmppi_cem_sample(cpmppi_handle* h, uint32_t E, const float* mean, const float* stdev, uint64_t seed, uint64_t offset,
                      uint32_t env_offset, float* Q_out, void* stream);
int cpmppi_cem_update(cpmppi_handle* h, uint32_t E, const float* S, const float* Q, uint32_t best_k, float stdev_min,
                      float* mean_out, float* stdev_out, uint32_t* elite_idx_out, void* stream);
/* cem-gmm (config_optimizers.yml:12-20): Q[E,N,H] = clip(centres[E, c, :] + stdev[E,H] * z) with the component c of every
 * rollout uniform over the K centres (the elite sequences of the previous iteration); component_out[E,N] may be NULL. */
int cpmppi_cem_gmm_sample(cpmppi_handle* h, uint32_t E, const float* centres, uint32_t K, const float* stdev, uint64_t seed,
                          uint64_t offset, uint32_t env_offset, float* Q_out, uint32_t* component_out, void* stream);

/* a16 alone: S[E,N], delta_u[E,N,H] -> weighted average [E,H] (controller_mppi_cartpole.py:306-321). */
int cpmppi_reward_weighted_average(cpmppi_handle* h, uint32_t E, const float* S, const float* delta_u, float* out,
                                   void* stream);

/* Plant (the CALLER of the hot path; SURVEY.md §8b harness row / §8f N1): advance E simulated cartpoles by
 * n_substeps simulation steps of dt_sim under the held controls Q[E] — Euler-Cromer + edge bounce + cos/sin + wrap as
 * CartPole/__init__.py:283-324 (cartpole_equations.py:367-378, :341-347).  s[E,6] is updated in place. */
int cpmppi_plant_advance(cpmppi_handle* h, uint32_t E, float* s, const float* Q, const float* L, uint32_t n_substeps,
                         float dt_sim, void* stream);

/* The same advance with the closed loop's recording in the same launch (the experiment loop of
 * CartPole/__init__.py:659-735 appends one row per control period): Q_log[row][E] = Q and states_log[row + 1][E][6] =
 * the advanced state; either log may be NULL.  row = the control-step index, or, when row_dev is given, *row_dev - 1:
 * the device step counter of cpmppi_step_args.offset_dev, which the preceding cpmppi_step has already advanced (a
 * captured graph of control steps then replays without any changing launch argument).  log_rows = the number of
 * control periods the logs hold (Q_log[log_rows][E], states_log[log_rows + 1][E][6]): a host `row` >= log_rows is
 * CPMPPI_ERR_BAD_ARG; a device counter that is still 0 or points past the logs advances the plant WITHOUT recording
 * (a graph replayed beyond the recording's end never writes outside the buffers). */
int cpmppi_plant_advance_record(cpmppi_handle* h, uint32_t E, float* s, const float* Q, const float* L, uint32_t n_substeps,
                                float dt_sim, float* states_log, float* Q_log, uint64_t log_rows, uint64_t row,
                                const void* row_dev, void* stream);

/* The plant with the reference's EXPERIMENT SCHEDULE: what CartPole.update_state does around the controller on
 * every SIMULATION step (CartPole/__init__.py:283-324) - update_parameters (:529-537: the pole length may change in time),
 * update_target_position (:360-378: target_position = random_track_f(time)), update_target_equilibrium (:380-388: flips after
 * keep_target_equilibrium_x_seconds_up / _down), integration + bounce + wrap, the controller every dt_control (Update_Q :475-527),
 * the second derivatives, save_csv_routine every dt_save (:403-433).  The schedule is a function of TIME only, so the host
 * tabulates it once per experiment (cartpolesimulation_amd/schedule.py) and the device loop indexes the tables with its own step
 * counter: a captured graph of control periods still replays without a changing launch argument.
 *   simulation step g = 0 is the initial state; control period c advances steps c*period_steps + 1 .. (c + 1)*period_steps;
 *   table row of step g = min(g / sched_stride, sched_rows - 1): the value the simulator holds AFTER step g's updates.
 * One call = one control period, ONE kernel:
 *   1. held control q = Q[env]; Q_log[c][E] = q (c < ctrl_rows); with a Q_disturbance_table the plant is driven by
 *      q_applied = (q + Q_disturbance_table[c][env]) + Q_bias instead (add_control_noise after Update_Q, :523-524)
 *   2. second derivatives of (s, q) with the pole length of step c*period_steps; dd_log[r][E][2] = (angleDD, positionDD) if that
 *      step is a saved one (r = step / save_every < save_rows) - the row whose state the PREVIOUS period stored, completed with
 *      the control computed from it (save_csv_routine runs after Update_Q, :316-324)
 *   3. n_substeps x { pole length and pole mass of the step (L_table, else L / config.L_default; m_pole_table, else m_pole /
 *      config.m_pole: update_parameters, :529-537); Euler-Cromer + edge bounce + cos / sin + wrap;
 *      second derivatives; if step % save_every == 0: states_log[step / save_every][E][6] = state, and dd_log unless the step ends
 *      the period }
 *   4. *_out[E] = row ((c + 1)*period_steps) of the tables: what the NEXT controller call is handed (updated_attributes
 *      target_position / target_equilibrium / L, :509-520) - point cpmppi_step_args.target_position / target_equilibrium / L there.
 * n_substeps = 0 records only (1. and 2.: the run's last controller call is followed by no plant step, :690-716).
 * period_dev: c = *period_dev - 1, the device step counter of cpmppi_step_args.offset_dev (already advanced by the step); a
 * counter that is still 0 advances the plant from table row 0 WITHOUT recording or publishing.  Any log / table / out pointer may
 * be NULL.  The pole MASS the controller computes with is
 * the handle's (config.m_pole / cpmppi_set_pole_mass), whatever the plant's: a per-env controller-side mass does not exist. */
typedef struct {
  uint32_t E;
  float* s;                             /* [E,6] in / out */
  const float* Q;                       /* [E] held controls */
  const float* L;                       /* [E] pole length when there is no L_table; NULL = config.L_default */
  uint32_t n_substeps;                  /* simulation steps to advance now: period_steps (fewer: a run's trailing partial period), or 0 (record only) */
  uint32_t period_steps;                /* simulation steps per control period (dt_control / dt_sim); 0 = n_substeps */
  float dt_sim;
  uint64_t period;                      /* c */
  const void* period_dev;               /* or NULL */
  float* states_log;                    /* [save_rows][E][6]; row 0 is the caller's (the initial state) */
  float* dd_log;                        /* [save_rows][E][2] */
  uint64_t save_rows;
  uint32_t save_every;                  /* simulation steps per saved row (dt_save / dt_sim); 0 = period_steps */
  float* Q_log;                         /* [ctrl_rows][E] */
  uint64_t ctrl_rows;
  const float* target_position_table;   /* [sched_rows][E] */
  const float* target_equilibrium_table;/* [sched_rows][E] */
  const float* L_table;                 /* [sched_rows][E] */
  uint64_t sched_rows;
  uint32_t sched_stride;                /* simulation steps per table row; 0 = 1 */
  float* target_position_out;           /* [E] */
  float* target_equilibrium_out;        /* [E] */
  float* L_out;                         /* [E] */
  uint32_t row_envs;                    /* envs per ROW of the logs and tables (0 = E): an env group that works on a slice of a larger
                                           batch's buffers passes the batch's env count and pointers to its first env (cpmppi_groups_run) */
  const float* m_pole;                  /* [E] the PLANT's pole mass when there is no m_pole_table; NULL = config.m_pole */
  const float* m_pole_table;            /* [sched_rows][E] a pole mass that changes in time, like L_table (the `m_pole:` updater) */
  const float* L_controller_table;      /* [sched_rows][E] what L_out publishes instead of L_table: the pole length the CONTROLLER is
                                           told (inform_controller_about_parameters_change, CartPole/controller_informer.py: the true
                                           value or the initial one); NULL = L_table */
  const float* Q_disturbance_table;     /* [ctrl_rows][E] the simulator's additive control disturbance (CartPole/noise_control_signal.py:
                                           14-16): the plant of period c is driven by Q_applied = (Q + table[c]) + Q_bias in float32,
                                           table = controlDisturbance * N(0,1) drawn on the host; Q_log keeps the CALCULATED control.
                                           NULL = none.  Needs ctrl_rows > 0 */
  float Q_bias;                         /* controlBias */
  float* Q_applied_out;                 /* [E] the control the plant was driven by this period: what the simulator hands the NEXT controller
                                           call as Q_ccrc / "Q_applied_-1" (CartPole/__init__.py:489, 517-518) - point
                                           cpmppi_step_args.previous_input there for the costs that read it; NULL = not needed */
  /* The measurement chain between plant and controller (CartPole.add_noise_and_latency, CartPole/__init__.py:336-356; latency 0,
   * noise OFF, offset 0 as shipped): with s_measured given, a period that is followed by a controller call ends with
   *   delayed  = state(g - latency_steps) + latency_frac * (state(g - latency_steps - 1) - state(g - latency_steps))   (float64;
   *              before the first step: zeros with cos = 1, CartPole/latency_adder.py:24-26, 63-67)
   *   noise      (measurement_noise_table row c + 1: angle += n0, wrap, cos / sin, position += n1, angleD += n2, positionD += n3)
   *   offset     angle = wrap(angle + angle_offset), cos / sin (:348-356); informed (informed_table, NULL = yes): taken out again
   *              with another wrap + cos / sin (:501-505)
   * and s_measured[E][6] = that state in float32 - point cpmppi_step_args.s0 there (the caller puts the initial state in before
   * the t = 0 call, which sees the true state, :869-870). */
  float* s_measured;                    /* [E][6] out; NULL = no measurement chain */
  float* state_history;                 /* [history_len][row_envs][6] ring: the state after simulation step g at slot g % history_len; the
                                           caller fills it with zeros and cos = 1 before the run.  Needed when latency > 0 */
  uint32_t history_len;                 /* >= latency_steps + 2 */
  uint32_t latency_steps;               /* int(latency / dt_sim) */
  double latency_frac;                  /* latency / dt_sim - latency_steps */
  const float* measurement_noise_table; /* [ctrl_rows][row_envs][4] sigma * N(0,1) for angle, position, angleD, positionD, or NULL */
  const double* angle_offset_table;     /* [sched_rows][row_envs] the vertical angle offset after the table row's step, or NULL = 0 */
  const uint8_t* informed_table;        /* [sched_rows][row_envs] 1 = the controller is informed (the offset is taken out), NULL = 1 */
} cpmppi_plant_args;
int cpmppi_plant_step(cpmppi_handle* h, const cpmppi_plant_args* args, void* stream);

/* Multi-GPU (no reference counterpart - its only fan-out is share-nothing SLURM job arrays,
 * others/EulerClusterScripts/ParallelDataGeneration.sh:2-17): one process per GPU, each owning a contiguous block of
 * envs, and ONE all-gather of the chosen control sequences per step over RCCL / xGMI, enqueued from C on a high-priority
 * side stream so that it runs under the NEXT step's rollout kernel:
 *   cpmppi_comm_unique_id  rank 0: the 128-byte RCCL id every rank needs (distribute it by any means - a
 *                          torch.distributed store, MPI, a file); rccl_path may be NULL (the RCCL already in the process,
 *                          else librccl.so by name)
 *   cpmppi_comm_init       collective over all ranks: creates the handle's communicator, side stream and events
 *   cpmppi_comm_gather     after step i on `stream`: recv_all[world][count] <- all-gather of send[count] on the side stream,
 *                          ordered after everything enqueued on `stream` so far; `stream` itself does not wait.  slot <
 *                          CPMPPI_COMM_SLOTS names the completion event of this gather
 *   cpmppi_comm_wait       `stream` waits (on the device) for the gather of `slot` - call it before the step that overwrites
 *                          that gather's send buffer (with the two u_nom buffers of cpmppi_step_args.u_nom_out: two steps later)
 *                          or before reading recv_all on `stream`
 *   cpmppi_comm_sync       host wait for every gather enqueued so far; reports (once) and clears a device-side timeout
 *   cpmppi_comm_set_timeout  how long a device-side wait of cpmppi_step_gather may last, in seconds (default 10; <= 0: for
 *                          ever): the finalize of a step that is about to overwrite a buffer an all-gather still reads
 *                          waits for that gather, which completes only when EVERY rank has joined it
 *   cpmppi_comm_get_info   what the communicator IS, as RCCL itself reports it (ncclCommCount / ncclCommUserRank), next to what
 *                          cpmppi_comm_init was told: a bench line can then state how many ranks the collective really spanned
 * The side stream's wait for a published step is a one-lane kernel of the library's with the same timeout (the default form); the
 * stream-memory-operation form of rounds 4-5 (environment: CPMPPI_COMM_WAITER=stream-ops; hipStreamWaitValue32 has no timeout of its
 * own) is covered by cpmppi_comm_sync and cpmppi_comm_destroy, which poll the side stream for at most the timeout and then release
 * the wait from the host, raise the error and return CPMPPI_ERR_COMM.  Either way a rollout launch that never published - failed,
 * aborted - cannot wedge them.  cpmppi_step_gather refuses a stream that is being captured.
 * PEERS: a rank whose device-side wait timed out keeps its buffers intact and its already-enqueued all-gathers still run, so the
 * other ranks receive a well-formed but STALE block from it and no error of their own.  Two ways to know:
 *   - cpmppi_comm_set_stamped(h, 1) (before the first cpmppi_step_gather, the same on every rank): every gathered block carries
 *     CPMPPI_GATHER_STAMP_FLOATS trailing words - the u_nom / u_nom_out buffers handed to cpmppi_step_gather are then
 *     [E*H + CPMPPI_GATHER_STAMP_FLOATS] floats (the caller zeroes the trailing words once), recv_all is
 *     [world][E*H + CPMPPI_GATHER_STAMP_FLOATS].  Word 0 of the trailer (read as uint32_t) is the STAMP: the number of the
 *     cpmppi_step_gather call (1, 2, ... per communicator; the same on every rank, collectives being called in the same order) that
 *     produced the block, written on the side stream between the step's publication and its all-gather - unless a device-side
 *     wait of the communicator has given up: a rank that dropped a step (timeout) leaves buffer AND stamp as they were, and stamps
 *     nothing until cpmppi_comm_sync has cleared the condition.  A receiver accepts block r of the n-th gather iff its stamp == n
 *     and otherwise keeps what it had for rank r (it may so discard a good block that was gathered while the error was up - never
 *     accept a stale one).  No extra collective, nothing on the launch stream; words 1.. of the trailer are reserved (0).
 *   - or treat gathered blocks as unverified until cpmppi_comm_sync has returned CPMPPI_OK on EVERY rank (exchange the return
 *     codes with the collective of your choice).
 * Errors: CPMPPI_ERR_COMM.  RCCL is bound at run time: the library loads without it.  rccl_path, when given, is loaded as given
 * and wins over an RCCL the process already holds (a site-specific build; the two-process test double of tests/fake_rccl). */
typedef struct {
  uint32_t world, rank;            /* as given to cpmppi_comm_init */
  int32_t rccl_ranks, rccl_rank;   /* ncclCommCount / ncclCommUserRank of the communicator; -1 = this RCCL does not export them */
  int32_t rccl_version;            /* ncclGetVersion (e.g. 22606), 0 = unknown */
  uint32_t stream_memory_ops;      /* 0 = the one-lane waiter kernel orders the side stream (default), 1 = hipStreamWaitValue32 / WriteValue32 */
  uint32_t gathers_enqueued;       /* cpmppi_step_gather calls so far (= the stamp of the most recent one) */
  uint32_t stamped;                /* cpmppi_comm_set_stamped */
} cpmppi_comm_info;
#define CPMPPI_COMM_ID_BYTES 128
#define CPMPPI_COMM_SLOTS 4
#define CPMPPI_GATHER_STAMP_FLOATS 4   /* 16 bytes: every rank's block in recv_all stays 16-byte aligned */
int cpmppi_comm_unique_id(void* id_out, const char* rccl_path);
int cpmppi_comm_init(cpmppi_handle* h, const void* id, int world, int rank, const char* rccl_path);
int cpmppi_comm_gather(cpmppi_handle* h, uint32_t slot, const float* send, float* recv_all, size_t count, void* stream);
int cpmppi_comm_wait(cpmppi_handle* h, uint32_t slot, void* stream);
int cpmppi_comm_sync(cpmppi_handle* h);
int cpmppi_comm_set_timeout(cpmppi_handle* h, double seconds);
int cpmppi_comm_destroy(cpmppi_handle* h);
int cpmppi_comm_get_info(cpmppi_handle* h, cpmppi_comm_info* out);
int cpmppi_comm_set_stamped(cpmppi_handle* h, int on);

/* cpmppi_step + the all-gather of its result in ONE call - the production form of the per-step collective:
 * recv_all[world][E*H] <- all-gather of the nominal sequences this step writes (args->u_nom_out, or args->u_nom when the
 * step runs in place).  The launch stream receives the rollout kernel and nothing else; step and gather are ordered
 * through device memory (the kernel's finalizing blocks publish the step number, a one-lane kernel on the side stream waits
 * for it - hipStreamWaitValue32 on signal memory with CPMPPI_COMM_WAITER=stream-ops; the finalize of a
 * later step that overwrites a buffer still being gathered waits for that gather).  Use two
 * u_nom buffers alternately (step i: u_nom = B[i & 1], u_nom_out = B[(i + 1) & 1]) so that the gather of step i runs
 * under step i + 1; in place is correct too, but then step i + 1's finalize waits for gather i.  recv_all must stay
 * untouched until cpmppi_comm_sync (host) returns or a later cpmppi_comm_gather/wait pair orders the reader.
 * A launch of more than 512 envs performs that wait once, with one lane, in FRONT of the kernel too (a one-lane kernel on `stream`,
 * ~5 us): thousands of finalizing blocks spinning for a late gather would hold every workgroup slot of the device (two ranks sharing
 * one device deadlocked on that until the timeout) - so with many envs and in place the step no longer overlaps the previous gather:
 * alternate two buffers.
 * A device-side wait that outlasts cpmppi_comm_set_timeout does NOT proceed: the step drops its result (the buffer being
 * gathered is not overwritten), so does every later step of the handle, and the next cpmppi_step_gather - or
 * cpmppi_comm_sync, which also clears the condition - returns CPMPPI_ERR_COMM. */
int cpmppi_step_gather(cpmppi_handle* h, const cpmppi_step_args* args, float* recv_all, void* stream);

/* Recording writer (HOST pointers, no GPU involved): E experiment recordings in the reference's CSV layout
 * (CartPole/csv_logger.py:10-58,125-159; column set and order CartPole/__init__.py:221-259), byte for byte what the reference's
 * csv.writer produces from the values its simulator logs - pinned to a recording written by the reference itself
 * (tests/golden/schedule.npz "csv_rows"): every column in the representation of the TYPE the reference holds it in -
 *   time, Q_calculated, target_position, L, m_pole, the vertical-angle-offset columns: Python floats -> repr(float) of the double
 *   the state, angleDD / positionDD, Q_applied, Q_ccrc, u: numpy float32 scalars -> str(numpy.float32), shortest float32 digits
 *   target_equilibrium: an int; L_for_controller / m_pole_for_controller: the controller informer's 'true' / 'default'
 *   Q_update_time: empty until the first controller update inside the loop, then `q_update_time` (the reference logs the wall
 *   clock of its controller call there: not reproducible by construction)
 * rows end with "\r\n".  One thread per file.
 *   paths[E]      one file per experiment; a path that already exists is an error (CPMPPI_ERR_IO) - the caller makes the names
 *                 unique (csv_logger.py:61-91) and nothing is ever appended to or overwritten.  Files are written under a
 *                 temporary name and renamed when complete; on any failure the files of THIS call are removed again.
 *   preamble      the comment block and the column-name row as ready-made bytes, the same for every file
 *   n_threads     0 = one per hardware thread (at most 32) */
typedef struct {
  uint32_t E, rows;
  const double* time;                   /* [rows] */
  const float* states;                  /* [rows][E][6] */
  const float* dd;                      /* [rows][E][2] angleDD, positionDD */
  const float* Q;                       /* [rows][E] the control in force when the row was saved (Q_calculated = Q_applied) */
  const float* Q_ccrc;                  /* [rows][E] the control before it */
  const double* target_position;        /* [rows][E] */
  const int32_t* target_equilibrium;    /* [rows][E] */
  const float* L;                       /* [rows][E] */
  double m_pole;
  float u_max;                          /* u = u_max * Q in float32 (CartPole/cartpole_equations.py:119-127) */
  uint32_t first_update_row;            /* rows before it have an empty Q_update_time */
  double q_update_time;
  const float* m_pole_rows;             /* [rows][E] a pole mass that changed during the run (the float32 values the simulator held), or
                                           NULL: `m_pole` in every row */
  const uint8_t* informed;              /* [rows][E] L_for_controller / m_pole_for_controller: 1 = 'true', 0 = 'default' (the controller
                                           informer's state when the row was saved); NULL = 'true' in every row (mode ON, as shipped) */
  const float* Q_applied;               /* [rows][E] the control the plant was driven by when it differs from the calculated one (control
                                           disturbance): the Q_applied and u columns; NULL = Q */
  const double* angle_offset;           /* [rows][E][3] the vertical angle offset, its cos and its sin when the row was saved, or NULL:
                                           0.0, 1.0, 0.0 in every row */
} cpmppi_recording;
int cpmppi_write_recordings(const char* const* paths, const char* preamble, size_t preamble_len, const cpmppi_recording* rec,
                            int n_threads);

/* A HIP stream with a hardware queue OF ITS OWN (hipExtStreamCreateWithCUMask with every CU enabled): the runtime multiplexes
 * ordinary streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and two streams that land on the same
 * queue run their kernels one after the other - which defeats env groups that are meant to run side by side
 * (cartpolesimulation_amd/pipeline.py; measured: two groups of C3 on pooled streams 327 us per step, serialised, against 175 us).
 * The returned hipStream_t is an ordinary stream for every other purpose; destroy it with cpmppi_stream_destroy. */
int cpmppi_stream_create(int device, void** stream_out);
int cpmppi_stream_destroy(void* stream);

/* ENV GROUPS: the E envs of one device as `groups` contiguous groups, each with a handle and a dedicated-queue stream of its own,
 * every group running its OWN chain of launches - independent MPPI problem instances need not march in step (a launch of a few
 * dozen envs ends with its slowest wave, waves differ by 20-40 %; measured on MI355X: 64 envs x 2048 x 50 74.5 -> 63.8 us per
 * step of all envs with two groups, 64 x 4096 x 100 224.6 -> 177.1 us).  No reference counterpart beyond its share-nothing job arrays
 * (others/EulerClusterScripts/ParallelDataGeneration.sh:2-17).  Philox keys are GLOBAL env indices (env_offset + env): a result
 * does not depend on the split (bit-identical to the unsplit launch whenever both pick the same lane mapping).
 *   cpmppi_groups_create   cfg->E = all envs of the device; env_offset = global index of env 0 (rank * E when sharded over GPUs)
 *   cpmppi_groups_slice / _handle / _stream   a group's envs [first, first + n), its handle (cost weights, GRU model, ... are set
 *                          per handle) and its hipStream_t
 *   cpmppi_groups_fork     every group stream waits for what `stream` has enqueued so far (uploads, allocations)
 *   cpmppi_groups_join     `stream` waits for every group
 *   cpmppi_groups_run      `periods` control periods of EVERY group enqueued from C, round robin (period k, group g: cpmppi_step
 *                          with the Philox step counter step->offset + k, then - if `plant` is given - cpmppi_plant_step with period
 *                          plant->period + k).  `step` and `plant` describe the FULL [E, ...] arrays exactly as for one handle over
 *                          all envs; every group works on its slice of them in place (plant->row_envs is filled in).  No group waits
 *                          for another.  step->offset_dev / plant->period_dev (one shared device counter) are refused.
 *   cpmppi_groups_comm_init   the env groups of a device under ONE communicator and ONE side stream (collective over all ranks, as
 *                          cpmppi_comm_init).  The communicator lives in group 0's handle: cpmppi_comm_set_timeout / _set_stamped /
 *                          _sync / _get_info / _destroy take cpmppi_groups_handle(g, 0)
 *   cpmppi_groups_run_gather  cpmppi_groups_run + the per-step all-gather of cpmppi_step_gather: per period ONE all-gather of the
 *                          device's whole u_nom[E, H] (recv_all[world][E*H], + the stamp words when stamped) on the side stream.  The
 *                          finalizing blocks of ALL groups count themselves on the communicator's shared arrival counter
 *                          (alternating with the step's parity: the groups drift by at most one step), the last env of the last
 *                          group publishes the step, and a group's finalize that is about to overwrite sequences a gather still
 *                          reads waits for it - exactly the single-handle protocol, with `envs` = all envs of the device.  With
 *                          step->u_nom_out given, the two buffers alternate per period of the call (period 0 reads u_nom and writes
 *                          u_nom_out, period 1 the other way round, ...); NULL = in place.  recv_all holds the LAST period's gather
 *                          once cpmppi_comm_sync(cpmppi_groups_handle(g, 0)) has returned.
 * Errors as for the single-handle calls; text in cpmppi_groups_last_error. */
typedef struct cpmppi_groups cpmppi_groups;
int cpmppi_groups_create(const cpmppi_config* cfg, int device, uint32_t groups, uint32_t env_offset, cpmppi_groups** out);
void cpmppi_groups_destroy(cpmppi_groups* g);
uint32_t cpmppi_groups_count(const cpmppi_groups* g);
int cpmppi_groups_slice(const cpmppi_groups* g, uint32_t group, uint32_t* first_env, uint32_t* n_envs);
cpmppi_handle* cpmppi_groups_handle(cpmppi_groups* g, uint32_t group);
void* cpmppi_groups_stream(cpmppi_groups* g, uint32_t group);
int cpmppi_groups_fork(cpmppi_groups* g, void* stream);
int cpmppi_groups_join(cpmppi_groups* g, void* stream);
int cpmppi_groups_run(cpmppi_groups* g, const cpmppi_step_args* step, const cpmppi_plant_args* plant, uint32_t periods);
int cpmppi_groups_comm_init(cpmppi_groups* g, const void* id, int world, int rank, const char* rccl_path);
int cpmppi_groups_run_gather(cpmppi_groups* g, const cpmppi_step_args* step, const cpmppi_plant_args* plant, uint32_t periods,
                             float* recv_all);
const char* cpmppi_groups_last_error(const cpmppi_groups* g);

/* The ABI version this library was BUILT as (CPMPPI_ABI_VERSION of its header): lets a client that was compiled against another
 * header notice before it passes a struct of the wrong layout (cpmppi_create refuses a mismatching cpmppi_config.abi_version). */
uint32_t cpmppi_abi_version(void);

/* Build info, e.g. "cpmppi 1 gfx950 hip-7.2". */
const char* cpmppi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CPMPPI_H */
