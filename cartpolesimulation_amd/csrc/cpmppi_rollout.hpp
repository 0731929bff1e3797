// cpmppi_rollout.hpp — the hot path: rollout_cost_kernel and what it shares with the other kernels (launch
// descriptor, nominal-sequence shift, the per-env finalize).  A header because the kernel is instantiated in two
// translation units compiled with different instruction-scheduling strategies (cpmppi_rollout_latency.hip /
// cpmppi_rollout_throughput.hip): a launch of at most one wave per SIMD is bound by the latency of a single wave's
// instruction stream, larger ones by issue throughput, and the compiler's schedulers differ measurably on the two.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cpmppi.h"
#include "cpmppi_device.hpp"

namespace cpmppi_k {
using namespace cpmppi;


constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
#ifndef CPMPPI_GRU_MIN_WAVES
#define CPMPPI_GRU_MIN_WAVES 2      // waves per SIMD the GRU kernels are compiled for (register budget 512 / this)
#endif
#ifndef CPMPPI_MIN_WAVES
#define CPMPPI_MIN_WAVES 1
#endif
#ifndef CPMPPI_DMA_TK
#define CPMPPI_DMA_TK 8             // control steps per direct-to-LDS tile (two tiles per wave)
#endif
#ifndef CPMPPI_DMA_TK_THROUGHPUT
#define CPMPPI_DMA_TK_THROUGHPUT 16 // ... in the throughput build (one tile per wave)
#endif
#ifndef CPMPPI_NOMINAL_IN_LANES
#define CPMPPI_NOMINAL_IN_LANES 3   // FAST, bit v = build VARIANT v: nominal sequence held in lanes, fetched with v_readlane_b32 (latency + throughput builds)
#endif
#ifndef CPMPPI_EVENTFUL_UNROLL
#define CPMPPI_EVENTFUL_UNROLL 1
#endif
#ifndef CPMPPI_ODE_TRACK_NEAR
#define CPMPPI_ODE_TRACK_NEAR 1     // predictor_ODE: boundary-cost flag from one pair of compares per control step (A/B switch)
#endif
#ifndef CPMPPI_WAVE_PRIORITY
#define CPMPPI_WAVE_PRIORITY 1
#endif
#ifndef CPMPPI_KNOT_SEGMENTS
#define CPMPPI_KNOT_SEGMENTS 1      // throughput build, in-kernel interpolation: knot segments as an outer loop (A/B switch)
#endif
#ifndef CPMPPI_QBGM_ACC
#define CPMPPI_QBGM_ACC 1           // FAST quadratic_boundary_grad_minimal: stage cost + correction accumulated with FMAs (A/B switch)
#endif
#ifndef CPMPPI_ENV_FOLD
#define CPMPPI_ENV_FOLD 1           // throughput build: per-env constants from fold_env_kernel's block instead of each wave's prologue (A/B switch)
#endif
#ifndef CPMPPI_TILED_SCATTER
#define CPMPPI_TILED_SCATTER 1      // tiled layout: the weighted column sums as a reduce-scatter over the wave (A/B switch)
#endif
#ifndef CPMPPI_ROLLBACK_PHASED
#define CPMPPI_ROLLBACK_PHASED 1    // phased mid-size build: the quiet control step with one edge test per three substeps too (A/B switch)
#endif
#ifndef CPMPPI_SPIN_BRANCH
#define CPMPPI_SPIN_BRANCH 1        // throughput build, two rollouts per lane: the spin test as one v_max + compare + branch (A/B switch)
#endif
constexpr size_t SAMPLER_LDS_MAX = 159 * 1024;   // gfx950: 160 KB of LDS per workgroup (sampler: [256][P+1] floats)

// Device-side ordering between a step and the all-gather of its result (cpmppi_step_gather, cpmppi_comm.hip) without any
// packet on the launch stream: flags[0] envs finalized by this launch, flags[1] steps published (read by the fallback
// waiter kernel), flags[2] gathers completed (written by the side stream), flags[3] a wait gave up (sticky until
// cpmppi_comm_sync: while it is set nothing waits and nothing is stored).
struct GatherSync {
  uint32_t* flags;          // NULL = no gather follows this step.  The block also holds what only the slow paths need, so that
                            // the kernel argument stays four words: [4,5] pointer to the signal memory the side stream's
                            // hipStreamWaitValue32 watches (0: the waiter kernel polls flags[1]), [6,7] pointer to the pinned
                            // host word that mirrors the error, [8,9] the 100 MHz ticks a wait may last (~0 = for ever)
  uint32_t publish;         // the step number every env of this launch publishes once its nominal sequence is written
  uint32_t need;            // flags[2] must have reached this before the output buffer may be overwritten (0 = no wait)
  uint32_t envs;            // envs in this launch
};
constexpr int GS_PUBLISHED = 4, GS_ERR_HOST = 6, GS_TIMEOUT = 8, GS_WORDS = 16;
__device__ __forceinline__ uint64_t gs_word64(const uint32_t* flags, int i) {
  return (uint64_t)flags[i] | ((uint64_t)flags[i + 1] << 32);
}

// flag >= need (wrap-safe), polled with system-scope loads (the side stream's hipStreamWriteValue32 is a write of the command
// processor: not through this XCD's L2).  Returns false - after raising the error for device and host - when the wait
// outlasts `timeout_ticks` or the error is already up: the caller then does NOT proceed to the stores the wait guards.
__device__ __forceinline__ bool spin_until_reached(uint32_t* flag, uint32_t need, const GatherSync& gs) {
  uint32_t* err = gs.flags + 3;
  if ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - need) >= 0)
    return __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    __builtin_amdgcn_s_sleep(4);
    if ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - need) >= 0)
      return __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
    if (__builtin_amdgcn_s_memrealtime() - t0 > gs_word64(gs.flags, GS_TIMEOUT)) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t* err_host = reinterpret_cast<uint32_t*>(gs_word64(gs.flags, GS_ERR_HOST));
      if (err_host) __hip_atomic_store(err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
  }
}

struct StepPtrs {
  const float* s0;
  const float* u_nom;
  const float* u_prev;
  const float* x_t;
  const float* te;
  const float* L;
  const float* noise;
  const float* prev_in; // [E] control applied before this step (quadratic_boundary_grad ccrc) or NULL
  uint64_t seed, offset;
  const unsigned long long* offset_dev;   // if set: the Philox step counter lives in device memory (graph replay)
  uint32_t stash;       // NOISE_PHILOX: the generated knots are parked in LDS ([P][R][BLOCK] after the weighted sums) for the reduction
  uint32_t env_offset;
  uint32_t nb;          // blocks per env
  uint32_t W;           // width of the weighted-sum vector (H in delta_u space, P in knot space)
  float* S_out;
  float* partial;       // [E][nb][2 + W]
  uint32_t* counter;    // [E] arrival tickets of the env's blocks (0 between launches); NULL = separate finalize kernel
  float* u_nom_out;     // fused finalize: where the updated nominal sequence goes (u_nom itself, or the caller's second buffer)
  float* Q_out;
  uint32_t* host_ticket; // cpmppi_step_host: counter in pinned host memory, +1 (system scope) per finalized env; NULL otherwise
  GatherSync gs;
  const EnvFold* env_fold;   // [envs of this launch] per-env constants (throughput build, FAST, predictor_ODE_v0: launch_rollout_math fills it first;
                             // last: every other field keeps the kernarg offset the latency builds were tuned with)
};

// 16 bytes per lane from a per-lane global address straight into LDS at (wave-uniform `lds`) + 16 * lane - gfx950's
// global_load_lds_dwordx4; tracked by vmcnt.  (The builtin exists in the device pass only; the host pass, which merely
// emits the kernel's launch stub, sees an empty body.)
__device__ __forceinline__ void load16_to_lds(const float* gptr, float* lds) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(gptr, lds, 16, 0, 0);
#else
  (void)gptr; (void)lds;
#endif
}

// The StepPtrs kernel argument re-read from the kernarg segment at the point of call (the kernels here take
// (const Params, const StepPtrs): the second argument sits at the first 8-byte boundary after the first).
__device__ __forceinline__ StepPtrs late_step_ptrs() {
  typedef const __attribute__((address_space(4))) char* kptr;
  kptr k = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(k));
  static_assert(alignof(StepPtrs) == 8 && sizeof(StepPtrs) % 8 == 0, "kernarg layout");
  const __attribute__((address_space(4))) uint64_t* w =
      (const __attribute__((address_space(4))) uint64_t*)(k + ((sizeof(Params) + 7u) & ~(size_t)7u));
  union { StepPtrs s; uint64_t w[sizeof(StepPtrs) / 8]; } u;
#pragma unroll
  for (size_t i = 0; i < sizeof(StepPtrs) / 8; ++i) u.w[i] = w[i];       // (only the words of fields used later survive)
  return u.s;
}

// Nominal control for stage k after the configured shift (a18).
__device__ __forceinline__ float shifted_nominal(const Params& p, const float* __restrict__ un, uint32_t k) {
  if (p.shift_mode == CPMPPI_SHIFT_NONE) return un[k];
  if (k + 1 < p.H) return un[k + 1];
  return (p.shift_mode == CPMPPI_SHIFT_REPEAT_LAST) ? un[p.H - 1] : 0.0f;
}

// Merge the per-block partials of one env (rescaled to the env-wide minimum), apply shift / update / clip, write u_nom
// and Q.  Executed by one whole block.  COHERENT = the partials were written by other workgroups of THIS launch: read
// them with agent-scope (sc1) loads that bypass this CU's L1.
template <bool KNOT_SPACE, bool COHERENT>
__device__ __forceinline__ void finalize_env(const Params& p, const float* partial, uint32_t nb, uint32_t W,
                                             const float* u_nom_in, float* u_nom_out, float* __restrict__ Q_out,
                                             uint32_t env, uint32_t* host_ticket = nullptr,
                                             const GatherSync gs = GatherSync{nullptr, 0u, 0u, 0u}) {
  __shared__ float u_new[CPMPPI_MAX_HORIZON];
  __shared__ float bz[KNOT_SPACE ? (CPMPPI_MAX_HORIZON + 2) : 1];
  const uint32_t tid = threadIdx.x, H = p.H;
  const float* pe = partial + (size_t)env * nb * (2 + W);
  auto ld = [&](size_t i) -> float {
    if constexpr (COHERENT) return __hip_atomic_load(pe + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return pe[i];
  };
  // One memory round trip per batch of eight blocks: the minimum, the weight sum and this thread's column of every block are
  // requested together (the loads do not depend on each other; issued one after the other as written they were three
  // dependent L2 round trips at the tail of every launch), then merged with the running minimum (identical arithmetic
  // to a min-first pass when nb <= 8).
  constexpr int NBB = 8;
  float a = 0.0f;
  auto merged = [&](uint32_t c) {
    float M = INFINITY, v = 0.0f;
    a = 0.0f;
    for (uint32_t b0 = 0; b0 < nb; b0 += NBB) {
      float mb[NBB], ab[NBB], vb[NBB];
#pragma unroll
      for (int u = 0; u < NBB; ++u) {
        const uint32_t b = (b0 + u < nb) ? b0 + u : nb - 1u;
        mb[u] = ld((size_t)b * (2 + W));
        ab[u] = ld((size_t)b * (2 + W) + 1);
        vb[u] = ld((size_t)b * (2 + W) + 2 + c);
      }
      float Mn = M;
#pragma unroll
      for (int u = 0; u < NBB; ++u) Mn = fminf(Mn, mb[u]);
      if (b0 != 0) {
        const float sc = expf((-1.0f / p.LBD) * (M - Mn));
        a *= sc;
        v *= sc;
      }
      M = Mn;
#pragma unroll
      for (int u = 0; u < NBB; ++u) {
        if (b0 + u < nb) {
          const float w = expf((-1.0f / p.LBD) * (mb[u] - M));
          a += ab[u] * w;
          v = __builtin_fmaf(vb[u], w, v);
        }
      }
    }
    return v;
  };
  if constexpr (KNOT_SPACE) {
    // every thread merges one column (clamped), so that every thread also holds the weight sum `a`
    const float v0 = merged(tid < W ? tid : W - 1u);
    if (tid < W) bz[tid] = v0;
    for (uint32_t c = tid + BLOCK; c < W; c += BLOCK) bz[c] = merged(c);
    __syncthreads();
  }
  const float* un = u_nom_in + (size_t)env * H;       // (may alias the output: every read precedes the barrier below)
  float* uo = u_nom_out + (size_t)env * H;
  for (uint32_t k = tid; k < H; k += BLOCK) {
    float bk;
    if constexpr (KNOT_SPACE) {
      const uint32_t j = k / p.period, i = k % p.period;
      bk = bz[j] + (bz[j + 1] - bz[j]) * ((float)i / (float)p.period);
    } else {
      bk = merged(k);
    }
    float v = shifted_nominal(p, un, k) + bk / a;
    if (p.control_mode == CPMPPI_CONTROL_CLIP) v = fminf(fmaxf(v, p.lo), p.hi);
    u_new[k] = v;
  }
  __syncthreads();                          // every read of the old nominal sequence is done
  bool store = true;
  if (gs.flags && gs.need) {
    // the all-gather that still reads the buffer written next (two steps back with alternating buffers) must be complete:
    // by now it has had a whole step to run, so this practically never spins.  A wait that gives up (a peer rank stalled
    // beyond cpmppi_comm_set_timeout) must NOT fall through to the stores - the gather would send a half-overwritten
    // buffer to every rank: this step's result is dropped instead, the error is raised for the host, and the launch still
    // publishes so that nothing behind it wedges.
    __shared__ uint32_t may_store;
    if (tid == 0) may_store = spin_until_reached(gs.flags + 2, gs.need, gs) ? 1u : 0u;
    __syncthreads();
    store = may_store != 0u;
  }
  if (store) {
    for (uint32_t k = tid; k < H; k += BLOCK) uo[k] = u_new[k];
    if (tid == 0 && Q_out) Q_out[env] = u_new[0];
  }
  if (tid == 0 && host_ticket) {
    // the simulator's host thread spins on this counter instead of waiting on the stream (cpmppi_step_host): Q_out lives
    // in the same pinned, fine-grained block; system-scope release so that the control is visible before the ticket
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_fetch_add(host_ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (gs.flags) {
    // publish "every env of this step has written its sequence" to the side stream's waiter: stores drained, block
    // barrier, one lane's agent-scope release + arrival count; the last env's block publishes the step number
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const uint32_t arrived = __hip_atomic_fetch_add(gs.flags, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (arrived == gs.envs - 1u) {
        __hip_atomic_store(gs.flags, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the side stream waits for this number: hipStreamWaitValue32 on signal memory (a one-lane blit kernel of the runtime's,
        // as the round-6 kernel trace shows), or - env groups; devices without stream memory operations - our one-lane kernel polling flags[1]
        uint32_t* published = reinterpret_cast<uint32_t*>(gs_word64(gs.flags, GS_PUBLISHED));
        if (published) __hip_atomic_store(published, gs.publish, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        else __hip_atomic_store(gs.flags + 1, gs.publish, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// The hot path.  R = rollouts per lane (1: latency mapping, 2: packed float2 throughput mapping, FAST only).
// A block of 256 threads owns 256*R consecutive rollouts of one env; wave w owns rows [w*64*R, (w+1)*64*R) and lane l
// integrates rows l (component 0) and l+64 (component 1).
// VARIANT selects the translation unit (hence the scheduling strategy) an instantiation is compiled in:
// 0 = latency build (one rollout per lane, launches of at most one wave per SIMD), 1 = throughput build, 2 = packed
// mapping for mid-sized launches (same code except where the loop constants live, see below).
// INTEG selects the in-tree ODE predictor the rollouts are integrated with: PREDICTOR_ODE_V0 (predictor_ODE_v0: simultaneous
// Euler, edge bounce, fmod wrap - the north-star path) or PREDICTOR_ODE (predictor_ODE: Euler-Cromer, no bounce, atan2 wrap;
// cpmppi_device.hpp).  The second has no events, hence no phased loop: it is built in the latency and throughput forms only.
template <int COST, bool FAST, int NOISE, int R, int VARIANT_, int INTEG = PREDICTOR_ODE_V0>
__global__ __launch_bounds__(BLOCK, CPMPPI_MIN_WAVES) void rollout_cost_kernel(const Params p, const StepPtrs a) {
  static_assert(INTEG == PREDICTOR_ODE_V0 || VARIANT_ != 2, "predictor_ODE: latency / throughput builds and the lone-wave form of the latter");
  // VARIANT_ 3 = the mid-size build for launches of at most ONE wave per SIMD: VARIANT 2 with the quiet control step's nine
  // substeps as straight-line code (a lone wave pays ~50 cycles per taken branch: C4 80.1 -> 77.4 us; with two or more waves
  // per SIMD the larger code costs 1.5-2.5 % instead, so those launches keep the loop)
  // (predictor_ODE has no events, hence no phased loop: its VARIANT_ 3 is the THROUGHPUT build's kernel with the substeps unrolled)
  constexpr int VARIANT = (VARIANT_ == 3) ? (INTEG == PREDICTOR_ODE ? 1 : 2) : VARIANT_;
  constexpr bool LONE_WAVE = VARIANT_ == 3;
  using F = typename Lanes<R>::F;
  static_assert(FAST || R == 1, "the PRECISE path is one rollout per lane");
  // wave-private tiles (direct-to-LDS loads): two of 8 control steps in the latency / mid-size builds (the next tile streams
  // in under the current one), ONE of 16 in the throughput build (same LDS; every 128-byte line of a 200-byte row is then
  // requested about twice instead of four times, and the three other waves of the SIMD cover the wait)
  constexpr uint32_t DMA_TK = (VARIANT == 1) ? CPMPPI_DMA_TK_THROUGHPUT : CPMPPI_DMA_TK;
  constexpr uint32_t DMA_BUFS = (VARIANT == 1) ? 1u : 2u;
  __shared__ float tile[NOISE == NOISE_DELTA_U ? WAVES * DMA_BUFS * 64 * R * DMA_TK : 1];
  __shared__ float red[2 * WAVES];
  extern __shared__ float bsum[];            // [WAVES][W]

  const uint32_t env = blockIdx.x / a.nb, blk = blockIdx.x % a.nb;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
#if CPMPPI_WAVE_PRIORITY
  // Small launches (one or two waves per SIMD) end with their slowest wave, and the all-gather of the previous step's result
  // runs UNDER this kernel on another stream (cpmppi_step_gather): a wave of that kernel sharing a SIMD with one of ours
  // takes issue slots from it for its whole duration.  Raised wave priority makes the arbiter serve the rollout wave first;
  // a lone rollout wave leaves more than half of the issue slots unused, so the guest still runs.
  if constexpr (VARIANT != 1 || LONE_WAVE) __builtin_amdgcn_s_setprio(3);
#endif
#ifdef CPMPPI_DEBUG_COUNTERS
  const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime();
  CPMPPI_DBG_STAMP(0);
#endif
  const uint32_t row0 = blk * (BLOCK * R) + wave * (64 * R);     // first rollout of this wave
  uint32_t n[R];
  bool valid[R];
#pragma unroll
  for (int i = 0; i < R; ++i) { n[i] = row0 + i * 64 + lane; valid[i] = n[i] < p.N; }
  const uint32_t H = p.H;
  const uint64_t step_offset = a.offset_dev ? (uint64_t)*a.offset_dev : a.offset;

  // ---- per-env, wave-uniform -------------------------------------------------------------------------------------
  // throughput build: the per-env constants come from the block fold_env_kernel wrote just before this launch (scalar loads
  // through the constant address space: the block is read-only for this kernel), everywhere else each wave forms them itself
  // (not default.py's cost fed with knots from memory, two rollouts per lane: with the block's 31 scalars live from the first
  // instruction that one instantiation runs out of SGPRs and spill lanes - 20 bytes of scratch; it keeps the in-kernel fold)
  constexpr bool ENV_FOLD = CPMPPI_ENV_FOLD != 0 && FAST && VARIANT_ == 1 && INTEG == PREDICTOR_ODE_V0 &&
                            !(COST == COST_DEFAULT && NOISE == NOISE_KNOTS && R == 2);
  typedef const __attribute__((address_space(4))) float* env_fold_ptr;
  // (the builds that fold in-kernel do so where they always did - further down for te, cos and the cost's constants: the latency
  // build's time moves by 3 % with the order of this prologue)
  EnvConst ec_;
  QbgmFolded qf_{};
  float cos0_ = 0.0f, inv_period_ = 0.0f, nearlim_ = 0.0f;
  if constexpr (ENV_FOLD) {
    env_fold_ptr ef = (env_fold_ptr)(uintptr_t)(a.env_fold + env);
#define CPMPPI_EF(field) ef[offsetof(EnvFold, field) / sizeof(float)]
    ec_.L = CPMPPI_EF(ec.L); ec_.Lh = CPMPPI_EF(ec.Lh); ec_.kp1 = CPMPPI_EF(ec.kp1); ec_.kp1_mt = CPMPPI_EF(ec.kp1_mt);
    ec_.mg = CPMPPI_EF(ec.mg); ec_.JinvLh = CPMPPI_EF(ec.JinvLh); ec_.kmLh = CPMPPI_EF(ec.kmLh); ec_.kM = CPMPPI_EF(ec.kM);
    ec_.g_i = CPMPPI_EF(ec.g_i); ec_.cT_i = CPMPPI_EF(ec.cT_i); ec_.inv_kLh = CPMPPI_EF(ec.inv_kLh);
    ec_.inv_halfL = CPMPPI_EF(ec.inv_halfL); ec_.uK_scale = CPMPPI_EF(ec.uK_scale); ec_.t1_i = CPMPPI_EF(ec.t1_i);
    ec_.tg_i = CPMPPI_EF(ec.tg_i); ec_.tcT_i = CPMPPI_EF(ec.tcT_i); ec_.tinv_kLh = CPMPPI_EF(ec.tinv_kLh); ec_.wlim = CPMPPI_EF(ec.wlim);
    qf_.c_dd = CPMPPI_EF(qf.c_dd); qf_.c_cc = CPMPPI_EF(qf.c_cc); qf_.neg_te = CPMPPI_EF(qf.neg_te);
    qf_.a_dd = CPMPPI_EF(qf.a_dd); qf_.a_ep = CPMPPI_EF(qf.a_ep); qf_.a_ekp = CPMPPI_EF(qf.a_ekp); qf_.a_db = CPMPPI_EF(qf.a_db);
    qf_.db_lim = CPMPPI_EF(qf.db_lim); qf_.a_u2 = CPMPPI_EF(qf.a_u2); qf_.k_a = CPMPPI_EF(qf.k_a);
    qf_.k_b_run = CPMPPI_EF(qf.k_b_run); qf_.k_b_nom = CPMPPI_EF(qf.k_b_nom); qf_.k_c_nom = CPMPPI_EF(qf.k_c_nom);
    cos0_ = CPMPPI_EF(cos0); inv_period_ = CPMPPI_EF(inv_period); nearlim_ = CPMPPI_EF(nearlim);
#undef CPMPPI_EF
  } else {
    const float L = a.L ? a.L[env] : p.L_default;
    ec_ = make_env_const_uniform(p, L);
  }
  const EnvConst ec = ec_;
  // Mid-size build (VARIANT 2 / 3, two rollouts per lane), phased horizon loop: quiet control steps and eventful ones - a
  // rollout of the wave ended the previous step at or beyond the track edge, or its pole spins beyond the rotation range -
  // run in SEPARATE loops over k (run_phased below).  The quiet loop is the throughput build's control step, untouched
  // (its substep loop handles the rare first event behind a branch); the eventful loop integrates with the event
  // arithmetic inline.  Kept apart like this, the quiet loop gets the registers and the layout of a kernel that has no
  // eventful code: section stamps at C4 (tools/dev/sections.py) showed the median wave of the throughput build at 2680
  // cycles per control step against 3240 for the build this replaced (three substeps at a time under a rollback, the
  // event loop as an alternative inside the same loop body), in EVERY section, identical source included - that build
  // paid for its event handling with a larger loop body (register copies, spill reloads), not with its triples.
  // Measured: C4 84 -> 78 us, C3 243 -> 235, 256 envs 121 -> 117, 1024 envs 379 -> 353 us; buffer-fed kernels alike
  // (C4 reference layout 87.5 -> 82.9 us).
  constexpr bool PHASED = FAST && VARIANT == 2 && R == 2;
  // throughput build, two rollouts per lane: ONE edge test per quiet control step, the step redone from its entry state on an
  // event (control_step_fast).  The entry state stays live through the step - 12 registers: within the 128 of four waves per
  // SIMD for quadratic_boundary_grad_minimal (113-122), beyond it for the other costs (130-157), which keep the per-substep test.
  // The phased mid-size build's quiet loop does the same (same cost only: the other costs' kernels grow by 10-25 registers,
  // past the 168 of three waves per SIMD) - there the compare -> scalar-branch hand-over a test costs a lone wave is paid three
  // times per control step instead of nine.
  constexpr bool ROLLBACK_TP = VARIANT == 1 && R == 2 && CPMPPI_SPIN_BRANCH != 0 && CPMPPI_ROLLBACK != 0 && COST == COST_QBGM;
  constexpr bool ROLLBACK = ROLLBACK_TP || (PHASED && CPMPPI_ROLLBACK != 0 && CPMPPI_ROLLBACK_PHASED != 0 && COST == COST_QBGM);
  const Params& ph = p;
  // (ROLLBACK kernels: three of the substep's wave-uniform constants are parked in vector registers - these kernels have twenty
  // to spare, while the scalar file is what they run out of: the Philox one was 20 bytes of scratch short)
  EnvConst eh_ = ec;
  if constexpr (ROLLBACK_TP) {
    asm volatile("v_mov_b32 %0, %1" : "=v"(eh_.kp1_mt) : "s"(ec.kp1_mt));
    asm volatile("v_mov_b32 %0, %1" : "=v"(eh_.mg) : "s"(ec.mg));
    asm volatile("v_mov_b32 %0, %1" : "=v"(eh_.inv_kLh) : "s"(ec.inv_kLh));
  }
  const EnvConst& eh = eh_;
  const float x_t = a.x_t[env], te = a.te[env];
  const float* __restrict__ s0 = a.s0 + (size_t)env * 6;
  const float* __restrict__ un = a.u_nom + (size_t)env * H;
  const float* __restrict__ up = (a.u_prev ? a.u_prev : a.u_nom) + (size_t)env * H;
  State<F> st{splat<F>(s0[0]), splat<F>(s0[1]), splat<F>(s0[2]), splat<F>(s0[3]), splat<F>(s0[4]), splat<F>(s0[5])};

  F cost = splat<F>(0.0f), corr = splat<F>(0.0f);
  float u_nom_sq = 0.0f;                     // QBGM_ACC with the correction on u_nom: sum of u_nom^2 over the stages (wave-uniform)
  F u_before = splat<F>(a.prev_in ? a.prev_in[env] : 0.0f);
  const bool qb_ccrc = COST == COST_DEFAULT && INTEG == PREDICTOR_ODE_V0 && p.qb_mode != 0u && a.prev_in != nullptr;   // quadratic_boundary.py:83-85
  F cosang = splat<F>(ENV_FOLD ? cos0_ : cosf(s0[0]));     // the cost plugins take cos(angle), not the stored angle_cos, at stage 0
  // `near` (wave-uniform): may any rollout of this wave sit at or beyond permissible_track_fraction * THL at the current
  // stage?  Only then does quadratic_boundary_grad_minimal's boundary term need evaluating (it is exactly zero below the
  // threshold).  The flag comes out of the previous control step's last substep, whose one pair of edge compares tests
  // against this coarser limit (substep_fast); stage 0 is the initial state all rollouts share.  Other costs: the limit
  // is the edge itself and the flag is unused.
  const QbgmFolded qf = ENV_FOLD ? qf_ : make_qbgm_folded(p, te);
  // quadratic_boundary_grad_minimal, FAST: stage cost and correction term accumulated term by term with FMAs (stage_qbgm_acc)
  constexpr bool QBGM_ACC = FAST && COST == COST_QBGM && CPMPPI_QBGM_FOLD != 0 && CPMPPI_QBGM_ACC != 0;
  // (not in the latency build: there the flag's compare -> scalar branch hand-over sits on the lone wave's critical path
  // once per control step - measured 56 -> 66 us for a single env - while the eight instructions it saves are hidden)
  constexpr bool TRACK_NEAR = FAST && COST == COST_QBGM && VARIANT != 0 && (INTEG == PREDICTOR_ODE_V0 || CPMPPI_ODE_TRACK_NEAR != 0);
  const float nearlim = (ENV_FOLD && TRACK_NEAR) ? nearlim_ : uniform_(TRACK_NEAR ? __builtin_fminf(p.w[6], 1.0f) * p.THL : p.THL);
  bool near = !TRACK_NEAR || !(__builtin_fabsf(s0[4]) < nearlim);

  // Latency build: the nominal control (and the legacy cost's previous sequence) of step k + 1 is requested while step k
  // integrates - a scalar load consumed a few instructions after its issue is ~100 ns of exposed latency per control step
  // for a wave that has its SIMD to itself (single env 60.5 -> 57.2 us; measured neutral at C4 and 8192 envs, +2 % at C3,
  // so the packed builds load it where it is used).
  // (measured, round 3: 8192 envs 2.61 -> 2.47 ms per launch; single env with knots from memory 57.6 -> 54.9 us, with
  // Philox / a delta_u buffer +0.5 / +1 % - those keep the one-step-ahead load; mid-size build: C4 -1..-3 %, C3 and 256 envs
  // +2 %, not enabled)
  constexpr bool NOMINAL_IN_LANES = FAST && ((((CPMPPI_NOMINAL_IN_LANES) >> VARIANT) & 1) != 0 || PHASED) && (VARIANT != 0 || NOISE == NOISE_KNOTS);
  constexpr bool PREFETCH_NOMINAL = (VARIANT == 0) && !NOMINAL_IN_LANES;
  float uk_next = PREFETCH_NOMINAL ? shifted_nominal(p, un, 0) : 0.0f;
  float up_next = (VARIANT == 0 && COST == COST_LEGACY) ? up[0] : 0.0f;
  // Round 3: the env's nominal sequence (after the configured shift) is held in ONE register, lane l holding stage 64 c + l of
  // the current chunk c of 64 stages, and a control step fetches its stage with v_readlane_b32: one vector load per 64
  // control steps instead of one per step.  (The sequence is written by this same launch's finalize, so the compiler may
  // not use scalar loads for it: it was a vector load plus s_waitcnt vmcnt(0) per control step.)
  float un_lane = 0.0f;
  if constexpr (NOMINAL_IN_LANES) {
    if (lane < H) un_lane = shifted_nominal(p, un, lane);
  }
#if defined(CPMPPI_DEBUG_COUNTERS) && defined(CPMPPI_SECTION_STAMPS)
  // sections: 0 two adjacent stamps (the stamp's own cost), 1 nominal + clamp + stage cost + correction, 2 rotation seed and
  // spin test, 3 intermediate substeps, 4 last substep, 5 between control steps (noise, interpolation, loop)
  unsigned sec[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, (unsigned)__builtin_amdgcn_s_memtime()};
  unsigned* const secp = sec;
#else
  unsigned* const secp = nullptr;
#endif
  // phased build: does a rollout of this wave sit at or beyond the track edge (or spin beyond the rotation range) as the
  // next control step starts?  (wave-uniform)
  bool at_edge = false;
  float run_hi_v = p.run_hi;
  if constexpr (FAST && VARIANT == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(run_hi_v) : "s"(p.run_hi));
  auto control_step = [&](uint32_t k, F du, auto eventful) __attribute__((always_inline)) {
    if (secp) { asm volatile("" : "+v"(du)); CPMPPI_SEC(secp, 5, st); CPMPPI_SEC(secp, 0, st); }
    float uk, upk = 0.0f;
    if constexpr (NOMINAL_IN_LANES) {
      if (__builtin_expect((k & 63u) == 0u && k != 0u, 0)) {
        const uint32_t kl = k + lane;
        un_lane = (kl < H) ? shifted_nominal(p, un, kl) : 0.0f;
      }
      uk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(un_lane), (int)(k & 63u)));
      if constexpr (COST == COST_LEGACY) {
        if constexpr (VARIANT == 0) {
          upk = up_next;
          if (k + 1 < H) up_next = up[k + 1];
        } else {
          upk = up[k];
        }
      }
    } else if constexpr (PREFETCH_NOMINAL) {
      uk = uk_next; upk = up_next;
      if (k + 1 < H) {
        uk_next = shifted_nominal(p, un, k + 1);
        if constexpr (COST == COST_LEGACY) up_next = up[k + 1];
      }
    } else {
      uk = shifted_nominal(p, un, k);
      if constexpr (COST == COST_LEGACY) upk = up[k];
    }
    F ur = splat<F>(uk) + du;
    if constexpr (FAST && VARIANT == 1) {
      // (v_med3_f32 takes ONE scalar operand: as plain kernel arguments the upper limit is copied into a vector register on
      // every control step; `run_hi_v` is that copy made once, behind an opaque asm so that it is not re-materialised)
#pragma unroll
      for (int i = 0; i < R; ++i) put(ur, i, __builtin_amdgcn_fmed3f(get(ur, i), p.run_lo, run_hi_v));
    } else {
      ur = clamp_(ur, p.run_lo, p.run_hi);
    }
    if constexpr (QBGM_ACC) {
      float b_nom = 0.0f;
      // (packed builds: the flag is re-formed from the kernel argument on every stage, behind an opaque copy - as a loop-invariant
      // bool the compiler keeps ONE lane mask for it and derives the negated one through a v_cndmask + v_cmp pair on every control
      // step; the latency build keeps the hoisted flag: there the three scalar instructions cost what vector ones do)
      uint32_t correction_u_now = p.correction_u;
      if constexpr (VARIANT != 0) asm volatile("" : "+s"(correction_u_now));
      const bool nom_mode = correction_u_now != CPMPPI_CORRECTION_U_RUN;
      if (__builtin_expect(nom_mode, 0)) {             // (wave-uniform; the correction takes u_nom: non-default glue)
        asm volatile("; correction term on u_nom");    // (keeps this a branch: if-converted it costs five instructions per stage)
        b_nom = uniform_(qf.k_b_nom * uk);
        u_nom_sq = __builtin_fmaf(uk, uk, u_nom_sq);
      }
      stage_qbgm_acc<F>(qf, st.x, cosang, st.w, ur, du, nom_mode, b_nom, x_t, near, cost, corr);
    } else if constexpr (COST == COST_QBGM) {
      cost += stage_qbgm<F, FAST>(p, st.x, cosang, st.w, ur, x_t, te, near, (FAST && CPMPPI_QBGM_FOLD != 0) ? &qf : nullptr);
      corr += mppi_correction<F>(p, p.correction_u == CPMPPI_CORRECTION_U_RUN ? ur : splat<F>(uk), du);
    } else if constexpr (COST == COST_DEFAULT) {
      cost += stage_default<F, FAST, (INTEG == PREDICTOR_ODE_V0)>(p, st.x, cosang, ur, x_t, te, u_before, qb_ccrc);
      corr += mppi_correction<F>(p, p.correction_u == CPMPPI_CORRECTION_U_RUN ? ur : splat<F>(uk), du);
      if (qb_ccrc) u_before = ur;               // (quadratic_boundary's control-change-rate term; wave-uniform)
    } else if constexpr (COST == COST_QBG) {
      cost += stage_qbg<F, FAST>(p, st.x, cosang, st.w, ur, u_before, x_t, te);
      corr += mppi_correction<F>(p, p.correction_u == CPMPPI_CORRECTION_U_RUN ? ur : splat<F>(uk), du);
      u_before = ur;
    } else {
      cost += stage_legacy<F, FAST>(p, st.x, cosang, st.w, st.v, uk, du, upk, x_t);
    }
    const F u = ur * splat<F>(p.u_max);     // Q2u, cartpole_equations.py:119-127
    if constexpr (INTEG == PREDICTOR_ODE) {
      if constexpr (FAST) {
        control_step_cromer_fast<F, (VARIANT == 0 || LONE_WAVE)>(st, ur * splat<F>(ec.uK_scale), p.S, p.t_step, p, ec);
        if constexpr (TRACK_NEAR) {             // (no edge test in this predictor to piggyback on: one pair of compares per control step)
          uint64_t m = 0;
#pragma unroll
          for (int i = 0; i < R; ++i) m |= __builtin_amdgcn_fcmpf(__builtin_fabsf(get(st.x, i)), nearlim, 11);   // unordered or >=
          near = m != 0;
        }
      } else {
        for (uint32_t sub = 0; sub < p.S; ++sub) substep_precise_cromer(st, u, p.t_step, p, ec);
      }
    } else if constexpr (FAST) {
      F uK = ur * splat<F>(ec.uK_scale);         // (k+1) u_max Q: the form in which the control enters positionDD's numerator
      if (secp) { asm volatile("" : "+v"(uK), "+v"(cost), "+v"(corr)); CPMPPI_SEC(secp, 1, st); }
      bool near_next;
      if constexpr (PHASED) {
        if constexpr (decltype(eventful)::value) near_next = control_step_fast_eventful<F, (LONE_WAVE && CPMPPI_EVENTFUL_UNROLL != 0)>(st, uK, p.S, p.t_step, ph, eh, nearlim, &at_edge);
        else near_next = control_step_fast<F, LONE_WAVE, false, ROLLBACK>(st, uK, p.S, p.t_step, ph, eh, nearlim, secp, &at_edge);
      } else {
        near_next = control_step_fast<F, false, (VARIANT == 1 && R == 2 && CPMPPI_SPIN_BRANCH != 0), ROLLBACK_TP>(st, uK, p.S, p.t_step, ph, eh, nearlim, secp, ROLLBACK_TP ? &at_edge : nullptr);
      }
      near = !TRACK_NEAR || near_next;
    } else {
      for (uint32_t sub = 0; sub < p.S; ++sub) substep_precise(st, u, p.t_step, p, ec);
    }
    cosang = st.c;
  };

  // ---- rollout over the horizon ----------------------------------------------------------------------------------
  // phased build: `step(k, eventful)` performs control step k (fetching its perturbation itself); quiet and eventful steps
  // in separate loops (see PHASED above)
  auto run_phased = [&](auto&& step) __attribute__((always_inline)) {
    uint32_t k = 0;
    while (k < H) {
      for (; k < H && !at_edge; ++k) step(k, std::false_type{});
      for (; k < H && at_edge; ++k) step(k, std::true_type{});
    }
  };
  if constexpr (NOISE == NOISE_DELTA_U) {
    // delta_u[E,N,H] in the REFERENCE's rollout-major layout (controller_mppi_cartpole.py:434-446,479-483: the tensor at
    // the optimizer / predictor seam).  A lane needs one row, a memory transaction wants neighbouring lanes on neighbouring
    // addresses: the transposition is done by the load itself.  global_load_lds_dwordx4 (gfx950) moves 16 bytes per lane
    // from a per-lane global address straight into LDS at (wave-uniform base) + 16 * lane, no vector registers in between:
    // lane l asks for columns [k0 + 4 part, +4) of ITS OWN row, so each of the tile's R * DTK/4 loads deposits one
    // "column piece" of 64 rows as 64 consecutive 16-byte slots, and at control step kk the lane reads word kk % 4 of its
    // slot in piece kk / 4.  Two tiles per wave: the next one streams in while the current one is integrated; the tile is
    // private to its wave, so one s_waitcnt vmcnt(0) orders load and use - no block barrier, no staging registers (round 2:
    // 16 predicated dword loads into 16 registers + 16 LDS stores + two block barriers per tile, 161 VGPRs).  Rows past N are
    // clamped to the env's last row (their lanes are masked out of every result); a horizon that is no multiple of the tile depth ends
    // with a tile that starts at H - DTK and overlaps its predecessor, so every load lies inside its row.
    constexpr uint32_t DTK = DMA_TK, NBUF = DMA_BUFS;
    static_assert(DTK % 4 == 0 && (NBUF == 1 || NBUF == 2), "tile depth: a multiple of 4; one or two tiles per wave");
    constexpr int PPR = DTK / 4;                      // 16-byte column pieces per tile
    constexpr uint32_t TILE_FLOATS = 64u * R * DTK;
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    float* const wave_tile = tile + wave_u * (NBUF * TILE_FLOATS);
    const float* __restrict__ env_src = a.noise + (size_t)env * p.N * H;     // wave-uniform
    uint32_t row_off[R];                                                       // floats from env_src to the lane's rows
#pragma unroll
    for (int i = 0; i < R; ++i) row_off[i] = (n[i] < p.N ? n[i] : p.N - 1u) * H;
    auto tile_start = [&](uint32_t t) __attribute__((always_inline)) -> uint32_t {
      const uint32_t k0 = t * DTK;
      return (k0 + DTK <= H) ? k0 : H - DTK;
    };
    auto dma = [&](uint32_t ks, uint32_t buf) __attribute__((always_inline)) {
#pragma unroll
      for (int part = 0; part < PPR; ++part)
#pragma unroll
        for (int i = 0; i < R; ++i)
          load16_to_lds(env_src + row_off[i] + ks + 4u * (uint32_t)part,
                        wave_tile + buf * TILE_FLOATS + (uint32_t)(part * R + i) * 256u);
    };
    const bool streamed = H >= DTK;          // (a horizon shorter than one tile is filled element by element, below)
    const uint32_t ntiles = streamed ? (H + DTK - 1u) / DTK : 1u;
    if (!streamed) {
      for (uint32_t k = 0; k < H; ++k)
#pragma unroll
        for (int i = 0; i < R; ++i)
          wave_tile[((k >> 2) * R + (uint32_t)i) * 256u + lane * 4u + (k & 3u)] = env_src[row_off[i] + k];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (NBUF == 2) {
      dma(0u, 0u);
    }
    if constexpr (PHASED) {
      // the same tiles, walked by control step instead of by nested tile / piece / word loops (the phased driver owns the
      // loop over k): tile t starts at step t * DTK; the quad is re-read every four columns of the tile; a last tile that
      // overlaps its predecessor is entered in its middle (o > 0)
      uint32_t t_next = 0u, ks = 0u;
      const float4* __restrict__ cur_tile = reinterpret_cast<const float4*>(wave_tile) + lane;
      float4 quad[R];
      auto read_quad = [&](uint32_t piece) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < R; ++i) quad[i] = cur_tile[(piece * R + (uint32_t)i) * 64u];
      };
      auto shift_quad = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < R; ++i) { quad[i].x = quad[i].y; quad[i].y = quad[i].z; quad[i].z = quad[i].w; }
      };
      run_phased([&](uint32_t k, auto eventful) __attribute__((always_inline)) {
        if (k == t_next * DTK) {
          const uint32_t t = t_next;
          ks = streamed ? tile_start(t) : 0u;
          if (streamed) {
            if (NBUF == 1) dma(ks, 0u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // tile t has landed in LDS
            if (NBUF == 2 && t + 1u < ntiles) dma(tile_start(t + 1u), (t + 1u) & 1u);
          }
          cur_tile = reinterpret_cast<const float4*>(wave_tile + (NBUF == 2 ? (t & 1u) : 0u) * TILE_FLOATS) + lane;
          t_next = t + 1u;
          const uint32_t o = k - ks;
          read_quad(o >> 2);
          for (uint32_t w = 0; w < (o & 3u); ++w) shift_quad();
        } else if (((k - ks) & 3u) == 0u) {
          read_quad((k - ks) >> 2);
        }
        F du;
#pragma unroll
        for (int i = 0; i < R; ++i) put(du, i, quad[i].x);
        shift_quad();
        control_step(k, du, eventful);
      });
    } else
    for (uint32_t t = 0; t < ntiles; ++t) {
      const uint32_t ks = streamed ? tile_start(t) : 0u, k_first = t * DTK;      // (k_first > ks only in an overlapping last tile)
      if (streamed) {
        if (NBUF == 1) dma(ks, 0u);                                             // (the other waves of the SIMD cover the wait)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // tile t has landed in LDS
        if (NBUF == 2 && t + 1u < ntiles) dma(tile_start(t + 1u), (t + 1u) & 1u);
      }
      const float4* __restrict__ cur_tile = reinterpret_cast<const float4*>(wave_tile + (NBUF == 2 ? (t & 1u) : 0u) * TILE_FLOATS) + lane;
      for (uint32_t q = 0; q < (uint32_t)PPR; ++q) {
        float4 quad[R];                                                         // one conflict-free 16-byte read per four control steps
#pragma unroll
        for (int i = 0; i < R; ++i) quad[i] = cur_tile[(q * R + (uint32_t)i) * 64u];
        for (uint32_t c = 0; c < 4u; ++c) {
          const uint32_t k = ks + 4u * q + c;
          F du;
#pragma unroll
          for (int i = 0; i < R; ++i) {
            put(du, i, quad[i].x);
            quad[i].x = quad[i].y; quad[i].y = quad[i].z; quad[i].z = quad[i].w;
          }
          if (k >= k_first && k < H) control_step(k, du, std::false_type{});
        }
      }
    }
  } else if constexpr (NOISE == NOISE_TILED) {
    // delta_u in the library's TILED layout [E][G = ceil(N/64)][Hq = ceil(H/4)][64 rows][4 steps] (cpmppi_sample_tiled /
    // cpmppi_tile_delta_u): lane l of row-group g reads ONE float4 per four control steps, and a wave-instruction reads
    // 1 KB of contiguous memory — every fetched byte is used, no LDS transpose.  The next quad is in flight while the
    // current one is integrated (four control steps = thousands of cycles of cover).
    const uint32_t G = (p.N + 63u) >> 6, Hq = (H + 3u) >> 2;
    const float4* __restrict__ src[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      uint32_t g = (row0 >> 6) + (uint32_t)i;
      g = g < G ? g : G - 1u;                                 // (rows of a group past the end are invalid anyway)
      src[i] = reinterpret_cast<const float4*>(a.noise) + ((size_t)env * G + g) * Hq * 64u + lane;
    }
    float4 cur[R], nxt[R];
#pragma unroll
    for (int i = 0; i < R; ++i) { cur[i] = src[i][0]; nxt[i] = cur[i]; }
    if constexpr (PHASED) {
      run_phased([&](uint32_t k, auto eventful) __attribute__((always_inline)) {
        if ((k & 3u) == 0u) {                                 // a new quad: the one requested four steps ago; request the next
          const uint32_t q = k >> 2;
          if (q != 0u) {
#pragma unroll
            for (int i = 0; i < R; ++i) cur[i] = nxt[i];
          }
          if (q + 1u < Hq) {
#pragma unroll
            for (int i = 0; i < R; ++i) nxt[i] = src[i][(size_t)(q + 1u) * 64u];
          }
        }
        F du;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          put(du, i, cur[i].x);
          cur[i].x = cur[i].y; cur[i].y = cur[i].z; cur[i].z = cur[i].w;
        }
        control_step(k, du, eventful);
      });
    } else
    for (uint32_t q = 0; q < Hq; ++q) {
      if (q + 1 < Hq) {
#pragma unroll
        for (int i = 0; i < R; ++i) nxt[i] = src[i][(size_t)(q + 1) * 64u];
      }
      const uint32_t kend = (H - 4u * q < 4u) ? (H - 4u * q) : 4u;
      for (uint32_t j = 0; j < kend; ++j) {
        F du;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          put(du, i, cur[i].x);
          // the quad moves down one step (three register moves) instead of a select on the wave-uniform j, which the
          // compiler turns into a tree of scalar branches per control step
          cur[i].x = cur[i].y; cur[i].y = cur[i].z; cur[i].z = cur[i].w;
        }
        control_step(4u * q + j, du, std::false_type{});
      }
#pragma unroll
      for (int i = 0; i < R; ++i) cur[i] = nxt[i];
    }
  } else {
    // Philox: one block yields FOUR consecutive knots (4q .. 4q+3); the other three are kept until needed.  Every knot is
    // also parked in LDS (when it fits) so that the soft-min reduction below does not generate the sequence again.
    float z_next[R][3];
    float* __restrict__ kstash = bsum + WAVES * a.W + tid;
    auto knot = [&](int i, uint32_t j) __attribute__((always_inline)) -> float {
      const uint32_t nn = valid[i] ? n[i] : 0;
      if constexpr (NOISE == NOISE_KNOTS) {
        return a.noise[((size_t)env * p.N + nn) * p.P + j];
      } else {
        float z;
        const uint32_t s = j & 3u;
        if (s == 0u) {
          float zq[4];
          philox_normal_quad(a.seed, step_offset, a.env_offset + env, nn, j >> 2, zq);
          z = p.sigma * zq[0];
          z_next[i][0] = p.sigma * zq[1]; z_next[i][1] = p.sigma * zq[2]; z_next[i][2] = p.sigma * zq[3];
        } else {
          z = (s == 1u) ? z_next[i][0] : ((s == 2u) ? z_next[i][1] : z_next[i][2]);
        }
        if (a.stash) kstash[(j * R + i) * BLOCK] = z;
        return z;
      }
    };
    constexpr bool F32_INTERP = FAST && NOISE == NOISE_PHILOX;       // our own noise: one FMA instead of the f64 form
    const float inv_period = ENV_FOLD ? inv_period_ : 1.0f / (float)p.period;
    float z_lo[R], z_hi[R], slope32[R];
    double slope[R];
    // knots from memory (the reference's own noise stream): the one after next is requested a whole knot period before it
    // is needed - a vector load consumed right after its issue is ~1 us of exposed latency per knot for a lone wave
    constexpr bool KNOT_AHEAD = (NOISE == NOISE_KNOTS);
    float z_ahead[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      z_lo[i] = knot(i, 0); z_hi[i] = knot(i, 1);
      z_ahead[i] = (KNOT_AHEAD && 2 < p.P) ? knot(i, 2) : 0.0f;
      if constexpr (F32_INTERP) slope32[i] = knot_slope32(z_lo[i], z_hi[i], inv_period);
      else slope[i] = knot_slope(z_lo[i], z_hi[i], p.period);
    }
    uint32_t ii = 0, j = 0;
    auto horizon_step = [&](uint32_t k, auto eventful) __attribute__((always_inline)) {
      F du;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        if constexpr (F32_INTERP) put(du, i, interp_from_slope32(slope32[i], z_lo[i], ii));
        else put(du, i, interp_from_slope(slope[i], z_lo[i], ii));
      }
      control_step(k, du, eventful);
      // (the branch weight is a LAYOUT hint: the nine of ten control steps that need no new knot fall through - 1024 envs
      // -2.9 %, 256 envs -2.4 %, C3 -1.5 %; FAST only: in one PRECISE kernel the other layout left a scratch slot)
      if (FAST ? __builtin_expect(++ii == p.period, 0) : (++ii == p.period)) {
        ii = 0; ++j;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          z_lo[i] = z_hi[i];
          if constexpr (KNOT_AHEAD) {
            if (j + 1 < p.P) z_hi[i] = z_ahead[i];
            if (j + 2 < p.P) z_ahead[i] = knot(i, j + 2);
          } else {
            if (j + 1 < p.P) z_hi[i] = knot(i, j + 1);
          }
          if constexpr (F32_INTERP) slope32[i] = knot_slope32(z_lo[i], z_hi[i], inv_period);
          else slope[i] = knot_slope(z_lo[i], z_hi[i], p.period);
        }
      }
    };
    if constexpr (PHASED) {
      run_phased(horizon_step);
    } else if constexpr (FAST && VARIANT == 1 && CPMPPI_KNOT_SEGMENTS != 0 && !(COST == COST_DEFAULT && NOISE == NOISE_PHILOX && R == 2)) {
      // throughput build: the horizon as NESTED loops - knot segments outside, the `period` control steps between two knots
      // inside, where the segment's knot and slope are loop invariants.  The flat loop above refreshes the knots behind a
      // branch inside the loop body, and the register allocator lines the hot path up with that branch's assignment by
      // shuffling (z_lo, z_hi, slope) through four v_mov_b64 on EVERY control step; here the refresh sits between two inner
      // loops.  Same knots in the same order, same interpolation (bit-identical).  (Not the `default`-cost Philox kernel with two
      // rollouts per lane: there this form costs two more scalar registers than the file has - a 20-byte scratch slot, which
      // tests/test_abi_and_host.py refuses.)
      uint32_t k = 0;
      for (uint32_t seg = 0; k < H; ++seg) {
        const uint32_t kend = (H - k < p.period) ? H : k + p.period;
        for (uint32_t i2 = 0; k < kend; ++k, ++i2) {
          F du;
#pragma unroll
          for (int i = 0; i < R; ++i) {
            if constexpr (F32_INTERP) put(du, i, interp_from_slope32(slope32[i], z_lo[i], i2));
            else put(du, i, interp_from_slope(slope[i], z_lo[i], i2));
          }
          control_step(k, du, std::false_type{});
        }
        if (k < H) {                                 // the next segment's knots (seg + 1, seg + 2)
          const uint32_t jn = seg + 1u;
#pragma unroll
          for (int i = 0; i < R; ++i) {
            z_lo[i] = z_hi[i];
            if constexpr (KNOT_AHEAD) {
              if (jn + 1 < p.P) z_hi[i] = z_ahead[i];
              if (jn + 2 < p.P) z_ahead[i] = knot(i, jn + 2);
            } else {
              if (jn + 1 < p.P) z_hi[i] = knot(i, jn + 1);
            }
            if constexpr (F32_INTERP) slope32[i] = knot_slope32(z_lo[i], z_hi[i], inv_period);
            else slope[i] = knot_slope(z_lo[i], z_hi[i], p.period);
          }
        }
      }
    } else {
      for (uint32_t k = 0; k < H; ++k) horizon_step(k, std::false_type{});
    }
  }

#ifdef CPMPPI_DEBUG_COUNTERS
  if (lane == 0 && blockIdx.x * WAVES + wave < 16384u)
    cpmppi::g_wave_cycles[blockIdx.x * WAVES + wave] = __builtin_amdgcn_s_memtime() - dbg_t0;
  CPMPPI_DBG_STAMP(1);
#ifdef CPMPPI_SECTION_STAMPS
  if (lane == 0 && blockIdx.x * WAVES + wave < 16384u) {
#pragma unroll
    for (int i = 0; i < 8; ++i) cpmppi::g_wave_sec[blockIdx.x * WAVES + wave][i] = sec[i];
  }
#endif
#endif
  // Everything the epilogue needs from the launch descriptor (output pointers, the partials workspace, the tickets) is read
  // from the kernarg segment HERE, behind an opaque copy of its address: as plain uses of `a` the compiler loads all of
  // them at kernel entry and keeps ~20 more scalar registers live through the horizon loop, which the packed builds pay
  // for with SGPR spills (v_writelane / v_readlane) inside the loop.
  const StepPtrs la = late_step_ptrs();
  // ---- per-rollout total cost ------------------------------------------------------------------------------------
  F S_total;
  if constexpr (COST == COST_LEGACY) {
    S_total = cost + terminal_indicator<F>(p, st.th, st.x, x_t);     // sum_k q + phi  (:197-199)
  } else {
    if constexpr (QBGM_ACC) {
      // (the two running sums of stage_qbgm_acc: the horizon aggregation's scale is in their weights; terminal cost zero)
      S_total = (cost + corr) + splat<F>(qf.k_c_nom * u_nom_sq);
    } else {
      const F term = (COST == COST_DEFAULT) ? terminal_indicator<F>(p, st.th, st.x, x_t) : splat<F>(0.0f);
      S_total = (p.horizon_reduce == CPMPPI_REDUCE_SUM) ? (cost + term) : (cost + term) / splat<F>((float)(H + 1));
      S_total += corr;
    }
  }
#pragma unroll
  for (int i = 0; i < R; ++i)
    if (la.S_out && valid[i]) la.S_out[(size_t)env * p.N + n[i]] = get(S_total, i);

  // ---- block-level soft-min partials (a16) -----------------------------------------------------------------------
  float m_l = INFINITY;
#pragma unroll
  for (int i = 0; i < R; ++i) m_l = fminf(m_l, valid[i] ? get(S_total, i) : INFINITY);
  const float m_w = wave_min(m_l);
  if (lane == 0) red[wave] = m_w;
  __syncthreads();
  float m_b = red[0];
#pragma unroll
  for (int w = 1; w < WAVES; ++w) m_b = fminf(m_b, red[w]);
  float e[R], e_l = 0.0f;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    e[i] = valid[i] ? expf((-1.0f / p.LBD) * (get(S_total, i) - m_b)) : 0.0f;
    e_l += e[i];
  }
  const float a_w = wave_sum(e_l);
  if (lane == 0) red[WAVES + wave] = a_w;

  const uint32_t W = la.W;
  float* __restrict__ my_bsum = bsum + wave * W;
  if constexpr (NOISE == NOISE_PHILOX) {
    const float* __restrict__ kstash = bsum + WAVES * W + tid;         // each lane reads back what it wrote itself
    for (uint32_t j = 0; j < W; ++j) {
      float v = 0.0f;
#pragma unroll
      for (int i = 0; i < R; ++i)
        v += e[i] * (la.stash ? kstash[(j * R + i) * BLOCK]
                             : philox_knot(a.seed, step_offset, a.env_offset + env, valid[i] ? n[i] : 0, j, p.sigma));
      v = wave_sum(v);
      if (lane == 0) my_bsum[j] = v;
    }
  } else if constexpr (NOISE == NOISE_TILED) {
    // second, coalesced sweep over the wave's quads: lane = row, the sum over the 64 rows of a group by wave reduction.
    // Four quads per batch with all their loads issued first: the sweep is a chain of load latencies otherwise (13 quads
    // at ~1 us each are 13 us of a 60 us single-env launch).
    const uint32_t G = (p.N + 63u) >> 6, Hq = (H + 3u) >> 2;
    const float4* __restrict__ src2[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      uint32_t g = (row0 >> 6) + (uint32_t)i;
      g = g < G ? g : G - 1u;
      src2[i] = reinterpret_cast<const float4*>(a.noise) + ((size_t)env * G + g) * Hq * 64u + lane;
    }
    constexpr int QB = 4;
#if CPMPPI_TILED_SCATTER
    // Round 4: the 16 column sums of a batch as a REDUCE-SCATTER over the wave instead of 16 full wave reductions.  Lane pairs at
    // distance 1, 2, 4, 8 each keep one half of their columns and hand the other half over (two selects + one add per column
    // pair: 8 + 4 + 2 + 1 pairs), after which a lane holds ONE column - number (lane & 15) of the batch - summed over its row of
    // 16 lanes; two more exchanges (distance 16, 32) add the four rows.  47 vector instructions per batch instead of 16 x 11;
    // the exchanges at distance >= 4 go through ds_swizzle / ds_bpermute (the LDS crossbar, not the vector ALU).
    const bool lb0 = (lane & 1u) != 0u, lb1 = (lane & 2u) != 0u, lb2 = (lane & 4u) != 0u, lb3 = (lane & 8u) != 0u;
    const int across = (int)((lane ^ 32u) << 2);
    auto swz = [](float x, auto pattern) __attribute__((always_inline)) {
      return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), decltype(pattern)::value));
    };
    for (uint32_t q0 = 0; q0 < Hq; q0 += QB) {
      float4 v[QB][R];
#pragma unroll
      for (int u = 0; u < QB; ++u) {
        const uint32_t q = (q0 + u < Hq) ? q0 + u : Hq - 1u;
#pragma unroll
        for (int i = 0; i < R; ++i) v[u][i] = src2[i][(size_t)q * 64u];
      }
      float c16[16];
#pragma unroll
      for (int u = 0; u < QB; ++u) {
        float4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < R; ++i) {
          acc.x = __builtin_fmaf(e[i], v[u][i].x, acc.x); acc.y = __builtin_fmaf(e[i], v[u][i].y, acc.y);
          acc.z = __builtin_fmaf(e[i], v[u][i].z, acc.z); acc.w = __builtin_fmaf(e[i], v[u][i].w, acc.w);
        }
        c16[4 * u + 0] = acc.x; c16[4 * u + 1] = acc.y; c16[4 * u + 2] = acc.z; c16[4 * u + 3] = acc.w;
      }
      float c8[8], c4[4], c2[2];
#pragma unroll
      for (int m = 0; m < 8; ++m)                 // distance 1: quad_perm [1,0,3,2]
        c8[m] = (lb0 ? c16[2 * m + 1] : c16[2 * m]) + dpp_<0xB1>(lb0 ? c16[2 * m] : c16[2 * m + 1]);
#pragma unroll
      for (int m = 0; m < 4; ++m)                 // distance 2: quad_perm [2,3,0,1]
        c4[m] = (lb1 ? c8[2 * m + 1] : c8[2 * m]) + dpp_<0x4E>(lb1 ? c8[2 * m] : c8[2 * m + 1]);
#pragma unroll
      for (int m = 0; m < 2; ++m)                 // distance 4: ds_swizzle, xor mask 4
        c4[m] = (lb2 ? c4[2 * m + 1] : c4[2 * m]) + swz(lb2 ? c4[2 * m] : c4[2 * m + 1], std::integral_constant<int, 0x101F>{});
      c2[0] = c4[0]; c2[1] = c4[1];
      float col = (lb3 ? c2[1] : c2[0]) + swz(lb3 ? c2[0] : c2[1], std::integral_constant<int, 0x201F>{});   // distance 8
      col += swz(col, std::integral_constant<int, 0x401F>{});                                                  // distance 16
      col += __int_as_float(__builtin_amdgcn_ds_bpermute(across, __float_as_int(col)));                        // distance 32
      const uint32_t k = 4u * q0 + (lane & 15u);   // the column this lane ended up with
      if (lane < 16u && k < W) my_bsum[k] = col;
    }
#else
    for (uint32_t q0 = 0; q0 < Hq; q0 += QB) {
      float4 v[QB][R];
#pragma unroll
      for (int u = 0; u < QB; ++u) {
        const uint32_t q = (q0 + u < Hq) ? q0 + u : Hq - 1u;
#pragma unroll
        for (int i = 0; i < R; ++i) v[u][i] = src2[i][(size_t)q * 64u];
      }
#pragma unroll
      for (int u = 0; u < QB; ++u) {
        float4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < R; ++i) {
          acc.x = __builtin_fmaf(e[i], v[u][i].x, acc.x); acc.y = __builtin_fmaf(e[i], v[u][i].y, acc.y);
          acc.z = __builtin_fmaf(e[i], v[u][i].z, acc.z); acc.w = __builtin_fmaf(e[i], v[u][i].w, acc.w);
        }
        acc.x = wave_sum(acc.x); acc.y = wave_sum(acc.y); acc.z = wave_sum(acc.z); acc.w = wave_sum(acc.w);
        const uint32_t k = 4u * (q0 + u);
        if (lane == 0 && q0 + u < Hq) {
          my_bsum[k] = acc.x;
          if (k + 1 < W) my_bsum[k + 1] = acc.y;
          if (k + 2 < W) my_bsum[k + 2] = acc.z;
          if (k + 3 < W) my_bsum[k + 3] = acc.w;
        }
      }
    }
#endif
  } else {
    // transposed pass: lane = column (time-step or knot), loop over the wave's rows, rows read coalesced (cache-hot)
    const float* __restrict__ src = a.noise + ((size_t)env * p.N + row0) * W;
    for (uint32_t c0 = 0; c0 < W; c0 += 64) {
      const uint32_t col = c0 + lane;
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const uint32_t base = row0 + i * 64;
        const uint32_t rows = (base < p.N) ? ((p.N - base < 64u) ? p.N - base : 64u) : 0u;
        // eight independent row loads in flight per batch: the pass is bound by load latency, not by its arithmetic
        const float* __restrict__ colp = src + (size_t)(i * 64) * W + (col < W ? col : 0u);
        uint32_t r = 0;
        for (; r + 8 <= rows; r += 8) {
          float x[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) x[u] = colp[(size_t)(r + u) * W];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            acc = __builtin_fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(e[i]), r + u)), x[u], acc);
        }
        for (; r < rows; ++r)
          acc = __builtin_fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(e[i]), r)), colp[(size_t)r * W], acc);
      }
      if (col < W) my_bsum[col] = acc;
    }
  }
  __syncthreads();
  float* __restrict__ out = la.partial + ((size_t)env * la.nb + blk) * (2 + W);
  if (tid == 0) {
    float a_b = red[WAVES];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) a_b += red[WAVES + w];
    out[0] = m_b;
    out[1] = a_b;
  }
  for (uint32_t c = tid; c < W; c += BLOCK) {
    float v = bsum[c];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) v += bsum[w * W + c];
    out[2 + c] = v;
  }
  CPMPPI_DBG_STAMP(2);
  // ---- fused finalize: the env's last-arriving block merges the partials (no second launch) -----------------------
  // Placement-independent hand-off (cdna_hip_programming.md Guideline 16): every storing wave drains its stores, the
  // block's barrier, one lane's agent-scope release, then the ticket; the consumer block does one agent-scope acquire
  // (invalidates this CU's L1), drains, barriers, and additionally reads the partials with sc1 loads.
  if (la.counter) {
    __shared__ uint32_t ticket;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ticket = __hip_atomic_fetch_add(la.counter + env, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (ticket == la.nb - 1) {                               // uniform over the block
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(la.counter + env, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      }
      __syncthreads();
      finalize_env<(NOISE == NOISE_KNOTS || NOISE == NOISE_PHILOX), true>(p, la.partial, la.nb, W, la.u_nom, la.u_nom_out, la.Q_out, env, la.host_ticket, la.gs);
    }
  }
  CPMPPI_DBG_STAMP(3);
}

}  // namespace cpmppi_k

// Every instantiation of rollout_cost_kernel, by the translation unit that compiles it.  X(COST, FAST, NOISE, R, VARIANT).
// The units define them with CPMPPI_DEFINE_ROLLOUT, cpmppi.hip declares ALL of them extern with CPMPPI_DECLARE_ROLLOUT —
// an instantiation missing there would be compiled a second time in cpmppi.hip with that unit's flags, and the runtime
// would launch whichever copy registered last.
#define CPMPPI_FOR_COSTS(X, FAST, NOISE, R, V) \
  X(COST_QBGM, FAST, NOISE, R, V) X(COST_DEFAULT, FAST, NOISE, R, V) X(COST_LEGACY, FAST, NOISE, R, V) X(COST_QBG, FAST, NOISE, R, V)
#define CPMPPI_FOR_NOISES(X, FAST, R, V)                                                        \
  CPMPPI_FOR_COSTS(X, FAST, NOISE_DELTA_U, R, V) CPMPPI_FOR_COSTS(X, FAST, NOISE_KNOTS, R, V)   \
  CPMPPI_FOR_COSTS(X, FAST, NOISE_PHILOX, R, V) CPMPPI_FOR_COSTS(X, FAST, NOISE_TILED, R, V)
// (the latency build is two units: the reference-layout buffer kernel schedules best with iterative-ilp, the others with
// max-memory-clause - single env 1024 x 50 on MI355X: Philox 59.2 vs 56.2 us, knots 59.6 vs 58.2, buffer 61.8 vs 68.0)
#define CPMPPI_LATENCY_INSTANCES(X)                                                                                     \
  CPMPPI_FOR_COSTS(X, true, NOISE_KNOTS, 1, 0) CPMPPI_FOR_COSTS(X, true, NOISE_TILED, 1, 0)                              \
  X(COST_QBGM, true, NOISE_PHILOX, 1, 0) X(COST_LEGACY, true, NOISE_PHILOX, 1, 0) X(COST_QBG, true, NOISE_PHILOX, 1, 0)
// (... and the `default`-cost Philox kernel, which max-memory-clause leaves with a 20-byte scratch slot for two spilled
// scalar registers - tests/test_abi_and_host.py keeps scratch out of every instantiation)
#define CPMPPI_LATENCY_BUFFER_INSTANCES(X) CPMPPI_FOR_COSTS(X, true, NOISE_DELTA_U, 1, 0) X(COST_DEFAULT, true, NOISE_PHILOX, 1, 0)
// (the mid-size build is two units as well, for compile time: 32 kernels each; VARIANT 3 = the build for launches of at
// most one wave per SIMD)
#define CPMPPI_MID_INSTANCES(X) CPMPPI_FOR_COSTS(X, true, NOISE_KNOTS, 2, 2) CPMPPI_FOR_COSTS(X, true, NOISE_PHILOX, 2, 2) \
  CPMPPI_FOR_COSTS(X, true, NOISE_KNOTS, 2, 3) CPMPPI_FOR_COSTS(X, true, NOISE_PHILOX, 2, 3)
#define CPMPPI_MID_BUFFER_INSTANCES(X) CPMPPI_FOR_COSTS(X, true, NOISE_DELTA_U, 2, 2) CPMPPI_FOR_COSTS(X, true, NOISE_TILED, 2, 2) \
  CPMPPI_FOR_COSTS(X, true, NOISE_DELTA_U, 2, 3) CPMPPI_FOR_COSTS(X, true, NOISE_TILED, 2, 3)
#define CPMPPI_THROUGHPUT_INSTANCES(X) \
  CPMPPI_FOR_NOISES(X, true, 1, 1) CPMPPI_FOR_NOISES(X, false, 1, 1) CPMPPI_FOR_NOISES(X, true, 2, 1)
// predictor_ODE (INTEG = PREDICTOR_ODE): latency build (one rollout per lane) and throughput build (both lane mappings, PRECISE)
#define CPMPPI_ODE_LATENCY_INSTANCES(X) CPMPPI_FOR_NOISES(X, true, 1, 0)
#define CPMPPI_ODE_THROUGHPUT_INSTANCES(X) \
  CPMPPI_FOR_NOISES(X, true, 1, 1) CPMPPI_FOR_NOISES(X, false, 1, 1) CPMPPI_FOR_NOISES(X, true, 2, 1)
// (two rollouts per lane in a launch of at most one wave per SIMD: the substeps as straight-line code, raised wave priority)
#define CPMPPI_ODE_LONE_INSTANCES(X) CPMPPI_FOR_NOISES(X, true, 2, 3)
#define CPMPPI_DEFINE_ROLLOUT_ODE(COST, FAST, NOISE, R, V) \
  template __global__ void rollout_cost_kernel<COST, FAST, NOISE, R, V, PREDICTOR_ODE>(const Params, const StepPtrs);
#define CPMPPI_DECLARE_ROLLOUT_ODE(COST, FAST, NOISE, R, V) \
  extern template __global__ void rollout_cost_kernel<COST, FAST, NOISE, R, V, PREDICTOR_ODE>(const Params, const StepPtrs);
#define CPMPPI_DEFINE_ROLLOUT(COST, FAST, NOISE, R, V) \
  template __global__ void rollout_cost_kernel<COST, FAST, NOISE, R, V>(const Params, const StepPtrs);
#define CPMPPI_DECLARE_ROLLOUT(COST, FAST, NOISE, R, V) \
  extern template __global__ void rollout_cost_kernel<COST, FAST, NOISE, R, V>(const Params, const StepPtrs);
