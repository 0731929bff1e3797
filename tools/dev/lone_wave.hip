// Micro-benchmark (development tool): what the rollout kernel's own control-step code costs a wave as a function of the
// waves that share its SIMD - the pieces of `control_step_fast<f2>` (cpmppi_device.hpp) and of the triples-with-rollback
// control step it replaced (rows "triple ...") timed in isolation, with the
// product's compiler flags, on 1 / 2 / 4 waves per SIMD.  Answers: is a lone wave (BASELINE C4: one packed wave per SIMD)
// bound by issue (~5 cycles per packed instruction), by dependent latency (~9), or by something else?
// What it told (round 3): the triple of the mid-size build runs at the lone wave's ISSUE limit (184 ns for 81 instructions =
// 2.27 ns each, the rate of independent v_pk_fma_f32), so no scheduling change can speed it up, and two independent triples
// interleaved are no faster per substep.  What it could NOT tell: the cost of the event test - variants predicted here to
// save 25-45 % of a control step (compare issued before the third substep and branched on after it; triples for the
// one-rollout-per-lane mapping) measured 0 to -3 % in the real kernel (tools/kbench.py), where registers, code layout
// and the code around the substeps differ.  Trust the A/B of the real kernel, not these rows, for anything but the
// issue-rate facts.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I include -I cartpolesimulation_amd/csrc -mllvm -amdgpu-sched-strategy=iterative-ilp
//         -mllvm -disable-vector-combine tools/dev/lone_wave.hip -o build_variants/lone_wave
#include "cpmppi_device.hpp"
#include <stdio.h>
#include <string.h>
#include <math.h>
using namespace cpmppi;

#define TO_VGPR(x) asm volatile("" : "+v"(x))

struct Env {
  Params ph;
  EnvConst eh;
};
__device__ __forceinline__ void make_env(const Params& p, Params& ph, EnvConst& eh, bool vgpr) {
  ph = p;
  eh = make_env_const_uniform(p, p.L_default);
  if (vgpr) {
    TO_VGPR(ph.m_pole); TO_VGPR(eh.kp1_mt); TO_VGPR(eh.mg); TO_VGPR(eh.JinvLh);
    TO_VGPR(eh.kmLh); TO_VGPR(eh.kM); TO_VGPR(eh.t1_i); TO_VGPR(eh.inv_kLh);
  }
}

__device__ __forceinline__ State<f2> load_state(const float* in, int o) {
  const int l = threadIdx.x & 63;
  const float a = in[l] * 1e-3f + 0.01f * o;
  State<f2> st;
  st.th = f2{0.3f + a, -0.2f - a};
  st.w = f2{0.5f + a, -0.7f + a};
  st.c = f2{cosf(st.th.x), cosf(st.th.y)};
  st.s = f2{sinf(st.th.x), sinf(st.th.y)};
  st.x = f2{0.01f + a, -0.02f - a};
  st.v = f2{0.1f * a, -0.1f * a};
  return st;
}
__device__ __forceinline__ void sink(float* out, const State<f2>& st) {
  const float z = st.th.x + st.th.y + st.w.x + st.w.y + st.c.x + st.c.y + st.s.x + st.s.y + st.x.x + st.x.y + st.v.x + st.v.y;
  if (z == 123.456f) out[0] = z;
}

// KIND 0: whole control steps (seed the rotation pair, nine substeps each with its test, last substep with wrap + sincos)
// KIND 1: triples only (the loop of the mid-size build: 3 substeps without event handling + v_max3 test)
// KIND 2: two independent triples interleaved in one wave (what more ILP would buy)
// KIND 3: last substep only (substep_fast: wrap + polynomial sincos + near test)
// KIND 4: triples without the test (counted loop)
// KIND 5: single substeps with the per-substep test behind a wave-uniform branch (BOUNCY = false)
template <int KIND, bool VG>
__global__ __launch_bounds__(256) void k(const float* in, float* out, Params p, int iters) {
  Params ph; EnvConst eh;
  make_env(p, ph, eh, VG);
  State<f2> st = load_state(in, 0), st2 = load_state(in, 1);
  const float t = p.t_step;
  f2 uK = f2{0.3f, -0.2f} * eh.uK_scale;
  f2 xlim = splat<f2>(p.THL);
  f2 cd, sd, cd2, sd2;
  rot_pair<f2>(st.w * splat<f2>(t), cd, sd);
  rot_pair<f2>(st2.w * splat<f2>(t), cd2, sd2);
  const float nearlim = uniform_(0.85f * p.THL);
  int it = 0;
  for (; it < iters; ++it) {
    if constexpr (KIND == 0) {
      control_step_fast<f2>(st, uK, p.S, t, ph, eh, nearlim);
      uK = -uK;
    } else if constexpr (KIND == 1 || KIND == 4) {
      substep_fast_rot_carried<f2, false, true>(st, uK, t, ph, eh, cd, sd, xlim, false);
      const f2 xa = st.x;
      substep_fast_rot_carried<f2, false, true>(st, uK, t, ph, eh, cd, sd, xlim, false);
      const f2 xb = st.x;
      substep_fast_rot_carried<f2, false, true>(st, uK, t, ph, eh, cd, sd, xlim, false);
      if (KIND == 1) {
        uint64_t fired = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float m = __builtin_fmaxf(__builtin_fabsf(get(xa, i)), __builtin_fmaxf(__builtin_fabsf(get(xb, i)), __builtin_fabsf(get(st.x, i))));
          fired |= __builtin_amdgcn_fcmpf(m, get(xlim, i), 3);
        }
        if (__builtin_expect(fired != 0, 0)) break;
      }
    } else if constexpr (KIND == 2) {
      substep_fast_rot_carried<f2, false, true>(st, uK, t, ph, eh, cd, sd, xlim, false);
      substep_fast_rot_carried<f2, false, true>(st2, uK, t, ph, eh, cd2, sd2, xlim, false);
      const f2 xa = st.x, xa2 = st2.x;
      substep_fast_rot_carried<f2, false, true>(st, uK, t, ph, eh, cd, sd, xlim, false);
      substep_fast_rot_carried<f2, false, true>(st2, uK, t, ph, eh, cd2, sd2, xlim, false);
      const f2 xb = st.x, xb2 = st2.x;
      substep_fast_rot_carried<f2, false, true>(st, uK, t, ph, eh, cd, sd, xlim, false);
      substep_fast_rot_carried<f2, false, true>(st2, uK, t, ph, eh, cd2, sd2, xlim, false);
      uint64_t fired = 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float m = __builtin_fmaxf(__builtin_fabsf(get(xa, i)), __builtin_fmaxf(__builtin_fabsf(get(xb, i)), __builtin_fabsf(get(st.x, i))));
        const float m2 = __builtin_fmaxf(__builtin_fabsf(get(xa2, i)), __builtin_fmaxf(__builtin_fabsf(get(xb2, i)), __builtin_fabsf(get(st2.x, i))));
        fired |= __builtin_amdgcn_fcmpf(__builtin_fmaxf(m, m2), get(xlim, i), 3);
      }
      if (__builtin_expect(fired != 0, 0)) break;
    } else if constexpr (KIND == 3) {
      substep_fast<f2>(st, uK, t, ph, eh, nearlim, true);
      uK = -uK;
    } else if constexpr (KIND == 5) {
      if (substep_fast_rot_carried<f2, false>(st, uK, t, ph, eh, cd, sd, xlim, true) != 0) uK = -uK;
    }
  }
  sink(out, st); sink(out, st2);
  if (it == -1) out[1] = cd.x + sd.y + cd2.x + sd2.y;
}

// The intermediate substep of the one-rollout-per-lane mapping (ode_euler_fast + rot_pair_lo + rotation, no event test) as
// ONE asm block in a hand-chosen order: the nine links of the loop-carried chain (w -> t1 -> numerator -> xDD -> aDD -> w')
// are spaced two or three independent instructions apart.  Same operations as the compiler's code (bit-identical results).
struct AsmConsts { float m, K, JLh, mg, kmLh, kM, t1i, ikLh, c5, c3, c4; };
__device__ __forceinline__ void substep_asm(State<float>& st, float uK, float t, const AsmConsts& k) {
  float twj, tcm, ww, d, t1, A, wk, d2, n, r, tt1, pc, ps, d3, cd, sd, xDD, ssd, csd, xc, aDD;
  asm volatile(
      "v_mul_f32 %[twj], %[w], %[JLh]\n"
      "v_mul_f32 %[tcm], %[c], %[m]\n"
      "v_mul_f32 %[ww], %[w], %[w]\n"
      "v_mul_f32 %[d], %[w], %[t]\n"
      "v_fma_f32 %[t1], %[mg], %[s], -%[twj]\n"
      "v_fma_f32 %[A], -%[tcm], %[c], %[K]\n"
      "v_mul_f32 %[wk], %[ww], %[kmLh]\n"
      "v_mul_f32 %[d2], %[d], %[d]\n"
      "v_fma_f32 %[n], %[c], %[t1], %[uK]\n"
      "v_rcp_f32 %[r], %[A]\n"
      "v_mul_f32 %[tt1], %[t1], %[t1i]\n"
      "v_fma_f32 %[pc], %[d2], %[c4], -0.5\n"
      "v_fma_f32 %[n], -%[wk], %[s], %[n]\n"
      "v_fma_f32 %[ps], %[d2], %[c5], %[c3]\n"
      "v_mul_f32 %[d3], %[d], %[d2]\n"
      "v_fma_f32 %[th], %[w], %[t], %[th]\n"
      "v_fma_f32 %[n], -%[kM], %[v], %[n]\n"
      "v_fma_f32 %[x], %[v], %[t], %[x]\n"
      "v_fma_f32 %[cd], %[d2], %[pc], 1.0\n"
      "v_fma_f32 %[sd], %[d3], %[ps], %[d]\n"
      "v_mul_f32 %[xDD], %[n], %[r]\n"
      "v_mul_f32 %[ssd], %[s], %[sd]\n"
      "v_mul_f32 %[csd], %[c], %[sd]\n"
      "v_mul_f32 %[xc], %[xDD], %[c]\n"
      "v_fma_f32 %[v], %[xDD], %[t], %[v]\n"
      "v_fma_f32 %[c], %[c], %[cd], -%[ssd]\n"
      "v_fma_f32 %[s], %[s], %[cd], %[csd]\n"
      "v_fma_f32 %[aDD], %[xc], %[ikLh], %[tt1]\n"
      "v_fma_f32 %[w], %[aDD], %[t], %[w]\n"
      : [th] "+v"(st.th), [w] "+v"(st.w), [c] "+v"(st.c), [s] "+v"(st.s), [x] "+v"(st.x), [v] "+v"(st.v),
        [twj] "=&v"(twj), [tcm] "=&v"(tcm), [ww] "=&v"(ww), [d] "=&v"(d), [t1] "=&v"(t1), [A] "=&v"(A), [wk] "=&v"(wk),
        [d2] "=&v"(d2), [n] "=&v"(n), [r] "=&v"(r), [tt1] "=&v"(tt1), [pc] "=&v"(pc), [ps] "=&v"(ps), [d3] "=&v"(d3),
        [cd] "=&v"(cd), [sd] "=&v"(sd), [xDD] "=&v"(xDD), [ssd] "=&v"(ssd), [csd] "=&v"(csd), [xc] "=&v"(xc), [aDD] "=&v"(aDD)
      : [uK] "v"(uK), [t] "s"(t), [m] "v"(k.m), [K] "v"(k.K), [JLh] "v"(k.JLh), [mg] "v"(k.mg), [kmLh] "v"(k.kmLh),
        [kM] "v"(k.kM), [t1i] "v"(k.t1i), [ikLh] "v"(k.ikLh), [c5] "v"(k.c5), [c3] "v"(k.c3), [c4] "v"(k.c4));
}

// One rollout per lane (the latency build's mapping): KIND 0 whole control steps, 1 intermediate substeps (rotation by
// polynomial, per-substep test behind a wave-uniform branch), 2 the same without test, 3 the last substep
template <int KIND>
__global__ __launch_bounds__(256) void k1(const float* in, float* out, Params p, int iters) {
  Params ph = p;
  EnvConst eh = make_env_const_uniform(p, p.L_default);
  const State<f2> s2 = load_state(in, 0);
  State<float> st{s2.th.x, s2.w.x, s2.c.x, s2.s.x, s2.x.x, s2.v.x};
  const float t = p.t_step;
  float uK = 0.3f * eh.uK_scale;
  const float nearlim = uniform_(p.THL);
  AsmConsts ak{ph.m_pole, eh.kp1_mt, eh.JinvLh, eh.mg, eh.kmLh, eh.kM, eh.t1_i, eh.inv_kLh, 8.3333333e-3f, -1.6666667e-1f, 4.1666667e-2f};
  TO_VGPR(ak.m); TO_VGPR(ak.K); TO_VGPR(ak.JLh); TO_VGPR(ak.mg); TO_VGPR(ak.kmLh); TO_VGPR(ak.kM); TO_VGPR(ak.t1i); TO_VGPR(ak.ikLh);
  TO_VGPR(ak.c5); TO_VGPR(ak.c3); TO_VGPR(ak.c4);
  const float ts = uniform_(t);
  if constexpr (KIND == 6) {            // self-check: the asm substep against the compiler's, bit for bit
    State<float> sa = st, sb = st;
    unsigned bad = 0;
    for (int i = 0; i < 1000; ++i) {
      substep_asm(sa, uK, ts, ak);
      substep_fast_rot<float, false, 0>(sb, uK, t, ph, eh);
      bad |= (__float_as_uint(sa.th) ^ __float_as_uint(sb.th)) | (__float_as_uint(sa.w) ^ __float_as_uint(sb.w)) | (__float_as_uint(sa.c) ^ __float_as_uint(sb.c)) |
             (__float_as_uint(sa.s) ^ __float_as_uint(sb.s)) | (__float_as_uint(sa.x) ^ __float_as_uint(sb.x)) | (__float_as_uint(sa.v) ^ __float_as_uint(sb.v));
      if ((i & 63) == 63) uK = -uK;
    }
    if (bad != 0u) atomicAdd(reinterpret_cast<unsigned*>(out) + 2, 1u);
    st = sa;
  }
  int it = 0;
  for (; it < ((KIND == 6) ? 0 : iters); ++it) {
    if constexpr (KIND == 5) {
      substep_asm(st, uK, ts, ak);
    } else if constexpr (KIND == 0) {
      control_step_fast<float>(st, uK, p.S, t, ph, eh, nearlim);
      uK = -uK;
    } else if constexpr (KIND == 1) {
      substep_fast_rot<float>(st, uK, t, ph, eh);
    } else if constexpr (KIND == 2) {
      substep_fast_rot<float, false, 0>(st, uK, t, ph, eh);
    } else if constexpr (KIND == 4) {
      // three substeps without event handling, ONE test: max3 of the positions against the edge, max3 of the rotation
      // angles against the polynomial's range
      const float d0 = st.w * t;
      substep_fast_rot<float, true, 0>(st, uK, t, ph, eh);
      const float xa = st.x, d1 = st.w * t;
      substep_fast_rot<float, true, 0>(st, uK, t, ph, eh);
      const float xb = st.x, d2 = st.w * t;
      substep_fast_rot<float, true, 0>(st, uK, t, ph, eh);
      const float mx = __builtin_fmaxf(__builtin_fabsf(xa), __builtin_fmaxf(__builtin_fabsf(xb), __builtin_fabsf(st.x)));
      const float md = __builtin_fmaxf(__builtin_fabsf(d0), __builtin_fmaxf(__builtin_fabsf(d1), __builtin_fabsf(d2)));
      const uint64_t fired = __builtin_amdgcn_fcmpf(mx, ph.THL, 3) | __builtin_amdgcn_fcmpf(md, ROT_LIMIT_LO, 2);
      if (__builtin_expect(fired != 0, 0)) break;
    } else {
      substep_fast<float, true, false>(st, uK, t, ph, eh, nearlim);
      uK = -uK;
    }
  }
  const float z = st.th + st.w + st.c + st.s + st.x + st.v;
  if (z == 123.456f || it == -1) out[0] = z;
}
template <int KIND>
float run1(const float* in, float* out, const Params& p, int iters, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k1<KIND>), dim3(blocks), dim3(256), 0, 0, in, out, p, 10);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k1<KIND>), dim3(blocks), dim3(256), 0, 0, in, out, p, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = fminf(best, ms);
  }
  return best * 1e6f / iters;
}

// v_pk_fma_f32 / v_pk_mul_f32 with every operand a VGPR pair (the rollout kernel's form), ILP independent chains per lane
template <int ILP, int MODE>
__global__ __launch_bounds__(256) void kv(const float* in, float* out, int iters) {
  const int l = threadIdx.x & 63;
  f2 y[ILP], a = f2{1.0001f + in[l] * 1e-6f, 0.9999f}, b = f2{0.5f, in[l] * 1e-3f};
#pragma unroll
  for (int i = 0; i < ILP; ++i) y[i] = f2{in[l] + i, in[l] + 1.0f + i};
  TO_VGPR(a); TO_VGPR(b);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (MODE == 0) y[i] = __builtin_elementwise_fma(y[i], a, b);        // 3 VGPR pairs
        else if (MODE == 1) y[i] = y[i] * a;                               // v_pk_mul, 2 VGPR pairs
        else if (MODE == 2) y[i] = __builtin_elementwise_fma(y[i], y[i], b);   // same pair twice
        else { y[i].x = __builtin_fmaf(y[i].x, a.x, b.x); y[i].y = __builtin_fmaf(y[i].y, a.y, b.y); }   // two plain FMAs (needs -fno-slp-vectorize to stay plain)
      }
    }
  }
  float z = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) z += y[i].x + y[i].y;
  if (z == 123.456f) out[0] = z;
}
template <int ILP, int MODE>
float runv(const float* in, float* out, int iters, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kv<ILP, MODE>), dim3(blocks), dim3(256), 0, 0, in, out, 10);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kv<ILP, MODE>), dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = fminf(best, ms);
  }
  return best * 1e6f / ((float)iters * 16.0f * ILP * (MODE == 3 ? 2 : 1));   // ns per wave instruction
}

template <int KIND, bool VG>
float run(const float* in, float* out, const Params& p, int iters, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, VG>), dim3(blocks), dim3(256), 0, 0, in, out, p, 10);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, VG>), dim3(blocks), dim3(256), 0, 0, in, out, p, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = fminf(best, ms);
  }
  return best * 1e6f / iters;     // ns per iteration
}

int main() {
  float *in, *out;
  hipMalloc(&in, 1024); hipMalloc(&out, 1024);
  float h[64]; for (int i = 0; i < 64; ++i) h[i] = (float)i;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  Params p; memset(&p, 0, sizeof(p));
  p.E = 1; p.N = 1024; p.H = 50; p.S = 10; p.P = 6; p.period = 10; p.t_step = 0.002f;
  p.k = 1.0f / 3.0f; p.m_cart = 0.23f; p.m_pole = 0.087f; p.g = 9.81f; p.J_fric = 5e-5f; p.M_fric = 3.22f; p.u_max = 1.77f;
  p.THL = 1.0e6f;                  // no rollout reaches the edge: the clean path
  p.L_default = 0.395f;
  p.w[6] = 0.85f;
  p.run_lo = -1.0f; p.run_hi = 1.0f; p.lo = -1.0f; p.hi = 1.0f;
  const int iters = 20000;
  printf("ns per loop iteration (min of 5 launches of %d iterations); per substep in brackets\n", iters);
  printf("%-44s %10s %10s %10s\n", "waves per SIMD", "1", "2", "4");
#define ROW(KIND, VG, name, subs)                                                                          \
  {                                                                                                        \
    printf("%-44s", name);                                                                                 \
    for (int blocks : {256, 512, 1024}) { float ns = run<KIND, VG>(in, out, p, iters, blocks); printf(" %7.1f (%5.1f)", ns, ns / (subs)); } \
    printf("\n");                                                                                          \
  }
  ROW(0, true, "control step (10 substeps), VGPR consts", 10)
  ROW(0, false, "control step (10 substeps), SGPR consts", 10)
  ROW(1, true, "triple + max3 test, VGPR consts", 3)
  ROW(1, false, "triple + max3 test, SGPR consts", 3)
  ROW(4, true, "triple, no test, VGPR consts", 3)
  ROW(2, true, "two independent triples, VGPR consts", 6)
  ROW(3, true, "last substep (wrap + sincos), VGPR consts", 1)
  ROW(5, true, "single substep, test behind branch, VGPR", 1)
  printf("\none rollout per lane (latency build's mapping)\n");
#define ROW1(KIND, name, subs)                                                                             \
  {                                                                                                        \
    printf("%-44s", name);                                                                                 \
    for (int blocks : {256, 512, 1024}) { float ns = run1<KIND>(in, out, p, iters, blocks); printf(" %7.1f (%5.1f)", ns, ns / (subs)); } \
    printf("\n");                                                                                          \
  }
  ROW1(0, "control step (10 substeps)", 10)
  ROW1(1, "intermediate substep, test behind branch", 1)
  ROW1(2, "intermediate substep, no test", 1)
  ROW1(5, "intermediate substep, no test, hand-ordered asm", 1)
  ROW1(4, "triple + one test (max3 x, max3 d)", 3)
  ROW1(3, "last substep (wrap + sincos)", 1)
  {
    hipMemset(out, 0, 64);
    hipLaunchKernelGGL((k1<6>), dim3(256), dim3(256), 0, 0, in, out, p, 1);
    unsigned h3[4]; hipMemcpy(h3, out, 16, hipMemcpyDeviceToHost);
    printf("asm substep vs compiled substep over 1000 substeps: %u waves with a differing bit (0 = bit-identical)\n", h3[2]);
  }
  printf("\nns per wave64 instruction, all operands VGPRs\n%-44s %10s %10s %10s\n", "waves per SIMD", "1", "2", "4");
#define ROWV(ILP, MODE, name)                                                                              \
  {                                                                                                        \
    printf("%-44s", name);                                                                                 \
    for (int blocks : {256, 512, 1024}) printf(" %10.3f", runv<ILP, MODE>(in, out, 20000, blocks));         \
    printf("\n");                                                                                          \
  }
  ROWV(1, 0, "v_pk_fma_f32 v,v,v   1 chain") ROWV(2, 0, "v_pk_fma_f32 v,v,v   2 chains") ROWV(3, 0, "v_pk_fma_f32 v,v,v   3 chains")
  ROWV(4, 0, "v_pk_fma_f32 v,v,v   4 chains") ROWV(8, 0, "v_pk_fma_f32 v,v,v   8 chains")
  ROWV(1, 1, "v_pk_mul_f32 v,v     1 chain") ROWV(2, 1, "v_pk_mul_f32 v,v     2 chains") ROWV(4, 1, "v_pk_mul_f32 v,v     4 chains")
  ROWV(1, 2, "v_pk_fma_f32 y,y,v   1 chain") ROWV(4, 2, "v_pk_fma_f32 y,y,v   4 chains")
  ROWV(1, 3, "2 x v_fma_f32 v,v,v  1 chain pair") ROWV(2, 3, "2 x v_fma_f32 v,v,v  2 chain pairs") ROWV(4, 3, "2 x v_fma_f32 v,v,v  4 chain pairs")
  return 0;
}
