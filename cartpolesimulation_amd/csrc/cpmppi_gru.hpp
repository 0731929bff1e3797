// cpmppi_gru.hpp — autoregressive GRU predictor (GRU-6IN-32H1-32H2-5OUT) on the f32 matrix cores of gfx950.
//
// BASELINE.json configs[4] / SURVEY.md §8f N3: "predictor_autoregressive_neural: GRU-32H1-32H2 forward as HIP kernel
// replacing ODE inside same MPPI loop".  The reference class lives in the absent SI_Toolkit submodule; what is in-tree
// is the model naming (SI_Toolkit_ASF/config_predictors.yml:8-13), the feature sets (config_training.yml:12-14), the
// alphabetical feature order (net-info file under GymlikeCartPole/Dense-7IN-32H1-32H2-1OUT-0/) and the output
// augmentation angle = atan2(sin, cos) (ToolkitCustomization/predictors_customization.py:121-127).  The cell follows
// the torch.nn.GRU convention (gates r, z, n; two bias vectors) and is pinned against torch.nn.GRU itself.
//
// Mapping.  Every product is Y[units x rollouts] = W[units x K] * X[K x rollouts] on v_mfma_f32_32x32x2_f32 (exact f32
// FMA chains, 64 FLOP/clk/SIMD): the 32 rollouts of a wave are the tile's COLUMNS (col = lane & 31) and the hidden units
// its ROWS, which sit in the 16 accumulator registers (row = (v&3) + 8*(v>>2) + 4*(lane>>5)).  A result tile is
// therefore already the B operand of the next product — k-step s takes accumulator register s, so lane-half 0 supplies
// row (s&3)+8(s>>2) and lane-half 1 that row + 4 — and the weights are stored in LDS as ready-made A fragments permuted
// to that k order (one conflict-free ds_read_b32 per MFMA).  Hidden states never leave registers, no LDS transposes.
// Biases enter through one extra MFMA per accumulator with B = 1.  The gate non-linearities are lane-local because the
// r, z, n pre-activations of unit j for rollout c all land in the same lane and register index.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cpmppi {

typedef float f16v __attribute__((ext_vector_type(16)));

// fragment indices inside the LDS image (each fragment = 64 floats, one per lane)
enum : int {
  GF_L1X = 0,            // 3 gates x 4 k-steps    (input tile rows 0..7: 5 state features, Q, 2 zero rows)
  GF_L1H = 12,           // 3 x 16
  GF_L1B = 60,           // 4 bias fragments: r (b_ir+b_hr), z (b_iz+b_hz), n_x (b_in), n_h (b_hn)
  GF_L2X = 64,           // 3 x 16   (input = layer-1 hidden tile)
  GF_L2H = 112,          // 3 x 16
  GF_L2B = 160,          // 4
  GF_DW = 164,           // 16       (dense head, rows 0..4 real)
  GF_DB = 180,           // 1
  GF_COUNT = 181
};
// After the fragments: plain vectors for the fused rollout kernel, which takes biases and the 5-row dense head off the
// matrix pipe (25 of 181 MFMAs per step).  Layout [lane half][register v] so that a lane reads 16 consecutive floats
// (4 x ds_read_b128, same address across a half = broadcast).
constexpr int GV_BIAS = GF_COUNT * 64;          // 8 vectors (layer 1: r, z, n_x, n_h; layer 2: the same) x [2][16]
constexpr int GV_HEAD = GV_BIAS + 8 * 32;       // [2][5][16]: w_out[o][row(v, half)]
constexpr int GV_HEADB = GV_HEAD + 2 * 5 * 16;  // b_out[5] (+3 pad)
constexpr int GRU_IMAGE_FLOATS = GV_HEADB + 8;

struct GruNorm {         // normalised = x*scale + shift ; order Q, angleD, angle_cos, angle_sin, position, positionD
  float in_scale[6], in_shift[6], out_scale[5], out_shift[5];
};

// row of a 32x32 accumulator tile held by (register v, lane half hf)
__host__ __device__ inline int gru_tile_row(int v, int hf) { return (v & 3) + 8 * (v >> 2) + 4 * hf; }

template <int KS>
__device__ __forceinline__ f16v gru_mm(f16v acc, const float* __restrict__ frag, const f16v& X, uint32_t lane) {
#pragma unroll
  for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[s * 64 + lane], X[s], acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ f16v gru_bias(const float* __restrict__ frag, uint32_t lane) {
  f16v z;
#pragma unroll
  for (int v = 0; v < 16; ++v) z[v] = 0.0f;
  return __builtin_amdgcn_mfma_f32_32x32x2f32(frag[lane], 1.0f, z, 0, 0, 0);
}

__device__ __forceinline__ float gru_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float gru_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

// One GRU layer for the wave's 32 rollouts.  KX = k-steps of the input tile (4 for the 8-row feature tile, 16 for a
// hidden tile).  h is updated in place.
template <int KX>
__device__ __forceinline__ void gru_layer(const float* __restrict__ lds, int fx, int fh, int fb, const f16v& x, f16v& h,
                                          uint32_t lane) {
  f16v ar = gru_bias(lds + (fb + 0) * 64, lane);
  f16v az = gru_bias(lds + (fb + 1) * 64, lane);
  f16v anx = gru_bias(lds + (fb + 2) * 64, lane);
  f16v anh = gru_bias(lds + (fb + 3) * 64, lane);
  ar = gru_mm<KX>(ar, lds + (fx + 0 * KX) * 64, x, lane);
  az = gru_mm<KX>(az, lds + (fx + 1 * KX) * 64, x, lane);
  anx = gru_mm<KX>(anx, lds + (fx + 2 * KX) * 64, x, lane);
  ar = gru_mm<16>(ar, lds + (fh + 0) * 64, h, lane);
  az = gru_mm<16>(az, lds + (fh + 16) * 64, h, lane);
  anh = gru_mm<16>(anh, lds + (fh + 32) * 64, h, lane);
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const float r = gru_sigmoid(ar[v]);
    const float z = gru_sigmoid(az[v]);
    const float n = gru_tanh(__builtin_fmaf(r, anh[v], anx[v]));
    h[v] = __builtin_fmaf(z, h[v] - n, n);                    // (1-z)*n + z*h
  }
}

// One autoregressive step: x (feature tile, registers 0..3 used) -> h1, h2 updated, normalised outputs in out[0..3]
// (rows 0..3 on lane-half 0, row 4 in register 0 of lane-half 1; rows 5..7 are exact zeros).
__device__ __forceinline__ f16v gru_step(const float* __restrict__ lds, const f16v& x, f16v& h1, f16v& h2, uint32_t lane) {
  gru_layer<4>(lds, GF_L1X, GF_L1H, GF_L1B, x, h1, lane);
  gru_layer<16>(lds, GF_L2X, GF_L2H, GF_L2B, h1, h2, lane);
  return gru_mm<16>(gru_bias(lds + GF_DB * 64, lane), lds + GF_DW * 64, h2, lane);
}


// ------------------------------------------------------------------------------------------------------------------
// Software-pipelined step for the fused rollout kernel.  A wave's MFMAs run in the matrix pipe while its VALU executes
// independent instructions, but within one autoregressive step the gate non-linearities (VALU: exp, rcp) depend on all
// products of their layer.  The products that do NOT depend on them are therefore issued interleaved with the gates:
//   * layer-2 hidden products  W_hh2 h2(t-1)          during the layer-1 gates of step t,
//   * layer-1 hidden products  W_hh1 h1(t) for step t+1  during the layer-2 gates of step t  (carried in GruCarry),
// three MFMAs (one k-step of the r, z, n accumulators) per gate register, A fragments fetched one chunk ahead; a
// scheduling barrier after every chunk keeps the compiler from clumping them again.  Summation order differs from
// gru_step only in that the hidden products enter an accumulator before the input products.
// development aid (-DCPMPPI_GRU_STAMPS): s_memtime at the phase boundaries of a step, summed per wave
#ifdef CPMPPI_GRU_STAMPS
__device__ unsigned long long g_gru_stamp_sum[8];
#define GRU_STAMP(i)                                                                   \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                      \
    stamp_acc[i] += now_ - stamp_prev;                                                 \
    stamp_prev = now_;                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
#else
#define GRU_STAMP(i) do {} while (0)
#endif

struct GruCarry {
  f16v ar, az, anh;      // layer-1 pre-activations so far: bias + W_hh1 h1 for the coming step
};

// bias vector b (0..7) as an accumulator tile: acc[v] = bias[row(v, lane half)]
__device__ __forceinline__ f16v gru_bias_v(const float* __restrict__ lds, int b, uint32_t lane) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4* __restrict__ src = reinterpret_cast<const f4*>(lds + GV_BIAS + b * 32 + (lane >> 5) * 16);
  f16v z;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f4 t = src[q];
    z[4 * q + 0] = t.x; z[4 * q + 1] = t.y; z[4 * q + 2] = t.z; z[4 * q + 3] = t.w;
  }
  return z;
}

// dense head on the VALU: every lane sums its 16 rows, the two halves are added across lanes; out[0..4] on ALL lanes
__device__ __forceinline__ void gru_head_valu(const float* __restrict__ lds, const f16v& h2, uint32_t lane, float out[5]) {
  const float* __restrict__ w = lds + GV_HEAD + (lane >> 5) * 80;
#pragma unroll
  for (int o = 0; o < 5; ++o) {
    float acc = 0.0f;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc = __builtin_fmaf(w[o * 16 + v], h2[v], acc);
    out[o] = acc + __shfl_xor(acc, 32, 64) + lds[GV_HEADB + o];
  }
}

__device__ __forceinline__ void gru_carry_init(const float* __restrict__ lds, const f16v& h1, uint32_t lane, GruCarry& c) {
  c.ar = gru_mm<16>(gru_bias_v(lds, 0, lane), lds + (GF_L1H + 0) * 64, h1, lane);
  c.az = gru_mm<16>(gru_bias_v(lds, 1, lane), lds + (GF_L1H + 16) * 64, h1, lane);
  c.anh = gru_mm<16>(gru_bias_v(lds, 3, lane), lds + (GF_L1H + 32) * 64, h1, lane);
}

// gates of one layer (registers ar, az, anx, anh -> h in place) interleaved with the hidden products
// pr/pz/pn += W[fh + {0,16,32} + s] * X[s] of ANOTHER layer / step.
__device__ __forceinline__ void gru_gates_overlapped(const f16v& ar, const f16v& az, const f16v& anx, const f16v& anh,
                                                     f16v& h, const float* __restrict__ lds, int fh, const f16v& X,
                                                     f16v& pr, f16v& pz, f16v& pn, uint32_t lane) {
  const float* __restrict__ f = lds + fh * 64 + lane;
  float a0 = f[0], a1 = f[16 * 64], a2 = f[32 * 64];
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f;
    if (v + 1 < 16) { n0 = f[(v + 1) * 64]; n1 = f[(16 + v + 1) * 64]; n2 = f[(32 + v + 1) * 64]; }
#ifndef GRU_EXP_NOMFMA
    pr = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, X[v], pr, 0, 0, 0);
    pz = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, X[v], pz, 0, 0, 0);
    pn = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, X[v], pn, 0, 0, 0);
#else
    pr[v] += a0; pz[v] += a1; pn[v] += a2;
#endif
#ifndef GRU_EXP_NOGATES
    const float r = gru_sigmoid(ar[v]);
    const float z = gru_sigmoid(az[v]);
    const float n = gru_tanh(__builtin_fmaf(r, anh[v], anx[v]));
    h[v] = __builtin_fmaf(z, h[v] - n, n);                    // (1-z)*n + z*h
#else
    h[v] = __builtin_fmaf(ar[v], az[v], anx[v] + anh[v]) * 1e-3f;
#endif
    a0 = n0; a1 = n1; a2 = n2;
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ void gru_step_pipelined(const float* __restrict__ lds, const f16v& x, f16v& h1, f16v& h2,
                                                   GruCarry& c, uint32_t lane, float out[5]
#ifdef CPMPPI_GRU_STAMPS
                                                   , unsigned long long* stamp_acc, unsigned long long& stamp_prev
#endif
                                                   ) {
  GRU_STAMP(0);
  // layer 1: input products on top of the carried hidden products
  const f16v ar = gru_mm<4>(c.ar, lds + (GF_L1X + 0) * 64, x, lane);
  const f16v az = gru_mm<4>(c.az, lds + (GF_L1X + 4) * 64, x, lane);
  const f16v anx = gru_mm<4>(gru_bias_v(lds, 2, lane), lds + (GF_L1X + 8) * 64, x, lane);
  f16v br = gru_bias_v(lds, 4, lane);
  f16v bz = gru_bias_v(lds, 5, lane);
  f16v bnx = gru_bias_v(lds, 6, lane);
  f16v bnh = gru_bias_v(lds, 7, lane);
  GRU_STAMP(1);
  gru_gates_overlapped(ar, az, anx, c.anh, h1, lds, GF_L2H, h2, br, bz, bnh, lane);       // gates 1 || W_hh2 h2
  GRU_STAMP(2);
  // layer 2: input products with the new h1
  br = gru_mm<16>(br, lds + (GF_L2X + 0) * 64, h1, lane);
  bz = gru_mm<16>(bz, lds + (GF_L2X + 16) * 64, h1, lane);
  bnx = gru_mm<16>(bnx, lds + (GF_L2X + 32) * 64, h1, lane);
  c.ar = gru_bias_v(lds, 0, lane);
  c.az = gru_bias_v(lds, 1, lane);
  c.anh = gru_bias_v(lds, 3, lane);
  GRU_STAMP(3);
  gru_gates_overlapped(br, bz, bnx, bnh, h2, lds, GF_L1H, h1, c.ar, c.az, c.anh, lane);   // gates 2 || W_hh1 h1 (next step)
  GRU_STAMP(4);
  gru_head_valu(lds, h2, lane, out);
  GRU_STAMP(5);
}

// Next state (without the angle) from head outputs held on every lane; cos(atan2(s, c)) = c / |(c, s)| spares the
// per-step atan2f + cosf of the augmentation (the angle itself is needed only for the terminal cost).
__device__ __forceinline__ void gru_output_state_fast(const GruNorm& nm, const float out[5], float st[6], float& cosang) {
  st[1] = __builtin_fmaf(out[0], nm.out_scale[0], nm.out_shift[0]);
  st[2] = __builtin_fmaf(out[1], nm.out_scale[1], nm.out_shift[1]);
  st[3] = __builtin_fmaf(out[2], nm.out_scale[2], nm.out_shift[2]);
  st[4] = __builtin_fmaf(out[3], nm.out_scale[3], nm.out_shift[3]);
  st[5] = __builtin_fmaf(out[4], nm.out_scale[4], nm.out_shift[4]);
  const float r2 = __builtin_fmaf(st[2], st[2], st[3] * st[3]);
  cosang = r2 > 0.0f ? st[2] * __builtin_amdgcn_rsqf(r2) : 1.0f;
}

// Hidden state of one rollout/env: hsrc[32] -> tile registers of this lane.
__device__ __forceinline__ f16v gru_load_hidden(const float* __restrict__ hsrc, uint32_t lane) {
  f16v h;
#pragma unroll
  for (int v = 0; v < 16; ++v) h[v] = hsrc ? hsrc[gru_tile_row(v, lane >> 5)] : 0.0f;
  return h;
}

// Feature tile of step 0 from a state (angle, angleD, cos, sin, position, positionD) and the first control.
__device__ __forceinline__ f16v gru_input_tile(const GruNorm& nm, const float* __restrict__ s, float Q, uint32_t lane) {
  f16v x;
#pragma unroll
  for (int v = 0; v < 16; ++v) x[v] = 0.0f;
  if ((lane >> 5) == 0) {
    x[0] = __builtin_fmaf(s[1], nm.in_scale[1], nm.in_shift[1]);      // angleD
    x[1] = __builtin_fmaf(s[2], nm.in_scale[2], nm.in_shift[2]);      // angle_cos
    x[2] = __builtin_fmaf(s[3], nm.in_scale[3], nm.in_shift[3]);      // angle_sin
    x[3] = __builtin_fmaf(s[4], nm.in_scale[4], nm.in_shift[4]);      // position
  } else {
    x[0] = __builtin_fmaf(s[5], nm.in_scale[5], nm.in_shift[5]);      // positionD (row 4)
    x[1] = __builtin_fmaf(Q, nm.in_scale[0], nm.in_shift[0]);         // Q         (row 5)
  }
  return x;
}

// De-normalised next state of the lane's rollout from the output tile (valid on lanes 0..31).
__device__ __forceinline__ void gru_output_state(const GruNorm& nm, const f16v& out, uint32_t lane, float st[6]) {
  const float r4 = __shfl(out[0], (int)((lane & 31u) + 32u), 64);    // positionD lives on the partner lane
  const float angleD = __builtin_fmaf(out[0], nm.out_scale[0], nm.out_shift[0]);
  const float c = __builtin_fmaf(out[1], nm.out_scale[1], nm.out_shift[1]);
  const float s = __builtin_fmaf(out[2], nm.out_scale[2], nm.out_shift[2]);
  st[0] = atan2f(s, c);                                               // predictors_customization.py:121-127
  st[1] = angleD; st[2] = c; st[3] = s;
  st[4] = __builtin_fmaf(out[3], nm.out_scale[3], nm.out_shift[3]);
  st[5] = __builtin_fmaf(r4, nm.out_scale[4], nm.out_shift[4]);
}

}  // namespace cpmppi
