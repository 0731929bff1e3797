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
constexpr int GRU_IMAGE_FLOATS = GF_COUNT * 64;

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
