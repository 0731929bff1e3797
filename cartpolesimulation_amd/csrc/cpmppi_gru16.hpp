// cpmppi_gru16.hpp — the GRU predictor of the FAST arithmetic: float32-equivalent products on the f16 matrix cores.
//
// Why.  Measured on MI355X (tools/gru_stamps.py, SQ_VALU_MFMA_COEXEC_CYCLES = 0): v_mfma_f32_32x32x2_f32 does not
// co-execute with vector instructions — its time and the gate math's time simply add (9 984 + ~3 800 cycles per
// 32-rollout tile step), so the exact-f32 kernel of cpmppi_gru.hpp sits at its floor.  The f16 matrix cores run
// 16 x more multiply-adds per cycle AND overlap with the VALU.  Every float32 operand is therefore split into two
// halves, x = hi + lo with hi = f16(x), lo = f16(x - hi)  (x - hi is exact in float32), and a product W X becomes
//      Whi Xhi + Whi Xlo + Wlo Xhi                                   (three v_mfma_f32_32x32x16_f16, f32 accumulate)
// — the dropped Wlo Xlo term is 2^-22 relative, and lo parts that fall into f16's subnormal range are rounded with an
// absolute error <= 3e-8, i.e. at the level of float32 rounding of the O(1) pre-activations.  Per 16 hidden units:
// 3 x 32 cycles instead of 8 x 64.  PRECISE math keeps the exact-f32 kernel.
//
// Mapping (same idea as cpmppi_gru.hpp).  Rollouts are the tile's columns (col = lane & 31), hidden units its rows,
// which live in the 16 accumulator registers: row = (v&3) + 8(v>>2) + 4(lane>>5).  The f16 MFMA takes 8 consecutive
// k-values per lane (k = 8(lane>>5) + t); the weights are stored with the k order permuted so that k-slot
// (block b, lane half, t) IS accumulator register v = 8b + t of that lane half: a result tile becomes the next
// product's B operand by conversion alone — registers 0..7 feed block 0, registers 8..15 block 1, no data movement.
// (operand layout checked on the device by tools/dev/mfma16_layout.hip)
#pragma once
#include "cpmppi_gru.hpp"

namespace cpmppi {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// LDS image, in 16-byte units per lane: fragment f = 64 lanes x 8 f16 = 1 KiB.  A (gate, block) pair owns two
// consecutive fragments: hi, lo.
enum : int {
  HF_L1X = 0,                   // 3 gates x 1 block  (x tile registers 0..7: 5 state features, Q, zeros)
  HF_L1H = HF_L1X + 3 * 1 * 2,  // 3 x 2
  HF_L2X = HF_L1H + 3 * 2 * 2,
  HF_L2H = HF_L2X + 3 * 2 * 2,
  HF_HEAD = HF_L2H + 3 * 2 * 2, // 1 x 2  (rows 0..4 real)
  HF_COUNT = HF_HEAD + 1 * 2 * 2
};
constexpr int G16_FRAG_BYTES = 64 * 16;
constexpr int G16_BIAS_OFF = HF_COUNT * G16_FRAG_BYTES;        // 9 vectors x [2][16] floats: 8 gate biases + head
constexpr int G16_IMAGE_BYTES = G16_BIAS_OFF + 9 * 32 * 4;

struct HSplit { h8 hi, lo; };      // one k-block of a tile as B operand

// registers [8b, 8b+8) of a float32 tile -> (hi, lo).  lo = f16(x - hi) with the difference formed by v_fma_mix_f32, which
// reads hi straight from its packed f16 register (one instruction per element instead of a conversion back to float32
// and a subtraction; the compiler does not form it from `x - (float)hi`).
__device__ __forceinline__ HSplit gru16_split(const f16v& t, int b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  HSplit s;
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const float x0 = t[8 * b + i], x1 = t[8 * b + i + 1];
    const h2 hp = {(_Float16)x0, (_Float16)x1};
    const uint32_t hbits = __builtin_bit_cast(uint32_t, hp);
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hbits), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hbits), "v"(x1));
    s.hi[i] = hp.x; s.hi[i + 1] = hp.y;
    s.lo[i] = (_Float16)l0; s.lo[i + 1] = (_Float16)l1;
  }
  return s;
}

__device__ __forceinline__ const h8* gru16_frag(const char* __restrict__ lds, int f, uint32_t lane) {
  return reinterpret_cast<const h8*>(lds + (size_t)f * G16_FRAG_BYTES + lane * 16);
}

// acc += W X for one (gate, block): fragments f (hi), f+1 (lo)
__device__ __forceinline__ f16v gru16_mm(f16v acc, const char* __restrict__ lds, int f, const HSplit& X, uint32_t lane) {
  const h8 whi = *gru16_frag(lds, f, lane), wlo = *gru16_frag(lds, f + 1, lane);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, X.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, X.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, X.hi, acc, 0, 0, 0);
  return acc;
}

// bias vector b (0..8) as an accumulator tile
__device__ __forceinline__ f16v gru16_bias(const char* __restrict__ lds, int b, uint32_t lane) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4* __restrict__ src = reinterpret_cast<const f4*>(lds + G16_BIAS_OFF + (b * 32 + (lane >> 5) * 16) * 4);
  f16v z;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f4 t = src[q];
    z[4 * q + 0] = t.x; z[4 * q + 1] = t.y; z[4 * q + 2] = t.z; z[4 * q + 3] = t.w;
  }
  return z;
}

struct Gru16Carry {
  f16v ar, az, anh;        // layer-1 pre-activations so far: bias + W_hh1 h1 for the coming step
};

struct Gru16State {
  f16v h1, h2;             // float32 hidden tiles
  HSplit h1s[2], h2s[2];   // their two k-blocks as B operands
};

__device__ __forceinline__ void gru16_resplit(const f16v& h, HSplit s[2]) {
  s[0] = gru16_split(h, 0);
  s[1] = gru16_split(h, 1);
}

// hidden products of layer `fh` (HF_L1H / HF_L2H) into (r, z, n_h)
__device__ __forceinline__ void gru16_hidden_products(const char* __restrict__ lds, int fh, const HSplit hs[2], f16v& pr,
                                                      f16v& pz, f16v& pn, uint32_t lane) {
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    pr = gru16_mm(pr, lds, fh + (0 * 2 + b) * 2, hs[b], lane);
    pz = gru16_mm(pz, lds, fh + (1 * 2 + b) * 2, hs[b], lane);
    pn = gru16_mm(pn, lds, fh + (2 * 2 + b) * 2, hs[b], lane);
  }
}

// Gates on pre-activations that the image builder has pre-scaled: the r and z rows of every weight matrix and bias by
// -log2(e), the n rows by 2 log2(e), so sigmoid(a) = 1 / (1 + 2^a') and tanh(y) = 1 - 2 / (2^y' + 1) need no multiply
// in front of v_exp_f32.
__device__ __forceinline__ void gru16_gates(const f16v& ar, const f16v& az, const f16v& anx, const f16v& anh, f16v& h) {
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ar[v]));
    const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(az[v]));
    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(r, anh[v], anx[v]));
    const float n = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
    h[v] = __builtin_fmaf(z, h[v] - n, n);                    // (1-z)*n + z*h
  }
}

// The gates of one layer (ar, az, anx, anh -> h in place) with the 18 MFMAs of ANOTHER layer's / step's hidden products
// pr / pz / pn += W[fh ...] hs placed BETWEEN them in program order, one MFMA per gate element.
// Why in program order: a wave issues in order, and a v_mfma_f32_32x32x16_f16 occupies the SIMD's matrix pipe for 32 cycles -
// the NEXT MFMA of the same wave cannot issue before that, so 18 MFMAs written back to back hold the wave for 18 x 32 cycles
// whatever follows them; only the instructions placed between two MFMAs run in that gap (MI355X_MICROARCH.md, per-instruction
// constants: an MFMA takes 8 cycles of the SIMD's vector issue, a plain VALU instruction 4, a transcendental 8; issue costs up
// to 24 cycles per gap are hidden).  A gate element is 7 plain + 6 transcendental instructions = 76 cycles of issue: the phase
// is bound by the gate math and the products disappear under it (round 3 issued the 18 MFMAs in front of the gates behind a
// scheduling barrier: 576 + ~1220 cycles per phase; now ~1220 + 18 x 8).  Accumulation order per accumulator is unchanged
// (block 0: hi hi, hi lo, lo hi; block 1 likewise), so the results are bit-identical.  Fragments are fetched two MFMAs ahead.
#ifndef CPMPPI_GRU_INTERLEAVE
#define CPMPPI_GRU_INTERLEAVE 1
#endif
#ifndef CPMPPI_GRU_PACKED_GATES
#define CPMPPI_GRU_PACKED_GATES 1
#endif
__device__ __forceinline__ void gru16_gates_overlapped(const f16v& ar, const f16v& az, const f16v& anx, const f16v& anh, f16v& h,
                                                       const char* __restrict__ lds, int fh, const HSplit hs[2], f16v& pr,
                                                       f16v& pz, f16v& pn, uint32_t lane) {
  // MFMA m = 0..17: block b = m / 9, term = (m % 9) / 3 (0: Whi Xhi, 1: Whi Xlo, 2: Wlo Xhi), gate g = m % 3 (r, z, n)
  auto frag_of = [&](int m) -> const h8* {
    const int b = m / 9, term = (m % 9) / 3, g = m % 3;
    return gru16_frag(lds, fh + (g * 2 + b) * 2 + (term == 2 ? 1 : 0), lane);
  };
  auto issue = [&](int m, const h8& w) __attribute__((always_inline)) {
    const int b = m / 9, term = (m % 9) / 3, g = m % 3;
    const h8& x = (term == 1) ? hs[b].lo : hs[b].hi;
    if (g == 0) pr = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, pr, 0, 0, 0);
    else if (g == 1) pz = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, pz, 0, 0, 0);
    else pn = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, pn, 0, 0, 0);
  };
  h8 w0 = *frag_of(0), w1 = *frag_of(1);
  // the two MFMAs that have no gate element of their own go first
  {
    const h8 w2 = *frag_of(2), w3 = *frag_of(3);
    issue(0, w0);
    issue(1, w1);
    w0 = w2; w1 = w3;
  }
  __builtin_amdgcn_sched_barrier(0);
#if CPMPPI_GRU_PACKED_GATES
  // FOUR gate elements at a time as two register pairs: accumulator registers (v, v+1) are an aligned pair, so the seven plain
  // instructions of an element - 1 + 2^a (twice), r*anh + anx, e + 1, 1 - 2q, h - n, the blend - become seven v_pk_* for TWO
  // elements (the six transcendentals per element stay scalar): 152 -> 124 cycles of vector issue per pair.  Two pairs are
  // evaluated side by side in four stages, each behind one MFMA, so that no instruction follows the one it depends on (a wave
  // issues in order: written pair by pair the chain exp -> add -> rcp -> fma -> add -> fma cost an s_nop per link).  Same
  // operations in the same order per element: bit-identical.
#pragma unroll
  for (int v = 0; v < 16; v += 4) {
    const int m = v + 2;
    h8 wa = w0, wb = w1, wc = w0, wd = w1;
    if (m + 2 < 18) wc = *frag_of(m + 2);
    if (m + 3 < 18) wd = *frag_of(m + 3);
    h8 na = w0, nb = w1;
    if (m + 4 < 18) na = *frag_of(m + 4);
    if (m + 5 < 18) nb = *frag_of(m + 5);
    const f2 one = splat<f2>(1.0f);
    // stage A: 2^ar, 2^az of the four elements, + 1
    issue(m, wa);
    const f2 er0 = f2{__builtin_amdgcn_exp2f(ar[v]), __builtin_amdgcn_exp2f(ar[v + 1])};
    const f2 ez0 = f2{__builtin_amdgcn_exp2f(az[v]), __builtin_amdgcn_exp2f(az[v + 1])};
    const f2 er1 = f2{__builtin_amdgcn_exp2f(ar[v + 2]), __builtin_amdgcn_exp2f(ar[v + 3])};
    const f2 ez1 = f2{__builtin_amdgcn_exp2f(az[v + 2]), __builtin_amdgcn_exp2f(az[v + 3])};
    const f2 dr0 = one + er0, dz0 = one + ez0, dr1 = one + er1, dz1 = one + ez1;
    __builtin_amdgcn_sched_barrier(0);
    // stage B: r, z; the candidate's pre-activation
    issue(m + 1, wb);
    const f2 r0 = rcp_(dr0), r1 = rcp_(dr1), z0 = rcp_(dz0), z1 = rcp_(dz1);
    const f2 y0 = fma_(r0, f2{anh[v], anh[v + 1]}, f2{anx[v], anx[v + 1]});
    const f2 y1 = fma_(r1, f2{anh[v + 2], anh[v + 3]}, f2{anx[v + 2], anx[v + 3]});
    __builtin_amdgcn_sched_barrier(0);
    // stage C: 2^y + 1
    issue(m + 2, wc);
    const f2 e0 = f2{__builtin_amdgcn_exp2f(y0.x), __builtin_amdgcn_exp2f(y0.y)};
    const f2 e1 = f2{__builtin_amdgcn_exp2f(y1.x), __builtin_amdgcn_exp2f(y1.y)};
    const f2 d0 = e0 + one, d1 = e1 + one;
    __builtin_amdgcn_sched_barrier(0);
    // stage D: n = 1 - 2 / (2^y + 1), h = (1 - z) n + z h
    issue(m + 3, wd);
    const f2 q0 = rcp_(d0), q1 = rcp_(d1);
    const f2 n0 = fma_(splat<f2>(-2.0f), q0, one), n1 = fma_(splat<f2>(-2.0f), q1, one);
    const f2 h0 = fma_(z0, f2{h[v], h[v + 1]} - n0, n0), h1 = fma_(z1, f2{h[v + 2], h[v + 3]} - n1, n1);
    h[v] = h0.x; h[v + 1] = h0.y; h[v + 2] = h1.x; h[v + 3] = h1.y;
    w0 = na; w1 = nb;
    __builtin_amdgcn_sched_barrier(0);
  }
#else
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int m = v + 2;
    h8 wn = w1;
    if (m + 2 < 18) wn = *frag_of(m + 2);
    issue(m, w0);
    const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ar[v]));
    const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(az[v]));
    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(r, anh[v], anx[v]));
    const float n = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
    h[v] = __builtin_fmaf(z, h[v] - n, n);                    // (1-z)*n + z*h
    w0 = w1; w1 = wn;
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
}

__device__ __forceinline__ void gru16_carry_init(const char* __restrict__ lds, Gru16State& s, uint32_t lane, Gru16Carry& c) {
  gru16_resplit(s.h1, s.h1s);
  gru16_resplit(s.h2, s.h2s);
  c.ar = gru16_bias(lds, 0, lane); c.az = gru16_bias(lds, 1, lane); c.anh = gru16_bias(lds, 3, lane);
  gru16_hidden_products(lds, HF_L1H, s.h1s, c.ar, c.az, c.anh, lane);
}

// One autoregressive step.  x: feature tile (registers 0..7 meaningful).  Returns the head tile (rows 0..3 in registers
// 0..3 of lane-half 0, row 4 in register 0 of lane-half 1).  The products that do not depend on a layer's gates are
// issued just before them (layer-2 hidden products before the layer-1 gates, the NEXT step's layer-1 hidden products
// before the layer-2 gates) so that the matrix pipe works while the VALU evaluates exp / rcp.
__device__ __forceinline__ f16v gru16_step(const char* __restrict__ lds, const f16v& x, Gru16State& s, Gru16Carry& c,
                                           uint32_t lane
#ifdef CPMPPI_GRU_STAMPS
                                           , unsigned long long* stamp_acc, unsigned long long& stamp_prev
#endif
                                           ) {
  __builtin_amdgcn_sched_barrier(0);
  GRU_STAMP(0);
  const HSplit xs = gru16_split(x, 0);
  // (scheduling barriers between the phases: without them the compiler hoists every fragment load of the step to
  // its top and spills)
  // layer 1: input products on top of the carried hidden products
  const f16v ar = gru16_mm(c.ar, lds, HF_L1X + 0, xs, lane);
  const f16v az = gru16_mm(c.az, lds, HF_L1X + 2, xs, lane);
  const f16v anx = gru16_mm(gru16_bias(lds, 2, lane), lds, HF_L1X + 4, xs, lane);
  const f16v anh = c.anh;
  __builtin_amdgcn_sched_barrier(0);
  GRU_STAMP(1);
  f16v br = gru16_bias(lds, 4, lane), bz = gru16_bias(lds, 5, lane), bnh = gru16_bias(lds, 7, lane);
#if CPMPPI_GRU_INTERLEAVE
  gru16_gates_overlapped(ar, az, anx, anh, s.h1, lds, HF_L2H, s.h2s, br, bz, bnh, lane);   // gates 1 || W_hh2 h2(t-1)
#else
  gru16_hidden_products(lds, HF_L2H, s.h2s, br, bz, bnh, lane);        // W_hh2 h2(t-1): independent of the gates below
  __builtin_amdgcn_sched_barrier(0);
  gru16_gates(ar, az, anx, anh, s.h1);
#endif
  gru16_resplit(s.h1, s.h1s);
  __builtin_amdgcn_sched_barrier(0);
  GRU_STAMP(2);
  // layer 2: input products with the new h1
  f16v bnx = gru16_bias(lds, 6, lane);
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    br = gru16_mm(br, lds, HF_L2X + (0 * 2 + b) * 2, s.h1s[b], lane);
    bz = gru16_mm(bz, lds, HF_L2X + (1 * 2 + b) * 2, s.h1s[b], lane);
    bnx = gru16_mm(bnx, lds, HF_L2X + (2 * 2 + b) * 2, s.h1s[b], lane);
  }
  __builtin_amdgcn_sched_barrier(0);
  GRU_STAMP(3);
  c.ar = gru16_bias(lds, 0, lane); c.az = gru16_bias(lds, 1, lane); c.anh = gru16_bias(lds, 3, lane);
#if CPMPPI_GRU_INTERLEAVE
  gru16_gates_overlapped(br, bz, bnx, bnh, s.h2, lds, HF_L1H, s.h1s, c.ar, c.az, c.anh, lane);   // gates 2 || W_hh1 h1(t) for step t+1
#else
  gru16_hidden_products(lds, HF_L1H, s.h1s, c.ar, c.az, c.anh, lane);  // W_hh1 h1(t) for step t+1
  __builtin_amdgcn_sched_barrier(0);
  gru16_gates(br, bz, bnx, bnh, s.h2);
#endif
  gru16_resplit(s.h2, s.h2s);
  __builtin_amdgcn_sched_barrier(0);
  GRU_STAMP(4);
  f16v out = gru16_bias(lds, 8, lane);
  out = gru16_mm(out, lds, HF_HEAD + 0, s.h2s[0], lane);
  out = gru16_mm(out, lds, HF_HEAD + 2, s.h2s[1], lane);
  __builtin_amdgcn_sched_barrier(0);
  GRU_STAMP(5);
  return out;
}

}  // namespace cpmppi
