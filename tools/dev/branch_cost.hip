// Micro-benchmark (development tool): what a loop back-edge and a VALU-compare -> scalar-branch hand-over cost a LONE wave
// (one wave per SIMD) on gfx950.  Body = B dependent v_pk_fma_f32; variants: plain counted loop; loop whose exit also
// depends on a v_cmp of the body's result (the rollout kernel's event test).  Fit time(B) = a + b*B: a = per-iteration
// control cost, b = per-instruction cost.
//   hipcc -O3 --offload-arch=gfx950 tools/dev/branch_cost.hip -o build_variants/branch_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int B, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b, float lim) {
  f2 y = {threadIdx.x * 1e-3f, threadIdx.x * 1e-3f + 1.0f};
  const f2 aa{a, a}, bb{b, b};
  int it = 0;
  for (; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < B; ++r) y = __builtin_elementwise_fma(y, aa, bb);
    if (KIND == 1) {            // wave-uniform exit on a per-lane compare of the fresh result (never taken here)
      const unsigned long long m = __builtin_amdgcn_fcmpf(__builtin_fabsf(y.x), lim, 3) | __builtin_amdgcn_fcmpf(__builtin_fabsf(y.y), lim, 3);
      if (m != 0ull) break;
    }
  }
  if (y.x + y.y == 123.456f || it == -1) out[0] = y.x;
}

template <int B, int KIND>
float run(float* d, int iters, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<B, KIND>), dim3(blocks), dim3(256), 0, 0, d, 10, 0.999f, 0.001f, 1e30f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<B, KIND>), dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 0.001f, 1e30f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / iters;     // ns per iteration
}

int main() {
  float* d; hipMalloc(&d, 1024);
  const int iters = 200000;
  for (int blocks : {256, 512, 1024}) {
    printf("waves per SIMD %d: ns per loop iteration, body of B dependent v_pk_fma_f32\n", blocks / 256);
    printf("  B        4       8      16      32   | fit: per-iteration control  per-instruction\n");
    for (int kind = 0; kind < 2; ++kind) {
      float t4 = kind ? run<4, 1>(d, iters, blocks) : run<4, 0>(d, iters, blocks);
      float t8 = kind ? run<8, 1>(d, iters, blocks) : run<8, 0>(d, iters, blocks);
      float t16 = kind ? run<16, 1>(d, iters, blocks) : run<16, 0>(d, iters, blocks);
      float t32 = kind ? run<32, 1>(d, iters, blocks) : run<32, 0>(d, iters, blocks);
      float b = (t32 - t8) / 24.0f, a = t8 - 8.0f * b;
      printf("  %-7s %6.1f  %6.1f  %6.1f  %6.1f  | %6.1f ns  %5.2f ns\n", kind ? "cmp+br" : "counted", t4, t8, t16, t32, a, b);
    }
  }
  return 0;
}
