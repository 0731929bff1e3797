// Micro-benchmark: sustained VALU issue rate on gfx950 vs occupancy and ILP (development tool).
// Each kernel runs ITER iterations of an unrolled body of dependent v_fma_f32 chains (ILP independent chains per lane).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ILP, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3f + i;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 y[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) y[i] = f2{x[i], x[i] + 1.0f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (KIND == 0) x[i] = __builtin_fmaf(x[i], a, b);
        else if (KIND == 1) { f2 aa{a, a}, bb{b, b}; y[i] = __builtin_elementwise_fma(y[i], aa, bb); }
        else if (KIND == 2) x[i] = __builtin_amdgcn_rcpf(x[i]) + b;            // 1 trans + 1 add
        else if (KIND == 3) x[i] = (x[i] > a) ? x[i] - b : x[i] + a;            // cmp + 2 add/sel mix
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += (KIND == 1) ? (y[i].x + y[i].y) : x[i];
  if (s == 123.456f) out[0] = s;
}

template <int ILP, int KIND>
double run(float* d, int blocks_per_cu, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL((k<ILP, KIND>), dim3(grid), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<ILP, KIND>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD: each block = 4 waves -> 1 wave per SIMD per block
  double wave_instr_per_simd = (double)blocks_per_cu * iters * 16.0 * ILP * (KIND == 2 ? 2 : (KIND == 3 ? 3 : 1));
  return ms * 1e-3 / wave_instr_per_simd;   // seconds per wave-instruction per SIMD
}

int main() {
  float* d; CHECK(hipMalloc(&d, 1024));
  const int iters = 20000;
  printf("ns per wave64-instruction per SIMD (x clock GHz = cycles); blocks/CU = waves/SIMD\n");
  printf("kind ilp  w=1     w=2     w=4     w=8\n");
#define ROW(ILP, KIND, name) { printf("%-10s %d ", name, ILP); for (int w : {1, 2, 4, 8}) printf(" %7.3f", run<ILP, KIND>(d, w, iters) * 1e9); printf("\n"); }
  ROW(1, 0, "fma") ROW(2, 0, "fma") ROW(4, 0, "fma") ROW(8, 0, "fma")
  ROW(1, 1, "pk_fma") ROW(2, 1, "pk_fma") ROW(4, 1, "pk_fma")
  ROW(1, 2, "rcp+add") ROW(4, 2, "rcp+add")
  ROW(1, 3, "cmp/sel") ROW(4, 3, "cmp/sel")
  return 0;
}
