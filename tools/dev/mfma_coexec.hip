// Micro-benchmark (development tool): do MFMA and VALU instructions of DIFFERENT waves on one SIMD overlap on gfx950?
// Blocks of 256 threads = one wave per SIMD; role 0 runs chains of v_mfma_f32_32x32x16_f16, role 1 runs v_fma_f32 chains
// (or transcendental chains), role 2 interleaves both in one wave.  Launches: role 0 alone, role 1 alone, roles 0 and 1
// together (2 blocks per CU, one of each on every SIMD), role 2 alone.  together ~ max(alone) => co-execution,
// together ~ sum => the two pipes serialise at issue.
//   hipcc -O3 --offload-arch=gfx950 tools/dev/mfma_coexec.hip -o build_variants/mfma_coexec && build_variants/mfma_coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int VKIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, int role_a, int role_b, int n_mfma, int n_valu, float a, float b) {
  const int role = (blockIdx.x < 256) ? role_a : role_b;
  h8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(threadIdx.x * 1e-3f + i); hb[i] = (_Float16)(0.5f + i); }
  f16v acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; ++it) {
    if (role == 0 || role == 2) {
      for (int r = 0; r < n_mfma; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[j], 0, 0, 0);
      }
    }
    if (role == 1 || role == 2) {
      for (int r = 0; r < n_valu; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (VKIND == 0) x[i] = __builtin_fmaf(x[i], a, b);
          else x[i] = __builtin_amdgcn_exp2f(x[i]) * a;
        }
      }
    }
  }
  float s = 0;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 123.456f) out[0] = s;
}

template <int VKIND>
float run(float* d, int grid, int ra, int rb, int iters, int nm, int nv) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<VKIND>), dim3(grid), dim3(256), 0, 0, d, 4, ra, rb, nm, nv, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<VKIND>), dim3(grid), dim3(256), 0, 0, d, iters, ra, rb, nm, nv, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

template <int VKIND>
void table(float* d, const char* name) {
  const int iters = 2000;
  // per iteration: nm x 4 MFMAs (32x32x16 f16: 8 passes = 32 cycles each... measured below) and nv x 8 VALU instructions
  for (int nv : {4, 16, 64}) {
    const int nm = 4;
    float t_m = run<VKIND>(d, 256, 0, 0, iters, nm, nv), t_v = run<VKIND>(d, 256, 1, 1, iters, nm, nv);
    float t_both = run<VKIND>(d, 512, 0, 1, iters, nm, nv), t_same = run<VKIND>(d, 256, 2, 2, iters, nm, nv);
    float t_mm = run<VKIND>(d, 512, 0, 0, iters, nm, nv), t_vv = run<VKIND>(d, 512, 1, 1, iters, nm, nv);
    printf("%-6s mfma/iter %2d valu/iter %3d | mfma alone %7.1f us  valu alone %7.1f | two waves (mfma + valu) %7.1f  [sum %7.1f max %7.1f] | "
           "one wave both %7.1f | 2x mfma %7.1f  2x valu %7.1f\n",
           name, nm * 4, nv * 8, t_m, t_v, t_both, t_m + t_v, t_m > t_v ? t_m : t_v, t_same, t_mm, t_vv);
  }
}

int main() {
  float* d; CHECK(hipMalloc(&d, 1024));
  table<0>(d, "fma");
  table<1>(d, "exp2");
  return 0;
}
