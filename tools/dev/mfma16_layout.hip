// Development check (GPU): operand / result layout of v_mfma_f32_32x32x16_f16 as assumed by cpmppi_gru16.hpp:
//   A: lane l holds A[i = l%32][k = 8*(l/32) + t], t = 0..7;  B: lane l holds B[k = 8*(l/32) + t][j = l%32];
//   C: register v of lane l holds C[row = (v&3) + 8*(v>>2) + 4*(l/32)][col = l%32].
// Build + run on the box: hipcc -O2 --offload-arch=gfx950 tools/dev/mfma16_layout.hip -o /tmp/m16 && /tmp/m16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const _Float16* A, const _Float16* B, float* C) {
  const int l = threadIdx.x;
  h8 a, b;
  for (int t = 0; t < 8; ++t) { a[t] = A[(l % 32) * 16 + 8 * (l / 32) + t]; b[t] = B[(8 * (l / 32) + t) * 32 + (l % 32)]; }
  f16v c;
  for (int v = 0; v < 16; ++v) c[v] = 0.0f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int v = 0; v < 16; ++v) C[((v & 3) + 8 * (v >> 2) + 4 * (l / 32)) * 32 + (l % 32)] = c[v];
}
int main() {
  _Float16 hA[32 * 16], hB[16 * 32];
  float hC[32 * 32], ref[32 * 32];
  srand(1);
  for (int i = 0; i < 512; ++i) { hA[i] = (_Float16)((rand() % 17 - 8) / 8.0f); hB[i] = (_Float16)((rand() % 13 - 6) / 4.0f); }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += (float)hA[i * 16 + kk] * (float)hB[kk * 32 + j]; ref[i * 32 + j] = s; }
  _Float16 *dA, *dB; float* dC;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(hC[i] - ref[i]));
  printf("max |C - ref| = %g  -> layout %s\n", err, err == 0 ? "CONFIRMED" : "WRONG");
  return err == 0 ? 0 : 1;
}
