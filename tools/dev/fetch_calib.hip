// Development check (GPU): what rocprofv3's FETCH_SIZE reports for THIS library's load shapes, on a known byte count
// (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count").
//   k_stream4 : 4 B/lane coalesced stream (a wave reads 256 contiguous bytes per instruction), every byte once
//   k_stream16: 16 B/lane coalesced stream (the guide's calibrated case: FETCH_SIZE = 1/2 of the bytes)
//   k_rows32  : the buffer mode's tile fetch: 32-byte segments of rows 200 bytes apart, every byte once overall
// 1 GiB each (beyond the 256 MiB Infinity Cache).  Run: rocprofv3 --pmc FETCH_SIZE -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_stream4(const float* __restrict__ x, size_t n, float* out) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += x[i];
  if (s == 12345.678f) *out = s;
}
__global__ void k_stream16(const float4* __restrict__ x, size_t n4, float* out) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = x[i]; s += v.x + v.y + v.z + v.w;
  }
  if (s == 12345.678f) *out = s;
}
// rows of 50 floats; a wave owns 128 consecutive rows and walks the row in 8-float tiles: lane l, load u fetches
// element (row = (l + 64u) / 8, col = (l + 64u) % 8) of the tile — the access pattern of rollout_cost_kernel's gload
__global__ void k_rows32(const float* __restrict__ x, size_t rows, float* out) {
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
  const size_t row0 = wave * 128;
  if (row0 + 128 > rows) return;
  float s = 0.f;
  for (int k0 = 0; k0 < 50; k0 += 8)
    for (int u = 0; u < 16; ++u) {
      const size_t idx = lane + 64 * u, r = idx / 8, c = idx % 8;
      if (k0 + c < 50) s += x[(row0 + r) * 50 + k0 + c];
    }
  if (s == 12345.678f) *out = s;
}
int main() {
  const size_t n = (size_t)1 << 28;             // 2^28 floats = 1 GiB
  float *x, *out;
  hipMalloc(&x, n * 4); hipMalloc(&out, 4); hipMemset(x, 0, n * 4);
  hipLaunchKernelGGL(k_stream4, dim3(256 * 32), dim3(256), 0, 0, x, n, out);
  hipLaunchKernelGGL(k_stream16, dim3(256 * 32), dim3(256), 0, 0, (const float4*)x, n / 4, out);
  const size_t rows = n / 50 / 128 * 128;
  hipLaunchKernelGGL(k_rows32, dim3((unsigned)(rows / 128 / 4)), dim3(256), 0, 0, x, rows, out);
  hipDeviceSynchronize();
  printf("bytes read once: stream4 %zu, stream16 %zu, rows32 %zu\n", n * 4, n * 4, rows * 200);
  return 0;
}
