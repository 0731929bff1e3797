// Development tool (GPU): where does the dispatcher put the workgroups of a mid-sized grid, and how does a VALU-bound
// dependent chain scale with the waves that share a SIMD?  Each 256-thread block records (XCC id, HW id, start, end)
// and runs ITER dependent v_fma_f32 per lane (ILP independent chains).  The host prints, per grid size and per LDS
// request (an LDS request caps the blocks a CU admits): CUs used, blocks per CU (min / max), wall time and
// ns per wave-instruction per SIMD assuming an even spread.
//   hipcc --offload-arch=gfx950 -O3 tools/dev/census.hip -o /tmp/census && /tmp/census
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <map>
#include <vector>
#include <algorithm>

struct Rec { uint32_t xcc, hwid; unsigned long long t0, t1, c0, c1; };

template <int ILP>
__global__ __launch_bounds__(256) void spin(Rec* rec, float* out, int iters, float a, float b) {
  extern __shared__ float lds[];
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  float x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < ILP; ++i) x[i] = __builtin_fmaf(x[i], a, b);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  if (s == 123.456f) { out[0] = s; lds[threadIdx.x] = s; }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    rec[blockIdx.x] = Rec{xcc & 0xf, hw, t0, t1, c0, c1};
  }
}

template <int ILP>
void run(int grid, size_t lds, int iters, Rec* drec, float* dout) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&spin<ILP>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
  hipLaunchKernelGGL(spin<ILP>, dim3(grid), dim3(256), lds, 0, drec, dout, 10, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(spin<ILP>, dim3(grid), dim3(256), lds, 0, drec, dout, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<Rec> r(grid);
  hipMemcpy(r.data(), drec, grid * sizeof(Rec), hipMemcpyDeviceToHost);
  // CU identity: XCC + (SE, SH, CU) bits of HW_ID (gfx9 layout: cu_id [11:8], sh_id [12], se_id [15:13])
  std::map<uint32_t, std::vector<std::pair<unsigned long long, unsigned long long>>> cu;
  for (auto& q : r) cu[(q.xcc << 16) | (q.hwid & 0xff00)].push_back({q.t0, q.t1});
  int mn = 1 << 30, mx = 0, maxconc = 0;
  for (auto& kv : cu) {
    int n = (int)kv.second.size();
    mn = std::min(mn, n); mx = std::max(mx, n);
    // peak concurrency on this CU
    std::vector<std::pair<unsigned long long, int>> ev;
    for (auto& iv : kv.second) { ev.push_back({iv.first, 1}); ev.push_back({iv.second, -1}); }
    std::sort(ev.begin(), ev.end());
    int c = 0;
    for (auto& e : ev) { c += e.second; maxconc = std::max(maxconc, c); }
  }
  const double instr = (double)iters * 16.0 * ILP;                  // wave-instructions per wave
  const double per_simd_even = instr * grid / 256.0;                 // if blocks are spread evenly, 1 wave/SIMD per block
  printf("ILP %d grid %5d lds %6zu B: %3zu CUs used, blocks/CU min %2d max %2d, peak concurrent blocks on a CU %2d, "
         "%.3f ms, %.3f ns per wave-instr per SIMD (even spread)\n",
         ILP, grid, lds, cu.size(), mn, mx, maxconc, ms, ms * 1e6 / per_simd_even);
}

// A series of short launches (as a small-config MPPI loop issues them): per launch wall time (HIP events) and the shader
// clock the waves saw (s_memtime ticks per 100 MHz s_memrealtime tick), to tell DVFS from placement effects.
void series(int grid, int iters, int launches, bool sync_each, Rec* drec, float* dout) {
  std::vector<hipEvent_t> ev(launches + 1);
  for (auto& e : ev) hipEventCreate(&e);
  std::vector<Rec> all((size_t)launches * grid);
  hipDeviceSynchronize();
  for (int l = 0; l < launches; ++l) {
    hipEventRecord(ev[l]);
    hipLaunchKernelGGL(spin<1>, dim3(grid), dim3(256), 0, 0, drec + (size_t)l * grid, dout, iters, 1.0001f, 0.5f);
    if (sync_each) hipDeviceSynchronize();
  }
  hipEventRecord(ev[launches]);
  hipDeviceSynchronize();
  hipMemcpy(all.data(), drec, all.size() * sizeof(Rec), hipMemcpyDeviceToHost);
  printf("series grid %d iters %d %s:\n  us/launch:", grid, iters, sync_each ? "host sync after each launch" : "back to back");
  for (int l = 0; l < launches; ++l) { float ms; hipEventElapsedTime(&ms, ev[l], ev[l + 1]); printf(" %.0f", ms * 1e3); }
  printf("\n  MHz      :");
  for (int l = 0; l < launches; ++l) {
    double sum = 0;
    for (int b = 0; b < grid; ++b) { const Rec& q = all[(size_t)l * grid + b]; sum += (double)(q.c1 - q.c0) / (double)(q.t1 - q.t0) * 100.0; }
    printf(" %.0f", sum / grid);
  }
  printf("\n");
}

int main() {
  Rec* drec; float* dout;
  hipMalloc(&drec, (size_t)64 * 4096 * sizeof(Rec)); hipMalloc(&dout, 1024);
  series(256, 2500, 40, false, drec, dout);
  series(256, 2500, 40, true, drec, dout);
  series(1024, 2500, 40, false, drec, dout);
  series(2048, 40000, 10, false, drec, dout);
  const int iters = 4000;
  for (int grid : {256, 512, 1024, 2048, 4096}) run<1>(grid, 0, iters, drec, dout);
  printf("-- LDS-shaped occupancy (40 KB per block = at most 4 blocks per CU; 80 KB = 2; 20 KB = 8)\n");
  for (int grid : {512, 1024, 2048}) {
    run<1>(grid, 20 * 1024, iters, drec, dout);
    run<1>(grid, 40 * 1024, iters, drec, dout);
    run<1>(grid, 80 * 1024, iters, drec, dout);
  }
  printf("-- two independent chains per lane\n");
  for (int grid : {256, 512, 1024, 2048}) run<2>(grid, 0, iters, drec, dout);
  return 0;
}
