// Explicit instantiations of the rollout kernel, throughput build (VARIANT 1: both lane mappings and the PRECISE
// arithmetic); compiled with -amdgpu-sched-strategy=max-ilp (see __graft_entry__.build).
#include "cpmppi_rollout.hpp"

namespace cpmppi_k {
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_DELTA_U, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, false, NOISE_DELTA_U, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_DELTA_U, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_KNOTS, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, false, NOISE_KNOTS, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_KNOTS, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_PHILOX, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, false, NOISE_PHILOX, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_PHILOX, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_DELTA_U, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, false, NOISE_DELTA_U, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_DELTA_U, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_KNOTS, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, false, NOISE_KNOTS, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_KNOTS, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_PHILOX, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, false, NOISE_PHILOX, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_PHILOX, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_DELTA_U, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, false, NOISE_DELTA_U, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_DELTA_U, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_KNOTS, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, false, NOISE_KNOTS, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_KNOTS, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_PHILOX, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, false, NOISE_PHILOX, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_PHILOX, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_DELTA_U, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, false, NOISE_DELTA_U, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_DELTA_U, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_KNOTS, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, false, NOISE_KNOTS, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_KNOTS, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_PHILOX, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, false, NOISE_PHILOX, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_PHILOX, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_TILED, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, false, NOISE_TILED, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_TILED, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_TILED, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, false, NOISE_TILED, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_TILED, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_TILED, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, false, NOISE_TILED, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_TILED, 2, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_TILED, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, false, NOISE_TILED, 1, 1>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_TILED, 2, 1>(const Params, const StepPtrs);
}  // namespace cpmppi_k

#ifdef CPMPPI_DEBUG_COUNTERS
// diagnostic build only: event counters and per-wave lifetimes of this unit's kernels (tools/dev/cold_counts.py)
extern "C" int cpmppi_debug_read(unsigned int* wave_cold, unsigned long long* wave_cycles, unsigned n_waves, int reset) {
  if (wave_cold && n_waves &&
      hipMemcpyFromSymbol(wave_cold, HIP_SYMBOL(cpmppi::g_wave_cold), (size_t)n_waves * sizeof(unsigned int)) != hipSuccess)
    return -1;
  if (wave_cycles && n_waves &&
      hipMemcpyFromSymbol(wave_cycles, HIP_SYMBOL(cpmppi::g_wave_cycles), (size_t)n_waves * sizeof(unsigned long long)) != hipSuccess)
    return -1;
  if (reset) {
    static unsigned int z[16384];
    if (hipMemcpyToSymbol(HIP_SYMBOL(cpmppi::g_wave_cold), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
