// Explicit instantiations of the rollout kernel, latency build (VARIANT 0: one rollout per lane, launches of at most
// one wave per SIMD); compiled with -amdgpu-sched-strategy=max-memory-clause (see __graft_entry__.build).
#include "cpmppi_rollout.hpp"

namespace cpmppi_k {
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_DELTA_U, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_KNOTS, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_PHILOX, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_DELTA_U, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_KNOTS, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_PHILOX, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_DELTA_U, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_KNOTS, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_PHILOX, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_DELTA_U, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_KNOTS, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_PHILOX, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_TILED, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_TILED, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_TILED, 1, 0>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_TILED, 1, 0>(const Params, const StepPtrs);
}  // namespace cpmppi_k
