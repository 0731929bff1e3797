// Explicit instantiations of the rollout kernel, mid-size build (VARIANT 2: two rollouts per lane, loop constants in
// vector registers; launches from one packed wave per SIMD up to ~2 M rollouts); compiled with
// -amdgpu-sched-strategy=max-ilp (see __graft_entry__.build).
#include "cpmppi_rollout.hpp"

namespace cpmppi_k {
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_DELTA_U, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_KNOTS, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_PHILOX, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_DELTA_U, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_KNOTS, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_PHILOX, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_DELTA_U, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_KNOTS, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_PHILOX, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_DELTA_U, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_KNOTS, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_PHILOX, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_TILED, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_TILED, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_TILED, 2, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_TILED, 2, 2>(const Params, const StepPtrs);
}  // namespace cpmppi_k
