// Explicit instantiations of the rollout kernel, two rollouts per lane (compiled with its own scheduling strategy,
// see __graft_entry__.build and cpmppi_rollout.hpp).
#include "cpmppi_rollout.hpp"

namespace cpmppi_k {
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_DELTA_U, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_KNOTS, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBGM, true, NOISE_PHILOX, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_DELTA_U, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_KNOTS, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_DEFAULT, true, NOISE_PHILOX, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_DELTA_U, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_KNOTS, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_LEGACY, true, NOISE_PHILOX, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_DELTA_U, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_KNOTS, 2>(const Params, const StepPtrs);
template __global__ void rollout_cost_kernel<COST_QBG, true, NOISE_PHILOX, 2>(const Params, const StepPtrs);
}  // namespace cpmppi_k
