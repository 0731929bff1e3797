// Explicit instantiations of the rollout kernel, latency build (VARIANT 0), reference-layout perturbation buffer
// delta_u[E,N,H]; compiled with -amdgpu-sched-strategy=iterative-ilp (see __graft_entry__.build).
#include "cpmppi_rollout.hpp"

namespace cpmppi_k {
CPMPPI_LATENCY_BUFFER_INSTANCES(CPMPPI_DEFINE_ROLLOUT)
}  // namespace cpmppi_k
