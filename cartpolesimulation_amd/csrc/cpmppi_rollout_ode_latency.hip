// Explicit instantiations of the rollout kernel for predictor_type "ODE" (Euler-Cromer, no edge bounce: cpmppi_device.hpp),
// latency build (VARIANT 0: one rollout per lane, launches of at most one wave per SIMD); compiled like the latency unit
// (see __graft_entry__.build).
#include "cpmppi_rollout.hpp"

namespace cpmppi_k {
CPMPPI_ODE_LATENCY_INSTANCES(CPMPPI_DEFINE_ROLLOUT_ODE)
}  // namespace cpmppi_k
