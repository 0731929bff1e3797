// Explicit instantiations of the rollout kernel for predictor_type "ODE" (Euler-Cromer, no edge bounce: cpmppi_device.hpp),
// throughput build (VARIANT 1: both lane mappings and the PRECISE arithmetic); compiled like the throughput unit
// (see __graft_entry__.build).
#include "cpmppi_rollout.hpp"

namespace cpmppi_k {
CPMPPI_ODE_THROUGHPUT_INSTANCES(CPMPPI_DEFINE_ROLLOUT_ODE)
}  // namespace cpmppi_k
