// Explicit instantiations of the rollout kernel, mid-size build, perturbations from a buffer (VARIANT 2, NOISE_DELTA_U /
// NOISE_TILED: triples with rollback, loop constants in vector registers); compiled with
// -amdgpu-sched-strategy=iterative-ilp (see __graft_entry__.build).
#include "cpmppi_rollout.hpp"

namespace cpmppi_k {
CPMPPI_MID_BUFFER_INSTANCES(CPMPPI_DEFINE_ROLLOUT)
}  // namespace cpmppi_k

#ifdef CPMPPI_DEBUG_COUNTERS
CPMPPI_DEBUG_READER(cpmppi_debug_read_mid_buffer)
#endif
