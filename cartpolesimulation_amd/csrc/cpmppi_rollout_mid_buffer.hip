// Explicit instantiations of the rollout kernel, mid-size build, perturbations from a buffer (VARIANT 2 / 3, NOISE_DELTA_U /
// NOISE_TILED: phased horizon loop, the tiles walked by control step); a unit of its own so that the mid-size kernels
// compile in parallel; same flags as cpmppi_rollout_mid.hip (see __graft_entry__.build).
#include "cpmppi_rollout.hpp"

namespace cpmppi_k {
CPMPPI_MID_BUFFER_INSTANCES(CPMPPI_DEFINE_ROLLOUT)
}  // namespace cpmppi_k

#ifdef CPMPPI_DEBUG_COUNTERS
CPMPPI_DEBUG_READER(cpmppi_debug_read_mid_buffer)
#endif
