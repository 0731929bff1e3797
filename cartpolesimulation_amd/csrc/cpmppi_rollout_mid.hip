// Explicit instantiations of the rollout kernel, mid-size build, noise generated or interpolated in the kernel (VARIANT 2
// and 3, NOISE_PHILOX / NOISE_KNOTS: two rollouts per lane, phased horizon loop; VARIANT 3 = launches of at most one wave
// per SIMD, VARIANT 2 up to ~1.5 M rollouts); compiled with the throughput unit's flags (see __graft_entry__.build).
#include "cpmppi_rollout.hpp"

namespace cpmppi_k {
CPMPPI_MID_INSTANCES(CPMPPI_DEFINE_ROLLOUT)
}  // namespace cpmppi_k

#ifdef CPMPPI_DEBUG_COUNTERS
CPMPPI_DEBUG_READER(cpmppi_debug_read_mid)
CPMPPI_SECTION_READER(cpmppi_debug_sections_mid)
CPMPPI_HW_READER(cpmppi_debug_hw_mid)
#endif
