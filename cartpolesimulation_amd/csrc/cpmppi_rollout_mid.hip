// Explicit instantiations of the rollout kernel, mid-size build (VARIANT 2: two rollouts per lane, loop constants in
// vector registers; launches from one packed wave per SIMD up to ~2 M rollouts); compiled with
// -amdgpu-sched-strategy=iterative-ilp (see __graft_entry__.build).
#include "cpmppi_rollout.hpp"

namespace cpmppi_k {
CPMPPI_MID_INSTANCES(CPMPPI_DEFINE_ROLLOUT)
}  // namespace cpmppi_k

#ifdef CPMPPI_DEBUG_COUNTERS
CPMPPI_DEBUG_READER(cpmppi_debug_read_mid)
CPMPPI_SECTION_READER(cpmppi_debug_sections_mid)
#endif
