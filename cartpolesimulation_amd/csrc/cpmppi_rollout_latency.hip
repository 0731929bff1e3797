// Explicit instantiations of the rollout kernel, latency build (VARIANT 0: one rollout per lane, launches of at most
// one wave per SIMD), noise from knots / in-kernel Philox / the tiled buffer; compiled with
// -amdgpu-sched-strategy=max-memory-clause (see __graft_entry__.build).
#include "cpmppi_rollout.hpp"

namespace cpmppi_k {
CPMPPI_LATENCY_INSTANCES(CPMPPI_DEFINE_ROLLOUT)
}  // namespace cpmppi_k

#ifdef CPMPPI_DEBUG_COUNTERS
CPMPPI_DEBUG_READER(cpmppi_debug_read_latency)
CPMPPI_SECTION_READER(cpmppi_debug_sections_latency)
CPMPPI_HW_READER(cpmppi_debug_hw_latency)
#endif
