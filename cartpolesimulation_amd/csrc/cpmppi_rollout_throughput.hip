// Explicit instantiations of the rollout kernel, throughput build (VARIANT 1: both lane mappings and the PRECISE
// arithmetic); compiled with -amdgpu-sched-strategy=max-ilp (see __graft_entry__.build).
#include "cpmppi_rollout.hpp"

namespace cpmppi_k {
CPMPPI_THROUGHPUT_INSTANCES(CPMPPI_DEFINE_ROLLOUT)
}  // namespace cpmppi_k

#ifdef CPMPPI_DEBUG_COUNTERS
// diagnostic build only: event counters, per-wave lifetimes and time stamps of this unit's kernels (tools/dev/cold_counts.py)
CPMPPI_DEBUG_READER(cpmppi_debug_read)
CPMPPI_SECTION_READER(cpmppi_debug_sections_throughput)
CPMPPI_HW_READER(cpmppi_debug_hw_throughput)
#endif
