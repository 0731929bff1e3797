// Explicit instantiations of the rollout kernel for predictor_type "ODE", two rollouts per lane in launches of at most one wave
// per SIMD (VARIANT_ 3: the throughput build's kernel with the substeps as straight-line code and raised wave priority - a lone
// wave pays ~50 cycles per taken branch); compiled like the throughput unit (see __graft_entry__.build).
#include "cpmppi_rollout.hpp"

namespace cpmppi_k {
CPMPPI_ODE_LONE_INSTANCES(CPMPPI_DEFINE_ROLLOUT_ODE)
}  // namespace cpmppi_k
