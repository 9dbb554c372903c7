// kernels_force_lj.hip — LDS-tiled single-centre LJ force kernel (placeholder until the tiled kernel lands:
// returns false so the caller uses the generic kernel).
#include "common.hpp"

namespace ls1 {
bool launch_force_lj(const ForceParams&, hipStream_t, uint32_t*, double*, size_t) { return false; }
}  // namespace ls1
