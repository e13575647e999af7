// ibvh_msd_finish_b.hip — instantiations of the finish kernel (ibvh_msd_finish.inc) for part of its geometries; the list
// lives in ibvh_msd_finish.hip (IBVH_FINISH_GEOMETRIES_B).
#if !defined(IBVH_PHASE_STAMPS) // (diagnostic builds: everything in ibvh_msd.hip's translation unit, one stamp buffer)
#include "ibvh_msd_finish.inc"
#include "ibvh_msd_finish_geometries.hpp"
namespace ibvh {
namespace msd {
#define IBVH_FIN(K, T, I) template int launch_finish<K, T, I>(const Plan &, const FinishArgs &, hipStream_t);
IBVH_FINISH_GEOMETRIES_B
#undef IBVH_FIN
} // namespace msd
} // namespace ibvh
#endif
