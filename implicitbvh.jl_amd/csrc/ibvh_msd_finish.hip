// ibvh_msd_finish.hip — the build's sort, second half: every cell of the partitioned records (and every window of sub-cells of a
// crowded cell) is sorted by one workgroup in LDS and written to its final place.  gfx950 only.  See ibvh_msd.hip for the whole
// picture (replaces AK.sort!(leaves, by = bv -> bv.morton), reference src/build.jl:248-253).
// This unit is the dispatcher only: the kernel is a template (ibvh_msd_finish.inc) instantiated in ibvh_msd_finish_a / _b / _c.hip.
#include "ibvh_msd_impl.hpp"
#include "ibvh_msd_finish_geometries.hpp"
#if defined(IBVH_PHASE_STAMPS) // (diagnostic builds: one translation unit — this file is included by ibvh_msd.hip)
#include "ibvh_msd_finish.inc"
#endif

namespace ibvh {
namespace msd {

template <class K, int FT, int FI> int launch_finish(const Plan &p, const FinishArgs &fa_in, hipStream_t st);

int run_finish(const Plan &p, int key_bytes, const FinishArgs &fa, hipStream_t st) {
    int rc = IBVH_ERR_INVALID_ARG;
#define IBVH_FIN(K, T, I) \
    if (sizeof(K) == (size_t)key_bytes && p.ftpb == T && p.fipt == I) rc = launch_finish<K, T, I>(p, fa, st);
    IBVH_FINISH_GEOMETRIES_A IBVH_FINISH_GEOMETRIES_B IBVH_FINISH_GEOMETRIES_C
#undef IBVH_FIN
    return rc;
}

} // namespace msd
} // namespace ibvh
