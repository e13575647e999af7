// ibvh_lvt_queue_self.hip — lvt_queue_kernel (ibvh_lvt_queue.inc) instantiated for traverse(bvh, LVTTraversal())
#include "ibvh_lvt_queue.inc"

namespace ibvh {
namespace lvt {
IBVH_FOR_BBOX_NODE_COMBOS(IBVH_INSTANTIATE_QUEUE, MODE_SELF)
} // namespace lvt
} // namespace ibvh

#ifdef IBVH_PHASE_STAMPS
extern "C" int ibvh_debug_lvt_ticks(unsigned long long *out /* 8 */, int reset) {
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(ibvh::lvt::g_lvt_ticks), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ibvh::lvt::g_lvt_ticks), sizeof(unsigned long long) * 8);
}
#endif
