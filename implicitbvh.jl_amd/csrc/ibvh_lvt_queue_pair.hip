// ibvh_lvt_queue_pair.hip — lvt_queue_kernel (ibvh_lvt_queue.inc) instantiated for traverse(bvh1, bvh2, LVTTraversal())
#include "ibvh_lvt_queue.inc"

namespace ibvh {
namespace lvt {
IBVH_FOR_BBOX_NODE_COMBOS(IBVH_INSTANTIATE_QUEUE, MODE_PAIR)
} // namespace lvt
} // namespace ibvh
