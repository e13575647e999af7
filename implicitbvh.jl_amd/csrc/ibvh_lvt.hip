// ibvh_lvt.hip — leaf-vs-tree traversal (LVTTraversal) on gfx950, entry points: one work item per leaf / ray walks
// the (same / other) implicit tree depth-first; two passes (count -> inclusive scan -> write) give
// the reference's deterministic contact order.
//
// Replaces src/traverse/leaf_vs_tree/traverse_single.jl, traverse_pair.jl and
// src/raytrace/leaf_vs_tree/leaf_vs_tree.jl.
//
// The walkers live in their own translation units (ibvh_lvt.hpp lists them); this one holds the type dispatch, the
// two-pass protocol and the extern "C" entry points, and instantiates walker 1 (lvt_joint_kernel, the exact walk).
#include "ibvh_lvt.hpp"

namespace ibvh {
namespace lvt {

template <class L, class N, class I, int MODE>
int launch(const Args<L, N, I> &a, const PairCache<I> &cache, bool write, hipStream_t st, const RayBins &rb = RayBins{}, bool *agg_zeroed = nullptr) {
    if (a.n_items == 0) return IBVH_OK;
    const bool count_work = a.work != nullptr; // (the counting pass of the COUNT instantiation; nothing else is launched)
    if (count_work && (!kWorkTypes<L, N, I> || write)) return IBVH_ERR_UNSUPPORTED;
    unsigned blocks = (unsigned)ceil_div(a.n_items, 256);
    if constexpr (MODE == MODE_RAYS) {
        return launch_rays<L, N, I>(a, cache, write, st, rb);
    } else {
        // BBox nodes with at least one node level below the start level: frontier descent + candidate queue (walker 2;
        // trees deeper than 31 levels excepted: its wave-uniform arithmetic is 32-bit); everything else (BSphere nodes,
        // start_level == levels): the exact joint walk
        if constexpr (N::kind == IBVH_BBOX) {
            if (a.start_level < a.tree.levels && a.tree.levels <= 31) return launch_queue<L, N, I, MODE>(a, cache, write, st, agg_zeroed);
        }
        if (count_work) return IBVH_ERR_UNSUPPORTED; // (BSphere nodes / start at the leaf level: the exact walk has no counters)
        // (the exact walk keeps the run-time narrow switch: NARROW = true covers both)
        if (write) IBVH_LAUNCH((lvt_joint_kernel<L, N, I, MODE, true, true>), dim3(blocks), dim3(256), 0, st, a, cache);
        else IBVH_LAUNCH((lvt_joint_kernel<L, N, I, MODE, false, true>), dim3(blocks), dim3(256), 0, st, a, cache);
    }
    IBVH_LAUNCH_CHECK();
    return IBVH_OK;
}


// shared driver of the six entry points
template <int MODE>
int run(const ibvh_bvh *drv, const ibvh_bvh *walk, const void *points, const void *dirs, int64_t n_items,
        int64_t start_level, int32_t narrow, int32_t flip, void *counts, int64_t *total_out, void *contacts, void *scratch,
        size_t scratch_bytes, hipStream_t st, bool enqueue = false, int64_t capacity = 0, int64_t *total_dev = nullptr,
        int64_t *total_host = nullptr, unsigned long long *work = nullptr) {
    // three shapes: count (contacts == nullptr), write (contacts, !enqueue), enqueue = count + scan + guarded write
    const bool write = contacts != nullptr && !enqueue;
    const int32_t positions = (narrow & IBVH_OUTPUT_POSITIONS) ? 1 : 0;
    if (narrow & ~(IBVH_NARROW_MASK | IBVH_OUTPUT_POSITIONS)) return IBVH_ERR_INVALID_ARG;
    narrow &= IBVH_NARROW_MASK;
    if (MODE == MODE_RAYS ? (narrow != IBVH_NARROW_NONE && narrow != IBVH_NARROW_RAY_ORIGIN_OUTSIDE)
                          : (narrow != IBVH_NARROW_NONE && narrow != IBVH_NARROW_MORTON_LT && narrow != IBVH_NARROW_INDEX_LT))
        return IBVH_ERR_INVALID_ARG;
    ibvh_layout lay;
    LeafLayout wl, dl;
    if (!layout_of(walk->types, lay, &wl)) return IBVH_ERR_UNSUPPORTED;
    dl = wl;
    if (drv && !layout_of(drv->types, lay, &dl)) return IBVH_ERR_UNSUPPORTED;
    if (!work && (!scratch || scratch_bytes < scan_scratch_bytes(n_items))) return IBVH_ERR_SCRATCH;
    // RAYS: the quantised shadow of the node levels lives at the END of the scratch when the caller sized it with
    // ibvh_rays_scratch_bytes (and the walk qualifies: rays_shadow_bytes); the contact cache gets what lies in between
    size_t shadow_bytes = 0;
    RayBinPlan bin_plan; // RAYS: the binned path's region, same place, when the tree and the batch qualify (it goes before the shadow)
    if (MODE == MODE_RAYS && !work) {
        bin_plan = rays_bin_plan(*walk, n_items);
        if (bin_plan.cut_level < start_level || scratch_bytes < scan_scratch_bytes(n_items) + bin_plan.bytes + 256) bin_plan = RayBinPlan{};
        if (bin_plan.depth == 0) shadow_bytes = rays_shadow_bytes(*walk, n_items);
        if (scratch_bytes < scan_scratch_bytes(n_items) + shadow_bytes + 256) shadow_bytes = 0;
    }
    // SELF / PAIR under BBox nodes: the rows of the shared descent (ibvh_lvt.hpp "BlockRows") live at the END of the scratch when
    // the caller sized it with ibvh_lvt_scratch_bytes; a smaller scratch simply has none (every wave descends on its own)
    // ... in front of them, every work item's .index (the counting pass writes it for the writing pass: Args::q_index_dense), when there
    // is room for both
    size_t rows_bytes = 0, qidx_bytes = 0;
    if (MODE != MODE_RAYS && !work && walk->types.node_kind == IBVH_BBOX) {
        rows_bytes = blk_rows_bytes(n_items, BLK_SHIFT_MIN);
        qidx_bytes = (size_t)align_up(n_items * (int64_t)(lay.pair_bytes / 2), 256);
        if (scratch_bytes < scan_scratch_bytes(n_items) + rows_bytes + qidx_bytes + 256) qidx_bytes = 0;
        if (scratch_bytes < scan_scratch_bytes(n_items) + rows_bytes + 256) rows_bytes = 0;
        if (!rows_bytes) qidx_bytes = 0;
        rows_bytes += qidx_bytes; // (one tail: [index array | rows])
    }
    const size_t tail_bytes = bin_plan.depth ? bin_plan.bytes : (shadow_bytes ? shadow_bytes : rows_bytes);
    const size_t cache_room = scratch_bytes - (tail_bytes ? tail_bytes + 256 : 0);
    char *tail_ptr = tail_bytes ? (char *)scratch + ((scratch_bytes - tail_bytes) & ~(size_t)255) : nullptr;
    char *shadow_ptr = shadow_bytes ? tail_ptr : nullptr;
    const RayBins bins = bin_plan.depth ? rays_bins_at(bin_plan, tail_ptr) : RayBins{};
    const int K = work ? 0 : cache_slots_for(cache_room, n_items, lay.pair_bytes);
    return dispatch_leaf_node(walk->types, [&](auto lt, auto nt) -> int {
        using L = typename decltype(lt)::type;
        using N = typename decltype(nt)::type;
        if constexpr (MODE == MODE_RAYS && !std::is_same<typename L::elt, typename N::elt>::value) {
            return (int)IBVH_ERR_UNSUPPORTED; // isintersection(::BBox{T}, ::NTuple{3,T}, ...) needs one T
        } else {
            return dispatch_index(walk->types.index_type, [&](auto it) -> int {
                using I = typename decltype(it)::type;
                Args<L, N, I> a;
                a.items = drv ? (const char *)drv->leaves : nullptr;
                a.items_lay = dl;
                a.points = (const typename L::elt *)points;
                a.dirs = (const typename L::elt *)dirs;
                a.n_items = n_items;
                a.leaves = (const char *)walk->leaves;
                a.lay = wl;
                a.nodes = (const N *)walk->nodes;
                a.tree = TreeDev{walk->tree.levels, walk->tree.real_leaves, walk->tree.virtual_leaves};
                a.start_level = start_level;
                a.built_level = walk->built_level;
                a.narrow = narrow;
                a.positions = positions;
                a.flip = flip;
                const int xcd_env = g_tuning.lvt_xcd;
                a.xcd_tiles = MODE != MODE_RAYS ? xcd_env : 0; // 0: round robin, 1: one contiguous range per XCD, n > 1: runs of n
                a.counts = (I *)counts;
                a.contacts = (IndexPair<I> *)contacts;
                a.guard_total = nullptr;
                a.guard_capacity = 0;
                a.work = work;
                a.shadow = shadow_ptr;
                a.rays_filter = 0;
                a.gate = nullptr;
                a.blk_rows = rows_bytes ? (uint32_t *)(tail_ptr + qidx_bytes) : nullptr;
                a.q_index_dense = qidx_bytes ? (I *)tail_ptr : nullptr;
                // (the scan's tile sums live behind the 64-byte header of the scratch: scan_counts)
                const bool may_fuse = !write && !work && MODE != MODE_RAYS;
                a.scan_agg = may_fuse ? (unsigned long long *)((int64_t *)scratch + 8) : nullptr;
                a.scan_nparts = (int32_t)ceil_div(n_items, (int64_t)SCAN_TILE);
                bool agg_zeroed = false;
                a.blk_shift = 0;
                const ibvh_bvh *qside = drv ? drv : walk;
                a.q_nodes = (const N *)qside->nodes;
                a.q_tree = TreeDev{qside->tree.levels, qside->tree.real_leaves, qside->tree.virtual_leaves};
                a.q_built_level = qside->built_level;
                PairCache<I> cache{K ? (IndexPair<I> *)((char *)scratch + scan_scratch_bytes(n_items)) : nullptr, K};
                if (int e = launch<L, N, I, MODE>(a, cache, write, st, bins, &agg_zeroed)) return e;
                if (write || work) return (int)IBVH_OK;
                if (int e = scan_counts<I>((I *)counts, n_items, enqueue ? nullptr : total_out, scratch, st, enqueue ? total_dev : nullptr,
                                           enqueue ? total_host : nullptr, nullptr, agg_zeroed)) return e;
                if (enqueue && capacity > 0) {
                    a.guard_total = total_dev ? (const int64_t *)total_dev : (const int64_t *)scratch; // the total contacts
                    a.guard_capacity = sizeof(I) == 4 && capacity > (int64_t)INT32_MAX ? (int64_t)INT32_MAX : capacity;
                    return launch<L, N, I, MODE>(a, cache, true, st, bins);
                }
                return (int)IBVH_OK;
            });
        }
    });
}

} // namespace lvt
} // namespace ibvh

using namespace ibvh;
using namespace ibvh::lvt;

extern "C" {

// Scratch for the *_count / *_write pair of calls on n_items work items.  cache_slots = contacts
// per work item kept from the counting pass (0 = none: the writing pass walks again; 8 is a good
// default at ~2 contacts per leaf).  The SAME buffer and size must be passed to both calls.
ibvh_status ibvh_lvt_scratch_bytes(const ibvh_types *types, int64_t n_items, int32_t cache_slots, size_t *bytes_out) {
    if (!types || !bytes_out || n_items < 0 || cache_slots < 0) return IBVH_ERR_INVALID_ARG;
    ibvh_layout lay;
    if (!layout_of(*types, lay)) return IBVH_ERR_UNSUPPORTED;
    if (cache_slots > MAX_CACHE_SLOTS) cache_slots = MAX_CACHE_SLOTS;
    *bytes_out = scan_scratch_bytes(n_items) + (size_t)cache_slots * (size_t)n_items * (size_t)lay.pair_bytes;
    if (types->node_kind == IBVH_BBOX) // room for the rows of the shared descent (2 bytes per work item), behind the contact cache
        *bytes_out = (size_t)align_up((int64_t)*bytes_out, 256) + blk_rows_bytes(n_items, BLK_SHIFT_MIN) + 512 +
                     (size_t)align_up(n_items * (int64_t)(lay.pair_bytes / 2), 256); // (+ every work item's .index, for the writing pass)
    return IBVH_OK;
}

// Scratch for the ray traversal entry points: ibvh_lvt_scratch_bytes for num_rays work items plus room for the shadow
// of the node levels the ray walker builds for itself (see include/ibvh.h).
ibvh_status ibvh_rays_scratch_bytes(const ibvh_bvh *bvh, int64_t num_rays, int32_t cache_slots, size_t *bytes_out) {
    if (!bvh || !bytes_out || num_rays < 0) return IBVH_ERR_INVALID_ARG;
    size_t base = 0;
    if (ibvh_status e = ibvh_lvt_scratch_bytes(&bvh->types, num_rays, cache_slots, &base)) return e;
    const RayBinPlan bp = rays_bin_plan(*bvh, num_rays);
    if (bp.depth) { // the binned path keeps no contact cache: its writing pass walks the subtrees out of LDS again
        if (ibvh_status e = ibvh_lvt_scratch_bytes(&bvh->types, num_rays, 0, &base)) return e;
        *bytes_out = (size_t)align_up((int64_t)base, 256) + bp.bytes + 512;
        return IBVH_OK;
    }
    const size_t sh = rays_shadow_bytes(*bvh, num_rays);
    *bytes_out = sh ? (size_t)align_up((int64_t)base, 256) + sh + 512 : base;
    return IBVH_OK;
}

// traverse(bvh, LVTTraversal()) — lvt/traverse_single.jl:1-79
ibvh_status ibvh_traverse_lvt_count(const ibvh_bvh *bvh, int64_t start_level, int32_t narrow, void *counts,
                                    int64_t *total_out, void *scratch, size_t scratch_bytes, void *stream) {
    if (!bvh || !total_out) return IBVH_ERR_INVALID_ARG;
    *total_out = 0;
    if (int e = check_levels(*bvh, start_level)) return (ibvh_status)e;
    if (bvh->tree.real_nodes <= 1) return IBVH_OK; // traverse_single.jl:17-21
    if (!counts || !scratch) return IBVH_ERR_INVALID_ARG;
    return (ibvh_status)run<MODE_SELF>(bvh, bvh, nullptr, nullptr, bvh->tree.real_leaves, start_level, narrow, 0, counts,
                                       total_out, nullptr, scratch, scratch_bytes, (hipStream_t)stream);
}
ibvh_status ibvh_traverse_lvt_write(const ibvh_bvh *bvh, int64_t start_level, int32_t narrow, const void *counts,
                                    void *contacts, void *scratch, size_t scratch_bytes, void *stream) {
    if (!bvh) return IBVH_ERR_INVALID_ARG;
    if (int e = check_levels(*bvh, start_level)) return (ibvh_status)e;
    if (bvh->tree.real_nodes <= 1) return IBVH_OK;
    if (!counts || !contacts) return IBVH_ERR_INVALID_ARG;
    int64_t dummy;
    return (ibvh_status)run<MODE_SELF>(bvh, bvh, nullptr, nullptr, bvh->tree.real_leaves, start_level, narrow, 0,
                                       (void *)counts, &dummy, contacts, scratch, scratch_bytes, (hipStream_t)stream);
}

// count + scan + writing pass in one go, WITHOUT the host read of the total in between (the reference blocks there,
// traverse_single.jl:53-60): the writing pass is launched right behind the scan and does nothing unless the total
// fits `capacity` pairs.  The total stays in the scratch header: read it with ibvh_lvt_total whenever convenient;
// if it exceeds `capacity`, call ibvh_traverse_lvt_write with a larger buffer (counts and scratch are ready for it).
ibvh_status ibvh_traverse_lvt_enqueue(const ibvh_bvh *bvh, int64_t start_level, int32_t narrow, void *counts, void *contacts,
                                      int64_t capacity, void *total_dev, void *total_host, void *scratch, size_t scratch_bytes,
                                      void *stream) {
    if (!bvh || capacity < 0) return IBVH_ERR_INVALID_ARG;
    if (int e = check_levels(*bvh, start_level)) return (ibvh_status)e;
    if (!scratch || scratch_bytes < scan_scratch_bytes(bvh->tree.real_leaves)) return IBVH_ERR_SCRATCH;
    if (bvh->tree.real_nodes <= 1) { // traverse_single.jl:17-21: no contacts
        if (hipMemsetAsync(total_dev ? total_dev : scratch, 0, 8, (hipStream_t)stream) != hipSuccess) return IBVH_ERR_HIP;
        if (total_host) *(volatile int64_t *)total_host = 0; // (host memory: nothing was launched that could write it later)
        return IBVH_OK;
    }
    if (!counts || (capacity > 0 && !contacts)) return IBVH_ERR_INVALID_ARG;
    return (ibvh_status)run<MODE_SELF>(bvh, bvh, nullptr, nullptr, bvh->tree.real_leaves, start_level, narrow, 0, counts,
                                       nullptr, contacts, scratch, scratch_bytes, (hipStream_t)stream, true, capacity,
                                       (int64_t *)total_dev, (int64_t *)total_host);
}
// blocking read of the total contact count a *_count / *_enqueue call left in the scratch header
ibvh_status ibvh_lvt_total(const void *scratch, int64_t *total_out, void *stream) {
    if (!scratch || !total_out) return IBVH_ERR_INVALID_ARG;
    if (hipMemcpyAsync(total_out, scratch, sizeof(int64_t), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return IBVH_ERR_HIP;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return IBVH_ERR_HIP;
    return IBVH_OK;
}

// Work of ONE counting pass (see include/ibvh.h)
ibvh_status ibvh_lvt_work_counters(const ibvh_bvh *bvh, const ibvh_bvh *bvh2, const void *points, const void *directions,
                                   int64_t num_rays, void *counts, void *work_out, void *stream) {
    if (!bvh || !counts || !work_out) return IBVH_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(work_out, 0, 4 * sizeof(unsigned long long), st) != hipSuccess) return IBVH_ERR_HIP;
    const int64_t sl = bvh->built_level > 1 ? bvh->built_level : 1;
    if (int e = check_levels(*bvh, sl)) return (ibvh_status)e; // (all three shapes: levels <= 32, built_level <= start level)
    if (points) {
        if (!directions || num_rays <= 0) return IBVH_ERR_INVALID_ARG;
        return (ibvh_status)run<MODE_RAYS>(nullptr, bvh, points, directions, num_rays, sl, 0, 0, counts, nullptr, nullptr, nullptr,
                                           0, st, false, 0, nullptr, nullptr, (unsigned long long *)work_out);
    }
    if (bvh2) {
        if (!same_types(bvh->types, bvh2->types)) return IBVH_ERR_UNSUPPORTED;
        const bool flip = !(bvh->tree.real_leaves >= bvh2->tree.real_leaves); // the BVH with more leaves drives (:15-36)
        const ibvh_bvh *drv = flip ? bvh2 : bvh, *oth = flip ? bvh : bvh2;
        const int64_t slo = oth->built_level > 1 ? oth->built_level : 1;
        if (int e = check_levels(*oth, slo)) return (ibvh_status)e;
        return (ibvh_status)run<MODE_PAIR>(drv, oth, nullptr, nullptr, drv->tree.real_leaves, slo, 0, flip ? 1 : 0, counts, nullptr,
                                           nullptr, nullptr, 0, st, false, 0, nullptr, nullptr, (unsigned long long *)work_out);
    }
    if (bvh->tree.real_nodes <= 1) return IBVH_OK;
    return (ibvh_status)run<MODE_SELF>(bvh, bvh, nullptr, nullptr, bvh->tree.real_leaves, sl, 0, 0, counts, nullptr, nullptr, nullptr, 0,
                                       st, false, 0, nullptr, nullptr, (unsigned long long *)work_out);
}

// traverse(bvh1, bvh2, LVTTraversal()) — lvt/traverse_pair.jl:1-116
static ibvh_status pair_common(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t sl1, int64_t sl2, int32_t narrow,
                               void *counts, int64_t *total_out, void *contacts, void *scratch, size_t scratch_bytes,
                               void *stream, bool enqueue = false, int64_t capacity = 0, void *total_dev = nullptr,
                               void *total_host = nullptr) {
    if (!bvh1 || !bvh2) return IBVH_ERR_INVALID_ARG;
    if (int e = check_levels(*bvh1, sl1)) return (ibvh_status)e;
    if (int e = check_levels(*bvh2, sl2)) return (ibvh_status)e;
    if (!same_types(bvh1->types, bvh2->types)) return IBVH_ERR_UNSUPPORTED;
    if (!counts) return IBVH_ERR_INVALID_ARG;
    // the BVH with more leaves supplies the work items; flip restores (bvh1, bvh2) order (:15-36).  IBVH_PAIR_SMALLER_DRIVES: the
    // other way round (the contact SET is the same; the list's order is the smaller BVH's leaf order)
    const bool smaller = (narrow & IBVH_PAIR_SMALLER_DRIVES) != 0;
    narrow &= ~IBVH_PAIR_SMALLER_DRIVES;
    const bool flip = smaller ? bvh1->tree.real_leaves > bvh2->tree.real_leaves : !(bvh1->tree.real_leaves >= bvh2->tree.real_leaves);
    const ibvh_bvh *drv = flip ? bvh2 : bvh1, *oth = flip ? bvh1 : bvh2;
    return (ibvh_status)run<MODE_PAIR>(drv, oth, nullptr, nullptr, drv->tree.real_leaves, flip ? sl1 : sl2, narrow,
                                       flip ? 1 : 0, counts, total_out, contacts, scratch, scratch_bytes,
                                       (hipStream_t)stream, enqueue, capacity, (int64_t *)total_dev, (int64_t *)total_host);
}
ibvh_status ibvh_traverse_pair_lvt_count(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t sl1, int64_t sl2,
                                         int32_t narrow, void *counts, int64_t *total_out, void *scratch,
                                         size_t scratch_bytes, void *stream) {
    if (!total_out || !scratch) return IBVH_ERR_INVALID_ARG;
    *total_out = 0;
    return pair_common(bvh1, bvh2, sl1, sl2, narrow, counts, total_out, nullptr, scratch, scratch_bytes, stream);
}
ibvh_status ibvh_traverse_pair_lvt_write(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t sl1, int64_t sl2,
                                         int32_t narrow, const void *counts, void *contacts, void *scratch,
                                         size_t scratch_bytes, void *stream) {
    if (!contacts) return IBVH_ERR_INVALID_ARG;
    int64_t dummy;
    return pair_common(bvh1, bvh2, sl1, sl2, narrow, (void *)counts, &dummy, contacts, scratch, scratch_bytes, stream);
}

ibvh_status ibvh_traverse_pair_lvt_enqueue(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t sl1, int64_t sl2,
                                           int32_t narrow, void *counts, void *contacts, int64_t capacity, void *total_dev,
                                           void *total_host, void *scratch, size_t scratch_bytes, void *stream) {
    if (!scratch || capacity < 0 || (capacity > 0 && !contacts)) return IBVH_ERR_INVALID_ARG;
    return pair_common(bvh1, bvh2, sl1, sl2, narrow, counts, nullptr, contacts, scratch, scratch_bytes, stream, true, capacity,
                       total_dev, total_host);
}

// traverse_rays(bvh, points, directions, LVTTraversal()) — raytrace/leaf_vs_tree/leaf_vs_tree.jl:1-90
static ibvh_status rays_common(const ibvh_bvh *bvh, const void *points, const void *dirs, int64_t num_rays, int64_t sl,
                               int32_t narrow, void *counts, int64_t *total_out, void *contacts, void *scratch, size_t scratch_bytes,
                               void *stream, bool enqueue = false, int64_t capacity = 0, void *total_dev = nullptr,
                               void *total_host = nullptr) {
    if (!bvh || num_rays < 0) return IBVH_ERR_INVALID_ARG;
    if (int e = check_levels(*bvh, sl)) return (ibvh_status)e;
    if (bvh->types.leaf_float != bvh->types.node_float) return IBVH_ERR_UNSUPPORTED;
    if (num_rays == 0) { // :22-26
        if (enqueue && (total_dev || scratch) && hipMemsetAsync(total_dev ? total_dev : scratch, 0, 8, (hipStream_t)stream) != hipSuccess)
            return IBVH_ERR_HIP;
        if (enqueue && total_host) *(volatile int64_t *)total_host = 0;
        return IBVH_OK;
    }
    if (!points || !dirs || !counts) return IBVH_ERR_INVALID_ARG;
    return (ibvh_status)run<MODE_RAYS>(nullptr, bvh, points, dirs, num_rays, sl, narrow, 0, counts, total_out, contacts,
                                       scratch, scratch_bytes, (hipStream_t)stream, enqueue, capacity, (int64_t *)total_dev,
                                       (int64_t *)total_host);
}
ibvh_status ibvh_traverse_rays_lvt_count(const ibvh_bvh *bvh, const void *points, const void *dirs, int64_t num_rays,
                                         int64_t sl, int32_t narrow, void *counts, int64_t *total_out, void *scratch,
                                         size_t scratch_bytes, void *stream) {
    if (!total_out) return IBVH_ERR_INVALID_ARG;
    *total_out = 0;
    if (num_rays > 0 && !scratch) return IBVH_ERR_INVALID_ARG;
    return rays_common(bvh, points, dirs, num_rays, sl, narrow, counts, total_out, nullptr, scratch, scratch_bytes, stream);
}
ibvh_status ibvh_traverse_rays_lvt_write(const ibvh_bvh *bvh, const void *points, const void *dirs, int64_t num_rays,
                                         int64_t sl, int32_t narrow, const void *counts, void *contacts, void *scratch,
                                         size_t scratch_bytes, void *stream) {
    if (num_rays > 0 && !contacts) return IBVH_ERR_INVALID_ARG;
    int64_t dummy;
    return rays_common(bvh, points, dirs, num_rays, sl, narrow, (void *)counts, &dummy, contacts, scratch, scratch_bytes, stream);
}
ibvh_status ibvh_traverse_rays_lvt_enqueue(const ibvh_bvh *bvh, const void *points, const void *dirs, int64_t num_rays,
                                           int64_t sl, int32_t narrow, void *counts, void *contacts, int64_t capacity,
                                           void *total_dev, void *total_host, void *scratch, size_t scratch_bytes, void *stream) {
    if (!scratch || capacity < 0 || (capacity > 0 && !contacts)) return IBVH_ERR_INVALID_ARG;
    return rays_common(bvh, points, dirs, num_rays, sl, narrow, counts, nullptr, contacts, scratch, scratch_bytes, stream, true, capacity,
                       total_dev, total_host);
}

} // extern "C"
