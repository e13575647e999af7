// ibvh_lvt.hip — leaf-vs-tree traversal (LVTTraversal) on gfx950: one work item per leaf / ray walks
// the (same / other) implicit tree depth-first; two passes (count -> inclusive scan -> write) give
// the reference's deterministic contact order.
//
// Replaces src/traverse/leaf_vs_tree/traverse_single.jl, traverse_pair.jl and
// src/raytrace/leaf_vs_tree/leaf_vs_tree.jl.
//
// The reference keeps a 32-entry per-thread stack of pending right children
// (traverse_single.jl:188-203).  The implicit tree makes that stack redundant: the pre-order
// successor of a finished subtree rooted at i is (i+1) >> ctz(i+1) — climb while i is a right
// child, then step to the sibling — clamped so the climb never rises above start_level (the
// reference iterates the start-level roots in order, which is the same sequence).  A virtual
// successor ends the walk: virtual nodes form a suffix of every level, so everything after it in
// pre-order is virtual too.  Visitation order, hence contact order, is identical to the stack
// version; no scratch memory, no private-array spills.
#include "ibvh_common.hpp"

namespace ibvh {
namespace lvt {

enum { MODE_SELF = 0, MODE_PAIR = 1, MODE_RAYS = 2 };

template <class L, class N, class I> struct Args {
    // work items
    const char *items;       // driving leaves (SELF/PAIR)
    LeafLayout items_lay;
    const typename L::elt *points; // RAYS: (3, n) column-major
    const typename L::elt *dirs;
    int64_t n_items;
    // the tree being walked
    const char *leaves;
    LeafLayout lay;
    const N *nodes;
    TreeDev tree;
    int64_t start_level;
    int32_t narrow;
    int32_t flip;
    // outputs
    I *counts;                 // count pass: per-item counts; write pass: inclusive prefix
    IndexPair<I> *contacts;
};

IBVH_D bool narrow_eval(int narrow, uint64_t ma, int64_t ia, uint64_t mb, int64_t ib) {
    if (narrow == IBVH_NARROW_MORTON_LT) return ma < mb;
    if (narrow == IBVH_NARROW_INDEX_LT) return ia < ib;
    return true;
}

template <class L, class N, class I, int MODE, bool WRITE>
__global__ __launch_bounds__(256) void lvt_kernel(Args<L, N, I> a) {
    using T = typename L::elt;
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (item >= a.n_items) return;

    // the query: a leaf (converted once to the node type for node tests, traverse_single.jl:154-155)
    // or a ray converted to the leaf float type (raytrace/leaf_vs_tree:116-125)
    L q_leaf;
    N q_node;
    I q_index = 0;
    uint64_t q_morton = 0;
    T p[3], d[3];
    if constexpr (MODE == MODE_RAYS) {
        p[0] = a.points[3 * item + 0];
        p[1] = a.points[3 * item + 1];
        p[2] = a.points[3 * item + 2];
        d[0] = a.dirs[3 * item + 0];
        d[1] = a.dirs[3 * item + 1];
        d[2] = a.dirs[3 * item + 2];
    } else {
        const char *rec = a.items + item * a.items_lay.stride;
        q_leaf = load_vol<L>(rec);
        q_node = convert_to(q_leaf, (N *)nullptr);
        q_index = load_index<I>(rec, a.items_lay);
        if (a.narrow == IBVH_NARROW_MORTON_LT) q_morton = load_morton(rec, a.items_lay);
    }

    const int64_t levels = a.tree.levels, vl = a.tree.virtual_leaves;
    const int64_t leaf_first = int64_t(1) << (levels - 1); // implicit index of leaf position 1
    const int64_t self_implicit = item + leaf_first;       // SELF: implicit index of this leaf

    int64_t cnt = 0;
    int64_t w = 0;
    if constexpr (WRITE) w = item == 0 ? 0 : (int64_t)a.counts[item - 1];

    int64_t level = a.start_level;
    int64_t inode = int64_t(1) << (level - 1);
    // level_num_real(start_level) >= 1 always, so the first root is real
    while (true) {
        bool descend = false;
        bool skip = false;
        if constexpr (MODE == MODE_SELF) {
            // ignore subtrees whose right-most reachable leaf is not to the right of this leaf
            // (traverse_single.jl:165-167): only partners j > i are reported
            int64_t rightmost = ((inode + 1) << (levels - level)) - 1;
            skip = rightmost <= self_implicit;
        }
        if (!skip) {
            if (level == levels) {
                const char *rec = a.leaves + (inode - leaf_first) * a.lay.stride;
                L leaf = load_vol<L>(rec);
                bool hit;
                if constexpr (MODE == MODE_RAYS) hit = isintersection(leaf, p, d);
                else hit = iscontact(q_leaf, leaf);
                if (hit) {
                    I lidx = load_index<I>(rec, a.lay);
                    if constexpr (MODE != MODE_RAYS) {
                        if (a.narrow != IBVH_NARROW_NONE) {
                            uint64_t lm = a.narrow == IBVH_NARROW_MORTON_LT ? load_morton(rec, a.lay) : 0;
                            // pair with flip: narrow(leaf, bv) (traverse_pair.jl:202)
                            hit = (MODE == MODE_PAIR && a.flip) ? narrow_eval(a.narrow, lm, lidx, q_morton, q_index)
                                                                : narrow_eval(a.narrow, q_morton, q_index, lm, lidx);
                        }
                    }
                    if (hit) {
                        if constexpr (WRITE) {
                            IndexPair<I> c;
                            if constexpr (MODE == MODE_SELF) {
                                c = q_index > lidx ? IndexPair<I>{lidx, q_index} : IndexPair<I>{q_index, lidx};
                            } else if constexpr (MODE == MODE_PAIR) {
                                c = a.flip ? IndexPair<I>{lidx, q_index} : IndexPair<I>{q_index, lidx};
                            } else {
                                c = IndexPair<I>{lidx, (I)(item + 1)};
                            }
                            a.contacts[w++] = c;
                        } else {
                            ++cnt;
                        }
                    }
                }
            } else {
                N node = load_vol<N>(a.nodes + (inode - level_skips(levels, vl, level) - 1));
                if constexpr (MODE == MODE_RAYS) descend = isintersection(node, p, d);
                else descend = iscontact(q_node, node);
            }
        }
        if (descend) { // the left child of a real node is always real
            inode = 2 * inode;
            level += 1;
            continue;
        }
        // pre-order successor, never climbing above start_level
        int64_t up = (int64_t)__builtin_ctzll((unsigned long long)(inode + 1));
        int64_t room = level - a.start_level;
        up = up < room ? up : room;
        inode = (inode + 1) >> up;
        level -= up;
        if (inode - (int64_t(1) << (level - 1)) >= level_num_real(levels, vl, level)) break; // virtual / past the level
    }
    if constexpr (!WRITE) a.counts[item] = (I)cnt;
}

// ---- inclusive scan of the per-item counts (AK.accumulate!, traverse_single.jl:57) ---------------
constexpr int SCAN_TPB = 256, SCAN_IPT = 16, SCAN_TILE = SCAN_TPB * SCAN_IPT;

IBVH_D int64_t block_sum(int64_t v, int64_t *s_w) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    int64_t t = 0;
#pragma unroll
    for (int i = 0; i < SCAN_TPB / 64; ++i) t += s_w[i];
    __syncthreads();
    return t;
}

template <class I> __global__ __launch_bounds__(SCAN_TPB) void scan_reduce_kernel(const I *c, int64_t n, int64_t *partials) {
    __shared__ int64_t s_w[SCAN_TPB / 64];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE, v = 0;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) {
        int64_t i = base + j * SCAN_TPB + threadIdx.x;
        if (i < n) v += (int64_t)c[i];
    }
    int64_t t = block_sum(v, s_w);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
// one workgroup: exclusive scan of the tile sums in place; total -> totals[0]
__global__ __launch_bounds__(SCAN_TPB) void scan_partials_kernel(int64_t *partials, int64_t nparts, int64_t *totals) {
    __shared__ int64_t s_w[SCAN_TPB / 64];
    int64_t carry = 0;
    for (int64_t base = 0; base < nparts; base += SCAN_TPB) {
        int64_t i = base + threadIdx.x;
        int64_t v = i < nparts ? partials[i] : 0, inc = v;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int64_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int64_t wb = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < SCAN_TPB / 64; ++k) {
            int64_t t = s_w[k];
            if (k < w) wb += t;
            tot += t;
        }
        __syncthreads();
        if (i < nparts) partials[i] = carry + wb + inc - v;
        carry += tot;
    }
    if (threadIdx.x == 0) totals[0] = carry;
}
template <class I>
__global__ __launch_bounds__(SCAN_TPB) void scan_apply_kernel(I *c, int64_t n, const int64_t *partials) {
    __shared__ int64_t s_w[SCAN_TPB / 64];
    // thread owns SCAN_IPT consecutive items so the in-thread running sum is in memory order
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_IPT;
    int64_t v[SCAN_IPT], sum = 0;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) {
        int64_t i = base + j;
        v[j] = i < n ? (int64_t)c[i] : 0;
        sum += v[j];
    }
    int64_t inc = sum;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int64_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int64_t wb = 0;
#pragma unroll
    for (int k = 0; k < SCAN_TPB / 64; ++k)
        if (k < w) wb += s_w[k];
    int64_t run = partials[blockIdx.x] + wb + inc - sum;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) {
        int64_t i = base + j;
        run += v[j];
        if (i < n) c[i] = (I)run;
    }
}

inline size_t scan_scratch_bytes(int64_t n) { return (size_t)(ceil_div(n > 0 ? n : 1, SCAN_TILE) + 8) * 8; }

// inclusive scan in place + blocking read of the total (the reference's @allowscalar, :60)
template <class I> int scan_counts(I *counts, int64_t n, int64_t *total_out, void *scratch, hipStream_t st) {
    int64_t nparts = ceil_div(n, SCAN_TILE);
    int64_t *totals = (int64_t *)scratch;
    int64_t *partials = totals + 8;
    IBVH_LAUNCH((scan_reduce_kernel<I>), dim3((unsigned)nparts), dim3(SCAN_TPB), 0, st, counts, n, partials);
    IBVH_LAUNCH(scan_partials_kernel, dim3(1), dim3(SCAN_TPB), 0, st, partials, nparts, totals);
    IBVH_LAUNCH((scan_apply_kernel<I>), dim3((unsigned)nparts), dim3(SCAN_TPB), 0, st, counts, n, partials);
    IBVH_LAUNCH_CHECK();
    int64_t total = 0;
    IBVH_HIP_CHECK(hipMemcpyAsync(&total, totals, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    IBVH_HIP_CHECK(hipStreamSynchronize(st));
    *total_out = total;
    if (sizeof(I) == 4 && total > (int64_t)INT32_MAX) return IBVH_ERR_OVERFLOW;
    return IBVH_OK;
}

inline int check_levels(const ibvh_bvh &b, int64_t start_level) {
    // @argcheck bvh.built_level <= start_level <= bvh.tree.levels <= 32 (traverse_single.jl:10)
    if (!(b.built_level <= start_level && start_level <= b.tree.levels && b.tree.levels <= 32)) return IBVH_ERR_INVALID_ARG;
    if (start_level < 1) return IBVH_ERR_INVALID_ARG;
    return IBVH_OK;
}
inline bool same_types(const ibvh_types &x, const ibvh_types &y) {
    return x.leaf_kind == y.leaf_kind && x.leaf_float == y.leaf_float && x.node_kind == y.node_kind &&
           x.node_float == y.node_float && x.index_type == y.index_type && x.morton_type == y.morton_type;
}

template <class L, class N, class I, int MODE>
int launch(const Args<L, N, I> &a, bool write, hipStream_t st) {
    if (a.n_items == 0) return IBVH_OK;
    unsigned blocks = (unsigned)ceil_div(a.n_items, 256);
    if (write) IBVH_LAUNCH((lvt_kernel<L, N, I, MODE, true>), dim3(blocks), dim3(256), 0, st, a);
    else IBVH_LAUNCH((lvt_kernel<L, N, I, MODE, false>), dim3(blocks), dim3(256), 0, st, a);
    IBVH_LAUNCH_CHECK();
    return IBVH_OK;
}

// shared driver of the six entry points
template <int MODE>
int run(const ibvh_bvh *drv, const ibvh_bvh *walk, const void *points, const void *dirs, int64_t n_items,
        int64_t start_level, int32_t narrow, int32_t flip, void *counts, int64_t *total_out, void *contacts, void *scratch,
        size_t scratch_bytes, hipStream_t st) {
    const bool write = contacts != nullptr;
    ibvh_layout lay;
    LeafLayout wl, dl;
    if (!layout_of(walk->types, lay, &wl)) return IBVH_ERR_UNSUPPORTED;
    dl = wl;
    if (drv && !layout_of(drv->types, lay, &dl)) return IBVH_ERR_UNSUPPORTED;
    if (!write && scratch_bytes < scan_scratch_bytes(n_items)) return IBVH_ERR_SCRATCH;
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
                a.narrow = narrow;
                a.flip = flip;
                a.counts = (I *)counts;
                a.contacts = (IndexPair<I> *)contacts;
                if (int e = launch<L, N, I, MODE>(a, write, st)) return e;
                if (!write) return scan_counts<I>((I *)counts, n_items, total_out, scratch, st);
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

ibvh_status ibvh_lvt_scratch_bytes(int64_t n_items, size_t *bytes_out) {
    if (!bytes_out || n_items < 0) return IBVH_ERR_INVALID_ARG;
    *bytes_out = scan_scratch_bytes(n_items);
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
                                    void *contacts, void *stream) {
    if (!bvh) return IBVH_ERR_INVALID_ARG;
    if (int e = check_levels(*bvh, start_level)) return (ibvh_status)e;
    if (bvh->tree.real_nodes <= 1) return IBVH_OK;
    if (!counts || !contacts) return IBVH_ERR_INVALID_ARG;
    int64_t dummy;
    return (ibvh_status)run<MODE_SELF>(bvh, bvh, nullptr, nullptr, bvh->tree.real_leaves, start_level, narrow, 0,
                                       (void *)counts, &dummy, contacts, nullptr, 0, (hipStream_t)stream);
}

// traverse(bvh1, bvh2, LVTTraversal()) — lvt/traverse_pair.jl:1-116
static ibvh_status pair_common(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t sl1, int64_t sl2, int32_t narrow,
                               void *counts, int64_t *total_out, void *contacts, void *scratch, size_t scratch_bytes,
                               void *stream) {
    if (!bvh1 || !bvh2) return IBVH_ERR_INVALID_ARG;
    if (int e = check_levels(*bvh1, sl1)) return (ibvh_status)e;
    if (int e = check_levels(*bvh2, sl2)) return (ibvh_status)e;
    if (!same_types(bvh1->types, bvh2->types)) return IBVH_ERR_UNSUPPORTED;
    if (!counts) return IBVH_ERR_INVALID_ARG;
    // the BVH with more leaves supplies the work items; flip restores (bvh1, bvh2) order (:15-36)
    const bool flip = !(bvh1->tree.real_leaves >= bvh2->tree.real_leaves);
    const ibvh_bvh *drv = flip ? bvh2 : bvh1, *oth = flip ? bvh1 : bvh2;
    return (ibvh_status)run<MODE_PAIR>(drv, oth, nullptr, nullptr, drv->tree.real_leaves, flip ? sl1 : sl2, narrow,
                                       flip ? 1 : 0, counts, total_out, contacts, scratch, scratch_bytes,
                                       (hipStream_t)stream);
}
ibvh_status ibvh_traverse_pair_lvt_count(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t sl1, int64_t sl2,
                                         int32_t narrow, void *counts, int64_t *total_out, void *scratch,
                                         size_t scratch_bytes, void *stream) {
    if (!total_out || !scratch) return IBVH_ERR_INVALID_ARG;
    *total_out = 0;
    return pair_common(bvh1, bvh2, sl1, sl2, narrow, counts, total_out, nullptr, scratch, scratch_bytes, stream);
}
ibvh_status ibvh_traverse_pair_lvt_write(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t sl1, int64_t sl2,
                                         int32_t narrow, const void *counts, void *contacts, void *stream) {
    if (!contacts) return IBVH_ERR_INVALID_ARG;
    int64_t dummy;
    return pair_common(bvh1, bvh2, sl1, sl2, narrow, (void *)counts, &dummy, contacts, nullptr, 0, stream);
}

// traverse_rays(bvh, points, directions, LVTTraversal()) — raytrace/leaf_vs_tree/leaf_vs_tree.jl:1-90
static ibvh_status rays_common(const ibvh_bvh *bvh, const void *points, const void *dirs, int64_t num_rays, int64_t sl,
                               void *counts, int64_t *total_out, void *contacts, void *scratch, size_t scratch_bytes,
                               void *stream) {
    if (!bvh || num_rays < 0) return IBVH_ERR_INVALID_ARG;
    if (int e = check_levels(*bvh, sl)) return (ibvh_status)e;
    if (bvh->types.leaf_float != bvh->types.node_float) return IBVH_ERR_UNSUPPORTED;
    if (num_rays == 0) return IBVH_OK; // :22-26
    if (!points || !dirs || !counts) return IBVH_ERR_INVALID_ARG;
    return (ibvh_status)run<MODE_RAYS>(nullptr, bvh, points, dirs, num_rays, sl, 0, 0, counts, total_out, contacts,
                                       scratch, scratch_bytes, (hipStream_t)stream);
}
ibvh_status ibvh_traverse_rays_lvt_count(const ibvh_bvh *bvh, const void *points, const void *dirs, int64_t num_rays,
                                         int64_t sl, void *counts, int64_t *total_out, void *scratch,
                                         size_t scratch_bytes, void *stream) {
    if (!total_out) return IBVH_ERR_INVALID_ARG;
    *total_out = 0;
    if (num_rays > 0 && !scratch) return IBVH_ERR_INVALID_ARG;
    return rays_common(bvh, points, dirs, num_rays, sl, counts, total_out, nullptr, scratch, scratch_bytes, stream);
}
ibvh_status ibvh_traverse_rays_lvt_write(const ibvh_bvh *bvh, const void *points, const void *dirs, int64_t num_rays,
                                         int64_t sl, const void *counts, void *contacts, void *stream) {
    if (num_rays > 0 && !contacts) return IBVH_ERR_INVALID_ARG;
    int64_t dummy;
    return rays_common(bvh, points, dirs, num_rays, sl, (void *)counts, &dummy, contacts, nullptr, 0, stream);
}

} // extern "C"
