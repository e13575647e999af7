// ibvh_lvt.hpp — what the translation units of the leaf-vs-tree traversal share (LVTTraversal on gfx950): the argument
// block, the per-lane query state, the exact wave-uniform walk (walker 1, also the fall-back of walker 2), the scan of
// the per-item counts, the geometry of the binned ray path, and the launchers each unit exports.
//
//   ibvh_lvt.hip          entry points (extern "C"), type dispatch, the two-pass protocol (count -> scan -> write)
//   ibvh_lvt_queue_*.hip  walker 2, lvt_queue_kernel (BBox nodes: frontier descent + candidate-pair queue), one unit
//                         per mode (self / pair) — ibvh_lvt_queue.inc holds the kernel
//   ibvh_lvt_rays.hip     walker 3, lvt_rays_kernel (per-lane ray walk)
//   ibvh_lvt_raybins.hip  walker 4, rays binned by subtree (rays_top / tilehist / binscan / scatter / subtree / place)
//
// Replaces src/traverse/leaf_vs_tree/traverse_single.jl, traverse_pair.jl and
// src/raytrace/leaf_vs_tree/leaf_vs_tree.jl.  The reference's 32-entry per-thread index stack
// (traverse_single.jl:188-203) is never needed: the tree is implicit, so "the pending right siblings of
// the current path" is one 32-bit mask.
#pragma once
#include <type_traits>

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
    int64_t built_level;       // nodes above it do not exist
    int32_t narrow;    // menu code (IBVH_NARROW_MASK already applied)
    int32_t positions; // IBVH_OUTPUT_POSITIONS: contacts hold 1-based leaf positions instead of user indices
    int32_t flip;
    int32_t xcd_tiles; // work items handed out so that each XCD walks one contiguous range (speed only)
    // outputs
    I *counts;                 // count pass: per-item counts; write pass: inclusive prefix
    IndexPair<I> *contacts;
    // *_enqueue: the writing pass was launched before the host knew the total; it does nothing unless
    // *guard_total <= guard_capacity (guard_total == nullptr: unguarded)
    const int64_t *guard_total;
    int64_t guard_capacity;
    // RAYS: the quantised 8-wide shadow of the node levels (RayShadow below; nullptr: the binary walk) and which rays a
    // launch serves: 0 all, 1 only IRREGULAR ones (a zero / non-finite direction component or a non-finite origin: the
    // slab test is not monotone under box inclusion for them), 2 only regular ones
    const char *shadow;
    int32_t rays_filter;
    // RAYS: the binary walker as the stand-by of the binned path (RayBins below): it returns at once unless *gate != 0
    const int32_t *gate;
    // ibvh_lvt_work_counters only (COUNT instantiations): [0] node tests, [1] leaf tests, [2] node records fetched,
    // [3] leaf records fetched, summed over the launch
    unsigned long long *work;
    // SELF / PAIR, walker 2: the shared part of the descent (BlockRows below).  blk_rows != nullptr: one row per block of
    // 2^blk_shift consecutive work items, written by lvt_block_frontier_kernel in front of the counting pass
    uint32_t *blk_rows;
    int32_t blk_shift;
    const N *q_nodes;    // the nodes of the tree the work items are the leaves of (SELF: == nodes)
    TreeDev q_tree;
    int64_t q_built_level;
    // SELF / PAIR, walker 2: every work item's .index, densely (4 / 8 bytes an item), written by the counting pass for the writing
    // pass — which needs nothing else of a leaf to put a cached pair together, and would otherwise pull every line of the leaf
    // records through the memory system for it (24-byte records: 240 MB at 1e7 leaves).  nullptr: read the leaf.
    I *q_index_dense;
    // walker 2, counting pass: the tile aggregates of the single-kernel scan that follows it (scan_fused_kernel) — its first
    // scan_nparts waves zero one word each (the scratch is the caller's, uninitialised).  nullptr: nothing to zero.
    unsigned long long *scan_agg;
    int32_t scan_nparts;
};

// ---- walker 2, the shared part of the descent ---------------------------------------------------------------------------
// Round 5.  The descent from the start level to the cut level is per-wave fixed cost that the ~16 neighbouring waves of a
// Morton-sorted range repeat almost identically (30 - 36 % of the counting pass, profiles/r04_lvt_sections.json).  So it is
// done ONCE per block of 2^shift consecutive sorted leaves: the block IS a node of the implicit tree the leaves belong to
// (shift levels above them), its box is that node's box, and lvt_block_frontier_kernel — one wave per block, in front of the
// counting pass — descends the walked tree with that one box and leaves the cut-level nodes it touches, in ascending order,
// in the block's row.  A wave of the counting pass then starts at the cut level: it filters its block's row with its own two
// boxes (the existing per-subtree test) instead of descending.  The row is a superset of what the wave's own descent finds
// PROVIDED the block's box contains the wave's queries — true whenever the merges below the block node were exact minima /
// maxima; a NaN anywhere below can be dropped or propagated by merge.jl's `a < b ? a : b`, so every wave CHECKS the
// containment of its valid queries (six compares per lane) and descends on its own when it fails, when the block's list
// overflowed its row, or when there are no rows (small trees, the writing pass, the work-counter instantiation).  Any
// conservative enumeration is exact here (ibvh_lvt.hpp header: box tests are monotone along a root-to-leaf path).
constexpr int BLK_ROW = 512;   // 32-bit words per row: [0] count (-1: no list), [2 .. 2 + sizeof(N) / 4) the block's box, [BLK_HEAD ..) nodes
constexpr int BLK_HEAD = 16;
constexpr int BLK_CAP = BLK_ROW - BLK_HEAD;
constexpr int BLK_FCAP = 512;  // frontier entries of a block's descent per level (LDS)
inline size_t blk_rows_bytes(int64_t n_items, int shift) { return (size_t)ceil_div(n_items > 0 ? n_items : 1, (int64_t)1 << shift) * BLK_ROW * 4; }
constexpr int BLK_SHIFT_MIN = 9; // (the scratch is sized for the smallest block the launch code may choose)

// per-lane work counters of the COUNT instantiations (nothing at all otherwise)
template <bool COUNT> struct Work {
    uint32_t v[4] = {0, 0, 0, 0};
    IBVH_D void add(int k, uint32_t n) {
        if constexpr (COUNT) v[k] += n;
    }
    IBVH_D void flush(unsigned long long *out) {
        if constexpr (COUNT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned long long t = v[k];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
                if ((threadIdx.x & 63) == 0 && t != 0) atomicAdd(out + k, t);
            }
        }
    }
};

IBVH_D int64_t load_total_uniform(const int64_t *p) {
    return *(const __attribute__((address_space(4))) int64_t *)(uintptr_t)p; // one s_load, same value for every lane
}

IBVH_D bool narrow_eval(int narrow, uint64_t ma, int64_t ia, uint64_t mb, int64_t ib) {
    if (narrow == IBVH_NARROW_MORTON_LT) return ma < mb;
    if (narrow == IBVH_NARROW_INDEX_LT) return ia < ib;
    return true;
}

// IBVH_NARROW_RAY_ORIGIN_OUTSIDE: (bv, p, d) -> p lies outside bv.volume (strictly outside the sphere: distance > r;
// outside the box: beyond a face on some axis), evaluated only for leaves the ray already hits
template <class T> IBVH_D bool origin_outside(const BSphere<T> &s, const T *p) {
    return dist3sq(p, s.x) > s.r * s.r;
}
template <class T> IBVH_D bool origin_outside(const BBox<T> &b, const T *p) {
    return (p[0] < b.lo[0]) | (p[0] > b.up[0]) | (p[1] < b.lo[1]) | (p[1] > b.up[1]) | (p[2] < b.lo[2]) | (p[2] > b.up[2]);
}

// ------------------------------------------------------------------------------------------
// Wave-cooperative walks for leaf queries (SELF / PAIR).
//
// The 64 work items of a wave are 64 consecutive Morton-sorted leaves — one compact cluster — so
// their individual walks visit almost the same nodes.  64 divergent walks make every load
// instruction touch 64 different cache lines (the L1/TA-bound regime measured first in round 1:
// 0.86 ms per pass at 1e6 leaves).  Instead the wave works on the UNION of its 64 walks:
//
// (1) lvt_joint_kernel — exact, any node type.  One wave-uniform pre-order walk: the current node,
//     its level and the stack of pending right siblings (a 32-bit mask, possible because the tree is
//     implicit) live in SGPRs; each step scalar-loads BOTH children from one address and every lane
//     tests them against its own query; ballots steer the wave; a lane keeps one bit "active on the
//     current path" and one 32-bit mask "hit the pending sibling of level l".  A lane reaches a leaf
//     iff it hit every ancestor from its start-level root down, exactly the reference's per-leaf walk
//     (traverse_single.jl:157-203), and pre-order visits leaves in increasing position, so each lane
//     emits its contacts in the reference's order.
//
// (2) lvt_queue_kernel — BBox nodes.  BBox parents are the exact min/max of their children
//     (merge.jl:30-40), so box tests are monotone along a root-to-leaf path: a query that touches the
//     box of a leaf's PARENT (level levels-1) touches every ancestor's box.  The reference's walk
//     therefore reports leaf j for query q iff q touches parent(j)'s box and the leaf test passes
//     (plus, for the self walk, j to the right of q): the interior levels only prune, they never
//     change the result, and ANY conservative enumeration of candidates followed by those two exact
//     tests gives the reference's list, provided each query's contacts come out in increasing leaf
//     position.  So:
//       a. frontier descent, lanes = NODES: level by level the wave tests up to 64 frontier nodes at
//          once against two boxes that cover its queries and compacts the children of the hits (ballot
//          + popcount) into the next frontier in LDS — one step per level instead of one per node;
//       b. at the cut level (subtrees of 128 leaves) lane k loads leaf-parent k of each surviving subtree
//          (coalesced) and the wave finds the (query, parent) candidates with the shorter of two loops:
//          over the active queries (broadcast a query box with v_readlane, test all 64 parents at once)
//          or over the parents that touch the wave's boxes (broadcast a parent, test all 64 queries);
//          candidates are appended to a per-wave LDS queue;
//       c. the queue is drained 64 candidates at a time with every lane busy: a lane gathers the two
//          leaves of its candidate parent, runs the exact leaf tests and ranks its hits among the lanes
//          that hold the same query.
//     BSphere nodes (rounded merges, not nested), start_level == levels and trees deeper than 31 levels
//     take kernel (1); trees of 29 .. 31 levels use 64-bit queue entries (WIDE).
//
// Contact cache (K * n_items pairs of scratch).  Kernel (1) stores the first K contacts of every work item
// slot-major (slot k of item i at [k * n_items + i]); its writing pass copies them to their final offsets and only
// waves with an item of more than K contacts walk again.  Kernels (2) and (3) pool the slots of a wave's 64 items
// and fill them densely (see the kernels): their writing pass walks again only if the whole wave overflowed.
// ------------------------------------------------------------------------------------------
template <class I> struct PairCache {
    IndexPair<I> *slots; // K * n_items pairs, slot-major; nullptr when K == 0
    int32_t K;
};

#ifndef IBVH_BRUTE_DEPTH
#define IBVH_BRUTE_DEPTH 7
#endif
constexpr int BRUTE_DEPTH = IBVH_BRUTE_DEPTH; // 2^7 = 128 leaves, 64 leaf-parents (one per lane) per brute-forced subtree
constexpr int FRONTIER_CAP = 256; // frontier entries per wave and level (LDS); overflow -> exact walk

// Per-lane query state + the emission rules shared by both kernels.
template <class L, class N, class I, int MODE, bool WRITE, bool NARROW> struct Query {
    using Cnt = typename std::conditional<sizeof(I) == 8, int64_t, int32_t>::type; // contact counters / offsets
    const Args<L, N, I> &a;
    PairCache<I> cache; // (a copy: the queue kernel's fallback switches the cache off for its wave)
    int64_t item;
    bool valid, lane_on;
    L q_leaf;
    N q_node;
    I q_index;
    uint64_t q_morton;
    Cnt w, cnt;

    // defer_leaf: the leaf is not read here but by load_leaf(), if at all (the queue walker's writing pass, which serves most waves
    // from the cache)
    IBVH_D Query(const Args<L, N, I> &a_, const PairCache<I> &c_, bool defer_leaf = false) : a(a_), cache(c_) {
        // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MiB L2; handing an
        // XCD RUNS of 64 consecutive workgroups (16 K Morton-sorted items) keeps neighbouring waves, which read the same
        // nodes and leaves, behind one L2.  Time-neutral for this issue-bound kernel, but L2-miss traffic drops
        // (rocprofv3 FETCH_SIZE per count launch at 1e6 leaves: 88 MB round robin, 38 MB with runs of 64, 22 MB with ONE
        // contiguous range per XCD).  One range per XCD is not the default because a workload whose cost sits in part
        // of the index range (config 4: two clouds overlapping by 10 %) then loads a few XCDs only (0.37 -> 0.47 ms);
        // runs of 64 keep it at 0.38 ms.
        const int blk = a.xcd_tiles == 1 ? xcd_remap((int)blockIdx.x, (int)gridDim.x)
                        : (a.xcd_tiles > 1 ? xcd_run_remap((int)blockIdx.x, (int)gridDim.x, a.xcd_tiles) : (int)blockIdx.x);
        item = (int64_t)blk * blockDim.x + threadIdx.x;
        valid = item < a.n_items;
        q_leaf = {};
        q_node = {};
        q_index = 0;
        q_morton = 0;
        w = 0;
        cnt = 0;
        if (!defer_leaf) load_leaf();
        lane_on = valid;
    }
    IBVH_D void load_leaf() {
        if (valid) {
            // (SELF: the work items ARE the walked tree's leaves — naming them through a.leaves / a.lay lets the compiler drop
            // a.items / a.items_lay, six scalar registers that would otherwise stay live through the whole kernel)
            const LeafLayout &il = MODE == MODE_SELF ? a.lay : a.items_lay;
            const char *rec = (MODE == MODE_SELF ? a.leaves : a.items) + item * il.stride;
            q_leaf = load_vol<L>(rec);
            q_node = convert_to(q_leaf, (N *)nullptr); // traverse_single.jl:154-155
            q_index = load_index<I>(rec, il);
            if constexpr (NARROW)
                if (a.narrow == IBVH_NARROW_MORTON_LT) q_morton = load_morton(rec, il);
        }
    }
    // WRITE pass: serve the item from the contact cache; returns false when the whole wave is done
    IBVH_D bool begin_write() {
        if (a.guard_total != nullptr && load_total_uniform(a.guard_total) > a.guard_capacity) return false;
        w = (valid && item > 0) ? (Cnt)a.counts[item - 1] : 0;
        const Cnt mine = valid ? (Cnt)a.counts[item] - w : 0;
        const bool over = mine > (Cnt)cache.K;
        if (valid && !over)
            for (Cnt k = 0; k < mine; ++k) a.contacts[(int64_t)w + k] = cache.slots[(int64_t)k * a.n_items + item];
        lane_on = over;
        return __ballot(over) != 0;
    }
    IBVH_D bool narrow_ok(uint64_t lm, I lidx) const {
        return (MODE == MODE_PAIR && a.flip) ? narrow_eval(a.narrow, lm, lidx, q_morton, q_index)
                                             : narrow_eval(a.narrow, q_morton, q_index, lm, lidx);
    }
    IBVH_D void emit(I lidx, int64_t lpos) { // lpos: 0-based position of the leaf in the walked tree's leaves
        IndexPair<I> c2;
        if (a.positions) { // (query, partner) / (bvh1, bvh2) positions, 1-based (include/ibvh.h, IBVH_OUTPUT_POSITIONS)
            const I qp = (I)(item + 1), lp = (I)(lpos + 1);
            c2 = (MODE == MODE_PAIR && a.flip) ? IndexPair<I>{lp, qp} : IndexPair<I>{qp, lp};
        } else if constexpr (MODE == MODE_SELF) c2 = q_index > lidx ? IndexPair<I>{lidx, q_index} : IndexPair<I>{q_index, lidx};
        else c2 = a.flip ? IndexPair<I>{lidx, q_index} : IndexPair<I>{q_index, lidx};
        if constexpr (WRITE) {
            a.contacts[(int64_t)w] = c2;
            ++w;
        } else {
            if (cnt < (Cnt)cache.K) cache.slots[(int64_t)cnt * a.n_items + item] = c2;
            ++cnt;
        }
    }
    IBVH_D void finish() {
        if constexpr (!WRITE)
            if (valid) a.counts[item] = (I)cnt;
    }
};

// ---- (1) exact wave-uniform pre-order walk ------------------------------------------------------
template <class L, class N, class I, int MODE, bool WRITE, bool NARROW>
IBVH_D void joint_walk(Query<L, N, I, MODE, WRITE, NARROW> &q, const Args<L, N, I> &a) {
    const int64_t levels = a.tree.levels, vl = a.tree.virtual_leaves;
    const uint32_t leaf_first = 1u << (levels - 1);
    const uint64_t self_next = (uint64_t)q.item + leaf_first + 1; // SELF: implicit index of this leaf, plus one

    // test one leaf (wave-uniform position, scalar loads) for the lanes in `hit`, emit in place
    auto leaf_step = [&](uint32_t c, bool hit) {
        if constexpr (MODE == MODE_SELF) hit = hit && !((uint64_t)c + 1 <= self_next); // leaves at or left of self
        const char *rec = a.leaves + (int64_t)(c - leaf_first) * a.lay.stride;            // uniform address
        const L leaf = load_vol_uniform<L>(rec);
        hit = hit && iscontact(q.q_leaf, leaf);
        if (__ballot(hit) == 0) return;
        const I lidx = load_index_uniform<I>(rec, a.lay);
        if constexpr (NARROW) {
            const uint64_t lm = a.narrow == IBVH_NARROW_MORTON_LT ? load_morton_uniform(rec, a.lay) : 0;
            hit = hit && q.narrow_ok(lm, lidx);
        }
        if (hit) q.emit(lidx, (int64_t)(c - leaf_first));
    };

    // pseudo-parents: the nodes one level above the start level are entered unconditionally, which
    // tests every start-level root exactly once (the reference's loop over inode_start:inode_end);
    // start_level == 1 uses the pseudo node 0, whose only real child is the root 1.
    const int64_t plevel = a.start_level - 1;
    const int64_t roots = level_num_real(levels, vl, a.start_level);
    const uint32_t pfirst = plevel >= 1 ? (1u << (plevel - 1)) : 0u;
    const uint32_t pcount = (uint32_t)((roots + 1) / 2);

    for (uint32_t pi = 0; pi < pcount; ++pi) {
        uint32_t inode = pfirst + pi; // wave-uniform
        int level = (int)plevel;      // wave-uniform
        uint32_t pend = 0;            // wave-uniform: pending right siblings by level
        uint32_t pendhit = 0;         // per lane: did this lane hit the pending sibling on level l
        bool act = q.lane_on;         // per lane: active on the current path
        while (true) {
            const int cl = level + 1;
            const uint32_t c0 = 2u * inode, c1 = c0 + 1u;
            const bool real0 = c0 != 0u;
            const bool real1 = (int64_t)(c1 - (1u << (cl - 1))) < level_num_real(levels, vl, cl);
            if (cl == levels) {
                // both children are leaves: test and emit, left then right
                if (real0 && __ballot(act) != 0) leaf_step(c0, act);
                if (real1 && __ballot(act) != 0) leaf_step(c1, act);
            } else {
                const int64_t sk = level_skips(levels, vl, cl);
                const N *np = a.nodes + ((int64_t)c0 - sk - 1); // uniform address; c1 follows contiguously
                bool h0 = false, h1 = false;
                if (real0) {
                    h0 = act;
                    if constexpr (MODE == MODE_SELF) h0 = h0 && !(((uint64_t)c0 + 1) <= (self_next >> (levels - cl)));
                    const N n0 = load_vol_uniform<N>(np);
                    h0 = h0 && iscontact(q.q_node, n0);
                }
                if (real1) {
                    h1 = act;
                    if constexpr (MODE == MODE_SELF) h1 = h1 && !(((uint64_t)c1 + 1) <= (self_next >> (levels - cl)));
                    const N n1 = load_vol_uniform<N>(np + 1);
                    h1 = h1 && iscontact(q.q_node, n1);
                }
                const bool go0 = __ballot(h0) != 0;
                const bool go1 = __ballot(h1) != 0;
                if (go0) {
                    if (go1) {
                        pend |= 1u << cl;
                        pendhit = h1 ? (pendhit | (1u << cl)) : (pendhit & ~(1u << cl));
                    }
                    inode = c0;
                    level = cl;
                    act = h0;
                    continue;
                }
                if (go1) {
                    inode = c1;
                    level = cl;
                    act = h1;
                    continue;
                }
            }
            // pop the deepest pending right sibling
            if (pend == 0) break;
            const int pl = 31 - __builtin_clz(pend);
            pend &= ~(1u << pl);
            inode = (inode >> (level - pl)) | 1u;
            level = pl;
            act = (pendhit >> pl) & 1u;
        }
    }
}

template <class L, class N, class I, int MODE, bool WRITE, bool NARROW>
__global__ __launch_bounds__(256) void lvt_joint_kernel(Args<L, N, I> a, PairCache<I> cache) {
    Query<L, N, I, MODE, WRITE, NARROW> q(a, cache);
    if constexpr (WRITE)
        if (!q.begin_write()) return;
    joint_walk(q, a);
    q.finish();
}

// ---- rays: constants and predicates shared by the per-lane walker and the binned path --------------------------------
constexpr int RAY_BITS = 10, RAY_BLOCK_MAX = 1 << RAY_BITS;

// a ray the shadow walk may serve: finite origin, finite non-zero direction with finite reciprocal
template <class T> IBVH_D bool ray_is_regular(const T *p, const T *d, const T *inv) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const T big = float_max<T>();
        ok = ok && (p[k] >= -big && p[k] <= big) && (d[k] >= -big && d[k] <= big) && d[k] != T(0) && (inv[k] >= -big && inv[k] <= big) &&
             inv[k] != T(0);
    }
    return ok;
}

// ---- rays binned by subtree: the tables of one call (kernels in ibvh_lvt_raybins.hip) --------------------------------
struct RayBins {
    unsigned long long *cursor; // header + 0: items emitted so far (may run past cap: that is the overflow test)
    int32_t *flag;              // header + 8: != 0 -> the item list overflowed, the binary walker serves this call
    int32_t *n_items;           // header + 12: min(cursor, cap), written by rays_binscan_kernel
    int32_t *n_chunks;          // header + 16: workgroups of rays_subtree_kernel that have work (rays_binscan_kernel)
    int32_t *reflag;            // header + 20: != 0 -> a region of the hit list overflowed, the writing pass walks the subtrees again
    int32_t *top_nan;           // header + 24: != 0 -> a node of levels 1 .. K holds a NaN (rays_topcheck_kernel): no fast slab test up there
    uint32_t *region_cursor;    // header + 1024: [RAY_REGIONS] records in each region of the hit list
    int64_t *dummy_total;       // header + 64: where the helper scans put their totals
    void *scan_scratch;         // tile sums of the helper scans (room for cap items)
    int32_t *ray_items;         // [rays] items a ray emitted; after the scan: inclusive prefix
    uint64_t *items;            // [cap] emission order: ray | subtree << 32 | ordinal << 48
    uint32_t *bin_count;        // [subtrees]
    uint32_t *bin_start;        // [subtrees + 1] exclusive prefix of bin_count
    uint32_t *bin_cursor;       // [subtrees]
    uint2 *bucket;              // [cap] {ray, g} grouped by subtree
    uint2 *chunk_tab;           // [subtrees + cap / RAYSUB_CHUNK] {subtree, chunk of its bucket}: one workgroup each
    void *hit_list;             // [regions][region_cap] RayHit<I>: the hits of the counting pass
    int32_t region_cap;
    int32_t regions;            // lists in use (a power of two <= RAY_REGIONS): a workgroup moves on to another one after every flush
    void *hits;                 // [cap] of I: hits of item g; after the scan: inclusive prefix
    int32_t cap;                // 0: the path is not in use
    int32_t cut_level;          // K
    int32_t depth;              // D = levels - K: a subtree holds 2^D leaves
    int32_t subtrees;           // real nodes on level K
    int32_t tail_lanes;         // rays_subtree_kernel (counting pass): a wave whose chunk ran dry hands its last <= tail_lanes walks to the workgroup's unit rounds (0: off)
};
constexpr int RAYTILE_IPT = 16;
constexpr int RAYSUB_CHUNK = 4096; // items of one rays_subtree_kernel workgroup: busy subtrees are shared by several (1,024: 2 % slower on config 3)
constexpr int RAYSUB_TPB = 256;
// waves of a workgroup that WALK (all of them load).  A wave lives as long as its longest item (config 3: 12 steps on
// average, ~150 for the longest of a bucket), so four walkers with 256 items each keep only ~20 % of their lanes busy — but
// fewer walkers lose more to latency than they gain in lane use (config 3, subtree pass: 1.47 ms with four, 2.09 ms with one
// per 512 items, 2.47 ms with one)
constexpr int RAYSUB_WALKERS = 4;
constexpr int RAYSUB_STAGE = 64;  // hit records a wave stages in LDS (a step adds at most 64 left and 64 right hits: two appends)
// The tail of a workgroup (round 6): walks still alive when their wave's chunk has run dry and <= tail_lanes lanes are busy are
// PARKED — at most RAYSUB_TAIL_MAX a wave — and finished by the whole workgroup as UNITS (item, node whose children are to be
// tested): a walk's pending right siblings are independent subtrees, so a long walk is no longer one chain of dependent steps.
constexpr int RAYSUB_TAIL_MAX = 8;                                    // parked walks per wave
constexpr int RAYSUB_TAIL_ITEMS = RAYSUB_TAIL_MAX * RAYSUB_WALKERS;   // per workgroup
constexpr int RAYSUB_TAIL_UNITS = 208;                                // units per round (two lists in the flushed hit stages, beside masks and item table)
constexpr int RAY_REGIONS = 256;  // the hit list is at most RAY_REGIONS lists with a cursor each: same-address atomics serialise
// a hit of the counting pass: the pair as it will be reported, the item it belongs to and its rank within the item; the
// writing pass puts it at scan[g - 1] + k (rays_place_kernel) instead of walking again
template <class I> struct RayHit {
    IndexPair<I> pair;
    uint32_t g, k;
};
constexpr uint32_t RAY_HIT_NONE = 0xffffffffu; // RayHit::g of a slot nobody filled (the unused end of a full list: rays_place_kernel skips it)
IBVH_HD size_t rays_subtree_lds(int depth, size_t node_bytes, size_t leaf_bytes, size_t index_bytes, size_t hit_bytes, bool write) {
    const size_t S = (size_t)1 << depth;
    size_t o = (S * node_bytes + 15) & ~(size_t)15;
    o += (S * leaf_bytes + 15) & ~(size_t)15;
    o += (S * index_bytes + 15) & ~(size_t)15;
    if (!write) o += (size_t)RAYSUB_WALKERS * RAYSUB_STAGE * hit_bytes;
    return o;
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

// (limit != nullptr: the array's length is min(n, *limit), known only on the device)
template <class I> __global__ __launch_bounds__(SCAN_TPB) void scan_reduce_kernel(const I *c, int64_t n, int64_t *partials, const int32_t *limit) {
    __shared__ int64_t s_w[SCAN_TPB / 64];
    if (limit != nullptr) n = (int64_t)*limit < n ? (int64_t)*limit : n;
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
// Every workgroup derives its own tile offset from the raw tile sums (a redundant reduction of <= a few thousand
// values) instead of waiting for a single-workgroup scan launch in between; the last tile also publishes the total.
template <class I>
__global__ __launch_bounds__(SCAN_TPB) void scan_apply_kernel(I *c, int64_t n, const int64_t *partials, int64_t *totals,
                                                              int64_t *total_host, const int32_t *limit) {
    __shared__ int64_t s_w[SCAN_TPB / 64], s_p[SCAN_TPB / 64];
    if (limit != nullptr) n = (int64_t)*limit < n ? (int64_t)*limit : n;
    int64_t before = 0;
    for (int64_t j = threadIdx.x; j < (int64_t)blockIdx.x; j += SCAN_TPB) before += partials[j];
    const int64_t tile_offset = block_sum(before, s_p);
    // The grand total is known to the last workgroup before it scans anything (the tile sums are all there): publish it
    // FIRST — the host may be polling its pinned copy (total_host), and every microsecond it learns the count earlier is
    // a microsecond more of the next step's launch work hidden behind this step's writing pass.
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const int64_t total = tile_offset + partials[blockIdx.x];
        totals[0] = total;
        if (total_host) __hip_atomic_store(total_host, total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // thread owns SCAN_IPT consecutive items so the in-thread running sum is in memory order
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_IPT;
    int64_t v[SCAN_IPT], sum = 0;
    // a thread's SCAN_IPT items are 64 (or 128) contiguous bytes: 16-byte loads and stores when the array allows it
    // (one 4-byte access per item makes every load instruction of a wave touch 64 different lines)
    constexpr int NV = SCAN_IPT * (int)sizeof(I) / 16;
    const bool vec = base + SCAN_IPT <= n && ((uintptr_t)c & 15) == 0;
    if (vec) {
        I raw[SCAN_IPT];
        const uint4 *src = (const uint4 *)(c + base);
#pragma unroll
        for (int k = 0; k < NV; ++k) ((uint4 *)raw)[k] = src[k];
#pragma unroll
        for (int j = 0; j < SCAN_IPT; ++j) {
            v[j] = (int64_t)raw[j];
            sum += v[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_IPT; ++j) {
            int64_t i = base + j;
            v[j] = i < n ? (int64_t)c[i] : 0;
            sum += v[j];
        }
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
    int64_t run = tile_offset + wb + inc - sum;
    if (vec) {
        I raw[SCAN_IPT];
#pragma unroll
        for (int j = 0; j < SCAN_IPT; ++j) {
            run += v[j];
            raw[j] = (I)run;
        }
        uint4 *dst = (uint4 *)(c + base);
#pragma unroll
        for (int k = 0; k < NV; ++k) dst[k] = ((const uint4 *)raw)[k];
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_IPT; ++j) {
            int64_t i = base + j;
            run += v[j];
            if (i < n) c[i] = (I)run;
        }
    }
}

// The same scan in ONE kernel (round 5), for counts whose producer zeroed the tile aggregates (walker 2's counting pass): every
// workgroup sums its tile, PUBLISHES the sum (bit 63 = "there"; one 64-bit agent-scope atomic store: value and flag travel together,
// no fence — an agent-scope fence on this part writes back and invalidates an XCD's whole L2), adds up the aggregates of the tiles
// before it (polling those that are not there yet) and scans its tile.  One launch and one dependent round trip less than reduce +
// apply: they are launch- and latency-bound (245 workgroups at 1e6 leaves).
// Whom a workgroup may wait for: only workgroups that are RUNNING OR DONE, whatever order the hardware starts them
// in — the grid never exceeds what the device holds at once (scan_counts: resident_scan_workgroups()).  Larger inputs go through
// scan_fused_grouped_kernel, whose workgroups own several consecutive tiles each: one aggregate and one look-back per group.
// (Round 6 first took the tile from an atomic ticket instead: 2,442 returning atomics on one word serialise at ~11 ns each —
// the 1e7-item scan 25 -> 56 us.)
template <class I>
__global__ __launch_bounds__(SCAN_TPB) void scan_fused_kernel(I *c, int64_t n, unsigned long long *agg, int64_t *totals, int64_t *total_host,
                                                              const int32_t *limit) {
    __shared__ int64_t s_w[SCAN_TPB / 64], s_p[SCAN_TPB / 64];
    constexpr unsigned long long THERE = 1ull << 63;
    // (limit: the array's length is min(n, *limit), known only on the device — tiles beyond it hold zeros and store nothing)
    if (limit != nullptr) n = (int64_t)*limit < n ? (int64_t)*limit : n;
    const uint32_t tile = blockIdx.x;
    // thread owns SCAN_IPT consecutive items so the in-thread running sum is in memory order
    const int64_t base = (int64_t)tile * SCAN_TILE + (int64_t)threadIdx.x * SCAN_IPT;
    int64_t v[SCAN_IPT], sum = 0;
    constexpr int NV = SCAN_IPT * (int)sizeof(I) / 16;
    const bool vec = base + SCAN_IPT <= n && ((uintptr_t)c & 15) == 0;
    if (vec) {
        I raw[SCAN_IPT];
        const uint4 *src = (const uint4 *)(c + base);
#pragma unroll
        for (int k = 0; k < NV; ++k) ((uint4 *)raw)[k] = src[k];
#pragma unroll
        for (int j = 0; j < SCAN_IPT; ++j) {
            v[j] = (int64_t)raw[j];
            sum += v[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_IPT; ++j) {
            const int64_t i = base + j;
            v[j] = i < n ? (int64_t)c[i] : 0;
            sum += v[j];
        }
    }
    int64_t inc = sum;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int64_t wb = 0, tile_total = 0;
#pragma unroll
    for (int k = 0; k < SCAN_TPB / 64; ++k) {
        if (k < w) wb += s_w[k];
        tile_total += s_w[k];
    }
    if (threadIdx.x == 0) __hip_atomic_store(&agg[tile], THERE | (unsigned long long)tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int64_t before = 0;
    for (int64_t j = threadIdx.x; j < (int64_t)tile; j += SCAN_TPB) {
        unsigned long long a = __hip_atomic_load(&agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (!(a & THERE)) {
            __builtin_amdgcn_s_sleep(1);
            a = __hip_atomic_load(&agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        before += (int64_t)(a & ~THERE);
    }
    const int64_t tile_offset = block_sum(before, s_p);
    if (tile == gridDim.x - 1 && threadIdx.x == 0) {
        const int64_t total = tile_offset + tile_total;
        totals[0] = total;
        if (total_host) __hip_atomic_store(total_host, total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    int64_t run = tile_offset + wb + inc - sum;
    if (vec) {
        I raw[SCAN_IPT];
#pragma unroll
        for (int j = 0; j < SCAN_IPT; ++j) {
            run += v[j];
            raw[j] = (I)run;
        }
        uint4 *dst = (uint4 *)(c + base);
#pragma unroll
        for (int k = 0; k < NV; ++k) dst[k] = ((const uint4 *)raw)[k];
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_IPT; ++j) {
            const int64_t i = base + j;
            run += v[j];
            if (i < n) c[i] = (I)run;
        }
    }
}

// The same scan for grids that would not be resident: a workgroup owns `tiles_per_group` CONSECUTIVE tiles.  Phase 1 sums them
// (one pass over its items), publishes ONE aggregate and looks back over the groups before it; phase 2 reads the items again
// (L2-hot) and scans tile by tile with a running base.  Grid = ceil(tiles / tiles_per_group) <= what the device holds at once.
template <class I>
__global__ __launch_bounds__(SCAN_TPB) void scan_fused_grouped_kernel(I *c, int64_t n, unsigned long long *agg, int64_t *totals, int64_t *total_host,
                                                                      int tiles_per_group, const int32_t *limit) {
    __shared__ int64_t s_w[SCAN_TPB / 64], s_p[SCAN_TPB / 64];
    constexpr unsigned long long THERE = 1ull << 63;
    const int64_t nparts = (n + SCAN_TILE - 1) / SCAN_TILE; // (of the launch: the grid was sized for it)
    if (limit != nullptr) n = (int64_t)*limit < n ? (int64_t)*limit : n;
    const int64_t t0 = (int64_t)blockIdx.x * tiles_per_group, t1 = t0 + tiles_per_group < nparts ? t0 + tiles_per_group : nparts;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    auto load16 = [&](int64_t base, int64_t (&v)[SCAN_IPT]) { // a thread's SCAN_IPT consecutive items of a tile
        constexpr int NV = SCAN_IPT * (int)sizeof(I) / 16;
        if (base + SCAN_IPT <= n && ((uintptr_t)c & 15) == 0) {
            I raw[SCAN_IPT];
            const uint4 *src = (const uint4 *)(c + base);
#pragma unroll
            for (int k = 0; k < NV; ++k) ((uint4 *)raw)[k] = src[k];
#pragma unroll
            for (int j = 0; j < SCAN_IPT; ++j) v[j] = (int64_t)raw[j];
        } else {
#pragma unroll
            for (int j = 0; j < SCAN_IPT; ++j) v[j] = base + j < n ? (int64_t)c[base + j] : 0;
        }
    };
    int64_t sum = 0;
    for (int64_t t = t0; t < t1; ++t) {
        int64_t v[SCAN_IPT];
        load16(t * SCAN_TILE + (int64_t)threadIdx.x * SCAN_IPT, v);
#pragma unroll
        for (int j = 0; j < SCAN_IPT; ++j) sum += v[j];
    }
    const int64_t group_total = block_sum(sum, s_p);
    if (threadIdx.x == 0) __hip_atomic_store(&agg[blockIdx.x], THERE | (unsigned long long)group_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int64_t before = 0;
    for (int64_t j = threadIdx.x; j < (int64_t)blockIdx.x; j += SCAN_TPB) {
        unsigned long long a = __hip_atomic_load(&agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (!(a & THERE)) {
            __builtin_amdgcn_s_sleep(1);
            a = __hip_atomic_load(&agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        before += (int64_t)(a & ~THERE);
    }
    __syncthreads(); // (s_p is reused)
    int64_t run_base = block_sum(before, s_p);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const int64_t total = run_base + group_total;
        totals[0] = total;
        if (total_host) __hip_atomic_store(total_host, total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int64_t t = t0; t < t1; ++t) {
        const int64_t base = t * SCAN_TILE + (int64_t)threadIdx.x * SCAN_IPT;
        int64_t v[SCAN_IPT], mine = 0;
        load16(base, v);
#pragma unroll
        for (int j = 0; j < SCAN_IPT; ++j) mine += v[j];
        int64_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t u = __shfl_up(inc, o, 64);
            if (lane >= o) inc += u;
        }
        __syncthreads(); // (s_w is reused)
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int64_t wb = 0, tile_total = 0;
#pragma unroll
        for (int k = 0; k < SCAN_TPB / 64; ++k) {
            if (k < w) wb += s_w[k];
            tile_total += s_w[k];
        }
        int64_t run = run_base + wb + inc - mine;
        const bool vec = base + SCAN_IPT <= n && ((uintptr_t)c & 15) == 0;
        if (vec) {
            constexpr int NV = SCAN_IPT * (int)sizeof(I) / 16;
            I raw[SCAN_IPT];
#pragma unroll
            for (int j = 0; j < SCAN_IPT; ++j) {
                run += v[j];
                raw[j] = (I)run;
            }
            uint4 *dst = (uint4 *)(c + base);
#pragma unroll
            for (int k = 0; k < NV; ++k) dst[k] = ((const uint4 *)raw)[k];
        } else {
#pragma unroll
            for (int j = 0; j < SCAN_IPT; ++j) {
                run += v[j];
                if (base + j < n) c[base + j] = (I)run;
            }
        }
        run_base += tile_total;
    }
}
// workgroups of SCAN_TPB threads the current device holds at once, halved (the margin for anything else that is running)
template <class I> inline int64_t resident_scan_workgroups() {
    static thread_local int memo_dev = -1;
    static thread_local int64_t memo = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != memo_dev) {
        int ncu = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 64;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)scan_fused_grouped_kernel<I>, SCAN_TPB, 0) != hipSuccess || per_cu <= 0) per_cu = 2;
        memo = (int64_t)ncu * per_cu / 2;
        if (memo < 16) memo = 16;
        memo_dev = dev;
    }
    return memo;
}

// scratch layout of the *_count / *_write calls:
//   [0, 64)            int64 header: [0] total contacts, [1] contact-cache slots K in use
//   [64, scan_bytes)   scan tile sums
//   [scan_bytes, ...)  contact cache: K * n_items IndexPair{I}, slot-major
inline size_t scan_scratch_bytes(int64_t n) {
    return (size_t)align_up((ceil_div(n > 0 ? n : 1, SCAN_TILE) + 9) * 8, 256); // header, one aggregate per tile, the fused scan's ticket
}
constexpr int MAX_CACHE_SLOTS = 64;
inline int cache_slots_for(size_t scratch_bytes, int64_t n_items, int64_t pair_bytes) {
    size_t sb = scan_scratch_bytes(n_items);
    if (scratch_bytes <= sb || n_items <= 0) return 0;
    int64_t k = (int64_t)((scratch_bytes - sb) / ((size_t)n_items * (size_t)pair_bytes));
    return (int)(k > MAX_CACHE_SLOTS ? MAX_CACHE_SLOTS : k);
}

// inclusive scan in place + (total_out != nullptr) blocking read of the total (the reference's @allowscalar, :60)
template <class I>
int scan_counts(I *counts, int64_t n, int64_t *total_out, void *scratch, hipStream_t st, int64_t *total_dev = nullptr,
                int64_t *total_host = nullptr, const int32_t *limit = nullptr, bool aggregates_zeroed = false) {
    int64_t nparts = ceil_div(n, SCAN_TILE);
    int64_t *totals = total_dev ? total_dev : (int64_t *)scratch; // where the device-side total goes
    int64_t *partials = (int64_t *)scratch + 8;
    if (aggregates_zeroed && g_tuning.lvt_scan_fused != 0) {
        int64_t room = resident_scan_workgroups<I>();
        if (g_tuning.lvt_scan_fused > 1 && g_tuning.lvt_scan_fused < room) room = g_tuning.lvt_scan_fused; // (development knob: a smaller grid)
        if (nparts > room) {
            const int64_t per = ceil_div(nparts, room);
            IBVH_LAUNCH((scan_fused_grouped_kernel<I>), dim3((unsigned)ceil_div(nparts, per)), dim3(SCAN_TPB), 0, st, counts, n, (unsigned long long *)partials, totals,
                        total_host, (int)per, limit);
        } else {
            IBVH_LAUNCH((scan_fused_kernel<I>), dim3((unsigned)nparts), dim3(SCAN_TPB), 0, st, counts, n, (unsigned long long *)partials, totals, total_host,
                        limit);
        }
    } else {
        IBVH_LAUNCH((scan_reduce_kernel<I>), dim3((unsigned)nparts), dim3(SCAN_TPB), 0, st, counts, n, partials, limit);
        IBVH_LAUNCH((scan_apply_kernel<I>), dim3((unsigned)nparts), dim3(SCAN_TPB), 0, st, counts, n, partials, totals, total_host, limit);
    }
    IBVH_LAUNCH_CHECK();
    if (!total_out) return IBVH_OK; // *_enqueue: the total stays in the scratch header, nobody waits
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

// the instantiations ibvh_lvt_work_counters may ask for: the bench types only
template <class L, class N, class I> constexpr bool kWorkTypes =
    std::is_same<L, BSphere<float>>::value && std::is_same<N, BBox<float>>::value && std::is_same<I, int32_t>::value;

// Geometry of the binned ray path (kernels (3c)) for a tree and a batch of rays; depth == 0: the binary walker serves it.
// The region lives at the END of the caller's scratch (ibvh_rays_scratch_bytes adds its size), offsets below are inside it.
struct RayBinPlan {
    int depth = 0, cut_level = 0, subtrees = 0;
    int64_t cap = 0;
    size_t bytes = 0, off_scan = 0, off_ray_items = 0, off_bin_count = 0, off_bin_start = 0, off_bin_cursor = 0, off_items = 0,
           off_bucket = 0, off_hits = 0, off_chunks = 0, off_hit_list = 0;
    int64_t region_cap = 0;
    int regions = RAY_REGIONS;
};
inline RayBinPlan rays_bin_plan(const ibvh_bvh &bvh, int64_t num_rays) {
    RayBinPlan p;
    const int mode = g_tuning.rays_binned; // 1 = where it pays, 2 = wherever the tree allows it (tests), 0 = never
    // (the tables take 40 bytes x 16 items per ray: batches beyond 8 M rays — 5.4 GB — are left to the per-lane walker)
    if (mode == 0 || num_rays <= 0 || num_rays > (int64_t)1 << 23) return p;
    if (bvh.types.leaf_float != bvh.types.node_float) return p; // (isintersection needs one float type: the entry points refuse the rest)
    ibvh_layout lay;
    if (!layout_of(bvh.types, lay)) return p;
    const int levels = (int)bvh.tree.levels;
    if (levels < 3 || levels > 32) return p;
    // 512-leaf subtrees: 26 KB of LDS a workgroup, six workgroups a CU (config 3: 1.47 ms for the subtree pass; 1,024 leaves, three
    // workgroups a CU: 2.50 ms; 256 leaves: 1.49 ms with a longer top walk); Float64 records are twice as wide: 256 leaves
    int D = g_tuning.rays_subtree_depth > 0 ? g_tuning.rays_subtree_depth : (bvh.types.leaf_float == IBVH_F64 ? 8 : 9);
    if (D > 11) D = 11;
    const size_t index_bytes = bvh.types.index_type == IBVH_I64 ? 8 : 4;
    while (D > 1 && rays_subtree_lds(D, (size_t)lay.node_bytes, (size_t)lay.volume_bytes, index_bytes, index_bytes * 2 + 8, false) > 144 * 1024) --D; // (the CU's LDS)
    const bool small_batch = num_rays <= 8192;
    if (mode == 1) {
        // subtrees of >= 64 leaves; enough of them to fill the chip (>= ~1,000: the cut at level 11 or below) unless the batch
        // is small anyway (then the walk is a chain of dependent fetches and cutting it pays on any tree: 1,000 rays on 3,200 /
        // 20,000 / 45,000 / 7.2 M leaves: 0.29 -> 0.18, 0.42 -> 0.20, 0.49 -> 0.22, 1.49 -> 0.51 ms; 64 rays on 7.2 M: 0.86 -> 0.35)
        const int k_min = small_batch ? 7 : 11;
        if (D > levels - k_min) D = levels - k_min;
        if (D < 6) return p;
        // a SMALL tree under MANY rays stays with the per-lane walker: it lives in L2 and the binning is pure overhead
        // (45 k leaves, 1e6 rays: 1.62 ms against 2.17 binned; 250 k leaves, 1e6 rays: 2.42 / 2.61; but 250 k, 1e5: 1.04 / 0.71)
        if (!small_batch && bvh.tree.real_leaves < ((int64_t)1 << 20) && num_rays > 2 * bvh.tree.real_leaves) return p;
    } else if (D > levels - 2) {
        D = levels - 2;
    }
    int K = levels - D;
    if (K < (int)bvh.built_level) { // the nodes above built_level do not exist
        K = (int)bvh.built_level;
        D = levels - K;
        if (D < 1) return p;
    }
    int64_t subtrees = level_num_real(bvh.tree.levels, bvh.tree.virtual_leaves, K);
    while (subtrees > 16384 && D < 11 && K - 1 >= (int)bvh.built_level) { // (one LDS counter per subtree in the binning kernels)
        ++D;
        --K;
        subtrees = level_num_real(bvh.tree.levels, bvh.tree.virtual_leaves, K);
    }
    if (subtrees > 16384) return p;
    // (few rays are no reason to stay away: a subtree nobody reaches is never loaded — 7.2 M-leaf mesh, 3e4 rays: 0.78 ms
    // against 1.89 ms for the binary walker, 1e5 rays: 1.01 / 2.29, 3e5: 1.52 / 2.52)
    const int per_ray = g_tuning.rays_items_per_ray > 0 ? g_tuning.rays_items_per_ray : 16;
    int64_t cap = num_rays * per_ray;
    if (cap > ((int64_t)1 << 30)) cap = (int64_t)1 << 30;
    p.depth = D;
    p.cut_level = K;
    p.subtrees = (int)subtrees;
    p.cap = cap;
    size_t o = 2048; // header
    p.off_scan = o, o += scan_scratch_bytes(cap > num_rays ? cap : num_rays);
    p.off_ray_items = o, o += (size_t)align_up(4 * num_rays, 256);
    p.off_bin_count = o, o += (size_t)align_up(4 * (subtrees + 1), 256);
    p.off_bin_start = o, o += (size_t)align_up(4 * (subtrees + 1), 256);
    p.off_bin_cursor = o, o += (size_t)align_up(4 * (subtrees + 1), 256);
    p.off_items = o, o += (size_t)cap * 8;
    p.off_bucket = o, o += (size_t)cap * 8;
    p.off_hits = o, o += (size_t)cap * 8;
    p.off_chunks = o, o += (size_t)align_up(8 * (subtrees + cap / RAYSUB_CHUNK + 1), 256);
    // As many records as items all together, in `regions` lists with a cursor each (same-address atomics serialise).  A list
    // should take at least 8,192 records — a workgroup's flush is a few hundred — so small batches get fewer lists (round 6: with
    // 256 lists for every batch, 1e5 rays left 6,250 records a list, one list overflowed and the writing pass walked every
    // subtree again; a 1,000-ray batch had 63 records a list and ALWAYS walked twice).
    while (p.regions > 1 && cap / p.regions < 8192) p.regions >>= 1;
    p.region_cap = (cap + p.regions - 1) / p.regions;
    p.off_hit_list = o, o += (size_t)p.region_cap * (size_t)p.regions * (bvh.types.index_type == IBVH_I64 ? 24 : 16);
    p.bytes = o;
    return p;
}
inline RayBins rays_bins_at(const RayBinPlan &p, char *base) {
    RayBins rb{};
    rb.cursor = (unsigned long long *)base;
    rb.flag = (int32_t *)(base + 8);
    rb.n_items = (int32_t *)(base + 12);
    rb.n_chunks = (int32_t *)(base + 16);
    rb.reflag = (int32_t *)(base + 20);
    rb.top_nan = (int32_t *)(base + 24);
    rb.region_cursor = (uint32_t *)(base + 1024);
    rb.dummy_total = (int64_t *)(base + 64);
    rb.scan_scratch = base + p.off_scan;
    rb.ray_items = (int32_t *)(base + p.off_ray_items);
    rb.bin_count = (uint32_t *)(base + p.off_bin_count);
    rb.bin_start = (uint32_t *)(base + p.off_bin_start);
    rb.bin_cursor = (uint32_t *)(base + p.off_bin_cursor);
    rb.items = (uint64_t *)(base + p.off_items);
    rb.bucket = (uint2 *)(base + p.off_bucket);
    rb.hits = base + p.off_hits;
    rb.chunk_tab = (uint2 *)(base + p.off_chunks);
    rb.hit_list = base + p.off_hit_list;
    rb.region_cap = (int32_t)p.region_cap;
    rb.regions = p.regions;
    rb.cap = (int32_t)p.cap;
    rb.cut_level = p.cut_level;
    rb.depth = p.depth;
    rb.tail_lanes = (p.depth <= 9 && g_tuning.rays_tail > 0) ? (g_tuning.rays_tail < RAYSUB_TAIL_MAX ? g_tuning.rays_tail : RAYSUB_TAIL_MAX) : 0;
    rb.subtrees = p.subtrees;
    return rb;
}
// the type combinations the binned path is compiled for (one float type throughout; everything else: the binary walker)
template <class L, class N> constexpr bool kRayBinTypes = std::is_same<typename L::elt, typename N::elt>::value; // (what ray traversal asks for anyway)

// ---- launchers, one per walker, each defined (and explicitly instantiated for every type combination the dispatch can
// reach: IBVH_FOR_* below) in its own translation unit ------------------------------------------------------------------
// walker 2: a.start_level < a.tree.levels <= 31, BBox nodes (ibvh_lvt_queue_self.hip / ibvh_lvt_queue_pair.hip)
template <class L, class N, class I, int MODE>
int launch_queue(const Args<L, N, I> &a, const PairCache<I> &cache, bool write, hipStream_t st, bool *agg_zeroed = nullptr);
// walkers 3 and 4: the whole ray traversal of one pass (ibvh_lvt_rays.hip; it hands over to launch_rays_binned when rb.cap > 0)
template <class L, class N, class I>
int launch_rays(const Args<L, N, I> &a, const PairCache<I> &cache, bool write, hipStream_t st, const RayBins &rb);
// walker 3 as the stand-by of the binned path: gated on a.gate, no contact cache (ibvh_lvt_rays.hip)
template <class L, class N, class I>
int launch_rays_standby(const Args<L, N, I> &standby, bool write, hipStream_t st, int ray_block, unsigned rblocks);
size_t rays_shadow_bytes(const ibvh_bvh &bvh, int64_t num_rays); // (ibvh_lvt_rays.hip; 0 in the product library)
// walker 4 (ibvh_lvt_raybins.hip)
template <class L, class N, class I>
int launch_rays_binned(const Args<L, N, I> &a, bool write, hipStream_t st, const RayBins &rb, int ray_block, unsigned rblocks);

// The (leaf, node, index) combinations dispatch_leaf_node / dispatch_index reach (ibvh_common.hpp): X(L, N, I, extra...)
#ifdef IBVH_ONLY_BENCH_TYPES
#define IBVH_FOR_INDEX(X, L_, N_, ...) X(L_, N_, int32_t, __VA_ARGS__)
#define IBVH_FOR_BBOX_NODE_COMBOS(X, ...)                        \
    IBVH_FOR_INDEX(X, BSphere<float>, BBox<float>, __VA_ARGS__) \
    IBVH_FOR_INDEX(X, BBox<float>, BBox<float>, __VA_ARGS__)
#define IBVH_FOR_SAME_FLOAT_COMBOS(X, ...)                          \
    IBVH_FOR_INDEX(X, BSphere<float>, BSphere<float>, __VA_ARGS__) \
    IBVH_FOR_BBOX_NODE_COMBOS(X, __VA_ARGS__)
#else
#define IBVH_FOR_INDEX(X, L_, N_, ...) X(L_, N_, int32_t, __VA_ARGS__) X(L_, N_, int64_t, __VA_ARGS__)
#define IBVH_FOR_BBOX_NODE_COMBOS_F32(X, ...)                     \
    IBVH_FOR_INDEX(X, BSphere<float>, BBox<float>, __VA_ARGS__)  \
    IBVH_FOR_INDEX(X, BSphere<double>, BBox<float>, __VA_ARGS__) \
    IBVH_FOR_INDEX(X, BBox<float>, BBox<float>, __VA_ARGS__)     \
    IBVH_FOR_INDEX(X, BBox<double>, BBox<float>, __VA_ARGS__)
#define IBVH_FOR_BBOX_NODE_COMBOS_F64(X, ...)                      \
    IBVH_FOR_INDEX(X, BSphere<float>, BBox<double>, __VA_ARGS__)  \
    IBVH_FOR_INDEX(X, BSphere<double>, BBox<double>, __VA_ARGS__) \
    IBVH_FOR_INDEX(X, BBox<float>, BBox<double>, __VA_ARGS__)     \
    IBVH_FOR_INDEX(X, BBox<double>, BBox<double>, __VA_ARGS__)
#define IBVH_FOR_BBOX_NODE_COMBOS(X, ...) IBVH_FOR_BBOX_NODE_COMBOS_F32(X, __VA_ARGS__) IBVH_FOR_BBOX_NODE_COMBOS_F64(X, __VA_ARGS__)
#define IBVH_FOR_SAME_FLOAT_COMBOS(X, ...)                            \
    IBVH_FOR_INDEX(X, BSphere<float>, BSphere<float>, __VA_ARGS__)   \
    IBVH_FOR_INDEX(X, BSphere<float>, BBox<float>, __VA_ARGS__)      \
    IBVH_FOR_INDEX(X, BBox<float>, BBox<float>, __VA_ARGS__)         \
    IBVH_FOR_INDEX(X, BSphere<double>, BSphere<double>, __VA_ARGS__) \
    IBVH_FOR_INDEX(X, BSphere<double>, BBox<double>, __VA_ARGS__)    \
    IBVH_FOR_INDEX(X, BBox<double>, BBox<double>, __VA_ARGS__)
#endif

} // namespace lvt
} // namespace ibvh
