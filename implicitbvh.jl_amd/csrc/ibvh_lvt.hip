// ibvh_lvt.hip — leaf-vs-tree traversal (LVTTraversal) on gfx950: one work item per leaf / ray walks
// the (same / other) implicit tree depth-first; two passes (count -> inclusive scan -> write) give
// the reference's deterministic contact order.
//
// Replaces src/traverse/leaf_vs_tree/traverse_single.jl, traverse_pair.jl and
// src/raytrace/leaf_vs_tree/leaf_vs_tree.jl.
//
// Three walkers, documented where they are defined: lvt_queue_kernel (BBox nodes: frontier descent +
// candidate-pair queue), lvt_joint_kernel (exact wave-uniform pre-order walk, any node type) and
// lvt_rays_kernel (per-lane walk with a bitmask stack).  The reference's 32-entry per-thread index stack
// (traverse_single.jl:188-203) is never needed: the tree is implicit, so "the pending right siblings of
// the current path" is one 32-bit mask.
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
};

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

    IBVH_D Query(const Args<L, N, I> &a_, const PairCache<I> &c_) : a(a_), cache(c_) {
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
        lane_on = valid;
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

// ---- (2) BBox nodes: frontier descent + brute-forced subtrees -----------------------------------
template <class T> IBVH_D T wave_min_all(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T t = __shfl_xor(v, o, 64);
        v = v < t ? v : t;
    }
    return v;
}
template <class T> IBVH_D T wave_max_all(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T t = __shfl_xor(v, o, 64);
        v = v > t ? v : t;
    }
    return v;
}

// Inclusive prefix min / max over the 64 lanes of a wave with DPP (lane k: min / max of lanes 0 .. k): four row_shr steps
// inside each row of 16 lanes, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 — six VALU
// instructions per value (the DPP operand rides on the v_min / v_max itself) where __shfl_up costs a ds_bpermute, its
// address arithmetic and a compare-select per step.  Lanes without a valid source are write-disabled and keep their own
// value.  Inputs are finite (the callers sanitise NaN), so v_min_f32 / v_max_f32 are exact here; all 64 lanes are active.
// Written as ONE assembly block over twelve independent values (six minima, six maxima), step by step across all of
// them: a DPP operand must not have been written by one of the two preceding VALU instructions, and with eleven other
// instructions between two steps of the same value no s_nop is needed.  (Through __builtin_amdgcn_update_dpp the compiler
// emits v_mov_b32_dpp + canonicalising v_max + v_min + a copy per step: four instructions instead of one.)
#define IBVH_DPP_STEP(CTRL)                              \
    "v_min_f32_dpp %0, %0, %0 " CTRL "\n\t"              \
    "v_min_f32_dpp %1, %1, %1 " CTRL "\n\t"              \
    "v_min_f32_dpp %2, %2, %2 " CTRL "\n\t"              \
    "v_min_f32_dpp %3, %3, %3 " CTRL "\n\t"              \
    "v_min_f32_dpp %4, %4, %4 " CTRL "\n\t"              \
    "v_min_f32_dpp %5, %5, %5 " CTRL "\n\t"              \
    "v_max_f32_dpp %6, %6, %6 " CTRL "\n\t"              \
    "v_max_f32_dpp %7, %7, %7 " CTRL "\n\t"              \
    "v_max_f32_dpp %8, %8, %8 " CTRL "\n\t"              \
    "v_max_f32_dpp %9, %9, %9 " CTRL "\n\t"              \
    "v_max_f32_dpp %10, %10, %10 " CTRL "\n\t"           \
    "v_max_f32_dpp %11, %11, %11 " CTRL "\n\t"
IBVH_D void wave_prefix_scans_dpp(float (&mn)[6], float (&mx)[6]) {
    asm volatile("s_nop 1\n\t" IBVH_DPP_STEP("row_shr:1 row_mask:0xf bank_mask:0xf") IBVH_DPP_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
                     IBVH_DPP_STEP("row_shr:4 row_mask:0xf bank_mask:0xf") IBVH_DPP_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
                         IBVH_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") IBVH_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(mn[0]), "+v"(mn[1]), "+v"(mn[2]), "+v"(mn[3]), "+v"(mn[4]), "+v"(mn[5]), "+v"(mx[0]), "+v"(mx[1]), "+v"(mx[2]),
                   "+v"(mx[3]), "+v"(mx[4]), "+v"(mx[5]));
}
#undef IBVH_DPP_STEP
// wave minimum of one value (every lane of the result's lane 63 holds it; read with v_readlane): one dependent chain,
// so each step waits out the DPP hazard with an s_nop
IBVH_D float wave_min_dpp_lane63(float v) {
    asm volatile("s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(v));
    return v;
}

// ---- (2) BBox nodes: frontier descent + candidate-pair queue ----------------------------------------
// Stage c in detail: each lane gathers the two leaves of its pair (48 contiguous bytes for BSphere{Float32}
// records), runs the exact leaf tests and ranks its hits among the lanes that hold the same query (6-ballot
// match-any on the query lane id), so the contacts of a query still come out in increasing leaf position:
// queue order is (subtree, then parent) ascending for any fixed query.  The per-query counters / output
// offsets live in LDS.  History (round 1, 1e6 random spheres, count pass): per-lane walks 0.86 ms; exact joint
// walk 0.72; subtrees brute-forced with per-lane candidate masks and one leaf-test round per candidate of the
// busiest lane (~22 % of the lanes busy) 0.40; this kernel with ONE union box per wave 0.40 (the kernel waited
// for the few waves whose 64 leaves straddle a big Z-curve jump: 489 subtrees against 23 on average); with the
// two-box split 0.25.
// Stage c's inner step for Float32 boxes, hand-scheduled: test the wave-uniform box S (SGPRs) against every lane's
// box V (iscontact: S.lo <= V.up and S.up >= V.lo per axis — the same ordered compares as the C++ operators), plus
// one unsigned compare (the self walk's "a leaf to the right of the query" prune), starting from the lanes in
// `init`; the surviving lanes append their entry `e` to the LDS queue at lds_base + 4 * (number of surviving lanes
// below).  v_cmpx narrows EXEC directly, so the chain needs no s_and per compare and the append runs under the
// result mask without a saveexec — 17 VALU + 4 SALU where the compiler's version took 21 + 14.  Returns the mask.
// Round 3: the step also advances the queue's LDS byte address itself (s_bcnt1 + s_lshl2_add: the caller no longer keeps
// a count that has to be shifted and added to a base every time), EXEC is restored to all-ones instead of being saved (the
// kernel's control flow is wave-uniform: all 64 lanes are active wherever this is called), and the self walk's prune in the
// parent-major loop compares the parent index itself (p >= (item + 1) >> 1  <=>  2p + 1 > item): 30 instructions per loop
// iteration where round 2 took 42 — and the loops are bound by the scalar instructions around the seven v_cmpx
// (count pass 0.155 -> 0.146 ms at 1e6 for the first five of them, measured).
// PRUNE = false drops the seventh compare: the pair walk has no prune, and the self walk needs none in a subtree that lies
// entirely to the right of the wave's own leaves (most of them) — one vector instruction of eleven per iteration, and an
// iteration's vector instructions are what the pass is bound by (six more of them per parent-major iteration, v_readlane
// instead of the scalar load: 125 -> 141 us, measured).
#define IBVH_TEST_AND_APPEND(SEVENTH)                                                                                     \
    asm volatile("s_mov_b64 exec, %[init]\n\t"                                                                           \
                 "v_cmpx_le_f32 %[slo0], %[vup0]\n\t"                                                                    \
                 "v_cmpx_ge_f32 %[sup0], %[vlo0]\n\t"                                                                    \
                 "v_cmpx_le_f32 %[slo1], %[vup1]\n\t"                                                                    \
                 "v_cmpx_ge_f32 %[sup1], %[vlo1]\n\t"                                                                    \
                 "v_cmpx_le_f32 %[slo2], %[vup2]\n\t"                                                                    \
                 "v_cmpx_ge_f32 %[sup2], %[vlo2]\n\t" SEVENTH "s_nop 2\n\t"                                             \
                 "v_mbcnt_lo_u32_b32 %[tmp], exec_lo, 0\n\t"                                                             \
                 "v_mbcnt_hi_u32_b32 %[tmp], exec_hi, %[tmp]\n\t"                                                        \
                 "v_lshl_add_u32 %[tmp], %[tmp], 2, %[addr]\n\t"                                                         \
                 "ds_write_b32 %[tmp], %[e]\n\t"                                                                         \
                 "s_bcnt1_i32_b64 %[cnt], exec\n\t"                                                                      \
                 "s_mov_b64 exec, -1\n\t"                                                                                \
                 "s_lshl2_add_u32 %[addr], %[cnt], %[addr]"                                                              \
                 : [tmp] "=&v"(tmp), [cnt] "=&s"(cnt), [addr] "+s"(lds_addr)                                             \
                 : [init] "s"(init), [slo0] "s"(slo0), [slo1] "s"(slo1), [slo2] "s"(slo2), [sup0] "s"(sup0), [sup1] "s"(sup1), \
                   [sup2] "s"(sup2), [vlo0] "v"(vlo0), [vlo1] "v"(vlo1), [vlo2] "v"(vlo2), [vup0] "v"(vup0), [vup1] "v"(vup1), \
                   [vup2] "v"(vup2), [sthr] "s"(sthr), [vcmp] "v"(vcmp), [e] "v"(e)                                      \
                 : "vcc", "scc", "memory")
template <bool THR_GE, bool PRUNE = true>
IBVH_D void test_and_append_f32(uint64_t init, float slo0, float slo1, float slo2, float sup0, float sup1, float sup2, float vlo0,
                                 float vlo1, float vlo2, float vup0, float vup1, float vup2, uint32_t sthr, uint32_t vcmp, uint32_t e,
                                 uint32_t &lds_addr) {
    uint32_t tmp, cnt;
    if constexpr (!PRUNE) IBVH_TEST_AND_APPEND("");
    else if constexpr (THR_GE) IBVH_TEST_AND_APPEND("v_cmpx_ge_u32 %[sthr], %[vcmp]\n\t");
    else IBVH_TEST_AND_APPEND("v_cmpx_lt_u32 %[sthr], %[vcmp]\n\t");
}
#undef IBVH_TEST_AND_APPEND

// `init` & iscontact(S, V) for the 64 lanes at once, S wave-uniform (scalar registers), V per lane: six v_cmpx narrow EXEC
// from `init` — 6 VALU + 3 SALU where the compiler's six v_cmp into SGPR pairs need five s_and on top, and the result is a
// scalar mask straight away (a ballot of a bool that crossed a branch is re-materialised with v_cndmask + v_cmp_ne).
// All 64 lanes are active at every call site (wave-uniform control flow): EXEC is restored to all-ones.
IBVH_D uint64_t contact_mask_f32(uint64_t init, const BBox<float> &S, const BBox<float> &V) {
    uint64_t m;
    asm volatile("s_mov_b64 exec, %[init]\n\t"
                 "v_cmpx_le_f32 %[slo0], %[vup0]\n\t"
                 "v_cmpx_ge_f32 %[sup0], %[vlo0]\n\t"
                 "v_cmpx_le_f32 %[slo1], %[vup1]\n\t"
                 "v_cmpx_ge_f32 %[sup1], %[vlo1]\n\t"
                 "v_cmpx_le_f32 %[slo2], %[vup2]\n\t"
                 "v_cmpx_ge_f32 %[sup2], %[vlo2]\n\t"
                 "s_mov_b64 %[m], exec\n\t"
                 "s_mov_b64 exec, -1"
                 : [m] "=&s"(m)
                 : [init] "s"(init), [slo0] "s"(S.lo[0]), [slo1] "s"(S.lo[1]), [slo2] "s"(S.lo[2]), [sup0] "s"(S.up[0]), [sup1] "s"(S.up[1]),
                   [sup2] "s"(S.up[2]), [vlo0] "v"(V.lo[0]), [vlo1] "v"(V.lo[1]), [vlo2] "v"(V.lo[2]), [vup0] "v"(V.up[0]), [vup1] "v"(V.up[1]),
                   [vup2] "v"(V.up[2])
                 : "vcc");
    return m;
}

// `init` & (iscontact(A, V) | iscontact(B, V)): the wave's two query boxes against a per-lane box, in one block (the
// second chain starts from `init` again; the masks are OR-ed on the scalar unit)
IBVH_D uint64_t contact_mask2_f32(uint64_t init, const BBox<float> &A, const BBox<float> &B, const BBox<float> &V) {
    uint64_t m, ma;
    asm volatile("s_mov_b64 exec, %[init]\n\t"
                 "v_cmpx_le_f32 %[alo0], %[vup0]\n\t"
                 "v_cmpx_ge_f32 %[aup0], %[vlo0]\n\t"
                 "v_cmpx_le_f32 %[alo1], %[vup1]\n\t"
                 "v_cmpx_ge_f32 %[aup1], %[vlo1]\n\t"
                 "v_cmpx_le_f32 %[alo2], %[vup2]\n\t"
                 "v_cmpx_ge_f32 %[aup2], %[vlo2]\n\t"
                 "s_mov_b64 %[ma], exec\n\t"
                 "s_mov_b64 exec, %[init]\n\t"
                 "v_cmpx_le_f32 %[blo0], %[vup0]\n\t"
                 "v_cmpx_ge_f32 %[bup0], %[vlo0]\n\t"
                 "v_cmpx_le_f32 %[blo1], %[vup1]\n\t"
                 "v_cmpx_ge_f32 %[bup1], %[vlo1]\n\t"
                 "v_cmpx_le_f32 %[blo2], %[vup2]\n\t"
                 "v_cmpx_ge_f32 %[bup2], %[vlo2]\n\t"
                 "s_or_b64 %[m], %[ma], exec\n\t"
                 "s_mov_b64 exec, -1"
                 : [m] "=&s"(m), [ma] "=&s"(ma)
                 : [init] "s"(init), [alo0] "s"(A.lo[0]), [alo1] "s"(A.lo[1]), [alo2] "s"(A.lo[2]), [aup0] "s"(A.up[0]), [aup1] "s"(A.up[1]),
                   [aup2] "s"(A.up[2]), [blo0] "s"(B.lo[0]), [blo1] "s"(B.lo[1]), [blo2] "s"(B.lo[2]), [bup0] "s"(B.up[0]), [bup1] "s"(B.up[1]),
                   [bup2] "s"(B.up[2]), [vlo0] "v"(V.lo[0]), [vlo1] "v"(V.lo[1]), [vlo2] "v"(V.lo[2]), [vup0] "v"(V.up[0]), [vup1] "v"(V.up[1]),
                   [vup2] "v"(V.up[2])
                 : "vcc", "scc");
    return m;
}

// Diagnostic build only (-DIBVH_PHASE_STAMPS, tools/phase_stamps.sh + tools/lvt_stamps.py): s_memtime ticks a wave of
// lvt_queue_kernel spends per section, summed over the launch's waves into a buffer no product code reads.  Coarse: a
// lap is a scalar memory round trip itself, and the per-subtree sections take ~70 of them per wave.
#ifdef IBVH_PHASE_STAMPS
__device__ unsigned long long g_lvt_ticks[8];
struct Sections {
    unsigned long long t0, acc[6] = {0, 0, 0, 0, 0, 0};
    IBVH_D void start() { t0 = __builtin_amdgcn_s_memtime(); }
    IBVH_D void lap(int k) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        acc[k] += t - t0;
        t0 = t;
    }
    IBVH_D void flush() {
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) atomicAdd(&g_lvt_ticks[k], acc[k]);
            atomicAdd(&g_lvt_ticks[7], 1ull);
        }
    }
};
#else
struct Sections {
    IBVH_D void start() {}
    IBVH_D void lap(int) {}
    IBVH_D void flush() {}
};
#endif
enum { SEC_PROLOGUE = 0, SEC_DESCENT = 1, SEC_SUBTREE = 2, SEC_LOOPS = 3, SEC_LEAVES = 4, SEC_EPILOGUE = 5 };

#ifndef IBVH_QUEUE_CAP
#define IBVH_QUEUE_CAP 512
#endif
constexpr int QUEUE_CAP = IBVH_QUEUE_CAP; // candidate pairs per wave (LDS); drained whenever fewer than 64 slots are free
#ifndef IBVH_QUEUE_WAVES
#define IBVH_QUEUE_WAVES 1
#endif
constexpr int QUEUE_WAVES = IBVH_QUEUE_WAVES; // waves per workgroup (they share nothing: a workgroup is only a unit of dispatch)
// Waves per SIMD the register allocator has to leave room for.  Round 3, measured on MI355X with the DPP prologue and
// the straight-line loads below (count pass, 1e6 / 1e7 leaves): 8 waves (64 VGPRs: 36 SGPR + 6 VGPR spills, 28 B of
// scratch per lane) 0.188 / 1.75 ms; 7 waves (70 VGPRs, 15 SGPR spills to VGPR lanes, NO scratch) 0.156 / 1.37 ms; the
// round-2 kernel at 8 waves (31 + 4 spills, 20 B of scratch) 0.165 / 1.42 ms.  (profiles/r03_lvt_variants.txt)
#ifndef IBVH_QUEUE_MINWAVES
#define IBVH_QUEUE_MINWAVES 7
#endif
#ifndef IBVH_LVT_QTABLE
#define IBVH_LVT_QTABLE 0
#endif
#ifndef IBVH_LVT_STRAIGHT
#define IBVH_LVT_STRAIGHT 7 // bit 0: descent loads, bit 1: leaf-parent loads, bit 2: leaf loads of the pair step — straight-line (clamped) instead of exec-masked
#endif
constexpr int QUEUE_MINWAVES = IBVH_QUEUE_MINWAVES; // waves per SIMD the register allocator has to leave room for (8: 64 VGPRs)

// Waves per SIMD a given instantiation can actually reach: Float64 volumes and 64-bit queue entries need more registers than
// the bench types, and asking for 7 waves there only makes the allocator spill and warn (-Wpass-failed, 48 times in round 3).
template <class L, class N, class I, bool WIDE, bool WRITE = false> constexpr int queue_min_waves() {
    if (sizeof(typename N::elt) == 8) return 4;
    if (sizeof(typename L::elt) == 8) return 5;
    if (WIDE && sizeof(I) == 8) return 6;
    // (the writing pass puts pairs together from 8-byte cache entries: a few registers more, one wave per SIMD fewer — it
    // is a streaming pass of 0.013 ms at 1e6 leaves)
    return WRITE ? QUEUE_MINWAVES - 1 : QUEUE_MINWAVES;
}
// WIDE: 64-bit queue entries for trees of 29 .. 31 levels (leaf-parent indices beyond 2^26), see launch().
template <class L, class N, class I, int MODE, bool WRITE, bool NARROW, bool WIDE, bool COUNT = false>
__global__ __launch_bounds__(64 * QUEUE_WAVES, (queue_min_waves<L, N, I, WIDE, WRITE>())) void lvt_queue_kernel(Args<L, N, I> a, PairCache<I> cache, int cut_level) {
    using TN = typename N::elt;
    Work<COUNT> work; // (COUNT: one lane-level box / sphere test = one count; lane 0 carries the wave-uniform parts)
    using Q = Query<L, N, I, MODE, WRITE, NARROW>;
    using Cnt = typename Q::Cnt;
    __shared__ uint32_t s_frontier[QUEUE_WAVES][2][FRONTIER_CAP];
    using QE = typename std::conditional<WIDE, uint64_t, uint32_t>::type; // queue entry: query lane | leaf-parent index << 6
    __shared__ QE s_queue[QUEUE_WAVES][QUEUE_CAP];
    __shared__ Cnt s_cnt[QUEUE_WAVES][64];
    __shared__ I s_qside[QUEUE_WAVES][64]; // writing pass from the cache: the query's half of a pair, by query lane
#if IBVH_LVT_QTABLE
    // The wave's 64 query leaves (volume, index) in LDS: the leaf-test step fetches its candidate's query with one or two
    // ds_read instead of five ds_bpermute out of registers, and the volume / index need not stay in VGPRs through the loops.
    struct QRec {
        L vol;
        I idx;
    };
    __shared__ QRec s_query[QUEUE_WAVES][64];
#endif
    Sections sec;
    sec.start();
    Q q(a, cache);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (uniform: LDS bases stay scalar)
#if IBVH_LVT_QTABLE
    s_query[wv][lane] = QRec{q.q_leaf, q.q_index};
#endif
    // Wave-dense contact cache.  The wave owns the scratch bytes its 64 items own in the slot-major layout of the
    // other walkers ([item0 * K, (item0 + 64) * K) pairs) but fills them densely, in discovery order, with
    // (partner, query lane | position within that query's list << 6) entries behind a 16-byte header {fill}: the
    // counting pass writes them with coalesced stores instead of one scattered 8-byte store per contact, and the
    // writing pass reads ~10 B per contact instead of touching K sparse slot arrays (measured at 1e7 leaves:
    // 0.72 GB fetched by the writing pass with the slot-major cache).  fill < 0: the wave found more contacts than
    // fit (or fell back to the exact walk) and walks again in the writing pass.
    // Round 4: an entry is 8 bytes, not 12 — the QUERY's half of the pair is a function of the query lane, which the
    // writing pass has anyway (it loads its 64 items), so only the partner's half (its index, or its 1-based position
    // with IBVH_OUTPUT_POSITIONS) is kept and the pair is put together when it is written out.
    struct Entry {
        I other;
        I meta;
    };
    const int64_t first_item = q.item - lane;
    const int64_t items_here = a.n_items - first_item < 64 ? a.n_items - first_item : 64;
    char *region = cache.K > 0 && items_here > 0 ? (char *)(cache.slots + first_item * (int64_t)cache.K) : nullptr;
    const int entry_cap = region ? (int)(((int64_t)items_here * cache.K * (int64_t)sizeof(IndexPair<I>) - 16) / (int64_t)sizeof(Entry)) : 0;
    Entry *entries = (Entry *)(region + 16);
    int wfill = 0; // wave-uniform: entries appended so far (may run past entry_cap: then nothing more is stored)
    if constexpr (WRITE) {
        if (a.guard_total != nullptr && load_total_uniform(a.guard_total) > a.guard_capacity) return;
        q.w = (q.valid && q.item > 0) ? (Cnt)a.counts[q.item - 1] : 0;
        const int fill = region ? __builtin_amdgcn_readfirstlane(*(const int *)region) : -1;
        if (fill >= 0) {
            // serve the whole wave from its cache: entry t goes to (prefix of its query) + (its position in the list)
            s_cnt[wv][lane] = q.w;
            s_qside[wv][lane] = a.positions ? (I)(q.item + 1) : q.q_index;
            __builtin_amdgcn_wave_barrier();
            for (int t = lane; t < fill; t += 64) {
                const Entry e = entries[t];
                const int qi = (int)(e.meta & 63);
                const int64_t dest = (int64_t)s_cnt[wv][qi] + (int64_t)(e.meta >> 6);
                const I qv = s_qside[wv][qi];
                IndexPair<I> c2; // (the same rules as put() below)
                if (a.positions) c2 = (MODE == MODE_PAIR && a.flip) ? IndexPair<I>{e.other, qv} : IndexPair<I>{qv, e.other};
                else if constexpr (MODE == MODE_SELF) c2 = qv > e.other ? IndexPair<I>{e.other, qv} : IndexPair<I>{qv, e.other};
                else c2 = a.flip ? IndexPair<I>{e.other, qv} : IndexPair<I>{qv, e.other};
                a.contacts[dest] = c2;
            }
            return;
        }
        q.lane_on = q.valid; // every item of the wave walks again
    }
    // Pair walk: a wave none of whose queries touches the other tree's ROOT box finds nothing — leave before the
    // two-box split and the descent (two partially overlapping clouds: most waves of the larger one).  The root
    // exists only in a fully built tree.
    if constexpr (MODE == MODE_PAIR) {
        if (a.built_level <= 1 && a.tree.levels >= 2) {
            const N root = load_vol_uniform<N>(a.nodes);
            work.add(0, q.lane_on);
            work.add(2, lane == 0);
            if (__builtin_amdgcn_ballot_w64(q.lane_on & iscontact(q.q_node, root)) == 0) {
                work.flush(a.work);
                if constexpr (!WRITE) {
                    if (q.valid) a.counts[q.item] = (I)0;
                    if (region && lane == 0) *(int *)region = 0;
                }
                return;
            }
        }
    }

    // Everything wave-uniform below is 32-bit on purpose (levels <= 31, so node indices and leaf positions stay
    // below 2^31): the scalar unit has no ordered 64-bit compare, a 64-bit uniform compare is done by the VALU, its
    // result counts as divergent and turns every loop that depends on it into an exec-masked one.
    const int levels = (int)a.tree.levels;
    const uint32_t vl = (uint32_t)a.tree.virtual_leaves;
    auto num_real = [&](int level) -> uint32_t { return (1u << (level - 1)) - (vl >> (levels - level)); };
    auto first_mem = [&](int level) -> uint32_t { // 0-based memory index of the level's first node
        const uint32_t v = vl >> (levels - (level - 1));
        return (1u << (level - 1)) - (2u * v - (uint32_t)__builtin_popcount(v)) - 1u;
    };
    const uint32_t leaf_first = 1u << (levels - 1);
    const uint32_t my_item = (uint32_t)q.item;
    const uint32_t wave_item0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)my_item);
    const uint32_t self_next = my_item + leaf_first + 1u;
    const uint32_t wave_next = wave_item0 + leaf_first + 1u;
    QE *queue = s_queue[wv];
    Cnt *cnts = s_cnt[wv];
    cnts[lane] = WRITE ? q.w : (Cnt)0; // next output offset (WRITE) / contacts so far (count pass) of query `lane`
    int qn = 0;                        // wave-uniform: queued pairs
    const uint64_t lane_on_mask = __builtin_amdgcn_ballot_w64(q.lane_on); // (fixed from here on)

    // Two boxes instead of one union box: 64 consecutive Morton-sorted leaves regularly straddle a big jump of
    // the Z-curve, and the single union box of such a wave spans a large part of the scene (measured at 1e6
    // random spheres: 23 cut-level subtrees per wave on average, 489 for the worst wave — and the kernel waits
    // for the worst wave).  The wave splits its queries at the lane k that minimises the half-area sum of
    // box[0..k] and box[k+1..63] (prefix / suffix min-max scans over the lanes); with the split the worst wave
    // sees 41 subtrees, the average one 18.  Any split is valid: the two boxes only have to cover the queries.
    N ubox_a, ubox_b;
    {
        const TN big = float_max<TN>();
        bool use = q.lane_on; // NaN boxes touch nothing and must not poison the min / max
#pragma unroll
        for (int k = 0; k < 3; ++k) use = use && q.q_node.lo[k] == q.q_node.lo[k] && q.q_node.up[k] == q.q_node.up[k];
        N pre, nxt_suf; // box of lanes 0 .. lane / of lanes lane+1 .. 63
        float cost;
        auto half_area = [](const N &b) {
            float d[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d[k] = (float)b.up[k] - (float)b.lo[k];
                d[k] = d[k] > 0.0f ? d[k] : 0.0f; // empty (or NaN) -> 0
            }
            return d[0] * d[1] + d[1] * d[2] + d[0] * d[2];
        };
        int ksplit;
        if constexpr (std::is_same<TN, float>::value) {
            // DPP scans (wave_prefix_dpp).  The suffix boxes come from the same prefix scan run on the lane-reversed
            // values: rsuf in lane j = box of lanes 63-j .. 63, so the box of lanes k+1 .. 63 sits in lane 62-k; only its
            // half-area has to travel back (one ds_bpermute), and the chosen boxes are read with v_readlane.
            N rev, rsuf;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                pre.lo[k] = use ? q.q_node.lo[k] : big;
                pre.up[k] = use ? q.q_node.up[k] : -big;
                rev.lo[k] = __shfl(pre.lo[k], 63 - lane, 64);
                rev.up[k] = __shfl(pre.up[k], 63 - lane, 64);
            }
            {
                float mn[6] = {pre.lo[0], pre.lo[1], pre.lo[2], rev.lo[0], rev.lo[1], rev.lo[2]};
                float mx[6] = {pre.up[0], pre.up[1], pre.up[2], rev.up[0], rev.up[1], rev.up[2]};
                wave_prefix_scans_dpp(mn, mx);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    pre.lo[k] = mn[k];
                    pre.up[k] = mx[k];
                    rsuf.lo[k] = mn[3 + k];
                    rsuf.up[k] = mx[3 + k];
                }
            }
            // cost(k) = area(lanes 0 .. k) + area(lanes k+1 .. 63); the latter is rsuf's area in lane 62-k (nothing for k = 63)
            const float ra = half_area(rsuf);
            float sa = __shfl(ra, 62 - lane, 64);
            sa = lane == 63 ? 0.0f : sa;
            cost = half_area(pre) + sa;
            cost = cost == cost ? cost : __builtin_inff();
            // argmin: wave min of the cost with the same DPP steps, then the first lane that attains it
            const float m = wave_min_dpp_lane63(cost);
            const float best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
            const uint64_t at = __builtin_amdgcn_ballot_w64(cost == best);
            ksplit = at != 0 ? (int)__builtin_ctzll(at) : 0;
            ubox_a = broadcast_from_lane(pre, ksplit);
            ubox_b = broadcast_from_lane(rsuf, ksplit == 63 ? 0 : 62 - ksplit);
            if (ksplit == 63) { // nothing to the right of the split: the empty box
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    ubox_b.lo[k] = big;
                    ubox_b.up[k] = -big;
                }
            }
        } else {
            N suf;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                pre.lo[k] = suf.lo[k] = use ? q.q_node.lo[k] : big;
                pre.up[k] = suf.up[k] = use ? q.q_node.up[k] : -big;
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { // lanes without a source keep their own value: min / max are idempotent
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    TN t = __shfl_up(pre.lo[k], o, 64);
                    pre.lo[k] = pre.lo[k] < t ? pre.lo[k] : t;
                    t = __shfl_up(pre.up[k], o, 64);
                    pre.up[k] = pre.up[k] > t ? pre.up[k] : t;
                    t = __shfl_down(suf.lo[k], o, 64);
                    suf.lo[k] = suf.lo[k] < t ? suf.lo[k] : t;
                    t = __shfl_down(suf.up[k], o, 64);
                    suf.up[k] = suf.up[k] > t ? suf.up[k] : t;
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                nxt_suf.lo[k] = __shfl_down(suf.lo[k], 1, 64);
                nxt_suf.up[k] = __shfl_down(suf.up[k], 1, 64);
                if (lane == 63) {
                    nxt_suf.lo[k] = big;
                    nxt_suf.up[k] = -big;
                }
            }
            cost = half_area(pre) + half_area(nxt_suf);
            cost = cost == cost ? cost : __builtin_inff();
            // argmin over the lanes: non-negative floats order like their bit patterns
            uint64_t key = ((uint64_t)__float_as_uint(cost) << 32) | (uint32_t)lane;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint64_t t = (uint64_t)__shfl_xor((long long)key, o, 64);
                key = t < key ? t : key;
            }
            ksplit = (int)(key & 63u);
            ubox_a = broadcast_from_lane(pre, ksplit);
            ubox_b = broadcast_from_lane(nxt_suf, ksplit);
        }
    }
    auto touches_wave = [&](const N &b) { return (bool)((int)iscontact(ubox_a, b) | (int)iscontact(ubox_b, b)); };
    const int lp = levels - 1;
    const uint32_t lp_real = num_real(lp);
    const N *lp_nodes = a.nodes + first_mem(lp);
    const uint32_t n_leaves = (uint32_t)a.tree.real_leaves;
    const uint64_t lt_mask = ((uint64_t)1 << lane) - 1;

    // c: leaf tests of queue[off, off + avail), one pair per lane
    auto pair_step = [&](int off, int avail) {
        const bool v = lane < avail;
        // (straight-line loads: lanes beyond the step re-read its first entry, a pair without a right leaf re-reads the left
        // one; both are masked by v / has_b afterwards)
#if IBVH_LVT_STRAIGHT & 4
        const QE e = queue[off + (v ? lane : 0)];
        const int qi = (int)(e & 63u);
        const uint32_t pos = 2u * (uint32_t)(e >> 6); // 0-based position of the pair's left leaf
        const bool has_b = v & (pos + 1u < n_leaves);
        uint64_t mor_a = 0, mor_b = 0;
        const char *rec = a.leaves + (int64_t)pos * a.lay.stride;
        const char *rec_b = (pos + 1u < n_leaves) ? rec + a.lay.stride : rec;
        const L leaf_a = load_vol<L>(rec), leaf_b = load_vol<L>(rec_b);
        const I idx_a = load_index<I>(rec, a.lay), idx_b = load_index<I>(rec_b, a.lay);
        if constexpr (NARROW) {
            if (a.narrow == IBVH_NARROW_MORTON_LT) {
                mor_a = load_morton(rec, a.lay);
                mor_b = load_morton(rec_b, a.lay);
            }
        }
#else
        const QE e = v ? queue[off + lane] : (QE)0;
        const int qi = (int)(e & 63u);
        const uint32_t pos = 2u * (uint32_t)(e >> 6); // 0-based position of the pair's left leaf
        const bool has_b = v & (pos + 1u < n_leaves);
        L leaf_a = {}, leaf_b = {};
        I idx_a = 0, idx_b = 0;
        uint64_t mor_a = 0, mor_b = 0;
        const char *rec = a.leaves + (int64_t)pos * a.lay.stride;
        if (v) {
            leaf_a = load_vol<L>(rec);
            idx_a = load_index<I>(rec, a.lay);
            if constexpr (NARROW)
                if (a.narrow == IBVH_NARROW_MORTON_LT) mor_a = load_morton(rec, a.lay);
        }
        if (has_b) {
            leaf_b = load_vol<L>(rec + a.lay.stride);
            idx_b = load_index<I>(rec + a.lay.stride, a.lay);
            if constexpr (NARROW)
                if (a.narrow == IBVH_NARROW_MORTON_LT) mor_b = load_morton(rec + a.lay.stride, a.lay);
        }
#endif
#if IBVH_LVT_QTABLE
        const QRec qr = s_query[wv][qi];
        const L ql = qr.vol;
        const I qidx = qr.idx;
#else
        const L ql = shuffle_from(q.q_leaf, qi);
        const I qidx = __shfl(q.q_index, qi, 64);
#endif
        const uint32_t item_q = wave_item0 + (uint32_t)qi;
        bool hit_a = v & iscontact(ql, leaf_a), hit_b = has_b & iscontact(ql, leaf_b);
        work.add(1, (uint32_t)v + (uint32_t)has_b);
        work.add(3, (uint32_t)v + (uint32_t)has_b);
        if constexpr (MODE == MODE_SELF) { // only partners to the right of the query
            hit_a = hit_a & (pos > item_q);
            hit_b = hit_b & (pos + 1u > item_q);
        }
        if constexpr (NARROW) {
            const uint64_t qm = (uint64_t)__shfl((long long)q.q_morton, qi, 64);
            const bool fl = MODE == MODE_PAIR && a.flip;
            hit_a = hit_a && (fl ? narrow_eval(a.narrow, mor_a, idx_a, qm, qidx) : narrow_eval(a.narrow, qm, qidx, mor_a, idx_a));
            hit_b = hit_b && (fl ? narrow_eval(a.narrow, mor_b, idx_b, qm, qidx) : narrow_eval(a.narrow, qm, qidx, mor_b, idx_b));
        }
        // lanes holding the same query (match-any on the 6-bit lane id).  Written on the 32-bit halves with the
        // three-input boolean op — same &= ~(ballot(bit) ^ -bit), table 0x90 = a & ~(b ^ c) — and with v_mbcnt as "popcount
        // below this lane": the compiler's 64-bit per-lane version of this block was 90 VALU instructions, this is 40.
        uint32_t same_lo, same_hi;
        {
            const uint64_t vm = avail >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << avail) - 1); // = ballot(v), scalar
            same_lo = (uint32_t)vm;
            same_hi = (uint32_t)(vm >> 32);
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int x = (int)((uint32_t)qi << (31 - b)) >> 31; // -bit
            const uint64_t m = __builtin_amdgcn_ballot_w64(x != 0);
            same_lo = __builtin_amdgcn_bitop3_b32(same_lo, (uint32_t)m, (uint32_t)x, 0x90);
            same_hi = __builtin_amdgcn_bitop3_b32(same_hi, (uint32_t)(m >> 32), (uint32_t)x, 0x90);
        }
        const uint64_t m_a = __builtin_amdgcn_ballot_w64(hit_a), m_b = __builtin_amdgcn_ballot_w64(hit_b);
        const uint32_t sa_lo = same_lo & (uint32_t)m_a, sa_hi = same_hi & (uint32_t)(m_a >> 32);
        const uint32_t sb_lo = same_lo & (uint32_t)m_b, sb_hi = same_hi & (uint32_t)(m_b >> 32);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi(sb_hi, __builtin_amdgcn_mbcnt_lo(sb_lo, __builtin_amdgcn_mbcnt_hi(sa_hi, __builtin_amdgcn_mbcnt_lo(sa_lo, 0u))));
        const int tot = __builtin_popcount(sa_lo) + __builtin_popcount(sa_hi) + __builtin_popcount(sb_lo) + __builtin_popcount(sb_hi);
        const bool group_first = __builtin_amdgcn_mbcnt_hi(same_hi, __builtin_amdgcn_mbcnt_lo(same_lo, 0u)) == 0u;
        const Cnt base = cnts[qi];
        __builtin_amdgcn_wave_barrier();
        if (v && group_first && tot > 0) cnts[qi] = base + (Cnt)tot;
        __builtin_amdgcn_wave_barrier();
        // WRITE: straight to the output; counting pass: appended to the wave's dense cache (slot = running fill +
        // number of hitting lanes below this one, a-hits of the step before its b-hits)
        auto put = [&](Cnt at, I lidx, int slot, uint32_t lpos) {
            if constexpr (WRITE) {
                IndexPair<I> c2;
                if (a.positions) { // 1-based positions, query / bvh1 first (include/ibvh.h, IBVH_OUTPUT_POSITIONS)
                    const I qp = (I)(item_q + 1u), lp = (I)(lpos + 1u);
                    c2 = (MODE == MODE_PAIR && a.flip) ? IndexPair<I>{lp, qp} : IndexPair<I>{qp, lp};
                } else if constexpr (MODE == MODE_SELF) c2 = qidx > lidx ? IndexPair<I>{lidx, qidx} : IndexPair<I>{qidx, lidx};
                else c2 = a.flip ? IndexPair<I>{lidx, qidx} : IndexPair<I>{qidx, lidx};
                a.contacts[(int64_t)at] = c2;
            } else {
                if (slot < entry_cap) entries[slot] = Entry{a.positions ? (I)(lpos + 1u) : lidx, (I)((I)qi | ((I)(at - 0) << 6))};
            }
        };
        const Cnt at = base + (Cnt)rank;
        const int n_a = __popcll(m_a);
        auto below = [](uint64_t m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
        if (hit_a) put(at, idx_a, wfill + below(m_a), pos);
        if (hit_b) put(at + (hit_a ? 1 : 0), idx_b, wfill + n_a + below(m_b), pos + 1u);
        if constexpr (!WRITE) wfill = __builtin_amdgcn_readfirstlane(wfill + n_a + __popcll(m_b));
    };
    // drain the full 64-pair steps (all == false) or everything (all == true); a remainder moves to the front
    auto drain = [&](bool all) {
        sec.lap(SEC_LOOPS);
        int done = 0;
        while (qn - done >= 64 || (all && qn - done > 0)) {
            const int avail = qn - done < 64 ? qn - done : 64;
            pair_step(done, avail);
            done += avail;
        }
        const int rem = qn - done;
        if (rem > 0 && done > 0) {
            const QE e = lane < rem ? queue[done + lane] : (QE)0;
            __builtin_amdgcn_wave_barrier();
            if (lane < rem) queue[lane] = e;
            __builtin_amdgcn_wave_barrier();
        }
        qn = __builtin_amdgcn_readfirstlane(rem);
        sec.lap(SEC_LEAVES);
    };

    // b: candidates of the subtree rooted at node c (level cut_level) whose box is `cbox`
    auto brute = [&](uint32_t c, const N &cbox) {
        sec.lap(SEC_DESCENT);
        work.add(0, q.lane_on);
        uint64_t on_mask;
        bool on; // (per lane: only the generic loop below reads it)
        if constexpr (std::is_same<TN, float>::value) {
            uint64_t init = lane_on_mask;
            if constexpr (MODE == MODE_SELF) init &= __builtin_amdgcn_ballot_w64(!((c + 1u) <= (self_next >> (levels - cut_level))));
            on_mask = contact_mask_f32(init, cbox, q.q_node);
            on = (on_mask >> lane) & 1u;
        } else {
            on = q.lane_on & iscontact(q.q_node, cbox);
            if constexpr (MODE == MODE_SELF) on = on & !((c + 1u) <= (self_next >> (levels - cut_level)));
            on_mask = __builtin_amdgcn_ballot_w64(on);
        }
        if (on_mask == 0) {
            sec.lap(SEC_SUBTREE);
            return;
        }
        const uint32_t first32 = (c - (1u << (cut_level - 1))) << (lp - cut_level); // 0-based, within level lp
        uint32_t last = first32 + (1u << (lp - cut_level));
        last = last < lp_real ? last : lp_real;
        const int np = (int)(last - first32); // <= 64
        // lanes without a parent (a ragged last subtree) re-read the last one and stay out of box_mask: whatever they
        // compute below is masked (straight-line load: no exec-masked region, no "empty box" to materialise)
#if IBVH_LVT_STRAIGHT & 2
        const N mybox = load_vol<N>(lp_nodes + (first32 + (uint32_t)(lane < np ? lane : np - 1)));
        const bool mine = lane < np;
#else
        N mybox; // lanes without a parent hold the empty box: it matches nothing
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mybox.lo[k] = float_max<TN>();
            mybox.up[k] = -float_max<TN>();
        }
        if (lane < np) mybox = load_vol<N>(lp_nodes + (first32 + (uint32_t)lane));
        const bool mine = true;
#endif
        work.add(2, lane < np);
        work.add(0, lane < np ? 2u : 0u); // against the wave's two boxes
        const uint32_t right_leaf = 2u * (first32 + (uint32_t)lane) + 1u; // of this lane's parent
        uint64_t box_mask;
        bool box_on; // (per lane: only the generic loop below reads it)
        if constexpr (std::is_same<TN, float>::value) {
            uint64_t init = __builtin_amdgcn_ballot_w64(mine);
            if constexpr (MODE == MODE_SELF) init &= __builtin_amdgcn_ballot_w64(right_leaf > wave_item0);
            box_mask = contact_mask2_f32(init, ubox_a, ubox_b, mybox);
            box_on = (box_mask >> lane) & 1u;
        } else {
            box_on = mine & touches_wave(mybox);
            if constexpr (MODE == MODE_SELF) box_on = box_on & (right_leaf > wave_item0);
            box_mask = __builtin_amdgcn_ballot_w64(box_on);
        }
        // shorter of the two loops: lanes = queries over the parents that touch the wave's boxes, or
        // lanes = parents over the active queries.  (Measured alternative: lanes = (query, parent) pairs
        // pulled together with ds_bpermute — as many steps as this loop has iterations, and slower.)
        uint32_t n_box = (uint32_t)__builtin_popcountll(box_mask), n_on = (uint32_t)__builtin_popcountll(on_mask);
        // (opaque to the optimiser: it otherwise compares the two 64-bit popcounts, which the scalar unit cannot do — a v_mov
        // and a v_cmp_lt_u64 per subtree — and counts the active queries early, parking the number in a vector register)
        asm volatile("" : "+s"(n_box), "+s"(n_on));
        const bool by_box = n_box < n_on;
        sec.lap(SEC_SUBTREE);
        const QE e_box = (QE)lane | ((QE)first32 << 6);             // + (u << 6)
        const QE e_qry = (QE)(first32 + (uint32_t)lane) << 6;       // | u
        if constexpr (std::is_same<TN, float>::value && !WIDE) {
            // hand-scheduled step (test_and_append_f32); the pair walk has no prune: thresholds that always pass.  The loops
            // keep the queue's LDS byte address (the step advances it) instead of the entry count.
            const uint32_t queue_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)queue;
            const uint32_t drain_at = queue_lds + 4u * (uint32_t)(QUEUE_CAP - 64);
            uint32_t qaddr = queue_lds + 4u * (uint32_t)qn;
            auto drain_if_full = [&]() {
                if (qaddr > drain_at) {
                    qn = (int)((qaddr - queue_lds) >> 2);
                    drain(false);
                    qaddr = queue_lds + 4u * (uint32_t)qn;
                }
            };
            // 2p + 1 > item  <=>  p >= (item + 1) >> 1: the parent index itself is the scalar operand
            const uint32_t half_item = MODE == MODE_SELF ? (my_item + 1u) >> 1 : 0u;
            auto loop_by_box = [&](auto prune) {
                for (uint64_t todo = box_mask; todo != 0;) {
                    drain_if_full();
                    const int u = __builtin_ctzll(todo);
                    asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(u)); // (todo &= todo - 1 costs three scalar instructions)
                    // (the parent box comes straight from memory with scalar loads — it is L2-hot, lane u just loaded it —
                    // instead of six v_readlane out of `mybox`: 3 % fewer VALU cycles, measured; 32-bit byte offset: leaf-parent
                    // indices stay below 2^26 here, and a 64-bit product costs four scalar instructions instead of one)
                    const uint32_t pidx = first32 + (uint32_t)u;
                    const N p = load_vol_uniform<N>((const char *)lp_nodes + pidx * (uint32_t)sizeof(N));
                    work.add(0, (uint32_t)(on_mask >> lane) & 1u);
                    work.add(2, lane == 0);
                    test_and_append_f32<true, decltype(prune)::value>(on_mask, p.lo[0], p.lo[1], p.lo[2], p.up[0], p.up[1], p.up[2], q.q_node.lo[0], q.q_node.lo[1],
                                              q.q_node.lo[2], q.q_node.up[0], q.q_node.up[1], q.q_node.up[2], pidx, half_item,
                                              e_box + ((uint32_t)u << 6), qaddr);
                }
            };
            auto loop_by_query = [&](auto prune) {
                for (uint64_t todo = on_mask; todo != 0;) {
                    drain_if_full();
                    const int u = __builtin_ctzll(todo);
                    asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(u));
                    const N qb = broadcast_from_lane(q.q_node, u);
                    work.add(0, (uint32_t)(box_mask >> lane) & 1u);
                    const uint32_t thr = MODE == MODE_SELF ? wave_item0 + (uint32_t)u : 0u; // < right_leaf
                    test_and_append_f32<false, decltype(prune)::value>(box_mask, qb.lo[0], qb.lo[1], qb.lo[2], qb.up[0], qb.up[1], qb.up[2], mybox.lo[0], mybox.lo[1],
                                               mybox.lo[2], mybox.up[0], mybox.up[1], mybox.up[2], thr, MODE == MODE_SELF ? right_leaf : 1u,
                                               e_qry | (uint32_t)u, qaddr);
                }
            };
            // (the pair walk has no prune: its loops carry six compares.  The self walk's prune drops nothing in a subtree
            // beyond the wave's last item — most of them — but a second copy of the loops for those costs more than the compare
            // it saves: 128 us against 125, six more SGPR spills; profiles/r03_lvt_variants.txt)
            using Prune = std::integral_constant<bool, MODE == MODE_SELF>;
            if (by_box) loop_by_box(Prune{});
            else loop_by_query(Prune{});
            qn = (int)((qaddr - queue_lds) >> 2);
        } else {
            for (uint64_t todo = by_box ? box_mask : on_mask; todo != 0; todo &= todo - 1) {
                if (qn > QUEUE_CAP - 64) drain(false);
                const int u = __builtin_ctzll(todo);
                bool h;
                QE e;
                work.add(0, (uint32_t)((by_box ? on_mask : box_mask) >> lane) & 1u);
                if (by_box) {
                    const N pbox = broadcast_from_lane(mybox, u);
                    h = on & iscontact(q.q_node, pbox);
                    if constexpr (MODE == MODE_SELF) h = h & (2u * (first32 + (uint32_t)u) + 1u > my_item);
                    e = e_box + ((QE)u << 6);
                } else {
                    const N qbox = broadcast_from_lane(q.q_node, u);
                    h = box_on & iscontact(qbox, mybox);
                    if constexpr (MODE == MODE_SELF) h = h & (right_leaf > wave_item0 + (uint32_t)u);
                    e = e_qry | (QE)u;
                }
                const uint64_t hm = __builtin_amdgcn_ballot_w64(h);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                if (h) queue[qn + rank] = e;
                qn = __builtin_amdgcn_readfirstlane(qn + __popcll(hm));
            }
        }
        sec.lap(SEC_LOOPS);
    };

    sec.lap(SEC_PROLOGUE);
    // a: frontier descent from the start level to the cut level.  Roots are taken in chunks so the first
    // frontier always fits; chunks and frontier entries stay in increasing node order.
    uint32_t *fr0 = s_frontier[wv][0], *fr1 = s_frontier[wv][1];
    const int start_level = (int)a.start_level;
    const uint32_t root_first = 1u << (start_level - 1);
    const int roots = (int)num_real(start_level);
    bool overflow = false;
    for (int r0 = 0; r0 < roots && !overflow; r0 += FRONTIER_CAP) {
        int count = __builtin_amdgcn_readfirstlane((roots - r0) < FRONTIER_CAP ? (roots - r0) : FRONTIER_CAP);
        for (int i = lane; i < count; i += 64) fr0[i] = root_first + (uint32_t)(r0 + i);
        __builtin_amdgcn_wave_barrier();
        uint32_t *cur = fr0, *nxt = fr1;
        for (int lvl = start_level; lvl <= cut_level && count > 0; ++lvl) {
            const N *lvl_nodes = a.nodes + first_mem(lvl);
            const uint32_t lvl_first = 1u << (lvl - 1);
            const uint32_t child_real = num_real(lvl + 1);
            int next_count = 0;
            for (int base = 0; base < count; base += 64) {
                const bool have = base + lane < count;
                // (straight-line: lanes beyond the frontier re-read its first entry and are masked afterwards — no exec-masked
                // region around the loads, and the ballot below is the compare mask itself)
                work.add(2, have);
                work.add(0, have ? 2u : 0u);
#if IBVH_LVT_STRAIGHT & 1
                const uint32_t idx = cur[have ? base + lane : base];
                const N box = load_vol<N>(lvl_nodes + (idx - lvl_first));
                bool hit;
                uint64_t hit_mask;
                if constexpr (std::is_same<TN, float>::value) {
                    bool pre = have;
                    if constexpr (MODE == MODE_SELF) pre = pre & !((idx + 1u) <= (wave_next >> (levels - lvl)));
                    const uint64_t init = __builtin_amdgcn_ballot_w64(pre);
                    hit_mask = contact_mask2_f32(init, ubox_a, ubox_b, box);
                    hit = (hit_mask >> lane) & 1u;
                } else {
                    hit = have & touches_wave(box);
                    if constexpr (MODE == MODE_SELF) hit = hit & !((idx + 1u) <= (wave_next >> (levels - lvl)));
                    hit_mask = __builtin_amdgcn_ballot_w64(hit);
                }
#else
                const uint32_t idx = have ? cur[base + lane] : 0u;
                N box;
                bool hit = false;
                if (have) {
                    box = load_vol<N>(lvl_nodes + (idx - lvl_first));
                    hit = touches_wave(box);
                    if constexpr (MODE == MODE_SELF) hit = hit & !((idx + 1u) <= (wave_next >> (levels - lvl)));
                }
                const uint64_t hit_mask = __builtin_amdgcn_ballot_w64(hit);
#endif
                const uint64_t hm = hit_mask;
                if (lvl == cut_level) {
                    for (uint64_t todo = hm; todo != 0; todo &= todo - 1) {
                        const int src = __builtin_ctzll(todo);
                        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)idx, src);
                        // (the cut node's box by scalar load — L2-hot, lane `src` has just fetched it — instead of six v_readlane)
                        const N cbox = load_vol_uniform<N>((const char *)lvl_nodes + (c - lvl_first) * (uint32_t)sizeof(N));
                        brute(c, cbox);
                    }
                } else {
                    const int before = __popcll(hm & lt_mask);
                    const int total = __popcll(hm);
                    const bool last_virtual = hm != 0 && [&] {
                        const int top = 63 - __builtin_clzll(hm);
                        const uint32_t ti = (uint32_t)__builtin_amdgcn_readlane((int)idx, top);
                        return (2u * ti + 1u - (1u << lvl)) >= child_real;
                    }();
                    const int add = __builtin_amdgcn_readfirstlane(2 * total - (last_virtual ? 1 : 0)); // (readfirstlane: tell the compiler it is uniform)
                    if (next_count + add > FRONTIER_CAP) {
                        overflow = true;
                        break;
                    }
                    if (hit) {
                        nxt[next_count + 2 * before] = 2u * idx;
                        if (2 * before + 1 < add) nxt[next_count + 2 * before + 1] = 2u * idx + 1u;
                    }
                    next_count = __builtin_amdgcn_readfirstlane(next_count + add);
                }
            }
            if (overflow) break;
            __builtin_amdgcn_wave_barrier();
            uint32_t *t = cur;
            cur = nxt;
            nxt = t;
            count = __builtin_amdgcn_readfirstlane(lvl == cut_level ? 0 : next_count);
        }
    }
    if (overflow) {
        // frontier too wide for LDS (heavily overlapping input): redo this wave with the exact walk from
        // scratch; whatever was already emitted is written again, identically
        q.cnt = 0;
        q.cache.K = 0; // (counting pass) no slot-major writes into the dense regions; the wave walks again when writing
        if constexpr (WRITE) q.w = (q.valid && q.item > 0) ? (Cnt)a.counts[q.item - 1] : 0;
        joint_walk(q, a);
        q.finish();
        work.flush(a.work); // (the exact walk's own tests are not counted: frontier overflow only happens on heavily overlapping input)
        if constexpr (!WRITE)
            if (region && lane == 0) *(int *)region = -1;
        return;
    }
    sec.lap(SEC_DESCENT);
    drain(true);
    work.flush(a.work);
    if constexpr (!WRITE) {
        if (q.valid) a.counts[q.item] = (I)cnts[lane];
        // positions within a query's list must fit the entry's meta field: 2^25 contacts of one leaf never happen
        // for Int32 lists that fit, but keep the check exact
        const bool meta_ok = __builtin_amdgcn_ballot_w64((int64_t)cnts[lane] >= ((int64_t)1 << (sizeof(I) * 8 - 7))) == 0;
        if (region && lane == 0) *(int *)region = (wfill <= entry_cap && meta_ok) ? wfill : -1;
    }
    sec.lap(SEC_EPILOGUE);
    sec.flush();
}


// ---- (2b) BBox nodes: wave-local DUAL descent + candidate-pair queue ----------------------------------
// Round 4.  lvt_queue_kernel pairs the wave's 64 queries with the tree in (wave, 128-leaf subtree) tiles: ~23 tiles per
// wave at 1e6 random spheres, each paying a prologue of ~100 instructions and ~3 loop iterations of 27 for ~14 candidates —
// 124 lane-level box tests per leaf, 4 % of them hits (profiles/r04_lvt_sections.json: descent 30 %, subtree prologues 19 %,
// candidate loops 30 % of a wave's time).  This kernel descends BOTH sides instead: the wave's queries get a hierarchy of
// their own (Q nodes: the lanes of an aligned group of 64 >> (d - 1) lanes on ONE side of the wave's best cut — the two-box
// split of lvt_queue_kernel is its depth 1 — down to single lanes at depth 7; boxes of depths 1 .. 6 in an LDS table), and
// the unit of work is a PAIR (Q node, tree node T) whose boxes touch.  A lane pops one pair, fetches T's two children (48
// contiguous bytes) and Q's two children and tests the <= 4 child pairs at once; passing pairs are appended, in order, to
// the next segment.  All pairs of a segment sit at the same (Q depth, T level): first only T is split (Q waits at depth 1),
// the last six steps split both sides, and the pairs that pass the last step are (single query, leaf parent) — exactly the
// candidates of lvt_queue_kernel, tested with the same exact box test — which the unchanged leaf-test step consumes.
// tools/sim_lvt_dual.py (the oracle's tree of config 2): 54 lane-level box tests per leaf instead of 124, 23.5 64-lane
// steps per wave (worst wave 32), 318 candidates per wave.
//
// Order.  Every query's candidates must reach the leaf-test step in increasing leaf-parent order.  Invariant: within a
// segment, the pairs of any fixed Q node appear in increasing T order; a pair's children replace it in place, T-major
// ((Qa,Ta) (Qb,Ta) (Qa,Tb) (Qb,Tb)), lanes in order — so it holds for the next segment, whatever subset passes.
//
// Memory.  Segments live in ONE LDS array used as a double-ended stack: even segments grow up from the bottom, odd ones
// down from the top; a consumed segment is popped.  While everything fits this is a plain level-synchronous descent
// (|segment s| + |segment s+1| <= capacity).  When the gap runs short the producer of segment s+1 pauses, segment s+1 is
// consumed first (recursively: depth-first on demand), popped, and the producer resumes into a fresh segment s+1 — order is
// preserved because a paused segment's remainder is only expanded after everything before it has left the pipeline.  A
// reserve of four entries per deeper step guarantees progress (the chunk shrinks to what fits), so there is no overflow
// path: heavily overlapping input degrades to smaller chunks instead of falling back to the exact walk.
#ifndef IBVH_DUAL_STACK
#define IBVH_DUAL_STACK 896
#endif
#ifndef IBVH_DUAL_MINWAVES
#define IBVH_DUAL_MINWAVES 7
#endif
constexpr int DUAL_STACK = IBVH_DUAL_STACK;
constexpr int DUAL_MINWAVES = IBVH_DUAL_MINWAVES;
constexpr int DUAL_QSLOTS = 70; // depths 1 .. 6: 2^(d-1) groups + 1 (the group the cut falls into has a part on either side) = 69
// first table slot of depth d (1 .. 6)
IBVH_D int dual_qoff(int d) { return (1 << (d - 1)) - 1 + (d - 1); }

// The dual step's four box tests for Float32 boxes, hand-scheduled like test_and_append_f32: chain k starts from the lanes in
// i_k and narrows EXEC with six v_cmpx (iscontact(Q, T): Q.lo <= T.up and Q.up >= T.lo per axis — the same ordered compares
// as the C++ operators), its surviving lanes are m_k.  All 64 lanes are active at the call site (wave-uniform control flow):
// EXEC is restored to all-ones.
#define IBVH_DUAL_CHAIN(I_, M_, Q_, T_)                  \
    "s_mov_b64 exec, %[" I_ "]\n\t"                      \
    "v_cmpx_le_f32 %[" Q_ "l0], %[" T_ "u0]\n\t"         \
    "v_cmpx_ge_f32 %[" Q_ "u0], %[" T_ "l0]\n\t"         \
    "v_cmpx_le_f32 %[" Q_ "l1], %[" T_ "u1]\n\t"         \
    "v_cmpx_ge_f32 %[" Q_ "u1], %[" T_ "l1]\n\t"         \
    "v_cmpx_le_f32 %[" Q_ "l2], %[" T_ "u2]\n\t"         \
    "v_cmpx_ge_f32 %[" Q_ "u2], %[" T_ "l2]\n\t"         \
    "s_mov_b64 %[" M_ "], exec\n\t"
#define IBVH_DUAL_BOX(P_, B_) [P_##l0] "v"(B_.lo[0]), [P_##l1] "v"(B_.lo[1]), [P_##l2] "v"(B_.lo[2]), [P_##u0] "v"(B_.up[0]), [P_##u1] "v"(B_.up[1]), [P_##u2] "v"(B_.up[2])
IBVH_D void dual_test4_f32(uint64_t i0, uint64_t i1, uint64_t i2, uint64_t i3, const BBox<float> &Qa, const BBox<float> &Qb, const BBox<float> &Ta,
                           const BBox<float> &Tb, uint64_t &m0, uint64_t &m1, uint64_t &m2, uint64_t &m3) {
    asm volatile(IBVH_DUAL_CHAIN("i0", "m0", "qa", "ta") IBVH_DUAL_CHAIN("i1", "m1", "qb", "ta") IBVH_DUAL_CHAIN("i2", "m2", "qa", "tb")
                     IBVH_DUAL_CHAIN("i3", "m3", "qb", "tb") "s_mov_b64 exec, -1"
                 : [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
                 : [i0] "s"(i0), [i1] "s"(i1), [i2] "s"(i2), [i3] "s"(i3), IBVH_DUAL_BOX(qa, Qa), IBVH_DUAL_BOX(qb, Qb), IBVH_DUAL_BOX(ta, Ta),
                   IBVH_DUAL_BOX(tb, Tb)
                 : "vcc");
}
// two tests of one Q box (the steps that split only the tree side)
IBVH_D void dual_test2_f32(uint64_t i0, uint64_t i2, const BBox<float> &Qa, const BBox<float> &Ta, const BBox<float> &Tb, uint64_t &m0, uint64_t &m2) {
    asm volatile(IBVH_DUAL_CHAIN("i0", "m0", "qa", "ta") IBVH_DUAL_CHAIN("i2", "m2", "qa", "tb") "s_mov_b64 exec, -1"
                 : [m0] "=&s"(m0), [m2] "=&s"(m2)
                 : [i0] "s"(i0), [i2] "s"(i2), IBVH_DUAL_BOX(qa, Qa), IBVH_DUAL_BOX(ta, Ta), IBVH_DUAL_BOX(tb, Tb)
                 : "vcc");
}
#undef IBVH_DUAL_CHAIN
#undef IBVH_DUAL_BOX
// the lanes of m_k store e_k at consecutive entries from LDS byte address `addr` on (`step` = +-4 bytes per entry, per lane)
IBVH_D void dual_push4(uint64_t m0, uint64_t m1, uint64_t m2, uint64_t m3, uint32_t addr, uint32_t step, uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3) {
    asm volatile("s_mov_b64 exec, %[m0]\n\t"
                 "ds_write_b32 %[addr], %[e0]\n\t"
                 "v_add_u32 %[addr], %[addr], %[step]\n\t"
                 "s_mov_b64 exec, %[m1]\n\t"
                 "ds_write_b32 %[addr], %[e1]\n\t"
                 "v_add_u32 %[addr], %[addr], %[step]\n\t"
                 "s_mov_b64 exec, %[m2]\n\t"
                 "ds_write_b32 %[addr], %[e2]\n\t"
                 "v_add_u32 %[addr], %[addr], %[step]\n\t"
                 "s_mov_b64 exec, %[m3]\n\t"
                 "ds_write_b32 %[addr], %[e3]\n\t"
                 "s_mov_b64 exec, -1"
                 : [addr] "+v"(addr)
                 : [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3), [step] "v"(step), [e0] "v"(e0), [e1] "v"(e1), [e2] "v"(e2), [e3] "v"(e3)
                 : "memory");
}
IBVH_D void dual_push2(uint64_t m0, uint64_t m2, uint32_t addr, uint32_t step, uint32_t e0, uint32_t e2) {
    asm volatile("s_mov_b64 exec, %[m0]\n\t"
                 "ds_write_b32 %[addr], %[e0]\n\t"
                 "v_add_u32 %[addr], %[addr], %[step]\n\t"
                 "s_mov_b64 exec, %[m2]\n\t"
                 "ds_write_b32 %[addr], %[e2]\n\t"
                 "s_mov_b64 exec, -1"
                 : [addr] "+v"(addr)
                 : [m0] "s"(m0), [m2] "s"(m2), [step] "v"(step), [e0] "v"(e0), [e2] "v"(e2)
                 : "memory");
}

// The lean phase's four tests: the wave's two side boxes A and B (wave-uniform: scalar operands) against a lane's two child
// boxes Ta (lanes ia) and Tb (lanes ib); same chains as dual_test4_f32.
#define IBVH_SIDE_CHAIN(I_, M_, S_, T_)                  \
    "s_mov_b64 exec, %[" I_ "]\n\t"                      \
    "v_cmpx_le_f32 %[" S_ "l0], %[" T_ "u0]\n\t"         \
    "v_cmpx_ge_f32 %[" S_ "u0], %[" T_ "l0]\n\t"         \
    "v_cmpx_le_f32 %[" S_ "l1], %[" T_ "u1]\n\t"         \
    "v_cmpx_ge_f32 %[" S_ "u1], %[" T_ "l1]\n\t"         \
    "v_cmpx_le_f32 %[" S_ "l2], %[" T_ "u2]\n\t"         \
    "v_cmpx_ge_f32 %[" S_ "u2], %[" T_ "l2]\n\t"         \
    "s_mov_b64 %[" M_ "], exec\n\t"
#define IBVH_VBOX(P_, B_) [P_##l0] "v"(B_.lo[0]), [P_##l1] "v"(B_.lo[1]), [P_##l2] "v"(B_.lo[2]), [P_##u0] "v"(B_.up[0]), [P_##u1] "v"(B_.up[1]), [P_##u2] "v"(B_.up[2])
#define IBVH_SBOX(P_, B_) [P_##l0] "s"(B_.lo[0]), [P_##l1] "s"(B_.lo[1]), [P_##l2] "s"(B_.lo[2]), [P_##u0] "s"(B_.up[0]), [P_##u1] "s"(B_.up[1]), [P_##u2] "s"(B_.up[2])
IBVH_D void dual_sides4_f32(uint64_t ia, uint64_t ib, const BBox<float> &A, const BBox<float> &B, const BBox<float> &Ta, const BBox<float> &Tb,
                            uint64_t &maa, uint64_t &mba, uint64_t &mab, uint64_t &mbb) {
    asm volatile(IBVH_SIDE_CHAIN("ia", "maa", "a", "ta") IBVH_SIDE_CHAIN("ia", "mba", "b", "ta") IBVH_SIDE_CHAIN("ib", "mab", "a", "tb")
                     IBVH_SIDE_CHAIN("ib", "mbb", "b", "tb") "s_mov_b64 exec, -1"
                 : [maa] "=&s"(maa), [mba] "=&s"(mba), [mab] "=&s"(mab), [mbb] "=&s"(mbb)
                 : [ia] "s"(ia), [ib] "s"(ib), IBVH_SBOX(a, A), IBVH_SBOX(b, B), IBVH_VBOX(ta, Ta), IBVH_VBOX(tb, Tb)
                 : "vcc");
}
#undef IBVH_SIDE_CHAIN
#undef IBVH_VBOX
#undef IBVH_SBOX

// The lane k after which the wave's 64 query boxes are best cut in two (minimum sum of the half-areas of box[0..k] and
// box[k+1..63]; lvt_queue_kernel's two-box split, see there), wave-uniform.  `box` is empty for lanes without a query.
template <class N> IBVH_D int dual_choose_cut(const N &box, int lane) {
    using TN = typename N::elt;
    const TN big = float_max<TN>();
    auto half_area = [](const N &b) {
        float d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d[k] = (float)b.up[k] - (float)b.lo[k];
            d[k] = d[k] > 0.0f ? d[k] : 0.0f; // empty (or NaN) -> 0
        }
        return d[0] * d[1] + d[1] * d[2] + d[0] * d[2];
    };
    if constexpr (std::is_same<TN, float>::value) {
        N pre = box, rev, rsuf;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rev.lo[k] = __shfl(pre.lo[k], 63 - lane, 64);
            rev.up[k] = __shfl(pre.up[k], 63 - lane, 64);
        }
        float mn[6] = {pre.lo[0], pre.lo[1], pre.lo[2], rev.lo[0], rev.lo[1], rev.lo[2]};
        float mx[6] = {pre.up[0], pre.up[1], pre.up[2], rev.up[0], rev.up[1], rev.up[2]};
        wave_prefix_scans_dpp(mn, mx);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pre.lo[k] = mn[k];
            pre.up[k] = mx[k];
            rsuf.lo[k] = mn[3 + k];
            rsuf.up[k] = mx[3 + k];
        }
        const float ra = half_area(rsuf);
        float sa = __shfl(ra, 62 - lane, 64);
        sa = lane == 63 ? 0.0f : sa;
        float cost = half_area(pre) + sa;
        cost = cost == cost ? cost : __builtin_inff();
        const float m = wave_min_dpp_lane63(cost);
        const float best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
        const uint64_t at = __builtin_amdgcn_ballot_w64(cost == best);
        return at != 0 ? (int)__builtin_ctzll(at) : 0;
    } else {
        N pre = box, suf = box, nxt_suf;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                TN t = __shfl_up(pre.lo[k], o, 64);
                pre.lo[k] = pre.lo[k] < t ? pre.lo[k] : t;
                t = __shfl_up(pre.up[k], o, 64);
                pre.up[k] = pre.up[k] > t ? pre.up[k] : t;
                t = __shfl_down(suf.lo[k], o, 64);
                suf.lo[k] = suf.lo[k] < t ? suf.lo[k] : t;
                t = __shfl_down(suf.up[k], o, 64);
                suf.up[k] = suf.up[k] > t ? suf.up[k] : t;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            nxt_suf.lo[k] = __shfl_down(suf.lo[k], 1, 64);
            nxt_suf.up[k] = __shfl_down(suf.up[k], 1, 64);
            if (lane == 63) {
                nxt_suf.lo[k] = big;
                nxt_suf.up[k] = -big;
            }
        }
        float cost = half_area(pre) + half_area(nxt_suf);
        cost = cost == cost ? cost : __builtin_inff();
        uint64_t key = ((uint64_t)__float_as_uint(cost) << 32) | (uint32_t)lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t t = (uint64_t)__shfl_xor((long long)key, o, 64);
            key = t < key ? t : key;
        }
        return (int)(key & 63u);
    }
}


// The schedule of a dual descent, made once per launch on the host (it depends on the tree's shape only): row[cur + 1]
// describes the step that consumes segment cur (cur = -1: the roots).
struct DualPlan {
    int32_t S;        // expansion steps 0 .. S-1 consume segments 0 .. S-1; segment S holds the candidates
    int32_t roots;    // real nodes of the level the descent starts at
    int32_t q_first;  // first step that splits the query side (the steps before it split only T)
    int32_t pad_;
    uint32_t row[34][8];
    // row: [0], [1] byte offset (low, high word) of the first node of the level the step's T children live on
    //      [2] T child b exists for t < this (0: the step does not split T)
    //      [3] bit 0: the step splits T; bit 1: its Q children are groups of the next depth (table rows / single lanes);
    //          bit 2: two Q children (four tests); bit 3: they are single lanes; bit 4: the pairs are the roots themselves;
    //          bits 8-9: code of Q child b - code of Q child a; bits 16-21: levels - (level of the T children);
    //          bits 24-27: depth of the Q children
    //      [4] table slot of the first Q child row
};
enum { DUAL_TS = 1, DUAL_QS = 2, DUAL_Q2 = 4, DUAL_SINGLES = 8, DUAL_IOTA = 16 };
inline DualPlan make_dual_plan(int levels, uint32_t vl, int L0, size_t node_bytes) {
    auto num_real = [&](int level) -> uint32_t { return (1u << (level - 1)) - (vl >> (levels - level)); };
    auto first_mem = [&](int level) -> uint32_t {
        const uint32_t v = vl >> (levels - (level - 1));
        return (1u << (level - 1)) - (2u * v - (uint32_t)__builtin_popcount(v)) - 1u;
    };
    DualPlan p{};
    const int lp = levels - 1, nT = lp - L0;
    p.S = nT > 6 ? nT : 6;
    p.roots = (int32_t)num_real(L0);
    const int t_first = p.S - nT, q_first = p.S - 6; // first step that splits T / Q
    p.q_first = q_first;
    for (int cur = -1; cur < p.S; ++cur) {
        const bool root = cur < 0, split_t = !root && cur >= t_first, split_q = root || cur >= q_first;
        const int tl_out = L0 + (root ? 0 : (cur >= t_first ? cur - t_first + 1 : 0));
        const int qd_out = root ? 1 : 1 + (cur >= q_first ? cur - q_first + 1 : 0);
        const bool qs = split_q && !root;
        uint32_t *r = p.row[cur + 1];
        const uint64_t off = (uint64_t)first_mem(tl_out) * node_bytes;
        r[0] = (uint32_t)off;
        r[1] = (uint32_t)(off >> 32);
        r[2] = split_t ? num_real(tl_out) >> 1 : 0u;
        r[3] = (split_t ? DUAL_TS : 0) | (qs ? DUAL_QS : 0) | (split_q ? DUAL_Q2 : 0) | (qd_out == 7 ? DUAL_SINGLES : 0) | (root ? DUAL_IOTA : 0) |
               ((root ? 1u : 2u) << 8) | ((uint32_t)(levels - tl_out) << 16) | ((uint32_t)qd_out << 24);
        r[4] = qs && qd_out <= 6 ? (uint32_t)((1 << (qd_out - 1)) - 1 + (qd_out - 1)) : 0u;
    }
    return p;
}
// the lanes below m (0 .. 64) as a mask: s_bfm_b64 takes the width modulo 64
IBVH_D uint64_t dual_mask_below(int m) {
    uint64_t r;
    asm("s_bfm_b64 %0, %1, 0" : "=s"(r) : "s"(m));
    return m >= 64 ? ~(uint64_t)0 : r;
}

// One butterfly step of the min / max all-reduce over aligned lane groups, on DPP for Float32 (six minima, six maxima in one
// block: a DPP operand must not have been written by the two preceding VALU instructions).  STEP 0 .. 3 complete the groups of
// 2, 4, 8 and 16 lanes (quad permutes, then the half-row and row mirrors: any pairing of the two halves of a group will do).
#define IBVH_DPP12(CTRL)                                                                                                       \
    asm volatile("s_nop 1\n\t"                                                                                                 \
                 "v_min_f32_dpp %0, %0, %0 " CTRL "\n\tv_min_f32_dpp %1, %1, %1 " CTRL "\n\tv_min_f32_dpp %2, %2, %2 " CTRL "\n\t"   \
                 "v_min_f32_dpp %3, %3, %3 " CTRL "\n\tv_min_f32_dpp %4, %4, %4 " CTRL "\n\tv_min_f32_dpp %5, %5, %5 " CTRL "\n\t"   \
                 "v_max_f32_dpp %6, %6, %6 " CTRL "\n\tv_max_f32_dpp %7, %7, %7 " CTRL "\n\tv_max_f32_dpp %8, %8, %8 " CTRL "\n\t"   \
                 "v_max_f32_dpp %9, %9, %9 " CTRL "\n\tv_max_f32_dpp %10, %10, %10 " CTRL "\n\tv_max_f32_dpp %11, %11, %11 " CTRL      \
                 : "+v"(mn[0]), "+v"(mn[1]), "+v"(mn[2]), "+v"(mn[3]), "+v"(mn[4]), "+v"(mn[5]), "+v"(mx[0]), "+v"(mx[1]), "+v"(mx[2]), \
                   "+v"(mx[3]), "+v"(mx[4]), "+v"(mx[5]))
template <int STEP> IBVH_D void group_reduce_step_dpp(float (&mn)[6], float (&mx)[6]) {
    if constexpr (STEP == 0) IBVH_DPP12("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    else if constexpr (STEP == 1) IBVH_DPP12("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    else if constexpr (STEP == 2) IBVH_DPP12("row_half_mirror row_mask:0xf bank_mask:0xf");
    else IBVH_DPP12("row_mirror row_mask:0xf bank_mask:0xf");
}
#undef IBVH_DPP12

template <class L, class N> constexpr int dual_min_waves() { // (Float64 volumes cannot reach the bench types' occupancy)
    return sizeof(typename N::elt) == 8 ? 4 : (sizeof(typename L::elt) == 8 ? 5 : DUAL_MINWAVES);
}
template <class L, class N, class I, int MODE, bool WRITE, bool NARROW, bool WIDE, bool COUNT = false>
__global__ __launch_bounds__(64, (dual_min_waves<L, N>())) void lvt_dual_kernel(Args<L, N, I> a, PairCache<I> cache, DualPlan plan) {
    using TN = typename N::elt;
    Work<COUNT> work;
    using Q = Query<L, N, I, MODE, WRITE, NARROW>;
    using Cnt = typename Q::Cnt;
    using QE = typename std::conditional<WIDE, uint64_t, uint32_t>::type; // pair: Q node (7 bits) | T index within its level << 7
    constexpr int CAP = WIDE ? DUAL_STACK / 2 : DUAL_STACK;
    __shared__ QE s_stack[CAP];
    constexpr int NW = (int)(sizeof(N) / 8); // a node box as 8-byte words
    __shared__ __attribute__((aligned(16))) uint64_t s_qtab[DUAL_QSLOTS * NW];
    __shared__ Cnt s_cnt[64];
    Q q(a, cache);
    const int lane = threadIdx.x;
    // wave-dense contact cache: see lvt_queue_kernel
    struct Entry {
        IndexPair<I> pair;
        I meta;
    };
    const int64_t first_item = q.item - lane;
    const int64_t items_here = a.n_items - first_item < 64 ? a.n_items - first_item : 64;
    char *region = cache.K > 0 && items_here > 0 ? (char *)(cache.slots + first_item * (int64_t)cache.K) : nullptr;
    const int entry_cap = region ? (int)(((int64_t)items_here * cache.K * (int64_t)sizeof(IndexPair<I>) - 16) / (int64_t)sizeof(Entry)) : 0;
    Entry *entries = (Entry *)(region + 16);
    int wfill = 0;
    if constexpr (WRITE) {
        if (a.guard_total != nullptr && load_total_uniform(a.guard_total) > a.guard_capacity) return;
        q.w = (q.valid && q.item > 0) ? (Cnt)a.counts[q.item - 1] : 0;
        const int fill = region ? __builtin_amdgcn_readfirstlane(*(const int *)region) : -1;
        if (fill >= 0) {
            s_cnt[lane] = q.w;
            __builtin_amdgcn_wave_barrier();
            for (int t = lane; t < fill; t += 64) {
                const Entry e = entries[t];
                const int64_t dest = (int64_t)s_cnt[(int)(e.meta & 63)] + (int64_t)(e.meta >> 6);
                a.contacts[dest] = e.pair;
            }
            return;
        }
        q.lane_on = q.valid;
    }
    if constexpr (MODE == MODE_PAIR) {
        if (a.built_level <= 1 && a.tree.levels >= 2) {
            const N root = load_vol_uniform<N>(a.nodes);
            work.add(0, q.lane_on);
            work.add(2, lane == 0);
            if (__builtin_amdgcn_ballot_w64(q.lane_on & iscontact(q.q_node, root)) == 0) {
                work.flush(a.work);
                if constexpr (!WRITE) {
                    if (q.valid) a.counts[q.item] = (I)0;
                    if (region && lane == 0) *(int *)region = 0;
                }
                return;
            }
        }
    }

    // (wave-uniform arithmetic is 32-bit on purpose, see lvt_queue_kernel)
    const int levels = (int)a.tree.levels;
    const uint32_t vl = (uint32_t)a.tree.virtual_leaves;
    auto num_real = [&](int level) -> uint32_t { return (1u << (level - 1)) - (vl >> (levels - level)); };
    auto first_mem = [&](int level) -> uint32_t {
        const uint32_t v = vl >> (levels - (level - 1));
        return (1u << (level - 1)) - (2u * v - (uint32_t)__builtin_popcount(v)) - 1u;
    };
    const uint32_t my_item = (uint32_t)q.item;
    const uint32_t wave_item0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)my_item);
    const uint32_t n_leaves = (uint32_t)a.tree.real_leaves;
    Cnt *cnts = s_cnt;
    cnts[lane] = WRITE ? q.w : (Cnt)0;

    // ---- the wave's own hierarchy ----
    N qbox; // this lane's query as a node box; empty for lanes without a query (NaN boxes touch nothing)
    {
        const TN big = float_max<TN>();
        bool use = q.lane_on;
#pragma unroll
        for (int k = 0; k < 3; ++k) use = use && q.q_node.lo[k] == q.q_node.lo[k] && q.q_node.up[k] == q.q_node.up[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            qbox.lo[k] = use ? q.q_node.lo[k] : big;
            qbox.up[k] = use ? q.q_node.up[k] : -big;
        }
    }
    const int ksplit = dual_choose_cut(qbox, lane); // side A = lanes 0 .. ksplit, side B = the rest
    N side_a, side_b_;
    {
        const TN big = float_max<TN>();
        const bool side_b = lane > ksplit;
        // the part of this lane's aligned group on side A (0 .. 2) / on side B (3 .. 5), minima and maxima
        TN mn[6], mx[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mn[k] = side_b ? big : qbox.lo[k];
            mx[k] = side_b ? -big : qbox.up[k];
            mn[3 + k] = side_b ? qbox.lo[k] : big;
            mx[3 + k] = side_b ? qbox.up[k] : -big;
        }
        auto store = [&](int slot, int part) { // (8-byte words: ds_write_b64)
            N b;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                b.lo[k] = mn[3 * part + k];
                b.up[k] = mx[3 * part + k];
            }
            uint64_t w[NW];
            __builtin_memcpy(w, &b, sizeof(N));
#pragma unroll
            for (int k = 0; k < NW; ++k) s_qtab[slot * NW + k] = w[k];
        };
        auto step = [&](auto dtag) {
            constexpr int d = decltype(dtag)::value; // depth whose groups (64 >> (d-1) lanes) are complete after this step
            constexpr int o = 1 << (6 - d);
            bool done = false;
            if constexpr (std::is_same<TN, float>::value && o <= 8) {
                group_reduce_step_dpp<6 - d>(mn, mx);
                done = true;
            }
            if (!done) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    TN t = __shfl_xor(mn[k], o, 64);
                    mn[k] = mn[k] < t ? mn[k] : t;
                    t = __shfl_xor(mx[k], o, 64);
                    mx[k] = mx[k] > t ? mx[k] : t;
                }
            }
            const int j = lane >> (7 - d), sidx = ksplit >> (7 - d);
            if ((lane & (2 * o - 1)) == 0) {
                if (j <= sidx) store(dual_qoff(d) + j, 0);
                if (j >= sidx) store(dual_qoff(d) + j + 1, 1);
            }
        };
        step(std::integral_constant<int, 6>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 1>{});
#pragma unroll
        for (int k = 0; k < 3; ++k) { // every lane now holds the two side boxes
            side_a.lo[k] = mn[k];
            side_a.up[k] = mx[k];
            side_b_.lo[k] = mn[3 + k];
            side_b_.up[k] = mx[3 + k];
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- schedule ----
    const int S = plan.S;               // expansion steps 0 .. S-1 consume segments 0 .. S-1; segment S holds the candidates
    const int roots = plan.roots;
    int lo_ptr = 0, hi_ptr = CAP;       // free entries: [lo_ptr, hi_ptr)
    auto seg_dir = [](int s) { return (s & 1) ? -1 : 1; };
    const uint32_t stack_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) QE *)s_stack;
    auto below = [](uint64_t mk, uint32_t acc) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, acc)); };

    // leaf tests of `avail` candidates (entries hd .. of the segment at org, direction dir), one pair per lane
    // (lvt_queue_kernel's stage c)
    auto pair_step = [&](int org, int dir, int hd, int avail) {
        const bool v = lane < avail;
        const QE e = s_stack[org + dir * (hd + (v ? lane : 0))];
        const int qi = (int)((uint32_t)(e >> 1) & 63u);
        const uint32_t pos = 2u * (uint32_t)(e >> 7);
        const bool has_b = v & (pos + 1u < n_leaves);
        uint64_t mor_a = 0, mor_b = 0;
        const char *rec = a.leaves + (int64_t)pos * a.lay.stride;
        const char *rec_b = (pos + 1u < n_leaves) ? rec + a.lay.stride : rec;
        const L leaf_a = load_vol<L>(rec), leaf_b = load_vol<L>(rec_b);
        const I idx_a = load_index<I>(rec, a.lay), idx_b = load_index<I>(rec_b, a.lay);
        if constexpr (NARROW) {
            if (a.narrow == IBVH_NARROW_MORTON_LT) {
                mor_a = load_morton(rec, a.lay);
                mor_b = load_morton(rec_b, a.lay);
            }
        }
        const L ql = shuffle_from(q.q_leaf, qi);
        const I qidx = __shfl(q.q_index, qi, 64);
        const uint32_t item_q = wave_item0 + (uint32_t)qi;
        bool hit_a = v & iscontact(ql, leaf_a), hit_b = has_b & iscontact(ql, leaf_b);
        work.add(1, (uint32_t)v + (uint32_t)has_b);
        work.add(3, (uint32_t)v + (uint32_t)has_b);
        if constexpr (MODE == MODE_SELF) {
            hit_a = hit_a & (pos > item_q);
            hit_b = hit_b & (pos + 1u > item_q);
        }
        if constexpr (NARROW) {
            const uint64_t qm = (uint64_t)__shfl((long long)q.q_morton, qi, 64);
            const bool fl = MODE == MODE_PAIR && a.flip;
            hit_a = hit_a && (fl ? narrow_eval(a.narrow, mor_a, idx_a, qm, qidx) : narrow_eval(a.narrow, qm, qidx, mor_a, idx_a));
            hit_b = hit_b && (fl ? narrow_eval(a.narrow, mor_b, idx_b, qm, qidx) : narrow_eval(a.narrow, qm, qidx, mor_b, idx_b));
        }
        uint32_t same_lo, same_hi;
        {
            const uint64_t vm = dual_mask_below(avail);
            same_lo = (uint32_t)vm;
            same_hi = (uint32_t)(vm >> 32);
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int x = (int)((uint32_t)qi << (31 - b)) >> 31;
            const uint64_t m = __builtin_amdgcn_ballot_w64(x != 0);
            same_lo = __builtin_amdgcn_bitop3_b32(same_lo, (uint32_t)m, (uint32_t)x, 0x90);
            same_hi = __builtin_amdgcn_bitop3_b32(same_hi, (uint32_t)(m >> 32), (uint32_t)x, 0x90);
        }
        const uint64_t m_a = __builtin_amdgcn_ballot_w64(hit_a), m_b = __builtin_amdgcn_ballot_w64(hit_b);
        const uint32_t sa_lo = same_lo & (uint32_t)m_a, sa_hi = same_hi & (uint32_t)(m_a >> 32);
        const uint32_t sb_lo = same_lo & (uint32_t)m_b, sb_hi = same_hi & (uint32_t)(m_b >> 32);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi(sb_hi, __builtin_amdgcn_mbcnt_lo(sb_lo, __builtin_amdgcn_mbcnt_hi(sa_hi, __builtin_amdgcn_mbcnt_lo(sa_lo, 0u))));
        const int tot = __builtin_popcount(sa_lo) + __builtin_popcount(sa_hi) + __builtin_popcount(sb_lo) + __builtin_popcount(sb_hi);
        const bool group_first = __builtin_amdgcn_mbcnt_hi(same_hi, __builtin_amdgcn_mbcnt_lo(same_lo, 0u)) == 0u;
        const Cnt base = cnts[qi];
        __builtin_amdgcn_wave_barrier();
        if (v && group_first && tot > 0) cnts[qi] = base + (Cnt)tot;
        __builtin_amdgcn_wave_barrier();
        auto put = [&](Cnt at, I lidx, int slot, uint32_t lpos) {
            IndexPair<I> c2;
            if (a.positions) {
                const I qp = (I)(item_q + 1u), lpp = (I)(lpos + 1u);
                c2 = (MODE == MODE_PAIR && a.flip) ? IndexPair<I>{lpp, qp} : IndexPair<I>{qp, lpp};
            } else if constexpr (MODE == MODE_SELF) c2 = qidx > lidx ? IndexPair<I>{lidx, qidx} : IndexPair<I>{qidx, lidx};
            else c2 = a.flip ? IndexPair<I>{lidx, qidx} : IndexPair<I>{qidx, lidx};
            if constexpr (WRITE) {
                a.contacts[(int64_t)at] = c2;
            } else {
                if (slot < entry_cap) entries[slot] = Entry{c2, (I)((I)qi | ((I)(at - 0) << 6))};
            }
        };
        const Cnt at = base + (Cnt)rank;
        const int n_a = __popcll(m_a);
        if (hit_a) put(at, idx_a, wfill + (int)below(m_a, 0u), pos);
        if (hit_b) put(at + (hit_a ? 1 : 0), idx_b, wfill + n_a + (int)below(m_b, 0u), pos + 1u);
        if constexpr (!WRITE) wfill = __builtin_amdgcn_readfirstlane(wfill + n_a + __popcll(m_b));
    };

    // ---- scheduler (wave-uniform): breadth-first while the segments fit, depth-first on demand ----
    // cur = the segment being consumed (-1: the roots), its state in scalar registers.  A segment that has to wait
    // half-consumed (space ran short: what it produced so far is consumed first) parks its state in lane cur + 1 of sg_*.
    int cur = -1, c_org = 0, c_cnt = roots, c_head = 0;
    uint32_t paused = 0;                // bit s + 1: segment s is parked
    int sg_head = 0, sg_cnt = 0, sg_org = 0;

    // ---- lean phase: while only T is split and the frontier fits the wave's 64 lanes, it lives in registers (lane j: the
    // j-th node, at most 64), both side boxes are tested at once from scalar registers and a node that touches either side
    // is kept ONCE; no segment bookkeeping.  The step that leaves the phase (the next step splits Q, or more than 64 nodes
    // pass) pushes (side, node) pairs into the segment the general scheduler continues from. ----
    if (roots <= 64) {
        const N A = broadcast_from_lane(side_a, 0), B = broadcast_from_lane(side_b_, 0);
        uint32_t ft = (uint32_t)lane; // the frontier: T index within its level
        int fcnt = roots;
        for (int s = -1;; ++s) {      // the step that consumes "segment" s
            const uint32_t *row = plan.row[s + 1];
            const uint32_t flags = row[3], tb_lim = row[2];
            const int ts = (int)(flags & DUAL_TS);
            const int sh = (int)((flags >> 16) & 63u);
            const char *lvl = (const char *)a.nodes + (((uint64_t)row[1] << 32) | row[0]);
            const uint64_t vm = dual_mask_below(fcnt);
            const uint32_t t = lane < fcnt ? ft : 0u;
            const uint64_t tbm = __builtin_amdgcn_ballot_w64(t < tb_lim);
            const uint32_t offa = t * ((uint32_t)sizeof(N) << ts);
            const uint32_t offb = (t < tb_lim) ? offa + (uint32_t)sizeof(N) : offa;
            const N Ta = load_vol<N>(lvl + offa);
            const N Tb = load_vol<N>(lvl + offb);
            work.add(2, lane < fcnt ? ((t < tb_lim) ? 2u : 1u) : 0u);
            uint64_t ia = vm, ib = vm & tbm;
            if constexpr (MODE == MODE_SELF) {
                const uint32_t thr = (wave_item0 + 1u) >> sh;
                ia &= __builtin_amdgcn_ballot_w64(t >= (ts ? (thr + 1u) >> 1 : thr));
                ib &= __builtin_amdgcn_ballot_w64(t >= (thr >> 1));
            }
            work.add(0, 2u * ((uint32_t)((ia >> lane) & 1u) + (uint32_t)((ib >> lane) & 1u)));
            uint64_t maa, mba, mab, mbb;
            if constexpr (std::is_same<TN, float>::value) dual_sides4_f32(ia, ib, A, B, Ta, Tb, maa, mba, mab, mbb);
            else {
                maa = ia & __builtin_amdgcn_ballot_w64(iscontact(A, Ta));
                mba = ia & __builtin_amdgcn_ballot_w64(iscontact(B, Ta));
                mab = ib & __builtin_amdgcn_ballot_w64(iscontact(A, Tb));
                mbb = ib & __builtin_amdgcn_ballot_w64(iscontact(B, Tb));
            }
            const uint64_t ma = maa | mba, mb = mab | mbb;
            const int n_merged = __popcll(ma) + __popcll(mb);
            const uint32_t ta = t << ts;
            if (s + 1 < plan.q_first && n_merged <= 64) {
                // stay: compact the surviving children (child a before child b, lanes in order) through the bottom of the stack
                if (n_merged == 0) {
                    cur = -2; // nothing touches the wave: done
                    break;
                }
                const uint32_t rank = below(mb, below(ma, 0u));
                if ((ma >> lane) & 1u) s_stack[rank] = (QE)ta;
                if ((mb >> lane) & 1u) s_stack[rank + (uint32_t)((ma >> lane) & 1u)] = (QE)(ta + 1u);
                __builtin_amdgcn_wave_barrier();
                ft = (uint32_t)s_stack[lane];
                __builtin_amdgcn_wave_barrier();
                fcnt = n_merged;
                continue;
            }
            // leave: (side, node) pairs into segment s + 1, T-major
            const int n_out = __popcll(maa) + __popcll(mba) + __popcll(mab) + __popcll(mbb);
            const int ddir = seg_dir(s + 1);
            const int d_org = ddir > 0 ? lo_ptr : hi_ptr - 1;
            const uint32_t rank = below(mbb, below(mab, below(mba, below(maa, 0u))));
            const QE e0 = (QE)ta << 7, e1 = e0 + 1u, e2 = e0 + 128u, e3 = e2 + 1u;
            if constexpr (!WIDE) {
                const uint32_t vstep = (uint32_t)(ddir * 4);
                dual_push4(maa, mba, mab, mbb, stack_lds + (uint32_t)(d_org * 4) + rank * vstep, vstep, e0, e1, e2, e3);
            } else {
                int j = (int)rank;
                if ((maa >> lane) & 1u) s_stack[d_org + ddir * j++] = e0;
                if ((mba >> lane) & 1u) s_stack[d_org + ddir * j++] = e1;
                if ((mab >> lane) & 1u) s_stack[d_org + ddir * j++] = e2;
                if ((mbb >> lane) & 1u) s_stack[d_org + ddir * j++] = e3;
            }
            __builtin_amdgcn_wave_barrier();
            if (ddir > 0) lo_ptr += n_out;
            else hi_ptr -= n_out;
            cur = n_out > 0 ? s + 1 : -2;
            c_org = d_org;
            c_cnt = n_out;
            c_head = 0;
            break;
        }
    }
    while (cur >= -1) {
        if (c_head == c_cnt) { // consumed (and popped): continue with the deepest parked segment, if any
            if (paused == 0) break;
            cur = 30 - __builtin_clz(paused);
            paused &= ~(1u << (cur + 1));
            c_org = __builtin_amdgcn_readlane(sg_org, cur + 1);
            c_cnt = __builtin_amdgcn_readlane(sg_cnt, cur + 1);
            c_head = __builtin_amdgcn_readlane(sg_head, cur + 1);
            continue;
        }
        if (cur == S) {
            for (; c_head < c_cnt; c_head += 64) pair_step(c_org, seg_dir(S), c_head, c_cnt - c_head < 64 ? c_cnt - c_head : 64);
            if (seg_dir(S) > 0) lo_ptr = c_org;
            else hi_ptr = c_org + 1;
            c_head = c_cnt = 0;
            continue;
        }
        // the step that consumes segment cur
        const uint32_t *row = plan.row[cur + 1];
        const uint32_t flags = row[3], tb_lim = row[2];
        const int ts = (int)(flags & DUAL_TS), qs = (int)((flags >> 1) & 1u);
        const uint32_t dq = (flags >> 8) & 3u;
        const int sh = (int)((flags >> 16) & 63u), qd_out = (int)((flags >> 24) & 15u);
        const char *lvl = (const char *)a.nodes + (((uint64_t)row[1] << 32) | row[0]);
        uint32_t thr_a = 0, thr_b = 0; // self walk: T child a / b is kept for t >= thr_a / thr_b (something right of the wave's first item below it)
        if constexpr (MODE == MODE_SELF) {
            const uint32_t thr = (wave_item0 + 1u) >> sh;
            thr_a = ts ? (thr + 1u) >> 1 : thr;
            thr_b = thr >> 1;
        }
        // Q children that do not exist: only the group the cut falls into has a part on either side
        uint32_t bad_a = ~0u, bad_b = ~0u;
        if (qs) {
            const uint32_t sidx_in = (uint32_t)ksplit >> (8 - qd_out), bit = ((uint32_t)ksplit >> (7 - qd_out)) & 1u;
            bad_a = (bit || qd_out == 7) ? 2u * sidx_in + 1u : ~0u; // side B of that group: its child a lies on side A
            bad_b = bit ? ~0u : 2u * sidx_in;                      // side A of that group: its child b lies on side B
        }
        const uint32_t qrow = row[4] * (uint32_t)NW; // first word of the Q children's table row
        const int fan_shift = ts + ((flags & DUAL_Q2) ? 1 : 0);
        const int res = 4 * (S - 1 - cur);
        const int need_full = res + (64 << fan_shift);
        const int sdir = cur < 0 ? 1 : seg_dir(cur), ddir = seg_dir(cur + 1);
        const int d_org = ddir > 0 ? lo_ptr : hi_ptr - 1; // (the segment this step produces is fresh: nothing deeper is pending)
        int d_cnt = 0, gap = hi_ptr - lo_ptr;
        const uint32_t vstep = (uint32_t)(ddir * (int)sizeof(QE));
        bool descend = false;
        auto chunks = [&](auto kind_tag) {
            constexpr int KIND = decltype(kind_tag)::value;
            while (c_head < c_cnt) {
                int m = c_cnt - c_head < 64 ? c_cnt - c_head : 64;
                if (gap < need_full) {
                    const int room = (gap - res) >> fan_shift;
                    if (room < m) {
                        if (d_cnt > 0) { // short of space: what this step produced so far is consumed first
                            descend = true;
                            break;
                        }
                        m = room;
                    }
                }
                const uint64_t vm = dual_mask_below(m);
                const int src = c_head + (lane < m ? lane : 0);
                uint32_t qn = 0, t = (uint32_t)src;
                if (KIND != 1 || !(flags & DUAL_IOTA)) {
                    const QE e = s_stack[c_org + sdir * src];
                    qn = (uint32_t)e & 127u;
                    t = (uint32_t)(e >> 7);
                }
                // tree side: the two children of T (or T itself while only Q is split); a missing child b re-reads child a
                const uint64_t tbm = __builtin_amdgcn_ballot_w64(t < tb_lim);
                const uint32_t offa = t * ((uint32_t)sizeof(N) << (KIND == 0 ? 1 : ts));
                const uint32_t offb = (t < tb_lim) ? offa + (uint32_t)sizeof(N) : offa;
                const N Ta = load_vol<N>(lvl + offa);
                const N Tb = load_vol<N>(lvl + offb);
                work.add(2, lane < m ? ((t < tb_lim) ? 2u : 1u) : 0u);
                // query side: the two children of Q (or Q itself while it waits at depth 1; a root meets both sides)
                N Qa, Qb;
                if constexpr (KIND == 2) {
                    const int la = (int)(qn & ~1u);
                    Qa = shuffle_from(qbox, la);
                    Qb = shuffle_from(qbox, la + 1);
                } else {
                    uint64_t w[2 * NW];
                    const uint32_t w0 = qrow + qn * (uint32_t)NW;
#pragma unroll
                    for (int k = 0; k < (KIND == 0 ? NW : 2 * NW); ++k) w[k] = s_qtab[w0 + k];
                    __builtin_memcpy(&Qa, w, sizeof(N));
                    if constexpr (KIND == 0) Qb = Qa;
                    else __builtin_memcpy(&Qb, w + NW, sizeof(N));
                }
                uint64_t am = vm, bm = 0;
                if constexpr (KIND != 0) {
                    am = vm & __builtin_amdgcn_ballot_w64(qn != bad_a);
                    bm = vm & __builtin_amdgcn_ballot_w64(qn != bad_b);
                }
                uint64_t ka = ~(uint64_t)0, kb = tbm;
                if constexpr (MODE == MODE_SELF) {
                    ka = __builtin_amdgcn_ballot_w64(t >= thr_a);
                    kb &= __builtin_amdgcn_ballot_w64(t >= thr_b);
                }
                const uint64_t i0 = am & ka, i1 = bm & ka, i2 = am & kb, i3 = bm & kb;
                work.add(0, (uint32_t)((i0 >> lane) & 1u) + (uint32_t)((i1 >> lane) & 1u) + (uint32_t)((i2 >> lane) & 1u) + (uint32_t)((i3 >> lane) & 1u));
                uint64_t m0, m1 = 0, m2, m3 = 0;
                if constexpr (std::is_same<TN, float>::value) {
                    if constexpr (KIND == 0) dual_test2_f32(i0, i2, Qa, Ta, Tb, m0, m2);
                    else dual_test4_f32(i0, i1, i2, i3, Qa, Qb, Ta, Tb, m0, m1, m2, m3);
                } else {
                    m0 = i0 & __builtin_amdgcn_ballot_w64(iscontact(Qa, Ta));
                    m2 = i2 & __builtin_amdgcn_ballot_w64(iscontact(Qa, Tb));
                    if constexpr (KIND != 0) {
                        m1 = i1 & __builtin_amdgcn_ballot_w64(iscontact(Qb, Ta));
                        m3 = i3 & __builtin_amdgcn_ballot_w64(iscontact(Qb, Tb));
                    }
                }
                const int n_out = __popcll(m0) + __popcll(m2) + (KIND != 0 ? __popcll(m1) + __popcll(m3) : 0);
                // entries: Q code | T index << 7
                const uint32_t qa = KIND == 0 ? qn : (KIND == 2 ? 2u * qn - (qn & 1u) : (qn << qs) - (qn & (uint32_t)qs));
                const QE e0 = ((QE)t << (7 + (KIND == 0 ? 1 : ts))) | (QE)qa, e1 = e0 + dq, e2 = e0 + 128u, e3 = e2 + dq;
                const uint32_t rank = KIND != 0 ? below(m3, below(m2, below(m1, below(m0, 0u)))) : below(m2, below(m0, 0u));
                if constexpr (!WIDE) {
                    const uint32_t addr = stack_lds + (uint32_t)((d_org + ddir * d_cnt) * 4) + rank * vstep;
                    if constexpr (KIND != 0) dual_push4(m0, m1, m2, m3, addr, vstep, e0, e1, e2, e3);
                    else dual_push2(m0, m2, addr, vstep, e0, e2);
                } else {
                    int j = d_cnt + (int)rank;
                    if ((m0 >> lane) & 1u) s_stack[d_org + ddir * j++] = e0;
                    if ((m1 >> lane) & 1u) s_stack[d_org + ddir * j++] = e1;
                    if ((m2 >> lane) & 1u) s_stack[d_org + ddir * j++] = e2;
                    if ((m3 >> lane) & 1u) s_stack[d_org + ddir * j++] = e3;
                }
                d_cnt += n_out;
                gap -= n_out;
                c_head += m;
                __builtin_amdgcn_wave_barrier();
            }
        };
        if (flags & DUAL_SINGLES) chunks(std::integral_constant<int, 2>{});
        else if (flags & DUAL_Q2) chunks(std::integral_constant<int, 1>{});
        else chunks(std::integral_constant<int, 0>{});
        if (ddir > 0) lo_ptr = d_org + d_cnt;
        else hi_ptr = d_org + 1 - d_cnt;
        if (descend) { // park this segment with its remainder
            sg_org = (lane == cur + 1) ? c_org : sg_org;
            sg_cnt = (lane == cur + 1) ? c_cnt : sg_cnt;
            sg_head = (lane == cur + 1) ? c_head : sg_head;
            paused |= 1u << (cur + 1);
        } else if (cur >= 0) { // consumed: pop it (everything deeper than cur + 1 is empty)
            if (sdir > 0) lo_ptr = c_org;
            else hi_ptr = c_org + 1;
        }
        ++cur;
        c_org = d_org;
        c_cnt = d_cnt;
        c_head = 0;
    }
    work.flush(a.work);
    if constexpr (!WRITE) {
        if (q.valid) a.counts[q.item] = (I)cnts[lane];
        const bool meta_ok = __builtin_amdgcn_ballot_w64((int64_t)cnts[lane] >= ((int64_t)1 << (sizeof(I) * 8 - 7))) == 0;
        if (region && lane == 0) *(int *)region = (wfill <= entry_cap && meta_ok) ? wfill : -1;
    }
}

// ---- (3) rays: per-lane walks, lanes refilled from the wave's block of rays -------------------------------
// The rays of a wave are not spatially coherent, so every lane walks its own ray — leaner than the reference's loop
// (raytrace/leaf_vs_tree/leaf_vs_tree.jl:187-225): a step tests BOTH children of the current node (adjacent in
// memory: one 48-byte fetch instead of two dependent ones) and the pending right siblings are a 32-bit mask instead
// of a 32-entry stack, possible because the tree is implicit.  Visit order is the reference's (left subtree, then the
// pending sibling, deepest first), so the hits of a ray come out in the same order.
//
// What bounds it (config 3: 1e6 rays, 7.2 M-triangle surface; measured in round 2): a ray takes 207 steps on average
// (2,621 at most), every step is a DEPENDENT fetch, and a wave's step costs what its lanes' different code paths cost
// one after the other — node level from global memory, leaf level, hit bookkeeping.  So:
//   * ONE fetch per step whatever the level: when a leaf record and a node have the same size (24 bytes for
//     BSphere{F32} leaves / BBox{F32} nodes), the lane computes ONE address — its two child nodes or its two leaf
//     records, 48 contiguous bytes either way — all lanes fetch together (three 16-byte requests each), and only the
//     arithmetic afterwards differs; other type combinations keep two fetch paths;
//   * the reciprocals 1/d are computed once per ray, not in every box test;
//   * a wave owns a BLOCK of 64 .. 256 consecutive rays and deals them to its lanes as they become free (39 % of
//     these rays hit nothing, the mean is 9.8 hits, the heaviest has 804: with one ray per lane for the life of a wave
//     the lanes were busy 10 % of the time by hit count); idle lanes take the next rays of the block whenever a
//     quarter of the wave is idle (ranked with v_mbcnt, no atomics).  The block size keeps ~3,000+ waves in the grid.
// A ray is walked by one lane from start to end, so its hits keep their order; the per-ray counts and the scanned
// output offsets make the result independent of which lane walked it.
// Hit cache as for leaf queries: the wave fills the scratch bytes of its block (block * K pairs) densely with
// (pair, ray-in-block | position in that ray's list << RAY_BITS) entries behind a 16-byte header {fill}; a wave walks
// again in the writing pass only if ALL its rays together found more than fits.
// (Tried and dropped in round 2: the top 10 levels of the tree in LDS — a third code path per step, no gain.)
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



template <class L, class N, class I, bool WRITE, bool COUNT = false>
__global__ __launch_bounds__(64) void lvt_rays_kernel(Args<L, N, I> a, PairCache<I> cache, int ray_block) {
    using T = typename L::elt;
    Work<COUNT> work;
    struct Entry { // 8 bytes: the ray's half of the pair follows from the meta field (ray-in-block | position << RAY_BITS)
        I leaf;
        I meta;
    };
    __shared__ int s_fill;
    if (a.gate != nullptr && *(const __attribute__((address_space(4))) int32_t *)(uintptr_t)a.gate == 0) return;
    const int lane = threadIdx.x;
    const int64_t first_item = (int64_t)blockIdx.x * ray_block;
    const int64_t left = a.n_items - first_item;
    const int items_here = (int)(left < ray_block ? left : ray_block);
    char *region = cache.K > 0 && items_here > 0 ? (char *)(cache.slots + first_item * (int64_t)cache.K) : nullptr;
    const int entry_cap = region ? (int)(((int64_t)items_here * cache.K * (int64_t)sizeof(IndexPair<I>) - 16) / (int64_t)sizeof(Entry)) : 0;
    Entry *entries = (Entry *)(region + 16);
    if (lane == 0) s_fill = 0;
    __builtin_amdgcn_wave_barrier();
    if constexpr (WRITE) {
        if (a.guard_total != nullptr && load_total_uniform(a.guard_total) > a.guard_capacity) return;
        const int fill = region ? __builtin_amdgcn_readfirstlane(*(const int *)region) : -1;
        if (fill >= 0) { // serve the whole block from its cache
            for (int t = lane; t < fill; t += 64) {
                const Entry e = entries[t];
                const int64_t ray = first_item + (int64_t)(e.meta & (RAY_BLOCK_MAX - 1));
                const int64_t w0 = ray > 0 ? (int64_t)a.counts[ray - 1] : 0;
                a.contacts[w0 + (int64_t)(e.meta >> RAY_BITS)] = IndexPair<I>{e.leaf, (I)(ray + 1)};
            }
            return;
        }
    }
    // tree constants (wave-uniform)
    const int levels = (int)a.tree.levels;
    const uint32_t vl = (uint32_t)a.tree.virtual_leaves; // < 2^(levels-1) <= 2^31
    const uint32_t leaf_first = 1u << (levels - 1);
    const int plevel = (int)a.start_level - 1;
    const int64_t roots = level_num_real(a.tree.levels, a.tree.virtual_leaves, a.start_level);
    const uint32_t pfirst = plevel >= 1 ? (1u << (plevel - 1)) : 0u;
    const uint32_t pcount = (uint32_t)((roots + 1) / 2); // pseudo-parents of the start-level roots
    // one fetch path for nodes and leaves when both are 24-byte records whose volume comes first
    constexpr bool SAME = sizeof(N) == 24 && sizeof(L) == 16;
    const bool unified = SAME && sizeof(I) == 4 && a.lay.stride == 24 && a.lay.index_off == 16;

    // per-lane ray state
    T p[3] = {0, 0, 0}, d[3] = {0, 0, 0}, inv[3] = {0, 0, 0}; // inv = 1 / d, once per ray (isintersection.jl:2-4)
    int ray = -1;          // ray-in-block this lane walks (-1: idle)
    uint32_t pi = 0;       // pseudo-parent being walked
    uint32_t inode = 0, pend = 0;
    int level = 0;
    int64_t w = 0, cnt = 0;
#ifdef IBVH_RAY_STEPS
    int64_t steps = 0; // diagnostic build: the per-ray STEP count goes where the hit count belongs
#endif
    bool meta_bad = false; // a position that does not fit the entry's meta field: the block walks again when writing
    int next = 0;          // wave-uniform: rays of the block handed out so far

    auto node_hit = [&](const N &n) {
        if constexpr (N::kind == IBVH_BBOX) return isintersection_inv(n, p, inv);
        else return isintersection(n, p, d);
    };
    auto emit = [&](I lidx, uint32_t lpos) {
        // (leaf.index, iray), raytrace/lvt:200 — or the leaf's 1-based position (IBVH_OUTPUT_POSITIONS)
        const IndexPair<I> c2{a.positions ? (I)(lpos + 1u) : lidx, (I)(first_item + ray + 1)};
        if constexpr (WRITE) {
            a.contacts[w++] = c2;
        } else {
            if (region) {
                const int slot = atomicAdd(&s_fill, 1);
                if (cnt >= ((int64_t)1 << (sizeof(I) * 8 - 1 - RAY_BITS))) meta_bad = true;
                if (slot < entry_cap) entries[slot] = Entry{c2.a, (I)((I)ray | ((I)cnt << RAY_BITS))};
            }
            ++cnt;
        }
    };

    for (;;) {
        // ---- refill: idle lanes take the next rays of the block
        const uint64_t idle = __builtin_amdgcn_ballot_w64(ray < 0);
        if (idle != 0 && next < items_here) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const int mine = next + rank;
            bool took = false;
            if (ray < 0 && mine < items_here) {
                const int64_t item = first_item + mine;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    p[k] = a.points[3 * item + k];
                    d[k] = a.dirs[3 * item + k];
                    inv[k] = T(1) / d[k];
                }
                // (behind the shadow walker this kernel serves the irregular rays only: Args::rays_filter)
                took = !(a.rays_filter == 1 && ray_is_regular(p, d, inv));
            }
            if (took) {
                const int64_t item = first_item + mine;
                ray = mine;
                pi = 0;
                inode = pfirst;
                level = plevel;
                pend = 0;
                cnt = 0;
                if constexpr (WRITE) w = item > 0 ? (int64_t)a.counts[item - 1] : 0;
            }
            const int taken = __popcll(idle);
            next = next + taken < items_here ? next + taken : items_here;
        }
        if (__builtin_amdgcn_ballot_w64(ray >= 0) == 0) {
            if (next >= items_here) break;
            continue; // (a whole draw of rays that are not this launch's: draw again)
        }
        // ---- walk: every busy lane advances its ray until a quarter of the wave has gone idle (or the block is used up
        // and everybody is done)
        for (;;) {
            if (ray >= 0) {
#ifdef IBVH_RAY_STEPS
                ++steps;
#endif
                const int cl = level + 1;
                const uint32_t c0 = 2u * inode, c1 = c0 + 1u;
                const uint32_t first = 1u << (cl - 1);
                const uint32_t nreal = first - (uint32_t)((uint64_t)vl >> (levels - cl));
                const bool real0 = c0 != 0u, real1 = (c1 - first) < nreal; // (c0 == 0: the pseudo node above the root)
                const bool at_leaves = cl == levels;
                work.add(at_leaves ? 1 : 0, (uint32_t)real0 + (uint32_t)real1);
                work.add(at_leaves ? 3 : 2, (uint32_t)real0 + (uint32_t)real1);
                const uint64_t v = (uint64_t)vl >> (levels - cl + 1);
                const uint32_t sk = (uint32_t)(2 * v) - (uint32_t)__popcll(v); // level_skips(cl)
                bool h0 = false, h1 = false, descended = false;
                I idx0 = 0, idx1 = 0;
                if (unified) {
                    if constexpr (SAME) {
                        // the two children — nodes or leaf records — are 48 contiguous bytes; a missing one re-reads its sibling
                        const char *base = at_leaves ? a.leaves + ((int64_t)c0 - (int64_t)leaf_first) * 24
                                                     : (const char *)(a.nodes + ((int64_t)c0 - (int64_t)sk - 1));
                        struct Raw {
                            uint32_t w[12];
                        } raw;
                        const char *lo = real0 ? base : base + 24;
                        if (real0 && real1) {
                            __builtin_memcpy(&raw, __builtin_assume_aligned(base, 8), 48);
                        } else {
                            __builtin_memcpy(&raw, __builtin_assume_aligned(lo, 8), 24);
                            __builtin_memcpy(&raw.w[6], &raw.w[0], 24);
                        }
                        if (at_leaves) {
                            L la, lb;
                            __builtin_memcpy(&la, &raw.w[0], 16);
                            __builtin_memcpy(&lb, &raw.w[6], 16);
                            h0 = real0 && isintersection(la, p, d);
                            h1 = real1 && isintersection(lb, p, d);
                            if (a.narrow == IBVH_NARROW_RAY_ORIGIN_OUTSIDE) { // raytrace/lvt:194: isintersection(...) && narrow(leaf, p, d)
                                h0 = h0 && origin_outside(la, p);
                                h1 = h1 && origin_outside(lb, p);
                            }
                            // .index sits right behind the 16-byte volume (4 or 8 bytes)
                            // (a 24-byte record with a 16-byte volume: the index is the 4 bytes behind it — a CONSTANT offset; the
                            // run-time a.lay.index_off made the compiler keep `raw` in LDS: 18 LDS instructions a step, SQ counters)
                            if constexpr (sizeof(I) == 4) {
                                idx0 = (I)raw.w[4];
                                idx1 = (I)raw.w[10];
                            }
                        } else {
                            N na, nb;
                            __builtin_memcpy(&na, &raw.w[0], 24);
                            __builtin_memcpy(&nb, &raw.w[6], 24);
                            h0 = real0 && node_hit(na);
                            h1 = real1 && node_hit(nb);
                        }
                    }
                } else if (at_leaves) {
                    const char *rec = a.leaves + ((int64_t)c0 - (int64_t)leaf_first) * a.lay.stride;
                    const L la = load_vol<L>(real0 ? rec : rec + a.lay.stride), lb = load_vol<L>(real1 ? rec + a.lay.stride : rec);
                    h0 = real0 && isintersection(la, p, d);
                    h1 = real1 && isintersection(lb, p, d);
                    if (a.narrow == IBVH_NARROW_RAY_ORIGIN_OUTSIDE) {
                        h0 = h0 && origin_outside(la, p);
                        h1 = h1 && origin_outside(lb, p);
                    }
                    if (h0) idx0 = load_index<I>(rec, a.lay);
                    if (h1) idx1 = load_index<I>(rec + a.lay.stride, a.lay);
                } else {
                    const N *np = a.nodes + ((int64_t)c0 - (int64_t)sk - 1);
                    struct Two {
                        N a, b;
                    };
                    Two ch;
                    if (real0 && real1) {
                        __builtin_memcpy(&ch, __builtin_assume_aligned(np, 8), sizeof(Two));
                    } else {
                        ch.a = load_vol<N>(real0 ? np : np + 1);
                        ch.b = ch.a;
                    }
                    h0 = real0 && node_hit(ch.a);
                    h1 = real1 && node_hit(ch.b);
                }
                if (at_leaves) {
                    if (h0) emit(idx0, c0 - leaf_first);
                    if (h1) emit(idx1, c1 - leaf_first);
                } else if (h0) {
                    if (h1) pend |= 1u << cl;
                    inode = c0;
                    level = cl;
                    descended = true;
                } else if (h1) {
                    inode = c1;
                    level = cl;
                    descended = true;
                }
                if (!descended) {
                    if (pend != 0) { // back to the deepest pending right sibling
                        const int pl = 31 - __builtin_clz(pend);
                        pend &= ~(1u << pl);
                        inode = (inode >> (level - pl)) | 1u;
                        level = pl;
                    } else if (++pi < pcount) { // next root pair of the start level
                        inode = pfirst + pi;
                        level = plevel;
                    } else { // ray finished
#ifdef IBVH_RAY_STEPS
                        if constexpr (!WRITE) a.counts[first_item + ray] = (I)steps;
                        steps = 0;
#else
                        if constexpr (!WRITE) a.counts[first_item + ray] = (I)cnt;
#endif
                        ray = -1;
                    }
                }
            }
            const uint64_t idle_now = __builtin_amdgcn_ballot_w64(ray < 0);
            if (idle_now == ~(uint64_t)0) break;
            if (next < items_here && __popcll(idle_now) >= 16) break;
        }
    }
    work.flush(a.work);
    if constexpr (!WRITE) {
        __builtin_amdgcn_wave_barrier();
        const bool ok = __builtin_amdgcn_ballot_w64(meta_bad) == 0;
        if (region && lane == 0) *(int *)region = (s_fill <= entry_cap && ok) ? s_fill : -1;
    }
}

// ---- (3b) rays over a quantised 8-wide SHADOW of the node levels --------------------------------------------------
// What bounds the binary ray walk is the number of dependent ~48-byte fetches (config 3: 206 node fetches per ray) and the
// bytes they move.  For a ray whose direction components are all finite and non-zero (and whose origin is finite) the slab
// test of isintersection.jl:1-33 is MONOTONE under box inclusion — every operation in it, (lo - p) * inv, min, max, is a
// weakly monotone function of its operands in floating point — and a BBox node is the exact min / max of its children
// (merge.jl:30-40), so such a ray reaches leaf j in the reference's walk (raytrace/leaf_vs_tree/leaf_vs_tree.jl:187-225)
// iff it hits the box of j's PARENT (that implies every ancestor) and then the leaf itself.  Any conservative enumeration
// of leaf parents followed by those two exact tests, in ascending leaf order, therefore reproduces the reference's hit
// list including its order — the interior levels only prune.  The enumeration used here is a shadow copy of the node
// levels in which ONE entry describes a node and its (up to) eight descendants three levels down:
//     { float lo[3], step[3]; uint32 valid; uint8 q[8][6] }   (80 bytes, 16-byte aligned)
// child c's box relative to the node's own exact box, 8 bits a coordinate, rounded OUTWARDS and verified against the very
// expression the walk evaluates (lo + float(q) * step, no contraction): dequantised boxes contain the exact ones, so by the
// same monotonicity a ray that hits an exact box hits its dequantised superset when the SAME slab function is applied.
// One 80-byte fetch thus replaces three levels of 48-byte fetches (config 3: 57 wide + 19 leaf-parent fetches per ray
// instead of 206), the children of an entry are tested from registers, and the walk is a depth-first visit in ascending
// child order whose stack is one byte per wide level (the tree is implicit: a child's index is (index << 3) | c).
// At the bottom a candidate leaf parent's exact box (24 bytes) and its two leaves (48 bytes) are fetched together and
// tested exactly.  Irregular rays (a zero or non-finite direction component, ...) are left to the binary walker, which is
// launched behind this kernel for them alone (Args::rays_filter) — config 3 has none, a launch that finds none returns at once.
// The shadow is rebuilt by every counting call (one streaming pass over the nodes: 0.05 ms for 7.2 M leaves) into the
// caller's scratch (ibvh_rays_scratch_bytes), so nothing outlives the call and a BVH needs no extra field.
constexpr int SHADOW_ENTRY_BYTES = 80;
constexpr int SHADOW_MAX_DEPTHS = 8; // wide levels: trees of up to 26 levels (the per-lane stack is one uint64)
struct RayShadow {
    int32_t depths;                       // wide levels K (0: no shadow)
    int32_t d0;                           // binary levels the top entry spans (1 .. 3); every other entry spans 3
    uint32_t base[SHADOW_MAX_DEPTHS + 1]; // first entry of wide level k (in entries); base[K] = total
};
IBVH_HD int shadow_level(const RayShadow &sh, int k) { return k == 0 ? 1 : 1 + sh.d0 + 3 * (k - 1); } // binary level of wide level k
inline RayShadow make_ray_shadow(const ibvh_tree &tree) {
    RayShadow sh{};
    const int64_t lp = tree.levels - 1; // leaf parents: the children of the bottom wide level
    if (lp < 7) return sh;
    const int K = (int)((lp - 1 + 2) / 3);
    if (K > SHADOW_MAX_DEPTHS) return sh;
    sh.depths = K;
    sh.d0 = (int)((lp - 1) - 3 * (K - 1));
    uint64_t run = 0;
    for (int k = 0; k < K; ++k) {
        sh.base[k] = (uint32_t)run;
        run += (uint64_t)level_num_real(tree.levels, tree.virtual_leaves, shadow_level(sh, k));
    }
    if (run >= ((uint64_t)1 << 32)) return RayShadow{};
    sh.base[K] = (uint32_t)run;
    return sh;
}
struct ShadowEntry {
    float lo[3], step[3];
    uint32_t valid;
    uint8_t q[8][6]; // child c: lo.x lo.y lo.z up.x up.y up.z
    uint32_t pad_;
};
static_assert(sizeof(ShadowEntry) == SHADOW_ENTRY_BYTES, "shadow entry layout");
// the ONE dequantisation expression (build-time verification and walk must agree bit for bit)
IBVH_D float shadow_dequant(float lo, float step, uint32_t q) { return lo + (float)q * step; }

template <class N>
__global__ __launch_bounds__(256) void ray_shadow_build_kernel(const N *__restrict__ nodes, TreeDev tree, RayShadow sh, ShadowEntry *__restrict__ out) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= sh.base[sh.depths]) return;
    int k = 0;
    while (k + 1 < sh.depths && e >= sh.base[k + 1]) ++k;
    const int level = shadow_level(sh, k), dep = k == 0 ? sh.d0 : 3, clevel = level + dep;
    const int64_t first = int64_t(1) << (level - 1), cfirst = int64_t(1) << (clevel - 1);
    const int64_t idx = first + (int64_t)(e - sh.base[k]); // implicit index of the node
    const N self = load_vol<N>(nodes + (idx - level_skips(tree.levels, tree.virtual_leaves, level) - 1));
    const int64_t creal = level_num_real(tree.levels, tree.virtual_leaves, clevel);
    const int64_t cskips = level_skips(tree.levels, tree.virtual_leaves, clevel);
    ShadowEntry en{};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = (float)self.lo[a], up = (float)self.up[a];
        float step = (up - lo) / 255.0f;
        if (!(step >= 0.0f) || !(step <= 3.0e38f)) step = 0.0f; // (NaN / inf boxes: every child dequantises to NaN / lo: see below)
        // the top of the frame must reach the node's own upper bound despite the roundings
        for (int it = 0; it < 64 && shadow_dequant(lo, step, 255u) < up; ++it) step = __int_as_float(__float_as_int(step) + 1); // (next float up: step is finite and >= 0)
        en.lo[a] = lo;
        en.step[a] = step;
    }
    uint32_t valid = 0;
    for (int c = 0; c < (1 << dep); ++c) {
        const int64_t ci = (idx << dep) | c;
        if (ci - cfirst >= creal) continue; // virtual child
        const N ch = load_vol<N>(nodes + (ci - cskips - 1));
        bool ok = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = en.lo[a], step = en.step[a];
            const float clo = (float)ch.lo[a], cup = (float)ch.up[a];
            // largest q with dequant(q) <= clo, smallest q with dequant(q) >= cup (verified with the walk's own expression)
            int ql = step > 0.0f ? (int)((clo - lo) / step) : 0;
            ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
            while (ql > 0 && !(shadow_dequant(lo, step, (uint32_t)ql) <= clo)) --ql;
            int qu = step > 0.0f ? (int)((cup - lo) / step) : 0;
            qu = qu < 0 ? 0 : (qu > 255 ? 255 : qu);
            while (qu < 255 && !(shadow_dequant(lo, step, (uint32_t)qu) >= cup)) ++qu;
            // a frame that cannot bracket the child (NaN or infinite coordinates): the entry says so and the walk treats
            // the child as hit unconditionally (conservative)
            ok = ok && shadow_dequant(lo, step, (uint32_t)ql) <= clo && shadow_dequant(lo, step, (uint32_t)qu) >= cup;
            en.q[c][a] = (uint8_t)ql;
            en.q[c][3 + a] = (uint8_t)qu;
        }
        valid |= 1u << c;
        if (!ok) valid |= 1u << (8 + c); // bits 8 .. 15: "always descend"
    }
    en.valid = valid;
    out[e] = en;
}

template <class L, class N, class I, bool WRITE>
__global__ __launch_bounds__(64) void lvt_rays_wide_kernel(Args<L, N, I> a, PairCache<I> cache, int ray_block, RayShadow sh) {
    using T = typename L::elt;
    static_assert(std::is_same<T, float>::value && std::is_same<typename N::elt, float>::value, "the shadow is single precision");
    struct Entry {
        IndexPair<I> pair;
        I meta;
    };
    __shared__ int s_fill;
    const int lane = threadIdx.x;
    const int64_t first_item = (int64_t)blockIdx.x * ray_block;
    const int64_t left = a.n_items - first_item;
    const int items_here = (int)(left < ray_block ? left : ray_block);
    char *region = cache.K > 0 && items_here > 0 ? (char *)(cache.slots + first_item * (int64_t)cache.K) : nullptr;
    const int entry_cap = region ? (int)(((int64_t)items_here * cache.K * (int64_t)sizeof(IndexPair<I>) - 16) / (int64_t)sizeof(Entry)) : 0;
    Entry *entries = (Entry *)(region + 16);
    if (lane == 0) s_fill = 0;
    __builtin_amdgcn_wave_barrier();
    if constexpr (WRITE) {
        if (a.guard_total != nullptr && load_total_uniform(a.guard_total) > a.guard_capacity) return;
        const int fill = region ? __builtin_amdgcn_readfirstlane(*(const int *)region) : -1;
        if (fill >= 0) { // serve the whole block from its cache (the hits of its REGULAR rays; the others are the binary walker's)
            for (int t = lane; t < fill; t += 64) {
                const Entry e = entries[t];
                const int64_t ray = first_item + (int64_t)(e.meta & (RAY_BLOCK_MAX - 1));
                const int64_t w0 = ray > 0 ? (int64_t)a.counts[ray - 1] : 0;
                a.contacts[w0 + (int64_t)(e.meta >> RAY_BITS)] = e.pair;
            }
            return;
        }
    }
    const int levels = (int)a.tree.levels, lp = levels - 1, K = sh.depths;
    const uint32_t vl = (uint32_t)a.tree.virtual_leaves;
    const uint32_t lp_first = 1u << (lp - 1);
    const uint32_t lp_real = lp_first - (vl >> 1);
    const uint32_t lp_skips = [&] {
        const uint32_t v = vl >> 2; // level_skips(lp) = 2v - popcount(v), v = vl >> (levels - (lp - 1))
        return 2u * v - (uint32_t)__builtin_popcount(v);
    }();
    const N *lp_nodes = a.nodes + ((int64_t)lp_first - (int64_t)lp_skips - 1);
    const uint32_t n_leaves = (uint32_t)a.tree.real_leaves;
    const ShadowEntry *shadow = (const ShadowEntry *)a.shadow;

    // per-lane ray state
    T p[3] = {0, 0, 0}, d[3] = {0, 0, 0}, inv[3] = {0, 0, 0};
    int ray = -1;        // ray-in-block this lane walks (-1: idle)
    int k = 0;           // wide level of the current entry
    uint32_t idx = 1;    // its node's implicit (binary) index
    uint32_t todo = 0;   // children of the current entry still to visit (bit c)
    uint64_t pend = 0;   // byte j: children of the path's entry at wide level j still to visit
    bool fetch = false;  // the current entry has not been fetched yet
    int64_t w = 0, cnt = 0;
    bool meta_bad = false;
    int next = 0;

    auto emit = [&](I lidx, uint32_t lpos) {
        const IndexPair<I> c2{a.positions ? (I)(lpos + 1u) : lidx, (I)(first_item + ray + 1)};
        if constexpr (WRITE) {
            a.contacts[w++] = c2;
        } else {
            if (region) {
                const int slot = atomicAdd(&s_fill, 1);
                if (cnt >= ((int64_t)1 << (sizeof(I) * 8 - 1 - RAY_BITS))) meta_bad = true;
                if (slot < entry_cap) entries[slot] = Entry{c2, (I)((I)ray | ((I)cnt << RAY_BITS))};
            }
            ++cnt;
        }
    };

    for (;;) {
        // ---- refill: idle lanes take the next rays of the block (irregular ones are skipped here: the binary walker's)
        const uint64_t idle = __builtin_amdgcn_ballot_w64(ray < 0);
        if (idle != 0 && next < items_here) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const int mine = next + rank;
            if (ray < 0 && mine < items_here) {
                const int64_t item = first_item + mine;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    p[j] = a.points[3 * item + j];
                    d[j] = a.dirs[3 * item + j];
                    inv[j] = T(1) / d[j];
                }
                if (ray_is_regular(p, d, inv)) {
                    ray = mine;
                    k = 0;
                    idx = 1u;
                    todo = 0;
                    pend = 0;
                    fetch = true;
                    cnt = 0;
                    if constexpr (WRITE) w = item > 0 ? (int64_t)a.counts[item - 1] : 0;
                }
            }
            const int taken = __popcll(idle);
            next = next + taken < items_here ? next + taken : items_here;
        }
        if (__builtin_amdgcn_ballot_w64(ray >= 0) == 0) {
            if (next >= items_here) break;
            continue; // (every lane drew an irregular ray: draw again)
        }
        // ---- walk
        for (;;) {
            if (ray >= 0) {
                if (fetch) {
                    // one 80-byte entry: the node's frame and its (up to) eight descendants three levels down
                    fetch = false;
                    const ShadowEntry en = shadow[sh.base[k] + (idx - (1u << (shadow_level(sh, k) - 1)))];
                    uint32_t hits = 0;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        N b;
#pragma unroll
                        for (int ax = 0; ax < 3; ++ax) {
                            b.lo[ax] = shadow_dequant(en.lo[ax], en.step[ax], en.q[c][ax]);
                            b.up[ax] = shadow_dequant(en.lo[ax], en.step[ax], en.q[c][3 + ax]);
                        }
                        const bool h = isintersection_inv(b, p, inv); // (the reference's slab function: monotone, see above)
                        hits |= (h || ((en.valid >> (8 + c)) & 1u)) ? (1u << c) : 0u;
                    }
                    todo = hits & en.valid & 0xffu;
                }
                if (todo != 0) {
                    const int c = __builtin_ctz(todo);
                    todo &= todo - 1;
                    const int dep = k == 0 ? sh.d0 : 3;
                    const uint32_t child = (idx << dep) | (uint32_t)c;
                    if (k + 1 < K) { // descend: remember what is left here
                        pend = (pend & ~((uint64_t)0xff << (8 * k))) | ((uint64_t)todo << (8 * k));
                        k += 1;
                        idx = child;
                        fetch = true;
                    } else {
                        // a candidate leaf parent: its EXACT box and its two leaves, fetched together, tested exactly
                        const uint32_t j = child - lp_first; // within level lp (real: the entry's valid mask)
                        const N pb = load_vol<N>(lp_nodes + j);
                        const uint32_t pos = 2u * j;
                        const bool has_b = pos + 1u < n_leaves;
                        const char *rec = a.leaves + (int64_t)pos * a.lay.stride;
                        const char *rec_b = has_b ? rec + a.lay.stride : rec;
                        const L la = load_vol<L>(rec), lb = load_vol<L>(rec_b);
                        const I ia = load_index<I>(rec, a.lay), ib = load_index<I>(rec_b, a.lay);
                        if (isintersection_inv(pb, p, inv)) { // raytrace/lvt:205-221 at the leaf parents' level
                            bool h0 = isintersection(la, p, d), h1 = has_b && isintersection(lb, p, d);
                            if (a.narrow == IBVH_NARROW_RAY_ORIGIN_OUTSIDE) {
                                h0 = h0 && origin_outside(la, p);
                                h1 = h1 && origin_outside(lb, p);
                            }
                            if (h0) emit(ia, pos);
                            if (h1) emit(ib, pos + 1u);
                        }
                    }
                } else if (!fetch) {
                    // this entry is exhausted: back to the deepest ancestor with children left
                    while (k > 0 && ((pend >> (8 * (k - 1))) & 0xffu) == 0) {
                        idx >>= (k - 1 == 0 ? sh.d0 : 3);
                        k -= 1;
                    }
                    if (k == 0) { // ray finished
                        if constexpr (!WRITE) a.counts[first_item + ray] = (I)cnt;
                        ray = -1;
                    } else {
                        idx >>= (k - 1 == 0 ? sh.d0 : 3);
                        k -= 1;
                        todo = (uint32_t)(pend >> (8 * k)) & 0xffu;
                        pend &= ~((uint64_t)0xff << (8 * k));
                    }
                }
            }
            const uint64_t idle_now = __builtin_amdgcn_ballot_w64(ray < 0);
            if (idle_now == ~(uint64_t)0) break;
            if (next < items_here && __popcll(idle_now) >= 16) break;
        }
    }
    if constexpr (!WRITE) {
        __builtin_amdgcn_wave_barrier();
        const bool ok = __builtin_amdgcn_ballot_w64(meta_bad) == 0;
        if (region && lane == 0) *(int *)region = (s_fill <= entry_cap && ok) ? s_fill : -1;
    }
}

// ---- (3c) rays BINNED BY SUBTREE: the bottom of the tree is walked out of LDS ---------------------------------------
// What the per-lane walk above costs on config 3 (SQ / TCC counters, round 4): 1,700 wave-steps per wave at 32 % of the lanes
// busy, 138 VALU + 94 SALU instructions a wave-step, and every step below level ~17 misses L2 — 237 M 128-byte lines come
// out of L2 for 48 useful bytes each (30 GB), 109 M of them out of HBM / Infinity Cache (14 GB, 31 x the algorithmic bytes).
// The RAYS are the small side (24 bytes each), so the bottom of the tree is turned node-major:
//   A. rays_top_kernel — the same per-lane walk, but only down to the CUT level K = levels - D (D = 9: subtrees of 512
//      leaves).  Levels 1 .. K are a few hundred KB: every fetch is an L2 hit.  A hit at level K is not descended into, it
//      is EMITTED as an item (ray, subtree j, ordinal of the item within its ray); items leave the wave through an LDS
//      stage in chunks (one global atomic per ~700 items).
//   B. the items are grouped by subtree: a counting sort whose tiles count in LDS first (rays_tilehist_kernel, rays_binscan_kernel,
//      rays_scatter_kernel) — the buckets are far from even (config 3: 1,400 items on average, 190,000 in the busiest) and
//      same-address global atomics serialise at ~11 ns each.  An item's rank in (ray, ordinal) order, g, is known once the
//      per-ray item counts are scanned and travels with it.
//   C. rays_subtree_kernel — a workgroup copies one subtree's node levels and leaves into LDS (the tree is read about once
//      per call, coalesced), then its lanes take up to RAYSUB_CHUNK of the subtree's items (busy subtrees are shared by several
//      workgroups) and finish the walk below the subtree's root out of LDS: the random access that remains is the item's
//      24-byte ray (bucket entries that carry the ray — one coalesced 32-byte read — cost the scatter more than they save
//      here: 0.18 -> 0.35 ms against 1.46 -> 1.44).  A hit is counted for its item AND kept as a record (pair, g, rank
//      within the item) in a list.
//   D. hits per item in g order -> inclusive scan -> an item's hits go to [scan[g-1], scan[g]); the per-ray counts the
//      entry points return are differences of that scan at the rays' item boundaries.  The writing pass only moves the kept
//      records to scan[g-1] + rank (rays_place_kernel); if the record list overflowed it walks the subtrees again instead.
// Order: a ray's walk visits subtrees left to right and emits its items in that order, so (ray, ordinal) order followed
// by the walk's own order inside the subtree is exactly the order in which the reference's loop
// (raytrace/leaf_vs_tree/leaf_vs_tree.jl:187-225) reports the ray's hits — the walk is the same walk, cut in two at
// level K; no property of the ray is assumed, so irregular rays (zero / infinite / NaN components) take this path too.
// The item list has a fixed capacity inside the caller's scratch (ibvh_rays_scratch_bytes: 16 items per ray; config 3
// emits 10.4); a call that overflows it raises *flag and every later kernel of the path returns at once, while the
// binary walker — launched behind it in every call, gated on that flag — serves the call instead.  No host round trip.
// The slab test of isintersection.jl:1-33 for a ray and a box that cannot produce a NaN: the ray is REGULAR (finite origin,
// finite non-zero direction with finite non-zero reciprocal: ray_is_regular) and the box holds no NaN — then every
// (bound - p) * inv is a number (possibly infinite), and on numbers the reference's `a < b ? a : b` / `a > b ? a : b` and
// the hardware's v_min_f32 / v_max_f32 differ at most in the sign of a zero, which no later min, max or comparison can
// tell apart: the same boolean, for half the instructions (packed subtract / multiply on the six bounds as they lie in
// memory, v_min3 / v_max3).  Rays and boxes that do not qualify take isintersection_inv.  Used by rays_top_kernel
// (0.72 -> 0.65 ms on config 3).
#ifdef IBVH_RAYS_NO_FAST_SLAB // (development builds: the walks without the second code path, tools/build_variant.sh)
constexpr bool kRaysFastSlab = false;
#else
constexpr bool kRaysFastSlab = true;
#endif
typedef float ray_f2 __attribute__((ext_vector_type(2)));
struct RayPk {
    ray_f2 p01, p20, p12, i01, i20, i12; // origin and reciprocals paired like a BBox{Float32}'s six floats: lo0 lo1 | lo2 up0 | up1 up2
};
IBVH_D RayPk ray_pk(const float *p, const float *inv) {
    return RayPk{ray_f2{p[0], p[1]}, ray_f2{p[2], p[0]}, ray_f2{p[1], p[2]}, ray_f2{inv[0], inv[1]}, ray_f2{inv[2], inv[0]}, ray_f2{inv[1], inv[2]}};
}
IBVH_D bool slab_fast(const BBox<float> &b, const RayPk &r) {
    const ray_f2 a = (ray_f2{b.lo[0], b.lo[1]} - r.p01) * r.i01; // t(lo0), t(lo1)
    const ray_f2 c = (ray_f2{b.lo[2], b.up[0]} - r.p20) * r.i20; // t(lo2), t(up0)
    const ray_f2 e = (ray_f2{b.up[1], b.up[2]} - r.p12) * r.i12; // t(up1), t(up2)
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(a.x, c.y), __builtin_fminf(a.y, e.x)), __builtin_fminf(c.x, e.y));
    const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(a.x, c.y), __builtin_fmaxf(a.y, e.x)), __builtin_fmaxf(c.x, e.y));
    return (tmin <= tmax) && (tmax >= 0.f);
}
IBVH_D bool word_has_nan(uint64_t w) { // either float of an 8-byte word
    return ((uint32_t)w & 0x7fffffffu) > 0x7f800000u || ((uint32_t)(w >> 32) & 0x7fffffffu) > 0x7f800000u;
}

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
    void *hit_list;             // [RAY_REGIONS][region_cap] RayHit<I>: the hits of the counting pass
    int32_t region_cap;
    void *hits;                 // [cap] of I: hits of item g; after the scan: inclusive prefix
    int32_t cap;                // 0: the path is not in use
    int32_t cut_level;          // K
    int32_t depth;              // D = levels - K: a subtree holds 2^D leaves
    int32_t subtrees;           // real nodes on level K
};

// any NaN in the node levels 1 .. K?  (a few hundred KB; decides whether rays_top_kernel may use slab_fast)
template <class N> __global__ __launch_bounds__(256) void rays_topcheck_kernel(const N *nodes, int64_t count, RayBins rb) {
    const uint64_t *w = (const uint64_t *)nodes;
    const int64_t words = count * (int64_t)(sizeof(N) / 8);
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (int64_t)gridDim.x * 256) bad |= word_has_nan(w[i]);
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) *rb.top_nan = 1;
}

constexpr int RAYTOP_STAGE = 768; // items a wave stages in LDS before it reserves room in the global list
template <class L, class N, class I>
__global__ __launch_bounds__(64) void rays_top_kernel(Args<L, N, I> a, RayBins rb, int ray_block) {
    using T = typename L::elt;
    __shared__ uint64_t s_items[RAYTOP_STAGE];
    const int lane = threadIdx.x;
    const int64_t first_item = (int64_t)blockIdx.x * ray_block;
    const int64_t left = a.n_items - first_item;
    const int items_here = (int)(left < ray_block ? left : ray_block);
    const int levels = (int)a.tree.levels;
    const uint32_t vl = (uint32_t)a.tree.virtual_leaves;
    const int K = rb.cut_level;
    const uint32_t kfirst = 1u << (K - 1);
    const int plevel = (int)a.start_level - 1;
    const int64_t roots = level_num_real(a.tree.levels, a.tree.virtual_leaves, a.start_level);
    const uint32_t pfirst = plevel >= 1 ? (1u << (plevel - 1)) : 0u;
    const uint32_t pcount = (uint32_t)((roots + 1) / 2);

    T p[3] = {0, 0, 0}, d[3] = {0, 0, 0}, inv[3] = {0, 0, 0};
    RayPk pk{};
    bool regular = true; // (idle lanes count as regular)
    constexpr bool kPacked = N::kind == IBVH_BBOX && std::is_same<T, float>::value; // (slab_fast is single precision)
    const bool top_clean = kRaysFastSlab && kPacked && *rb.top_nan == 0; // (the knob rays_fast_slab = 0 stores -1 there)
    int ray = -1;
    uint32_t pi = 0, inode = 0, pend = 0, ord = 0;
    int level = 0;
    int next = 0; // wave-uniform: rays of the block handed out so far
    int fill = 0; // wave-uniform: items staged

    auto node_hit = [&](const N &n) {
        if constexpr (N::kind == IBVH_BBOX) return isintersection_inv(n, p, inv);
        else return isintersection(n, p, d);
    };
    auto flush = [&]() {
        if (fill == 0) return;
        __syncthreads(); // (one wave: orders the stage's writes before the reads below)
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(rb.cursor, (unsigned long long)fill);
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base + (unsigned long long)fill > (unsigned long long)rb.cap) {
            if (lane == 0) *rb.flag = 1;
        } else {
            for (int t = lane; t < fill; t += 64) {
                const uint64_t it = s_items[t];
                rb.items[base + t] = it;
            }
        }
        __syncthreads();
        fill = 0;
    };

    for (;;) {
        const uint64_t idle = __builtin_amdgcn_ballot_w64(ray < 0);
        if (idle != 0 && next < items_here) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const int mine = next + rank;
            if (ray < 0 && mine < items_here) {
                const int64_t item = first_item + mine;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    p[k] = a.points[3 * item + k];
                    d[k] = a.dirs[3 * item + k];
                    inv[k] = T(1) / d[k];
                }
                if constexpr (kPacked) {
                    regular = ray_is_regular(p, d, inv);
                    pk = ray_pk(p, inv);
                }
                ray = mine;
                pi = 0;
                inode = pfirst;
                level = plevel;
                pend = 0;
                ord = 0;
            }
            const int taken = __popcll(idle);
            next = next + taken < items_here ? next + taken : items_here;
        }
        if (__builtin_amdgcn_ballot_w64(ray >= 0) == 0) break; // (every lane idle after a refill: the block is used up)
        // (wave-uniform, fixed between refills: slab_fast serves the wave while all its rays are regular)
        const bool fast = top_clean && __builtin_amdgcn_ballot_w64(ray >= 0 && !regular) == 0;
        for (;;) {
            bool e0 = false, e1 = false;
            uint64_t it0 = 0, it1 = 0;
            if (ray >= 0) {
                const int cl = level + 1;
                const uint32_t c0 = 2u * inode, c1 = c0 + 1u;
                const uint32_t first = 1u << (cl - 1);
                const uint32_t nreal = first - (uint32_t)((uint64_t)vl >> (levels - cl));
                const bool real0 = c0 != 0u, real1 = (c1 - first) < nreal; // (c0 == 0: the pseudo node above the root)
                const uint64_t v = (uint64_t)vl >> (levels - cl + 1);
                const uint32_t sk = (uint32_t)(2 * v) - (uint32_t)__popcll(v); // level_skips(cl)
                const N *np = a.nodes + ((int64_t)c0 - (int64_t)sk - 1);
                struct Two {
                    N a, b;
                };
                Two ch;
                if (real0 && real1) {
                    __builtin_memcpy(&ch, __builtin_assume_aligned(np, 8), sizeof(Two));
                } else {
                    ch.a = load_vol<N>(real0 ? np : np + 1);
                    ch.b = ch.a;
                }
                bool h0, h1;
                if constexpr (kPacked) {
                    if (fast) {
                        h0 = real0 && slab_fast(ch.a, pk);
                        h1 = real1 && slab_fast(ch.b, pk);
                    } else {
                        h0 = real0 && node_hit(ch.a);
                        h1 = real1 && node_hit(ch.b);
                    }
                } else {
                    h0 = real0 && node_hit(ch.a);
                    h1 = real1 && node_hit(ch.b);
                }
                bool descended = false;
                if (cl == K) { // the cut: hits become items, left before right
                    const uint64_t r64 = (uint64_t)(first_item + ray);
                    e0 = h0;
                    it0 = r64 | ((uint64_t)(c0 - kfirst) << 32) | ((uint64_t)ord << 48);
                    ord += h0 ? 1u : 0u;
                    e1 = h1;
                    it1 = r64 | ((uint64_t)(c1 - kfirst) << 32) | ((uint64_t)ord << 48);
                    ord += h1 ? 1u : 0u;
                } else if (h0) {
                    if (h1) pend |= 1u << cl;
                    inode = c0;
                    level = cl;
                    descended = true;
                } else if (h1) {
                    inode = c1;
                    level = cl;
                    descended = true;
                }
                if (!descended) {
                    if (pend != 0) {
                        const int pl = 31 - __builtin_clz(pend);
                        pend &= ~(1u << pl);
                        inode = (inode >> (level - pl)) | 1u;
                        level = pl;
                    } else if (++pi < pcount) {
                        inode = pfirst + pi;
                        level = plevel;
                    } else {
                        rb.ray_items[first_item + ray] = (int32_t)ord;
                        ray = -1;
                        regular = true;
                    }
                }
            }
            const uint64_t m0 = __builtin_amdgcn_ballot_w64(e0), m1 = __builtin_amdgcn_ballot_w64(e1);
            if ((m0 | m1) != 0) {
                const int n0 = __popcll(m0);
                if (e0) s_items[fill + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))] = it0;
                if (e1) s_items[fill + n0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u))] = it1;
                fill += n0 + __popcll(m1);
                if (fill > RAYTOP_STAGE - 128) flush();
            }
            const uint64_t idle_now = __builtin_amdgcn_ballot_w64(ray < 0);
            if (idle_now == ~(uint64_t)0) break;
            if (next < items_here && __popcll(idle_now) >= 16) break;
        }
    }
    flush();
}

// items per subtree.  The distribution is far from even (config 3: mean 1,400 items, the busiest subtree 190,000) and
// same-address global atomics serialise (~11 ns each), so a tile of items is counted in LDS first and every non-empty bin
// of the tile costs ONE global atomic.
// (a tile is 16 items a thread; 1,024-thread tiles amortise the walk over the bins: config 3, 1e7 items, 14 k bins: the two
// kernels 0.36 -> 0.23 ms against 256-thread tiles; small batches keep the small tiles so that the grid still fills the chip)
constexpr int RAYTILE_IPT = 16;
constexpr int RAYSUB_CHUNK = 4096; // items of one rays_subtree_kernel workgroup: busy subtrees are shared by several (1,024: 2 % slower on config 3)
template <int RAYTILE_TPB> __global__ __launch_bounds__(RAYTILE_TPB) void rays_tilehist_kernel(RayBins rb) {
    constexpr int RAYTILE = RAYTILE_TPB * RAYTILE_IPT;
    extern __shared__ uint32_t s_hist[];
    if (*rb.flag != 0) return;
    const unsigned long long cur = *rb.cursor;
    const int64_t n = (int64_t)(cur < (unsigned long long)rb.cap ? cur : (unsigned long long)rb.cap);
    const int64_t base = (int64_t)blockIdx.x * RAYTILE;
    if (base >= n) return;
    const int64_t end = base + RAYTILE < n ? base + RAYTILE : n;
    for (int b = threadIdx.x; b < rb.subtrees; b += RAYTILE_TPB) s_hist[b] = 0;
    __syncthreads();
    for (int64_t i = base + threadIdx.x; i < end; i += RAYTILE_TPB) atomicAdd(&s_hist[(uint32_t)(rb.items[i] >> 32) & 0xffffu], 1u);
    __syncthreads();
    for (int b = threadIdx.x; b < rb.subtrees; b += RAYTILE_TPB) {
        const uint32_t c = s_hist[b];
        if (c != 0) atomicAdd(&rb.bin_count[b], c);
    }
}

// one workgroup: bin_start = exclusive prefix of bin_count, the table of rays_subtree_kernel's workgroups (one per
// RAYSUB_CHUNK items of a bucket), the item count clipped to the capacity
__global__ __launch_bounds__(1024) void rays_binscan_kernel(RayBins rb) {
    __shared__ unsigned long long s_w[16];
    const unsigned long long cur = *rb.cursor;
    if (threadIdx.x == 0) *rb.n_items = (int32_t)(cur < (unsigned long long)rb.cap ? cur : (unsigned long long)rb.cap);
    if (*rb.flag != 0) {
        if (threadIdx.x == 0) *rb.n_chunks = 0;
        return;
    }
    const int per = (rb.subtrees + 1023) / 1024;
    const int b = (int)threadIdx.x * per;
    // items in the low word, chunks in the high word: one scan for both (items <= 2^30)
    unsigned long long sum = 0;
    for (int k = 0; k < per; ++k) {
        const uint32_t c = b + k < rb.subtrees ? rb.bin_count[b + k] : 0u;
        sum += (unsigned long long)c | ((unsigned long long)((c + RAYSUB_CHUNK - 1) / RAYSUB_CHUNK) << 32);
    }
    unsigned long long inc = sum;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    unsigned long long run = inc - sum;
    for (int k = 0; k < w; ++k) run += s_w[k];
    for (int k = 0; k < per; ++k) {
        if (b + k < rb.subtrees) {
            const uint32_t c = rb.bin_count[b + k];
            const uint32_t chunks = (c + RAYSUB_CHUNK - 1) / RAYSUB_CHUNK, c0 = (uint32_t)(run >> 32);
            rb.bin_start[b + k] = (uint32_t)run;
            for (uint32_t q = 0; q < chunks; ++q) rb.chunk_tab[c0 + q] = make_uint2((uint32_t)(b + k), q);
            run += (unsigned long long)c | ((unsigned long long)chunks << 32);
            if (b + k == rb.subtrees - 1) {
                rb.bin_start[rb.subtrees] = (uint32_t)run;
                *rb.n_chunks = (int32_t)(run >> 32);
            }
        }
    }
}

// items -> buckets by subtree; an item's rank g in (ray, ordinal) order comes from the scanned per-ray item counts.  Same
// tiles and the same LDS counting as rays_tilehist_kernel: a tile reserves its share of a bucket with one global atomic.
template <int RAYTILE_TPB> __global__ __launch_bounds__(RAYTILE_TPB) void rays_scatter_kernel(RayBins rb) {
    constexpr int RAYTILE = RAYTILE_TPB * RAYTILE_IPT;
    extern __shared__ uint32_t s_hist[];
    if (*rb.flag != 0) return;
    const int64_t n = *rb.n_items;
    const int64_t base = (int64_t)blockIdx.x * RAYTILE;
    if (base >= n) return;
    constexpr int IPT = RAYTILE / RAYTILE_TPB;
    for (int b = threadIdx.x; b < rb.subtrees; b += RAYTILE_TPB) s_hist[b] = 0;
    __syncthreads();
    uint32_t rank[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int64_t i = base + k * RAYTILE_TPB + threadIdx.x;
        rank[k] = i < n ? atomicAdd(&s_hist[(uint32_t)(rb.items[i] >> 32) & 0xffffu], 1u) : 0u;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < rb.subtrees; b += RAYTILE_TPB) {
        const uint32_t c = s_hist[b];
        if (c != 0) s_hist[b] = rb.bin_start[b] + atomicAdd(&rb.bin_cursor[b], c);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int64_t i = base + k * RAYTILE_TPB + threadIdx.x;
        if (i < n) {
            const uint64_t it = rb.items[i];
            const uint32_t ray = (uint32_t)it, j = (uint32_t)(it >> 32) & 0xffffu, ord = (uint32_t)(it >> 48);
            const uint32_t g = (ray > 0 ? (uint32_t)rb.ray_items[ray - 1] : 0u) + ord;
            rb.bucket[s_hist[j] + rank[k]] = make_uint2(ray, g);
        }
    }
}

constexpr int RAYSUB_TPB = 256;
// waves of a workgroup that WALK (all of them load).  A wave lives as long as its longest item (config 3: 12 steps on
// average, ~150 for the longest of a bucket), so four walkers with 256 items each keep only ~20 % of their lanes busy — but
// fewer walkers lose more to latency than they gain in lane use (config 3, subtree pass: 1.47 ms with four, 2.09 ms with one
// per 512 items, 2.47 ms with one)
constexpr int RAYSUB_WALKERS = 4;
constexpr int RAYSUB_STAGE = 64;  // hit records a wave stages in LDS (a step adds at most 64 left and 64 right hits: two appends)
constexpr int RAY_REGIONS = 256;  // the hit list is RAY_REGIONS lists with a cursor each: same-address atomics serialise
// a hit of the counting pass: the pair as it will be reported, the item it belongs to and its rank within the item; the
// writing pass puts it at scan[g - 1] + k (rays_place_kernel) instead of walking again
template <class I> struct RayHit {
    IndexPair<I> pair;
    uint32_t g, k;
};
IBVH_HD size_t rays_subtree_lds(int depth, size_t node_bytes, size_t leaf_bytes, size_t index_bytes, size_t hit_bytes, bool write) {
    const size_t S = (size_t)1 << depth;
    size_t o = (S * node_bytes + 15) & ~(size_t)15;
    o += (S * leaf_bytes + 15) & ~(size_t)15;
    o += (S * index_bytes + 15) & ~(size_t)15;
    if (!write) o += (size_t)RAYSUB_WALKERS * RAYSUB_STAGE * hit_bytes;
    return o;
}

template <class L, class N, class I, bool WRITE>
__global__ __launch_bounds__(RAYSUB_TPB) void rays_subtree_kernel(Args<L, N, I> a, RayBins rb) {
    using T = typename L::elt;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ uint32_t s_next;
    if (*rb.flag != 0) return;
    if constexpr (WRITE) {
        if (*rb.reflag == 0) return; // the counting pass kept every hit: rays_place_kernel writes them
        if (a.guard_total != nullptr && load_total_uniform(a.guard_total) > a.guard_capacity) return;
    }
    if ((int32_t)blockIdx.x >= *rb.n_chunks) return;
    const uint2 chunk = rb.chunk_tab[blockIdx.x];
    const uint32_t j = chunk.x; // the subtree; this workgroup takes items [chunk.y * RAYSUB_CHUNK, ...) of its bucket
    const uint32_t b0 = rb.bin_start[j] + chunk.y * RAYSUB_CHUNK;
    const uint32_t n_here = rb.bin_start[j + 1] - b0 < (uint32_t)RAYSUB_CHUNK ? rb.bin_start[j + 1] - b0 : (uint32_t)RAYSUB_CHUNK;
    const int levels = (int)a.tree.levels, K = rb.cut_level, D = rb.depth;
    const uint32_t S = 1u << D;
    const uint32_t vl = (uint32_t)a.tree.virtual_leaves;
    const uint32_t real_leaves = (uint32_t)a.tree.real_leaves;
    // LDS: nodes by heap index t (1 = the subtree's root, never read; children of t are 2t, 2t + 1), the leaves' volumes,
    // what a hit reports for them (user index, or 1-based position), the waves' hit stages
    size_t o = 0;
    N *s_nodes = (N *)s_raw;
    o += ((size_t)S * sizeof(N) + 15) & ~(size_t)15;
    L *s_leaves = (L *)(s_raw + o);
    o += ((size_t)S * sizeof(L) + 15) & ~(size_t)15;
    I *s_index = (I *)(s_raw + o);
    o += ((size_t)S * sizeof(I) + 15) & ~(size_t)15;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    RayHit<I> *s_stage = (RayHit<I> *)(s_raw + o) + wave * RAYSUB_STAGE;
    {
        constexpr uint32_t WPN = sizeof(N) / 8;
        const uint64_t *src = (const uint64_t *)a.nodes;
        uint64_t *dst = (uint64_t *)s_nodes;
#pragma unroll 4
        for (uint32_t wd = 2 * WPN + tid; wd < S * WPN; wd += RAYSUB_TPB) {
            const uint32_t h = wd / WPN, part = wd - h * WPN;
            const int dl = 31 - __builtin_clz(h), level = K + dl;
            const uint32_t gi = (j << dl) + (h - (1u << dl)), first = 1u << (level - 1);
            const uint32_t nreal = first - (uint32_t)((uint64_t)vl >> (levels - level));
            const uint64_t v = (uint64_t)vl >> (levels - level + 1);
            const uint32_t sk = (uint32_t)(2 * v) - (uint32_t)__popcll(v);
            if (gi < nreal) dst[wd] = src[((int64_t)first + (int64_t)gi - (int64_t)sk - 1) * WPN + part];
        }
        const uint32_t g0 = j << D;
        const uint32_t cnt = g0 >= real_leaves ? 0u : (real_leaves - g0 < S ? real_leaves - g0 : S);
        for (uint32_t t = tid; t < cnt; t += RAYSUB_TPB) {
            const char *rec = a.leaves + (int64_t)(g0 + t) * a.lay.stride;
            s_leaves[t] = load_vol<L>(rec);
            s_index[t] = a.positions ? (I)(g0 + t + 1u) : load_index<I>(rec, a.lay);
        }
    }
    if (tid == 0) s_next = 0;
    __syncthreads();
    if (wave >= RAYSUB_WALKERS) return;

    const I *hits = (const I *)rb.hits;
    const uint32_t region = blockIdx.x & (RAY_REGIONS - 1);
    T p[3] = {0, 0, 0}, d[3] = {0, 0, 0}, inv[3] = {0, 0, 0};
    bool busy = false, more = true; // more: wave-uniform, the chunk may still hold items
    uint32_t ray = 0, g = 0, tn = 1, pend = 0;
    int dl = 0;
    int64_t w = 0;
    uint32_t cnt = 0;
    int fill = 0; // wave-uniform: hit records staged
    auto node_hit = [&](const N &n) {
        if constexpr (N::kind == IBVH_BBOX) return isintersection_inv(n, p, inv);
        else return isintersection(n, p, d);
    };
    auto flush = [&]() {
        if (fill == 0) return;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&rb.region_cursor[region], (uint32_t)fill);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if ((uint64_t)base + (uint32_t)fill > (uint64_t)rb.region_cap) {
            if (lane == 0) *rb.reflag = 1;
        } else {
            RayHit<I> *dst = (RayHit<I> *)rb.hit_list + (size_t)region * rb.region_cap + base;
            for (int t = lane; t < fill; t += 64) dst[t] = s_stage[t];
        }
        fill = 0;
    };
    for (;;) {
        const uint64_t idle = __builtin_amdgcn_ballot_w64(!busy);
        if (idle != 0 && more) {
            const int want = __popcll(idle);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_next, (uint32_t)want);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            more = base + (uint32_t)want < n_here;
            const uint32_t mine = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (!busy && mine < n_here) {
                const uint2 e = rb.bucket[b0 + mine];
                ray = e.x;
                g = e.y;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    p[k] = a.points[3 * (int64_t)ray + k];
                    d[k] = a.dirs[3 * (int64_t)ray + k];
                    inv[k] = T(1) / d[k];
                }
                busy = true;
                tn = 1;
                dl = 0;
                pend = 0;
                cnt = 0;
                if constexpr (WRITE) w = g > 0 ? (int64_t)hits[g - 1] : 0;
            }
        }
        if (__builtin_amdgcn_ballot_w64(busy) == 0) break;
        for (;;) {
            bool h0 = false, h1 = false;
            uint32_t li = 0;
            if (busy) {
                const int cd = dl + 1;
                const uint32_t c0 = 2u * tn, c1 = c0 + 1u;
                bool descended = false;
                if (cd == D) { // the two leaves under tn
                    li = c0 - S;
                    const bool real1 = (j << D) + li + 1u < real_leaves; // (a real parent's left child is real)
                    const L la = s_leaves[li], lb = s_leaves[li + 1];
                    h0 = isintersection(la, p, d);
                    h1 = real1 && isintersection(lb, p, d);
                    if (a.narrow == IBVH_NARROW_RAY_ORIGIN_OUTSIDE) { // raytrace/lvt:194: isintersection(...) && narrow(leaf, p, d)
                        h0 = h0 && origin_outside(la, p);
                        h1 = h1 && origin_outside(lb, p);
                    }
                    if constexpr (WRITE) {
                        if (h0) a.contacts[w++] = IndexPair<I>{s_index[li], (I)((int64_t)ray + 1)};
                        if (h1) a.contacts[w++] = IndexPair<I>{s_index[li + 1], (I)((int64_t)ray + 1)};
                    }
                } else {
                    const int level = K + cd;
                    const uint32_t nreal = (1u << (level - 1)) - (uint32_t)((uint64_t)vl >> (levels - level));
                    const bool real1 = (j << cd) + (c1 - (1u << cd)) < nreal;
                    const N na = s_nodes[c0], nb = s_nodes[c1];
                    // (the packed slab test of rays_top_kernel as a second code path in this loop made it slower: 1.65 against 1.55 ms)
                    const bool n0 = node_hit(na), n1 = real1 && node_hit(nb);
                    if (n0) {
                        if (n1) pend |= 1u << cd;
                        tn = c0;
                        dl = cd;
                        descended = true;
                    } else if (n1) {
                        tn = c1;
                        dl = cd;
                        descended = true;
                    }
                }
                if (!descended) {
                    if (pend != 0) {
                        const int pl = 31 - __builtin_clz(pend);
                        pend &= ~(1u << pl);
                        tn = (tn >> (dl - pl)) | 1u;
                        dl = pl;
                    } else {
                        if constexpr (!WRITE) ((I *)rb.hits)[g] = (I)(cnt + (h0 ? 1u : 0u) + (h1 ? 1u : 0u));
                        busy = false;
                    }
                }
            }
            if constexpr (!WRITE) {
                const uint64_t m0 = __builtin_amdgcn_ballot_w64(h0), m1 = __builtin_amdgcn_ballot_w64(h1);
                if ((m0 | m1) != 0) {
                    const int n0 = __popcll(m0), n1 = __popcll(m1);
                    if (n0 != 0) {
                        if (fill + n0 > RAYSUB_STAGE) flush();
                        if (h0) {
                            const int s0 = fill + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
                            s_stage[s0] = RayHit<I>{IndexPair<I>{s_index[li], (I)((int64_t)ray + 1)}, g, cnt};
                        }
                        fill += n0;
                    }
                    if (n1 != 0) {
                        if (fill + n1 > RAYSUB_STAGE) flush();
                        if (h1) {
                            const int s1 = fill + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
                            s_stage[s1] = RayHit<I>{IndexPair<I>{s_index[li + 1], (I)((int64_t)ray + 1)}, g, cnt + (h0 ? 1u : 0u)};
                        }
                        fill += n1;
                    }
                    cnt += (h0 ? 1u : 0u) + (h1 ? 1u : 0u);
                }
            }
            const uint64_t idle_now = __builtin_amdgcn_ballot_w64(!busy);
            if (idle_now == ~(uint64_t)0) break;
            if (more && __popcll(idle_now) >= 16) break;
        }
    }
    if constexpr (!WRITE) flush();
}

// the writing pass when the counting pass kept every hit: records -> their places in the contact list
template <class I> __global__ __launch_bounds__(256) void rays_place_kernel(RayBins rb, IndexPair<I> *contacts, const int64_t *guard_total, int64_t guard_capacity) {
    if (*rb.flag != 0 || *rb.reflag != 0) return;
    if (guard_total != nullptr && load_total_uniform(guard_total) > guard_capacity) return;
    const uint32_t region = blockIdx.y;
    const uint32_t n = rb.region_cursor[region] < (uint32_t)rb.region_cap ? rb.region_cursor[region] : (uint32_t)rb.region_cap;
    const RayHit<I> *src = (const RayHit<I> *)rb.hit_list + (size_t)region * rb.region_cap;
    const I *h = (const I *)rb.hits;
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
        const RayHit<I> r = src[t];
        const int64_t at = (r.g > 0 ? (int64_t)h[r.g - 1] : 0) + (int64_t)r.k;
        contacts[at] = r.pair;
    }
}

// per-ray hit counts from the scanned per-item hits: the difference of the scan at the ray's item boundaries
template <class I> __global__ __launch_bounds__(256) void rays_counts_kernel(RayBins rb, I *counts, int64_t n_rays) {
    if (*rb.flag != 0) return;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rays) return;
    const I *h = (const I *)rb.hits;
    const int32_t e1 = rb.ray_items[r], e0 = r > 0 ? rb.ray_items[r - 1] : 0;
    const int64_t s1 = e1 > 0 ? (int64_t)h[e1 - 1] : 0, s0 = e0 > 0 ? (int64_t)h[e0 - 1] : 0;
    counts[r] = (I)(s1 - s0);
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

// scratch layout of the *_count / *_write calls:
//   [0, 64)            int64 header: [0] total contacts, [1] contact-cache slots K in use
//   [64, scan_bytes)   scan tile sums
//   [scan_bytes, ...)  contact cache: K * n_items IndexPair{I}, slot-major
inline size_t scan_scratch_bytes(int64_t n) {
    return (size_t)align_up((ceil_div(n > 0 ? n : 1, SCAN_TILE) + 8) * 8, 256);
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
                int64_t *total_host = nullptr, const int32_t *limit = nullptr) {
    int64_t nparts = ceil_div(n, SCAN_TILE);
    int64_t *totals = total_dev ? total_dev : (int64_t *)scratch; // where the device-side total goes
    int64_t *partials = (int64_t *)scratch + 8;
    IBVH_LAUNCH((scan_reduce_kernel<I>), dim3((unsigned)nparts), dim3(SCAN_TPB), 0, st, counts, n, partials, limit);
    IBVH_LAUNCH((scan_apply_kernel<I>), dim3((unsigned)nparts), dim3(SCAN_TPB), 0, st, counts, n, partials, totals, total_host, limit);
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
    p.region_cap = (cap + 255) / 256; // (RAY_REGIONS lists, as many records as items all together)
    p.off_hit_list = o, o += (size_t)p.region_cap * 256 * (bvh.types.index_type == IBVH_I64 ? 24 : 16);
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
    rb.cap = (int32_t)p.cap;
    rb.cut_level = p.cut_level;
    rb.depth = p.depth;
    rb.subtrees = p.subtrees;
    return rb;
}
// the type combinations the binned path is compiled for (one float type throughout; everything else: the binary walker)
template <class L, class N> constexpr bool kRayBinTypes = std::is_same<typename L::elt, typename N::elt>::value; // (what ray traversal asks for anyway)

template <class L, class N, class I, int MODE>
int launch(const Args<L, N, I> &a, const PairCache<I> &cache, bool write, hipStream_t st, const RayBins &rb = RayBins{}) {
    if (a.n_items == 0) return IBVH_OK;
    const bool count_work = a.work != nullptr; // (the counting pass of the COUNT instantiation; nothing else is launched)
    if (count_work && (!kWorkTypes<L, N, I> || write)) return IBVH_ERR_UNSUPPORTED;
    unsigned blocks = (unsigned)ceil_div(a.n_items, 256);
    if constexpr (MODE == MODE_RAYS) {
        // rays of one wave are not spatially coherent: each lane walks on its own
        // (A breadth-first variant with 16 lanes per ray and per-level LDS frontiers was measured in round 1: with
        // frontiers that fit it halves the time of SMALL batches (1e5 rays: 1.1 -> 0.67 ms, the heaviest ray no longer
        // walks on one lane), but rays grazing the surface outgrow any LDS slice that still allows a decent occupancy
        // and at 1e6 rays it was 2x slower than this walk, so it was dropped.)
        // one wave per workgroup: a wave's time is its heaviest ray, and a finished wave should hand its slot back at once
        // rays per wave: the largest block of 64 / 128 / 256 that still leaves ~3,000 waves in the grid (measured on config
        // 3, 1e6 rays: 5.41 / 5.09 / 4.25 ms with 64 / 128 / 256, 5.25 with 512; 1e5 rays: 2.35 / 3.18 ms with 64 / 256)
        int ray_block = 64;
        while (ray_block < 256 && a.n_items / (2 * ray_block) >= 3000) ray_block *= 2;
        const int forced_block = g_tuning.ray_block;
        if (forced_block >= 64 && forced_block <= RAY_BLOCK_MAX && (forced_block & (forced_block - 1)) == 0) ray_block = forced_block;
        const unsigned rblocks = (unsigned)ceil_div(a.n_items, (int64_t)ray_block);
        if constexpr (kRayBinTypes<L, N>) {
            if (rb.cap > 0 && !count_work) {
                // the binned path (3c); the binary walker stands by behind it, gated on the overflow flag
                static_assert(RAY_REGIONS == 256, "rays_bin_plan sizes the hit list for 256 regions");
                const size_t lds = rays_subtree_lds(rb.depth, sizeof(N), sizeof(L), sizeof(I), sizeof(RayHit<I>), write);
                Args<L, N, I> standby = a;
                standby.gate = rb.flag;
                standby.shadow = nullptr;
                const PairCache<I> none{nullptr, 0};
                if (!write) {
                    IBVH_HIP_CHECK(hipMemsetAsync(rb.cursor, 0, 2048, st));
                    IBVH_HIP_CHECK(hipMemsetAsync(rb.bin_count, 0, (size_t)((char *)rb.items - (char *)rb.bin_count), st)); // counts, starts, cursors
                    if (!g_tuning.rays_fast_slab) IBVH_HIP_CHECK(hipMemsetAsync(rb.top_nan, 0xff, 4, st)); // (-1: no fast slab test anywhere)
                    if constexpr (N::kind == IBVH_BBOX && std::is_same<typename N::elt, float>::value) {
                        const int64_t top_first = level_start(a.tree.levels, a.tree.virtual_leaves, a.built_level) - 1; // (memory index of the first node that exists)
                        const int64_t top_count = level_start(a.tree.levels, a.tree.virtual_leaves, rb.cut_level + 1) - 1 - top_first;
                        IBVH_LAUNCH((rays_topcheck_kernel<N>), dim3((unsigned)(ceil_div(top_count * 3, 256) < 256 ? ceil_div(top_count * 3, 256) : 256)), dim3(256), 0,
                                    st, a.nodes + top_first, top_count, rb);
                    }
                    IBVH_LAUNCH((rays_top_kernel<L, N, I>), dim3(rblocks), dim3(64), 0, st, a, rb, ray_block);
                    if (int e = scan_counts<int32_t>(rb.ray_items, a.n_items, nullptr, rb.scan_scratch, st, rb.dummy_total)) return e;
                    const bool big_tiles = rb.cap >= (1 << 22);
                    const unsigned tiles = (unsigned)ceil_div((int64_t)rb.cap, (big_tiles ? 1024 : 256) * RAYTILE_IPT);
                    const size_t hist_lds = (size_t)rb.subtrees * 4;
                    const unsigned chunks = (unsigned)(rb.subtrees + rb.cap / RAYSUB_CHUNK);
                    if (big_tiles) IBVH_LAUNCH((rays_tilehist_kernel<1024>), dim3(tiles), dim3(1024), hist_lds, st, rb);
                    else IBVH_LAUNCH((rays_tilehist_kernel<256>), dim3(tiles), dim3(256), hist_lds, st, rb);
                    IBVH_LAUNCH((rays_binscan_kernel), dim3(1), dim3(1024), 0, st, rb);
                    if (big_tiles) IBVH_LAUNCH((rays_scatter_kernel<1024>), dim3(tiles), dim3(1024), hist_lds, st, rb);
                    else IBVH_LAUNCH((rays_scatter_kernel<256>), dim3(tiles), dim3(256), hist_lds, st, rb);
                    if (lds > 64 * 1024)
                        IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)rays_subtree_kernel<L, N, I, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    IBVH_LAUNCH((rays_subtree_kernel<L, N, I, false>), dim3(chunks), dim3(RAYSUB_TPB), lds, st, a, rb);
                    if (int e = scan_counts<I>((I *)rb.hits, (int64_t)rb.cap, nullptr, rb.scan_scratch, st, rb.dummy_total, nullptr, rb.n_items)) return e;
                    IBVH_LAUNCH((rays_counts_kernel<I>), dim3((unsigned)ceil_div(a.n_items, 256)), dim3(256), 0, st, rb, a.counts, a.n_items);
                    IBVH_LAUNCH((lvt_rays_kernel<L, N, I, false>), dim3(rblocks), dim3(64), 0, st, standby, none, ray_block);
                } else {
                    IBVH_LAUNCH((rays_place_kernel<I>), dim3((unsigned)ceil_div((int64_t)rb.region_cap, 1024), RAY_REGIONS), dim3(256), 0, st, rb, a.contacts,
                                a.guard_total, a.guard_capacity);
                    if (lds > 64 * 1024)
                        IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)rays_subtree_kernel<L, N, I, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    IBVH_LAUNCH((rays_subtree_kernel<L, N, I, true>), dim3((unsigned)(rb.subtrees + rb.cap / RAYSUB_CHUNK)), dim3(RAYSUB_TPB), lds, st, a, rb);
                    IBVH_LAUNCH((lvt_rays_kernel<L, N, I, true>), dim3(rblocks), dim3(64), 0, st, standby, none, ray_block);
                }
                IBVH_LAUNCH_CHECK();
                return IBVH_OK;
            }
        }
        if constexpr (std::is_same<typename L::elt, float>::value && std::is_same<N, BBox<float>>::value) {
            if (a.shadow != nullptr && !count_work) {
                // regular rays over the 8-wide shadow; the irregular ones (if any) by the binary walker behind it, without
                // a cache of its own (the block headers belong to the shadow walker)
                const ibvh_tree t{a.tree.levels, a.tree.real_leaves, 0, a.tree.virtual_leaves, 0};
                const RayShadow sh = make_ray_shadow(t);
                Args<L, N, I> irr = a;
                irr.rays_filter = 1;
                const PairCache<I> none{nullptr, 0};
                if (write) {
                    IBVH_LAUNCH((lvt_rays_wide_kernel<L, N, I, true>), dim3(rblocks), dim3(64), 0, st, a, cache, ray_block, sh);
                    IBVH_LAUNCH((lvt_rays_kernel<L, N, I, true>), dim3(rblocks), dim3(64), 0, st, irr, none, ray_block);
                } else {
                    IBVH_LAUNCH((lvt_rays_wide_kernel<L, N, I, false>), dim3(rblocks), dim3(64), 0, st, a, cache, ray_block, sh);
                    IBVH_LAUNCH((lvt_rays_kernel<L, N, I, false>), dim3(rblocks), dim3(64), 0, st, irr, none, ray_block);
                }
                IBVH_LAUNCH_CHECK();
                return IBVH_OK;
            }
        }
        if constexpr (kWorkTypes<L, N, I>) {
            if (count_work) {
                IBVH_LAUNCH((lvt_rays_kernel<L, N, I, false, true>), dim3(rblocks), dim3(64), 0, st, a, cache, ray_block);
                IBVH_LAUNCH_CHECK();
                return IBVH_OK;
            }
        }
        if (write) IBVH_LAUNCH((lvt_rays_kernel<L, N, I, true>), dim3(rblocks), dim3(64), 0, st, a, cache, ray_block);
        else IBVH_LAUNCH((lvt_rays_kernel<L, N, I, false>), dim3(rblocks), dim3(64), 0, st, a, cache, ray_block);
    } else {
        // BBox nodes with at least one node level below the start level: frontier descent + brute force;
        // everything else (BSphere nodes, start_level == levels): the exact joint walk
        if constexpr (N::kind == IBVH_BBOX) {
            if (a.start_level < a.tree.levels) {
                const bool force_wide = g_tuning.lvt_wide != 0; // test knob: 64-bit queue entries for every tree
                // queue entries pack (leaf-parent index << 6 | lane): 32 bits up to 28 levels (134 M leaves), 64 bits up to
                // 31 levels (wave-uniform arithmetic is 32-bit: positions + 2^(levels-1) must stay below 2^32); deeper
                // trees take the exact walk
                if (a.tree.levels <= 31) {
                    const bool wide = force_wide || a.tree.levels > 28;
                    // With BBox nodes the contact list does not depend on the start level (monotone box tests, see the
                    // header comment), so the descent always starts where one 64-lane step covers all roots (level 7,
                    // or the highest built level below it) whatever level the caller named.
                    Args<L, N, I> aq = a;
                    aq.xcd_tiles = a.xcd_tiles > 1 ? a.xcd_tiles * 4 / QUEUE_WAVES : a.xcd_tiles; // (runs are counted in 256-item units)
                    const unsigned qblocks = (unsigned)ceil_div(a.n_items, (int64_t)64 * QUEUE_WAVES);
                    int64_t top = a.built_level > 7 ? a.built_level : 7; // level 7: 64 nodes, one 64-lane step
                    if (top > a.tree.levels - 1) top = a.tree.levels - 1;
                    aq.start_level = top; // also when the caller named a HIGHER level: levels 1..6 hold < 64 nodes each
                    if (g_tuning.lvt_dual) {
                        // the dual descent (lvt_dual_kernel): pairs pack (T index within its level << 7 | Q node): 32 bits up to 27 levels
                        const bool dwide = force_wide || a.tree.levels > 27;
                        const DualPlan dplan = make_dual_plan((int)aq.tree.levels, (uint32_t)aq.tree.virtual_leaves, (int)aq.start_level, sizeof(N));
                        if constexpr (kWorkTypes<L, N, I>) {
                            if (count_work) {
                                if (dwide || aq.narrow != IBVH_NARROW_NONE) return IBVH_ERR_UNSUPPORTED;
                                IBVH_LAUNCH((lvt_dual_kernel<L, N, I, MODE, false, false, false, true>), dim3(qblocks), dim3(64), 0, st, aq, cache, dplan);
                                IBVH_LAUNCH_CHECK();
                                return IBVH_OK;
                            }
                        }
                        const int dvariant = (write ? 1 : 0) | (aq.narrow != IBVH_NARROW_NONE ? 2 : 0) | (dwide ? 4 : 0);
#define IBVH_DUAL_LAUNCH(W_, N_, D_) IBVH_LAUNCH((lvt_dual_kernel<L, N, I, MODE, W_, N_, D_>), dim3(qblocks), dim3(64), 0, st, aq, cache, dplan)
                        switch (dvariant) {
                        case 0: IBVH_DUAL_LAUNCH(false, false, false); break;
                        case 1: IBVH_DUAL_LAUNCH(true, false, false); break;
                        case 2: IBVH_DUAL_LAUNCH(false, true, false); break;
                        case 3: IBVH_DUAL_LAUNCH(true, true, false); break;
                        case 4: IBVH_DUAL_LAUNCH(false, false, true); break;
                        case 5: IBVH_DUAL_LAUNCH(true, false, true); break;
                        case 6: IBVH_DUAL_LAUNCH(false, true, true); break;
                        default: IBVH_DUAL_LAUNCH(true, true, true); break;
                        }
#undef IBVH_DUAL_LAUNCH
                        IBVH_LAUNCH_CHECK();
                        return IBVH_OK;
                    }
                    const int64_t c = aq.tree.levels - BRUTE_DEPTH;
                    const int cut = (int)(c > aq.start_level ? c : aq.start_level);
                    if constexpr (kWorkTypes<L, N, I>) {
                        if (count_work) {
                            if (wide || aq.narrow != IBVH_NARROW_NONE) return IBVH_ERR_UNSUPPORTED;
                            IBVH_LAUNCH((lvt_queue_kernel<L, N, I, MODE, false, false, false, true>), dim3(qblocks), dim3(64 * QUEUE_WAVES), 0, st,
                                        aq, cache, cut);
                            IBVH_LAUNCH_CHECK();
                            return IBVH_OK;
                        }
                    }
                    const int variant = (write ? 1 : 0) | (aq.narrow != IBVH_NARROW_NONE ? 2 : 0) | (wide ? 4 : 0);
#define IBVH_QUEUE_LAUNCH(W_, N_, D_)                                                                                 \
    IBVH_LAUNCH((lvt_queue_kernel<L, N, I, MODE, W_, N_, D_>), dim3(qblocks), dim3(64 * QUEUE_WAVES), 0, st, aq, cache, cut)
                    switch (variant) {
                    case 0: IBVH_QUEUE_LAUNCH(false, false, false); break;
                    case 1: IBVH_QUEUE_LAUNCH(true, false, false); break;
                    case 2: IBVH_QUEUE_LAUNCH(false, true, false); break;
                    case 3: IBVH_QUEUE_LAUNCH(true, true, false); break;
                    case 4: IBVH_QUEUE_LAUNCH(false, false, true); break;
                    case 5: IBVH_QUEUE_LAUNCH(true, false, true); break;
                    case 6: IBVH_QUEUE_LAUNCH(false, true, true); break;
                    default: IBVH_QUEUE_LAUNCH(true, true, true); break;
                    }
#undef IBVH_QUEUE_LAUNCH
                    IBVH_LAUNCH_CHECK();
                    return IBVH_OK;
                }
            }
        }
        if (count_work) return IBVH_ERR_UNSUPPORTED; // (BSphere nodes / start at the leaf level: the exact walk has no counters)
        // (the exact walk keeps the run-time narrow switch: NARROW = true covers both)
        if (write) IBVH_LAUNCH((lvt_joint_kernel<L, N, I, MODE, true, true>), dim3(blocks), dim3(256), 0, st, a, cache);
        else IBVH_LAUNCH((lvt_joint_kernel<L, N, I, MODE, false, true>), dim3(blocks), dim3(256), 0, st, a, cache);
    }
    IBVH_LAUNCH_CHECK();
    return IBVH_OK;
}

// Bytes of the quantised shadow a ray traversal of `bvh` with `num_rays` rays uses, 0 when the binary walk serves it:
// single-precision leaves under BBox{Float32} nodes, a fully built tree of 8 .. 26 levels, and enough rays for the one
// streaming pass over the nodes that builds the shadow to pay (at least one ray per 64 leaves).
inline size_t rays_shadow_bytes(const ibvh_bvh &bvh, int64_t num_rays) {
    if (bvh.types.node_kind != IBVH_BBOX || bvh.types.node_float != IBVH_F32 || bvh.types.leaf_float != IBVH_F32) return 0;
    if (!g_tuning.rays_shadow) return 0; // (development knob, off: see the note at the kernel)
    if (bvh.built_level > 1 || num_rays * 64 < bvh.tree.real_leaves) return 0;
    const RayShadow sh = make_ray_shadow(bvh.tree);
    return sh.depths ? (size_t)sh.base[sh.depths] * SHADOW_ENTRY_BYTES : 0;
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
    const size_t tail_bytes = bin_plan.depth ? bin_plan.bytes : shadow_bytes;
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
                if constexpr (MODE == MODE_RAYS && std::is_same<N, BBox<float>>::value && std::is_same<typename L::elt, float>::value) {
                    if (shadow_ptr && !write) { // (the writing pass of a _count / _write pair finds the shadow where the count left it)
                        const RayShadow sh = make_ray_shadow(walk->tree);
                        IBVH_LAUNCH((ray_shadow_build_kernel<N>), dim3((unsigned)ceil_div((int64_t)sh.base[sh.depths], 256)), dim3(256), 0, st,
                                    a.nodes, a.tree, sh, (ShadowEntry *)shadow_ptr);
                    }
                }
                PairCache<I> cache{K ? (IndexPair<I> *)((char *)scratch + scan_scratch_bytes(n_items)) : nullptr, K};
                if (int e = launch<L, N, I, MODE>(a, cache, write, st, bins)) return e;
                if (write || work) return (int)IBVH_OK;
                if (int e = scan_counts<I>((I *)counts, n_items, enqueue ? nullptr : total_out, scratch, st, enqueue ? total_dev : nullptr,
                                           enqueue ? total_host : nullptr)) return e;
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
    // the BVH with more leaves supplies the work items; flip restores (bvh1, bvh2) order (:15-36)
    const bool flip = !(bvh1->tree.real_leaves >= bvh2->tree.real_leaves);
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

#ifdef IBVH_PHASE_STAMPS
extern "C" int ibvh_debug_lvt_ticks(unsigned long long *out /* 8 */, int reset) {
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(ibvh::lvt::g_lvt_ticks), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ibvh::lvt::g_lvt_ticks), sizeof(unsigned long long) * 8);
}
#endif
