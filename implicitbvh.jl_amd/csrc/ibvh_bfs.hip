// ibvh_bfs.hip — breadth-first traversal (BFSTraversal) on gfx950: level-synchronous expansion of a
// bounding-volume-test-tree held in two ping-pong pair queues; the "dynamic contact-pair work queue".
//
// Replaces src/traverse/breadth_first/traverse_single{,_gpu}.jl, traverse_pair{,_gpu}.jl and
// src/raytrace/breadth_first/{breadth_first,raytrace_gpu}.jl.
//
// The reference's KernelAbstractions kernels do one LDS atomic per thread to reserve slots in a
// workgroup staging buffer and one global atomic per workgroup (bfs/traverse_single_gpu.jl:65-99).
// Here slot reservation inside a wave is atomic-free: every lane produces 0..4 pairs, the count is
// split into bit planes and wave64 ballots + popcounts give each lane its exclusive offset; waves
// combine through LDS, the workgroup takes ONE global atomic on the level's queue tail, stages its
// pairs in LDS and writes them out as one contiguous, coalesced run.  Output order across workgroups
// is arbitrary — as in the reference's GPU path, whose tests compare after sorting
// (test/gputests.jl:71-78).  One kernel template serves all ten reference kernels through a policy.
#include <type_traits>

#include "ibvh_common.hpp"

namespace ibvh {
namespace bfs {

constexpr int TPB = 256;
constexpr int GEN_SLOTS = 16; // words per step the workgroups spread their check counts over (level_kernel)

// Workgroup barrier that orders LDS accesses only: __syncthreads() also drains every outstanding global store and
// load (s_waitcnt vmcnt(0)), which would make a workgroup wait for its queue writes before it may read its next chunk.
IBVH_D void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <class L, class N> struct TreeRef {
    const char *leaves;
    LeafLayout lay;
    const N *nodes;
    int64_t levels, virtual_leaves;
    int64_t level;      // level of this tree's entries in the source queue
    int64_t skips;      // level_skips(level)
    int64_t leaf_first; // 2^(levels-1)
    int64_t child_real; // real nodes on level+1
    IBVH_D N node(int64_t implicit) const { return load_vol<N>(nodes + (implicit - skips - 1)); }
    IBVH_D const char *leaf_rec(int64_t implicit) const { return leaves + (implicit - leaf_first) * lay.stride; }
    // unsafe_isvirtual(tree, 2*implicit + 1): is the right child virtual?
    IBVH_D bool right_child_virtual(int64_t implicit) const {
        int64_t c = 2 * implicit + 1;
        return c - (int64_t(1) << level) >= child_real;
    }
};

template <class L, class N> TreeRef<L, N> make_ref(const ibvh_bvh &b, const LeafLayout &lay, int64_t level) {
    TreeRef<L, N> r;
    r.leaves = (const char *)b.leaves;
    r.lay = lay;
    r.nodes = (const N *)b.nodes;
    r.levels = b.tree.levels;
    r.virtual_leaves = b.tree.virtual_leaves;
    r.level = level;
    r.skips = level_skips(r.levels, r.virtual_leaves, level);
    r.leaf_first = int64_t(1) << (r.levels - 1);
    r.child_real = level < r.levels ? level_num_real(r.levels, r.virtual_leaves, level + 1) : 0;
    return r;
}

// `narrow` arguments carry the menu code and the IBVH_OUTPUT_POSITIONS flag (include/ibvh.h)
inline bool narrow_arg_ok(int32_t narrow, bool rays) {
    if (narrow & ~(IBVH_NARROW_MASK | IBVH_OUTPUT_POSITIONS)) return false;
    const int code = narrow & IBVH_NARROW_MASK;
    return rays ? (code == IBVH_NARROW_NONE || code == IBVH_NARROW_RAY_ORIGIN_OUTSIDE)
                : (code == IBVH_NARROW_NONE || code == IBVH_NARROW_MORTON_LT || code == IBVH_NARROW_INDEX_LT);
}
template <class T> IBVH_D bool origin_outside(const BSphere<T> &s, const T *p) { return dist3sq(p, s.x) > s.r * s.r; }
template <class T> IBVH_D bool origin_outside(const BBox<T> &b, const T *p) {
    return (p[0] < b.lo[0]) | (p[0] > b.up[0]) | (p[1] < b.lo[1]) | (p[1] > b.up[1]) | (p[2] < b.lo[2]) | (p[2] > b.up[2]);
}

IBVH_D bool narrow_eval(int narrow, uint64_t ma, int64_t ia, uint64_t mb, int64_t ib) {
    if (narrow == IBVH_NARROW_MORTON_LT) return ma < mb;
    if (narrow == IBVH_NARROW_INDEX_LT) return ia < ib;
    return true;
}

// ------------------------------------------------------------------------------------------
// policies.  The reference's kernels CHECK a pair of the level's queue and, if it passes, append its child pairs for
// the next level to check (bfs/traverse_single_gpu.jl:30-120): every generated pair is written to the queue, read
// back one launch later and costs two unrelated volume fetches there.  Here a step is shifted by half a level: the
// queues hold pairs that have already PASSED their check; a step enumerates the children of such a pair (no memory
// access) and checks them at once — the four children of (a, b) need the volumes 2a, 2a+1, 2b, 2b+1: two adjacent
// pairs, fetched once — and only the passing ones reach the queue (about a third).  The same pairs are checked as in
// the reference, one launch earlier: num_checks (every generated pair, bfs/traverse_single.jl:25,48) and the
// contact set are unchanged; queue traffic and volume fetches drop by 3 - 4 x.
// A policy = check(pair, result) (does the pair pass at its level? `result`: what goes into the queue — the pair
// itself, or the contact at leaf level) + children(pair, out) (the next level's pairs of a passing pair).
// ------------------------------------------------------------------------------------------
// Register storage for a node volume OR a leaf volume, whichever the launch's (uniform) level flag says: what a lane
// prefetches for a child stays as small as the larger of the two instead of holding both.
template <class A, class B> struct Either {
    static constexpr int W = (int)((sizeof(A) > sizeof(B) ? sizeof(A) : sizeof(B)) / 4);
    uint32_t w[W];
    template <class X> IBVH_D void set(const X &x) { __builtin_memcpy(w, &x, sizeof(X)); }
    template <class X> IBVH_D X get() const {
        X x;
        __builtin_memcpy(&x, w, sizeof(X));
        return x;
    }
};

template <class I> struct Identity { // the initial queue: "children" of nothing, still to be checked
    static constexpr int MAXOUT = 1;
    static constexpr bool kIdentity = true;
    IBVH_D int children(IndexPair<I> s, IndexPair<I> *out) const {
        out[0] = s;
        return 1;
    }
};

// _traverse_nodes_gpu! — bfs/traverse_single_gpu.jl:30-120 (same rules as traverse_single_cpu.jl:64-133) — and
// _traverse_leaves_gpu! — :153-211
template <class L, class N, class I> struct SelfStep {
    using leaf_t = L;
    using node_t = N;
    static constexpr int MAXOUT = 4;
    TreeRef<L, N> t;
    int leaf;        // entries are leaf pairs
    int self_checks; // (nodes) :44
    int narrow;      // (leaves) menu code
    int positions;   // (leaves) IBVH_OUTPUT_POSITIONS: 1-based leaf positions instead of user indices, left leaf first
    // What one check needs from memory, fetched UNCONDITIONALLY (`s` is always a valid pair of this level: the kernel
    // hands over a real child or safe()): every lane's loads for all its <= 4 children are in flight before the first
    // test — four dependent round trips per lane otherwise, and the kernel is bound by exactly that latency.
    struct Loaded {
        Either<N, L> a, b;
        I ia, ib;
    };
    IBVH_D IndexPair<I> safe() const { const I f = I(int64_t(1) << (t.level - 1)); return {f, f}; } // the level's first node: always real
    IBVH_D Loaded load(IndexPair<I> s) const {
        Loaded x{};
        if (leaf) {
            const char *r1 = t.leaf_rec(s.a), *r2 = t.leaf_rec(s.b);
            x.a.set(load_vol<L>(r1));
            x.b.set(load_vol<L>(r2));
            x.ia = load_index<I>(r1, t.lay);
            x.ib = load_index<I>(r2, t.lay);
        } else {
            x.a.set(t.node(s.a));
            x.b.set(t.node(s.b));
        }
        return x;
    }
    IBVH_D bool test(IndexPair<I> s, const Loaded &x, IndexPair<I> &res) const {
        if (leaf) {
            if (!iscontact(x.a.template get<L>(), x.b.template get<L>())) return false;
            if (narrow != IBVH_NARROW_NONE) { // (the Morton codes are fetched only here: the menu predicates are the rare case)
                const uint64_t m1 = narrow == IBVH_NARROW_MORTON_LT ? load_morton(t.leaf_rec(s.a), t.lay) : 0;
                const uint64_t m2 = narrow == IBVH_NARROW_MORTON_LT ? load_morton(t.leaf_rec(s.b), t.lay) : 0;
                if (!narrow_eval(narrow, m1, x.ia, m2, x.ib)) return false;
            }
            if (positions) res = {I(s.a - t.leaf_first + 1), I(s.b - t.leaf_first + 1)}; // (a is left of b)
            else res = x.ia > x.ib ? IndexPair<I>{x.ib, x.ia} : IndexPair<I>{x.ia, x.ib};
            return true;
        }
        res = s;
        return s.a == s.b || iscontact(x.a.template get<N>(), x.b.template get<N>()); // (a node against itself is not tested, :52-70)
    }
    // The children of a source pair (a, b) are combinations of {2a, 2a+1} x {2b, 2b+1}: FOUR volumes (two adjacent
    // pairs in memory) serve all of them — fetched once per source pair instead of twice per child pair.
    // (called on the CHILD level's policy with the parent-level source pair)
    static constexpr bool kIdentity = false;
    struct Sides {
        Either<N, L> v[2][2]; // [side a / b][left / right child]
        I idx[2][2];
    };
    IBVH_D Sides load_sides(IndexPair<I> s, const SelfStep &) const {
        Sides x{};
        const int64_t first = int64_t(1) << (t.level - 1), nreal = level_num_real(t.levels, t.virtual_leaves, t.level);
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int64_t c0 = 2 * (int64_t)(side ? s.b : s.a);
            const int64_t c1 = (c0 + 1 - first) < nreal ? c0 + 1 : c0; // a virtual right child re-reads the left one (never selected)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int64_t c = k ? c1 : c0;
                if (leaf) {
                    const char *r = t.leaf_rec(c);
                    x.v[side][k].set(load_vol<L>(r));
                    x.idx[side][k] = load_index<I>(r, t.lay);
                } else {
                    x.v[side][k].set(t.node(c));
                }
            }
        }
        return x;
    }
    IBVH_D bool test_sides(IndexPair<I> s, const Sides &x, IndexPair<I> &res) const {
        const int ka = (int)(s.a & 1), kb = (int)(s.b & 1);
        Loaded y;
        y.a = ka ? x.v[0][1] : x.v[0][0];
        y.b = kb ? x.v[1][1] : x.v[1][0];
        y.ia = ka ? x.idx[0][1] : x.idx[0][0];
        y.ib = kb ? x.idx[1][1] : x.idx[1][0];
        return test(s, y, res);
    }
    IBVH_D int children(IndexPair<I> s, IndexPair<I> *out) const {
        const I a = s.a, b = s.b;
        if (a == b) {
            if (t.right_child_virtual(a)) {
                if (!self_checks) return 0;
                out[0] = {I(2 * a), I(2 * a)};
                return 1;
            }
            if (self_checks) {
                out[0] = {I(2 * a), I(2 * a)};
                out[1] = {I(2 * a), I(2 * a + 1)};
                out[2] = {I(2 * a + 1), I(2 * a + 1)};
                return 3;
            }
            out[0] = {I(2 * a), I(2 * a + 1)};
            return 1;
        }
        // node a is left of node b, so its children are real (traverse_single_cpu.jl:103-105)
        out[0] = {I(2 * a), I(2 * b)};
        if (t.right_child_virtual(b)) {
            out[1] = {I(2 * a + 1), I(2 * b)};
            return 2;
        }
        out[1] = {I(2 * a), I(2 * b + 1)};
        out[2] = {I(2 * a + 1), I(2 * b)};
        out[3] = {I(2 * a + 1), I(2 * b + 1)};
        return 4;
    }
};

// The six pair kernels of bfs/traverse_pair_gpu.jl:34-609 in one policy: which side is at leaf level
// (leaf1 / leaf2) and which side descends (d1 / d2).
template <class L, class N, class I> struct PairStep {
    using leaf_t = L;
    using node_t = N;
    static constexpr int MAXOUT = 4;
    TreeRef<L, N> t1, t2;
    int narrow, positions;
    int leaf1, leaf2, d1, d2;
    struct Loaded { // (see SelfStep::Loaded)
        Either<N, L> a, b;
        I ia, ib;
    };
    IBVH_D IndexPair<I> safe() const { return {I(int64_t(1) << (t1.level - 1)), I(int64_t(1) << (t2.level - 1))}; }
    IBVH_D Loaded load(IndexPair<I> s) const {
        Loaded x{};
        const I a = s.a, b = s.b;
        if (leaf1) {
            const char *r1 = t1.leaf_rec(a);
            x.a.set(load_vol<L>(r1));
            if (leaf2) x.ia = load_index<I>(r1, t1.lay);
        } else {
            x.a.set(t1.node(a));
        }
        if (leaf2) {
            const char *r2 = t2.leaf_rec(b);
            x.b.set(load_vol<L>(r2));
            if (leaf1) x.ib = load_index<I>(r2, t2.lay);
        } else {
            x.b.set(t2.node(b));
        }
        return x;
    }
    IBVH_D bool test(IndexPair<I> s, const Loaded &x, IndexPair<I> &res) const {
        const I a = s.a, b = s.b;
        res = s;
        if (leaf1 && leaf2) {
            // _traverse_leaves_pair_gpu! (:556-609): (leaf1.index, leaf2.index), not re-ordered
            if (!iscontact(x.a.template get<L>(), x.b.template get<L>())) return false;
            if (narrow != IBVH_NARROW_NONE) {
                const uint64_t m1 = narrow == IBVH_NARROW_MORTON_LT ? load_morton(t1.leaf_rec(a), t1.lay) : 0;
                const uint64_t m2 = narrow == IBVH_NARROW_MORTON_LT ? load_morton(t2.leaf_rec(b), t2.lay) : 0;
                if (!narrow_eval(narrow, m1, x.ia, m2, x.ib)) return false;
            }
            res = positions ? IndexPair<I>{I(a - t1.leaf_first + 1), I(b - t2.leaf_first + 1)} : IndexPair<I>{x.ia, x.ib};
            return true;
        }
        if (leaf1) return iscontact(x.a.template get<L>(), x.b.template get<N>()); // iscontact(leaf1.volume, node2), :461-527
        if (leaf2) return iscontact(x.a.template get<N>(), x.b.template get<L>()); // :360-426
        return iscontact(x.a.template get<N>(), x.b.template get<N>());
    }
    static constexpr bool kIdentity = false;
    struct Sides { // (see SelfStep::Sides; a side that did not descend in the parent step holds its one node twice)
        Either<N, L> v[2][2];
        I idx[2][2];
    };
    IBVH_D Sides load_sides(IndexPair<I> s, const PairStep &parent) const {
        Sides x{};
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const TreeRef<L, N> &t = side ? t2 : t1;
            const bool is_leaf = side ? leaf2 : leaf1, descended = side ? parent.d2 : parent.d1;
            const int64_t p = (int64_t)(side ? s.b : s.a);
            const int64_t first = int64_t(1) << (t.level - 1), nreal = level_num_real(t.levels, t.virtual_leaves, t.level);
            const int64_t c0 = descended ? 2 * p : p;
            const int64_t c1 = (descended && (c0 + 1 - first) < nreal) ? c0 + 1 : c0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int64_t c = k ? c1 : c0;
                if (is_leaf) {
                    const char *r = t.leaf_rec(c);
                    x.v[side][k].set(load_vol<L>(r));
                    x.idx[side][k] = load_index<I>(r, t.lay);
                } else {
                    x.v[side][k].set(t.node(c));
                }
            }
        }
        return x;
    }
    IBVH_D bool test_sides(IndexPair<I> s, const Sides &x, IndexPair<I> &res, const PairStep &parent) const {
        const int ka = parent.d1 ? (int)(s.a & 1) : 0, kb = parent.d2 ? (int)(s.b & 1) : 0;
        Loaded y;
        y.a = ka ? x.v[0][1] : x.v[0][0];
        y.b = kb ? x.v[1][1] : x.v[1][0];
        y.ia = ka ? x.idx[0][1] : x.idx[0][0];
        y.ib = kb ? x.idx[1][1] : x.idx[1][0];
        return test(s, y, res);
    }
    IBVH_D int children(IndexPair<I> s, IndexPair<I> *out) const {
        const I a = s.a, b = s.b;
        if (d1 && d2) { // _traverse_nodes_pair_gpu! (:34-119)
            const bool v1 = t1.right_child_virtual(a), v2 = t2.right_child_virtual(b);
            out[0] = {I(2 * a), I(2 * b)};
            if (v1) {
                if (v2) return 1;
                out[1] = {I(2 * a), I(2 * b + 1)};
                return 2;
            }
            if (v2) {
                out[1] = {I(2 * a + 1), I(2 * b)};
                return 2;
            }
            out[1] = {I(2 * a), I(2 * b + 1)};
            out[2] = {I(2 * a + 1), I(2 * b)};
            out[3] = {I(2 * a + 1), I(2 * b + 1)};
            return 4;
        }
        if (d1) { // _left_ kernels (:155-222, :360-426)
            out[0] = {I(2 * a), b};
            if (t1.right_child_virtual(a)) return 1;
            out[1] = {I(2 * a + 1), b};
            return 2;
        }
        // _right_ kernels (:258-325, :461-527)
        out[0] = {a, I(2 * b)};
        if (t2.right_child_virtual(b)) return 1;
        out[1] = {a, I(2 * b + 1)};
        return 2;
    }
};

// _traverse_rays_nodes_gpu! / _traverse_rays_leaves_gpu! — raytrace/breadth_first/raytrace_gpu.jl:26-179
template <class L, class N, class I> struct RayStep {
    using leaf_t = L;
    using node_t = N;
    static constexpr int MAXOUT = 2;
    using T = typename L::elt;
    TreeRef<L, N> t;
    const T *points, *dirs;
    int leaf;
    int narrow, positions;
    struct Loaded { // (see SelfStep::Loaded)
        T p[3], d[3];
        Either<N, L> v;
        I idx;
    };
    IBVH_D IndexPair<I> safe() const { return {I(int64_t(1) << (t.level - 1)), I(1)}; } // (num_rays >= 1 whenever a level runs)
    IBVH_D Loaded load(IndexPair<I> s) const {
        Loaded x{};
        const int64_t r0 = 3 * ((int64_t)s.b - 1);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            x.p[k] = points[r0 + k];
            x.d[k] = dirs[r0 + k];
        }
        if (leaf) {
            const char *r = t.leaf_rec(s.a);
            x.v.set(load_vol<L>(r));
            x.idx = load_index<I>(r, t.lay);
        } else {
            x.v.set(t.node(s.a));
        }
        return x;
    }
    IBVH_D bool test(IndexPair<I> s, const Loaded &x, IndexPair<I> &res) const {
        const I a = s.a, iray = s.b;
        res = s;
        if (leaf) {
            const L lv = x.v.template get<L>();
            if (!isintersection(lv, x.p, x.d)) return false;
            if (narrow == IBVH_NARROW_RAY_ORIGIN_OUTSIDE && !origin_outside(lv, x.p)) return false; // raytrace_gpu.jl:159
            res = {positions ? I(a - t.leaf_first + 1) : x.idx, iray};
            return true;
        }
        return isintersection(x.v.template get<N>(), x.p, x.d);
    }
    static constexpr bool kIdentity = false;
    struct Sides { // the ray once + the two children of its node
        T p[3], d[3];
        Either<N, L> v[2];
        I idx[2];
    };
    IBVH_D Sides load_sides(IndexPair<I> s, const RayStep &) const {
        Sides x{};
        const int64_t r0 = 3 * ((int64_t)s.b - 1);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            x.p[k] = points[r0 + k];
            x.d[k] = dirs[r0 + k];
        }
        const int64_t first = int64_t(1) << (t.level - 1), nreal = level_num_real(t.levels, t.virtual_leaves, t.level);
        const int64_t c0 = 2 * (int64_t)s.a, c1 = (c0 + 1 - first) < nreal ? c0 + 1 : c0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int64_t c = k ? c1 : c0;
            if (leaf) {
                const char *r = t.leaf_rec(c);
                x.v[k].set(load_vol<L>(r));
                x.idx[k] = load_index<I>(r, t.lay);
            } else {
                x.v[k].set(t.node(c));
            }
        }
        return x;
    }
    IBVH_D bool test_sides(IndexPair<I> s, const Sides &x, IndexPair<I> &res) const {
        const int k = (int)(s.a & 1);
        Loaded y;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            y.p[j] = x.p[j];
            y.d[j] = x.d[j];
        }
        y.v = k ? x.v[1] : x.v[0];
        y.idx = k ? x.idx[1] : x.idx[0];
        return test(s, y, res);
    }
    IBVH_D int children(IndexPair<I> s, IndexPair<I> *out) const {
        out[0] = {I(2 * s.a), s.b};
        if (t.right_child_virtual(s.a)) return 1;
        out[1] = {I(2 * s.a + 1), s.b};
        return 2;
    }
};

// ------------------------------------------------------------------------------------------
// the level kernel: children of the source pairs, checked at once; wave64 ballot compaction of the passing ones +
// one global atomic per workgroup
// ------------------------------------------------------------------------------------------
// counters[0]          overflow flag: 0, or 1 + the index of the first step whose destination queue was too small
// counters[1 + s]      entries in the SOURCE queue of step s (counters[1] = the initial queue; step s produces
//                      counters[2 + s], counting on past `capacity` so that the exact need is known)
// counters[chk * (1 + k) + s]   pairs step s generated and checked, slot k < GEN_SLOTS (num_checks is their sum; chk =
//                      total_levels + 8 = the length of one block).  Same-address global atomics are serialised at
//                      ~11 ns each on this part (measured: a level's time was 11 ns x (workgroups + waves) whatever
//                      its size — 108 us for 0.5 M source pairs, 140 us for 5.2 M), so a workgroup adds its count
//                      ONCE, to the slot blockIdx % GEN_SLOTS (different cache lines), not once per wave to one word
// No host read between the levels: a step takes its source count from the device word the previous step
// accumulated, so the grid is a fixed number of workgroups that stride over the source queue; once a step has
// overflowed, the later steps (already enqueued) return at once and the caller resumes from that step with larger
// queues (the source queue of the overflowed step is intact).
template <class I, class PolA, class PolB>
__global__ __launch_bounds__(TPB) void level_kernel(const IndexPair<I> *__restrict__ src, IndexPair<I> *__restrict__ dst,
                                                    int64_t capacity, unsigned long long *__restrict__ counters, int step, int chk,
                                                    PolA pa, PolB pb) {
    constexpr int MAXOUT = PolA::MAXOUT;
    // Results of MANY chunks are collected in LDS and leave in one coalesced run behind ONE global atomic on the level's
    // tail (round 2 took that atomic per chunk: every chunk of a workgroup then waited out a ~2 us round trip to L2 —
    // ~34 chunks a workgroup and level at 1e6 leaves, most of the 1.4 ms of config 2).  A wave reserves its share of the
    // staging area with one LDS atomic (no workgroup barrier for the prefix); one barrier per chunk tells everybody
    // whether the area could overflow with the next chunk and must be flushed first.
    constexpr int STAGE = (sizeof(I) == 4 ? 24 : 48) * 1024 / (int)sizeof(IndexPair<I>); // 3,072 pairs: six workgroups a CU (the registers allow five)
    static_assert(STAGE >= 2 * MAXOUT * TPB, "the staging area holds at least two worst-case chunks");
    __shared__ IndexPair<I> staged[STAGE];
    __shared__ int s_fill;
    __shared__ unsigned long long s_base, s_gen;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) {
        s_fill = 0;
        s_gen = 0ull;
    }
    lds_barrier();
    // an EARLIER step overflowed: its destination (our source) is incomplete.  The step that overflows itself keeps going
    // in every workgroup — also those that start, or reach their next chunk, after a sibling raised the flag — so that
    // counters[2 + step] ends as the exact number of pairs the step produces (required_capacity).
    const unsigned long long flag = counters[0];
    if (flag != 0ull && flag - 1ull < (unsigned long long)step) return;
    const int64_t num_src = (int64_t)counters[1 + step];
    if (num_src > capacity) return;   // (defensive: the flag covers this)
    const uint64_t lt = ((uint64_t)1 << lane) - 1;
    unsigned long long generated = 0;
    // staging area -> destination queue: one global atomic for everything collected since the last flush
    auto flush = [&]() {
        const int total = s_fill; // (all waves are behind a barrier: stable)
        if (total > 0) {
            if (threadIdx.x == 0) s_base = atomicAdd(&counters[2 + step], (unsigned long long)total);
            lds_barrier();
            const unsigned long long base = s_base;
            if (base + (unsigned long long)total > (unsigned long long)capacity) {
                if (threadIdx.x == 0) atomicMax(&counters[0], (unsigned long long)(1 + step)); // the tail keeps counting
            } else {
                for (int p = threadIdx.x; p < total; p += TPB) dst[base + p] = staged[p];
            }
        }
        lds_barrier();
        if (threadIdx.x == 0) s_fill = 0;
        lds_barrier(); // staged / s_fill / s_base are reused (LDS only: the stores above stay in flight)
    };
    // Software pipeline over the workgroup's chunks (round 3): the source pair of chunk c + 2 and the volumes of chunk
    // c + 1 are in flight while chunk c is tested, staged and flushed.  A chunk used to be three DEPENDENT round trips
    // (source pair -> volumes -> tail atomic of a flush), with nothing else for the workgroup's waves to do meanwhile: the
    // level kernels were ~82 % waiting.  Lanes beyond the queue's end carry safe() (a valid pair of the level), so every
    // load is unconditional; vmcnt counts in order, so the wait for chunk c + 1's volumes sits at the register rotation at
    // the end of the body, a whole test-and-stage phase after they were requested.
    using Fetched = typename std::conditional<PolA::kIdentity, typename PolB::Loaded, typename PolB::Sides>::type;
    const int64_t stride = (int64_t)gridDim.x;
    auto source_of = [&](int64_t chunk) -> IndexPair<I> {
        const int64_t i = chunk * TPB + threadIdx.x;
        if constexpr (PolA::kIdentity) return i < num_src ? src[i] : pb.safe();
        else return i < num_src ? src[i] : pa.safe();
    };
    auto fetch = [&](IndexPair<I> sp) -> Fetched {
        if constexpr (PolA::kIdentity) return pb.load(sp);
        else return pb.load_sides(sp, pa);
    };
    IndexPair<I> s_cur = source_of((int64_t)blockIdx.x), s_nxt = source_of((int64_t)blockIdx.x + stride);
    Fetched f_cur = fetch(s_cur);
    for (int64_t chunk = blockIdx.x; chunk * TPB < num_src; chunk += stride) {
        const int64_t i = chunk * TPB + threadIdx.x;
        const Fetched f_nxt = fetch(s_nxt);
        const IndexPair<I> s_nn = source_of(chunk + 2 * stride);
        // (results stay in registers, statically indexed: `out[k++]` would put the array into scratch memory)
        IndexPair<I> kids[MAXOUT], res[MAXOUT];
        uint32_t okm = 0; // bit j: child j passed
        int nk = 0;
        if (i < num_src) nk = pa.children(s_cur, kids);
        generated += (unsigned long long)nk;
        if constexpr (PolA::kIdentity) {
            if (nk > 0 && pb.test(s_cur, f_cur, res[0])) okm = 1u;
        } else {
            // the source pair's children share their volumes: {2a, 2a+1} x {2b, 2b+1} (or one side kept): four volume
            // fetches per lane serve up to four checks
#pragma unroll
            for (int j = 0; j < MAXOUT; ++j) {
                bool ok;
                if constexpr (std::is_same<PolB, PairStep<typename PolB::leaf_t, typename PolB::node_t, I>>::value) ok = j < nk && pb.test_sides(kids[j], f_cur, res[j], pa);
                else ok = j < nk && pb.test_sides(kids[j], f_cur, res[j]);
                okm |= ok ? (1u << j) : 0u;
            }
        }
        const int k = __popc(okm);

        // exclusive offset inside the wave from ballots over the bit planes of k (k <= 4)
        uint64_t b0 = __ballot(k & 1), b1 = __ballot(k & 2), b2 = __ballot(k & 4);
        int wave_prefix = __popcll(b0 & lt) + 2 * __popcll(b1 & lt) + 4 * __popcll(b2 & lt);
        int wave_sum = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
        int wave_base = 0;
        if (lane == 0 && wave_sum > 0) wave_base = atomicAdd(&s_fill, wave_sum); // (LDS atomic: the wave's share of the staging area)
        wave_base = __builtin_amdgcn_readfirstlane(wave_base);
        const int at = wave_base + wave_prefix;
#pragma unroll
        for (int j = 0; j < MAXOUT; ++j)
            if ((okm >> j) & 1u) staged[at + __popc(okm & ((1u << j) - 1u))] = res[j];
        lds_barrier(); // every wave's share is reserved and written
        if (s_fill > STAGE - MAXOUT * TPB) flush(); // (uniform: read after the barrier) the next chunk might not fit
        s_cur = s_nxt;
        f_cur = f_nxt;
        s_nxt = s_nn;
    }
    flush();
    // pairs checked by this workgroup
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) generated += __shfl_xor(generated, o, 64);
    if (lane == 0 && generated != 0ull) atomicAdd(&s_gen, generated); // (LDS)
    lds_barrier();
    if (threadIdx.x == 0 && s_gen != 0ull) atomicAdd(&counters[chk * (1 + (int)(blockIdx.x % GEN_SLOTS)) + step], s_gen);
}

// ------------------------------------------------------------------------------------------
// initial queues — fill_initial_bvtt_* (bfs/traverse_single.jl:102-167, traverse_pair.jl:195-219,
// raytrace/breadth_first/breadth_first.jl:116-137), in the CPU branch's order
// ------------------------------------------------------------------------------------------
// rows i = 0..n-1; row i holds (n - i) entries when the self pair is included, (n - 1 - i) otherwise
IBVH_D int64_t tri_before(int64_t i, int64_t n, bool with_self) {
    return with_self ? i * n - i * (i - 1) / 2 : i * (2 * n - i - 1) / 2;
}
template <class I>
__global__ __launch_bounds__(TPB) void fill_self_kernel(IndexPair<I> *q, int64_t n, int64_t first, int with_self, int64_t total) {
    const int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (k >= total) return;
    // exact unranking: double-precision guess of the row, then integer correction
    const double t = with_self ? 2.0 * (double)n + 1.0 : 2.0 * (double)n - 1.0;
    double disc = t * t - 8.0 * (double)k;
    int64_t i = (int64_t)((t - sqrt(disc < 0.0 ? 0.0 : disc)) * 0.5);
    const int64_t imax = with_self ? n - 1 : n - 2;
    i = i < 0 ? 0 : (i > imax ? imax : i);
    while (i > 0 && tri_before(i, n, with_self) > k) --i;
    while (i < imax && tri_before(i + 1, n, with_self) <= k) ++i;
    const int64_t off = k - tri_before(i, n, with_self);
    const int64_t j = with_self ? i + off : i + 1 + off;
    q[k] = {I(first + i), I(first + j)};
}
template <class I>
__global__ __launch_bounds__(TPB) void fill_product_kernel(IndexPair<I> *q, int64_t rows, int64_t cols, int64_t first_row,
                                                          int64_t first_col) {
    const int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (k >= rows * cols) return;
    q[k] = {I(first_row + k / cols), I(first_col + k % cols)};
}

// ------------------------------------------------------------------------------------------
// host drivers
// ------------------------------------------------------------------------------------------
// A traversal is a fixed sequence of steps (one level kernel each).  All steps are enqueued back to back; ONE host
// read at the end fetches the overflow flag and every step's count.
struct Run {
    hipStream_t st;
    unsigned long long *counters;
    int64_t capacity;
    int64_t total_levels;
    int chk = 0;           // index of the first per-step check counter
    int step = 0;          // index of the next step
    int first = 0;         // steps below `first` were completed by an earlier call (resume): they are not launched
    bool swapped = false;  // false: step `step` reads q[0]
    void *q[2];
    void *src() const { return q[swapped ? 1 : 0]; }
    void *dst() const { return q[swapped ? 0 : 1]; }
};

inline int level_grid() {
    static const int cus = [] {
        int dev = 0, c = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        return c;
    }();
    // (default 4 a CU: one resident round at the self / pair kernels' occupancy; every workgroup costs a serialised tail
    // atomic per flush)
    const int per = g_tuning.bfs_wg_per_cu < 1 ? 1 : (g_tuning.bfs_wg_per_cu > 16 ? 16 : g_tuning.bfs_wg_per_cu);
    return cus * per;
}

template <class I, class PolA, class PolB> int step(Run &r, const PolA &pa, const PolB &pb) {
    if (r.step >= r.first) {
        IBVH_LAUNCH((level_kernel<I, PolA, PolB>), dim3(level_grid()), dim3(TPB), 0, r.st, (const IndexPair<I> *)r.src(),
                    (IndexPair<I> *)r.dst(), r.capacity, r.counters, r.step, r.chk, pa, pb);
        IBVH_LAUNCH_CHECK();
    }
    r.step += 1;
    r.swapped = !r.swapped;
    return IBVH_OK;
}

// flag + per-step queue counts, then per-step check counts
inline size_t counters_bytes(int64_t total_levels) { return (size_t)(total_levels + 8) * 8 * (1 + GEN_SLOTS); }

// fresh run: counters zeroed, counters[1] = initial queue size.  Resume (res->resume_step > 0): the queue the
// overflowed step reads — res->contacts_in says which buffer — still holds res->resume_num entries; the flag and
// the counts from that step on are cleared, the earlier counts stay (they are part of num_checks).
inline int begin(Run &r, void *bvtt1, void *bvtt2, int64_t capacity, void *counters, int64_t total_levels, hipStream_t st,
                 const ibvh_bfs_result *res, int64_t initial) {
    r.st = st;
    r.counters = (unsigned long long *)counters;
    r.capacity = capacity;
    r.total_levels = total_levels;
    r.q[0] = bvtt1;
    r.q[1] = bvtt2;
    r.chk = (int)(total_levels + 8);
    if (res->resume_step > 0) {
        if (res->resume_step > total_levels + 4) return IBVH_ERR_INVALID_ARG;
        r.first = (int)res->resume_step;
        const size_t half = (size_t)(total_levels + 8) * 8;
        const size_t keep = (size_t)(2 + r.first) * 8; // flag + counts of the sources of steps 0 .. first
        IBVH_HIP_CHECK(hipMemsetAsync(counters, 0, 8, st));
        IBVH_HIP_CHECK(hipMemsetAsync((char *)counters + keep, 0, half - keep, st));
        // the checks of steps 0 .. first - 1 stay; step `first` runs again from the start
        for (int k = 0; k < GEN_SLOTS; ++k)
            IBVH_HIP_CHECK(hipMemsetAsync((char *)counters + half * (size_t)(1 + k) + (size_t)r.first * 8, 0, half - (size_t)r.first * 8, st));
        return IBVH_OK;
    }
    IBVH_HIP_CHECK(hipMemsetAsync(counters, 0, counters_bytes(total_levels), st));
    unsigned long long init = (unsigned long long)initial;
    IBVH_HIP_CHECK(hipMemcpyAsync((char *)counters + 8, &init, 8, hipMemcpyHostToDevice, st)); // (pageable: copied before return)
    return IBVH_OK;
}

// the one blocking read: flag + all counts
inline int finish(const Run &r, ibvh_bfs_result *res) {
    unsigned long long host[80 * (1 + GEN_SLOTS)];
    const int nsteps = r.step;
    if (nsteps + 2 > r.chk || r.chk > 80) return IBVH_ERR_INVALID_ARG;
    IBVH_HIP_CHECK(hipMemcpyAsync(host, r.counters, (size_t)r.chk * (1 + GEN_SLOTS) * 8, hipMemcpyDeviceToHost, r.st));
    IBVH_HIP_CHECK(hipStreamSynchronize(r.st));
    const unsigned long long flag = host[0];
    if (flag != 0ull) {
        const int s = (int)flag - 1; // the first step whose destination was too small
        const int64_t need = (int64_t)host[2 + s];
        res->num_contacts = 0;
        res->num_checks = 0;
        // the source queue of step s: buffer 1 if s is even (nothing swapped yet), buffer 2 otherwise
        res->contacts_in = (s % 2 == 0) ? 1 : 2;
        res->required_capacity = need + need / 4 + 1024; // exact need of the step that overflowed + headroom
        res->resume_step = s;
        res->resume_num = (int64_t)host[1 + s];
        return IBVH_ERR_CAPACITY;
    }
    // num_checks (bfs/traverse_single.jl:25,48, traverse_pair.jl:146-150): every entry of every queue of the reference
    // is one check — the initial queue plus each node-level result, i.e. every pair a step here generated and checked;
    // what passes the last (leaf) step is the contact list
    int64_t checks = 0;
    for (int k = 0; k < GEN_SLOTS; ++k)
        for (int s = 0; s < nsteps; ++s) checks += (int64_t)host[r.chk * (1 + k) + s];
    res->num_contacts = (int64_t)host[1 + nsteps];
    res->num_checks = checks;
    res->contacts_in = r.swapped ? 2 : 1;
    res->required_capacity = 0;
    res->resume_step = 0;
    res->resume_num = 0;
    return IBVH_OK;
}
inline int capacity_error(ibvh_bfs_result *res, int64_t need) {
    res->num_contacts = 0;
    res->num_checks = 0;
    res->contacts_in = 1;
    res->required_capacity = need + need / 4 + 1024;
    res->resume_step = 0; // nothing to keep: the initial queue itself did not fit
    res->resume_num = 0;
    return IBVH_ERR_CAPACITY;
}

inline int64_t self_initial(const ibvh_bvh &b, int64_t start_level) {
    int64_t n = level_num_real(b.tree.levels, b.tree.virtual_leaves, start_level);
    return start_level != b.tree.levels ? n * (n - 1) / 2 + n : n * (n - 1) / 2;
}

inline bool same_types(const ibvh_types &x, const ibvh_types &y) {
    return x.leaf_kind == y.leaf_kind && x.leaf_float == y.leaf_float && x.node_kind == y.node_kind &&
           x.node_float == y.node_float && x.index_type == y.index_type && x.morton_type == y.morton_type;
}

template <class L, class N, class I>
int run_self(const ibvh_bvh &b, int64_t start_level, int narrow, void *bvtt1, void *bvtt2, int64_t capacity, void *counters,
             ibvh_bfs_result *res, hipStream_t st) {
    ibvh_layout lay;
    LeafLayout dl;
    layout_of(b.types, lay, &dl);
    Run r;
    const int64_t levels = b.tree.levels;
    const int64_t n0 = level_num_real(levels, b.tree.virtual_leaves, start_level);
    const int64_t total0 = self_initial(b, start_level);
    if (res->resume_step == 0 && total0 > capacity) return capacity_error(res, total0);
    if (int e = begin(r, bvtt1, bvtt2, capacity, counters, levels, st, res, total0)) return e;
    if (r.first == 0 && total0 > 0)
        IBVH_LAUNCH((fill_self_kernel<I>), dim3((unsigned)ceil_div(total0, TPB)), dim3(TPB), 0, st, (IndexPair<I> *)bvtt1,
                           n0, int64_t(1) << (start_level - 1), start_level != levels ? 1 : 0, total0);
    auto at = [&](int64_t level) {
        return SelfStep<L, N, I>{make_ref<L, N>(b, dl, level), level == levels ? 1 : 0, level < levels - 1 ? 1 : 0 /* self_checks (:44) */,
                                 narrow & IBVH_NARROW_MASK, (narrow & IBVH_OUTPUT_POSITIONS) ? 1 : 0};
    };
    if (int e = step<I>(r, Identity<I>{}, at(start_level))) return e; // the initial queue is checked ...
    for (int64_t level = start_level; level < levels; ++level)       // ... then children of what passed, level by level
        if (int e = step<I>(r, at(level), at(level + 1))) return e;
    return finish(r, res);
}

template <class L, class N, class I>
int run_pair(const ibvh_bvh &b1, const ibvh_bvh &b2, int64_t sl1, int64_t sl2, int narrow, void *bvtt1, void *bvtt2,
             int64_t capacity, void *counters, ibvh_bfs_result *res, hipStream_t st) {
    ibvh_layout lay;
    LeafLayout dl;
    layout_of(b1.types, lay, &dl);
    Run r;
    const int64_t L1 = b1.tree.levels, L2 = b2.tree.levels;
    const int64_t nr1 = level_num_real(L1, b1.tree.virtual_leaves, sl1), nr2 = level_num_real(L2, b2.tree.virtual_leaves, sl2);
    const int64_t total0 = nr1 * nr2;
    if (res->resume_step == 0 && total0 > capacity) return capacity_error(res, total0);
    if (int e = begin(r, bvtt1, bvtt2, capacity, counters, L1 + L2, st, res, total0)) return e;
    if (r.first == 0)
        IBVH_LAUNCH((fill_product_kernel<I>), dim3((unsigned)ceil_div(total0, TPB)), dim3(TPB), 0, st, (IndexPair<I> *)bvtt1,
                           nr1, nr2, int64_t(1) << (sl1 - 1), int64_t(1) << (sl2 - 1));
    // the six-phase descent of bfs/traverse_pair.jl:50-143 as a list of (levels, which side is a leaf, which descends)
    PairStep<L, N, I> seq[132];
    int ns = 0;
    int64_t l1 = sl1, l2 = sl2;
    auto add = [&](bool leaf1, bool leaf2, bool d1, bool d2) {
        PairStep<L, N, I> p;
        p.t1 = make_ref<L, N>(b1, dl, l1);
        p.t2 = make_ref<L, N>(b2, dl, l2);
        p.narrow = narrow & IBVH_NARROW_MASK;
        p.positions = (narrow & IBVH_OUTPUT_POSITIONS) ? 1 : 0;
        p.leaf1 = leaf1, p.leaf2 = leaf2, p.d1 = d1, p.d2 = d2;
        seq[ns++] = p;
    };
    while (l1 < L1 - 1 && l2 < L2 - 1) {
        add(false, false, true, true);
        ++l1, ++l2;
    }
    while (l1 < L1 - 1 && l2 == L2 - 1) {
        add(false, false, true, false);
        ++l1;
    }
    while (l2 < L2 - 1 && l1 == L1 - 1) {
        add(false, false, false, true);
        ++l2;
    }
    while (l2 == L2 && l1 < L1) {
        add(false, true, true, false);
        ++l1;
    }
    while (l1 == L1 && l2 < L2) {
        add(true, false, false, true);
        ++l2;
    }
    if (l1 == L1 - 1 && l2 == L2 - 1) {
        add(false, false, true, true);
        ++l1, ++l2;
    }
    // leaf-leaf: what passes is the contact list (num_checks is not incremented after it, bfs/traverse_pair.jl:146-150)
    add(true, true, false, false);
    if (int e = step<I>(r, Identity<I>{}, seq[0])) return e;
    for (int i = 0; i + 1 < ns; ++i)
        if (int e = step<I>(r, seq[i], seq[i + 1])) return e;
    return finish(r, res);
}

template <class L, class N, class I>
int run_rays(const ibvh_bvh &b, const void *points, const void *dirs, int64_t num_rays, int64_t start_level, int narrow,
             void *bvtt1, void *bvtt2, int64_t capacity, void *counters, ibvh_bfs_result *res, hipStream_t st) {
    using T = typename L::elt;
    ibvh_layout lay;
    LeafLayout dl;
    layout_of(b.types, lay, &dl);
    Run r;
    const int64_t levels = b.tree.levels;
    const int64_t nr = level_num_real(levels, b.tree.virtual_leaves, start_level);
    const int64_t total0 = nr * num_rays;
    if (res->resume_step == 0 && total0 > capacity) return capacity_error(res, total0);
    if (int e = begin(r, bvtt1, bvtt2, capacity, counters, levels, st, res, total0)) return e;
    if (r.first == 0)
        IBVH_LAUNCH((fill_product_kernel<I>), dim3((unsigned)ceil_div(total0, TPB)), dim3(TPB), 0, st, (IndexPair<I> *)bvtt1, nr,
                           num_rays, int64_t(1) << (start_level - 1), int64_t(1));
    auto at = [&](int64_t level) {
        return RayStep<L, N, I>{make_ref<L, N>(b, dl, level), (const T *)points, (const T *)dirs, level == levels ? 1 : 0,
                                narrow & IBVH_NARROW_MASK, (narrow & IBVH_OUTPUT_POSITIONS) ? 1 : 0};
    };
    if (int e = step<I>(r, Identity<I>{}, at(start_level))) return e;
    for (int64_t level = start_level; level < levels; ++level)
        if (int e = step<I>(r, at(level), at(level + 1))) return e;
    return finish(r, res);
}

inline int check_levels(const ibvh_bvh &b, int64_t sl) {
    // @argcheck bvh.tree.levels >= start_level >= bvh.built_level (bfs/traverse_single.jl:9-11)
    if (!(b.tree.levels >= sl && sl >= b.built_level && sl >= 1)) return IBVH_ERR_INVALID_ARG;
    if (b.tree.levels > 62) return IBVH_ERR_INVALID_ARG;
    return IBVH_OK;
}

template <class F> int dispatch(const ibvh_types &t, F &&f) {
    return dispatch_leaf_node(t, [&](auto lt, auto nt) -> int {
        return dispatch_index(t.index_type, [&](auto it) -> int { return f(lt, nt, it); });
    });
}

} // namespace bfs
} // namespace ibvh

using namespace ibvh;
using namespace ibvh::bfs;

extern "C" {

ibvh_status ibvh_bfs_counters_bytes(int64_t total_levels, size_t *bytes_out) {
    if (!bytes_out || total_levels < 0) return IBVH_ERR_INVALID_ARG;
    *bytes_out = counters_bytes(total_levels);
    return IBVH_OK;
}

ibvh_status ibvh_bfs_initial_capacity(const ibvh_bvh *bvh, int64_t start_level, int64_t *pairs_out) {
    if (!bvh || !pairs_out) return IBVH_ERR_INVALID_ARG;
    if (int e = check_levels(*bvh, start_level)) return (ibvh_status)e;
    *pairs_out = self_initial(*bvh, start_level);
    return IBVH_OK;
}
ibvh_status ibvh_bfs_pair_initial_capacity(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t sl1, int64_t sl2,
                                           int64_t *pairs_out) {
    if (!bvh1 || !bvh2 || !pairs_out) return IBVH_ERR_INVALID_ARG;
    if (int e = check_levels(*bvh1, sl1)) return (ibvh_status)e;
    if (int e = check_levels(*bvh2, sl2)) return (ibvh_status)e;
    *pairs_out = level_num_real(bvh1->tree.levels, bvh1->tree.virtual_leaves, sl1) *
                 level_num_real(bvh2->tree.levels, bvh2->tree.virtual_leaves, sl2);
    return IBVH_OK;
}
ibvh_status ibvh_bfs_rays_initial_capacity(const ibvh_bvh *bvh, int64_t num_rays, int64_t start_level, int64_t *pairs_out) {
    if (!bvh || !pairs_out || num_rays < 0) return IBVH_ERR_INVALID_ARG;
    if (int e = check_levels(*bvh, start_level)) return (ibvh_status)e;
    *pairs_out = level_num_real(bvh->tree.levels, bvh->tree.virtual_leaves, start_level) * num_rays;
    return IBVH_OK;
}

ibvh_status ibvh_traverse_bfs(const ibvh_bvh *bvh, int64_t start_level, int32_t narrow, void *bvtt1, void *bvtt2,
                              int64_t capacity, void *counters, ibvh_bfs_result *result, void *stream) {
    if (!bvh || !result) return IBVH_ERR_INVALID_ARG;
    const ibvh_bfs_result in = *result;
    *result = {0, 0, 1, 0, in.resume_step, in.resume_num};
    if (in.resume_step < 0 || in.resume_num < 0) return IBVH_ERR_INVALID_ARG;
    if (int e = check_levels(*bvh, start_level)) return (ibvh_status)e;
    if (!narrow_arg_ok(narrow, false)) return IBVH_ERR_INVALID_ARG;
    if (bvh->tree.real_nodes <= 1) return IBVH_OK; // bfs/traverse_single.jl:17-21
    if (!bvtt1 || !bvtt2 || !counters || capacity < 1) return IBVH_ERR_INVALID_ARG;
    if (bvh->types.index_type == IBVH_I32 && bvh->tree.levels > 31) return IBVH_ERR_OVERFLOW;
    return (ibvh_status)dispatch(bvh->types, [&](auto lt, auto nt, auto it) -> int {
        return run_self<typename decltype(lt)::type, typename decltype(nt)::type, typename decltype(it)::type>(
            *bvh, start_level, narrow, bvtt1, bvtt2, capacity, counters, result, (hipStream_t)stream);
    });
}

ibvh_status ibvh_traverse_pair_bfs(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t sl1, int64_t sl2, int32_t narrow,
                                   void *bvtt1, void *bvtt2, int64_t capacity, void *counters, ibvh_bfs_result *result,
                                   void *stream) {
    if (!bvh1 || !bvh2 || !result) return IBVH_ERR_INVALID_ARG;
    const ibvh_bfs_result in = *result;
    *result = {0, 0, 1, 0, in.resume_step, in.resume_num};
    if (in.resume_step < 0 || in.resume_num < 0) return IBVH_ERR_INVALID_ARG;
    if (int e = check_levels(*bvh1, sl1)) return (ibvh_status)e;
    if (int e = check_levels(*bvh2, sl2)) return (ibvh_status)e;
    if (!same_types(bvh1->types, bvh2->types)) return IBVH_ERR_UNSUPPORTED;
    if (!narrow_arg_ok(narrow, false)) return IBVH_ERR_INVALID_ARG;
    if (!bvtt1 || !bvtt2 || !counters || capacity < 1) return IBVH_ERR_INVALID_ARG;
    if (bvh1->types.index_type == IBVH_I32 && (bvh1->tree.levels > 31 || bvh2->tree.levels > 31)) return IBVH_ERR_OVERFLOW;
    return (ibvh_status)dispatch(bvh1->types, [&](auto lt, auto nt, auto it) -> int {
        return run_pair<typename decltype(lt)::type, typename decltype(nt)::type, typename decltype(it)::type>(
            *bvh1, *bvh2, sl1, sl2, narrow, bvtt1, bvtt2, capacity, counters, result, (hipStream_t)stream);
    });
}

ibvh_status ibvh_traverse_rays_bfs(const ibvh_bvh *bvh, const void *points, const void *dirs, int64_t num_rays,
                                   int64_t start_level, int32_t narrow, void *bvtt1, void *bvtt2, int64_t capacity,
                                   void *counters, ibvh_bfs_result *result, void *stream) {
    if (!bvh || !result || num_rays < 0 || !narrow_arg_ok(narrow, true)) return IBVH_ERR_INVALID_ARG;
    const ibvh_bfs_result in = *result;
    *result = {0, 0, 1, 0, in.resume_step, in.resume_num};
    if (in.resume_step < 0 || in.resume_num < 0) return IBVH_ERR_INVALID_ARG;
    if (int e = check_levels(*bvh, start_level)) return (ibvh_status)e;
    if (bvh->types.leaf_float != bvh->types.node_float) return IBVH_ERR_UNSUPPORTED;
    if (num_rays == 0) return IBVH_OK;
    if (!points || !dirs || !bvtt1 || !bvtt2 || !counters || capacity < 1) return IBVH_ERR_INVALID_ARG;
    if (bvh->types.index_type == IBVH_I32 && (bvh->tree.levels > 31 || num_rays > INT32_MAX)) return IBVH_ERR_OVERFLOW;
    return (ibvh_status)dispatch(bvh->types, [&](auto lt, auto nt, auto it) -> int {
        using L = typename decltype(lt)::type;
        using N = typename decltype(nt)::type;
        using I = typename decltype(it)::type;
        if constexpr (!std::is_same<typename L::elt, typename N::elt>::value) return (int)IBVH_ERR_UNSUPPORTED;
        else
            return run_rays<L, N, I>(*bvh, points, dirs, num_rays, start_level, narrow, bvtt1, bvtt2, capacity, counters, result,
                                     (hipStream_t)stream);
    });
}

} // extern "C"
