// ibvh_oracle.cpp — CPU ORACLE (test infrastructure, NOT product code); see ibvh_oracle.hpp.
//
// extern "C" surface: `oracle_*` twins of the libibvh entry points in include/ibvh.h, taking HOST
// pointers, so a parity test passes the same descriptors to both and compares buffers.
#include "ibvh_oracle.hpp"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <thread>

using namespace orc;

namespace {

// ------------------------------------------------------------------------------------------
// type dispatch over the supported combinations
// ------------------------------------------------------------------------------------------
template <class T> struct Tag {
    using type = T;
};

template <class F> int dispatch_index(int index_type, F &&f) {
    switch (index_type) {
    case IBVH_I32: return f(Tag<int32_t>{});
    case IBVH_I64: return f(Tag<int64_t>{});
    }
    return IBVH_ERR_UNSUPPORTED;
}
template <class F> int dispatch_morton(int morton_type, F &&f) {
    switch (morton_type) {
    case IBVH_U16: return f(Tag<uint16_t>{});
    case IBVH_U32: return f(Tag<uint32_t>{});
    case IBVH_U64: return f(Tag<uint64_t>{});
    }
    return IBVH_ERR_UNSUPPORTED;
}
template <class F> int dispatch_volume(int kind, int flt, F &&f) {
    if (kind == IBVH_BSPHERE && flt == IBVH_F32) return f(Tag<BSphere<float>>{});
    if (kind == IBVH_BSPHERE && flt == IBVH_F64) return f(Tag<BSphere<double>>{});
    if (kind == IBVH_BBOX && flt == IBVH_F32) return f(Tag<BBox<float>>{});
    if (kind == IBVH_BBOX && flt == IBVH_F64) return f(Tag<BBox<double>>{});
    return IBVH_ERR_UNSUPPORTED;
}
// leaf/node combinations that exist in the reference AND are instantiated here:
// NodeType(leaf) must exist (merge.jl): sphere->sphere, sphere->box, box->box; any node float type (build.jl:198-205
// builds whatever node_type it is given: narrower, equal or WIDER than the leaves' — the constructors of merge.jl compute in
// Julia's promoted type and convert once).
inline bool combo_ok(const ibvh_types &t) {
    if (t.node_kind == IBVH_BSPHERE && t.leaf_kind != IBVH_BSPHERE) return false;
    return true;
}
template <class F> int dispatch_leaf_node(const ibvh_types &t, F &&f) {
    if (!combo_ok(t)) return IBVH_ERR_UNSUPPORTED;
    return dispatch_volume(t.leaf_kind, t.leaf_float, [&](auto lt) -> int {
        using L = typename decltype(lt)::type;
        return dispatch_volume(t.node_kind, t.node_float, [&](auto nt) -> int {
            using N = typename decltype(nt)::type;
            constexpr bool ok = !(N::kind == IBVH_BSPHERE && L::kind != IBVH_BSPHERE);
            if constexpr (ok) return f(lt, nt);
            else return IBVH_ERR_UNSUPPORTED;
        });
    });
}
template <class F> int dispatch_all(const ibvh_types &t, F &&f) {
    return dispatch_leaf_node(t, [&](auto lt, auto nt) -> int {
        return dispatch_index(t.index_type, [&](auto it) -> int {
            return dispatch_morton(t.morton_type, [&](auto mt) -> int { return f(lt, nt, it, mt); });
        });
    });
}

// ------------------------------------------------------------------------------------------
// BVH view used by the traversals
// ------------------------------------------------------------------------------------------
template <class L, class N, class I, class M> struct View {
    using Rec = BoundingVolume<L, I, M>;
    ibvh_tree tree;
    const Rec *leaves;
    const N *nodes;
    const I *skips;
};
template <class L, class N, class I, class M> View<L, N, I, M> view_of(const ibvh_bvh &b) {
    return {b.tree, (const BoundingVolume<L, I, M> *)b.leaves, (const N *)b.nodes, (const I *)b.skips};
}

template <class Rec> inline bool narrow_eval(int narrow, const Rec &a, const Rec &b) {
    switch (narrow) {
    case IBVH_NARROW_MORTON_LT: return a.morton < b.morton;
    case IBVH_NARROW_INDEX_LT: return a.index < b.index;
    default: return true;
    }
}

// ------------------------------------------------------------------------------------------
// build — build.jl
// ------------------------------------------------------------------------------------------
// _aggregate_last_level_at! (build.jl:427-457) + _aggregate_level_at! (build.jl:503-523)
template <class L, class N, class I, class M>
void aggregate_oibvh(N *nodes, const BoundingVolume<L, I, M> *leaves, const ibvh_tree &tree, int64_t built_level) {
    // aggregate_last_level! — build.jl:381-405
    {
        int64_t level = tree.levels - 1;
        int64_t start_pos = memory_index(tree, pow2(level - 1));
        int64_t num_nodes = pow2(level - 1) - jl_shr(tree.virtual_leaves, 1);
        int64_t num_nodes_next = tree.real_leaves;
        for (int64_t i = 1; i <= num_nodes; ++i) {
            int64_t l = 2 * i - 1, r = 2 * i;
            bool rvirt = r > num_nodes_next;
            N out;
            if constexpr (std::is_same<L, N>::value) { // same_leaf_node (bounding_volumes.jl:65-70)
                if (rvirt) out = leaves[l - 1].volume;
                else out = merge_to(leaves[l - 1].volume, leaves[r - 1].volume, (N *)nullptr);
            } else {
                if (rvirt) out = convert_to(leaves[l - 1].volume, (N *)nullptr);
                else out = merge_to(leaves[l - 1].volume, leaves[r - 1].volume, (N *)nullptr);
            }
            nodes[start_pos - 1 + i - 1] = out;
        }
    }
    // aggregate_oibvh! loop — build.jl:371-375
    for (int64_t level = tree.levels - 2; level >= built_level; --level) {
        int64_t start_pos = memory_index(tree, pow2(level - 1));
        int64_t num_nodes = pow2(level - 1) - jl_shr(tree.virtual_leaves, tree.levels - level);
        int64_t start_pos_next = memory_index(tree, pow2(level));
        int64_t num_nodes_next = pow2(level) - jl_shr(tree.virtual_leaves, tree.levels - (level + 1));
        for (int64_t i = 1; i <= num_nodes; ++i) {
            int64_t l = start_pos_next + 2 * i - 2, r = start_pos_next + 2 * i - 1;
            if (r > start_pos_next + num_nodes_next - 1) nodes[start_pos - 1 + i - 1] = nodes[l - 1];
            else nodes[start_pos - 1 + i - 1] = merge_to(nodes[l - 1], nodes[r - 1], (N *)nullptr);
        }
    }
}

template <class L, class N, class I, class M>
int build_impl(const ibvh_build_desc &d, const void *volumes, void *leaves_v, void *nodes_v, void *skips_v,
               void *extrema_out) {
    using Rec = BoundingVolume<L, I, M>;
    using T = typename L::elt;
    ibvh_tree tree;
    if (!tree_shape(d.n, tree)) return IBVH_ERR_DOMAIN;
    if (d.built_level < 1 || d.built_level > tree.levels) return IBVH_ERR_INVALID_ARG; // build.jl:314
    Rec *leaves = (Rec *)leaves_v;
    const int64_t n = d.n;
    // wrap_bounding_volumes — build.jl:328-352
    if (!d.already_wrapped) {
        const L *vols = (const L *)volumes;
        for (int64_t i = 0; i < n; ++i) {
            std::memset(&leaves[i], 0, sizeof(Rec)); // deterministic padding bytes
            leaves[i].volume = vols[i];
            leaves[i].index = I(i + 1);
            leaves[i].morton = M(0);
        }
    }
    // compute_skips! — build.jl:232-239
    std::vector<int64_t> sk(tree.levels);
    compute_skips(tree, sk.data());
    for (int64_t i = 0; i < tree.levels; ++i) ((I *)skips_v)[i] = I(sk[i]);
    // morton_encode! — morton/default.jl:43-82
    T ext[6];
    if (d.compute_extrema) extrema_of(leaves, n, true, ext);
    else
        for (int k = 0; k < 3; ++k) {
            ext[k] = T(d.mins[k]);
            ext[3 + k] = T(d.maxs[k]);
        }
    if (extrema_out) std::memcpy(extrema_out, ext, sizeof(ext));
    for (int64_t i = 0; i < n; ++i) {
        T c[3];
        center(leaves[i].volume, c);
        leaves[i].morton = morton_encode_single<M>(c, ext, ext + 3);
    }
    // AK.sort!(by = bv -> bv.morton) — build.jl:248-253.  Third-party; restated by its call-site
    // contract as a STABLE ascending sort (Base.sort! on one task).  Tie order is unpinned.
    std::stable_sort(leaves, leaves + n, [](const Rec &a, const Rec &b) { return a.morton < b.morton; });
    // aggregate — build.jl:265-268
    if (tree.real_nodes >= 2) aggregate_oibvh<L, N, I, M>((N *)nodes_v, leaves, tree, d.built_level);
    return IBVH_OK;
}

// Fork-join over `threads` contiguous ranges on the OpenMP runtime (persistent team, spinning barriers): the phases
// of the baseline are short (a tree level, a radix pass) and spawning 256 std::threads for each of them cost more
// than the work itself.  Every logical range is executed exactly once whatever team size the runtime grants.
template <class F> static void parallel_ranges(int64_t n, int threads, F &&f) {
    if (threads <= 1 || n < 2 * threads) {
        f(0, int64_t(0), n);
        return;
    }
#pragma omp parallel for schedule(static, 1) num_threads(threads)
    for (int t = 0; t < threads; ++t) f(t, n * t / threads, n * (t + 1) / threads);
}
// The same, with the range cut into chunks that the threads take as they become free (the leaf-vs-tree walk of one
// leaf costs anything from a few node tests to hundreds: equal contiguous shares leave threads idle at the end).
// Results do not depend on who processes which chunk: every leaf writes its own count / its own output range.
template <class F> static void parallel_chunks(int64_t n, int threads, int64_t chunk, F &&f) {
    if (threads <= 1 || n < 2 * threads) {
        f(0, int64_t(0), n);
        return;
    }
    const int64_t nchunks = (n + chunk - 1) / chunk;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int64_t c = 0; c < nchunks; ++c) f(0, c * chunk, (c + 1) * chunk < n ? (c + 1) * chunk : n);
}

// Work counters of the walk (SURVEY.md §8d "touched bytes"): node tests = calls of the node predicate (one node volume
// fetched and tested), leaf tests = leaves reached (one leaf record fetched and tested).  The reference has such a
// counter for BFS only (num_checks, bfs/traverse_single.jl:25,48); this is the same idea for the leaf-vs-tree walk.
struct TestCounts {
    int64_t node = 0, leaf = 0;
};
template <class P> struct CountingPolicy {
    P &p;
    TestCounts &tc;
    bool skip(int64_t inode, int64_t ilevel) const { return p.skip(inode, ilevel); }
    void leaf(int64_t pos) const {
        ++tc.leaf;
        p.leaf(pos);
    }
    bool node(int64_t mem) const {
        ++tc.node;
        return p.node(mem);
    }
};
template <class I, class ViewT, class Policy> inline void lvt_walk(const ViewT &bvh, int64_t start_level, Policy &&pol);
template <class I, class ViewT, class P> inline void lvt_walk_counted(const ViewT &bvh, int64_t start_level, P &p, TestCounts *tc) {
    if (tc) {
        CountingPolicy<P> cp{p, *tc};
        lvt_walk<I>(bvh, start_level, cp);
    } else {
        lvt_walk<I>(bvh, start_level, p);
    }
}

// ------------------------------------------------------------------------------------------
// LVT walkers — lvt/traverse_single.jl:136-208, lvt/traverse_pair.jl:176-244,
//               raytrace/leaf_vs_tree/leaf_vs_tree.jl:170-228
// ------------------------------------------------------------------------------------------
// One DFS with the reference's explicit 32-entry stack: descend left while touching, push the
// real right child.  `Visit` supplies the three policy hooks.
template <class I, class ViewT, class Policy> inline void lvt_walk(const ViewT &bvh, int64_t start_level, Policy &&pol) {
    const ibvh_tree &tree = bvh.tree;
    int64_t inode_start = pow2(start_level - 1);
    int64_t nreal = pow2(start_level - 1) - jl_shr(tree.virtual_leaves, tree.levels - start_level);
    int64_t inode_end = inode_start + nreal - 1;
    int64_t stack[64];
    for (int64_t inode_root = inode_start; inode_root <= inode_end; ++inode_root) {
        int64_t istack = 0;
        int64_t inode = inode_root;
        while (true) {
            int64_t ilevel = ilog2_down(inode) + 1;
            if (pol.skip(inode, ilevel)) {
                // subtree skipped (self-traversal de-dup, traverse_single.jl:165-167)
            } else if (ilevel == tree.levels) {
                pol.leaf(inode - pow2(tree.levels - 1) + 1); // 1-based leaf position
            } else {
                if (pol.node(inode - (int64_t)bvh.skips[ilevel - 1])) { // nodes[inode - skips[ilevel]]
                    if (!isvirtual(tree, 2 * inode + 1)) stack[istack++] = 2 * inode + 1;
                    inode = 2 * inode;
                    continue;
                }
            }
            if (istack == 0) break;
            inode = stack[--istack];
        }
    }
}

// traverse_lvt_single! for leaf `ileaf` (1-based).  emit(first, second) per contact, in order.
template <class L, class N, class I, class M, class Emit>
inline void lvt_single_leaf(const View<L, N, I, M> &bvh, int64_t ileaf, int64_t start_level, int narrow, Emit &&emit,
                            TestCounts *tc = nullptr) {
    const auto &bv = bvh.leaves[ileaf - 1];
    // bv_node = bv.volume isa NodeType ? bv.volume : NodeType(bv.volume) — traverse_single.jl:154-155
    N bv_node = convert_to(bv.volume, (N *)nullptr);
    struct P {
        const View<L, N, I, M> &bvh;
        const BoundingVolume<L, I, M> &bv;
        const N &bv_node;
        int64_t ileaf;
        int narrow;
        Emit &emit;
        bool skip(int64_t inode, int64_t ilevel) const {
            int64_t rightmost = ((inode + 1) << (bvh.tree.levels - ilevel)) - 1;
            return rightmost <= ileaf + pow2(bvh.tree.levels - 1) - 1;
        }
        void leaf(int64_t pos) const {
            const auto &leaf = bvh.leaves[pos - 1];
            if (iscontact(bv.volume, leaf.volume) && narrow_eval(narrow, bv, leaf)) {
                if (bv.index > leaf.index) emit(leaf.index, bv.index);
                else emit(bv.index, leaf.index);
            }
        }
        bool node(int64_t mem) const { return iscontact(bv_node, bvh.nodes[mem - 1]); }
    } p{bvh, bv, bv_node, ileaf, narrow, emit};
    lvt_walk_counted<I>(bvh, start_level, p, tc);
}

// traverse_lvt_pair! — bv from the driving BVH against the other tree; FLIP restores order.
template <class L, class N, class I, class M, class Emit>
inline void lvt_pair_leaf(const BoundingVolume<L, I, M> &bv, const View<L, N, I, M> &bvh, int64_t start_level,
                          int narrow, bool flip, Emit &&emit, TestCounts *tc = nullptr) {
    N bv_node = convert_to(bv.volume, (N *)nullptr);
    struct P {
        const View<L, N, I, M> &bvh;
        const BoundingVolume<L, I, M> &bv;
        const N &bv_node;
        int narrow;
        bool flip;
        Emit &emit;
        bool skip(int64_t, int64_t) const { return false; }
        void leaf(int64_t pos) const {
            const auto &leaf = bvh.leaves[pos - 1];
            if (iscontact(bv.volume, leaf.volume) &&
                (flip ? narrow_eval(narrow, leaf, bv) : narrow_eval(narrow, bv, leaf))) {
                if (flip) emit(leaf.index, bv.index);
                else emit(bv.index, leaf.index);
            }
        }
        bool node(int64_t mem) const { return iscontact(bv_node, bvh.nodes[mem - 1]); }
    } p{bvh, bv, bv_node, narrow, flip, emit};
    lvt_walk_counted<I>(bvh, start_level, p, tc);
}

// traverse_ray_lvt!
template <class L, class N, class I, class M, class Emit>
inline void lvt_ray(const typename L::elt *point, const typename L::elt *dir, int64_t iray,
                    const View<L, N, I, M> &bvh, int64_t start_level, Emit &&emit, TestCounts *tc = nullptr) {
    struct P {
        const View<L, N, I, M> &bvh;
        const typename L::elt *p;
        const typename L::elt *d;
        int64_t iray;
        Emit &emit;
        bool skip(int64_t, int64_t) const { return false; }
        void leaf(int64_t pos) const {
            const auto &leaf = bvh.leaves[pos - 1];
            if (isintersection(leaf.volume, p, d)) emit(leaf.index, I(iray));
        }
        bool node(int64_t mem) const { return isintersection(bvh.nodes[mem - 1], p, d); }
    } p{bvh, point, dir, iray, emit};
    lvt_walk_counted<I>(bvh, start_level, p, tc);
}

// check + inclusive scan of counts (AK.accumulate!, traverse_single.jl:57) with overflow guard
template <class I> int scan_counts(I *counts, int64_t n, int64_t *total_out) {
    int64_t acc = 0;
    for (int64_t i = 0; i < n; ++i) {
        acc += (int64_t)counts[i];
        if (acc > (int64_t)std::numeric_limits<I>::max()) return IBVH_ERR_OVERFLOW;
        counts[i] = I(acc);
    }
    *total_out = acc;
    return IBVH_OK;
}

inline int check_levels(const ibvh_bvh &b, int64_t start_level, bool lvt) {
    // @argcheck bvh.built_level <= start_level <= bvh.tree.levels <= 32
    if (!(b.built_level <= start_level && start_level <= b.tree.levels)) return IBVH_ERR_INVALID_ARG;
    if (lvt && b.tree.levels > 32) return IBVH_ERR_INVALID_ARG;
    return IBVH_OK;
}
inline bool same_types(const ibvh_types &a, const ibvh_types &b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }

// ------------------------------------------------------------------------------------------
// BFS — bfs/traverse_single.jl + traverse_single_cpu.jl (single-task order),
//       bfs/traverse_pair.jl + traverse_pair_cpu.jl, raytrace/breadth_first/*.jl
// ------------------------------------------------------------------------------------------
struct Pair64 {
    int64_t a, b;
};

template <class L, class N, class I, class M>
void bfs_single(const View<L, N, I, M> &bvh, int64_t start_level, int narrow, std::vector<IndexPair<I>> &contacts,
                int64_t &num_checks, int64_t &peak) {
    const ibvh_tree &tree = bvh.tree;
    std::vector<Pair64> src, dst;
    // fill_initial_bvtt_single! CPU branch — bfs/traverse_single.jl:151-166
    int64_t level_nodes = pow2(start_level - 1);
    int64_t num_real = level_nodes - jl_shr(tree.virtual_leaves, tree.levels - start_level);
    for (int64_t i = level_nodes; i <= level_nodes + num_real - 1; ++i) {
        if (start_level != tree.levels) src.push_back({i, i});
        for (int64_t j = i + 1; j <= level_nodes + num_real - 1; ++j) src.push_back({i, j});
    }
    num_checks = (int64_t)src.size();
    peak = (int64_t)src.size();
    int64_t level = start_level;
    while (level < tree.levels) {
        bool self_checks = level < tree.levels - 1; // bfs/traverse_single.jl:44
        // traverse_nodes_range! — traverse_single_cpu.jl:64-133
        int64_t vnl = jl_shr(tree.virtual_leaves, tree.levels - (level - 1));
        int64_t num_skips = 2 * vnl - count_ones(vnl);
        dst.clear();
        for (const Pair64 &pr : src) {
            int64_t i1 = pr.a, i2 = pr.b;
            if (i1 == i2) {
                if (isvirtual(tree, 2 * i1 + 1)) {
                    if (self_checks) dst.push_back({2 * i1, 2 * i1});
                } else if (self_checks) {
                    dst.push_back({2 * i1, 2 * i1});
                    dst.push_back({2 * i1, 2 * i1 + 1});
                    dst.push_back({2 * i1 + 1, 2 * i1 + 1});
                } else {
                    dst.push_back({2 * i1, 2 * i1 + 1});
                }
            } else {
                const N &n1 = bvh.nodes[i1 - num_skips - 1];
                const N &n2 = bvh.nodes[i2 - num_skips - 1];
                if (iscontact(n1, n2)) {
                    if (isvirtual(tree, 2 * i2 + 1)) {
                        dst.push_back({2 * i1, 2 * i2});
                        dst.push_back({2 * i1 + 1, 2 * i2});
                    } else {
                        dst.push_back({2 * i1, 2 * i2});
                        dst.push_back({2 * i1, 2 * i2 + 1});
                        dst.push_back({2 * i1 + 1, 2 * i2});
                        dst.push_back({2 * i1 + 1, 2 * i2 + 1});
                    }
                }
            }
        }
        num_checks += (int64_t)dst.size();
        peak = std::max<int64_t>(peak, (int64_t)dst.size());
        src.swap(dst);
        ++level;
    }
    // traverse_leaves_range! — traverse_single_cpu.jl:184-219
    int64_t num_above = pow2(tree.levels - 1) - 1;
    contacts.clear();
    for (const Pair64 &pr : src) {
        const auto &l1 = bvh.leaves[pr.a - num_above - 1];
        const auto &l2 = bvh.leaves[pr.b - num_above - 1];
        if (iscontact(l1.volume, l2.volume) && narrow_eval(narrow, l1, l2)) {
            if (l1.index > l2.index) contacts.push_back({l2.index, l1.index});
            else contacts.push_back({l1.index, l2.index});
        }
    }
}

template <class L, class N, class I, class M>
void bfs_pair(const View<L, N, I, M> &b1, const View<L, N, I, M> &b2, int64_t start_level1, int64_t start_level2,
              int narrow, std::vector<IndexPair<I>> &contacts, int64_t &num_checks, int64_t &peak) {
    const ibvh_tree &t1 = b1.tree, &t2 = b2.tree;
    std::vector<Pair64> src, dst;
    // fill_initial_bvtt_pair! CPU branch — bfs/traverse_pair.jl:206-218
    int64_t ln1 = pow2(start_level1 - 1), ln2 = pow2(start_level2 - 1);
    int64_t nr1 = ln1 - jl_shr(t1.virtual_leaves, t1.levels - start_level1);
    int64_t nr2 = ln2 - jl_shr(t2.virtual_leaves, t2.levels - start_level2);
    for (int64_t i = ln1; i <= ln1 + nr1 - 1; ++i)
        for (int64_t j = ln2; j <= ln2 + nr2 - 1; ++j) src.push_back({i, j});
    num_checks = (int64_t)src.size();
    peak = (int64_t)src.size();
    auto skips_at = [](const ibvh_tree &t, int64_t level) {
        int64_t vnl = jl_shr(t.virtual_leaves, t.levels - (level - 1));
        return 2 * vnl - count_ones(vnl);
    };
    auto finish_level = [&]() {
        num_checks += (int64_t)dst.size();
        peak = std::max<int64_t>(peak, (int64_t)dst.size());
        src.swap(dst);
    };
    // traverse_nodes_pair_range! — traverse_pair_cpu.jl:67-126
    auto nodes_pair = [&](int64_t level1, int64_t level2) {
        int64_t s1 = skips_at(t1, level1), s2 = skips_at(t2, level2);
        dst.clear();
        for (const Pair64 &pr : src) {
            if (!iscontact(b1.nodes[pr.a - s1 - 1], b2.nodes[pr.b - s2 - 1])) continue;
            bool v1 = isvirtual(t1, 2 * pr.a + 1), v2 = isvirtual(t2, 2 * pr.b + 1);
            dst.push_back({2 * pr.a, 2 * pr.b});
            if (v1) {
                if (!v2) dst.push_back({2 * pr.a, 2 * pr.b + 1});
            } else if (v2) {
                dst.push_back({2 * pr.a + 1, 2 * pr.b});
            } else {
                dst.push_back({2 * pr.a, 2 * pr.b + 1});
                dst.push_back({2 * pr.a + 1, 2 * pr.b});
                dst.push_back({2 * pr.a + 1, 2 * pr.b + 1});
            }
        }
        finish_level();
    };
    // traverse_nodes_left_range! / _right_range! — traverse_pair_cpu.jl:195-235, 304-344
    auto nodes_left = [&](int64_t level1, int64_t level2) {
        int64_t s1 = skips_at(t1, level1), s2 = skips_at(t2, level2);
        dst.clear();
        for (const Pair64 &pr : src) {
            if (!iscontact(b1.nodes[pr.a - s1 - 1], b2.nodes[pr.b - s2 - 1])) continue;
            dst.push_back({2 * pr.a, pr.b});
            if (!isvirtual(t1, 2 * pr.a + 1)) dst.push_back({2 * pr.a + 1, pr.b});
        }
        finish_level();
    };
    auto nodes_right = [&](int64_t level1, int64_t level2) {
        int64_t s1 = skips_at(t1, level1), s2 = skips_at(t2, level2);
        dst.clear();
        for (const Pair64 &pr : src) {
            if (!iscontact(b1.nodes[pr.a - s1 - 1], b2.nodes[pr.b - s2 - 1])) continue;
            dst.push_back({pr.a, 2 * pr.b});
            if (!isvirtual(t2, 2 * pr.b + 1)) dst.push_back({pr.a, 2 * pr.b + 1});
        }
        finish_level();
    };
    // traverse_nodes_leaves_left_range! / _right_range! — traverse_pair_cpu.jl:408-451, 515-558
    auto nodes_leaves_left = [&](int64_t level1) {
        int64_t s1 = skips_at(t1, level1);
        int64_t above2 = pow2(t2.levels - 1) - 1;
        dst.clear();
        for (const Pair64 &pr : src) {
            if (!iscontact(b1.nodes[pr.a - s1 - 1], b2.leaves[pr.b - above2 - 1].volume)) continue;
            dst.push_back({2 * pr.a, pr.b});
            if (!isvirtual(t1, 2 * pr.a + 1)) dst.push_back({2 * pr.a + 1, pr.b});
        }
        finish_level();
    };
    auto nodes_leaves_right = [&](int64_t level2) {
        int64_t s2 = skips_at(t2, level2);
        int64_t above1 = pow2(t1.levels - 1) - 1;
        dst.clear();
        for (const Pair64 &pr : src) {
            if (!iscontact(b1.leaves[pr.a - above1 - 1].volume, b2.nodes[pr.b - s2 - 1])) continue;
            dst.push_back({pr.a, 2 * pr.b});
            if (!isvirtual(t2, 2 * pr.b + 1)) dst.push_back({pr.a, 2 * pr.b + 1});
        }
        finish_level();
    };
    // six-phase descent — bfs/traverse_pair.jl:50-143
    int64_t level1 = start_level1, level2 = start_level2;
    while (level1 < t1.levels - 1 && level2 < t2.levels - 1) {
        nodes_pair(level1, level2);
        ++level1;
        ++level2;
    }
    while (level1 < t1.levels - 1 && level2 == t2.levels - 1) {
        nodes_left(level1, level2);
        ++level1;
    }
    while (level2 < t2.levels - 1 && level1 == t1.levels - 1) {
        nodes_right(level1, level2);
        ++level2;
    }
    while (level2 == t2.levels && level1 < t1.levels) {
        nodes_leaves_left(level1);
        ++level1;
    }
    while (level1 == t1.levels && level2 < t2.levels) {
        nodes_leaves_right(level2);
        ++level2;
    }
    if (level1 == t1.levels - 1 && level2 == t2.levels - 1) {
        nodes_pair(level1, level2);
        ++level1;
        ++level2;
    }
    // traverse_leaves_pair_range! — traverse_pair_cpu.jl:614-645
    int64_t above1 = pow2(t1.levels - 1) - 1, above2 = pow2(t2.levels - 1) - 1;
    contacts.clear();
    for (const Pair64 &pr : src) {
        const auto &l1 = b1.leaves[pr.a - above1 - 1];
        const auto &l2 = b2.leaves[pr.b - above2 - 1];
        if (iscontact(l1.volume, l2.volume) && narrow_eval(narrow, l1, l2)) contacts.push_back({l1.index, l2.index});
    }
}

template <class L, class N, class I, class M>
void bfs_rays(const View<L, N, I, M> &bvh, const typename L::elt *points, const typename L::elt *dirs,
              int64_t num_rays, int64_t start_level, std::vector<IndexPair<I>> &contacts, int64_t &num_checks,
              int64_t &peak) {
    const ibvh_tree &tree = bvh.tree;
    std::vector<Pair64> src, dst;
    // fill_initial_bvtt_rays! CPU branch — raytrace/breadth_first/breadth_first.jl:126-136
    int64_t level_nodes = pow2(start_level - 1);
    int64_t num_real = level_nodes - jl_shr(tree.virtual_leaves, tree.levels - start_level);
    for (int64_t i = level_nodes; i <= level_nodes + num_real - 1; ++i)
        for (int64_t j = 1; j <= num_rays; ++j) src.push_back({i, j});
    num_checks = (int64_t)src.size();
    peak = num_checks;
    int64_t level = start_level;
    while (level < tree.levels) {
        // traverse_rays_nodes_range! — raytrace_cpu.jl:62-101
        int64_t vnl = jl_shr(tree.virtual_leaves, tree.levels - (level - 1));
        int64_t num_skips = 2 * vnl - count_ones(vnl);
        dst.clear();
        for (const Pair64 &pr : src) {
            const N &node = bvh.nodes[pr.a - num_skips - 1];
            if (isintersection(node, points + 3 * (pr.b - 1), dirs + 3 * (pr.b - 1))) {
                dst.push_back({2 * pr.a, pr.b});
                if (!isvirtual(tree, 2 * pr.a + 1)) dst.push_back({2 * pr.a + 1, pr.b});
            }
        }
        num_checks += (int64_t)dst.size();
        peak = std::max<int64_t>(peak, (int64_t)dst.size());
        src.swap(dst);
        ++level;
    }
    // traverse_rays_leaves_range! — raytrace_cpu.jl:151-182
    int64_t num_above = pow2(tree.levels - 1) - 1;
    contacts.clear();
    for (const Pair64 &pr : src) {
        const auto &leaf = bvh.leaves[pr.a - num_above - 1];
        if (isintersection(leaf.volume, points + 3 * (pr.b - 1), dirs + 3 * (pr.b - 1)))
            contacts.push_back({leaf.index, I(pr.b)});
    }
}

template <class I>
int bfs_finish(const std::vector<IndexPair<I>> &contacts, int64_t num_checks, int64_t peak, void *bvtt1,
               int64_t capacity, ibvh_bfs_result *res) {
    res->num_checks = num_checks;
    res->num_contacts = (int64_t)contacts.size();
    res->contacts_in = 1;
    res->required_capacity = peak;
    if (peak > capacity) return IBVH_ERR_CAPACITY;
    if (!contacts.empty()) std::memcpy(bvtt1, contacts.data(), contacts.size() * sizeof(IndexPair<I>));
    return IBVH_OK;
}

// SplitMix64 counter-based generator shared (by specification, not by code) with libibvh:
//   z = seed + (ctr+1)*0x9E3779B97F4A7C15 ; z ^= z>>30; z*=0xBF58476D1CE4E5B9; z^=z>>27;
//   z *= 0x94D049BB133111EB; z ^= z>>31 ; u = (z >> 40) * 2^-24  in [0,1)
inline uint64_t splitmix64(uint64_t seed, uint64_t ctr) {
    uint64_t z = seed + (ctr + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline float u01(uint64_t seed, uint64_t ctr) { return float(splitmix64(seed, ctr) >> 40) * (1.0f / 16777216.0f); }

} // namespace

// ==========================================================================================
// extern "C" surface
// ==========================================================================================
extern "C" {

int oracle_tree_shape(int64_t n, ibvh_tree *out) { return tree_shape(n, *out) ? IBVH_OK : IBVH_ERR_DOMAIN; }
int oracle_compute_skips(const ibvh_tree *t, int64_t *skips) {
    compute_skips(*t, skips);
    return IBVH_OK;
}
int oracle_memory_index(const ibvh_tree *t, int64_t idx, int64_t *out) {
    if (!(1 <= idx && idx <= pow2(t->levels) - 1)) return IBVH_ERR_INVALID_ARG;
    *out = memory_index(*t, idx);
    return IBVH_OK;
}
int oracle_level_indices(const ibvh_tree *t, int64_t level, int64_t *start, int64_t *stop) {
    if (!(1 <= level && level <= t->levels)) return IBVH_ERR_INVALID_ARG;
    level_indices(*t, level, *start, *stop);
    return IBVH_OK;
}
int oracle_isvirtual(const ibvh_tree *t, int64_t idx, int32_t *out) {
    if (!(1 <= idx && idx <= pow2(t->levels) - 1)) return IBVH_ERR_INVALID_ARG;
    *out = isvirtual(*t, idx) ? 1 : 0;
    return IBVH_OK;
}
int oracle_compute_build_level(const ibvh_tree *t, double frac, int64_t *out) {
    if (!(0 <= frac && frac <= 1)) return IBVH_ERR_INVALID_ARG;
    *out = compute_build_level(*t, frac);
    return IBVH_OK;
}
int oracle_layout_of(const ibvh_types *t, ibvh_layout *out) {
    return dispatch_all(*t, [&](auto lt, auto nt, auto it, auto mt) -> int {
        using L = typename decltype(lt)::type;
        using N = typename decltype(nt)::type;
        using I = typename decltype(it)::type;
        using M = typename decltype(mt)::type;
        using Rec = BoundingVolume<L, I, M>;
        out->volume_bytes = sizeof(L);
        out->node_bytes = sizeof(N);
        out->index_off = offsetof(Rec, index);
        out->morton_off = offsetof(Rec, morton);
        out->leaf_bytes = sizeof(Rec);
        out->pair_bytes = sizeof(IndexPair<I>);
        return IBVH_OK;
    });
}

uint16_t oracle_morton_split3_u16(uint16_t v) { return morton_split3(v); }
uint32_t oracle_morton_split3_u32(uint32_t v) { return morton_split3(v); }
uint64_t oracle_morton_split3_u64(uint64_t v) { return morton_split3(v); }

// ---- geometry single-shots (for the reference's unit-test goldens) -----------------------
int oracle_iscontact(int kind_a, int flt_a, const void *a, int kind_b, int flt_b, const void *b, int32_t *out) {
    return dispatch_volume(kind_a, flt_a, [&](auto ta) -> int {
        using A = typename decltype(ta)::type;
        return dispatch_volume(kind_b, flt_b, [&](auto tb) -> int {
            using B = typename decltype(tb)::type;
            *out = iscontact(*(const A *)a, *(const B *)b) ? 1 : 0;
            return IBVH_OK;
        });
    });
}
int oracle_isintersection(int kind, int flt, const void *vol, const void *p, const void *d, int32_t *out) {
    return dispatch_volume(kind, flt, [&](auto tv) -> int {
        using V = typename decltype(tv)::type;
        using T = typename V::elt;
        *out = isintersection(*(const V *)vol, (const T *)p, (const T *)d) ? 1 : 0;
        return IBVH_OK;
    });
}
// node = NodeType(a) when b == NULL else NodeType(a, b)
int oracle_merge(const ibvh_types *t, const void *a, const void *b, void *out) {
    return dispatch_leaf_node(*t, [&](auto lt, auto nt) -> int {
        using L = typename decltype(lt)::type;
        using N = typename decltype(nt)::type;
        if (b) *(N *)out = merge_to(*(const L *)a, *(const L *)b, (N *)nullptr);
        else *(N *)out = convert_to(*(const L *)a, (N *)nullptr);
        return IBVH_OK;
    });
}
int oracle_volumes_from_triangles(int kind, int flt, const void *tris, int64_t n, void *out) {
    return dispatch_volume(kind, flt, [&](auto tv) -> int {
        using V = typename decltype(tv)::type;
        using T = typename V::elt;
        const T *p = (const T *)tris;
        for (int64_t i = 0; i < n; ++i) {
            if constexpr (V::kind == IBVH_BSPHERE) ((V *)out)[i] = bsphere_from_triangle(p + 9 * i, p + 9 * i + 3, p + 9 * i + 6);
            else ((V *)out)[i] = bbox_from_triangle(p + 9 * i, p + 9 * i + 3, p + 9 * i + 6);
        }
        return IBVH_OK;
    });
}

// ---- Morton pieces ----------------------------------------------------------------------
int oracle_extrema(const ibvh_types *t, const void *records, int32_t wrapped, int64_t n, int32_t expand, void *out) {
    return dispatch_all(*t, [&](auto lt, auto, auto it, auto mt) -> int {
        using L = typename decltype(lt)::type;
        using I = typename decltype(it)::type;
        using M = typename decltype(mt)::type;
        using T = typename L::elt;
        if (wrapped) extrema_of((const BoundingVolume<L, I, M> *)records, n, expand != 0, (T *)out);
        else extrema_of((const L *)records, n, expand != 0, (T *)out);
        return IBVH_OK;
    });
}
int oracle_morton_keys(const ibvh_types *t, const void *records, int32_t wrapped, int64_t n, const void *extrema,
                       void *keys_out) {
    return dispatch_all(*t, [&](auto lt, auto, auto it, auto mt) -> int {
        using L = typename decltype(lt)::type;
        using I = typename decltype(it)::type;
        using M = typename decltype(mt)::type;
        using T = typename L::elt;
        const T *ext = (const T *)extrema;
        for (int64_t i = 0; i < n; ++i) {
            T c[3];
            if (wrapped) center(((const BoundingVolume<L, I, M> *)records)[i].volume, c);
            else center(((const L *)records)[i], c);
            M m = morton_encode_single<M>(c, ext, ext + 3);
            if (sizeof(M) == 8) ((uint64_t *)keys_out)[i] = (uint64_t)m;
            else ((uint32_t *)keys_out)[i] = (uint32_t)m;
        }
        return IBVH_OK;
    });
}
// stable sort of (key, value) pairs by key — the AK.sort! contract
int oracle_sort_pairs(int32_t key_bytes, int64_t n, void *keys, uint32_t *vals) {
    std::vector<int64_t> perm(n);
    for (int64_t i = 0; i < n; ++i) perm[i] = i;
    if (key_bytes == 4) {
        uint32_t *k = (uint32_t *)keys;
        std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return k[a] < k[b]; });
        std::vector<uint32_t> k2(n), v2(n);
        for (int64_t i = 0; i < n; ++i) {
            k2[i] = k[perm[i]];
            v2[i] = vals[perm[i]];
        }
        std::memcpy(k, k2.data(), n * 4);
        std::memcpy(vals, v2.data(), n * 4);
    } else if (key_bytes == 8) {
        uint64_t *k = (uint64_t *)keys;
        std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return k[a] < k[b]; });
        std::vector<uint64_t> k2(n);
        std::vector<uint32_t> v2(n);
        for (int64_t i = 0; i < n; ++i) {
            k2[i] = k[perm[i]];
            v2[i] = vals[perm[i]];
        }
        std::memcpy(k, k2.data(), n * 8);
        std::memcpy(vals, v2.data(), n * 4);
    } else
        return IBVH_ERR_INVALID_ARG;
    return IBVH_OK;
}

// ---- build -------------------------------------------------------------------------------
int oracle_build(const ibvh_build_desc *d, const void *volumes, void *leaves, void *nodes, void *skips,
                 void *extrema_out) {
    return dispatch_all(d->types, [&](auto lt, auto nt, auto it, auto mt) -> int {
        return build_impl<typename decltype(lt)::type, typename decltype(nt)::type, typename decltype(it)::type,
                          typename decltype(mt)::type>(*d, volumes, leaves, nodes, skips, extrema_out);
    });
}
int oracle_aggregate(const ibvh_types *t, const ibvh_tree *tree, int64_t built_level, const void *leaves,
                     void *nodes) {
    return dispatch_all(*t, [&](auto lt, auto nt, auto it, auto mt) -> int {
        using L = typename decltype(lt)::type;
        using N = typename decltype(nt)::type;
        using I = typename decltype(it)::type;
        using M = typename decltype(mt)::type;
        if (tree->real_nodes >= 2)
            aggregate_oibvh<L, N, I, M>((N *)nodes, (const BoundingVolume<L, I, M> *)leaves, *tree, built_level);
        return IBVH_OK;
    });
}

// ---- LVT ---------------------------------------------------------------------------------
int oracle_traverse_lvt_count(const ibvh_bvh *bvh, int64_t start_level, int32_t narrow, void *counts,
                              int64_t *total_out) {
    if (int e = check_levels(*bvh, start_level, true)) return e;
    *total_out = 0;
    if (bvh->tree.real_nodes <= 1) return IBVH_OK; // traverse_single.jl:17-21
    return dispatch_all(bvh->types, [&](auto lt, auto nt, auto it, auto mt) -> int {
        using L = typename decltype(lt)::type;
        using N = typename decltype(nt)::type;
        using I = typename decltype(it)::type;
        using M = typename decltype(mt)::type;
        auto v = view_of<L, N, I, M>(*bvh);
        I *c = (I *)counts;
        int64_t n = bvh->tree.real_leaves;
        for (int64_t i = 1; i <= n; ++i) {
            int64_t cnt = 0;
            auto emit = [&](I, I) { ++cnt; };
            lvt_single_leaf(v, i, start_level, narrow, emit);
            if (cnt > (int64_t)std::numeric_limits<I>::max()) return IBVH_ERR_OVERFLOW;
            c[i - 1] = I(cnt);
        }
        return scan_counts(c, n, total_out);
    });
}
int oracle_traverse_lvt_write(const ibvh_bvh *bvh, int64_t start_level, int32_t narrow, const void *counts,
                              void *contacts) {
    if (int e = check_levels(*bvh, start_level, true)) return e;
    if (bvh->tree.real_nodes <= 1) return IBVH_OK;
    return dispatch_all(bvh->types, [&](auto lt, auto nt, auto it, auto mt) -> int {
        using L = typename decltype(lt)::type;
        using N = typename decltype(nt)::type;
        using I = typename decltype(it)::type;
        using M = typename decltype(mt)::type;
        auto v = view_of<L, N, I, M>(*bvh);
        const I *c = (const I *)counts;
        IndexPair<I> *out = (IndexPair<I> *)contacts;
        int64_t n = bvh->tree.real_leaves;
        for (int64_t i = 1; i <= n; ++i) {
            int64_t w = (i == 1) ? 0 : (int64_t)c[i - 2]; // iwrite = prefix[i-1] + 1 (1-based), :117-121
            auto emit = [&](I a, I b) { out[w++] = {a, b}; };
            lvt_single_leaf(v, i, start_level, narrow, emit);
        }
        return IBVH_OK;
    });
}

static int pair_lvt(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t sl1, int64_t sl2, int32_t narrow,
                    void *counts, int64_t *total_out, void *contacts, int threads = 1, TestCounts *tc_out = nullptr) {
    if (int e = check_levels(*bvh1, sl1, true)) return e;
    if (int e = check_levels(*bvh2, sl2, true)) return e;
    if (!same_types(bvh1->types, bvh2->types)) return IBVH_ERR_UNSUPPORTED;
    // traverse_pair.jl:15-36: the BVH with more leaves drives; flip restores (bvh1, bvh2) order
    bool flip = !(bvh1->tree.real_leaves >= bvh2->tree.real_leaves);
    const ibvh_bvh *drv = flip ? bvh2 : bvh1, *oth = flip ? bvh1 : bvh2;
    int64_t sl_other = flip ? sl1 : sl2;
    return dispatch_all(bvh1->types, [&](auto lt, auto nt, auto it, auto mt) -> int {
        using L = typename decltype(lt)::type;
        using N = typename decltype(nt)::type;
        using I = typename decltype(it)::type;
        using M = typename decltype(mt)::type;
        auto vd = view_of<L, N, I, M>(*drv);
        auto vo = view_of<L, N, I, M>(*oth);
        I *c = (I *)counts;
        int64_t n = drv->tree.real_leaves;
        // (threads > 1: contiguous chunks handed out dynamically, lvt/traverse_pair.jl:119-173's task ranges; every
        // driving leaf writes its own count / its own output range, so the result does not depend on the team)
        if (!contacts) {
            std::mutex mu;
            parallel_chunks(n, threads, 2048, [&](int, int64_t lo, int64_t hi) {
                TestCounts tc;
                for (int64_t i = lo + 1; i <= hi; ++i) {
                    int64_t cnt = 0;
                    auto emit = [&](I, I) { ++cnt; };
                    lvt_pair_leaf(vd.leaves[i - 1], vo, sl_other, narrow, flip, emit, tc_out ? &tc : nullptr);
                    c[i - 1] = I(cnt);
                }
                if (tc_out) {
                    std::lock_guard<std::mutex> g(mu);
                    tc_out->node += tc.node;
                    tc_out->leaf += tc.leaf;
                }
            });
            return scan_counts(c, n, total_out);
        }
        IndexPair<I> *out = (IndexPair<I> *)contacts;
        parallel_chunks(n, threads, 2048, [&](int, int64_t lo, int64_t hi) {
            for (int64_t i = lo + 1; i <= hi; ++i) {
                int64_t w = (i == 1) ? 0 : (int64_t)c[i - 2];
                auto emit = [&](I a, I b) { out[w++] = {a, b}; };
                lvt_pair_leaf(vd.leaves[i - 1], vo, sl_other, narrow, flip, emit);
            }
        });
        return IBVH_OK;
    });
}
int oracle_traverse_pair_lvt_count(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t sl1, int64_t sl2,
                                   int32_t narrow, void *counts, int64_t *total_out) {
    *total_out = 0;
    return pair_lvt(bvh1, bvh2, sl1, sl2, narrow, counts, total_out, nullptr);
}
int oracle_traverse_pair_lvt_write(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t sl1, int64_t sl2,
                                   int32_t narrow, const void *counts, void *contacts) {
    int64_t dummy;
    return pair_lvt(bvh1, bvh2, sl1, sl2, narrow, (void *)counts, &dummy, contacts);
}

static int rays_lvt(const ibvh_bvh *bvh, const void *points, const void *dirs, int64_t num_rays, int64_t sl,
                    void *counts, int64_t *total_out, void *contacts, int threads = 1, TestCounts *tc_out = nullptr) {
    if (int e = check_levels(*bvh, sl, true)) return e;
    // isintersection(::BBox{T}, ::NTuple{3,T}, ...) needs one float type (isintersection.jl:1-5)
    if (bvh->types.leaf_float != bvh->types.node_float) return IBVH_ERR_UNSUPPORTED;
    if (num_rays == 0) return IBVH_OK;
    return dispatch_all(bvh->types, [&](auto lt, auto nt, auto it, auto mt) -> int {
        using L = typename decltype(lt)::type;
        using N = typename decltype(nt)::type;
        using I = typename decltype(it)::type;
        using M = typename decltype(mt)::type;
        using T = typename L::elt;
        if constexpr (!std::is_same<T, typename N::elt>::value) return IBVH_ERR_UNSUPPORTED;
        else {
            auto v = view_of<L, N, I, M>(*bvh);
            const T *p = (const T *)points, *d = (const T *)dirs;
            I *c = (I *)counts;
            if (!contacts) {
                std::mutex mu;
                parallel_chunks(num_rays, threads, 1024, [&](int, int64_t lo, int64_t hi) {
                    TestCounts tc;
                    for (int64_t i = lo + 1; i <= hi; ++i) {
                        int64_t cnt = 0;
                        auto emit = [&](I, I) { ++cnt; };
                        lvt_ray(p + 3 * (i - 1), d + 3 * (i - 1), i, v, sl, emit, tc_out ? &tc : nullptr);
                        c[i - 1] = I(cnt);
                    }
                    if (tc_out) {
                        std::lock_guard<std::mutex> g(mu);
                        tc_out->node += tc.node;
                        tc_out->leaf += tc.leaf;
                    }
                });
                return scan_counts(c, num_rays, total_out);
            }
            IndexPair<I> *out = (IndexPair<I> *)contacts;
            parallel_chunks(num_rays, threads, 1024, [&](int, int64_t lo, int64_t hi) {
                for (int64_t i = lo + 1; i <= hi; ++i) {
                    int64_t w = (i == 1) ? 0 : (int64_t)c[i - 2];
                    auto emit = [&](I a, I b) { out[w++] = {a, b}; };
                    lvt_ray(p + 3 * (i - 1), d + 3 * (i - 1), i, v, sl, emit);
                }
            });
            return IBVH_OK;
        }
    });
}
int oracle_traverse_rays_lvt_count(const ibvh_bvh *bvh, const void *points, const void *dirs, int64_t num_rays,
                                   int64_t sl, void *counts, int64_t *total_out) {
    *total_out = 0;
    return rays_lvt(bvh, points, dirs, num_rays, sl, counts, total_out, nullptr);
}
int oracle_traverse_rays_lvt_write(const ibvh_bvh *bvh, const void *points, const void *dirs, int64_t num_rays,
                                   int64_t sl, const void *counts, void *contacts) {
    int64_t dummy;
    return rays_lvt(bvh, points, dirs, num_rays, sl, (void *)counts, &dummy, contacts);
}

// ---- BFS ---------------------------------------------------------------------------------
int oracle_traverse_bfs(const ibvh_bvh *bvh, int64_t start_level, int32_t narrow, void *bvtt1, void *,
                        int64_t capacity, ibvh_bfs_result *res) {
    if (int e = check_levels(*bvh, start_level, false)) return e;
    *res = {0, 0, 1, 0};
    if (bvh->tree.real_nodes <= 1) return IBVH_OK; // bfs/traverse_single.jl:17-21
    return dispatch_all(bvh->types, [&](auto lt, auto nt, auto it, auto mt) -> int {
        using L = typename decltype(lt)::type;
        using N = typename decltype(nt)::type;
        using I = typename decltype(it)::type;
        using M = typename decltype(mt)::type;
        std::vector<IndexPair<I>> contacts;
        int64_t checks, peak;
        bfs_single(view_of<L, N, I, M>(*bvh), start_level, narrow, contacts, checks, peak);
        return bfs_finish(contacts, checks, peak, bvtt1, capacity, res);
    });
}
int oracle_traverse_pair_bfs(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t sl1, int64_t sl2, int32_t narrow,
                             void *bvtt1, void *, int64_t capacity, ibvh_bfs_result *res) {
    if (int e = check_levels(*bvh1, sl1, false)) return e;
    if (int e = check_levels(*bvh2, sl2, false)) return e;
    if (!same_types(bvh1->types, bvh2->types)) return IBVH_ERR_UNSUPPORTED;
    *res = {0, 0, 1, 0};
    return dispatch_all(bvh1->types, [&](auto lt, auto nt, auto it, auto mt) -> int {
        using L = typename decltype(lt)::type;
        using N = typename decltype(nt)::type;
        using I = typename decltype(it)::type;
        using M = typename decltype(mt)::type;
        std::vector<IndexPair<I>> contacts;
        int64_t checks, peak;
        bfs_pair(view_of<L, N, I, M>(*bvh1), view_of<L, N, I, M>(*bvh2), sl1, sl2, narrow, contacts, checks, peak);
        return bfs_finish(contacts, checks, peak, bvtt1, capacity, res);
    });
}
int oracle_traverse_rays_bfs(const ibvh_bvh *bvh, const void *points, const void *dirs, int64_t num_rays,
                             int64_t sl, void *bvtt1, void *, int64_t capacity, ibvh_bfs_result *res) {
    if (int e = check_levels(*bvh, sl, false)) return e;
    if (bvh->types.leaf_float != bvh->types.node_float) return IBVH_ERR_UNSUPPORTED;
    *res = {0, 0, 1, 0};
    if (num_rays == 0) return IBVH_OK;
    return dispatch_all(bvh->types, [&](auto lt, auto nt, auto it, auto mt) -> int {
        using L = typename decltype(lt)::type;
        using N = typename decltype(nt)::type;
        using I = typename decltype(it)::type;
        using M = typename decltype(mt)::type;
        using T = typename L::elt;
        if constexpr (!std::is_same<T, typename N::elt>::value) return IBVH_ERR_UNSUPPORTED;
        else {
            std::vector<IndexPair<I>> contacts;
            int64_t checks, peak;
            bfs_rays(view_of<L, N, I, M>(*bvh), (const T *)points, (const T *)dirs, num_rays, sl, contacts, checks,
                     peak);
            return bfs_finish(contacts, checks, peak, bvtt1, capacity, res);
        }
    });
}

// ---- brute force (the reference tests' own oracle, runtests.jl:851-859, 1024-1033) --------
// contacts (i, j), i < j (1-based positions in `volumes`), row-major order; returns count.
int64_t oracle_brute_force_self(int kind, int flt, const void *volumes, int64_t n, int64_t *pairs_out,
                                int64_t capacity) {
    int64_t cnt = 0;
    dispatch_volume(kind, flt, [&](auto tv) -> int {
        using V = typename decltype(tv)::type;
        const V *v = (const V *)volumes;
        for (int64_t i = 0; i < n; ++i)
            for (int64_t j = i + 1; j < n; ++j)
                if (iscontact(v[i], v[j])) {
                    if (cnt < capacity) {
                        pairs_out[2 * cnt] = i + 1;
                        pairs_out[2 * cnt + 1] = j + 1;
                    }
                    ++cnt;
                }
        return 0;
    });
    return cnt;
}
int64_t oracle_brute_force_pair(int kind, int flt, const void *va, int64_t na, const void *vb, int64_t nb,
                                int64_t *pairs_out, int64_t capacity) {
    int64_t cnt = 0;
    dispatch_volume(kind, flt, [&](auto tv) -> int {
        using V = typename decltype(tv)::type;
        const V *a = (const V *)va, *b = (const V *)vb;
        for (int64_t i = 0; i < na; ++i)
            for (int64_t j = 0; j < nb; ++j)
                if (iscontact(a[i], b[j])) {
                    if (cnt < capacity) {
                        pairs_out[2 * cnt] = i + 1;
                        pairs_out[2 * cnt + 1] = j + 1;
                    }
                    ++cnt;
                }
        return 0;
    });
    return cnt;
}
int64_t oracle_brute_force_rays(int kind, int flt, const void *volumes, int64_t n, const void *points,
                                const void *dirs, int64_t num_rays, int64_t *pairs_out, int64_t capacity) {
    int64_t cnt = 0;
    dispatch_volume(kind, flt, [&](auto tv) -> int {
        using V = typename decltype(tv)::type;
        using T = typename V::elt;
        const V *v = (const V *)volumes;
        const T *p = (const T *)points, *d = (const T *)dirs;
        for (int64_t r = 0; r < num_rays; ++r)
            for (int64_t i = 0; i < n; ++i)
                if (isintersection(v[i], p + 3 * r, d + 3 * r)) {
                    if (cnt < capacity) {
                        pairs_out[2 * cnt] = i + 1;
                        pairs_out[2 * cnt + 1] = r + 1;
                    }
                    ++cnt;
                }
        return 0;
    });
    return cnt;
}

// ---- batch re-checks for full-size property tests -------------------------------------------
// number of (leaf position, ray) pairs (1-based, as reported) that do NOT satisfy isintersection
int64_t oracle_count_bad_ray_hits(int kind, int flt, const void *volumes, const void *points, const void *dirs,
                                  const int64_t *pairs, int64_t npairs) {
    int64_t bad = 0;
    dispatch_volume(kind, flt, [&](auto tv) -> int {
        using V = typename decltype(tv)::type;
        using T = typename V::elt;
        const V *v = (const V *)volumes;
        const T *p = (const T *)points, *d = (const T *)dirs;
        for (int64_t k = 0; k < npairs; ++k) {
            int64_t i = pairs[2 * k] - 1, r = pairs[2 * k + 1] - 1;
            if (!isintersection(v[i], p + 3 * r, d + 3 * r)) ++bad;
        }
        return 0;
    });
    return bad;
}
// number of (i, j) pairs (1-based) with !iscontact(a[i], b[j])
int64_t oracle_count_bad_contacts(int kind, int flt, const void *va, const void *vb, const int64_t *pairs, int64_t npairs) {
    int64_t bad = 0;
    dispatch_volume(kind, flt, [&](auto tv) -> int {
        using V = typename decltype(tv)::type;
        const V *a = (const V *)va, *b = (const V *)vb;
        for (int64_t k = 0; k < npairs; ++k)
            if (!iscontact(a[pairs[2 * k] - 1], b[pairs[2 * k + 1] - 1])) ++bad;
        return 0;
    });
    return bad;
}

// ---- synthetic inputs (same specification as ibvh_generate_spheres_f32) -------------------
int oracle_generate_spheres_f32(int64_t n, uint64_t seed, int64_t first_index, const float origin[3],
                                const float extent[3], float r0, void *out) {
    BSphere<float> *s = (BSphere<float> *)out;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t g = (uint64_t)(first_index + i);
        float u0 = u01(seed, 4 * g + 0), u1 = u01(seed, 4 * g + 1), u2 = u01(seed, 4 * g + 2), u3 = u01(seed, 4 * g + 3);
        s[i].x[0] = origin[0] + extent[0] * u0;
        s[i].x[1] = origin[1] + extent[1] * u1;
        s[i].x[2] = origin[2] + extent[2] * u2;
        s[i].r = r0 * (0.5f + 0.5f * u3);
    }
    return IBVH_OK;
}

// ---- multithreaded timed baseline ("CPU restatement (T threads)") ---------
// Same arithmetic and output as oracle_build + oracle_traverse_lvt_*; parallel over contiguous
// ranges the way the reference's CPU path is (AK.itask_partition, lvt/traverse_single.jl:94-111):
// extrema, encode, chunked stable sort + stable merges, per-level merges, two-pass LVT.
} // extern "C"
extern "C" {

// SphereF32 leaves / BBoxF32 nodes / I32 / U32 — the bench types (benchmark/bvh_contact.jl:21-27)
int oracle_bench_build_traverse_f32(const void *volumes, int64_t n, int threads, void *leaves_out, void *nodes_out,
                                    void *skips_out, void *counts, void *contacts_out, int64_t contacts_capacity,
                                    int64_t *num_contacts, double *t_build_s, double *t_traverse_s) {
    using L = BSphere<float>;
    using N = BBox<float>;
    using I = int32_t;
    using M = uint32_t;
    using Rec = BoundingVolume<L, I, M>;
    ibvh_tree tree;
    if (!tree_shape(n, tree)) return IBVH_ERR_DOMAIN;
    auto t0 = std::chrono::steady_clock::now();
    Rec *leaves = (Rec *)leaves_out;
    const L *vols = (const L *)volumes;
    parallel_ranges(n, threads, [&](int, int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) leaves[i] = Rec{vols[i], I(i + 1), 0};
    });
    std::vector<int64_t> sk(tree.levels);
    compute_skips(tree, sk.data());
    for (int64_t i = 0; i < tree.levels; ++i) ((I *)skips_out)[i] = I(sk[i]);
    // extrema
    std::vector<float> part(6 * std::max(threads, 1));
    int used = (threads <= 1 || n < 2 * threads) ? 1 : threads;
    parallel_ranges(n, threads, [&](int t, int64_t lo, int64_t hi) { extrema_of(leaves + lo, hi - lo, false, &part[6 * t]); });
    float ext[6];
    for (int k = 0; k < 6; ++k) ext[k] = part[k];
    for (int t = 1; t < used; ++t)
        for (int k = 0; k < 3; ++k) {
            ext[k] = ext[k] < part[6 * t + k] ? ext[k] : part[6 * t + k];
            ext[3 + k] = ext[3 + k] > part[6 * t + 3 + k] ? ext[3 + k] : part[6 * t + 3 + k];
        }
    {
        const float rp = relative_precision<float>(), fm = std::numeric_limits<float>::min();
        for (int k = 0; k < 3; ++k) {
            float a = rp * std::fabs(ext[k]);
            ext[k] = (ext[k] - a) - fm;
            float b = rp * std::fabs(ext[3 + k]);
            ext[3 + k] = (ext[3 + k] + b) + fm;
        }
    }
    parallel_ranges(n, threads, [&](int, int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) leaves[i].morton = morton_encode_single<M>(leaves[i].volume.x, ext, ext + 3);
    });
    // stable sort by Morton code: parallel LSB radix sort of (code, position) pairs, 8-bit digits, then one gather of
    // the records (same result as a stable comparison sort of the records; per-thread histograms over contiguous
    // chunks keep it stable)
    if (used == 1) {
        auto cmp = [](const Rec &a, const Rec &b) { return a.morton < b.morton; };
        std::stable_sort(leaves, leaves + n, cmp);
    } else {
        std::vector<uint32_t> ka(n), kb(n), va(n), vb(n);
        parallel_ranges(n, threads, [&](int, int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) {
                ka[i] = leaves[i].morton;
                va[i] = (uint32_t)i;
            }
        });
        std::vector<int64_t> hist((size_t)used * 256), offs((size_t)used * 256);
        uint32_t *kin = ka.data(), *kout = kb.data(), *vin = va.data(), *vout = vb.data();
        for (int shift = 0; shift < 30; shift += 8) {
            parallel_ranges(n, threads, [&](int t, int64_t lo, int64_t hi) {
                int64_t *h = &hist[(size_t)t * 256];
                for (int d = 0; d < 256; ++d) h[d] = 0;
                for (int64_t i = lo; i < hi; ++i) ++h[(kin[i] >> shift) & 255u];
            });
            int64_t run = 0;
            for (int d = 0; d < 256; ++d)
                for (int t = 0; t < used; ++t) {
                    offs[(size_t)t * 256 + d] = run;
                    run += hist[(size_t)t * 256 + d];
                }
            parallel_ranges(n, threads, [&](int t, int64_t lo, int64_t hi) {
                int64_t *o = &offs[(size_t)t * 256];
                for (int64_t i = lo; i < hi; ++i) {
                    const int64_t dst = o[(kin[i] >> shift) & 255u]++;
                    kout[dst] = kin[i];
                    vout[dst] = vin[i];
                }
            });
            std::swap(kin, kout);
            std::swap(vin, vout);
        }
        parallel_ranges(n, threads, [&](int, int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) leaves[i] = Rec{vols[vin[i]], I(vin[i] + 1), kin[i]};
        });
    }
    // merges, level by level
    N *nodes = (N *)nodes_out;
    if (tree.real_nodes >= 2) {
        {
            int64_t level = tree.levels - 1;
            int64_t start_pos = memory_index(tree, pow2(level - 1));
            int64_t num_nodes = pow2(level - 1) - jl_shr(tree.virtual_leaves, 1);
            parallel_ranges(num_nodes, threads, [&](int, int64_t lo, int64_t hi) {
                for (int64_t i = lo + 1; i <= hi; ++i) {
                    int64_t l = 2 * i - 1, r = 2 * i;
                    nodes[start_pos - 1 + i - 1] = r > n ? convert_to(leaves[l - 1].volume, (N *)nullptr)
                                                         : merge_to(leaves[l - 1].volume, leaves[r - 1].volume, (N *)nullptr);
                }
            });
        }
        for (int64_t level = tree.levels - 2; level >= 1; --level) {
            int64_t start_pos = memory_index(tree, pow2(level - 1));
            int64_t num_nodes = pow2(level - 1) - jl_shr(tree.virtual_leaves, tree.levels - level);
            int64_t spn = memory_index(tree, pow2(level));
            int64_t nnn = pow2(level) - jl_shr(tree.virtual_leaves, tree.levels - (level + 1));
            parallel_ranges(num_nodes, threads, [&](int, int64_t lo, int64_t hi) {
                for (int64_t i = lo + 1; i <= hi; ++i) {
                    int64_t l = spn + 2 * i - 2, r = spn + 2 * i - 1;
                    nodes[start_pos - 1 + i - 1] =
                        r > spn + nnn - 1 ? nodes[l - 1] : merge_to(nodes[l - 1], nodes[r - 1], (N *)nullptr);
                }
            });
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    // LVT two-pass
    View<L, N, I, M> v{tree, leaves, nodes, (const I *)skips_out};
    I *c = (I *)counts;
    int64_t total = 0;
    if (tree.real_nodes > 1) {
        parallel_chunks(n, threads, 2048, [&](int, int64_t lo, int64_t hi) {
            for (int64_t i = lo + 1; i <= hi; ++i) {
                int64_t cnt = 0;
                auto emit = [&](I, I) { ++cnt; };
                lvt_single_leaf(v, i, 1, 0, emit);
                c[i - 1] = I(cnt);
            }
        });
        if (int e = scan_counts(c, n, &total)) return e;
        if (total <= contacts_capacity) {
            IndexPair<I> *out = (IndexPair<I> *)contacts_out;
            parallel_chunks(n, threads, 2048, [&](int, int64_t lo, int64_t hi) {
                for (int64_t i = lo + 1; i <= hi; ++i) {
                    int64_t w = (i == 1) ? 0 : (int64_t)c[i - 2];
                    auto emit = [&](I a, I b) { out[w++] = {a, b}; };
                    lvt_single_leaf(v, i, 1, 0, emit);
                }
            });
        }
    }
    auto t2 = std::chrono::steady_clock::now();
    *num_contacts = total;
    *t_build_s = std::chrono::duration<double>(t1 - t0).count();
    *t_traverse_s = std::chrono::duration<double>(t2 - t1).count();
    return total <= contacts_capacity ? IBVH_OK : IBVH_ERR_CAPACITY;
}

// ---- timed CPU baselines of the pair and ray traversals (bench.py's cpu_baseline legs for configs 3 and 4) ----------
// The reference's two-pass protocol on `threads` OpenMP threads with dynamically handed-out chunks
// (lvt/traverse_pair.jl:40-116, raytrace/leaf_vs_tree/leaf_vs_tree.jl:1-90): count -> inclusive scan -> write.
// Protocol of benchmark/bvh_contact_pair.jl:38-46 / bvh_rays.jl:36-58: the BVHs are built beforehand.
int oracle_bench_pair_lvt(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t sl1, int64_t sl2, int threads, void *counts,
                          void *contacts, int64_t capacity, int64_t *num_contacts, double *t_s) {
    auto t0 = std::chrono::steady_clock::now();
    int64_t total = 0;
    if (int e = pair_lvt(bvh1, bvh2, sl1, sl2, 0, counts, &total, nullptr, threads)) return e;
    *num_contacts = total;
    if (total > capacity) return IBVH_ERR_CAPACITY;
    if (int e = pair_lvt(bvh1, bvh2, sl1, sl2, 0, counts, &total, contacts, threads)) return e;
    *t_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return IBVH_OK;
}
int oracle_bench_rays_lvt(const ibvh_bvh *bvh, const void *points, const void *dirs, int64_t num_rays, int64_t sl, int threads,
                          void *counts, void *contacts, int64_t capacity, int64_t *num_contacts, double *t_s) {
    auto t0 = std::chrono::steady_clock::now();
    int64_t total = 0;
    if (int e = rays_lvt(bvh, points, dirs, num_rays, sl, counts, &total, nullptr, threads)) return e;
    *num_contacts = total;
    if (total > capacity) return IBVH_ERR_CAPACITY;
    if (int e = rays_lvt(bvh, points, dirs, num_rays, sl, counts, &total, contacts, threads)) return e;
    *t_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return IBVH_OK;
}

// ---- work counters of the reference's leaf-vs-tree walk: node tests and leaf tests of ONE pass ------
// bvh2 != NULL: pair walk (the BVH with more leaves drives); points != NULL: rays; otherwise the self walk.
// counts: scratch for the per-item counts (max(n1, n2) / num_rays / n entries of the index type).
int oracle_lvt_test_counts(const ibvh_bvh *bvh, const ibvh_bvh *bvh2, const void *points, const void *dirs, int64_t num_rays,
                           int64_t sl1, int64_t sl2, int threads, void *counts, int64_t *node_tests, int64_t *leaf_tests,
                           int64_t *num_contacts) {
    TestCounts tc;
    int64_t total = 0;
    int rc = IBVH_OK;
    if (points) {
        rc = rays_lvt(bvh, points, dirs, num_rays, sl1, counts, &total, nullptr, threads, &tc);
    } else if (bvh2) {
        rc = pair_lvt(bvh, bvh2, sl1, sl2, 0, counts, &total, nullptr, threads, &tc);
    } else {
        if (int e = check_levels(*bvh, sl1, true)) return e;
        if (bvh->tree.real_nodes > 1)
            rc = dispatch_all(bvh->types, [&](auto lt, auto nt, auto it, auto mt) -> int {
                using L = typename decltype(lt)::type;
                using N = typename decltype(nt)::type;
                using I = typename decltype(it)::type;
                using M = typename decltype(mt)::type;
                auto v = view_of<L, N, I, M>(*bvh);
                std::mutex mu;
                parallel_chunks(bvh->tree.real_leaves, threads, 2048, [&](int, int64_t lo, int64_t hi) {
                    TestCounts t;
                    int64_t cnt = 0;
                    auto emit = [&](I, I) { ++cnt; };
                    for (int64_t i = lo + 1; i <= hi; ++i) lvt_single_leaf(v, i, sl1, 0, emit, &t);
                    std::lock_guard<std::mutex> g(mu);
                    tc.node += t.node;
                    tc.leaf += t.leaf;
                    total += cnt;
                });
                return IBVH_OK;
            });
    }
    *node_tests = tc.node;
    *leaf_tests = tc.leaf;
    *num_contacts = total;
    return rc;
}

int oracle_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

} // extern "C"
