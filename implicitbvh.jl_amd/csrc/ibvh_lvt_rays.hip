// ibvh_lvt_rays.hip — walker 3 of the leaf-vs-tree traversal: lvt_rays_kernel, the per-lane ray walk
// (raytrace/leaf_vs_tree/leaf_vs_tree.jl:187-225), and the launcher of a whole ray pass.
#include "ibvh_lvt.hpp"

namespace ibvh {
namespace lvt {

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

#ifdef IBVH_VARIANTS
#include "../../variants/rays_shadow.inc"
#endif

// Bytes of the quantised shadow a ray traversal of `bvh` with `num_rays` rays uses, 0 when the binary walk serves it — always,
// in the product library: the shadow walker is a development variant (variants/rays_shadow.inc, knob "rays_shadow").
size_t rays_shadow_bytes(const ibvh_bvh &bvh, int64_t num_rays) {
#ifdef IBVH_VARIANTS
    // single-precision leaves under BBox{Float32} nodes, a fully built tree of 8 .. 26 levels, and enough rays for the one
    // streaming pass over the nodes that builds the shadow to pay (at least one ray per 64 leaves)
    if (bvh.types.node_kind != IBVH_BBOX || bvh.types.node_float != IBVH_F32 || bvh.types.leaf_float != IBVH_F32) return 0;
    if (!g_tuning.rays_shadow) return 0;
    if (bvh.built_level > 1 || num_rays * 64 < bvh.tree.real_leaves) return 0;
    const RayShadow sh = make_ray_shadow(bvh.tree);
    return sh.depths ? (size_t)sh.base[sh.depths] * SHADOW_ENTRY_BYTES : 0;
#else
    (void)bvh, (void)num_rays;
    return 0;
#endif
}

template <class L, class N, class I>
int launch_rays_standby(const Args<L, N, I> &standby, bool write, hipStream_t st, int ray_block, unsigned rblocks) {
    const PairCache<I> none{nullptr, 0};
    if (write) IBVH_LAUNCH((lvt_rays_kernel<L, N, I, true>), dim3(rblocks), dim3(64), 0, st, standby, none, ray_block);
    else IBVH_LAUNCH((lvt_rays_kernel<L, N, I, false>), dim3(rblocks), dim3(64), 0, st, standby, none, ray_block);
    return IBVH_OK;
}

template <class L, class N, class I>
int launch_rays(const Args<L, N, I> &a, const PairCache<I> &cache, bool write, hipStream_t st, const RayBins &rb) {
    const bool count_work = a.work != nullptr;
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
        if (rb.cap > 0 && !count_work) return launch_rays_binned<L, N, I>(a, write, st, rb, ray_block, rblocks); // (3c), ibvh_lvt_raybins.hip
    }
#ifdef IBVH_VARIANTS
    if constexpr (std::is_same<typename L::elt, float>::value && std::is_same<N, BBox<float>>::value) {
        if (a.shadow != nullptr && !count_work) {
            // regular rays over the 8-wide shadow; the irregular ones (if any) by the binary walker behind it, without
            // a cache of its own (the block headers belong to the shadow walker)
            const ibvh_tree t{a.tree.levels, a.tree.real_leaves, 0, a.tree.virtual_leaves, 0};
            const RayShadow sh = make_ray_shadow(t);
            if (!write) // (the writing pass of a _count / _write pair finds the shadow where the count left it)
                IBVH_LAUNCH((ray_shadow_build_kernel<N>), dim3((unsigned)ceil_div((int64_t)sh.base[sh.depths], 256)), dim3(256), 0, st, a.nodes,
                            a.tree, sh, (ShadowEntry *)a.shadow);
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
#endif
    if constexpr (kWorkTypes<L, N, I>) {
        if (count_work) {
            IBVH_LAUNCH((lvt_rays_kernel<L, N, I, false, true>), dim3(rblocks), dim3(64), 0, st, a, cache, ray_block);
            IBVH_LAUNCH_CHECK();
            return IBVH_OK;
        }
    }
    if (write) IBVH_LAUNCH((lvt_rays_kernel<L, N, I, true>), dim3(rblocks), dim3(64), 0, st, a, cache, ray_block);
    else IBVH_LAUNCH((lvt_rays_kernel<L, N, I, false>), dim3(rblocks), dim3(64), 0, st, a, cache, ray_block);
    IBVH_LAUNCH_CHECK();
    return IBVH_OK;
}

#define IBVH_INSTANTIATE_RAYS(L_, N_, I_, ...)                                                                                  \
    template int launch_rays<L_, N_, I_>(const Args<L_, N_, I_> &, const PairCache<I_> &, bool, hipStream_t, const RayBins &); \
    template int launch_rays_standby<L_, N_, I_>(const Args<L_, N_, I_> &, bool, hipStream_t, int, unsigned);
IBVH_FOR_SAME_FLOAT_COMBOS(IBVH_INSTANTIATE_RAYS, 0)

} // namespace lvt
} // namespace ibvh
