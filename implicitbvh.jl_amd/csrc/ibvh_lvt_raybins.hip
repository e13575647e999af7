// ibvh_lvt_raybins.hip — walker 4 of the leaf-vs-tree traversal: rays binned by subtree (the default ray path of large
// trees).  raytrace/leaf_vs_tree/leaf_vs_tree.jl:170-228 cut in two at a level K.
#include "ibvh_lvt.hpp"

namespace ibvh {
namespace lvt {

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
    { // the tile aggregates of the scan over the items' hits (scan_counts, one-kernel route): the scan over the rays' items has used them
        unsigned long long *agg = (unsigned long long *)rb.scan_scratch + 8;
        const int words = (rb.cap + SCAN_TILE - 1) / SCAN_TILE + 1;
        for (int k = (int)threadIdx.x; k < words; k += 1024) agg[k] = 0ull;
    }
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

// Room for `count` hit records of this wave in the hit list: the list `region` points at, or — when that one is full — one of the
// next few (what was left of a full list is marked unused; a wave moves on to another list after every reservation, so the lists
// fill evenly whatever a subtree's share of the hits).  false: no room found, the writing pass walks the subtrees again (*reflag).
template <class I> IBVH_D bool reserve_hits(const RayBins &rb, uint32_t &region, uint32_t count, int lane, RayHit<I> *&dst) {
    const uint32_t mask = (uint32_t)(rb.regions - 1);
    const int tries = rb.regions < 8 ? rb.regions : 8;
    uint32_t r = region;
    for (int attempt = 0; attempt < tries; ++attempt) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&rb.region_cursor[r], count);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        RayHit<I> *list = (RayHit<I> *)rb.hit_list + (size_t)r * rb.region_cap;
        r = (r + 61u) & mask;
        if ((uint64_t)base + count <= (uint64_t)rb.region_cap) {
            dst = list + base;
            region = r;
            return true;
        }
        for (uint64_t t = (uint64_t)base + (uint32_t)lane; t < (uint64_t)rb.region_cap; t += 64) list[t].g = RAY_HIT_NONE;
    }
    region = r;
    if (lane == 0) *rb.reflag = 1;
    return false;
}

#ifdef IBVH_RAYSUB_HIST // (diagnostic build, tools/dbg_raysub_hist.py: wave-steps of the counting pass by number of busy lanes; [8], [9]: steps
                        // after the wave's chunk ran dry with < 8 / >= 8 lanes busy)
__device__ unsigned long long g_raysub_hist[16];
#endif
// ---- the tail of a rays_subtree_kernel workgroup (counting pass, round 6) -----------------------------------------------------
// Walks still alive when their wave's chunk has run dry and <= tail_lanes lanes are busy are PARKED (<= RAYSUB_TAIL_MAX a wave) and
// finished by the WHOLE workgroup as units (item, node whose two children are still to be tested): the walk's current node and
// every pending right sibling are independent subtrees, so a long walk is no longer one chain of dependent steps that three
// quarters of the wave-steps of the pass waited for with < 8 lanes busy (profiles/r05_variants.txt).  Rounds: every unit goes down
// one level, children that hit become the next round's units (two lists in LDS, <= D rounds); a hit leaf sets its bit in the
// item's mask (one bit per leaf of the subtree: pre-order = leaf order); when the rounds are through, an item's hits are its
// mask's bits in order, with ranks that go on from the hits the walk had made before it was parked.  A unit that finds the next
// list full is finished depth-first by its lane on the spot.  All of it lives in the four waves' flushed hit stages (4 KB): two
// unit lists, the masks, the parked items' (ray, g, hits so far).  A function of its own on purpose: inlined behind the walking
// loop its address arithmetic and uniform values were kept live across that loop (68 -> 106 SGPRs, the hot loop +12 %).
template <class L, class N, class I>
IBVH_D void rays_tail_phase(const Args<L, N, I> &a, const RayBins &rb, unsigned char *stage, const N *s_nodes, const L *s_leaves,
                                                     const I *s_index, uint32_t *s_npark, uint32_t *s_nunits, bool busy, uint32_t ray, uint32_t g,
                                                     uint32_t tn, int dl, uint32_t pend, uint32_t cnt, uint32_t j, uint32_t region) {
    using T = typename L::elt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int levels = (int)a.tree.levels, K = rb.cut_level, D = rb.depth;
    const uint32_t S = 1u << D;
    const uint32_t vl = (uint32_t)a.tree.virtual_leaves, real_leaves = (uint32_t)a.tree.real_leaves;
    constexpr int UNITS = RAYSUB_TAIL_UNITS;
    uint32_t *units = (uint32_t *)stage;                     // [2][UNITS]: item | depth << 6 | node << 10 | pending siblings << 20 (a parked walk's first unit only)
    uint32_t *masks = units + 2 * UNITS;                     // [RAYSUB_TAIL_ITEMS][S / 32]
    const int mask_words = (int)(S >> 5) > 0 ? (int)(S >> 5) : 1;
    uint32_t *p_ray = masks + RAYSUB_TAIL_ITEMS * 16, *p_g = p_ray + RAYSUB_TAIL_ITEMS, *p_cnt = p_g + RAYSUB_TAIL_ITEMS;
    static_assert((2 * RAYSUB_TAIL_UNITS + RAYSUB_TAIL_ITEMS * 19) * 4 <= RAYSUB_WALKERS * RAYSUB_STAGE * 16, "the tail lives in the hit stages");
    // one unit down one level; returns the children that hit (node tests) or marks the leaves that hit (cd == D)
    auto expand = [&](uint32_t pid, uint32_t t, int cd, const T (&up)[3], const T (&ud)[3], const T (&uinv)[3], bool &n0, bool &n1) {
        const uint32_t c0 = 2u * t, c1 = c0 + 1u;
        n0 = n1 = false;
        if (cd == D) { // the two leaves under t
            const uint32_t li = c0 - S;
            const bool real1 = (j << D) + li + 1u < real_leaves;
            const L la = s_leaves[li], lb = s_leaves[li + 1];
            bool h0 = isintersection(la, up, ud), h1 = real1 && isintersection(lb, up, ud);
            if (a.narrow == IBVH_NARROW_RAY_ORIGIN_OUTSIDE) {
                h0 = h0 && origin_outside(la, up);
                h1 = h1 && origin_outside(lb, up);
            }
            const uint32_t bits = (h0 ? 1u << (li & 31u) : 0u) | (h1 ? 1u << ((li + 1u) & 31u) : 0u); // (li is even: one word)
            if (bits != 0) atomicOr(&masks[pid * mask_words + (li >> 5)], bits);
        } else {
            const int level = K + cd;
            const uint32_t nreal = (1u << (level - 1)) - (uint32_t)((uint64_t)vl >> (levels - level));
            const bool real1 = (j << cd) + (c1 - (1u << cd)) < nreal;
            const N na = s_nodes[c0], nb = s_nodes[c1];
            if constexpr (N::kind == IBVH_BBOX) {
                n0 = isintersection_inv(na, up, uinv);
                n1 = real1 && isintersection_inv(nb, up, uinv);
            } else {
                n0 = isintersection(na, up, ud);
                n1 = real1 && isintersection(nb, up, ud);
            }
        }
    };
    __syncthreads(); // every wave has left the walking loop and flushed: the stages are free
    if (busy) {
        const uint32_t pid = atomicAdd(s_npark, 1u);
        p_ray[pid] = ray;
        p_g[pid] = g;
        p_cnt[pid] = cnt;
        for (int k = 0; k < mask_words; ++k) masks[pid * mask_words + k] = 0u;
        // ONE unit per parked walk: it stands at tn (its children are next) and carries the walk's pending right siblings, which the
        // lane that takes the unit turns into units of their own
        const uint32_t slot = atomicAdd(&s_nunits[0], 1u); // (< RAYSUB_TAIL_ITEMS <= UNITS)
        units[slot] = pid | ((uint32_t)dl << 6) | (tn << 10) | (pend << 20);
    }
    __syncthreads();
    int cur = 0;
    for (int round = 0; round <= D; ++round) {
        uint32_t n_units = s_nunits[cur];
        if (n_units == 0) break;
        n_units = n_units < (uint32_t)UNITS ? n_units : (uint32_t)UNITS;
        __syncthreads(); // (everybody has read the count)
        if (tid == 0) s_nunits[cur ^ 1] = 0;
        __syncthreads();
        const uint32_t *in = units + cur * UNITS;
        uint32_t *out = units + (cur ^ 1) * UNITS;
        for (uint32_t u = tid; u < n_units; u += RAYSUB_TPB) {
            const uint32_t unit = in[u];
            const uint32_t pid = unit & 63u;
            const int64_t r = (int64_t)p_ray[pid];
            T up[3], ud[3], uinv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                up[k] = a.points[3 * r + k];
                ud[k] = a.dirs[3 * r + k];
                uinv[k] = T(1) / ud[k];
            }
            // this unit one level down; a child that finds the next list full is walked depth-first, here and now
            uint32_t t = (unit >> 10) & 1023u, spill_pend = 0;
            int d = (int)((unit >> 6) & 15u);
            for (uint32_t up_pend = unit >> 20; up_pend != 0;) { // a parked walk's pending right siblings: units of their own
                const int pl = 31 - __builtin_clz(up_pend);
                up_pend &= ~(1u << pl);
                const uint32_t slot = atomicAdd(&s_nunits[cur ^ 1], 1u);
                if (slot < (uint32_t)UNITS) out[slot] = pid | ((uint32_t)pl << 6) | (((t >> (d - pl)) | 1u) << 10);
                else spill_pend |= 1u << pl; // (no room: this lane's depth-first walk below takes it)
            }
            bool first = true;
            for (;;) {
                bool n0, n1;
                expand(pid, t, d + 1, up, ud, uinv, n0, n1);
                bool descended = false;
                if (first) {
                    first = false;
                    uint32_t keep = 0; // children that did not fit the next list
                    if (n0) {
                        const uint32_t slot = atomicAdd(&s_nunits[cur ^ 1], 1u);
                        if (slot < (uint32_t)UNITS) out[slot] = pid | ((uint32_t)(d + 1) << 6) | ((2u * t) << 10);
                        else keep |= 1u;
                    }
                    if (n1) {
                        const uint32_t slot = atomicAdd(&s_nunits[cur ^ 1], 1u);
                        if (slot < (uint32_t)UNITS) out[slot] = pid | ((uint32_t)(d + 1) << 6) | ((2u * t + 1u) << 10);
                        else keep |= 2u;
                    }
                    n0 = (keep & 1u) != 0;
                    n1 = (keep & 2u) != 0;
                }
                if (n0) { // (the walking loop's step, hits marked instead of staged)
                    if (n1) spill_pend |= 1u << (d + 1);
                    t = 2u * t;
                    d += 1;
                    descended = true;
                } else if (n1) {
                    t = 2u * t + 1u;
                    d += 1;
                    descended = true;
                }
                if (!descended) {
                    if (spill_pend == 0) break;
                    const int pl = 31 - __builtin_clz(spill_pend);
                    spill_pend &= ~(1u << pl);
                    t = (t >> (d - pl)) | 1u;
                    d = pl;
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    __syncthreads();
    // ranked records: item pid's hits are its mask's bits in leaf order, ranks go on from the hits its walk had made
    const uint32_t n_park = *s_npark;
    if (wave == 0) {
        uint32_t total = 0;
        if ((uint32_t)lane < n_park)
            for (int k = 0; k < mask_words; ++k) total += (uint32_t)__popc(masks[lane * mask_words + k]);
        uint32_t inc = total;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, off, 64);
            if (lane >= off) inc += v;
        }
        const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        if ((uint32_t)lane < n_park) ((I *)rb.hits)[p_g[lane]] = (I)(p_cnt[lane] + total);
        if (all != 0) {
            RayHit<I> *room = nullptr;
            if (reserve_hits<I>(rb, region, all, lane, room) && (uint32_t)lane < n_park && total != 0) {
                RayHit<I> *dst = room + (inc - total);
                uint32_t k = p_cnt[lane];
                const I ray1 = (I)((int64_t)p_ray[lane] + 1);
                for (int wd = 0; wd < mask_words; ++wd) {
                    uint32_t m = masks[lane * mask_words + wd];
                    while (m != 0) {
                        const int b = __builtin_ctz(m);
                        m &= m - 1u;
                        *dst++ = RayHit<I>{IndexPair<I>{s_index[wd * 32 + b], ray1}, p_g[lane], k++};
                    }
                }
            }
        }
    }
}

template <class L, class N, class I, bool WRITE>
__global__ __launch_bounds__(RAYSUB_TPB) void rays_subtree_kernel(Args<L, N, I> a, RayBins rb) {
    using T = typename L::elt;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ uint32_t s_next;
    // the workgroup's tail (counting pass; RAYSUB_TAIL_*): parked walks and the counters of the unit rounds
    __shared__ uint32_t s_npark, s_nunits[2];
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
    if (tid == 0) {
        s_next = 0;
        s_npark = 0;
        s_nunits[0] = s_nunits[1] = 0;
    }
    __syncthreads();
    static_assert(RAYSUB_WALKERS * 64 == RAYSUB_TPB, "every wave walks: the tail's barriers count on all of them");

    const I *hits = (const I *)rb.hits;
    // (the list this wave appends to next — the records carry their place, any list serves: reserve_hits)
    uint32_t region = blockIdx.x & (uint32_t)(rb.regions - 1);
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
        RayHit<I> *dst = nullptr;
        if (reserve_hits<I>(rb, region, (uint32_t)fill, lane, dst))
            for (int t = lane; t < fill; t += 64) dst[t] = s_stage[t];
        fill = 0;
    };
    // tail_lanes > 0 (counting pass): the loop is left with <= tail_lanes walks alive once the chunk is dry; they are parked below
    const int tail_lanes = WRITE ? 0 : rb.tail_lanes;
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
#ifdef IBVH_RAYSUB_HIST
            if constexpr (!WRITE) {
                if (lane == 0) {
                    const int nb = 64 - __popcll(idle_now);
                    atomicAdd(&g_raysub_hist[nb >= 64 ? 7 : nb / 8], 1ull);
                    if (!more) atomicAdd(&g_raysub_hist[8 + (nb >= 8 ? 1 : 0)], 1ull);
                }
            }
#endif
            if (idle_now == ~(uint64_t)0) break;
            if (more && __popcll(idle_now) >= 16) break;
            if (!more && 64 - __popcll(idle_now) <= tail_lanes) break;
        }
        if (!more && tail_lanes > 0 && 64 - __popcll(__builtin_amdgcn_ballot_w64(!busy)) <= tail_lanes) break;
    }
    if constexpr (!WRITE) {
        flush();
        if (tail_lanes > 0) rays_tail_phase<L, N, I>(a, rb, s_raw + o, s_nodes, s_leaves, s_index, &s_npark, s_nunits, busy, ray, g, tn, dl, pend, cnt, j, region);
    }
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
        if (r.g == RAY_HIT_NONE) continue; // (the unused end of a list that filled up: reserve_hits)
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

template <class L, class N, class I>
int launch_rays_binned(const Args<L, N, I> &a, bool write, hipStream_t st, const RayBins &rb, int ray_block, unsigned rblocks) {
    // the binned path (3c); the binary walker stands by behind it, gated on the overflow flag
    const size_t lds = rays_subtree_lds(rb.depth, sizeof(N), sizeof(L), sizeof(I), sizeof(RayHit<I>), write);
    Args<L, N, I> standby = a;
    standby.gate = rb.flag;
    standby.shadow = nullptr;
    const PairCache<I> none{nullptr, 0};
    if (!write) {
        // (header + the helper scans' header and tile aggregates right behind it: both scans take the one-kernel route, the second
        // one's aggregates are zeroed again by rays_binscan_kernel)
        const int64_t scan_words = 8 + ceil_div((int64_t)rb.cap > a.n_items ? (int64_t)rb.cap : a.n_items, (int64_t)SCAN_TILE) + 1;
        IBVH_HIP_CHECK(hipMemsetAsync(rb.cursor, 0, 2048 + (size_t)scan_words * 8, st));
        IBVH_HIP_CHECK(hipMemsetAsync(rb.bin_count, 0, (size_t)((char *)rb.items - (char *)rb.bin_count), st)); // counts, starts, cursors
        if (!g_tuning.rays_fast_slab) IBVH_HIP_CHECK(hipMemsetAsync(rb.top_nan, 0xff, 4, st)); // (-1: no fast slab test anywhere)
        if constexpr (N::kind == IBVH_BBOX && std::is_same<typename N::elt, float>::value) {
            const int64_t top_first = level_start(a.tree.levels, a.tree.virtual_leaves, a.built_level) - 1; // (memory index of the first node that exists)
            const int64_t top_count = level_start(a.tree.levels, a.tree.virtual_leaves, rb.cut_level + 1) - 1 - top_first;
            IBVH_LAUNCH((rays_topcheck_kernel<N>), dim3((unsigned)(ceil_div(top_count * 3, 256) < 256 ? ceil_div(top_count * 3, 256) : 256)), dim3(256), 0,
                        st, a.nodes + top_first, top_count, rb);
        }
        IBVH_LAUNCH((rays_top_kernel<L, N, I>), dim3(rblocks), dim3(64), 0, st, a, rb, ray_block);
        if (int e = scan_counts<int32_t>(rb.ray_items, a.n_items, nullptr, rb.scan_scratch, st, rb.dummy_total, nullptr, nullptr, true)) return e;
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
        if (int e = scan_counts<I>((I *)rb.hits, (int64_t)rb.cap, nullptr, rb.scan_scratch, st, rb.dummy_total, nullptr, rb.n_items, true)) return e;
        IBVH_LAUNCH((rays_counts_kernel<I>), dim3((unsigned)ceil_div(a.n_items, 256)), dim3(256), 0, st, rb, a.counts, a.n_items);
        if (int e = launch_rays_standby<L, N, I>(standby, false, st, ray_block, rblocks)) return e;
    } else {
        IBVH_LAUNCH((rays_place_kernel<I>), dim3((unsigned)ceil_div((int64_t)rb.region_cap, 1024), (unsigned)rb.regions), dim3(256), 0, st, rb, a.contacts,
                    a.guard_total, a.guard_capacity);
        if (lds > 64 * 1024)
            IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)rays_subtree_kernel<L, N, I, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        IBVH_LAUNCH((rays_subtree_kernel<L, N, I, true>), dim3((unsigned)(rb.subtrees + rb.cap / RAYSUB_CHUNK)), dim3(RAYSUB_TPB), lds, st, a, rb);
        if (int e = launch_rays_standby<L, N, I>(standby, true, st, ray_block, rblocks)) return e;
    }
    IBVH_LAUNCH_CHECK();
    return IBVH_OK;
}

#define IBVH_INSTANTIATE_RAYBINS(L_, N_, I_, ...) \
    template int launch_rays_binned<L_, N_, I_>(const Args<L_, N_, I_> &, bool, hipStream_t, const RayBins &, int, unsigned);
IBVH_FOR_SAME_FLOAT_COMBOS(IBVH_INSTANTIATE_RAYBINS, 0)

} // namespace lvt
} // namespace ibvh

#ifdef IBVH_RAYSUB_HIST
extern "C" int ibvh_debug_raysub_hist(unsigned long long *out /* 16 */, int reset) {
    if (reset) {
        unsigned long long z[16] = {};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(ibvh::lvt::g_raysub_hist), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ibvh::lvt::g_raysub_hist), sizeof(unsigned long long) * 16);
}
#endif
