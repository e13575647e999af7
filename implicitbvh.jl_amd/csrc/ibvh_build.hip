// ibvh_build.hip — BVH construction on gfx950: extrema -> Morton keys -> radix sort -> gather ->
// bottom-up merge.  Replaces src/build.jl:198-271 and src/morton/*.jl of the reference.
//
// Data flow (BSphere{F32} leaves, n of them; bytes per leaf in brackets):
//   extrema_partial   read volumes [16]                         -> per-workgroup min/max
//   extrema_final     (1 workgroup) + epsilon expansion         -> 6 scalars in HBM (no host readback)
//   encode            read volumes [16], write key [4]          (values are implicit positions)
//   radix sort        ibvh_sort.hip
//   gather            read perm [4] + key [4] + volume [16 random], write record [24]
//   aggregate         read records [24 stride, 16 used], write nodes [24 * ~1]
#include "ibvh_common.hpp"
#include "ibvh_radix.hpp"
#include "ibvh_msd.hpp"

namespace ibvh {
namespace rsort { // ibvh_sort.hip: LSD passes over (key, position) pairs (+ the MSD / in-LDS hybrid on pairs)
struct FirstPassPlan {
    int tpb, ipt, num_tiles;
    uint32_t *tile_hist;
    uint32_t mask;
    int shift, bits;
};
FirstPassPlan first_pass_plan(int64_t n, int key_bits, int key_bytes, void *scratch);
bool uses_hybrid(int64_t n, int key_bits, int key_bytes);
int sort_pairs(int key_bytes, int key_bits, int64_t n, void *keys, void *vals, void *keys_alt, void *vals_alt,
               bool vals_implicit, int32_t *result_in_alt, void *scratch, size_t scratch_sz, hipStream_t st,
               bool first_hist_done, const RecordArgs *records);
size_t scratch_bytes(int64_t n);
} // namespace rsort

namespace build {

constexpr int EXT_TPB = 256;
constexpr int EXT_MAX_BLOCKS = 2048;
constexpr int EXT_FOLD_BLOCKS = 512; // ibvh_build: the partials are folded by every encode workgroup (ExtremaFold)

// ------------------------------------------------------------------------------------------
// M1: extrema of the centres — morton/utils.jl:1-72
// ------------------------------------------------------------------------------------------
template <class T> IBVH_D T wave_min(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T t = __shfl_xor(v, o, 64);
        v = v < t ? v : t;
    }
    return v;
}
template <class T> IBVH_D T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T t = __shfl_xor(v, o, 64);
        v = v > t ? v : t;
    }
    return v;
}

// min/max are exact and order-free (no NaNs), so any reduction tree reproduces the reference's
// mapreduce bit for bit.  Inits: min <- floatmax(T); max <- floatmin(T) (sic, utils.jl:39-40).
template <class V>
__global__ __launch_bounds__(EXT_TPB) void extrema_partial_kernel(const char *__restrict__ recs, int64_t stride, int64_t n,
                                                                  typename V::elt *__restrict__ partials) {
    using T = typename V::elt;
    T mn[3] = {float_max<T>(), float_max<T>(), float_max<T>()};
    T mx[3] = {float_min_normal<T>(), float_min_normal<T>(), float_min_normal<T>()};
    for (int64_t i = (int64_t)blockIdx.x * EXT_TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EXT_TPB) {
        V v = load_vol<V>(recs + i * stride);
        T c[3];
        center(v, c);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mn[k] = mn[k] < c[k] ? mn[k] : c[k];
            mx[k] = mx[k] > c[k] ? mx[k] : c[k];
        }
    }
    __shared__ T s[EXT_TPB / 64][6];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        T a = wave_min(mn[k]), b = wave_max(mx[k]);
        if (lane == 0) {
            s[w][k] = a;
            s[w][3 + k] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        T v = s[0][threadIdx.x];
        for (int i = 1; i < EXT_TPB / 64; ++i) {
            T t = s[i][threadIdx.x];
            v = threadIdx.x < 3 ? (v < t ? v : t) : (v > t ? v : t);
        }
        partials[(int64_t)blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// compute_skips! on device memory (build.jl:232-239), folded into whichever single-workgroup kernel the build launches
// first (one launch less than a kernel of its own)
struct SkipsOut {
    TreeDev tree;
    void *skips; // nullptr: nothing to write
    int index_bytes;
};
IBVH_D void write_skips(const SkipsOut &so) {
    const int64_t level = (int64_t)threadIdx.x + 1;
    if (so.skips == nullptr || level > so.tree.levels) return;
    const int64_t v = level_skips(so.tree.levels, so.tree.virtual_leaves, level);
    if (so.index_bytes == 4) ((int32_t *)so.skips)[level - 1] = (int32_t)v;
    else ((int64_t *)so.skips)[level - 1] = v;
}

// One workgroup folds the partials and applies the epsilon expansion of
// bounding_volumes_extrema (utils.jl:63-69): x -/+ rp*abs(x) -/+ floatmin, two roundings a side.
template <class T>
__global__ __launch_bounds__(EXT_TPB) void extrema_final_kernel(const T *__restrict__ partials, int nparts, int expand,
                                                                T *__restrict__ out, T *__restrict__ out2, SkipsOut so) {
    write_skips(so);
    T mn[3] = {float_max<T>(), float_max<T>(), float_max<T>()};
    T mx[3] = {float_min_normal<T>(), float_min_normal<T>(), float_min_normal<T>()};
    for (int i = threadIdx.x; i < nparts; i += EXT_TPB) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            T a = partials[i * 6 + k], b = partials[i * 6 + 3 + k];
            mn[k] = mn[k] < a ? mn[k] : a;
            mx[k] = mx[k] > b ? mx[k] : b;
        }
    }
    __shared__ T s[EXT_TPB / 64][6];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        T a = wave_min(mn[k]), b = wave_max(mx[k]);
        if (lane == 0) {
            s[w][k] = a;
            s[w][3 + k] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        T v = s[0][threadIdx.x];
        for (int i = 1; i < EXT_TPB / 64; ++i) {
            T t = s[i][threadIdx.x];
            v = threadIdx.x < 3 ? (v < t ? v : t) : (v > t ? v : t);
        }
        if (expand) {
            const T rp = relative_precision<T>(), fm = float_min_normal<T>();
            T a = rp * ibvh_abs(v);
            v = threadIdx.x < 3 ? (v - a) - fm : (v + a) + fm;
        }
        out[threadIdx.x] = v;
        if (out2) out2[threadIdx.x] = v; // the caller's copy (ibvh_build's extrema_out): no separate 24-byte memcpy
    }
}

template <class T> __global__ void extrema_set_kernel(T *out, T *out2, double a0, double a1, double a2, double b0, double b1, double b2, SkipsOut so) {
    write_skips(so);
    if (threadIdx.x == 0) {
        out[0] = T(a0);
        out[1] = T(a1);
        out[2] = T(a2);
        out[3] = T(b0);
        out[4] = T(b1);
        out[5] = T(b2);
        if (out2)
            for (int k = 0; k < 6; ++k) out2[k] = out[k];
    }
}

// ------------------------------------------------------------------------------------------
// M2: Morton keys — morton/default.jl:63-108
// ------------------------------------------------------------------------------------------
template <class V, class K>
__global__ __launch_bounds__(256) void encode_kernel(const char *__restrict__ recs, int64_t stride, int64_t n,
                                                     const typename V::elt *__restrict__ ext, int morton_type,
                                                     K *__restrict__ keys) {
    using T = typename V::elt;
    const T mins[3] = {ext[0], ext[1], ext[2]}, maxs[3] = {ext[3], ext[4], ext[5]};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        V v = load_vol<V>(recs + i * stride);
        T c[3];
        center(v, c);
        keys[i] = (K)morton_encode_single(c, mins, maxs, morton_type);
    }
}

// Same, fused with the radix sort's first per-tile digit histogram: the workgroup encodes exactly the
// keys of one sort tile and counts their lowest digit in LDS, so the sort's first hist pass (a 4 B/leaf
// re-read of the keys plus a launch) disappears.
// The fold of the extrema partials rides along (ExtremaFold): every workgroup folds the <= EXT_FOLD_BLOCKS partial results
// itself (a few KB of L2 hits) and applies the epsilon expansion — the identical operations in every workgroup, min / max
// being order-free — instead of waiting for a one-workgroup launch in between (extrema_final_kernel: ~6 us of pure
// launch / dependency latency at 1e6 leaves); workgroup 0 also publishes the result (the build's `extrema` output) and
// writes the skips.
template <class T> struct ExtremaFold {
    const T *partials; // nullptr: `ext` already holds the bounds
    int nparts;
    T *out, *out2;
    SkipsOut so;
};
template <class V, class K>
__global__ __launch_bounds__(1024) void encode_hist_kernel(const char *__restrict__ recs, int64_t stride, int64_t n,
                                                          const typename V::elt *__restrict__ ext, int morton_type,
                                                          K *__restrict__ keys, int tile_elems, int shift, uint32_t mask,
                                                          uint32_t *__restrict__ tile_hist, int num_tiles, int tile_major,
                                                          ExtremaFold<typename V::elt> fold) {
    using T = typename V::elt;
    extern __shared__ uint32_t h[]; // mask + 1 counters
    __shared__ T s_fold[16][6];
    __shared__ T s_ext[6];
    for (int i = threadIdx.x; i <= (int)mask; i += blockDim.x) h[i] = 0;
    if (fold.partials != nullptr) {
        if (blockIdx.x == 0) write_skips(fold.so);
        T mn[3] = {float_max<T>(), float_max<T>(), float_max<T>()};
        T mx[3] = {float_min_normal<T>(), float_min_normal<T>(), float_min_normal<T>()};
        for (int i = threadIdx.x; i < fold.nparts; i += blockDim.x) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const T a = fold.partials[i * 6 + k], b = fold.partials[i * 6 + 3 + k];
                mn[k] = mn[k] < a ? mn[k] : a;
                mx[k] = mx[k] > b ? mx[k] : b;
            }
        }
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T a = wave_min(mn[k]), b = wave_max(mx[k]);
            if (lane == 0) {
                s_fold[w][k] = a;
                s_fold[w][3 + k] = b;
            }
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            T v = s_fold[0][threadIdx.x];
            for (int i = 1; i < (int)(blockDim.x >> 6); ++i) {
                const T t = s_fold[i][threadIdx.x];
                v = threadIdx.x < 3 ? (v < t ? v : t) : (v > t ? v : t);
            }
            // bounding_volumes_extrema (utils.jl:63-69): x -/+ rp*abs(x) -/+ floatmin, two roundings a side
            const T rp = relative_precision<T>(), fm = float_min_normal<T>();
            const T a = rp * ibvh_abs(v);
            v = threadIdx.x < 3 ? (v - a) - fm : (v + a) + fm;
            s_ext[threadIdx.x] = v;
            if (blockIdx.x == 0) {
                fold.out[threadIdx.x] = v;
                if (fold.out2) fold.out2[threadIdx.x] = v;
            }
        }
    }
    __syncthreads();
    const T *bounds = fold.partials != nullptr ? s_ext : ext;
    const T mins[3] = {bounds[0], bounds[1], bounds[2]}, maxs[3] = {bounds[3], bounds[4], bounds[5]};
    const int64_t base = (int64_t)blockIdx.x * tile_elems;
    for (int j = threadIdx.x; j < tile_elems; j += blockDim.x) {
        const int64_t i = base + j;
        if (i < n) {
            V v = load_vol<V>(recs + i * stride);
            T c[3];
            center(v, c);
            const K k = (K)morton_encode_single(c, mins, maxs, morton_type);
            keys[i] = k;
            atomicAdd(&h[(uint32_t)(k >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    // digit-major ([digit][tile]) for the LSD passes of ibvh_sort.hip, tile-major ([tile][digit], one coalesced row)
    // for the MSD partition of ibvh_msd.hip
    if (tile_major)
        for (int d = threadIdx.x; d <= (int)mask; d += blockDim.x) tile_hist[(int64_t)blockIdx.x * (mask + 1) + d] = h[d];
    else
        for (int d = threadIdx.x; d <= (int)mask; d += blockDim.x) tile_hist[(int64_t)d * num_tiles + blockIdx.x] = h[d];
}

// ------------------------------------------------------------------------------------------
// gather: sorted (key, position) -> BoundingVolume records in Morton order
//   index = position + 1 for freshly wrapped volumes (wrap_bounding_volumes, build.jl:345-349)
//           or the source record's own .index (build.jl:220-222)
// ------------------------------------------------------------------------------------------
// Each wave assembles 64 records in LDS and writes them as ONE contiguous byte range with fully coalesced
// 8-byte stores (a record-per-lane store would touch every line of the range with each of its 3 partial
// stores); the random volume fetch of the next chunk is issued before the current chunk is written out.
template <class V, class I, class K>
__global__ __launch_bounds__(256) void gather_kernel(const char *__restrict__ src, int64_t src_stride, int src_wrapped,
                                                     LeafLayout lay, const K *__restrict__ keys,
                                                     const uint32_t *__restrict__ perm, int64_t n, char *__restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int words = lay.stride >> 3; // 8-byte words per record
    uint64_t *stage = (uint64_t *)g_smem + (size_t)wv * 64 * words;
    const int64_t nchunks = ceil_div_dev(n, 256);
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t i = c * 256 + threadIdx.x;
        const int64_t wave_first = c * 256 + wv * 64;
        if (i < n) {
            const uint32_t p = perm[i];
            const char *s = src + (int64_t)p * src_stride;
            V v;
            if (sizeof(V) % 16 == 0 && (src_stride & 15) == 0) v = load_vol16<V>(s); // one 16-byte request per leaf
            else v = load_vol<V>(s);
            I idx = src_wrapped ? load_index<I>(s, lay) : (I)((int64_t)p + 1);
            char *d = (char *)(stage + (size_t)lane * words);
            store_vol(d, v);
            *(I *)(d + lay.index_off) = idx;
            store_morton(d, lay, (uint64_t)keys[i]);
        }
        __builtin_amdgcn_wave_barrier();
        const int64_t wave_n = n - wave_first < 64 ? n - wave_first : 64; // records of this wave's chunk
        if (wave_n > 0) {
            uint64_t *out = (uint64_t *)(dst + wave_first * lay.stride);
            const int total = (int)wave_n * words;
            for (int g = lane; g < total; g += 64) out[g] = stage[g];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// A1 + A2: bottom-up merge — build.jl:366-523.  One launch builds up to CH_LEVELS tree levels:
// a workgroup owns 2*TPB consecutive inputs (leaves, or nodes of the level the previous launch
// ended on), merges them pairwise, and keeps folding through LDS, writing every level it makes.
// The reference launches once per level (build.jl:371-375: ~levels launches); this needs
// ceil((levels-1)/CH_LEVELS).
// ------------------------------------------------------------------------------------------
constexpr int AGG_TPB = 256;
constexpr int CH_LEVELS = 9; // 2*256 inputs -> 256,128,...,1 nodes

template <class L, class N, bool FROM_LEAVES>
__global__ __launch_bounds__(AGG_TPB) void aggregate_kernel(const char *__restrict__ in, int64_t in_stride, int64_t in_level,
                                                            TreeDev tree, int64_t built_level, N *__restrict__ nodes) {
    __shared__ N s_nodes[AGG_TPB];
    const int t = threadIdx.x;
    // level being produced first: in_level - 1
    int64_t level = in_level - 1;
    const int64_t in_real = FROM_LEAVES ? tree.real_leaves : level_num_real(tree.levels, tree.virtual_leaves, in_level);
    int64_t i = (int64_t)blockIdx.x * AGG_TPB + t; // 0-based node index within `level`
    int64_t nreal = level_num_real(tree.levels, tree.virtual_leaves, level);
    N mine;
    bool real = i < nreal;
    if (real) {
        int64_t l = 2 * i, r = 2 * i + 1; // 0-based children within in_level
        if constexpr (FROM_LEAVES) {
            L a = load_vol<L>(in + l * in_stride);
            if (r < in_real) {
                L b = load_vol<L>(in + r * in_stride);
                mine = merge_to(a, b, (N *)nullptr);
            } else {
                mine = convert_to(a, (N *)nullptr);
            }
        } else {
            N a = load_vol<N>(in + l * in_stride);
            if (r < in_real) {
                N b = load_vol<N>(in + r * in_stride);
                mine = merge_to(a, b, (N *)nullptr);
            } else {
                mine = a;
            }
        }
        nodes[level_start(tree.levels, tree.virtual_leaves, level) - 1 + i] = mine;
    }
    // fold upwards inside the workgroup
    int count = AGG_TPB;
#pragma unroll 1
    for (int s = 1; s < CH_LEVELS; ++s) {
        if (level - 1 < built_level) break; // uniform
        __syncthreads();
        if (t < count) s_nodes[t] = mine;
        __syncthreads();
        count >>= 1;
        level -= 1;
        const int64_t child_real = nreal;
        nreal = level_num_real(tree.levels, tree.virtual_leaves, level);
        i = (int64_t)blockIdx.x * count + t;
        real = (t < count) && (i < nreal);
        if (real) {
            N a = s_nodes[2 * t];
            if (2 * i + 1 < child_real) mine = merge_to(a, s_nodes[2 * t + 1], (N *)nullptr);
            else mine = a;
            nodes[level_start(tree.levels, tree.virtual_leaves, level) - 1 + i] = mine;
        }
    }
}

// The top of the tree: one workgroup folds all levels above `in_level` (<= AGG_TOP_MAX inputs), a barrier per level; same
// merge_to as below, so the nodes are bit-identical.  IN_LDS (round 3; the inputs fit 128 KB): the level being folded
// lives in LDS — inputs read from memory once, every level's nodes stored as they are made, and the next level folds the
// LDS copy — instead of each level re-reading through L2 what the level before has just stored (eleven dependent
// store -> load round trips at 1e6 leaves: 8.3 us for a kernel whose work is 4 k merges).
constexpr int AGG_TOP_TPB = 1024, AGG_TOP_MAX = 4096, AGG_TOP_LDS_BYTES = 128 * 1024;
template <class N, bool IN_LDS>
__global__ __launch_bounds__(AGG_TOP_TPB) void aggregate_top_kernel(TreeDev tree, int64_t in_level, int64_t built_level, N *nodes) {
    if constexpr (IN_LDS) {
        extern __shared__ __attribute__((aligned(16))) unsigned char top_smem[];
        N *s = (N *)top_smem;
        int64_t have = level_num_real(tree.levels, tree.virtual_leaves, in_level); // nodes of the level held in LDS
        {
            const N *in = nodes + (level_start(tree.levels, tree.virtual_leaves, in_level) - 1);
            for (int64_t i = threadIdx.x; i < have; i += AGG_TOP_TPB) s[i] = load_vol<N>(in + i);
        }
        __syncthreads();
        for (int64_t level = in_level - 1; level >= built_level && level >= 1; --level) {
            const int64_t nreal = level_num_real(tree.levels, tree.virtual_leaves, level);
            N *out = nodes + (level_start(tree.levels, tree.virtual_leaves, level) - 1);
            // (nreal <= AGG_TOP_MAX / 2 <= 2 * AGG_TOP_TPB: at most two nodes per thread, read before anything is overwritten)
            N mine[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int64_t i = threadIdx.x + (int64_t)k * AGG_TOP_TPB;
                if (i < nreal) mine[k] = (2 * i + 1 < have) ? merge_to(s[2 * i], s[2 * i + 1], (N *)nullptr) : s[2 * i];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int64_t i = threadIdx.x + (int64_t)k * AGG_TOP_TPB;
                if (i < nreal) {
                    s[i] = mine[k];
                    store_vol(out + i, mine[k]);
                }
            }
            __syncthreads();
            have = nreal;
        }
        return;
    }
    for (int64_t level = in_level - 1; level >= built_level && level >= 1; --level) {
        const int64_t nreal = level_num_real(tree.levels, tree.virtual_leaves, level);
        const int64_t child_real = level_num_real(tree.levels, tree.virtual_leaves, level + 1);
        const N *in = nodes + (level_start(tree.levels, tree.virtual_leaves, level + 1) - 1);
        N *out = nodes + (level_start(tree.levels, tree.virtual_leaves, level) - 1);
        for (int64_t i = threadIdx.x; i < nreal; i += AGG_TOP_TPB) {
            const N a = load_vol<N>(in + 2 * i);
            if (2 * i + 1 < child_real) store_vol(out + i, merge_to(a, load_vol<N>(in + 2 * i + 1), (N *)nullptr));
            else store_vol(out + i, a);
        }
        __threadfence_block();
        __syncthreads();
    }
}

template <class L, class N>
int aggregate(const char *leaves, int64_t leaf_stride, const ibvh_tree &tree, int64_t built_level, N *nodes, hipStream_t st) {
    if (tree.real_nodes < 2) return IBVH_OK; // build.jl:266
    TreeDev td{tree.levels, tree.real_leaves, tree.virtual_leaves};
    // launch 1: leaves (level `levels`) -> levels-1 .. levels-CH_LEVELS.  The last-level merge always
    // runs (aggregate_last_level!, build.jl:369), even when built_level == levels.
    int64_t in_level = tree.levels;
    {
        int64_t nreal = level_num_real(tree.levels, tree.virtual_leaves, in_level - 1);
        int64_t eff_built = built_level > tree.levels - 1 ? tree.levels - 1 : built_level;
        IBVH_LAUNCH((aggregate_kernel<L, N, true>), dim3((unsigned)ceil_div(nreal, AGG_TPB)), dim3(AGG_TPB), 0, st,
                           leaves, leaf_stride, in_level, td, eff_built, nodes);
        IBVH_LAUNCH_CHECK();
        in_level = in_level - CH_LEVELS;
    }
    while (in_level - 1 >= built_level && in_level - 1 >= 1) {
        if (level_num_real(tree.levels, tree.virtual_leaves, in_level) <= AGG_TOP_MAX) {
            // few nodes left: ONE workgroup folds every remaining level (a launch costs more than these levels)
            const size_t top_bytes = (size_t)level_num_real(tree.levels, tree.virtual_leaves, in_level) * sizeof(N);
            if (top_bytes <= (size_t)AGG_TOP_LDS_BYTES) {
                IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)aggregate_top_kernel<N, true>, hipFuncAttributeMaxDynamicSharedMemorySize, AGG_TOP_LDS_BYTES));
                IBVH_LAUNCH((aggregate_top_kernel<N, true>), dim3(1), dim3(AGG_TOP_TPB), top_bytes, st, td, in_level, built_level, nodes);
            } else {
                IBVH_LAUNCH((aggregate_top_kernel<N, false>), dim3(1), dim3(AGG_TOP_TPB), 0, st, td, in_level, built_level, nodes);
            }
            IBVH_LAUNCH_CHECK();
            break;
        }
        int64_t nreal = level_num_real(tree.levels, tree.virtual_leaves, in_level - 1);
        const char *in = (const char *)(nodes + (level_start(tree.levels, tree.virtual_leaves, in_level) - 1));
        IBVH_LAUNCH((aggregate_kernel<L, N, false>), dim3((unsigned)ceil_div(nreal, AGG_TPB)), dim3(AGG_TPB), 0, st,
                           in, (int64_t)sizeof(N), in_level, td, built_level, nodes);
        IBVH_LAUNCH_CHECK();
        in_level -= CH_LEVELS;
    }
    return IBVH_OK;
}

// compute_skips! on device memory in the index type I (build.jl:232-239)
template <class I> __global__ void skips_kernel(TreeDev tree, I *skips) {
    int64_t level = threadIdx.x + 1;
    if (level <= tree.levels) skips[level - 1] = (I)level_skips(tree.levels, tree.virtual_leaves, level);
}

inline int grid_for(int64_t n, int tpb, int max_blocks) {
    int64_t b = ceil_div(n, tpb);
    return (int)(b < 1 ? 1 : (b > max_blocks ? max_blocks : b));
}

// scratch carve-up -----------------------------------------------------------------------------
struct Scratch {
    char *partials;  // EXT_MAX_BLOCKS * 6 * 8
    char *extrema;   // 6 * 8
    char *keys, *keys_alt; // n * key_bytes
    char *vals, *vals_alt; // n * 4
    char *records;   // n * leaf_bytes: the partitioned records (and the staging of an in-place LSD build)
    char *records2;  // n * leaf_bytes: second partition level (oversized cells only)
    char *sort;      // rsort::scratch_bytes(n)
    size_t total;
};
inline int morton_key_bits(int morton_type);
inline Scratch carve(char *base, int64_t n, int key_bytes, int64_t leaf_bytes, int morton_type) {
    Scratch s;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + off : nullptr;
        off += (size_t)align_up((int64_t)bytes, 256);
        return p;
    };
    s.partials = take((size_t)EXT_MAX_BLOCKS * 6 * 8);
    s.extrema = take(64);
    s.keys = take((size_t)n * key_bytes);
    s.keys_alt = take((size_t)n * key_bytes);
    s.vals = take((size_t)n * 4);
    s.vals_alt = take((size_t)n * 4);
    s.records = take((size_t)n * leaf_bytes);
    s.records2 = take((size_t)n * leaf_bytes);
    const size_t a = rsort::scratch_bytes(n), b = msd::scratch_bytes(n, morton_key_bits(morton_type), key_bytes, (int)leaf_bytes);
    s.sort = take(a > b ? a : b);
    s.total = off;
    return s;
}

// the first half alone: per-workgroup partial extrema for a consumer that folds them itself (encode_hist_kernel)
template <class V> int extrema_partials(const char *recs, int64_t stride, int64_t n, char *partials, hipStream_t st, int *nparts) {
    using T = typename V::elt;
    const int blocks = grid_for(n, EXT_TPB * 4, EXT_FOLD_BLOCKS);
    IBVH_LAUNCH((extrema_partial_kernel<V>), dim3(blocks), dim3(EXT_TPB), 0, st, recs, stride, n, (T *)partials);
    *nparts = blocks;
    return IBVH_OK;
}
template <class V>
int extrema(const char *recs, int64_t stride, int64_t n, int expand, typename V::elt *out, char *partials, hipStream_t st,
            typename V::elt *out2 = nullptr, SkipsOut so = SkipsOut{TreeDev{0, 0, 0}, nullptr, 4}) {
    using T = typename V::elt;
    int blocks = grid_for(n, EXT_TPB * 4, EXT_MAX_BLOCKS);
    IBVH_LAUNCH((extrema_partial_kernel<V>), dim3(blocks), dim3(EXT_TPB), 0, st, recs, stride, n, (T *)partials);
    IBVH_LAUNCH((extrema_final_kernel<T>), dim3(1), dim3(EXT_TPB), 0, st, (const T *)partials, blocks, expand, out, out2, so);
    IBVH_LAUNCH_CHECK();
    return IBVH_OK;
}

template <class V>
int encode(const char *recs, int64_t stride, int64_t n, const typename V::elt *ext, int morton_type, void *keys, hipStream_t st) {
    int blocks = grid_for(n, 256, 256 * 16);
    if (morton_type == IBVH_U64)
        IBVH_LAUNCH((encode_kernel<V, uint64_t>), dim3(blocks), dim3(256), 0, st, recs, stride, n, ext, morton_type,
                           (uint64_t *)keys);
    else
        IBVH_LAUNCH((encode_kernel<V, uint32_t>), dim3(blocks), dim3(256), 0, st, recs, stride, n, ext, morton_type,
                           (uint32_t *)keys);
    IBVH_LAUNCH_CHECK();
    return IBVH_OK;
}

inline int morton_key_bits(int morton_type) { return morton_type == IBVH_U16 ? 15 : (morton_type == IBVH_U32 ? 30 : 63); }

} // namespace build
} // namespace ibvh

using namespace ibvh;
using namespace ibvh::build;

extern "C" {

ibvh_status ibvh_build_scratch_bytes(const ibvh_types *types, int64_t n, size_t *bytes_out) {
    if (!types || !bytes_out) return IBVH_ERR_INVALID_ARG;
    ibvh_layout lay;
    if (!layout_of(*types, lay)) return IBVH_ERR_UNSUPPORTED;
    if (n < 1) return IBVH_ERR_DOMAIN;
    int kb = types->morton_type == IBVH_U64 ? 8 : 4;
    *bytes_out = carve(nullptr, n, kb, lay.leaf_bytes, types->morton_type).total;
    return IBVH_OK;
}

ibvh_status ibvh_extrema(const ibvh_types *types, const void *records, int32_t wrapped, int64_t n, int32_t expand,
                         void *extrema_out, void *scratch, size_t scratch_bytes, void *stream) {
    if (!types || !records || !extrema_out || !scratch) return IBVH_ERR_INVALID_ARG;
    if (n < 1) return IBVH_ERR_DOMAIN;
    if (scratch_bytes < (size_t)EXT_MAX_BLOCKS * 6 * 8) return IBVH_ERR_SCRATCH;
    ibvh_layout lay;
    if (!layout_of(*types, lay)) return IBVH_ERR_UNSUPPORTED;
    int64_t stride = wrapped ? lay.leaf_bytes : lay.volume_bytes;
    return (ibvh_status)dispatch_volume(types->leaf_kind, types->leaf_float, [&](auto lt) -> int {
        using V = typename decltype(lt)::type;
        return extrema<V>((const char *)records, stride, n, expand, (typename V::elt *)extrema_out, (char *)scratch,
                          (hipStream_t)stream);
    });
}

ibvh_status ibvh_morton_keys(const ibvh_types *types, const void *records, int32_t wrapped, int64_t n, const void *ext,
                             void *keys_out, void *stream) {
    if (!types || !records || !ext || !keys_out) return IBVH_ERR_INVALID_ARG;
    if (n < 1) return IBVH_ERR_DOMAIN;
    ibvh_layout lay;
    if (!layout_of(*types, lay)) return IBVH_ERR_UNSUPPORTED;
    int64_t stride = wrapped ? lay.leaf_bytes : lay.volume_bytes;
    return (ibvh_status)dispatch_volume(types->leaf_kind, types->leaf_float, [&](auto lt) -> int {
        using V = typename decltype(lt)::type;
        return encode<V>((const char *)records, stride, n, (const typename V::elt *)ext, types->morton_type, keys_out,
                         (hipStream_t)stream);
    });
}

ibvh_status ibvh_aggregate(const ibvh_types *types, const ibvh_tree *tree, int64_t built_level, const void *leaves,
                           void *nodes, void *stream) {
    if (!types || !tree || !leaves) return IBVH_ERR_INVALID_ARG;
    if (built_level < 1 || built_level > tree->levels) return IBVH_ERR_INVALID_ARG;
    ibvh_layout lay;
    if (!layout_of(*types, lay)) return IBVH_ERR_UNSUPPORTED;
    if (tree->real_nodes >= 2 && !nodes) return IBVH_ERR_INVALID_ARG;
    return (ibvh_status)dispatch_leaf_node(*types, [&](auto lt, auto nt) -> int {
        using L = typename decltype(lt)::type;
        using N = typename decltype(nt)::type;
        return aggregate<L, N>((const char *)leaves, lay.leaf_bytes, *tree, built_level, (N *)nodes, (hipStream_t)stream);
    });
}

ibvh_status ibvh_build(const ibvh_build_desc *desc, const void *volumes, void *leaves, void *nodes, void *skips,
                       void *extrema_out, void *scratch, size_t scratch_bytes, void *stream) {
    if (!desc || !leaves || !skips || !scratch) return IBVH_ERR_INVALID_ARG;
    const ibvh_types &ty = desc->types;
    ibvh_layout lay;
    LeafLayout dlay;
    if (!layout_of(ty, lay, &dlay)) return IBVH_ERR_UNSUPPORTED;
    ibvh_tree tree;
    if (ibvh_status e = ibvh_tree_shape(desc->n, &tree)) return e;
    if (desc->n >= (int64_t(1) << 32)) return IBVH_ERR_INVALID_ARG; // positions are uint32 in the sort
    if (ty.index_type == IBVH_I32 && tree.real_nodes > INT32_MAX) return IBVH_ERR_OVERFLOW;
    if (desc->built_level < 1 || desc->built_level > tree.levels) return IBVH_ERR_INVALID_ARG; // build.jl:314
    if (!desc->already_wrapped && !volumes) return IBVH_ERR_INVALID_ARG;
    if (tree.real_nodes >= 2 && !nodes) return IBVH_ERR_INVALID_ARG;
    const int64_t n = desc->n;
    const int key_bytes = ty.morton_type == IBVH_U64 ? 8 : 4;
    const bool wrapped = desc->already_wrapped != 0;
    Scratch sc = carve((char *)scratch, n, key_bytes, lay.leaf_bytes, ty.morton_type);
    if (scratch_bytes < sc.total) return IBVH_ERR_SCRATCH;
    hipStream_t st = (hipStream_t)stream;

    // already_wrapped with a non-NULL `volumes`: `volumes` holds the source RECORDS and `leaves` receives the sorted
    // ones (out of place: no staging copy); with volumes == NULL the caller's `leaves` are sorted in place.
    const bool out_of_place = wrapped && volumes != nullptr;
    const char *src = wrapped ? (out_of_place ? (const char *)volumes : (const char *)leaves) : (const char *)volumes;
    const int64_t src_stride = wrapped ? lay.leaf_bytes : lay.volume_bytes;

    int rc = dispatch_leaf_node(ty, [&](auto lt, auto nt) -> int {
        using L = typename decltype(lt)::type;
        using N = typename decltype(nt)::type;
        using T = typename L::elt;
        T *ext = (T *)sc.extrema;
        // skips
        TreeDev td{tree.levels, tree.real_leaves, tree.virtual_leaves};
        const SkipsOut so{td, skips, ty.index_type == IBVH_I32 ? 4 : 8};
        // extrema (or caller-fixed bounds: morton/default.jl:52-57)
        ExtremaFold<T> fold{nullptr, 0, ext, (T *)extrema_out, so};
        // Small builds are launch-latency bound: the encode kernel's workgroups fold the partial extrema themselves (one
        // launch less on the critical path: -6 us at 1e6 leaves).  With thousands of encode workgroups the redundant folds
        // cost more than the launch they save (1e7 leaves: encode 35 -> 45 us), so large builds keep the one-workgroup fold.
        const bool fold_in_encode = n <= (int64_t)1 << 21;
        if (desc->compute_extrema && fold_in_encode) {
            int nparts = 0;
            if (int e = extrema_partials<L>(src, src_stride, n, sc.partials, st, &nparts)) return e;
            fold.partials = (const T *)sc.partials;
            fold.nparts = nparts;
        } else if (desc->compute_extrema) {
            if (int e = extrema<L>(src, src_stride, n, 1, ext, sc.partials, st, (T *)extrema_out, so)) return e;
        } else {
            IBVH_LAUNCH((extrema_set_kernel<T>), dim3(1), dim3(64), 0, st, ext, (T *)extrema_out, desc->mins[0], desc->mins[1],
                               desc->mins[2], desc->maxs[0], desc->maxs[1], desc->maxs[2], so);
        }
        // keys, fused with the first per-tile digit histogram of the sort
        const int key_bits = morton_key_bits(ty.morton_type);
        const msd::Plan mp = msd::make_plan(n, key_bits, key_bytes, (int)lay.leaf_bytes, sc.sort);
        rsort::FirstPassPlan plan;
        if (mp.bits) plan = rsort::FirstPassPlan{mp.ptpb, mp.pipt, mp.num_tiles, mp.tb.tile_hist, (1u << mp.bits) - 1u, mp.shift, mp.bits};
        else plan = rsort::first_pass_plan(n, key_bits, key_bytes, sc.sort);
        if (key_bytes == 8)
            IBVH_LAUNCH((encode_hist_kernel<L, uint64_t>), dim3(plan.num_tiles), dim3(plan.tpb), ((size_t)plan.mask + 1) * 4, st,
                        src, src_stride, n, ext, ty.morton_type, (uint64_t *)sc.keys, plan.tpb * plan.ipt, plan.shift, plan.mask,
                        plan.tile_hist, plan.num_tiles, mp.bits ? 1 : 0, fold);
        else
            IBVH_LAUNCH((encode_hist_kernel<L, uint32_t>), dim3(plan.num_tiles), dim3(plan.tpb), ((size_t)plan.mask + 1) * 4, st,
                        src, src_stride, n, ext, ty.morton_type, (uint32_t *)sc.keys, plan.tpb * plan.ipt, plan.shift, plan.mask,
                        plan.tile_hist, plan.num_tiles, mp.bits ? 1 : 0, fold);
        IBVH_LAUNCH_CHECK();
        if (mp.bits) {
            // the default: ONE partition of the finished records by the top bits of their key, buckets finished in LDS
            // (ibvh_msd.hip).  index = position + 1 for fresh volumes (build.jl:345-349) or the source record's own
            // index (:220-222); in-place builds read `leaves`, stage in scratch and write `leaves`: no extra copy.
            rsort::RecordArgs ra{src, sc.records, src_stride, wrapped ? 1 : 0, (int32_t)(lay.volume_bytes / 8),
                                 ty.index_type == IBVH_I32 ? 4 : 8, dlay};
            if (int e = msd::sort_records(mp, key_bytes, key_bits, sc.keys, n, ra, sc.records2, (char *)leaves, sc.keys_alt, (uint32_t *)sc.vals_alt, sc.keys,
                                          (uint32_t *)sc.vals, desc->sort_levels, desc->sort_equalize != 0, desc->skew_flag, st))
                return e;
            return aggregate<L, N>((const char *)leaves, lay.leaf_bytes, tree, desc->built_level, (N *)nodes, st);
        }
        if (desc->skew_flag) IBVH_HIP_CHECK(hipMemsetAsync(desc->skew_flag, 0, 4, st)); // (LSD passes do not care about skew)
        // stable LSB radix sort of (key, position), then the records in Morton order
        // (index = position + 1 for fresh volumes, build.jl:345-349, or the source record's own index, :220-222).
        // In-place (already wrapped) builds go through scratch records.
        // Below ~4 M leaves the LAST radix pass writes the finished records itself (one launch and one
        // (key, position) round trip fewer: 0.157 -> 0.140 ms at 1e6); above, the dedicated gather kernel's higher
        // occupancy serves the random volume reads better (measured at 1e7: 0.049 + 0.276 ms vs 0.350 ms fused).
        char *dst = (wrapped && !out_of_place) ? sc.records : (char *)leaves;
        // (with the MSD + in-LDS hybrid, the default up to ~6 M leaves, the bucket kernel always writes the records)
        const bool fuse_records = rsort::uses_hybrid(n, key_bits, key_bytes) || n < (int64_t(1) << 22);
        rsort::RecordArgs ra{src, dst, src_stride, wrapped ? 1 : 0, (int32_t)(lay.volume_bytes / 8),
                             ty.index_type == IBVH_I32 ? 4 : 8, dlay};
        int32_t in_alt = 0;
        if (int e = rsort::sort_pairs(key_bytes, key_bits, n, sc.keys, sc.vals, sc.keys_alt, sc.vals_alt, true, &in_alt,
                                      sc.sort, rsort::scratch_bytes(n), st, true, fuse_records ? &ra : nullptr))
            return e;
        if (!fuse_records) {
            const void *skeys = in_alt ? sc.keys_alt : sc.keys;
            const uint32_t *sperm = (const uint32_t *)(in_alt ? sc.vals_alt : sc.vals);
            int gblocks = grid_for(n, 256, 256 * 16);
            auto launch_gather = [&](auto it) -> int {
                using I = typename decltype(it)::type;
                const size_t gsm = (size_t)256 * lay.leaf_bytes;
                if (key_bytes == 8)
                    IBVH_LAUNCH((gather_kernel<L, I, uint64_t>), dim3(gblocks), dim3(256), gsm, st, src, src_stride,
                                wrapped ? 1 : 0, dlay, (const uint64_t *)skeys, sperm, n, dst);
                else
                    IBVH_LAUNCH((gather_kernel<L, I, uint32_t>), dim3(gblocks), dim3(256), gsm, st, src, src_stride,
                                wrapped ? 1 : 0, dlay, (const uint32_t *)skeys, sperm, n, dst);
                return IBVH_OK;
            };
            if (int e = dispatch_index(ty.index_type, launch_gather)) return e;
            IBVH_LAUNCH_CHECK();
        }
        if (wrapped && !out_of_place)
            IBVH_HIP_CHECK(hipMemcpyAsync(leaves, sc.records, (size_t)n * lay.leaf_bytes, hipMemcpyDeviceToDevice, st));
        // merge
        return aggregate<L, N>((const char *)leaves, lay.leaf_bytes, tree, desc->built_level, (N *)nodes, st);
    });
    return (ibvh_status)rc;
}

} // extern "C"
