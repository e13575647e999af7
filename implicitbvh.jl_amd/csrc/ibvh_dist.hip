// ibvh_dist.hip — device pieces of the multi-GPU build (no counterpart in the reference, which is
// single-device): the global centre AABB is an RCCL all-reduce of per-GPU extrema, the Morton sort a
// distributed radix sort — splitter keys found by refining digit histograms that are all-reduced, one
// all-to-all of packed BoundingVolume records, then the ordinary local stable sort (ibvh_build).
// The collectives themselves are issued by the host side (implicitbvh.jl_amd/dist.py) through
// torch.distributed (backend "nccl" = RCCL over xGMI); this file holds the kernels around them.
#include "ibvh_common.hpp"

namespace ibvh {
namespace rsort { // ibvh_sort.hip
struct RecordArgs;
int sort_pairs(int key_bytes, int key_bits, int64_t n, void *keys, void *vals, void *keys_alt, void *vals_alt, bool vals_implicit,
               int32_t *result_in_alt, void *scratch, size_t scratch_sz, hipStream_t st, bool first_hist_done,
               const RecordArgs *records);
size_t scratch_bytes(int64_t n);
} // namespace rsort
namespace distk {

// destination rank of every key: the number of splitters <= key (keys in [k_r, k_{r+1}) go to rank r)
constexpr int MAX_SPLITTERS = 255;
struct Splitters {
    uint64_t v[MAX_SPLITTERS];
};
// counts (optional, zeroed by the caller): leaves per destination rank, accumulated through an LDS histogram (one global
// atomic per workgroup and non-empty destination)
template <class K>
__global__ __launch_bounds__(256) void dest_kernel(const K *__restrict__ keys, int64_t n, Splitters sp, int nsplit, uint32_t *__restrict__ dest,
                                                   unsigned long long *__restrict__ counts) {
    __shared__ uint32_t s_cnt[MAX_SPLITTERS + 1];
    if (counts) {
        for (int i = threadIdx.x; i <= nsplit; i += 256) s_cnt[i] = 0;
        __syncthreads();
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint64_t k = (uint64_t)keys[i];
        int lo = 0, hi = nsplit; // first splitter > k
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sp.v[mid] <= k) lo = mid + 1;
            else hi = mid;
        }
        dest[i] = (uint32_t)lo;
        if (counts) atomicAdd(&s_cnt[lo], 1u);
    }
    if (counts) {
        __syncthreads();
        for (int i = threadIdx.x; i <= nsplit; i += 256)
            if (s_cnt[i]) atomicAdd(&counts[i], (unsigned long long)s_cnt[i]);
    }
}

// epsilon expansion of bounding_volumes_extrema (morton/utils.jl:63-69) applied to already reduced
// extrema: mins - rp*|mins| - floatmin, maxs + rp*|maxs| + floatmin, two roundings per side
template <class T> __global__ void expand_kernel(T *ext) {
    const int k = threadIdx.x;
    if (k < 6) {
        const T rp = relative_precision<T>(), fm = float_min_normal<T>();
        T v = ext[k];
        T a = rp * ibvh_abs(v);
        ext[k] = k < 3 ? (v - a) - fm : (v + a) + fm;
    }
}

// The all-reduce(MAX) vector of the distributed build, assembled on device in one launch:
// vec = [-mins(3), maxs(3), one-hot leaf counts(nranks)] as float64 (float -> double is exact and min x = -max(-x)),
// with the reference's neutral elements (morton/utils.jl:29-40: floatmax for minima, floatmin for maxima) when this
// rank has no leaves.
template <class T> __global__ void pack_extrema_kernel(const T *ext, int has_data, int rank, int nranks, double n_local, double *vec) {
    const int k = threadIdx.x;
    if (k < 3) vec[k] = has_data ? -(double)ext[k] : -(double)float_max<T>();
    else if (k < 6) vec[k] = has_data ? (double)ext[k] : (double)float_min_normal<T>();
    else if (k < 6 + nranks) vec[k] = (k - 6 == rank) ? n_local : 0.0;
}
// ... and taken apart again after the collective: global extrema in the leaf float type, epsilon-expanded
// (double -> float of values that came from floats is exact).
template <class T> __global__ void unpack_extrema_kernel(const double *vec, T *ext) {
    const int k = threadIdx.x;
    if (k < 6) {
        const T rp = relative_precision<T>(), fm = float_min_normal<T>();
        const T v = k < 3 ? (T)(-vec[k]) : (T)vec[k];
        const T a = rp * ibvh_abs(v);
        ext[k] = k < 3 ? (v - a) - fm : (v + a) + fm;
    }
}

// Digit histograms for the splitter search: out[j][d] = #keys with (key >> prefix_shift) == prefix[j]
// and digit d = (key >> shift) & mask; nprefix == 0: one histogram over all keys.  LDS-staged.
constexpr int HIST_TPB = 256;
constexpr int MAX_PREFIX = 15;
struct Prefixes {
    uint64_t v[MAX_PREFIX];
};
template <class K>
__global__ __launch_bounds__(HIST_TPB) void key_hist_kernel(const K *__restrict__ keys, int64_t n, int shift, int bits,
                                                            int prefix_shift, Prefixes pre, int nprefix,
                                                            uint32_t *__restrict__ out) {
    extern __shared__ uint32_t sh[];
    const int nb = 1 << bits, rows = nprefix > 0 ? nprefix : 1;
    for (int i = threadIdx.x; i < rows * nb; i += HIST_TPB) sh[i] = 0;
    __syncthreads();
    const uint32_t mask = (uint32_t)nb - 1u;
    for (int64_t i = (int64_t)blockIdx.x * HIST_TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * HIST_TPB) {
        const uint64_t k = (uint64_t)keys[i];
        const uint32_t d = (uint32_t)(k >> shift) & mask;
        if (nprefix == 0) {
            atomicAdd(&sh[d], 1u);
        } else {
            const uint64_t p = prefix_shift >= 64 ? 0 : (k >> prefix_shift);
            for (int j = 0; j < nprefix; ++j)
                if (p == pre.v[j]) atomicAdd(&sh[j * nb + d], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * nb; i += HIST_TPB) {
        uint32_t c = sh[i];
        if (c) atomicAdd(&out[i], c);
    }
}

// records for the exchange: out[i] = BoundingVolume{ volumes[p], index_base + p + 1, keys[p] }, p = perm[i] or i
template <class V, class I, class K>
__global__ __launch_bounds__(256) void pack_kernel(const V *__restrict__ vols, const K *__restrict__ keys,
                                                   const uint32_t *__restrict__ perm, int64_t index_base, int64_t n,
                                                   LeafLayout lay, char *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t p = perm ? (int64_t)perm[i] : i;
        char *d = out + i * lay.stride;
        store_vol(d, load_vol<V>(vols + p));
        *(I *)(d + lay.index_off) = (I)(index_base + p + 1);
        store_morton(d, lay, (uint64_t)keys[p]);
    }
}

} // namespace distk
} // namespace ibvh

using namespace ibvh;

extern "C" {

ibvh_status ibvh_expand_extrema(int32_t flt, void *extrema, void *stream) {
    if (!extrema) return IBVH_ERR_INVALID_ARG;
    if (flt == IBVH_F32) IBVH_LAUNCH((distk::expand_kernel<float>), dim3(1), dim3(64), 0, (hipStream_t)stream, (float *)extrema);
    else if (flt == IBVH_F64) IBVH_LAUNCH((distk::expand_kernel<double>), dim3(1), dim3(64), 0, (hipStream_t)stream, (double *)extrema);
    else return IBVH_ERR_INVALID_ARG;
    return hipGetLastError() == hipSuccess ? IBVH_OK : IBVH_ERR_HIP;
}

ibvh_status ibvh_dist_pack_extrema(int32_t flt, const void *extrema, int32_t has_data, int32_t rank, int32_t nranks,
                                   int64_t n_local, void *vec_out, void *stream) {
    if (!vec_out || nranks < 1 || nranks > 1018 || rank < 0 || rank >= nranks || n_local < 0 || (has_data && !extrema)) return IBVH_ERR_INVALID_ARG;
    if (flt == IBVH_F32)
        IBVH_LAUNCH((distk::pack_extrema_kernel<float>), dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float *)extrema, has_data,
                    rank, nranks, (double)n_local, (double *)vec_out);
    else if (flt == IBVH_F64)
        IBVH_LAUNCH((distk::pack_extrema_kernel<double>), dim3(1), dim3(1024), 0, (hipStream_t)stream, (const double *)extrema, has_data,
                    rank, nranks, (double)n_local, (double *)vec_out);
    else return IBVH_ERR_INVALID_ARG;
    return hipGetLastError() == hipSuccess ? IBVH_OK : IBVH_ERR_HIP;
}
ibvh_status ibvh_dist_unpack_extrema(int32_t flt, const void *vec, void *extrema_out, void *stream) {
    if (!vec || !extrema_out) return IBVH_ERR_INVALID_ARG;
    if (flt == IBVH_F32) IBVH_LAUNCH((distk::unpack_extrema_kernel<float>), dim3(1), dim3(64), 0, (hipStream_t)stream, (const double *)vec, (float *)extrema_out);
    else if (flt == IBVH_F64) IBVH_LAUNCH((distk::unpack_extrema_kernel<double>), dim3(1), dim3(64), 0, (hipStream_t)stream, (const double *)vec, (double *)extrema_out);
    else return IBVH_ERR_INVALID_ARG;
    return hipGetLastError() == hipSuccess ? IBVH_OK : IBVH_ERR_HIP;
}

// Stable partition of the local leaves by destination rank: perm_out[j] = source position of the j-th leaf in
// (destination rank, source position) order.  One destination kernel + ONE stable radix pass over (rank, position).
ibvh_status ibvh_dist_partition_scratch_bytes(int64_t n, size_t *bytes_out) {
    if (!bytes_out || n < 0) return IBVH_ERR_INVALID_ARG;
    *bytes_out = (size_t)align_up(n * 4, 256) * 3 + rsort::scratch_bytes(n);
    return IBVH_OK;
}
ibvh_status ibvh_dist_partition(int32_t key_bytes, const void *keys, int64_t n, const uint64_t *splitters, int32_t nranks,
                                void *perm_out, void *counts_out, void *scratch, size_t scratch_bytes, void *stream) {
    if (n < 0 || nranks < 1 || (key_bytes != 4 && key_bytes != 8)) return IBVH_ERR_INVALID_ARG;
    if (nranks - 1 > distk::MAX_SPLITTERS) return IBVH_ERR_UNSUPPORTED;
    if (counts_out && hipMemsetAsync(counts_out, 0, (size_t)nranks * 8, (hipStream_t)stream) != hipSuccess) return IBVH_ERR_HIP;
    if (n == 0) return IBVH_OK;
    if (!keys || !perm_out || !scratch || (nranks > 1 && !splitters)) return IBVH_ERR_INVALID_ARG;
    size_t need;
    ibvh_dist_partition_scratch_bytes(n, &need);
    if (scratch_bytes < need) return IBVH_ERR_SCRATCH;
    hipStream_t st = (hipStream_t)stream;
    const size_t slab = (size_t)align_up(n * 4, 256);
    uint32_t *dest = (uint32_t *)scratch, *dest_alt = (uint32_t *)((char *)scratch + slab), *vals_pri = (uint32_t *)((char *)scratch + 2 * slab);
    void *sort_scratch = (char *)scratch + 3 * slab;
    distk::Splitters sp;
    for (int i = 0; i < nranks - 1; ++i) sp.v[i] = splitters[i];
    const int blocks = (int)(ceil_div(n, 256) < 4096 ? ceil_div(n, 256) : 4096);
    unsigned long long *cnt = (unsigned long long *)counts_out;
    if (key_bytes == 8) IBVH_LAUNCH((distk::dest_kernel<uint64_t>), dim3(blocks), dim3(256), 0, st, (const uint64_t *)keys, n, sp, nranks - 1, dest, cnt);
    else IBVH_LAUNCH((distk::dest_kernel<uint32_t>), dim3(blocks), dim3(256), 0, st, (const uint32_t *)keys, n, sp, nranks - 1, dest, cnt);
    int bits = 1;
    while ((1 << bits) < nranks) ++bits; // <= 8: exactly one LSD pass, whose output lands in the alternate buffers
    int32_t in_alt = 0;
    if (int e = rsort::sort_pairs(4, bits, n, dest, vals_pri, dest_alt, perm_out, true, &in_alt, sort_scratch, rsort::scratch_bytes(n), st,
                                  false, nullptr))
        return (ibvh_status)e;
    if (!in_alt) // (cannot happen for one pass; keep the contract anyway)
        if (hipMemcpyAsync(perm_out, vals_pri, (size_t)n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return IBVH_ERR_HIP;
    return hipGetLastError() == hipSuccess ? IBVH_OK : IBVH_ERR_HIP;
}

ibvh_status ibvh_key_histogram(int32_t key_bytes, const void *keys, int64_t n, int32_t shift, int32_t bits,
                               int32_t prefix_shift, const uint64_t *prefixes, int32_t nprefix, void *out, void *stream) {
    if (n < 0 || bits < 1 || bits > 12 || shift < 0 || nprefix < 0 || nprefix > distk::MAX_PREFIX || !out) return IBVH_ERR_INVALID_ARG;
    if (key_bytes != 4 && key_bytes != 8) return IBVH_ERR_INVALID_ARG;
    if (nprefix > 0 && !prefixes) return IBVH_ERR_INVALID_ARG;
    const int rows = nprefix > 0 ? nprefix : 1;
    const size_t smem = (size_t)rows * ((size_t)1 << bits) * 4;
    if (smem > 160 * 1024) return IBVH_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, smem, st) != hipSuccess) return IBVH_ERR_HIP;
    if (n == 0) return IBVH_OK;
    distk::Prefixes pre{};
    for (int j = 0; j < nprefix; ++j) pre.v[j] = prefixes[j];
    int64_t b = ceil_div(n, distk::HIST_TPB * 16);
    unsigned blocks = (unsigned)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
    if (key_bytes == 4) {
        if (hipFuncSetAttribute((const void *)distk::key_hist_kernel<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) // (always the same value: concurrent callers cannot interleave badly)
            return IBVH_ERR_HIP;
        IBVH_LAUNCH((distk::key_hist_kernel<uint32_t>), dim3(blocks), dim3(distk::HIST_TPB), smem, st, (const uint32_t *)keys, n,
                    shift, bits, prefix_shift, pre, nprefix, (uint32_t *)out);
    } else {
        if (hipFuncSetAttribute((const void *)distk::key_hist_kernel<uint64_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) // (always the same value: concurrent callers cannot interleave badly)
            return IBVH_ERR_HIP;
        IBVH_LAUNCH((distk::key_hist_kernel<uint64_t>), dim3(blocks), dim3(distk::HIST_TPB), smem, st, (const uint64_t *)keys, n,
                    shift, bits, prefix_shift, pre, nprefix, (uint32_t *)out);
    }
    return hipGetLastError() == hipSuccess ? IBVH_OK : IBVH_ERR_HIP;
}

ibvh_status ibvh_pack_records(const ibvh_types *types, const void *volumes, const void *keys, const void *perm,
                              int64_t index_base, int64_t n, void *records_out, void *stream) {
    if (!types || n < 0) return IBVH_ERR_INVALID_ARG;
    if (n == 0) return IBVH_OK;
    if (!volumes || !keys || !records_out) return IBVH_ERR_INVALID_ARG;
    ibvh_layout lay;
    LeafLayout dl;
    if (!layout_of(*types, lay, &dl)) return IBVH_ERR_UNSUPPORTED;
    int64_t b = ceil_div(n, 256);
    unsigned blocks = (unsigned)(b > 4096 ? 4096 : b);
    hipStream_t st = (hipStream_t)stream;
    return (ibvh_status)dispatch_volume(types->leaf_kind, types->leaf_float, [&](auto vt) -> int {
        using V = typename decltype(vt)::type;
        return dispatch_index(types->index_type, [&](auto it) -> int {
            using I = typename decltype(it)::type;
            if (types->morton_type == IBVH_U64)
                IBVH_LAUNCH((distk::pack_kernel<V, I, uint64_t>), dim3(blocks), dim3(256), 0, st, (const V *)volumes,
                            (const uint64_t *)keys, (const uint32_t *)perm, index_base, n, dl, (char *)records_out);
            else
                IBVH_LAUNCH((distk::pack_kernel<V, I, uint32_t>), dim3(blocks), dim3(256), 0, st, (const V *)volumes,
                            (const uint32_t *)keys, (const uint32_t *)perm, index_base, n, dl, (char *)records_out);
            IBVH_LAUNCH_CHECK();
            return (int)IBVH_OK;
        });
    });
}

} // extern "C"
