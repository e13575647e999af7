// ibvh_sort.hip — stable LSB radix sort of (Morton key, uint32 position) pairs for gfx950.
//
// Replaces AK.sort!(leaves, by = bv -> bv.morton) (reference src/build.jl:248-253).  The
// reference sorts whole 24-byte records with a third-party merge sort; here only (key, position)
// pairs move through the passes and the records are gathered once at the end (ibvh_build.hip).
//
// One pass (8-bit digit) = three launches, no inter-workgroup communication inside a launch:
//   hist    : per-tile digit histogram, LDS-staged (ds_add_u32), written digit-major
//   scan    : one workgroup per digit: exclusive scan of that digit's row over tiles + digit total
//   scatter : per tile: wave64 ballot "match" ranking (stable), LDS-staged per-wave histograms,
//             tile reorder in LDS, then coalesced runs written to global.
// Stability: ranks are assigned in tile order = memory order, so equal keys keep their order;
// that is the single-task Base.sort! behaviour the oracle restates (tie order is unpinned in the
// reference).
#include <cstdlib>

#include "ibvh_common.hpp"
#include "ibvh_radix.hpp"

namespace ibvh {
namespace rsort {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;

// ---- hist ---------------------------------------------------------------------------------
template <class K, int TPB, int IPT>
__global__ __launch_bounds__(TPB) void hist_kernel(const K *__restrict__ keys, int64_t n, int shift, uint32_t mask,
                                                   uint32_t *__restrict__ tile_hist, int num_tiles) {
    // one private copy of the histogram per wave: the LDS atomics of different waves never collide
    constexpr int W = TPB / 64;
    __shared__ uint32_t h[W][RADIX];
    for (int i = threadIdx.x; i < W * RADIX; i += TPB) (&h[0][0])[i] = 0;
    __syncthreads();
    const int tile = blockIdx.x;
    const int64_t base = (int64_t)tile * (TPB * IPT);
    uint32_t *mine = h[threadIdx.x >> 6];
    K key[IPT];
#pragma unroll
    for (int j = 0; j < IPT; ++j) { // all loads first, then the atomics
        const int64_t i = base + j * TPB + threadIdx.x;
        key[j] = i < n ? keys[i] : (K)0;
    }
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int64_t i = base + j * TPB + threadIdx.x;
        if (i < n) atomicAdd(&mine[(uint32_t)(key[j] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < RADIX; d += TPB) {
        uint32_t s = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) s += h[w][d];
        tile_hist[(int64_t)d * num_tiles + tile] = s;
    }
}

// ---- scatter ------------------------------------------------------------------------------
// RECORDS = true (last pass of a BVH build): instead of writing (key, position) and gathering later, the
// pass fetches the source volume at `position` and writes the finished BoundingVolume record straight to
// its sorted place — the separate gather kernel and one (key, position) round trip through HBM disappear.
// Type-generic at run time (volumes move as 8-byte words), so no extra template axis.
// Occupancy the LDS budget allows, pinned so that co-compiled instantiations cannot push the VGPR count over a
// waves-per-SIMD step (observed: 128 -> 132 VGPRs, 4 -> 3 waves/SIMD, scatter 48 -> 66 us at 1e7 keys).
constexpr int scatter_min_waves(int tpb, int ipt, int key_bytes) {
    return key_bytes == 8 ? ((tpb * ipt <= 2048) ? 4 : 2) : ((tpb * ipt <= 2048) ? 6 : 4);
}

template <class K, int TPB, int IPT, bool RECORDS>
__global__ __launch_bounds__(TPB, scatter_min_waves(TPB, IPT, sizeof(K))) void scatter_kernel(const K *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                      K *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                      int64_t n, int shift, uint32_t mask,
                                                      const uint32_t *__restrict__ tile_hist,
                                                      const uint32_t *__restrict__ digit_total, int num_tiles,
                                                      RecordArgs rec) {
    constexpr int W = TPB / 64;
    constexpr int TILE = TPB * IPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    K *s_keys = (K *)smem;                                   // TILE
    uint32_t *s_vals = (uint32_t *)(s_keys + TILE);          // TILE
    uint32_t *whist = s_vals + TILE;                         // W * RADIX
    uint32_t *local_base = whist + W * RADIX;                // RADIX
    uint32_t *delta = local_base + RADIX;                    // RADIX
    uint32_t *wave_tot = delta + RADIX;                      // W

    const int tile = xcd_remap(blockIdx.x, num_tiles);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t tile_base = (int64_t)tile * TILE;
    const int64_t wave_base = tile_base + (int64_t)w * (64 * IPT);
    const int valid = (int)((n - tile_base) < (int64_t)TILE ? (n - tile_base) : (int64_t)TILE);

    for (int i = threadIdx.x; i < W * RADIX; i += TPB) whist[i] = 0;

    // phase A: coalesced loads, wave-striped so that (w, j, lane) order == memory order
    K key[IPT];
    uint32_t val[IPT];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        int64_t i = wave_base + j * 64 + lane;
        bool ok = i < n;
        key[j] = ok ? keys_in[i] : (K)~(K)0;
        val[j] = ok ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;
    }
    __syncthreads();

    // phase B: stable rank inside the wave by ballot matching, running per-wave digit counters in LDS
    uint16_t rank[IPT];
    uint32_t *my_hist = whist + w * RADIX;
    const uint64_t lt_mask = ((uint64_t)1 << lane) - 1;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        uint32_t d = (uint32_t)(key[j] >> shift) & mask;
        uint64_t peers = ~(uint64_t)0;
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            bool bit = (d >> b) & 1u;
            uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        uint32_t prev = my_hist[d];
        rank[j] = (uint16_t)(prev + (uint32_t)__popcll(peers & lt_mask));
        if ((peers & lt_mask) == 0) my_hist[d] = prev + (uint32_t)__popcll(peers); // lowest peer lane writes
    }
    __syncthreads();

    // phase C: per digit, exclusive prefix over waves; tile-local digit bases; global deltas
    uint32_t tot_d = 0, gtot = 0;
    if (threadIdx.x < RADIX) {
        const int d = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < W; ++i) {
            uint32_t c = whist[i * RADIX + d];
            whist[i * RADIX + d] = run;
            run += c;
        }
        tot_d = run;
        gtot = digit_total[d];
    }
    uint32_t lb = block_exclusive_scan<TPB>(tot_d, wave_tot, nullptr);
    uint32_t gb = block_exclusive_scan<TPB>(gtot, wave_tot, nullptr);
    if (threadIdx.x < RADIX) {
        const int d = threadIdx.x;
        local_base[d] = lb;
        delta[d] = gb + tile_hist[(int64_t)d * num_tiles + tile] - lb; // mod 2^32
    }
    __syncthreads();

    // phase D: tile reorder in LDS
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        uint32_t d = (uint32_t)(key[j] >> shift) & mask;
        uint32_t pos = local_base[d] + my_hist[d] + rank[j];
        s_keys[pos] = key[j];
        s_vals[pos] = val[j];
    }
    __syncthreads();

    // phase E: coalesced runs to global
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        int pos = k * TPB + threadIdx.x;
        if (pos < valid) {
            K kk = s_keys[pos];
            uint32_t d = (uint32_t)(kk >> shift) & mask;
            uint32_t dest = (uint32_t)pos + delta[d];
            if constexpr (RECORDS) {
                const uint32_t p = s_vals[pos];
                const char *sp = rec.src + (int64_t)p * rec.src_stride;
                char *dp = rec.dst + (int64_t)dest * rec.lay.stride;
                const uint64_t *sw = (const uint64_t *)sp;
                uint64_t *dw = (uint64_t *)dp;
#pragma unroll
                for (int wd = 0; wd < 6; ++wd)
                    if (wd < rec.vol_words) dw[wd] = sw[wd];
                if (rec.index_bytes == 4)
                    *(int32_t *)(dp + rec.lay.index_off) = rec.src_wrapped ? *(const int32_t *)(sp + rec.lay.index_off) : (int32_t)(p + 1u);
                else
                    *(int64_t *)(dp + rec.lay.index_off) = rec.src_wrapped ? *(const int64_t *)(sp + rec.lay.index_off) : (int64_t)p + 1;
                store_morton(dp, rec.lay, (uint64_t)kk);
            } else {
                keys_out[dest] = kk;
                vals_out[dest] = s_vals[pos];
            }
        }
    }
}

// =============================================================================================
// MSD + in-LDS hybrid (the default for up to ~6 M keys; larger inputs use the LSD passes above).
//
// LSD needs ceil(bits/8) full passes over the (key, position) pairs.  Morton codes of a point cloud
// spread over their top bits, so ONE stable most-significant-digit partition (up to 11 bits = 2048
// buckets) leaves buckets of a few thousand keys that one workgroup sorts ENTIRELY in LDS on the
// remaining bits (stable LSD passes of 8 bits, LDS to LDS) and then writes out — as finished
// BoundingVolume records when called from the build.  HBM traffic per key drops from
// 4 + 16*4 + (8+16+24) B to 8+8 (partition) + 8 + 16 + 24 (bucket sort + record).
// Stability: the partition is stable and the in-LDS passes are stable, so equal keys keep input order.
// A bucket larger than the workgroup's LDS capacity (clustered input) is sorted by the same workgroup
// with a tiled LSD between the two global buffers: slower, never wrong.
// =============================================================================================
constexpr int MSD_MAX_BITS = 11;

// per-tile histogram of a digit of up to MSD_MAX_BITS bits (the build fuses this into the key encoder)
template <class K, int TPB, int IPT>
__global__ __launch_bounds__(TPB) void hist_wide_kernel(const K *__restrict__ keys, int64_t n, int shift, int bits,
                                                        uint32_t *__restrict__ tile_hist, int num_tiles) {
    extern __shared__ uint32_t hw[];
    const int radix = 1 << bits;
    for (int i = threadIdx.x; i < radix; i += TPB) hw[i] = 0;
    __syncthreads();
    const uint32_t mask = (uint32_t)radix - 1u;
    const int64_t base = (int64_t)blockIdx.x * (TPB * IPT);
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int64_t i = base + j * TPB + threadIdx.x;
        if (i < n) atomicAdd(&hw[(uint32_t)(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < radix; d += TPB) tile_hist[(int64_t)d * num_tiles + blockIdx.x] = hw[d];
}

// stable partition of (key, position) by the digit (key >> shift) of `bits` <= 11 bits; positions are implicit
template <class K, int TPB, int IPT>
__global__ __launch_bounds__(TPB) void scatter_wide_kernel(const K *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                           K *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                           int64_t n, int shift, int bits,
                                                           const uint32_t *__restrict__ tile_hist,
                                                           const uint32_t *__restrict__ digit_total, int num_tiles) {
    constexpr int W = TPB / 64;
    constexpr int TILE = TPB * IPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int radix = 1 << bits;
    const uint32_t mask = (uint32_t)radix - 1u;
    K *s_keys = (K *)smem;                              // TILE
    uint32_t *s_vals = (uint32_t *)(s_keys + TILE);     // TILE
    uint32_t *local_base = s_vals + TILE;               // radix
    uint32_t *delta = local_base + radix;               // radix
    uint32_t *wave_tot = delta + radix;                 // 16
    uint16_t *whist = (uint16_t *)(wave_tot + 16);      // W * radix

    const int tile = xcd_remap(blockIdx.x, num_tiles);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t tile_base = (int64_t)tile * TILE;
    const int64_t wave_base = tile_base + (int64_t)w * (64 * IPT);
    const int valid = (int)((n - tile_base) < (int64_t)TILE ? (n - tile_base) : (int64_t)TILE);

    for (int i = threadIdx.x; i < W * radix; i += TPB) whist[i] = 0;
    K key[IPT];
    uint32_t val[IPT];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int64_t i = wave_base + j * 64 + lane;
        const bool ok = i < n;
        key[j] = ok ? keys_in[i] : (K) ~(K)0;
        val[j] = (ok && vals_in) ? vals_in[i] : (uint32_t)i;
    }
    __syncthreads();
    uint16_t rank[IPT];
    uint16_t *my_hist = whist + w * radix;
    wave_rank<K, IPT>(key, shift, mask, bits, my_hist, lane, rank);
    __syncthreads();
    for (int d = threadIdx.x; d < radix; d += TPB) {
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const uint32_t c = whist[i * radix + d];
            whist[i * radix + d] = (uint16_t)run;
            run += c;
        }
        local_base[d] = run;
        delta[d] = digit_total[d];
    }
    __syncthreads();
    lds_exclusive_scan<TPB>(local_base, radix, wave_tot);
    lds_exclusive_scan<TPB>(delta, radix, wave_tot);
    for (int d = threadIdx.x; d < radix; d += TPB)
        delta[d] = delta[d] + tile_hist[(int64_t)d * num_tiles + tile] - local_base[d]; // mod 2^32
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const uint32_t d = (uint32_t)(key[j] >> shift) & mask;
        const uint32_t pos = local_base[d] + my_hist[d] + rank[j];
        s_keys[pos] = key[j];
        s_vals[pos] = val[j];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int pos = k * TPB + threadIdx.x;
        if (pos < valid) {
            const K kk = s_keys[pos];
            const uint32_t d = (uint32_t)(kk >> shift) & mask;
            const uint32_t dest = (uint32_t)pos + delta[d];
            keys_out[dest] = kk;
            vals_out[dest] = s_vals[pos];
        }
    }
}
template <class K, int TPB, int IPT> inline size_t scatter_wide_smem(int bits) {
    return (size_t)TPB * IPT * (sizeof(K) + 4) + ((size_t)2 << bits) * 4 + 64 + (size_t)(TPB / 64) * ((size_t)1 << bits) * 2 + 64;
}

// one workgroup per bucket: sort the bucket's pairs on the low `low_bits` bits and write them out
// (pairs to kout/vout, or finished records when RECORDS)
template <class K, int TPB, int IPT, bool RECORDS>
__global__ __launch_bounds__(TPB) void bucket_sort_kernel(K *__restrict__ kalt, uint32_t *__restrict__ valt,
                                                          K *__restrict__ kpri, uint32_t *__restrict__ vpri,
                                                          const uint32_t *__restrict__ bucket_start /* digit totals */, int low_bits,
                                                          RecordArgs rec) {
    constexpr int W = TPB / 64;
    constexpr int CAP = TPB * IPT;
    constexpr int R = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
    K *s_keys = (K *)bsm;                                 // CAP
    uint32_t *s_vals = (uint32_t *)(s_keys + CAP);        // CAP
    uint32_t *local_base = s_vals + CAP;                  // R
    uint32_t *gbase = local_base + R;                     // R
    uint32_t *wave_tot = gbase + R;                       // 16
    uint16_t *whist = (uint16_t *)(wave_tot + 16);        // W * R
    // the bucket's range: every workgroup sums the digit totals in front of its own digit (<= 2047 values) instead of
    // waiting for a one-workgroup prefix-sum launch
    const int64_t m = (int64_t)bucket_start[blockIdx.x]; // (digit totals, not yet scanned)
    if (m == 0) return;
    uint32_t before = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += TPB) before += bucket_start[i];
    uint32_t start32 = 0;
    block_exclusive_scan<TPB>(before, wave_tot, &start32);
    const int64_t start = (int64_t)start32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint16_t *my_hist = whist + w * R;
    const int passes = (low_bits + 7) / 8;

    // one stable 8-bit pass over the CAP keys held in registers (wave-striped); result in s_keys / s_vals
    auto lds_pass = [&](K(&key)[IPT], uint32_t(&val)[IPT], int shift, int bits, uint32_t &tot_d, int jmax) {
        const uint32_t mask = (1u << bits) - 1u;
        for (int i = threadIdx.x; i < W * R; i += TPB) whist[i] = 0;
        __syncthreads();
        uint16_t rank[IPT];
        wave_rank<K, IPT>(key, shift, mask, bits, my_hist, lane, rank, jmax);
        __syncthreads();
        tot_d = 0;
        if (threadIdx.x < R) {
            const int d = threadIdx.x;
            uint32_t run = 0;
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const uint32_t c = whist[i * R + d];
                whist[i * R + d] = (uint16_t)run;
                run += c;
            }
            tot_d = run;
        }
        const uint32_t lb = block_exclusive_scan<TPB>(tot_d, wave_tot, nullptr);
        if (threadIdx.x < R) local_base[threadIdx.x] = lb;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            if (j >= jmax) break;
            const uint32_t d = (uint32_t)(key[j] >> shift) & mask;
            const uint32_t pos = local_base[d] + my_hist[d] + rank[j];
            s_keys[pos] = key[j];
            s_vals[pos] = val[j];
        }
        __syncthreads();
    };

    if (m <= CAP) {
        // ---- fast path: the whole bucket lives in LDS --------------------------------------------
        // The m keys are dealt to the waves in equal contiguous shares of `chunk` (a multiple of 64), so that a
        // bucket of CAP/4 keys keeps all waves busy with a quarter of the ranking work each instead of leaving
        // it to wave 0 while the others rank padding.  (w, j, lane) order is still memory order.
        const int chunk = (int)((m + W * 64 - 1) / (W * 64)) * 64;
        const int jmax = chunk / 64; // <= IPT
        K key[IPT];
        uint32_t val[IPT];
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int idx = w * chunk + j * 64 + lane;
            const bool ok = j < jmax && idx < m;
            key[j] = ok ? kalt[start + idx] : (K) ~(K)0; // sentinels: maximal digit every pass, last in order
            val[j] = ok ? valt[start + idx] : 0u;
        }
        if (passes == 0) {
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const int idx = w * chunk + j * 64 + lane;
                if (j < jmax) {
                    s_keys[idx] = key[j];
                    s_vals[idx] = val[j];
                }
            }
            __syncthreads();
        }
        for (int p = 0; p < passes; ++p) {
            const int shift = 8 * p;
            const int bits = low_bits - shift < 8 ? low_bits - shift : 8;
            uint32_t tot_d;
            lds_pass(key, val, shift, bits, tot_d, jmax);
            if (p + 1 < passes) {
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    const int idx = w * chunk + j * 64 + lane;
                    if (j < jmax) {
                        key[j] = s_keys[idx];
                        val[j] = s_vals[idx];
                    }
                }
                __syncthreads();
            }
        }
        for (int pos = threadIdx.x; pos < m; pos += TPB) {
            const K kk = s_keys[pos];
            const uint32_t p = s_vals[pos];
            if constexpr (RECORDS) {
                write_record(rec, p, (uint64_t)(start + pos), (uint64_t)kk);
            } else {
                kpri[start + pos] = kk;
                vpri[start + pos] = p;
            }
        }
        return;
    }

    // ---- slow path: bucket larger than the LDS capacity; tiled LSD between the two global buffers ----
    K *src_k = kalt, *dst_k = kpri;
    uint32_t *src_v = valt, *dst_v = vpri;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        const int bits = low_bits - shift < 8 ? low_bits - shift : 8;
        const uint32_t mask = (1u << bits) - 1u;
        const bool last = p + 1 == passes;
        if (threadIdx.x < R) gbase[threadIdx.x] = 0;
        __syncthreads();
        for (int64_t i = threadIdx.x; i < m; i += TPB) atomicAdd(&gbase[(uint32_t)(src_k[start + i] >> shift) & mask], 1u);
        __syncthreads();
        lds_exclusive_scan<TPB>(gbase, R, wave_tot);
        for (int64_t t0 = 0; t0 < m; t0 += CAP) {
            const int64_t cnt = m - t0 < CAP ? m - t0 : CAP;
            K key[IPT];
            uint32_t val[IPT];
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const int idx = w * 64 * IPT + j * 64 + lane;
                const bool ok = idx < cnt;
                key[j] = ok ? src_k[start + t0 + idx] : (K) ~(K)0;
                val[j] = ok ? src_v[start + t0 + idx] : 0u;
            }
            uint32_t tot_d;
            lds_pass(key, val, shift, bits, tot_d, IPT);
            for (int pos = threadIdx.x; pos < cnt; pos += TPB) {
                const K kk = s_keys[pos];
                const uint32_t d = (uint32_t)(kk >> shift) & mask;
                const uint64_t dest = (uint64_t)start + gbase[d] + ((uint32_t)pos - local_base[d]);
                if (RECORDS && last) {
                    write_record(rec, s_vals[pos], dest, (uint64_t)kk);
                } else {
                    dst_k[dest] = kk;
                    dst_v[dest] = s_vals[pos];
                }
            }
            __syncthreads();
            // sentinels of a partial tile were counted in the last digit: real count there = cnt - local_base
            if (threadIdx.x < R) {
                uint32_t real = tot_d;
                if ((int)threadIdx.x == (int)mask && cnt < CAP) real = (uint32_t)cnt - local_base[mask];
                gbase[threadIdx.x] += real;
            }
            __syncthreads();
        }
        // make this pass's global writes visible to the next pass's reads (same workgroup, other lanes)
        __threadfence_block();
        __syncthreads();
        K *tk = src_k;
        src_k = dst_k;
        dst_k = tk;
        uint32_t *tv = src_v;
        src_v = dst_v;
        dst_v = tv;
    }
    // after the loop `src` holds the sorted pairs (unless records were written by the last pass)
    if (passes == 0 || !RECORDS) {
        if (RECORDS) {
            for (int64_t i = threadIdx.x; i < m; i += TPB) write_record(rec, src_v[start + i], (uint64_t)(start + i), (uint64_t)src_k[start + i]);
        } else if (src_k != kpri) {
            for (int64_t i = threadIdx.x; i < m; i += TPB) {
                kpri[start + i] = src_k[start + i];
                vpri[start + i] = src_v[start + i];
            }
        }
    }
}

template <class K, int TPB, int IPT> constexpr size_t bucket_smem() {
    return (size_t)TPB * IPT * (sizeof(K) + 4) + 2 * 256 * 4 + 64 + (size_t)(TPB / 64) * 256 * 2 + 64;
}
template <class K, int TPB, int IPT> constexpr size_t scatter_smem() {
    return (size_t)TPB * IPT * (sizeof(K) + 4) + (size_t)(TPB / 64) * RADIX * 4 + RADIX * 8 + (TPB / 64) * 4 + 64;
}

struct Geometry {
    int tpb, ipt;
    int tile() const { return tpb * ipt; }
};
struct FirstPassPlan {
    int tpb, ipt, num_tiles;
    uint32_t *tile_hist;
    uint32_t mask;
    int shift, bits; // the digit the first pass sorts on: (key >> shift) & mask
};
// Small inputs get small tiles so the grid still covers the 256 CUs; large inputs get 8192-element
// tiles so that one digit's run in a tile is >= 128 B on average.
inline Geometry choose_geometry(int64_t n) {
    const int forced = g_tuning.sort_tile; // 2048, 4096, 8192 or 16384 keys per tile
    switch (forced) {
    case 2048: return Geometry{256, 8};
    case 4096: return Geometry{256, 16};
    case 8192: return Geometry{512, 16};
    case 16384: return Geometry{1024, 16};
    }
    return n >= (int64_t(1) << 22) ? Geometry{512, 16} : Geometry{256, 8};
}

// MSD + in-LDS hybrid: which digit width and which bucket-kernel capacity, or {0, 0} for plain LSD
struct MsdPlan {
    int bits;     // MSD digit width (0 = do not use the hybrid)
    int capacity; // keys a bucket workgroup sorts in LDS: 2048, 4096 or 8192
    int btpb;     // threads of the bucket workgroup (capacity / btpb keys per thread)
};
inline MsdPlan choose_msd(int64_t n, int key_bits, int key_bytes) {
    const int mode = g_tuning.sort_lsd ? 1 : 0;
    if (mode == 1 || n < 2048 || key_bits <= 8) return {0, 0, 0};
    const int msd_avg_max = g_tuning.sort_msd_avg;
    int bits = 1;
    while (bits < MSD_MAX_BITS && bits < key_bits && (n >> bits) > msd_avg_max) ++bits;
    const int64_t avg = n >> bits;
    // Measured on MI355X (round 1): the hybrid beats plain LSD while buckets fit 4096-key workgroups
    // (Morton+sort phase 0.366 vs 0.422 ms at 6e6 leaves, 0.109 vs 0.139 ms at 1e6) and loses with 8192-key
    // workgroups (0.652 vs 0.589 ms at 1e7: the bucket kernel is barrier-bound), so larger inputs keep LSD.
    const int cap_max = 4096;
    int cap = 2048;
    while (cap < cap_max && avg * 4 > cap * 3) cap *= 2; // average bucket <= 3/4 of the capacity
    if (avg * 4 > (int64_t)cap * 3) return {0, 0, 0};   // too many keys for one partition level: LSD
    const int forced_tpb = g_tuning.bucket_tpb;
    int btpb = cap == 2048 ? 256 : (cap == 4096 ? 512 : 1024);
    if (forced_tpb == 256 || forced_tpb == 512 || forced_tpb == 1024) btpb = forced_tpb;
    if (cap / btpb < 8) btpb = cap / 8;
    if (cap / btpb > 32) btpb = cap / 32;
    return {bits, cap, btpb};
}

size_t scratch_bytes(int64_t n) {
    int64_t tiles = ceil_div(n, 256 * 8); // upper bound over all geometries
    // tile histograms for the widest digit (MSD partition), digit totals, bucket starts
    return (size_t)align_up(((int64_t)1 << MSD_MAX_BITS) * tiles * 4, 256) + (((size_t)1 << MSD_MAX_BITS) + 8) * 4 * 2 + 512;
}

template <class K, int TPB, int IPT>
int run_passes(K *keys, uint32_t *vals, K *keys_alt, uint32_t *vals_alt, int64_t n, int key_bits, bool vals_implicit,
               int32_t *result_in_alt, void *scratch, hipStream_t st, bool first_hist_done, const RecordArgs *records) {
    const int num_tiles = (int)ceil_div(n, TPB * IPT);
    uint32_t *tile_hist = (uint32_t *)scratch;
    uint32_t *digit_total = (uint32_t *)((char *)scratch + align_up((int64_t)RADIX * num_tiles * 4, 256));
    constexpr size_t smem = scatter_smem<K, TPB, IPT>();
    // (per call, not once per process: the attribute is per DEVICE, and a host may drive several GPUs from one process)
    IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)scatter_kernel<K, TPB, IPT, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)scatter_kernel<K, TPB, IPT, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    K *kin = keys, *kout = keys_alt;
    uint32_t *vin = vals, *vout = vals_alt;
    int flips = 0;
    for (int shift = 0; shift < key_bits; shift += RADIX_BITS) {
        int bits = key_bits - shift < RADIX_BITS ? key_bits - shift : RADIX_BITS;
        uint32_t mask = (1u << bits) - 1u;
        if (!(shift == 0 && first_hist_done))
            IBVH_LAUNCH((hist_kernel<K, TPB, IPT>), dim3(num_tiles), dim3(TPB), 0, st, kin, n, shift, mask, tile_hist,
                        num_tiles);
        IBVH_LAUNCH((scan_kernel<256>), dim3(RADIX), dim3(256), 0, st, tile_hist, num_tiles, digit_total);
        const uint32_t *vsrc = (shift == 0 && vals_implicit) ? (const uint32_t *)nullptr : vin;
        const bool last = shift + RADIX_BITS >= key_bits;
        if (last && records)
            IBVH_LAUNCH((scatter_kernel<K, TPB, IPT, true>), dim3(num_tiles), dim3(TPB), smem, st, kin, vsrc, kout, vout, n,
                        shift, mask, tile_hist, digit_total, num_tiles, *records);
        else
            IBVH_LAUNCH((scatter_kernel<K, TPB, IPT, false>), dim3(num_tiles), dim3(TPB), smem, st, kin, vsrc, kout, vout, n,
                        shift, mask, tile_hist, digit_total, num_tiles, RecordArgs{});
        IBVH_LAUNCH_CHECK();
        K *tk = kin;
        kin = kout;
        kout = tk;
        uint32_t *tv = vin;
        vin = vout;
        vout = tv;
        ++flips;
    }
    *result_in_alt = flips & 1;
    return IBVH_OK;
}

template <class K, int TPB, int IPT, int BT, int BI>
int run_msd(K *keys, uint32_t *vals, K *keys_alt, uint32_t *vals_alt, int64_t n, int key_bits, bool vals_implicit,
            int32_t *result_in_alt, void *scratch, hipStream_t st, bool first_hist_done, const RecordArgs *records, int msd_bits) {
    const int num_tiles = (int)ceil_div(n, TPB * IPT);
    const int radix = 1 << msd_bits;
    const int shift = key_bits - msd_bits;
    uint32_t *tile_hist = (uint32_t *)scratch;
    uint32_t *digit_total = (uint32_t *)((char *)scratch + align_up((int64_t)radix * num_tiles * 4, 256));
    const size_t ssm = scatter_wide_smem<K, TPB, IPT>(msd_bits);
    constexpr size_t bsm = bucket_smem<K, BT, BI>();
    IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)scatter_wide_kernel<K, TPB, IPT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_wide_smem<K, TPB, IPT>(MSD_MAX_BITS)));
    IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)bucket_sort_kernel<K, BT, BI, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bsm));
    IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)bucket_sort_kernel<K, BT, BI, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bsm));
    if (!first_hist_done)
        IBVH_LAUNCH((hist_wide_kernel<K, TPB, IPT>), dim3(num_tiles), dim3(TPB), (size_t)radix * 4, st, keys, n, shift, msd_bits,
                    tile_hist, num_tiles);
    IBVH_LAUNCH((scan_kernel<256>), dim3(radix), dim3(256), 0, st, tile_hist, num_tiles, digit_total);
    IBVH_LAUNCH((scatter_wide_kernel<K, TPB, IPT>), dim3(num_tiles), dim3(TPB), ssm, st, keys,
                vals_implicit ? (const uint32_t *)nullptr : vals, keys_alt, vals_alt, n, shift, msd_bits, tile_hist, digit_total,
                num_tiles);
    if (records)
        IBVH_LAUNCH((bucket_sort_kernel<K, BT, BI, true>), dim3(radix), dim3(BT), bsm, st, keys_alt, vals_alt, keys, vals,
                    digit_total, shift, *records);
    else
        IBVH_LAUNCH((bucket_sort_kernel<K, BT, BI, false>), dim3(radix), dim3(BT), bsm, st, keys_alt, vals_alt, keys, vals,
                    digit_total, shift, RecordArgs{});
    IBVH_LAUNCH_CHECK();
    *result_in_alt = 0; // sorted pairs (when no records are requested) end in the primary buffers
    return IBVH_OK;
}

bool uses_hybrid(int64_t n, int key_bits, int key_bytes);

// vals_implicit: the values of the first pass are the element positions 0..n-1 (vals is not read).
// Where the first pass expects its per-tile histogram ([RADIX][num_tiles], digit-major) and the tile
// geometry it will use, for a producer that fuses that histogram into its own pass (ibvh_build.hip).
FirstPassPlan first_pass_plan(int64_t n, int key_bits, int key_bytes, void *scratch) {
    Geometry g = choose_geometry(n);
    if (key_bytes == 8 && g.tpb == 1024) g = Geometry{512, 16}; // same fallback as sort_pairs
    FirstPassPlan p;
    p.tpb = g.tpb;
    p.ipt = g.ipt;
    p.num_tiles = (int)ceil_div(n, g.tile());
    p.tile_hist = (uint32_t *)scratch;
    const MsdPlan mp = uses_hybrid(n, key_bits, key_bytes) ? choose_msd(n, key_bits, key_bytes) : MsdPlan{0, 0, 0};
    if (mp.bits) { // MSD partition first: histogram of the TOP digit
        p.bits = mp.bits;
        p.shift = key_bits - mp.bits;
    } else {
        p.bits = key_bits < RADIX_BITS ? key_bits : RADIX_BITS;
        p.shift = 0;
    }
    p.mask = (1u << p.bits) - 1u;
    return p;
}

bool uses_hybrid(int64_t n, int key_bits, int key_bytes) {
    if (choose_msd(n, key_bits, key_bytes).bits == 0) return false;
    Geometry g = choose_geometry(n);
    return g.tpb != 1024; // forced 16384-key tiles have no partition instantiation
}

int sort_pairs(int key_bytes, int key_bits, int64_t n, void *keys, void *vals, void *keys_alt, void *vals_alt,
               bool vals_implicit, int32_t *result_in_alt, void *scratch, size_t scratch_sz, hipStream_t st,
               bool first_hist_done, const RecordArgs *records) {
    if (n < 0 || n >= (int64_t(1) << 32) || key_bits < 1 || key_bits > key_bytes * 8) return IBVH_ERR_INVALID_ARG;
    if (scratch_sz < scratch_bytes(n)) return IBVH_ERR_SCRATCH;
    *result_in_alt = 0;
    if (n == 0) return IBVH_OK;
    Geometry g = choose_geometry(n);
    const MsdPlan mp = uses_hybrid(n, key_bits, key_bytes) ? choose_msd(n, key_bits, key_bytes) : MsdPlan{0, 0, 0};
    if (mp.bits) {
        if (key_bytes == 8 && g.tpb == 1024) g = Geometry{512, 16};
#define IBVH_MSD_CASE(K, T, P, BT, BI)                                                                                \
    if (g.tpb == T && g.ipt == P && mp.capacity == BT * BI && mp.btpb == BT)                                          \
        return run_msd<K, T, P, BT, BI>((K *)keys, (uint32_t *)vals, (K *)keys_alt, (uint32_t *)vals_alt, n, key_bits, \
                                        vals_implicit, result_in_alt, scratch, st, first_hist_done, records, mp.bits);
#define IBVH_MSD_GEOM(K, T, P)                                                                                         \
    IBVH_MSD_CASE(K, T, P, 256, 8) IBVH_MSD_CASE(K, T, P, 512, 8) IBVH_MSD_CASE(K, T, P, 1024, 8) IBVH_MSD_CASE(K, T, P, 256, 16) \
    IBVH_MSD_CASE(K, T, P, 256, 32) IBVH_MSD_CASE(K, T, P, 512, 16)
        if (key_bytes == 4) {
            IBVH_MSD_GEOM(uint32_t, 256, 8)
            IBVH_MSD_GEOM(uint32_t, 256, 16)
            IBVH_MSD_GEOM(uint32_t, 512, 16)
        } else {
            IBVH_MSD_CASE(uint64_t, 256, 8, 256, 8)
            IBVH_MSD_CASE(uint64_t, 256, 8, 512, 8)
            IBVH_MSD_CASE(uint64_t, 256, 8, 256, 16)
            IBVH_MSD_CASE(uint64_t, 256, 16, 256, 8)
            IBVH_MSD_CASE(uint64_t, 256, 16, 512, 8)
            IBVH_MSD_CASE(uint64_t, 256, 16, 256, 16)
            IBVH_MSD_CASE(uint64_t, 512, 16, 256, 8)
            IBVH_MSD_CASE(uint64_t, 512, 16, 512, 8)
            IBVH_MSD_CASE(uint64_t, 512, 16, 256, 16)
        }
#undef IBVH_MSD_GEOM
#undef IBVH_MSD_CASE
        // (forced 16384-key tiles have no partition instantiation: fall through to LSD)
    }
#define IBVH_SORT_CASE(K, T, P)                                                                                       \
    if (g.tpb == T && g.ipt == P)                                                                                     \
        return run_passes<K, T, P>((K *)keys, (uint32_t *)vals, (K *)keys_alt, (uint32_t *)vals_alt, n, key_bits,     \
                                   vals_implicit, result_in_alt, scratch, st, first_hist_done, records);
    if (key_bytes == 4) {
        IBVH_SORT_CASE(uint32_t, 256, 8)
        IBVH_SORT_CASE(uint32_t, 256, 16)
        IBVH_SORT_CASE(uint32_t, 512, 16)
        IBVH_SORT_CASE(uint32_t, 1024, 16)
    }
    if (key_bytes == 8) {
        IBVH_SORT_CASE(uint64_t, 256, 8)
        IBVH_SORT_CASE(uint64_t, 256, 16)
        IBVH_SORT_CASE(uint64_t, 512, 16)
        if (g.tpb == 1024) // 16384 x 12 B does not fit the LDS: fall back to 8192
            return run_passes<uint64_t, 512, 16>((uint64_t *)keys, (uint32_t *)vals, (uint64_t *)keys_alt, (uint32_t *)vals_alt, n,
                                                 key_bits, vals_implicit, result_in_alt, scratch, st, first_hist_done, records);
    }
#undef IBVH_SORT_CASE
    return IBVH_ERR_INVALID_ARG;
}

} // namespace rsort
} // namespace ibvh

extern "C" {

ibvh_status ibvh_sort_scratch_bytes(int32_t key_bytes, int64_t n, size_t *bytes_out) {
    if (!bytes_out || (key_bytes != 4 && key_bytes != 8) || n < 0) return IBVH_ERR_INVALID_ARG;
    *bytes_out = ibvh::rsort::scratch_bytes(n);
    return IBVH_OK;
}

ibvh_status ibvh_sort_pairs(int32_t key_bytes, int32_t key_bits, int64_t n, void *keys, void *vals, void *keys_alt,
                            void *vals_alt, int32_t *result_in_alt, void *scratch, size_t scratch_bytes, void *stream) {
    if (!result_in_alt) return IBVH_ERR_INVALID_ARG;
    if (n > 0 && (!keys || !vals || !keys_alt || !vals_alt || !scratch)) return IBVH_ERR_INVALID_ARG;
    return (ibvh_status)ibvh::rsort::sort_pairs(key_bytes, key_bits, n, keys, vals, keys_alt, vals_alt, false,
                                                result_in_alt, scratch, scratch_bytes, (hipStream_t)stream, false, nullptr);
}

} // extern "C"
