// ibvh_radix.hpp — wave64 radix-sort building blocks shared by ibvh_sort.hip (LSD passes over (key, position)
// pairs) and ibvh_msd.hip (MSD partition of whole records + in-LDS bucket finish).  gfx950 only.
//
// Everything here is stable: ranks are handed out in (row, lane) = memory order, so equal keys keep their input
// order — the single-task Base.sort! behaviour the oracle restates for AK.sort! (reference src/build.jl:248-253;
// tie order is unpinned there).
#pragma once
#include <type_traits>
#include "ibvh_common.hpp"

#ifndef IBVH_PASS_STAMP
#define IBVH_PASS_STAMP(k) // (diagnostic builds of ibvh_msd.hip define it)
#endif

namespace ibvh {
namespace rsort {

// Workgroup barrier that orders LDS accesses only.  __syncthreads() carries a workgroup-scope fence, for which the
// compiler drains EVERY outstanding memory operation (s_waitcnt vmcnt(0)) — including global loads issued on purpose
// long before their use.  Where only LDS contents are handed from wave to wave, this barrier waits for the LDS
// operations alone and such loads stay in flight across it.
IBVH_D void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <bool LDS_ONLY> IBVH_D void wg_barrier() {
    if constexpr (LDS_ONLY) lds_barrier();
    else __syncthreads();
}

template <int TPB, bool LDS_ONLY = false> IBVH_D uint32_t block_exclusive_scan(uint32_t v, uint32_t *wave_tot /* TPB/64 */, uint32_t *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[w] = inc;
    wg_barrier<LDS_ONLY>();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < TPB / 64; ++i) {
        uint32_t t = wave_tot[i];
        if (i < w) base += t;
        tot += t;
    }
    if (total) *total = tot;
    wg_barrier<LDS_ONLY>();
    return base + inc - v;
}

// in-place exclusive scan of an LDS array of `count` values by the whole workgroup; returns the total
template <int TPB, bool LDS_ONLY = false> IBVH_D uint32_t lds_exclusive_scan(uint32_t *arr, int count, uint32_t *wave_tot) {
    const int per = (count + TPB - 1) / TPB;
    const int lo = threadIdx.x * per;
    uint32_t sum = 0;
    for (int k = 0; k < per; ++k)
        if (lo + k < count) sum += arr[lo + k];
    uint32_t total;
    uint32_t run = block_exclusive_scan<TPB, LDS_ONLY>(sum, wave_tot, &total);
    for (int k = 0; k < per; ++k)
        if (lo + k < count) {
            const uint32_t v = arr[lo + k];
            arr[lo + k] = run;
            run += v;
        }
    wg_barrier<LDS_ONLY>();
    return total;
}

// wave64 "match" ranking of IPT keys per lane on a digit of `bits` bits: rank[j] = number of keys of the same
// digit that precede key j in (j, lane) order within this wave, counted through my_hist (per-wave LDS counters).
// Two loops on purpose: the ballots of all rows are independent of each other (the compiler interleaves them; one
// row's chain of `bits` dependent mask updates otherwise waits for itself: measured 1,400 cycles a row at two waves
// per SIMD), only the counter updates are a serial chain (a row reads the counters the previous row wrote).
// Per row and digit bit: one bit-field extract (0 / -1 by the lane's bit), one compare (the ballot), and per 32-lane
// half an xor (lanes whose bit differs from mine) and an or into the running mismatch mask: the ranking is VALU-issue
// bound (measured: 60 % of the bucket-finish kernel), so its inner loop is kept to ~6 vector instructions per row-bit
// and rows beyond the wave's share are skipped.
template <class K, int IPT>
IBVH_D void wave_rank(const K (&key)[IPT], int shift, uint32_t mask, int bits, uint16_t *my_hist, int lane,
                      uint16_t (&rank)[IPT], int jmax = IPT) {
    const uint64_t lt_mask = ((uint64_t)1 << lane) - 1;
    constexpr int CH = IPT < 8 ? IPT : 8; // rows whose ballots are interleaved
#pragma unroll
    for (int j0 = 0; j0 < IPT; j0 += CH) {
        if (j0 >= jmax) break; // (wave-uniform) rows beyond the wave's share hold nothing
        uint32_t dig[CH], mlo[CH], mhi[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            dig[j] = (uint32_t)(key[j0 + j] >> shift);
            mlo[j] = 0;
            mhi[j] = 0;
        }
        for (int b = 0; b < bits; ++b) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (j0 + j < jmax) { // (wave-uniform)
                    const int32_t t = __builtin_amdgcn_sbfe((int32_t)dig[j], b, 1); // 0 or -1
                    const uint64_t bal = __ballot(t != 0);
                    mlo[j] |= (uint32_t)bal ^ (uint32_t)t;          // lanes whose bit b differs from this lane's
                    mhi[j] |= (uint32_t)(bal >> 32) ^ (uint32_t)t;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (j0 + j < jmax) {
                const uint64_t peers = ~(((uint64_t)mhi[j] << 32) | mlo[j]);
                const uint32_t d = dig[j] & mask;
                const uint32_t prev = my_hist[d];
                rank[j0 + j] = (uint16_t)(prev + (uint32_t)__popcll(peers & lt_mask));
                if ((peers & lt_mask) == 0) my_hist[d] = (uint16_t)(prev + (uint32_t)__popcll(peers));
            }
        }
    }
}

// in-place exclusive scans of TWO LDS arrays of `count` values each in one sweep (shared barriers); wave_tot: 2 * TPB/64
template <int TPB, bool LDS_ONLY = false> IBVH_D void lds_exclusive_scan_pair(uint32_t *a, uint32_t *b, int count, uint32_t *wave_tot) {
    constexpr int W = TPB / 64;
    const int per = (count + TPB - 1) / TPB;
    const int lo = threadIdx.x * per;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t sa = 0, sb = 0;
    for (int k = 0; k < per; ++k)
        if (lo + k < count) {
            sa += a[lo + k];
            sb += b[lo + k];
        }
    uint32_t ia = sa, ib = sb;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t ta = __shfl_up(ia, o, 64), tb = __shfl_up(ib, o, 64);
        if (lane >= o) {
            ia += ta;
            ib += tb;
        }
    }
    if (lane == 63) {
        wave_tot[w] = ia;
        wave_tot[W + w] = ib;
    }
    wg_barrier<LDS_ONLY>();
    uint32_t ra = ia - sa, rb = ib - sb;
#pragma unroll
    for (int i = 0; i < W; ++i)
        if (i < w) {
            ra += wave_tot[i];
            rb += wave_tot[W + i];
        }
    for (int k = 0; k < per; ++k)
        if (lo + k < count) {
            const uint32_t va = a[lo + k], vb = b[lo + k];
            a[lo + k] = ra;
            b[lo + k] = rb;
            ra += va;
            rb += vb;
        }
    wg_barrier<LDS_ONLY>();
}

// grid = radix workgroups; workgroup d turns row d of tile_hist ([radix][num_tiles], digit-major) into its
// exclusive prefix over tiles and writes the row sum to digit_total[d].
template <int TPB>
__global__ __launch_bounds__(TPB) void scan_kernel(uint32_t *__restrict__ tile_hist, int num_tiles,
                                                   uint32_t *__restrict__ digit_total) {
    __shared__ uint32_t wave_tot[TPB / 64];
    uint32_t *row = tile_hist + (int64_t)blockIdx.x * num_tiles;
    uint32_t carry = 0;
    for (int base = 0; base < num_tiles; base += TPB) {
        int i = base + threadIdx.x;
        uint32_t v = i < num_tiles ? row[i] : 0u;
        uint32_t tot;
        uint32_t ex = block_exclusive_scan<TPB>(v, wave_tot, &tot);
        if (i < num_tiles) row[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = carry;
}

// How a finished BoundingVolume record is produced from a source volume / record (type-generic at run time:
// volumes move as 8-byte words, so there is no template axis for the leaf type).
struct RecordArgs {
    const char *src;      // raw volumes or BoundingVolume records
    char *dst;            // BoundingVolume records
    int64_t src_stride;
    int32_t src_wrapped;  // 1: keep the source record's .index, 0: index = position + 1
    int32_t vol_words;    // sizeof(V) / 8
    int32_t index_bytes;  // 4 or 8
    LeafLayout lay;
};

IBVH_D void write_record(const RecordArgs &rec, uint32_t p, uint64_t dest, uint64_t key) {
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
    store_morton(dp, rec.lay, key);
}

// One stable pass over the keys a workgroup holds in registers (wave-striped: (w, j, lane) order == sequence
// order) on the digit (key >> shift) & (2^bits - 1), bits <= RBITS: afterwards s_keys / s_vals hold the sequence
// in the new order.  whist: W * 2^RBITS 16-bit counters, local_base: 2^RBITS words, wave_tot: TPB/64 words.
struct NoVal {}; // VT of a pass that moves keys only (the payload is packed into the key's low bits)
template <class K, class VT, int TPB, int IPT, int RBITS>
IBVH_D void lds_radix_pass(const K (&key)[IPT], const VT (&val)[IPT], int shift, int bits, int jmax, K *s_keys, VT *s_vals,
                           uint32_t *local_base, uint32_t *wave_tot, uint16_t *whist, uint32_t *tot_d_out = nullptr) {
    constexpr int W = TPB / 64;
    constexpr int R = 1 << RBITS;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t mask = (1u << bits) - 1u;
    uint16_t *my_hist = whist + w * R;
    IBVH_PASS_STAMP(5);
    // (every wave zeroes ITS OWN counters — nobody else touches them before the barrier behind the ranking, and a wave's LDS
    // operations execute in order — so no workgroup barrier, and no wait for the slowest wave, between zeroing and ranking)
    for (int i = lane; i < R / 2; i += 64) ((uint32_t *)my_hist)[i] = 0;
    __builtin_amdgcn_wave_barrier();
    IBVH_PASS_STAMP(6);
    uint16_t rank[IPT];
    wave_rank<K, IPT>(key, shift, mask, bits, my_hist, lane, rank, jmax);
    __syncthreads();
    IBVH_PASS_STAMP(7);
    uint32_t tot_d = 0;
    for (int d0 = 0; d0 < R; d0 += TPB) { // (R <= TPB in every instantiation: one trip)
        const int d = d0 + threadIdx.x;
        tot_d = 0;
        if (d < R) {
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
        if (d < R) local_base[d] = lb;
    }
    if (tot_d_out) *tot_d_out = tot_d;
    __syncthreads();
    IBVH_PASS_STAMP(8);
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        if (j >= jmax) break;
        const uint32_t d = (uint32_t)(key[j] >> shift) & mask;
        const uint32_t pos = local_base[d] + my_hist[d] + rank[j];
        s_keys[pos] = key[j];
        if constexpr (!std::is_same<VT, NoVal>::value) s_vals[pos] = val[j];
    }
    __syncthreads();
    IBVH_PASS_STAMP(9);
}

} // namespace rsort
} // namespace ibvh
