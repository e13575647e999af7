// ibvh_msd_finish.hip — the build's sort, second half: every cell of the partitioned records (and every window of sub-cells of a
// crowded cell) is sorted by one workgroup in LDS and written to its final place.  gfx950 only.  See ibvh_msd.hip for the whole
// picture (replaces AK.sort!(leaves, by = bv -> bv.morton), reference src/build.jl:248-253).
#if !defined(IBVH_PHASE_STAMPS) || defined(IBVH_MSD_SINGLE_TU) // (diagnostic builds: one translation unit, one stamp buffer)
#include "ibvh_msd_impl.hpp"

namespace ibvh {
namespace msd {

template <class K, int TPB, int IPT> struct FinishLds {
    static constexpr int W = TPB / 64, CAP = TPB * IPT, RB = 8, R = 1 << RB;
    K *s_keys;            // CAP
    uint16_t *s_idx;      // CAP
    uint32_t *s_vals32;   // CAP / 2 (slow path: tiles of CAP / 2 keys + 32-bit positions in the same bytes)
    uint32_t *local_base; // R
    uint32_t *gbase;      // R
    uint32_t *wave_tot;   // 16
    uint16_t *whist;      // W * R
    // resident_off != 0 (the resident kernel): the 16-bit positions of the plain path — which then only serves the ranges too
    // large to be resident — live in the record area, so that the record area is as large as possible
    IBVH_D explicit FinishLds(unsigned char *p, uint32_t resident_off = 0) {
        s_keys = (K *)p;
        s_idx = resident_off ? (uint16_t *)(p + resident_off) : (uint16_t *)(s_keys + CAP);
        s_vals32 = (uint32_t *)(s_keys + CAP / 2);
        local_base = resident_off ? (uint32_t *)(s_keys + CAP) : (uint32_t *)(s_idx + CAP);
        gbase = local_base + R;
        wave_tot = gbase + R;
        whist = (uint16_t *)(wave_tot + 16);
    }
};

// records [start, start + m) of `part`, whose keys are key_base + (an nbits-bit number): sorted into out[start ...)
template <class K, int TPB, int IPT, bool RES>
IBVH_D void finish_range(const FinishArgs &fa, const FinishLds<K, TPB, IPT> &l, const char *part, int64_t start, int64_t m, K key_base,
                         int nbits) {
    constexpr int W = TPB / 64;
    constexpr int CAP = TPB * IPT;
    constexpr int RB = 8, R = 1 << RB;
    static_assert(R <= TPB, "one digit counter per thread");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int stride = fa.lay.stride;
    const char *bucket = part + start * stride;
    const int passes = (nbits + RB - 1) / RB;
    IBVH_STAMP(1, 1);
    // ---- resident path (round 4): the range's RECORDS live in LDS.  The fast path below reads every record twice from
    // memory — the strided key loads pull the range's lines through L2, the gather by sorted position fetches them again
    // (and with 512 cells of ~117 KB in flight the second read often misses: FETCH_SIZE showed the records read ~2.8 x) —
    // here each record is read from memory exactly ONCE, coalesced, into LDS; the keys are taken from the LDS copy, sorted
    // as one 32-bit word (key - base) << IDXB | position (a stable LSD on the key bits only: the position rides along), and
    // the output is gathered out of LDS.  32-bit keys whose varying bits + IDXB fit 32 bits only.
    if constexpr (RES && sizeof(K) == 4) {
        constexpr int IDXB = 32 - __builtin_clz((unsigned)(CAP - 1)); // bits of a position inside the range
        if (m <= CAP && nbits + IDXB <= 32 && (uint32_t)m * fa.words <= fa.resident_words) {
            uint64_t *s_rec = (uint64_t *)((unsigned char *)l.s_keys + fa.resident_off);
            const uint64_t *__restrict__ src = (const uint64_t *)bucket;
            const uint32_t total = (uint32_t)m * fa.words;
            constexpr int U = 8;
            for (uint32_t g0 = threadIdx.x; g0 < total; g0 += TPB * U) {
                uint64_t v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t g = g0 + u * TPB;
                    v[u] = src[g < total ? g : 0u];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t g = g0 + u * TPB;
                    if (g < total) s_rec[g] = v[u];
                }
            }
            __syncthreads();
            IBVH_STAMP(1, 2);
            const int chunk = (int)((m + W * 64 - 1) / (W * 64)) * 64;
            const int jmax = chunk / 64; // <= IPT
            const uint32_t mw = (uint32_t)fa.lay.morton_off >> 3, msh = ((uint32_t)fa.lay.morton_off & 7u) * 8u;
            const uint32_t mmask = fa.lay.morton_bytes == 2 ? 0xffffu : 0xffffffffu;
            K key[IPT];
            const rsort::NoVal none[IPT] = {};
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const int idx = w * chunk + j * 64 + lane;
                const bool ok = j < jmax && idx < m;
                const uint32_t k = (uint32_t)(s_rec[(uint32_t)(ok ? idx : 0) * fa.words + mw] >> msh) & mmask;
                key[j] = ok ? (K)(((k - (uint32_t)key_base) << IDXB) | (uint32_t)idx) : (K) ~(K)0; // sentinels sort last
            }
            if (passes == 0) {
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    const int idx = w * chunk + j * 64 + lane;
                    if (j < jmax) l.s_keys[idx] = key[j];
                }
                __syncthreads();
            }
            int done = 0;
            for (int p = 0; p < passes; ++p) {
                const int bits = (nbits - done + (passes - p) - 1) / (passes - p);
                lds_radix_pass<K, rsort::NoVal, TPB, IPT, RB>(key, none, IDXB + done, bits, jmax, l.s_keys, (rsort::NoVal *)nullptr, l.local_base,
                                                              l.wave_tot, l.whist);
                done += bits;
                if (p + 1 < passes) {
#pragma unroll
                    for (int j = 0; j < IPT; ++j) {
                        const int idx = w * chunk + j * 64 + lane;
                        if (j < jmax) key[j] = l.s_keys[idx];
                    }
                    __syncthreads();
                }
            }
            IBVH_STAMP(1, 3);
            uint64_t *__restrict__ dst = (uint64_t *)(fa.out + start * stride);
            for (uint32_t g0 = threadIdx.x; g0 < total; g0 += TPB * U) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t g = g0 + u * TPB;
                    if (g < total) {
                        const uint32_t r = __umulhi(g, fa.inv_words);
                        const uint32_t part_w = g - r * fa.words;
                        dst[g] = s_rec[((uint32_t)l.s_keys[r] & ((1u << IDXB) - 1u)) * fa.words + part_w];
                    }
                }
            }
            IBVH_STAMP(1, 4);
            __syncthreads(); // (the LDS arrays are reused by the workgroup's next range)
            return;
        }
    }
    if (m <= CAP) {
        // ---- fast path: the range's keys live in LDS -------------------------------------------------------
        // the m keys are dealt to the waves in equal contiguous shares of `chunk` (a multiple of 64): a range of
        // CAP/4 keys keeps every wave busy with a quarter of the ranking work; (w, j, lane) order is memory order
        const int chunk = (int)((m + W * 64 - 1) / (W * 64)) * 64;
        const int jmax = chunk / 64; // <= IPT
        K key[IPT];
        uint16_t val[IPT];
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int idx = w * chunk + j * 64 + lane;
            const bool ok = j < jmax && idx < m;
            // (the strided key loads touch every line of the range's records: they are L2 hits for the copy below;
            // sentinels sort last)
            key[j] = ok ? (K)((K)load_morton(bucket + (int64_t)idx * stride, fa.lay) - key_base) : (K) ~(K)0;
            val[j] = (uint16_t)idx;
        }
        if (passes == 0) {
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const int idx = w * chunk + j * 64 + lane;
                if (j < jmax) l.s_idx[idx] = val[j];
            }
            __syncthreads();
        }
        int done = 0;
        IBVH_STAMP(1, 2);
        for (int p = 0; p < passes; ++p) {
            const int bits = (nbits - done + (passes - p) - 1) / (passes - p); // even split of the remaining bits
            lds_radix_pass<K, uint16_t, TPB, IPT, RB>(key, val, done, bits, jmax, l.s_keys, l.s_idx, l.local_base, l.wave_tot, l.whist);
            done += bits;
            if (p + 1 < passes) {
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    const int idx = w * chunk + j * 64 + lane;
                    if (j < jmax) {
                        key[j] = l.s_keys[idx];
                        val[j] = l.s_idx[idx];
                    }
                }
                __syncthreads();
            }
        }
        // records: lane <-> 8-byte word of the output range (fully coalesced stores; the loads hit the range's
        // partitioned records, which the key loads above have just pulled through L2)
        IBVH_STAMP(1, 3);
        const uint64_t *__restrict__ src = (const uint64_t *)bucket;
        uint64_t *__restrict__ dst = (uint64_t *)(fa.out + start * stride);
        const uint32_t total = (uint32_t)m * fa.words;
        constexpr int U = 8;
        for (uint32_t g0 = threadIdx.x; g0 < total; g0 += TPB * U) {
            uint64_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t g = g0 + u * TPB;
                const uint32_t gc = g < total ? g : 0u;
                const uint32_t r = __umulhi(gc, fa.inv_words);
                const uint32_t part_w = gc - r * fa.words;
                v[u] = src[(uint32_t)l.s_idx[r] * fa.words + part_w];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t g = g0 + u * TPB;
                if (g < total) dst[g] = v[u];
            }
        }
        IBVH_STAMP(1, 4);
        __syncthreads(); // (the LDS arrays are reused by the workgroup's next range)
        return;
    }

    // ---- slow path: more records than the LDS holds; tiled LSD between the two (key, position) arrays ----
    K *src_k = (K *)fa.kalt + start, *dst_k = (K *)fa.kpri + start;
    uint32_t *src_v = fa.valt + start, *dst_v = fa.vpri + start;
    for (int64_t i = threadIdx.x; i < m; i += TPB) {
        src_k[i] = (K)((K)load_morton(bucket + i * stride, fa.lay) - key_base);
        src_v[i] = (uint32_t)i;
    }
    __threadfence_block();
    __syncthreads();
    for (int p = 0; p < passes; ++p) {
        const int shift = RB * p;
        const int bits = nbits - shift < RB ? nbits - shift : RB;
        const uint32_t mask = (1u << bits) - 1u;
        if (threadIdx.x < R) l.gbase[threadIdx.x] = 0;
        __syncthreads();
        for (int64_t i = threadIdx.x; i < m; i += TPB) atomicAdd(&l.gbase[(uint32_t)(src_k[i] >> shift) & mask], 1u);
        __syncthreads();
        lds_exclusive_scan<TPB>(l.gbase, R, l.wave_tot);
        constexpr int SIPT = IPT / 2, SCAP = CAP / 2;
        for (int64_t t0 = 0; t0 < m; t0 += SCAP) {
            const int64_t cnt = m - t0 < SCAP ? m - t0 : SCAP;
            K key[SIPT];
            uint32_t val[SIPT];
#pragma unroll
            for (int j = 0; j < SIPT; ++j) {
                const int idx = w * 64 * SIPT + j * 64 + lane;
                const bool ok = idx < cnt;
                key[j] = ok ? src_k[t0 + idx] : (K) ~(K)0;
                val[j] = ok ? src_v[t0 + idx] : 0u;
            }
            uint32_t tot_d;
            lds_radix_pass<K, uint32_t, TPB, SIPT, RB>(key, val, shift, bits, SIPT, l.s_keys, l.s_vals32, l.local_base, l.wave_tot, l.whist, &tot_d);
            for (int pos = threadIdx.x; pos < cnt; pos += TPB) {
                const K kk = l.s_keys[pos];
                const uint32_t d = (uint32_t)(kk >> shift) & mask;
                const uint32_t dest = l.gbase[d] + ((uint32_t)pos - l.local_base[d]);
                dst_k[dest] = kk;
                dst_v[dest] = l.s_vals32[pos];
            }
            __syncthreads();
            // sentinels of a partial tile were counted in the last digit: real count there = cnt - local_base
            if (threadIdx.x < R) {
                uint32_t real = tot_d;
                if ((int)threadIdx.x == (int)mask && cnt < SCAP) real = (uint32_t)cnt - l.local_base[mask];
                l.gbase[threadIdx.x] += real;
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
    // `src_v` holds the range's positions in sorted order
    const uint64_t *src = (const uint64_t *)bucket;
    uint64_t *dst = (uint64_t *)(fa.out + start * stride);
    const uint64_t total = (uint64_t)m * fa.words;
    for (uint64_t g = threadIdx.x; g < total; g += TPB) {
        const uint64_t r = g / fa.words;
        const uint32_t part_w = (uint32_t)(g - r * fa.words);
        dst[g] = src[(uint64_t)src_v[r] * fa.words + part_w];
    }
    __syncthreads();
}

IBVH_D int bit_length(uint32_t v) { return v == 0 ? 0 : 32 - __builtin_clz(v); }
// workgroup-uniform values loaded through vector loads: moved to scalar registers (they are live across the sort)
IBVH_D uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
IBVH_D uint64_t uni(uint64_t v) { return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }

// grid = R + F2 workgroups: workgroup b < R finishes cell b (unless it became a segment); the others stride over the
// tiles of every extra level: tile t of segment k finishes the window of sub-cells that START inside records
// [t * tile, (t + 1) * tile) of the segment (consecutive sub-cells: one LDS sort with the sub-cell number as the top
// bits); a window that does not fit is finished sub-cell by sub-cell, skipping those the next level took
template <class K, int TPB, int IPT, bool RES = false>
__global__ __launch_bounds__(TPB) void finish_kernel(Tables tb, int radix, FinishArgs fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
    const FinishLds<K, TPB, IPT> l(bsm, RES ? fa.resident_off : 0u);
    IBVH_STAMP(1, 0);
    if (blockIdx.x == 0 && threadIdx.x == 0 && fa.skew_flag)
        *fa.skew_flag = (int32_t)(tb.needed[0] | (tb.needed[1] << 8) |
                                  (fa.eq_key_bits ? ((tb.needed[2] & 1u) << 16) | (fa.levels > 0 ? (tb.needed[3] & 1u) << 17 : 0u) : 0u));
    if ((int)blockIdx.x < radix) {
        const uint32_t d = blockIdx.x;
        const uint32_t start = uni(tb.cell_start[d]), m = uni(tb.cell_start[d + 1]) - start;
        if (m == 0 || (m > fa.cap && fa.levels > 0)) return; // (crowded cells are finished window by window below)
        if (fa.eq_key_bits) {
            K lo;
            int nb;
            cell_range<K>(tb, d, radix, fa.eq_key_bits, &lo, &nb);
            finish_range<K, TPB, IPT, RES>(fa, l, fa.buf[0], (int64_t)start, (int64_t)m, (K)uni(lo), __builtin_amdgcn_readfirstlane(nb));
            return;
        }
        finish_range<K, TPB, IPT, RES>(fa, l, fa.buf[0], (int64_t)start, (int64_t)m, (K)((K)d << fa.shift1), fa.shift1);
        return;
    }
    constexpr int C = 1 << L2_BITS;
    for (int li = 0; li < fa.levels; ++li) {
        const Level L = tb.lvl[li];
        const uint32_t ntiles = uni(L.hdr[1]);
        const char *buf = fa.buf[(li + 1) & 1];
        const bool handed_down = li + 1 < fa.levels; // crowded sub-cells are segments of the next level
        for (uint32_t t = blockIdx.x - radix; t < ntiles; t += gridDim.x - radix) {
            const uint32_t k = uni(L.tile_seg[t]);
            const uint64_t ka = uni(L.seg_and[k]), ko = uni(L.seg_or[k]);
            const Digit dg = level_digit(ka, ko);
            if (dg.terminal) continue; // the partition wrote the segment to `out`, sorted
            const uint32_t *ssp = L.sub_start + (int64_t)k * (C + 1);
            auto ss = [&](uint32_t i) { return uni(ssp[i]); };
            const uint32_t count = uni(L.seg_count[k]), reps = segment_reps(count, fa.tile);
            const uint32_t t_in_seg = t - uni(L.seg_tile[k]);
            for (uint32_t sub = 0; sub < reps; ++sub) {
            const uint32_t lo = (t_in_seg * reps + sub) * fa.tile, hi = lo + fa.tile;
            if (lo >= count) break;
            // e0 = first sub-cell starting at or after lo, e1 = first one starting at or after hi (binary searches; the
            // whole workgroup walks the same path)
            uint32_t e0 = 0, e1 = 0;
            {
                uint32_t a = 0, b = C; // first e with ss[e] >= lo
                while (a < b) {
                    const uint32_t mid = (a + b) >> 1;
                    if (ss(mid) < lo) a = mid + 1;
                    else b = mid;
                }
                e0 = a;
                a = e0, b = C;
                while (a < b) {
                    const uint32_t mid = (a + b) >> 1;
                    if (ss(mid) < hi) a = mid + 1;
                    else b = mid;
                }
                e1 = a;
            }
            if (e0 == e1) continue; // no sub-cell starts in this window (a large one covers it)
            if (ss(e1) == ss(e0)) continue; // empty sub-cells only
            const int64_t seg0 = (int64_t)uni(L.seg_start[k]);
            const K prefix = (K)common_prefix(ka, ko);
            // runs of consecutive sub-cells that fit one LDS sort (normally the whole window), one sort each; a single
            // crowded sub-cell is a segment of the next level — or, after the last level, takes the slow path
            for (uint32_t e = e0; e < e1;) {
                const uint32_t s0 = ss(e);
                uint32_t f = e + 1;
                const bool crowded = ss(f) - s0 > fa.cap;
                if (!crowded) { // the largest f <= e1 with ss(f) - s0 <= cap (ss ascends: binary search, usually f = e1 at once)
                    if (ss(e1) - s0 <= fa.cap) {
                        f = e1;
                    } else {
                        uint32_t lo_f = f, hi_f = e1; // ss(lo_f) fits, ss(hi_f) does not
                        while (hi_f - lo_f > 1) {
                            const uint32_t mid = (lo_f + hi_f) >> 1;
                            if (ss(mid) - s0 <= fa.cap) lo_f = mid;
                            else hi_f = mid;
                        }
                        f = lo_f;
                    }
                }
                const uint32_t m = ss(f) - s0;
                if (m != 0 && !(crowded && handed_down))
                    finish_range<K, TPB, IPT, RES>(fa, l, buf, seg0 + s0, (int64_t)m, (K)(prefix + ((K)e << dg.shift)),
                                              dg.shift + bit_length(f - e - 1));
                e = f;
            }
            } // sub
        }
    }
}
template <class K, int TPB, int IPT> constexpr size_t finish_smem() {
    return (size_t)TPB * IPT * (sizeof(K) + 2) + 2 * 256 * 4 + 64 + (size_t)(TPB / 64) * 256 * 2 + 64;
}

template <class K, int FT, int FI>
static int launch_finish(const Plan &p, const FinishArgs &fa_in, hipStream_t st) {
    FinishArgs fa = fa_in;
    size_t smem = finish_smem<K, FT, FI>();
    // resident path: room for a full range of records behind the sort's arrays, as long as the workgroup stays within the
    // LDS budget (tuning msd_resident_kb; 0 = off: the records are then gathered from memory as in rounds 2 and 3)
    if (sizeof(K) == 4 && (g_tuning.msd_resident_kb > 0 || p.resident)) {
        const size_t cap = (size_t)FT * FI;
        const size_t off = (size_t)align_up((int64_t)(smem - cap * 2), 16); // (no 16-bit positions in front of the record area)
        const size_t want = g_tuning.msd_resident_kb > 0 ? (size_t)g_tuning.msd_resident_kb * 1024 : (size_t)kMaxLds;
        const size_t budget = want < (size_t)kMaxLds ? want : (size_t)kMaxLds;
        size_t rec = cap * (size_t)fa.lay.stride;  // a full range, or what the budget leaves (larger ranges take the plain path)
        if (off + rec > budget) rec = budget > off ? ((budget - off) / 8) * 8 : 0;
        if (rec >= cap * 2 && rec >= 1024 * (size_t)fa.lay.stride) {
            fa.resident_off = (uint32_t)off;
            fa.resident_words = (uint32_t)(rec / 8);
            smem = off + rec;
        }
    }
    // (development knob: ask for more LDS than needed, i.e. fewer workgroups per CU — a smaller footprint in flight per L2)
    if ((size_t)g_tuning.msd_finish_pad_kb * 1024 > smem && g_tuning.msd_finish_pad_kb <= 160) smem = (size_t)g_tuning.msd_finish_pad_kb * 1024;
    const int f2 = fa.levels <= 0 ? 0 : (p.max_tiles2 < 1024 ? p.max_tiles2 : 1024); // workgroups that stride over the extra levels' windows
    if constexpr (sizeof(K) == 4) {
        if (fa.resident_words) { // (a kernel of its own: the resident branch must not cost the plain one registers)
            IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)finish_kernel<K, FT, FI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
            IBVH_LAUNCH((finish_kernel<K, FT, FI, true>), dim3((1u << p.bits) + f2), dim3(FT), smem, st, p.tb, 1 << p.bits, fa);
            return IBVH_OK;
        }
    }
    IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)finish_kernel<K, FT, FI>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
    IBVH_LAUNCH((finish_kernel<K, FT, FI>), dim3((1u << p.bits) + f2), dim3(FT), smem, st, p.tb, 1 << p.bits, fa);
    return IBVH_OK;
}

int run_finish(const Plan &p, int key_bytes, const FinishArgs &fa, hipStream_t st) {
    int rc;
    rc = IBVH_ERR_INVALID_ARG;
#define IBVH_FIN(K, T, I) \
    if (p.ftpb == T && p.fipt == I) rc = launch_finish<K, T, I>(p, fa, st);
    if (key_bytes == 4) {
        IBVH_FIN(uint32_t, 256, 6) IBVH_FIN(uint32_t, 256, 10) IBVH_FIN(uint32_t, 256, 11) IBVH_FIN(uint32_t, 256, 12)
        IBVH_FIN(uint32_t, 256, 8) IBVH_FIN(uint32_t, 256, 16) IBVH_FIN(uint32_t, 256, 32) IBVH_FIN(uint32_t, 512, 8)
        IBVH_FIN(uint32_t, 512, 16) IBVH_FIN(uint32_t, 512, 32) IBVH_FIN(uint32_t, 1024, 8) IBVH_FIN(uint32_t, 1024, 16)
        IBVH_FIN(uint32_t, 1024, 3) IBVH_FIN(uint32_t, 1024, 4) IBVH_FIN(uint32_t, 512, 6) IBVH_FIN(uint32_t, 1024, 6) IBVH_FIN(uint32_t, 512, 11)
        IBVH_FIN(uint32_t, 512, 12) IBVH_FIN(uint32_t, 512, 5) IBVH_FIN(uint32_t, 512, 7)
    } else {
        IBVH_FIN(uint64_t, 256, 8) IBVH_FIN(uint64_t, 256, 16) IBVH_FIN(uint64_t, 256, 32) IBVH_FIN(uint64_t, 512, 8)
        IBVH_FIN(uint64_t, 512, 16) IBVH_FIN(uint64_t, 1024, 8)
    }
#undef IBVH_FIN
    return rc;
}

} // namespace msd
} // namespace ibvh
#endif
