// ibvh_msd_impl.hpp — what ibvh_msd.hip (tables, sample, partition, host driver) and ibvh_msd_finish.hip (the in-LDS finish) share:
// two translation units because the finish kernel's 26 geometries alone take a minute to compile.  Not an interface: see ibvh_msd.hpp.
#pragma once
#include <cstdlib>

#include "ibvh_common.hpp"
#ifdef IBVH_PHASE_STAMPS
namespace ibvh { namespace msd { extern __device__ unsigned long long g_stamps[2][12][4096]; } }
#define IBVH_PASS_STAMP(k)                                                                                         \
    do {                                                                                                           \
        if (threadIdx.x == 0 && shift == 0) ::ibvh::msd::g_stamps[1][k][blockIdx.x & 4095] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#endif
#include "ibvh_radix.hpp"
#include "ibvh_msd.hpp"

namespace ibvh {
namespace msd {

using rsort::block_exclusive_scan;
using rsort::lds_barrier;
using rsort::lds_exclusive_scan;
using rsort::lds_exclusive_scan_pair;
using rsort::lds_radix_pass;
using rsort::RecordArgs;
using rsort::wave_rank;

constexpr int MSD_MAX_BITS = 12;
constexpr int L2_BITS = 8; // sub-cells per oversized cell: 2^8

// Diagnostic build only (-DIBVH_PHASE_STAMPS, tools/phase_stamps.sh): thread 0 of every workgroup stamps s_memtime
// at the phase boundaries into a buffer no product code reads.  In the product build the macro is empty.
#ifdef IBVH_PHASE_STAMPS
__device__ unsigned long long g_stamps[2][12][4096];
#define IBVH_STAMP(kern, k)                                                                              \
    do {                                                                                                 \
        if (threadIdx.x == 0) g_stamps[kern][k][blockIdx.x & 4095] = __builtin_amdgcn_s_memtime();       \
    } while (0)
#else
#define IBVH_STAMP(kern, k)
#endif


struct Digit {
    int shift, bits;
    uint32_t mask;
    bool terminal;
};
IBVH_D Digit level_digit(uint64_t a, uint64_t o) {
    const uint64_t x = a ^ o;
    const int hi = x ? 64 - __builtin_clzll(x) : 0;
    int bits = hi < L2_BITS ? hi : L2_BITS;
    if (bits < 1) bits = 1;
    const int shift = hi > bits ? hi - bits : 0;
    return Digit{shift, bits, (1u << bits) - 1u, shift == 0};
}
IBVH_D uint64_t common_prefix(uint64_t a, uint64_t o) { // the bits above the varying ones (equal in every key)
    const uint64_t x = a ^ o;
    const int hi = x ? 64 - __builtin_clzll(x) : 0;
    return hi >= 64 ? 0 : (a >> hi) << hi;
}

// A segment's tiles are `reps` partition tiles long (reps = 1 up to MAX_ROWS * tile records): no segment has more than
// MAX_ROWS of them, which bounds the column scan one workgroup does per segment
constexpr uint32_t MAX_ROWS = 256;
IBVH_D uint32_t segment_reps(uint32_t count, uint32_t tile) {
    const uint32_t r = (count + MAX_ROWS * tile - 1) / (MAX_ROWS * tile);
    return r ? r : 1u;
}
IBVH_D uint32_t segment_tiles(uint32_t count, uint32_t tile) {
    const uint32_t macro = segment_reps(count, tile) * tile;
    return (count + macro - 1) / macro;
}
// cell d of the equalised route holds the keys in [lo, lo + 2^nbits): its first splitter and the bits that may vary behind it
template <class K> IBVH_D void cell_range(const Tables &tb, uint32_t d, int radix, int key_bits, K *lo_out, int *nbits_out) {
    const K *spl = (const K *)tb.splitters;
    const K lo = spl[d];
    const K hi = (K)((d + 1 < (uint32_t)radix ? spl[d + 1] : (K)((K)1 << key_bits)) - (K)1); // (inclusive; a non-empty cell has hi >= lo)
    const uint64_t span = (uint64_t)(K)(hi - lo);
    *lo_out = lo;
    *nbits_out = span == 0 ? 0 : 64 - __builtin_clzll(span);
}

constexpr int kMaxLds = 160 * 1024; // LDS of a gfx950 CU
constexpr int kRescuers = 512;       // rescue workgroups of the finish kernel that scratch is carved for; a launch uses at most half of what the device holds at once (they WAIT for the ordinary workgroups)


struct FinishArgs {
    const char *buf[2]; // the two record buffers: level 1 writes buf[0]; extra level li reads buf[li & 1], writes buf[(li + 1) & 1]
    char *out;          // sorted records
    LeafLayout lay;
    uint32_t words;     // lay.stride / 8
    uint32_t inv_words; // ceil(2^32 / words): g / words == __umulhi(g, inv_words) for g < 2^29
    uint32_t cap;       // cells / sub-cells above this size are segments of the next level
    uint32_t tile;      // tile of the extra levels = window of sub-cells one workgroup finishes
    int levels;         // extra levels that ran; what is still crowded after them takes the slow path
    int shift1;         // key >> shift1 = cell
    int eq_key_bits;    // != 0: equalised cells (cell d = the keys between splitters d and d + 1); the number of key bits
    int32_t *skew_flag; // caller's hint word (may be null): receives `needed`
    // slow path only: (key, position) arrays of n entries each
    void *kalt, *kpri;
    uint32_t *valt, *vpri;
    // resident path (32-bit keys): 8-byte words of LDS behind the sort's arrays that hold a range's RECORDS (0: off), and
    // where they start (bytes from the base of the dynamic LDS)
    uint32_t resident_words, resident_off;
    // rescue path (ranges too large for one workgroup that no partition level is left to split): the grid's first normal_wgs
    // workgroups are the ordinary ones, the `rescuers` behind them share those ranges (0: off — the one-workgroup slow path)
    uint32_t normal_wgs, rescuers;
    uint32_t *rescue_priv;
};


// the finish launch for the plan's geometry (ibvh_msd_finish.hip)
int run_finish(const Plan &p, int key_bytes, const FinishArgs &fa, hipStream_t st);

} // namespace msd
} // namespace ibvh
