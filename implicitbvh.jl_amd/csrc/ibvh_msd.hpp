// ibvh_msd.hpp — interface of ibvh_msd.hip (the build's sort: MSD partition of whole records + in-LDS finish) for
// ibvh_build.hip.
#pragma once
#include "ibvh_common.hpp"
#include "ibvh_radix.hpp"

namespace ibvh {
namespace msd {

constexpr int MAX_LEVELS = 4; // partition levels beyond the first (levels 2 .. 5)

// One extra partition level: the SEGMENTS (contiguous record ranges with a common key prefix) the level before found
// crowded, each split into 256 sub-cells by the top 8 of the key bits that actually vary inside it.
struct Level {
    uint32_t *hdr;       // [0] segments, [1] tiles
    uint32_t *seg_start; // [S]       first record (absolute)
    uint32_t *seg_count; // [S]
    uint32_t *seg_tile;  // [S]       first tile of the segment
    uint64_t *seg_and;   // [S]       AND of the segment's keys
    uint64_t *seg_or;    // [S]       OR of the segment's keys: bits of (and ^ or) are the ones that vary
    uint32_t *sub_start; // [S][257]  exclusive prefix of the sub-cell sizes inside the segment; [256] = seg_count
    uint32_t *tile_seg;  // [T2]      tile -> segment
};

// Device-side tables shared by the kernels below (all in the sort scratch; sizes: R = 2^bits cells, S = the most
// crowded segments a level can have = n / cap + 1, T2 = the most tiles a level can have = num_tiles + S).
struct Tables {
    uint32_t *tile_hist;  // [num_tiles][R]   counts            (histogram kernel)
    uint32_t *tile_scan;  // [num_tiles][R]   exclusive prefix over the tiles (scan)
    uint32_t *cell_total; // [R]
    uint32_t *cell_start; // [R + 1]          exclusive prefix of cell_total (plan)
    uint32_t *needed;     // [16]             [0..3] extra levels this input would have used, fullest cell, equalised route's verdicts (the caller's
                          //                  hint); [4..7] the finish kernel's rescue counters (ibvh_msd_finish.hip), zeroed with [0]
    uint32_t *tile_hist2; // [T2][256]        per-tile sub-cell counts of the level being run (shared by the levels)
    uint32_t *tile_scan2; // [T2][256]
    uint64_t *tile_and;   // [T2]             AND / OR of the keys of every tile of the first extra level (range_kernel)
    uint64_t *tile_or;    // [T2]
    void *splitters;      // [R + 1] keys     equalised route: cell d holds the keys in [splitters[d], splitters[d + 1])
    uint16_t *dig;        // [n]              equalised route: every source leaf's cell
    uint32_t *rescue;     // [S] x 64 bytes   ranges the finish found too large for one workgroup (RescueEntry)
    uint32_t *rescue_priv; // [H][2 * cap]    private scratch of the H rescue workgroups
    Level lvl[MAX_LEVELS];
};

struct Plan {
    int bits;            // MSD digit width; 0: this path does not apply (tiny input) -> ibvh_sort.hip
    int shift;           // key_bits - bits: the cell is key >> shift
    int ptpb, pipt;      // partition (and histogram) tile geometry
    int num_tiles;
    int ftpb, fipt;      // finish workgroup: threads, keys per thread (capacity = ftpb * fipt)
    bool resident;       // the finish keeps a cell's records in LDS (ibvh_msd.hip, finish_range)
    int max_seg;         // S
    int max_tiles2;      // T2
    int rescuers;        // H: rescue workgroups at the end of the finish grid (scratch is carved for them)
    Tables tb;
};

Plan make_plan(int64_t n, int key_bits, int key_bytes, int leaf_bytes, void *sort_scratch);
size_t scratch_bytes(int64_t n, int key_bits, int key_bytes, int leaf_bytes);
int sort_records(const Plan &p, int key_bytes, int key_bits, const void *keys, int64_t n, const rsort::RecordArgs &ra, char *part2, char *out,
                 void *kalt, uint32_t *valt, void *kpri, uint32_t *vpri, int levels, bool equalize, void *skew_flag, hipStream_t st);

} // namespace msd
} // namespace ibvh
