// ibvh_msd.hpp — interface of ibvh_msd.hip (the build's sort: MSD partition of whole records + in-LDS finish) for
// ibvh_build.hip.
#pragma once
#include "ibvh_common.hpp"
#include "ibvh_radix.hpp"

namespace ibvh {
namespace msd {

// Device-side tables shared by the kernels below (all in the sort scratch; sizes: R = 2^bits cells, T2 = the most
// level-2 tiles an input of n records can have = n / tile + R).
struct Tables {
    uint32_t *tile_hist;      // [num_tiles][R]   counts            (histogram kernel)
    uint32_t *tile_scan;      // [num_tiles][R]   exclusive prefix over the tiles (scan)
    uint32_t *cell_total;     // [R]
    uint32_t *cell_start;     // [R + 1]          exclusive prefix of cell_total (plan)
    uint32_t *hdr;            // [0] oversized cells, [1] level-2 tiles
    uint32_t *over_cell;      // [R]              k -> cell
    uint32_t *over_tile_base; // [R + 1]          k -> first level-2 tile of oversized cell k
    uint32_t *tile_cell;      // [T2]             level-2 tile -> k
    uint32_t *tile_hist2;     // [T2][256]
    uint32_t *tile_scan2;     // [T2][256]
    uint32_t *sub_total;      // [R][256]         (indexed by k)
    uint32_t *sub_start;      // [R][256]         exclusive prefix of sub_total within the cell
};

struct Plan {
    int bits;            // MSD digit width; 0: this path does not apply (tiny input) -> ibvh_sort.hip
    int shift;           // key_bits - bits: the cell is key >> shift
    int ptpb, pipt;      // partition (and histogram) tile geometry
    int num_tiles;
    int ftpb, fipt;      // finish workgroup: threads, keys per thread (capacity = ftpb * fipt)
    int max_tiles2;      // most level-2 tiles this input can have
    Tables tb;
};

Plan make_plan(int64_t n, int key_bits, int key_bytes, int leaf_bytes, void *sort_scratch);
size_t scratch_bytes(int64_t n, int key_bits, int key_bytes, int leaf_bytes);
int sort_records(const Plan &p, int key_bytes, const void *keys, int64_t n, const rsort::RecordArgs &ra, char *part2, char *out,
                 void *kalt, uint32_t *valt, void *kpri, uint32_t *vpri, int two_level, void *skew_flag, hipStream_t st);

} // namespace msd
} // namespace ibvh
