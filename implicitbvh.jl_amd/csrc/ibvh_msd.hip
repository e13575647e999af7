// ibvh_msd.hip — the build's sort: a most-significant-digit partition of whole BoundingVolume records, then an
// in-LDS finish per cell of the Morton grid.  gfx950 only.
//
// Replaces AK.sort!(leaves, by = bv -> bv.morton) + wrap_bounding_volumes (reference src/build.jl:248-253, 328-352).
//
// Why not LSD passes over (key, position) pairs + a gather (ibvh_sort.hip, round 1): after the last LSD pass the
// source position of sorted element i is random, so the gather pays one 64-byte HBM sector per 16-byte volume
// (measured at 1e7 leaves: 2 x FETCH + WRITE = 1.58 GB for 0.48 GB of algorithmic bytes, 0.23 ms of the phase's
// 0.54).  Morton codes of a cloud spread over their top bits, so here the records themselves are partitioned ONCE
// by the top `bits` (<= 12) bits of their key — a CELL of the Morton grid per digit; source volumes are read in
// order (streaming), a tile's records are staged in LDS and leave as coalesced per-cell runs — and every cell (a
// few thousand records, L2-sized) is then finished by one workgroup: keys -> LDS, stable LSD passes on the remaining
// bits entirely in LDS, records copied from the cell's partitioned range to their final place with coalesced stores.
// HBM bytes per leaf (BSphere{F32} / Int32 / UInt32): hist 16 + 4, partition 4 + 16 + 24, finish 24 + 24 = 112
// against 4 + 4*16 + 44 (+ 48 of sector over-fetch) before.
//
// Skew: a cell that holds more records than a finish workgroup sorts in LDS (clustered clouds, surfaces: a mesh
// fills a fraction of the grid's cells, duplicates) becomes a SEGMENT of the next partition level, which touches only
// such segments: range (which key bits vary inside each segment; first extra level only, the others take "everything
// below the parent's digit") -> hist -> scan -> partition on the top 8 VARYING bits (range and hist read a COMPACT copy of the keys that the partition before wrote next to its records — 4 or 8
// bytes a record instead of a strided walk over whole records), the tiles of all segments side by side in one grid, so a single huge cell is still shared by many workgroups.
// Sub-cells that are still crowded become the segments of the level after (up to MAX_LEVELS extra levels, ping-pong
// between the two record buffers); a segment with at most 8 varying bits left is partitioned straight into the
// output, sorted (a cell of identical keys: one tiled copy).  The finish kernel sorts windows of consecutive
// sub-cells of every level.  Levels are launched up to a depth the caller chooses (`levels`); for a uniform cloud
// their launches find nothing to do and return at once.  What is still crowded at the last launched level is sorted
// by one workgroup with a tiled LSD through scratch arrays: slower, never wrong; `needed` tells the caller how many
// levels the input would have used.
//
// Stability: partitions rank in memory order and the LDS passes are stable, so equal keys keep input order (the
// oracle's definition of the unpinned AK.sort! tie order).
#include "ibvh_msd_impl.hpp"

namespace ibvh {
namespace msd {

// ------------------------------------------------------------------------------------------------------------
// scan: tile_hist is TILE-major ([num_tiles][radix]: the histogram kernel writes, and the partition kernel reads,
// one contiguous row per tile; a digit-major matrix costs both a 128-byte line per 4-byte counter).  tile_scan
// receives every column's (digit's) exclusive prefix over the tiles; digit_total[d] = the column sum.
// Workgroup (db, c): 64 digits (lane = digit: a wave reads 256 contiguous bytes of a row), tile rows
// [c*rows_per_chunk, ...).  The sum of the rows above its chunk is re-derived by the workgroup itself (rows dealt
// round-robin to its 16 waves) — redundant reads of an L2-resident matrix instead of a second launch.
// ------------------------------------------------------------------------------------------------------------
constexpr int SCAN_TPB = 1024;
__global__ __launch_bounds__(SCAN_TPB) void scan_tiles_kernel(const uint32_t *__restrict__ tile_hist, uint32_t *__restrict__ tile_scan,
                                                              int num_tiles, int radix, int rows_per_chunk,
                                                              uint32_t *__restrict__ digit_total) {
    constexpr int W = SCAN_TPB / 64;
    __shared__ uint32_t s_part[W][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ndb = radix >> 6;
    const int db = blockIdx.x % ndb, c = blockIdx.x / ndb;
    // (out of place: other workgroups are still summing the counts of this chunk's rows)
    const uint32_t *col = tile_hist + db * 64 + lane;
    uint32_t *out = tile_scan + db * 64 + lane;
    const int r0 = c * rows_per_chunk;
    const int r1 = r0 + rows_per_chunk < num_tiles ? r0 + rows_per_chunk : num_tiles;
    if (r0 >= num_tiles) return;
    // rows above the chunk
    uint32_t above = 0;
    {
        uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int r = w;
        for (; r + 7 * W < r0; r += 8 * W) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += col[(int64_t)(r + u * W) * radix];
        }
        for (; r < r0; r += W) acc[0] += col[(int64_t)r * radix];
        s_part[w][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        __syncthreads();
#pragma unroll
        for (int i = 0; i < W; ++i) above += s_part[i][lane];
        __syncthreads();
    }
    // own rows: wave w takes the contiguous share [a, b)
    const int share = (r1 - r0 + W - 1) / W;
    const int a = r0 + w * share;
    const int b = a + share < r1 ? a + share : r1;
    uint32_t mine = 0;
    for (int r = a; r < b; ++r) mine += col[(int64_t)r * radix];
    s_part[w][lane] = mine;
    __syncthreads();
    uint32_t run = above, chunk_total = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint32_t t = s_part[i][lane];
        if (i < w) run += t;
        chunk_total += t;
    }
    for (int r = a; r < b; ++r) {
        const uint32_t v = col[(int64_t)r * radix];
        out[(int64_t)r * radix] = run;
        run += v;
    }
    if (r1 == num_tiles && w == 0) digit_total[db * 64 + lane] = above + chunk_total;
}

// ------------------------------------------------------------------------------------------------------------
// Equalised cells (round 5; the caller asks for them: ibvh_build_desc.sort_equalize).  "Cell = top bits of the key" is right
// for a cloud that fills its bounding box; a surface mesh fills a fraction of that grid's cells and a clustered cloud a
// handful, and every crowded cell then goes through an extra partition level — a second move of (nearly) all the records.
// Here the cells are key RANGES of about equal population instead: one workgroup sorts a strided sample of the keys in
// LDS and takes every (S / R)-th one as a splitter; a histogram pass finds every leaf's cell by binary search over the
// R splitters (in LDS) and leaves it in a 2-byte side array, so that the partition ranks by a table look-up instead of a
// shift.  Cells are still contiguous, ascending key ranges: stability and everything behind the partition (finish by key
// range, extra levels for cells that are crowded regardless — many equal keys) stay as they are.
// ------------------------------------------------------------------------------------------------------------
// The sample: S = 1,024 * G keys (32,768 32-bit ones: 8 - 32 per cell, so that a cell of twice the average population — what the
// finish workgroups are sized for — is a 5-sigma event) at evenly spaced positions, dealt round-robin to G lists.
// sample_sort_kernel (one small workgroup per list, each on its own CU: one workgroup sorting 16,384 keys took 77 us — the LDS radix
// passes are latency chains — G of 1,024 take ~8) sorts every list in LDS; sample_rank_kernel gives every sample its rank in the union (its place in its own list +
// a binary search in each of the others, ties broken by list number: a strict total order) and the samples whose rank is a
// multiple of S / R become the splitter candidates.
template <class K> struct SampleGeom {
    static constexpr int LIST = 1024, G = sizeof(K) == 8 ? 16 : 32, S = LIST * G; // (S keys must fit the rank kernel's LDS: 128 KiB)
    static constexpr int TPB = 256, IPT = LIST / TPB, RB = 8;
    static constexpr size_t sort_smem = (size_t)LIST * sizeof(K) + ((size_t)4 << RB) + 32 * 4 + (size_t)(TPB / 64) * ((size_t)2 << RB);
};
template <class K>
__global__ __launch_bounds__(256) void sample_sort_kernel(const K *__restrict__ keys, int64_t n, K *__restrict__ lists, int key_bits) {
    using G = SampleGeom<K>;
    constexpr int TPB = G::TPB, IPT = G::IPT, RB = G::RB, R = 1 << RB;
    extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
    K *s_keys = (K *)bsm;
    uint32_t *local_base = (uint32_t *)(s_keys + G::LIST);
    uint32_t *wave_tot = local_base + R;
    uint16_t *whist = (uint16_t *)(wave_tot + 32);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int chunk = 64 * IPT;
    K key[IPT];
    const rsort::NoVal none[IPT] = {};
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        // sample number (place in this list) * G + list: every list spans the whole input (n < S: leaves are taken more than once — harmless)
        const uint64_t smp = (uint64_t)(w * chunk + j * 64 + lane) * (uint64_t)G::G + blockIdx.x;
        key[j] = keys[(smp * (uint64_t)n) / (uint64_t)G::S]; // (S is a power of two: a shift)
    }
    const int passes = (key_bits + RB - 1) / RB;
    int done = 0;
    for (int p = 0; p < passes; ++p) {
        const int b = (key_bits - done + (passes - p) - 1) / (passes - p);
        lds_radix_pass<K, rsort::NoVal, TPB, IPT, RB>(key, none, done, b, IPT, s_keys, (rsort::NoVal *)nullptr, local_base, wave_tot, whist);
        done += b;
#pragma unroll
        for (int j = 0; j < IPT; ++j) key[j] = s_keys[w * chunk + j * 64 + lane];
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < IPT; ++j) lists[(int64_t)blockIdx.x * G::LIST + w * chunk + j * 64 + lane] = key[j];
}
// number of keys of the sorted list L[0 .. 1,024) that are < e (OR_EQUAL: <= e)
template <class K, bool OR_EQUAL> IBVH_D uint32_t list_count(const K *L, K e) {
    uint32_t base = 0;
#pragma unroll
    for (uint32_t half = 512; half >= 1; half >>= 1) {
        const K v = L[base + half - 1];
        base += (OR_EQUAL ? v <= e : v < e) ? half : 0u;
    }
    const K v = L[base];
    return base + ((OR_EQUAL ? v <= e : v < e) ? 1u : 0u);
}
// grid = G * 4 workgroups: workgroup (g, q) ranks samples [256 q, 256 (q + 1)) of list g, four threads per sample (each takes every fourth
// of the other lists) — the kernel is bound by the LDS reads of its searches per CU, so it is spread over 4 G CUs, each holding
// the whole sample in LDS
template <class K>
__global__ __launch_bounds__(1024) void sample_rank_kernel(const K *__restrict__ lists, K *__restrict__ cand, int bits) {
    using G = SampleGeom<K>;
    extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
    K *s = (K *)bsm;
    for (int i = threadIdx.x; i < G::S; i += 1024) s[i] = lists[i];
    __syncthreads();
    const int g = blockIdx.x >> 2, q = blockIdx.x & 3;
    const uint32_t at = (uint32_t)(q * 256 + (threadIdx.x >> 2)), part = threadIdx.x & 3;
    const K e = s[g * G::LIST + at];
    uint32_t rank = part == 0 ? at : 0u;
#pragma unroll
    for (int o4 = 0; o4 < G::G; o4 += 4) {
        const int o = o4 + (int)part;
        if (o == g) continue;
        rank += o < g ? list_count<K, true>(s + o * G::LIST, e) : list_count<K, false>(s + o * G::LIST, e);
    }
    rank += __shfl_xor(rank, 1, 64);
    rank += __shfl_xor(rank, 2, 64);
    const uint32_t per = (uint32_t)G::S >> bits;
    if (part == 0 && (rank & (per - 1u)) == 0) cand[rank / per] = e;
}

// per tile of `tile` source leaves: every leaf's cell (the last splitter <= its key) and the tile's row of the histogram.  Every
// workgroup makes the splitters out of the candidates for itself (workgroup 0 also leaves them, and the verdict for the caller's
// hint, for the kernels behind it): splitter d = candidate d (the sample of rank d * S / R), except that a key that fills a whole
// cell's worth of samples — two candidates in a row are equal — gets a cell of its own, [v, v + 1): however many records carry it,
// they need no sorting (one extra level copies them out), and they no longer share a cell and its fate with what follows them.
template <class K>
__global__ __launch_bounds__(512) void bucket_hist_kernel(const K *__restrict__ keys, int64_t n, Tables tb, const K *__restrict__ cand, int bits,
                                                         int tile, int key_bits, uint32_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
    const int radix = 1 << bits;
    // The splitters as an implicit search tree in breadth-first order (node k: children 2k, 2k + 1; level L = nodes [2^L, 2^(L+1))):
    // a binary search over the SORTED array reads addresses a large power of two apart in its first steps — all in one LDS bank,
    // a 2^L-way conflict at step L — while here the nodes of a level are neighbours: no conflicts down to level 5, random below.
    K *eyt = (K *)bsm; // [1, radix): node k of level L, place j in its level = splitter (2j + 1) * radix / 2^(L+1)
    uint32_t *h = (uint32_t *)(eyt + radix);
    K *spl = (K *)(h + radix); // the sorted splitters (staged: the tree is a permutation of them — gathered from LDS, not from memory)
    for (int d = threadIdx.x; d < radix; d += 512) {
        h[d] = 0;
        const K c = cand[d], prev = cand[d > 0 ? d - 1 : 0];
        const K v = d == 0 ? (K)0 : ((d >= 2 && c == prev) ? (K)(c + 1) : c);
        spl[d] = v;
        if (blockIdx.x == 0) ((K *)tb.splitters)[d] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x + 1; k < radix; k += 512) {
        const int L = 31 - __builtin_clz((unsigned)k), j = k - (1 << L);
        eyt[k] = spl[(2 * j + 1) * (radix >> (L + 1))];
    }
    if (blockIdx.x == 0) {
        // the verdict: would the plain grid (cell = top `bits` bits) have had a crowded cell?  Crowded = a full finish workgroup's share
        // of the samples (lambda) in one grid cell, in whole candidates (m consecutive candidates in one cell = m * per .. (m + 1) *
        // per samples there): the threshold sits AT the capacity, not above it — rounds 5's lambda + 4 sqrt(lambda) + one candidate
        // said "fits" for a fullest cell of 1.0 .. 1.4 (250 k leaves) or .. 2 (1e6) workgroup shares, the chain went back to the plain
        // grid, met the crowded cell, came back, and so on every other build.  A cloud that fills its box holds <= 5/8 lambda per
        // cell (make_plan) and stays clear.  While the grid would be crowded the caller keeps asking for equalised cells.
        constexpr int S = SampleGeom<K>::S;
        const int shift = key_bits - bits, per = S >> bits;
        const float lambda = (float)cap * (float)S / (float)(n > 0 ? n : 1);
        const float kf = lambda / (float)per;
        const int m = kf >= (float)radix ? radix : (kf < 1.0f ? 1 : (int)kf);
        int crowded = 0;
        for (int d = threadIdx.x + 1; d + m < radix; d += 512) crowded |= (int)((cand[d] >> shift) == (cand[d + m] >> shift));
        crowded = __syncthreads_or(crowded);
        if (threadIdx.x == 0) {
            tb.needed[2] = crowded ? 1u : 0u;
            ((K *)tb.splitters)[radix] = (K) ~(K)0;
        }
    }
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * tile;
    constexpr int U = 4;
    for (int j0 = threadIdx.x; j0 < tile; j0 += 512 * U) {
        K k[U];
        uint32_t lo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + j0 + u * 512;
            k[u] = keys[i < n ? i : n - 1];
            lo[u] = 1;
        }
        for (int step = 0; step < bits; ++step) { // (the cell = the number of splitters 1 .. radix - 1 that are <= the key)
#pragma unroll
            for (int u = 0; u < U; ++u) lo[u] = 2u * lo[u] + (eyt[lo[u]] <= k[u] ? 1u : 0u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) lo[u] -= (uint32_t)radix;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + j0 + u * 512;
            const bool valid = j0 + u * 512 < tile && i < n;
            // (input in a coherent order — a mesh — puts most of a wave's 64 consecutive leaves into ONE cell: one LDS atomic for
            // all lanes that share the first lane's cell instead of 64 on the same address)
            const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo[u]);
            const uint64_t shared = __ballot(valid && lo[u] == first);
            if (valid) {
                if (lo[u] != first) atomicAdd(&h[lo[u]], 1u);
                else if ((shared & (((uint64_t)1 << (threadIdx.x & 63)) - 1)) == 0) atomicAdd(&h[first], (uint32_t)__popcll(shared));
                tb.dig[i] = (uint16_t)lo[u];
            }
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < radix; d += 512) tb.tile_hist[(int64_t)blockIdx.x * radix + d] = h[d];
}

// ------------------------------------------------------------------------------------------------------------
// plan (one workgroup): cell starts (every later kernel reads them instead of re-deriving them); the crowded cells
// (more than `cap` records) become the segments of the first extra level, with their tiles (`tile` records each)
// ------------------------------------------------------------------------------------------------------------
constexpr int PLAN_TPB = 1024;
IBVH_D uint32_t segment_tiles(uint32_t count, uint32_t tile);
__global__ __launch_bounds__(PLAN_TPB) void plan_kernel(Tables tb, int radix, uint32_t cap, uint32_t tile, int levels, int shift1, int eq_key_bytes,
                                                        int key_bits) {
    constexpr int PER = (1 << MSD_MAX_BITS) / PLAN_TPB; // cells per thread, at most
    __shared__ uint32_t wave_tot[PLAN_TPB / 64];
    __shared__ uint32_t s_tbase[(1 << MSD_MAX_BITS) + 1]; // first tile of every segment, for the tile -> segment search
    __shared__ uint32_t s_nover, s_ntiles;
    const Level L = tb.lvl[0];
    const int lo = threadIdx.x * PER;
    uint32_t tot[PER], sum = 0, nov = 0, nt = 0, biggest = 0, crowded_sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int d = lo + k;
        tot[k] = d < radix ? tb.cell_total[d] : 0u;
        sum += tot[k];
        biggest = tot[k] > biggest ? tot[k] : biggest;
        if (tot[k] > cap) {
            nov += 1;
            nt += segment_tiles(tot[k], tile);
            crowded_sum += tot[k];
        }
    }
    uint32_t total_n = 0, total_over = 0, total_tiles = 0;
    uint32_t run = block_exclusive_scan<PLAN_TPB>(sum, wave_tot, &total_n);
    uint32_t kk = block_exclusive_scan<PLAN_TPB>(nov, wave_tot, &total_over);
    uint32_t tt = block_exclusive_scan<PLAN_TPB>(nt, wave_tot, &total_tiles);
    uint32_t total_crowded = 0; // records in crowded cells
    block_exclusive_scan<PLAN_TPB>(crowded_sum, wave_tot, &total_crowded);
    // the fullest cell, in 1/128 of what a finish workgroup sorts (saturating at 255): the second byte of the caller's hint
    // word — how close a so far uniform input is to needing an extra level (include/ibvh.h, skew_flag)
    __shared__ uint32_t s_biggest;
    if (threadIdx.x == 0) s_biggest = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { // (one LDS atomic per wave: 1,024 same-address ones cost 3 us)
        const uint32_t t = (uint32_t)__shfl_xor((int)biggest, o, 64);
        biggest = t > biggest ? t : biggest;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(&s_biggest, biggest);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int d = lo + k;
        if (d < radix) {
            tb.cell_start[d] = run;
            if (tot[k] > cap) {
                if (levels > 0) {
                    L.seg_start[kk] = run;
                    L.seg_count[kk] = tot[k];
                    L.seg_tile[kk] = tt;
                    // (bounds; range_kernel + hist_level_kernel replace them by what the keys really span)
                    if (eq_key_bytes) { // equalised cells: the key range between two splitters -> its common prefix, everything below
                        uint64_t lo;
                        int nb;
                        if (eq_key_bytes == 8) {
                            uint64_t l8;
                            cell_range<uint64_t>(tb, (uint32_t)d, radix, key_bits, &l8, &nb);
                            lo = l8;
                        } else {
                            uint32_t l4;
                            cell_range<uint32_t>(tb, (uint32_t)d, radix, key_bits, &l4, &nb);
                            lo = l4;
                        }
                        const uint64_t hi = lo + (nb >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << nb) - 1)); // (>= the cell's last key)
                        const uint64_t x = lo ^ hi;
                        const int hb = x ? 64 - __builtin_clzll(x) : 0;
                        const uint64_t fixed = hb >= 64 ? 0 : (lo >> hb) << hb;
                        L.seg_and[kk] = fixed;
                        L.seg_or[kk] = fixed | (hb >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << hb) - 1));
                    } else {
                        L.seg_and[kk] = (uint64_t)d << shift1;
                        L.seg_or[kk] = ((uint64_t)d << shift1) | (((uint64_t)1 << shift1) - 1);
                    }
                }
                s_tbase[kk] = tt;
                kk += 1;
                tt += segment_tiles(tot[k], tile);
            }
            run += tot[k];
        }
    }
    if (threadIdx.x == 0) {
        tb.cell_start[radix] = total_n;
        s_tbase[total_over] = total_tiles;
        *tb.needed = total_over > 0 ? 1u : 0u;
        tb.needed[4] = tb.needed[5] = tb.needed[6] = tb.needed[7] = 0u; // (the finish kernel's rescue counters)
        {
            const uint64_t occ = ((uint64_t)s_biggest * 128u + cap - 1) / cap;
            tb.needed[1] = (uint32_t)(occ > 255 ? 255 : occ);
            // equalised cells that leave most records in crowded cells all the same (runs of equal keys: nothing a choice of cells
            // can do) have cost their launches for nothing: bit 17 of the hint, the caller goes back to the plain grid for a while
            tb.needed[3] = (eq_key_bytes && (uint64_t)total_crowded * 2 > total_n) ? 1u : 0u;
        }
        if (levels <= 0) total_over = total_tiles = 0; // the finish kernel sorts crowded cells by itself
#pragma unroll
        for (int l = 0; l < MAX_LEVELS; ++l) tb.lvl[l].hdr[0] = tb.lvl[l].hdr[1] = 0;
        L.hdr[0] = total_over;
        L.hdr[1] = total_tiles;
        s_nover = total_over;
        s_ntiles = total_tiles;
    }
    __syncthreads();
    const uint32_t nover = s_nover, ntiles = s_ntiles;
    for (uint32_t t = threadIdx.x; t < ntiles; t += PLAN_TPB) { // k with tbase[k] <= t < tbase[k + 1]
        uint32_t a = 0, b = nover;
        while (b - a > 1) {
            const uint32_t mid = (a + b) >> 1;
            if (s_tbase[mid] <= t) a = mid;
            else b = mid;
        }
        L.tile_seg[t] = a;
    }
}

// ------------------------------------------------------------------------------------------------------------
// extra levels.  The digit of a segment: the top L2_BITS of the bits that vary among its keys (and ^ or).  With at
// most L2_BITS varying bits the digit is all of them and the partition leaves the segment sorted (terminal).
// ------------------------------------------------------------------------------------------------------------
struct TileRange {
    uint32_t seg, first, cnt;
};
IBVH_D TileRange level_tile(const Level &L, uint32_t t, uint32_t tile) {
    const uint32_t s = L.tile_seg[t], count = L.seg_count[s];
    const uint32_t macro = segment_reps(count, tile) * tile;
    const uint32_t first = L.seg_start[s] + (t - L.seg_tile[s]) * macro, end = L.seg_start[s] + count;
    return TileRange{s, first, end - first < macro ? end - first : macro};
}

// which key bits vary inside every segment of level li
template <class K>
__global__ __launch_bounds__(256) void range_kernel(Tables tb, int li, const K *__restrict__ keys, uint32_t tile) {
    __shared__ uint64_t s_and[4], s_or[4];
    const Level L = tb.lvl[li];
    const uint32_t ntiles = L.hdr[1];
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const TileRange r = level_tile(L, t, tile);
        uint64_t a = ~(uint64_t)0, o = 0;
        for (uint32_t i0 = threadIdx.x; i0 < r.cnt; i0 += 256 * 8) { // (8 independent loads in flight)
            uint64_t k[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t i = i0 + u * 256;
                k[u] = (uint64_t)keys[r.first + (i < r.cnt ? i : i0)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a &= k[u];
                o |= k[u];
            }
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            a &= __shfl_xor(a, m, 64);
            o |= __shfl_xor(o, m, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            s_and[threadIdx.x >> 6] = a;
            s_or[threadIdx.x >> 6] = o;
        }
        __syncthreads();
        if (threadIdx.x == 0) { // per tile, no atomics: the histogram kernel merges a segment's (<= MAX_ROWS) tiles
            tb.tile_and[t] = (s_and[0] & s_and[1]) & (s_and[2] & s_and[3]);
            tb.tile_or[t] = (s_or[0] | s_or[1]) | (s_or[2] | s_or[3]);
        }
        __syncthreads();
    }
}

// per tile of a segment: counts of the segment's digit
template <class K>
__global__ __launch_bounds__(256) void hist_level_kernel(Tables tb, int li, const K *__restrict__ keys, uint32_t tile, int measured) {
    __shared__ uint32_t h[1 << L2_BITS];
    __shared__ uint64_t s_and[4], s_or[4];
    const Level L = tb.lvl[li];
    const uint32_t ntiles = L.hdr[1];
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        h[threadIdx.x] = 0;
        const TileRange r = level_tile(L, t, tile);
        uint64_t sa, so;
        if (measured) { // merge what range_kernel found in the segment's tiles; its first tile records the result
            const uint32_t t0 = L.seg_tile[r.seg], nt = segment_tiles(L.seg_count[r.seg], tile);
            uint64_t a = ~(uint64_t)0, o = 0;
            for (uint32_t i = threadIdx.x; i < nt; i += 256) {
                a &= tb.tile_and[t0 + i];
                o |= tb.tile_or[t0 + i];
            }
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                a &= __shfl_xor(a, m, 64);
                o |= __shfl_xor(o, m, 64);
            }
            if ((threadIdx.x & 63) == 0) {
                s_and[threadIdx.x >> 6] = a;
                s_or[threadIdx.x >> 6] = o;
            }
            __syncthreads();
            sa = (s_and[0] & s_and[1]) & (s_and[2] & s_and[3]);
            so = (s_or[0] | s_or[1]) | (s_or[2] | s_or[3]);
            if (t == t0 && threadIdx.x == 0) {
                L.seg_and[r.seg] = sa;
                L.seg_or[r.seg] = so;
            }
        } else {
            sa = L.seg_and[r.seg];
            so = L.seg_or[r.seg];
        }
        __syncthreads();
        const Digit dg = level_digit(sa, so);
        for (uint32_t i0 = threadIdx.x; i0 < r.cnt; i0 += 256 * 8) {
            uint64_t k[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t i = i0 + u * 256;
                k[u] = (uint64_t)keys[r.first + (i < r.cnt ? i : i0)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * 256 < r.cnt) atomicAdd(&h[(uint32_t)(k[u] >> dg.shift) & dg.mask], 1u);
        }
        __syncthreads();
        tb.tile_hist2[(int64_t)t * (1 << L2_BITS) + threadIdx.x] = h[threadIdx.x];
        __syncthreads();
    }
}

// one workgroup per segment: per sub-cell, the exclusive prefix over the segment's tiles; the sub-cell starts inside
// the segment; sub-cells that are still crowded become segments of the next level (or raise `needed`)
__global__ __launch_bounds__(1024) void scan_level_kernel(Tables tb, int li, int levels, uint32_t cap, uint32_t tile) {
    constexpr int C = 1 << L2_BITS, G = 1024 / C; // G row groups x C columns
    __shared__ uint32_t part[G][C];
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t s_base[2];
    const Level L = tb.lvl[li];
    const uint32_t nseg = L.hdr[0];
    const int col = threadIdx.x % C, rg = threadIdx.x / C;
    for (uint32_t k = blockIdx.x; k < nseg; k += gridDim.x) {
        const uint32_t count = L.seg_count[k];
        const uint32_t t0 = L.seg_tile[k], t1 = t0 + segment_tiles(count, tile);
        const uint32_t share = (t1 - t0 + G - 1) / G;
        const uint32_t a = t0 + rg * share < t1 ? t0 + rg * share : t1, b = a + share < t1 ? a + share : t1;
        uint32_t mine = 0;
        {
            uint32_t r = a;
            for (; r + 8 <= b; r += 8) { // (8 independent loads in flight)
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = tb.tile_hist2[(int64_t)(r + u) * C + col];
                mine += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            for (; r < b; ++r) mine += tb.tile_hist2[(int64_t)r * C + col];
        }
        part[rg][col] = mine;
        __syncthreads();
        uint32_t run = 0, total = 0;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const uint32_t v = part[i][col];
            if (i < rg) run += v;
            total += v;
        }
        {
            uint32_t r = a;
            for (; r + 8 <= b; r += 8) {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = tb.tile_hist2[(int64_t)(r + u) * C + col];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    tb.tile_scan2[(int64_t)(r + u) * C + col] = run;
                    run += v[u];
                }
            }
            for (; r < b; ++r) {
                const uint32_t v = tb.tile_hist2[(int64_t)r * C + col];
                tb.tile_scan2[(int64_t)r * C + col] = run;
                run += v;
            }
        }
        const uint32_t ex = block_exclusive_scan<1024>(rg == 0 ? total : 0u, wave_tot, nullptr); // (threads 0..C-1 carry the totals)
        if (rg == 0) L.sub_start[(int64_t)k * (C + 1) + col] = ex;
        if (threadIdx.x == 0) L.sub_start[(int64_t)k * (C + 1) + C] = count;
        // still crowded?
        const Digit dg = level_digit(L.seg_and[k], L.seg_or[k]);
        const bool crowded = rg == 0 && total > cap && !dg.terminal;
        const uint32_t my_tiles = crowded ? segment_tiles(total, tile) : 0u;
        uint32_t n_crowded = 0, n_tiles = 0;
        const uint32_t ci = block_exclusive_scan<1024>(crowded ? 1u : 0u, wave_tot, &n_crowded);
        const uint32_t ti = block_exclusive_scan<1024>(my_tiles, wave_tot, &n_tiles);
        if (n_crowded) { // (workgroup-uniform)
            if (li + 1 < levels) {
                const Level N = tb.lvl[li + 1];
                if (threadIdx.x == 0) {
                    s_base[0] = atomicAdd(&N.hdr[0], n_crowded);
                    s_base[1] = atomicAdd(&N.hdr[1], n_tiles);
                }
                __syncthreads();
                if (crowded) {
                    const uint32_t s = s_base[0] + ci, tb0 = s_base[1] + ti;
                    N.seg_start[s] = L.seg_start[k] + ex;
                    N.seg_count[s] = total;
                    N.seg_tile[s] = tb0;
                    // which bits vary inside it: at most those below this level's digit (only the first extra level
                    // measures its segments — range_kernel; a pass over the keys per level costs more than it saves)
                    const uint64_t fixed = common_prefix(L.seg_and[k], L.seg_or[k]) | ((uint64_t)col << dg.shift);
                    N.seg_and[s] = fixed;
                    N.seg_or[s] = fixed | (((uint64_t)1 << dg.shift) - 1);
                    for (uint32_t t = 0; t < my_tiles; ++t) N.tile_seg[tb0 + t] = s;
                }
            }
            if (threadIdx.x == 0) atomicMax(tb.needed, (uint32_t)(li + 2 < MAX_LEVELS ? li + 2 : MAX_LEVELS));
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------
// partition: a tile ranks its keys by a digit and writes every leaf's finished record to its cell's range in dst.
// Level 1 (L2 = false): tile t = TILE consecutive SOURCE leaves; digit = top `bits` bits; records assembled from the
//   source volumes (or copied from wrapped source records).
// Extra levels (L2 = true): tile t = TILE consecutive records of a crowded segment in the output of the level
//   before; digit = the segment's own (level_digit); records copied as they are, into the other record buffer — or
//   into `out` when the segment is terminal (sorted by this very partition).
// ------------------------------------------------------------------------------------------------------------
// The record of source leaf i is assembled word by word (8-byte words; every layout is a multiple of 8 with the
// volume first): volume words are copied, the words behind the volume carry .index and .morton.
struct TailLayout {
    int32_t vol_bytes, index_off, morton_off, index_bytes, morton_bytes;
};
IBVH_D uint64_t tail_word(const TailLayout &t, int word /* of the record */, uint64_t index, uint64_t key, uint64_t old) {
    const int base = word * 8;
    uint64_t w = old;
    const int io = t.index_off - base, mo = t.morton_off - base;
    if (io >= 0 && io < 8) {
        const uint64_t m = t.index_bytes == 8 ? ~(uint64_t)0 : (uint64_t)0xffffffffu;
        w = (w & ~(m << (8 * io))) | ((index & m) << (8 * io));
    }
    if (mo >= 0 && mo < 8) {
        const uint64_t m = t.morton_bytes == 8 ? ~(uint64_t)0 : (t.morton_bytes == 4 ? (uint64_t)0xffffffffu : (uint64_t)0xffffu);
        w = (w & ~(m << (8 * mo))) | ((key & m) << (8 * mo));
    }
    return w;
}

// Assemble the records of G rows of a wave (lane = one source leaf per row) and put them at their tile-local sorted
// positions in the LDS stage: NW source words per leaf are loaded for all G rows before the first LDS store (NW =
// sizeof(V)/8 for fresh volumes; the whole record for wrapped sources, whose .index is kept), the words behind the
// volume are assembled from (index, key).  NW / OW are compile-time: the kernel switches once, wave-uniformly, on
// the run-time layout.
template <int NW, int OW, bool WRAPPED, class K, int G>
IBVH_D void stage_rows(const uint64_t *__restrict__ src, uint32_t src_words, uint64_t *stage, const TailLayout &tl,
                       int64_t first /* source leaf of row 0, lane 0 */, int64_t n, const K *key, const uint32_t *pos) {
    const int lane = threadIdx.x & 63;
    uint64_t v[G][NW];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        int64_t i = first + j * 64 + lane;
        if (i >= n) i = n - 1; // (a valid address; the value is not stored)
        const uint64_t *p = src + (uint64_t)i * src_words;
#pragma unroll
        for (int k = 0; k < NW; ++k) v[j][k] = p[k];
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int64_t i = first + j * 64 + lane;
        if (i < n) {
            uint64_t *q = stage + (uint32_t)pos[j] * OW;
#pragma unroll
            for (int k = 0; k < OW; ++k) {
                if constexpr (WRAPPED) q[k] = tail_word(tl, k, 0, (uint64_t)key[j], v[j][k]); // only .morton is new
                else q[k] = k < NW ? v[j][k < NW ? k : 0] : tail_word(tl, k, (uint64_t)i + 1u, (uint64_t)key[j], 0);
            }
        }
    }
}

// occupancy the LDS stage allows (24-byte records): pinned so the layout switch cannot push the VGPR count over a step
constexpr int partition_min_waves(int tpb, int ipt) { return tpb * ipt <= 2048 ? 3 : 2; }

// EQ (level 1 only): the digit of a leaf is its equalised cell, read from tb.dig (bucket_hist_kernel) instead of shifted out of the key
template <class K, int TPB, int IPT, bool L2, bool EQ = false>
__global__ __launch_bounds__(TPB, partition_min_waves(TPB, IPT)) void partition_kernel(const K *__restrict__ keys, int64_t n, int shift, int bits,
                                                        Tables tb, int num_tiles, RecordArgs rec, uint32_t inv_words, int digit_bits,
                                                        int li, char *out, K *__restrict__ side_out) {
    constexpr int W = TPB / 64;
    constexpr int TILE = TPB * IPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int radix = 1 << bits;                             // table size (extra levels: always 2^L2_BITS columns)
    uint32_t mask = ((uint32_t)1 << digit_bits) - 1u;      // the digit itself may be narrower (few key bits left)
    uint64_t *__restrict__ dst = (uint64_t *)rec.dst;
    // layout: local_base[radix] | delta[radix] | wave_tot[32] | { whist[W * radix] (u16), later stage[TILE records] }
    uint32_t *local_base = (uint32_t *)smem;         // radix: tile-local start of digit d
    uint32_t *delta = local_base + radix;            // radix: (global position) - (tile-local sorted position) of digit d
    uint32_t *wave_tot = delta + radix;              // 32
    uint64_t *stage = (uint64_t *)(wave_tot + 32);
    uint16_t *whist = (uint16_t *)stage;             // W * radix
    uint16_t *sdig = (uint16_t *)((unsigned char *)stage + (size_t)TILE * (size_t)rec.lay.stride); // EQ: the digit of every staged record

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t tile_base, end;         // the tile's records are [tile_base, min(tile_base + TILE, end))
    int reps = 1;                   // extra levels: partition tiles this workgroup does one after the other
    const uint32_t *scan_row;       // this tile's row of the scanned histogram
    const uint32_t *digit_start;    // where each digit's range starts in dst (relative to dst_first)
    uint32_t dst_first = 0;
    if constexpr (L2) {
        const Level L = tb.lvl[li];
        const uint32_t t = blockIdx.x;
        if (t >= L.hdr[1]) return;
        const uint32_t k = L.tile_seg[t];
        dst_first = L.seg_start[k];
        reps = (int)segment_reps(L.seg_count[k], TILE);
        tile_base = (int64_t)dst_first + (int64_t)(t - L.seg_tile[k]) * reps * TILE;
        end = (int64_t)dst_first + L.seg_count[k];
        scan_row = tb.tile_scan2 + (int64_t)t * radix;
        digit_start = L.sub_start + (int64_t)k * (radix + 1);
        const Digit dg = level_digit(L.seg_and[k], L.seg_or[k]);
        shift = dg.shift, mask = dg.mask, digit_bits = dg.bits;
        if (dg.terminal) dst = (uint64_t *)out;
    } else {
        const int tile = xcd_remap(blockIdx.x, num_tiles);
        tile_base = (int64_t)tile * TILE;
        end = n;
        scan_row = tb.tile_scan + (int64_t)tile * radix;
        digit_start = tb.cell_start;
    }

    IBVH_STAMP(0, 0);
    // this tile's row of the (scanned, tile-major) histogram and the digit starts: coalesced, in flight while the keys
    // are ranked
    constexpr int DPT = (1 << MSD_MAX_BITS) / TPB; // digits per thread, at most
    // (straight-line, unconditional loads with clamped indices: the compiler can then wait for exactly the values it
    // needs — s_waitcnt vmcnt(N) — instead of draining everything before the first use)
    uint32_t tile_off_a[DPT], tile_off_b[DPT];
    // Level 1 without extra levels (li < 0): no plan_kernel ran — a launch that depends on its predecessor costs ~5 us.  The
    // digit starts are the exclusive prefix of the <= 4,096 cell totals: every workgroup scans them for itself, in the same
    // sweep (shared barriers) as its tile-local digit bases; workgroup 0 also leaves the starts and the hint words where
    // the finish kernel reads them.
    const bool own_starts = !L2 && li < 0;
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
        const int d = k * TPB + threadIdx.x;
        const int dc = d < radix ? d : radix - 1;
        tile_off_a[k] = scan_row[dc];
        tile_off_b[k] = own_starts ? tb.cell_total[dc] : digit_start[dc];
    }
    for (int rp = 0; rp < (L2 ? reps : 1); ++rp, tile_base += TILE) {
    if (L2 && tile_base >= end) break;
    const int64_t wave_base = tile_base + (int64_t)w * (64 * IPT);
    for (int i = threadIdx.x; i < W * radix / 2; i += TPB) ((uint32_t *)whist)[i] = 0;
    K key[IPT];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int64_t i = wave_base + j * 64 + lane;
        const int64_t ic = i < end ? i : end - 1;
        key[j] = keys[ic]; // (extra levels: the compact keys the partition before wrote; same index space as the records)
    }
    uint32_t dk[EQ ? IPT : 1];
    if constexpr (EQ) {
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int64_t i = wave_base + j * 64 + lane;
            dk[j] = i < end ? (uint32_t)tb.dig[i] : mask; // (past the end: the last digit, like the sentinel keys)
        }
    }
    // 16-byte fresh volumes (BSphere{Float32}, the common case) are requested now and stay in flight while the keys are
    // ranked (the barriers below do not wait for them); other layouts are fetched when they are staged
    const uint32_t src_words = (uint32_t)(rec.src_stride / 8);
    const uint32_t words = (uint32_t)rec.lay.stride / 8u;
    const bool wrapped = rec.src_wrapped != 0;
    const uint64_t *src = (const uint64_t *)rec.src;
    const int code = (wrapped ? 100 : 0) + rec.vol_words * 10 + (int)words;
    const bool preload = !L2 && (code == 23 || code == 24);
    // (issued for EVERY layout — the first 16 bytes of a volume or record are always there — so that the code stays
    // straight-line and the compiler can count the loads in flight exactly; only the 16-byte layouts use the values)
    uint64_t pre[IPT][2];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        int64_t i = wave_base + j * 64 + lane;
        if (i >= end) i = end - 1;
        const uint64_t *pv = src + (uint64_t)i * src_words;
        pre[j][0] = pv[0];
        pre[j][1] = pv[1];
    }
#pragma unroll
    for (int j = 0; j < IPT; ++j)
        if (wave_base + j * 64 + lane >= end) key[j] = (K) ~(K)0; // past the end: sentinels, ranked last
    lds_barrier();
    IBVH_STAMP(0, 1);
    uint16_t rank[IPT];
    uint16_t *my_hist = whist + w * radix;
    if constexpr (EQ) wave_rank<uint32_t, IPT>(dk, 0, mask, digit_bits, my_hist, lane, rank);
    else wave_rank<K, IPT>(key, shift, mask, digit_bits, my_hist, lane, rank);
    lds_barrier();
    IBVH_STAMP(0, 2);
    // per digit: exclusive prefix over the waves (in place), tile total
    uint32_t tile_cnt[DPT];
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
        const int d = k * TPB + threadIdx.x;
        tile_cnt[k] = 0;
        if (d < radix) {
            uint32_t run = 0;
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const uint32_t c = whist[i * radix + d];
                whist[i * radix + d] = (uint16_t)run;
                run += c;
            }
            local_base[d] = run;
            tile_cnt[k] = run;
            if (own_starts) delta[d] = tile_off_b[k]; // (the cell's total)
        }
    }
    lds_barrier();
    IBVH_STAMP(0, 3);
    if (own_starts) {
        lds_exclusive_scan_pair<TPB, true>(local_base, delta, radix, wave_tot);
        if (blockIdx.x == 0) { // what plan_kernel would have left for the finish kernel
            uint32_t biggest = 0;
#pragma unroll
            for (int k = 0; k < DPT; ++k) {
                const int d = k * TPB + threadIdx.x;
                if (d < radix) {
                    tb.cell_start[d] = delta[d];
                    biggest = tile_off_b[k] > biggest ? tile_off_b[k] : biggest;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t t = (uint32_t)__shfl_xor((int)biggest, o, 64);
                biggest = t > biggest ? t : biggest;
            }
            if (lane == 0) wave_tot[w] = biggest;
            lds_barrier();
            if (threadIdx.x == 0) {
                for (int i = 1; i < W; ++i) biggest = wave_tot[i] > biggest ? wave_tot[i] : biggest;
                const uint32_t cap = (uint32_t)-li; // (li = -(what a finish workgroup sorts) says "no plan")
                const uint64_t occ = ((uint64_t)biggest * 128u + cap - 1) / cap;
                tb.cell_start[radix] = (uint32_t)n;
                tb.needed[0] = biggest > cap ? 1u : 0u;
                tb.needed[1] = (uint32_t)(occ > 255 ? 255 : occ);
                tb.needed[4] = tb.needed[5] = tb.needed[6] = tb.needed[7] = 0u; // (the finish kernel's rescue counters)
            }
            lds_barrier(); // (wave_tot is reused)
        }
    } else {
        lds_exclusive_scan<TPB, true>(local_base, radix, wave_tot);
    }
    IBVH_STAMP(0, 4);
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
        const int d = k * TPB + threadIdx.x;
        if (d < radix) delta[d] = tile_off_a[k] + (own_starts ? delta[d] : tile_off_b[k]) + dst_first - local_base[d]; // mod 2^32
    }
    uint32_t pos[IPT];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        uint32_t d;
        if constexpr (EQ) d = dk[j];
        else d = (uint32_t)(key[j] >> shift) & mask;
        pos[j] = local_base[d] + my_hist[d] + rank[j];
    }
    lds_barrier(); // every wave is done with whist: its bytes become the stage
    if constexpr (EQ) {
#pragma unroll
        for (int j = 0; j < IPT; ++j)
            if (wave_base + j * 64 + lane < end) sdig[pos[j]] = (uint16_t)dk[j];
    }
    IBVH_STAMP(0, 5);
    // the records: sources are read in memory order (coalesced) and land at their sorted place in the stage
    const TailLayout tl{rec.vol_words * 8, wrapped ? -64 : rec.lay.index_off, rec.lay.morton_off, rec.index_bytes, rec.lay.morton_bytes};
    static_assert(IPT % 4 == 0, "rows are moved 4 or 2 at a time");
#define IBVH_MOVE(NW, OW, WR)                                                                                           \
    for (int j0 = 0; j0 < IPT; j0 += (NW <= 3 ? 4 : 2))                                                                 \
        stage_rows<NW, OW, WR, K, (NW <= 3 ? 4 : 2)>(src, src_words, stage, tl, wave_base + j0 * 64, end, key + j0, pos + j0);
    // (volume words, record words) of every layout layout_of() can produce; wrapped sources carry whole records
    if (preload) {
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int64_t i = wave_base + j * 64 + lane;
            if (i < end) {
                uint64_t *q = stage + pos[j] * words;
                q[0] = pre[j][0];
                q[1] = pre[j][1];
                q[2] = tail_word(tl, 2, (uint64_t)i + 1u, (uint64_t)key[j], 0);
                if (words == 4) q[3] = tail_word(tl, 3, (uint64_t)i + 1u, (uint64_t)key[j], 0);
            }
        }
    } else
    switch (code) {
    case 23: IBVH_MOVE(2, 3, false) break;  // BSphere{F32}, 24-byte record (Int32 / UInt16|UInt32)
    case 24: IBVH_MOVE(2, 4, false) break;  // BSphere{F32}, Int64 and/or UInt64
    case 34: IBVH_MOVE(3, 4, false) break;  // BBox{F32}
    case 35: IBVH_MOVE(3, 5, false) break;
    case 45: IBVH_MOVE(4, 5, false) break;  // BSphere{F64}
    case 46: IBVH_MOVE(4, 6, false) break;
    case 67: IBVH_MOVE(6, 7, false) break;  // BBox{F64}
    case 68: IBVH_MOVE(6, 8, false) break;
    case 123: IBVH_MOVE(3, 3, true) break;
    case 124: case 134: IBVH_MOVE(4, 4, true) break;
    case 135: case 145: IBVH_MOVE(5, 5, true) break;
    case 146: IBVH_MOVE(6, 6, true) break;
    case 167: IBVH_MOVE(7, 7, true) break;
    case 168: IBVH_MOVE(8, 8, true) break;
    default: break; // (unreachable: the host refuses other layouts)
    }
#undef IBVH_MOVE
    lds_barrier();
    IBVH_STAMP(0, 6);
    // out: lane <-> 8-byte word of the tile's sorted records; a digit's run goes to consecutive addresses
    const int64_t left = end - tile_base;
    const uint32_t valid = left < (int64_t)TILE ? (uint32_t)left : (uint32_t)TILE;
    const uint32_t total = valid * words;
    const char *stage_bytes = (const char *)stage;
    constexpr int U = 4; // LDS reads of U words are issued before the first global store
    for (uint32_t g0 = threadIdx.x; g0 < total; g0 += TPB * U) {
        uint64_t v[U];
        uint32_t kd[U], rr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t g = g0 + u * TPB;
            const uint32_t gc = g < total ? g : 0u;
            rr[u] = __umulhi(gc, inv_words);
            v[u] = stage[gc];
            if constexpr (EQ) kd[u] = sdig[rr[u]];
            else kd[u] = (uint32_t)((K)load_morton(stage_bytes + rr[u] * (uint32_t)rec.lay.stride, rec.lay) >> shift) & mask;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) kd[u] = delta[kd[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t g = g0 + u * TPB;
            if (g < total) dst[(uint64_t)(rr[u] + kd[u]) * words + (g - rr[u] * words)] = v[u];
        }
    }
    IBVH_STAMP(0, 7);
    // a compact copy of the keys, in the records' new order, for the range and histogram kernels of the next level —
    // only when that level has segments at all (one scalar load; a uniform cloud never pays this)
    {
        const int nli = L2 ? li + 1 : 0;
        if (side_out != nullptr && nli < MAX_LEVELS && tb.lvl[nli < MAX_LEVELS ? nli : 0].hdr[0] != 0) {
            for (uint32_t r = threadIdx.x; r < valid; r += TPB) {
                const K kk = (K)load_morton(stage_bytes + r * (uint32_t)rec.lay.stride, rec.lay);
                if constexpr (EQ) side_out[r + delta[sdig[r]]] = kk;
                else side_out[r + delta[(uint32_t)(kk >> shift) & mask]] = kk;
            }
        }
    }
    if constexpr (L2) { // the next tile of this workgroup starts where this one's digits ended
#pragma unroll
        for (int k = 0; k < DPT; ++k) tile_off_a[k] += tile_cnt[k];
        lds_barrier();
    }
    } // rp
}
template <class K, int TPB, int IPT> inline size_t partition_smem(int bits, int stride, bool eq = false) {
    const size_t hist = (size_t)(TPB / 64) * ((size_t)2 << bits), st = (size_t)TPB * IPT * (size_t)stride + (eq ? (size_t)TPB * IPT * 2 : 0);
    return ((size_t)8 << bits) + 128 + (hist > st ? hist : st);
}

// ------------------------------------------------------------------------------------------------------------
// finish: sort a contiguous range of partitioned records — one cell of level 1, or a window of consecutive
// sub-cells of an oversized cell — on the key bits that still vary inside it, and write the records to `out`
// ------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
static size_t carve_tables(Tables *tb, char *base, int radix, int num_tiles, int max_seg, int max_tiles2, int64_t n, int rescuers, int cap) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        uint32_t *p = base ? (uint32_t *)(base + off) : nullptr;
        off += (size_t)align_up((int64_t)bytes, 256);
        return p;
    };
    Tables t;
    t.tile_hist = take((size_t)radix * num_tiles * 4);
    t.tile_scan = take((size_t)radix * num_tiles * 4);
    t.cell_total = take((size_t)radix * 4);
    t.cell_start = take((size_t)(radix + 1) * 4);
    t.needed = take(64);
    t.tile_hist2 = take((size_t)max_tiles2 * 4 << L2_BITS);
    t.tile_scan2 = take((size_t)max_tiles2 * 4 << L2_BITS);
    t.tile_and = (uint64_t *)take((size_t)max_tiles2 * 8);
    t.tile_or = (uint64_t *)take((size_t)max_tiles2 * 8);
    t.splitters = take(((size_t)(radix + 1) * 2 + 32768) * 8); // splitters, candidates, the sorted sample lists
    t.dig = (uint16_t *)take((size_t)n * 2);
    t.rescue = take((size_t)max_seg * 64);
    t.rescue_priv = take((size_t)rescuers * 2 * (size_t)cap * 4);
    for (int l = 0; l < MAX_LEVELS; ++l) {
        Level &L = t.lvl[l];
        L.hdr = take(64);
        L.seg_start = take((size_t)max_seg * 4);
        L.seg_count = take((size_t)max_seg * 4);
        L.seg_tile = take((size_t)max_seg * 4);
        L.seg_and = (uint64_t *)take((size_t)max_seg * 8);
        L.seg_or = (uint64_t *)take((size_t)max_seg * 8);
        L.sub_start = take((size_t)max_seg * 4 * ((1 << L2_BITS) + 1));
        L.tile_seg = take((size_t)max_tiles2 * 4);
    }
    if (tb) *tb = t;
    return off;
}

Plan make_plan(int64_t n, int key_bits, int key_bytes, int leaf_bytes, void *sort_scratch) {
    // development knobs (ibvh_set_tuning): msd = 0 disables the path, msd_bits / _cap / _tile / _ftpb force a geometry
    const int enabled = g_tuning.msd, f_bits = g_tuning.msd_bits, f_cap = g_tuning.msd_cap, f_tile = g_tuning.msd_tile,
              f_avg = g_tuning.msd_avg, f_ftpb = g_tuning.msd_ftpb;
    Plan p{};
    if (!enabled || n < 4096 || key_bits <= 8 || n >= ((int64_t)1 << 32) - 65536) return p;
    int bits = 6; // (>= 6: the scan kernel works on blocks of 64 digits)
    while (bits < 11 && bits < key_bits - 1 && (n >> bits) > f_avg) ++bits;
    // 1.3e7 .. 2.7e7 leaves: 4,096 cells keep the average cell within the 8,192-record finish workgroup (the 16,384-
    // record one runs one workgroup per CU: 2e7 leaves 0.73 -> 0.40 ms for the finish, +0.05 for the wider partition)
    constexpr int64_t kCellMax8k = 8192 * 8 / 10;
    if (bits == 11 && key_bits > 13 && (n >> 11) > kCellMax8k && (n >> 12) <= kCellMax8k) bits = 12;
    if (f_bits) bits = f_bits;
    if (bits > MSD_MAX_BITS) bits = MSD_MAX_BITS;
    if (bits >= key_bits) bits = key_bits - 1;
    if (bits < 6) return p;
    const int64_t avg = n >> bits;
    // capacity of the finish workgroup: the average cell fills at most 5/8 of it (uniform clouds vary by a few per
    // cent; denser cells get the rest of the headroom before the second partition level takes them)
    int cap = 2048;
    const int cap_max = key_bytes == 8 ? 8192 : 16384; // (16384 x 10 B does not fit the LDS)
    // (the last step is taken later — the average cell may fill 8/10 of the 8,192-record workgroup, a Poisson cell count
    // then stays below it — because the 16,384-record workgroup is so much slower: 1.25e7 leaves 0.43 -> 0.26 ms)
    while (cap < cap_max && (cap < 8192 ? avg * 8 > (int64_t)cap * 5 : avg * 10 > (int64_t)cap * 8)) cap *= 2;
    if (f_cap) cap = f_cap;
    if (cap > cap_max) cap = cap_max;
    // partition tile: its records are staged in LDS (tile * leaf_bytes + 2 tables of 2^bits words <= 160 KiB)
    int tile = (n >= (int64_t(1) << 22) && leaf_bytes <= 32) ? 4096 : 2048;
    if (f_tile) tile = f_tile;
    while (tile > 1024 && (size_t)tile * leaf_bytes + ((size_t)8 << bits) + 128 > 160 * 1024) tile >>= 1;
    while (!f_tile && tile > 1024 && tile > cap / 2) tile >>= 1; // a window of sub-cells (< tile + one sub-cell) should fit the LDS sort
    switch (tile) {
    case 1024: p.ptpb = 256, p.pipt = 4; break;
    case 2048: p.ptpb = 256, p.pipt = 8; break;
    default: tile = 4096, p.ptpb = 512, p.pipt = 8; break;
    }
    if ((size_t)tile * leaf_bytes + ((size_t)8 << bits) + 128 > 160 * 1024) return p;
    // finish workgroups are small (4 or 8 waves, many keys per thread): several ranges per CU at different phases
    switch (cap) {
    case 2048: p.ftpb = 256; break;
    case 4096: p.ftpb = 256; break;
    case 8192: p.ftpb = 512; break;
    default: p.ftpb = 512; break;
    }
    // Round 4: cells of the 8,192-record geometry (~2.5 .. 6.5 k records of 24 bytes: 1e7-leaf builds) are finished with
    // their RECORDS resident in LDS — read from memory once instead of ~2.8 times (finish_range, resident path) — by one
    // 1,024-thread workgroup per CU (the memory phases of a lone workgroup need that many loads in flight); cells beyond the
    // ~5,000 records that fit beside the sort's arrays take the plain path inside the same kernel.  Measured at 1e7 leaves:
    // finish 212 -> 196 us, Morton+sort 463 -> 448 us (tools/ab_sort.py); smaller geometries lose (1e6: 24 -> 32 us: fewer
    // workgroups per CU) and keep the plain kernel.
    // (cells of ~3,000 records — 1.25e7 leaves at 12 bits — lose: 289 against 265 us; the rule asks for >= 4,096 on average)
    p.resident = cap == 8192 && avg >= 4096 && key_bytes == 4 && leaf_bytes <= 24 && g_tuning.msd_resident_kb >= 0 && !f_ftpb;
    if (p.resident) p.ftpb = 1024;
    if (f_ftpb) p.ftpb = f_ftpb;
    p.fipt = cap / p.ftpb;
    p.bits = bits;
    p.shift = key_bits - bits;
    p.num_tiles = (int)ceil_div(n, (int64_t)tile);
    // segments of an extra level hold more than cap records each; a segment's last tile may be partial
    p.max_seg = (int)(n / cap + 1);
    p.max_tiles2 = p.num_tiles + p.max_seg;
    // rescue workgroups of the finish kernel (ibvh_msd_finish.inc): no more than there can be windows to sort (the launch
    // clamps them to half of what the device holds at once)
    p.rescuers = g_tuning.msd_rescue ? (p.max_seg < kRescuers ? p.max_seg : kRescuers) : 0;
    carve_tables(&p.tb, (char *)sort_scratch, 1 << bits, p.num_tiles, p.max_seg, p.max_tiles2, n, p.rescuers, cap);
    return p;
}

// scratch for the tables of the plan make_plan() chooses for this input (0 when the path does not apply)
size_t scratch_bytes(int64_t n, int key_bits, int key_bytes, int leaf_bytes) {
    const Plan p = make_plan(n, key_bits, key_bytes, leaf_bytes, nullptr);
    if (!p.bits) return 0;
    return carve_tables(nullptr, nullptr, 1 << p.bits, p.num_tiles, p.max_seg, p.max_tiles2, n, p.rescuers, p.ftpb * p.fipt) + 4096;
}

template <class K, int PT, int PI>
static int launch_partitions(const Plan &p, const K *keys, int64_t n, const RecordArgs &ra, char *part2, char *out, K *side0, K *side1, int levels,
                             bool eq, hipStream_t st) {
    const size_t smem = partition_smem<K, PT, PI>(p.bits, ra.lay.stride, eq);
    if (smem > 160 * 1024) return IBVH_ERR_INVALID_ARG; // (make_plan sizes the tile for the record)
    // The attribute is per function and per device, and calls may come from several host threads with different geometries:
    // always the SAME value (the whole LDS), so that the order of concurrent calls cannot matter; a launch still only
    // allocates what it asks for.
    IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)partition_kernel<K, PT, PI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
    const uint32_t words = (uint32_t)ra.lay.stride / 8u;
    const uint32_t inv_words = (uint32_t)((((uint64_t)1 << 32) + words - 1) / words);
    if (eq) {
        IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)partition_kernel<K, PT, PI, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        IBVH_LAUNCH((partition_kernel<K, PT, PI, false, true>), dim3(p.num_tiles), dim3(PT), smem, st, keys, n, p.shift, p.bits, p.tb,
                    p.num_tiles, ra, inv_words, p.bits, levels > 0 ? 0 : -(p.ftpb * p.fipt), out, levels > 0 ? side0 : (K *)nullptr);
    } else
    IBVH_LAUNCH((partition_kernel<K, PT, PI, false>), dim3(p.num_tiles), dim3(PT), smem, st, keys, n, p.shift, p.bits, p.tb,
                p.num_tiles, ra, inv_words, p.bits, levels > 0 ? 0 : -(p.ftpb * p.fipt), out, levels > 0 ? side0 : (K *)nullptr);
    if (levels <= 0) return IBVH_OK;
    // extra levels: only the segments the level before found crowded (none for a uniform cloud: every workgroup
    // returns at once)
    const uint32_t tile = (uint32_t)PT * PI, cap = (uint32_t)(p.ftpb * p.fipt);
    const LeafLayout lay = ra.lay;
    const size_t smem2 = partition_smem<K, PT, PI>(L2_BITS, ra.lay.stride);
    IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)partition_kernel<K, PT, PI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
    char *buf[2] = {(char *)ra.dst, part2};
    K *side[2] = {side0, side1}; // compact keys of the records in buf[0] / buf[1]
    const unsigned stride_grid = (unsigned)(p.max_tiles2 < 2048 ? p.max_tiles2 : 2048);
    const unsigned seg_grid = (unsigned)(p.max_seg < 1024 ? p.max_seg : 1024);
    for (int li = 0; li < levels; ++li) {
        const char *src = buf[li & 1];
        const int measure = g_tuning.msd_range; // 0 = first extra level on the next 8 bits, unmeasured
        const bool measured = li == 0 && measure != 0;
        if (measured) IBVH_LAUNCH((range_kernel<K>), dim3(stride_grid), dim3(256), 0, st, p.tb, li, (const K *)side[li & 1], tile);
        IBVH_LAUNCH((hist_level_kernel<K>), dim3(stride_grid), dim3(256), 0, st, p.tb, li, (const K *)side[li & 1], tile, measured ? 1 : 0);
        IBVH_LAUNCH((scan_level_kernel), dim3(seg_grid), dim3(1024), 0, st, p.tb, li, levels, cap, tile);
        RecordArgs r2 = ra;
        r2.src = src; // whole records, copied as they are
        r2.dst = buf[(li + 1) & 1];
        r2.src_stride = ra.lay.stride;
        r2.src_wrapped = 1;
        IBVH_LAUNCH((partition_kernel<K, PT, PI, true>), dim3(p.max_tiles2), dim3(PT), smem2, st, (const K *)side[li & 1], n, 0, L2_BITS, p.tb,
                    p.max_tiles2, r2, inv_words, L2_BITS, li, out, li + 1 < levels ? side[(li + 1) & 1] : (K *)nullptr);
    }
    return IBVH_OK;
}
// keys: n Morton keys (uint32 / uint64) in source order, their top-digit tile histogram already in p.tb.tile_hist.
// ra: source -> level-1 partitioned records (ra.dst: n records of scratch); part2: n more records of scratch (the
// extra levels ping-pong between the two); out: the sorted records.  (kalt, valt, kpri, vpri): n-entry scratch arrays:
// kalt / kpri carry the compact key copies of the extra levels, then all four serve the slow path (kpri may alias
// `keys`: the source keys are dead once the first partition has run).  levels: extra partition levels to launch.
int sort_records(const Plan &p, int key_bytes, int key_bits, const void *keys, int64_t n, const RecordArgs &ra, char *part2, char *out, void *kalt,
                 uint32_t *valt, void *kpri, uint32_t *vpri, int levels, bool equalize, void *skew_flag, hipStream_t st) {
    if (levels < 0 || levels > MAX_LEVELS) levels = MAX_LEVELS;
    // equalised cells: only where the partition's LDS stage has room for the 2-byte digit of every staged record
    const size_t tile_elems = (size_t)p.ptpb * p.pipt;
    bool eq = equalize && g_tuning.msd_equalize >= 0 && tile_elems * ((size_t)ra.lay.stride + 2) + ((size_t)8 << p.bits) + 128 <= (size_t)160 * 1024;
    if (g_tuning.msd_equalize > 0) eq = tile_elems * ((size_t)ra.lay.stride + 2) + ((size_t)8 << p.bits) + 128 <= (size_t)160 * 1024; // (test knob: always)
    if (eq) {
        // sorted sample -> splitter candidates -> every leaf's cell + the tile histogram (three launches in place of the histogram
        // the encode kernel fused; the lists and the candidates live behind the splitters in the sort scratch)
        const size_t hsm = ((size_t)(2 * key_bytes + 4)) << p.bits;
        const uint32_t cap = (uint32_t)(p.ftpb * p.fipt);
        auto run = [&](auto kt) -> int {
            using K = decltype(kt);
            using SG = SampleGeom<K>;
            K *cand = (K *)p.tb.splitters + ((size_t)1 << p.bits) + 1, *lists = cand + ((size_t)1 << p.bits) + 1;
            IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)sample_rank_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
            // (4,096 cells of 64-bit codes: 80 KB of splitters, tree and counters; bucket_hist_kernel has a few static bytes of its own)
            IBVH_HIP_CHECK(hipFuncSetAttribute((const void *)bucket_hist_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 8192));
            IBVH_LAUNCH((sample_sort_kernel<K>), dim3(SG::G), dim3(SG::TPB), SG::sort_smem, st, (const K *)keys, n, lists, key_bits);
            IBVH_LAUNCH((sample_rank_kernel<K>), dim3(SG::G * 4), dim3(1024), (size_t)SG::S * sizeof(K), st, (const K *)lists, cand, p.bits);
            IBVH_LAUNCH((bucket_hist_kernel<K>), dim3(p.num_tiles), dim3(512), hsm, st, (const K *)keys, n, p.tb, (const K *)cand, p.bits, (int)tile_elems,
                        key_bits, cap);
            return IBVH_OK;
        };
        if (int e = key_bytes == 8 ? run(uint64_t{}) : run(uint32_t{})) return e;
    }
    {
        const int radix = 1 << p.bits, ndb = radix >> 6;
        int chunks = 512 / ndb < 1 ? 1 : 512 / ndb; // ~512 workgroups
        if (chunks > (p.num_tiles + 15) / 16) chunks = (p.num_tiles + 15) / 16;
        const int rows = (p.num_tiles + chunks - 1) / chunks;
        chunks = (p.num_tiles + rows - 1) / rows;
        IBVH_LAUNCH((scan_tiles_kernel), dim3(ndb * chunks), dim3(SCAN_TPB), 0, st, p.tb.tile_hist, p.tb.tile_scan, p.num_tiles, radix, rows,
                    p.tb.cell_total);
        // (no extra levels: nothing but the cell starts and the hint would come out of the plan, and the partition kernel
        // derives those from the cell totals itself — one dependent launch less)
        if (levels > 0)
            IBVH_LAUNCH((plan_kernel), dim3(1), dim3(PLAN_TPB), 0, st, p.tb, radix, (uint32_t)(p.ftpb * p.fipt), (uint32_t)(p.ptpb * p.pipt),
                        levels, p.shift, eq ? key_bytes : 0, key_bits);
    }
    int rc = IBVH_ERR_INVALID_ARG;
#define IBVH_PART(K, T, I) \
    if (p.ptpb == T && p.pipt == I) rc = launch_partitions<K, T, I>(p, (const K *)keys, n, ra, part2, out, (K *)kalt, (K *)kpri, levels, eq, st);
    if (key_bytes == 4) {
        IBVH_PART(uint32_t, 256, 4) IBVH_PART(uint32_t, 256, 8) IBVH_PART(uint32_t, 512, 8)
    } else {
        IBVH_PART(uint64_t, 256, 4) IBVH_PART(uint64_t, 256, 8) IBVH_PART(uint64_t, 512, 8)
    }
#undef IBVH_PART
    if (rc) return rc;
    FinishArgs fa{};
    fa.buf[0] = ra.dst;
    fa.buf[1] = part2;
    fa.out = out;
    fa.lay = ra.lay;
    fa.words = (uint32_t)ra.lay.stride / 8u;
    fa.inv_words = (uint32_t)((((uint64_t)1 << 32) + fa.words - 1) / fa.words);
    fa.cap = (uint32_t)(p.ftpb * p.fipt);
    fa.tile = (uint32_t)(p.ptpb * p.pipt);
    fa.levels = levels;
    fa.shift1 = p.shift;
    fa.eq_key_bits = eq ? key_bits : 0;
    fa.skew_flag = (int32_t *)skew_flag;
    fa.kalt = kalt;
    fa.kpri = kpri;
    fa.valt = valt;
    fa.vpri = vpri;
    fa.rescuers = (uint32_t)p.rescuers;
    fa.rescue_priv = p.tb.rescue_priv;
    rc = run_finish(p, key_bytes, fa, st);
    if (rc) return rc;
    IBVH_LAUNCH_CHECK();
    return IBVH_OK;
}

} // namespace msd
} // namespace ibvh

#ifdef IBVH_PHASE_STAMPS // (diagnostic builds: the finish kernels in this translation unit, one stamp buffer)
#define IBVH_MSD_SINGLE_TU
#include "ibvh_msd_finish.hip"
extern "C" int ibvh_debug_stamps(unsigned long long *out /* 2 * 12 * 4096 */) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ibvh::msd::g_stamps), sizeof(ibvh::msd::g_stamps));
}
#endif
