/*
 * ibvh.h — C ABI of libibvh, the MI355X (gfx950) implicit-BVH engine.
 *
 * This is the drop-in boundary for ImplicitBVH.jl's hot path (Morton encode ->
 * stable radix sort by Morton code (an MSD partition of whole records + an in-LDS
 * LSD finish; plain LSD passes for tiny inputs: the result is the stable order
 * either way) -> bottom-up ImplicitTree merge -> LVT / BFS traversal of one BVH,
 * two BVHs, or rays).  The reference has no FFI: its back-end seam is
 * Julia method dispatch on the array type (src/build.jl:198, src/traverse/
 * leaf_vs_tree/traverse_single.jl:1, src/raytrace/raytrace.jl:71).  A package
 * extension for AMDGPU.jl's ROCArray `ccall`s the entry points below (see
 * INTEGRATION.md); every one cites the reference function it replaces.
 *
 * Conventions
 *   - plain C, no HIP/torch types: `void *stream` is a hipStream_t (NULL = default
 *     stream); all buffer pointers are DEVICE pointers owned by the caller;
 *   - record layouts are the C layouts Julia gives its isbits structs:
 *       BSphere{T}            { T x[3]; T r; }                  (bsphere.jl:26-29)
 *       BBox{T}               { T lo[3]; T up[3]; }             (bbox.jl:35-38)
 *       BoundingVolume{V,I,M} { V volume; I index; M morton; }  (bounding_volumes.jl:55-59)
 *       IndexPair{I}          { I first; I second; }            (traverse.jl:6)
 *   - indices are 1-based exactly as in the reference;
 *   - every function returns an ibvh_status; nothing is allocated or freed by the
 *     library; calls are asynchronous on `stream` unless stated otherwise.
 */
#ifndef IBVH_H
#define IBVH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct layout or an entry point's argument list changes.  A binding checks
 * ibvh_abi_version() == IBVH_ABI_VERSION (the header it was written against) right after loading the
 * library and refuses a mismatch: a stale caller would otherwise hand the GPU garbage pointers.
 *   1: round 1   2: ibvh_build_desc.sort_levels / skew_flag, ibvh_bfs_result.resume_*, *_enqueue(total_dev)
 *   3: *_enqueue(total_host), ibvh_set_tuning, ibvh_lvt_work_counters, ray `narrow`, contact positions
 *   4: the multi-GPU driver (ibvh_comm, ibvh_dist_*)
 *   5: ibvh_dist_cross_* (boundary leaves), ibvh_comm_release, ibvh_build_desc.sort_equalize (was reserved_: same layout) */
#define IBVH_ABI_VERSION 5
int32_t ibvh_abi_version(void);

/* ----------------------------------------------------------------------------------- */
/* status codes -> exceptions the Julia shim raises                                      */
/* ----------------------------------------------------------------------------------- */
typedef enum ibvh_status {
    IBVH_OK = 0,
    IBVH_ERR_INVALID_ARG = 1, /* ArgumentError (@argcheck in build.jl:207,235,260; lvt/traverse_single.jl:10) */
    IBVH_ERR_DOMAIN = 2,      /* DomainError: fewer than one leaf (implicit_tree.jl:78-80)                    */
    IBVH_ERR_UNSUPPORTED = 3, /* a type combination this library does not instantiate                        */
    IBVH_ERR_CAPACITY = 4,    /* caller buffer too small; the required size is reported                      */
    IBVH_ERR_OVERFLOW = 5,    /* a count does not fit the index type I                                       */
    IBVH_ERR_HIP = 6,         /* a HIP runtime call failed                                                    */
    IBVH_ERR_SCRATCH = 7,     /* scratch buffer smaller than ibvh_build_scratch_bytes()                      */
    IBVH_ERR_PEER = 8         /* multi-GPU calls: another rank's arguments were not acceptable — every rank returns
                                 together instead of waiting for it in a collective (that rank reports its own error) */
} ibvh_status;

/* ----------------------------------------------------------------------------------- */
/* type descriptors                                                                     */
/* ----------------------------------------------------------------------------------- */
enum { IBVH_BSPHERE = 0, IBVH_BBOX = 1 };       /* volume kinds                       */
enum { IBVH_F32 = 0, IBVH_F64 = 1 };            /* float types                        */
enum { IBVH_I32 = 0, IBVH_I64 = 1 };            /* index types (BVHOptions.index)     */
enum { IBVH_U16 = 0, IBVH_U32 = 1, IBVH_U64 = 2 }; /* Morton types (morton/default.jl:21) */

/* narrow-phase menu.  The reference takes an arbitrary Julia closure that is only ever
 * evaluated as `iscontact(...) && narrow(a, b)` at leaf level
 * (lvt/traverse_single.jl:170); closures cannot cross a C ABI, so a fixed menu is offered. */
enum {
    IBVH_NARROW_NONE = 0,      /* (a, b) -> true / (bv, p, d) -> true  (the reference defaults)        */
    IBVH_NARROW_MORTON_LT = 1, /* self / pair: (a, b) -> a.morton < b.morton (runtests.jl:1239)        */
    IBVH_NARROW_INDEX_LT = 2,  /* self / pair: (a, b) -> a.index < b.index                             */
    IBVH_NARROW_RAY_ORIGIN_OUTSIDE = 3, /* rays (raytrace/raytrace.jl:76): (bv, p, d) -> p lies outside
                                  bv.volume — drops the leaves a ray STARTS in (rays cast from a surface) */
    IBVH_NARROW_MASK = 0xff,
    /* Any other pure `narrow`: OR this flag into the `narrow` argument of a traversal (of its _count AND its _write /
     * _enqueue call) and the contact list holds leaf POSITIONS instead of user indices, in the same order:
     *   one BVH   (position of the query leaf, position of its partner) in bvh.leaves, 1-based, query first
     *             (the reference evaluates narrow(query, partner); the query is the leaf with the smaller position)
     *   two BVHs  (position in bvh1.leaves, position in bvh2.leaves)
     *   rays      (position of the leaf in bvh.leaves, iray)
     * so the caller can evaluate the predicate on the records itself and keep what passes: `narrow` is only ever
     * evaluated as `iscontact(...) && narrow(...)` at leaf level (lvt/traverse_single.jl:170,
     * bfs/traverse_single_gpu.jl:187, raytrace/leaf_vs_tree/leaf_vs_tree.jl:194), i.e. a post-filter. */
    IBVH_OUTPUT_POSITIONS = 0x100,
    /* Pair LVT traversals only: let the BVH with FEWER leaves supply the work items (the reference lets the larger one,
     * lvt/traverse_pair.jl:15-36 — and so does the library without this flag).  Same pairs, still (bvh1, bvh2) order inside a
     * pair, but the LIST is ordered by the smaller BVH's leaves, so it is not the reference's order: for callers that want the
     * contact SET.  A handful of leaves against a large tree then costs (few leaves) x (tree depth) instead of one work item —
     * and 8 cached contacts of scratch — per leaf of the large tree; the cross-shard completion uses it (a slice's thin boundary
     * shell against the neighbour's whole slice).  Size the scratch and `counts` for min(n1, n2) items then. */
    IBVH_PAIR_SMALLER_DRIVES = 0x200
};

typedef struct ibvh_types {
    int32_t leaf_kind;   /* IBVH_BSPHERE | IBVH_BBOX */
    int32_t leaf_float;  /* IBVH_F32 | IBVH_F64      */
    int32_t node_kind;
    int32_t node_float;
    int32_t index_type;  /* IBVH_I32 | IBVH_I64      */
    int32_t morton_type; /* IBVH_U16 | IBVH_U32 | IBVH_U64 */
} ibvh_types;

/* ImplicitTree{I} (implicit_tree.jl:52-67), widened to int64 at the boundary. */
typedef struct ibvh_tree {
    int64_t levels;
    int64_t real_leaves;
    int64_t real_nodes;
    int64_t virtual_leaves;
    int64_t virtual_nodes;
} ibvh_tree;

/* Byte layout of one BoundingVolume{V,I,M} record. */
typedef struct ibvh_layout {
    int64_t volume_bytes; /* sizeof(V)                       */
    int64_t node_bytes;   /* sizeof(N), the node volume type */
    int64_t index_off;
    int64_t morton_off;
    int64_t leaf_bytes;   /* sizeof(BoundingVolume{V,I,M})   */
    int64_t pair_bytes;   /* sizeof(IndexPair{I})            */
} ibvh_layout;

/* A built BVH as the traversals see it: struct BVH (build.jl:155-166). */
typedef struct ibvh_bvh {
    ibvh_types types;
    ibvh_tree tree;
    int64_t built_level;
    const void *leaves; /* real_leaves x BoundingVolume records, Morton-sorted    */
    const void *nodes;  /* (real_nodes - real_leaves) x node volumes              */
    const void *skips;  /* levels x I, see ibvh_compute_skips                     */
} ibvh_bvh;

#define IBVH_MAX_SORT_LEVELS 4

/* Inputs of one BVH construction: BVH(bounding_volumes, node_type; built_level, options)
 * (build.jl:198-271) with options.morton = DefaultMortonAlgorithm (morton/default.jl:21-40). */
typedef struct ibvh_build_desc {
    ibvh_types types;
    int64_t n;               /* number of bounding volumes (>= 1)                              */
    int64_t built_level;     /* integer level 1..levels (use ibvh_compute_build_level)         */
    int32_t already_wrapped; /* 1: `leaves` already holds BoundingVolume records whose .index
                                is kept (build.jl:220-222); 0: wrap `volumes`, index = 1..n  */
    int32_t compute_extrema; /* 1: derive mins/maxs from the centres (morton/default.jl:53)    */
    double mins[3];          /* used when compute_extrema == 0 (alg.mins / alg.maxs)           */
    double maxs[3];
    /* Skewed inputs (clustered clouds, surfaces, duplicates).  The sort first splits the leaves into the cells of a
     * coarse Morton grid; a cell too crowded for one workgroup is split again, by the key bits that vary inside it, by
     * up to IBVH_MAX_SORT_LEVELS further partition levels.  For a uniform cloud those levels find nothing to do but
     * still cost their launches (4 per level, ~2 us each), so their number is the caller's choice:
     *   sort_levels = k (0 .. IBVH_MAX_SORT_LEVELS; negative or larger = all): launch k extra levels.  Whatever is
     *   still crowded after them is sorted by one workgroup per piece — always correct, slow when pieces are large.
     * skew_flag (optional, may be NULL): 4 bytes the GPU can write — device memory or mapped pinned host memory.  Every
     * build stores there, in the low byte, how many extra levels its input would have used (0 for a uniform cloud; at
     * most one more than it was given) and, in the second byte, how full the fullest cell of the coarse grid was, in
     * 1/128 of what one workgroup sorts (128 = exactly full, saturating at 255), so a caller that rebuilds every time
     * step can pass sort_levels = (the levels the previous build reported, plus one spare level when that is not 0 or
     * the fullest cell was close to full) without ever synchronising (build.jl:109-126 reuse pattern); a cold build
     * should pass 2 or more. */
    int32_t sort_levels;
    /* sort_equalize != 0: the cells of the first partition are key RANGES of about equal population (splitters taken from a
     * sorted sample of the keys) instead of the cells of a regular grid: a surface mesh or a clustered cloud fills few grid
     * cells, and every crowded one costs a second move of its records (the extra levels above); with equalised cells only
     * runs of EQUAL keys longer than a workgroup sorts still need a level.  Costs two small launches (~15 - 25 us) that a
     * cloud filling its box does not need; the result is byte-identical either way.  A build that ran with it stores bit 16
     * of skew_flag = 1 while the plain grid would have had a crowded cell: a caller that rebuilds every step asks for
     * equalised cells when the previous build reported extra levels (low byte != 0) or that bit.  Bit 17 (equalised builds
     * with sort_levels > 0): more than half of the records sat in crowded cells all the same — runs of equal keys, which
     * no choice of cells can split; such a chain is better off with the plain grid (the sample's launches buy nothing).
     * Bit 18: this build ran with equalised cells (so a caller that goes back to the plain grid on the strength of bit 16 = 0
     * knows it is doing so on an ESTIMATE, and can give that one build a spare level). */
    int32_t sort_equalize;
    void *skew_flag;
} ibvh_build_desc;

/* ----------------------------------------------------------------------------------- */
/* host-side shape math (no GPU needed)                                                 */
/* ----------------------------------------------------------------------------------- */

/* ImplicitTree{I}(num_leaves) — implicit_tree.jl:77-90.  IBVH_ERR_DOMAIN when n < 1. */
ibvh_status ibvh_tree_shape(int64_t num_leaves, ibvh_tree *out);

/* compute_skips! — implicit_tree.jl:100-113; writes tree->levels int64 values (host). */
ibvh_status ibvh_compute_skips(const ibvh_tree *tree, int64_t *skips_out);

/* memory_index / level_indices / isvirtual — implicit_tree.jl:128-199 (host helpers). */
ibvh_status ibvh_memory_index(const ibvh_tree *tree, int64_t implicit_index, int64_t *out);
ibvh_status ibvh_level_indices(const ibvh_tree *tree, int64_t level, int64_t *start, int64_t *stop);
ibvh_status ibvh_isvirtual(const ibvh_tree *tree, int64_t implicit_index, int32_t *out);

/* compute_build_level for a fractional argument — build.jl:309-325:
 * round(I, levels + (1 - levels) * frac), ties to even like Julia's round. */
ibvh_status ibvh_compute_build_level(const ibvh_tree *tree, double frac, int64_t *out);

/* Record layout of the given type combination; IBVH_ERR_UNSUPPORTED if not instantiated. */
ibvh_status ibvh_layout_of(const ibvh_types *types, ibvh_layout *out);

/* ----------------------------------------------------------------------------------- */
/* build                                                                                */
/* ----------------------------------------------------------------------------------- */

/* Scratch bytes ibvh_build needs for n leaves of the given types. */
ibvh_status ibvh_build_scratch_bytes(const ibvh_types *types, int64_t n, size_t *bytes_out);

/* The whole of BVH(...) on device — build.jl:198-271:
 *   wrap (build.jl:328-352) -> extrema (morton/utils.jl:1-72) -> Morton encode
 *   (morton/default.jl:63-157) -> stable ascending sort by .morton (build.jl:248-253)
 *   -> compute_skips! -> aggregate_oibvh! down to built_level (build.jl:366-523).
 * volumes : n x V raw volumes; with desc->already_wrapped: NULL (sort `leaves` in place) or n source
 *           BoundingVolume records that are left untouched (`leaves` then receives the sorted copy)
 * leaves  : n x BoundingVolume records, written (or sorted in place, see above)
 * nodes   : (real_nodes - real_leaves) x N
 * skips   : levels x I
 * extrema_out : optional 6 x double device->host copy target is NOT provided; pass NULL or a
 *           DEVICE pointer to 6 values of the leaf float type (mins then maxs, expanded).
 */
ibvh_status ibvh_build(const ibvh_build_desc *desc, const void *volumes, void *leaves, void *nodes,
                       void *skips, void *extrema_out, void *scratch, size_t scratch_bytes,
                       void *stream);

/* Stand-alone pieces of the build, exposed for tests, profiling and the multi-GPU driver. */

/* bounding_volumes_extrema (morton/utils.jl:55-72): 6 values of the leaf float type
 * (xmin,ymin,zmin,xmax,ymax,zmax), epsilon-expanded when `expand` != 0, written to DEVICE
 * memory `extrema_out`.  `records` are raw volumes (stride volume_bytes) when wrapped == 0. */
ibvh_status ibvh_extrema(const ibvh_types *types, const void *records, int32_t wrapped, int64_t n,
                         int32_t expand, void *extrema_out, void *scratch, size_t scratch_bytes,
                         void *stream);

/* _morton_encode! (morton/default.jl:63-108) into a separate key array (uint32 for U16/U32,
 * uint64 for U64) using DEVICE extrema (6 x leaf float, already expanded). */
ibvh_status ibvh_morton_keys(const ibvh_types *types, const void *records, int32_t wrapped,
                             int64_t n, const void *extrema, void *keys_out, void *stream);

/* Stable radix sort of (key, value=uint32) pairs (LSD passes, or one MSD partition + in-LDS bucket sort, chosen
 * from n: same result); key_bits = significant low bits.
 * keys/vals are sorted into keys_out/vals_out (may alias the alt buffers as documented in
 * DESIGN.md); key_bytes is 4 or 8. */
ibvh_status ibvh_sort_pairs(int32_t key_bytes, int32_t key_bits, int64_t n, void *keys, void *vals,
                            void *keys_alt, void *vals_alt, int32_t *result_in_alt, void *scratch,
                            size_t scratch_bytes, void *stream);
ibvh_status ibvh_sort_scratch_bytes(int32_t key_bytes, int64_t n, size_t *bytes_out);

/* aggregate_oibvh! alone (build.jl:366-523) over already-sorted leaves. */
ibvh_status ibvh_aggregate(const ibvh_types *types, const ibvh_tree *tree, int64_t built_level,
                           const void *leaves, void *nodes, void *stream);

/* ----------------------------------------------------------------------------------- */
/* leaf-vs-tree traversal (LVTTraversal, the reference default)                          */
/* ----------------------------------------------------------------------------------- */
/* The reference's own two-pass protocol (lvt/traverse_single.jl:53-75):
 *   _count : pass 1 + inclusive scan of the per-work-item counts; SYNCHRONISES the stream and
 *            returns the total (the reference's `@allowscalar thread_ncontacts[end]`, :60);
 *   _write : pass 2, contact k of work item i lands at counts[i-1] + k (1-based), which makes
 *            the contact list order deterministic and identical to the reference's.
 * counts : one I per work item (cache2 of BVHTraversal on the GPU path, :31-32).
 * scratch: ibvh_lvt_scratch_bytes() bytes of device memory: scan tile sums + the contact cache + (BBox nodes) the rows of
 *          the shared descent: one list of cut-level nodes per block of 2,048 consecutive leaves, made by one small kernel
 *          in front of the counting pass and read by both passes (csrc/ibvh_lvt.hpp "BlockRows"), and a dense copy of the
 *          work items' .index (4 / 8 bytes an item) that the counting pass leaves for the writing pass, which then does not
 *          touch the leaf records at all; a scratch without room for either is served without it.
 * NaN: with BBox nodes the walkers rely on parents being the exact minima / maxima of their children (merge.jl:30-40): a
 *      contact is decided by the leaf parent's box and the leaf test, the levels above only prune.  On volumes whose boxes
 *      hold no NaN that is the reference's list, element for element.  A NaN leaf box (a NaN radius; Inf - Inf in the
 *      sphere -> box conversion) can make merge.jl's `a < b ? a : b` produce a NaN NODE box above NaN-free leaves; the
 *      reference's walk then prunes those leaves at that node, while this library tests the node levels from level 7 down to
 *      the level of the 128-leaf subtrees and then the leaf parents: on such trees its list is a superset of the
 *      reference's (tests/test_gpu_lvt_blocks.py::test_nan_and_infinite_radii pins the relation).  Infinite boxes are exact.
 */
/* cache_slots: contacts per work item (on average) the counting pass keeps for the writing pass (0 = none: the
 * writing pass walks the tree again; 8 suits ~2 contacts per leaf).  The BBox-node and ray walkers pool the slots of
 * the 64 work items of a wave (a wave walks again only if ALL its items together found more than ~42 * cache_slots
 * contacts); the exact walk keeps the first cache_slots contacts of every item.  Pass the SAME scratch buffer and size
 * to the _count call and its _write call. */
ibvh_status ibvh_lvt_scratch_bytes(const ibvh_types *types, int64_t n_items, int32_t cache_slots,
                                   size_t *bytes_out);
ibvh_status ibvh_traverse_lvt_count(const ibvh_bvh *bvh, int64_t start_level, int32_t narrow,
                                    void *counts, int64_t *total_out, void *scratch,
                                    size_t scratch_bytes, void *stream);
ibvh_status ibvh_traverse_lvt_write(const ibvh_bvh *bvh, int64_t start_level, int32_t narrow,
                                    const void *counts, void *contacts, void *scratch,
                                    size_t scratch_bytes, void *stream);

/* traverse(bvh1, bvh2, LVTTraversal()) — lvt/traverse_pair.jl:1-244.  The BVH with more leaves
 * supplies the work items (:15-36); contacts are always (index in bvh1, index in bvh2).
 * counts needs max(n1, n2) entries. */
ibvh_status ibvh_traverse_pair_lvt_count(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2,
                                         int64_t start_level1, int64_t start_level2,
                                         int32_t narrow, void *counts, int64_t *total_out,
                                         void *scratch, size_t scratch_bytes, void *stream);
ibvh_status ibvh_traverse_pair_lvt_write(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2,
                                         int64_t start_level1, int64_t start_level2,
                                         int32_t narrow, const void *counts, void *contacts,
                                         void *scratch, size_t scratch_bytes, void *stream);

/* traverse_rays(bvh, points, directions, LVTTraversal()) — raytrace/leaf_vs_tree/
 * leaf_vs_tree.jl:1-228.  points/directions: (3, num_rays) column-major arrays of the leaf
 * float type; contacts are (leaf.index, iray); `narrow`: IBVH_NARROW_NONE / _RAY_ORIGIN_OUTSIDE (| IBVH_OUTPUT_POSITIONS).
 * Scratch: ibvh_rays_scratch_bytes().  For trees of >= 17 levels (>= 13 under batches of <= 8,192 rays; not for a tree of
 * < 2^20 leaves under more than two rays per leaf) it holds the tables
 * of the BINNED path (csrc/ibvh_lvt.hip "(3c)": the walk is cut at a level, the hits at that level are grouped by subtree
 * and the walks are finished subtree by subtree out of LDS — the reference's walk cut in two, the same hit list in the same
 * order): 40 bytes x 16 items per ray (batches of up to 8 M rays).  A call that needs more raises a flag on the device and is served by the binary
 * walker in the same launch sequence; a caller that passes a smaller buffer (ibvh_lvt_scratch_bytes) gets the binary
 * walker from the start.  Otherwise it is ibvh_lvt_scratch_bytes(num_rays work items) — plus, ONLY while the development
 * knob "rays_shadow" is set (off by default: measured slower than the binary walk), room for a quantised
 * 8-wide shadow of the node levels that the counting call builds for itself and walks instead of the binary tree (one
 * 80-byte fetch per three levels; leaf parents' exact boxes and leaves still tested exactly: the hit list, order
 * included, is unchanged — csrc/ibvh_lvt.hip "(3b)").  Pass the SAME scratch buffer and size to a _count call and its
 * _write call. */
ibvh_status ibvh_rays_scratch_bytes(const ibvh_bvh *bvh, int64_t num_rays, int32_t cache_slots, size_t *bytes_out);
ibvh_status ibvh_traverse_rays_lvt_count(const ibvh_bvh *bvh, const void *points,
                                         const void *directions, int64_t num_rays,
                                         int64_t start_level, int32_t narrow, void *counts,
                                         int64_t *total_out, void *scratch, size_t scratch_bytes,
                                         void *stream);
ibvh_status ibvh_traverse_rays_lvt_write(const ibvh_bvh *bvh, const void *points,
                                         const void *directions, int64_t num_rays,
                                         int64_t start_level, int32_t narrow, const void *counts,
                                         void *contacts, void *scratch, size_t scratch_bytes,
                                         void *stream);

/* _enqueue : _count + _write without the host read in between, for callers that already own a contact
 *   buffer (the reference's `cache=` reuse, traverse.jl:54-107): pass 1, the scan and pass 2 are enqueued
 *   back to back and NOTHING synchronises the stream; pass 2 does nothing unless the total fits
 *   `capacity` pairs.  The total is written to `total_dev`, a DEVICE pointer to one int64 owned by the caller
 *   (NULL: the first 8 bytes of `scratch`), and — when `total_host` is not NULL — also to that int64 in MAPPED
 *   PINNED HOST memory (hipHostMalloc; the store is a system-scope release, so a host that set the word to a
 *   sentinel before the call can poll it instead of synchronising the stream: the reference's blocking
 *   `@allowscalar` read, lvt/traverse_single.jl:60, without the copy and the stream sync).  Otherwise
 *   fetch it with ibvh_lvt_total() when it is needed; if it exceeds
 *   `capacity`, grow the buffer and call the matching _write (counts and scratch are ready for it).  A caller that
 *   chains traversals through one scratch buffer and reads the totals late gives every call its own `total_dev`
 *   word: the scratch (header included) is rewritten by the next call.  Replaces the same reference lines as
 *   _count/_write; only the place of the blocking read (`@allowscalar`, lvt/traverse_single.jl:60) moves. */
ibvh_status ibvh_traverse_lvt_enqueue(const ibvh_bvh *bvh, int64_t start_level, int32_t narrow,
                                      void *counts, void *contacts, int64_t capacity, void *total_dev,
                                      void *total_host, void *scratch, size_t scratch_bytes, void *stream);
ibvh_status ibvh_traverse_pair_lvt_enqueue(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2,
                                           int64_t start_level1, int64_t start_level2, int32_t narrow,
                                           void *counts, void *contacts, int64_t capacity, void *total_dev,
                                           void *total_host, void *scratch, size_t scratch_bytes, void *stream);
ibvh_status ibvh_traverse_rays_lvt_enqueue(const ibvh_bvh *bvh, const void *points,
                                           const void *directions, int64_t num_rays,
                                           int64_t start_level, int32_t narrow, void *counts, void *contacts,
                                           int64_t capacity, void *total_dev, void *total_host, void *scratch,
                                           size_t scratch_bytes, void *stream);
/* blocking read of the total a _count / _enqueue call left behind: pass the call's `total_dev`, or its `scratch`
 * when total_dev was NULL */
ibvh_status ibvh_lvt_total(const void *total_dev_or_scratch, int64_t *total_out, void *stream);

/* ----------------------------------------------------------------------------------- */
/* breadth-first traversal (BFSTraversal): level-synchronous pair queues                 */
/* ----------------------------------------------------------------------------------- */
typedef struct ibvh_bfs_result {
    int64_t num_contacts;
    int64_t num_checks;        /* BVHTraversal.num_checks (bfs/traverse_single.jl:25,48)          */
    int64_t contacts_in;       /* 1: contacts are in bvtt1, 2: in bvtt2; with IBVH_ERR_CAPACITY: the queue that
                                  holds the pairs still to be expanded (keep its first resume_num entries)     */
    int64_t required_capacity; /* pairs each queue must hold; set when IBVH_ERR_CAPACITY          */
    /* IN/OUT — resuming after IBVH_ERR_CAPACITY instead of starting over (the reference grows its queue between two
     * levels, bfs/traverse_single.jl:38-53): zero both for a fresh traversal.  With IBVH_ERR_CAPACITY the library
     * reports the step whose destination queue was too small; call again with the SAME result struct, both queues
     * grown to required_capacity (contents of the `contacts_in` queue preserved, e.g. Julia's resize!) and the same
     * `counters` buffer: the traversal continues from that step.                                                   */
    int64_t resume_step;
    int64_t resume_num;
} ibvh_bfs_result;

/* Pairs the initial queue needs: initial_bvtt (bfs/traverse_single.jl:64-99). */
ibvh_status ibvh_bfs_initial_capacity(const ibvh_bvh *bvh, int64_t start_level, int64_t *pairs_out);
ibvh_status ibvh_bfs_pair_initial_capacity(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2,
                                           int64_t start_level1, int64_t start_level2,
                                           int64_t *pairs_out);
ibvh_status ibvh_bfs_rays_initial_capacity(const ibvh_bvh *bvh, int64_t num_rays,
                                           int64_t start_level, int64_t *pairs_out);
/* Bytes of the `counters` device scratch for a traversal that walks `total_levels` levels
 * (levels of the BVH; levels1 + levels2 for a pair). */
ibvh_status ibvh_bfs_counters_bytes(int64_t total_levels, size_t *bytes_out);

/* traverse(bvh, BFSTraversal()) — bfs/traverse_single.jl:1-61.  bvtt1/bvtt2: two queues of
 * `capacity` IndexPair{I} each (cache1/cache2).  counters: DEVICE scratch of
 * ibvh_bfs_counters_bytes() bytes (overflow flag + a queue count and a check count per level).  A queue holds pairs
 * that have already passed their check: a step enumerates their children and checks them at once (the same checks as the
 * reference's, one launch earlier; num_checks is unchanged).  All levels are enqueued back to back — a
 * level takes its queue length from the device word the previous level accumulated — and the stream is synchronised
 * ONCE, at the end (the reference reads one count per level, bfs/traverse_single_gpu.jl:24).  On IBVH_ERR_CAPACITY
 * grow both queues to result->required_capacity and call again with the same `result` (see ibvh_bfs_result: the
 * traversal resumes at the level that overflowed; the reference's resize!, bfs/traverse_single.jl:40). */
ibvh_status ibvh_traverse_bfs(const ibvh_bvh *bvh, int64_t start_level, int32_t narrow, void *bvtt1,
                              void *bvtt2, int64_t capacity, void *counters,
                              ibvh_bfs_result *result, void *stream);

/* traverse(bvh1, bvh2, BFSTraversal()) — bfs/traverse_pair.jl:1-151 (six-phase descent). */
ibvh_status ibvh_traverse_pair_bfs(const ibvh_bvh *bvh1, const ibvh_bvh *bvh2, int64_t start_level1,
                                   int64_t start_level2, int32_t narrow, void *bvtt1, void *bvtt2,
                                   int64_t capacity, void *counters, ibvh_bfs_result *result,
                                   void *stream);

/* traverse_rays(bvh, points, directions, BFSTraversal()) — raytrace/breadth_first/
 * breadth_first.jl:1-66. */
ibvh_status ibvh_traverse_rays_bfs(const ibvh_bvh *bvh, const void *points, const void *directions,
                                   int64_t num_rays, int64_t start_level, int32_t narrow, void *bvtt1, void *bvtt2,
                                   int64_t capacity, void *counters, ibvh_bfs_result *result,
                                   void *stream);

/* ----------------------------------------------------------------------------------- */
/* multi-GPU build: device pieces (collectives are issued by the host side over RCCL)    */
/* ----------------------------------------------------------------------------------- */
/* Epsilon expansion of bounding_volumes_extrema (morton/utils.jl:63-69) applied in place to 6
 * already reduced values (mins then maxs) of float type `flt` in DEVICE memory: used after the
 * all-reduce of the per-GPU extrema (ibvh_extrema with expand = 0). */
ibvh_status ibvh_expand_extrema(int32_t flt, void *extrema, void *stream);

/* The all-reduce(MAX) vector of the distributed build, assembled / taken apart on device (one launch each, no
 * host round trip): vec_out (DEVICE, 6 + nranks float64) = [-mins(3), maxs(3), one-hot leaf counts]; `extrema`:
 * this GPU's 6 UNEXPANDED centre extrema of float type `flt` (ibvh_extrema with expand = 0), ignored when
 * has_data == 0 (neutral elements of morton/utils.jl:29-40 are written).  _unpack converts the reduced vector back
 * to `flt` and applies the epsilon expansion (morton/utils.jl:63-69). */
ibvh_status ibvh_dist_pack_extrema(int32_t flt, const void *extrema, int32_t has_data, int32_t rank,
                                   int32_t nranks, int64_t n_local, void *vec_out, void *stream);
ibvh_status ibvh_dist_unpack_extrema(int32_t flt, const void *vec, void *extrema_out, void *stream);

/* Stable partition of the local leaves by destination rank (keys in [splitters[r-1], splitters[r]) go to rank r;
 * `splitters`: nranks - 1 ascending keys in HOST memory): perm_out[j] (DEVICE, n x uint32) = source position of the
 * j-th leaf in (destination, source position) order; counts_out (DEVICE, nranks x uint64, optional): leaves per
 * destination rank — for a caller that does not already know its row of the send matrix.  nranks <= 256. */
ibvh_status ibvh_dist_partition_scratch_bytes(int64_t n, size_t *bytes_out);
ibvh_status ibvh_dist_partition(int32_t key_bytes, const void *keys, int64_t n, const uint64_t *splitters,
                                int32_t nranks, void *perm_out, void *counts_out, void *scratch,
                                size_t scratch_bytes, void *stream);

/* Digit histograms for the splitter search of the distributed radix sort.  out (DEVICE,
 * max(nprefix,1) x 2^bits uint32, zeroed here): out[j][d] = number of keys whose
 * (key >> prefix_shift) == prefixes[j] and whose digit (key >> shift) & (2^bits - 1) == d;
 * nprefix == 0 counts all keys.  bits <= 12, nprefix <= 15; `prefixes` is a HOST array. */
ibvh_status ibvh_key_histogram(int32_t key_bytes, const void *keys, int64_t n, int32_t shift,
                               int32_t bits, int32_t prefix_shift, const uint64_t *prefixes,
                               int32_t nprefix, void *out, void *stream);

/* Pack BoundingVolume records for the exchange: out[i] = { volumes[p], index_base + p + 1,
 * keys[p] } with p = perm[i] (uint32) or i when perm == NULL. */
ibvh_status ibvh_pack_records(const ibvh_types *types, const void *volumes, const void *keys,
                              const void *perm, int64_t index_base, int64_t n, void *records_out,
                              void *stream);

/* ----------------------------------------------------------------------------------- */
/* multi-GPU build: the driver (round 4).  No reference counterpart: ImplicitBVH.jl is    */
/* single-device; BASELINE.json's north star shards the BUILD over the GPUs of one node   */
/* ("RCCL allreduce over xGMI for the global AABB and a distributed radix-sort exchange") */
/* and keeps the host in Julia behind "a thin C-ABI (ccall) shim" — this is that shim.    */
/* ----------------------------------------------------------------------------------- */
#define IBVH_DIST_MAX_RANKS 256
enum { IBVH_COMM_F64 = 0, IBVH_COMM_I64 = 1, IBVH_COMM_I32 = 2 };  /* element type of an all-reduce */
enum { IBVH_COMM_MAX = 0, IBVH_COMM_SUM = 1, IBVH_COMM_MIN = 2 };  /* its operation                 */

/* Collectives the driver needs, as a vtable: every rank calls the same sequence.  Buffers are DEVICE pointers; a call is
 * ordered behind earlier work on `stream` and later work on `stream` sees its result (RCCL semantics).  Return 0 on
 * success.  all_to_all_v: `send` is partitioned contiguously by destination (send_bytes[p] bytes for rank p), `recv`
 * contiguously by source (recv_bytes[p]); both count arrays are HOST arrays of `size` entries. */
typedef struct ibvh_comm {
    void *ctx;
    int32_t rank, size;
    int32_t (*all_reduce)(void *ctx, void *buf, int64_t count, int32_t dtype, int32_t op, void *stream); /* in place */
    int32_t (*all_gather)(void *ctx, const void *send, void *recv, int64_t bytes_per_rank, void *stream);
    int32_t (*all_to_all_v)(void *ctx, const void *send, const int64_t *send_bytes, void *recv, const int64_t *recv_bytes,
                            void *stream);
} ibvh_comm;

/* The vtable over RCCL for an ncclComm_t the caller created (ncclCommInitRank), collectives issued on the stream each
 * call is handed.  librccl.so is resolved with dlopen on first use (libibvh.so does not link it): IBVH_ERR_UNSUPPORTED
 * when it cannot be found.  all_to_all_v = grouped ncclSend / ncclRecv: xGMI is point-to-point. */
ibvh_status ibvh_comm_from_rccl(void *nccl_comm, int32_t rank, int32_t size, ibvh_comm *out);

/* Splitter search of the distributed radix sort — HOST arithmetic only (no GPU), identical on every rank: keys
 * k_1 <= ... <= k_{P-1}, rank r receives the keys in [k_r, k_{r+1}).  Protocol: init; then, until all_done: make the GLOBAL
 * histogram of the digit (key >> next_shift) & (2^next_bits - 1) — one row over all keys at level 0, later one row per
 * prefix rows[j] (keys with key >> (next_shift + next_bits) == rows[j]), num_rows rows — and hand it to _step.  A splitter
 * stops refining once the bucket it landed in holds at most tolerance * n_global / size keys (0: full key resolution). */
typedef struct ibvh_splitter_search {
    int32_t size, key_bits, decided /* key bits decided so far */, all_done;
    int32_t next_bits, next_shift, num_rows, reserved_;
    int64_t n_global;
    double tolerance;
    uint64_t rows[IBVH_DIST_MAX_RANKS];      /* prefixes the next histogram needs, ascending */
    uint64_t prefix[IBVH_DIST_MAX_RANKS];    /* per splitter: the bits decided so far        */
    uint64_t splitters[IBVH_DIST_MAX_RANKS]; /* per splitter: the final key, once done       */
    int64_t below[IBVH_DIST_MAX_RANKS];      /* per splitter: global number of keys strictly below its decided prefix range */
    int32_t row_of[IBVH_DIST_MAX_RANKS];     /* per splitter: its row of the next histogram  */
    uint8_t done[IBVH_DIST_MAX_RANKS];
} ibvh_splitter_search;
ibvh_status ibvh_splitter_search_init(ibvh_splitter_search *s, int32_t size, int32_t key_bits, int64_t n_global, double tolerance);
ibvh_status ibvh_splitter_search_step(ibvh_splitter_search *s, const int64_t *hist);

/* What ibvh_dist_plan found out (HOST memory). */
typedef struct ibvh_dist_plan_t {
    int32_t size, levels_used /* key bits the splitter search decided */;
    int64_t n_local, n_global;
    int64_t base;             /* global 0-based number of this rank's first local leaf (records carry base + p + 1)   */
    int64_t n_slice;          /* records this rank receives = leaves of its slice of the globally sorted sequence    */
    int64_t record_bytes;
    double extrema[6];        /* global centre extrema, epsilon-expanded: mins / maxs for the local ibvh_build        */
    uint64_t splitters[IBVH_DIST_MAX_RANKS];
    int64_t send_counts[IBVH_DIST_MAX_RANKS], recv_counts[IBVH_DIST_MAX_RANKS]; /* records per peer */
} ibvh_dist_plan_t;

/* Scratch for ibvh_dist_plan + ibvh_dist_exchange (the same buffer must be passed to both, untouched in between: the plan
 * leaves the keys and the partition's permutation there). */
ibvh_status ibvh_dist_scratch_bytes(const ibvh_types *types, int64_t n_local, int32_t size, size_t *bytes_out);

/* Everything up to the exchange (see csrc/ibvh_distdrv.hip): global AABB (one all-reduce(MAX) that also carries every
 * rank's leaf count), Morton keys, splitters (one all-gather; further all-reduce(SUM) levels only while a splitter's bucket
 * is heavier than `tolerance` of a shard — 0.005 is the Python mirror's default), stable partition by destination.  BLOCKS
 * on `stream` once (twice when the send matrix does not follow from the first histogram): the record counts must reach the
 * host before RCCL can be told the transfer sizes — the reference's own "count, then size, then write" shape.
 * IBVH_ERR_DOMAIN (on EVERY rank, so that none is left waiting in a collective) when there are fewer leaves than ranks or a
 * rank would receive none. */
ibvh_status ibvh_dist_plan(const ibvh_types *types, const ibvh_comm *comm, const void *volumes, int64_t n_local, double tolerance,
                           void *scratch, size_t scratch_bytes, ibvh_dist_plan_t *plan_out, void *stream);

/* Pack the local leaves into BoundingVolume records with GLOBAL 1-based indices and exchange them: ONE all-to-all.
 * records_out: DEVICE, plan->n_slice records, grouped by source rank in source order — so the stable local sort of
 * ibvh_build (already_wrapped = 1, compute_extrema = 0, mins / maxs = plan->extrema) keeps global input order among equal
 * keys, and rank r ends with the r-th slice of the single-device sorted sequence, bit for bit.  Asynchronous on `stream`. */
ibvh_status ibvh_dist_exchange(const ibvh_types *types, const ibvh_comm *comm, const void *volumes, const ibvh_dist_plan_t *plan,
                               void *scratch, size_t scratch_bytes, void *records_out, void *stream);

/* Cross-shard contact completion: the contacts between leaves of DIFFERENT slices, which the
 * per-slice self-traversals cannot see.  Root boxes and leaf counts of all slices are all-gathered; for every pair of slices
 * r < s whose boxes touch, rank s sends rank r the leaves whose own box touches one of r's boxes (a slice is described by
 * <= 16 node boxes of its tree, refined from the root by always splitting the largest: a Morton slice is not convex) — a thin
 * shell of its slice, not its tree — in ONE all_to_all_v over the same vtable; rank r builds an ordinary BVH over each set it received
 * (ibvh_build, in place) and runs the ordinary pair traversal (traverse(bvh_r, bvh_s), lvt/traverse_pair.jl) against it.
 * Per-slice self contacts + these pairs = the contact set of the whole cloud, every pair once.  Count -> size -> write:
 *   _plan     collective (one all-gather, one 8-bytes-a-peer all_to_all_v); ONE host synchronisation (every slice's boxes and how
 *             many leaves every peer gets, together); a rank with unacceptable arguments still takes part — its status travels in
 *             its record — and EVERY rank returns an error: its own, or IBVH_ERR_PEER; fills the plan:
 *             export_bytes / import_bytes (the two buffers the caller hands to _exchange) and scratch_bytes (the scratch of
 *             _count / _write: traversal scratch per imported set, cache_slots as in ibvh_lvt_scratch_bytes, + the build's).
 *             `scratch` here and in _exchange: IBVH_DIST_CROSS_SCRATCH(size) bytes of device memory.
 *   _exchange collective (every rank calls it, also one that neither sends nor receives); asynchronous on `stream`.
 *   _count    per imported set: ibvh_build + the pair traversal's counting pass; totals_out[k] (may be NULL), *total_out: pairs.
 *   _write    contacts_out: *total_out IndexPair{I}, the pairs against imported set 0 first: (index in THIS slice, index in
 *             the other slice), both GLOBAL 1-based leaf numbers (the records carry them).  `totals`: what _count returned.
 * The BVH must be fully built (built_level = 1).  The pairs are a SET: the order in which a sender's boundary leaves arrive is
 * not deterministic, so leaves of equal Morton codes — and their cross contacts — may change places from run to run (the per-slice
 * lists keep the reference's order).  Errors inside a collective sequence that are not argument errors (a failed HIP call, a
 * failed collective; a rank without a communicator or scratch): the caller must abort the communicator — a rank that returns
 * early leaves its peers waiting. */
#define IBVH_DIST_CROSS_BOXES 16
typedef struct ibvh_dist_cross_plan_t {
    int32_t size, rank, n_recv /* leaf sets this rank imports */, cache_slots;
    int64_t import_bytes, scratch_bytes, export_bytes, build_offset /* where the build's scratch starts in the scratch */;
    int32_t recv_rank[IBVH_DIST_MAX_RANKS];      /* [n_recv] ascending: the ranks leaves are imported from                   */
    int64_t recv_leaves[IBVH_DIST_MAX_RANKS];    /* [n_recv] how many                                                        */
    int64_t recv_offset[IBVH_DIST_MAX_RANKS];    /* [n_recv] byte offset of that set's leaves in the import buffer (the sets are contiguous; the room for their trees' nodes and skips follows the last set) */
    int64_t scratch_offset[IBVH_DIST_MAX_RANKS]; /* [n_recv] byte offset of its counts + traversal scratch                   */
    int64_t slice_leaves[IBVH_DIST_MAX_RANKS];   /* [size] leaves of every rank's slice                                      */
    int32_t touches[IBVH_DIST_MAX_RANKS];        /* [size] 1: this rank's root box touches rank r's (r != rank)              */
    int64_t send_leaves[IBVH_DIST_MAX_RANKS];    /* [size] own leaves rank r gets (r < rank, boxes touching r's root box)    */
    int64_t send_offset[IBVH_DIST_MAX_RANKS];    /* [size] where they are compacted in the export buffer (contiguous, by rank) */
    int32_t n_boxes[IBVH_DIST_MAX_RANKS];        /* [size] boxes that describe rank r's slice (1 .. IBVH_DIST_CROSS_BOXES)   */
    double boxes[IBVH_DIST_MAX_RANKS][IBVH_DIST_CROSS_BOXES][6]; /* [size] ... node boxes of its tree, refined greedily (lo, up) */
} ibvh_dist_cross_plan_t;
/* device bytes _plan and _exchange need as `scratch` for `size` ranks */
#define IBVH_DIST_CROSS_SCRATCH(size) ((size_t)(IBVH_DIST_CROSS_BOXES * 48 + 16) * ((size_t)(size) + 1) + (size_t)16 * (size_t)(size) + 512)
ibvh_status ibvh_dist_cross_plan(const ibvh_comm *comm, const ibvh_bvh *bvh, int32_t cache_slots, void *scratch, size_t scratch_bytes,
                                 ibvh_dist_cross_plan_t *plan_out, void *stream);
ibvh_status ibvh_dist_cross_exchange(const ibvh_comm *comm, const ibvh_bvh *bvh, const ibvh_dist_cross_plan_t *plan, void *export_buf,
                                     void *import_buf, void *scratch, size_t scratch_bytes, void *stream);
ibvh_status ibvh_dist_cross_count(const ibvh_bvh *bvh, const ibvh_dist_cross_plan_t *plan, void *import_buf, void *scratch,
                                  size_t scratch_bytes, int64_t *totals_out, int64_t *total_out, void *stream);
ibvh_status ibvh_dist_cross_write(const ibvh_bvh *bvh, const ibvh_dist_cross_plan_t *plan, const void *import_buf, void *scratch,
                                  size_t scratch_bytes, const int64_t *totals, void *contacts_out, void *stream);

/* Release what ibvh_comm_from_rccl allocated for `comm` (the ncclComm_t itself stays the caller's). */
ibvh_status ibvh_comm_release(ibvh_comm *comm);

/* ----------------------------------------------------------------------------------- */
/* input preparation adjacent to the path                                               */
/* ----------------------------------------------------------------------------------- */
/* BSphere{T}(p1,p2,p3) (bsphere.jl:43-112) / BBox{T}(p1,p2,p3) (bbox.jl:59-70) for n triangles
 * stored as n x 9 values (p1 p2 p3) of float type `flt`; writes n volumes of `kind`. */
ibvh_status ibvh_volumes_from_triangles(int32_t kind, int32_t flt, const void *triangles, int64_t n,
                                        void *volumes_out, void *stream);

/* Deterministic synthetic inputs shared by bench, tests and the oracle (SplitMix64 counter
 * based).  Writes n BSphere{F32} with centres uniform in
 * [origin, origin+extent)^3 and radius r0*(0.5+0.5u). */
ibvh_status ibvh_generate_spheres_f32(int64_t n, uint64_t seed, int64_t first_index,
                                      const float origin[3], const float extent[3], float r0,
                                      void *volumes_out, void *stream);

/* Per-launch timing, used by bench.py for the roofline figures: when enabled, every kernel the
 * library launches is bracketed by HIP events on its launch stream.  _get synchronises on the
 * record's stop event; names are the kernel instantiations' source spellings. */
ibvh_status ibvh_profile_enable(int32_t on); /* also clears the records */
ibvh_status ibvh_profile_count(int64_t *count_out);
ibvh_status ibvh_profile_get(int64_t i, const char **name_out, float *ms_out);

/* Work of ONE counting pass of the leaf-vs-tree walkers (SURVEY.md §8d "touched bytes"; the reference's own counter,
 * BVHTraversal.num_checks, exists for BFS only: bfs/traverse_single.jl:25,48).  Runs the counting pass of the walk the
 * ordinary entry points would take — self (bvh2 = points = NULL), pair (bvh2), or rays (points / directions, num_rays) —
 * in an instantiation that also counts, and leaves in `work_out` (DEVICE, 4 x uint64, zeroed here):
 *   [0] node tests   (one lane-level box-box / ray-box test against a node volume)
 *   [1] leaf tests   (one exact leaf-leaf / ray-leaf test)
 *   [2] node records fetched   [3] leaf records fetched
 * `counts`: per-work-item counts as for the _count calls (max(n1, n2) / num_rays / n entries), not scanned.
 * Measurement only; instantiated for BSphere{Float32} leaves, BBox{Float32} nodes, Int32 indices, default start
 * levels, no `narrow` (anything else: IBVH_ERR_UNSUPPORTED). */
ibvh_status ibvh_lvt_work_counters(const ibvh_bvh *bvh, const ibvh_bvh *bvh2, const void *points,
                                   const void *directions, int64_t num_rays, void *counts, void *work_out,
                                   void *stream);

/* Development knobs, for measurements and tests only (the defaults are the shipped behaviour; results never depend
 * on them, only speed and which code path is taken).  One process-wide table: set a knob BEFORE the calls it should
 * affect and not concurrently with them.  The library never reads the environment.  Names (exactly the table in
 * csrc/ibvh_core.hip; meanings: csrc/ibvh_common.hpp, struct Tuning): "ray_block", "lvt_wide", "lvt_xcd", "sort_tile",
 * "sort_lsd", "sort_msd_avg", "bucket_tpb", "msd", "msd_bits", "msd_cap", "msd_tile", "msd_ftpb", "msd_avg",
 * "msd_range", "msd_equalize", "msd_rescue", "lvt_scan_fused", "bfs_wg_per_cu", "lvt_blocks", "lvt_block_shift",
 * "lvt_blocks_min_items", "lvt_blocks_paired_below", "rays_binned" (1 = the binned ray path where it pays, 2 =
 * wherever the tree allows it, 0 = never), "rays_fast_slab", "rays_subtree_depth", "rays_items_per_ray", "rays_tail",
 * "msd_resident_kb", "msd_finish_pad_kb".  Unknown name: IBVH_ERR_INVALID_ARG — that includes the names of
 * development variants whose kernels are not in libibvh.so (variants/ builds with -DIBVH_VARIANTS add their own). */
ibvh_status ibvh_set_tuning(const char *name, int32_t value);
ibvh_status ibvh_get_tuning(const char *name, int32_t *value_out);

/* Library / device introspection. */
const char *ibvh_version(void);
const char *ibvh_status_string(int32_t status);

#ifdef __cplusplus
}
#endif
#endif /* IBVH_H */
