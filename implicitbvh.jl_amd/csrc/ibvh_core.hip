// ibvh_core.hip — host-side shape math of libibvh (no kernels): ImplicitTree, skips, layouts.
// Reference: src/implicit_tree.jl, src/build.jl:309-325.
#include <cmath>

#include "ibvh_common.hpp"

#include <mutex>
#include <string>
#include <vector>

using namespace ibvh;

// ---- per-launch timing ---------------------------------------------------------------------
namespace ibvh {
namespace prof {
bool enabled = false;
struct Rec {
    std::string name;
    hipEvent_t a, b;
};
static std::vector<Rec> recs;
static std::vector<hipEvent_t> pool;
static std::mutex mu; // launches may come from several host threads (one stream each)
static thread_local size_t last = 0; // the record begin() of THIS thread opened
static hipEvent_t take() {
    hipEvent_t e;
    if (!pool.empty()) {
        e = pool.back();
        pool.pop_back();
        return e;
    }
    (void)hipEventCreate(&e);
    return e;
}
void begin(const char *name, hipStream_t st) {
    std::lock_guard<std::mutex> g(mu);
    Rec r{name, take(), take()};
    (void)hipEventRecord(r.a, st);
    recs.push_back(r);
    last = recs.size() - 1;
}
void end(hipStream_t st) {
    std::lock_guard<std::mutex> g(mu);
    if (last < recs.size()) (void)hipEventRecord(recs[last].b, st);
}
static void reset() {
    std::lock_guard<std::mutex> g(mu);
    for (auto &r : recs) {
        pool.push_back(r.a);
        pool.push_back(r.b);
    }
    recs.clear();
}
} // namespace prof
} // namespace ibvh

namespace ibvh {
Tuning g_tuning;
}
namespace {
struct Knob {
    const char *name;
    int Tuning::*field;
};
const Knob kKnobs[] = {
    {"ray_block", &Tuning::ray_block},   {"lvt_wide", &Tuning::lvt_wide},     {"lvt_xcd", &Tuning::lvt_xcd},
    {"sort_tile", &Tuning::sort_tile},   {"sort_lsd", &Tuning::sort_lsd},     {"sort_msd_avg", &Tuning::sort_msd_avg},
    {"bucket_tpb", &Tuning::bucket_tpb}, {"msd", &Tuning::msd},               {"msd_bits", &Tuning::msd_bits},
    {"msd_cap", &Tuning::msd_cap},       {"msd_tile", &Tuning::msd_tile},     {"msd_ftpb", &Tuning::msd_ftpb},
    {"msd_avg", &Tuning::msd_avg},       {"msd_range", &Tuning::msd_range},   {"msd_equalize", &Tuning::msd_equalize}, {"msd_rescue", &Tuning::msd_rescue}, {"lvt_scan_fused", &Tuning::lvt_scan_fused}, {"bfs_wg_per_cu", &Tuning::bfs_wg_per_cu},
    {"lvt_blocks", &Tuning::lvt_blocks}, {"lvt_block_shift", &Tuning::lvt_block_shift}, {"lvt_blocks_min_items", &Tuning::lvt_blocks_min_items}, {"lvt_blocks_paired_below", &Tuning::lvt_blocks_paired_below},
    {"rays_binned", &Tuning::rays_binned}, {"rays_fast_slab", &Tuning::rays_fast_slab}, {"rays_subtree_depth", &Tuning::rays_subtree_depth},
    {"rays_items_per_ray", &Tuning::rays_items_per_ray}, {"rays_tail", &Tuning::rays_tail}, {"msd_resident_kb", &Tuning::msd_resident_kb}, {"msd_finish_pad_kb", &Tuning::msd_finish_pad_kb},
#ifdef IBVH_VARIANTS // (development builds only: the kernels behind these knobs are not in libibvh.so, variants/*.inc)
    {"rays_shadow", &Tuning::rays_shadow}, {"lvt_dual", &Tuning::lvt_dual},
#endif
};
inline int64_t ilog2_down(int64_t n) { return 63 - __builtin_clzll((unsigned long long)n); }
} // namespace

extern "C" {

// ImplicitTree{I}(num_leaves) — implicit_tree.jl:77-90
ibvh_status ibvh_tree_shape(int64_t num_leaves, ibvh_tree *out) {
    if (!out) return IBVH_ERR_INVALID_ARG;
    if (num_leaves < 1) return IBVH_ERR_DOMAIN;
    if (num_leaves > (int64_t(1) << 60)) return IBVH_ERR_INVALID_ARG;
    int64_t lr = num_leaves;
    int64_t fl = ilog2_down(lr);
    int64_t levels = ((lr & (lr - 1)) == 0 ? fl : fl + 1) + 1;
    int64_t lv = (int64_t(1) << (levels - 1)) - lr;
    out->levels = levels;
    out->real_leaves = lr;
    out->real_nodes = 2 * lr - 1 + popc64(lv);
    out->virtual_leaves = lv;
    out->virtual_nodes = 2 * lv - popc64(lv);
    return IBVH_OK;
}

// compute_skips! — implicit_tree.jl:100-113
ibvh_status ibvh_compute_skips(const ibvh_tree *tree, int64_t *skips_out) {
    if (!tree || !skips_out) return IBVH_ERR_INVALID_ARG;
    for (int64_t level = 1; level <= tree->levels; ++level)
        skips_out[level - 1] = level_skips(tree->levels, tree->virtual_leaves, level);
    return IBVH_OK;
}

// memory_index — implicit_tree.jl:128-148
ibvh_status ibvh_memory_index(const ibvh_tree *tree, int64_t implicit_index, int64_t *out) {
    if (!tree || !out) return IBVH_ERR_INVALID_ARG;
    if (!(1 <= implicit_index && implicit_index <= (int64_t(1) << tree->levels) - 1)) return IBVH_ERR_INVALID_ARG;
    int64_t level = ilog2_down(implicit_index) + 1;
    *out = implicit_index - level_skips(tree->levels, tree->virtual_leaves, level);
    return IBVH_OK;
}

// level_indices — implicit_tree.jl:156-171
ibvh_status ibvh_level_indices(const ibvh_tree *tree, int64_t level, int64_t *start, int64_t *stop) {
    if (!tree || !start || !stop) return IBVH_ERR_INVALID_ARG;
    if (!(1 <= level && level <= tree->levels)) return IBVH_ERR_INVALID_ARG;
    *start = level_start(tree->levels, tree->virtual_leaves, level);
    *stop = *start + level_num_real(tree->levels, tree->virtual_leaves, level) - 1;
    return IBVH_OK;
}

// isvirtual — implicit_tree.jl:179-199
ibvh_status ibvh_isvirtual(const ibvh_tree *tree, int64_t implicit_index, int32_t *out) {
    if (!tree || !out) return IBVH_ERR_INVALID_ARG;
    if (!(1 <= implicit_index && implicit_index <= (int64_t(1) << tree->levels) - 1)) return IBVH_ERR_INVALID_ARG;
    int64_t level = ilog2_down(implicit_index) + 1;
    int64_t first = int64_t(1) << (level - 1);
    *out = (implicit_index - first + 1 > level_num_real(tree->levels, tree->virtual_leaves, level)) ? 1 : 0;
    return IBVH_OK;
}

// compute_build_level(tree, built_level::AbstractFloat) — build.jl:316-318
ibvh_status ibvh_compute_build_level(const ibvh_tree *tree, double frac, int64_t *out) {
    if (!tree || !out) return IBVH_ERR_INVALID_ARG;
    if (!(0.0 <= frac && frac <= 1.0)) return IBVH_ERR_INVALID_ARG; // @argcheck 0 <= built_level <= 1
    double x = double(tree->levels) + double(1 - tree->levels) * frac;
    *out = (int64_t)std::nearbyint(x); // Julia round(): ties to even
    return IBVH_OK;
}

ibvh_status ibvh_layout_of(const ibvh_types *types, ibvh_layout *out) {
    if (!types || !out) return IBVH_ERR_INVALID_ARG;
    return layout_of(*types, *out) ? IBVH_OK : IBVH_ERR_UNSUPPORTED;
}

ibvh_status ibvh_profile_enable(int32_t on) {
    prof::reset();
    prof::enabled = on != 0;
    return IBVH_OK;
}
ibvh_status ibvh_profile_count(int64_t *count_out) {
    if (!count_out) return IBVH_ERR_INVALID_ARG;
    *count_out = (int64_t)prof::recs.size();
    return IBVH_OK;
}
ibvh_status ibvh_profile_get(int64_t i, const char **name_out, float *ms_out) {
    if (!name_out || !ms_out || i < 0 || i >= (int64_t)prof::recs.size()) return IBVH_ERR_INVALID_ARG;
    auto &r = prof::recs[(size_t)i];
    if (hipEventSynchronize(r.b) != hipSuccess) return IBVH_ERR_HIP;
    if (hipEventElapsedTime(ms_out, r.a, r.b) != hipSuccess) return IBVH_ERR_HIP;
    *name_out = r.name.c_str();
    return IBVH_OK;
}

ibvh_status ibvh_set_tuning(const char *name, int32_t value) {
    if (!name) return IBVH_ERR_INVALID_ARG;
    for (const Knob &k : kKnobs)
        if (std::string(name) == k.name) {
            g_tuning.*(k.field) = value;
            return IBVH_OK;
        }
    return IBVH_ERR_INVALID_ARG;
}
ibvh_status ibvh_get_tuning(const char *name, int32_t *value_out) {
    if (!name || !value_out) return IBVH_ERR_INVALID_ARG;
    for (const Knob &k : kKnobs)
        if (std::string(name) == k.name) {
            *value_out = g_tuning.*(k.field);
            return IBVH_OK;
        }
    return IBVH_ERR_INVALID_ARG;
}

const char *ibvh_version(void) { return "libibvh 0.3.0 (gfx950)"; }
int32_t ibvh_abi_version(void) { return IBVH_ABI_VERSION; }

const char *ibvh_status_string(int32_t status) {
    switch (status) {
    case IBVH_OK: return "ok";
    case IBVH_ERR_INVALID_ARG: return "invalid argument";
    case IBVH_ERR_DOMAIN: return "domain error: must have at least one geometry";
    case IBVH_ERR_UNSUPPORTED: return "unsupported type combination";
    case IBVH_ERR_CAPACITY: return "caller buffer too small";
    case IBVH_ERR_OVERFLOW: return "count overflows the index type";
    case IBVH_ERR_HIP: return "HIP runtime error";
    case IBVH_ERR_SCRATCH: return "scratch buffer too small";
    case IBVH_ERR_PEER: return "another rank's arguments were not acceptable";
    }
    return "unknown status";
}

} // extern "C"
