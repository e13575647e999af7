// ibvh_distdrv.hip — the distributed build behind the C ABI (north star: "host code stays Julia, calling … through a thin
// C-ABI (ccall) shim"; the build "shards leaves across the 8 GPUs of one node with RCCL allreduce over xGMI for the global AABB
// and a distributed radix-sort exchange").  The reference is single-device: there is no reference file for this layer.
//
// One process per GPU.  A rank calls
//   ibvh_dist_plan      local extrema -> ONE all-reduce(MAX) of [-mins, maxs, one-hot leaf counts] -> global AABB (bit-identical
//                       to the single-device reduce: min / max are exact) -> Morton keys -> 12-bit digit histogram, ALL-GATHERED
//                       -> ONE device -> host copy -> splitters (refined by further 12-bit levels, one all-reduce(SUM) each, only
//                       while a splitter's bucket is heavier than `tolerance` of a shard) -> stable partition by destination
//                       (ibvh_dist_partition) -> how many records this rank sends / receives;
//   (the caller sizes its record array from plan->n_slice — the reference's "count, then size, then write" protocol)
//   ibvh_dist_exchange  pack records with GLOBAL 1-based indices, ONE all-to-all over xGMI;
//   ibvh_build          ordinary local build over the received records (already_wrapped, fixed global extrema): rank r ends
//                       with the r-th slice of the globally stable-sorted sequence.
// Collectives go through a small vtable (ibvh_comm): ibvh_comm_from_rccl() fills it for an ncclComm_t (librccl.so is
// resolved with dlopen on first use: libibvh.so itself does not link RCCL); hosts without RCCL bindings and the CPU /
// virtual-rank tests inject their own.  The splitter arithmetic is host-only (ibvh_splitter_search_*): the gloo tests on
// CPU drive the very same code with histograms made by numpy.
#include <dlfcn.h>

#include <cmath>
#include <limits>
#include <cstring>
#include <mutex>
#include <vector>

#include "ibvh_common.hpp"

using namespace ibvh;

namespace {

constexpr int DIGIT_BITS = 12;
constexpr int MAX_RANKS = IBVH_DIST_MAX_RANKS;

int key_bits_of(const ibvh_types &t) { return t.morton_type == IBVH_U16 ? 15 : (t.morton_type == IBVH_U32 ? 30 : 63); }
int key_bytes_of(const ibvh_types &t) { return t.morton_type == IBVH_U64 ? 8 : 4; }
int float_bytes(int flt) { return flt == IBVH_F64 ? 8 : 4; }

__global__ void widen_u32_i64_kernel(const uint32_t *in, long long *out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (long long)in[i];
}

// where everything lives in the caller's scratch: a function of (types, n_local, size) only, so that ibvh_dist_exchange
// finds what ibvh_dist_plan left
struct Layout {
    size_t ext_local, vec, ext, hist, hist64, allh, cnt, keys, perm, part, records, ext_scratch, total;
    size_t part_bytes;
};
Layout make_layout(const ibvh_types &t, int64_t n, int size) {
    ibvh_layout lay;
    layout_of(t, lay);
    Layout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (size_t)align_up((int64_t)(bytes ? bytes : 8), 256);
        return at;
    };
    L.ext_local = take(64);
    L.vec = take((size_t)(7 + size) * 8); // (-mins, maxs, one-hot leaf counts, the largest status of all ranks)
    L.ext = take(64);
    L.hist = take((size_t)16 * (1 << DIGIT_BITS) * 4);
    L.hist64 = take((size_t)16 * (1 << DIGIT_BITS) * 8);
    L.allh = take((size_t)size * (1 << DIGIT_BITS) * 4);
    L.cnt = take((size_t)4 * size * 8);
    L.ext_scratch = take((size_t)1 << 17);
    L.keys = take((size_t)n * key_bytes_of(t));
    L.perm = take((size_t)n * 4);
    ibvh_dist_partition_scratch_bytes(n, &L.part_bytes);
    L.part = take(L.part_bytes);
    L.records = take((size_t)n * lay.leaf_bytes);
    L.total = off;
    return L;
}

// ---- RCCL, resolved lazily ----------------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool ok = false;
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // The library that CREATED the caller's ncclComm_t must serve it: a librccl already loaded into the process (torch
        // bundles its own copy) is found through the global scope first; dlopen by name is the fallback for hosts that
        // have not loaded one yet (a communicator must not cross two copies of the library).
        if (dlsym(RTLD_DEFAULT, "ncclAllReduce") != nullptr) {
            r.lib = RTLD_DEFAULT;
        } else {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (r.lib) break;
            }
            if (!r.lib) return;
        }
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
        r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
        r.Send = (decltype(r.Send))dlsym(r.lib, "ncclSend");
        r.Recv = (decltype(r.Recv))dlsym(r.lib, "ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
        r.ok = r.AllReduce && r.AllGather && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
    });
    return r;
}
// nccl.h: ncclUint8 = 1, ncclInt32 = 2, ncclInt64 = 4, ncclFloat64 = 8; ncclSum = 0, ncclMax = 2, ncclMin = 3
struct RcclCtx {
    void *comm;
    int rank, size;
};
int32_t rccl_all_reduce(void *ctx, void *buf, int64_t count, int32_t dtype, int32_t op, void *stream) {
    const RcclCtx *c = (const RcclCtx *)ctx;
    const int dt = dtype == IBVH_COMM_F64 ? 8 : (dtype == IBVH_COMM_I64 ? 4 : 2);
    const int ro = op == IBVH_COMM_MAX ? 2 : (op == IBVH_COMM_MIN ? 3 : 0);
    return rccl().AllReduce(buf, buf, (size_t)count, dt, ro, c->comm, (hipStream_t)stream) == 0 ? 0 : IBVH_ERR_HIP;
}
int32_t rccl_all_gather(void *ctx, const void *send, void *recv, int64_t bytes, void *stream) {
    const RcclCtx *c = (const RcclCtx *)ctx;
    return rccl().AllGather(send, recv, (size_t)bytes, 1, c->comm, (hipStream_t)stream) == 0 ? 0 : IBVH_ERR_HIP;
}
int32_t rccl_all_to_all_v(void *ctx, const void *send, const int64_t *send_bytes, void *recv, const int64_t *recv_bytes, void *stream) {
    const RcclCtx *c = (const RcclCtx *)ctx;
    Rccl &r = rccl();
    // grouped point-to-point transfers: xGMI is point-to-point, every pair of GPUs has its own link(s)
    if (r.GroupStart() != 0) return IBVH_ERR_HIP;
    size_t so = 0, ro = 0;
    int bad = 0;
    for (int p = 0; p < c->size; ++p) {
        if (send_bytes[p] > 0) bad |= r.Send((const char *)send + so, (size_t)send_bytes[p], 1, p, c->comm, (hipStream_t)stream);
        if (recv_bytes[p] > 0) bad |= r.Recv((char *)recv + ro, (size_t)recv_bytes[p], 1, p, c->comm, (hipStream_t)stream);
        so += (size_t)send_bytes[p];
        ro += (size_t)recv_bytes[p];
    }
    bad |= r.GroupEnd();
    return bad == 0 ? 0 : IBVH_ERR_HIP;
}

int d2h(void *dst, const void *src, size_t bytes, hipStream_t st) {
    IBVH_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
    return IBVH_OK;
}

} // namespace

#define DIST_HIP_CHECK(expr)                                  \
    do {                                                      \
        if ((expr) != hipSuccess) return IBVH_ERR_HIP;        \
    } while (0)

extern "C" {

ibvh_status ibvh_comm_from_rccl(void *nccl_comm, int32_t rank, int32_t size, ibvh_comm *out) {
    if (!nccl_comm || !out || size < 1 || rank < 0 || rank >= size) return IBVH_ERR_INVALID_ARG;
    if (!rccl().ok) return IBVH_ERR_UNSUPPORTED; // librccl.so not found
    RcclCtx *c = new RcclCtx{nccl_comm, rank, size}; // (released by ibvh_comm_release)
    out->ctx = c;
    out->rank = rank;
    out->size = size;
    out->all_reduce = rccl_all_reduce;
    out->all_gather = rccl_all_gather;
    out->all_to_all_v = rccl_all_to_all_v;
    return IBVH_OK;
}

ibvh_status ibvh_comm_release(ibvh_comm *comm) {
    if (!comm) return IBVH_ERR_INVALID_ARG;
    if (comm->all_reduce == rccl_all_reduce && comm->ctx) delete (RcclCtx *)comm->ctx; // (a caller's own vtable owns its own context)
    std::memset(comm, 0, sizeof(*comm));
    return IBVH_OK;
}

// ---- splitter search (host only; identical arithmetic on every rank) ------------------------------------------
// Keys k_1 <= ... <= k_{P-1}: rank r receives the keys in [k_r, k_{r+1}).  Digit histograms are refined from the top of
// the key, 12 bits per level.  After a level, splitter s sits on the digit whose cumulative count first exceeds the
// balanced target s*N/P; it may stop there (k_s = prefix << remaining bits) once the bucket it landed in holds at most
// tolerance*N/P keys — the imbalance it can cause.  tolerance = 0 refines to full key resolution.
ibvh_status ibvh_splitter_search_init(ibvh_splitter_search *s, int32_t size, int32_t key_bits, int64_t n_global, double tolerance) {
    if (!s || size < 1 || size > MAX_RANKS || key_bits < 1 || key_bits > 64 || n_global < 0 || !(tolerance >= 0.0)) return IBVH_ERR_INVALID_ARG;
    std::memset(s, 0, sizeof(*s));
    s->size = size;
    s->key_bits = key_bits;
    s->n_global = n_global;
    s->tolerance = tolerance;
    s->all_done = size == 1 ? 1 : 0;
    s->next_bits = key_bits < DIGIT_BITS ? key_bits : DIGIT_BITS;
    s->next_shift = key_bits - s->next_bits;
    s->num_rows = 0; // level 0: one row over all keys
    return IBVH_OK;
}

// hist: max(num_rows, 1) rows of 2^next_bits GLOBAL counts (row j: the keys whose decided prefix is rows[j]; level 0: all keys)
ibvh_status ibvh_splitter_search_step(ibvh_splitter_search *s, const int64_t *hist) {
    if (!s || !hist) return IBVH_ERR_INVALID_ARG;
    if (s->all_done) return IBVH_OK;
    const int P = s->size, bits = s->next_bits, shift = s->next_shift, width = 1 << bits;
    const double allowed = s->tolerance * (double)s->n_global / (double)P;
    for (int k = 0; k < P - 1; ++k) {
        if (s->done[k]) continue;
        const int64_t *row = hist + (size_t)(s->decided == 0 ? 0 : s->row_of[k]) * width;
        const int64_t target = (int64_t)(k + 1) * s->n_global / P;
        const int64_t rem = target - s->below[k];
        // first digit whose inclusive cumulative count exceeds rem
        int64_t cum = 0;
        int d = 0;
        for (; d < width; ++d) {
            if (cum + row[d] > rem) break;
            cum += row[d];
        }
        if (d >= width) {
            d = width - 1;
            cum -= row[d];
        }
        s->below[k] += cum;
        s->prefix[k] = (s->prefix[k] << bits) | (uint64_t)d;
        if (shift == 0 || (double)row[d] <= allowed) {
            s->done[k] = 1;
            s->splitters[k] = shift >= 64 ? 0 : s->prefix[k] << shift;
        }
    }
    s->decided += bits;
    bool all = true;
    for (int k = 0; k < P - 1; ++k) all = all && s->done[k];
    s->all_done = (all || s->decided >= s->key_bits) ? 1 : 0;
    if (!s->all_done) { // the next level: one histogram row per distinct undecided prefix
        s->next_bits = s->key_bits - s->decided < DIGIT_BITS ? s->key_bits - s->decided : DIGIT_BITS;
        s->next_shift = s->key_bits - s->decided - s->next_bits;
        int nr = 0;
        for (int k = 0; k < P - 1; ++k) {
            if (s->done[k]) continue;
            int j = 0;
            for (; j < nr; ++j)
                if (s->rows[j] == s->prefix[k]) break;
            if (j == nr) { // insert sorted (the prefixes of ascending splitters ascend)
                int at = nr;
                while (at > 0 && s->rows[at - 1] > s->prefix[k]) {
                    s->rows[at] = s->rows[at - 1];
                    --at;
                }
                s->rows[at] = s->prefix[k];
                ++nr;
            }
        }
        s->num_rows = nr;
        for (int k = 0; k < P - 1; ++k) {
            s->row_of[k] = 0;
            if (s->done[k]) continue;
            for (int j = 0; j < nr; ++j)
                if (s->rows[j] == s->prefix[k]) s->row_of[k] = j;
        }
    }
    return IBVH_OK;
}

// ---- the driver --------------------------------------------------------------------------------------------------
ibvh_status ibvh_dist_scratch_bytes(const ibvh_types *types, int64_t n_local, int32_t size, size_t *bytes_out) {
    if (!types || !bytes_out || n_local < 0 || size < 1 || size > MAX_RANKS) return IBVH_ERR_INVALID_ARG;
    ibvh_layout lay;
    if (!layout_of(*types, lay)) return IBVH_ERR_UNSUPPORTED;
    *bytes_out = make_layout(*types, n_local, size).total;
    return IBVH_OK;
}

ibvh_status ibvh_dist_plan(const ibvh_types *types, const ibvh_comm *comm, const void *volumes, int64_t n_local, double tolerance,
                           void *scratch, size_t scratch_bytes, ibvh_dist_plan_t *plan, void *stream) {
    // what cannot be agreed on: no communicator, no plan to fill
    if (!comm || !plan) return IBVH_ERR_INVALID_ARG;
    const int P = comm->size, me = comm->rank;
    if (P < 1 || P > MAX_RANKS || me < 0 || me >= P) return IBVH_ERR_INVALID_ARG;
    if (P > 1 && (!comm->all_reduce || !comm->all_gather || !comm->all_to_all_v)) return IBVH_ERR_INVALID_ARG;
    // everything else is this rank's STATUS (round 6): a rank whose arguments are not acceptable still takes part in the first
    // all-reduce — which carries the largest status of all ranks in an element of its own — and every rank returns before the next
    // collective: its own error, or IBVH_ERR_PEER.  (It needs room for its parts of the two collectives that run before the host
    // looks at anything — the all-reduce's vector and the all-gather of the first histogram, ~150 KB at 8 ranks; a rank without
    // that returns at once and its peers wait: abort the communicator.)
    const size_t width0_bytes = (size_t)4 << DIGIT_BITS;
    const size_t room_to_take_part = (size_t)8 * (7 + (size_t)P) + ((size_t)P + 1) * width0_bytes + 1024;
    if (!scratch || scratch_bytes < room_to_take_part) return IBVH_ERR_SCRATCH;
    hipStream_t st = (hipStream_t)stream;
    int status = IBVH_OK;
    ibvh_layout lay{};
    if (!types || n_local < 0 || (n_local > 0 && !volumes) || !(tolerance >= 0.0)) status = IBVH_ERR_INVALID_ARG;
    else if (!layout_of(*types, lay) || !combo_ok(*types)) status = IBVH_ERR_UNSUPPORTED;
    else if (scratch_bytes < make_layout(*types, n_local, P).total) status = IBVH_ERR_SCRATCH;
    if (status != IBVH_OK) {
        if (P > 1) { // neutral values + the status: [0, 6 + P) as ibvh_dist_pack_extrema lays them out, [6 + P] the status
            std::vector<double> v(7 + (size_t)P, 0.0);
            for (int k = 0; k < 6; ++k) v[k] = -std::numeric_limits<double>::infinity();
            v[6 + (size_t)P] = (double)status;
            char *tmp = (char *)align_up((int64_t)(uintptr_t)scratch, 256);
            DIST_HIP_CHECK(hipMemcpyAsync(tmp, v.data(), v.size() * 8, hipMemcpyHostToDevice, st));
            DIST_HIP_CHECK(hipStreamSynchronize(st));
            if (int e = comm->all_reduce(comm->ctx, tmp, 7 + P, IBVH_COMM_F64, IBVH_COMM_MAX, stream)) return (ibvh_status)e;
            // (the peers gather the first histogram before their host looks at the reduced vector: an empty one from here)
            char *hist = tmp + align_up((int64_t)8 * (7 + P), 256);
            DIST_HIP_CHECK(hipMemsetAsync(hist, 0, width0_bytes, st));
            if (int e = comm->all_gather(comm->ctx, hist, hist + width0_bytes, (int64_t)width0_bytes, stream)) return (ibvh_status)e;
            DIST_HIP_CHECK(hipStreamSynchronize(st));
        }
        return (ibvh_status)status;
    }
    const Layout L = make_layout(*types, n_local, P);
    char *base = (char *)scratch;
    const int flt = types->leaf_float, fb = float_bytes(flt);
    const int kb = key_bytes_of(*types), key_bits = key_bits_of(*types);
    std::memset(plan, 0, sizeof(*plan));
    plan->size = P;
    plan->n_local = n_local;
    plan->record_bytes = lay.leaf_bytes;
#define IBVH_TRY(expr)                        \
    do {                                      \
        const int e_ = (int)(expr);           \
        if (e_ != IBVH_OK) return (ibvh_status)e_; \
    } while (0)
    // 1. extrema -> [-mins, maxs, one-hot counts] -> ONE all-reduce(MAX) -> global, expanded extrema
    if (n_local > 0)
        IBVH_TRY(ibvh_extrema(types, volumes, 0, n_local, 0, base + L.ext_local, base + L.ext_scratch, (size_t)1 << 17, stream));
    IBVH_TRY(ibvh_dist_pack_extrema(flt, base + L.ext_local, n_local > 0 ? 1 : 0, me, P, n_local, base + L.vec, stream));
    if (P > 1) {
        DIST_HIP_CHECK(hipMemsetAsync(base + L.vec + (size_t)8 * (6 + P), 0, 8, st)); // element 6 + P: the largest status of all ranks (0.0: fine)
        IBVH_TRY(comm->all_reduce(comm->ctx, base + L.vec, 7 + P, IBVH_COMM_F64, IBVH_COMM_MAX, stream));
    }
    IBVH_TRY(ibvh_dist_unpack_extrema(flt, base + L.vec, base + L.ext, stream));
    // 2. keys, first digit histogram (all-gathered: every rank then knows the whole send matrix when one level suffices)
    const int bits0 = key_bits < DIGIT_BITS ? key_bits : DIGIT_BITS, shift0 = key_bits - bits0, width0 = 1 << bits0;
    if (n_local > 0) {
        IBVH_TRY(ibvh_morton_keys(types, volumes, 0, n_local, base + L.ext, base + L.keys, stream));
        IBVH_TRY(ibvh_key_histogram(kb, base + L.keys, n_local, shift0, bits0, 64, nullptr, 0, base + L.hist, stream));
    } else {
        DIST_HIP_CHECK(hipMemsetAsync(base + L.hist, 0, (size_t)width0 * 4, st));
    }
    if (P > 1) IBVH_TRY(comm->all_gather(comm->ctx, base + L.hist, base + L.allh, (int64_t)width0 * 4, stream));
    else DIST_HIP_CHECK(hipMemcpyAsync(base + L.allh, base + L.hist, (size_t)width0 * 4, hipMemcpyDeviceToDevice, st));
    // ONE device -> host round trip: the reduced vector (leaf counts), the extrema, every rank's histogram
    std::vector<double> vec(7 + P, 0.0);
    unsigned char ext_raw[48];
    std::vector<uint32_t> H((size_t)P * width0);
    IBVH_TRY(d2h(vec.data(), base + L.vec, vec.size() * 8, st));
    IBVH_TRY(d2h(ext_raw, base + L.ext, (size_t)6 * fb, st));
    IBVH_TRY(d2h(H.data(), base + L.allh, H.size() * 4, st));
    DIST_HIP_CHECK(hipStreamSynchronize(st));
    if (P > 1 && vec[6 + (size_t)P] > 0.0) return IBVH_ERR_PEER; // (every healthy rank reads the same element: all of them stop here)
    int64_t n_global = 0, before = 0;
    for (int r = 0; r < P; ++r) {
        const int64_t c = (int64_t)std::llround(vec[6 + r]);
        if (r < me) before += c;
        n_global += c;
    }
    plan->n_global = n_global;
    plan->base = before;
    for (int k = 0; k < 6; ++k) plan->extrema[k] = fb == 8 ? ((const double *)ext_raw)[k] : (double)((const float *)ext_raw)[k];
    if (n_global < P) return IBVH_ERR_DOMAIN; // fewer leaves than ranks
    // 3. splitters
    ibvh_splitter_search ss;
    IBVH_TRY(ibvh_splitter_search_init(&ss, P, key_bits, n_global, tolerance));
    if (P > 1) {
        std::vector<int64_t> sum((size_t)width0, 0);
        for (int r = 0; r < P; ++r)
            for (int d = 0; d < width0; ++d) sum[d] += H[(size_t)r * width0 + d];
        IBVH_TRY(ibvh_splitter_search_step(&ss, sum.data()));
        std::vector<int64_t> rows_host;
        while (!ss.all_done) { // (rare: a splitter landed in a bucket heavier than the tolerance allows)
            const int nr = ss.num_rows, width = 1 << ss.next_bits;
            rows_host.resize((size_t)nr * width);
            // (the histogram tables hold 16 rows: more undecided prefixes than that — many ranks under tolerance 0, or a tightly
            // clustered cloud — go through them 15 rows at a time, one all-reduce per batch)
            for (int r0 = 0; r0 < nr; r0 += 15) {
                const int nb = nr - r0 < 15 ? nr - r0 : 15;
                if (n_local > 0)
                    IBVH_TRY(ibvh_key_histogram(kb, base + L.keys, n_local, ss.next_shift, ss.next_bits, ss.next_shift + ss.next_bits, ss.rows + r0, nb,
                                                base + L.hist, stream));
                else DIST_HIP_CHECK(hipMemsetAsync(base + L.hist, 0, (size_t)nb * width * 4, st));
                const int64_t cnt = (int64_t)nb * width;
                widen_u32_i64_kernel<<<dim3((unsigned)ceil_div(cnt, 256)), dim3(256), 0, st>>>((const uint32_t *)(base + L.hist), (long long *)(base + L.hist64), cnt);
                IBVH_TRY(comm->all_reduce(comm->ctx, base + L.hist64, cnt, IBVH_COMM_I64, IBVH_COMM_SUM, stream));
                IBVH_TRY(d2h(rows_host.data() + (size_t)r0 * width, base + L.hist64, (size_t)cnt * 8, st));
                DIST_HIP_CHECK(hipStreamSynchronize(st));
            }
            IBVH_TRY(ibvh_splitter_search_step(&ss, rows_host.data()));
        }
    }
    plan->levels_used = ss.decided;
    for (int k = 0; k < P - 1; ++k) plan->splitters[k] = ss.splitters[k];
    // 4. who sends what to whom
    bool matrix_known = P > 1 && ss.decided <= bits0; // splitters sit on first-level bucket boundaries
    if (P == 1) {
        plan->send_counts[0] = plan->recv_counts[0] = n_local;
    } else if (matrix_known) {
        std::vector<int64_t> edge(P + 1);
        edge[0] = 0;
        edge[P] = width0;
        for (int r = 1; r < P; ++r) edge[r] = (int64_t)(plan->splitters[r - 1] >> shift0);
        for (int dst = 0; dst < P; ++dst) {
            int64_t mine = 0, col = 0;
            for (int src = 0; src < P; ++src) {
                int64_t c = 0;
                for (int64_t d = edge[dst]; d < edge[dst + 1]; ++d) c += H[(size_t)src * width0 + d];
                if (src == me) mine = c;
                if (dst == me) plan->recv_counts[src] = c;
                col += c;
            }
            plan->send_counts[dst] = mine;
            if (col < 1) return IBVH_ERR_DOMAIN; // a rank would receive nothing: EVERY rank stops here, together
        }
    }
    // stable partition of the local leaves by destination (always inside the library); unknown matrix: it counts as well
    if (P > 1 && n_local > 0) {
        IBVH_TRY(ibvh_dist_partition(kb, base + L.keys, n_local, plan->splitters, P, base + L.perm, matrix_known ? nullptr : base + L.cnt,
                                     base + L.part, L.part_bytes, stream));
    } else if (P > 1 && !matrix_known) {
        DIST_HIP_CHECK(hipMemsetAsync(base + L.cnt, 0, (size_t)P * 8, st));
    }
    if (P > 1 && !matrix_known) {
        // exchange the counts (8 bytes per peer), then make sure no rank is left empty
        std::vector<int64_t> eight(P, 8);
        IBVH_TRY(comm->all_to_all_v(comm->ctx, base + L.cnt, eight.data(), base + L.cnt + (size_t)P * 8, eight.data(), stream));
        IBVH_TRY(d2h(plan->send_counts, base + L.cnt, (size_t)P * 8, st));
        IBVH_TRY(d2h(plan->recv_counts, base + L.cnt + (size_t)P * 8, (size_t)P * 8, st));
        DIST_HIP_CHECK(hipStreamSynchronize(st));
        int64_t got = 0;
        for (int r = 0; r < P; ++r) got += plan->recv_counts[r];
        DIST_HIP_CHECK(hipMemcpyAsync(base + L.cnt + (size_t)2 * P * 8, &got, 8, hipMemcpyHostToDevice, st));
        IBVH_TRY(comm->all_reduce(comm->ctx, base + L.cnt + (size_t)2 * P * 8, 1, IBVH_COMM_I64, IBVH_COMM_MIN, stream));
        int64_t min_recv = 0;
        IBVH_TRY(d2h(&min_recv, base + L.cnt + (size_t)2 * P * 8, 8, st));
        DIST_HIP_CHECK(hipStreamSynchronize(st));
        if (min_recv < 1) return IBVH_ERR_DOMAIN;
    }
    int64_t n_slice = 0;
    for (int r = 0; r < P; ++r) n_slice += plan->recv_counts[r];
    plan->n_slice = n_slice;
    if (n_slice < 1) return IBVH_ERR_DOMAIN;
    return IBVH_OK;
}

ibvh_status ibvh_dist_exchange(const ibvh_types *types, const ibvh_comm *comm, const void *volumes, const ibvh_dist_plan_t *plan,
                               void *scratch, size_t scratch_bytes, void *records_out, void *stream) {
    if (!types || !comm || !plan || !records_out) return IBVH_ERR_INVALID_ARG;
    const int P = comm->size;
    if (P != plan->size || P < 1 || P > MAX_RANKS) return IBVH_ERR_INVALID_ARG;
    ibvh_layout lay;
    if (!layout_of(*types, lay)) return IBVH_ERR_UNSUPPORTED;
    const int64_t n_local = plan->n_local;
    if (n_local > 0 && !volumes) return IBVH_ERR_INVALID_ARG;
    const Layout L = make_layout(*types, n_local, P);
    if (!scratch || scratch_bytes < L.total) return IBVH_ERR_SCRATCH;
    hipStream_t st = (hipStream_t)stream;
    char *base = (char *)scratch;
    const int64_t rb = lay.leaf_bytes;
    if (P == 1) { // records straight into the caller's array
        if (n_local > 0) IBVH_TRY(ibvh_pack_records(types, volumes, base + L.keys, nullptr, plan->base, n_local, records_out, stream));
        return IBVH_OK;
    }
    if (n_local > 0) IBVH_TRY(ibvh_pack_records(types, volumes, base + L.keys, base + L.perm, plan->base, n_local, base + L.records, stream));
    int64_t sb[MAX_RANKS], rbts[MAX_RANKS];
    for (int r = 0; r < P; ++r) {
        sb[r] = plan->send_counts[r] * rb;
        rbts[r] = plan->recv_counts[r] * rb;
    }
    IBVH_TRY(comm->all_to_all_v(comm->ctx, base + L.records, sb, records_out, rbts, stream));
    (void)st;
    return IBVH_OK;
#undef IBVH_TRY
}

} // extern "C"

// ---- cross-shard contact completion behind the boundary ---------------------------------------------------
// Per-slice trees do not see contacts between leaves of different slices.  Every rank publishes its slice's root box and leaf
// count (ONE all-gather of 64 bytes a rank); for every pair of slices r < s whose boxes touch, rank s sends rank r the leaves
// that can matter there — those whose own box touches r's root box (a thin shell of the slice: Morton slices of a cloud meet at
// faces), NOT its tree: round 4 shipped whole trees, 600 MB a peer at config 5 for 0.5 % of the contacts — and rank r builds an
// ordinary BVH over what it received (ibvh_build: extrema, Morton sort, merge) and runs the ordinary pair traversal
// (ibvh_traverse_pair_lvt_*) of its own tree against it.  Per-slice self contacts and these pairs together are the contact set
// of the whole cloud, every pair exactly once (tests/test_gpu_dist_procs.py, test_gpu_parity.py, test_gpu_dist.py).
//   _plan     the slice's description (<= 16 boxes, refined greedily ON THE DEVICE: describe_kernel) + leaf count + this rank's
//             status, all-gathered (ONE all-gather of ~800 bytes a rank); per lower rank a counting pass over the own leaves, which
//             reads the receiver's boxes from the gathered records and leaves at once when the two slices do not touch; the counts
//             exchanged (8 bytes a peer, one all_to_all_v); ONE host synchronisation for records and counts together; fills the
//             plan: who sends how many leaves to whom and how large the caller's three buffers are (export, import, traversal
//             scratch).  A rank whose arguments are not acceptable still takes part in both collectives (its status travels in
//             its record) and EVERY rank returns an error together: IBVH_ERR_PEER on the ranks that were fine (round 6).
//   _exchange selected leaves -> export buffer, contiguous by receiver (one compaction pass per receiver), then ONE all_to_all_v
//             (round 6; P - 1 sequential rounds before): the import buffer holds the received leaf sets contiguously by sender
//             (what all_to_all_v delivers), the room for their trees' nodes and skips behind them
//   _count    per imported set: ibvh_build in place, pair-traversal counting pass;  _write: the writing passes.
// The order in which a sender's selected leaves arrive is NOT deterministic (one atomic cursor per workgroup): the receiver's
// stable Morton sort then orders leaves of EQUAL codes differently from run to run, and with them the order of their cross
// contacts — cross contacts are a SET (include/ibvh.h); the per-slice lists keep the reference's order.
namespace {
// the box a leaf is tested with, in double
template <class T> IBVH_D void leaf_box(const BSphere<T> &s, double (&lo)[3], double (&up)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = (double)(s.x[k] - s.r);
        up[k] = (double)(s.x[k] + s.r);
    }
}
template <class T> IBVH_D void leaf_box(const BBox<T> &b, double (&lo)[3], double (&up)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = (double)b.lo[k];
        up[k] = (double)b.up[k];
    }
}
constexpr int CROSS_BOXES = IBVH_DIST_CROSS_BOXES; // boxes a slice is described by: nodes of its tree, refined greedily (Morton slices are not convex)
// one rank's record in the all-gather: the boxes its slice is described by, its leaf count, n_boxes (or -status when the rank's
// arguments were not acceptable: the others then stop with IBVH_ERR_PEER instead of waiting in a later collective)
struct CrossRec {
    double box[CROSS_BOXES][6];
    int64_t leaves, n_boxes;
};
static_assert(sizeof(CrossRec) == IBVH_DIST_CROSS_BOXES * 48 + 16, "record layout (IBVH_DIST_CROSS_SCRATCH)");

template <class V> IBVH_D void box_of(const V &v, double out[6]) { // the volume's box in double: never smaller than the box any node type would hold
    double lo[3], up[3];
    leaf_box(v, lo, up);
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = lo[k], out[3 + k] = up[k];
}
// This rank's record.  A Morton slice is not convex — where it straddles a big jump of the Z-curve one node box spans the scene
// and every leaf of every other slice "touches" it — so the description is refined greedily: start with the root, replace the box
// of the largest volume by its two children, until CROSS_BOXES boxes (the straddling node is split again and again, one tight
// child peeled off each time, down to the level where the jump sits).  One thread: ~15 dependent 48-byte reads (round 5 made
// them from the host, one stream synchronisation each).
template <class NV, class LV>
__global__ void describe_kernel(const char *nodes, const char *leaves, int64_t node_bytes, TreeDev tree, bool from_leaf, int64_t n_leaves, int32_t status,
                                CrossRec *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    CrossRec r;
    for (int i = 0; i < CROSS_BOXES; ++i)
        for (int k = 0; k < 6; ++k) r.box[i][k] = 0.0;
    r.leaves = n_leaves;
    r.n_boxes = -(int64_t)status;
    if (status == 0) {
        if (from_leaf) box_of(load_vol<LV>(leaves), r.box[0]);
        else box_of(load_vol<NV>(nodes), r.box[0]);
        int64_t node[CROSS_BOXES], level[CROSS_BOXES]; // heap index and level of every box
        node[0] = 1, level[0] = 1;
        int nb = 1;
        const int64_t levels = tree.levels, vl = tree.virtual_leaves;
        while (!from_leaf && nb < CROSS_BOXES) {
            int pick = -1;
            double best = 0.0;
            for (int i = 0; i < nb; ++i) {
                if (level[i] + 1 > levels - 1) continue; // (its children are leaves: not node boxes)
                double v = 1.0;
                for (int k = 0; k < 3; ++k) {
                    const double d = r.box[i][3 + k] - r.box[i][k];
                    v *= d > 0 ? d : 0.0;
                }
                v = v == v ? v : 0.0; // (a NaN box touches nothing: nothing to refine)
                if (pick < 0 || v > best) pick = i, best = v;
            }
            if (pick < 0 || best <= 0.0) break;
            const int64_t cl = level[pick] + 1, c0 = 2 * node[pick], c1 = c0 + 1;
            const bool real1 = (c1 - ((int64_t)1 << (cl - 1))) < level_num_real(levels, vl, cl);
            const int64_t mem0 = level_start(levels, vl, cl) - 1 + (c0 - ((int64_t)1 << (cl - 1))); // 0-based memory index of the left child
            box_of(load_vol<NV>(nodes + mem0 * node_bytes), r.box[pick]);
            node[pick] = c0, level[pick] = cl;
            if (real1) {
                box_of(load_vol<NV>(nodes + (mem0 + 1) * node_bytes), r.box[nb]);
                node[nb] = c1, level[nb] = cl;
                ++nb;
            }
        }
        r.n_boxes = nb;
    }
    *out = r;
}
IBVH_HD bool recs_touch(const CrossRec &a, const CrossRec &b) { // any box of a against any box of b (iscontact.jl:20-28; a NaN box touches nothing)
    if (a.n_boxes <= 0 || b.n_boxes <= 0) return false;
    for (int64_t i = 0; i < a.n_boxes; ++i)
        for (int64_t j = 0; j < b.n_boxes; ++j) {
            bool t = true;
            for (int k = 0; k < 3; ++k) t = t && a.box[i][3 + k] >= b.box[j][k] && a.box[i][k] <= b.box[j][3 + k];
            if (t) return true;
        }
    return false;
}
// leaves of a slice whose box touches one of the RECEIVER's boxes (widened by a few ulps of the narrower float type: the filter must
// keep every leaf the pair traversal could report, and its own box tests round differently): counted (out == nullptr) or copied,
// unordered (the receiver sorts them again when it builds its tree), behind one atomic per workgroup.  recs: the gathered
// records on the device; nothing to do when the receiver's slice (rank `to`, lower) and this one (rank `me`) do not touch.
template <class V> __global__ __launch_bounds__(256) void cross_filter_kernel(const char *leaves, LeafLayout lay, int64_t n, const CrossRec *recs, int to,
                                                                              int me, double eps, unsigned long long *cursor, char *out) {
    __shared__ uint32_t s_cnt, s_base;
    __shared__ double s_lo[CROSS_BOXES][3], s_up[CROSS_BOXES][3];
    // do the two slices touch at all?  One (box of `to`, box of `me`) pair per thread: 16 x 16 = 256
    const int nb = (int)recs[to].n_boxes, nm = (int)recs[me].n_boxes;
    bool t = false;
    {
        const int i = threadIdx.x / CROSS_BOXES, j = threadIdx.x % CROSS_BOXES;
        if (i < nb && j < nm) {
            t = true;
            for (int k = 0; k < 3; ++k) t = t && recs[to].box[i][3 + k] >= recs[me].box[j][k] && recs[to].box[i][k] <= recs[me].box[j][3 + k];
        }
    }
    if (!__syncthreads_or(t ? 1 : 0)) return;
    if ((int)threadIdx.x < 3 * CROSS_BOXES) {
        const int b = threadIdx.x / 3, k = threadIdx.x % 3;
        if (b < nb) {
            const double lo = recs[to].box[b][k], up = recs[to].box[b][3 + k];
            const double ext = fabs(up - lo) + fabs(lo) + fabs(up);
            s_lo[b][k] = lo - eps * ext - 1e-300;
            s_up[b][k] = up + eps * ext + 1e-300;
        }
    }
    for (int64_t c0 = (int64_t)blockIdx.x * 256; c0 < n; c0 += (int64_t)gridDim.x * 256) { // (a bounded grid: an idle launch costs microseconds)
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const int64_t i = c0 + threadIdx.x;
        bool keep = false;
        if (i < n) {
            const V v = load_vol<V>(leaves + i * lay.stride);
            double lo[3], up[3];
            leaf_box(v, lo, up);
            for (int b = 0; b < nb; ++b) {
                bool in = true;
#pragma unroll
                for (int k = 0; k < 3; ++k) in = in && lo[k] <= s_up[b][k] && up[k] >= s_lo[b][k];
                keep = keep || in;
            }
        }
        uint32_t at = 0;
        if (keep) at = atomicAdd(&s_cnt, 1u);
        __syncthreads();
        if (threadIdx.x == 0 && s_cnt != 0) s_base = (uint32_t)atomicAdd(cursor, (unsigned long long)s_cnt);
        __syncthreads();
        if (keep && out != nullptr) {
            const uint64_t *src = (const uint64_t *)(leaves + i * lay.stride);
            uint64_t *dst = (uint64_t *)(out + ((int64_t)s_base + at) * lay.stride);
            for (int w = 0; w < lay.stride / 8; ++w) dst[w] = src[w];
        }
    }
}
int launch_cross_filter(const ibvh_types &t, const void *leaves, int64_t n, const CrossRec *recs, int to, int me, unsigned long long *cursor, void *out,
                        hipStream_t st) {
    ibvh_layout lay;
    LeafLayout dl;
    if (!layout_of(t, lay, &dl)) return IBVH_ERR_UNSUPPORTED;
    const double eps = (t.leaf_float == IBVH_F32 || t.node_float == IBVH_F32) ? 1e-5 : 1e-13;
    if (n <= 0) return IBVH_OK;
    return dispatch_volume(t.leaf_kind, t.leaf_float, [&](auto vt) -> int {
        using V = typename decltype(vt)::type;
        const int64_t wgs = ceil_div(n, 256);
        hipLaunchKernelGGL((cross_filter_kernel<V>), dim3((unsigned)(wgs < 8192 ? wgs : 8192)), dim3(256), 0, st, (const char *)leaves, dl, n, recs, to, me, eps, cursor,
                           (char *)out);
        return hipGetLastError() == hipSuccess ? (int)IBVH_OK : (int)IBVH_ERR_HIP;
    });
}
struct CrossSizes {
    int64_t leaf_bytes, node_bytes, skip_bytes, tree_bytes; // one imported set: leaves | nodes | skips, each padded to 256 bytes
    size_t counts_bytes, lvt_bytes, build_bytes;             // its share of the traversal scratch; the build's scratch (shared)
};
bool cross_sizes(const ibvh_types &t, int64_t n_other, int64_t n_mine, int32_t cache_slots, CrossSizes &z) {
    ibvh_layout lay;
    if (!layout_of(t, lay)) return false;
    ibvh_tree tree;
    if (ibvh_tree_shape(n_other, &tree) != IBVH_OK) return false;
    z.leaf_bytes = n_other * lay.leaf_bytes;
    z.node_bytes = (tree.real_nodes - tree.real_leaves) * lay.node_bytes;
    z.skip_bytes = tree.levels * (t.index_type == IBVH_I64 ? 8 : 4);
    z.tree_bytes = align_up(z.leaf_bytes, 256) + align_up(z.node_bytes, 256) + align_up(z.skip_bytes, 256);
    const int64_t items = n_other < n_mine ? n_other : n_mine; // IBVH_PAIR_SMALLER_DRIVES: the thin boundary shell walks the whole slice's tree
    z.counts_bytes = (size_t)align_up(items * (t.index_type == IBVH_I64 ? 8 : 4), 256);
    if (ibvh_lvt_scratch_bytes(&t, items, cache_slots, &z.lvt_bytes) != IBVH_OK) return false;
    z.lvt_bytes = (size_t)align_up((int64_t)z.lvt_bytes, 256);
    if (ibvh_build_scratch_bytes(&t, n_other, &z.build_bytes) != IBVH_OK) return false;
    return true;
}
// import buffer: [leaf set 0 | leaf set 1 | ...] contiguous, by ascending sender (what ONE all_to_all_v delivers), then, from the
// next 256-byte boundary, [nodes 0 | skips 0 | nodes 1 | skips 1 | ...] (each padded to 256 bytes): the trees ibvh_build makes in place
int64_t import_aux_offset(const ibvh_types &t, const ibvh_dist_cross_plan_t &plan, int k, int64_t *total_out) {
    ibvh_layout lay;
    layout_of(t, lay);
    int64_t leaves = 0;
    for (int i = 0; i < plan.n_recv; ++i) leaves += plan.recv_leaves[i] * lay.leaf_bytes;
    int64_t off = align_up(leaves, 256), mine = off;
    for (int i = 0; i < plan.n_recv; ++i) {
        if (i == k) mine = off;
        ibvh_tree tree;
        ibvh_tree_shape(plan.recv_leaves[i], &tree);
        off += align_up((tree.real_nodes - tree.real_leaves) * lay.node_bytes, 256) + align_up(tree.levels * (t.index_type == IBVH_I64 ? 8 : 4), 256);
    }
    if (total_out) *total_out = off;
    return mine;
}
ibvh_bvh imported_tree(const ibvh_bvh &mine, const ibvh_dist_cross_plan_t &plan, int k, const void *import_buf) {
    ibvh_bvh o{};
    o.types = mine.types;
    ibvh_tree_shape(plan.recv_leaves[k], &o.tree);
    o.built_level = 1;
    ibvh_layout lay;
    layout_of(mine.types, lay);
    o.leaves = (const char *)import_buf + plan.recv_offset[k];
    o.nodes = (const char *)import_buf + import_aux_offset(mine.types, plan, k, nullptr);
    o.skips = (const char *)o.nodes + align_up((o.tree.real_nodes - o.tree.real_leaves) * lay.node_bytes, 256);
    return o;
}
} // namespace

extern "C" {

ibvh_status ibvh_dist_cross_plan(const ibvh_comm *comm, const ibvh_bvh *bvh, int32_t cache_slots, void *scratch, size_t scratch_bytes,
                                 ibvh_dist_cross_plan_t *plan, void *stream) {
    // what cannot be agreed on: no communicator, no plan to fill, no scratch to take part in the collectives with
    if (!comm || !plan) return IBVH_ERR_INVALID_ARG;
    const int P = comm->size, me = comm->rank;
    if (P < 1 || P > MAX_RANKS || me < 0 || me >= P) return IBVH_ERR_INVALID_ARG;
    if (P > 1 && (!comm->all_gather || !comm->all_to_all_v)) return IBVH_ERR_INVALID_ARG;
    if (!scratch || scratch_bytes < IBVH_DIST_CROSS_SCRATCH(P)) return IBVH_ERR_SCRATCH;
    // everything else is this rank's STATUS: it travels in the rank's record, the rank takes part in both collectives, and all
    // ranks return together (a rank that returned here would leave its peers waiting in the all-gather)
    ibvh_layout lay{};
    int status = IBVH_OK;
    if (!bvh || cache_slots < 0) status = IBVH_ERR_INVALID_ARG;
    else if (!layout_of(bvh->types, lay)) status = IBVH_ERR_UNSUPPORTED;
    else if (bvh->built_level > 1 && bvh->tree.real_nodes > bvh->tree.real_leaves) status = IBVH_ERR_UNSUPPORTED; // (the root must exist)
    else if (!bvh->leaves || (bvh->tree.real_nodes > bvh->tree.real_leaves && !bvh->nodes)) status = IBVH_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    std::memset(plan, 0, sizeof(*plan));
    plan->size = P;
    plan->rank = me;
    plan->cache_slots = cache_slots;
    const int64_t n_mine = status == IBVH_OK ? bvh->tree.real_leaves : 0;
    char *base = (char *)scratch;
    CrossRec *recs = (CrossRec *)(base + sizeof(CrossRec)); // [P] the gathered records; base[0]: this rank's own
    const size_t gather_bytes = sizeof(CrossRec) * (size_t)(P + 1);
    unsigned long long *cnt_send = (unsigned long long *)(base + align_up((int64_t)gather_bytes, 256)), *cnt_recv = cnt_send + P;
    // 1. this rank's record, on the device
    if (status == IBVH_OK) {
        const bool from_leaf = bvh->tree.real_nodes <= bvh->tree.real_leaves;
        const TreeDev td{bvh->tree.levels, bvh->tree.real_leaves, bvh->tree.virtual_leaves};
        const int e = dispatch_volume(bvh->types.node_kind, bvh->types.node_float, [&](auto nt) -> int {
            using NV = typename decltype(nt)::type;
            return dispatch_volume(bvh->types.leaf_kind, bvh->types.leaf_float, [&](auto lt) -> int {
                using LV = typename decltype(lt)::type;
                hipLaunchKernelGGL((describe_kernel<NV, LV>), dim3(1), dim3(64), 0, st, (const char *)bvh->nodes, (const char *)bvh->leaves, (int64_t)lay.node_bytes, td,
                                   from_leaf, n_mine, (int32_t)0, (CrossRec *)base);
                return hipGetLastError() == hipSuccess ? (int)IBVH_OK : (int)IBVH_ERR_HIP;
            });
        });
        if (e != IBVH_OK) status = e;
    }
    if (status != IBVH_OK) { // (the record of a rank that cannot take part in the completion: no boxes, its status)
        CrossRec bad{};
        bad.n_boxes = -(int64_t)status;
        DIST_HIP_CHECK(hipMemcpyAsync(base, &bad, sizeof(bad), hipMemcpyHostToDevice, st));
        DIST_HIP_CHECK(hipStreamSynchronize(st)); // (`bad` is a stack object)
    }
    // 2. every rank's record
    if (P > 1) {
        if (int e = comm->all_gather(comm->ctx, base, recs, (int64_t)sizeof(CrossRec), stream)) return (ibvh_status)e;
    } else {
        DIST_HIP_CHECK(hipMemcpyAsync(recs, base, sizeof(CrossRec), hipMemcpyDeviceToDevice, st));
    }
    // 3. how many of the own leaves every LOWER rank gets: a counting pass per lower rank (it leaves at once when the slices do
    //    not touch), then the counts change hands
    DIST_HIP_CHECK(hipMemsetAsync(cnt_send, 0, (size_t)16 * P, st));
    if (P > 1) {
        if (status == IBVH_OK)
            for (int r = 0; r < me; ++r)
                if (int e = launch_cross_filter(bvh->types, bvh->leaves, n_mine, recs, r, me, cnt_send + r, nullptr, st)) status = e;
        std::vector<int64_t> eight(P, 8);
        if (int e = comm->all_to_all_v(comm->ctx, cnt_send, eight.data(), cnt_recv, eight.data(), stream)) return (ibvh_status)e;
    }
    // 4. ONE device -> host round trip: the records, the counts
    std::vector<CrossRec> all(P);
    std::vector<unsigned long long> host(2 * (size_t)P);
    DIST_HIP_CHECK(hipMemcpyAsync(all.data(), recs, sizeof(CrossRec) * (size_t)P, hipMemcpyDeviceToHost, st));
    DIST_HIP_CHECK(hipMemcpyAsync(host.data(), cnt_send, (size_t)16 * P, hipMemcpyDeviceToHost, st));
    DIST_HIP_CHECK(hipStreamSynchronize(st));
    if (status != IBVH_OK) return (ibvh_status)status;
    for (int r = 0; r < P; ++r)
        if (all[r].n_boxes < 0) return IBVH_ERR_PEER; // (every rank sees the same records: all of them stop here)
    for (int r = 0; r < P; ++r) {
        plan->slice_leaves[r] = all[r].leaves;
        plan->touches[r] = (r != me && recs_touch(all[me < r ? me : r], all[me < r ? r : me])) ? 1 : 0;
        plan->n_boxes[r] = (int32_t)all[r].n_boxes;
        for (int i = 0; i < CROSS_BOXES; ++i)
            for (int k = 0; k < 6; ++k) plan->boxes[r][i][k] = all[r].box[i][k];
    }
    if (P > 1) {
        int64_t eo = 0;
        for (int r = 0; r < P; ++r) { // the export buffer: contiguous by receiver, ascending (all_to_all_v's send layout)
            plan->send_leaves[r] = (int64_t)host[r];
            plan->send_offset[r] = eo;
            eo += (int64_t)host[r] * lay.leaf_bytes;
        }
        plan->export_bytes = eo;
        int64_t off = 0, scr = 0;
        size_t build_max = 0;
        for (int r = me + 1; r < P; ++r) { // leaf sets this rank imports, by ascending rank: contiguous (all_to_all_v's receive layout)
            const int64_t got = (int64_t)host[(size_t)P + r];
            if (got <= 0) continue;
            CrossSizes z;
            if (!cross_sizes(bvh->types, got, n_mine, cache_slots, z)) return IBVH_ERR_UNSUPPORTED;
            const int k = plan->n_recv++;
            plan->recv_rank[k] = r;
            plan->recv_leaves[k] = got;
            plan->recv_offset[k] = off;
            plan->scratch_offset[k] = scr;
            off += z.leaf_bytes;
            scr += (int64_t)(z.counts_bytes + z.lvt_bytes);
            build_max = z.build_bytes > build_max ? z.build_bytes : build_max;
        }
        int64_t total = 0;
        import_aux_offset(bvh->types, *plan, 0, &total);
        plan->import_bytes = plan->n_recv ? total : 0;
        plan->build_offset = scr;
        plan->scratch_bytes = scr + (int64_t)align_up((int64_t)build_max, 256);
    }
    return IBVH_OK;
}

ibvh_status ibvh_dist_cross_exchange(const ibvh_comm *comm, const ibvh_bvh *bvh, const ibvh_dist_cross_plan_t *plan, void *export_buf,
                                     void *import_buf, void *scratch, size_t scratch_bytes, void *stream) {
    if (!comm || !bvh || !plan) return IBVH_ERR_INVALID_ARG;
    const int P = comm->size, me = comm->rank;
    if (P != plan->size || me != plan->rank || P < 1 || P > MAX_RANKS) return IBVH_ERR_INVALID_ARG;
    if ((plan->import_bytes > 0 && !import_buf) || (plan->export_bytes > 0 && !export_buf)) return IBVH_ERR_INVALID_ARG;
    if (P == 1) return IBVH_OK;
    if (!scratch || scratch_bytes < IBVH_DIST_CROSS_SCRATCH(P)) return IBVH_ERR_SCRATCH;
    ibvh_layout lay;
    if (!layout_of(bvh->types, lay)) return IBVH_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // the records again (from the plan: the scratch need not be the one _plan was given), then the selected leaves of every
    // receiver, compacted into the export buffer (the counting pass of _plan, now copying)
    char *base = (char *)scratch;
    CrossRec *recs = (CrossRec *)(base + sizeof(CrossRec));
    {
        std::vector<CrossRec> all(P);
        for (int r = 0; r < P; ++r) {
            all[r].leaves = plan->slice_leaves[r];
            all[r].n_boxes = plan->n_boxes[r];
            for (int i = 0; i < CROSS_BOXES; ++i)
                for (int k = 0; k < 6; ++k) all[r].box[i][k] = plan->boxes[r][i][k];
        }
        DIST_HIP_CHECK(hipMemcpyAsync(recs, all.data(), sizeof(CrossRec) * (size_t)P, hipMemcpyHostToDevice, st));
        DIST_HIP_CHECK(hipStreamSynchronize(st)); // (`all` is freed on return)
    }
    unsigned long long *cursor = (unsigned long long *)(base + align_up((int64_t)(sizeof(CrossRec) * (size_t)(P + 1)), 256));
    DIST_HIP_CHECK(hipMemsetAsync(cursor, 0, (size_t)8 * P, st));
    for (int r = 0; r < me; ++r)
        if (plan->touches[r] && plan->send_leaves[r] > 0)
            if (int e = launch_cross_filter(bvh->types, bvh->leaves, bvh->tree.real_leaves, recs, r, me, cursor + r, (char *)export_buf + plan->send_offset[r], st))
                return (ibvh_status)e;
    // ONE all_to_all_v: rank s sends every lower touching rank its share; both buffers are laid out the way it wants them
    int64_t sb[MAX_RANKS], rbts[MAX_RANKS];
    for (int r = 0; r < P; ++r) sb[r] = rbts[r] = 0;
    for (int r = 0; r < me; ++r) sb[r] = plan->send_leaves[r] * lay.leaf_bytes;
    for (int k = 0; k < plan->n_recv; ++k) rbts[plan->recv_rank[k]] = plan->recv_leaves[k] * lay.leaf_bytes;
    const void *send = export_buf ? export_buf : (const void *)scratch;
    void *recv = import_buf ? import_buf : scratch;
    if (int e = comm->all_to_all_v(comm->ctx, send, sb, recv, rbts, stream)) return (ibvh_status)e;
    return IBVH_OK;
}

ibvh_status ibvh_dist_cross_count(const ibvh_bvh *bvh, const ibvh_dist_cross_plan_t *plan, void *import_buf, void *scratch,
                                  size_t scratch_bytes, int64_t *totals_out, int64_t *total_out, void *stream) {
    if (!bvh || !plan || !total_out) return IBVH_ERR_INVALID_ARG;
    *total_out = 0;
    if (plan->n_recv == 0) return IBVH_OK;
    if (!import_buf || !scratch || scratch_bytes < (size_t)plan->scratch_bytes) return IBVH_ERR_SCRATCH;
    for (int k = 0; k < plan->n_recv; ++k) {
        CrossSizes z;
        if (!cross_sizes(bvh->types, plan->recv_leaves[k], bvh->tree.real_leaves, plan->cache_slots, z)) return IBVH_ERR_UNSUPPORTED;
        const ibvh_bvh other = imported_tree(*bvh, *plan, k, import_buf);
        // an ordinary BVH over the received leaves, in place: their own extrema, Morton sort, bottom-up merge (thin shells of a
        // slice: the sort's extra levels stay on)
        ibvh_build_desc desc{};
        desc.types = bvh->types;
        desc.n = plan->recv_leaves[k];
        desc.built_level = 1;
        desc.already_wrapped = 1;
        desc.compute_extrema = 1;
        desc.sort_levels = 2;
        const size_t build_room = (size_t)plan->scratch_bytes - (size_t)plan->build_offset;
        if (ibvh_status e = ibvh_build(&desc, nullptr, (void *)other.leaves, (void *)other.nodes, (void *)other.skips, nullptr,
                                       (char *)scratch + plan->build_offset, build_room, stream)) return e;
        char *counts = (char *)scratch + plan->scratch_offset[k];
        // (own slice first: the pairs come out as (index in this slice, index in the other slice), both GLOBAL 1-based numbers.)
        // Enqueued with NO contact buffer: counting pass + scan, the total stays in the first 8 bytes of this set's traversal
        // scratch — the host reads all sets' totals in ONE round trip below instead of synchronising per set (round 6)
        const ibvh_status e = ibvh_traverse_pair_lvt_enqueue(bvh, &other, bvh->built_level > 1 ? bvh->built_level : 1, 1, IBVH_NARROW_NONE | IBVH_PAIR_SMALLER_DRIVES, counts,
                                                             nullptr, 0, nullptr, nullptr, counts + z.counts_bytes, z.lvt_bytes, stream);
        if (e != IBVH_OK) return e;
    }
    int64_t totals[MAX_RANKS];
    for (int k = 0; k < plan->n_recv; ++k) {
        CrossSizes z;
        cross_sizes(bvh->types, plan->recv_leaves[k], bvh->tree.real_leaves, plan->cache_slots, z);
        DIST_HIP_CHECK(hipMemcpyAsync(&totals[k], (char *)scratch + plan->scratch_offset[k] + z.counts_bytes, 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
    }
    DIST_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    const bool i32 = bvh->types.index_type == IBVH_I32;
    for (int k = 0; k < plan->n_recv; ++k) {
        if (i32 && totals[k] > (int64_t)INT32_MAX) return IBVH_ERR_OVERFLOW;
        if (totals_out) totals_out[k] = totals[k];
        *total_out += totals[k];
    }
    return IBVH_OK;
}

ibvh_status ibvh_dist_cross_write(const ibvh_bvh *bvh, const ibvh_dist_cross_plan_t *plan, const void *import_buf, void *scratch,
                                  size_t scratch_bytes, const int64_t *totals, void *contacts_out, void *stream) {
    if (!bvh || !plan) return IBVH_ERR_INVALID_ARG;
    if (plan->n_recv == 0) return IBVH_OK;
    if (!import_buf || !scratch || scratch_bytes < (size_t)plan->scratch_bytes || !totals) return IBVH_ERR_INVALID_ARG;
    // (a rank that imported boundary leaves and found no contact among them has nothing to write: NULL is fine then)
    bool any = false;
    for (int k = 0; k < plan->n_recv; ++k) any = any || totals[k] > 0;
    if (!any) return IBVH_OK;
    if (!contacts_out) return IBVH_ERR_INVALID_ARG;
    ibvh_layout lay;
    if (!layout_of(bvh->types, lay)) return IBVH_ERR_UNSUPPORTED;
    int64_t at = 0;
    for (int k = 0; k < plan->n_recv; ++k) {
        CrossSizes z;
        if (!cross_sizes(bvh->types, plan->recv_leaves[k], bvh->tree.real_leaves, plan->cache_slots, z)) return IBVH_ERR_UNSUPPORTED;
        const ibvh_bvh other = imported_tree(*bvh, *plan, k, import_buf);
        char *counts = (char *)scratch + plan->scratch_offset[k];
        if (totals[k] > 0) {
            const ibvh_status e = ibvh_traverse_pair_lvt_write(bvh, &other, bvh->built_level > 1 ? bvh->built_level : 1, 1, IBVH_NARROW_NONE | IBVH_PAIR_SMALLER_DRIVES, counts,
                                                               (char *)contacts_out + at * lay.pair_bytes, counts + z.counts_bytes, z.lvt_bytes, stream);
            if (e != IBVH_OK) return e;
        }
        at += totals[k];
    }
    return IBVH_OK;
}

} // extern "C"
