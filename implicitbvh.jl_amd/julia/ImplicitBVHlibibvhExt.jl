# ImplicitBVHlibibvhExt.jl — package extension a maintainer adds to ImplicitBVH.jl so that ROCArray inputs run
# on libibvh (hand-written HIP for MI355X / gfx950) instead of the generic KernelAbstractions path.
#
# Project.toml additions:
#   [weakdeps]   AMDGPU = "21141c5a-9bdb-4563-92ae-f87d6854732e"
#   [extensions] ImplicitBVHlibibvhExt = "AMDGPU"
# and ENV["LIBIBVH"] (or a JLL) pointing at libibvh.so.  The C ABI is include/ibvh.h.
#
# The extension only adds MORE SPECIFIC METHODS for the user entry points of the hot path:
#     BVH(::ROCVector, ...)                                            build.jl:198-271
#     traverse(bvh, ::LVTTraversal | ::BFSTraversal; ...)              lvt/traverse_single.jl, bfs/traverse_single.jl
#     traverse(bvh1, bvh2, ::LVTTraversal | ::BFSTraversal; ...)       lvt/traverse_pair.jl, bfs/traverse_pair.jl
#     traverse_rays(bvh, points, directions, ::LVTTraversal | ::BFSTraversal; ...)   raytrace/*
# Keyword surface, return types (BVH, BVHTraversal) and error behaviour are the reference's.  All buffers are
# ROCArrays owned by Julia (cache= reuse keeps working); the library never allocates.
#
# `narrow`: an arbitrary Julia closure cannot cross a C ABI.  The reference's default and two callable structs defined
# here (MortonLess, IndexLess: the predicates its tests use, runtests.jl:1239) run on the device; ANY other `narrow`
# falls back to the reference's own generic method through `invoke` — it is never silently dropped.
#
# One `ccall` per C entry point lives in the `c_*` wrappers below; tests/test_host_cpu.py parses their argument-type
# tuples and the POD structs and checks them against the ctypes binding (implicitbvh.jl_amd/lib.py, abi.py) the GPU
# parity tests run on, so the two bindings cannot drift apart.
#
# NOTE: written against include/ibvh.h; Julia is not available in the build environment of libibvh, so this file is
# delivered as source and the identical call sequences are exercised by the Python mirror (implicitbvh.jl_amd/api.py).
module ImplicitBVHlibibvhExt

using ImplicitBVH
using ImplicitBVH: BVH, BVHOptions, BVHTraversal, BoundingVolume, BSphere, BBox, ImplicitTree, IndexPair,
                   LVTTraversal, BFSTraversal, DefaultMortonAlgorithm, default_start_level, get_index_type,
                   compute_build_level
using AMDGPU
using AMDGPU: ROCArray, ROCVector, ROCMatrix

const libibvh = get(ENV, "LIBIBVH", "libibvh.so")

# The header this file was written against (include/ibvh.h, IBVH_ABI_VERSION).  A library with another struct layout or
# argument list would make the GPU write through garbage pointers, so a mismatch is refused when the extension loads.
const IBVH_ABI_VERSION = Int32(5)
function __init__()
    got = ccall((:ibvh_abi_version, libibvh), Int32, ())
    got == IBVH_ABI_VERSION ||
        error("libibvh ABI version $got, ImplicitBVHlibibvhExt was written against $IBVH_ABI_VERSION: rebuild / update one of them")
end

# ---- POD descriptors (must match include/ibvh.h) ------------------------------------------------------
struct IbvhTypes
    leaf_kind::Int32; leaf_float::Int32; node_kind::Int32; node_float::Int32; index_type::Int32; morton_type::Int32
end
struct IbvhTree
    levels::Int64; real_leaves::Int64; real_nodes::Int64; virtual_leaves::Int64; virtual_nodes::Int64
end
struct IbvhBvh
    types::IbvhTypes; tree::IbvhTree; built_level::Int64
    leaves::Ptr{Cvoid}; nodes::Ptr{Cvoid}; skips::Ptr{Cvoid}
end
struct IbvhBuildDesc
    types::IbvhTypes; n::Int64; built_level::Int64; already_wrapped::Int32; compute_extrema::Int32
    mins::NTuple{3, Float64}; maxs::NTuple{3, Float64}
    sort_levels::Int32; sort_equalize::Int32; skew_flag::Ptr{Cvoid}
end
mutable struct IbvhBfsResult
    num_contacts::Int64; num_checks::Int64; contacts_in::Int64; required_capacity::Int64
    resume_step::Int64; resume_num::Int64
end
# multi-GPU build (include/ibvh.h "multi-GPU build: the driver")
# (IbvhComm is filled by ibvh_comm_from_rccl; its last three fields are C function pointers)
struct IbvhComm
    ctx::Ptr{Cvoid}; rank::Int32; size::Int32
    all_reduce::Ptr{Cvoid}; all_gather::Ptr{Cvoid}; all_to_all_v::Ptr{Cvoid}
end
mutable struct IbvhDistPlan
    size::Int32; levels_used::Int32; n_local::Int64; n_global::Int64; base::Int64; n_slice::Int64; record_bytes::Int64
    extrema::NTuple{6, Float64}
    splitters::NTuple{256, UInt64}; send_counts::NTuple{256, Int64}; recv_counts::NTuple{256, Int64}
    IbvhDistPlan() = new()
end
mutable struct IbvhDistCrossPlan
    size::Int32; rank::Int32; n_recv::Int32; cache_slots::Int32
    import_bytes::Int64; scratch_bytes::Int64; export_bytes::Int64; build_offset::Int64
    recv_rank::NTuple{256, Int32}; recv_leaves::NTuple{256, Int64}; recv_offset::NTuple{256, Int64}; scratch_offset::NTuple{256, Int64}
    slice_leaves::NTuple{256, Int64}; touches::NTuple{256, Int32}
    send_leaves::NTuple{256, Int64}; send_offset::NTuple{256, Int64}
    n_boxes::NTuple{256, Int32}
    boxes::NTuple{24576, Float64}
    IbvhDistCrossPlan() = new()
end

# Type codes of include/ibvh.h.  Every table ends in a CATCH-ALL returning -1: a volume, float, index or Morton type the
# library has no instantiation for (BSphere{Float16}, index=UInt32, a user-defined volume, ...) is not an error here — the
# methods below are MORE SPECIFIC than the reference's, so they must hand every input they cannot serve to the reference's
# own generic method (`invoke`), never throw a MethodError for something that works without the extension
# (test/runtests.jl:480,510-538 builds Float16 volumes; utils.jl:54-71 takes any Integer exemplar).
kind(::Type{<:BSphere}) = Int32(0);  kind(::Type{<:BBox}) = Int32(1);  kind(::Type) = Int32(-1)
fltcode(::Type{Float32}) = Int32(0); fltcode(::Type{Float64}) = Int32(1); fltcode(::Type) = Int32(-1)
idxcode(::Type{Int32}) = Int32(0);   idxcode(::Type{Int64}) = Int32(1);   idxcode(::Type) = Int32(-1)
morcode(::Type{UInt16}) = Int32(0);  morcode(::Type{UInt32}) = Int32(1); morcode(::Type{UInt64}) = Int32(2); morcode(::Type) = Int32(-1)
# eltype of a volume TYPE (bsphere.jl:32, bbox.jl:41); Any for a type that does not define it
volume_float(::Type{V}) where {V} = V <: Union{BSphere, BBox} ? eltype(V) : Any

# The library's type descriptor of (leaf volume L, node N, index I, Morton M) — or `nothing` when any of the four has no
# code: the caller then takes the reference's generic path.  (A combination of KNOWN codes the library does not instantiate
# is reported by the library itself: IBVH_ERR_UNSUPPORTED from the *_scratch_bytes queries.)
function native_types(::Type{L}, ::Type{N}, ::Type{I}, ::Type{M}) where {L, N, I, M}
    codes = (kind(L), fltcode(volume_float(L)), kind(N), fltcode(volume_float(N)), idxcode(I), morcode(M))
    any(c -> c < 0, codes) && return nothing
    IbvhTypes(codes...)
end

const IBVH_ERR_UNSUPPORTED = Cint(3)
const IBVH_ERR_CAPACITY = Cint(4)

# status -> the exception the reference throws in the same situation
function check(status::Cint, what)
    status == 0 && return
    status == 1 && throw(ArgumentError("$what: invalid argument"))                 # @argcheck
    status == 2 && throw(DomainError(0, "must have at least one geometry!"))       # implicit_tree.jl:78-80
    status == 3 && throw(ArgumentError("$what: type combination not instantiated in libibvh"))
    status == 5 && throw(OverflowError("$what: count does not fit the index type"))
    status == 8 && error("$what: another rank's arguments were not acceptable (every rank returned together)")
    error("$what: libibvh status $status")
end

stream_ptr() = Ptr{Cvoid}(UInt(AMDGPU.stream().stream))   # hipStream_t of the task-local stream
devptr(a::ROCArray) = length(a) == 0 ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(UInt(pointer(a)))
devptr(::Nothing) = Ptr{Cvoid}(C_NULL)

tree_of(t::ImplicitTree) = IbvhTree(t.levels, t.real_leaves, t.real_nodes, t.virtual_leaves, t.virtual_nodes)

const RocBVH{I} = BVH{I, <:ROCVector, <:ROCVector, <:ROCVector}

# The library's view of a built BVH, or `nothing` when its types have no code (-> generic path, as native_types).
function bvh_desc(bvh::BVH{I, <:ROCVector, <:ROCVector{N}, <:ROCVector{BoundingVolume{L, I, M}}}) where {I, N, L, M}
    types = native_types(L, N, I, M)
    isnothing(types) && return nothing
    IbvhBvh(types, tree_of(bvh.tree), Int64(bvh.built_level),
            devptr(bvh.leaves), devptr(bvh.nodes), devptr(bvh.skips))
end
bvh_desc(bvh::BVH) = nothing   # (leaves that are not BoundingVolume records of the BVH's own index type)

# ---- narrow --------------------------------------------------------------------------------------------
const DEFAULT_NARROW = (bv1, bv2) -> true          # the default of every method below (=== comparable)
const DEFAULT_RAY_NARROW = (bv, p, d) -> true
"`narrow=MortonLess()`: (a, b) -> a.morton < b.morton, evaluated on the device (IBVH_NARROW_MORTON_LT)."
struct MortonLess end
(::MortonLess)(a, b) = a.morton < b.morton
"`narrow=IndexLess()`: (a, b) -> a.index < b.index, evaluated on the device (IBVH_NARROW_INDEX_LT)."
struct IndexLess end
(::IndexLess)(a, b) = a.index < b.index
# the device-side menu, or `nothing`: the caller then hands the whole call to the reference's generic method.
# The default is recognised by identity or by TYPE (a non-capturing closure is a singleton: every value of the default
# closure's type is the default).  NB an explicit `narrow=(a, b) -> true` written by the caller is a DIFFERENT closure type:
# it cannot be told apart from any other user predicate and takes the generic KernelAbstractions path of the reference —
# pass nothing (or `DEFAULT_NARROW`) to stay on the native path.
is_default_narrow(f, d) = f === d || typeof(f) === typeof(d)
narrow_code(f) = is_default_narrow(f, DEFAULT_NARROW) ? Int32(0) : f isa MortonLess ? Int32(1) : f isa IndexLess ? Int32(2) : nothing
"`narrow=OriginOutside()` for traverse_rays: (bv, p, d) -> the ray's origin lies outside bv.volume, evaluated on the device
(IBVH_NARROW_RAY_ORIGIN_OUTSIDE): drops the leaves a ray starts in."
struct OriginOutside end
(::OriginOutside)(bv::BoundingVolume{<:BSphere}, p, d) =
    (p[1] - bv.volume.x[1])^2 + (p[2] - bv.volume.x[2])^2 + (p[3] - bv.volume.x[3])^2 > bv.volume.r * bv.volume.r
(::OriginOutside)(bv::BoundingVolume{<:BBox}, p, d) = any(p .< bv.volume.lo) || any(p .> bv.volume.up)
ray_narrow_code(f) = is_default_narrow(f, DEFAULT_RAY_NARROW) ? Int32(0) : f isa OriginOutside ? Int32(3) : nothing
# Any other PURE predicate can also be served without the generic path: OR IBVH_OUTPUT_POSITIONS (0x100) into the code
# and the contact list holds leaf positions ((query, partner) / (bvh1, bvh2) / (leaf, iray), include/ibvh.h) on which the
# caller evaluates the predicate itself — `narrow` is only ever used as `iscontact(...) && narrow(...)` at leaf level.
# The methods below keep handing unknown closures to the reference's own generic method instead: a closure is not
# known to be pure.

# ---- scratch: one growing ROCVector{UInt8} per (task, purpose) instead of an allocation per call -------------
function scratch!(purpose::Symbol, nbytes::Integer)
    pool = get!(() -> Dict{Symbol, Any}(), task_local_storage(), :libibvh_scratch)::Dict{Symbol, Any}
    buf = get(pool, purpose, nothing)
    if buf === nothing || length(buf) < nbytes
        buf = ROCVector{UInt8}(undef, nbytes)
        pool[purpose] = buf
    end
    buf::ROCVector{UInt8}
end
# the device-side totals of enqueued LVT traversals: a ring of 64 words, one per call (include/ibvh.h, `total_dev`)
function next_total_word()
    ring = get!(() -> (AMDGPU.zeros(Int64, 64), Ref(0)), task_local_storage(), :libibvh_totals)
    words, calls = ring
    slot = calls[] % 64
    calls[] += 1
    Ptr{Cvoid}(UInt(pointer(words)) + 8 * slot)
end

# The host-side mirror of a total (include/ibvh.h, `total_host`): words of ONE block of mapped pinned host memory,
# allocated once per process and never freed (a kernel still in flight can only ever write into memory this module owns).
# The scan kernel stores the total there with a system-scope release; the host polls the word instead of synchronising
# the stream and copying 8 bytes back (the reference's blocking `@allowscalar`, lvt/traverse_single.jl:60).
const HOST_WORDS = 4096      # totals: words [0, HOST_WORDS)
const HINT_WORDS = 1024      # skew hints of build chains: words [HOST_WORDS, HOST_WORDS + HINT_WORDS) — a ring of their own, so a
                             # build still in flight can at worst overwrite another chain's HINT (speed only), never a total
const HOST_PENDING = typemin(Int64) >> 1
const host_block = Ref{Ptr{Int64}}(C_NULL)
const host_next = Threads.Atomic{Int}(0)
const host_lock = ReentrantLock()
function host_words()
    if host_block[] == C_NULL
        lock(host_lock) do
            if host_block[] == C_NULL
                p = Ref{Ptr{Cvoid}}(C_NULL)
                AMDGPU.HIP.hipHostMalloc(p, 8 * (HOST_WORDS + HINT_WORDS), 0) |> AMDGPU.HIP.check   # hipHostMallocDefault: mapped, coherent
                host_block[] = Ptr{Int64}(p[])
            end
        end
    end
    host_block[]
end
function next_host_word()
    w = host_words() + 8 * (Threads.atomic_add!(host_next, 1) % HOST_WORDS)
    unsafe_store!(w, HOST_PENDING)
    w
end
function poll_total(w::Ptr{Int64}, tdev, stream)
    # An ATOMIC (acquire) load: a plain unsafe_load in a side-effect-free loop may be hoisted or the loop deleted by LLVM,
    # and the poll would then never see the GPU's store.  Bounded by time, not by iteration count.
    t0 = time_ns()
    while time_ns() - t0 < 2_000_000_000
        v = unsafe_load(w, :acquire)
        v != HOST_PENDING && return v
        GC.safepoint()
    end
    total = Ref{Int64}(0)           # never arrived (should not happen): the blocking device read
    check(c_lvt_total(tdev, total, stream), "ibvh_lvt_total")
    total[]
end

# ---- one ccall per C entry point (include/ibvh.h) ---------------------------------------------------------
c_build_scratch_bytes(types, n, out) =
    ccall((:ibvh_build_scratch_bytes, libibvh), Cint,
          (Ref{IbvhTypes}, Int64, Ref{Csize_t}),
          types, n, out)
c_build(desc, volumes, leaves, nodes, skips, extrema_out, scratch, scratch_bytes, stream) =
    ccall((:ibvh_build, libibvh), Cint,
          (Ref{IbvhBuildDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
          desc, volumes, leaves, nodes, skips, extrema_out, scratch, scratch_bytes, stream)
c_comm_from_rccl(nccl_comm, rank, size, out) =
    ccall((:ibvh_comm_from_rccl, libibvh), Cint,
          (Ptr{Cvoid}, Int32, Int32, Ref{IbvhComm}), nccl_comm, rank, size, out)
c_dist_scratch_bytes(types, n_local, size, out) =
    ccall((:ibvh_dist_scratch_bytes, libibvh), Cint,
          (Ref{IbvhTypes}, Int64, Int32, Ref{Csize_t}), types, n_local, size, out)
c_dist_plan(types, comm, volumes, n_local, tolerance, scratch, sb, plan, stream) =
    ccall((:ibvh_dist_plan, libibvh), Cint,
          (Ref{IbvhTypes}, Ref{IbvhComm}, Ptr{Cvoid}, Int64, Float64, Ptr{Cvoid}, Csize_t, Ref{IbvhDistPlan}, Ptr{Cvoid}),
          types, comm, volumes, n_local, tolerance, scratch, sb, plan, stream)
c_dist_exchange(types, comm, volumes, plan, scratch, sb, records, stream) =
    ccall((:ibvh_dist_exchange, libibvh), Cint,
          (Ref{IbvhTypes}, Ref{IbvhComm}, Ptr{Cvoid}, Ref{IbvhDistPlan}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}, Ptr{Cvoid}),
          types, comm, volumes, plan, scratch, sb, records, stream)
c_dist_cross_plan(comm, bvh, cache_slots, scratch, sb, plan, stream) =
    ccall((:ibvh_dist_cross_plan, libibvh), Cint,
          (Ref{IbvhComm}, Ref{IbvhBvh}, Int32, Ptr{Cvoid}, Csize_t, Ref{IbvhDistCrossPlan}, Ptr{Cvoid}),
          comm, bvh, cache_slots, scratch, sb, plan, stream)
c_dist_cross_exchange(comm, bvh, plan, export_buf, import_buf, scratch, sb, stream) =
    ccall((:ibvh_dist_cross_exchange, libibvh), Cint,
          (Ref{IbvhComm}, Ref{IbvhBvh}, Ref{IbvhDistCrossPlan}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
          comm, bvh, plan, export_buf, import_buf, scratch, sb, stream)
c_dist_cross_count(bvh, plan, import_buf, scratch, sb, totals, total, stream) =
    ccall((:ibvh_dist_cross_count, libibvh), Cint,
          (Ref{IbvhBvh}, Ref{IbvhDistCrossPlan}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ref{Int64}, Ref{Int64}, Ptr{Cvoid}),
          bvh, plan, import_buf, scratch, sb, totals, total, stream)
c_dist_cross_write(bvh, plan, import_buf, scratch, sb, totals, contacts, stream) =
    ccall((:ibvh_dist_cross_write, libibvh), Cint,
          (Ref{IbvhBvh}, Ref{IbvhDistCrossPlan}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ref{Int64}, Ptr{Cvoid}, Ptr{Cvoid}),
          bvh, plan, import_buf, scratch, sb, totals, contacts, stream)
c_lvt_scratch_bytes(types, n_items, cache_slots, out) =
    ccall((:ibvh_lvt_scratch_bytes, libibvh), Cint,
          (Ref{IbvhTypes}, Int64, Int32, Ref{Csize_t}),
          types, n_items, cache_slots, out)
c_lvt_total(total_dev, out, stream) =
    ccall((:ibvh_lvt_total, libibvh), Cint,
          (Ptr{Cvoid}, Ref{Int64}, Ptr{Cvoid}),
          total_dev, out, stream)
c_traverse_lvt_count(bvh, sl, narrow, counts, total, scratch, sb, stream) =
    ccall((:ibvh_traverse_lvt_count, libibvh), Cint,
          (Ref{IbvhBvh}, Int64, Int32, Ptr{Cvoid}, Ref{Int64}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
          bvh, sl, narrow, counts, total, scratch, sb, stream)
c_traverse_lvt_write(bvh, sl, narrow, counts, contacts, scratch, sb, stream) =
    ccall((:ibvh_traverse_lvt_write, libibvh), Cint,
          (Ref{IbvhBvh}, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
          bvh, sl, narrow, counts, contacts, scratch, sb, stream)
c_traverse_lvt_enqueue(bvh, sl, narrow, counts, contacts, capacity, total_dev, total_host, scratch, sb, stream) =
    ccall((:ibvh_traverse_lvt_enqueue, libibvh), Cint,
          (Ref{IbvhBvh}, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
          bvh, sl, narrow, counts, contacts, capacity, total_dev, total_host, scratch, sb, stream)
c_traverse_pair_lvt_count(bvh1, bvh2, sl1, sl2, narrow, counts, total, scratch, sb, stream) =
    ccall((:ibvh_traverse_pair_lvt_count, libibvh), Cint,
          (Ref{IbvhBvh}, Ref{IbvhBvh}, Int64, Int64, Int32, Ptr{Cvoid}, Ref{Int64}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
          bvh1, bvh2, sl1, sl2, narrow, counts, total, scratch, sb, stream)
c_traverse_pair_lvt_write(bvh1, bvh2, sl1, sl2, narrow, counts, contacts, scratch, sb, stream) =
    ccall((:ibvh_traverse_pair_lvt_write, libibvh), Cint,
          (Ref{IbvhBvh}, Ref{IbvhBvh}, Int64, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
          bvh1, bvh2, sl1, sl2, narrow, counts, contacts, scratch, sb, stream)
c_traverse_pair_lvt_enqueue(bvh1, bvh2, sl1, sl2, narrow, counts, contacts, capacity, total_dev, total_host, scratch, sb, stream) =
    ccall((:ibvh_traverse_pair_lvt_enqueue, libibvh), Cint,
          (Ref{IbvhBvh}, Ref{IbvhBvh}, Int64, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
          bvh1, bvh2, sl1, sl2, narrow, counts, contacts, capacity, total_dev, total_host, scratch, sb, stream)
c_rays_scratch_bytes(bvh, num_rays, slots, out) =
    ccall((:ibvh_rays_scratch_bytes, libibvh), Cint,
          (Ref{IbvhBvh}, Int64, Int32, Ref{Csize_t}),
          bvh, num_rays, slots, out)
c_traverse_rays_lvt_count(bvh, points, dirs, num_rays, sl, narrow, counts, total, scratch, sb, stream) =
    ccall((:ibvh_traverse_rays_lvt_count, libibvh), Cint,
          (Ref{IbvhBvh}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int32, Ptr{Cvoid}, Ref{Int64}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
          bvh, points, dirs, num_rays, sl, narrow, counts, total, scratch, sb, stream)
c_traverse_rays_lvt_write(bvh, points, dirs, num_rays, sl, narrow, counts, contacts, scratch, sb, stream) =
    ccall((:ibvh_traverse_rays_lvt_write, libibvh), Cint,
          (Ref{IbvhBvh}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
          bvh, points, dirs, num_rays, sl, narrow, counts, contacts, scratch, sb, stream)
c_traverse_rays_lvt_enqueue(bvh, points, dirs, num_rays, sl, narrow, counts, contacts, capacity, total_dev, total_host, scratch, sb, stream) =
    ccall((:ibvh_traverse_rays_lvt_enqueue, libibvh), Cint,
          (Ref{IbvhBvh}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
          bvh, points, dirs, num_rays, sl, narrow, counts, contacts, capacity, total_dev, total_host, scratch, sb, stream)
c_bfs_initial_capacity(bvh, sl, out) =
    ccall((:ibvh_bfs_initial_capacity, libibvh), Cint,
          (Ref{IbvhBvh}, Int64, Ref{Int64}),
          bvh, sl, out)
c_bfs_pair_initial_capacity(bvh1, bvh2, sl1, sl2, out) =
    ccall((:ibvh_bfs_pair_initial_capacity, libibvh), Cint,
          (Ref{IbvhBvh}, Ref{IbvhBvh}, Int64, Int64, Ref{Int64}),
          bvh1, bvh2, sl1, sl2, out)
c_bfs_rays_initial_capacity(bvh, num_rays, sl, out) =
    ccall((:ibvh_bfs_rays_initial_capacity, libibvh), Cint,
          (Ref{IbvhBvh}, Int64, Int64, Ref{Int64}),
          bvh, num_rays, sl, out)
c_bfs_counters_bytes(total_levels, out) =
    ccall((:ibvh_bfs_counters_bytes, libibvh), Cint,
          (Int64, Ref{Csize_t}),
          total_levels, out)
c_traverse_bfs(bvh, sl, narrow, bvtt1, bvtt2, capacity, counters, result, stream) =
    ccall((:ibvh_traverse_bfs, libibvh), Cint,
          (Ref{IbvhBvh}, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ref{IbvhBfsResult}, Ptr{Cvoid}),
          bvh, sl, narrow, bvtt1, bvtt2, capacity, counters, result, stream)
c_traverse_pair_bfs(bvh1, bvh2, sl1, sl2, narrow, bvtt1, bvtt2, capacity, counters, result, stream) =
    ccall((:ibvh_traverse_pair_bfs, libibvh), Cint,
          (Ref{IbvhBvh}, Ref{IbvhBvh}, Int64, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ref{IbvhBfsResult}, Ptr{Cvoid}),
          bvh1, bvh2, sl1, sl2, narrow, bvtt1, bvtt2, capacity, counters, result, stream)
c_traverse_rays_bfs(bvh, points, dirs, num_rays, sl, narrow, bvtt1, bvtt2, capacity, counters, result, stream) =
    ccall((:ibvh_traverse_rays_bfs, libibvh), Cint,
          (Ref{IbvhBvh}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ref{IbvhBfsResult}, Ptr{Cvoid}),
          bvh, points, dirs, num_rays, sl, narrow, bvtt1, bvtt2, capacity, counters, result, stream)

# ---- host policies of a rebuild chain (the same rules as the Python mirror, implicitbvh.jl_amd/api.py) -----------------
# What a `cache=` rebuild asks the sort for is decided on the host from ONE word the previous build of the chain left in
# mapped pinned memory (include/ibvh.h, ibvh_build_desc.skew_flag: low byte = extra partition levels that build would have
# used, second byte = its fullest cell in 1/128 of what a finish workgroup sorts, bit 16 = it ran with equalised cells and
# the plain grid would still have been crowded, bit 17 = equalising did not help).  tests/test_host_cpu.py parses the
# constants and the two rule functions below and evaluates them against api.sort_hint_rule / api._cache_slots on a table
# of inputs: the two hosts cannot drift apart.  (Both functions are written in the small subset that test understands:
# integer arithmetic, comparisons, `ifelse`, if / elseif / else, `min` / `max` / `cld` / `nextpow`.)
const COLD_SORT_LEVELS = 2      # extra partition levels a build without cache= launches (include/ibvh.h, sort_levels)
const SPARE_OCCUPANCY = 96      # (of 128) fullest coarse cell from which a cached build launches a spare extra level
const EQ_SPARE_OCCUPANCY = 120  # the same for a chain that builds with equalised cells (they re-fit the cells every time)
const EQ_HOLDOFF = 32           # rebuilds a chain stays with the plain grid after an equalised build reported no gain (bit 17)
const EQ_STICKY = 32            # rebuilds a chain stays with equalised cells after a return to the plain grid turned out crowded
const SPARE_ALWAYS_FROM = 16777216   # leaves from which a cached build always launches the spare level
const EQUALIZE = 1              # 0: never ask for equalised cells
const MAX_SORT_LEVELS = 4       # IBVH_MAX_SORT_LEVELS

# (sort_levels, sort_equalize, holdoff') of a rebuild of n leaves — api.sort_hint_rule.  holdoff > 0: rebuilds left on the plain
# grid after bit 17; -1: probation (the previous request went back to the plain grid on an equalised build's ESTIMATE, with a
# spare level as insurance); <= -2: sticky (the probation build was crowded after all: equalised cells for EQ_STICKY rebuilds
# whatever bit 16 says)
function sort_hint_rule(word::Int64, n::Int64, holdoff::Int64)
    used = word & 0xff
    occupancy = (word >> 8) & 0xff
    eq = (word >> 16) & 1
    nohelp = (word >> 17) & 1
    ran_eq = (word >> 18) & 1
    sticky = 0
    if holdoff <= -2
        if nohelp == 1
            holdoff = EQ_HOLDOFF
        else
            sticky = 1
            holdoff = holdoff + 1
            if holdoff == -1
                holdoff = 0
            end
        end
    elseif holdoff == -1
        if used > 0
            sticky = 1
            holdoff = -2 - EQ_STICKY
        else
            holdoff = 0
        end
    elseif nohelp == 1 && holdoff == 0
        holdoff = EQ_HOLDOFF
    elseif holdoff > 0
        holdoff = holdoff - 1
    end
    equalize = ifelse((used > 0 || eq == 1 || sticky == 1) && EQUALIZE == 1 && holdoff <= 0, 1, 0)
    probation = ifelse(ran_eq == 1 && equalize == 0 && holdoff == 0 && EQUALIZE == 1, 1, 0)
    if probation == 1
        holdoff = -1
    end
    threshold = ifelse(eq == 1 && equalize == 1, EQ_SPARE_OCCUPANCY, SPARE_OCCUPANCY)
    spare = ifelse(used > 0 || occupancy >= threshold || n >= SPARE_ALWAYS_FROM || probation == 1, 1, 0)
    return (min(used + spare, MAX_SORT_LEVELS), equalize, holdoff)
end

# contact-cache slots per work item: the default, or — when a previous traversal's contact buffer is reused — about four
# times the contacts per item that buffer was sized for (a power of two, at most 64) — api._cache_slots
function cache_slots_rule(capacity::Int64, n_items::Int64, default::Int64)
    if capacity <= 0 || n_items <= 0
        return default
    end
    per_item = cld(capacity, n_items)
    want = nextpow(2, max(4 * per_item, 1))
    return max(default, min(64, want))
end
cache_slots(cache, n_items, default) =
    Int32(cache_slots_rule(isnothing(cache) ? Int64(0) : Int64(length(cache.cache1)), Int64(n_items), Int64(default)))

# The chain's state lives beside the reference's BVH struct (build.jl:155-166 has no spare field): keyed by the node array,
# which a `cache=` rebuild keeps (resize! preserves identity) and which dies with the chain.
mutable struct ChainState
    slot::Int          # index of the chain's word in the hint ring
    generation::Int    # of that slot when the chain took it: a recycled slot is another chain's
    holdoff::Int64
end
const hint_generation = zeros(Int, HINT_WORDS)
const hint_next = Ref(0)
const chains = WeakKeyDict{Any, ChainState}()
hint_ptr(st::ChainState) = host_words() + 8 * (HOST_WORDS + st.slot)
function new_chain()
    lock(host_lock) do
        slot = hint_next[]
        hint_next[] = (slot + 1) % HINT_WORDS
        hint_generation[slot + 1] += 1
        st = ChainState(slot, hint_generation[slot + 1], 0)
        unsafe_store!(hint_ptr(st), Int64(COLD_SORT_LEVELS))
        st
    end
end
# the state of the chain `cache` belongs to, if its word has not been recycled since
function chain_of(cache)
    isnothing(cache) && return nothing
    st = lock(() -> get(chains, cache.nodes, nothing), host_lock)
    (isnothing(st) || hint_generation[st.slot + 1] != st.generation) ? nothing : st
end
# (sort_levels, sort_equalize, skew_flag, state) of this build: cold builds launch COLD_SORT_LEVELS levels on the plain grid
# and start a chain; a rebuild applies sort_hint_rule to the word (read ONCE, without synchronising: the latest value that
# has arrived is good enough for a hint)
function build_policy(cache, n)
    st = chain_of(cache)
    if isnothing(st)
        st = new_chain()
        return (Int32(COLD_SORT_LEVELS), Int32(0), Ptr{Cvoid}(hint_ptr(st)), st)
    end
    levels, equalize, holdoff = sort_hint_rule(unsafe_load(hint_ptr(st), :monotonic), Int64(n), st.holdoff)
    st.holdoff = holdoff
    (Int32(levels), Int32(equalize), Ptr{Cvoid}(hint_ptr(st)), st)
end

# ---- BVH(...) — build.jl:198-271 ------------------------------------------------------------------------
function ImplicitBVH.BVH(
    bounding_volumes::ROCVector{L},
    node_type::Type{N}=BBox{Float32};
    built_level::Union{Integer, AbstractFloat}=1,
    cache::Union{Nothing, BVH}=nothing,
    options=BVHOptions(),
) where {L, N}
    I = get_index_type(options)
    wrapped = L <: BoundingVolume
    V = wrapped ? fieldtype(L, :volume) : L
    # What the library cannot serve goes to the reference's own method, unchanged: a Morton algorithm other than the default
    # one, or any type without a code (Float16 volumes, index=UInt32, user volumes) — native_types; and a combination of
    # known codes the library has no instantiation for — IBVH_ERR_UNSUPPORTED from the scratch query.  This is decided
    # BEFORE anything is allocated or checked: the generic method does its own checks.
    types = options.morton isa DefaultMortonAlgorithm ? native_types(V, N, I, eltype(options.morton)) : nothing
    need = Ref{Csize_t}(0)
    if isnothing(types) || c_build_scratch_bytes(types, length(bounding_volumes), need) == IBVH_ERR_UNSUPPORTED
        return invoke(ImplicitBVH.BVH, Tuple{AbstractVector, Type}, bounding_volumes, node_type;
                      built_level=built_level, cache=cache, options=options)
    end
    M = eltype(options.morton)
    if wrapped   # check_bounding_volume_types, build.jl:355-361
        fieldtype(L, :index) === I || throw(ArgumentError("BoundingVolume index type does not match BVHOptions"))
        fieldtype(L, :morton) === M || throw(ArgumentError("BoundingVolume morton type does not match BVHOptions"))
    end
    numbv = length(bounding_volumes)
    tree = ImplicitTree{I}(numbv)                       # DomainError for numbv < 1
    built_ilevel = compute_build_level(tree, built_level)

    leaves = wrapped ? bounding_volumes : similar(bounding_volumes, BoundingVolume{V, I, M}, numbv)
    skips = isnothing(cache) ? similar(bounding_volumes, I, tree.levels) : begin
        eltype(cache.skips) === I || throw(ArgumentError("eltype(cache.skips) === I must hold"))
        length(cache.skips) == tree.levels || resize!(cache.skips, tree.levels); cache.skips end
    num_nodes = Int(tree.real_nodes - tree.real_leaves)
    nodes = isnothing(cache) ? similar(bounding_volumes, N, num_nodes) : begin
        eltype(cache.nodes) === N || throw(ArgumentError("eltype(cache.nodes) === N must hold"))
        length(cache.nodes) == num_nodes || resize!(cache.nodes, num_nodes); cache.nodes end

    check(c_build_scratch_bytes(types, numbv, need), "ibvh_build_scratch_bytes")
    scratch = scratch!(:build, need[])

    alg = options.morton
    # sort_levels / sort_equalize / skew_flag: the chain's policy (build_policy above; include/ibvh.h)
    sort_levels, sort_equalize, skew_flag, chain = build_policy(cache, numbv)
    desc = IbvhBuildDesc(types, numbv, Int64(built_ilevel), wrapped ? 1 : 0, alg.compute_extrema ? 1 : 0,
                         Float64.(alg.mins), Float64.(alg.maxs),   # NB alg.mins/maxs, not options.mins (default.jl:55-56)
                         sort_levels, sort_equalize, skew_flag)
    check(c_build(desc, wrapped ? C_NULL : devptr(bounding_volumes), devptr(leaves), devptr(nodes), devptr(skips),
                  C_NULL, devptr(scratch), need[], stream_ptr()), "ibvh_build")
    lock(() -> (chains[nodes] = chain), host_lock)
    BVH(I(built_ilevel), tree, skips, nodes, leaves)
end

# ---- multi-GPU build: one Julia process per GPU, leaves sharded over the ranks -----------------------------------
# No reference counterpart (ImplicitBVH.jl is single-device): BASELINE.json's north star.  `nccl_comm` is the ncclComm_t of
# this rank (RCCL's C API: ncclGetUniqueId on rank 0, broadcast it, ncclCommInitRank on every rank — e.g. through MPI.jl or
# a file); `local_volumes` this rank's share of the leaves.  Returns this rank's slice of the globally sorted sequence as an
# ordinary BVH (leaf .index = GLOBAL 1-based number) that traverse() takes like any other; contacts across slices are NOT
# found by it: dist_cross_contacts below completes them (the trees of touching slices travel over the same communicator).
"""
    dist_comm(nccl_comm::Ptr{Cvoid}, rank, size) -> IbvhComm
"""
function dist_comm(nccl_comm::Ptr{Cvoid}, rank::Integer, size::Integer)
    out = Ref{IbvhComm}()
    check(c_comm_from_rccl(nccl_comm, Int32(rank), Int32(size), out), "ibvh_comm_from_rccl")
    out[]
end

"""
    dist_BVH(comm::IbvhComm, local_volumes::ROCVector, node_type=BBox{Float32}; tolerance=0.005, cache=nothing, options=BVHOptions())
"""
function dist_BVH(comm::IbvhComm, local_volumes::ROCVector{V}, node_type::Type{N}=BBox{Float32};
                  tolerance::Float64=0.005, cache::Union{Nothing, BVH}=nothing, options=BVHOptions()) where {V, N}
    I = get_index_type(options)
    M = eltype(options.morton)
    types = native_types(V, N, I, M)
    isnothing(types) && throw(ArgumentError("dist_BVH: no libibvh instantiation for $V leaves, $N nodes, $I indices, $M codes"))
    n_local = length(local_volumes)
    need = Ref{Csize_t}(0)
    check(c_dist_scratch_bytes(types, n_local, comm.size, need), "ibvh_dist_scratch_bytes")
    scratch = scratch!(:dist, need[])
    plan = IbvhDistPlan()
    # blocks once: the record counts must reach the host before RCCL can be told the transfer sizes
    check(c_dist_plan(types, comm, devptr(local_volumes), n_local, tolerance, devptr(scratch), need[], plan, stream_ptr()), "ibvh_dist_plan")
    n_slice = Int(plan.n_slice)
    records = similar(local_volumes, BoundingVolume{V, I, M}, n_slice)       # sized from the plan: count, then size, then write
    check(c_dist_exchange(types, comm, devptr(local_volumes), plan, devptr(scratch), need[], devptr(records), stream_ptr()), "ibvh_dist_exchange")
    # the ordinary local build over the received slice: pre-wrapped records, the GLOBAL extrema fixed
    e = plan.extrema
    # (morton/default.jl:30-40: the keyword method takes the code type or an exemplar; utils.jl:54-71: the keyword is `index`)
    fixed = DefaultMortonAlgorithm(zero(M); compute_extrema=false, mins=(e[1], e[2], e[3]), maxs=(e[4], e[5], e[6]))
    opts = BVHOptions(index=options.index_exemplar, morton=fixed, num_threads=options.num_threads,
                      min_mortons_per_thread=options.min_mortons_per_thread, min_sorts_per_thread=options.min_sorts_per_thread,
                      min_boundings_per_thread=options.min_boundings_per_thread,
                      min_traversals_per_thread=options.min_traversals_per_thread, block_size=options.block_size)
    ImplicitBVH.BVH(records, N; cache=cache, options=opts)
end

"""
    dist_cross_contacts(comm::IbvhComm, bvh::BVH) -> ROCVector{IndexPair{I}}

The contacts between the leaves of THIS rank's slice (`bvh`, from `dist_BVH`) and the leaves of the other ranks' slices that
this rank is responsible for (pairs of slices r < s are handled by rank r): `(index in this slice, index in the other slice)`,
both global 1-based leaf numbers.  Collective — every rank calls it.  `traverse(bvh)` on every rank plus these pairs is the
contact set of the whole cloud, every pair once (include/ibvh.h "Cross-shard contact completion").
"""
function dist_cross_contacts(comm::IbvhComm, bvh::RocBVH{I}; cache_slots::Integer=LVT_CACHE_SLOTS) where {I}
    desc = bvh_desc(bvh)
    isnothing(desc) && throw(ArgumentError("dist_cross_contacts: no libibvh instantiation for this BVH's types"))
    nsmall = (16 * 48 + 16) * (Int(comm.size) + 1) + 16 * Int(comm.size) + 512    # IBVH_DIST_CROSS_SCRATCH
    small = scratch!(:dist_cross_plan, nsmall)
    plan = IbvhDistCrossPlan()
    # blocks twice: the root boxes of all slices, then how many leaves every peer gets, must reach the host before the transfers can be sized
    check(c_dist_cross_plan(comm, desc, Int32(cache_slots), devptr(small), nsmall, plan, stream_ptr()), "ibvh_dist_cross_plan")
    export_buf = similar(bvh.leaves, UInt8, max(Int(plan.export_bytes), 1))   # the own leaves whose box touches a lower rank's root box
    import_buf = similar(bvh.leaves, UInt8, max(Int(plan.import_bytes), 1))   # what the higher ranks send, + room for the trees built over it
    check(c_dist_cross_exchange(comm, desc, plan, devptr(export_buf), devptr(import_buf), devptr(small), nsmall, stream_ptr()), "ibvh_dist_cross_exchange")
    scratch = scratch!(:dist_cross, max(Int(plan.scratch_bytes), 1))
    totals = zeros(Int64, 256)
    total = Ref{Int64}(0)
    check(c_dist_cross_count(desc, plan, devptr(import_buf), devptr(scratch), Int(plan.scratch_bytes), totals, total, stream_ptr()), "ibvh_dist_cross_count")
    contacts = similar(bvh.leaves, IndexPair{I}, Int(total[]))          # count, then size, then write
    total[] > 0 && check(c_dist_cross_write(desc, plan, devptr(import_buf), devptr(scratch), Int(plan.scratch_bytes), totals, devptr(contacts),
                                            stream_ptr()), "ibvh_dist_cross_write")
    contacts
end

# ---- leaf-vs-tree: count -> (cache) -> write, or enqueue against a cached contact buffer ------------------------
const LVT_CACHE_SLOTS = 8    # contacts per work item kept from the counting pass (include/ibvh.h); cache_slots() widens it
const RAY_CACHE_SLOTS = 32   # hits per ray

cached(cache, field::Symbol, ::Type{T}, n, like) where {T} =
    if isnothing(cache)
        similar(like, T, n)
    else
        buf = getfield(cache, field)
        eltype(buf) === T || throw(ArgumentError("eltype(cache.$field) === $T must hold"))
        length(buf) < n && resize!(buf, n)
        buf
    end

# `count`, `write`, `enqueue`: closures over the entry points of one traversal shape; they take the buffers only.
function lvt_two_pass(::Type{I}, like, n_items, types, slots, cache, count, write, enqueue; rays_of=nothing) where {I}
    counts = cached(cache, :cache2, I, n_items, like)
    need = Ref{Csize_t}(0)
    if isnothing(rays_of)
        check(c_lvt_scratch_bytes(types, n_items, slots, need), "ibvh_lvt_scratch_bytes")
    else   # rays: room for the walker's quantised shadow of the node levels as well (include/ibvh.h)
        check(c_rays_scratch_bytes(rays_of, n_items, slots, need), "ibvh_rays_scratch_bytes")
    end
    scratch = scratch!(:lvt, need[])
    total = Ref{Int64}(0)
    if !isnothing(cache) && length(cache.cache1) > 0
        # cache reuse: pass 1 + scan + a guarded pass 2 against the cached buffer are enqueued without a host read in
        # between (the GPU never idles); the total is read afterwards — the reference's @allowscalar (:60), after the
        # work is queued — and the ordinary _write runs only if the cached buffer turned out too small.
        eltype(cache.cache1) === IndexPair{I} || throw(ArgumentError("eltype(cache.cache1) === IndexPair{I} must hold"))
        tdev = next_total_word()
        thost = next_host_word()
        check(enqueue(counts, cache.cache1, length(cache.cache1), tdev, Ptr{Cvoid}(thost), scratch, need[]), "ibvh_traverse_*_lvt_enqueue")
        total[] = poll_total(thost, tdev, stream_ptr())
        I === Int32 && total[] > typemax(Int32) && throw(OverflowError("more than typemax(Int32) contacts"))
        if total[] > length(cache.cache1)
            resize!(cache.cache1, total[])
            check(write(counts, cache.cache1, scratch, need[]), "ibvh_traverse_*_lvt_write")
        end
        return Int(total[]), cache.cache1, counts
    end
    check(count(counts, total, scratch, need[]), "ibvh_traverse_*_lvt_count")   # synchronises: @allowscalar (:60)
    contacts = cached(cache, :cache1, IndexPair{I}, total[], like)
    total[] > 0 && check(write(counts, contacts, scratch, need[]), "ibvh_traverse_*_lvt_write")
    Int(total[]), contacts, counts
end

# traverse(bvh, LVTTraversal()) — lvt/traverse_single.jl:1-79
function ImplicitBVH.traverse(
    bvh::RocBVH{I}, alg::LVTTraversal;
    start_level::Int=default_start_level(bvh, alg),
    narrow=DEFAULT_NARROW,
    cache::Union{Nothing, BVHTraversal}=nothing,
    options=BVHOptions(),
) where {I}
    code = narrow_code(narrow)
    d = bvh_desc(bvh)
    # an arbitrary closure, or types without a libibvh instantiation: the reference's generic (KernelAbstractions) method
    if isnothing(code) || isnothing(d)
        return invoke(ImplicitBVH.traverse, Tuple{BVH, LVTTraversal}, bvh, alg;
                      start_level=start_level, narrow=narrow, cache=cache, options=options)
    end
    bvh.built_level <= start_level <= bvh.tree.levels <= 32 ||
        throw(ArgumentError("bvh.built_level <= start_level <= bvh.tree.levels <= 32 must hold"))
    if bvh.tree.real_nodes <= 1   # :17-21
        return BVHTraversal(Int(start_level), 0, 0, similar(bvh.nodes, IndexPair{I}, 0), similar(bvh.nodes, I, 0))
    end
    s = stream_ptr()
    n_items = length(bvh.leaves)
    total, contacts, counts = lvt_two_pass(I, bvh.nodes, n_items, d.types, cache_slots(cache, n_items, LVT_CACHE_SLOTS), cache,
        (cn, tot, sc, sb) -> c_traverse_lvt_count(d, start_level, code, devptr(cn), tot, devptr(sc), sb, s),
        (cn, ct, sc, sb) -> c_traverse_lvt_write(d, start_level, code, devptr(cn), devptr(ct), devptr(sc), sb, s),
        (cn, ct, cap, td, th, sc, sb) -> c_traverse_lvt_enqueue(d, start_level, code, devptr(cn), devptr(ct), cap, td, th, devptr(sc), sb, s))
    BVHTraversal(Int(start_level), 0, total, contacts, counts)
end

# traverse(bvh1, bvh2, LVTTraversal()) — lvt/traverse_pair.jl:1-116 (the library picks the BVH with more leaves as
# the driver and flips the pairs back: contacts are always (index in bvh1, index in bvh2))
function ImplicitBVH.traverse(
    bvh1::RocBVH{I}, bvh2::RocBVH, alg::LVTTraversal;
    start_level1::Int=default_start_level(bvh1, alg),
    start_level2::Int=default_start_level(bvh2, alg),
    narrow=DEFAULT_NARROW,
    cache::Union{Nothing, BVHTraversal}=nothing,
    options=BVHOptions(),
) where {I}
    code = narrow_code(narrow)
    d1, d2 = bvh_desc(bvh1), bvh_desc(bvh2)
    # closures, types without an instantiation, and pairs of different leaf / node types: generic method
    if isnothing(code) || isnothing(d1) || isnothing(d2) || d1.types != d2.types
        return invoke(ImplicitBVH.traverse, Tuple{BVH, BVH, LVTTraversal}, bvh1, bvh2, alg;
                      start_level1=start_level1, start_level2=start_level2, narrow=narrow, cache=cache, options=options)
    end
    bvh1.built_level <= start_level1 <= bvh1.tree.levels <= 32 ||
        throw(ArgumentError("bvh1.built_level <= start_level1 <= bvh1.tree.levels <= 32 must hold"))
    bvh2.built_level <= start_level2 <= bvh2.tree.levels <= 32 ||
        throw(ArgumentError("bvh2.built_level <= start_level2 <= bvh2.tree.levels <= 32 must hold"))
    get_index_type(bvh2) === I || throw(ArgumentError("get_index_type(bvh2) === I must hold"))   # :50-52
    s = stream_ptr()
    n_items = max(length(bvh1.leaves), length(bvh2.leaves))
    total, contacts, counts = lvt_two_pass(I, bvh1.nodes, n_items, d1.types, cache_slots(cache, n_items, LVT_CACHE_SLOTS), cache,
        (cn, tot, sc, sb) -> c_traverse_pair_lvt_count(d1, d2, start_level1, start_level2, code, devptr(cn), tot, devptr(sc), sb, s),
        (cn, ct, sc, sb) -> c_traverse_pair_lvt_write(d1, d2, start_level1, start_level2, code, devptr(cn), devptr(ct), devptr(sc), sb, s),
        (cn, ct, cap, td, th, sc, sb) -> c_traverse_pair_lvt_enqueue(d1, d2, start_level1, start_level2, code, devptr(cn), devptr(ct), cap, td, th, devptr(sc), sb, s))
    BVHTraversal(Int(start_level1), Int(start_level2), 0, total, contacts, counts)
end

# rays as the library wants them: (3, N) column-major matrices of the leaf float type (raytrace/lvt:116-125 converts
# every coordinate with T.(...) too)
function ray_matrices(bvh::BVH{I, <:ROCVector, <:ROCVector, <:ROCVector{BoundingVolume{L, I, M}}}, points, directions) where {I, L, M}
    size(points, 1) == size(directions, 1) == 3 || throw(ArgumentError("size(points, 1) == size(directions, 1) == 3 must hold"))
    size(points, 2) == size(directions, 2) || throw(ArgumentError("size(points, 2) == size(directions, 2) must hold"))
    T = eltype(L)
    p = points isa ROCMatrix{T} ? points : ROCMatrix{T}(points)
    d = directions isa ROCMatrix{T} ? directions : ROCMatrix{T}(directions)
    p, d
end

# traverse_rays(bvh, points, directions, LVTTraversal()) — raytrace/leaf_vs_tree/leaf_vs_tree.jl:1-90
function ImplicitBVH.traverse_rays(
    bvh::RocBVH{I},
    points::AbstractMatrix, directions::AbstractMatrix,
    alg::LVTTraversal;
    start_level::Int=1,
    narrow=DEFAULT_RAY_NARROW,
    cache::Union{Nothing, BVHTraversal}=nothing,
    options=BVHOptions(),
) where {I}
    d = bvh_desc(bvh)
    code = ray_narrow_code(narrow)
    if isnothing(code) || isnothing(d) || d.types.leaf_float != d.types.node_float   # (isintersection needs one T)
        return invoke(ImplicitBVH.traverse_rays, Tuple{BVH, AbstractMatrix, AbstractMatrix, LVTTraversal},
                      bvh, points, directions, alg; start_level=start_level, narrow=narrow, cache=cache, options=options)
    end
    bvh.built_level <= start_level <= bvh.tree.levels <= 32 ||
        throw(ArgumentError("bvh.built_level <= start_level <= bvh.tree.levels <= 32 must hold"))
    p, dr = ray_matrices(bvh, points, directions)
    nr = size(p, 2)
    if nr == 0   # :22-26
        return BVHTraversal(start_level, 0, 0, similar(bvh.nodes, IndexPair{I}, 0), similar(bvh.nodes, IndexPair{I}, 0))
    end
    s = stream_ptr()
    total, contacts, counts = lvt_two_pass(I, bvh.nodes, nr, d.types, cache_slots(cache, nr, RAY_CACHE_SLOTS), cache,
        (cn, tot, sc, sb) -> c_traverse_rays_lvt_count(d, devptr(p), devptr(dr), nr, start_level, code, devptr(cn), tot, devptr(sc), sb, s),
        (cn, ct, sc, sb) -> c_traverse_rays_lvt_write(d, devptr(p), devptr(dr), nr, start_level, code, devptr(cn), devptr(ct), devptr(sc), sb, s),
        (cn, ct, cap, td, th, sc, sb) -> c_traverse_rays_lvt_enqueue(d, devptr(p), devptr(dr), nr, start_level, code, devptr(cn), devptr(ct), cap, td, th, devptr(sc), sb, s);
        rays_of=d)
    BVHTraversal(Int(start_level), 0, total, contacts, counts)
end

# ---- breadth-first: two caller-owned pair queues, grown when the library reports the capacity it needs ------------
# `run(bvtt1, bvtt2, capacity, counters, result)` issues the entry point of one traversal shape.
function bfs_run(::Type{I}, like, initial_pairs, total_levels, cache, run, what) where {I}
    capacity = 4 * max(initial_pairs, 1)                    # the reference's initial sizing (bfs/traverse_single.jl:73)
    bvtt1 = cached(cache, :cache1, IndexPair{I}, capacity, like)
    bvtt2 = cached(cache, :cache2, IndexPair{I}, capacity, like)
    nb = Ref{Csize_t}(0)
    check(c_bfs_counters_bytes(total_levels, nb), "ibvh_bfs_counters_bytes")
    counters = scratch!(:bfs_counters, nb[])
    fill!(counters, 0x00)
    res = IbvhBfsResult(0, 0, 1, 0, 0, 0)
    while true
        st = run(bvtt1, bvtt2, min(length(bvtt1), length(bvtt2)), counters, res)
        if st == IBVH_ERR_CAPACITY                          # the reference's resize! (bfs/traverse_single.jl:40)
            # resize! keeps the contents: the queue res.contacts_in still holds the res.resume_num pairs of the level
            # that overflowed, and the next call (same `res`, same counters) resumes there instead of starting over
            newcap = max(res.required_capacity, 4 * min(length(bvtt1), length(bvtt2)))   # (deeper levels need more still)
            resize!(bvtt1, newcap); resize!(bvtt2, newcap)
            continue
        end
        check(st, what); break
    end
    contacts, other = res.contacts_in == 1 ? (bvtt1, bvtt2) : (bvtt2, bvtt1)
    Int(res.num_checks), Int(res.num_contacts), contacts, other
end

# traverse(bvh, BFSTraversal()) — bfs/traverse_single.jl:1-61
function ImplicitBVH.traverse(
    bvh::RocBVH{I}, alg::BFSTraversal;
    start_level::Int=default_start_level(bvh, alg),
    narrow=DEFAULT_NARROW,
    cache::Union{Nothing, BVHTraversal}=nothing,
    options=BVHOptions(),
) where {I}
    code = narrow_code(narrow)
    d = bvh_desc(bvh)
    if isnothing(code) || isnothing(d)
        return invoke(ImplicitBVH.traverse, Tuple{BVH, BFSTraversal}, bvh, alg;
                      start_level=start_level, narrow=narrow, cache=cache, options=options)
    end
    bvh.tree.levels >= start_level >= bvh.built_level ||
        throw(ArgumentError("bvh.tree.levels >= start_level >= bvh.built_level must hold"))
    if bvh.tree.real_nodes <= 1
        return BVHTraversal(start_level, 0, 0, similar(bvh.nodes, IndexPair{I}, 0), similar(bvh.nodes, IndexPair{I}, 0))
    end
    s = stream_ptr()
    cap = Ref{Int64}(0)
    check(c_bfs_initial_capacity(d, start_level, cap), "ibvh_bfs_initial_capacity")
    checks, total, contacts, other = bfs_run(I, bvh.nodes, cap[], bvh.tree.levels, cache,
        (q1, q2, c, ctr, res) -> c_traverse_bfs(d, start_level, code, devptr(q1), devptr(q2), c, devptr(ctr), res, s),
        "ibvh_traverse_bfs")
    BVHTraversal(start_level, checks, total, contacts, other)
end

# traverse(bvh1, bvh2, BFSTraversal()) — bfs/traverse_pair.jl:1-151 (six-phase descent)
function ImplicitBVH.traverse(
    bvh1::RocBVH{I}, bvh2::RocBVH, alg::BFSTraversal;
    start_level1::Int=default_start_level(bvh1, alg),
    start_level2::Int=default_start_level(bvh2, alg),
    narrow=DEFAULT_NARROW,
    cache::Union{Nothing, BVHTraversal}=nothing,
    options=BVHOptions(),
) where {I}
    code = narrow_code(narrow)
    d1, d2 = bvh_desc(bvh1), bvh_desc(bvh2)
    if isnothing(code) || isnothing(d1) || isnothing(d2) || d1.types != d2.types
        return invoke(ImplicitBVH.traverse, Tuple{BVH, BVH, BFSTraversal}, bvh1, bvh2, alg;
                      start_level1=start_level1, start_level2=start_level2, narrow=narrow, cache=cache, options=options)
    end
    bvh1.tree.levels >= start_level1 >= bvh1.built_level ||
        throw(ArgumentError("bvh1.tree.levels >= start_level1 >= bvh1.built_level must hold"))
    bvh2.tree.levels >= start_level2 >= bvh2.built_level ||
        throw(ArgumentError("bvh2.tree.levels >= start_level2 >= bvh2.built_level must hold"))
    s = stream_ptr()
    cap = Ref{Int64}(0)
    check(c_bfs_pair_initial_capacity(d1, d2, start_level1, start_level2, cap), "ibvh_bfs_pair_initial_capacity")
    checks, total, contacts, other = bfs_run(I, bvh1.nodes, cap[], bvh1.tree.levels + bvh2.tree.levels, cache,
        (q1, q2, c, ctr, res) -> c_traverse_pair_bfs(d1, d2, start_level1, start_level2, code, devptr(q1), devptr(q2), c, devptr(ctr), res, s),
        "ibvh_traverse_pair_bfs")
    BVHTraversal(start_level1, start_level2, checks, total, contacts, other)
end

# traverse_rays(bvh, points, directions, BFSTraversal()) — raytrace/breadth_first/breadth_first.jl:1-66
function ImplicitBVH.traverse_rays(
    bvh::RocBVH{I},
    points::AbstractMatrix, directions::AbstractMatrix,
    alg::BFSTraversal;
    start_level::Int=1,
    narrow=DEFAULT_RAY_NARROW,
    cache::Union{Nothing, BVHTraversal}=nothing,
    options=BVHOptions(),
) where {I}
    d = bvh_desc(bvh)
    code = ray_narrow_code(narrow)
    if isnothing(code) || isnothing(d) || d.types.leaf_float != d.types.node_float
        return invoke(ImplicitBVH.traverse_rays, Tuple{BVH, AbstractMatrix, AbstractMatrix, BFSTraversal},
                      bvh, points, directions, alg; start_level=start_level, narrow=narrow, cache=cache, options=options)
    end
    bvh.tree.levels >= start_level >= bvh.built_level ||
        throw(ArgumentError("bvh.tree.levels >= start_level >= bvh.built_level must hold"))
    p, dr = ray_matrices(bvh, points, directions)
    nr = size(p, 2)
    if nr == 0
        return BVHTraversal(start_level, 0, 0, similar(bvh.nodes, IndexPair{I}, 0), similar(bvh.nodes, IndexPair{I}, 0))
    end
    s = stream_ptr()
    cap = Ref{Int64}(0)
    check(c_bfs_rays_initial_capacity(d, nr, start_level, cap), "ibvh_bfs_rays_initial_capacity")
    checks, total, contacts, other = bfs_run(I, bvh.nodes, cap[], bvh.tree.levels, cache,
        (q1, q2, c, ctr, res) -> c_traverse_rays_bfs(d, devptr(p), devptr(dr), nr, start_level, code, devptr(q1), devptr(q2), c, devptr(ctr), res, s),
        "ibvh_traverse_rays_bfs")
    BVHTraversal(start_level, checks, total, contacts, other)
end

end # module
