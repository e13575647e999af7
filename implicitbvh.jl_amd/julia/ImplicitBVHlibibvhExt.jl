# ImplicitBVHlibibvhExt.jl — package extension a maintainer adds to ImplicitBVH.jl so that ROCArray inputs run
# on libibvh (hand-written HIP for MI355X / gfx950) instead of the generic KernelAbstractions path.
#
# Project.toml additions:
#   [weakdeps]   AMDGPU = "21141c5a-9bdb-4563-92ae-f87d6854732e"
#   [extensions] ImplicitBVHlibibvhExt = "AMDGPU"
# and ENV["LIBIBVH"] (or a JLL) pointing at libibvh.so.  The C ABI is include/ibvh.h.
#
# The extension only adds MORE SPECIFIC METHODS for the three user entry points; keyword surface, return
# types (BVH, BVHTraversal) and error behaviour are the reference's.  All buffers are ROCArrays owned by Julia
# (cache= reuse keeps working); the library never allocates.
#
# NOTE: written against include/ibvh.h; Julia is not available in the build environment of libibvh, so this
# file is delivered as source and the identical call sequence is exercised by the Python mirror
# (implicitbvh.jl_amd/api.py) and its GPU parity tests.
module ImplicitBVHlibibvhExt

using ImplicitBVH
using ImplicitBVH: BVH, BVHOptions, BVHTraversal, BoundingVolume, BSphere, BBox, ImplicitTree, IndexPair,
                   LVTTraversal, BFSTraversal, DefaultMortonAlgorithm, default_start_level, get_index_type,
                   compute_build_level
using AMDGPU
using AMDGPU: ROCArray, ROCVector, ROCMatrix

const libibvh = get(ENV, "LIBIBVH", "libibvh.so")

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
end
mutable struct IbvhBfsResult
    num_contacts::Int64; num_checks::Int64; contacts_in::Int64; required_capacity::Int64
end

kind(::Type{<:BSphere}) = Int32(0);  kind(::Type{<:BBox}) = Int32(1)
fltcode(::Type{Float32}) = Int32(0); fltcode(::Type{Float64}) = Int32(1)
idxcode(::Type{Int32}) = Int32(0);   idxcode(::Type{Int64}) = Int32(1)
morcode(::Type{UInt16}) = Int32(0);  morcode(::Type{UInt32}) = Int32(1); morcode(::Type{UInt64}) = Int32(2)

ibvh_types(::Type{L}, ::Type{N}, ::Type{I}, ::Type{M}) where {L, N, I, M} =
    IbvhTypes(kind(L), fltcode(eltype(L)), kind(N), fltcode(eltype(N)), idxcode(I), morcode(M))

# status -> the exception the reference throws in the same situation
function check(status::Cint, what)
    status == 0 && return
    status == 1 && throw(ArgumentError("$what: invalid argument"))                 # @argcheck
    status == 2 && throw(DomainError(0, "must have at least one geometry!"))       # implicit_tree.jl:78-80
    status == 3 && throw(ArgumentError("$what: type combination not instantiated in libibvh"))
    status == 5 && throw(OverflowError("$what: count does not fit the index type"))
    error("$what: libibvh status $status")
end

stream_ptr() = Ptr{Cvoid}(UInt(AMDGPU.stream().stream))   # hipStream_t of the task-local stream
devptr(a::ROCArray) = Ptr{Cvoid}(UInt(pointer(a)))

tree_of(t::ImplicitTree) = IbvhTree(t.levels, t.real_leaves, t.real_nodes, t.virtual_leaves, t.virtual_nodes)

function bvh_desc(bvh::BVH{I, <:ROCVector, <:ROCVector{N}, <:ROCVector{BoundingVolume{L, I, M}}}) where {I, N, L, M}
    IbvhBvh(ibvh_types(L, N, I, M), tree_of(bvh.tree), Int64(bvh.built_level),
            devptr(bvh.leaves), devptr(bvh.nodes), devptr(bvh.skips))
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
    M = eltype(options.morton)
    wrapped = L <: BoundingVolume
    V = wrapped ? fieldtype(L, :volume) : L
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

    types = ibvh_types(V, N, I, M)
    need = Ref{Csize_t}(0)
    check(ccall((:ibvh_build_scratch_bytes, libibvh), Cint, (Ref{IbvhTypes}, Int64, Ref{Csize_t}), types, numbv, need),
          "ibvh_build_scratch_bytes")
    scratch = ROCVector{UInt8}(undef, need[])           # a real shim keeps this in a task-local pool

    alg = options.morton
    desc = IbvhBuildDesc(types, numbv, Int64(built_ilevel), wrapped ? 1 : 0, alg.compute_extrema ? 1 : 0,
                         Float64.(alg.mins), Float64.(alg.maxs))   # NB alg.mins/maxs, not options.mins (default.jl:55-56)
    check(ccall((:ibvh_build, libibvh), Cint,
                (Ref{IbvhBuildDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
                desc, wrapped ? C_NULL : devptr(bounding_volumes), devptr(leaves), devptr(nodes), devptr(skips),
                C_NULL, devptr(scratch), need[], stream_ptr()), "ibvh_build")
    BVH(I(built_ilevel), tree, skips, nodes, leaves)
end

# ---- traverse(bvh, LVTTraversal()) — lvt/traverse_single.jl:1-79 ----------------------------------------
const LVT_CACHE_SLOTS = Int32(8)
narrow_code(narrow) = Int32(0)   # default (a, b) -> true; map known closures to IBVH_NARROW_* here, otherwise
                                 # fall back to the generic method (invoke) or post-filter `.contacts`

function ImplicitBVH.traverse(
    bvh::BVH{I, <:ROCVector, <:ROCVector, <:ROCVector}, alg::LVTTraversal;
    start_level::Int=default_start_level(bvh, alg),
    narrow=(bv1, bv2) -> true,
    cache::Union{Nothing, BVHTraversal}=nothing,
    options=BVHOptions(),
) where {I}
    bvh.built_level <= start_level <= bvh.tree.levels <= 32 || throw(ArgumentError("start_level out of range"))
    if bvh.tree.real_nodes <= 1
        return BVHTraversal(Int(start_level), 0, 0, similar(bvh.nodes, IndexPair{I}, 0), similar(bvh.nodes, I, 0))
    end
    n = length(bvh.leaves)
    counts = isnothing(cache) ? similar(bvh.nodes, I, n) : begin
        eltype(cache.cache2) === I || throw(ArgumentError("eltype(cache.cache2) === I must hold"))
        length(cache.cache2) < n && resize!(cache.cache2, n); cache.cache2 end
    d = bvh_desc(bvh)
    need = Ref{Csize_t}(0)
    check(ccall((:ibvh_lvt_scratch_bytes, libibvh), Cint, (Ref{IbvhTypes}, Int64, Int32, Ref{Csize_t}),
                d.types, n, LVT_CACHE_SLOTS, need), "ibvh_lvt_scratch_bytes")
    scratch = ROCVector{UInt8}(undef, need[])
    total = Ref{Int64}(0)
    if !isnothing(cache) && length(cache.cache1) > 0
        # cache reuse: enqueue pass 1 + scan + a guarded pass 2 against the cached buffer without a host read in between
        # (the GPU never idles); read the total afterwards and fall through to the ordinary _write only if it did not fit.
        eltype(cache.cache1) === IndexPair{I} || throw(ArgumentError("eltype(cache.cache1) === IndexPair{I} must hold"))
        check(ccall((:ibvh_traverse_lvt_enqueue, libibvh), Cint,
                    (Ref{IbvhBvh}, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
                    d, start_level, narrow_code(narrow), devptr(counts), devptr(cache.cache1), length(cache.cache1),
                    devptr(scratch), need[], stream_ptr()), "ibvh_traverse_lvt_enqueue")
        check(ccall((:ibvh_lvt_total, libibvh), Cint, (Ptr{Cvoid}, Ref{Int64}, Ptr{Cvoid}), devptr(scratch), total, stream_ptr()),
              "ibvh_lvt_total")                             # the reference's @allowscalar (:60), after the work is queued
        if total[] <= length(cache.cache1)
            return BVHTraversal(Int(start_level), 0, Int(total[]), cache.cache1, counts)
        end
        resize!(cache.cache1, total[])
        check(ccall((:ibvh_traverse_lvt_write, libibvh), Cint,
                    (Ref{IbvhBvh}, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
                    d, start_level, narrow_code(narrow), devptr(counts), devptr(cache.cache1), devptr(scratch), need[],
                    stream_ptr()), "ibvh_traverse_lvt_write")
        return BVHTraversal(Int(start_level), 0, Int(total[]), cache.cache1, counts)
    end
    check(ccall((:ibvh_traverse_lvt_count, libibvh), Cint,
                (Ref{IbvhBvh}, Int64, Int32, Ptr{Cvoid}, Ref{Int64}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
                d, start_level, narrow_code(narrow), devptr(counts), total, devptr(scratch), need[], stream_ptr()),
          "ibvh_traverse_lvt_count")                        # synchronises: the reference's @allowscalar (:60)
    contacts = isnothing(cache) ? similar(bvh.nodes, IndexPair{I}, total[]) : begin
        eltype(cache.cache1) === IndexPair{I} || throw(ArgumentError("eltype(cache.cache1) === IndexPair{I} must hold"))
        length(cache.cache1) < total[] && resize!(cache.cache1, total[]); cache.cache1 end
    if total[] > 0
        check(ccall((:ibvh_traverse_lvt_write, libibvh), Cint,
                    (Ref{IbvhBvh}, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
                    d, start_level, narrow_code(narrow), devptr(counts), devptr(contacts), devptr(scratch), need[],
                    stream_ptr()), "ibvh_traverse_lvt_write")
    end
    BVHTraversal(Int(start_level), 0, Int(total[]), contacts, counts)
end

# traverse(bvh1, bvh2, LVTTraversal()) and traverse_rays(bvh, points, directions, LVTTraversal()) follow the same
# count -> allocate -> write shape with ibvh_traverse_pair_lvt_* / ibvh_traverse_rays_lvt_*; `points` and
# `directions` are (3, N) ROCMatrix{T} (converted to the leaf eltype first, raytrace/lvt:116-125) and are passed as is:
# Julia's column-major (3, N) is the layout the library expects.

# ---- traverse(bvh, BFSTraversal()) — bfs/traverse_single.jl:1-61 ----------------------------------------
function ImplicitBVH.traverse(
    bvh::BVH{I, <:ROCVector, <:ROCVector, <:ROCVector}, alg::BFSTraversal;
    start_level::Int=default_start_level(bvh, alg),
    cache::Union{Nothing, BVHTraversal}=nothing,
    narrow=(bv1, bv2) -> true,
    options=BVHOptions(),
) where {I}
    bvh.tree.levels >= start_level >= bvh.built_level || throw(ArgumentError("start_level out of range"))
    if bvh.tree.real_nodes <= 1
        return BVHTraversal(start_level, 0, 0, similar(bvh.nodes, IndexPair{I}, 0), similar(bvh.nodes, IndexPair{I}, 0))
    end
    d = bvh_desc(bvh)
    cap = Ref{Int64}(0)
    check(ccall((:ibvh_bfs_initial_capacity, libibvh), Cint, (Ref{IbvhBvh}, Int64, Ref{Int64}), d, start_level, cap),
          "ibvh_bfs_initial_capacity")
    capacity = 4 * cap[]                                    # the reference's initial_number (bfs/traverse_single.jl:73)
    bvtt1 = isnothing(cache) ? similar(bvh.nodes, IndexPair{I}, capacity) : cache.cache1
    bvtt2 = isnothing(cache) ? similar(bvh.nodes, IndexPair{I}, capacity) : cache.cache2
    nb = Ref{Csize_t}(0)
    ccall((:ibvh_bfs_counters_bytes, libibvh), Cint, (Int64, Ref{Csize_t}), bvh.tree.levels, nb)
    counters = AMDGPU.zeros(UInt8, nb[])
    res = IbvhBfsResult(0, 0, 1, 0)
    while true
        capacity = min(length(bvtt1), length(bvtt2))
        st = ccall((:ibvh_traverse_bfs, libibvh), Cint,
                   (Ref{IbvhBvh}, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ref{IbvhBfsResult}, Ptr{Cvoid}),
                   d, start_level, narrow_code(narrow), devptr(bvtt1), devptr(bvtt2), capacity, devptr(counters), res,
                   stream_ptr())
        if st == 4                                          # IBVH_ERR_CAPACITY: the reference's resize! (:40)
            resize!(bvtt1, res.required_capacity); resize!(bvtt2, res.required_capacity)
            continue
        end
        check(st, "ibvh_traverse_bfs"); break
    end
    contacts, other = res.contacts_in == 1 ? (bvtt1, bvtt2) : (bvtt2, bvtt1)
    BVHTraversal(start_level, Int(res.num_checks), Int(res.num_contacts), contacts, other)
end

end # module
