// ibvh_common.hpp — shared device/host definitions of libibvh (gfx950 only).
//
// Record types, geometry predicates and implicit-tree index math used by every kernel.  All
// floating-point code here is compiled with -ffp-contract=off: the reference (Julia) never
// contracts a*b+c, and results must match it bit for bit.  min/max are the reference's
// `a < b ? a : b` ternaries (utils.jl:177-181), never fminf/fmaxf (NaN semantics differ).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ibvh.h"

#define IBVH_HD __host__ __device__ __forceinline__
#define IBVH_D __device__ __forceinline__

namespace ibvh {

// ------------------------------------------------------------------------------------------
// records (layouts fixed by the Julia side, include/ibvh.h)
// ------------------------------------------------------------------------------------------
template <class T> struct BSphere {
    using elt = T;
    static constexpr int kind = IBVH_BSPHERE;
    T x[3];
    T r;
};
template <class T> struct BBox {
    using elt = T;
    static constexpr int kind = IBVH_BBOX;
    T lo[3];
    T up[3];
};
template <class I> struct IndexPair {
    I a, b;
};

// Where the fields of one BoundingVolume{V,I,M} record live; the Morton type only changes the
// stride and the width of the morton field, so kernels take it at run time instead of being
// instantiated per Morton type.
struct LeafLayout {
    int32_t stride;       // sizeof(BoundingVolume{V,I,M}); always a multiple of 8
    int32_t index_off;    // offsetof(.index)
    int32_t morton_off;   // offsetof(.morton)
    int32_t morton_bytes; // 2, 4 or 8
};

// Every record size is a multiple of 8 and the volume sits at offset 0, so a volume can always
// be moved as 8-byte words (global_load_dwordx2); hipMalloc'd bases are 256-byte aligned.
template <class V> IBVH_D V load_vol(const void *p) {
    static_assert(sizeof(V) % 8 == 0, "volumes are multiples of 8 bytes");
    V v;
    __builtin_memcpy(&v, __builtin_assume_aligned(p, 8), sizeof(V));
    return v;
}
// Same, for a WAVE-UNIFORM address: reading through the constant address space makes the compiler
// emit scalar loads (s_load_dwordx*), so the words land in SGPRs, cost no vector-memory issue slots
// and one request serves the whole wave.  Only for buffers no thread of the kernel writes.
typedef const __attribute__((address_space(4))) uint64_t *uniform_words;
template <class V> IBVH_D V load_vol_uniform(const void *p) {
    static_assert(sizeof(V) % 8 == 0, "volumes are multiples of 8 bytes");
    V v;
    uniform_words s = (uniform_words)(uintptr_t)p;
    uint64_t *d = (uint64_t *)&v;
#pragma unroll
    for (int k = 0; k < (int)(sizeof(V) / 8); ++k) d[k] = s[k];
    return v;
}
template <class I> IBVH_D I load_index_uniform(const char *rec, const LeafLayout &lay) {
    typedef const __attribute__((address_space(4))) I *ip;
    return *(ip)(uintptr_t)(rec + lay.index_off);
}
IBVH_D uint64_t load_morton_uniform(const char *rec, const LeafLayout &lay) {
    const uintptr_t p = (uintptr_t)(rec + lay.morton_off);
    if (lay.morton_bytes == 4) return *(const __attribute__((address_space(4))) uint32_t *)p;
    if (lay.morton_bytes == 8) return *(const __attribute__((address_space(4))) uint64_t *)p;
    return *(const __attribute__((address_space(4))) uint16_t *)p;
}
// Broadcast a whole record from lane `src` (wave-uniform) to every lane through SGPRs
// (v_readlane_b32 per 32-bit word): no LDS, no memory.
template <class V> IBVH_D V broadcast_from_lane(const V &v, int src) {
    static_assert(sizeof(V) % 4 == 0, "records are multiples of 4 bytes");
    V out;
    const int *s = (const int *)&v;
    int *d = (int *)&out;
#pragma unroll
    for (int k = 0; k < (int)(sizeof(V) / 4); ++k) d[k] = __builtin_amdgcn_readlane(s[k], src);
    return out;
}
// Pull a whole record out of lane `src` (per-lane index): ds_bpermute_b32 per 32-bit word.
template <class V> IBVH_D V shuffle_from(const V &v, int src) {
    static_assert(sizeof(V) % 4 == 0, "records are multiples of 4 bytes");
    V out;
    const int *s = (const int *)&v;
    int *d = (int *)&out;
#pragma unroll
    for (int k = 0; k < (int)(sizeof(V) / 4); ++k) d[k] = __shfl(s[k], src, 64);
    return out;
}
template <class V> IBVH_D void store_vol(void *p, const V &v) {
    __builtin_memcpy(__builtin_assume_aligned(p, 8), &v, sizeof(V));
}
// 16-byte-aligned variant (raw BSphere{F32}/{F64}, BBox{F64} arrays): global_load_dwordx4
template <class V> IBVH_D V load_vol16(const void *p) {
    V v;
    __builtin_memcpy(&v, __builtin_assume_aligned(p, 16), sizeof(V));
    return v;
}
template <class I> IBVH_D I load_index(const char *rec, const LeafLayout &lay) {
    return *(const I *)(rec + lay.index_off);
}
IBVH_D uint64_t load_morton(const char *rec, const LeafLayout &lay) {
    const char *p = rec + lay.morton_off;
    if (lay.morton_bytes == 4) return *(const uint32_t *)p;
    if (lay.morton_bytes == 8) return *(const uint64_t *)p;
    return *(const uint16_t *)p;
}
IBVH_D void store_morton(char *rec, const LeafLayout &lay, uint64_t m) {
    char *p = rec + lay.morton_off;
    if (lay.morton_bytes == 4) *(uint32_t *)p = (uint32_t)m;
    else if (lay.morton_bytes == 8) *(uint64_t *)p = m;
    else *(uint16_t *)p = (uint16_t)m;
}

// ------------------------------------------------------------------------------------------
// scalar helpers — utils.jl:160-181
// ------------------------------------------------------------------------------------------
template <class A, class B> IBVH_HD auto minimum2(A a, B b) -> decltype(a + b) { return a < b ? a : b; }
template <class A, class B> IBVH_HD auto maximum2(A a, B b) -> decltype(a + b) { return a > b ? a : b; }
template <class T> IBVH_HD T minimum3(T a, T b, T c) { return a < b ? minimum2(a, c) : minimum2(b, c); }
template <class T> IBVH_HD T maximum3(T a, T b, T c) { return a > b ? maximum2(a, c) : maximum2(b, c); }

template <class A, class B> IBVH_HD auto dist3sq(const A *x, const B *y) -> decltype(x[0] - y[0]) {
    return (x[0] - y[0]) * (x[0] - y[0]) + (x[1] - y[1]) * (x[1] - y[1]) + (x[2] - y[2]) * (x[2] - y[2]);
}
IBVH_HD float ibvh_sqrt(float v) { return sqrtf(v); }   // correctly rounded (IEEE) on gfx950
IBVH_HD double ibvh_sqrt(double v) { return sqrt(v); }
IBVH_HD float ibvh_abs(float v) { return fabsf(v); }
IBVH_HD double ibvh_abs(double v) { return fabs(v); }
template <class A, class B> IBVH_HD auto dist3(const A *x, const B *y) -> decltype(x[0] - y[0]) {
    return ibvh_sqrt(dist3sq(x, y));
}

// ------------------------------------------------------------------------------------------
// centres — bsphere.jl:142, bbox.jl:100-102
// ------------------------------------------------------------------------------------------
template <class T> IBVH_HD void center(const BSphere<T> &b, T c[3]) {
    c[0] = b.x[0];
    c[1] = b.x[1];
    c[2] = b.x[2];
}
template <class T> IBVH_HD void center(const BBox<T> &b, T c[3]) {
    c[0] = T(0.5) * (b.lo[0] + b.up[0]);
    c[1] = T(0.5) * (b.lo[1] + b.up[1]);
    c[2] = T(0.5) * (b.lo[2] + b.up[2]);
}

// ------------------------------------------------------------------------------------------
// conversions / merges — merge.jl:2-85.  Arithmetic in Julia's promoted type — the leaf float type TL unless the node's
// TN is wider and a T-typed constant enters the expression (sphere merge) — and one conversion into TN at the end.
// ------------------------------------------------------------------------------------------
template <class TN, class TL> IBVH_HD BSphere<TN> convert_to(const BSphere<TL> &a, BSphere<TN> *) {
    return {{TN(a.x[0]), TN(a.x[1]), TN(a.x[2])}, TN(a.r)};
}
template <class TN, class TL> IBVH_HD BBox<TN> convert_to(const BBox<TL> &a, BBox<TN> *) {
    return {{TN(a.lo[0]), TN(a.lo[1]), TN(a.lo[2])}, {TN(a.up[0]), TN(a.up[1]), TN(a.up[2])}};
}
template <class TN, class TL> IBVH_HD BBox<TN> convert_to(const BSphere<TL> &a, BBox<TN> *) { // merge.jl:47-51
    return {{TN(a.x[0] - a.r), TN(a.x[1] - a.r), TN(a.x[2] - a.r)},
            {TN(a.x[0] + a.r), TN(a.x[1] + a.r), TN(a.x[2] + a.r)}};
}
template <class TN, class TL>
IBVH_HD BSphere<TN> merge_to(const BSphere<TL> &a, const BSphere<TL> &b, BSphere<TN> *) { // merge.jl:2-22
    TL length = dist3(a.x, b.x);
    if (length + a.r <= b.r) return convert_to(b, (BSphere<TN> *)nullptr);
    if (length + b.r <= a.r) return convert_to(a, (BSphere<TN> *)nullptr);
    using TP = decltype(TL() + TN()); // promote_type(TL, TN): TL for every combination but Float32 leaves under Float64 nodes
    TP frac = TP(0.5) * ((b.r - a.r) / length + TP(1));
    TP c0 = a.x[0] + frac * (b.x[0] - a.x[0]);
    TP c1 = a.x[1] + frac * (b.x[1] - a.x[1]);
    TP c2 = a.x[2] + frac * (b.x[2] - a.x[2]);
    TP radius = TP(0.5) * (length + a.r + b.r);
    return {{TN(c0), TN(c1), TN(c2)}, TN(radius)};
}
template <class TN, class TL> IBVH_HD BBox<TN> merge_to(const BBox<TL> &a, const BBox<TL> &b, BBox<TN> *) { // :30-40
    BBox<TN> o;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o.lo[k] = TN(minimum2(a.lo[k], b.lo[k]));
        o.up[k] = TN(maximum2(a.up[k], b.up[k]));
    }
    return o;
}
template <class TN, class TL>
IBVH_HD BBox<TN> merge_to(const BSphere<TL> &a, const BSphere<TL> &b, BBox<TN> *) { // merge.jl:58-81
    TL length = dist3(a.x, b.x);
    if (length + a.r <= b.r) return convert_to(b, (BBox<TN> *)nullptr);
    if (length + b.r <= a.r) return convert_to(a, (BBox<TN> *)nullptr);
    BBox<TN> o;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o.lo[k] = TN(minimum2(a.x[k] - a.r, b.x[k] - b.r));
        o.up[k] = TN(maximum2(a.x[k] + a.r, b.x[k] + b.r));
    }
    return o;
}

// ------------------------------------------------------------------------------------------
// iscontact — iscontact.jl:2-28
// ------------------------------------------------------------------------------------------
template <class TA, class TB> IBVH_HD bool iscontact(const BSphere<TA> &a, const BSphere<TB> &b) {
    return dist3sq(a.x, b.x) <= (a.r + b.r) * (a.r + b.r);
}
template <class TA, class TB> IBVH_HD bool iscontact(const BBox<TA> &a, const BBox<TB> &b) {
    // (bitwise &: six independent compares, no short-circuit branches on the GPU)
    return (a.up[0] >= b.lo[0]) & (a.lo[0] <= b.up[0]) & (a.up[1] >= b.lo[1]) & (a.lo[1] <= b.up[1]) &
           (a.up[2] >= b.lo[2]) & (a.lo[2] <= b.up[2]);
}
template <class TA, class TB> IBVH_HD bool iscontact(const BSphere<TA> &a, const BBox<TB> &b) {
    BBox<TA> ab = {{a.x[0] - a.r, a.x[1] - a.r, a.x[2] - a.r}, {a.x[0] + a.r, a.x[1] + a.r, a.x[2] + a.r}};
    return iscontact(ab, b);
}
template <class TA, class TB> IBVH_HD bool iscontact(const BBox<TA> &a, const BSphere<TB> &b) {
    return iscontact(b, a);
}

// ------------------------------------------------------------------------------------------
// isintersection — isintersection.jl:1-65 (inf/NaN paths for zero direction components kept)
// ------------------------------------------------------------------------------------------
template <class T> IBVH_HD bool isintersection(const BBox<T> &b, const T *p, const T *d) {
    T inv0 = T(1) / d[0], inv1 = T(1) / d[1], inv2 = T(1) / d[2];
    T t1 = (b.lo[0] - p[0]) * inv0;
    T t2 = (b.up[0] - p[0]) * inv0;
    T tmin = minimum2(t1, t2);
    T tmax = maximum2(t1, t2);
    t1 = (b.lo[1] - p[1]) * inv1;
    t2 = (b.up[1] - p[1]) * inv1;
    tmin = maximum2(tmin, minimum2(t1, t2));
    tmax = minimum2(tmax, maximum2(t1, t2));
    t1 = (b.lo[2] - p[2]) * inv2;
    t2 = (b.up[2] - p[2]) * inv2;
    tmin = maximum2(tmin, minimum2(t1, t2));
    tmax = minimum2(tmax, maximum2(t1, t2));
    return (tmin <= tmax) && (tmax >= T(0));
}
// The same slab test with the three reciprocals 1/d[k] computed by the caller (they depend on the ray alone: a walk
// that tests hundreds of boxes against one ray divides once) — the identical operations in the identical order.
template <class T> IBVH_HD bool isintersection_inv(const BBox<T> &b, const T *p, const T *inv) {
    T t1 = (b.lo[0] - p[0]) * inv[0];
    T t2 = (b.up[0] - p[0]) * inv[0];
    T tmin = minimum2(t1, t2);
    T tmax = maximum2(t1, t2);
    t1 = (b.lo[1] - p[1]) * inv[1];
    t2 = (b.up[1] - p[1]) * inv[1];
    tmin = maximum2(tmin, minimum2(t1, t2));
    tmax = minimum2(tmax, maximum2(t1, t2));
    t1 = (b.lo[2] - p[2]) * inv[2];
    t2 = (b.up[2] - p[2]) * inv[2];
    tmin = maximum2(tmin, minimum2(t1, t2));
    tmax = minimum2(tmax, maximum2(t1, t2));
    return (tmin <= tmax) && (tmax >= T(0));
}
template <class T> IBVH_HD bool isintersection_inv(const BSphere<T> &s, const T *p, const T *inv, const T *d) { return isintersection(s, p, d); }
template <class T> IBVH_HD bool isintersection(const BSphere<T> &s, const T *p, const T *d) {
    T a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    T b = T(2) * ((p[0] - s.x[0]) * d[0] + (p[1] - s.x[1]) * d[1] + (p[2] - s.x[2]) * d[2]);
    T c = ((p[0] - s.x[0]) * (p[0] - s.x[0]) + (p[1] - s.x[1]) * (p[1] - s.x[1]) +
           (p[2] - s.x[2]) * (p[2] - s.x[2])) -
          s.r * s.r;
    T disc = b * b - T(4) * a * c;
    if (disc >= T(0)) {
        if (b <= T(0)) return true;
        return T(0) >= c;
    }
    return false;
}

// ------------------------------------------------------------------------------------------
// triangle constructors — bsphere.jl:43-112, bbox.jl:59-70
// ------------------------------------------------------------------------------------------
template <class T> IBVH_HD T eps_of();
template <> IBVH_HD float eps_of<float>() { return 1.1920928955078125e-07f; }
template <> IBVH_HD double eps_of<double>() { return 2.220446049250313e-16; }

template <class T> IBVH_HD BSphere<T> bsphere_from_triangle(const T *a, const T *b, const T *c) {
    T abab = (b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1]) + (b[2] - a[2]) * (b[2] - a[2]);
    T abac = (b[0] - a[0]) * (c[0] - a[0]) + (b[1] - a[1]) * (c[1] - a[1]) + (b[2] - a[2]) * (c[2] - a[2]);
    T acac = (c[0] - a[0]) * (c[0] - a[0]) + (c[1] - a[1]) * (c[1] - a[1]) + (c[2] - a[2]) * (c[2] - a[2]);
    T d = T(2) * (abab * acac - abac * abac);
    T centre[3];
    T radius;
    if (ibvh_abs(d) <= eps_of<T>()) {
        T upper[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            T lower = minimum3(a[k], b[k], c[k]);
            upper[k] = maximum3(a[k], b[k], c[k]);
            centre[k] = T(0.5) * (lower + upper[k]);
        }
        radius = dist3(centre, upper);
    } else {
        T s = (abab * acac - acac * abac) / d;
        T t = (acac * abab - abab * abac) / d;
        if (s <= T(0)) {
#pragma unroll
            for (int k = 0; k < 3; ++k) centre[k] = T(0.5) * (a[k] + c[k]);
            radius = dist3(centre, a);
        } else if (t <= T(0)) {
#pragma unroll
            for (int k = 0; k < 3; ++k) centre[k] = T(0.5) * (a[k] + b[k]);
            radius = dist3(centre, a);
        } else if (s + t >= T(1)) {
#pragma unroll
            for (int k = 0; k < 3; ++k) centre[k] = T(0.5) * (b[k] + c[k]);
            radius = dist3(centre, b);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) centre[k] = a[k] + s * (b[k] - a[k]) + t * (c[k] - a[k]);
            radius = dist3(centre, a);
        }
    }
    return {{centre[0], centre[1], centre[2]}, radius};
}
template <class T> IBVH_HD BBox<T> bbox_from_triangle(const T *a, const T *b, const T *c) {
    BBox<T> o;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o.lo[k] = minimum3(a[k], b[k], c[k]);
        o.up[k] = maximum3(a[k], b[k], c[k]);
    }
    return o;
}

// ------------------------------------------------------------------------------------------
// Morton — morton/default.jl:91-181
// ------------------------------------------------------------------------------------------
IBVH_HD uint32_t split3_u16(uint32_t v) { // default.jl:118-127 (UInt16 arithmetic)
    uint32_t s = v & 0x001fu;
    s = (s | ((s << 8) & 0xffffu)) & 0x100fu;
    s = (s | ((s << 4) & 0xffffu)) & 0x10c3u;
    s = (s | ((s << 2) & 0xffffu)) & 0x1249u;
    return s;
}
IBVH_HD uint32_t split3_u32(uint32_t v) { // default.jl:130-143
    uint32_t s = v & 0x000003ffu;
    s = (s | s << 16) & 0x30000ffu;
    s = (s | s << 8) & 0x0300f00fu;
    s = (s | s << 4) & 0x30c30c3u;
    s = (s | s << 2) & 0x9249249u;
    return s;
}
IBVH_HD uint64_t split3_u64(uint64_t v) { // default.jl:146-157
    uint64_t s = v & 0x00000000001fffffull;
    s = (s | s << 32) & 0x1f00000000ffffull;
    s = (s | s << 16) & 0x1f0000ff0000ffull;
    s = (s | s << 8) & 0x100f00f00f00f00full;
    s = (s | s << 4) & 0x10c30c30c30c30c3ull;
    s = (s | s << 2) & 0x1249249249249249ull;
    return s;
}
// morton_encode_single — default.jl:91-108; morton_type selects scaling 2^5 / 2^10 / 2^21 and the
// split ladder.  The multiply by a power of two is exact; the trunc is a plain fptoui.
template <class T> IBVH_HD uint64_t morton_encode_single(const T c[3], const T mins[3], const T maxs[3], int morton_type) {
    T s1 = (c[0] - mins[0]) / (maxs[0] - mins[0]);
    T s2 = (c[1] - mins[1]) / (maxs[1] - mins[1]);
    T s3 = (c[2] - mins[2]) / (maxs[2] - mins[2]);
    if (morton_type == IBVH_U32) {
        const T sc = T(1024);
        uint32_t i1 = (uint32_t)(s1 * sc), i2 = (uint32_t)(s2 * sc), i3 = (uint32_t)(s3 * sc);
        return (uint64_t)((split3_u32(i1) << 2) | (split3_u32(i2) << 1) | split3_u32(i3));
    } else if (morton_type == IBVH_U64) {
        const T sc = T(2097152);
        uint64_t i1 = (uint64_t)(s1 * sc), i2 = (uint64_t)(s2 * sc), i3 = (uint64_t)(s3 * sc);
        return (split3_u64(i1) << 2) | (split3_u64(i2) << 1) | split3_u64(i3);
    } else {
        const T sc = T(32);
        uint32_t i1 = (uint32_t)(s1 * sc) & 0xffffu, i2 = (uint32_t)(s2 * sc) & 0xffffu, i3 = (uint32_t)(s3 * sc) & 0xffffu;
        return (uint64_t)((((split3_u16(i1) << 2) | (split3_u16(i2) << 1) | split3_u16(i3))) & 0xffffu);
    }
}

template <class T> IBVH_HD T float_max();
template <> IBVH_HD float float_max<float>() { return 3.4028234663852886e+38f; }
template <> IBVH_HD double float_max<double>() { return 1.7976931348623157e+308; }
template <class T> IBVH_HD T float_min_normal(); // Julia's floatmin(T)
template <> IBVH_HD float float_min_normal<float>() { return 1.1754943508222875e-38f; }
template <> IBVH_HD double float_min_normal<double>() { return 2.2250738585072014e-308; }
template <class T> IBVH_HD T relative_precision(); // default.jl:179-181
template <> IBVH_HD float relative_precision<float>() { return 1e-5f; }
template <> IBVH_HD double relative_precision<double>() { return 1e-14; }

// ------------------------------------------------------------------------------------------
// implicit tree index math — implicit_tree.jl (levels <= 62 here; Julia's >> saturates to 0)
// ------------------------------------------------------------------------------------------
struct TreeDev {
    int64_t levels;
    int64_t real_leaves;
    int64_t virtual_leaves;
};
IBVH_HD int64_t shr_sat(int64_t v, int64_t s) { return s >= 63 ? 0 : (v >> s); }
IBVH_HD int64_t popc64(int64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll((unsigned long long)v);
#else
    return __builtin_popcountll((unsigned long long)v);
#endif
}
// number of real nodes on `level`: pow2(level-1) - virtual_leaves >> (levels - level)
IBVH_HD int64_t level_num_real(int64_t levels, int64_t virtual_leaves, int64_t level) {
    return (int64_t(1) << (level - 1)) - shr_sat(virtual_leaves, levels - level);
}
// skips[level] (compute_skips!, implicit_tree.jl:100-113): virtual nodes before `level`
IBVH_HD int64_t level_skips(int64_t levels, int64_t virtual_leaves, int64_t level) {
    int64_t v = shr_sat(virtual_leaves, levels - (level - 1));
    return 2 * v - popc64(v);
}
// memory index (1-based) of the first node of `level`
IBVH_HD int64_t level_start(int64_t levels, int64_t virtual_leaves, int64_t level) {
    return (int64_t(1) << (level - 1)) - level_skips(levels, virtual_leaves, level);
}

// ------------------------------------------------------------------------------------------
// host-side helpers
// ------------------------------------------------------------------------------------------
struct Tag32 {};
template <class T> struct Tag {
    using type = T;
};

inline bool combo_ok(const ibvh_types &t) {
    if (t.leaf_kind < 0 || t.leaf_kind > 1 || t.node_kind < 0 || t.node_kind > 1) return false;
    if (t.leaf_float < 0 || t.leaf_float > 1 || t.node_float < 0 || t.node_float > 1) return false;
    if (t.index_type < 0 || t.index_type > 1 || t.morton_type < 0 || t.morton_type > 2) return false;
    if (t.node_kind == IBVH_BSPHERE && t.leaf_kind != IBVH_BSPHERE) return false; // no BSphere(BBox) ctor
    return true; // (any node float type: narrower, equal or wider than the leaves', build.jl:198-205)
}
inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
inline bool layout_of(const ibvh_types &t, ibvh_layout &out, LeafLayout *dev = nullptr) {
    if (!combo_ok(t)) return false;
    int64_t fl = t.leaf_float == IBVH_F32 ? 4 : 8, fn = t.node_float == IBVH_F32 ? 4 : 8;
    int64_t vb = (t.leaf_kind == IBVH_BSPHERE ? 4 : 6) * fl;
    int64_t nb = (t.node_kind == IBVH_BSPHERE ? 4 : 6) * fn;
    int64_t ib = t.index_type == IBVH_I32 ? 4 : 8;
    int64_t mb = t.morton_type == IBVH_U16 ? 2 : (t.morton_type == IBVH_U32 ? 4 : 8);
    int64_t io = align_up(vb, ib);
    int64_t mo = align_up(io + ib, mb);
    int64_t al = fl > ib ? fl : ib;
    al = al > mb ? al : mb;
    out.volume_bytes = vb;
    out.node_bytes = nb;
    out.index_off = io;
    out.morton_off = mo;
    out.leaf_bytes = align_up(mo + mb, al);
    out.pair_bytes = 2 * ib;
    if (dev) *dev = {(int32_t)out.leaf_bytes, (int32_t)io, (int32_t)mo, (int32_t)mb};
    return true;
}

// IBVH_ONLY_BENCH_TYPES (development builds only, never the shipped library): instantiate nothing but Float32
// volumes and Int32 indices, so that one kernel can be recompiled in seconds while it is being tuned.
template <class F> int dispatch_volume(int kind, int flt, F &&f) {
    if (kind == IBVH_BSPHERE && flt == IBVH_F32) return f(Tag<BSphere<float>>{});
    if (kind == IBVH_BBOX && flt == IBVH_F32) return f(Tag<BBox<float>>{});
#ifndef IBVH_ONLY_BENCH_TYPES
    if (kind == IBVH_BSPHERE && flt == IBVH_F64) return f(Tag<BSphere<double>>{});
    if (kind == IBVH_BBOX && flt == IBVH_F64) return f(Tag<BBox<double>>{});
#endif
    return IBVH_ERR_UNSUPPORTED;
}
template <class F> int dispatch_leaf_node(const ibvh_types &t, F &&f) {
    if (!combo_ok(t)) return IBVH_ERR_UNSUPPORTED;
    return dispatch_volume(t.leaf_kind, t.leaf_float, [&](auto lt) -> int {
        using L = typename decltype(lt)::type;
        return dispatch_volume(t.node_kind, t.node_float, [&](auto nt) -> int {
            using N = typename decltype(nt)::type;
            constexpr bool ok = !(N::kind == IBVH_BSPHERE && L::kind != IBVH_BSPHERE);
            if constexpr (ok) return f(lt, nt);
            else return (int)IBVH_ERR_UNSUPPORTED;
        });
    });
}
template <class F> int dispatch_index(int index_type, F &&f) {
    if (index_type == IBVH_I32) return f(Tag<int32_t>{});
#ifndef IBVH_ONLY_BENCH_TYPES
    if (index_type == IBVH_I64) return f(Tag<int64_t>{});
#endif
    return IBVH_ERR_UNSUPPORTED;
}

#define IBVH_HIP_CHECK(expr)                                  \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return (int)IBVH_ERR_HIP;       \
    } while (0)
#define IBVH_LAUNCH_CHECK()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return (int)IBVH_ERR_HIP; \
    } while (0)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
IBVH_HD int64_t ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Development knobs (ibvh_set_tuning in include/ibvh.h): one process-wide table, written only through that entry
// point and read by the launch code.  The defaults are the shipped geometry; the library itself never reads the
// environment.
struct Tuning {
    int ray_block = 0;      // rays per wave of lvt_rays_kernel (0 = chosen from the batch size; a power of two, 64 .. 1024)
    int lvt_wide = 0;       // 1 = 64-bit queue entries for every tree (otherwise only for 29 .. 31 levels)
    int msd_rescue = 1;     // 0 = ranges too large for one finish workgroup take the one-workgroup slow path of rounds 2 - 5 instead of the rescue workgroups (ibvh_msd_finish.hip)
    int lvt_dual = 0;       // -DIBVH_VARIANTS builds only (variants/lvt_dual.inc): 1 = BBox-node leaf queries take the dual descent (7 - 25 % slower than lvt_queue_kernel, round 4)
    int lvt_blocks = 1;     // walker 2: 1 = the descent is shared per block of leaves (lvt_block_frontier_kernel), 0 = every wave descends on its own
    int lvt_block_shift = 0; // log2 of the leaves per block (0 = 11; 9 .. 12)
    int lvt_blocks_paired_below = -1; // block grids smaller than this take two levels per trip (-1 = 4096)
    int lvt_blocks_min_items = 0; // fewest work items for the shared descent (0 = 2^17)
    int lvt_xcd = 64;       // LVT item placement: 0 = round robin, 1 = one range per XCD, n = runs of n workgroups
    int sort_tile = 0;      // LSD path: keys per tile (0 = by size; 2048, 4096, 8192, 16384)
    int sort_lsd = 0;       // 1 = ibvh_sort_pairs always takes the plain LSD passes
    int sort_msd_avg = 1536; // ibvh_sort_pairs: largest average bucket before another partition bit is taken
    int bucket_tpb = 0;     // ibvh_sort_pairs: threads of a bucket workgroup (0 = by capacity)
    int msd = 1;            // 0 = the build never takes the MSD partition path
    int msd_bits = 0, msd_cap = 0, msd_tile = 0, msd_ftpb = 0; // forced partition geometry (0 = chosen from n)
    int msd_avg = 1024;     // largest average cell before another first-level bit is taken
    int msd_range = 1;      // 0 = first extra level on the next 8 bits, unmeasured
    int lvt_scan_fused = 1; // the scan behind walker 2's counting pass in one kernel (scan_fused_kernel / scan_fused_grouped_kernel); 0 = reduce + apply; N > 1 = at most N workgroups (development: the default is half of what the device holds at once)
    int msd_equalize = 0;   // equalised cells (ibvh_msd.hip): 0 = when the build asks (ibvh_build_desc.sort_equalize), 1 = always, -1 = never
    int msd_finish_pad_kb = 0; // LDS (KiB) a finish workgroup asks for at least: limits the workgroups per CU (0 = what it needs)
    int msd_resident_kb = 0; // LDS budget (KiB) of a finish workgroup that keeps its range's RECORDS in LDS: 0 = the plan decides
                             // (8,192-record geometry only), > 0 = every geometry with this budget, < 0 = never
    int bfs_wg_per_cu = 4;  // workgroups per CU of the BFS level kernels' fixed grid (ibvh_bfs.hip, level_grid)
    int rays_shadow = 0;    // -DIBVH_VARIANTS builds only (variants/rays_shadow.inc): 1 = ray traversals walk the quantised 8-wide shadow of the node levels when the scratch has room
                            // (ibvh_rays_scratch_bytes); measured slower than the binary walk on config 3 (4.6 vs 4.3 ms): off
    int rays_binned = 1;    // ray traversals (F32 trees) cut the walk at a level and finish it subtree by subtree out of LDS:
                            // 1 = where it pays (rays_bin_plan: >= 17 levels, or >= 13 under <= 8,192 rays; not a small tree under many
                            // rays), 2 = wherever the tree allows it, 0 = never
    int rays_subtree_depth = 0; // levels of such a subtree below its root (0 = 9: 512 leaves; at most 11)
    int rays_fast_slab = 1;     // 0 = the binned path tests every box with isintersection_inv (A/B of the packed / v_min3 slab test)
    int rays_tail = 8;          // binned rays, subtree pass: walks a wave parks for the workgroup's unit rounds once its chunk is dry (0 = never: every walk is finished by its lane)
    int rays_items_per_ray = 0; // capacity of the (ray, subtree) item list per ray (0 = 16); a call that overflows it is served by the binary walker
};
extern Tuning g_tuning;

// Optional per-launch timing (ibvh_profile_* in include/ibvh.h): when enabled every kernel launch is
// bracketed by a pair of HIP events recorded on the launch stream.  Off by default: one predictable
// branch per launch.
namespace prof {
extern bool enabled;
void begin(const char *name, hipStream_t st);
void end(hipStream_t st);
} // namespace prof
#define IBVH_LAUNCH(kernel, grid, block, smem, st, ...)                          \
    do {                                                                         \
        if (::ibvh::prof::enabled) ::ibvh::prof::begin(#kernel, st);             \
        hipLaunchKernelGGL(kernel, grid, block, smem, st, __VA_ARGS__);          \
        if (::ibvh::prof::enabled) ::ibvh::prof::end(st);                        \
    } while (0)

// XCD-aware workgroup -> tile assignment.  Workgroups b and b+8 share an XCD (round-robin dispatch,
// MI355X_MICROARCH.md "Workgroup dispatch"), so handing XCD x the contiguous tile range
// [x*q, (x+1)*q) keeps neighbouring tiles — which read each other's data (adjacent Morton ranges) —
// behind one 4 MiB L2.  Bijective for any grid size; affects speed only, never results.
IBVH_D int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// Block-cyclic variant: XCD x gets RUNS of `run` consecutive tiles, the runs dealt round-robin — neighbouring tiles
// still share an L2 while a workload whose cost is concentrated in part of the index range (two partially
// overlapping clouds) is still spread over all XCDs.  Bijective for any grid size (the ragged tail is left as is).
IBVH_D int xcd_run_remap(int b, int nwg, int run) {
    if ((run & (run - 1)) == 0) { // (the launch code hands out power-of-two runs: shifts instead of two integer divisions per wave)
        const int sh = __builtin_ctz((unsigned)run);
        const int full = (nwg >> (sh + 3)) << (sh + 3);
        if (b >= full) return b;
        const int xcd = b & 7, k = b >> 3;
        return ((((k >> sh) << 3) + xcd) << sh) + (k & (run - 1));
    }
    const int period = 8 * run;
    const int full = (nwg / period) * period;
    if (b >= full) return b;
    const int xcd = b & 7, k = b >> 3; // k-th workgroup that lands on this XCD
    return ((k / run) * 8 + xcd) * run + (k % run);
}

// wave64 helpers -----------------------------------------------------------------------------
IBVH_D int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

} // namespace ibvh
