// ibvh_oracle.hpp — CPU ORACLE (test infrastructure, NOT product code).
//
// A single-threaded, strict-IEEE (build with -ffp-contract=off -fno-fast-math) restatement of
// the hot path of StellaOrg/ImplicitBVH.jl v0.7.1, function by function, each citing the
// reference file:line it follows (paths relative to the reference root).  It exists so the HIP
// library can be checked bit-for-bit; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it.  The product (libibvh.so) never links or calls anything here.
//
// Pinning: checked in tests/test_oracle_golden.py against every known-answer the reference's own
// tests and doctests hold for this path: tree shapes, morton_split3, the
// 0x06186186 Morton KAT, the 5-sphere contact lists, the pair and ray doctests, the build
// structure test, the ray-box / ray-sphere truth tables and the merge / triangle expectations.
// NOT pinned (third-party AcceleratedKernels, source absent): the tie order of equal Morton
// codes after AK.sort! — this oracle defines it as *stable*, the single-task CPU behaviour.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <type_traits>
#include <utility>
#include <vector>

#include "../include/ibvh.h"

namespace orc {

// ------------------------------------------------------------------------------------------
// records — bounding_volumes/bsphere.jl:26-29, bbox.jl:35-38, bounding_volumes.jl:55-59
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
template <class V, class I, class M> struct BoundingVolume {
    V volume;
    I index;
    M morton;
};
template <class I> struct IndexPair {
    I a, b;
};

// ------------------------------------------------------------------------------------------
// scalar helpers — utils.jl:160-181
// ------------------------------------------------------------------------------------------
// minimum2(a, b) = a < b ? a : b ; maximum2(a, b) = a > b ? a : b  (NaN-asymmetric ternaries)
template <class A, class B> inline auto minimum2(A a, B b) -> decltype(a + b) { return a < b ? a : b; }
template <class A, class B> inline auto maximum2(A a, B b) -> decltype(a + b) { return a > b ? a : b; }
template <class T> inline T minimum3(T a, T b, T c) { return a < b ? minimum2(a, c) : minimum2(b, c); }
template <class T> inline T maximum3(T a, T b, T c) { return a > b ? maximum2(a, c) : maximum2(b, c); }

// dot3 — utils.jl:163-165
template <class A, class B> inline auto dot3(const A *x, const B *y) -> decltype(x[0] * y[0]) {
    return x[0] * y[0] + x[1] * y[1] + x[2] * y[2];
}
// dist3sq — utils.jl:168-172 ; evaluation order ((dx*dx)+(dy*dy))+(dz*dz)
template <class A, class B> inline auto dist3sq(const A *x, const B *y) -> decltype(x[0] - y[0]) {
    return (x[0] - y[0]) * (x[0] - y[0]) + (x[1] - y[1]) * (x[1] - y[1]) +
           (x[2] - y[2]) * (x[2] - y[2]);
}
// dist3 — utils.jl:175
template <class A, class B> inline auto dist3(const A *x, const B *y) -> decltype(x[0] - y[0]) {
    return std::sqrt(dist3sq(x, y));
}

// Julia's `>>` on a non-negative Int: shifting by >= the bit width gives 0 (no UB).
inline int64_t jl_shr(int64_t v, int64_t s) { return s >= 63 ? 0 : (s < 0 ? v << (-s) : v >> s); }
inline int64_t pow2(int64_t n) { return int64_t(1) << n; } // utils.jl:160
inline int64_t count_ones(int64_t v) { return __builtin_popcountll((unsigned long long)v); }
// unsafe_ilog2(n, RoundDown) — utils.jl:129-133 (msbindex - leading_zeros)
inline int64_t ilog2_down(int64_t n) { return 63 - __builtin_clzll((unsigned long long)n); }
// ilog2(x, RoundUp) — utils.jl:116
inline int64_t ilog2_up(int64_t n) {
    bool isp2 = (n & (n - 1)) == 0;
    return isp2 ? ilog2_down(n) : ilog2_down(n) + 1;
}

// ------------------------------------------------------------------------------------------
// ImplicitTree — implicit_tree.jl
// ------------------------------------------------------------------------------------------
// ImplicitTree{T}(num_leaves) — implicit_tree.jl:77-90
inline bool tree_shape(int64_t num_leaves, ibvh_tree &t) {
    if (num_leaves < 1) return false; // DomainError, :78-80
    int64_t lr = num_leaves;
    int64_t levels = ilog2_up(lr) + 1;          // :83
    int64_t lv = pow2(levels - 1) - lr;         // :85
    int64_t nv = 2 * lv - count_ones(lv);       // :86
    int64_t nr = 2 * lr - 1 + count_ones(lv);   // :87
    t.levels = levels;
    t.real_leaves = lr;
    t.real_nodes = nr;
    t.virtual_leaves = lv;
    t.virtual_nodes = nv;
    return true;
}
// compute_skips! — implicit_tree.jl:100-113 (1-based i; skips[i-1] here)
inline void compute_skips(const ibvh_tree &t, int64_t *skips) {
    for (int64_t i = 1; i <= t.levels; ++i) {
        int64_t vnl = jl_shr(t.virtual_leaves, t.levels - (i - 1)); // :107
        skips[i - 1] = 2 * vnl - count_ones(vnl);                    // :108
    }
}
// memory_index — implicit_tree.jl:128-148
inline int64_t memory_index(const ibvh_tree &t, int64_t implicit_index) {
    int64_t level = ilog2_down(implicit_index) + 1;
    int64_t vnl = jl_shr(t.virtual_leaves, t.levels - (level - 1));
    int64_t before = 2 * vnl - count_ones(vnl);
    return implicit_index - before;
}
// level_indices — implicit_tree.jl:156-171 ; NB `a - b >> c` parses as a - (b >> c) in Julia
inline void level_indices(const ibvh_tree &t, int64_t level, int64_t &start, int64_t &stop) {
    start = memory_index(t, pow2(level - 1));
    int64_t nreal = pow2(level - 1) - jl_shr(t.virtual_leaves, t.levels - level);
    stop = start + nreal - 1;
}
// unsafe_isvirtual — implicit_tree.jl:191-199
inline bool isvirtual(const ibvh_tree &t, int64_t implicit_index) {
    int64_t level = ilog2_down(implicit_index) + 1;
    int64_t level_first = pow2(level - 1);
    int64_t nreal = level_first - jl_shr(t.virtual_leaves, t.levels - level);
    return implicit_index - level_first + 1 > nreal;
}
// number of real nodes on a level (used all over build.jl / traversals as
// pow2(level-1) - virtual_leaves >> (levels-level))
inline int64_t level_num_real(const ibvh_tree &t, int64_t level) {
    return pow2(level - 1) - jl_shr(t.virtual_leaves, t.levels - level);
}
// Julia round(I, x): ties to even — build.jl:318
inline int64_t compute_build_level(const ibvh_tree &t, double frac) {
    double x = double(t.levels) + double(1 - t.levels) * frac;
    return (int64_t)std::nearbyint(x); // default rounding mode = ties-to-even
}

// ------------------------------------------------------------------------------------------
// centres — bsphere.jl:142, bbox.jl:100-102
// ------------------------------------------------------------------------------------------
template <class T> inline void center(const BSphere<T> &b, T c[3]) {
    c[0] = b.x[0];
    c[1] = b.x[1];
    c[2] = b.x[2];
}
template <class T> inline void center(const BBox<T> &b, T c[3]) {
    c[0] = T(0.5) * (b.lo[0] + b.up[0]);
    c[1] = T(0.5) * (b.lo[1] + b.up[1]);
    c[2] = T(0.5) * (b.lo[2] + b.up[2]);
}

// ------------------------------------------------------------------------------------------
// triangle constructors — bsphere.jl:43-112, bbox.jl:59-70
// ------------------------------------------------------------------------------------------
template <class T> inline BSphere<T> bsphere_from_triangle(const T *p1, const T *p2, const T *p3) {
    const T a[3] = {p1[0], p1[1], p1[2]}, b[3] = {p2[0], p2[1], p2[2]}, c[3] = {p3[0], p3[1], p3[2]};
    T abab = (b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1]) + (b[2] - a[2]) * (b[2] - a[2]);
    T abac = (b[0] - a[0]) * (c[0] - a[0]) + (b[1] - a[1]) * (c[1] - a[1]) + (b[2] - a[2]) * (c[2] - a[2]);
    T acac = (c[0] - a[0]) * (c[0] - a[0]) + (c[1] - a[1]) * (c[1] - a[1]) + (c[2] - a[2]) * (c[2] - a[2]);
    T d = T(2.) * (abab * acac - abac * abac); // :66
    BSphere<T> out;
    if (std::fabs(d) <= std::numeric_limits<T>::epsilon()) { // :68, eps(T)
        T lower[3] = {minimum3(a[0], b[0], c[0]), minimum3(a[1], b[1], c[1]), minimum3(a[2], b[2], c[2])};
        T upper[3] = {maximum3(a[0], b[0], c[0]), maximum3(a[1], b[1], c[1]), maximum3(a[2], b[2], c[2])};
        T centre[3] = {T(0.5) * (lower[0] + upper[0]), T(0.5) * (lower[1] + upper[1]), T(0.5) * (lower[2] + upper[2])};
        T radius = dist3(centre, upper);
        out = {{centre[0], centre[1], centre[2]}, radius};
        return out;
    }
    T s = (abab * acac - acac * abac) / d; // :83
    T t = (acac * abab - abab * abac) / d; // :84
    T centre[3];
    const T *ref;
    if (s <= T(0)) { // :86
        for (int k = 0; k < 3; ++k) centre[k] = T(0.5) * (a[k] + c[k]);
        ref = a;
    } else if (t <= T(0)) { // :92
        for (int k = 0; k < 3; ++k) centre[k] = T(0.5) * (a[k] + b[k]);
        ref = a;
    } else if (s + t >= T(1)) { // :98
        for (int k = 0; k < 3; ++k) centre[k] = T(0.5) * (b[k] + c[k]);
        ref = b;
    } else { // :104
        for (int k = 0; k < 3; ++k) centre[k] = a[k] + s * (b[k] - a[k]) + t * (c[k] - a[k]);
        ref = a;
    }
    T radius = dist3(centre, ref);
    out = {{centre[0], centre[1], centre[2]}, radius};
    return out;
}
template <class T> inline BBox<T> bbox_from_triangle(const T *p1, const T *p2, const T *p3) {
    BBox<T> o;
    for (int k = 0; k < 3; ++k) {
        o.lo[k] = minimum3(p1[k], p2[k], p3[k]);
        o.up[k] = maximum3(p1[k], p2[k], p3[k]);
    }
    return o;
}

// ------------------------------------------------------------------------------------------
// conversions and merges — merge.jl
// ------------------------------------------------------------------------------------------
// BSphere{T}(x::BSphere) / BBox{T}(x::BBox): field-wise convert (round to nearest)
template <class TN, class TL> inline BSphere<TN> convert_to(const BSphere<TL> &a, BSphere<TN> *) {
    return {{TN(a.x[0]), TN(a.x[1]), TN(a.x[2])}, TN(a.r)};
}
template <class TN, class TL> inline BBox<TN> convert_to(const BBox<TL> &a, BBox<TN> *) {
    return {{TN(a.lo[0]), TN(a.lo[1]), TN(a.lo[2])}, {TN(a.up[0]), TN(a.up[1]), TN(a.up[2])}};
}
// BBox{T}(a::BSphere) — merge.jl:47-51 (arithmetic in the sphere's type, then convert)
template <class TN, class TL> inline BBox<TN> convert_to(const BSphere<TL> &a, BBox<TN> *) {
    TL lo[3] = {a.x[0] - a.r, a.x[1] - a.r, a.x[2] - a.r};
    TL up[3] = {a.x[0] + a.r, a.x[1] + a.r, a.x[2] + a.r};
    return {{TN(lo[0]), TN(lo[1]), TN(lo[2])}, {TN(up[0]), TN(up[1]), TN(up[2])}};
}
// BSphere{T}(a::BSphere, b::BSphere) — merge.jl:2-22
template <class TN, class TL>
inline BSphere<TN> merge_to(const BSphere<TL> &a, const BSphere<TL> &b, BSphere<TN> *) {
    TL length = dist3(a.x, b.x);
    if (length + a.r <= b.r) return convert_to(b, (BSphere<TN> *)nullptr); // :6
    if (length + b.r <= a.r) return convert_to(a, (BSphere<TN> *)nullptr); // :10
    // :15-19 in Julia's promoted type TP = promote_type(TL, TN): T(0.5) and T(1) are exact in every float type, so with TL
    // at least as wide as TN all of it is TL arithmetic; with a WIDER node type the leaf-typed sub-expressions
    // ((b.r - a.r) / length, b.x - a.x, length + a.r + b.r) are still evaluated in TL and only their combination with a
    // T-typed value is promoted — which is exactly what C++'s usual arithmetic conversions do with these expressions
    using TP = decltype(TL() + TN());
    TP frac = TP(0.5) * ((b.r - a.r) / length + TP(1));
    TP centre[3] = {a.x[0] + frac * (b.x[0] - a.x[0]), a.x[1] + frac * (b.x[1] - a.x[1]),
                    a.x[2] + frac * (b.x[2] - a.x[2])};
    TP radius = TP(0.5) * (length + a.r + b.r);
    return {{TN(centre[0]), TN(centre[1]), TN(centre[2])}, TN(radius)};
}
// BBox{T}(a::BBox, b::BBox) — merge.jl:30-40
template <class TN, class TL>
inline BBox<TN> merge_to(const BBox<TL> &a, const BBox<TL> &b, BBox<TN> *) {
    BBox<TN> o;
    for (int k = 0; k < 3; ++k) {
        o.lo[k] = TN(minimum2(a.lo[k], b.lo[k]));
        o.up[k] = TN(maximum2(a.up[k], b.up[k]));
    }
    return o;
}
// BBox{T}(a::BSphere, b::BSphere) — merge.jl:58-81
template <class TN, class TL>
inline BBox<TN> merge_to(const BSphere<TL> &a, const BSphere<TL> &b, BBox<TN> *) {
    TL length = dist3(a.x, b.x);
    if (length + a.r <= b.r) return convert_to(b, (BBox<TN> *)nullptr); // :62-63
    if (length + b.r <= a.r) return convert_to(a, (BBox<TN> *)nullptr); // :66-67
    BBox<TN> o;
    for (int k = 0; k < 3; ++k) {
        o.lo[k] = TN(minimum2(a.x[k] - a.r, b.x[k] - b.r)); // :71-73
        o.up[k] = TN(maximum2(a.x[k] + a.r, b.x[k] + b.r)); // :75-77
    }
    return o;
}

// ------------------------------------------------------------------------------------------
// iscontact — iscontact.jl:2-28 (mixed float types compare after exact promotion, as in Julia)
// ------------------------------------------------------------------------------------------
template <class TA, class TB> inline bool iscontact(const BSphere<TA> &a, const BSphere<TB> &b) {
    return dist3sq(a.x, b.x) <= (a.r + b.r) * (a.r + b.r);
}
template <class TA, class TB> inline bool iscontact(const BBox<TA> &a, const BBox<TB> &b) {
    return (a.up[0] >= b.lo[0] && a.lo[0] <= b.up[0]) && (a.up[1] >= b.lo[1] && a.lo[1] <= b.up[1]) &&
           (a.up[2] >= b.lo[2] && a.lo[2] <= b.up[2]);
}
template <class TA, class TB> inline bool iscontact(const BSphere<TA> &a, const BBox<TB> &b) {
    BBox<TA> ab = {{a.x[0] - a.r, a.x[1] - a.r, a.x[2] - a.r}, {a.x[0] + a.r, a.x[1] + a.r, a.x[2] + a.r}};
    return iscontact(ab, b);
}
template <class TA, class TB> inline bool iscontact(const BBox<TA> &a, const BSphere<TB> &b) {
    return iscontact(b, a);
}

// ------------------------------------------------------------------------------------------
// isintersection — isintersection.jl:1-65
// ------------------------------------------------------------------------------------------
template <class T> inline bool isintersection(const BBox<T> &b, const T *p, const T *d) {
    T inv_d[3] = {T(1) / d[0], T(1) / d[1], T(1) / d[2]};
    T t1 = (b.lo[0] - p[0]) * inv_d[0];
    T t2 = (b.up[0] - p[0]) * inv_d[0];
    T tmin = minimum2(t1, t2);
    T tmax = maximum2(t1, t2);
    t1 = (b.lo[1] - p[1]) * inv_d[1];
    t2 = (b.up[1] - p[1]) * inv_d[1];
    tmin = maximum2(tmin, minimum2(t1, t2));
    tmax = minimum2(tmax, maximum2(t1, t2));
    t1 = (b.lo[2] - p[2]) * inv_d[2];
    t2 = (b.up[2] - p[2]) * inv_d[2];
    tmin = maximum2(tmin, minimum2(t1, t2));
    tmax = minimum2(tmax, maximum2(t1, t2));
    return (tmin <= tmax) && (tmax >= 0);
}
template <class T> inline bool isintersection(const BSphere<T> &s, const T *p, const T *d) {
    T a = dot3(d, d);
    T b = T(2) * ((p[0] - s.x[0]) * d[0] + (p[1] - s.x[1]) * d[1] + (p[2] - s.x[2]) * d[2]);
    T c = ((p[0] - s.x[0]) * (p[0] - s.x[0]) + (p[1] - s.x[1]) * (p[1] - s.x[1]) +
           (p[2] - s.x[2]) * (p[2] - s.x[2])) -
          s.r * s.r;
    T discriminant = b * b - T(4) * a * c;
    if (discriminant >= T(0)) {
        if (b <= T(0)) return true;
        return T(0) >= c;
    }
    return false;
}

// ------------------------------------------------------------------------------------------
// Morton — morton/default.jl, morton/utils.jl
// ------------------------------------------------------------------------------------------
inline uint16_t morton_split3(uint16_t v) { // default.jl:118-127
    uint16_t s = v & 0x001f;
    s = (s | uint16_t(s << 8)) & 0x100f;
    s = (s | uint16_t(s << 4)) & 0x10c3;
    s = (s | uint16_t(s << 2)) & 0x1249;
    return s;
}
inline uint32_t morton_split3(uint32_t v) { // default.jl:130-143
    uint32_t s = v & 0x000003ffu;
    s = (s | s << 16) & 0x30000ffu;
    s = (s | s << 8) & 0x0300f00fu;
    s = (s | s << 4) & 0x30c30c3u;
    s = (s | s << 2) & 0x9249249u;
    return s;
}
inline uint64_t morton_split3(uint64_t v) { // default.jl:146-157
    uint64_t s = v & 0x00000000001fffffull;
    s = (s | s << 32) & 0x1f00000000ffffull;
    s = (s | s << 16) & 0x1f0000ff0000ffull;
    s = (s | s << 8) & 0x100f00f00f00f00full;
    s = (s | s << 4) & 0x10c30c30c30c30c3ull;
    s = (s | s << 2) & 0x1249249249249249ull;
    return s;
}
template <class M> constexpr int morton_scaling_bits() { // default.jl:167-169 (2^5, 2^10, 2^21)
    return sizeof(M) == 2 ? 5 : (sizeof(M) == 4 ? 10 : 21);
}
template <class T> constexpr T relative_precision() { // default.jl:179-181
    return sizeof(T) == 4 ? T(1e-5) : T(1e-14);
}
// morton_encode_single — default.jl:91-108
template <class M, class T> inline M morton_encode_single(const T centre[3], const T mins[3], const T maxs[3]) {
    const T scaling = T(int64_t(1) << morton_scaling_bits<M>()); // Int promoted to T (:93,101)
    T s1 = (centre[0] - mins[0]) / (maxs[0] - mins[0]);
    T s2 = (centre[1] - mins[1]) / (maxs[1] - mins[1]);
    T s3 = (centre[2] - mins[2]) / (maxs[2] - mins[2]);
    // unsafe_trunc(U, x) == fptoui
    M i1 = (M)(uint64_t)(s1 * scaling);
    M i2 = (M)(uint64_t)(s2 * scaling);
    M i3 = (M)(uint64_t)(s3 * scaling);
    return M((M(morton_split3(i1) << 2)) | (M(morton_split3(i2) << 1)) | morton_split3(i3));
}
// _compute_extrema + bounding_volumes_extrema — morton/utils.jl:1-72
template <class Rec, class T> inline void extrema_of(const Rec *recs, int64_t n, bool expand, T out[6]) {
    T mn[3] = {std::numeric_limits<T>::max(), std::numeric_limits<T>::max(), std::numeric_limits<T>::max()}; // floatmax
    // NB: the max-reduce is initialised with floatmin(T), the smallest positive NORMAL number,
    // not -floatmax (utils.jl:39-40).  Load-bearing for parity.
    T mx[3] = {std::numeric_limits<T>::min(), std::numeric_limits<T>::min(), std::numeric_limits<T>::min()};
    for (int64_t i = 0; i < n; ++i) {
        T c[3];
        center(volume_of(recs[i]), c);
        for (int k = 0; k < 3; ++k) {
            mn[k] = mn[k] < c[k] ? mn[k] : c[k]; // min_centers(a, b): a[k] < b[k] ? a[k] : b[k]
            mx[k] = mx[k] > c[k] ? mx[k] : c[k];
        }
    }
    if (expand) { // utils.jl:63-69 — two roundings per side, never fused
        const T rp = relative_precision<T>();
        const T fm = std::numeric_limits<T>::min();
        for (int k = 0; k < 3; ++k) {
            T a = rp * std::fabs(mn[k]);
            mn[k] = (mn[k] - a) - fm;
            T b = rp * std::fabs(mx[k]);
            mx[k] = (mx[k] + b) + fm;
        }
    }
    for (int k = 0; k < 3; ++k) {
        out[k] = mn[k];
        out[3 + k] = mx[k];
    }
}
// volume_of: a raw volume is its own volume; a BoundingVolume record yields .volume
template <class T> inline const BSphere<T> &volume_of(const BSphere<T> &v) { return v; }
template <class T> inline const BBox<T> &volume_of(const BBox<T> &v) { return v; }
template <class V, class I, class M> inline const V &volume_of(const BoundingVolume<V, I, M> &b) { return b.volume; }

} // namespace orc
