// ibvh_misc.hip — input preparation adjacent to the hot path: triangle -> bounding volume
// (reference src/bounding_volumes/bsphere.jl:43-112, bbox.jl:59-70; README "Compute bounding
// volumes") and the deterministic synthetic-input generator shared with the oracle by
// specification (DESIGN.md §Synthetic inputs).
#include "ibvh_common.hpp"

namespace ibvh {
namespace misc {

template <class V>
__global__ __launch_bounds__(256) void tri_kernel(const typename V::elt *__restrict__ tris, int64_t n, V *__restrict__ out) {
    using T = typename V::elt;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        T a[3], b[3], c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            a[k] = tris[9 * i + k];
            b[k] = tris[9 * i + 3 + k];
            c[k] = tris[9 * i + 6 + k];
        }
        if constexpr (V::kind == IBVH_BSPHERE) out[i] = bsphere_from_triangle(a, b, c);
        else out[i] = bbox_from_triangle(a, b, c);
    }
}

// SplitMix64 of (seed, counter): z = seed + (ctr+1)*0x9E3779B97F4A7C15, then the standard finaliser;
// u = (z >> 40) * 2^-24 in [0, 1).
IBVH_HD uint64_t splitmix64(uint64_t seed, uint64_t ctr) {
    uint64_t z = seed + (ctr + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
IBVH_HD float u01(uint64_t seed, uint64_t ctr) { return float(splitmix64(seed, ctr) >> 40) * (1.0f / 16777216.0f); }

struct F3 {
    float v[3];
};
__global__ __launch_bounds__(256) void gen_spheres_kernel(int64_t n, uint64_t seed, int64_t first, F3 origin, F3 extent,
                                                          float r0, BSphere<float> *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint64_t g = (uint64_t)(first + i);
        float u0 = u01(seed, 4 * g + 0), u1 = u01(seed, 4 * g + 1), u2 = u01(seed, 4 * g + 2), u3 = u01(seed, 4 * g + 3);
        BSphere<float> s;
        s.x[0] = origin.v[0] + extent.v[0] * u0;
        s.x[1] = origin.v[1] + extent.v[1] * u1;
        s.x[2] = origin.v[2] + extent.v[2] * u2;
        s.r = r0 * (0.5f + 0.5f * u3);
        out[i] = s;
    }
}

} // namespace misc
} // namespace ibvh

using namespace ibvh;

extern "C" {

ibvh_status ibvh_volumes_from_triangles(int32_t kind, int32_t flt, const void *triangles, int64_t n, void *volumes_out,
                                        void *stream) {
    if (n < 0 || (n > 0 && (!triangles || !volumes_out))) return IBVH_ERR_INVALID_ARG;
    if (n == 0) return IBVH_OK;
    int64_t b = ceil_div(n, 256);
    unsigned blocks = (unsigned)(b > 8192 ? 8192 : b);
    return (ibvh_status)dispatch_volume(kind, flt, [&](auto vt) -> int {
        using V = typename decltype(vt)::type;
        IBVH_LAUNCH((misc::tri_kernel<V>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const typename V::elt *)triangles, n, (V *)volumes_out);
        IBVH_LAUNCH_CHECK();
        return IBVH_OK;
    });
}

ibvh_status ibvh_generate_spheres_f32(int64_t n, uint64_t seed, int64_t first_index, const float origin[3],
                                      const float extent[3], float r0, void *volumes_out, void *stream) {
    if (n < 0 || !origin || !extent || (n > 0 && !volumes_out)) return IBVH_ERR_INVALID_ARG;
    if (n == 0) return IBVH_OK;
    int64_t b = ceil_div(n, 256);
    unsigned blocks = (unsigned)(b > 8192 ? 8192 : b);
    misc::F3 o{{origin[0], origin[1], origin[2]}}, e{{extent[0], extent[1], extent[2]}};
    IBVH_LAUNCH(misc::gen_spheres_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, seed, first_index, o, e,
                       r0, (BSphere<float> *)volumes_out);
    return hipGetLastError() == hipSuccess ? IBVH_OK : IBVH_ERR_HIP;
}

} // extern "C"
