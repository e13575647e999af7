// ibvh_msd_finish_geometries.hpp — (key type, threads, keys per thread) of every finish workgroup make_plan() can choose
// (ibvh_msd.hip) or a development knob can force, split over the three translation units that compile them.
#pragma once
#define IBVH_FINISH_GEOMETRIES_A IBVH_FIN(uint32_t, 256, 6) IBVH_FIN(uint32_t, 256, 10) IBVH_FIN(uint32_t, 256, 11) IBVH_FIN(uint32_t, 256, 12) IBVH_FIN(uint32_t, 256, 8) IBVH_FIN(uint32_t, 256, 16) IBVH_FIN(uint32_t, 256, 32)
#define IBVH_FINISH_GEOMETRIES_B IBVH_FIN(uint32_t, 512, 8) IBVH_FIN(uint32_t, 512, 16) IBVH_FIN(uint32_t, 512, 32) IBVH_FIN(uint32_t, 1024, 8) IBVH_FIN(uint32_t, 1024, 16) IBVH_FIN(uint32_t, 1024, 3) IBVH_FIN(uint32_t, 1024, 4)
#define IBVH_FINISH_GEOMETRIES_C IBVH_FIN(uint32_t, 512, 6) IBVH_FIN(uint32_t, 1024, 6) IBVH_FIN(uint32_t, 512, 11) IBVH_FIN(uint32_t, 512, 12) IBVH_FIN(uint32_t, 512, 5) IBVH_FIN(uint32_t, 512, 7) IBVH_FIN(uint64_t, 256, 8) IBVH_FIN(uint64_t, 256, 16) IBVH_FIN(uint64_t, 256, 32) IBVH_FIN(uint64_t, 512, 8) IBVH_FIN(uint64_t, 512, 16) IBVH_FIN(uint64_t, 1024, 8)
