// rm_kernels.h -- launchers of the gfx950 kernels in rm_kernels.hip.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#endif

#include "rm_types.h"

struct RmDiagDevice {  // accumulator of rm_reduce_counters_device (32 bytes)
    unsigned long long total_sdf;
    unsigned long long total_iters;
    unsigned int max_sdf;
    unsigned int min_sdf;
    unsigned long long pad;
};

// rm_kernels.hip and rm_render_v2.hip are compiled twice: as they are (vec3.length = Math.hypot, the gl-matrix 3.0 -
// 3.4.3 form) and with -DRM_LENGTH_SQRT (vec3.length = Math.sqrt(x*x + y*y + z*z); entry points carry the suffix
// _sqrt).  Option `length` picks the set (rm_device.h, vec3_length).
#ifdef RM_LENGTH_SQRT
#define RM_LEN_VARIANT(name) name##_sqrt
#define RM_LEN_TAG " [length=sqrt]"
#define RM_LEN_IS_SQRT true
#else
#define RM_LEN_VARIANT(name) name
#define RM_LEN_TAG ""
#define RM_LEN_IS_SQRT false
#endif

#ifndef __HIPCC_RTC__  // (host side: the launchers)
// Renders rows [y_start, y_end) (runRaymarcher + optional fused shade).  *kernel_name (optional) receives the
// instantiation that was launched (static string).
hipError_t rm_launch_render(const RmRenderParams &p, hipStream_t stream, const char **kernel_name);
hipError_t rm_launch_render_sqrt(const RmRenderParams &p, hipStream_t stream, const char **kernel_name);

// Asked by rm_launch_render_v2 for every launch that carries a context (RmRenderParams::rtc_ctx): the hipFunction_t of the
// wave loop compiled for this launch's configuration (rm_v2_fields.h), or null -- the library's own instantiation runs.
// Implemented in rm_api.cpp (it counts how often a configuration comes back and owns the compile policy).
const void *rm_rtc_v2_hook(const RmRenderParams &p, int accel, bool lds, bool ur, bool rel, bool length_sqrt);

// v2 kernel (rm_render_v2.hip); called by rm_launch_render when p.variant == 2
hipError_t rm_launch_render_v2(const RmRenderParams &p, hipStream_t stream, const char **kernel_name);
hipError_t rm_launch_render_v2_sqrt(const RmRenderParams &p, hipStream_t stream, const char **kernel_name);

// the octree's node boxes relative to one camera position, for render_kernel_oct (rm_kernels.hip)
hipError_t rm_launch_oct_frame_table(const RmOctNode *nodes, int n, const double origin[3], RmOctFrameNode *out, hipStream_t stream);

// ShadingModel.shade over n = width * height pixels.
hipError_t rm_launch_shade(int shader, int64_t n, const uint8_t *depth, const uint8_t *normal,
                           const uint16_t *sdf, const uint16_t *iters, uint8_t *rgba,
                           const float light[3], hipStream_t stream);

// sums 0, max 0, min UINT_MAX
hipError_t rm_launch_reduce_init(RmDiagDevice *acc, hipStream_t stream);

// diagnostics reduction into *acc (must be initialised: sums 0, max 0, min UINT_MAX)
hipError_t rm_launch_reduce(const uint16_t *sdf, const uint16_t *iters, int64_t n, RmDiagDevice *acc,
                            hipStream_t stream);

// Scene.getDistance for a batch of points
hipError_t rm_launch_distance(const RmRenderParams &p, const float *points, int64_t n, double *dist,
                              uint32_t *count, hipStream_t stream);
hipError_t rm_launch_distance_sqrt(const RmRenderParams &p, const float *points, int64_t n, double *dist,
                                   uint32_t *count, hipStream_t stream);

// v2: builds the longest-first item order of the next launch from the previous launch's recorded costs (rm_render_v2.hip)
hipError_t rm_launch_lpt_sort(const uint8_t *cost_prev, uint16_t *perm, int stride, int tiles_x, int tiles_y, hipStream_t stream);

// Rank 0 of a sharded frame (rm_frame_ops.hip): copies every stripe of a gathered [world x rank_stride] buffer to its
// place in the row-major frame and combines the ranks' partial diagnostics accumulators.
hipError_t rm_launch_assemble(const unsigned char *gathered, int64_t rank_stride, int64_t section_offset, int32_t row_bytes,
                              int32_t height, int32_t stripe_rows, const int32_t *stripe_src, int32_t n_stripes,
                              unsigned char *frame, int64_t acc_offset, int32_t world, RmDiagDevice *acc, hipStream_t stream);

hipError_t rm_launch_hypot(const float *xyz, int64_t n, double *out, hipStream_t stream);
// rm_jsmath.h on the device: fn 0 sin, 1 cos, 2 atan2, 3 asin, 4 log, 5 pow, 6 round, 7 atan
hipError_t rm_launch_jsmath(int fn, const double *a, const double *b, int64_t n, double *out, hipStream_t stream);

// compares the two device forms of Math.hypot on n generated triples; adds mismatches
hipError_t rm_launch_fastdiv_selftest(uint64_t seed, int64_t n, unsigned long long *d_mismatches, hipStream_t stream);
hipError_t rm_launch_recip_selftest(int mode, unsigned long long *d_mismatches, hipStream_t stream);
#endif  // !__HIPCC_RTC__
