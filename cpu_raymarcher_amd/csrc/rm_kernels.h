// rm_kernels.h -- launchers of the gfx950 kernels in rm_kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "rm_types.h"

struct RmDiagDevice {  // accumulator of rm_reduce_counters_device (32 bytes)
    unsigned long long total_sdf;
    unsigned long long total_iters;
    unsigned int max_sdf;
    unsigned int min_sdf;
    unsigned long long pad;
};

// Renders rows [y_start, y_end) (runRaymarcher + optional fused shade).
hipError_t rm_launch_render(const RmRenderParams &p, hipStream_t stream);

// v2 kernel (rm_render_v2.hip); called by rm_launch_render when p.variant == 2
hipError_t rm_launch_render_v2(const RmRenderParams &p, hipStream_t stream);

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

hipError_t rm_launch_hypot(const float *xyz, int64_t n, double *out, hipStream_t stream);
// rm_jsmath.h on the device: fn 0 sin, 1 cos, 2 atan2, 3 asin, 4 log, 5 pow, 6 round, 7 atan
hipError_t rm_launch_jsmath(int fn, const double *a, const double *b, int64_t n, double *out, hipStream_t stream);

// compares the two device forms of Math.hypot on n generated triples; adds mismatches
hipError_t rm_launch_fastdiv_selftest(uint64_t seed, int64_t n, unsigned long long *d_mismatches, hipStream_t stream);
hipError_t rm_launch_recip_selftest(int mode, unsigned long long *d_mismatches, hipStream_t stream);
