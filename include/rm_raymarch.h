/*
 * rm_raymarch.h -- C ABI of the MI355X-native sphere-tracing render path.
 *
 * Drop-in boundary for ONE path of vxlerian/cpu-raymarcher: the per-pixel render path
 *   worker tile dispatch -> Raymarcher.runRaymarcher -> SphereTracer.rayMarch
 *   -> BVH / Octree callbacks -> Scene.getDistance -> Sphere.sdf, then ShadingModel.shade
 * (reference files cited per entry point, paths relative to the reference's src/).
 *
 * Plain pointers and sizes only; no C++/torch types.  Every function returns an
 * rm_status (0 = ok, negative = error) unless stated; nothing throws or aborts across
 * the boundary (the reference has no error path on this route: bad inputs are clamped
 * or defaulted exactly as the reference does, see each entry).
 *
 * Buffers follow the reference's typed arrays (raymarchWorker.ts:42-46):
 *   depth  : uint8  [W*h]      Uint8ClampedArray
 *   normal : uint8  [W*h*3]    Uint8ClampedArray, RGB interleaved
 *   sdf    : uint16 [W*h]      Uint16Array  (SDF evaluations per pixel, wraps mod 65536)
 *   iters  : uint16 [W*h]      Uint16Array  (march iterations per pixel)
 *   rgba   : uint8  [W*h*4]    Uint8ClampedArray (ImageData)
 * all tile-local, row-major, h = max(0, yEnd - yStart).  The caller owns every buffer;
 * the library never frees or retains one (the reference transfers ownership back to the
 * main thread, raymarchWorker.ts:86-91).
 *
 * Threading: an rm_ctx is bound to one GPU and is not thread-safe; use one ctx per host
 * thread / GPU (the reference keeps one job in flight per worker, main.ts:447-490).
 */
#ifndef RM_RAYMARCH_H
#define RM_RAYMARCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RM_API __attribute__((visibility("default")))

typedef struct rm_ctx rm_ctx;

typedef enum rm_status {
    RM_OK = 0,
    RM_E_INVALID = -1,     /* null pointer, negative size, non-finite camera ...            */
    RM_E_UNSUPPORTED = -2, /* valid request the native path cannot serve (operator tree     */
                           /* nested deeper than 15, BVH leaf above 255 primitives): the    */
                           /* host keeps its own CPU path for that job                      */
    RM_E_NO_DEVICE = -3,   /* ctx was created host-only, or no HIP device                   */
    RM_E_HIP = -4,         /* a HIP runtime call failed; rm_last_error has the text         */
    RM_E_NO_SCENE = -5,    /* render requested before any scene was set                     */
    RM_E_NOMEM = -6
} rm_status;

/* Scene.accelerationStructure strings "None" | "Octree" | "BVH" (scene.ts:21,32-36) */
typedef enum rm_accel { RM_ACCEL_NONE = 0, RM_ACCEL_OCTREE = 1, RM_ACCEL_BVH = 2 } rm_accel;

/* Job.algorithm strings (raymarchWorker.ts:50-68) */
typedef enum rm_algorithm {
    RM_ALG_SPHERE_TRACER = 0, /* 'sphere-tracer' and every unknown string (default branch) */
    RM_ALG_FIXED_STEP = 1,
    RM_ALG_ADAPTIVE_STEP = 2,
    RM_ALG_ADAPTIVE_STEP_V2 = 3,
    RM_ALG_ADAPTIVE_STEP_V3 = 4
} rm_algorithm;

/* shading model strings (main.ts:33-45) */
typedef enum rm_shader {
    RM_SHADE_NORMAL = 0, /* 'normal' and every unknown string */
    RM_SHADE_PHONG = 1,
    RM_SHADE_SDF_HEATMAP = 2,
    RM_SHADE_ITERATION_HEATMAP = 3
} rm_shader;

/* scene_preset_index value that selects the scene last given to rm_scene_from_spheres */
#define RM_SCENE_UPLOADED INT32_MIN

/* Replaces the worker message `Job` (raymarchWorker.ts:10-22).  The reference worker
 * rebuilds the scene from (scenePresetIndex, accelerationStructure) for every job
 * (raymarchWorker.ts:37-38); here the ctx caches the built, device-resident scene and
 * rebuilds only when those two fields change. */
typedef struct rm_job {
    int32_t width;                  /* Job.width  : full-frame width                     */
    int32_t height;                 /* Job.height : full-frame height                    */
    double  time;                   /* Job.time   : unused by sphere primitives          */
    int32_t y_start;                /* Job.yStart                                        */
    int32_t y_end;                  /* Job.yEnd   (rows [y_start, y_end))                */
    double  camera_pitch;           /* Job.camera.pitch (clamped to +-pi/2, camera.ts:59) */
    double  camera_yaw;             /* Job.camera.yaw                                    */
    int32_t algorithm;              /* rm_algorithm                                      */
    int32_t scene_preset_index;     /* clamped to [0, 18] (scene.ts:39) or RM_SCENE_UPLOADED */
    int32_t acceleration_structure; /* rm_accel                                          */
    int32_t reserved;
    double  overshoot_factor;       /* Job.overshootFactor, AdaptiveStepV2/V3; NaN = JS undefined -> 1.2 */
    double  step_size;              /* Job.stepSize, FixedStep;               NaN = JS undefined -> 0.1 */
} rm_job;

/* what rm_scene_get_info reports about the built acceleration structure */
typedef struct rm_scene_info {
    int32_t n_prims;
    int32_t accel;           /* rm_accel */
    int32_t preset_index;    /* or RM_SCENE_UPLOADED */
    int32_t bvh_nodes, bvh_leaves, bvh_depth;
    int32_t oct_nodes, oct_leaves, oct_empty_leaves, oct_max_leaf_prims;
    float   root_min[3], root_max[3];
    int32_t nodes_in_lds;    /* 1 when the flattened node table is staged in LDS */
    int32_t reserved;
} rm_scene_info;

/* diagnostics of main.ts:528-548 */
typedef struct rm_diagnostics {
    uint64_t total_sdf_calls;
    uint64_t total_iterations;
    uint32_t max_sdf_calls;
    uint32_t min_sdf_calls;  /* Number.MAX_SAFE_INTEGER in the reference when n == 0; UINT32_MAX here */
    uint64_t total_pixels;
} rm_diagnostics;

/* ---- context ------------------------------------------------------------------- */

/* device >= 0: bind to that HIP device.  device == -1: host-only ctx (scene building and
 * camera only; every render entry returns RM_E_NO_DEVICE).  There is no CPU fallback. */
RM_API int rm_create(int device, rm_ctx **out);
RM_API void rm_destroy(rm_ctx *ctx);
RM_API const char *rm_last_error(const rm_ctx *ctx); /* never NULL */
/* the render-kernel instantiation the most recent render entry launched ("" before the first; never NULL):
 * measurement aid, so that a benchmark line names the kernel that really ran */
RM_API const char *rm_last_kernel(const rm_ctx *ctx);
RM_API const char *rm_version(void);

/* string -> enum with the reference's defaulting rules (never fail) */
RM_API int rm_algorithm_from_string(const char *s); /* raymarchWorker.ts:50-68 */
RM_API int rm_accel_from_string(const char *s);     /* scene.ts:32-36          */
RM_API int rm_shader_from_string(const char *s);    /* main.ts:33-45           */
RM_API int rm_preset_count(void);                   /* sceneManager.ts:363-365 */

/* ---- scene --------------------------------------------------------------------- */

/* Replaces `new Scene(accel); scene.loadPreset(index)` (scene.ts:24-59,
 * raymarchWorker.ts:37-38): builds the primitive list, the BVH (bvh.ts:29-92) or Octree
 * (octree.ts:36-191), flattens it and uploads it.  All 19 presets are native: 0..4 (spheres),
 * 5, 7, 8, 9 (torus, boxes), 6 and 10..18 (SDF operators, Mandelbulb; index clamps like
 * sceneManager.ts:359-361). */
RM_API int rm_scene_from_preset(rm_ctx *ctx, int32_t preset_index, int32_t accel);

/* Build-defined scene entry: n spheres as SceneManager.createSphere(x, y, z, r) without
 * rotation would make them (sceneManager.ts:21-41): centre f32, radius a double
 * (sphere.ts:5-9).  Selected in jobs by scene_preset_index = RM_SCENE_UPLOADED. */
RM_API int rm_scene_from_spheres(rm_ctx *ctx, const float *centers_xyz, const double *radii,
                                 int32_t n, int32_t accel);

/* General primitives: Sphere / Box / Torus with any world->local matrix
 * (primitives/sphere.ts, box.ts, torus.ts; Primitive.sdf primitive.ts:33-39). */
typedef enum rm_prim_type { RM_PRIM_SPHERE = 0, RM_PRIM_BOX = 1, RM_PRIM_TORUS = 2 } rm_prim_type;
typedef struct rm_prim {
    int32_t type;               /* rm_prim_type */
    int32_t reserved;
    float   world_to_local[16]; /* Primitive.transform, gl-matrix layout (column-major) */
    double  params[3];          /* sphere: radius; box: halfSize x,y,z (stored f32, box.ts:10); */
                                /* torus: majorRadius, minorRadius                               */
    double  reserved2;
} rm_prim;

/* n primitives of any of the three kinds, selected in jobs by RM_SCENE_UPLOADED. */
RM_API int rm_scene_from_prims(rm_ctx *ctx, const rm_prim *prims, int32_t n, int32_t accel);

/* SceneManager.getTransform(x, y, z, rotation?) (sceneManager.ts:21-37): the world->local
 * matrix of a primitive; rotation_xyz may be NULL (no rotation argument) or three Euler
 * angles stored as binary32 like a gl-matrix vec3. */
RM_API int rm_make_transform(double x, double y, double z, const float *rotation_xyz, float *world_to_local16);

/* SDF expression forests: the operator classes of util/primitive_operations/
 * (round.ts, smoothUnion.ts, smoothSubstraction.ts, twist.ts, repetition.ts, animatedTranslate.ts)
 * over Sphere / Box / Torus / Mandelbulb (primitives/mandelbulb.ts) leaves.  All 19 presets of
 * sceneManager.ts:102-357 are served natively through rm_scene_from_preset / rm_job; this entry
 * takes an arbitrary forest.  Operands must precede the node that uses them. */
typedef enum rm_node_type {
    RM_NODE_SPHERE = 0, RM_NODE_BOX = 1, RM_NODE_TORUS = 2, RM_NODE_MANDELBULB = 3,
    RM_NODE_ROUND = 10, RM_NODE_SMOOTH_UNION = 11, RM_NODE_SMOOTH_SUBTRACTION = 12,
    RM_NODE_TWIST = 13, RM_NODE_REPETITION = 14, RM_NODE_ANIMATED_TRANSLATE = 15
} rm_node_type;
typedef struct rm_node {
    int32_t type;               /* rm_node_type */
    int32_t child_a, child_b;   /* operand node indices (< own index); -1 when unused */
    int32_t reserved;
    float   world_to_local[16]; /* leaves: Primitive.transform.  Operators derive theirs as their
                                 * constructors do (wrappers: the operand's; unions: identity) */
    double  params[6];          /* sphere: radius | box: halfSize | torus: major, minor |
                                 * mandelbulb: power, iterations, enableAnimation, animationSpeed |
                                 * round: radius | unions: smoothness | twist: twistAmount |
                                 * repetition: spacing x,y,z | animated translate: the NORMALISED
                                 * direction x,y,z (animatedTranslate.ts:22-23), amplitude, speed */
} rm_node;

/* roots[] lists the nodes that are Scene.objects, in order; selected in jobs by RM_SCENE_UPLOADED.
 * RM_E_UNSUPPORTED when a tree is deeper than the device interpreter's 15 nested operators. */
RM_API int rm_scene_from_nodes(rm_ctx *ctx, const rm_node *nodes, int32_t n_nodes, const int32_t *roots,
                               int32_t n_roots, int32_t accel);

/* gl-matrix mat4.scale(m, m, [x, y, z]) in place: SceneManager.createMandelbulb post-scales the
 * world->local matrix by 0.5 (sceneManager.ts:63). */
RM_API int rm_scale_transform(float *world_to_local16, double x, double y, double z);

/* Scene.updateTime(time) (scene.ts:135-140) for rm_scene_distance; renders take rm_job.time. */
RM_API int rm_scene_set_time(rm_ctx *ctx, double time);

RM_API int rm_scene_get_info(const rm_ctx *ctx, rm_scene_info *out);

/* Camera.setAngles + getRotationMatrix/getPosition (camera.ts:38-44,58-69,81-88):
 * writes mat3.fromMat4 of the rotation (9 floats, column-major) and the origin. */
RM_API int rm_camera_from_angles(double pitch, double yaw, float *rot9, float *origin3);

/* Scene.getDistance(position, counter) (scene.ts:144-190) for a batch of points, on the
 * device: dist[i] and count[i] (primitives evaluated) for points xyz f32[3n] (host). */
RM_API int rm_scene_distance(rm_ctx *ctx, const float *points_xyz, int64_t n, double *dist,
                             uint32_t *count);

/* ---- render -------------------------------------------------------------------- */

/* Replaces the worker's onmessage (raymarchWorker.ts:33-92) = Raymarcher.runRaymarcher
 * (raymarcher.ts:46-109) for one row tile, host buffers (synchronous).  Counters are
 * (re)initialised per pixel (raymarcher.ts:79-80); buffers need no pre-clear. */
RM_API int rm_render_tile(rm_ctx *ctx, const rm_job *job, uint8_t *depth, uint8_t *normal,
                          uint16_t *sdf, uint16_t *iters);

/* Same with device pointers (hipMalloc / torch CUDA tensors), asynchronous on `stream`
 * (a hipStream_t passed as void*, NULL = default stream).  rgba may be NULL; when not,
 * ShadingModel.shade (shading_models/, all four) is fused behind the march.  depth, normal, sdf, iters may
 * each be NULL when the caller does not want that G-buffer (rgba then still sees them). */
RM_API int rm_render_tile_device(rm_ctx *ctx, const rm_job *job, int32_t shader, void *d_depth,
                                 void *d_normal, void *d_sdf, void *d_iters, void *d_rgba,
                                 void *stream);

/* Multi-GPU sharding of one Job (replaces the contiguous ceil(H/N) split of main.ts:444-449
 * by a load-balanced one): the rows [y_start, y_end) are cut into stripes of `stripe_rows`
 * rows dealt round-robin over `n_parts`; this call renders, in ONE launch, the stripes of
 * `part`, packed in increasing y (rm_stripe_rows rows of `width` pixels).  Pixels are pure
 * functions of (x, y, W, H, camera, scene) (raymarcher.ts:72-76,83), so any row subset is exact. */
RM_API int rm_render_stripes_device(rm_ctx *ctx, const rm_job *job, int32_t shader, int32_t stripe_rows,
                                    int32_t n_parts, int32_t part, void *d_depth, void *d_normal,
                                    void *d_sdf, void *d_iters, void *d_rgba, void *stream);
/* number of rows part `part` owns (>= 0), or RM_E_INVALID */
RM_API int rm_stripe_rows(int32_t y_start, int32_t y_end, int32_t stripe_rows, int32_t n_parts, int32_t part);

/* Weighted deal of the stripes of `rows` rows over n_parts parts (replaces the equal ceil(H/N) shares of
 * main.ts:444-449 when the parts are NOT equally loaded: the root of the gather also reassembles the frame, so it
 * gets a smaller share).  owner[s] receives the part of stripe s (s < ceil(rows / stripe_rows) = the return value);
 * weights[p] > 0 are relative shares (NULL: equal).  Smooth weighted round-robin: every part's stripes stay spread
 * over the whole frame (load balance: the top and bottom of a frame are mostly root-box misses), and equal weights
 * give exactly the round-robin deal of rm_render_stripes_device.  Deterministic integer arithmetic: every rank
 * computes the same deal from the same weights.  Host-side, no ctx. */
RM_API int rm_deal_stripes(int32_t rows, int32_t stripe_rows, int32_t n_parts, const int32_t *weights, int32_t *owner);

/* One launch for an explicit list of stripes: stripe_ids[0 .. n_stripes) (strictly increasing, host memory, copied),
 * stripe k of the list being rows [y_start + id * stripe_rows, ...) of the Job, packed in list order.  Buffers must
 * hold the rows the list covers (the frame's last stripe may be partial). */
RM_API int rm_render_stripe_list_device(rm_ctx *ctx, const rm_job *job, int32_t shader, int32_t stripe_rows,
                                        const int32_t *stripe_ids, int32_t n_stripes, void *d_depth, void *d_normal,
                                        void *d_sdf, void *d_iters, void *d_rgba, void *stream);

/* Rank 0's fan-in (replaces `buffer.set(tile, yStart * width)` per worker result, main.ts:461-468, and the combined
 * diagnostics of main.ts:528-548) as ONE kernel: d_gathered holds `world` per-rank packed buffers rank_stride bytes
 * apart (what a gather delivers); the section at section_offset of each holds that rank's stripes, packed in
 * increasing y, rows of row_bytes = width * bytes-per-pixel.  owner[s] (host, n_stripes = ceil(height / stripe_rows)
 * entries, as rm_deal_stripes returns them) names the rank of frame stripe s.  Writes the row-major frame to d_frame.
 * acc_offset >= 0: each rank's 32-byte partial diagnostics accumulator (rm_reduce_counters_enqueue layout) sits at
 * that offset of its packed buffer; their combination is written to d_acc (acc_offset < 0 or d_acc NULL: skipped). */
RM_API int rm_assemble_frame_device(rm_ctx *ctx, const void *d_gathered, int64_t rank_stride, int64_t section_offset,
                                    int32_t row_bytes, int32_t height, int32_t stripe_rows, const int32_t *owner,
                                    int32_t n_stripes, int32_t world, void *d_frame, int64_t acc_offset, void *d_acc,
                                    void *stream);

/* Replaces ShadingModel.shade(shaded, depth, normal, sdfEval, iters, width, height)
 * (shadingModel.ts:8-17 and the four models), host buffers. */
RM_API int rm_shade(rm_ctx *ctx, int32_t shader, int32_t width, int32_t height,
                    const uint8_t *depth, const uint8_t *normal, const uint16_t *sdf,
                    const uint16_t *iters, uint8_t *rgba);
RM_API int rm_shade_device(rm_ctx *ctx, int32_t shader, int32_t width, int32_t height,
                           const void *d_depth, const void *d_normal, const void *d_sdf,
                           const void *d_iters, void *d_rgba, void *stream);

/* Replaces the diagnostics pass of main.ts:528-548. */
RM_API int rm_reduce_counters(rm_ctx *ctx, const uint16_t *sdf, const uint16_t *iters, int64_t n,
                              rm_diagnostics *out);
/* device buffers; result written to host `out` after a stream sync */
RM_API int rm_reduce_counters_device(rm_ctx *ctx, const void *d_sdf, const void *d_iters, int64_t n,
                                     rm_diagnostics *out, void *stream);

/* Asynchronous form for a frame loop: enqueues init + reduction on `stream` and leaves the
 * result in device memory `d_acc` (32 bytes: u64 total_sdf, u64 total_iters, u32 max_sdf,
 * u32 min_sdf, u64 pad); no host synchronisation. */
RM_API int rm_reduce_counters_enqueue(rm_ctx *ctx, const void *d_sdf, const void *d_iters, int64_t n,
                                      void *d_acc, void *stream);

/* The same diagnostics WITHOUT a second pass over the counters: the next render call on this context
 * (rm_render_tile_device, rm_render_stripes_device or rm_render_stripe_list_device) also leaves, in device memory
 * `d_acc` (32 bytes, 8-byte aligned, the layout of rm_reduce_counters_enqueue), the sum / max / min of the sdfEval
 * values and the sum of the iteration values of exactly the pixels it renders -- as stored, Uint16Array wrap included
 * (raymarcher.ts:79-80,119; main.ts:534-543).  The render kernel accumulates them from the registers it stores the
 * counters from and its last wave writes the result: no initialisation of d_acc, no further launch, and the counter
 * buffers themselves may be NULL.  A call that renders no pixel writes the neutral elements (sums 0, max 0, min
 * UINT32_MAX).  One-shot: consumed by the next render call whether it succeeds or not; d_acc = NULL cancels. */
RM_API int rm_render_attach_diagnostics(rm_ctx *ctx, void *d_acc);

/* ---- tile partition (main.ts:444-450) ------------------------------------------- */

/* rows of worker i of n: [min(i*r, H), min((i+1)*r, H)) with r = ceil(H / n) */
RM_API int rm_partition_rows(int32_t height, int32_t n_workers, int32_t i, int32_t *y_start,
                             int32_t *y_end);

/* ---- device numerics self-test (parity aid, not part of the reference surface) ---- */

/* V8 Math.hypot of n float triples evaluated by the device code path used in Sphere.sdf */
RM_API int rm_selftest_hypot(rm_ctx *ctx, const float *xyz, int64_t n, double *out);

/* JS Math.* as the device computes them (csrc/rm_jsmath.h, fdlibm restated like V8's ieee754.cc):
 * fn 0 sin, 1 cos, 2 atan2(a, b), 3 asin, 4 log, 5 pow(a, b), 6 round, 7 atan; b may be NULL. */
RM_API int rm_selftest_jsmath(rm_ctx *ctx, int32_t fn, const double *a, const double *b, int64_t n, double *out);

/* Device check of the shared-reciprocal division used by the v2 kernel's sphere SDF: evaluates
 * Math.hypot with the compiler's IEEE divisions and with the shared reciprocal on n generated
 * binary32 triples (zeros, denormals, equal magnitudes included) and counts bitwise mismatches. */
RM_API int rm_selftest_fastdiv(rm_ctx *ctx, uint64_t seed, int64_t n, uint64_t *mismatches);

/* Device self-test of the range-restricted division of ray set-up (rm_device.h div_in_range) against the compiler's IEEE
 * division, EXHAUSTIVE over its domain: mode 0 = 1.0 / d for every finite non-zero binary32 d; mode 1 = x / W for all
 * integers 0 <= x < 65536, 1 <= W < 65536.  Counts bitwise mismatches (2^32 cases per mode, about a second). */
RM_API int rm_selftest_recip(rm_ctx *ctx, int mode, uint64_t *mismatches);

/* Diagnostic builds only (make EXTRA=-DRM_STAMPS): reads and clears eight per-section cycle
 * accumulators of the v2 wave loop (all zero in the product build). */
RM_API int rm_debug_read_stamps(rm_ctx *ctx, uint64_t *out8);

/* Run-time specialisation of expression forests (option `specialise`, default 1).  The reference evaluates an operator tree
 * by virtual dispatch (src/util/primitives/primitive.ts:33-39 and the overrides in src/util/primitive_operations/ *.ts); here the
 * active scene's trees are emitted as straight-line HIP and compiled for gfx950 with hiprtc into the one-ray-per-lane kernels,
 * once per (acceleration structure, marcher family) the scene is rendered with, at the first such render (1.5 - 3 s,
 * synchronous); loaded kernels are cached for the life of the process (up to 256, keyed by the generated source), so a scene
 * that comes back -- a preset menu -- does not compile again.  Without libhiprtc.so, for forests above 32 objects / 512 instructions, or with `specialise` = 0 the device
 * interpreter serves the scene (same results: both call the same formula functions in the same order).
 *   rm_rtc_source         the generated source of the active scene (NUL-terminated, truncated to cap; *needed = full size)
 *   rm_rtc_compile_check  compiles it for (accel, other != 0: the marchers other than the sphere tracer) without loading the
 *                         result -- works on a host-only context; log receives the compiler's resource-usage remarks
 *   rm_rtc_status         kernels compiled / failed for the active scene and the most recent compile log (or why hiprtc is absent) */
RM_API int rm_rtc_source(rm_ctx *ctx, char *out, int64_t cap, int64_t *needed);
RM_API int rm_rtc_compile_check(rm_ctx *ctx, int32_t accel, int32_t other, char *log, int64_t cap, double *seconds);
RM_API int rm_rtc_status(rm_ctx *ctx, int32_t *compiled, int32_t *failed, char *log, int64_t cap);

/* Diagnostic builds only (make EXTRA=-DRM_COUNTS): reads and clears the execution counts of sixteen events of the
 * v2 wave loop (scripts/counts.py) or of the v1 octree kernels (scripts/counts_v1.py) -- out32[i] wave-level executions, out32[i + 16] lanes active in them (all zero in the product build). */
RM_API int rm_debug_read_counts(rm_ctx *ctx, uint64_t *out32);
/* The item durations (units of 2.56 us, one byte per work item: 64 queues x 4096 slots) the last v2 launch recorded for the
 * longest-first order of the next one (option lpt); scripts/lpt_costs.py. */
RM_API int rm_debug_read_lpt_costs(rm_ctx *ctx, uint8_t *out, int64_t n);
/* Diagnostic builds only (make EXTRA=-DRM_STAMPS): start and end time (100 MHz ticks) of the first 8192 waves of the last
 * v2 launch -- out[w] start of the wave loop, out[8192 + w] end, out[16384 + w] kernel entry, 0 where no wave ran
 * (scripts/tail_hist.py). */
RM_API int rm_debug_read_wave_times(rm_ctx *ctx, uint64_t *out24576);
/* Diagnostic builds only (EXTRA="-DRM_STAMPS -DRM_STAMPS_LOG"): low 32 bits of the 100 MHz clock at the start of the first 96
 * batches of the first 2048 waves of the last v2 launches, out[wave * 96 + k]; 0 = no such batch (scripts/batch_timeline.py). */
RM_API int rm_debug_read_batch_log(rm_ctx *ctx, uint32_t *out196608);

/* Kernel-variant knobs for measurement; unknown keys or values are RM_E_INVALID.  They NEVER change results
 * (tests/test_gpu_parity.py renders every combination and compares the bytes).
 *   kernel 0 auto | 1 one ray per lane (v1) | 2 uniform wave loop (v2)      tile_w 8|16|32|64 pixels per wave row
 *   filter 0|1 conservative binary32 bound before exact evaluations         coop 0|1 wave-cooperative all-primitive loop
 *   nodes_in_lds 0|1 scene tables staged in LDS (v2)                        list_cap 1..64 hit-leaf list entries per ray (v2)
 *   grid 0|1 leaf grid for BVH.getPrimitivesAt (v2)                         nn 0|1|2 nearest-candidate grid off|on|auto
 *   recs, lut, sub 0|1 octree: leaf-ordered records, findNode cell table, sub-cell candidate lists
 *   blocks_per_cu 1..8, refill 1..64, hw_xcd 0|1, item_px 64|128|256       persistent-kernel scheduling (v2)
 *   static 0..95 percent of every tile queue assigned to the waves without atomics (v2; for overlapping frames)
 *   uniform 0|1   scenes whose spheres share one radius: rank leaf candidates by squared centre distance (v2, default 1)
 *   rel 0|1       BVH node boxes relative to the frame's ray origin, as doubles in LDS, when they fit (v2, default 1)
 *   cull 0|1      whole 64-pixel batches find their hit BVH leaves by a bundle-frustum cull (v2, <= 256 leaves, default 1)
 *   lds_kb 0|16..64  LDS budget per workgroup the v2 launcher trims the per-ray hit lists to (0, default: as many workgroups per CU
 *                 as the kernel's registers allow -- six, 26 880 bytes each --, then five, then four; 32: five; 40: four)
 *   n0_batch 1..64 BVH (v2): getNormal is deferred until no lane of the wave needs a march distance, then evaluated for all waiting
 *                 rays in one round, the three offset samples taken from the sphere that gave d0 where provably the minimum;
 *                 lanes waiting that trigger that round early (64: never early; default 64)
 *   lpt 0|1       v2: hand out a launch's work items longest-first using the item durations the previous launch recorded (any
 *                 order gives the same bytes; default 1: a frame alone 1.07 against 1.14 ms; bench.py turns it off with frames in flight)
 *   multi_step 0|1  v2 BVH: a lane takes further march steps inside a round while the leaf set and the winning sphere provably
 *                 stay the same (default 1; same bytes either way)
 *   lds_fill 0|1  v2: pad the LDS request so that exactly blocks_per_cu workgroups fit a CU (default 0; measurement knob)
 *   item_wide 0|1 v2: the 64-pixel batches of a work item side by side (item = tile_w * item_px / 64 pixels wide) instead of one
 *                 above the other (default 0: measured no gain in write traffic, 2 % slower with frames in flight)
 *   specialise 0|1|2  small scenes: 1 (default) = the scene compiled into the kernel at run time (rm_rtc_* above), waiting for the
 *                 compile at the first render; 2 = the same without waiting -- a background thread compiles, the ahead-of-time
 *                 kernels render meanwhile (same bytes), the scene's own kernel takes over when it is ready; 0 = the
 *                 ahead-of-time kernels only (the device interpreter of csrc/rm_program.h for operator trees)
 *   specialise_v2_after 0..  the v2 wave loop: after this many launches with the same CONFIGURATION (frame size, shader, option
 *                 switches, the scene's counts and grids, tile geometry, LDS layout -- csrc/rm_v2_fields.h; not the camera, not
 *                 the rows of a launch) the kernel is compiled with that configuration's parameters as literals (~2 s; waiting
 *                 for it or not as `specialise` says) and used from then on: C3 at 4K 1 190 -> 1 310 frames/s.  Default 3;
 *                 0: never.  rm_last_kernel marks such launches "[launch constants compiled in]"
 *   rtc_spheres 0..33  sphere lists of fewer spheres than this (default 16) and primitive lists of up to 32 primitives are compiled
 *                 too, one single-leaf object per primitive, when their BVH has at most eight leaves (emitted as code: no node
 *                 walks in memory); they then run in the one-ray-per-lane kernel instead of the v2 wave loop
 *   prune 0|1     specialised kernels: smooth unions / subtractions over spheres, boxes and tori skip operands whose binary32
 *                 interval proves they cannot matter (exact: csrc/rm_rtc.cpp; default 1; read when a scene is built)
 *   v1_lists 0|1  v1 BVH kernels: per-ray hit-leaf lists (as v2) instead of one tree walk per interval advance (default 1)
 *   v1_block 64|128|256  v1 kernels: threads per workgroup (default 64: one wave, so wave slots refill one by one); without an
 *                 explicit tile_w the v1 kernels use 8 x 8-pixel wave tiles
 *   oct_lean 0|1  octree + sphere scene + sphere tracer: the lean kernel render_kernel_oct (march and getNormal as phases of one
 *                 loop, eight waves per SIMD, node boxes relative to the camera position from a per-camera table) instead of
 *                 render_kernel<1, false, 0> (default 1; needs recs, lut and filter on)
 * The one option that is NOT a measurement knob but part of the numeric contract:
 *   length 0|1    gl-matrix vec3.length / vec3.distance (sphere.ts:12-14, box.ts:26,33, mandelbulb.ts:46, smoothUnion.ts:45):
 *                 0 = Math.hypot(x, y, z) (gl-matrix 3.0 - 3.4.3, default), 1 = Math.sqrt(x*x + y*y + z*z) (the form a later
 *                 3.4.x release may use).  Results differ by <= 1 ulp(f64) per distance; the active scene is
 *                 rebuilt (bounding radii of boxes and smooth unions use it too).
 *                 NOTE: the reference pins gl-matrix 3.4.4 (package.json:25), whose source is not available offline, while the
 *                 DEFAULT follows the formula of 3.0 - 3.4.3.  Nothing in the reference tree decides between the two (no
 *                 fixture, no test): parity is unpinned on this point.  Goldens, the cross-check shim and the benchmark exist
 *                 for both modes (tests/golden/, bench.py --opt length=1), so the default is a one-line change
 *                 (rm_api.cpp: opt_length) once the pinned version's formula is known. */
RM_API int rm_set_option(rm_ctx *ctx, const char *key, int64_t value);
RM_API int rm_get_option(const rm_ctx *ctx, const char *key, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* RM_RAYMARCH_H */
