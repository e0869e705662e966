// rm_api.cpp -- the C ABI of include/rm_raymarch.h on top of the HIP kernels.
//
// Host side of the tile dispatch (reference src/workers/raymarchWorker.ts:33-92): a job
// names (preset, accel, camera angles, row range); the ctx keeps the built scene resident
// in HBM and rebuilds it only when (preset, accel) changes -- the reference rebuilds it
// twice per tile per frame (scene.ts:24-29 then raymarchWorker.ts:38).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <unordered_map>
#include <new>
#include <string>
#include <vector>

#include "../../include/rm_raymarch.h"
#include "rm_diag.h"
#include "rm_kernels.h"
#include "rm_rtc.h"
#include "rm_scene_host.h"

namespace {

struct DeviceScene {
    RmSphere *spheres = nullptr;
    double *radii = nullptr;
    RmBvhNode *bvh = nullptr;
    int32_t *bvh_prims = nullptr;
    RmOctNode *oct = nullptr;
    int32_t *oct_prims = nullptr;
    uint32_t *pq_cells = nullptr;
    uint16_t *pq_list = nullptr;
    uint32_t *nn_cells = nullptr;
    uint16_t *nn_list = nullptr;
    RmPrim *prims = nullptr;
    RmInstr *prog = nullptr;
    int32_t *obj_ranges = nullptr;
    RmSphereRec *oct_recs = nullptr;
    int32_t *oct_lut = nullptr;
    uint32_t *oct_sub_hdr = nullptr;
    uint8_t *oct_sub_list = nullptr;
    uint16_t *bvh_leaves = nullptr;  // node indices of the non-empty BVH leaves, increasing (v2 bundle cull)
    int32_t bvh_leaf_count = 0;      // 0: no cull (no BVH, too many leaves, or a leaf box not inside its ancestors')
};

}  // namespace

struct rm_ctx {
    int device = -1;
    bool has_device = false;
    std::string err = "";
    hipStream_t stream = nullptr;  // for the host-buffer entry points

    bool have_scene = false;
    bool scene_is_uploaded = false;  // active scene came from rm_scene_from_spheres
    int scene_preset = 0;
    rmh::HostScene host;
    DeviceScene dev;

    bool have_uploaded = false;  // sphere list kept for accel changes by later jobs
    std::vector<float> up_centers;
    std::vector<double> up_radii;
    std::vector<rmh::PrimDesc> up_prims;  // when the uploaded scene came from rm_scene_from_prims
    bool up_general = false;
    std::vector<rmh::NodeDesc> up_nodes;  // when it came from rm_scene_from_nodes
    std::vector<int> up_roots;
    bool up_program = false;
    double time = 0.0;  // Scene.updateTime: the last job's time, or rm_scene_set_time

    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    RmDiagDevice *d_diag = nullptr;
    float light[3] = {0, 0, 0};

    int64_t opt_tile_w = 8;  // 8 x 8-pixel batches (a frame alone, round 3: 128-pixel items of 8 x 16: 1.09 ms; of 16 x 8: 1.17 ms)
    bool tile_w_set = false;  // rm_set_option("tile_w") was called: the value then holds for every kernel (else v1 kernels use 8 x 8 wave tiles)
    int64_t opt_filter = 1;
    int64_t opt_lds = 1;
    int64_t opt_kernel = 0;  // 0 = auto: v2 for BVH / no acceleration, v1 for the octree (its lean kernel for sphere scenes: opt_oct_lean) and small scenes
    int64_t opt_list_cap = 32;
    int64_t opt_coop = 1;
    int64_t opt_grid = 1;
    int64_t opt_nn = 2;  // per-cell nearest-candidate lists for the all-primitive fallback: 0 off, 1 on, 2 auto (scenes of
                         // <= 512 spheres, where the 48^3 candidate grid keeps the lists short: C3 2.65 -> 2.62 ms)
    int64_t opt_blocks_per_cu = 6;  // persistent workgroups per launch and CU: what the kernel's 80 VGPRs and 26 KB of LDS allow (a frame alone, round 3: 4: 1.42 ms, 6: 1.32, 7: 1.30)
    int64_t opt_lds_fill = 0;
    int64_t opt_item_wide = 0;  // v2: the 64-pixel batches of an item side by side (1) or one above the other (0)
    int64_t opt_multi_step = 1;  // v2 BVH: in-round march steps (rm_render_v2.hip, section M)
    int64_t opt_lds_kb = 0;         // v2: LDS budget per workgroup the launcher trims the hit lists to (0: as many workgroups per CU as the kernel's registers allow; 32: five per CU, 40: four)
    int64_t opt_refill = 64;
    int64_t opt_recs = 1;  // octree leaves read leaf-ordered sphere records
    int64_t opt_static = 0;  // accepted and ignored since round 3 (a static share of every tile queue, overtaken in round 2; the 64 queue heads of round 3 removed its reason)
    int64_t opt_lut = 1;   // Octree.findNode through the 64^3 cell table
    int64_t opt_cull = 1;     // v2: whole 64-pixel batches find their hit leaves by a bundle-frustum cull instead of the tree walk
    int64_t opt_rel = 1;      // v2: BVH node boxes relative to the frame's ray origin, as doubles in LDS (slab test without conversions)
    int64_t opt_uniform = 1;  // v2: scenes whose spheres all have one radius rank candidates by squared centre distance
    int64_t opt_sub = 1;   // crowded octree leaves scan their sub-cell candidate lists
    int64_t opt_hw_xcd = 1;
    int64_t opt_v1_lists = 1;  // v1 BVH kernels: per-ray hit-leaf lists instead of one tree walk per interval advance
    int64_t opt_v1_block = 64;  // v1 kernels: threads per workgroup (one wave: wave slots refill one by one)
    int64_t opt_oct_lean = 1;  // octree, sphere scenes, sphere tracer: render_kernel_oct (rm_kernels.hip) instead of render_kernel<1, false, 0>
    int64_t opt_lpt = 1;  // v2: longest-first item order from the previous frame's item durations (shortens the tail of a frame that runs alone: 1.13 -> 1.08 ms)
    // LPT buffers: a ring of slots, one per launch in flight (a launch sorts from the previous launch's costs into its own
    // permutation and records its own costs); geometry changes restart the feedback
    static constexpr int kLptSlots = 16;
    static constexpr int kLptQueues = 64;     // rm_render_v2.hip RM_QUEUES
    static constexpr int kLptStride = 4096;   // items per queue; larger frames render without LPT
    uint8_t *d_lpt_cost = nullptr;    // [kLptSlots][kLptQueues * kLptStride]
    uint16_t *d_lpt_perm = nullptr;   // [kLptSlots][kLptQueues * kLptStride]
    unsigned int lpt_launch = 0;
    long long lpt_geometry = -1;
    hipEvent_t lpt_done[16] = {};  // recorded after the launch that owns the slot: a slot is reused only once that launch is over
    // render_kernel_oct: the octree relative to a camera position (RmOctFrameNode), one table per position in a ring.  A table is
    // shared by every launch with that position (frames in flight with one camera build it once) and rewritten only when the
    // ring comes round, after every launch that read it has finished (one event per stream that used it).
    struct OctFrameSlot {
        double origin[3] = {0, 0, 0};
        unsigned long long gen = 0;
        bool valid = false;
        RmOctFrameNode *dev = nullptr;
        size_t nodes = 0;
        std::vector<std::pair<hipStream_t, hipEvent_t>> users;
        hipStream_t builder = nullptr;  // the stream oct_frame_table_kernel ran on ...
        hipEvent_t built = nullptr;     // ... and the event recorded right behind it: a launch on ANY OTHER stream waits for it
    };
    static constexpr int kOctFrameSlots = 32;
    OctFrameSlot oct_frames[kOctFrameSlots];
    unsigned oct_frame_next = 0;
    int oct_frame_cur = -1;
    unsigned long long scene_gen = 0;  // bumped by every upload_scene
    int64_t opt_n0_batch = 64;  // v2 BVH: see RmRenderParams::n0_batch
    int64_t opt_length = 0;  // vec3.length: 0 Math.hypot (gl-matrix 3.0 - 3.4.3), 1 Math.sqrt(x*x + y*y + z*z)
    const char *last_kernel = "";
    // Expression forests: the scene's programs compiled into the one-ray-per-lane kernels at run time (rm_rtc.h; option
    // `specialise`, default on).  One kernel per (acceleration structure, marcher family, vec3.length form) the scene is
    // rendered with, compiled at its first launch (1.5 - 3 s, synchronous) and kept until the scene is replaced.  A compile
    // that fails is remembered with its log (rm_rtc_status) and the interpreter of rm_program.h serves the scene.
    int64_t opt_specialise = 1;
    // sphere lists of fewer spheres than this run in the one-ray-per-lane kernels as the scene's own code (C2, nine spheres at
    // 1080p: 0.22 ms alone against 0.32 in the v2 wave loop, 8 780 against 8 340 frames/s in flight); without hiprtc the rule is < 8
    int64_t opt_rtc_spheres = 16;
    int64_t opt_prune = 1;  // specialised kernels: exact pruning of smooth unions / subtractions (rm_rtc.cpp); takes effect at the next scene build
    std::string rtc_src;
    std::map<int, rmrtc::Kernel> rtc_kernels;   // handles; the modules belong to rm_rtc's process-wide cache ...
    std::vector<rmrtc::Kernel> rtc_owned;       // ... except these (compiled while the cache was full)
    std::map<int, std::string> rtc_failed;
    std::string rtc_log;  // of the most recent compile
    // The v2 wave loop compiled for a launch configuration that keeps coming back (rm_v2_fields.h, rm_rtc_v2_hook below): after
    // `specialise_v2_after` launches with the same configuration (default 3; 0: never) the kernel is compiled with that
    // configuration's parameters as literals (~2 s; synchronous with `specialise` = 1, in the background with 2) and used from
    // then on.  Keyed by a hash of the configuration; the modules live in rm_rtc's process-wide cache.
    struct V2Special {
        int launches = 0;
        int state = 0;  // 0 counting, 2 compiling in the background, 1 ready, -1 failed
        bool lite = false;  // the first compile (with the scene's counts as literals) was refused: without them
        rmrtc::Kernel k;
    };
    std::unordered_map<uint64_t, V2Special> v2_special;
    int64_t opt_v2_after = 3;
    // small host tables the sharded entry points need on the device (stripe lists, stripe -> source maps): cached by
    // content, each in its own allocation, so a table a launch in flight still reads is never overwritten
    struct DevTable {
        std::vector<int32_t> host;
        int32_t *dev = nullptr;
    };
    std::vector<DevTable> tables;
    int64_t opt_item_px = 128;  // two 64-pixel batches per queue claim: 7-10 % faster than 64 at the end of round 1, 256 loses
    unsigned int *d_counters = nullptr;  // ring of 1024 x 64 queue heads, 256 bytes apart (rm_render_v2.hip RM_QSTRIDE): a launch owns its slot until 1023 later launches
                                         // have been enqueued (frames in flight on several streams each need their own)
    unsigned int counter_slot = 0;
    // Fused diagnostics (rm_diag.h): a ring of accumulator blocks, one per launch in flight like the queue heads above;
    // zeroed once here, left zeroed by every launch's last wave.  diag_next: rm_render_attach_diagnostics, one-shot.
    static constexpr unsigned kDiagBlocks = 1024;
    RmDiagBlock *d_diag_blocks = nullptr;
    unsigned int diag_slot = 0;
    void *diag_next = nullptr;
    unsigned long long *d_stamps = nullptr;  // diagnostic build only
    int num_cus = 256;
};

namespace {

int fail(rm_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    return code;
}

int hip_fail(rm_ctx *ctx, hipError_t e, const char *what) {
    return fail(ctx, RM_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

// the kernels exist twice: vec3.length = Math.hypot, and = sqrt(x*x + y*y + z*z) (rm_kernels.h, option `length`)
// the scene's specialised kernel for (accel, marcher family, length form), compiled on first use; nullptr: the interpreter serves
const rmrtc::Kernel *specialised_kernel(rm_ctx *ctx, int accel, bool other) {
    if (!ctx->opt_specialise || ctx->rtc_src.empty() || !ctx->has_device) return nullptr;
    const int key = (accel * 2 + (other ? 1 : 0)) * 2 + (ctx->opt_length ? 1 : 0);
    auto it = ctx->rtc_kernels.find(key);
    if (it != ctx->rtc_kernels.end()) return &it->second;
    if (ctx->rtc_failed.count(key)) return nullptr;
    rmrtc::Kernel k;
    bool cached = false;
    if (ctx->opt_specialise == 2) {  // never wait: the ahead-of-time kernels render until the background compile has finished
        const int st = rmrtc::compile_async(ctx->device, ctx->rtc_src, accel, other, ctx->opt_length != 0, k, ctx->rtc_log);
        if (st < 0) ctx->rtc_failed[key] = ctx->rtc_log;
        if (st <= 0) return nullptr;
        return &(ctx->rtc_kernels[key] = k);
    }
    if (!rmrtc::compile_cached(ctx->device, ctx->rtc_src, accel, other, ctx->opt_length != 0, k, ctx->rtc_log, &cached)) {
        ctx->rtc_failed[key] = ctx->rtc_log;
        return nullptr;
    }
    if (!cached) ctx->rtc_owned.push_back(k);  // the process-wide cache is full: this context unloads the module with the scene
    return &(ctx->rtc_kernels[key] = k);
}

}  // namespace

const void *rm_rtc_v2_hook(const RmRenderParams &p, int accel, bool lds, bool ur, bool rel, bool length_sqrt) {
    rm_ctx *ctx = static_cast<rm_ctx *>(p.rtc_ctx);
    if (!ctx || !ctx->opt_specialise || ctx->opt_v2_after <= 0) return nullptr;
    const int bits = (accel & 3) | (lds ? 4 : 0) | (ur ? 8 : 0) | (rel ? 16 : 0) | rmrtc::kV2Bit;
    rm_ctx::V2Special &e = ctx->v2_special[rmrtc::v2_fixed_hash(p, bits | (length_sqrt ? 512 : 0))];
    if (e.state == 1) return e.k.render;
    if (e.state < 0) return nullptr;
    if (e.state == 0 && ++e.launches < ctx->opt_v2_after) return nullptr;
    // with the scene's counts as literals first; a kernel refused for spilling (small scenes: their loops unroll) once more without
    for (;;) {
        const std::string src = rmrtc::v2_fixed_source(p, !e.lite);
        bool failed = false;
        if (ctx->opt_specialise == 2) {
            const int st = rmrtc::compile_async(ctx->device, src, bits, false, length_sqrt, e.k, ctx->rtc_log);
            e.state = st > 0 ? 1 : (st < 0 ? -1 : 2);
            failed = st < 0;
        } else {
            bool cached = false;
            failed = !rmrtc::compile_cached(ctx->device, src, bits, false, length_sqrt, e.k, ctx->rtc_log, &cached);
            e.state = failed ? -1 : 1;
            if (!failed && !cached) ctx->rtc_owned.push_back(e.k);
        }
        if (!failed || e.lite) break;
        e.lite = true;
        e.state = 0;
    }
    return e.state == 1 ? e.k.render : nullptr;
}

namespace {

hipError_t launch_render(rm_ctx *ctx, const RmRenderParams &p_in, hipStream_t stream) {
    RmRenderParams p = p_in;
#if !defined(RM_STAMPS) && !defined(RM_COUNTS) && !defined(RM_STAMPS_LOG) && !defined(RM_STAMPS_CLAIM)  // (diagnostic builds keep their instrumented kernels)
    p.rtc_ctx = ctx;
#endif
    // every launch of the one-ray-per-lane kernels (not the v2 wave loop, not the lean octree kernel) takes the scene's own kernel if it has one
    const bool lean_oct = p.accel == 1 && p.oct_lean && !p.general && p.algorithm == 0 && p.oct_lut && p.oct_recs && p.filter;
    const rmrtc::Kernel *special = (p.variant == 2 && p.algorithm == 0) || lean_oct ? nullptr : specialised_kernel(ctx, p.accel, p.algorithm != 0);
    p.rtc_function = special ? special->render : nullptr;
    rm_ctx::OctFrameSlot *oct_slot = nullptr;
    if (lean_oct) {
        const size_t n = static_cast<size_t>(p.oct_nodes);
        auto matches = [&](const rm_ctx::OctFrameSlot &sl) {
            return sl.valid && sl.gen == ctx->scene_gen && sl.origin[0] == p.origin_d[0] && sl.origin[1] == p.origin_d[1] && sl.origin[2] == p.origin_d[2];
        };
        if (ctx->oct_frame_cur >= 0 && matches(ctx->oct_frames[ctx->oct_frame_cur])) oct_slot = &ctx->oct_frames[ctx->oct_frame_cur];
        for (int k = 0; !oct_slot && k < rm_ctx::kOctFrameSlots; ++k)
            if (matches(ctx->oct_frames[k])) {
                oct_slot = &ctx->oct_frames[k];
                ctx->oct_frame_cur = k;
            }
        // A table found here may still be under construction on the stream that first saw this camera position (frames
        // in flight share one table): a reader on another stream orders itself behind the builder (the `users`
        // events only made the next WRITER wait for the readers, never a reader for the writer).
        bool seen = false;  // a stream that has read this table before is already ordered behind the builder
        if (oct_slot)
            for (auto &u : oct_slot->users) seen = seen || u.first == stream;
        if (oct_slot && oct_slot->builder != stream && !seen) {
            if (oct_slot->built) {
                if (hipStreamWaitEvent(stream, oct_slot->built, 0) != hipSuccess) (void)hipStreamSynchronize(oct_slot->builder);
            } else {
                (void)hipStreamSynchronize(oct_slot->builder);  // no event could be created when it was built
            }
        }
        if (!oct_slot) {  // a new camera position: the oldest table is rewritten once its readers are done
            const int k = static_cast<int>(ctx->oct_frame_next++ % rm_ctx::kOctFrameSlots);
            rm_ctx::OctFrameSlot &sl = ctx->oct_frames[k];
            for (auto &u : sl.users) {
                (void)hipEventSynchronize(u.second);
                (void)hipEventDestroy(u.second);
            }
            sl.users.clear();  // `seen` above is about THIS table, not the one the slot held before
            sl.valid = false;
            if (sl.nodes < n) {
                (void)hipFree(sl.dev);
                sl.dev = nullptr;
                sl.nodes = 0;
                if (hipMalloc(reinterpret_cast<void **>(&sl.dev), n * sizeof(RmOctFrameNode)) == hipSuccess) sl.nodes = n;
                else (void)hipGetLastError();
            }
            if (sl.dev && rm_launch_oct_frame_table(p.oct, p.oct_nodes, p.origin_d, sl.dev, stream) == hipSuccess) {
                for (int a = 0; a < 3; ++a) sl.origin[a] = p.origin_d[a];
                sl.gen = ctx->scene_gen;
                sl.valid = true;
                sl.builder = stream;
                if (!sl.built && hipEventCreateWithFlags(&sl.built, hipEventDisableTiming) != hipSuccess) sl.built = nullptr;
                if (!sl.built || hipEventRecord(sl.built, stream) != hipSuccess) (void)hipStreamSynchronize(stream);  // built before anyone else can look
                oct_slot = &sl;
                ctx->oct_frame_cur = k;
            }
        }
        p.oct_frame = oct_slot ? oct_slot->dev : nullptr;  // null: the launcher takes render_kernel<1, false, 0>
    }
    if (ctx->opt_lpt && p.variant == 2 && p.algorithm == 0) {  // longest-first item order (rm_render_v2.hip, lpt_sort_kernel)
        const size_t per_slot = static_cast<size_t>(rm_ctx::kLptQueues) * rm_ctx::kLptStride;
        if (!ctx->d_lpt_cost) {
            if (hipMalloc(reinterpret_cast<void **>(&ctx->d_lpt_cost), rm_ctx::kLptSlots * per_slot) != hipSuccess ||
                hipMalloc(reinterpret_cast<void **>(&ctx->d_lpt_perm), rm_ctx::kLptSlots * per_slot * sizeof(uint16_t)) != hipSuccess) {
                (void)hipFree(ctx->d_lpt_cost);
                ctx->d_lpt_cost = nullptr;
                ctx->d_lpt_perm = nullptr;
            }
        }
        if (ctx->d_lpt_cost) {
            // the costs are per item of THIS tiling of THIS row set: any change restarts the feedback (costs read as zero)
            const long long geom = ((static_cast<long long>(p.width) * 65536 + p.local_rows) * 1024 + p.tile_w) * 1024 + p.item_px + (p.item_wide ? 512 : 0) +
                                   (p.stripe_rows ? (1ll << 62) + p.part * 131 + p.n_parts : 0) + p.y_start * 7919ll;
            const unsigned slot = ctx->lpt_launch % rm_ctx::kLptSlots, prev = (ctx->lpt_launch + rm_ctx::kLptSlots - 1) % rm_ctx::kLptSlots;
            const bool have_prev = ctx->lpt_launch > 0 && geom == ctx->lpt_geometry;
            ctx->lpt_geometry = geom;
            ctx->lpt_launch++;
            if (ctx->lpt_done[slot]) (void)hipEventSynchronize(ctx->lpt_done[slot]);  // 16 launches ago: normally long finished
            p.lpt_stride = rm_ctx::kLptStride;
            p.lpt_cost_prev = have_prev ? ctx->d_lpt_cost + prev * per_slot : nullptr;
            p.lpt_cost_out = ctx->d_lpt_cost + slot * per_slot;
            p.lpt_perm_out = ctx->d_lpt_perm + slot * per_slot;
        }
    }
    // Fused diagnostics (rm_render_attach_diagnostics; one-shot).  v2 launches always get an accumulator block: their last
    // wave also re-zeroes the launch's tile-queue heads.  A launch without pixels runs no kernel: neutral elements then.
    p.diag_out = static_cast<RmDiagDevice *>(ctx->diag_next);
    ctx->diag_next = nullptr;
    p.diag_block = nullptr;
    const bool empty = p.local_rows <= 0 || p.width <= 0;
    const bool v2 = p.variant == 2 && p.algorithm == 0;
    // The one-ray-per-lane kernels run one-wave workgroups: 129 600 waves per 4K frame, each of which would end with five
    // device-scope atomics and a wait for four of them.  Measured (round 3, frames in flight): Pyramid of Boxes 1 348 frames/s
    // with that epilogue against 1 561 with the two reduction launches over the stored counters, C5 493 against 512.  So
    // when the call stores both counter buffers the diagnostics of a v1 launch come from rm_launch_reduce behind it (still one
    // call for the host, no host synchronisation); the in-kernel epilogue serves the calls that store no counters.
    RmDiagDevice *reduce_after = nullptr;
    if (p.diag_out && !v2 && !empty && p.sdf && p.iters) {
        reduce_after = p.diag_out;
        p.diag_out = nullptr;
    }
    if (!empty && ctx->d_diag_blocks && (p.diag_out || v2))
        p.diag_block = ctx->d_diag_blocks + (ctx->diag_slot++ % rm_ctx::kDiagBlocks);
    if (p.diag_out && empty) {
        const hipError_t ei = rm_launch_reduce_init(p.diag_out, stream);
        if (ei != hipSuccess) return ei;
    }
    hipError_t e = ctx->opt_length ? rm_launch_render_sqrt(p, stream, &ctx->last_kernel) : rm_launch_render(p, stream, &ctx->last_kernel);
    if (special && !empty) ctx->last_kernel = special->name.c_str();
    if (e == hipSuccess && reduce_after) {
        e = rm_launch_reduce_init(reduce_after, stream);
        if (e == hipSuccess) e = rm_launch_reduce(p.sdf, p.iters, static_cast<int64_t>(p.local_rows) * p.width, reduce_after, stream);
    }
    if (oct_slot) {  // this stream now reads the table: whoever rewrites it waits for this launch
        hipEvent_t *ev = nullptr;
        for (auto &u : oct_slot->users)
            if (u.first == stream) ev = &u.second;
        if (!ev) {
            hipEvent_t fresh = nullptr;
            if (hipEventCreateWithFlags(&fresh, hipEventDisableTiming) == hipSuccess) {
                oct_slot->users.emplace_back(stream, fresh);
                ev = &oct_slot->users.back().second;
            }
        }
        if (ev) (void)hipEventRecord(*ev, stream);
        else (void)hipStreamSynchronize(stream);  // no event to be had: the table must not outlive its reader unseen
    }
    if (e == hipSuccess && p.lpt_perm_out) {
        const unsigned slot = (ctx->lpt_launch + rm_ctx::kLptSlots - 1) % rm_ctx::kLptSlots;
        if (!ctx->lpt_done[slot]) (void)hipEventCreateWithFlags(&ctx->lpt_done[slot], hipEventDisableTiming);
        if (ctx->lpt_done[slot]) (void)hipEventRecord(ctx->lpt_done[slot], stream);
    }
    return e;
}

#define RM_HIP(ctx, call)                                       \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

void free_device_scene(rm_ctx *ctx) {
    for (auto &k : ctx->rtc_owned) rmrtc::release(k);  // (callers have synchronised the device)
    ctx->rtc_owned.clear();
    ctx->rtc_kernels.clear();
    ctx->v2_special.clear();  // (a new scene is a new configuration anyway: its counts and LDS layout are in the key)
    ctx->rtc_failed.clear();
    ctx->last_kernel = "";
    if (!ctx->has_device) return;
    DeviceScene &d = ctx->dev;
    (void)hipFree(d.spheres);
    (void)hipFree(d.radii);
    (void)hipFree(d.bvh);
    (void)hipFree(d.bvh_prims);
    (void)hipFree(d.oct);
    (void)hipFree(d.oct_prims);
    (void)hipFree(d.pq_cells);
    (void)hipFree(d.pq_list);
    (void)hipFree(d.nn_cells);
    (void)hipFree(d.nn_list);
    (void)hipFree(d.prims);
    (void)hipFree(d.prog);
    (void)hipFree(d.obj_ranges);
    (void)hipFree(d.oct_recs);
    (void)hipFree(d.oct_lut);
    (void)hipFree(d.oct_sub_hdr);
    (void)hipFree(d.oct_sub_list);
    (void)hipFree(d.bvh_leaves);
    d = DeviceScene();
}

// device copy of a small host table, cached by content (see rm_ctx::tables)
int device_table(rm_ctx *ctx, const int32_t *host, size_t n, const int32_t **out) {
    for (auto &t : ctx->tables)
        if (t.host.size() == n && std::memcmp(t.host.data(), host, n * sizeof(int32_t)) == 0) {
            *out = t.dev;
            return RM_OK;
        }
    if (ctx->tables.size() >= 256) {  // far more than any run deals; start over rather than grow without bound
        RM_HIP(ctx, hipDeviceSynchronize());
        for (auto &t : ctx->tables) (void)hipFree(t.dev);
        ctx->tables.clear();
    }
    rm_ctx::DevTable t;
    t.host.assign(host, host + n);
    RM_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&t.dev), (n ? n : 1) * sizeof(int32_t)));
    if (n) RM_HIP(ctx, hipMemcpy(t.dev, host, n * sizeof(int32_t), hipMemcpyHostToDevice));
    *out = t.dev;
    ctx->tables.push_back(std::move(t));
    return RM_OK;
}

template <typename T>
int upload_vec(rm_ctx *ctx, const std::vector<T> &v, T **out) {
    *out = nullptr;
    const size_t bytes = (v.empty() ? 1 : v.size()) * sizeof(T);
    RM_HIP(ctx, hipMalloc(reinterpret_cast<void **>(out), bytes));
    if (!v.empty()) RM_HIP(ctx, hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return RM_OK;
}

int upload_scene(rm_ctx *ctx) {
    // Scenes the one-ray-per-lane kernels serve and that are small enough to be code (rm_rtc.h): expression forests, and --
    // as one single-leaf object per primitive -- primitive lists (up to 32) and sphere lists of fewer than `rtc_spheres` (16)
    // spheres, whose BVH has at most eight leaves (larger ones run in the v2 wave loop).
    ctx->rtc_src.clear();
    if (ctx->host.program) {
        ctx->rtc_src = rmrtc::scene_source(ctx->host.prog, ctx->host.obj_ranges, ctx->host.prog_tree, ctx->host.prog_roots, ctx->opt_prune != 0, ctx->host.bvh,
                                           ctx->host.bvh_prims, false);
    } else if (ctx->host.general ? ctx->host.prims.size() <= 32 : static_cast<int64_t>(ctx->host.spheres.size()) < ctx->opt_rtc_spheres) {
        std::vector<RmInstr> prog;
        std::vector<int32_t> ranges, roots;
        std::vector<rmh::ProgTreeNode> tree;
        if (rmh::leaf_objects(ctx->host, prog, ranges, tree, roots))
            ctx->rtc_src = rmrtc::scene_source(prog, ranges, tree, roots, false, ctx->host.bvh, ctx->host.bvh_prims, true);
    }
    if (!ctx->has_device) {
        free_device_scene(ctx);
        return RM_OK;
    }
    RM_HIP(ctx, hipSetDevice(ctx->device));
    RM_HIP(ctx, hipDeviceSynchronize());  // nothing may still read the old tables
    free_device_scene(ctx);
    ctx->scene_gen++;  // the octree frame tables of the old scene are stale
    int rc;
    if ((rc = upload_vec(ctx, ctx->host.spheres, &ctx->dev.spheres))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.radii, &ctx->dev.radii))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.bvh, &ctx->dev.bvh))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.bvh_prims, &ctx->dev.bvh_prims))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.oct, &ctx->dev.oct))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.oct_prims, &ctx->dev.oct_prims))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.pq_cells, &ctx->dev.pq_cells))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.pq_list, &ctx->dev.pq_list))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.nn_cells, &ctx->dev.nn_cells))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.nn_list, &ctx->dev.nn_list))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.prims, &ctx->dev.prims))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.prog, &ctx->dev.prog))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.obj_ranges, &ctx->dev.obj_ranges))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.oct_recs, &ctx->dev.oct_recs))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.oct_lut, &ctx->dev.oct_lut))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.oct_sub_hdr, &ctx->dev.oct_sub_hdr))) return rc;
    if ((rc = upload_vec(ctx, ctx->host.oct_sub_list, &ctx->dev.oct_sub_list))) return rc;
    // Leaf table of the bundle cull (rm_render_v2.hip, bvh_prologue_cull).  The cull tests leaves directly, which
    // equals the reference's traversal only if a hit leaf implies hit ancestors: every box must lie inside its
    // parent's, bit for bit (the builder takes unions, so it does; verified here rather than assumed).
    {
        const auto &bvh = ctx->host.bvh;
        std::vector<uint16_t> leaves;
        bool ok = !bvh.empty() && bvh.size() < 65536;
        std::vector<int> parent(bvh.size(), -1);
        for (size_t i = 0; ok && i < bvh.size(); ++i) {
            if (bvh[i].leaf >= 0) continue;
            for (size_t c = i + 1; c < static_cast<size_t>(bvh[i].skip) && c < bvh.size(); c = static_cast<size_t>(bvh[c].skip)) {
                parent[c] = static_cast<int>(i);
                if (bvh[c].skip <= static_cast<int>(c)) { ok = false; break; }
            }
        }
        for (size_t i = 0; ok && i < bvh.size(); ++i) {
            if (parent[i] >= 0)
                for (int k = 0; k < 3; ++k)
                    if (!(bvh[i].lo[k] >= bvh[parent[i]].lo[k] && bvh[i].hi[k] <= bvh[parent[i]].hi[k])) ok = false;
            for (int k = 0; k < 3; ++k)  // a negative radius gives lo > hi: the slab test then is not monotone in the box
                if (!(bvh[i].lo[k] <= bvh[i].hi[k])) ok = false;
            if (i > 0 && parent[i] < 0) ok = false;  // not reached through the skip links: unknown shape
            if (bvh[i].leaf >= 0 && (bvh[i].leaf & 0xFF) > 0) leaves.push_back(static_cast<uint16_t>(i));
        }
        if (!ok || leaves.size() > 256) leaves.clear();
        ctx->dev.bvh_leaf_count = static_cast<int32_t>(leaves.size());
        if ((rc = upload_vec(ctx, leaves, &ctx->dev.bvh_leaves))) return rc;
    }
    return RM_OK;
}

int set_scene(rm_ctx *ctx, const float *centers, const double *radii, int n, int accel, bool uploaded, int preset) {
    std::string err;
    rmh::HostScene hs;
    rmh::set_length_mode(static_cast<int>(ctx->opt_length));
    if (!rmh::build_scene(hs, centers, radii, n, accel, err)) {
        const bool unsupported = err.find("BVH leaf") != std::string::npos;
        return fail(ctx, unsupported ? RM_E_UNSUPPORTED : RM_E_INVALID, err);
    }
    hs.preset = preset;
    ctx->host = std::move(hs);
    ctx->have_scene = true;
    ctx->scene_is_uploaded = uploaded;
    ctx->scene_preset = preset;
    return upload_scene(ctx);
}

int set_scene_general(rm_ctx *ctx, const rmh::PrimDesc *prims, int n, int accel, bool uploaded, int preset) {
    std::string err;
    rmh::HostScene hs;
    rmh::set_length_mode(static_cast<int>(ctx->opt_length));
    if (!rmh::build_scene_general(hs, prims, n, accel, err)) {
        const bool unsupported = err.find("BVH leaf") != std::string::npos;
        return fail(ctx, unsupported ? RM_E_UNSUPPORTED : RM_E_INVALID, err);
    }
    hs.preset = preset;
    ctx->host = std::move(hs);
    ctx->have_scene = true;
    ctx->scene_is_uploaded = uploaded;
    ctx->scene_preset = preset;
    return upload_scene(ctx);
}

int set_scene_nodes(rm_ctx *ctx, const rmh::NodeDesc *nodes, int n_nodes, const int *roots, int n_roots, int accel,
                    bool uploaded, int preset) {
    std::string err;
    rmh::HostScene hs;
    rmh::set_length_mode(static_cast<int>(ctx->opt_length));
    if (!rmh::build_scene_nodes(hs, nodes, n_nodes, roots, n_roots, accel, err)) {
        const bool unsupported = err.find("BVH leaf") != std::string::npos || err.find("RM_PROG_MAX") != std::string::npos;
        return fail(ctx, unsupported ? RM_E_UNSUPPORTED : RM_E_INVALID, err);
    }
    hs.preset = preset;
    ctx->host = std::move(hs);
    ctx->have_scene = true;
    ctx->scene_is_uploaded = uploaded;
    ctx->scene_preset = preset;
    return upload_scene(ctx);
}

int scene_n_prims(const rm_ctx *ctx) {
    if (ctx->host.program) return static_cast<int>(ctx->host.obj_ranges.size() / 2);
    return static_cast<int>(ctx->host.general ? ctx->host.prims.size() : ctx->host.spheres.size());
}

// which primitive representation the kernels read (RmRenderParams::general)
void fill_scene_repr(const rm_ctx *ctx, RmRenderParams &p) {
    p.general = ctx->host.general ? 1 : 0;
    if (ctx->host.program) {
        p.general = 2;
        for (const RmInstr &ins : ctx->host.prog)
            if (ins.op == 3) p.general = 3;  // Mandelbulb leaf: the instantiation that carries the fdlibm code
    }
    p.prims = ctx->dev.prims;
    p.prog = ctx->dev.prog;
    p.obj_ranges = ctx->dev.obj_ranges;
    p.prog_slots = ctx->host.prog_slots;
    p.prog_vals = ctx->host.prog_vals;
}

int clamp_preset(int idx) { return idx < 0 ? 0 : (idx > rmh::kPresetCount - 1 ? rmh::kPresetCount - 1 : idx); }
int norm_accel(int a) { return (a == RM_ACCEL_OCTREE || a == RM_ACCEL_BVH) ? a : RM_ACCEL_NONE; }

// makes the scene named by the job the active one (raymarchWorker.ts:37-38)
int ensure_scene(rm_ctx *ctx, int32_t preset_index, int32_t accel_in) {
    const int accel = norm_accel(accel_in);
    if (preset_index == RM_SCENE_UPLOADED) {
        if (!ctx->have_uploaded) return fail(ctx, RM_E_NO_SCENE, "no scene uploaded with rm_scene_from_spheres");
        if (ctx->have_scene && ctx->scene_is_uploaded && ctx->host.accel == accel) return RM_OK;
        if (ctx->up_program)
            return set_scene_nodes(ctx, ctx->up_nodes.data(), static_cast<int>(ctx->up_nodes.size()), ctx->up_roots.data(),
                                   static_cast<int>(ctx->up_roots.size()), accel, true, RM_SCENE_UPLOADED);
        if (ctx->up_general)
            return set_scene_general(ctx, ctx->up_prims.data(), static_cast<int>(ctx->up_prims.size()), accel, true,
                                     RM_SCENE_UPLOADED);
        return set_scene(ctx, ctx->up_centers.data(), ctx->up_radii.data(), static_cast<int>(ctx->up_radii.size()),
                         accel, true, RM_SCENE_UPLOADED);
    }
    const int preset = clamp_preset(preset_index);
    if (ctx->have_scene && !ctx->scene_is_uploaded && ctx->scene_preset == preset && ctx->host.accel == accel)
        return RM_OK;
    std::vector<float> c;
    std::vector<double> r;
    if (rmh::preset_spheres(preset, c, r))
        return set_scene(ctx, c.data(), r.data(), static_cast<int>(r.size()), accel, false, preset);
    std::vector<rmh::PrimDesc> prims;  // torus / box presets: general primitive records
    if (rmh::preset_prims(preset, prims))
        return set_scene_general(ctx, prims.data(), static_cast<int>(prims.size()), accel, false, preset);
    std::vector<rmh::NodeDesc> nodes;  // operator / Mandelbulb presets: expression programs
    std::vector<int> roots;
    if (rmh::preset_nodes(preset, nodes, roots))
        return set_scene_nodes(ctx, nodes.data(), static_cast<int>(nodes.size()), roots.data(),
                               static_cast<int>(roots.size()), accel, false, preset);
    return fail(ctx, RM_E_UNSUPPORTED, "scene preset " + std::to_string(preset) + " is not on the native path");
}

int ensure_scratch(rm_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return RM_OK;
    if (ctx->scratch) {
        RM_HIP(ctx, hipDeviceSynchronize());
        (void)hipFree(ctx->scratch);
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    RM_HIP(ctx, hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return RM_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int fill_params(rm_ctx *ctx, const rm_job *job, RmRenderParams &p) {
    if (!job) return fail(ctx, RM_E_INVALID, "null job");
    if (job->height <= 0 || job->width < 0) return fail(ctx, RM_E_INVALID, "width must be >= 0 and height > 0");
    if (!std::isfinite(job->camera_pitch) || !std::isfinite(job->camera_yaw))
        return fail(ctx, RM_E_INVALID, "non-finite camera angle");
    if (job->algorithm < RM_ALG_SPHERE_TRACER || job->algorithm > RM_ALG_ADAPTIVE_STEP_V3)
        return fail(ctx, RM_E_INVALID, "unknown algorithm enum (use rm_algorithm_from_string)");
    int rc = ensure_scene(ctx, job->scene_preset_index, job->acceleration_structure);
    if (rc) return rc;
    std::memset(&p, 0, sizeof p);
    p.width = job->width;
    p.height = job->height;
    p.y_start = job->y_start;
    p.y_end = job->y_end;
    p.algorithm = job->algorithm;
    // NaN stands for JS `undefined`: the constructors' defaults (adaptiveStepV2.ts:13, fixedStep.ts:13)
    p.overshoot = std::isnan(job->overshoot_factor) ? 1.2 : job->overshoot_factor;
    p.step_size = std::isnan(job->step_size) ? 0.1 : job->step_size;
    p.local_rows = job->y_end > job->y_start ? job->y_end - job->y_start : 0;
    p.stripe_rows = 0;
    p.n_parts = 1;
    p.part = 0;
    rmh::camera_from_angles(job->camera_pitch, job->camera_yaw, p.rot, p.origin);
    std::memcpy(p.light, ctx->light, sizeof p.light);
    for (int k = 0; k < 9; ++k) p.rot_d[k] = p.rot[k];
    for (int k = 0; k < 3; ++k) {
        p.origin_d[k] = p.origin[k];
        p.light_d[k] = p.light[k];
    }
    p.n_prims = scene_n_prims(ctx);
    fill_scene_repr(ctx, p);
    if (!std::isfinite(job->time)) return fail(ctx, RM_E_INVALID, "non-finite time");
    ctx->time = job->time;  // raymarcher.ts:58-59 scene.updateTime(time)
    p.time = job->time;
    p.accel = ctx->host.accel;
    p.bvh_nodes = static_cast<int32_t>(ctx->host.bvh.size());
    p.oct_nodes = static_cast<int32_t>(ctx->host.oct.size());
    p.tile_w = static_cast<int32_t>(ctx->opt_tile_w);
    p.nodes_in_lds = static_cast<int32_t>(ctx->opt_lds);
    p.filter = static_cast<int32_t>(ctx->opt_filter);
    p.variant = static_cast<int32_t>(ctx->opt_kernel);
    // auto: the octree runs in the one-ray-per-lane kernels; so do the smallest scenes.  (Until round 2 the 9-sphere grid of C2
    // was faster there too; since the in-round march steps of round 3 the uniform wave loop wins it: 0.308 against 0.331 ms
    // alone, 9 680 against 8 920 frames/s in flight.)
    if (p.variant == 0)
        p.variant = (ctx->host.accel == RM_ACCEL_OCTREE || ctx->host.spheres.size() < 8 ||
                     (ctx->opt_specialise && !ctx->rtc_src.empty() && rmrtc::available(nullptr) &&
                      static_cast<int64_t>(ctx->host.spheres.size()) < ctx->opt_rtc_spheres))
                        ? 1
                        : 2;
    if (p.algorithm != RM_ALG_SPHERE_TRACER) p.variant = 1;  // the other marchers live in the v1 kernel
    if (ctx->host.general) p.variant = 1;                    // so do boxes, tori and rotated primitives
    // v1: square 8 x 8 wave tiles keep a wave's rays in the same leaves / intervals (N3-mixed 2.31 -> 2.07 ms together with
    // one-wave workgroups, option `v1_block`)
    if (p.variant == 1 && !ctx->tile_w_set) p.tile_w = 8;
    p.list_cap = static_cast<int32_t>(ctx->opt_list_cap);
    p.coop = static_cast<int32_t>(ctx->opt_coop);
    p.bvh_prim_count = static_cast<int32_t>(ctx->host.bvh_prims.size());
    p.oct_prim_count = static_cast<int32_t>(ctx->host.oct_prims.size());
    p.blocks_per_cu = static_cast<int32_t>(ctx->opt_blocks_per_cu);
    p.num_cus = ctx->num_cus;
    p.refill_threshold = static_cast<int32_t>(ctx->opt_refill);
    p.hw_xcd = static_cast<int32_t>(ctx->opt_hw_xcd);
    p.item_px = static_cast<int32_t>(ctx->opt_item_px);
    p.tile_counters = ctx->d_counters ? ctx->d_counters + 64 * 64 * (ctx->counter_slot++ % 1024) : nullptr;
    p.stamps = ctx->d_stamps;
    for (int k = 0; k < 3; ++k) {
        p.pq_dim[k] = ctx->host.pq_dim[k];
        p.pq_origin[k] = ctx->host.pq_origin[k];
        p.pq_inv[k] = ctx->host.pq_inv[k];
        // cell size, rounded down a little: it scales a distance that must not be overstated
        p.pq_cell[k] = ctx->host.pq_inv[k] > 0.f ? static_cast<float>((1.0 / static_cast<double>(ctx->host.pq_inv[k])) * 0.9999) : 0.f;
    }
    p.pq_cell_count = static_cast<int32_t>(ctx->host.pq_cells.size());
    p.pq_list_count = static_cast<int32_t>(ctx->host.pq_list.size());
    p.use_grid = (ctx->opt_grid && !ctx->host.pq_cells.empty()) ? 1 : 0;
    p.pq_cells = ctx->dev.pq_cells;
    p.pq_list = ctx->dev.pq_list;
    p.bvh_leaves = ctx->dev.bvh_leaves;
    p.bvh_leaf_count = (ctx->opt_cull && ctx->host.accel == RM_ACCEL_BVH) ? ctx->dev.bvh_leaf_count : 0;
    p.nn_cells = ctx->dev.nn_cells;
    p.nn_list = ctx->dev.nn_list;
    p.nn_cell_count = static_cast<int32_t>(ctx->host.nn_cells.size());
    p.nn_list_count = static_cast<int32_t>(ctx->host.nn_list.size());
    const bool nn_on = ctx->opt_nn == 1 || (ctx->opt_nn == 2 && ctx->host.spheres.size() <= 512);
    p.use_nn = (p.use_grid && nn_on && !ctx->host.nn_cells.empty()) ? 1 : 0;
    p.leaf_order = ctx->host.leaf_order ? 1 : 0;
    p.rel_boxes = static_cast<int32_t>(ctx->opt_rel);
    p.n0_batch = static_cast<int32_t>(ctx->opt_n0_batch);
    p.v1_lists = static_cast<int32_t>(ctx->opt_v1_lists);
    p.v1_block = static_cast<int32_t>(ctx->opt_v1_block);
    p.oct_lean = static_cast<int32_t>(ctx->opt_oct_lean);  // the camera is finite (fill_params), so every march point is
    p.v1_list_offset = -1;
    p.prim_filter = (ctx->opt_filter && ctx->host.general && !ctx->host.program && ctx->host.prim_filter_ok &&
                     ctx->host.spheres.size() == ctx->host.prims.size()) ? 1 : 0;
    p.lds_budget_kb = static_cast<int32_t>(ctx->opt_lds_kb);
    p.lds_fill = static_cast<int32_t>(ctx->opt_lds_fill);
    p.item_wide = static_cast<int32_t>(ctx->opt_item_wide);
    p.multi_step = static_cast<int32_t>(ctx->opt_multi_step);
    p.uniform_radius = 0;
    if (ctx->opt_uniform && !ctx->host.general && ctx->host.spheres.size() >= 2 &&
        ctx->host.spheres.size() <= 256) {  // one radius, bit for bit; at most 256 spheres: the scan keys carry the id in eight bits
        const auto &sp = ctx->host.spheres;
        const auto &rd = ctx->host.radii;
        bool same = rd.size() == sp.size();
        for (size_t k = 1; same && k < sp.size(); ++k)
            same = std::memcmp(&sp[k].rf, &sp[0].rf, sizeof(float)) == 0 && std::memcmp(&rd[k], &rd[0], sizeof(double)) == 0;
        p.uniform_radius = same ? 1 : 0;
    }
    for (int k = 0; k < 3; ++k) {
        p.nn_dim[k] = ctx->host.nn_dim[k];
        p.nn_inv[k] = ctx->host.nn_inv[k];
    }
    p.spheres = ctx->dev.spheres;
    p.radii = ctx->dev.radii;
    p.bvh = ctx->dev.bvh;
    p.bvh_prims = ctx->dev.bvh_prims;
    p.oct = ctx->dev.oct;
    p.oct_prims = ctx->dev.oct_prims;
    p.oct_recs = ctx->opt_recs ? ctx->dev.oct_recs : nullptr;
    p.oct_lut = ctx->opt_lut ? ctx->dev.oct_lut : nullptr;
    p.oct_sub_hdr = (ctx->opt_sub && ctx->opt_recs) ? ctx->dev.oct_sub_hdr : nullptr;
    p.oct_sub_list = ctx->dev.oct_sub_list;
    return RM_OK;
}

int norm_shader(int s) { return (s >= RM_SHADE_NORMAL && s <= RM_SHADE_ITERATION_HEATMAP) ? s : RM_SHADE_NORMAL; }

}  // namespace

extern "C" {

const char *rm_version(void) { return "cpu-raymarcher_amd 0.1 (gfx950)"; }

int rm_create(int device, rm_ctx **out) {
    if (!out) return RM_E_INVALID;
    *out = nullptr;
    rm_ctx *ctx = new (std::nothrow) rm_ctx();
    if (!ctx) return RM_E_NOMEM;
    rmh::phong_light_dir(ctx->light);
    if (device >= 0) {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || device >= count) {
            delete ctx;
            return RM_E_NO_DEVICE;
        }
        e = hipSetDevice(device);
        if (e == hipSuccess) e = hipStreamCreate(&ctx->stream);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&ctx->d_diag), sizeof(RmDiagDevice));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&ctx->d_counters), 1024 * 64 * 64 * sizeof(unsigned int));
        if (e == hipSuccess) e = hipMemset(ctx->d_counters, 0, 1024 * 64 * 64 * sizeof(unsigned int));  // every launch leaves its heads zeroed (rm_diag.h)
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&ctx->d_diag_blocks), rm_ctx::kDiagBlocks * sizeof(RmDiagBlock));
        if (e == hipSuccess) e = hipMemset(ctx->d_diag_blocks, 0, rm_ctx::kDiagBlocks * sizeof(RmDiagBlock));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&ctx->d_stamps), (40 + 3 * 8192 + 2048 * 48) * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemset(ctx->d_stamps, 0, (40 + 3 * 8192 + 2048 * 48) * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemset(ctx->d_stamps + 6, 0xFF, sizeof(unsigned long long));
        if (e != hipSuccess) {
            delete ctx;
            return RM_E_HIP;
        }
        ctx->device = device;
        ctx->has_device = true;
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
            ctx->num_cus = cus;
    }
    *out = ctx;
    return RM_OK;
}

void rm_destroy(rm_ctx *ctx) {
    if (!ctx) return;
    if (ctx->has_device) {
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
        free_device_scene(ctx);
        (void)hipFree(ctx->scratch);
        (void)hipFree(ctx->d_diag);
        (void)hipFree(ctx->d_counters);
        (void)hipFree(ctx->d_diag_blocks);
        (void)hipFree(ctx->d_stamps);
        for (auto &t : ctx->tables) (void)hipFree(t.dev);
        (void)hipFree(ctx->d_lpt_cost);
        (void)hipFree(ctx->d_lpt_perm);
        for (hipEvent_t ev : ctx->lpt_done)
            if (ev) (void)hipEventDestroy(ev);
        for (auto &sl : ctx->oct_frames) {
            (void)hipFree(sl.dev);
            for (auto &u : sl.users) (void)hipEventDestroy(u.second);
            if (sl.built) (void)hipEventDestroy(sl.built);
        }
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

const char *rm_last_error(const rm_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
const char *rm_last_kernel(const rm_ctx *ctx) { return ctx && ctx->last_kernel ? ctx->last_kernel : ""; }

int rm_algorithm_from_string(const char *s) {
    if (!s) return RM_ALG_SPHERE_TRACER;
    if (!std::strcmp(s, "fixed-step")) return RM_ALG_FIXED_STEP;
    if (!std::strcmp(s, "adaptive-step")) return RM_ALG_ADAPTIVE_STEP;
    if (!std::strcmp(s, "adaptive-step-v2")) return RM_ALG_ADAPTIVE_STEP_V2;
    if (!std::strcmp(s, "adaptive-step-v3")) return RM_ALG_ADAPTIVE_STEP_V3;
    return RM_ALG_SPHERE_TRACER;  // 'sphere-tracer' and the default branch
}

int rm_accel_from_string(const char *s) {
    if (s && !std::strcmp(s, "Octree")) return RM_ACCEL_OCTREE;
    if (s && !std::strcmp(s, "BVH")) return RM_ACCEL_BVH;
    return RM_ACCEL_NONE;
}

int rm_shader_from_string(const char *s) {
    if (s && !std::strcmp(s, "phong")) return RM_SHADE_PHONG;
    if (s && !std::strcmp(s, "sdf-heatmap")) return RM_SHADE_SDF_HEATMAP;
    if (s && !std::strcmp(s, "iteration-heatmap")) return RM_SHADE_ITERATION_HEATMAP;
    return RM_SHADE_NORMAL;
}

int rm_preset_count(void) { return rmh::kPresetCount; }

int rm_scene_from_preset(rm_ctx *ctx, int32_t preset_index, int32_t accel) {
    if (!ctx) return RM_E_INVALID;
    if (preset_index == RM_SCENE_UPLOADED) return fail(ctx, RM_E_INVALID, "not a preset index");
    // force a rebuild only when something changed
    return ensure_scene(ctx, preset_index, accel);
}

int rm_scene_from_spheres(rm_ctx *ctx, const float *centers_xyz, const double *radii, int32_t n, int32_t accel) {
    if (!ctx) return RM_E_INVALID;
    if (n < 0 || (n > 0 && (!centers_xyz || !radii))) return fail(ctx, RM_E_INVALID, "bad sphere list");
    int rc = set_scene(ctx, centers_xyz, radii, n, norm_accel(accel), true, RM_SCENE_UPLOADED);
    if (rc) return rc;
    ctx->up_centers.assign(centers_xyz, centers_xyz + 3 * static_cast<size_t>(n));
    ctx->up_radii.assign(radii, radii + n);
    ctx->up_general = false;
    ctx->up_program = false;
    ctx->have_uploaded = true;
    return RM_OK;
}

int rm_scene_from_nodes(rm_ctx *ctx, const rm_node *nodes, int32_t n_nodes, const int32_t *roots, int32_t n_roots,
                        int32_t accel) {
    if (!ctx) return RM_E_INVALID;
    if (n_nodes < 0 || n_roots < 0 || (n_nodes > 0 && !nodes) || (n_roots > 0 && !roots))
        return fail(ctx, RM_E_INVALID, "bad node list");
    std::vector<rmh::NodeDesc> d(static_cast<size_t>(n_nodes));
    for (int i = 0; i < n_nodes; ++i) {
        d[i].type = nodes[i].type;
        d[i].a = nodes[i].child_a;
        d[i].b = nodes[i].child_b;
        std::memcpy(d[i].m, nodes[i].world_to_local, sizeof d[i].m);
        for (int k = 0; k < 6; ++k) d[i].params[k] = nodes[i].params[k];
    }
    std::vector<int> r(roots, roots + n_roots);
    int rc = set_scene_nodes(ctx, d.data(), n_nodes, r.data(), n_roots, norm_accel(accel), true, RM_SCENE_UPLOADED);
    if (rc) return rc;
    ctx->up_nodes = std::move(d);
    ctx->up_roots = std::move(r);
    ctx->up_program = true;
    ctx->up_general = false;
    ctx->have_uploaded = true;
    return RM_OK;
}

int rm_scale_transform(float *m, double x, double y, double z) {
    if (!m || !std::isfinite(x) || !std::isfinite(y) || !std::isfinite(z)) return RM_E_INVALID;
    rmh::scale_transform(m, x, y, z);
    return RM_OK;
}

int rm_scene_set_time(rm_ctx *ctx, double time) {
    if (!ctx) return RM_E_INVALID;
    if (!std::isfinite(time)) return fail(ctx, RM_E_INVALID, "non-finite time");
    ctx->time = time;
    return RM_OK;
}

int rm_scene_from_prims(rm_ctx *ctx, const rm_prim *prims, int32_t n, int32_t accel) {
    if (!ctx) return RM_E_INVALID;
    if (n < 0 || (n > 0 && !prims)) return fail(ctx, RM_E_INVALID, "bad primitive list");
    std::vector<rmh::PrimDesc> d(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
        d[i].type = prims[i].type;
        std::memcpy(d[i].m, prims[i].world_to_local, sizeof d[i].m);
        for (int k = 0; k < 3; ++k) d[i].params[k] = prims[i].params[k];
    }
    int rc = set_scene_general(ctx, d.data(), n, norm_accel(accel), true, RM_SCENE_UPLOADED);
    if (rc) return rc;
    ctx->up_prims = std::move(d);
    ctx->up_general = true;
    ctx->up_program = false;
    ctx->have_uploaded = true;
    return RM_OK;
}

int rm_make_transform(double x, double y, double z, const float *rotation_xyz, float *world_to_local16) {
    if (!world_to_local16 || !std::isfinite(x) || !std::isfinite(y) || !std::isfinite(z)) return RM_E_INVALID;
    if (rotation_xyz)
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(rotation_xyz[k])) return RM_E_INVALID;
    rmh::make_transform(x, y, z, rotation_xyz, world_to_local16);
    return RM_OK;
}

int rm_scene_get_info(const rm_ctx *ctx, rm_scene_info *out) {
    if (!ctx || !out) return RM_E_INVALID;
    if (!ctx->have_scene) return RM_E_NO_SCENE;
    std::memset(out, 0, sizeof *out);
    const rmh::HostScene &h = ctx->host;
    out->n_prims = scene_n_prims(ctx);
    out->accel = h.accel;
    out->preset_index = ctx->scene_preset;
    out->bvh_nodes = static_cast<int32_t>(h.bvh.size());
    out->bvh_leaves = h.bvh_leaves;
    out->bvh_depth = h.bvh_depth;
    out->oct_nodes = static_cast<int32_t>(h.oct.size());
    out->oct_leaves = h.oct_leaves;
    out->oct_empty_leaves = h.oct_empty;
    out->oct_max_leaf_prims = h.oct_max_leaf;
    std::memcpy(out->root_min, h.root_min, sizeof h.root_min);
    std::memcpy(out->root_max, h.root_max, sizeof h.root_max);
    out->nodes_in_lds = static_cast<int32_t>(ctx->opt_lds);
    return RM_OK;
}

int rm_camera_from_angles(double pitch, double yaw, float *rot9, float *origin3) {
    if (!rot9 || !origin3 || !std::isfinite(pitch) || !std::isfinite(yaw)) return RM_E_INVALID;
    rmh::camera_from_angles(pitch, yaw, rot9, origin3);
    return RM_OK;
}

int rm_render_tile_device(rm_ctx *ctx, const rm_job *job, int32_t shader, void *d_depth, void *d_normal, void *d_sdf,
                          void *d_iters, void *d_rgba, void *stream) {
    if (!ctx) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context: there is no CPU render path");
    RmRenderParams p;
    int rc = fill_params(ctx, job, p);
    if (rc) return rc;
    p.shader = norm_shader(shader);
    p.depth = static_cast<uint8_t *>(d_depth);
    p.normal = static_cast<uint8_t *>(d_normal);
    p.sdf = static_cast<uint16_t *>(d_sdf);
    p.iters = static_cast<uint16_t *>(d_iters);
    p.rgba = static_cast<uint8_t *>(d_rgba);
    RM_HIP(ctx, hipSetDevice(ctx->device));
    RM_HIP(ctx, launch_render(ctx, p, static_cast<hipStream_t>(stream)));
    return RM_OK;
}

int rm_render_attach_diagnostics(rm_ctx *ctx, void *d_acc) {
    if (!ctx) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    if (d_acc && (reinterpret_cast<uintptr_t>(d_acc) & 7)) return fail(ctx, RM_E_INVALID, "the accumulator must be 8-byte aligned");
    ctx->diag_next = d_acc;
    return RM_OK;
}

int rm_stripe_rows(int32_t y_start, int32_t y_end, int32_t stripe_rows, int32_t n_parts, int32_t part) {
    if (stripe_rows <= 0 || n_parts <= 0 || part < 0 || part >= n_parts) return RM_E_INVALID;
    const int64_t rows = y_end > y_start ? static_cast<int64_t>(y_end) - y_start : 0;
    int64_t mine = 0;
    for (int64_t s = part; s * stripe_rows < rows; s += n_parts) {
        const int64_t a = s * stripe_rows, b = a + stripe_rows;
        mine += (b < rows ? b : rows) - a;
    }
    return static_cast<int>(mine);
}

int rm_render_stripes_device(rm_ctx *ctx, const rm_job *job, int32_t shader, int32_t stripe_rows, int32_t n_parts,
                             int32_t part, void *d_depth, void *d_normal, void *d_sdf, void *d_iters, void *d_rgba,
                             void *stream) {
    if (!ctx) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context: there is no CPU render path");
    if (!job) return fail(ctx, RM_E_INVALID, "null job");
    const int mine = rm_stripe_rows(job->y_start, job->y_end, stripe_rows, n_parts, part);
    if (mine < 0) return fail(ctx, RM_E_INVALID, "bad stripe partition");
    RmRenderParams p;
    int rc = fill_params(ctx, job, p);
    if (rc) return rc;
    p.local_rows = mine;
    p.stripe_rows = stripe_rows;
    p.n_parts = n_parts;
    p.part = part;
    p.shader = norm_shader(shader);
    p.depth = static_cast<uint8_t *>(d_depth);
    p.normal = static_cast<uint8_t *>(d_normal);
    p.sdf = static_cast<uint16_t *>(d_sdf);
    p.iters = static_cast<uint16_t *>(d_iters);
    p.rgba = static_cast<uint8_t *>(d_rgba);
    RM_HIP(ctx, hipSetDevice(ctx->device));
    RM_HIP(ctx, launch_render(ctx, p, static_cast<hipStream_t>(stream)));
    return RM_OK;
}

int rm_deal_stripes(int32_t rows, int32_t stripe_rows, int32_t n_parts, const int32_t *weights, int32_t *owner) {
    if (rows < 0 || stripe_rows <= 0 || n_parts <= 0 || n_parts > 65535 || !owner) return RM_E_INVALID;
    int64_t total = 0;
    for (int p = 0; p < n_parts; ++p) {
        const int64_t w = weights ? weights[p] : 1;
        if (w <= 0 || w > (1 << 20)) return RM_E_INVALID;
        total += w;
    }
    const int n = static_cast<int>((static_cast<int64_t>(rows) + stripe_rows - 1) / stripe_rows);
    // smooth weighted round-robin: credit every part its weight, hand the stripe to the richest (lowest index on a
    // tie), charge it the total.  Equal weights: 0, 1, ..., n_parts - 1, 0, 1, ...
    std::vector<int64_t> credit(static_cast<size_t>(n_parts), 0);
    for (int s = 0; s < n; ++s) {
        int best = 0;
        for (int p = 0; p < n_parts; ++p) {
            credit[p] += weights ? weights[p] : 1;
            if (credit[p] > credit[best]) best = p;
        }
        credit[best] -= total;
        owner[s] = best;
    }
    return n;
}

int rm_render_stripe_list_device(rm_ctx *ctx, const rm_job *job, int32_t shader, int32_t stripe_rows, const int32_t *stripe_ids,
                                 int32_t n_stripes, void *d_depth, void *d_normal, void *d_sdf, void *d_iters, void *d_rgba,
                                 void *stream) {
    if (!ctx) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context: there is no CPU render path");
    if (!job) return fail(ctx, RM_E_INVALID, "null job");
    if (stripe_rows <= 0 || n_stripes < 0 || (n_stripes > 0 && !stripe_ids)) return fail(ctx, RM_E_INVALID, "bad stripe list");
    const int64_t rows = job->y_end > job->y_start ? static_cast<int64_t>(job->y_end) - job->y_start : 0;
    const int64_t total = (rows + stripe_rows - 1) / stripe_rows;
    int64_t mine = 0;
    for (int k = 0; k < n_stripes; ++k) {
        if (stripe_ids[k] < 0 || stripe_ids[k] >= total || (k > 0 && stripe_ids[k] <= stripe_ids[k - 1]))
            return fail(ctx, RM_E_INVALID, "stripe ids must be strictly increasing and inside the row range");
        const int64_t a = static_cast<int64_t>(stripe_ids[k]) * stripe_rows, b = a + stripe_rows;
        mine += (b < rows ? b : rows) - a;
    }
    RmRenderParams p;
    int rc = fill_params(ctx, job, p);
    if (rc) return rc;
    if (mine == 0) return RM_OK;
    RM_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = device_table(ctx, stripe_ids, static_cast<size_t>(n_stripes), &p.stripe_ids))) return rc;
    p.local_rows = static_cast<int32_t>(mine);
    p.stripe_rows = stripe_rows;
    p.n_parts = 1;
    p.part = 0;
    p.shader = norm_shader(shader);
    p.depth = static_cast<uint8_t *>(d_depth);
    p.normal = static_cast<uint8_t *>(d_normal);
    p.sdf = static_cast<uint16_t *>(d_sdf);
    p.iters = static_cast<uint16_t *>(d_iters);
    p.rgba = static_cast<uint8_t *>(d_rgba);
    RM_HIP(ctx, launch_render(ctx, p, static_cast<hipStream_t>(stream)));
    return RM_OK;
}

int rm_assemble_frame_device(rm_ctx *ctx, const void *d_gathered, int64_t rank_stride, int64_t section_offset, int32_t row_bytes,
                             int32_t height, int32_t stripe_rows, const int32_t *owner, int32_t n_stripes, int32_t world,
                             void *d_frame, int64_t acc_offset, void *d_acc, void *stream) {
    if (!ctx) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    if (!d_gathered || !d_frame || !owner) return fail(ctx, RM_E_INVALID, "null buffer");
    if (row_bytes <= 0 || height < 0 || stripe_rows <= 0 || world <= 0 || world > 65535 || rank_stride < 0 || section_offset < 0)
        return fail(ctx, RM_E_INVALID, "bad frame geometry");
    if (n_stripes != (height + stripe_rows - 1) / stripe_rows || n_stripes > 65535)
        return fail(ctx, RM_E_INVALID, "owner[] must name every stripe of the frame (at most 65535)");
    if (acc_offset >= 0 && (acc_offset % 8 != 0 || rank_stride % 8 != 0)) return fail(ctx, RM_E_INVALID, "accumulators must be 8-byte aligned");
    // frame stripe s -> (rank, position among that rank's stripes): the rank packed its stripes in increasing y
    std::vector<int32_t> src(static_cast<size_t>(n_stripes));
    std::vector<int32_t> seen(static_cast<size_t>(world), 0);
    for (int s = 0; s < n_stripes; ++s) {
        if (owner[s] < 0 || owner[s] >= world) return fail(ctx, RM_E_INVALID, "owner out of range");
        src[s] = (owner[s] << 16) | seen[owner[s]]++;
    }
    RM_HIP(ctx, hipSetDevice(ctx->device));
    const int32_t *d_src = nullptr;
    int rc = device_table(ctx, src.data(), src.size(), &d_src);
    if (rc) return rc;
    RM_HIP(ctx, rm_launch_assemble(static_cast<const unsigned char *>(d_gathered), rank_stride, section_offset, row_bytes, height,
                                   stripe_rows, d_src, n_stripes, static_cast<unsigned char *>(d_frame), acc_offset, world,
                                   static_cast<RmDiagDevice *>(d_acc), static_cast<hipStream_t>(stream)));
    return RM_OK;
}

int rm_render_tile(rm_ctx *ctx, const rm_job *job, uint8_t *depth, uint8_t *normal, uint16_t *sdf, uint16_t *iters) {
    if (!ctx) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context: there is no CPU render path");
    if (!job) return fail(ctx, RM_E_INVALID, "null job");
    if (!depth || !normal || !sdf || !iters) return fail(ctx, RM_E_INVALID, "null output buffer");
    const int64_t rows = job->y_end > job->y_start ? static_cast<int64_t>(job->y_end) - job->y_start : 0;
    const size_t npx = static_cast<size_t>(rows) * static_cast<size_t>(job->width > 0 ? job->width : 0);
    RM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t o_depth = 0, o_normal = align_up(npx, 256), o_sdf = o_normal + align_up(3 * npx, 256),
                 o_iters = o_sdf + align_up(2 * npx, 256), total = o_iters + align_up(2 * npx, 256);
    int rc = ensure_scratch(ctx, total ? total : 256);
    if (rc) return rc;
    char *base = static_cast<char *>(ctx->scratch);
    rc = rm_render_tile_device(ctx, job, RM_SHADE_NORMAL, base + o_depth, base + o_normal, base + o_sdf,
                               base + o_iters, nullptr, ctx->stream);
    if (rc) return rc;
    if (npx) {
        RM_HIP(ctx, hipMemcpyAsync(depth, base + o_depth, npx, hipMemcpyDeviceToHost, ctx->stream));
        RM_HIP(ctx, hipMemcpyAsync(normal, base + o_normal, 3 * npx, hipMemcpyDeviceToHost, ctx->stream));
        RM_HIP(ctx, hipMemcpyAsync(sdf, base + o_sdf, 2 * npx, hipMemcpyDeviceToHost, ctx->stream));
        RM_HIP(ctx, hipMemcpyAsync(iters, base + o_iters, 2 * npx, hipMemcpyDeviceToHost, ctx->stream));
    }
    RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RM_OK;
}

int rm_shade_device(rm_ctx *ctx, int32_t shader, int32_t width, int32_t height, const void *d_depth,
                    const void *d_normal, const void *d_sdf, const void *d_iters, void *d_rgba, void *stream) {
    if (!ctx) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    if (width < 0 || height < 0) return fail(ctx, RM_E_INVALID, "negative size");
    if (!d_depth || !d_normal || !d_sdf || !d_iters || !d_rgba) return fail(ctx, RM_E_INVALID, "null buffer");
    RM_HIP(ctx, hipSetDevice(ctx->device));
    RM_HIP(ctx, rm_launch_shade(norm_shader(shader), static_cast<int64_t>(width) * height,
                                static_cast<const uint8_t *>(d_depth), static_cast<const uint8_t *>(d_normal),
                                static_cast<const uint16_t *>(d_sdf), static_cast<const uint16_t *>(d_iters),
                                static_cast<uint8_t *>(d_rgba), ctx->light, static_cast<hipStream_t>(stream)));
    return RM_OK;
}

int rm_shade(rm_ctx *ctx, int32_t shader, int32_t width, int32_t height, const uint8_t *depth, const uint8_t *normal,
             const uint16_t *sdf, const uint16_t *iters, uint8_t *rgba) {
    if (!ctx) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    if (width < 0 || height < 0) return fail(ctx, RM_E_INVALID, "negative size");
    if (!depth || !normal || !sdf || !iters || !rgba) return fail(ctx, RM_E_INVALID, "null buffer");
    const size_t npx = static_cast<size_t>(width) * static_cast<size_t>(height);
    if (!npx) return RM_OK;
    RM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t o_depth = 0, o_normal = align_up(npx, 256), o_sdf = o_normal + align_up(3 * npx, 256),
                 o_iters = o_sdf + align_up(2 * npx, 256), o_rgba = o_iters + align_up(2 * npx, 256),
                 total = o_rgba + align_up(4 * npx, 256);
    int rc = ensure_scratch(ctx, total);
    if (rc) return rc;
    char *base = static_cast<char *>(ctx->scratch);
    RM_HIP(ctx, hipMemcpyAsync(base + o_depth, depth, npx, hipMemcpyHostToDevice, ctx->stream));
    RM_HIP(ctx, hipMemcpyAsync(base + o_normal, normal, 3 * npx, hipMemcpyHostToDevice, ctx->stream));
    RM_HIP(ctx, hipMemcpyAsync(base + o_sdf, sdf, 2 * npx, hipMemcpyHostToDevice, ctx->stream));
    RM_HIP(ctx, hipMemcpyAsync(base + o_iters, iters, 2 * npx, hipMemcpyHostToDevice, ctx->stream));
    rc = rm_shade_device(ctx, shader, width, height, base + o_depth, base + o_normal, base + o_sdf, base + o_iters,
                         base + o_rgba, ctx->stream);
    if (rc) return rc;
    RM_HIP(ctx, hipMemcpyAsync(rgba, base + o_rgba, 4 * npx, hipMemcpyDeviceToHost, ctx->stream));
    RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RM_OK;
}

int rm_reduce_counters_enqueue(rm_ctx *ctx, const void *d_sdf, const void *d_iters, int64_t n, void *d_acc,
                               void *stream) {
    if (!ctx || !d_acc) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    if (n < 0 || (n > 0 && (!d_sdf || !d_iters))) return fail(ctx, RM_E_INVALID, "bad buffers");
    RM_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    RM_HIP(ctx, rm_launch_reduce_init(static_cast<RmDiagDevice *>(d_acc), st));
    RM_HIP(ctx, rm_launch_reduce(static_cast<const uint16_t *>(d_sdf), static_cast<const uint16_t *>(d_iters), n,
                                 static_cast<RmDiagDevice *>(d_acc), st));
    return RM_OK;
}

int rm_reduce_counters_device(rm_ctx *ctx, const void *d_sdf, const void *d_iters, int64_t n, rm_diagnostics *out,
                              void *stream) {
    if (!ctx || !out) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc0 = rm_reduce_counters_enqueue(ctx, d_sdf, d_iters, n, ctx->d_diag, stream);
    if (rc0) return rc0;
    RmDiagDevice res;
    RM_HIP(ctx, hipMemcpyAsync(&res, ctx->d_diag, sizeof res, hipMemcpyDeviceToHost, st));
    RM_HIP(ctx, hipStreamSynchronize(st));
    out->total_sdf_calls = res.total_sdf;
    out->total_iterations = res.total_iters;
    out->max_sdf_calls = res.max_sdf;
    out->min_sdf_calls = res.min_sdf;
    out->total_pixels = static_cast<uint64_t>(n);
    return RM_OK;
}

int rm_reduce_counters(rm_ctx *ctx, const uint16_t *sdf, const uint16_t *iters, int64_t n, rm_diagnostics *out) {
    if (!ctx || !out) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    if (n < 0 || (n > 0 && (!sdf || !iters))) return fail(ctx, RM_E_INVALID, "bad buffers");
    RM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = 2 * static_cast<size_t>(n);
    const size_t o_iters = align_up(bytes, 256);
    int rc = ensure_scratch(ctx, o_iters + align_up(bytes, 256) + 256);
    if (rc) return rc;
    char *base = static_cast<char *>(ctx->scratch);
    if (n) {
        RM_HIP(ctx, hipMemcpyAsync(base, sdf, bytes, hipMemcpyHostToDevice, ctx->stream));
        RM_HIP(ctx, hipMemcpyAsync(base + o_iters, iters, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    return rm_reduce_counters_device(ctx, base, base + o_iters, n, out, ctx->stream);
}

int rm_partition_rows(int32_t height, int32_t n_workers, int32_t i, int32_t *y_start, int32_t *y_end) {
    if (!y_start || !y_end || n_workers <= 0 || height < 0 || i < 0) return RM_E_INVALID;
    // main.ts:444-449
    const int64_t rows = (static_cast<int64_t>(height) + n_workers - 1) / n_workers;
    const int64_t a = static_cast<int64_t>(i) * rows, b = (static_cast<int64_t>(i) + 1) * rows;
    *y_start = static_cast<int32_t>(a < height ? a : height);
    *y_end = static_cast<int32_t>(b < height ? b : height);
    return RM_OK;
}

int rm_scene_distance(rm_ctx *ctx, const float *points_xyz, int64_t n, double *dist, uint32_t *count) {
    if (!ctx) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    if (!ctx->have_scene) return fail(ctx, RM_E_NO_SCENE, "no scene set");
    if (n < 0 || (n > 0 && (!points_xyz || !dist || !count))) return fail(ctx, RM_E_INVALID, "bad buffers");
    if (!n) return RM_OK;
    for (int64_t i = 0; i < 3 * n; ++i)
        if (!std::isfinite(points_xyz[i])) return fail(ctx, RM_E_INVALID, "non-finite point");
    RM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t o_dist = align_up(12 * static_cast<size_t>(n), 256), o_cnt = o_dist + align_up(8 * static_cast<size_t>(n), 256);
    int rc = ensure_scratch(ctx, o_cnt + align_up(4 * static_cast<size_t>(n), 256));
    if (rc) return rc;
    char *base = static_cast<char *>(ctx->scratch);
    RmRenderParams p;
    std::memset(&p, 0, sizeof p);
    p.n_prims = scene_n_prims(ctx);
    fill_scene_repr(ctx, p);
    p.time = ctx->time;
    p.accel = ctx->host.accel;
    p.bvh_nodes = static_cast<int32_t>(ctx->host.bvh.size());
    p.oct_nodes = static_cast<int32_t>(ctx->host.oct.size());
    p.filter = static_cast<int32_t>(ctx->opt_filter);
    p.tile_w = 8;
    p.spheres = ctx->dev.spheres;
    p.radii = ctx->dev.radii;
    p.bvh = ctx->dev.bvh;
    p.bvh_prims = ctx->dev.bvh_prims;
    p.oct = ctx->dev.oct;
    p.oct_prims = ctx->dev.oct_prims;
    p.oct_lut = ctx->opt_lut ? ctx->dev.oct_lut : nullptr;
    {
        const rmrtc::Kernel *special = specialised_kernel(ctx, p.accel, false);
        p.rtc_function = special ? special->distance : nullptr;
    }
    RM_HIP(ctx, hipMemcpyAsync(base, points_xyz, 12 * static_cast<size_t>(n), hipMemcpyHostToDevice, ctx->stream));
    RM_HIP(ctx, (ctx->opt_length ? rm_launch_distance_sqrt : rm_launch_distance)(
                    p, reinterpret_cast<const float *>(base), n, reinterpret_cast<double *>(base + o_dist),
                    reinterpret_cast<uint32_t *>(base + o_cnt), ctx->stream));
    RM_HIP(ctx, hipMemcpyAsync(dist, base + o_dist, 8 * static_cast<size_t>(n), hipMemcpyDeviceToHost, ctx->stream));
    RM_HIP(ctx, hipMemcpyAsync(count, base + o_cnt, 4 * static_cast<size_t>(n), hipMemcpyDeviceToHost, ctx->stream));
    RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RM_OK;
}

int rm_selftest_hypot(rm_ctx *ctx, const float *xyz, int64_t n, double *out) {
    if (!ctx) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    if (n < 0 || (n > 0 && (!xyz || !out))) return fail(ctx, RM_E_INVALID, "bad buffers");
    if (!n) return RM_OK;
    RM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t o_out = align_up(12 * static_cast<size_t>(n), 256);
    int rc = ensure_scratch(ctx, o_out + 8 * static_cast<size_t>(n));
    if (rc) return rc;
    char *base = static_cast<char *>(ctx->scratch);
    RM_HIP(ctx, hipMemcpyAsync(base, xyz, 12 * static_cast<size_t>(n), hipMemcpyHostToDevice, ctx->stream));
    RM_HIP(ctx, rm_launch_hypot(reinterpret_cast<const float *>(base), n, reinterpret_cast<double *>(base + o_out),
                                ctx->stream));
    RM_HIP(ctx, hipMemcpyAsync(out, base + o_out, 8 * static_cast<size_t>(n), hipMemcpyDeviceToHost, ctx->stream));
    RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RM_OK;
}

int rm_selftest_jsmath(rm_ctx *ctx, int32_t fn, const double *a, const double *b, int64_t n, double *out) {
    if (!ctx) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    if (n < 0 || fn < 0 || fn > 7 || (n > 0 && (!a || !out))) return fail(ctx, RM_E_INVALID, "bad buffers");
    if (!n) return RM_OK;
    RM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t stride = align_up(8 * static_cast<size_t>(n), 256);
    int rc = ensure_scratch(ctx, 3 * stride);
    if (rc) return rc;
    char *base = static_cast<char *>(ctx->scratch);
    RM_HIP(ctx, hipMemcpyAsync(base, a, 8 * static_cast<size_t>(n), hipMemcpyHostToDevice, ctx->stream));
    if (b) {
        RM_HIP(ctx, hipMemcpyAsync(base + stride, b, 8 * static_cast<size_t>(n), hipMemcpyHostToDevice, ctx->stream));
    } else {
        RM_HIP(ctx, hipMemsetAsync(base + stride, 0, 8 * static_cast<size_t>(n), ctx->stream));
    }
    RM_HIP(ctx, rm_launch_jsmath(fn, reinterpret_cast<const double *>(base), reinterpret_cast<const double *>(base + stride),
                                 n, reinterpret_cast<double *>(base + 2 * stride), ctx->stream));
    RM_HIP(ctx, hipMemcpyAsync(out, base + 2 * stride, 8 * static_cast<size_t>(n), hipMemcpyDeviceToHost, ctx->stream));
    RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RM_OK;
}

int rm_selftest_fastdiv(rm_ctx *ctx, uint64_t seed, int64_t n, uint64_t *mismatches) {
    if (!ctx || !mismatches) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    if (n < 0) return fail(ctx, RM_E_INVALID, "negative count");
    RM_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_scratch(ctx, 256);
    if (rc) return rc;
    unsigned long long *d = static_cast<unsigned long long *>(ctx->scratch);
    RM_HIP(ctx, hipMemsetAsync(d, 0, sizeof *d, ctx->stream));
    RM_HIP(ctx, rm_launch_fastdiv_selftest(seed, n, d, ctx->stream));
    unsigned long long h = 0;
    RM_HIP(ctx, hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *mismatches = h;
    return RM_OK;
}

int rm_selftest_recip(rm_ctx *ctx, int mode, uint64_t *mismatches) {
    if (!ctx || !mismatches) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    if (mode != 0 && mode != 1) return fail(ctx, RM_E_INVALID, "mode must be 0 or 1");
    RM_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_scratch(ctx, 256);
    if (rc) return rc;
    unsigned long long *d = static_cast<unsigned long long *>(ctx->scratch);
    RM_HIP(ctx, hipMemsetAsync(d, 0, sizeof *d, ctx->stream));
    RM_HIP(ctx, rm_launch_recip_selftest(mode, d, ctx->stream));
    unsigned long long h = 0;
    RM_HIP(ctx, hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *mismatches = h;
    return RM_OK;
}

int rm_debug_read_stamps(rm_ctx *ctx, uint64_t *out8) {
    if (!ctx || !out8) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    RM_HIP(ctx, hipSetDevice(ctx->device));
    RM_HIP(ctx, hipDeviceSynchronize());
    RM_HIP(ctx, hipMemcpy(out8, ctx->d_stamps, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    RM_HIP(ctx, hipMemset(ctx->d_stamps, 0, 8 * sizeof(uint64_t)));
    RM_HIP(ctx, hipMemset(ctx->d_stamps + 6, 0xFF, sizeof(uint64_t)));  // slot 6: minimum start time of the next launch's waves
    return RM_OK;
}

// -DRM_STAMPS builds: start / end time (s_memrealtime, 100 MHz) of the first 8192 waves of the last v2 launch
int rm_debug_read_wave_times(rm_ctx *ctx, uint64_t *out16384) {
    if (!ctx || !out16384) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    RM_HIP(ctx, hipSetDevice(ctx->device));
    RM_HIP(ctx, hipDeviceSynchronize());
    RM_HIP(ctx, hipMemcpy(out16384, ctx->d_stamps + 40, 3 * 8192 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    RM_HIP(ctx, hipMemset(ctx->d_stamps + 40, 0, 3 * 8192 * sizeof(uint64_t)));
    return RM_OK;
}

// -DRM_STAMPS -DRM_STAMPS_LOG builds: the low 32 bits of s_memrealtime at the start of the first 96 batches of the first 2048 waves
int rm_debug_read_batch_log(rm_ctx *ctx, uint32_t *out196608) {
    if (!ctx || !out196608) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    RM_HIP(ctx, hipSetDevice(ctx->device));
    RM_HIP(ctx, hipDeviceSynchronize());
    RM_HIP(ctx, hipMemcpy(out196608, ctx->d_stamps + 40 + 3 * 8192, 2048 * 96 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    RM_HIP(ctx, hipMemset(ctx->d_stamps + 40 + 3 * 8192, 0, 2048 * 96 * sizeof(uint32_t)));
    return RM_OK;
}

// ---- run-time specialisation of expression forests (rm_rtc.h) -------------------------------------------------------------
static int copy_text(const std::string &t, char *out, int64_t cap, int64_t *needed) {
    if (needed) *needed = static_cast<int64_t>(t.size()) + 1;
    if (out && cap > 0) {
        const size_t n = std::min(static_cast<size_t>(cap - 1), t.size());
        std::memcpy(out, t.data(), n);
        out[n] = '\0';
    }
    return RM_OK;
}

int rm_rtc_source(rm_ctx *ctx, char *out, int64_t cap, int64_t *needed) {
    if (!ctx) return RM_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, RM_E_NO_SCENE, "no scene");
    return copy_text(ctx->rtc_src, out, cap, needed);
}

int rm_rtc_compile_check(rm_ctx *ctx, int32_t accel, int32_t other, char *log, int64_t cap, double *seconds) {
    if (!ctx) return RM_E_INVALID;
    if (!ctx->have_scene || ctx->rtc_src.empty()) return fail(ctx, RM_E_NO_SCENE, "the active scene has no specialised source");
    rmrtc::Kernel k;
    std::string text;
    const bool ok = rmrtc::compile(ctx->rtc_src, norm_accel(accel), other != 0, ctx->opt_length != 0, false, true, k, text);
    if (seconds) *seconds = k.compile_seconds;
    copy_text(text, log, cap, nullptr);
    return ok ? RM_OK : fail(ctx, RM_E_INVALID, "hiprtc: the specialised kernel did not compile (see the log)");
}

int rm_rtc_status(rm_ctx *ctx, int32_t *compiled, int32_t *failed, char *log, int64_t cap) {
    if (!ctx) return RM_E_INVALID;
    int v2_ready = 0, v2_failed = 0;
    for (const auto &kv : ctx->v2_special) {
        v2_ready += kv.second.state == 1;
        v2_failed += kv.second.state < 0;
    }
    if (compiled) *compiled = static_cast<int32_t>(ctx->rtc_kernels.size()) + v2_ready;
    if (failed) *failed = static_cast<int32_t>(ctx->rtc_failed.size()) + v2_failed;
    std::string why;
    if (!rmrtc::available(&why)) return copy_text(why, log, cap, nullptr);
    return copy_text(ctx->rtc_log, log, cap, nullptr);
}

// the item costs (durations in units of 2.56 us, one byte per item, 64 queues x kLptStride slots) the LAST v2 launch recorded
int rm_debug_read_lpt_costs(rm_ctx *ctx, uint8_t *out, int64_t n) {
    if (!ctx || !out) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    const size_t per_slot = static_cast<size_t>(rm_ctx::kLptQueues) * rm_ctx::kLptStride;
    if (!ctx->d_lpt_cost || ctx->lpt_launch == 0 || n < 0 || static_cast<size_t>(n) > per_slot) return fail(ctx, RM_E_INVALID, "no costs recorded");
    RM_HIP(ctx, hipSetDevice(ctx->device));
    RM_HIP(ctx, hipDeviceSynchronize());
    const unsigned slot = (ctx->lpt_launch + rm_ctx::kLptSlots - 1) % rm_ctx::kLptSlots;
    RM_HIP(ctx, hipMemcpy(out, ctx->d_lpt_cost + slot * per_slot, static_cast<size_t>(n), hipMemcpyDeviceToHost));
    return RM_OK;
}

int rm_debug_read_counts(rm_ctx *ctx, uint64_t *out32) {
    if (!ctx || !out32) return RM_E_INVALID;
    if (!ctx->has_device) return fail(ctx, RM_E_NO_DEVICE, "host-only context");
    RM_HIP(ctx, hipSetDevice(ctx->device));
    RM_HIP(ctx, hipDeviceSynchronize());
    RM_HIP(ctx, hipMemcpy(out32, ctx->d_stamps + 8, 32 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    RM_HIP(ctx, hipMemset(ctx->d_stamps + 8, 0, 32 * sizeof(uint64_t)));
    return RM_OK;
}

int rm_set_option(rm_ctx *ctx, const char *key, int64_t value) {
    if (!ctx || !key) return RM_E_INVALID;
    if (!std::strcmp(key, "tile_w")) {
        if (value != 8 && value != 16 && value != 32 && value != 64) return fail(ctx, RM_E_INVALID, "tile_w must be 8, 16, 32 or 64");
        ctx->opt_tile_w = value;
        ctx->tile_w_set = true;
        return RM_OK;
    }
    if (!std::strcmp(key, "filter")) {
        ctx->opt_filter = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "nodes_in_lds")) {
        ctx->opt_lds = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "kernel")) {
        if (value < 0 || value > 2) return fail(ctx, RM_E_INVALID, "kernel must be 0 (auto), 1 or 2");
        ctx->opt_kernel = value;
        return RM_OK;
    }
    if (!std::strcmp(key, "list_cap")) {
        if (value < 1 || value > 64) return fail(ctx, RM_E_INVALID, "list_cap must be in [1, 64]");
        ctx->opt_list_cap = value;
        return RM_OK;
    }
    if (!std::strcmp(key, "coop")) {
        ctx->opt_coop = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "recs")) {
        ctx->opt_recs = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "lut")) {
        ctx->opt_lut = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "uniform")) {
        ctx->opt_uniform = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "lds_kb")) {
        if (value != 0 && (value < 16 || value > 64)) return fail(ctx, RM_E_INVALID, "lds_kb must be 0 (auto) or in [16, 64]");
        ctx->opt_lds_kb = value;
        return RM_OK;
    }
    if (!std::strcmp(key, "multi_step")) {
        ctx->opt_multi_step = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "item_wide")) {
        ctx->opt_item_wide = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "prune")) {
        ctx->opt_prune = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "rtc_spheres")) {
        if (value < 0 || value > 33) return fail(ctx, RM_E_INVALID, "rtc_spheres must be 0..33");
        ctx->opt_rtc_spheres = value;
        return RM_OK;
    }
    if (!std::strcmp(key, "specialise_v2_after")) {
        if (value < 0 || value > 1000000) return fail(ctx, RM_E_INVALID, "specialise_v2_after must be 0 (never) .. 1000000 launches");
        ctx->opt_v2_after = value;
        return RM_OK;
    }
    if (!std::strcmp(key, "specialise")) {
        if (value < 0 || value > 2) return fail(ctx, RM_E_INVALID, "specialise must be 0, 1 or 2");
        ctx->opt_specialise = value;
        return RM_OK;
    }
    if (!std::strcmp(key, "lds_fill")) {
        ctx->opt_lds_fill = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "cull")) {
        ctx->opt_cull = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "rel")) {
        ctx->opt_rel = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "static")) {
        if (value < 0 || value > 95) return fail(ctx, RM_E_INVALID, "static must be in [0, 95] percent");
        ctx->opt_static = value;
        return RM_OK;
    }
    if (!std::strcmp(key, "sub")) {
        ctx->opt_sub = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "refill")) {
        if (value < 1 || value > 64) return fail(ctx, RM_E_INVALID, "refill must be in [1, 64]");
        ctx->opt_refill = value;
        return RM_OK;
    }
    if (!std::strcmp(key, "item_px")) {
        if (value != 64 && value != 128 && value != 256) return fail(ctx, RM_E_INVALID, "item_px must be 64, 128 or 256");
        ctx->opt_item_px = value;
        return RM_OK;
    }
    if (!std::strcmp(key, "hw_xcd")) {
        ctx->opt_hw_xcd = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "grid")) {
        ctx->opt_grid = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "nn")) {
        if (value < 0 || value > 2) return fail(ctx, RM_E_INVALID, "nn must be 0 (off), 1 (on) or 2 (auto)");
        ctx->opt_nn = value;
        return RM_OK;
    }
    if (!std::strcmp(key, "blocks_per_cu")) {
        if (value < 1 || value > 8) return fail(ctx, RM_E_INVALID, "blocks_per_cu must be in [1, 8]");
        ctx->opt_blocks_per_cu = value;
        return RM_OK;
    }
    if (!std::strcmp(key, "v1_lists")) {
        ctx->opt_v1_lists = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "v1_block")) {
        if (value != 64 && value != 128 && value != 256) return fail(ctx, RM_E_INVALID, "v1_block must be 64, 128 or 256");
        ctx->opt_v1_block = value;
        return RM_OK;
    }
    if (!std::strcmp(key, "oct_lean")) {
        ctx->opt_oct_lean = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "lpt")) {
        ctx->opt_lpt = value ? 1 : 0;
        return RM_OK;
    }
    if (!std::strcmp(key, "n0_batch")) {
        if (value < 1 || value > 64) return fail(ctx, RM_E_INVALID, "n0_batch must be in [1, 64]");
        ctx->opt_n0_batch = value;
        return RM_OK;
    }
    if (!std::strcmp(key, "length")) {  // part of the numeric contract, not a measurement knob (rm_raymarch.h)
        if (value != 0 && value != 1) return fail(ctx, RM_E_INVALID, "length must be 0 (Math.hypot) or 1 (Math.sqrt)");
        if (ctx->opt_length == value) return RM_OK;
        ctx->opt_length = value;
        if (!ctx->have_scene) return RM_OK;
        // bounding radii of boxes and smooth unions use vec3.length / vec3.distance too: rebuild the active scene
        ctx->have_scene = false;
        return ensure_scene(ctx, ctx->scene_is_uploaded ? RM_SCENE_UPLOADED : ctx->scene_preset, ctx->host.accel);
    }
    return fail(ctx, RM_E_INVALID, std::string("unknown option ") + key);
}

int rm_get_option(const rm_ctx *ctx, const char *key, int64_t *value) {
    if (!ctx || !key || !value) return RM_E_INVALID;
    if (!std::strcmp(key, "tile_w")) *value = ctx->opt_tile_w;
    else if (!std::strcmp(key, "filter")) *value = ctx->opt_filter;
    else if (!std::strcmp(key, "nodes_in_lds")) *value = ctx->opt_lds;
    else if (!std::strcmp(key, "kernel")) *value = ctx->opt_kernel;
    else if (!std::strcmp(key, "list_cap")) *value = ctx->opt_list_cap;
    else if (!std::strcmp(key, "coop")) *value = ctx->opt_coop;
    else if (!std::strcmp(key, "grid")) *value = ctx->opt_grid;
    else if (!std::strcmp(key, "nn")) *value = ctx->opt_nn;
    else if (!std::strcmp(key, "refill")) *value = ctx->opt_refill;
    else if (!std::strcmp(key, "recs")) *value = ctx->opt_recs;
    else if (!std::strcmp(key, "lut")) *value = ctx->opt_lut;
    else if (!std::strcmp(key, "sub")) *value = ctx->opt_sub;
    else if (!std::strcmp(key, "static")) *value = ctx->opt_static;
    else if (!std::strcmp(key, "uniform")) *value = ctx->opt_uniform;
    else if (!std::strcmp(key, "rel")) *value = ctx->opt_rel;
    else if (!std::strcmp(key, "cull")) *value = ctx->opt_cull;
    else if (!std::strcmp(key, "lds_kb")) *value = ctx->opt_lds_kb;
    else if (!std::strcmp(key, "lds_fill")) *value = ctx->opt_lds_fill;
    else if (!std::strcmp(key, "specialise")) *value = ctx->opt_specialise;
    else if (!std::strcmp(key, "specialise_v2_after")) *value = ctx->opt_v2_after;
    else if (!std::strcmp(key, "rtc_spheres")) *value = ctx->opt_rtc_spheres;
    else if (!std::strcmp(key, "prune")) *value = ctx->opt_prune;
    else if (!std::strcmp(key, "item_wide")) *value = ctx->opt_item_wide;
    else if (!std::strcmp(key, "multi_step")) *value = ctx->opt_multi_step;
    else if (!std::strcmp(key, "hw_xcd")) *value = ctx->opt_hw_xcd;
    else if (!std::strcmp(key, "item_px")) *value = ctx->opt_item_px;
    else if (!std::strcmp(key, "blocks_per_cu")) *value = ctx->opt_blocks_per_cu;
    else if (!std::strcmp(key, "length")) *value = ctx->opt_length;
    else if (!std::strcmp(key, "n0_batch")) *value = ctx->opt_n0_batch;
    else if (!std::strcmp(key, "lpt")) *value = ctx->opt_lpt;
    else if (!std::strcmp(key, "v1_lists")) *value = ctx->opt_v1_lists;
    else if (!std::strcmp(key, "oct_lean")) *value = ctx->opt_oct_lean;
    else if (!std::strcmp(key, "v1_block")) *value = ctx->opt_v1_block;
    else return RM_E_INVALID;
    return RM_OK;
}

}  // extern "C"
