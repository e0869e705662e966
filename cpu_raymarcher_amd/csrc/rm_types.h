// rm_types.h -- device-resident data layout shared by the host builder and the HIP kernels.
//
// Layout in HBM (all read-only during a render, replicated per GPU):
//   spheres   : RmSphere[n]     16 B  centre f32x3 + f32(radius) (radius copy feeds the
//                                      conservative f32 candidate filter only)
//   radii     : double[n]        8 B  Sphere.radius stays a JS double (sphere.ts:5-9)
//   bvh       : RmBvhNode[nodes] 32 B  right-first pre-order with skip links (see below)
//   bvh_prims : int32[...]             leaf primitive ids, leaf order
//   oct       : RmOctNode[nodes] 48 B  children of a node are 8 consecutive entries
//   oct_prims : int32[...]             leaf primitive ids
// Dense Sphere Grid: 125 spheres (2 KB + 1 KB), 127 BVH nodes (4 KB).  10k synthetic:
// 160 KB + 80 KB spheres, 2441 octree nodes (117 KB) + ~1.2 MB of leaf ids.
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#else  // hiprtc keeps the fixed-width names in a namespace of its own
using __hip_internal::int8_t;
using __hip_internal::int16_t;
using __hip_internal::int32_t;
using __hip_internal::int64_t;
using __hip_internal::uint8_t;
using __hip_internal::uint16_t;
using __hip_internal::uint32_t;
using __hip_internal::uint64_t;
#endif

// One BVH node (bvh.ts:6-22), flattened.  Nodes are stored in the order in which
// BVH.findRayIntersections (bvh.ts:136-173) pops them: it pushes left then right, so the
// RIGHT child is visited first.  In that order the first child of node i is i + 1 and
// `skip` is the next node to visit once the subtree of i is finished or pruned; a
// traversal is `i = descend ? i + 1 : skip` with no stack, and the node index doubles as
// the traversal sequence number that breaks ties in the stable sort of bvh.ts:176.
struct RmBvhNode {
    float lo[3];
    int32_t skip;  // == node count for "end of traversal"
    float hi[3];
    int32_t leaf;  // -1: internal node; else (first << 8) | count into bvh_prims
};

// One octree node (octree.ts:6-26), flattened.
struct RmOctNode {
    float lo[3];
    int32_t first_child;  // -1: leaf; else index of child 0 (children are consecutive,
                          // index = x + 2y + 4z, octree.ts:69-88)
    float hi[3];
    int32_t prim_first;   // into oct_prims
    double min_distance;  // octree.ts:149-191 (0 unless isEmpty)
    int32_t prim_count;
    int32_t is_empty;
    float center[3];      // f32 split point (boundingBox.ts:108-114) = children[0].hi; internal nodes
    int32_t sub_first;    // leaves with many spheres: first of RM_OCT_SUB^3 sub-cell headers in oct_sub_hdr, else -1
};

#define RM_OCT_SUB 12  // sub-cells per axis of a crowded octree leaf

// One octree node relative to a camera position (render_kernel_oct): the operands of intersectRayBox's
// (bounds - origin) * invDir (octree.ts:200-203) and marchRay's minDistance * 0.99 (octree.ts:282), 64 B like RmOctNode so
// one byte offset addresses both.
struct RmOctFrameNode {
    double lo[3];  // (double)lo - origin
    double cap;    // min_distance * 0.99
    double hi[3];  // (double)hi - origin
    double pad;
};

struct RmSphere {
    float cx, cy, cz;
    float rf;  // (float)radius, used only by the conservative candidate filter
};

// Octree leaf lists, sphere scenes: the leaf's spheres copied out in leaf order (centre, f32 radius for
// the filter, the double radius), so a leaf is one contiguous stream of 32-B records instead of the
// dependent chain id -> sphere -> radius (10k-sphere scene: 0.3 M records, 9.6 MB, L2/MALL resident).
struct RmSphereRec {
    float cx, cy, cz, rf;
    double radius;
    int32_t id, pad;
};

// General primitive: world->local matrix as gl-matrix stores it (column-major
// Float32Array) + the local SDF's parameters.  Scenes that contain anything but unrotated
// spheres use these records (v1 kernel); pure sphere scenes keep the compact RmSphere path.
struct RmPrim {
    float m[16];
    int32_t type;   // low byte: 0 sphere, 1 box, 2 torus; bit 8: m has the bottom row (0,0,0,1) (no w divide);
                    // bit 9: m is a pure translation
    float half[3];  // box: halfSize (Float32Array, box.ts:8-11)
    double a, b;    // sphere: radius, -; torus: majorRadius, minorRadius
};

// Expression trees (primitive_operations/*.ts, mandelbulb.ts) are compiled by the host
// into a linear program per scene object, evaluated by every lane with a small position-slot file
// and a value stack (rm_program.h).  An operator node becomes a PRE instruction (computes the point
// its operands are evaluated at, into slot `dst`), the operands' instructions, and -- for Round and
// the smooth unions -- a POST instruction that combines the values on the stack.
//   op: 0 sphere, 1 box, 2 torus, 3 mandelbulb (leaves: push localSdf(T * pos[src]))
//       10 round, 11 smooth union, 12 smooth subtraction, 13 twist, 14 repetition, 15 animated translate (PRE)
//       20 round, 21 smooth union, 22 smooth subtraction (POST)
//   p:  sphere r | box halfSize | torus major, minor | mandelbulb power, iterations, enableAnimation, speed |
//       round radius | unions smoothness | twist amount | repetition spacing | animated: direction, amplitude, speed
struct RmInstr {
    float T[16];     // world -> local of the node (Primitive.transform)
    float Tinv[16];  // mat4.invert(T) as the operators recompute it per call (identity when singular)
    double p[6];
    int32_t op, src, dst;
    int32_t flags;   // bit 0 / 1: T / Tinv has the bottom row (0,0,0,1) (transform_mat4 skips the w divide);
                     // bit 2 / 3: T / Tinv is a pure translation (upper 3x3 identity): one f32 add per component
};
#define RM_PROG_MAX_SLOTS 16
#define RM_PROG_MAX_VALS 16

#define RM_BVH_LEAF_MAX 255
#define RM_MAX_STEPS 100
#define RM_MAX_DIST 10.0
#define RM_EPSILON 0.001

struct RmDiagBlock;   // rm_diag.h
struct RmDiagDevice;  // rm_kernels.h

// Kernel parameters (passed by value).
struct RmRenderParams {
    int32_t width, height, y_start, y_end;
    float rot[9];     // mat3.fromMat4(camera rotation), column-major
    float origin[3];  // camera position
    float light[3];   // normalize(f32(1,-1,1.5)) as phongModel.ts:15-16 computes it
    double rot_d[9], origin_d[3], light_d[3];  // the same binary32 values widened on the host (exact)
    int32_t n_prims;
    int32_t accel;    // rm_accel
    int32_t shader;   // rm_shader, used when rgba != nullptr
    int32_t bvh_nodes;
    int32_t oct_nodes;
    int32_t tile_w;   // pixels per wave row (64, 32, 16 or 8); wave tile = tile_w x (64 / tile_w)
    int32_t nodes_in_lds;  // 1: kernel stages the node / sphere tables in LDS (host decides)
    int32_t filter;        // 1: conservative f32 candidate filter for N-primitive loops
    int32_t variant;       // 1: v1 (one ray per lane, divergent loops); 2: v2 (uniform wave loop)
    int32_t list_cap;      // v2: per-ray hit-leaf list capacity in LDS (entries of 2 B per lane)
    int32_t coop;          // v2: wave-cooperative N-primitive fallback
    int32_t bvh_prim_count;
    int32_t oct_prim_count;
    int32_t blocks_per_cu;  // v2: persistent workgroups per CU
    int32_t num_cus;
    int32_t pq_dim[3];      // BVH point-query grid (rm_scene_host.cpp), 0 = absent
    int32_t pq_cell_count;
    int32_t pq_list_count;
    int32_t use_grid;
    float pq_origin[3];
    float pq_inv[3];
    float pq_cell[3];          // cell size of the point-query grid per axis, rounded DOWN (1 / pq_inv: rm_render_v2.hip rho_cell)
    int32_t refill_threshold;  // v2: idle lanes that trigger a ballot/prefix refill (64 = whole wave)
    int32_t hw_xcd;            // v2: read the XCD id from HW_REG_XCC_ID instead of blockIdx % 8
    int32_t item_px;           // v2: pixels per work item (64, 128, 256)
    int32_t local_rows;        // packed rows this launch renders (= y_end - y_start without striping)
    int32_t stripe_rows;       // > 0: rows are dealt in stripes of this many rows, round-robin over n_parts;
    int32_t n_parts, part;     //      this launch renders the stripes of `part`, packed in increasing y
    int32_t rel_boxes;         // v2: stage origin-relative node boxes (doubles) in LDS when they fit (option `rel`)
    // v2 longest-first item order from the previous frame's costs (rm_render_v2.hip, lpt_sort_kernel): entry k of queue x is
    // item lpt_perm[x * lpt_stride + k] of that queue (null: the queue's own order); every finished item's cost (wave-loop
    // iterations, saturated at 255) goes to lpt_cost_out[x * lpt_stride + item]
    const uint16_t *lpt_perm;
    uint8_t *lpt_cost_out;
    const uint8_t *lpt_cost_prev;
    uint16_t *lpt_perm_out;  // what the sort kernel of this launch writes (== lpt_perm)
    int32_t lpt_stride, lpt_pad;
    int32_t v1_lists;          // v1 BVH: per-ray hit-leaf lists in LDS (option `v1_lists`), placed at v1_list_offset by the launcher
    int32_t v1_list_offset;
    uint32_t lds_off[10];      // v2: byte offsets of the staged tables in LDS (nodes, prims, cells, list, oct, oct_prims, spheres,
                               // radii, rel, end), computed by the launcher: a section rebuilds its view from one scalar load
    int32_t prim_filter;       // general primitives (RmPrim): `spheres` holds a bounding sphere per primitive (rigid transforms only)
    int32_t n0_batch;          // v2 BVH: lanes waiting for getNormal that trigger the normal round while others still march (64: never)
    const int32_t *stripe_ids; // non-null (with stripe_rows > 0): the launch renders the stripes stripe_ids[0 .. ) in this
                               // order (increasing), packed; any deal of stripes to parts, e.g. a weighted one
    unsigned int *tile_counters;  // v2: 64 work-queue heads (8 XCDs x 8 sub-queues), zero before and after a launch
    unsigned long long *stamps;   // diagnostic build (-DRM_STAMPS) only: 8 cycle accumulators
    const uint32_t *pq_cells;
    const uint16_t *pq_list;
    const uint16_t *bvh_leaves;  // v2 bundle cull: node indices of the non-empty leaves, increasing; bvh_leaf_count entries (0: off)
    int32_t bvh_leaf_count;
    int32_t v1_block;  // v1: threads per workgroup (64, 128 or 256; option `v1_block`)
    int32_t oct_lean;  // v1, octree, sphere scenes with the cell table: the lean sphere-tracer kernel (render_kernel_oct; option `oct_lean`)
    const uint32_t *nn_cells;  // nearest-candidate lists per grid cell (all-primitive fallback)
    const uint16_t *nn_list;
    int32_t nn_cell_count, nn_list_count, use_nn;
    // v2 tile geometry precomputed by the launcher so the refill section has no integer division (tile_w and item_px
    // are powers of two; k / tiles_x through a magic multiplier, exact for k * tiles_x < 2^32)
    int32_t tile_w_log2, tile_h_log2, tiles_x, tiles_y;
    // an item's 64-pixel batches sit side by side (option `item_wide`: item = (tile_w * item_px / 64) x (64 / tile_w) pixels, the
    // batches of one wave's item complete whole 128-byte lines of every buffer) or one above the other (item = tile_w x item_px / tile_w)
    int32_t item_wide, item_w_log2, sub_dx, sub_dy;
    // Host side only (the launchers): the hipFunction_t of this scene's run-time specialised kernel (rm_rtc.h), or null
    const void *rtc_function;
    void *rtc_ctx;  // host side only: the context whose policy decides about a run-time compiled copy of the v2 kernel (rm_rtc_v2_hook); null: never
    uint32_t tiles_x_magic;
    int32_t lds_budget_kb;  // v2: LDS budget per workgroup the launcher aims for (option `lds_kb`; 0 = six workgroups per CU, then five, then four)
    int32_t leaf_order;  // BVH leaf lists are consecutive: leaf = spheres[first .. first+count), no id reads
    int32_t nn_dim[3];   // the nearest-candidate grid has its own (finer) resolution over the root box
    float nn_inv[3];
    int32_t algorithm;   // rm_algorithm; 0 = sphere tracer, 1..4 the other marchers (v1 kernel)
    int32_t general;     // 0: RmSphere records; 1: RmPrim records (`prims`); 2: expression programs (`prog`); 3: programs with a Mandelbulb
    int32_t uniform_radius;  // v2: every sphere has the same radius (candidates are ranked by squared centre distance)
    const RmPrim *prims;
    const RmInstr *prog;         // general == 2: instructions of every scene object, concatenated
    const int32_t *obj_ranges;   // general == 2: (first, count) into prog per scene object
    int32_t prog_slots, prog_vals;  // general >= 2: position slots / pending values of the deepest program (LDS sizing)
    double time;                 // Scene.updateTime(time) (raymarcher.ts:58-59): animated primitives
    double overshoot;    // AdaptiveStepV2/V3 overshootFactor (default 1.2)
    double step_size;    // FixedStep stepSize (default 0.1)
    const RmSphere *spheres;
    const double *radii;
    const RmBvhNode *bvh;
    const int32_t *bvh_prims;
    const RmOctNode *oct;
    const int32_t *oct_prims;
    const RmSphereRec *oct_recs;  // parallel to oct_prims (sphere scenes only, else null)
    const int32_t *oct_lut;       // 64^3 finest-level cells -> leaf node index (null: descend the tree)
    const RmOctFrameNode *oct_frame;  // the nodes relative to this launch's camera position (rm_api.cpp attaches it; null: no lean octree kernel)
    const uint32_t *oct_sub_hdr;  // per crowded leaf RM_OCT_SUB^3 sub-cells: (offset << 8) | count into oct_sub_list
    const uint8_t *oct_sub_list;  // positions within the leaf's record list of the spheres that can be nearest there
    uint8_t *depth;
    uint8_t *normal;
    uint16_t *sdf;
    uint16_t *iters;
    uint8_t *rgba;
    // Fused diagnostics (rm_diag.h; main.ts:528-548): the launch's accumulator block (from a ring in rm_api.cpp; all zero
    // before and after the launch) and where the launch's last wave writes the 32-byte result (null: no diagnostics).
    // v2 launches always carry a block: the last wave also re-zeroes the launch's tile-queue heads.
    int32_t lds_fill;    // v2: option `lds_fill` (rm_render_v2.hip: pad the LDS request so that exactly blocks_per_cu workgroups fit a CU)
    int32_t multi_step;  // v2 BVH: march steps inside a round while the leaf set and the winning sphere provably stay the same (option `multi_step`)
    RmDiagBlock *diag_block;
    RmDiagDevice *diag_out;
};
