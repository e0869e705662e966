// rm_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the sphere-tracing render path.
//
// One ray per lane.  A 64-lane wavefront owns a tile_w x (64 / tile_w) pixel tile (8 x 8 unless option tile_w says
// otherwise); a workgroup is one wave (option v1_block: up to four tiles stacked vertically) and workgroup ids are dealt
// to tiles in runs of eight per XCD (v1_tile_of_block).  There is no dense contraction
// anywhere on this path, so no MFMA: the work is FP64 VALU (the reference computes in JS
// doubles) with binary32 rounding exactly where the reference stores into Float32Array,
// plus f32 compares for box tests.  Build with -ffp-contract=off: JS never fuses a*b+c.
//
// Reference functions restated here (paths relative to the reference's src/):
//   Raymarcher.runRaymarcher / getSceneDistance / getNormal   cpu_algorithms/raymarcher.ts:46-135
//   SphereTracer.rayMarch                                     cpu_algorithms/sphereTracer.ts:15-83
//   Scene.getDistance                                         util/scene.ts:144-190
//   Primitive.sdf / Sphere.localSdf                           util/primitives/primitive.ts:33-39, sphere.ts:12-14
//   BVH.getPrimitivesAt / findRayIntersections / onRayMarch*  acceleration_structures/bvh.ts:95-240
//   BoundingBox.contains / intersectRay                       acceleration_structures/boundingBox.ts:15-21,69-105
//   Octree.findNode / marchRay / intersectRayBox              acceleration_structures/octree.ts:195-294
//   ShadingModel.shade (4 models)                             util/shading_models/*.ts
//   diagnostics                                               main.ts:528-548
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#include "rm_device.h"

// Diagnostic build only (-DRM_COUNTS, scripts/counts_v1.py): wave-level executions (slot i) and active lanes (slot i + 16)
// per event, straight into P.stamps[8 .. 39] with global atomics (slow; the build exists to count, not to be timed).
#ifdef RM_COUNTS
__device__ unsigned long long *rm_cnt_g;
#define RM_CNT(i)                                                                                  \
    {                                                                                              \
        const unsigned long long n_ = static_cast<unsigned long long>(__popcll(__ballot(1)));      \
        const int l_ = static_cast<int>(__lane_id());                                              \
        if (__builtin_amdgcn_readfirstlane(l_) == l_ && rm_cnt_g) {                                \
            atomicAdd(&rm_cnt_g[8 + (i)], 1ull);                                                   \
            atomicAdd(&rm_cnt_g[8 + (i) + 16], n_);                                                \
        }                                                                                          \
    }
#define RM_CNT1(i) RM_CNT(i)
#else
#define RM_CNT1(i) {}
#endif

#include "rm_bvh_list.h"
#include "rm_program.h"
#ifdef RM_RTC  // the run-time specialiser's build of this file (rm_rtc.cpp): one scene's expression forest as straight-line code
#include "rm_rtc_scene.inc"
#endif
#include "rm_kernels.h"
#include "rm_diag.h"

namespace {

using namespace rmd;

// Fused diagnostics of the one-ray-per-lane kernels (rm_diag.h): a wave stores its 64 pixels once, at its end; the four
// totals of the wave are formed with LDS atomics (lanes outside the frame contribute nothing; eight slots, lane & 7: 64
// lanes on one address would be 64 serial read-modify-writes) and flushed by the wave.  Reached by EVERY lane of every wave
// of the launch (no early return above it), all 64 lanes active.  Out of line, plain arguments: the lean octree kernel
// lives on exactly 64 VGPRs, and inlined this epilogue cost it two spills.
__device__ __attribute__((noinline)) static void v1_diag_flush(RmDiagBlock *blk, RmDiagDevice *out, bool has_pixel, uint32_t c16, uint32_t i16) {
    __shared__ unsigned int v1_diag_s[4][4][8];
    const int lane = static_cast<int>(__lane_id());
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    unsigned int(*w)[8] = v1_diag_s[wave];
    if (lane < 8) w[0][lane] = w[1][lane] = w[2][lane] = w[3][lane] = 0;  // (one wave's LDS operations execute in order)
    if (has_pixel) {
        const int k = lane & 7;
        atomicAdd(&w[0][k], c16);
        atomicAdd(&w[1][k], i16);
        atomicMax(&w[2][k], c16);
        atomicMax(&w[3][k], 0xFFFFFFFFu - c16);
    }
    unsigned int ts = 0, ti = 0, mx = 0, mi = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ts += w[0][k];
        ti += w[1][k];
        mx = w[2][k] > mx ? w[2][k] : mx;
        mi = w[3][k] > mi ? w[3][k] : mi;
    }
    const unsigned int wpw = blockDim.x >> 6;
    diag_flush_wave<true>(blk, out, nullptr, ts, ti, mx, mi, blockIdx.x * wpw + static_cast<unsigned int>(wave), gridDim.x * wpw, lane);
}
__device__ __forceinline__ void v1_diag_epilogue(const RmRenderParams &P, bool has_pixel, uint32_t c16, uint32_t i16) {
    if (P.diag_block) v1_diag_flush(P.diag_block, P.diag_out, has_pixel, c16, i16);
}

// ------------------------------------------------------------------ Scene.getDistance

// min over a primitive list for either scene representation
// GEN is a compile-time property of the kernel instantiation: a run-time `if (P.general)` in
// front of the sphere loop changed the results of render_kernel<0, true> (lane-grouping
// dependent wrong normals with hipcc 7.2), so the two representations never share a function body.
// GEN: 0 RmSphere records, 1 RmPrim records, 2 expression programs (rm_program.h), 3 programs with a Mandelbulb,
// 4 (RM_RTC builds only) the scene's programs as generated straight-line code.
template <int GEN>
__device__ __forceinline__ double list_min(const RmRenderParams &P, const int32_t *ids, int n, const Vec3f &p,
                                           double closest) {
#ifdef RM_RTC
    if (GEN == 4) {  // the scene's own code (generated: rm_rtc.cpp); per-lane object ids, no instruction stream
        for (int k = 0; k < n; ++k) closest = js_min_nan(rm_rtc_object_sdf(ids ? ids[k] : k, p, P.time), closest);
        return closest;
    }
#endif
    if (GEN >= 2) {  // Math.min(primitive.sdf(position), closestDistance), NaN-propagating
        for (int k = 0; k < n; ++k) {
            const int obj = ids ? ids[k] : k;
            // The interpreter reads its instruction stream through the scalar cache, so it runs ONE object at a time: lanes whose
            // lists name different objects at this position (different leaves) take turns (a waterfall over the distinct ids).
            double d = 0.0;
            for (bool done = false; !done;) {
                const int u = __builtin_amdgcn_readfirstlane(obj);
                if (obj == u) {
                    d = program_sdf<GEN == 3>(P.prog, P.obj_ranges[2 * u], P.obj_ranges[2 * u + 1], p, P.time, P.prog_slots);
                    done = true;
                }
            }
            closest = js_min_nan(d, closest);
        }
        return closest;
    }
    if (GEN == 1) {
        for (int k = 0; k < n; ++k) closest = min_dist(prim_sdf_general(P.prims[ids ? ids[k] : k], p), closest);
        return closest;
    }
#ifdef RM_REPRO_GENERAL_BRANCH  // round 1's form: the representation chosen by a RUN-TIME branch in the sphere instantiation
    if (P.general) {
        for (int k = 0; k < n; ++k) closest = min_dist(prim_sdf_general(P.prims[ids ? ids[k] : k], p), closest);
        return closest;
    }
#endif
    if (P.filter && n >= 2) return prims_min_best<int32_t>(P.spheres, P.radii, ids, n, 0, p, closest);  // scan, then one exact evaluation
    return prims_min<true>(P.spheres, P.radii, ids, n, p, closest, false);  // shared-reciprocal Math.hypot (bit-identical: rm_selftest_fastdiv)
}

// scene.ts:183-189 and the BVH fallback scene.ts:173: every primitive, all counted
// General primitives: the reference evaluates all N (scene.ts:173,183-189); with rigid transforms every primitive has a
// bounding sphere (centre c, radius R; host: build_scene_general) with |p - c| - R <= sdf(p) <= |p - c| + R, so a primitive
// whose lower bound exceeds the smallest upper bound (or 10, the starting value of `closest`) cannot attain
// min(10, min_j sdf_j(p)).  Binary32 bounds with the same error margin as the spheres' filter (sphere_sdf_estimate); only
// the survivors -- a handful of 40 -- are evaluated exactly (transformMat4 + Box / Torus / Sphere localSdf in binary64).
// Math.min is order independent, so the value is bit-identical; all N are counted.
__device__ double general_prims_filtered(const RmRenderParams &P, const Vec3f &p) {
    float ub = 10.0f;
    for (int j = 0; j < P.n_prims; ++j) {
        const RmSphere b = P.spheres[j];
        const float dx = p.x - b.cx, dy = p.y - b.cy, dz = p.z - b.cz;
        const float len = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
        const float hi = (len + b.rf) + (len + b.rf + 1.0f) * 4e-6f;
        ub = hi < ub ? hi : ub;
    }
    double closest = RM_MAX_DIST;
    for (int j = 0; j < P.n_prims; ++j) {
        const RmSphere b = P.spheres[j];
        const float dx = p.x - b.cx, dy = p.y - b.cy, dz = p.z - b.cz;
        const float len = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
        const float lo = (len - b.rf) - (len + b.rf + 1.0f) * 4e-6f;
        if (lo <= ub) closest = min_dist(prim_sdf_general(P.prims[j], p), closest);
    }
    return closest;
}

template <int GEN>
__device__ double all_prims_distance(const RmRenderParams &P, const Vec3f &p, uint32_t &count) {
    double closest = RM_MAX_DIST;
    if (GEN == 1 && P.prim_filter && P.n_prims >= 8) closest = general_prims_filtered(P, p);
    else closest = list_min<GEN>(P, nullptr, P.n_prims, p, closest);
    count += static_cast<uint32_t>(P.n_prims);
    return closest;
}

// scene.ts:167-181 with BVH.getPrimitivesAt (bvh.ts:95-121): union of the primitives of
// every leaf whose box contains p (a primitive lives in exactly one leaf), else all.
template <int GEN>
__device__ double bvh_distance(const RmRenderParams &P, const Vec3f &p, uint32_t &count) {
#ifdef RM_RTC_BVH_LEAVES  // this scene's leaves are code (rm_rtc.cpp, emit_bvh): no node loads, the objects called by name
    if (GEN == 4) return rm_rtc_bvh_distance(p, P.time, count);
#endif
    double closest = RM_MAX_DIST;
    uint32_t found = 0;
    int i = 0;
    int n = P.bvh_nodes;
    if (GEN < 2 && P.use_grid) {  // (not for expression programs: few objects, and their kernels are at the register limit)
        // BVH.getPrimitivesAt through the leaf grid (rm_scene_host.cpp build_point_query_grid): the leaves listed for p's
        // cell are a superset of the leaves whose box contains p; each is re-tested with the reference's inclusive
        // binary32 compares, and outside the root box no leaf can contain p.  Crowded cells keep the tree walk.
        const RmBvhNode root = P.bvh[0];
        bool walk = false;
        if (box_contains(root.lo, root.hi, p)) {
            const int cx = min(max(static_cast<int>((p.x - P.pq_origin[0]) * P.pq_inv[0]), 0), P.pq_dim[0] - 1);
            const int cy = min(max(static_cast<int>((p.y - P.pq_origin[1]) * P.pq_inv[1]), 0), P.pq_dim[1] - 1);
            const int cz = min(max(static_cast<int>((p.z - P.pq_origin[2]) * P.pq_inv[2]), 0), P.pq_dim[2] - 1);
            const uint32_t cell = P.pq_cells[(cz * P.pq_dim[1] + cy) * P.pq_dim[0] + cx];
            const int ccnt = static_cast<int>(cell & 0xFFu);
            if (ccnt == 255) walk = true;
            else {
                const uint16_t *lst = P.pq_list + (cell >> 8);
                for (int e = 0; e < ccnt; ++e) {
                    const RmBvhNode node = P.bvh[lst[e]];
                    if (!box_contains(node.lo, node.hi, p)) continue;
                    const int first = node.leaf >> 8, cnt = node.leaf & 0xFF;
                    closest = list_min<GEN>(P, P.bvh_prims + first, cnt, p, closest);
                    found += static_cast<uint32_t>(cnt);
                }
            }
        }
        if (!walk) n = 0;  // the grid answered: skip the walk
    }
    while (i < n) {
        const RmBvhNode node = P.bvh[i];
        if (!box_contains(node.lo, node.hi, p)) {
            i = node.skip;
            continue;
        }
        if (node.leaf < 0) {
            i = i + 1;
            continue;
        }
        const int first = node.leaf >> 8, cnt = node.leaf & 0xFF;
        closest = list_min<GEN>(P, P.bvh_prims + first, cnt, p, closest);
        found += static_cast<uint32_t>(cnt);
        i = node.skip;
    }
    if (found == 0) return all_prims_distance<GEN>(P, p, count);
    count += found;
    return closest;
}

// Octree.findNode (octree.ts:223-248): -1 when p is outside the root cube.  Children tile
// their parent at the f32 centre and `contains` is inclusive, so the first child (index
// order x + 2y + 4z) that contains p takes the low half on every axis where p <= centre.
// With the cell table: on each axis the leaf is the cell k with b_k < p <= b_{k+1} (b_k = -10 + 0.3125 k,
// exact in binary32; the lowest cell also takes p = -10), which is what "first child that contains p" selects.
// k is guessed from a binary32 product (off by at most one cell) and corrected with exact compares.
__device__ __forceinline__ int oct_cell(float v) {
    int k = static_cast<int>((v + 10.0f) * 3.2f);
    k = k < 0 ? 0 : (k > 63 ? 63 : k);
    const float b = -10.0f + 0.3125f * static_cast<float>(k);  // exact
    if (v <= b && k > 0) k -= 1;
    else if (v > b + 0.3125f && k < 63) k += 1;
    return k;
}

__device__ int oct_find(const RmRenderParams &P, const Vec3f &p) {
    const RmOctNode *nodes = P.oct;
    if (!box_contains(nodes[0].lo, nodes[0].hi, p)) return -1;
    if (P.oct_lut) return P.oct_lut[(oct_cell(p.z) * 64 + oct_cell(p.y)) * 64 + oct_cell(p.x)];
    int i = 0;
    for (;;) {
        const int first = nodes[i].first_child;
        if (first < 0) return i;
        const float cx = nodes[i].center[0], cy = nodes[i].center[1], cz = nodes[i].center[2];
        i = first + (p.x > cx ? 1 : 0) + (p.y > cy ? 2 : 0) + (p.z > cz ? 4 : 0);
    }
}

// One leaf-ordered record against the running minimum (same filter as prims_min)
__device__ __forceinline__ void rec_consider(const float4 c, const RmSphereRec *rec, const Vec3f &p, double &closest, float &ub) {
    const float dx = p.x - c.x, dy = p.y - c.y, dz = p.z - c.z;
    const float len = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
    const float err = (len + __builtin_fabsf(c.w) + 1.0f) * 4e-6f;  // sphere_sdf_estimate
    if ((len - c.w) - err <= ub) {
        const double e = vec3_length(dx, dy, dz) - rec->radius;
        if (e < closest) {
            closest = e;
            ub = f32_upper_bound(e);
        }
    }
}

// min over a leaf's sphere records (RmSphereRec).  Lanes of a wave sit in different leaves, so a loop that
// evaluates a sphere exactly as soon as its bound passes runs the 40-instruction FP64 body once per distinct
// position at which SOME lane passes -- measured ~9 times per leaf visit on the 10k-sphere scene, each for a few
// lanes.  Instead the scan only keeps, per lane, the sphere with the smallest upper bound (k1, hi1, lb1) and the
// smallest lower bound among all the others (lb2); afterwards every lane evaluates its k1 exactly in the SAME
// instruction stream.  If lb2 exceeds that exact value no other sphere can be closer (exact_j >= lb_j >= lb2);
// otherwise (near ties) the list is rescanned with the ordinary filter.  Four 16-B loads are in flight per step.
// `sub` != nullptr: only the listed positions are scanned (the sub-cell's candidates, rm_scene_host.h); the near-tie
// rescan always covers the whole list.
__device__ double recs_min(const RmSphereRec *recs, int n, const Vec3f &p, double closest, const uint8_t *sub = nullptr,
                           int n_sub = 0) {
    if (n <= 0) return closest;
    const float inf = __builtin_inff();
    int k1 = 0;
    float hi1 = inf, lb1 = inf, lb2 = inf;
    auto scan = [&](const float4 c, int k) {
        const float dx = p.x - c.x, dy = p.y - c.y, dz = p.z - c.z;
        const float len = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
        const float err = (len + __builtin_fabsf(c.w) + 1.0f) * 4e-6f;  // sphere_sdf_estimate
        const float a = len - c.w, lb = a - err, hi = a + err;
        const bool better = hi < hi1;
        lb2 = __builtin_fminf(lb2, better ? lb1 : lb);
        k1 = better ? k : k1;
        lb1 = better ? lb : lb1;
        hi1 = better ? hi : hi1;
    };
    int k = 0;
    if (sub) {
        for (int e = 0; e < n_sub; ++e) {
            RM_CNT1(6)
            const int j = sub[e];
            scan(*reinterpret_cast<const float4 *>(recs + j), j);
        }
        k = n;  // skip the full scan below
    }
    for (; k + 4 <= n; k += 4) {
        RM_CNT1(7)
        const float4 c0 = *reinterpret_cast<const float4 *>(recs + k);
        const float4 c1 = *reinterpret_cast<const float4 *>(recs + k + 1);
        const float4 c2 = *reinterpret_cast<const float4 *>(recs + k + 2);
        const float4 c3 = *reinterpret_cast<const float4 *>(recs + k + 3);
        scan(c0, k);
        scan(c1, k + 1);
        scan(c2, k + 2);
        scan(c3, k + 3);
    }
    for (; k < n; ++k) {
        RM_CNT1(15)
        scan(*reinterpret_cast<const float4 *>(recs + k), k);
    }
    {
        RM_CNT1(8)
        const float4 c = *reinterpret_cast<const float4 *>(recs + k1);
        const double e = vec3_length(p.x - c.x, p.y - c.y, p.z - c.z) - recs[k1].radius;
        closest = e < closest ? e : closest;
    }
    float ub = f32_upper_bound(closest);
    if (lb2 <= ub) {  // another sphere may tie or win: ordinary filtered pass over the whole list
        RM_CNT1(9)
        for (k = 0; k < n; ++k) rec_consider(*reinterpret_cast<const float4 *>(recs + k), recs + k, p, closest, ub);
    }
    return closest;
}

// scene.ts:148-166 given the node findNode returned
template <int GEN>
__device__ double oct_node_distance(const RmRenderParams &P, int node, const Vec3f &p, uint32_t &count) {
    if (node < 0) {
        RM_CNT1(12)
        return all_prims_distance<GEN>(P, p, count);  // outside the cube: scene.ts:166,183-189
    }
    const RmOctNode nd = P.oct[node];
    double closest = RM_MAX_DIST;
    if (nd.prim_count > 0) {
        RM_CNT1(2)
        if (GEN == 0 && P.oct_recs && P.filter) {
            const uint8_t *sub = nullptr;
            int n_sub = 0;
            if (P.oct_sub_hdr && nd.sub_first >= 0) {  // crowded leaf: the candidates of p's sub-cell
                const float ix = nd.center[0], iy = nd.center[1], iz = nd.center[2];  // RM_OCT_SUB / (hi - lo), from the host
                const int sx = min(max(static_cast<int>((p.x - nd.lo[0]) * ix), 0), RM_OCT_SUB - 1);
                const int sy = min(max(static_cast<int>((p.y - nd.lo[1]) * iy), 0), RM_OCT_SUB - 1);
                const int sz = min(max(static_cast<int>((p.z - nd.lo[2]) * iz), 0), RM_OCT_SUB - 1);
                const uint32_t hdr = P.oct_sub_hdr[nd.sub_first + (sz * RM_OCT_SUB + sy) * RM_OCT_SUB + sx];
                sub = P.oct_sub_list + (hdr >> 8);
                n_sub = static_cast<int>(hdr & 0xFFu);
            }
            closest = recs_min(P.oct_recs + nd.prim_first, nd.prim_count, p, closest, sub, n_sub);
        }
        else closest = list_min<GEN>(P, P.oct_prims + nd.prim_first, nd.prim_count, p, closest);
        count += static_cast<uint32_t>(nd.prim_count);
    } else if (nd.is_empty) {
        RM_CNT1(11)
        closest = min_dist(nd.min_distance * 0.99, closest);  // Math.min(closest, minDistance * safety)
    }
    return closest;
}

template <int ACCEL, int GEN>
__device__ __forceinline__ double scene_distance(const RmRenderParams &P, const Vec3f &p, uint32_t &count) {
    if (ACCEL == 2) return bvh_distance<GEN>(P, p, count);
    if (ACCEL == 1) return oct_node_distance<GEN>(P, oct_find(P, p), p, count);
    return all_prims_distance<GEN>(P, p, count);
}

// ------------------------------------------------------------------ BVH ray intervals

// The reference materialises every leaf interval of the ray, stable-sorts them by tEnter
// (bvh.ts:126-178) and walks the list with an index (bvh.ts:204-240).  This returns the
// element that follows key (keyT, keyOrd) in exactly that order -- ascending tEnter, ties
// in traversal order -- by one stackless traversal, so no per-ray list is stored.
__device__ bool bvh_next_interval(const RmRenderParams &P, const Ray &r, const RayInv &ri, double keyT, int keyOrd,
                                  Interval &out) {
#ifdef RM_RTC_BVH_LEAVES
    return rm_rtc_bvh_next_interval(r, ri, keyT, keyOrd, out);
#else
    bool have = false;
    int i = 0;
    const int n = P.bvh_nodes;
    while (i < n) {
        const RmBvhNode node = P.bvh[i];
        double tE, tX;
        // 1 / direction[i] (boundingBox.ts:78) is the same quotient for every box: computed once per ray
        if (!slab_inv(node.lo, node.hi, r, ri, tE, tX) || tX < 0.0 || tE > RM_MAX_DIST) {  // bvh.ts:145,151
            i = node.skip;
            continue;
        }
        if (node.leaf < 0) {
            i = i + 1;
            continue;
        }
        if ((node.leaf & 0xFF) > 0) {  // bvh.ts:165
            const double cE = tE > 0.0 ? tE : 0.0;                  // Math.max(tEnter, tMin = 0)
            const double cX = tX < RM_MAX_DIST ? tX : RM_MAX_DIST;  // Math.min(tExit, tMax = 10)
            const bool after = cE > keyT || (cE == keyT && i > keyOrd);
            const bool better = !have || cE < out.tEnter;  // equal tEnter: earlier node wins, seen first
            if (after && better) {
                out.tEnter = cE;
                out.tExit = cX;
                out.ord = i;
                have = true;
            }
        }
        i = node.skip;
    }
    return have;
#endif
}

// ------------------------------------------------------------------ the render kernel

template <int ACCEL, int GEN>
__device__ double normal_and_store(const RmRenderParams &P, const Ray &ray, double depth, uint32_t &count,
                                   uint8_t nb[3]) {
    // raymarcher.ts:94-105
    Vec3f hit;
    hit.x = to_f32(static_cast<double>(ray.o.x) + static_cast<double>(ray.d.x) * depth);
    hit.y = to_f32(static_cast<double>(ray.o.y) + static_cast<double>(ray.d.y) * depth);
    hit.z = to_f32(static_cast<double>(ray.o.z) + static_cast<double>(ray.d.z) * depth);
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (!(depth >= RM_MAX_DIST)) {
        RM_CNT1(13)
        // raymarcher.ts:123-135 getNormal
        const double d0 = scene_distance<ACCEL, GEN>(P, hit, count);
        Vec3f q = hit;
        q.x = to_f32(static_cast<double>(hit.x) - 0.01);
        nx = to_f32(d0 - scene_distance<ACCEL, GEN>(P, q, count));
        q = hit;
        q.y = to_f32(static_cast<double>(hit.y) - 0.01);
        ny = to_f32(d0 - scene_distance<ACCEL, GEN>(P, q, count));
        q = hit;
        q.z = to_f32(static_cast<double>(hit.z) - 0.01);
        nz = to_f32(d0 - scene_distance<ACCEL, GEN>(P, q, count));
        double len = static_cast<double>(nx) * nx + static_cast<double>(ny) * ny + static_cast<double>(nz) * nz;
        if (len > 0) len = 1 / __builtin_sqrt(len);
        nx = to_f32(nx * len);
        ny = to_f32(ny * len);
        nz = to_f32(nz * len);
    }
    nb[0] = u8clamp((static_cast<double>(nx) + 1) * 0.5 * 255);
    nb[1] = u8clamp((static_cast<double>(ny) + 1) * 0.5 * 255);
    nb[2] = u8clamp((static_cast<double>(nz) + 1) * 0.5 * 255);
    return depth;
}

// The per-ray hit-leaf list of rm_bvh_list.h in this kernel: BVH.onRayMarchStart walks the tree once and records the
// leaves the ray hits; every later "next interval" scans that short list instead of walking the whole tree again with a
// slab test per node (what bvh_next_interval does; it remains for launches without the list).  The list lives in dynamic
// LDS behind the expression programs' slots (list_lds_offset), RM_V1_LIST_CAP entries of 2 B per lane.
#define RM_V1_LIST_CAP 16
__device__ __forceinline__ SceneView v1_view(const RmRenderParams &P) {
    SceneView S;
    S.nodes = P.bvh;
    S.bvh_prims = P.bvh_prims;
    S.oct = P.oct;
    S.oct_prims = P.oct_prims;
    S.spheres = P.spheres;
    S.radii = P.radii;
    S.pq_cells = P.pq_cells;
    S.pq_list = P.pq_list;
    S.nn_cells = P.nn_cells;
    S.nn_list = P.nn_list;
    S.rel = nullptr;
    S.n_prims = P.n_prims;
    S.bvh_nodes = P.bvh_nodes;
    return S;
}
__device__ __forceinline__ RayList v1_ray_list(const RmRenderParams &P) {
    extern __shared__ __align__(16) unsigned char v1_smem[];
    RayList L;
    L.cap = RM_V1_LIST_CAP;
    L.cnt = L.live = L.cur_pos = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    L.col = reinterpret_cast<uint16_t *>(v1_smem + P.v1_list_offset) + (static_cast<size_t>(wave) * RM_V1_LIST_CAP) * 64 + lane;
    return L;
}

// SphereTracer.rayMarch (sphereTracer.ts:15-83)
template <int ACCEL, int GEN>
__device__ double ray_march(const RmRenderParams &P, const Ray &ray, uint32_t &count, uint32_t &iters) {
    double t = 0.0;
    Interval cur;
    bool haveCur = false;
    RayInv ri;
    OctInv oinv;
    RayList L;
    const bool lists = ACCEL == 2 && GEN < 2 && P.v1_list_offset >= 0;  // expression programs: a handful of objects, nothing to gain
    if (ACCEL == 1) oinv = make_oct_inv(ray);
    if (ACCEL == 2) {
        ri = make_ray_inv(ray);
        if (lists) {
            L = v1_ray_list(P);
            haveCur = bvh_prologue<false>(v1_view(P), ray, ri, L, cur);  // onRayMarchStart
        } else {
            haveCur = bvh_next_interval(P, ray, ri, -__builtin_inf(), -1, cur);
        }
        if (!haveCur) return RM_MAX_DIST;                                // bvh.ts:190-192
    }
    for (int i = 0; i < RM_MAX_STEPS; ++i) {
        RM_CNT1(0)
        Vec3f p;
        p.x = to_f32(static_cast<double>(ray.o.x) + static_cast<double>(ray.d.x) * t);
        p.y = to_f32(static_cast<double>(ray.o.y) + static_cast<double>(ray.d.y) * t);
        p.z = to_f32(static_cast<double>(ray.o.z) + static_cast<double>(ray.d.z) * t);
        int onode = -1;
        if (ACCEL == 2) {
            // BVH.onRayMarchStep (bvh.ts:204-240)
            double skip = 0.0;
            if (!haveCur) return RM_MAX_DIST;  // currentIntervalIdx >= length
            if (t < cur.tEnter) skip = cur.tEnter - t;
            else if (t > cur.tExit) {
                const Interval prev = cur;
                haveCur = lists ? bvh_next<false>(v1_view(P), ray, ri, L, prev.tEnter, prev.ord, cur)
                                : bvh_next_interval(P, ray, ri, prev.tEnter, prev.ord, cur);  // idx++
                if (!haveCur) return RM_MAX_DIST;
                if (cur.tEnter > t) skip = cur.tEnter - t;
            }
            if (skip > 0.0) {
                t += skip;
                if (t > RM_MAX_DIST) break;
                continue;
            }
        } else if (ACCEL == 1) {
            onode = oct_find(P, p);  // marchRay recomputes this same point
            if (onode >= 0) {
                const double skip = oct_skip(P.oct[onode], ray, t, oinv);
                if (skip > 0.0) {
                    RM_CNT1(1)
                    t += skip;
                    if (t > RM_MAX_DIST) break;
                    continue;
                }
            }
        }
        double dist;
        if (ACCEL == 1) dist = oct_node_distance<GEN>(P, onode, p, count);
        else dist = scene_distance<ACCEL, GEN>(P, p, count);
        t += dist;
        iters += 1;
        if (dist < RM_EPSILON) break;
        if (t > RM_MAX_DIST) break;
    }
    return t;
}

// The four other marchers of the Algorithm plugin (raymarchWorker.ts:49-68) share the accel
// prologue and skip protocol of the sphere tracer and differ in the step rule:
//   1 FixedStep       fixedStep.ts:21-94        MAX_STEPS 200, returns MAX_DIST unless it hit
//   2 AdaptiveStep    adaptiveStep.ts:22-105    MAX_STEPS 200, step = clamp(0.8 d, 0.025, 0.5) or 0.01 near
//   3 AdaptiveStepV2  adaptiveStepV2.ts:22-124  overshoot by overshootFactor, step back when spheres do not overlap
//   4 AdaptiveStepV3  adaptiveStepV3.ts:22-137  as V2 plus the "bridging" third evaluation
template <int ACCEL, int GEN>
__device__ double ray_march_other(const RmRenderParams &P, const Ray &ray, uint32_t &count, uint32_t &iters) {
    const int alg = P.algorithm;
    const int max_steps = (alg == 1 || alg == 2) ? 200 : 100;
    const double FIXED_STEP_SIZE = 0.1, STEP_SCALE = 0.8;
    const double MIN_STEP = FIXED_STEP_SIZE * 0.25, MAX_STEP = FIXED_STEP_SIZE * 5.0;
    const double NEAR_DIST = 0.1, NEAR_STEP = 0.01;
    double t = 0.0, prevSDF = 0.0, prevStep = 0.0;
    bool hit = false;
    Interval cur;
    bool haveCur = false;
    RayInv ri;
    OctInv oinv;
    RayList L;
    const bool lists = ACCEL == 2 && GEN < 2 && P.v1_list_offset >= 0;  // expression programs: a handful of objects, nothing to gain
    if (ACCEL == 1) oinv = make_oct_inv(ray);
    if (ACCEL == 2) {
        ri = make_ray_inv(ray);
        if (lists) {
            L = v1_ray_list(P);
            haveCur = bvh_prologue<false>(v1_view(P), ray, ri, L, cur);
        } else {
            haveCur = bvh_next_interval(P, ray, ri, -__builtin_inf(), -1, cur);
        }
        if (!haveCur) return RM_MAX_DIST;
    }
    for (int i = 0; i < max_steps; ++i) {
        Vec3f p = point_at(ray, t);
        int onode = -1;
        double skip = 0.0;
        if (ACCEL == 2) {  // BVH.onRayMarchStep (bvh.ts:204-240)
            if (!haveCur) return RM_MAX_DIST;
            if (t < cur.tEnter) skip = cur.tEnter - t;
            else if (t > cur.tExit) {
                const Interval prev = cur;
                haveCur = lists ? bvh_next<false>(v1_view(P), ray, ri, L, prev.tEnter, prev.ord, cur)
                                : bvh_next_interval(P, ray, ri, prev.tEnter, prev.ord, cur);
                if (!haveCur) return RM_MAX_DIST;
                if (cur.tEnter > t) skip = cur.tEnter - t;
            }
        } else if (ACCEL == 1) {
            onode = oct_find(P, p);
            if (onode >= 0) skip = oct_skip(P.oct[onode], ray, t, oinv);
        }
        if (skip > 0.0) {
            t += skip;
            if (t > RM_MAX_DIST) break;
            prevSDF = 0.0;  // adaptiveStepV2.ts:73-74 (harmless for the marchers without this state)
            prevStep = 0.0;
            continue;
        }
        const double dist = ACCEL == 1 ? oct_node_distance<GEN>(P, onode, p, count) : scene_distance<ACCEL, GEN>(P, p, count);
        iters += 1;
        if (alg == 1 || alg == 2) {
            if (dist < RM_EPSILON) {
                hit = true;
                break;
            }
            double step;
            if (alg == 1) step = P.step_size;
            else if (dist < NEAR_DIST) step = NEAR_STEP;
            else {
                step = STEP_SCALE * dist;
                if (step < MIN_STEP) step = MIN_STEP;
                if (step > MAX_STEP) step = MAX_STEP;
            }
            t += step;
            if (t > RM_MAX_DIST) break;
            continue;
        }
        // AdaptiveStepV2 / V3
        if (dist < RM_EPSILON) break;
        if (t > RM_MAX_DIST) break;
        if (i == 0 || prevSDF == 0.0) {
            t += dist;
            prevSDF = dist;
            prevStep = dist;
            continue;
        }
        if (prevStep <= (prevSDF + dist)) {  // spheresOverlapped
            const double step = dist * P.overshoot;
            t += step;
            prevSDF = dist;
            prevStep = step;
            continue;
        }
        if (alg == 3) {  // adaptiveStepV2.ts:108-114
            t -= prevStep;
            t += prevSDF;
            prevStep = prevSDF;
            continue;
        }
        // adaptiveStepV3.ts:103-130
        const double originalPos = t - prevStep;
        t = originalPos + prevSDF;
        p = point_at(ray, t);
        const double d3 = scene_distance<ACCEL, GEN>(P, p, count);
        iters += 1;
        if (prevSDF + dist + d3 >= prevStep) {
            t = originalPos + prevStep + dist;
            prevSDF = dist;
            prevStep = dist;
            continue;
        }
        prevSDF = d3;
        prevStep = d3;
        t += d3;
    }
    if (alg == 1 || alg == 2) return hit ? t : RM_MAX_DIST;
    return t;
}

// Workgroup id -> tile for the one-ray-per-lane kernels.  Workgroups go to the eight XCDs round-robin (id % 8), each with
// its own L2; a wave tile is 8 pixels wide, so a tile row writes 32-byte pieces of 128-byte lines.  Tiles are therefore
// dealt in runs of eight: ids 0, 8, 16 ... 56 (all on XCD 0, back to back) are tiles 0 ... 7, ids 1, 9, ... are tiles
// 8 ... 15, and so on: the pieces of a line meet in one L2 and leave as whole lines.  (The launcher rounds the grid up to
// a multiple of 64; ids past the last tile return.)
__device__ __forceinline__ int v1_tile_of_block(int b) {
    const int r = b & 63;
    return (b & ~63) + ((r & 7) << 3) + (r >> 3);
}

#ifndef RM_RTC  // (the lean octree kernel serves sphere scenes only)
// ------------------------------------------------------------------ the octree sphere tracer, lean form
//
// render_kernel<1, false, 0> spends 60 % of its instructions in the part of the march step that precedes the distance
// (scripts/counts_v1.py on the 10 000-sphere scene: 61 loop trips per wave, 38 of them end in a skip; ~135 VALU
// instructions per trip for findNode + marchRay, and ten dependent loads in a row).  The same arithmetic with less
// around it, for sphere scenes whose octree has the cell table (the launcher checks: rm_launch_render):
//  * findNode: the root cube is [-10, 10]^3 (scene.ts:81-85), so `contains` is max(|x|, |y|, |z|) <= 10 (the march
//    point is finite: finite camera, finite t); the cell index is guessed from a product that is biased UPWARDS
//    (3.20001f: never below the true cell, at most one above -- |error| of the two binary32 roundings and of the
//    constant < 4e-6 relative, cells <= 64) and corrected with ONE exact compare against the cell's lower face
//    (b_k = -10 + 0.3125 k, exact in binary32, formed with an fma that therefore rounds nothing).
//  * marchRay / intersectRayBox (octree.ts:195-220,250-294): the reference swaps t0 / t1 when 1 / d < 0; here the ray
//    knows from its direction signs which face is the far and which the near one on every axis and loads them with
//    per-lane byte offsets: no selects, same products.  The comparisons of the binary32 tMin / tMax are made in
//    binary32 (exact either way) and without branches.
//  * every load is base (SGPR pair) + 32-bit byte offset, and everything a step can need from the node record is
//    requested in ONE batch right after the cell table's answer.
//  * (face - origin) is the same binary64 difference for every ray of a frame: oct_frame_table_kernel forms it once per
//    node and camera position (RmOctFrameNode; rm_api.cpp keeps the tables of the last few camera positions), and the
//    march step multiplies it by the ray's 1 / d.  minDistance * 0.99 (octree.ts:282) sits beside it.
//  * getNormal's four samples are four more trips of the SAME loop (a phase per ray, as in the v2 kernel): one copy of
//    the distance code, and a ray that has finished marching shares its evaluations with neighbours that still march.
// Leaf records, sub-cell candidate lists, filter margins, shading and counters are those of render_kernel<1, ...>,
// which stays the independent cross-check (option `oct_lean` = 0).
__device__ __forceinline__ uint32_t oct_cell_up(float v) {
    const float g = __builtin_truncf((v + 10.0f) * 3.20001f);          // in [0, 64]
    const float b = __builtin_fmaf(g, 0.3125f, -10.0f);                // exact: lower face of cell g
    const int k = static_cast<int>(g) - (v <= b ? 1 : 0);              // the lowest cell also takes v = -10 (clamp)
    return static_cast<uint32_t>(min(max(k, 0), 63));
}

template <typename T>
__device__ __forceinline__ T ld_off(const void *base, uint32_t byte_off) {  // global_load ... v_off, s[base:base+1]
    return *reinterpret_cast<const T *>(static_cast<const char *>(base) + byte_off);
}

// outside the cube: every primitive (scene.ts:166,183-189), as all_prims_distance<0>.  Out of line and with plain
// arguments: a reference to the kernel's parameter block would force a copy of it into scratch.
__device__ __attribute__((noinline)) double oct_outside_distance(const RmSphere *spheres, const double *radii, int n, bool filter, float px,
                                                                 float py, float pz) {
    const Vec3f p = {px, py, pz};
    if (filter && n >= 2) return prims_min_best<int32_t>(spheres, radii, nullptr, n, 0, p, RM_MAX_DIST);
    return prims_min<true>(spheres, radii, static_cast<const int32_t *>(nullptr), n, p, RM_MAX_DIST, false);
}

// the near-tie pass of recs_min (another sphere may tie or win): ordinary filtered pass over the whole leaf, out of line
__device__ __attribute__((noinline)) double oct_leaf_rescan(const RmSphereRec *recs, int n, float px, float py, float pz, double closest) {
    const Vec3f p = {px, py, pz};
    float ub = f32_upper_bound(closest);
    for (int k = 0; k < n; ++k) rec_consider(*reinterpret_cast<const float4 *>(recs + k), recs + k, p, closest, ub);
    return closest;
}

__global__ __launch_bounds__(64, 8) void render_kernel_oct(const RmRenderParams P) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tw = P.tile_w, th = 64 / tw;
    const int tiles_x = (P.width + tw - 1) / tw;
    const int tile = v1_tile_of_block(static_cast<int>(blockIdx.x));
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int x = bx * tw + (lane % tw);
    const int wpw = static_cast<int>(blockDim.x >> 6);             // waves per workgroup (the launcher uses 1)
    const int row = by * (wpw * th) + wave * th + (lane / tw);  // tile-local row
#ifdef RM_COUNTS
    if (threadIdx.x == 0) rm_cnt_g = P.stamps;
    __syncthreads();
#endif
    // (no early return: every wave takes part in the diagnostics flush at the end; a lane outside the frame starts finished)
    const bool in_frame = x < P.width && row < P.local_rows;
    if (in_frame) RM_CNT1(14)
    const int y = row_to_y(P, row);
    const Ray ray = make_ray(P, x, y);
    // 1 / direction (octree.ts:200) and, per axis, the byte offset inside RmOctNode of the face the ray leaves through
    // (lo: 0, hi: 16, + 4 per axis): `if (invDir < 0) swap(t0, t1)` (octree.ts:204-208) decided once per ray
    const double inv0 = 1.0 / static_cast<double>(ray.d.x), inv1 = 1.0 / static_cast<double>(ray.d.y), inv2 = 1.0 / static_cast<double>(ray.d.z);
    const uint32_t far0 = inv0 < 0.0 ? 0u : 32u, far1 = inv1 < 0.0 ? 8u : 40u, far2 = inv2 < 0.0 ? 16u : 48u;  // into RmOctFrameNode
    // tEnter needs no evaluation when tExit is clearly ahead.  The march point p = fl32(o + d t) lies inside the leaf's box
    // (findNode), so on every axis the ray's real entry parameter is at most t + eps / |d_a| with eps < 6e-7 (the binary32
    // rounding of a coordinate of magnitude <= 10; the binary64 roundings are nine orders smaller), and the computed tMin of
    // that axis exceeds the real one by < 1e-7 relative (two binary64 roundings, one binary32).  With
    // E = 4e-6 max_a |1 / d_a| (>= 4e-6): tEnter <= (t + 0.15 E)(1 + 1e-7) < t + E for t <= 10 (0.85 E > 1.01e-6), so `tEnter > tExit`
    // (octree.ts:215) is false when tExit - t > E.  A zero direction component makes E infinite: such rays always take the full test.
    const double near_margin = 4e-6 * __builtin_fmax(__builtin_fmax(__builtin_fabs(inv0), __builtin_fabs(inv1)), __builtin_fabs(inv2));

    enum { PH_MARCH = 0, PH_N0 = 1, PH_N1 = 2, PH_N2 = 3, PH_N3 = 4, PH_DONE = 5 };
    // Registers decide the occupancy here (64 VGPRs: eight waves per SIMD).  The two Uint16Array counters share one: the march
    // iterations (<= 100) in the low half, the SDF evaluations in the high half, where `+=` wraps exactly as the Uint16Array
    // store would (raymarcher.ts:103-106).  The trip counter of the march loop rides above the phase (phase in bits 0-2).
    uint32_t counters = 0;
    int phase = in_frame ? PH_MARCH : PH_DONE;  // | steps << 3
    double t = 0.0, d0 = 0.0;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    const void *const nodes = P.oct;
    const void *const frame = P.oct_frame;
    // One structured body per trip (no `continue` / `break`: each of them costs a loop level of mask bookkeeping in the
    // compiled code): a trip either skips, or evaluates one distance and consumes it according to the ray's phase.
    while (phase != PH_DONE) {
        RM_CNT1(0)
        asm volatile("" : "+v"(phase));  // opaque across the back edge: otherwise the optimiser threads the known phase values through
                                         // the loop and the structuriser turns the result into a six-deep loop nest
        if ((phase & 7) == PH_MARCH) {
            if (phase >= (RM_MAX_STEPS << 3)) phase = (t >= RM_MAX_DIST) ? PH_DONE : PH_N0;  // loop exhausted: return totalDist
            else phase += 8;
        }
        if (phase != PH_DONE) {
            // the march point (sphereTracer.ts:44-45), the hit position (raymarcher.ts:94-95) or one of getNormal's offsets (:126-131)
            Vec3f p = point_at(ray, t);
            if (static_cast<unsigned>(phase) - PH_N1 <= static_cast<unsigned>(PH_N3 - PH_N1)) {  // x - 0.01 in binary64, stored to the Float32Array; x - 0.0 is x
                p.x = to_f32(static_cast<double>(p.x) - (phase == PH_N1 ? 0.01 : 0.0));
                p.y = to_f32(static_cast<double>(p.y) - (phase == PH_N2 ? 0.01 : 0.0));
                p.z = to_f32(static_cast<double>(p.z) - (phase == PH_N3 ? 0.01 : 0.0));
            }
            double dist = RM_MAX_DIST;
            bool evaluated = true;
            const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(p.x), __builtin_fabsf(p.y)), __builtin_fabsf(p.z));
            if (m <= 10.0f) {  // Octree.findNode through the cell table
                const uint32_t cell = (oct_cell_up(p.z) * 64u + oct_cell_up(p.y)) * 64u + oct_cell_up(p.x);
                const uint32_t node = static_cast<uint32_t>(ld_off<int32_t>(P.oct_lut, cell * 4u));
                __builtin_assume(node < (1u << 24));
                const uint32_t nb = node * 64u;
                // one batch: the far faces and the skip cap of the frame's table, primCount / isEmpty, and what a leaf evaluation reads
                const double ff0 = ld_off<double>(frame, nb + far0), ff1 = ld_off<double>(frame, nb + far1), ff2 = ld_off<double>(frame, nb + far2);
                const double cap = ld_off<double>(frame, nb + 24u);  // minDistance * 0.99
                const int2 pc_ie = ld_off<int2>(nodes, nb + 40u);    // prim_count, is_empty
                const int prim_first = ld_off<int32_t>(nodes, nb + 28u);
                const float4 sub = ld_off<float4>(nodes, nb + 48u);  // sub-cells per unit length x 3, sub_first
                const int prim_count = pc_ie.x;
                if ((phase & 7) == PH_MARCH && pc_ie.y) {  // Octree.marchRay (octree.ts:250-294): skip the rest of an empty leaf
                    const float tf0 = to_f32(ff0 * inv0), tf1 = to_f32(ff1 * inv1), tf2 = to_f32(ff2 * inv2);  // (face - origin) * invDir
                    // JS: tExit = Math.min(...) is NaN when an operand is, and then no comparison holds: skip 0; a NaN tEnter is ignored
                    const bool tf_nan = (tf0 != tf0) | (tf1 != tf1) | (tf2 != tf2);
                    const float xt = __builtin_fminf(__builtin_fminf(tf0, tf1), tf2);
                    bool bad = tf_nan | (xt < 0.0f);
                    const double ahead = static_cast<double>(xt) - t;
                    if (!(ahead > near_margin)) {  // tExit is not clearly ahead (or NaN): the near faces decide (octree.ts:215)
                        const float tn0 = to_f32(ld_off<double>(frame, nb + (far0 ^ 32u)) * inv0), tn1 = to_f32(ld_off<double>(frame, nb + (far1 ^ 32u)) * inv1),
                                    tn2 = to_f32(ld_off<double>(frame, nb + (far2 ^ 32u)) * inv2);
                        const bool tn_nan = (tn0 != tn0) | (tn1 != tn1) | (tn2 != tn2);
                        const float et = __builtin_fmaxf(__builtin_fmaxf(tn0, tn1), tn2);
                        bad |= !tn_nan & (et > xt);
                    }
                    double step = __builtin_fmax(ahead, 0.0);  // Math.max(0, tExit - t)
                    step = step < cap ? step : cap;
                    if (!bad && step > 0.0) {
                        RM_CNT1(1)
                        t += step + 0.001;
                        if (t > RM_MAX_DIST) phase = PH_DONE;  // `break`: depth >= MAX_DIST, no normal
                        evaluated = false;                      // `continue`
                        // Skip chain: a skip that minDistance cut short leaves the ray INSIDE this empty leaf, and the next trips of
                        // the reference's loop find the same node and compute the same tExit -- only t differs.  While the
                        // new t is clearly short of tExit (by near_margin: the march point then lies inside the box by more than
                        // its binary32 rounding on every axis, so findNode returns this leaf again, and tEnter, which only shrinks
                        // relative to t, cannot exceed tExit), the next skip needs neither the cell table nor the node: it is
                        // min(tExit - t, cap) + 0.001 again.  Each is a trip of the march loop (sphereTracer.ts:43,59-64).
                        while (phase != PH_DONE && phase < (RM_MAX_STEPS << 3)) {
                            const double ahead2 = static_cast<double>(xt) - t;
                            if (!(ahead2 > near_margin)) break;
                            RM_CNT1(3)
                            phase += 8;
                            const double step2 = ahead2 < cap ? ahead2 : cap;  // ahead2 > 0; cap > 0 (step > 0 above)
                            t += step2 + 0.001;
                            if (t > RM_MAX_DIST) phase = PH_DONE;
                        }
                    }
                }
                if (evaluated) {
                    if (prim_count > 0) {
                        RM_CNT1(2)
                        counters += static_cast<uint32_t>(prim_count) << 16;
                        const uint32_t rb = static_cast<uint32_t>(prim_first) * 32u;  // the leaf's records (RmSphereRec, 32 B)
                        const float inf = __builtin_inff();
                        int k1 = 0;
                        float hi1 = inf, lb1 = inf, lb2 = inf;
                        auto scan = [&](const float4 c, int k) {  // as recs_min: the smallest upper bound, and the smallest lower bound of the others
                            const float dx = p.x - c.x, dy = p.y - c.y, dz = p.z - c.z;
                            const float len = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
                            const float err = (len + __builtin_fabsf(c.w) + 1.0f) * 4e-6f;  // sphere_sdf_estimate
                            const float a = len - c.w, lb = a - err, hi = a + err;
                            const bool better = hi < hi1;
                            const float other = better ? lb1 : lb;  // (no NaN here: finite point, finite spheres)
                            lb2 = other < lb2 ? other : lb2;
                            k1 = better ? k : k1;
                            lb1 = better ? lb : lb1;
                            hi1 = better ? hi : hi1;
                        };
                        const int sub_first = __builtin_bit_cast(int32_t, sub.w);
                        if (P.oct_sub_hdr && sub_first >= 0) {  // crowded leaf: the candidates of p's sub-cell
                            const float lo0 = ld_off<float>(nodes, nb), lo1 = ld_off<float>(nodes, nb + 4u), lo2 = ld_off<float>(nodes, nb + 8u);
                            const int sx = min(max(static_cast<int>((p.x - lo0) * sub.x), 0), RM_OCT_SUB - 1);
                            const int sy = min(max(static_cast<int>((p.y - lo1) * sub.y), 0), RM_OCT_SUB - 1);
                            const int sz = min(max(static_cast<int>((p.z - lo2) * sub.z), 0), RM_OCT_SUB - 1);
                            const uint32_t sc = __umul24(__umul24(static_cast<uint32_t>(sz), RM_OCT_SUB) + static_cast<uint32_t>(sy), RM_OCT_SUB) +
                                                static_cast<uint32_t>(sx);  // 24-bit multiplies: full rate, v_mul_lo_u32 is not
                            const uint32_t hdr = ld_off<uint32_t>(P.oct_sub_hdr, (static_cast<uint32_t>(sub_first) + sc) * 4u);
                            const uint32_t lb0 = hdr >> 8;
                            const int n_sub = static_cast<int>(hdr & 0xFFu);
                            for (int e = 0; e < n_sub; ++e) {
                                RM_CNT1(6)
                                const int j = ld_off<uint8_t>(P.oct_sub_list, lb0 + static_cast<uint32_t>(e));
                                scan(ld_off<float4>(P.oct_recs, rb + static_cast<uint32_t>(j) * 32u), j);
                            }
                        } else {
                            for (int k = 0; k < prim_count; ++k) {
                                RM_CNT1(7)
                                scan(ld_off<float4>(P.oct_recs, rb + static_cast<uint32_t>(k) * 32u), k);
                            }
                        }
                        RM_CNT1(8)
                        const float4 c = ld_off<float4>(P.oct_recs, rb + static_cast<uint32_t>(k1) * 32u);
                        const double e =
                            vec3_length(p.x - c.x, p.y - c.y, p.z - c.z) - ld_off<double>(P.oct_recs, rb + static_cast<uint32_t>(k1) * 32u + 16u);
                        dist = e < dist ? e : dist;
                        if (lb2 <= f32_upper_bound(dist)) {
                            RM_CNT1(9)
                            dist = oct_leaf_rescan(P.oct_recs + prim_first, prim_count, p.x, p.y, p.z, dist);
                        }
                    } else if (pc_ie.y) {
                        RM_CNT1(11)
                        dist = min_dist(cap, dist);  // Math.min(closest, minDistance * safety) (scene.ts:160-163)
                    }
                }
            } else {
                RM_CNT1(12)
                counters += static_cast<uint32_t>(P.n_prims) << 16;
                dist = oct_outside_distance(P.spheres, P.radii, P.n_prims, P.filter != 0, p.x, p.y, p.z);
            }
            if (evaluated) {
                if ((phase & 7) == PH_MARCH) {
                    t += dist;
                    counters += 1;
                    if (dist < RM_EPSILON || t > RM_MAX_DIST) phase = (t >= RM_MAX_DIST) ? PH_DONE : PH_N0;  // raymarcher.ts:97-99
                } else if (phase == PH_N0) {
                    RM_CNT1(13)
                    d0 = dist;
                    phase = PH_N1;
                } else if (phase == PH_N1) {
                    nx = to_f32(d0 - dist);
                    phase = PH_N2;
                } else if (phase == PH_N2) {
                    ny = to_f32(d0 - dist);
                    phase = PH_N3;
                } else {
                    nz = to_f32(d0 - dist);
                    normalize3(nx, ny, nz);
                    phase = PH_DONE;
                }
            }
        }
    }
    if (in_frame) store_pixel(P, static_cast<size_t>(row) * P.width + x, t, nx, ny, nz, counters >> 16, counters & 0xFFFFu);
    v1_diag_epilogue(P, in_frame, counters >> 16, counters & 0xFFFFu);
}

#endif  // !RM_RTC

template <int ACCEL, bool OTHER, int GEN>
__device__ __forceinline__ void render_body(const RmRenderParams &P) {
    // wave tile: tile_w x (64 / tile_w); the waves of a workgroup (blockDim / 64: option `v1_block`) stacked vertically
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tw = P.tile_w, th = 64 / tw;
    const int tiles_x = (P.width + tw - 1) / tw;
    const int tile = v1_tile_of_block(static_cast<int>(blockIdx.x));
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int x = bx * tw + (lane % tw);
    const int row = by * (static_cast<int>(blockDim.x >> 6) * th) + wave * th + (lane / tw);  // tile-local row
    const int rows = P.local_rows;
#ifdef RM_COUNTS
    if (threadIdx.x == 0) rm_cnt_g = P.stamps;
    __syncthreads();
#endif
    const bool in_frame = x < P.width && row < rows;  // (no early return: the diagnostics flush at the end takes every wave)
    uint16_t c16 = 0, i16 = 0;
    if (in_frame) {
    RM_CNT1(14)
    const int y = row_to_y(P, row);
    const size_t idx = static_cast<size_t>(row) * P.width + x;

    // raymarcher.ts:73,83-88
    const double v = (static_cast<double>(y) / static_cast<double>(P.height) - 0.5) * 2.0;
    const double u = (static_cast<double>(x) / static_cast<double>(P.width) - 0.5) * 2.0;
    const double ax = to_f32(u), ay = to_f32(v), az = -1.0;
    float dx = to_f32(ax * P.rot[0] + ay * P.rot[3] + az * P.rot[6]);
    float dy = to_f32(ax * P.rot[1] + ay * P.rot[4] + az * P.rot[7]);
    float dz = to_f32(ax * P.rot[2] + ay * P.rot[5] + az * P.rot[8]);
    double len = static_cast<double>(dx) * dx + static_cast<double>(dy) * dy + static_cast<double>(dz) * dz;
    if (len > 0) len = 1 / __builtin_sqrt(len);
    Ray ray;
    ray.d.x = to_f32(dx * len);
    ray.d.y = to_f32(dy * len);
    ray.d.z = to_f32(dz * len);
    ray.o.x = P.origin[0];
    ray.o.y = P.origin[1];
    ray.o.z = P.origin[2];
    ray.od[0] = P.origin_d[0];
    ray.od[1] = P.origin_d[1];
    ray.od[2] = P.origin_d[2];

    uint32_t count = 0, iters = 0;
    const double depth = OTHER ? ray_march_other<ACCEL, GEN>(P, ray, count, iters) : ray_march<ACCEL, GEN>(P, ray, count, iters);
    uint8_t nb[3];
    normal_and_store<ACCEL, GEN>(P, ray, depth, count, nb);
    const uint8_t db = u8clamp(depth);
    c16 = static_cast<uint16_t>(count & 0xFFFFu);  // Uint16Array += wraps
    i16 = static_cast<uint16_t>(iters & 0xFFFFu);
    if (P.depth) P.depth[idx] = db;
    if (P.normal) {
        P.normal[3 * idx] = nb[0];
        P.normal[3 * idx + 1] = nb[1];
        P.normal[3 * idx + 2] = nb[2];
    }
    if (P.sdf) P.sdf[idx] = c16;
    if (P.iters) P.iters[idx] = i16;
    if (P.rgba) reinterpret_cast<uchar4 *>(P.rgba)[idx] = shade_pixel(P.shader, db, nb[0], nb[1], nb[2], c16, i16, P.light_d);
    }
    v1_diag_epilogue(P, in_frame, c16, i16);
}

template <int ACCEL, int GEN>
__device__ __forceinline__ void distance_body(const RmRenderParams &P, const float *pts, int64_t n, double *dist, uint32_t *count) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Vec3f p{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    uint32_t c = 0;
    dist[i] = scene_distance<ACCEL, GEN>(P, p, c);
    count[i] = c;
}

#ifdef RM_RTC
}  // namespace
extern "C" __global__ __launch_bounds__(256) void rm_rtc_render(const RmRenderParams P) { render_body<RM_RTC_ACCEL, RM_RTC_OTHER != 0, 4>(P); }
extern "C" __global__ __launch_bounds__(256) void rm_rtc_distance(const RmRenderParams P, const float *pts, int64_t n, double *dist, uint32_t *count) {
    distance_body<RM_RTC_ACCEL, 4>(P, pts, n, dist, count);
}
#else
template <int ACCEL, bool OTHER, int GEN>
__global__ __launch_bounds__(256) void render_kernel(const RmRenderParams P) {
    render_body<ACCEL, OTHER, GEN>(P);
}

#ifndef RM_LENGTH_SQRT  // scene-independent kernels exist once (this file is compiled a second time with -DRM_LENGTH_SQRT)
// ------------------------------------------------------------------ small kernels

// The octree's boxes relative to one camera position (render_kernel_oct): intersectRayBox computes (bounds - origin) * invDir
// per ray and axis (octree.ts:200-203); the difference does not depend on the ray.
__global__ __launch_bounds__(256) void oct_frame_table_kernel(const RmOctNode *nodes, int n, double ox, double oy, double oz,
                                                              RmOctFrameNode *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const RmOctNode nd = nodes[i];
    RmOctFrameNode f;
    f.lo[0] = static_cast<double>(nd.lo[0]) - ox;
    f.lo[1] = static_cast<double>(nd.lo[1]) - oy;
    f.lo[2] = static_cast<double>(nd.lo[2]) - oz;
    f.cap = nd.min_distance * 0.99;
    f.hi[0] = static_cast<double>(nd.hi[0]) - ox;
    f.hi[1] = static_cast<double>(nd.hi[1]) - oy;
    f.hi[2] = static_cast<double>(nd.hi[2]) - oz;
    f.pad = 0.0;
    out[i] = f;
}

__global__ __launch_bounds__(256) void shade_kernel(int shader, int64_t n, const uint8_t *depth, const uint8_t *normal,
                                                    const uint16_t *sdf, const uint16_t *iters, uchar4 *rgba, float l0,
                                                    float l1, float l2) {
    const double light[3] = {l0, l1, l2};
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        rgba[i] = shade_pixel(shader, depth[i], normal[3 * i], normal[3 * i + 1], normal[3 * i + 2], sdf[i], iters[i],
                              light);
    }
}

__global__ __launch_bounds__(256) void reduce_kernel(const uint16_t *sdf, const uint16_t *iters, int64_t n,
                                                     RmDiagDevice *acc) {
    unsigned long long s = 0, it = 0;
    unsigned int mx = 0, mn = 0xFFFFFFFFu;
    const int64_t tid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    const int64_t nthreads = static_cast<int64_t>(gridDim.x) * blockDim.x;
    int64_t done = 0;
    // 16-byte loads (8 counters per lane per load) when both streams are 16-byte aligned
    if (((reinterpret_cast<uintptr_t>(sdf) | reinterpret_cast<uintptr_t>(iters)) & 15) == 0) {
        const int64_t nvec = n / 8;
        const uint4 *s4 = reinterpret_cast<const uint4 *>(sdf);
        const uint4 *i4 = reinterpret_cast<const uint4 *>(iters);
        auto eat = [&](const uint4 a, const uint4 b) {
            const unsigned int aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned int lo = aw[k] & 0xFFFFu, hi = aw[k] >> 16;
                s += lo + hi;
                mx = lo > mx ? lo : mx;
                mx = hi > mx ? hi : mx;
                mn = lo < mn ? lo : mn;
                mn = hi < mn ? hi : mn;
                it += (bw[k] & 0xFFFFu) + (bw[k] >> 16);
            }
        };
        int64_t v = tid;
        for (; v + 3 * nthreads < nvec; v += 4 * nthreads) {  // eight 16-B loads in flight per lane
            const uint4 a0 = s4[v], b0 = i4[v], a1 = s4[v + nthreads], b1 = i4[v + nthreads];
            const uint4 a2 = s4[v + 2 * nthreads], b2 = i4[v + 2 * nthreads], a3 = s4[v + 3 * nthreads], b3 = i4[v + 3 * nthreads];
            eat(a0, b0);
            eat(a1, b1);
            eat(a2, b2);
            eat(a3, b3);
        }
        for (; v < nvec; v += nthreads) eat(s4[v], i4[v]);
        done = nvec * 8;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) {
        const unsigned int c = sdf[i];
        s += c;
        it += iters[i];
        mx = c > mx ? c : mx;
        mn = c < mn ? c : mn;
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off);
        it += __shfl_down(it, off);
        const unsigned int omx = __shfl_down(mx, off), omn = __shfl_down(mn, off);
        mx = omx > mx ? omx : mx;
        mn = omn < mn ? omn : mn;
    }
    // one atomic set per workgroup: the four words share a cache line, so per-wave atomics from
    // thousands of waves serialise (measured: 0.39 ms for 8192 waves vs the ~15 us the loads take)
    __shared__ unsigned long long sh_s[4], sh_it[4];
    __shared__ unsigned int sh_mx[4], sh_mn[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh_s[w] = s;
        sh_it[w] = it;
        sh_mx[w] = mx;
        sh_mn[w] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            s += sh_s[k];
            it += sh_it[k];
            mx = sh_mx[k] > mx ? sh_mx[k] : mx;
            mn = sh_mn[k] < mn ? sh_mn[k] : mn;
        }
        atomicAdd(&acc->total_sdf, s);
        atomicAdd(&acc->total_iters, it);
        atomicMax(&acc->max_sdf, mx);
        atomicMin(&acc->min_sdf, mn);
    }
}

__global__ void reduce_init_kernel(RmDiagDevice *acc) {
    acc->total_sdf = 0;
    acc->total_iters = 0;
    acc->max_sdf = 0;
    acc->min_sdf = 0xFFFFFFFFu;
    acc->pad = 0;
}

#endif  // !RM_LENGTH_SQRT

template <int ACCEL, int GEN>
__global__ __launch_bounds__(256) void distance_kernel(const RmRenderParams P, const float *pts, int64_t n, double *dist,
                                                       uint32_t *count) {
    distance_body<ACCEL, GEN>(P, pts, n, dist, count);
}

#ifndef RM_LENGTH_SQRT
__global__ __launch_bounds__(256) void hypot_kernel(const float *xyz, int64_t n, double *out) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    out[i] = hypot3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
}

// hypot3 (compiler's IEEE division) vs hypot3_shared_rcp on counter-generated binary32
// triples: random mantissas, exponents spread over [2^-149, 2^60], zeros, equal and
// near-equal magnitudes.  Counts bitwise mismatches.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ float gen_f32(uint64_t r, int mode) {
    const uint32_t mant = static_cast<uint32_t>(r) & 0x7FFFFFu;
    const uint32_t sign = static_cast<uint32_t>(r >> 63) << 31;
    uint32_t exp;
    const uint32_t sel = static_cast<uint32_t>(r >> 40) & 0xFFu;
    if (mode == 0) exp = 96 + sel % 64;        // around 1: 2^-31 .. 2^32
    else if (mode == 1) exp = sel % 188;       // whole low range incl. denormals (exp 0)
    else exp = 120 + sel % 10;                 // same binade neighbourhood
    if ((r >> 50 & 0x3F) == 0) return __uint_as_float(sign);  // +-0
    return __uint_as_float(sign | (exp << 23) | mant);
}

__global__ __launch_bounds__(256) void fastdiv_selftest_kernel(uint64_t seed, int64_t n, unsigned long long *mismatches) {
    unsigned long long bad = 0;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const uint64_t base = seed + 3ull * static_cast<uint64_t>(i);
        const int mode = static_cast<int>(i % 3);
        float a = gen_f32(mix64(base), mode), b = gen_f32(mix64(base + 1), mode), c = gen_f32(mix64(base + 2), mode);
        if ((i & 15) == 7) b = a;                     // equal magnitudes: quotient exactly 1
        if ((i & 31) == 19) c = -a;
        if ((i & 63) == 33) b = __uint_as_float(__float_as_uint(a) + 1);  // one ulp apart
        const double x = hypot3(a, b, c), y = hypot3_shared_rcp(a, b, c);
        if (__double_as_longlong(x) != __double_as_longlong(y)) bad++;
    }
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(mismatches, bad);
}

// mode 0: 1.0 / d for every finite non-zero binary32 d (2^32 bit patterns); mode 1: x / W for 0 <= x < 2^16, 1 <= W < 2^16
__global__ __launch_bounds__(256) void recip_selftest_kernel(int mode, unsigned long long *mismatches) {
    unsigned long long bad = 0;
    for (uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < (1ull << 32);
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        double a, b;
        if (mode == 0) {
            const float d = __uint_as_float(static_cast<uint32_t>(i));
            if (!(__builtin_fabsf(d) <= 3.4028234663852886e38f) || d == 0.0f) continue;
            a = 1.0;
            b = static_cast<double>(d);
        } else {
            const uint32_t x = static_cast<uint32_t>(i >> 16), w = static_cast<uint32_t>(i & 0xFFFFu);
            if (w == 0) continue;
            a = static_cast<double>(x);
            b = static_cast<double>(w);
        }
        // opaque to the optimiser, so that the reference quotient is the compiler's full IEEE expansion
        asm volatile("" : "+v"(a), "+v"(b));
        const double want = a / b, got = div_in_range(a, b);
        if (__double_as_longlong(want) != __double_as_longlong(got)) bad++;
    }
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(mismatches, bad);
}

#endif  // !RM_LENGTH_SQRT

}  // namespace

#ifndef RM_LENGTH_SQRT
hipError_t rm_launch_recip_selftest(int mode, unsigned long long *d_mismatches, hipStream_t stream) {
    hipLaunchKernelGGL(recip_selftest_kernel, dim3(4096), dim3(256), 0, stream, mode, d_mismatches);
    return hipGetLastError();
}

hipError_t rm_launch_fastdiv_selftest(uint64_t seed, int64_t n, unsigned long long *d_mismatches, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fastdiv_selftest_kernel, dim3(2048), dim3(256), 0, stream, seed, n, d_mismatches);
    return hipGetLastError();
}

#endif  // !RM_LENGTH_SQRT

// ---------------------------------------------------------------------- launchers

hipError_t RM_LEN_VARIANT(rm_launch_render)(const RmRenderParams &p, hipStream_t stream, const char **kernel_name) {
    const int rows = p.local_rows;
    if (kernel_name) *kernel_name = "";
    if (rows <= 0 || p.width <= 0) return hipSuccess;
    if (p.variant == 2 && p.algorithm == 0) return RM_LEN_VARIANT(rm_launch_render_v2)(p, stream, kernel_name);
    const int tw = p.tile_w, th = 64 / tw;
    const int wpw = p.v1_block >= 256 ? 4 : (p.v1_block >= 128 ? 2 : 1);  // waves per workgroup (option `v1_block`)
    const int threads = 64 * wpw;
    const int tiles_x = (p.width + tw - 1) / tw;
    const int tiles_y = (rows + wpw * th - 1) / (wpw * th);
    const dim3 grid((static_cast<unsigned>(tiles_x) * static_cast<unsigned>(tiles_y) + 63u) & ~63u), block(static_cast<unsigned>(threads));  // v1_tile_of_block
    // expression programs keep their position slots and pending values in LDS (rm_program.h)
    size_t shmem = p.general >= 2 && !p.rtc_function ? (static_cast<size_t>(p.prog_slots) * 12 + static_cast<size_t>(p.prog_vals) * 8) * threads : 0;
    RmRenderParams pl = p;
    pl.v1_list_offset = -1;
    const size_t list_bytes = static_cast<size_t>(2) * RM_V1_LIST_CAP * threads;
    if (p.accel == 2 && p.v1_lists && p.general < 2 && !p.rtc_function && p.bvh_nodes < 65536 && shmem + list_bytes <= 64 * 1024) {
        pl.v1_list_offset = static_cast<int32_t>((shmem + 15) & ~static_cast<size_t>(15));  // per-ray hit-leaf lists behind the program slots
        shmem = static_cast<size_t>(pl.v1_list_offset) + list_bytes;
    }
    if (p.rtc_function) {  // this scene's run-time specialised kernel (rm_rtc.h): same grid, nothing in LDS (set by the API layer for these launches only)
        size_t bytes = sizeof pl;
        void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pl, HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END};
        return hipModuleLaunchKernel(reinterpret_cast<hipFunction_t>(const_cast<void *>(p.rtc_function)), grid.x, 1, 1, block.x, 1, 1, 0, stream, nullptr,
                                     extra);
    }
#define RM_V1(A, O, G)                                                                       \
    {                                                                                        \
        hipLaunchKernelGGL((render_kernel<A, O, G>), grid, block, shmem, stream, pl);         \
        if (kernel_name) *kernel_name = "render_kernel<" #A ", " #O ", " #G ">" RM_LEN_TAG;  \
    }
#define RM_V1A(O, G) { if (p.accel == 2) RM_V1(2, O, G) else if (p.accel == 1) RM_V1(1, O, G) else RM_V1(0, O, G) }
    if (p.accel == 1 && p.general == 0 && p.algorithm == 0 && p.oct_lean && p.oct_frame && p.oct_lut && p.oct_recs && p.filter && p.oct_nodes < (1 << 24) &&
        p.oct_prim_count < (1 << 26)) {  // 32-bit byte offsets into the node and record tables
        // one wave per workgroup, 8 x 8 pixels: wave slots refill one by one (four-wave workgroups wait for a free slot on every
        // SIMD: 3.10 -> 2.89 ms on the 10 000-sphere frame), and a square tile keeps the rays of a wave in the same leaves
        pl.tile_w = 8;
        const int ty = (rows + 7) / 8;
        hipLaunchKernelGGL(render_kernel_oct, dim3((static_cast<unsigned>((p.width + 7) / 8) * static_cast<unsigned>(ty) + 63u) & ~63u), dim3(64), 0, stream,
                           pl);  // the lean octree sphere tracer (finite camera: rm_api.cpp)
        if (kernel_name) *kernel_name = "render_kernel_oct" RM_LEN_TAG;
    } else if (p.general == 3) {
        if (p.algorithm == 0) RM_V1A(false, 3) else RM_V1A(true, 3)
    } else if (p.general == 2) {
        if (p.algorithm == 0) RM_V1A(false, 2) else RM_V1A(true, 2)
    } else if (p.general) {
        if (p.algorithm == 0) RM_V1A(false, 1) else RM_V1A(true, 1)
    } else {
        if (p.algorithm == 0) RM_V1A(false, 0) else RM_V1A(true, 0)
    }
#undef RM_V1A
#undef RM_V1
    return hipGetLastError();
}

#ifndef RM_LENGTH_SQRT
hipError_t rm_launch_oct_frame_table(const RmOctNode *nodes, int n, const double origin[3], RmOctFrameNode *out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(oct_frame_table_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, nodes, n, origin[0], origin[1],
                       origin[2], out);
    return hipGetLastError();
}

hipError_t rm_launch_shade(int shader, int64_t n, const uint8_t *depth, const uint8_t *normal, const uint16_t *sdf,
                           const uint16_t *iters, uint8_t *rgba, const float light[3], hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(shade_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, shader, n, depth,
                       normal, sdf, iters, reinterpret_cast<uchar4 *>(rgba), light[0], light[1], light[2]);
    return hipGetLastError();
}

hipError_t rm_launch_reduce_init(RmDiagDevice *acc, hipStream_t stream) {
    hipLaunchKernelGGL(reduce_init_kernel, dim3(1), dim3(1), 0, stream, acc);
    return hipGetLastError();
}

hipError_t rm_launch_reduce(const uint16_t *sdf, const uint16_t *iters, int64_t n, RmDiagDevice *acc,
                            hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 128) blocks = 128;  // one atomic set per workgroup on one cache line: they serialise (256 workgroups: 17.6 us for init +
                                     // reduce of a 4K frame, 128: 12.6 us, of which two launches are ~12; 64 x 1024 threads: 12.0 us)
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(reduce_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, sdf, iters, n, acc);
    return hipGetLastError();
}

#endif  // !RM_LENGTH_SQRT

hipError_t RM_LEN_VARIANT(rm_launch_distance)(const RmRenderParams &p, const float *points, int64_t n, double *dist, uint32_t *count,
                              hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const dim3 grid(static_cast<unsigned>((n + 255) / 256)), block(256);
    const size_t shmem = p.general >= 2 ? (static_cast<size_t>(p.prog_slots) * 12 + static_cast<size_t>(p.prog_vals) * 8) * 256 : 0;
    if (p.rtc_function) {  // rm_rtc_distance(RmRenderParams, const float *, int64_t, double *, uint32_t *)
        struct Args {
            RmRenderParams p;
            const float *points;
            int64_t n;
            double *dist;
            uint32_t *count;
        } args{p, points, n, dist, count};
        static_assert(sizeof(RmRenderParams) % 8 == 0 && alignof(RmRenderParams) == 8, "kernel argument layout");
        size_t bytes = sizeof args;
        void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END};
        return hipModuleLaunchKernel(reinterpret_cast<hipFunction_t>(const_cast<void *>(p.rtc_function)), grid.x, 1, 1, 256, 1, 1, 0, stream, nullptr, extra);
    }
#define RM_DK(A, G) hipLaunchKernelGGL((distance_kernel<A, G>), grid, block, shmem, stream, p, points, n, dist, count)
    if (p.general == 3) { if (p.accel == 2) RM_DK(2, 3); else if (p.accel == 1) RM_DK(1, 3); else RM_DK(0, 3); }
    else if (p.general == 2) { if (p.accel == 2) RM_DK(2, 2); else if (p.accel == 1) RM_DK(1, 2); else RM_DK(0, 2); }
    else if (p.general) { if (p.accel == 2) RM_DK(2, 1); else if (p.accel == 1) RM_DK(1, 1); else RM_DK(0, 1); }
    else { if (p.accel == 2) RM_DK(2, 0); else if (p.accel == 1) RM_DK(1, 0); else RM_DK(0, 0); }
#undef RM_DK
    return hipGetLastError();
}

#ifndef RM_LENGTH_SQRT
__global__ __launch_bounds__(256) void jsmath_kernel(int fn, const double *a, const double *b, int64_t n, double *out) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b[i];
    double r;
    switch (fn) {
        case 0: r = js_sin(x); break;
        case 1: r = js_cos(x); break;
        case 2: r = js_atan2(x, y); break;
        case 3: r = js_asin(x); break;
        case 4: r = js_log(x); break;
        case 5: r = js_pow(x, y); break;
        case 6: r = js_round(x); break;
        default: r = js_atan(x); break;
    }
    out[i] = r;
}

hipError_t rm_launch_jsmath(int fn, const double *a, const double *b, int64_t n, double *out, hipStream_t stream) {
    hipLaunchKernelGGL(jsmath_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, fn, a, b, n, out);
    return hipGetLastError();
}

hipError_t rm_launch_hypot(const float *xyz, int64_t n, double *out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(hypot_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, xyz, n, out);
    return hipGetLastError();
}
#endif  // !RM_LENGTH_SQRT
#endif  // !RM_RTC
