// rm_bvh_list.h -- per-ray BVH interval bookkeeping shared by the v1 (rm_kernels.hip) and v2 (rm_render_v2.hip) kernels:
// the reference materialises every leaf interval of a ray, stable-sorts them by tEnter (bvh.ts:126-178) and walks the
// list with an index (bvh.ts:204-240).  Here BVH.onRayMarchStart records the LEAVES the ray hits (node ids, 2 B each, in
// an LDS list laid out entry x 64 + lane so one lane's column is bank-conflict free) and "the next interval in
// stable-sort order" is a scan of that short list that recomputes the slab test of the listed leaves (same arithmetic,
// same values).  A ray that hits more leaves than the list holds walks the tree again instead -- same results.
#pragma once
#include "rm_device.h"

#ifndef RM_CNT
#define RM_CNT(i) {}
#define RM_BVH_LIST_OWN_CNT
#endif

namespace rmd {

struct SceneView {
    const RmBvhNode *nodes;
    const int32_t *bvh_prims;
    const RmOctNode *oct;
    const int32_t *oct_prims;
    const RmSphere *spheres;
    const double *radii;
    const uint32_t *pq_cells;
    const uint16_t *pq_list;
    const uint32_t *nn_cells;
    const uint16_t *nn_list;
    const double *rel;  // REL kernels: per BVH node {lo - origin, hi - origin} as doubles (LDS, built per launch)
    int n_prims, bvh_nodes;
};

struct RayList {
    uint16_t *col;  // this lane's column: entry e at col[e * 64]
    int cap;
    int cnt;        // leaves hit (may exceed cap -> overflow)
    int live;       // entries still listed: the current interval's entry and the ones after it
    int cur_pos;    // list position of the current interval's entry (bvh_next removes it first)
};

// BVH.onRayMarchStart (bvh.ts:181-202): one traversal; records hit leaves, returns the first
// interval in sorted order (min tEnter, ties: first in traversal order)
template <bool REL>
__device__ __forceinline__ bool node_slab(const SceneView &S, const RmBvhNode &node, int i, const Ray &r, const RayInv &ri,
                                          double &tE, double &tX) {
    if (REL) return slab_rel(S.rel + 6 * i, ri, tE, tX);
    return slab_inv(node.lo, node.hi, r, ri, tE, tX);
}

template <bool REL>
__device__ bool bvh_prologue(const SceneView &S, const Ray &r, const RayInv &ri, RayList &L, Interval &first) {
    bool have = false;
    int i = 0;
    const int n = S.bvh_nodes;
    L.cnt = 0;
    while (i < n) {
        RM_CNT(5)
        const RmBvhNode node = S.nodes[i];
        double tE, tX;
        if (!node_slab<REL>(S, node, i, r, ri, tE, tX) || tX < 0.0 || tE > RM_MAX_DIST) {  // bvh.ts:145,151
            i = node.skip;
            continue;
        }
        if (node.leaf < 0) {
            i = i + 1;
            continue;
        }
        if ((node.leaf & 0xFF) > 0) {  // bvh.ts:165
            const double cE = __builtin_fmax(tE, 0.0);
            const double cX = __builtin_fmin(tX, RM_MAX_DIST);
            if (L.cnt < L.cap) L.col[L.cnt * 64] = static_cast<uint16_t>(i);
            L.cnt++;
            if (!have || cE < first.tEnter) {
                first.tEnter = cE;
                first.tExit = cX;
                first.ord = i;
                have = true;
                L.cur_pos = L.cnt - 1;
            }
        }
        i = node.skip;
    }
    L.live = L.cnt;
    return have;
}

// successor of key (keyT, keyOrd) in the stable-sorted interval order of bvh.ts:176
template <bool REL>
__device__ bool bvh_next(const SceneView &S, const Ray &r, const RayInv &ri, RayList &L, double keyT, int keyOrd,
                         Interval &out) {
    bool have = false;
    RM_CNT(3)
    if (L.cnt <= L.cap) {
        // The list holds the current interval's entry and the entries AFTER it in the order of bvh.ts:176 -- nothing
        // else: the first interval is the minimum of all of them, every later one the minimum of what remained.  So the
        // successor is the minimum of the list without the current entry; that entry is removed by position (the last
        // entry takes its place) before the scan, no entry needs the `after` test, and the list a ray scans shrinks by
        // one per advance: c (c - 1) / 2 slab tests over a ray's life instead of c^2.  The list is not in traversal order
        // any more, so ties between equal tEnter go to the smaller node index explicitly (traversal order = increasing
        // index: the stackless walk only moves forward).
        L.live -= 1;
        if (L.cur_pos != L.live) L.col[L.cur_pos * 64] = L.col[L.live * 64];
        for (int e = 0; e < L.live; ++e) {
            RM_CNT(4)
            const int id = L.col[e * 64];
            const RmBvhNode node = S.nodes[id];
            double tE, tX;
            node_slab<REL>(S, node, id, r, ri, tE, tX);  // hit by construction; same arithmetic, same values
            const double cE = __builtin_fmax(tE, 0.0);
            const double cX = __builtin_fmin(tX, RM_MAX_DIST);
            if (!have || cE < out.tEnter || (cE == out.tEnter && id < out.ord)) {
                out.tEnter = cE;
                out.tExit = cX;
                out.ord = id;
                have = true;
                L.cur_pos = e;
            }
        }
        return have;
    }
    int i = 0;  // overflow: the list is incomplete, walk the tree again
    const int n = S.bvh_nodes;
    while (i < n) {
        const RmBvhNode node = S.nodes[i];
        double tE, tX;
        if (!node_slab<REL>(S, node, i, r, ri, tE, tX) || tX < 0.0 || tE > RM_MAX_DIST) {
            i = node.skip;
            continue;
        }
        if (node.leaf < 0) {
            i = i + 1;
            continue;
        }
        if ((node.leaf & 0xFF) > 0) {
            const double cE = __builtin_fmax(tE, 0.0);
            const double cX = __builtin_fmin(tX, RM_MAX_DIST);
            const bool after = cE > keyT || (cE == keyT && i > keyOrd);
            if (after && (!have || cE < out.tEnter)) {
                out.tEnter = cE;
                out.tExit = cX;
                out.ord = i;
                have = true;
            }
        }
        i = node.skip;
    }
    return have;
}


}  // namespace rmd

#ifdef RM_BVH_LIST_OWN_CNT
#undef RM_CNT
#undef RM_BVH_LIST_OWN_CNT
#endif
