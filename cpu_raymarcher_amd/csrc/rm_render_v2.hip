// rm_render_v2.hip -- render kernel, second generation, for gfx950 (MI355X, CDNA4).
//
// Same results as the v1 kernel (rm_kernels.hip) bit for bit; different execution shape:
//
//  * Uniform wave loop.  A ray's life is: BVH prologue -> march steps -> 4 normal samples ->
//    stores.  The expensive unit common to march steps and normal samples is one
//    Scene.getDistance(point).  Every iteration of the wave loop lets each lane do its cheap
//    bookkeeping (interval state machine, empty-space skips) until it needs a distance, then
//    all lanes that need one evaluate it together.  Lanes in different phases of their rays
//    therefore still share the heavy instructions.
//  * Per-ray hit-leaf list in LDS.  The prologue traverses the BVH once and records the
//    leaves the ray hits (node ids, 2 B each, lane-interleaved so the column of one lane is
//    bank-conflict free).  "Next interval in stable tEnter order" (bvh.ts:176,223-236) is then
//    a scan of that short list (mean 2, max 13 on the dense grid) instead of a traversal.
//    A ray that hits more leaves than the list holds falls back to the full traversal.
//  * Wave-cooperative N-primitive fallback (scene.ts:173): when a point lies in no leaf box
//    the reference evaluates ALL primitives.  The wave serves such lanes one at a time: the
//    point is broadcast (v_readlane), each lane bounds 1/64 of the spheres with a conservative
//    binary32 estimate, a wave min gives an upper bound, and only spheres whose lower bound
//    does not exceed it are evaluated exactly (in parallel, one per lane), followed by a wave
//    min.  min() is order independent and the skipped spheres are provably farther, so the
//    result is bit-identical; the counter still advances by N.
//  * Scene tables (nodes, leaf ids, spheres, radii) staged in LDS when they fit.
//  * XCD-aware tile order: workgroups b, b+8, b+16 ... share an XCD (round-robin dispatch);
//    they are given horizontally adjacent tiles of one tile row so the partial-line stores of
//    neighbouring tiles merge in that XCD's L2 before they reach HBM.
#include <hip/hip_runtime.h>

#include "rm_device.h"
#include "rm_kernels.h"

namespace {

using namespace rmd;

enum Phase : int { PH_MARCH = 0, PH_N0 = 1, PH_N1 = 2, PH_N2 = 3, PH_N3 = 4, PH_DONE = 5 };

struct SceneView {
    const RmBvhNode *nodes;
    const int32_t *bvh_prims;
    const RmOctNode *oct;
    const int32_t *oct_prims;
    const RmSphere *spheres;
    const double *radii;
    int n_prims, bvh_nodes;
};

__device__ __forceinline__ float wave_min_f32(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ float readlane_f32(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// round-up conversion so that the binary32 value is an upper bound of the double
__device__ __forceinline__ float f32_upper(double v) {
    return __double2float_ru(v);
}

// min over an id list (or 0..n-1 when ids == nullptr) of the exact sphere distance, one
// lane alone: exact evaluation only for spheres whose conservative lower bound does not
// exceed the best upper bound so far.  Skipped spheres satisfy exact > best >= result.
__device__ double lane_min_filtered(const SceneView &S, const int32_t *ids, int n, const Vec3f &p, double closest) {
    float ub = f32_upper(closest);
    for (int k = 0; k < n; ++k) {
        const int id = ids ? ids[k] : k;
        const RmSphere s = S.spheres[id];
        float err;
        const float a = sphere_sdf_estimate(s, p, err);
        if (a - err <= ub) {
            const double e = sphere_sdf_fast(s, S.radii[id], p);
            if (e < closest) {
                closest = e;
                ub = f32_upper(e);
            }
        }
    }
    return closest;
}

__device__ double lane_min_exact(const SceneView &S, const int32_t *ids, int n, const Vec3f &p, double closest) {
    for (int k = 0; k < n; ++k) {
        const int id = ids ? ids[k] : k;
        closest = min_dist(sphere_sdf_fast(S.spheres[id], S.radii[id], p), closest);
    }
    return closest;
}

// all primitives for ONE point held by every lane (wave-uniform b); all 64 lanes take part
__device__ double coop_all_prims(const SceneView &S, const Vec3f &b, int lane) {
    float ub = 10.0f;
    for (int j = lane; j < S.n_prims; j += 64) {
        float err;
        const float a = sphere_sdf_estimate(S.spheres[j], b, err);
        const float hi = a + err;
        ub = hi < ub ? hi : ub;
    }
    ub = wave_min_f32(ub);
    double best = RM_MAX_DIST;
    for (int j = lane; j < S.n_prims; j += 64) {
        const RmSphere s = S.spheres[j];
        float err;
        const float a = sphere_sdf_estimate(s, b, err);
        if (a - err <= ub) best = min_dist(sphere_sdf_fast(s, S.radii[j], b), best);
    }
    return wave_min_f64(best);
}

// Scene.getDistance over all primitives (scene.ts:183-189 and the fallback scene.ts:173) for
// the lanes in `need`; must be reached by the whole wave.  Returns this lane's distance.
__device__ double all_prims_wave(const SceneView &S, bool need, const Vec3f &q, int lane, bool coop, bool filter) {
    double closest = RM_MAX_DIST;
    unsigned long long fb = __ballot(need);
    if (fb == 0) return closest;
    const int n = S.n_prims;
    const int m = __popcll(fb);
    // per fallback ray: cooperative ~ 28 * ceil(n/64) + 250 wave instructions, serialised over
    // the m rays; per-lane filtered loop ~ 16 n + 300 once for all lanes
    const bool use_coop = coop && n > 8 && static_cast<long long>(m) * (28 * ((n + 63) / 64) + 250) < 16ll * n + 300;
    if (use_coop) {
        while (fb) {
            const int src = __builtin_ctzll(fb);
            fb &= fb - 1;
            Vec3f b;
            b.x = readlane_f32(q.x, src);
            b.y = readlane_f32(q.y, src);
            b.z = readlane_f32(q.z, src);
            const double r = coop_all_prims(S, b, lane);
            if (lane == src) closest = r;
        }
    } else if (need) {
        closest = (filter && n >= 2) ? lane_min_filtered(S, nullptr, n, q, closest) : lane_min_exact(S, nullptr, n, q, closest);
    }
    return closest;
}

// BVH branch of Scene.getDistance (scene.ts:167-181); whole wave must call
__device__ double bvh_distance_wave(const SceneView &S, bool need, const Vec3f &q, uint32_t &count, int lane, bool coop,
                                    bool filter) {
    double closest = RM_MAX_DIST;
    uint32_t found = 0;
    if (need) {
        int i = 0;
        const int n = S.bvh_nodes;
        while (i < n) {  // BVH.getPrimitivesAt (bvh.ts:95-121), stackless
            const RmBvhNode node = S.nodes[i];
            if (!box_contains(node.lo, node.hi, q)) {
                i = node.skip;
                continue;
            }
            if (node.leaf < 0) {
                i = i + 1;
                continue;
            }
            const int first = node.leaf >> 8, cnt = node.leaf & 0xFF;
            closest = (filter && cnt >= 2) ? lane_min_filtered(S, S.bvh_prims + first, cnt, q, closest)
                                           : lane_min_exact(S, S.bvh_prims + first, cnt, q, closest);
            found += static_cast<uint32_t>(cnt);
            i = node.skip;
        }
    }
    const bool fallback = need && found == 0;
    const double all = all_prims_wave(S, fallback, q, lane, coop, filter);
    if (fallback) {
        closest = all;
        count += static_cast<uint32_t>(S.n_prims);
    } else if (need) {
        count += found;
    }
    return closest;
}

// Octree.findNode (octree.ts:223-248), see rm_kernels.hip
__device__ __forceinline__ int oct_find(const SceneView &S, const Vec3f &p) {
    const RmOctNode *nodes = S.oct;
    if (!box_contains(nodes[0].lo, nodes[0].hi, p)) return -1;
    int i = 0;
    for (;;) {
        const int first = nodes[i].first_child;
        if (first < 0) return i;
        const float cx = nodes[first].hi[0], cy = nodes[first].hi[1], cz = nodes[first].hi[2];
        i = first + (p.x > cx ? 1 : 0) + (p.y > cy ? 2 : 0) + (p.z > cz ? 4 : 0);
    }
}

// Octree branch of Scene.getDistance (scene.ts:148-166) for the node findNode returned
__device__ double oct_distance_lane(const SceneView &S, int node, const Vec3f &q, uint32_t &count, bool filter) {
    if (node < 0) {  // outside the cube: all primitives (scene.ts:166,183-189)
        count += static_cast<uint32_t>(S.n_prims);
        return (filter && S.n_prims >= 2) ? lane_min_filtered(S, nullptr, S.n_prims, q, RM_MAX_DIST)
                                         : lane_min_exact(S, nullptr, S.n_prims, q, RM_MAX_DIST);
    }
    const RmOctNode nd = S.oct[node];
    double closest = RM_MAX_DIST;
    if (nd.prim_count > 0) {
        const int32_t *ids = S.oct_prims + nd.prim_first;
        closest = (filter && nd.prim_count >= 2) ? lane_min_filtered(S, ids, nd.prim_count, q, closest)
                                                : lane_min_exact(S, ids, nd.prim_count, q, closest);
        count += static_cast<uint32_t>(nd.prim_count);
    } else if (nd.is_empty) {
        closest = min_dist(nd.min_distance * 0.99, closest);
    }
    return closest;
}

// ---- per-ray BVH interval bookkeeping ----------------------------------------------------

struct RayList {
    uint16_t *col;  // this lane's column: entry e at col[e * 64]
    int cap;
    int cnt;        // leaves hit (may exceed cap -> overflow)
};

// BVH.onRayMarchStart (bvh.ts:181-202): one traversal; records hit leaves, returns the first
// interval in sorted order (min tEnter, ties: first in traversal order)
__device__ bool bvh_prologue(const SceneView &S, const Ray &r, const RayInv &ri, RayList &L, Interval &first) {
    bool have = false;
    int i = 0;
    const int n = S.bvh_nodes;
    L.cnt = 0;
    while (i < n) {
        const RmBvhNode node = S.nodes[i];
        double tE, tX;
        if (!slab_inv(node.lo, node.hi, r, ri, tE, tX) || tX < 0.0 || tE > RM_MAX_DIST) {  // bvh.ts:145,151
            i = node.skip;
            continue;
        }
        if (node.leaf < 0) {
            i = i + 1;
            continue;
        }
        if ((node.leaf & 0xFF) > 0) {  // bvh.ts:165
            const double cE = tE > 0.0 ? tE : 0.0;
            const double cX = tX < RM_MAX_DIST ? tX : RM_MAX_DIST;
            if (L.cnt < L.cap) L.col[L.cnt * 64] = static_cast<uint16_t>(i);
            L.cnt++;
            if (!have || cE < first.tEnter) {
                first.tEnter = cE;
                first.tExit = cX;
                first.ord = i;
                have = true;
            }
        }
        i = node.skip;
    }
    return have;
}

// successor of key (keyT, keyOrd) in the stable-sorted interval order of bvh.ts:176
__device__ bool bvh_next(const SceneView &S, const Ray &r, const RayInv &ri, const RayList &L, double keyT, int keyOrd,
                         Interval &out) {
    bool have = false;
    if (L.cnt <= L.cap) {
        for (int e = 0; e < L.cnt; ++e) {
            const int id = L.col[e * 64];
            const RmBvhNode node = S.nodes[id];
            double tE, tX;
            slab_inv(node.lo, node.hi, r, ri, tE, tX);  // hit by construction; same arithmetic, same values
            const double cE = tE > 0.0 ? tE : 0.0;
            const double cX = tX < RM_MAX_DIST ? tX : RM_MAX_DIST;
            const bool after = cE > keyT || (cE == keyT && id > keyOrd);
            if (after && (!have || cE < out.tEnter)) {  // list is in traversal order: first seen wins ties
                out.tEnter = cE;
                out.tExit = cX;
                out.ord = id;
                have = true;
            }
        }
        return have;
    }
    int i = 0;  // overflow: the list is incomplete, walk the tree again
    const int n = S.bvh_nodes;
    while (i < n) {
        const RmBvhNode node = S.nodes[i];
        double tE, tX;
        if (!slab_inv(node.lo, node.hi, r, ri, tE, tX) || tX < 0.0 || tE > RM_MAX_DIST) {
            i = node.skip;
            continue;
        }
        if (node.leaf < 0) {
            i = i + 1;
            continue;
        }
        if ((node.leaf & 0xFF) > 0) {
            const double cE = tE > 0.0 ? tE : 0.0;
            const double cX = tX < RM_MAX_DIST ? tX : RM_MAX_DIST;
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

// ---- the kernel ------------------------------------------------------------------------------

template <typename T>
__device__ __forceinline__ const T *stage(unsigned char *smem, size_t &off, const T *src, int count) {
    off = (off + 15) & ~static_cast<size_t>(15);
    T *dst = reinterpret_cast<T *>(smem + off);
    const int words = static_cast<int>((static_cast<size_t>(count) * sizeof(T) + 3) / 4);
    const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
    uint32_t *d = reinterpret_cast<uint32_t *>(dst);
    for (int i = threadIdx.x; i < words; i += blockDim.x) d[i] = s[i];
    off += static_cast<size_t>(words) * 4;
    return dst;
}

template <int ACCEL, bool LDS>
__global__ __launch_bounds__(256) void render_kernel_v2(const RmRenderParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // XCD-aware tile order: blocks with equal (blockIdx % 8) walk along one tile row
    const int tw = P.tile_w, th = 64 / tw;
    const int rows = P.y_end - P.y_start;
    const int tiles_x = (P.width + tw - 1) / tw;
    const int tiles_y = (rows + 4 * th - 1) / (4 * th);
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int tile_row = (k / tiles_x) * 8 + xcd, tile_col = k % tiles_x;
    if (tile_row >= tiles_y) return;  // whole workgroup

    SceneView S;
    S.nodes = P.bvh;
    S.bvh_prims = P.bvh_prims;
    S.oct = P.oct;
    S.oct_prims = P.oct_prims;
    S.spheres = P.spheres;
    S.radii = P.radii;
    S.n_prims = P.n_prims;
    S.bvh_nodes = P.bvh_nodes;
    size_t off = 0;
    if (LDS) {
        if (ACCEL == 2) {
            S.nodes = stage(smem, off, P.bvh, P.bvh_nodes);
            S.bvh_prims = stage(smem, off, P.bvh_prims, P.bvh_prim_count);
        } else if (ACCEL == 1) {
            S.oct = stage(smem, off, P.oct, P.oct_nodes);
            S.oct_prims = stage(smem, off, P.oct_prims, P.oct_prim_count);
        }
        S.spheres = stage(smem, off, P.spheres, P.n_prims);
        S.radii = stage(smem, off, P.radii, P.n_prims);
        off = (off + 15) & ~static_cast<size_t>(15);
        __syncthreads();
    }
    RayList L;
    L.cap = P.list_cap;
    L.cnt = 0;
    L.col = reinterpret_cast<uint16_t *>(smem + off) + (static_cast<size_t>(wave) * L.cap) * 64 + lane;

    const int x = tile_col * tw + (lane % tw);
    const int row = tile_row * (4 * th) + wave * th + (lane / tw);  // tile-local row
    const bool active = x < P.width && row < rows;
    const int y = P.y_start + row;
    const size_t idx = static_cast<size_t>(row) * P.width + x;
    const bool coop = P.coop != 0, filter = P.filter != 0;

    Ray ray = make_ray(P, active ? x : 0, active ? y : P.y_start);
    RayInv ri;
    uint32_t count = 0, iters = 0;
    int phase = active ? PH_MARCH : PH_DONE;
    int loopi = 0;
    double t = 0.0, depth = RM_MAX_DIST, d0 = 0.0;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    Vec3f hit = {0.f, 0.f, 0.f};
    Interval cur;
    cur.tEnter = 0.0;
    cur.tExit = 0.0;
    cur.ord = -1;
    bool haveCur = false;

    if (ACCEL == 2) {
        ri = make_ray_inv(ray);
        if (active) {
            haveCur = bvh_prologue(S, ray, ri, L, cur);
            if (!haveCur) {  // bvh.ts:190-192 -> sphereTracer.ts:38-40: exactly MAX_DIST, zero normal
                depth = RM_MAX_DIST;
                phase = PH_DONE;
            }
        }
    }

    // march finished with distance `dist_total` (raymarcher.ts:91-102)
    auto finish_march = [&](double dist_total) {
        depth = dist_total;
        hit = point_at(ray, depth);
        phase = (depth >= RM_MAX_DIST) ? PH_DONE : PH_N0;
    };

    for (;;) {
        // ---- A: bookkeeping until this lane needs a distance (sphereTracer.ts:43-64) ------
        bool need = false;
        Vec3f q = {0.f, 0.f, 0.f};
        int onode = -1;
        if (phase == PH_MARCH) {
            for (;;) {
                if (loopi >= RM_MAX_STEPS) {  // loop exhausted: return totalDist
                    finish_march(t);
                    break;
                }
                loopi++;
                const Vec3f p = point_at(ray, t);
                if (ACCEL == 2) {
                    // BVH.onRayMarchStep (bvh.ts:204-240)
                    double skip = 0.0;
                    bool terminate = !haveCur;
                    if (!terminate) {
                        if (t < cur.tEnter) skip = cur.tEnter - t;
                        else if (t > cur.tExit) {
                            const Interval prev = cur;
                            haveCur = bvh_next(S, ray, ri, L, prev.tEnter, prev.ord, cur);
                            if (!haveCur) terminate = true;
                            else if (cur.tEnter > t) skip = cur.tEnter - t;
                        }
                    }
                    if (terminate) {  // -1: return MAX_DIST
                        finish_march(RM_MAX_DIST);
                        break;
                    }
                    if (skip > 0.0) {
                        t += skip;
                        if (t > RM_MAX_DIST) {
                            finish_march(t);
                            break;
                        }
                        continue;
                    }
                } else if (ACCEL == 1) {
                    onode = oct_find(S, p);
                    if (onode >= 0) {
                        const double skip = oct_skip(S.oct[onode], ray, t);
                        if (skip > 0.0) {
                            t += skip;
                            if (t > RM_MAX_DIST) {
                                finish_march(t);
                                break;
                            }
                            continue;
                        }
                    }
                }
                q = p;
                need = true;
                break;
            }
        }
        if (phase >= PH_N0 && phase <= PH_N3) {  // raymarcher.ts:123-132 sample points
            q = hit;
            if (phase == PH_N1) q.x = to_f32(static_cast<double>(hit.x) - 0.01);
            if (phase == PH_N2) q.y = to_f32(static_cast<double>(hit.y) - 0.01);
            if (phase == PH_N3) q.z = to_f32(static_cast<double>(hit.z) - 0.01);
            if (ACCEL == 1) onode = oct_find(S, q);
            need = true;
        }
        if (!__any(need)) break;

        // ---- B: one Scene.getDistance per needing lane --------------------------------------
        double dist;
        if (ACCEL == 2) dist = bvh_distance_wave(S, need, q, count, lane, coop, filter);
        else if (ACCEL == 1) dist = need ? oct_distance_lane(S, onode, q, count, filter) : RM_MAX_DIST;
        else {
            dist = all_prims_wave(S, need, q, lane, coop, filter);
            if (need) count += static_cast<uint32_t>(S.n_prims);
        }

        // ---- C: consume -------------------------------------------------------------------------
        if (need) {
            if (phase == PH_MARCH) {
                t += dist;
                iters += 1;
                if (dist < RM_EPSILON || t > RM_MAX_DIST) finish_march(t);
            } else if (phase == PH_N0) {
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
    if (active) store_pixel(P, idx, depth, nx, ny, nz, count, iters);
}

size_t scene_lds_bytes(const RmRenderParams &p) {
    auto up = [](size_t v) { return (v + 15) & ~static_cast<size_t>(15); };
    size_t b = 0;
    if (p.accel == 2) b += up(static_cast<size_t>(p.bvh_nodes) * sizeof(RmBvhNode)) + up(static_cast<size_t>(p.bvh_prim_count) * 4);
    else if (p.accel == 1) b += up(static_cast<size_t>(p.oct_nodes) * sizeof(RmOctNode)) + up(static_cast<size_t>(p.oct_prim_count) * 4);
    b += up(static_cast<size_t>(p.n_prims) * sizeof(RmSphere)) + up(static_cast<size_t>(p.n_prims) * 8);
    return b + 16;
}

}  // namespace

hipError_t rm_launch_render_v2(const RmRenderParams &p_in, hipStream_t stream) {
    RmRenderParams p = p_in;
    const int rows = p.y_end - p.y_start;
    if (rows <= 0 || p.width <= 0) return hipSuccess;
    const int tw = p.tile_w, th = 64 / tw;
    const int tiles_x = (p.width + tw - 1) / tw;
    const int tiles_y = (rows + 4 * th - 1) / (4 * th);
    const unsigned blocks = 8u * static_cast<unsigned>(tiles_x) * static_cast<unsigned>((tiles_y + 7) / 8);
    if (p.list_cap < 1) p.list_cap = 1;
    const size_t list_bytes = p.accel == 2 ? static_cast<size_t>(4) * p.list_cap * 128 : 0;
    const size_t scene_bytes = scene_lds_bytes(p);
    // stage the scene in LDS when it leaves room for >= 2 workgroups per CU (160 KB LDS)
    const bool lds = p.nodes_in_lds != 0 && scene_bytes + list_bytes + 16 <= 64 * 1024;
    const size_t shmem = (lds ? scene_bytes : 0) + list_bytes + 16;
    const dim3 grid(blocks), block(256);
#define RM_V2(A, L) hipLaunchKernelGGL((render_kernel_v2<A, L>), grid, block, shmem, stream, p)
    if (p.accel == 2) { if (lds) RM_V2(2, true); else RM_V2(2, false); }
    else if (p.accel == 1) { if (lds) RM_V2(1, true); else RM_V2(1, false); }
    else { if (lds) RM_V2(0, true); else RM_V2(0, false); }
#undef RM_V2
    return hipGetLastError();
}
