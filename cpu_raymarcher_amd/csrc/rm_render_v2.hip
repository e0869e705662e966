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
#ifndef __HIPCC_RTC__  // (hiprtc brings its own runtime declarations: rm_rtc.cpp compiles this file too, see RM_RTC_V2 below)
#include <hip/hip_runtime.h>
#endif

#include "rm_device.h"
#ifdef RM_RTC_V2  // the run-time specialiser's build: this launch configuration's parameters as literals (rm_v2_fields.h)
#include "rm_v2_fixed.inc"
#endif

// Diagnostic build only (-DRM_COUNTS): how often each part of the wave loop executes -- per event the number of
// wave-level executions (slot i) and the number of lanes active in them (slot i + 16), accumulated in LDS and added
// to P.stamps[8 .. 39] at the end.  Instructions per launch ~ sum over events of executions x the event's static
// instruction count; lanes / (64 x executions) is the event's lane utilisation.
#ifdef RM_COUNTS
__shared__ unsigned int rm_cnt_s[32];
#define RM_CNT(i)                                                                                  \
    {                                                                                              \
        const unsigned int n_ = static_cast<unsigned int>(__popcll(__ballot(1)));                  \
        const int l_ = static_cast<int>(__lane_id());                                              \
        if (__builtin_amdgcn_readfirstlane(l_) == l_) {                                            \
            atomicAdd(&rm_cnt_s[i], 1u);                                                           \
            atomicAdd(&rm_cnt_s[(i) + 16], n_);                                                    \
        }                                                                                          \
    }
#else
#define RM_CNT(i) {}
#endif

#include "rm_bvh_list.h"  // after RM_CNT: the diagnostic build counts the list scans too
#include "rm_kernels.h"
#include "rm_diag.h"

namespace {

using namespace rmd;

// Diagnostic build only (-DRM_STAMPS): per-section cycle shares of the wave loop, accumulated per
// wave and added to P.stamps[0..7] at the end.  No stamp executes in the product build.
#ifdef RM_STAMPS
#define RM_T0() unsigned long long t_prev_ = __builtin_amdgcn_s_memtime(); unsigned long long t_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
    const unsigned long long t_start_ = __builtin_amdgcn_s_memrealtime();                                                          \
    if (lane == 0 && P.stamps) atomicMin(&P.stamps[6], t_start_);                                                                  \
    if (lane == 0 && P.stamps && blockIdx.x * 4 + (threadIdx.x >> 6) < 8192) P.stamps[40 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t_start_;
#define RM_T(i)                                                  \
    {                                                            \
        const unsigned long long t_now_ = __builtin_amdgcn_s_memtime(); \
        t_acc_[i] += t_now_ - t_prev_;                           \
        t_prev_ = t_now_;                                        \
    }
// ... and a histogram of the waves' finishing times (32 buckets of 64 us after the launch's first wave started, read
// through rm_debug_read_counts): how long is the frame's tail?
#define RM_TEND()                                                                      \
    if (lane == 0 && P.stamps) {                                                        \
        for (int i_ = 0; i_ < 8; ++i_) if (i_ != 6) atomicAdd(&P.stamps[i_], t_acc_[i_]); \
        const unsigned long long first_ = atomicMin(&P.stamps[6], t_start_);            \
        const unsigned long long rel_ = __builtin_amdgcn_s_memrealtime() - (first_ < t_start_ ? first_ : t_start_); \
        unsigned long long b_ = rel_ / 6400ull;  /* 100 MHz: 64 us */                  \
        atomicAdd(&P.stamps[8 + (b_ > 31 ? 31 : b_)], 1ull);                            \
        if (blockIdx.x * 4 + (threadIdx.x >> 6) < 8192) P.stamps[40 + 8192 + blockIdx.x * 4 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime(); \
    }
#else
#define RM_T0()
#define RM_T(i)
#define RM_TEND()
#endif

enum Phase : int { PH_MARCH = 0, PH_N0 = 1, PH_N1 = 2, PH_N2 = 3, PH_N3 = 4, PH_DONE = 5 };

// Fused diagnostics (rm_diag.h): a wave's running totals live in LDS, not in registers -- the wave loop has none to spare.
// Every pixel store adds the two counters AS STORED (Uint16Array wrap included) with LDS atomics; the wave flushes once, at
// its end.  Eight slots per wave (lane & 7), structure of arrays: 64 lanes on ONE address are 64 serial read-modify-writes
// per instruction, four instructions per batch, and the LDS pipeline of the CU stands still meanwhile (measured at six waves
// per SIMD: 932 -> 748 frames/s); eight addresses in eight banks are eight.
struct WaveDiag {
    unsigned long long sdf[8], iters[8];
    unsigned int mx[8], mn_inv[8];
};
__device__ __forceinline__ void wave_diag_add(WaveDiag *w, int lane, uint32_t count, uint32_t iters) {
    const unsigned int c16 = count & 0xFFFFu, i16 = iters & 0xFFFFu;
    const int k = lane & 7;
    atomicAdd(&w->sdf[k], static_cast<unsigned long long>(c16));
    atomicAdd(&w->iters[k], static_cast<unsigned long long>(i16));
    atomicMax(&w->mx[k], c16);
    atomicMax(&w->mn_inv[k], 0xFFFFFFFFu - c16);
}

// Wave-wide minimum of a binary32 value with DPP row operations (no LDS traffic; the
// ds_bpermute form of __shfl_xor cost 18 LDS round trips per fallback ray).  Every lane of the
// wave must be active.  Steps: xor 1, xor 2 inside quads, half-row mirror, row mirror (each
// row of 16 now holds its minimum in every lane), row_bcast15 into rows 1 and 3, row_bcast31
// into rows 2 and 3; lane 63 then holds the minimum of all 64 lanes.
__device__ __forceinline__ float wave_min_f32(float v) {
    int b = __float_as_int(v);
#define RM_DPP_MIN(ctrl, rowmask)                                                              \
    {                                                                                          \
        const float o = __int_as_float(__builtin_amdgcn_update_dpp(b, b, ctrl, rowmask, 0xF, false)); \
        const float c = __int_as_float(b);                                                     \
        b = __float_as_int(o < c ? o : c);                                                     \
    }
    RM_DPP_MIN(0xB1, 0xF)   // quad_perm [1,0,3,2]
    RM_DPP_MIN(0x4E, 0xF)   // quad_perm [2,3,0,1]
    RM_DPP_MIN(0x141, 0xF)  // row_half_mirror
    RM_DPP_MIN(0x140, 0xF)  // row_mirror
    RM_DPP_MIN(0x142, 0xA)  // row_bcast15 -> rows 1, 3
    RM_DPP_MIN(0x143, 0xC)  // row_bcast31 -> rows 2, 3
#undef RM_DPP_MIN
    return __int_as_float(__builtin_amdgcn_readlane(b, 63));
}

// minimum over the lanes in `mask` of a double each of them holds (typically one or two
// candidate lanes): scalar loop over the set bits with v_readlane, no cross-lane network
__device__ __forceinline__ double masked_min_f64(double v, unsigned long long mask, double init) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double best = init;
    while (mask) {
        const int src = __builtin_ctzll(mask);
        mask &= mask - 1;
        const double o = __hiloint2double(__builtin_amdgcn_readlane(hi, src), __builtin_amdgcn_readlane(lo, src));
        best = o < best ? o : best;
    }
    return best;
}

__device__ __forceinline__ float readlane_f32(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// round-up conversion so that the binary32 value is an upper bound of the double
__device__ __forceinline__ float f32_upper(double v) {
    return f32_upper_bound(v);
}

// min over an id list (or 0..n-1 when ids == nullptr) of the exact sphere distance, one
// lane alone: exact evaluation only for spheres whose conservative lower bound does not
// exceed the best upper bound so far.  Skipped spheres satisfy exact > best >= result.
__device__ double lane_min_filtered(const SceneView &S, const int32_t *ids, int n, const Vec3f &p, double closest,
                                    int base = 0) {
    return prims_min_best<int32_t>(S.spheres, S.radii, ids, n, base, p, closest);
}

__device__ double lane_min_exact(const SceneView &S, const int32_t *ids, int n, const Vec3f &p, double closest, int base = 0) {
    for (int k = 0; k < n; ++k) {
        const int id = ids ? ids[k] : base + k;
        closest = min_dist(sphere_sdf_fast(S.spheres[id], S.radii[id], p), closest);
    }
    return closest;
}

// all primitives for ONE point held by every lane (wave-uniform b); all 64 lanes take part
__device__ double coop_all_prims(const SceneView &S, const Vec3f &b, int lane) {
    if (S.n_prims <= 128) {  // two spheres per lane: the lower bounds of pass 1 stay in registers
        const int j0 = lane, j1 = lane + 64;
        const float inf = __builtin_inff();
        float lb0 = inf, lb1 = inf, ub = 10.0f;
        if (j0 < S.n_prims) {
            float err;
            const float a = sphere_sdf_estimate(S.spheres[j0], b, err);
            lb0 = a - err;
            ub = a + err < ub ? a + err : ub;
        }
        if (j1 < S.n_prims) {
            float err;
            const float a = sphere_sdf_estimate(S.spheres[j1], b, err);
            lb1 = a - err;
            ub = a + err < ub ? a + err : ub;
        }
        ub = wave_min_f32(ub);
        const bool c0 = lb0 <= ub, c1 = lb1 <= ub;
        double e0 = RM_MAX_DIST, e1 = RM_MAX_DIST;
        if (c0) e0 = sphere_sdf_fast(S.spheres[j0], S.radii[j0], b);
        double best = masked_min_f64(e0, __ballot(c0), RM_MAX_DIST);
        if (c1) e1 = sphere_sdf_fast(S.spheres[j1], S.radii[j1], b);
        return masked_min_f64(e1, __ballot(c1), best);
    }
    float ub = 10.0f;
    for (int j = lane; j < S.n_prims; j += 64) {
        float err;
        const float a = sphere_sdf_estimate(S.spheres[j], b, err);
        const float hi = a + err;
        ub = hi < ub ? hi : ub;
    }
    ub = wave_min_f32(ub);
    double best = RM_MAX_DIST;
    for (int j0 = 0; j0 < S.n_prims; j0 += 64) {  // wave-uniform trip count
        const int j = j0 + lane;
        bool cand = false;
        double e = RM_MAX_DIST;
        if (j < S.n_prims) {
            const RmSphere s = S.spheres[j];
            float err;
            const float a = sphere_sdf_estimate(s, b, err);
            cand = a - err <= ub;
            if (cand) e = sphere_sdf_fast(s, S.radii[j], b);
        }
        best = masked_min_f64(e, __ballot(cand), best);
    }
    return best;
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
            RM_CNT(12)
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
__device__ double bvh_distance_wave_seq(const RmRenderParams &P, const SceneView &S, bool need, const Vec3f &q,
                                    uint32_t &count, int lane, bool coop, bool filter, bool use_grid,
                                    unsigned long long *dbg_fallback_cycles) {
    double closest = RM_MAX_DIST;
    uint32_t found = 0;
    bool walk_tree = false;
    bool in_root = false;
    if (need && use_grid) {
        // BVH.getPrimitivesAt through the leaf grid: the leaves listed for p's cell are a superset
        // of the leaves whose box contains p; each is re-tested with the reference's inclusive
        // f32 compares.  Outside the root box no leaf can contain p.
        const RmBvhNode root = S.nodes[0];
        if (box_contains(root.lo, root.hi, q)) {
            const int cx = min(max(static_cast<int>((q.x - P.pq_origin[0]) * P.pq_inv[0]), 0), P.pq_dim[0] - 1);
            const int cy = min(max(static_cast<int>((q.y - P.pq_origin[1]) * P.pq_inv[1]), 0), P.pq_dim[1] - 1);
            const int cz = min(max(static_cast<int>((q.z - P.pq_origin[2]) * P.pq_inv[2]), 0), P.pq_dim[2] - 1);
            in_root = true;
            const uint32_t cell = S.pq_cells[(cz * P.pq_dim[1] + cy) * P.pq_dim[0] + cx];
            const int ccnt = static_cast<int>(cell & 0xFFu);
            if (ccnt == 255) walk_tree = true;  // crowded cell: fall back to the tree walk below
            else {
                const uint16_t *lst = S.pq_list + (cell >> 8);
                for (int e = 0; e < ccnt; ++e) {
                    const RmBvhNode node = S.nodes[lst[e]];
                    if (!box_contains(node.lo, node.hi, q)) continue;
                    const int first = node.leaf >> 8, cnt = node.leaf & 0xFF;
                    const int32_t *ids = P.leaf_order ? nullptr : S.bvh_prims + first;
                    closest = (filter && cnt >= 2) ? lane_min_filtered(S, ids, cnt, q, closest, first)
                                                   : lane_min_exact(S, ids, cnt, q, closest, first);
                    found += static_cast<uint32_t>(cnt);
                }
            }
        }
    }
    if (need && (!use_grid || walk_tree)) {
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
            const int32_t *ids = P.leaf_order ? nullptr : S.bvh_prims + first;
            closest = (filter && cnt >= 2) ? lane_min_filtered(S, ids, cnt, q, closest, first)
                                           : lane_min_exact(S, ids, cnt, q, closest, first);
            found += static_cast<uint32_t>(cnt);
            i = node.skip;
        }
    }
    bool fallback = need && found == 0;
    bool served = false;
    if (fallback && P.use_nn && in_root) {
        // scene.ts:173 fallback through the cell's nearest-candidate list: every sphere that can
        // attain the minimum for a point of this cell is in the list (rm_scene_host.cpp), so the
        // minimum over the list equals the minimum over all N primitives; N are still counted.
        const int nx = min(max(static_cast<int>((q.x - P.pq_origin[0]) * P.nn_inv[0]), 0), P.nn_dim[0] - 1);
        const int ny = min(max(static_cast<int>((q.y - P.pq_origin[1]) * P.nn_inv[1]), 0), P.nn_dim[1] - 1);
        const int nz = min(max(static_cast<int>((q.z - P.pq_origin[2]) * P.nn_inv[2]), 0), P.nn_dim[2] - 1);
        const uint32_t cell = S.nn_cells[(nz * P.nn_dim[1] + ny) * P.nn_dim[0] + nx];  // global memory: ~1 MB of tables
        const int ccnt = static_cast<int>(cell & 0xFFu);
        if (ccnt != 255) {
            closest = prims_min_best<uint16_t>(S.spheres, S.radii, S.nn_list + (cell >> 8), ccnt, 0, q, RM_MAX_DIST);
            served = true;
        }
    }
#ifdef RM_STAMPS
    const unsigned long long t_fb0 = __builtin_amdgcn_s_memtime();
#endif
    const double all = all_prims_wave(S, fallback && !served, q, lane, coop, filter);
#ifdef RM_STAMPS
    if (dbg_fallback_cycles) *dbg_fallback_cycles += __builtin_amdgcn_s_memtime() - t_fb0;
#endif
    if (fallback) {
        if (!served) closest = all;
        count += static_cast<uint32_t>(S.n_prims);
    } else if (need) {
        count += found;
    }
    return closest;
}

// Scan state of "evaluate once": the sphere with the smallest upper bound and the smallest lower bound of the rest
struct BestScan {
    int k1;
    float hi1, lb1, lb2;
    uint32_t kbest, ksecond;  // UR: order keys (see scan_sphere)
};
// UR (every sphere of the scene has the same radius): distance order = order of the squared centre distances, so
// the scan ranks by s2 = |p - c|^2 (hi1 holds the smallest, lb2 the smallest of the others) and the square root,
// the radius and the error margin are applied once, after the scan (scan_finish), instead of once per sphere.
// UR keys: s2 >= 0, so its bit pattern orders like the value; the low eight mantissa bits give way to the sphere id (UR
// kernels run scenes of at most 256 spheres), which makes "smallest and its id, and the smallest of the others" three
// integer min / max instructions with no compare and no select:  lo = min(best, key), hi = max(best, key), second =
// min(second, hi).  Clearing mantissa bits only lowers a key, so the runner-up's bound stays a lower bound; two spheres
// closer than 2^-15 (relative, in s2) may be ranked the wrong way round, which the near-tie test after the exact
// evaluation catches like any other near tie (the wrongly ranked one's lower bound cannot exceed the exact value).
template <bool UR>
__device__ __forceinline__ void scan_sphere(BestScan &b, const RmSphere &s, int id, const Vec3f &p) {
    if (UR) {
        const float dx = p.x - s.cx, dy = p.y - s.cy, dz = p.z - s.cz;
        const float s2 = dx * dx + dy * dy + dz * dz;
        const uint32_t key = (__float_as_uint(s2) & 0xFFFFFF00u) | static_cast<uint32_t>(id);
        const uint32_t lo = min(key, b.kbest), hi = max(key, b.kbest);
        b.kbest = lo;
        b.ksecond = min(hi, b.ksecond);
        return;
    }
    float err;
    const float a = sphere_sdf_estimate(s, p, err);
    const float lb = a - err, hi = a + err;
    const bool better = hi < b.hi1;
    b.lb2 = __builtin_fminf(b.lb2, better ? b.lb1 : lb);
    b.k1 = better ? id : b.k1;
    b.lb1 = better ? lb : b.lb1;
    b.hi1 = better ? hi : b.hi1;
}
// UR: lb2 (the runner-up's squared centre distance) -> the conservative lower bound sphere_sdf_estimate would have
// given that sphere.  Every other sphere j has s2_j >= lb2, and the bound is monotone in s2 up to the two ulps of
// v_sqrt_f32 and of the subtraction, which the margin (12x the worst-case error of the estimate) absorbs.
__device__ __forceinline__ void scan_finish_uniform(BestScan &b, float rf) {
    const float len = __builtin_amdgcn_sqrtf(b.lb2);
    const float lb = (len - rf) - (len + __builtin_fabsf(rf) + 1.0f) * 4e-6f;
    b.lb2 = b.lb2 < __builtin_inff() ? lb : __builtin_inff();
}

// BVH branch of Scene.getDistance (scene.ts:167-181); whole wave must call.  Every source of candidates of a lane
// -- the leaves whose box contains its point (through the leaf grid or the tree walk) or, when there is none, the
// cell's nearest-candidate list (scene.ts:173) -- feeds ONE scan; the exact FP64 evaluation then happens once per
// call, for all lanes together, instead of once per list position at which some lane's bound passes.  Near ties
// (the runner-up's lower bound does not exceed the exact value) are recomputed by the sequential form above.
// What the fused normal evaluation (render_kernel_v2, section N) needs to know about the evaluation at the hit point:
// which sphere gave the value, how far every other candidate is at least (binary32 lower bound), how many primitives
// were counted, and whether the set of leaves that contain the point is the same for every point within NRM_DELTA of it.
struct NormalAux {
    int k1;
    float lb2;
    uint32_t found;
    float rho;       // every point within rho (per axis) of q lies in exactly those of the cell's LISTED leaves that q lies in (0: unknown)
    float rho_cell;  // every point within rho_cell (per axis) of q is covered by the leaf list of q's cell: the host grows every
                     // cell's list by 0.0101 (RM_NRM_DELTA and a margin), plus q's own distance to the faces of its cell
    bool ok;
};
// The three offset points of getNormal (raymarcher.ts:126-132) are hit - 0.01 e_i, stored as binary32:
// |q_i - hit| <= 0.01 + half an ulp of a coordinate (< 2^-21 for |x| < 16); the box faces compared against are
// binary32.  0.010003 covers all of it with a factor of 100 to spare.
#define RM_NRM_DELTA 0.010003f
// BoundingBox.contains (boundingBox.ts:15-21) and how far the answer is from changing, in one value: m = the smallest of
// the six binary32 differences p - lo, hi - p.  A difference of two binary32 numbers has the sign of the real difference
// and is zero only for equal operands (denormals are kept: float_denorm_mode_32 = 3), so  contains(p) <=> m >= 0  exactly
// as the six inclusive compares decide it; and every point within |m| of p per axis gets the same answer (inside by m on
// every axis, or outside by |m| on the axis that gives the minimum).  Nine instructions instead of six compares and five
// mask operations -- and the stability radius of the leaf set, which the fused normal and the in-round march steps of
// render_kernel_v2 need, comes with it.
__device__ __forceinline__ float box_margin(const float lo[3], const float hi[3], const Vec3f &p) {
    const float a = __builtin_fminf(__builtin_fminf(p.x - lo[0], p.y - lo[1]), p.z - lo[2]);
    const float b = __builtin_fminf(__builtin_fminf(hi[0] - p.x, hi[1] - p.y), hi[2] - p.z);
    return __builtin_fminf(a, b);
}

template <bool UR>
__device__ double bvh_distance_wave(const RmRenderParams &P, const SceneView &S, bool need, const Vec3f &q,
                                    uint32_t &count, int lane, bool coop, bool filter, bool use_grid,
                                    unsigned long long *dbg_fallback_cycles, NormalAux *aux) {
    aux->ok = false;
    aux->rho = 0.f;
    aux->rho_cell = 0.f;
    if (!filter) return bvh_distance_wave_seq(P, S, need, q, count, lane, coop, filter, use_grid, dbg_fallback_cycles);
    double closest = RM_MAX_DIST;
    uint32_t found = 0;
    bool walk_tree = false;
    bool in_root = false;
    float rho = __builtin_inff();  // the leaf set is the same within rho of q (per axis): min |box_margin| over the root and the cell's leaves
    float rho_cell = 0.f;
    BestScan bs;
    bs.k1 = -1;
    bs.hi1 = bs.lb1 = bs.lb2 = __builtin_inff();
    bs.kbest = bs.ksecond = 0xFFFFFFFFu;
    auto scan_leaf = [&](const RmBvhNode &node) {
        const int first = node.leaf >> 8, cnt = node.leaf & 0xFF;
        for (int k = 0; k < cnt; ++k) {
            const int id = P.leaf_order ? first + k : S.bvh_prims[first + k];
            RM_CNT(8)
            scan_sphere<UR>(bs, S.spheres[id], id, q);
        }
        found += static_cast<uint32_t>(cnt);
    };
    if (need && use_grid) {
        const RmBvhNode root = S.nodes[0];
        const float m_root = box_margin(root.lo, root.hi, q);
        if (m_root >= 0.f) {
            const float gx = (q.x - P.pq_origin[0]) * P.pq_inv[0], gy = (q.y - P.pq_origin[1]) * P.pq_inv[1], gz = (q.z - P.pq_origin[2]) * P.pq_inv[2];
            const int cx = min(max(static_cast<int>(gx), 0), P.pq_dim[0] - 1);
            const int cy = min(max(static_cast<int>(gy), 0), P.pq_dim[1] - 1);
            const int cz = min(max(static_cast<int>(gz), 0), P.pq_dim[2] - 1);
            in_root = true;
            {   // how far q is from the faces of its cell, in world units (the cell coordinates carry ~1e-5 of a cell of
                // rounding: the 1e-4 the host's 0.0101 has over the 0.0100 used here covers it forty times)
                const float fx = gx - static_cast<float>(cx), fy = gy - static_cast<float>(cy), fz = gz - static_cast<float>(cz);
                const float in_cell = __builtin_fminf(__builtin_fminf(__builtin_fminf(fx, 1.0f - fx) * P.pq_cell[0], __builtin_fminf(fy, 1.0f - fy) * P.pq_cell[1]),
                                                      __builtin_fminf(fz, 1.0f - fz) * P.pq_cell[2]);
                rho_cell = 0.0100f + __builtin_fmaxf(in_cell, 0.f);
            }
            // (the margins of RM_NRM_DELTA assume half an ulp of a coordinate below 2^-21: |coordinate| < 16)
            rho = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(q.x), __builtin_fabsf(q.y)), __builtin_fabsf(q.z)) < 16.0f ? m_root : 0.f;
            const uint32_t cell = S.pq_cells[(cz * P.pq_dim[1] + cy) * P.pq_dim[0] + cx];
            const int ccnt = static_cast<int>(cell & 0xFFu);
            if (ccnt == 255) walk_tree = true;  // crowded cell: the tree walk below
            else {
                const uint16_t *lst = S.pq_list + (cell >> 8);
                for (int e = 0; e < ccnt; ++e) {
                    RM_CNT(7)
                    const RmBvhNode node = S.nodes[lst[e]];
                    const float m = box_margin(node.lo, node.hi, q);
                    rho = __builtin_fminf(rho, __builtin_fabsf(m));
                    if (m >= 0.f) scan_leaf(node);
                }
            }
        }
    }
    if (need && (!use_grid || walk_tree)) {
        rho = 0.f;  // the fused normal evaluation and the in-round steps rely on the cell's leaf list (grown by RM_NRM_DELTA on the host)
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
            scan_leaf(node);
            i = node.skip;
        }
    }
    const bool fallback = need && found == 0;
    bool served = false;
    if (fallback && P.use_nn && in_root) {  // scene.ts:173 through the cell's nearest-candidate list
        const int nx = min(max(static_cast<int>((q.x - P.pq_origin[0]) * P.nn_inv[0]), 0), P.nn_dim[0] - 1);
        const int ny = min(max(static_cast<int>((q.y - P.pq_origin[1]) * P.nn_inv[1]), 0), P.nn_dim[1] - 1);
        const int nz = min(max(static_cast<int>((q.z - P.pq_origin[2]) * P.nn_inv[2]), 0), P.nn_dim[2] - 1);
        const uint32_t cell = S.nn_cells[(nz * P.nn_dim[1] + ny) * P.nn_dim[0] + nx];
        const int ccnt = static_cast<int>(cell & 0xFFu);
        if (ccnt != 255) {
            const uint16_t *lst = S.nn_list + (cell >> 8);
            for (int e = 0; e < ccnt; ++e) {
                const int id = lst[e];
                RM_CNT(9)
                scan_sphere<UR>(bs, S.spheres[id], id, q);
            }
            served = true;
        }
    }
    // Points no list serves (outside the root box: a ray that overshoots the grid after passing a sphere still takes one
    // step per remaining interval, each an evaluation of ALL primitives, scene.ts:173).  When many lanes of the wave hold
    // such a point -- whole batches do -- every lane feeds all N spheres into the SAME scan as the leaf candidates (twelve
    // instructions per sphere for one-radius scenes) and joins the one exact evaluation below; a few such lanes are
    // served one at a time by the whole wave afterwards (all_prims_wave).  Round 3: these rounds were 8 % of C3's VALU
    // instructions with the separate estimate-and-evaluate loop.
    if (fallback && !served) RM_CNT(13)
    {
        const unsigned long long unserved = __ballot(fallback && !served);
        if (unserved) {
            const int n = S.n_prims, m = __popcll(unserved);
            const bool one_by_one = coop && n > 8 && static_cast<long long>(m) * (28 * ((n + 63) / 64) + 250) < 14ll * n + 200;
            if (!one_by_one && (!UR || n <= 256)) {
                if (fallback && !served) {
                    for (int id = 0; id < n; ++id) scan_sphere<UR>(bs, S.spheres[id], id, q);
                    served = true;
                }
            }
        }
    }
    // the one exact evaluation of this call
    bool redo = false;
    if (UR && bs.kbest != 0xFFFFFFFFu) {  // decode the keys: best id, and the runner-up's squared distance (rounded down)
        bs.k1 = static_cast<int>(bs.kbest & 0xFFu);
        bs.lb2 = bs.ksecond == 0xFFFFFFFFu ? __builtin_inff() : __uint_as_float(bs.ksecond & 0xFFFFFF00u);
    }
    if (bs.k1 >= 0) {
        RM_CNT(10)
        const RmSphere s1 = S.spheres[bs.k1];
        if (UR) scan_finish_uniform(bs, s1.rf);
        closest = sphere_sdf_fast(s1, S.radii[bs.k1], q);
        if (closest > RM_MAX_DIST) closest = RM_MAX_DIST;  // Math.min(sdf, closestDistance = 10)
        redo = bs.lb2 <= f32_upper_bound(closest);
    }
    {   // the value came from sphere k1 alone, every other candidate of the leaf set (the same within rho) is at least lb2 away
        aux->k1 = bs.k1;
        aux->lb2 = bs.lb2;
        aux->found = found;
        aux->rho = rho;
        aux->rho_cell = rho_cell;
        aux->ok = need && in_root && found > 0 && bs.k1 >= 0 && !redo;
    }
    if (__any(redo)) {  // near tie somewhere in the wave: those lanes take the sequential form (same result by construction)
        if (redo) RM_CNT(11)
        uint32_t dummy = 0;
        const double r = bvh_distance_wave_seq(P, S, redo, q, dummy, lane, coop, filter, use_grid, nullptr);
        if (redo) closest = r;
    }
#ifdef RM_STAMPS
    const unsigned long long t_fb0 = __builtin_amdgcn_s_memtime();
#endif
    const double all = all_prims_wave(S, fallback && !served, q, lane, coop, filter);
#ifdef RM_STAMPS
    if (dbg_fallback_cycles) *dbg_fallback_cycles += __builtin_amdgcn_s_memtime() - t_fb0;
#endif
    if (fallback) {
        if (!served) closest = all;
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
        const float cx = nodes[i].center[0], cy = nodes[i].center[1], cz = nodes[i].center[2];
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

// The same result for a whole 64-pixel batch without the tree walk.  A leaf's box lies inside every ancestor's (checked
// by the host), and every operation of the slab test is monotone, so a ray hits a leaf exactly when the leaf's own
// test passes: the set of recorded leaves is {non-empty leaves whose test passes}, in increasing node index.  The
// batch's rays form a bundle through the pixel rectangle [u0, u1] x [v0, v1]: a point s (u c0 + v c1 - c2), s >= 0, of
// such a ray satisfies q.(c0 + u0 c2) >= 0, q.(c0 + u1 c2) <= 0 and the same with c1 and v, so a box that lies
// entirely on the wrong side of one of those four planes through the origin is hit by no ray of the bundle.  Each lane
// culls one leaf (binary32; the rectangle widened by half a pixel spacing -- a hundred times the rounding error of
// the ray directions -- and a tolerance for the dot products' own roundings, bundle_misses_box), a ballot collects
// the survivors, and all lanes run the exact test on those only.  Rays outside the frame (`active` false) take part in the cull but record nothing.
struct Bundle {
    float n[4][3];  // plane normals: left, right, bottom, top (keep side: >= 0, <= 0, >= 0, <= 0)
};
__device__ __forceinline__ Bundle make_bundle(const RmRenderParams &C, int x0, int x1, int y0, int y1) {
    const float mu = __builtin_fmaxf(1.0f / static_cast<float>(C.width), 1e-4f);
    const float mv = __builtin_fmaxf(1.0f / static_cast<float>(C.height), 1e-4f);
    const float u0 = (static_cast<float>(x0) / static_cast<float>(C.width) - 0.5f) * 2.0f - mu;
    const float u1 = (static_cast<float>(x1) / static_cast<float>(C.width) - 0.5f) * 2.0f + mu;
    const float v0 = (static_cast<float>(y0) / static_cast<float>(C.height) - 0.5f) * 2.0f - mv;
    const float v1 = (static_cast<float>(y1) / static_cast<float>(C.height) - 0.5f) * 2.0f + mv;
    Bundle b;
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // make_ray: d = u c0 + v c1 - c2 with c0 = rot[0..2], c1 = rot[3..5], c2 = rot[6..8]
        b.n[0][k] = C.rot[k] + u0 * C.rot[6 + k];
        b.n[1][k] = C.rot[k] + u1 * C.rot[6 + k];
        b.n[2][k] = C.rot[3 + k] + v0 * C.rot[6 + k];
        b.n[3][k] = C.rot[3 + k] + v1 * C.rot[6 + k];
    }
    return b;
}
__device__ __forceinline__ bool bundle_misses_box(const Bundle &b, const float lo[3], const float hi[3], const float o[3]) {
    const float ql[3] = {lo[0] - o[0], lo[1] - o[1], lo[2] - o[2]};
    const float qh[3] = {hi[0] - o[0], hi[1] - o[1], hi[2] - o[2]};
    bool miss = false;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float mx = 0.f, mn = 0.f, mag = 0.f;  // max / min over the box's corners of q . n; sum of the term magnitudes
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float a = ql[k] * b.n[p][k], c = qh[k] * b.n[p][k];
            mx += __builtin_fmaxf(a, c);
            mn += __builtin_fminf(a, c);
            mag += __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(c));
        }
        // the binary32 sums are within 1e-6 * mag of the real ones (roundings of q, n, the products and the two
        // additions: < 6e-7 * mag): a box is only dropped when it is outside by more than that, whatever its size
        // (the half-pixel widening alone shrinks with the distance to the camera; a large box that passes close to
        // the camera, almost in a side plane, needs the absolute term)
        const float tol = 1e-6f * mag;
        miss = miss || ((p & 1) ? mn > tol : mx < -tol);
    }
    return miss;
}

template <bool REL>
__device__ bool bvh_prologue_cull(const SceneView &S, const RmRenderParams &C, const Bundle &B, bool active, const Ray &r,
                                  const RayInv &ri, RayList &L, Interval &first, int lane) {
    bool have = false;
    L.cnt = 0;
    {   // the root first: a batch of background rays ends here, as it does in the tree walk
        const RmBvhNode root = S.nodes[0];
        double tE = 0.0, tX = 0.0;
        const bool hit = active && node_slab<REL>(S, root, 0, r, ri, tE, tX) && !(tX < 0.0) && !(tE > RM_MAX_DIST);
        if (!__any(hit)) {
            L.live = 0;
            return false;
        }
        active = hit;  // a ray that misses the root hits no leaf
    }
    const float o[3] = {C.origin[0], C.origin[1], C.origin[2]};
    for (int j0 = 0; j0 < C.bvh_leaf_count; j0 += 64) {  // wave-uniform trip count
        const int j = j0 + lane;
        int li = 0;
        bool cand = false;
        if (j < C.bvh_leaf_count) {
            li = C.bvh_leaves[j];
            const RmBvhNode nd = S.nodes[li];
            cand = !bundle_misses_box(B, nd.lo, nd.hi, o);
        }
        unsigned long long m = __ballot(cand);
        while (m) {  // survivors in increasing node index = traversal order
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const int i = __builtin_amdgcn_readlane(li, src);
            RM_CNT(5)
            if (active) {
                const RmBvhNode node = S.nodes[i];
                double tE, tX;
                if (node_slab<REL>(S, node, i, r, ri, tE, tX) && !(tX < 0.0) && !(tE > RM_MAX_DIST)) {  // bvh.ts:145,151,165
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
            }
        }
    }
    L.live = L.cnt;
    return have;
}

// ---- the kernel ------------------------------------------------------------------------------

// Where every staged table lives in LDS: a pure function of the launch parameters (counts), evaluated once by the launcher
// and handed over in RmRenderParams::lds_off, so that any section of the wave loop can rebuild its SceneView from freshly
// loaded parameters (cold_params) with one scalar load, instead of keeping a dozen table addresses and scene constants in SGPRs across the whole loop -- where
// they do not fit: the first build spilled 130 SGPRs to VGPR lanes and a tenth of the wave loop's straight-line
// instructions were v_readlane reloads (4.7 issue cycles each, profiles/r02/valu_issue_costs.json).
struct LdsLayout {
    uint32_t nodes, prims, cells, list, oct, oct_prims, spheres, radii, rel, end;
};
// the one definition of the layout, used by the launcher (which writes it into RmRenderParams::lds_off)
inline LdsLayout lds_layout_host(const RmRenderParams &C, int accel, bool lds, bool rel) {
    auto up = [](uint32_t v) { return (v + 15u) & ~15u; };
    auto words = [](uint32_t count, uint32_t size) { return ((count * size + 3u) / 4u) * 4u; };
    LdsLayout o = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t off = 0;
    if (lds) {
        if (accel == 2) {
            o.nodes = off = up(off);
            off += words(static_cast<uint32_t>(C.bvh_nodes), sizeof(RmBvhNode));
            o.prims = off = up(off);
            off += words(static_cast<uint32_t>(C.bvh_prim_count), 4);
            o.cells = off = up(off);
            off += words(static_cast<uint32_t>(C.pq_cell_count), 4);
            o.list = off = up(off);
            off += words(static_cast<uint32_t>(C.pq_list_count), 2);
        } else if (accel == 1) {
            o.oct = off = up(off);
            off += words(static_cast<uint32_t>(C.oct_nodes), sizeof(RmOctNode));
            o.oct_prims = off = up(off);
            off += words(static_cast<uint32_t>(C.oct_prim_count), 4);
        }
        o.spheres = off = up(off);
        off += words(static_cast<uint32_t>(C.n_prims), sizeof(RmSphere));
        o.radii = off = up(off);
        off += words(static_cast<uint32_t>(C.n_prims), 8);
        off = up(off);
        if (rel) {
            o.rel = off;
            off += static_cast<uint32_t>(C.bvh_nodes) * 48u;
        }
    }
    o.end = off;
    return o;
}
template <int ACCEL, bool LDS, bool REL>
__device__ __forceinline__ LdsLayout lds_layout(const RmRenderParams &C) {
    LdsLayout o;
    o.nodes = C.lds_off[0];
    o.prims = C.lds_off[1];
    o.cells = C.lds_off[2];
    o.list = C.lds_off[3];
    o.oct = C.lds_off[4];
    o.oct_prims = C.lds_off[5];
    o.spheres = C.lds_off[6];
    o.radii = C.lds_off[7];
    o.rel = C.lds_off[8];
    o.end = C.lds_off[9];
    return o;
}
template <int ACCEL, bool LDS, bool REL>
__device__ __forceinline__ SceneView scene_view(const RmRenderParams &C, unsigned char *smem) {
    SceneView S;
    S.nodes = C.bvh;
    S.bvh_prims = C.bvh_prims;
    S.oct = C.oct;
    S.oct_prims = C.oct_prims;
    S.spheres = C.spheres;
    S.radii = C.radii;
    S.pq_cells = C.pq_cells;
    S.pq_list = C.pq_list;
    S.nn_cells = C.nn_cells;
    S.nn_list = C.nn_list;
    S.rel = nullptr;
    S.n_prims = C.n_prims;
    S.bvh_nodes = C.bvh_nodes;
    if (LDS) {
        const LdsLayout o = lds_layout<ACCEL, LDS, REL>(C);
        if (ACCEL == 2) {
            S.nodes = reinterpret_cast<const RmBvhNode *>(smem + o.nodes);
            S.bvh_prims = reinterpret_cast<const int32_t *>(smem + o.prims);
            S.pq_cells = reinterpret_cast<const uint32_t *>(smem + o.cells);
            S.pq_list = reinterpret_cast<const uint16_t *>(smem + o.list);
        } else if (ACCEL == 1) {
            S.oct = reinterpret_cast<const RmOctNode *>(smem + o.oct);
            S.oct_prims = reinterpret_cast<const int32_t *>(smem + o.oct_prims);
        }
        S.spheres = reinterpret_cast<const RmSphere *>(smem + o.spheres);
        S.radii = reinterpret_cast<const double *>(smem + o.radii);
        if (REL) S.rel = reinterpret_cast<const double *>(smem + o.rel);
    }
    return S;
}

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

// Wave-granular work queues: a work item is one tile of item_px (64, 128 or 256) pixels, tile_w wide and item_px / tile_w
// tall.  XCD x owns the tile rows r with r % 8 == x, so the partial-line stores of horizontally adjacent tiles meet in
// one L2.  Round 3: every XCD's rows are dealt over RM_SUBQ sub-queues (row x + 8 m belongs to sub-queue m % RM_SUBQ), 64
// queue heads instead of 8.  A queue head is ONE address that every claim of that queue must pass through, and a
// device-scope read-modify-write on one address completes every ~146 ns (measured: a 4K frame of 64-pixel items took
// 16 200 claims x 146 ns = 2.37 ms however many waves were resident): with eight heads a frame alone could not use items
// smaller than 128 pixels, and its slowest items (a 256-pixel item whose four batches each hold a 100-step grazing ray:
// 400 wave-loop trips in series) set its duration.  A wave serves one home queue (its XCD, sub-queue by workgroup) and,
// when that is empty, looks at all 64 heads at once (one per lane) and takes from the nearest queue that has entries left.
// The heads sit RM_QSTRIDE words (256 bytes) apart: device-scope atomics on one 128-byte line are performed one after the
// other whichever word they name (microbenchmark, 6144 waves x 32 claims: 64 heads 4 bytes apart 4.7 ns per claim over
// the whole chip, 256 bytes apart 0.25 ns), and with thousands of waves in the queue for that line a claim took 13 us.
#define RM_SUBQ 8
#define RM_QUEUES (8 * RM_SUBQ)
#define RM_QSTRIDE 64
inline int queue_rows_host(int tiles_y, int qid) {
    const int x = qid / RM_SUBQ, s = qid % RM_SUBQ, Rx = (tiles_y - x + 7) >> 3;
    return Rx > s ? (Rx - s + RM_SUBQ - 1) / RM_SUBQ : 0;
}
struct TileQueue {
    unsigned int *counters;  // RM_QUEUES heads, zero before the launch (the previous user's last wave leaves them so)
    int tiles_x, tiles_y, tile_w, tile_h, item_px;
    unsigned int tiles_x_magic;  // floor(2^32 / tiles_x) + 1; 0 for tiles_x == 1
    const uint16_t *perm;        // longest-first order of every queue (null: the queue's own order), perm_stride entries apart
    int perm_stride;
};

// tile rows of queue qid = x * RM_SUBQ + s: the rows x + 8 m with m % RM_SUBQ == s
__device__ __forceinline__ int queue_rows(int tiles_y, int qid) {
    const int x = qid / RM_SUBQ, s = qid % RM_SUBQ;
    const int Rx = (tiles_y - x + 7) >> 3;  // rows of XCD x (may be <= 0)
    return Rx > s ? (Rx - s + RM_SUBQ - 1) / RM_SUBQ : 0;
}

// entry k of queue qid -> tile (row, col); false beyond the queue's end.  A queue's rows are handed out from the middle
// of the frame outwards, so the light rows near the top and bottom edges (rays that miss everything) come last.
__device__ __forceinline__ bool queue_entry(const TileQueue &Q, int qid, unsigned int k, int &tile_col, int &tile_row) {
    const int R = queue_rows(Q.tiles_y, qid);
    if (k >= static_cast<unsigned int>(R) * static_cast<unsigned int>(Q.tiles_x)) return false;
    if (Q.perm) k = Q.perm[static_cast<size_t>(qid) * Q.perm_stride + k];  // the k-th item to hand out is the one the previous frame found k-th longest
    const unsigned int q = Q.tiles_x_magic ? __umulhi(k, Q.tiles_x_magic) : k;  // k / tiles_x (exact: k * tiles_x < 2^32)
    const int qi = static_cast<int>(q), mid = R >> 1;
    const int j = (qi & 1) ? mid - ((qi + 1) >> 1) : mid + (qi >> 1);
    tile_row = (qid / RM_SUBQ) + 8 * ((qid % RM_SUBQ) + RM_SUBQ * j);
    tile_col = static_cast<int>(k - q * static_cast<unsigned int>(Q.tiles_x));
    return true;
}
// ... and back: the slot (row slot * tiles_x + col) of a tile in its queue (for the cost feedback of the longest-first order)
__device__ __forceinline__ unsigned int queue_slot_of(const TileQueue &Q, int qid, int tile_col, int tile_row) {
    const int R = queue_rows(Q.tiles_y, qid), mid = R >> 1;
    const int j = (tile_row >> 3) / RM_SUBQ;
    const int qi = j >= mid ? 2 * (j - mid) : 2 * (mid - j) - 1;
    return static_cast<unsigned int>(qi) * static_cast<unsigned int>(Q.tiles_x) + static_cast<unsigned int>(tile_col);
}

// next tile for this wave; false when every queue is exhausted.  `first_claim`: lane 0's result of an atomicAdd on the
// home queue that the caller issued earlier (so that its round trip overlaps the pixel stores).  When the home queue is
// empty, lane l reads head l (device scope) and the wave takes from the first queue after `home` -- the rest of its own
// XCD's sub-queues come first -- that still has entries; a claim that loses the race for a queue's last entry looks again.
// Heads only grow, so every look either succeeds or finds one more queue exhausted: the loop ends.
__device__ __forceinline__ bool pull_tile(const TileQueue &Q, int &home, int &tile_col, int &tile_row, int lane,
                                          unsigned int first_claim) {
    unsigned int k = static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(first_claim)));
    if (queue_entry(Q, home, k, tile_col, tile_row)) return true;
    const unsigned int mine = static_cast<unsigned int>(queue_rows(Q.tiles_y, lane)) * static_cast<unsigned int>(Q.tiles_x);  // RM_QUEUES == 64 lanes
    for (int attempt = 0; attempt < 4096; ++attempt) {  // (the bound is never reached: see above)
        const unsigned int head = __hip_atomic_load(&Q.counters[lane * RM_QSTRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long avail = __ballot(head < mine);
        if (avail == 0) return false;
        const int h = (home + 1) & (RM_QUEUES - 1);
        const unsigned long long rot = h ? ((avail >> h) | (avail << (RM_QUEUES - h))) : avail;
        const int x = (h + __builtin_ctzll(rot)) & (RM_QUEUES - 1);
        k = 0;
        if (lane == 0) k = atomicAdd(&Q.counters[x * RM_QSTRIDE], 1u);
        k = static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(k)));
        if (queue_entry(Q, x, k, tile_col, tile_row)) {
            home = x;
            return true;
        }
    }
    return false;
}

// The kernel parameters live in the kernarg segment.  Sections that run once per 64 pixels (refill: ray
// setup, pixel stores, tile queue) read what they need through this freshly laundered pointer, so the
// ~60 SGPRs of camera matrix, output pointers and frame geometry are loaded there (s_load) and dead
// again afterwards, instead of being hoisted out of the wave loop and spilled to VGPR lanes around it
// (the first build carried 198 spilled SGPRs and reloaded ~100 of them per loop iteration).
__device__ __forceinline__ RmRenderParams cold_params() {
#if __HIP_DEVICE_COMPILE__  // the host pass only parses this body; address spaces exist in the device pass
    typedef const __attribute__((address_space(4))) RmRenderParams *KernArgs;
    KernArgs p = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();  // P is the kernel's only argument
    asm volatile("" : "+s"(p));
#ifdef RM_RTC_V2  // what is configuration is a literal (rm_v2_fix, generated); what is left -- pointers, camera, rows -- is loaded
    RmRenderParams r = *p;
    rm_v2_fix(r);
    return r;
#else
    return *p;
#endif
#else
    return RmRenderParams();
#endif
}

template <int ACCEL, bool LDS, bool UR = false, bool REL = false>
// Round 3: 80 VGPRs without a spill and without scratch in every instantiation, vec3.length = sqrt build included: six
// waves per SIMD (tests/test_build_invariants.py holds all of them to that).  Round 2 needed 96 (five waves; the REL and sqrt
// builds spilled at five and ran at four); what changed is the lane state (see "lane state" below), the wave index as an
// SGPR, and -amdgpu-inline-max-bb (Makefile).
#define RM_V2_WAVES 6
__device__ __forceinline__ void render_v2_body(const RmRenderParams &P) {
    // (RM_RTC_V2 builds: the literals are applied where the wave loop re-loads the block, cold_params; what runs once per
    // workgroup here reads the kernel argument as it is -- a patched local copy of the block indexed at run time, as the REL
    // staging loop does, would live in scratch)
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ WaveDiag wave_diag[4];
    const int lane0 = threadIdx.x & 63;  // (used before the wave loop only: inside it the lane id is re-derived where needed, see lane_now)
    const int lane = lane0;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));  // wave-uniform by construction: an SGPR, not a VGPR held for the whole kernel
    if (lane < 8) {  // (LDS operations of one wave execute in order: no barrier between this and the wave's own atomics)
        wave_diag[wave].sdf[lane] = 0;
        wave_diag[wave].iters[lane] = 0;
        wave_diag[wave].mx[lane] = 0;
        wave_diag[wave].mn_inv[lane] = 0;
    }
#ifdef RM_STAMPS  // diagnostic builds: per-wave times (rm_debug_read_wave_times): kernel entry here, wave-loop start and end in RM_T0 / RM_TEND
    if (lane == 0 && P.stamps && blockIdx.x * 4 + wave < 8192) P.stamps[40 + 2 * 8192 + blockIdx.x * 4 + wave] = __builtin_amdgcn_s_memrealtime();
#endif

    {   // staged once per persistent workgroup, at the places lds_layout() names
        const LdsLayout lay = lds_layout<ACCEL, LDS, REL>(P);
        if (LDS) {
            size_t off;
            if (ACCEL == 2) {
                off = lay.nodes;
                stage(smem, off, P.bvh, P.bvh_nodes);
                off = lay.prims;
                stage(smem, off, P.bvh_prims, P.bvh_prim_count);
                // staged unconditionally: a pointer that is LDS on one path and global on the other is a generic
                // pointer, and every list read in the leaf loop became a flat_load with a full s_waitcnt
                off = lay.cells;
                stage(smem, off, P.pq_cells, P.pq_cell_count);
                off = lay.list;
                stage(smem, off, P.pq_list, P.pq_list_count);
            } else if (ACCEL == 1) {
                off = lay.oct;
                stage(smem, off, P.oct, P.oct_nodes);
                off = lay.oct_prims;
                stage(smem, off, P.oct_prims, P.oct_prim_count);
            }
            off = lay.spheres;
            stage(smem, off, P.spheres, P.n_prims);
            off = lay.radii;
            stage(smem, off, P.radii, P.n_prims);
            if (REL) {  // node boxes relative to this frame's ray origin (slab_rel)
                double *rel = reinterpret_cast<double *>(smem + lay.rel);
                for (int k = threadIdx.x; k < P.bvh_nodes * 6; k += blockDim.x) {
                    const int node = k / 6, c = k - node * 6;
                    const float v = c < 3 ? P.bvh[node].lo[c] : P.bvh[node].hi[c - 3];
                    rel[k] = static_cast<double>(v) - P.origin_d[c < 3 ? c : c - 3];
                }
            }
            __syncthreads();
        }
    }
    const int list_cap = P.list_cap;
    // The lane id inside the wave loop: two v_mbcnt where it is used (opaque to the optimiser: a value derived from
    // threadIdx would be kept in a register for the whole kernel, and the kernel has exactly as many as six waves allow).
    auto lane_now = []() {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    // this lane's column of the wave's hit-leaf lists (entry e at [e * 64]).  Formed where it is used, from parameters loaded
    // there: a loop-invariant address would be hoisted into a register that lives for the whole kernel.
    auto list_column = [&](const RmRenderParams &C) {
        const uint32_t end = C.lds_off[9];
        return reinterpret_cast<uint16_t *>(smem + end) + (static_cast<uint32_t>(wave) * static_cast<uint32_t>(C.list_cap)) * 64u + static_cast<uint32_t>(lane_now());
    };
    const int item_px = P.item_px;
    // home queue: the XCD this wave really runs on (HW_REG_XCC_ID, id 20, bits [3:0]; blockIdx % 8 otherwise), and the
    // sub-queue its workgroup's number within the XCD selects (workgroups go to the XCDs round-robin)
    int home = (P.hw_xcd ? (__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7) : (static_cast<int>(blockIdx.x) & 7)) * RM_SUBQ +
               static_cast<int>((blockIdx.x >> 3) % RM_SUBQ);
    const int refill_at = P.refill_threshold;  // refill as soon as this many lanes are idle

    // ---- wave state: the tile being consumed ------------------------------------------------
    int tile_col = -1, tile_row = 0, qpos = item_px;  // tile_col < 0: no item yet; qpos: next pixel of the current tile (item_px = used up)
    bool no_more = false;
    // cost feedback for the next frame's longest-first order: when the current item was taken (s_memrealtime, 100 MHz; a
    // cost byte counts units of 10.24 us: the slowest 64-pixel batches of C3 take 0.9 ms).  Until round 3 the cost was the item's
    // wave-loop iterations; the in-round march steps take time without taking iterations.
    unsigned int item_t0 = 0;

    // ---- lane state (round 3: 21 registers instead of ~44; registers decide this kernel's occupancy) ----------------
    //  * nothing wave-uniform: the ray origin is rebuilt from freshly loaded parameters where a section needs it
    //    (a `Ray` assigned under a divergent branch kept its nine origin words in VGPRs);
    //  * the march state {cur.tEnter, cur.tExit, cur.ord} and the normal state {d0, (nx, ny), nz} share sA / sB / sC: a
    //    ray is in one phase or the other (bvh.ts:204-240 against raymarcher.ts:123-135);
    //  * both Uint16Array counters in one word (iterations <= 100 in the low half; the SDF evaluations in the high half,
    //    where `+=` wraps as the store would), the loop trip counter and the ray's flags beside the phase, the list
    //    bookkeeping in one word, the pixel as its buffer index.
    enum : int { ST_PHASE = 7, ST_TRIP_SHIFT = 3, ST_TRIP_MASK = 127 << 3, ST_HAVECUR = 1 << 10, ST_PAR_SHIFT = 11, ST_PAR_MASK = 7 << 11,
                 ST_PIXEL = 1 << 14,
                 ST_LIVE_SHIFT = 15, ST_POS_SHIFT = 22, ST_OVF = 1 << 29, ST_LIST_MASK = (0x3FFF << 15) | (1 << 29) };  // hit-leaf list: live (7 bits), cur_pos (7), overflow
    Vec3f rd = {0.f, 0.f, -1.f};              // ray direction
    double inv0 = 0.0, inv1 = 0.0, inv2 = 0.0;  // 1 / direction (BVH slab tests)
    double t = RM_MAX_DIST;                   // distance marched; after the march it is the returned depth
    double sA = 0.0, sB = 0.0;                // march: cur.tEnter, cur.tExit | normal: d0, (nx, ny) as two binary32 words
    int sC = -1;                              // march: cur.ord | normal: nz
    uint32_t counters = 0;                    // sdfEval << 16 | iterations
    uint32_t pidx = 0;                        // tile-local pixel index (row * width + column; the launcher checks it fits)
    int st = PH_DONE;                         // phase | trips << 3 | flags | hit-leaf list bookkeeping

    auto ray_of = [&](const RmRenderParams &C) {
        Ray r;
        r.d = rd;
        r.o = {C.origin[0], C.origin[1], C.origin[2]};
        r.od[0] = C.origin_d[0];
        r.od[1] = C.origin_d[1];
        r.od[2] = C.origin_d[2];
        return r;
    };
    auto inv_of = [&]() {
        RayInv ri;
        ri.inv[0] = inv0;
        ri.inv[1] = inv1;
        ri.inv[2] = inv2;
        ri.par[0] = (st & (1 << ST_PAR_SHIFT)) != 0;
        ri.par[1] = (st & (2 << ST_PAR_SHIFT)) != 0;
        ri.par[2] = (st & (4 << ST_PAR_SHIFT)) != 0;
        ri.any_par = (st & ST_PAR_MASK) != 0;
        return ri;
    };
    auto list_of = [&](const RmRenderParams &C) {
        RayList L;
        L.col = list_column(C);
        L.cap = list_cap;
        L.live = (st >> ST_LIVE_SHIFT) & 127;
        L.cur_pos = (st >> ST_POS_SHIFT) & 127;
        L.cnt = (st & ST_OVF) ? list_cap + 1 : L.live;  // only `cnt <= cap` is ever asked after the prologue
        return L;
    };
    auto keep_list = [&](const RayList &L) {
        // (list_cap <= 64: live <= 64 and cur_pos <= 63 whenever the list is complete; beyond that only the overflow bit is read)
        st = (st & ~ST_LIST_MASK) | ((L.live & 127) << ST_LIVE_SHIFT) | ((L.cur_pos & 127) << ST_POS_SHIFT) | (L.cnt > list_cap ? ST_OVF : 0);
    };
    auto set_phase = [&](int ph) { st = (st & ~ST_PHASE) | ph; };
    auto nrm_x = [&]() { return __int_as_float(__double2loint(sB)); };
    auto nrm_y = [&]() { return __int_as_float(__double2hiint(sB)); };
    auto set_normal = [&](float nx, float ny, float nz) {
        sB = __hiloint2double(__float_as_int(ny), __float_as_int(nx));
        sC = __float_as_int(nz);
    };
    // march finished with distance `dist_total` (raymarcher.ts:91-102).  No normal is formed when depth >= MAX_DIST
    // (raymarcher.ts:97-99): the store applies that very test to t, so nothing is zeroed here.
    auto finish_march = [&](double dist_total) {
        t = dist_total;
        set_phase((t >= RM_MAX_DIST) ? PH_DONE : PH_N0);
    };
    auto store_mine = [&](const RmRenderParams &C) {
        const bool miss = t >= RM_MAX_DIST;
        const float nx = miss ? 0.f : nrm_x(), ny = miss ? 0.f : nrm_y(), nz = miss ? 0.f : __int_as_float(sC);
        store_pixel(C, static_cast<size_t>(pidx), t, nx, ny, nz, counters >> 16, counters & 0xFFFFu);
        if (C.diag_out) wave_diag_add(&wave_diag[wave], lane_now(), counters >> 16, counters & 0xFFFFu);
    };

    RM_T0()
#ifdef RM_STAMPS_LOG
    int refill_k_ = 0;
#endif
#ifdef RM_COUNTS
    if (threadIdx.x < 32) rm_cnt_s[threadIdx.x] = 0;
    __syncthreads();
#endif
    for (;;) {
        RM_T(7)
        RM_CNT(0)
        // ---- R: active-ray compaction.  Lanes whose ray is finished store their pixel and take
        // the next pixels of the wave's tile stream, assigned by ballot + prefix count. ----------
        const unsigned long long idle = __ballot((st & ST_PHASE) == PH_DONE);
        const int n_idle = __popcll(idle);
        if (n_idle >= refill_at || n_idle == 64) {
            RM_CNT(1)
#ifdef RM_STAMPS_LOG  // diagnostic: when does this wave begin its k-th batch?  (rm_debug_read_batch_log, scripts/batch_timeline.py)
            if (lane == 0 && P.stamps && blockIdx.x * 4 + wave < 2048 && refill_k_ < 96)
                reinterpret_cast<unsigned int *>(P.stamps + 40 + 3 * 8192)[(blockIdx.x * 4 + wave) * 96 + refill_k_] =
                    static_cast<unsigned int>(__builtin_amdgcn_s_memrealtime());
            refill_k_ += 1;
#endif
            const RmRenderParams C = cold_params();
            TileQueue Q;
            Q.counters = C.tile_counters;
            Q.tile_w = C.tile_w;
            Q.item_px = C.item_px;
            Q.tile_h = 1 << C.tile_h_log2;
            Q.tiles_x = C.tiles_x;
            Q.tiles_y = C.tiles_y;
            Q.tiles_x_magic = C.tiles_x_magic;
            Q.perm = C.lpt_perm;
            Q.perm_stride = C.lpt_stride;
            const bool want_tile = !no_more && qpos >= Q.item_px;
            const int lane_r = lane_now();
            if (want_tile) {
                // (before the pixel stores are issued: the wait for the claim's answer then covers no store of this refill.
                // Tried in round 3 and dropped: claiming one item AHEAD so that the answer is never waited for -- no gain with
                // frames in flight (919 against 949 frames/s), none alone, one more register held across the wave loop.)
                if (C.lpt_cost_out && tile_col >= 0) {  // the item just finished: what it cost, at its slot of the queue it was pulled from (= home)
                    const unsigned int took = (static_cast<unsigned int>(__builtin_amdgcn_s_memrealtime()) - item_t0) >> 10;  // units of 10.24 us
                    if (lane_r == 0) C.lpt_cost_out[static_cast<size_t>(home) * C.lpt_stride + queue_slot_of(Q, home, tile_col, tile_row)] =
                        static_cast<uint8_t>(took < 255u ? (took ? took : 1u) : 255u);
                }
#ifdef RM_STAMPS_CLAIM  // diagnostic: how long does the wave wait for its claim?
                const unsigned long long tc0 = __builtin_amdgcn_s_memrealtime();
#endif
                unsigned int claim = 0;
                if (lane_r == 0) claim = atomicAdd(&Q.counters[home * RM_QSTRIDE], 1u);
                if (pull_tile(Q, home, tile_col, tile_row, lane_r, claim)) {
                    qpos = 0;
                    if (C.lpt_cost_out) item_t0 = static_cast<unsigned int>(__builtin_amdgcn_s_memrealtime());
                } else no_more = true;
#ifdef RM_STAMPS_CLAIM
                const unsigned long long tc1 = __builtin_amdgcn_s_memrealtime();
                if (lane_r == 0 && P.stamps) {
                    atomicAdd(&P.stamps[38], tc1 - tc0);
                    atomicAdd(&P.stamps[39], 1ull);
                }
#endif
            }
            if ((st & (ST_PHASE | ST_PIXEL)) == (PH_DONE | ST_PIXEL)) {
                store_mine(C);
                st &= ~ST_PIXEL;
            }
            if (!no_more) {
                const int remaining = Q.item_px - qpos;
                const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned int>(idle >> 32),
                                                           __builtin_amdgcn_mbcnt_lo(static_cast<unsigned int>(idle), 0u));
                const bool take = (st & ST_PHASE) == PH_DONE && rank < remaining;
                bool in_frame = false;  // this lane received a pixel inside the frame
                // a whole batch from one 64-pixel sub-tile (the normal case): its hit leaves come from the bundle cull
                const bool whole_batch = ACCEL == 2 && C.bvh_leaf_count > 0 && n_idle == 64 && (qpos & 63) == 0;
                const int batch_sub = qpos >> 6;
                Ray ray = ray_of(C);
                RayInv ri = inv_of();
                if (take) {
                    const int n = qpos + rank;  // pixel n of the tile, in 64-pixel sub-tile order
                    const int sub = n >> 6, l = n & 63;
                    const int px = (tile_col << C.item_w_log2) + sub * C.sub_dx + (l & (Q.tile_w - 1));
                    const int prow = (tile_row << C.tile_h_log2) + sub * C.sub_dy + (l >> C.tile_w_log2);
                    if (px < C.width && prow < C.local_rows) {
                        in_frame = true;
                        pidx = static_cast<uint32_t>(prow) * static_cast<uint32_t>(C.width) + static_cast<uint32_t>(px);
                        ray = make_ray(C, px, row_to_y(C, prow));
                        rd = ray.d;
                        counters = 0;
                        t = 0.0;
                        st = PH_MARCH | ST_PIXEL;  // no trips yet, no interval, no parallel axis
                        if (ACCEL == 2) {
                            ri = make_ray_inv(ray);
                            inv0 = ri.inv[0];
                            inv1 = ri.inv[1];
                            inv2 = ri.inv[2];
                            st |= ((ri.par[0] ? 1 : 0) | (ri.par[1] ? 2 : 0) | (ri.par[2] ? 4 : 0)) << ST_PAR_SHIFT;
                        }
                    }
                }
                if (ACCEL == 2) {
#ifdef RM_STAMPS
                    const unsigned long long t_pr0 = __builtin_amdgcn_s_memtime();
#endif
                    RayList L = list_of(C);
                    Interval cur;
                    cur.tEnter = sA;
                    cur.tExit = sB;
                    cur.ord = sC;
                    bool hc = false;
                    if (whole_batch) {
                        const int x0 = (tile_col << C.item_w_log2) + batch_sub * C.sub_dx;
                        const int r0 = (tile_row << C.tile_h_log2) + batch_sub * C.sub_dy;
                        const Bundle B = make_bundle(C, x0, x0 + Q.tile_w - 1, row_to_y(C, r0),
                                                     row_to_y(C, r0 + (64 >> C.tile_w_log2) - 1));
                        hc = bvh_prologue_cull<REL>(scene_view<ACCEL, LDS, REL>(C, smem), C, B, in_frame, ray, ri, L, cur, lane_now());
                    } else if (in_frame) {
                        RM_CNT(15)
                        hc = bvh_prologue<REL>(scene_view<ACCEL, LDS, REL>(C, smem), ray, ri, L, cur);
                    }
#ifdef RM_STAMPS
                    t_acc_[5] += __builtin_amdgcn_s_memtime() - t_pr0;
                    t_prev_ += __builtin_amdgcn_s_memtime() - t_pr0;
#endif
                    if (in_frame) {
                        keep_list(L);
                        if (hc) {
                            sA = cur.tEnter;
                            sB = cur.tExit;
                            sC = cur.ord;
                            st |= ST_HAVECUR;
                        } else {  // bvh.ts:190-192: exactly MAX_DIST, zero normal
                            t = RM_MAX_DIST;
                            set_phase(PH_DONE);
                        }
                    }
                }
                qpos += n_idle < remaining ? n_idle : remaining;
            }
        }
        if (no_more && !__any((st & ST_PHASE) != PH_DONE)) {
            if (st & ST_PIXEL) store_mine(cold_params());
            break;
        }
        RM_T(0)

        // ---- A: bookkeeping until this lane needs a distance (sphereTracer.ts:43-64) ------
        // Every section works on a SceneView rebuilt from freshly loaded parameters (see lds_layout): nothing of it
        // lives in SGPRs across the wave loop.
        const RmRenderParams CA = cold_params();
        const SceneView S = scene_view<ACCEL, LDS, REL>(CA, smem);
        const Ray ray = ray_of(CA);
        bool need = false;
        Vec3f q = {0.f, 0.f, 0.f};
        int onode = -1;
        if ((st & ST_PHASE) == PH_MARCH) {
            for (;;) {
                if ((st & ST_TRIP_MASK) >= (RM_MAX_STEPS << ST_TRIP_SHIFT)) {  // loop exhausted: return totalDist
                    finish_march(t);
                    break;
                }
                st += 1 << ST_TRIP_SHIFT;
                RM_CNT(2)
                const Vec3f p = point_at(ray, t);
                if (ACCEL == 2) {
                    // BVH.onRayMarchStep (bvh.ts:204-240)
                    double skip = 0.0;
                    bool terminate = !(st & ST_HAVECUR);
                    if (!terminate) {
                        if (t < sA) skip = sA - t;
                        else if (t > sB) {
                            RayList L = list_of(CA);
                            Interval cur;
                            cur.tEnter = sA;
                            cur.tExit = sB;
                            cur.ord = sC;
                            const bool hc = bvh_next<REL>(S, ray, inv_of(), L, sA, sC, cur);
                            keep_list(L);
                            if (!hc) {
                                st &= ~ST_HAVECUR;
                                terminate = true;
                            } else {
                                sA = cur.tEnter;
                                sB = cur.tExit;
                                sC = cur.ord;
                                if (sA > t) skip = sA - t;
                            }
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
        // BVH kernels defer getNormal: a ray that has finished marching waits in PH_N0 until no lane of the wave needs a
        // march distance any more; then all of them take ONE round for the hit point (n0_go) and, where the evaluation
        // allows it, get the three offset distances from the same sphere without three more rounds (section N below).
        bool n0_go = true;
        if (ACCEL == 2) n0_go = !__any(need) || __popcll(__ballot((st & ST_PHASE) == PH_N0)) >= CA.n0_batch;
        {
            const int ph = st & ST_PHASE;
            if (ph >= PH_N0 && ph <= PH_N3 && (ph != PH_N0 || n0_go)) {  // raymarcher.ts:123-132 sample points
                RM_CNT(14)
                q = point_at(ray, t);  // hitPosition (raymarcher.ts:94-95), recomputed: 3 VGPRs fewer
                if (ph == PH_N1) q.x = to_f32(static_cast<double>(q.x) - 0.01);
                if (ph == PH_N2) q.y = to_f32(static_cast<double>(q.y) - 0.01);
                if (ph == PH_N3) q.z = to_f32(static_cast<double>(q.z) - 0.01);
                if (ACCEL == 1) onode = oct_find(S, q);
                need = true;
            }
        }
        RM_T(1)
        if (!__any(need)) continue;  // every live ray just finished: go and refill

        // ---- B: one Scene.getDistance per needing lane --------------------------------------
        if (need) RM_CNT(6)
        const RmRenderParams CB = cold_params();
        const SceneView SB = scene_view<ACCEL, LDS, REL>(CB, smem);
        const bool coop = CB.coop != 0, filter = CB.filter != 0;
        const bool use_grid = ACCEL == 2 && CB.use_grid != 0;
        double dist;
        const int lane_b = lane_now();
        uint32_t evaluated = 0;  // primitives this round counts (raymarcher.ts:117-119)
        NormalAux aux;
        aux.ok = false;
        aux.k1 = 0;
        aux.lb2 = 0.f;
        aux.found = 0;
        aux.rho = 0.f;
        aux.rho_cell = 0.f;
        if (ACCEL == 2) {
#ifdef RM_STAMPS
            unsigned long long fbc = 0;
            dist = bvh_distance_wave<UR>(CB, SB, need, q, evaluated, lane_b, coop, filter, use_grid, &fbc, &aux);
            t_acc_[4] += fbc;
            t_prev_ += fbc;  // keep section 2 = query + leaf evaluation only
#else
            dist = bvh_distance_wave<UR>(CB, SB, need, q, evaluated, lane_b, coop, filter, use_grid, nullptr, &aux);
#endif
        }
        else if (ACCEL == 1) dist = need ? oct_distance_lane(SB, onode, q, evaluated, filter) : RM_MAX_DIST;
        else {
            dist = all_prims_wave(SB, need, q, lane_b, coop, filter);
            if (need) evaluated = static_cast<uint32_t>(SB.n_prims);
        }
        counters += evaluated << 16;  // Uint16Array += : the carry out of the upper half is the wrap

        RM_T(2)
        // ---- C: consume -------------------------------------------------------------------------
        const bool fuse = ACCEL == 2 && need && (st & ST_PHASE) == PH_N0 && aux.ok && aux.rho > RM_NRM_DELTA;
        const bool was_marching = need && (st & ST_PHASE) == PH_MARCH;
        if (need) {
            const int ph = st & ST_PHASE;
            if (ph == PH_MARCH) {
                t += dist;
                counters += 1;
                if (dist < RM_EPSILON || t > RM_MAX_DIST) finish_march(t);
            } else if (ph == PH_N0) {
                sA = dist;  // d0
                set_phase(PH_N1);
            } else if (ph == PH_N1) {
                sB = __hiloint2double(__double2hiint(sB), __float_as_int(to_f32(sA - dist)));  // nx
                set_phase(PH_N2);
            } else if (ph == PH_N2) {
                sB = __hiloint2double(__float_as_int(to_f32(sA - dist)), __double2loint(sB));  // ny
                set_phase(PH_N3);
            } else {
                float nx = nrm_x(), ny = nrm_y(), nz = to_f32(sA - dist);
                normalize3(nx, ny, nz);
                set_normal(nx, ny, nz);
                set_phase(PH_DONE);
            }
        }
        // ---- M: more march steps inside this round (sphereTracer.ts:43-75, bvh.ts:204-240, scene.ts:167-181), for as long as
        // they provably see what this round saw.  A ray close to a surface takes step after step against the same sphere: a
        // ray that grazes one takes up to a hundred, each a whole round of the wave loop with one or two lanes at work --
        // the slowest rays of a batch set its number of rounds.  The next step is taken HERE, by the lane alone, when
        //  * the march goes on (trip count, epsilon and MAX_DIST tests as the loop makes them) and the point's parameter
        //    lies inside the current interval, so that onRayMarchStep answers 0 without touching its state;
        //  * the new point p2 is within rho of this round's point q (distance measured between the two binary32 points,
        //    rounded up), rho being the stability radius of the leaf set (box_margin of every leaf the cell lists, and not
        //    beyond what that list covers: the cell grown by 0.0100): getPrimitivesAt returns the same leaves, the counter
        //    advances by the same `found`;
        //  * every other candidate, at least lb2 away at q and 1-Lipschitz, is still farther at p2 than the exact
        //    distance of the sphere k1 that won at q: Math.min over the candidates is that distance.
        // The step is then exactly the one the ordinary path would take (same point, same value, same counters); a lane
        // that fails a test has committed nothing of that step and takes it in the next round.
        if (ACCEL == 2 && CB.multi_step) {
            const bool go = was_marching && aux.ok && (st & ST_PHASE) == PH_MARCH;
            if (__any(go)) {
                if (go) {
                    const SceneView SM = scene_view<ACCEL, LDS, REL>(cold_params(), smem);
                    const RmSphere s1 = SM.spheres[aux.k1];
                    const double r1 = SM.radii[aux.k1];
                    const float rho = __builtin_fminf(aux.rho, aux.rho_cell);
                    for (;;) {
                        if ((st & ST_TRIP_MASK) >= (RM_MAX_STEPS << ST_TRIP_SHIFT)) {  // loop exhausted: return totalDist
                            finish_march(t);
                            break;
                        }
                        if (t < sA || t > sB) break;  // the interval state machine has work to do: the ordinary path
                        const Vec3f p2 = point_at(ray, t);
                        const float ex = p2.x - q.x, ey = p2.y - q.y, ez = p2.z - q.z;
                        const float dd = __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez) * 1.00001f + 1e-7f;  // >= |p2 - q|
                        if (!(dd < rho)) break;
                        const double e = sphere_sdf_fast(s1, r1, p2);
                        if (!(aux.lb2 - dd > f32_upper_bound(e))) break;
                        RM_CNT(15)
                        st += 1 << ST_TRIP_SHIFT;
                        counters += (aux.found << 16) + 1u;
                        const double d2 = __builtin_fmin(e, RM_MAX_DIST);  // Math.min(sdf, closestDistance = 10)
                        t += d2;
                        if (d2 < RM_EPSILON || t > RM_MAX_DIST) {
                            finish_march(t);
                            break;
                        }
                    }
                }
            }
        }
        // ---- N: the three offset samples of getNormal (raymarcher.ts:126-132) from the sphere that gave d0.  The leaf
        // set is the same at the offset points (aux.ok: box_answer_is_stable for every listed leaf and the root), so the
        // same spheres are the candidates there and the counter advances by the same number; Sphere.sdf is 1-Lipschitz, so
        // every candidate other than k1 is at least lb2 - delta away at an offset point.  If that exceeds the exact distance
        // of k1 there, k1 is the minimum (Math.min is order independent): three exact evaluations replace three rounds.
        // A lane for which one of the three checks fails keeps d0 and goes on through PH_N1..PH_N3 as before.
        if (ACCEL == 2 && __any(fuse)) {
            if (fuse) {
                const SceneView SN = scene_view<ACCEL, LDS, REL>(cold_params(), smem);
                const RmSphere s1 = SN.spheres[aux.k1];
                const double r1 = SN.radii[aux.k1];
                Vec3f qx = q, qy = q, qz = q;
                qx.x = to_f32(static_cast<double>(q.x) - 0.01);
                qy.y = to_f32(static_cast<double>(q.y) - 0.01);
                qz.z = to_f32(static_cast<double>(q.z) - 0.01);
                const double e1 = sphere_sdf_fast(s1, r1, qx), e2 = sphere_sdf_fast(s1, r1, qy), e3 = sphere_sdf_fast(s1, r1, qz);
                const double emax = __builtin_fmax(__builtin_fmax(e1, e2), e3);
                if (aux.lb2 - RM_NRM_DELTA > f32_upper_bound(emax)) {
                    counters += (3u * aux.found) << 16;
                    float nx = to_f32(sA - __builtin_fmin(e1, RM_MAX_DIST));  // Math.min(sdf, closestDistance = 10)
                    float ny = to_f32(sA - __builtin_fmin(e2, RM_MAX_DIST));
                    float nz = to_f32(sA - __builtin_fmin(e3, RM_MAX_DIST));
                    normalize3(nx, ny, nz);
                    set_normal(nx, ny, nz);
                    set_phase(PH_DONE);
                }
            }
        }
        RM_T(3)
    }
    RM_TEND()
    {   // the wave's totals to the launch's accumulator block; the launch's last wave publishes the result and re-zeroes
        // the block and the tile-queue heads (rm_diag.h).  All 64 lanes are active here (the loop's exit is wave-uniform).
        const RmRenderParams C = cold_params();
        if (C.diag_block) {
            const WaveDiag *w = &wave_diag[wave];
            unsigned long long ts = 0, ti = 0;
            unsigned int mx = 0, mi = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {  // (every lane reads the same eight slots: broadcasts)
                ts += w->sdf[k];
                ti += w->iters[k];
                mx = w->mx[k] > mx ? w->mx[k] : mx;
                mi = w->mn_inv[k] > mi ? w->mn_inv[k] : mi;
            }
            diag_flush_wave(C.diag_block, C.diag_out, C.tile_counters, ts, ti, mx, mi, blockIdx.x * 4u + static_cast<unsigned int>(wave),
                            gridDim.x * 4u, lane_now());
        }
    }
#ifdef RM_COUNTS
    __syncthreads();
    if (threadIdx.x < 32 && P.stamps) atomicAdd(&P.stamps[8 + threadIdx.x], static_cast<unsigned long long>(rm_cnt_s[threadIdx.x]));
#endif
}

#ifdef RM_RTC_V2
}  // namespace
extern "C" __global__ __launch_bounds__(256, RM_V2_WAVES) void rm_rtc_render_v2(const RmRenderParams P) {
    render_v2_body<RM_RTC_V2_ACCEL, RM_RTC_V2_LDS != 0, RM_RTC_V2_UR != 0, RM_RTC_V2_REL != 0>(P);
}
#else
template <int ACCEL, bool LDS, bool UR = false, bool REL = false>
__global__ __launch_bounds__(256, RM_V2_WAVES) void render_kernel_v2(const RmRenderParams P) {
    render_v2_body<ACCEL, LDS, UR, REL>(P);
}

size_t scene_lds_bytes(const RmRenderParams &p) {
    auto up = [](size_t v) { return (v + 15) & ~static_cast<size_t>(15); };
    size_t b = 0;
    if (p.accel == 2) {
        b += up(static_cast<size_t>(p.bvh_nodes) * sizeof(RmBvhNode)) + up(static_cast<size_t>(p.bvh_prim_count) * 4);
        b += up(static_cast<size_t>(p.pq_cell_count) * 4) + up(static_cast<size_t>(p.pq_list_count) * 2 + 4);
    } else if (p.accel == 1) {
        b += up(static_cast<size_t>(p.oct_nodes) * sizeof(RmOctNode)) + up(static_cast<size_t>(p.oct_prim_count) * 4);
    }
    b += up(static_cast<size_t>(p.n_prims) * sizeof(RmSphere)) + up(static_cast<size_t>(p.n_prims) * 8);
    return b + 16;
}

}  // namespace

#ifndef RM_LENGTH_SQRT
namespace {
// Longest-first order of a frame's work items from the costs the PREVIOUS frame recorded (lpt_cost_prev; a frame loop renders
// nearly the same picture again, and any permutation is a valid order, so stale or missing costs only cost balance).
// A frame that runs alone ends with a ramp: its 4096 waves finish over the last ~0.5 ms (scripts/tail_hist.py), because a
// wave's last item may be one with a 100-step grazing ray.  Handing out the expensive items first leaves cheap ones for the
// end.  One workgroup per queue (RM_QUEUES); four cost classes (>= 4, 2, 1.25 x the mean, the rest) and a STABLE partition, so that
// inside a class the queue keeps its own order -- horizontally adjacent tiles stay adjacent in time and their partial-line
// stores still merge in L2.  The costs are copied to LDS first: the previous launch may still be writing them, and the
// permutation must be built from ONE snapshot to be a permutation.
__global__ __launch_bounds__(1024) void lpt_sort_kernel(const uint8_t *cost_prev, uint16_t *perm, int stride, int tiles_x, int tiles_y) {
    extern __shared__ unsigned char lpt_smem[];
    const int x = blockIdx.x;  // queue id
    const int R = queue_rows(tiles_y, x);
    const int n = R * tiles_x;
    uint8_t *cost = lpt_smem;                                                   // n bytes
    unsigned int *cnt = reinterpret_cast<unsigned int *>(lpt_smem + ((stride + 15) & ~15));  // [4][1024]
    __shared__ unsigned int sum_s, nz_s, base_s[4];
    if (threadIdx.x == 0) sum_s = nz_s = 0;
    __syncthreads();
    unsigned int sum = 0, nz = 0;
    for (int e = threadIdx.x; e < n; e += 1024) {
        const uint8_t c = cost_prev ? cost_prev[static_cast<size_t>(x) * stride + e] : 0;
        cost[e] = c;
        sum += c;
        nz += c != 0;
    }
    atomicAdd(&sum_s, sum);
    atomicAdd(&nz_s, nz);
    __syncthreads();
    const float mean = nz_s ? static_cast<float>(sum_s) / static_cast<float>(nz_s) : 1e9f;
    auto cls = [&](uint8_t c) { const float f = c; return f >= 4.f * mean ? 0 : (f >= 2.f * mean ? 1 : (f >= 1.25f * mean ? 2 : 3)); };
    const int chunk = (n + 1023) / 1024, e0 = threadIdx.x * chunk, e1 = min(n, e0 + chunk);
    unsigned int mine[4] = {0, 0, 0, 0};
    for (int e = e0; e < e1; ++e) mine[cls(cost[e])]++;
    for (int c = 0; c < 4; ++c) cnt[c * 1024 + threadIdx.x] = mine[c];
    __syncthreads();
    if (threadIdx.x < 4) {  // exclusive scan of one class over the 1024 chunks
        unsigned int run = 0;
        for (int t = 0; t < 1024; ++t) {
            const unsigned int v = cnt[threadIdx.x * 1024 + t];
            cnt[threadIdx.x * 1024 + t] = run;
            run += v;
        }
        base_s[threadIdx.x] = run;  // class total
    }
    __syncthreads();
    unsigned int start[4], acc = 0;
    for (int c = 0; c < 4; ++c) {
        start[c] = acc + cnt[c * 1024 + threadIdx.x];
        acc += base_s[c];
    }
    for (int e = e0; e < e1; ++e) {
        const int c = cls(cost[e]);
        perm[static_cast<size_t>(x) * stride + start[c]++] = static_cast<uint16_t>(e);
    }
}
}  // namespace
hipError_t rm_launch_lpt_sort(const uint8_t *cost_prev, uint16_t *perm, int stride, int tiles_x, int tiles_y, hipStream_t stream) {
    const size_t shmem = static_cast<size_t>((stride + 15) & ~15) + 4 * 1024 * sizeof(unsigned int);
    hipLaunchKernelGGL(lpt_sort_kernel, dim3(RM_QUEUES), dim3(1024), shmem, stream, cost_prev, perm, stride, tiles_x, tiles_y);
    return hipGetLastError();
}
#endif

hipError_t RM_LEN_VARIANT(rm_launch_render_v2)(const RmRenderParams &p_in, hipStream_t stream, const char **kernel_name) {
    RmRenderParams p = p_in;
    const int rows = p.local_rows;
    if (kernel_name) *kernel_name = "";
    if (rows <= 0 || p.width <= 0) return hipSuccess;
    if (!p.tile_counters) return hipErrorInvalidValue;
    if (p.leaf_order) p.bvh_prim_count = 0;  // the leaf lists are the identity (spheres stored in leaf order): nothing reads them, nothing is staged
    if (p.item_px != 64 && p.item_px != 128 && p.item_px != 256) p.item_px = 64;
    if (p.item_px < p.tile_w) p.item_px = p.tile_w;
    const int tw = p.tile_w;
    const bool wide = p.item_wide != 0 && tw < 64 && p.item_px > 64;  // batches side by side: item_px / 64 of them, each tw x 64 / tw
    const int iw = wide ? tw * (p.item_px / 64) : tw, th = wide ? 64 / tw : p.item_px / tw;
    const int tiles_x = (p.width + iw - 1) / iw;
    const int tiles_y = (rows + th - 1) / th;
    auto log2i = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    if ((tw & (tw - 1)) != 0 || tiles_x <= 0 || static_cast<long long>(tiles_x) * (tiles_y + 8) >= (1ll << 31)) return hipErrorInvalidValue;
    p.tile_w_log2 = log2i(tw);
    p.tile_h_log2 = log2i(th);
    p.item_w_log2 = log2i(iw);
    p.sub_dx = wide ? tw : 0;
    p.sub_dy = wide ? 0 : 64 / tw;
    p.tiles_x = tiles_x;
    p.tiles_y = tiles_y;
    p.tiles_x_magic = tiles_x == 1 ? 0u : static_cast<uint32_t>((1ull << 32) / static_cast<unsigned long long>(tiles_x)) + 1u;  // 0: k / 1
    const unsigned needed = static_cast<unsigned>((static_cast<long long>(tiles_x) * tiles_y + 3) / 4);  // 4 waves each
    unsigned resident = 0, blocks = 0;  // set below, once the LDS footprint (workgroups per CU) is known
    if (p.list_cap < 1) p.list_cap = 1;
    if (p.refill_threshold < 1) p.refill_threshold = 1;
    if (p.refill_threshold > 64) p.refill_threshold = 64;
    size_t list_bytes = p.accel == 2 ? static_cast<size_t>(4) * p.list_cap * 128 : 0;
    const size_t scene_bytes = scene_lds_bytes(p);
    // stage the scene in LDS when it leaves room for >= 2 workgroups per CU (160 KB LDS)
    const bool lds = p.nodes_in_lds != 0 && scene_bytes + list_bytes + 16 <= 64 * 1024;
    // LDS budget per workgroup.  The kernels hold 80 VGPRs: six waves per SIMD = six four-wave workgroups per CU, if six
    // fit the CU's 160 KB of LDS.  LDS is handed out in granules of 1 280 bytes (measured in round 3: a workgroup of
    // 32 096 bytes, dynamic + static, no longer ran five to a CU -- 26 granules = 33 280 -- and the frame rate with frames in
    // flight fell 5 %; hipOccupancyMaxActiveBlocksPerMultiprocessor does not model it), so the budgets are whole granules:
    // 26 880 bytes for six, 32 000 for five, 40 960 for four.  The per-ray hit lists give way down to 12 entries to reach a
    // budget (rays that hit more leaves take the tree-walk form of bvh_next, as they do beyond any cap); origin-relative
    // node boxes (48 B per node) ride along when they fit the same budget.  Option `lds_kb` (> 0) names a budget itself.
    bool rel = false;
    const size_t rel_bytes = static_cast<size_t>(p.bvh_nodes) * 48;
    if (lds && p.accel == 2) {
        constexpr size_t kLdsPerCu = 160 * 1024, kGranule = 1280, kStatic = sizeof(WaveDiag) * 4;
        auto budget_for = [&](size_t bytes) { return bytes / kGranule * kGranule - kStatic; };  // dynamic bytes of a workgroup
        auto fits = [&](size_t extra, int cap, size_t budget) { return scene_bytes + extra + static_cast<size_t>(4) * cap * 128 + 16 <= budget; };
        auto trim = [&](size_t extra, size_t budget) {  // largest cap <= list_cap (>= 12, or list_cap itself if smaller) that fits; 0 if none
            int cap = p.list_cap;
            while (cap > 12 && !fits(extra, cap, budget)) cap -= 1;
            return fits(extra, cap, budget) ? cap : 0;
        };
        size_t budgets[4];
        int nb = 0;
        if (p.lds_budget_kb > 0) budgets[nb++] = budget_for(static_cast<size_t>(p.lds_budget_kb) * 1024);
        else {
            budgets[nb++] = budget_for(kLdsPerCu / RM_V2_WAVES);
            budgets[nb++] = budget_for(kLdsPerCu / 5);
        }
        budgets[nb++] = budget_for(kLdsPerCu / 4);
        for (int b = 0; b < nb; ++b) {
            int cap = p.rel_boxes ? trim(rel_bytes + 16, budgets[b]) : 0;
            if (cap > 0) rel = true;
            else cap = trim(0, budgets[b]);
            if (cap > 0) {
                p.list_cap = cap;
                list_bytes = static_cast<size_t>(4) * cap * 128;
                break;
            }
            rel = false;
        }
    }
    size_t shmem = (lds ? scene_bytes : 0) + (rel ? rel_bytes + 16 : 0) + list_bytes + 16;
    // Option `lds_fill`: a launch of k persistent workgroups per CU asks for as much LDS as still lets k of them share a CU, so
    // that the dispatcher CANNOT put more than k on one CU (and fewer on another): a frame that runs alone gets exactly k on
    // every CU.  Never with one workgroup per CU (frames in flight: the CU is to be shared with the other launches).
    if (p.lds_fill && p.blocks_per_cu >= 2 && p.blocks_per_cu <= 8) {
        const size_t room = (160 * 1024 / static_cast<size_t>(p.blocks_per_cu)) / 1280 * 1280 - sizeof(WaveDiag) * 4;
        if (room > shmem && room <= 64 * 1024) shmem = room;
    }
    {
        const LdsLayout lay = lds_layout_host(p, p.accel, lds, rel && p.accel == 2);
        const uint32_t v[10] = {lay.nodes, lay.prims, lay.cells, lay.list, lay.oct, lay.oct_prims, lay.spheres, lay.radii, lay.rel, lay.end};
        for (int k = 0; k < 10; ++k) p.lds_off[k] = v[k];
        if (lay.end + list_bytes > shmem) return hipErrorInvalidValue;  // the layout and the allocation come from two formulas: they must agree
    }
    resident = static_cast<unsigned>(p.num_cus > 0 ? p.num_cus : 256) * static_cast<unsigned>(p.blocks_per_cu > 0 ? p.blocks_per_cu : 4);
    blocks = needed < resident ? (needed ? needed : 1u) : resident;
    // the tile-queue heads are zero: a launch's last wave leaves them so (rm_diag.h).  Without an accumulator block
    // (a caller below the API layer) they are cleared here.
    hipError_t e = hipSuccess;
    if (!p.diag_block) e = hipMemsetAsync(p.tile_counters, 0, RM_QUEUES * RM_QSTRIDE * sizeof(unsigned int), stream);
    if (e != hipSuccess) return e;
    // longest-first item order (option `lpt`): the api layer hands in the cost / permutation buffers and their stride
    p.lpt_perm = nullptr;
    if (p.lpt_perm_out) {
        const long long per_queue = static_cast<long long>(queue_rows_host(tiles_y, 0)) * tiles_x;  // queue 0 is never shorter than another
        if (per_queue <= p.lpt_stride && per_queue <= 32768) {
            e = rm_launch_lpt_sort(p.lpt_cost_prev, p.lpt_perm_out, p.lpt_stride, tiles_x, tiles_y, stream);
            if (e != hipSuccess) return e;
            p.lpt_perm = p.lpt_perm_out;
        } else {
            p.lpt_cost_out = nullptr;
        }
    } else {
        p.lpt_cost_out = nullptr;
    }
    const dim3 grid(blocks), block(256);
    // A launch configuration that keeps coming back runs in a copy of this kernel compiled for it, its configuration
    // parameters literals (rm_v2_fields.h; rm_api.cpp decides when, rm_rtc.cpp compiles): same grid, same LDS, same bytes.
#define RM_V2X(A, L, U, R)                                                                                                   \
    {                                                                                                                        \
        const void *fn_ = p.rtc_ctx ? rm_rtc_v2_hook(p, A, L, U, R, RM_LEN_IS_SQRT) : nullptr;                               \
        if (fn_) {                                                                                                           \
            size_t bytes_ = sizeof p;                                                                                        \
            void *extra_[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes_, HIP_LAUNCH_PARAM_END}; \
            e = hipModuleLaunchKernel(reinterpret_cast<hipFunction_t>(const_cast<void *>(fn_)), grid.x, 1, 1, 256, 1, 1,     \
                                      static_cast<unsigned>(shmem), stream, nullptr, extra_);                               \
            if (e != hipSuccess) return e;                                                                                   \
            if (kernel_name) *kernel_name = "render_kernel_v2<" #A ", " #L ", " #U ", " #R ">" RM_LEN_TAG " [launch constants compiled in]"; \
        } else {                                                                                                             \
            hipLaunchKernelGGL((render_kernel_v2<A, L, U, R>), grid, block, shmem, stream, p);                               \
            if (kernel_name) *kernel_name = "render_kernel_v2<" #A ", " #L ", " #U ", " #R ">" RM_LEN_TAG;                   \
        }                                                                                                                    \
    }
#define RM_V2(A, L) RM_V2X(A, L, false, false)
    if (p.accel == 2) {
        if (lds && rel && p.uniform_radius) RM_V2X(2, true, true, true)
        else if (lds && rel) RM_V2X(2, true, false, true)
        else if (lds && p.uniform_radius) RM_V2X(2, true, true, false)
        else if (lds) RM_V2(2, true)
        else RM_V2(2, false)
    }
    else if (p.accel == 1) { if (lds) RM_V2(1, true) else RM_V2(1, false) }
    else { if (lds) RM_V2(0, true) else RM_V2(0, false) }
#undef RM_V2X
#undef RM_V2
    return hipGetLastError();
}
#endif  // !RM_RTC_V2
