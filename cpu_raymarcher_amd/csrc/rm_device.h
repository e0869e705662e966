// rm_device.h -- device-side building blocks shared by the gfx950 kernels: JS number
// semantics, the sphere SDF, box tests, the octree empty-space skip, ray generation,
// shading and the pixel store.  Everything here restates a reference function; the citation
// is on each helper (paths relative to the reference's src/).
//
// Build with -ffp-contract=off: JS arithmetic has one rounding per operation, no FMA.
#pragma once
#ifndef __HIPCC_RTC__  // (hiprtc brings its own runtime declarations: the run-time specialiser of rm_rtc.cpp compiles these files too)
#include <hip/hip_runtime.h>
#endif

#include "rm_types.h"

namespace rmd {

// ------------------------------------------------------------------ number semantics

__device__ __forceinline__ float to_f32(double v) { return static_cast<float>(v); }  // Float32Array store

// Uint8ClampedArray store: NaN -> 0, clamp, round half to even (v_rndne_f64)
__device__ __forceinline__ uint8_t u8clamp(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 255.0) return 255;
    return static_cast<uint8_t>(__builtin_rint(v));
}

// V8 Math.hypot(x, y, z) for finite binary32-valued inputs: scale by the largest
// magnitude, Kahan-sum the squares in argument order, sqrt, rescale.  The first Kahan
// step is folded by hand (sum = 0, compensation = 0 make it exact).
__device__ __forceinline__ double hypot3(float lx, float ly, float lz) {
    const double ax = __builtin_fabs(static_cast<double>(lx));
    const double ay = __builtin_fabs(static_cast<double>(ly));
    const double az = __builtin_fabs(static_cast<double>(lz));
    const double big = __builtin_fmax(__builtin_fmax(ax, ay), az);  // no NaN among the operands
    if (big == 0.0) return 0.0;
    const double nx = ax / big, ny = ay / big, nz = az / big;
    double sum = nx * nx;
    const double sy = ny * ny;  // summand - compensation(=0)
    double next = sum + sy;
    const double comp = (next - sum) - sy;
    sum = next;
    const double sz = nz * nz - comp;
    sum = sum + sz;
    return __builtin_sqrt(sum) * big;
}

// The same value with the three divisions by `big` sharing one reciprocal refinement.
// hipcc expands an IEEE f64 division x / y (no fast-math) as
//     r0 = v_rcp_f64(y); e0 = fma(-y,r0,1); r1 = fma(r0,e0,r0); e1 = fma(-y,r1,1); r2 = fma(r1,e1,r1);
//     m = x*r2; e = fma(-y,m,x); q = fma(e,r2,m)      (+ v_div_scale / v_div_fmas / v_div_fixup)
// where the scale/fixup instructions only act on operands near the ends of the binary64
// range or on zero/inf/NaN.  Here y = max(|lx|,|ly|,|lz|) > 0 and x <= y are binary32
// magnitudes (>= 2^-149, < 2^128), so no scaling ever triggers, x = 0 gives q = 0 on both
// paths, and r2 depends on y alone: computing it once and reusing it yields bit-identical
// quotients with 14 instead of 30 instructions (one quarter-rate v_rcp_f64 instead of three).
// tests/test_gpu_parity.py::test_device_hypot_is_v8_math_hypot compares the two on device.
// sqrt for x in [1, 4): the compiler's IEEE expansion of __builtin_sqrt (v_rsq_f64, one coupled
// Goldschmidt step, two residual corrections) without its input scaling for x < 2^-767 and its
// zero / infinity fix-up, neither of which can trigger in this range: same ten instructions in the
// middle, eight fewer around them, same correctly rounded result (rm_selftest_fastdiv compares).
__device__ __forceinline__ double sqrt_unit_range(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = r * 0.5;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return g;
}

__device__ __forceinline__ double hypot3_shared_rcp(float lx, float ly, float lz) {
    const double ax = __builtin_fabs(static_cast<double>(lx));
    const double ay = __builtin_fabs(static_cast<double>(ly));
    const double az = __builtin_fabs(static_cast<double>(lz));
    const double big = __builtin_fmax(__builtin_fmax(ax, ay), az);
    if (big == 0.0) return 0.0;
    const double r0 = __builtin_amdgcn_rcp(big);
    const double e0 = __builtin_fma(-big, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-big, r1, 1.0);
    const double r2 = __builtin_fma(r1, e1, r1);
    const double mx = ax * r2, my = ay * r2, mz = az * r2;
    const double nx = __builtin_fma(__builtin_fma(-big, mx, ax), r2, mx);
    const double ny = __builtin_fma(__builtin_fma(-big, my, ay), r2, my);
    const double nz = __builtin_fma(__builtin_fma(-big, mz, az), r2, mz);
    double sum = nx * nx;
    const double sy = ny * ny;
    double next = sum + sy;
    const double comp = (next - sum) - sy;
    sum = next;
    const double sz = nz * nz - comp;
    sum = sum + sz;
    return sqrt_unit_range(sum) * big;  // the largest scaled square is exactly 1: sum in [1, 3]
}

// a / b as the compiler's IEEE expansion computes it when v_div_scale has nothing to scale (reciprocal estimate, two
// Newton steps, quotient, one correction) without the scale / fmas / fixup bracket: 8 instead of ~30 instructions.
// Used only where the operand ranges are known and the equality was checked EXHAUSTIVELY on the device
// (rm_selftest_recip): 1.0 / d for every finite non-zero binary32 d, and x / W for all integers 0 <= x < 2^16,
// 1 <= W < 2^16.
__device__ __forceinline__ double div_in_range(double a, double b) {
    const double r0 = __builtin_amdgcn_rcp(b);
    const double e0 = __builtin_fma(-b, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-b, r1, 1.0);
    const double r2 = __builtin_fma(r1, e1, r1);
    const double q = a * r2;
    return __builtin_fma(__builtin_fma(-b, q, a), r2, q);
}

// gl-matrix vec3.length / vec3.distance (sphere.ts:13, box.ts:26, mandelbulb.ts:46).  3.0 - 3.4.3 compute
// Math.hypot(x, y, z); the pinned 3.4.4 is not available offline and a later patch release may use
// Math.sqrt(x*x + y*y + z*z), so the formula is ONE switchable function: the kernels are
// compiled twice, and option `length` = 1 selects the objects built with -DRM_LENGTH_SQRT.  vec3_length is the
// form with the shared reciprocal (bit-identical to hypot3), vec3_length_plain the one with the compiler's divisions.
#ifdef RM_LENGTH_SQRT
__device__ __forceinline__ double vec3_length(float lx, float ly, float lz) {
    const double x = lx, y = ly, z = lz;
    return __builtin_sqrt(x * x + y * y + z * z);
}
__device__ __forceinline__ double vec3_length_plain(float lx, float ly, float lz) { return vec3_length(lx, ly, lz); }
#else
__device__ __forceinline__ double vec3_length(float lx, float ly, float lz) { return hypot3_shared_rcp(lx, ly, lz); }
__device__ __forceinline__ double vec3_length_plain(float lx, float ly, float lz) { return hypot3(lx, ly, lz); }
#endif

struct Vec3f {
    float x, y, z;
};

struct Ray {
    Vec3f o, d;
    double od[3];  // origin again as doubles, supplied by the host: wave-uniform values stay in
                   // SGPRs instead of being widened per lane (v_cvt_f64_f32) and hoisted into VGPRs
};

// Primitive.sdf (primitive.ts:33-39) for a translation-only world->local transform:
// transformMat4 reduces to f32(f64(p - c)), which equals the binary32 subtraction (double
// rounding is innocuous for +,- when the wide format has >= 2p+2 bits), then
// Sphere.localSdf (sphere.ts:12-14).
__device__ __forceinline__ double sphere_sdf(const RmSphere &s, double radius, const Vec3f &p) {
    return vec3_length_plain(p.x - s.cx, p.y - s.cy, p.z - s.cz) - radius;
}

__device__ __forceinline__ double sphere_sdf_fast(const RmSphere &s, double radius, const Vec3f &p) {
    return vec3_length(p.x - s.cx, p.y - s.cy, p.z - s.cz) - radius;
}

// A binary32 value >= v for the filter's running upper bound: round to nearest, then add more than the
// rounding error (2^-24 |v|, or half the smallest denormal).  One ulp looser than a directed rounding, three
// instructions instead of the fourteen of __double2float_ru's emulation.
__device__ __forceinline__ float f32_upper_bound(double v) {
    const float f = static_cast<float>(v);
    return f + (__builtin_fabsf(f) * 1.2e-7f + 1e-30f);
}

// Conservative binary32 estimate of the same distance: |estimate - exact| <= err.
// Error budget: 2^-24 (4.5 len + r + |a|) <= 3.3e-7 (len + r) with the raw v_sqrt_f32 (1 ulp; the IEEE
// expansion of sqrtf costs 15 more instructions and buys nothing here); err is ~12x that, which also
// absorbs the roundings of the +/- err comparisons themselves.
__device__ __forceinline__ float sphere_sdf_estimate(const RmSphere &s, const Vec3f &p, float &err) {
    const float dx = p.x - s.cx, dy = p.y - s.cy, dz = p.z - s.cz;
    const float len = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
    err = (len + __builtin_fabsf(s.rf) + 1.0f) * 4e-6f;  // |r|: a negative radius is legal input (sphere.ts:12-14 just subtracts it)
    return len - s.rf;
}

// Math.min(candidate, closest) -- operands are never NaN here (inputs validated finite)
// (v_min_f64; it differs from the select only for NaN or for the sign of a zero, neither of
// which can reach an output: DESIGN.md section 2)
__device__ __forceinline__ double min_dist(double candidate, double closest) {
    return __builtin_fmin(candidate, closest);
}

// min(closest, min over ids[0..n) -- or 0..n-1 when ids is null -- of Sphere.sdf), the loop of
// scene.ts:155-158,175-178,183-187.  With `filter` a sphere is evaluated exactly only when its
// conservative binary32 lower bound does not exceed an upper bound of the best value so far;
// a skipped sphere has exact > best >= result, and min() does not depend on order, so the
// value is bit-identical to the plain loop.  FAST selects the shared-reciprocal hypot.
template <bool FAST, typename IdT = int32_t>
__device__ __forceinline__ double prims_min(const RmSphere *spheres, const double *radii, const IdT *ids, int n,
                                            const Vec3f &p, double closest, bool filter) {
    if (!filter || n < 2) {
        for (int k = 0; k < n; ++k) {
            const int id = ids ? ids[k] : k;
            const double e = FAST ? sphere_sdf_fast(spheres[id], radii[id], p) : sphere_sdf(spheres[id], radii[id], p);
            closest = min_dist(e, closest);
        }
        return closest;
    }
    float ub = f32_upper_bound(closest);
    for (int k = 0; k < n; ++k) {
        const int id = ids ? ids[k] : k;
        const RmSphere s = spheres[id];
        float err;
        const float a = sphere_sdf_estimate(s, p, err);
        if (a - err <= ub) {
            const double e = FAST ? sphere_sdf_fast(s, radii[id], p) : sphere_sdf(s, radii[id], p);
            if (e < closest) {
                closest = e;
                ub = f32_upper_bound(e);
            }
        }
    }
    return closest;
}

// The same minimum with the exact evaluation taken out of the scan.  Lanes of a wave work on different lists, so a
// loop that evaluates a sphere exactly as soon as its bound passes executes the FP64 body once per list position
// at which SOME lane passes.  Here the scan keeps per lane the sphere with the smallest upper bound (k1, hi1, lb1)
// and the smallest lower bound among the others (lb2); afterwards every lane evaluates its k1 in one common
// instruction stream.  If lb2 exceeds that exact value no other sphere can be closer (exact_j >= lb_j >= lb2);
// otherwise (near ties) the list is rescanned with the ordinary filter.  ids == nullptr: spheres base .. base+n-1.
template <typename IdT>
__device__ __forceinline__ double prims_min_best(const RmSphere *spheres, const double *radii, const IdT *ids, int n, int base,
                                                 const Vec3f &p, double closest) {
    if (n <= 0) return closest;
    const float inf = __builtin_inff();
    int k1 = 0;
    float hi1 = inf, lb1 = inf, lb2 = inf;
    for (int k = 0; k < n; ++k) {
        const int id = ids ? static_cast<int>(ids[k]) : base + k;
        float err;
        const float a = sphere_sdf_estimate(spheres[id], p, err);
        const float lb = a - err, hi = a + err;
        const bool better = hi < hi1;
        lb2 = __builtin_fminf(lb2, better ? lb1 : lb);
        k1 = better ? id : k1;
        lb1 = better ? lb : lb1;
        hi1 = better ? hi : hi1;
    }
    {
        const double e = sphere_sdf_fast(spheres[k1], radii[k1], p);
        closest = e < closest ? e : closest;
    }
    float ub = f32_upper_bound(closest);
    if (lb2 <= ub) {
        for (int k = 0; k < n; ++k) {
            const int id = ids ? static_cast<int>(ids[k]) : base + k;
            const RmSphere s = spheres[id];
            float err;
            const float a = sphere_sdf_estimate(s, p, err);
            if (a - err <= ub) {
                const double e = sphere_sdf_fast(s, radii[id], p);
                if (e < closest) {
                    closest = e;
                    ub = f32_upper_bound(e);
                }
            }
        }
    }
    return closest;
}

// Primitive.sdf (primitive.ts:33-39) with the full vec3.transformMat4 (w = w || 1.0), then
// Sphere / Box / Torus .localSdf (sphere.ts:12-14, box.ts:13-30, torus.ts:14-25).
__device__ __forceinline__ double prim_sdf_general(const RmPrim &q, const Vec3f &p) {
    const double x = p.x, y = p.y, z = p.z;
    float lx, ly, lz;
    if (q.type & 0x200) {  // pure translation: f32(x + t), see sphere_sdf
        lx = p.x + q.m[12];
        ly = p.y + q.m[13];
        lz = p.z + q.m[14];
    } else if (q.type & 0x100) {  // bottom row (0,0,0,1): w = 1 exactly for finite points, x / 1.0 = x
        lx = to_f32(q.m[0] * x + q.m[4] * y + q.m[8] * z + q.m[12]);
        ly = to_f32(q.m[1] * x + q.m[5] * y + q.m[9] * z + q.m[13]);
        lz = to_f32(q.m[2] * x + q.m[6] * y + q.m[10] * z + q.m[14]);
    } else {
        double w = q.m[3] * x + q.m[7] * y + q.m[11] * z + q.m[15];
        if (!(w != 0.0)) w = 1.0;  // 0, -0 and NaN are falsy
        lx = to_f32((q.m[0] * x + q.m[4] * y + q.m[8] * z + q.m[12]) / w);
        ly = to_f32((q.m[1] * x + q.m[5] * y + q.m[9] * z + q.m[13]) / w);
        lz = to_f32((q.m[2] * x + q.m[6] * y + q.m[10] * z + q.m[14]) / w);
    }
    const int type = q.type & 0xFF;
    if (type == 1) {
        const float e0 = to_f32(__builtin_fabs(static_cast<double>(lx)) - static_cast<double>(q.half[0]));
        const float e1 = to_f32(__builtin_fabs(static_cast<double>(ly)) - static_cast<double>(q.half[1]));
        const float e2 = to_f32(__builtin_fabs(static_cast<double>(lz)) - static_cast<double>(q.half[2]));
        const float o0 = e0 > 0.f ? e0 : 0.f, o1 = e1 > 0.f ? e1 : 0.f, o2 = e2 > 0.f ? e2 : 0.f;  // Math.max(q, 0)
        const double outside = vec3_length(o0, o1, o2);
        const float big = e0 > (e1 > e2 ? e1 : e2) ? e0 : (e1 > e2 ? e1 : e2);  // Math.max(q0, Math.max(q1, q2))
        const double inside = big < 0.f ? static_cast<double>(big) : 0.0;       // Math.min(., 0)
        return outside + inside;
    }
    if (type == 2) {
        const double dx = lx, dy = ly, dz = lz;
        const double qx = __builtin_sqrt(dx * dx + dz * dz) - q.a;
        return __builtin_sqrt(qx * qx + dy * dy) - q.b;
    }
    return vec3_length(lx, ly, lz) - q.a;
}

// BoundingBox.contains (boundingBox.ts:15-21)
__device__ __forceinline__ bool box_contains(const float lo[3], const float hi[3], const Vec3f &p) {
    return p.x >= lo[0] && p.x <= hi[0] && p.y >= lo[1] && p.y <= hi[1] && p.z >= lo[2] && p.z <= hi[2];
}

// BoundingBox.intersectRay (boundingBox.ts:69-105); false = null
__device__ __forceinline__ bool slab(const float lo[3], const float hi[3], const Ray &r, double &tEnter,
                                     double &tExit) {
    double tMin = -__builtin_inf(), tMax = __builtin_inf();
    const float o[3] = {r.o.x, r.o.y, r.o.z};
    const float d[3] = {r.d.x, r.d.y, r.d.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (__builtin_fabs(static_cast<double>(d[a])) < 1e-10) {
            if (o[a] < lo[a] || o[a] > hi[a]) return false;
        } else {
            const double inv = 1.0 / static_cast<double>(d[a]);
            double t0 = (static_cast<double>(lo[a]) - r.od[a]) * inv;
            double t1 = (static_cast<double>(hi[a]) - r.od[a]) * inv;
            if (t0 > t1) {
                const double t = t0;
                t0 = t1;
                t1 = t;
            }
            tMin = tMin > t0 ? tMin : t0;  // Math.max; no NaN possible (|d| >= 1e-10)
            tMax = tMax < t1 ? tMax : t1;  // Math.min
            if (tMin > tMax) return false;
        }
    }
    tEnter = tMin;
    tExit = tMax;
    return true;
}

// Same test with 1/direction[i] hoisted out (the reference recomputes the identical
// quotient for every box): inv[a] = 1.0 / d[a], par[a] = |d[a]| < 1e-10.
struct RayInv {
    double inv[3];
    bool par[3];
    bool any_par;  // some axis has |d| < 1e-10 (only the rays through the image's centre column / row)
};

__device__ __forceinline__ RayInv make_ray_inv(const Ray &r) {
    RayInv ri;
    const float d[3] = {r.d.x, r.d.y, r.d.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        ri.par[a] = __builtin_fabs(static_cast<double>(d[a])) < 1e-10;
        ri.inv[a] = ri.par[a] ? 0.0 : div_in_range(1.0, static_cast<double>(d[a]));  // d finite, |d| >= 1e-10
    }
    ri.any_par = ri.par[0] || ri.par[1] || ri.par[2];
    return ri;
}

__device__ __forceinline__ bool slab_inv(const float lo[3], const float hi[3], const Ray &r, const RayInv &ri,
                                         double &tEnter, double &tExit) {
    if (__ballot(ri.any_par) == 0) {
        // No active lane has a parallel axis (the normal case): straight-line code.  tMin only grows and tMax
        // only shrinks, so the per-axis `if (tMin > tMax) return null` of the reference equals one test at the end.
        const double ax = (static_cast<double>(lo[0]) - r.od[0]) * ri.inv[0], bx = (static_cast<double>(hi[0]) - r.od[0]) * ri.inv[0];
        const double ay = (static_cast<double>(lo[1]) - r.od[1]) * ri.inv[1], by = (static_cast<double>(hi[1]) - r.od[1]) * ri.inv[1];
        const double az = (static_cast<double>(lo[2]) - r.od[2]) * ri.inv[2], bz = (static_cast<double>(hi[2]) - r.od[2]) * ri.inv[2];
        const double tMin = __builtin_fmax(__builtin_fmax(__builtin_fmin(ax, bx), __builtin_fmin(ay, by)), __builtin_fmin(az, bz));
        const double tMax = __builtin_fmin(__builtin_fmin(__builtin_fmax(ax, bx), __builtin_fmax(ay, by)), __builtin_fmax(az, bz));
        tEnter = tMin;
        tExit = tMax;
        return !(tMin > tMax);
    }
    double tMin = -__builtin_inf(), tMax = __builtin_inf();
    const float o[3] = {r.o.x, r.o.y, r.o.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (ri.par[a]) {
            if (o[a] < lo[a] || o[a] > hi[a]) return false;
        } else {
            const double ta = (static_cast<double>(lo[a]) - r.od[a]) * ri.inv[a];
            const double tb = (static_cast<double>(hi[a]) - r.od[a]) * ri.inv[a];
            const double t0 = __builtin_fmin(ta, tb), t1 = __builtin_fmax(ta, tb);  // if (t0 > t1) swap
            tMin = __builtin_fmax(tMin, t0);  // Math.max / Math.min: operands are never NaN here
            tMax = __builtin_fmin(tMax, t1);
            if (tMin > tMax) return false;
        }
    }
    tEnter = tMin;
    tExit = tMax;
    return true;
}

// The same test on a node box stored relative to the ray origin: rel = {lo - o, hi - o} as doubles, the very
// differences `(this.min[i] - ray.origin[i])` the reference forms (boundingBox.ts:84-85), computed once per frame and
// node when the persistent workgroup stages the scene (the origin is the same for every ray of a frame).  Saves the
// six conversions and six subtractions of every test.  Parallel axes: o < lo <=> lo - o > 0 and o > hi <=> hi - o < 0
// exactly (the difference of two distinct binary32 values is a non-zero double).
__device__ __forceinline__ bool slab_rel(const double *rel, const RayInv &ri, double &tEnter, double &tExit) {
    if (__ballot(ri.any_par) == 0) {
        const double ax = rel[0] * ri.inv[0], bx = rel[3] * ri.inv[0];
        const double ay = rel[1] * ri.inv[1], by = rel[4] * ri.inv[1];
        const double az = rel[2] * ri.inv[2], bz = rel[5] * ri.inv[2];
        const double tMin = __builtin_fmax(__builtin_fmax(__builtin_fmin(ax, bx), __builtin_fmin(ay, by)), __builtin_fmin(az, bz));
        const double tMax = __builtin_fmin(__builtin_fmin(__builtin_fmax(ax, bx), __builtin_fmax(ay, by)), __builtin_fmax(az, bz));
        tEnter = tMin;
        tExit = tMax;
        return !(tMin > tMax);
    }
    double tMin = -__builtin_inf(), tMax = __builtin_inf();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (ri.par[a]) {
            if (rel[a] > 0.0 || rel[a + 3] < 0.0) return false;
        } else {
            const double ta = rel[a] * ri.inv[a];
            const double tb = rel[a + 3] * ri.inv[a];
            const double t0 = __builtin_fmin(ta, tb), t1 = __builtin_fmax(ta, tb);
            tMin = __builtin_fmax(tMin, t0);
            tMax = __builtin_fmin(tMax, t1);
            if (tMin > tMax) return false;
        }
    }
    tEnter = tMin;
    tExit = tMax;
    return true;
}

struct Interval {
    double tEnter, tExit;
    int ord;  // node index == position in the traversal order of bvh.ts:136-173
};

// Octree.marchRay (octree.ts:252-278) for the node that contains the current point.
// intersectRayBox (octree.ts:195-220) has no zero-direction guard and stores t0/t1 in a
// Float32Array; 0 * Infinity = NaN then flows through Math.max/min (NaN-propagating) and
// every comparison with NaN is false.
// 1 / direction[i] as intersectRayBox computes it for every call (octree.ts:200; no zero guard: +-Infinity for a
// zero component): the same three quotients for every node of a ray, so the marchers compute them once.
struct OctInv {
    double inv[3];
};
__device__ __forceinline__ OctInv make_oct_inv(const Ray &r) {
    OctInv o;
    o.inv[0] = 1.0 / static_cast<double>(r.d.x);
    o.inv[1] = 1.0 / static_cast<double>(r.d.y);
    o.inv[2] = 1.0 / static_cast<double>(r.d.z);
    return o;
}

__device__ __forceinline__ double oct_skip(const RmOctNode &nd, const Ray &r, double t, const OctInv &oi) {
    if (!nd.is_empty) return 0.0;
    float tn[3], tf[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double inv = oi.inv[a];
        double t0 = (static_cast<double>(nd.lo[a]) - r.od[a]) * inv;
        double t1 = (static_cast<double>(nd.hi[a]) - r.od[a]) * inv;
        if (inv < 0.0) {
            const double s = t0;
            t0 = t1;
            t1 = s;
        }
        tn[a] = to_f32(t0);
        tf[a] = to_f32(t1);
    }
    // JS: tEnter = Math.max(tMin[0..2]), tExit = Math.min(tMax[0..2]) are NaN when any
    // operand is; `tEnter > tExit || tExit < 0` is false whenever a NaN takes part, and only
    // tExit is used afterwards, so a NaN tEnter is ignored while a NaN tExit ends in 0.
    if (tf[0] != tf[0] || tf[1] != tf[1] || tf[2] != tf[2]) return 0.0;
    float x = tf[0] < tf[1] ? tf[0] : tf[1];
    x = tf[2] < x ? tf[2] : x;
    const double tExit = x;
    if (!(tn[0] != tn[0] || tn[1] != tn[1] || tn[2] != tn[2])) {
        float e = tn[0] > tn[1] ? tn[0] : tn[1];
        e = tn[2] > e ? tn[2] : e;
        if (static_cast<double>(e) > tExit) return 0.0;
    }
    if (tExit < 0.0) return 0.0;
    double toExit = tExit - t;
    toExit = toExit > 0.0 ? toExit : 0.0;  // Math.max(0, .)
    const double cap = nd.min_distance * 0.99;
    double step = toExit < cap ? toExit : cap;
    step = step > 0.0 ? step : 0.0;
    return step > 0.0 ? step + 0.001 : 0.0;
}
__device__ __forceinline__ double oct_skip(const RmOctNode &nd, const Ray &r, double t) {
    return oct_skip(nd, r, t, make_oct_inv(r));
}

// ------------------------------------------------------------------ ray generation / stores

// packed (launch-local) row -> frame row y.  Without striping the tile is the contiguous range
// [y_start, y_end) of the Job (raymarchWorker.ts:14-15); with striping (multi-GPU sharding) the
// launch owns every n_parts-th stripe of stripe_rows rows, or the stripes its list names (stripe_ids).
__device__ __forceinline__ int row_to_y(const RmRenderParams &P, int r) {
    if (P.stripe_rows <= 0) return P.y_start + r;
    const int s = r / P.stripe_rows;
    int stripe;
    if (P.stripe_ids) {
        // rows past the launch's last row are asked for too (the bundle of a batch that hangs over the end of the tile
        // rows): never read past the list, and keep the mapping increasing there like the round-robin rule does
        const int last = (P.local_rows - 1) / P.stripe_rows;
        const int k = s < last ? s : last;
        stripe = P.stripe_ids[k] + (s - k);
    } else {
        stripe = s * P.n_parts + P.part;
    }
    return P.y_start + stripe * P.stripe_rows + (r - s * P.stripe_rows);
}

// raymarcher.ts:73,83-88: u, v from full-frame W, H; fromValues, transformMat3, normalize
__device__ __forceinline__ Ray make_ray(const RmRenderParams &P, int x, int y) {
    double qy, qx;
    if (P.width < 65536 && P.height < 65536) {  // wave-uniform: the quotients are in the exhaustively checked range
        qy = div_in_range(static_cast<double>(y), static_cast<double>(P.height));
        qx = div_in_range(static_cast<double>(x), static_cast<double>(P.width));
    } else {
        qy = static_cast<double>(y) / static_cast<double>(P.height);
        qx = static_cast<double>(x) / static_cast<double>(P.width);
    }
    const double v = (qy - 0.5) * 2.0;
    const double u = (qx - 0.5) * 2.0;
    const double ax = to_f32(u), ay = to_f32(v), az = -1.0;
    const float dx = to_f32(ax * P.rot_d[0] + ay * P.rot_d[3] + az * P.rot_d[6]);
    const float dy = to_f32(ax * P.rot_d[1] + ay * P.rot_d[4] + az * P.rot_d[7]);
    const float dz = to_f32(ax * P.rot_d[2] + ay * P.rot_d[5] + az * P.rot_d[8]);
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
    return ray;
}

// vec3.scaleAndAdd(out, origin, dir, t) (sphereTracer.ts:44-45, raymarcher.ts:94-95)
__device__ __forceinline__ Vec3f point_at(const Ray &r, double t) {
    Vec3f p;
    p.x = to_f32(r.od[0] + static_cast<double>(r.d.x) * t);
    p.y = to_f32(r.od[1] + static_cast<double>(r.d.y) * t);
    p.z = to_f32(r.od[2] + static_cast<double>(r.d.z) * t);
    return p;
}

// PhongModel.shade for one pixel (phongModel.ts:33-72).  A real function, not inlined: the device library's pow keeps
// ~18 double constants in VGPRs, and inlined into the render kernels they were hoisted out of the wave loop and held
// for the whole kernel (36 of its 128 VGPRs), whichever shader the frame used.
__device__ __attribute__((noinline)) static uchar4 shade_phong(uint8_t depth, uint8_t n0, uint8_t n1, uint8_t n2, double l0,
                                                                double l1, double l2) {
    if (depth >= 255) return make_uchar4(10, 10, 20, 255);
    float nx = to_f32(static_cast<double>(n0) / 127.5 - 1.0);
    float ny = to_f32(static_cast<double>(n1) / 127.5 - 1.0);
    float nz = to_f32(static_cast<double>(n2) / 127.5 - 1.0);
    double len = static_cast<double>(nx) * nx + static_cast<double>(ny) * ny + static_cast<double>(nz) * nz;
    if (len > 0) len = 1 / __builtin_sqrt(len);
    nx = to_f32(nx * len);
    ny = to_f32(ny * len);
    nz = to_f32(nz * len);
    const double ndl = static_cast<double>(nx) * l0 + static_cast<double>(ny) * l1 +
                       static_cast<double>(nz) * l2;
    const double diffuse = ndl > 0.0 ? ndl : 0.0;
    const double k2 = 2 * ndl;
    float rx = to_f32(nx * k2), ry = to_f32(ny * k2), rz = to_f32(nz * k2);
    rx = to_f32(static_cast<double>(rx) - l0);
    ry = to_f32(static_cast<double>(ry) - l1);
    rz = to_f32(static_cast<double>(rz) - l2);
    double rl = static_cast<double>(rx) * rx + static_cast<double>(ry) * ry + static_cast<double>(rz) * rz;
    if (rl > 0) rl = 1 / __builtin_sqrt(rl);
    rz = to_f32(rz * rl);
    const double vdr = static_cast<double>(rz);  // dot((0,0,1), reflect)
    const double base = vdr > 0.0 ? vdr : 0.0;
    // Math.pow is not correctly rounded on either side; the byte is within 1 LSB
    const double spec = 0.5 * pow(base, 32.0);
    double inten = 0.1 + diffuse + spec;
    inten = inten < 1.0 ? inten : 1.0;
    const double color = 255 * inten * (1 - static_cast<double>(depth) / 255);
    const uint8_t c = u8clamp(color);
    return make_uchar4(c, c, c, 255);
}

// ShadingModel.shade for one pixel.  Heatmaps (SDFHeatmap.ts:24-29, IterationHeatmap.ts:24-29)
// and Normal (normalModel.ts:21-24) are pure integer; Phong follows phongModel.ts:33-72 in
// double with f32 stores.
__device__ __forceinline__ uchar4 shade_pixel(int shader, uint8_t depth, uint8_t n0, uint8_t n1, uint8_t n2,
                                              uint16_t sdf, uint16_t iters, const double light[3]) {
    if (shader == 2 || shader == 3) {
        const uint32_t c = shader == 2 ? sdf : iters;
        const uint32_t k = (c * 5u) & 255u;                              // counter * 5 % 256
        const uint32_t r = 2u * k < 255u ? 2u * k : 255u;                 // Math.min(2k, 255)
        const uint32_t g = 512u - 2u * k < 255u ? 512u - 2u * k : 255u;  // Math.min(-2k + 512, 255)
        return make_uchar4(static_cast<uint8_t>(r), static_cast<uint8_t>(g), 0, 255);
    }
    if (shader == 1) return shade_phong(depth, n0, n1, n2, light[0], light[1], light[2]);
    return make_uchar4(n0, n1, n2, 255);
}

// raymarcher.ts:103-106 + fused shade: the five stores of one pixel
__device__ __forceinline__ void store_pixel(const RmRenderParams &P, size_t idx, double depth, float nx, float ny,
                                            float nz, uint32_t count, uint32_t iters) {
    const uint8_t n0 = u8clamp((static_cast<double>(nx) + 1) * 0.5 * 255);
    const uint8_t n1 = u8clamp((static_cast<double>(ny) + 1) * 0.5 * 255);
    const uint8_t n2 = u8clamp((static_cast<double>(nz) + 1) * 0.5 * 255);
    const uint8_t db = u8clamp(depth);
    const uint16_t c16 = static_cast<uint16_t>(count & 0xFFFFu);  // Uint16Array += wraps
    const uint16_t i16 = static_cast<uint16_t>(iters & 0xFFFFu);
    if (P.depth) P.depth[idx] = db;
    if (P.normal) {
        P.normal[3 * idx] = n0;
        P.normal[3 * idx + 1] = n1;
        P.normal[3 * idx + 2] = n2;
    }
    if (P.sdf) P.sdf[idx] = c16;
    if (P.iters) P.iters[idx] = i16;
    if (P.rgba) reinterpret_cast<uchar4 *>(P.rgba)[idx] = shade_pixel(P.shader, db, n0, n1, n2, c16, i16, P.light_d);
}

// vec3.normalize (raymarcher.ts:133)
__device__ __forceinline__ void normalize3(float &nx, float &ny, float &nz) {
    double len = static_cast<double>(nx) * nx + static_cast<double>(ny) * ny + static_cast<double>(nz) * nz;
    if (len > 0) len = 1 / __builtin_sqrt(len);
    nx = to_f32(nx * len);
    ny = to_f32(ny * len);
    nz = to_f32(nz * len);
}

}  // namespace rmd
