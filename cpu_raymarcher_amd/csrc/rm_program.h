// rm_program.h -- device interpreter for SDF expression programs (RmInstr, rm_types.h):
// Primitive.sdf (primitive.ts:33-39) of Round / SmoothUnion / SmoothSubtraction / Twist / Repetition /
// AnimatedTranslate (primitive_operations/*.ts) over Sphere / Box / Torus / Mandelbulb leaves.
// Every wave executes the instruction stream of one object uniformly (the caller, list_min in rm_kernels.hip, lets lanes
// with different objects take turns; the program counter and the slot indices are wave-uniform and the stream is read
// through the scalar cache); only the Mandelbulb's escape loop diverges per lane.
#pragma once
#include "rm_device.h"
#include "rm_jsmath.h"

// Position slots and pending values of the interpreter live in LDS (dynamic shared memory of the launch,
// sized by the host from the scene's deepest program): slot s, component c of thread t is word
// (s*3 + c) * blockDim + t, so a wave's access is one conflict-free row.  Private arrays indexed by the
// (uniform, but run-time) slot number had gone to scratch: three vector-memory round trips per instruction.
extern __shared__ __align__(16) unsigned char rm_prog_smem[];

namespace rmd {

// The instruction stream is the same for every lane: read it through the scalar cache.  A pointer handed to an out-of-line
// function arrives in VGPRs as a generic pointer, and every field read became a flat_load with a full wait behind it
// (three or four dependent round trips per interpreted instruction); made wave-uniform and typed as constant address
// space, the same reads are s_load_dwordx4/x8/x16 into SGPRs that the VALU instructions use directly.
typedef const RmInstr __attribute__((address_space(4))) *RmInstrConstPtr;
__device__ __forceinline__ RmInstrConstPtr uniform_program(const RmInstr *prog) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(prog);
    const unsigned int lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned int>(a)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned int>(a >> 32));
    return reinterpret_cast<RmInstrConstPtr>((static_cast<unsigned long long>(hi) << 32) | lo);
}

// What every instruction needs, requested in ONE batch at the top of the interpreter loop: the last 64 bytes of the record
// (six parameters, op, src, dst, flags: one s_load_dwordx16) and the translation columns of T and Tinv (all a pure
// translation needs).  Read field by field the record cost three or four dependent scalar-cache round trips.
typedef int rm_i32x16 __attribute__((ext_vector_type(16)));
typedef float rm_f32x4 __attribute__((ext_vector_type(4)));
struct InstrHead {
    double p[6];
    int op, src, dst, flags;
    float t[3], ti[3];  // T[12..14], Tinv[12..14]
};
__device__ __forceinline__ InstrHead load_head(RmInstrConstPtr I) {
    const rm_i32x16 w = *reinterpret_cast<const rm_i32x16 __attribute__((address_space(4))) *>(&I->p[0]);
    const rm_f32x4 t = *reinterpret_cast<const rm_f32x4 __attribute__((address_space(4))) *>(&I->T[12]);
    const rm_f32x4 ti = *reinterpret_cast<const rm_f32x4 __attribute__((address_space(4))) *>(&I->Tinv[12]);
    InstrHead h;
#pragma unroll
    for (int k = 0; k < 6; ++k)
        h.p[k] = __builtin_bit_cast(double, (static_cast<unsigned long long>(static_cast<unsigned int>(w[2 * k + 1])) << 32) |
                                                static_cast<unsigned int>(w[2 * k]));
    h.op = w[12];
    h.src = w[13];
    h.dst = w[14];
    h.flags = w[15];
    h.t[0] = t[0], h.t[1] = t[1], h.t[2] = t[2];
    h.ti[0] = ti[0], h.ti[1] = ti[1], h.ti[2] = ti[2];
    asm volatile("" ::"s"(h.op), "s"(h.t[0]), "s"(h.ti[0]));  // keep the three loads together here (not sunk into the branches that use them)
    return h;
}
// Math.min / Math.max: NaN if either argument is NaN (the Mandelbulb can produce NaN at its pole)
__device__ __forceinline__ double js_min_nan(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a < b ? a : b); }
__device__ __forceinline__ double js_max_nan(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a > b ? a : b); }

// vec3.transformMat4 into a Float32Array (w = w || 1.0).  `affine`: the host saw the bottom row
// (m3, m7, m11, m15) = (0, 0, 0, 1) exactly; then w = 0*x + 0*y + 0*z + 1 is 1 for every finite point
// (and NaN -> `|| 1.0` -> 1 for a non-finite one), and a division by 1.0 returns its numerator:
// the three IEEE divisions are skipped with identical results.
// `flags` bit 2 (of the two bits handed in): additionally the upper 3x3 block is exactly the identity
// (translations, and the identity itself for the smooth unions): 1*x + 0*y + 0*z + t = x + t for every
// finite point, signed zeros included (the trailing + t absorbs them), and f32(f64(x) + f64(t)) equals the
// binary32 sum (double rounding is innocuous for + when the wide format has >= 2p+2 bits): one v_add_f32.
template <typename MatPtr>
__device__ __forceinline__ void transform_mat4(MatPtr m, int flags, float fx, float fy, float fz, float &ox, float &oy, float &oz) {
    if (flags & 4) {
        ox = fx + m[12];
        oy = fy + m[13];
        oz = fz + m[14];
        return;
    }
    const double x = fx, y = fy, z = fz;
    if (flags & 1) {
        ox = to_f32(m[0] * x + m[4] * y + m[8] * z + m[12]);
        oy = to_f32(m[1] * x + m[5] * y + m[9] * z + m[13]);
        oz = to_f32(m[2] * x + m[6] * y + m[10] * z + m[14]);
        return;
    }
    double w = m[3] * x + m[7] * y + m[11] * z + m[15];
    if (!(w != 0.0)) w = 1.0;
    ox = to_f32((m[0] * x + m[4] * y + m[8] * z + m[12]) / w);
    oy = to_f32((m[1] * x + m[5] * y + m[9] * z + m[13]) / w);
    oz = to_f32((m[2] * x + m[6] * y + m[10] * z + m[14]) / w);
}

// transform_mat4 with the translation column already at hand
template <typename MatPtr>
__device__ __forceinline__ void transform_head(MatPtr m, const float *tcol, int flags, float fx, float fy, float fz, float &ox, float &oy, float &oz) {
    if (flags & 4) {
        ox = fx + tcol[0];
        oy = fy + tcol[1];
        oz = fz + tcol[2];
        return;
    }
    transform_mat4(m, flags, fx, fy, fz, ox, oy, oz);
}

// Out-of-line copies for the Mandelbulb's escape loop: inlined, the five fdlibm bodies push the kernel to 226
// VGPRs (2 waves/SIMD); as calls the register need is the largest callee's, not their sum.
// In a scene's run-time specialised kernel (RM_RTC) they are inlined: 136 VGPRs, three waves per SIMD, no scratch -- 11.1
// against 12.0 ms on the 1080p Mandelbulb (RM_RTC_DEFINES="RM_MB_CALLS" restores the calls; the compile takes 8 s instead of 1.4).
#if defined(RM_RTC) && !defined(RM_MB_CALLS)
#define RM_MB_CALL __forceinline__
#else
#define RM_MB_CALL __attribute__((noinline))
#endif
// (Two independent evaluations share a call: the fdlibm bodies are long dependent chains, and at three waves per SIMD the
// second chain fills the issue slots the first one leaves empty.)
__device__ RM_MB_CALL void mb_angles(double y, double x, double s, double &theta, double &phi) {
    theta = js_atan2(y, x);
    phi = js_asin(s);
}
__device__ RM_MB_CALL double mb_log(double x) { return js_log(x); }
__device__ RM_MB_CALL void mb_pow_pair(double x, double ya, double yb, double &ra, double &rb) { js_pow_pair(x, ya, yb, ra, rb); }
__device__ RM_MB_CALL void mb_sincos2(double a, double b, double &sa, double &ca, double &sb, double &cb) {
    js_sincos(a, sa, ca);
    js_sincos(b, sb, cb);
}

// mandelbulb.ts:37-78; z is a Float32Array: each component store rounds to binary32
template <typename PrmPtr>
__device__ inline double mandelbulb_sdf(PrmPtr prm, float lx, float ly, float lz, double time) {
    const double power = prm[0], speed = prm[3];
    const int iterations = static_cast<int>(prm[1]);
    const bool animate = prm[2] != 0.0;
    const float p0 = lx, p1 = lz, p2 = ly;  // p.xyz = p.xzy
    float z0 = p0, z1 = p1, z2 = p2;
    double dr = 1.0, r = 0.0;
    for (int i = 0; i < iterations; ++i) {
        r = vec3_length(z0, z1, z2);
        if (r > 2.0) break;
        double theta, phi;
        mb_angles(z1, z0, static_cast<double>(z2) / r, theta, phi);
        if (animate) phi += time * speed;
        double pw_m1, pw;  // Math.pow(r, power - 1), Math.pow(r, power): one log2(r) for both
        mb_pow_pair(r, power - 1.0, power, pw_m1, pw);
        dr = pw_m1 * dr * power + 1.0;
        r = pw;
        theta = theta * power;
        phi = phi * power;
        double sth, cth, sph, cph;  // one argument reduction per angle
        mb_sincos2(theta, phi, sth, cth, sph, cph);
        z0 = to_f32(r * cth * cph + static_cast<double>(p0));
        z1 = to_f32(r * sth * cph + static_cast<double>(p1));
        z2 = to_f32(r * sph + static_cast<double>(p2));
    }
    return 0.5 * mb_log(r) * r / dr;
}

// The arithmetic of every node kind, shared by the interpreter below and by the straight-line code the run-time specialiser
// generates for a scene (rm_rtc.cpp): one statement of each formula, two ways of sequencing them.
__device__ __forceinline__ double leaf_box(float lx, float ly, float lz, double hx, double hy, double hz) {  // box.ts:13-30
    const float e0 = to_f32(__builtin_fabs(static_cast<double>(lx)) - hx);
    const float e1 = to_f32(__builtin_fabs(static_cast<double>(ly)) - hy);
    const float e2 = to_f32(__builtin_fabs(static_cast<double>(lz)) - hz);
    const float o0 = e0 > 0.f ? e0 : 0.f, o1 = e1 > 0.f ? e1 : 0.f, o2 = e2 > 0.f ? e2 : 0.f;
    const float big = e0 > (e1 > e2 ? e1 : e2) ? e0 : (e1 > e2 ? e1 : e2);
    return vec3_length(o0, o1, o2) + (big < 0.f ? static_cast<double>(big) : 0.0);
}
__device__ __forceinline__ double leaf_torus(float lx, float ly, float lz, double major, double minor) {  // torus.ts:14-25
    const double dx = lx, dy = ly, dz = lz;
    const double qx = __builtin_sqrt(dx * dx + dz * dz) - major;
    return __builtin_sqrt(qx * qx + dy * dy) - minor;
}
__device__ __forceinline__ double leaf_sphere(float lx, float ly, float lz, double r) { return vec3_length(lx, ly, lz) - r; }  // sphere.ts:12-14
__device__ __forceinline__ double post_smooth_union(double d1, double d2, double k4) {  // smoothUnion.ts:31-34 (k4 = k * 4.0)
    const double h = js_max_nan(k4 - __builtin_fabs(d1 - d2), 0.0);
    return js_min_nan(d1, d2) - h * h * 0.25 / k4;
}
__device__ __forceinline__ double post_smooth_subtraction(double d1, double d2, double k4) {  // smoothSubstraction.ts:30-33
    const double h = js_max_nan(k4 - __builtin_fabs(d1 + d2), 0.0);
    return js_max_nan(d1, -d2) + h * h * 0.25 / k4;
}
// animatedTranslate.ts:34-49: local - direction * (sin(time*speed)*amplitude)
__device__ __forceinline__ void pre_animated_translate(float lx, float ly, float lz, double d0, double d1, double d2, double amplitude, double speed,
                                                       double time, float &wx, float &wy, float &wz) {
    const double offset = js_sin(time * speed) * amplitude;
    wx = to_f32(static_cast<double>(lx) - static_cast<double>(to_f32(d0 * offset)));
    wy = to_f32(static_cast<double>(ly) - static_cast<double>(to_f32(d1 * offset)));
    wz = to_f32(static_cast<double>(lz) - static_cast<double>(to_f32(d2 * offset)));
}
__device__ __forceinline__ void pre_twist(double k, float &wx, float wy, float &wz) {  // twist.ts:21-33
    const double a = k * static_cast<double>(wy);
    double c, s;
    js_sincos(a, s, c);
    const float tx = to_f32(c * static_cast<double>(wx) - s * static_cast<double>(wz));
    const float tz = to_f32(s * static_cast<double>(wx) + c * static_cast<double>(wz));
    wx = tx;
    wz = tz;
}
__device__ __forceinline__ void pre_repetition(double s0, double s1, double s2, float &wx, float &wy, float &wz) {  // repetition.ts:20-24
    wx = to_f32(static_cast<double>(wx) - s0 * js_round(static_cast<double>(wx) / s0));
    wy = to_f32(static_cast<double>(wy) - s1 * js_round(static_cast<double>(wy) / s1));
    wz = to_f32(static_cast<double>(wz) - s2 * js_round(static_cast<double>(wz) / s2));
}

// One scene object: returns Primitive.sdf(p) of the root node.  MB = the scene has a Mandelbulb leaf: its fdlibm
// code doubles the register need (200 against 106 VGPRs), so scenes without one get an instantiation that runs
// four waves per SIMD instead of two.
template <bool MB>
__device__ __attribute__((noinline)) double program_sdf(const RmInstr *prog_, int first_, int count_, const Vec3f &p, double time,
                                                        int n_slots_) {
    const RmInstrConstPtr prog = uniform_program(prog_);
    const int first = __builtin_amdgcn_readfirstlane(first_), count = __builtin_amdgcn_readfirstlane(count_),
              n_slots = __builtin_amdgcn_readfirstlane(n_slots_);
    const int nt = blockDim.x, tid = threadIdx.x;
    float *pos = reinterpret_cast<float *>(rm_prog_smem) + tid;                              // [(s*3 + c) * nt]
    double *val = reinterpret_cast<double *>(rm_prog_smem + static_cast<size_t>(n_slots) * 3 * nt * 4) + tid;  // [k * nt]
    int sp = 0;
    pos[0] = p.x;
    pos[nt] = p.y;
    pos[2 * nt] = p.z;
    // (requesting the NEXT instruction's head one trip ahead was measured slower: 4.36 against 3.75 ms on the Chicken preset)
    for (int pc = first; pc < first + count; ++pc) {
        const RmInstrConstPtr I = prog + pc;
        const InstrHead H = load_head(I);
        const int op = H.op;
        if (op >= 20) {  // POST
            if (op == 20) {  // round.ts:24
                val[(sp - 1) * nt] = val[(sp - 1) * nt] - H.p[0];
            } else {
                const double d1 = val[(sp - 2) * nt], d2 = val[(sp - 1) * nt];
                const double k = H.p[0] * 4.0;
                sp -= 1;
                val[(sp - 1) * nt] = op == 21 ? post_smooth_union(d1, d2, k) : post_smooth_subtraction(d1, d2, k);
            }
            continue;
        }
        float lx, ly, lz;
        {
            const float *src = pos + H.src * 3 * nt;
            transform_head(I->T, H.t, (H.flags & 1) | ((H.flags >> 2) & 1) << 2, src[0], src[nt], src[2 * nt], lx, ly, lz);  // primitive.ts:34-35
        }
        if (op < 10) {  // leaves
            double d;
            if (op == 1) d = leaf_box(lx, ly, lz, H.p[0], H.p[1], H.p[2]);
            else if (op == 2) d = leaf_torus(lx, ly, lz, H.p[0], H.p[1]);
            else if (MB && op == 3) d = mandelbulb_sdf(H.p, lx, ly, lz, time);
            else d = leaf_sphere(lx, ly, lz, H.p[0]);
            val[sp * nt] = d;
            sp += 1;
            continue;
        }
        // PRE: the point the operands see
        float wx, wy, wz;
        if (op == 15) {
            pre_animated_translate(lx, ly, lz, H.p[0], H.p[1], H.p[2], H.p[3], H.p[4], time, wx, wy, wz);
        } else {
            transform_head(I->Tinv, H.ti, ((H.flags >> 1) & 1) | ((H.flags >> 3) & 1) << 2, lx, ly, lz, wx, wy, wz);  // "convert local position back to world space"
            if (op == 13) pre_twist(H.p[0], wx, wy, wz);
            else if (op == 14) pre_repetition(H.p[0], H.p[1], H.p[2], wx, wy, wz);
        }
        float *dst = pos + H.dst * 3 * nt;
        dst[0] = wx;
        dst[nt] = wy;
        dst[2 * nt] = wz;
    }
    return val[0];
}

}  // namespace rmd
