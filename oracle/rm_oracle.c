/*
 * rm_oracle.c -- CPU restatement of the reference's per-pixel sphere-tracing path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (cpu_raymarcher_amd/)
 * never calls into this file.
 *
 * PARITY UNPINNED: the reference (vxlerian/cpu-raymarcher) ships no tests, golden
 * vectors or fixtures for this path, cannot be transpiled here (TypeScript, no tsc) and
 * its arithmetic dependency gl-matrix@3.4.4 (package.json:25) is not vendored.  This
 * file is pinned by (1) hand-derived known answers from the reference source
 * (tests/test_oracle_kat.py), (2) byte-for-byte agreement with an independently written
 * JS restatement (oracle/rm_oracle.js) run on a real JS engine in the build container.
 *
 * Number semantics: every scalar is IEEE binary64 with one
 * rounding per operation and no FMA contraction (build with -ffp-contract=off); every
 * value the reference stores into a gl-matrix vec/mat (Float32Array) is rounded to
 * binary32 at the store; Uint8ClampedArray / Uint16Array store rules; JS Math.min/max
 * NaN and signed-zero rules; V8's Math.hypot (scaled Kahan sum).
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference/src).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ro_jsmath.h" /* Math.sin/cos/atan2/asin/pow/log/round as V8 computes them (fdlibm ports) */

#if defined(__FP_FAST_FMA) && defined(__FMA__)
/* fine: we still rely on -ffp-contract=off so a*b+c is never fused */
#endif

/* ------------------------------------------------------------------------- */
/* JS number semantics                                                       */
/* ------------------------------------------------------------------------- */

static inline float f32(double x) { return (float)x; } /* Float32Array store: RNE */

/* Math.min(a, b): NaN if either is NaN; -0 < +0 */
static inline double js_min(double a, double b) {
    if (a != a || b != b) return NAN;
    if (a == 0.0 && b == 0.0) return signbit(a) ? a : b;
    return a < b ? a : b;
}
/* Math.max(a, b) */
static inline double js_max(double a, double b) {
    if (a != a || b != b) return NAN;
    if (a == 0.0 && b == 0.0) return signbit(a) ? b : a;
    return a > b ? a : b;
}

/* V8 Math.hypot(x, y, z): max-scaled Kahan sum of squares */
double ro_hypot3(double x, double y, double z) {
    double a[3] = {x, y, z};
    int one_nan = 0;
    double max = 0.0;
    for (int i = 0; i < 3; i++) {
        if (a[i] != a[i]) { one_nan = 1; a[i] = 0.0; }
        else { a[i] = fabs(a[i]); if (a[i] > max) max = a[i]; }
    }
    if (max == INFINITY) return INFINITY;
    if (one_nan) return NAN;
    if (max == 0.0) return 0.0;
    double sum = 0.0, comp = 0.0;
    for (int i = 0; i < 3; i++) {
        double n = a[i] / max;
        double summand = (n * n) - comp;
        double prelim = sum + summand;
        comp = (prelim - sum) - summand;
        sum = prelim;
    }
    return sqrt(sum) * max;
}

/* gl-matrix vec3.length: Math.hypot in 3.0-3.4.3; switchable */
static int g_length_uses_sqrt = 0;
void ro_set_length_mode(int use_sqrt) { g_length_uses_sqrt = use_sqrt; }
static inline double vec3_length(const float *a) {
    double x = a[0], y = a[1], z = a[2];
    if (g_length_uses_sqrt) return sqrt(x * x + y * y + z * z);
    return ro_hypot3(x, y, z);
}

/* Uint8ClampedArray store (ToUint8Clamp): NaN->0, clamp, round half to even */
uint8_t ro_u8clamp(double x) {
    if (!(x > 0.0)) return 0; /* NaN, <= 0 */
    if (x >= 255.0) return 255;
    double f = floor(x);
    if (f + 0.5 < x) return (uint8_t)(f + 1.0);
    if (x < f + 0.5) return (uint8_t)f;
    uint8_t fi = (uint8_t)f;
    return (fi & 1) ? (uint8_t)(fi + 1) : fi;
}

/* ------------------------------------------------------------------------- */
/* gl-matrix restatements; all outputs are Float32Array  */
/* ------------------------------------------------------------------------- */

static void mat4_identity(float *o) {
    memset(o, 0, 16 * sizeof(float));
    o[0] = o[5] = o[10] = o[15] = 1.0f;
}

/* mat4.fromRotationTranslationScale(out, q, v, s) */
static void mat4_fromRTS(float *out, const double *q, const double *v, const double *s) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double x2 = x + x, y2 = y + y, z2 = z + z;
    double xx = x * x2, xy = x * y2, xz = x * z2;
    double yy = y * y2, yz = y * z2, zz = z * z2;
    double wx = w * x2, wy = w * y2, wz = w * z2;
    double sx = s[0], sy = s[1], sz = s[2];
    out[0] = f32((1 - (yy + zz)) * sx);
    out[1] = f32((xy + wz) * sx);
    out[2] = f32((xz - wy) * sx);
    out[3] = 0;
    out[4] = f32((xy - wz) * sy);
    out[5] = f32((1 - (xx + zz)) * sy);
    out[6] = f32((yz + wx) * sy);
    out[7] = 0;
    out[8] = f32((xz + wy) * sz);
    out[9] = f32((yz - wx) * sz);
    out[10] = f32((1 - (xx + yy)) * sz);
    out[11] = 0;
    out[12] = f32(v[0]);
    out[13] = f32(v[1]);
    out[14] = f32(v[2]);
    out[15] = 1;
}

/* mat4.invert(out, a); returns 0 when !det (gl-matrix returns null) */
static int mat4_invert(float *out, const float *a) {
    double a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3];
    double a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7];
    double a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11];
    double a30 = a[12], a31 = a[13], a32 = a[14], a33 = a[15];
    double b00 = a00 * a11 - a01 * a10;
    double b01 = a00 * a12 - a02 * a10;
    double b02 = a00 * a13 - a03 * a10;
    double b03 = a01 * a12 - a02 * a11;
    double b04 = a01 * a13 - a03 * a11;
    double b05 = a02 * a13 - a03 * a12;
    double b06 = a20 * a31 - a21 * a30;
    double b07 = a20 * a32 - a22 * a30;
    double b08 = a20 * a33 - a23 * a30;
    double b09 = a21 * a32 - a22 * a31;
    double b10 = a21 * a33 - a23 * a31;
    double b11 = a22 * a33 - a23 * a32;
    double det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06;
    if (!(det != 0.0)) return 0; /* !det : 0, -0 or NaN */
    det = 1.0 / det;
    float o[16];
    o[0] = f32((a11 * b11 - a12 * b10 + a13 * b09) * det);
    o[1] = f32((a02 * b10 - a01 * b11 - a03 * b09) * det);
    o[2] = f32((a31 * b05 - a32 * b04 + a33 * b03) * det);
    o[3] = f32((a22 * b04 - a21 * b05 - a23 * b03) * det);
    o[4] = f32((a12 * b08 - a10 * b11 - a13 * b07) * det);
    o[5] = f32((a00 * b11 - a02 * b08 + a03 * b07) * det);
    o[6] = f32((a32 * b02 - a30 * b05 - a33 * b01) * det);
    o[7] = f32((a20 * b05 - a22 * b02 + a23 * b01) * det);
    o[8] = f32((a10 * b10 - a11 * b08 + a13 * b06) * det);
    o[9] = f32((a01 * b08 - a00 * b10 - a03 * b06) * det);
    o[10] = f32((a30 * b04 - a31 * b02 + a33 * b00) * det);
    o[11] = f32((a21 * b02 - a20 * b04 - a23 * b00) * det);
    o[12] = f32((a11 * b07 - a10 * b09 - a12 * b06) * det);
    o[13] = f32((a00 * b09 - a01 * b07 + a02 * b06) * det);
    o[14] = f32((a31 * b01 - a30 * b03 - a32 * b00) * det);
    o[15] = f32((a20 * b03 - a21 * b01 + a22 * b00) * det);
    memcpy(out, o, sizeof o);
    return 1;
}

/* mat4.rotateY(out, a, rad), out != a */
static void mat4_rotateY(float *out, const float *a, double rad) {
    double s = sin(rad), c = cos(rad);
    double a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3];
    double a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11];
    out[4] = a[4]; out[5] = a[5]; out[6] = a[6]; out[7] = a[7];
    out[12] = a[12]; out[13] = a[13]; out[14] = a[14]; out[15] = a[15];
    out[0] = f32(a00 * c - a20 * s);
    out[1] = f32(a01 * c - a21 * s);
    out[2] = f32(a02 * c - a22 * s);
    out[3] = f32(a03 * c - a23 * s);
    out[8] = f32(a00 * s + a20 * c);
    out[9] = f32(a01 * s + a21 * c);
    out[10] = f32(a02 * s + a22 * c);
    out[11] = f32(a03 * s + a23 * c);
}

/* mat4.rotateX(out, a, rad), out != a */
static void mat4_rotateX(float *out, const float *a, double rad) {
    double s = sin(rad), c = cos(rad);
    double a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7];
    double a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11];
    out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; out[3] = a[3];
    out[12] = a[12]; out[13] = a[13]; out[14] = a[14]; out[15] = a[15];
    out[4] = f32(a10 * c + a20 * s);
    out[5] = f32(a11 * c + a21 * s);
    out[6] = f32(a12 * c + a22 * s);
    out[7] = f32(a13 * c + a23 * s);
    out[8] = f32(a20 * c - a10 * s);
    out[9] = f32(a21 * c - a11 * s);
    out[10] = f32(a22 * c - a12 * s);
    out[11] = f32(a23 * c - a13 * s);
}

/* mat4.translate(out, a, v), out != a; v is a Float32Array vec3 */
static void mat4_translate(float *out, const float *a, const float *v) {
    double x = v[0], y = v[1], z = v[2];
    double a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3];
    double a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7];
    double a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11];
    for (int i = 0; i < 12; i++) out[i] = a[i];
    out[12] = f32(a00 * x + a10 * y + a20 * z + a[12]);
    out[13] = f32(a01 * x + a11 * y + a21 * z + a[13]);
    out[14] = f32(a02 * x + a12 * y + a22 * z + a[14]);
    out[15] = f32(a03 * x + a13 * y + a23 * z + a[15]);
}

/* vec3.transformMat4(out, a, m) */
static void vec3_transformMat4(float *out, const float *a, const float *m) {
    double x = a[0], y = a[1], z = a[2];
    double w = m[3] * x + m[7] * y + m[11] * z + m[15];
    if (!(w != 0.0)) w = 1.0; /* w = w || 1.0 : 0, -0, NaN are falsy */
    float o0 = f32((m[0] * x + m[4] * y + m[8] * z + m[12]) / w);
    float o1 = f32((m[1] * x + m[5] * y + m[9] * z + m[13]) / w);
    float o2 = f32((m[2] * x + m[6] * y + m[10] * z + m[14]) / w);
    out[0] = o0; out[1] = o1; out[2] = o2;
}

/* vec3.transformMat3(out, a, m) */
static void vec3_transformMat3(float *out, const float *a, const float *m) {
    double x = a[0], y = a[1], z = a[2];
    float o0 = f32(x * m[0] + y * m[3] + z * m[6]);
    float o1 = f32(x * m[1] + y * m[4] + z * m[7]);
    float o2 = f32(x * m[2] + y * m[5] + z * m[8]);
    out[0] = o0; out[1] = o1; out[2] = o2;
}

/* vec3.normalize(out, a) */
static void vec3_normalize(float *out, const float *a) {
    double x = a[0], y = a[1], z = a[2];
    double len = x * x + y * y + z * z;
    if (len > 0) len = 1 / sqrt(len);
    out[0] = f32(a[0] * len);
    out[1] = f32(a[1] * len);
    out[2] = f32(a[2] * len);
}

/* vec3.scaleAndAdd(out, a, b, scale) */
static void vec3_scaleAndAdd(float *out, const float *a, const float *b, double s) {
    out[0] = f32(a[0] + b[0] * s);
    out[1] = f32(a[1] + b[1] * s);
    out[2] = f32(a[2] + b[2] * s);
}

static inline double vec3_dot(const float *a, const float *b) {
    return (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2];
}

/* ------------------------------------------------------------------------- */
/* Scene model                                                               */
/* ------------------------------------------------------------------------- */

typedef struct { float min[3], max[3]; } BBox; /* boundingBox.ts:5-12 (vec3.clone -> f32) */

/* primitives/primitive.ts:3-44, primitives/{sphere,box,torus,mandelbulb}.ts and the operator
 * classes of primitive_operations/ (each "extends Primitive" and wraps one or two operands) */
enum { PRIM_SPHERE = 0, PRIM_BOX = 1, PRIM_TORUS = 2, PRIM_MANDELBULB = 3,
       OP_ROUND = 10, OP_SMOOTH_UNION = 11, OP_SMOOTH_SUB = 12, OP_TWIST = 13, OP_REPETITION = 14, OP_ANIM = 15 };
typedef struct Prim {
    float transform[16]; /* world -> local; wrappers share their operand's (round.ts:9 ...), unions use identity */
    int type;
    double radius;       /* Sphere: stays a JS double (sphere.ts:5-9) */
    float halfSize[3];   /* Box: vec3.clone(halfSize) -> Float32Array (box.ts:8-11) */
    double majorRadius, minorRadius; /* Torus (torus.ts:8-12) */
    struct Prim *a, *b;  /* operands (owned) */
    double k;            /* Round.radius | smoothness | twistAmount | AnimatedTranslate.amplitude */
    double speed;        /* AnimatedTranslate.speed | Mandelbulb.animationSpeed */
    float vec[3];        /* Repetition.spacing | AnimatedTranslate.direction (normalised, f32) */
    double power;        /* Mandelbulb */
    int iterations, enableAnimation;
} Prim;

/* Scene.updateTime (scene.ts:135-140) hands `time` to every animated primitive before a render
 * (raymarcher.ts:58-59); renders run in parallel threads over one const scene, so the value
 * lives in thread-local storage for the duration of a call. */
static __thread double g_time = 0.0;

typedef struct BVHNode {
    BBox bounds;
    int *prims; int nprims;
    struct BVHNode *left, *right;
} BVHNode; /* bvh.ts:6-22 */

typedef struct OctNode {
    BBox bounds;
    int *prims; int nprims;
    struct OctNode **children; /* 8 or NULL */
    int level;
    int isEmpty;
    double minDistance;
} OctNode; /* octree.ts:6-26 */

enum { ACCEL_NONE = 0, ACCEL_OCTREE = 1, ACCEL_BVH = 2 };

typedef struct ro_scene {
    Prim *prims; int n;
    int accel;
    BVHNode *bvh;
    OctNode *octree;
    BBox *primBounds; /* octree.ts:47 cache */
    int bvh_leaves, bvh_nodes, bvh_depth;
    int oct_nodes, oct_leaves, oct_empty, oct_maxleafprims;
    /* camera.ts:3-19 */
    double pitch, yaw;
    float cameraTransform[16];
} ro_scene;

static inline double len3d(double x, double y, double z) {
    if (g_length_uses_sqrt) return sqrt(x * x + y * y + z * z);
    return ro_hypot3(x, y, z);
}

/* primitive.ts:20-30 getWorldPosition and its overrides (round.ts:30-33, twist.ts:42-45,
 * repetition.ts:35-38, animatedTranslate.ts:55-57: the operand's; smoothUnion.ts:52-60: the
 * midpoint; smoothSubstraction.ts:41-44: the first operand's) */
static void prim_world_position(const Prim *p, float *out) {
    if (p->type == OP_SMOOTH_UNION) {
        float p1[3], p2[3];
        prim_world_position(p->a, p1);
        prim_world_position(p->b, p2);
        for (int i = 0; i < 3; i++) out[i] = f32(((double)p1[i] + (double)p2[i]) / 2);
        return;
    }
    if (p->type >= OP_ROUND) { prim_world_position(p->a, out); return; }
    float l2w[16];
    mat4_identity(l2w);
    mat4_invert(l2w, p->transform);
    out[0] = l2w[12]; out[1] = l2w[13]; out[2] = l2w[14];
}

/* mandelbulb.ts:37-78 localSdf.  z is a vec3 (Float32Array): every z[i] store rounds to binary32 */
static double mandelbulb_local_sdf(const Prim *m, const float *local) {
    const float p[3] = {local[0], local[2], local[1]};
    float z[3] = {p[0], p[1], p[2]};
    double dr = 1.0, r = 0.0;
    for (int i = 0; i < m->iterations; i++) {
        r = vec3_length(z);
        if (r > 2.0) break;
        double theta = js_atan2(z[1], z[0]);
        double phi = js_asin((double)z[2] / r);
        if (m->enableAnimation) phi += g_time * m->speed;
        dr = js_pow(r, m->power - 1.0) * dr * m->power + 1.0;
        r = js_pow(r, m->power);
        theta = theta * m->power;
        phi = phi * m->power;
        z[0] = f32(r * js_cos(theta) * js_cos(phi) + (double)p[0]);
        z[1] = f32(r * js_sin(theta) * js_cos(phi) + (double)p[1]);
        z[2] = f32(r * js_sin(phi) + (double)p[2]);
    }
    return 0.5 * js_log(r) * r / dr;
}

static double prim_sdf(const Prim *p, const float *pos);

/* the "convert local position back to world space" prologue the operators share
 * (round.ts:17-21, twist.ts:16-19, repetition.ts:14-18, smoothUnion.ts:20-23) */
static void op_world_pos(const Prim *p, const float *local, float *world) {
    float l2w[16];
    mat4_identity(l2w);
    mat4_invert(l2w, p->transform);
    vec3_transformMat4(world, local, l2w);
}

/* primitive.ts:33-39 sdf + the localSdf of each class */
static double prim_sdf(const Prim *p, const float *pos) {
    float local[3];
    vec3_transformMat4(local, pos, p->transform);
    if (p->type >= OP_ROUND) {
        float w[3];
        switch (p->type) {
        case OP_ROUND: /* round.ts:16-25 */
            op_world_pos(p, local, w);
            return prim_sdf(p->a, w) - p->k;
        case OP_TWIST: { /* twist.ts:14-36 */
            op_world_pos(p, local, w);
            double c = js_cos(p->k * (double)w[1]), sn = js_sin(p->k * (double)w[1]);
            float t[3] = {f32(c * (double)w[0] - sn * (double)w[2]), w[1], f32(sn * (double)w[0] + c * (double)w[2])};
            return prim_sdf(p->a, t);
        }
        case OP_REPETITION: { /* repetition.ts:13-29 */
            op_world_pos(p, local, w);
            float q[3];
            for (int i = 0; i < 3; i++)
                q[i] = f32((double)w[i] - (double)p->vec[i] * js_round((double)w[i] / (double)p->vec[i]));
            return prim_sdf(p->a, q);
        }
        case OP_ANIM: { /* animatedTranslate.ts:34-49: the operand re-applies its own transform */
            double offset = js_sin(g_time * p->speed) * p->k;
            float adj[3];
            for (int i = 0; i < 3; i++) {
                float off = f32((double)p->vec[i] * offset);
                adj[i] = f32((double)local[i] - (double)off);
            }
            return prim_sdf(p->a, adj);
        }
        case OP_SMOOTH_UNION: { /* smoothUnion.ts:18-35 */
            op_world_pos(p, local, w);
            double d1 = prim_sdf(p->a, w), d2 = prim_sdf(p->b, w);
            double k = p->k * 4.0;
            double h = js_max(k - fabs(d1 - d2), 0.0);
            return js_min(d1, d2) - h * h * 0.25 / k;
        }
        default: { /* OP_SMOOTH_SUB, smoothSubstraction.ts:16-34 */
            op_world_pos(p, local, w);
            double d1 = prim_sdf(p->a, w), d2 = prim_sdf(p->b, w);
            double k = p->k * 4.0;
            double h = js_max(k - fabs(d1 + d2), 0.0);
            return js_max(d1, -d2) + h * h * 0.25 / k;
        }
        }
    }
    if (p->type == PRIM_MANDELBULB) return mandelbulb_local_sdf(p, local);
    if (p->type == PRIM_BOX) { /* box.ts:13-30 */
        float q[3], outside[3];
        for (int i = 0; i < 3; i++) {
            q[i] = f32(fabs((double)local[i]) - (double)p->halfSize[i]);
            outside[i] = f32(js_max(q[i], 0));
        }
        double outsideDist = vec3_length(outside);
        double insideDist = js_min(js_max(q[0], js_max(q[1], q[2])), 0);
        return outsideDist + insideDist;
    }
    if (p->type == PRIM_TORUS) { /* torus.ts:14-25 */
        double x = local[0], y = local[1], z = local[2];
        double qx = sqrt(x * x + z * z) - p->majorRadius;
        double qy = y;
        return sqrt(qx * qx + qy * qy) - p->minorRadius;
    }
    return vec3_length(local) - p->radius;
}

/* getLocalBoundingRadius: sphere.ts:16-18, box.ts:32-34, torus.ts:27-29, mandelbulb.ts:80-83,
 * round.ts:27-30, twist.ts:38-41, repetition.ts:31-34, animatedTranslate.ts:51-54,
 * smoothUnion.ts:37-49, smoothSubstraction.ts:36-39 */
static double prim_local_radius(const Prim *p) {
    switch (p->type) {
    case PRIM_MANDELBULB: return 2.5;
    case OP_ROUND: return prim_local_radius(p->a) + p->k;
    case OP_TWIST: return prim_local_radius(p->a);
    case OP_REPETITION: return INFINITY;
    case OP_ANIM: return prim_local_radius(p->a) + p->k;
    case OP_SMOOTH_SUB: return prim_local_radius(p->a);
    case OP_SMOOTH_UNION: {
        double r1 = prim_local_radius(p->a), r2 = prim_local_radius(p->b);
        float p1[3], p2[3];
        prim_world_position(p->a, p1);
        prim_world_position(p->b, p2);
        double centerDist = len3d((double)p2[0] - (double)p1[0], (double)p2[1] - (double)p1[1], (double)p2[2] - (double)p1[2]);
        return js_max(r1, r2) + centerDist * 0.5;
    }
    default: break;
    }
    if (p->type == PRIM_BOX) return vec3_length(p->halfSize);
    if (p->type == PRIM_TORUS) return p->majorRadius + p->minorRadius;
    return p->radius;
}

/* boundingBox.ts:15-21 */
static inline int bbox_contains(const BBox *b, const float *pt) {
    return pt[0] >= b->min[0] && pt[0] <= b->max[0] &&
           pt[1] >= b->min[1] && pt[1] <= b->max[1] &&
           pt[2] >= b->min[2] && pt[2] <= b->max[2];
}
/* boundingBox.ts:24-30 */
static inline int bbox_intersects(const BBox *a, const BBox *o) {
    return a->min[0] <= o->max[0] && a->max[0] >= o->min[0] &&
           a->min[1] <= o->max[1] && a->max[1] >= o->min[1] &&
           a->min[2] <= o->max[2] && a->max[2] >= o->min[2];
}
/* boundingBox.ts:33-47 */
static double bbox_distanceToBox(const BBox *a, const BBox *o) {
    double d[3];
    for (int i = 0; i < 3; i++) {
        d[i] = 0;
        if (a->max[i] < o->min[i]) d[i] = (double)o->min[i] - (double)a->max[i];
        else if (o->max[i] < a->min[i]) d[i] = (double)a->min[i] - (double)o->max[i];
    }
    return ro_hypot3(d[0], d[1], d[2]);
}
/* boundingBox.ts:69-105; returns 0 for null */
static int bbox_intersectRay(const BBox *b, const float *origin, const float *dir,
                             double *outMin, double *outMax) {
    double tMin = -INFINITY, tMax = INFINITY;
    for (int i = 0; i < 3; i++) {
        if (fabs((double)dir[i]) < 1e-10) {
            if (origin[i] < b->min[i] || origin[i] > b->max[i]) return 0;
        } else {
            double invD = 1.0 / (double)dir[i];
            double t0 = ((double)b->min[i] - (double)origin[i]) * invD;
            double t1 = ((double)b->max[i] - (double)origin[i]) * invD;
            if (t0 > t1) { double t = t0; t0 = t1; t1 = t; }
            tMin = js_max(tMin, t0);
            tMax = js_min(tMax, t1);
            if (tMin > tMax) return 0;
        }
    }
    *outMin = tMin; *outMax = tMax;
    return 1;
}
/* boundingBox.ts:108-114 */
static void bbox_center(const BBox *b, float *c) {
    for (int i = 0; i < 3; i++) c[i] = f32(((double)b->min[i] + (double)b->max[i]) / 2);
}
/* boundingBox.ts:133-154 */
static void bbox_fromPrimitive(const Prim *p, BBox *out) {
    float wp[3];
    prim_world_position(p, wp);
    double localRadius = prim_local_radius(p);
    float l2w[16];
    mat4_identity(l2w);
    int ok = mat4_invert(l2w, p->transform);
    const float *m = ok ? l2w : p->transform;
    double scaleX = ro_hypot3(m[0], m[1], m[2]);
    double scaleY = ro_hypot3(m[4], m[5], m[6]);
    double scaleZ = ro_hypot3(m[8], m[9], m[10]);
    double maxScale = js_max(js_max(scaleX, scaleY), scaleZ);
    double r = localRadius * maxScale * 1.5;
    for (int i = 0; i < 3; i++) {
        out->min[i] = f32((double)wp[i] - r);
        out->max[i] = f32((double)wp[i] + r);
    }
}
/* boundingBox.ts:117-130 */
static void bbox_merge(BBox *a, const BBox *o) {
    for (int i = 0; i < 3; i++) {
        a->min[i] = f32(js_min(a->min[i], o->min[i]));
        a->max[i] = f32(js_max(a->max[i], o->max[i]));
    }
}
/* boundingBox.ts:158-169 */
static void compute_bounds(const ro_scene *s, const int *ids, int n, BBox *out) {
    if (n == 0) { memset(out, 0, sizeof *out); return; }
    bbox_fromPrimitive(&s->prims[ids[0]], out);
    for (int i = 1; i < n; i++) {
        BBox b;
        bbox_fromPrimitive(&s->prims[ids[i]], &b);
        bbox_merge(out, &b);
    }
}

/* stable merge sort of ids by key (Array.prototype.sort is stable, Appendix A.7);
 * comparator aPos - bPos : a before b iff key[a] < key[b]; ties keep order */
static void stable_sort_by_key(int *ids, const double *keyOfId, int n, int *tmp) {
    if (n < 2) return;
    int mid = n / 2;
    stable_sort_by_key(ids, keyOfId, mid, tmp);
    stable_sort_by_key(ids + mid, keyOfId, n - mid, tmp);
    int i = 0, j = mid, k = 0;
    while (i < mid && j < n) {
        if (keyOfId[ids[j]] - keyOfId[ids[i]] < 0) tmp[k++] = ids[j++];
        else tmp[k++] = ids[i++];
    }
    while (i < mid) tmp[k++] = ids[i++];
    while (j < n) tmp[k++] = ids[j++];
    memcpy(ids, tmp, (size_t)n * sizeof(int));
}

/* ------------------------------------------------------------------------- */
/* BVH (bvh.ts)                                                              */
/* ------------------------------------------------------------------------- */

static BVHNode *bvh_build(ro_scene *s, const int *ids, int n, const BBox *bounds, int depth,
                          const float *worldPos /* n_total x 3 */) {
    /* bvh.ts:44-92 */
    BVHNode *node = (BVHNode *)calloc(1, sizeof *node);
    node->bounds = *bounds;
    s->bvh_nodes++;
    if (depth > s->bvh_depth) s->bvh_depth = depth;
    if (depth >= 20 || n <= 2) {
        node->prims = (int *)malloc((size_t)(n ? n : 1) * sizeof(int));
        memcpy(node->prims, ids, (size_t)n * sizeof(int));
        node->nprims = n;
        s->bvh_leaves++;
        return node;
    }
    float size[3];
    for (int i = 0; i < 3; i++) size[i] = f32((double)bounds->max[i] - (double)bounds->min[i]);
    int axis = 0;
    if (size[1] > size[0]) axis = 1;
    if (size[2] > size[axis]) axis = 2;

    int *sorted = (int *)malloc((size_t)n * sizeof(int));
    int *tmp = (int *)malloc((size_t)n * sizeof(int));
    memcpy(sorted, ids, (size_t)n * sizeof(int));
    double *keys = (double *)malloc((size_t)s->n * sizeof(double));
    for (int i = 0; i < n; i++) keys[ids[i]] = worldPos[3 * ids[i] + axis];
    stable_sort_by_key(sorted, keys, n, tmp);
    free(keys); free(tmp);

    int mid = n / 2;
    if (mid == 0 || n - mid == 0) { /* bvh.ts:78-81 */
        node->prims = (int *)malloc((size_t)n * sizeof(int));
        memcpy(node->prims, ids, (size_t)n * sizeof(int));
        node->nprims = n;
        s->bvh_leaves++;
        free(sorted);
        return node;
    }
    BBox lb, rb;
    compute_bounds(s, sorted, mid, &lb);
    compute_bounds(s, sorted + mid, n - mid, &rb);
    node->left = bvh_build(s, sorted, mid, &lb, depth + 1, worldPos);
    node->right = bvh_build(s, sorted + mid, n - mid, &rb, depth + 1, worldPos);
    free(sorted);
    return node;
}

static void bvh_free(BVHNode *n) {
    if (!n) return;
    bvh_free(n->left); bvh_free(n->right);
    free(n->prims); free(n);
}

/* bvh.ts:101-121 queryNode; appends leaf prims to out (a prim lives in one leaf) */
static void bvh_query(const BVHNode *node, const float *pt, int *out, int *nout) {
    if (!bbox_contains(&node->bounds, pt)) return;
    if (!node->left && !node->right) {
        for (int i = 0; i < node->nprims; i++) {
            int id = node->prims[i], dup = 0;
            for (int k = 0; k < *nout; k++) if (out[k] == id) { dup = 1; break; } /* Set */
            if (!dup) out[(*nout)++] = id;
        }
        return;
    }
    if (node->left) bvh_query(node->left, pt, out, nout);
    if (node->right) bvh_query(node->right, pt, out, nout);
}

typedef struct { double tEnter, tExit; } Interval;

/* bvh.ts:126-178 findRayIntersections; returns count; list sorted stably by tEnter */
static int bvh_find_intervals(const ro_scene *s, const float *origin, const float *dir,
                              double tMinArg, double tMaxArg, Interval *list,
                              const BVHNode **stack) {
    int count = 0, sp = 0;
    stack[sp++] = s->bvh;
    while (sp > 0) {
        const BVHNode *node = stack[--sp];
        double tEnter, tExit;
        if (!bbox_intersectRay(&node->bounds, origin, dir, &tEnter, &tExit)) continue;
        if (tExit < tMinArg || tEnter > tMaxArg) continue;
        double cEnter = js_max(tEnter, tMinArg);
        double cExit = js_min(tExit, tMaxArg);
        if (node->left || node->right) {
            if (node->left) stack[sp++] = node->left;
            if (node->right) stack[sp++] = node->right;
        } else if (node->nprims > 0) {
            list[count].tEnter = cEnter;
            list[count].tExit = cExit;
            count++;
        }
    }
    /* stable insertion sort by tEnter (comparator a.tEnter - b.tEnter) */
    for (int i = 1; i < count; i++) {
        Interval v = list[i];
        int j = i - 1;
        while (j >= 0 && (v.tEnter - list[j].tEnter) < 0) { list[j + 1] = list[j]; j--; }
        list[j + 1] = v;
    }
    return count;
}

/* ------------------------------------------------------------------------- */
/* Octree (octree.ts)                                                        */
/* ------------------------------------------------------------------------- */

static OctNode *oct_new(const BBox *b, int level) {
    OctNode *n = (OctNode *)calloc(1, sizeof *n);
    n->bounds = *b; n->level = level; n->isEmpty = 1; n->minDistance = 0;
    return n;
}

static OctNode *oct_build(ro_scene *s, const int *ids, int n, const BBox *bounds, int depth) {
    /* octree.ts:52-118 */
    OctNode *node = oct_new(bounds, depth);
    s->oct_nodes++;
    if (depth >= 6 || n <= 4) {
        node->prims = (int *)malloc((size_t)(n ? n : 1) * sizeof(int));
        memcpy(node->prims, ids, (size_t)n * sizeof(int));
        node->nprims = n;
        return node;
    }
    float c[3];
    bbox_center(bounds, c);
    BBox cb[8];
    int k = 0;
    for (int zs = 0; zs < 2; zs++)
        for (int ys = 0; ys < 2; ys++)
            for (int xs = 0; xs < 2; xs++) {
                cb[k].min[0] = xs == 0 ? bounds->min[0] : c[0];
                cb[k].max[0] = xs == 0 ? c[0] : bounds->max[0];
                cb[k].min[1] = ys == 0 ? bounds->min[1] : c[1];
                cb[k].max[1] = ys == 0 ? c[1] : bounds->max[1];
                cb[k].min[2] = zs == 0 ? bounds->min[2] : c[2];
                cb[k].max[2] = zs == 0 ? c[2] : bounds->max[2];
                k++;
            }
    int *child_ids[8]; int child_n[8];
    for (int i = 0; i < 8; i++) { child_ids[i] = (int *)malloc((size_t)n * sizeof(int)); child_n[i] = 0; }
    for (int j = 0; j < n; j++) {
        BBox pb;
        bbox_fromPrimitive(&s->prims[ids[j]], &pb);
        for (int i = 0; i < 8; i++)
            if (bbox_intersects(&cb[i], &pb)) child_ids[i][child_n[i]++] = ids[j];
    }
    node->children = (OctNode **)malloc(8 * sizeof(OctNode *));
    for (int i = 0; i < 8; i++) {
        if (child_n[i] > 0) node->children[i] = oct_build(s, child_ids[i], child_n[i], &cb[i], depth + 1);
        else { node->children[i] = oct_new(&cb[i], depth + 1); s->oct_nodes++; }
        free(child_ids[i]);
    }
    return node;
}

/* octree.ts:149-191 */
static int oct_compute_min_distances(ro_scene *s, OctNode *node) {
    if (!node->children) {
        int hasPrims = node->nprims > 0;
        node->isEmpty = !hasPrims;
        s->oct_leaves++;
        if (hasPrims && node->nprims > s->oct_maxleafprims) s->oct_maxleafprims = node->nprims;
        if (!hasPrims) {
            s->oct_empty++;
            double minD = INFINITY;
            for (int i = 0; i < s->n; i++) {
                double d = bbox_distanceToBox(&node->bounds, &s->primBounds[i]);
                if (d < minD) minD = d;
            }
            node->minDistance = minD != INFINITY ? js_max(0, minD) : 0;
        } else node->minDistance = 0;
        return hasPrims;
    }
    int sub = 0;
    for (int i = 0; i < 8; i++) if (oct_compute_min_distances(s, node->children[i])) sub = 1;
    node->isEmpty = !sub;
    if (node->isEmpty) {
        double minD = INFINITY;
        for (int i = 0; i < s->n; i++) {
            double d = bbox_distanceToBox(&node->bounds, &s->primBounds[i]);
            if (d < minD) minD = d;
        }
        node->minDistance = minD != INFINITY ? js_max(0, minD) : 0;
    } else node->minDistance = 0;
    return sub;
}

static void oct_free(OctNode *n) {
    if (!n) return;
    if (n->children) { for (int i = 0; i < 8; i++) oct_free(n->children[i]); free(n->children); }
    free(n->prims); free(n);
}

/* octree.ts:227-248 findNodeRecursive */
static const OctNode *oct_find(const OctNode *node, const float *pt) {
    if (!bbox_contains(&node->bounds, pt)) return NULL;
    if (!node->children || node->level == 6) return node;
    for (int i = 0; i < 8; i++) {
        const OctNode *f = oct_find(node->children[i], pt);
        if (f) return f;
    }
    return node;
}

/* octree.ts:195-220 intersectRayBox (tMin/tMax are Float32Array); 0 for null */
static int oct_intersectRayBox(const float *o, const float *d, const BBox *box,
                               double *tEnterOut, double *tExitOut) {
    float tMin[3], tMax[3];
    for (int i = 0; i < 3; i++) {
        double invD = 1.0 / (double)d[i];
        double t0 = ((double)box->min[i] - (double)o[i]) * invD;
        double t1 = ((double)box->max[i] - (double)o[i]) * invD;
        if (invD < 0.0) { double t = t0; t0 = t1; t1 = t; }
        tMin[i] = f32(t0);
        tMax[i] = f32(t1);
    }
    double tEnter = js_max(js_max(tMin[0], tMin[1]), tMin[2]);
    double tExit = js_min(js_min(tMax[0], tMax[1]), tMax[2]);
    if (tEnter > tExit || tExit < 0) return 0;
    *tEnterOut = js_max(0, tEnter);
    *tExitOut = tExit;
    return 1;
}

/* octree.ts:252-278 marchRay */
static double oct_marchRay(const ro_scene *s, const float *o, const float *d, double currentDist) {
    float cur[3];
    vec3_scaleAndAdd(cur, o, d, currentDist);
    const OctNode *node = oct_find(s->octree, cur);
    if (!node) return 0;
    if (node->isEmpty) {
        double tEnter, tExit;
        if (oct_intersectRayBox(o, d, &node->bounds, &tEnter, &tExit)) {
            double toExit = js_max(0, tExit - currentDist);
            double step = js_max(0, js_min(toExit, node->minDistance * 0.99));
            return step > 0 ? step + 0.001 : 0;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Scene (scene.ts)                                                          */
/* ------------------------------------------------------------------------- */

/* camera.ts:81-88 */
static void camera_update(ro_scene *s) {
    float ident[16], tmp[16], orbit[16];
    mat4_identity(ident);
    mat4_identity(tmp);
    mat4_identity(orbit);
    mat4_rotateY(tmp, ident, s->yaw);
    mat4_rotateX(orbit, tmp, s->pitch);
    float v[3] = {0.0f, 0.0f, f32(fabs(3.0))};
    mat4_identity(s->cameraTransform);
    mat4_translate(s->cameraTransform, orbit, v);
}

/* camera.ts:58-62 */
void ro_scene_set_angles(ro_scene *s, double pitch, double yaw) {
    const double HALF_PI = 3.141592653589793 / 2;
    s->pitch = js_min(js_max(pitch, -HALF_PI), HALF_PI);
    s->yaw = yaw;
    camera_update(s);
}

static int parse_accel(const char *a) { /* scene.ts:32-36 */
    if (a && strcmp(a, "Octree") == 0) return ACCEL_OCTREE;
    if (a && strcmp(a, "BVH") == 0) return ACCEL_BVH;
    return ACCEL_NONE;
}

static void scene_build_accel(ro_scene *s) { /* scene.ts:38-70 */
    int *all = (int *)malloc((size_t)(s->n ? s->n : 1) * sizeof(int));
    for (int i = 0; i < s->n; i++) all[i] = i;
    if (s->accel == ACCEL_BVH) {
        size_t nwp = s->n > 0 ? (size_t)s->n : 1;
        float *wp = (float *)malloc(nwp * 3 * sizeof(float));
        for (int i = 0; i < s->n; i++) prim_world_position(&s->prims[i], wp + 3 * i);
        BBox root;
        compute_bounds(s, all, s->n, &root); /* bvh.ts:38-41: ctor arg ignored */
        s->bvh = bvh_build(s, all, s->n, &root, 0, wp);
        free(wp);
    } else if (s->accel == ACCEL_OCTREE) {
        BBox root = {{-10, -10, -10}, {10, 10, 10}}; /* scene.ts:81-85 */
        s->primBounds = (BBox *)malloc((size_t)(s->n ? s->n : 1) * sizeof(BBox));
        for (int i = 0; i < s->n; i++) bbox_fromPrimitive(&s->prims[i], &s->primBounds[i]);
        s->octree = oct_build(s, all, s->n, &root, 0);
        oct_compute_min_distances(s, s->octree);
    }
    free(all);
}

/* sceneManager.ts:21-41 getTransform (no rotation) + createSphere */
static void make_sphere(Prim *p, double x, double y, double z, double radius) {
    float model[16];
    mat4_identity(model);
    double q[4] = {0, 0, 0, 1}, v[3] = {x, y, z}, sc[3] = {1, 1, 1};
    mat4_fromRTS(model, q, v, sc);
    mat4_identity(p->transform);
    mat4_invert(p->transform, model);
    p->radius = radius;
}

/* mat4.fromTranslation + mat4.rotateZ (in place forms have the same arithmetic) */
static void mat4_rotateZ(float *out, const float *a, double rad) {
    double s = sin(rad), c = cos(rad);
    double a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3];
    double a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7];
    float o[16];
    memcpy(o, a, sizeof o);
    o[0] = f32(a00 * c + a10 * s); o[1] = f32(a01 * c + a11 * s); o[2] = f32(a02 * c + a12 * s); o[3] = f32(a03 * c + a13 * s);
    o[4] = f32(a10 * c - a00 * s); o[5] = f32(a11 * c - a01 * s); o[6] = f32(a12 * c - a02 * s); o[7] = f32(a13 * c - a03 * s);
    memcpy(out, o, sizeof o);
}

/* sceneManager.ts:21-37 getTransform; rot == NULL: the fromRotationTranslationScale branch.
 * `rot` is a gl-matrix vec3 (Float32Array), e.g. vec3.fromValues(-Math.PI/2, 0, 0). */
static void get_transform(float *worldToLocal, double x, double y, double z, const float *rot) {
    float model[16];
    mat4_identity(model);
    if (rot) {
        float t[16], u[16];
        model[12] = f32(x); model[13] = f32(y); model[14] = f32(z); /* mat4.fromTranslation */
        mat4_rotateX(t, model, rot[0]);
        mat4_rotateY(u, t, rot[1]);
        mat4_rotateZ(model, u, rot[2]);
    } else {
        double q[4] = {0, 0, 0, 1}, v[3] = {x, y, z}, sc[3] = {1, 1, 1};
        mat4_fromRTS(model, q, v, sc);
    }
    mat4_identity(worldToLocal);
    mat4_invert(worldToLocal, model);
}

/* sceneManager.ts:43-49 createBox / createTorus */
static void make_box(Prim *p, double x, double y, double z, double hx, double hy, double hz, const float *rot) {
    get_transform(p->transform, x, y, z, rot);
    p->type = PRIM_BOX;
    p->halfSize[0] = f32(hx); p->halfSize[1] = f32(hy); p->halfSize[2] = f32(hz);
}
static void make_torus(Prim *p, double x, double y, double z, double radius, const float *rot) {
    get_transform(p->transform, x, y, z, rot);
    p->type = PRIM_TORUS;
    p->majorRadius = radius;
    p->minorRadius = radius / 4;
}

/* gl-matrix mat4.scale(out, a, v), in place */
static void mat4_scale(float *m, double x, double y, double z) {
    for (int i = 0; i < 4; i++) {
        m[i] = f32((double)m[i] * x);
        m[4 + i] = f32((double)m[4 + i] * y);
        m[8 + i] = f32((double)m[8 + i] * z);
    }
}

/* by-value constructors for expression trees (sceneManager.ts:39-100) */
static Prim mk_sphere(double x, double y, double z, double radius, const float *rot) {
    Prim p; memset(&p, 0, sizeof p);
    get_transform(p.transform, x, y, z, rot);
    p.type = PRIM_SPHERE; p.radius = radius;
    return p;
}
static Prim mk_box(double x, double y, double z, double hx, double hy, double hz, const float *rot) {
    Prim p; memset(&p, 0, sizeof p);
    make_box(&p, x, y, z, hx, hy, hz, rot);
    return p;
}
static Prim mk_torus(double x, double y, double z, double radius, const float *rot) {
    Prim p; memset(&p, 0, sizeof p);
    make_torus(&p, x, y, z, radius, rot);
    return p;
}
/* sceneManager.ts:51-73 createMandelbulb: the world->local matrix is post-scaled by 0.5 */
static Prim mk_mandelbulb(double x, double y, double z, double power, int iterations, int anim, double speed, const float *rot) {
    Prim p; memset(&p, 0, sizeof p);
    get_transform(p.transform, x, y, z, rot);
    mat4_scale(p.transform, 0.5, 0.5, 0.5);
    p.type = PRIM_MANDELBULB; p.power = power; p.iterations = iterations; p.enableAnimation = anim; p.speed = speed;
    return p;
}
static Prim *heap_prim(Prim v) {
    Prim *h = (Prim *)malloc(sizeof *h);
    *h = v;
    return h;
}
/* Round / Twist / Repetition / AnimatedTranslate: super(primitive.transform) */
static Prim mk_wrap(int type, Prim operand, double k) {
    Prim p; memset(&p, 0, sizeof p);
    memcpy(p.transform, operand.transform, sizeof p.transform);
    p.type = type; p.k = k; p.a = heap_prim(operand);
    return p;
}
static Prim mk_repetition(Prim operand, double sx, double sy, double sz) {
    Prim p = mk_wrap(OP_REPETITION, operand, 0);
    p.vec[0] = f32(sx); p.vec[1] = f32(sy); p.vec[2] = f32(sz);
    return p;
}
/* animatedTranslate.ts:14-27: direction is normalised into a fresh vec3 */
static Prim mk_anim(Prim operand, double dx, double dy, double dz, double amplitude, double speed) {
    Prim p = mk_wrap(OP_ANIM, operand, amplitude);
    float d[3] = {f32(dx), f32(dy), f32(dz)};
    vec3_normalize(p.vec, d);
    p.speed = speed;
    return p;
}
/* SmoothUnion / SmoothSubtraction: super(mat4.create()) */
static Prim mk_smooth(int type, Prim a, Prim b, double k) {
    Prim p; memset(&p, 0, sizeof p);
    mat4_identity(p.transform);
    p.type = type; p.k = k; p.a = heap_prim(a); p.b = heap_prim(b);
    return p;
}
static void prim_free_children(Prim *p) {
    if (p->a) { prim_free_children(p->a); free(p->a); }
    if (p->b) { prim_free_children(p->b); free(p->b); }
}

static ro_scene *scene_alloc(int n, const char *accel) {
    ro_scene *s = (ro_scene *)calloc(1, sizeof *s);
    s->n = n;
    s->prims = (Prim *)calloc((size_t)(n ? n : 1), sizeof(Prim));
    s->accel = parse_accel(accel);
    s->pitch = 0; s->yaw = 0;
    camera_update(s);
    return s;
}

#define RO_PRESET_COUNT 19 /* sceneManager.ts:102-357 */

/* sceneManager.ts:102-357 -- all 19 presets */
ro_scene *ro_scene_from_preset(int index, const char *accel) {
    /* scene.ts:39 / sceneManager.ts:359-361 clamp */
    if (index < 0) index = 0;
    if (index > RO_PRESET_COUNT - 1) index = RO_PRESET_COUNT - 1;
    ro_scene *s = NULL;
    switch (index) {
    case 0:
        s = scene_alloc(1, accel);
        make_sphere(&s->prims[0], 0, 0, 0, 1.5);
        break;
    case 1: {
        static const double v[7][4] = {
            {0.8, -0.3, 0.2, 0.4}, {-0.5, 0.9, -0.1, 0.5}, {0.2, 0.1, 0.8, 0.3},
            {-0.9, -0.4, -0.6, 0.6}, {0.4, -0.8, 0.5, 0.35}, {-0.2, 0.6, -0.9, 0.4},
            {0.7, 0.3, -0.4, 0.25}};
        s = scene_alloc(7, accel);
        for (int i = 0; i < 7; i++) make_sphere(&s->prims[i], v[i][0], v[i][1], v[i][2], v[i][3]);
        break;
    }
    case 2: {
        s = scene_alloc(9, accel);
        int k = 0;
        for (int y = -1; y <= 1; y++)
            for (int x = -1; x <= 1; x++) make_sphere(&s->prims[k++], x, y, 0, 0.3);
        break;
    }
    case 3: {
        int gridSize = 5;
        double spacing = 0.6;
        double offset = (gridSize - 1) * spacing / 2;
        s = scene_alloc(125, accel);
        int k = 0;
        for (int x = 0; x < gridSize; x++)
            for (int y = 0; y < gridSize; y++)
                for (int z = 0; z < gridSize; z++)
                    make_sphere(&s->prims[k++], x * spacing - offset, y * spacing - offset,
                                z * spacing - offset, 0.15);
        break;
    }
    case 4: {
        static const double v[7][4] = {
            {0, 0, 0, 0.5}, {1.2, 0, 0, 0.3}, {-1.2, 0, 0, 0.3}, {0, 1.2, 0, 0.3},
            {0, -1.2, 0, 0.3}, {0, 0, 1.2, 0.3}, {0, 0, -1.2, 0.3}};
        s = scene_alloc(7, accel);
        for (int i = 0; i < 7; i++) make_sphere(&s->prims[i], v[i][0], v[i][1], v[i][2], v[i][3]);
        break;
    }
    case 5: { /* "Torus", sceneManager.ts:171-176 */
        const float rot[3] = {f32(-3.141592653589793 / 2), 0.0f, 0.0f};
        s = scene_alloc(1, accel);
        make_torus(&s->prims[0], 0, 0, 0, 1.3, rot);
        break;
    }
    case 7: /* "Cube" */
        s = scene_alloc(1, accel);
        make_box(&s->prims[0], 0, 0, 0, 1, 1, 1, NULL);
        break;
    case 8: /* "Sphere and Cube" */
        s = scene_alloc(2, accel);
        make_sphere(&s->prims[0], -0.7, 0, 0, 0.5);
        make_box(&s->prims[1], 1, 0, 0, 0.5, 0.5, 0.5, NULL);
        break;
    case 9: /* "Pyramid of Boxes" */
        s = scene_alloc(3, accel);
        make_box(&s->prims[0], 0, 0.5, 0, 0.9, 0.25, 0.9, NULL);
        make_box(&s->prims[1], 0, 0, 0, 0.6, 0.25, 0.6, NULL);
        make_box(&s->prims[2], 0, -0.5, 0, 0.3, 0.25, 0.3, NULL);
        break;
    case 6: /* "Rounded Box", sceneManager.ts:178-186 */
        s = scene_alloc(1, accel);
        s->prims[0] = mk_wrap(OP_ROUND, mk_box(0, 0, 0, 0.4, 0.4, 0.4, NULL), 0.3);
        break;
    case 10: /* "Smooth Union" */
        s = scene_alloc(1, accel);
        s->prims[0] = mk_smooth(OP_SMOOTH_UNION, mk_sphere(0, 0, 0, 0.5, NULL), mk_box(0, 0.5, 0, 1, 0.2, 1, NULL), 0.2);
        break;
    case 11: { /* "Smooth Subtraction" */
        const float rot[3] = {0.0f, f32(3.141592653589793 / 4), 0.0f};
        s = scene_alloc(1, accel);
        s->prims[0] = mk_smooth(OP_SMOOTH_SUB, mk_wrap(OP_ROUND, mk_box(0, 0, 0, 1, 1, 1, rot), 0.1),
                                mk_sphere(0, 0, 0, 0.9, NULL), 0.2);
        break;
    }
    case 12: /* "Smooth Union [A]" */
        s = scene_alloc(1, accel);
        s->prims[0] = mk_smooth(OP_SMOOTH_UNION, mk_anim(mk_sphere(0, 0, 0, 1, NULL), 1, 0, 0, 3.0, 0.005),
                                mk_sphere(0, 0, 0, 1, NULL), 0.2);
        break;
    case 13: /* "Mandelbulb [A]" */
        s = scene_alloc(1, accel);
        s->prims[0] = mk_mandelbulb(0, 0, 0, 8, 80, 1, -0.0001, NULL);
        break;
    case 14: { /* "Twisted Torus" */
        const float rot[3] = {f32(-3.141592653589793 / 2), 0.0f, 0.0f};
        s = scene_alloc(1, accel);
        s->prims[0] = mk_wrap(OP_TWIST, mk_torus(0, 0, 0, 1.3, rot), 3);
        break;
    }
    case 15: /* "Infinite Spheres" */
        s = scene_alloc(1, accel);
        s->prims[0] = mk_repetition(mk_sphere(0, 0, 0, 0.3, NULL), 1.5, 1.5, 1.5);
        break;
    case 16: /* "Screw" */
        s = scene_alloc(1, accel);
        s->prims[0] = mk_wrap(OP_ROUND, mk_wrap(OP_TWIST, mk_box(0, 0, 0, 0.4, 1.5, 0.4, NULL), 4.0), 0.1);
        break;
    case 17: { /* "Chicken": a left-nested chain of nine smooth unions over ten boxes */
        static const double b[10][6] = {
            {0, 0, 0, 0.6, 0.6, 0.8}, {0, -0.2, 0, 0.8, 0.4, 0.6}, {0, -0.8, 0.8, 0.4, 0.6, 0.3},
            {0, -0.8, 1.2, 0.4, 0.2, 0.2}, {0, -0.4, 1.0, 0.2, 0.2, 0.2}, {0.3, 1, 0, 0.1, 0.6, 0.01},
            {-0.3, 1, 0, 0.1, 0.6, 0.01}, {0, 1.6, 0.2, 0.6, 0.01, 0.2}, {0.3, 1.6, 0.5, 0.1, 0.01, 0.1},
            {-0.3, 1.6, 0.5, 0.1, 0.01, 0.1}};
        s = scene_alloc(1, accel);
        Prim acc = mk_box(b[0][0], b[0][1], b[0][2], b[0][3], b[0][4], b[0][5], NULL);
        for (int i = 1; i < 10; i++)
            acc = mk_smooth(OP_SMOOTH_UNION, acc, mk_box(b[i][0], b[i][1], b[i][2], b[i][3], b[i][4], b[i][5], NULL), 0.0001);
        s->prims[0] = acc;
        break;
    }
    default: { /* 18: "67" */
        const double PI = 3.141592653589793;
        const float r5[3] = {0.0f, 0.0f, f32(PI / 5)}, r7[3] = {0.0f, 0.0f, f32(PI / 7)}, r2[3] = {0.0f, 0.0f, f32(PI / 2)};
        const float rt[3] = {f32(-PI / 2), 0.0f, 0.0f};
        s = scene_alloc(2, accel);
        s->prims[0] = mk_smooth(OP_SMOOTH_UNION,
                                mk_wrap(OP_ROUND, mk_box(-1.25, -0.8, 0, 0.05, 0.7, 0.05, r5), 0.20),
                                mk_wrap(OP_ROUND, mk_torus(-1.25, 0.5, 0, 0.8, rt), 0.05), 0.0001);
        s->prims[1] = mk_smooth(OP_SMOOTH_UNION,
                                mk_wrap(OP_ROUND, mk_box(1.35, 0, 0, 0.05, 1.5, 0.05, r7), 0.20),
                                mk_wrap(OP_ROUND, mk_box(1.25, -1.4, 0, 0.05, 0.8, 0.05, r2), 0.20), 0.0001);
        break;
    }
    }
    scene_build_accel(s);
    return s;
}

/* Build-defined entry for synthetic scenes: spheres placed with
 * createSphere(x, y, z, r) semantics, x/y/z given as doubles. */
ro_scene *ro_scene_from_spheres(const double *xyz, const double *radii, int n, const char *accel) {
    ro_scene *s = scene_alloc(n, accel);
    for (int i = 0; i < n; i++) make_sphere(&s->prims[i], xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], radii[i]);
    scene_build_accel(s);
    return s;
}

/* Build-defined generic entry for tests: n primitives, each described by 11 doubles
 * {type, x, y, z, rotX, rotY, rotZ (NaN rotX = no rotation argument), p0, p1, p2, unused}:
 * sphere p0 = radius; box p0..p2 = halfSize; torus p0 = radius (minor = radius / 4). */
ro_scene *ro_scene_from_prims(const double *desc, int n, const char *accel) {
    ro_scene *s = scene_alloc(n, accel);
    for (int i = 0; i < n; i++) {
        const double *d = desc + 11 * i;
        float rot[3] = {f32(d[4]), f32(d[5]), f32(d[6])};
        const float *r = (d[4] != d[4]) ? NULL : rot;
        int type = (int)d[0];
        if (type == PRIM_BOX) make_box(&s->prims[i], d[1], d[2], d[3], d[7], d[8], d[9], r);
        else if (type == PRIM_TORUS) make_torus(&s->prims[i], d[1], d[2], d[3], d[7], r);
        else {
            get_transform(s->prims[i].transform, d[1], d[2], d[3], r);
            s->prims[i].type = PRIM_SPHERE;
            s->prims[i].radius = d[7];
        }
    }
    scene_build_accel(s);
    return s;
}

/* Build-defined generic entry for tests: an expression forest.  16 doubles per node:
 * {type, x, y, z, rotX, rotY, rotZ (NaN rotX = no rotation argument), p0..p5, a, b, unused};
 * leaves take position/rotation and p0.. as in ro_scene_from_prims (Mandelbulb: power,
 * iterations, enableAnimation, animationSpeed); operators take operand node indices a (and b):
 * Round p0 = radius; SmoothUnion/SmoothSubtraction p0 = smoothness; Twist p0 = amount;
 * Repetition p0..p2 = spacing; AnimatedTranslate p0..p2 = direction, p3 = amplitude, p4 = speed.
 * roots[] lists the nodes that are Scene.objects, in order. */
static Prim node_build(const double *desc, int idx) {
    const double *d = desc + 16 * idx;
    float rot[3] = {f32(d[4]), f32(d[5]), f32(d[6])};
    const float *r = (d[4] != d[4]) ? NULL : rot;
    const int type = (int)d[0];
    switch (type) {
    case PRIM_BOX: return mk_box(d[1], d[2], d[3], d[7], d[8], d[9], r);
    case PRIM_TORUS: return mk_torus(d[1], d[2], d[3], d[7], r);
    case PRIM_MANDELBULB: return mk_mandelbulb(d[1], d[2], d[3], d[7], (int)d[8], d[9] != 0, d[10], r);
    case OP_ROUND: case OP_TWIST: return mk_wrap(type, node_build(desc, (int)d[13]), d[7]);
    case OP_REPETITION: return mk_repetition(node_build(desc, (int)d[13]), d[7], d[8], d[9]);
    case OP_ANIM: return mk_anim(node_build(desc, (int)d[13]), d[7], d[8], d[9], d[10], d[11]);
    case OP_SMOOTH_UNION: case OP_SMOOTH_SUB:
        return mk_smooth(type, node_build(desc, (int)d[13]), node_build(desc, (int)d[14]), d[7]);
    default: return mk_sphere(d[1], d[2], d[3], d[7], r);
    }
}
ro_scene *ro_scene_from_nodes(const double *desc, int n_nodes, const int *roots, int n_roots, const char *accel) {
    (void)n_nodes;
    ro_scene *s = scene_alloc(n_roots, accel);
    for (int i = 0; i < n_roots; i++) s->prims[i] = node_build(desc, roots[i]);
    scene_build_accel(s);
    return s;
}

/* Flattened expression forest as the product's rm_scene_from_nodes takes it: per node type,
 * operand indices, the world->local matrix (16 f32) and 6 doubles of parameters (layout of
 * rm_node.params in include/rm_raymarch.h).  Returns the node count; call with NULL buffers to size. */
static int flatten(const Prim *p, int *types, int *kids, float *transforms, double *params, int *n) {
    int ia = -1, ib = -1;
    if (p->a) ia = flatten(p->a, types, kids, transforms, params, n);
    if (p->b) ib = flatten(p->b, types, kids, transforms, params, n);
    const int me = (*n)++;
    if (types) {
        types[me] = p->type; kids[2 * me] = ia; kids[2 * me + 1] = ib;
        memcpy(transforms + 16 * me, p->transform, 16 * sizeof(float));
        double *q = params + 6 * me;
        for (int i = 0; i < 6; i++) q[i] = 0;
        switch (p->type) {
        case PRIM_SPHERE: q[0] = p->radius; break;
        case PRIM_BOX: q[0] = p->halfSize[0]; q[1] = p->halfSize[1]; q[2] = p->halfSize[2]; break;
        case PRIM_TORUS: q[0] = p->majorRadius; q[1] = p->minorRadius; break;
        case PRIM_MANDELBULB: q[0] = p->power; q[1] = p->iterations; q[2] = p->enableAnimation; q[3] = p->speed; break;
        case OP_REPETITION: q[0] = p->vec[0]; q[1] = p->vec[1]; q[2] = p->vec[2]; break;
        case OP_ANIM: q[0] = p->vec[0]; q[1] = p->vec[1]; q[2] = p->vec[2]; q[3] = p->k; q[4] = p->speed; break;
        default: q[0] = p->k; break;
        }
    }
    return me;
}
int ro_scene_nodes(const ro_scene *s, int *types, int *kids, float *transforms, double *params, int *roots) {
    int n = 0;
    for (int i = 0; i < s->n; i++) {
        int r = flatten(&s->prims[i], types, kids, transforms, params, &n);
        if (roots) roots[i] = r;
    }
    return n;
}

/* ro_jsmath.h entry for tests: fn 0 sin, 1 cos, 2 atan2(a,b), 3 asin, 4 log, 5 pow(a,b), 6 round, 7 atan */
void ro_jsmath_eval(int fn, const double *a, const double *b, double *out, long n) {
    for (long i = 0; i < n; ++i) {
        switch (fn) {
        case 0: out[i] = js_sin(a[i]); break;
        case 1: out[i] = js_cos(a[i]); break;
        case 2: out[i] = js_atan2(a[i], b[i]); break;
        case 3: out[i] = js_asin(a[i]); break;
        case 4: out[i] = js_log(a[i]); break;
        case 5: out[i] = js_pow(a[i], b[i]); break;
        case 6: out[i] = js_round(a[i]); break;
        default: out[i] = js_atan(a[i]); break;
        }
    }
}

/* Scene.updateTime for ro_scene_distance (renders take `time` as an argument) */
void ro_set_time(double t) { g_time = t; }

/* per primitive: type, world->local transform (16 f32), params (3 doubles) -- what the product's
 * rm_scene_from_prims takes */
void ro_scene_prims(const ro_scene *s, int *types, float *transforms, double *params) {
    for (int i = 0; i < s->n; i++) {
        const Prim *p = &s->prims[i];
        types[i] = p->type;
        memcpy(transforms + 16 * i, p->transform, 16 * sizeof(float));
        if (p->type == PRIM_BOX) { params[3 * i] = p->halfSize[0]; params[3 * i + 1] = p->halfSize[1]; params[3 * i + 2] = p->halfSize[2]; }
        else if (p->type == PRIM_TORUS) { params[3 * i] = p->majorRadius; params[3 * i + 1] = p->minorRadius; params[3 * i + 2] = 0; }
        else { params[3 * i] = p->radius; params[3 * i + 1] = 0; params[3 * i + 2] = 0; }
    }
}

void ro_scene_free(ro_scene *s) {
    if (!s) return;
    bvh_free(s->bvh); oct_free(s->octree);
    for (int i = 0; i < s->n; i++) prim_free_children(&s->prims[i]);
    free(s->primBounds); free(s->prims); free(s);
}

/* per-call scratch (the reference allocates per step; we reuse) */
typedef struct {
    int *cand;             /* n */
    Interval *intervals;   /* bvh_leaves */
    const BVHNode **stack; /* bvh_nodes */
} Scratch;

/* scene.ts:144-190 getDistance */
static double scene_get_distance(const ro_scene *s, const float *pos, uint32_t *count, Scratch *sc) {
    const double MAX_DIST = 10;
    double closest = MAX_DIST;
    if (s->accel == ACCEL_OCTREE && s->octree) {
        const OctNode *node = oct_find(s->octree, pos);
        if (node) {
            if (node->nprims > 0) {
                for (int i = 0; i < node->nprims; i++) {
                    (*count)++;
                    closest = js_min(prim_sdf(&s->prims[node->prims[i]], pos), closest);
                }
            } else if (node->isEmpty) {
                const double safety = 0.99;
                closest = js_min(closest, node->minDistance * safety);
            }
            return closest;
        }
    } else if (s->accel == ACCEL_BVH && s->bvh) {
        int nc = 0;
        bvh_query(s->bvh, pos, sc->cand, &nc);
        if (nc == 0) { /* scene.ts:173 fallback: all primitives */
            for (int i = 0; i < s->n; i++) {
                (*count)++;
                closest = js_min(prim_sdf(&s->prims[i], pos), closest);
            }
        } else {
            for (int i = 0; i < nc; i++) {
                (*count)++;
                closest = js_min(prim_sdf(&s->prims[sc->cand[i]], pos), closest);
            }
        }
        return closest;
    }
    for (int i = 0; i < s->n; i++) {
        (*count)++;
        closest = js_min(prim_sdf(&s->prims[i], pos), closest);
    }
    return closest;
}

/* exported for unit tests */
double ro_scene_distance(const ro_scene *s, const float *pos, uint32_t *count) {
    Scratch sc;
    sc.cand = (int *)malloc((size_t)(s->n ? s->n : 1) * sizeof(int));
    sc.intervals = NULL; sc.stack = NULL;
    uint32_t c = 0;
    double d = scene_get_distance(s, pos, &c, &sc);
    if (count) *count = c;
    free(sc.cand);
    return d;
}

/* raymarcher.ts:111-121 getSceneDistance: u16 += wraps */
static inline double get_scene_distance(const ro_scene *s, const float *pos, uint16_t *sdfCell, Scratch *sc) {
    uint32_t c = 0;
    double d = scene_get_distance(s, pos, &c, sc);
    *sdfCell = (uint16_t)(((uint32_t)*sdfCell + c) & 0xFFFFu);
    return d;
}

#define MAX_DIST 10.0
#define EPSILON 0.001

/* The accel prologue and step callback every marcher shares (sphereTracer.ts:26-64,
 * fixedStep.ts:33-72, adaptiveStep.ts:34-72, adaptiveStepV2.ts:37-76, adaptiveStepV3.ts:34-69). */
typedef struct {
    int haveState;  /* accelState truthy */
    int nIntervals, curIdx;
} AccelState;

/* onRayMarchStart; returns 0 when the structure says "terminate" (bvh.ts:190-192) */
static int accel_start(const ro_scene *s, const float *origin, const float *dir, Scratch *sc, AccelState *st) {
    st->haveState = 0; st->nIntervals = 0; st->curIdx = 0;
    if (s->accel == ACCEL_BVH && s->bvh) {
        st->nIntervals = bvh_find_intervals(s, origin, dir, 0, MAX_DIST, sc->intervals, sc->stack);
        if (st->nIntervals == 0) return 0;
        st->haveState = 1;
    } else if (s->accel == ACCEL_OCTREE && s->octree) {
        st->haveState = 1; /* {data: null} is truthy, octree.ts:281-284 */
    }
    return 1;
}

/* onRayMarchStep: 0 = march normally, > 0 = skip, -1 = terminate */
static double accel_step(const ro_scene *s, const float *origin, const float *dir, double totalDist,
                         Scratch *sc, AccelState *st) {
    if (!st->haveState) return 0;
    if (s->accel == ACCEL_BVH) { /* bvh.ts:204-240 */
        if (st->curIdx >= st->nIntervals) return -1;
        const Interval *cur = &sc->intervals[st->curIdx];
        if (totalDist < cur->tEnter) return cur->tEnter - totalDist;
        if (totalDist > cur->tExit) {
            st->curIdx++;
            if (st->curIdx < st->nIntervals) {
                const Interval *nx = &sc->intervals[st->curIdx];
                if (nx->tEnter > totalDist) return nx->tEnter - totalDist;
            } else return -1;
        }
        return 0;
    }
    return oct_marchRay(s, origin, dir, totalDist); /* octree.ts:286-294 */
}

/* sphereTracer.ts:15-83 rayMarch (MAX_STEPS = 100) */
static double sphere_tracer_march(const ro_scene *s, const float *origin, const float *dir,
                                  uint16_t *sdfCell, uint16_t *iterCell, Scratch *sc) {
    double totalDist = 0;
    AccelState st;
    if (!accel_start(s, origin, dir, sc, &st)) return MAX_DIST;
    for (int i = 0; i < 100; i++) {
        float p[3];
        vec3_scaleAndAdd(p, origin, dir, totalDist);
        double skip = accel_step(s, origin, dir, totalDist, sc, &st);
        if (skip == -1) return MAX_DIST;
        else if (skip > 0) {
            totalDist += skip;
            if (totalDist > MAX_DIST) break;
            continue;
        }
        double dist = get_scene_distance(s, p, sdfCell, sc);
        totalDist += dist;
        *iterCell = (uint16_t)(*iterCell + 1);
        if (dist < EPSILON) break;
        if (totalDist > MAX_DIST) break;
    }
    return totalDist;
}

/* fixedStep.ts:21-94 (MAX_STEPS = 200; returns MAX_DIST unless it hit) */
static double fixed_step_march(const ro_scene *s, const float *origin, const float *dir, uint16_t *sdfCell,
                               uint16_t *iterCell, Scratch *sc, double stepSize) {
    double totalDist = 0;
    int hit = 0;
    AccelState st;
    if (!accel_start(s, origin, dir, sc, &st)) return MAX_DIST;
    for (int i = 0; i < 200; i++) {
        float p[3];
        vec3_scaleAndAdd(p, origin, dir, totalDist);
        double skip = accel_step(s, origin, dir, totalDist, sc, &st);
        if (skip == -1) return MAX_DIST;
        else if (skip > 0) {
            totalDist += skip;
            if (totalDist > MAX_DIST) break;
            continue;
        }
        double dist = get_scene_distance(s, p, sdfCell, sc);
        *iterCell = (uint16_t)(*iterCell + 1);
        if (dist < EPSILON) { hit = 1; break; }
        totalDist += stepSize;
        if (totalDist > MAX_DIST) break;
    }
    return hit ? totalDist : MAX_DIST;
}

/* adaptiveStep.ts:22-105 (MAX_STEPS = 200) */
static double adaptive_step_march(const ro_scene *s, const float *origin, const float *dir, uint16_t *sdfCell,
                                  uint16_t *iterCell, Scratch *sc) {
    const double FIXED_STEP_SIZE = 0.1, STEP_SCALE = 0.8;
    const double MIN_STEP = FIXED_STEP_SIZE * 0.25, MAX_STEP = FIXED_STEP_SIZE * 5.0;
    const double NEAR_DIST = 0.1, NEAR_STEP = 0.01;
    double totalDist = 0;
    int hit = 0;
    AccelState st;
    if (!accel_start(s, origin, dir, sc, &st)) return MAX_DIST;
    for (int i = 0; i < 200; i++) {
        float p[3];
        vec3_scaleAndAdd(p, origin, dir, totalDist);
        double skip = accel_step(s, origin, dir, totalDist, sc, &st);
        if (skip == -1) return MAX_DIST;
        else if (skip > 0) {
            totalDist += skip;
            if (totalDist > MAX_DIST) break;
            continue;
        }
        double dist = get_scene_distance(s, p, sdfCell, sc);
        *iterCell = (uint16_t)(*iterCell + 1);
        if (dist < EPSILON) { hit = 1; break; }
        double step;
        if (dist < NEAR_DIST) step = NEAR_STEP;
        else {
            step = STEP_SCALE * dist;
            if (step < MIN_STEP) step = MIN_STEP;
            if (step > MAX_STEP) step = MAX_STEP;
        }
        totalDist += step;
        if (totalDist > MAX_DIST) break;
    }
    return hit ? totalDist : MAX_DIST;
}

/* adaptiveStepV2.ts:22-124 (v3 == 0) and adaptiveStepV3.ts:22-137 (v3 == 1); MAX_STEPS = 100 */
static double adaptive_v23_march(const ro_scene *s, const float *origin, const float *dir, uint16_t *sdfCell,
                                 uint16_t *iterCell, Scratch *sc, double overshootFactor, int v3) {
    double totalDist = 0, prevSDF = 0, prevStep = 0;
    AccelState st;
    if (!accel_start(s, origin, dir, sc, &st)) return MAX_DIST;
    for (int i = 0; i < 100; i++) {
        float p[3];
        vec3_scaleAndAdd(p, origin, dir, totalDist);
        double skip = accel_step(s, origin, dir, totalDist, sc, &st);
        if (skip == -1) return MAX_DIST;
        else if (skip > 0) {
            totalDist += skip;
            if (totalDist > MAX_DIST) break;
            prevSDF = 0;
            prevStep = 0;
            continue;
        }
        double newSDF = get_scene_distance(s, p, sdfCell, sc);
        *iterCell = (uint16_t)(*iterCell + 1);
        if (newSDF < EPSILON) break;
        if (totalDist > MAX_DIST) break;
        if (i == 0 || prevSDF == 0) {
            double step = newSDF;
            totalDist += step;
            prevSDF = newSDF;
            prevStep = step;
            continue;
        }
        int spheresOverlapped = prevStep <= (prevSDF + newSDF);
        if (spheresOverlapped) {
            double step = newSDF * overshootFactor;
            totalDist += step;
            prevSDF = newSDF;
            prevStep = step;
            continue;
        }
        if (!v3) { /* adaptiveStepV2.ts:108-114 */
            totalDist -= prevStep;
            totalDist += prevSDF;
            prevStep = prevSDF;
            continue;
        }
        /* adaptiveStepV3.ts:103-130 */
        double originalPos = totalDist - prevStep;
        totalDist = originalPos + prevSDF;
        vec3_scaleAndAdd(p, origin, dir, totalDist);
        double d3 = get_scene_distance(s, p, sdfCell, sc);
        *iterCell = (uint16_t)(*iterCell + 1);
        if (prevSDF + newSDF + d3 >= prevStep) {
            totalDist = originalPos + prevStep + newSDF;
            prevSDF = newSDF;
            prevStep = newSDF;
            continue;
        }
        prevSDF = d3;
        prevStep = d3;
        totalDist += d3;
    }
    return totalDist;
}

/* raymarcher.ts:123-135 getNormal */
static void get_normal(const ro_scene *s, const float *pos, uint16_t *sdfCell, Scratch *sc, float *n) {
    double d = get_scene_distance(s, pos, sdfCell, sc);
    const double e0 = 0.01;
    float q[3];
    q[0] = f32((double)pos[0] - e0); q[1] = pos[1]; q[2] = pos[2];
    n[0] = f32(d - get_scene_distance(s, q, sdfCell, sc));
    q[0] = pos[0]; q[1] = f32((double)pos[1] - e0); q[2] = pos[2];
    n[1] = f32(d - get_scene_distance(s, q, sdfCell, sc));
    q[0] = pos[0]; q[1] = pos[1]; q[2] = f32((double)pos[2] - e0);
    n[2] = f32(d - get_scene_distance(s, q, sdfCell, sc));
    vec3_normalize(n, n);
}

/* raymarchWorker.ts:50-68: 0 = sphere tracer (also the default), >0 = other marchers */
static int parse_algorithm(const char *a) {
    if (!a) return 0;
    if (strcmp(a, "fixed-step") == 0) return 1;
    if (strcmp(a, "adaptive-step") == 0) return 2;
    if (strcmp(a, "adaptive-step-v2") == 0) return 3;
    if (strcmp(a, "adaptive-step-v3") == 0) return 4;
    return 0;
}

/* raymarcher.ts:46-109 runRaymarcher (+ raymarchWorker.ts:33-92 algorithm pick).
 * overshootFactor / stepSize: NaN stands for JS `undefined` (constructor defaults 1.2 / 0.1,
 * adaptiveStepV2.ts:13, fixedStep.ts:13). */
int ro_run_raymarcher_ex(const ro_scene *s, const char *algorithm, uint8_t *depthBuffer,
                         uint8_t *normalBuffer, uint16_t *sdfBuffer, uint16_t *iterBuffer,
                         int width, int height, double time, int yStart, int yEnd,
                         double overshootFactor, double stepSize) {
    g_time = time; /* raymarcher.ts:58-59 scene.updateTime(time) */
    const int alg = parse_algorithm(algorithm);
    if (overshootFactor != overshootFactor) overshootFactor = 1.2;
    if (stepSize != stepSize) stepSize = 0.1;
    /* raymarcher.ts:62-67 */
    float rotMat3[9];
    const float *ct = s->cameraTransform;
    /* camera.ts:38-44 getRotationMatrix, then mat3.fromMat4 */
    rotMat3[0] = ct[0]; rotMat3[1] = ct[1]; rotMat3[2] = ct[2];
    rotMat3[3] = ct[4]; rotMat3[4] = ct[5]; rotMat3[5] = ct[6];
    rotMat3[6] = ct[8]; rotMat3[7] = ct[9]; rotMat3[8] = ct[10];
    float rayOrigin[3] = {ct[12], ct[13], ct[14]};

    Scratch sc;
    sc.cand = (int *)malloc((size_t)(s->n ? s->n : 1) * sizeof(int));
    sc.intervals = (Interval *)malloc((size_t)(s->bvh_leaves ? s->bvh_leaves : 1) * sizeof(Interval));
    sc.stack = (const BVHNode **)malloc((size_t)(s->bvh_nodes ? s->bvh_nodes : 1) * sizeof(BVHNode *));

    for (int y = yStart; y < yEnd; y++) {
        int localY = y - yStart;
        double v = ((double)y / (double)height - 0.5) * 2.0;
        for (int x = 0; x < width; x++) {
            size_t idx = (size_t)localY * (size_t)width + (size_t)x;
            size_t nIdx = idx * 3;
            sdfBuffer[idx] = 0;
            iterBuffer[idx] = 0;
            double u = ((double)x / (double)width - 0.5) * 2.0;
            float rayDir[3] = {f32(u), f32(v), -1.0f};
            vec3_transformMat3(rayDir, rayDir, rotMat3);
            vec3_normalize(rayDir, rayDir);

            double depth;
            uint16_t *sc_ = &sdfBuffer[idx], *ic_ = &iterBuffer[idx];
            switch (alg) {
            case 1: depth = fixed_step_march(s, rayOrigin, rayDir, sc_, ic_, &sc, stepSize); break;
            case 2: depth = adaptive_step_march(s, rayOrigin, rayDir, sc_, ic_, &sc); break;
            case 3: depth = adaptive_v23_march(s, rayOrigin, rayDir, sc_, ic_, &sc, overshootFactor, 0); break;
            case 4: depth = adaptive_v23_march(s, rayOrigin, rayDir, sc_, ic_, &sc, overshootFactor, 1); break;
            default: depth = sphere_tracer_march(s, rayOrigin, rayDir, sc_, ic_, &sc);
            }

            float hit[3];
            vec3_scaleAndAdd(hit, rayOrigin, rayDir, depth);
            float normal[3] = {0, 0, 0};
            if (!(depth >= MAX_DIST)) get_normal(s, hit, &sdfBuffer[idx], &sc, normal);
            normalBuffer[nIdx] = ro_u8clamp(((double)normal[0] + 1) * 0.5 * 255);
            normalBuffer[nIdx + 1] = ro_u8clamp(((double)normal[1] + 1) * 0.5 * 255);
            normalBuffer[nIdx + 2] = ro_u8clamp(((double)normal[2] + 1) * 0.5 * 255);
            depthBuffer[idx] = ro_u8clamp(depth);
        }
    }
    free(sc.cand); free(sc.intervals); free((void *)sc.stack);
    return 0;
}

int ro_run_raymarcher(const ro_scene *s, const char *algorithm, uint8_t *depthBuffer,
                      uint8_t *normalBuffer, uint16_t *sdfBuffer, uint16_t *iterBuffer,
                      int width, int height, double time, int yStart, int yEnd) {
    return ro_run_raymarcher_ex(s, algorithm, depthBuffer, normalBuffer, sdfBuffer, iterBuffer, width, height, time,
                                yStart, yEnd, NAN, NAN);
}

/* ------------------------------------------------------------------------- */
/* Shading (shading_models/, all four models) and diagnostics (main.ts:528-548)           */
/* ------------------------------------------------------------------------- */

/* main.ts:33-45: 'phong' | 'sdf-heatmap' | 'iteration-heatmap' | default normal */
void ro_shade(const char *model, uint8_t *shaded, const uint8_t *depthBuffer,
              const uint8_t *normalBuffer, const uint16_t *sdfBuffer, const uint16_t *iterBuffer,
              int width, int height) {
    size_t npx = (size_t)width * (size_t)height;
    if (model && (strcmp(model, "sdf-heatmap") == 0 || strcmp(model, "iteration-heatmap") == 0)) {
        /* SDFHeatmap.ts:5-34 / IterationHeatmap.ts:5-34 */
        const uint16_t *src = strcmp(model, "sdf-heatmap") == 0 ? sdfBuffer : iterBuffer;
        for (size_t i = 0; i < npx; i++) {
            double inten = fmod((double)src[i] * 5, 256);
            shaded[4 * i + 0] = ro_u8clamp(js_min(2 * inten, 255));
            shaded[4 * i + 1] = ro_u8clamp(js_min(-2 * inten + 512, 255));
            shaded[4 * i + 2] = 0;
            shaded[4 * i + 3] = 255;
        }
        return;
    }
    if (model && strcmp(model, "phong") == 0) {
        /* phongModel.ts:6-77 */
        float lightDir[3] = {f32(1), f32(-1), f32(1.5)};
        vec3_normalize(lightDir, lightDir);
        float viewDir[3] = {0, 0, 1};
        const double ambient = 0.1, specularStrength = 0.5, shininess = 32;
        for (size_t i = 0; i < npx; i++) {
            double depth = depthBuffer[i];
            if (depth >= 255) {
                shaded[4 * i + 0] = 10; shaded[4 * i + 1] = 10; shaded[4 * i + 2] = 20; shaded[4 * i + 3] = 255;
                continue;
            }
            float normal[3], refl[3];
            normal[0] = f32(normalBuffer[3 * i] / 127.5 - 1.0);
            normal[1] = f32(normalBuffer[3 * i + 1] / 127.5 - 1.0);
            normal[2] = f32(normalBuffer[3 * i + 2] / 127.5 - 1.0);
            vec3_normalize(normal, normal);
            double diffuse = js_max(vec3_dot(normal, lightDir), 0);
            double sc2 = 2 * vec3_dot(normal, lightDir);
            refl[0] = f32(normal[0] * sc2); refl[1] = f32(normal[1] * sc2); refl[2] = f32(normal[2] * sc2);
            refl[0] = f32((double)refl[0] - lightDir[0]);
            refl[1] = f32((double)refl[1] - lightDir[1]);
            refl[2] = f32((double)refl[2] - lightDir[2]);
            vec3_normalize(refl, refl);
            double specular = specularStrength * pow(js_max(vec3_dot(viewDir, refl), 0), shininess);
            double intensity = js_min(ambient + diffuse + specular, 1);
            double depthFactor = 1 - depth / 255;
            double color = 255 * intensity * depthFactor;
            uint8_t c = ro_u8clamp(color);
            shaded[4 * i + 0] = c; shaded[4 * i + 1] = c; shaded[4 * i + 2] = c; shaded[4 * i + 3] = 255;
        }
        return;
    }
    /* normalModel.ts:6-29 */
    for (size_t i = 0; i < npx; i++) {
        shaded[4 * i + 0] = normalBuffer[3 * i];
        shaded[4 * i + 1] = normalBuffer[3 * i + 1];
        shaded[4 * i + 2] = normalBuffer[3 * i + 2];
        shaded[4 * i + 3] = 255;
    }
}

/* main.ts:528-548: out = {totalSDFCalls, maxSDFCalls, minSDFCalls, totalIterations} */
void ro_diagnostics(const uint16_t *sdfBuffer, const uint16_t *iterBuffer, size_t totalPixels, double *out) {
    double total = 0, totalIt = 0, mx = 0, mn = 9007199254740991.0;
    for (size_t i = 0; i < totalPixels; i++) {
        double c = sdfBuffer[i];
        total += c;
        totalIt += iterBuffer[i];
        if (c > mx) mx = c;
        if (c < mn) mn = c;
    }
    out[0] = total; out[1] = mx; out[2] = mn; out[3] = totalIt;
}

/* ------------------------------------------------------------------------- */
/* Introspection for tests                                                   */
/* ------------------------------------------------------------------------- */

/* out: [bvh_nodes, bvh_leaves, bvh_depth, oct_nodes, oct_leaves, oct_empty, oct_maxleafprims, n] */
void ro_scene_stats(const ro_scene *s, int *out) {
    out[0] = s->bvh_nodes; out[1] = s->bvh_leaves; out[2] = s->bvh_depth;
    out[3] = s->oct_nodes; out[4] = s->oct_leaves; out[5] = s->oct_empty;
    out[6] = s->oct_maxleafprims; out[7] = s->n;
}

/* root bounds of the active accel structure (6 floats), returns 0 if none */
int ro_scene_root_bounds(const ro_scene *s, float *out) {
    const BBox *b = s->bvh ? &s->bvh->bounds : (s->octree ? &s->octree->bounds : NULL);
    if (!b) return 0;
    memcpy(out, b->min, 12); memcpy(out + 3, b->max, 12);
    return 1;
}

/* camera: rot 3x3 (column-major as mat3.fromMat4) + origin */
void ro_scene_camera(const ro_scene *s, float *rot9, float *origin3) {
    const float *ct = s->cameraTransform;
    rot9[0] = ct[0]; rot9[1] = ct[1]; rot9[2] = ct[2];
    rot9[3] = ct[4]; rot9[4] = ct[5]; rot9[5] = ct[6];
    rot9[6] = ct[8]; rot9[7] = ct[9]; rot9[8] = ct[10];
    origin3[0] = ct[12]; origin3[1] = ct[13]; origin3[2] = ct[14];
}

/* sphere list as the product boundary takes it: centre f32x3 (= world position), radius f64 */
void ro_scene_spheres(const ro_scene *s, float *centers, double *radii) {
    for (int i = 0; i < s->n; i++) {
        prim_world_position(&s->prims[i], centers + 3 * i);
        radii[i] = s->prims[i].radius;
    }
}

/* histogram of per-ray BVH interval-list lengths (design aid for the GPU kernel's list
 * capacity; not part of the reference's path).  hist has nbins entries; last bin = overflow */
void ro_debug_interval_hist(const ro_scene *s, int width, int height, long *hist, int nbins) {
    if (!s->bvh) return;
    const float *ct = s->cameraTransform;
    float rotMat3[9] = {ct[0], ct[1], ct[2], ct[4], ct[5], ct[6], ct[8], ct[9], ct[10]};
    float rayOrigin[3] = {ct[12], ct[13], ct[14]};
    Interval *iv = (Interval *)malloc((size_t)s->bvh_leaves * sizeof(Interval));
    const BVHNode **stack = (const BVHNode **)malloc((size_t)s->bvh_nodes * sizeof(BVHNode *));
    for (int y = 0; y < height; y++) {
        double v = ((double)y / (double)height - 0.5) * 2.0;
        for (int x = 0; x < width; x++) {
            double u = ((double)x / (double)width - 0.5) * 2.0;
            float rayDir[3] = {f32(u), f32(v), -1.0f};
            vec3_transformMat3(rayDir, rayDir, rotMat3);
            vec3_normalize(rayDir, rayDir);
            int n = bvh_find_intervals(s, rayOrigin, rayDir, 0, MAX_DIST, iv, stack);
            hist[n < nbins - 1 ? n : nbins - 1]++;
        }
    }
    free(iv); free((void *)stack);
}
