// rm_jsmath.h -- JS `Math.sin / cos / atan2 / asin / log / pow / round` for the device.
//
// The reference's operator and Mandelbulb SDFs call them per point (twist.ts:24-25,
// animatedTranslate.ts:36, mandelbulb.ts:50-77, repetition.ts:22-24).  V8 computes them with ports
// of Sun's fdlibm 5.3 (src/base/ieee754.cc: k_sin.c, k_cos.c, e_rem_pio2.c, s_atan.c, e_atan2.c,
// e_asin.c, e_log.c, e_pow.c).  The same published algorithms are restated here for gfx950: plain
// FP64 VALU arithmetic, one rounding per operation (-ffp-contract=off), IEEE division and square
// root, the word-level tests done on the bit pattern.  `rm_selftest_jsmath` evaluates them on the
// GPU for the parity tests.  Argument reduction is complete (medium path up to 2^19*pi/2, Payne-Hanek
// beyond; the latter is an out-of-line function with its tables in scratch, reached only by huge arguments).
//
// The algorithms and the polynomial / table constants restated below are those of fdlibm 5.3, which carries this notice:
//
//   ====================================================
//   Copyright (C) 1993-2004 by Sun Microsystems, Inc. All rights reserved.
//
//   Developed at SunSoft, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

namespace rmd {

__device__ inline uint64_t jm_bits(double x) { return __builtin_bit_cast(uint64_t, x); }
__device__ inline double jm_from(uint64_t u) { return __builtin_bit_cast(double, u); }
__device__ inline int32_t jm_hi(double x) { return static_cast<int32_t>(jm_bits(x) >> 32); }
__device__ inline uint32_t jm_lo(double x) { return static_cast<uint32_t>(jm_bits(x)); }
__device__ inline double jm_set_hi(double x, int32_t hi) { return jm_from((static_cast<uint64_t>(static_cast<uint32_t>(hi)) << 32) | jm_lo(x)); }
__device__ inline double jm_set_lo(double x, uint32_t lo) { return jm_from((jm_bits(x) & 0xFFFFFFFF00000000ull) | lo); }
__device__ inline double jm_words(int32_t hi, uint32_t lo) { return jm_from((static_cast<uint64_t>(static_cast<uint32_t>(hi)) << 32) | lo); }

/* k_sin.c */
__device__ inline double jm_ksin(double x, double y, int iy) {
    constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                        S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                        S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    int32_t ix = jm_hi(x) & 0x7fffffff;
    if (ix < 0x3e400000) { if ((int)x == 0) return x; }
    double z = x * x, v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (iy == 0) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

/* k_cos.c */
__device__ inline double jm_kcos(double x, double y) {
    constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                        C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                        C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    int32_t ix = jm_hi(x) & 0x7fffffff;
    if (ix < 0x3e400000) { if ((int)x == 0) return 1.0; }
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    if (ix < 0x3FD33333) return 1.0 - (0.5 * z - (z * r - x * y));
    double qx;
    if (ix > 0x3fe90000) qx = 0.28125;
    else qx = jm_words(ix - 0x00200000, 0);
    double hz = 0.5 * z - qx, a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}

/* k_rem_pio2.c: Payne-Hanek reduction for |x| > 2^19*pi/2.  x[0..nx) are 24-bit pieces of |x| scaled by
 * 2^-e0; ipio2 holds 2/pi in 24-bit pieces (1584 bits, generated from pi with integer arithmetic and
 * equal to the published table).  prec = 2 (double): jk = 4. */
__device__ __attribute__((noinline)) int jm_kernel_rem_pio2(const double *x, double *y, int e0, int nx) {
    constexpr int32_t ipio2[66] = {
        0xA2F983, 0x6E4E44, 0x1529FC, 0x2757D1, 0xF534DD, 0xC0DB62, 0x95993C, 0x439041,
        0xFE5163, 0xABDEBB, 0xC561B7, 0x246E3A, 0x424DD2, 0xE00649, 0x2EEA09, 0xD1921C,
        0xFE1DEB, 0x1CB129, 0xA73EE8, 0x8235F5, 0x2EBB44, 0x84E99C, 0x7026B4, 0x5F7E41,
        0x3991D6, 0x398353, 0x39F49C, 0x845F8B, 0xBDF928, 0x3B1FF8, 0x97FFDE, 0x05980F,
        0xEF2F11, 0x8B5A0A, 0x6D1F6D, 0x367ECF, 0x27CB09, 0xB74F46, 0x3F669E, 0x5FEA2D,
        0x7527BA, 0xC7EBE5, 0xF17B3D, 0x0739F7, 0x8A5292, 0xEA6BFB, 0x5FB11F, 0x8D5D08,
        0x560330, 0x46FC7B, 0x6BABF0, 0xCFBC20, 0x9AF436, 0x1DA9E3, 0x91615E, 0xE61B08,
        0x659985, 0x5F14A0, 0x68408D, 0xFFD880, 0x4D7327, 0x310606, 0x1556CA, 0x73A8C9,
        0x60E27B, 0xC08C6B};
    constexpr double PIo2[8] = {1.57079625129699707031e+00, 7.54978941586159635335e-08, 5.39030252995776476554e-15,
                                   3.28200341580791294123e-22, 1.27065575308067607349e-29, 1.22933308981111328932e-36,
                                   2.73370053816464559624e-44, 2.16741683877804819444e-51};
    const double two24 = 1.67772160000000000000e+07, twon24 = 5.96046447753906250000e-08;
    int32_t jz, jx, jv, jp, jk, carry, n, iq[20], i, j, k, m, q0, ih;
    double z, fw, f[20], fq[20], q[20];
    jk = 4;
    jp = jk;
    jx = nx - 1;
    jv = (e0 - 3) / 24;
    if (jv < 0) jv = 0;
    q0 = e0 - 24 * (jv + 1);
    j = jv - jx;
    m = jx + jk;
    for (i = 0; i <= m; i++, j++) f[i] = (j < 0) ? 0.0 : (double)ipio2[j];
    for (i = 0; i <= jk; i++) {
        for (j = 0, fw = 0.0; j <= jx; j++) fw += x[j] * f[jx + i - j];
        q[i] = fw;
    }
    jz = jk;
    for (;;) {  /* recompute: */
        for (i = 0, j = jz, z = q[jz]; j > 0; i++, j--) {
            fw = (double)((int32_t)(twon24 * z));
            iq[i] = (int32_t)(z - two24 * fw);
            z = q[j - 1] + fw;
        }
        z = __builtin_ldexp(z, q0);
        z -= 8.0 * __builtin_floor(z * 0.125);
        n = (int32_t)z;
        z -= (double)n;
        ih = 0;
        if (q0 > 0) {
            i = (iq[jz - 1] >> (24 - q0));
            n += i;
            iq[jz - 1] -= i << (24 - q0);
            ih = iq[jz - 1] >> (23 - q0);
        } else if (q0 == 0) ih = iq[jz - 1] >> 23;
        else if (z >= 0.5) ih = 2;
        if (ih > 0) {
            n += 1;
            carry = 0;
            for (i = 0; i < jz; i++) {
                j = iq[i];
                if (carry == 0) {
                    if (j != 0) { carry = 1; iq[i] = 0x1000000 - j; }
                } else iq[i] = 0xffffff - j;
            }
            if (q0 > 0) {
                switch (q0) {
                case 1: iq[jz - 1] &= 0x7fffff; break;
                case 2: iq[jz - 1] &= 0x3fffff; break;
                }
            }
            if (ih == 2) {
                z = 1.0 - z;
                if (carry != 0) z -= __builtin_ldexp(1.0, q0);
            }
        }
        if (z == 0.0) {
            j = 0;
            for (i = jz - 1; i >= jk; i--) j |= iq[i];
            if (j == 0) { /* need recomputation */
                for (k = 1; iq[jk - k] == 0; k++) {}
                for (i = jz + 1; i <= jz + k; i++) {
                    f[jx + i] = (double)ipio2[jv + i];
                    for (j = 0, fw = 0.0; j <= jx; j++) fw += x[j] * f[jx + i - j];
                    q[i] = fw;
                }
                jz += k;
                continue;
            }
        }
        break;
    }
    if (z == 0.0) {
        jz -= 1;
        q0 -= 24;
        while (iq[jz] == 0) { jz--; q0 -= 24; }
    } else {
        z = __builtin_ldexp(z, -q0);
        if (z >= two24) {
            fw = (double)((int32_t)(twon24 * z));
            iq[jz] = (int32_t)(z - two24 * fw);
            jz += 1;
            q0 += 24;
            iq[jz] = (int32_t)fw;
        } else iq[jz] = (int32_t)z;
    }
    fw = __builtin_ldexp(1.0, q0);
    for (i = jz; i >= 0; i--) { q[i] = fw * (double)iq[i]; fw *= twon24; }
    for (i = jz; i >= 0; i--) {
        for (fw = 0.0, k = 0; k <= jp && k <= jz - i; k++) fw += PIo2[k] * q[i + k];
        fq[jz - i] = fw;
    }
    fw = 0.0;
    for (i = jz; i >= 0; i--) fw += fq[i];
    y[0] = (ih == 0) ? fw : -fw;
    fw = fq[0] - fw;
    for (i = 1; i <= jz; i++) fw += fq[i];
    y[1] = (ih == 0) ? fw : -fw;
    return n & 7;
}

/* e_rem_pio2.c; returns n, y[0]+y[1] = x - n*pi/2 (*ok is always 1 since the large path exists) */
__device__ inline int jm_rem_pio2(double x, double *y, int *ok) {
    constexpr int32_t npio2_hw[32] = {
        0x3FF921FB, 0x400921FB, 0x4012D97C, 0x401921FB, 0x401F6A7A, 0x4022D97C, 0x4025FDBB, 0x402921FB,
        0x402C463A, 0x402F6A7A, 0x4031475C, 0x4032D97C, 0x40346B9C, 0x4035FDBB, 0x40378FDB, 0x403921FB,
        0x403AB41B, 0x403C463A, 0x403DD85A, 0x403F6A7A, 0x40407E4C, 0x4041475C, 0x4042106C, 0x4042D97C,
        0x4043A28C, 0x40446B9C, 0x404534AC, 0x4045FDBB, 0x4046C6CB, 0x40478FDB, 0x404858EB, 0x404921FB};
    constexpr double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                        pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
                        pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
                        pio2_3t = 8.47842766036889956997e-32;
    int32_t hx = jm_hi(x), ix = hx & 0x7fffffff;
    *ok = 1;
    if (ix <= 0x3fe921fb) { y[0] = x; y[1] = 0; return 0; }
    if (ix < 0x4002d97c) {
        double z;
        if (hx > 0) {
            z = x - pio2_1;
            if (ix != 0x3ff921fb) { y[0] = z - pio2_1t; y[1] = (z - y[0]) - pio2_1t; }
            else { z -= pio2_2; y[0] = z - pio2_2t; y[1] = (z - y[0]) - pio2_2t; }
            return 1;
        }
        z = x + pio2_1;
        if (ix != 0x3ff921fb) { y[0] = z + pio2_1t; y[1] = (z - y[0]) + pio2_1t; }
        else { z += pio2_2; y[0] = z + pio2_2t; y[1] = (z - y[0]) + pio2_2t; }
        return -1;
    }
    if (ix <= 0x413921fb) {
        double t = __builtin_fabs(x);
        int n = (int)(t * invpio2 + 0.5);
        double fn = (double)n;
        double r = t - fn * pio2_1, w = fn * pio2_1t;
        if (n < 32 && ix != npio2_hw[n - 1]) {
            y[0] = r - w;
        } else {
            int j = ix >> 20;
            y[0] = r - w;
            int i = j - ((jm_hi(y[0]) >> 20) & 0x7ff);
            if (i > 16) {
                t = r; w = fn * pio2_2; r = t - w; w = fn * pio2_2t - ((t - r) - w); y[0] = r - w;
                i = j - ((jm_hi(y[0]) >> 20) & 0x7ff);
                if (i > 49) { t = r; w = fn * pio2_3; r = t - w; w = fn * pio2_3t - ((t - r) - w); y[0] = r - w; }
            }
        }
        y[1] = (r - y[0]) - w;
        if (hx < 0) { y[0] = -y[0]; y[1] = -y[1]; return -n; }
        return n;
    }
    if (ix >= 0x7ff00000) { y[0] = y[1] = x - x; return 0; }
    {   /* set z = scalbn(|x|, ilogb(x) - 23), split it into three 24-bit pieces */
        const int e0 = (ix >> 20) - 1046;
        double z = jm_words(ix - (int32_t)((uint32_t)e0 << 20), jm_lo(x));
        double tx[3];
        for (int i = 0; i < 2; i++) {
            tx[i] = (double)((int32_t)z);
            z = (z - tx[i]) * 1.67772160000000000000e+07;
        }
        tx[2] = z;
        int nx = 3;
        while (tx[nx - 1] == 0.0) nx--;
        const int n = jm_kernel_rem_pio2(tx, y, e0, nx);
        if (hx < 0) { y[0] = -y[0]; y[1] = -y[1]; return -n; }
        return n;
    }
}

__device__ inline double js_sin(double x) {
    int32_t ix = jm_hi(x) & 0x7fffffff;
    if (ix <= 0x3fe921fb) return jm_ksin(x, 0.0, 0);
    if (ix >= 0x7ff00000) return x - x;
    double y[2]; int ok; int n = jm_rem_pio2(x, y, &ok);
    if (!ok) return __builtin_nan("");
    switch (n & 3) {
        case 0: return jm_ksin(y[0], y[1], 1);
        case 1: return jm_kcos(y[0], y[1]);
        case 2: return -jm_ksin(y[0], y[1], 1);
        default: return -jm_kcos(y[0], y[1]);
    }
}

__device__ inline double js_cos(double x) {
    int32_t ix = jm_hi(x) & 0x7fffffff;
    if (ix <= 0x3fe921fb) return jm_kcos(x, 0.0);
    if (ix >= 0x7ff00000) return x - x;
    double y[2]; int ok; int n = jm_rem_pio2(x, y, &ok);
    if (!ok) return __builtin_nan("");
    switch (n & 3) {
        case 0: return jm_kcos(y[0], y[1]);
        case 1: return -jm_ksin(y[0], y[1], 1);
        case 2: return -jm_kcos(y[0], y[1]);
        default: return jm_ksin(y[0], y[1], 1);
    }
}

/* s_atan.c */
__device__ inline double js_atan(double x) {
    const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01,
                                     9.82793723247329054082e-01, 1.57079632679489655800e+00};
    const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17,
                                     1.39033110312309984516e-17, 6.12323399573676603587e-17};
    const double aT[11] = {3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01,
                                  -1.11111104054623557880e-01, 9.09088713343650656196e-02, -7.69187620504482999495e-02,
                                  6.66107313738753120669e-02, -5.83357013379057348645e-02, 4.97687799461593236017e-02,
                                  -3.65315727442169155270e-02, 1.62858201153657823623e-02};
    int32_t hx = jm_hi(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x44100000) {
        if (ix > 0x7ff00000 || (ix == 0x7ff00000 && jm_lo(x) != 0)) return x + x;
        if (hx > 0) return atanhi[3] + atanlo[3];
        return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3fdc0000) {
        if (ix < 0x3e200000) { if (1.0e300 + x > 1.0) return x; }
        id = -1;
    } else {
        x = __builtin_fabs(x);
        if (ix < 0x3ff30000) {
            if (ix < 0x3fe60000) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); }
            else { id = 1; x = (x - 1.0) / (x + 1.0); }
        } else {
            if (ix < 0x40038000) { id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); }
            else { id = 3; x = -1.0 / x; }
        }
    }
    double z = x * x, w = z * z;
    double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -z : z;
}

/* e_atan2.c */
__device__ inline double js_atan2(double y, double x) {
    constexpr double tiny = 1.0e-300, pi_o_4 = 7.8539816339744827900E-01, pi_o_2 = 1.5707963267948965580E+00,
                        pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
    int32_t hx = jm_hi(x), ix = hx & 0x7fffffff, hy = jm_hi(y), iy = hy & 0x7fffffff;
    uint32_t lx = jm_lo(x), ly = jm_lo(y);
    if (((uint32_t)ix | ((lx | (0u - lx)) >> 31)) > 0x7ff00000u || ((uint32_t)iy | ((ly | (0u - ly)) >> 31)) > 0x7ff00000u)
        return x + y;
    if (((hx - 0x3ff00000) | (int32_t)lx) == 0) return js_atan(y);
    int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if ((iy | (int32_t)ly) == 0) {
        switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; }
    }
    if ((ix | (int32_t)lx) == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7ff00000) {
        if (iy == 0x7ff00000) {
            switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny;
                         case 2: return 3.0 * pi_o_4 + tiny; default: return -3.0 * pi_o_4 - tiny; }
        } else {
            switch (m) { case 0: return 0.0; case 1: return -0.0; case 2: return pi + tiny; default: return -pi - tiny; }
        }
    }
    if (iy == 0x7ff00000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int k = (iy - ix) >> 20;
    double z;
    if (k > 60) { z = pi_o_2 + 0.5 * pi_lo; m &= 1; }
    else if (hx < 0 && k < -60) z = 0.0;
    else z = js_atan(__builtin_fabs(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

/* e_asin.c */
__device__ inline double js_asin(double x) {
    constexpr double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                        pio4_hi = 7.85398163397448278999e-01,
                        pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                        pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                        qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                        qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    int32_t hx = jm_hi(x), ix = hx & 0x7fffffff;
    double t = 0.0, w, p, q, c, r, s;
    if (ix >= 0x3ff00000) {
        if (((ix - 0x3ff00000) | (int32_t)jm_lo(x)) == 0) return x * pio2_hi + x * pio2_lo;
        return (x - x) / (x - x);
    } else if (ix < 0x3fe00000) {
        if (ix < 0x3e400000) { if (1.0e300 + x > 1.0) return x; }
        else t = x * x;
        p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
        q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
        w = p / q;
        return x + x * w;
    }
    w = 1.0 - __builtin_fabs(x);
    t = w * 0.5;
    p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
    q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
    s = __builtin_sqrt(t);
    if (ix >= 0x3FEF3333) {
        w = p / q;
        t = pio2_hi - (2.0 * (s + s * w) - pio2_lo);
    } else {
        w = jm_set_lo(s, 0);
        c = (t - w * w) / (s + w);
        r = p / q;
        p = 2.0 * s * r - (pio2_lo - 2.0 * c);
        q = pio4_hi - 2.0 * w;
        t = pio4_hi - (p - q);
    }
    return (hx > 0) ? t : -t;
}

/* e_log.c */
__device__ inline double js_log(double x) {
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        two54 = 1.80143985094819840000e+16,
                        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    int32_t hx = jm_hi(x), k = 0, i, j;
    uint32_t lx = jm_lo(x);
    double f, s, z, R, w, t1, t2, dk, hfsq;
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | (int32_t)lx) == 0) return -__builtin_inf();
        if (hx < 0) return __builtin_nan("");
        k -= 54; x *= two54; hx = jm_hi(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    i = (hx + 0x95f64) & 0x100000;
    x = jm_set_hi(x, hx | (i ^ 0x3ff00000));
    k += (i >> 20);
    f = x - 1.0;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.0) {
            if (k == 0) return 0.0;
            dk = (double)k; return dk * ln2_hi + dk * ln2_lo;
        }
        R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        dk = (double)k; return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    s = f / (2.0 + f);
    dk = (double)k;
    z = s * s;
    i = hx - 0x6147a;
    w = z * z;
    j = 0x6b851 - hx;
    t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    R = t2 + t1;
    if (i > 0) {
        hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* e_pow.c */
__device__ inline double js_pow(double x, double y) {
    const double bp[2] = {1.0, 1.5}, dp_h[2] = {0.0, 5.84962487220764160156e-01},
                 dp_l[2] = {0.0, 1.35003920212974897128e-08};
    constexpr double two53 = 9007199254740992.0, huge = 1.0e300, tiny = 1.0e-300,
                        L1 = 5.99999999999994648725e-01, L2 = 4.28571428578550184252e-01, L3 = 3.33333329818377432918e-01,
                        L4 = 2.72728123808534006489e-01, L5 = 2.30660745775561754067e-01, L6 = 2.06975017800338417784e-01,
                        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08,
                        lg2 = 6.93147180559945286227e-01, lg2_h = 6.93147182464599609375e-01, lg2_l = -1.90465429995776804525e-09,
                        ovt = 8.0085662595372944372e-0017, cp = 9.61796693925975554329e-01, cp_h = 9.61796700954437255859e-01,
                        cp_l = -7.02846165095275826516e-09, ivln2 = 1.44269504088896338700e+00,
                        ivln2_h = 1.44269502162933349609e+00, ivln2_l = 1.92596299112661746887e-08;
    double z, ax, z_h, z_l, p_h, p_l, y1, t1, t2, r, s, t, u, v, w;
    int32_t i, j, k, yisint, n;
    int32_t hx = jm_hi(x), hy = jm_hi(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    uint32_t lx = jm_lo(x), ly = jm_lo(y);
    if ((iy | (int32_t)ly) == 0) return 1.0;
    if (ix > 0x7ff00000 || (ix == 0x7ff00000 && lx != 0) || iy > 0x7ff00000 || (iy == 0x7ff00000 && ly != 0)) return x + y;
    yisint = 0;
    if (hx < 0) {
        if (iy >= 0x43400000) yisint = 2;
        else if (iy >= 0x3ff00000) {
            k = (iy >> 20) - 0x3ff;
            if (k > 20) {
                uint32_t jj = ly >> (52 - k);
                if ((jj << (52 - k)) == ly) yisint = 2 - (int32_t)(jj & 1);
            } else if (ly == 0) {
                j = iy >> (20 - k);
                if ((j << (20 - k)) == iy) yisint = 2 - (j & 1);
            }
        }
    }
    if (ly == 0) {
        if (iy == 0x7ff00000) {
            if (((ix - 0x3ff00000) | (int32_t)lx) == 0) return y - y;
            else if (ix >= 0x3ff00000) return (hy >= 0) ? y : 0.0;
            else return (hy < 0) ? -y : 0.0;
        }
        if (iy == 0x3ff00000) { if (hy < 0) return 1.0 / x; else return x; }
        if (hy == 0x40000000) return x * x;
        if (hy == 0x3fe00000) { if (hx >= 0) return __builtin_sqrt(x); }
    }
    ax = __builtin_fabs(x);
    if (lx == 0) {
        if (ix == 0x7ff00000 || ix == 0 || ix == 0x3ff00000) {
            z = ax;
            if (hy < 0) z = 1.0 / z;
            if (hx < 0) {
                if (((ix - 0x3ff00000) | yisint) == 0) z = (z - z) / (z - z);
                else if (yisint == 1) z = -z;
            }
            return z;
        }
    }
    n = (hx >> 31) + 1;
    if ((n | yisint) == 0) return (x - x) / (x - x);
    s = 1.0;
    if ((n | (yisint - 1)) == 0) s = -1.0;
    if (iy > 0x41e00000) {
        if (iy > 0x43f00000) {
            if (ix <= 0x3fefffff) return (hy < 0) ? huge * huge : tiny * tiny;
            if (ix >= 0x3ff00000) return (hy > 0) ? huge * huge : tiny * tiny;
        }
        if (ix < 0x3fefffff) return (hy < 0) ? s * huge * huge : s * tiny * tiny;
        if (ix > 0x3ff00000) return (hy > 0) ? s * huge * huge : s * tiny * tiny;
        t = ax - 1.0;
        w = (t * t) * (0.5 - t * (0.3333333333333333333333 - t * 0.25));
        u = ivln2_h * t;
        v = t * ivln2_l - w * ivln2;
        t1 = jm_set_lo(u + v, 0);
        t2 = v - (t1 - u);
    } else {
        double ss, s2, s_h, s_l, t_h, t_l;
        n = 0;
        if (ix < 0x00100000) { ax *= two53; n -= 53; ix = jm_hi(ax); }
        n += (ix >> 20) - 0x3ff;
        j = ix & 0x000fffff;
        ix = j | 0x3ff00000;
        if (j <= 0x3988E) k = 0;
        else if (j < 0xBB67A) k = 1;
        else { k = 0; n += 1; ix -= 0x00100000; }
        ax = jm_set_hi(ax, ix);
        u = ax - bp[k];
        v = 1.0 / (ax + bp[k]);
        ss = u * v;
        s_h = jm_set_lo(ss, 0);
        t_h = jm_words(((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18), 0);
        t_l = ax - (t_h - bp[k]);
        s_l = v * ((u - s_h * t_h) - s_h * t_l);
        s2 = ss * ss;
        r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
        r += s_l * (s_h + ss);
        s2 = s_h * s_h;
        t_h = jm_set_lo(3.0 + s2 + r, 0);
        t_l = r - ((t_h - 3.0) - s2);
        u = s_h * t_h;
        v = s_l * t_h + t_l * ss;
        p_h = jm_set_lo(u + v, 0);
        p_l = v - (p_h - u);
        z_h = cp_h * p_h;
        z_l = cp_l * p_h + p_l * cp + dp_l[k];
        t = (double)n;
        t1 = jm_set_lo(((z_h + z_l) + dp_h[k]) + t, 0);
        t2 = z_l - (((t1 - t) - dp_h[k]) - z_h);
    }
    y1 = jm_set_lo(y, 0);
    p_l = (y - y1) * t1 + y * t2;
    p_h = y1 * t1;
    z = p_l + p_h;
    j = jm_hi(z);
    i = (int32_t)jm_lo(z);
    if (j >= 0x40900000) {
        if (((j - 0x40900000) | i) != 0) return s * huge * huge;
        if (p_l + ovt > z - p_h) return s * huge * huge;
    } else if ((j & 0x7fffffff) >= 0x4090cc00) {
        if (((j - (int32_t)0xc090cc00) | i) != 0) return s * tiny * tiny;
        if (p_l <= z - p_h) return s * tiny * tiny;
    }
    i = j & 0x7fffffff;
    k = (i >> 20) - 0x3ff;
    n = 0;
    if (i > 0x3fe00000) {
        n = j + (0x00100000 >> (k + 1));
        k = ((n & 0x7fffffff) >> 20) - 0x3ff;
        t = jm_words(n & ~(0x000fffff >> k), 0);
        n = ((n & 0x000fffff) | 0x00100000) >> (20 - k);
        if (j < 0) n = -n;
        p_h -= t;
    }
    t = jm_set_lo(p_l + p_h, 0);
    u = t * lg2_h;
    v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
    z = u + v;
    w = v - (z - u);
    t = z * z;
    t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    r = (z * t1) / (t1 - 2.0) - (w + z * w);
    z = 1.0 - (r - z);
    j = jm_hi(z);
    j += (int32_t)((uint32_t)n << 20);
    if ((j >> 20) <= 0) z = __builtin_ldexp(z, n);
    else z = jm_set_hi(z, j);
    return s * z;
}

/* The two halves of js_pow for an ordinary base (finite, > 0, not 1) and an ordinary exponent (finite, not 0, +-1, 2,
 * 0.5, |y| <= 2^31), as separate functions with the identical arithmetic: the Mandelbulb raises the same r to
 * power - 1 and to power (mandelbulb.ts:61-64), and the extended-precision log2 of the base is the larger half. */
__device__ inline void jm_pow_log2(double ax, int32_t ix, double &t1, double &t2) {
    const double bp[2] = {1.0, 1.5}, dp_h[2] = {0.0, 5.84962487220764160156e-01},
                 dp_l[2] = {0.0, 1.35003920212974897128e-08};
    constexpr double two53 = 9007199254740992.0, huge = 1.0e300, tiny = 1.0e-300,
                        L1 = 5.99999999999994648725e-01, L2 = 4.28571428578550184252e-01, L3 = 3.33333329818377432918e-01,
                        L4 = 2.72728123808534006489e-01, L5 = 2.30660745775561754067e-01, L6 = 2.06975017800338417784e-01,
                        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08,
                        lg2 = 6.93147180559945286227e-01, lg2_h = 6.93147182464599609375e-01, lg2_l = -1.90465429995776804525e-09,
                        ovt = 8.0085662595372944372e-0017, cp = 9.61796693925975554329e-01, cp_h = 9.61796700954437255859e-01,
                        cp_l = -7.02846165095275826516e-09, ivln2 = 1.44269504088896338700e+00,
                        ivln2_h = 1.44269502162933349609e+00, ivln2_l = 1.92596299112661746887e-08;
    double z_h, z_l, p_h, p_l, r, t, u, v;
    int32_t j, k, n;
    (void)huge; (void)tiny; (void)P1; (void)P2; (void)P3; (void)P4; (void)P5; (void)lg2; (void)lg2_h; (void)lg2_l; (void)ovt;
    (void)ivln2; (void)ivln2_h; (void)ivln2_l;
    {
        double ss, s2, s_h, s_l, t_h, t_l;
        n = 0;
        if (ix < 0x00100000) { ax *= two53; n -= 53; ix = jm_hi(ax); }
        n += (ix >> 20) - 0x3ff;
        j = ix & 0x000fffff;
        ix = j | 0x3ff00000;
        if (j <= 0x3988E) k = 0;
        else if (j < 0xBB67A) k = 1;
        else { k = 0; n += 1; ix -= 0x00100000; }
        ax = jm_set_hi(ax, ix);
        u = ax - bp[k];
        v = 1.0 / (ax + bp[k]);
        ss = u * v;
        s_h = jm_set_lo(ss, 0);
        t_h = jm_words(((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18), 0);
        t_l = ax - (t_h - bp[k]);
        s_l = v * ((u - s_h * t_h) - s_h * t_l);
        s2 = ss * ss;
        r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
        r += s_l * (s_h + ss);
        s2 = s_h * s_h;
        t_h = jm_set_lo(3.0 + s2 + r, 0);
        t_l = r - ((t_h - 3.0) - s2);
        u = s_h * t_h;
        v = s_l * t_h + t_l * ss;
        p_h = jm_set_lo(u + v, 0);
        p_l = v - (p_h - u);
        z_h = cp_h * p_h;
        z_l = cp_l * p_h + p_l * cp + dp_l[k];
        t = (double)n;
        t1 = jm_set_lo(((z_h + z_l) + dp_h[k]) + t, 0);
        t2 = z_l - (((t1 - t) - dp_h[k]) - z_h);
    }
}

__device__ inline double jm_pow_exp2(double y, double t1, double t2) {
    const double bp[2] = {1.0, 1.5}, dp_h[2] = {0.0, 5.84962487220764160156e-01},
                 dp_l[2] = {0.0, 1.35003920212974897128e-08};
    constexpr double two53 = 9007199254740992.0, huge = 1.0e300, tiny = 1.0e-300,
                        L1 = 5.99999999999994648725e-01, L2 = 4.28571428578550184252e-01, L3 = 3.33333329818377432918e-01,
                        L4 = 2.72728123808534006489e-01, L5 = 2.30660745775561754067e-01, L6 = 2.06975017800338417784e-01,
                        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08,
                        lg2 = 6.93147180559945286227e-01, lg2_h = 6.93147182464599609375e-01, lg2_l = -1.90465429995776804525e-09,
                        ovt = 8.0085662595372944372e-0017, cp = 9.61796693925975554329e-01, cp_h = 9.61796700954437255859e-01,
                        cp_l = -7.02846165095275826516e-09, ivln2 = 1.44269504088896338700e+00,
                        ivln2_h = 1.44269502162933349609e+00, ivln2_l = 1.92596299112661746887e-08;
    const double s = 1.0;
    double z, p_h, p_l, y1, r, t, u, v, w;
    int32_t i, j, k, n;
    (void)bp; (void)dp_h; (void)dp_l; (void)two53; (void)L1; (void)L2; (void)L3; (void)L4; (void)L5; (void)L6; (void)cp; (void)cp_h;
    (void)cp_l; (void)ivln2; (void)ivln2_h; (void)ivln2_l;
    y1 = jm_set_lo(y, 0);
    p_l = (y - y1) * t1 + y * t2;
    p_h = y1 * t1;
    z = p_l + p_h;
    j = jm_hi(z);
    i = (int32_t)jm_lo(z);
    if (j >= 0x40900000) {
        if (((j - 0x40900000) | i) != 0) return s * huge * huge;
        if (p_l + ovt > z - p_h) return s * huge * huge;
    } else if ((j & 0x7fffffff) >= 0x4090cc00) {
        if (((j - (int32_t)0xc090cc00) | i) != 0) return s * tiny * tiny;
        if (p_l <= z - p_h) return s * tiny * tiny;
    }
    i = j & 0x7fffffff;
    k = (i >> 20) - 0x3ff;
    n = 0;
    if (i > 0x3fe00000) {
        n = j + (0x00100000 >> (k + 1));
        k = ((n & 0x7fffffff) >> 20) - 0x3ff;
        t = jm_words(n & ~(0x000fffff >> k), 0);
        n = ((n & 0x000fffff) | 0x00100000) >> (20 - k);
        if (j < 0) n = -n;
        p_h -= t;
    }
    t = jm_set_lo(p_l + p_h, 0);
    u = t * lg2_h;
    v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
    z = u + v;
    w = v - (z - u);
    t = z * z;
    t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    r = (z * t1) / (t1 - 2.0) - (w + z * w);
    z = 1.0 - (r - z);
    j = jm_hi(z);
    j += (int32_t)((uint32_t)n << 20);
    if ((j >> 20) <= 0) z = __builtin_ldexp(z, n);
    else z = jm_set_hi(z, j);
    return s * z;
}

__device__ inline bool jm_pow_ordinary_exponent(double y) {
    const int32_t hy = jm_hi(y), iy = hy & 0x7fffffff;
    const uint32_t ly = jm_lo(y);
    if ((iy | (int32_t)ly) == 0 || iy >= 0x7ff00000 || iy > 0x41e00000) return false;
    if (ly == 0 && (iy == 0x3ff00000 || hy == 0x40000000 || hy == 0x3fe00000)) return false;
    return true;
}

/* ra = Math.pow(x, ya), rb = Math.pow(x, yb) */
__device__ inline void js_pow_pair(double x, double ya, double yb, double &ra, double &rb) {
    const int32_t hx = jm_hi(x), ix = hx & 0x7fffffff;
    const bool base_ok = hx > 0 && ix < 0x7ff00000 && !(jm_lo(x) == 0 && ix == 0x3ff00000);
    if (base_ok && jm_pow_ordinary_exponent(ya) && jm_pow_ordinary_exponent(yb)) {
        double t1, t2;
        jm_pow_log2(x, ix, t1, t2);
        ra = jm_pow_exp2(ya, t1, t2);
        rb = jm_pow_exp2(yb, t1, t2);
        return;
    }
    ra = js_pow(x, ya);
    rb = js_pow(x, yb);
}

/* Math.sin(x) and Math.cos(x) with one argument reduction (same kernels, same values) */
__device__ inline void js_sincos(double x, double &sn, double &cs) {
    int32_t ix = jm_hi(x) & 0x7fffffff;
    if (ix <= 0x3fe921fb) {
        sn = jm_ksin(x, 0.0, 0);
        cs = jm_kcos(x, 0.0);
        return;
    }
    if (ix >= 0x7ff00000) {
        sn = cs = x - x;
        return;
    }
    double y[2]; int ok; int n = jm_rem_pio2(x, y, &ok);
    if (!ok) {
        sn = cs = __builtin_nan("");
        return;
    }
    const double ks = jm_ksin(y[0], y[1], 1), kc = jm_kcos(y[0], y[1]);
    switch (n & 3) {
        case 0: sn = ks; cs = kc; break;
        case 1: sn = kc; cs = -ks; break;
        case 2: sn = -ks; cs = -kc; break;
        default: sn = -kc; cs = ks; break;
    }
}

/* Math.round: __builtin_floor(x + 0.5) with the ties and signed-zero rules of ECMA-262 21.3.2.28 */
__device__ inline double js_round(double x) {
    if (!(__builtin_fabs(x) < 4503599627370496.0)) return x; /* NaN, inf and integers beyond 2^52 */
    double f = __builtin_floor(x);
    double r = (x - f >= 0.5) ? f + 1.0 : f;
    if (r == 0.0) return (x < 0.0 || (x == 0.0 && __builtin_signbit(x))) ? -0.0 : 0.0; /* [-0.5, -0] -> -0 */
    return r;
}


}  // namespace rmd
