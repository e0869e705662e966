'use strict';
/*
 * glmatrix_shim.js -- the subset of gl-matrix 3.x the reference's render path calls (vec3 / mat3 / mat4, 25 functions),
 * RESTATED from the library's published algorithms because gl-matrix 3.4.4 (package.json:25) is not in this image and
 * there is no network.  It is a stand-in, not the library: the cross-check that uses it (run.js) exercises the
 * REFERENCE'S control flow, not gl-matrix's arithmetic, and does not pin parity (SURVEY 8c, DESIGN.md 5).
 * Float32Array-backed like the library's default ARRAY_TYPE; formulas as in SURVEY Appendix B, binary64 evaluation in
 * the order written, one binary32 rounding at each element store.  vec3.length: Math.hypot (3.0 - 3.4.3), or
 * Math.sqrt(x*x + y*y + z*z) when RM_XCHECK_LENGTH_SQRT=1.
 */
const SQRT_LEN = process.env.RM_XCHECK_LENGTH_SQRT === '1';
const vec3 = {
  create() { return new Float32Array(3); },
  fromValues(x, y, z) { const o = new Float32Array(3); o[0] = x; o[1] = y; o[2] = z; return o; },
  clone(a) { const o = new Float32Array(3); o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; return o; },
  copy(o, a) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; return o; },
  set(o, x, y, z) { o[0] = x; o[1] = y; o[2] = z; return o; },
  add(o, a, b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; return o; },
  subtract(o, a, b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; return o; },
  scale(o, a, s) { o[0] = a[0] * s; o[1] = a[1] * s; o[2] = a[2] * s; return o; },
  scaleAndAdd(o, a, b, s) { o[0] = a[0] + b[0] * s; o[1] = a[1] + b[1] * s; o[2] = a[2] + b[2] * s; return o; },
  dot(a, b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; },
  length(a) { const x = a[0], y = a[1], z = a[2]; return SQRT_LEN ? Math.sqrt(x * x + y * y + z * z) : Math.hypot(x, y, z); },
  distance(a, b) { const x = b[0] - a[0], y = b[1] - a[1], z = b[2] - a[2]; return SQRT_LEN ? Math.sqrt(x * x + y * y + z * z) : Math.hypot(x, y, z); },
  normalize(o, a) {
    const x = a[0], y = a[1], z = a[2];
    let len = x * x + y * y + z * z;
    if (len > 0) len = 1 / Math.sqrt(len);
    o[0] = a[0] * len; o[1] = a[1] * len; o[2] = a[2] * len;
    return o;
  },
  transformMat3(o, a, m) {
    const x = a[0], y = a[1], z = a[2];
    o[0] = x * m[0] + y * m[3] + z * m[6];
    o[1] = x * m[1] + y * m[4] + z * m[7];
    o[2] = x * m[2] + y * m[5] + z * m[8];
    return o;
  },
  transformMat4(o, a, m) {
    const x = a[0], y = a[1], z = a[2];
    let w = m[3] * x + m[7] * y + m[11] * z + m[15];
    w = w || 1.0;
    o[0] = (m[0] * x + m[4] * y + m[8] * z + m[12]) / w;
    o[1] = (m[1] * x + m[5] * y + m[9] * z + m[13]) / w;
    o[2] = (m[2] * x + m[6] * y + m[10] * z + m[14]) / w;
    return o;
  },
};
vec3.sub = vec3.subtract; vec3.len = vec3.length; vec3.dist = vec3.distance;

const mat3 = {
  create() { const o = new Float32Array(9); o[0] = o[4] = o[8] = 1; return o; },
  fromMat4(o, a) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[4]; o[4] = a[5]; o[5] = a[6]; o[6] = a[8]; o[7] = a[9]; o[8] = a[10]; return o; },
};

const mat4 = {
  create() { const o = new Float32Array(16); o[0] = o[5] = o[10] = o[15] = 1; return o; },
  copy(o, a) { for (let i = 0; i < 16; i++) o[i] = a[i]; return o; },
  fromTranslation(o, v) {
    o[0] = 1; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 0; o[5] = 1; o[6] = 0; o[7] = 0; o[8] = 0; o[9] = 0; o[10] = 1; o[11] = 0;
    o[12] = v[0]; o[13] = v[1]; o[14] = v[2]; o[15] = 1;
    return o;
  },
  fromRotationTranslationScale(o, q, v, s) {
    const x = q[0], y = q[1], z = q[2], w = q[3];
    const x2 = x + x, y2 = y + y, z2 = z + z;
    const xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2, wx = w * x2, wy = w * y2, wz = w * z2;
    const sx = s[0], sy = s[1], sz = s[2];
    o[0] = (1 - (yy + zz)) * sx; o[1] = (xy + wz) * sx; o[2] = (xz - wy) * sx; o[3] = 0;
    o[4] = (xy - wz) * sy; o[5] = (1 - (xx + zz)) * sy; o[6] = (yz + wx) * sy; o[7] = 0;
    o[8] = (xz + wy) * sz; o[9] = (yz - wx) * sz; o[10] = (1 - (xx + yy)) * sz; o[11] = 0;
    o[12] = v[0]; o[13] = v[1]; o[14] = v[2]; o[15] = 1;
    return o;
  },
  invert(o, a) {
    const a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7];
    const a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11], a30 = a[12], a31 = a[13], a32 = a[14], a33 = a[15];
    const b00 = a00 * a11 - a01 * a10, b01 = a00 * a12 - a02 * a10, b02 = a00 * a13 - a03 * a10, b03 = a01 * a12 - a02 * a11;
    const b04 = a01 * a13 - a03 * a11, b05 = a02 * a13 - a03 * a12, b06 = a20 * a31 - a21 * a30, b07 = a20 * a32 - a22 * a30;
    const b08 = a20 * a33 - a23 * a30, b09 = a21 * a32 - a22 * a31, b10 = a21 * a33 - a23 * a31, b11 = a22 * a33 - a23 * a32;
    let det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06;
    if (!det) return null;
    det = 1.0 / det;
    o[0] = (a11 * b11 - a12 * b10 + a13 * b09) * det;
    o[1] = (a02 * b10 - a01 * b11 - a03 * b09) * det;
    o[2] = (a31 * b05 - a32 * b04 + a33 * b03) * det;
    o[3] = (a22 * b04 - a21 * b05 - a23 * b03) * det;
    o[4] = (a12 * b08 - a10 * b11 - a13 * b07) * det;
    o[5] = (a00 * b11 - a02 * b08 + a03 * b07) * det;
    o[6] = (a32 * b02 - a30 * b05 - a33 * b01) * det;
    o[7] = (a20 * b05 - a22 * b02 + a23 * b01) * det;
    o[8] = (a10 * b10 - a11 * b08 + a13 * b06) * det;
    o[9] = (a01 * b08 - a00 * b10 - a03 * b06) * det;
    o[10] = (a30 * b04 - a31 * b02 + a33 * b00) * det;
    o[11] = (a21 * b02 - a20 * b04 - a23 * b00) * det;
    o[12] = (a11 * b07 - a10 * b09 - a12 * b06) * det;
    o[13] = (a00 * b09 - a01 * b07 + a02 * b06) * det;
    o[14] = (a31 * b01 - a30 * b03 - a32 * b00) * det;
    o[15] = (a20 * b03 - a21 * b01 + a22 * b00) * det;
    return o;
  },
  rotateX(o, a, rad) {
    const s = Math.sin(rad), c = Math.cos(rad);
    const a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7], a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11];
    if (a !== o) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[12] = a[12]; o[13] = a[13]; o[14] = a[14]; o[15] = a[15]; }
    o[4] = a10 * c + a20 * s; o[5] = a11 * c + a21 * s; o[6] = a12 * c + a22 * s; o[7] = a13 * c + a23 * s;
    o[8] = a20 * c - a10 * s; o[9] = a21 * c - a11 * s; o[10] = a22 * c - a12 * s; o[11] = a23 * c - a13 * s;
    return o;
  },
  rotateY(o, a, rad) {
    const s = Math.sin(rad), c = Math.cos(rad);
    const a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11];
    if (a !== o) { o[4] = a[4]; o[5] = a[5]; o[6] = a[6]; o[7] = a[7]; o[12] = a[12]; o[13] = a[13]; o[14] = a[14]; o[15] = a[15]; }
    o[0] = a00 * c - a20 * s; o[1] = a01 * c - a21 * s; o[2] = a02 * c - a22 * s; o[3] = a03 * c - a23 * s;
    o[8] = a00 * s + a20 * c; o[9] = a01 * s + a21 * c; o[10] = a02 * s + a22 * c; o[11] = a03 * s + a23 * c;
    return o;
  },
  rotateZ(o, a, rad) {
    const s = Math.sin(rad), c = Math.cos(rad);
    const a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7];
    if (a !== o) { o[8] = a[8]; o[9] = a[9]; o[10] = a[10]; o[11] = a[11]; o[12] = a[12]; o[13] = a[13]; o[14] = a[14]; o[15] = a[15]; }
    o[0] = a00 * c + a10 * s; o[1] = a01 * c + a11 * s; o[2] = a02 * c + a12 * s; o[3] = a03 * c + a13 * s;
    o[4] = a10 * c - a00 * s; o[5] = a11 * c - a01 * s; o[6] = a12 * c - a02 * s; o[7] = a13 * c - a03 * s;
    return o;
  },
  translate(o, a, v) {
    const x = v[0], y = v[1], z = v[2];
    if (a === o) {
      o[12] = a[0] * x + a[4] * y + a[8] * z + a[12]; o[13] = a[1] * x + a[5] * y + a[9] * z + a[13];
      o[14] = a[2] * x + a[6] * y + a[10] * z + a[14]; o[15] = a[3] * x + a[7] * y + a[11] * z + a[15];
    } else {
      const a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7], a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11];
      o[0] = a00; o[1] = a01; o[2] = a02; o[3] = a03; o[4] = a10; o[5] = a11; o[6] = a12; o[7] = a13; o[8] = a20; o[9] = a21; o[10] = a22; o[11] = a23;
      o[12] = a00 * x + a10 * y + a20 * z + a[12]; o[13] = a01 * x + a11 * y + a21 * z + a[13];
      o[14] = a02 * x + a12 * y + a22 * z + a[14]; o[15] = a03 * x + a13 * y + a23 * z + a[15];
    }
    return o;
  },
  scale(o, a, v) {
    const x = v[0], y = v[1], z = v[2];
    o[0] = a[0] * x; o[1] = a[1] * x; o[2] = a[2] * x; o[3] = a[3] * x; o[4] = a[4] * y; o[5] = a[5] * y; o[6] = a[6] * y; o[7] = a[7] * y;
    o[8] = a[8] * z; o[9] = a[9] * z; o[10] = a[10] * z; o[11] = a[11] * z; o[12] = a[12]; o[13] = a[13]; o[14] = a[14]; o[15] = a[15];
    return o;
  },
};

module.exports = { vec3, mat3, mat4 };
