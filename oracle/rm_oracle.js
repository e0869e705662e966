'use strict';
/*
 * rm_oracle.js -- second, independently written restatement of the reference's per-pixel
 * sphere-tracing path, for a real JS engine (node).
 *
 * TEST INFRASTRUCTURE ONLY.  It exists so that the C oracle (rm_oracle.c) can be
 * cross-checked, byte for byte, against the same algorithm evaluated with genuine JS
 * number semantics (Math.hypot, Math.min/max, Float32Array / Uint8ClampedArray /
 * Uint16Array stores, Array.prototype.sort stability), and to produce the fixtures under
 * tests/golden/ (tests/golden/make_golden.py drives it).  It never runs on the GPU box and
 * nothing in the product imports it.  PARITY UNPINNED (see rm_oracle.c header): the
 * gl-matrix helpers below restate the published 3.x formulas, gl-matrix itself is absent.
 *
 * Written in a different shape from the C file on purpose (flat node tables, iterative
 * point queries) so that agreement between the two means something.
 *
 * usage: node rm_oracle.js render <config.json> <outdir>
 *        node rm_oracle.js hypot  <in.f64 triples> <out.f64>
 *        node rm_oracle.js camera <in.f64 pitch,yaw pairs> <out.f32 x12>
 *        node rm_oracle.js jsmath <fn 0..8> <a.f64> <b.f64> <out.f64>   (the engine's Math.*; 8 = the fdlibm pow port)
 */
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');

const fr = Math.fround; // == one Float32Array element store

// ---------------------------------------------------------------- gl-matrix subset
function m4identity() { const m = new Float32Array(16); m[0] = m[5] = m[10] = m[15] = 1; return m; }

function m4fromRTS(q, v, s) { // mat4.fromRotationTranslationScale
  const o = new Float32Array(16);
  const x = q[0], y = q[1], z = q[2], w = q[3];
  const x2 = x + x, y2 = y + y, z2 = z + z;
  const xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2;
  const wx = w * x2, wy = w * y2, wz = w * z2;
  o[0] = (1 - (yy + zz)) * s[0]; o[1] = (xy + wz) * s[0]; o[2] = (xz - wy) * s[0]; o[3] = 0;
  o[4] = (xy - wz) * s[1]; o[5] = (1 - (xx + zz)) * s[1]; o[6] = (yz + wx) * s[1]; o[7] = 0;
  o[8] = (xz + wy) * s[2]; o[9] = (yz - wx) * s[2]; o[10] = (1 - (xx + yy)) * s[2]; o[11] = 0;
  o[12] = v[0]; o[13] = v[1]; o[14] = v[2]; o[15] = 1;
  return o;
}

function m4invert(out, a) { // mat4.invert, cofactor form; null when det is falsy
  const a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7];
  const a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11], a30 = a[12], a31 = a[13], a32 = a[14], a33 = a[15];
  const b00 = a00 * a11 - a01 * a10, b01 = a00 * a12 - a02 * a10, b02 = a00 * a13 - a03 * a10;
  const b03 = a01 * a12 - a02 * a11, b04 = a01 * a13 - a03 * a11, b05 = a02 * a13 - a03 * a12;
  const b06 = a20 * a31 - a21 * a30, b07 = a20 * a32 - a22 * a30, b08 = a20 * a33 - a23 * a30;
  const b09 = a21 * a32 - a22 * a31, b10 = a21 * a33 - a23 * a31, b11 = a22 * a33 - a23 * a32;
  let det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06;
  if (!det) return null;
  det = 1.0 / det;
  out[0] = (a11 * b11 - a12 * b10 + a13 * b09) * det;
  out[1] = (a02 * b10 - a01 * b11 - a03 * b09) * det;
  out[2] = (a31 * b05 - a32 * b04 + a33 * b03) * det;
  out[3] = (a22 * b04 - a21 * b05 - a23 * b03) * det;
  out[4] = (a12 * b08 - a10 * b11 - a13 * b07) * det;
  out[5] = (a00 * b11 - a02 * b08 + a03 * b07) * det;
  out[6] = (a32 * b02 - a30 * b05 - a33 * b01) * det;
  out[7] = (a20 * b05 - a22 * b02 + a23 * b01) * det;
  out[8] = (a10 * b10 - a11 * b08 + a13 * b06) * det;
  out[9] = (a01 * b08 - a00 * b10 - a03 * b06) * det;
  out[10] = (a30 * b04 - a31 * b02 + a33 * b00) * det;
  out[11] = (a21 * b02 - a20 * b04 - a23 * b00) * det;
  out[12] = (a11 * b07 - a10 * b09 - a12 * b06) * det;
  out[13] = (a00 * b09 - a01 * b07 + a02 * b06) * det;
  out[14] = (a31 * b01 - a30 * b03 - a32 * b00) * det;
  out[15] = (a20 * b03 - a21 * b01 + a22 * b00) * det;
  return out;
}

function m4rotY(a, rad) { // mat4.rotateY into a fresh matrix
  const o = new Float32Array(a), s = Math.sin(rad), c = Math.cos(rad);
  for (let k = 0; k < 4; k++) { o[k] = a[k] * c - a[8 + k] * s; o[8 + k] = a[k] * s + a[8 + k] * c; }
  return o;
}
function m4rotX(a, rad) { // mat4.rotateX into a fresh matrix
  const o = new Float32Array(a), s = Math.sin(rad), c = Math.cos(rad);
  for (let k = 0; k < 4; k++) { o[4 + k] = a[4 + k] * c + a[8 + k] * s; o[8 + k] = a[8 + k] * c - a[4 + k] * s; }
  return o;
}
function m4translate(a, v) { // mat4.translate into a fresh matrix
  const o = new Float32Array(a), x = v[0], y = v[1], z = v[2];
  for (let k = 0; k < 4; k++) o[12 + k] = a[k] * x + a[4 + k] * y + a[8 + k] * z + a[12 + k];
  return o;
}

// ---------------------------------------------------------------- camera (camera.ts)
function cameraMatrix(pitch, yaw) {
  const p = Math.min(Math.max(pitch, -Math.PI / 2), Math.PI / 2);
  const orbit = m4rotX(m4rotY(m4identity(), yaw), p);
  return m4translate(orbit, new Float32Array([0, 0, Math.abs(3)]));
}

// ---------------------------------------------------------------- scene tables
// prims: {T: Float32Array(16) world->local, r: double, c: Float32Array(3) world position,
//         lo/hi: Float32Array(3) padded AABB}
function m4rotZ(a, rad) { // mat4.rotateZ into a fresh matrix
  const o = new Float32Array(a), s = Math.sin(rad), c = Math.cos(rad);
  for (let k = 0; k < 4; k++) { o[k] = a[k] * c + a[4 + k] * s; o[4 + k] = a[4 + k] * c - a[k] * s; }
  return o;
}

// desc: [x, y, z, r] (sphere, no rotation argument) or
//       {type: 'sphere'|'box'|'torus', pos: [x,y,z], rot: Float32Array(3)|undefined, r | half | radius}
const OPS = { round: 1, twist: 1, repetition: 1, anim: 1, smoothUnion: 2, smoothSub: 2 };
function finishBounds(q, localRadius) { // boundingBox.ts:133-154
  const back = m4identity(); const ok = m4invert(back, q.T);
  const m = ok ? back : q.T;
  const sc = Math.max(Math.hypot(m[0], m[1], m[2]), Math.hypot(m[4], m[5], m[6]), Math.hypot(m[8], m[9], m[10]));
  q.localRadius = localRadius;
  q.back = back; // what mat4.invert(localToWorld, this.transform) yields inside the operators
  const pad = localRadius * sc * 1.5, c = q.c;
  q.lo = new Float32Array([c[0] - pad, c[1] - pad, c[2] - pad]);
  q.hi = new Float32Array([c[0] + pad, c[1] + pad, c[2] + pad]);
  return q;
}
function makePrim(desc) {
  if (Array.isArray(desc)) desc = { type: 'sphere', pos: [desc[0], desc[1], desc[2]], r: desc[3] };
  if (OPS[desc.type]) { // primitive_operations/*.ts
    const a = makePrim(desc.a), b = OPS[desc.type] === 2 ? makePrim(desc.b) : null;
    const q = { type: desc.type, a, b };
    if (b) { // smoothUnion.ts / smoothSubstraction.ts: identity transform
      q.T = m4identity(); q.k = desc.k;
      if (desc.type === 'smoothUnion') {
        q.c = new Float32Array([(a.c[0] + b.c[0]) / 2, (a.c[1] + b.c[1]) / 2, (a.c[2] + b.c[2]) / 2]);
        const dist = Math.hypot(b.c[0] - a.c[0], b.c[1] - a.c[1], b.c[2] - a.c[2]);
        return finishBounds(q, Math.max(a.localRadius, b.localRadius) + dist * 0.5);
      }
      q.c = a.c;
      return finishBounds(q, a.localRadius);
    }
    q.T = a.T; q.c = a.c;
    if (desc.type === 'round') { q.k = desc.radius; return finishBounds(q, a.localRadius + desc.radius); }
    if (desc.type === 'twist') { q.k = desc.amount; return finishBounds(q, a.localRadius); }
    if (desc.type === 'repetition') { q.spacing = new Float32Array(desc.spacing); return finishBounds(q, Infinity); }
    const d = new Float32Array(desc.direction); // animatedTranslate.ts:22-23 vec3.normalize
    let len = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (len > 0) len = 1 / Math.sqrt(len);
    q.dir = new Float32Array([d[0] * len, d[1] * len, d[2] * len]);
    q.amplitude = desc.amplitude; q.speed = desc.speed;
    return finishBounds(q, a.localRadius + desc.amplitude);
  }
  let model;
  if (desc.rot) { // sceneManager.ts:24-29: fromTranslation, rotateX, rotateY, rotateZ
    model = m4identity(); model[12] = desc.pos[0]; model[13] = desc.pos[1]; model[14] = desc.pos[2];
    model = m4rotZ(m4rotY(m4rotX(model, desc.rot[0]), desc.rot[1]), desc.rot[2]);
  } else model = m4fromRTS([0, 0, 0, 1], desc.pos, [1, 1, 1]);
  const T = m4identity(); m4invert(T, model);
  if (desc.type === 'mandelbulb') { // sceneManager.ts:63 mat4.scale(transform, transform, [.5,.5,.5])
    for (let k = 0; k < 12; k++) T[k] = T[k] * 0.5;
  }
  const back = m4identity(); m4invert(back, T);
  const c = new Float32Array([back[12], back[13], back[14]]);
  const q = { T, c, type: desc.type };
  let localRadius;
  if (desc.type === 'box') { q.half = new Float32Array(desc.half); localRadius = Math.hypot(q.half[0], q.half[1], q.half[2]); }
  else if (desc.type === 'torus') { q.major = desc.radius; q.minor = desc.radius / 4; localRadius = q.major + q.minor; }
  else if (desc.type === 'mandelbulb') {
    q.power = desc.power; q.iterations = desc.iterations; q.animate = !!desc.animate; q.speed = desc.speed; localRadius = 2.5;
  } else { q.r = desc.r; localRadius = q.r; }
  return finishBounds(q, localRadius);
}

function presetSpheres(index) {
  const i = Math.max(0, Math.min(index, 18));
  const out = [];
  if (i === 0) out.push([0, 0, 0, 1.5]);
  else if (i === 1) out.push([0.8, -0.3, 0.2, 0.4], [-0.5, 0.9, -0.1, 0.5], [0.2, 0.1, 0.8, 0.3], [-0.9, -0.4, -0.6, 0.6],
    [0.4, -0.8, 0.5, 0.35], [-0.2, 0.6, -0.9, 0.4], [0.7, 0.3, -0.4, 0.25]);
  else if (i === 2) { for (let y = -1; y <= 1; y++) for (let x = -1; x <= 1; x++) out.push([x, y, 0, 0.3]); }
  else if (i === 3) {
    const g = 5, sp = 0.6, off = (g - 1) * sp / 2;
    for (let x = 0; x < g; x++) for (let y = 0; y < g; y++) for (let z = 0; z < g; z++) out.push([x * sp - off, y * sp - off, z * sp - off, 0.15]);
  } else if (i === 4) out.push([0, 0, 0, 0.5], [1.2, 0, 0, 0.3], [-1.2, 0, 0, 0.3], [0, 1.2, 0, 0.3], [0, -1.2, 0, 0.3], [0, 0, 1.2, 0.3], [0, 0, -1.2, 0.3]);
  else if (i === 5) out.push({ type: 'torus', pos: [0, 0, 0], radius: 1.3, rot: new Float32Array([-Math.PI / 2, 0, 0]) });
  else if (i === 7) out.push({ type: 'box', pos: [0, 0, 0], half: [1, 1, 1] });
  else if (i === 8) out.push([-0.7, 0, 0, 0.5], { type: 'box', pos: [1, 0, 0], half: [0.5, 0.5, 0.5] });
  else if (i === 9) out.push({ type: 'box', pos: [0, 0.5, 0], half: [0.9, 0.25, 0.9] }, { type: 'box', pos: [0, 0, 0], half: [0.6, 0.25, 0.6] },
    { type: 'box', pos: [0, -0.5, 0], half: [0.3, 0.25, 0.3] });
  else {
    const PI = Math.PI, f3 = (x, y, z) => new Float32Array([x, y, z]);
    const box = (x, y, z, half, rot) => ({ type: 'box', pos: [x, y, z], half, rot });
    const sph = (x, y, z, r) => ({ type: 'sphere', pos: [x, y, z], r });
    const tor = (x, y, z, radius, rot) => ({ type: 'torus', pos: [x, y, z], radius, rot });
    const round = (a, radius) => ({ type: 'round', a, radius });
    const su = (a, b, k) => ({ type: 'smoothUnion', a, b, k });
    if (i === 6) out.push(round(box(0, 0, 0, [0.4, 0.4, 0.4]), 0.3));
    else if (i === 10) out.push(su(sph(0, 0, 0, 0.5), box(0, 0.5, 0, [1, 0.2, 1]), 0.2));
    else if (i === 11) out.push({ type: 'smoothSub', a: round(box(0, 0, 0, [1, 1, 1], f3(0, PI / 4, 0)), 0.1), b: sph(0, 0, 0, 0.9), k: 0.2 });
    else if (i === 12) out.push(su({ type: 'anim', a: sph(0, 0, 0, 1), direction: [1, 0, 0], amplitude: 3.0, speed: 0.005 }, sph(0, 0, 0, 1), 0.2));
    else if (i === 13) out.push({ type: 'mandelbulb', pos: [0, 0, 0], power: 8, iterations: 80, animate: true, speed: -0.0001 });
    else if (i === 14) out.push({ type: 'twist', a: tor(0, 0, 0, 1.3, f3(-PI / 2, 0, 0)), amount: 3 });
    else if (i === 15) out.push({ type: 'repetition', a: sph(0, 0, 0, 0.3), spacing: [1.5, 1.5, 1.5] });
    else if (i === 16) out.push(round({ type: 'twist', a: box(0, 0, 0, [0.4, 1.5, 0.4]), amount: 4.0 }, 0.1));
    else if (i === 17) {
      const b = [[0, 0, 0, 0.6, 0.6, 0.8], [0, -0.2, 0, 0.8, 0.4, 0.6], [0, -0.8, 0.8, 0.4, 0.6, 0.3], [0, -0.8, 1.2, 0.4, 0.2, 0.2],
        [0, -0.4, 1.0, 0.2, 0.2, 0.2], [0.3, 1, 0, 0.1, 0.6, 0.01], [-0.3, 1, 0, 0.1, 0.6, 0.01], [0, 1.6, 0.2, 0.6, 0.01, 0.2],
        [0.3, 1.6, 0.5, 0.1, 0.01, 0.1], [-0.3, 1.6, 0.5, 0.1, 0.01, 0.1]].map(v => box(v[0], v[1], v[2], [v[3], v[4], v[5]]));
      out.push(b.slice(1).reduce((acc, x) => su(acc, x, 0.0001), b[0]));
    } else {
      out.push(su(round(box(-1.25, -0.8, 0, [0.05, 0.7, 0.05], f3(0, 0, PI / 5)), 0.20), round(tor(-1.25, 0.5, 0, 0.8, f3(-PI / 2, 0, 0)), 0.05), 0.0001));
      out.push(su(round(box(1.35, 0, 0, [0.05, 1.5, 0.05], f3(0, 0, PI / 7)), 0.20), round(box(1.25, -1.4, 0, [0.05, 0.8, 0.05], f3(0, 0, PI / 2)), 0.20), 0.0001));
    }
  }
  return out;
}

function unionBox(prims, ids) { // computeBounds: fold of merge, f32 at every step
  if (ids.length === 0) return { lo: new Float32Array(3), hi: new Float32Array(3) };
  const lo = new Float32Array(prims[ids[0]].lo), hi = new Float32Array(prims[ids[0]].hi);
  for (let k = 1; k < ids.length; k++) for (let a = 0; a < 3; a++) {
    lo[a] = Math.min(lo[a], prims[ids[k]].lo[a]); hi[a] = Math.max(hi[a], prims[ids[k]].hi[a]);
  }
  return { lo, hi };
}

// BVH as parallel arrays; node = index.  bvh.ts:29-92
function buildBVH(prims) {
  const B = { lo: [], hi: [], L: [], R: [], P: [], leaves: 0, depth: 0 };
  function rec(ids, box, depth) {
    const me = B.lo.length;
    B.lo.push(box.lo); B.hi.push(box.hi); B.L.push(-1); B.R.push(-1); B.P.push(null);
    if (depth > B.depth) B.depth = depth;
    if (depth >= 20 || ids.length <= 2) { B.P[me] = ids; B.leaves++; return me; }
    const ext = new Float32Array(3);
    for (let a = 0; a < 3; a++) ext[a] = box.hi[a] - box.lo[a];
    let axis = 0; if (ext[1] > ext[0]) axis = 1; if (ext[2] > ext[axis]) axis = 2;
    const order = ids.slice().sort((p, q) => prims[p].c[axis] - prims[q].c[axis]);
    const mid = Math.floor(order.length / 2);
    const lh = order.slice(0, mid), rh = order.slice(mid);
    if (lh.length === 0 || rh.length === 0) { B.P[me] = ids; B.leaves++; return me; }
    B.L[me] = rec(lh, unionBox(prims, lh), depth + 1);
    B.R[me] = rec(rh, unionBox(prims, rh), depth + 1);
    return me;
  }
  const all = prims.map((_, i) => i);
  rec(all, unionBox(prims, all), 0);
  return B;
}

function inBox(lo, hi, p) {
  return p[0] >= lo[0] && p[0] <= hi[0] && p[1] >= lo[1] && p[1] <= hi[1] && p[2] >= lo[2] && p[2] <= hi[2];
}

function slab(lo, hi, o, d) { // boundingBox.ts:69-105
  let tn = -Infinity, tf = Infinity;
  for (let a = 0; a < 3; a++) {
    if (Math.abs(d[a]) < 1e-10) { if (o[a] < lo[a] || o[a] > hi[a]) return null; }
    else {
      const inv = 1.0 / d[a];
      let t0 = (lo[a] - o[a]) * inv, t1 = (hi[a] - o[a]) * inv;
      if (t0 > t1) { const t = t0; t0 = t1; t1 = t; }
      tn = Math.max(tn, t0); tf = Math.min(tf, t1);
      if (tn > tf) return null;
    }
  }
  return [tn, tf];
}

function bvhIntervals(B, o, d, tLo, tHi) { // bvh.ts:126-178
  const found = [], todo = [0];
  while (todo.length) {
    const n = todo.pop();
    const h = slab(B.lo[n], B.hi[n], o, d);
    if (!h) continue;
    if (h[1] < tLo || h[0] > tHi) continue;
    if (B.L[n] >= 0 || B.R[n] >= 0) { if (B.L[n] >= 0) todo.push(B.L[n]); if (B.R[n] >= 0) todo.push(B.R[n]); }
    else if (B.P[n] && B.P[n].length > 0) found.push({ a: Math.max(h[0], tLo), b: Math.min(h[1], tHi) });
  }
  found.sort((u, v) => u.a - v.a);
  return found;
}

function bvhPointPrims(B, p) { // bvh.ts:95-121 (left before right, Set semantics)
  const seen = new Set(), todo = [0];
  while (todo.length) {
    const n = todo.pop();
    if (!inBox(B.lo[n], B.hi[n], p)) continue;
    if (B.L[n] < 0 && B.R[n] < 0) { for (const id of B.P[n]) seen.add(id); continue; }
    if (B.R[n] >= 0) todo.push(B.R[n]);
    if (B.L[n] >= 0) todo.push(B.L[n]);
  }
  return Array.from(seen);
}

// Octree as parallel arrays.  octree.ts:36-191
function boxGap(alo, ahi, blo, bhi) { // boundingBox.ts:33-47
  const g = [0, 0, 0];
  for (let a = 0; a < 3; a++) {
    if (ahi[a] < blo[a]) g[a] = blo[a] - ahi[a]; else if (bhi[a] < alo[a]) g[a] = alo[a] - bhi[a];
  }
  return Math.hypot(g[0], g[1], g[2]);
}
function buildOctree(prims) {
  const O = { lo: [], hi: [], kids: [], P: [], level: [], empty: [], minD: [] };
  function add(lo, hi, level) {
    O.lo.push(lo); O.hi.push(hi); O.kids.push(null); O.P.push([]); O.level.push(level); O.empty.push(true); O.minD.push(0);
    return O.lo.length - 1;
  }
  function rec(ids, lo, hi, depth) {
    const me = add(lo, hi, depth);
    if (depth >= 6 || ids.length <= 4) { O.P[me] = ids; return me; }
    const c = new Float32Array([(lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, (lo[2] + hi[2]) / 2]);
    const cl = [], ch = [];
    for (let k = 0; k < 8; k++) {
      const bx = k & 1, by = (k >> 1) & 1, bz = (k >> 2) & 1;
      cl.push(new Float32Array([bx ? c[0] : lo[0], by ? c[1] : lo[1], bz ? c[2] : lo[2]]));
      ch.push(new Float32Array([bx ? hi[0] : c[0], by ? hi[1] : c[1], bz ? hi[2] : c[2]]));
    }
    const share = [[], [], [], [], [], [], [], []];
    for (const id of ids) for (let k = 0; k < 8; k++) {
      const q = prims[id];
      if (cl[k][0] <= q.hi[0] && ch[k][0] >= q.lo[0] && cl[k][1] <= q.hi[1] && ch[k][1] >= q.lo[1] && cl[k][2] <= q.hi[2] && ch[k][2] >= q.lo[2]) share[k].push(id);
    }
    const kids = [];
    for (let k = 0; k < 8; k++) kids.push(share[k].length > 0 ? rec(share[k], cl[k], ch[k], depth + 1) : add(cl[k], ch[k], depth + 1));
    O.kids[me] = kids;
    return me;
  }
  rec(prims.map((_, i) => i), new Float32Array([-10, -10, -10]), new Float32Array([10, 10, 10]), 0);
  function nearest(n) {
    let best = Infinity;
    for (const q of prims) { const g = boxGap(O.lo[n], O.hi[n], q.lo, q.hi); if (g < best) best = g; }
    return best !== Infinity ? Math.max(0, best) : 0;
  }
  function mark(n) {
    if (!O.kids[n]) { const has = O.P[n].length > 0; O.empty[n] = !has; O.minD[n] = has ? 0 : nearest(n); return has; }
    let any = false;
    for (const k of O.kids[n]) if (mark(k)) any = true;
    O.empty[n] = !any; O.minD[n] = any ? 0 : nearest(n);
    return any;
  }
  mark(0);
  return O;
}
function octFind(O, p) { // octree.ts:223-248
  function rec(n) {
    if (!inBox(O.lo[n], O.hi[n], p)) return -1;
    if (!O.kids[n] || O.level[n] === 6) return n;
    for (const k of O.kids[n]) { const f = rec(k); if (f >= 0) return f; }
    return n;
  }
  return rec(0);
}
const _tA = new Float32Array(3), _tB = new Float32Array(3);
function octSkip(O, o, d, t) { // octree.ts:252-278 + 195-220
  const p = new Float32Array([o[0] + d[0] * t, o[1] + d[1] * t, o[2] + d[2] * t]);
  const n = octFind(O, p);
  if (n < 0) return 0;
  if (O.empty[n]) {
    for (let a = 0; a < 3; a++) {
      const inv = 1.0 / d[a];
      let t0 = (O.lo[n][a] - o[a]) * inv, t1 = (O.hi[n][a] - o[a]) * inv;
      if (inv < 0.0) { const s = t0; t0 = t1; t1 = s; }
      _tA[a] = t0; _tB[a] = t1;
    }
    const tIn = Math.max(_tA[0], _tA[1], _tA[2]), tOut = Math.min(_tB[0], _tB[1], _tB[2]);
    if (!(tIn > tOut || tOut < 0)) {
      const toExit = Math.max(0, tOut - t);
      const step = Math.max(0, Math.min(toExit, O.minD[n] * 0.99));
      return step > 0 ? step + 0.001 : 0;
    }
  }
  return 0;
}

// ---------------------------------------------------------------- scene distance
// Math.pow is the one transcendental whose node-12 (V8 7.8) value is not the fdlibm one that
// current V8 computes (ieee754::legacy::pow); e_pow.c is therefore restated here as well.
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
const _f64 = new Float64Array(1), _u32 = new Uint32Array(_f64.buffer);
function hiWord(x) { _f64[0] = x; return _u32[1] | 0; }
function loWord(x) { _f64[0] = x; return _u32[0] >>> 0; }
function withLo(x, lo) { _f64[0] = x; _u32[0] = lo; return _f64[0]; }
function withHi(x, hi) { _f64[0] = x; _u32[1] = hi; return _f64[0]; }
function fromWords(hi, lo) { _u32[1] = hi; _u32[0] = lo; return _f64[0]; }
function fdlibmPow(x, y) {
  const bp = [1.0, 1.5], dp_h = [0.0, 5.84962487220764160156e-01], dp_l = [0.0, 1.35003920212974897128e-08];
  const two53 = 9007199254740992.0, huge = 1.0e300, tiny = 1.0e-300,
    L1 = 5.99999999999994648725e-01, L2 = 4.28571428578550184252e-01, L3 = 3.33333329818377432918e-01,
    L4 = 2.72728123808534006489e-01, L5 = 2.30660745775561754067e-01, L6 = 2.06975017800338417784e-01,
    P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
    P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08,
    lg2 = 6.93147180559945286227e-01, lg2_h = 6.93147182464599609375e-01, lg2_l = -1.90465429995776804525e-09,
    ovt = 8.0085662595372944372e-0017, cp = 9.61796693925975554329e-01, cp_h = 9.61796700954437255859e-01,
    cp_l = -7.02846165095275826516e-09, ivln2 = 1.44269504088896338700e+00, ivln2_h = 1.44269502162933349609e+00,
    ivln2_l = 1.92596299112661746887e-08;
  const hx = hiWord(x), lx = loWord(x), hy = hiWord(y), ly = loWord(y);
  let ix = hx & 0x7fffffff; const iy = hy & 0x7fffffff;
  if ((iy | ly) === 0) return 1.0;
  if (ix > 0x7ff00000 || (ix === 0x7ff00000 && lx !== 0) || iy > 0x7ff00000 || (iy === 0x7ff00000 && ly !== 0)) return x + y;
  let yisint = 0, k, j, n, i;
  if (hx < 0) {
    if (iy >= 0x43400000) yisint = 2;
    else if (iy >= 0x3ff00000) {
      k = (iy >> 20) - 0x3ff;
      if (k > 20) { j = ly >>> (52 - k); if (((j << (52 - k)) >>> 0) === ly) yisint = 2 - (j & 1); }
      else if (ly === 0) { j = iy >> (20 - k); if ((j << (20 - k)) === iy) yisint = 2 - (j & 1); }
    }
  }
  if (ly === 0) {
    if (iy === 0x7ff00000) {
      if (((ix - 0x3ff00000) | lx) === 0) return y - y;
      else if (ix >= 0x3ff00000) return (hy >= 0) ? y : 0.0;
      else return (hy < 0) ? -y : 0.0;
    }
    if (iy === 0x3ff00000) return (hy < 0) ? 1.0 / x : x;
    if (hy === 0x40000000) return x * x;
    if (hy === 0x3fe00000) { if (hx >= 0) return Math.sqrt(x); }
  }
  let ax = Math.abs(x), z;
  if (lx === 0) {
    if (ix === 0x7ff00000 || ix === 0 || ix === 0x3ff00000) {
      z = ax;
      if (hy < 0) z = 1.0 / z;
      if (hx < 0) {
        if (((ix - 0x3ff00000) | yisint) === 0) z = (z - z) / (z - z);
        else if (yisint === 1) z = -z;
      }
      return z;
    }
  }
  n = (hx >> 31) + 1;
  if ((n | yisint) === 0) return (x - x) / (x - x);
  let s = 1.0;
  if ((n | (yisint - 1)) === 0) s = -1.0;
  let t, u, v, w, t1, t2, r, p_h, p_l, z_h, z_l;
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
    t1 = withLo(u + v, 0);
    t2 = v - (t1 - u);
  } else {
    n = 0;
    if (ix < 0x00100000) { ax *= two53; n -= 53; ix = hiWord(ax); }
    n += (ix >> 20) - 0x3ff;
    j = ix & 0x000fffff;
    ix = j | 0x3ff00000;
    if (j <= 0x3988E) k = 0;
    else if (j < 0xBB67A) k = 1;
    else { k = 0; n += 1; ix -= 0x00100000; }
    ax = withHi(ax, ix);
    u = ax - bp[k];
    v = 1.0 / (ax + bp[k]);
    const ss = u * v;
    const s_h = withLo(ss, 0);
    let t_h = fromWords(((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18), 0);
    let t_l = ax - (t_h - bp[k]);
    const s_l = v * ((u - s_h * t_h) - s_h * t_l);
    let s2 = ss * ss;
    r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
    r += s_l * (s_h + ss);
    s2 = s_h * s_h;
    t_h = withLo(3.0 + s2 + r, 0);
    t_l = r - ((t_h - 3.0) - s2);
    u = s_h * t_h;
    v = s_l * t_h + t_l * ss;
    p_h = withLo(u + v, 0);
    p_l = v - (p_h - u);
    z_h = cp_h * p_h;
    z_l = cp_l * p_h + p_l * cp + dp_l[k];
    t = n;
    t1 = withLo(((z_h + z_l) + dp_h[k]) + t, 0);
    t2 = z_l - (((t1 - t) - dp_h[k]) - z_h);
  }
  const y1 = withLo(y, 0);
  p_l = (y - y1) * t1 + y * t2;
  p_h = y1 * t1;
  z = p_l + p_h;
  j = hiWord(z); i = loWord(z) | 0;
  if (j >= 0x40900000) {
    if (((j - 0x40900000) | i) !== 0) return s * huge * huge;
    if (p_l + ovt > z - p_h) return s * huge * huge;
  } else if ((j & 0x7fffffff) >= 0x4090cc00) {
    if (((j - (0xc090cc00 | 0)) | i) !== 0) return s * tiny * tiny;
    if (p_l <= z - p_h) return s * tiny * tiny;
  }
  i = j & 0x7fffffff;
  k = (i >> 20) - 0x3ff;
  n = 0;
  if (i > 0x3fe00000) {
    n = (j + (0x00100000 >> (k + 1))) | 0;
    k = ((n & 0x7fffffff) >> 20) - 0x3ff;
    t = fromWords(n & ~(0x000fffff >> k), 0);
    n = ((n & 0x000fffff) | 0x00100000) >> (20 - k);
    if (j < 0) n = -n;
    p_h -= t;
  }
  t = withLo(p_l + p_h, 0);
  u = t * lg2_h;
  v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
  z = u + v;
  w = v - (z - u);
  t = z * z;
  t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  r = (z * t1) / (t1 - 2.0) - (w + z * w);
  z = 1.0 - (r - z);
  j = hiWord(z);
  j = (j + (n << 20)) | 0;
  if ((j >> 20) <= 0) { // scalbn(z, n), n < 0: subnormal result; one rounding, at the last multiplication
    let e = n;
    if (e < -1022) { z *= fromWords((0x3ff - 969) << 20, 0); e += 969; if (e < -1022) { z *= fromWords((0x3ff - 969) << 20, 0); e += 969; if (e < -1022) e = -1022; } }
    z = z * fromWords((0x3ff + e) << 20, 0);
  }
  else z = withHi(z, j);
  return s * z;
}

let gTime = 0; // Scene.updateTime (scene.ts:135-140)
function xform(m, x, y, z, out) { // vec3.transformMat4
  let w = m[3] * x + m[7] * y + m[11] * z + m[15]; w = w || 1.0;
  out[0] = (m[0] * x + m[4] * y + m[8] * z + m[12]) / w;
  out[1] = (m[1] * x + m[5] * y + m[9] * z + m[13]) / w;
  out[2] = (m[2] * x + m[6] * y + m[10] * z + m[14]) / w;
  return out;
}
function mandelbulbSdf(q, lx, ly, lz, useSqrt) { // mandelbulb.ts:37-78
  const p = new Float32Array([lx, lz, ly]), z = new Float32Array(p);
  let dr = 1.0, r = 0.0;
  for (let i = 0; i < q.iterations; i++) {
    r = useSqrt ? Math.sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]) : Math.hypot(z[0], z[1], z[2]);
    if (r > 2.0) break;
    let theta = Math.atan2(z[1], z[0]);
    let phi = Math.asin(z[2] / r);
    if (q.animate) phi += gTime * q.speed;
    dr = fdlibmPow(r, q.power - 1.0) * dr * q.power + 1.0;
    r = fdlibmPow(r, q.power);
    theta = theta * q.power;
    phi = phi * q.power;
    z[0] = r * Math.cos(theta) * Math.cos(phi) + p[0];
    z[1] = r * Math.sin(theta) * Math.cos(phi) + p[1];
    z[2] = r * Math.sin(phi) + p[2];
  }
  return 0.5 * Math.log(r) * r / dr;
}
function opSdf(q, lx, ly, lz, useSqrt) { // primitive_operations/*.ts localSdf
  if (q.type === 'anim') { // animatedTranslate.ts:34-49
    const offset = Math.sin(gTime * q.speed) * q.amplitude;
    const off = new Float32Array([q.dir[0] * offset, q.dir[1] * offset, q.dir[2] * offset]);
    const adj = new Float32Array([lx - off[0], ly - off[1], lz - off[2]]);
    return sphereSdf(q.a, adj, useSqrt);
  }
  const w = xform(q.back, lx, ly, lz, new Float32Array(3));
  if (q.type === 'round') return sphereSdf(q.a, w, useSqrt) - q.k;
  if (q.type === 'twist') {
    const c = Math.cos(q.k * w[1]), s = Math.sin(q.k * w[1]);
    return sphereSdf(q.a, new Float32Array([c * w[0] - s * w[2], w[1], s * w[0] + c * w[2]]), useSqrt);
  }
  if (q.type === 'repetition') {
    const sp = q.spacing;
    const r = new Float32Array([w[0] - sp[0] * Math.round(w[0] / sp[0]), w[1] - sp[1] * Math.round(w[1] / sp[1]),
      w[2] - sp[2] * Math.round(w[2] / sp[2])]);
    return sphereSdf(q.a, r, useSqrt);
  }
  const d1 = sphereSdf(q.a, w, useSqrt), d2 = sphereSdf(q.b, w, useSqrt), k = q.k * 4.0;
  if (q.type === 'smoothUnion') { const h = Math.max(k - Math.abs(d1 - d2), 0.0); return Math.min(d1, d2) - h * h * 0.25 / k; }
  const h = Math.max(k - Math.abs(d1 + d2), 0.0);
  return Math.max(d1, -d2) + h * h * 0.25 / k;
}

function sphereSdf(q, p, useSqrt) { // primitive.ts:33-39 + every class's localSdf
  const m = q.T, x = p[0], y = p[1], z = p[2];
  let w = m[3] * x + m[7] * y + m[11] * z + m[15]; w = w || 1.0;
  const lx = fr((m[0] * x + m[4] * y + m[8] * z + m[12]) / w);
  const ly = fr((m[1] * x + m[5] * y + m[9] * z + m[13]) / w);
  const lz = fr((m[2] * x + m[6] * y + m[10] * z + m[14]) / w);
  if (OPS[q.type]) return opSdf(q, lx, ly, lz, useSqrt);
  if (q.type === 'mandelbulb') return mandelbulbSdf(q, lx, ly, lz, useSqrt);
  if (q.type === 'box') { // box.ts:13-30
    const e = new Float32Array([Math.abs(lx) - q.half[0], Math.abs(ly) - q.half[1], Math.abs(lz) - q.half[2]]);
    const out = new Float32Array([Math.max(e[0], 0), Math.max(e[1], 0), Math.max(e[2], 0)]);
    const od = useSqrt ? Math.sqrt(out[0] * out[0] + out[1] * out[1] + out[2] * out[2]) : Math.hypot(out[0], out[1], out[2]);
    return od + Math.min(Math.max(e[0], Math.max(e[1], e[2])), 0);
  }
  if (q.type === 'torus') { // torus.ts:14-25
    const qx = Math.sqrt(lx * lx + lz * lz) - q.major;
    return Math.sqrt(qx * qx + ly * ly) - q.minor;
  }
  const len = useSqrt ? Math.sqrt(lx * lx + ly * ly + lz * lz) : Math.hypot(lx, ly, lz);
  return len - q.r;
}

function makeScene(spheres, accel, useSqrt) {
  const prims = spheres.map(makePrim);
  const S = { prims, accel, bvh: null, oct: null, useSqrt: !!useSqrt };
  if (accel === 'BVH') S.bvh = buildBVH(prims);
  else if (accel === 'Octree') S.oct = buildOctree(prims);
  else S.accel = 'None';
  return S;
}

function sceneDistance(S, p, tally) { // scene.ts:144-190; tally[0] += evaluations
  let best = 10;
  const prims = S.prims;
  if (S.accel === 'Octree') {
    const n = octFind(S.oct, p);
    if (n >= 0) {
      const ids = S.oct.P[n];
      if (ids.length > 0) { for (const id of ids) { tally[0]++; best = Math.min(sphereSdf(prims[id], p, S.useSqrt), best); } }
      else if (S.oct.empty[n]) best = Math.min(best, S.oct.minD[n] * 0.99);
      return best;
    }
  } else if (S.accel === 'BVH') {
    let ids = bvhPointPrims(S.bvh, p);
    if (ids.length === 0) ids = prims.map((_, i) => i);
    for (const id of ids) { tally[0]++; best = Math.min(sphereSdf(prims[id], p, S.useSqrt), best); }
    return best;
  }
  for (let id = 0; id < prims.length; id++) { tally[0]++; best = Math.min(sphereSdf(prims[id], p, S.useSqrt), best); }
  return best;
}

// ---------------------------------------------------------------- tile render
function renderTile(S, cam, W, H, y0, y1, algorithm, overshootFactor, stepSize) {
  const rows = Math.max(0, y1 - y0);
  const depth = new Uint8ClampedArray(W * rows), normal = new Uint8ClampedArray(W * rows * 3);
  const sdf = new Uint16Array(W * rows), iters = new Uint16Array(W * rows);
  const R = [cam[0], cam[1], cam[2], cam[4], cam[5], cam[6], cam[8], cam[9], cam[10]]; // mat3.fromMat4
  const o = new Float32Array([cam[12], cam[13], cam[14]]);
  const d = new Float32Array(3), p = new Float32Array(3), q = new Float32Array(3), nrm = new Float32Array(3);
  const tally = [0];
  function dist(at, px) { tally[0] = 0; const v = sceneDistance(S, at, tally); sdf[px] += tally[0]; return v; }

  // accel prologue / step callback shared by the five marchers
  let list = null, cur = 0;
  function accelStart() {
    list = null; cur = 0;
    if (S.accel === 'BVH') { list = bvhIntervals(S.bvh, o, d, 0, 10); if (list.length === 0) return false; }
    return true;
  }
  function accelStep(t) { // 0: march, > 0: skip, -1: terminate
    if (S.accel === 'None') return 0;
    if (S.accel === 'BVH') {
      if (cur >= list.length) return -1;
      if (t < list[cur].a) return list[cur].a - t;
      if (t > list[cur].b) {
        cur++;
        if (cur < list.length) { if (list[cur].a > t) return list[cur].a - t; }
        else return -1;
      }
      return 0;
    }
    return octSkip(S.oct, o, d, t);
  }
  function at(t) { p[0] = o[0] + d[0] * t; p[1] = o[1] + d[1] * t; p[2] = o[2] + d[2] * t; return p; }

  function marchSphere(px) { // sphereTracer.ts:15-83
    let t = 0;
    if (!accelStart()) return 10;
    for (let i = 0; i < 100; i++) {
      at(t);
      const skip = accelStep(t);
      if (skip === -1) return 10;
      if (skip > 0) { t += skip; if (t > 10) break; continue; }
      const dd = dist(p, px);
      t += dd;
      iters[px] += 1;
      if (dd < 0.001) break;
      if (t > 10) break;
    }
    return t;
  }
  function marchFixed(px, stepSize) { // fixedStep.ts:21-94
    let t = 0, hit = false;
    if (!accelStart()) return 10;
    for (let i = 0; i < 200; i++) {
      at(t);
      const skip = accelStep(t);
      if (skip === -1) return 10;
      if (skip > 0) { t += skip; if (t > 10) break; continue; }
      const dd = dist(p, px);
      iters[px] += 1;
      if (dd < 0.001) { hit = true; break; }
      t += stepSize;
      if (t > 10) break;
    }
    return hit ? t : 10;
  }
  function marchAdaptive(px) { // adaptiveStep.ts:22-105
    const FIXED = 0.1, MINS = FIXED * 0.25, MAXS = FIXED * 5.0;
    let t = 0, hit = false;
    if (!accelStart()) return 10;
    for (let i = 0; i < 200; i++) {
      at(t);
      const skip = accelStep(t);
      if (skip === -1) return 10;
      if (skip > 0) { t += skip; if (t > 10) break; continue; }
      const dd = dist(p, px);
      iters[px] += 1;
      if (dd < 0.001) { hit = true; break; }
      let step;
      if (dd < 0.1) step = 0.01;
      else { step = 0.8 * dd; if (step < MINS) step = MINS; if (step > MAXS) step = MAXS; }
      t += step;
      if (t > 10) break;
    }
    return hit ? t : 10;
  }
  function marchV23(px, k, v3) { // adaptiveStepV2.ts:22-124, adaptiveStepV3.ts:22-137
    let t = 0, prevSDF = 0, prevStep = 0;
    if (!accelStart()) return 10;
    for (let i = 0; i < 100; i++) {
      at(t);
      const skip = accelStep(t);
      if (skip === -1) return 10;
      if (skip > 0) { t += skip; if (t > 10) break; prevSDF = 0; prevStep = 0; continue; }
      const nw = dist(p, px);
      iters[px] += 1;
      if (nw < 0.001) break;
      if (t > 10) break;
      if (i === 0 || prevSDF === 0) { t += nw; prevSDF = nw; prevStep = nw; continue; }
      if (prevStep <= (prevSDF + nw)) { const st = nw * k; t += st; prevSDF = nw; prevStep = st; continue; }
      if (!v3) { t -= prevStep; t += prevSDF; prevStep = prevSDF; continue; }
      const orig = t - prevStep;
      t = orig + prevSDF;
      at(t);
      const d3 = dist(p, px);
      iters[px] += 1;
      if (prevSDF + nw + d3 >= prevStep) { t = orig + prevStep + nw; prevSDF = nw; prevStep = nw; continue; }
      prevSDF = d3; prevStep = d3; t += d3;
    }
    return t;
  }
  function march(px) { // raymarchWorker.ts:49-68
    switch (algorithm) {
      case 'fixed-step': return marchFixed(px, stepSize === undefined ? 0.1 : stepSize);
      case 'adaptive-step': return marchAdaptive(px);
      case 'adaptive-step-v2': return marchV23(px, overshootFactor === undefined ? 1.2 : overshootFactor, false);
      case 'adaptive-step-v3': return marchV23(px, overshootFactor === undefined ? 1.2 : overshootFactor, true);
      default: return marchSphere(px);
    }
  }

  for (let y = y0; y < y1; y++) {
    const v = (y / H - 0.5) * 2.0;
    for (let x = 0; x < W; x++) {
      const px = (y - y0) * W + x;
      sdf[px] = 0; iters[px] = 0;
      const u = (x / W - 0.5) * 2.0;
      d[0] = u; d[1] = v; d[2] = -1;
      const ax = d[0], ay = d[1], az = d[2];
      d[0] = ax * R[0] + ay * R[3] + az * R[6];
      d[1] = ax * R[1] + ay * R[4] + az * R[7];
      d[2] = ax * R[2] + ay * R[5] + az * R[8];
      let len = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      if (len > 0) len = 1 / Math.sqrt(len);
      d[0] = d[0] * len; d[1] = d[1] * len; d[2] = d[2] * len;

      const t = march(px);
      q[0] = o[0] + d[0] * t; q[1] = o[1] + d[1] * t; q[2] = o[2] + d[2] * t;
      nrm[0] = 0; nrm[1] = 0; nrm[2] = 0;
      if (!(t >= 10)) { // raymarcher.ts:123-135
        const base = dist(q, px);
        const s = new Float32Array(3);
        s[0] = q[0] - 0.01; s[1] = q[1]; s[2] = q[2]; nrm[0] = base - dist(s, px);
        s[0] = q[0]; s[1] = q[1] - 0.01; s[2] = q[2]; nrm[1] = base - dist(s, px);
        s[0] = q[0]; s[1] = q[1]; s[2] = q[2] - 0.01; nrm[2] = base - dist(s, px);
        let l2 = nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2];
        if (l2 > 0) l2 = 1 / Math.sqrt(l2);
        nrm[0] = nrm[0] * l2; nrm[1] = nrm[1] * l2; nrm[2] = nrm[2] * l2;
      }
      normal[3 * px] = (nrm[0] + 1) * 0.5 * 255;
      normal[3 * px + 1] = (nrm[1] + 1) * 0.5 * 255;
      normal[3 * px + 2] = (nrm[2] + 1) * 0.5 * 255;
      depth[px] = t;
    }
  }
  return { depth, normal, sdf, iters };
}

// ---------------------------------------------------------------- shading
function shade(model, depth, normal, sdf, iters, W, H) {
  const out = new Uint8ClampedArray(W * H * 4), n = W * H;
  if (model === 'sdf-heatmap' || model === 'iteration-heatmap') {
    const src = model === 'sdf-heatmap' ? sdf : iters;
    for (let i = 0; i < n; i++) {
      const k = src[i] * 5 % 256;
      out[4 * i] = Math.min(2 * k, 255); out[4 * i + 1] = Math.min(-2 * k + 512, 255); out[4 * i + 2] = 0; out[4 * i + 3] = 255;
    }
  } else if (model === 'phong') {
    const L = new Float32Array([1, -1, 1.5]);
    let ll = L[0] * L[0] + L[1] * L[1] + L[2] * L[2]; ll = 1 / Math.sqrt(ll);
    L[0] = L[0] * ll; L[1] = L[1] * ll; L[2] = L[2] * ll;
    const N = new Float32Array(3), Rf = new Float32Array(3);
    for (let i = 0; i < n; i++) {
      const dz = depth[i];
      if (dz >= 255) { out[4 * i] = 10; out[4 * i + 1] = 10; out[4 * i + 2] = 20; out[4 * i + 3] = 255; continue; }
      N[0] = normal[3 * i] / 127.5 - 1.0; N[1] = normal[3 * i + 1] / 127.5 - 1.0; N[2] = normal[3 * i + 2] / 127.5 - 1.0;
      let l2 = N[0] * N[0] + N[1] * N[1] + N[2] * N[2]; if (l2 > 0) l2 = 1 / Math.sqrt(l2);
      N[0] = N[0] * l2; N[1] = N[1] * l2; N[2] = N[2] * l2;
      const ndl = N[0] * L[0] + N[1] * L[1] + N[2] * L[2];
      const diffuse = Math.max(ndl, 0);
      const k2 = 2 * ndl;
      Rf[0] = N[0] * k2; Rf[1] = N[1] * k2; Rf[2] = N[2] * k2;
      Rf[0] = Rf[0] - L[0]; Rf[1] = Rf[1] - L[1]; Rf[2] = Rf[2] - L[2];
      let r2 = Rf[0] * Rf[0] + Rf[1] * Rf[1] + Rf[2] * Rf[2]; if (r2 > 0) r2 = 1 / Math.sqrt(r2);
      Rf[0] = Rf[0] * r2; Rf[1] = Rf[1] * r2; Rf[2] = Rf[2] * r2;
      const vdr = 0 * Rf[0] + 0 * Rf[1] + 1 * Rf[2];
      const spec = 0.5 * Math.pow(Math.max(vdr, 0), 32);
      const inten = Math.min(0.1 + diffuse + spec, 1);
      const col = 255 * inten * (1 - dz / 255);
      out[4 * i] = col; out[4 * i + 1] = col; out[4 * i + 2] = col; out[4 * i + 3] = 255;
    }
  } else {
    for (let i = 0; i < n; i++) { out[4 * i] = normal[3 * i]; out[4 * i + 1] = normal[3 * i + 1]; out[4 * i + 2] = normal[3 * i + 2]; out[4 * i + 3] = 255; }
  }
  return out;
}

// ---------------------------------------------------------------- driver
function sha(buf) { return crypto.createHash('sha256').update(Buffer.from(buf.buffer, buf.byteOffset, buf.byteLength)).digest('hex'); }

function cmdRender(cfgPath, outDir) {
  const cfg = JSON.parse(fs.readFileSync(cfgPath, 'utf8'));
  let spheres;
  if (cfg.spheres_file) {
    const raw = fs.readFileSync(cfg.spheres_file);
    const f = new Float64Array(raw.buffer, raw.byteOffset, raw.byteLength / 8);
    spheres = [];
    for (let i = 0; i + 3 < f.length; i += 4) spheres.push([f[i], f[i + 1], f[i + 2], f[i + 3]]);
  } else if (cfg.prims) {
    const conv = d => {
      const o = Object.assign({}, d);
      if (d.rot) o.rot = new Float32Array(d.rot); else delete o.rot;
      if (d.a) o.a = conv(d.a);
      if (d.b) o.b = conv(d.b);
      return o;
    };
    spheres = cfg.prims.map(conv);
  } else spheres = presetSpheres(cfg.preset);
  const S = makeScene(spheres, cfg.accel, cfg.length_sqrt);
  const cam = cameraMatrix(cfg.pitch || 0, cfg.yaw || 0);
  const W = cfg.width, H = cfg.height;
  const y0 = cfg.yStart === undefined ? 0 : cfg.yStart, y1 = cfg.yEnd === undefined ? H : cfg.yEnd;
  const t0 = process.hrtime.bigint();
  gTime = cfg.time || 0; // raymarcher.ts:58-59 scene.updateTime(time)
  const r = renderTile(S, cam, W, H, y0, y1, cfg.algorithm, cfg.overshootFactor, cfg.stepSize);
  const t1 = process.hrtime.bigint();
  const rgba = shade(cfg.shader || 'normal', r.depth, r.normal, r.sdf, r.iters, W, y1 - y0);
  let sumS = 0, sumI = 0, mx = 0, mn = Number.MAX_SAFE_INTEGER;
  for (let i = 0; i < r.sdf.length; i++) { const c = r.sdf[i]; sumS += c; sumI += r.iters[i]; if (c > mx) mx = c; if (c < mn) mn = c; }
  if (outDir) {
    fs.mkdirSync(outDir, { recursive: true });
    const w = (name, a) => fs.writeFileSync(path.join(outDir, name), Buffer.from(a.buffer, a.byteOffset, a.byteLength));
    w('depth.bin', r.depth); w('normal.bin', r.normal); w('sdf.bin', r.sdf); w('iters.bin', r.iters); w('rgba.bin', rgba);
  }
  const stats = {
    width: W, height: H, yStart: y0, yEnd: y1, n_prims: spheres.length,
    sum_sdf: sumS, max_sdf: mx, min_sdf: mn, sum_iters: sumI, render_ms: Number(t1 - t0) / 1e6,
    sha256: { depth: sha(r.depth), normal: sha(r.normal), sdf: sha(r.sdf), iters: sha(r.iters), rgba: sha(rgba) },
    bvh: S.bvh ? { nodes: S.bvh.lo.length, leaves: S.bvh.leaves, depth: S.bvh.depth } : null,
    octree: S.oct ? { nodes: S.oct.lo.length } : null,
    engine: 'node ' + process.version + ' v8 ' + process.versions.v8,
  };
  process.stdout.write(JSON.stringify(stats) + '\n');
}

function cmdHypot(inPath, outPath) {
  const raw = fs.readFileSync(inPath);
  const f = new Float64Array(raw.buffer, raw.byteOffset, raw.byteLength / 8);
  const out = new Float64Array(f.length / 3);
  for (let i = 0; i < out.length; i++) out[i] = Math.hypot(f[3 * i], f[3 * i + 1], f[3 * i + 2]);
  fs.writeFileSync(outPath, Buffer.from(out.buffer));
}

function cmdCamera(inPath, outPath) {
  const raw = fs.readFileSync(inPath);
  const f = new Float64Array(raw.buffer, raw.byteOffset, raw.byteLength / 8);
  const out = new Float32Array(f.length / 2 * 12);
  for (let i = 0; i < f.length / 2; i++) {
    const c = cameraMatrix(f[2 * i], f[2 * i + 1]);
    const v = [c[0], c[1], c[2], c[4], c[5], c[6], c[8], c[9], c[10], c[12], c[13], c[14]];
    for (let k = 0; k < 12; k++) out[12 * i + k] = v[k];
  }
  fs.writeFileSync(outPath, Buffer.from(out.buffer));
}

function cmdJsMath(fn, aPath, bPath, outPath) {
  const rd = f => { const raw = fs.readFileSync(f); return new Float64Array(raw.buffer, raw.byteOffset, raw.byteLength / 8); };
  const A = rd(aPath), B = rd(bPath), out = new Float64Array(A.length);
  const f = [a => Math.sin(a), a => Math.cos(a), (a, b) => Math.atan2(a, b), a => Math.asin(a), a => Math.log(a),
    (a, b) => Math.pow(a, b), a => Math.round(a), a => Math.atan(a), (a, b) => fdlibmPow(a, b)][+fn];
  for (let i = 0; i < A.length; i++) out[i] = f(A[i], B[i]);
  fs.writeFileSync(outPath, Buffer.from(out.buffer));
}

// The reference's frame loop on its own terms (main.ts:318, 434-548 minus the DOM): a pool of
// N = max(1, min(4, cores - 1)) workers (main.ts:318), per frame one Job per worker over contiguous ceil(H/N)-row
// tiles (main.ts:444-449), tile buffers transferred back and set into the frame buffers (main.ts:461-468), then
// ShadingModel.shade (main.ts:493-501) and the diagnostics pass (main.ts:528-548) on the main thread.  `frames`
// whole frames are timed after one warm-up frame.  cfg.rowStride = k > 1 renders every k-th row of every tile
// (a bounded sample spread over the whole frame; the caller scales the time by k and says so).  Like the
// reference's worker (raymarchWorker.ts:37-38) every job builds its scene anew (once here, twice there).
function poolWorker() {
  const { parentPort } = require('worker_threads');
  parentPort.on('message', (job) => {
    const S = makeScene(presetSpheres(job.scenePresetIndex), job.accelerationStructure, false);
    const cam = cameraMatrix(job.camera.pitch, job.camera.yaw);
    const k = job.rowStride || 1;
    const rowsList = [];
    for (let y = job.yStart; y < job.yEnd; y += k) rowsList.push(y);
    const W = job.width, n = rowsList.length;
    const depth = new Uint8ClampedArray(W * n), normal = new Uint8ClampedArray(W * n * 3);
    const sdf = new Uint16Array(W * n), iters = new Uint16Array(W * n);
    gTime = job.time || 0;
    if (k === 1) {
      const r = renderTile(S, cam, W, job.height, job.yStart, job.yEnd, job.algorithm);
      depth.set(r.depth); normal.set(r.normal); sdf.set(r.sdf); iters.set(r.iters);
    } else {
      rowsList.forEach((y, j) => {
        const r = renderTile(S, cam, W, job.height, y, y + 1, job.algorithm);
        depth.set(r.depth, j * W); normal.set(r.normal, j * W * 3); sdf.set(r.sdf, j * W); iters.set(r.iters, j * W);
      });
    }
    parentPort.postMessage({ yStart: job.yStart, yEnd: job.yEnd, rows: n, depth, normal, sdfEval: sdf, iters },
      [depth.buffer, normal.buffer, sdf.buffer, iters.buffer]);
  });
}

function cmdPool(cfgPath) {
  const { Worker } = require('worker_threads');
  const os = require('os');
  const cfg = JSON.parse(fs.readFileSync(cfgPath, 'utf8'));
  const cores = cfg.cores || os.cpus().length;
  const N = cfg.workers || Math.max(1, Math.min(4, cores - 1)); // main.ts:318
  const W = cfg.width, H = cfg.height, k = cfg.rowStride || 1, frames = cfg.frames || 5;
  const workers = [];
  for (let i = 0; i < N; i++) workers.push(new Worker(__filename, { argv: ['pool-worker'] }));
  const rowsPerWorker = Math.ceil(H / N); // main.ts:444
  const tiles = [];
  let sampled = 0;
  for (let i = 0; i < N; i++) {
    const y0 = Math.min(i * rowsPerWorker, H), y1 = Math.min((i + 1) * rowsPerWorker, H); // main.ts:448-449
    tiles.push([y0, y1, sampled]);
    sampled += Math.ceil(Math.max(0, y1 - y0) / k);
  }
  const depthB = new Uint8ClampedArray(W * sampled), normalB = new Uint8ClampedArray(W * sampled * 3);
  const sdfB = new Uint16Array(W * sampled), itersB = new Uint16Array(W * sampled), out = new Uint8ClampedArray(W * sampled * 4);
  let diag = null;
  const frame = (f) => Promise.all(workers.map((w, i) => new Promise((resolve) => {
    w.once('message', (r) => {
      const off = tiles[i][2] * W; // r.yStart * width in the reference; sampled rows are packed here
      depthB.set(r.depth, off); normalB.set(r.normal, off * 3); sdfB.set(r.sdfEval, off); itersB.set(r.iters, off);
      resolve();
    });
    w.postMessage({ width: W, height: H, time: 0, yStart: tiles[i][0], yEnd: tiles[i][1], rowStride: k,
      camera: { pitch: cfg.pitch || 0, yaw: cfg.yaw || 0 }, algorithm: cfg.algorithm || 'sphere-tracer',
      scenePresetIndex: cfg.preset, accelerationStructure: cfg.accel });
  }))).then(() => {
    const rgba = shade(cfg.shader || 'normal', depthB, normalB, sdfB, itersB, W, sampled); // main.ts:493-501
    out.set(rgba);
    let sumS = 0, sumI = 0, mx = 0, mn = Number.MAX_SAFE_INTEGER; // main.ts:528-548
    for (let i = 0; i < sdfB.length; i++) { const c = sdfB[i]; sumS += c; sumI += itersB[i]; if (c > mx) mx = c; if (c < mn) mn = c; }
    diag = { sum_sdf: sumS, sum_iters: sumI, max_sdf: mx, min_sdf: mn };
  });
  (async () => {
    await frame(-1); // warm-up (JIT, scene build code paths)
    const times = [];
    for (let f = 0; f < frames; f++) {
      const t0 = process.hrtime.bigint();
      await frame(f);
      times.push(Number(process.hrtime.bigint() - t0) / 1e6);
    }
    await Promise.all(workers.map((w) => w.terminate()));
    process.stdout.write(JSON.stringify({ workers: N, cores, frames, row_stride: k, sampled_rows: sampled, frame_ms: times,
      diagnostics: diag, engine: 'node ' + process.version + ' v8 ' + process.versions.v8 }) + '\n');
  })();
}

const [cmd, a1, a2, a3, a4] = process.argv.slice(2);
if (cmd === 'pool-worker') poolWorker();
else if (cmd === 'pool') cmdPool(a1);
else if (cmd === 'render') cmdRender(a1, a2);
else if (cmd === 'jsmath') cmdJsMath(a1, a2, a3, a4);
else if (cmd === 'hypot') cmdHypot(a1, a2);
else if (cmd === 'camera') cmdCamera(a1, a2);
else { process.stderr.write('usage: node rm_oracle.js render|pool|hypot|camera|jsmath ...\n'); process.exit(2); }
