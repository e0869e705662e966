'use strict';
/*
 * run.js -- reference-control-flow cross-check (VERDICT r1 #7, SURVEY 8c "optional cross-check").  BUILD CONTAINER ONLY.
 *
 * What it does: reads the reference's TypeScript sources as TEXT at run time from the directory given on the command line
 * (/root/reference/src -- never copied, never written anywhere), removes the type syntax in memory (strip_ts.js), resolves
 * `gl-matrix` to glmatrix_shim.js, evaluates the modules under node, and runs the render path the way the reference's
 * worker does (raymarchWorker.ts:33-92: new Scene(accel), loadPreset, camera.setAngles, the algorithm's runRaymarcher),
 * then the shading model (main.ts:33-45,493-501) and the diagnostics pass (main.ts:528-548).  It prints ONLY output
 * vectors' digests: SHA-256 per buffer and the counter sums, as JSON on stdout.
 *
 * What it is NOT: it is not oracle/_ref and not a reference build.  gl-matrix 3.4.4 is absent from this image, so the
 * arithmetic library underneath is a restatement (the shim) -- a stand-in for a library the image lacks -- and the
 * result cannot pin parity (DESIGN.md 5 keeps "parity unpinned").  Its purpose is narrower: both oracle restatements
 * (rm_oracle.c, rm_oracle.js) come from one author's reading of bvh.ts / octree.ts / scene.ts / the marchers; this runs
 * the reference's OWN statements for that control flow, so a misreading would show up as a digest mismatch.
 *
 * usage: node run.js <reference src dir> <config.json>
 *   config: {preset, accel, width, height, yStart?, yEnd?, pitch?, yaw?, algorithm?, overshootFactor?, stepSize?, time?, shader?}
 */
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const vm = require('vm');
const { strip } = require('./strip_ts.js');
const shim = require('./glmatrix_shim.js');

const SRC = path.resolve(process.argv[2]);
const cfg = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'));
const cache = new Map();

function load(absNoExt) {
  const file = absNoExt.endsWith('.ts') ? absNoExt : absNoExt + '.ts';
  if (cache.has(file)) return cache.get(file);
  const exports_ = {};
  cache.set(file, exports_);  // circular imports see the partially filled object, as with real ES module bindings here
  const text = fs.readFileSync(file, 'utf8');
  const js = strip(text, path.relative(SRC, file));
  const require_ = (spec) => {
    if (spec === 'gl-matrix') return shim;
    if (spec.startsWith('.')) return load(path.resolve(path.dirname(file), spec));
    throw new Error('module ' + spec + ' is not on the render path (imported by ' + file + ')');
  };
  let fn;
  try {
    fn = new vm.Script('(function (require_, exports_) {"use strict";\n' + js + '\n})', { filename: 'stripped:' + path.relative(SRC, file) }).runInThisContext();
  } catch (e) {
    process.stderr.write('type stripping left invalid JavaScript in ' + file + ': ' + e.message + '\n');
    if (process.env.RM_XCHECK_DUMP) fs.writeFileSync(process.env.RM_XCHECK_DUMP, js);
    process.exit(3);
  }
  fn(require_, exports_);
  return exports_;
}

// sceneManager.ts touches `document` only in populateSceneDropdown (never called here)
const { Scene } = load(path.join(SRC, 'util', 'scene'));
const algs = {
  'sphere-tracer': () => new (load(path.join(SRC, 'cpu_algorithms', 'sphereTracer')).SphereTracer)(),
  'fixed-step': () => new (load(path.join(SRC, 'cpu_algorithms', 'fixedStep')).FixedStep)(cfg.stepSize),
  'adaptive-step': () => new (load(path.join(SRC, 'cpu_algorithms', 'adaptiveStep')).AdaptiveStep)(),
  'adaptive-step-v2': () => new (load(path.join(SRC, 'cpu_algorithms', 'adaptiveStepV2')).AdaptiveStepV2)(cfg.overshootFactor),
  'adaptive-step-v3': () => new (load(path.join(SRC, 'cpu_algorithms', 'adaptiveStepV3')).AdaptiveStepV3)(cfg.overshootFactor),
};
const shaders = {
  phong: () => new (load(path.join(SRC, 'util', 'shading_models', 'phongModel')).PhongModel)(),
  'sdf-heatmap': () => new (load(path.join(SRC, 'util', 'shading_models', 'SDFHeatmap')).SDFHeatmap)(),
  'iteration-heatmap': () => new (load(path.join(SRC, 'util', 'shading_models', 'IterationHeatmap')).IterationHeatmap)(),
  normal: () => new (load(path.join(SRC, 'util', 'shading_models', 'normalModel')).NormalModel)(),
};

// raymarchWorker.ts:33-81
const width = cfg.width, height = cfg.height;
const yStart = cfg.yStart === undefined ? 0 : cfg.yStart, yEnd = cfg.yEnd === undefined ? height : cfg.yEnd;
const scene = new Scene(cfg.accel);
scene.loadPreset(cfg.preset);
scene.camera.setAngles(cfg.pitch || 0, cfg.yaw || 0);
const tileHeight = Math.max(0, yEnd - yStart);
const depth = new Uint8ClampedArray(width * tileHeight);
const normal = new Uint8ClampedArray(width * tileHeight * 3);
const sdfEval = new Uint16Array(width * tileHeight);
const iters = new Uint16Array(width * tileHeight);
const alg = (algs[cfg.algorithm] || algs['sphere-tracer'])();
const t0 = process.hrtime.bigint();
alg.runRaymarcher(scene, depth, normal, sdfEval, iters, width, height, cfg.time || 0, yStart, yEnd);
const ms = Number(process.hrtime.bigint() - t0) / 1e6;
// main.ts:493-501
const shaded = new Uint8ClampedArray(width * tileHeight * 4);
(shaders[cfg.shader] || shaders.normal)().shade(shaded, depth, normal, sdfEval, iters, width, tileHeight);
// main.ts:528-548
let totalSDFCalls = 0, totalIterations = 0, maxSDFCalls = 0, minSDFCalls = Number.MAX_SAFE_INTEGER;
for (let i = 0; i < sdfEval.length; i++) {
  const c = sdfEval[i];
  totalSDFCalls += c; totalIterations += iters[i];
  if (c > maxSDFCalls) maxSDFCalls = c;
  if (c < minSDFCalls) minSDFCalls = c;
}
const sha = (a) => crypto.createHash('sha256').update(Buffer.from(a.buffer, a.byteOffset, a.byteLength)).digest('hex');
process.stdout.write(JSON.stringify({
  sha256: { depth: sha(depth), normal: sha(normal), sdf: sha(sdfEval), iters: sha(iters), rgba: sha(shaded) },
  total_sdf: totalSDFCalls, total_iters: totalIterations, max_sdf: maxSDFCalls, min_sdf: minSDFCalls,
  n_objects: scene.objectSDFs.length, render_ms: ms, modules: cache.size,
}) + '\n');
