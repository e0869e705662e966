'use strict';
// node native/render_cli.js <config.json> <outdir>: renders one Job through the N-API addon with
// the reference's own call sequence (worker onmessage -> shade -> diagnostics) and writes the
// five buffers, for tests/test_napi_host.py to compare with the oracle.
const fs = require('fs');
const path = require('path');
const R = require('./host/raymarcher.js');
const cfg = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const out = process.argv[3];
const H = cfg.height, W = cfg.width;
const job = { width: W, height: H, time: cfg.time || 0, yStart: cfg.yStart === undefined ? 0 : cfg.yStart,
  yEnd: cfg.yEnd === undefined ? H : cfg.yEnd, camera: { pitch: cfg.pitch || 0, yaw: cfg.yaw || 0 },
  algorithm: cfg.algorithm || 'sphere-tracer', scenePresetIndex: cfg.preset, accelerationStructure: cfg.accel,
  overshootFactor: cfg.overshootFactor, stepSize: cfg.stepSize };
let res;
try { res = R.onmessage(job); } catch (e) { process.stdout.write(JSON.stringify({ error: e.message, code: e.code }) + '\n'); process.exit(e.code === -2 ? 3 : 1); }
const rows = job.yEnd - job.yStart;
const rgba = new Uint8ClampedArray(W * rows * 4);
R.createShadingModelFromValue(cfg.shader).shade(rgba, res.depth, res.normal, res.sdfEval, res.iters, W, rows);
const d = R.diagnostics(res.sdfEval, res.iters);
fs.mkdirSync(out, { recursive: true });
const w = (n, a) => fs.writeFileSync(path.join(out, n), Buffer.from(a.buffer, a.byteOffset, a.byteLength));
w('depth.bin', res.depth); w('normal.bin', res.normal); w('sdf.bin', res.sdfEval); w('iters.bin', res.iters); w('rgba.bin', rgba);
process.stdout.write(JSON.stringify(d) + '\n');
