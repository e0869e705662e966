'use strict';
// node native/pool_cli.js <config.json> <outdir>: the reference's worker pool (main.ts:318-321) and per-frame
// fan-out / fan-in (main.ts:444-468) over worker_threads, every worker rendering its row tile through the N-API
// addon with ITS OWN rm_ctx (the addon keeps the ctx in the env's instance data).  cfg.workers workers,
// cfg.frames frames; cfg.device = -1 runs the lifetime logic on host-only contexts (every renderTile then returns
// RM_E_NO_DEVICE = -3, which is what the test expects: no worker may see another status, a crash or a stale ctx).
const { Worker, isMainThread, parentPort, workerData } = require('worker_threads');
const fs = require('fs');
const path = require('path');

if (!isMainThread) {
  const native = require(path.join(__dirname, 'build', 'rm_addon.node'));
  let rc0 = native.create(workerData.device);
  parentPort.on('message', (job) => {
    if (job.recreate) rc0 = native.create(workerData.device);  // a worker may re-create its own ctx at any time
    const h = Math.max(0, job.yEnd - job.yStart);
    const depth = new Uint8ClampedArray(job.width * h), normal = new Uint8ClampedArray(job.width * h * 3);
    const sdfEval = new Uint16Array(job.width * h), iters = new Uint16Array(job.width * h);
    const rc = rc0 !== 0 ? rc0 : native.renderTile(job, depth, normal, sdfEval, iters);
    parentPort.postMessage({ rc, err: rc ? native.lastError() : '', yStart: job.yStart, yEnd: job.yEnd, depth, normal, sdfEval, iters },
      [depth.buffer, normal.buffer, sdfEval.buffer, iters.buffer]);  // raymarchWorker.ts:83-91
  });
} else {
  const cfg = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
  const out = process.argv[3];
  const W = cfg.width, H = cfg.height, N = cfg.workers || 4, frames = cfg.frames || 1;
  const device = cfg.device === undefined ? 0 : cfg.device;
  const workers = [];
  for (let i = 0; i < N; ++i) workers.push(new Worker(__filename, { workerData: { device } }));
  const depthB = new Uint8ClampedArray(W * H), normalB = new Uint8ClampedArray(W * H * 3);
  const sdfB = new Uint16Array(W * H), itersB = new Uint16Array(W * H);
  const statuses = [];
  const frame = (f) => {
    const rowsPerWorker = Math.ceil(H / N);  // main.ts:444
    return Promise.all(workers.map((w, i) => new Promise((resolve) => {
      const yStart = Math.min(i * rowsPerWorker, H), yEnd = Math.min((i + 1) * rowsPerWorker, H);  // main.ts:448-449
      w.once('message', (r) => {
        statuses.push(r.rc);
        if (r.rc === 0) {  // main.ts:461-468
          depthB.set(r.depth, r.yStart * W); normalB.set(r.normal, r.yStart * W * 3);
          sdfB.set(r.sdfEval, r.yStart * W); itersB.set(r.iters, r.yStart * W);
        }
        resolve();
      });
      w.postMessage({ width: W, height: H, time: cfg.time || 0, yStart, yEnd, camera: { pitch: cfg.pitch || 0, yaw: (cfg.yaw || 0) + 0.015 * f },
        algorithm: cfg.algorithm || 'sphere-tracer', scenePresetIndex: cfg.preset, accelerationStructure: cfg.accel,
        recreate: !!cfg.recreate && ((f + i) % 2 === 1) });
    })));
  };
  (async () => {
    for (let f = 0; f < frames; ++f) await frame(f);
    await Promise.all(workers.map((w) => w.terminate()));
    if (out) {
      fs.mkdirSync(out, { recursive: true });
      const wr = (n, a) => fs.writeFileSync(path.join(out, n), Buffer.from(a.buffer, a.byteOffset, a.byteLength));
      wr('depth.bin', depthB); wr('normal.bin', normalB); wr('sdf.bin', sdfB); wr('iters.bin', itersB);
    }
    process.stdout.write(JSON.stringify({ workers: N, frames, statuses }) + '\n');
  })();
}
