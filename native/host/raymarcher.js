'use strict';
/*
 * JS host mirror of the reference's plugin surface for the render path, on top of the N-API
 * addon (native/rm_addon.cc -> include/rm_raymarch.h).  Same class and method names, argument
 * order and defaulting as the TypeScript (paths relative to the reference's src/):
 *
 *   Scene(accelerationStructure)            util/scene.ts:24-59
 *   Camera.setAngles / rotateCamera         util/camera.ts:27-62
 *   SphereTracer.runRaymarcher(...)         cpu_algorithms/raymarcher.ts:46-57
 *   createShadingModelFromValue(name)       main.ts:33-45
 *   onmessage(job) -> Result                workers/raymarchWorker.ts:33-92
 *
 * Buffers are the reference's typed arrays.  RM_E_UNSUPPORTED (-2) surfaces as an Error with
 * .code === -2 so a host can keep its TypeScript CPU path for that job.
 */
const path = require('path');
const native = require(path.join(__dirname, '..', 'build', 'rm_addon.node'));

let created = false;
function ensure(device) {
  if (!created) {
    const rc = native.create(device === undefined ? 0 : device);
    if (rc !== 0) { const e = new Error('rm_create failed (' + rc + '): no usable HIP device'); e.code = rc; throw e; }
    created = true;
  }
}
function check(rc) {
  if (rc === 0) return;
  const e = new Error(native.lastError() + ' (' + rc + ')');
  e.code = rc;
  throw e;
}

class Camera { // util/camera.ts
  constructor() { this.pitch = 0; this.yaw = 0; }
  setAngles(pitch, yaw) { this.pitch = Math.min(Math.max(pitch, -Math.PI / 2), Math.PI / 2); this.yaw = yaw; }
  rotateCamera(pitch, yaw) { this.pitch = Math.min(Math.max(this.pitch + pitch, -Math.PI / 2), Math.PI / 2); this.yaw += yaw; }
  getAngles() { return [this.pitch, this.yaw]; }
}

class Scene { // util/scene.ts
  constructor(accelerationStructure = 'None') {
    ensure();
    this.accelerationStructure = accelerationStructure;
    this.camera = new Camera();
    this.currentPresetIndex = 0;
  }
  loadPreset(index) { this.currentPresetIndex = Math.max(0, Math.min(index, 18)); }
}

class Raymarcher { // cpu_algorithms/raymarcher.ts:19-57
  constructor() { this.algorithm = 'sphere-tracer'; }
  getMaxDistance() { return 10; }
  runRaymarcher(scene, depthBuffer, normalBuffer, SDFevaluationBuffer, iterationsBuffer, width, height, time,
    yStart = 0, yEnd = height) {
    check(native.renderTile({
      width, height, time, yStart, yEnd, camera: { pitch: scene.camera.pitch, yaw: scene.camera.yaw },
      algorithm: this.algorithm, scenePresetIndex: scene.currentPresetIndex,
      accelerationStructure: scene.accelerationStructure, overshootFactor: this.overshootFactor, stepSize: this.stepSize,
    }, depthBuffer, normalBuffer, SDFevaluationBuffer, iterationsBuffer));
  }
}
class SphereTracer extends Raymarcher {}
class FixedStep extends Raymarcher { constructor(stepSize) { super(); this.algorithm = 'fixed-step'; this.stepSize = stepSize; } }
class AdaptiveStep extends Raymarcher { constructor() { super(); this.algorithm = 'adaptive-step'; } }
class AdaptiveStepV2 extends Raymarcher { constructor(o) { super(); this.algorithm = 'adaptive-step-v2'; this.overshootFactor = o; } }
class AdaptiveStepV3 extends Raymarcher { constructor(o) { super(); this.algorithm = 'adaptive-step-v3'; this.overshootFactor = o; } }

class ShadingModel { // util/shading_models/shadingModel.ts:8-17
  constructor(name) { this.name = name || 'normal'; }
  shade(shadedBuffer, depthBuffer, normalBuffer, SDFevaluationBuffer, iterationsBuffer, width, height) {
    ensure();
    check(native.shade(this.name, width, height, depthBuffer, normalBuffer, SDFevaluationBuffer, iterationsBuffer, shadedBuffer));
    return shadedBuffer;
  }
}
function createShadingModelFromValue(selectedModel) { // main.ts:33-45
  return new ShadingModel(['phong', 'sdf-heatmap', 'iteration-heatmap'].includes(selectedModel) ? selectedModel : 'normal');
}

function onmessage(job) { // workers/raymarchWorker.ts:33-92
  const scene = new Scene(job.accelerationStructure);
  scene.loadPreset(job.scenePresetIndex);
  scene.camera.setAngles(job.camera.pitch, job.camera.yaw);
  const tileHeight = Math.max(0, job.yEnd - job.yStart);
  const depth = new Uint8ClampedArray(job.width * tileHeight);
  const normal = new Uint8ClampedArray(job.width * tileHeight * 3);
  const sdfEval = new Uint16Array(job.width * tileHeight);
  const iters = new Uint16Array(job.width * tileHeight);
  let alg;
  switch (job.algorithm) {
    case 'fixed-step': alg = new FixedStep(job.stepSize); break;
    case 'adaptive-step': alg = new AdaptiveStep(); break;
    case 'adaptive-step-v2': alg = new AdaptiveStepV2(job.overshootFactor); break;
    case 'adaptive-step-v3': alg = new AdaptiveStepV3(job.overshootFactor); break;
    default: alg = new SphereTracer();
  }
  alg.runRaymarcher(scene, depth, normal, sdfEval, iters, job.width, job.height, job.time, job.yStart, job.yEnd);
  return { yStart: job.yStart, yEnd: job.yEnd, depth, normal, sdfEval, iters };
}

function diagnostics(SDFevaluationBuffer, iterationsBuffer) { // main.ts:528-548
  ensure();
  const d = native.diagnostics(SDFevaluationBuffer, iterationsBuffer);
  if (typeof d === 'number') check(d);
  d.averageSDFCalls = d.totalSDFCalls / Math.max(1, d.totalPixels);
  d.averageIterations = d.totalIterations / Math.max(1, d.totalPixels);
  return d;
}

module.exports = { Camera, Scene, Raymarcher, SphereTracer, FixedStep, AdaptiveStep, AdaptiveStepV2, AdaptiveStepV3,
  ShadingModel, createShadingModelFromValue, onmessage, diagnostics, native };
