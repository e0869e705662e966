"""Host-side mirror of the reference's plugin surface for the render path, on top of the C ABI.

Same names, argument meaning and defaulting as the TypeScript (paths relative to the
reference's src/), so parity tests read like the reference's call sites:

  Scene(accelerationStructure)        util/scene.ts:24-59      loadPreset, camera, getDistance
  Camera.setAngles / rotateCamera     util/camera.ts:27-62
  SphereTracer().runRaymarcher(...)   cpu_algorithms/raymarcher.ts:46-57
  createShadingModelFromValue(name)   main.ts:33-45  -> NormalModel | PhongModel | SDFHeatmap | IterationHeatmap
  ShadingModel.shade(...)             util/shading_models/shadingModel.ts:8-17
  Job / Result / onmessage            workers/raymarchWorker.ts:10-92
  renderFrame (fan-out / fan-in)      main.ts:444-468,493-501
  diagnostics                         main.ts:528-548

Buffers are numpy arrays (host entry points) or torch CUDA tensors (device entry points,
asynchronous on torch's current stream).
"""
import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _native as N
from .context import Context, _is_torch, partition_rows

ALGORITHMS = ("sphere-tracer", "fixed-step", "adaptive-step", "adaptive-step-v2", "adaptive-step-v3")
SHADERS = ("normal", "phong", "sdf-heatmap", "iteration-heatmap")


class Camera:
    """util/camera.ts: orbit camera; the matrices themselves are derived inside the library."""

    def __init__(self):
        self.pitch = 0.0
        self.yaw = 0.0

    def setAngles(self, pitch, yaw):  # camera.ts:58-62
        self.pitch = min(max(pitch, -math.pi / 2), math.pi / 2)
        self.yaw = yaw

    def rotateCamera(self, pitch, yaw):  # camera.ts:27-32
        self.pitch = min(max(self.pitch + pitch, -math.pi / 2), math.pi / 2)
        self.yaw += yaw

    def getAngles(self):  # camera.ts:35-37
        return [self.pitch, self.yaw]


class Scene:
    """util/scene.ts Scene: primitives + camera + chosen acceleration structure.  The built
    structure lives in HBM inside `ctx`; loadPreset(i) rebuilds only on change."""

    def __init__(self, accelerationStructure="None", ctx: Optional[Context] = None, device=0):
        self.ctx = ctx if ctx is not None else Context(device)
        self.accelerationStructure = accelerationStructure
        self.camera = Camera()
        self._uploaded = False
        self.currentPresetIndex = 0
        self.loadPreset(0)  # scene.ts:28

    @property
    def accel_enum(self):
        return N.lib().rm_accel_from_string(str(self.accelerationStructure).encode())

    def loadPreset(self, index):  # scene.ts:38-59
        count = N.lib().rm_preset_count()
        self.currentPresetIndex = max(0, min(int(index), count - 1))
        self._uploaded = False
        self.ctx.scene_from_preset(self.currentPresetIndex, self.accel_enum)

    def loadSpheres(self, centers, radii):
        """Build-defined: n spheres as SceneManager.createSphere(x,y,z,r) makes them."""
        self.ctx.scene_from_spheres(centers, radii, self.accel_enum)
        self._uploaded = True

    def loadPrims(self, prims):
        """Build-defined: spheres / boxes / tori as (type, world_to_local[16], params)."""
        self.ctx.scene_from_prims(prims, self.accel_enum)
        self._uploaded = True

    def loadNodes(self, nodes, roots):
        """Build-defined: an SDF expression forest (operators of util/primitive_operations/ over
        sphere / box / torus / mandelbulb leaves); see Context.scene_from_nodes."""
        self.ctx.scene_from_nodes(nodes, roots, self.accel_enum)
        self._uploaded = True

    def updateTime(self, time):  # scene.ts:135-140
        self.ctx.scene_set_time(time)

    @property
    def preset_for_job(self):
        return N.RM_SCENE_UPLOADED if self._uploaded else self.currentPresetIndex

    def getDistance(self, position, sdfEvaluationCounter=None):  # scene.ts:144-190
        self._activate()
        d, c = self.ctx.scene_distance(np.asarray(position, np.float32).reshape(1, 3))
        if sdfEvaluationCounter is not None:
            sdfEvaluationCounter["count"] += int(c[0])
        return float(d[0])

    def getDistances(self, points):
        self._activate()
        return self.ctx.scene_distance(points)

    def info(self):
        self._activate()
        return self.ctx.scene_info()

    def _activate(self):
        if self._uploaded:
            return  # rm_scene_from_spheres already made it active
        self.ctx.scene_from_preset(self.currentPresetIndex, self.accel_enum)


def _job(scene, width, height, time, yStart, yEnd, algorithm, overshootFactor=None, stepSize=None):
    j = N.rm_job()
    j.width, j.height, j.time = int(width), int(height), float(time)
    j.y_start, j.y_end = int(yStart), int(yEnd)
    j.camera_pitch, j.camera_yaw = float(scene.camera.pitch), float(scene.camera.yaw)
    j.algorithm = N.lib().rm_algorithm_from_string(str(algorithm).encode())
    j.scene_preset_index = scene.preset_for_job
    j.acceleration_structure = scene.accel_enum
    # None mirrors JS `undefined` (constructor defaults 1.2 / 0.1); the ABI encodes it as NaN
    j.overshoot_factor = float(overshootFactor) if overshootFactor is not None else float("nan")
    j.step_size = float(stepSize) if stepSize is not None else float("nan")
    return j


class Raymarcher:
    """cpu_algorithms/raymarcher.ts:19-57 (abstract); subclasses name the algorithm."""
    algorithm = "sphere-tracer"

    def getMaxDistance(self):
        return 10

    def runRaymarcher(self, scene, depthBuffer, normalBuffer, SDFevaluationBuffer, iterationsBuffer,
                      width, height, time=0.0, yStart=0, yEnd=None, shadedBuffer=None, shader="normal", diagnostics=None):
        """raymarcher.ts:46-57.  shadedBuffer/shader are an extension: fused ShadingModel.shade; so is `diagnostics`
        (device buffers only): a CUDA tensor of 32 bytes that receives the sums / max / min of main.ts:528-548 over the
        rendered rows from the render kernel itself (rm_render_attach_diagnostics; read with Context.decode_acc)."""
        if yEnd is None:
            yEnd = height
        job = _job(scene, width, height, time, yStart, yEnd, self.algorithm,
                   getattr(self, "overshootFactor", None), getattr(self, "stepSize", None))
        sh = N.lib().rm_shader_from_string(str(shader).encode())
        scene.ctx.render_tile(job, depthBuffer, normalBuffer, SDFevaluationBuffer, iterationsBuffer,
                              rgba=shadedBuffer, shader=sh, diag=diagnostics)


class SphereTracer(Raymarcher):  # cpu_algorithms/sphereTracer.ts
    algorithm = "sphere-tracer"


class FixedStep(Raymarcher):  # cpu_algorithms/fixedStep.ts
    algorithm = "fixed-step"

    def __init__(self, stepSize=None):
        self.stepSize = stepSize


class AdaptiveStep(Raymarcher):  # cpu_algorithms/adaptiveStep.ts
    algorithm = "adaptive-step"


class AdaptiveStepV2(Raymarcher):
    algorithm = "adaptive-step-v2"

    def __init__(self, overshootFactor=None):
        self.overshootFactor = overshootFactor


class AdaptiveStepV3(AdaptiveStepV2):
    algorithm = "adaptive-step-v3"


def createRaymarcher(algorithm, overshootFactor=None, stepSize=None):
    """raymarchWorker.ts:49-68 (unknown -> SphereTracer)."""
    if algorithm == "fixed-step":
        return FixedStep(stepSize)
    if algorithm == "adaptive-step":
        return AdaptiveStep()
    if algorithm == "adaptive-step-v2":
        return AdaptiveStepV2(overshootFactor)
    if algorithm == "adaptive-step-v3":
        return AdaptiveStepV3(overshootFactor)
    return SphereTracer()


class ShadingModel:
    """util/shading_models/shadingModel.ts:8-17."""
    name = "normal"

    def __init__(self, ctx: Optional[Context] = None, device=0):
        self._ctx = ctx
        self._device = device

    def shade(self, shadedBuffer, depthBuffer, normalBuffer, SDFevaluationBuffer, iterationsBuffer, width, height):
        if self._ctx is None:
            self._ctx = Context(self._device)
        sh = N.lib().rm_shader_from_string(self.name.encode())
        self._ctx.shade(sh, width, height, depthBuffer, normalBuffer, SDFevaluationBuffer, iterationsBuffer,
                        shadedBuffer)
        return shadedBuffer


class NormalModel(ShadingModel):
    name = "normal"


class PhongModel(ShadingModel):
    name = "phong"


class SDFHeatmap(ShadingModel):
    name = "sdf-heatmap"


class IterationHeatmap(ShadingModel):
    name = "iteration-heatmap"


def createShadingModelFromValue(selectedModel, ctx=None):  # main.ts:33-45
    return {"phong": PhongModel, "sdf-heatmap": SDFHeatmap,
            "iteration-heatmap": IterationHeatmap}.get(selectedModel, NormalModel)(ctx)


@dataclass
class Job:  # raymarchWorker.ts:10-22
    width: int
    height: int
    time: float = 0.0
    yStart: int = 0
    yEnd: int = 0
    camera: dict = field(default_factory=lambda: {"pitch": 0.0, "yaw": 0.0})
    algorithm: str = "sphere-tracer"
    scenePresetIndex: int = 0
    accelerationStructure: str = "None"
    overshootFactor: Optional[float] = None
    stepSize: Optional[float] = None


@dataclass
class Result:  # raymarchWorker.ts:24-31
    yStart: int
    yEnd: int
    depth: np.ndarray
    normal: np.ndarray
    sdfEval: np.ndarray
    iters: np.ndarray


class RaymarchWorker:
    """workers/raymarchWorker.ts: stateless between jobs as far as results go; the scene the
    job names stays resident in HBM between jobs."""

    def __init__(self, ctx: Optional[Context] = None, device=0):
        self.ctx = ctx if ctx is not None else Context(device)
        self._scene = None

    def onmessage(self, job: Job) -> Result:  # raymarchWorker.ts:33-92
        if self._scene is None or self._scene.accelerationStructure != job.accelerationStructure:
            self._scene = Scene(job.accelerationStructure, ctx=self.ctx)
        scene = self._scene
        scene.loadPreset(job.scenePresetIndex)
        scene.camera.setAngles(job.camera["pitch"], job.camera["yaw"])
        tileHeight = max(0, job.yEnd - job.yStart)
        depth = np.zeros(job.width * tileHeight, np.uint8)
        normal = np.zeros(job.width * tileHeight * 3, np.uint8)
        sdfEval = np.zeros(job.width * tileHeight, np.uint16)
        iters = np.zeros(job.width * tileHeight, np.uint16)
        alg = createRaymarcher(job.algorithm, job.overshootFactor, job.stepSize)
        alg.runRaymarcher(scene, depth, normal, sdfEval, iters, job.width, job.height, job.time, job.yStart, job.yEnd)
        return Result(job.yStart, job.yEnd, depth, normal, sdfEval, iters)


def renderFrame(workers, width, height, scenePresetIndex, accelerationStructure, pitch=0.0, yaw=0.0,
                algorithm="sphere-tracer", time=0.0):
    """main.ts:444-468: partition rows over the workers, gather tiles into frame buffers."""
    n = len(workers)
    depthBuffer = np.zeros(width * height, np.uint8)
    normalBuffer = np.zeros(width * height * 3, np.uint8)
    SDFevaluationBuffer = np.zeros(width * height, np.uint16)
    iterationsBuffer = np.zeros(width * height, np.uint16)
    for i, w in enumerate(workers):
        yStart, yEnd = partition_rows(height, n, i)
        if yStart >= yEnd:
            continue
        r = w.onmessage(Job(width, height, time, yStart, yEnd, {"pitch": pitch, "yaw": yaw}, algorithm,
                            scenePresetIndex, accelerationStructure))
        pixelOffset = r.yStart * width
        depthBuffer[pixelOffset:pixelOffset + r.depth.size] = r.depth
        SDFevaluationBuffer[pixelOffset:pixelOffset + r.sdfEval.size] = r.sdfEval
        iterationsBuffer[pixelOffset:pixelOffset + r.iters.size] = r.iters
        normalBuffer[pixelOffset * 3:pixelOffset * 3 + r.normal.size] = r.normal
    return depthBuffer, normalBuffer, SDFevaluationBuffer, iterationsBuffer


def diagnostics(ctx, SDFevaluationBuffer, iterationsBuffer):
    """main.ts:528-548 on the device: averages as the page shows them."""
    d = ctx.reduce_counters(SDFevaluationBuffer, iterationsBuffer)
    n = max(1, d["total_pixels"])
    d["average_sdf_calls"] = d["total_sdf"] / n
    d["average_iterations"] = d["total_iters"] / n
    return d
