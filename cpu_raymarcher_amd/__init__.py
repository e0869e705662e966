"""cpu_raymarcher_amd -- MI355X-native drop-in for the per-pixel sphere-tracing render path of
vxlerian/cpu-raymarcher (hand-written HIP for gfx950 behind the C ABI of include/rm_raymarch.h).

Importing this package loads cpu_raymarcher_amd/librm_hip.so and fails loudly if it is not
built: there is no CPU fallback on the product path.
"""
from . import _native
from ._native import RmError, RmUnsupported, RM_SCENE_UPLOADED

_native.lib()  # raise now, not at first render, when the HIP library is missing

from .context import Context, camera_from_angles, make_transform, partition_rows, scale_transform  # noqa: E402
from .host import (ALGORITHMS, SHADERS, AdaptiveStep, AdaptiveStepV2, AdaptiveStepV3, Camera, FixedStep,  # noqa: E402
                   IterationHeatmap, Job, NormalModel, PhongModel, Raymarcher, RaymarchWorker, Result, Scene,
                   SDFHeatmap, ShadingModel, SphereTracer, createRaymarcher, createShadingModelFromValue,
                   diagnostics, renderFrame)

__all__ = [n for n in dir() if not n.startswith("_")]
