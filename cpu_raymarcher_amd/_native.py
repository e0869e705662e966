"""Loader for the C-ABI library librm_hip.so (include/rm_raymarch.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be loaded
this module raises, loudly.  `build()` compiles it in-tree with hipcc for gfx950.
"""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
# RM_HIP_LIB: another build of the same library (A/B measurements of two builds in one process each)
LIB_PATH = os.environ.get("RM_HIP_LIB") or os.path.join(_PKG, "librm_hip.so")
CSRC = os.path.join(_PKG, "csrc")

RM_OK = 0
RM_E_INVALID = -1
RM_E_UNSUPPORTED = -2
RM_E_NO_DEVICE = -3
RM_E_HIP = -4
RM_E_NO_SCENE = -5
RM_E_NOMEM = -6
RM_SCENE_UPLOADED = -(2 ** 31)

_STATUS_NAMES = {0: "RM_OK", -1: "RM_E_INVALID", -2: "RM_E_UNSUPPORTED", -3: "RM_E_NO_DEVICE",
                 -4: "RM_E_HIP", -5: "RM_E_NO_SCENE", -6: "RM_E_NOMEM"}


class RmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (_STATUS_NAMES.get(code, "?"), code, msg))
        self.code = code


class RmUnsupported(RmError):
    """RM_E_UNSUPPORTED: the host may fall back to its own CPU path."""


class rm_job(C.Structure):  # include/rm_raymarch.h: struct rm_job
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("time", C.c_double),
                ("y_start", C.c_int32), ("y_end", C.c_int32),
                ("camera_pitch", C.c_double), ("camera_yaw", C.c_double),
                ("algorithm", C.c_int32), ("scene_preset_index", C.c_int32),
                ("acceleration_structure", C.c_int32), ("reserved", C.c_int32),
                ("overshoot_factor", C.c_double), ("step_size", C.c_double)]


class rm_scene_info(C.Structure):
    _fields_ = [("n_prims", C.c_int32), ("accel", C.c_int32), ("preset_index", C.c_int32),
                ("bvh_nodes", C.c_int32), ("bvh_leaves", C.c_int32), ("bvh_depth", C.c_int32),
                ("oct_nodes", C.c_int32), ("oct_leaves", C.c_int32), ("oct_empty_leaves", C.c_int32),
                ("oct_max_leaf_prims", C.c_int32), ("root_min", C.c_float * 3), ("root_max", C.c_float * 3),
                ("nodes_in_lds", C.c_int32), ("reserved", C.c_int32)]


class rm_node(C.Structure):  # include/rm_raymarch.h: struct rm_node
    _fields_ = [("type", C.c_int32), ("child_a", C.c_int32), ("child_b", C.c_int32), ("reserved", C.c_int32),
                ("world_to_local", C.c_float * 16), ("params", C.c_double * 6)]


class rm_prim(C.Structure):  # include/rm_raymarch.h: struct rm_prim
    _fields_ = [("type", C.c_int32), ("reserved", C.c_int32), ("world_to_local", C.c_float * 16),
                ("params", C.c_double * 3), ("reserved2", C.c_double)]


class rm_diagnostics(C.Structure):
    _fields_ = [("total_sdf_calls", C.c_uint64), ("total_iterations", C.c_uint64),
                ("max_sdf_calls", C.c_uint32), ("min_sdf_calls", C.c_uint32), ("total_pixels", C.c_uint64)]


# every symbol include/rm_raymarch.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
SIGNATURES = {
    "rm_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "rm_destroy": (None, [_VP]),
    "rm_last_error": (C.c_char_p, [_VP]),
    "rm_last_kernel": (C.c_char_p, [_VP]),
    "rm_version": (C.c_char_p, []),
    "rm_algorithm_from_string": (C.c_int, [C.c_char_p]),
    "rm_accel_from_string": (C.c_int, [C.c_char_p]),
    "rm_shader_from_string": (C.c_int, [C.c_char_p]),
    "rm_preset_count": (C.c_int, []),
    "rm_scene_from_preset": (C.c_int, [_VP, C.c_int32, C.c_int32]),
    "rm_scene_from_spheres": (C.c_int, [_VP, _VP, _VP, C.c_int32, C.c_int32]),
    "rm_scene_from_prims": (C.c_int, [_VP, C.POINTER(rm_prim), C.c_int32, C.c_int32]),
    "rm_make_transform": (C.c_int, [C.c_double, C.c_double, C.c_double, _VP, _VP]),
    "rm_scene_from_nodes": (C.c_int, [_VP, C.POINTER(rm_node), C.c_int32, _VP, C.c_int32, C.c_int32]),
    "rm_scale_transform": (C.c_int, [_VP, C.c_double, C.c_double, C.c_double]),
    "rm_scene_set_time": (C.c_int, [_VP, C.c_double]),
    "rm_selftest_jsmath": (C.c_int, [_VP, C.c_int32, _VP, _VP, C.c_int64, _VP]),
    "rm_scene_get_info": (C.c_int, [_VP, C.POINTER(rm_scene_info)]),
    "rm_camera_from_angles": (C.c_int, [C.c_double, C.c_double, _VP, _VP]),
    "rm_scene_distance": (C.c_int, [_VP, _VP, C.c_int64, _VP, _VP]),
    "rm_render_tile": (C.c_int, [_VP, C.POINTER(rm_job), _VP, _VP, _VP, _VP]),
    "rm_render_tile_device": (C.c_int, [_VP, C.POINTER(rm_job), C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP]),
    "rm_render_stripes_device": (C.c_int, [_VP, C.POINTER(rm_job), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           _VP, _VP, _VP, _VP, _VP, _VP]),
    "rm_stripe_rows": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "rm_deal_stripes": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _VP, _VP]),
    "rm_render_stripe_list_device": (C.c_int, [_VP, C.POINTER(rm_job), C.c_int32, C.c_int32, _VP, C.c_int32,
                                               _VP, _VP, _VP, _VP, _VP, _VP]),
    "rm_assemble_frame_device": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _VP,
                                           C.c_int32, C.c_int32, _VP, C.c_int64, _VP, _VP]),
    "rm_shade": (C.c_int, [_VP, C.c_int32, C.c_int32, C.c_int32, _VP, _VP, _VP, _VP, _VP]),
    "rm_shade_device": (C.c_int, [_VP, C.c_int32, C.c_int32, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP]),
    "rm_reduce_counters": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.POINTER(rm_diagnostics)]),
    "rm_reduce_counters_device": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.POINTER(rm_diagnostics), _VP]),
    "rm_reduce_counters_enqueue": (C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, _VP]),
    "rm_render_attach_diagnostics": (C.c_int, [_VP, _VP]),
    "rm_partition_rows": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rm_selftest_hypot": (C.c_int, [_VP, _VP, C.c_int64, _VP]),
    "rm_selftest_fastdiv": (C.c_int, [_VP, C.c_uint64, C.c_int64, C.POINTER(C.c_uint64)]),
    "rm_selftest_recip": (C.c_int, [_VP, C.c_int, _VP]),
    "rm_debug_read_stamps": (C.c_int, [_VP, _VP]),
    "rm_debug_read_wave_times": (C.c_int, [_VP, _VP]),
    "rm_debug_read_batch_log": (C.c_int, [_VP, _VP]),
    "rm_debug_read_counts": (C.c_int, [_VP, _VP]),
    "rm_debug_read_lpt_costs": (C.c_int, [_VP, _VP, C.c_int64]),
    "rm_rtc_source": (C.c_int, [_VP, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]),
    "rm_rtc_compile_check": (C.c_int, [_VP, C.c_int32, C.c_int32, C.c_char_p, C.c_int64, C.POINTER(C.c_double)]),
    "rm_rtc_status": (C.c_int, [_VP, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_int64]),
    "rm_set_option": (C.c_int, [_VP, C.c_char_p, C.c_int64]),
    "rm_get_option": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_int64)]),
}

_lib = None


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> cpu_raymarcher_amd/librm_hip.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_PKG), "include", "rm_raymarch.h"))
    if not force and os.path.exists(LIB_PATH) and all(
            os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    cmd = ["make", "-j%d" % min(8, os.cpu_count() or 1), "-C", CSRC, "all"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def lib():
    """The loaded library; raises if it is absent (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "cpu_raymarcher_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C cpu_raymarcher_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7, and a
        # second runtime opened later finds no GPU ("No HIP GPUs are available").  Loading
        # torch first makes librm_hip.so's NEEDED libamdhip64.so.7 resolve to that copy.
        # Pure C / N-API consumers (no torch) get /opt/rocm's runtime through RUNPATH.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("RM_HIP_LIB") and not hasattr(L, name):
                continue  # A/B measurements against an OLDER build of the library (scripts/kbench.py): newer entry points are absent there
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(ctx, rc):
    if rc == RM_OK:
        return
    msg = lib().rm_last_error(ctx).decode() if ctx else ""
    raise (RmUnsupported if rc == RM_E_UNSUPPORTED else RmError)(rc, msg)
