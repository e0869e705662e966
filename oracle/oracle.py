"""ctypes wrapper around oracle/librm_oracle.so (the CPU restatement, rm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  PARITY UNPINNED, see rm_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librm_oracle.so")
_lib = None


def build(force=False):
    """Compile rm_oracle.c with gcc (Makefile in this directory)."""
    srcs = [os.path.join(_HERE, f) for f in ("rm_oracle.c", "ro_jsmath.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(f) for f in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "librm_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.ro_hypot3.restype = C.c_double
        L.ro_hypot3.argtypes = [C.c_double] * 3
        L.ro_u8clamp.restype = C.c_uint8
        L.ro_u8clamp.argtypes = [C.c_double]
        L.ro_set_length_mode.argtypes = [C.c_int]
        L.ro_scene_from_preset.restype = C.c_void_p
        L.ro_scene_from_preset.argtypes = [C.c_int, C.c_char_p]
        L.ro_scene_from_spheres.restype = C.c_void_p
        L.ro_scene_from_spheres.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p]
        L.ro_scene_from_prims.restype = C.c_void_p
        L.ro_scene_from_prims.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.ro_scene_prims.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ro_scene_from_nodes.restype = C.c_void_p
        L.ro_scene_from_nodes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_char_p]
        L.ro_scene_nodes.restype = C.c_int
        L.ro_scene_nodes.argtypes = [C.c_void_p] * 6
        L.ro_set_time.argtypes = [C.c_double]
        L.ro_scene_free.argtypes = [C.c_void_p]
        L.ro_scene_set_angles.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.ro_scene_distance.restype = C.c_double
        L.ro_scene_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ro_run_raymarcher.restype = C.c_int
        L.ro_run_raymarcher.argtypes = [C.c_void_p, C.c_char_p] + [C.c_void_p] * 4 + \
            [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
        L.ro_run_raymarcher_ex.restype = C.c_int
        L.ro_run_raymarcher_ex.argtypes = [C.c_void_p, C.c_char_p] + [C.c_void_p] * 4 + \
            [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double]
        L.ro_shade.argtypes = [C.c_char_p] + [C.c_void_p] * 5 + [C.c_int, C.c_int]
        L.ro_diagnostics.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.ro_scene_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_scene_root_bounds.restype = C.c_int
        L.ro_scene_root_bounds.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_scene_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ro_scene_spheres.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


NODE_TYPES = {"sphere": 0, "box": 1, "torus": 2, "mandelbulb": 3, "round": 10, "smoothUnion": 11, "smoothSub": 12,
              "twist": 13, "repetition": 14, "anim": 15}


def _flatten_nodes(prims):
    """Expression forest (nested dicts, the format rm_oracle.js takes as cfg.prims) -> the 16-double
    records of ro_scene_from_nodes + root indices."""
    rows, roots = [], []

    def add(d):
        t = d["type"]
        r = [0.0] * 16
        r[0] = NODE_TYPES[t]
        r[4] = float("nan")
        r[13] = r[14] = -1
        if t in ("sphere", "box", "torus", "mandelbulb"):
            r[1:4] = [float(v) for v in d["pos"]]
            if d.get("rot") is not None:
                r[4:7] = [float(v) for v in d["rot"]]
            if t == "sphere":
                r[7] = d["r"]
            elif t == "box":
                r[7:10] = [float(v) for v in d["half"]]
            elif t == "torus":
                r[7] = d["radius"]
            else:
                r[7:11] = [d["power"], d["iterations"], 1.0 if d["animate"] else 0.0, d["speed"]]
        else:
            r[13] = add(d["a"])
            if t in ("smoothUnion", "smoothSub"):
                r[14] = add(d["b"])
                r[7] = d["k"]
            elif t == "round":
                r[7] = d["radius"]
            elif t == "twist":
                r[7] = d["amount"]
            elif t == "repetition":
                r[7:10] = [float(v) for v in d["spacing"]]
            else:
                r[7:10] = [float(v) for v in d["direction"]]
                r[10], r[11] = d["amplitude"], d["speed"]
        rows.append(r)
        return len(rows) - 1

    for d in prims:
        roots.append(add(d))
    return np.array(rows, np.float64).reshape(-1, 16), np.array(roots, np.int32)


class OracleScene:
    """Mirrors `new Scene(accel); scene.loadPreset(i); scene.camera.setAngles(p, y)`
    (reference src/workers/raymarchWorker.ts:37-39)."""

    def __init__(self, preset=None, accel="None", spheres=None, prims=None):
        """prims: list of dicts {type: 'sphere'|'box'|'torus', pos: (x,y,z), rot: (rx,ry,rz)|None,
        r | half | radius} placed like SceneManager.createSphere/createBox/createTorus."""
        L = lib()
        self.accel = accel
        if prims is not None and any(d["type"] not in ("sphere", "box", "torus") for d in prims):
            desc, roots = _flatten_nodes(prims)
            self._h = L.ro_scene_from_nodes(_p(desc), len(desc), _p(roots), len(roots), accel.encode())
        elif prims is not None:
            desc = np.zeros((len(prims), 11), np.float64)
            for i, d in enumerate(prims):
                desc[i, 0] = {"sphere": 0, "box": 1, "torus": 2}[d["type"]]
                desc[i, 1:4] = d["pos"]
                desc[i, 4:7] = d["rot"] if d.get("rot") is not None else (np.nan, 0, 0)
                if d["type"] == "box":
                    desc[i, 7:10] = d["half"]
                else:
                    desc[i, 7] = d["r"] if d["type"] == "sphere" else d["radius"]
            self._h = L.ro_scene_from_prims(_p(desc), len(prims), accel.encode())
        elif spheres is not None:
            s = np.ascontiguousarray(spheres, dtype=np.float64).reshape(-1, 4)
            xyz = np.ascontiguousarray(s[:, :3])
            rad = np.ascontiguousarray(s[:, 3])
            self._h = L.ro_scene_from_spheres(_p(xyz), _p(rad), len(s), accel.encode())
        else:
            self._h = L.ro_scene_from_preset(int(preset), accel.encode())
        if not self._h:
            raise ValueError("preset %r needs non-sphere primitives (out of scope)" % (preset,))

    def close(self):
        if self._h:
            lib().ro_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_angles(self, pitch, yaw):
        lib().ro_scene_set_angles(self._h, float(pitch), float(yaw))

    def camera(self):
        rot = np.zeros(9, np.float32)
        org = np.zeros(3, np.float32)
        lib().ro_scene_camera(self._h, _p(rot), _p(org))
        return rot, org

    def stats(self):
        out = np.zeros(8, np.int32)
        lib().ro_scene_stats(self._h, _p(out))
        keys = ["bvh_nodes", "bvh_leaves", "bvh_depth", "oct_nodes", "oct_leaves", "oct_empty",
                "oct_maxleafprims", "n"]
        return dict(zip(keys, (int(v) for v in out)))

    def root_bounds(self):
        out = np.zeros(6, np.float32)
        ok = lib().ro_scene_root_bounds(self._h, _p(out))
        return out if ok else None

    def spheres(self):
        n = self.stats()["n"]
        c = np.zeros((n, 3), np.float32)
        r = np.zeros(n, np.float64)
        lib().ro_scene_spheres(self._h, _p(c), _p(r))
        return c, r

    def prims(self):
        """(type, world_to_local float32[16], params float64[3]) per primitive: what the product's
        rm_scene_from_prims takes."""
        n = self.stats()["n"]
        t = np.zeros(n, np.int32)
        m = np.zeros((n, 16), np.float32)
        par = np.zeros((n, 3), np.float64)
        lib().ro_scene_prims(self._h, _p(t), _p(m), _p(par))
        return [(int(t[i]), m[i].copy(), par[i].copy()) for i in range(n)]

    def nodes(self):
        """Flattened expression forest as the product's rm_scene_from_nodes takes it:
        (list of (type, child_a, child_b, world_to_local float32[16], params float64[6]), roots)."""
        n = lib().ro_scene_nodes(self._h, None, None, None, None, None)
        t = np.zeros(n, np.int32)
        kids = np.zeros((n, 2), np.int32)
        m = np.zeros((n, 16), np.float32)
        par = np.zeros((n, 6), np.float64)
        roots = np.zeros(self.stats()["n"], np.int32)
        lib().ro_scene_nodes(self._h, _p(t), _p(kids), _p(m), _p(par), _p(roots))
        return [(int(t[i]), int(kids[i, 0]), int(kids[i, 1]), m[i].copy(), par[i].copy()) for i in range(n)], roots.tolist()

    def distance(self, p, time=0.0):
        lib().ro_set_time(float(time))
        pos = np.asarray(p, np.float32)
        cnt = C.c_uint32(0)
        d = lib().ro_scene_distance(self._h, _p(pos), C.byref(cnt))
        return d, cnt.value

    def render(self, width, height, y_start=0, y_end=None, algorithm="sphere-tracer", time=0.0,
               overshoot_factor=None, step_size=None):
        """runRaymarcher (reference src/cpu_algorithms/raymarcher.ts:46-109): tile-local buffers."""
        if y_end is None:
            y_end = height
        rows = max(0, y_end - y_start)
        depth = np.zeros(width * rows, np.uint8)
        normal = np.zeros(width * rows * 3, np.uint8)
        sdf = np.zeros(width * rows, np.uint16)
        iters = np.zeros(width * rows, np.uint16)
        nan = float("nan")
        rc = lib().ro_run_raymarcher_ex(self._h, algorithm.encode(), _p(depth), _p(normal), _p(sdf),
                                        _p(iters), width, height, float(time), y_start, y_end,
                                        nan if overshoot_factor is None else float(overshoot_factor),
                                        nan if step_size is None else float(step_size))
        if rc != 0:
            raise RuntimeError("ro_run_raymarcher_ex failed (%d)" % rc)
        return depth, normal, sdf, iters


def shade(model, depth, normal, sdf, iters, width, height):
    """ShadingModel.shade (reference src/util/shading_models/*.ts)."""
    out = np.zeros(width * height * 4, np.uint8)
    lib().ro_shade(model.encode(), _p(out), _p(depth), _p(normal), _p(sdf), _p(iters), width, height)
    return out


def diagnostics(sdf, iters):
    """main.ts:528-548 -> dict(total_sdf, max_sdf, min_sdf, total_iters)."""
    out = np.zeros(4, np.float64)
    lib().ro_diagnostics(_p(sdf), _p(iters), sdf.size, _p(out))
    return {"total_sdf": int(out[0]), "max_sdf": int(out[1]), "min_sdf": int(out[2]),
            "total_iters": int(out[3])}


def synthetic_mixed_prims(n=40, seed=7):
    """Deterministic mix of spheres, boxes and tori, half of them with a rotation argument."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        kind = ("sphere", "box", "torus")[i % 3]
        d = {"type": kind, "pos": [float(np.float32(v)) for v in rng.uniform(-1.3, 1.3, 3)],
             "rot": [float(np.float32(v)) for v in rng.uniform(-3.2, 3.2, 3)] if i % 2 else None}
        if kind == "sphere":
            d["r"] = float(rng.uniform(0.08, 0.3))
        elif kind == "box":
            d["half"] = [float(np.float32(v)) for v in rng.uniform(0.05, 0.3, 3)]
        else:
            d["radius"] = float(rng.uniform(0.1, 0.3))
        out.append(d)
    return out


def synthetic_spheres(n=10000, seed=0x5EED5EED):
    """splitmix64, centres uniform in [-1.5,1.5]^3 then fround, radii uniform
    in [0.01,0.04] kept as doubles.  Returns float64 [n,4] (x,y,z,r)."""
    mask = (1 << 64) - 1
    state = seed & mask

    def nxt():
        nonlocal state
        state = (state + 0x9E3779B97F4A7C15) & mask
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z = z ^ (z >> 31)
        return (z >> 11) * (1.0 / 9007199254740992.0)

    out = np.zeros((n, 4), np.float64)
    for i in range(n):
        for k in range(3):
            out[i, k] = np.float32(-1.5 + 3.0 * nxt())
        out[i, 3] = 0.01 + 0.03 * nxt()
    return out
