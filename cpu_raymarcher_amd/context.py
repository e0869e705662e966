"""One rm_ctx per GPU (include/rm_raymarch.h "Threading"): owns the device-resident scene."""
import ctypes as C

import numpy as np

from . import _native as N


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x):
    """void* of a numpy array (host) or a torch tensor (device); None -> NULL."""
    if x is None:
        return None
    if _is_torch(x):
        return C.c_void_p(x.data_ptr())
    return x.ctypes.data_as(C.c_void_p)


_BPP = {"depth": 1, "normal": 3, "sdf": 2, "iters": 2, "rgba": 4}


def _check_buffers(npx, bufs, need_device=None):
    """The C ABI takes raw pointers: everything it assumes about a buffer is checked here.  bufs: name -> buffer
    or None.  Every buffer must be C-contiguous, of the reference's element size (depth / normal / rgba 1 byte, sdf / iters
    2 bytes: Uint8ClampedArray / Uint16Array) and hold at least npx pixels; all of them on the host or all on one
    device.  Returns True when they are device (torch CUDA) buffers."""
    on_dev = None
    for name, b in bufs.items():
        if b is None:
            continue
        esz = 2 if name in ("sdf", "iters") else 1
        if _is_torch(b):
            dev, contiguous, itemsize, nbytes = b.is_cuda, b.is_contiguous(), b.element_size(), b.numel() * b.element_size()
        elif isinstance(b, np.ndarray):
            dev, contiguous, itemsize, nbytes = False, b.flags["C_CONTIGUOUS"], b.itemsize, b.nbytes
            if not b.flags["WRITEABLE"] and name != "_in":
                raise ValueError("%s buffer is read-only" % name)
        else:
            raise ValueError("%s must be a numpy array or a torch tensor, not %s" % (name, type(b).__name__))
        if not contiguous:
            raise ValueError("%s buffer must be C-contiguous" % name)
        if itemsize != esz and not (dev and itemsize == 1):  # device side: byte views into a packed gather buffer are fine
            raise ValueError("%s buffer must have %d-byte elements (got %d)" % (name, esz, itemsize))
        if nbytes < npx * _BPP[name]:
            raise ValueError("%s buffer holds %d bytes, the tile needs %d" % (name, nbytes, npx * _BPP[name]))
        if on_dev is None:
            on_dev = dev
        elif on_dev != dev:
            raise ValueError("buffers must be all on the host or all on the device")
    if need_device is not None and on_dev is not None and on_dev != need_device:
        raise ValueError("this entry point takes %s buffers" % ("device" if need_device else "host"))
    return bool(on_dev)


def _current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Context:
    """`device` >= 0 binds a GPU; device=None creates a host-only context (scene building
    and camera only -- every render entry then fails with RM_E_NO_DEVICE: no CPU path)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        self.device = -1 if device is None else int(device)
        rc = N.lib().rm_create(self.device, C.byref(self._h))
        if rc != N.RM_OK:
            raise N.RmError(rc, "rm_create(device=%d) failed: no usable HIP device" % self.device)

    def close(self):
        if self._h:
            N.lib().rm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- options (never change results) -------------------------------------------
    def set_option(self, key, value):
        N.check(self._h, N.lib().rm_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int64(0)
        N.check(self._h, N.lib().rm_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    # ---- scene ----------------------------------------------------------------------
    def scene_from_preset(self, index, accel):
        N.check(self._h, N.lib().rm_scene_from_preset(self._h, int(index), int(accel)))

    def scene_from_spheres(self, centers, radii, accel):
        c = np.ascontiguousarray(centers, dtype=np.float32).reshape(-1, 3)
        r = np.ascontiguousarray(radii, dtype=np.float64).reshape(-1)
        if len(c) != len(r):
            raise ValueError("centers and radii differ in length")
        N.check(self._h, N.lib().rm_scene_from_spheres(self._h, _ptr(c), _ptr(r), len(r), int(accel)))

    def scene_from_prims(self, prims, accel):
        """prims: iterable of (type, world_to_local[16], params[<=3]); type 0 sphere, 1 box, 2 torus."""
        prims = list(prims)
        arr = (N.rm_prim * max(1, len(prims)))()
        for i, (t, m, par) in enumerate(prims):
            arr[i].type = int(t)
            for k in range(16):
                arr[i].world_to_local[k] = float(m[k])
            for k in range(3):
                arr[i].params[k] = float(par[k]) if k < len(par) else 0.0
        N.check(self._h, N.lib().rm_scene_from_prims(self._h, arr, len(prims), int(accel)))

    def scene_from_nodes(self, nodes, roots, accel):
        """SDF expression forest (rm_scene_from_nodes): nodes = iterable of
        (type, child_a, child_b, world_to_local[16] | None, params[<=6]); roots = Scene.objects."""
        nodes = list(nodes)
        arr = (N.rm_node * max(1, len(nodes)))()
        for i, (t, a, b, m, par) in enumerate(nodes):
            arr[i].type, arr[i].child_a, arr[i].child_b = int(t), int(a), int(b)
            for k in range(16):
                arr[i].world_to_local[k] = 0.0 if m is None else float(m[k])
            for k in range(6):
                arr[i].params[k] = float(par[k]) if k < len(par) else 0.0
        r = np.ascontiguousarray(roots, dtype=np.int32)
        N.check(self._h, N.lib().rm_scene_from_nodes(self._h, arr, len(nodes), _ptr(r), len(r), int(accel)))

    def scene_set_time(self, time):
        """Scene.updateTime(time) for scene_distance (renders take the job's time)."""
        N.check(self._h, N.lib().rm_scene_set_time(self._h, float(time)))

    def selftest_jsmath(self, fn, a, b=None):
        """Math.sin/cos/atan2/asin/log/pow/round/atan (fn 0..7) as the device evaluates them."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
        out = np.zeros_like(a)
        N.check(self._h, N.lib().rm_selftest_jsmath(self._h, int(fn), _ptr(a), _ptr(b), a.size, _ptr(out)))
        return out

    def scene_info(self):
        info = N.rm_scene_info()
        N.check(self._h, N.lib().rm_scene_get_info(self._h, C.byref(info)))
        d = {k: getattr(info, k) for k, _ in info._fields_ if k not in ("root_min", "root_max", "reserved")}
        d["root_min"] = [float(v) for v in info.root_min]
        d["root_max"] = [float(v) for v in info.root_max]
        return d

    def scene_distance(self, points):
        """Scene.getDistance for a batch of points -> (dist float64[n], count uint32[n])."""
        p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        dist = np.zeros(len(p), np.float64)
        cnt = np.zeros(len(p), np.uint32)
        N.check(self._h, N.lib().rm_scene_distance(self._h, _ptr(p), len(p), _ptr(dist), _ptr(cnt)))
        return dist, cnt

    # ---- render ---------------------------------------------------------------------
    def _attach_diag(self, diag):
        """rm_render_attach_diagnostics for the render call that follows: `diag` is a CUDA tensor of 32 bytes that
        receives (total_sdf, total_iters, max_sdf | min_sdf << 32, pad) of the pixels that call renders (decode_acc)."""
        if diag is None:
            return
        if not (_is_torch(diag) and diag.is_cuda and diag.is_contiguous() and diag.numel() * diag.element_size() >= 32):
            raise ValueError("diag must be a contiguous CUDA tensor of 32 bytes")
        self._same_device(dict(diag=diag))
        N.check(self._h, N.lib().rm_render_attach_diagnostics(self._h, _ptr(diag)))

    def render_tile(self, job, depth, normal, sdf, iters, rgba=None, shader=0, diag=None):
        """Host numpy buffers -> rm_render_tile (+ rm_shade when rgba is given); torch CUDA
        tensors -> rm_render_tile_device on torch's current stream (fused shade; `diag`: fused diagnostics)."""
        npx = max(0, job.y_end - job.y_start) * max(0, job.width)
        bufs = dict(depth=depth, normal=normal, sdf=sdf, iters=iters, rgba=rgba)
        if _check_buffers(npx, bufs) or (diag is not None and all(b is None for b in bufs.values())):
            self._same_device(bufs)
            self._attach_diag(diag)
            N.check(self._h, N.lib().rm_render_tile_device(
                self._h, C.byref(job), int(shader), _ptr(depth), _ptr(normal), _ptr(sdf), _ptr(iters),
                _ptr(rgba), _current_stream_ptr()))
            return
        if diag is not None:
            raise ValueError("fused diagnostics need device buffers")
        if depth is None or normal is None or sdf is None or iters is None:
            raise ValueError("the host entry point needs all four G-buffers")
        N.check(self._h, N.lib().rm_render_tile(self._h, C.byref(job), _ptr(depth), _ptr(normal),
                                                 _ptr(sdf), _ptr(iters)))
        if rgba is not None:
            rows = max(0, job.y_end - job.y_start)
            self.shade(shader, job.width, rows, depth, normal, sdf, iters, rgba)

    def _same_device(self, bufs):
        for name, b in bufs.items():
            if b is not None and _is_torch(b) and b.is_cuda and b.device.index != self.device:
                raise ValueError("%s buffer is on cuda:%s, the context on cuda:%d" % (name, b.device.index, self.device))

    def render_stripes(self, job, stripe_rows, n_parts, part, depth, normal, sdf, iters, rgba=None, shader=0, diag=None):
        """One launch for every stripe of `part` (rm_render_stripes_device); device buffers only."""
        rows = N.lib().rm_stripe_rows(job.y_start, job.y_end, int(stripe_rows), int(n_parts), int(part))
        if rows < 0:
            raise ValueError("bad stripe partition")
        bufs = dict(depth=depth, normal=normal, sdf=sdf, iters=iters, rgba=rgba)
        _check_buffers(rows * max(0, job.width), bufs, need_device=True)
        self._same_device(bufs)
        self._attach_diag(diag)
        N.check(self._h, N.lib().rm_render_stripes_device(
            self._h, C.byref(job), int(shader), int(stripe_rows), int(n_parts), int(part), _ptr(depth), _ptr(normal),
            _ptr(sdf), _ptr(iters), _ptr(rgba), _current_stream_ptr()))

    def render_stripe_list(self, job, stripe_rows, stripe_ids, depth, normal, sdf, iters, rgba=None, shader=0, diag=None):
        """One launch for the listed stripes (strictly increasing ids; rm_render_stripe_list_device), packed in list
        order; device buffers only.  Serves any deal of stripes to ranks (deal_stripes)."""
        ids = np.ascontiguousarray(stripe_ids, dtype=np.int32)
        height = max(0, job.y_end - job.y_start)
        rows = sum(min(height, (int(i) + 1) * stripe_rows) - int(i) * stripe_rows for i in ids)
        bufs = dict(depth=depth, normal=normal, sdf=sdf, iters=iters, rgba=rgba)
        _check_buffers(max(0, rows) * max(0, job.width), bufs, need_device=True)
        self._same_device(bufs)
        self._attach_diag(diag)
        N.check(self._h, N.lib().rm_render_stripe_list_device(
            self._h, C.byref(job), int(shader), int(stripe_rows), _ptr(ids), len(ids), _ptr(depth), _ptr(normal),
            _ptr(sdf), _ptr(iters), _ptr(rgba), _current_stream_ptr()))

    def assemble_frame(self, gathered, rank_stride, section_offset, row_bytes, height, stripe_rows, owner, world, frame,
                       acc_offset=-1, acc=None):
        """Rank 0's fan-in in one kernel (rm_assemble_frame_device): stripes of the gathered per-rank buffers -> row-major
        frame, and the ranks' 32-byte partial diagnostics (at acc_offset of each rank's buffer) -> acc."""
        own = np.ascontiguousarray(owner, dtype=np.int32)
        for name, t, need in (("gathered", gathered, world * rank_stride), ("frame", frame, row_bytes * height)):
            if not (_is_torch(t) and t.is_cuda and t.is_contiguous()) or t.numel() * t.element_size() < need:
                raise ValueError("%s must be a contiguous CUDA tensor of at least %d bytes" % (name, need))
        if acc is not None and (not (_is_torch(acc) and acc.is_cuda) or acc.numel() * acc.element_size() < 32):
            raise ValueError("acc must be a CUDA tensor of 32 bytes")
        N.check(self._h, N.lib().rm_assemble_frame_device(
            self._h, _ptr(gathered), int(rank_stride), int(section_offset), int(row_bytes), int(height), int(stripe_rows),
            _ptr(own), len(own), int(world), _ptr(frame), int(acc_offset), _ptr(acc), _current_stream_ptr()))

    def last_kernel(self):
        """The render-kernel instantiation the last render entry launched (rm_last_kernel)."""
        return N.lib().rm_last_kernel(self._h).decode()

    # ---- run-time specialised kernels of expression forests (rm_rtc_*) -------------------
    def rtc_source(self):
        """The straight-line HIP source generated for the active scene's expression forest ("" if none)."""
        need = C.c_int64(0)
        N.check(self._h, N.lib().rm_rtc_source(self._h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(max(1, need.value))
        N.check(self._h, N.lib().rm_rtc_source(self._h, buf, len(buf), None))
        return buf.value.decode()

    def rtc_compile_check(self, accel=0, other=False):
        """Compiles the active scene's specialised kernel for gfx950 without loading it (no GPU needed).
        Returns (compiler log with the resource-usage remarks, seconds)."""
        buf = C.create_string_buffer(1 << 16)
        secs = C.c_double(0)
        rc = N.lib().rm_rtc_compile_check(self._h, int(accel), int(bool(other)), buf, len(buf), C.byref(secs))
        if rc != 0:
            raise RuntimeError("rm_rtc_compile_check: %s\n%s" % (N.lib().rm_last_error(self._h).decode(), buf.value.decode()))
        return buf.value.decode(), secs.value

    def rtc_status(self):
        """(kernels compiled for the active scene, compiles that failed, most recent compile log)."""
        done, bad = C.c_int32(0), C.c_int32(0)
        buf = C.create_string_buffer(1 << 16)
        N.check(self._h, N.lib().rm_rtc_status(self._h, C.byref(done), C.byref(bad), buf, len(buf)))
        return done.value, bad.value, buf.value.decode()

    def shade(self, shader, width, height, depth, normal, sdf, iters, rgba):
        npx = max(0, int(width)) * max(0, int(height))
        bufs = dict(depth=depth, normal=normal, sdf=sdf, iters=iters, rgba=rgba)
        if any(b is None for b in bufs.values()):
            raise ValueError("shade needs all five buffers")
        if _check_buffers(npx, bufs):
            self._same_device(bufs)
            N.check(self._h, N.lib().rm_shade_device(self._h, int(shader), width, height, _ptr(depth),
                                                      _ptr(normal), _ptr(sdf), _ptr(iters), _ptr(rgba),
                                                      _current_stream_ptr()))
        else:
            N.check(self._h, N.lib().rm_shade(self._h, int(shader), width, height, _ptr(depth), _ptr(normal),
                                               _ptr(sdf), _ptr(iters), _ptr(rgba)))

    def _check_counters(self, sdf, iters):
        if sdf is None or iters is None:
            raise ValueError("null counter buffer")
        n = int(sdf.numel()) if _is_torch(sdf) else int(sdf.size)
        m = int(iters.numel()) if _is_torch(iters) else int(iters.size)
        if n != m:
            raise ValueError("sdfEval and iters differ in length (%d, %d)" % (n, m))
        dev = _check_buffers(n, dict(sdf=sdf, iters=iters))
        if dev:
            self._same_device(dict(sdf=sdf, iters=iters))
        return n, dev

    def reduce_counters(self, sdf, iters):
        out = N.rm_diagnostics()
        n, dev = self._check_counters(sdf, iters)
        if dev:
            N.check(self._h, N.lib().rm_reduce_counters_device(self._h, _ptr(sdf), _ptr(iters), n,
                                                                C.byref(out), _current_stream_ptr()))
        else:
            N.check(self._h, N.lib().rm_reduce_counters(self._h, _ptr(sdf), _ptr(iters), n, C.byref(out)))
        return {"total_sdf": int(out.total_sdf_calls), "total_iters": int(out.total_iterations),
                "max_sdf": int(out.max_sdf_calls), "min_sdf": int(out.min_sdf_calls),
                "total_pixels": int(out.total_pixels)}

    def reduce_counters_enqueue(self, sdf, iters, acc):
        """Async diagnostics for a frame loop: `acc` is a CUDA int64[4] tensor that receives
        (total_sdf, total_iters, max_sdf | min_sdf << 32, pad); read it with decode_acc()."""
        n, dev = self._check_counters(sdf, iters)
        if not dev and n:
            raise ValueError("reduce_counters_enqueue takes device buffers")
        if not (_is_torch(acc) and acc.is_cuda and acc.numel() * acc.element_size() >= 32):
            raise ValueError("acc must be a CUDA tensor of 32 bytes")
        N.check(self._h, N.lib().rm_reduce_counters_enqueue(self._h, _ptr(sdf), _ptr(iters), n,
                                                            _ptr(acc), _current_stream_ptr()))

    @staticmethod
    def decode_acc(acc):
        v = [int(x) for x in acc.cpu().tolist()]
        return {"total_sdf": v[0], "total_iters": v[1], "max_sdf": v[2] & 0xFFFFFFFF,
                "min_sdf": (v[2] >> 32) & 0xFFFFFFFF}

    def selftest_hypot(self, xyz):
        a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(len(a), np.float64)
        N.check(self._h, N.lib().rm_selftest_hypot(self._h, _ptr(a), len(a), _ptr(out)))
        return out


def _selftest_fastdiv(self, seed, n):
    """Device-side comparison of the two Math.hypot forms; returns the mismatch count."""
    out = C.c_uint64(0)
    N.check(self._h, N.lib().rm_selftest_fastdiv(self._h, int(seed), int(n), C.byref(out)))
    return out.value


Context.selftest_fastdiv = _selftest_fastdiv


def _selftest_recip(self, mode):
    """Exhaustive device-side check of the range-restricted division of ray set-up (mode 0: 1.0 / d for every finite
    non-zero binary32 d; mode 1: x / W for 0 <= x < 65536, 1 <= W < 65536); returns the mismatch count."""
    out = C.c_uint64(0)
    N.check(self._h, N.lib().rm_selftest_recip(self._h, int(mode), C.byref(out)))
    return out.value


Context.selftest_recip = _selftest_recip


def make_transform(x, y, z, rotation=None):
    """SceneManager.getTransform (sceneManager.ts:21-37) -> world->local float32[16]."""
    out = np.zeros(16, np.float32)
    rot = None if rotation is None else np.ascontiguousarray(rotation, dtype=np.float32)
    rc = N.lib().rm_make_transform(float(x), float(y), float(z), _ptr(rot), _ptr(out))
    if rc != N.RM_OK:
        raise N.RmError(rc, "rm_make_transform")
    return out


def scale_transform(m, x, y, z):
    """gl-matrix mat4.scale(m, m, [x, y, z]) (sceneManager.ts:63) -> new float32[16]."""
    out = np.ascontiguousarray(m, dtype=np.float32).copy()
    rc = N.lib().rm_scale_transform(_ptr(out), float(x), float(y), float(z))
    if rc != N.RM_OK:
        raise N.RmError(rc, "rm_scale_transform")
    return out


def camera_from_angles(pitch, yaw):
    rot = np.zeros(9, np.float32)
    org = np.zeros(3, np.float32)
    rc = N.lib().rm_camera_from_angles(float(pitch), float(yaw), _ptr(rot), _ptr(org))
    if rc != N.RM_OK:
        raise N.RmError(rc, "rm_camera_from_angles")
    return rot, org


def deal_stripes(rows, stripe_rows, n_parts, weights=None):
    """rm_deal_stripes: owner of every stripe of `rows` rows (smooth weighted round-robin; equal weights = round-robin)."""
    n = (max(0, int(rows)) + int(stripe_rows) - 1) // int(stripe_rows)
    owner = np.zeros(max(1, n), np.int32)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.int32)
    if w is not None and len(w) != n_parts:
        raise ValueError("one weight per part")
    rc = N.lib().rm_deal_stripes(int(rows), int(stripe_rows), int(n_parts), _ptr(w), _ptr(owner))
    if rc < 0:
        raise N.RmError(rc, "rm_deal_stripes")
    return owner[:rc]


def partition_rows(height, n_workers, i):
    """main.ts:444-449."""
    a, b = C.c_int32(0), C.c_int32(0)
    rc = N.lib().rm_partition_rows(int(height), int(n_workers), int(i), C.byref(a), C.byref(b))
    if rc != N.RM_OK:
        raise N.RmError(rc, "rm_partition_rows")
    return a.value, b.value
