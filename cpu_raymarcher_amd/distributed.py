"""Row-tile sharding of one frame over the GPUs of a node, one process per GPU, and the
gather of the per-rank tile buffers to rank 0 (RCCL over xGMI through torch.distributed).

Replaces the reference's worker fan-out / fan-in (src/main.ts:444-468): there the rows are
split in contiguous ceil(H/N) blocks over <= 4 Web Workers and the tiles come back by
postMessage; here every rank renders its rows of the frame on its own GPU and one
`gather` moves {RGBA, sdfEval, iters} (8 B/pixel; + depth, normal = 12 B/pixel on request)
to rank 0.  Pixels are independent functions of (x, y, W, H, camera, scene)
(raymarcher.ts:72-76,83), so any row subset is exact.

Two partitions:
  * "contiguous": the reference's rule, rows [min(i r, H), min((i+1) r, H)), r = ceil(H/N).
    Load-imbalanced: top and bottom blocks are mostly root-box misses.
  * "interleaved" (default): stripes of `stripe` rows dealt round-robin, rank = (y // stripe) % N,
    or -- `weights` -- by a smooth WEIGHTED round-robin (rm_deal_stripes): rank 0 also reassembles the
    frame and runs the receiving side of the gather, so an equal deal makes it the slowest rank
    (VERDICT r1); it gets a smaller share, every rank's stripes still spread over the whole frame.
A direct gather lets the root receive from its 7 peers over 7 different xGMI links at once;
a ring would be bound by one link.
"""
import numpy as np

from .context import deal_stripes, partition_rows

SECTION_BYTES = {"rgba": 4, "sdf": 2, "iters": 2, "depth": 1, "normal": 3}


def owned_rows(height, world, rank, mode="interleaved", stripe=16, weights=None):
    """List of (y_start, y_end) row ranges of `rank`, in increasing y."""
    if world <= 1:
        return [(0, height)] if height > 0 else []
    if mode == "contiguous":
        a, b = partition_rows(height, world, rank)
        return [(a, b)] if b > a else []
    if weights is not None:
        owner = deal_stripes(height, stripe, world, weights)
        return [(int(k) * stripe, min(height, (int(k) + 1) * stripe)) for k in np.nonzero(owner == rank)[0]]
    out = []
    k = rank
    while k * stripe < height:
        out.append((k * stripe, min(height, (k + 1) * stripe)))
        k += world
    return out


def max_local_rows(height, world, mode="interleaved", stripe=16, weights=None):
    return max((sum(b - a for a, b in owned_rows(height, world, r, mode, stripe, weights)) for r in range(world)), default=0)


def balanced_weights(world, root_overhead, shard_time, scale=1000):
    """Integer weights for rm_deal_stripes that equalise the ranks' per-frame GPU time when rank 0 carries
    `root_overhead` seconds of extra work per frame (reassembly, combined diagnostics, the receiving side of the
    gather) and an equal 1/world share renders in `shard_time` seconds: with per-frame time a_r + b s_r (b = world x
    shard_time) equal over the ranks and the shares summing to one, s_0 = 1/N - a_0 (N - 1) / (N b), the others
    share the rest equally.  Rank 0 keeps at least a quarter of an equal share."""
    if world <= 1:
        return [scale]
    b = max(1e-12, world * float(shard_time))
    s0 = 1.0 / world - float(root_overhead) * (world - 1) / (world * b)
    s0 = min(max(s0, 0.25 / world), 1.0 / world)
    rest = (1.0 - s0) / (world - 1)
    return [max(1, int(round(s0 / rest * scale)))] + [scale] * (world - 1)


class FrameLayout:
    """Packed per-rank byte buffer: sections [rgba | sdf | iters | depth | normal], each sized for
    `cap` rows (the largest share of any rank, so every rank sends the same byte count)."""

    def __init__(self, width, height, world, sections=("rgba", "sdf", "iters"), mode="interleaved", stripe=16, tail=0,
                 weights=None):
        """tail: extra bytes after the sections (rounded up to 256) that travel with the gather, e.g. the 32-byte
        diagnostics accumulator of this rank's rows (tail_offset).  weights: one positive integer per rank for the
        weighted stripe deal (None: equal, plain round-robin)."""
        self.width, self.height, self.world = width, height, world
        self.sections, self.mode, self.stripe = tuple(sections), mode, stripe
        self.weights = None if weights is None or mode != "interleaved" or world <= 1 else [int(w) for w in weights]
        # owner[s]: rank of stripe s (interleaved partitions; the native assembler and the list render take it)
        self.owner = deal_stripes(height, stripe, world, self.weights) if mode == "interleaved" and world >= 1 else None
        self.cap = max_local_rows(height, world, mode, stripe, self.weights)
        self.offsets = {}
        off = 0
        for s in self.sections:
            self.offsets[s] = off
            off += (SECTION_BYTES[s] * width * self.cap + 255) // 256 * 256
        self.tail_offset = off
        self.nbytes = max(off + (int(tail) + 255) // 256 * 256, 256)

    def rows(self, rank):
        return owned_rows(self.height, self.world, rank, self.mode, self.stripe, self.weights)

    def stripe_ids(self, rank):
        """Stripe indices of `rank`, increasing (interleaved partitions)."""
        return np.nonzero(self.owner == rank)[0].astype(np.int32)

    def section(self, buf, name, rows=None):
        """View of one section of a packed uint8 buffer (torch tensor or numpy array)."""
        n = SECTION_BYTES[name] * self.width * (self.cap if rows is None else rows)
        return buf[self.offsets[name]:self.offsets[name] + n]

    def scatter_into_frame(self, packed_per_rank, frame):
        """Rank 0 after the gather: copy every rank's packed rows to their place in the
        row-major full frame.  frame: dict name -> flat uint8 buffer of H*W*bytes."""
        W = self.width
        for r, buf in enumerate(packed_per_rank):
            local = 0
            for (a, b) in self.rows(r):
                for s in self.sections:
                    bpp = SECTION_BYTES[s]
                    src = self.section(buf, s)
                    frame[s][a * W * bpp:b * W * bpp] = src[local * W * bpp:(local + b - a) * W * bpp]
                local += b - a


class ShardedFrameRenderer:
    """One frame = render the rows this rank owns + gather to rank 0.

    render_rows(y_start, y_end, local_row, packed) must write the tile [y_start, y_end) into the
    sections of `packed` starting at tile-local row `local_row` (on the GPU: one
    rm_render_tile_device launch per range; the CPU gloo test injects the oracle)."""

    def __init__(self, layout, rank, world, render_rows, new_buffer, dist=None, double_buffer=True, render_all=None,
                 frames_in_flight=None, streams=None):
        """frames_in_flight: buffer sets (default 2 = double buffering).  streams: one device stream per buffer
        set; frame n is rendered, gathered and assembled on stream n % frames_in_flight, so the tail of one
        frame's persistent kernel overlaps the start of the next (None: everything on the current stream)."""
        self.layout, self.rank, self.world = layout, rank, world
        self.render_rows = render_rows
        self.render_all = render_all  # optional: one call that fills every owned row of `packed`
        self.dist = dist
        self.nbuf = int(frames_in_flight) if frames_in_flight else (2 if double_buffer else 1)
        self.streams = streams
        self.send = [new_buffer(layout.nbytes) for _ in range(self.nbuf)]
        self.recv = [[new_buffer(layout.nbytes) for _ in range(world)] if (rank == 0 and world > 1) else None
                     for _ in range(self.nbuf)]
        self.work = [None] * self.nbuf
        self.step = 0

    def render_local(self, slot):
        if self.render_all is not None:
            self.render_all(self.send[slot])
            return
        local = 0
        for (a, b) in self.layout.rows(self.rank):
            self.render_rows(a, b, local, self.send[slot])
            local += b - a

    def on_stream(self, slot):
        """Context manager: the device stream of buffer set `slot` (a no-op without streams)."""
        if self.streams is None:
            import contextlib
            return contextlib.nullcontext()
        import torch
        return torch.cuda.stream(self.streams[slot])

    def submit(self):
        """Render this rank's rows for the next frame and start its gather; returns the slot."""
        slot = self.step % self.nbuf
        self.step += 1
        with self.on_stream(slot):
            self._submit(slot)
        return slot

    def _submit(self, slot):
        if self.work[slot] is not None:
            self.work[slot].wait()  # the buffer's previous gather has been consumed
            self.work[slot] = None
        self.render_local(slot)
        if self.world > 1:
            self.work[slot] = self.dist.gather(self.send[slot], self.recv[slot] if self.rank == 0 else None, dst=0,
                                               async_op=True)

    def finish(self, slot, frame=None):
        """Wait for the gather of `slot`; on rank 0 optionally assemble the full frame."""
        with self.on_stream(slot):
            self._finish(slot, frame)

    def _finish(self, slot, frame=None):
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None
        if frame is not None and self.rank == 0:
            parts = self.recv[slot] if self.world > 1 else [self.send[slot]]
            self.layout.scatter_into_frame(parts, frame)

    def drain(self):
        for s in range(self.nbuf):
            if self.work[s] is not None:
                self.work[s].wait()
                self.work[s] = None


def new_frame(layout, zeros):
    """Full-frame buffers for rank 0: dict section -> flat uint8 of H*W*bytes."""
    return {s: zeros(SECTION_BYTES[s] * layout.width * layout.height) for s in layout.sections}


def gpu_render_rows(ctx, scene, width, height, shader, layout):
    """render_rows callback for the GPU: rm_render_tile_device straight into the packed buffer."""
    from . import _native as N
    from .host import _job
    sh = N.lib().rm_shader_from_string(str(shader).encode())
    want = set(layout.sections)

    def render_rows(y0, y1, local, packed):
        W = width
        def sec(name):
            if name not in want:
                return None
            bpp = SECTION_BYTES[name]
            return layout.section(packed, name)[local * W * bpp:]
        job = _job(scene, width, height, 0.0, y0, y1, "sphere-tracer")
        ctx.render_tile(job, sec("depth"), sec("normal"), sec("sdf"), sec("iters"), rgba=sec("rgba"), shader=sh)
    return render_rows


def gpu_render_all(ctx, scene, width, height, shader, layout, rank, extra=None, diag_in_tail=False):
    """render_all callback for the GPU and the interleaved partition: ONE rm_render_stripes_device
    launch writes every stripe of this rank into the packed buffer.  extra(packed) -> dict name -> device buffer for
    outputs that are not sections of the layout (rank-local sdfEval / iters when only RGBA is gathered).
    diag_in_tail: the same launch leaves the diagnostics of this rank's rows (main.ts:528-548; sums, max and min
    combine exactly over ranks) in the 32 bytes at layout.tail_offset, which travel with the gather
    (rm_render_attach_diagnostics: no reduction launch)."""
    from . import _native as N
    from .host import _job
    if layout.mode != "interleaved":
        return None
    sh = N.lib().rm_shader_from_string(str(shader).encode())
    want = set(layout.sections)

    def render_all(packed):
        local = extra(packed) if extra else {}

        def sec(name):
            return layout.section(packed, name) if name in want else local.get(name)
        job = _job(scene, width, height, 0.0, 0, height, "sphere-tracer")
        diag = packed[layout.tail_offset:layout.tail_offset + 32] if diag_in_tail else None
        if layout.weights is not None:  # any deal of stripes: the launch takes the list
            ctx.render_stripe_list(job, layout.stripe, ids, sec("depth"), sec("normal"), sec("sdf"), sec("iters"),
                                   rgba=sec("rgba"), shader=sh, diag=diag)
        else:
            ctx.render_stripes(job, layout.stripe, layout.world, rank, sec("depth"), sec("normal"), sec("sdf"),
                               sec("iters"), rgba=sec("rgba"), shader=sh, diag=diag)
    ids = layout.stripe_ids(rank)
    return render_all


class GpuFrameAssembler:
    """Rank 0's reassembly on the device: the gather lands in one [world, nbytes] tensor per slot; ONE native
    kernel per section (rm_assemble_frame_device, csrc/rm_frame_ops.hip) copies every stripe to its place in the
    row-major frame and, with `acc`, combines the ranks' partial diagnostics that ride in the tail of their packed
    buffers.  (Round 1 used torch index kernels plus eight small tensor ops per frame for the diagnostics: rank 0's
    host thread and GPU both paid for them)  `ctx` is the rank's cpu_raymarcher_amd.Context."""

    def __init__(self, layout, device, nbuf=2, ctx=None):
        import torch
        if ctx is None:
            raise ValueError("GpuFrameAssembler needs the rank's Context (the assembly is a native kernel)")
        self.layout, self.torch, self.ctx = layout, torch, ctx
        W, H = layout.width, layout.height
        if layout.mode == "interleaved":
            self.stripe_rows, self.owner = layout.stripe, np.asarray(layout.owner, dtype=np.int32)
        else:  # contiguous ceil(H/N) blocks (main.ts:444-449) = stripes of that many rows dealt in order
            r = max(1, -(-H // max(1, layout.world)))
            self.stripe_rows, self.owner = r, np.arange(-(-H // r), dtype=np.int32)
        self.recv2d = [torch.zeros(layout.world, layout.nbytes, dtype=torch.uint8, device=device) for _ in range(nbuf)]
        # one output frame per buffer set: frames in flight on different streams must not share it
        self.frames = [{s: torch.zeros(SECTION_BYTES[s] * W * H, dtype=torch.uint8, device=device)
                        for s in layout.sections} for _ in range(nbuf)]
        self.frame = self.frames[0]  # the most recently assembled frame

    def gather_lists(self):
        """Per slot, the list of per-rank views torch.distributed.gather writes into."""
        return [list(t.unbind(0)) for t in self.recv2d]

    def assemble(self, slot, acc=None):
        """acc: CUDA int64[4] that receives the combination of the ranks' partial diagnostics (layout.tail_offset)."""
        L = self.layout
        for k, s in enumerate(L.sections):
            self.ctx.assemble_frame(self.recv2d[slot], L.nbytes, L.offsets[s], SECTION_BYTES[s] * L.width, L.height,
                                    self.stripe_rows, self.owner, L.world, self.frames[slot][s],
                                    acc_offset=L.tail_offset if (acc is not None and k == 0) else -1,
                                    acc=acc if k == 0 else None)
        self.frame = self.frames[slot]
        return self.frame
