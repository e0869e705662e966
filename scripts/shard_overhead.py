"""Per-rank frame time of the N-GPU row-tile shard, measured on ONE GPU without any communication:
every rank's per-frame device work -- render its stripes + reduce its counters, and for rank 0 also the native
reassembly of the gathered frame + combined diagnostics (from stale receive buffers) -- with S frames in flight on
S streams, for the equal stripe deal and for the weighted deal bench.py computes (distributed.balanced_weights from
the same two measurements).  The N-GPU job's frame rate is bounded by its slowest rank: the table shows rank 0 is that
rank under the equal deal and is not under the weighted one.
usage: python scripts/shard_overhead.py [frames] [S=12] [N=8 ...] [WL=C3|C5]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cpu_raymarcher_amd as R
from cpu_raymarcher_amd import distributed as D


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 240
    S = ([int(a[2:]) for a in sys.argv[1:] if a.startswith("S=")] or [12])[0]
    worlds = [int(a[2:]) for a in sys.argv[1:] if a.startswith("N=")] or [8]
    wl = ([a[3:] for a in sys.argv[1:] if a.startswith("WL=")] or ["C3"])[0]
    W, H = 3840, 2160
    dev = torch.device("cuda:0")
    ctx = R.Context(0)
    ctx.set_option("blocks_per_cu", 1)  # what bench.py uses with frames in flight
    ctx.set_option("lpt", 0)            # likewise: overlapping frames hide the tail the longest-first order is for
    if wl == "C5":
        from cpu_raymarcher_amd.synthetic import synthetic_spheres
        scene = R.Scene("Octree", ctx=ctx)
        sp = synthetic_spheres(10000)
        scene.loadSpheres(sp[:, :3], sp[:, 3])
    else:
        scene = R.Scene("BVH", ctx=ctx)
        scene.loadPreset(3)
    streams = [torch.cuda.Stream() for _ in range(S)]
    u8 = lambda n: torch.zeros(n, dtype=torch.uint8, device=dev)  # noqa: E731
    i16 = lambda n: torch.zeros(n, dtype=torch.int16, device=dev)  # noqa: E731
    print("workload %s, %dx%d, %d frames in flight, %d frames per measurement" % (wl, W, H, S, frames), flush=True)

    def rank_frame_ms(layout, rank, with_asm):
        cur = [None]
        ra = D.gpu_render_all(ctx, scene, W, H, "iteration-heatmap", layout, rank, extra=lambda packed: cur[0])
        sets = [dict(p=u8(layout.nbytes), sdf=i16(layout.cap * W), iters=i16(layout.cap * W)) for _ in range(S)]
        px = W * sum(b - a for a, b in layout.rows(rank))
        asm = D.GpuFrameAssembler(layout, dev, S, ctx=ctx) if with_asm else None
        acc = torch.zeros(4, dtype=torch.int64, device=dev)

        def run(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in range(n):
                k = f % S
                b = sets[k]
                with torch.cuda.stream(streams[k]):
                    cur[0] = b
                    ra(b["p"])
                    ctx.reduce_counters_enqueue(b["sdf"][:px], b["iters"][:px],
                                                b["p"][layout.tail_offset:layout.tail_offset + 32].view(torch.int64))
                    if asm:
                        asm.assemble(k, acc)
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / n, 1e3 * t_host / n
        run(2 * S)
        return run(frames)

    for world in worlds:
        equal = D.FrameLayout(W, H, world, ("rgba",), "interleaved", 16, tail=32)
        t_shard, h_shard = rank_frame_ms(equal, 1 % world, False)
        t_root, h_root = rank_frame_ms(equal, 0, True)
        weights = D.balanced_weights(world, max(0.0, t_root - t_shard) * 1e-3, t_shard * 1e-3)
        bal = D.FrameLayout(W, H, world, ("rgba",), "interleaved", 16, tail=32, weights=weights)
        b_root, hb_root = rank_frame_ms(bal, 0, True)
        b_other, hb_other = rank_frame_ms(bal, 1 % world, False)
        rows = lambda lay, r: sum(b - a for a, b in lay.rows(r))  # noqa: E731
        print("N=%d  equal deal   : rank 0 (render %4d rows + reassembly + diagnostics) %.3f ms/frame | rank 1 (render %4d rows) %.3f ms/frame"
              "  -> job <= %.0f frames/s   [host enqueue %.3f / %.3f ms]"
              % (world, rows(equal, 0), t_root, rows(equal, 1 % world), t_shard, 1e3 / max(t_root, t_shard), h_root, h_shard), flush=True)
        print("N=%d  weighted deal: rank 0 (render %4d rows + reassembly + diagnostics) %.3f ms/frame | rank 1 (render %4d rows) %.3f ms/frame"
              "  -> job <= %.0f frames/s   weights %s [host enqueue %.3f / %.3f ms]"
              % (world, rows(bal, 0), b_root, rows(bal, 1 % world), b_other, 1e3 / max(b_root, b_other), weights[:2], hb_root, hb_other), flush=True)


if __name__ == "__main__":
    main()
