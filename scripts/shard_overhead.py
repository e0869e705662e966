"""Per-rank frame time of the N-GPU row-tile shard, measured on ONE GPU without any communication:
every rank's per-frame device work -- render its stripes + reduce its counters, and for rank 0 also the native
reassembly of the gathered frame + combined diagnostics (from stale receive buffers) -- with S frames in flight on
S streams, for the equal stripe deal and for the weighted deal bench.py computes (distributed.balanced_weights from
the same two measurements).  The N-GPU job's frame rate is bounded by its slowest rank: the table shows rank 0 is that
rank under the equal deal and is not under the weighted one.
Round 3: the rank's diagnostics come from the render launch itself (FUSED=1, default; FUSED=0: round 2's two reduction
launches), and GATHER=1 adds the call the real job makes per frame -- an asynchronous RCCL gather of the packed buffer on
the frame's stream, through a ONE-rank process group (a one-GPU box cannot host two ranks: the transfer is a local copy,
but the host-side cost of issuing it and of Work.wait() is the real one) -- to the host-enqueue column.
usage: python scripts/shard_overhead.py [frames] [S=12] [N=8 ...] [WL=C3|C5] [FUSED=0|1] [GATHER=0|1] [OPT=key=value ...]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cpu_raymarcher_amd as R
from cpu_raymarcher_amd import distributed as D


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 240
    S = ([int(a[2:]) for a in sys.argv[1:] if a.startswith("S=")] or [12])[0]
    worlds = [int(a[2:]) for a in sys.argv[1:] if a.startswith("N=")] or [8]
    wl = ([a[3:] for a in sys.argv[1:] if a.startswith("WL=")] or ["C3"])[0]
    fused = ([int(a[6:]) for a in sys.argv[1:] if a.startswith("FUSED=")] or [1])[0]
    gather = ([int(a[7:]) for a in sys.argv[1:] if a.startswith("GATHER=")] or [0])[0]
    W, H = 3840, 2160
    dev = torch.device("cuda:0")
    if gather:
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    ctx = R.Context(0)
    ctx.set_option("blocks_per_cu", 1)  # what bench.py uses with frames in flight
    ctx.set_option("lpt", 0)            # likewise: overlapping frames hide the tail the longest-first order is for
    for a in sys.argv[1:]:
        if a.startswith("OPT="):
            k, v = a[4:].split("=")
            ctx.set_option(k, int(v))
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
    print("workload %s, %dx%d, %d frames in flight, %d frames per measurement, diagnostics %s, %s"
          % (wl, W, H, S, frames, "fused into the render launch" if fused else "two reduction launches",
             "one-rank RCCL gather issued per frame" if gather else "no gather call"), flush=True)

    def rank_frame_ms(layout, rank, with_asm):
        cur = [None]
        ra = D.gpu_render_all(ctx, scene, W, H, "iteration-heatmap", layout, rank, extra=lambda packed: cur[0], diag_in_tail=bool(fused))
        sets = [dict(p=u8(layout.nbytes), sdf=i16(layout.cap * W), iters=i16(layout.cap * W), recv=[u8(layout.nbytes)] if gather else None,
                     work=None) for _ in range(S)]
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
                    if b["work"] is not None:
                        b["work"].wait()  # the buffer's previous gather has been consumed (ShardedFrameRenderer._submit)
                    ra(b["p"])
                    if not fused:
                        ctx.reduce_counters_enqueue(b["sdf"][:px], b["iters"][:px],
                                                    b["p"][layout.tail_offset:layout.tail_offset + 32].view(torch.int64))
                    if gather:
                        b["work"] = dist.gather(b["p"], b["recv"], dst=0, async_op=True)
                        if asm:
                            b["work"].wait()
                            b["work"] = None
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


    if gather:
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
