"""Rank 0's per-frame cost in the sharded path, measured on ONE GPU without any communication:
render the stripes of part 0 of N, reassemble a frame from N (stale) receive buffers, reduce the
counters.  It bounds the frames/s the N-GPU job can reach if the gather is hidden completely, and
shows how much of a frame is host-side launch overhead.
usage: python scripts/shard_overhead.py [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cpu_raymarcher_amd as R
from cpu_raymarcher_amd import distributed as D


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    W, H = 3840, 2160
    dev = torch.device("cuda:0")
    ctx = R.Context(0)
    scene = R.Scene("BVH", ctx=ctx)
    scene.loadPreset(3)
    acc = torch.zeros(4, dtype=torch.int64, device=dev)
    for world in (1, 2, 4, 8):
        layout = D.FrameLayout(W, H, world, ("rgba", "sdf", "iters"), "interleaved", 16)
        render_all = D.gpu_render_all(ctx, scene, W, H, "iteration-heatmap", layout, 0)
        packed = torch.zeros(layout.nbytes, dtype=torch.uint8, device=dev)
        asm = D.GpuFrameAssembler(layout, dev, 1)

        def frame():
            render_all(packed)
            fr = asm.assemble(0)
            ctx.reduce_counters_enqueue(fr["sdf"].view(torch.int16), fr["iters"].view(torch.int16), acc)

        for _ in range(5):
            frame()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            frame()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            render_all(packed)
        e1.record()
        torch.cuda.synchronize()
        print("N=%d  rank-0 frame: %.3f ms wall (host enqueue %.3f ms)  render of 1/%d of the rows: %.3f ms  -> <= %.0f frames/s"
              % (world, 1e3 * t_all / frames, 1e3 * t_host / frames, world, e0.elapsed_time(e1) / 20, frames / t_all), flush=True)


if __name__ == "__main__":
    main()
