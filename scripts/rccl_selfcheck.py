"""One-rank RCCL check of the calls the sharded bench path makes (a one-GPU box cannot host two RCCL ranks):
process group with device_id, async gather issued on a side stream into unbind() views of one [world, nbytes]
tensor, Work.wait() on that stream, the assembler, barrier and the MAX all-reduce of the elapsed time.
usage: python scripts/rccl_selfcheck.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

import cpu_raymarcher_amd as R
from cpu_raymarcher_amd import distributed as D


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    W, H = 1920, 1080
    ctx = R.Context(0)
    scene = R.Scene("BVH", ctx=ctx)
    scene.loadPreset(3)
    layout = D.FrameLayout(W, H, 1, ("rgba", "sdf", "iters"), "interleaved", 16)
    render_all = D.gpu_render_all(ctx, scene, W, H, "iteration-heatmap", layout, 0)
    S = 3
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    u8 = lambda n: torch.zeros(n, dtype=torch.uint8, device=dev)  # noqa: E731
    shr = D.ShardedFrameRenderer(layout, 0, 1, None, u8, dist, render_all=render_all, frames_in_flight=S,
                                 streams=streams)
    shr.world = 2  # take the gather branch although there is one rank: rank 0 gathers from itself
    asm = D.GpuFrameAssembler(layout, dev, S, ctx=ctx)
    shr.recv = asm.gather_lists()
    slots = [shr.submit() for _ in range(S)]
    for s in slots:
        shr.finish(s)
        with shr.on_stream(s):
            asm.assemble(s)
    dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tl = [torch.zeros(2, dtype=torch.float64, device=dev)]  # the calibration exchange of bench.py --root-share auto
    dist.all_gather(tl, torch.tensor([0.25, 0.5], dtype=torch.float64, device=dev))
    assert tl[0].tolist() == [0.25, 0.5]
    ok = True
    for s in slots:
        ok &= bool(torch.equal(asm.recv2d[s][0], shr.send[s]))
        ok &= bool(torch.equal(asm.frames[s]["rgba"], asm.frames[0]["rgba"]))
    ok &= int(asm.frames[0]["rgba"].view(torch.int32).ne(0).sum()) > 0 and float(t.item()) == 1.5
    print("rccl one-rank gather on side streams:", "ok" if ok else "MISMATCH")
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
