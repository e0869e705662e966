"""Worker for tests/test_distributed.py: world_size-N gloo job on CPU.  The tile renderer is
the oracle (this is a test of the shard / gather / reassembly plumbing, which is the same
code the GPU ranks run with rm_render_tile_device as the renderer)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpu_raymarcher_amd import distributed as D  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    mode, stripe, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 96, 70  # H not a multiple of stripe * world
    sections = ("rgba", "sdf", "iters", "depth", "normal")
    weights = [int(w) for w in sys.argv[5].split(",")] if len(sys.argv) > 5 and sys.argv[5] else None
    layout = D.FrameLayout(W, H, world, sections, mode, stripe, weights=weights)
    sc = O.OracleScene(preset=3, accel="BVH")
    sc.set_angles(0.2, 0.4)

    def render_rows(y0, y1, local, packed):
        d, n, s, i = sc.render(W, H, y0, y1)
        rgba = O.shade("iteration-heatmap", d, n, s, i, W, y1 - y0)
        pk = packed.numpy()
        for name, arr in (("rgba", rgba), ("sdf", s.view(np.uint8)), ("iters", i.view(np.uint8)), ("depth", d), ("normal", n)):
            bpp = D.SECTION_BYTES[name]
            D.FrameLayout.section(layout, pk, name)[local * W * bpp:local * W * bpp + arr.size] = arr

    in_flight = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    r = D.ShardedFrameRenderer(layout, rank, world, render_rows, lambda n: torch.zeros(n, dtype=torch.uint8), dist,
                               frames_in_flight=in_flight)
    frame = D.new_frame(layout, lambda n: torch.zeros(n, dtype=torch.uint8)) if rank == 0 else None
    slots = [r.submit() for _ in range(in_flight + 1)]  # one more frame than buffer sets: a set is reused
    for s in slots[-in_flight:]:
        r.finish(s, frame)
    r.drain()
    ok = True
    if rank == 0:
        d, n, s, i = sc.render(W, H)
        rgba = O.shade("iteration-heatmap", d, n, s, i, W, H)
        want = {"rgba": rgba, "sdf": s.view(np.uint8), "iters": i.view(np.uint8), "depth": d, "normal": n}
        for k in sections:
            ok = ok and np.array_equal(frame[k].numpy(), want[k])
        rows = sorted(sum((layout.rows(q) for q in range(world)), []))
        ok = ok and rows[0][0] == 0 and rows[-1][1] == H and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
        with open(out, "w") as f:
            f.write("OK" if ok else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
