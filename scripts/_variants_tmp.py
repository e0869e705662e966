import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import cpu_raymarcher_amd as rm
sys.path.insert(0, "/root/repo/tests")
from test_gpu_parity import gpu_render, NAMES
ctx = rm.Context(0)
base = dict(kernel=2, coop=1, filter=1, nodes_in_lds=1, list_cap=32, tile_w=8, grid=1, refill=64, hw_xcd=1, blocks_per_cu=3, recs=1, lut=1, nn=2, sub=1, item_px=128, static=0, uniform=1, rel=1, lds_kb=32, cull=1, n0_batch=64, lpt=1)
variants = [dict(kernel=1), dict(kernel=2, list_cap=2), dict(kernel=2, list_cap=2, rel=0), dict(kernel=2, lds_kb=64), dict(kernel=2, lds_kb=64, rel=0),
            dict(kernel=2, list_cap=2, lpt=0), dict(kernel=2, list_cap=2, cull=0), dict(kernel=2, list_cap=2, uniform=0), dict(kernel=2, list_cap=2, n0_batch=1),
            dict(kernel=2, list_cap=2, rel=0, cull=0, uniform=0, lpt=0), dict(kernel=2, list_cap=4), dict(kernel=2, list_cap=8), dict(kernel=2, list_cap=2, nodes_in_lds=0)]
ref = None
for v in variants:
    for k, val in list(base.items()) + list(v.items()):
        try: ctx.set_option(k, val)
        except Exception: pass
    out = gpu_render(rm, ctx, 3, "BVH", 300, 170, (0.25, 0.6))
    if ref is None: ref = out
    d = [int((a != b).sum()) for a, b in zip(out, ref)]
    print(v, d, ctx.last_kernel(), flush=True)
