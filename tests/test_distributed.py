"""Multi-process CPU tests (gloo, world_size 2 and 3) of the row-tile shard + gather path."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,mode,stripe,in_flight,weights", [
    (2, "interleaved", 16, 2, ""), (2, "contiguous", 16, 2, ""), (3, "interleaved", 4, 2, ""), (2, "interleaved", 8, 4, ""),
    (3, "interleaved", 4, 2, "400,1000,1000"),  # weighted deal: the gather's root renders a smaller share
    (2, "interleaved", 8, 3, "1,3")])
def test_shard_gather_reassemble(tmp_path, world, mode, stripe, in_flight, weights):
    out = tmp_path / "result.txt"
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), mode, str(stripe),
                                       str(out), str(in_flight), weights], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    assert out.read_text() == "OK"


def test_row_partitions_cover_the_frame(rm):
    from cpu_raymarcher_amd import distributed as D
    for H in (1, 15, 16, 17, 270, 2160):
        for world in (1, 2, 4, 8):
            for mode in ("interleaved", "contiguous"):
                rows = sorted(sum((D.owned_rows(H, world, r, mode, 16) for r in range(world)), []))
                assert rows[0][0] == 0 and rows[-1][1] == H
                assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
                cap = D.max_local_rows(H, world, mode, 16)
                assert all(sum(b - a for a, b in D.owned_rows(H, world, r, mode, 16)) <= cap for r in range(world))
    # the reference's contiguous rule (main.ts:444-449)
    assert D.owned_rows(2160, 8, 7, "contiguous") == [(1890, 2160)]


def test_layout_tail_travels_after_the_sections(rm):
    """bench.py gathers RGBA only and lets every rank's 32-byte diagnostics accumulator ride in the tail of its
    packed buffer: the tail must not overlap a section and the partial accumulators must combine exactly."""
    import numpy as np
    from cpu_raymarcher_amd import distributed as D
    plain = D.FrameLayout(3840, 2160, 8, ("rgba",), "interleaved", 16)
    tailed = D.FrameLayout(3840, 2160, 8, ("rgba",), "interleaved", 16, tail=32)
    assert tailed.offsets == plain.offsets and tailed.tail_offset == plain.nbytes
    assert tailed.nbytes == plain.nbytes + 256 and tailed.tail_offset % 256 == 0
    buf = np.zeros(tailed.nbytes, np.uint8)
    assert tailed.section(buf, "rgba").size == 4 * 3840 * tailed.cap <= tailed.tail_offset
    # sum / max / min of per-rank partial accumulators = the accumulator of the whole frame (main.ts:528-548)
    rng = np.random.default_rng(5)
    sdf = rng.integers(0, 900, size=2160 * 64).astype(np.uint16)
    parts = [sdf[r::8] for r in range(8)]
    assert sum(int(p.sum()) for p in parts) == int(sdf.sum())
    assert max(int(p.max()) for p in parts) == int(sdf.max()) and min(int(p.min()) for p in parts) == int(sdf.min())


def test_weighted_stripe_deal(rm):
    """rm_deal_stripes: equal weights give the round-robin deal of rm_render_stripes_device; any weights deal every
    stripe exactly once, in proportion, and spread every rank's stripes over the whole frame (no rank's stripes bunch
    at the light top / bottom of the frame)."""
    import numpy as np
    from cpu_raymarcher_amd import distributed as D
    from cpu_raymarcher_amd.context import deal_stripes
    for rows, stripe, n in ((2160, 16, 8), (2160, 7, 8), (131, 16, 3), (1, 16, 4), (0, 16, 2)):
        own = deal_stripes(rows, stripe, n)
        assert list(own) == [s % n for s in range(-(-rows // stripe))]
        assert D.owned_rows(rows, n, 1, "interleaved", stripe) == D.owned_rows(rows, n, 1, "interleaved", stripe, [5] * n)
    w = [550, 1000, 1000, 1000, 1000, 1000, 1000, 1000]
    own = deal_stripes(2160, 16, 8, w)
    cnt = np.bincount(own, minlength=8)
    assert cnt.sum() == 135 and all(abs(c - 135 * wi / sum(w)) < 1.0 for c, wi in zip(cnt, w))
    for r in range(8):  # gaps between consecutive stripes of a rank stay near total / weight
        ids = np.nonzero(own == r)[0]
        assert np.diff(ids).max() <= np.ceil(sum(w) / w[r]) + 1
    assert D.balanced_weights(8, 0.0, 0.2e-3) == [1000] * 8
    bw = D.balanced_weights(8, 0.05e-3, 0.2e-3)  # 50 us of root overhead against a 200 us shard
    assert bw[1:] == [1000] * 7 and 700 < bw[0] < 800
    assert D.balanced_weights(8, 1.0, 0.2e-3)[0] >= 200  # rank 0 keeps at least a quarter share
    lay = D.FrameLayout(3840, 2160, 8, ("rgba",), "interleaved", 16, tail=32, weights=bw)
    assert sorted(sum((lay.rows(r) for r in range(8)), [])) == [(a, min(a + 16, 2160)) for a in range(0, 2160, 16)]
    assert lay.cap == max(sum(b - a for a, b in lay.rows(r)) for r in range(8))
