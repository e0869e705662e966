"""Build-container only (skipped wherever /root/reference or node is absent, e.g. on the GPU box): the reference's own
TypeScript control flow -- read as text at run time, type syntax removed in memory, a restated gl-matrix underneath --
must produce the oracle's bytes (scripts/ref_crosscheck.py).  Not a reference build; it does not pin parity
(DESIGN.md 5), it guards the oracle's reading of bvh.ts / octree.ts / scene.ts / the marchers against misreadings."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/src") or shutil.which("node") is None,
                                reason="reference sources / node not present (build container only)")


def test_reference_control_flow_matches_the_oracle():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ref_crosscheck.py"), "--quick"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=1500)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    assert "8 configurations, 0 differ" in out, out


def test_type_stripper_leaves_no_type_syntax():
    """Every module of the render path compiles after stripping (run.js exits 3 otherwise) for each marcher and shader."""
    import json
    import tempfile
    for alg, shader in (("fixed-step", "phong"), ("adaptive-step", "sdf-heatmap"), ("adaptive-step-v2", "iteration-heatmap"),
                        ("adaptive-step-v3", "normal")):
        with tempfile.TemporaryDirectory() as td:
            cfg = os.path.join(td, "c.json")
            json.dump(dict(preset=9, accel="Octree", width=24, height=16, algorithm=alg, shader=shader), open(cfg, "w"))
            out = subprocess.check_output(["node", os.path.join(ROOT, "scripts", "ref_crosscheck", "run.js"), "/root/reference/src", cfg],
                                          timeout=300)
            assert json.loads(out)["n_objects"] > 0
