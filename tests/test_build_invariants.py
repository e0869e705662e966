"""Build-time invariants of the gfx950 kernels that a parity test cannot see until it is too late (CPU only: hipcc
cross-compiles).  Round 2 found one the hard way: the v2 REL instantiations spilled 21 VGPRs to scratch at five waves per
SIMD and rendered wrong pixels, non-deterministically.  Round 3: EVERY v2 instantiation, in both vec3.length builds, holds
80 VGPRs -- six waves per SIMD -- without a spilled VGPR and without scratch, and so do the sphere / primitive
instantiations of the one-ray-per-lane kernel (the expression-program ones, GEN = 2 / 3, keep their interpreter's stack:
listed here with their bound, so that a change that makes it grow shows up)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cpu_raymarcher_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("c++filt") is None, reason="hipcc / c++filt not present")


def resource_usage(extra=(), source="rm_render_v2.hip"):
    # the flags of csrc/Makefile (DEVFLAGS included: the inliner's basic-block limit decides what is inlined)
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only",
                          "-mllvm", "-amdgpu-inline-max-bb=100000",
                          "-Rpass-analysis=kernel-resource-usage", "-o", os.devnull, os.path.join(CSRC, source), *extra],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900).stdout.decode()
    rows, cur = [], None
    for line in out.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key in ("VGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "VGPRs"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and cur is not None and key not in cur:
                cur[key] = int(m.group(1))
    names = subprocess.check_output(["c++filt"] + [r["name"] for r in rows]).decode().split("\n")
    return {n.replace("(anonymous namespace)::", ""): r for n, r in zip(names, rows)}


def test_makefile_uses_the_flags_checked_here():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "-amdgpu-inline-max-bb=100000" in mk and "-ffp-contract=off" in mk


@pytest.mark.parametrize("extra", [(), ("-DRM_LENGTH_SQRT",)])
def test_v2_render_kernels_hold_six_waves_without_spills_or_scratch(extra):
    usage = resource_usage(extra)
    kernels = {n: r for n, r in usage.items() if "render_kernel_v2<" in n}
    assert len(kernels) >= 9, sorted(usage)
    for name, r in kernels.items():
        assert r["VGPRs Spill"] == 0 and r["ScratchSize [bytes/lane]"] == 0, (name, r)
        assert r["Occupancy [waves/SIMD]"] >= 6 and r["VGPRs"] <= 80, (name, r)


@pytest.mark.parametrize("extra", [(), ("-DRM_LENGTH_SQRT",)])
def test_one_ray_per_lane_kernels_spill_no_vgpr(extra):
    """render_kernel<ACCEL, OTHER, GEN> and distance_kernel<ACCEL, GEN>: no VGPR spill anywhere; no scratch for spheres (GEN 0)
    and general primitives (GEN 1).  The expression-program instantiations (GEN 2 / 3) run an interpreter whose out-of-line
    fdlibm calls and value stack live in scratch: bounded here, not yet removed."""
    usage = resource_usage(extra, "rm_kernels.hip")
    kernels = {n: r for n, r in usage.items() if n.startswith(("void render_kernel<", "void distance_kernel<"))}
    assert len(kernels) >= 24 + 12, sorted(usage)
    for name, r in kernels.items():
        gen = int(name.split("<")[1].split(">")[0].split(",")[-1])
        assert r["VGPRs Spill"] == 0, (name, r)
        if gen <= 1:
            assert r["ScratchSize [bytes/lane]"] == 0, (name, r)
        else:
            assert r["ScratchSize [bytes/lane]"] <= 800, (name, r)


@pytest.mark.parametrize("extra", [(), ("-DRM_LENGTH_SQRT",)])
def test_lean_octree_kernel_keeps_eight_waves_without_spills(extra):
    """render_kernel_oct is compiled for 64 VGPRs (eight waves per SIMD: its speed against render_kernel<1, false, 0> is
    occupancy as much as instruction count); a change that makes it spill VGPRs to scratch shows up here, not as a slow frame."""
    usage = resource_usage(extra, "rm_kernels.hip")
    k = [r for n, r in usage.items() if n.startswith("render_kernel_oct(")]
    assert len(k) == 1, sorted(usage)
    assert k[0]["VGPRs Spill"] == 0 and k[0]["ScratchSize [bytes/lane]"] == 0 and k[0]["Occupancy [waves/SIMD]"] == 8, k[0]
