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


def listing(source, extra=()):
    out = os.path.join("/tmp", "rm_inv_%d_%s.s" % (os.getpid(), os.path.basename(source)))
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only",
                    "-mllvm", "-amdgpu-inline-max-bb=100000", "-o", out, os.path.join(CSRC, source), *extra],
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900, check=True)
    try:
        return open(out).read().split("\n")
    finally:
        os.remove(out)


def spill_code_ahead_of_exec_restore(lines):
    """The code-generation defect behind round 2's wrong pixels (profiles/r03/spill_exec_hazard.txt): hipcc 7.2 placed a
    folded VGPR spill at the top of a JOIN block, ahead of the s_or_b64 exec that re-enables the lanes which skipped the
    branch -- those lanes never wrote the spill slot and reloaded garbage later.  Returns the (line number, text) of every
    folded VGPR spill / reload that sits between a block label and an EXEC-widening instruction of the same block."""
    bad, pending, whole_wave = [], [], False
    for n, raw in enumerate(lines):
        t = raw.strip()
        if not t or t.startswith((";", "//")):
            continue
        if t.endswith(":") or t.startswith(".LBB") or t.startswith("; %bb."):  # a new block: nothing pending carries over
            pending, whole_wave = [], False
            continue
        if t.startswith("."):
            continue
        op = t.split()[0]
        # (a function's prologue / epilogue saves its callee-saved VGPRs in whole-wave mode: s_or_saveexec_b64 sN, -1 ... spill ...
        # s_mov_b64 exec, sN.  There the spill runs with EVERY lane enabled and the following write of EXEC narrows: correct.)
        if re.match(r"(s_or_saveexec_b64\s+\S+,\s*-1|s_mov_b64\s+exec,\s*-1)", t):
            whole_wave = True
            continue
        if op.startswith("scratch_") and "Folded" in t:
            if not whole_wave:
                pending.append((n + 1, t))
            continue
        if whole_wave and re.match(r"\S+\s+exec\b", t):
            whole_wave = False
            continue
        widens = (op in ("s_or_b64", "s_mov_b64", "s_or_saveexec_b64", "s_xor_b64", "s_orn2_b64") and re.match(r"\S+\s+exec\b", t) is not None)
        if widens and pending:
            bad.extend(pending)
            pending = []
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
            pending = []
    return bad


def test_the_scanner_finds_the_round_2_pattern():
    sample = [".LBB3_549:", "\tscratch_store_dword off, v49, off offset:136 ; 4-byte Folded Spill", "\ts_or_b64 exec, exec, s[0:1]",
              "\tv_mov_b32_e32 v0, 0", ".LBB3_550:", "\ts_or_b64 exec, exec, s[2:3]", "\tscratch_load_dword v1, off, off offset:4 ; 4-byte Folded Reload"]
    assert [n for n, _ in spill_code_ahead_of_exec_restore(sample)] == [2]


@pytest.mark.parametrize("source,extra", [("rm_render_v2.hip", ()), ("rm_render_v2.hip", ("-DRM_LENGTH_SQRT",)), ("rm_kernels.hip", ()),
                                          ("rm_kernels.hip", ("-DRM_LENGTH_SQRT",))])
def test_no_spill_code_ahead_of_an_exec_restore(source, extra):
    bad = spill_code_ahead_of_exec_restore(listing(source, extra))
    assert not bad, bad[:5]
