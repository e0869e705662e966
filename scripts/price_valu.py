"""Adds the `valu` block to profiles/<round>/pmc_<workload>.json: the render kernel's measured VALU instruction mix priced
with the issue costs measured on this chip ("measure the real VALU ceiling").

  counts   SQ_INSTS_VALU and the SQ_INSTS_VALU_* buckets per launch (rocprofv3, scripts/profile_round.sh)
  costs    profiles/<round>/valu_issue_costs.json (scripts/valu_issue_bench): cycles one wave64 instruction costs a SIMD with
           five waves resident = the slowest wave's cycles / (instructions x 5)  [the mean over waves is lower only
           because waves that start late run part of the time with fewer than five on the SIMD]
  OTHER    = SQ_INSTS_VALU - sum(buckets): compares, selects, min / max, moves, lane reads ... have no PMC bucket.
           Their average cost is taken from the kernel's STATIC instruction mix (the .s listing of the instantiation
           that ran; run `make -C cpu_raymarcher_amd/csrc asm` first): every mnemonic outside the arithmetic buckets,
           weighted by its static count, priced with the measured class it belongs to.  Static, not dynamic, weights:
           stated here and in the JSON.
  weighted_issue_floor_ms = sum(count x cycles) / (SIMDs x clock): the time the chip needs merely to ISSUE this mix with
           every SIMD busy every cycle and five waves resident.  frac (in bench.py) = floor / kernel time.
usage: python scripts/price_valu.py C3 [C2 C5 ...]   (run in the build container, where the .s listings are)"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cpu_raymarcher_amd", "csrc")
PROF = os.path.join(ROOT, "profiles", os.environ.get("RM_ROUND", "r03"))

# mnemonic (regex) -> (PMC bucket or OTHER, measured class in valu_issue_costs.json)
RULES = [
    (r"v_(add|sub|subrev)_f64", "ADD_F64", "v_add_f64"), (r"v_mul_f64", "MUL_F64", "v_mul_f64"), (r"v_(fma|fmac)_f64", "FMA_F64", "v_fma_f64"),
    (r"v_(rcp|rsq|sqrt)_f64", "TRANS_F64", "v_rcp_f64"),
    (r"v_(add|sub|subrev)_f32", "ADD_F32", "v_add_f32"), (r"v_mul_f32", "MUL_F32", "v_mul_f32"), (r"v_(fma|fmac|mad)_f32", "FMA_F32", "v_fma_f32"),
    (r"v_pk_add_f32", "ADD_F32", "v_pk_add_f32"), (r"v_pk_mul_f32", "MUL_F32", "v_pk_mul_f32"), (r"v_pk_fma_f32", "FMA_F32", "v_pk_fma_f32"),
    (r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f32", "TRANS_F32", "v_sqrt_f32"),
    (r"v_cvt_", "CVT", "v_cvt_f32_f64"),
    (r"v_(and|or|xor|not)_b32|v_(add|sub|subrev)_(u32|i32|co_u32)|v_(addc|subb)_co_u32", "INT32", "v_and_b32"),
    (r"v_(lshl_add|add_lshl|lshl_or|and_or|or3|xad|add3)_u32|v_(bfe|bfi|perm|alignbit|lshlrev|lshrrev|ashrrev|mul_lo|mul_hi|mad_u32|mad_i32|mul_u32|mul_i32|min_u|max_u|min_i|max_i|med3_i|med3_u|sad)",
     "INT32", "v_lshl_add_u32"),
    (r"v_(lshlrev|lshrrev|ashrrev)_b64|v_lshl_add_u64|v_mad_u64|v_mad_i64", "INT64", "v_mul_lo_u32"),
    # ---- no PMC bucket ("OTHER")
    (r"v_mov_b32_dpp|v_.*_dpp", "OTHER", "v_mov_b32_dpp"),
    (r"v_mov_b64|v_mov_b32|v_accvgpr|v_swap", "OTHER", "v_mov_b32"),
    (r"v_readlane|v_readfirstlane", "OTHER", "v_readlane_b32"), (r"v_writelane", "OTHER", "v_writelane_b32"),
    (r"v_cndmask", "OTHER", "v_cndmask_b32 (SGPR-pair mask, VOP3)"),  # the VOP2 form is not slower in the real kernels (A/B)
    (r"v_cmpx?_.*_f64|v_cmpx?_class_f64", "OTHER", "v_cmp_lt_f64"), (r"v_cmpx?_", "OTHER", "v_cmp_lt_f32"),
    (r"v_(min|max)_f64", "OTHER", "v_min_f64"), (r"v_(min|max|med3)_f32|v_(min3|max3)", "OTHER", "v_min_f32"),
    (r"v_(rndne|trunc|floor|ceil|fract|frexp_mant|frexp_exp)_f64|v_ldexp_f64", "OTHER", "v_rndne_f64"),
    (r"v_div_scale", "OTHER", "v_div_scale_f64"), (r"v_div_fmas", "OTHER", "v_div_fmas_f64"), (r"v_div_fixup", "OTHER", "v_div_fixup_f64"),
    (r"v_mbcnt", "OTHER", "v_mbcnt_lo_u32_b32"), (r"v_(rndne|trunc|floor|ceil|fract|ldexp|frexp).*_f32", "OTHER", "v_min_f32"),
    (r"v_nop", "OTHER", "v_mov_b32"),
]


def classify(mn):
    for rx, bucket, cls in RULES:
        if re.match(rx, mn):
            return bucket, cls
    return "OTHER", "v_min_f64"  # unknown VALU mnemonic: priced in the 4-cycle class


def static_mix(kernel_demangled):
    """Static VALU mnemonic counts of the kernel in the .s listings (matched through llvm-cxxfilt)."""
    filt = "c++filt"
    want = re.sub(r"^void ", "", kernel_demangled or "").replace("(anonymous namespace)::", "")
    for fn in ("rm_render_v2.s", "rm_kernels.s"):
        path = os.path.join(CSRC, fn)
        if not os.path.exists(path):
            continue
        lines = open(path).read().split("\n")
        labels = [(i, l.split(":")[0]) for i, l in enumerate(lines) if l.startswith("_Z") and ":" in l]
        names = subprocess.check_output([filt] + [l for _, l in labels]).decode().split("\n") if labels else []
        for (i, _), dem in zip(labels, names):
            dem = re.sub(r"^void ", "", dem).replace("(anonymous namespace)::", "")
            if dem.replace(" ", "") == want.replace(" ", ""):
                mix = collections.Counter()
                for l in lines[i + 1:]:
                    if "s_endpgm" in l:
                        break
                    m = re.match(r"^\s+(v_[a-z0-9_]+)", l)
                    if m:
                        mix[re.sub(r"_e(32|64)$", "", m.group(1))] += 1
                return mix, fn
    return None, None


def main():
    costs = json.load(open(os.path.join(PROF, "valu_issue_costs.json")))
    cls = costs["classes"]
    # waves per SIMD the kernel runs at: v2 kernels six (round 3), the lean octree kernel eight (priced at the most the
    # micro-benchmark measures: six)
    W = "w6" if "w6" in next(iter(cls.values())) else "w5"
    cyc = lambda n: cls[n][W]["cycles_slowest_wave"]  # noqa: E731
    clock = sorted(v[W]["clock_mhz"] for v in cls.values())[len(cls) // 2]
    simds = costs["compute_units"] * 4
    for wl in sys.argv[1:]:
        path = os.path.join(PROF, "pmc_%s.json" % wl)
        pj = json.load(open(path))
        c = pj["counters_per_launch"]
        kname = pj.get("kernel")
        if kname and kname.startswith("rm_rtc_render_v2") and pj.get("kernel_reported"):
            # the wave loop compiled for its configuration: the static mix of the unbucketed instructions is taken from the
            # ahead-of-time instantiation it is a copy of (the literals remove scalar loads and dead branches, not the mix's shape)
            kname = pj["kernel_reported"].split(" [")[0] + "(RmRenderParams)"
        mix, listing = static_mix(kname)
        if mix is None:
            print("%s: kernel %r not found in the .s listings (run `make -C cpu_raymarcher_amd/csrc asm`)" % (wl, pj.get("kernel")))
            continue
        # static cost per bucket: each bucket's mnemonics weighted by static count
        num, den = collections.defaultdict(float), collections.defaultdict(float)
        other_detail = collections.Counter()
        for mn, n in mix.items():
            b, k = classify(mn)
            num[b] += n * cyc(k)
            den[b] += n
            if b == "OTHER":
                other_detail[mn] += n
        cost = {b: num[b] / den[b] for b in den}
        buckets = ["ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT"]
        by = {b: c.get("SQ_INSTS_VALU_" + b, 0.0) for b in buckets}
        total = c["SQ_INSTS_VALU"]
        by["OTHER"] = max(0.0, total - sum(by.values()))
        dflt = {"ADD_F64": "v_add_f64", "MUL_F64": "v_mul_f64", "FMA_F64": "v_fma_f64", "TRANS_F64": "v_rcp_f64", "ADD_F32": "v_add_f32",
                "MUL_F32": "v_mul_f32", "FMA_F32": "v_fma_f32", "TRANS_F32": "v_sqrt_f32", "INT32": "v_lshl_add_u32", "INT64": "v_mul_lo_u32",
                "CVT": "v_cvt_f32_f64", "OTHER": "v_min_f64"}
        cpi = {b: cost.get(b, cyc(dflt[b])) for b in by}
        cycles = sum(by[b] * cpi[b] for b in by)
        floor_ms = cycles / simds / (clock * 1e6) * 1e3
        lane_util = None
        if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
            lane_util = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] / 64.0
        pj["valu"] = {
            "insts": total, "by_class": by, "cycles_per_inst": cpi, "issue_cycles_by_class": {b: by[b] * cpi[b] for b in by},
            "weighted_issue_floor_ms": floor_ms, "kernel_ms_rocprof": pj.get("kernel_ms_rocprof"),
            "frac_of_kernel_alone": floor_ms / pj["kernel_ms_rocprof"] if pj.get("kernel_ms_rocprof") else None,
            "lane_util": lane_util, "waves_per_simd_priced": int(W[1:]), "clock_mhz": clock, "simds": simds,
            "salu_insts": c.get("SQ_INSTS_SALU"), "lds_insts": c.get("SQ_INSTS_LDS"),
            "other_static_top": dict(other_detail.most_common(12)), "static_listing": listing,
            "source": "SQ_INSTS_VALU_* per launch (rocprofv3) x profiles/<round>/valu_issue_costs.json (cycles_slowest_wave at %s waves per "
                      "SIMD); per-bucket costs weighted by the kernel's STATIC mnemonic mix (%s)" % (W[1:], listing)}
        json.dump(pj, open(path, "w"), indent=1)
        print("%s: %.3g VALU insts, issue floor %.3f ms vs kernel alone %.3f ms (%.0f %%), lane util %.1f %%; cycles/inst %s"
              % (wl, total, floor_ms, pj.get("kernel_ms_rocprof") or 0, 100 * floor_ms / (pj.get("kernel_ms_rocprof") or 1),
                 100 * (lane_util or 0), {b: round(v, 2) for b, v in cpi.items()}))


if __name__ == "__main__":
    main()
