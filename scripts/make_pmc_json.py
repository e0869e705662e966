"""Digest of a scripts/profile_r02.sh capture -> profiles/r02/pmc_<workload>.json (what bench.py reads for
roofline.traffic and roofline.valu) and pmc_<workload>.txt (the per-kernel tables a reader can recompute from).

  traffic_bytes = (FETCH_SIZE + WRITE_SIZE) x 1024 per render-kernel launch (rocprofv3 reports KB; separate passes).
                  gfx950: FETCH_SIZE counts wide coalesced streaming reads at half their size (MI355X_MICROARCH.md);
                  this kernel's reads are ~1 MB of scene tables, no streaming input, so no doubling is applied.
  valu          = the kernel's VALU instruction mix (SQ_INSTS_VALU_* per launch) priced with the issue costs that
                  scripts/valu_issue_bench measured on this chip at 5 waves per SIMD (profiles/r02/valu_issue_costs.json):
                  weighted_issue_floor_ms = sum_class(count x cycles) / (SIMDs x clock).  Classes without a PMC bucket
                  (compares, selects, min / max, moves, lane reads: "OTHER" = SQ_INSTS_VALU - the buckets) are priced
                  at the mean of the measured OTHER-class instructions weighted 1 : 1 between 32-bit and 64-bit forms
                  unless the kernel's static ISA mix is given (--isa-mix file).
The file is stamped with the hash of the kernel sources and the option set; bench.py refuses a stale one."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_launch(out_dir, name, kernel_substr="render_kernel"):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(os.path.join(out_dir, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_substr not in r.get("Kernel_Name", ""):
                continue
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
    return {k: tot[k] / n[k] for k in tot}, (max(n.values()) if n else 0)


def main():
    wl, out_dir = sys.argv[1], sys.argv[2]
    import bench
    counters, launches = {}, 0
    for name in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_mix1", "pmc_mix2", "pmc_mix3"):
        c, n = per_launch(out_dir, name)
        counters.update(c)
        launches = max(launches, n)
    kstats = []
    for f in sorted(glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_stats.csv"), recursive=True)):
        kstats = [r for r in csv.DictReader(open(f))]
    render = [r for r in kstats if "render_kernel" in r.get("Name", "")]
    kernel_name = render[0]["Name"] if render else None
    kernel_ms = float(render[0]["AverageNs"]) * 1e-6 if render else None
    args = open(os.path.join(out_dir, "bench_args.txt")).read().split()
    options = {}
    for i, a in enumerate(args):
        if a == "--opt":
            k, v = args[i + 1].split("=")
            options[k] = int(v)
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:  # noqa: BLE001 -- the GPU box has no .git
        commit = None
    res = {"workload": wl, "source_sha16": bench.kernel_source_hash(), "commit_at_capture": commit, "options": options,
           "bench_args": " ".join(args), "kernel": kernel_name, "kernel_ms_rocprof": kernel_ms, "launches_counted": launches,
           "counters_per_launch": {k: counters[k] for k in sorted(counters)}}
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        res["fetch_kb"], res["write_kb"] = counters["FETCH_SIZE"], counters["WRITE_SIZE"]
        res["traffic_bytes"] = int(round((counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024))
    # ---- VALU mix priced with the measured issue costs
    costs_path = os.path.join(ROOT, "profiles", "r02", "valu_issue_costs.json")
    if "SQ_INSTS_VALU" in counters and os.path.exists(costs_path):
        costs = json.load(open(costs_path))
        cls = costs["classes"]
        w = "w5"
        cyc = lambda n: cls[n][w]["cycles"]  # noqa: E731
        clock_mhz = sorted(v[w]["clock_mhz"] for v in cls.values())[len(cls) // 2]
        bucket_cost = {  # PMC bucket -> cycles per wave-instruction (the instructions the kernel uses in that bucket)
            "ADD_F64": cyc("v_add_f64"), "MUL_F64": cyc("v_mul_f64"), "FMA_F64": cyc("v_fma_f64"),
            "TRANS_F64": 0.5 * (cyc("v_rcp_f64") + cyc("v_rsq_f64")),
            "ADD_F32": cyc("v_add_f32"), "MUL_F32": cyc("v_mul_f32"), "FMA_F32": cyc("v_fma_f32"), "TRANS_F32": cyc("v_sqrt_f32"),
            "INT32": 0.25 * (cyc("v_and_b32") + cyc("v_add_u32") + cyc("v_lshl_add_u32") + cyc("v_mul_lo_u32")),
            "INT64": cyc("v_mul_lo_u32"),
            "CVT": (cyc("v_cvt_f64_f32") + cyc("v_cvt_f32_f64") + cyc("v_cvt_i32_f32")) / 3.0,
        }
        total = counters["SQ_INSTS_VALU"]
        by = {k: counters.get("SQ_INSTS_VALU_" + k, 0.0) for k in bucket_cost}
        other = max(0.0, total - sum(by.values()))
        # OTHER: compares / selects / min-max / moves / lane reads.  Static ISA mix of the kernel if given, else an even
        # split between the 32-bit forms (2-cycle class) and the 64-bit forms (v_cmp_f64, v_min/max_f64).
        o32 = (cyc("v_cndmask_b32") + cyc("v_mov_b32") + cyc("v_cmp_lt_f32") + cyc("v_min_f32") + cyc("v_readlane_b32") + cyc("v_writelane_b32")) / 6.0
        o64 = (cyc("v_cmp_lt_f64") + cyc("v_min_f64") + cyc("v_max_f64") + cyc("v_rndne_f64")) / 4.0
        share64 = 0.5
        mixf = os.path.join(ROOT, "profiles", "r02", "isa_other_mix_%s.json" % wl)
        if os.path.exists(mixf):
            share64 = json.load(open(mixf))["other_share_64bit"]
        other_cost = (1 - share64) * o32 + share64 * o64
        cycles = sum(by[k] * bucket_cost[k] for k in by) + other * other_cost
        simds = costs["compute_units"] * 4
        floor_ms = cycles / simds / (clock_mhz * 1e6) * 1e3
        lane_util = None
        if counters.get("SQ_ACTIVE_INST_VALU") and counters.get("SQ_THREAD_CYCLES_VALU"):
            lane_util = counters["SQ_THREAD_CYCLES_VALU"] / counters["SQ_ACTIVE_INST_VALU"] / 64.0
        res["valu"] = {"insts": total, "by_class": dict(by, OTHER=other), "cycles_per_inst": dict(bucket_cost, OTHER=other_cost),
                       "other_share_64bit": share64, "waves_per_simd_priced": 5, "clock_mhz": clock_mhz, "simds": simds,
                       "weighted_issue_floor_ms": floor_ms, "lane_util": lane_util,
                       "salu_insts": counters.get("SQ_INSTS_SALU"), "lds_insts": counters.get("SQ_INSTS_LDS"),
                       "source": "SQ_INSTS_VALU_* per launch x profiles/r02/valu_issue_costs.json (w5)"}
    dst = os.path.join(ROOT, "profiles", "r02")
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(dst, "pmc_%s.json" % wl), "w") as f:
        json.dump(res, f, indent=1)
    with open(os.path.join(dst, "pmc_%s.txt" % wl), "w") as f:
        f.write("# %s  (scripts/profile_r02.sh; bench.py %s)\n" % (wl, " ".join(args)))
        f.write("== kernel stats (rocprofv3 --kernel-trace --stats)\n")
        for r in kstats:
            f.write("%-90s calls %-5s avg_ns %-12s total_ns %-12s pct %s\n" % (r.get("Name", "")[:90], r.get("Calls"), r.get("AverageNs"),
                                                                                r.get("TotalDurationNs"), r.get("Percentage")))
        f.write("== counters per render-kernel launch (mean over %d launches)\n" % launches)
        for k in sorted(counters):
            f.write("%-28s %16.6g\n" % (k, counters[k]))
    # gpurun merges gpurun_out back; profiles/ written on the GPU box is lost unless copied there too
    os.makedirs(os.path.join(ROOT, "gpurun_out", "profiles_r02"), exist_ok=True)
    for n in ("pmc_%s.json" % wl, "pmc_%s.txt" % wl):
        with open(os.path.join(dst, n)) as a, open(os.path.join(ROOT, "gpurun_out", "profiles_r02", n), "w") as b:
            b.write(a.read())
    print(json.dumps({k: res.get(k) for k in ("workload", "kernel", "kernel_ms_rocprof", "traffic_bytes")}))
    if "valu" in res:
        print(json.dumps(res["valu"]))


if __name__ == "__main__":
    main()
