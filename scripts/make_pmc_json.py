"""Digest of a scripts/profile_round.sh capture -> profiles/<round>/pmc_<workload>.json (what bench.py reads for
roofline.traffic and roofline.valu) and pmc_<workload>.txt (the per-kernel tables a reader can recompute from).

  traffic_bytes = (FETCH_SIZE + WRITE_SIZE) x 1024 per render-kernel launch (rocprofv3 reports KB; separate passes).
                  gfx950: FETCH_SIZE counts wide coalesced streaming reads at half their size (MI355X_MICROARCH.md);
                  this kernel's reads are ~1 MB of scene tables, no streaming input, so no doubling is applied.
  valu          = added afterwards by scripts/price_valu.py (the instruction mix priced with the measured issue costs).
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


RENDER_KERNELS = ("render_kernel", "rm_rtc_render")  # the AOT kernels and a scene's run-time specialised one (rm_rtc.h)


def per_launch(out_dir, name, kernel_substr=RENDER_KERNELS):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    subs = (kernel_substr,) if isinstance(kernel_substr, str) else kernel_substr
    rows = [r for f in glob.glob(os.path.join(out_dir, name, "**", "*counter_collection.csv"), recursive=True) for r in csv.DictReader(open(f))
            if any(k in r.get("Kernel_Name", "") for k in subs)]
    # a configuration's first launches run in the library's own instantiation, the rest in the copy compiled for it
    # (rm_rtc_render_v2; option specialise_v2_after): the steady state is what is profiled
    if any("rm_rtc_render" in r["Kernel_Name"] for r in rows):
        rows = [r for r in rows if "rm_rtc_render" in r["Kernel_Name"]]
    for _f in (0,):
        for r in rows:
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
    render = [r for r in kstats if any(k in r.get("Name", "") for k in RENDER_KERNELS)]
    if any("rm_rtc_render" in r["Name"] for r in render):
        render = [r for r in render if "rm_rtc_render" in r["Name"]]
    kernel_name = render[0]["Name"] if render else None
    kernel_ms = float(render[0]["AverageNs"]) * 1e-6 if render else None
    # what bench.py itself reported as the kernel alone: for a run-time compiled copy of the v2 wave loop the instantiation it is a
    # copy of ("render_kernel_v2<...> [launch constants compiled in]"; the profiler only sees the symbol rm_rtc_render_v2)
    reported = None
    for f in sorted(glob.glob(os.path.join(out_dir, "*.log"))):
        for line in open(f, errors="replace"):
            if line.startswith("{") and '"roofline"' in line:
                try:
                    reported = json.loads(line)["roofline"].get("kernel") or reported
                except ValueError:
                    pass
    args = open(os.path.join(out_dir, "bench_args.txt")).read().split()
    options = {}
    for i, a in enumerate(args):
        if a == "--opt":
            k, v = args[i + 1].split("=")
            options[k] = int(v)
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:  # noqa: BLE001 -- the GPU box has no .git: scripts/gpu_battery.sh leaves the commit in .build_commit
        try:
            commit = open(os.path.join(ROOT, ".build_commit")).read().strip() or None
        except OSError:
            commit = None
    res = {"workload": wl, "source_sha16": bench.kernel_source_hash(), "commit_at_capture": commit, "options": options,
           "bench_args": " ".join(args), "kernel": kernel_name, "kernel_reported": reported, "kernel_ms_rocprof": kernel_ms, "launches_counted": launches,
           "counters_per_launch": {k: counters[k] for k in sorted(counters)}}
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        res["fetch_kb"], res["write_kb"] = counters["FETCH_SIZE"], counters["WRITE_SIZE"]
        res["traffic_bytes"] = int(round((counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024))
    # the `valu` block (mix priced with the measured issue costs) is added by scripts/price_valu.py in the build container,
    # where the kernel's .s listing is at hand
    rnd = os.environ.get("RM_ROUND", "r03")
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(dst, "pmc_%s.json" % wl), "w") as f:
        json.dump(res, f, indent=1)
    with open(os.path.join(dst, "pmc_%s.txt" % wl), "w") as f:
        f.write("# %s  (scripts/profile_round.sh; bench.py %s; commit %s)\n" % (wl, " ".join(args), commit))
        f.write("== kernel stats (rocprofv3 --kernel-trace --stats)\n")
        for r in kstats:
            f.write("%-90s calls %-5s avg_ns %-12s total_ns %-12s pct %s\n" % (r.get("Name", "")[:90], r.get("Calls"), r.get("AverageNs"),
                                                                                r.get("TotalDurationNs"), r.get("Percentage")))
        f.write("== counters per render-kernel launch (mean over %d launches)\n" % launches)
        for k in sorted(counters):
            f.write("%-28s %16.6g\n" % (k, counters[k]))
    # gpurun merges gpurun_out back; profiles/ written on the GPU box is lost unless copied there too
    os.makedirs(os.path.join(ROOT, "gpurun_out", "profiles_" + rnd), exist_ok=True)
    for n in ("pmc_%s.json" % wl, "pmc_%s.txt" % wl):
        with open(os.path.join(dst, n)) as a, open(os.path.join(ROOT, "gpurun_out", "profiles_" + rnd, n), "w") as b:
            b.write(a.read())
    print(json.dumps({k: res.get(k) for k in ("workload", "kernel", "kernel_ms_rocprof", "traffic_bytes")}))


if __name__ == "__main__":
    main()
