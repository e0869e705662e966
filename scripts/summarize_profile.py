"""Condenses a scripts/profile.sh capture into a small text summary for profiles/."""
import csv, glob, os, sys, collections
d = sys.argv[1]
out = []
for f in sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    out.append("== kernel stats (%s)" % os.path.relpath(f, d))
    out.extend(l.rstrip() for l in open(f))
for name in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
    for f in sorted(glob.glob(os.path.join(d, name, "**", "*counter_collection.csv"), recursive=True)):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name"))
            agg[k][0] += float(row.get("Counter_Value", 0)); agg[k][1] += 1
        out.append("== %s: per-kernel counter mean per dispatch (dispatches)" % name)
        for (kn, cn), (s, n) in sorted(agg.items()):
            out.append("%-62s %-24s %18.1f  (%d)" % (kn, cn, s / n, n))
print("\n".join(out))
