"""rocprofv3 kernel_stats.csv -> the markdown table kept under profiles/ (usage: stats_md.py <csv> '<title>' > out.md)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("# %s\n" % sys.argv[2])
print("| kernel | calls | avg ms | min ms | max ms | % |\n|---|---|---|---|---|---|")
for r in rows:
    name = r.get("Name") or r.get("KernelName")
    print("| `%s` | %s | %.3f | %.3f | %.3f | %s |" % (name, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6,
                                                       float(r["MaxNs"]) / 1e6, r["Percentage"]))
