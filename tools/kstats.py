"""Per-step summary of a rocprofv3 kernel_stats.csv (dev tool): python tools/kstats.py <csv> <steps> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel ms/step %.3f  launches/step %.1f" % (tot / 1e6 / steps, sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:top]:
    print("  %-56s calls/step %5.1f  avg %8.1f us  us/step %8.1f" % (r["Name"][:56], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                                      float(r["TotalDurationNs"]) / 1e3 / steps))
