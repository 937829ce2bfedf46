"""Per-forward kernel table from a rocprofv3 --kernel-trace --stats CSV: python scripts/kstats_csv.py <kernel_stats.csv> <n_forwards>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    per = int(r["Calls"]) / n
    us = float(r["AverageNs"]) / 1000
    tot += per * us
    print("%-84s per_fwd=%6.1f avg=%7.2fus sum=%7.1fus" % (r["Name"][:84], per, us, per * us))
print("kernel us per forward (rows shown): %.1f; launches per forward: %.1f" % (tot, sum(int(r["Calls"]) for r in rows) / n))
