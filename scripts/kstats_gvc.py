"""gvc:: kernels of a rocprofv3 --kernel-trace --stats CSV, per forward: python scripts/kstats_gvc.py <kernel_stats.csv> <n_forwards> [name filter]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else "gvc::"
tot = cnt = 0.0
for r in rows:
    if flt not in r["Name"]:
        continue
    per = int(r["Calls"]) / n
    us = float(r["AverageNs"]) / 1000
    if per < 0.5:
        continue
    tot += per * us; cnt += per
    print("%-70s per_fwd=%5.1f avg=%7.2f min=%6.2f max=%6.2f us  sum=%7.1f us" % (r["Name"].replace("void ", "")[:70], per, us, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, per * us))
print("kernel time per forward: %.1f us over %.1f launches" % (tot, cnt))
