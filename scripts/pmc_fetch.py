"""Average FETCH_SIZE / WRITE_SIZE (KB) per kernel and grid size from a rocprofv3 --pmc ... --output-format csv run."""
import collections
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not files:
    print("no counter_collection.csv under", sys.argv[1], glob.glob(sys.argv[1] + "/**/*", recursive=True)[:20])
    sys.exit(0)
agg = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if len(sys.argv) < 3 or any(p in r["Kernel_Name"] for p in sys.argv[2:]):
            agg[(r["Kernel_Name"][:60], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, "n", len(v), "avg KB %.0f" % (sum(v) / len(v)))
