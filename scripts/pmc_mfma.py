"""Per-kernel MFMA utilisation from a `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv` run (the *_counter_collection.csv it writes).
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); GFLOP = MOPS x 512 / 1e9."""
import csv
import glob
import sys
from collections import defaultdict

path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(path + "/**/*counter_collection.csv", recursive=True)
agg = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
dur = defaultdict(dict)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
        dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("kernel,dispatches,avg_duration_us_under_pmc,avg_SQ_VALU_MFMA_BUSY_CYCLES,avg_GRBM_GUI_ACTIVE_sum_over_8_XCD,avg_MFMA_F32_GFLOP,MfmaUtil_percent,achieved_TFLOPs")
rows = []
for k, c in agg.items():
    n = len(disp[k])
    d = sum(dur[k].values()) / n
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / n
    gf = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) / n * 512 / 1e9
    util = 100.0 * busy / (gui / 8 * 1024) if gui else 0.0
    rows.append((d * n, k, n, d, busy, gui, gf, util, gf / d * 1e-3 * 1e3 if d else 0.0))
for _, k, n, d, busy, gui, gf, util, tf in sorted(rows, reverse=True)[:16]:
    print(f"\"{k[:90].replace(',', ';')}\",{n},{d:.1f},{busy:.0f},{gui:.0f},{gf:.3f},{util:.1f},{gf / (d * 1e-6) / 1e3:.1f}")
