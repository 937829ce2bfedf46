"""Kernel-by-kernel timeline of the LAST forward in a rocprofv3 kernel trace: python scripts/trace_timeline.py <kernel_trace.csv> <first-kernel substring>
prints name, duration, gap to the previous kernel's end (µs) and the totals."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "k_hb_conv0"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
seg = rows[idx[-1]:]
prev = None
busy = gaps = 0.0
for r in seg:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev) / 1e3 if prev is not None else 0.0
    name = r["Kernel_Name"].replace("void ", "").replace("gvc::", "")[:48]
    print(f"{name:48s} {(en - st) / 1e3:7.2f}  gap {gap:6.2f}  grid {r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')}x{r.get('Grid_Size_Z', '?')}")
    busy += (en - st) / 1e3
    gaps += max(gap, 0.0)
    prev = en
print(f"{len(seg)} kernels: busy {busy:.1f} us, gaps {gaps:.1f} us, span {(int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3:.1f} us")
