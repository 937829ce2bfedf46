"""Last forward of a rocprofv3 kernel trace: python scripts/trace_tail.py <kernel_trace.csv> <first-kernel substring> [n]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gvc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]][-1]
prev_end, first = None, None
for r in rows[idx:idx + (int(sys.argv[3]) if len(sys.argv) > 3 else 16)]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    first = first or st
    print(f"{r['Kernel_Name'][:64]:64s} dur {(en - st) / 1000:6.2f} gap {((st - prev_end) / 1000 if prev_end else 0):6.2f} end@ {(en - first) / 1000:7.2f} "
          f"wgs {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
    prev_end = en
