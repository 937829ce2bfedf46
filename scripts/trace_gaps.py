"""GPU idle time inside the last utterance of a traced bench run: python scripts/trace_gaps.py <kernel_trace.csv> [chunks per utterance]"""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nchunk = int(sys.argv[2]) if len(sys.argv) > 2 else 10
idx = [i for i, r in enumerate(rows) if "k_hb_conv0" in r["Kernel_Name"]]
seg = rows[idx[-nchunk]:]
t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"wall {(t1 - t0) / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms, {len(seg)} kernels")
gaps, prev = Counter(), None
for r in seg:
    st = int(r["Start_Timestamp"])
    if prev is not None and st - prev > 2000:
        gaps[r["Kernel_Name"][:60]] += (st - prev) / 1000
    prev = max(prev or 0, int(r["End_Timestamp"]))
print(f"idle in gaps > 2 us: {sum(gaps.values()) / 1000:.2f} ms")
for n, g in gaps.most_common(12):
    print(f"{g:9.1f} us  before {n}")
