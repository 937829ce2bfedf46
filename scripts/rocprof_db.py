"""Per-kernel table from a rocprofv3 sqlite output (rocprofv3 --kernel-trace -d DIR -o NAME): count, mean/min/max microseconds."""
import glob
import sqlite3
import sys

path = sys.argv[1]
dbs = [path] if path.endswith(".db") else glob.glob(path + "/**/*.db", recursive=True)
for db in dbs:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    by_grid = len(sys.argv) > 3 and sys.argv[3] == "grid"          # also split by launch geometry (the same kernel at different shapes)
    name = "s.kernel_name || ' grid ' || d.grid_size_x || 'x' || d.grid_size_y" if by_grid else "s.kernel_name"
    q = (f"select {name}, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start) "
         f"from {disp} d join {sym} s on d.kernel_id=s.id group by 1 order by 6 desc")
    rows = list(c.execute(q))
    total = sum(r[5] for r in rows)
    print(f"{db}: {total / 1e3:.1f} us of kernel time")
    for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
        print(f"{r[5] / total * 100:5.1f}%  n={r[1]:5d}  avg {r[2] / 1e3:8.1f}  min {r[3] / 1e3:8.1f}  max {r[4] / 1e3:8.1f} us  {r[0][:110]}")
