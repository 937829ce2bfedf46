import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, count(*), avg(end-start), min(end-start), sum(end-start) from kernels group by name order by sum(end-start) desc").fetchall()
tot = sum(r[4] for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{r[0][:62]:62s} n={r[1]:6d} avg={r[2]/1000:7.2f}us min={r[3]/1000:6.2f} {100*r[4]/tot:5.1f}%")
