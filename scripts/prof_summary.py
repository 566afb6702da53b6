"""Summarise a rocprofv3 rocpd database: kernel time per (name, grid) group.  python scripts/prof_summary.py DB [filter] [n_solves]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
nsolve = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = list(c.execute("select name, grid_x, count(*), sum(end-start), avg(end-start), min(end-start) from kernels group by name, grid_x order by 4 desc"))
tot = sum(r[3] for r in rows)
print("total kernel ms %.2f (per solve %.2f)" % (tot / 1e6, tot / 1e6 / nsolve))
for r in rows:
    if flt and flt not in r[0]: continue
    if r[3] < 0.002 * tot: continue
    print("%-78s g=%-8d n=%-5d tot %8.3f ms avg %8.1f us min %8.1f us" % (r[0][:78], r[1], r[2], r[3] / 1e6, r[4] / 1e3, r[5] / 1e3))
