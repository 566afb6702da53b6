"""One CG iteration and one multigrid setup, kernel by kernel (name, grid, duration, gap to the previous kernel), from a
rocprofv3 rocpd database of bench.py:  python scripts/prof_iteration.py DB [iteration|setup]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
what = sys.argv[2] if len(sys.argv) > 2 else "iteration"
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
src = "kernels" if "kernels" in tabs else next(t for t in tabs if "kernel" in t.lower())
rows = list(c.execute(f"select name, grid_x, start, end from {src} order by start"))
def short(n):
    n = n.replace("padne::", "").replace("void ", "")
    return n.split("(")[0][:56]
if what == "iteration":
    k = [i for i, r in enumerate(rows) if "pcg_update_p_z_kernel" in r[0]]
    a, b = k[-12], k[-11]          # one whole iteration in the middle of the last solve
    seg = rows[a + 1:b + 1]
else:
    k = [i for i, r in enumerate(rows) if "abs_range_kernel" in r[0]]
    e = [i for i, r in enumerate(rows) if "pcg_init_plain_kernel" in r[0]]
    a = k[-1]; b = [x for x in e if x > a][0]
    seg = rows[a:b]
t_prev = seg[0][2]; tot = 0
for n, g, s, e_ in seg:
    print(f"{short(n):58s} g={g:<7d} {(e_-s)/1e3:8.1f} us   gap {(s-t_prev)/1e3:7.1f}")
    tot += e_ - s; t_prev = e_
print(f"kernels {len(seg)}  busy {tot/1e3:.1f} us  span {(seg[-1][3]-seg[0][2])/1e3:.1f} us")
