"""Mean idle time in front of every kernel of the CG loop (start minus the previous kernel's end, same stream order), from a
rocprofv3 rocpd database of bench.py:  python scripts/prof_gaps.py DB"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
src = "kernels" if "kernels" in tabs else next(t for t in tabs if "kernel" in t.lower())
rows = list(c.execute(f"select name, grid_x, start, end from {src} order by start"))
def short(n):
    return n.replace("padne::", "").replace("void ", "").split("(")[0][:70]
k = [i for i, r in enumerate(rows) if "pcg_update_p_z_kernel" in r[0]]
a, b = k[-27], k[-1]                      # the iterations of the last solve
gaps = collections.defaultdict(list)
for i in range(a + 1, b + 1):
    gaps[(short(rows[i][0]), rows[i][1])].append((rows[i][2] - rows[i - 1][3]) / 1e3)
tot = 0.0
for (n, g), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:72s} g={g:<8d} n={len(v):3d} mean gap {sum(v)/len(v):6.2f} us  max {max(v):6.2f}")
    tot += sum(v)
print(f"idle between kernels: {tot / 26:.1f} us per iteration")
