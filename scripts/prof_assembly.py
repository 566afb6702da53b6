"""Assembly and reduction kernels of one bench run from a rocprofv3 rocpd database (python scripts/prof_assembly.py DB)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
rows = list(c.execute("select name,start,end from kernels order by start"))
k = [i for i, r in enumerate(rows) if 'reduce_count' in r[0] or 'map_is_injective' in r[0]][0]
k2 = [i for i, r in enumerate(rows) if 'xw_plan' in r[0] or 'gershgorin' in r[0]][0]
for label, seg in (("assembly", rows[:k]), ("reduction", rows[k:k2])):
    tot = 0.0
    for r in seg:
        d = (r[2] - r[1]) / 1e3; tot += d
        if d > 20: print("%9.1f us %s" % (d, r[0].split('(')[0][-60:]))
    print("%s kernels total %.2f ms" % (label, tot / 1e3))
