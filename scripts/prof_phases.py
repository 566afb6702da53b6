"""Per-solve span / busy / idle of the setup and of the iterations from a rocprofv3 rocpd database of bench.py."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
rows = list(c.execute("select name,start,end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'xw_plan_kernel' in r[0]]
for a, b in zip(idx[-4:-1], idx[-3:]):
    seg = rows[a:b]
    k = [i for i, r in enumerate(seg) if 'pcg_update_xr_entry' in r[0]]
    out = []
    for nm, sg in (("setup", seg[:k[0]]), ("iter", seg[k[0]:])):
        sp = (sg[-1][2] - sg[0][1]) / 1e6; bz = sum(r[2] - r[1] for r in sg) / 1e6
        out.append("%s span %.2f busy %.2f idle %.2f kernels %d" % (nm, sp, bz, sp - bz, len(sg)))
    print(" | ".join(out))
