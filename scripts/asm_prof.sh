#!/bin/bash
# kernel trace of the device-resident assembly of a config: scripts/asm_prof.sh [C4] -> gpurun_out/asm_<cfg>_timeline.txt
CFG=${1:-C4}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_asm
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/prof_asm -o p -- python3 scripts/asm_only.py $CFG > gpurun_out/asm_${CFG}.log 2>&1 || { tail -20 gpurun_out/asm_${CFG}.log; exit 1; }
DB=$(ls gpurun_out/prof_asm/*.db | head -1)
python3 - $DB > gpurun_out/asm_${CFG}_timeline.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
src = "kernels" if "kernels" in tabs else next(t for t in tabs if "kernel" in t.lower())
rows = list(c.execute(f"select name, grid_x, start, end from {src} order by start"))
k = [i for i, r in enumerate(rows) if "asm_count_tri" in r[0]]
a = k[-1]
seg = rows[a - 8:]
t_prev = seg[0][2]; tot = 0
for n, g, s, e in seg:
    n = n.replace("padne::", "").replace("void ", "").split("(")[0][:56]
    print(f"{n:58s} g={g:<9d} {(e-s)/1e3:8.1f} us   gap {(s-t_prev)/1e3:7.1f}")
    tot += e - s; t_prev = e
print(f"kernels {len(seg)}  busy {tot/1e3:.1f} us  span {(seg[-1][3]-seg[0][2])/1e3:.1f} us")
PY
rm -f gpurun_out/prof_asm/*.db
tail -4 gpurun_out/asm_${CFG}.log
