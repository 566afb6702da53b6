#!/bin/bash
# derived counters of the kernels of the lockstep solve (config C5): scripts/pmc_c5.sh OUTDIR "CTR1 CTR2" ...
OUT="$GRAFT_REPO_ROOT/$1"; shift; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $SET --output-format csv -d "$OUT/set$i" -o run -- python3 $GRAFT_REPO_ROOT/scripts/c5_only.py > "$OUT/set$i.log" 2>&1 || { tail -5 "$OUT/set$i.log"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/set*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("padne::", "")[:70] + " g=" + row.get("Grid_Size", row.get("Grid_Size_X", "?"))
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(85), " ".join(n.rjust(16) for n in names))
for k in sorted(acc):
    if not any(w in k for w in ("spmm", "pcg8")): continue
    v = acc[k]
    print(k.ljust(85), " ".join(("%16.2f" % (sum(v[n]) / len(v[n])) if v.get(n) else " " * 16) for n in names))
PY
