#!/bin/bash
# counters of the assembly kernels of a config: scripts/pmc_asm.sh OUTDIR CFG "CTR1 CTR2 ..." ...   (one pass per counter set)
OUT="$GRAFT_REPO_ROOT/$1"; shift; CFG=$1; shift; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $SET --output-format csv -d "$OUT/set$i" -o run -- python3 $GRAFT_REPO_ROOT/scripts/asm_only.py $CFG > "$OUT/set$i.log" 2>&1 || { tail -5 "$OUT/set$i.log"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/set*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("padne::", "")[:44] + " g=" + row.get("Grid_Size", row.get("Grid_Size_X", "?"))
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(60), " ".join(n.rjust(18) for n in names))
for k in sorted(acc):
    if not any(w in k for w in ("asm_", "merge_rows")): continue
    v = acc[k]
    print(k.ljust(60), " ".join(("%18.0f" % (sum(v[n]) / len(v[n])) if v.get(n) else " " * 18) for n in names))
PY
