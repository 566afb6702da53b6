#!/bin/bash
# derived counters of the largest kernels of one bench step: scripts/pmc_kernels.sh OUTDIR "CTR1 CTR2 ..." (one pass per counter set)
OUT="$GRAFT_REPO_ROOT/$1"; shift; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $SET --output-format csv -d "$OUT/set$i" -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-seam --no-c5 --no-rank-proxy --no-small --no-dist-one-rank --no-launch-count > "$OUT/set$i.log" 2>&1 || { tail -5 "$OUT/set$i.log"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/set*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("padne::", "")[:60] + " g=" + row.get("Grid_Size", row.get("Grid_Size_X", "?"))
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
import os
want = tuple(os.environ["PMC_KERNELS"].split(",")) if os.environ.get("PMC_KERNELS") else ("spgemm_rows_lds", "spgemm_rows_lanes", "csr_spmv_kernel<2", "csr_spmv_kernel<6", "csr_spmv_kernel<1", "transpose_fill", "strength_mark", "agg_join", "prolong_rows_xw", "w_from_slots", "nbr_max_xw", "spgemm_count")
names = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(75), " ".join(n.rjust(16) for n in names))
for k in sorted(acc):
    if not any(w in k for w in want): continue
    v = acc[k]
    if max(len(x) for x in v.values()) == 0: continue
    print(k.ljust(75), " ".join(("%16.2f" % (sum(v[n]) / len(v[n])) if v.get(n) else " " * 16) for n in names))
PY
