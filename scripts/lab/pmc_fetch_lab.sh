#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every lab variant (one dispatch row per launch)
OUT="$GRAFT_REPO_ROOT/$1"; mkdir -p "$OUT"; cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $GRAFT_REPO_ROOT/scripts/bin/spmv_lab 8 1118 > "$OUT/fetch.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/fetch/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"].split("(")[0][-45:], row["Grid_Size"], float(row["Counter_Value"])))
rows.sort()
last = None
for d, k, g, v in rows:
    key = (k, g)
    if key != last:
        print(f"{k:46s} grid={g:>9s} FETCH_SIZE={v/1024:9.1f} MB  x2={2*v/1024:9.1f} MB")
        last = key
PY
