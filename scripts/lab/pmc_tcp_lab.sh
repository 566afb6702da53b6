#!/bin/bash
OUT="$GRAFT_REPO_ROOT/$1"; mkdir -p "$OUT"; cd /tmp
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d "$OUT/tcp" -- $GRAFT_REPO_ROOT/scripts/bin/spmv_lab 8 1118 > "$OUT/tcp.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d "$OUT/sq" -- $GRAFT_REPO_ROOT/scripts/bin/spmv_lab 8 1118 > "$OUT/sq.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tab = collections.OrderedDict()
for sub in ("tcp", "sq"):
    rows = []
    for f in glob.glob(out + f"/{sub}/*/*_counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"].split("(")[0][-40:], row["Grid_Size"], row["Counter_Name"], float(row["Counter_Value"])))
    rows.sort()
    for d, k, g, c, v in rows:
        tab.setdefault((k, g), {})[c] = v     # last dispatch of each (kernel, grid) wins
cols = ["TCP_TOTAL_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum",
        "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"]
print("kernel grid " + " ".join(c.replace("TCP_", "").replace("SQ_", "").replace("_sum", "") for c in cols))
for (k, g), d in tab.items():
    if "fill" in k: continue
    print(f"{k:40s} {g:>8s} " + " ".join(f"{d.get(c, float('nan'))/1e6:9.2f}" for c in cols))
PY
