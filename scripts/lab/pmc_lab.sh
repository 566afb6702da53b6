#!/bin/bash
# usage: scripts/pmc_lab.sh <variant substring> <outdir>   (run on the GPU box)
V="$1"; OUT="$2"; case "$OUT" in /*) ;; *) OUT="$GRAFT_REPO_ROOT/$OUT";; esac; mkdir -p "$OUT"; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- $GRAFT_REPO_ROOT/scripts/bin/spmv_lab 8 1118 "$V" > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" "$V" <<'PY'
import csv, glob, sys, collections
out, v = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/p*/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "spmv" in row["Kernel_Name"] or "stream" in row["Kernel_Name"]:
            agg[(row["Kernel_Name"].split("(")[0][-40:], row["Counter_Name"])].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for (k, c), vals in sorted(agg.items()):
        vals = vals[1:] if len(vals) > 1 else vals    # first dispatch = reference-result launch / warmup
        fh.write(f"{k:42s} {c:40s} n={len(vals):3d} mean={sum(vals)/len(vals):16.1f}\n")
print(open(out + "/summary.txt").read())
PY
