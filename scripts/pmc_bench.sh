#!/bin/bash
# HBM traffic of the dominant kernel per launch: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes over one
# bench step (MI355X_MICROARCH.md, HBM section: FETCH_SIZE x2 on gfx950, calibrated here on a pure streaming kernel).
# usage (on the GPU box, from the repo root):  bash scripts/pmc_bench.sh gpurun_out/pmc_r01
OUT="$GRAFT_REPO_ROOT/$1"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-seam --no-c5 --no-rank-proxy --no-small --no-dist-one-rank --no-launch-count > "$OUT/$C.log" 2>&1 || exit 1
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{C}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            per[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    acc[C] = per
def big(name, C):      # mean over the launches on the finest level (the largest values)
    v = sorted(acc[C].get(name, []))
    v = [x for x in v if x > 0.5 * v[-1]] if v else []
    return sum(v) / len(v) if v else float("nan"), len(v)
names = [k for k in acc["FETCH_SIZE"] if "csr_spmv_kernel<1, double, double, double" in k]
dot = names[0]
# calibration kernel: the p update of the multigrid loop (pcg.hip, pcg_update_p_z_kernel; since round 6 x is formed from the
# kept search directions, so the update touches no x): per row it READS p^ (4 B) and z32 (4 B) = 8 B and WRITES the next p^
# (4 B, a place of its own), nothing else
cal = [k for k in acc["FETCH_SIZE"] if "pcg_update_p_z_kernel" in k][0]
f_dot, n_dot = big(dot, "FETCH_SIZE"); w_dot, _ = big(dot, "WRITE_SIZE")
f_cal, n_cal = big(cal, "FETCH_SIZE"); w_cal, _ = big(cal, "WRITE_SIZE")
rec = {"kernel": dot, "workload": "C4 (N=10M, nnz=70M)", "round": 6, "launches_averaged": n_dot,
       "FETCH_SIZE_KB": f_dot, "WRITE_SIZE_KB": w_dot,
       "bytes_per_launch": 2 * f_dot * 1024 + w_dot * 1024,
       "algorithmic_bytes_per_launch": 999452960,
       "calibration": {"kernel": cal, "launches_averaged": n_cal, "FETCH_SIZE_KB": f_cal, "WRITE_SIZE_KB": w_cal,
                       "bytes_per_row_read": 8, "bytes_per_row_written": 4,
                       "expected_read_bytes": 8 * 9999391, "expected_write_bytes": 4 * 9999391,
                       "read_ratio_with_x2": 2 * f_cal * 1024 / (8 * 9999391), "write_ratio": w_cal * 1024 / (4 * 9999391)},
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section) -> x2, checked on the "
                     "calibration kernel of the same run whose traffic is known exactly; WRITE_SIZE exact"}
json.dump(rec, open(out + "/spmv_traffic.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
