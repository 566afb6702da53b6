#!/bin/bash
# Roofline evidence on CGAL-shaped input: scripts/unstructured_roofline.py once for the times, then its --spmv-only form under
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes (counter traffic of the f64 product per launch, x2 on the
# fetches as everywhere on gfx950).  usage (GPU box, repo root):  bash scripts/pmc_unstructured.sh gpurun_out/r04_unstructured [side]
OUT="$GRAFT_REPO_ROOT/$1"; SIDE=${2:-1620}; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
PADNE_VERBOSE=xw timeout -k 10 900 python3 $GRAFT_REPO_ROOT/scripts/unstructured_roofline.py --side $SIDE > "$OUT/run.log" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
echo "timing run done"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o run -- python3 $GRAFT_REPO_ROOT/scripts/unstructured_roofline.py --side $SIDE --spmv-only > "$OUT/$C.log" 2>&1 || exit 2
  echo "$C done"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections, re
out = sys.argv[1]
rec = json.loads([l for l in open(out + "/run.log") if l.startswith("{")][-1])
plans = re.findall(r"x-window plan: (\d+) of (\d+) tiles qualify with runs of (\d+)", open(out + "/run.err").read())
rec["x_window_plans"] = [{"tiles_on_the_window_path": int(a), "tiles": int(b), "run": int(c), "share": int(a) / int(b)} for a, b, c in plans]
acc = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{C}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            per[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    acc[C] = per
name = [k for k in acc["FETCH_SIZE"] if "csr_spmv_kernel<0, double, double, double" in k or "csr_spmv_kernel<1, double, double, double" in k]
if name:
    k = name[0]
    f = acc["FETCH_SIZE"][k]; w = acc["WRITE_SIZE"][k]
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    s = rec["strip_numbering"]
    s["counter"] = {"kernel": k, "launches": len(f), "FETCH_SIZE_KB": fm, "WRITE_SIZE_KB": wm,
                    "bytes_per_launch": 2 * fm * 1024 + wm * 1024,
                    "ratio_to_algorithmic": (2 * fm * 1024 + wm * 1024) / s["spmv_bytes_algorithmic"],
                    "gbs_standalone_by_counter_bytes": (2 * fm * 1024 + wm * 1024) / (s["spmv_us_standalone"] * 1e-6) / 1e9,
                    "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section), WRITE_SIZE exact; separate --pmc passes"}
json.dump(rec, open(out + "/unstructured.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
