"""HBM-side traffic of one multigrid setup from the two --pmc passes of scripts/pmc_bench.sh (FETCH_SIZE, WRITE_SIZE per
kernel launch of `bench.py --steps 1 --warmup 0`): bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950 correction,
MI355X_MICROARCH.md), summed over the kernels a setup launches (everything that is not a kernel of the CG loop), per kernel
and in total.   python scripts/pmc_setup_sum.py gpurun_out/final/pmc [out.csv]"""
import collections, csv, glob, sys
out = sys.argv[1]
LOOP = ("csr_spmv_kernel<1,", "csr_spmv_kernel<2, float", "csr_spmv_kernel<6, float", "csr_spmv_kernel<7, float", "csr_spmv_kernel<3, float",
        "csr_spmv_kernel<4, float", "csr_spmv_wpr_kernel", "dense_gemv", "pcg_update", "pcg_init", "fold_partials", "residual_kernel",
        "pcg_set_tolerance", "csr_spmv_kernel<0, double", "mail_post", "kkt_", "halo_")
acc = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{C}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            per[row["Kernel_Name"].split("(")[0].replace("void ", "").replace("padne::", "")].append(float(row["Counter_Value"]))
    acc[C] = per
rows = []
for k, v in acc["FETCH_SIZE"].items():
    if any(t in k for t in LOOP):
        continue
    w = acc["WRITE_SIZE"].get(k, [0.0] * len(v))
    rows.append((2 * sum(v) * 1024 + sum(w) * 1024, k, len(v), sum(v), sum(w)))
rows.sort(reverse=True)
total = sum(r[0] for r in rows)
lines = ["kernel,launches_per_setup,FETCH_SIZE_KB_sum,WRITE_SIZE_KB_sum,bytes_2xFETCH_plus_WRITE"]
for b, k, n, f, w in rows:
    lines.append('"%s",%d,%.0f,%.0f,%.0f' % (k, n, f, w, b))
lines.append('"TOTAL (one setup)",%d,%.0f,%.0f,%.0f' % (sum(r[2] for r in rows), sum(r[3] for r in rows), sum(r[4] for r in rows), total))
text = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print(text if len(rows) < 15 else "\n".join(lines[:16] + lines[-1:]))
print("setup traffic: %.2f GB" % (total / 1e9))
