"""HBM-side traffic of one timed step of bench.py (multigrid setup + CG iterations) from the two --pmc passes of
scripts/pmc_bench.sh (FETCH_SIZE, WRITE_SIZE per kernel launch of `bench.py --steps 1 --warmup 0 --no-seam --no-launch-count`
and none of the side blocks):
bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950 correction, MI355X_MICROARCH.md, calibrated in spmv_traffic.json).
That command runs TWO solves (the timed step and the one that samples the SpMV with events): sums are halved.  Kernels of
the workload build (mesh generation, assembly, reduction) are left out.
    python scripts/pmc_setup_sum.py gpurun_out/final/pmc [per_kernel.csv] [step_traffic.json]"""
import collections, csv, glob, json, sys
out = sys.argv[1]
LOOP = ("csr_spmv_kernel<1,", "csr_spmv_kernel<2, float", "csr_spmv_kernel<6, float", "csr_spmv_kernel<7, float", "csr_spmv_kernel<3, float",
        "csr_spmv_kernel<4, float", "csr_spmv_kernel<0, float", "csr_spmv_kernel<8, float", "p_hat_from_z", "csr_spmv_wpr_kernel", "dense_gemv", "pcg_update_xr_entry", "pcg_update_p_z", "pcg_x_flush",
        "pcg_init", "fold_partials", "residual_kernel", "pcg_set_tolerance", "csr_spmv_kernel<0, double", "mail_post")
BUILD = ("asm_", "grid_mesh", "generate", "relabel", "reduce_", "map_is_injective", "map_is_compaction", "merge_rows", "compact_rows", "nn_", "kkt_", "halo_",
         "sort_long_rows_wave<1024, 4>", "fill_value_i32")
SOLVES = 2
acc = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{C}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            per[row["Kernel_Name"].split("(")[0].replace("void ", "").replace("padne::", "")].append(float(row["Counter_Value"]))
    acc[C] = per
groups = {"setup": [], "loop": [], "build": []}
for k, v in acc["FETCH_SIZE"].items():
    w = acc["WRITE_SIZE"].get(k, [0.0] * len(v))
    if "csr_spmv_kernel<0, double" in k and len(v) > 8 * SOLVES:
        # the plain product: once per solve for the true residual -- and 55 back-to-back launches of bench.py's standalone
        # timing behind the solves, which are no part of a step: keep one launch per solve
        big = sorted(v)[len(v) // 2]
        v = [big] * SOLVES
        w = [sorted(w)[len(w) // 2]] * SOLVES
    if "copyBuffer" in k:
        # device-to-device copies: the small ones belong to the setup, the 160 / 240 MB ones are the mesh copy an assembled
        # matrix keeps (bench.py times the assembly eleven times before the solves): workload build, not part of a step
        keep = [i for i in range(len(v)) if v[i] < 50000.0]
        v = [v[i] for i in keep]
        w = [w[i] for i in keep if i < len(w)]
    rec = (2 * sum(v) * 1024 + sum(w) * 1024, k, len(v), sum(v), sum(w))
    g = "loop" if any(t in k for t in LOOP) else ("build" if any(t in k for t in BUILD) else "setup")
    # compact_rows / merge kernels also run inside the setup (products of the coarse levels): only their launches on
    # 10 M-row inputs belong to the assembly -- they cannot be told apart by name, so they are all counted with the setup
    if g == "build" and k.startswith(("compact_rows", "merge_rows")):
        g = "setup"
    groups[g].append(rec)
for g in groups:
    groups[g].sort(reverse=True)
tot = {g: sum(r[0] for r in rows) / SOLVES for g, rows in groups.items()}
lines = ["group,kernel,launches_per_solve,FETCH_SIZE_KB_per_solve,WRITE_SIZE_KB_per_solve,bytes_per_solve_2xFETCH_plus_WRITE"]
for g in ("setup", "loop"):
    for b, k, n, f, w in groups[g]:
        lines.append('%s,"%s",%.1f,%.0f,%.0f,%.0f' % (g, k, n / SOLVES, f / SOLVES, w / SOLVES, b / SOLVES))
    lines.append('%s,"TOTAL",%.1f,%.0f,%.0f,%.0f' % (g, sum(r[2] for r in groups[g]) / SOLVES, sum(r[3] for r in groups[g]) / SOLVES,
                                                   sum(r[4] for r in groups[g]) / SOLVES, tot[g]))
text = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
rec = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-seam` "
                 "(two solves: sums halved); bytes = 2 * FETCH_SIZE + WRITE_SIZE (KB -> B), gfx950 correction as in spmv_traffic.json",
       "workload": "C4", "setup_bytes": tot["setup"], "loop_bytes": tot["loop"], "step_bytes": tot["setup"] + tot["loop"],
       "largest_setup_kernels_GB": {r[1]: round(r[0] / SOLVES / 1e9, 2) for r in groups["setup"][:12]}}
if len(sys.argv) > 3:
    json.dump(rec, open(sys.argv[3], "w"), indent=1)
print("\n".join(lines[:14]))
print("setup %.2f GB  loop %.2f GB  step %.2f GB per solve" % (tot["setup"] / 1e9, tot["loop"] / 1e9, (tot["setup"] + tot["loop"]) / 1e9))
