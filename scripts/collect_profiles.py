"""Copy the judged summaries of scripts/measure_all.sh (gpurun_out/final/) into profiles/ and derive the per-kernel PMC
table and the SpMV-by-grid summary from the raw csv files."""
import csv, glob, json, collections, shutil, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RD = os.environ.get("PADNE_ROUND", "r05")        # prefix of the files written under profiles/
F = os.path.join(R, "gpurun_out", "final"); P = os.path.join(R, "profiles")
for src, dst in (("bench_c4_1gpu_amg.json", RD + "_bench_c4_1gpu_amg.json"), ("bench_c4_1gpu_amg.json", RD + "_bench_c4_1gpu.json"),
                 ("bench_c4_1gpu_jacobi.json", RD + "_bench_c4_1gpu_jacobi.json"), ("stats/run_kernel_stats.csv", RD + "_bench_c4_amg_kernel_stats.csv"),
                 ("pmc/spmv_traffic.json", "spmv_traffic.json"), ("configs.json", RD + "_configs.json"),
                 ("step_traffic.json", RD + "_step_traffic.json"), ("pmc_per_kernel_per_solve.csv", RD + "_pmc_per_kernel_per_solve.csv")):
    shutil.copy(os.path.join(F, src), os.path.join(P, dst))
with open(os.path.join(P, RD + "_assembly_timeline.txt"), "w") as f:
    f.write("Kernel timeline of the device-resident assembly (scripts/asm_prof.sh: rocprofv3 --kernel-trace of scripts/asm_only.py), last of four runs\n")
    for c in ("C2", "C3", "C4"):
        f.write(f"\n== config {c} ==\n" + open(os.path.join(F, f"asm_{c}_timeline.txt")).read())
    f.write("\n== config C4: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB, separate passes, mean of four runs; scripts/pmc_asm.sh) ==\n")
    f.write(open(os.path.join(F, "asm_C4_pmc.txt")).read())
# counter traffic of one assembly (the bench line's assembly.traffic): 2 x FETCH_SIZE + WRITE_SIZE over the assembly kernels
try:
    rows_a = [l.split() for l in open(os.path.join(F, "asm_C4_pmc.txt")) if l.startswith(("asm_", "merge_rows"))]
    tot = sum(2 * float(r[-2]) * 1024 + float(r[-1]) * 1024 for r in rows_a)
    json.dump({"workload": "C4", "bytes_per_assembly": tot, "kernels": {" ".join(r[:-2]): {"FETCH_SIZE_KB": float(r[-2]), "WRITE_SIZE_KB": float(r[-1])} for r in rows_a},
               "source": "scripts/pmc_asm.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over scripts/asm_only.py C4, mean of four assemblies; bytes = 2 * FETCH_SIZE + WRITE_SIZE (gfx950)",
               "algorithmic_bytes": 1239009368}, open(os.path.join(P, RD + "_assembly_traffic.json"), "w"), indent=1)
except Exception as exc:
    print("assembly traffic not written:", exc)
acc = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for f in glob.glob(f"{F}/pmc/{C}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            per[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    acc[C] = per
rows = []
for k, v in acc["FETCH_SIZE"].items():
    w = acc["WRITE_SIZE"].get(k, [0.0])
    rows.append((sum(v), k, len(v), sum(v) / len(v), max(v), sum(w) / len(w), max(w)))
rows.sort(reverse=True)
with open(os.path.join(P, RD + "_pmc_fetch_write_per_kernel.csv"), "w") as f:
    f.write("kernel,launches,FETCH_SIZE_KB_mean,FETCH_SIZE_KB_max,WRITE_SIZE_KB_mean,WRITE_SIZE_KB_max,note\n")
    for _, k, n, fm, fx, wm, wx in rows[:40]:
        f.write('"%s",%d,%.1f,%.1f,%.1f,%.1f,separate --pmc passes over `bench.py --steps 1 --warmup 0`; HBM bytes = 2*FETCH*1024 + WRITE*1024 (gfx950)\n' % (k, n, fm, fx, wm, wx))
per = collections.defaultdict(list)
for row in csv.DictReader(open(f"{F}/stats/run_kernel_trace.csv")):
    if "csr_spmv_kernel<1, double, double, double" in row["Kernel_Name"]:
        per[int(row.get("Grid_Size_X", row.get("Grid_Size", 0)))].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
st = [r for r in csv.DictReader(open(f"{F}/stats/run_kernel_stats.csv")) if "csr_spmv_kernel<1, double, double, double" in r["Name"]][0]
rec = {"source": "rocprofv3 --kernel-trace --stats --output-format csv of `python bench.py --no-cpu-baseline` (profiles/" + RD + "_bench_c4_amg_kernel_stats.csv is the --stats table of the same run)",
       "kernel": "padne::csr_spmv_kernel<SPMV_DOT = 1, double, double, double, ..., float>  (the CG loop's q = A p with p stored in single precision; the Lanczos estimates of the setup use the twin instantiation <5, ...>)",
       "stats": {"calls": int(st["Calls"]), "average_us": float(st["AverageNs"]) / 1e3, "min_us": float(st["MinNs"]) / 1e3, "max_us": float(st["MaxNs"]) / 1e3},
       "by_grid": {str(g): {"launches": len(v), "mean_us": sum(v) / len(v)} for g, v in per.items()},
       "algorithmic_bytes_per_launch": 999452960,      # 12 nnz + 16 n + 4: p is stored in single precision since the second half of round 4
       "achieved_GBs_from_stats_average": 999452960 / (float(st["AverageNs"]) * 1e-9) / 1e9}
json.dump(rec, open(os.path.join(P, RD + "_spmv_dot_by_level.json"), "w"), indent=1)
print(json.dumps(rec["stats"]), rec["achieved_GBs_from_stats_average"])
