#!/bin/bash
# The round's measurement suite on one MI355X; everything lands in gpurun_out/final/ (copy what is judged into profiles/)
O=gpurun_out/final; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 600 python bench.py > $O/bench_default.log 2>&1 || exit 1
grep '^{' $O/bench_default.log > $O/bench_c4_1gpu_amg.json; echo bench done
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-seam --no-c5 --no-rank-proxy --no-small --no-dist-one-rank > $GRAFT_REPO_ROOT/$O/stats.log 2>&1 ) || exit 2
echo stats done
timeout -k 10 600 bash scripts/pmc_bench.sh $O/pmc > $O/pmc.log 2>&1 || exit 3
python scripts/pmc_setup_sum.py $O/pmc $O/pmc_per_kernel_per_solve.csv $O/step_traffic.json > $O/pmc_sum.log 2>&1; tail -2 $O/pmc_sum.log
echo pmc done
timeout -k 10 600 python scripts/run_configs.py > $O/configs.log 2>&1 || exit 4
cp gpurun_out/configs.json $O/ ; echo configs done
timeout -k 10 300 python bench.py --precond jacobi --steps 1 --warmup 0 --no-cpu-baseline --no-c5 --no-rank-proxy --no-small --no-dist-one-rank > $O/bench_jacobi.log 2>&1 || exit 5
grep '^{' $O/bench_jacobi.log > $O/bench_c4_1gpu_jacobi.json; echo jacobi done
for c in C2 C3 C4; do bash scripts/asm_prof.sh $c > $O/asm_$c.log 2>&1 || exit 6; cp gpurun_out/asm_${c}_timeline.txt $O/; done
timeout -k 10 400 bash scripts/pmc_asm.sh $O/pmc_asm C4 "FETCH_SIZE" "WRITE_SIZE" > $O/asm_C4_pmc.txt 2>&1 || exit 7
echo assembly done
