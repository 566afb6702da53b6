#!/bin/bash
# parity evidence of a round's FINAL build: the randomised sweeps, the per-fixture parity report, the exchange timing between
# processes, a one-stream kernel trace of the setup (standalone kernel times) -> gpurun_out/evidence/ (copied to profiles/<round>_*)
O=gpurun_out/evidence; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( echo "scripts/fuzz_parity.py 600 cases (seed 5), product path vs the reference's direct solve on the same KKT system:"; timeout -k 10 900 python scripts/fuzz_parity.py 600 5 2>&1 | tail -4
  echo; echo "scripts/fuzz_assembly.py 300 cases (seed 5), device assembly vs the oracle, bit for bit:"; timeout -k 10 600 python scripts/fuzz_assembly.py 300 5 2>&1 | tail -3 ) > $O/fuzz.txt 2>&1
echo fuzz done
timeout -k 10 300 python scripts/parity_report.py > $O/parity_report.log 2>&1 && cp gpurun_out/parity_report.json $O/; echo parity done
timeout -k 10 900 python scripts/exp_p2p_exchange.py > $O/p2p.log 2>&1 && cp gpurun_out/p2p_exchange.json $O/; echo p2p done
bash scripts/gpu_prof.sh evid_c4 "" --no-c5 --no-rank-proxy --no-small --no-dist-one-rank > $O/prof_c4.log 2>&1; cp gpurun_out/evid_c4_setup.txt gpurun_out/evid_c4_iteration.txt gpurun_out/evid_c4_streams.txt $O/ 2>/dev/null; echo profile done
