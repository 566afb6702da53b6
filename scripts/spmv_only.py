"""Profiling target: assemble config C4 (or --workload) and launch the SpMV kernel a few times.

Used under rocprofv3 for the per-kernel trace and the PMC (FETCH_SIZE / WRITE_SIZE) passes:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python scripts/spmv_only.py
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from padne_amd import _hip, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C4")
ap.add_argument("--launches", type=int, default=20)
ap.add_argument("--iters", type=int, default=0, help="also run this many PCG iterations")
args = ap.parse_args()

ctx = _hip.Context(0)
sysm = synthetic.config(args.workload)
nv = sysm.n_vertices
N = nv + 1
xy, tri, mvo, mto, sig = bench.flat(sysm)
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
L = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
imap = np.arange(N, dtype=np.int32)
imap[sysm.ground] = -1
imap[imap > sysm.ground] -= 1
imap[N - 1] = -1
A = L.reduce(imap, nv - 1, -1.0)
L.close()
x = ctx.to_device(np.random.default_rng(1).uniform(-1, 1, A.shape[1]))
y = ctx.empty(A.shape[0])
t = A.spmv_time(x, y, 3, args.launches)
print(f"spmv {t*1e6:.1f} us/launch  {A.spmv_bytes/t/1e9:.1f} GB/s  bytes={A.spmv_bytes} rows={A.shape[0]} nnz={A.nnz}")
if args.iters:
    keep = np.flatnonzero(imap[:nv] >= 0)
    b = ctx.to_device(-rhs[keep])
    xs = ctx.empty(A.shape[0])
    r = A.solve_spd_dev(b, xs, max_iter=args.iters, raise_on_fail=False)
    print(f"pcg {r.iterations} iterations, {r.seconds/max(r.iterations,1)*1e6:.1f} us/iteration")
