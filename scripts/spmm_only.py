"""Standalone timing of the 8-vector SpMM against 8 SpMVs on one of the synthetic configs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from padne_amd import _hip, synthetic
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
sysm = synthetic.config(name)
ctx = _hip.Context(0)
nv = sysm.n_vertices
N = nv + 1
xy, tri, mvo, mto, sig = bench.flat(sysm)
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
L = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
A = L.reduce(imap, nv - 1, -1.0); L.close()
n = A.shape[0]
rng = np.random.default_rng(1)
x1 = ctx.to_device(rng.uniform(-1, 1, n)); y1 = ctx.empty(n)
x8 = ctx.to_device(rng.uniform(-1, 1, n * 8)); y8 = ctx.empty(n * 8)
t1 = A.spmv_time(x1, y1, 5, 50)
t8 = A.spmm8_time(x8, y8, 5, 30)
print(f"[{name}] n={n} nnz={A.nnz}: SpMV {t1*1e6:.1f} us ({A.spmv_bytes/t1/1e9:.0f} GB/s, {A.spmv_bytes/t1/8e12:.1%} of 8 TB/s); "
      f"SpMM8 {t8*1e6:.1f} us = {t8/t1:.2f} SpMV ({A.spmm8_bytes/t8/1e9:.0f} GB/s algorithmic, {A.spmm8_bytes/t8/8e12:.1%}); "
      f"8 vectors {8*t1/t8:.2f}x faster than 8 SpMVs", flush=True)
