"""What the 8-wide product spends on its gathers (would x windows pay?): the product of config C3's
matrix against the same matrix with every column folded into [0, 64) -- same rows, same stream of values and indices, every
gather an L1 hit -- and against a copy whose columns are shuffled over the whole vector (every gather a miss).
    python scripts/lab/exp_spmm_gather_floor.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from padne_amd import _hip, synthetic
import bench

sysm = synthetic.config("C3")
ctx = _hip.Context(0)
nv = sysm.n_vertices
N = nv + 1
xy, tri, mvo, mto, sig = bench.flat(sysm)
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
L = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
A = L.reduce(imap, nv - 1, -1.0); L.close()
n = A.shape[0]
M = A.to_scipy()
rng = np.random.default_rng(1)
x8 = ctx.to_device(rng.uniform(-1, 1, n * 8)); y8 = ctx.empty(n * 8)
x1 = ctx.to_device(rng.uniform(-1, 1, n)); y1 = ctx.empty(n)
def t8(mat):
    return min(mat.spmm8_time(x8, y8, 5, 30) for _ in range(3))
print(f"as it is:            {t8(A)*1e6:7.1f} us  (SpMV {A.spmv_time(x1, y1, 5, 50)*1e6:.1f} us)", flush=True)
# (rows with repeated, unsorted columns are fine for a timing: uploaded through the C ABI as they are)
h = _hip._P()
ip, ix, dt = M.indptr.astype(np.int32), (M.indices % 64).astype(np.int32), M.data.astype(np.float64)
_hip._check(ctx._lib.padne_csr_from_host(ctx._h, n, n, _hip._ptr(ip, _hip._PI32), _hip._ptr(ix, _hip._PI32), _hip._ptr(dt, _hip._PF64), _hip.C.byref(h)))
dF = _hip.CsrMatrix(ctx, h)
print(f"columns mod 64:      {t8(dF)*1e6:7.1f} us  (every gather an L1 hit: the floor without gathers)", flush=True)
ix2 = rng.integers(0, n, len(ix)).astype(np.int32)
h2 = _hip._P()
_hip._check(ctx._lib.padne_csr_from_host(ctx._h, n, n, _hip._ptr(ip, _hip._PI32), _hip._ptr(ix2, _hip._PI32), _hip._ptr(dt, _hip._PF64), _hip.C.byref(h2)))
dR = _hip.CsrMatrix(ctx, h2)
print(f"columns at random:   {t8(dR)*1e6:7.1f} us  (every gather a miss)", flush=True)
print(f"algorithmic bytes {A.spmm8_bytes/1e9:.3f} GB: at 8 TB/s {A.spmm8_bytes/8e12*1e6:.0f} us")
