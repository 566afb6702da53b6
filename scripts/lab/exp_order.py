"""Experiment: SpMV time with natural (row-major) vs Morton-ordered unknowns (C4 size)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from padne_amd import _hip, synthetic

def morton_keys(xy, bits=16):
    mn = xy.min(axis=0); span = np.maximum(xy.max(axis=0) - mn, 1e-300)
    q = np.minimum(((xy - mn) / span * (2**bits - 1)).astype(np.uint64), 2**bits - 1)
    def spread(v):
        v = (v | (v << 16)) & np.uint64(0x0000FFFF0000FFFF)
        v = (v | (v << 8)) & np.uint64(0x00FF00FF00FF00FF)
        v = (v | (v << 4)) & np.uint64(0x0F0F0F0F0F0F0F0F)
        v = (v | (v << 2)) & np.uint64(0x3333333333333333)
        v = (v | (v << 1)) & np.uint64(0x5555555555555555)
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1))

ctx = _hip.Context(0)
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
sysm = synthetic.config(name); nv = sysm.n_vertices; N = nv + 1
xy, tri, mvo, mto, sig = bench.flat(sysm); rows, cols, vals, rhs = bench.stamps_of(sysm, N)
L = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
A = L.reduce(imap, nv - 1, -1.0)
x = ctx.to_device(np.random.default_rng(1).uniform(-1, 1, A.shape[1])); y = ctx.empty(A.shape[0])
for _ in range(2):
    t = A.spmv_time(x, y, 5, 50); print(f"natural : {t*1e6:.1f} us {A.spmv_bytes/t/1e9:.0f} GB/s", flush=True)
t0 = time.time()
mesh_id = np.repeat(np.arange(len(sysm.meshes)), np.diff(mvo)).astype(np.uint64)
key = (mesh_id << np.uint64(40)) | morton_keys(xy)
order = np.argsort(key, kind="stable")            # new position -> old vertex
newpos = np.empty(nv, dtype=np.int64); newpos[order] = np.arange(nv)
print("host morton+argsort", time.time() - t0, flush=True)
# compose: vertex g -> position among non-ground vertices in morton order
pm = np.full(N, -1, dtype=np.int32)
pos = newpos.copy(); gpos = pos[sysm.ground]; pos[pos > gpos] -= 1
pm[:nv] = pos; pm[sysm.ground] = -1
t0 = time.time(); A2 = L.reduce(pm, nv - 1, -1.0); ctx.synchronize(); print("reduce(morton)", time.time() - t0, flush=True)
for _ in range(2):
    t = A2.spmv_time(x, y, 5, 50); print(f"morton  : {t*1e6:.1f} us {A2.spmv_bytes/t/1e9:.0f} GB/s", flush=True)
keep = np.flatnonzero(pm[:nv] >= 0)
b = np.zeros(nv - 1); b[pm[keep]] = -rhs[keep]
bd = ctx.to_device(b); xs = ctx.empty(nv - 1)
r = A2.solve_spd_dev(bd, xs, rtol=1e-12, time_spmv=True)
print(f"pcg morton: iters={r.iterations} {r.seconds:.3f}s us/iter={r.seconds/r.iterations*1e6:.1f} spmv_in_situ={r.spmv_seconds*1e6:.1f}us relres={r.rel_residual:.2e}", flush=True)
