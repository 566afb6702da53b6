"""How many runs of L consecutive columns cover the columns of a 64-row tile of the hierarchy's gather-path operators
(restriction, first coarse operator, second restriction)?  python scripts/exp_runs.py [layers nx ny]  (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from padne_amd import _hip, synthetic
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import padne_oracle as O

nl, nx, ny = ([int(a) for a in sys.argv[1:4]] + [2, 700, 700])[:3] if len(sys.argv) > 3 else (2, 700, 700)
sysm = synthetic.layered_system(nl, nx, ny, via_lattice=6)
els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
n = sysm.n_vertices
A = (-Lo[1:n, 1:n]).tocsr(); A.sort_indices()
b = -ro[1:n]
ctx = _hip.Context(0)
d = ctx.csr_from_scipy(A)
res = d.solve_spd(b, precond="amg")
print("levels", res.levels, "iterations", res.iterations, flush=True)

def runs_needed(M, L):
    M = M.tocsr(); M.sort_indices()
    nt = (M.shape[0] + 63) // 64
    out = np.zeros(nt, dtype=np.int32)
    for t in range(nt):
        c = np.unique(M.indices[M.indptr[64 * t]:M.indptr[min(64 * t + 64, M.shape[0])]])
        k = 0; bound = -1
        for v in c:
            if v > bound:
                k += 1; bound = v + L - 1
        out[t] = k
    return out

for lvl in range(min(res.levels - 1, 2)):
    ops = {"A%d" % lvl: d.amg_level(lvl, "A"), "R%d" % lvl: d.amg_level(lvl, "R")}
    for name, M in ops.items():
        if lvl == 0 and name == "A0": continue
        print(name, M.shape, "nnz/row %.1f" % (M.nnz / M.shape[0]))
        for L in (32, 64, 128, 256, 512):
            r = runs_needed(M, L)
            q = np.percentile(r, [50, 90, 95, 99])
            print("   runs of %4d: median %d  p90 %d  p95 %d  p99 %d  max %d   positions at p95 %d" % (L, q[0], q[1], q[2], q[3], r.max(), int(q[2]) * L))
