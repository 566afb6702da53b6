"""Debug: strongly graded Delaunay mesh, two layers with a conductivity jump."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.spatial, scipy.sparse as sp
from oracle import padne_oracle as O
from padne_amd import _hip
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
ex = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rng = np.random.default_rng(21)
u = rng.uniform(0, 1, (npts, 2))
rad = 30.0 * u[:, 0] ** ex                         # isotropic grading: dense around the origin
pts = np.column_stack([rad * np.cos(2 * np.pi * u[:, 1]), rad * np.sin(2 * np.pi * u[:, 1])])
pts = np.unique(np.round(pts, 9), axis=0)
tri = scipy.spatial.Delaunay(pts).simplices.astype(np.int32)
a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
cross = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
tri = tri[np.abs(cross) > 1e-12]; cross = cross[np.abs(cross) > 1e-12]
tri[cross < 0] = tri[cross < 0][:, [0, 2, 1]]
used = np.unique(tri); remap = -np.ones(len(pts), dtype=np.int64); remap[used] = np.arange(len(used))
xy, tri = pts[used], remap[tri].astype(np.int32)
n1 = len(xy)
ms = [(xy, tri, 2082.5), (xy.copy(), tri.copy(), 52.0)]
ties = np.random.default_rng(5).choice(n1, 25, replace=False)
els = [("R", int(t), int(n1 + t), 2e-3) for t in ties]
src, snk = int(np.argmin(xy.sum(axis=1))), int(n1 + np.argmax(xy.sum(axis=1)))
els += [("I", src, snk, 3.0)]
Lo, ro = O.assemble_system(ms, 0, els, 11)
n = 2 * n1
keep = np.array([i for i in range(n) if i != 11])
A = (-Lo[keep][:, keep]).tocsr(); A.sort_indices()
b = -ro[keep]
d = A.diagonal()
print("n", A.shape[0], "diag range", d.min(), d.max(), "offdiag positive entries:", int((A - sp.diags(d)).max() > 0), flush=True)
rowsum = np.asarray(A.sum(axis=1)).ravel()
print("row sums: min", rowsum.min(), "max", rowsum.max(), flush=True)
ctx = _hip.Context(0)
dA = ctx.csr_from_scipy(A)
for pc in ("amg", "jacobi"):
    try:
        res = dA.solve_spd(b, precond=pc, raise_on_fail=False, max_iter=60000)
        print(pc, "iterations", res.iterations, "levels", res.levels, "fallbacks", res.precond_fallbacks, "relres", res.rel_residual,
              "status", res.status, flush=True)
    except Exception as e:
        print(pc, "failed:", e, flush=True)
