"""CPU (scipy) experiment: smoothed aggregation on the fine level only, plain (unsmoothed) aggregation below -- the coarse
levels' A P, R = P^T and R (A P) would then be index bookkeeping instead of sparse products.  python scripts/exp_unsmoothed_coarse.py 8 220"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, scipy.sparse as sp
import amg_proto as P
import exp_aggressive as E
from oracle import padne_oracle as O
from padne_amd import synthetic as S


def build(A, smooth_levels, coarse_n=2048, overcorrect=1.0):
    levels = []; lvl = 0
    while A.shape[0] > coarse_n:
        agg, nc = E.aggregate(A, 1)
        n = A.shape[0]
        T = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nc))
        Pm = T
        if lvl < smooth_levels:
            AF = E.filtered(A); dF = AF.diagonal()
            lamF = min(E.gersh(AF), E.gersh(A))
            Pm = (Pm - (1.5 / lamF) * (sp.diags(1.0 / dF) @ (AF @ Pm))).tocsr()
        lam = E.gersh(A) if lvl == 0 else 1.08 * E.lam_max(A)
        levels.append((A, Pm, lam, 1.0 if lvl < smooth_levels else overcorrect))
        A = (Pm.T @ A @ Pm).tocsr(); lvl += 1
    levels.append((A, None, 2.0, 1.0))
    print(f"   sizes {[l[0].shape[0] for l in levels]} nnz {[l[0].nnz for l in levels]} complexity {sum(l[0].nnz for l in levels)/levels[0][0].nnz:.3f}")
    return levels


def make_vc(levels, nu_coarse=1):
    import scipy.sparse.linalg as spla
    lu = spla.splu(levels[-1][0].tocsc())
    def vc(b, l=0):
        Al, Pm, lam, oc = levels[l]
        if Pm is None: return lu.solve(b)
        dinv = 1.0 / Al.diagonal(); c = 1.0 / (0.55 * lam)
        nu = 1 if l == 0 else nu_coarse
        x = c * dinv * b
        for _ in range(nu - 1): x = x + c * dinv * (b - Al @ x)
        x = x + oc * (Pm @ vc(Pm.T @ (b - Al @ x), l + 1))
        for _ in range(nu): x = x + c * dinv * (b - Al @ x)
        return x
    return vc


if __name__ == "__main__":
    nl, nx = int(sys.argv[1]), int(sys.argv[2])
    sysm = S.layered_system(nl, nx, nx, via_lattice=max(2, 32 * nx // 1118))
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)] + [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    L, r = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    n = sysm.n_vertices
    A = (-L[1:n, 1:n]).tocsr(); b = -r[1:n]
    for name, sl, oc, nu in (("smoothed everywhere (shipped)", 99, 1.0, 1), ("smoothed on level 0 only", 1, 1.0, 1),
                             ("level 0 only, over-correction 1.5", 1, 1.5, 1), ("level 0 only, V(2,2) below", 1, 1.0, 2),
                             ("levels 0-1 smoothed", 2, 1.0, 1)):
        print(name, flush=True)
        lv = build(A, sl, overcorrect=oc)
        x, it = P.pcg(A, b, make_vc(lv, nu))
        print(f"   => PCG iterations {it}", flush=True)
