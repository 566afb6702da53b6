"""CPU (scipy) experiment: iteration counts of the device's smoothed-aggregation recipe with more aggressive
coarsening on selected levels (fewer / smaller coarse levels = less launch-bound work in setup and cycle).
Recipe as csrc/amg.hip: theta = 0.08 strength, MIS-2 aggregates, filtered prolongator smoothing with
omega = 1.5 / Gershgorin(filtered), V(1,1) damped Jacobi c = 1 / (0.55 lambda), exact coarsest solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import amg_proto as P
from oracle import padne_oracle as O
from padne_amd import synthetic as S

TH = 0.08
def strength_graph(A, theta=TH):
    d = A.diagonal(); C = A.tocoo()
    keep = (np.abs(C.data) >= theta * np.sqrt(d[C.row] * d[C.col])) | (C.row == C.col)
    return sp.csr_matrix((C.data[keep], (C.row[keep], C.col[keep])), shape=A.shape)
def filtered(A, theta=TH):
    d = A.diagonal(); C = A.tocoo()
    strong = (np.abs(C.data) >= theta * np.sqrt(d[C.row] * d[C.col])) & (C.row != C.col)
    weak = ~strong & (C.row != C.col)
    lump = np.bincount(C.row[weak], weights=C.data[weak], minlength=A.shape[0])
    return (sp.csr_matrix((C.data[strong], (C.row[strong], C.col[strong])), shape=A.shape) + sp.diags(d + lump)).tocsr()
def gersh(A): return (abs(A).sum(axis=1).A1 / A.diagonal()).max()
def lam_max(A):
    Dm = sp.diags(1.0 / np.sqrt(A.diagonal()))
    return spla.eigsh(Dm @ A @ Dm, k=1, which="LA", return_eigenvectors=False, tol=1e-3)[0]

def aggregate(A, passes):
    Sg = strength_graph(A)
    agg, nc, _ = P.mis2_aggregate(Sg)
    for _ in range(passes - 1):
        n = A.shape[0]
        T = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nc))
        G = (T.T @ abs(Sg) @ T).tocsr()          # graph of the aggregates (unsmoothed Galerkin pattern)
        G = (G - sp.diags(G.diagonal())) * -1.0 + sp.diags(np.ones(nc))
        agg2, nc2, _ = P.mis2_aggregate(G.tocsr())
        agg, nc = agg2[agg], nc2
    return agg, nc

def build(A, passes_by_level, coarse_n=2048, psmooth=1):
    levels = []; lvl = 0
    while A.shape[0] > coarse_n:
        passes = passes_by_level[min(lvl, len(passes_by_level) - 1)]
        agg, nc = aggregate(A, passes)
        n = A.shape[0]
        T = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nc))
        AF = filtered(A); dF = AF.diagonal()
        lamF = min(gersh(AF), gersh(A))
        Pm = T
        for _ in range(psmooth if passes == 1 else psmooth):
            Pm = (Pm - (1.5 / lamF) * (sp.diags(1.0 / dF) @ (AF @ Pm))).tocsr()
        lam = gersh(A) if lvl == 0 else 1.08 * lam_max(A)
        levels.append((A, Pm, lam))
        print(f"   level {lvl}: n={n} nnz={A.nnz} -> {nc} (x{n/nc:.1f}) nnz(P)={Pm.nnz}", flush=True)
        A = (Pm.T @ A @ Pm).tocsr(); lvl += 1
    levels.append((A, None, 2.0))
    print(f"   coarsest n={A.shape[0]} nnz={A.nnz}; operator complexity {sum(l[0].nnz for l in levels)/levels[0][0].nnz:.3f}")
    return levels

def make_vc(levels):
    lu = spla.splu(levels[-1][0].tocsc())
    def vc(b, l=0):
        Al, Pm, lam = levels[l]
        if Pm is None: return lu.solve(b)
        dinv = 1.0 / Al.diagonal(); c = 1.0 / (0.55 * lam)
        x = c * dinv * b
        x = x + Pm @ vc(Pm.T @ (b - Al @ x), l + 1)
        return x + c * dinv * (b - Al @ x)
    return vc

if __name__ == "__main__":
    nl, nx = int(sys.argv[1]), int(sys.argv[2])
    sysm = S.layered_system(nl, nx, nx, via_lattice=max(2, 32 * nx // 1118))
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)] + [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    L, r = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    n = sysm.n_vertices
    A = (-L[1:n, 1:n]).tocsr(); b = -r[1:n]
    for name, passes, cn in (("baseline", [1], 2048), ("double from level 1", [1, 2], 2048), ("double from level 2", [1, 1, 2], 2048),
                             ("double everywhere", [2], 2048)):
        print(name, flush=True)
        t = time.time(); lv = build(A, passes, cn)
        x, it = P.pcg(A, b, make_vc(lv))
        print(f"   => PCG iterations {it} (setup+solve {time.time()-t:.0f} s)", flush=True)
