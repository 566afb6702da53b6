"""Compare the device V-cycle with the scipy prototype on the same matrix (debug)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import amg_proto as P
from oracle import padne_oracle as O
from padne_amd import _hip, synthetic as S

def strength_graph(A, theta):
    d = A.diagonal(); C = A.tocoo()
    keep = (np.abs(C.data) >= theta * np.sqrt(d[C.row] * d[C.col])) | (C.row == C.col)
    return sp.csr_matrix((C.data[keep], (C.row[keep], C.col[keep])), shape=A.shape)
def filtered(A, theta):
    d = A.diagonal(); C = A.tocoo()
    strong = (np.abs(C.data) >= theta * np.sqrt(d[C.row] * d[C.col])) & (C.row != C.col)
    weak = ~strong & (C.row != C.col)
    lump = np.bincount(C.row[weak], weights=C.data[weak], minlength=A.shape[0])
    return (sp.csr_matrix((C.data[strong], (C.row[strong], C.col[strong])), shape=A.shape) + sp.diags(d + lump)).tocsr()
def gersh(A): return (abs(A).sum(axis=1).A1 / A.diagonal()).max()

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 120
sysm = S.layered_system(1, nx, nx, via_lattice=2)
els = [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
L, r = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
n = sysm.n_vertices
A = (-L[1:n, 1:n]).tocsr(); A.sort_indices()
ctx = _hip.Context(0)
Ad = ctx.csr_from_scipy(A)
rng = np.random.default_rng(0); rv = rng.uniform(-1, 1, A.shape[0])
zd = Ad.amg_apply(rv)
# prototype hierarchy, same recipe as the device (filtered smoothing, Gershgorin lambdas, coarse_n = 512)
levels = []; Al = A; lvl = 0
while Al.shape[0] > 512:
    agg, nc, _ = P.mis2_aggregate(strength_graph(Al, 0.1))
    nl = Al.shape[0]
    T = sp.csr_matrix((np.ones(nl), (np.arange(nl), agg)), shape=(nl, nc))
    AF = filtered(Al, 0.1); dF = AF.diagonal()
    lam = gersh(Al)
    lamF = min((abs(AF).sum(axis=1).A1 / dF).max(), lam)
    Pm = (T - (4.0 / 3.0 / lamF) * (sp.diags(1.0 / dF) @ (AF @ T))).tocsr()
    levels.append((Al, Pm, lam)); print(f"proto level {lvl}: n={nl} -> {nc}  nnz(P)={Pm.nnz} lam={lam:.3f}")
    Al = (Pm.T @ Al @ Pm).tocsr(); lvl += 1
levels.append((Al, None, 2.0)); print("proto coarsest", Al.shape[0], Al.nnz)
for l in range(len(levels) - 1):
    Pd = Ad.amg_level(l, "P"); Pp = levels[l][1]; Pp.sort_indices()
    same = Pd.shape == Pp.shape and np.array_equal(Pd.indptr, Pp.indptr) and np.array_equal(Pd.indices, Pp.indices)
    print(f"P level {l}: structure equal {same}; max|dP| = {abs(Pd - Pp).max() if Pd.shape == Pp.shape else -1:.3e}")
    A1d = Ad.amg_level(l + 1, "A"); A1p = levels[l + 1][0]
    print(f"A level {l+1}: max|dA| = {abs(A1d - A1p).max() if A1d.shape == A1p.shape else -1:.3e}  max|A| = {abs(A1p).max():.3e}")
    Rd = Ad.amg_level(l, "R"); print(f"R == P^T: {abs(Rd - Pd.T).max():.1e}")
lu = spla.splu(Al.tocsc())
def vc(b, l=0):
    Al, Pm, lam = levels[l]
    if Pm is None: return lu.solve(b)
    dinv = 1.0 / Al.diagonal(); c = 1.0 / (0.55 * lam)
    x = c * dinv * b
    x = x + Pm @ vc(Pm.T @ (b - Al @ x), l + 1)
    return x + c * dinv * (b - Al @ x)
zp = vc(rv)
print("||z_dev - z_proto|| / ||z_proto|| =", np.linalg.norm(zd - zp) / np.linalg.norm(zp))
# symmetry / definiteness of the device operator
r2 = rng.uniform(-1, 1, A.shape[0]); z2 = Ad.amg_apply(r2)
print("symmetry: r2.M r1 =", r2 @ zd, " r1.M r2 =", rv @ z2, " r.Mr =", rv @ zd)
b = -r[1:n]
for M, name in ((vc, "proto"), (lambda v: Ad.amg_apply(v), "device")):
    x, it = P.pcg(A, b, M); print(name, "PCG iterations", it)
res = Ad.solve_spd(b, precond="amg"); print("device solve_spd iterations", res.iterations)

# ---- dissect level 1 ---------------------------------------------------------------------------------
A1 = Ad.amg_level(1, "A"); P1d = Ad.amg_level(1, "P")
agg, nc, _ = P.mis2_aggregate(strength_graph(A1, 0.1))
T = sp.csr_matrix((np.ones(A1.shape[0]), (np.arange(A1.shape[0]), agg)), shape=(A1.shape[0], nc))
AF = filtered(A1, 0.1); dF = AF.diagonal()
gF = (abs(AF).sum(axis=1).A1 / dF).max(); g = gersh(A1)
print("level 1: gersh_F", gF, "gersh", g, " min(dF/d) =", (dF / A1.diagonal()).min(), " #positive offdiag =", int(((A1 - sp.diags(A1.diagonal())).data > 0).sum()))
for lam in (gF, g, min(gF, g)):
    Pp = (T - (4.0 / 3.0 / lam) * (sp.diags(1.0 / dF) @ (AF @ T))).tocsr()
    print(f"  omega from lambda={lam:.4f}: max|P_dev - P_proto| = {abs(P1d - Pp).max():.3e}")
# unfiltered variant
Pu = (T - (4.0 / 3.0 / g) * (sp.diags(1.0 / A1.diagonal()) @ (A1 @ T))).tocsr()
print("  unfiltered proto vs device:", abs(P1d - Pu).max() if Pu.shape == P1d.shape else "shape differs")
D = (P1d - Pp).tocoo(); k = np.argmax(np.abs(D.data)); i = D.row[k]
print("  worst row", i, "dev", P1d[i].toarray()[0][P1d[i].indices], "proto", Pp[i].toarray()[0][Pp[i].indices], "dF/d", dF[i] / A1.diagonal()[i])
