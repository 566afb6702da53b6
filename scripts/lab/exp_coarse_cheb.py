"""CPU (scipy) experiment: the exact solve on the coarsest level (a dense inverse: 2 ms of Gauss-Jordan in every setup)
replaced by k Chebyshev sweeps on D^-1 A (a fixed polynomial: still a fixed SPD preconditioner).  python scripts/exp_coarse_cheb.py 8 220"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import amg_proto as P
import exp_aggressive as E
from oracle import padne_oracle as O
from padne_amd import synthetic as S


def make_vc(levels, k):
    Ac = levels[-1][0]
    lu = spla.splu(Ac.tocsc())
    dinv = 1.0 / Ac.diagonal()
    Dm = sp.diags(np.sqrt(dinv))
    ev = np.linalg.eigvalsh((Dm @ Ac @ Dm).toarray())
    lmin, lmax = ev[0], ev[-1]
    print(f"   coarsest: n={Ac.shape[0]} spectrum of D^-1 A in [{lmin:.3e}, {lmax:.3f}] (condition {lmax/lmin:.0f})")
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)

    def cheb(b):                      # k steps of the Chebyshev iteration for D^-1 A x = D^-1 b from x = 0
        x = np.zeros_like(b); r = dinv * b
        sigma = theta / delta; rho = 1.0 / sigma; d = r / theta
        for _ in range(k):
            x = x + d
            r = r - dinv * (Ac @ d)
            rho_new = 1.0 / (2.0 * sigma - rho)
            d = rho_new * rho * d + (2.0 * rho_new / delta) * r
            rho = rho_new
        return x

    def vc(b, l=0):
        Al, Pm, lam = levels[l]
        if Pm is None: return lu.solve(b) if k == 0 else cheb(b)
        dinv_l = 1.0 / Al.diagonal(); c = 1.0 / (0.55 * lam)
        x = c * dinv_l * b
        x = x + Pm @ vc(Pm.T @ (b - Al @ x), l + 1)
        return x + c * dinv_l * (b - Al @ x)
    return vc


if __name__ == "__main__":
    nl, nx = int(sys.argv[1]), int(sys.argv[2])
    sysm = S.layered_system(nl, nx, nx, via_lattice=max(2, 32 * nx // 1118))
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)] + [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    L, r = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    n = sysm.n_vertices
    A = (-L[1:n, 1:n]).tocsr(); b = -r[1:n]
    lv = E.build(A, [1], 2048)
    for k in (0, 8, 16, 32, 64):
        x, it = P.pcg(A, b, make_vc(lv, k))
        print(f"coarsest level: {'exact solve' if k == 0 else str(k) + ' Chebyshev sweeps'} => PCG iterations {it}", flush=True)
