"""CPU (scipy) experiment: does a nested-iteration (full multigrid) initial guess save CG iterations on the shipped
recipe?  x0 = FMG(b): restrict b to every level, solve the coarsest exactly, and on the way up interpolate and apply one
V(1,1) cycle of that level's own hierarchy to the level's residual.  Cost of the start: about 1.2 cycles.
python scripts/exp_fmg.py 8 220"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import amg_proto as P
import exp_aggressive as E
from oracle import padne_oracle as O
from padne_amd import synthetic as S


def pcg_from(A, b, M, x0, rtol=1e-12, maxit=500):
    x = x0.copy(); r = b - A @ x; z = M(r); p = z.copy(); rz = r @ z; bn = np.linalg.norm(b)
    if np.linalg.norm(r) <= rtol * bn: return x, 0, np.linalg.norm(r) / bn
    r0 = np.linalg.norm(r) / bn
    for it in range(1, maxit + 1):
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q
        if np.linalg.norm(r) <= rtol * bn: return x, it, r0
        z = M(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return x, maxit, r0


if __name__ == "__main__":
    nl, nx = int(sys.argv[1]), int(sys.argv[2])
    sysm = S.layered_system(nl, nx, nx, via_lattice=max(2, 32 * nx // 1118))
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)] + [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    L, r = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    n = sysm.n_vertices
    A = (-L[1:n, 1:n]).tocsr(); b = -r[1:n]
    lv = E.build(A, [1], 2048)
    vc = E.make_vc(lv)

    def fmg(bf, sweeps=1):
        bs = [bf]
        for (Al, Pm, lam) in lv[:-1]:
            bs.append(Pm.T @ bs[-1])
        x = vc(bs[-1], len(lv) - 1)                      # exact on the coarsest level
        for l in range(len(lv) - 2, -1, -1):
            x = lv[l][1] @ x
            for _ in range(sweeps):
                x = x + vc(bs[l] - lv[l][0] @ x, l)
        return x
    for name, x0 in (("zero start", np.zeros_like(b)), ("one cycle", vc(b)), ("nested iteration", fmg(b)),
                     ("nested iteration, 2 cycles per level", fmg(b, 2))):
        x, it, r0 = pcg_from(A, b, vc, x0)
        print(f"{name:40s}: initial relative residual {r0:.2e}, PCG iterations {it}", flush=True)
