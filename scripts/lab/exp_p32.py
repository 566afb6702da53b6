"""CPU (scipy) experiment: PCG with the search direction rounded to single precision every iteration (x and r updated with
the rounded vector, so b - A x stays tracked) against the plain loop, on the device's multigrid recipe (exp_aggressive.build)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import exp_aggressive as E
import amg_proto as P
from oracle import padne_oracle as O
from padne_amd import synthetic as S
def pcg_p32(A, b, M, rtol=1e-12, maxit=500, p32=True, z32=True):
    x = np.zeros_like(b); r = b.copy()
    def prec(r):
        z = M(r)
        return z.astype(np.float32).astype(np.float64) if z32 else z
    z = prec(r); p = z.copy()
    if p32: p = p.astype(np.float32).astype(np.float64)
    rz = r @ z; bn = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q
        if np.linalg.norm(r) <= rtol * bn: return x, it
        z = prec(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
        if p32: p = p.astype(np.float32).astype(np.float64)
    return x, maxit
for nl, nx in ((8, 400), (4, 300)):
    sysm = S.layered_system(nl, nx, nx, via_lattice=max(2, 32 * nx // 1118))
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)] + [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    L, r = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    n = sysm.n_vertices
    A = (-L[1:n, 1:n]).tocsr(); b = -r[1:n]
    lv = E.build(A, [1], 2500)
    vc = E.make_vc(lv)
    for p32 in (False, True):
        x, it = pcg_p32(A, b, vc, p32=p32)
        print(nl, nx, "p32" if p32 else "p64", it, np.linalg.norm(b - A @ x) / np.linalg.norm(b), flush=True)
