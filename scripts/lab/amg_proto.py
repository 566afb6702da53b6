"""Numerical prototype (CPU, scipy) of the smoothed-aggregation AMG preconditioner planned for the device.
Dev tool only: explores aggregation / smoothing choices and iteration counts before writing kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
from oracle import padne_oracle as O
from padne_amd import synthetic as S

def hash32(i):
    i = (i.astype(np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    i ^= i >> np.uint64(15); i = (i * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF); i ^= i >> np.uint64(13)
    return i

def neighbor_max(A, val):
    """max over row neighbours (incl. self) of val -- one 'SpMV-like' pass."""
    n = A.shape[0]
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    out = val.copy()
    np.maximum.at(out, rows, val[A.indices])
    return out

def mis2_aggregate(A):
    """Deterministic Luby-style MIS-2 + two assignment passes. Returns agg[n], n_agg."""
    n = A.shape[0]
    # state: 0 undecided, 1 root, 2 covered (within distance 2 of a root)
    state = np.zeros(n, np.int8)
    prio = (hash32(np.arange(n)) << np.uint64(32)) | np.arange(n, dtype=np.uint64)   # unique
    rounds = 0
    while (state == 0).any():
        rounds += 1
        p = np.where(state == 0, prio, np.uint64(0))
        m1 = neighbor_max(A, p)
        m2 = neighbor_max(A, m1)
        newroot = (state == 0) & (m2 == prio)
        state[newroot] = 1
        r = np.where(state == 1, np.uint64(1), np.uint64(0))
        c1 = neighbor_max(A, r); c2 = neighbor_max(A, c1)
        state[(state == 0) & (c2 > 0)] = 2
    roots = np.flatnonzero(state == 1)
    agg = -np.ones(n, np.int64); agg[roots] = np.arange(len(roots))
    # pass 1: distance-1 neighbours of a root join it (strongest root if several)
    n_rows = np.repeat(np.arange(n), np.diff(A.indptr))
    w = -A.data  # positive couplings
    for _ in range(2):
        cand = agg[A.indices]
        ok = (cand >= 0) & (agg[n_rows] < 0) & (A.indices != n_rows)
        # strongest connection wins: sort by (row, weight)
        order = np.lexsort((w[ok], n_rows[ok]))
        rr = n_rows[ok][order]; cc = cand[ok][order]
        last = np.r_[rr[1:] != rr[:-1], True]
        agg_new = agg.copy(); agg_new[rr[last]] = cc[last]
        agg = agg_new
    # leftovers (isolated): own aggregate
    left = np.flatnonzero(agg < 0)
    agg[left] = len(roots) + np.arange(len(left))
    return agg, len(roots) + len(left), rounds

def build_hierarchy(A, max_levels=12, coarse_n=400, omega=2.0/3.0, smooth=True):
    levels = []
    while A.shape[0] > coarse_n and len(levels) < max_levels:
        agg, nc, rounds = mis2_aggregate(A)
        n = A.shape[0]
        T = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nc))
        if smooth:
            Dinv = sp.diags(1.0 / A.diagonal())
            P = (T - omega * (Dinv @ (A @ T))).tocsr()
        else:
            P = T
        Ac = (P.T @ A @ P).tocsr()
        levels.append((A, P))
        print(f"  level {len(levels)-1}: n={n} nnz={A.nnz} -> nc={nc} (ratio {n/nc:.1f}) nnz(P)={P.nnz} mis rounds={rounds}", flush=True)
        A = Ac
    levels.append((A, None))
    print(f"  coarsest: n={A.shape[0]} nnz={A.nnz}")
    return levels

def vcycle(levels, b, lvl=0, omega=2.0/3.0, nu=1, coarse_sweeps=40):
    A, P = levels[lvl]
    dinv = 1.0 / A.diagonal()
    if P is None:
        x = np.zeros_like(b)
        for _ in range(coarse_sweeps):
            x += omega * dinv * (b - A @ x)
        return x
    x = omega * dinv * b
    for _ in range(nu - 1):
        x += omega * dinv * (b - A @ x)
    r = b - A @ x
    xc = vcycle(levels, P.T @ r, lvl + 1, omega, nu, coarse_sweeps)
    x += P @ xc
    for _ in range(nu):
        x += omega * dinv * (b - A @ x)
    return x

def pcg(A, b, M, rtol=1e-12, maxit=500):
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z; bn = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q
        if np.linalg.norm(r) <= rtol * bn: return x, it
        z = M(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return x, maxit

if __name__ == "__main__":
    nl, nx = int(sys.argv[1]), int(sys.argv[2])
    sysm = S.layered_system(nl, nx, nx, via_lattice=max(2, 32 * nx // 1118))
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)] + [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    L, r = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
    n = sysm.n_vertices
    A = (-L[1:n, 1:n]).tocsr(); b = -r[1:n]
    for smooth in (True, False):
        t = time.time(); lv = build_hierarchy(A, smooth=smooth); print("setup", time.time() - t)
        cx = sum(l[0].nnz for l in lv) / A.nnz
        for nu in (1, 2):
            x, it = pcg(A, b, lambda rr: vcycle(lv, rr, nu=nu))
            print(f"smooth={smooth} nu={nu}: PCG iterations {it}, operator complexity {cx:.2f}, true relres {np.linalg.norm(b - A@x)/np.linalg.norm(b):.2e}", flush=True)
