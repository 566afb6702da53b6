"""Debug companion of exp_unstructured.py: the reduced SPD system of the shuffled Delaunay two-layer problem, solved
with the multigrid in single / double precision and with Jacobi; prints iterations and the residual reached."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.spatial
from padne_amd import _hip, reduction


def strip_hull_slivers(p, t, min_deg=12.0):
    """Drop sliver triangles that sit on the boundary (the hull of jittered grid points carries triangles with angles
    of 0 / 180 degrees; a quality mesher never emits those), repeatedly, interior triangles stay."""
    def min_angle(t):
        out = []
        for k in range(3):
            u = p[t[:, (k + 1) % 3]] - p[t[:, k]]; v = p[t[:, (k + 2) % 3]] - p[t[:, k]]
            out.append(np.degrees(np.arccos(np.clip((u * v).sum(1) / np.linalg.norm(u, axis=1) / np.linalg.norm(v, axis=1), -1, 1))))
        return np.stack(out, 1).min(1)
    for _ in range(20):
        e = np.sort(np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]), axis=1).astype(np.int64)
        code = e[:, 0] * (len(p) + 1) + e[:, 1]
        _, inv, cnt = np.unique(code, return_inverse=True, return_counts=True)
        on_boundary = (cnt[inv] == 1).reshape(3, -1).any(0)
        bad = on_boundary & (min_angle(t) < min_deg)
        if not bad.any():
            break
        t = t[~bad]
    return t

side = int(sys.argv[1]) if len(sys.argv) > 1 else 700
shuffle = "--noshuffle" not in sys.argv
rng = np.random.default_rng(7)
gx, gy = np.meshgrid(np.arange(side, dtype=np.float64), np.arange(side, dtype=np.float64), indexing="xy")
pts = np.stack([gx.ravel(), gy.ravel()], 1) * 0.5 + rng.uniform(-0.15, 0.15, (side * side, 2))
tri = scipy.spatial.Delaunay(pts).simplices.astype(np.int32)
a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
cr = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
tri[cr < 0] = tri[cr < 0][:, [0, 2, 1]]
# element quality
def angles(p, t):
    out = []
    for k in range(3):
        u = p[t[:, (k + 1) % 3]] - p[t[:, k]]; v = p[t[:, (k + 2) % 3]] - p[t[:, k]]
        out.append(np.degrees(np.arccos(np.clip((u * v).sum(1) / np.linalg.norm(u, axis=1) / np.linalg.norm(v, axis=1), -1, 1))))
    return np.stack(out, 1)
tri = strip_hull_slivers(pts, tri)
ang = angles(pts, tri)
print(f"n={len(pts)} triangles={len(tri)} min angle {ang.min():.2f} deg, max {ang.max():.2f}; obtuse fraction {(ang.max(1) > 90).mean():.3f}", flush=True)
n = len(pts)
if shuffle:
    perm = rng.permutation(n); inv = np.empty_like(perm); inv[perm] = np.arange(n)
    pts, tri = pts[perm], inv[tri].astype(np.int32)
ctx = _hip.Context(0)
xy = np.concatenate([pts, pts]); tr = np.concatenate([tri, tri])
mvo = np.array([0, n, 2 * n], dtype=np.int64); mto = np.array([0, len(tri), 2 * len(tri)], dtype=np.int64)
nv = 2 * n; N = nv + 1
vias = rng.choice(n, 64, replace=False); g = 500.0
rows = np.concatenate([vias, vias, n + vias, n + vias]); cols = np.concatenate([vias, n + vias, vias, n + vias])
vals = np.concatenate([-g * np.ones(64), g * np.ones(64), g * np.ones(64), -g * np.ones(64)])
L = ctx.assemble_system(N, xy, tr, mvo, mto, [2082.5, 2082.5], rows, cols, vals)
ground = n // 2
imap = np.arange(N, dtype=np.int32); imap[ground] = -1; imap[imap > ground] -= 1; imap[N - 1] = -1
if shuffle:
    # strip ordering of the free unknowns, like solve_system
    st = reduction.strip_index(xy, np.repeat([0, 1], n).astype(np.int64)) if hasattr(reduction, "strip_index") else None
    key = np.lexsort((xy[:, 0], st, np.repeat([0, 1], n)))
    rank = np.empty(nv, dtype=np.int64); rank[key] = np.arange(nv)
    free = imap[:nv] >= 0
    order = np.argsort(rank[free], kind="stable"); newid = np.empty(free.sum(), dtype=np.int32); newid[order] = np.arange(free.sum(), dtype=np.int32)
    imap[:nv][free] = newid
A = L.reduce(imap, nv - 1, -1.0)
rhs = np.zeros(nv); rhs[0] += 1.0; rhs[nv - 1] -= 1.0
b = np.zeros(nv - 1); keep = imap[:nv] >= 0; b[imap[:nv][keep]] = rhs[keep]
for label, env, prec in (("amg f32", {}, "amg"), ("amg f64", {"PADNE_AMG_F64": "1"}, "amg"), ("amg f32, no x windows", {"PADNE_NO_XWINDOW": "1"}, "amg"), ("jacobi", {}, "jacobi")):
    for k, v in env.items(): os.environ[k] = v
    A2 = L.reduce(imap, nv - 1, -1.0)
    t = time.perf_counter(); res = A2.solve_spd(b, rtol=1e-12, precond=prec, raise_on_fail=False); w = time.perf_counter() - t
    x = res.x
    print(f"{label:24s}: status {res.status} iterations {res.iterations} levels {res.levels} relres {res.rel_residual:.2e} restarts {getattr(res, 'restarts', None)} wall {w*1e3:.0f} ms", flush=True)
    for k in env: del os.environ[k]
    A2.close()
