"""A CGAL-like input at scale: Delaunay triangulation of ~2 M jittered points (unstructured connectivity, good element
quality), vertex ids shuffled as an insertion-order mesher leaves them, two layers tied by a handful of via resistors,
through the Python seam (assemble_from_arrays + solve_system, which renumbers in strips).  Reports whether the
x-window plan engages, iterations, times.  python scripts/exp_unstructured.py [points_per_side]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.spatial
from padne_amd import mesh, solver


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

side = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1000
rng = np.random.default_rng(7)
gx, gy = np.meshgrid(np.arange(side, dtype=np.float64), np.arange(side, dtype=np.float64), indexing="xy")
pts = np.stack([gx.ravel(), gy.ravel()], 1) * 0.5 + rng.uniform(-0.15, 0.15, (side * side, 2))
t = time.perf_counter(); tri = scipy.spatial.Delaunay(pts).simplices.astype(np.int32); t_del = time.perf_counter() - t
a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
cr = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
tri[cr < 0] = tri[cr < 0][:, [0, 2, 1]]
tri = strip_hull_slivers(pts, tri)
n = len(pts)
perm = rng.permutation(n); inv = np.empty_like(perm); inv[perm] = np.arange(n)
sxy, stri = pts[perm], inv[tri].astype(np.int32)
print(f"Delaunay of {n} points: {len(tri)} triangles in {t_del:.1f} s; ids shuffled", flush=True)
meshes = [mesh.Mesh(sxy, stri), mesh.Mesh(sxy.copy(), stri.copy())]
sig = [2082.5, 2082.5]
nv = 2 * n; N = nv + 1
ctx = solver.get_context()
for rep in range(3):
    t0 = time.perf_counter()
    st = solver.StampList(N)
    vias = rng.choice(n, 64, replace=False)
    g = 1.0 / 0.002
    for v in vias:                                  # resistor between the two layers
        st.rows += [int(v), int(v), int(n + v), int(n + v)]; st.cols += [int(v), int(n + v), int(v), int(n + v)]; st.vals += [-g, g, g, -g]
    r = np.zeros(N); src, snk = int(inv[0]), int(n + inv[n - 1]); r[src] += 1.0; r[snk] -= 1.0
    solver.setup_ground_node(int(inv[n // 2]), st, r)
    L = solver.assemble_from_arrays(meshes, sig, st, nv)
    ctx.synchronize(); t1 = time.perf_counter()
    v, info = solver.solve_system(L, r)
    t2 = time.perf_counter()
    L.dev.close()
    print(f"assemble {t1-t0:.3f} s | solve_system {t2-t1:.3f} s [{info.iterations} it, residual {info.residual_norm:.2e}, device solve {info.solve_seconds*1e3:.1f} ms] "
          f"| drop source->sink {v[src]-v[snk]:.6f} V", flush=True)
if "--profile" in sys.argv:
    import cProfile, pstats
    st = solver.StampList(N)
    for v in vias:
        st.rows += [int(v), int(v), int(n + v), int(n + v)]; st.cols += [int(v), int(n + v), int(v), int(n + v)]; st.vals += [-g, g, g, -g]
    r = np.zeros(N); r[src] += 1.0; r[snk] -= 1.0
    solver.setup_ground_node(int(inv[n // 2]), st, r)
    L = solver.assemble_from_arrays(meshes, sig, st, nv)
    pr = cProfile.Profile(); pr.enable(); solver.solve_system(L, r); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
