"""Randomised bit-for-bit check of the device assembly against the oracle (GPU): 1-3 random Delaunay meshes per case
(with and without a hole, clustered points for high-degree vertices), random conductances, random stamps on mesh
vertices and internal nodes.  Structure and every value must be identical.  python scripts/fuzz_assembly.py [n] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.spatial
from oracle import padne_oracle as O
from padne_amd import _hip


def random_mesh(rng, n_points, hole, clusters):
    pts = rng.uniform(0, 40, (n_points, 2))
    for _ in range(clusters):                                   # a fan: many points on a small circle around a centre
        c = rng.uniform(5, 35, 2); k = int(rng.integers(10, 22)); rad = rng.uniform(0.2, 0.6)
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        pts = pts[np.hypot(pts[:, 0] - c[0], pts[:, 1] - c[1]) > 1.5 * rad]
        pts = np.concatenate([pts, [c], c + rad * np.stack([np.cos(ang), np.sin(ang)], 1)])
    if hole:
        pts = pts[np.hypot(pts[:, 0] - 20, pts[:, 1] - 20) > 6.0]
    tri = scipy.spatial.Delaunay(pts).simplices.astype(np.int32)
    if hole:
        c = pts[tri].mean(axis=1)
        tri = tri[np.hypot(c[:, 0] - 20, c[:, 1] - 20) > 6.5]
    a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
    cross = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
    tri = tri[cross != 0]
    cross = cross[cross != 0]
    tri[cross < 0] = tri[cross < 0][:, [0, 2, 1]]
    used = np.unique(tri)
    remap = -np.ones(len(pts), dtype=np.int64); remap[used] = np.arange(len(used))
    return pts[used], remap[tri].astype(np.int32)


def run(n_cases=40, seed0=0, verbose=True):
    ctx = _hip.Context(0)
    t0 = time.perf_counter(); max_deg = 0
    for case in range(n_cases):
        rng = np.random.default_rng(seed0 * 1000 + case)
        meshes = []
        for _ in range(int(rng.integers(1, 4))):
            xy, tri = random_mesh(rng, int(rng.integers(200, 6000)), bool(rng.integers(2)), int(rng.integers(0, 4)))
            try:
                O.check_manifold(len(xy), tri)
            except Exception:
                continue
            meshes.append((xy, tri, float(rng.choice([2082.5, 1041.25, 333.3]))))
        if not meshes:
            continue
        n_vert = sum(len(m[0]) for m in meshes)
        n_int = int(rng.integers(0, 3))
        els = []
        for _ in range(int(rng.integers(0, 40))):
            a, b = int(rng.integers(0, n_vert + n_int)), int(rng.integers(0, n_vert + n_int))
            if a != b:
                els.append(("R", a, b, float(10 ** rng.uniform(-3, 2))))
        hub = int(rng.integers(0, n_vert))                      # one vertex with many stamps (a long slot row)
        for _ in range(int(rng.integers(0, 30))):
            b = int(rng.integers(0, n_vert))
            if b != hub:
                els.append(("R", hub, b, float(10 ** rng.uniform(-3, 1))))
        Lo, ro = O.assemble_system(meshes, n_int, els, 0)
        Lo.sort_indices()
        N = Lo.shape[0]
        rows, cols, vals = [], [], []
        for e in els:
            g = 1 / e[3]
            rows += [e[1], e[1], e[2], e[2]]; cols += [e[1], e[2], e[2], e[1]]; vals += [-g, g, -g, g]
        rows += [N - 1, 0]; cols += [0, N - 1]; vals += [1.0, 1.0]                 # ground constraint (setup_ground_node)
        xy = np.concatenate([m[0] for m in meshes]); tri = np.concatenate([m[1] for m in meshes])
        mvo = np.concatenate([[0], np.cumsum([len(m[0]) for m in meshes])]); mto = np.concatenate([[0], np.cumsum([len(m[1]) for m in meshes])])
        d = ctx.assemble_system(N, xy, tri, mvo, mto, [m[2] for m in meshes], rows, cols, vals)
        got = d.to_scipy(); d.close()
        deg = max(np.bincount(m[1].reshape(-1)).max() for m in meshes); max_deg = max(max_deg, deg)
        ok = got.shape == Lo.shape and np.array_equal(got.indptr, Lo.indptr) and np.array_equal(got.indices, Lo.indices) and np.array_equal(got.data, Lo.data)
        if verbose or not ok:
            print(f"case {case:3d}: meshes {len(meshes)} N {N:6d} nnz {Lo.nnz:7d} stamps {len(els):3d} max triangles at a vertex {deg:2d}  {'identical' if ok else 'DIFFERENT'}", flush=True)
        assert ok, f"assembly differs from the oracle in case {case} of seed {seed0}"
    if verbose:
        print(f"{n_cases} cases identical to the oracle bit for bit (largest vertex degree {max_deg}) in {time.perf_counter()-t0:.0f} s")
    return max_deg


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
