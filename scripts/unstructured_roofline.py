"""Roofline evidence on CGAL-SHAPED input: Delaunay triangulations of jittered points with the hull's
slivers removed (unstructured connectivity, 20 % obtuse triangles), vertex ids SHUFFLED as an insertion-order mesher leaves
them (mesh.py:778-786), two layers tied by via resistors -- >= 5 M vertices in all.  The system goes through the product's own
path: assemble_from_arrays, the device plan of solve_system with the strip numbering made on the device, and then the
reduced matrix A is measured like the bench measures config C4:

  * the f64 product q = A p standalone (back-to-back launches) and in situ inside CG (HIP events), GB/s by the algorithmic
    bytes 12 nnz + 20 N + 4 of SURVEY 8d, share of the 64-row tiles on the x-window path / the gather path,
  * one complete solve with the hierarchy rebuilt: setup ms, iterations, us per iteration,
  * the same product on the matrix AS NUMBERED BY THE MESHER (no strip numbering), for the difference the numbering makes.

    python scripts/unstructured_roofline.py [--side 1620] [--spmv-only]     (--spmv-only: the rocprofv3 --pmc target)
The mesh is cached in /tmp (the triangulation takes half a minute of one core).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
if os.environ.get("PADNE_AB_LIB"):      # another build of the library (same-box A/B)
    from padne_amd import _hip as _hip_ab
    _hip_ab.LIB_PATH = os.path.abspath(os.environ["PADNE_AB_LIB"])

ap = argparse.ArgumentParser()
ap.add_argument("--side", type=int, default=1620)
ap.add_argument("--spmv-only", action="store_true")
ap.add_argument("--launches", type=int, default=20)
args = ap.parse_args()


def delaunay_mesh(side):
    path = f"/tmp/padne_delaunay_{side}.npz"
    if os.path.exists(path):
        z = np.load(path)
        return z["xy"], z["tri"], float(z["seconds"])
    import scipy.spatial
    from exp_unstructured_mesh import strip_hull_slivers
    rng = np.random.default_rng(7)
    gx, gy = np.meshgrid(np.arange(side, dtype=np.float64), np.arange(side, dtype=np.float64), indexing="xy")
    pts = np.stack([gx.ravel(), gy.ravel()], 1) * 0.5 + rng.uniform(-0.15, 0.15, (side * side, 2))
    t0 = time.perf_counter()
    tri = scipy.spatial.Delaunay(pts).simplices.astype(np.int32)
    dt = time.perf_counter() - t0
    a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
    cr = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
    tri[cr < 0] = tri[cr < 0][:, [0, 2, 1]]
    tri = strip_hull_slivers(pts, tri)
    n = len(pts)
    perm = rng.permutation(n)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(n)
    xy, tri = pts[perm], inv[tri].astype(np.int32)              # ids as an insertion-order mesher leaves them: scattered
    np.savez(path, xy=xy, tri=tri, seconds=dt)
    return xy, tri, dt


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
xy, tri, t_del = delaunay_mesh(args.side)
n = len(xy)
from padne_amd import _hip, mesh, solver  # noqa: E402
from padne_amd.reduction import build_reduction, floating_component_pins  # noqa: E402

ctx = solver.get_context()
meshes = [mesh.Mesh(xy, tri), mesh.Mesh(xy.copy(), tri.copy())]
nv = 2 * n
N = nv + 1
rng = np.random.default_rng(3)
st = solver.StampList(N)
g = 1.0 / 0.002
for v in rng.choice(n, 256, replace=False):                      # via resistors between the two layers
    st.rows += [int(v), int(v), int(n + v), int(n + v)]
    st.cols += [int(v), int(n + v), int(v), int(n + v)]
    st.vals += [-g, g, g, -g]
r = np.zeros(N)
src, snk = 17, n + n // 3
r[src] += 1.0
r[snk] -= 1.0
solver.setup_ground_node(n // 2, st, r)
t0 = time.perf_counter()
L = solver.assemble_from_arrays(meshes, [2082.5, 2082.5], st, nv)
ctx.synchronize()
t_asm = time.perf_counter() - t0
red = build_reduction(L.layout, [])
out = {"what": "two layers of a shuffled Delaunay mesh (scripts/unstructured_roofline.py)", "points_per_layer": n,
       "triangles_per_layer": int(len(tri)), "unknowns": nv, "delaunay_seconds": t_del, "assemble_seconds_host_arrays": t_asm}


def measure(plan, label):
    A = plan.reduced_matrix()
    nr, nnz = A.shape[0], A.nnz
    x = ctx.to_device(np.random.default_rng(1).uniform(-1, 1, A.shape[1]))
    y = ctx.empty(nr)
    t = A.spmv_time(x, y, 5, args.launches)                      # (the first launch builds the x-window plan: PADNE_VERBOSE=xw reports it)
    rec = {"rows": nr, "nnz": nnz, "nnz_per_row": nnz / nr, "spmv_bytes_algorithmic": A.spmv_bytes,
           "spmv_us_standalone": t * 1e6, "spmv_gbs_standalone": A.spmv_bytes / t / 1e9,
           "spmv_frac_of_8TBs_standalone": A.spmv_bytes / t / 1e9 / 8000.0}
    if not args.spmv_only:
        keep_rhs = np.zeros(nr)
        keep_rhs[0], keep_rhs[nr // 2] = 1.0, -1.0
        b = ctx.to_device(keep_rhs)
        xs = ctx.empty(nr)
        A.solve_spd_dev(b, xs, rtol=1e-12, precond="amg", rebuild=True)                     # warm-up
        res = A.solve_spd_dev(b, xs, rtol=1e-12, precond="amg", rebuild=True, time_spmv=True)
        rec.update({"iterations": int(res.iterations), "rel_residual": float(res.rel_residual), "levels": int(res.levels),
                    "operator_complexity": float(res.operator_complexity), "setup_ms": res.setup_seconds * 1e3,
                    "solve_ms": res.seconds * 1e3, "us_per_iteration": res.seconds / max(res.iterations, 1) * 1e6,
                    "spmv_us_in_situ": res.spmv_seconds * 1e6,
                    # (in the loop the product multiplies a search direction stored in single precision: 4 bytes of x per row less)
                    "spmv_bytes_in_situ": A.spmv_bytes - 4 * nr,
                    "spmv_gbs_in_situ": (A.spmv_bytes - 4 * nr) / res.spmv_seconds / 1e9 if res.spmv_seconds > 0 else None,
                    "spmv_frac_of_8TBs_in_situ": (A.spmv_bytes - 4 * nr) / res.spmv_seconds / 1e9 / 8000.0 if res.spmv_seconds > 0 else None})
    out[label] = rec


# the numbering solve_system uses for a scattered mesh: strips, made on the device
plan = _hip.KktPlan(L.dev, L.layout.n_potential, red.elim, red.tied, red.n_free, strip_order=True)
measure(plan, "strip_numbering")
plan.close()
if not args.spmv_only:
    plan = _hip.KktPlan(L.dev, L.layout.n_potential, red.elim, red.tied, red.n_free, strip_order=False)
    measure(plan, "as_numbered_by_the_mesher")
    plan.close()
    # the whole seam on the same system, host vectors in and out
    t0 = time.perf_counter()
    v, info = solver.solve_system(L, r)
    out["solve_system_seconds_first"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for plan_ in list(L._plans.values()):
        plan_.close()
    L._plans.clear()
    v, info = solver.solve_system(L, r)
    out["solve_system_seconds"] = time.perf_counter() - t0
    out["solve_system_iterations"] = int(info.iterations)
    out["solve_system_residual_norm"] = float(info.residual_norm)
print(json.dumps(out), flush=True)
