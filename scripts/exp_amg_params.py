"""Sweep of the multigrid parameters (environment overrides of amg.hip) on a synthetic config: iterations and times."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from padne_amd import _hip, synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
ctx = _hip.Context(0)
if name in ("delaunay", "graded"):
    import scipy.spatial
    from padne_amd import reduction
    rng = np.random.default_rng(3)
    npts = 1000000
    if name == "delaunay":
        pts = rng.uniform(0, 100, (npts, 2))
    else:
        u = rng.uniform(0, 1, (npts, 2)); rad = 100.0 * u[:, 0] ** 2.5
        pts = np.column_stack([rad * np.cos(2 * np.pi * u[:, 1]), rad * np.sin(2 * np.pi * u[:, 1])])
        pts = np.unique(np.round(pts, 9), axis=0)
    dt = scipy.spatial.Delaunay(pts).simplices.astype(np.int32)
    a_, b_, c_ = pts[dt[:, 0]], pts[dt[:, 1]], pts[dt[:, 2]]
    cr = (b_[:, 0] - a_[:, 0]) * (c_[:, 1] - a_[:, 1]) - (b_[:, 1] - a_[:, 1]) * (c_[:, 0] - a_[:, 0])
    dt = dt[np.abs(cr) > 1e-12]; cr = cr[np.abs(cr) > 1e-12]
    dt[cr < 0] = dt[cr < 0][:, [0, 2, 1]]
    used = np.unique(dt); remap = -np.ones(len(pts), dtype=np.int64); remap[used] = np.arange(len(used))
    pts, dt = pts[used], remap[dt].astype(np.int32)
    n = len(pts)
    e = np.zeros(0, np.int64)
    L = ctx.assemble_system(n, pts, dt, [0, n], [0, len(dt)], [2082.5], e, e, np.zeros(0))
    st = reduction.strip_index(pts, np.zeros(n, dtype=np.int64))
    order = np.lexsort((pts[:, 0], st))
    rank = np.full(n, -1, dtype=np.int32)
    keepv = order[order != 11]                                  # vertex 11 is the ground
    rank[keepv] = np.arange(n - 1, dtype=np.int32)
    A = L.reduce(rank, n - 1, -1.0); L.close()
    rhs_full = np.zeros(n); rhs_full[int(np.argmin(pts.sum(1)))] = 1.0; rhs_full[int(np.argmax(pts.sum(1)))] = -1.0
    bb = np.zeros(n - 1); bb[rank[rank >= 0]] = -rhs_full[rank >= 0]
    b = ctx.to_device(bb); x = ctx.empty(n - 1)
else:
    sysm = synthetic.config(name); nv = sysm.n_vertices; N = nv + 1
    xy, tri, mvo, mto, sig = bench.flat(sysm); rows, cols, vals, rhs = bench.stamps_of(sysm, N)
    L = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
    imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
    A = L.reduce(imap, nv - 1, -1.0); L.close()
    keep = np.flatnonzero(imap[:nv] >= 0)
    b = ctx.to_device(-rhs[keep]); x = ctx.empty(nv - 1)
def run(env):
    for k in ("PADNE_AMG_CHEB_RATIO", "PADNE_AMG_OMEGA", "PADNE_AMG_THETA", "PADNE_AMG_F64", "PADNE_AMG_THETA_DECAY", "PADNE_AMG_COARSE_N"): os.environ.pop(k, None)
    os.environ.update(env)
    A.solve_spd_dev(b, x, precond="amg", rebuild=True, raise_on_fail=False)
    t = time.perf_counter(); r = A.solve_spd_dev(b, x, precond="amg", rebuild=True, raise_on_fail=False); w = time.perf_counter() - t
    print(f"{env}: fallbacks {r.precond_fallbacks} status {r.status} iterations {r.iterations} levels {r.levels} complexity {r.operator_complexity:.3f} setup {r.setup_seconds*1e3:.1f} ms "
          f"solve {r.seconds*1e3:.1f} ms wall {w*1e3:.1f} ms relres {r.rel_residual:.1e}", flush=True)
run({})
if len(sys.argv) > 2:
    run({"PADNE_AMG_F64": "1"})
    rj = A.solve_spd_dev(b, x, precond="jacobi", raise_on_fail=False)
    print("jacobi:", rj.iterations, rj.rel_residual, rj.status, flush=True)
    sys.exit(0)
for dec in ("0.7", "0.5", "0.35"):
    run({"PADNE_AMG_THETA_DECAY": dec})
for cn in ("512", "2048", "3000"):
    os.environ["PADNE_AMG_COARSE_N"] = cn
    run({"PADNE_AMG_COARSE_N": cn})
os.environ.pop("PADNE_AMG_COARSE_N", None)
for th, om in itertools.product(("0.07", "0.09"), ("1.4", "1.5", "1.6")):
    run({"PADNE_AMG_THETA": th, "PADNE_AMG_OMEGA": om})
