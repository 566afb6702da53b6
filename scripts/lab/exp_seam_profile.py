"""Host profile of solve_system(L, r) on config C4 with the plan cached (a further right-hand side): where the time between
the device solve and the caller goes.  python scripts/exp_seam_profile.py"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from padne_amd import _hip, solver, synthetic
from padne_amd.reduction import Constraint, KKTLayout

ctx = _hip.Context(0)
sysm, xy, tri = synthetic.config_on_device(ctx, "C4")
nv = sysm.n_vertices
N = nv + 1
sig = np.array([m[2] for m in sysm.meshes])
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
L = ctx.assemble_system(N, xy, tri, sysm.mesh_offsets, sysm._tri_offsets, sig, rows, cols, vals)
solver.set_context(ctx)
layout = KKTLayout(size=N, n_potential=N - 1, constraints=[Constraint(index=N - 1, p=int(sysm.ground), n=-1, value=0.0)])
Ls = solver.SystemMatrix(L, layout)
for _ in range(2):
    v, info = solver.solve_system(Ls, rhs)
t0 = time.perf_counter()
v, info = solver.solve_system(Ls, rhs)
print("cached-plan call: %.2f ms wall, device solve %.2f ms, iterations %d" % ((time.perf_counter() - t0) * 1e3, info.solve_seconds * 1e3, info.iterations))
pr = cProfile.Profile()
pr.enable()
v, info = solver.solve_system(Ls, rhs)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(25)
print(s.getvalue()[:6000])
