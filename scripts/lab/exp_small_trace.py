"""A cold solve_system(L, r) at config C1's size (11 k unknowns), for a kernel trace: rocprofv3 --kernel-trace -- python3 scripts/lab/exp_small_trace.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from padne_amd import _hip, solver, synthetic
from padne_amd.reduction import Constraint, KKTLayout
ctx = _hip.Context(0)
solver.set_context(ctx)
sysm = synthetic.layered_system(4, 53, 53, via_lattice=2)
N = sysm.n_vertices + 1
xy, tri, mvo, mto, sig = bench.flat(sysm)
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
L = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
Ls = solver.SystemMatrix(L, KKTLayout(size=N, n_potential=N - 1, constraints=[Constraint(index=N - 1, p=int(sysm.ground), n=-1, value=0.0)]))
for k in range(4):
    for plan in Ls._plans.values():
        plan.close()
    Ls._plans.clear()
    ctx.synchronize()
    t0 = time.perf_counter()
    v, info = solver.solve_system(Ls, rhs)
    print(f"cold solve_system: {(time.perf_counter() - t0) * 1e3:.2f} ms, setup {info.__dict__.get('setup_seconds', 0)}", flush=True)
