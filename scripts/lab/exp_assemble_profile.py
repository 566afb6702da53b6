import cProfile, os, pstats, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from padne_amd import mesh, solver, synthetic
sysm = synthetic.config("C4"); nv = sysm.n_vertices; N = nv + 1
ctx = solver.get_context()
meshes = [mesh.Mesh(m[0], m[1]) for m in sysm.meshes]; sig = [m[2] for m in sysm.meshes]
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
def build():
    st = solver.StampList(N)
    st.rows, st.cols, st.vals = list(rows[:-2]), list(cols[:-2]), list(vals[:-2])
    r = rhs.copy()
    solver.setup_ground_node(sysm.ground, st, r)
    return st
st = build(); L = solver.assemble_from_arrays(meshes, sig, st, nv); L.dev.close()
st = build()
pr = cProfile.Profile(); pr.enable(); L = solver.assemble_from_arrays(meshes, sig, st, nv); ctx.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
