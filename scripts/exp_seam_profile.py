"""cProfile of solver.solve_system at bench scale (where does the host time of the Python seam go?)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from padne_amd import mesh, solver, synthetic

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
sysm = synthetic.config(name); nv = sysm.n_vertices; N = nv + 1
ctx = solver.get_context()
meshes = [mesh.Mesh(m[0], m[1]) for m in sysm.meshes]; sig = [m[2] for m in sysm.meshes]
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
st = solver.StampList(N)
st.rows, st.cols, st.vals = list(rows[:-2]), list(cols[:-2]), list(vals[:-2])
r = rhs.copy()
solver.setup_ground_node(sysm.ground, st, r)
L = solver.assemble_from_arrays(meshes, sig, st, nv)
solver.solve_system(L, r)                      # warm-up
t0 = time.perf_counter(); v, info = solver.solve_system(L, r); t1 = time.perf_counter()
print(f"solve_system {t1-t0:.3f} s (device solve {info.solve_seconds:.3f} s, {info.iterations} it)", flush=True)
pr = cProfile.Profile(); pr.enable(); solver.solve_system(L, r); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
