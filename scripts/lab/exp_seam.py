"""Host-inclusive cost of the Python seam at bench scale: assemble_from_arrays + solve_system + power density
through padne_amd.solver (host buffers in, host buffers out), config C4."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from padne_amd import mesh, solver, synthetic
from padne_amd.reduction import Constraint

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
sysm = synthetic.config(name); nv = sysm.n_vertices; N = nv + 1
ctx = solver.get_context()
meshes = [mesh.Mesh(m[0], m[1]) for m in sysm.meshes]; sig = [m[2] for m in sysm.meshes]
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
for rep in range(3):
    t0 = time.perf_counter()
    st = solver.StampList(N)
    st.rows, st.cols, st.vals = list(rows[:-2]), list(cols[:-2]), list(vals[:-2])
    r = rhs.copy()
    solver.setup_ground_node(sysm.ground, st, r)
    t1 = time.perf_counter()
    L = solver.assemble_from_arrays(meshes, sig, st, nv)
    ctx.synchronize(); t2 = time.perf_counter()
    v, info = solver.solve_system(L, r)
    t3 = time.perf_counter()
    pd = L.dev.power_density(v[:nv], len(L.tri))          # the mesh stayed on the device with the assembled system
    t4 = time.perf_counter()
    L.dev.close()
    print(f"[{name}] stamps(list) {t1-t0:.3f} s | assemble (H2D 0.4 GB + kernels) {t2-t1:.3f} s | solve_system (host reduction + device) {t3-t2:.3f} s "
          f"[{info.iterations} it, residual {info.residual_norm:.2e}, device solve {info.solve_seconds:.3f} s] | power density {t4-t3:.3f} s | total {t4-t0:.3f} s", flush=True)
