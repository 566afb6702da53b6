"""Where solve_system's time goes at bench scale: wall time of every device call of one solve_system."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from padne_amd import _hip, mesh, solver, synthetic

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
log = []
def timed(cls, meth):
    orig = getattr(cls, meth)
    def wrap(self, *a, **k):
        t = time.perf_counter(); out = orig(self, *a, **k); ctx.synchronize(); dt = time.perf_counter() - t
        extra = ""
        if meth == "solve_spd":
            extra = f" setup {out.setup_seconds*1e3:.1f} ms, iterations {out.seconds*1e3:.1f} ms ({out.iterations} it, {out.restarts} restarts, levels {out.levels})"
        log.append(f"   {meth:16s} {dt*1e3:8.1f} ms{extra}")
        return out
    setattr(cls, meth, wrap)
for m in ("solve_spd", "matvec", "reduce", "residual_norm", "close"):
    timed(_hip.CsrMatrix, m)
for k in range(3):
    log.clear()
    t0 = time.perf_counter(); v, info = solver.solve_system(L, r); t1 = time.perf_counter()
    print(f"solve_system {1e3*(t1-t0):.1f} ms"); print("\n".join(log), flush=True)
