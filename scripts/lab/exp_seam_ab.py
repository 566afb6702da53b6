"""Wall time of solve_system(L, r) on config C4 (cold plan and cached plan, medians of 7) with a given build of the library:
    python scripts/lab/exp_seam_ab.py padne_amd/libpadne_hip.so [other.so ...]   (one child process per library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 or (len(sys.argv) == 2 and not sys.argv[1].startswith("--child=")):
    for lib in sys.argv[1:]:
        subprocess.call([sys.executable, os.path.abspath(__file__), "--child=" + os.path.abspath(lib)])
    sys.exit(0)
sys.path.insert(0, ROOT)
import time
import numpy as np
from padne_amd import _hip
_hip.LIB_PATH = sys.argv[1].split("=", 1)[1]
import bench
from padne_amd import solver, synthetic
from padne_amd.reduction import Constraint, KKTLayout
ctx = _hip.Context(0)
sysm, xy, tri = synthetic.config_on_device(ctx, "C4")
N = sysm.n_vertices + 1
sig = np.array([m[2] for m in sysm.meshes])
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
L = ctx.assemble_system(N, xy, tri, sysm.mesh_offsets, sysm._tri_offsets, sig, rows, cols, vals)
solver.set_context(ctx)
Ls = solver.SystemMatrix(L, KKTLayout(size=N, n_potential=N - 1, constraints=[Constraint(index=N - 1, p=int(sysm.ground), n=-1, value=0.0)]))
solver.solve_system(Ls, rhs)
cold, cached = [], []
for _ in range(7):
    for plan in Ls._plans.values():
        plan.close()
    Ls._plans.clear()
    ctx.synchronize()
    t0 = time.perf_counter(); solver.solve_system(Ls, rhs); cold.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); solver.solve_system(Ls, rhs); cached.append(time.perf_counter() - t0)
print(f"{os.path.basename(_hip.LIB_PATH):24s} cold plan {np.median(cold)*1e3:6.2f} ms   cached plan {np.median(cached)*1e3:6.2f} ms", flush=True)
