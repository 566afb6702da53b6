"""Host-vector solve (padne_solve_spd) against the device-vector solve (padne_solve_spd_dev) of the same reduced system."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from padne_amd import _hip, synthetic
ctx = _hip.Context(0)
sysm, xy, tri = synthetic.config_on_device(ctx, sys.argv[1] if len(sys.argv) > 1 else "C4")
nv = sysm.n_vertices; N = nv + 1
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
sig = np.array([m[2] for m in sysm.meshes])
L = ctx.assemble_system(N, xy, tri, sysm.mesh_offsets, sysm._tri_offsets, sig, rows, cols, vals)
imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
A = L.reduce(imap, nv - 1, -1.0)
keep = np.flatnonzero(imap[:nv] >= 0)
b = -rhs[keep]
bd = ctx.to_device(b); xd = ctx.empty(nv - 1)
for k in range(3):
    t = time.perf_counter(); r = A.solve_spd_dev(bd, xd, precond="amg", rebuild=True); w = time.perf_counter() - t
    print(f"dev  : wall {w*1e3:7.1f} ms  setup {r.setup_seconds*1e3:6.1f}  solve {r.seconds*1e3:6.1f}  it {r.iterations} restarts {r.restarts}", flush=True)
for k in range(3):
    t = time.perf_counter(); r = A.solve_spd(b, precond="amg", rebuild=True); w = time.perf_counter() - t
    print(f"host : wall {w*1e3:7.1f} ms  setup {r.setup_seconds*1e3:6.1f}  solve {r.seconds*1e3:6.1f}  it {r.iterations} restarts {r.restarts}", flush=True)
print("a new reduced matrix for every solve (what solve_system does):", flush=True)
A.close()
for k in range(4):
    t = time.perf_counter(); A = L.reduce(imap, nv - 1, -1.0); ctx.synchronize(); tr = time.perf_counter() - t
    t = time.perf_counter(); r = A.solve_spd_dev(bd, xd, precond="amg"); w = time.perf_counter() - t
    print(f"new A: reduce {tr*1e3:5.1f} ms  wall {w*1e3:7.1f} ms  setup {r.setup_seconds*1e3:6.1f}  solve {r.seconds*1e3:6.1f}  it {r.iterations}", flush=True)
    A.close()
print("the device calls of solve_system in its order:", flush=True)
v = np.zeros(N); rfull = rhs.copy()
for variant in ("reduce+solve", "+matvec", "+residual_norm"):
    for k in range(3):
        A = L.reduce(imap, nv - 1, -1.0)
        t = time.perf_counter(); r = A.solve_spd(b, precond="amg"); w = time.perf_counter() - t
        if variant != "reduce+solve":
            v[keep] = r.x
            L.matvec(v)
        if variant == "+residual_norm":
            L.residual_norm(v, rfull)
        A.close()
        print(f"{variant:16s}: wall {w*1e3:7.1f} ms  setup {r.setup_seconds*1e3:6.1f}  solve {r.seconds*1e3:6.1f}  it {r.iterations}", flush=True)
print("a threaded BLAS call (np.linalg.norm of 10 M doubles) right before the solve:", flush=True)
A = L.reduce(imap, nv - 1, -1.0)
for k in range(4):
    nb = np.linalg.norm(b)
    t = time.perf_counter(); r = A.solve_spd_dev(bd, xd, precond="amg", rebuild=True); w = time.perf_counter() - t
    print(f"after norm: wall {w*1e3:7.1f} ms  setup {r.setup_seconds*1e3:6.1f}  solve {r.seconds*1e3:6.1f}  it {r.iterations}", flush=True)
