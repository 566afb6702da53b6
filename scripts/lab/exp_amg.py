"""GPU experiment: multigrid-preconditioned CG vs Jacobi-PCG on the synthetic configs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from padne_amd import _hip, synthetic
from oracle import padne_oracle as O

ctx = _hip.Context(0)
for name in sys.argv[1:] or ["S", "C2", "C4"]:
    if name == "S":
        sysm = synthetic.layered_system(2, 200, 200, via_lattice=6)
    elif name.startswith("L"):          # L<layers>x<nx>x<lattice>
        nl_, nx_, lat_ = (int(t) for t in name[1:].split("x"))
        sysm = synthetic.layered_system(nl_, nx_, nx_, via_lattice=lat_)
    else:
        sysm = synthetic.config(name)
    nv = sysm.n_vertices; N = nv + 1
    xy, tri, mvo, mto, sig = bench.flat(sysm); rows, cols, vals, rhs = bench.stamps_of(sysm, N)
    L = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
    imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
    A = L.reduce(imap, nv - 1, -1.0); L.close()
    keep = np.flatnonzero(imap[:nv] >= 0)
    b = ctx.to_device(-rhs[keep]); x = ctx.empty(A.shape[0])
    t = time.time(); r = A.solve_spd_dev(b, x, rtol=1e-12, precond="amg", raise_on_fail=False); w = time.time() - t
    print(f"[{name}] n={A.shape[0]} AMG: iters={r.iterations} restarts={r.restarts} relres={r.rel_residual:.2e} status={r.status} "
          f"levels={r.levels} cx={r.operator_complexity:.2f} setup={r.setup_seconds*1e3:.1f} ms solve={r.seconds*1e3:.1f} ms wall={w*1e3:.1f} ms "
          f"ms/iter={r.seconds/max(r.iterations,1)*1e3:.3f}", flush=True)
    xa = x.numpy()
    for rep in range(2):
        r2 = A.solve_spd_dev(b, x, rtol=1e-12, precond="amg", raise_on_fail=False, time_spmv=True)
        print(f"      cached hierarchy: iters={r2.iterations} solve={r2.seconds*1e3:.1f} ms spmv={r2.spmv_seconds*1e6:.1f} us", flush=True)
    if A.shape[0] < 3e6:
        r3 = A.solve_spd_dev(b, x, rtol=1e-12, precond="jacobi", raise_on_fail=False)
        xj = x.numpy()
        print(f"      Jacobi: iters={r3.iterations} solve={r3.seconds*1e3:.1f} ms ; max|x_amg-x_jac|/max|x| = {np.abs(xa-xj).max()/np.abs(xj).max():.2e}", flush=True)
    if A.shape[0] < 2e5:
        els = [("R", int(a_), int(b_), float(r_)) for a_, b_, r_ in zip(*sysm.resistors)] + [("I", int(f), int(t_), float(i)) for f, t_, i in zip(*sysm.current_sources)]
        Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, 0)
        v = O.solve_system(Lo, ro)[0]
        print(f"      vs spsolve: {np.abs(xa - v[keep]).max()/np.abs(v[:nv]).max():.2e}", flush=True)
    del A, b, x
