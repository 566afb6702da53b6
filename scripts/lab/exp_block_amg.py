"""Single-GPU emulation of the multi-GPU preconditioner: CG on the full C4 matrix, multigrid built on the
matrix with the inter-layer (via) couplings removed = block-Jacobi across 8 layer ranks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from padne_amd import _hip, synthetic

ctx = _hip.Context(0)
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
sysm = synthetic.config(name) if not name.startswith("L") else synthetic.layered_system(*[int(t) for t in name[1:].split("x")][:2], int(name[1:].split("x")[1]), via_lattice=int(name[1:].split("x")[2]))
nv = sysm.n_vertices; N = nv + 1
xy, tri, mvo, mto, sig = bench.flat(sysm); rows, cols, vals, rhs = bench.stamps_of(sysm, N)
imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
L = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
A = L.reduce(imap, nv - 1, -1.0); L.close()
diag_only = rows == cols
Lb = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows[diag_only | (rows == N - 1) | (cols == N - 1)], cols[diag_only | (rows == N - 1) | (cols == N - 1)], vals[diag_only | (rows == N - 1) | (cols == N - 1)])
Ab = Lb.reduce(imap, nv - 1, -1.0); Lb.close()
keep = np.flatnonzero(imap[:nv] >= 0)
b = ctx.to_device(-rhs[keep]); x = ctx.empty(A.shape[0])
r = A.solve_spd_dev(b, x, precond="amg"); print(f"full AMG        : iters={r.iterations} setup={r.setup_seconds*1e3:.0f} ms solve={r.seconds*1e3:.0f} ms", flush=True)
A.set_preconditioner_block(Ab)
r = A.solve_spd_dev(b, x, precond="amg", raise_on_fail=False); print(f"layer-block AMG : iters={r.iterations} setup={r.setup_seconds*1e3:.0f} ms solve={r.seconds*1e3:.0f} ms relres={r.rel_residual:.2e}", flush=True)
