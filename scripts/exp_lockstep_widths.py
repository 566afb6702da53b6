"""What a lockstep iteration of width 2 / 4 / 8 costs against single solves, config C3 (N = 5 M): n right-hand sides
(distinct source / sink pairs) solved through padne_solve_spd_dev with the hierarchy in place, narrow widths on and off.
Writes gpurun_out/lockstep_widths.json (-> profiles/r04_lockstep_widths.json)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from padne_amd import _hip, synthetic

ctx = _hip.Context(0)
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
sysm, xy, tri = synthetic.config_on_device(ctx, name)
nv = sysm.n_vertices; N = nv + 1
rows, cols, vals, rhs = bench.stamps_of(sysm, N)
L = ctx.assemble_system(N, xy, tri, sysm.mesh_offsets, sysm._tri_offsets, np.array([m[2] for m in sysm.meshes]), rows, cols, vals)
imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
A = L.reduce(imap, nv - 1, -1.0); L.close()
keep = np.flatnonzero(imap[:nv] >= 0)
f, tt = synthetic.multi_rhs_pairs(sysm, 8)
B = np.zeros((8, nv - 1))
for k in range(8):
    full = np.zeros(nv); full[f[k]] += 1.0; full[tt[k]] -= 1.0
    B[k] = full[keep]
out = {"config": name, "n": int(A.shape[0])}
x1 = ctx.empty(nv - 1)
b1 = ctx.to_device(B[0])
A.solve_spd_dev(b1, x1, precond="amg", rebuild=True)
single = min(A.solve_spd_dev(b1, x1, precond="amg").seconds for _ in range(3))
r1 = A.solve_spd_dev(b1, x1, precond="amg")
out["single"] = {"solve_ms": single * 1e3, "iterations": int(r1.iterations)}
for n_rhs in (2, 3, 4, 8):
    b = ctx.to_device(B[:n_rhs].copy()); x = ctx.empty((n_rhs, nv - 1))
    rec = {}
    for label, env in (("lockstep", None), ("one_at_a_time", "1")):
        if env: os.environ["PADNE_NO_BATCH"] = env; ctx.reload_options()
        A.solve_spd_dev(b, x, n_rhs=n_rhs, precond="amg")
        best = None
        for _ in range(3):
            r = A.solve_spd_dev(b, x, n_rhs=n_rhs, precond="amg")
            if best is None or r.seconds < best.seconds: best = r
        os.environ.pop("PADNE_NO_BATCH", None); ctx.reload_options()
        rec[label] = {"solve_ms": best.seconds * 1e3, "iterations_total": int(best.iterations), "in_single_solves": best.seconds / single}
        sol = x.numpy().copy()
        if label == "lockstep": ref = sol
        else: rec["max_rel_difference"] = float(np.abs(ref - sol).max() / np.abs(sol).max())
    out[f"n_rhs_{n_rhs}"] = rec
    print(n_rhs, json.dumps(rec), flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "lockstep_widths.json"), "w"), indent=1)
