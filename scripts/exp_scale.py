"""Larger-than-headline systems on one MI355X: the C4 geometry at 40 M and 80 M unknowns (does everything still fit in
32-bit indices, how do setup / iterations / SpMV rate scale).  python scripts/exp_scale.py [nx ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from padne_amd import _hip, synthetic

ctx = _hip.Context(0)
out = {}
for nx in [int(a) for a in (sys.argv[1:] or ["2236", "3162"])]:
    t = time.perf_counter(); sysm = synthetic.layered_system(8, nx, nx, name=f"8-layer {nx}x{nx}"); t_gen = time.perf_counter() - t
    nv = sysm.n_vertices; N = nv + 1
    print(f"generated N={nv} in {t_gen:.1f} s", flush=True)
    xy, tri, mvo, mto, sig = bench.flat(sysm); rows, cols, vals, rhs = bench.stamps_of(sysm, N)
    t = time.perf_counter(); L = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals); ctx.synchronize(); t_asm = time.perf_counter() - t
    imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
    t = time.perf_counter(); A = L.reduce(imap, nv - 1, -1.0); ctx.synchronize(); t_red = time.perf_counter() - t
    L.close(); del xy, tri
    keep = np.flatnonzero(imap[:nv] >= 0)
    b = ctx.to_device(-rhs[keep]); x = ctx.empty(nv - 1)
    xr = ctx.to_device(np.random.default_rng(1).uniform(-1, 1, A.shape[1])); y = ctx.empty(A.shape[0])
    t_spmv = min(A.spmv_time(xr, y, 3, 20) for _ in range(3))
    best = None
    for _ in range(2):
        t = time.perf_counter(); r = A.solve_spd_dev(b, x, precond="amg", rebuild=True); w = time.perf_counter() - t
        if best is None or w < best[0]: best = (w, r)
    w, r = best
    rec = {"n": A.shape[0], "nnz": A.nnz, "assemble_s_incl_h2d": t_asm, "reduce_s": t_red, "spmv_us": t_spmv * 1e6,
           "spmv_frac_of_8TBs": A.spmv_bytes / t_spmv / 8e12, "iterations": r.iterations, "levels": r.levels,
           "operator_complexity": r.operator_complexity, "setup_ms": r.setup_seconds * 1e3, "solve_ms": r.seconds * 1e3,
           "wall_ms": w * 1e3, "rel_residual": r.rel_residual}
    out[str(nx)] = rec
    print(json.dumps(rec), flush=True)
    del A, b, x, xr, y, L
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", os.environ.get("PADNE_ROUND", "r02") + "_scale.json"), "w"), indent=1)
