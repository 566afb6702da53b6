"""Measure every BASELINE.json config that fits one GPU (C2, C3, C4, C5) and write gpurun_out/configs.json (scripts/collect_profiles.py copies it to profiles/<round>_configs.json)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from padne_amd import _hip, synthetic

ctx = _hip.Context(0)
out = {}
for name in ["C2", "C3", "C4", "C5"]:
    sysm = synthetic.config(name); nv = sysm.n_vertices; N = nv + 1
    xy, tri, mvo, mto, sig = bench.flat(sysm); rows, cols, vals, rhs = bench.stamps_of(sysm, N)
    t = time.perf_counter(); L = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals); ctx.synchronize(); t_asm = time.perf_counter() - t
    # the same with the meshes generated on the device (no host arrays, no PCIe): generation, then assembly
    t = time.perf_counter(); sd, xy_d, tri_d = synthetic.config_on_device(ctx, name); ctx.synchronize(); t_gen = time.perf_counter() - t
    t_dev = 1e9
    for _ in range(3):
        t = time.perf_counter(); Ld = ctx.assemble_system(N, xy_d, tri_d, sd.mesh_offsets, sd._tri_offsets, sig, rows, cols, vals); ctx.synchronize()
        t_dev = min(t_dev, time.perf_counter() - t); Ld.close()
    xy_d.free(); tri_d.free(); del xy, tri
    imap = np.arange(N, dtype=np.int32); imap[sysm.ground] = -1; imap[imap > sysm.ground] -= 1; imap[N - 1] = -1
    t = time.perf_counter(); A = L.reduce(imap, nv - 1, -1.0); ctx.synchronize(); t_red = time.perf_counter() - t
    L.close()
    keep = np.flatnonzero(imap[:nv] >= 0)
    xr = ctx.to_device(np.random.default_rng(1).uniform(-1, 1, A.shape[1])); y = ctx.empty(A.shape[0])
    t_spmv = min(A.spmv_time(xr, y, 5, 50) for _ in range(3))
    rec = {"n": A.shape[0], "nnz": A.nnz, "assemble_s_incl_h2d": t_asm, "assemble_s_device_resident": t_dev,
           "generate_on_device_s": t_gen, "assembly_algorithmic_gbs": 124.0 * nv / t_dev / 1e9, "reduce_s": t_red,
           "spmv_us": t_spmv * 1e6, "spmv_gbs": A.spmv_bytes / t_spmv / 1e9, "spmv_frac_of_8TBs": A.spmv_bytes / t_spmv / 8e12}
    if name == "C5":
        f, tt = synthetic.multi_rhs_pairs(sysm, 8)
        B = np.zeros((8, nv - 1))
        for k in range(8):
            full = np.zeros(nv); full[f[k]] += 1.0; full[tt[k]] -= 1.0
            B[k] = full[keep]
        b = ctx.to_device(B); x = ctx.empty((8, nv - 1))
        A.solve_spd_dev(b, x, n_rhs=8, precond="amg", rebuild=True)        # warm the allocator
        t = time.perf_counter(); r = A.solve_spd_dev(b, x, n_rhs=8, precond="amg", rebuild=True); w = time.perf_counter() - t
        xb = x.numpy()
        rec.update({"n_rhs": 8, "iterations_total": r.iterations, "setup_ms": r.setup_seconds * 1e3, "solve_ms_all_rhs": r.seconds * 1e3,
                    "wall_ms": w * 1e3, "rel_residual_max": r.rel_residual,
                    "note": "8 right-hand sides advanced in lockstep (SpMM, one pass over the operators per iteration)"})
        xs8 = ctx.to_device(np.random.default_rng(2).uniform(-1, 1, A.shape[1] * 8)); ys8 = ctx.empty(A.shape[0] * 8)
        t8 = min(A.spmm8_time(xs8, ys8, 5, 30) for _ in range(3))
        rec.update({"spmm8_us": t8 * 1e6, "spmm8_gbs_algorithmic": A.spmm8_bytes / t8 / 1e9, "spmm8_vs_8_spmv": 8 * t_spmv / t8})
        os.environ["PADNE_NO_BATCH"] = "1"; ctx.reload_options()
        t = time.perf_counter(); r1 = A.solve_spd_dev(b, x, n_rhs=8, precond="amg", rebuild=True); w1 = time.perf_counter() - t
        del os.environ["PADNE_NO_BATCH"]; ctx.reload_options()
        rec.update({"one_at_a_time": {"iterations_total": r1.iterations, "solve_ms_all_rhs": r1.seconds * 1e3, "wall_ms": w1 * 1e3},
                    "batched_vs_one_at_a_time_max_rel_diff": float(np.abs(xb - x.numpy()).max() / np.abs(xb).max())})
        del xs8, ys8
    else:
        b = ctx.to_device(-rhs[keep]); x = ctx.empty(nv - 1)
        best = None
        for _ in range(3):
            t = time.perf_counter(); r = A.solve_spd_dev(b, x, precond="amg", rebuild=True); w = time.perf_counter() - t
            if best is None or w < best[0]: best = (w, r)
        w, r = best
        rec.update({"iterations": r.iterations, "levels": r.levels, "operator_complexity": r.operator_complexity,
                    "setup_ms": r.setup_seconds * 1e3, "solve_ms": r.seconds * 1e3, "wall_ms": w * 1e3, "solves_per_s": 1.0 / w,
                    "rel_residual": r.rel_residual})
        if name != "C4":
            xa = x.numpy()
            rj = A.solve_spd_dev(b, x, precond="jacobi")
            rec.update({"jacobi_iterations": rj.iterations, "jacobi_solve_ms": rj.seconds * 1e3,
                        "amg_vs_jacobi_max_rel_diff": float(np.abs(xa - x.numpy()).max() / np.abs(xa).max())})
    out[name] = rec
    print(name, json.dumps(rec), flush=True)
    del A, b, x, xr, y
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "configs.json"), "w"), indent=1)
