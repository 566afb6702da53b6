#!/usr/bin/env python3
"""Full-size check of the north-star bar: HIP potentials vs the reference's direct solve, at the sizes of BASELINE.json.

    python scripts/direct_full.py [--configs C3,C4] [--out gpurun_out/r02_direct_full.json] [--cap-seconds 900]

For every config the SAME un-reduced system (mesh Laplacian in the reference's sign convention, via stamps, ground
row / column: what ``assemble_system`` returns, solver.py:783-812) is

  1. assembled on the device and solved by the product path (``padne_amd.solver.solve_system``),
  2. downloaded and solved on this host exactly as the reference does (solver.py:772-775):
     ``L.tocsc()``, ``scipy.sparse.linalg.spsolve(L_csc, r)``, ``norm(L_csc @ v - r)``, timed with perf_counter
     (SuperLU of the scipy wheel: one thread),

and the two potential vectors are compared: max |v_hip - v_lu| / max |v_lu|.  The record (seconds, peak RSS,
errors, residuals) is rewritten after every stage so a run that hits the wall cap still leaves what it measured.
BASELINE.md section 4.4; VERDICT r01 "next" 1(b).  scipy is the third-party home of the reference's solve
(pyproject.toml:46); no reference source is needed or read here.
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import resource
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rss_gb() -> float:
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def heartbeat(stop: threading.Event, label: str):
    t0 = time.perf_counter()
    while not stop.wait(45.0):
        print(f"[direct_full] {label}: {time.perf_counter() - t0:6.0f} s, peak RSS {rss_gb():.1f} GB", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C3,C4")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r02_direct_full.json"))
    args = ap.parse_args()

    import scipy
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from padne_amd import solver, synthetic
    from padne_amd.reduction import Constraint, KKTLayout
    sys.path.insert(0, ROOT)
    from bench import flat, stamps_of

    record = {"host": {"cpu": cpu_model(), "cores_available": os.cpu_count(), "cores_used": 1,
                       "scipy": scipy.__version__, "numpy": np.__version__},
              "what": "reference solve step (tocsc + spsolve + residual, solver.py:772-775) vs padne_amd.solver.solve_system "
                      "on the same assembled system", "configs": {}}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)

    def dump():
        with open(args.out, "w") as fh:
            json.dump(record, fh, indent=1)

    ctx = solver.get_context()
    for name in args.configs.split(","):
        sysm = synthetic.config(name)
        nv = sysm.n_vertices
        N = nv + 1
        xy, tri, mvo, mto, sig = flat(sysm)
        rows, cols, vals, rhs = stamps_of(sysm, N)
        t0 = time.perf_counter()
        Ld = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
        ctx.synchronize()
        t_asm = time.perf_counter() - t0
        layout = KKTLayout(size=N, n_potential=nv,
                           constraints=[Constraint(index=N - 1, p=int(sysm.ground), n=-1, value=0.0)])
        L = solver.SystemMatrix(Ld, layout)
        solver.solve_system(L, rhs)                      # warm-up (allocator pool, first-call costs)
        t0 = time.perf_counter()
        v_hip, info = solver.solve_system(L, rhs)
        t_hip = time.perf_counter() - t0
        rec = {"n": int(N), "nnz": int(Ld.nnz), "hip": {
            "assemble_seconds_incl_h2d": t_asm, "solve_system_seconds": t_hip, "device_solve_seconds": info.solve_seconds,
            "iterations": info.iterations, "rel_residual": info.rel_residual, "residual_norm": info.residual_norm,
            "ground_node_current": info.ground_node_current}}
        record["configs"][name] = rec
        dump()
        print(f"[direct_full] {name}: N={N} nnz={Ld.nnz} hip solve_system {t_hip * 1e3:.1f} ms "
              f"(device {info.solve_seconds * 1e3:.1f} ms, {info.iterations} it), residual {info.residual_norm:.2e}", flush=True)

        Lh = Ld.to_scipy()
        Ld.close()
        del xy, tri
        stop = threading.Event()
        hb = threading.Thread(target=heartbeat, args=(stop, f"{name} spsolve"), daemon=True)
        hb.start()
        try:
            t0 = time.perf_counter()
            L_csc = Lh.tocsc()
            t_csc = time.perf_counter() - t0
            t1 = time.perf_counter()
            v_lu = spla.spsolve(L_csc, rhs)
            t_lu = time.perf_counter() - t1
            t2 = time.perf_counter()
            res_lu = float(np.linalg.norm(L_csc @ v_lu - rhs))
            t_res = time.perf_counter() - t2
        finally:
            stop.set()
        total = t_csc + t_lu + t_res
        scale = float(np.abs(v_lu[:nv]).max())
        err = float(np.abs(v_hip[:nv] - v_lu[:nv]).max() / scale)
        # the solution is determined up to rounding relative to its own spread
        spread = float(v_lu[:nv].max() - v_lu[:nv].min())
        rec["reference_cpu"] = {"tocsc_seconds": t_csc, "spsolve_seconds": t_lu, "residual_seconds": t_res,
                                "total_seconds": total, "residual_norm": res_lu, "peak_rss_gb": rss_gb(),
                                "ground_node_current": float(v_lu[-1])}
        rec["parity"] = {"max_abs_diff": float(np.abs(v_hip[:nv] - v_lu[:nv]).max()), "max_abs_potential": scale,
                         "potential_spread": spread, "max_rel_error": err, "max_error_over_spread": float(np.abs(v_hip[:nv] - v_lu[:nv]).max() / spread),
                         "ground_current_abs_diff": abs(float(v_hip[-1]) - float(v_lu[-1])), "bar": 1e-8,
                         "within_bar": bool(err <= 1e-8)}
        rec["speedup"] = {"vs_device_solve": total / info.solve_seconds, "vs_solve_system_wall": total / t_hip}
        # a sample of the direct solve's potentials becomes a committed fixture (tests/golden/direct_<config>.npz): the
        # full-size GPU test then checks the product against the reference's own solve call at BASELINE scale
        pick = np.unique(np.concatenate([np.random.default_rng(2026).integers(0, nv, 4096),
                                         [int(np.argmax(v_lu[:nv])), int(np.argmin(v_lu[:nv])), 0, nv - 1]]))
        np.savez_compressed(os.path.join(os.path.dirname(args.out), f"direct_{name}.npz"), index=pick.astype(np.int64),
                            potential=v_lu[pick], max_abs_potential=np.float64(scale), n_vertices=np.int64(nv),
                            residual_norm=np.float64(res_lu), spsolve_seconds=np.float64(t_lu))
        # scipy CSR product on this host, same byte formula as the GPU figure
        x = np.random.default_rng(1).uniform(-1, 1, N)
        best = min(_t(lambda: Lh @ x) for _ in range(5))
        rec["reference_cpu"]["csr_spmv_seconds"] = best
        rec["reference_cpu"]["csr_spmv_gbs"] = (12 * Lh.nnz + 20 * N + 4) / best / 1e9
        dump()
        print(f"[direct_full] {name}: spsolve {t_lu:.1f} s (+tocsc {t_csc:.1f} s, residual {t_res:.2f} s), residual {res_lu:.2e}, "
              f"peak RSS {rss_gb():.1f} GB; max rel potential error {err:.2e}; speed-up {total / info.solve_seconds:.0f}x "
              f"(device solve) / {total / t_hip:.0f}x (solve_system wall)", flush=True)
        del Lh, L_csc, v_lu
    dump()
    print(json.dumps(record))


def _t(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


if __name__ == "__main__":
    main()
