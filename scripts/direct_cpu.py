#!/usr/bin/env python3
"""Direct-solve fixture of a BASELINE config WITHOUT any device code: oracle assembly + the reference's solve call.

    python scripts/direct_cpu.py [--config C2] [--out tests/golden/direct_C2.npz]

``scripts/direct_full.py`` (C3, C4) solves the DEVICE-assembled matrix on both sides, so its fixtures pin the solve but
not the assembly kernels at full size (missing item 4).  Here the un-reduced system comes from the CPU
oracle (``oracle.padne_oracle.assemble_system``: ``mesh.py:124-139``, ``solver.py:171-213, 563-575, 469-560``
restated and pinned against the reference's goldens) and is solved exactly as the reference does (``solver.py:772-775``:
``tocsc``, ``scipy.sparse.linalg.spsolve``, residual).  4100 sampled potentials + the ground current are committed; the
``-m gpu`` test then assembles the same config on the device, compares the matrix BIT FOR BIT with the oracle's and the
product's potentials with these samples at 1e-8.  Runs on any host (no GPU): C2 takes about a minute of one core.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def oracle_system(name: str):
    """(L_csr, r, n_vertices) of a named config, assembled by the oracle."""
    from oracle import padne_oracle as O
    from padne_amd import synthetic
    sysm = synthetic.config(name)
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
    els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    L, r = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, sysm.ground)
    return L, r, sysm.n_vertices


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    out = args.out or os.path.join(ROOT, "tests", "golden", f"direct_{args.config}.npz")
    import scipy.sparse.linalg as spla
    t0 = time.perf_counter()
    L, r, nv = oracle_system(args.config)
    t_asm = time.perf_counter() - t0
    print(f"[direct_cpu] {args.config}: oracle assembly {t_asm:.1f} s, N={L.shape[0]} nnz={L.nnz}", flush=True)
    t0 = time.perf_counter()
    L_csc = L.tocsc()
    v = spla.spsolve(L_csc, r)
    res = float(np.linalg.norm(L_csc @ v - r))
    t_lu = time.perf_counter() - t0
    scale = float(np.abs(v[:nv]).max())
    pick = np.unique(np.concatenate([np.random.default_rng(2026).integers(0, nv, 4096),
                                     [int(np.argmax(v[:nv])), int(np.argmin(v[:nv])), 0, nv - 1]]))
    np.savez_compressed(out, index=pick.astype(np.int64), potential=v[pick], max_abs_potential=np.float64(scale),
                        n_vertices=np.int64(nv), residual_norm=np.float64(res), spsolve_seconds=np.float64(t_lu),
                        ground_node_current=np.float64(v[-1]), nnz=np.int64(L.nnz),
                        assembled_by=np.array("oracle.padne_oracle.assemble_system"))
    print(f"[direct_cpu] {args.config}: tocsc+spsolve+residual {t_lu:.1f} s, residual {res:.2e}, "
          f"max |v| {scale:.6g}, ground current {v[-1]:.2e} -> {out}", flush=True)


if __name__ == "__main__":
    main()
