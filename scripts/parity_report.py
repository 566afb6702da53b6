#!/usr/bin/env python3
"""What the product path actually achieves on every golden fixture (GPU box): potential error vs the reference's
direct solve, source/ground current error, residual norm of the original system.  Used to set measured test bounds."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from padne_amd import solver  # noqa: E402
from oracle import padne_oracle as O  # noqa: E402

out = {}
for name in H.golden_names():
    g = H.load_golden(name)
    meshes, sig, stamps, r, n_pot = H.product_system(g)
    L = solver.assemble_from_arrays(meshes, sig, stamps, n_pot)
    v, info = solver.solve_system(L, r)
    v_chk = O.solve_system(L.tocsr(), r)[0]
    sp_ = max(np.abs(g["v"][:n_pot]).max(), 1e-300)
    sc = max(np.abs(g["v"][n_pot:]).max(), 1e-300)
    out[name] = dict(pot_rel_vs_checker_same_matrix=float(np.abs(v[:n_pot] - v_chk[:n_pot]).max() / max(np.abs(v_chk[:n_pot]).max(), 1e-300)), pot_rel=float(np.abs(v[:n_pot] - g["v"][:n_pot]).max() / sp_),
                     cur_abs=float(np.abs(v[n_pot:] - g["v"][n_pot:]).max()), cur_scale=float(sc),
                     residual=float(info.residual_norm), ref_residual=float(g["residual_norm"]),
                     gc=float(info.ground_node_current), ref_gc=float(g["ground_node_current"]), it=int(info.iterations))
    L.dev.close()
    print(name, json.dumps(out[name]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity_report.json"), "w"), indent=1)
