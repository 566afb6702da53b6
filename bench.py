#!/usr/bin/env python3
"""Headline benchmark: CG solves/s (and SpMV GB/s vs the HBM roofline) on the synthetic layered
Laplacian of BASELINE.json, MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one complete solve of the reduced SPD system (the work of the reference's
``solve_system``, solver.py:767-780, which factorises and solves): build the multigrid hierarchy
(the counterpart of the factorisation; it is rebuilt inside EVERY timed step), then preconditioned CG
from x0 = 0 to ||b - A x|| <= 1e-12 ||b||, including the final true-residual check, with the matrix
and the right-hand side already resident in HBM.  Workload at every GPU count: config C4 of SURVEY.md section 8d -- 8 copper layers of
1118 x 1118 vertices (N = 10M unknowns, ~70M non-zeros) stitched by via rings; with N GPUs the
layers are dealt to the ranks (strong scaling: the problem is fixed, ``scaling: "strong"``).

One JSON line is printed by rank 0 (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md:36
RTOL = 1e-12
ASSEMBLY_REPEATS = 10          # warm repetitions of the device-resident assembly behind the cold call


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C4", help="C2 | C3 | C4 (default, the headline config)")
    ap.add_argument("--precond", default="amg", choices=["amg", "jacobi"],
                    help="amg: smoothed-aggregation multigrid V-cycle (rebuilt inside every step); jacobi: diagonal")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-seam", action="store_true",
                    help="skip the solve_system(L, r) wall-time measurement behind the timed steps (profiling runs)")
    ap.add_argument("--no-rank-proxy", action="store_true",
                    help="skip the one-rank-of-eight measurement (one layer of C4 through the row-partitioned path)")
    ap.add_argument("--no-c5", action="store_true", help="skip the batched right-hand sides of config C5")
    ap.add_argument("--no-launch-count", action="store_true",
                    help="skip the three extra solves that count the library's launches per step / setup / iteration (counter runs)")
    ap.add_argument("--no-small", action="store_true", help="skip solve_system vs spsolve at 11 k / 100 k / 1 M unknowns")
    ap.add_argument("--no-dist-one-rank", action="store_true",
                    help="skip config C4 through the row-partitioned path with a one-rank RCCL communicator")
    ap.add_argument("--small-cpu-limit", type=int, default=300000,
                    help="largest system whose CPU spsolve is timed live in the `small` block (above: profiles/r06_small.json)")
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the row-partitioned code path (RCCL communicator, halo plan) even on 1 GPU")
    ap.add_argument("--cpu-sample-nx", type=int, default=400,
                    help="grid edge of the bounded CPU-baseline sample (8 layers of nx*nx)")
    return ap.parse_args()


def stamps_of(sysm, N):
    """Lumped COO stamps + rhs of a SyntheticSystem, in the reference's stamp order (solver.py:475-492, 558-560)."""
    a, b, r = sysm.resistors
    g = 1.0 / r
    rows = np.stack([a, a, b, b], 1).reshape(-1)
    cols = np.stack([a, b, b, a], 1).reshape(-1)
    vals = np.stack([-g, g, -g, g], 1).reshape(-1)
    gi = sysm.ground
    rows = np.concatenate([rows, [N - 1, gi]])
    cols = np.concatenate([cols, [gi, N - 1]])
    vals = np.concatenate([vals, [1.0, 1.0]])
    rhs = np.zeros(N)
    f, t, i = sysm.current_sources
    np.add.at(rhs, f, i)
    np.add.at(rhs, t, -i)
    return rows, cols, vals, rhs


def flat(sysm):
    xy = np.concatenate([m[0] for m in sysm.meshes])
    tri = np.concatenate([m[1] for m in sysm.meshes])
    mvo = sysm.mesh_offsets
    mto = np.concatenate([[0], np.cumsum([m[1].shape[0] for m in sysm.meshes])]).astype(np.int64)
    sig = np.array([m[2] for m in sysm.meshes])
    return xy, tri, mvo, mto, sig


def cpu_baseline(nx: int):
    """The reference's solve step (tocsc + spsolve + residual, solver.py:772-775) on a bounded sample
    of the same workload, timed on this host."""
    from oracle import padne_oracle as O
    from padne_amd import synthetic
    lattice = max(2, int(round(32 * nx / 1118)))
    sysm = synthetic.layered_system(8, nx, nx, via_lattice=lattice)
    els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
    els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
    L, r = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, sysm.ground)
    t0 = time.perf_counter()
    v, gc, res = O.solve_system(L, r)
    dt = time.perf_counter() - t0
    n = L.shape[0]
    # scipy CSR SpMV on the same matrix
    x = np.random.default_rng(1).uniform(-1, 1, n)
    best = min(_time(lambda: L @ x) for _ in range(5))
    spmv_gbs = (12 * L.nnz + 20 * n + 4) / best / 1e9
    n_full = 8 * 1118 * 1118
    return {
        "value": 1.0 / dt, "unit": "solves/s", "cores": 1, "kind": "port",
        "sample": f"8-layer {nx}x{nx} via-stitched Laplacian, N={n} (1/{n_full // n} of the workload): "
                  f"oracle assembly, then the reference's tocsc+spsolve+residual (solver.py:772-775), "
                  f"{dt:.2f} s, residual {res:.1e}; SuperLU is single-threaded ({os.cpu_count()} cores available)",
        "seconds": dt, "n": n, "cpu_spmv_gbs": spmv_gbs,
        "full_size": full_size_cpu_record(),
    }


def full_size_cpu_record():
    """The reference's solve call on the WHOLE workload, measured once per round on a GPU box host (scripts/direct_full.py ->
    profiles/<round>_direct_full.json; 11 minutes of one core, far beyond the bounded sample this bench may time)."""
    for rd in ("r06", "r05", "r02"):
        path = os.path.join(ROOT, "profiles", rd + "_direct_full.json")
        try:
            doc = json.load(open(path))
            rec = doc["configs"]["C4"]
            return {"static_from": f"profiles/{rd}_direct_full.json",
                    "source": f"profiles/{rd}_direct_full.json (scripts/direct_full.py, measured once in round {int(rd[1:])} on a GPU-box "
                              "host on that round's device-assembled matrix, not extrapolated, NOT re-measured in this run)",
                    "cpu": doc["host"]["cpu"], "n": rec["n"],
                    "spsolve_seconds": rec["reference_cpu"]["spsolve_seconds"], "total_seconds": rec["reference_cpu"]["total_seconds"],
                    "peak_rss_gb": rec["reference_cpu"]["peak_rss_gb"],
                    "max_rel_potential_error_of_the_hip_solve": rec["parity"]["max_rel_error"]}
        except Exception:
            continue
    return None


def step_counter_traffic():
    """HBM-side bytes of one timed step by the hardware counters (profiles/<round>_step_traffic.json, written by
    scripts/pmc_setup_sum.py from separate rocprofv3 --pmc passes of the same command; the newest round that has one);
    (None, None) if there is none."""
    for rd in ("r06", "r05", "r04", "r03"):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", rd + "_step_traffic.json"))), f"profiles/{rd}_step_traffic.json"
        except Exception:
            continue
    return None, None


def step_algorithmic_bytes(shapes, iterations: int):
    """Bytes a whole step has to move at least, by the textbook CSR accounting of SURVEY.md section 8d (f64 values, i32
    indices; 12 B per non-zero + 20 B per row for a product, x counted once): the multigrid setup reads every level
    operator once and writes P, R = P^T and the next operator once; a CG iteration is the 232 N bytes of the textbook
    loop with z = M r replaced by one V(1,1) cycle (per level two products with A_l, one with P_l and R_l each, five
    vector passes).  An implementation that moves fewer real bytes (float cycle, 1-byte window positions) simply
    scores a higher fraction, as the survey says for the PCG figure."""
    def csr(t):                       # store / stream a matrix once
        return 12 * t[2] + 4 * (t[0] + 1)

    def product(t):                   # y = M x
        return 12 * t[2] + 4 * (t[0] + 1) + 8 * t[0] + 8 * t[1]
    setup = 0
    cycle = 0
    for lv in shapes:
        setup += csr(lv["A"])                                        # read A_l (strength, smoothing, A P)
        if "P" in lv:
            setup += csr(lv["P"]) + csr(lv["R"])                     # write P, R
            cycle += 2 * product(lv["A"]) + product(lv["P"]) + product(lv["R"]) + 5 * 8 * lv["A"][0]
        else:
            setup += 8 * lv["A"][0] * lv["A"][0]                     # dense inverse of the coarsest operator
            cycle += 8 * lv["A"][0] * lv["A"][0]
    setup += sum(csr(lv["A"]) for lv in shapes[1:])                   # write A_1 ... A_L
    a0 = shapes[0]["A"]
    n = a0[0]
    cg = product(a0) + 16 * n + 24 * n + 24 * n + 16 * n + 24 * n     # q = A p, p.q, x, r, r.z / r.r, p  (232 N at 7 nnz/row)
    return {"setup": int(setup), "per_iteration": int(cg + cycle), "total": int(setup + iterations * (cg + cycle))}


def seam_timing(ctx, L_dev, sysm, rhs, repeats: int):
    """Wall time of the call the reference actually makes: ``solve_system(L, r)`` (solver.py:767-780) with r and v as
    host numpy arrays -- reduction plan, reduced matrix, multigrid hierarchy, CG, expansion, multiplier recovery and
    ||L v - r||, r up and v down over PCIe.  ``ms_per_solve``: everything rebuilt in every call (a new system each time,
    like a factorisation); ``ms_per_solve_cached_plan``: a further right-hand side on the same assembled system."""
    from padne_amd import solver
    from padne_amd.reduction import Constraint, KKTLayout
    solver.set_context(ctx)
    N = L_dev.shape[0]
    layout = KKTLayout(size=N, n_potential=N - 1, constraints=[Constraint(index=N - 1, p=int(sysm.ground), n=-1, value=0.0)])
    Ls = solver.SystemMatrix(L_dev, layout)

    def drop_plans():
        for plan in Ls._plans.values():
            plan.close()
        Ls._plans.clear()
    v, info = solver.solve_system(Ls, rhs)                      # warm-up: allocator pool, first-call costs
    cold, cached = [], []
    for _ in range(repeats):
        drop_plans()
        ctx.synchronize()
        t0 = time.perf_counter()
        v, info = solver.solve_system(Ls, rhs)
        cold.append(time.perf_counter() - t0)
    for _ in range(repeats):
        t0 = time.perf_counter()
        v, info_c = solver.solve_system(Ls, rhs)
        cached.append(time.perf_counter() - t0)
    drop_plans()
    return {"ms_per_solve": float(np.mean(cold)) * 1e3, "ms_per_solve_min": float(np.min(cold)) * 1e3,
            "ms_per_solve_cached_plan": float(np.mean(cached)) * 1e3, "repeats": repeats,
            "iterations": int(info.iterations), "residual_norm": float(info.residual_norm),
            "device_solve_ms": float(info.solve_seconds) * 1e3,
            "what": "padne_amd.solver.solve_system(L, r): host r in, host v out (PCIe inclusive), everything derived from L "
                    "rebuilt per call; never part of `value`"}


def rank_proxy(steps: int):
    """What ONE rank of an 8-GPU run has to do per solve, measured on this GPU: one layer of config C4 (1118 x 1118 vertices,
    a rank's share of the 10 M unknowns) through the ROW-PARTITIONED path -- the hierarchy of amg_setup_dist, the
    single-reduction CG loop, the collectives of an RCCL communicator of one rank -- next to the same layer through the
    one-GPU path.  No other rank exists, so nothing waits for a peer: ``ms_per_step`` is the part of an 8-GPU step that
    no communication is in, and ms_per_step(C4 on one GPU) / rank_proxy.ms_per_step is an UPPER BOUND of the 8-GPU speed-up
    (the exchanges, the gathered tail of the hierarchy and load imbalance only lower it).  A bound, not a scaling claim."""
    import torch
    import torch.distributed as dist
    from padne_amd import _hip, distributed, synthetic
    sysm = synthetic.layered_system(1, 1118, 1118, name="one layer of C4")
    nv = sysm.n_vertices
    N = nv + 1
    out = {"workload": f"one layer of config C4: {nv} vertices, 1 A source/sink, amg-PCG to rtol {RTOL:g}, hierarchy rebuilt "
                       "in every step", "rows": int(nv - 1)}

    def timed(solve, sync):
        solve(rebuild=True)
        solve(rebuild=True)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = solve(rebuild=True)
        sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        # launches: a whole step, a solve on the cached hierarchy, a shorter solve on it
        c0 = _hip.launch_count()
        full = solve(rebuild=True)
        sync()
        c1 = _hip.launch_count()
        cached = solve(rebuild=False)
        sync()
        c2 = _hip.launch_count()
        short = solve(rebuild=False, rtol=1e-5)
        sync()
        c3 = _hip.launch_count()
        d_it = max(int(cached.iterations) - int(short.iterations), 1)
        return {"ms_per_step": ms, "setup_ms": float(last.setup_seconds) * 1e3,
                "us_per_iteration": float(last.seconds) / max(int(last.iterations), 1) * 1e6, "iterations": int(last.iterations),
                "launches_per_step": int(c1 - c0), "launches_per_setup": int((c1 - c0) - (c2 - c1)),
                "launches_per_iteration": round(((c2 - c1) - (c3 - c2)) / d_it, 1)}

    # (a) the one-GPU path
    ctx1 = _hip.Context(0)
    xy, tri, mvo, mto, sig = flat(sysm)
    rows, cols, vals, rhs = stamps_of(sysm, N)
    L = ctx1.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
    imap = np.arange(N, dtype=np.int32)
    imap[sysm.ground] = -1
    imap[imap > sysm.ground] -= 1
    imap[N - 1] = -1
    A = L.reduce(imap, nv - 1, -1.0)
    L.close()
    keep = np.flatnonzero(imap[:nv] >= 0)
    b = ctx1.to_device(-rhs[keep])
    x = ctx1.empty(A.shape[0])
    out["one_gpu_path"] = timed(lambda rebuild, rtol=RTOL: A.solve_spd_dev(b, x, rtol=rtol, precond="amg", rebuild=rebuild),
                                ctx1.synchronize)
    A.close()
    b.free()
    x.free()
    ctx1.close()
    # (b) the row-partitioned path with a communicator of one rank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ctx2 = _hip.Context(0)
    plan = distributed.build_layer_partition(sysm, 0, 1)
    ds = distributed.DistributedSolver(ctx2, plan, dist)
    rec = timed(lambda rebuild, rtol=RTOL: ds.solve(rtol=rtol, precond="amg", rebuild=rebuild), ctx2.synchronize)
    out.update(rec)
    out["path"] = "row-partitioned (amg_setup_dist, single-reduction CG, RCCL communicator of one rank)"
    out["p2p_exchange"] = None
    for rd in ("r06", "r05"):
        try:
            out["p2p_exchange"] = json.load(open(os.path.join(ROOT, "profiles", rd + "_p2p_exchange.json")))
            out["p2p_exchange"]["static_from"] = (f"profiles/{rd}_p2p_exchange.json (tests/two_process_rank.py, two processes on one GPU; "
                                                  "not measured in this run)")
            break
        except Exception:
            continue
    ds.close()
    ctx2.close()
    return out


def dist_one_rank(steps: int):
    """The HEADLINE config (C4, all eight layers) through the row-partitioned path with an RCCL communicator of one rank:
    amg_setup_dist, single-reduction CG, the collectives of a communicator -- what the code path of an N-GPU run costs
    before there is a second rank, next to ``ms_per_step`` of the one-GPU path in the same line."""
    import torch
    import torch.distributed as dist
    from padne_amd import _hip, distributed, synthetic
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    sysm = synthetic.config("C4")
    ctx = _hip.Context(0)
    plan = distributed.build_layer_partition(sysm, 0, 1)
    ds = distributed.DistributedSolver(ctx, plan, dist)
    del sysm, plan
    ds.solve(rtol=RTOL, precond="amg", rebuild=True)
    ds.solve(rtol=RTOL, precond="amg", rebuild=True)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = ds.solve(rtol=RTOL, precond="amg", rebuild=True)
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    c0 = _hip.launch_count()
    ds.solve(rtol=RTOL, precond="amg", rebuild=True)
    ctx.synchronize()
    c1 = _hip.launch_count()
    rec = {"ms_per_step": ms, "setup_ms": float(last.setup_seconds) * 1e3,
           "us_per_iteration": float(last.seconds) / max(int(last.iterations), 1) * 1e6, "iterations": int(last.iterations),
           "launches_per_step": int(c1 - c0), "steps": steps,
           "path": "config C4 through DistributedSolver (row-partitioned path) with an RCCL communicator of one rank"}
    ds.close()
    ctx.close()
    return rec


def c5_block(ctx, steps: int):
    """Config C5: 8 current-source configurations on the N = 5 M matrix of C3, advanced in lockstep (one pass over every
    operator per iteration for all eight), next to ONE of them solved alone; hierarchy rebuilt in every call as in the
    headline step.  ``solves_equiv`` = what the eight cost in single solves."""
    from padne_amd import synthetic
    sysm, xy, tri = synthetic.config_on_device(ctx, "C5")
    nv = sysm.n_vertices
    N = nv + 1
    sig = np.array([m[2] for m in sysm.meshes])
    rows, cols, vals, rhs = stamps_of(sysm, N)
    L = ctx.assemble_system(N, xy, tri, sysm.mesh_offsets, sysm._tri_offsets, sig, rows, cols, vals)
    xy.free()
    tri.free()
    imap = np.arange(N, dtype=np.int32)
    imap[sysm.ground] = -1
    imap[imap > sysm.ground] -= 1
    imap[N - 1] = -1
    A = L.reduce(imap, nv - 1, -1.0)
    L.close()
    keep = np.flatnonzero(imap[:nv] >= 0)
    f, t = synthetic.multi_rhs_pairs(sysm, 8)
    B = np.zeros((8, nv - 1))
    for k in range(8):
        full = np.zeros(nv)
        full[f[k]] += 1.0
        full[t[k]] -= 1.0
        B[k] = full[keep]
    b8 = ctx.to_device(B)
    x8 = ctx.empty((8, nv - 1))
    b1 = ctx.to_device(B[0])
    x1 = ctx.empty(nv - 1)

    def run(bb, xx, k):
        A.solve_spd_dev(bb, xx, n_rhs=k, rtol=RTOL, precond="amg", rebuild=True)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = A.solve_spd_dev(bb, xx, n_rhs=k, rtol=RTOL, precond="amg", rebuild=True)
        ctx.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, r
    ms8, r8 = run(b8, x8, 8)
    ms1, r1 = run(b1, x1, 1)
    xs8 = ctx.to_device(np.random.default_rng(2).uniform(-1, 1, A.shape[1] * 8))
    ys8 = ctx.empty(A.shape[0] * 8)
    t8 = min(A.spmm8_time(xs8, ys8, 5, 30) for _ in range(3))
    rec = {"workload": f"{sysm.name}: {nv} nodes, 8 right-hand sides in lockstep, amg-PCG to rtol {RTOL:g}",
           "ms": ms8, "setup_ms": float(r8.setup_seconds) * 1e3, "iterations_total": int(r8.iterations),
           "rel_residual_max": float(r8.rel_residual), "single_solve_ms": ms1, "single_iterations": int(r1.iterations),
           "solves_equiv": ms8 / ms1, "rhs_per_s": 8e3 / ms8,
           # the same ratio without the hierarchy both sides rebuild: what the LOOPS cost (8 right-hand sides in lockstep in
           # units of one right-hand side alone)
           "single_setup_ms": float(r1.setup_seconds) * 1e3,
           "solve_parts_equiv": (ms8 - float(r8.setup_seconds) * 1e3) / max(ms1 - float(r1.setup_seconds) * 1e3, 1e-9),
           "spmm8_us": t8 * 1e6, "spmm8_gbs_algorithmic": A.spmm8_bytes / t8 / 1e9,
           "spmm8_frac": A.spmm8_bytes / t8 / 1e9 / HBM_PEAK_GBS, "spmm8_bytes_rule": "12 nnz + 4 N + 16 N k (= 216 N at k = 8)"}
    for d in (b8, x8, b1, x1, xs8, ys8):
        d.free()
    A.close()
    return rec


def small_block(ctx, live_cpu_limit: int = 300000):
    """Where the drop-in pays off: ``solve_system(L, r)`` -- host r in, host v out, everything derived from L built inside
    the call -- against the reference's own solve step (tocsc + spsolve + residual, solver.py:772-775) on the same host
    and the same system, at ~1 k and ~3 k unknowns (where the crossover lies), at the size of the shipped projects (config
    C1: ~11 k unknowns), at ~100 k and at ~1 M.  The CPU
    time of sizes above ``live_cpu_limit`` unknowns is read from profiles/r06_small.json (measured once on a GPU-box
    host by this very function, `scripts/small_sizes.py`); the rest is timed in this run."""
    from oracle import padne_oracle as O
    from padne_amd import solver, synthetic
    from padne_amd.reduction import Constraint, KKTLayout
    solver.set_context(ctx)
    static = {}
    try:
        static = {int(e["n"]): e for e in json.load(open(os.path.join(ROOT, "profiles", "r06_small.json")))["small"]}
    except Exception:
        static = {}
    out = []
    for layers, nx in ((4, 16), (4, 28), (4, 53), (4, 158), (4, 500)):
        sysm = synthetic.layered_system(layers, nx, nx, via_lattice=max(2, int(round(32 * nx / 1118))))
        nv = sysm.n_vertices
        N = nv + 1
        xy, tri, mvo, mto, sig = flat(sysm)
        rows, cols, vals, rhs = stamps_of(sysm, N)
        L = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
        layout = KKTLayout(size=N, n_potential=N - 1, constraints=[Constraint(index=N - 1, p=int(sysm.ground), n=-1, value=0.0)])
        Ls = solver.SystemMatrix(L, layout)
        v, info = solver.solve_system(Ls, rhs)                 # warm-up (pool, code objects)
        times = []
        for _ in range(5):
            for plan in Ls._plans.values():
                plan.close()
            Ls._plans.clear()
            ctx.synchronize()
            t0 = time.perf_counter()
            v, info = solver.solve_system(Ls, rhs)
            times.append(time.perf_counter() - t0)
        cached = []
        for _ in range(5):
            t0 = time.perf_counter()
            v, info_c = solver.solve_system(Ls, rhs)
            cached.append(time.perf_counter() - t0)
        rec = {"n": int(N), "hip_solve_system_ms": float(np.median(times)) * 1e3,
               "hip_solve_system_cached_plan_ms": float(np.median(cached)) * 1e3,
               "iterations": int(info.iterations), "residual_norm": float(info.residual_norm)}
        if N <= live_cpu_limit:
            els = [("R", int(a), int(b), float(r)) for a, b, r in zip(*sysm.resistors)]
            els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
            Lo, ro = O.assemble_system([(m[0], m[1], m[2]) for m in sysm.meshes], 0, els, sysm.ground)
            cpu = []
            for _ in range(3 if N < 50000 else 1):
                t0 = time.perf_counter()
                v_ref, gc, res = O.solve_system(Lo, ro)
                cpu.append(time.perf_counter() - t0)
            rec["cpu_spsolve_ms"] = float(np.median(cpu)) * 1e3
            rec["max_rel_potential_error"] = float(np.abs(v[:nv] - v_ref[:nv]).max() / np.abs(v_ref[:nv]).max())
        elif int(N) in static and "cpu_spsolve_ms" in static[int(N)]:
            rec["cpu_spsolve_ms"] = static[int(N)]["cpu_spsolve_ms"]
            rec["cpu_static_from"] = "profiles/r06_small.json (this function with live_cpu_limit raised, on a GPU-box host; not re-measured in this run)"
        if "cpu_spsolve_ms" in rec:
            rec["speedup"] = rec["cpu_spsolve_ms"] / rec["hip_solve_system_ms"]
        for plan in Ls._plans.values():
            plan.close()
        Ls._plans.clear()
        L.close()
        out.append(rec)
    return out


def launch_ranks(n: int) -> int:
    """Start one rank per GPU with torch.distributed.run (the command the driver uses) and wait for them."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def _time(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing has touched the GPU yet (no torch, no HIP
        # call), the ranks are CHILD processes (never an exec), and rank 0 prints the JSON line on the inherited stdout.
        sys.exit(launch_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")
    import torch  # plumbing only: rendezvous, barrier, max-over-ranks
    import torch.distributed as dist
    from padne_amd import _hip, synthetic

    distributed_path = world > 1 or args.force_distributed
    if distributed_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    ctx = _hip.Context(local_rank)

    def barrier():
        if distributed_path:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    # ---- build the workload (untimed) -----------------------------------------------------------
    t_setup0 = time.perf_counter()
    if not distributed_path:
        # meshes generated on the device (padne_generate_grid_mesh: the arrays of synthetic.config, bit for bit, without
        # seconds of numpy and 0.4 GB of PCIe); the lumped elements are host-side index lists as before
        sysm, xy, tri = synthetic.config_on_device(ctx, args.workload)
        nv = sysm.n_vertices
        N = nv + 1
        mvo, mto = sysm.mesh_offsets, sysm._tri_offsets
        sig = np.array([m[2] for m in sysm.meshes])
        rows, cols, vals, rhs = stamps_of(sysm, N)
        ctx.synchronize()
        t0 = time.perf_counter()
        L = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
        ctx.synchronize()
        t_assemble = time.perf_counter() - t0
        # the same assembly WARM (pool blocks in place, code objects loaded): device-resident meshes in, CSR matrix out,
        # wall time per call with its host looks; the cold call above pays first-use costs and is reported apart
        asm_warm = []
        for rep in range(ASSEMBLY_REPEATS + 1):
            ctx.synchronize()
            t0 = time.perf_counter()
            L_again = ctx.assemble_system(N, xy, tri, mvo, mto, sig, rows, cols, vals)
            ctx.synchronize()
            if rep > 0:      # (the first call beside a live result still draws its 1 GB of result arrays from hipMalloc: 1.5-40 ms by box)
                asm_warm.append(time.perf_counter() - t0)
            assert L_again.nnz == L.nnz
            L_again.close()
        imap = np.arange(N, dtype=np.int32)
        imap[sysm.ground] = -1
        imap[imap > sysm.ground] -= 1
        imap[N - 1] = -1
        t0 = time.perf_counter()
        A = L.reduce(imap, nv - 1, -1.0)
        ctx.synchronize()
        t_reduce = time.perf_counter() - t0
        red_warm, red_warm_host = [], []
        imap_dev = ctx.to_device(imap)      # the index map is an input like the mesh: resident in HBM when the timed region starts
        for host_map in (True, False):
            for _ in range(ASSEMBLY_REPEATS):
                ctx.synchronize()
                t0 = time.perf_counter()
                A_again = L.reduce(imap if host_map else imap_dev, nv - 1, -1.0)
                ctx.synchronize()
                (red_warm_host if host_map else red_warm).append(time.perf_counter() - t0)
                assert A_again.nnz == A.nnz
                A_again.close()
        imap_dev.free()
        L_nnz = L.nnz
        L_full = L                   # kept for the seam measurement below (solve_system on host vectors)
        keep = np.flatnonzero(imap[:nv] >= 0)
        b = ctx.to_device(-rhs[keep])
        x = ctx.empty(A.shape[0])
        solver = lambda time_spmv=False: A.solve_spd_dev(b, x, rtol=RTOL, time_spmv=time_spmv, precond=args.precond,  # noqa: E731
                                                         rebuild=True)
        standalone = lambda: A.spmv_time(b, x, 5, 50)  # noqa: E731   same matrix, back-to-back launches
        n_local, nnz_local = A.shape[0], A.nnz
        spmv_bytes = A.spmv_bytes
        hierarchy_shapes = (lambda: A.amg_shapes()) if args.precond == "amg" else None
    else:
        from padne_amd import distributed
        sysm = synthetic.config(args.workload)
        nv = sysm.n_vertices
        N = nv + 1
        plan = distributed.build_layer_partition(sysm, rank, world)
        dsolver = distributed.DistributedSolver(ctx, plan, dist)
        t_assemble, t_reduce = dsolver.t_assemble, dsolver.t_reduce
        solver = lambda time_spmv=False: dsolver.solve(rtol=RTOL, time_spmv=time_spmv, precond=args.precond,  # noqa: E731
                                                       rebuild=True)
        n_local, nnz_local = dsolver.n_owned, dsolver.nnz
        spmv_bytes = dsolver.spmv_bytes
        standalone = None
        hierarchy_shapes = None
        halo_exchange = ("peer-to-peer stores into hipIpc mailboxes, device-side arrival flags" if dsolver.p2p
                         else "none (one rank)" if world == 1
                         else "ncclAllGather (the mailboxes could not be shared or did not pass their self-test)")
    t_setup = time.perf_counter() - t_setup0

    # ---- warmup + timed steps ---------------------------------------------------------------------
    for _ in range(args.warmup):
        solver()
    barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = solver()
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed_path:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # one more (untimed) solve with HIP events around every 16th SpMV launch, on the kernels' stream
    prof = solver(time_spmv=True)
    barrier()
    t_standalone = standalone() if standalone is not None else None
    seam = None
    if not distributed_path and args.precond == "amg" and not args.no_seam:
        seam = seam_timing(ctx, L_full, sysm, rhs, max(2, min(args.steps, 5)))
    if not distributed_path:
        L_full.close()               # (the assembled KKT matrix, ~1 GB at C4: only the seam measurement needed it)
    shapes_now = None
    if hierarchy_shapes is not None:
        try:
            shapes_now = hierarchy_shapes()      # (of the last timed step's hierarchy, while the matrix is alive)
        except Exception:
            shapes_now = None
    # launches of one step, of its setup and per iteration (library-side count, outside the timed region): a whole step, a
    # solve on the cached hierarchy, a shorter solve on it
    launches = None
    if not distributed_path and args.precond == "amg" and not args.no_launch_count:
        try:
            c0 = _hip.launch_count()
            solver()
            ctx.synchronize()
            c1 = _hip.launch_count()
            cached = A.solve_spd_dev(b, x, rtol=RTOL, precond=args.precond, rebuild=False)
            ctx.synchronize()
            c2 = _hip.launch_count()
            short = A.solve_spd_dev(b, x, rtol=1e-5, precond=args.precond, rebuild=False)
            ctx.synchronize()
            c3 = _hip.launch_count()
            d_it = max(int(cached.iterations) - int(short.iterations), 1)
            launches = {"per_step": int(c1 - c0), "per_setup": int((c1 - c0) - (c2 - c1)),
                        "per_iteration": round(((c2 - c1) - (c3 - c2)) / d_it, 1)}
        except Exception as exc:
            launches = {"error": repr(exc)}
    c5 = proxy = small = d1 = None
    if not distributed_path and args.precond == "amg" and args.workload == "C4":
        A.close()
        b.free()
        x.free()
        if not args.no_c5:
            try:
                c5 = c5_block(ctx, max(2, min(args.steps, 5)))
            except Exception as exc:
                c5 = {"error": repr(exc)}
        if not args.no_small:
            try:
                small = small_block(ctx, args.small_cpu_limit)
            except Exception as exc:
                small = {"error": repr(exc)}
        if not args.no_rank_proxy:
            try:
                proxy = rank_proxy(max(3, min(args.steps, 10)))
            except Exception as exc:
                proxy = {"error": repr(exc)}
        if not args.no_dist_one_rank:
            try:
                d1 = dist_one_rank(max(3, min(args.steps, 10)))
            except Exception as exc:
                d1 = {"error": repr(exc)}
        try:
            import torch.distributed as _d
            if _d.is_initialized():
                _d.destroy_process_group()
        except Exception:
            pass

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        t_spmv = prof.spmv_seconds
        # the product of the multigrid-preconditioned loop on one GPU multiplies a search direction STORED in single precision
        # (pcg.hip): 4 instead of 8 bytes of x per row; the standalone launches (and every other path) multiply doubles
        p_stored_f32 = (args.precond == "amg" and not distributed_path and "PADNE_PCG_P64" not in os.environ
                        and "PADNE_AMG_F64" not in os.environ)
        spmv_bytes_standalone = spmv_bytes
        if p_stored_f32:
            spmv_bytes = spmv_bytes - 4 * n_local
        achieved = spmv_bytes / t_spmv / 1e9 if t_spmv > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "spmv_traffic.json")
        if os.path.exists(tpath) and not distributed_path:      # measured on the 1-GPU launch; a rank's launch is smaller
            try:
                traffic = json.load(open(tpath)).get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "CG solves/s",
            "value": args.steps / elapsed,
            "unit": "solves/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{sysm.name}: {len(sysm.meshes)}-layer jittered triangular Laplacian, "
                                   f"{nv} nodes, via-ring stitched, 1 A source/sink, {args.precond}-PCG to rtol {RTOL:g}",
                       "n_unknowns": int(nv - 1), "nnz_per_rank": int(nnz_local), "rows_per_rank": int(n_local),
                       "parallelism": f"layer-partitioned x{args.gpus}",
                       **({"halo_exchange": halo_exchange} if distributed_path else {})},
            "preconditioner": {"kind": args.precond, "levels": int(last.levels),
                               "operator_complexity": float(last.operator_complexity),
                               "setup_ms_per_step": float(last.setup_seconds) * 1e3,
                               "solve_ms_per_step": float(last.seconds) * 1e3,
                               **({"launches": launches} if launches is not None else {})},
            "iterations": int(last.iterations), "restarts": int(last.restarts),
            "rel_residual": float(last.rel_residual),
            "us_per_iteration": last.seconds / max(last.iterations, 1) * 1e6,
            "pcg_textbook_gbs": (232.0 * n_local * last.iterations / last.seconds / 1e9
                                 if last.seconds > 0 and args.precond == "jacobi" else None),
            "setup_seconds": {"total": t_setup, "assemble": t_assemble, "reduce": t_reduce},
            "roofline": {"bound": "hbm", "kernel": "csr_spmv_kernel<SPMV_DOT, double, double, double" +
                                                   (", x stored as float" if p_stored_f32 else "") + "> (q = A p with p.q epilogue)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_static_from": "profiles/spmv_traffic.json (rocprofv3 --pmc passes of the round's final build, "
                                                "scripts/pmc_bench.sh; not collected in this run)" if traffic is not None else None,
                         "bytes_per_launch": int(spmv_bytes), "seconds_per_launch": t_spmv,
                         "note": "achieved/frac: kernel timed in situ inside the CG loop; standalone_frac: the same "
                                 "kernel launched back to back (no dirty predecessor)",
                         "bytes_rule": ("12 nnz + 16 n + 4 (values 8 + columns 4 per non-zero; row pointer 4, x 4, y 8 per row)"
                                        if p_stored_f32 else "12 nnz + 20 n + 4 (values 8 + columns 4 per non-zero; row pointer 4, x 8, y 8 per row)"),
                         "standalone_frac": (spmv_bytes_standalone / t_standalone / 1e9 / HBM_PEAK_GBS) if t_standalone else None},
        }
        if shapes_now is not None:
            try:
                sb = step_algorithmic_bytes(shapes_now, int(last.iterations))
                ach = sb["total"] / (elapsed / args.steps) / 1e9
                out["roofline"]["step"] = {
                    "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "bytes_per_step": sb["total"], "setup_bytes": sb["setup"], "bytes_per_iteration": sb["per_iteration"],
                    "setup_frac": sb["setup"] / max(float(last.setup_seconds), 1e-12) / 1e9 / HBM_PEAK_GBS,
                    "iteration_frac": sb["per_iteration"] * int(last.iterations) / max(float(last.seconds), 1e-12) / 1e9 / HBM_PEAK_GBS,
                    "note": "whole timed step (multigrid setup + all CG iterations) by the textbook CSR byte count of "
                            "SURVEY 8d; the entry above is the dominant kernel alone"}
                step_traffic, step_traffic_file = step_counter_traffic()
                if step_traffic is not None:
                    # what the implementation really moves (float cycle, one-byte window positions; but also every re-fetch):
                    # counter bytes of one step over the time of one step
                    out["roofline"]["step"].update({
                        "traffic": step_traffic["step_bytes"], "traffic_setup": step_traffic["setup_bytes"],
                        "traffic_loop": step_traffic["loop_bytes"],
                        "traffic_frac": step_traffic["step_bytes"] / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                        "traffic_static_from": step_traffic_file + " (scripts/pmc_setup_sum.py over the --pmc passes of "
                                               "that round's final build; not collected in this run)"})
            except Exception as exc:
                out["roofline"]["step"] = {"error": repr(exc)}
        if not distributed_path:
            # per-element stiffness assembly (mesh.py:124-139, solver.py:171-213, 563-575) from device-resident meshes, timed
            # warm in THIS run; algorithmic bytes per SURVEY 8d: 16 N (xy) + 12 T (tri) + 12 nnz (out) = 124 N
            asm_ms = float(np.mean(asm_warm)) * 1e3
            red_ms = float(np.mean(red_warm)) * 1e3
            asm_bytes = 16 * nv + 12 * int(mto[-1]) + 12 * int(L_nnz)
            asm_traffic, asm_traffic_file = None, None
            for rd in ("r06", "r05", "r04"):
                try:
                    asm_traffic = json.load(open(os.path.join(ROOT, "profiles", rd + "_assembly_traffic.json")))
                    asm_traffic_file = f"profiles/{rd}_assembly_traffic.json"
                    break
                except Exception:
                    continue
            out["assembly"] = {
                "ms": asm_ms, "ms_min": float(np.min(asm_warm)) * 1e3, "ms_cold_first_call": t_assemble * 1e3,
                "ms_each": [round(float(t) * 1e3, 3) for t in asm_warm],
                "repeats": ASSEMBLY_REPEATS, "algorithmic_bytes": int(asm_bytes), "bytes_rule": "16 N + 12 T + 12 nnz (= 124 N at 7 nnz/row)",
                "achieved": asm_bytes / (asm_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": asm_bytes / (asm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": asm_traffic.get("bytes_per_assembly") if asm_traffic else None,
                "traffic_static_from": (asm_traffic_file + " (rocprofv3 --pmc passes, scripts/pmc_asm.sh; not collected in this run)")
                                       if asm_traffic else None,
                "reduce_ms": red_ms, "reduce_ms_host_map": float(np.mean(red_warm_host)) * 1e3,
                "what": "padne_assemble_system on device-resident meshes + host stamp lists -> CSR L (wall time per call, warm, "
                        "host looks included); reduce_ms: L -> A = -P^T L P (padne_csr_reduce: drop the ground vertex and the "
                        "multiplier row, flip the sign) with the index map resident on the device, reduce_ms_host_map: the same "
                        "with the 40 MB map uploaded from pageable host memory in every call"}
            out["value_with_assembly"] = 1e3 / (ms_per_step + asm_ms + red_ms)
            out["ms_per_step_with_assembly"] = ms_per_step + asm_ms + red_ms
        if seam is not None:
            out["seam"] = seam
            out["seam_ms_per_solve"] = seam["ms_per_solve"]
        if c5 is not None:
            out["c5"] = c5
        if proxy is not None:
            if "ms_per_step" in proxy:
                proxy["speedup_bound_at_8_gpus"] = ms_per_step / proxy["ms_per_step"]
                proxy["note"] = ("ms_per_step of the headline config on this GPU / ms_per_step of one rank's share with nothing to "
                                 "wait for: an upper bound of the 8-GPU speed-up, not a measurement of it")
            out["rank_proxy"] = proxy
        if d1 is not None:
            if "ms_per_step" in d1:
                d1["vs_one_gpu_path"] = d1["ms_per_step"] / ms_per_step
            out["dist_one_rank"] = d1
        if small is not None:
            out["small"] = small
        if args.gpus == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample_nx)
            except Exception as exc:  # the GPU numbers stand on their own
                out["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(out), flush=True)
    if distributed_path:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
