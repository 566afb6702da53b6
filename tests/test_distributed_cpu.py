"""world_size-2 (and 4) gloo tests of the multi-GPU plan on the CPU.

The product's partition / halo plan (padne_amd.distributed, host index logic) is executed by a
numpy restatement of what the device does per iteration (pack -> all-gather -> local SpMV ->
all-reduce of the dot products); the distributed PCG must reproduce the single-process direct solve.
The local matrices are assembled by the oracle, so no GPU is involved."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from oracle import padne_oracle as O  # noqa: E402
from padne_amd import distributed, synthetic  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _local_matrix(plan):
    n_loc = plan.n_local_unknowns
    blocks = [(-1.0) * 0 for _ in ()]
    rows, cols, vals = [], [], []
    off = 0
    for xy, tri, sigma, _ in plan.meshes:
        Lm = O.laplace_operator(xy, tri)
        rows.append(Lm.row + off)
        cols.append(Lm.col + off)
        vals.append(sigma * Lm.data)
        off += len(xy)
    rows.append(plan.coo_rows)
    cols.append(plan.coo_cols)
    vals.append(plan.coo_vals)
    L = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n_loc, n_loc)).tocsr()
    imap, n_owned, export_red = distributed.reduced_local_map(plan)
    keep = np.flatnonzero(imap >= 0)
    A = (-L[keep][:, keep]).tocsr()
    b = -plan.rhs_local[np.flatnonzero(imap[:plan.n_owned_vertices] >= 0)]
    return A, b, n_owned, export_red


def _worker(rank, world, port, nl, nx, ny, lattice, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sysm = synthetic.layered_system(nl, nx, ny, via_lattice=lattice)
        plan = distributed.build_layer_partition(sysm, rank, world)
        A, b, n_owned, export_red = _local_matrix(plan)
        m = plan.m
        A_own = A[:n_owned]
        dinv = 1.0 / A_own.diagonal()

        def allsum(v):
            t = torch.tensor([v], dtype=torch.float64)
            dist.all_reduce(t)
            return float(t.item())

        def extended(v):
            seg = torch.zeros(m, dtype=torch.float64)
            seg[:len(export_red)] = torch.from_numpy(v[export_red])
            parts = [torch.zeros(m, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, seg)
            return np.concatenate([v] + [p.numpy() for p in parts])

        x = np.zeros(n_owned)
        r = b.copy()
        z = dinv * r
        p = z.copy()
        rz = allsum(r @ z)
        bb = allsum(b @ b)
        it = 0
        while allsum(r @ r) > (1e-13 ** 2) * bb and it < 20000:
            q = A_own @ extended(p)
            alpha = rz / allsum(p @ q)
            x += alpha * p
            r -= alpha * q
            z = dinv * r
            rz_new = allsum(r @ z)
            p = z + (rz_new / rz) * p
            rz = rz_new
            it += 1
        gathered = [None] * world
        dist.all_gather_object(gathered, (plan.g0, plan.g1, plan.ground_local, x))
        if rank == 0:
            v = np.zeros(sysm.n_vertices)
            for g0, g1, gl, xs in gathered:
                idx = np.arange(g0, g1)
                if gl >= 0:
                    idx = np.delete(idx, gl)
                v[idx] = xs
            els = [("R", int(a), int(b_), float(rr)) for a, b_, rr in zip(*sysm.resistors)]
            els += [("I", int(f), int(t), float(i)) for f, t, i in zip(*sysm.current_sources)]
            Lf, rf = O.assemble_system([(mm[0], mm[1], mm[2]) for mm in sysm.meshes], 0, els, sysm.ground)
            v_ref, _, _ = O.solve_system(Lf, rf)
            err = np.abs(v - v_ref[:sysm.n_vertices]).max() / np.abs(v_ref[:sysm.n_vertices]).max()
            out.put((it, err, plan.m))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nl", [(2, 2), (2, 4), (4, 4)])
def test_layer_partitioned_pcg_matches_direct_solve(world, nl):
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nl, 26, 22, 3, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    it, err, m = out.get()
    assert it > 50 and m > 0
    assert err <= 1e-8


def test_partition_bookkeeping():
    sysm = synthetic.layered_system(4, 20, 16, via_lattice=3)
    n_per = 20 * 16
    plans = [distributed.build_layer_partition(sysm, r, 2) for r in range(2)]
    assert [(p.g0, p.g1) for p in plans] == [(0, 2 * n_per), (2 * n_per, 4 * n_per)]
    assert plans[0].ground_local == 0 and plans[1].ground_local == -1
    assert plans[0].m == plans[1].m and plans[0].m >= len(plans[0].export_local)
    # every stamp row is owned, every column is either owned or inside the exchange area
    for p in plans:
        n_own = p.n_owned_vertices
        assert p.coo_rows.min() >= 0 and p.coo_rows.max() < n_own
        assert p.coo_cols.min() >= 0 and p.coo_cols.max() < p.n_local_unknowns
        remote = p.coo_cols >= n_own
        seg = (p.coo_cols[remote] - n_own) // p.m
        assert np.all(seg != p.rank)
    # stamps are conserved: each resistor contributes 4 entries in total
    assert sum(len(p.coo_vals) for p in plans) == 4 * len(sysm.resistors[0])
    assert np.isclose(sum(p.rhs_local.sum() for p in plans), 0.0)
    with pytest.raises(ValueError):
        distributed.layer_ranges(2, 4)
    assert distributed.layer_ranges(8, 4) == [(0, 2), (2, 4), (4, 6), (6, 8)]
