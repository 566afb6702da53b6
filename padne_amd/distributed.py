"""Layer partition of a meshed system across the GPUs of one node (one process per GPU).

No counterpart in the reference (single process); the plan follows
SURVEY.md section 8e: rows are partitioned *by layer* -- every mesh block is already contiguous in
the global numbering (``solver.py:221-229``) and the mesh Laplacian of a layer has no off-rank
columns.  The only cross-rank couplings are lumped elements whose terminals sit on different
ranks (via resistor rings).  Per PCG iteration a rank therefore needs

* the current values of the few thousand remote vertices its via resistors touch: every rank
  packs the owned values somebody else needs (``export`` list) and one ``ncclAllGather`` of
  ``m = max_r len(export_r)`` doubles per rank delivers them (``padne_ctx_set_halo``);
* the global dot products: one-workgroup fold + ``ncclAllReduce`` of 1-2 doubles.

This module is host-side index bookkeeping only: it decides who owns what, renumbers the lumped
stamps into each rank's local index space and hands the local triangles + stamps to the same
device assembly the single-GPU path uses.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

from .synthetic import SyntheticSystem


@dataclass
class RankPlan:
    rank: int
    world: int
    meshes: list                   # this rank's (xy, tri, sigma, layer) blocks
    g0: int                        # first owned global vertex
    g1: int                        # one past the last owned global vertex
    ground_local: int              # local index of the ground vertex, or -1 if another rank owns it
    m: int                         # exchange segment length (max export count over ranks)
    export_local: np.ndarray       # int32: owned local indices other ranks need, in export order
    n_local_unknowns: int          # owned vertices + world*m exchange slots
    coo_rows: np.ndarray           # lumped stamps of the owned rows, local indices, stamp order
    coo_cols: np.ndarray
    coo_vals: np.ndarray
    rhs_local: np.ndarray          # r restricted to the owned vertices
    owned_global: np.ndarray = field(default=None)   # global vertex id of each owned local vertex

    @property
    def n_owned_vertices(self) -> int:
        return self.g1 - self.g0


def layer_ranges(n_layers: int, world: int):
    """Contiguous, balanced assignment of layers to ranks."""
    if world > n_layers:
        raise ValueError(f"{world} ranks but only {n_layers} layers: strip partitioning of a layer is not implemented")
    cuts = [(r * n_layers) // world for r in range(world + 1)]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def build_layer_partition(sysm: SyntheticSystem, rank: int, world: int) -> RankPlan:
    """Plan for ``rank``; deterministic and identical on every rank for the shared parts."""
    offs = sysm.mesh_offsets
    n_layers = len(sysm.meshes)
    ranges = layer_ranges(n_layers, world)
    vert_range = [(int(offs[a]), int(offs[b])) for a, b in ranges]
    owner_cut = np.array([v[0] for v in vert_range] + [int(offs[-1])], dtype=np.int64)

    def owner_of(g):
        return np.searchsorted(owner_cut, g, side="right") - 1

    ra, rb, rr = sysm.resistors
    oa, ob = owner_of(ra), owner_of(rb)
    cross = oa != ob
    # export lists: vertices of rank r referenced from a row owned by another rank; sorted & unique
    exports = []
    for r in range(world):
        need = np.concatenate([ra[cross & (oa == r)], rb[cross & (ob == r)]])
        exports.append(np.unique(need))
    m = int(max((len(e) for e in exports), default=0))
    g0, g1 = vert_range[rank]
    n_own = g1 - g0

    def local_col(g):
        """Local column of global vertex g: owned -> g - g0, remote -> its slot in the exchange area."""
        g = np.asarray(g, dtype=np.int64)
        out = g - g0
        o = owner_of(g)
        rem = o != rank
        if rem.any():
            pos = np.empty(g.shape, dtype=np.int64)
            for r in range(world):
                sel = rem & (o == r)
                if sel.any():
                    pos[sel] = n_own + r * m + np.searchsorted(exports[r], g[sel])
            out = np.where(rem, pos, out)
        return out

    g = 1.0 / rr
    # stamp order of the reference per resistor: (a,a,-g) (a,b,+g) (b,b,-g) (b,a,+g)   solver.py:480-484
    rows = np.stack([ra, ra, rb, rb], 1).reshape(-1)
    cols = np.stack([ra, rb, rb, ra], 1).reshape(-1)
    vals = np.stack([-g, g, -g, g], 1).reshape(-1)
    mine = owner_of(rows) == rank
    rows, cols, vals = rows[mine], cols[mine], vals[mine]
    coo_rows = rows - g0
    coo_cols = local_col(cols)
    rhs = np.zeros(n_own)
    f, t, cur = sysm.current_sources
    for ff, tt, ii in zip(f, t, cur):
        if g0 <= ff < g1:
            rhs[ff - g0] += ii
        if g0 <= tt < g1:
            rhs[tt - g0] += -ii
    ground_local = int(sysm.ground - g0) if g0 <= sysm.ground < g1 else -1
    a, b = ranges[rank]
    return RankPlan(rank=rank, world=world, meshes=sysm.meshes[a:b], g0=g0, g1=g1, ground_local=ground_local, m=m,
                    export_local=(exports[rank] - g0).astype(np.int32), n_local_unknowns=n_own + world * m,
                    coo_rows=coo_rows.astype(np.int64), coo_cols=coo_cols.astype(np.int64), coo_vals=vals,
                    rhs_local=rhs, owned_global=np.arange(g0, g1, dtype=np.int64))


def reduced_local_map(plan: RankPlan):
    """Index map local unknown -> local reduced unknown: the ground vertex (if owned) is dropped, owned
    vertices keep their order, exchange slots follow.  Returns (map int32, n_owned_reduced, export_reduced)."""
    n_own = plan.n_owned_vertices
    n_loc = plan.n_local_unknowns
    imap = np.arange(n_loc, dtype=np.int32)
    if plan.ground_local >= 0:
        imap[plan.ground_local] = -1
        imap[plan.ground_local + 1:] -= 1
    n_owned_red = n_own - (1 if plan.ground_local >= 0 else 0)
    export_red = imap[plan.export_local]
    if (export_red < 0).any():
        # the ground vertex is exported: its value is 0 by definition; keep the slot, feed it from any
        # owned unknown times zero is not possible -> such stamps are Dirichlet terms of the remote rows.
        raise NotImplementedError("a via resistor lands on the ground vertex; choose another ground")
    return imap, n_owned_red, export_red.astype(np.int32)


class DistributedSolver:
    """Per-rank driver: local assembly, reduction, halo plan, RCCL communicator, solve."""

    def __init__(self, ctx, plan: RankPlan, dist=None, team=None, block_preconditioner: bool = False):
        """``dist``: an initialised ``torch.distributed`` module (one process per GPU, RCCL), or ``team``: a
        ``_hip.LocalTeam`` whose members are contexts of this process (single-GPU rehearsal, one thread per rank).

        The multigrid preconditioner is one hierarchy over all ranks (aggregates stay inside a rank; every level
        has its own exchange plan).  ``block_preconditioner=True`` selects block-Jacobi instead: one V-cycle of
        each rank's own diagonal block with no communication inside the cycle (3-6x more CG iterations)."""
        self.ctx, self.plan = ctx, plan
        if team is not None:
            team.join(ctx, plan.rank)
        else:
            # RCCL communicator: rank 0 creates the id, torch.distributed broadcasts the 128 bytes
            import torch
            if plan.rank == 0:
                uid = np.frombuffer(ctx.comm_unique_id(), dtype=np.uint8).copy()
            else:
                uid = np.zeros(128, dtype=np.uint8)
            t = torch.from_numpy(uid).cuda()
            dist.broadcast(t, src=0)
            ctx.comm_init(bytes(t.cpu().numpy().tobytes()), plan.rank, plan.world)
        # local assembly on this GPU
        xy = np.concatenate([mm[0] for mm in plan.meshes])
        tri = np.concatenate([mm[1] for mm in plan.meshes])
        mvo = np.concatenate([[0], np.cumsum([len(mm[0]) for mm in plan.meshes])]).astype(np.int64)
        mto = np.concatenate([[0], np.cumsum([len(mm[1]) for mm in plan.meshes])]).astype(np.int64)
        sig = np.array([mm[2] for mm in plan.meshes])
        t0 = time.perf_counter()
        L = ctx.assemble_system(plan.n_local_unknowns, xy, tri, mvo, mto, sig, plan.coo_rows, plan.coo_cols,
                                plan.coo_vals)
        ctx.synchronize()
        self.t_assemble = time.perf_counter() - t0
        imap, n_owned, export_red = reduced_local_map(plan)
        t0 = time.perf_counter()
        # this rank's rows of A = -L_vv: owned rows x [owned | world * m exchange slots]
        rmap = imap.copy()
        rmap[plan.n_owned_vertices:] = -1
        self.A = L.relabel(rmap, n_owned, imap, n_owned + plan.world * plan.m, -1.0)
        self.A_block = None
        if block_preconditioner:
            # owned x owned diagonal block (couplings to other ranks dropped)
            self.A_block = L.reduce(rmap, n_owned, -1.0)
            self.A.set_preconditioner_block(self.A_block)
        ctx.synchronize()
        self.t_reduce = time.perf_counter() - t0
        L.close()
        self.n_owned = n_owned
        ctx.set_halo(n_owned, plan.m, export_red)
        keep = np.flatnonzero(imap[:plan.n_owned_vertices] >= 0)
        self.b = ctx.to_device(-plan.rhs_local[keep])
        self.x = ctx.empty(n_owned)
        self.nnz = self.A.nnz
        self.spmv_bytes = 12 * self.A.nnz + 20 * n_owned + 4

    def solve(self, rtol=1e-12, time_spmv=False, precond="amg", rebuild=False):
        return self.A.solve_spd_dev(self.b, self.x, rtol=rtol, time_spmv=time_spmv, precond=precond, rebuild=rebuild)

    def solution(self) -> np.ndarray:
        return self.x.numpy()
