"""Partition of a meshed system across the GPUs of one node (one process per GPU).

No counterpart in the reference (single process); the plan follows SURVEY.md section 8e:

* with at least as many layers as ranks the rows are partitioned *by layer* -- every mesh block is contiguous in the
  global numbering (``solver.py:221-229``) and the mesh Laplacian of a layer has no off-rank columns, so the only
  cross-rank couplings are lumped elements whose terminals sit on different ranks (via resistor rings);
* with fewer layers than ranks a layer is cut into horizontal *strips* (SURVEY 8e fallback): the rank also assembles
  the ring of vertices around its strip, whose rows it then drops, and the mesh edges across a cut join the via
  resistors as cross-rank couplings;
* voltage sources and regulators: what is partitioned are the unknowns of the REDUCED system (``reduction.py``: the
  unknowns a source ties together are one unknown, known potentials none) -- a group's rows all live on the rank of
  its smallest member, and the multiplier currents come back from the residual rows of the group members, each
  evaluated where that row was assembled (:func:`build_problem_partition`, :func:`solve_partitioned`).

Either way a rank's matrix is ``owned rows x [owned | world * m exchange slots]``: per product every rank packs the
owned values somebody else needs (``export`` list) and one ``ncclAllGather`` of ``m = max_r len(export_r)`` values per
rank delivers them (``padne_ctx_set_halo``); dot products are a one-workgroup fold + ``ncclAllReduce``.

This module is host-side index bookkeeping only: it decides who owns what, renumbers triangles and lumped stamps into
each rank's local index space and hands them to the same device assembly the single-GPU path uses.  Every rank holds
the whole problem description on the host (as ``solve()`` does after meshing) and derives ALL ranks' export lists from
it, so no rank ever decides anything from data another rank lacks.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

from . import _hip
from .synthetic import SyntheticSystem


@dataclass
class RankPlan:
    rank: int
    world: int
    meshes: list                   # this rank's (xy, tri, sigma, layer) blocks: owned vertices + the ring around them
    local_global: np.ndarray       # int64: global unknown of every local assembly unknown (mesh blocks first)
    owned_global: np.ndarray       # int64: global unknowns this rank owns (ascending), the ground included if it is here
    ground_local: int              # position of the ground in owned_global, or -1 if another rank owns it
    m: int                         # exchange segment length (max export count over ranks)
    export_owned: np.ndarray       # int64: positions in owned_global of the unknowns other ranks need, export order
    n_local_unknowns: int          # local assembly unknowns (len(local_global))
    coo_rows: np.ndarray           # lumped stamps of the owned rows, local assembly indices, stamp order
    coo_cols: np.ndarray
    coo_vals: np.ndarray
    rhs_local: np.ndarray          # r restricted to owned_global
    row_map: np.ndarray            # int32 over local assembly unknowns: reduced owned row, or -1 (ring, ground)
    col_map: np.ndarray            # int32: reduced owned column | exchange slot n_owned_reduced + q*m + k | -1 (ground)
    n_owned_reduced: int
    export_reduced: np.ndarray     # int32: reduced owned indices of the exported unknowns, export order
    partial_mesh: bool = False     # the mesh blocks are pieces of larger meshes (strip partition)
    # index reduction (voltage sources, several known potentials): an unknown is either KNOWN (c = its potential) or a
    # member of a free group, v[u] = y[group] + c[u]; a group is named by its smallest member (its representative) and
    # owned -- all its members' rows -- by the representative's rank
    reps_owned: np.ndarray = None  # int64: representatives this rank owns, ascending = order of the reduced owned rows
    rhs_rows: np.ndarray = None    # r over the local assembly unknowns (zero on rows of other ranks)
    rhs_owner_mask: np.ndarray = None   # bool over the local assembly unknowns: this rank assembled the whole row
    c_local: np.ndarray = None     # known part over the local assembly unknowns
    rep_global: np.ndarray = None  # int64 over all unknowns: representative, -1 = known
    c_global: np.ndarray = None    # known part over all unknowns
    reduction: object = None       # the reduction.Reduction behind index_map / c (multiplier recovery, regulators)
    rhs_full: np.ndarray = None    # r of the whole KKT system (multiplier rows included)

    @property
    def n_owned_vertices(self) -> int:           # (name kept from the layer-only plan: owned unknowns incl. ground)
        return len(self.owned_global)

    @property
    def g0(self) -> int:
        return int(self.owned_global[0]) if len(self.owned_global) else 0

    @property
    def g1(self) -> int:
        return int(self.owned_global[-1]) + 1 if len(self.owned_global) else 0

    @property
    def n_cols(self) -> int:
        return self.n_owned_reduced + self.world * self.m


@dataclass
class Partition:
    """Where this process stands in a row-partitioned solve: ``solver.solve_meshed(..., partition=Partition(...))``.

    ``dist``: the initialised ``torch.distributed`` module (one process per GPU, backend "nccl" = RCCL); or ``team``: a
    ``_hip.LocalTeam`` with one thread per rank on one GPU (tests), in which case ``gather`` must collect one Python
    object per rank into a list (``torch.distributed.all_gather_object`` does that for processes)."""
    rank: int
    world: int
    dist: object = None
    team: object = None
    gather: object = None
    device: int = None            # GPU of this rank (default: rank for processes, 0 for a team)


def layer_ranges(n_layers: int, world: int):
    """Contiguous, balanced assignment of layers to ranks (needs n_layers >= world; fewer layers are cut into strips by
    :func:`owners_of_unknowns`)."""
    if world > n_layers:
        raise ValueError(f"{world} ranks but only {n_layers} layers: use owners_of_unknowns (strip partition)")
    cuts = [(r * n_layers) // world for r in range(world + 1)]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def owners_of_unknowns(meshes, n_unknowns: int, world: int, links=None) -> np.ndarray:
    """Owner rank of every unknown.  ``meshes``: (xy, tri, sigma, layer) in global numbering order.

    * n_meshes >= world: whole meshes, contiguous and balanced (the layer partition);
    * fewer meshes than ranks: the ranks are dealt to the meshes in proportion to their sizes and each mesh is cut into
      that many horizontal strips of (nearly) equal vertex counts (cuts at quantiles of y, ties kept together by index).
    Unknowns behind the vertices (internal nodes) go to the owner of the smallest unknown they are linked to
    (``links``: pairs of unknowns coupled by lumped stamps), rank 0 if none."""
    sizes = np.array([len(m[0]) for m in meshes], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    n_vert = int(offs[-1])
    owner = np.zeros(n_unknowns, dtype=np.int32)
    n_m = len(meshes)
    if n_m >= world:
        for r, (a, b) in enumerate(layer_ranges(n_m, world)):
            owner[offs[a]:offs[b]] = r
    elif n_m > 0:
        # ranks per mesh: at least one each, the rest by largest remainder of the size shares
        share = sizes / max(sizes.sum(), 1) * world
        per = np.maximum(np.floor(share).astype(int), 1)
        while per.sum() > world:
            per[np.argmax(per)] -= 1
        order = np.argsort(-(share - np.floor(share)), kind="stable")
        k = 0
        while per.sum() < world:
            per[order[k % n_m]] += 1
            k += 1
        r0 = 0
        for mi, (xy, _tri, _s, _l) in enumerate(meshes):
            n = len(xy)
            k_strips = int(per[mi])
            rank_in_mesh = np.zeros(n, dtype=np.int32)
            if k_strips > 1 and n > 0:
                by_y = np.lexsort((np.arange(n), xy[:, 1]))
                rank_in_mesh[by_y] = np.minimum((np.arange(n, dtype=np.int64) * k_strips) // n, k_strips - 1)
            owner[offs[mi]:offs[mi + 1]] = r0 + rank_in_mesh
            r0 += k_strips
    if n_unknowns > n_vert:
        owner[n_vert:] = -1
        if links is not None and len(links):
            lk = np.asarray(links, dtype=np.int64).reshape(-1, 2)
            for _ in range(4):                                   # chains of internal nodes: a few sweeps settle them
                for a, b in ((lk[:, 0], lk[:, 1]), (lk[:, 1], lk[:, 0])):
                    sel = (owner[a] < 0) & (owner[b] >= 0)
                    if sel.any():
                        o = np.argsort(b[sel], kind="stable")[::-1]          # smallest linked unknown wins (written last)
                        owner[a[sel][o]] = owner[b[sel][o]]
        owner[owner < 0] = 0
    return owner


def build_partition(meshes, n_unknowns: int, coo, rhs: np.ndarray, ground: int, owner: np.ndarray, rank: int,
                    world: int, index_map: np.ndarray = None, c: np.ndarray = None) -> RankPlan:
    """Plan of ``rank`` for an arbitrary ownership of the unknowns.

    ``meshes``: (xy, tri, sigma, layer) in global numbering order (vertices first, ``solver.py:221-229``);
    ``coo = (rows, cols, vals)``: lumped stamps on global unknowns in stamp order, WITHOUT the ground row / column
    (``ground`` is eliminated: it is 0 V by definition, ``solver.py:558-560``).  Deterministic and identical on every
    rank for the shared parts (export lists of all ranks).

    ``index_map`` / ``c`` (``reduction.Reduction.index_map`` / ``.c`` over the potential unknowns): the index reduction of
    a system with voltage sources -- unknowns with the same non-negative ``index_map`` form one reduced unknown
    (``v[u] = y + c[u]``), negative ones are known (``v[u] = c[u]``: the ground, nodes tied to it by sources, pins of
    floating copper).  The rows of a group are summed on the rank that owns its representative (smallest member); known
    unknowns have no row and no exchange slot, their columns go into the right-hand side (``DistributedSolver``)."""
    rows_g, cols_g, vals_g = (np.asarray(a) for a in coo)
    rows_g = rows_g.astype(np.int64)
    cols_g = cols_g.astype(np.int64)
    owner = np.array(owner, dtype=np.int32)
    sizes = np.array([len(m[0]) for m in meshes], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n_vert = int(offs[-1])
    if index_map is None:                                        # the only constraint is the ground
        known = np.zeros(n_unknowns, dtype=bool)
        known[ground] = True
        rep = np.arange(n_unknowns, dtype=np.int64)
        rep[ground] = -1
        c = np.zeros(n_unknowns)
    else:
        imap = np.asarray(index_map)[:n_unknowns].astype(np.int64)
        c = np.asarray(c, dtype=np.float64)[:n_unknowns]
        known = imap < 0
        free = np.flatnonzero(~known)
        first = np.full(int(imap.max()) + 1 if len(free) else 0, n_unknowns, dtype=np.int64)
        np.minimum.at(first, imap[free], free)
        rep = np.full(n_unknowns, -1, dtype=np.int64)
        rep[free] = first[imap[free]]
        owner[free] = owner[rep[free]]                           # a group lives where its representative lives
    # ---- who needs what: (unknown, needing rank) pairs from mesh edges across a cut and from lumped stamps ----------
    need_u, need_r = [], []
    tri_mine = []
    for mi, (xy, tri, _s, _l) in enumerate(meshes):
        t = np.asarray(tri, dtype=np.int64) + offs[mi]
        if len(t) == 0:
            tri_mine.append(np.zeros(0, dtype=np.int64))
            continue
        ot = owner[t]                                            # [n_tri, 3]
        mixed = (ot[:, 0] != ot[:, 1]) | (ot[:, 1] != ot[:, 2])
        if mixed.any():
            tm, om = t[mixed], ot[mixed]
            for k in range(3):
                for j in range(3):
                    if j != k:
                        sel = om[:, k] != om[:, j]
                        need_u.append(tm[sel, k])
                        need_r.append(om[sel, j])
        tri_mine.append(np.flatnonzero((ot == rank).any(axis=1)))
    cross = owner[rows_g] != owner[cols_g]
    need_u.append(cols_g[cross])
    need_r.append(owner[rows_g[cross]])
    need_u = np.concatenate(need_u) if need_u else np.zeros(0, np.int64)
    need_r = np.concatenate(need_r) if need_r else np.zeros(0, np.int32)
    keep = ~known[need_u]                                        # known potentials are never exchanged
    need_u, need_r = rep[need_u[keep]], need_r[keep]             # what travels is the group's value: asked of its representative
    keep = owner[need_u] != need_r                               # (a member of a group this very rank owns)
    need_u, need_r = need_u[keep], need_r[keep]
    exports = [np.unique(need_u[owner[need_u] == q]) for q in range(world)]
    m = int(max((len(e) for e in exports), default=0))
    # ---- local assembly unknowns: owned + ring (triangles) + remote stamp terminals ------------------------------------
    owned = np.flatnonzero(owner == rank).astype(np.int64)
    mine_rows = owner[rows_g] == rank
    parts = [owned, cols_g[mine_rows]]
    local_meshes = []
    for mi, (xy, tri, s, l) in enumerate(meshes):
        t = np.asarray(tri, dtype=np.int64)[tri_mine[mi]] + offs[mi]
        parts.append(t.reshape(-1))
    U = np.unique(np.concatenate(parts))
    partial = False
    for mi, (xy, tri, s, l) in enumerate(meshes):
        a, b = np.searchsorted(U, [offs[mi], offs[mi + 1]])
        if b == a:
            continue
        vg = U[a:b]
        t = np.asarray(tri, dtype=np.int64)[tri_mine[mi]] + offs[mi]
        tl = (np.searchsorted(vg, t.reshape(-1)).reshape(-1, 3)).astype(np.int32)
        if 0 < len(tri_mine[mi]) < len(tri):                     # a piece of the mesh: owned strip + ring
            partial = True
        local_meshes.append((np.ascontiguousarray(np.asarray(xy)[vg - offs[mi]]), tl, float(s), int(l)))
    # ---- relabelling maps ------------------------------------------------------------------------------------------------
    own_red = np.unique(rep[owned][~known[owned]])               # representatives owned here = the reduced owned rows
    n_red = len(own_red)
    is_owned = owner[U] == rank
    row_map = np.full(len(U), -1, dtype=np.int32)
    sel = is_owned & ~known[U]
    row_map[sel] = np.searchsorted(own_red, rep[U[sel]]).astype(np.int32)
    col_map = row_map.copy()
    rem = ~is_owned & ~known[U]
    if rem.any():
        ur = rep[U[rem]]
        orr = owner[ur]
        slot = np.empty(len(ur), dtype=np.int64)
        for q in range(world):
            s_q = orr == q
            if not s_q.any():
                continue
            ex = exports[q]
            if len(ex) == 0:
                slot[s_q] = -1
                continue
            pos = np.minimum(np.searchsorted(ex, ur[s_q]), len(ex) - 1)
            # ring vertices that no owned ROW references (they only complete a triangle) are nobody's import: no slot
            slot[s_q] = np.where(ex[pos] == ur[s_q], n_red + q * m + pos, -1)
        col_map[rem] = slot.astype(np.int32)
    export_red = np.searchsorted(own_red, exports[rank]).astype(np.int32)
    export_owned = np.searchsorted(owned, exports[rank]).astype(np.int64)
    coo_rows = np.searchsorted(U, rows_g[mine_rows])
    coo_cols = np.searchsorted(U, cols_g[mine_rows])
    g_pos = np.flatnonzero(owned == ground)
    rhs = np.asarray(rhs, dtype=np.float64)
    rhs_rows = np.where(is_owned, rhs[U], 0.0)
    return RankPlan(rank=rank, world=world, meshes=local_meshes, local_global=U, owned_global=owned,
                    ground_local=int(g_pos[0]) if len(g_pos) else -1, m=m, export_owned=export_owned,
                    n_local_unknowns=len(U), coo_rows=coo_rows.astype(np.int64), coo_cols=coo_cols.astype(np.int64),
                    coo_vals=np.asarray(vals_g, dtype=np.float64)[mine_rows], rhs_local=np.asarray(rhs, dtype=np.float64)[owned],
                    row_map=row_map, col_map=col_map, n_owned_reduced=n_red, export_reduced=export_red,
                    partial_mesh=partial, reps_owned=own_red, rhs_rows=rhs_rows, rhs_owner_mask=is_owned,
                    c_local=c[U].copy(), rep_global=rep, c_global=c)


def resistor_stamps(ra, rb, rr):
    """COO stamps of resistors in the reference's order per element: (a,a,-g) (a,b,+g) (b,b,-g) (b,a,+g), solver.py:480-484."""
    g = 1.0 / np.asarray(rr, dtype=np.float64)
    rows = np.stack([ra, ra, rb, rb], 1).reshape(-1)
    cols = np.stack([ra, rb, rb, ra], 1).reshape(-1)
    vals = np.stack([-g, g, -g, g], 1).reshape(-1)
    return rows.astype(np.int64), cols.astype(np.int64), vals


def build_layer_partition(sysm: SyntheticSystem, rank: int, world: int) -> RankPlan:
    """Plan of ``rank`` for a :class:`SyntheticSystem` (resistors + current sources): by layer, or by strips of layers
    when there are fewer layers than ranks."""
    n = sysm.n_vertices + sysm.n_internal
    ra, rb, rr = sysm.resistors
    coo = resistor_stamps(ra, rb, rr)
    rhs = np.zeros(n)
    f, t, cur = sysm.current_sources
    np.add.at(rhs, f, cur)
    np.add.at(rhs, t, -np.asarray(cur))
    links = np.stack([ra, rb], axis=1) if len(ra) else None
    owner = owners_of_unknowns(sysm.meshes, n, world, links)
    return build_partition(sysm.meshes, n, coo, rhs, int(sysm.ground), owner, rank, world)


def build_problem_partition(meshes, conductances, mesh_layers, stamps, rhs: np.ndarray, n_potential: int, rank: int,
                            world: int) -> RankPlan:
    """Plan of ``rank`` for an assembled Problem (what ``solver.solve_meshed`` has after numbering and stamp listing):
    ``meshes`` are :class:`padne_amd.mesh.Mesh`, ``stamps`` the :class:`padne_amd.solver.StampList``.

    Resistors, current sources, vias, VOLTAGE SOURCES and REGULATORS: the index reduction of ``reduction.py`` (ground,
    source-tied groups, pins of floating copper) is computed on every rank -- it is index logic on the lumped elements --
    and the partition is one of the REDUCED unknowns (see :func:`build_partition`).  A regulator (a constraint whose
    multiplier also enters other rows, ``Constraint.gamma``) costs one extra solve with the same matrix
    (:func:`solve_partitioned`)."""
    from .reduction import KKTLayout, build_reduction, floating_component_pins
    cons = list(stamps.constraints)
    rhs = np.asarray(rhs, dtype=np.float64)
    for cst in cons:                                             # multiplier rows take their right-hand side from r
        cst.value = float(rhs[cst.index])
    layout = KKTLayout(size=int(stamps.shape[0]), n_potential=int(n_potential), constraints=cons)
    ground = int(layout.ground_constraint.p)
    rows, cols, vals = stamps.arrays()
    keep = (rows < n_potential) & (cols < n_potential)            # drop the multiplier rows / columns of the KKT layout
    ms = [(m.points, m.triangles, float(s), int(l)) for m, s, l in zip(meshes, conductances, mesh_layers)]
    off = rows != cols
    links = np.stack([rows[keep & off], cols[keep & off]], axis=1)
    mesh_offsets = np.concatenate([[0], np.cumsum([len(m.points) for m in meshes])]).astype(np.int64)
    pins = floating_component_pins(n_potential, ground, cons, mesh_offsets=mesh_offsets,
                                   links=np.unique(np.sort(links, axis=1), axis=0) if len(links) else links)
    red = build_reduction(layout, pins)
    ties = np.array([[cst.p, cst.n] for cst in cons if cst.n >= 0], dtype=np.int64).reshape(-1, 2)
    owner = owners_of_unknowns(ms, n_potential, world, np.concatenate([links, ties]) if len(ties) else links)
    plain = len(cons) == 1 and not pins                           # only the ground: the plan of the earlier rounds, bit for bit
    plan = build_partition(ms, n_potential, (rows[keep], cols[keep], vals[keep]), rhs[:n_potential], ground, owner, rank,
                           world, None if plain else red.index_map[:n_potential], None if plain else red.c[:n_potential])
    if not plain:
        plan.reduction = red
        plan.rhs_full = rhs
    return plan


def reduced_local_map(plan: RankPlan):
    """(row_map, col_map, n_owned_reduced, export_reduced) of the plan: what ``padne_csr_relabel`` turns the local
    assembly into this rank's rows of A = -L_vv, ``owned rows x [owned | world * m exchange slots]``."""
    return plan.row_map, plan.col_map, plan.n_owned_reduced, plan.export_reduced


class DistributedSolver:
    """Per-rank driver: local assembly, reduction, halo plan, RCCL communicator, solve."""

    def __init__(self, ctx, plan: RankPlan, dist=None, team=None, block_preconditioner: bool = False,
                 keep_assembly: bool = False):
        """``dist``: an initialised ``torch.distributed`` module (one process per GPU, RCCL), or ``team``: a
        ``_hip.LocalTeam`` whose members are contexts of this process (single-GPU rehearsal, one thread per rank).

        The multigrid preconditioner is one hierarchy over all ranks (aggregates stay inside a rank; every level
        has its own exchange plan).  ``block_preconditioner=True`` selects block-Jacobi instead: one V-cycle of
        each rank's own diagonal block with no communication inside the cycle (3-6x more CG iterations)."""
        self.ctx, self.plan = ctx, plan
        self.p2p = False                                      # halo exchanges as peer-to-peer stores into shared mailboxes?
        if team is not None:
            team.join(ctx, plan.rank)
            self.p2p = True
        else:
            import torch
            backend = dist.get_backend() if hasattr(dist, "get_backend") else "nccl"
            if backend == "gloo":
                # ranks RCCL cannot connect (two processes on one GPU) or a host without RCCL: the library's collectives go
                # through this process group's all-gather on host memory (padne_ctx_comm_init_host)
                def allgather(send):
                    t = torch.from_numpy(send)
                    parts = [torch.empty_like(t) for _ in range(plan.world)]
                    dist.all_gather(parts, t)
                    return torch.cat(parts).numpy().tobytes()
                ctx.comm_init_host(plan.rank, plan.world, allgather)
            else:
                # RCCL communicator: rank 0 creates the id, torch.distributed broadcasts the 128 bytes
                if plan.rank == 0:
                    uid = np.frombuffer(ctx.comm_unique_id(), dtype=np.uint8).copy()
                else:
                    uid = np.zeros(128, dtype=np.uint8)
                t = torch.from_numpy(uid).cuda()
                dist.broadcast(t, src=0)
                ctx.comm_init(bytes(t.cpu().numpy().tobytes()), plan.rank, plan.world)
            self.p2p = self._share_mailboxes(dist)
        # local assembly on this GPU
        ms = plan.meshes
        xy = np.concatenate([mm[0] for mm in ms]) if ms else np.zeros((0, 2))
        tri = np.concatenate([mm[1] for mm in ms]) if ms else np.zeros((0, 3), np.int32)
        mvo = np.concatenate([[0], np.cumsum([len(mm[0]) for mm in ms])]).astype(np.int64)
        mto = np.concatenate([[0], np.cumsum([len(mm[1]) for mm in ms])]).astype(np.int64)
        sig = np.array([mm[2] for mm in ms])
        t0 = time.perf_counter()
        L = ctx.assemble_system(plan.n_local_unknowns, xy, tri, mvo, mto, sig, plan.coo_rows, plan.coo_cols,
                                plan.coo_vals, partial_mesh=plan.partial_mesh)
        ctx.synchronize()
        self.t_assemble = time.perf_counter() - t0
        n_owned = plan.n_owned_reduced
        t0 = time.perf_counter()
        # this rank's rows of A = -L_vv: owned rows x [owned | world * m exchange slots]
        self.A = L.relabel(plan.row_map, n_owned, plan.col_map, plan.n_cols, -1.0)
        self.A_block = None
        if block_preconditioner:
            # owned x owned diagonal block (couplings to other ranks dropped)
            self.A_block = L.reduce(plan.row_map, n_owned, -1.0)
            self.A.set_preconditioner_block(self.A_block)
        ctx.synchronize()
        self.t_reduce = time.perf_counter() - t0
        self._Lc = L.matvec(plan.c_local) if plan.c_local is not None and np.any(plan.c_local) else None
        self.L_local = L if keep_assembly else None               # kept for the residual rows of the multiplier recovery
        if not keep_assembly:
            L.close()
        self.n_owned = n_owned
        ctx.set_halo(n_owned, plan.m, plan.export_reduced)
        # b = sum over a group's rows of (L c - r): c = known part of the potentials (zero without sources)
        resid = -np.asarray(plan.rhs_rows, dtype=np.float64)
        if plan.c_local is not None and np.any(plan.c_local):
            resid = resid + self._Lc
        sel = plan.row_map >= 0
        b_red = np.bincount(plan.row_map[sel], weights=resid[sel], minlength=n_owned).astype(np.float64)
        self.owned_reduced_global = plan.reps_owned
        self.b_norm2 = float(np.einsum("i,i->", b_red, b_red))    # this rank's share of ||b||^2 (einsum, not a threaded BLAS call: its spinning workers delay the HIP runtime)
        self.b = ctx.to_device(b_red)
        self.x = ctx.empty(n_owned)
        self.nnz = self.A.nnz
        self.spmv_bytes = 12 * self.A.nnz + 20 * n_owned + 4

    def _share_mailboxes(self, dist) -> bool:
        """The peer-to-peer halo exchange between processes (include/padne_hip.h, padne_ctx_p2p_*): every rank allocates its
        mailbox and all-gathers the hipIpc handles; if ANY rank cannot allocate, export or map one, all ranks close theirs and
        the exchanges stay all-gathers -- the decision is taken from gathered flags, so every rank takes the same path."""
        ctx, plan = self.ctx, self.plan
        if plan.world < 2 or plan.m <= 0:
            return False
        slots = max(4 * int(plan.m), 16384)                   # the levels of the hierarchy exchange fewer values than the matrix

        def gather(obj):
            parts = [None] * plan.world
            dist.all_gather_object(parts, obj)
            return parts
        try:
            handle = ctx.p2p_export(slots)
        except _hip.HipError:
            handle = None
        handles = gather(handle)
        ok = all(h is not None for h in handles)
        if ok:
            try:
                ctx.p2p_import(b"".join(handles), plan.world)
            except _hip.HipError:
                ok = False
        if not all(gather(ok)):
            ctx.p2p_close()
            return False
        # ... and one real exchange of known values through them before any solve depends on it (2 s bound)
        try:
            ok = ctx.p2p_selftest()
        except _hip.HipError:
            ok = False
        if not all(gather(ok)):
            ctx.p2p_close()
            return False
        return True

    def solve(self, rtol=1e-12, time_spmv=False, precond="amg", rebuild=False):
        return self.A.solve_spd_dev(self.b, self.x, rtol=rtol, time_spmv=time_spmv, precond=precond, rebuild=rebuild)

    def set_rhs(self, b_owned: np.ndarray):
        """Another right-hand side over this rank's reduced rows (same matrix, same hierarchy)."""
        self.b.set(np.ascontiguousarray(b_owned, dtype=np.float64))

    def project_rows(self, rows: dict) -> np.ndarray:
        """P^T of a sparse vector {unknown: value} restricted to this rank's reduced rows."""
        plan = self.plan
        out = np.zeros(self.n_owned)
        for x, val in rows.items():
            rep = int(plan.rep_global[x]) if plan.rep_global is not None else int(x)
            if rep < 0:
                continue
            pos = int(np.searchsorted(plan.reps_owned, rep))
            if pos < len(plan.reps_owned) and plan.reps_owned[pos] == rep:
                out[pos] += val
        return out

    def residual_rows(self, v_all: np.ndarray, rows) -> dict:
        """rho_x = r_x - (L v)_x for the listed unknowns whose rows this rank assembled completely (its own)."""
        plan = self.plan
        Lv = self.L_local.matvec(np.ascontiguousarray(v_all[plan.local_global]))
        out = {}
        for x in rows:
            pos = int(np.searchsorted(plan.local_global, x))
            if pos < len(plan.local_global) and plan.local_global[pos] == x and plan.rhs_owner_mask[pos]:
                out[int(x)] = float(plan.rhs_rows[pos] - Lv[pos])
        return out

    def residual_sq_of_own_rows(self, v_all: np.ndarray, corrections: dict) -> float:
        """Sum over the potential rows this rank assembled completely of (r_x - (L v)_x - corrections[x])^2: this rank's
        share of ||L v - r||^2 on the ORIGINAL system (``solver.py:775``).  ``corrections``: what the multiplier columns
        add to (L v)_x -- they are not part of the local assembly -- as {unknown: value}."""
        plan = self.plan
        Lv = self.L_local.matvec(np.ascontiguousarray(v_all[plan.local_global]))
        rho = np.where(plan.rhs_owner_mask, plan.rhs_rows - Lv, 0.0)
        for x, val in corrections.items():
            pos = int(np.searchsorted(plan.local_global, x))
            if pos < len(plan.local_global) and plan.local_global[pos] == x and plan.rhs_owner_mask[pos]:
                rho[pos] -= val
        return float(np.einsum("i,i->", rho, rho))

    def close(self):
        if self.L_local is not None:
            self.L_local.close()
            self.L_local = None

    def solution(self) -> np.ndarray:
        """Potentials of the owned unknowns except the ground, in the order of ``owned_reduced_global``."""
        return self.x.numpy()


class _NoSolve:
    """Result of a solve whose right-hand side vanishes on every rank."""
    abs_residual = 0.0
    iterations = 0
    rel_residual = 0.0
    seconds = 0.0


def solve_partitioned(plan: RankPlan, ctx, dist=None, team=None, rtol: float = 1e-12, gather=None):
    """Solve and return ``(v, result)`` on every rank.  ``gather(obj)`` collects one Python object per rank into a list;
    with ``dist`` it defaults to ``all_gather_object``.

    Plans without an index reduction (the ground is the only constraint): ``v`` = potentials of all unknowns, ground 0.
    Plans of :func:`build_problem_partition` with sources: ``v`` is the whole solution vector of the KKT system like
    ``solver.solve_system`` returns it -- potentials, then the multiplier currents (sources, regulators, ground row),
    recovered from the residual rows r - L v of the source-tied unknowns, each computed by the rank that assembled that
    row.  Every regulator adds one solve with the same matrix and hierarchy (``solver.py`` module docstring, step 3)."""
    red = plan.reduction
    ds = DistributedSolver(ctx, plan, dist=dist, team=team, keep_assembly=red is not None)
    try:
        def collect(obj):
            if gather is not None:
                return gather(obj)
            parts = [None] * plan.world
            dist.all_gather_object(parts, obj)
            return parts
        # the reference's absolute residual bar (1e-9, tests/test_solver.py:2083-2089) as on one GPU: a large right-hand
        # side tightens the relative tolerance (solver.ABS_RESIDUAL_TARGET), decided from the global norm so that all ranks agree
        from . import solver as _solver

        def solve_for(b_norm2_local):
            b_norm = float(np.sqrt(sum(collect(b_norm2_local))))
            tol = rtol
            if b_norm > 0.0 and rtol * b_norm > _solver.ABS_RESIDUAL_TARGET:
                tol = max(_solver.ABS_RESIDUAL_TARGET / b_norm, _solver.RTOL_FLOOR)
            if b_norm == 0.0:                                     # nothing to solve for (all ranks agree)
                return _NoSolve(), np.zeros(ds.n_owned)
            result = ds.solve(rtol=tol)
            return result, ds.solution()

        n = len(plan.rep_global) if plan.rep_global is not None else 0

        def at_representatives(y_owned):
            parts = collect((ds.owned_reduced_global, y_owned))
            size = max([n, int(plan.owned_global.max()) + 1 if len(plan.owned_global) else 0] +
                       [int(p[0].max()) + 1 if len(p[0]) else 0 for p in parts])
            y = np.zeros(size)                                    # value of every group at its representative
            for idx, vals in parts:
                y[idx] = vals
            return y
        res, y_owned = solve_for(ds.b_norm2)
        totals = {"iterations": int(res.iterations), "seconds": float(res.seconds), "rel_residual": float(res.rel_residual)}
        y = at_representatives(y_owned)
        if plan.rep_global is None:
            return y, res
        free = np.flatnonzero(plan.rep_global >= 0)

        def expand(y_rep, with_known):
            v = np.zeros(len(y_rep))
            if with_known:
                v[:len(plan.c_global)] = plan.c_global
            v[free] += y_rep[plan.rep_global[free]]
            return v
        v_pot = expand(y, True)
        if red is None:
            return v_pot, res
        # ---- regulators and multiplier currents, as solver.solve_system does on one GPU ----------------------------------
        N = red.layout.size
        n_pot = red.layout.n_potential
        members = sorted({int(x) for mem, cons, _ in red.groups if cons for x in mem})

        def residual_at_members(v_potentials):
            rho = np.zeros(N)
            for part in collect(ds.residual_rows(v_potentials, members)):
                for x, val in part.items():
                    rho[x] = val
            return rho
        Z = []
        for cst in red.regulators:
            b_k = ds.project_rows(cst.gamma)
            ds.set_rhs(b_k)
            res_k, z_owned = solve_for(float(np.einsum("i,i->", b_k, b_k)))
            totals["iterations"] += int(res_k.iterations)
            totals["seconds"] += float(res_k.seconds)
            totals["rel_residual"] = max(totals["rel_residual"], float(res_k.rel_residual))
            Z.append(expand(at_representatives(z_owned), False))
        mult_known = {}
        if red.regulators:
            keys = [cst.index for cst in red.regulators]
            K = len(keys)

            def currents_for(i_vec):
                vv = v_pot + sum(i_vec[k] * Z[k] for k in range(K))
                return red.multipliers(residual_at_members(vv), dict(zip(keys, i_vec)))
            base = currents_for(np.zeros(K))
            F0 = np.array([base[k] for k in keys])
            J = np.zeros((K, K))
            for k in range(K):
                e = np.zeros(K)
                e[k] = 1.0
                ck = currents_for(e)
                J[:, k] = np.array([ck[q] for q in keys]) - F0
            i_reg = np.linalg.solve(np.eye(K) - J, F0)
            v_pot = v_pot + sum(i_reg[k] * Z[k] for k in range(K))
            mult_known = dict(zip(keys, i_reg))
        v = np.zeros(N)
        v[:n_pot] = v_pot[:n_pot]
        for idx, val in red.multipliers(residual_at_members(v_pot), mult_known).items():
            if idx >= 0:                        # negative: the current through the pin of a floating component
                v[idx] = val
        for idx, val in mult_known.items():
            v[idx] = val
        # ||L v - r||_2 of the RETURNED vector on the original KKT system, as solve_system reports it on one GPU
        # (solver.py:775): every rank evaluates the potential rows it assembled, the multiplier columns
        # (+i at p, -i at n, gain entries, the ground column) are added from the constraint list, the multiplier rows
        # v_p - v_n = U are evaluated on every rank alike
        corr: dict = {}
        sq_rows = 0.0
        for cst in red.layout.constraints:
            i_c = float(v[cst.index])
            corr[cst.p] = corr.get(cst.p, 0.0) + i_c
            if cst.n >= 0:
                corr[cst.n] = corr.get(cst.n, 0.0) - i_c
            for row, gain in cst.gamma.items():
                corr[row] = corr.get(row, 0.0) + gain * i_c
            lhs = v[cst.p] - (v[cst.n] if cst.n >= 0 else 0.0)
            sq_rows += float(lhs - plan.rhs_full[cst.index]) ** 2
        v_ext = v_pot if len(v_pot) >= n_pot else np.concatenate([v_pot, np.zeros(n_pot - len(v_pot))])
        sq = sum(collect(ds.residual_sq_of_own_rows(v_ext, corr))) + sq_rows
        import types
        out = types.SimpleNamespace(**{k: getattr(res, k, None) for k in
                                       ("restarts", "status", "spmv_seconds", "setup_seconds", "operator_complexity", "levels",
                                        "precond_fallbacks")})
        out.abs_residual = float(np.sqrt(sq))
        out.reduced_abs_residual = float(res.abs_residual)
        out.iterations, out.seconds, out.rel_residual = totals["iterations"], totals["seconds"], totals["rel_residual"]
        return v, out
    finally:
        ds.close()
