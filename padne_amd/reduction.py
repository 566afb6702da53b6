"""Host-side index logic that turns the reference's KKT system into an SPD one and back.

The matrix assembled by the reference (``solver.py:783-812``) is indefinite: every
``VoltageSource`` / ``VoltageRegulator`` and the ground add a multiplier row/column
(``solver.py:493-538, 544-560``), and a regulator's gain column makes it non-symmetric.
CG cannot run on that.  Algebraically equivalent rewriting, with *no arithmetic on matrix
entries on the host* (the matrix work -- ``P^T L P``, products ``L c`` -- runs on the device):

1. constraint rows say ``v_p - v_n = U`` (and ``v_g = 0``): union-find with potentials merges
   the tied nodes into groups, ``v_x = y_G + c_x``; the ground group is known outright.
2. adding the KCL rows of a group cancels the multiplier currents (``+i`` in row p, ``-i`` in
   row n), leaving ``A y = b`` with ``A = -P^T L_vv P`` (SPD: negated Dirichlet Laplacian) and
   ``b = -P^T (r - L c)``.
3. after the solve the multipliers (source currents, ground current) follow from the KCL
   residual of the merged rows, peeled leaf by leaf along each group's tree of sources.
4. a gain column ``gamma_k`` (regulator mirror, ``solver.py:537-538``) makes the right-hand side
   depend on the unknown current ``i_k``: ``y = y0 + sum_k i_k z_k`` with ``A z_k = P^T gamma_k``
   (one extra solve per regulator), closed by a k*k linear system for ``i``.

Everything here is O(#constraints) bookkeeping plus a few O(N) numpy gathers.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


class SingularSystemError(ValueError):
    """The constraint rows are contradictory or redundant (the reference's LU would be singular)."""


@dataclass
class Constraint:
    """One multiplier unknown ``index``: row says v[p] - v[n] = value (n = -1: v[p] = value).

    ``gamma``: extra entries of the multiplier's *column* beyond the +1/-1 at p/n, as
    ``{row: coefficient}`` (regulator gain stamps).
    """
    index: int
    p: int
    n: int
    value: float
    gamma: dict = field(default_factory=dict)


@dataclass
class KKTLayout:
    size: int                       # N
    n_potential: int                # vertices + internal nodes (indices [0, n_potential))
    constraints: list               # list[Constraint], includes the ground row (n = -1)

    @property
    def ground_constraint(self) -> Constraint:
        for c in self.constraints:
            if c.n < 0:
                return c
        raise SingularSystemError("system has no ground row")


def infer_layout(L_csr, r: np.ndarray, n_potential: int | None = None) -> KKTLayout:
    """Recover the KKT structure from a matrix in the reference's layout.

    Used when ``solve_system(L, r)`` is handed a bare scipy matrix (e.g. one produced by the
    reference's own ``assemble_system``).  The reference numbers the multiplier unknowns after all
    potentials (``solver.py:441-460, 757-760``), so they are the longest *suffix* of rows that look
    like constraint rows: zero diagonal, one or two entries of +-1, all of them in columns in front
    of the row itself (the terminals are potentials).  A potential row can also have a zero diagonal
    and +-1 entries -- an internal node that touches only source terminals, e.g. between two voltage
    sources in series -- but its entries sit in multiplier *columns*, behind it, which ends the
    suffix.  ``n_potential`` (if the caller knows it) fixes the split outright.
    """
    N = L_csr.shape[0]
    indptr, indices, data = L_csr.indptr, L_csr.indices, L_csr.data
    diag = L_csr.diagonal()

    def looks_like_constraint_row(k: int) -> bool:
        a, b = indptr[k], indptr[k + 1]
        return (diag[k] == 0 and 1 <= b - a <= 2 and bool(np.all(np.abs(data[a:b]) == 1.0))
                and bool(np.all(indices[a:b] < k)))

    if n_potential is None:
        n_pot = N
        while n_pot > 0 and looks_like_constraint_row(n_pot - 1):
            n_pot -= 1
    else:
        n_pot = int(n_potential)
        if not 0 <= n_pot <= N or not all(looks_like_constraint_row(k) for k in range(n_pot, N)):
            raise SingularSystemError("rows behind n_potential are not constraint rows of the form v_p - v_n = U")
    mult = list(range(n_pot, N))
    mult_set = set(mult)
    if not mult:
        raise SingularSystemError("no ground / multiplier rows found: not a padne system matrix")
    csc = L_csr.tocsc()
    cons = []
    for k in mult:
        cols = indices[indptr[k]:indptr[k + 1]]
        vals = data[indptr[k]:indptr[k + 1]]
        if len(cols) == 1:
            if vals[0] != 1.0:
                raise SingularSystemError("unexpected single-entry multiplier row")
            p, n = int(cols[0]), -1
        else:
            pos = cols[vals > 0]
            neg = cols[vals < 0]
            if len(pos) != 1 or len(neg) != 1:
                raise SingularSystemError("multiplier row is not of the form v_p - v_n")
            p, n = int(pos[0]), int(neg[0])
        gamma = {}
        rws = csc.indices[csc.indptr[k]:csc.indptr[k + 1]]
        cv = csc.data[csc.indptr[k]:csc.indptr[k + 1]]
        for rr, vv in zip(rws, cv):
            rr = int(rr)
            if rr in mult_set:
                raise SingularSystemError("multiplier-multiplier coupling is not supported")
            base = (1.0 if rr == p else 0.0) - (1.0 if rr == n else 0.0)
            if vv - base != 0.0:
                gamma[rr] = float(vv - base)
        cons.append(Constraint(index=k, p=p, n=n, value=float(r[k]), gamma=gamma))
    return KKTLayout(size=N, n_potential=n_pot, constraints=cons)


class _UnionFind:
    """Union-find with potentials: pot[x] = v[x] - v[find(x)]."""

    def __init__(self):
        self.parent: dict = {}
        self.pot: dict = {}

    def find(self, x):
        if x not in self.parent:
            self.parent[x] = x
            self.pot[x] = 0.0
            return x
        path = []
        while self.parent[x] != x:
            path.append(x)
            x = self.parent[x]
        root = x
        # path compression, accumulating potentials from the root outwards
        for node in reversed(path):
            par = self.parent[node]
            self.pot[node] = self.pot[node] + (self.pot[par] if par != root else 0.0)
            self.parent[node] = root
        return root

    def offset(self, x) -> float:
        self.find(x)
        return self.pot[x]

    def union(self, a, b, d) -> bool:
        """Impose v[a] - v[b] = d.  Returns False if a and b were already tied."""
        ra, rb = self.find(a), self.find(b)
        if ra == rb:
            return False
        # attach ra under rb:  v[ra] - v[rb] = d + pot[b] - pot[a]
        self.parent[ra] = rb
        self.pot[ra] = d + self.pot[b] - self.pot[a]
        return True


class Reduction:
    """The index reduction of one KKT system.  SPARSE first: what distinguishes it from the identity is listed --
    ``elim`` (sorted potentials that have no reduced unknown of their own: known potentials and the non-representative
    members of source-tied groups), ``tied`` (member, representative) pairs of the free groups, ``known`` {unknown: known
    part c of its potential} -- in O(#constraints); the device builds its index map from these lists
    (``padne_kkt_create``).  The dense ``index_map`` (int32[N]: reduced unknown of each unknown, -1 = eliminated) and
    ``c`` (f64[N]) of the host restatement (``expand`` / ``rhs``, the row-partitioned plan, the locality ordering) are
    made on first access."""

    def __init__(self, layout: KKTLayout, n_free: int, elim: np.ndarray, tied: list, known: dict, groups: list,
                 regulators: list):
        self.layout = layout
        self.n_free = int(n_free)
        self.elim = np.asarray(elim, dtype=np.int64)     # sorted, < n_potential
        self.tied = list(tied)                           # [(member, representative)], member > representative
        self.known = dict(known)                         # {unknown: c}, members of constraint groups with c != 0
        self.groups = groups                             # list of (members list[int], constraint list[Constraint], root or None)
        self.regulators = regulators                     # constraints with a non-empty gamma
        self._index_map = None
        self._c = None
        self.reordered = False                           # apply_locality_ordering relabelled the dense map

    @classmethod
    def from_dense(cls, index_map: np.ndarray, n_free: int, c: np.ndarray, layout=None, groups=(), regulators=()):
        """A reduction given by its dense arrays (tests of the host restatement)."""
        red = cls(layout, n_free, np.zeros(0, np.int64), [], {int(i): float(c[i]) for i in np.flatnonzero(c)},
                  list(groups), list(regulators))
        red._index_map = np.asarray(index_map, dtype=np.int32)
        red._c = np.asarray(c, dtype=np.float64)
        red.reordered = True                       # the sparse lists do not describe this map
        return red

    @property
    def index_map(self) -> np.ndarray:
        if self._index_map is None:
            N, n_pot = self.layout.size, self.layout.n_potential
            eliminated = np.zeros(N, dtype=bool)
            eliminated[n_pot:] = True
            eliminated[self.elim] = True
            imap = np.full(N, -1, dtype=np.int32)
            keep = ~eliminated
            imap[keep] = np.arange(int(keep.sum()), dtype=np.int32)
            for member, rep in self.tied:
                imap[member] = imap[rep]
            self._index_map = imap
        return self._index_map

    @property
    def c(self) -> np.ndarray:
        if self._c is None:
            c = np.zeros(self.layout.size, dtype=np.float64)
            for x, val in self.known.items():
                c[x] = val
            self._c = c
        return self._c

    def index_of(self, x: int) -> int:
        """Reduced unknown of potential ``x`` (-1: known / multiplier) without the dense map."""
        if self._index_map is not None:
            return int(self._index_map[x])
        if x >= self.layout.n_potential:
            return -1
        for member, rep in self.tied:
            if member == x:
                x = rep
                break
        k = int(np.searchsorted(self.elim, x))
        if k < len(self.elim) and self.elim[k] == x:
            return -1
        return int(x - k)

    @property
    def has_known_part(self) -> bool:
        """Is c non-zero anywhere?  c lives on the members of the constraint groups only: a handful of entries."""
        return any(val != 0.0 for val in self.known.values())

    def _plan(self):
        """Split the index map into long contiguous runs (index_map[i0:i1] == arange(t0, t0 + i1 - i0)), which are
        copied with slices, and the few remaining entries (ground, source-tied groups, run borders).  A reduced
        system is almost the identity with a handful of holes, so at N = 10 M this replaces two boolean-mask
        gathers of 80 MB (45 ms each) by slice copies."""
        plan = getattr(self, "_run_plan", None)
        if plan is not None and plan[0] is self.index_map:
            return plan
        imap = self.index_map
        n = imap.shape[0]
        brk = np.flatnonzero((imap[1:] != imap[:-1] + 1) | (imap[1:] < 0) | (imap[:-1] < 0)) + 1
        starts = np.concatenate([[0], brk])
        ends = np.concatenate([brk, [n]])
        runs, rest = [], []
        if len(starts) <= 4096:
            for i0, i1 in zip(starts.tolist(), ends.tolist()):
                if imap[i0] >= 0 and i1 - i0 >= 256:
                    runs.append((i0, i1, int(imap[i0])))
                elif imap[i0] >= 0 or i1 - i0 > 1:
                    rest.append(np.arange(i0, i1))
            rest = np.concatenate(rest) if rest else np.zeros(0, dtype=np.int64)
            rest = rest[imap[rest] >= 0]
        else:                                   # scattered map (e.g. after the locality reordering): generic path
            runs, rest = [], np.flatnonzero(imap >= 0)
        plan = (imap, runs, rest, imap[rest].astype(np.int64))
        self._run_plan = plan
        return plan

    def expand(self, y: np.ndarray) -> np.ndarray:
        """v (multipliers still zero) from the reduced solution."""
        _, runs, rest, rest_t = self._plan()
        v = self.c.copy()
        for i0, i1, t0 in runs:
            v[i0:i1] += y[t0:t0 + (i1 - i0)]
        if len(rest):
            v[rest] += y[rest_t]
        return v

    def rhs(self, r: np.ndarray, Lc: np.ndarray | None) -> np.ndarray:
        """b = -P^T (r - L c) on the free groups."""
        resid = r if Lc is None else r - Lc
        _, runs, rest, rest_t = self._plan()
        b = np.zeros(self.n_free)
        if len(rest) * 8 > self.n_free:
            b -= np.bincount(rest_t, weights=resid[rest], minlength=self.n_free)
        elif len(rest):
            np.subtract.at(b, rest_t, resid[rest])
        for i0, i1, t0 in runs:
            b[t0:t0 + (i1 - i0)] -= resid[i0:i1]
        return b

    def project(self, vec_rows: dict) -> np.ndarray:
        """P^T applied to a sparse row-indexed vector {row: value}."""
        out = np.zeros(self.n_free)
        for row, val in vec_rows.items():
            t = self.index_of(row)
            if t >= 0:
                out[t] += val
        return out

    def multipliers(self, kcl_residual: np.ndarray, known: dict | None = None) -> dict:
        """Multiplier currents from rho = r - L v (v with zero multipliers).

        Row x of the original system reads  L_x.v + sum_s sigma_xs i_s + sum_k gamma_k[x] i_k = r_x,
        so rho_x = sum_s sigma_xs i_s + sum_k gamma_k[x] i_k.  ``known`` fixes the regulator currents
        that enter through gamma.  Peels each group's tree of constraints leaf by leaf.
        """
        rho = {}
        out = {}
        gam = {}
        if known:
            for cst in self.regulators:
                ik = known.get(cst.index, 0.0)
                for row, g in cst.gamma.items():
                    gam[row] = gam.get(row, 0.0) + g * ik
        for members, cons, root in self.groups:
            if not cons:
                continue
            for x in members:
                rho[x] = float(kcl_residual[x]) - gam.get(x, 0.0)
            # adjacency of the constraint tree
            inc = {x: [] for x in members}
            ground_c = None
            for cst in cons:
                if cst.n < 0:
                    ground_c = cst
                    continue
                inc[cst.p].append(cst)
                inc[cst.n].append(cst)
            done = set()
            leaves = [x for x in members if len(inc[x]) == 1 and x != root]
            while leaves:
                x = leaves.pop()
                live = [cst for cst in inc[x] if cst.index not in done]
                if len(live) != 1:
                    continue
                cst = live[0]
                sign = 1.0 if cst.p == x else -1.0
                i_s = rho[x] / sign
                out[cst.index] = i_s
                done.add(cst.index)
                other = cst.n if cst.p == x else cst.p
                rho[other] -= (-sign) * i_s
                rho[x] = 0.0
                rem = [q for q in inc[other] if q.index not in done]
                if len(rem) == 1 and other != root:
                    leaves.append(other)
            if ground_c is not None:
                out[ground_c.index] = rho[ground_c.p]       # L[g, -1] = 1  (solver.py:559)
        return out


def build_reduction(layout: KKTLayout, pins: list | None = None) -> Reduction:
    """``pins``: unknowns of *floating* components (copper that no path of resistors or sources ties to the ground,
    see :func:`floating_component_pins`) that are held at 0 V like a second ground.  The reference's matrix is
    singular there (its LU returns whatever the rounding leaves); a pin makes the component's block definite, and the
    current it carries is the component's net injected current (reported in ``Reduction.pin_currents``)."""
    N, n_pot = layout.size, layout.n_potential
    uf = _UnionFind()
    ground = layout.ground_constraint
    dirichlet = [ground] + [Constraint(index=-(k + 1), p=int(x), n=-1, value=0.0) for k, x in enumerate(pins or [])]
    all_cons = list(layout.constraints) + dirichlet[1:]
    for cst in all_cons:
        if cst.n < 0:
            uf.find(cst.p)
            continue
        if cst.p == cst.n:
            raise SingularSystemError("voltage source with both terminals on one node")
        if not uf.union(cst.p, cst.n, cst.value):
            raise SingularSystemError("loop of voltage sources: the constraint rows are linearly dependent")
    # collect groups among the nodes touched by constraints
    members: dict = {}
    for x in list(uf.parent.keys()):
        members.setdefault(uf.find(x), []).append(x)
    gcons: dict = {root: [] for root in members}
    for cst in all_cons:
        gcons[uf.find(cst.p)].append(cst)
    # groups whose potential is known outright: root -> (v[root], the unknown the Dirichlet row names)
    known_roots: dict = {}
    for cst in dirichlet:
        root = uf.find(cst.p)
        if root in known_roots:
            if cst is not ground and cst.index < 0:
                gcons[root].remove(cst)            # a pin inside an already grounded group is redundant
                continue
            raise SingularSystemError("two ground rows tie the same group of nodes")
        known_roots[root] = (cst.value - uf.offset(cst.p), cst.p)
    # representative of a free group = its smallest member, so singletons keep their place
    rep_of = {}
    elim, tied, known = [], [], {}
    for root, mem in members.items():
        if root in known_roots:
            v_root = known_roots[root][0]
            for x in mem:
                elim.append(x)
                val = v_root + uf.offset(x)
                if val != 0.0:
                    known[x] = val
        else:
            rep = min(mem)
            rep_of[root] = rep
            for x in mem:
                val = uf.offset(x) - uf.offset(rep)
                if val != 0.0:
                    known[x] = val
                if x != rep:
                    elim.append(x)                # numbered through its representative
                    tied.append((x, rep))
    for x in elim:
        if not 0 <= x < n_pot:
            raise SingularSystemError("constraint names an unknown that is not a potential")
    elim = np.unique(np.asarray(elim, dtype=np.int64))
    n_free = n_pot - len(elim)
    groups = []
    for root, mem in members.items():
        r_node = known_roots[root][1] if root in known_roots else rep_of[root]
        groups.append((sorted(mem), gcons[root], r_node))
    regs = [cst for cst in layout.constraints if cst.gamma]
    return Reduction(layout=layout, n_free=n_free, elim=elim, tied=sorted(tied), known=known, groups=groups, regulators=regs)


def floating_component_pins(n_potential: int, ground: int, constraints, *, mesh_offsets=None, links=None,
                            matrix=None) -> list:
    """One unknown (the smallest) of every connected component of potentials that does not contain the ground.

    Connectivity = shared mesh (a mesh is one connected triangulation), resistor stamps (``links``: pairs of
    unknowns), voltage-source / regulator-output ties (constraints).  With a bare ``matrix`` (reference layout, host
    CSR) the components are those of its potential block.  Work is O(#meshes + #lumped elements), or one
    ``connected_components`` pass for a bare matrix."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import connected_components
    ties = [(c.p, c.n) for c in constraints if c.n >= 0]
    if matrix is not None:
        blk = sp.csr_matrix(matrix)[:n_potential, :n_potential]
        if ties:
            t = np.asarray(ties, dtype=np.int64)
            blk = blk + sp.coo_matrix((np.ones(len(t)), (t[:, 0], t[:, 1])), shape=blk.shape)
        n_comp, label = connected_components(blk, directed=False)
        if n_comp <= 1:
            return []
        first = np.full(n_comp, n_potential, dtype=np.int64)
        np.minimum.at(first, label, np.arange(n_potential))
        return [int(first[k]) for k in range(n_comp) if k != label[ground]]
    offs = np.asarray(mesh_offsets if mesh_offsets is not None else [0], dtype=np.int64)
    n_vert, n_mesh = int(offs[-1]), len(offs) - 1
    n_super = n_mesh + (n_potential - n_vert)

    def super_of(u):
        u = np.asarray(u, dtype=np.int64)
        return np.where(u < n_vert, np.searchsorted(offs, u, side="right") - 1, n_mesh + (u - n_vert))

    pairs = [np.zeros((0, 2), dtype=np.int64)]
    if links is not None and len(links):
        pairs.append(np.asarray(links, dtype=np.int64).reshape(-1, 2))
    if ties:
        pairs.append(np.asarray(ties, dtype=np.int64))
    e = np.concatenate(pairs)
    a, b = super_of(e[:, 0]), super_of(e[:, 1])
    n_comp, label = connected_components(sp.coo_matrix((np.ones(len(a)), (a, b)), shape=(n_super, n_super)), directed=False)
    sizes = np.concatenate([np.diff(offs), np.ones(n_potential - n_vert, dtype=np.int64)])
    first_unknown = np.concatenate([offs[:-1], np.arange(n_vert, n_potential, dtype=np.int64)])
    g_label = label[int(super_of(ground))]
    pins = {}
    for sn in range(n_super):
        if sizes[sn] == 0 or label[sn] == g_label:
            continue
        pins[label[sn]] = min(pins.get(label[sn], n_potential), int(first_unknown[sn]))
    return sorted(pins.values())


# ---- locality ordering ------------------------------------------------------------------------------------

def _spread_bits(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.uint64)
    v = (v | (v << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
    v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
    v = (v | (v << np.uint64(2))) & np.uint64(0x3333333333333333)
    v = (v | (v << np.uint64(1))) & np.uint64(0x5555555555555555)
    return v


def morton_keys(xy: np.ndarray, bits: int = 16) -> np.ndarray:
    """Z-order key of every point (quantised to `bits` per axis inside the bounding box)."""
    lo = xy.min(axis=0)
    span = np.maximum(xy.max(axis=0) - lo, 1e-300)
    q = np.minimum(((xy - lo) / span * (2 ** bits - 1)).astype(np.uint64), np.uint64(2 ** bits - 1))
    return _spread_bits(q[:, 0]) | (_spread_bits(q[:, 1]) << np.uint64(1))


def ordering_is_scattered(tri: np.ndarray, n_vert: int) -> bool:
    """True if mesh neighbours are far apart in the numbering (CGAL insertion order, random order):
    mean index distance along triangle edges well beyond the O(sqrt(n)) of a scan-line numbering."""
    if len(tri) == 0 or n_vert < 4096:
        return False
    step = max(1, len(tri) // 200000)
    t = tri[::step].astype(np.int64)
    mean_dist = float(np.mean(np.abs(t[:, 0] - t[:, 1]) + np.abs(t[:, 1] - t[:, 2])) / 2)
    return mean_dist > 8.0 * np.sqrt(n_vert)


def strip_index(xy: np.ndarray, mesh_id: np.ndarray) -> np.ndarray:
    """Horizontal strip of every point, per mesh: strips are about three mean vertex spacings high
    (3.4 * sqrt(bounding-box area / n)), so the neighbours of a vertex lie in its own strip or the two adjacent ones.
    Sorting by (strip, x) then gives a *band* numbering: the columns of 64 consecutive rows fall into three short
    index ranges, which is what the x-window path of the SpMV needs (on a random Delaunay mesh 96 % of the 64-row
    tiles are covered by 3 runs of 128 entries; in Z-order none are) and what a scan-line numbering has anyway."""
    strip = np.zeros(len(xy), dtype=np.int64)
    mesh_id = np.asarray(mesh_id)
    if len(xy) == 0:
        return strip
    cuts = np.flatnonzero(mesh_id[1:] != mesh_id[:-1]) + 1
    if len(cuts) + 1 != len(np.unique(mesh_id[np.concatenate([[0], cuts])])):
        segments = [np.flatnonzero(mesh_id == m) for m in np.unique(mesh_id)]      # meshes interleaved: index lists
    else:
        edges = np.concatenate([[0], cuts, [len(xy)]])                              # contiguous blocks: views, no copies
        segments = [slice(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:])]
    for sel in segments:
        px, py = xy[sel, 0], xy[sel, 1]
        if len(px) < 2:
            continue
        y0 = float(py.min())
        area = max((float(px.max()) - float(px.min())) * (float(py.max()) - y0), 1e-300)
        height = 3.4 * np.sqrt(area / len(px))
        strip[sel] = np.floor((py - y0) / height).astype(np.int64)
    return strip


def _strip_order(n_free: int, has: np.ndarray, mesh_id: np.ndarray, xy: np.ndarray, owner: np.ndarray) -> np.ndarray:
    """Permutation (new position -> old reduced index) that sorts the mesh unknowns by (mesh, strip, x) and leaves
    the others behind them in their old order.  One sort of 64-bit keys [mesh:16 | strip:16 | x:32] -- numpy's
    vectorised quicksort is 5x faster than ``lexsort`` on three keys, which was 60 % of ``solve_system`` on a 2 M-vertex
    unstructured mesh; equal keys (two vertices of one strip with the same quantised x) are then put in index order, so
    that the numbering does not depend on the sort implementation."""
    strip = np.zeros(n_free, dtype=np.int64)
    pts = xy[owner[has]]
    m = mesh_id[has]
    strip[has] = strip_index(pts, m)
    n_mesh = int(m.max()) + 1 if has.any() else 0
    if n_mesh >= 0xFFFF or (has.any() and int(strip.max()) >= 0x10000) or n_free >= 2 ** 32:
        k_mesh = np.where(has, mesh_id, np.int64(2 ** 40))
        k_x = np.arange(n_free, dtype=np.float64)
        k_x[has] = pts[:, 0]
        return np.lexsort((k_x, strip, k_mesh))
    key = np.arange(n_free, dtype=np.uint64) | (np.uint64(0xFFFF) << np.uint64(48))     # non-mesh unknowns: last, old order
    if has.any():
        x = pts[:, 0]
        lo = np.full(n_mesh, np.inf)
        hi = np.full(n_mesh, -np.inf)
        if n_mesh > 64:
            np.minimum.at(lo, m, x)
            np.maximum.at(hi, m, x)
        else:
            for k in range(n_mesh):
                sel = m == k
                if sel.any():
                    lo[k], hi[k] = x[sel].min(), x[sel].max()
        span = np.maximum(hi - lo, 1e-300)
        xq = np.minimum((x - lo[m]) / span[m] * (2.0 ** 32 - 1), 2.0 ** 32 - 1).astype(np.uint64)
        key[has] = (m.astype(np.uint64) << np.uint64(48)) | (strip[has].astype(np.uint64) << np.uint64(32)) | xq
    order = np.argsort(key)
    ks = key[order]
    tie = np.flatnonzero(ks[1:] == ks[:-1])
    if tie.size:
        gid = np.cumsum(np.concatenate([[True], ks[1:] != ks[:-1]])) - 1
        in_tie = np.zeros(n_free, dtype=bool)
        in_tie[tie] = True
        in_tie[tie + 1] = True
        sub = np.flatnonzero(in_tie)
        order[sub] = order[sub][np.lexsort((order[sub], gid[sub]))]
    return order


def apply_locality_ordering(red: "Reduction", xy: np.ndarray, mesh_offsets: np.ndarray, kind: str = "strip") -> None:
    """Relabel the reduced unknowns so that vertices close in space are close in index (``kind="strip"``: band
    numbering by horizontal strips, see ``strip_index``; ``"morton"``: Z-order) inside each mesh; mesh blocks stay
    contiguous and in order; non-mesh unknowns keep their place behind them.
    Purely internal: ``index_map`` is the only thing that changes, ``expand``/``rhs``/``project`` follow it."""
    n_vert = int(mesh_offsets[-1])
    imap = red.index_map
    owner = np.full(red.n_free, -1, dtype=np.int64)           # representative vertex of every reduced unknown
    vert = np.flatnonzero(imap[:n_vert] >= 0)
    # the first vertex that maps to a group represents it (groups are numbered by their smallest member)
    first = np.unique(imap[vert], return_index=True)
    owner[first[0]] = vert[first[1]]
    mesh_id = np.searchsorted(mesh_offsets, np.maximum(owner, 0), side="right") - 1
    has = owner >= 0
    if kind == "morton":
        key = np.full(red.n_free, np.uint64(2 ** 63), dtype=np.uint64)    # non-mesh unknowns sort last, stably
        mk = morton_keys(xy[owner[has]])
        key[has] = (mesh_id[has].astype(np.uint64) << np.uint64(40)) | mk
        order = np.argsort(key, kind="stable")                            # new position -> old reduced index
    else:
        order = _strip_order(red.n_free, has, mesh_id, xy, owner)
    new_of_old = np.empty(red.n_free, dtype=np.int32)
    new_of_old[order] = np.arange(red.n_free, dtype=np.int32)
    free = imap >= 0
    imap[free] = new_of_old[imap[free]]
    red.reordered = True
