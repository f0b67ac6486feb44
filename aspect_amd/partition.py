"""Domain decomposition of a box hierarchy: one rank <-> one GPU <-> one subdomain.

The reference partitions the p4est forest along its Morton curve and lets deal.II number the
locally owned DoFs of every level contiguously, ghosts behind them
(stokes_matrix_free_local_smoothing.cc:1548-1700, ``setup_dofs``).  This module does
the same for the structured stand-in host of ``boxmesh``: rank r owns a contiguous run of coarse
cells (forest trees) and all their descendants on every level, so every level is balanced and a
parent always lives on the rank of its children.

Per rank and level it produces the flat tables of ``agmg_level_desc`` in LOCAL numbering:

* local cells = owned cells (a contiguous range of the global Morton order);
* local nodes = every node an owned cell touches; a node is owned by the lowest rank touching it;
  owned nodes come first (ascending global index), ghosts behind them grouped by owner rank
  (ascending global index inside a group);
* the ghost exchange plan: per neighbour rank the owned nodes it ghosts (``send_nodes``, in the
  order of its ghost group) and the offset of its group among my ghosts (``recv_offsets``).
  ``update ghosts`` sends owner -> ghost along it, ``compress`` sends the ghost partial sums back.
"""
from __future__ import annotations

import numpy as np


def default_replicated_levels(n_global_cells, max_cells=32768):
    """How many of the coarsest levels every rank holds whole (agmg_level_desc::replicated): those with at most
    `max_cells` cells in the whole mesh -- a level that small costs a GPU its launch latency whether it holds an
    eighth of it or all of it, and replicated it needs no message at all ("levels <= 3 have <= 512 cells
    and go to the host or are gathered on one rank").  The default follows from measured kernel times on one MI355X: an
    A-block application costs 23 us on 32768 cells and 15 us on 4096 (launch-bound), so holding a 32768-cell level whole
    beats an eighth of it plus two message rounds as soon as a round costs more than 4 us; at 262144 cells (93 us
    against 23 us) it does not.  Never the finest level.  n_global_cells: per level, coarsest first."""
    k = 0
    while k < len(n_global_cells) - 1 and n_global_cells[k] <= max_cells:
        k += 1
    return k


def cell_ranks(glev, size):
    """Owner rank of every cell of a global level (contiguous runs of coarse cells)."""
    # trees of the forest: the coarse cells of a box (reps), or of another geometry (n_trees: the shell's 96)
    n_coarse = int(getattr(glev, "n_trees", 0)) or int(np.prod(glev.reps))
    if n_coarse % size:
        raise ValueError(f"{n_coarse} coarse cells do not split over {size} ranks")
    per_coarse = glev.n_cells // n_coarse
    return (np.arange(glev.n_cells, dtype=np.int64) // per_coarse) // (n_coarse // size)


def _touch_and_owner(cell_nodes, crank, n_nodes, size):
    touch = np.zeros(n_nodes, dtype=np.uint64)
    for r in range(size):
        touch[np.unique(cell_nodes[crank == r])] |= np.uint64(1) << np.uint64(r)
    owner = np.full(n_nodes, -1, dtype=np.int64)
    for r in range(size - 1, -1, -1):
        owner[(touch >> np.uint64(r)) & np.uint64(1) == 1] = r
    return touch, owner


class _Space:
    """Local numbering + exchange plan of one node space (Q2 velocity or Q1 pressure nodes)."""

    def __init__(self, cell_nodes, crank, n_nodes, rank, size):
        touch, owner = _touch_and_owner(cell_nodes, crank, n_nodes, size)
        mine = ((touch >> np.uint64(rank)) & np.uint64(1)) == 1
        owned = np.flatnonzero(mine & (owner == rank))
        ghosts = np.flatnonzero(mine & (owner != rank))
        ghosts = ghosts[np.lexsort((ghosts, owner[ghosts]))]
        self.l2g = np.concatenate([owned, ghosts]).astype(np.int64)
        self.n_owned = owned.size
        self.ghost_owner = owner[ghosts]
        g2l = np.full(n_nodes, -1, dtype=np.int64)
        g2l[self.l2g] = np.arange(self.l2g.size)
        self.g2l = g2l
        self.touch_local = touch[self.l2g]
        self.rank, self.size = rank, size
        # ranks ghosting my owned nodes
        self.export = {}
        otouch = touch[owned]
        for k in range(size):
            if k == rank:
                continue
            sel = ((otouch >> np.uint64(k)) & np.uint64(1)) == 1
            if sel.any():
                self.export[k] = np.flatnonzero(sel)          # local (owned) indices, ascending global
        self.neighbors = set(self.export) | set(int(o) for o in np.unique(self.ghost_owner))

    def sym_plan(self):
        """The symmetric exchange plan (agmg_level_desc::sym_*): per rank that touches a node of mine, ascending, my
        local nodes it touches too, ordered by global index -- both sides list the shared nodes in the same order, so
        entry i of my message to rank k and entry i of its message to me belong to the same node."""
        nbrs, off, nodes = [], [0], []
        for k in range(self.size):
            if k == self.rank:
                continue
            idx = np.flatnonzero(((self.touch_local >> np.uint64(k)) & np.uint64(1)) == 1)
            if idx.size:
                idx = idx[np.argsort(self.l2g[idx], kind="stable")]
                nbrs.append(k)
                nodes.append(idx)
                off.append(off[-1] + idx.size)
        return (np.asarray(nbrs, dtype=np.int32), np.asarray(off, dtype=np.int32),
                (np.concatenate(nodes) if nodes else np.zeros(0)).astype(np.uint32))

    def plan(self, neighbors):
        send_off, send_nodes, recv_off = [0], [], []
        for k in neighbors:
            e = self.export.get(k, np.zeros(0, dtype=np.int64))
            send_nodes.append(e)
            send_off.append(send_off[-1] + e.size)
            recv_off.append(int(np.searchsorted(self.ghost_owner, k, side="left")))
        recv_off.append(int(self.ghost_owner.size))
        # every ghost owner is a neighbour, so consecutive offsets delimit each neighbour's group
        return (np.asarray(send_off, dtype=np.int32),
                (np.concatenate(send_nodes) if send_nodes else np.zeros(0)).astype(np.uint32),
                np.asarray(recv_off, dtype=np.int32))


class _LocalVectors:
    """local <-> global vector helpers shared by the two ways a rank's level comes about (needs self.v.l2g,
    self.p.l2g, the owned counts and the constraint masks)."""

    @property
    def n_dofs_v(self):
        return self.dim * self.n_nodes_v

    @property
    def n_dofs_p(self):
        return self.n_nodes_p

    def constrained_dofs_v(self):
        bits = (self.cmask_v[:, None] >> np.arange(self.dim)[None, :]) & 1
        return bits.reshape(-1).astype(bool)

    # local <-> global vectors (velocity vectors are [node][component])
    def to_local_v(self, g):
        return np.ascontiguousarray(g.reshape(-1, self.dim)[self.v.l2g].reshape(-1))

    def to_local_p(self, g):
        return np.ascontiguousarray(g[self.p.l2g])

    def owned_dofs_v(self):
        """(local dof indices, global dof indices) of the owned velocity DoFs."""
        n, d = self.n_owned_nodes_v, self.dim
        loc = np.arange(n * d)
        glo = (self.v.l2g[:n, None] * d + np.arange(d)[None, :]).reshape(-1)
        return loc, glo

    def owned_dofs_p(self):
        n = self.n_owned_nodes_p
        return np.arange(n), self.p.l2g[:n]

class LocalLevel(_LocalVectors):
    """One rank's share of a ``boxmesh.Level``; duck-types it for ``capi.Solver``."""

    def __init__(self, glev, rank, size, replicated=False, parent_is_replicated=False):
        self.dim, self.level = glev.dim, glev.level
        self.replicated = bool(replicated) and size > 1
        if self.replicated:          # the whole level, numbered as one rank would number it; no exchange plan
            rank, size = 0, 1
        self.rank, self.size = rank, size
        crank = cell_ranks(glev, size)
        self.cells = np.flatnonzero(crank == rank)            # global cell ids, contiguous
        assert self.cells.size and np.all(np.diff(self.cells) == 1)
        self.n_cells = int(self.cells.size)
        self.v = _Space(glev.cell_nodes_v.astype(np.int64), crank, glev.n_nodes_v, rank, size)
        self.p = _Space(glev.cell_nodes_p.astype(np.int64), crank, glev.n_nodes_p, rank, size)
        self.n_nodes_v, self.n_owned_nodes_v = int(self.v.l2g.size), int(self.v.n_owned)
        self.n_nodes_p, self.n_owned_nodes_p = int(self.p.l2g.size), int(self.p.n_owned)
        self.cell_nodes_v = self.v.g2l[glev.cell_nodes_v[self.cells].astype(np.int64)].astype(np.uint32)
        self.cell_nodes_p = self.p.g2l[glev.cell_nodes_p[self.cells].astype(np.int64)].astype(np.uint32)
        self.cmask_v = glev.cmask_v[self.v.l2g]
        self.cmask_p = glev.cmask_p[self.p.l2g]
        self.cell_h = glev.cell_h[self.cells]
        self.general = getattr(glev, "general", False)
        if self.general:
            self.inv_jacobian = glev.inv_jacobian[self.cells]
            self.jxw = glev.jxw[self.cells]
        self.neighbors = sorted(self.v.neighbors | self.p.neighbors)
        self.send_offsets_v, self.send_nodes_v, self.recv_offsets_v = self.v.plan(self.neighbors)
        self.send_offsets_p, self.send_nodes_p, self.recv_offsets_p = self.p.plan(self.neighbors)
        self.sym_neighbors_p, self.sym_offsets_p, self.sym_nodes_p = self.p.sym_plan()
        self.sym_neighbors_v, self.sym_offsets_v, self.sym_nodes_v = self.v.sym_plan()
        self.global_nodes_v = self.v.l2g.astype(np.uint32)
        self.global_nodes_p = self.p.l2g.astype(np.uint32)
        self.n_global_nodes_v, self.n_global_nodes_p = glev.n_nodes_v, glev.n_nodes_p
        if parent_is_replicated and size > 1:
            # the first distributed level: its parents are cells of a level held whole (global cell order: the 2^dim
            # children of a cell are consecutive)
            nch = 1 << self.dim
            self.parent_cell = (self.cells // nch).astype(np.int32)
            self.child_index = (self.cells % nch).astype(np.uint8)


class LocalHierarchy:
    def __init__(self, ghier, rank, size, n_replicated=0):
        self.dim = ghier.dim
        self.rank, self.size = rank, size
        self.n_replicated = min(int(n_replicated), len(ghier.levels) - 1) if size > 1 else 0
        self.levels = [LocalLevel(g, rank, size, replicated=l < self.n_replicated,
                                  parent_is_replicated=(l == self.n_replicated and l > 0))
                       for l, g in enumerate(ghier.levels)]

    @property
    def finest(self):
        return self.levels[-1]

    @property
    def n_levels(self):
        return len(self.levels)


class LocalProblem:
    """Rank-local view of a built ``cases.StokesProblem``: local tables, coefficient tables and
    right-hand side of the rank's subdomain (what each MPI rank of the reference holds)."""

    def __init__(self, gproblem, rank, size, n_replicated=0):
        self.g = gproblem
        self.rank, self.size = rank, size
        for k in ("name", "dim", "tol", "restart", "n_cheap", "n_expensive", "krylov", "idr_s",
                  "pressure_scaling", "eta_min", "eta_max", "is_compressible"):
            setattr(self, k, getattr(gproblem, k, None))
        self.hier = LocalHierarchy(gproblem.hier, rank, size, n_replicated)
        self.eta_levels = [np.ascontiguousarray(t[lev.cells]) for t, lev in zip(gproblem.eta_levels, self.hier.levels)]
        self.eta_active = np.ascontiguousarray(gproblem.eta_active[self.hier.finest.cells])
        fin = self.hier.finest
        # free-surface faces of this rank's cells (every rank passes its list, possibly empty: the
        # call is collective in a distributed context)
        self.free_surface = None
        fs = getattr(gproblem, "free_surface", None)
        if fs is not None:
            cells_g, fnos, stab, normal, jxw = fs
            lut = -np.ones(gproblem.hier.finest.n_cells, dtype=np.int64)
            lut[np.asarray(fin.cells, dtype=np.int64)] = np.arange(len(fin.cells))
            loc = lut[np.asarray(cells_g, dtype=np.int64)]
            keep = loc >= 0
            self.free_surface = (loc[keep].astype(np.uint32), fnos[keep], stab[keep],
                                 None if normal is None else normal[keep], None if jxw is None else jxw[keep])
        if hasattr(gproblem, "rhs_u"):
            self.rhs_u = fin.to_local_v(gproblem.rhs_u)
            self.rhs_p = fin.to_local_p(gproblem.rhs_p)
            self.p0 = fin.to_local_p(gproblem.p0)

    def to_local(self, g):
        """global block vector [u; p] -> local block vector (owned and ghost entries filled)."""
        gf, fin = self.g.hier.finest, self.hier.finest
        return np.concatenate([fin.to_local_v(g[:gf.n_dofs_v]), fin.to_local_p(g[gf.n_dofs_v:])])

    def owned_block(self):
        """(local indices, global indices) of the owned entries of a block vector [u; p]."""
        gf, fin = self.g.hier.finest, self.hier.finest
        lu, gu = fin.owned_dofs_v()
        lp, gp = fin.owned_dofs_p()
        return np.concatenate([lu, fin.n_dofs_v + lp]), np.concatenate([gu, gf.n_dofs_v + gp])


# ---------------------------------------------------------------------------
# Rank-local set-up of a structured box: the same tables as LocalLevel / LocalProblem, generated from the rank's own
# subdomain in closed form -- no rank ever holds the global mesh (what each MPI rank of the reference does:
# p4est hands it its octants, deal.II numbers its own DoFs, core.cc:141-154, ...local_smoothing.cc:1548-1700).
# Host memory and set-up time per rank depend on the cells of the rank, not on the number of ranks.
# ---------------------------------------------------------------------------
class _ClosedFormSpace:
    """_Space of a node lattice (Q2: 2 lattice steps per cell, Q1: 1) for the cells of `rank`, from lattice
    coordinates alone: who touches a node follows from the coarse cells whose closed extent contains it."""

    def __init__(self, cell_lattice_ijk, n1, step, nd, reps, l, rank, size):
        dim = cell_lattice_ijk.shape[1]
        strides = np.ones(dim, dtype=np.int64)
        for d in range(1, dim):
            strides[d] = strides[d - 1] * nd[d - 1]
        loc = np.indices((n1,) * dim).reshape(dim, -1)[::-1]         # x fastest
        gid = np.zeros((cell_lattice_ijk.shape[0], n1 ** dim), dtype=np.int64)
        for d in range(dim):
            gid += (step * cell_lattice_ijk[:, d:d + 1] + loc[d][None, :]) * strides[d]
        uniq, inv = np.unique(gid.reshape(-1), return_inverse=True)   # the nodes this rank touches, ascending
        # lattice coordinates of the touched nodes
        X = []
        rem = uniq
        for d in range(dim):
            X.append(rem % nd[d])
            rem = rem // nd[d]
        # coarse cells containing a node: one per direction, two on an inner coarse face
        S = step << l                                                  # lattice steps per coarse cell
        n_coarse = int(np.prod(reps))
        per_rank = n_coarse // size
        lo, hi = [], []
        for d in range(dim):
            c_hi = np.minimum(X[d] // S, reps[d] - 1)
            on_face = (X[d] % S == 0) & (X[d] > 0) & (X[d] < nd[d] - 1)
            lo.append(np.where(on_face, c_hi - 1, c_hi))
            hi.append(c_hi)
        touch = np.zeros(uniq.size, dtype=np.uint64)
        owner = np.full(uniq.size, size, dtype=np.int64)
        for combo in range(1 << dim):
            cid = np.zeros(uniq.size, dtype=np.int64)
            mul = 1
            for d in range(dim):
                cid += (hi[d] if (combo >> d) & 1 else lo[d]) * mul
                mul *= int(reps[d])
            r = cid // per_rank
            touch |= np.uint64(1) << r.astype(np.uint64)
            owner = np.minimum(owner, r)
        assert np.all((touch >> np.uint64(rank)) & np.uint64(1) == 1)
        own_sel = owner == rank
        owned = np.flatnonzero(own_sel)
        ghosts = np.flatnonzero(~own_sel)
        ghosts = ghosts[np.lexsort((uniq[ghosts], owner[ghosts]))]
        order = np.concatenate([owned, ghosts])                         # local index -> position in uniq
        self.l2g = uniq[order]
        self.n_owned = int(owned.size)
        self.ghost_owner = owner[ghosts]
        pos2loc = np.empty(uniq.size, dtype=np.int64)
        pos2loc[order] = np.arange(uniq.size)
        self.cell_nodes = pos2loc[inv].reshape(gid.shape).astype(np.uint32)
        self.lattice = np.stack([x[order] for x in X], axis=1)        # lattice coordinates of the local nodes
        self.touch_local = touch[order]
        self.rank, self.size = rank, size
        self.export = {}
        otouch = touch[owned]
        for k in range(size):
            if k == rank:
                continue
            sel = ((otouch >> np.uint64(k)) & np.uint64(1)) == 1
            if sel.any():
                self.export[k] = np.flatnonzero(sel)
        self.neighbors = set(self.export) | set(int(o) for o in np.unique(self.ghost_owner))

    plan = _Space.plan
    sym_plan = _Space.sym_plan


class RankLocalBoxLevel(_LocalVectors):
    """Level `l` of a box of `reps` coarse cells as rank `rank` of `size` sees it (a contiguous run of coarse cells
    and their descendants, cells in Morton order inside a coarse cell): the tables of LocalLevel(boxmesh.Level(...))
    entry for entry, and enough of boxmesh.Level's interface (cell_ijk, h, qpoint_coords, node coordinates) for the
    coefficient and right-hand side builders to run on the rank's cells.  Cartesian, non-periodic, lexicographic
    global numbering."""

    def __init__(self, dim, extents, reps, l, bc, rank, size, replicated=False, parent_is_replicated=False):
        from . import boxmesh as bm
        self.replicated = bool(replicated) and size > 1
        true_rank, true_size = rank, size
        if self.replicated:          # the whole level, as one rank would hold it; no exchange plan
            rank, size = 0, 1
        self.dim, self.level, self.rank, self.size = dim, l, rank, size
        self.extents = np.asarray(extents, dtype=np.float64)
        self.reps = np.asarray(reps, dtype=np.int64)
        self.ncd = self.reps * (1 << l)
        self.h = self.extents / self.ncd
        n_coarse = int(np.prod(self.reps))
        if n_coarse % size:
            raise ValueError(f"{n_coarse} coarse cells do not split over {size} ranks")
        per_coarse = 1 << (dim * l)
        per_rank = n_coarse // size
        self.cells = np.arange(rank * per_rank * per_coarse, (rank + 1) * per_rank * per_coarse, dtype=np.int64)
        self.n_cells = int(self.cells.size)
        self.n_global_cells = n_coarse * per_coarse
        coarse = self.cells // per_coarse
        local = bm._morton_decode(self.cells % per_coarse, dim, l)
        cc, rem = [], coarse
        for d in range(dim):
            cc.append(rem % self.reps[d])
            rem = rem // self.reps[d]
        self.cell_ijk = np.stack([cc[d] * (1 << l) + local[d] for d in range(dim)], axis=1)
        self.nvd, self.npd = 2 * self.ncd + 1, self.ncd + 1
        self.n_global_nodes_v, self.n_global_nodes_p = int(np.prod(self.nvd)), int(np.prod(self.npd))
        self.v = _ClosedFormSpace(self.cell_ijk, 3, 2, self.nvd, self.reps, l, rank, size)
        self.p = _ClosedFormSpace(self.cell_ijk, 2, 1, self.npd, self.reps, l, rank, size)
        self.cell_nodes_v, self.cell_nodes_p = self.v.cell_nodes, self.p.cell_nodes
        self.n_nodes_v, self.n_owned_nodes_v = int(self.v.l2g.size), self.v.n_owned
        self.n_nodes_p, self.n_owned_nodes_p = int(self.p.l2g.size), self.p.n_owned
        # constraint masks (boxmesh.Level._constraint_mask on the rank's nodes)
        mask = np.zeros(self.n_nodes_v, dtype=np.uint8)
        allbits = (1 << dim) - 1
        for d in range(dim):
            for side in range(2):
                code = bc[2 * d + side]
                on = self.v.lattice[:, d] == (0 if side == 0 else self.nvd[d] - 1)
                if code == bm.BC_ZERO:
                    mask[on] |= allbits
                elif code == bm.BC_TANGENTIAL:
                    mask[on] |= (1 << d)
                elif code >= bm.BC_COMPONENTS:
                    mask[on] |= (code - bm.BC_COMPONENTS) & allbits
        self.cmask_v = mask
        self.cmask_p = np.zeros(self.n_nodes_p, dtype=np.uint8)
        self.cell_h = np.tile(self.h, (self.n_cells, 1))
        self.general = False
        self.periodic = [False] * dim
        self.neighbors = sorted(self.v.neighbors | self.p.neighbors)
        self.send_offsets_v, self.send_nodes_v, self.recv_offsets_v = self.v.plan(self.neighbors)
        self.send_offsets_p, self.send_nodes_p, self.recv_offsets_p = self.p.plan(self.neighbors)
        self.sym_neighbors_p, self.sym_offsets_p, self.sym_nodes_p = self.p.sym_plan()
        self.sym_neighbors_v, self.sym_offsets_v, self.sym_nodes_v = self.v.sym_plan()
        self.global_nodes_v = self.v.l2g.astype(np.uint32)
        self.global_nodes_p = self.p.l2g.astype(np.uint32)
        if parent_is_replicated and true_size > 1:
            nch = 1 << dim
            self.parent_cell = (self.cells // nch).astype(np.int32)
            self.child_index = (self.cells % nch).astype(np.uint8)

    # -- the part of boxmesh.Level's interface the table builders use, on the rank's cells and nodes
    def node_coords_v(self):
        return self.v.lattice * (self.h / 2)[None, :]

    def node_coords_p(self):
        return self.p.lattice * self.h[None, :]

    def qpoint_coords(self, cell_ijk=None):
        from . import boxmesh as bm
        ijk = self.cell_ijk if cell_ijk is None else cell_ijk
        q = np.indices((3,) * self.dim).reshape(self.dim, -1)[::-1]
        out = np.zeros((ijk.shape[0], 3 ** self.dim, self.dim))
        for d in range(self.dim):
            out[:, :, d] = (ijk[:, d:d + 1] + bm.GAUSS3_X[q[d]][None, :]) * self.h[d]
        return out

    def halo_cells(self):
        """Lattice positions [n, dim] of the cells of OTHER ranks that touch a node of this rank (one layer around the
        subdomain): their element vectors complete the sums at the rank's shared nodes without any exchange."""
        dim = self.dim
        shared = np.zeros(self.n_nodes_v, dtype=bool)
        shared[self.n_owned_nodes_v:] = True
        for e in self.v.export.values():
            shared[e] = True
        surf = self.cell_ijk[shared[self.cell_nodes_v.astype(np.int64)].any(axis=1)]
        if surf.shape[0] == 0:
            return np.zeros((0, dim), dtype=np.int64)
        offs = np.indices((3,) * dim).reshape(dim, -1).T - 1
        cand = (surf[:, None, :] + offs[None, :, :]).reshape(-1, dim)
        ok = np.all((cand >= 0) & (cand < self.ncd[None, :]), axis=1)
        cand = cand[ok]
        mul = np.concatenate([[1], np.cumprod(self.ncd[:-1])]).astype(np.int64)
        ids = np.unique(cand @ mul)
        ids = ids[~np.isin(ids, self.cell_ijk @ mul)]
        out = np.zeros((ids.size, dim), dtype=np.int64)
        rem = ids
        for d in range(dim):
            out[:, d] = rem % self.ncd[d]
            rem = rem // self.ncd[d]
        return out

    def local_nodes_of(self, cell_ijk):
        """Local Q2 node indices [n, 3^dim] of arbitrary cells of the level (-1: not a node of this rank)."""
        dim = self.dim
        strides = np.ones(dim, dtype=np.int64)
        for d in range(1, dim):
            strides[d] = strides[d - 1] * self.nvd[d - 1]
        loc = np.indices((3,) * dim).reshape(dim, -1)[::-1]
        gid = np.zeros((cell_ijk.shape[0], 3 ** dim), dtype=np.int64)
        for d in range(dim):
            gid += (2 * cell_ijk[:, d:d + 1] + loc[d][None, :]) * strides[d]
        srt = np.argsort(self.v.l2g, kind="stable")
        pos = np.searchsorted(self.v.l2g[srt], gid)
        pos = np.minimum(pos, srt.size - 1)
        hit = self.v.l2g[srt][pos] == gid
        return np.where(hit, srt[pos], -1)


class RankLocalHierarchy:
    def __init__(self, dim, refinement, extents, reps, bc, rank, size, only_finest=False, n_replicated=0):
        self.dim, self.rank, self.size = dim, rank, size
        first = refinement if only_finest else 0
        self.n_replicated = 0 if (only_finest or size == 1) else min(int(n_replicated), refinement)
        self.levels = [RankLocalBoxLevel(dim, extents, reps, l, bc, rank, size, replicated=l < self.n_replicated,
                                         parent_is_replicated=(l == self.n_replicated and l > 0))
                       for l in range(first, refinement + 1)]

    @property
    def finest(self):
        return self.levels[-1]

    @property
    def n_levels(self):
        return len(self.levels)


class RankLocalProblem:
    """LocalProblem without a global problem: built from an UNBUILT ``cases.StokesProblem`` of a Cartesian box.
    Coefficient tables come from the material at the rank's own quadrature points; the right-hand side from the
    rank's cells plus the one layer of neighbouring cells that share its nodes (same sums as the global assembly up
    to their order).  eta_min / eta_max are the rank's own (reporting only)."""

    def __init__(self, case, rank, size, only_finest=False, n_replicated=0, gather=None):
        """n_replicated: the levels 0..n_replicated-1 are held whole by every rank (agmg_level_desc::replicated); their
        coefficient tables need the cell values of the coarsest distributed level of ALL ranks: `gather(array)` returns
        the arrays of all ranks concatenated in rank order (parallel.Ranks.gather_concat)."""
        from . import boxmesh as bm
        if n_replicated and size > 1 and gather is None:
            raise ValueError("replicated levels need a gather over the ranks for their coefficient tables")
        if case.deform is not None or case.periodic or case.make_hierarchy is not None or case.numbering != "lexicographic":
            raise ValueError("rank-local set-up: Cartesian non-periodic boxes with lexicographic numbering only")
        if case.boundary_traction or case.dilation_rhs is not None or case.free_surface_spec or case.boundary_velocity or \
           case.eta_q is not None or case.pressure_scaling_from_mesh is not None:
            raise ValueError("rank-local set-up: this deck needs tables that are only built from the global mesh")
        self.case, self.rank, self.size = case, rank, size
        for k in ("name", "dim", "tol", "restart", "n_cheap", "n_expensive", "krylov", "idr_s", "pressure_scaling", "is_compressible"):
            setattr(self, k, getattr(case, k, None))
        dim = case.dim
        extents = [1.0] * dim if case.extents is None else case.extents
        reps = [1] * dim if case.reps is None else case.reps
        self.hier = RankLocalHierarchy(dim, case.refinement, extents, reps, case.bc, rank, size, only_finest, n_replicated)
        fin = self.hier.finest
        if only_finest:
            nq = 3 ** dim
            eta_q = case.eta(fin.qpoint_coords().reshape(-1, dim)).reshape(fin.n_cells, nq)
            self.eta_active = (nq / np.sum(1.0 / eta_q, axis=1))[:, None] if case.averaging == bm.AVG_HARMONIC else eta_q
            self.eta_levels = [self.eta_active]
            self.eta_min, self.eta_max = float(self.eta_active.min()), float(self.eta_active.max())
        else:
            self.eta_active, self.eta_levels, self.eta_min, self.eta_max = bm.viscosity_tables(
                self.hier, case.eta, case.averaging, gather=gather if self.hier.n_replicated else None,
                n_replicated=self.hier.n_replicated)
        self.free_surface = None
        self.dilation = None
        # right-hand side: own cells, then the halo cells' shares of the rank's nodes
        Nq = bm.tensor_shape([bm.N2] * dim)
        W = bm.tensor_shape([bm.GAUSS3_W.reshape(3, 1)] * dim).reshape(-1) * np.prod(fin.h)
        out = np.zeros((fin.n_nodes_v, dim))

        def add(cell_ijk, nodes):
            if case.force_q is not None:
                raise ValueError("rank-local set-up: FE-field densities need the global mesh")
            xq = fin.qpoint_coords(cell_ijk)
            f = case.force(xq.reshape(-1, dim)).reshape(cell_ijk.shape[0], -1, dim)
            loc = np.matmul((Nq * W[:, None]).T[None, :, :], f)          # [cell][i][d] = sum_q N_i(q) W_q f(q)
            keep = nodes >= 0
            np.add.at(out, nodes[keep], loc[keep])

        add(fin.cell_ijk, fin.cell_nodes_v.astype(np.int64))
        halo = fin.halo_cells()
        if halo.shape[0]:
            add(halo, fin.local_nodes_of(halo))
        out = out.reshape(-1)
        out[fin.constrained_dofs_v()] = 0.0
        self.rhs_u = out
        self.rhs_p = np.zeros(fin.n_dofs_p)
        if case.initial_pressure is not None:
            self.p0 = case.initial_pressure(fin.node_coords_p()) / self.pressure_scaling
        else:
            self.p0 = np.zeros(fin.n_dofs_p)
        self.n_global_dofs_v = fin.n_global_nodes_v * dim
        self.n_global_dofs_p = fin.n_global_nodes_p

    def to_local(self, g):
        """global block vector [u; p] -> local block vector (tests on small meshes)."""
        fin = self.hier.finest
        return np.concatenate([fin.to_local_v(g[:self.n_global_dofs_v]), fin.to_local_p(g[self.n_global_dofs_v:])])

    def owned_block(self):
        fin = self.hier.finest
        lu, gu = fin.owned_dofs_v()
        lp, gp = fin.owned_dofs_p()
        return np.concatenate([lu, fin.n_dofs_v + lp]), np.concatenate([gu, self.n_global_dofs_v + gp])

    def global_index_vector(self):
        """Global index of every local entry of a block vector [u; p] (synthetic inputs that are a function of the
        global index, generated without a global vector)."""
        fin = self.hier.finest
        d = self.dim
        gu = (fin.v.l2g[:, None] * d + np.arange(d)[None, :]).reshape(-1)
        return np.concatenate([gu, self.n_global_dofs_v + fin.p.l2g])
