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


class LocalLevel:
    """One rank's share of a ``boxmesh.Level``; duck-types it for ``capi.Solver``."""

    def __init__(self, glev, rank, size):
        self.dim, self.level = glev.dim, glev.level
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
        self.global_nodes_v = self.v.l2g.astype(np.uint32)
        self.global_nodes_p = self.p.l2g.astype(np.uint32)
        self.n_global_nodes_v, self.n_global_nodes_p = glev.n_nodes_v, glev.n_nodes_p

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


class LocalHierarchy:
    def __init__(self, ghier, rank, size):
        self.dim = ghier.dim
        self.rank, self.size = rank, size
        self.levels = [LocalLevel(g, rank, size) for g in ghier.levels]

    @property
    def finest(self):
        return self.levels[-1]

    @property
    def n_levels(self):
        return len(self.levels)


class LocalProblem:
    """Rank-local view of a built ``cases.StokesProblem``: local tables, coefficient tables and
    right-hand side of the rank's subdomain (what each MPI rank of the reference holds)."""

    def __init__(self, gproblem, rank, size):
        self.g = gproblem
        self.rank, self.size = rank, size
        for k in ("name", "dim", "tol", "restart", "n_cheap", "n_expensive", "krylov", "idr_s",
                  "pressure_scaling", "eta_min", "eta_max", "is_compressible"):
            setattr(self, k, getattr(gproblem, k, None))
        self.hier = LocalHierarchy(gproblem.hier, rank, size)
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
