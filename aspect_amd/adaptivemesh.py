"""Adaptively refined box mesh with hanging nodes -> flat tables for the C-ABI (first slice of SURVEY.md 8 f1).

The reference refines its p4est forest adaptively (2:1 balanced across faces, edges and corners, as deal.II asks of
p4est), numbers every node of every active cell as a DoF -- hanging ones included -- and ties the hanging nodes to
the nodes of the coarser neighbour with `DoFTools::make_hanging_node_constraints`
(stokes_matrix_free_local_smoothing.cc:1597-1605, 1675-1682); MatrixFree resolves them when it reads and writes
cell values.  This module is the stand-in host for that: the ACTIVE mesh of a box refined uniformly `base` times and
then locally, as one level for `agmg_level_desc`:

* active cells in the hierarchical (Morton) order of the forest: a refined cell is replaced by its 2^dim children;
* Q2 / Q1 nodes numbered in the order a walk over the cells first meets them;
* hanging nodes: a node of a fine cell that lies on a face or edge of a coarser active cell without being one of
  its nodes; its value is that cell's Q2 (Q1) interpolant there, i.e. a few weights from the 1-D shape functions at
  1/4, 1/2, 3/4;
* constraint masks, cell extents and quadrature-point coordinates as `boxmesh.Level` has them.

Multigrid levels of such a mesh (local smoothing: refinement-edge masks, interface operators, transfers between
partially refined levels) are NOT built: a level of this class is used on its own (`capi.Solver(..., single_level=True)`).
"""
from __future__ import annotations

import numpy as np

from . import boxmesh as bm


def _q2_1d(xi):
    return np.array([(1 - xi) * (1 - 2 * xi), 4 * xi * (1 - xi), xi * (2 * xi - 1)])


def _q1_1d(xi):
    return np.array([1 - xi, xi])


class AdaptiveBoxLevel:
    """Active mesh of a box: `base` global refinements, then `rounds` of local refinement; `marker(centers, level,
    round)` returns the cells to refine (boolean per active cell).  2:1 balance over faces, edges and corners is
    enforced by refining further cells."""

    def __init__(self, dim, base, marker, rounds=1, extents=None, reps=None, bc=None):
        self.dim = dim
        self.extents = np.asarray([1.0] * dim if extents is None else extents, dtype=np.float64)
        self.reps = np.asarray([1] * dim if reps is None else reps, dtype=np.int64)
        self.bc = [bm.BC_TANGENTIAL] * (2 * dim) if bc is None else list(bc)
        self.general = False
        self.periodic = [False] * dim
        self.level = 0            # (used on its own: level 0 of a one-level context)
        # leaves in forest order: (level, i, j, k) with ijk on the level's own cell grid
        uni = bm.Level(dim, self.extents, self.reps, base, self.bc)
        leaves = [(base,) + tuple(int(v) for v in ijk) for ijk in uni.cell_ijk]
        for r in range(rounds):
            lv = np.array([c[0] for c in leaves])
            ijk = np.array([c[1:] for c in leaves], dtype=np.float64)
            h = self.extents[None, :] / (self.reps[None, :] * (2.0 ** lv)[:, None])
            mark = np.asarray(marker((ijk + 0.5) * h, lv, r), dtype=bool)
            leaves = self._refine(leaves, mark)
        self.leaves = leaves
        self.n_cells = len(leaves)
        self.cell_level = np.array([c[0] for c in leaves], dtype=np.int64)
        self.cell_ijk = np.array([c[1:] for c in leaves], dtype=np.int64)
        self.max_level = int(self.cell_level.max())
        self.n_levels_of_mesh = self.max_level + 1
        self.cell_h = self.extents[None, :] / (self.reps[None, :] * (2.0 ** self.cell_level)[:, None])
        self._number()

    # -- refinement with 2:1 balance -------------------------------------------------------------------------
    def _refine(self, leaves, mark):
        dim = self.dim
        index = {c: n for n, c in enumerate(leaves)}
        mark = mark.copy()
        offs = [o for o in np.indices((3,) * dim).reshape(dim, -1).T - 1 if np.any(o != 0)]
        ncd0 = self.reps
        work = list(np.flatnonzero(mark))
        while work:
            n = work.pop()
            l, ijk = leaves[n][0], np.array(leaves[n][1:])
            if l == 0:
                continue
            # after refinement this cell's children are at level l + 1: every neighbour (face, edge, corner) must be
            # at level >= l, i.e. a neighbour leaf at level l - 1 has to be refined as well
            for o in offs:
                q = ijk + o
                if np.any(q < 0) or np.any(q >= ncd0 * (1 << l)):
                    continue
                key = (l - 1,) + tuple(int(v) for v in q // 2)
                m = index.get(key)
                if m is not None and not mark[m]:
                    mark[m] = True
                    work.append(m)
        out = []
        for n, c in enumerate(leaves):
            if not mark[n]:
                out.append(c)
                continue
            l, ijk = c[0], c[1:]
            for ch in range(1 << dim):
                out.append((l + 1,) + tuple(2 * ijk[d] + ((ch >> d) & 1) for d in range(dim)))
        return out

    # -- numbering and hanging nodes -------------------------------------------------------------------------
    def _number(self):
        dim, L = self.dim, self.max_level
        s = 1 << (L - self.cell_level)                    # node spacing of a cell on the finest Q2 lattice
        self.lattice_n = 2 * self.reps * (1 << L) + 1
        mul = np.concatenate([[1], np.cumprod(self.lattice_n[:-1])]).astype(np.int64)
        loc3 = np.indices((3,) * dim).reshape(dim, -1)[::-1].T          # x fastest
        loc2 = np.indices((2,) * dim).reshape(dim, -1)[::-1].T
        Xv = (2 * self.cell_ijk[:, None, :] + loc3[None, :, :]) * s[:, None, None]        # [cell][27][dim]
        Xp = (2 * self.cell_ijk[:, None, :] + 2 * loc2[None, :, :]) * s[:, None, None]    # [cell][8][dim]

        def number(X, vertex_of_cell):
            # deal.II hands out DoFs per vertex, line, quad and hex: where a refined line or face meets its unrefined
            # neighbour, the middle vertex of the children and the coarse line / quad DoF at the same point are TWO
            # DoFs (tied by a constraint with weight 1).  Same point + same role (vertex of its cell or not) = same DoF.
            key = 2 * (X.reshape(-1, dim) @ mul) + np.tile(vertex_of_cell, X.shape[0])
            uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
            order = np.argsort(first, kind="stable")                    # first-touch order of the cell walk
            new_of = np.empty(uniq.size, dtype=np.int64)
            new_of[order] = np.arange(uniq.size)
            coords = X.reshape(-1, dim)[first[order]]
            return new_of[inv].reshape(X.shape[:2]).astype(np.uint32), coords
        self.cell_nodes_v, self.lattice_v = number(Xv, np.all(loc3 != 1, axis=1).astype(np.int64))
        self.cell_nodes_p, self.lattice_p = number(Xp, np.ones(loc2.shape[0], dtype=np.int64))
        self.n_nodes_v, self.n_nodes_p = self.lattice_v.shape[0], self.lattice_p.shape[0]
        leaf_at = {c: n for n, c in enumerate(self.leaves)}
        self.hanging_v = self._hanging(self.lattice_v, self.cell_nodes_v, leaf_at, q2=True)
        self.hanging_p = self._hanging(self.lattice_p, self.cell_nodes_p, leaf_at, q2=False)
        # constraint masks (boxmesh.Level._constraint_mask on these nodes)
        mask = np.zeros(self.n_nodes_v, dtype=np.uint8)
        allbits = (1 << dim) - 1
        for d in range(dim):
            for side in range(2):
                code = self.bc[2 * d + side]
                on = self.lattice_v[:, d] == (0 if side == 0 else self.lattice_n[d] - 1)
                if code == bm.BC_ZERO:
                    mask[on] |= allbits
                elif code == bm.BC_TANGENTIAL:
                    mask[on] |= (1 << d)
                elif code >= bm.BC_COMPONENTS:
                    mask[on] |= (code - bm.BC_COMPONENTS) & allbits
        self.cmask_v = mask
        self.cmask_p = np.zeros(self.n_nodes_p, dtype=np.uint8)

    def _hanging(self, lattice, cell_nodes, leaf_at, q2):
        """(nodes, ptr, masters, weights): CSR of the hanging nodes of one space."""
        dim, L = self.dim, self.max_level
        nodes, ptr, masters, weights = [], [0], [], []
        shape = _q2_1d if q2 else _q1_1d
        n1 = 3 if q2 else 2
        loc = np.indices((n1,) * dim).reshape(dim, -1)[::-1].T
        ncd0 = self.reps
        # coarsest level of the cells that list a node
        own_level = np.full(lattice.shape[0], L + 1, dtype=np.int64)
        np.minimum.at(own_level, cell_nodes.reshape(-1).astype(np.int64), np.repeat(self.cell_level, cell_nodes.shape[1]))
        for node in range(lattice.shape[0]):
            X = lattice[node]
            found = None
            for l in range(int(own_level[node])):                    # an active cell coarser than every cell listing the node,
                size = 2 << (L - l)                                  # whose closed box holds the point: the node hangs on it
                cands = [[int(X[d] // size)] + ([int(X[d] // size) - 1] if X[d] % size == 0 else []) for d in range(dim)]
                for combo in np.array(np.meshgrid(*cands, indexing="ij")).reshape(dim, -1).T:
                    if np.any(combo < 0) or np.any(combo >= ncd0 * (1 << l)):
                        continue
                    leaf = leaf_at.get((l,) + tuple(int(v) for v in combo))
                    if leaf is not None:
                        found = (leaf, l, combo, size)
                        break
                if found:
                    break
            if not found:
                continue
            leaf, l, combo, size = found
            xi = (X - combo * size) / size
            w1 = [shape(xi[d]) for d in range(dim)]
            w = np.ones(loc.shape[0])
            for d in range(dim):
                w *= w1[d][loc[:, d]]
            keep = np.abs(w) > 1e-14
            nodes.append(node)
            masters.extend(int(v) for v in cell_nodes[leaf][keep])
            weights.extend(float(v) for v in w[keep])
            ptr.append(len(masters))
        hn = np.asarray(nodes, dtype=np.uint32)
        hm = np.asarray(masters, dtype=np.uint32)
        assert not np.isin(hm, hn).any(), "a master node is hanging itself: the mesh is not 2:1 balanced"
        return hn, np.asarray(ptr, dtype=np.int32), hm, np.asarray(weights, dtype=np.float64)

    # -- what the table builders of cases / capi use ---------------------------------------------------------
    @property
    def n_dofs_v(self):
        return self.dim * self.n_nodes_v

    @property
    def n_dofs_p(self):
        return self.n_nodes_p

    @property
    def h_fine(self):
        return self.extents / (self.reps * (1 << self.max_level))

    def node_coords_v(self):
        return self.lattice_v * (self.h_fine / 2)[None, :]

    def node_coords_p(self):
        return self.lattice_p * (self.h_fine / 2)[None, :]

    def qpoint_coords(self):
        q = np.indices((3,) * self.dim).reshape(self.dim, -1)[::-1]
        out = np.zeros((self.n_cells, 3 ** self.dim, self.dim))
        for d in range(self.dim):
            out[:, :, d] = (self.cell_ijk[:, d:d + 1] + bm.GAUSS3_X[q[d]][None, :]) * self.cell_h[:, d:d + 1]
        return out

    def constrained_dofs_v(self):
        bits = (self.cmask_v[:, None] >> np.arange(self.dim)[None, :]) & 1
        return bits.reshape(-1).astype(bool)

    def hanging_dofs_v(self):
        out = np.zeros((self.n_nodes_v, self.dim), dtype=bool)
        out[self.hanging_v[0].astype(np.int64)] = True
        return out.reshape(-1)

    def interpolate_v(self, f):
        """Nodal values of a vector field f(x[n, dim]) -> [n, dim]; hanging nodes take their constrained values."""
        u = f(self.node_coords_v())
        hn, hp, hm, hw = self.hanging_v
        for i, h in enumerate(hn):
            u[h] = (hw[hp[i]:hp[i + 1], None] * u[hm[hp[i]:hp[i + 1]].astype(np.int64)]).sum(axis=0)
        return u.reshape(-1)


class AdaptiveHierarchy:
    """One-level 'hierarchy' around an AdaptiveBoxLevel (what capi.Solver(single_level=True) walks)."""

    def __init__(self, level):
        self.dim = level.dim
        self.levels = [level]

    @property
    def finest(self):
        return self.levels[-1]

    @property
    def n_levels(self):
        return 1


class AdaptiveProblem:
    """The minimum capi.Solver needs: the level and its viscosity table (cell-constant or per quadrature point)."""

    def __init__(self, level, eta, per_q=True, pressure_scaling=1.0, is_compressible=False):
        self.hier = AdaptiveHierarchy(level)
        self.dim = level.dim
        self.name = "adaptive_box"
        self.tol, self.restart, self.n_cheap, self.n_expensive, self.krylov, self.idr_s = 1e-7, 100, 200, 0, "gmres", 2
        self.pressure_scaling = pressure_scaling
        self.is_compressible = is_compressible
        eq = eta(level.qpoint_coords().reshape(-1, level.dim)).reshape(level.n_cells, -1)
        self.eta_active = eq if per_q else (eq.shape[1] / np.sum(1.0 / eq, axis=1))[:, None]
        self.eta_levels = [self.eta_active]
        self.eta_min, self.eta_max = float(self.eta_active.min()), float(self.eta_active.max())
        self.free_surface = None
