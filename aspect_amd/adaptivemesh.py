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

`AdaptiveMGHierarchy` adds the multigrid levels of local smoothing (stokes_matrix_free_local_smoothing.cc:1760-1877):
level l holds ALL cells of refinement level l (active or refined), numbered without hanging nodes; its refinement-edge
mask marks the nodes on faces whose neighbour of the same level does not exist (MGTools::extract_inner_interface_dofs);
its copy table pairs the nodes of the cells that are active on the level with the active-mesh nodes, refinement-edge
nodes left out (MGLevelGlobalTransfer::fill_copy_indices); parents and child numbers drive the transfers.
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


    def nodal_to_qpoints(self, nodal):
        """Q2 field at the velocity nodes (hanging nodes holding their constrained values) -> [n_cells][nq]."""
        return nodal[self.cell_nodes_v.astype(np.int64)] @ bm.tensor_shape([bm.N2] * self.dim).T

    def condense(self, vec_nodes, hanging):
        """AffineConstraints::condense of a right-hand side [n_nodes][ncomp]: rows of hanging nodes go to their masters."""
        hn, hp, hm, hw = hanging
        for i, h in enumerate(hn):
            k = slice(hp[i], hp[i + 1])
            np.add.at(vec_nodes, hm[k].astype(np.int64), hw[k, None] * vec_nodes[h][None, :])
            vec_nodes[h] = 0.0
        return vec_nodes

    def assemble_rhs_u(self, force_func=None, force_q=None):
        """int rho g . phi_u over the active cells (Gauss-3), hanging rows condensed, constrained rows zeroed."""
        dim = self.dim
        f = force_q(self) if force_q is not None else force_func(self.qpoint_coords().reshape(-1, dim)).reshape(self.n_cells, -1, dim)
        Nq = bm.tensor_shape([bm.N2] * dim)
        W = bm.tensor_shape([bm.GAUSS3_W.reshape(3, 1)] * dim).reshape(-1)
        loc = np.einsum("qi,q,c,cqd->cid", Nq, W, np.prod(self.cell_h, axis=1), f)
        out = np.zeros((self.n_nodes_v, dim))
        np.add.at(out, self.cell_nodes_v.astype(np.int64), loc)
        out = self.condense(out, self.hanging_v).reshape(-1)
        out[self.constrained_dofs_v()] = 0.0
        return out

    def distribute_v(self, u):
        """AffineConstraints::distribute on a velocity vector: hanging nodes take their constrained values."""
        u = np.array(u, dtype=np.float64).reshape(self.n_nodes_v, self.dim)
        hn, hp, hm, hw = self.hanging_v
        for i, h in enumerate(hn):
            u[h] = (hw[hp[i]:hp[i + 1], None] * u[hm[hp[i]:hp[i + 1]].astype(np.int64)]).sum(axis=0)
        return u.reshape(-1)


class AdaptiveMGLevel:
    """All cells of refinement level `l` of an adaptively refined box (one level of local smoothing)."""

    def __init__(self, active, l, cells, parent_cell, child_index):
        dim = self.dim = active.dim
        self.level = l
        self.general = False
        self.periodic = [False] * dim
        self.cell_ijk = np.asarray(cells, dtype=np.int64).reshape(-1, dim)
        self.n_cells = self.cell_ijk.shape[0]
        self.parent_cell = None if parent_cell is None else np.asarray(parent_cell, dtype=np.int32)
        self.child_index = None if child_index is None else np.asarray(child_index, dtype=np.uint8)
        ncd = active.reps * (1 << l)
        self.h = active.extents / ncd
        self.cell_h = np.repeat(self.h[None, :], self.n_cells, axis=0)
        lat_n = 2 * ncd + 1
        mul = np.concatenate([[1], np.cumprod(lat_n[:-1])]).astype(np.int64)
        loc3 = np.indices((3,) * dim).reshape(dim, -1)[::-1].T
        loc2 = np.indices((2,) * dim).reshape(dim, -1)[::-1].T

        def number(X):
            key = X.reshape(-1, dim) @ mul
            uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
            order = np.argsort(first, kind="stable")
            new_of = np.empty(uniq.size, dtype=np.int64)
            new_of[order] = np.arange(uniq.size)
            return new_of[inv].reshape(X.shape[:2]).astype(np.uint32), X.reshape(-1, dim)[first[order]]
        self.cell_nodes_v, self.lattice_v = number(2 * self.cell_ijk[:, None, :] + loc3[None, :, :])
        self.cell_nodes_p, self.lattice_p = number(2 * self.cell_ijk[:, None, :] + 2 * loc2[None, :, :])
        self.n_nodes_v, self.n_nodes_p = self.lattice_v.shape[0], self.lattice_p.shape[0]
        mask = np.zeros(self.n_nodes_v, dtype=np.uint8)
        allbits = (1 << dim) - 1
        for d in range(dim):
            for side in range(2):
                code = active.bc[2 * d + side]
                on = self.lattice_v[:, d] == (0 if side == 0 else lat_n[d] - 1)
                if code == bm.BC_ZERO:
                    mask[on] |= allbits
                elif code == bm.BC_TANGENTIAL:
                    mask[on] |= (1 << d)
                elif code >= bm.BC_COMPONENTS:
                    mask[on] |= (code - bm.BC_COMPONENTS) & allbits
        self.cmask_v = mask
        self.cmask_p = np.zeros(self.n_nodes_p, dtype=np.uint8)
        # refinement edge: faces of a cell whose neighbour on this level does not exist (and is not the domain boundary)
        here = {tuple(int(v) for v in c) for c in self.cell_ijk}
        self.edge_v = np.zeros(self.n_nodes_v, dtype=np.uint8)
        self.edge_p = np.zeros(self.n_nodes_p, dtype=np.uint8)
        for n, c in enumerate(self.cell_ijk):
            for d in range(dim):
                for side in range(2):
                    q = c.copy()
                    q[d] += 1 if side else -1
                    if q[d] < 0 or q[d] >= ncd[d] or tuple(int(v) for v in q) in here:
                        continue
                    self.edge_v[self.cell_nodes_v[n][loc3[:, d] == 2 * side]] = 1
                    self.edge_p[self.cell_nodes_p[n][loc2[:, d] == side]] = 1
        self.index_of = {tuple(int(v) for v in c): n for n, c in enumerate(self.cell_ijk)}

    @property
    def n_dofs_v(self):
        return self.dim * self.n_nodes_v

    @property
    def n_dofs_p(self):
        return self.n_nodes_p

    def qpoint_coords(self):
        q = np.indices((3,) * self.dim).reshape(self.dim, -1)[::-1]
        out = np.zeros((self.n_cells, 3 ** self.dim, self.dim))
        for d in range(self.dim):
            out[:, :, d] = (self.cell_ijk[:, d:d + 1] + bm.GAUSS3_X[q[d]][None, :]) * self.h[d]
        return out

    def constrained_dofs_v(self):
        bits = (self.cmask_v[:, None] >> np.arange(self.dim)[None, :]) & 1
        return bits.reshape(-1).astype(bool)


class AdaptiveMGHierarchy:
    """Levels 0..max_level of local smoothing under an AdaptiveBoxLevel (`active`); `finest` is the ACTIVE mesh (the
    vectors of the solver live there), `levels` the multigrid levels."""

    def __init__(self, active):
        self.dim = dim = active.dim
        self.active = active
        leaves = {c: n for n, c in enumerate(active.leaves)}
        coarse = bm.Level(dim, active.extents, active.reps, 0, active.bc)
        cells = [tuple(int(v) for v in c) for c in coarse.cell_ijk]
        parent, child = None, None
        self.levels = []
        for l in range(active.max_level + 1):
            lev = AdaptiveMGLevel(active, l, cells, parent, child)
            lev.active_cell = np.array([leaves.get((l,) + c, -1) for c in cells], dtype=np.int64)
            self.levels.append(lev)
            nxt, parent, child = [], [], []
            for n, c in enumerate(cells):
                if lev.active_cell[n] >= 0:
                    continue
                for ch in range(1 << dim):
                    nxt.append(tuple(2 * c[d] + ((ch >> d) & 1) for d in range(dim)))
                    parent.append(n)
                    child.append(ch)
            cells = nxt
        assert not cells and sum(int((lv.active_cell >= 0).sum()) for lv in self.levels) == active.n_cells
        for lev in self.levels:                      # copy tables: (active node, level node) of the cells active on the level
            act = np.flatnonzero(lev.active_cell >= 0)
            for name, cn_a, cn_l, edge in (("v", active.cell_nodes_v, lev.cell_nodes_v, lev.edge_v),
                                           ("p", active.cell_nodes_p, lev.cell_nodes_p, lev.edge_p)):
                a = cn_a[lev.active_cell[act]].reshape(-1).astype(np.int64)
                b = cn_l[act].reshape(-1).astype(np.int64)
                keep = edge[b] == 0
                pairs = np.unique(np.stack([a[keep], b[keep]], axis=1), axis=0) if keep.any() else np.zeros((0, 2), dtype=np.int64)
                assert np.unique(pairs[:, 1]).size == pairs.shape[0] and np.unique(pairs[:, 0]).size == pairs.shape[0]
                setattr(lev, "copy_active_" + name, pairs[:, 0].astype(np.uint32))
                setattr(lev, "copy_level_" + name, pairs[:, 1].astype(np.uint32))

    @property
    def finest(self):
        return self.active

    @property
    def n_levels(self):
        return len(self.levels)

    def level_tables(self, active_table, restrict):
        """Coefficient tables of the levels from the table of the active cells: a cell that is active on its level keeps
        its row, a refined cell takes `restrict(rows of its 2^dim children [parents][children][k])` (the DG transfer of
        the reference, MGTransferMF::interpolate_to_mg, local_smoothing.cc:388-445)."""
        out = [None] * self.n_levels
        for l in range(self.n_levels - 1, -1, -1):
            lev = self.levels[l]
            t = np.zeros((lev.n_cells, active_table.shape[1]))
            act = lev.active_cell >= 0
            t[act] = active_table[lev.active_cell[act]]
            if l + 1 < self.n_levels and self.levels[l + 1].n_cells:
                nxt = self.levels[l + 1]
                kids = out[l + 1].reshape(-1, 1 << self.dim, active_table.shape[1])
                t[nxt.parent_cell[::1 << self.dim]] = restrict(kids)
            out[l] = t
        return out


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

    def __init__(self, level, eta, per_q=True, pressure_scaling=1.0, is_compressible=False, multigrid=False,
                 force=None, force_q=None, initial_pressure=None, tol=1e-7, n_cheap=200, n_expensive=0):
        """multigrid: also the levels of local smoothing (AdaptiveMGHierarchy; level tables: cell-constant
        viscosities, child means on refined cells) and the right-hand side of `force` / `force_q`."""
        self.hier = AdaptiveMGHierarchy(level) if multigrid else AdaptiveHierarchy(level)
        self.dim = level.dim
        self.name = "adaptive_box"
        self.tol, self.restart, self.n_cheap, self.n_expensive, self.krylov, self.idr_s = tol, 100, n_cheap, n_expensive, "gmres", 2
        self.pressure_scaling = pressure_scaling
        self.is_compressible = is_compressible
        eq = eta(level.qpoint_coords().reshape(-1, level.dim)).reshape(level.n_cells, -1)
        self.eta_active = eq if per_q else (eq.shape[1] / np.sum(1.0 / eq, axis=1))[:, None]
        self.eta_min, self.eta_max = float(self.eta_active.min()), float(self.eta_active.max())
        self.free_surface = None
        if not multigrid:
            self.eta_levels = [self.eta_active]
            return
        if per_q:       # (levels carry cell-constant tables: the mean over the q-points, then child means)
            cellv = (eq @ bm.tensor_shape([bm.GAUSS3_W.reshape(3, 1)] * level.dim).reshape(-1))[:, None]
        else:
            cellv = self.eta_active
        self.eta_levels = self.hier.level_tables(cellv, lambda kids: kids.mean(axis=1))
        self.rhs_u = level.assemble_rhs_u(force, force_q) if (force is not None or force_q is not None) else np.zeros(level.n_dofs_v)
        self.rhs_p = np.zeros(level.n_dofs_p)
        self.p0 = np.zeros(level.n_dofs_p) if initial_pressure is None else initial_pressure(level.node_coords_p()) / pressure_scaling
        self.p0[level.hanging_p[0].astype(np.int64)] = 0.0       # (constrained entries of the start vector are zero)
        self.u0 = None
        self.dilation = None
