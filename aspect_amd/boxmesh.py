"""Structured box mesh hierarchy -> flat tables for the C-ABI.

The reference keeps deal.II/p4est for mesh, DoF and constraint handling
(stokes_matrix_free_local_smoothing.cc:1548-1878, ``setup_dofs``) and hands
MatrixFree the cell->DoF tables, constraints and mapping data.  Neither
deal.II nor p4est exists in this image, so this module is the stand-in host:
it produces exactly the flat tables the glue of INTEGRATION.md would flatten
out of ``MatrixFree`` for a box with a multigrid hierarchy
of uniformly refined levels:

* cells in hierarchical (Morton) order inside each coarse cell, so the
  children of cell ``P`` on level ``l-1`` are cells ``2^dim*P + c`` on level
  ``l`` (p4est/deal.II child order: bit0 -> x, bit1 -> y, bit2 -> z);
* Q2 velocity nodes / Q1 pressure nodes numbered either lexicographically on
  the level grid or (``numbering="cellwise"``) in the order in which a walk
  over the cells first meets them -- the locality deal.II's own numbering has
  (DoFs are distributed cell by cell along the space-filling curve), which
  keeps the nodes of a run of consecutive cells close together in memory;
  velocity DoF = ``dim*node + component``;
* per-node constraint bit masks (bit c: component c is a homogeneous
  Dirichlet row: zero velocity -> all bits, axis-aligned no-normal-flux ->
  the normal component only; local_smoothing.cc:1597-1620, 1675-1698,
  1794-1821);
* Cartesian geometry (MappingCartesian, core.cc:118-119): cell extents; or,
  with a ``deform`` map, general geometry: inverse Jacobians and JxW at the
  Gauss points of the isoparametric Q2 mapping through the displaced nodes
  (the tables MatrixFree's MappingInfo holds for a MappingQ mesh,
  core.cc:120-121).

Element conventions follow SURVEY.md A.1: lexicographic cell-local DoF order,
Gauss-3 quadrature per direction for every operator.
"""
from __future__ import annotations

import numpy as np

# ---------------------------------------------------------------------------
# Reference element
# ---------------------------------------------------------------------------
GAUSS3_X = np.array([0.5 - 0.5 * np.sqrt(0.6), 0.5, 0.5 + 0.5 * np.sqrt(0.6)])
GAUSS3_W = np.array([5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0])


def q2_shape(x):
    x = np.asarray(x, dtype=np.float64)
    return np.stack([(1 - x) * (1 - 2 * x), 4 * x * (1 - x), x * (2 * x - 1)], axis=-1)


def q2_dshape(x):
    x = np.asarray(x, dtype=np.float64)
    return np.stack([4 * x - 3, 4 - 8 * x, 4 * x - 1], axis=-1)


def q1_shape(x):
    x = np.asarray(x, dtype=np.float64)
    return np.stack([1 - x, x], axis=-1)


N2 = q2_shape(GAUSS3_X)    # [q][i]
D2 = q2_dshape(GAUSS3_X)   # [q][i]
N1 = q1_shape(GAUSS3_X)    # [q][i]


def tensor_shape(mats):
    """Tensor-product shape table [q_lex][i_lex] with x fastest."""
    out = mats[0]
    for m in mats[1:]:
        out = np.kron(m, out)
    return out


def _morton_decode(m, dim, nbits):
    coords = [np.zeros_like(m) for _ in range(dim)]
    for k in range(nbits):
        for d in range(dim):
            coords[d] |= ((m >> (dim * k + d)) & 1) << k
    return coords


BC_NONE, BC_ZERO, BC_TANGENTIAL = 0, 1, 2
BC_COMPONENTS = 16      # BC_COMPONENTS + bits: the velocity components in `bits` are constrained (`bottom x: function`)


class Level:
    """Flat tables of one mesh level (what ``agmg_level_desc`` carries)."""

    def __init__(self, dim, extents, reps, l, bc, deform=None, periodic=None, numbering="lexicographic"):
        self.dim = dim
        self.numbering = numbering
        self.lex_of_v = self.lex_of_p = None   # new node index -> lexicographic index (cellwise only)
        self.deform = deform
        # periodic directions (geometry_model/box.cc "X periodic"): the nodes of the upper face are
        # the nodes of the lower face, i.e. periodicity is plain cell -> node connectivity here
        # (deal.II keeps both DoFs and constrains one to the other: the same reduced system)
        self.periodic = [False] * dim if periodic is None else [bool(b) for b in periodic]
        self.level = l
        self.extents = np.asarray(extents, dtype=np.float64)
        self.reps = np.asarray(reps, dtype=np.int64)
        self.ncd = self.reps * (1 << l)              # cells per direction
        self.n_cells = int(np.prod(self.ncd))
        self.h = self.extents / self.ncd
        per_coarse = 1 << (dim * l)
        ids = np.arange(self.n_cells, dtype=np.int64)
        coarse = ids // per_coarse
        local = _morton_decode(ids % per_coarse, dim, l)
        cc = []
        rem = coarse
        for d in range(dim):
            cc.append(rem % self.reps[d])
            rem = rem // self.reps[d]
        self.cell_ijk = np.stack([cc[d] * (1 << l) + local[d] for d in range(dim)], axis=1)

        # velocity (Q2) nodes
        wrap = np.array([1 if b else 0 for b in self.periodic], dtype=np.int64)
        self.nvd = 2 * self.ncd + 1 - wrap
        self.n_nodes_v = int(np.prod(self.nvd))
        self.npd = self.ncd + 1 - wrap
        self.n_nodes_p = int(np.prod(self.npd))
        self.cell_nodes_v = self._cell_nodes(3, 2, self.nvd)
        self.cell_nodes_p = self._cell_nodes(2, 1, self.npd)
        if numbering == "cellwise":
            self.cell_nodes_v, self.lex_of_v = self._first_touch(self.cell_nodes_v, self.n_nodes_v)
            self.cell_nodes_p, self.lex_of_p = self._first_touch(self.cell_nodes_p, self.n_nodes_p)
        elif numbering != "lexicographic":
            raise ValueError(f"unknown numbering {numbering!r}")
        self.cmask_v = self._constraint_mask(bc)
        self.cmask_p = np.zeros(self.n_nodes_p, dtype=np.uint8)
        self.cell_h = np.tile(self.h, (self.n_cells, 1))
        self.general = deform is not None
        if self.general:
            self._general_geometry()

    def _general_geometry(self):
        """inv_jacobian [cell][q][e][d] = d xi_e / d x_d and jxw [cell][q] of the Q2 mapping."""
        dim = self.dim
        X = self.node_coords_v()[self.cell_nodes_v.astype(np.int64)]     # [cell][i][d]
        self.inv_jacobian = np.zeros((self.n_cells, 3 ** dim, dim, dim))
        J = np.zeros((self.n_cells, 3 ** dim, dim, dim))                 # [cell][q][d][e] = d x_d / d xi_e
        for e in range(dim):
            dN = tensor_shape([D2 if k == e else N2 for k in range(dim)])   # [q][i]
            J[:, :, :, e] = np.einsum("qi,cid->cqd", dN, X)
        det = np.linalg.det(J)
        if not np.all(det > 0):
            raise ValueError("deformation inverts cells")
        self.inv_jacobian = np.linalg.inv(J)
        W = tensor_shape([GAUSS3_W.reshape(3, 1)] * dim).reshape(-1)
        self.jxw = det * W[None, :]

    # -- tables -----------------------------------------------------------
    @staticmethod
    def _first_touch(cell_nodes, n_nodes):
        """Renumber nodes in the order the cell walk first meets them (cell-local order inside a cell)."""
        flat = cell_nodes.reshape(-1).astype(np.int64)
        uniq, first = np.unique(flat, return_index=True)
        assert uniq.size == n_nodes
        lex_of = uniq[np.argsort(first, kind="stable")]
        new_of = np.empty(n_nodes, dtype=np.int64)
        new_of[lex_of] = np.arange(n_nodes)
        return new_of[flat].reshape(cell_nodes.shape).astype(np.uint32), lex_of

    def _cell_nodes(self, n1, stride, nd):
        dim = self.dim
        loc = np.indices((n1,) * dim).reshape(dim, -1)[::-1]  # x fastest
        strides = np.ones(dim, dtype=np.int64)
        for d in range(1, dim):
            strides[d] = strides[d - 1] * nd[d - 1]
        out = np.zeros((self.n_cells, n1 ** dim), dtype=np.int64)
        for d in range(dim):
            out += ((stride * self.cell_ijk[:, d:d + 1] + loc[d][None, :]) % nd[d]) * strides[d]
        return out.astype(np.uint32)

    def node_coords_v(self):
        idx = np.indices(tuple(self.nvd[::-1])).reshape(self.dim, -1)[::-1]
        x = np.stack([idx[d] * (self.h[d] / 2) for d in range(self.dim)], axis=1)
        if self.lex_of_v is not None:
            x = x[self.lex_of_v]
        return self._deformed(x)

    def _deformed(self, x):
        """deform(x) for a smooth map of the box, deform(x, level) for maps that depend on the level (the Q1 Eulerian
        mappings of a deformed mesh: per-level vertex displacements, mesh_deformation/interface.cc:1548-1570)."""
        if self.deform is None:
            return x
        return self.deform(x, self) if getattr(self.deform, "per_level", False) else self.deform(x)

    def node_coords_p(self):
        idx = np.indices(tuple(self.npd[::-1])).reshape(self.dim, -1)[::-1]
        x = np.stack([idx[d] * self.h[d] for d in range(self.dim)], axis=1)
        if self.lex_of_p is not None:
            x = x[self.lex_of_p]
        return self._deformed(x)

    def _constraint_mask(self, bc):
        """bc: list of 2*dim codes in deal.II box boundary-id order
        (0: x-, 1: x+, 2: y-, 3: y+, 4: z-, 5: z+; geometry_model/box.cc)."""
        dim = self.dim
        idx = np.indices(tuple(self.nvd[::-1])).reshape(dim, -1)[::-1]
        mask = np.zeros(self.n_nodes_v, dtype=np.uint8)
        allbits = (1 << dim) - 1
        for d in range(dim):
            if self.periodic[d]:
                continue
            for side in range(2):
                code = bc[2 * d + side]
                on = idx[d] == (0 if side == 0 else self.nvd[d] - 1)
                if code == BC_ZERO:
                    mask[on] |= allbits
                elif code == BC_TANGENTIAL:
                    mask[on] |= (1 << d)
                elif code >= BC_COMPONENTS:
                    mask[on] |= (code - BC_COMPONENTS) & allbits
        return mask if self.lex_of_v is None else mask[self.lex_of_v]

    def qpoint_coords(self):
        """Physical coordinates of the 3^dim Gauss points, [n_cells][nq][dim]."""
        dim = self.dim
        if self.general:
            X = self.node_coords_v()[self.cell_nodes_v.astype(np.int64)]
            return np.einsum("qi,cid->cqd", tensor_shape([N2] * dim), X)
        q = np.indices((3,) * dim).reshape(dim, -1)[::-1]
        out = np.zeros((self.n_cells, 3 ** dim, dim))
        for d in range(dim):
            out[:, :, d] = (self.cell_ijk[:, d:d + 1] + GAUSS3_X[q[d]][None, :]) * self.h[d]
        return out

    def boundary_faces(self, boundary_id):
        """Cells with a face on box boundary ``boundary_id`` (deal.II order 2*d + side) -> (cells, face numbers)."""
        d, side = boundary_id // 2, boundary_id % 2
        if self.periodic[d]:
            return np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint8)
        on = self.cell_ijk[:, d] == (self.ncd[d] - 1 if side else 0)
        cells = np.nonzero(on)[0].astype(np.uint32)
        return cells, np.full(cells.size, boundary_id, dtype=np.uint8)

    def face_local_nodes(self, face_number):
        """Cell-local indices of the 3^(dim-1) Q2 nodes of a face, lexicographic over the tangential directions."""
        d, side = face_number // 2, face_number % 2
        loc = np.indices((3,) * self.dim).reshape(self.dim, -1)[::-1]     # x fastest
        return np.nonzero(loc[d] == (2 if side else 0))[0]

    def face_geometry(self, cells, face_numbers):
        """Face q-point coordinates, outward unit normals and JxW of the listed faces (Gauss-3 per tangential
        direction, lexicographic): what FEFaceEvaluation / FEFaceValues hold for them.
        -> x [faces][nqf][dim], normal [faces][nqf][dim], jxw [faces][nqf]"""
        dim = self.dim
        nqf = 3 ** (dim - 1)
        nf = len(cells)
        x = np.zeros((nf, nqf, dim)); normal = np.zeros((nf, nqf, dim)); jxw = np.zeros((nf, nqf))
        X = self.node_coords_v()
        qt = np.indices((3,) * (dim - 1)).reshape(dim - 1, -1)[::-1]      # tangential q indices, lower direction fastest
        W = np.prod(GAUSS3_W[qt], axis=0)
        for fno in np.unique(face_numbers):
            sel = np.nonzero(face_numbers == fno)[0]
            d, side = int(fno) // 2, int(fno) % 2
            tang = [k for k in range(dim) if k != d]
            # reference coordinates of the face q-points inside the cell
            xi = np.zeros((nqf, dim))
            xi[:, d] = float(side)
            for t, k in enumerate(tang):
                xi[:, k] = GAUSS3_X[qt[t]]
            shp = [q2_shape(xi[:, k]) for k in range(dim)]              # [q][i] per direction
            dshp = [q2_dshape(xi[:, k]) for k in range(dim)]
            loc = np.indices((3,) * dim).reshape(dim, -1)[::-1]
            N = np.ones((nqf, 3 ** dim))
            for k in range(dim):
                N *= shp[k][:, loc[k]]
            Xc = X[self.cell_nodes_v[cells[sel]].astype(np.int64)]       # [face][i][dim]
            x[sel] = np.einsum("qi,fid->fqd", N, Xc)
            J = np.zeros((sel.size, nqf, dim, dim))                      # d x_d / d xi_e
            for e in range(dim):
                dN = np.ones((nqf, 3 ** dim))
                for k in range(dim):
                    dN *= (dshp[k] if k == e else shp[k])[:, loc[k]]
                J[:, :, :, e] = np.einsum("qi,fid->fqd", dN, Xc)
            # n dS = det(J) J^-T n_ref dxi_tangential
            det = np.linalg.det(J)
            nref = np.zeros(dim); nref[d] = 1.0 if side else -1.0
            v = det[..., None] * np.einsum("fqed,e->fqd", np.linalg.inv(J), nref)
            area = np.linalg.norm(v, axis=-1)
            normal[sel] = v / area[..., None]
            jxw[sel] = area * W[None, :]
        return x, normal, jxw

    def nodal_to_qpoints(self, nodal):
        """Q2 field given at the velocity nodes -> values at the Gauss points, [n_cells][nq]."""
        return nodal[self.cell_nodes_v.astype(np.int64)] @ tensor_shape([N2] * self.dim).T

    def cell_centers(self):
        return (self.cell_ijk + 0.5) * self.h[None, :]

    @property
    def n_dofs_v(self):
        return self.dim * self.n_nodes_v

    @property
    def n_dofs_p(self):
        return self.n_nodes_p

    def constrained_dofs_v(self):
        """Boolean [n_dofs_v] mask of constrained velocity DoFs."""
        bits = (self.cmask_v[:, None] >> np.arange(self.dim)[None, :]) & 1
        return bits.reshape(-1).astype(bool)


class BoxHierarchy:
    """Levels 0..refinement of a box; level ``refinement`` is also the active mesh."""

    def __init__(self, dim, refinement, extents=None, reps=None, bc=None, only_finest=False, deform=None,
                 periodic=None, numbering="lexicographic"):
        self.dim = dim
        extents = [1.0] * dim if extents is None else extents
        reps = [1] * dim if reps is None else reps
        bc = [BC_TANGENTIAL] * (2 * dim) if bc is None else bc
        self.bc = list(bc)
        first = refinement if only_finest else 0
        self.levels = [Level(dim, extents, reps, l, bc, deform, periodic, numbering) for l in range(first, refinement + 1)]

    @property
    def finest(self):
        return self.levels[-1]

    @property
    def n_levels(self):
        return len(self.levels)


# ---------------------------------------------------------------------------
# Coefficient tables (local_smoothing.cc:172-445, utilities.cc:2697-2770,
# material_model/interface.cc:542-561)
# ---------------------------------------------------------------------------
AVG_NONE = "none"
AVG_HARMONIC = "harmonic average only viscosity"
AVG_PROJECT_Q1 = "project to Q1 only viscosity"
AVG_GEOMETRIC = "geometric average only viscosity"


def _dgq1_tables(dim):
    """Cellwise L2 projection q-values -> DGQ1 nodal values and back."""
    Nq = tensor_shape([N1] * dim)                      # [q][i]
    W = tensor_shape([GAUSS3_W.reshape(3, 1)] * dim).reshape(-1)
    M = Nq.T @ (W[:, None] * Nq)
    proj = np.linalg.solve(M, Nq.T * W[None, :])       # [i][q]
    return Nq, proj


def _dgq1_restriction(dim):
    """FE_DGQ(1) child->parent restriction = L2 projection of the piecewise
    child function onto the parent (deal.II FETools::compute_projection_matrices).
    Returns R[child][i_parent][j_child]."""
    xg, wg = np.polynomial.legendre.leggauss(4)
    xg = 0.5 * (xg + 1)
    wg = 0.5 * wg
    M1 = np.array([[1 / 3, 1 / 6], [1 / 6, 1 / 3]])
    R1 = []
    for c in range(2):
        xp = 0.5 * (xg + c)        # parent coordinate of child quadrature point
        rhs = (q1_shape(xp) * (0.5 * wg)[:, None]).T @ q1_shape(xg)   # [i_parent][j_child]
        R1.append(np.linalg.solve(M1, rhs))
    out = []
    for c in range(1 << dim):
        mats = [R1[(c >> d) & 1] for d in range(dim)]
        out.append(tensor_shape(mats))
    return np.stack(out)


def viscosity_tables(hier, eta_func, averaging, eta_q=None, gather=None, n_replicated=0):
    """Return (active_table, [level tables], eta_min, eta_max).  eta_q: the viscosity at the q-points of the finest
    level [n_cells][3^dim] in place of eta_func (solution-dependent rheologies).

    Rank-local hierarchies whose levels 0..n_replicated-1 are held whole by every rank (partition.RankLocalHierarchy):
    `gather(array)` returns the arrays of all ranks concatenated in rank order (= global cell order, a rank's cells
    being a contiguous run of it); it is called once, on the cell values of the coarsest distributed level, before
    they are coarsened onto the replicated levels, and once on the extreme values.

    Tables are ``[n_cells][1]`` (DGQ0) or ``[n_cells][3^dim]`` (per q-point)
    exactly as OperatorCellData::viscosity's second extent
    (matrix_free_operators.h:126-131).
    """
    dim = hier.dim
    fin = hier.finest
    if eta_q is None:
        eta_q = eta_func(fin.qpoint_coords().reshape(-1, dim)).reshape(fin.n_cells, -1)
    eta_min, eta_max = float(eta_q.min()), float(eta_q.max())
    nL = hier.n_levels
    ch = 1 << dim
    if averaging in (AVG_HARMONIC, AVG_GEOMETRIC):
        # unweighted harmonic mean N / sum(1/eta_q), or geometric mean exp(sum(log eta_q) / N) (material_model/
        # interface.cc, average_property); min/max are taken AFTER averaging (local_smoothing.cc:247-275)
        if averaging == AVG_HARMONIC:
            cellv = eta_q.shape[1] / np.sum(1.0 / eta_q, axis=1)
        else:
            cellv = np.exp(np.log(eta_q).mean(axis=1))
        eta_min, eta_max = float(cellv.min()), float(cellv.max())
        active = cellv[:, None].copy()
        levels = [None] * nL
        cur = cellv
        levels[nL - 1] = cur[:, None].copy()
        for l in range(nL - 2, -1, -1):
            if gather is not None and l == n_replicated - 1:
                cur = gather(cur)
            # FE_DGQ(0) restriction: arithmetic mean of the children
            cur = cur.reshape(-1, ch).mean(axis=1)
            levels[l] = cur[:, None].copy()
        return active, levels, eta_min, eta_max
    if averaging == AVG_NONE:
        # fe_projection is DGQ0 (local_smoothing.cc:101-113): the active table
        # keeps the raw q-point values (:255-266), the level tables come from
        # the cellwise L2 projection onto a constant = JxW-weighted mean
        # (utilities.cc:2697-2770), then child means.
        W = tensor_shape([GAUSS3_W.reshape(3, 1)] * dim).reshape(-1)
        cur = eta_q @ W
        if fin.general:
            cur = (eta_q * fin.jxw).sum(axis=1) / fin.jxw.sum(axis=1)
        levels = [None] * nL
        levels[nL - 1] = cur[:, None].copy()
        for l in range(nL - 2, -1, -1):
            if gather is not None and l == n_replicated - 1:
                cur = gather(cur)
            cur = cur.reshape(-1, ch).mean(axis=1)
            levels[l] = cur[:, None].copy()
        return eta_q.copy(), levels, eta_min, eta_max
    if averaging == AVG_PROJECT_Q1:
        Nq, proj = _dgq1_tables(dim)
        if getattr(fin, "general", False):
            # the cellwise L2 projection carries the cell's own JxW (compute_projection_matrix, material_model/interface.cc)
            M = np.einsum("qi,cq,qj->cij", Nq, fin.jxw, Nq)
            rhs = np.einsum("qi,cq,cq->ci", Nq, fin.jxw, eta_q)
            nodal = np.linalg.solve(M, rhs[:, :, None])[:, :, 0]
        else:
            nodal = eta_q @ proj.T                     # [cell][2^dim]
        R = _dgq1_restriction(dim)
        levels = [None] * nL
        cur = nodal
        if gather is not None and n_replicated > 0:
            # (the level tables are clamped to the extreme values of the whole mesh, local_smoothing.cc:432-437)
            ext = gather(np.array([eta_min, -eta_max]))
            eta_min, eta_max = float(ext[0::2].min()), float(-ext[1::2].min())
        for l in range(nL - 1, -1, -1):
            if l < nL - 1:
                if gather is not None and l == n_replicated - 1:
                    cur = gather(cur)
                child = cur.reshape(-1, ch, ch)        # [parent][child][j]
                cur = np.einsum("cij,pcj->pi", R, child)
            levels[l] = np.clip(cur @ Nq.T, eta_min, eta_max)
        return levels[nL - 1].copy(), levels, eta_min, eta_max
    raise ValueError(averaging)


# ---------------------------------------------------------------------------
# Right-hand side  f_u = int rho g . phi_u  (assemblers/stokes.cc:433-442)
# ---------------------------------------------------------------------------
def assemble_rhs_p(level, func):
    """int func(x) phi_p dx at the pressure nodes (Gauss-3 per direction)."""
    dim = level.dim
    f = func(level.qpoint_coords().reshape(-1, dim)).reshape(level.n_cells, -1)
    Nq = tensor_shape([N1] * dim)                      # [q][i]
    if level.general:
        loc = np.einsum("qi,cq,cq->ci", Nq, level.jxw, f)
    else:
        W = tensor_shape([GAUSS3_W.reshape(3, 1)] * dim).reshape(-1) * np.prod(level.h)
        loc = np.einsum("qi,q,cq->ci", Nq, W, f)
    out = np.zeros(level.n_nodes_p)
    np.add.at(out, level.cell_nodes_p.astype(np.int64), loc)
    return out


def assemble_boundary_traction(level, boundary_id, traction):
    """int_Gamma phi_i . t dS over the faces of one box boundary (Gauss-3 per tangential direction), as a
    velocity vector [node][component]; constrained rows are zeroed like the rest of the right-hand side."""
    dim = level.dim
    cells, fnos = level.boundary_faces(boundary_id)
    rhs = np.zeros((level.n_nodes_v, dim))
    if len(cells) == 0:
        return rhs.reshape(-1)
    x, normal, jxw = level.face_geometry(cells, fnos)
    nqf = 3 ** (dim - 1)
    qt = np.indices((3,) * (dim - 1)).reshape(dim - 1, -1)[::-1]
    loc = np.indices((3,) * dim).reshape(dim, -1)[::-1]
    t = traction(x.reshape(-1, dim)).reshape(x.shape)
    for fno in np.unique(fnos):
        sel = np.nonzero(fnos == fno)[0]
        d, side = int(fno) // 2, int(fno) % 2
        tang = [k for k in range(dim) if k != d]
        xi = np.zeros((nqf, dim))
        xi[:, d] = float(side)
        for m, k in enumerate(tang):
            xi[:, k] = GAUSS3_X[qt[m]]
        N = np.ones((nqf, 3 ** dim))
        for k in range(dim):
            N *= q2_shape(xi[:, k])[:, loc[k]]
        contrib = np.einsum("qi,fq,fqd->fid", N, jxw[sel], t[sel])          # [face][local node][component]
        np.add.at(rhs, level.cell_nodes_v[cells[sel]].astype(np.int64), contrib)
    bits = ((level.cmask_v[:, None] >> np.arange(dim)[None, :]) & 1).astype(bool)
    rhs[bits] = 0.0
    return rhs.reshape(-1)


def assemble_rhs_u(level, force_func, force_q=None):
    """force_func(x[n,dim]) -> rho*g [n,dim], or force_q(level) -> rho*g at the q-points
    [n_cells,nq,dim]; constrained rows are zeroed."""
    dim = level.dim
    if force_q is not None:
        f = force_q(level)
    else:
        xq = level.qpoint_coords()
        f = force_func(xq.reshape(-1, dim)).reshape(level.n_cells, -1, dim)
    Nq = tensor_shape([N2] * dim)                      # [q][i]
    if level.general:
        loc = np.einsum("qi,cq,cqd->cid", Nq, level.jxw, f)
    else:
        W = tensor_shape([GAUSS3_W.reshape(3, 1)] * dim).reshape(-1) * np.prod(level.h)
        loc = np.einsum("qi,q,cqd->cid", Nq, W, f)     # [cell][i][d]
    out = np.zeros((level.n_nodes_v, dim))
    np.add.at(out, level.cell_nodes_v.astype(np.int64), loc)
    out = out.reshape(-1)
    out[level.constrained_dofs_v()] = 0.0
    return out
