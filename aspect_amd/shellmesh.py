"""Host-side spherical-shell mesh hierarchy for BASELINE configs[3] (nsinker_spherical_shell) and
the shell goldens of the reference: the stand-in for what deal.II / p4est give ASPECT's
``spherical shell`` geometry model (source/geometry_model/spherical_shell.cc:315-328:
``GridGenerator::hyper_shell(..., 96, true)`` + ``SphericalManifold``) and its ``MappingQCache(4)``
(source/simulator/core.cc:115-116).  Test / benchmark input generation, like boxmesh.py.

Coarse mesh.  [deal.II] hyper_shell with 96 cells is the 12-cell shell over the faces of a rhombic
dodecahedron, refined once along the spherical manifold: 48 quadrilaterals on the sphere x 2 radial
layers.  That this is the topology the reference runs on is corroborated by its goldens: 150
pressure and 970 velocity nodes on the coarse mesh (50 vertices, 96 edges, 48 faces on the sphere;
(50) x 3 = 150, (50 + 96 + 48) x 5 = 970; tests/spherical_nsinker_gmg_gc/screen-output:9,
tests/hollow_sphere_gmg/screen-output:5).

Geometry.  Every coarse cell is a tree; inside a tree the map from the reference cube is
  x(xi) = [r0 + xi_2 (r1 - r0)] * d(xi_0, xi_1)
with d the transfinite (Coons) blend of the great-circle arcs between the tree's four corner
directions, normalised.  On the tree's edges this is the geodesic SphericalManifold refines along,
at a face centre it has the manifold's weights (edge midpoints 1/2, vertices -1/4), radii are
averaged linearly: the mesh a SphericalManifold produces by recursive bisection differs from it
inside the trees by higher-order terms only, and the reference's exact vertex positions cannot be
reproduced without deal.II anyway (so numbers that depend on them -- the viscosity range of
tests/spherical_nsinker_gmg_gc -- are pinned approximately, DoF counts exactly).
Jacobians: per cell the degree-4 Lagrange interpolant of that map on the Gauss-Lobatto points
(what MappingQ(4) holds), differentiated at the 27 Gauss points.

Cells are numbered tree by tree and in Morton order inside a tree (child c of cell k is 8 k + c, bit
d of c = upper half in local direction d), so the tables have the layout boxmesh.Level has.
"""
from __future__ import annotations

import numpy as np

from . import boxmesh as bm


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _slerp(a, b, t):
    """Point at parameter t on the great-circle arc from unit vectors a to b (broadcasts)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64)[..., None]
    om = np.arccos(np.clip(np.sum(a * b, axis=-1, keepdims=True), -1.0, 1.0))
    so = np.sin(om)
    return (np.sin((1.0 - t) * om) * a + np.sin(t * om) * b) / so


def rhombic_dodecahedron_faces():
    """The 12 rhombic faces as 4 unit directions each, ordered as a quadrilateral in lexicographic
    local order (d00, d10, d01, d11), outward orientation (xi_0 x xi_1 points outwards)."""
    axis = [np.array(v, dtype=np.float64) for v in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1))]
    faces = []
    # one face per pair of adjacent axis vertices (12 edges of the octahedron); its other two corners
    # are the two cube vertices adjacent to both
    for a in range(6):
        for b in range(a + 1, 6):
            if np.dot(axis[a], axis[b]) != 0.0:
                continue
            n = np.cross(axis[a], axis[b])
            c0 = _unit(axis[a] + axis[b] + n)
            c1 = _unit(axis[a] + axis[b] - n)
            quad = [axis[a], c0, c1, axis[b]]          # d00, d10, d01, d11
            # outward orientation
            e0, e1 = quad[1] - quad[0], quad[2] - quad[0]
            if np.dot(np.cross(e0, e1), quad[0] + quad[3]) < 0:
                quad = [quad[0], quad[2], quad[1], quad[3]]
            faces.append(np.stack(quad))
    assert len(faces) == 12
    return faces


def _coons(corners, s, t):
    """Transfinite blend of the four great-circle edges of a spherical quadrilateral with corner
    directions corners[(d00, d10, d01, d11)], normalised.  s, t arrays of the same shape."""
    d00, d10, d01, d11 = corners
    bottom, top = _slerp(d00, d10, s), _slerp(d01, d11, s)
    left, right = _slerp(d00, d01, t), _slerp(d10, d11, t)
    s_, t_ = s[..., None], t[..., None]
    bil = (1 - s_) * (1 - t_) * d00 + s_ * (1 - t_) * d10 + (1 - s_) * t_ * d01 + s_ * t_ * d11
    return _unit((1 - t_) * bottom + t_ * top + (1 - s_) * left + s_ * right - bil)


def coarse_trees(r_inner, r_outer, n_coarse=96):
    """The trees as (corner directions [4][3], r0, r1): 96 (the default hyper_shell of the geometry
    model) or 12 (`Cells along circumference = 12`: the rhombic dodecahedron itself, one radial layer,
    tests/hollow_sphere_gmg.prm; its first refinement is the 96-cell mesh)."""
    if n_coarse == 12:
        return [(quad, r_inner, r_outer) for quad in rhombic_dodecahedron_faces()]
    assert n_coarse == 96
    trees = []
    half = np.array([0.0, 0.5, 1.0])
    rm = 0.5 * (r_inner + r_outer)
    for quad in rhombic_dodecahedron_faces():
        S, Tt = np.meshgrid(half, half, indexing="xy")         # [t][s]
        P = _coons(quad, S, Tt)                                # 3 x 3 directions of the once-refined face
        for j in range(2):
            for i in range(2):
                corners = np.stack([P[j, i], P[j, i + 1], P[j + 1, i], P[j + 1, i + 1]])
                for (ra, rb) in ((r_inner, rm), (rm, r_outer)):
                    trees.append((corners, ra, rb))
    assert len(trees) == 96
    return trees


# Gauss-Lobatto points of degree 4 on [0, 1] (MappingQ support points) and the derivative of the
# Lagrange basis on them at the Gauss-3 points
_GLL5 = 0.5 * (np.array([-1.0, -np.sqrt(3.0 / 7.0), 0.0, np.sqrt(3.0 / 7.0), 1.0]) + 1.0)


def _lagrange(nodes, x):
    """values [len(x)][len(nodes)] and derivatives of the Lagrange basis on `nodes` at x."""
    n = len(nodes)
    V = np.ones((len(x), n))
    D = np.zeros((len(x), n))
    for i in range(n):
        for j in range(n):
            if j != i:
                V[:, i] *= (x - nodes[j]) / (nodes[i] - nodes[j])
        for k in range(n):
            if k == i:
                continue
            term = np.full(len(x), 1.0 / (nodes[i] - nodes[k]))
            for j in range(n):
                if j != i and j != k:
                    term *= (x - nodes[j]) / (nodes[i] - nodes[j])
            D[:, i] += term
    return V, D


class ShellLevel:
    """Flat tables of one level of the shell (the interface of boxmesh.Level that capi.Solver, cases
    and the tests use)."""

    dim = 3
    general = True
    numbering = "first touch"
    periodic = [False, False, False]

    def __init__(self, trees, level, r_inner, r_outer, zero_velocity=True):
        self.trees = trees
        self.level = level
        self.r_inner, self.r_outer = float(r_inner), float(r_outer)
        n1 = 1 << level
        per_tree = n1 ** 3
        self.n_cells = len(trees) * per_tree
        ids = np.arange(self.n_cells, dtype=np.int64)
        self.cell_tree = ids // per_tree
        loc = bm._morton_decode(ids % per_tree, 3, level)
        self.cell_ijk = np.stack(loc, axis=1)            # position inside the tree
        self.h_ref = 1.0 / n1
        corners = np.stack([t[0] for t in trees])        # [tree][4][3]
        self._corners = corners
        self._r0 = np.array([t[1] for t in trees])
        self._r1 = np.array([t[2] for t in trees])
        # nodes: physical positions of the cell-local Q2 / Q1 nodes, identified across cells and trees
        self.cell_nodes_v, self._xv = self._number_nodes(2)
        self.cell_nodes_p, self._xp = self._number_nodes(1)
        self.n_nodes_v, self.n_nodes_p = self._xv.shape[0], self._xp.shape[0]
        rv = np.linalg.norm(self._xv, axis=1)
        tol = 1e-9 * self.r_outer
        on_boundary = (np.abs(rv - self.r_inner) < tol) | (np.abs(rv - self.r_outer) < tol)
        self.cmask_v = np.where(on_boundary & bool(zero_velocity), 7, 0).astype(np.uint8)
        self.cmask_p = np.zeros(self.n_nodes_p, dtype=np.uint8)
        self.cell_h = np.ones((self.n_cells, 3))          # unused (general geometry)
        self._geometry()

    # -- geometry ---------------------------------------------------------
    def map_points(self, cells, xi):
        """Physical positions of reference points xi [n_pts][3] (cell-local, in [0,1]^3) of `cells`:
        [len(cells)][n_pts][3]."""
        tr = self.cell_tree[cells]
        X = (self.cell_ijk[cells][:, None, :] + xi[None, :, :]) * self.h_ref      # tree coordinates
        c = self._corners[tr]                                                      # [cell][4][3]
        d = _coons((c[:, None, 0], c[:, None, 1], c[:, None, 2], c[:, None, 3]), X[:, :, 0], X[:, :, 1])
        r = self._r0[tr][:, None] + X[:, :, 2] * (self._r1[tr] - self._r0[tr])[:, None]
        return r[:, :, None] * d

    def _number_nodes(self, degree):
        n1 = degree + 1
        loc = np.indices((n1,) * 3).reshape(3, -1)[::-1].T / degree             # [n1^3][3], x fastest
        chunk = 1 << 15
        pos = np.empty((self.n_cells, n1 ** 3, 3))
        for a in range(0, self.n_cells, chunk):
            cells = np.arange(a, min(a + chunk, self.n_cells))
            pos[a:a + len(cells)] = self.map_points(cells, loc)
        flat = pos.reshape(-1, 3)
        # identify coincident points: positions computed from different trees agree to rounding, so
        # round to a grid much finer than the mesh and much coarser than the rounding error
        scale = 2.0 ** 30 / self.r_outer
        for shift in (0.0, 0.25, 0.5):
            key = np.floor(flat * scale + shift + 0.5).astype(np.int64)
            uniq, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
            if uniq.shape[0] == self._expected_nodes(degree):
                break
        else:
            raise RuntimeError("shell mesh: node identification failed")
        inv = inv.reshape(-1)
        # number the nodes in the order the cell walk first meets them
        order = np.argsort(first, kind="stable")
        new_of = np.empty(uniq.shape[0], dtype=np.int64)
        new_of[order] = np.arange(uniq.shape[0])
        x = flat[first[order]]
        return new_of[inv].reshape(self.n_cells, n1 ** 3).astype(np.uint32), x

    def _expected_nodes(self, degree):
        # closed quadrilateral surface mesh: F faces, E = 2F edges, V = F + 2 vertices
        coarse_faces, coarse_layers = (48, 2) if len(self.trees) == 96 else (12, 1)
        F = coarse_faces * 4 ** self.level
        V, E = F + 2, 2 * F
        layers = coarse_layers * (1 << self.level)
        if degree == 1:
            return V * (layers + 1)
        return (V + E + F) * (2 * layers + 1)

    def _geometry(self):
        """inv_jacobian [cell][q][e][d] = d xi_e / d x_d and jxw [cell][q] of the degree-4 mapping."""
        V5, D5 = _lagrange(_GLL5, bm.GAUSS3_X)                                  # [q][k]
        sup = np.indices((5,) * 3).reshape(3, -1)[::-1].T                       # [125][3] indices, x fastest
        xi = _GLL5[sup]
        q = np.indices((3,) * 3).reshape(3, -1)[::-1].T                         # [27][3]
        # dN/dxi_e at q-point for support point k: product of 1-D values / derivative
        G = np.empty((3, 27, 125))
        for e in range(3):
            g = np.ones((27, 125))
            for d in range(3):
                M = D5 if d == e else V5
                g *= M[q[:, d]][:, sup[:, d]]
            G[e] = g
        Nq = np.ones((27, 125))
        for d in range(3):
            Nq *= V5[q[:, d]][:, sup[:, d]]
        W = bm.tensor_shape([bm.GAUSS3_W.reshape(3, 1)] * 3).reshape(-1)
        self.inv_jacobian = np.empty((self.n_cells, 27, 3, 3))
        self.jxw = np.empty((self.n_cells, 27))
        self._xq = np.empty((self.n_cells, 27, 3))
        chunk = 1 << 13
        for a in range(0, self.n_cells, chunk):
            cells = np.arange(a, min(a + chunk, self.n_cells))
            X = self.map_points(cells, xi)                                      # [cell][125][3]
            J = np.einsum("eqk,ckd->cqde", G, X)                                # d x_d / d xi_e
            det = np.linalg.det(J)
            if not np.all(det > 0):
                raise ValueError("shell mesh: inverted cell")
            self.inv_jacobian[a:a + len(cells)] = np.linalg.inv(J)
            self.jxw[a:a + len(cells)] = det * W[None, :]
            self._xq[a:a + len(cells)] = np.einsum("qk,ckd->cqd", Nq, X)

    def mapped_quadrature(self, x1d, w1d):
        """Physical positions [cell][q][3] and JxW [cell][q] of the tensor quadrature (x1d, w1d) per
        direction (q lexicographic, x fastest) under the degree-4 mapping: what FEValues gives the
        error postprocessors."""
        nq1 = len(x1d)
        V5, D5 = _lagrange(_GLL5, np.asarray(x1d))
        sup = np.indices((5,) * 3).reshape(3, -1)[::-1].T
        xi = _GLL5[sup]
        q = np.indices((nq1,) * 3).reshape(3, -1)[::-1].T
        nq = q.shape[0]
        G = np.empty((3, nq, 125))
        Nq = np.ones((nq, 125))
        for e in range(3):
            g = np.ones((nq, 125))
            for d in range(3):
                g *= (D5 if d == e else V5)[q[:, d]][:, sup[:, d]]
            G[e] = g
            Nq *= V5[q[:, e]][:, sup[:, e]]
        W = np.asarray(w1d)[q[:, 0]] * np.asarray(w1d)[q[:, 1]] * np.asarray(w1d)[q[:, 2]]
        X = self.map_points(np.arange(self.n_cells), xi)
        J = np.einsum("eqk,ckd->cqde", G, X)
        return np.einsum("qk,ckd->cqd", Nq, X), np.linalg.det(J) * W[None, :]

    def mapped_face_quadrature(self, cells, x1d, w1d, top=True):
        """Positions [face][q][3] and JxW [face][q] on the radial faces (xi_2 = 1 or 0) of `cells`."""
        nq1 = len(x1d)
        V5, D5 = _lagrange(_GLL5, np.asarray(x1d))
        Vz, _ = _lagrange(_GLL5, np.array([1.0 if top else 0.0]))
        sup = np.indices((5,) * 3).reshape(3, -1)[::-1].T
        xi = _GLL5[sup]
        q = np.indices((nq1,) * 2).reshape(2, -1)[::-1].T
        nq = q.shape[0]
        N = V5[q[:, 0]][:, sup[:, 0]] * V5[q[:, 1]][:, sup[:, 1]] * Vz[0][sup[:, 2]][None, :]
        T0 = D5[q[:, 0]][:, sup[:, 0]] * V5[q[:, 1]][:, sup[:, 1]] * Vz[0][sup[:, 2]][None, :]
        T1 = V5[q[:, 0]][:, sup[:, 0]] * D5[q[:, 1]][:, sup[:, 1]] * Vz[0][sup[:, 2]][None, :]
        X = self.map_points(np.asarray(cells, dtype=np.int64), xi)
        t0, t1 = np.einsum("qk,ckd->cqd", T0, X), np.einsum("qk,ckd->cqd", T1, X)
        area = np.linalg.norm(np.cross(t0, t1), axis=-1)
        W = np.asarray(w1d)[q[:, 0]] * np.asarray(w1d)[q[:, 1]]
        return np.einsum("qk,ckd->cqd", N, X), area * W[None, :]

    # -- the rest of boxmesh.Level's interface -------------------------------
    def node_coords_v(self):
        return self._xv

    def node_coords_p(self):
        return self._xp

    def qpoint_coords(self):
        return self._xq

    def cell_centers(self):
        return self.map_points(np.arange(self.n_cells), np.array([[0.5, 0.5, 0.5]]))[:, 0, :]

    def nodal_to_qpoints(self, nodal):
        return nodal[self.cell_nodes_v.astype(np.int64)] @ bm.tensor_shape([bm.N2] * 3).T

    def boundary_faces(self, boundary_id):
        """boundary_id 0 = inner ('bottom'), 1 = outer ('top'): cells and deal.II face numbers (local
        direction 2 is radial in every tree)."""
        n1 = 1 << self.level
        tr = self.cell_tree
        if boundary_id == 0:
            on = (self.cell_ijk[:, 2] == 0) & (np.abs(self._r0[tr] - self.r_inner) < 1e-12)
            fno = 4
        else:
            on = (self.cell_ijk[:, 2] == n1 - 1) & (np.abs(self._r1[tr] - self.r_outer) < 1e-12)
            fno = 5
        cells = np.nonzero(on)[0].astype(np.uint32)
        return cells, np.full(cells.size, fno, dtype=np.uint8)

    @property
    def n_trees(self):
        return len(self.trees)

    @property
    def n_dofs_v(self):
        return 3 * self.n_nodes_v

    @property
    def n_dofs_p(self):
        return self.n_nodes_p

    def constrained_dofs_v(self):
        bits = (self.cmask_v[:, None] >> np.arange(3)[None, :]) & 1
        return bits.reshape(-1).astype(bool)

    def volume(self):
        return float(self.jxw.sum())


class ShellHierarchy:
    """Levels 0..refinement of the 96-cell shell; level `refinement` is also the active mesh."""

    dim = 3

    def __init__(self, refinement, r_inner=0.54, r_outer=1.0, zero_velocity=True, only_finest=False, n_coarse=96):
        trees = coarse_trees(r_inner, r_outer, n_coarse)
        first = refinement if only_finest else 0
        self.levels = [ShellLevel(trees, l, r_inner, r_outer, zero_velocity) for l in range(first, refinement + 1)]
        self.bc = "zero velocity on both spheres" if zero_velocity else "none"

    @property
    def finest(self):
        return self.levels[-1]

    @property
    def n_levels(self):
        return len(self.levels)
