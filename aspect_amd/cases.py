"""Synthetic problem definitions restating the reference's input decks
(SURVEY.md App. C): material/density/gravity functions, boundary conditions,
pressure scaling and solver parameters for the BASELINE.json configs and for
the golden-pinned regression tests of SURVEY.md section 4.

Everything here is host-side input generation (the part ASPECT's material and
geometry plugins do); the outputs are the flat tables the C-ABI consumes.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

import os

from . import boxmesh as bm

# first 8 of the 75 sinker centres, benchmarks/nsinker/nsinker.cc:203-210
NSINKER_CENTERS = np.array([
    [0.24257829890, 0.013469574514, 0.38313885004],
    [0.41465269048, 0.067768972864, 0.99312692973],
    [0.48430804651, 0.76533776604, 0.031833815403],
    [0.030935481671, 0.93264044027, 0.88787953411],
    [0.59132973039, 0.47877868473, 0.83335433660],
    [0.18633519681, 0.73565270739, 0.11505317181],
    [0.69865863058, 0.35560411138, 0.63830000658],
    [0.90821050755, 0.29400041480, 0.26497158886],
])


def nsinker_chi(x, n_sinkers, delta=200.0, omega=0.1):
    """benchmarks/nsinker/nsinker.cc:61-76."""
    chi = np.ones(x.shape[0])
    for s in range(n_sinkers):
        d = np.sqrt(((x - NSINKER_CENTERS[s][None, :]) ** 2).sum(axis=1))
        t = np.maximum(0.0, d - omega / 2)
        chi *= 1.0 - np.exp(-delta * t * t)
    return chi


@dataclass
class StokesProblem:
    name: str
    dim: int
    refinement: int
    bc: list
    eta: callable
    force: callable                 # x -> rho*g vector [n, dim]
    averaging: str
    pressure_scaling: float
    extents: list = None
    reps: list = None
    tol: float = 1e-7
    restart: int = 100
    n_cheap: int = 1000
    n_expensive: int = 1000
    krylov: str = "gmres"
    idr_s: int = 2
    initial_pressure: callable = None   # x -> p0 (unscaled), static pressure guess
    deform: callable = None             # x -> x' smooth map of the box (general / MappingQ geometry)
    periodic: list = None               # periodic directions of the box
    numbering: str = "lexicographic"    # node numbering of the host mesh (boxmesh.Level)
    force_q: callable = None            # level -> rho*g at the q-points [n_cells, nq, dim] (FE-field densities)
    # free-surface stabilisation (mesh deformation with free-surface boundary indicators): dict with
    # boundaries (box boundary ids), density x -> rho, gravity x -> g vector, dt, theta
    free_surface_spec: dict = None
    free_surface: tuple = field(default=None, repr=False)
    hier: bm.BoxHierarchy = field(default=None, repr=False)
    make_hierarchy: callable = None     # (refinement, only_finest) -> hierarchy (other geometries than the box)
    boundary_velocity: callable = None  # x -> prescribed velocity on the constrained nodes (inhomogeneous Dirichlet)
    pressure_scaling_from_mesh: callable = None   # finest level -> lambda (helper_functions.cc:252-350)
    # prescribed boundary tractions {box boundary id: x -> traction vector [n, dim]}: rhs_u += int_Gamma phi . t dS
    # (source/simulator/assemblers/stokes.cc, StokesBoundaryTraction)
    boundary_traction: dict = None
    # prescribed dilation (Formulation: Enable prescribed dilation; assemblers/stokes.cc:453-480): div u = R enters
    # the pressure rows of the right-hand side as -lambda int R phi_p, the operator of the active mesh becomes
    # deviatoric and carries -lambda^2 int L p q in its pressure block.  dilation_rhs / dilation_lhs: x -> R, L
    dilation_rhs: callable = None
    dilation_lhs: callable = None
    # material_model.is_compressible(): every operator carries the -2/3 eta div u term (matrix_free_operators.cc:243-246)
    is_compressible: bool = False
    eta_q: callable = None              # finest level -> viscosity at its q-points [n_cells, nq] (FE-field viscosities)

    def _hierarchy(self, only_finest):
        if self.make_hierarchy is not None:
            return self.make_hierarchy(self.refinement, only_finest)
        return bm.BoxHierarchy(self.dim, self.refinement, self.extents, self.reps, self.bc, only_finest=only_finest,
                               deform=self.deform, periodic=self.periodic, numbering=self.numbering)

    def build(self):
        self.hier = self._hierarchy(False)
        if self.pressure_scaling_from_mesh is not None:
            self.pressure_scaling = self.pressure_scaling_from_mesh(self.hier.finest)
        self.eta_active, self.eta_levels, self.eta_min, self.eta_max = \
            bm.viscosity_tables(self.hier, self.eta, self.averaging,
                                eta_q=None if self.eta_q is None else self.eta_q(self.hier.finest))
        fin = self.hier.finest
        self.rhs_u = bm.assemble_rhs_u(fin, self.force, self.force_q)
        for b, traction in (self.boundary_traction or {}).items():
            self.rhs_u += bm.assemble_boundary_traction(fin, b, traction)
        self.rhs_p = np.zeros(fin.n_dofs_p)
        self.dilation = None
        if self.dilation_rhs is not None:
            self.rhs_p = -self.pressure_scaling * bm.assemble_rhs_p(fin, self.dilation_rhs)
            xq = fin.qpoint_coords().reshape(-1, self.dim)
            lhs = np.zeros(xq.shape[0]) if self.dilation_lhs is None else self.dilation_lhs(xq)
            self.dilation = (np.ascontiguousarray(lhs.reshape(fin.n_cells, -1)), None, None)
        if self.initial_pressure is not None:
            self.p0 = self.initial_pressure(fin.node_coords_p()) / self.pressure_scaling
        else:
            self.p0 = np.zeros(fin.n_dofs_p)
        self._free_surface_tables(fin)
        self.u0 = None
        if self.boundary_velocity is not None:
            # constraints.distribute(0): the zero vector carrying the boundary values (local_smoothing.cc:775-787)
            u = np.zeros((fin.n_nodes_v, self.dim))
            con = fin.cmask_v != 0
            u[con] = self.boundary_velocity(fin.node_coords_v()[con])
            bits = ((fin.cmask_v[:, None] >> np.arange(self.dim)[None, :]) & 1).astype(bool)
            u[~bits] = 0.0
            self.u0 = np.concatenate([u.reshape(-1), np.zeros(fin.n_dofs_p)])
        return self

    def _free_surface_tables(self, fin):
        """free_surface_stabilization_term_table = rho * dt * theta * |g| * g_hat at the face q-points of
        the faces carrying a free-surface indicator (local_smoothing.cc:665-756)."""
        spec = self.free_surface_spec
        if spec is None:
            return
        parts = [fin.boundary_faces(b) for b in spec["boundaries"]]
        cells = np.concatenate([p[0] for p in parts])
        fnos = np.concatenate([p[1] for p in parts])
        x, normal, jxw = fin.face_geometry(cells, fnos)
        xf = x.reshape(-1, self.dim)
        g = spec["gravity"](xf)
        gn = np.linalg.norm(g, axis=1)
        ghat = np.where(gn[:, None] == 0.0, 0.0, g / np.where(gn == 0.0, 1.0, gn)[:, None])
        pert = spec["density"](xf) * spec["dt"] * spec["theta"] * gn
        stab = (pert[:, None] * ghat).reshape(x.shape)
        if fin.general:
            self.free_surface = (cells, fnos, stab, normal, jxw)
        else:
            self.free_surface = (cells, fnos, stab, None, None)

    def build_single_level(self):
        """Only the active mesh (operator benchmarks on big levels: no hierarchy, no RHS)."""
        self.hier = self._hierarchy(True)
        fin = self.hier.finest
        if self.pressure_scaling_from_mesh is not None:
            self.pressure_scaling = self.pressure_scaling_from_mesh(fin)
        nq = 3 ** self.dim
        if self.averaging == bm.AVG_HARMONIC:
            eta_q = self.eta(fin.qpoint_coords().reshape(-1, self.dim)).reshape(fin.n_cells, nq)
            self.eta_active = (nq / np.sum(1.0 / eta_q, axis=1))[:, None]
        else:
            self.eta_active = self.eta(fin.qpoint_coords().reshape(-1, self.dim)).reshape(fin.n_cells, nq)
        self.eta_levels = [self.eta_active]
        self.eta_min, self.eta_max = float(self.eta_active.min()), float(self.eta_active.max())
        self._free_surface_tables(fin)
        return self


def nsinker_box(refinement, n_sinkers=8, ratio=1e6, averaging=bm.AVG_HARMONIC,
                krylov="gmres", tol=1e-6, restart=100, n_cheap=200, n_expensive=0, average_density=False):
    """benchmarks/nsinker/nsinker_gmg.prm + nsinker.cc (config 3).  average_density: the benchmark deck itself asks for
    `Material averaging = harmonic average`, which averages the density of a cell too (material_model/interface.cc:895-896);
    the regression tests of the reference (tests/nsinker_gmg_*.prm) override it with `harmonic average only viscosity`."""
    sr = np.sqrt(ratio)

    def eta(x):
        chi = nsinker_chi(x, n_sinkers)
        return (sr - 1.0 / sr) * (1.0 - chi) + 1.0 / sr

    def force(x):
        chi = nsinker_chi(x, n_sinkers)
        f = np.zeros_like(x)
        f[:, -1] = -1.0 * 10.0 * (1.0 - chi)     # gravity vertical magnitude 1, pointing down
        return f

    def force_q(level):
        # `Material averaging = harmonic average` averages the density of a cell too (material_model/interface.cc:895-896)
        xq = level.qpoint_coords()
        rho = 10.0 * (1.0 - nsinker_chi(xq.reshape(-1, 3), n_sinkers)).reshape(level.n_cells, -1)
        with np.errstate(divide="ignore"):
            rho = np.repeat((rho.shape[1] / (1.0 / rho).sum(axis=1))[:, None], rho.shape[1], axis=1)
        f = np.zeros(xq.shape)
        f[:, :, -1] = -1.0 * rho
        return f

    return StokesProblem(
        name=f"nsinker_r{refinement}", dim=3, refinement=refinement,
        bc=[bm.BC_ZERO] * 6, eta=eta, force=force, force_q=force_q if (average_density and averaging == bm.AVG_HARMONIC) else None, averaging=averaging,
        pressure_scaling=1.0,                     # nsinker.cc:320-331 forces lambda = 1
        tol=tol, restart=restart, n_cheap=n_cheap, n_expensive=n_expensive, krylov=krylov)


def convection_box(dim, refinement):
    """cookbooks/convection-box (2-D, config 1) / convection_box_3d (configs 2, 5)."""
    if dim == 2:
        g, alpha = 1e4, 1.0
    else:
        g, alpha = 1e16, 1e-10

    def temperature(x):
        z = x[:, -1]
        t = (1 - z) - 0.01 * np.cos(np.pi * x[:, 0]) * np.sin(np.pi * z)
        if dim == 3:
            t = (1 - z) - 0.01 * np.cos(np.pi * x[:, 0]) * np.sin(np.pi * z) * x[:, 1] ** 3
        return t

    def force(x):
        rho = 1.0 * (1.0 - alpha * temperature(x))
        f = np.zeros_like(x)
        f[:, -1] = -g * rho
        return f

    def p_static(x):
        # adiabatic (reference-profile) pressure used as the first linearisation
        # point: rho_ref * g * depth with the reference temperature T = 0
        return 1.0 * g * (1.0 - x[:, -1])

    return StokesProblem(
        name=f"convection_box_{dim}d_r{refinement}", dim=dim, refinement=refinement,
        bc=[bm.BC_TANGENTIAL] * (2 * dim), eta=lambda x: np.ones(x.shape[0]),
        force=force, averaging=bm.AVG_PROJECT_Q1, pressure_scaling=100.0,
        initial_pressure=p_static)


def free_surface_blob(dim, refinement, dt_theta=0.5, deform=None, side_surface=False):
    """A dense blob under a free surface (the setting of tests/free_surface_blob_gmg.prm, whose time
    stepping cannot run here): box with free slip on the sides and bottom, a free top boundary that
    carries the stabilisation term; ``side_surface`` also frees the x+ side (faces of two directions
    meeting in an edge)."""
    g0 = 10.0

    def rho(x):
        r2 = np.sum((x - 0.5) ** 2, axis=1)
        return 1.0 + 0.5 * np.exp(-r2 / 0.02)

    def gravity(x):
        g = np.zeros_like(x)
        g[:, -1] = -g0
        return g

    def force(x):
        return rho(x)[:, None] * gravity(x)

    bc = [bm.BC_TANGENTIAL] * (2 * dim)
    free = [2 * dim - 1] + ([1] if side_surface else [])
    for b in free:
        bc[b] = bm.BC_NONE
    return StokesProblem(
        name=f"free_surface_blob_{dim}d_r{refinement}", dim=dim, refinement=refinement, bc=bc,
        eta=lambda x: np.exp(2.0 * x[:, 0] + np.sin(3.0 * x[:, -1] + x[:, 0])), force=force, averaging=bm.AVG_HARMONIC, pressure_scaling=1.0,
        initial_pressure=lambda x: g0 * (1.0 - x[:, -1]), deform=deform, tol=1e-7, n_cheap=200, n_expensive=0,
        free_surface_spec=dict(boundaries=free, density=rho, gravity=gravity, dt=dt_theta, theta=1.0))


def solcx(refinement, averaging=bm.AVG_HARMONIC, jump=1e6, n_cheap=1000, average_density=False):
    """benchmarks/solcx: eta = 1 (x < 1/2) | jump, rho = -sin(pi z) cos(pi x)
    (solcx.h:2969-2970, 3018); gravity vertical magnitude 1 default; free slip.
    average_density: `Material averaging = harmonic average` (tests/no_cheap_gmg.prm:57) averages the
    density over the cell's q-points too (material_model/interface.cc:542-561, 890-896: unweighted
    N / sum(1/rho_q), and 0 if any rho_q is exactly 0), unlike `harmonic average only viscosity`."""

    def eta(x):
        return np.where(x[:, 0] < 0.5, 1.0, jump)

    def density(x):
        return -np.sin(np.pi * x[:, 1]) * np.cos(np.pi * x[:, 0])

    def force(x):
        f = np.zeros_like(x)
        f[:, 1] = -1.0 * density(x)
        return f

    def force_q(level):
        xq = level.qpoint_coords()
        rho = density(xq.reshape(-1, 2)).reshape(level.n_cells, -1)
        with np.errstate(divide="ignore"):
            avg = rho.shape[1] / np.sum(1.0 / rho, axis=1)
        avg[np.any(rho == 0.0, axis=1)] = 0.0
        f = np.zeros(rho.shape + (2,))
        f[:, :, 1] = -1.0 * avg[:, None]
        return f

    # pressure scaling: exp(mean log eta)/L with L = 0.01 (helper_functions.cc:286-345)
    lam = np.exp(0.5 * np.log(1.0) + 0.5 * np.log(jump)) / 0.01
    return StokesProblem(
        name=f"solcx_r{refinement}", dim=2, refinement=refinement,
        bc=[bm.BC_TANGENTIAL] * 4, eta=eta, force=force, averaging=averaging,
        pressure_scaling=lam, n_cheap=n_cheap, force_q=force_q if average_density else None)


def stokes_sinker_box(refinement):
    """tests/stokes.prm: 3-D isoviscous box 2890 km, free slip, eta = 1e22, `simple` material:
    rho = 3300 (1 - 2e-5 (T - 293)) + 100 C with T = 0, g = 9.81.  C is the compositional FIELD
    (Q2, interpolated from the indicator of a sphere r = 200 km at the centre), evaluated at the
    q-points, not the indicator itself."""
    L = 2890000.0
    rho0 = 3300.0 * (1.0 + 2e-5 * 293.0)      # "Reference density (kg/m^3): 3319.34" of the golden

    def force_q(level):
        xn = level.node_coords_v()
        c_nodal = np.where(np.sqrt(((xn - 1445000.0) ** 2).sum(axis=1)) > 200000.0, 0.0, 1.0)
        cq = level.nodal_to_qpoints(c_nodal)
        f = np.zeros(cq.shape + (3,))
        f[:, :, 2] = -9.81 * (rho0 + 100.0 * cq)
        return f

    def p_static(x):
        return rho0 * 9.81 * (L - x[:, 2])

    return StokesProblem(
        name=f"stokes_sinker_r{refinement}", dim=3, refinement=refinement,
        bc=[bm.BC_TANGENTIAL] * 6, eta=lambda x: np.full(x.shape[0], 1e22),
        force=None, force_q=force_q, averaging=bm.AVG_PROJECT_Q1, pressure_scaling=1e22 / (0.01 * L),
        extents=[L, L, L], initial_pressure=p_static)


def deformed_sinker_box(dim, refinement, amplitude=0.06, ratio=1e3, averaging=bm.AVG_NONE, **kw):
    """Curved-cell stand-in for the MappingQ meshes of config 4 (spherical shell,
    SURVEY.md 8 config 4): the unit box under a smooth interior displacement that
    vanishes on the boundary (so zero-velocity walls stay exact), every cell with a
    full, q-point-dependent Jacobian; sinker viscosity/density as in nsinker."""
    sr = np.sqrt(ratio)
    centers = NSINKER_CENTERS[:4, :dim]

    def deform(x):
        bump = np.prod(np.sin(np.pi * x), axis=1)
        y = x.copy()
        for d in range(dim):
            y[:, d] += amplitude * bump * np.cos(2.0 * np.pi * x[:, (d + 1) % dim] + 0.7 * d)
        return y

    def chi(x):
        c = np.ones(x.shape[0])
        for s in range(centers.shape[0]):
            r = np.sqrt(((x - centers[s][None, :]) ** 2).sum(axis=1))
            t = np.maximum(0.0, r - 0.05)
            c *= 1.0 - np.exp(-200.0 * t * t)
        return c

    def eta(x):
        return (sr - 1.0 / sr) * (1.0 - chi(x)) + 1.0 / sr

    def force(x):
        f = np.zeros_like(x)
        f[:, -1] = -10.0 * (1.0 - chi(x))
        return f

    args = dict(tol=1e-6, n_cheap=200, n_expensive=0)
    args.update(kw)
    return StokesProblem(
        name=f"deformed_sinker_{dim}d_r{refinement}", dim=dim, refinement=refinement,
        bc=[bm.BC_ZERO] * (2 * dim), eta=eta, force=force, averaging=averaging,
        pressure_scaling=1.0, deform=deform, **args)


def periodic_box_2d(refinement=5, no_slip_walls=False):
    """tests/periodic_box_gmg.prm, first Stokes solve: 2-D box 1e6 x 5e5, periodic in x, free slip on
    top and bottom, isoviscous 1e20 (harmonic averaging), two hot blobs (T = 800) in a cold box with
    the bottom boundary temperature 1000 imposed on the bottom nodes; density 3300 (1 - 4e-5 T),
    g = 10; tolerance 1e-8, 100 cheap steps.
    With free slip the A block has the x-translation in its kernel (the deck removes it afterwards,
    "Remove nullspace = net x translation"), so the 100-step CG eigenvalue estimate of the 12-DoF
    coarse level runs on rounding noise and everything downstream of it (smoother ranges, iteration
    counts) is rounding-dependent; no_slip_walls = True gives the well-posed variant."""
    X, Y = 1.0e6, 5.0e5

    def temperature_nodal(x):
        t = np.where((np.sqrt((x[:, 0] - 1e5) ** 2 + (x[:, 1] - 4.0e5) ** 2) < 5.0e4) |
                     (np.sqrt((x[:, 0] - 3e5) ** 2 + (x[:, 1] - 2e5) ** 2) < 1.0e5), 800.0, 0.0)
        t[x[:, 1] <= 0.0] = 1000.0           # fixed temperature boundary: bottom 1000, top 0
        t[x[:, 1] >= Y] = 0.0
        return t

    def force_q(level):
        tq = level.nodal_to_qpoints(temperature_nodal(level.node_coords_v()))   # Q2 temperature field
        f = np.zeros(tq.shape + (2,))
        f[:, :, 1] = -10.0 * 3300.0 * (1.0 - 4e-5 * tq)
        return f

    def p_static(x):
        return 3300.0 * 10.0 * (Y - x[:, 1])

    return StokesProblem(
        name=f"periodic_box_2d_r{refinement}", dim=2, refinement=refinement,
        bc=[bm.BC_NONE, bm.BC_NONE] + [bm.BC_ZERO if no_slip_walls else bm.BC_TANGENTIAL] * 2,
        eta=lambda x: np.full(x.shape[0], 1e20), force=None, force_q=force_q,
        averaging=bm.AVG_HARMONIC, pressure_scaling=1e20 / (0.01 * X), extents=[X, Y],
        periodic=[True, False], tol=1e-8, n_cheap=100, n_expensive=1000, initial_pressure=p_static)


def prescribed_dilation_2d(refinement=3, eta=10.0):
    """tests/prescribed_dilation_gmg.prm (+ prescribed_dilation.prm / .cc): manufactured solution u = (cos y,
    sin x + x y), p = -2/3 eta (x - 1) on [0, 2]^2 with div u = x prescribed as dilation, the body force
    (-eta (1 - cos y), eta sin x) as additional Stokes right-hand side, the analytic velocity on all four
    boundaries; eta = 10, harmonic averaging, tolerance 1e-12, 200 cheap steps."""
    X = 2.0

    def force(x):
        return np.stack([-eta * (1.0 - np.cos(x[:, 1])), eta * np.sin(x[:, 0])], axis=1)

    def velocity(x):
        return np.stack([np.cos(x[:, 1]), np.sin(x[:, 0]) + x[:, 0] * x[:, 1]], axis=1)

    pr = StokesProblem(
        name=f"prescribed_dilation_2d_r{refinement}", dim=2, refinement=refinement, bc=[bm.BC_ZERO] * 4,
        eta=lambda x: np.full(x.shape[0], eta), force=force, averaging=bm.AVG_HARMONIC,
        pressure_scaling=eta / (0.01 * X), extents=[X, X], tol=1e-12, restart=100, n_cheap=200, n_expensive=1000,
        boundary_velocity=velocity, dilation_rhs=lambda x: x[:, 0])
    pr.analytic_velocity = velocity
    pr.analytic_pressure = lambda x: -2.0 / 3.0 * eta * (x[:, 0] - 1.0)
    return pr


def geometric_average_box(refinement=1):
    """tests/geometric_average_only_viscosity.prm: box 100 x 100, T = 200 - x, `simple` material with
    eta = 1e14 exp(-ln(1e4) (T - 100) / 100) (1e10 .. 1e14), rho = 3300 (1 - 2e-5 (T - 100)), g = 10, free slip on all
    sides, geometric averaging of the viscosity per cell; default solver parameters (tolerance 1e-7, 200 cheap steps,
    restart 100 since 3.1.0-pre); pressure scaling = exp(mean log eta at the cell midpoints) / (0.01 X)
    (helper_functions.cc:252-350)."""
    X = 100.0

    def temperature(x):
        return 200.0 - x[:, 0]

    def eta(x):
        return 1e14 * np.clip(np.exp(-9.210340372 * (temperature(x) - 100.0) / 100.0), 1e-6, 1e2)

    def force(x):
        f = np.zeros_like(x)
        f[:, 1] = -10.0 * 3300.0 * (1.0 - 2e-5 * (temperature(x) - 100.0))
        return f

    def scaling(fin):
        return float(np.exp(np.log(eta(fin.cell_centers())).mean())) / (0.01 * X)

    return StokesProblem(
        name=f"geometric_average_box_r{refinement}", dim=2, refinement=refinement, bc=[bm.BC_TANGENTIAL] * 4,
        eta=eta, force=force, averaging=bm.AVG_GEOMETRIC, pressure_scaling=1.0, extents=[X, X],
        tol=1e-7, restart=100, n_cheap=200, n_expensive=1000, pressure_scaling_from_mesh=scaling)


def free_surface_blob_deck(refinement=3):
    """tests/free_surface_blob_gmg.prm (+ free_surface_blob.prm), first Stokes solve (time step 0 on the first, uniform
    mesh): box 500 km x 200 km of 5 x 2 coarse cells, eta = 1e21, a hot blob (T = 200 within 25 km of the centre, as
    a Q2 field) with rho = 3300 (1 - 4e-5 T), g = 10; free slip left and right, no slip at the bottom, the top is the
    free surface: open, and its stabilisation term vanishes with dt = 0.  No cheap steps: FGMRES with the inner CG
    solves only; tolerance 1e-7."""
    X, Y = 500.0e3, 200.0e3

    def temperature_nodal(x):
        return np.where(np.sqrt((x[:, 0] - 250e3) ** 2 + (x[:, 1] - 100e3) ** 2) < 25e3, 200.0, 0.0)

    def force_q(level):
        tq = level.nodal_to_qpoints(temperature_nodal(level.node_coords_v()))
        rho = 3300.0 * (1.0 - 4e-5 * tq)
        # `Material averaging = harmonic average` averages the density too (material_model/interface.cc:895-896)
        rho = np.repeat((rho.shape[1] / (1.0 / rho).sum(axis=1))[:, None], rho.shape[1], axis=1)
        f = np.zeros(tq.shape + (2,))
        f[:, :, 1] = -10.0 * rho
        return f

    return StokesProblem(
        name=f"free_surface_blob_deck_r{refinement}", dim=2, refinement=refinement,
        bc=[bm.BC_TANGENTIAL, bm.BC_TANGENTIAL, bm.BC_ZERO, bm.BC_NONE],
        eta=lambda x: np.full(x.shape[0], 1e21), force=None, force_q=force_q, averaging=bm.AVG_HARMONIC,
        pressure_scaling=1e21 / (0.01 * X), extents=[X, Y], reps=[5, 2], tol=1e-7, restart=100, n_cheap=0, n_expensive=1000,
        initial_pressure=lambda x: 3300.0 * 10.0 * (Y - x[:, 1]))


def poiseuille_2d(refinement=4):
    """tests/poiseuille_2d_pressure_bc_gmg.prm: box 2 x 1 (one coarse cell), eta = 1, no gravity, no slip on top and
    bottom, the traction (2, 0) prescribed on the open left and right boundaries (pressure 2 at the inlet, -2 at the
    outlet); harmonic averaging, tolerance 1e-6, 200 cheap steps, restart 100."""
    X, Y = 2.0, 1.0

    def traction(x):
        t = np.zeros_like(x)
        t[:, 0] = 2.0
        return t

    return StokesProblem(
        name=f"poiseuille_2d_r{refinement}", dim=2, refinement=refinement,
        bc=[bm.BC_NONE, bm.BC_NONE, bm.BC_ZERO, bm.BC_ZERO],
        eta=lambda x: np.ones(x.shape[0]), force=lambda x: np.zeros_like(x),
        averaging=bm.AVG_HARMONIC, pressure_scaling=1.0 / (0.01 * X), extents=[X, Y],
        tol=1e-6, restart=100, n_cheap=200, n_expensive=0, boundary_traction={0: traction, 1: traction})


def simple_compressibility_box(refinement=3):
    """tests/simple_compressibility_iterated_stokes_gmg.prm (+ simple_compressibility_iterated_stokes.prm): unit box, the
    `simple compressible` material (rho = rho0 exp(beta p), rho0 = 0.90909..., beta = 0.01, eta = 1), vertical gravity 10,
    free slip on the sides, inflow (0, 1) through the bottom, open top; isentropic compression, harmonic averaging.  The
    density and with it both right-hand sides depend on the iterate: tests/solution_checks.py drives the iterated-Stokes
    loop and fills rhs_u / rhs_p per nonlinear iteration."""
    def inflow(x):
        u = np.zeros_like(x)
        u[:, 1] = 1.0
        return u

    p = StokesProblem(
        name=f"simple_compressibility_r{refinement}", dim=2, refinement=refinement,
        bc=[bm.BC_TANGENTIAL, bm.BC_TANGENTIAL, bm.BC_ZERO, bm.BC_NONE],
        eta=lambda x: np.ones(x.shape[0]), force=lambda x: np.zeros_like(x),
        averaging=bm.AVG_HARMONIC, pressure_scaling=1.0 / 0.01, extents=[1.0, 1.0],
        tol=1e-7, restart=100, n_cheap=1000, n_expensive=1000, boundary_velocity=inflow, is_compressible=True)
    p.rho0, p.beta, p.gravity = 0.909090909090, 0.01, np.array([0.0, -10.0])
    return p


def nonlinear_channel_flow_tractions(refinement=4):
    """tests/nonlinear_channel_flow_tractions_Newton_Stokes_GMG.prm (benchmarks/newton_solver_benchmark_set/
    nonlinear_channel_flow/input_t.prm + input_v.prm): box 10 km x 8 km from one coarse cell, no slip on the sides,
    u_x = 0 on top and bottom with the traction (0, 1e9) on the bottom and (0, 0) on the top (`bottom y: function`), no
    gravity, power-law rheology `simple nonlinear` (n = 3, A = 1e-37, 1e19 <= eta <= 1e24), harmonic averaging, linear
    tolerance 1e-7.  The viscosity depends on the iterate: tests/solution_checks.py drives the defect-correction Picard /
    Newton loop; the tables built here are those of the first iteration (u = 0: eta = 1e24)."""
    X, Y = 10e3, 8e3

    def traction(x):
        t = np.zeros_like(x)
        t[:, 1] = np.where(x[:, 1] > 4e3, 0.0, 1e9)
        return t

    p = StokesProblem(
        name=f"nonlinear_channel_tractions_r{refinement}", dim=2, refinement=refinement,
        bc=[bm.BC_ZERO, bm.BC_ZERO, bm.BC_COMPONENTS + 1, bm.BC_COMPONENTS + 1],
        eta=lambda x: np.full(x.shape[0], 1e24), force=lambda x: np.zeros_like(x),
        averaging=bm.AVG_HARMONIC, pressure_scaling=1e24 / (0.01 * X), extents=[X, Y],
        tol=1e-7, restart=100, n_cheap=1000, n_expensive=1000, boundary_traction={2: traction, 3: traction})
    p.rheology = dict(prefactor=1e-37, n=3.0, eta_min=1e19, eta_max=1e24, min_strain_rate=1.96e-40)
    p.length_scale = 0.01 * X
    return p


def nonlinear_channel_flow_velocities(refinement=4):
    """tests/nonlinear_channel_flow_velocities_Newton_Stokes_GMG.prm (+ nonlinear_channel_flow_velocities_Newton_Stokes.prm):
    the channel of `nonlinear_channel_flow_tractions` with the analytic power-law profile prescribed as velocity on top and
    bottom (all boundaries Dirichlet: the pressure is fixed up to a constant), minimum strain rate 1e-16, the rheology on
    the deviator of the strain rate, linear tolerance 1e-6; 30 defect-correction Picard steps (the Newton switch is off)."""
    X, Y, n = 10e3, 8e3, 3.0

    def profile(x):
        u = np.zeros_like(x)
        u[:, 1] = (1e-37 / (n + 1)) * ((1e9 / 8e3) ** n) * ((5e3) ** (n + 1) - (x[:, 0] - 5e3) ** (n + 1))
        return u

    p = StokesProblem(
        name=f"nonlinear_channel_velocities_r{refinement}", dim=2, refinement=refinement, bc=[bm.BC_ZERO] * 4,
        eta=lambda x: np.full(x.shape[0], 1e24), force=lambda x: np.zeros_like(x),
        averaging=bm.AVG_HARMONIC, pressure_scaling=1e24 / (0.01 * X), extents=[X, Y],
        tol=1e-6, restart=100, n_cheap=1000, n_expensive=1000, boundary_velocity=profile)
    p.rheology = dict(prefactor=1e-37, n=n, eta_min=1e19, eta_max=1e24, min_strain_rate=1e-16, use_deviator=True)
    p.length_scale = 0.01 * X
    return p


def tosi_benchmark(refinement=4):
    """tests/tosi_benchmark_gmg_newton_solver.prm (benchmarks/newton_solver_benchmark_set/tosi_et_al_2015/input.prm +
    benchmarks/tosi_et_al_2015_gcubed/tosi.cc), time step 0: unit box, free slip, gravity 1e8, rho = 1 - 1e-6 T with
    T = (1 - z) + 0.01 cos(pi x) sin(pi z) interpolated on the Q2 temperature space, visco-plastic Tosi rheology (eta = 0.1
    where the strain rate vanishes), harmonic averaging (density too), tolerance 1e-6, 200 cheap steps.  Viscosity and both
    right-hand sides depend on the iterate: tests/solution_checks.py drives the Picard / Newton loop."""
    pi = 3.1415926536            # the deck's own constant

    def temperature(x):
        return (1.0 - x[:, 1]) + 0.01 * np.cos(pi * x[:, 0]) * np.sin(pi * x[:, 1])

    def force_q(level):
        Tq = level.nodal_to_qpoints(temperature(level.node_coords_v()))
        rho = 1.0 * (1.0 - 1e-6 * (Tq - 0.0))
        rho = np.repeat((rho.shape[1] / (1.0 / rho).sum(axis=1))[:, None], rho.shape[1], axis=1)
        f = np.zeros(rho.shape + (2,))
        f[:, :, 1] = -1e8 * rho
        return f

    p = StokesProblem(
        name=f"tosi_r{refinement}", dim=2, refinement=refinement, bc=[bm.BC_TANGENTIAL] * 4,
        eta=lambda x: np.full(x.shape[0], 0.1), force=None, force_q=force_q,
        averaging=bm.AVG_HARMONIC, pressure_scaling=0.1 / 0.01, extents=[1.0, 1.0],
        tol=1e-6, restart=100, n_cheap=200, n_expensive=1000,
        initial_pressure=lambda x: 1e8 * (1.0 - x[:, 1]))          # adiabatic profile of rho = 1 (compute_profile.cc)
    p.temperature = temperature
    p.rheology = dict(kind="tosi", eta_T=1e5, eta_Z=10.0, sigma_yield=1.0, eta_asterisk=1e-3, eta_initial=1e-1,
                      eta_min=1e-6, eta_max=10.0)
    p.length_scale = 0.01
    return p


def tangurnis_tala(refinement=3):
    """tests/tangurnis_tala_gmg.prm (benchmarks/tangurnis/tala/tan.prm + code/tangurnis.cc): unit box, free slip, gravity 1,
    Di = 0.5, gamma = 1, a = 0: eta = Di, rho = -Di T exp(Di depth) with the analytic T = sin(pi y) cos(pi x); truncated
    anelastic: `Mass conservation = reference density profile` with rho_ref = exp(Di depth); compressible operators, harmonic
    averaging of the viscosity only.  The pressure right-hand side depends on the iterate (tests/solution_checks.py)."""
    Di, gamma = 0.5, 1.0

    def force(x):
        T = np.sin(np.pi * x[:, 1]) * np.cos(np.pi * x[:, 0])
        rho = Di * (-1.0 * T) * np.exp(Di / gamma * (1.0 - x[:, 1]))
        f = np.zeros_like(x)
        f[:, 1] = -1.0 * rho
        return f

    p = StokesProblem(
        name=f"tangurnis_tala_r{refinement}", dim=2, refinement=refinement, bc=[bm.BC_TANGENTIAL] * 4,
        eta=lambda x: np.full(x.shape[0], Di), force=force, averaging=bm.AVG_HARMONIC,
        pressure_scaling=Di / 0.01, extents=[1.0, 1.0], tol=1e-7, restart=100, n_cheap=1000, n_expensive=1000,
        is_compressible=True)
    p.Di, p.gamma, p.gravity = Di, gamma, np.array([0.0, -1.0])
    return p


def q1_eulerian_deformation(dim, refinement, extents, boundary_displacement):
    """The mesh of `Mesh deformation` with an initial deformation: the Q1 displacement field that solves the vector Laplace
    problem of MeshDeformationHandler::compute_mesh_displacements (mesh_deformation/interface.cc:832-985) on the finest mesh
    with the displacement `boundary_displacement(x) -> [n, dim]` on the top boundary and zero on the others (the zero
    constraints win in the corners, :821), and the per-level Q1 Eulerian mappings: level vertices carry the fine values
    (interpolate_to_mg, :1562), every other point of a level cell is mapped bilinearly.  -> deform(x, level), 2-D boxes."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    assert dim == 2
    n = 1 << refinement
    hx, hy = extents[0] / n, extents[1] / n
    nv = n + 1
    idx = lambda i, j: j * nv + i
    # Q1 Laplace stiffness of one hx x hy cell (exact for the 2 x 2 Gauss rule)
    a, b = hy / hx, hx / hy
    Kx = np.array([[2, -2, 1, -1], [-2, 2, -1, 1], [1, -1, 2, -2], [-1, 1, -2, 2]]) * a / 6.0
    Ky = np.array([[2, 1, -2, -1], [1, 2, -1, -2], [-2, -1, 2, 1], [-1, -2, 1, 2]]) * b / 6.0
    Ke = Kx + Ky                                         # local order (0,0), (1,0), (0,1), (1,1)
    rows, cols, vals = [], [], []
    for j in range(n):
        for i in range(n):
            loc = [idx(i, j), idx(i + 1, j), idx(i, j + 1), idx(i + 1, j + 1)]
            for r in range(4):
                for c in range(4):
                    rows.append(loc[r]); cols.append(loc[c]); vals.append(Ke[r, c])
    K = sp.csr_matrix((vals, (rows, cols)), shape=(nv * nv, nv * nv))
    ii, jj = np.meshgrid(np.arange(nv), np.arange(nv), indexing="xy")
    X = np.stack([ii.reshape(-1) * hx, jj.reshape(-1) * hy], axis=1)
    on_boundary = (ii == 0) | (ii == n) | (jj == 0) | (jj == n)
    top = (jj == n) & (ii > 0) & (ii < n)
    d = np.zeros((nv * nv, 2))
    d[top.reshape(-1)] = boundary_displacement(X[top.reshape(-1)])
    free = ~on_boundary.reshape(-1)
    for c in range(2):
        rhs = -(K[free][:, ~free] @ d[~free, c])
        d[free, c] = spla.spsolve(K[free][:, free].tocsc(), rhs)
    D = d.reshape(nv, nv, 2)                              # [j][i]

    def deform(x, level):
        m = 1 << level.level                              # cells per direction on this level
        step = n // m
        Dl = D[::step, ::step]                            # level vertices: the fine values
        fx, fy = x[:, 0] / (extents[0] / m), x[:, 1] / (extents[1] / m)
        ix = np.minimum(np.floor(fx + 1e-12).astype(int), m - 1)
        iy = np.minimum(np.floor(fy + 1e-12).astype(int), m - 1)
        tx, ty = (fx - ix)[:, None], (fy - iy)[:, None]
        disp = ((1 - tx) * (1 - ty) * Dl[iy, ix] + tx * (1 - ty) * Dl[iy, ix + 1]
                + (1 - tx) * ty * Dl[iy + 1, ix] + tx * ty * Dl[iy + 1, ix + 1])
        return x + disp

    deform.per_level = True
    return deform


def mesh_deform_prescribed_box(refinement=3):
    """tests/gmg_mesh_deform_prescribed.prm (+ box_initial_mesh_deformation.prm / .cc): box 660 km x 660 km, top boundary
    displaced by 20 km cos(2 pi x / 660 km) against gravity before the first time step, no slip everywhere, eta = 1e21,
    constant temperature (uniform density 3340 (1 - 3e-5 (1613 - 293))), gravity 10: the exact solution is u = 0 with a
    hydrostatic pressure; the solve starts from the adiabatic pressure rho g depth with depth clipped at the undeformed
    surface (geometry_model/box.cc:261-273).  Pins the per-level Q1 Eulerian mappings of a deformed mesh."""
    L = 660000.0
    rho = 3340.0 * (1.0 - 3e-5 * (1613.0 - 293.0))

    def top(x):
        d = np.zeros_like(x)
        d[:, 1] = 20000.0 * np.cos(2.0 * np.pi * x[:, 0] / 660000.0)
        return d

    def force(x):
        f = np.zeros_like(x)
        f[:, 1] = -10.0 * rho
        return f

    return StokesProblem(
        name=f"mesh_deform_prescribed_r{refinement}", dim=2, refinement=refinement, bc=[bm.BC_ZERO] * 4,
        eta=lambda x: np.full(x.shape[0], 1e21), force=force, averaging=bm.AVG_HARMONIC,
        pressure_scaling=1e21 / (0.01 * L), extents=[L, L], tol=1e-7, restart=100, n_cheap=200, n_expensive=1000,
        deform=q1_eulerian_deformation(2, refinement, [L, L], top),
        initial_pressure=lambda x: rho * 10.0 * np.clip(L - x[:, 1], 0.0, L))


def crameri_case_1(refinement=1):
    """tests/crameri_benchmark_1_gmg.prm (benchmarks/crameri_et_al/case_1/crameri_benchmark_1.prm + .cc), first Stokes solve:
    `rebound box` 2800 km x 700 km of 60 x 15 coarse cells whose vertices are lifted by 7 km cos(2 pi x / X) y / Y (children
    of a coarse cell are bilinear in it), one global refinement; eta = 1e21 * 100^C with the Q2 composition field C
    interpolated from `y > 600 km`, rho = 3300, g = 10, free slip on the sides, no slip at the bottom, free surface on top
    (its stabilisation vanishes with dt = 0); harmonic averaging, no cheap steps, no pressure normalisation; the solve starts
    from the adiabatic pressure rho g (Y + 7 km - y)."""
    X, Y, A = 28.0e5, 7.0e5, 7.0e3
    reps = [60, 15]

    def deform(x, level):
        # bilinear in the coarse cells: displacement of the coarse vertices, interpolated
        hx, hy = X / reps[0], Y / reps[1]
        fx, fy = x[:, 0] / hx, x[:, 1] / hy
        ix = np.minimum(np.floor(fx + 1e-12).astype(int), reps[0] - 1)
        iy = np.minimum(np.floor(fy + 1e-12).astype(int), reps[1] - 1)
        tx, ty = fx - ix, fy - iy
        lift = lambda i, j: np.cos(2.0 * np.pi * (i * hx) / X) * A * (j * hy) / Y
        dy = ((1 - tx) * (1 - ty) * lift(ix, iy) + tx * (1 - ty) * lift(ix + 1, iy)
              + (1 - tx) * ty * lift(ix, iy + 1) + tx * ty * lift(ix + 1, iy + 1))
        out = x.copy()
        out[:, 1] += dy
        return out

    deform.per_level = True

    def eta_of_c(c):
        return 10.0 ** ((1.0 - c) * np.log10(1e21) + c * np.log10(1e21 * 100.0))

    def eta_q(level):
        c = np.where(level.node_coords_v()[:, 1] > 600.0e3, 1.0, 0.0)
        return eta_of_c(level.nodal_to_qpoints(c))

    def scaling(level):
        mid = np.array([0.5])
        c = np.where(level.node_coords_v()[:, 1] > 600.0e3, 1.0, 0.0)[level.cell_nodes_v.astype(np.int64)]
        cm = (c @ bm.tensor_shape([bm.q2_shape(mid)] * 2).T)[:, 0]
        w = level.jxw.sum(axis=1)
        return float(np.exp((np.log(eta_of_c(cm)) * w).sum() / w.sum())) / (0.01 * X)

    def force(x):
        f = np.zeros_like(x)
        f[:, 1] = -10.0 * 3300.0
        return f

    return StokesProblem(
        name=f"crameri_case_1_r{refinement}", dim=2, refinement=refinement,
        bc=[bm.BC_TANGENTIAL, bm.BC_TANGENTIAL, bm.BC_ZERO, bm.BC_NONE],
        eta=None, eta_q=eta_q, force=force, averaging=bm.AVG_HARMONIC, pressure_scaling=1.0, extents=[X, Y], reps=reps,
        tol=1e-7, restart=100, n_cheap=0, n_expensive=1000, deform=deform, pressure_scaling_from_mesh=scaling,
        initial_pressure=lambda x: 3300.0 * 10.0 * (Y + A - x[:, 1]))


def mesh_deform_topo_first_solve(refinement=4):
    """tests/gmg_mesh_deform_topo.prm, first Stokes solve (t = 0: the mesh is still undeformed): unit box, free slip on top
    and bottom, no slip on the sides, eta = 1, gravity 1e5, rho = 1 - T with the Q2 interpolant of T = (1 - z) - 0.01
    cos(pi x) sin(pi z) and harmonic averaging (density too), adiabatic start pressure 1e5 (1 - z); tolerance 1e-7, 200 cheap
    steps."""
    pi = 3.1415926536

    def temperature(x):
        return (1.0 - x[:, 1]) - 0.01 * np.cos(1 * pi * x[:, 0] / 1.0) * np.sin(pi * x[:, 1])

    def force_q(level):
        rho = 1.0 - level.nodal_to_qpoints(temperature(level.node_coords_v()))
        rho = np.repeat((rho.shape[1] / (1.0 / rho).sum(axis=1))[:, None], rho.shape[1], axis=1)
        f = np.zeros(rho.shape + (2,))
        f[:, :, 1] = -1e5 * rho
        return f

    return StokesProblem(
        name=f"mesh_deform_topo_r{refinement}", dim=2, refinement=refinement,
        bc=[bm.BC_ZERO, bm.BC_ZERO, bm.BC_TANGENTIAL, bm.BC_TANGENTIAL],
        eta=lambda x: np.ones(x.shape[0]), force=None, force_q=force_q, averaging=bm.AVG_HARMONIC,
        pressure_scaling=1.0 / 0.01, extents=[1.0, 1.0], tol=1e-7, restart=100, n_cheap=200, n_expensive=1000,
        initial_pressure=lambda x: 1e5 * (1.0 - x[:, 1]))


def convection_box_phase_function(refinement=3):
    """tests/convection_box_phase_function.prm, first Stokes solve: box 1350 km square, free slip, eta = 1e20, gravity 10,
    `latent heat` material: rho = (1000 + 200 X) (1 - 1e-4 (T - 500)) with the phase function X = (1 + tanh((depth - 675 km
    + 405 m/K (T - 500 K)) / 54 km)) / 2 (material_model/latent_heat.cc:84-175, utilities.cc:1424-1479: transition by depth,
    Clapeyron slope -4.05e6 Pa/K over the adiabatic pressure gradient 1e4 Pa/m), T the Q2 interpolant of the deck's
    error-function profile; only the viscosity is averaged.  Start pressure 1e4 Pa/m x depth."""
    from scipy.special import erfc
    H, pi = 1350000.0, 3.1416

    def temperature(x):
        X, z = x[:, 0], x[:, 1]
        return (500 + 500 * (erfc(z / (H * 0.1)) - erfc((1 - z / H) / 0.1)) + 10 * np.cos(pi * X / H) * np.sin(pi * z / H)
                + 10 * np.cos(2 * pi * X / H) * np.sin(pi * z / H) + 10 * np.cos(pi * X / H) * np.sin(2 * pi * z / H)
                + 10 * np.cos(2 * pi * X / H) * np.sin(2 * pi * z / H))

    def force_q(level):
        T = level.nodal_to_qpoints(temperature(level.node_coords_v()))
        depth = np.clip(H - level.qpoint_coords()[:, :, 1], 0.0, H)
        dev = depth - 675000.0 - (-4050000.0 / 1e4) * (T - 500.0)
        phase = 0.5 * (1.0 + np.tanh(dev / 54000.0))
        rho = (1000.0 + 200.0 * phase) * (1.0 - 1e-4 * (T - 500.0))
        f = np.zeros(rho.shape + (2,))
        f[:, :, 1] = -10.0 * rho
        return f

    return StokesProblem(
        name=f"convection_box_phase_function_r{refinement}", dim=2, refinement=refinement, bc=[bm.BC_TANGENTIAL] * 4,
        eta=lambda x: np.full(x.shape[0], 1e20), force=None, force_q=force_q, averaging=bm.AVG_HARMONIC,
        pressure_scaling=1e20 / (0.01 * H), extents=[H, H], tol=1e-7, restart=100, n_cheap=1000, n_expensive=1000,
        initial_pressure=lambda x: 1e4 * np.clip(H - x[:, 1], 0.0, H))


NSINKER_ALL_CENTERS = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "nsinker_centers.npy"))


def nsinker_shell(refinement, n_sinkers=5, ratio=1e2, jump=100.0, averaging="harmonic average",
                  tol=1e-6, restart=200, n_cheap=2000, n_expensive=0):
    """benchmarks/nsinker_spherical_shell/gmg.prm + nsinker.cc (BASELINE configs[3]): shell R in
    [0.54, 1], zero velocity on both spheres, radial gravity of magnitude 1, the 75 sinker centres of
    the box benchmark mapped c -> 2c - 1 with those outside the shell removed (order kept,
    nsinker.cc:285-296), delta = 400, omega = 0.2 (:204-205), viscosity x `jump` below r = 0.89583
    (:62-63); `Material averaging = harmonic average` averages the density too."""
    from . import shellmesh as sm
    sr = np.sqrt(ratio)
    c = 2.0 * NSINKER_ALL_CENTERS - 1.0
    nrm = np.linalg.norm(c, axis=1)
    centers = c[(nrm >= 0.54) & (nrm <= 1.0)][:n_sinkers]
    assert centers.shape[0] == n_sinkers

    def chi(x):
        out = np.ones(x.shape[0])
        for s_ in range(n_sinkers):
            d = np.sqrt(((x - centers[s_][None, :]) ** 2).sum(axis=1))
            t = np.maximum(0.0, d - 0.2 / 2)
            out *= 1.0 - np.exp(-400.0 * t * t)
        return out

    def eta(x):
        e = (sr - 1.0 / sr) * (1.0 - chi(x)) + 1.0 / sr
        return np.where(np.linalg.norm(x, axis=1) < 0.89583, e * jump, e)

    def density(x):
        return 10.0 * (1.0 - chi(x))

    def force_q(level):
        xq = level.qpoint_coords()
        flat = xq.reshape(-1, 3)
        rho = density(flat).reshape(level.n_cells, -1)
        if averaging == "harmonic average":       # density averaged too; 0 if any value is exactly 0
            with np.errstate(divide="ignore"):
                avg = rho.shape[1] / np.sum(1.0 / rho, axis=1)
            avg[np.any(rho == 0.0, axis=1)] = 0.0
            rho = np.broadcast_to(avg[:, None], rho.shape)
        rhat = flat / np.linalg.norm(flat, axis=1, keepdims=True)
        return (-1.0 * rho.reshape(-1)[:, None] * rhat).reshape(xq.shape)     # gravity points inwards

    def scaling(fin):
        # exp(volume mean of log eta at the cell mid-points) / L, L = 1e4 / 2855000 * (R1 - R0)
        # (helper_functions.cc:286-345, spherical_shell.cc:665-676); mid-point rule weights = cell volumes
        le = np.log(eta(fin.cell_centers()))
        vol = fin.jxw.sum(axis=1)
        return float(np.exp((le * vol).sum() / vol.sum()) / (1e4 / (6336000.0 - 3481000.0) * (1.0 - 0.54)))

    return StokesProblem(
        name=f"nsinker_shell_r{refinement}", dim=3, refinement=refinement, bc=None, eta=eta, force=None,
        force_q=force_q, averaging=bm.AVG_HARMONIC, pressure_scaling=1.0, tol=tol, restart=restart,
        n_cheap=n_cheap, n_expensive=n_expensive,
        make_hierarchy=lambda r, only_finest: sm.ShellHierarchy(r, 0.54, 1.0, zero_velocity=True, only_finest=only_finest),
        pressure_scaling_from_mesh=scaling)


def _hollow_sphere_constants(m):
    R1, R2, gam = 0.5, 1.0, 1.0
    if m == -1:
        alpha = -gam * (R2 ** 3 - R1 ** 3) / (R2 ** 3 * np.log(R1) - R1 ** 3 * np.log(R2))
        beta = -3 * gam * (np.log(R2) - np.log(R1)) / (R1 ** 3 * np.log(R2) - R2 ** 3 * np.log(R1))
    else:
        alpha = gam * (m + 1) * (R1 ** -3 - R2 ** -3) / (R1 ** (-m - 4) - R2 ** (-m - 4))
        beta = -3 * gam * (R1 ** (m + 1) - R2 ** (m + 1)) / (R1 ** (m + 4) - R2 ** (m + 4))
    return R1, R2, gam, alpha, beta


def hollow_sphere_solution(x, m=-1):
    """Analytic velocity [n][3] and pressure [n] of the hollow-sphere benchmark (Thieulot, Solid Earth
    2017, viscosity mu = r^(m+1)) as benchmarks/hollow_sphere/hollow_sphere.cc:47-128 evaluates it:
    gamma = 1, R1 = 0.5, R2 = 1, g = 1, rho_0 = 1000."""
    R1, R2, gam, alpha, beta = _hollow_sphere_constants(m)
    r = np.linalg.norm(x, axis=1)
    theta = np.arccos(np.clip(x[:, 2] / r, -1.0, 1.0))
    phi = np.arctan2(x[:, 1], x[:, 0])
    if m == -1:
        fr = alpha / r ** 2 + beta * r
        gr = -2.0 / r ** 2 * (alpha * np.log(r) + beta / 3.0 * r ** 3 + gam)
        hr = 2.0 / r * gr
    else:
        fr = alpha / r ** (m + 3) + beta * r
        gr = -2.0 / r ** 2 * (-alpha / (m + 1) * r ** (-m - 1) + beta / 3.0 * r ** 3 + gam)
        hr = (m + 3) / r * gr * r ** (m + 1)
    vr, vt, vp = gr * np.cos(theta), fr * np.sin(theta), fr * np.sin(theta)
    st, ct, sp, cp = np.sin(theta), np.cos(theta), np.sin(phi), np.cos(phi)
    v = np.stack([st * cp * vr + ct * cp * vt - sp * vp, st * sp * vr + ct * sp * vt + cp * vp, ct * vr - st * vt], axis=1)
    p = hr * np.cos(theta) + 1000.0 * (R2 - r)
    return v, p


def hollow_sphere(refinement=1, tol=1e-8, n_cheap=200):
    """tests/hollow_sphere_gmg.prm: shell R in [0.5, 1] on the 12-cell coarse mesh, prescribed
    (analytic) velocity on both spheres, mu = 1, the benchmark's density (hollow_sphere.cc:352-375,
    m = -1) under `Material averaging = harmonic average` (density averaged too), radial gravity 1,
    tolerance 1e-8, 200 cheap steps; first linearisation point = the adiabatic pressure rho_0 g (R2 - r)."""
    from . import shellmesh as sm
    R1, R2, gam = 0.5, 1.0, 1.0
    alpha = -gam * (R2 ** 3 - R1 ** 3) / (R2 ** 3 * np.log(R1) - R1 ** 3 * np.log(R2))
    beta = -3 * gam * (np.log(R2) - np.log(R1)) / (R1 ** 3 * np.log(R2) - R2 ** 3 * np.log(R1))

    def density(x):
        r = np.linalg.norm(x, axis=1)
        ct = x[:, 2] / r
        return -(alpha / r ** 4 * (8 * np.log(r) - 6) + 8.0 / 3.0 * beta / r + 8 * gam / r ** 4) * ct + 1000.0

    def force_q(level):
        xq = level.qpoint_coords()
        flat = xq.reshape(-1, 3)
        rho = density(flat).reshape(level.n_cells, -1)
        avg = rho.shape[1] / np.sum(1.0 / rho, axis=1)
        rhat = flat / np.linalg.norm(flat, axis=1, keepdims=True)
        return (-np.repeat(avg, rho.shape[1])[:, None] * rhat).reshape(xq.shape)

    return StokesProblem(
        name=f"hollow_sphere_r{refinement}", dim=3, refinement=refinement, bc=None,
        eta=lambda x: np.ones(x.shape[0]), force=None, force_q=force_q, averaging=bm.AVG_HARMONIC,
        pressure_scaling=1.0 / (1e4 / (6336000.0 - 3481000.0) * (R2 - R1)), tol=tol, n_cheap=n_cheap, n_expensive=1000,
        initial_pressure=lambda x: 1000.0 * (R2 - np.linalg.norm(x, axis=1)),
        boundary_velocity=lambda x: hollow_sphere_solution(x)[0],
        make_hierarchy=lambda r, only_finest: sm.ShellHierarchy(r, R1, R2, zero_velocity=True, only_finest=only_finest, n_coarse=12))


def hollow_sphere_project_q1(refinement=1):
    """tests/hollow_sphere_gmg_project_q1.prm: the hollow-sphere benchmark with its default viscosity parameter m = 3
    (mu = r^4, density hollow_sphere.cc:386-390) under `Material averaging = project to Q1 only viscosity` (the density stays
    at the quadrature points), tolerance 1e-8, 100 cheap steps, on the mesh of `hollow_sphere`."""
    from . import shellmesh as sm
    m = 3
    R1, R2, gam, alpha, beta = _hollow_sphere_constants(m)

    def density(x):
        r = np.linalg.norm(x, axis=1)
        ct = x[:, 2] / r
        return -(2 * alpha * r ** -4 * (m + 3) / (m + 1) * (m - 1) - 2 * beta / 3 * (m - 1) * (m + 3) * r ** m
                 - m * (m + 5) * 2 * gam * r ** (m - 3)) * ct + 1000.0

    def force_q(level):
        xq = level.qpoint_coords()
        flat = xq.reshape(-1, 3)
        rhat = flat / np.linalg.norm(flat, axis=1, keepdims=True)
        return (-density(flat)[:, None] * rhat).reshape(xq.shape)

    def scaling(level):      # exp(mean log eta at the cell midpoints) / length scale (helper_functions.cc:252-320)
        xm = level.qpoint_coords()[:, 13, :]
        w = level.jxw[:, 13]
        eta_ref = np.exp((np.log(np.linalg.norm(xm, axis=1) ** (m + 1)) * w).sum() / w.sum())
        return float(eta_ref) / (1e4 / (6336000.0 - 3481000.0) * (R2 - R1))

    return StokesProblem(
        name=f"hollow_sphere_project_q1_r{refinement}", dim=3, refinement=refinement, bc=None,
        eta=lambda x: np.linalg.norm(x, axis=1) ** (m + 1), force=None, force_q=force_q, averaging=bm.AVG_PROJECT_Q1,
        pressure_scaling=1.0, pressure_scaling_from_mesh=scaling, tol=1e-8, n_cheap=100, n_expensive=1000,
        initial_pressure=lambda x: 1000.0 * (R2 - np.linalg.norm(x, axis=1)),
        boundary_velocity=lambda x: hollow_sphere_solution(x, m)[0],
        make_hierarchy=lambda r, only_finest: sm.ShellHierarchy(r, R1, R2, zero_velocity=True, only_finest=only_finest, n_coarse=12))
