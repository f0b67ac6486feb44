"""ctypes binding of the C-ABI declared in include/agmg.h.

``Library(path, prefix)`` binds one shared object exporting ``<prefix>create``,
``<prefix>vmult`` ... with the signatures of agmg.h.  The product library is
``aspect_amd/csrc/libaspect_amd.so`` with prefix ``agmg_``; it is loaded by
``load()`` and nothing else is ever substituted for it: a missing library is
an ImportError, not a fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ACTIVE = -1
OP_STOKES, OP_A, OP_BT, OP_MASS = 0, 1, 2, 3
GEOM_CARTESIAN, GEOM_GENERAL = 0, 1
KRYLOV_GMRES, KRYLOV_IDR = 0, 1

c_dp = C.POINTER(C.c_double)
c_u32p = C.POINTER(C.c_uint32)
c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)


class Options(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("degree_v", C.c_int32), ("device", C.c_int32), ("n_levels", C.c_int32),
        ("linear_tolerance", C.c_double),
        ("n_cheap_steps", C.c_int32), ("n_expensive_steps", C.c_int32),
        ("gmres_restart", C.c_int32), ("krylov_type", C.c_int32), ("idr_s", C.c_int32),
        ("A_block_tolerance", C.c_double), ("S_block_tolerance", C.c_double),
        ("A_block_is_symmetric", C.c_int32), ("coarse_on_host", C.c_int32),
        ("smoother_degree", C.c_int32), ("smoothing_range", C.c_double),
        ("eig_cg_n_iterations", C.c_int32), ("coarse_degree", C.c_int32),
        ("coarse_range", C.c_double), ("coarse_eig_cg_n_iterations", C.c_int32),
        ("schur_smoother_steps", C.c_int32),
        ("world_size", C.c_int32), ("rank", C.c_int32),
        ("nccl_id", C.c_uint8 * 128),
    ]


class LevelDesc(C.Structure):
    _fields_ = [
        ("n_cells", C.c_int32), ("n_nodes_v", C.c_int32), ("n_nodes_p", C.c_int32),
        ("n_owned_nodes_v", C.c_int32), ("n_owned_nodes_p", C.c_int32),
        ("cell_nodes_v", c_u32p), ("cell_nodes_p", c_u32p),
        ("constrained_v", c_u8p), ("constrained_p", c_u8p),
        ("geometry_kind", C.c_int32),
        ("cell_h", c_dp), ("inv_jacobian", c_dp), ("jxw", c_dp),
        ("parent_cell", c_i32p), ("child_index", c_u8p),
        ("n_neighbors", C.c_int32), ("neighbor_rank", c_i32p),
        ("send_offsets_v", c_i32p), ("send_nodes_v", c_u32p), ("recv_offsets_v", c_i32p),
        ("send_offsets_p", c_i32p), ("send_nodes_p", c_u32p), ("recv_offsets_p", c_i32p),
        ("global_nodes_v", c_u32p), ("global_nodes_p", c_u32p),
        ("n_hanging_v", C.c_int32), ("n_hanging_p", C.c_int32),
        ("hanging_nodes_v", c_u32p), ("hanging_ptr_v", c_i32p), ("hanging_master_v", c_u32p), ("hanging_weight_v", c_dp),
        ("hanging_nodes_p", c_u32p), ("hanging_ptr_p", c_i32p), ("hanging_master_p", c_u32p), ("hanging_weight_p", c_dp),
        ("refinement_edge_v", c_u8p), ("refinement_edge_p", c_u8p),
        ("n_copy_v", C.c_int32), ("n_copy_p", C.c_int32),
        ("copy_active_v", c_u32p), ("copy_level_v", c_u32p), ("copy_active_p", c_u32p), ("copy_level_p", c_u32p),
        ("replicated", C.c_int32),
        ("n_sym_neighbors_p", C.c_int32), ("sym_neighbor_rank_p", c_i32p), ("sym_offsets_p", c_i32p), ("sym_nodes_p", c_u32p),
        ("n_sym_neighbors_v", C.c_int32), ("sym_neighbor_rank_v", c_i32p), ("sym_offsets_v", c_i32p), ("sym_nodes_v", c_u32p),
    ]


class CellData(C.Structure):
    _fields_ = [
        ("is_compressible", C.c_int32), ("enable_newton_derivatives", C.c_int32),
        ("enable_prescribed_dilation", C.c_int32), ("symmetrize_newton_system", C.c_int32),
        ("pressure_scaling", C.c_double),
        ("n_q_viscosity", C.c_int32), ("viscosity", c_dp),
        ("strain_rate_table", c_dp), ("newton_factor_wrt_strain_rate", c_dp),
        ("newton_factor_wrt_pressure", c_dp),
        ("dilation_lhs_term", c_dp), ("dilation_derivative_wrt_strain_rate", c_dp),
        ("dilation_derivative_wrt_pressure", c_dp),
    ]


class FaceData(C.Structure):
    _fields_ = [
        ("n_faces", C.c_int32), ("face_cell", c_u32p), ("face_number", c_u8p),
        ("stabilization_term", c_dp), ("normal", c_dp), ("jxw", c_dp),
    ]


class SolveReport(C.Structure):
    _fields_ = [
        ("n_cheap_iterations", C.c_int32), ("n_expensive_iterations", C.c_int32),
        ("n_A_preconditioner_cheap", C.c_int32), ("n_A_preconditioner_expensive", C.c_int32),
        ("n_S_preconditioner_cheap", C.c_int32), ("n_S_preconditioner_expensive", C.c_int32),
        ("initial_nonlinear_residual", C.c_double), ("final_linear_residual", C.c_double),
        ("solver_tolerance", C.c_double), ("solve_seconds", C.c_double), ("setup_seconds", C.c_double),
        ("n_history", C.c_int32), ("history", C.c_double * 2048),
    ]


# every symbol include/agmg.h declares (suffix after the prefix)
SYMBOLS = [
    "default_options", "last_error", "create", "destroy", "set_communicator",
    "rccl_get_unique_id", "set_communicator_rccl", "set_level", "set_coefficients",
    "set_free_surface_faces",
    "compute_diagonals", "rhs_correction", "vmult", "vmult_device", "vmult_device_noalloc", "get_inverse_diagonal", "prolongate_add",
    "restrict_add", "setup_smoothers", "get_eigenvalue_estimates", "vcycle", "precondition",
    "solve", "bench_vmult", "execute_solver_timings", "get_level_kernel_path", "get_overlap_statistics", "get_cell_data_memory_consumption",
    "get_mg_transfer_memory_consumption", "get_constraint_memory_consumption",
    "get_dof_handler_memory_consumption",
]


def _ptr(a, ctype):
    if a is None:
        return C.cast(None, C.POINTER(ctype))
    return a.ctypes.data_as(C.POINTER(ctype))


class AgmgError(RuntimeError):
    pass


class Library:
    def __init__(self, path, prefix="agmg_", optional=()):
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: build it first (python -c 'import __graft_entry__ as g; g.build()')")
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        L, p = self.lib, prefix

        def fn(name, restype, argtypes):
            try:
                f = getattr(L, p + name)
            except AttributeError:
                if name in optional:
                    return None
                raise
            f.restype = restype
            f.argtypes = argtypes
            return f

        vp = C.c_void_p
        self.default_options = fn("default_options", None, [C.POINTER(Options)])
        self.last_error = fn("last_error", C.c_char_p, [])
        self.create = fn("create", vp, [C.POINTER(Options)])
        self.destroy = fn("destroy", None, [vp])
        self.set_communicator = fn("set_communicator", C.c_int, [vp, vp])
        self.rccl_get_unique_id = fn("rccl_get_unique_id", C.c_int, [C.POINTER(C.c_uint8)])
        self.set_communicator_rccl = fn("set_communicator_rccl", C.c_int, [vp, C.POINTER(C.c_uint8)])
        self.set_level = fn("set_level", C.c_int, [vp, C.c_int, C.POINTER(LevelDesc)])
        self.set_coefficients = fn("set_coefficients", C.c_int, [vp, C.c_int, C.POINTER(CellData)])
        self.set_free_surface_faces = fn("set_free_surface_faces", C.c_int, [vp, C.POINTER(FaceData)])
        self.compute_diagonals = fn("compute_diagonals", C.c_int, [vp])
        self.rhs_correction = fn("rhs_correction", C.c_int, [vp, c_dp, c_dp])
        self.vmult = fn("vmult", C.c_int, [vp, C.c_int, C.c_int, c_dp, c_dp])
        self.vmult_device = fn("vmult_device", C.c_int, [vp, C.c_int, C.c_int, vp, vp])
        self.vmult_device_noalloc = fn("vmult_device_noalloc", C.c_int, [vp, C.c_int, C.c_int, vp, vp])
        self.vmult_dense = fn("vmult_dense", C.c_int, [vp, C.c_int, C.c_int, c_dp, c_dp])
        # (checker library only: the CPU baseline leg of bench.py, SIMD across cells; returns the lanes used)
        self.vmult_simd = fn("vmult_simd", C.c_int, [vp, C.c_int, C.c_int, c_dp, c_dp]) if "vmult_simd" in optional or prefix != "agmg_" else None
        self.bench_vmult_simd = (fn("bench_vmult_simd", C.c_int, [vp, C.c_int, C.c_int, C.c_int, c_dp, c_dp])
                                 if prefix != "agmg_" else None)
        self.get_inverse_diagonal = fn("get_inverse_diagonal", C.c_int, [vp, C.c_int, C.c_int, c_dp])
        self.prolongate_add = fn("prolongate_add", C.c_int, [vp, C.c_int, C.c_int, c_dp, c_dp])
        self.restrict_add = fn("restrict_add", C.c_int, [vp, C.c_int, C.c_int, c_dp, c_dp])
        self.setup_smoothers = fn("setup_smoothers", C.c_int, [vp])
        self.get_eigenvalue_estimates = fn("get_eigenvalue_estimates", C.c_int,
                                           [vp, C.c_int, C.c_int, c_dp, c_dp])
        self.vcycle = fn("vcycle", C.c_int, [vp, C.c_int, c_dp, c_dp])
        self.precondition = fn("precondition", C.c_int, [vp, C.c_int, c_dp, c_dp])
        self.solve = fn("solve", C.c_int, [vp, c_dp, c_dp, C.c_int, C.POINTER(SolveReport)])
        self.bench_vmult = fn("bench_vmult", C.c_int, [vp, C.c_int, C.c_int, C.c_int, c_dp])
        self.execute_solver_timings = fn("execute_solver_timings", C.c_int, [vp, C.c_int, c_dp])
        self.get_overlap_statistics = fn("get_overlap_statistics", C.c_int, [vp, C.c_int, c_i32p, c_i32p, c_i32p])
        self.get_level_kernel_path = fn("get_level_kernel_path", C.c_int, [vp, C.c_int, c_i32p, c_i32p, c_i32p])
        for g in ("cell_data", "mg_transfer", "constraint", "dof_handler"):
            setattr(self, f"get_{g}_memory_consumption",
                    fn(f"get_{g}_memory_consumption", C.c_size_t, [vp]))

    def has(self, name):
        return hasattr(self.lib, self.prefix + name)


_PRODUCT = None


def product_library_path():
    # AGMG_LIB: developer knob for kernel ablation builds (tools/ablate.sh); still a HIP build of
    # the same sources, never a CPU substitute
    override = os.environ.get("AGMG_LIB")
    if override:
        return override
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libaspect_amd.so")


def load():
    """Load the product HIP library; raises ImportError when it is missing."""
    global _PRODUCT
    if _PRODUCT is None:
        _PRODUCT = Library(product_library_path(), "agmg_", optional=("vmult_dense", "vmult_simd"))
    return _PRODUCT


class Solver:
    """One context of a Library fed from a cases.StokesProblem (host-side
    equivalent of StokesMatrixFreeHandler's setup_dofs/assemble/
    build_preconditioner/solve sequence)."""

    def __init__(self, lib: Library, problem, single_level=False, transport=None, **opt_overrides):
        self.lib = lib
        self.problem = problem
        hier = problem.hier
        opt = Options()
        lib.default_options(C.byref(opt))
        opt.dim = hier.dim
        opt.n_levels = hier.n_levels
        opt.linear_tolerance = problem.tol
        opt.n_cheap_steps = problem.n_cheap
        opt.n_expensive_steps = problem.n_expensive
        opt.gmres_restart = problem.restart
        opt.krylov_type = KRYLOV_IDR if problem.krylov == "idr" else KRYLOV_GMRES
        opt.idr_s = problem.idr_s
        if transport is not None:       # distributed: `problem` is this rank's partition.LocalProblem
            opt.world_size, opt.rank = transport.struct.size, transport.struct.rank
        for k, v in opt_overrides.items():
            setattr(opt, k, v)
        self.opt = opt
        self.ctx = lib.create(C.byref(opt))
        if not self.ctx:
            raise AgmgError(lib.last_error().decode())
        self._keep = []
        self.transport = transport
        if transport is not None:
            if getattr(transport, "native_rccl", False):   # parallel.RcclTransport: in-library RCCL
                ident = (C.c_uint8 * 128)(*transport.unique_id(lib))
                self._check(lib.set_communicator_rccl(self.ctx, ident))
            else:
                self._check(lib.set_communicator(self.ctx, C.addressof(transport.struct)))
        for l, lev in enumerate(hier.levels):
            self._set_level(l, lev)
        if getattr(hier, "active", None) is not None:       # adaptive refinement: the active mesh has tables of its own
            self._set_level(ACTIVE, hier.active)
        else:
            self._check(lib.set_level(self.ctx, ACTIVE, None))
        comp = int(bool(getattr(problem, "is_compressible", False)))   # material_model.is_compressible()
        for l, lev in enumerate(hier.levels):
            self._set_coeff(l, problem.eta_levels[l], is_compressible=comp)
        self._set_coeff(ACTIVE, problem.eta_active, is_compressible=comp, dilation=getattr(problem, "dilation", None))
        if getattr(problem, "free_surface", None) is not None:
            self.set_free_surface_faces(*problem.free_surface)
        if not single_level:
            self._check(lib.compute_diagonals(self.ctx))
        fin = hier.finest
        self.n_u, self.n_p = fin.n_dofs_v, fin.n_dofs_p

    def _check(self, rc, allow=(0,)):
        if rc not in allow:
            raise AgmgError(f"status {rc}: {self.lib.last_error().decode()}")
        return rc

    def _set_level(self, l, lev):
        d = LevelDesc()
        cnv = np.ascontiguousarray(lev.cell_nodes_v, dtype=np.uint32)
        cnp = np.ascontiguousarray(lev.cell_nodes_p, dtype=np.uint32)
        cmv = np.ascontiguousarray(lev.cmask_v, dtype=np.uint8)
        cmp_ = np.ascontiguousarray(lev.cmask_p, dtype=np.uint8)
        ch = np.ascontiguousarray(lev.cell_h, dtype=np.float64)
        d.n_cells = lev.n_cells
        d.n_nodes_v = d.n_owned_nodes_v = lev.n_nodes_v
        d.n_nodes_p = d.n_owned_nodes_p = lev.n_nodes_p
        d.cell_nodes_v = _ptr(cnv, C.c_uint32)
        d.cell_nodes_p = _ptr(cnp, C.c_uint32)
        d.constrained_v = _ptr(cmv, C.c_uint8)
        d.constrained_p = _ptr(cmp_, C.c_uint8)
        keep = [cnv, cnp, cmv, cmp_, ch]
        if getattr(lev, "general", False):
            ij = np.ascontiguousarray(lev.inv_jacobian, dtype=np.float64)
            jw = np.ascontiguousarray(lev.jxw, dtype=np.float64)
            d.geometry_kind = GEOM_GENERAL
            d.inv_jacobian = _ptr(ij, C.c_double)
            d.jxw = _ptr(jw, C.c_double)
            keep += [ij, jw]
        else:
            d.geometry_kind = GEOM_CARTESIAN
            d.cell_h = _ptr(ch, C.c_double)
        # a level every rank holds whole (distributed contexts; agmg_level_desc::replicated): no exchange plan
        d.replicated = int(bool(getattr(lev, "replicated", False)) and self.transport is not None)
        if getattr(lev, "neighbors", None) is not None and self.transport is not None and not d.replicated:
            d.n_owned_nodes_v, d.n_owned_nodes_p = lev.n_owned_nodes_v, lev.n_owned_nodes_p
            nb = np.ascontiguousarray(lev.neighbors, dtype=np.int32)
            plan = [np.ascontiguousarray(a) for a in (lev.send_offsets_v, lev.send_nodes_v, lev.recv_offsets_v,
                                                      lev.send_offsets_p, lev.send_nodes_p, lev.recv_offsets_p,
                                                      lev.global_nodes_v, lev.global_nodes_p)]
            d.n_neighbors = nb.size
            d.neighbor_rank = _ptr(nb, C.c_int32)
            d.send_offsets_v, d.send_nodes_v, d.recv_offsets_v = _ptr(plan[0], C.c_int32), _ptr(plan[1], C.c_uint32), _ptr(plan[2], C.c_int32)
            d.send_offsets_p, d.send_nodes_p, d.recv_offsets_p = _ptr(plan[3], C.c_int32), _ptr(plan[4], C.c_uint32), _ptr(plan[5], C.c_int32)
            d.global_nodes_v, d.global_nodes_p = _ptr(plan[6], C.c_uint32), _ptr(plan[7], C.c_uint32)
            keep += [nb] + plan
            if getattr(lev, "sym_neighbors_p", None) is not None and len(lev.sym_neighbors_p):
                sym = [np.ascontiguousarray(lev.sym_neighbors_p, dtype=np.int32), np.ascontiguousarray(lev.sym_offsets_p, dtype=np.int32),
                       np.ascontiguousarray(lev.sym_nodes_p, dtype=np.uint32)]
                d.n_sym_neighbors_p = sym[0].size
                d.sym_neighbor_rank_p, d.sym_offsets_p, d.sym_nodes_p = _ptr(sym[0], C.c_int32), _ptr(sym[1], C.c_int32), _ptr(sym[2], C.c_uint32)
                keep += sym
            if getattr(lev, "sym_neighbors_v", None) is not None and len(lev.sym_neighbors_v):
                sym = [np.ascontiguousarray(lev.sym_neighbors_v, dtype=np.int32), np.ascontiguousarray(lev.sym_offsets_v, dtype=np.int32),
                       np.ascontiguousarray(lev.sym_nodes_v, dtype=np.uint32)]
                d.n_sym_neighbors_v = sym[0].size
                d.sym_neighbor_rank_v, d.sym_offsets_v, d.sym_nodes_v = _ptr(sym[0], C.c_int32), _ptr(sym[1], C.c_int32), _ptr(sym[2], C.c_uint32)
                keep += sym
        if getattr(lev, "hanging_v", None) is not None:
            # hanging-node constraints of an adaptive mesh: (nodes, ptr, masters, weights) per space
            for name, (hn, hp, hm, hw) in (("v", lev.hanging_v), ("p", lev.hanging_p)):
                arrs = [np.ascontiguousarray(hn, dtype=np.uint32), np.ascontiguousarray(hp, dtype=np.int32),
                        np.ascontiguousarray(hm, dtype=np.uint32), np.ascontiguousarray(hw, dtype=np.float64)]
                setattr(d, "n_hanging_" + name, arrs[0].size)
                setattr(d, "hanging_nodes_" + name, _ptr(arrs[0], C.c_uint32))
                setattr(d, "hanging_ptr_" + name, _ptr(arrs[1], C.c_int32))
                setattr(d, "hanging_master_" + name, _ptr(arrs[2], C.c_uint32))
                setattr(d, "hanging_weight_" + name, _ptr(arrs[3], C.c_double))
                keep += arrs
        if getattr(lev, "edge_v", None) is not None:       # a level of local smoothing under an adaptive mesh
            arrs = [np.ascontiguousarray(lev.edge_v, dtype=np.uint8), np.ascontiguousarray(lev.edge_p, dtype=np.uint8)]
            d.refinement_edge_v, d.refinement_edge_p = _ptr(arrs[0], C.c_uint8), _ptr(arrs[1], C.c_uint8)
            for name in ("v", "p"):
                ca = np.ascontiguousarray(getattr(lev, "copy_active_" + name), dtype=np.uint32)
                cl = np.ascontiguousarray(getattr(lev, "copy_level_" + name), dtype=np.uint32)
                setattr(d, "n_copy_" + name, ca.size)
                setattr(d, "copy_active_" + name, _ptr(ca, C.c_uint32))
                setattr(d, "copy_level_" + name, _ptr(cl, C.c_uint32))
                arrs += [ca, cl]
            keep += arrs
        if getattr(lev, "parent_cell", None) is not None:
            par = np.ascontiguousarray(lev.parent_cell, dtype=np.int32)
            chi = np.ascontiguousarray(lev.child_index, dtype=np.uint8)
            d.parent_cell = _ptr(par, C.c_int32)
            d.child_index = _ptr(chi, C.c_uint8)
            keep += [par, chi]
        elif l > 0 and lev.level > 0 and not hasattr(lev, "edge_v"):
            nch = 1 << lev.dim
            par = (np.arange(lev.n_cells, dtype=np.int32) // nch).astype(np.int32)
            chi = (np.arange(lev.n_cells, dtype=np.int32) % nch).astype(np.uint8)
            d.parent_cell = _ptr(par, C.c_int32)
            d.child_index = _ptr(chi, C.c_uint8)
            keep += [par, chi]
        self._check(self.lib.set_level(self.ctx, l, C.byref(d)))
        self._keep.append(keep)

    def _set_coeff(self, l, table, is_compressible=0, newton=None, symmetrize=0, dilation=None):
        """newton = (strain_rate [cells][nq][ns], newton_factor_wrt_strain_rate [cells][nq][ns],
        newton_factor_wrt_pressure [cells][nq]): enables the Newton-derivative terms (active level)."""
        t = np.ascontiguousarray(table, dtype=np.float64)
        cd = CellData()
        cd.is_compressible = is_compressible
        if l == ACTIVE:   # (operators.Tvmult: the Stokes operator stops being symmetric)
            self.unsymmetric_newton = newton is not None and not symmetrize
        if newton is not None:
            nt = [np.ascontiguousarray(a, dtype=np.float64) for a in newton]
            cd.enable_newton_derivatives = 1
            cd.symmetrize_newton_system = int(symmetrize)
            cd.strain_rate_table = _ptr(nt[0], C.c_double)
            cd.newton_factor_wrt_strain_rate = _ptr(nt[1], C.c_double)
            cd.newton_factor_wrt_pressure = _ptr(nt[2], C.c_double)
        if dilation is not None:   # (lhs term [cells][nq], d/d strain rate [cells][nq][ns] | None, d/dp [cells][nq] | None)
            dt = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in dilation]
            cd.enable_prescribed_dilation = 1
            cd.dilation_lhs_term = _ptr(dt[0], C.c_double)
            if len(dt) > 1 and dt[1] is not None:
                cd.dilation_derivative_wrt_strain_rate = _ptr(dt[1], C.c_double)
                cd.dilation_derivative_wrt_pressure = _ptr(dt[2], C.c_double)
        cd.pressure_scaling = self.problem.pressure_scaling
        cd.n_q_viscosity = t.shape[1]
        cd.viscosity = _ptr(t, C.c_double)
        self._check(self.lib.set_coefficients(self.ctx, l, C.byref(cd)))

    def set_free_surface_faces(self, face_cell=None, face_number=None, stabilization_term=None, normal=None, jxw=None):
        """Free-surface stabilisation faces of the active Stokes operator (no arguments: remove them)."""
        if face_cell is None:
            return self._check(self.lib.set_free_surface_faces(self.ctx, None))
        fd = FaceData()
        fc = np.ascontiguousarray(face_cell, dtype=np.uint32)
        fn = np.ascontiguousarray(face_number, dtype=np.uint8)
        st = np.ascontiguousarray(stabilization_term, dtype=np.float64)
        nqf = 3 ** (self.problem.hier.dim - 1)
        assert fn.size == fc.size and st.shape == (fc.size, nqf, self.problem.hier.dim)
        fd.n_faces = fc.size
        fd.face_cell, fd.face_number, fd.stabilization_term = _ptr(fc, C.c_uint32), _ptr(fn, C.c_uint8), _ptr(st, C.c_double)
        if normal is not None:
            nr = np.ascontiguousarray(normal, dtype=np.float64)
            jw = np.ascontiguousarray(jxw, dtype=np.float64)
            assert nr.shape == st.shape and jw.shape == (fc.size, nqf)
            fd.normal, fd.jxw = _ptr(nr, C.c_double), _ptr(jw, C.c_double)
        self._check(self.lib.set_free_surface_faces(self.ctx, C.byref(fd)))

    # -- operator interface ---------------------------------------------
    def size(self, which, level=ACTIVE):
        lev = self.problem.hier.finest if level == ACTIVE else self.problem.hier.levels[level]
        if which == OP_A:
            return lev.n_dofs_v
        if which == OP_MASS:
            return lev.n_dofs_p
        return lev.n_dofs_v + lev.n_dofs_p

    def vmult(self, which, src, level=ACTIVE, dense=False):
        src = np.ascontiguousarray(src, dtype=np.float64)
        assert src.size == self.size(which, level)
        dst = np.zeros_like(src)
        f = self.lib.vmult_dense if dense else self.lib.vmult
        self._check(f(self.ctx, which, level, _ptr(dst, C.c_double), _ptr(src, C.c_double)))
        return dst

    def vmult_simd(self, which, src, level=ACTIVE):
        """Checker library only (oracle/cpu_simd.cpp): the operator with SIMD across cells; returns (dst, lanes)."""
        src = np.ascontiguousarray(src, dtype=np.float64)
        dst = np.zeros_like(src)
        lanes = self.lib.vmult_simd(self.ctx, which, level, _ptr(dst, C.c_double), _ptr(src, C.c_double))
        if lanes < 0:
            raise RuntimeError("vmult_simd does not cover this level")
        return dst, lanes

    def bench_vmult_simd(self, which, src, n_repeat, level=ACTIVE):
        """Checker library only: seconds per application of vmult_simd, timed inside the library on buffers allocated
        once (the reference's harness times the vmult alone, local_smoothing.cc:1236-1264); returns (seconds, lanes)."""
        src = np.ascontiguousarray(src, dtype=np.float64)
        sec = C.c_double(0.0)
        lanes = self.lib.bench_vmult_simd(self.ctx, which, level, int(n_repeat), _ptr(src, C.c_double), C.byref(sec))
        if lanes < 0:
            raise RuntimeError("vmult_simd does not cover this level")
        return sec.value, lanes

    def rhs_correction(self, u0, rhs):
        """correct_stokes_rhs(): rhs - M u0 with u0 read without the constraints."""
        u0 = np.ascontiguousarray(u0, dtype=np.float64)
        out = np.ascontiguousarray(rhs, dtype=np.float64).copy()
        assert u0.size == out.size == self.n_u + self.n_p
        self._check(self.lib.rhs_correction(self.ctx, _ptr(u0, C.c_double), _ptr(out, C.c_double)))
        return out

    def inverse_diagonal(self, which, level):
        dst = np.zeros(self.size(which, level))
        self._check(self.lib.get_inverse_diagonal(self.ctx, which, level, _ptr(dst, C.c_double)))
        return dst

    def prolongate_add(self, which, to_level, dst, src):
        dst = np.ascontiguousarray(dst, dtype=np.float64).copy()
        src = np.ascontiguousarray(src, dtype=np.float64)
        assert dst.size == self.size(which, to_level) and src.size == self.size(which, to_level - 1)
        self._check(self.lib.prolongate_add(self.ctx, which, to_level, _ptr(dst, C.c_double), _ptr(src, C.c_double)))
        return dst

    def restrict_add(self, which, from_level, dst, src):
        dst = np.ascontiguousarray(dst, dtype=np.float64).copy()
        src = np.ascontiguousarray(src, dtype=np.float64)
        assert src.size == self.size(which, from_level) and dst.size == self.size(which, from_level - 1)
        self._check(self.lib.restrict_add(self.ctx, which, from_level, _ptr(dst, C.c_double), _ptr(src, C.c_double)))
        return dst

    def setup_smoothers(self):
        self._check(self.lib.setup_smoothers(self.ctx))

    def eigenvalue_estimates(self, which, level):
        a, b = C.c_double(), C.c_double()
        self._check(self.lib.get_eigenvalue_estimates(self.ctx, which, level, C.byref(a), C.byref(b)))
        return a.value, b.value

    def vcycle(self, which, src):
        src = np.ascontiguousarray(src, dtype=np.float64)
        dst = np.zeros_like(src)
        self._check(self.lib.vcycle(self.ctx, which, _ptr(dst, C.c_double), _ptr(src, C.c_double)))
        return dst

    def precondition(self, src, expensive=False):
        src = np.ascontiguousarray(src, dtype=np.float64)
        dst = np.zeros_like(src)
        self._check(self.lib.precondition(self.ctx, int(expensive), _ptr(dst, C.c_double), _ptr(src, C.c_double)))
        return dst

    def solve(self, rhs=None, x0=None, newton=False, allow_failure=False):
        p = self.problem
        if rhs is None:
            rhs = np.concatenate([p.rhs_u, p.rhs_p])
        if x0 is None:
            x0 = np.concatenate([np.zeros(self.n_u), p.p0])
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        x = np.ascontiguousarray(x0, dtype=np.float64).copy()
        rep = SolveReport()
        rc = self.lib.solve(self.ctx, _ptr(rhs, C.c_double), _ptr(x, C.c_double), int(newton), C.byref(rep))
        self._check(rc, allow=(0, 1) if allow_failure else (0,))
        return x, rep, rc

    def bench_vmult(self, which, level, n_repeat):
        t = C.c_double()
        self._check(self.lib.bench_vmult(self.ctx, which, level, n_repeat, C.byref(t)))
        return t.value

    def solver_timings(self, n_repeat=5):
        """Seconds per call of the 'Execute solver timings' sections: the reference's stokes_vmult,
        stokes_preconditioner, A_preconditioner, S_preconditioner (local_smoothing.cc:1268-1340), then two of this
        library's own (A_block_vmult, smoother_setup) where the reference times its two cheap solves."""
        out = (C.c_double * 6)()
        self._check(self.lib.execute_solver_timings(self.ctx, n_repeat, out))
        names = ("stokes_vmult", "stokes_preconditioner", "A_preconditioner", "S_preconditioner", "A_block_vmult", "smoother_setup")
        return dict(zip(names, list(out)))

    def overlap_statistics(self, level):
        """(interior blocks, interface blocks, overlapped operator applications so far)."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.get_overlap_statistics(self.ctx, level, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def kernel_path(self, level=ACTIVE):
        """(bricks, layers per brick, blocks) of the kernels that apply the velocity rows of a level."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.get_level_kernel_path(self.ctx, level, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def close(self):
        if self.ctx:
            self.lib.destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
