"""Host-side mirror of the reference's operator / handler interface for this path, over the
C-ABI (names, argument meaning and error behaviour follow
include/aspect/simulator/solver/matrix_free_operators.h:208-507 and stokes_matrix_free.h:46-98),
so parity tests read like the reference's own usage.  The compute is libaspect_amd.so only."""
from __future__ import annotations

import numpy as np

from . import capi


class _Operator:
    which = None

    def __init__(self, handler, level=capi.ACTIVE):
        self.handler, self.level = handler, level

    def m(self):
        return self.handler.solver.size(self.which, self.level)

    def vmult(self, dst, src):
        """dst = Op * src (deal.II MatrixFreeOperators::Base::vmult semantics)."""
        dst[:] = self.handler.solver.vmult(self.which, src, self.level)

    def vmult_add(self, dst, src):
        dst += self.handler.solver.vmult(self.which, src, self.level)

    def Tvmult(self, dst, src):
        """Transpose application.  The A block, the mass operator and the plain Stokes operator are symmetric
        (vmult serves); the Stokes operator with unsymmetrised Newton derivatives is not, and neither the
        reference (MatrixFreeOperators::Base::Tvmult is never called on this path) nor the C ABI applies its
        transpose; B^T is only used through vmult as well."""
        newton = getattr(self.handler.solver, "unsymmetric_newton", False)
        if self.which == capi.OP_BT or (self.which == capi.OP_STOKES and newton):
            raise NotImplementedError("Tvmult of an unsymmetric operator is not part of the path (nor of the reference's)")
        self.vmult(dst, src)


class StokesOperator(_Operator):
    """matrix_free_operators.h:208-273; compute_diagonal is not implemented there either."""
    which = capi.OP_STOKES

    def compute_diagonal(self):
        raise NotImplementedError("StokesOperator::compute_diagonal (matrix_free_operators.cc:158-167)")


class BTBlockOperator(_Operator):
    which = capi.OP_BT


class ABlockOperator(_Operator):
    which = capi.OP_A

    def get_matrix_diagonal_inverse(self):
        lvl = self.handler.problem.hier.n_levels - 1 if self.level == capi.ACTIVE else self.level
        return self.handler.solver.inverse_diagonal(self.which, lvl)


class MassMatrixOperator(ABlockOperator):
    which = capi.OP_MASS


class StokesMatrixFreeHandler:
    """stokes_matrix_free.h:46-98: setup_dofs / assemble / build_preconditioner / solve."""

    def __init__(self, problem, lib=None, **options):
        self.problem = problem
        self.lib = lib or capi.load()
        self.options = options
        self.solver = None

    def name(self):
        return "GMG"                    # local_smoothing.cc:147

    def setup_dofs(self):
        if self.problem.hier is None:
            self.problem.build()

    def assemble(self):
        """evaluate_material_model + correct_stokes_rhs: tables are produced by cases/boxmesh."""
        self.setup_dofs()

    def build_preconditioner(self):
        self.solver = capi.Solver(self.lib, self.problem, **self.options)   # includes compute_diagonal

    def solve(self, rhs=None, x0=None, solve_newton_system=False):
        if self.solver is None:
            self.build_preconditioner()
        x, rep, rc = self.solver.solve(rhs, x0, newton=solve_newton_system, allow_failure=True)
        if rc == 1:
            raise capi.AgmgError("The Stokes solver did not converge (SolverControl::NoConvergence)")
        outputs = {"initial_nonlinear_residual": rep.initial_nonlinear_residual,
                   "final_linear_residual": rep.final_linear_residual,
                   "iterations": (rep.n_cheap_iterations, rep.n_expensive_iterations)}
        return x, outputs

    def operators(self, level=capi.ACTIVE):
        return (StokesOperator(self, level), ABlockOperator(self, level), BTBlockOperator(self, level),
                MassMatrixOperator(self, level))

    def get_cell_data_memory_consumption(self):
        return self.lib.get_cell_data_memory_consumption(self.solver.ctx)

    def get_mg_transfer_memory_consumption(self):
        return self.lib.get_mg_transfer_memory_consumption(self.solver.ctx)

    def get_constraint_memory_consumption(self):
        return self.lib.get_constraint_memory_consumption(self.solver.ctx)

    def get_dof_handler_memory_consumption(self):
        return self.lib.get_dof_handler_memory_consumption(self.solver.ctx)
