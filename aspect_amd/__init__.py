"""aspect_amd: MI355X-native matrix-free block-GMG Stokes solve.

Host-side Python layer: synthetic mesh/table generation (standing in for the
deal.II/p4est host that the reference keeps), ctypes bindings to the C-ABI
library (``libaspect_amd.so``, HIP for gfx950) and a mirror of the reference's
operator/handler interface (``MatrixFreeStokesOperators``,
``StokesMatrixFreeHandler``).  The product compute path is the HIP library
only; nothing here falls back to a CPU implementation.
"""

__all__ = ["boxmesh", "cases", "capi", "operators"]
