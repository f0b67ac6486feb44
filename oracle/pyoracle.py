"""ctypes loader of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (aspect_amd/) never imports
this module.
"""
import os
import subprocess

from aspect_amd import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = capi.Library(path, "oracle_",
                            optional=("vmult_device", "vmult_device_noalloc", "get_level_kernel_path", "get_overlap_statistics", "execute_solver_timings", "bench_vmult", "set_communicator", "rccl_get_unique_id", "set_communicator_rccl",
                                      "get_cell_data_memory_consumption",
                                      "get_mg_transfer_memory_consumption",
                                      "get_constraint_memory_consumption",
                                      "get_dof_handler_memory_consumption"))
    return _LIB


def solver(problem, **kw):
    return capi.Solver(load(), problem, **kw)
