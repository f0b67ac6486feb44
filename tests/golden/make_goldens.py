#!/usr/bin/env python3
"""Extracts the facts the reference's own regression goldens pin for the block-GMG Stokes solve
(tests/<name>/screen-output of tjhei/aspect 3.1.0-pre) into tests/golden/gmg_goldens.json.
Only numbers are kept (data, not source).  Run in the build container, where /root/reference
exists:   python tests/golden/make_goldens.py"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
CASES = ["sol_cx_2_gmg", "sol_cx_4_gmg", "sol_cx_4_gmg_no_avg", "no_cheap_gmg", "nsinker_gmg_idr",
         "nsinker_gmg_no_avg", "stokes", "refinement_nonadiabatic_temperature",
         "particle_interpolator_bilinear_3d", "periodic_box_gmg", "nsinker_gmg_gc",
         "poiseuille_2d_pressure_bc_gmg", "prescribed_dilation_gmg", "geometric_average_only_viscosity", "free_surface_blob_gmg",
         "simple_compressibility_iterated_stokes_gmg", "prescribed_dilation_newton",
         "nonlinear_channel_flow_tractions_Newton_Stokes_GMG",
         "nonlinear_channel_flow_tractions_dc_picard_Stokes_GMG",
         "nonlinear_channel_flow_velocities_Newton_Stokes_GMG",
         "tosi_benchmark_gmg_newton_solver", "tosi_benchmark_gmg_dc_picard_solver", "tangurnis_tala_gmg", "gmg_mesh_deform_prescribed", "crameri_benchmark_1_gmg", "free_surface_iterated_stokes_gmg",
         "solver_scheme_no_advection_single_stokes", "gmg_mesh_deform_topo", "convection_box_phase_function"]
SHELL_CASES = ["hollow_sphere_gmg", "spherical_nsinker_gmg_gc", "shell_3d_gmg", "hollow_sphere_gmg_project_q1"]


def parse(path):
    out = {"solves": []}
    cur = None
    for line in open(path):
        m = re.search(r"Number of active cells: ([\d,]+) \(on (\d+) levels\)", line)
        if m:
            out.setdefault("meshes", []).append({"cells": int(m.group(1).replace(",", "")), "levels": int(m.group(2))})
        m = re.search(r"Number of degrees of freedom: ([\d,]+) \(([\d,+]+)\)", line)
        if m:
            parts = [int(x.replace(",", "")) for x in m.group(2).split("+")]
            out["meshes"][-1]["dofs_u"], out["meshes"][-1]["dofs_p"] = parts[0], parts[1]
        m = re.search(r"Initial Newton Stokes residual = (\S+), v = (\S+), p = (\S+)", line)
        if m:
            out.setdefault("initial_newton_residuals", []).append([float(m.group(i)) for i in range(1, 4)])
        if "Solving Stokes system (GMG" in line:
            cur = {"mesh": len(out.get("meshes", [])) - 1}
            out["solves"].append(cur)
            m = re.search(r"\.\.\.\s*(\d+)\+(\d+) iterations", line)
            if m:
                cur["cheap"], cur["expensive"] = int(m.group(1)), int(m.group(2))
        if cur is None:
            continue
        m = re.search(r"GMG coarse size A: (\d+), coarse size S: (\d+)", line)
        if m:
            cur["coarse_A"], cur["coarse_S"] = int(m.group(1)), int(m.group(2))
        m = re.search(r"GMG n_levels: (\d+)", line)
        if m:
            cur["n_levels"] = int(m.group(1))
        m = re.search(r"Viscosity range: (\S+) - (\S+)", line)
        if m:
            cur["eta_min"], cur["eta_max"] = float(m.group(1)), float(m.group(2))
        m = re.search(r"Stokes solver: (\d+)\+(\d+) iterations", line)
        if m:
            cur["cheap"], cur["expensive"] = int(m.group(1)), int(m.group(2))
        m = re.search(r"Schur complement preconditioner: (\d+)\+(\d+) iterations", line)
        if m:
            cur["S_cheap"], cur["S_expensive"] = int(m.group(1)), int(m.group(2))
        m = re.search(r"A block preconditioner: (\d+)\+(\d+) iterations", line)
        if m:
            cur["A_cheap"], cur["A_expensive"] = int(m.group(1)), int(m.group(2))
        m = re.search(r"Relative nonlinear residual \((?:Stokes|total) system\) after nonlinear iteration (\d+): (\S+)", line)
        if m:
            out.setdefault("nonlinear_residuals", []).append(float(m.group(2)))
        m = re.search(r"Newton system information: Norm of the rhs: ([^,\s]+)(?:, Derivative scaling factor: (\S+))?", line)
        if m:
            out.setdefault("newton_rhs_norms", []).append(float(m.group(1)))
            out.setdefault("derivative_scaling_factors", []).append(float(m.group(2)) if m.group(2) else None)
        m = re.search(r"The linear solver tolerance is set to ([^.\s]+(?:\.\d+)?(?:e[-+]?\d+)?)\.", line)
        if m:
            out.setdefault("linear_tolerances", []).append(float(m.group(1)))
        m = re.search(r"RMS, max velocity:\s+(\S+) m/s, (\S+) m/s", line)
        if m:
            out.setdefault("rms_max_velocity", []).append([float(m.group(1)), float(m.group(2))])
        m = re.search(r"RMS, max velocity:\s+(\S+) m/year, (\S+) m/year", line)
        if m:
            out.setdefault("rms_max_velocity_per_year", []).append([float(m.group(1)), float(m.group(2))])
        m = re.search(r"Topography min/max:\s+(\S+) m, (\S+) m", line)
        if m:
            out.setdefault("topography_min_max", []).append([float(m.group(1)), float(m.group(2))])
        m = re.search(r"Pressure min/avg/max:\s+(\S+) Pa, (\S+) Pa, (\S+) Pa", line)
        if m:
            out.setdefault("pressure_min_avg_max", []).append([float(m.group(i)) for i in range(1, 4)])
        m = re.search(r"Mass fluxes through boundary parts:\s+(.*)$", line)
        if m:
            out.setdefault("mass_fluxes", []).append([float(t.split()[0]) for t in m.group(1).split(",")])
        m = re.search(r"Errors u_L2, p_L2:\s+([^,\s]+),\s*([^,\s]+)", line)
        if m:
            out.setdefault("errors_uL2_pL2", []).append([float(m.group(1)), float(m.group(2))])
        m = re.search(r"Errors u_L1, p_L1, u_L2, p_L2(?: topo_L2)?:?\s+([^,\s]+),\s*([^,\s]+),\s*([^,\s]+),\s*([^,\s]+)", line)
        if m:
            out["errors_uL1_pL1_uL2_pL2"] = [float(m.group(i)) for i in range(1, 5)]
    return out


def main():
    res = {"_source": "tjhei/aspect 3.1.0-pre tests/<case>/screen-output (deal.II 9.6 goldens)"}
    for c in CASES:
        res[c] = parse(os.path.join(REF, "tests", c, "screen-output"))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gmg_goldens.json")
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst)
    shell = {"_source": res["_source"]}
    for c in SHELL_CASES:
        shell[c] = parse(os.path.join(REF, "tests", c, "screen-output"))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shell_goldens.json")
    json.dump(shell, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst)


if __name__ == "__main__":
    main()
