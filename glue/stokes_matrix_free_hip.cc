// glue/stokes_matrix_free_hip.cc -- ASPECT-side binding of libaspect_amd.so (include/agmg.h).
//
// NOT COMPILED IN THIS REPOSITORY: deal.II and ASPECT's build are absent from the image.  It is the complete
// translation unit a maintainer would add to source/simulator/solver/ (plus the three-line patch to the handler's
// header and to create_matrix_free_solver shown in INTEGRATION.md); every deal.II / ASPECT call it makes is named
// with the reference line that makes the same call today.  Nothing in tests/, bench.py or the product library
// depends on this file.
//
// What it replaces in the reference (tjhei/aspect 3.1.0-pre):
//   StokesMatrixFreeHandlerLocalSmoothingImplementation<dim,2>::setup_dofs / assemble / build_preconditioner / solve
//   (stokes_matrix_free_local_smoothing.cc:1548-1878, 172-898, 1882-1894, 902-1545) keep building their DoFHandlers,
//   constraints, MatrixFree objects and OperatorCellData tables exactly as they do; the HIP backend takes those
//   objects, flattens them into the tables of agmg_level_desc / agmg_cell_data once per mesh / coefficient change,
//   and runs steps 1-7 of solve() on the GPU.
//
// Header patch (include/aspect/simulator/solver/stokes_matrix_free_local_smoothing.h, inside the class):
//     friend class HipBackend<dim, velocity_degree>;
//     std::unique_ptr<HipBackend<dim, velocity_degree>> hip;      // created in the constructor when
//                                                                 // "Stokes solver type = block GMG (HIP)"
// and in setup_dofs() / assemble() / build_preconditioner() / solve() a first line
//     if (hip) return hip->setup_dofs();   (etc.)
#include <aspect/simulator/solver/stokes_matrix_free_local_smoothing.h>
#include <aspect/mesh_deformation/interface.h>
#include <aspect/utilities.h>

#include <deal.II/base/mpi.h>
#include <deal.II/base/partitioner.h>
#include <deal.II/fe/fe_tools.h>
#include <deal.II/lac/la_parallel_block_vector.h>
#include <deal.II/matrix_free/fe_evaluation.h>
#include <deal.II/matrix_free/matrix_free.h>

#include <hip/hip_runtime_api.h>

extern "C" {
#include "agmg.h"
}

#include <algorithm>
#include <array>
#include <cstdint>
#include <map>
#include <vector>

namespace aspect
{
  using namespace dealii;

  template <int dim, int velocity_degree>
  class HipBackend
  {
      static_assert(velocity_degree == 2, "libaspect_amd.so implements the Q2-Q1 element (agmg_options.degree_v)");
      using Handler = StokesMatrixFreeHandlerLocalSmoothingImplementation<dim, velocity_degree>;
      using VectorType = dealii::LinearAlgebra::distributed::Vector<double>;
      using BlockVectorType = dealii::LinearAlgebra::distributed::BlockVector<double>;

    public:
      HipBackend(Handler &handler)
        : h(handler), ctx(nullptr)
      {}

      ~HipBackend()
      {
        if (ctx != nullptr)
          agmg_destroy(ctx);
      }

      // ------------------------------------------------------------------ create_matrix_free_solver + constructor
      // (stokes_matrix_free.cc:29-62; parameters.cc:432-584)
      void create(const unsigned int n_levels)
      {
        if (ctx != nullptr)
          agmg_destroy(ctx);
        const Parameters<dim> &prm = h.get_parameters();
        agmg_options opt;
        agmg_default_options(&opt);
        opt.dim = dim;
        opt.degree_v = velocity_degree;
        int device_count = 1;
        (void)hipGetDeviceCount(&device_count);
        opt.device = static_cast<int>(Utilities::MPI::this_mpi_process(h.get_mpi_communicator())) % device_count;
        opt.n_levels = static_cast<int>(n_levels);
        opt.linear_tolerance = prm.linear_stokes_solver_tolerance;
        opt.n_cheap_steps = static_cast<int>(prm.n_cheap_stokes_solver_steps);
        opt.n_expensive_steps = static_cast<int>(prm.n_expensive_stokes_solver_steps);
        opt.gmres_restart = static_cast<int>(prm.stokes_gmres_restart_length);
        opt.krylov_type = (prm.stokes_krylov_type == Parameters<dim>::StokesKrylovType::idr_s) ? AGMG_KRYLOV_IDR : AGMG_KRYLOV_GMRES;
        opt.idr_s = static_cast<int>(prm.idr_s_parameter);
        opt.A_block_tolerance = prm.linear_solver_A_block_tolerance;
        opt.S_block_tolerance = prm.linear_solver_S_block_tolerance;
        // BiCGStab instead of CG for the A block of the expensive preconditioner when the free-surface
        // stabilisation makes it non-symmetric (helper_functions.cc:1265-1275, stokes_matrix_free.h:302-330)
        opt.A_block_is_symmetric = h.sim.stokes_A_block_is_symmetric() ? 1 : 0;
        opt.world_size = static_cast<int>(Utilities::MPI::n_mpi_processes(h.get_mpi_communicator()));
        opt.rank = static_cast<int>(Utilities::MPI::this_mpi_process(h.get_mpi_communicator()));
        ctx = agmg_create(&opt);
        AssertThrow(ctx != nullptr, ExcMessage(std::string("agmg_create: ") + agmg_last_error()));
        if (opt.world_size > 1)
          {
            // inside one node the library talks RCCL itself; the 128-byte id travels by MPI
            uint8_t id[128];
            if (opt.rank == 0)
              check(agmg_rccl_get_unique_id(id), "agmg_rccl_get_unique_id");
            MPI_Bcast(id, 128, MPI_BYTE, 0, h.get_mpi_communicator());
            check(agmg_set_communicator_rccl(ctx, id), "agmg_set_communicator_rccl");
          }
      }

      // ------------------------------------------------------------------ setup_dofs()
      // Runs after the handler's own setup_dofs() (stokes_matrix_free_local_smoothing.cc:1548-1878) has distributed
      // the DoFs, built constraints_v / constraints_p, mg_constrained_dofs_* and the MatrixFree objects.
      void setup_dofs()
      {
        const unsigned int n_levels = h.get_triangulation().n_global_levels();
        AssertThrow(h.get_triangulation().n_global_levels() ==
                    Utilities::MPI::max(h.get_triangulation().n_levels(), h.get_mpi_communicator()),
                    ExcInternalError());
        AssertThrow(!h.get_triangulation().has_hanging_nodes(),
                    ExcMessage("this glue flattens uniformly refined meshes only.  The library takes adaptive ones on one "
                               "rank (agmg_level_desc: hanging-node CSR from AffineConstraints::get_constraint_entries, "
                               "refinement_edge_* from mg_constrained_dofs.get_refinement_edge_indices(level), copy_* from "
                               "the copy_indices of MGLevelGlobalTransfer, agmg_set_level(AGMG_ACTIVE, desc)); flattening "
                               "them here is the next step of the glue"));
        create(n_levels);
        level_cells.assign(n_levels, {});
        for (unsigned int level = 0; level < n_levels; ++level)
          {
            // MatrixFree object of the level: the one mg_matrices_A_block[level] was initialised with
            // (stokes_matrix_free_local_smoothing.cc:1843-1877)
            const MatrixFree<dim, double> &mf = *h.mg_matrices_A_block[level].get_matrix_free();
            const MatrixFree<dim, double> &mf_p = *h.mg_matrices_Schur_complement[level].get_matrix_free();
            flatten_level(mf, mf_p, level, &h.mg_constrained_dofs_A_block, level > 0 ? &level_cells[level - 1] : nullptr,
                          level_cells[level]);
          }
        // The library's active mesh IS its finest level (agmg_set_level(AGMG_ACTIVE, NULL): uniformly refined meshes).
        // deal.II numbers the active DoFs and the level DoFs of the finest level independently, so vectors and
        // tables of the active mesh are permuted into the level's numbering / cell order here: the same cells carry
        // both numberings.
        check(agmg_set_level(ctx, AGMG_ACTIVE, nullptr), "agmg_set_level(active)");
        build_active_permutation(n_levels - 1);
      }

      // active local dof -> local dof of the finest level (velocity and pressure), and the position of every active
      // cell in the level's flattened order
      void build_active_permutation(const unsigned int finest)
      {
        const MatrixFree<dim, double> &mf_a = *h.matrix_free_objects[0];
        const MatrixFree<dim, double> &mf_l = *h.mg_matrices_A_block[finest].get_matrix_free();
        const MatrixFree<dim, double> &mf_lp = *h.mg_matrices_Schur_complement[finest].get_matrix_free();
        const Utilities::MPI::Partitioner &pa_v = *mf_a.get_vector_partitioner(0), &pa_p = *mf_a.get_vector_partitioner(1);
        const Utilities::MPI::Partitioner &pl_v = *mf_l.get_vector_partitioner(0), &pl_p = *mf_lp.get_vector_partitioner(0);
        perm_v.assign(pa_v.locally_owned_size() + pa_v.n_ghost_indices(), numbers::invalid_unsigned_int);
        perm_p.assign(pa_p.locally_owned_size() + pa_p.n_ghost_indices(), numbers::invalid_unsigned_int);
        std::vector<types::global_dof_index> a(h.fe_v.n_dofs_per_cell()), l(h.fe_v.n_dofs_per_cell());
        std::vector<types::global_dof_index> ap(h.fe_p.n_dofs_per_cell()), lp(h.fe_p.n_dofs_per_cell());
        active_position.clear();
        for (unsigned int batch = 0; batch < mf_a.n_cell_batches(); ++batch)
          for (unsigned int lane = 0; lane < mf_a.n_active_entries_per_cell_batch(batch); ++lane)
            {
              const auto cv = mf_a.get_cell_iterator(batch, lane, 0);
              const auto cp = mf_a.get_cell_iterator(batch, lane, 1);
              AssertThrow(static_cast<unsigned int>(cv->level()) == finest, ExcMessage("active cell below the finest level"));
              cv->get_dof_indices(a);
              cv->get_mg_dof_indices(l);
              cp->get_dof_indices(ap);
              cp->get_mg_dof_indices(lp);
              for (unsigned int i = 0; i < a.size(); ++i)
                perm_v[pa_v.global_to_local(a[i])] = pl_v.global_to_local(l[i]);
              for (unsigned int i = 0; i < ap.size(); ++i)
                perm_p[pa_p.global_to_local(ap[i])] = pl_p.global_to_local(lp[i]);
              active_position.push_back(level_cells[finest].at(cv->id()));
            }
        n_level_dofs_v = pl_v.locally_owned_size() + pl_v.n_ghost_indices();
        n_level_dofs_p = pl_p.locally_owned_size() + pl_p.n_ghost_indices();
        n_level_owned_v = pl_v.locally_owned_size();
      }

      // ------------------------------------------------------------------ assemble() = evaluate_material_model() +
      // correct_stokes_rhs() (stokes_matrix_free_local_smoothing.cc:172-898)
      void assemble()
      {
        h.evaluate_material_model();       // fills active_cell_data / level_cell_data exactly as today
        const unsigned int n_levels = h.get_triangulation().n_global_levels();
        for (unsigned int level = 0; level < n_levels; ++level)
          push_cell_data(static_cast<int>(level), h.level_cell_data[level], *h.mg_matrices_A_block[level].get_matrix_free());
        push_cell_data(AGMG_ACTIVE, h.active_cell_data, *h.matrix_free_objects[0]);
        push_free_surface_faces();
        // correct_stokes_rhs() (:761-898): u0 = zero vector carrying the inhomogeneous boundary values
        if (h.sim.boundary_velocity_manager.get_prescribed_boundary_velocity_indicators().size() > 0 ||
            !h.get_parameters().prescribed_velocity_boundary_indicators.empty())
          {
            BlockVectorType u0(2), rhs(2);
            h.stokes_matrix.initialize_dof_vector(u0);
            h.stokes_matrix.initialize_dof_vector(rhs);
            u0 = 0.;
            h.sim.current_constraints.distribute(u0);                    // :775-787
            u0.update_ghost_values();
            internal::ChangeVectorTypes::copy(rhs, h.sim.system_rhs);    // matrix_free_operators.cc:60-93
            std::vector<double> u0_flat, rhs_flat;
            to_flat(u0, u0_flat);
            to_flat(rhs, rhs_flat);
            check(agmg_rhs_correction(ctx, u0_flat.data(), rhs_flat.data()), "agmg_rhs_correction");
            from_flat(rhs_flat, rhs);
            internal::ChangeVectorTypes::copy(h.sim.system_rhs, rhs);
          }
      }

      // ------------------------------------------------------------------ build_preconditioner() (:1882-1894)
      void build_preconditioner()
      {
        check(agmg_compute_diagonals(ctx), "agmg_compute_diagonals");
      }

      // ------------------------------------------------------------------ solve() (:902-1545)
      StokesSolver::SolverOutputs
      solve(const LinearAlgebra::BlockVector &system_rhs, const bool solve_newton_system,
            const double last_pressure_normalization_adjustment, LinearAlgebra::BlockVector &solution_vector)
      {
        StokesSolver::SolverOutputs outputs;
        const unsigned int block_vel = h.introspection().block_indices.velocities;
        const unsigned int block_p = h.introspection().block_indices.pressure;

        // steps 1-2 of the reference's solve(): the start vector (:1085-1119)
        LinearAlgebra::BlockVector distributed_stokes_solution(h.introspection().index_sets.stokes_partitioning, h.get_mpi_communicator());
        LinearAlgebra::BlockVector initial_guess(h.introspection().index_sets.stokes_partitioning, h.get_mpi_communicator());
        if (!solve_newton_system)
          {
            initial_guess.block(block_vel) = h.get_current_linearization_point().block(block_vel);
            initial_guess.block(block_p) = h.get_current_linearization_point().block(block_p);
            h.denormalize_pressure(last_pressure_normalization_adjustment, initial_guess);
          }
        else
          {
            initial_guess.block(block_vel) = 0;
            initial_guess.block(block_p) = 0;
          }
        h.get_current_constraints().set_zero(initial_guess);
        initial_guess.block(block_p) /= h.get_pressure_scaling();

        // ChangeVectorTypes::copy hand-off (matrix_free_operators.cc:60-93): Trilinos -> deal.II distributed vectors
        // laid out by the MatrixFree partitioners, which is the rank-local layout the C ABI takes
        BlockVectorType x(2), rhs(2);
        h.stokes_matrix.initialize_dof_vector(x);
        h.stokes_matrix.initialize_dof_vector(rhs);
        internal::ChangeVectorTypes::copy(x, initial_guess);
        internal::ChangeVectorTypes::copy(rhs, system_rhs);
        std::vector<double> x_flat, rhs_flat;
        to_flat(x, x_flat);
        to_flat(rhs, rhs_flat);

        // steps 3-7: tolerance, cheap solve, fall-through to the expensive one, report (:1121-1495)
        agmg_solve_report rep;
        const int status = agmg_solve(ctx, rhs_flat.data(), x_flat.data(), solve_newton_system ? 1 : 0, &rep);
        if (status < 0)
          AssertThrow(false, ExcMessage(std::string("agmg_solve: ") + agmg_last_error()));
        if (status == 1)
          {
            // both solvers failed: the reference's failure path (:1428-1468, utilities.cc:2786-2848)
            std::vector<SolverControl> controls;
            SolverControl cheap(rep.n_cheap_iterations, rep.solver_tolerance, true);
            SolverControl expensive(rep.n_expensive_iterations, rep.solver_tolerance, true);
            cheap.enable_history_data();
            expensive.enable_history_data();
            for (int i = 0; i < rep.n_history; ++i)
              (i <= rep.n_cheap_iterations ? cheap : expensive).check(i <= rep.n_cheap_iterations ? i : i - rep.n_cheap_iterations - 1,
                                                                      rep.history[i]);
            controls.push_back(cheap);
            controls.push_back(expensive);
            Utilities::throw_linear_solver_failure_exception("iterative Stokes solver (HIP)",
                                                             "HipBackend::solve",
                                                             controls,
                                                             ExcMessage("no convergence within the cheap and the expensive iterations"),
                                                             h.get_mpi_communicator(),
                                                             h.get_parameters().output_directory + "solver_history.txt");
          }
        h.get_pcout() << rep.n_cheap_iterations << '+' << rep.n_expensive_iterations << " iterations." << std::endl;   // :1421-1425
        if (h.print_details)
          {
            h.get_pcout() << "     Schur complement preconditioner: " << rep.n_S_preconditioner_cheap << '+'
                          << rep.n_S_preconditioner_expensive << " iterations." << std::endl;
            h.get_pcout() << "     A block preconditioner: " << rep.n_A_preconditioner_cheap << '+'
                          << rep.n_A_preconditioner_expensive << " iterations." << std::endl;
          }
        {
          // signal successful solver (:1497-1502): the statistics columns are filled from the two SolverControls
          SolverControl cheap(rep.n_cheap_iterations + 1, rep.solver_tolerance), expensive(rep.n_expensive_iterations + 1, rep.solver_tolerance);
          cheap.enable_history_data();
          expensive.enable_history_data();
          for (int i = 0; i < rep.n_history; ++i)
            if (i <= rep.n_cheap_iterations)
              cheap.check(i, rep.history[i]);
            else
              expensive.check(i - rep.n_cheap_iterations - 1, rep.history[i]);
          h.get_signals().post_stokes_solver(h.sim,
                                             rep.n_S_preconditioner_cheap + rep.n_S_preconditioner_expensive,
                                             rep.n_A_preconditioner_cheap + rep.n_A_preconditioner_expensive,
                                             cheap, expensive);
        }

        // back: constraints, pressure scaling, null space, normalisation (:1505-1541)
        from_flat(x_flat, x);
        x.update_ghost_values();
        internal::ChangeVectorTypes::copy(distributed_stokes_solution, x);
        h.get_current_constraints().distribute(distributed_stokes_solution);
        distributed_stokes_solution.block(block_p) *= h.get_pressure_scaling();
        solution_vector.block(block_vel) = distributed_stokes_solution.block(block_vel);
        solution_vector.block(block_p) = distributed_stokes_solution.block(block_p);
        h.remove_nullspace(solution_vector, distributed_stokes_solution);
        outputs.initial_nonlinear_residual = rep.initial_nonlinear_residual;
        outputs.final_linear_residual = rep.final_linear_residual;
        if (!solve_newton_system)
          outputs.pressure_normalization_adjustment = h.normalize_pressure(solution_vector);
        return outputs;
      }

    private:
      Handler &h;
      agmg_ctx *ctx;
      // position of every cell in the flattened order of its level (parents are looked up by CellId)
      std::vector<std::map<CellId, int>> level_cells;
      std::vector<unsigned int> perm_v, perm_p;      // active local dof -> local dof of the finest level
      std::vector<int> active_position;              // active cell (MatrixFree batch order) -> cell of the finest level
      std::size_t n_level_dofs_v = 0, n_level_dofs_p = 0, n_level_owned_v = 0;

      void check(const int status, const char *what) const
      {
        AssertThrow(status == 0, ExcMessage(std::string(what) + ": " + agmg_last_error()));
      }

      // ---- block vector (active numbering, MatrixFree partitioner layout: owned entries first, ghosts behind them)
      // <-> the flat rank-local vector of the library [velocity (dof = dim * node + component) ; pressure] in the
      // numbering of the finest level.  A rank owns the same cells on the active mesh and on the finest level, so an
      // owned active dof maps to an owned level dof.
      void to_flat(const BlockVectorType &v, std::vector<double> &flat) const
      {
        const VectorType &u = v.block(0), &p = v.block(1);
        flat.assign(n_level_dofs_v + n_level_dofs_p, 0.0);
        for (unsigned int i = 0; i < perm_v.size(); ++i)
          if (perm_v[i] != numbers::invalid_unsigned_int)
            flat[perm_v[i]] = u.local_element(i);                         // (ghost entries are refreshed by the library)
        for (unsigned int i = 0; i < perm_p.size(); ++i)
          if (perm_p[i] != numbers::invalid_unsigned_int)
            flat[n_level_dofs_v + perm_p[i]] = p.local_element(i);
      }

      void from_flat(const std::vector<double> &flat, BlockVectorType &v) const
      {
        VectorType &u = v.block(0), &p = v.block(1);
        for (unsigned int i = 0; i < u.locally_owned_size(); ++i)
          u.local_element(i) = flat[perm_v[i]];
        for (unsigned int i = 0; i < p.locally_owned_size(); ++i)
          p.local_element(i) = flat[n_level_dofs_v + perm_p[i]];
      }

      // ---- one multigrid level -> agmg_level_desc
      // mf: MatrixFree whose dof handler 0 is dof_handler_v; mf_p: the one whose last dof handler is dof_handler_p
      // (the level objects of the A block and of the Schur complement; on the active mesh both are matrix_free_objects[0])
      void flatten_level(const MatrixFree<dim, double> &mf, const MatrixFree<dim, double> &mf_p, const unsigned int level,
                         const MGConstrainedDoFs *mg_constrained_v, const std::map<CellId, int> *coarser,
                         std::map<CellId, int> &positions)
      {
        const bool active = false;
        const unsigned int comp_v = 0, comp_p = (&mf == &mf_p) ? 1 : 0;
        const Utilities::MPI::Partitioner &part_v = *mf.get_vector_partitioner(comp_v);
        const Utilities::MPI::Partitioner &part_p = *mf_p.get_vector_partitioner(comp_p);
        constexpr unsigned int nv = Utilities::pow(3, dim), np = Utilities::pow(2, dim);

        // lexicographic position -> cell-local dof of COMPONENT 0 of FESystem(FE_Q(2), dim); the other components of
        // the same node are the following dofs of the node (deal.II numbers the dim components of a support point
        // consecutively unless a component-wise renumbering is asked for; dof_handler_v has none, :1575-1590)
        const std::vector<unsigned int> lex_q2 = FETools::lexicographic_to_hierarchic_numbering<dim>(2);
        const std::vector<unsigned int> lex_q1 = FETools::lexicographic_to_hierarchic_numbering<dim>(1);
        std::array<unsigned int, nv> lex_to_cell_v;
        for (unsigned int i = 0; i < nv; ++i)
          lex_to_cell_v[i] = h.fe_v.component_to_system_index(0, lex_q2[i]);

        std::vector<uint32_t> cnv, cnp;
        std::vector<double> cell_h, inv_jac, jxw;
        std::vector<int32_t> parent;
        std::vector<uint8_t> child;
        const bool cartesian = dynamic_cast<const MappingCartesian<dim> *>(&h.get_mapping()) != nullptr &&
                               !h.get_parameters().mesh_deformation_enabled;      // core.cc:115-121
        std::vector<types::global_dof_index> dofs_v(h.fe_v.n_dofs_per_cell()), dofs_p(h.fe_p.n_dofs_per_cell());
        FEEvaluation<dim, velocity_degree, velocity_degree + 1, dim, double> phi(mf, comp_v);
        int position = 0;
        positions.clear();
        for (unsigned int batch = 0; batch < mf.n_cell_batches(); ++batch)
          {
            if (!cartesian)
              phi.reinit(batch);
            for (unsigned int lane = 0; lane < mf.n_active_entries_per_cell_batch(batch); ++lane, ++position)
              {
                const auto cell_v = mf.get_cell_iterator(batch, lane, comp_v);
                const auto cell_p = mf_p.get_cell_iterator(batch, lane, comp_p);    // same triangulation, same batches
                AssertThrow(cell_v->id() == cell_p->id(), ExcInternalError());
                positions[cell_v->id()] = position;
                if (active)
                  {
                    cell_v->get_dof_indices(dofs_v);
                    cell_p->get_dof_indices(dofs_p);
                  }
                else
                  {
                    cell_v->get_mg_dof_indices(dofs_v);
                    cell_p->get_mg_dof_indices(dofs_p);
                  }
                for (unsigned int i = 0; i < nv; ++i)
                  {
                    const unsigned int local = part_v.global_to_local(dofs_v[lex_to_cell_v[i]]);
                    AssertThrow(local % dim == 0, ExcMessage("velocity dofs of a support point are not numbered consecutively"));
                    cnv.push_back(local / dim);
                  }
                for (unsigned int i = 0; i < np; ++i)
                  cnp.push_back(part_p.global_to_local(dofs_p[lex_q1[i]]));
                if (cartesian)
                  for (unsigned int d = 0; d < dim; ++d)
                    cell_h.push_back(cell_v->extent_in_direction(d));
                else
                  for (unsigned int q = 0; q < nv; ++q)
                    {
                      // FEEvaluation::inverse_jacobian(q) is J^{-T}: entry [d][e] = d xi_e / d x_d; the ABI wants [e][d]
                      const auto jit = phi.inverse_jacobian(q);
                      for (unsigned int e = 0; e < dim; ++e)
                        for (unsigned int d = 0; d < dim; ++d)
                          inv_jac.push_back(jit[d][e][lane]);
                      jxw.push_back(phi.JxW(q)[lane]);
                    }
                if (!active && level > 0)
                  {
                    const auto p = cell_v->parent();
                    const auto it = coarser->find(p->id());
                    AssertThrow(it != coarser->end(), ExcMessage("parent of a level cell is not a local cell of the coarser level"));
                    parent.push_back(it->second);
                    uint8_t c = 0;
                    for (unsigned int k = 0; k < GeometryInfo<dim>::max_children_per_cell; ++k)
                      if (p->child(k) == cell_v)
                        c = static_cast<uint8_t>(k);                   // deal.II child k: bit d = upper half in direction d
                    child.push_back(c);
                  }
              }
          }

        // constraint bits per node: Dirichlet rows of the level (boundary indices + no-normal-flux constraints merged
        // into level_constraints_v, :1771-1822) or of the active mesh (constraints_v, :1597-1620).  Rows with entries
        // (curved free slip) are refused: they are not homogeneous Dirichlet rows.
        const unsigned int n_nodes_v = (part_v.locally_owned_size() + part_v.n_ghost_indices()) / dim;
        const unsigned int n_nodes_p = part_p.locally_owned_size() + part_p.n_ghost_indices();
        std::vector<uint8_t> mask_v(n_nodes_v, 0), mask_p(n_nodes_p, 0);
        const AffineConstraints<double> &con = mf.get_affine_constraints(comp_v);
        for (unsigned int local = 0; local < n_nodes_v * dim; ++local)
          {
            const types::global_dof_index g = part_v.local_to_global(local);
            if (con.is_constrained(g))
              {
                AssertThrow(con.get_constraint_entries(g) == nullptr || con.get_constraint_entries(g)->empty(),
                            ExcMessage("libaspect_amd.so: constraints with entries are not supported"));
                mask_v[local / dim] |= static_cast<uint8_t>(1u << (local % dim));
              }
          }
        (void)mg_constrained_v;      // (already merged into the level's AffineConstraints by the handler)

        // ghost exchange plan from the partitioner: ghost_targets() = (owner rank, number of my ghosts it owns), in the
        // order of the ghost range; import_targets() / import_indices() = (rank, count) and the owned local indices it
        // ghosts, in the order that rank stores them
        agmg_level_desc d = {};
        std::vector<int32_t> neighbor, send_off_v, recv_off_v, send_off_p, recv_off_p;
        std::vector<uint32_t> send_nodes_v, send_nodes_p;
        build_plan(part_v, dim, part_p, neighbor, send_off_v, send_nodes_v, recv_off_v, send_off_p, send_nodes_p, recv_off_p);
        std::vector<uint32_t> global_v(n_nodes_v), global_p(n_nodes_p);
        for (unsigned int n = 0; n < n_nodes_v; ++n)
          global_v[n] = static_cast<uint32_t>(part_v.local_to_global(n * dim) / dim);
        for (unsigned int n = 0; n < n_nodes_p; ++n)
          global_p[n] = static_cast<uint32_t>(part_p.local_to_global(n));

        d.n_cells = position;
        d.n_nodes_v = static_cast<int32_t>(n_nodes_v);
        d.n_nodes_p = static_cast<int32_t>(n_nodes_p);
        d.n_owned_nodes_v = static_cast<int32_t>(part_v.locally_owned_size() / dim);
        d.n_owned_nodes_p = static_cast<int32_t>(part_p.locally_owned_size());
        d.cell_nodes_v = cnv.data();
        d.cell_nodes_p = cnp.data();
        d.constrained_v = mask_v.data();
        d.constrained_p = mask_p.data();
        d.geometry_kind = cartesian ? AGMG_GEOM_CARTESIAN : AGMG_GEOM_GENERAL;
        d.cell_h = cartesian ? cell_h.data() : nullptr;
        d.inv_jacobian = cartesian ? nullptr : inv_jac.data();
        d.jxw = cartesian ? nullptr : jxw.data();
        d.parent_cell = parent.empty() ? nullptr : parent.data();
        d.child_index = child.empty() ? nullptr : child.data();
        d.n_neighbors = static_cast<int32_t>(neighbor.size());
        d.neighbor_rank = neighbor.data();
        d.send_offsets_v = send_off_v.data();
        d.send_nodes_v = send_nodes_v.data();
        d.recv_offsets_v = recv_off_v.data();
        d.send_offsets_p = send_off_p.data();
        d.send_nodes_p = send_nodes_p.data();
        d.recv_offsets_p = recv_off_p.data();
        d.global_nodes_v = global_v.data();
        d.global_nodes_p = global_p.data();
        // one message round per mass application inside the Schur smoother (optional)
        std::vector<int32_t> sym_rank, sym_off;
        std::vector<uint32_t> sym_nodes;
        if (Utilities::MPI::n_mpi_processes(this->get_mpi_communicator()) > 1)
          {
            build_symmetric_plan(part_p, this->get_mpi_communicator(), sym_rank, sym_off, sym_nodes);
            d.n_sym_neighbors_p = static_cast<int32_t>(sym_rank.size());
            d.sym_neighbor_rank_p = sym_rank.data();
            d.sym_offsets_p = sym_off.data();
            d.sym_nodes_p = sym_nodes.data();
            // (the same for the velocity nodes, on part_v with dim dofs per node, gives d.sym_*_v: the A-block smoother)
          }
        // (d.replicated stays 0: deal.II partitions every level.  Holding the coarsest levels whole on every rank, INTEGRATION.md,
        // needs the level's cells and DoF tables of all ranks: a serial copy of the coarse levels of the triangulation.)
        check(agmg_set_level(ctx, static_cast<int>(level), &d), "agmg_set_level");
      }

      // neighbours = union of the ranks of both spaces, ascending; per neighbour the owned nodes it ghosts and the
      // offset of its group among my ghosts (ghosts are stored grouped by owner in ascending rank order because owners
      // hold contiguous ranges of the global numbering)
      static void space_plan(const Utilities::MPI::Partitioner &part, const unsigned int dofs_per_node,
                             std::map<unsigned int, std::vector<uint32_t>> &send, std::map<unsigned int, int32_t> &recv_count)
      {
        unsigned int k = 0;
        const auto &ranges = part.import_indices();            // (begin, end) ranges of owned local indices
        std::vector<unsigned int> flat;
        for (const auto &r : ranges)
          for (unsigned int i = r.first; i < r.second; ++i)
            flat.push_back(i);
        for (const auto &t : part.import_targets())
          {
            std::vector<uint32_t> &nodes = send[t.first];
            for (unsigned int i = 0; i < t.second; ++i, ++k)
              if (flat[k] % dofs_per_node == 0)
                nodes.push_back(flat[k] / dofs_per_node);
          }
        for (const auto &t : part.ghost_targets())
          recv_count[t.first] = static_cast<int32_t>(t.second / dofs_per_node);
      }

      static void build_plan(const Utilities::MPI::Partitioner &part_v, const unsigned int dim_, const Utilities::MPI::Partitioner &part_p,
                             std::vector<int32_t> &neighbor, std::vector<int32_t> &send_off_v, std::vector<uint32_t> &send_nodes_v,
                             std::vector<int32_t> &recv_off_v, std::vector<int32_t> &send_off_p, std::vector<uint32_t> &send_nodes_p,
                             std::vector<int32_t> &recv_off_p)
      {
        std::map<unsigned int, std::vector<uint32_t>> sv, sp;
        std::map<unsigned int, int32_t> rv, rp;
        space_plan(part_v, dim_, sv, rv);
        space_plan(part_p, 1, sp, rp);
        std::map<unsigned int, bool> all;
        for (const auto &x : sv) all[x.first] = true;
        for (const auto &x : sp) all[x.first] = true;
        for (const auto &x : rv) all[x.first] = true;
        for (const auto &x : rp) all[x.first] = true;
        send_off_v.assign(1, 0); send_off_p.assign(1, 0); recv_off_v.assign(1, 0); recv_off_p.assign(1, 0);
        for (const auto &x : all)
          {
            neighbor.push_back(static_cast<int32_t>(x.first));
            const auto &a = sv[x.first];
            send_nodes_v.insert(send_nodes_v.end(), a.begin(), a.end());
            send_off_v.push_back(static_cast<int32_t>(send_nodes_v.size()));
            const auto &b = sp[x.first];
            send_nodes_p.insert(send_nodes_p.end(), b.begin(), b.end());
            send_off_p.push_back(static_cast<int32_t>(send_nodes_p.size()));
            recv_off_v.push_back(recv_off_v.back() + rv[x.first]);
            recv_off_p.push_back(recv_off_p.back() + rp[x.first]);
          }
      }

      // Symmetric exchange plan of the pressure nodes (agmg_level_desc::sym_*; optional: the one-round Schur smoother).
      // deal.II's partitioner knows owner -> ghost only, so the owner mediates: it knows who ghosts each of its nodes
      // (import_targets / import_indices) and tells every holder of a node who the other holders are (one
      // some_to_some); each rank then lists, per rank it shares a node with, ascending, its local nodes in the order of
      // the global index -- both sides build the same order without talking to each other.
      static void build_symmetric_plan(const Utilities::MPI::Partitioner &part, const MPI_Comm comm,
                                       std::vector<int32_t> &sym_rank, std::vector<int32_t> &sym_off, std::vector<uint32_t> &sym_nodes)
      {
        const unsigned int me = Utilities::MPI::this_mpi_process(comm);
        // holders[local owned node] = ranks that ghost it
        std::map<unsigned int, std::vector<unsigned int>> holders;
        {
          std::vector<unsigned int> flat;
          for (const auto &r : part.import_indices())
            for (unsigned int i = r.first; i < r.second; ++i) flat.push_back(i);
          unsigned int k = 0;
          for (const auto &t : part.import_targets())
            for (unsigned int i = 0; i < t.second; ++i, ++k) holders[flat[k]].push_back(t.first);
        }
        // shared[rank] = global indices of the nodes I hold together with that rank
        std::map<unsigned int, std::vector<types::global_dof_index>> shared, tell;
        for (const auto &h : holders)
          {
            const types::global_dof_index g = part.local_to_global(h.first);
            for (const unsigned int a : h.second)
              {
                shared[a].push_back(g);                       // owner <-> ghoster
                for (const unsigned int b : h.second)
                  if (b != a) { tell[a].push_back(g); tell[a].push_back(b); }   // ghoster <-> ghoster, by way of the owner
              }
          }
        for (const auto &msg : Utilities::MPI::some_to_some(comm, tell))
          for (std::size_t i = 0; i + 1 < msg.second.size(); i += 2)
            shared[static_cast<unsigned int>(msg.second[i + 1])].push_back(msg.second[i]);
        // my ghosts and their owners
        for (unsigned int gh = 0; gh < part.n_ghost_indices(); ++gh)
          {
            const types::global_dof_index g = part.local_to_global(part.locally_owned_size() + gh);
            unsigned int owner = 0, seen = 0;
            for (const auto &t : part.ghost_targets())
              { if (gh < seen + t.second) { owner = t.first; break; } seen += t.second; }
            shared[owner].push_back(g);
          }
        sym_rank.clear(); sym_off.assign(1, 0); sym_nodes.clear();
        for (auto &x : shared)                                // std::map: ascending ranks
          {
            if (x.first == me || x.second.empty()) continue;
            std::sort(x.second.begin(), x.second.end());
            x.second.erase(std::unique(x.second.begin(), x.second.end()), x.second.end());
            sym_rank.push_back(static_cast<int32_t>(x.first));
            for (const types::global_dof_index g : x.second) sym_nodes.push_back(static_cast<uint32_t>(part.global_to_local(g)));
            sym_off.push_back(static_cast<int32_t>(sym_nodes.size()));
          }
      }

      // ---- OperatorCellData (matrix_free_operators.h:91-203) -> agmg_cell_data.  The reference's tables are indexed
      // [cell batch][q] with one VectorizedArray lane per cell; the ABI wants [cell in flattened order][q].
      void push_cell_data(const int level, const MatrixFreeStokesOperators::OperatorCellData<dim, double> &cd,
                          const MatrixFree<dim, double> &mf)
      {
        const unsigned int nq_table = cd.viscosity.size(1);
        constexpr unsigned int nq = Utilities::pow(3, dim), nsym = dim * (dim + 1) / 2;
        std::vector<double> eta, strain, dEde, dEdp, dil_lhs, dil_de, dil_dp;
        const bool newton = cd.enable_newton_derivatives, dilation = cd.enable_prescribed_dilation;
        // (level tables: the flattened order IS the batch order of the level's MatrixFree.  Active tables: rows are
        // produced in the batch order of the active MatrixFree and moved to the level's cell order afterwards.)
        for (unsigned int batch = 0; batch < mf.n_cell_batches(); ++batch)
          for (unsigned int lane = 0; lane < mf.n_active_entries_per_cell_batch(batch); ++lane)
            {
              for (unsigned int q = 0; q < nq_table; ++q)
                eta.push_back(cd.viscosity(batch, q)[lane]);
              if (newton)
                for (unsigned int q = 0; q < nq; ++q)
                  {
                    // deal.II unrolls symmetric tensors diagonal first, then (0,1), (0,2), (1,2): the order of the ABI
                    for (unsigned int s = 0; s < nsym; ++s)
                      {
                        const TableIndices<2> ij = SymmetricTensor<2, dim>::unrolled_to_component_indices(s);
                        strain.push_back(cd.strain_rate_table(batch, q)[ij][lane]);
                        dEde.push_back(cd.newton_factor_wrt_strain_rate_table(batch, q)[ij][lane]);
                      }
                    dEdp.push_back(cd.newton_factor_wrt_pressure_table(batch, q)[lane]);
                  }
              if (dilation)
                for (unsigned int q = 0; q < nq; ++q)
                  {
                    dil_lhs.push_back(cd.dilation_lhs_term_table(batch, q)[lane]);
                    if (newton)
                      {
                        for (unsigned int s = 0; s < nsym; ++s)
                          dil_de.push_back(cd.dilation_derivative_wrt_strain_rate_table(batch, q)[SymmetricTensor<2, dim>::unrolled_to_component_indices(s)][lane]);
                        dil_dp.push_back(cd.dilation_derivative_wrt_pressure_table(batch, q)[lane]);
                      }
                  }
            }
        if (level == AGMG_ACTIVE)
          {
            const auto reorder = [&](std::vector<double> &t, const unsigned int width)
            {
              if (t.empty())
                return;
              std::vector<double> out(t.size());
              for (std::size_t cell = 0; cell < active_position.size(); ++cell)
                std::copy(t.begin() + cell * width, t.begin() + (cell + 1) * width, out.begin() + std::size_t(active_position[cell]) * width);
              t.swap(out);
            };
            reorder(eta, nq_table);
            reorder(strain, nq * nsym); reorder(dEde, nq * nsym); reorder(dEdp, nq);
            reorder(dil_lhs, nq); reorder(dil_de, nq * nsym); reorder(dil_dp, nq);
          }
        agmg_cell_data d = {};
        d.is_compressible = cd.is_compressible ? 1 : 0;
        d.enable_newton_derivatives = newton ? 1 : 0;
        d.enable_prescribed_dilation = dilation ? 1 : 0;
        d.symmetrize_newton_system = cd.symmetrize_newton_system ? 1 : 0;
        d.pressure_scaling = cd.pressure_scaling;
        d.n_q_viscosity = static_cast<int32_t>(nq_table);
        d.viscosity = eta.data();
        d.strain_rate_table = newton ? strain.data() : nullptr;
        d.newton_factor_wrt_strain_rate = newton ? dEde.data() : nullptr;
        d.newton_factor_wrt_pressure = newton ? dEdp.data() : nullptr;
        d.dilation_lhs_term = dilation ? dil_lhs.data() : nullptr;
        d.dilation_derivative_wrt_strain_rate = (dilation && newton) ? dil_de.data() : nullptr;
        d.dilation_derivative_wrt_pressure = (dilation && newton) ? dil_dp.data() : nullptr;
        check(agmg_set_coefficients(ctx, level, &d), "agmg_set_coefficients");
      }

      // ---- free-surface stabilisation faces of the active Stokes operator (:665-756; matrix_free_operators.cc:294-333)
      void push_free_surface_faces()
      {
        const auto &cd = h.active_cell_data;
        agmg_face_data fd = {};
        if (!cd.apply_stabilization_free_surface_faces)
          {
            check(agmg_set_free_surface_faces(ctx, &fd), "agmg_set_free_surface_faces");
            return;
          }
        const MatrixFree<dim, double> &mf = *h.matrix_free_objects[0];
        constexpr unsigned int nqf = Utilities::pow(3, dim - 1);
        std::vector<uint32_t> face_cell;
        std::vector<uint8_t> face_number;
        std::vector<double> stab, normal, jxw;
        const bool cartesian = dynamic_cast<const MappingCartesian<dim> *>(&h.get_mapping()) != nullptr &&
                               !h.get_parameters().mesh_deformation_enabled;
        FEFaceEvaluation<dim, velocity_degree, velocity_degree + 1, dim, double> phi(mf, true, 0);
        const unsigned int n_inner = mf.n_inner_face_batches();
        for (unsigned int face = n_inner; face < n_inner + mf.n_boundary_face_batches(); ++face)
          {
            if (cd.free_surface_boundary_indicators.count(mf.get_boundary_id(face)) == 0)
              continue;
            phi.reinit(face);
            for (unsigned int lane = 0; lane < mf.n_active_entries_per_face_batch(face); ++lane)
              {
                const auto cf = mf.get_face_iterator(face, lane, true);        // (cell, face number)
                face_cell.push_back(static_cast<uint32_t>(level_cells.back().at(cf.first->id())));
                face_number.push_back(static_cast<uint8_t>(cf.second));
                for (unsigned int q = 0; q < nqf; ++q)
                  {
                    for (unsigned int d = 0; d < dim; ++d)
                      stab.push_back(cd.free_surface_stabilization_term_table(face - n_inner, q)[d][lane]);
                    if (!cartesian)
                      {
                        for (unsigned int d = 0; d < dim; ++d)
                          normal.push_back(phi.normal_vector(q)[d][lane]);
                        jxw.push_back(phi.JxW(q)[lane]);
                      }
                  }
              }
          }
        fd.n_faces = static_cast<int32_t>(face_cell.size());
        fd.face_cell = face_cell.data();
        fd.face_number = face_number.data();
        fd.stabilization_term = stab.data();
        fd.normal = cartesian ? nullptr : normal.data();
        fd.jxw = cartesian ? nullptr : jxw.data();
        check(agmg_set_free_surface_faces(ctx, &fd), "agmg_set_free_surface_faces");     // collective
      }
  };

  // explicit instantiations, as the reference instantiates its handler (stokes_matrix_free_local_smoothing.cc, end of file)
  template class HipBackend<2, 2>;
  template class HipBackend<3, 2>;
}
