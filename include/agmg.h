/* agmg.h -- C-ABI of the MI355X-native matrix-free block-GMG Stokes solve.
 *
 * Drop-in boundary for the hot path of tjhei/aspect (ASPECT 3.1.0-pre):
 *   include/aspect/simulator/solver/matrix_free_operators.h:91-507
 *       OperatorCellData, StokesOperator, BTBlockOperator, MassMatrixOperator,
 *       ABlockOperator
 *   include/aspect/simulator/solver/stokes_matrix_free.h:46-98, 113-354
 *       StokesMatrixFreeHandler, BlockSchurGMGPreconditioner
 *   source/simulator/solver/stokes_matrix_free_local_smoothing.cc:902-1544
 *       StokesMatrixFreeHandlerLocalSmoothingImplementation::solve
 *
 * The reference has no FFI for this path: the boundary is C++ virtual dispatch
 * plus deal.II's operator concept.  A thin C++ glue keeps those
 * classes intact and forwards to the entry points below, flattening deal.II
 * MatrixFree data into the tables declared here.
 *
 * Conventions
 *  - plain C, plain pointers and sizes; no exception crosses the ABI.
 *  - every int return is a status: 0 ok, 1 not converged, <0 error
 *    (message via agmg_last_error()).
 *  - one context per rank/GPU; calls are collective over the ranks of the
 *    context; not thread-safe.
 *  - host tables are copied during the call; the caller keeps ownership.
 *  - velocity DoF numbering: dof = dim*node + component ("node-major"),
 *    nodes being the Q2 support points in whatever (local) numbering the host
 *    chooses; pressure DoF = Q1 node.  Cell-local node order is lexicographic
 *    (x fastest), as in deal.II FEEvaluation.
 *  - all arithmetic is FP64 (GMGNumberType = double, stokes_matrix_free.h:40).
 */
#ifndef AGMG_H
#define AGMG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct agmg_ctx agmg_ctx;

/* which operator, for agmg_vmult (matrix_free_operators.h:208-507) */
enum { AGMG_OP_STOKES = 0, AGMG_OP_A = 1, AGMG_OP_BT = 2, AGMG_OP_MASS = 3 };
/* geometry kinds: MappingCartesian (core.cc:118-119) vs MappingQ (core.cc:115-116) */
enum { AGMG_GEOM_CARTESIAN = 0, AGMG_GEOM_GENERAL = 1 };
/* Krylov method for the cheap phase (parameters.cc:466-482) */
enum { AGMG_KRYLOV_GMRES = 0, AGMG_KRYLOV_IDR = 1 };
/* level index of the active mesh in set_level/set_coefficients/vmult */
#define AGMG_ACTIVE (-1)

/* Solver parameters; mirrors "Solver parameters/Stokes solver parameters"
 * (source/simulator/parameters.cc:432-584) and the constants hard-wired in
 * local_smoothing.cc:918-969. */
typedef struct agmg_options {
  int32_t dim;                   /* 2 or 3 */
  int32_t degree_v;              /* Stokes velocity polynomial degree; 2 supported */
  int32_t device;                /* HIP device ordinal */
  int32_t n_levels;              /* number of multigrid levels (n_global_levels) */
  double  linear_tolerance;      /* Linear solver tolerance, default 1e-7 */
  int32_t n_cheap_steps;         /* Number of cheap Stokes solver steps, default 1000 */
  int32_t n_expensive_steps;     /* Maximum number of expensive Stokes solver steps, 1000 */
  int32_t gmres_restart;         /* GMRES solver restart length, default 100 */
  int32_t krylov_type;           /* AGMG_KRYLOV_* */
  int32_t idr_s;                 /* IDR(s) parameter, default 2 */
  double  A_block_tolerance;     /* Linear solver A block tolerance, 1e-2 */
  double  S_block_tolerance;     /* Linear solver S block tolerance, 1e-6 */
  int32_t A_block_is_symmetric;  /* sim.stokes_A_block_is_symmetric() */
  int32_t coarse_on_host;        /* 1: level 0 smoother/solver runs on the host */
  /* Chebyshev smoother constants, local_smoothing.cc:926-937, 947-948, 954-965 */
  int32_t smoother_degree;       /* 4 */
  double  smoothing_range;       /* 15 */
  int32_t eig_cg_n_iterations;   /* 10 */
  int32_t coarse_degree;         /* 8 */
  double  coarse_range;          /* 1e-3 */
  int32_t coarse_eig_cg_n_iterations; /* 100 */
  int32_t schur_smoother_steps;  /* mg_smoother_Schur(4) */
  /* multi-GPU: world size/rank of this context (one rank <-> one GPU <-> one
   * subdomain).  world_size > 1 needs a transport before any table is set:
   * agmg_set_communicator_rccl (in-library RCCL) or agmg_set_communicator (host
   * callbacks).  nccl_id is unused (the id goes to agmg_set_communicator_rccl). */
  int32_t world_size;
  int32_t rank;
  uint8_t nccl_id[128];
} agmg_options;

/* Transport of a distributed context.  The reference runs this path over MPI
 * (LinearAlgebra::distributed::Vector::update_ghost_values / compress and
 * MPI_Allreduce inside deal.II's vector dot products); the
 * library needs exactly those two operations and takes them as callbacks so the
 * host can bind whatever it runs on: RCCL through torch.distributed
 * (aspect_amd/parallel.py), GPU-aware MPI in the ASPECT glue.
 *   exchange: n_msgs pairwise messages; message i sends n_send[i] doubles from
 *     send[i] to rank peer[i] and receives n_recv[i] doubles from it into
 *     recv[i] (either count may be 0).  Both sides list their messages to each
 *     other in the same order.
 *   allreduce_sum: in-place sum over all ranks of n doubles.
 * device_buffers = 1: pointers are device memory and the callback must order
 *   its work after everything already enqueued on `stream` and make later work
 *   on `stream` wait for it (no host synchronisation needed);
 * device_buffers = 0: the library stages through pinned host memory, the
 *   pointers are host memory, `stream` is NULL and the callback blocks.
 * Return 0 on success. */
typedef struct agmg_comm {
  int32_t rank, size;
  int32_t device_buffers;
  void *user;
  int (*exchange)(void *user, int32_t n_msgs, const int32_t *peer,
                  const double *const *send, const int64_t *n_send,
                  double *const *recv, const int64_t *n_recv, void *stream);
  int (*allreduce_sum)(void *user, double *values, int32_t n, void *stream);
} agmg_comm;

void agmg_default_options(agmg_options *opt);

/* One mesh level (an MG level 0..n_levels-1, or the active mesh AGMG_ACTIVE);
 * what MatrixFree + MGConstrainedDoFs + MGTransferMF hold for it
 * (local_smoothing.cc:1710-1877). */
typedef struct agmg_level_desc {
  int32_t n_cells;
  int32_t n_nodes_v;              /* locally relevant Q2 nodes (owned first, then ghosts) */
  int32_t n_nodes_p;              /* locally relevant Q1 nodes */
  int32_t n_owned_nodes_v;        /* == n_nodes_v on one rank */
  int32_t n_owned_nodes_p;
  const uint32_t *cell_nodes_v;   /* [n_cells][3^dim] */
  const uint32_t *cell_nodes_p;   /* [n_cells][2^dim] */
  const uint8_t  *constrained_v;  /* [n_nodes_v] bit c: component c is a homogeneous Dirichlet row */
  const uint8_t  *constrained_p;  /* [n_nodes_p] or NULL */
  int32_t geometry_kind;          /* AGMG_GEOM_* */
  const double *cell_h;           /* CARTESIAN: [n_cells][dim] cell extents */
  const double *inv_jacobian;     /* GENERAL:   [n_cells][3^dim][dim][dim], (J^-1)_{ed} = d xi_e / d x_d */
  const double *jxw;              /* GENERAL:   [n_cells][3^dim] */
  /* transfer to the next coarser level (NULL on level 0 and on the active mesh):
   * parent cell index on level-1 and child number (bit d: upper half in direction d) */
  const int32_t *parent_cell;     /* [n_cells] */
  const uint8_t *child_index;     /* [n_cells] */
  /* ghost exchange plan (world_size > 1), per neighbour rank; nodes are local indices */
  int32_t n_neighbors;
  const int32_t *neighbor_rank;   /* [n_neighbors] */
  const int32_t *send_offsets_v;  /* [n_neighbors+1] into send_nodes_v */
  const uint32_t *send_nodes_v;   /* owned nodes whose values the neighbour ghosts */
  const int32_t *recv_offsets_v;  /* [n_neighbors+1]; ghost node g of neighbour k lives at
                                     local index n_owned_nodes_v + recv_offsets_v[k] + g */
  const int32_t *send_offsets_p;
  const uint32_t *send_nodes_p;
  const int32_t *recv_offsets_p;
  /* global index of every local node (world_size > 1; NULL = the local index).
   * Only used for the start vector of the Chebyshev eigenvalue estimate
   * (SURVEY.md A.5), so that it does not depend on the partition. */
  const uint32_t *global_nodes_v; /* [n_nodes_v] */
  const uint32_t *global_nodes_p; /* [n_nodes_p] */
  /* Hanging-node constraints of an adaptively refined mesh (what AffineConstraints holds after
   * DoFTools::make_hanging_node_constraints, local_smoothing.cc:1597-1605, and what MatrixFree resolves
   * in read_dof_values / distribute_local_to_global): CSR over the hanging nodes, node h =
   * hanging_nodes_x[i] takes the value sum_k hanging_weight_x[k] * node hanging_master_x[k],
   * k in [hanging_ptr_x[i], hanging_ptr_x[i+1]); all components of a velocity node share the weights;
   * masters are not hanging themselves.  n_hanging_x == 0 (and NULL tables): none.
   * Accepted on the active mesh of an adaptive context (agmg_set_level(AGMG_ACTIVE, desc), below) and on
   * the level of a one-level, one-rank context.  The edge masks are what the level operators of local
   * smoothing need (MGConstrainedDoFs::get_refinement_edge_indices, :1804-1813): a level whose mask has a
   * node set runs its operator with those rows as identity rows (MatrixFreeOperators::Base) and the
   * interface operators (vmult_interface_down / _up, set_edge_matrices :1045, 1053) in the V-cycle.
   * One rank.  The constraints (resolve / transpose / identity rows) and the level copies are index
   * kernels around the device operators (kernels_vec.hip); agmg_rhs_correction, free-surface faces and
   * agmg_vmult_device_noalloc return -2 in adaptive contexts. */
  int32_t n_hanging_v, n_hanging_p;
  const uint32_t *hanging_nodes_v; const int32_t *hanging_ptr_v; const uint32_t *hanging_master_v; const double *hanging_weight_v;
  const uint32_t *hanging_nodes_p; const int32_t *hanging_ptr_p; const uint32_t *hanging_master_p; const double *hanging_weight_p;
  const uint8_t *refinement_edge_v;  /* [n_nodes_v] 1: node on the refinement edge of this level, or NULL */
  const uint8_t *refinement_edge_p;  /* [n_nodes_p] or NULL */
  /* Local smoothing on an adaptively refined mesh (the active mesh is then set with a descriptor of
   * its own, agmg_set_level(AGMG_ACTIVE, desc), and level l holds ALL cells of refinement level l):
   * MGLevelGlobalTransfer::copy_indices of the level: node copy_level_x[i] of this level is node
   * copy_active_x[i] of the active mesh (the nodes of the cells that are active on the level, its
   * refinement-edge nodes left out).  copy_to_mg fills the level defects through them, copy_from_mg
   * gathers the level solutions (local_smoothing.cc:1040-1058, PreconditionMG::vmult).  n_copy_x == 0
   * on every level: uniformly refined mesh, the active mesh is the finest level. */
  int32_t n_copy_v, n_copy_p;
  const uint32_t *copy_active_v; const uint32_t *copy_level_v;
  const uint32_t *copy_active_p; const uint32_t *copy_level_p;
  /* Distributed contexts (world_size > 1): 1 = every rank holds this level WHOLE, with the same tables in the
   * same (global) numbering, and no exchange plan (n_neighbors == 0, every node owned).  SURVEY.md 8e: "levels
   * <= 3 have <= 512 cells and go to the host or are gathered on one rank": here they are replicated instead, so
   * that nobody waits for anybody.  Operators, smoothers, eigenvalue estimates and the coarse solve of such a
   * level run as on one rank (no ghost exchange, no all-reduce; the one-launch smoother sequences, the assembled
   * mass rows and the host-resident level 0 apply); the restriction from the first distributed level onto the
   * last replicated one ends in ONE sum over ranks of the coarse defect (every rank restricts the cells it owns
   * on the finer level: parent_cell of that level indexes the whole coarser level, and coarse cells without a
   * child on this rank take no part), the prolongation back needs no message.  Replicated levels are the
   * levels 0..k for some k; the finest level is never replicated.  (No counterpart in the reference: deal.II
   * partitions every level.) */
  int32_t replicated;
  /* Optional symmetric exchange plan of the Q1 pressure nodes (distributed levels; no counterpart in deal.II, whose
   * vectors know owner -> ghost and ghost -> owner only): per rank that touches a node of this rank, in ascending rank
   * order, this rank's local nodes (owned or ghost) it touches too, ordered by global index, so that both sides list
   * the shared nodes in the same order.  With it the Chebyshev smoother of the Schur-complement mass operator needs ONE
   * message round per operator application instead of two (ghost import + compress): every rank that touches a node
   * receives the shares of all the others, adds them in rank order and applies the smoother update to its own copy of
   * the row, so the ghost copies of the iterate never go stale inside a smoother sequence.  n_sym_neighbors_p == 0:
   * not given, two rounds per application as in the reference. */
  int32_t n_sym_neighbors_p;
  const int32_t *sym_neighbor_rank_p;   /* [n_sym_neighbors_p], ascending */
  const int32_t *sym_offsets_p;         /* [n_sym_neighbors_p + 1] into sym_nodes_p */
  const uint32_t *sym_nodes_p;          /* local node indices */
  /* the same plan for the Q2 velocity nodes: the Chebyshev smoother of the A block in one message round per
   * application on the levels that run the block kernels (brick levels keep ghost import + compress) */
  int32_t n_sym_neighbors_v;
  const int32_t *sym_neighbor_rank_v;
  const int32_t *sym_offsets_v;
  const uint32_t *sym_nodes_v;
} agmg_level_desc;

/* Mirrors OperatorCellData<dim,double> (matrix_free_operators.h:91-203). */
typedef struct agmg_cell_data {
  int32_t is_compressible;
  int32_t enable_newton_derivatives;
  int32_t enable_prescribed_dilation;
  int32_t symmetrize_newton_system;
  double  pressure_scaling;
  int32_t n_q_viscosity;          /* second extent of the viscosity table: 1 or 3^dim */
  const double *viscosity;        /* [n_cells][n_q_viscosity] */
  /* Newton tables (active level only, local_smoothing.cc:447-663), or NULL.  Symmetric tensors
   * are unrolled as deal.II does: diagonal first, then (0,1), (0,2), (1,2). */
  const double *strain_rate_table;                 /* [n_cells][nq][dim(dim+1)/2] */
  const double *newton_factor_wrt_strain_rate;     /* [n_cells][nq][dim(dim+1)/2] */
  const double *newton_factor_wrt_pressure;        /* [n_cells][nq] */
  /* prescribed plastic dilation (active level only, local_smoothing.cc:503-510, 632-643): the
   * LHS term, and with Newton derivatives its two derivative tables; or NULL */
  const double *dilation_lhs_term;                       /* [n_cells][nq] */
  const double *dilation_derivative_wrt_strain_rate;     /* [n_cells][nq][dim(dim+1)/2] */
  const double *dilation_derivative_wrt_pressure;        /* [n_cells][nq] */
} agmg_cell_data;

/* Free-surface stabilisation faces of the active Stokes operator: the
 * apply_stabilization_free_surface_faces branch of OperatorCellData
 * (matrix_free_operators.h:91-203) with its free_surface_stabilization_term_table
 * (filled in local_smoothing.cc:665-756: pressure_perturbation * g_hat per face
 * q-point) and free_surface_boundary_indicators: the caller lists the boundary
 * faces that carry such an indicator.  Face q-points and face nodes are ordered
 * lexicographically over the tangential directions (lower direction fastest),
 * Gauss-3 per direction (face_quadratures.velocities). */
typedef struct agmg_face_data {
  int32_t n_faces;                  /* 0 switches the term off */
  const uint32_t *face_cell;        /* [n_faces] cell of the active mesh */
  const uint8_t  *face_number;      /* [n_faces] deal.II face number 2*direction + side */
  const double *stabilization_term; /* [n_faces][3^(dim-1)][dim] */
  const double *normal;             /* [n_faces][3^(dim-1)][dim] outward unit normals, or NULL
                                       (Cartesian geometry: +-e_direction) */
  const double *jxw;                /* [n_faces][3^(dim-1)] face JxW, or NULL with normal == NULL */
} agmg_face_data;

/* Mirrors the screen/statistics output of solve()
 * (local_smoothing.cc:1421-1425, 1526-1536; SolverOutputs interface.h:36-58). */
typedef struct agmg_solve_report {
  int32_t n_cheap_iterations;
  int32_t n_expensive_iterations;
  int32_t n_A_preconditioner_cheap;      /* "A block preconditioner: a+b" */
  int32_t n_A_preconditioner_expensive;
  int32_t n_S_preconditioner_cheap;
  int32_t n_S_preconditioner_expensive;
  double  initial_nonlinear_residual;
  double  final_linear_residual;
  double  solver_tolerance;              /* absolute tolerance used */
  double  solve_seconds;                 /* wall time of the on-device solve */
  double  setup_seconds;                 /* smoother/eigenvalue setup inside solve() */
  int32_t n_history;                     /* entries written to history */
  double  history[2048];                 /* residual per outer iteration (cheap then expensive) */
} agmg_solve_report;

const char *agmg_last_error(void);

/* create_matrix_free_solver + constructor (stokes_matrix_free.cc:29-62) */
agmg_ctx *agmg_create(const agmg_options *opt);
void      agmg_destroy(agmg_ctx *ctx);

/* Bind the transport of a distributed context (opt.world_size > 1); the struct
 * is copied, the callbacks and `user` must outlive the context.  Replaces the
 * MPI communicator the reference takes from SimulatorAccess
 * (get_mpi_communicator(), used throughout local_smoothing.cc).  In a
 * distributed context every vector crossing this ABI is the rank-LOCAL vector
 * (owned entries first, ghost entries behind them, as in agmg_level_desc):
 * owned entries of inputs must be valid, ghost entries are refreshed
 * internally (the library writes them, also in `src` arguments on the
 * device); only owned entries of outputs are meaningful. */
int agmg_set_communicator(agmg_ctx *ctx, const agmg_comm *comm);

/* In-library transport: RCCL (xGMI inside a node) called directly on the
 * context's stream: one grouped ncclSend/ncclRecv launch per ghost exchange, one
 * ncclAllReduce per fused dot product, no host synchronisation.  Rank 0 obtains
 * an id with agmg_rccl_get_unique_id and hands the 128 bytes to every rank by any
 * means (the reference would MPI_Bcast them); agmg_set_communicator_rccl is then
 * collective over the ranks of the context (ncclCommInitRank), with the
 * context's device current.  Takes the place of agmg_set_communicator. */
int agmg_rccl_get_unique_id(uint8_t id[128]);
int agmg_set_communicator_rccl(agmg_ctx *ctx, const uint8_t id[128]);

/* setup_dofs(): upload the tables of one level (local_smoothing.cc:1548-1878).
 * level in [0, n_levels) or AGMG_ACTIVE. desc == NULL for AGMG_ACTIVE aliases
 * the active mesh to the finest level (uniformly refined mesh, identical
 * numbering); a descriptor of its own makes the context adaptive: the levels
 * then hold all cells of their refinement level (set them first, with edge masks
 * and copy tables), the vectors of vmult / vcycle / precondition / solve are
 * those of the active mesh. */
int agmg_set_level(agmg_ctx *ctx, int level, const agmg_level_desc *desc);

/* evaluate_material_model() -> OperatorCellData per level
 * (local_smoothing.cc:172-445; set_cell_data matrix_free_operators.cc:151, 381, 511, 698) */
int agmg_set_coefficients(agmg_ctx *ctx, int level, const agmg_cell_data *data);

/* The boundary-face term of StokesOperator (local_apply_boundary_face,
 * matrix_free_operators.cc:294-333; apply_add :337-358): dst_u += sum over the
 * listed faces of integral phi_i . (-(g . u) n).  Active Stokes operator only
 * (the reference has no multilevel surface terms, local_smoothing.cc:668-670);
 * data == NULL or n_faces == 0 removes it.  In a distributed context the call
 * is collective: every rank passes the faces of its own cells (possibly none);
 * sums that belong to rows of other ranks travel in one more compress step per
 * operator application. */
int agmg_set_free_surface_faces(agmg_ctx *ctx, const agmg_face_data *data);

/* build_preconditioner(): ABlockOperator/MassMatrixOperator::compute_diagonal on
 * every level (local_smoothing.cc:1882-1894; matrix_free_operators.cc:592-639, 799-829) */
int agmg_compute_diagonals(agmg_ctx *ctx);

/* {Stokes,ABlock,BTBlock,MassMatrix}Operator::vmult on host vectors
 * (matrix_free_operators.cc:171-278, 399-432, 518-576, 760-783; wrapper
 * semantics of deal.II MatrixFreeOperators::Base::vmult).
 * STOKES/BT: src,dst = [n_dofs_v + n_dofs_p]; A: [n_dofs_v]; MASS: [n_dofs_p]. */
int agmg_vmult(agmg_ctx *ctx, int which, int level, double *dst, const double *src);
/* same on device pointers, asynchronous on the context's stream */
int agmg_vmult_device(agmg_ctx *ctx, int which, int level, double *dst, const double *src);

/* correct_stokes_rhs() (local_smoothing.cc:761-898): rhs += -M u0 on the active mesh, host vectors
 * [n_dofs_v + n_dofs_p].  u0 is the zero vector with the inhomogeneous boundary values distributed
 * (constraints.distribute(u0), :775-787); it is read WITHOUT the constraints
 * (read_dof_values_plain, :813, 817); rows of constrained DoFs receive nothing
 * (integrate_scatter); no Newton terms; compressible / dilation models keep the
 * deviatoric term only (:835-838); free-surface faces with plain reads (:853-888).
 * Call after agmg_set_coefficients(AGMG_ACTIVE or the finest level) once per assemble. */
int agmg_rhs_correction(agmg_ctx *ctx, const double *u0, double *rhs);

/* agmg_vmult_device for callers that cannot allocate: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart
 * in the reference.  Same arguments and results as agmg_vmult_device; the slot /
 * partial-sum buffers of the kernels live in a static 64 MiB device array of the
 * library (one copy per GPU, held until the process ends) instead of the
 * context's workspace, and no hipMalloc, hipFree or allocating routine is called.
 * The need follows the kernels the level runs: brick partial sums or
 * block-boundary slots (a 3-D box of 64^3 cells: 19 MB) or, on the slot path, one
 * slot per cell dof (170 MB there).  Levels that need more than the array holds are
 * rejected (status -3, the message names both sizes); one-rank contexts only (-2
 * otherwise).  All calls on
 * one GPU share the array: calls that overlap on one GPU race. */
int agmg_vmult_device_noalloc(agmg_ctx *ctx, int which, int level, double *dst, const double *src);

/* get_matrix_diagonal_inverse() of the level A / mass operators, copied to host */
int agmg_get_inverse_diagonal(agmg_ctx *ctx, int which, int level, double *dst);

/* MGTransferMF::prolongate_and_add / restrict_and_add for the velocity (which =
 * AGMG_OP_A) or pressure (AGMG_OP_MASS) space, host vectors.
 * prolongate: dst (level to_level) += P src (level to_level-1);
 * restrict:   dst (level from_level-1) += P^T src (level from_level). */
int agmg_prolongate_add(agmg_ctx *ctx, int which, int to_level, double *dst, const double *src);
int agmg_restrict_add(agmg_ctx *ctx, int which, int from_level, double *dst, const double *src);

/* Build the Chebyshev smoothers and estimate their eigenvalues on every level
 * (local_smoothing.cc:918-993).  Called implicitly by agmg_solve; explicit for
 * agmg_vcycle/agmg_precondition. */
int agmg_setup_smoothers(agmg_ctx *ctx);
/* eigenvalue estimates (max after the 1.2 safety factor, min as used) */
int agmg_get_eigenvalue_estimates(agmg_ctx *ctx, int which, int level, double *lambda_min, double *lambda_max);

/* PreconditionMG::vmult for the A block (AGMG_OP_A) or the Schur complement
 * mass matrix (AGMG_OP_MASS): one V-cycle, active-mesh host vectors
 * (local_smoothing.cc:1040-1058) */
int agmg_vcycle(agmg_ctx *ctx, int which, double *dst, const double *src);

/* BlockSchurGMGPreconditioner::vmult (stokes_matrix_free.h:237-354), host
 * vectors [n_dofs_v + n_dofs_p]; expensive != 0 selects do_solve_A/do_solve_Schur */
int agmg_precondition(agmg_ctx *ctx, int expensive, double *dst, const double *src);

/* StokesMatrixFreeHandler::solve (local_smoothing.cc:902-1544), steps 1-7 of
 * x holds the initial guess (velocity, scaled pressure) on entry
 * and the solution on return; constrained entries of x are zeroed as
 * current_constraints.set_zero does.  solve_newton_system selects the Newton
 * tolerance branch (:1169-1183). */
int agmg_solve(agmg_ctx *ctx, const double *rhs, double *x, int solve_newton_system,
               agmg_solve_report *report);

/* device-resident benchmark helpers: vectors live in the context
 *   agmg_bench_vmult: n_repeat vmults of `which` on `level`, returns mean
 *   seconds per vmult measured with HIP events on the context's stream
 *   (the in-solver harness local_smoothing.cc:1234-1375). */
int agmg_bench_vmult(agmg_ctx *ctx, int which, int level, int n_repeat, double *seconds_per_vmult);

/* The reference's in-solver micro-benchmark ("Solver parameters/Matrix Free/Execute solver
 * timings", local_smoothing.cc:57-62, 1234-1375), device-resident: one warm-up call, then the mean
 * wall time of n_repeat calls of each section.  seconds[0] stokes_vmult, [1] stokes_preconditioner
 * (cheap: 2 V-cycles + B^T), [2] A_preconditioner (one V-cycle), [3] S_preconditioner (one V-cycle),
 * [4] A_block_vmult, [5] set-up of the smoothers (eigenvalue estimates on every level: what the
 * reference redoes inside every solve(), :918-1058).  Sections 0-3 are the reference's; 4 and 5 are
 * this library's own (the reference times Stokes_solve_cheap_idr and Stokes_solve_cheap_gmres there,
 * :1341-1374: time agmg_solve for those). */
int agmg_execute_solver_timings(agmg_ctx *ctx, int n_repeat, double seconds[6]);

/* Which kernels apply the velocity-row operators of a level (diagnostics and benchmark reports; the
 * reference's counterpart is the "Output details" block of local_smoothing.cc:1007-1017): bricks of
 * 8 x 8 x layers cells (structured 3-D levels), else blocks of consecutive cells, else (both 0) the
 * slot path. */
int agmg_get_level_kernel_path(const agmg_ctx *ctx, int level, int32_t *n_bricks, int32_t *layers_per_brick, int32_t *n_blocks);

/* Distributed contexts: how a level's blocks split into interior blocks (no ghost node: they run
 * while the ghost import of the source vector is in flight, as deal.II's cell_loop overlaps the
 * import with its interior cell batches, matrix_free_operators.cc:357, 794) and interface blocks,
 * and how many operator applications of the context have taken the overlapped path so far. */
int agmg_get_overlap_statistics(const agmg_ctx *ctx, int level, int32_t *n_interior_blocks, int32_t *n_interface_blocks,
                                int32_t *n_overlapped_applications);

/* memory getters (stokes_matrix_free.h:75-98; postprocess/memory_statistics.cc:44-59) */
size_t agmg_get_cell_data_memory_consumption(const agmg_ctx *ctx);
size_t agmg_get_mg_transfer_memory_consumption(const agmg_ctx *ctx);
size_t agmg_get_constraint_memory_consumption(const agmg_ctx *ctx);
size_t agmg_get_dof_handler_memory_consumption(const agmg_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* AGMG_H */
