// agmg_internal.h -- device-side data layout and kernel launch interfaces of
// libaspect_amd.so (gfx950 only).  See DESIGN.md for the layout rationale.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/agmg.h"

#define AGMG_HIP_CHECK(expr)                                                        \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) {                                                         \
      agmg::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));           \
      return -100;                                                                  \
    }                                                                               \
  } while (0)

namespace agmg {

void set_error(const std::string &msg);

constexpr uint32_t NODE_MASK = 0x1FFFFFFFu;  // low 29 bits: node index; high 3 bits: constraint bits
constexpr uint32_t NO_CHILD = 0xFFFFFFFFu;   // host side of the transfer tables: the coarse cell is not refined (local smoothing)
#ifndef AGMG_WAVES_PER_BLOCK
#define AGMG_WAVES_PER_BLOCK 4
#endif
constexpr int WAVES_PER_BLOCK = AGMG_WAVES_PER_BLOCK;
// cell kernels: one lane per (cell, velocity component); 21 cells (3-D) or 32
// cells (2-D) per 64-lane wavefront
constexpr int cells_per_wave(int dim) { return 64 / dim; }
// doubles of LDS per wavefront in the block-scatter kernel: the slot tile [3^dim][64], which also
// hosts (aliased, earlier in the wave's life) the gradient-exchange tile [3][64*dim+2] and the
// per-q viscosity tile [3^dim][cells_per_wave]
// row stride of the slot tile in doubles: odd, so that the node-owner reads (which differ in the
// row a far more than in the lane) spread over the LDS banks
constexpr int BLK_SLOT_ROW = 65;
constexpr int LID_ROW = 32;   // uint16 entries per cell row of BlockTables::lidT (3^dim <= 27 used)
constexpr int blk_wave_tile(int dim) {
  const int nv = dim == 3 ? 27 : 9;
  const int slots = nv * BLK_SLOT_ROW, scratch = 3 * (64 * dim + 2) + nv * cells_per_wave(dim);
  return ((slots > scratch ? slots : scratch) + 7) / 8 * 8;
}

// epilogues of the pull (node-owner) kernels
enum Epilogue {
  EPI_STORE = 0,    // dst = A x                        (constrained rows: x, or 0 if x == nullptr)
  EPI_RESID = 1,    // dst = b - A x
  EPI_CHEB = 2,     // dst = (1+f1) x - f1 xold + f2 dinv (b - A x)   (dst may alias xold)
  EPI_ADD = 3,      // dst += sum  (constrained rows untouched)        [restriction]
  EPI_DIAG = 4,     // dst = 1 / (constrained ? 1 : sum)               [inverse diagonal]
  EPI_DIAG_RAW = 5, // dst = constrained ? 1 : sum
  EPI_SUB = 6       // dst -= sum  (constrained rows untouched)        [right-hand side correction]
};

struct CoeffData {     // OperatorCellData of one level (or of the active mesh)
  bool set = false;
  double *etaT = nullptr;  // [nqv][cstride]
  int nqv = 0;
  double lambda = 1.0;
  int is_compressible = 0;
  // Newton-derivative terms of the Stokes operator (active level only): tables [q][NS][cstride]
  // (symmetric tensors unrolled: diagonal, then 01, 02, 12) and [q][cstride]
  bool newton = false;
  int nw_symmetrize = 0;
  double *nw_strain = nullptr, *nw_dEde = nullptr, *nw_dEdp = nullptr;
  // prescribed dilation (active level only): LHS term [q][cstride], with Newton its derivatives
  bool dilation = false;
  double *dil_lhs = nullptr, *dil_de = nullptr, *dil_dp = nullptr;
  double *ell_val = nullptr;   // assembled mass rows [ell_w][ell_stride] of the level these coefficients belong to
  int ell_stride = 0;          // 0: not assembled
  // brick levels (kernels_brickq.hip): the viscosity rows and the cell extents in brick order,
  // [brick][layer][nqv + 3][64]; nullptr: not built (general geometry, no brick tables)
  double *brick_geo = nullptr;
  size_t bytes = 0;
};

// Block decomposition of the velocity nodes of one level ("block scatter"): cells are
// cut into blocks of `block_cells` = WAVES_PER_BLOCK * (64 / dim) consecutive cells (one
// workgroup); a block sums the contributions of its cells per node out of an LDS slot tile
// (up to 8 slots per node, fixed order), stores nodes no other block touches ("exclusive")
// straight to dst with the vector epilogue fused, and hands the partial sums of shared nodes to
// a fix-up kernel.
struct BlockTables {
  bool built = false;
  int block_cells = 0, n_blocks = 0, max_unodes = 0, n_shared = 0;
  size_t n_shared_slots = 0;
  uint32_t *blk_hdr = nullptr;    // [n_blocks][4]: unique nodes, exclusive nodes (they come first), first shared slot, 0
  uint32_t *unodes = nullptr;     // [n_blocks][max_unodes] global node | cmask << 29 of (block, local id)
  uint16_t *uent8 = nullptr;      // [n_blocks][max_unodes][8] LDS slot indices, padded with the zero slot
  uint16_t *lidT = nullptr;       // [cstride][LID_ROW] local id (within its block) of each cell node: one 64-byte
                                  // row per cell (four 16-byte loads per lane instead of 3^dim 2-byte ones)
  uint32_t *sh_nodes = nullptr;   // [n_shared] global node
  uint32_t *sh_ptr = nullptr;     // [n_shared + 1]
  uint32_t *sh_list = nullptr;    // slot positions, in block order; positions >= n_shared_slots address
                                  // the receive buffer of the ghost exchange (remote partial sums)
  int n_shared_owned = 0;         // shared nodes [0, n_shared_owned) are owned, the rest ghosts
  // distributed levels: blocks without a ghost node ("interior") first, then the blocks that read
  // ghost values ("interface"): the interior ones run while the ghost import is in flight
  // (deal.II's cell_loop overlaps the same way, matrix_free_operators.cc:357, 794)
  uint32_t *blk_order = nullptr;  // [n_blocks]
  int n_interior = 0;
  size_t bytes = 0;
};

// Brick decomposition of a structured 3-D level ("bricks"): the cells of the level are
// tiled by bricks of 8 x 8 x Z cells (Z layers of 64 cells); one workgroup marches a brick layer by
// layer (kernels_brick.hip).  A brick's (17 x 17 x (2Z+1)) lattice of velocity nodes is listed in
// `gnode`; a node that several bricks touch is owned by the first of them (brick order): the owner
// stores the row to dst (epilogue fused, with its own share of the sum), every other brick writes
// its share to a partial-sum slot, and a fix-up pass adds those shares to the owner's row.
constexpr uint32_t BRICK_NODE_MASK = 0x0FFFFFFFu;   // low 28 bits: node
constexpr uint32_t BRICK_PARTIAL = 0x10000000u;     // bit 28: this brick does not own the node
constexpr int BRICK_NX = 8, BRICK_LAT = 17, BRICK_PLANE = 17 * 17;
struct BrickTables {
  bool built = false;
  int Z = 0, n_bricks = 0, latn = 0, npb = 0;   // layers per brick, bricks, lattice nodes and surface slots per brick
  uint32_t *gnode = nullptr;    // [n_bricks][latn]: node | BRICK_PARTIAL | cmask << 29
  uint32_t *bcell = nullptr;    // [n_bricks][Z][64]: cell of (layer, lane = i + 8 j)
  uint16_t *pidx = nullptr;     // [latn]: surface enumeration of the lattice (0xFFFF inside)
  int n_fix = 0;                // owned nodes that receive partial sums
  uint32_t *fx_nodes = nullptr; // [n_fix] node | cmask << 29
  uint32_t *fx_ptr = nullptr;   // [n_fix + 1]
  uint32_t *fx_list = nullptr;  // partial slots (brick * npb + surface index), brick order
  size_t n_partial_slots = 0;   // n_bricks * npb (doubles needed: * 3)
  // distributed levels: a ghost node is owned by no brick (every brick writes its share to a partial slot); the sums
  // of the ghost nodes go to Exchange::ghostbuf for the compress step, and the fx lists of owned nodes that other
  // ranks ghost carry entries >= n_partial_slots addressing the receive buffer (remote partial sums)
  int n_interior = 0;           // bricks [0, n_interior) touch no ghost node: they run while the ghost import is in flight
  int n_gh = 0;
  uint32_t *gh_nodes = nullptr; // [n_gh] ghost node
  uint32_t *gh_ptr = nullptr;   // [n_gh + 1]
  uint32_t *gh_list = nullptr;  // partial slots, brick order
  size_t bytes = 0;
};

// Ghost exchange of one node space (Q2 velocity nodes or Q1 pressure nodes) of one level
// (deal.II update_ghost_values / compress(add) of LinearAlgebra::distributed::Vector).
// Ghost nodes sit behind the owned ones, grouped by owner rank; `send_nodes` lists per neighbour
// the owned nodes it ghosts, in the order of its ghost group.
struct Exchange {
  int nc = 1;                      // values per node (dim for velocity, 1 for pressure)
  int n_owned = 0, n_ghost = 0, n_export = 0;
  std::vector<int> nbr, send_off, recv_off;   // node units; [n_nbr], [n_nbr+1], [n_nbr+1]
  uint32_t *send_nodes = nullptr;  // device [n_export]
  double *expbuf = nullptr;        // device [n_export*nc]: packed owner values out / remote partial sums in
  double *ghostbuf = nullptr;      // device [n_ghost*nc]:  ghost partial sums out
  bool active() const { return !nbr.empty(); }
};

// Symmetric exchange of the pressure nodes of a distributed level (agmg_level_desc::sym_*): every rank that touches a
// node sends its share of a sum to every other rank that touches it; entry i of the message to neighbour k and entry i
// of the message from it belong to the same node, so one position table serves the send and the receive buffer.
struct SymExchange {
  std::vector<int> nbr, off;       // ascending ranks; [n_nbr + 1] offsets into the buffers
  int n_sym = 0, n_sh = 0;         // buffer entries; distinct shared local nodes
  uint32_t *sh_nodes = nullptr;    // [n_sh] local node
  uint32_t *ptr = nullptr;         // [n_sh + 1] into pos
  uint32_t *pos = nullptr;         // buffer positions of a shared node, ascending neighbour rank
  uint8_t *split = nullptr;        // [n_sh] how many of them belong to ranks below this one (rank-ordered sums)
  int32_t *row = nullptr;          // [n_nodes] index among the shared nodes, or -1
  double *sendbuf = nullptr, *recvbuf = nullptr, *ypart = nullptr;   // [n_sym], [n_sym], [n_sh]
  bool active() const { return !nbr.empty(); }
};

struct DeviceLevel {
  bool set = false;
  int dim = 3;
  int n_cells = 0, n_nodes_v = 0, n_nodes_p = 0;
  int n_owned_v = 0, n_owned_p = 0;  // owned nodes come first; == n_nodes_* on one rank
  bool replicated = false;           // distributed context: every rank holds this level whole (agmg_level_desc::replicated)
  Exchange xv, xp;
  SymExchange sp;                  // pressure nodes, optional
  bool sym_agreed = false;         // every rank can run the one-round mass application on this level (agmg_compute_diagonals)
  // the same for the velocity nodes (A block on the block path): buffers hold dim values per entry; sv_shidx[r] = index of
  // shared row r in the block tables' list of shared nodes, bt_symrow[i] = the inverse (-1: the node is shared between
  // blocks of this rank only)
  SymExchange sv;
  uint32_t *sv_shidx = nullptr;
  int32_t *bt_symrow = nullptr;
  bool sym_agreed_A = false;
  uint32_t *gid_v = nullptr, *gid_p = nullptr;  // global node index of each local node (nullptr: identity)
  int cstride = 0;                 // padded cell count (SoA row stride)
  // SoA tables: [local index][cstride]
  uint32_t *cnvT = nullptr;        // [3^dim][cstride]: node | cmask << 29
  uint32_t *cnpT = nullptr;        // [2^dim][cstride]: node | cmask << 29
  uint8_t *cmask_v = nullptr;      // [n_nodes_v]
  uint8_t *cmask_p = nullptr;      // [n_nodes_p]
  int geom = AGMG_GEOM_CARTESIAN;
  double *cell_hT = nullptr;       // CARTESIAN [dim][cstride]
  double *invJT = nullptr;         // GENERAL   [3^dim][dim*dim][cstride]
  double *jxwT = nullptr;          // GENERAL   [3^dim][cstride]
  // pull CSR (node -> contributing slots); entries a*cstride + cell
  uint32_t *pull_ptr_v = nullptr;  // [n_nodes_v + 1]
  uint32_t *pull_list_v = nullptr; // [n_cells * 3^dim]
  uint32_t *pull_ptr_p = nullptr;
  uint32_t *pull_list_p = nullptr; // [n_cells * 2^dim]
  BlockTables bt;
  BrickTables bk;
  CoeffData coeff;
  double *inv_diag_A = nullptr, *inv_diag_M = nullptr;
  // transfer, stored on the FINE level: per fine node the parent cell (on
  // level-1) and the position in the parent's (2k+1)^dim patch; per parent the
  // child cells; 1/multiplicity of fine nodes over parent patches
  uint32_t *prol_parent_v = nullptr; uint8_t *prol_pos_v = nullptr;
  uint32_t *prol_parent_p = nullptr; uint8_t *prol_pos_p = nullptr;
  double *inv_mult_v = nullptr, *inv_mult_p = nullptr;
  uint32_t *childrenT = nullptr;   // [2^dim][cstride of level-1]: child cell ids on this level
  uint32_t *prol_mask_v = nullptr; // [4][cstride of level-1]: bit p of a parent = it is the designated parent of the fine node at
                                   // position p of its (2k+1)^dim velocity patch (the per-parent prolongation writes only those)
  bool has_transfer = false;
  bool mass_fusable = false;       // the level fits the one-workgroup fused mass smoother
  // assembled rows of the Q1 mass operator: column structure, ELL [ell_w][ell_stride]
  // (unused entries point at the row's own node and carry the value 0); values live in CoeffData
  uint32_t *ell_col = nullptr;
  int ell_w = 0, ell_stride = 0;
  // column structure by class: on a structured numbering almost every row has the columns row + (one of a few offset
  // vectors); ell_cls[row] names the vector, ell_cls_off[class][ell_w] holds it.  nullptr: too many classes (an
  // arbitrary numbering), the kernel reads ell_col.  Saves 4 of the 12 bytes per entry the operator moves.
  uint16_t *ell_cls = nullptr;
  int32_t *ell_cls_off = nullptr;
  // distributed levels: per owned pressure node its row in the CSR of received shares (-1: nobody ghosts the node)
  int32_t *ell_xrow = nullptr;
  uint32_t *ell_xptr = nullptr, *ell_xent = nullptr;
  size_t bytes_tables = 0, bytes_transfer = 0, bytes_constraints = 0;
  int n_dofs_v() const { return dim * n_nodes_v; }
  int n_dofs_p() const { return n_nodes_p; }
};

struct EpiArgs {
  int epi = EPI_STORE;
  double *dst = nullptr;
  const double *x = nullptr;     // the vector A was applied to (identity rows, Chebyshev)
  const double *b = nullptr;
  const double *xold = nullptr;
  const double *dinv = nullptr;
  double f1 = 0, f2 = 0;
};

// one-launch Chebyshev sequences of the A block on single-block levels (k_cells_blk_seq)
enum { SEQ_POINTWISE = 0, SEQ_CHEB = 1, SEQ_RESID = 2, SEQ_MAX_STEPS = 12 };
struct SeqStep {
  int kind = SEQ_CHEB;
  double f1 = 0, f2 = 0;
  const double *x = nullptr, *xold = nullptr;
  double *dst = nullptr;
};
struct SeqArgs {
  int n = 0, n_dofs = 0;
  const double *b = nullptr, *dinv = nullptr;
  SeqStep s[SEQ_MAX_STEPS];
};
struct DeviceLevel;
struct CoeffData;
bool blk_seq_possible(const DeviceLevel &L, const CoeffData &cd);
int launch_op_seq(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, const SeqArgs &q, double *slots_u);

// ---- operator launches (kernels_ops.hip) ----
// cell kernels write per-cell contributions to `slots`; pull kernels sum them per DoF.
// which: AGMG_OP_STOKES (reads u,p; writes slots_u, slots_p), AGMG_OP_A (u -> slots_u),
// AGMG_OP_BT (p -> slots_u)
// plain: read the source without the constraints (right-hand side correction)
int launch_op_cells(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, int which,
                    const double *src_u, const double *src_p, double *slots_u, double *slots_p, bool plain = false);
int launch_mass_cells(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, const double *src_p,
                      double *slots_p);
int launch_diag_cells(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, double *slots_u,
                      double *slots_p);
// phase (distributed levels): PH_ALL everything at once (one rank); PH_GHOST the part before the
// ghost exchange (cell phase + partial sums of ghost nodes into Exchange::ghostbuf); PH_OWNED the
// part after it (sums of owned nodes including the received remote partial sums + epilogue)
// overlapped form of PH_GHOST: PH_INTERIOR = the cell phase of the blocks that read no ghost value
// (may run before the ghost import has landed), PH_INTERFACE = the remaining blocks + the partial sums
// of the ghost nodes
// PH_CELLS = the cell phase of every block and nothing else (the one-round form finishes the shared rows itself)
enum Phase { PH_ALL = 0, PH_GHOST = 1, PH_OWNED = 2, PH_INTERIOR = 3, PH_INTERFACE = 4, PH_CELLS = 5 };
int launch_pull_v(hipStream_t s, const DeviceLevel &L, const double *slots, const EpiArgs &e, int phase = PH_ALL);
// Whole velocity-row application: cell phase + reduction + epilogue `eu` on the velocity rows.
// Uses the block-scatter kernels when L.bt is built (and use_blocks), else cells -> slots -> pull.
// Stokes pressure rows are left in slots_p for launch_pull_p(..., dim, ...), or finished here with
// epilogue *ep (one rank: the velocity fix-up and the pressure pull share one launch).
// assembled Q1 mass rows: cd.ell_val = rows of lambda^2 (1/eta) mass matrix (set-up), and dst = epilogue(M src)
int launch_mass_ell_assemble(hipStream_t s, const DeviceLevel &L, const CoeffData &cd);
// phase (distributed levels): PH_GHOST = the ghost rows' raw sums into Exchange::ghostbuf, PH_OWNED = the owned rows with the
// received shares added before the epilogue
int launch_mass_ell(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, const double *src_p, const EpiArgs &e, int phase = 0);
// one message round per application (SymExchange): part 0 = the shares of the shared rows into the send buffer, part 1
// (after the exchange) = every local row, owned or ghost, with the epilogue; the shares of a shared row are added in
// rank order, so every rank that holds the row computes the same sum
int launch_mass_ell_sym(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, const double *src_p, const EpiArgs &e, int part);
// A block on the block path, one message round (DeviceLevel::sv): behind launch_op(..., PH_CELLS), part 0 = the shares of
// the rows other ranks hold too (sums of this rank's slots) into the send buffer, part 1 (after the exchange) = every
// shared row of the block tables, owned or ghost, with the epilogue; shares added in rank order
int launch_fix_sym(hipStream_t s, const DeviceLevel &L, const double *slots_u, const EpiArgs &eu, int part);

// free-surface stabilisation faces of the active Stokes operator (matrix_free_operators.cc:294-333),
// organised by node: every surface node lists the (face, local index) pairs that touch it
struct FaceTerm {
  int n_faces = 0, n_nodes = 0;
  uint32_t *nodes = nullptr;   // [n_nodes] surface nodes
  uint32_t *ptr = nullptr;     // [n_nodes + 1] into ent
  uint32_t *ent = nullptr;     // face << 4 | face-local node index
  uint32_t *fnode = nullptr;   // [n_faces][3^(dim-1)] nodes of each face
  double *stab = nullptr;      // [n_faces][nqf][dim] pressure_perturbation * g_hat
  double *njxw = nullptr;      // [n_faces][nqf][dim] outward normal * JxW
  // distributed contexts: some rank holds faces (agreed by all ranks at set-up); owned nodes other
  // ranks ghost, with their entries of the receive buffer (CSR), for the sums coming back
  bool global_active = false;
  int n_exp_nodes = 0;
  uint32_t *exp_nodes = nullptr, *exp_ptr = nullptr, *exp_ent = nullptr;
};
// dst_u += sign * (face term applied to src_u), unconstrained rows only.  ghost_out != nullptr
// (distributed): the sums of ghost nodes (index >= n_owned) go, unsigned, to
// ghost_out[(node - n_owned) * dim + c] for the compress step instead.
int launch_face_term(hipStream_t s, int dim, const FaceTerm &ft, const uint8_t *cmask_v, const double *src_u,
                     double *dst_u, double sign, double *ghost_out = nullptr, int n_owned = 0, bool plain = false);
// dst_u[node] += sign * (sum of the node's receive-buffer entries), fixed order, unconstrained rows
int launch_face_add_received(hipStream_t s, int dim, const FaceTerm &ft, const uint8_t *cmask_v, const double *recv,
                             double *dst_u, double sign);

// brick path (kernels_brick.hip): whole operator application on a level with BrickTables; partial sums go
// to `partials` (3 * n_partial_slots doubles).  ep (Stokes): also finish the pressure rows.
// phase (distributed levels, as launch_op): PH_GHOST = main kernel + sums of the ghost nodes into Exchange::ghostbuf
// (or PH_INTERIOR = the main kernel over the bricks without a ghost node, then PH_INTERFACE = the other bricks + the
// ghost sums), PH_OWNED = the fix-up of the owned rows including the received remote partial sums
int launch_op_bricks(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, int which, const double *src_u,
                     const double *src_p, double *partials, double *slots_p, const EpiArgs &eu, const EpiArgs *ep,
                     int phase = 0 /* PH_ALL */);
// quad-per-cell brick kernels (kernels_brickq.hip): main kernel only, the fix-up is launch_op_bricks'
bool brickq_supported(const DeviceLevel &L, const CoeffData &cd, int which);
size_t brick_geo_doubles(const DeviceLevel &L, int nqv);
int launch_brick_geo(hipStream_t s, const DeviceLevel &L, const CoeffData &cd);
// bricks [brick0, brick0 + n_launch) (n_launch < 0: all from brick0 on)
int launch_brickq(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, int which, const double *src_u, const double *src_p,
                  double *partials, double *slots_p, const EpiArgs &eu, int brick0 = 0, int n_launch = -1);
int launch_op(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, int which, const double *src_u,
              const double *src_p, double *slots_u, double *slots_p, const EpiArgs &eu, bool use_blocks,
              int phase = PH_ALL, const EpiArgs *ep = nullptr);
// n_parts: partial values per (cell, i) slot (dim for the Stokes pressure rows, 1 for mass)
int launch_pull_p(hipStream_t s, const DeviceLevel &L, const double *slots, int n_parts, const EpiArgs &e,
                  int phase = PH_ALL);

// ---- fused Chebyshev smoother of the Schur-complement mass operator on small levels -----------
// One workgroup runs a whole mg::SmootherRelaxation sequence (every Chebyshev step of every
// smoothing step, optionally the residual behind it) in ONE launch: per-cell coefficients and the
// node's vector entries stay in registers, the iterate lives in LDS.  The same arithmetic as the
// unfused path (k_mass + k_pull + epilogue), step by step.
struct SmootherStep { double f1, f2; int kind, use_xold; };   // kind: SM_POINTWISE x = f2*dinv*b | SM_CHEB | SM_RESID
enum { SM_POINTWISE = 0, SM_CHEB = 1, SM_RESID = 2 };
constexpr int SM_MAX_STEPS = 40;
struct SmootherProgram { int n = 0; SmootherStep s[SM_MAX_STEPS]; };
constexpr int MS_THREADS = 1024, MS_MAX_SLOTS = 4096, MS_MAX_NODES = 768;
// x_in == nullptr: the program starts with SM_POINTWISE; resid_out receives b - M x of an SM_RESID step
int launch_mass_smoother(hipStream_t s, const DeviceLevel &L, const CoeffData &cd, const SmootherProgram &prog,
                         const double *dinv, const double *b, const double *x_in, double *x_out, double *resid_out,
                         bool use_ell = false);   // use_ell: from the assembled rows (k_mass_smoother_ell)

// ---- transfer (kernels_transfer.hip) ----
int launch_prolongate_add(hipStream_t s, const DeviceLevel &fine, const DeviceLevel &coarse, bool velocity,
                          double *dst_fine, const double *src_coarse);
// writes coarse-level slots; follow with launch_pull_{v,p}(coarse, slots, EPI_ADD)
int launch_restrict_cells(hipStream_t s, const DeviceLevel &fine, const DeviceLevel &coarse, bool velocity,
                          const double *src_fine, double *slots);

// ---- in-library RCCL transport (transport_rccl.cpp) ----
int rccl_unique_id(uint8_t id[128]);
int rccl_comm_create(const uint8_t id[128], int rank, int size, agmg_comm *out, void **state);
void rccl_comm_destroy(void *state);

// ---- static device scratch (scratch.hip): for the entry point that allocates nothing ----
constexpr size_t STATIC_SCRATCH_BYTES = (size_t)64 << 20;   // per GPU, held until the process ends
int static_scratch_init(int device);      // once per device, that device current (agmg_create)
void *static_scratch_ptr(int device);     // device address of this GPU's copy, or nullptr

// ---- vector kernels (kernels_vec.hip) ----
int vec_set(hipStream_t s, double *x, double v, size_t n);
int vec_copy(hipStream_t s, double *dst, const double *src, size_t n);
int vec_scale(hipStream_t s, double *x, double a, size_t n);
int vec_axpy(hipStream_t s, double *y, double a, const double *x, size_t n);              // y += a x
int vec_sadd(hipStream_t s, double *y, double sy, double a, const double *x, size_t n);   // y = sy*y + a*x
int vec_lincomb(hipStream_t s, double *dst, double a, const double *x, double b, const double *y, size_t n);
int vec_pointwise_scale(hipStream_t s, double *dst, double f, const double *d, const double *b, size_t n);  // dst = f*d*b
int vec_zero_constrained(hipStream_t s, double *x, const uint8_t *cmask, int n_nodes, int ncomp);
int vec_fill_index_mod(hipStream_t s, double *x, size_t n, int mod);
// dst[q][cstride] = src[cell][nq] (cells >= n_cells: 1.0); *bad (device int) is set where !(value > 0)
int vec_transpose_table(hipStream_t s, double *dst, const double *src, size_t n_cells, int nq, size_t cstride, int *bad);
// x[node*nc + c] = (gid[node]*nc + c) % mod
int vec_fill_gid_mod(hipStream_t s, double *x, const uint32_t *gid, int n_nodes, int nc, int mod);
// out[k*nc + c] = x[nodes[k]*nc + c]
int vec_gather_nodes(hipStream_t s, double *out, const double *x, const uint32_t *nodes, int n, int nc);
int vec_shift(hipStream_t s, double *x, double a, size_t n);  // x += a
// hanging-node constraints of one node space on the device: the CSR hanging node -> (master, weight) as the caller
// gave it, its transpose master -> (hanging node, weight) in the same order of terms, and room for the entries of the
// hanging nodes as they came in (identity rows)
struct HangTables {
  int n_hang = 0, n_masters = 0;
  uint32_t *hn = nullptr, *hm = nullptr, *tm = nullptr, *th = nullptr;
  int32_t *hptr = nullptr, *tptr = nullptr;
  double *hw = nullptr, *tw = nullptr, *saved = nullptr;
};
int vec_hang_resolve(hipStream_t s, double *x, const HangTables &t, const uint8_t *cmask, int nc);
int vec_hang_distribute(hipStream_t s, double *y, const HangTables &t, const uint8_t *cmask, int nc);
int vec_hang_restore(hipStream_t s, double *y, const HangTables &t, int nc);
// dst[dst_nodes[k]*nc + c] = src[src_nodes[k]*nc + c]
int vec_copy_nodes(hipStream_t s, double *dst, const uint32_t *dst_nodes, const double *src, const uint32_t *src_nodes, int n_nodes, int nc);
constexpr int DOT_BLOCKS = 256;
constexpr int DOT_MAX_VECS = 16;   // (8 until round 3: a sweep over the Krylov basis re-reads the vector it is taken against once per chunk)
// out[k] = <x, ys[k]> for k < nvec <= DOT_MAX_VECS; result on device in `out`
// (workspace `partials` holds DOT_BLOCKS*DOT_MAX_VECS doubles)
struct VecPtrs { const double *p[DOT_MAX_VECS]; };
// logical entry i lives at i (i < seg1) or i + gap: skips the ghost section of a block vector
int vec_multi_dot(hipStream_t s, const double *x, const VecPtrs &ys, int nvec, size_t n, size_t seg1, size_t gap,
                  double *partials, double *out);
struct Coeffs { double a[DOT_MAX_VECS]; };
// y += sum_k a[k] * xs[k]
int vec_multi_axpy(hipStream_t s, double *y, const VecPtrs &xs, const Coeffs &a, int nvec, size_t n);

}  // namespace agmg
