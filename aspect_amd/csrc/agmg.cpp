// agmg.cpp -- host side of libaspect_amd.so: the C-ABI of include/agmg.h.
//
// Mirrors, for the hot path only, what the reference keeps in
//   StokesMatrixFreeHandlerLocalSmoothingImplementation (setup_dofs / evaluate_material_model /
//   build_preconditioner / solve, stokes_matrix_free_local_smoothing.cc:172-445, 902-1544, 1548-1894)
//   BlockSchurGMGPreconditioner::vmult              (stokes_matrix_free.h:237-354)
// and the deal.II pieces that path instantiates (PreconditionChebyshev, Multigrid::level_v_step,
// mg::SmootherRelaxation, PreconditionMG, SolverGMRES/FGMRES/IDR/CG).
// All vectors of the solve live in HBM; the host only sees scalars (dot products, Givens
// rotations) and -- by default, as north_star asks -- the coarsest multigrid level.
#include <algorithm>
#include <array>
#include <unordered_map>
#include <thread>
#include <vector>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <random>

#include "agmg_internal.h"

namespace agmg {
static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
}  // namespace agmg

using namespace agmg;

#define CHECK(x)            \
  do {                      \
    const int rc_ = (x);    \
    if (rc_ != 0) return rc_; \
  } while (0)

namespace {

inline int ipow(int b, int e) { int r = 1; while (e--) r *= b; return r; }
inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct HostLevel {  // host copies kept for building the transfer tables
  std::vector<uint32_t> cnv, cnp;
  std::vector<int32_t> parent;
  std::vector<uint8_t> child;
  // hanging-node constraints (agmg_level_desc; first slice: resolved on the host around agmg_vmult): CSR over the
  // hanging nodes of the velocity [0] and pressure [1] spaces, and the constraint masks of the two spaces
  std::vector<uint32_t> hang_n[2], hang_m[2];
  std::vector<int32_t> hang_ptr[2];
  std::vector<double> hang_w[2];
  std::vector<uint8_t> cmask[2];
  bool has_hanging() const { return !hang_n[0].empty() || !hang_n[1].empty(); }
};

struct Smoother {  // PreconditionChebyshev state
  double lmin = 0, lmax = 0, theta = 1, delta = 0;
  int degree = 4;
};

struct CoarseOp {  // host-resident level 0: dense operator + Jacobi-Chebyshev "solver"
  int n = 0;
  std::vector<double> A, dinv;
};

// extreme eigenvalues of a symmetric tridiagonal matrix by Sturm-sequence bisection
void tridiag_extreme_eigs(const std::vector<double> &d, const std::vector<double> &e, double &emin, double &emax) {
  const int n = (int)d.size();
  double lo = std::numeric_limits<double>::max(), hi = -lo;
  for (int i = 0; i < n; ++i) {
    const double r = (i > 0 ? std::fabs(e[i - 1]) : 0.0) + (i + 1 < n ? std::fabs(e[i]) : 0.0);
    lo = std::min(lo, d[i] - r);
    hi = std::max(hi, d[i] + r);
  }
  auto count_below = [&](double x) {  // number of eigenvalues < x
    int cnt = 0;
    double q = 1.0;
    for (int i = 0; i < n; ++i) {
      const double off2 = i > 0 ? e[i - 1] * e[i - 1] : 0.0;
      q = (d[i] - x) - (i > 0 ? off2 / q : 0.0);
      if (q == 0.0) q = 1e-300;
      if (q < 0) ++cnt;
    }
    return cnt;
  };
  auto kth = [&](int k) {  // k-th smallest eigenvalue (0-based)
    double a = lo, b = hi;
    for (int it = 0; it < 200; ++it) {
      const double m = 0.5 * (a + b);
      if (m <= a || m >= b) break;
      if (count_below(m) > k) b = m; else a = m;
    }
    return 0.5 * (a + b);
  };
  emin = kth(0);
  emax = kth(n - 1);
}

// builds the CG tridiagonal incrementally (deal.II SolverCG eigenvalue signal)
struct Lanczos {
  std::vector<double> diag, off;
  double alpha_old = 0;
  void push(int it, double alpha, double beta) {
    if (it == 1) diag.push_back(1.0 / alpha);
    else { diag.push_back(1.0 / alpha + beta / alpha_old); off.push_back(std::sqrt(beta) / alpha_old); }
    alpha_old = alpha;
  }
};

void finish_smoother(Smoother &sm, const Lanczos &lz, double range) {
  double emin = 1.0, emax = 1.0;
  if (!lz.diag.empty()) {
    double a, b;
    tridiag_extreme_eigs(lz.diag, lz.off, a, b);
    emin = a;
    emax = 1.2 * b;
  }
  sm.lmax = emax;
  const double alpha = range > 1.0 ? emax / range : std::min(0.9 * emax, emin);
  sm.lmin = alpha;
  sm.delta = 0.5 * (emax - alpha);
  sm.theta = 0.5 * (emax + alpha);
}

enum { ST_SUCCESS = 0, ST_ITERATE = 1, ST_FAILURE = 2 };
struct Control {  // deal.II SolverControl(max_steps, tol, log_history)
  int max_steps;
  double tol;
  int last_step = 0;
  double last_value = 0;
  std::vector<double> history;
  int check(int step, double v) {
    last_step = step;
    last_value = v;
    history.push_back(v);
    if (v <= tol) return ST_SUCCESS;
    if (step >= max_steps || !std::isfinite(v)) return ST_FAILURE;
    return ST_ITERATE;
  }
};

}  // namespace

struct agmg_ctx {
  agmg_options opt;
  hipStream_t stream = nullptr;
  std::vector<DeviceLevel> L;
  std::vector<HostLevel> H;
  bool has_hanging() const { for (const auto &h : H) if (h.has_hanging()) return true; return false; }
  // ---- local smoothing on adaptively refined meshes: the active mesh has tables of its own (its
  // hanging nodes are resolved around the device operators), every level carries its copy table, and a level with a
  // refinement edge exists twice: L[l] with boundary + edge rows constrained (the level operator of
  // MatrixFreeOperators::Base), ad.Lb[l] with the boundary rows alone (the interface operators)
  struct Adaptive {
    bool on = false;
    DeviceLevel act;
    HostLevel hact;                                     // (with the hanging-node CSRs of the active mesh)
    std::vector<DeviceLevel> Lb;
    std::vector<HostLevel> Hb;
    std::vector<uint8_t *> edge[2], notedge[2];        // [velocity | pressure][level] device masks, nullptr without an edge
    std::vector<std::vector<uint32_t>> copy_act[2], copy_lev[2];   // [velocity | pressure][level] MGLevelGlobalTransfer::copy_indices
    std::vector<uint32_t *> copy_act_d[2], copy_lev_d[2];          // the same tables on the device
    HangTables hang[2];                                            // hanging-node CSRs of the active mesh on the device
    // pull tables of level l - 1 that leave out the cells without children (restriction from level l), [space][l]
    std::vector<uint32_t *> rpull_ptr[2], rpull_list[2];
    double *xres = nullptr;      // a source vector with its hanging entries resolved [n_u + n_p]
    size_t xres_cap = 0;
  } ad;
  bool has_edge(int l) const { return ad.on && l >= 0 && l < (int)ad.Lb.size() && ad.Lb[l].set; }
  void free_adaptive_level(int l);
  void free_adaptive_active();
  // operators of the active mesh with the hanging nodes resolved: dst = op(src), or b - op(src) with b given
  int active_apply(int which, double *dst, const double *src, const double *b);
  int apply_on(const DeviceLevel &lv, int which01, const CoeffData &cd, const double *x, const EpiArgs &e);
  int interface_up_into_defect(int which, int l);
  int vcycle_adaptive(int which, double *dst, const double *src);
  CoeffData active;
  FaceTerm faces;   // free-surface stabilisation faces of the active Stokes operator
  void free_faces() {
    dfree(faces.nodes); dfree(faces.ptr); dfree(faces.ent); dfree(faces.fnode); dfree(faces.stab); dfree(faces.njxw);
    dfree(faces.exp_nodes); dfree(faces.exp_ptr); dfree(faces.exp_ent);
    faces = FaceTerm();
  }
  double *slots_u = nullptr, *slots_p = nullptr;
  size_t slots_u_cap = 0, slots_p_cap = 0;
  // multigrid work vectors, [0] = A block (velocity), [1] = Schur complement (pressure)
  std::vector<double *> mg_def[2], mg_sol[2], mg_tmp[2], mg_t[2];
  std::vector<Smoother> sm[2];
  CoarseOp coarse[2];
  bool mg_alloc = false, diag_ready = false, smoothers_ready = false;
  double *dot_partials = nullptr, *dot_out_d = nullptr, *dot_out_h = nullptr;
  std::vector<double *> pool;  // Krylov work vectors of size n_u + n_p
  std::vector<double *> bicg_pool;  // three more velocity vectors for BiCGStab on a non-symmetric A block
  int nA = 0, nS = 0;          // V-cycle / inner iteration counters of the current preconditioner
  std::vector<double> h0, h1;  // host staging for the coarse level
  // pinned staging of the host-resident coarse solve, per block: [0, n) the defect, [n, 2n) the solution.  The solve runs
  // as a host function in stream order (hipLaunchHostFunc): no synchronisation, and the V-cycle stays capturable
  double *coarse_pin[2] = {nullptr, nullptr};
  size_t coarse_pin_cap[2] = {0, 0};
  struct CoarseJob { agmg_ctx *c; int which; } coarse_job[2];
  void coarse_host_compute(int which, const double *b, double *xout) const;
  // one hipGraph per V-cycle (A block, Schur block): the cycle is a fixed sequence of some hundred small launches on
  // fixed buffers.  Opt-in (AGMG_GRAPHS=1): measured on MI355X the replay saves 3 % of the A-block cycle and 10 % of the
  // Schur cycle at r5 (the cycles are bound by the dependent kernels' own latency, not by the host's launches), and
  // rocprofv3 --kernel-trace of ROCm 7.2 segfaults on a process that captures (profiles/README.md).  state: 0 = not run yet (the
  // first cycle runs eagerly: lazy allocations and kernel attributes), 1 = capture at the next cycle, 2 = replay,
  // -1 = capture failed, eager from then on
  bool use_graphs = false;
  struct VGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int state = 0; } vgraph[2];
  void free_graphs() {
    for (auto &g : vgraph) {
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
      if (g.graph) (void)hipGraphDestroy(g.graph);
      g = VGraph();
    }
  }

  // ---- distributed contexts: transport callbacks + host staging
  agmg_comm comm{};
  bool comm_set = false;
  void *rccl_state = nullptr;   // in-library RCCL transport (transport_rccl.cpp), or nullptr
  double *stage_h = nullptr;   // pinned; device_buffers == 0
  size_t stage_cap = 0;
  // (a 1-rank context with a bound transport runs the distributed code paths too: tests)
  bool dist() const { return opt.world_size > 1 || comm_set; }
  // a level of a distributed context that is really partitioned (a replicated level, agmg_level_desc::replicated, is held
  // whole by every rank and runs the one-rank code paths)
  bool ldist(const DeviceLevel &lv) const { return dist() && !lv.replicated; }
  // in-place sum over ranks of a device vector (the restriction onto the last replicated level)
  int allreduce_vec(double *d, size_t n);
  struct Msg { int peer; const double *send; int64_t n_send; double *recv; int64_t n_recv; };
  int exchange(const std::vector<Msg> &msgs);
  // owner -> ghost copies of `vec` (nc values per node, nc <= x.nc)
  int add_update(std::vector<Msg> &msgs, const Exchange &x, double *vec, int nc);
  // ghost partial sums (x.ghostbuf) -> owners (x.expbuf)
  void add_compress(std::vector<Msg> &msgs, const Exchange &x, int nc);
  // ---- overlap of the ghost import with the interior blocks: the import runs on a second stream
  hipStream_t cstream = nullptr;
  hipEvent_t ev_src_ready = nullptr, ev_import_done = nullptr;
  bool use_overlap = true;    // AGMG_NO_OVERLAP=1: import, then all blocks (the order of round 1)
  int n_overlapped = 0;       // operator applications that took the overlapped path (tests)
  bool can_overlap(const DeviceLevel &lv, const EpiArgs &e, const CoeffData &cd, int which) const {
    return use_overlap && use_blocks && lv.bt.built && lv.bt.blk_order && lv.bt.n_interior > 0 && cstream &&
           (e.epi == EPI_STORE || e.epi == EPI_RESID || e.epi == EPI_CHEB) &&
           !((cd.newton || cd.dilation) && which == AGMG_OP_STOKES);
  }
  // run `import` (the ghost update of the source vector) on the communication stream, behind everything
  // already enqueued on the main stream; returns with the main stream free to run the interior blocks
  int begin_import(const std::function<int()> &import) {
    AGMG_HIP_CHECK(hipEventRecord(ev_src_ready, stream));
    AGMG_HIP_CHECK(hipStreamWaitEvent(cstream, ev_src_ready, 0));
    std::swap(stream, cstream);
    const int rc = import();
    std::swap(stream, cstream);
    if (rc) return rc;
    AGMG_HIP_CHECK(hipEventRecord(ev_import_done, cstream));
    return 0;
  }
  int end_import() {
    AGMG_HIP_CHECK(hipStreamWaitEvent(stream, ev_import_done, 0));
    ++n_overlapped;
    return 0;
  }
  int update_ghosts(const Exchange &x, const double *vec) {
    if (!x.active()) return 0;
    std::vector<Msg> m;
    CHECK(add_update(m, x, const_cast<double *>(vec), x.nc));
    return exchange(m);
  }
  int compress(const Exchange &x) {
    if (!x.active()) return 0;
    std::vector<Msg> m;
    add_compress(m, x, x.nc);
    return exchange(m);
  }
  int allreduce_host(double *v, int n);
  // A vector of the solve: n local entries (owned + ghost); dot products run over the owned
  // entries only: logical entry i < n_dot lives at i (i < seg1) or at i + gap
  // local: the vector lives on a replicated level, every rank holds all of it and computes the same sums (no all-reduce)
  struct Span { size_t n, n_dot, seg1, gap; bool local = false; };
  Span span_of(int which, int l) const {
    const DeviceLevel &lv = L[l];
    if (which == 0) { const size_t o = (size_t)lv.n_owned_v * lv.dim; return {(size_t)lv.n_dofs_v(), o, o, 0, lv.replicated}; }
    return {(size_t)lv.n_dofs_p(), (size_t)lv.n_owned_p, (size_t)lv.n_owned_p, 0, lv.replicated};
  }
  Span span_up() const {
    const DeviceLevel &lv = ad.on ? ad.act : L.back();
    const size_t nu = lv.n_dofs_v(), ou = (size_t)lv.n_owned_v * lv.dim;
    return {nu + (size_t)lv.n_dofs_p(), ou + (size_t)lv.n_owned_p, ou, nu - ou};
  }

  int nl() const { return (int)L.size(); }
  // level 0 lives on the host (dense operator + Jacobi-Chebyshev) while it is small: one coarse cell
  // per tree of a box is 81 + 8 DoFs; the 96-cell shell (2910 + 150) would spend the V-cycle in a
  // 2910^2 host matvec, so larger coarse levels run the same smoother on the device
  static constexpr size_t COARSE_HOST_MAX_DOFS = 600;
  bool coarse_host(int which) const { return opt.coarse_on_host && !L.empty() && !ldist(L[0]) && size_of(which, 0) <= COARSE_HOST_MAX_DOFS; }
  DeviceLevel &fin() { return L.back(); }
  // the active mesh: the finest level (uniform refinement) or the tables set with agmg_set_level(AGMG_ACTIVE, desc)
  DeviceLevel &act() { return ad.on ? ad.act : L.back(); }
  const DeviceLevel &act() const { return ad.on ? ad.act : L.back(); }
  const CoeffData &active_coeff() { return active.set ? active : fin().coeff; }
  Span span_act(int which) const {
    const DeviceLevel &lv = act();
    if (which == 0) { const size_t o = (size_t)lv.n_owned_v * lv.dim; return {(size_t)lv.n_dofs_v(), o, o, 0}; }
    return {(size_t)lv.n_dofs_p(), (size_t)lv.n_owned_p, (size_t)lv.n_owned_p, 0};
  }
  size_t size_of(int which, int l) const { return which == 0 ? (size_t)L[l].n_dofs_v() : (size_t)L[l].n_dofs_p(); }

  template <class T> int dalloc(T **p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    AGMG_HIP_CHECK(hipMalloc((void **)p, n * sizeof(T)));
    return 0;
  }
  template <class T> void dfree(T *&p) {
    if (p) (void)hipFree((void *)p);
    p = nullptr;
  }
  template <class T> int upload(T **p, const std::vector<T> &v) {
    CHECK(dalloc(p, v.size()));
    if (!v.empty()) AGMG_HIP_CHECK(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
  }
  int sync() { AGMG_HIP_CHECK(hipStreamSynchronize(stream)); return 0; }
  int h2d(double *d, const double *h, size_t n) {
    AGMG_HIP_CHECK(hipMemcpyAsync(d, h, n * sizeof(double), hipMemcpyHostToDevice, stream));
    return sync();
  }
  int d2h(double *h, const double *d, size_t n) {
    AGMG_HIP_CHECK(hipMemcpyAsync(h, d, n * sizeof(double), hipMemcpyDeviceToHost, stream));
    return sync();
  }

  // doubles of slots_u the block path of a level writes beyond the cell slots (partial sums of shared nodes)
  size_t scratch_u_need(const DeviceLevel &lv) const {
    return std::max(lv.bt.built ? lv.bt.n_shared_slots * (size_t)lv.dim : 0, lv.bk.n_partial_slots * 3);
  }
  bool use_blocks = true;
  bool use_ell = true;     // assembled rows for the Q1 mass operator (AGMG_NO_ELL=1: matrix-free cell kernel + pull)
  bool use_ell_classes = true;   // columns of the assembled rows by offset class (AGMG_NO_ELL_CLASSES=1: explicit columns)
  bool mass_ell(const DeviceLevel &lv, const CoeffData &cd) const {
    return use_ell && lv.ell_col && cd.ell_val && cd.ell_stride == lv.ell_stride;
  }
  // one message round per application of the mass operator inside a smoother sequence (SymExchange; AGMG_NO_SYM=1:
  // ghost import + compress, two rounds).  sym_now: the vectors of the running operator application carry valid ghost
  // entries (set by the smoother around its sequence)
  bool use_sym = true, sym_now = false, sym_now_A = false;
  bool fine_ghosts_valid = false;   // the vector about to be restricted came out of a one-round application: no import needed
  std::vector<const double *> sym_rhs_valid;   // per level: the right-hand side whose ghost entries the pre-smoother made valid
  int n_sym_applications = 0;     // (tests)
  bool mass_sym_local(const DeviceLevel &lv, const CoeffData &cd) const { return use_sym && ldist(lv) && lv.sp.active() && mass_ell(lv, cd); }
  // (the message pattern differs from the two-round form: every rank must take the same path, agreed at agmg_compute_diagonals)
  bool mass_sym(const DeviceLevel &lv, const CoeffData &cd) const { return lv.sym_agreed && &cd == &lv.coeff && mass_sym_local(lv, cd); }
  int apply_mass_sym(const DeviceLevel &lv, const CoeffData &cd, const double *x, const EpiArgs &e);
  // the same for the A block on the levels that run the block kernels (brick levels keep ghost import + compress)
  bool a_sym_local(const DeviceLevel &lv, const CoeffData &cd) const {
    return use_sym && ldist(lv) && lv.sv.active() && lv.bt_symrow && use_blocks && lv.bt.built && !cd.newton && !cd.dilation &&
           !(bricks_on(lv) && brickq_supported(lv, cd, AGMG_OP_A));
  }
  bool a_sym(const DeviceLevel &lv, const CoeffData &cd) const { return lv.sym_agreed_A && &cd == &lv.coeff && a_sym_local(lv, cd); }
  int apply_A_sym(const DeviceLevel &lv, const CoeffData &cd, const double *x, const EpiArgs &e);
  bool use_batched_setup = true;   // eigenvalue estimates of all levels in lock step (AGMG_NO_BATCHED_SETUP=1: one by one)
  bool use_fused = true;   // one-launch smoother sequences on small levels (AGMG_NO_FUSED=1 disables)
  // Chebyshev coefficient sequences of cheb_vmult / cheb_step as a program for the fused kernels
  void program_vmult(const Smoother &s, SmootherProgram &p) const;
  void program_step(const Smoother &s, SmootherProgram &p) const;
  bool fused_mass(int l) const { return use_fused && L[l].mass_fusable; }
  // A block on a level that is one workgroup: the Chebyshev sequence (and the residual behind it) in one launch
  bool fused_A(int l) const { return use_fused && use_blocks && !ldist(L[l]) && blk_seq_possible(L[l], L[l].coeff); }
  int smooth_A_seq(int l, bool zero_start, const double *rhs, double *resid_dst);
  void free_level(DeviceLevel &l);
  void free_coeff(CoeffData &cd) {
    dfree(cd.etaT); dfree(cd.nw_strain); dfree(cd.nw_dEde); dfree(cd.nw_dEdp);
    dfree(cd.dil_lhs); dfree(cd.dil_de); dfree(cd.dil_dp); dfree(cd.ell_val); dfree(cd.brick_geo);
    cd = CoeffData();
  }
  int build_block_tables(DeviceLevel &lv, const HostLevel &hl, const std::vector<uint8_t> &cmv,
                         const std::vector<uint32_t> &send_nodes);
  int build_brick_tables(DeviceLevel &lv, const HostLevel &hl, const std::vector<uint8_t> &cmv, const std::vector<uint32_t> &send_nodes);
  // quad-per-cell brick kernel (kernels_brickq.hip) for the A block of structured Cartesian 3-D levels: faster than
  // the block kernels from 512 cubes of 8^3 cells on (r6: 121 -> 99 us, r7: 891 -> 776 us), slower below (r5: 23.3 ->
  // 24.5 us), so by default only those levels get brick tables.  AGMG_BRICKS=1: every level that tiles into
  // cubes; AGMG_BRICKS=0: none.
  int use_bricks = 2;       // 0 off, 1 every level that qualifies, 2 large levels only
  int brick_z_override = 0; // AGMG_BRICK_Z: layers per brick (1, 2, 4 or 8) instead of the size-based choice
  int brick_stack_override = 0; // AGMG_BRICK_STACK: cubes per brick along z (1, 2 or 4) instead of the size-based choice
  bool bricks_on(const DeviceLevel &lv) const { return use_bricks && lv.bk.built && lv.geom == AGMG_GEOM_CARTESIAN; }
  int ensure_workspace();
  int ensure_transfers();
  int ensure_mg_vectors();
  int get_vec(size_t idx, double **p);

  // out[k] = <x, ys[k]>, synchronous
  int dots(const double *x, const std::vector<const double *> &ys, const Span &sp, double *out);
  int norm(const double *x, const Span &sp, double *out) {
    double v;
    CHECK(dots(x, {x}, sp, &v));
    *out = std::sqrt(v);
    return 0;
  }

  // velocity-row operators (A, Stokes, B^T) of one rank: brick kernels where the level has brick tables,
  // else the block / slot path
  int op_velocity(const DeviceLevel &lv, const CoeffData &cd, int which, const double *src_u, const double *src_p,
                  const EpiArgs &eu, const EpiArgs *ep = nullptr) {
    const bool epi_ok = eu.epi == EPI_STORE || eu.epi == EPI_RESID || eu.epi == EPI_CHEB;
    if (bricks_on(lv) && epi_ok && brickq_supported(lv, cd, which))
      return launch_op_bricks(stream, lv, cd, which, src_u, src_p, slots_u, slots_p, eu, ep);
    return launch_op(stream, lv, cd, which, src_u, src_p, slots_u, slots_p, eu, use_blocks, PH_ALL, ep);
  }
  // level operators with fused epilogue: which 0 = A block, 1 = Schur mass matrix
  int apply_level(int which, int l, const CoeffData &cd, const double *x, const EpiArgs &e);
  int stokes_apply(const double *src, const EpiArgs &eu_in, double *dst);
  int stokes_vmult(double *dst, const double *src);
  int stokes_residual(double *dst, const double *rhs, const double *x);

  int cheb_vmult(int which, int l, double *&x, double *&xo, const double *b);
  int cheb_step(int which, int l, double *&x, double *&xo, const double *b);
  int smoother_apply(int which, int l, const double *rhs);
  int smoother_smooth(int which, int l, const double *rhs);
  int coarse_solve(int which);
  int level_v_step(int which, int l);
  int vcycle(int which, double *dst, const double *src);
  int estimate_device(int which, int l);
  int estimate_batched(const std::vector<std::pair<int, int>> &problems);
  int build_coarse(int which);
  int setup_smoothers();

  int precondition(bool expensive, double *dst, const double *src, int *status);
  int solver_cg(const std::function<int(double *, const double *)> &A,
                const std::function<int(double *, const double *)> &P, double *x, const double *b, const Span &sp,
                int max_it, double tol, double *work[4], int *its, bool *converged);
  int solver_bicgstab(const std::function<int(double *, const double *)> &A,
                      const std::function<int(double *, const double *)> &P, double *x, const double *b, const Span &sp,
                      int max_it, double tol, double *work[7], int *its, bool *converged);
  int bt_residual(double *dst, const double *b, const double *p);
  int restrict_to(int which, int l, const double *src_fine, const EpiArgs &e);
  int prolongate_from(int which, int l, double *dst_fine, const double *src_coarse);
  int global_multiplicity(const Exchange &x, const std::vector<uint32_t> &send_nodes_h, std::vector<double> &mult);
  std::vector<std::vector<uint32_t>> send_nodes_h[2];   // [velocity|pressure][level]: host copy of Exchange::send_nodes
  std::vector<uint32_t> gid_fin_h[2];                   // [pressure|velocity]: global node ids of the finest level (host)
};

// ===================================================================== memory / setup
void agmg_ctx::free_level(DeviceLevel &l) {
  dfree(l.cnvT); dfree(l.cnpT); dfree(l.cmask_v); dfree(l.cmask_p); dfree(l.cell_hT); dfree(l.invJT); dfree(l.jxwT);
  dfree(l.pull_ptr_v); dfree(l.pull_list_v); dfree(l.pull_ptr_p); dfree(l.pull_list_p);
  dfree(l.coeff.etaT); dfree(l.inv_diag_A); dfree(l.inv_diag_M);
  dfree(l.prol_parent_v); dfree(l.prol_pos_v); dfree(l.prol_parent_p); dfree(l.prol_pos_p);
  dfree(l.inv_mult_v); dfree(l.inv_mult_p); dfree(l.childrenT); dfree(l.prol_mask_v);
  dfree(l.bt.blk_hdr); dfree(l.bt.unodes); dfree(l.bt.uent8); dfree(l.bt.lidT);
  dfree(l.bt.sh_nodes); dfree(l.bt.sh_ptr); dfree(l.bt.sh_list); dfree(l.bt.blk_order);
  dfree(l.bk.gnode); dfree(l.bk.bcell); dfree(l.bk.pidx); dfree(l.bk.fx_nodes); dfree(l.bk.fx_ptr); dfree(l.bk.fx_list);
  dfree(l.bk.gh_nodes); dfree(l.bk.gh_ptr); dfree(l.bk.gh_list);
  for (Exchange *x : {&l.xv, &l.xp}) { dfree(x->send_nodes); dfree(x->expbuf); dfree(x->ghostbuf); }
  dfree(l.sp.sh_nodes); dfree(l.sp.ptr); dfree(l.sp.pos); dfree(l.sp.split); dfree(l.sp.row);
  dfree(l.sp.sendbuf); dfree(l.sp.recvbuf); dfree(l.sp.ypart);
  dfree(l.sv.sh_nodes); dfree(l.sv.ptr); dfree(l.sv.pos); dfree(l.sv.split); dfree(l.sv.row);
  dfree(l.sv.sendbuf); dfree(l.sv.recvbuf); dfree(l.sv.ypart); dfree(l.sv_shidx); dfree(l.bt_symrow);
  dfree(l.gid_v); dfree(l.gid_p);
  dfree(l.ell_col); dfree(l.coeff.ell_val); dfree(l.coeff.brick_geo);
  dfree(l.ell_xrow); dfree(l.ell_xptr); dfree(l.ell_xent); dfree(l.ell_cls); dfree(l.ell_cls_off);
  l = DeviceLevel();
}

// Block-scatter tables (agmg_internal.h BlockTables).  Blocks are runs of consecutive cells in the
// host's cell order (Morton order keeps them compact); a velocity node is "exclusive" when all
// cells touching it lie in one block.
// Distributed levels: ghost nodes and owned nodes another rank ghosts (`send_nodes`) are always
// "shared": their sums wait for / feed the ghost exchange in the fix-up kernel.
// host-side table building runs on a few threads (std::thread: no OpenMP runtime next to the caller's);
// AGMG_HOST_THREADS overrides the default of min(16, hardware threads)
static int host_threads() {
  static const int n = [] {
    const char *e = getenv("AGMG_HOST_THREADS");
    int t = e ? atoi(e) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    return t < 1 ? 1 : t;
  }();
  return n;
}
template <class F> static void parallel_for(const size_t n, const F &fn) {   // fn(begin, end, thread)
  const int T = (int)std::min<size_t>((size_t)host_threads(), std::max<size_t>(1, n / 1024));
  if (T <= 1) { fn((size_t)0, n, 0); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t) th.emplace_back([&, t] { fn(n * t / T, n * (t + 1) / T, t); });
  for (auto &x : th) x.join();
}

int agmg_ctx::build_block_tables(DeviceLevel &lv, const HostLevel &hl, const std::vector<uint8_t> &cmv,
                                 const std::vector<uint32_t> &send_nodes) {
  BlockTables &bt = lv.bt;
  const int dim = lv.dim, NV = ipow(3, dim), nn = lv.n_nodes_v;
  const int cpw = cells_per_wave(dim), bc = cpw * WAVES_PER_BLOCK;
  const int nb = (lv.n_cells + bc - 1) / bc;
  // pass 1: the sorted unique nodes of every block; whether one block or several touch each node
  std::vector<std::vector<uint32_t>> bl_nodes(nb);
  std::vector<uint8_t> nblk(nn, 0);
  parallel_for((size_t)nb, [&](size_t b0, size_t b1, int) {
    for (size_t b = b0; b < b1; ++b) {
      const size_t c0 = b * bc, c1 = std::min(c0 + (size_t)bc, (size_t)lv.n_cells);
      std::vector<uint32_t> &u = bl_nodes[b];
      u.assign(hl.cnv.begin() + c0 * NV, hl.cnv.begin() + c1 * NV);
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
      // (only "one block" against "several" is used: bit 0 = seen, bit 1 = seen again; an OR cannot wrap)
      for (uint32_t node : u)
        if (__atomic_fetch_or(&nblk[node], (uint8_t)1, __ATOMIC_RELAXED) & 1) __atomic_fetch_or(&nblk[node], (uint8_t)2, __ATOMIC_RELAXED);
    }
  });
  // shared = touched by several blocks; on distributed levels also every ghost node and every owned
  // node another rank ghosts (the count of local blocks stays in nblk)
  std::vector<uint8_t> shared(nn, 0);
  parallel_for((size_t)nn, [&](size_t i0, size_t i1, int) {
    for (size_t i = i0; i < i1; ++i) shared[i] = nblk[i] > 1 || (int)i >= lv.n_owned_v;
  });
  for (uint32_t node : send_nodes) shared[node] = 1;
  // pass 2: per block exclusive nodes first, then the shared ones (each group ascending)
  std::vector<uint32_t> blk_hdr((size_t)nb * 4, 0);
  std::vector<uint8_t> too_big(nb, 0);
  parallel_for((size_t)nb, [&](size_t b0, size_t b1, int) {
    std::vector<uint32_t> sh;
    for (size_t b = b0; b < b1; ++b) {
      std::vector<uint32_t> &u = bl_nodes[b];
      sh.clear();
      size_t ne = 0;
      for (uint32_t node : u) { if (shared[node]) sh.push_back(node); else u[ne++] = node; }
      std::copy(sh.begin(), sh.end(), u.begin() + ne);
      if ((int)u.size() > bc * NV / 2) too_big[b] = 1;
      blk_hdr[b * 4 + 0] = (uint32_t)u.size();
      blk_hdr[b * 4 + 1] = (uint32_t)ne;
    }
  });
  size_t nslots = 0;
  int max_u = 0;
  for (int b = 0; b < nb; ++b) {
    if (too_big[b]) return 0;  // (degenerate meshes) kernel's per-thread metadata bound: keep the slot path
    max_u = std::max(max_u, (int)blk_hdr[(size_t)b * 4]);
    blk_hdr[(size_t)b * 4 + 2] = (uint32_t)nslots;
    nslots += blk_hdr[(size_t)b * 4] - blk_hdr[(size_t)b * 4 + 1];
  }
  // per block: node words, local ids of the cells' nodes, up to 8 LDS slots per node in (cell, local dof) order
  const int wtile = blk_wave_tile(dim);
  const uint16_t zero_slot = (uint16_t)(WAVES_PER_BLOCK * wtile);  // entries are LDS slot indices (doubles)
  std::vector<uint32_t> unodes((size_t)nb * max_u, 0);
  std::vector<uint16_t> uent8((size_t)nb * max_u * 8, zero_slot), lidT((size_t)LID_ROW * lv.cstride, 0);
  std::vector<uint8_t> valence_ok(nb, 1);
  parallel_for((size_t)nb, [&](size_t b0, size_t b1, int) {
    std::vector<uint8_t> cnt;
    for (size_t b = b0; b < b1; ++b) {
      const size_t c0 = b * bc, c1 = std::min(c0 + (size_t)bc, (size_t)lv.n_cells);
      const size_t m0 = b * max_u;
      const std::vector<uint32_t> &u = bl_nodes[b];
      const int U = (int)u.size(), ne = (int)blk_hdr[b * 4 + 1];
      for (int l = 0; l < U; ++l) unodes[m0 + l] = u[l] | ((uint32_t)cmv[u[l]] << 29);
      cnt.assign(U, 0);
      for (size_t cell = c0; cell < c1; ++cell) {
        const int w = (int)(cell - c0) / cpw, jj = (int)(cell - c0) % cpw;
        for (int a = 0; a < NV; ++a) {
          const uint32_t node = hl.cnv[cell * NV + a];
          const auto first = shared[node] ? u.begin() + ne : u.begin(), last = shared[node] ? u.end() : u.begin() + ne;
          const int l = (int)(std::lower_bound(first, last, node) - u.begin());
          lidT[cell * LID_ROW + a] = (uint16_t)l;
          if (cnt[l] >= 8) { valence_ok[b] = 0; continue; }  // node of valence > 8 inside one block
          uent8[(m0 + l) * 8 + cnt[l]++] = (uint16_t)(w * wtile + a * BLK_SLOT_ROW + jj * dim);
        }
      }
    }
  });
  for (int b = 0; b < nb; ++b) if (!valence_ok[b]) return 0;   // keep the slot path
  // slots of the shared nodes, block by block
  std::vector<uint32_t> sh_count(nn, 0);
  std::vector<std::pair<uint32_t, uint32_t>> sh_pairs;  // (node, slot position)
  sh_pairs.reserve(nslots + send_nodes.size());
  for (int b = 0; b < nb; ++b) {
    const std::vector<uint32_t> &u = bl_nodes[b];
    size_t slot = blk_hdr[(size_t)b * 4 + 2];
    for (size_t l = blk_hdr[(size_t)b * 4 + 1]; l < u.size(); ++l) {
      sh_pairs.emplace_back(u[l], (uint32_t)slot++);
      ++sh_count[u[l]];
    }
  }
  std::vector<std::vector<uint32_t>>().swap(bl_nodes);
  // remote partial sums: entry k of the receive buffer belongs to owned node send_nodes[k]
  for (size_t k = 0; k < send_nodes.size(); ++k) {
    sh_pairs.emplace_back(send_nodes[k], (uint32_t)(nslots + k));
    ++sh_count[send_nodes[k]];
  }
  // fix-up CSR over shared nodes (ascending node order; slots of a node in block order, then remote)
  std::vector<uint32_t> sh_nodes, sh_ptr(1, 0), sh_list(nslots + send_nodes.size());
  std::vector<uint32_t> sh_index(nn, 0);
  for (int node = 0; node < nn; ++node)
    if (sh_count[node]) {
      if (node < lv.n_owned_v) bt.n_shared_owned = (int)sh_nodes.size() + 1;
      sh_index[node] = (uint32_t)sh_nodes.size();
      sh_nodes.push_back((uint32_t)node | ((uint32_t)cmv[node] << 29));
      sh_ptr.push_back(sh_ptr.back() + sh_count[node]);
    }
  std::vector<uint32_t> fill(sh_ptr.begin(), sh_ptr.end() - 1);
  for (auto &pr : sh_pairs) sh_list[fill[sh_index[pr.first]]++] = pr.second;
  if (lv.n_owned_v < nn) {   // distributed level: interior blocks (no ghost node) first, then interface blocks
    std::vector<uint32_t> order, iface;
    for (int b = 0; b < nb; ++b) {
      bool ghost = false;
      const int U = (int)blk_hdr[(size_t)b * 4];
      for (int l = 0; l < U && !ghost; ++l) ghost = (int)(unodes[(size_t)b * max_u + l] & NODE_MASK) >= lv.n_owned_v;
      // (interior blocks of the Stokes and B^T operators read the pressure source before its ghost import has
      // landed: a block with a ghost pressure node is an interface block too, whatever rule the caller's
      // partition used for the two spaces)
      if (!hl.cnp.empty()) {
        const int NPc = 1 << dim;
        const size_t c0 = (size_t)b * bc, c1 = std::min(c0 + (size_t)bc, (size_t)lv.n_cells);
        for (size_t i = c0 * NPc; i < c1 * NPc && !ghost; ++i) ghost = (int)hl.cnp[i] >= lv.n_owned_p;
      }
      (ghost ? iface : order).push_back((uint32_t)b);
    }
    bt.n_interior = (int)order.size();
    order.insert(order.end(), iface.begin(), iface.end());
    CHECK(upload(&bt.blk_order, order));
  }
  bt.block_cells = bc; bt.n_blocks = nb; bt.max_unodes = max_u;
  bt.n_shared = (int)sh_nodes.size(); bt.n_shared_slots = nslots;
  CHECK(upload(&bt.blk_hdr, blk_hdr));
  CHECK(upload(&bt.unodes, unodes)); CHECK(upload(&bt.uent8, uent8)); CHECK(upload(&bt.lidT, lidT));
  CHECK(upload(&bt.sh_nodes, sh_nodes)); CHECK(upload(&bt.sh_ptr, sh_ptr)); CHECK(upload(&bt.sh_list, sh_list));
  bt.bytes = (blk_hdr.size() + unodes.size() + sh_nodes.size() + sh_ptr.size() + sh_list.size()) * 4 + (uent8.size() + lidT.size()) * 2;
  bt.built = true;
  return 0;
}

// Brick tables (agmg_internal.h BrickTables).  The structure is recovered from the cell -> node
// table alone: two cells are x-neighbours when the +x face centre of one is the -x face centre of the
// other (cell-local node order is lexicographic, so this also says their local axes agree); a
// breadth-first walk over these neighbour links gives every cell integer coordinates inside its
// consistently oriented region (a tree of the forest, or several); the regions are cut into cubes of
// 8 x 8 x 8 cells and the cubes into bricks of Z layers.  Every lattice position of a brick must
// receive the same node from all cells around it, every cube must be complete and every cell must
// land in a brick: otherwise the level keeps the block path (returns 0 with bk.built == false).
int agmg_ctx::build_brick_tables(DeviceLevel &lv, const HostLevel &hl, const std::vector<uint8_t> &cmv, const std::vector<uint32_t> &send_nodes) {
  BrickTables &bk = lv.bk;
  const int n = lv.n_cells, nn = lv.n_nodes_v;
  if (lv.dim != 3 || n < 512 || n % 512 != 0 || (unsigned)nn > BRICK_NODE_MASK) return 0;
  if (lv.geom != AGMG_GEOM_CARTESIAN) return 0;                 // (the brick kernel reads cell extents)
  if (use_bricks == 2 && n < 512 * 512) return 0;               // small levels: the block kernels are faster
  const uint32_t *cnv = hl.cnv.data();
  // face centres: local nodes (0,1,1) (2,1,1) (1,0,1) (1,2,1) (1,1,0) (1,1,2)
  static const int FC[6] = {12, 14, 10, 16, 4, 22};
  std::vector<int32_t> first(nn, -1);
  std::vector<int32_t> nbr((size_t)n * 6, -1);
  for (int cell = 0; cell < n; ++cell)
    for (int f = 0; f < 6; ++f) {
      const uint32_t node = cnv[(size_t)cell * 27 + FC[f]];
      const int32_t me = cell * 6 + f;
      if (first[node] < 0) { first[node] = me; continue; }
      const int32_t other = first[node];
      const int oc = other / 6, of = other % 6;
      if ((of ^ 1) == f && oc != cell && nbr[(size_t)cell * 6 + f] < 0 && nbr[(size_t)oc * 6 + of] < 0) {
        nbr[(size_t)cell * 6 + f] = oc;
        nbr[(size_t)oc * 6 + of] = cell;
      }
    }
  std::vector<int32_t>().swap(first);
  // integer coordinates by breadth-first search; regions numbered in order of their first cell
  std::vector<int32_t> coord((size_t)n * 3, 0), region(n, -1);
  std::vector<int32_t> queue;
  queue.reserve(n);
  int n_regions = 0;
  std::vector<std::array<int32_t, 3>> rmin;
  for (int seed = 0; seed < n; ++seed) {
    if (region[seed] >= 0) continue;
    const int r = n_regions++;
    region[seed] = r;
    std::array<int32_t, 3> mn = {0, 0, 0};
    size_t head = queue.size();
    queue.push_back(seed);
    while (head < queue.size()) {
      const int cell = queue[head++];
      for (int f = 0; f < 6; ++f) {
        const int32_t o = nbr[(size_t)cell * 6 + f];
        if (o < 0 || region[o] >= 0) continue;
        region[o] = r;
        for (int d = 0; d < 3; ++d) coord[(size_t)o * 3 + d] = coord[(size_t)cell * 3 + d];
        coord[(size_t)o * 3 + f / 2] += (f & 1) ? 1 : -1;
        for (int d = 0; d < 3; ++d) mn[d] = std::min(mn[d], coord[(size_t)o * 3 + d]);
        queue.push_back(o);
      }
    }
    rmin.push_back(mn);
  }
  std::vector<int32_t>().swap(queue);
  // cubes: key (region, i >> 3, j >> 3, k >> 3), numbered in order of their first cell
  std::unordered_map<uint64_t, int32_t> cube_of;
  std::vector<int32_t> cube_cells;   // [cube][512], index (k & 7) * 64 + (j & 7) * 8 + (i & 7)
  for (int cell = 0; cell < n; ++cell) {
    int32_t q[3];
    for (int d = 0; d < 3; ++d) {
      q[d] = coord[(size_t)cell * 3 + d] - rmin[region[cell]][d];
      coord[(size_t)cell * 3 + d] = q[d];
    }
    if ((q[0] >> 3) >= (1 << 14) || (q[1] >> 3) >= (1 << 14) || (q[2] >> 3) >= (1 << 14) || region[cell] >= (1 << 20)) return 0;
    const uint64_t key = ((uint64_t)region[cell] << 42) | ((uint64_t)(q[2] >> 3) << 28) | ((uint64_t)(q[1] >> 3) << 14) | (uint64_t)(q[0] >> 3);
    auto it = cube_of.find(key);
    int32_t cube;
    if (it == cube_of.end()) {
      cube = (int32_t)(cube_cells.size() / 512);
      cube_of.emplace(key, cube);
      cube_cells.resize(cube_cells.size() + 512, -1);
    } else cube = it->second;
    int32_t &slot = cube_cells[(size_t)cube * 512 + (q[2] & 7) * 64 + (q[1] & 7) * 8 + (q[0] & 7)];
    if (slot >= 0) return 0;   // two cells at one position: not a lattice
    slot = cell;
  }
  const int n_cubes = (int)(cube_cells.size() / 512);
  if ((size_t)n_cubes * 512 != (size_t)n) return 0;   // some cube is incomplete
  // layers per brick: enough bricks to fill the chip on small levels, whole cubes on large ones, and on very large
  // ones STACKS of S cubes along z (a brick of 8 S layers has no partial sums on the cube faces inside it, and its
  // prologue and top plane are paid once per 8 S layers), where every cube sits in a complete stack and enough bricks
  // remain to keep the 512 workgroup slots of the chip in balance
  int Z = 8;
  if (n_cubes < 512) { Z = 1; while (Z < 8 && n_cubes * 8 / (2 * Z) >= 512) Z *= 2; }
  if (brick_z_override == 1 || brick_z_override == 2 || brick_z_override == 4 || brick_z_override == 8) Z = brick_z_override;
  else if (brick_stack_override == 1 || brick_stack_override == 2 || brick_stack_override == 4) Z = 8;   // (stacks are made of whole cubes)
  const auto key_of = [](uint64_t region_, uint64_t cx, uint64_t cy, uint64_t cz) { return (region_ << 42) | (cz << 28) | (cy << 14) | cx; };
  std::vector<std::array<int32_t, 4>> cube_pos(n_cubes);     // (region, cx, cy, cz) of every cube
  for (const auto &kv : cube_of)
    cube_pos[kv.second] = {(int32_t)(kv.first >> 42), (int32_t)(kv.first & 0x3FFF), (int32_t)((kv.first >> 14) & 0x3FFF), (int32_t)((kv.first >> 28) & 0x3FFF)};
  const auto stackable = [&](int S_) {
    if (n_cubes % S_) return false;
    for (int cube = 0; cube < n_cubes; ++cube) {
      const auto &q = cube_pos[cube];
      const int base = q[3] - q[3] % S_;
      for (int s_ = 0; s_ < S_; ++s_)
        if (!cube_of.count(key_of(q[0], q[1], q[2], base + s_))) return false;
    }
    return true;
  };
  int S = 1;
  if (Z == 8) {
    if (brick_stack_override == 1 || brick_stack_override == 2 || brick_stack_override == 4) {
      if (stackable(brick_stack_override)) S = brick_stack_override;
    } else {
      // (measured at r7 = 4096 cubes: 763 us per A-block vmult with one cube per brick, 744 with two, 751 with four:
      // fewer, longer bricks save partial sums but balance worse over the 512 workgroup slots)
      if (n_cubes / 4 >= 4096 && stackable(4)) S = 4;
      else if (n_cubes / 2 >= 1024 && stackable(2)) S = 2;
    }
  }
  // bricks in the order of their first cube; stack_cubes[brick][s] = the cube of layers 8 s .. 8 s + 7
  std::vector<int32_t> stack_cubes;
  if (S > 1) {
    for (int cube = 0; cube < n_cubes; ++cube) {
      const auto &q = cube_pos[cube];
      if (q[3] % S) continue;
      for (int s_ = 0; s_ < S; ++s_) stack_cubes.push_back(cube_of.at(key_of(q[0], q[1], q[2], q[3] + s_)));
    }
    if ((int)stack_cubes.size() != n_cubes) return 0;
    Z = 8 * S;
  }
  const int per_cube = S > 1 ? 1 : 8 / Z, nb = S > 1 ? n_cubes / S : n_cubes * per_cube;
  const int nplanes = 2 * Z + 1, latn = BRICK_PLANE * nplanes;
  if ((uint64_t)nb * latn > 0xFFFFFFFFull) return 0;
  // surface enumeration of the lattice
  std::vector<uint16_t> pidx(latn, 0xFFFF);
  int npb = 0;
  for (int K = 0; K < nplanes; ++K)
    for (int J = 0; J < BRICK_LAT; ++J)
      for (int I = 0; I < BRICK_LAT; ++I)
        if (K == 0 || K == nplanes - 1 || J == 0 || J == BRICK_LAT - 1 || I == 0 || I == BRICK_LAT - 1)
          pidx[(size_t)K * BRICK_PLANE + J * BRICK_LAT + I] = (uint16_t)npb++;
  if ((uint64_t)nb * npb > 0xFFFFFFFFull) return 0;
  std::vector<uint32_t> gnode((size_t)nb * latn, 0xFFFFFFFFu), bcell((size_t)nb * Z * 64);
  for (int first = 0; first < (S > 1 ? nb : n_cubes); ++first)
    for (int sub = 0; sub < per_cube; ++sub) {
      const int brick = first * per_cube + sub;
      uint32_t *lat = &gnode[(size_t)brick * latn];
      for (int kl = 0; kl < Z; ++kl)
        for (int jj = 0; jj < 8; ++jj)
          for (int ii = 0; ii < 8; ++ii) {
            const int cube = S > 1 ? stack_cubes[(size_t)first * S + kl / 8] : first;
            const int layer = S > 1 ? kl % 8 : sub * Z + kl;
            const int cell = cube_cells[(size_t)cube * 512 + layer * 64 + jj * 8 + ii];
            bcell[((size_t)brick * Z + kl) * 64 + jj * 8 + ii] = (uint32_t)cell;
            for (int a = 0; a < 27; ++a) {
              const int ax = a % 3, ay = (a / 3) % 3, az = a / 9;
              uint32_t &e = lat[(size_t)(2 * kl + az) * BRICK_PLANE + (2 * jj + ay) * BRICK_LAT + 2 * ii + ax];
              const uint32_t node = cnv[(size_t)cell * 27 + a];
              if (e == 0xFFFFFFFFu) e = node;
              else if (e != node) return 0;   // neighbouring cells disagree: not a structured brick
            }
          }
    }
  // ownership: the first brick (and lattice position) that lists a node owns it
  // (distributed levels: a ghost node is owned by no brick, every share of it goes through a partial slot)
  const uint32_t n_owned = (uint32_t)lv.n_owned_v;
  bk.n_interior = 0;
  if (n_owned < (uint32_t)nn) {
    // bricks without a ghost node first: they run while the ghost import is in flight (deal.II's cell_loop overlaps
    // the same way, matrix_free_operators.cc:357, 794); the order inside the two groups stays
    std::vector<int> order;
    std::vector<uint8_t> iface(nb, 0);
    for (int brick = 0; brick < nb; ++brick)
      for (int l = 0; l < latn && !iface[brick]; ++l) iface[brick] = gnode[(size_t)brick * latn + l] >= n_owned;
    for (int pass = 0; pass < 2; ++pass)
      for (int brick = 0; brick < nb; ++brick)
        if (iface[brick] == pass) order.push_back(brick);
    for (int brick = 0; brick < nb; ++brick) bk.n_interior += !iface[brick];
    std::vector<uint32_t> g2(gnode.size()), c2(bcell.size());
    for (int k = 0; k < nb; ++k) {
      std::copy(gnode.begin() + (size_t)order[k] * latn, gnode.begin() + (size_t)(order[k] + 1) * latn, g2.begin() + (size_t)k * latn);
      std::copy(bcell.begin() + (size_t)order[k] * Z * 64, bcell.begin() + (size_t)(order[k] + 1) * Z * 64, c2.begin() + (size_t)k * Z * 64);
    }
    gnode.swap(g2); bcell.swap(c2);
  }
  if ((uint64_t)nb * npb + send_nodes.size() > 0xFFFFFFF0ull) return 0;
  std::vector<uint32_t> owner(nn, 0xFFFFFFFFu), fx_count(nn, 0);
  std::vector<std::pair<uint32_t, uint32_t>> pairs;   // (node, partial slot)
  for (int brick = 0; brick < nb; ++brick)
    for (int l = 0; l < latn; ++l) {
      uint32_t &e = gnode[(size_t)brick * latn + l];
      const uint32_t node = e;
      if (node >= n_owned) {
        if (pidx[l] == 0xFFFF) return 0;   // a ghost node inside a brick
        owner[node] = 0xFFFFFFFEu;
        e = node | BRICK_PARTIAL | ((uint32_t)cmv[node] << 29);
        pairs.emplace_back(node, (uint32_t)((size_t)brick * npb + pidx[l]));
        ++fx_count[node];
      } else if (owner[node] == 0xFFFFFFFFu) {
        owner[node] = (uint32_t)((size_t)brick * latn + l);
        e = node | ((uint32_t)cmv[node] << 29);
      } else {
        if ((owner[node] / (uint32_t)latn) == (uint32_t)brick || pidx[l] == 0xFFFF) return 0;   // a node twice in one brick / shared from inside
        e = node | BRICK_PARTIAL | ((uint32_t)cmv[node] << 29);
        pairs.emplace_back(node, (uint32_t)((size_t)brick * npb + pidx[l]));
        ++fx_count[node];
      }
    }
  for (int node = 0; node < nn; ++node)
    if (owner[node] == 0xFFFFFFFFu) return 0;   // a node no cell touches
  std::vector<uint32_t>().swap(owner);
  // remote partial sums of the owned nodes other ranks ghost: entry k of the receive buffer belongs to send_nodes[k]
  for (uint32_t node : send_nodes) {
    if (node >= n_owned) return 0;
    ++fx_count[node];
  }
  // owned nodes [0, n_owned) form the fix-up lists, ghost nodes the lists of the compress step
  std::vector<uint32_t> fx_nodes, fx_ptr(1, 0), fx_index(nn, 0), gh_nodes, gh_ptr(1, 0);
  for (int node = 0; node < nn; ++node)
    if (fx_count[node]) {
      std::vector<uint32_t> &nodes = (uint32_t)node < n_owned ? fx_nodes : gh_nodes, &ptr = (uint32_t)node < n_owned ? fx_ptr : gh_ptr;
      fx_index[node] = (uint32_t)nodes.size();
      nodes.push_back((uint32_t)node < n_owned ? ((uint32_t)node | ((uint32_t)cmv[node] << 29)) : (uint32_t)node);
      ptr.push_back(ptr.back() + fx_count[node]);
    }
  std::vector<uint32_t> fx_list(fx_ptr.back()), gh_list(gh_ptr.back());
  std::vector<uint32_t> fill(fx_ptr.begin(), fx_ptr.end() - 1), gfill(gh_ptr.begin(), gh_ptr.end() - 1);
  for (auto &pr : pairs) {   // pairs are in brick order
    if (pr.first < n_owned) fx_list[fill[fx_index[pr.first]]++] = pr.second;
    else gh_list[gfill[fx_index[pr.first]]++] = pr.second;
  }
  for (size_t k = 0; k < send_nodes.size(); ++k) fx_list[fill[fx_index[send_nodes[k]]]++] = (uint32_t)((size_t)nb * npb + k);
  bk.Z = Z; bk.n_bricks = nb; bk.latn = latn; bk.npb = npb;
  bk.n_fix = (int)fx_nodes.size();
  bk.n_partial_slots = (size_t)nb * npb;
  CHECK(upload(&bk.gnode, gnode)); CHECK(upload(&bk.bcell, bcell)); CHECK(upload(&bk.pidx, pidx));
  CHECK(upload(&bk.fx_nodes, fx_nodes)); CHECK(upload(&bk.fx_ptr, fx_ptr)); CHECK(upload(&bk.fx_list, fx_list));
  bk.n_gh = (int)gh_nodes.size();
  if (bk.n_gh) { CHECK(upload(&bk.gh_nodes, gh_nodes)); CHECK(upload(&bk.gh_ptr, gh_ptr)); CHECK(upload(&bk.gh_list, gh_list)); }
  bk.bytes = (gnode.size() + bcell.size() + fx_nodes.size() + fx_ptr.size() + fx_list.size() + gh_nodes.size() + gh_ptr.size() + gh_list.size()) * 4 + pidx.size() * 2;
  bk.built = true;
  return 0;
}

int agmg_ctx::ensure_workspace() {
  size_t nu = 0, np = 0;
  const int dim = opt.dim, NV = ipow(3, dim), NP = ipow(2, dim);
  auto grow = [&](const DeviceLevel &l) {
    nu = std::max(nu, (size_t)NV * l.cstride * dim);
    nu = std::max(nu, l.bk.n_partial_slots * 3);
    np = std::max(np, (size_t)NP * l.cstride * dim);
  };
  for (auto &l : L) grow(l);
  for (auto &l : ad.Lb) if (l.set) grow(l);
  if (ad.on) grow(ad.act);
  if (nu > slots_u_cap) { dfree(slots_u); CHECK(dalloc(&slots_u, nu)); slots_u_cap = nu; }
  if (np > slots_p_cap) { dfree(slots_p); CHECK(dalloc(&slots_p, np)); slots_p_cap = np; }
  if (!dot_partials) {
    CHECK(dalloc(&dot_partials, (size_t)DOT_BLOCKS * DOT_MAX_VECS));
    CHECK(dalloc(&dot_out_d, 256));
    AGMG_HIP_CHECK(hipHostMalloc((void **)&dot_out_h, 256 * sizeof(double)));
  }
  return 0;
}

int agmg_ctx::ensure_mg_vectors() {
  if (mg_alloc) return 0;
  for (int w = 0; w < 2; ++w) {
    for (auto *v : {&mg_def[w], &mg_sol[w], &mg_tmp[w], &mg_t[w]}) {
      for (auto &p : *v) dfree(p);
      v->assign(nl(), nullptr);
    }
    for (int l = 0; l < nl(); ++l) {
      const size_t n = size_of(w, l);
      CHECK(dalloc(&mg_def[w][l], n)); CHECK(dalloc(&mg_sol[w][l], n));
      CHECK(dalloc(&mg_tmp[w][l], n)); CHECK(dalloc(&mg_t[w][l], n));
    }
    sm[w].assign(nl(), Smoother());
  }
  mg_alloc = true;
  return 0;
}

int agmg_ctx::get_vec(size_t idx, double **p) {
  const size_t n = (size_t)act().n_dofs_v() + act().n_dofs_p();
  while (pool.size() <= idx) {
    double *v;
    CHECK(dalloc(&v, n));
    pool.push_back(v);
  }
  *p = pool[idx];
  return 0;
}

int agmg_ctx::dots(const double *x, const std::vector<const double *> &ys, const Span &sp, double *out) {
  const int total = (int)ys.size();
  if (total > 256) { set_error("dots: too many vectors"); return -1; }
  const bool red = dist() && !sp.local;   // (a replicated level: every rank computes the whole sum itself)
  for (int k0 = 0; k0 < total; k0 += DOT_MAX_VECS) {
    const int nv = std::min(DOT_MAX_VECS, total - k0);
    VecPtrs vp;
    for (int k = 0; k < DOT_MAX_VECS; ++k) vp.p[k] = ys[k0 + std::min(k, nv - 1)];
    // one rank: the final reduction writes straight into the pinned (device-visible, coherent) host
    // buffer: no copy kernel between the reduction and the host's wait
    CHECK(vec_multi_dot(stream, x, vp, nv, sp.n_dot, sp.seg1, sp.gap, dot_partials, (red ? dot_out_d : dot_out_h) + k0));
  }
  if (red && comm.device_buffers) {
    if (comm.allreduce_sum(comm.user, dot_out_d, total, (void *)stream)) { set_error("all-reduce callback failed"); return -101; }
  }
  if (red) AGMG_HIP_CHECK(hipMemcpyAsync(dot_out_h, dot_out_d, total * sizeof(double), hipMemcpyDeviceToHost, stream));
  CHECK(sync());
  if (red && !comm.device_buffers) {
    if (comm.allreduce_sum(comm.user, dot_out_h, total, nullptr)) { set_error("all-reduce callback failed"); return -101; }
  }
  for (int k = 0; k < total; ++k) out[k] = dot_out_h[k];
  return 0;
}

// sum over ranks of a few host scalars (setup paths)
int agmg_ctx::allreduce_host(double *v, int n) {
  if (!dist()) return 0;
  if (n > 256) { set_error("allreduce_host: too many values"); return -1; }
  if (!comm.device_buffers) {
    if (comm.allreduce_sum(comm.user, v, n, nullptr)) { set_error("all-reduce callback failed"); return -101; }
    return 0;
  }
  CHECK(ensure_workspace());
  AGMG_HIP_CHECK(hipMemcpyAsync(dot_out_d, v, n * sizeof(double), hipMemcpyHostToDevice, stream));
  if (comm.allreduce_sum(comm.user, dot_out_d, n, (void *)stream)) { set_error("all-reduce callback failed"); return -101; }
  AGMG_HIP_CHECK(hipMemcpyAsync(dot_out_h, dot_out_d, n * sizeof(double), hipMemcpyDeviceToHost, stream));
  CHECK(sync());
  for (int k = 0; k < n; ++k) v[k] = dot_out_h[k];
  return 0;
}

int agmg_ctx::allreduce_vec(double *d, size_t n) {
  if (!dist() || n == 0) return 0;
  if (!comm_set) { set_error("distributed context without a communicator (agmg_set_communicator)"); return -1; }
  if (n > 0x7FFFFFFFull) { set_error("allreduce_vec: vector too long"); return -1; }
  if (comm.device_buffers) {
    if (comm.allreduce_sum(comm.user, d, (int32_t)n, (void *)stream)) { set_error("all-reduce callback failed"); return -101; }
    return 0;
  }
  if (n > stage_cap) {
    if (stage_h) (void)hipHostFree(stage_h);
    stage_h = nullptr; stage_cap = 0;
    AGMG_HIP_CHECK(hipHostMalloc((void **)&stage_h, (n + n / 2 + 64) * sizeof(double)));
    stage_cap = n + n / 2 + 64;
  }
  CHECK(d2h(stage_h, d, n));
  if (comm.allreduce_sum(comm.user, stage_h, (int32_t)n, nullptr)) { set_error("all-reduce callback failed"); return -101; }
  return h2d(d, stage_h, n);
}

// ===================================================================== ghost exchange
int agmg_ctx::add_update(std::vector<Msg> &msgs, const Exchange &x, double *vec, int nc) {
  if (!x.active()) return 0;
  if (x.n_export > 0) CHECK(vec_gather_nodes(stream, x.expbuf, vec, x.send_nodes, x.n_export, nc));
  for (size_t k = 0; k < x.nbr.size(); ++k)
    msgs.push_back({x.nbr[k], x.expbuf + (size_t)x.send_off[k] * nc, (int64_t)(x.send_off[k + 1] - x.send_off[k]) * nc,
                    vec + ((size_t)x.n_owned + x.recv_off[k]) * nc, (int64_t)(x.recv_off[k + 1] - x.recv_off[k]) * nc});
  return 0;
}
void agmg_ctx::add_compress(std::vector<Msg> &msgs, const Exchange &x, int nc) {
  for (size_t k = 0; k < x.nbr.size(); ++k)
    msgs.push_back({x.nbr[k], x.ghostbuf + (size_t)x.recv_off[k] * nc, (int64_t)(x.recv_off[k + 1] - x.recv_off[k]) * nc,
                    x.expbuf + (size_t)x.send_off[k] * nc, (int64_t)(x.send_off[k + 1] - x.send_off[k]) * nc});
}
int agmg_ctx::exchange(const std::vector<Msg> &msgs) {
  if (msgs.empty()) return 0;
  if (!comm_set) { set_error("distributed context without a communicator (agmg_set_communicator)"); return -1; }
  const int n = (int)msgs.size();
  std::vector<int32_t> peer(n);
  std::vector<const double *> snd(n);
  std::vector<double *> rcv(n);
  std::vector<int64_t> ns(n), nr(n);
  for (int i = 0; i < n; ++i) { peer[i] = msgs[i].peer; ns[i] = msgs[i].n_send; nr[i] = msgs[i].n_recv; }
  if (comm.device_buffers) {
    for (int i = 0; i < n; ++i) { snd[i] = msgs[i].send; rcv[i] = msgs[i].recv; }
    if (comm.exchange(comm.user, n, peer.data(), snd.data(), ns.data(), rcv.data(), nr.data(), (void *)stream)) {
      set_error("exchange callback failed"); return -101;
    }
    return 0;
  }
  size_t tot = 0;
  for (int i = 0; i < n; ++i) tot += (size_t)(ns[i] + nr[i]);
  if (tot > stage_cap) {
    if (stage_h) (void)hipHostFree(stage_h);
    stage_h = nullptr; stage_cap = 0;
    AGMG_HIP_CHECK(hipHostMalloc((void **)&stage_h, (tot + tot / 2 + 64) * sizeof(double)));
    stage_cap = tot + tot / 2 + 64;
  }
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    snd[i] = stage_h + off;
    if (ns[i]) AGMG_HIP_CHECK(hipMemcpyAsync(stage_h + off, msgs[i].send, ns[i] * sizeof(double), hipMemcpyDeviceToHost, stream));
    off += (size_t)ns[i];
    rcv[i] = stage_h + off;
    off += (size_t)nr[i];
  }
  CHECK(sync());
  if (comm.exchange(comm.user, n, peer.data(), snd.data(), ns.data(), rcv.data(), nr.data(), nullptr)) {
    set_error("exchange callback failed"); return -101;
  }
  for (int i = 0; i < n; ++i)
    if (nr[i]) AGMG_HIP_CHECK(hipMemcpyAsync(msgs[i].recv, rcv[i], nr[i] * sizeof(double), hipMemcpyHostToDevice, stream));
  // the staging buffer is reused by the next exchange: the uploads must have left it
  return sync();
}

// number of parent patches containing each fine node, summed over ranks (restriction weights)
int agmg_ctx::global_multiplicity(const Exchange &x, const std::vector<uint32_t> &snh, std::vector<double> &mult) {
  if (!x.active()) return 0;
  const size_t nn = mult.size();
  double *d;
  CHECK(dalloc(&d, nn));
  int rc = h2d(d, mult.data(), nn);
  std::vector<Msg> m;
  if (!rc && x.n_ghost) rc = vec_copy(stream, x.ghostbuf, d + x.n_owned, x.n_ghost);
  if (!rc) { add_compress(m, x, 1); rc = exchange(m); }
  std::vector<double> got(x.n_export);
  if (!rc && x.n_export) rc = d2h(got.data(), x.expbuf, x.n_export);
  if (!rc) {
    for (int k = 0; k < x.n_export; ++k) mult[snh[k]] += got[k];
    rc = h2d(d, mult.data(), nn);
  }
  m.clear();
  if (!rc) rc = add_update(m, x, d, 1);
  if (!rc) rc = exchange(m);
  if (!rc) rc = d2h(mult.data(), d, nn);
  dfree(d);
  return rc;
}

int agmg_ctx::ensure_transfers() {
  const int dim = opt.dim, nch = 1 << dim;
  for (int l = 1; l < nl(); ++l) {
    DeviceLevel &F = L[l];
    if (F.has_transfer) continue;
    if (!F.set || !L[l - 1].set) { set_error("ensure_transfers: level not set"); return -1; }
    const HostLevel &HF = H[l];
    if (HF.parent.empty()) { set_error("level " + std::to_string(l) + " has no parent_cell/child_index table"); return -1; }
    const DeviceLevel &C = L[l - 1];
    // the first distributed level above the replicated ones: its parent table indexes the whole coarser level, the coarse
    // cells of other ranks have no child here
    const bool mixed = dist() && C.replicated && !F.replicated;
    if (dist() && F.replicated && !C.replicated) { set_error("replicated levels must be the coarsest ones"); return -1; }
    std::vector<uint32_t> children((size_t)nch * C.cstride, NO_CHILD);
    for (int f = 0; f < F.n_cells; ++f) {
      const int p = HF.parent[f], ch = HF.child[f];
      if (p < 0 || p >= C.n_cells || ch >= nch) { set_error("bad parent_cell/child_index"); return -1; }
      children[(size_t)ch * C.cstride + p] = f;
    }
    // a coarse cell has all its children or (local smoothing on an adaptive mesh: the cell is active on its level) none
    for (int p = 0; p < C.n_cells; ++p) {
      int n_kids = 0;
      for (int ch = 0; ch < nch; ++ch) n_kids += children[(size_t)ch * C.cstride + p] != NO_CHILD;
      if (n_kids != nch && !(n_kids == 0 && (ad.on || mixed))) { set_error("parent cell without 2^dim children (cells that are active on a coarse level need the active mesh set with AGMG_ACTIVE first)"); return -1; }
    }
    for (int p = C.n_cells; p < C.cstride; ++p)
      for (int ch = 0; ch < nch; ++ch) children[(size_t)ch * C.cstride + p] = 0;
    // The kernels walk every coarse cell.  A cell without children is given child 0 of the level in the DEVICE table (a
    // valid cell): the prolongation writes a fine node only from its designated parent, which such a cell never is, and
    // what the restriction leaves in its slots is never pulled (tables below).
    bool childless = false;
    std::vector<uint8_t> refined(C.n_cells, 1);
    std::vector<uint32_t> children_dev(children);
    for (int p = 0; p < C.n_cells; ++p)
      if (children[p] == NO_CHILD) {
        childless = true;
        refined[p] = 0;
        for (int ch = 0; ch < nch; ++ch) children_dev[(size_t)ch * C.cstride + p] = 0;
      }
    if (ad.on || mixed)
      for (int vel = 0; vel < 2; ++vel) {
        dfree(ad.rpull_ptr[vel ? 0 : 1][l]); dfree(ad.rpull_list[vel ? 0 : 1][l]);
        if (!childless) continue;
        const int nloc = ipow(vel ? 3 : 2, dim), nn = vel ? C.n_nodes_v : C.n_nodes_p;
        const std::vector<uint32_t> &cnc = vel ? H[l - 1].cnv : H[l - 1].cnp;
        std::vector<uint32_t> ptr((size_t)nn + 1, 0), list;
        for (int p = 0; p < C.n_cells; ++p)
          if (refined[p]) for (int a = 0; a < nloc; ++a) ++ptr[cnc[(size_t)p * nloc + a] + 1];
        for (int i = 0; i < nn; ++i) ptr[i + 1] += ptr[i];
        list.resize(ptr[nn]);
        std::vector<uint32_t> fill(ptr.begin(), ptr.end() - 1);
        for (int p = 0; p < C.n_cells; ++p)
          if (refined[p]) for (int a = 0; a < nloc; ++a) list[fill[cnc[(size_t)p * nloc + a]]++] = (uint32_t)((size_t)a * C.cstride + p);
        if (list.empty()) list.push_back(0);
        CHECK(upload(&ad.rpull_ptr[vel ? 0 : 1][l], ptr));
        CHECK(upload(&ad.rpull_list[vel ? 0 : 1][l], list));
      }
    CHECK(upload(&F.childrenT, children_dev));
    F.bytes_transfer = children.size() * 4;
    for (int vel = 0; vel < 2; ++vel) {
      const int K = vel ? 2 : 1, n1 = K + 1, nf = 2 * K + 1, nloc = ipow(n1, dim);
      const std::vector<uint32_t> &cn = vel ? HF.cnv : HF.cnp;
      const int nn = vel ? F.n_nodes_v : F.n_nodes_p;
      std::vector<uint32_t> par(nn, 0xFFFFFFFFu);
      std::vector<uint8_t> pos(nn, 0);
      // multiplicity = occurrences (parent, patch position) of a fine node (on a periodic mesh one
      // cell wide both faces of a patch are the same fine nodes and count twice)
      std::vector<double> mult(nn, 0.0);
      std::vector<int> seen(ipow(nf, dim), -1);
      std::vector<uint32_t> own_mask(vel ? (size_t)4 * C.cstride : 0, 0u);   // (5^3 = 125 patch positions: 4 words)
      for (int p = 0; p < C.n_cells; ++p)
        for (int ch = 0; ch < nch; ++ch) {
          const uint32_t f = children[(size_t)ch * C.cstride + p];
          if (f == NO_CHILD) continue;   // (host table: cells without children take no part)
          for (int i = 0; i < nloc; ++i) {
            int rem = i, pidx = 0, stride = 1;
            for (int d = 0; d < dim; ++d) {
              const int id = rem % n1;
              rem /= n1;
              pidx += (id + K * ((ch >> d) & 1)) * stride;
              stride *= nf;
            }
            const uint32_t fn = cn[(size_t)f * nloc + i];
            if (seen[pidx] != p) { seen[pidx] = p; mult[fn] += 1.0; }
            if (par[fn] == 0xFFFFFFFFu) {
              par[fn] = p; pos[fn] = (uint8_t)pidx;
              if (vel) own_mask[(size_t)(pidx >> 5) * C.cstride + p] |= 1u << (pidx & 31);
            }
          }
        }
      for (int i = 0; i < nn; ++i)
        if (par[i] == 0xFFFFFFFFu) { par[i] = 0; mult[i] = 1.0; }  // node not touched by any cell
      if (dist()) CHECK(global_multiplicity(vel ? F.xv : F.xp, send_nodes_h[vel ? 0 : 1][l], mult));
      for (int i = 0; i < nn; ++i) mult[i] = 1.0 / mult[i];
      if (vel) { CHECK(upload(&F.prol_parent_v, par)); CHECK(upload(&F.prol_pos_v, pos)); CHECK(upload(&F.inv_mult_v, mult)); dfree(F.prol_mask_v); CHECK(upload(&F.prol_mask_v, own_mask)); }
      else { CHECK(upload(&F.prol_parent_p, par)); CHECK(upload(&F.prol_pos_p, pos)); CHECK(upload(&F.inv_mult_p, mult)); }
      F.bytes_transfer += (size_t)nn * (4 + 1 + 8);
    }
    F.has_transfer = true;
  }
  return 0;
}

// ===================================================================== operators
int agmg_ctx::apply_level(int which, int l, const CoeffData &cd, const double *x, const EpiArgs &e) {
  const DeviceLevel &lv = L[l];
  if (!ldist(lv)) {
    if (which == 0) return op_velocity(lv, cd, AGMG_OP_A, x, nullptr, e);
    if (mass_ell(lv, cd)) return launch_mass_ell(stream, lv, cd, x, e);
    CHECK(launch_mass_cells(stream, lv, cd, x, slots_p));
    return launch_pull_p(stream, lv, slots_p, 1, e);
  }
  // distributed: src.update_ghost_values(); cell loop; dst.compress(add); epilogue on the owned rows
  if (which == 0 && sym_now_A && a_sym(lv, cd) && (e.epi == EPI_STORE || e.epi == EPI_RESID || e.epi == EPI_CHEB))
    return apply_A_sym(lv, cd, x, e);
  if (which == 0 && bricks_on(lv) && brickq_supported(lv, cd, AGMG_OP_A) &&
      (e.epi == EPI_STORE || e.epi == EPI_RESID || e.epi == EPI_CHEB)) {
    // brick levels: ghost import, the brick kernel (shares of ghost nodes and of rows another brick owns go to partial
    // slots), the ghost sums to their owners, then the fix-up of the owned rows with the received sums
    if (use_overlap && cstream && lv.bk.n_interior > 0 && lv.xv.active()) {
      CHECK(begin_import([&] { return update_ghosts(lv.xv, x); }));
      CHECK(launch_op_bricks(stream, lv, cd, AGMG_OP_A, x, nullptr, slots_u, slots_p, e, nullptr, PH_INTERIOR));
      CHECK(end_import());
      CHECK(launch_op_bricks(stream, lv, cd, AGMG_OP_A, x, nullptr, slots_u, slots_p, e, nullptr, PH_INTERFACE));
    } else {
      CHECK(update_ghosts(lv.xv, x));
      CHECK(launch_op_bricks(stream, lv, cd, AGMG_OP_A, x, nullptr, slots_u, slots_p, e, nullptr, PH_GHOST));
    }
    CHECK(compress(lv.xv));
    return launch_op_bricks(stream, lv, cd, AGMG_OP_A, x, nullptr, slots_u, slots_p, e, nullptr, PH_OWNED);
  }
  if (which == 0) {
    if (can_overlap(lv, e, cd, AGMG_OP_A)) {
      CHECK(begin_import([&] { return update_ghosts(lv.xv, x); }));
      CHECK(launch_op(stream, lv, cd, AGMG_OP_A, x, nullptr, slots_u, nullptr, e, use_blocks, PH_INTERIOR));
      CHECK(end_import());
      CHECK(launch_op(stream, lv, cd, AGMG_OP_A, x, nullptr, slots_u, nullptr, e, use_blocks, PH_INTERFACE));
    } else {
      CHECK(update_ghosts(lv.xv, x));
      CHECK(launch_op(stream, lv, cd, AGMG_OP_A, x, nullptr, slots_u, nullptr, e, use_blocks, PH_GHOST));
    }
    CHECK(compress(lv.xv));
    return launch_op(stream, lv, cd, AGMG_OP_A, x, nullptr, slots_u, nullptr, e, use_blocks, PH_OWNED);
  }
  if (sym_now && mass_sym(lv, cd)) return apply_mass_sym(lv, cd, x, e);
  CHECK(update_ghosts(lv.xp, x));
  if (mass_ell(lv, cd)) {
    // assembled rows of the rank's own cells: the ghost rows are the shares their owners still miss
    CHECK(launch_mass_ell(stream, lv, cd, x, e, PH_GHOST));
    CHECK(compress(lv.xp));
    return launch_mass_ell(stream, lv, cd, x, e, PH_OWNED);
  }
  CHECK(launch_mass_cells(stream, lv, cd, x, slots_p));
  CHECK(launch_pull_p(stream, lv, slots_p, 1, e, PH_GHOST));
  CHECK(compress(lv.xp));
  return launch_pull_p(stream, lv, slots_p, 1, e, PH_OWNED);
}

// Mass operator of a distributed level in ONE message round (SymExchange): the source carries valid ghost entries (the
// smoother keeps them valid: every rank that holds a row applies the update to its own copy), every rank computes the
// shares of the rows others hold too, sends them to all of those, adds what it receives in rank order and runs the
// epilogue on EVERY local row, owned or ghost.  b, dinv and xold of the epilogue need valid ghost entries as well.
int agmg_ctx::apply_mass_sym(const DeviceLevel &lv, const CoeffData &cd, const double *x, const EpiArgs &e) {
  const SymExchange &sx = lv.sp;
  CHECK(launch_mass_ell_sym(stream, lv, cd, x, e, 0));
  std::vector<Msg> m;
  for (size_t k = 0; k < sx.nbr.size(); ++k) {
    const int64_t n = sx.off[k + 1] - sx.off[k];
    m.push_back({sx.nbr[k], sx.sendbuf + sx.off[k], n, sx.recvbuf + sx.off[k], n});
  }
  CHECK(exchange(m));
  ++n_sym_applications;
  return launch_mass_ell_sym(stream, lv, cd, x, e, 1);
}

// The A block of a distributed level in one message round: the cell phase of every block (rows inside a block are
// finished there), the shares of the rows other ranks hold too, one exchange, then every shared row -- owned or ghost --
// from its shares in rank order.  Source, b, dinv and xold carry valid ghost entries (the smoother sees to that).
int agmg_ctx::apply_A_sym(const DeviceLevel &lv, const CoeffData &cd, const double *x, const EpiArgs &e) {
  const SymExchange &sx = lv.sv;
  CHECK(launch_op(stream, lv, cd, AGMG_OP_A, x, nullptr, slots_u, nullptr, e, use_blocks, PH_CELLS));
  CHECK(launch_fix_sym(stream, lv, slots_u, e, 0));
  std::vector<Msg> m;
  for (size_t k = 0; k < sx.nbr.size(); ++k) {
    const int64_t n = (int64_t)(sx.off[k + 1] - sx.off[k]) * lv.dim;
    m.push_back({sx.nbr[k], sx.sendbuf + (size_t)sx.off[k] * lv.dim, n, sx.recvbuf + (size_t)sx.off[k] * lv.dim, n});
  }
  CHECK(exchange(m));
  ++n_sym_applications;
  return launch_fix_sym(stream, lv, slots_u, e, 1);
}

// one-rank level operators on given tables: which01 0 = A block, 1 = mass
int agmg_ctx::apply_on(const DeviceLevel &lv, int which01, const CoeffData &cd, const double *x, const EpiArgs &e) {
  if (which01 == 0) return op_velocity(lv, cd, AGMG_OP_A, x, nullptr, e);
  if (mass_ell(lv, cd)) return launch_mass_ell(stream, lv, cd, x, e);
  CHECK(launch_mass_cells(stream, lv, cd, x, slots_p));
  return launch_pull_p(stream, lv, slots_p, 1, e);
}

// Operators of the active mesh of an adaptive context.  MatrixFree resolves the hanging-node constraints when it reads
// cell values and applies their transpose when it writes (read_dof_values / distribute_local_to_global); constrained
// rows are identity rows: y = C^T A C x around the device operators, which treat every node as its own DoF.
// C and C^T are index kernels over the constraint CSR and its transpose (vec_hang_resolve / _distribute / _restore); the
// vector never leaves the device.
// src: [u], [p] or [u; p] as `which` asks; b != NULL: dst = b - op(src).
int agmg_ctx::active_apply(int which, double *dst, const double *src, const double *b) {
  const DeviceLevel &lv = ad.act;
  const CoeffData &cd = active_coeff();
  const int dim = lv.dim;
  const size_t nu = lv.n_dofs_v(), np = lv.n_dofs_p();
  const bool has_u = which != AGMG_OP_MASS, has_p = which != AGMG_OP_A;
  const size_t poff = has_u ? nu : 0, n = (has_u ? nu : 0) + (has_p ? np : 0);
  if (ad.xres_cap < nu + np) { dfree(ad.xres); CHECK(dalloc(&ad.xres, nu + np)); ad.xres_cap = nu + np; }
  double *xr = ad.xres;
  CHECK(vec_copy(stream, xr, src, n));
  if (has_u) CHECK(vec_hang_resolve(stream, xr, ad.hang[0], lv.cmask_v, dim));
  if (has_p) CHECK(vec_hang_resolve(stream, xr + poff, ad.hang[1], lv.cmask_p, 1));
  EpiArgs e;
  e.epi = EPI_STORE; e.dst = dst; e.x = xr;
  switch (which) {
    case AGMG_OP_A: CHECK(op_velocity(lv, cd, AGMG_OP_A, xr, nullptr, e)); break;
    case AGMG_OP_MASS:
      CHECK(launch_mass_cells(stream, lv, cd, xr, slots_p));
      CHECK(launch_pull_p(stream, lv, slots_p, 1, e));
      break;
    case AGMG_OP_STOKES: {
      EpiArgs ep = e;
      ep.dst = dst + nu; ep.x = xr + nu;
      CHECK(op_velocity(lv, cd, AGMG_OP_STOKES, xr, xr + nu, e, &ep));
      break;
    }
    default:
      CHECK(op_velocity(lv, cd, AGMG_OP_BT, nullptr, xr + nu, e));
      CHECK(vec_set(stream, dst + nu, 0.0, np));
  }
  if (has_u) {
    CHECK(vec_hang_distribute(stream, dst, ad.hang[0], lv.cmask_v, dim));
    CHECK(vec_hang_restore(stream, dst, ad.hang[0], dim));
  }
  if (which == AGMG_OP_MASS || which == AGMG_OP_STOKES) {   // (B^T writes no pressure rows)
    CHECK(vec_hang_distribute(stream, dst + poff, ad.hang[1], lv.cmask_p, 1));
    CHECK(vec_hang_restore(stream, dst + poff, ad.hang[1], 1));
  }
  if (b) CHECK(vec_lincomb(stream, dst, 1.0, b, -1.0, dst, n));
  return 0;
}

// dst = b - B^T p on the active mesh (velocity rows; stokes_matrix_free.h:299-303)
int agmg_ctx::bt_residual(double *dst, const double *b, const double *p) {
  if (ad.on) {
    // [0; p] through the B^T operator of the active mesh (the identity rows of the hanging velocity nodes read the zero
    // block), velocity rows kept; pool vector 4 is free while the block preconditioner forms this residual
    const DeviceLevel &la = ad.act;
    const size_t nu = la.n_dofs_v(), np = la.n_dofs_p();
    double *in, *out;
    CHECK(get_vec(3, &in)); CHECK(get_vec(4, &out));
    CHECK(vec_set(stream, in, 0.0, nu));
    CHECK(vec_copy(stream, in + nu, p, np));
    CHECK(active_apply(AGMG_OP_BT, out, in, nullptr));
    if (b) return vec_lincomb(stream, dst, 1.0, b, -1.0, out, nu);
    return vec_copy(stream, dst, out, nu);
  }
  const DeviceLevel &lv = fin();
  EpiArgs e;
  e.epi = b ? EPI_RESID : EPI_STORE; e.dst = dst; e.b = b; e.x = nullptr;
  if (!dist()) return op_velocity(lv, active_coeff(), AGMG_OP_BT, nullptr, p, e);
  if (can_overlap(lv, e, active_coeff(), AGMG_OP_BT)) {
    CHECK(begin_import([&] { return update_ghosts(lv.xp, p); }));
    CHECK(launch_op(stream, lv, active_coeff(), AGMG_OP_BT, nullptr, p, slots_u, nullptr, e, use_blocks, PH_INTERIOR));
    CHECK(end_import());
    CHECK(launch_op(stream, lv, active_coeff(), AGMG_OP_BT, nullptr, p, slots_u, nullptr, e, use_blocks, PH_INTERFACE));
  } else {
    CHECK(update_ghosts(lv.xp, p));
    CHECK(launch_op(stream, lv, active_coeff(), AGMG_OP_BT, nullptr, p, slots_u, nullptr, e, use_blocks, PH_GHOST));
  }
  CHECK(compress(lv.xv));
  return launch_op(stream, lv, active_coeff(), AGMG_OP_BT, nullptr, p, slots_u, nullptr, e, use_blocks, PH_OWNED);
}

// defect[l-1] (epilogue e on level l-1) from R * src_fine, src_fine on level l
int agmg_ctx::restrict_to(int which, int l, const double *src_fine, const EpiArgs &e) {
  const DeviceLevel &F = L[l], &C = L[l - 1];
  const bool vel = which == 0;
  double *slots = vel ? slots_u : slots_p;
  if (dist() && !fine_ghosts_valid) CHECK(update_ghosts(vel ? F.xv : F.xp, src_fine));
  CHECK(launch_restrict_cells(stream, F, C, vel, src_fine, slots));
  const bool mixed = dist() && C.replicated && !F.replicated;
  if ((ad.on || mixed) && l < (int)ad.rpull_ptr[vel ? 0 : 1].size() && ad.rpull_ptr[vel ? 0 : 1][l]) {
    // local smoothing, or a replicated coarse level under a distributed one: the slots of the coarse cells without
    // children (here) hold nothing to pull (ensure_transfers)
    DeviceLevel Cr = C;
    if (vel) { Cr.pull_ptr_v = ad.rpull_ptr[0][l]; Cr.pull_list_v = ad.rpull_list[0][l]; }
    else { Cr.pull_ptr_p = ad.rpull_ptr[1][l]; Cr.pull_list_p = ad.rpull_list[1][l]; }
    CHECK(vel ? launch_pull_v(stream, Cr, slots, e) : launch_pull_p(stream, Cr, slots, 1, e));
    if (!mixed) return 0;
  } else if (mixed) {
    CHECK(vel ? launch_pull_v(stream, C, slots, e) : launch_pull_p(stream, C, slots, 1, e));
  }
  if (mixed) {
    // every rank restricted the cells it owns on the finer level: the defect of the replicated level is their sum
    if (e.epi != EPI_STORE || e.x) { set_error("restriction onto a replicated level: store epilogue only"); return -1; }
    return allreduce_vec(e.dst, vel ? (size_t)C.n_dofs_v() : (size_t)C.n_dofs_p());
  }
  if (!ldist(C)) return vel ? launch_pull_v(stream, C, slots, e) : launch_pull_p(stream, C, slots, 1, e);
  CHECK(vel ? launch_pull_v(stream, C, slots, e, PH_GHOST) : launch_pull_p(stream, C, slots, 1, e, PH_GHOST));
  CHECK(compress(vel ? C.xv : C.xp));
  return vel ? launch_pull_v(stream, C, slots, e, PH_OWNED) : launch_pull_p(stream, C, slots, 1, e, PH_OWNED);
}
int agmg_ctx::prolongate_from(int which, int l, double *dst_fine, const double *src_coarse) {
  const DeviceLevel &F = L[l], &C = L[l - 1];
  const bool vel = which == 0;
  if (dist()) CHECK(update_ghosts(vel ? C.xv : C.xp, src_coarse));
  return launch_prolongate_add(stream, F, C, vel, dst_fine, src_coarse);
}

// Stokes operator on the active mesh; epilogue kind/b taken from eu_in (STORE or RESID)
int agmg_ctx::stokes_apply(const double *src, const EpiArgs &eu_in, double *dst) {
  if (ad.on) return active_apply(AGMG_OP_STOKES, dst, src, eu_in.epi == EPI_RESID ? eu_in.b : nullptr);
  const DeviceLevel &lv = fin();
  const size_t nu = lv.n_dofs_v();
  EpiArgs eu = eu_in, ep = eu_in;
  eu.dst = dst; eu.x = src;
  ep.dst = dst + nu; ep.x = src + nu;
  if (eu_in.b) ep.b = eu_in.b + nu;
  if (!dist()) {
    CHECK(op_velocity(lv, active_coeff(), AGMG_OP_STOKES, src, src + nu, eu, &ep));
    // boundary-face loop of StokesOperator::apply_add (matrix_free_operators.cc:343-353)
    return launch_face_term(stream, lv.dim, faces, lv.cmask_v, src, dst, eu_in.epi == EPI_RESID ? -1.0 : 1.0);
  }
  std::vector<Msg> m;
  auto import = [&]() -> int {
    CHECK(add_update(m, lv.xv, const_cast<double *>(src), lv.xv.nc));
    CHECK(add_update(m, lv.xp, const_cast<double *>(src) + nu, 1));
    return exchange(m);
  };
  // (the pressure rows of a phase ride in the launch of the velocity fix-up of that phase)
  if (can_overlap(lv, eu, active_coeff(), AGMG_OP_STOKES)) {
    CHECK(begin_import(import));
    CHECK(launch_op(stream, lv, active_coeff(), AGMG_OP_STOKES, src, src + nu, slots_u, slots_p, eu, use_blocks, PH_INTERIOR));
    CHECK(end_import());
    CHECK(launch_op(stream, lv, active_coeff(), AGMG_OP_STOKES, src, src + nu, slots_u, slots_p, eu, use_blocks, PH_INTERFACE, &ep));
  } else {
    CHECK(import());
    CHECK(launch_op(stream, lv, active_coeff(), AGMG_OP_STOKES, src, src + nu, slots_u, slots_p, eu, use_blocks, PH_GHOST, &ep));
  }
  m.clear();
  add_compress(m, lv.xv, lv.xv.nc);
  add_compress(m, lv.xp, 1);
  CHECK(exchange(m));
  CHECK(launch_op(stream, lv, active_coeff(), AGMG_OP_STOKES, src, src + nu, slots_u, slots_p, eu, use_blocks, PH_OWNED, &ep));
  if (!faces.global_active) return 0;
  // boundary-face loop (matrix_free_operators.cc:343-353): the sums of the local faces go to the owned
  // rows directly; those of ghost rows travel to their owners in one more compress step
  const double sign = eu_in.epi == EPI_RESID ? -1.0 : 1.0;
  if (lv.xv.n_ghost) AGMG_HIP_CHECK(hipMemsetAsync(lv.xv.ghostbuf, 0, (size_t)lv.xv.n_ghost * lv.xv.nc * sizeof(double), stream));
  CHECK(launch_face_term(stream, lv.dim, faces, lv.cmask_v, src, dst, sign, lv.xv.ghostbuf, lv.n_owned_v));
  CHECK(compress(lv.xv));
  return launch_face_add_received(stream, lv.dim, faces, lv.cmask_v, lv.xv.expbuf, dst, sign);
}
int agmg_ctx::stokes_vmult(double *dst, const double *src) {
  EpiArgs e;
  e.epi = EPI_STORE;
  return stokes_apply(src, e, dst);
}
int agmg_ctx::stokes_residual(double *dst, const double *rhs, const double *x) {
  EpiArgs e;
  e.epi = EPI_RESID;
  e.b = rhs;
  return stokes_apply(x, e, dst);
}

// ===================================================================== Chebyshev
int agmg_ctx::cheb_vmult(int which, int l, double *&x, double *&xo, const double *b) {
  const Smoother &s = sm[which][l];
  const size_t n = size_of(which, l);
  const double *dinv = which == 0 ? L[l].inv_diag_A : L[l].inv_diag_M;
  CHECK(vec_pointwise_scale(stream, x, 1.0 / s.theta, dinv, b, n));
  if (s.degree < 2 || std::fabs(s.delta) < 1e-40) return 0;
  double rhok = s.delta / s.theta;
  const double sigma = s.theta / s.delta;
  for (int k = 0; k < s.degree - 1; ++k) {
    const double rhokp = 1.0 / (2.0 * sigma - rhok);
    EpiArgs e;
    e.epi = EPI_CHEB; e.dst = xo; e.x = x; e.b = b; e.dinv = dinv;
    e.xold = k == 0 ? nullptr : xo;
    e.f1 = rhokp * rhok; e.f2 = 2.0 * rhokp / s.delta;
    rhok = rhokp;
    CHECK(apply_level(which, l, L[l].coeff, x, e));
    std::swap(x, xo);
  }
  return 0;
}
int agmg_ctx::cheb_step(int which, int l, double *&x, double *&xo, const double *b) {
  const Smoother &s = sm[which][l];
  const double *dinv = which == 0 ? L[l].inv_diag_A : L[l].inv_diag_M;
  {
    EpiArgs e;
    e.epi = EPI_CHEB; e.dst = xo; e.x = x; e.b = b; e.dinv = dinv; e.xold = nullptr; e.f1 = 0.0; e.f2 = 1.0 / s.theta;
    CHECK(apply_level(which, l, L[l].coeff, x, e));
    std::swap(x, xo);
  }
  if (s.degree < 2 || std::fabs(s.delta) < 1e-40) return 0;
  double rhok = s.delta / s.theta;
  const double sigma = s.theta / s.delta;
  for (int k = 0; k < s.degree - 1; ++k) {
    const double rhokp = 1.0 / (2.0 * sigma - rhok);
    EpiArgs e;
    e.epi = EPI_CHEB; e.dst = xo; e.x = x; e.b = b; e.dinv = dinv; e.xold = xo;
    e.f1 = rhokp * rhok; e.f2 = 2.0 * rhokp / s.delta;
    rhok = rhokp;
    CHECK(apply_level(which, l, L[l].coeff, x, e));
    std::swap(x, xo);
  }
  return 0;
}

void agmg_ctx::program_vmult(const Smoother &s, SmootherProgram &p) const {
  p.s[p.n++] = {0.0, 1.0 / s.theta, SM_POINTWISE, 0};
  if (s.degree < 2 || std::fabs(s.delta) < 1e-40) return;
  double rhok = s.delta / s.theta;
  const double sigma = s.theta / s.delta;
  for (int k = 0; k < s.degree - 1; ++k) {
    const double rhokp = 1.0 / (2.0 * sigma - rhok);
    p.s[p.n++] = {rhokp * rhok, 2.0 * rhokp / s.delta, SM_CHEB, k == 0 ? 0 : 1};
    rhok = rhokp;
  }
}
void agmg_ctx::program_step(const Smoother &s, SmootherProgram &p) const {
  p.s[p.n++] = {0.0, 1.0 / s.theta, SM_CHEB, 0};
  if (s.degree < 2 || std::fabs(s.delta) < 1e-40) return;
  double rhok = s.delta / s.theta;
  const double sigma = s.theta / s.delta;
  for (int k = 0; k < s.degree - 1; ++k) {
    const double rhokp = 1.0 / (2.0 * sigma - rhok);
    p.s[p.n++] = {rhokp * rhok, 2.0 * rhokp / s.delta, SM_CHEB, 1};
    rhok = rhokp;
  }
}

// cheb_vmult (zero_start) or cheb_step of the A block as one launch on a single-block level; resid_dst: also
// resid_dst = rhs - A x behind it (the defect Multigrid::level_v_step restricts).  Same steps, vectors and swaps as the
// launch-per-step forms above.
int agmg_ctx::smooth_A_seq(int l, bool zero_start, const double *rhs, double *resid_dst) {
  const Smoother &sp = sm[0][l];
  double *&x = mg_sol[0][l], *&xo = mg_tmp[0][l];
  SeqArgs q;
  q.n_dofs = (int)size_of(0, l);
  q.b = rhs; q.dinv = L[l].inv_diag_A;
  auto push = [&](int kind, double f1, double f2, const double *xin, const double *xold, double *dst) {
    SeqStep &st = q.s[q.n++];
    st.kind = kind; st.f1 = f1; st.f2 = f2; st.x = xin; st.xold = xold; st.dst = dst;
  };
  const bool recur = !(sp.degree < 2 || std::fabs(sp.delta) < 1e-40);
  if (sp.degree + 2 > SEQ_MAX_STEPS) { set_error("smooth_A_seq: smoother degree too large"); return -1; }
  if (zero_start) push(SEQ_POINTWISE, 0.0, 1.0 / sp.theta, nullptr, nullptr, x);
  else { push(SEQ_CHEB, 0.0, 1.0 / sp.theta, x, nullptr, xo); std::swap(x, xo); }
  if (recur) {
    double rhok = sp.delta / sp.theta;
    const double sigma = sp.theta / sp.delta;
    for (int k = 0; k < sp.degree - 1; ++k) {
      const double rhokp = 1.0 / (2.0 * sigma - rhok);
      push(SEQ_CHEB, rhokp * rhok, 2.0 * rhokp / sp.delta, x, (zero_start && k == 0) ? nullptr : xo, xo);
      rhok = rhokp;
      std::swap(x, xo);
    }
  }
  if (resid_dst) push(SEQ_RESID, 0.0, 0.0, x, nullptr, resid_dst);
  return launch_op_seq(stream, L[l], L[l].coeff, q, slots_u);
}

// mg::SmootherRelaxation(steps): apply = vmult + (steps-1) step; smooth = steps x step
namespace {
struct FlagGuard {   // sets a flag for the life of a scope
  bool &f, was;
  FlagGuard(bool &flag, bool v) : f(flag), was(flag) { f = v; }
  ~FlagGuard() { f = was; }
};
}  // namespace

int agmg_ctx::smoother_apply(int which, int l, const double *rhs) {
  const int steps = which == 0 ? 1 : opt.schur_smoother_steps;
  const bool sym = which == 1 && mass_sym(L[l], L[l].coeff), symA = which == 0 && a_sym(L[l], L[l].coeff);
  if (sym || symA) {   // (the iterate starts from dinv * rhs on every local entry)
    CHECK(update_ghosts(which == 0 ? L[l].xv : L[l].xp, rhs));
    if ((int)sym_rhs_valid.size() != 2 * nl()) sym_rhs_valid.assign(2 * nl(), nullptr);
    sym_rhs_valid[2 * l + which] = rhs;
  }
  FlagGuard guard(sym_now, sym), guardA(sym_now_A, symA);
  if (which == 1 && fused_mass(l) && steps * sm[1][l].degree + 1 <= SM_MAX_STEPS) {
    SmootherProgram p;
    program_vmult(sm[1][l], p);
    for (int i = 1; i < steps; ++i) program_step(sm[1][l], p);
    return launch_mass_smoother(stream, L[l], L[l].coeff, p, L[l].inv_diag_M, rhs, nullptr, mg_sol[1][l], nullptr, mass_ell(L[l], L[l].coeff));
  }
  if (which == 0 && fused_A(l)) return smooth_A_seq(l, true, rhs, nullptr);
  CHECK(cheb_vmult(which, l, mg_sol[which][l], mg_tmp[which][l], rhs));
  for (int i = 1; i < steps; ++i) CHECK(cheb_step(which, l, mg_sol[which][l], mg_tmp[which][l], rhs));
  return 0;
}
int agmg_ctx::smoother_smooth(int which, int l, const double *rhs) {
  const int steps = which == 0 ? 1 : opt.schur_smoother_steps;
  const bool sym = which == 1 && mass_sym(L[l], L[l].coeff), symA = which == 0 && a_sym(L[l], L[l].coeff);
  if (sym || symA) {   // the right-hand side and the iterate (coarse-grid correction added) with valid ghost entries
    const Exchange &x = which == 0 ? L[l].xv : L[l].xp;
    if ((int)sym_rhs_valid.size() != 2 * nl() || sym_rhs_valid[2 * l + which] != rhs) CHECK(update_ghosts(x, rhs));
    CHECK(update_ghosts(x, mg_sol[which][l]));
  }
  FlagGuard guard(sym_now, sym), guardA(sym_now_A, symA);
  if (which == 1 && fused_mass(l) && steps * sm[1][l].degree + 1 <= SM_MAX_STEPS) {
    SmootherProgram p;
    for (int i = 0; i < steps; ++i) program_step(sm[1][l], p);
    return launch_mass_smoother(stream, L[l], L[l].coeff, p, L[l].inv_diag_M, rhs, mg_sol[1][l], mg_sol[1][l], nullptr, mass_ell(L[l], L[l].coeff));
  }
  if (which == 0 && fused_A(l)) return smooth_A_seq(l, false, rhs, nullptr);
  for (int i = 0; i < steps; ++i) CHECK(cheb_step(which, l, mg_sol[which][l], mg_tmp[which][l], rhs));
  return 0;
}

namespace {
void coarse_host_fn(void *p) {
  auto *job = static_cast<agmg_ctx::CoarseJob *>(p);
  const int n = job->c->coarse[job->which].n;
  job->c->coarse_host_compute(job->which, job->c->coarse_pin[job->which], job->c->coarse_pin[job->which] + n);
}
}  // namespace

void agmg_ctx::coarse_host_compute(int which, const double *b, double *xout) const {
  const CoarseOp &co = coarse[which];
  const Smoother &s = sm[which][0];
  const int n = co.n, steps = which == 0 ? 1 : opt.schur_smoother_steps;
  std::vector<double> x(n), xo(n, 0.0), t(n);
  auto matvec = [&](std::vector<double> &y, const std::vector<double> &v) {
    for (int i = 0; i < n; ++i) {
      double acc = 0;
      const double *row = &co.A[(size_t)i * n];
      for (int j = 0; j < n; ++j) acc += row[j] * v[j];
      y[i] = acc;
    }
  };
  auto update = [&](bool first, double f1, double f2) {
    for (int i = 0; i < n; ++i) {
      double v = (1.0 + f1) * x[i] + f2 * co.dinv[i] * (b[i] - t[i]);
      if (!first) v -= f1 * xo[i];
      xo[i] = v;
    }
    x.swap(xo);
  };
  auto recurrence = [&](bool zero_start) {
    if (s.degree < 2 || std::fabs(s.delta) < 1e-40) return;
    double rhok = s.delta / s.theta;
    const double sigma = s.theta / s.delta;
    for (int k = 0; k < s.degree - 1; ++k) {
      matvec(t, x);
      const double rhokp = 1.0 / (2.0 * sigma - rhok);
      const double f1 = rhokp * rhok, f2 = 2.0 * rhokp / s.delta;
      rhok = rhokp;
      update(zero_start && k == 0, f1, f2);
    }
  };
  for (int i = 0; i < n; ++i) x[i] = (1.0 / s.theta) * co.dinv[i] * b[i];
  recurrence(true);
  for (int st = 1; st < steps; ++st) {
    matvec(t, x);
    update(true, 0.0, 1.0 / s.theta);
    recurrence(false);
  }
  for (int i = 0; i < n; ++i) xout[i] = x[i];
}

// level 0: MGCoarseGridApplySmoother = smoother apply (local_smoothing.cc:999-1004)
int agmg_ctx::coarse_solve(int which) {
  if (!coarse_host(which)) return smoother_apply(which, 0, mg_def[which][0]);
  // defect to the host, Jacobi-Chebyshev on the dense operator there, solution back: three stream-ordered nodes
  const size_t n = (size_t)coarse[which].n;
  if (coarse_pin_cap[which] < 2 * n) { set_error("coarse_solve: staging not allocated (setup_smoothers)"); return -1; }
  coarse_job[which] = {this, which};
  AGMG_HIP_CHECK(hipMemcpyAsync(coarse_pin[which], mg_def[which][0], n * sizeof(double), hipMemcpyDeviceToHost, stream));
  AGMG_HIP_CHECK(hipLaunchHostFunc(stream, coarse_host_fn, &coarse_job[which]));
  AGMG_HIP_CHECK(hipMemcpyAsync(mg_sol[which][0], coarse_pin[which] + n, n * sizeof(double), hipMemcpyHostToDevice, stream));
  return 0;
}

// Multigrid::level_v_step; uniformly refined meshes: no edge matrices
int agmg_ctx::level_v_step(int which, int l) {
  if (l == 0) return coarse_solve(which);
  const DeviceLevel &F = L[l];
  const int sm_steps = which == 0 ? 1 : opt.schur_smoother_steps;
  if (has_edge(l)) {
    // Local smoothing, a level with a refinement edge (Multigrid::level_v_step with set_edge_matrices(interface,
    // interface), local_smoothing.cc:1045, 1053).  The smoother leaves the edge entries of the solution at zero (identity
    // rows, zero defect there), so matrix->vmult plus edge_out->vmult_add (vmult_interface_down: the edge rows of the
    // operator applied to the other entries) is ONE application of the operator with the boundary rows alone constrained.
    CHECK(smoother_apply(which, l, mg_def[which][l]));
    EpiArgs e;
    e.epi = EPI_RESID; e.dst = mg_t[which][l]; e.b = mg_def[which][l]; e.x = mg_sol[which][l];
    CHECK(apply_on(ad.Lb[l], which, ad.Lb[l].coeff, mg_sol[which][l], e));
    EpiArgs er;   // (the defect of the coarser level already holds what copy_to_mg put there: restrict_and_add)
    er.epi = EPI_ADD; er.dst = mg_def[which][l - 1]; er.x = nullptr;
    CHECK(restrict_to(which, l, mg_t[which][l], er));
    CHECK(level_v_step(which, l - 1));
    CHECK(prolongate_from(which, l, mg_sol[which][l], mg_sol[which][l - 1]));
    CHECK(interface_up_into_defect(which, l));
    return smoother_smooth(which, l, mg_def[which][l]);
  }
  if (which == 1 && fused_mass(l) && sm_steps * sm[1][l].degree + 1 <= SM_MAX_STEPS) {
    // pre-smoothing and the residual behind it in one launch
    SmootherProgram p;
    program_vmult(sm[1][l], p);
    for (int i = 1; i < sm_steps; ++i) program_step(sm[1][l], p);
    p.s[p.n++] = {0.0, 0.0, SM_RESID, 0};
    CHECK(launch_mass_smoother(stream, F, F.coeff, p, F.inv_diag_M, mg_def[1][l], nullptr, mg_sol[1][l], mg_t[1][l], mass_ell(F, F.coeff)));
  } else if (which == 0 && fused_A(l)) {
    CHECK(smooth_A_seq(l, true, mg_def[0][l], mg_t[0][l]));   // pre-smoothing and t = defect - A solution
  } else {
  CHECK(smoother_apply(which, l, mg_def[which][l]));
  {  // t = defect - A solution
    EpiArgs e;
    e.epi = EPI_RESID; e.dst = mg_t[which][l]; e.b = mg_def[which][l];
    // (behind the symmetric smoother sequence the iterate and the defect carry valid ghost entries: one round here too)
    FlagGuard guard(sym_now, which == 1 && mass_sym(F, F.coeff)), guardA(sym_now_A, which == 0 && a_sym(F, F.coeff));
    CHECK(apply_level(which, l, F.coeff, mg_sol[which][l], e));
  }
  }
  {  // defect[l-1] = R t   (coarse Dirichlet rows dropped); adaptive contexts: += (copy_to_mg filled the coarser defect)
    EpiArgs e;
    e.epi = ad.on ? EPI_ADD : EPI_STORE; e.dst = mg_def[which][l - 1]; e.x = nullptr;
    // (a residual out of a one-round application carries its ghost entries already)
    FlagGuard guard(fine_ghosts_valid, which == 1 ? mass_sym(F, F.coeff) : a_sym(F, F.coeff));
    CHECK(restrict_to(which, l, mg_t[which][l], e));
  }
  CHECK(level_v_step(which, l - 1));
  CHECK(prolongate_from(which, l, mg_sol[which][l], mg_sol[which][l - 1]));
  return smoother_smooth(which, l, mg_def[which][l]);
}

// edge_in->Tvmult (vmult_interface_up) after the coarse-grid correction: defect -= (rows off the edge of the operator with
// the boundary rows alone constrained) applied to the edge entries of the solution
int agmg_ctx::interface_up_into_defect(int which, int l) {
  const DeviceLevel &B = ad.Lb[l];
  const size_t n = size_of(which, l);
  const int nn = which == 0 ? B.n_nodes_v : B.n_nodes_p, nc = which == 0 ? B.dim : 1;
  double *xe = mg_tmp[which][l], *y = mg_t[which][l];
  CHECK(vec_copy(stream, xe, mg_sol[which][l], n));
  CHECK(vec_zero_constrained(stream, xe, ad.notedge[which][l], nn, nc));
  EpiArgs e;
  e.epi = EPI_STORE; e.dst = y; e.x = xe;
  CHECK(apply_on(B, which, B.coeff, xe, e));
  CHECK(vec_zero_constrained(stream, y, ad.edge[which][l], nn, nc));
  CHECK(vec_zero_constrained(stream, y, which == 0 ? B.cmask_v : B.cmask_p, nn, nc));   // (apply_add writes no boundary rows)
  return vec_axpy(stream, mg_def[which][l], -1.0, y, n);
}

// PreconditionMG::vmult of an adaptive context: copy_to_mg fills the defect of EVERY level through its copy table (the
// cells that are active on the level), the cycle runs from the finest level, copy_from_mg gathers the solutions.
int agmg_ctx::vcycle_adaptive(int which, double *dst, const double *src) {
  const int top = nl() - 1, nc = which == 0 ? opt.dim : 1;
  const size_t n = which == 0 ? (size_t)ad.act.n_dofs_v() : (size_t)ad.act.n_dofs_p();
  for (int l = 0; l <= top; ++l) {
    CHECK(vec_set(stream, mg_def[which][l], 0.0, size_of(which, l)));
    CHECK(vec_copy_nodes(stream, mg_def[which][l], ad.copy_lev_d[which][l], src, ad.copy_act_d[which][l],
                         (int)ad.copy_act[which][l].size(), nc));
  }
  CHECK(level_v_step(which, top));
  CHECK(vec_set(stream, dst, 0.0, n));
  for (int l = 0; l <= top; ++l)
    CHECK(vec_copy_nodes(stream, dst, ad.copy_act_d[which][l], mg_sol[which][l], ad.copy_lev_d[which][l],
                         (int)ad.copy_act[which][l].size(), nc));
  return 0;
}

// PreconditionMG::vmult on device vectors of the active (= finest) mesh
int agmg_ctx::vcycle(int which, double *dst, const double *src) {
  if (ad.on) return vcycle_adaptive(which, dst, src);
  const int top = nl() - 1;
  VGraph &g = vgraph[which];
  if (use_graphs && !dist() && g.state > 0) {
    // replay (or capture, then replay) on the cycle's own defect buffer: the source moves from call to call
    CHECK(vec_copy(stream, mg_def[which][top], src, size_of(which, top)));
    if (g.state == 1) {
      int rc = 0;
      if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) rc = -1;
      else {
        rc = level_v_step(which, top);
        hipGraph_t graph = nullptr;
        if (hipStreamEndCapture(stream, &graph) != hipSuccess || !graph) rc = rc ? rc : -1;
        if (!rc && hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) != hipSuccess) rc = -1;
        if (rc) { if (graph) (void)hipGraphDestroy(graph); g.exec = nullptr; }
        else g.graph = graph;
      }
      if (rc) { (void)hipGetLastError(); g.state = -1; }
      else g.state = 2;
    }
    if (g.state == 2) {
      AGMG_HIP_CHECK(hipGraphLaunch(g.exec, stream));
      return vec_copy(stream, dst, mg_sol[which][top], size_of(which, top));
    }
  }
  double *saved = mg_def[which][top];
  mg_def[which][top] = const_cast<double *>(src);  // read-only at the top level
  const int rc = level_v_step(which, top);
  mg_def[which][top] = saved;
  CHECK(rc);
  if (use_graphs && !dist() && g.state == 0) g.state = 1;
  return vec_copy(stream, dst, mg_sol[which][top], size_of(which, top));
}

// eigenvalue estimate by eig_cg_n_iterations steps of Jacobi-preconditioned CG on the device
int agmg_ctx::estimate_device(int which, int l) {
  const Span sp = span_of(which, l);
  const size_t n = sp.n;
  const bool coarse_lvl = l == 0;
  Smoother &s = sm[which][l];
  s.degree = coarse_lvl ? opt.coarse_degree : opt.smoother_degree;
  const double range = coarse_lvl ? opt.coarse_range : opt.smoothing_range;
  const int max_it = coarse_lvl ? opt.coarse_eig_cg_n_iterations : opt.eig_cg_n_iterations;
  const double *dinv = which == 0 ? L[l].inv_diag_A : L[l].inv_diag_M;
  double *g = mg_def[which][l], *h = mg_sol[which][l], *d = mg_tmp[which][l], *Ad = mg_t[which][l];
  // start vector: (index mod 11) minus its mean
  // (by GLOBAL index in a distributed context, so the estimate does not depend on the partition)
  const uint32_t *gid = which == 0 ? L[l].gid_v : L[l].gid_p;
  if (gid) CHECK(vec_fill_gid_mod(stream, g, gid, which == 0 ? L[l].n_nodes_v : L[l].n_nodes_p, which == 0 ? L[l].dim : 1, 11));
  else CHECK(vec_fill_index_mod(stream, g, n, 11));
  double n_glob = (double)sp.n_dot;
  if (!sp.local) CHECK(allreduce_host(&n_glob, 1));
  double sum = 0;
  {
    const size_t ng = (size_t)n_glob, full = ng / 11, rem = ng % 11;
    sum = (double)full * 55.0 + (double)(rem * (rem - 1) / 2);
  }
  CHECK(vec_shift(stream, g, -sum / n_glob, n));
  Lanczos lz;
  double res;
  CHECK(norm(g, sp, &res));
  const double tol = 1e-10;
  if (res > tol && max_it > 0) {
    double gh = 0, alpha = 0, beta = 0;
    for (int it = 1;; ++it) {
      CHECK(vec_pointwise_scale(stream, h, 1.0, dinv, g, n));
      const double gh_old = gh;
      CHECK(dots(g, {h}, sp, &gh));
      if (it > 1) {
        beta = gh / gh_old;
        CHECK(vec_sadd(stream, d, beta, 1.0, h, n));
      } else CHECK(vec_copy(stream, d, h, n));
      EpiArgs e;
      e.epi = EPI_STORE; e.dst = Ad; e.x = d;
      CHECK(apply_level(which, l, L[l].coeff, d, e));
      double dAd;
      CHECK(dots(d, {Ad}, sp, &dAd));
      alpha = gh / dAd;
      CHECK(vec_axpy(stream, g, -alpha, Ad, n));
      CHECK(norm(g, sp, &res));
      lz.push(it, alpha, beta);
      if (res <= tol || it >= max_it || !std::isfinite(res)) break;
    }
  }
  finish_smoother(s, lz, range);
  return 0;
}

// The same estimates for several (operator, level) pairs at once, one rank: the CG iterations are
// independent of each other, so every phase is enqueued for all of them before the host waits
// once (2 waits per iteration in total instead of 3 per iteration and problem; the reductions
// land directly in the pinned host buffer).  Same kernels, same order per problem as
// estimate_device.
int agmg_ctx::estimate_batched(const std::vector<std::pair<int, int>> &problems) {
  struct St {
    int which, l, max_it;
    Span sp;
    size_t n;
    double range;
    const double *dinv;
    double *g, *h, *d, *Ad;
    Lanczos lz;
    double gh = 0, alpha = 0, beta = 0, res = 0;
    bool active = true;
  };
  std::vector<St> P;
  for (auto &pr : problems) {
    St s;
    s.which = pr.first; s.l = pr.second;
    s.sp = span_of(s.which, s.l);
    s.n = s.sp.n;
    const bool coarse_lvl = s.l == 0;
    Smoother &sm_ = sm[s.which][s.l];
    sm_.degree = coarse_lvl ? opt.coarse_degree : opt.smoother_degree;
    s.range = coarse_lvl ? opt.coarse_range : opt.smoothing_range;
    s.max_it = coarse_lvl ? opt.coarse_eig_cg_n_iterations : opt.eig_cg_n_iterations;
    s.dinv = s.which == 0 ? L[s.l].inv_diag_A : L[s.l].inv_diag_M;
    s.g = mg_def[s.which][s.l]; s.h = mg_sol[s.which][s.l]; s.d = mg_tmp[s.which][s.l]; s.Ad = mg_t[s.which][s.l];
    P.push_back(s);
  }
  if (2 * P.size() > 256) { set_error("estimate_batched: too many problems"); return -1; }
  const double tol = 1e-10;
  // start vectors, h = D^-1 g, (g.h, g.g)
  auto phase_a = [&](St &s, int slot) -> int {
    CHECK(vec_pointwise_scale(stream, s.h, 1.0, s.dinv, s.g, s.n));
    VecPtrs vp;
    for (int k = 0; k < DOT_MAX_VECS; ++k) vp.p[k] = k == 0 ? s.h : s.g;
    return vec_multi_dot(stream, s.g, vp, 2, s.sp.n_dot, s.sp.seg1, s.sp.gap, dot_partials, dot_out_h + 2 * slot);   // (one stream: the partials buffer is free again)
  };
  for (size_t i = 0; i < P.size(); ++i) {
    St &s = P[i];
    CHECK(vec_fill_index_mod(stream, s.g, s.n, 11));
    const size_t full = s.n / 11, rem = s.n % 11;
    const double sum = (double)full * 55.0 + (double)(rem * (rem - 1) / 2);
    CHECK(vec_shift(stream, s.g, -sum / (double)s.n, s.n));
    CHECK(phase_a(s, (int)i));
  }
  CHECK(sync());
  for (size_t i = 0; i < P.size(); ++i) {
    P[i].gh = dot_out_h[2 * i];
    P[i].res = std::sqrt(dot_out_h[2 * i + 1]);
    P[i].active = P[i].res > tol && P[i].max_it > 0;
  }
  for (int it = 1;; ++it) {
    bool any = false;
    for (size_t i = 0; i < P.size(); ++i) {
      St &s = P[i];
      if (!s.active) continue;
      any = true;
      if (it > 1) CHECK(vec_sadd(stream, s.d, s.beta, 1.0, s.h, s.n));
      else CHECK(vec_copy(stream, s.d, s.h, s.n));
      EpiArgs e;
      e.epi = EPI_STORE; e.dst = s.Ad; e.x = s.d;
      CHECK(apply_level(s.which, s.l, L[s.l].coeff, s.d, e));
      VecPtrs vp;
      for (int k = 0; k < DOT_MAX_VECS; ++k) vp.p[k] = s.Ad;
      CHECK(vec_multi_dot(stream, s.d, vp, 1, s.sp.n_dot, s.sp.seg1, s.sp.gap, dot_partials, dot_out_h + 2 * i));
    }
    if (!any) break;
    CHECK(sync());
    for (size_t i = 0; i < P.size(); ++i) {
      St &s = P[i];
      if (!s.active) continue;
      s.alpha = s.gh / dot_out_h[2 * i];
      CHECK(vec_axpy(stream, s.g, -s.alpha, s.Ad, s.n));
      CHECK(phase_a(s, (int)i));
    }
    CHECK(sync());
    for (size_t i = 0; i < P.size(); ++i) {
      St &s = P[i];
      if (!s.active) continue;
      const double gh_new = dot_out_h[2 * i];
      s.res = std::sqrt(dot_out_h[2 * i + 1]);
      s.lz.push(it, s.alpha, s.beta);
      if (s.res <= tol || it >= s.max_it || !std::isfinite(s.res)) { s.active = false; continue; }
      s.beta = gh_new / s.gh;
      s.gh = gh_new;
    }
  }
  for (St &s : P) finish_smoother(sm[s.which][s.l], s.lz, s.range);
  return 0;
}

// dense level-0 operator (by columns) + host eigenvalue estimate
int agmg_ctx::build_coarse(int which) {
  CoarseOp &co = coarse[which];
  const int n = (int)size_of(which, 0);
  co.n = n;
  co.A.assign((size_t)n * n, 0.0);
  co.dinv.resize(n);
  if (n <= 4096) {
    // all columns back to back on the stream (identity in, operator columns out), one copy back
    std::vector<double> eye((size_t)n * n, 0.0), cols((size_t)n * n);
    for (int j = 0; j < n; ++j) eye[(size_t)j * n + j] = 1.0;
    double *X = nullptr, *Y = nullptr;
    CHECK(dalloc(&X, eye.size()));
    int rc = dalloc(&Y, eye.size());
    if (!rc) rc = h2d(X, eye.data(), eye.size());
    for (int j = 0; j < n && !rc; ++j) {
      EpiArgs e;
      e.epi = EPI_STORE; e.dst = Y + (size_t)j * n; e.x = X + (size_t)j * n;
      rc = apply_level(which, 0, L[0].coeff, X + (size_t)j * n, e);
    }
    if (!rc) rc = d2h(cols.data(), Y, cols.size());
    dfree(X); dfree(Y);
    if (rc) return rc;
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) co.A[(size_t)i * n + j] = cols[(size_t)j * n + i];
  } else {
    double *x = mg_sol[which][0], *y = mg_t[which][0];
    CHECK(vec_set(stream, x, 0.0, n));
    std::vector<double> col(n);
    const double one = 1.0, zero = 0.0;
    for (int j = 0; j < n; ++j) {
      AGMG_HIP_CHECK(hipMemcpyAsync(x + j, &one, sizeof(double), hipMemcpyHostToDevice, stream));
      EpiArgs e;
      e.epi = EPI_STORE; e.dst = y; e.x = x;
      CHECK(apply_level(which, 0, L[0].coeff, x, e));
      CHECK(d2h(col.data(), y, n));
      AGMG_HIP_CHECK(hipMemcpyAsync(x + j, &zero, sizeof(double), hipMemcpyHostToDevice, stream));
      for (int i = 0; i < n; ++i) co.A[(size_t)i * n + j] = col[i];
    }
  }
  CHECK(d2h(co.dinv.data(), which == 0 ? L[0].inv_diag_A : L[0].inv_diag_M, n));
  // host CG for the eigenvalue estimate
  Smoother &s = sm[which][0];
  s.degree = opt.coarse_degree;
  std::vector<double> g(n), h(n), d(n), Ad(n);
  double mean = 0;
  for (int i = 0; i < n; ++i) { g[i] = (double)(i % 11); mean += g[i]; }
  mean /= n;
  for (auto &v : g) v -= mean;
  auto dot = [&](const std::vector<double> &a, const std::vector<double> &b) {
    double acc = 0;
    for (int i = 0; i < n; ++i) acc += a[i] * b[i];
    return acc;
  };
  Lanczos lz;
  double res = std::sqrt(dot(g, g));
  const double tol = 1e-10;
  const int max_it = opt.coarse_eig_cg_n_iterations;
  if (res > tol && max_it > 0) {
    double gh = 0, alpha = 0, beta = 0;
    for (int it = 1;; ++it) {
      for (int i = 0; i < n; ++i) h[i] = co.dinv[i] * g[i];
      const double gh_old = gh;
      gh = dot(g, h);
      if (it > 1) {
        beta = gh / gh_old;
        for (int i = 0; i < n; ++i) d[i] = h[i] + beta * d[i];
      } else d = h;
      for (int i = 0; i < n; ++i) {
        double acc = 0;
        for (int j = 0; j < n; ++j) acc += co.A[(size_t)i * n + j] * d[j];
        Ad[i] = acc;
      }
      alpha = gh / dot(d, Ad);
      for (int i = 0; i < n; ++i) g[i] -= alpha * Ad[i];
      res = std::sqrt(dot(g, g));
      lz.push(it, alpha, beta);
      if (res <= tol || it >= max_it || !std::isfinite(res)) break;
    }
  }
  finish_smoother(s, lz, opt.coarse_range);
  return 0;
}

int agmg_ctx::setup_smoothers() {
  if (!diag_ready) { set_error("setup_smoothers: call agmg_compute_diagonals first"); return -1; }
  CHECK(ensure_workspace());
  CHECK(ensure_transfers());
  CHECK(ensure_mg_vectors());
  CHECK(sync());
  free_graphs();           // smoother coefficients are kernel arguments of the captured cycles
  std::vector<std::pair<int, int>> batch;
  for (int w = 0; w < 2; ++w)
    for (int l = 0; l < nl(); ++l) {
      if (l == 0 && coarse_host(w)) {
        CHECK(build_coarse(w));
        const size_t need = 2 * (size_t)coarse[w].n;
        if (coarse_pin_cap[w] < need) {
          if (coarse_pin[w]) (void)hipHostFree(coarse_pin[w]);
          coarse_pin[w] = nullptr; coarse_pin_cap[w] = 0;
          AGMG_HIP_CHECK(hipHostMalloc((void **)&coarse_pin[w], need * sizeof(double)));
          coarse_pin_cap[w] = need;
        }
      }
      else if (!ldist(L[l]) && use_batched_setup) batch.emplace_back(w, l);
      else CHECK(estimate_device(w, l));
    }
  if (!batch.empty()) CHECK(estimate_batched(batch));
  smoothers_ready = true;
  return 0;
}

// ===================================================================== block preconditioner
// deal.II SolverCG on device vectors; work = {g, d, h, Ad}
int agmg_ctx::solver_cg(const std::function<int(double *, const double *)> &A,
                        const std::function<int(double *, const double *)> &P, double *x, const double *b, const Span &sp,
                        int max_it, double tol, double *work[4], int *its, bool *converged) {
  const size_t n = sp.n;
  double *g = work[0], *d = work[1], *h = work[2], *Ad = work[3];
  CHECK(A(g, x));
  CHECK(vec_lincomb(stream, g, 1.0, b, -1.0, g, n));
  double res;
  CHECK(norm(g, sp, &res));
  *its = 0;
  *converged = false;
  if (res <= tol) { *converged = true; return 0; }
  double gh = 0, alpha = 0, beta = 0;
  for (int it = 1;; ++it) {
    CHECK(P(h, g));
    const double gh_old = gh;
    CHECK(dots(g, {h}, sp, &gh));
    if (it > 1) {
      beta = gh / gh_old;
      CHECK(vec_sadd(stream, d, beta, 1.0, h, n));
    } else CHECK(vec_copy(stream, d, h, n));
    CHECK(A(Ad, d));
    double dAd;
    CHECK(dots(d, {Ad}, sp, &dAd));
    alpha = gh / dAd;
    CHECK(vec_axpy(stream, x, alpha, d, n));
    CHECK(vec_axpy(stream, g, -alpha, Ad, n));
    CHECK(norm(g, sp, &res));
    *its = it;
    if (res <= tol) { *converged = true; break; }
    if (it >= max_it || !std::isfinite(res)) break;
  }
  return 0;
}

// deal.II SolverBicgstab, AdditionalData(exact_residual = false), as BlockSchurGMGPreconditioner uses
// it for a non-symmetric A block (stokes_matrix_free.h:321-333); work = {r, rbar, p, y, z, t, v}
int agmg_ctx::solver_bicgstab(const std::function<int(double *, const double *)> &A,
                              const std::function<int(double *, const double *)> &P, double *x, const double *b,
                              const Span &sp, int max_it, double tol, double *work[7], int *its, bool *converged) {
  const size_t n = sp.n;
  double *r = work[0], *rbar = work[1], *p = work[2], *y = work[3], *z = work[4], *t = work[5], *v = work[6];
  const double breakdown = std::numeric_limits<double>::min();
  int step = 0;
  *its = 0;
  *converged = false;
  auto status = [&](double res) {
    *its = step;
    if (res <= tol) return ST_SUCCESS;
    if (step >= max_it || !std::isfinite(res)) return ST_FAILURE;
    return ST_ITERATE;
  };
  for (int restart = 0; restart < 1000; ++restart) {
    CHECK(A(r, x));
    CHECK(vec_lincomb(stream, r, 1.0, b, -1.0, r, n));
    double res;
    CHECK(norm(r, sp, &res));
    int state = status(res);
    if (state == ST_SUCCESS) { *converged = true; return 0; }
    if (state == ST_FAILURE) return 0;
    double alpha = 1, omega = 1, rho = 1;
    CHECK(vec_copy(stream, rbar, r, n));
    bool startup = true, broke = false;
    do {
      ++step;
      double rhobar;
      CHECK(dots(r, {rbar}, sp, &rhobar));
      if (std::fabs(rhobar) < breakdown) { broke = true; break; }
      const double beta = rhobar * alpha / (rho * omega);
      rho = rhobar;
      if (startup) { CHECK(vec_copy(stream, p, r, n)); startup = false; }
      else {
        CHECK(vec_sadd(stream, p, beta, 1.0, r, n));
        CHECK(vec_axpy(stream, p, -beta * omega, v, n));
      }
      CHECK(P(y, p));
      CHECK(A(v, y));
      CHECK(dots(rbar, {v}, sp, &rhobar));
      if (std::fabs(rhobar) < breakdown) { broke = true; break; }
      alpha = rho / rhobar;
      CHECK(vec_axpy(stream, r, -alpha, v, n));
      CHECK(norm(r, sp, &res));
      if (status(res) == ST_SUCCESS) {
        CHECK(vec_axpy(stream, x, alpha, y, n));
        *converged = true;
        return 0;
      }
      CHECK(P(z, r));
      CHECK(A(t, z));
      double d2[2];
      CHECK(dots(t, {r, t}, sp, d2));
      if (d2[1] < breakdown) { broke = true; break; }
      omega = d2[0] / d2[1];
      CHECK(vec_axpy(stream, x, alpha, y, n));
      CHECK(vec_axpy(stream, x, omega, z, n));
      CHECK(vec_axpy(stream, r, -omega, t, n));
      CHECK(norm(r, sp, &res));
      state = status(res);
    } while (state == ST_ITERATE);
    if (!broke) { *converged = state == ST_SUCCESS; return 0; }
  }
  return 0;
}

// BlockSchurGMGPreconditioner::vmult (stokes_matrix_free.h:237-354) on device block vectors
int agmg_ctx::precondition(bool expensive, double *dst, const double *src, int *status) {
  const DeviceLevel &lv = act();
  const size_t nu = lv.n_dofs_v(), np = lv.n_dofs_p();
  const int top = nl() - 1;
  double *dst_u = dst, *dst_p = dst + nu;
  const double *src_u = src, *src_p = src + nu;
  double *utmp;
  CHECK(get_vec(0, &utmp));
  if (expensive) {
    double nrm;
    CHECK(norm(src_p, span_act(1), &nrm));
    CHECK(vec_set(stream, dst_p, 0.0, np));
    if (nrm > 1e-50) {
      double *w0, *w1;
      CHECK(get_vec(1, &w0)); CHECK(get_vec(2, &w1));
      double *work[4] = {w0, w0 + np, w1, w1 + np};  // np <= nu: two pressure vectors fit one block vector
      auto Aop = [&](double *d, const double *s) {
        if (ad.on) return active_apply(AGMG_OP_MASS, d, s, nullptr);
        EpiArgs e; e.epi = EPI_STORE; e.dst = d; e.x = s;
        return apply_level(1, top, active_coeff(), s, e);
      };
      auto Pop = [&](double *d, const double *s) { return vcycle(1, d, s); };
      int its; bool conv;
      CHECK(solver_cg(Aop, Pop, dst_p, src_p, span_act(1), 100, nrm * opt.S_block_tolerance, work, &its, &conv));
      nS += its;
      if (!conv && status) *status = -2;
    }
  } else {
    CHECK(vcycle(1, dst_p, src_p));
    nS += 1;
  }
  CHECK(vec_scale(stream, dst_p, -1.0, np));
  CHECK(bt_residual(utmp, src_u, dst_p));  // utmp = src_u - B^T dst_p
  if (expensive) {
    double nrm;
    CHECK(norm(utmp, span_act(0), &nrm));
    CHECK(vec_set(stream, dst_u, 0.0, nu));
    double *w[7];
    for (int i = 0; i < 4; ++i) CHECK(get_vec(1 + i, &w[i]));
    if (!opt.A_block_is_symmetric) {
      while (bicg_pool.size() < 3) {
        double *v;
        CHECK(dalloc(&v, nu));
        bicg_pool.push_back(v);
      }
      for (int i = 0; i < 3; ++i) w[4 + i] = bicg_pool[i];
    }
    auto Aop = [&](double *d, const double *s) {
      if (ad.on) return active_apply(AGMG_OP_A, d, s, nullptr);
      EpiArgs e; e.epi = EPI_STORE; e.dst = d; e.x = s;
      return apply_level(0, top, active_coeff(), s, e);
    };
    auto Pop = [&](double *d, const double *s) { return vcycle(0, d, s); };
    int its; bool conv;
    if (opt.A_block_is_symmetric)
      CHECK(solver_cg(Aop, Pop, dst_u, utmp, span_act(0), 1000, nrm * opt.A_block_tolerance, w, &its, &conv));
    else
      CHECK(solver_bicgstab(Aop, Pop, dst_u, utmp, span_act(0), 1000, nrm * opt.A_block_tolerance, w, &its, &conv));
    nA += its;
    if (!conv && status) *status = -3;
  } else {
    CHECK(vcycle(0, dst_u, utmp));
    nA += 1;
  }
  return 0;
}

// ===================================================================== Krylov solvers
namespace {

using Op = std::function<int(double *, const double *)>;

struct Hessenberg {  // Givens-rotated least-squares state shared by GMRES and FGMRES
  int m;
  std::vector<double> gamma, ci, si;
  std::vector<std::vector<double>> H;
  explicit Hessenberg(int m_) : m(m_), gamma(m_ + 1, 0.0), ci(m_, 0.0), si(m_, 0.0), H(m_ + 1, std::vector<double>(m_, 0.0)) {}
  double push(int j, std::vector<double> &h) {  // h has j+2 entries
    for (int i = 0; i < j; ++i) {
      const double t = h[i];
      h[i] = ci[i] * t + si[i] * h[i + 1];
      h[i + 1] = -si[i] * t + ci[i] * h[i + 1];
    }
    const double r = std::hypot(h[j], h[j + 1]);
    // r == 0: the Krylov space is exhausted (happy breakdown); identity rotation, zero pivot guarded in solve()
    ci[j] = r > 0.0 ? h[j] / r : 1.0;
    si[j] = r > 0.0 ? h[j + 1] / r : 0.0;
    h[j] = r;
    gamma[j + 1] = -si[j] * gamma[j];
    gamma[j] *= ci[j];
    for (int i = 0; i <= j; ++i) H[i][j] = h[i];
    return std::fabs(gamma[j + 1]);
  }
  std::vector<double> solve(int dim) const {
    std::vector<double> y(dim);
    for (int i = dim - 1; i >= 0; --i) {
      double s = gamma[i];
      for (int k = i + 1; k < dim; ++k) s -= H[i][k] * y[k];
      y[i] = H[i][i] != 0.0 ? s / H[i][i] : 0.0;
    }
    return y;
  }
};

// y += sum_i a[i] * xs[i]
int multi_axpy(agmg_ctx *c, double *y, const std::vector<const double *> &xs, const std::vector<double> &a, size_t n) {
  const int total = (int)xs.size();
  for (int k0 = 0; k0 < total; k0 += DOT_MAX_VECS) {
    const int nv = std::min(DOT_MAX_VECS, total - k0);
    VecPtrs vp;
    Coeffs co;
    for (int k = 0; k < DOT_MAX_VECS; ++k) { vp.p[k] = xs[k0 + std::min(k, nv - 1)]; co.a[k] = k < nv ? a[k0 + k] : 0.0; }
    CHECK(vec_multi_axpy(c->stream, y, vp, co, nv, n));
  }
  return 0;
}

// orthogonalise w against V[0..j] (classical Gram-Schmidt, fused dots, one re-orthogonalisation);
// h[0..j] receives the coefficients, h[j+1] the norm of the remainder.  Two reductions (= two host
// waits, two all-reduces in distributed contexts) per call: the second pass also carries w . w, and
// with V orthonormal |w - V c|^2 = w . w - |c|^2, where the c of the second pass are rounding-sized
int orthogonalize(agmg_ctx *c, double *w, const std::vector<double *> &V, int j, const agmg_ctx::Span &sp, std::vector<double> &h) {
  const size_t n = sp.n;
  std::vector<const double *> vs(V.begin(), V.begin() + j + 1);
  std::vector<double> hh(j + 2), neg(j + 1);
  h.assign(j + 2, 0.0);
  CHECK(c->dots(w, vs, sp, hh.data()));
  for (int i = 0; i <= j; ++i) { h[i] += hh[i]; neg[i] = -hh[i]; }
  CHECK(multi_axpy(c, w, vs, neg, n));
  std::vector<const double *> vw(vs);
  vw.push_back(w);
  CHECK(c->dots(w, vw, sp, hh.data()));
  double ww = hh[j + 1], corr = 0.0;
  for (int i = 0; i <= j; ++i) { h[i] += hh[i]; neg[i] = -hh[i]; corr += hh[i] * hh[i]; }
  CHECK(multi_axpy(c, w, vs, neg, n));
  if (!(corr <= 1e-8 * ww)) {   // (the basis has lost orthogonality, or w lies in its span: measure the remainder)
    double s;
    CHECK(c->norm(w, sp, &s));
    h[j + 1] = s;
  } else {
    h[j + 1] = std::sqrt(ww - corr);
  }
  return 0;
}

// deal.II SolverGMRES, right preconditioning: x += P^-1 (V y) at convergence / restart
int solver_gmres(agmg_ctx *c, const Op &A, const Op &P, double *x, const double *b, const agmg_ctx::Span &sp, int restart,
                 Control &ctl, size_t pool0, int *state_out) {
  const size_t n = sp.n;
  std::vector<double *> V;
  double *p, *w;
  CHECK(c->get_vec(pool0, &p));
  CHECK(c->get_vec(pool0 + 1, &w));
  auto basis = [&](int i, double **v) { return c->get_vec(pool0 + 2 + i, v); };
  int acc = 0, state = ST_ITERATE;
  do {
    V.clear();
    double *v0;
    CHECK(basis(0, &v0));
    CHECK(c->stokes_residual(v0, b, x));
    double rho;
    CHECK(c->norm(v0, sp, &rho));
    state = ctl.check(acc, rho);
    if (state != ST_ITERATE) break;
    CHECK(vec_scale(c->stream, v0, 1.0 / rho, n));
    V.push_back(v0);
    Hessenberg hs(restart);
    hs.gamma[0] = rho;
    int dim = 0;
    for (int j = 0; j < restart && state == ST_ITERATE; ++j) {
      ++acc;
      CHECK(P(p, V[j]));
      CHECK(A(w, p));
      std::vector<double> h;
      CHECK(orthogonalize(c, w, V, j, sp, h));
      double *vn;
      CHECK(basis(j + 1, &vn));
      const double s = h[j + 1];
      CHECK(vec_lincomb(c->stream, vn, s != 0 ? 1.0 / s : 0.0, w, 0.0, w, n));
      V.push_back(vn);
      rho = hs.push(j, h);
      dim = j + 1;
      state = ctl.check(acc, rho);
    }
    const std::vector<double> y = hs.solve(dim);
    CHECK(vec_set(c->stream, w, 0.0, n));
    CHECK(multi_axpy(c, w, std::vector<const double *>(V.begin(), V.begin() + dim), y, n));
    CHECK(P(p, w));
    CHECK(vec_axpy(c->stream, x, 1.0, p, n));
  } while (state == ST_ITERATE);
  *state_out = state;
  return 0;
}

// deal.II SolverFGMRES(max_basis_size)
int solver_fgmres(agmg_ctx *c, const Op &A, const Op &P, double *x, const double *b, const agmg_ctx::Span &sp, int basis_size,
                  Control &ctl, size_t pool0, int *state_out) {
  const size_t n = sp.n;
  double *aux;
  CHECK(c->get_vec(pool0, &aux));
  auto vvec = [&](int i, double **v) { return c->get_vec(pool0 + 1 + 2 * i, v); };
  auto zvec = [&](int i, double **v) { return c->get_vec(pool0 + 2 + 2 * i, v); };
  int acc = 0, state = ST_ITERATE;
  do {
    CHECK(c->stokes_residual(aux, b, x));
    double beta;
    CHECK(c->norm(aux, sp, &beta));
    state = ctl.check(acc, beta);
    if (state != ST_ITERATE) break;
    std::vector<double *> V, Z;
    Hessenberg hs(basis_size);
    hs.gamma[0] = beta;
    double *v;
    CHECK(vvec(0, &v));
    CHECK(vec_lincomb(c->stream, v, 1.0 / beta, aux, 0.0, aux, n));
    int dim = 0;
    for (int j = 0; j < basis_size && state == ST_ITERATE; ++j) {
      V.push_back(v);
      double *z;
      CHECK(zvec(j, &z));
      CHECK(P(z, V[j]));
      Z.push_back(z);
      CHECK(A(aux, z));
      std::vector<double> h;
      CHECK(orthogonalize(c, aux, V, j, sp, h));
      CHECK(vvec(j + 1, &v));
      const double a = h[j + 1];
      CHECK(vec_lincomb(c->stream, v, a != 0 ? 1.0 / a : 0.0, aux, 0.0, aux, n));
      const double res = hs.push(j, h);
      dim = j + 1;
      state = ctl.check(++acc, res);
    }
    const std::vector<double> y = hs.solve(dim);
    CHECK(multi_axpy(c, x, std::vector<const double *>(Z.begin(), Z.begin() + dim), y, n));
  } while (state == ST_ITERATE);
  *state_out = state;
  return 0;
}

// deal.II SolverIDR(s), right preconditioned (van Gijzen & Sonneveld, bi-orthogonal variant)
int solver_idr(agmg_ctx *c, const Op &A, const Op &P, double *x, const double *b, const agmg_ctx::Span &sp, int s, Control &ctl,
               size_t pool0, int *state_out) {
  const size_t n = sp.n;
  double *r, *v, *vhat, *uhat, *ghat;
  size_t k0 = pool0;
  CHECK(c->get_vec(k0++, &r)); CHECK(c->get_vec(k0++, &v)); CHECK(c->get_vec(k0++, &vhat));
  CHECK(c->get_vec(k0++, &uhat)); CHECK(c->get_vec(k0++, &ghat));
  std::vector<double *> G(s), U(s), Q(s);
  for (int i = 0; i < s; ++i) { CHECK(c->get_vec(k0++, &G[i])); CHECK(c->get_vec(k0++, &U[i])); CHECK(c->get_vec(k0++, &Q[i])); }
  hipStream_t st = c->stream;
  CHECK(c->stokes_residual(r, b, x));
  double res;
  CHECK(c->norm(r, sp, &res));
  int step = 0;
  int state = ctl.check(step, res);
  *state_out = state;
  if (state != ST_ITERATE) return 0;
  std::vector<std::vector<double>> M(s, std::vector<double>(s, 0.0));
  std::mt19937 rng(5489u);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> hq(n);
  // Distributed: the shadow vectors are defined by GLOBAL dof index (every rank draws the whole
  // serial sequence and keeps its own entries), so that iteration counts do not depend on the
  // partition.  (deal.II draws per rank; its parallel IDR runs are partition dependent.)
  const DeviceLevel &lf = c->fin();
  const bool by_gid = c->dist() && c->gid_fin_h[1].size() == (size_t)lf.n_nodes_v && c->gid_fin_h[0].size() == (size_t)lf.n_nodes_p;
  double ng[2] = {(double)lf.n_owned_v, (double)lf.n_owned_p};   // global node counts
  if (by_gid) CHECK(c->allreduce_host(ng, 2));
  std::vector<double> hglob(by_gid ? (size_t)ng[0] * lf.dim + (size_t)ng[1] : 0);
  for (int i = 0; i < s; ++i) {
    CHECK(vec_set(st, G[i], 0.0, n));
    CHECK(vec_set(st, U[i], 0.0, n));
    if (i != 0) {
      if (by_gid) {
        for (auto &e : hglob) e = nd(rng);
        const size_t nu_l = (size_t)lf.n_nodes_v * lf.dim, nu_g = (size_t)ng[0] * lf.dim;
        for (int node = 0; node < lf.n_nodes_v; ++node)
          for (int cc = 0; cc < lf.dim; ++cc) hq[(size_t)node * lf.dim + cc] = hglob[(size_t)c->gid_fin_h[1][node] * lf.dim + cc];
        for (int node = 0; node < lf.n_nodes_p; ++node) hq[nu_l + node] = hglob[nu_g + c->gid_fin_h[0][node]];
      } else {
        for (auto &e : hq) e = nd(rng);
      }
      CHECK(c->h2d(Q[i], hq.data(), n));
    } else CHECK(vec_copy(st, Q[i], r, n));
    for (int j = 0; j < i; ++j) {
      double d2[2];
      CHECK(c->dots(Q[j], {Q[i], Q[j]}, sp, d2));
      CHECK(vec_axpy(st, Q[i], -d2[0] / d2[1], Q[j], n));
    }
    if (i != 0) {
      double nn;
      CHECK(c->norm(Q[i], sp, &nn));
      CHECK(vec_scale(st, Q[i], 1.0 / nn, n));
    }
    M[i][i] = 1.0;
  }
  double omega = 1.0;
  bool early = false;
  while (state == ST_ITERATE) {
    ++step;
    std::vector<double> phi(s);
    CHECK(c->dots(r, std::vector<const double *>(Q.begin(), Q.end()), sp, phi.data()));
    for (int k = 0; k < s; ++k) {
      std::vector<double> gamma(s - k);
      for (int i = k; i < s; ++i) {
        double sum = phi[i];
        for (int j = k; j < i; ++j) sum -= M[i][j] * gamma[j - k];
        gamma[i - k] = sum / M[i][i];
      }
      CHECK(vec_copy(st, v, r, n));
      for (int i = k; i < s; ++i) CHECK(vec_axpy(st, v, -gamma[i - k], G[i], n));
      CHECK(P(vhat, v));
      CHECK(vec_lincomb(st, uhat, omega, vhat, 0.0, vhat, n));
      for (int i = k; i < s; ++i) CHECK(vec_axpy(st, uhat, gamma[i - k], U[i], n));
      CHECK(A(ghat, uhat));
      if (k > 0) {
        double dq;
        CHECK(c->dots(Q[0], {ghat}, sp, &dq));
        double alpha = dq / M[0][0];
        for (int i = 1; i < k; ++i) {
          CHECK(vec_axpy(st, ghat, -alpha, G[i - 1], n));
          CHECK(vec_axpy(st, uhat, -alpha, U[i - 1], n));
          CHECK(c->dots(Q[i], {ghat}, sp, &dq));
          alpha = dq / M[i][i];
        }
        CHECK(vec_axpy(st, ghat, -alpha, G[k - 1], n));
        CHECK(vec_axpy(st, uhat, -alpha, U[k - 1], n));
      }
      std::swap(G[k], ghat);
      std::swap(U[k], uhat);
      {
        std::vector<double> col(s - k);
        CHECK(c->dots(G[k], std::vector<const double *>(Q.begin() + k, Q.end()), sp, col.data()));
        for (int i = k; i < s; ++i) M[i][k] = col[i - k];
      }
      const double beta = phi[k] / M[k][k];
      CHECK(vec_axpy(st, r, -beta, G[k], n));
      CHECK(vec_axpy(st, x, beta, U[k], n));
      CHECK(c->norm(r, sp, &res));
      state = ctl.check(step, res);
      if (state != ST_ITERATE) { early = true; break; }
      if (k + 1 < s) {
        for (int i = 0; i < k + 1; ++i) phi[i] = 0.0;
        for (int i = k + 1; i < s; ++i) phi[i] -= beta * M[i][k];
      }
    }
    if (early) break;
    CHECK(P(vhat, r));
    CHECK(A(v, vhat));
    double d2[2];
    CHECK(c->dots(v, {r, v}, sp, d2));
    omega = d2[0] / d2[1];
    CHECK(vec_axpy(st, r, -omega, v, n));
    CHECK(vec_axpy(st, x, omega, vhat, n));
    CHECK(c->norm(r, sp, &res));
    state = ctl.check(step, res);
  }
  *state_out = state;
  return 0;
}

}  // namespace

// ===================================================================== C API
extern "C" {

const char *agmg_last_error(void) { return agmg::g_error.c_str(); }

void agmg_default_options(agmg_options *o) {
  std::memset(o, 0, sizeof(*o));
  o->dim = 3; o->degree_v = 2; o->device = 0; o->n_levels = 1;
  o->linear_tolerance = 1e-7; o->n_cheap_steps = 1000; o->n_expensive_steps = 1000;
  o->gmres_restart = 100; o->krylov_type = AGMG_KRYLOV_GMRES; o->idr_s = 2;
  o->A_block_tolerance = 1e-2; o->S_block_tolerance = 1e-6; o->A_block_is_symmetric = 1;
  o->coarse_on_host = 1;
  o->smoother_degree = 4; o->smoothing_range = 15.0; o->eig_cg_n_iterations = 10;
  o->coarse_degree = 8; o->coarse_range = 1e-3; o->coarse_eig_cg_n_iterations = 100;
  o->schur_smoother_steps = 4;
  o->world_size = 1; o->rank = 0;
}

agmg_ctx *agmg_create(const agmg_options *opt) {
  if (!opt) { set_error("null options"); return nullptr; }
  if (opt->dim != 2 && opt->dim != 3) { set_error("dim must be 2 or 3"); return nullptr; }
  if (opt->degree_v != 2) { set_error("only Stokes velocity polynomial degree 2 (Q2-Q1) is built"); return nullptr; }
  if (opt->n_levels < 1) { set_error("n_levels must be >= 1"); return nullptr; }
  if (opt->world_size < 1 || opt->rank < 0 || opt->rank >= opt->world_size) { set_error("bad world_size / rank"); return nullptr; }
  // (the fused multi-dot returns at most 256 scalars per call: restart + 2 basis vectors)
  if (opt->gmres_restart < 1 || opt->gmres_restart > 250) { set_error("gmres_restart must be in [1, 250]"); return nullptr; }
  if (opt->idr_s < 1 || opt->idr_s > 64) { set_error("idr_s must be in [1, 64]"); return nullptr; }
  if (opt->n_cheap_steps < 0 || opt->n_expensive_steps < 0) { set_error("step counts must be >= 0"); return nullptr; }
  if (opt->smoother_degree < 1 || opt->coarse_degree < 1 || opt->schur_smoother_steps < 1 ||
      opt->eig_cg_n_iterations < 1 || opt->coarse_eig_cg_n_iterations < 1) {
    set_error("smoother degrees, steps and eigenvalue iterations must be >= 1"); return nullptr;
  }

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device: this library has no CPU path"); return nullptr; }
  if (opt->device < 0 || opt->device >= ndev) { set_error("bad device ordinal"); return nullptr; }
  if (hipSetDevice(opt->device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
  if (static_scratch_init(opt->device)) return nullptr;   // device address of this GPU's static scratch
  auto *c = new agmg_ctx();
  c->opt = *opt;
  if (const char *e = getenv("AGMG_NO_BLOCKS")) c->use_blocks = atoi(e) == 0;
  if (const char *e = getenv("AGMG_BRICKS")) c->use_bricks = atoi(e) != 0 ? 1 : 0;
  if (const char *e = getenv("AGMG_BRICK_Z")) c->brick_z_override = atoi(e);
  if (const char *e = getenv("AGMG_BRICK_STACK")) c->brick_stack_override = atoi(e);
  if (const char *e = getenv("AGMG_NO_FUSED")) c->use_fused = atoi(e) == 0;
  if (const char *e = getenv("AGMG_NO_ELL")) c->use_ell = atoi(e) == 0;
  if (const char *e = getenv("AGMG_NO_SYM")) c->use_sym = atoi(e) == 0;
  if (const char *e = getenv("AGMG_NO_ELL_CLASSES")) c->use_ell_classes = atoi(e) == 0;
  if (const char *e = getenv("AGMG_GRAPHS")) c->use_graphs = atoi(e) != 0;
  if (const char *e = getenv("AGMG_NO_BATCHED_SETUP")) c->use_batched_setup = atoi(e) == 0;
  c->L.resize(opt->n_levels);
  c->H.resize(opt->n_levels);
  for (int w = 0; w < 2; ++w) c->send_nodes_h[w].resize(opt->n_levels);
  if (hipStreamCreate(&c->stream) != hipSuccess) { set_error("hipStreamCreate failed"); delete c; return nullptr; }
  if (const char *e = getenv("AGMG_NO_OVERLAP")) c->use_overlap = atoi(e) == 0;
  if (opt->world_size > 1 &&
      (hipStreamCreate(&c->cstream) != hipSuccess || hipEventCreateWithFlags(&c->ev_src_ready, hipEventDisableTiming) != hipSuccess ||
       hipEventCreateWithFlags(&c->ev_import_done, hipEventDisableTiming) != hipSuccess)) {
    set_error("creating the communication stream failed"); delete c; return nullptr;
  }
  return c;
}

int agmg_set_communicator(agmg_ctx *c, const agmg_comm *comm) {
  if (!c || !comm) { set_error("null argument"); return -1; }
  if (!comm->exchange || !comm->allreduce_sum) { set_error("communicator without callbacks"); return -1; }
  if (comm->size != c->opt.world_size || comm->rank != c->opt.rank) { set_error("communicator rank/size differ from the context's options"); return -1; }
  c->comm = *comm;
  c->comm_set = true;
  return 0;
}

int agmg_rccl_get_unique_id(uint8_t id[128]) {
  if (!id) { set_error("null argument"); return -1; }
  return rccl_unique_id(id);
}

int agmg_set_communicator_rccl(agmg_ctx *c, const uint8_t id[128]) {
  if (!c || !id) { set_error("null argument"); return -1; }
  if (c->rccl_state) { set_error("RCCL transport already bound"); return -1; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  CHECK(rccl_comm_create(id, c->opt.rank, c->opt.world_size, &c->comm, &c->rccl_state));
  c->comm_set = true;
  return 0;
}

void agmg_destroy(agmg_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->opt.device);
  (void)hipStreamSynchronize(c->stream);
  for (auto &l : c->L) c->free_level(l);
  for (int l = 0; l < (int)c->ad.Lb.size(); ++l) c->free_adaptive_level(l);
  c->free_adaptive_active();
  c->free_coeff(c->active);
  c->dfree(c->slots_u); c->dfree(c->slots_p);
  for (auto &p : c->bicg_pool) c->dfree(p);
  c->free_faces();
  for (int w = 0; w < 2; ++w)
    for (auto *v : {&c->mg_def[w], &c->mg_sol[w], &c->mg_tmp[w], &c->mg_t[w]})
      for (auto &p : *v) c->dfree(p);
  for (auto &p : c->pool) c->dfree(p);
  c->dfree(c->dot_partials); c->dfree(c->dot_out_d);
  if (c->dot_out_h) (void)hipHostFree(c->dot_out_h);
  if (c->stage_h) (void)hipHostFree(c->stage_h);
  c->free_graphs();
  for (int w = 0; w < 2; ++w) if (c->coarse_pin[w]) (void)hipHostFree(c->coarse_pin[w]);
  rccl_comm_destroy(c->rccl_state);
  if (c->cstream) { (void)hipStreamSynchronize(c->cstream); (void)hipStreamDestroy(c->cstream); }
  if (c->ev_src_ready) (void)hipEventDestroy(c->ev_src_ready);
  if (c->ev_import_done) (void)hipEventDestroy(c->ev_import_done);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

namespace {
// AGMG_TIMING=1: wall time of the host-side steps of agmg_set_level on stderr (per mesh change)
struct StepTimer {
  const bool on = getenv("AGMG_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char *what, int level) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[agmg timing] level %d %-28s %8.1f ms\n", level, what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};
}  // namespace

// Tables of one mesh (a multigrid level, its boundary-rows-only twin on a level with a refinement edge, or the active mesh
// of an adaptive context) with the constraint masks cmv / cmp.  primary: the level itself (exchange plan, global ids,
// state flags of the context); the twins and the active mesh of adaptive contexts live on one rank.
static int set_level_tables(agmg_ctx *c, DeviceLevel &lv, HostLevel &hl, int level, const agmg_level_desc *d,
                            const std::vector<uint8_t> &cmv, const std::vector<uint8_t> &cmp, bool primary, bool structured, StepTimer &tm) {
  const int dim = c->opt.dim, NV = ipow(3, dim), NP = ipow(2, dim);
  const bool has_plan = primary && (d->n_owned_nodes_v != d->n_nodes_v || d->n_owned_nodes_p != d->n_nodes_p || d->n_neighbors != 0);
  static const std::vector<uint32_t> no_send_nodes;
  auto send_nodes_of = [&](int w) -> const std::vector<uint32_t> & { return primary ? c->send_nodes_h[w][level] : no_send_nodes; };
  if (has_plan && c->dist()) {
    if (!c->comm_set) { set_error("set_level: bind the transport first (agmg_set_communicator)"); return -1; }
    if (d->n_neighbors < 0 || (d->n_neighbors > 0 && (!d->neighbor_rank || !d->send_offsets_v || !d->recv_offsets_v ||
                                                      !d->send_offsets_p || !d->recv_offsets_p))) {
      set_error("incomplete ghost exchange plan"); return -1;
    }
    if (d->n_owned_nodes_v < 0 || d->n_owned_nodes_v > d->n_nodes_v || d->n_owned_nodes_p < 0 || d->n_owned_nodes_p > d->n_nodes_p) {
      set_error("bad owned node counts"); return -1;
    }
    for (int k = 0; k < d->n_neighbors; ++k)
      if (d->neighbor_rank[k] < 0 || d->neighbor_rank[k] >= c->opt.world_size || d->neighbor_rank[k] == c->opt.rank) {
        set_error("bad neighbour rank"); return -1;
      }
    if (d->n_neighbors > 0 &&
        (d->recv_offsets_v[d->n_neighbors] != d->n_nodes_v - d->n_owned_nodes_v ||
         d->recv_offsets_p[d->n_neighbors] != d->n_nodes_p - d->n_owned_nodes_p)) {
      set_error("ghost groups do not cover the ghost nodes"); return -1;
    }
  } else if (has_plan) {
    set_error("ghost nodes / neighbours given but world_size == 1");
    return -1;
  }
  c->free_level(lv);
  lv.dim = dim;
  lv.replicated = primary && c->dist() && d->replicated != 0;
  if (lv.replicated && (has_plan || level == c->nl() - 1)) { set_error("a replicated level has no exchange plan and is not the finest level"); return -1; }
  lv.n_cells = d->n_cells; lv.n_nodes_v = d->n_nodes_v; lv.n_nodes_p = d->n_nodes_p;
  lv.cstride = (d->n_cells + 63) / 64 * 64;
  const size_t cs = lv.cstride;
  hl.cnv.assign(d->cell_nodes_v, d->cell_nodes_v + (size_t)lv.n_cells * NV);
  hl.cnp.assign(d->cell_nodes_p, d->cell_nodes_p + (size_t)lv.n_cells * NP);
  lv.n_owned_v = lv.n_nodes_v; lv.n_owned_p = lv.n_nodes_p;
  if (primary) for (int w = 0; w < 2; ++w) c->send_nodes_h[w][level].clear();
  if (has_plan) {  // ghost exchange plans of the two node spaces
    lv.n_owned_v = d->n_owned_nodes_v; lv.n_owned_p = d->n_owned_nodes_p;
    for (int vel = 0; vel < 2; ++vel) {
      Exchange &x = vel ? lv.xv : lv.xp;
      const int32_t *so = vel ? d->send_offsets_v : d->send_offsets_p, *ro = vel ? d->recv_offsets_v : d->recv_offsets_p;
      const uint32_t *sn = vel ? d->send_nodes_v : d->send_nodes_p;
      x.nc = vel ? dim : 1;
      x.n_owned = vel ? lv.n_owned_v : lv.n_owned_p;
      x.n_ghost = (vel ? lv.n_nodes_v : lv.n_nodes_p) - x.n_owned;
      x.nbr.assign(d->neighbor_rank, d->neighbor_rank + d->n_neighbors);
      x.send_off.assign(so, so + d->n_neighbors + 1);
      x.recv_off.assign(ro, ro + d->n_neighbors + 1);
      x.n_export = d->n_neighbors ? so[d->n_neighbors] : 0;
      for (int k = 0; k < d->n_neighbors; ++k)
        if (so[k + 1] < so[k] || ro[k + 1] < ro[k] || so[0] != 0 || ro[0] != 0) { set_error("exchange offsets must be non-decreasing from 0"); return -1; }
      if (x.n_export > 0 && !sn) { set_error("send_nodes missing"); return -1; }
      std::vector<uint32_t> &snh = c->send_nodes_h[vel ? 0 : 1][level];
      snh.assign(sn, sn + x.n_export);
      for (uint32_t node : snh) if ((int)node >= x.n_owned) { set_error("send_nodes must be owned nodes"); return -1; }
      CHECK(c->upload(&x.send_nodes, snh));
      CHECK(c->dalloc(&x.expbuf, (size_t)x.n_export * x.nc));
      CHECK(c->dalloc(&x.ghostbuf, (size_t)x.n_ghost * x.nc));
      lv.bytes_tables += snh.size() * 4;
    }
  }
  // symmetric exchange plans (SymExchange) of the pressure nodes and of the velocity nodes
  auto build_sym = [&](SymExchange &sx, const int nn_, const int32_t *ranks, const int32_t *offs, const uint32_t *nodes, const int n_nodes, const int nc) -> int {
    if (!ranks || !offs || !nodes || offs[0] != 0) { set_error("incomplete symmetric exchange plan"); return -1; }
    sx.nbr.assign(ranks, ranks + nn_);
    sx.off.assign(offs, offs + nn_ + 1);
    for (int k = 0; k < nn_; ++k) {
      if (sx.nbr[k] < 0 || sx.nbr[k] >= c->opt.world_size || sx.nbr[k] == c->opt.rank || (k > 0 && sx.nbr[k] <= sx.nbr[k - 1]) || sx.off[k + 1] < sx.off[k]) {
        set_error("symmetric exchange plan: neighbour ranks ascending, offsets non-decreasing"); return -1;
      }
    }
    sx.n_sym = sx.off[nn_];
    std::vector<int32_t> row(n_nodes, -1);
    std::vector<uint32_t> sh_nodes, cnt;
    for (int i = 0; i < sx.n_sym; ++i) {
      const uint32_t node = nodes[i];
      if ((int)node >= n_nodes) { set_error("symmetric exchange plan: node out of range"); return -1; }
      if (row[node] < 0) { row[node] = (int32_t)sh_nodes.size(); sh_nodes.push_back(node); cnt.push_back(0); }
      ++cnt[row[node]];
    }
    sx.n_sh = (int)sh_nodes.size();
    std::vector<uint32_t> ptr(sx.n_sh + 1, 0), pos(sx.n_sym);
    for (int r = 0; r < sx.n_sh; ++r) ptr[r + 1] = ptr[r] + cnt[r];
    std::vector<uint32_t> fill(ptr.begin(), ptr.end() - 1);
    std::vector<uint8_t> split(sx.n_sh, 0);
    for (int k = 0; k < nn_; ++k)      // neighbours ascending: the positions of a node come out in rank order
      for (int i = sx.off[k]; i < sx.off[k + 1]; ++i) {
        const int r = row[nodes[i]];
        pos[fill[r]++] = (uint32_t)i;
        if (sx.nbr[k] < c->opt.rank) ++split[r];
      }
    CHECK(c->upload(&sx.sh_nodes, sh_nodes)); CHECK(c->upload(&sx.ptr, ptr)); CHECK(c->upload(&sx.pos, pos));
    CHECK(c->upload(&sx.split, split)); CHECK(c->upload(&sx.row, row));
    CHECK(c->dalloc(&sx.sendbuf, (size_t)sx.n_sym * nc)); CHECK(c->dalloc(&sx.recvbuf, (size_t)sx.n_sym * nc)); CHECK(c->dalloc(&sx.ypart, (size_t)sx.n_sh * nc));
    lv.bytes_tables += (sh_nodes.size() + ptr.size() + pos.size() + row.size()) * 4 + split.size();
    return 0;
  };
  if (has_plan && d->n_sym_neighbors_p > 0 && !lv.replicated)
    CHECK(build_sym(lv.sp, d->n_sym_neighbors_p, d->sym_neighbor_rank_p, d->sym_offsets_p, d->sym_nodes_p, d->n_nodes_p, 1));
  if (has_plan && d->n_sym_neighbors_v > 0 && !lv.replicated)
    CHECK(build_sym(lv.sv, d->n_sym_neighbors_v, d->sym_neighbor_rank_v, d->sym_offsets_v, d->sym_nodes_v, d->n_nodes_v, dim));
  for (int vel = 0; vel < 2; ++vel) {
    const uint32_t *gn = vel ? d->global_nodes_v : d->global_nodes_p;
    if (gn && c->dist() && primary) {
      std::vector<uint32_t> g(gn, gn + (vel ? lv.n_nodes_v : lv.n_nodes_p));
      CHECK(c->upload(vel ? &lv.gid_v : &lv.gid_p, g));
      if (level == c->nl() - 1) c->gid_fin_h[vel] = g;
    }
  }
  tm.lap("copy + exchange plan", level);
  for (int vel = 0; vel < 2; ++vel) {
    const int nloc = vel ? NV : NP, nn = vel ? lv.n_nodes_v : lv.n_nodes_p;
    const std::vector<uint32_t> &cn = vel ? hl.cnv : hl.cnp;
    const std::vector<uint8_t> &cm = vel ? cmv : cmp;
    const std::vector<uint32_t> &snh = send_nodes_of(vel ? 0 : 1);
    std::vector<uint32_t> T((size_t)nloc * cs);
    std::vector<uint32_t> ptr((size_t)nn + 1, 0), list((size_t)lv.n_cells * nloc + snh.size());
    {
      std::vector<uint8_t> bad(64, 0);
      parallel_for((size_t)lv.n_cells, [&](size_t c0, size_t c1, int t) {
        for (size_t cell = c0; cell < c1; ++cell)
          for (int a = 0; a < nloc; ++a) {
            const uint32_t node = cn[cell * nloc + a];
            if ((int)node >= nn) { bad[t & 63] = 1; continue; }
            T[(size_t)a * cs + cell] = node | ((uint32_t)cm[node] << 29);
          }
      });
      for (uint8_t f : bad) if (f) { set_error(vel ? "cell_nodes_v out of range" : "cell_nodes_p out of range"); return -1; }
    }
    for (size_t k = 0, ne = (size_t)lv.n_cells * nloc; k < ne; ++k) ++ptr[cn[k] + 1];
    for (size_t cell = lv.n_cells; cell < cs; ++cell)
      for (int a = 0; a < nloc; ++a) T[(size_t)a * cs + cell] = T[(size_t)a * cs + lv.n_cells - 1];
    for (uint32_t node : snh) ++ptr[node + 1];   // remote partial sums (receive buffer entries)
    for (int i = 0; i < nn; ++i) ptr[i + 1] += ptr[i];
    std::vector<uint32_t> fill(ptr.begin(), ptr.end() - 1);
    for (int cell = 0; cell < lv.n_cells; ++cell)
      for (int a = 0; a < nloc; ++a) {
        const uint32_t node = cn[(size_t)cell * nloc + a];
        list[fill[node]++] = (uint32_t)((size_t)a * cs + cell);
      }
    for (size_t k = 0; k < snh.size(); ++k) list[fill[snh[k]]++] = (uint32_t)((size_t)nloc * cs + k);
    if (!vel) {
      uint32_t max_val = 0;
      for (int i = 0; i < nn; ++i) max_val = std::max(max_val, ptr[i + 1] - ptr[i]);
      lv.mass_fusable = !c->ldist(lv) && lv.n_cells <= MS_THREADS && (size_t)lv.n_cells * nloc <= (size_t)MS_MAX_SLOTS &&
                        nn <= MS_MAX_NODES && max_val <= 8;
      // column structure of the assembled mass rows: the nodes of the cells around each node,
      // ascending; a patch with more than 3^dim nodes (irregular vertex) keeps the matrix-free path
      if (c->use_ell) {
        const int W = NV;   // 3^dim
        const size_t npad = ((size_t)nn + 63) / 64 * 64;
        std::vector<uint32_t> col((size_t)W * npad);
        std::vector<uint8_t> irregular(64, 0);
        parallel_for((size_t)nn, [&](size_t i0, size_t i1, int t) {
          uint32_t nb[64];
          for (size_t i = i0; i < i1; ++i) {
            int cnt = 0;
            bool ok_i = true;
            for (uint32_t k = ptr[i]; k < ptr[i + 1] && ok_i; ++k) {
              if (list[k] >= (uint32_t)((size_t)nloc * cs)) continue;   // (an entry of the receive buffer)
              const size_t cell = list[k] % cs;
              for (int a = 0; a < nloc; ++a) {
                const uint32_t node = cn[cell * nloc + a];
                int q = 0;
                while (q < cnt && nb[q] != node) ++q;
                if (q == cnt) {
                  if (cnt == W) { ok_i = false; break; }
                  nb[cnt++] = node;
                }
              }
            }
            if (!ok_i) { irregular[t & 63] = 1; continue; }
            std::sort(nb, nb + cnt);
            for (int k = 0; k < W; ++k) col[(size_t)k * npad + i] = k < cnt ? nb[k] : (uint32_t)i;
          }
        });
        bool ok = true;
        for (uint8_t f : irregular) if (f) ok = false;
        for (size_t i = nn; i < npad; ++i)
          for (int k = 0; k < W; ++k) col[(size_t)k * npad + i] = 0;
        if (ok) {
          CHECK(c->upload(&lv.ell_col, col));
          lv.ell_w = W;
          lv.ell_stride = (int)npad;
          lv.bytes_tables += col.size() * 4;
          if (c->use_ell_classes) {
            // classes of rows by their column offsets (structured numberings have a few dozen; give up beyond 1024)
            std::map<std::vector<int32_t>, uint16_t> classes;
            std::vector<uint16_t> cls(nn);
            std::vector<int32_t> key(W), off;
            bool few = true;
            std::vector<int32_t> last;
            uint16_t last_cls = 0;
            for (int i = 0; i < nn && few; ++i) {
              for (int k = 0; k < W; ++k) key[k] = (int32_t)col[(size_t)k * npad + i] - (int32_t)i;
              if (key == last) { cls[i] = last_cls; continue; }   // (neighbouring rows mostly share their class)
              auto it = classes.find(key);
              if (it == classes.end()) {
                if (classes.size() >= 1024) { few = false; break; }
                it = classes.emplace(key, (uint16_t)classes.size()).first;
                off.insert(off.end(), key.begin(), key.end());
              }
              cls[i] = it->second;
              last = key; last_cls = it->second;
            }
            if (few) {
              CHECK(c->upload(&lv.ell_cls, cls)); CHECK(c->upload(&lv.ell_cls_off, off));
              lv.bytes_tables += cls.size() * 2 + off.size() * 4;
            }
          }
          if (!snh.empty()) {   // received shares of the owned rows other ranks ghost: entry k of the receive buffer belongs to snh[k]
            std::vector<int32_t> xrow(lv.n_owned_p, -1);
            std::vector<uint32_t> xptr(1, 0), xent(snh.size());
            for (uint32_t node : snh) if (xrow[node] < 0) { xrow[node] = (int32_t)xptr.size() - 1; xptr.push_back(0); }
            for (uint32_t node : snh) ++xptr[xrow[node] + 1];
            for (size_t k = 1; k < xptr.size(); ++k) xptr[k] += xptr[k - 1];
            std::vector<uint32_t> xfill(xptr.begin(), xptr.end() - 1);
            for (size_t k = 0; k < snh.size(); ++k) xent[xfill[xrow[snh[k]]]++] = (uint32_t)k;
            CHECK(c->upload(&lv.ell_xrow, xrow)); CHECK(c->upload(&lv.ell_xptr, xptr)); CHECK(c->upload(&lv.ell_xent, xent));
            lv.bytes_tables += (xrow.size() + xptr.size() + xent.size()) * 4;
          }
        }
      }
    }
    if (vel) { CHECK(c->upload(&lv.cnvT, T)); CHECK(c->upload(&lv.pull_ptr_v, ptr)); CHECK(c->upload(&lv.pull_list_v, list)); CHECK(c->upload(&lv.cmask_v, cmv)); }
    else { CHECK(c->upload(&lv.cnpT, T)); CHECK(c->upload(&lv.pull_ptr_p, ptr)); CHECK(c->upload(&lv.pull_list_p, list)); CHECK(c->upload(&lv.cmask_p, cmp)); }
    lv.bytes_tables += (T.size() + ptr.size() + list.size()) * 4;
    lv.bytes_constraints += cm.size();
    tm.lap(vel ? "velocity cell/pull tables" : "pressure tables + ELL columns", level);
  }
  lv.geom = d->geometry_kind;
  const int NQ = NV;
  if (lv.geom == AGMG_GEOM_CARTESIAN) {
    if (!d->cell_h) { set_error("cell_h missing"); return -1; }
    std::vector<double> hT((size_t)dim * cs, 1.0);
    for (int cell = 0; cell < lv.n_cells; ++cell)
      for (int k = 0; k < dim; ++k) {
        const double h = d->cell_h[(size_t)cell * dim + k];
        if (!(h > 0)) { set_error("cell_h must be positive"); return -1; }
        hT[(size_t)k * cs + cell] = h;
      }
    CHECK(c->upload(&lv.cell_hT, hT));
    lv.bytes_tables += hT.size() * 8;
  } else if (lv.geom == AGMG_GEOM_GENERAL) {
    if (!d->inv_jacobian || !d->jxw) { set_error("inv_jacobian / jxw missing"); return -1; }
    const int dd = dim * dim;
    std::vector<double> J((size_t)NQ * dd * cs, 0.0), W((size_t)NQ * cs, 1.0);
    for (int cell = 0; cell < lv.n_cells; ++cell)
      for (int q = 0; q < NQ; ++q) {
        for (int e = 0; e < dd; ++e) J[((size_t)q * dd + e) * cs + cell] = d->inv_jacobian[((size_t)cell * NQ + q) * dd + e];
        W[(size_t)q * cs + cell] = d->jxw[(size_t)cell * NQ + q];
      }
    CHECK(c->upload(&lv.invJT, J));
    CHECK(c->upload(&lv.jxwT, W));
    lv.bytes_tables += (J.size() + W.size()) * 8;
  } else { set_error("bad geometry_kind"); return -1; }
  hl.parent.clear(); hl.child.clear();
  if (d->parent_cell && d->child_index) {
    hl.parent.assign(d->parent_cell, d->parent_cell + lv.n_cells);
    hl.child.assign(d->child_index, d->child_index + lv.n_cells);
  }
  tm.lap("geometry", level);
  CHECK(c->build_block_tables(lv, hl, cmv, send_nodes_of(0)));
  if (lv.sv.active() && lv.bt.built) {
    // every row of the velocity exchange must be a shared node of the block tables (interface nodes always are)
    std::vector<uint32_t> shn(lv.bt.n_shared), svn(lv.sv.n_sh);
    if (lv.bt.n_shared) AGMG_HIP_CHECK(hipMemcpy(shn.data(), lv.bt.sh_nodes, shn.size() * 4, hipMemcpyDeviceToHost));
    if (lv.sv.n_sh) AGMG_HIP_CHECK(hipMemcpy(svn.data(), lv.sv.sh_nodes, svn.size() * 4, hipMemcpyDeviceToHost));
    std::vector<int32_t> at(lv.n_nodes_v, -1), symrow(lv.bt.n_shared, -1);
    for (int i = 0; i < lv.bt.n_shared; ++i) at[shn[i] & NODE_MASK] = i;
    std::vector<uint32_t> shidx(lv.sv.n_sh, 0);
    bool ok = true;
    for (int r = 0; r < lv.sv.n_sh; ++r) {
      const int i = at[svn[r]];
      if (i < 0) { ok = false; break; }
      shidx[r] = (uint32_t)i; symrow[i] = r;
    }
    if (ok) { CHECK(c->upload(&lv.sv_shidx, shidx)); CHECK(c->upload(&lv.bt_symrow, symrow)); }
  }
  tm.lap("block tables", level);
  lv.bytes_tables += lv.bt.bytes;
  if (c->use_bricks && structured) {   // (not for the partly refined meshes of adaptive contexts)
    CHECK(c->build_brick_tables(lv, hl, cmv, send_nodes_of(0)));
    lv.bytes_tables += lv.bk.bytes;
  }
  CHECK(c->dalloc(&lv.inv_diag_A, (size_t)lv.n_dofs_v()));
  CHECK(c->dalloc(&lv.inv_diag_M, (size_t)lv.n_dofs_p()));
  lv.set = true;
  return 0;
}

void agmg_ctx::free_adaptive_level(int l) {
  if (l < (int)ad.Lb.size()) { free_level(ad.Lb[l]); ad.Hb[l] = HostLevel(); }
  for (int w = 0; w < 2; ++w) {
    if (l < (int)ad.edge[w].size()) { dfree(ad.edge[w][l]); dfree(ad.notedge[w][l]); }
    if (l < (int)ad.copy_act[w].size()) { ad.copy_act[w][l].clear(); ad.copy_lev[w][l].clear(); dfree(ad.copy_act_d[w][l]); dfree(ad.copy_lev_d[w][l]); }
    for (int k = l; k <= l + 1; ++k)   // restriction tables that involve this level as the fine or the coarse one
      if (k < (int)ad.rpull_ptr[w].size()) { dfree(ad.rpull_ptr[w][k]); dfree(ad.rpull_list[w][k]); }
  }
}
void agmg_ctx::free_adaptive_active() {
  free_level(ad.act);
  ad.hact = HostLevel();
  dfree(ad.xres);
  ad.xres_cap = 0;
  for (HangTables &t : ad.hang) {
    dfree(t.hn); dfree(t.hm); dfree(t.tm); dfree(t.th); dfree(t.hptr); dfree(t.tptr); dfree(t.hw); dfree(t.tw); dfree(t.saved);
    t = HangTables();
  }
  ad.on = false;
}

int agmg_set_level(agmg_ctx *c, int level, const agmg_level_desc *d) {
  if (!c) { set_error("null context"); return -1; }
  StepTimer tm;
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  CHECK(c->sync());
  if (level == AGMG_ACTIVE && !d) {   // uniformly refined mesh: the active mesh is the finest level
    if (c->ad.on) { c->free_adaptive_active(); c->free_faces(); c->free_coeff(c->active); }
    return 0;
  }
  if (level != AGMG_ACTIVE && (level < 0 || level >= c->nl())) { set_error("bad level"); return -1; }
  if (!d || d->n_cells <= 0 || !d->cell_nodes_v || !d->cell_nodes_p || !d->constrained_v) { set_error("incomplete level descriptor"); return -1; }
  const int dim = c->opt.dim;
  if ((unsigned)d->n_nodes_v > NODE_MASK || (unsigned)d->n_nodes_p > NODE_MASK) { set_error("too many nodes for 29-bit indices"); return -1; }
  std::vector<uint8_t> cmv(d->constrained_v, d->constrained_v + d->n_nodes_v);
  std::vector<uint8_t> cmp(d->n_nodes_p, 0);
  if (d->constrained_p) cmp.assign(d->constrained_p, d->constrained_p + d->n_nodes_p);
  const uint8_t vmask = (uint8_t)((1u << dim) - 1u);
  for (auto &m : cmv) m &= vmask;
  for (auto &m : cmp) m &= 1u;
  const bool active = level == AGMG_ACTIVE;
  if (active && c->dist()) { set_error("an active mesh with tables of its own (adaptive refinement): one-rank contexts only"); return -2; }
  HostLevel &hl = active ? c->ad.hact : c->H[level];
  // hanging-node constraints: on the active mesh of an adaptive context (resolved around the device operators), or on
  // the single level of a one-level context (the first slice: resolved on the host around agmg_vmult)
  HostLevel parsed;
  for (int sp = 0; sp < 2; ++sp) {
    const int nh = sp == 0 ? d->n_hanging_v : d->n_hanging_p, nn = sp == 0 ? d->n_nodes_v : d->n_nodes_p;
    const uint32_t *hn = sp == 0 ? d->hanging_nodes_v : d->hanging_nodes_p, *hm = sp == 0 ? d->hanging_master_v : d->hanging_master_p;
    const int32_t *hp = sp == 0 ? d->hanging_ptr_v : d->hanging_ptr_p;
    const double *hw = sp == 0 ? d->hanging_weight_v : d->hanging_weight_p;
    parsed.cmask[sp] = sp == 0 ? cmv : cmp;
    if (nh <= 0) continue;
    if (!active && (c->nl() != 1 || c->dist())) { set_error("hanging-node constraints belong to the active mesh (AGMG_ACTIVE) or to the level of a one-level, one-rank context"); return -2; }
    if (!hn || !hp || !hm || !hw) { set_error("incomplete hanging-node tables"); return -1; }
    if (hp[0] != 0) { set_error("hanging_ptr must start at 0"); return -1; }
    for (int i = 0; i < nh; ++i) if (hp[i + 1] < hp[i]) { set_error("hanging_ptr must be non-decreasing"); return -1; }
    parsed.hang_n[sp].assign(hn, hn + nh); parsed.hang_ptr[sp].assign(hp, hp + nh + 1);
    parsed.hang_m[sp].assign(hm, hm + hp[nh]); parsed.hang_w[sp].assign(hw, hw + hp[nh]);
    std::vector<uint8_t> is_h(nn, 0);
    for (uint32_t h : parsed.hang_n[sp]) { if ((int)h >= nn || is_h[h]) { set_error("hanging node out of range or listed twice"); return -1; } is_h[h] = 1; }
    for (uint32_t m : parsed.hang_m[sp]) if ((int)m >= nn || is_h[m]) { set_error("hanging-node master out of range or hanging itself"); return -1; }
  }
  // refinement edge and copy table of a level of local smoothing
  bool any_edge = false;
  std::vector<uint8_t> ev, ep;
  if (!active && (d->refinement_edge_v || d->refinement_edge_p)) {
    if (!d->refinement_edge_v || !d->refinement_edge_p) { set_error("refinement-edge masks come for both spaces"); return -1; }
    ev.assign(d->refinement_edge_v, d->refinement_edge_v + d->n_nodes_v);
    ep.assign(d->refinement_edge_p, d->refinement_edge_p + d->n_nodes_p);
    for (uint8_t e : ev) any_edge |= e != 0;
    for (uint8_t e : ep) any_edge |= e != 0;
    if (any_edge && c->dist()) { set_error("refinement edges (local smoothing on adaptive meshes): one-rank contexts only"); return -2; }
  }
  if (!active && (d->n_copy_v > 0 || d->n_copy_p > 0)) {
    if (c->dist()) { set_error("copy tables (local smoothing on adaptive meshes): one-rank contexts only"); return -2; }
    for (int sp = 0; sp < 2; ++sp) {
      const int nc = sp == 0 ? d->n_copy_v : d->n_copy_p, nn = sp == 0 ? d->n_nodes_v : d->n_nodes_p;
      const uint32_t *ca = sp == 0 ? d->copy_active_v : d->copy_active_p, *cl = sp == 0 ? d->copy_level_v : d->copy_level_p;
      if (nc < 0 || (nc > 0 && (!ca || !cl))) { set_error("incomplete copy tables"); return -1; }
      std::vector<uint8_t> seen(nn, 0);
      for (int i = 0; i < nc; ++i) {
        if ((int)cl[i] >= nn || seen[cl[i]]) { set_error("copy table: level node out of range or listed twice"); return -1; }
        seen[cl[i]] = 1;
      }
    }
  }
  if (active) {
    c->free_adaptive_active();
    c->free_faces();
    c->free_coeff(c->active);
  } else {
    c->free_adaptive_level(level);
    if (level == c->nl() - 1 && !c->ad.on) {   // face tables and the active coefficients refer to the cells of the active mesh
      c->free_faces();
      c->free_coeff(c->active);
    }
  }
  DeviceLevel &lv = active ? c->ad.act : c->L[level];
  std::vector<uint8_t> cmv_op = cmv, cmp_op = cmp;       // the masks of the level operator: boundary + refinement edge
  if (any_edge) {
    for (int i = 0; i < d->n_nodes_v; ++i) if (ev[i]) cmv_op[i] |= vmask;
    for (int i = 0; i < d->n_nodes_p; ++i) if (ep[i]) cmp_op[i] |= 1u;
  }
  const bool structured = !active && !d->refinement_edge_v && d->n_copy_v <= 0;
  CHECK(set_level_tables(c, lv, hl, active ? -1 : level, d, cmv_op, cmp_op, !active, structured, tm));
  for (int sp = 0; sp < 2; ++sp) {
    hl.hang_n[sp] = parsed.hang_n[sp]; hl.hang_m[sp] = parsed.hang_m[sp]; hl.hang_ptr[sp] = parsed.hang_ptr[sp]; hl.hang_w[sp] = parsed.hang_w[sp];
    hl.cmask[sp] = parsed.cmask[sp];
  }
  if (!active) {
    const int nl = c->nl();
    agmg_ctx::Adaptive &ad = c->ad;
    if ((int)ad.Lb.size() != nl) {
      ad.Lb.resize(nl); ad.Hb.resize(nl);
      for (int w = 0; w < 2; ++w) {
        ad.edge[w].assign(nl, nullptr); ad.notedge[w].assign(nl, nullptr);
        ad.copy_act[w].assign(nl, {}); ad.copy_lev[w].assign(nl, {});
        ad.copy_act_d[w].assign(nl, nullptr); ad.copy_lev_d[w].assign(nl, nullptr);
        ad.rpull_ptr[w].assign(nl, nullptr); ad.rpull_list[w].assign(nl, nullptr);
      }
    }
    if (any_edge) {
      // the twin with the boundary rows alone, and the masks that split a vector into edge and other entries
      CHECK(set_level_tables(c, ad.Lb[level], ad.Hb[level], level, d, cmv, cmp, false, false, tm));
      for (int w = 0; w < 2; ++w) {
        const std::vector<uint8_t> &e = w == 0 ? ev : ep;
        const uint8_t all = w == 0 ? vmask : (uint8_t)1;
        std::vector<uint8_t> on(e.size()), off(e.size());
        for (size_t i = 0; i < e.size(); ++i) { on[i] = e[i] ? all : 0; off[i] = e[i] ? 0 : all; }
        CHECK(c->upload(&ad.edge[w][level], on));
        CHECK(c->upload(&ad.notedge[w][level], off));
      }
    }
    for (int w = 0; w < 2; ++w) {
      const int nc = w == 0 ? d->n_copy_v : d->n_copy_p;
      if (nc <= 0) continue;
      const uint32_t *ca = w == 0 ? d->copy_active_v : d->copy_active_p, *cl = w == 0 ? d->copy_level_v : d->copy_level_p;
      ad.copy_act[w][level].assign(ca, ca + nc);
      ad.copy_lev[w][level].assign(cl, cl + nc);
      CHECK(c->upload(&ad.copy_act_d[w][level], ad.copy_act[w][level]));
      CHECK(c->upload(&ad.copy_lev_d[w][level], ad.copy_lev[w][level]));
    }
    if (level + 1 < nl) c->L[level + 1].has_transfer = false;
    lv.has_transfer = false;
  } else {
    // the hanging-node CSRs and their transposes (master -> hanging nodes, terms in the order of the table) on the device
    agmg_ctx::Adaptive &ad = c->ad;
    for (int sp = 0; sp < 2; ++sp) {
      HangTables &t = ad.hang[sp];
      const int nh = (int)hl.hang_n[sp].size(), nn = sp == 0 ? d->n_nodes_v : d->n_nodes_p;
      t.n_hang = nh;
      std::vector<int32_t> slot(nn, -1), tptr(1, 0);
      std::vector<uint32_t> tm;
      for (uint32_t m : hl.hang_m[sp]) if (slot[m] < 0) { slot[m] = (int32_t)tm.size(); tm.push_back(m); tptr.push_back(0); }
      for (uint32_t m : hl.hang_m[sp]) ++tptr[slot[m] + 1];
      for (size_t k = 1; k < tptr.size(); ++k) tptr[k] += tptr[k - 1];
      std::vector<int32_t> fill(tptr.begin(), tptr.end() - 1);
      std::vector<uint32_t> th(hl.hang_m[sp].size());
      std::vector<double> tw(hl.hang_m[sp].size());
      for (int i = 0; i < nh; ++i)
        for (int k = hl.hang_ptr[sp][i]; k < hl.hang_ptr[sp][i + 1]; ++k) {
          const int at = fill[slot[hl.hang_m[sp][k]]]++;
          th[at] = hl.hang_n[sp][i]; tw[at] = hl.hang_w[sp][k];
        }
      t.n_masters = (int)tm.size();
      if (hl.hang_ptr[sp].empty()) hl.hang_ptr[sp].assign(1, 0);
      CHECK(c->upload(&t.hn, hl.hang_n[sp])); CHECK(c->upload(&t.hptr, hl.hang_ptr[sp]));
      CHECK(c->upload(&t.hm, hl.hang_m[sp])); CHECK(c->upload(&t.hw, hl.hang_w[sp]));
      CHECK(c->upload(&t.tm, tm)); CHECK(c->upload(&t.tptr, tptr)); CHECK(c->upload(&t.th, th)); CHECK(c->upload(&t.tw, tw));
      CHECK(c->dalloc(&t.saved, (size_t)nh * (sp == 0 ? dim : 1)));
    }
    ad.on = true;
  }
  c->mg_alloc = false; c->diag_ready = false; c->smoothers_ready = false;
  c->free_graphs();
  for (auto &p : c->pool) c->dfree(p);
  c->pool.clear();
  for (auto &p : c->bicg_pool) c->dfree(p);
  c->bicg_pool.clear();
  return 0;
}

static int set_coefficients_on(agmg_ctx *c, DeviceLevel &lv, CoeffData &cd, bool is_active, const agmg_cell_data *d);

int agmg_set_coefficients(agmg_ctx *c, int level, const agmg_cell_data *d) {
  if (!c || !d) { set_error("null argument"); return -1; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  if (d->enable_prescribed_dilation) {   // active level only (local_smoothing.cc:456-462)
    if (level != AGMG_ACTIVE) { set_error("prescribed-dilation tables belong to the active level (AGMG_ACTIVE)"); return -1; }
    if (!d->dilation_lhs_term) { set_error("dilation_lhs_term missing"); return -1; }
  }
  if (d->enable_newton_derivatives) {
    // the tables exist on the active level only (local_smoothing.cc:447-663); the level operators
    // of the preconditioner stay Picard operators
    if (level != AGMG_ACTIVE) { set_error("Newton-derivative tables belong to the active level (AGMG_ACTIVE)"); return -1; }
    if (!d->strain_rate_table || !d->newton_factor_wrt_strain_rate || !d->newton_factor_wrt_pressure) {
      set_error("Newton-derivative tables missing"); return -1;
    }
  }
  if (level != AGMG_ACTIVE && (level < 0 || level >= c->nl())) { set_error("bad level"); return -1; }
  DeviceLevel &lv0 = level == AGMG_ACTIVE ? c->act() : c->L[level];
  if (!lv0.set) { set_error("set_coefficients before set_level"); return -1; }
  if (d->n_q_viscosity != 1 && d->n_q_viscosity != ipow(3, c->opt.dim)) { set_error("n_q_viscosity must be 1 or 3^dim"); return -1; }
  if (!d->viscosity) { set_error("viscosity table missing"); return -1; }
  CHECK(set_coefficients_on(c, lv0, level == AGMG_ACTIVE ? c->active : lv0.coeff, level == AGMG_ACTIVE, d));
  // a level with a refinement edge: its twin with the boundary rows alone carries the same coefficients
  if (level != AGMG_ACTIVE && c->has_edge(level)) CHECK(set_coefficients_on(c, c->ad.Lb[level], c->ad.Lb[level].coeff, false, d));
  c->diag_ready = false; c->smoothers_ready = false;
  return 0;
}

static int set_coefficients_on(agmg_ctx *c, DeviceLevel &lv, CoeffData &cd, bool is_active, const agmg_cell_data *d) {
  const int level = is_active ? AGMG_ACTIVE : 0;
  const int NQ = ipow(3, c->opt.dim);
  if (d->n_q_viscosity != 1 && d->n_q_viscosity != NQ) { set_error("n_q_viscosity must be 1 or 3^dim"); return -1; }
  if (!d->viscosity) { set_error("viscosity table missing"); return -1; }
  const size_t cs = lv.cstride;
  {
    // the table crosses PCIe as it is ([cell][q]); transposition to [q][cstride] and the positivity
    // check run on the device (a host loop over 5.7e7 strided entries cost 0.6 s at r7)
    const size_t n_in = (size_t)lv.n_cells * d->n_q_viscosity, n_out = (size_t)d->n_q_viscosity * cs;
    double *raw = nullptr, *etaT = nullptr;
    int *bad = nullptr;
    CHECK(c->dalloc(&raw, n_in));
    int rc = c->dalloc(&etaT, n_out);
    if (!rc) rc = c->dalloc(&bad, 1);
    int bad_h = 0;
    if (!rc && hipMemcpyAsync(raw, d->viscosity, n_in * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = -3;
    if (!rc && hipMemsetAsync(bad, 0, sizeof(int), c->stream) != hipSuccess) rc = -3;
    if (!rc) rc = vec_transpose_table(c->stream, etaT, raw, lv.n_cells, d->n_q_viscosity, cs, bad);
    if (!rc && hipMemcpyAsync(&bad_h, bad, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = -3;
    if (!rc) rc = c->sync();
    c->dfree(raw); c->dfree(bad);
    if (rc || bad_h) {
      c->dfree(etaT);
      if (bad_h) { set_error("viscosity must be positive"); return -1; }
      if (rc == -3) set_error("HIP error while uploading the viscosity table");
      return rc;
    }
    c->dfree(cd.etaT);
    cd.etaT = etaT;
  }
  const size_t T_size = (size_t)d->n_q_viscosity * cs;
  cd.nqv = d->n_q_viscosity;
  cd.lambda = d->pressure_scaling;
  cd.is_compressible = d->is_compressible;
  cd.bytes = T_size * 8;
  c->dfree(cd.nw_strain); c->dfree(cd.nw_dEde); c->dfree(cd.nw_dEdp);
  cd.newton = false;
  if (d->enable_newton_derivatives) {
    const int dim = c->opt.dim, NS = dim * (dim + 1) / 2;
    std::vector<double> E((size_t)NQ * NS * cs, 0.0), D((size_t)NQ * NS * cs, 0.0), P((size_t)NQ * cs, 0.0);
    for (int cell = 0; cell < lv.n_cells; ++cell)
      for (int q = 0; q < NQ; ++q) {
        for (int k = 0; k < NS; ++k) {
          E[((size_t)q * NS + k) * cs + cell] = d->strain_rate_table[((size_t)cell * NQ + q) * NS + k];
          D[((size_t)q * NS + k) * cs + cell] = d->newton_factor_wrt_strain_rate[((size_t)cell * NQ + q) * NS + k];
        }
        P[(size_t)q * cs + cell] = d->newton_factor_wrt_pressure[(size_t)cell * NQ + q];
      }
    CHECK(c->upload(&cd.nw_strain, E)); CHECK(c->upload(&cd.nw_dEde, D)); CHECK(c->upload(&cd.nw_dEdp, P));
    cd.newton = true;
    cd.nw_symmetrize = d->symmetrize_newton_system ? 1 : 0;
    cd.bytes += (E.size() + D.size() + P.size()) * 8;
  }
  c->dfree(cd.dil_lhs); c->dfree(cd.dil_de); c->dfree(cd.dil_dp);
  cd.dilation = false;
  if (d->enable_prescribed_dilation) {
    const int dim = c->opt.dim, NS = dim * (dim + 1) / 2;
    std::vector<double> Lh((size_t)NQ * cs, 0.0);
    for (int cell = 0; cell < lv.n_cells; ++cell)
      for (int q = 0; q < NQ; ++q) Lh[(size_t)q * cs + cell] = d->dilation_lhs_term[(size_t)cell * NQ + q];
    CHECK(c->upload(&cd.dil_lhs, Lh));
    cd.bytes += Lh.size() * 8;
    if (d->enable_newton_derivatives && d->dilation_derivative_wrt_strain_rate && d->dilation_derivative_wrt_pressure) {
      std::vector<double> De((size_t)NQ * NS * cs, 0.0), Dp((size_t)NQ * cs, 0.0);
      for (int cell = 0; cell < lv.n_cells; ++cell)
        for (int q = 0; q < NQ; ++q) {
          for (int k = 0; k < NS; ++k)
            De[((size_t)q * NS + k) * cs + cell] = d->dilation_derivative_wrt_strain_rate[((size_t)cell * NQ + q) * NS + k];
          Dp[(size_t)q * cs + cell] = d->dilation_derivative_wrt_pressure[(size_t)cell * NQ + q];
        }
      CHECK(c->upload(&cd.dil_de, De)); CHECK(c->upload(&cd.dil_dp, Dp));
      cd.bytes += (De.size() + Dp.size()) * 8;
    }
    cd.dilation = true;
  }
  cd.set = true;
  c->diag_ready = false; c->smoothers_ready = false;
  // brick levels: viscosity rows and cell extents in brick order (kernels_brickq.hip)
  c->dfree(cd.brick_geo);
  if (lv.bk.built && lv.geom == AGMG_GEOM_CARTESIAN && c->use_bricks) {
    CHECK(c->dalloc(&cd.brick_geo, brick_geo_doubles(lv, cd.nqv)));
    CHECK(launch_brick_geo(c->stream, lv, cd));
    cd.bytes += brick_geo_doubles(lv, cd.nqv) * 8;
  }
  // assembled rows of the mass operator with these coefficients (MassMatrixOperator of this level)
  if (cd.ell_stride != lv.ell_stride) { c->dfree(cd.ell_val); cd.ell_stride = 0; }
  // (not for the active-mesh coefficients: the solve only applies the level operators' mass
  // matrices; an active-mesh mass vmult keeps the matrix-free kernels)
  if (lv.ell_col && c->use_ell && level != AGMG_ACTIVE) {
    if (!cd.ell_val) CHECK(c->dalloc(&cd.ell_val, (size_t)lv.ell_w * lv.ell_stride));
    cd.ell_stride = lv.ell_stride;
    CHECK(launch_mass_ell_assemble(c->stream, lv, cd));
    cd.bytes += (size_t)lv.ell_w * lv.ell_stride * 8;
  }
  return 0;
}

int agmg_set_free_surface_faces(agmg_ctx *c, const agmg_face_data *d) {
  if (!c) { set_error("null context"); return -1; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  CHECK(c->sync());
  c->free_faces();
  if (c->dist()) {   // collective: every rank calls this, also ranks without faces
    double n = d ? (double)d->n_faces : 0.0;
    CHECK(c->allreduce_host(&n, 1));
    c->faces.global_active = n > 0;
    if (c->faces.global_active && !c->L.empty() && c->fin().set) {
      // owned nodes that other ranks ghost -> their entries of the receive buffer
      const std::vector<uint32_t> &sn = c->send_nodes_h[0][c->nl() - 1];
      std::map<uint32_t, std::vector<uint32_t>> by_node;
      for (size_t k = 0; k < sn.size(); ++k) by_node[sn[k]].push_back((uint32_t)k);
      std::vector<uint32_t> en, ep(1, 0), ee;
      for (auto &kv : by_node) {
        en.push_back(kv.first);
        ee.insert(ee.end(), kv.second.begin(), kv.second.end());
        ep.push_back((uint32_t)ee.size());
      }
      CHECK(c->upload(&c->faces.exp_nodes, en)); CHECK(c->upload(&c->faces.exp_ptr, ep)); CHECK(c->upload(&c->faces.exp_ent, ee));
      c->faces.n_exp_nodes = (int)en.size();
    }
  }
  if (!d || d->n_faces == 0) return 0;
  if (d->n_faces < 0 || !d->face_cell || !d->face_number || !d->stabilization_term) { set_error("free-surface face tables missing"); return -1; }
  if ((d->normal == nullptr) != (d->jxw == nullptr)) { set_error("face normal and jxw come together"); return -1; }
  if (c->L.empty() || !c->fin().set) { set_error("set_free_surface_faces before set_level"); return -1; }
  if (c->ad.on) { set_error("free-surface faces are not built for adaptive contexts (SURVEY.md 8 f1)"); return -2; }
  const DeviceLevel &lv = c->fin();
  const HostLevel &hl = c->H[c->nl() - 1];
  const int dim = lv.dim, NV = ipow(3, dim), NF = ipow(3, dim - 1), NQF = NF, nf = d->n_faces;
  if (!d->normal && !lv.cell_hT) { set_error("general geometry needs face normals and JxW"); return -1; }
  if ((uint64_t)nf >= (1ull << 28)) { set_error("too many free-surface faces"); return -1; }
  for (int f = 0; f < nf; ++f)
    if ((int)d->face_cell[f] >= lv.n_cells || d->face_number[f] >= 2 * dim) { set_error("free-surface face table out of range"); return -1; }
  std::vector<double> hT;
  if (!d->normal) {
    hT.resize((size_t)dim * lv.cstride);
    AGMG_HIP_CHECK(hipMemcpy(hT.data(), lv.cell_hT, hT.size() * sizeof(double), hipMemcpyDeviceToHost));
  }
  const double wq[3] = {5. / 18., 8. / 18., 5. / 18.};
  std::vector<uint32_t> fnode((size_t)nf * NF);
  std::vector<double> stab(d->stabilization_term, d->stabilization_term + (size_t)nf * NQF * dim), njxw((size_t)nf * NQF * dim);
  std::map<uint32_t, std::vector<uint32_t>> by_node;   // node -> (face << 4 | a), faces ascending
  for (int f = 0; f < nf; ++f) {
    const int cell = d->face_cell[f], dir = d->face_number[f] / 2, side = d->face_number[f] % 2;
    for (int a = 0; a < NF; ++a) {
      int idx = 0, stride = 1, rem = a;
      for (int k = 0; k < dim; ++k) {
        int coord;
        if (k == dir) coord = side ? 2 : 0;
        else { coord = rem % 3; rem /= 3; }
        idx += coord * stride;
        stride *= 3;
      }
      const uint32_t node = hl.cnv[(size_t)cell * NV + idx];
      fnode[(size_t)f * NF + a] = node;
      by_node[node].push_back(((uint32_t)f << 4) | (uint32_t)a);
    }
    for (int q = 0; q < NQF; ++q) {
      double n[3] = {0, 0, 0}, jxw;
      if (d->normal) {
        for (int cc = 0; cc < dim; ++cc) n[cc] = d->normal[((size_t)f * NQF + q) * dim + cc];
        jxw = d->jxw[(size_t)f * NQF + q];
      } else {
        n[dir] = side ? 1.0 : -1.0;
        jxw = 1.0;
        int qq = q;
        for (int k = 0; k < dim; ++k) {
          if (k == dir) continue;
          jxw *= wq[qq % 3] * hT[(size_t)k * lv.cstride + cell];
          qq /= 3;
        }
      }
      for (int cc = 0; cc < dim; ++cc) njxw[((size_t)f * NQF + q) * dim + cc] = n[cc] * jxw;
    }
  }
  std::vector<uint32_t> nodes, ptr(1, 0), ent;
  for (auto &kv : by_node) {
    nodes.push_back(kv.first);
    ent.insert(ent.end(), kv.second.begin(), kv.second.end());
    ptr.push_back((uint32_t)ent.size());
  }
  FaceTerm &F = c->faces;
  CHECK(c->upload(&F.nodes, nodes)); CHECK(c->upload(&F.ptr, ptr)); CHECK(c->upload(&F.ent, ent));
  CHECK(c->upload(&F.fnode, fnode)); CHECK(c->upload(&F.stab, stab)); CHECK(c->upload(&F.njxw, njxw));
  F.n_faces = nf;
  F.n_nodes = (int)nodes.size();
  return 0;
}

int agmg_compute_diagonals(agmg_ctx *c) {
  if (!c) { set_error("null context"); return -1; }
  if (c->has_hanging()) { set_error("agmg_compute_diagonals: the context holds hanging-node constraints; only agmg_vmult resolves them yet (SURVEY.md 8 f1, first slice)"); return -2; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  for (auto &l : c->L) if (!l.set || !l.coeff.set) { set_error("compute_diagonals: every level needs tables and coefficients"); return -1; }
  CHECK(c->ensure_workspace());
  for (auto &l : c->L) {
    CHECK(launch_diag_cells(c->stream, l, l.coeff, c->slots_u, c->slots_p));
    EpiArgs eA, eM;
    eA.epi = eM.epi = EPI_DIAG;
    eA.dst = l.inv_diag_A;
    eM.dst = l.inv_diag_M;
    if (!c->dist()) {
      CHECK(launch_pull_v(c->stream, l, c->slots_u, eA));
      CHECK(launch_pull_p(c->stream, l, c->slots_p, 1, eM));
    } else {  // cell diagonals of interface nodes are summed over the ranks sharing them
      CHECK(launch_pull_v(c->stream, l, c->slots_u, eA, PH_GHOST));
      CHECK(launch_pull_p(c->stream, l, c->slots_p, 1, eM, PH_GHOST));
      std::vector<agmg_ctx::Msg> m;
      c->add_compress(m, l.xv, l.xv.nc);
      c->add_compress(m, l.xp, 1);
      CHECK(c->exchange(m));
      CHECK(launch_pull_v(c->stream, l, c->slots_u, eA, PH_OWNED));
      CHECK(launch_pull_p(c->stream, l, c->slots_p, 1, eM, PH_OWNED));
      // the one-round smoother sequences apply the update to the ghost copies too, so they read the diagonal there; they
      // run only where EVERY rank can (a rank without assembled rows, e.g. at an irregular vertex of its mesh, would
      // otherwise wait in another message pattern)
      if (c->ldist(l)) {
        CHECK(c->update_ghosts(l.xp, l.inv_diag_M));
        CHECK(c->update_ghosts(l.xv, l.inv_diag_A));
        double can[2] = {c->mass_sym_local(l, l.coeff) ? 1.0 : 0.0, c->a_sym_local(l, l.coeff) ? 1.0 : 0.0};
        CHECK(c->allreduce_host(can, 2));
        l.sym_agreed = can[0] == (double)c->opt.world_size;
        l.sym_agreed_A = can[1] == (double)c->opt.world_size;
      }
    }
  }
  CHECK(c->sync());
  c->diag_ready = true;
  c->smoothers_ready = false;
  return 0;
}

static int vmult_sizes(agmg_ctx *c, int which, int level, size_t *n) {
  if (level != AGMG_ACTIVE && (level < 0 || level >= c->nl())) { set_error("bad level"); return -1; }
  const DeviceLevel &lv = level == AGMG_ACTIVE ? c->act() : c->L[level];
  const size_t nu = lv.n_dofs_v(), np = lv.n_dofs_p();
  switch (which) {
    case AGMG_OP_STOKES: case AGMG_OP_BT: *n = nu + np; break;
    case AGMG_OP_A: *n = nu; break;
    case AGMG_OP_MASS: *n = np; break;
    default: set_error("bad operator"); return -1;
  }
  return 0;
}

// the operator application itself, on the context's current slot buffers (slots_u, slots_p)
static int vmult_device_impl(agmg_ctx *c, int which, int level, double *dst, const double *src) {
  if (level == AGMG_ACTIVE && c->ad.on) return c->active_apply(which, dst, src, nullptr);
  const int l = level == AGMG_ACTIVE ? c->nl() - 1 : level;
  const DeviceLevel &lv = c->L[l];
  const CoeffData &cd = level == AGMG_ACTIVE ? c->active_coeff() : lv.coeff;
  const size_t nu = lv.n_dofs_v(), np = lv.n_dofs_p();
  EpiArgs e;
  e.epi = EPI_STORE; e.dst = dst; e.x = src;
  switch (which) {
    case AGMG_OP_A: return c->apply_level(0, l, cd, src, e);
    case AGMG_OP_MASS: return c->apply_level(1, l, cd, src, e);
    case AGMG_OP_STOKES: {
      if (c->dist()) {
        if (l != c->nl() - 1 || &cd != &c->active_coeff()) { set_error("distributed Stokes vmult: active mesh only"); return -2; }
        return c->stokes_vmult(dst, src);
      }
      if (level == AGMG_ACTIVE) return c->stokes_vmult(dst, src);   // with the active operator's face terms
      EpiArgs ep = e;
      ep.dst = dst + nu; ep.x = src + nu;
      return c->op_velocity(lv, cd, AGMG_OP_STOKES, src, src + nu, e, &ep);
    }
    default: {  // B^T: both blocks of dst are written (vmult zeroes dst first)
      if (c->dist()) {
        if (l != c->nl() - 1 || &cd != &c->active_coeff()) { set_error("distributed B^T vmult: active mesh only"); return -2; }
        CHECK(c->bt_residual(dst, nullptr, src + nu));
      } else {
        CHECK(c->op_velocity(lv, cd, AGMG_OP_BT, nullptr, src + nu, e));
      }
      return vec_set(c->stream, dst + nu, 0.0, np);
    }
  }
}

int agmg_vmult_device(agmg_ctx *c, int which, int level, double *dst, const double *src) {
  if (!c) { set_error("null context"); return -1; }
  if (c->has_hanging()) { set_error("agmg_vmult_device: the context holds hanging-node constraints; only agmg_vmult resolves them yet (SURVEY.md 8 f1, first slice)"); return -2; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  size_t n;
  CHECK(vmult_sizes(c, which, level, &n));
  CHECK(c->ensure_workspace());
  return vmult_device_impl(c, which, level, dst, src);
}

// agmg_vmult_device for callers that may not allocate (HIP graph capture, serving loops): the slot
// buffers come out of this GPU's static scratch (scratch.hip) instead of the context's workspace.
// No hipMalloc / hipFree / allocating routine on this path.  The scratch keeps the previous call's
// contents: every kernel on this path writes a slot before any kernel reads it (nothing has to start
// at zero).  Calls that overlap on one GPU share the array and race.
int agmg_vmult_device_noalloc(agmg_ctx *c, int which, int level, double *dst, const double *src) {
  if (!c) { set_error("null context"); return -1; }
  if (c->has_hanging()) { set_error("agmg_vmult_device_noalloc: the context holds hanging-node constraints; only agmg_vmult resolves them yet (SURVEY.md 8 f1, first slice)"); return -2; }
  if (!dst || !src) { set_error("null vector"); return -1; }
  if (c->dist()) { set_error("agmg_vmult_device_noalloc: one-rank contexts only (the ghost exchange stages through buffers it allocates)"); return -2; }
  if (c->ad.on) { set_error("agmg_vmult_device_noalloc: not in adaptive contexts (the resolved source vector lives in a buffer the context allocates)"); return -2; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  size_t n;
  CHECK(vmult_sizes(c, which, level, &n));
  double *base = static_cast<double *>(static_scratch_ptr(c->opt.device));
  if (!base) { set_error("static scratch not initialised on this device"); return -100; }
  const int l = level == AGMG_ACTIVE ? c->nl() - 1 : level;
  const DeviceLevel &lv = c->L[l];
  if (!lv.set) { set_error("level not set"); return -1; }
  const int dim = lv.dim, NV = ipow(3, dim), NP = ipow(2, dim);
  // doubles the kernels of this operator write to the slot buffers, by the predicates op_velocity and launch_op
  // choose the path with: brick partial sums, block-boundary slots, or one slot per (cell, local dof)
  const CoeffData &cd = level == AGMG_ACTIVE ? c->active_coeff() : lv.coeff;
  size_t su = 0, sp = 0;
  if (which != AGMG_OP_MASS) {
    if (c->bricks_on(lv) && brickq_supported(lv, cd, which)) su = lv.bk.n_partial_slots * 3;
    else if (c->use_blocks && lv.bt.built && !((cd.newton || cd.dilation) && which == AGMG_OP_STOKES)) su = lv.bt.n_shared_slots * (size_t)dim;
    else su = (size_t)NV * lv.cstride * dim;
  }
  if (which == AGMG_OP_STOKES || which == AGMG_OP_MASS) sp = (size_t)NP * lv.cstride;   // one value per (cell, pressure node)
  su = (su + 31) / 32 * 32;   // keep the second buffer 256-byte aligned
  const size_t need = (su + sp) * sizeof(double);
  if (need > STATIC_SCRATCH_BYTES) {
    set_error("agmg_vmult_device_noalloc: this level needs " + std::to_string(need) + " bytes of scratch, the static scratch holds " +
              std::to_string(STATIC_SCRATCH_BYTES) + " (use agmg_vmult_device)");
    return -3;
  }
  double *const save_u = c->slots_u, *const save_p = c->slots_p;
  c->slots_u = base;
  c->slots_p = base + su;
  const int rc = vmult_device_impl(c, which, level, dst, src);
  c->slots_u = save_u;
  c->slots_p = save_p;
  return rc;
}

// Hanging-node constraints around a device operator that treats every node as its own DoF (what MatrixFree's
// read_dof_values / distribute_local_to_global do with AffineConstraints; constrained rows are identity rows):
// y = C^T A C x.  Masters with a homogeneous Dirichlet bit contribute zero and receive nothing.  Host vectors.
static int vmult_hanging_host(agmg_ctx *c, int which, int level, double *dst, const double *src, size_t n) {
  const int l = level == AGMG_ACTIVE ? c->nl() - 1 : level;
  const HostLevel &hl = c->H[l];
  const DeviceLevel &lv = c->L[l];
  const int dim = lv.dim;
  const bool has_u = which != AGMG_OP_MASS, has_p = which != AGMG_OP_A;
  const size_t poff = has_u ? lv.n_dofs_v() : 0;
  std::vector<double> xs(src, src + n), ys(n);
  auto con = [&](int sp, uint32_t node, int comp) { return sp == 0 ? ((hl.cmask[0][node] >> comp) & 1) != 0 : (hl.cmask[1][node] & 1) != 0; };
  auto resolve = [&](int sp, size_t off, int nc) {
    for (size_t i = 0; i < hl.hang_n[sp].size(); ++i)
      for (int comp = 0; comp < nc; ++comp) {
        double v = 0;
        for (int k = hl.hang_ptr[sp][i]; k < hl.hang_ptr[sp][i + 1]; ++k)
          if (!con(sp, hl.hang_m[sp][k], comp)) v += hl.hang_w[sp][k] * src[off + (size_t)hl.hang_m[sp][k] * nc + comp];
        xs[off + (size_t)hl.hang_n[sp][i] * nc + comp] = v;
      }
  };
  auto distribute = [&](int sp, size_t off, int nc) {
    for (size_t i = 0; i < hl.hang_n[sp].size(); ++i) {
      const uint32_t h = hl.hang_n[sp][i];
      for (int comp = 0; comp < nc; ++comp) {
        if (!con(sp, h, comp))
          for (int k = hl.hang_ptr[sp][i]; k < hl.hang_ptr[sp][i + 1]; ++k)
            if (!con(sp, hl.hang_m[sp][k], comp)) ys[off + (size_t)hl.hang_m[sp][k] * nc + comp] += hl.hang_w[sp][k] * ys[off + (size_t)h * nc + comp];
        ys[off + (size_t)h * nc + comp] = src[off + (size_t)h * nc + comp];
      }
    }
  };
  if (has_u) resolve(0, 0, dim);
  if (has_p) resolve(1, poff, 1);
  double *ds, *dd;
  CHECK(c->dalloc(&ds, n)); CHECK(c->dalloc(&dd, n));
  int rc = c->h2d(ds, xs.data(), n);
  if (!rc) rc = c->ensure_workspace();
  if (!rc) rc = vmult_device_impl(c, which, level, dd, ds);
  if (!rc) rc = c->d2h(ys.data(), dd, n);
  c->dfree(ds); c->dfree(dd);
  if (rc) return rc;
  if (has_u) distribute(0, 0, dim);
  if (which == AGMG_OP_MASS || which == AGMG_OP_STOKES) distribute(1, poff, 1);   // (B^T writes no pressure rows)
  std::copy(ys.begin(), ys.end(), dst);
  return 0;
}

int agmg_vmult(agmg_ctx *c, int which, int level, double *dst, const double *src) {
  if (!c) { set_error("null context"); return -1; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  size_t n;
  CHECK(vmult_sizes(c, which, level, &n));
  if (c->has_hanging()) return vmult_hanging_host(c, which, level, dst, src, n);
  double *ds, *dd;
  CHECK(c->dalloc(&ds, n)); CHECK(c->dalloc(&dd, n));
  int rc = c->h2d(ds, src, n);
  if (!rc) rc = agmg_vmult_device(c, which, level, dd, ds);
  if (!rc) rc = c->d2h(dst, dd, n);
  c->dfree(ds); c->dfree(dd);
  return rc;
}

// correct_stokes_rhs (local_smoothing.cc:761-898): rhs += -M u0, u0 read without the constraints
int agmg_rhs_correction(agmg_ctx *c, const double *u0_h, double *rhs_h) {
  if (!c || !u0_h || !rhs_h) { set_error("null argument"); return -1; }
  if (c->has_hanging()) { set_error("agmg_rhs_correction: the context holds hanging-node constraints; only agmg_vmult resolves them yet (SURVEY.md 8 f1, first slice)"); return -2; }
  if (c->ad.on) { set_error("agmg_rhs_correction: not built for adaptive contexts (inhomogeneous constraints next to hanging nodes; SURVEY.md 8 f1)"); return -2; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  const DeviceLevel &lv = c->fin();
  if (!lv.set || !c->active_coeff().set) { set_error("rhs_correction: active mesh tables / coefficients not set"); return -1; }
  CHECK(c->ensure_workspace());
  const size_t nu = lv.n_dofs_v(), np = lv.n_dofs_p(), n = nu + np;
  double *u0, *rhs;
  CHECK(c->get_vec(5, &u0)); CHECK(c->get_vec(6, &rhs));
  CHECK(c->h2d(u0, u0_h, n));
  CHECK(c->h2d(rhs, rhs_h, n));
  // no Newton terms, no dilation LHS term; the deviatoric part stays (:835-838)
  CoeffData cd = c->active_coeff();
  cd.is_compressible = (cd.is_compressible || cd.dilation) ? 1 : 0;
  cd.newton = false; cd.dilation = false;
  EpiArgs eu, ep;
  eu.epi = ep.epi = EPI_SUB;
  eu.dst = rhs; ep.dst = rhs + nu;
  hipStream_t st = c->stream;
  if (!c->dist()) {
    CHECK(launch_op_cells(st, lv, cd, AGMG_OP_STOKES, u0, u0 + nu, c->slots_u, c->slots_p, true));
    CHECK(launch_pull_v(st, lv, c->slots_u, eu));
    CHECK(launch_pull_p(st, lv, c->slots_p, 1, ep));
    CHECK(launch_face_term(st, lv.dim, c->faces, lv.cmask_v, u0, rhs, -1.0, nullptr, 0, true));
  } else {
    std::vector<agmg_ctx::Msg> m;
    CHECK(c->add_update(m, lv.xv, u0, lv.xv.nc));
    CHECK(c->add_update(m, lv.xp, u0 + nu, 1));
    CHECK(c->exchange(m));
    CHECK(launch_op_cells(st, lv, cd, AGMG_OP_STOKES, u0, u0 + nu, c->slots_u, c->slots_p, true));
    CHECK(launch_pull_v(st, lv, c->slots_u, eu, PH_GHOST));
    CHECK(launch_pull_p(st, lv, c->slots_p, 1, ep, PH_GHOST));
    m.clear();
    c->add_compress(m, lv.xv, lv.xv.nc);
    c->add_compress(m, lv.xp, 1);
    CHECK(c->exchange(m));
    CHECK(launch_pull_v(st, lv, c->slots_u, eu, PH_OWNED));
    CHECK(launch_pull_p(st, lv, c->slots_p, 1, ep, PH_OWNED));
    if (c->faces.global_active) {
      if (lv.xv.n_ghost) AGMG_HIP_CHECK(hipMemsetAsync(lv.xv.ghostbuf, 0, (size_t)lv.xv.n_ghost * lv.xv.nc * sizeof(double), st));
      CHECK(launch_face_term(st, lv.dim, c->faces, lv.cmask_v, u0, rhs, -1.0, lv.xv.ghostbuf, lv.n_owned_v, true));
      CHECK(c->compress(lv.xv));
      CHECK(launch_face_add_received(st, lv.dim, c->faces, lv.cmask_v, lv.xv.expbuf, rhs, -1.0));
    }
  }
  return c->d2h(rhs_h, rhs, n);
}

int agmg_get_inverse_diagonal(agmg_ctx *c, int which, int level, double *dst) {
  if (!c || level < 0 || level >= c->nl()) { set_error("bad level"); return -1; }
  if (c->has_hanging()) { set_error("agmg_get_inverse_diagonal: the context holds hanging-node constraints; only agmg_vmult resolves them yet (SURVEY.md 8 f1, first slice)"); return -2; }
  if (!c->diag_ready) { set_error("diagonals not computed"); return -1; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  const DeviceLevel &lv = c->L[level];
  if (which == AGMG_OP_A) return c->d2h(dst, lv.inv_diag_A, lv.n_dofs_v());
  if (which == AGMG_OP_MASS) return c->d2h(dst, lv.inv_diag_M, lv.n_dofs_p());
  set_error("diagonal exists for A and MASS only (StokesOperator::compute_diagonal is ExcNotImplemented)");
  return -1;
}

static int transfer_host(agmg_ctx *c, int which, int fine_level, bool prolong, double *dst, const double *src) {
  if (!c || fine_level < 1 || fine_level >= c->nl()) { set_error("bad level for transfer"); return -1; }
  if (which != AGMG_OP_A && which != AGMG_OP_MASS) { set_error("transfer: which must be A or MASS"); return -1; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  CHECK(c->ensure_workspace());
  CHECK(c->ensure_transfers());
  const bool vel = which == AGMG_OP_A;
  const DeviceLevel &F = c->L[fine_level], &C = c->L[fine_level - 1];
  const size_t nf = vel ? F.n_dofs_v() : F.n_dofs_p(), ncz = vel ? C.n_dofs_v() : C.n_dofs_p();
  const size_t nd = prolong ? nf : ncz, ns = prolong ? ncz : nf;
  double *ds, *dd;
  CHECK(c->dalloc(&ds, ns)); CHECK(c->dalloc(&dd, nd));
  int rc = c->h2d(ds, src, ns);
  if (!rc) rc = c->h2d(dd, dst, nd);
  if (!rc) {
    if (prolong) rc = c->prolongate_from(vel ? 0 : 1, fine_level, dd, ds);
    else if (c->dist() && C.replicated && !F.replicated) {
      // onto the last replicated level: the restriction stores the sum over ranks (constrained rows: zero), added here
      double *tmp;
      CHECK(c->dalloc(&tmp, ncz));
      EpiArgs e;
      e.epi = EPI_STORE; e.dst = tmp;
      rc = c->restrict_to(vel ? 0 : 1, fine_level, ds, e);
      if (!rc) rc = vec_axpy(c->stream, dd, 1.0, tmp, ncz);
      if (!rc) rc = c->sync();
      c->dfree(tmp);
    } else {
      EpiArgs e;
      e.epi = EPI_ADD; e.dst = dd;
      rc = c->restrict_to(vel ? 0 : 1, fine_level, ds, e);
    }
  }
  if (!rc) rc = c->d2h(dst, dd, nd);
  c->dfree(ds); c->dfree(dd);
  return rc;
}
int agmg_prolongate_add(agmg_ctx *c, int which, int to_level, double *dst, const double *src) {
  return transfer_host(c, which, to_level, true, dst, src);
}
int agmg_restrict_add(agmg_ctx *c, int which, int from_level, double *dst, const double *src) {
  return transfer_host(c, which, from_level, false, dst, src);
}

int agmg_setup_smoothers(agmg_ctx *c) {
  if (!c) { set_error("null context"); return -1; }
  if (c->has_hanging()) { set_error("agmg_setup_smoothers: the context holds hanging-node constraints; only agmg_vmult resolves them yet (SURVEY.md 8 f1, first slice)"); return -2; }
  if (c->has_hanging()) { set_error("agmg_restrict_add: the context holds hanging-node constraints; only agmg_vmult resolves them yet (SURVEY.md 8 f1, first slice)"); return -2; }
  if (c->has_hanging()) { set_error("agmg_prolongate_add: the context holds hanging-node constraints; only agmg_vmult resolves them yet (SURVEY.md 8 f1, first slice)"); return -2; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  return c->setup_smoothers();
}

int agmg_get_eigenvalue_estimates(agmg_ctx *c, int which, int level, double *lmin, double *lmax) {
  if (!c || level < 0 || level >= c->nl() || !c->smoothers_ready) { set_error("smoothers not set up"); return -1; }
  if (c->has_hanging()) { set_error("agmg_get_eigenvalue_estimates: the context holds hanging-node constraints; only agmg_vmult resolves them yet (SURVEY.md 8 f1, first slice)"); return -2; }
  const Smoother &s = c->sm[which == AGMG_OP_A ? 0 : 1][level];
  *lmin = s.lmin; *lmax = s.lmax;
  return 0;
}

int agmg_vcycle(agmg_ctx *c, int which, double *dst, const double *src) {
  if (!c) { set_error("null context"); return -1; }
  if (c->has_hanging()) { set_error("agmg_vcycle: the context holds hanging-node constraints; only agmg_vmult resolves them yet (SURVEY.md 8 f1, first slice)"); return -2; }
  if (which != AGMG_OP_A && which != AGMG_OP_MASS) { set_error("vcycle: which must be A or MASS"); return -1; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  if (!c->smoothers_ready) CHECK(c->setup_smoothers());
  const int w = which == AGMG_OP_A ? 0 : 1;
  const size_t n = w == 0 ? (size_t)c->act().n_dofs_v() : (size_t)c->act().n_dofs_p();
  double *ds, *dd;
  CHECK(c->dalloc(&ds, n)); CHECK(c->dalloc(&dd, n));
  int rc = c->h2d(ds, src, n);
  if (!rc) rc = c->vcycle(w, dd, ds);
  if (!rc) rc = c->d2h(dst, dd, n);
  c->dfree(ds); c->dfree(dd);
  return rc;
}

int agmg_precondition(agmg_ctx *c, int expensive, double *dst, const double *src) {
  if (!c) { set_error("null context"); return -1; }
  if (c->has_hanging()) { set_error("agmg_precondition: the context holds hanging-node constraints; only agmg_vmult resolves them yet (SURVEY.md 8 f1, first slice)"); return -2; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  if (!c->smoothers_ready) CHECK(c->setup_smoothers());
  const size_t n = (size_t)c->act().n_dofs_v() + c->act().n_dofs_p();
  double *ds, *dd;
  CHECK(c->dalloc(&ds, n)); CHECK(c->dalloc(&dd, n));
  int status = 0;
  int rc = c->h2d(ds, src, n);
  if (!rc) rc = c->precondition(expensive != 0, dd, ds, &status);
  if (!rc) rc = c->d2h(dst, dd, n);
  c->dfree(ds); c->dfree(dd);
  if (!rc && status) { set_error("inner CG of the expensive preconditioner did not converge"); return 1; }
  return rc;
}

int agmg_solve(agmg_ctx *c, const double *rhs_h, double *x_h, int newton, agmg_solve_report *rep) {
  if (!c || !rhs_h || !x_h || !rep) { set_error("null argument"); return -1; }
  if (c->has_hanging()) { set_error("agmg_solve: the context holds hanging-node constraints; only agmg_vmult resolves them yet (SURVEY.md 8 f1, first slice)"); return -2; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  std::memset(rep, 0, sizeof(*rep));
  rep->initial_nonlinear_residual = rep->final_linear_residual = std::numeric_limits<double>::quiet_NaN();
  const double t_begin = now_s();
  if (!c->smoothers_ready) CHECK(c->setup_smoothers());
  CHECK(c->sync());
  const double t_setup = now_s();
  rep->setup_seconds = t_setup - t_begin;
  const DeviceLevel &lv = c->act();
  const size_t nu = lv.n_dofs_v(), np = lv.n_dofs_p(), n = nu + np;
  const agmg_ctx::Span sp = c->span_up();
  hipStream_t st = c->stream;
  CHECK(c->ensure_workspace());
  // pool layout: 0..4 preconditioner scratch, 5 rhs, 6 x, 7 tmp, 8 x0, 9.. Krylov
  double *rhs, *x, *tmp, *x0;
  CHECK(c->get_vec(5, &rhs)); CHECK(c->get_vec(6, &x)); CHECK(c->get_vec(7, &tmp)); CHECK(c->get_vec(8, &x0));
  CHECK(c->h2d(rhs, rhs_h, n));
  if (c->ad.on) {   // (hanging nodes are constrained entries too: zero in the start vector, identity rows in the operator)
    std::vector<double> xs(x_h, x_h + n);
    for (uint32_t h : c->ad.hact.hang_n[0]) for (int k = 0; k < lv.dim; ++k) xs[(size_t)h * lv.dim + k] = 0.0;
    for (uint32_t h : c->ad.hact.hang_n[1]) xs[nu + h] = 0.0;
    CHECK(c->h2d(x, xs.data(), n));
  } else
  CHECK(c->h2d(x, x_h, n));
  // current_constraints.set_zero(initial guess), local_smoothing.cc:1116
  CHECK(vec_zero_constrained(st, x, lv.cmask_v, lv.n_nodes_v, lv.dim));
  double tol;
  if (!newton) {  // local_smoothing.cc:1132-1167
    CHECK(c->stokes_residual(tmp, rhs, x));
    double r0;
    CHECK(c->norm(tmp, sp, &r0));
    rep->initial_nonlinear_residual = r0;
    CHECK(vec_copy(st, x0, x, n));
    CHECK(vec_set(st, x0, 0.0, nu));
    CHECK(c->stokes_residual(tmp, rhs, x0));
    double d2[2];
    CHECK(c->dots(tmp, {tmp}, c->span_act(0), &d2[0]));
    CHECK(c->dots(rhs + nu, {rhs + nu}, c->span_act(1), &d2[1]));
    tol = c->opt.linear_tolerance * std::sqrt(d2[0] + d2[1]);
  } else {  // :1169-1183
    double r0;
    CHECK(c->norm(rhs, sp, &r0));
    tol = c->opt.linear_tolerance * r0;
    rep->initial_nonlinear_residual = r0;
  }
  rep->solver_tolerance = tol;
  int inner_status = 0;
  int nA_c = 0, nS_c = 0, nA_e = 0, nS_e = 0;
  Op Aop = [&](double *d, const double *s) { return c->stokes_vmult(d, s); };
  Op Pc = [&](double *d, const double *s) { return c->precondition(false, d, s, nullptr); };
  Op Pe = [&](double *d, const double *s) { return c->precondition(true, d, s, &inner_status); };
  Control cc{c->opt.n_cheap_steps, tol}, ce{c->opt.n_expensive_steps, tol};
  int status = 0, state = ST_FAILURE;
  const size_t pool0 = 9;
  c->nA = c->nS = 0;
  if (c->opt.n_cheap_steps > 0) {
    if (c->opt.krylov_type == AGMG_KRYLOV_GMRES) CHECK(solver_gmres(c, Aop, Pc, x, rhs, sp, c->opt.gmres_restart, cc, pool0, &state));
    else CHECK(solver_idr(c, Aop, Pc, x, rhs, sp, c->opt.idr_s, cc, pool0, &state));
  }
  nA_c = c->nA; nS_c = c->nS;
  rep->n_cheap_iterations = cc.last_step;
  rep->final_linear_residual = cc.last_value;
  if (state != ST_SUCCESS) {
    if (c->opt.n_expensive_steps == 0) status = 1;
    else {
      c->nA = c->nS = 0;
      CHECK(solver_fgmres(c, Aop, Pe, x, rhs, sp, c->opt.gmres_restart, ce, pool0, &state));
      nA_e = c->nA; nS_e = c->nS;
      rep->n_expensive_iterations = ce.last_step;
      rep->final_linear_residual = ce.last_value;
      if (state != ST_SUCCESS || inner_status) status = 1;
    }
  }
  rep->n_A_preconditioner_cheap = nA_c; rep->n_S_preconditioner_cheap = nS_c;
  rep->n_A_preconditioner_expensive = nA_e; rep->n_S_preconditioner_expensive = nS_e;
  for (double v : cc.history) if (rep->n_history < 2048) rep->history[rep->n_history++] = v;
  for (double v : ce.history) if (rep->n_history < 2048) rep->history[rep->n_history++] = v;
  CHECK(c->d2h(x_h, x, n));
  if (c->ad.on) {   // current_constraints.distribute(solution): hanging nodes take the values of their constraints
    const HostLevel &hl = c->ad.hact;
    for (int spc = 0; spc < 2; ++spc) {
      const int nc = spc == 0 ? lv.dim : 1;
      const size_t off = spc == 0 ? 0 : nu;
      for (size_t i = 0; i < hl.hang_n[spc].size(); ++i)
        for (int k = 0; k < nc; ++k) {
          double v = 0;
          for (int q = hl.hang_ptr[spc][i]; q < hl.hang_ptr[spc][i + 1]; ++q) v += hl.hang_w[spc][q] * x_h[off + (size_t)hl.hang_m[spc][q] * nc + k];
          x_h[off + (size_t)hl.hang_n[spc][i] * nc + k] = v;
        }
    }
  }
  rep->solve_seconds = now_s() - t_setup;
  if (status == 1) set_error("Stokes solver did not converge within the allowed steps");
  return status;
}

int agmg_bench_vmult(agmg_ctx *c, int which, int level, int n_repeat, double *seconds_per_vmult) {
  if (!c || n_repeat < 1 || !seconds_per_vmult) { set_error("bad argument"); return -1; }
  if (c->has_hanging()) { set_error("agmg_bench_vmult: the context holds hanging-node constraints; only agmg_vmult resolves them yet (SURVEY.md 8 f1, first slice)"); return -2; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  size_t n;
  CHECK(vmult_sizes(c, which, level, &n));
  const DeviceLevel &lv = level == AGMG_ACTIVE ? c->act() : c->L[level];
  double *src, *dst;
  CHECK(c->dalloc(&src, n)); CHECK(c->dalloc(&dst, n));
  {
    std::vector<double> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = std::sin(0.37 * (double)i);
    CHECK(c->h2d(src, h.data(), n));
    if (which != AGMG_OP_MASS) CHECK(vec_zero_constrained(c->stream, src, lv.cmask_v, lv.n_nodes_v, lv.dim));
  }
  int rc = agmg_vmult_device(c, which, level, dst, src);  // warm-up
  hipEvent_t e0, e1;
  AGMG_HIP_CHECK(hipEventCreate(&e0));
  AGMG_HIP_CHECK(hipEventCreate(&e1));
  if (!rc) {
    AGMG_HIP_CHECK(hipEventRecord(e0, c->stream));
    for (int i = 0; i < n_repeat && !rc; ++i) rc = agmg_vmult_device(c, which, level, dst, src);
    AGMG_HIP_CHECK(hipEventRecord(e1, c->stream));
    AGMG_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    AGMG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *seconds_per_vmult = 1e-3 * (double)ms / n_repeat;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  c->dfree(src); c->dfree(dst);
  return rc;
}

// The in-solver micro-benchmark of the reference ("Execute solver timings", local_smoothing.cc:1234-1375):
// device-resident, one warm-up call then the mean wall time of n_repeat calls per section, the
// stream drained on both sides.  seconds[0..5] = stokes_vmult, stokes_preconditioner (cheap),
// A_preconditioner (one V-cycle), S_preconditioner (one V-cycle), A_block_vmult, setup of the
// smoothers (eigenvalue estimates of every level, what the reference redoes in every solve()).
int agmg_execute_solver_timings(agmg_ctx *c, int n_repeat, double seconds[6]) {
  if (!c || !seconds || n_repeat < 1) { set_error("bad argument"); return -1; }
  if (c->has_hanging()) { set_error("agmg_execute_solver_timings: the context holds hanging-node constraints; only agmg_vmult resolves them yet (SURVEY.md 8 f1, first slice)"); return -2; }
  AGMG_HIP_CHECK(hipSetDevice(c->opt.device));
  {
    CHECK(c->sync());
    const double t0 = now_s();
    CHECK(c->setup_smoothers());
    CHECK(c->sync());
    seconds[5] = now_s() - t0;
  }
  const DeviceLevel &lv = c->act();
  const size_t nu = lv.n_dofs_v(), np = lv.n_dofs_p(), n = nu + np;
  double *src, *dst;
  CHECK(c->get_vec(5, &src)); CHECK(c->get_vec(6, &dst));
  {
    std::vector<double> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = std::sin(0.37 * (double)i);
    CHECK(c->h2d(src, h.data(), n));
    CHECK(vec_zero_constrained(c->stream, src, lv.cmask_v, lv.n_nodes_v, lv.dim));
  }
  auto timeit = [&](const std::function<int()> &f, double *out) -> int {
    CHECK(f());
    CHECK(c->sync());
    const double t0 = now_s();
    for (int i = 0; i < n_repeat; ++i) CHECK(f());
    CHECK(c->sync());
    *out = (now_s() - t0) / n_repeat;
    return 0;
  };
  EpiArgs e;
  e.epi = EPI_STORE; e.dst = dst; e.x = src;
  CHECK(timeit([&] { return c->stokes_vmult(dst, src); }, &seconds[0]));
  CHECK(timeit([&] { return c->precondition(false, dst, src, nullptr); }, &seconds[1]));
  CHECK(timeit([&] { return c->vcycle(0, dst, src); }, &seconds[2]));
  CHECK(timeit([&] { return c->vcycle(1, dst + nu, src + nu); }, &seconds[3]));
  CHECK(timeit([&] { return c->ad.on ? c->active_apply(AGMG_OP_A, dst, src, nullptr) : c->apply_level(0, c->nl() - 1, c->active_coeff(), src, e); }, &seconds[4]));
  return 0;
}

int agmg_get_level_kernel_path(const agmg_ctx *c, int level, int32_t *n_bricks, int32_t *layers_per_brick, int32_t *n_blocks) {
  if (!c || (level != AGMG_ACTIVE && (level < 0 || level >= (int)c->L.size()))) { set_error("bad level"); return -1; }
  const DeviceLevel &lv = level == AGMG_ACTIVE ? c->act() : c->L[level];
  const bool bricks = c->bricks_on(lv);
  if (n_bricks) *n_bricks = bricks ? lv.bk.n_bricks : 0;
  if (layers_per_brick) *layers_per_brick = bricks ? lv.bk.Z : 0;
  if (n_blocks) *n_blocks = (c->use_blocks && lv.bt.built) ? lv.bt.n_blocks : 0;
  return 0;
}

int agmg_get_overlap_statistics(const agmg_ctx *c, int level, int32_t *n_interior_blocks, int32_t *n_interface_blocks,
                                int32_t *n_overlapped_applications) {
  if (!c || level < 0 || level >= (int)c->L.size()) { set_error("bad level"); return -1; }
  const BlockTables &bt = c->L[level].bt;
  const bool split = bt.built && bt.blk_order;
  if (n_interior_blocks) *n_interior_blocks = split ? bt.n_interior : 0;
  if (n_interface_blocks) *n_interface_blocks = split ? bt.n_blocks - bt.n_interior : 0;
  if (n_overlapped_applications) *n_overlapped_applications = c->n_overlapped;
  return 0;
}

size_t agmg_get_cell_data_memory_consumption(const agmg_ctx *c) {
  size_t s = c->active.bytes;
  for (auto &l : c->L) s += l.coeff.bytes;
  return s;
}
size_t agmg_get_mg_transfer_memory_consumption(const agmg_ctx *c) {
  size_t s = 0;
  for (auto &l : c->L) s += l.bytes_transfer;
  return s;
}
size_t agmg_get_constraint_memory_consumption(const agmg_ctx *c) {
  size_t s = 0;
  for (auto &l : c->L) s += l.bytes_constraints;
  return s;
}
size_t agmg_get_dof_handler_memory_consumption(const agmg_ctx *c) {
  size_t s = 0;
  for (auto &l : c->L) s += l.bytes_tables;
  return s;
}

}  // extern "C"
