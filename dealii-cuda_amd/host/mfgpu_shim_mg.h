// Multigrid side of the shim: the classes poisson_mg.cu / bmop_mg.cu use around the level operators
// and the level transfer, on top of the C-ABI.  From the reference tree: MGTransferMatrixFreeGpu
// (matrix_free_gpu/mg_transfer_matrix_free_gpu.h:140-252), LaplaceOperatorGpu::reinit(dof_handler,
// mg_constrained_dofs, level) (laplace_operator_gpu.h:154-186).  From deal.II (library code the reference links
// against, restated here in the small form its callers need): MGLevelObject, PreconditionChebyshev,
// Multigrid (V-cycle), PreconditionMG, mg::Matrix.  Globally refined level hierarchies only (CUBE and BALL domains): the
// interface ("edge") matrices of local smoothing on adaptive meshes (laplace_operator_gpu.h:306-352) are identity
// operations here and are not built.
#ifndef MFGPU_SHIM_MG_H
#define MFGPU_SHIM_MG_H

#include <cmath>
#include <memory>

#include "mfgpu_shim.h"

namespace mfgpu_shim {

// deal.II MGConstrainedDoFs: the level Dirichlet sets come with the level meshes of the stand-in
class MGConstrainedDoFs {
public:
  template <typename DH>
  void initialize(const DH &) {}
  template <typename DH, typename S>
  void make_zero_boundary_constraints(const DH &, const S &) {}
};

template <typename T>
class MGLevelObject {
public:
  void resize(unsigned int lo, unsigned int hi) {
    minlevel = lo;
    objects.clear();
    for (unsigned int l = lo; l <= hi; ++l) objects.push_back(std::make_shared<T>());
  }
  void clear_elements() { objects.clear(); }
  T &operator[](unsigned int l) { return *objects[l - minlevel]; }
  const T &operator[](unsigned int l) const { return *objects[l - minlevel]; }
  unsigned int min_level() const { return minlevel; }
  unsigned int max_level() const { return minlevel + (unsigned int)objects.size() - 1; }

private:
  unsigned int minlevel = 0;
  std::vector<std::shared_ptr<T>> objects;
};

// level hierarchy of a stand-in mesh (DoFHandler::distribute_mg_dofs, poisson_mg.cu:152).  Globally refined CUBE /
// BALL: level l = the domain refined l times, the finest level is the active mesh.  ADAPTIVE_GRID: the
// pseudo-adaptive recipe with Triangulation::limit_level_difference_at_vertices (poisson_mg.cu:131), level l = all
// cells of level l (mfgpu_mg_hierarchy), with refinement edges, transfers over the refined parents and copy pairs.
template <int dim>
class MGDoFHandler {
public:
  explicit MGDoFHandler(const Triangulation<dim> &tria) : tria(&tria) {}
  ~MGDoFHandler() { clear(); }
  MGDoFHandler(const MGDoFHandler &) = delete;
  void clear() {
    if (hierarchy) {
      mfgpu_mg_hierarchy_destroy(hierarchy);  // owns its level meshes
      hierarchy = nullptr;
      mfgpu_mesh_destroy(active);
    } else {
      for (mfgpu_mesh *m : levels) mfgpu_mesh_destroy(m);
    }
    active = nullptr;
    levels.clear();
  }
  void distribute_mg_dofs(const FE_Q<dim> &fe, int number_type = MFGPU_F64) {
    clear();
    degree = fe.degree;
    if (tria->adaptive) {
      if (tria->domain != CUBE) throw std::runtime_error("multigrid on adaptive meshes: CUBE domain only");
      check(mfgpu_mesh_create_adaptive_mg(dim, (int)fe.degree, tria->n_ref, number_type, &active), "active mesh");
      check(mfgpu_mg_hierarchy_create(active, &hierarchy), "level hierarchy");
      for (int l = 0; l < mfgpu_mg_n_levels(hierarchy); ++l)
        levels.push_back(const_cast<mfgpu_mesh *>(mfgpu_mg_level_mesh(hierarchy, l)));
      return;
    }
    for (int l = 0; l <= tria->n_ref; ++l) {
      mfgpu_mesh *m = nullptr;
      if (tria->domain == BALL) {
        check(mfgpu_mesh_create_ball(dim, (int)fe.degree, l, number_type, &m), "level mesh");
      } else {
        uint32_t nper[3] = {1u << l, 1u << l, 1u << l};
        check(mfgpu_mesh_create_uniform(dim, (int)fe.degree, nper, -1.0, 1.0, 0, 0, number_type, &m), "level mesh");
      }
      levels.push_back(m);
    }
    active = levels.back();
  }
  bool is_adaptive() const { return hierarchy != nullptr; }
  unsigned int n_levels() const { return (unsigned int)levels.size(); }
  unsigned int n_dofs(unsigned int level) const {
    mfgpu_desc d;
    check(mfgpu_mesh_desc(levels[level], &d), "mesh desc");
    return d.n_dofs;
  }
  unsigned int n_dofs() const {
    mfgpu_desc d;
    check(mfgpu_mesh_desc(active, &d), "mesh desc");
    return d.n_dofs;
  }
  // MGConstrainedDoFs::get_refinement_edge_indices(level)
  std::vector<uint32_t> refinement_edge_indices(unsigned int level) const {
    if (!hierarchy) return {};
    const uint32_t *p = nullptr;
    const int64_t n = mfgpu_mg_edge_dofs(hierarchy, (int)level, &p);
    return std::vector<uint32_t>(p, p + (n > 0 ? n : 0));
  }
  const Triangulation<dim> *tria;
  std::vector<mfgpu_mesh *> levels;
  mfgpu_mesh *active = nullptr;  // the active mesh (== levels.back() on globally refined meshes)
  mfgpu_mg_hierarchy *hierarchy = nullptr;
  unsigned int degree = 0;
};

// level-local operator (laplace_operator_gpu.h:154-186): the operator of the level mesh; the level's Dirichlet rows
// are identity rows as on the active mesh (constraint_handler.reinit(mg_constrained_dofs, level))
template <int dim, int fe_degree, typename Number>
class LevelOperatorGpu {
public:
  typedef Number value_type;
  typedef GpuVector<Number> VectorType;
  ~LevelOperatorGpu() { clear(); }
  void clear() {
    mfgpu_level_destroy(lev);
    mfgpu_destroy(own);
    lev = nullptr;
    own = nullptr;
    handle = nullptr;
  }
  // the level's refinement-edge dofs (MGConstrainedDoFs::get_refinement_edge_indices(level)) come from the hierarchy;
  // none on globally refined meshes
  void reinit(const MGDoFHandler<dim> &dof_handler, const MGConstrainedDoFs &, const unsigned int level_) {
    clear();
    level = level_;
    mfgpu_desc d;
    check(mfgpu_mesh_desc(dof_handler.levels[level], &d), "mesh desc");
    if (d.number_type != number_type<Number>()) throw std::runtime_error("mesh / operator number type mismatch");
    const std::vector<uint32_t> edges = dof_handler.refinement_edge_indices(level);
    check(mfgpu_level_create(&d, edges.data(), (uint32_t)edges.size(), &lev), "LevelOperatorGpu::reinit");
    handle = mfgpu_level_operator(lev);  // owned by the level
    n_dofs = d.n_dofs;
    n_edge = (unsigned int)edges.size();
  }
  // the system matrix on the ACTIVE mesh (LaplaceOperatorGpu::reinit(dof_handler, constraints), :120-151; hanging nodes)
  void reinit_active(const MGDoFHandler<dim> &dof_handler) {
    clear();
    mfgpu_desc d;
    check(mfgpu_mesh_desc(dof_handler.active, &d), "mesh desc");
    if (d.number_type != number_type<Number>()) throw std::runtime_error("mesh / operator number type mismatch");
    check(mfgpu_create(&d, &own), "LevelOperatorGpu::reinit_active");
    handle = own;
    n_dofs = d.n_dofs;
    n_edge = 0;
  }
  bool has_edges() const { return n_edge > 0; }
  // laplace_operator_gpu.h:306-352: the edge matrices of deal.II's Multigrid::set_edge_matrices
  void vmult_interface_down(VectorType &dst, const VectorType &src) const {
    check(mfgpu_level_vmult_interface_down(lev, dst.getData(), src.getDataRO(), nullptr), "vmult_interface_down");
  }
  void vmult_interface_up(VectorType &dst, const VectorType &src) const {
    check(mfgpu_level_vmult_interface_up(lev, dst.getData(), src.getDataRO(), nullptr), "vmult_interface_up");
  }
  unsigned int m() const { return n_dofs; }
  unsigned int n() const { return n_dofs; }
  void vmult(VectorType &dst, const VectorType &src) const {
    check(mfgpu_vmult(handle, dst.getData(), src.getDataRO(), nullptr), "vmult");
  }
  void vmult_add(VectorType &dst, const VectorType &src) const {
    check(mfgpu_vmult_add(handle, dst.getData(), src.getDataRO(), nullptr), "vmult_add");
  }
  void compute_diagonal() {
    inverse_diagonal_matrix = std::make_shared<DiagonalMatrix<Number>>();
    inverse_diagonal_matrix->get_vector().reinit(n_dofs);
    check(mfgpu_compute_inverse_diagonal(handle, inverse_diagonal_matrix->get_vector().getData(), nullptr), "compute_diagonal");
  }
  const std::shared_ptr<DiagonalMatrix<Number>> get_diagonal_inverse() const { return inverse_diagonal_matrix; }
  void set_constrained_values(VectorType &v, Number value) const {
    check(mfgpu_set_constrained_values(handle, v.getData(), (double)value, nullptr), "set_constrained_values");
  }
  std::size_t memory_consumption() const { return mfgpu_memory_consumption(handle); }
  unsigned int level = 0;

private:
  mfgpu_level *lev = nullptr;
  mfgpu_handle *own = nullptr;     // reinit_active
  mfgpu_handle *handle = nullptr;  // the level's operator (owned by lev) or own
  unsigned int n_dofs = 0, n_edge = 0;
  std::shared_ptr<DiagonalMatrix<Number>> inverse_diagonal_matrix;
};

// mg_transfer_matrix_free_gpu.h:140-252
template <int dim, typename Number>
class MGTransferMatrixFreeGpu {
public:
  MGTransferMatrixFreeGpu() = default;
  explicit MGTransferMatrixFreeGpu(const MGConstrainedDoFs &) {}
  ~MGTransferMatrixFreeGpu() { clear(); }
  void clear() {
    for (mfgpu_transfer *t : transfers) mfgpu_transfer_destroy(t);
    for (mfgpu_index_pairs *p : to_mg) mfgpu_index_pairs_destroy(p);
    for (mfgpu_index_pairs *p : from_mg) mfgpu_index_pairs_destroy(p);
    transfers.clear();
    to_mg.clear();
    from_mg.clear();
  }
  // :150-330 build: one transfer per level pair (adaptive meshes: over the refined cells of the coarser level), and the
  // index pairs of copy_to_mg / copy_from_mg
  void build(const MGDoFHandler<dim> &dof_handler) {
    clear();
    for (unsigned int l = 1; l < dof_handler.n_levels(); ++l) {
      mfgpu_transfer *t = nullptr;
      if (dof_handler.is_adaptive()) {
        const uint32_t *cd = nullptr, *fd = nullptr;
        const int64_t nc = mfgpu_mg_transfer_arrays(dof_handler.hierarchy, (int)l, &cd, &fd);
        mfgpu_desc dc, df;
        check(mfgpu_mesh_desc(dof_handler.levels[l - 1], &dc), "mesh desc");
        check(mfgpu_mesh_desc(dof_handler.levels[l], &df), "mesh desc");
        check(mfgpu_transfer_create(dim, (int)dof_handler.degree, number_type<Number>(), (uint32_t)nc, cd, fd, dc.n_dofs,
                                    df.n_dofs, dc.constrained_dofs, dc.n_constrained, nullptr, &t), "transfer build");
      } else {
        check(mfgpu_transfer_create_from_meshes(dof_handler.levels[l - 1], dof_handler.levels[l], &t), "transfer build");
      }
      transfers.push_back(t);
    }
    if (dof_handler.is_adaptive())
      for (unsigned int l = 0; l < dof_handler.n_levels(); ++l) {
        const uint32_t *a = nullptr, *b = nullptr;
        const int64_t n = mfgpu_mg_copy_pairs(dof_handler.hierarchy, (int)l, &a, &b);
        mfgpu_index_pairs *to = nullptr, *from = nullptr;
        check(mfgpu_index_pairs_create(b, a, (uint32_t)n, &to), "copy_to_mg pairs");      // level <- active
        check(mfgpu_index_pairs_create(a, b, (uint32_t)n, &from), "copy_from_mg pairs");  // active <- level
        to_mg.push_back(to);
        from_mg.push_back(from);
      }
  }
  void prolongate(const unsigned int to_level, GpuVector<Number> &dst, const GpuVector<Number> &src) const {  // .cu:595-627
    check(mfgpu_transfer_prolongate(transfers.at(to_level - 1), dst.getData(), src.getDataRO(), nullptr), "prolongate");
  }
  void restrict_and_add(const unsigned int from_level, GpuVector<Number> &dst, const GpuVector<Number> &src) const {  // :631-660
    check(mfgpu_transfer_restrict_and_add(transfers.at(from_level - 1), dst.getData(), src.getDataRO(), nullptr), "restrict_and_add");
  }
  // copy_to_mg / copy_from_mg (:690-760).  Globally refined: the active vector IS the finest level's.  Adaptive: the
  // copy_indices of every level (dofs of the cells active on it, off its refinement edge)
  void copy_to_mg(const MGDoFHandler<dim> &dh, MGLevelObject<GpuVector<Number>> &dst, const GpuVector<Number> &src) const {
    for (unsigned int l = dst.min_level(); l <= dst.max_level(); ++l) {
      if (dst[l].size() != dh.n_dofs(l)) dst[l].reinit(dh.n_dofs(l));
      dst[l] = Number(0);
      if (!to_mg.empty())
        check(mfgpu_vec_copy_pairs(to_mg[l], dst[l].getData(), src.getDataRO(), number_type<Number>(), nullptr), "copy_to_mg");
    }
    if (to_mg.empty()) dst[dst.max_level()].equ(1, src);
  }
  void copy_from_mg(const MGDoFHandler<dim> &dh, GpuVector<Number> &dst, const MGLevelObject<GpuVector<Number>> &src) const {
    if (from_mg.empty()) {
      dst.equ(1, src[src.max_level()]);
      return;
    }
    if (dst.size() != dh.n_dofs()) dst.reinit(dh.n_dofs());
    dst = Number(0);
    for (unsigned int l = src.min_level(); l <= src.max_level(); ++l)
      check(mfgpu_vec_copy_pairs(from_mg[l], dst.getData(), src[l].getDataRO(), number_type<Number>(), nullptr), "copy_from_mg");
  }
  std::size_t memory_consumption() const {
    std::size_t s = 0;
    for (mfgpu_transfer *t : transfers) s += mfgpu_transfer_memory_consumption(t);
    return s;
  }

private:
  std::vector<mfgpu_transfer *> transfers;
  std::vector<mfgpu_index_pairs *> to_mg, from_mg;
};

// deal.II PreconditionChebyshev as poisson_mg.cu:343-362 configures it (degree 5, smoothing_range 15, inner
// preconditioner = inverse diagonal).  The largest eigenvalue of D^-1 A is estimated by power iteration (deal.II: the
// Lanczos values of eig_cg_n_iterations CG steps): at least that many steps, on until the estimate settles to 1 %, and
// enlarged by 20 %.
template <typename MatrixType, typename VectorType>
class PreconditionChebyshev {
public:
  typedef typename VectorType::value_type Number;
  struct AdditionalData {
    unsigned int degree = 5;
    double smoothing_range = 15.;
    unsigned int eig_cg_n_iterations = 15;
    std::shared_ptr<DiagonalMatrix<Number>> preconditioner;
  };
  void initialize(const MatrixType &A, const AdditionalData &d) {
    matrix = &A;
    data = d;
    const unsigned int N = A.m();
    r.reinit(N);
    t.reinit(N);
    upd.reinit(N);
    std::vector<Number> init(N);
    for (unsigned int i = 0; i < N; ++i) init[i] = (Number)(std::sin(0.7 * i) + 0.3);
    VectorType v(init), w(N);
    // power iteration on D^-1 A: |w| / |v| approaches lambda_max from BELOW, so eig_cg_n_iterations steps are only
    // the minimum; iterate on until the estimate moves by less than 1 % per step (an underestimate would leave the
    // top modes outside [lambda_min, lambda_max], where the Chebyshev polynomial amplifies them), then add 20 %
    double lam = 1.0, prev = 0.0;
    for (unsigned int k = 0; k < 200; ++k) {
      A.vmult(w, v);
      w.scale(d.preconditioner->get_vector());
      const double nw = w.l2_norm(), nv = v.l2_norm();
      lam = nw / nv;
      v.equ((Number)(1.0 / nw), w);
      if (k + 1 >= std::max(5u, d.eig_cg_n_iterations) && std::fabs(lam - prev) <= 0.01 * lam) break;
      prev = lam;
    }
    lambda_max = 1.2 * lam;
    lambda_min = lambda_max / d.smoothing_range;
  }
  // dst = p(A) src, zero start (the pre-smoothing step and the coarse "smoother")
  void vmult(VectorType &dst, const VectorType &src) const { run(dst, src, true); }
  // one more sweep on a non-zero iterate (post-smoothing)
  void step(VectorType &dst, const VectorType &src) const { run(dst, src, false); }
  double lambda_max = 0, lambda_min = 0;

private:
  void run(VectorType &x, const VectorType &b, bool zero_start) const {
    const double theta = 0.5 * (lambda_max + lambda_min), delta = 0.5 * (lambda_max - lambda_min);
    const double sigma = theta / delta;
    double rho = 1.0 / sigma;
    const VectorType &dinv = data.preconditioner->get_vector();
    r.equ(1, b);
    if (!zero_start) {
      matrix->vmult(t, x);
      r.add(-1, t);
    }
    upd.equ((Number)(1.0 / theta), r);
    upd.scale(dinv);
    if (zero_start)
      x.equ(1, upd);
    else
      x.add(1, upd);
    for (unsigned int k = 1; k < data.degree; ++k) {
      matrix->vmult(t, upd);
      r.add(-1, t);
      const double rho_new = 1.0 / (2.0 * sigma - rho);
      t.equ((Number)(2.0 * rho_new / delta), r);
      t.scale(dinv);
      upd.sadd((Number)(rho_new * rho), 1, t);
      x.add(1, upd);
      rho = rho_new;
    }
  }
  const MatrixType *matrix = nullptr;
  AdditionalData data;
  mutable VectorType r, t, upd;
};

// deal.II Multigrid (V-cycle) + PreconditionMG for a globally refined hierarchy
template <int dim, typename LevelMatrixType, typename Number, typename CoarseSolver>
class MultigridPreconditioner {
public:
  typedef GpuVector<Number> VectorType;
  typedef PreconditionChebyshev<LevelMatrixType, VectorType> Smoother;
  MultigridPreconditioner(const MGDoFHandler<dim> &dh, const MGLevelObject<LevelMatrixType> &matrices,
                          const CoarseSolver &coarse, const MGTransferMatrixFreeGpu<dim, Number> &transfer,
                          const MGLevelObject<Smoother> &smoother)
      : dof_handler(&dh), matrices(&matrices), coarse(&coarse), transfer(&transfer), smoother(&smoother) {
    const unsigned int top = matrices.max_level();
    defect.resize(0, top);
    solution.resize(0, top);
    tmp.resize(0, top);
    edge.resize(0, top);
    for (unsigned int l = 0; l <= top; ++l) {
      defect[l].reinit(dh.n_dofs(l));
      solution[l].reinit(dh.n_dofs(l));
      tmp[l].reinit(dh.n_dofs(l));
      edge[l].reinit(dh.n_dofs(l));
    }
  }
  // PreconditionMG::vmult: copy_to_mg, one V-cycle, copy_from_mg
  void vmult(VectorType &dst, const VectorType &src) const {
    transfer->copy_to_mg(*dof_handler, defect, src);
    level_v_step(matrices->max_level());
    transfer->copy_from_mg(*dof_handler, dst, solution);
  }

private:
  // deal.II Multigrid::level_v_step with the edge matrices of set_edge_matrices (poisson_mg.cu:365-375): the level
  // matrix treats the refinement-edge dofs as constrained; their rows enter the residual through the interface matrix
  // (edge_out), and after the coarse-grid correction the edge values act on the interior rows (edge_in)
  void level_v_step(unsigned int l) const {
    if (l == 0) {
      (*coarse)(0, solution[0], defect[0]);
      return;
    }
    const LevelMatrixType &Al = (*matrices)[l];
    (*smoother)[l].vmult(solution[l], defect[l]);       // pre-smoothing from zero
    Al.vmult(tmp[l], solution[l]);                      // t = A x
    if (Al.has_edges()) {
      Al.vmult_interface_down(edge[l], solution[l]);
      tmp[l].add(1, edge[l]);
    }
    tmp[l].sadd(-1, 1, defect[l]);                      // t = defect - t
    transfer->restrict_and_add(l, defect[l - 1], tmp[l]);
    solution[l - 1] = Number(0);
    level_v_step(l - 1);
    transfer->prolongate(l, tmp[l], solution[l - 1]);   // coarse-grid correction
    solution[l].add(1, tmp[l]);
    if (Al.has_edges()) {
      Al.vmult_interface_up(edge[l], solution[l]);
      tmp[l].equ(1, defect[l]);
      tmp[l].add(-1, edge[l]);
      (*smoother)[l].step(solution[l], tmp[l]);         // post-smoothing on the corrected right-hand side
    } else {
      (*smoother)[l].step(solution[l], defect[l]);      // post-smoothing
    }
  }
  const MGDoFHandler<dim> *dof_handler;
  const MGLevelObject<LevelMatrixType> *matrices;
  const CoarseSolver *coarse;
  const MGTransferMatrixFreeGpu<dim, Number> *transfer;
  const MGLevelObject<Smoother> *smoother;
  mutable MGLevelObject<VectorType> defect, solution, tmp, edge;
};

}  // namespace mfgpu_shim
#endif
