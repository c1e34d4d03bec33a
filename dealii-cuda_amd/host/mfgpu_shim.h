// Header-only C++ shim over the C-ABI (include/mfgpu.h) that keeps the reference's class surface for
// the operator-apply path, so bmop / poisson style callers compile with the deal.II mesh types replaced
// by the minimal stand-ins below.  File:line citations are relative to the reference tree.
//
//   GpuVector<Number>                      matrix_free_gpu/gpu_vec.h:21-176  (only the pieces on the path)
//   ConstraintHandlerGpu<Number>           matrix_free_gpu/constraint_handler_gpu.h:13-59
//   MatrixFreeGpu<dim,Number>              matrix_free_gpu/matrix_free_gpu.h:80-229
//   LaplaceOperatorGpu<dim,degree,Number>  laplace_operator_gpu.h:35-96
//   Triangulation / FE_Q / DoFHandler / ConstraintMatrix / QGauss: just enough of deal.II for
//   bmop.cu:111-130 and bmop_common.h:108-120
//
// Errors: the C-ABI returns codes; the shim throws std::runtime_error (the reference throws
// dealii::ExcMessage, cuda_utils.cuh:15-25).
#ifndef MFGPU_SHIM_H
#define MFGPU_SHIM_H

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "mfgpu.h"

namespace mfgpu_shim {

inline void check(int rc, const char *what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + mfgpu_last_error());
}
template <typename Number>
constexpr int number_type() {
  static_assert(std::is_same<Number, double>::value || std::is_same<Number, float>::value, "double or float");
  return std::is_same<Number, double>::value ? MFGPU_F64 : MFGPU_F32;
}

// ---- deal.II stand-ins (setup side only) -----------------------------------------------------
enum domain_case_t { CUBE, BALL };  // poisson_common.h:27-30

template <int dim>
class Triangulation {
public:
  // bmop_setup_mesh (bmop_common.h:108-120) records the recipe; cells are created by DoFHandler
  bool adaptive = false;
  domain_case_t domain = CUBE;
  int n_ref = 0;
  void refine_global(int times) { n_ref += times; }
};

template <int dim>
void bmop_setup_mesh(Triangulation<dim> &tria, domain_case_t domain, bool pseudo_adaptive_grid, int n_ref) {
  // BALL (hyper_ball + spherical boundary manifold, poisson_common.h:65-70): global refinement only -- the
  // pseudo-adaptive recipe on the ball (bmop_common.h:58-70) needs hanging nodes on an unstructured mesh
  if (domain == BALL && pseudo_adaptive_grid)
    throw std::runtime_error("BALL_GRID with ADAPTIVE_GRID is not implemented (uniformly refined ball only)");
  tria.domain = domain;
  tria.adaptive = pseudo_adaptive_grid;
  tria.n_ref = n_ref;
}

template <int dim>
struct FE_Q {
  explicit FE_Q(unsigned int degree) : degree(degree) {}
  unsigned int degree;
};

template <int dim_>
struct QGauss {
  explicit QGauss(unsigned int n) : n(n) {}
  unsigned int size() const { return n; }
  unsigned int n;
};

class ConstraintMatrix {
public:
  void clear() { closed = false; }
  void close() { closed = true; }
  bool closed = false;  // boundary values + hanging nodes come with the mesh stand-in
};

template <int dim>
class DoFHandler {
public:
  explicit DoFHandler(const Triangulation<dim> &tria) : tria(&tria) {}
  ~DoFHandler() { clear(); }
  DoFHandler(const DoFHandler &) = delete;
  DoFHandler &operator=(const DoFHandler &) = delete;
  void clear() {
    if (mesh) mfgpu_mesh_destroy(mesh);
    mesh = nullptr;
  }
  // DoFHandler::distribute_dofs (bmop.cu:116) + interpolate_boundary_values + make_hanging_node_
  // constraints (:118-124): all produced by the mesh stand-in
  template <typename Number = double>
  void distribute_dofs(const FE_Q<dim> &fe, int number_type = MFGPU_F64) {
    clear();
    degree = fe.degree;
    if (tria->domain == BALL) {
      check(mfgpu_mesh_create_ball(dim, (int)fe.degree, tria->n_ref, number_type, &mesh), "mesh");
    } else if (tria->adaptive) {
      check(mfgpu_mesh_create_adaptive(dim, (int)fe.degree, tria->n_ref, number_type, &mesh), "mesh");
    } else {
      uint32_t nper[3] = {1u << tria->n_ref, 1u << tria->n_ref, 1u << tria->n_ref};
      check(mfgpu_mesh_create_uniform(dim, (int)fe.degree, nper, -1.0, 1.0, 0, 0, number_type, &mesh), "mesh");
    }
    check(mfgpu_mesh_desc(mesh, &desc), "mesh desc");
  }
  unsigned int n_dofs() const { return desc.n_dofs; }
  const Triangulation<dim> *tria;
  mfgpu_mesh *mesh = nullptr;
  mfgpu_desc desc{};
  unsigned int degree = 0;
};

// ---- GpuVector -------------------------------------------------------------------------------
template <typename Number>
class GpuVector {
public:
  typedef Number value_type;
  GpuVector() = default;
  explicit GpuVector(unsigned int s) { reinit(s); }  // zero-filled (gpu_vec.cu:24-40)
  explicit GpuVector(const std::vector<Number> &host) { *this = host; }
  GpuVector(const GpuVector &) = delete;
  GpuVector &operator=(const GpuVector &) = delete;
  ~GpuVector() { mfgpu_vec_free(vec_dev); }
  void reinit(unsigned int s) {  // gpu_vec.cu:166-182
    mfgpu_vec_free(vec_dev);
    vec_dev = nullptr;
    check(mfgpu_vec_alloc(&vec_dev, s, number_type<Number>()), "GpuVector::reinit");
    _size = s;
  }
  unsigned int size() const { return _size; }
  Number *getData() { return static_cast<Number *>(vec_dev); }
  const Number *getDataRO() const { return static_cast<const Number *>(vec_dev); }
  GpuVector &operator=(const Number v) {  // vec_init (gpu_vec.cu:281-291,373-381)
    check(mfgpu_vec_fill(vec_dev, _size, number_type<Number>(), (double)v, nullptr), "GpuVector::operator=");
    return *this;
  }
  GpuVector &operator=(const std::vector<Number> &host) {
    if (host.size() != _size) reinit((unsigned int)host.size());
    check(mfgpu_vec_from_host(vec_dev, host.data(), _size, number_type<Number>()), "GpuVector::fromHost");
    return *this;
  }
  std::vector<Number> toVector() const {  // gpu_vec.h:77-81 with std::vector for dealii::Vector
    std::vector<Number> v(_size);
    check(mfgpu_vec_to_host(v.data(), vec_dev, _size, number_type<Number>()), "GpuVector::toVector");
    return v;
  }
  void swap(GpuVector &other) {  // gpu_vec.h:164-172
    std::swap(vec_dev, other.vec_dev);
    std::swap(_size, other._size);
  }
  unsigned int memory_consumption() const { return _size * sizeof(Number); }

  // ---- BLAS-1 and reductions (gpu_vec.h:105-157)
  Number operator*(const GpuVector &v) const {  // scalar product
    double r = 0;
    check(mfgpu_vec_dot(vec_dev, v.vec_dev, _size, number_type<Number>(), nullptr, &r), "GpuVector::operator*");
    return (Number)r;
  }
  void add(const GpuVector &V) { sadd(1, 1, V); }
  void add(const Number a, const GpuVector &V) { sadd(1, a, V); }
  void sadd(const Number s, const GpuVector &V) { sadd(s, 1, V); }
  void sadd(const Number s, const Number a, const GpuVector &V) {  // this = s*this + a*V
    check(mfgpu_vec_sadd(vec_dev, (double)s, (double)a, V.vec_dev, _size, number_type<Number>(), nullptr), "GpuVector::sadd");
  }
  GpuVector &operator+=(const GpuVector &x) { sadd(1, 1, x); return *this; }
  GpuVector &operator-=(const GpuVector &x) { sadd(1, -1, x); return *this; }
  Number add_and_dot(const Number a, const GpuVector &x, const GpuVector &v) {  // this += a*x; return this . v
    double r = 0;
    check(mfgpu_vec_add_and_dot(vec_dev, (double)a, x.vec_dev, v.vec_dev, _size, number_type<Number>(), nullptr, &r),
          "GpuVector::add_and_dot");
    return (Number)r;
  }
  void scale(const GpuVector &v) {  // element-wise multiplication
    check(mfgpu_vec_scale(vec_dev, v.vec_dev, _size, number_type<Number>(), nullptr), "GpuVector::scale");
  }
  GpuVector &operator/=(const GpuVector &x) {  // element-wise division
    check(mfgpu_vec_divide(vec_dev, x.vec_dev, _size, number_type<Number>(), nullptr), "GpuVector::operator/=");
    return *this;
  }
  GpuVector &invert() {
    check(mfgpu_vec_invert(vec_dev, _size, number_type<Number>(), nullptr), "GpuVector::invert");
    return *this;
  }
  void equ(const Number a, const GpuVector &x) {  // this = a*x
    if (x._size != _size) reinit(x._size);
    check(mfgpu_vec_equ(vec_dev, (double)a, x.vec_dev, _size, number_type<Number>(), nullptr), "GpuVector::equ");
  }
  GpuVector &operator*=(const Number a) {
    check(mfgpu_vec_mul(vec_dev, (double)a, _size, number_type<Number>(), nullptr), "GpuVector::operator*=");
    return *this;
  }
  Number l2_norm() const {
    double r = 0;
    check(mfgpu_vec_l2_norm(vec_dev, _size, number_type<Number>(), nullptr, &r), "GpuVector::l2_norm");
    return (Number)r;
  }
  bool all_zero() const {
    int r = 0;
    check(mfgpu_vec_all_zero(vec_dev, _size, number_type<Number>(), nullptr, &r), "GpuVector::all_zero");
    return r != 0;
  }

private:
  void *vec_dev = nullptr;
  unsigned int _size = 0;
};

// DiagonalMatrix<GpuVector<Number>> as poisson.cu:242-250 uses it: vmult = element-wise scaling
template <typename Number>
class DiagonalMatrix {
public:
  GpuVector<Number> &get_vector() { return diagonal; }
  const GpuVector<Number> &get_vector() const { return diagonal; }
  void vmult(GpuVector<Number> &dst, const GpuVector<Number> &src) const {
    dst.equ(1, src);
    dst.scale(diagonal);
  }

private:
  GpuVector<Number> diagonal;
};

// ---- ConstraintHandlerGpu ---------------------------------------------------------------------
// The reference brackets the cell loop with save / load_and_add kernels over this index list
// (constraint_handler_gpu.cu:126-193).  In this build the bracket is fused into the cell-loop kernel
// (constrained rows read as zero, identity rows written by the owning batch), so the class only keeps
// the list it would have uploaded.
template <typename Number>
class ConstraintHandlerGpu {
public:
  void reinit(const ConstraintMatrix &, const mfgpu_desc &d) {
    constrained_indices.assign(d.constrained_dofs, d.constrained_dofs + d.n_constrained);
  }
  unsigned int n_constrained_dofs() const { return (unsigned int)constrained_indices.size(); }
  std::size_t memory_consumption() const { return constrained_indices.size() * sizeof(uint32_t); }
  std::vector<uint32_t> constrained_indices;
};

// ---- MatrixFreeGpu ------------------------------------------------------------------------------
template <int dim, typename Number>
class MatrixFreeGpu {
public:
  struct AdditionalData {
    bool use_coloring = false;  // MATRIX_FREE_COLOR (laplace_operator_gpu.h:132-136): selects
                                // MFGPU_COLORED_SCATTER; default is the two-pass scatter
  };
  unsigned int n_cells_tot = 0, n_dofs = 0, fe_degree = 0, dofs_per_cell = 0, qpts_per_cell = 0, num_colors = 0;
  bool use_coloring = false;

  // matrix_free_gpu.cu:448-563
  void reinit(const DoFHandler<dim> &dof_handler, const ConstraintMatrix &, const QGauss<1> &quad,
              const AdditionalData additional_data = AdditionalData()) {
    free();
    if (quad.size() != dof_handler.degree + 1)
      throw std::runtime_error("n_q_points_1d must be equal to fe_degree+1.");  // matrix_free_gpu.cu:485
    mfgpu_desc d = dof_handler.desc;
    if (d.number_type != number_type<Number>()) throw std::runtime_error("mesh / operator number type mismatch");
    use_coloring = additional_data.use_coloring;
    if (use_coloring) d.flags |= MFGPU_COLORED_SCATTER;
    check(mfgpu_create(&d, &handle), "MatrixFreeGpu::reinit");
    n_cells_tot = d.n_cells;
    n_dofs = d.n_dofs;
    fe_degree = (unsigned int)d.degree;
    dofs_per_cell = qpts_per_cell = 1;
    for (int i = 0; i < dim; ++i) dofs_per_cell *= fe_degree + 1;
    qpts_per_cell = dofs_per_cell;
    uint64_t st[8];
    mfgpu_plan_stats(handle, st);
    num_colors = (unsigned int)st[1];
  }
  void free() {  // matrix_free_gpu.cu:566-596
    mfgpu_destroy(handle);
    handle = nullptr;
  }
  std::size_t memory_consumption() const { return mfgpu_memory_consumption(handle); }
  ~MatrixFreeGpu() { free(); }
  mfgpu_handle *handle = nullptr;
};

// ---- LaplaceOperatorGpu ---------------------------------------------------------------------------
template <int dim, int fe_degree, typename Number>
class LaplaceOperatorGpu {
public:
  typedef Number value_type;
  typedef GpuVector<Number> VectorType;

  void clear() { data.free(); }  // laplace_operator_gpu.h:110-117

  // laplace_operator_gpu.h:120-151: MatrixFreeGpu::reinit + ConstraintHandlerGpu::reinit +
  // evaluate_coefficient (on the device, from the quadrature points)
  void reinit(const DoFHandler<dim> &dof_handler, const ConstraintMatrix &constraints) {
    if ((int)dof_handler.degree != fe_degree) throw std::runtime_error("FE degree mismatch");
    typename MatrixFreeGpu<dim, Number>::AdditionalData additional_data;
#ifdef MATRIX_FREE_COLOR
    additional_data.use_coloring = true;
#endif
    data.reinit(dof_handler, constraints, QGauss<1>(fe_degree + 1), additional_data);
    constraint_handler.reinit(constraints, dof_handler.desc);
  }
  unsigned int m() const { return data.n_dofs; }
  unsigned int n() const { return data.n_dofs; }

  void vmult(VectorType &dst, const VectorType &src) const {  // :216-223
    check(mfgpu_vmult(data.handle, dst.getData(), src.getDataRO(), nullptr), "vmult");
  }
  void Tvmult(VectorType &dst, const VectorType &src) const { vmult(dst, src); }  // symmetric, :227-234
  void vmult_add(VectorType &dst, const VectorType &src) const {                  // :286-303
    check(mfgpu_vmult_add(data.handle, dst.getData(), src.getDataRO(), nullptr), "vmult_add");
  }
  void Tvmult_add(VectorType &dst, const VectorType &src) const { vmult_add(dst, src); }
  Number el(unsigned int, unsigned int) const { throw std::runtime_error("matrix-free: no element access"); }
  // laplace_operator_gpu.h:401-418: inverse diagonal, constrained rows 1
  void compute_diagonal() {
    if (!inverse_diagonal_matrix) inverse_diagonal_matrix = std::make_shared<DiagonalMatrix<Number>>();
    VectorType &inv_diag = inverse_diagonal_matrix->get_vector();
    inv_diag.reinit(m());
    check(mfgpu_compute_inverse_diagonal(data.handle, inv_diag.getData(), nullptr), "compute_diagonal");
    diagonal_is_available = true;
  }
  const std::shared_ptr<DiagonalMatrix<Number>> get_diagonal_inverse() const {  // :420-429
    if (!diagonal_is_available) throw std::runtime_error("get_diagonal_inverse: call compute_diagonal first");
    return inverse_diagonal_matrix;
  }
  void set_constrained_values(VectorType &v, Number value) const {  // constraint_handler_gpu.cu:126-137
    check(mfgpu_set_constrained_values(data.handle, v.getData(), (double)value, nullptr), "set_constrained_values");
  }
  std::size_t memory_consumption() const {  // :434-445
    return data.memory_consumption() + constraint_handler.memory_consumption();
  }

private:
  MatrixFreeGpu<dim, Number> data;
  mutable ConstraintHandlerGpu<Number> constraint_handler;
  std::shared_ptr<DiagonalMatrix<Number>> inverse_diagonal_matrix;
  bool diagonal_is_available = false;
};

}  // namespace mfgpu_shim
#endif
