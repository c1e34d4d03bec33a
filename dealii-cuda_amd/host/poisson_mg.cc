// poisson_mg: conjugate gradients preconditioned by one geometric-multigrid V-cycle, assembled as the reference's
// poisson_mg.cu / bmop_mg.cu assemble it (:199-209 level matrices, :325-326 transfer, :334-335 coarse CG, :343-362
// Chebyshev smoothers of degree 5 with smoothing range 15, :369-380 Multigrid + PreconditionMG, CG to 1e-12).  deal.II's
// classes are the stand-ins of mfgpu_shim.h / mfgpu_shim_mg.h; everything on the device goes through the C-ABI.  The
// right-hand side is A x* for a known x*, so the driver checks its own answer.
// Output:  dim  degree  n_dofs  levels  cg_iterations  wall_seconds  rel_error
// usage: poisson-mg-<dim>d-p<k> n_ref          (-DBALL_GRID: the BALL domain; -DADAPTIVE_GRID: the pseudo-adaptive
// mesh with hanging nodes, local smoothing with refinement-edge matrices, poisson_mg.cu:365-375)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <iostream>

#include "mfgpu_shim_mg.h"

using namespace mfgpu_shim;

#ifndef DEGREE_FE
#define DEGREE_FE 4
#endif
#ifndef DIMENSION
#define DIMENSION 3
#endif
typedef double number;
typedef GpuVector<number> VectorType;

// coarse solver (poisson_mg.cu:61-83): unpreconditioned CG, relative tolerance 1e-10
template <typename MatrixType>
class MGCoarseIterative {
public:
  void initialize(const MatrixType &matrix) {
    coarse_matrix = &matrix;
    const unsigned int N = matrix.m();
    r.reinit(N);
    p.reinit(N);
    q.reinit(N);
  }
  void operator()(const unsigned int, VectorType &dst, const VectorType &src) const {
    dst = number(0);
    r.equ(1, src);
    p.equ(1, r);
    number rr = r * r;
    const number tol = 1e-10 * std::sqrt(rr);
    for (unsigned int it = 0; it < 10000 && std::sqrt(rr) > tol; ++it) {
      coarse_matrix->vmult(q, p);
      const number alpha = rr / (p * q);
      dst.add(alpha, p);
      r.add(-alpha, q);
      const number rr_new = r * r;
      p.sadd(rr_new / rr, 1, r);
      rr = rr_new;
    }
  }
  const MatrixType *coarse_matrix = nullptr;
  mutable VectorType r, p, q;
};

template <int dim, int fe_degree>
int run(int n_ref) {
  typedef LevelOperatorGpu<dim, fe_degree, number> LevelMatrixType;
  Triangulation<dim> triangulation;
#if defined(BALL_GRID)
  bmop_setup_mesh(triangulation, BALL, false, n_ref);
#elif defined(ADAPTIVE_GRID)
  bmop_setup_mesh(triangulation, CUBE, true, n_ref);
#else
  bmop_setup_mesh(triangulation, CUBE, false, n_ref);
#endif
  FE_Q<dim> fe(fe_degree);
  MGDoFHandler<dim> dof_handler(triangulation);
  dof_handler.distribute_mg_dofs(fe, number_type<number>());
  const unsigned int nlevels = dof_handler.n_levels();

  MGConstrainedDoFs mg_constrained_dofs;
  MGLevelObject<LevelMatrixType> mg_matrices;
  mg_matrices.resize(0, nlevels - 1);
  for (unsigned int level = 0; level < nlevels; ++level) {
    mg_matrices[level].reinit(dof_handler, mg_constrained_dofs, level);
    mg_matrices[level].compute_diagonal();
  }
  // the system matrix: on a globally refined mesh the finest level's; on the adaptive mesh the active cells' operator
  // with hanging nodes
  LevelMatrixType active_matrix;
  if (dof_handler.is_adaptive()) active_matrix.reinit_active(dof_handler);
  const LevelMatrixType &system_matrix = dof_handler.is_adaptive() ? active_matrix : mg_matrices[nlevels - 1];
  const unsigned int N = system_matrix.n();

  MGTransferMatrixFreeGpu<dim, number> mg_transfer(mg_constrained_dofs);
  mg_transfer.build(dof_handler);
  MGCoarseIterative<LevelMatrixType> mg_coarse;
  mg_coarse.initialize(mg_matrices[0]);
  typedef PreconditionChebyshev<LevelMatrixType, VectorType> SMOOTHER;
  MGLevelObject<SMOOTHER> mg_smoother;
  mg_smoother.resize(0, nlevels - 1);
  for (unsigned int level = 0; level < nlevels; ++level) {
    typename SMOOTHER::AdditionalData sd;
    sd.smoothing_range = 15.;
    sd.degree = 5;
    sd.eig_cg_n_iterations = 15;
    sd.preconditioner = mg_matrices[level].get_diagonal_inverse();
    mg_smoother[level].initialize(mg_matrices[level], sd);
  }
  MultigridPreconditioner<dim, LevelMatrixType, number, MGCoarseIterative<LevelMatrixType>> preconditioner(
      dof_handler, mg_matrices, mg_coarse, mg_transfer, mg_smoother);

  // x*: zero on the Dirichlet dofs; b = A x*
  std::vector<number> xs(N);
  for (unsigned int i = 0; i < N; ++i) xs[i] = std::sin(0.37 * i) + 0.5 * std::cos(0.011 * i);
  VectorType x_star(xs), b(N), x(N), r(N), z(N), p(N), q(N);
  system_matrix.set_constrained_values(x_star, 0);
  system_matrix.vmult(b, x_star);

  mfgpu_device_synchronize();
  const auto t0 = std::chrono::steady_clock::now();
  x = number(0);
  r.equ(1, b);
  preconditioner.vmult(z, r);
  p.equ(1, z);
  number rz = r * z;
  const number tol = 1e-12 * b.l2_norm();
  unsigned int it = 0;
  for (it = 1; it <= 1000; ++it) {
    system_matrix.vmult(q, p);
    const number alpha = rz / (p * q);
    x.add(alpha, p);
    r.add(-alpha, q);
    if (r.l2_norm() <= tol) break;
    preconditioner.vmult(z, r);
    const number rz_new = r * z;
    p.sadd(rz_new / rz, 1, z);
    rz = rz_new;
  }
  mfgpu_device_synchronize();
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  x.add(-1, x_star);
  const double err = x.l2_norm() / x_star.l2_norm();
  printf("%8d %8d %12u %8u %8u %14.8g %12.4g\n", dim, fe_degree, N, nlevels, it, wall, err);
  return (it <= 1000 && err < 1e-8) ? 0 : 2;
}

int main(int argc, char **argv) {
  try {
    const int n_ref = argc > 1 ? atoi(argv[1]) : 3;
    return run<DIMENSION, DEGREE_FE>(n_ref);
  } catch (std::exception &exc) {
    std::cerr << "Exception on processing: " << exc.what() << std::endl;
    return 1;
  }
}
