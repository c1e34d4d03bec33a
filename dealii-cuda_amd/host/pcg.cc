// pcg: the caller the SURVEY.md 8(f) rows N1 / N2 exist for, in the shape of the reference's poisson.cu
// (:223 compute_diagonal, :237-260 preconditioned CG on LaplaceOperatorGpu / GpuVector).  deal.II's
// SolverCG + PreconditionChebyshev (library code, out of scope) are replaced by a plain Jacobi-
// preconditioned CG written on the shim's GpuVector operations; everything on the device goes through
// the C-ABI.  The right-hand side is A x* for a known x*, so the driver checks its own answer.
// Output (QUIET line of poisson.cu:271-272):  dim  degree  n_dofs  iterations  wall_seconds  rel_error
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <iostream>

#include "mfgpu_shim.h"

using namespace mfgpu_shim;

#ifndef DEGREE_FE
#define DEGREE_FE 4
#endif
#ifndef DIMENSION
#define DIMENSION 3
#endif
typedef double number;

template <int dim, int fe_degree>
int run(int n_ref, bool adaptive) {
  Triangulation<dim> triangulation;
  FE_Q<dim> fe(fe_degree);
  DoFHandler<dim> dof_handler(triangulation);
  ConstraintMatrix constraints;
  LaplaceOperatorGpu<dim, fe_degree, number> A;
  bmop_setup_mesh(triangulation, CUBE, adaptive, n_ref);
  dof_handler.distribute_dofs(fe, number_type<number>());
  constraints.clear();
  constraints.close();
  A.reinit(dof_handler, constraints);
  const unsigned int N = A.n();

  // x*: smooth, zero on constrained dofs; b = A x*
  std::vector<number> xs(N);
  for (unsigned int i = 0; i < N; ++i) xs[i] = std::sin(0.37 * i) + 0.5 * std::cos(0.011 * i);
  GpuVector<number> x_star(xs), b(N), x(N), r(N), z(N), p(N), q(N);
  A.set_constrained_values(x_star, 0);
  A.vmult(b, x_star);

  A.compute_diagonal();
  const auto prec = A.get_diagonal_inverse();

  mfgpu_device_synchronize();
  const auto t0 = std::chrono::steady_clock::now();
  x = number(0);
  r.equ(1, b);
  prec->vmult(z, r);
  p.equ(1, z);
  number rz = r * z;
  const number tol = 1e-12 * b.l2_norm();
  unsigned int it = 0;
  for (it = 1; it <= 10000; ++it) {
    A.vmult(q, p);
    const number alpha = rz / (p * q);
    x.add(alpha, p);
    r.add(-alpha, q);
    if (r.l2_norm() <= tol) break;
    prec->vmult(z, r);
    const number rz_new = r * z;
    p.sadd(rz_new / rz, 1, z);
    rz = rz_new;
  }
  mfgpu_device_synchronize();
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  x.add(-1, x_star);
  const double err = x.l2_norm() / x_star.l2_norm();
  printf("%8d %8d %12u %8u %14.8g %12.4g\n", dim, fe_degree, N, it, wall, err);
  return (it <= 10000 && err < 1e-8) ? 0 : 2;
}

int main(int argc, char **argv) {
  try {
    const int n_ref = argc > 1 ? atoi(argv[1]) : 3;
    const bool adaptive = argc > 2 && atoi(argv[2]) != 0;
    return run<DIMENSION, DEGREE_FE>(n_ref, adaptive);
  } catch (std::exception &exc) {
    std::cerr << "Exception on processing: " << exc.what() << std::endl;
    return 1;
  }
}
