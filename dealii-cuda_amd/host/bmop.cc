// bmop: operator benchmark, same command line, compile-time switches and output line as the
// reference's bmop.cu (:45-64 N_ITERATIONS / DEGREE_FE / DIMENSION / BMOP_USE_FLOATS, :160-170
// BALL_GRID / ADAPTIVE_GRID, :186-192 argv max_ref [min_ref], :152 "dim\tdegree\tn_dofs\tsec_per_vmult").
// deal.II's mesh / dof classes are replaced by the stand-ins of mfgpu_shim.h; the operator runs
// through the C-ABI on the MI355X.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <iostream>

#include "mfgpu_shim.h"

using namespace mfgpu_shim;

#define N_ITERATIONS 100

#ifdef DEGREE_FE
const unsigned int degree_finite_element = DEGREE_FE;
#else
const unsigned int degree_finite_element = 4;
#endif

#ifdef DIMENSION
const unsigned int dimension = DIMENSION;
#else
const unsigned int dimension = 3;
#endif

#ifdef BMOP_USE_FLOATS
typedef float number;
#else
typedef double number;
#endif

template <int dim, int fe_degree>
class LaplaceProblem {
public:
  LaplaceProblem() : fe(fe_degree), dof_handler(triangulation), n_iterations(N_ITERATIONS) {}
  void run(int n_ref) {
#ifdef ADAPTIVE_GRID
    const bool pseudo_adaptive_grid = true;
#else
    const bool pseudo_adaptive_grid = false;
#endif
#ifdef BALL_GRID
    const domain_case_t domain = BALL;
#else
    const domain_case_t domain = CUBE;
#endif
    bmop_setup_mesh(triangulation, domain, pseudo_adaptive_grid, n_ref);
    setup_system();
    solve();
  }

private:
  void setup_system() {
    system_matrix.clear();
    dof_handler.distribute_dofs(fe, number_type<number>());
    constraints.clear();
    constraints.close();
    system_matrix.reinit(dof_handler, constraints);
    dst.reinit(system_matrix.n());
    src.reinit(system_matrix.n());
  }
  void solve() {
    mfgpu_device_synchronize();
    const auto t0 = std::chrono::steady_clock::now();
    dst = number(0.1);  // IC
    for (unsigned int i = 0; i < n_iterations; ++i) {
      dst.swap(src);
      system_matrix.vmult(dst, src);
    }
    mfgpu_device_synchronize();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%d\t%d\t%u\t%g\n", dim, fe_degree, dof_handler.n_dofs(), wall / n_iterations);
  }

  Triangulation<dim> triangulation;
  FE_Q<dim> fe;
  DoFHandler<dim> dof_handler;
  ConstraintMatrix constraints;
  LaplaceOperatorGpu<dim, fe_degree, number> system_matrix;
  GpuVector<number> src, dst;
  unsigned int n_iterations;
};

int main(int argc, char **argv) {
  try {
    int max_refinement = 1, min_refinement = 0;
    if (argc > 1) max_refinement = atoi(argv[1]);
    if (argc > 2) min_refinement = atoi(argv[2]);
    for (int r = min_refinement; r <= max_refinement; r++) {
      LaplaceProblem<dimension, degree_finite_element> laplace_problem;
      laplace_problem.run(r);
    }
  } catch (std::exception &exc) {
    std::cerr << "\n\n----------------------------------------------------\n"
              << "Exception on processing: \n" << exc.what() << "\nAborting!\n"
              << "----------------------------------------------------" << std::endl;
    return 1;
  }
  return 0;
}
