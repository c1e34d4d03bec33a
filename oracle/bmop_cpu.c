/* bmop-cpu driver on the CPU twin (TEST INFRASTRUCTURE / cpu_baseline: built by oracle/Makefile, never linked into
 * libmfgpu.so or the host/ product drivers).  Same command line and output as the reference's bmop-cpu.cc:
 *
 *     bmop-cpu-<dim>d-p<degree> [max_refinement [min_refinement]]            (bmop-cpu.cc:184-196)
 *     one line per refinement r: "dim \t degree \t n_dofs \t seconds per vmult"   (bmop-cpu.cc:154)
 *
 * for the uniform cube: hyper_cube(-1, 1), refine_global(r) (poisson_common.h:62-64, bmop_common.h:119), zero Dirichlet
 * values on the boundary, coefficient 1 / (0.05 + 2 |x|^2) (poisson_common.h:149-151), dst = 0.1 and N_ITERATIONS = 100
 * x { swap; vmult } (bmop-cpu.cc:137-155).  BASELINE.json configs[0] is `bmop-cpu-2d-p2 5`.
 *
 * The reference's mesh, dof numbering, quadrature and shape tables come from deal.II, which is not available: this
 * driver takes them from the build's deal.II stand-in (mfgpu_mesh_*: host-only code of libmfgpu.so, what
 * Triangulation + DoFHandler + FEValues hand over), and applies the operator with oracle/cpu_ref.c -- no GPU code
 * runs.  DIMENSION and DEGREE_FE are compile-time, as in the reference (bmop-cpu.cc:47-57). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "../include/mfgpu.h"

#ifndef DIMENSION
#define DIMENSION 2
#endif
#ifndef DEGREE_FE
#define DEGREE_FE 2
#endif
#define N_ITERATIONS 100

int cpu_ref_bmop(int dim, int n, uint32_t n_dofs, const uint32_t *l2g, const double *coef, const double *jxw,
                 const double *j0, const double *sv, const double *sg, const uint32_t *constrained, uint32_t ncon,
                 const uint32_t *color_off, int ncolors, const uint32_t *cell_order, double *a, double *b, double init,
                 int n_iter);

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int run(int r) {
  const int dim = DIMENSION, n = DEGREE_FE + 1;
  uint32_t per_dir[3];
  for (int d = 0; d < dim; ++d) per_dir[d] = 1u << r;
  mfgpu_mesh *mesh = NULL;
  mfgpu_desc desc;
  if (mfgpu_mesh_create_uniform(dim, DEGREE_FE, per_dir, -1.0, 1.0, 0, per_dir[dim - 1], MFGPU_F64, &mesh) ||
      mfgpu_mesh_desc(mesh, &desc)) {
    fprintf(stderr, "mesh setup failed: %s\n", mfgpu_last_error());
    return 1;
  }
  int nd = 1;
  for (int d = 0; d < dim; ++d) nd *= n;
  const size_t nq = (size_t)desc.n_cells * nd;
  /* setup_system: coefficient at the quadrature points (laplace_operator_cpu.cc:95-117) */
  double *coef = (double *)malloc(nq * sizeof(double));
  const double *qp = (const double *)desc.quadrature_points;
  for (size_t q = 0; q < nq; ++q) {
    double r2 = 0.0;
    for (int d = 0; d < dim; ++d) r2 += qp[q * dim + d] * qp[q * dim + d];
    coef[q] = 1.0 / (0.05 + 2.0 * r2);
  }
  /* cells of one colour share no dof: parity colouring of the lexicographic cell order (stands in for deal.II's
   * partition_color scheme, laplace_operator_cpu.cc:51-52) */
  const int ncolors = 1 << dim;
  uint32_t *color_off = (uint32_t *)calloc((size_t)ncolors + 1, sizeof(uint32_t));
  uint32_t *cell_order = (uint32_t *)malloc((size_t)desc.n_cells * sizeof(uint32_t));
  uint32_t *color = (uint32_t *)malloc((size_t)desc.n_cells * sizeof(uint32_t));
  for (uint32_t c = 0; c < desc.n_cells; ++c) {
    uint32_t idx = c, col = 0;
    for (int d = 0; d < dim; ++d) {
      col |= ((idx % per_dir[d]) & 1u) << d;
      idx /= per_dir[d];
    }
    color[c] = col;
    color_off[col + 1]++;
  }
  for (int k = 0; k < ncolors; ++k) color_off[k + 1] += color_off[k];
  uint32_t *fill = (uint32_t *)malloc((size_t)ncolors * sizeof(uint32_t));
  for (int k = 0; k < ncolors; ++k) fill[k] = color_off[k];
  for (uint32_t c = 0; c < desc.n_cells; ++c) cell_order[fill[color[c]]++] = c;
  double *a = (double *)malloc((size_t)desc.n_dofs * sizeof(double));
  double *b = (double *)calloc((size_t)desc.n_dofs, sizeof(double));
  /* solve (bmop-cpu.cc:137-155): the Timer covers the initial value and the 100 applies */
  const double t0 = now_s();
  cpu_ref_bmop(dim, n, desc.n_dofs, desc.loc2glob, coef, (const double *)desc.JxW, (const double *)desc.inv_jac,
               (const double *)desc.shape_values, (const double *)desc.shape_gradients, desc.constrained_dofs,
               desc.n_constrained, color_off, ncolors, cell_order, a, b, 0.1, N_ITERATIONS);
  const double wall = now_s() - t0;
  printf("%d\t%d\t%u\t%g\n", dim, DEGREE_FE, desc.n_dofs, wall / N_ITERATIONS);
  free(coef);
  free(color_off);
  free(cell_order);
  free(color);
  free(fill);
  free(a);
  free(b);
  mfgpu_mesh_destroy(mesh);
  return 0;
}

int main(int argc, char **argv) {
  int max_refinement = 1, min_refinement = 0;
  if (argc > 1) max_refinement = atoi(argv[1]);
  if (argc > 2) min_refinement = atoi(argv[2]);
  for (int r = min_refinement; r <= max_refinement; r++)
    if (run(r)) return 1;
  return 0;
}
