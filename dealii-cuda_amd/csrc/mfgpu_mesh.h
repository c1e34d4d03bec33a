// Host-side mesh / DoF / constraint stand-in (see mfgpu_mesh.cpp).
#ifndef MFGPU_MESH_H
#define MFGPU_MESH_H

#include <array>
#include <vector>

#include "mfgpu_internal.h"

namespace mfgpu {

void gauss_01(int n, std::vector<double> &x, std::vector<double> &w);
void gll_01(int p, std::vector<double> &x);
void lagrange_eval(const std::vector<double> &nodes, double x, std::vector<double> &val,
                   std::vector<double> &der);

struct Mesh {
  int dim = 0, degree = 0, number_type = MFGPU_F64;
  bool general = false;  // inv_jac holds a full J^-1 per quadrature point (no MFGPU_UNIFORM_J0): ball meshes
  uint32_t n_dofs = 0, n_cells = 0;
  // structure the multigrid transfer setup needs: 0 full uniform cube (lexicographic cells, nper per direction),
  // 1 refinement tree in depth-first order with lexicographic children (ball), -1 anything else
  int mg_kind = -1;
  uint32_t nper[3] = {1, 1, 1};
  std::vector<double> nodes, xq, wq;  // 1D support points / Gauss points / weights on [0,1]
  std::vector<double> shape_values, shape_gradients, weights;
  std::vector<uint32_t> loc2glob, constraint_mask, constrained;
  std::vector<double> JxW, inv_jac, qpoints, dof_coords;
  std::vector<uint32_t> iface[2];
  std::vector<uint32_t> cell_levels;  // adaptive meshes: (level, cx, cy, cz) per cell
  // Number-typed copies when number_type == F32
  std::vector<unsigned char> t_JxW, t_inv_jac, t_qpoints, t_sv, t_sg;

  void init_tables();
  void finalize_typed();
  void fill_desc(mfgpu_desc &d) const;
};

int build_uniform(Mesh &M, const uint32_t *nper, double lo, double hi, uint32_t sb, uint32_t se);
int build_adaptive(Mesh &M, int n_ref, bool balance_vertices);
int build_ball(Mesh &M, int n_ref);
// (p+1)^dim dofs of every coarse cell and (2p+1)^dim dofs of its children's patch, lexicographic (mg_transfer_matrix_
// free_gpu.h:246 level_dof_indices); fine must be the global refinement of coarse
int mesh_transfer_patches(const Mesh &coarse, const Mesh &fine, std::vector<uint32_t> &coarse_cell_dofs,
                          std::vector<uint32_t> &fine_patch_dofs);
void default_prolongation_1d(int p, std::vector<double> &P1);
int build_from_tree_leaves(Mesh &M, int dim, std::vector<std::array<uint32_t, 4>> leaves);

}  // namespace mfgpu

struct mfgpu_mesh {
  mfgpu::Mesh mesh;
};

#endif
