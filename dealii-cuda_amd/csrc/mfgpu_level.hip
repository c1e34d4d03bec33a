// Level operator of a multigrid hierarchy with refinement edges: LaplaceOperatorGpu::reinit(dof_handler,
// mg_constrained_dofs, level) and the interface ("edge") matrices vmult_interface_down / vmult_interface_up
// (reference laplace_operator_gpu.h:154-186, 306-352) that deal.II's Multigrid applies on adaptively refined meshes
// (mg.set_edge_matrices, poisson_mg.cu:375).
//
// The reference keeps ONE matrix-free structure per level without constraints and brackets its cell loop with index
// kernels of ConstraintHandlerGpu (save / zero / copy_edge_values / load).  Here the operator fuses its constrained
// rows into the cell loop, so a level holds two operators over the same level mesh:
//   A  : constrained rows = the level's Dirichlet dofs AND its refinement-edge dofs (what vmult / the smoother see)
//   Ab : K itself, NO constrained rows -- the reference's cell loop runs on a structure without constraints (:174-176)
//        (built only if the level has edge dofs)
// and the interface matrices are compositions (C = Dirichlet + edge, E = edge):
//   down: dst = 0 except dst[E] = (K (src with C zeroed))[E]                                          (:306-330)
//   up  : dst = K (src restricted to E), then dst[C] = 0                                              (:332-352)
// An edge dof may also be a Dirichlet dof (a refinement edge that reaches the domain boundary): down then returns
// (K x)[E] on it and up includes src[E] there, as the reference does; identity rows in Ab would lose both.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "mfgpu_internal.h"

struct mfgpu_level {
  mfgpu_handle *A = nullptr, *Ab = nullptr;
  uint32_t *d_c = nullptr, *d_e = nullptr;  // C = Dirichlet + edge, E = edge (device index lists)
  uint32_t n_c = 0, n_e = 0, n_dofs = 0;
  int number_type = MFGPU_F64;
  void *tmp_x = nullptr, *tmp_y = nullptr;
};

// index pairs on the device: copy_to_mg / copy_from_mg (mg_transfer_matrix_free_gpu.cu:690-760, copy_indices)
struct mfgpu_index_pairs {
  uint32_t *d_dst = nullptr, *d_src = nullptr;
  uint32_t n = 0;
};

namespace {

template <typename T>
__global__ void copy_pairs_kernel(T *dst, const T *src, const uint32_t *di, const uint32_t *si, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[di[i]] = src[si[i]];
}

template <typename T>
__global__ void set_indexed_kernel(T *v, const uint32_t *idx, uint32_t n, T value) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[idx[i]] = value;
}
template <typename T>
__global__ void copy_indexed_kernel(T *dst, const T *src, const uint32_t *idx, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[idx[i]] = src[idx[i]];
}

template <typename T>
int interface_typed(mfgpu_level *L, bool down, T *dst, const T *src, hipStream_t st) {
  const size_t bytes = (size_t)L->n_dofs * sizeof(T);
  T *x = (T *)L->tmp_x, *y = (T *)L->tmp_y;
  const unsigned gc = (L->n_c + 255) / 256, ge = (L->n_e + 255) / 256;
  if (L->n_e == 0) {  // no refinement edge on this level: both matrices are zero
    if (hipMemsetAsync(dst, 0, bytes, st) != hipSuccess) return MFGPU_EHIP;
    return 0;
  }
  if (down) {
    // x = src with C zeroed (constraint_handler.save_constrained_values, :315); y = K x; dst = 0, dst[E] = y[E]
    if (hipMemcpyAsync(x, src, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return MFGPU_EHIP;
    hipLaunchKernelGGL(set_indexed_kernel<T>, dim3(gc), dim3(256), 0, st, x, L->d_c, L->n_c, T(0));
    int rc = mfgpu_vmult(L->Ab, y, x, st);
    if (rc) return rc;
    if (hipMemsetAsync(dst, 0, bytes, st) != hipSuccess) return MFGPU_EHIP;
    hipLaunchKernelGGL(copy_indexed_kernel<T>, dim3(ge), dim3(256), 0, st, dst, (const T *)y, L->d_e, L->n_e);
  } else {
    // x = 0 except the edge values of src (copy_edge_values, :343); dst = K x; dst[C] = 0 (:351)
    if (hipMemsetAsync(x, 0, bytes, st) != hipSuccess) return MFGPU_EHIP;
    hipLaunchKernelGGL(copy_indexed_kernel<T>, dim3(ge), dim3(256), 0, st, x, src, L->d_e, L->n_e);
    int rc = mfgpu_vmult(L->Ab, dst, x, st);
    if (rc) return rc;
    hipLaunchKernelGGL(set_indexed_kernel<T>, dim3(gc), dim3(256), 0, st, dst, L->d_c, L->n_c, T(0));
  }
  return hipGetLastError() == hipSuccess ? 0 : MFGPU_EHIP;
}

}  // namespace

extern "C" {

int mfgpu_level_create(const mfgpu_desc *desc, const uint32_t *edge_dofs, uint32_t n_edge, mfgpu_level **out) {
  using mfgpu::set_error;
  if (!desc || !out || (n_edge && !edge_dofs)) {
    set_error("mfgpu_level_create: null argument");
    return MFGPU_EINVAL;
  }
  if (desc->flags & MFGPU_HANGING_NODES) {
    set_error("mfgpu_level_create: level meshes have no hanging nodes (laplace_operator_gpu.h:174-176)");
    return MFGPU_EINVAL;
  }
  std::vector<uint32_t> e(edge_dofs, edge_dofs + n_edge), c(desc->constrained_dofs, desc->constrained_dofs + desc->n_constrained);
  for (uint32_t g : e)
    if (g >= desc->n_dofs) {
      set_error("mfgpu_level_create: edge dof out of range");
      return MFGPU_EINVAL;
    }
  std::sort(e.begin(), e.end());
  e.erase(std::unique(e.begin(), e.end()), e.end());
  c.insert(c.end(), e.begin(), e.end());
  std::sort(c.begin(), c.end());
  c.erase(std::unique(c.begin(), c.end()), c.end());
  mfgpu_level *L = new mfgpu_level();
  L->n_dofs = desc->n_dofs;
  L->number_type = desc->number_type;
  L->n_e = (uint32_t)e.size();
  L->n_c = (uint32_t)c.size();
  auto fail = [&](int rc) {
    mfgpu_level_destroy(L);
    return rc;
  };
  mfgpu_desc da = *desc;
  da.constrained_dofs = c.data();
  da.n_constrained = (uint32_t)c.size();
  int rc = mfgpu_create(&da, &L->A);
  if (rc) return fail(rc);
  if (L->n_e) {
    mfgpu_desc db = *desc;
    db.constrained_dofs = nullptr;
    db.n_constrained = 0;
    if ((rc = mfgpu_create(&db, &L->Ab))) return fail(rc);
    const size_t es = desc->number_type == MFGPU_F64 ? 8 : 4;
    if (hipMalloc((void **)&L->d_c, c.size() * 4) != hipSuccess || hipMalloc((void **)&L->d_e, e.size() * 4) != hipSuccess ||
        hipMalloc(&L->tmp_x, (size_t)L->n_dofs * es) != hipSuccess || hipMalloc(&L->tmp_y, (size_t)L->n_dofs * es) != hipSuccess ||
        hipMemcpy(L->d_c, c.data(), c.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(L->d_e, e.data(), e.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
      set_error("mfgpu_level_create: device allocation failed");
      return fail(MFGPU_ENOMEM);
    }
  }
  *out = L;
  return 0;
}

mfgpu_handle *mfgpu_level_operator(mfgpu_level *L) { return L ? L->A : nullptr; }

int mfgpu_level_vmult_interface_down(mfgpu_level *L, void *dst, const void *src, void *stream) {
  if (!L || !dst || !src || dst == src) {
    mfgpu::set_error("mfgpu_level_vmult_interface_down: null or aliasing argument");
    return MFGPU_EINVAL;
  }
  return L->number_type == MFGPU_F64 ? interface_typed<double>(L, true, (double *)dst, (const double *)src, (hipStream_t)stream)
                                     : interface_typed<float>(L, true, (float *)dst, (const float *)src, (hipStream_t)stream);
}

int mfgpu_level_vmult_interface_up(mfgpu_level *L, void *dst, const void *src, void *stream) {
  if (!L || !dst || !src || dst == src) {
    mfgpu::set_error("mfgpu_level_vmult_interface_up: null or aliasing argument");
    return MFGPU_EINVAL;
  }
  return L->number_type == MFGPU_F64 ? interface_typed<double>(L, false, (double *)dst, (const double *)src, (hipStream_t)stream)
                                     : interface_typed<float>(L, false, (float *)dst, (const float *)src, (hipStream_t)stream);
}

int mfgpu_index_pairs_create(const uint32_t *dst_idx, const uint32_t *src_idx, uint32_t n, mfgpu_index_pairs **out) {
  if (!out || (n && (!dst_idx || !src_idx))) {
    mfgpu::set_error("mfgpu_index_pairs_create: null argument");
    return MFGPU_EINVAL;
  }
  mfgpu_index_pairs *p = new mfgpu_index_pairs();
  p->n = n;
  if (n && (hipMalloc((void **)&p->d_dst, (size_t)n * 4) != hipSuccess || hipMalloc((void **)&p->d_src, (size_t)n * 4) != hipSuccess ||
            hipMemcpy(p->d_dst, dst_idx, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p->d_src, src_idx, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess)) {
    mfgpu::set_error("mfgpu_index_pairs_create: device allocation failed");
    mfgpu_index_pairs_destroy(p);
    return MFGPU_ENOMEM;
  }
  *out = p;
  return 0;
}

int mfgpu_vec_copy_pairs(const mfgpu_index_pairs *p, void *dst, const void *src, int number_type, void *stream) {
  if (!p || !dst || !src) {
    mfgpu::set_error("mfgpu_vec_copy_pairs: null argument");
    return MFGPU_EINVAL;
  }
  if (p->n == 0) return 0;
  const unsigned grid = (p->n + 255) / 256;
  if (number_type == MFGPU_F64)
    hipLaunchKernelGGL(copy_pairs_kernel<double>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (double *)dst,
                       (const double *)src, p->d_dst, p->d_src, p->n);
  else
    hipLaunchKernelGGL(copy_pairs_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float *)dst,
                       (const float *)src, p->d_dst, p->d_src, p->n);
  return hipGetLastError() == hipSuccess ? 0 : MFGPU_EHIP;
}

void mfgpu_index_pairs_destroy(mfgpu_index_pairs *p) {
  if (!p) return;
  hipFree(p->d_dst);
  hipFree(p->d_src);
  delete p;
}

void mfgpu_level_destroy(mfgpu_level *L) {
  if (!L) return;
  mfgpu_destroy(L->A);
  mfgpu_destroy(L->Ab);
  hipFree(L->d_c);
  hipFree(L->d_e);
  hipFree(L->tmp_x);
  hipFree(L->tmp_y);
  delete L;
}

}  // extern "C"
