// Multigrid level transfer: MGTransferMatrixFreeGpu::prolongate / restrict_and_add
// (reference matrix_free_gpu/mg_transfer_matrix_free_gpu.cu:391-660) between two globally refined levels.
//
// Semantics (what the reference's weighted, atomically accumulated cell loops add up to):
//   prolongate        dst_fine    = P (src_coarse with the coarse level's Dirichlet dofs read as 0)        (:595-627)
//   restrict_and_add  dst_coarse += Z P^T src_fine,  Z = zero on the coarse level's Dirichlet dofs          (:631-660)
// with P the embedding of the coarse FE space into the fine one: per coarse cell the tensor product of the 1D
// matrix P1[(2p+1) x (p+1)] (the coarse cell's 1D basis at the 2p+1 nodes of its two children).
//
// Re-design for one pass each and no scratch vectors:
//  * every fine dof has ONE owner patch (the first coarse cell whose children list it; bit 31 of a fine patch entry
//    marks "not the owner").  Prolongation is consistent across patches (the spaces are conforming), so the owner
//    alone stores the value with a plain store: no dst = 0 pass, no 3^dim weights, no atomics (reference: :417-435,
//    :375-386).  Restriction counts every fine dof once for the same reason: non-owned entries read as 0.
//  * Dirichlet dofs of the coarse level are flagged in the coarse cell's dof list (bit 31): read as 0 in
//    prolongate (no src copy + set_mg_constrained_dofs, :607-608), skipped in restrict (no increment vector, :642-657).
//  * restriction accumulates into the coarse vector with hardware floating-point atomics (several coarse cells share
//    a coarse dof), as the reference does; prolongation is deterministic.
// One 256-thread workgroup per coarse cell at a time (grid-stride); the three 1D contractions go through LDS.
// Bound: HBM (one read of the fine vector + index lists); small next to the smoother's operator applies.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "mfgpu_internal.h"
#include "mfgpu_mesh.h"

struct mfgpu_transfer {
  int dim = 0, degree = 0, number_type = MFGPU_F64;
  uint32_t n_coarse_cells = 0, n_coarse_dofs = 0, n_fine_dofs = 0;
  uint32_t *d_coarse = nullptr, *d_fine = nullptr;  // [cells][(p+1)^dim], [cells][(2p+1)^dim]
  void *d_p1 = nullptr;                             // P1[(2p+1) * (p+1)], X-major
  bool covers_all = true;                           // every fine dof is listed by some patch
  size_t device_bytes = 0;
};

namespace mfgpu {

namespace {

#define HIP_TRY_T(expr)                                                            \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                \
      return e_ == hipErrorOutOfMemory ? MFGPU_ENOMEM : MFGPU_EHIP;                \
    }                                                                              \
  } while (0)

constexpr int ipow_c(int a, int e) { return e == 0 ? 1 : a * ipow_c(a, e - 1); }

// sizes of the intermediate arrays: after contracting directions 0..k-1 the array is nf^k x nc^(dim-k)
template <int dim, int p, typename T, bool RESTRICT>
__global__ void __launch_bounds__(256)
transfer_kernel(T *__restrict__ dst, const T *__restrict__ src, const uint32_t *__restrict__ coarse,
                const uint32_t *__restrict__ fine, const T *__restrict__ p1, uint32_t n_cells) {
  constexpr int nc = p + 1, nf = 2 * p + 1;
  constexpr int NC = ipow_c(nc, dim), NF = ipow_c(nf, dim);
  __shared__ T A[NF], B[NF], P[nf * nc];
  const int tid = threadIdx.x;
  for (int t = tid; t < nf * nc; t += 256) P[t] = p1[t];
  for (uint32_t cell = blockIdx.x; cell < n_cells; cell += gridDim.x) {
    const uint32_t *cd = coarse + (size_t)cell * NC, *fd = fine + (size_t)cell * NF;
    __syncthreads();
    if (!RESTRICT) {
      for (int t = tid; t < NC; t += 256) {
        const uint32_t g = cd[t];
        A[t] = (g >> 31) ? T(0) : src[g];  // set_mg_constrained_dofs(src, to_level - 1, 0), :607-608
      }
    } else {
      for (int t = tid; t < NF; t += 256) {
        const uint32_t g = fd[t];
        A[t] = (g >> 31) ? T(0) : src[g];  // a fine dof enters through its owner patch only
      }
    }
    __syncthreads();
    T *in = A, *out = B;
#pragma unroll
    for (int k = 0; k < dim; ++k) {
      // prolongate: direction k goes nc -> nf; directions < k are already nf long, directions > k still nc
      // restrict  : direction k goes nf -> nc; directions < k are already nc long, directions > k still nf
      const int lo = RESTRICT ? ipow_c(nc, k) : ipow_c(nf, k);            // product of the sizes below k
      const int hi = RESTRICT ? ipow_c(nf, dim - 1 - k) : ipow_c(nc, dim - 1 - k);  // ... above k
      const int n_in = RESTRICT ? nf : nc, n_out = RESTRICT ? nc : nf;
      const int total = lo * n_out * hi;
      for (int t = tid; t < total; t += 256) {
        const int a = t % lo, o = (t / lo) % n_out, b = t / (lo * n_out);
        const T *v = in + a + (size_t)b * lo * n_in;
        T s = T(0);
        for (int i = 0; i < n_in; ++i) s += (RESTRICT ? P[i * nc + o] : P[o * nc + i]) * v[i * lo];
        out[t] = s;
      }
      __syncthreads();
      T *tmp = in;
      in = out;
      out = tmp;
    }
    if (!RESTRICT) {
      for (int t = tid; t < NF; t += 256) {
        const uint32_t g = fd[t];
        if (!(g >> 31)) dst[g] = in[t];
      }
    } else {
      for (int t = tid; t < NC; t += 256) {
        const uint32_t g = cd[t];
        if (!(g >> 31)) atomicAdd(dst + g, in[t]);  // several coarse cells share the dof (:515-519)
      }
    }
  }
}

template <typename T, bool RESTRICT>
hipError_t launch(const mfgpu_transfer *t, T *dst, const T *src, hipStream_t st) {
  const uint32_t n = t->n_coarse_cells;
  if (n == 0) return hipSuccess;
  const unsigned grid = n < 8192u ? n : 8192u;
#define TR_CASE(D, PP)                                                                                           \
  case D * 10 + PP:                                                                                              \
    hipLaunchKernelGGL((transfer_kernel<D, PP, T, RESTRICT>), dim3(grid), dim3(256), 0, st, dst, src, t->d_coarse, \
                       t->d_fine, (const T *)t->d_p1, n);                                                         \
    break;
  switch (t->dim * 10 + t->degree) {
    TR_CASE(2, 1) TR_CASE(2, 2) TR_CASE(2, 3) TR_CASE(2, 4) TR_CASE(2, 5) TR_CASE(2, 6)
    TR_CASE(3, 1) TR_CASE(3, 2) TR_CASE(3, 3) TR_CASE(3, 4) TR_CASE(3, 5) TR_CASE(3, 6)
    default: return hipErrorInvalidValue;
  }
#undef TR_CASE
  return hipGetLastError();
}

template <typename T>
__global__ void zero_kernel(T *v, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) v[i] = T(0);
}

}  // namespace

// FE_Q(p) on Gauss-Lobatto nodes: P1[X * (p+1) + x] = phi_x(xi_X), xi_X the 2p+1 child nodes on the parent's [0,1]
void default_prolongation_1d(int p, std::vector<double> &P1) {
  std::vector<double> nodes, val, der;
  gll_01(p, nodes);
  const int nc = p + 1, nf = 2 * p + 1;
  P1.assign((size_t)nf * nc, 0.0);
  for (int X = 0; X < nf; ++X) {
    const double xi = X <= p ? 0.5 * nodes[X] : 0.5 + 0.5 * nodes[X - p];
    lagrange_eval(nodes, xi, val, der);
    for (int x = 0; x < nc; ++x) P1[(size_t)X * nc + x] = val[x];
  }
  // exact zeros / ones where a child node coincides with a parent node
  for (double &v : P1)
    if (std::fabs(v) < 1e-15) v = 0.0;
    else if (std::fabs(v - 1.0) < 1e-15) v = 1.0;
}

}  // namespace mfgpu

extern "C" {

int mfgpu_transfer_create(int dim, int degree, int number_type, uint32_t n_coarse_cells, const uint32_t *coarse_cell_dofs,
                          const uint32_t *fine_patch_dofs, uint32_t n_coarse_dofs, uint32_t n_fine_dofs,
                          const uint32_t *coarse_dirichlet, uint32_t n_coarse_dirichlet, const double *prolongation_1d,
                          mfgpu_transfer **out) {
  using namespace mfgpu;
  if (!out || (dim != 2 && dim != 3) || degree < 1 || degree > 6 || (n_coarse_cells && (!coarse_cell_dofs || !fine_patch_dofs)) ||
      (n_coarse_dirichlet && !coarse_dirichlet) || (number_type != MFGPU_F64 && number_type != MFGPU_F32) ||
      n_coarse_dofs >= (1u << 31) || n_fine_dofs >= (1u << 31)) {
    set_error("mfgpu_transfer_create: bad argument");
    return MFGPU_EINVAL;
  }
  const int nc = degree + 1, nf = 2 * degree + 1;
  const size_t NC = (size_t)ipow(nc, dim), NF = (size_t)ipow(nf, dim);
  std::vector<uint8_t> dir(n_coarse_dofs, 0), seen(n_fine_dofs, 0);
  for (uint32_t i = 0; i < n_coarse_dirichlet; ++i) {
    if (coarse_dirichlet[i] >= n_coarse_dofs) {
      set_error("mfgpu_transfer_create: Dirichlet dof out of range");
      return MFGPU_EINVAL;
    }
    dir[coarse_dirichlet[i]] = 1;
  }
  std::vector<uint32_t> cd((size_t)n_coarse_cells * NC), fd((size_t)n_coarse_cells * NF);
  for (size_t i = 0; i < cd.size(); ++i) {
    const uint32_t g = coarse_cell_dofs[i];
    if (g >= n_coarse_dofs) {
      set_error("mfgpu_transfer_create: coarse dof out of range");
      return MFGPU_EINVAL;
    }
    cd[i] = g | (dir[g] ? 0x80000000u : 0u);
  }
  size_t covered = 0;
  for (size_t i = 0; i < fd.size(); ++i) {
    const uint32_t g = fine_patch_dofs[i];
    if (g >= n_fine_dofs) {
      set_error("mfgpu_transfer_create: fine dof out of range");
      return MFGPU_EINVAL;
    }
    fd[i] = g | (seen[g] ? 0x80000000u : 0u);  // the first patch that lists a fine dof owns it
    if (!seen[g]) ++covered;
    seen[g] = 1;
  }
  std::vector<double> P1;
  if (prolongation_1d)
    P1.assign(prolongation_1d, prolongation_1d + (size_t)nf * nc);
  else
    default_prolongation_1d(degree, P1);
  mfgpu_transfer *t = new mfgpu_transfer();
  t->dim = dim;
  t->degree = degree;
  t->number_type = number_type;
  t->n_coarse_cells = n_coarse_cells;
  t->n_coarse_dofs = n_coarse_dofs;
  t->n_fine_dofs = n_fine_dofs;
  t->covers_all = covered == n_fine_dofs;
  auto fail = [&](int rc) {
    mfgpu_transfer_destroy(t);
    return rc;
  };
  std::vector<float> P1f(P1.begin(), P1.end());
  const void *p1src = number_type == MFGPU_F64 ? (const void *)P1.data() : (const void *)P1f.data();
  const size_t p1b = P1.size() * (number_type == MFGPU_F64 ? 8 : 4);
  if ((cd.size() && hipMalloc((void **)&t->d_coarse, cd.size() * 4) != hipSuccess) ||
      (fd.size() && hipMalloc((void **)&t->d_fine, fd.size() * 4) != hipSuccess) || hipMalloc(&t->d_p1, p1b) != hipSuccess) {
    set_error("mfgpu_transfer_create: device allocation failed");
    return fail(MFGPU_ENOMEM);
  }
  if ((cd.size() && hipMemcpy(t->d_coarse, cd.data(), cd.size() * 4, hipMemcpyHostToDevice) != hipSuccess) ||
      (fd.size() && hipMemcpy(t->d_fine, fd.data(), fd.size() * 4, hipMemcpyHostToDevice) != hipSuccess) ||
      hipMemcpy(t->d_p1, p1src, p1b, hipMemcpyHostToDevice) != hipSuccess) {
    set_error("mfgpu_transfer_create: upload failed");
    return fail(MFGPU_EHIP);
  }
  t->device_bytes = (cd.size() + fd.size()) * 4 + p1b;
  *out = t;
  return 0;
}

int mfgpu_transfer_create_from_meshes(const mfgpu_mesh *coarse, const mfgpu_mesh *fine, mfgpu_transfer **out) {
  using namespace mfgpu;
  if (!coarse || !fine || !out) {
    set_error("mfgpu_transfer_create_from_meshes: null argument");
    return MFGPU_EINVAL;
  }
  std::vector<uint32_t> cd, fd;
  int rc = mesh_transfer_patches(coarse->mesh, fine->mesh, cd, fd);
  if (rc) return rc;
  return mfgpu_transfer_create(coarse->mesh.dim, coarse->mesh.degree, fine->mesh.number_type, coarse->mesh.n_cells, cd.data(),
                               fd.data(), coarse->mesh.n_dofs, fine->mesh.n_dofs, coarse->mesh.constrained.data(),
                               (uint32_t)coarse->mesh.constrained.size(), nullptr, out);
}

int mfgpu_transfer_prolongate(mfgpu_transfer *t, void *dst_fine, const void *src_coarse, void *stream) {
  using namespace mfgpu;
  if (!t || !dst_fine || !src_coarse) {
    set_error("mfgpu_transfer_prolongate: null argument");
    return MFGPU_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  if (t->number_type == MFGPU_F64) {
    if (!t->covers_all) hipLaunchKernelGGL(zero_kernel<double>, dim3(2048), dim3(256), 0, st, (double *)dst_fine, (size_t)t->n_fine_dofs);
    HIP_TRY_T((launch<double, false>(t, (double *)dst_fine, (const double *)src_coarse, st)));
  } else {
    if (!t->covers_all) hipLaunchKernelGGL(zero_kernel<float>, dim3(2048), dim3(256), 0, st, (float *)dst_fine, (size_t)t->n_fine_dofs);
    HIP_TRY_T((launch<float, false>(t, (float *)dst_fine, (const float *)src_coarse, st)));
  }
  return 0;
}

int mfgpu_transfer_restrict_and_add(mfgpu_transfer *t, void *dst_coarse, const void *src_fine, void *stream) {
  using namespace mfgpu;
  if (!t || !dst_coarse || !src_fine) {
    set_error("mfgpu_transfer_restrict_and_add: null argument");
    return MFGPU_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  if (t->number_type == MFGPU_F64)
    HIP_TRY_T((launch<double, true>(t, (double *)dst_coarse, (const double *)src_fine, st)));
  else
    HIP_TRY_T((launch<float, true>(t, (float *)dst_coarse, (const float *)src_fine, st)));
  return 0;
}

size_t mfgpu_transfer_memory_consumption(const mfgpu_transfer *t) { return t ? t->device_bytes : 0; }

void mfgpu_transfer_destroy(mfgpu_transfer *t) {
  if (!t) return;
  hipFree(t->d_coarse);
  hipFree(t->d_fine);
  hipFree(t->d_p1);
  delete t;
}

}  // extern "C"
