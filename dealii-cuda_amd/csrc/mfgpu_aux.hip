// what a CG caller (poisson.cu:223-260) needs besides vmult.
//   N1  inverse diagonal of the operator (LaplaceOperatorGpu::compute_diagonal,
//       laplace_operator_gpu.h:355-418) and set_constrained_values (constraint_handler_gpu.cu:126-137)
//   N2  GpuVector BLAS-1 and reductions (gpu_vec.cu:222-617)
// All kernels are plain streaming kernels (bound: HBM); reductions are two-stage and deterministic
// (fixed grid, fixed summation order) with wave64 shuffles -- the reference's warp-synchronous tail
// (gpu_vec.cu:413-440) assumes 32 lanes and finishes with one atomic per block.
#include <hip/hip_runtime.h>

#include <mutex>

#include "mfgpu_cell.h"
#include "mfgpu_kernels.h"

namespace mfgpu {

// resolve_hanging_nodes_shmem<TRANSPOSE> on ONE cell's local vector in LDS (x, then y, then z;
// hanging_nodes.cuh:767-777); mask is uniform over the workgroup; ends with a barrier when mask != 0
template <int dim, int n, typename T>
__device__ __forceinline__ void hn_transpose_local(T *loc, const T *Wl, unsigned mask, int tid) {
  constexpr int n2 = n * n, nd = (dim == 3) ? n2 * n : n2, P = nd / n;
  if (!mask) return;
  T v[n];
  bool type;
  const int pa = (dim == 3) ? tid % n : tid, pb = (dim == 3) ? tid / n : 0;
  const bool on = tid < P;
  if (dim == 3) {
    if (on && hn_flag3<n, 0>(mask, pa, pb, type)) {
      lds_load<n>(loc + n * pa + n2 * pb, 1, v);
      hn_pencil<n, T, true>(Wl, type, v);
      lds_store<n>(loc + n * pa + n2 * pb, 1, v);
    }
    __syncthreads();
    if (on && hn_flag3<n, 1>(mask, pb, pa, type)) {
      lds_load<n>(loc + pa + n2 * pb, n, v);
      hn_pencil<n, T, true>(Wl, type, v);
      lds_store<n>(loc + pa + n2 * pb, n, v);
    }
    __syncthreads();
    if (on && hn_flag3<n, 2>(mask, pa, pb, type)) {
      lds_load<n>(loc + pa + n * pb, n2, v);
      hn_pencil<n, T, true>(Wl, type, v);
      lds_store<n>(loc + pa + n * pb, n2, v);
    }
  } else {
    if (on && hn_flag2<n, 0>(mask, pa, type)) {
      lds_load<n>(loc + n * pa, 1, v);
      hn_pencil<n, T, true>(Wl, type, v);
      lds_store<n>(loc + n * pa, 1, v);
    }
    __syncthreads();
    if (on && hn_flag2<n, 1>(mask, pa, type)) {
      lds_load<n>(loc + pa, n, v);
      hn_pencil<n, T, true>(Wl, type, v);
      lds_store<n>(loc + pa, n, v);
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// N1: diagonal
// ---------------------------------------------------------------------------------------------
// The reference applies the cell kernel to every local unit vector and keeps entry i of the i-th
// result (DiagonalLocalOperator, laplace_operator_gpu.h:355-399), nd kernel applications per cell.
// The same number is K_ii = sum_q c_q |grad phi_i(x_q)|^2, c = a J0^2 JxW (the folded coefficient of
// the apply kernel), and with tensor-product shape functions
//   |grad phi_i(x_q)|^2 = sum_d G2[i_d][q_d] prod_{e != d} S2[i_e][q_e],  S2 = S.^2, G2 = G.^2,
// evaluated directly here (setup-time kernel: nd * nd terms per cell).  The local diagonal VECTOR is
// then distributed like any cell result: hanging-node resolution TRANSPOSE, add into the global vector
// (distribute_local_to_global, fee_gpu.cuh:346-363).  One workgroup per batch; adds are float/double
// atomics in global memory (order of the <= 8 contributions per dof is not fixed: last-bit differences
// between runs are possible, which a Jacobi / Chebyshev diagonal does not care about).
template <int dim, int n, typename T>
__global__ void __launch_bounds__(256)
diag_kernel(T *diag, const uint32_t *batch_cell_off, const uint32_t *batch_dof_off, const uint32_t *bdofs,
            const uint16_t *lmap, const T *coef, const uint32_t *cmask, const T *hn_weights, const T *tab2) {
  constexpr int n2 = n * n, nd = (dim == 3) ? n2 * n : n2;
  __shared__ T S2[n2], G2[n2], Wl[n2], loc[nd], cf[nd];
  const int tid = threadIdx.x;
  for (int t = tid; t < n2; t += 256) {
    S2[t] = tab2[t];
    G2[t] = tab2[n2 + t];
    Wl[t] = hn_weights ? hn_weights[t] : T(0);
  }
  const uint32_t b = blockIdx.x;
  const uint32_t c0 = batch_cell_off[b], c1 = batch_cell_off[b + 1], d0 = batch_dof_off[b];
  for (uint32_t c = c0; c < c1; ++c) {
    __syncthreads();
    for (int i = tid; i < nd; i += 256) cf[i] = coef[(size_t)c * nd + i];
    __syncthreads();
    for (int i = tid; i < nd; i += 256) {
      const int ix = i % n, iy = (i / n) % n, iz = i / n2;
      T sum = T(0);
      if (dim == 3) {
        for (int qz = 0; qz < n; ++qz)
          for (int qy = 0; qy < n; ++qy) {
            const T syz = S2[iy * n + qy] * S2[iz * n + qz];
            const T gyz = G2[iy * n + qy] * S2[iz * n + qz] + S2[iy * n + qy] * G2[iz * n + qz];
            for (int qx = 0; qx < n; ++qx)
              sum += cf[qx + n * qy + n2 * qz] * (G2[ix * n + qx] * syz + S2[ix * n + qx] * gyz);
          }
      } else {
        for (int qy = 0; qy < n; ++qy)
          for (int qx = 0; qx < n; ++qx)
            sum += cf[qx + n * qy] * (G2[ix * n + qx] * S2[iy * n + qy] + S2[ix * n + qx] * G2[iy * n + qy]);
      }
      loc[i] = sum;
    }
    __syncthreads();
    hn_transpose_local<dim, n, T>(loc, Wl, cmask ? cmask[c] : 0u, tid);
    for (int i = tid; i < nd; i += 256) {
      const uint32_t g = bdofs[d0 + lmap[(size_t)c * nd + i]];
      if (!(g >> 31)) atomicAdd(diag + g, loc[i]);  // constrained rows are set by set_values_kernel
    }
  }
}

// General-geometry variant (no MFGPU_UNIFORM_J0, 3D): K_ii = sum_q ghat_i(q)^T M(q) ghat_i(q) with the folded
// symmetric metric M = a JxW J J^T of apply_batches_g ([cell][entry][q], entries 00 01 02 11 12 22) and
// ghat_i(q) = (G[ix][qx] S[iy][qy] S[iz][qz], S G S, S S G).  tab2 here holds the UNSQUARED tables [S | G].
template <int n, typename T>
__global__ void __launch_bounds__(256)
diag_general_kernel(T *diag, const uint32_t *batch_cell_off, const uint32_t *batch_dof_off, const uint32_t *bdofs,
                    const uint16_t *lmap, const T *metric, const uint32_t *cmask, const T *hn_weights, const T *tab) {
  constexpr int n2 = n * n, nd = n2 * n;
  __shared__ T S[n2], G[n2], Wl[n2], loc[nd], m[6 * nd];
  const int tid = threadIdx.x;
  for (int t = tid; t < n2; t += 256) {
    S[t] = tab[t];
    G[t] = tab[n2 + t];
    Wl[t] = hn_weights ? hn_weights[t] : T(0);
  }
  const uint32_t b = blockIdx.x;
  const uint32_t c0 = batch_cell_off[b], c1 = batch_cell_off[b + 1], d0 = batch_dof_off[b];
  for (uint32_t c = c0; c < c1; ++c) {
    __syncthreads();
    for (int i = tid; i < 6 * nd; i += 256) m[i] = metric[(size_t)c * (6 * nd) + i];
    __syncthreads();
    for (int i = tid; i < nd; i += 256) {
      const int ix = i % n, iy = (i / n) % n, iz = i / n2;
      T sum = T(0);
      for (int qz = 0; qz < n; ++qz)
        for (int qy = 0; qy < n; ++qy) {
          const T syz = S[iy * n + qy] * S[iz * n + qz];
          const T gy0 = G[iy * n + qy] * S[iz * n + qz];
          const T gz0 = S[iy * n + qy] * G[iz * n + qz];
          for (int qx = 0; qx < n; ++qx) {
            const int q = qx + n * qy + n2 * qz;
            const T gx = G[ix * n + qx] * syz, gy = S[ix * n + qx] * gy0, gz = S[ix * n + qx] * gz0;
            sum += m[q] * gx * gx + m[3 * nd + q] * gy * gy + m[5 * nd + q] * gz * gz +
                   T(2) * (m[nd + q] * gx * gy + m[2 * nd + q] * gx * gz + m[4 * nd + q] * gy * gz);
          }
        }
      loc[i] = sum;
    }
    __syncthreads();
    hn_transpose_local<3, n, T>(loc, Wl, cmask ? cmask[c] : 0u, tid);
    for (int i = tid; i < nd; i += 256) {
      const uint32_t g = bdofs[d0 + lmap[(size_t)c * nd + i]];
      if (!(g >> 31)) atomicAdd(diag + g, loc[i]);
    }
  }
}

template <typename T>
hipError_t diag_general_launch(int n, T *diag, uint32_t n_batches, const uint32_t *batch_cell_off,
                               const uint32_t *batch_dof_off, const uint32_t *bdofs, const uint16_t *lmap,
                               const T *metric, const uint32_t *cmask, const T *hn_weights, const T *tab,
                               hipStream_t st) {
  if (n_batches == 0) return hipSuccess;
#define DG_CASE(N)                                                                                           \
  case N:                                                                                                    \
    hipLaunchKernelGGL((diag_general_kernel<N, T>), dim3(n_batches), dim3(256), 0, st, diag, batch_cell_off, \
                       batch_dof_off, bdofs, lmap, metric, cmask, hn_weights, tab);                          \
    break;
  switch (n) {
    DG_CASE(2) DG_CASE(3) DG_CASE(4) DG_CASE(5) DG_CASE(6) DG_CASE(7)
    default: return hipErrorInvalidValue;
  }
#undef DG_CASE
  return hipGetLastError();
}

// the same in 2D: metric[cell][e][q], e = xx, xy, yy
template <int n, typename T>
__global__ void __launch_bounds__(256)
diag_general2_kernel(T *diag, const uint32_t *batch_cell_off, const uint32_t *batch_dof_off, const uint32_t *bdofs,
                     const uint16_t *lmap, const T *metric, const uint32_t *cmask, const T *hn_weights, const T *tab) {
  constexpr int nd = n * n;
  __shared__ T S[nd], G[nd], Wl[nd], loc[nd], m[3 * nd];
  const int tid = threadIdx.x;
  for (int t = tid; t < nd; t += 256) {
    S[t] = tab[t];
    G[t] = tab[nd + t];
    Wl[t] = hn_weights ? hn_weights[t] : T(0);
  }
  const uint32_t b = blockIdx.x;
  const uint32_t c0 = batch_cell_off[b], c1 = batch_cell_off[b + 1], d0 = batch_dof_off[b];
  for (uint32_t c = c0; c < c1; ++c) {
    __syncthreads();
    for (int i = tid; i < 3 * nd; i += 256) m[i] = metric[(size_t)c * (3 * nd) + i];
    __syncthreads();
    for (int i = tid; i < nd; i += 256) {
      const int ix = i % n, iy = i / n;
      T sum = T(0);
      for (int qy = 0; qy < n; ++qy)
        for (int qx = 0; qx < n; ++qx) {
          const int q = qx + n * qy;
          const T gx = G[ix * n + qx] * S[iy * n + qy], gy = S[ix * n + qx] * G[iy * n + qy];
          sum += m[q] * gx * gx + m[2 * nd + q] * gy * gy + T(2) * m[nd + q] * gx * gy;
        }
      loc[i] = sum;
    }
    __syncthreads();
    hn_transpose_local<2, n, T>(loc, Wl, cmask ? cmask[c] : 0u, tid);
    for (int i = tid; i < nd; i += 256) {
      const uint32_t g = bdofs[d0 + lmap[(size_t)c * nd + i]];
      if (!(g >> 31)) atomicAdd(diag + g, loc[i]);
    }
  }
}

template <typename T>
hipError_t diag_general2_launch(int n, T *diag, uint32_t n_batches, const uint32_t *batch_cell_off,
                                const uint32_t *batch_dof_off, const uint32_t *bdofs, const uint16_t *lmap,
                                const T *metric, const uint32_t *cmask, const T *hn_weights, const T *tab,
                                hipStream_t st) {
  if (n_batches == 0) return hipSuccess;
#define DG2_CASE(N)                                                                                           \
  case N:                                                                                                     \
    hipLaunchKernelGGL((diag_general2_kernel<N, T>), dim3(n_batches), dim3(256), 0, st, diag, batch_cell_off, \
                       batch_dof_off, bdofs, lmap, metric, cmask, hn_weights, tab);                           \
    break;
  switch (n) {
    DG2_CASE(2) DG2_CASE(3) DG2_CASE(4) DG2_CASE(5) DG2_CASE(6) DG2_CASE(7)
    default: return hipErrorInvalidValue;
  }
#undef DG2_CASE
  return hipGetLastError();
}

template <typename T>
__global__ void set_values_kernel(T *v, const uint32_t *idx, uint32_t n, T value) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[idx[i]] = value;
}

template <typename T>
hipError_t diag_launch(int dim, int n, T *diag, uint32_t n_batches, const uint32_t *batch_cell_off,
                       const uint32_t *batch_dof_off, const uint32_t *bdofs, const uint16_t *lmap, const T *coef,
                       const uint32_t *cmask, const T *hn_weights, const T *tab2, hipStream_t st) {
  if (n_batches == 0) return hipSuccess;
#define DIAG_CASE(D, N)                                                                                      \
  case D * 10 + N:                                                                                           \
    hipLaunchKernelGGL((diag_kernel<D, N, T>), dim3(n_batches), dim3(256), 0, st, diag, batch_cell_off,      \
                       batch_dof_off, bdofs, lmap, coef, cmask, hn_weights, tab2);                           \
    break;
  switch (dim * 10 + n) {
    DIAG_CASE(2, 2) DIAG_CASE(2, 3) DIAG_CASE(2, 4) DIAG_CASE(2, 5) DIAG_CASE(2, 6) DIAG_CASE(2, 7)
    DIAG_CASE(3, 2) DIAG_CASE(3, 3) DIAG_CASE(3, 4) DIAG_CASE(3, 5) DIAG_CASE(3, 6) DIAG_CASE(3, 7)
    default: return hipErrorInvalidValue;
  }
#undef DIAG_CASE
  return hipGetLastError();
}

template <typename T>
hipError_t set_values_launch(T *v, const uint32_t *idx, uint32_t n, T value, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(set_values_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, st, v, idx, n, value);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// N2: vector operations
// ---------------------------------------------------------------------------------------------
enum VecOp { OP_SADD, OP_EQU, OP_SCALE, OP_DIVIDE, OP_INVERT, OP_MUL };

constexpr unsigned kVecBlocks = 2048;  // 256 CUs x 8 resident blocks of 256 threads, grid-stride

template <int OP, typename T>
__global__ void __launch_bounds__(256) vec_map_kernel(T *__restrict__ v, const T *__restrict__ w, T s, T a, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (OP == OP_SADD) v[i] = s * v[i] + a * w[i];   // sadd        gpu_vec.cu:222-240,308-314
    if (OP == OP_EQU) v[i] = a * w[i];               // equ         gpu_vec.cu:257-266,346-352
    if (OP == OP_SCALE) v[i] = v[i] * w[i];          // scale       gpu_vec.cu:244-254,320-325
    if (OP == OP_DIVIDE) v[i] = v[i] / w[i];         // operator/=  gpu_vec.cu:244-254,328-333
    if (OP == OP_INVERT) v[i] = T(1) / v[i];         // invert      gpu_vec.cu:294-303,336-342
    if (OP == OP_MUL) v[i] = a * v[i];               // operator*=  gpu_vec.cu:269-279,357-362
  }
}

template <typename T>
hipError_t vec_map_launch(int op, T *v, const T *w, T s, T a, size_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  size_t blocks = (n + 255) / 256;
  if (blocks > kVecBlocks) blocks = kVecBlocks;
#define MAP_CASE(OP)                                                                                         \
  case OP:                                                                                                   \
    hipLaunchKernelGGL((vec_map_kernel<OP, T>), dim3((unsigned)blocks), dim3(256), 0, st, v, w, s, a, n);    \
    break;
  switch (op) {
    MAP_CASE(OP_SADD) MAP_CASE(OP_EQU) MAP_CASE(OP_SCALE) MAP_CASE(OP_DIVIDE) MAP_CASE(OP_INVERT) MAP_CASE(OP_MUL)
    default: return hipErrorInvalidValue;
  }
#undef MAP_CASE
  return hipGetLastError();
}

enum RedOp { RED_DOT, RED_ADD_AND_DOT, RED_NONZERO };

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

// stage 1: one partial per block, accumulated in double (also for float vectors), fixed order
template <int OP, typename T>
__global__ void __launch_bounds__(256)
vec_reduce_kernel(double *__restrict__ partial, T *__restrict__ v, const T *__restrict__ x, const T *__restrict__ w,
                  T a, size_t n) {
  __shared__ double red[4];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (OP == RED_DOT) acc += (double)v[i] * (double)w[i];  // operator*, l2_norm   gpu_vec.cu:473-510,542-563,367-369
    if (OP == RED_ADD_AND_DOT) {                             // add_and_dot          gpu_vec.cu:568-617
      const T t = v[i] + a * x[i];
      v[i] = t;
      acc += (double)t * (double)w[i];
    }
    if (OP == RED_NONZERO) acc += (v[i] != T(0)) ? 1.0 : 0.0;  // all_zero           gpu_vec.cu:445-470,512-540
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// stage 2: one block sums the partials in a fixed order
__global__ void __launch_bounds__(256) vec_reduce_final(double *__restrict__ result, const double *__restrict__ partial,
                                                         unsigned np) {
  __shared__ double red[4];
  double acc = 0.0;
  for (unsigned i = threadIdx.x; i < np; i += 256) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *result = (red[0] + red[1]) + (red[2] + red[3]);
}

// scratch of the reductions: kVecBlocks partials + the result, one set per device, allocated on first
// use and kept (reductions are serialised by the mutex: they end in a blocking copy of the result anyway)
static std::mutex g_red_mutex;
static double *g_red_buf[64] = {};

template <typename T>
hipError_t vec_reduce_launch(int op, T *v, const T *x, const T *w, T a, size_t n, hipStream_t st, double *out) {
  std::lock_guard<std::mutex> lock(g_red_mutex);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!g_red_buf[dev]) {
    e = hipMalloc((void **)&g_red_buf[dev], (kVecBlocks + 1) * sizeof(double));
    if (e != hipSuccess) return e;
  }
  double *partial = g_red_buf[dev], *result = partial + kVecBlocks;
  size_t blocks = (n + 255) / 256;
  if (blocks > kVecBlocks) blocks = kVecBlocks;
  if (blocks == 0) blocks = 1;
#define RED_CASE(OP)                                                                                         \
  case OP:                                                                                                   \
    hipLaunchKernelGGL((vec_reduce_kernel<OP, T>), dim3((unsigned)blocks), dim3(256), 0, st, partial, v, x, w, a, n); \
    break;
  switch (op) {
    RED_CASE(RED_DOT) RED_CASE(RED_ADD_AND_DOT) RED_CASE(RED_NONZERO)
    default: return hipErrorInvalidValue;
  }
#undef RED_CASE
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(vec_reduce_final, dim3(1), dim3(256), 0, st, result, partial, (unsigned)blocks);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(out, result, sizeof(double), hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(st);
}

#define INST(T)                                                                                              \
  template hipError_t diag_launch<T>(int, int, T *, uint32_t, const uint32_t *, const uint32_t *, const uint32_t *, \
                                     const uint16_t *, const T *, const uint32_t *, const T *, const T *, hipStream_t); \
  template hipError_t diag_general_launch<T>(int, T *, uint32_t, const uint32_t *, const uint32_t *, const uint32_t *, \
                                             const uint16_t *, const T *, const uint32_t *, const T *, const T *, \
                                             hipStream_t);                                                   \
  template hipError_t diag_general2_launch<T>(int, T *, uint32_t, const uint32_t *, const uint32_t *, const uint32_t *, \
                                              const uint16_t *, const T *, const uint32_t *, const T *, const T *, \
                                              hipStream_t);                                                  \
  template hipError_t set_values_launch<T>(T *, const uint32_t *, uint32_t, T, hipStream_t);                  \
  template hipError_t vec_map_launch<T>(int, T *, const T *, T, T, size_t, hipStream_t);                      \
  template hipError_t vec_reduce_launch<T>(int, T *, const T *, const T *, T, size_t, hipStream_t, double *);
INST(double)
INST(float)

}  // namespace mfgpu
