// General-Jacobian cell loop in 2D (the reference's default geometry path, fee_gpu.cuh:235-241,
// 275-281, is dimension-generic): bmop / poisson on the 2D BALL domain and any 2D description without
// MFGPU_UNIFORM_J0.  Two-pass scatter mode, with or without hanging nodes.
//
// Per quadrature point the symmetric 2x2 metric M = a JxW J^-1 J^-T is folded at setup (fold_general2_kernel: 24 B
// per point instead of 32 + 8 + 8), so a cell computes
//   v = S_y S_x u;  g = (D_x v, D_y v);  f = M g;  out = S_x^T S_y^T (D_x^T f_x + D_y^T f_y)
// 2D problems of bmop's sizes are small (C1: 4 225 dofs) and launch-bound; this kernel keeps the batch data flow of
// apply_batches (gather once per batch dof, LDS accumulator, interior dofs stored, shared dofs to the halo buffer)
// but a plain one-thread-per-node cell phase: n*n threads per cell, every 1D contraction reads its operands from LDS
// (the reference's scheme, tensor_ops.cuh:84-116), six barrier-separated steps per chunk of 256 / n^2 cells.
#include <hip/hip_runtime.h>

#include "mfgpu_kernels.h"

namespace mfgpu {

namespace {

// hanging_nodes.cuh:38-50 (2D subset)
constexpr unsigned kTypeX = 1u << 0, kTypeY = 1u << 1, kFaceX = 1u << 3, kFaceY = 1u << 4;

template <int n, typename T, bool HN>
__global__ void __launch_bounds__(256)
apply_batches_g2(const ApplyArgs<T> A) {
  constexpr int nd = n * n, CH = 256 / nd, p = n - 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double *acc = reinterpret_cast<double *>(smem_raw);
  T *usrc = reinterpret_cast<T *>(acc + A.nb_max);
  T *X = usrc + A.nb_max;   // four scratch arrays of CH cells
  T *Y = X + CH * nd;
  T *FX = Y + CH * nd;
  T *FY = FX + CH * nd;
  T *St = FY + CH * nd;     // S[i*n+q], Dt[q*n+t], W[i*n+j] (broadcast reads)
  T *Dt = St + nd;
  T *Wt = Dt + nd;
  const int tid = threadIdx.x;
  for (int t = tid; t < nd; t += 256) {
    St[t] = A.tabS[t];
    Dt[t] = A.tabDt[t];
    Wt[t] = HN ? A.hn_weights[t] : T(0);
  }
  const int lc = tid / nd, q = tid - lc * nd, i = q % n, j = q / n;
  const bool lane_cell = lc < CH;
  T *x = X + lc * nd, *y = Y + lc * nd, *fx = FX + lc * nd, *fy = FY + lc * nd;

  for (uint32_t b = A.batch0 + blockIdx.x; b < A.batch_end; b += gridDim.x) {
    const uint32_t c0 = A.batch_cell_off[b], ncell = A.batch_cell_off[b + 1] - c0;
    const uint32_t d0 = A.batch_dof_off[b], nb = A.batch_dof_off[b + 1] - d0;
    const uint32_t nint = A.batch_nint[b];
    __syncthreads();  // the previous batch's scatter has read acc; the tables are staged
    // ---- gather (fee_gpu.cuh:323-331); constrained rows read as 0, the owning batch writes the identity row
    for (uint32_t t = tid; t < nb; t += 256) {
      const uint32_t g = A.bdofs[d0 + t];
      const T sv = A.src[g & 0x7fffffffu];
      const bool con = (g >> 31) != 0;
      usrc[t] = con ? T(0) : sv;
      acc[t] = 0.0;
      if (con && t < nint) {
        T *d = A.dst + (g & 0x7fffffffu);
        *d = A.add ? *d + sv : sv;
      }
    }
    __syncthreads();
    for (uint32_t base = 0; base < ncell; base += CH) {
      const bool act = lane_cell && base + lc < ncell;
      const size_t cell = (size_t)c0 + base + (act ? lc : 0);
      const uint16_t lm = A.lmap[cell * nd + q];
      unsigned mask = 0;
      if (HN && act) mask = A.cmask[cell];
      // hanging-node interpolation (hanging_nodes.cuh:698-758): along x on the line y = oy, then along y on x = ox
      const int oy = (mask & kTypeY) ? 0 : p, ox = (mask & kTypeX) ? 0 : p;
      auto hn_pass = [&](const T *in, int dir, bool transpose) -> T {
        const bool on = dir == 0 ? ((mask & kFaceY) && j == oy) : ((mask & kFaceX) && i == ox);
        if (!on) return in[q];
        const bool typ = (mask & (dir == 0 ? kTypeX : kTypeY)) != 0;
        const int r = dir == 0 ? i : j;
        T s = T(0);
        for (int k = 0; k < n; ++k) {
          // M[r][k] = typ ? W[r][k] : W[p-r][p-k];  transposed: M[k][r]
          const int a = transpose ? k : r, c = transpose ? r : k;
          const T w = typ ? Wt[a * n + c] : Wt[(p - a) * n + (p - c)];
          s += w * in[dir == 0 ? k + n * j : i + n * k];
        }
        return s;
      };
      if (act) x[q] = usrc[lm];
      __syncthreads();
      if (HN) {
        if (act) y[q] = hn_pass(x, 0, false);
        __syncthreads();
        if (act) x[q] = hn_pass(y, 1, false);
        __syncthreads();
      }
      // v = S_y S_x u
      if (act) {
        T s = T(0);
        for (int k = 0; k < n; ++k) s += St[k * n + i] * x[k + n * j];
        y[q] = s;
      }
      __syncthreads();
      if (act) {
        T s = T(0);
        for (int k = 0; k < n; ++k) s += St[k * n + j] * y[i + n * k];
        x[q] = s;
      }
      __syncthreads();
      // reference-cell gradient, metric, per point
      if (act) {
        T gx = T(0), gy = T(0);
        for (int t = 0; t < n; ++t) {
          gx += Dt[i * n + t] * x[t + n * j];
          gy += Dt[j * n + t] * x[i + n * t];
        }
        const T *m = A.coef + cell * (3 * nd) + q;  // [cell][e][q], e = xx, xy, yy
        const T mxx = m[0], mxy = m[nd], myy = m[2 * nd];
        fx[q] = mxx * gx + mxy * gy;
        fy[q] = mxy * gx + myy * gy;
      }
      __syncthreads();
      // w = D_x^T f_x + D_y^T f_y, then S_y^T, S_x^T
      if (act) {
        T s = T(0);
        for (int t = 0; t < n; ++t) s += Dt[t * n + i] * fx[t + n * j] + Dt[t * n + j] * fy[i + n * t];
        y[q] = s;
      }
      __syncthreads();
      if (act) {
        T s = T(0);
        for (int k = 0; k < n; ++k) s += St[j * n + k] * y[i + n * k];
        x[q] = s;
      }
      __syncthreads();
      T out = T(0);
      if (act) {
        for (int k = 0; k < n; ++k) out += St[i * n + k] * x[k + n * j];
      }
      if (HN) {  // transposed resolution (fee_gpu.cuh:349-351): y^T then x^T, the exact adjoint of the forward order
        __syncthreads();
        if (act) y[q] = out;
        __syncthreads();
        if (act) x[q] = hn_pass(y, 1, true);
        __syncthreads();
        if (act) out = hn_pass(x, 0, true);
      }
      if (act) atomicAdd(&acc[lm], (double)out);
      __syncthreads();
    }
    // ---- scatter (fee_gpu.cuh:346-363): interior dofs are this batch's alone and final; partial sums of the others
    // go to the batch's halo slots (pass 2)
    T *halo = A.halo + A.halo_off[b];
    for (uint32_t t = tid; t < nb; t += 256) {
      const uint32_t g = A.bdofs[d0 + t];
      if (t < nint) {
        if (!(g >> 31)) A.dst[g] = A.add ? A.dst[g] + (T)acc[t] : (T)acc[t];
      } else {
        halo[t - nint] = (T)acc[t];
      }
    }
  }
}

// M[cell][e][q] = a JxW (J^-1 J^-T)[e], e = xx, xy, yy, for the cell at plan position `cell`
template <typename T>
__global__ void fold_general2_kernel(T *M, const T *coef, const T *jxw, const T *jinv, const uint32_t *order,
                                     uint32_t n_cells, uint32_t nd) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_cells * nd) return;
  const uint32_t cell = (uint32_t)(i / nd), q = (uint32_t)(i - (size_t)cell * nd);
  const size_t s = (size_t)order[cell] * nd + q;
  const T *J = jinv + s * 4;
  const T a = coef[s] * jxw[s];
  T *m = M + (size_t)cell * (3 * nd) + q;
  m[0] = a * (J[0] * J[0] + J[1] * J[1]);
  m[nd] = a * (J[0] * J[2] + J[1] * J[3]);
  m[2 * nd] = a * (J[2] * J[2] + J[3] * J[3]);
}

template <int n, typename T>
size_t g2_lds_bytes(uint32_t nb_max) {
  constexpr int nd = n * n, CH = 256 / nd;
  return (size_t)nb_max * sizeof(double) + ((size_t)nb_max + 4 * CH * nd + 3 * nd) * sizeof(T);
}

template <int n, typename T>
hipError_t g2_run(const ApplyArgs<T> &a, bool hn, uint32_t grid, hipStream_t st, bool configure_only, size_t *lds_out,
                  int *occupancy) {
  const size_t lds = g2_lds_bytes<n, T>(a.nb_max);
  if (lds_out) *lds_out = lds;
  const void *fn = hn ? (const void *)apply_batches_g2<n, T, true> : (const void *)apply_batches_g2<n, T, false>;
  if (configure_only) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && occupancy) {
      e = hn ? hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, apply_batches_g2<n, T, true>, 256, lds)
             : hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, apply_batches_g2<n, T, false>, 256, lds);
    }
    return e;
  }
  if (hn)
    hipLaunchKernelGGL((apply_batches_g2<n, T, true>), dim3(grid), dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL((apply_batches_g2<n, T, false>), dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}

}  // namespace

template <typename T>
hipError_t g2_launch(int n, const ApplyArgs<T> &a, bool hn, uint32_t grid, hipStream_t st, bool configure_only,
                     size_t *lds_out, int *occupancy) {
  switch (n) {
    case 2: return g2_run<2, T>(a, hn, grid, st, configure_only, lds_out, occupancy);
    case 3: return g2_run<3, T>(a, hn, grid, st, configure_only, lds_out, occupancy);
    case 4: return g2_run<4, T>(a, hn, grid, st, configure_only, lds_out, occupancy);
    case 5: return g2_run<5, T>(a, hn, grid, st, configure_only, lds_out, occupancy);
    case 6: return g2_run<6, T>(a, hn, grid, st, configure_only, lds_out, occupancy);
    case 7: return g2_run<7, T>(a, hn, grid, st, configure_only, lds_out, occupancy);
    default: return hipErrorInvalidValue;
  }
}
template hipError_t g2_launch<double>(int, const ApplyArgs<double> &, bool, uint32_t, hipStream_t, bool, size_t *, int *);
template hipError_t g2_launch<float>(int, const ApplyArgs<float> &, bool, uint32_t, hipStream_t, bool, size_t *, int *);

template <typename T>
hipError_t fold_general2_launch(T *M, const T *coef, const T *jxw, const T *jinv, const uint32_t *order,
                                uint32_t n_cells, uint32_t nd, hipStream_t st) {
  const size_t tot = (size_t)n_cells * nd;
  hipLaunchKernelGGL(fold_general2_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, M, coef, jxw, jinv,
                     order, n_cells, nd);
  return hipGetLastError();
}
template hipError_t fold_general2_launch<double>(double *, const double *, const double *, const double *,
                                                 const uint32_t *, uint32_t, uint32_t, hipStream_t);
template hipError_t fold_general2_launch<float>(float *, const float *, const float *, const float *, const uint32_t *,
                                                uint32_t, uint32_t, hipStream_t);

}  // namespace mfgpu
