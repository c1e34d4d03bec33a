// general (non-Cartesian) geometry, the reference's default code path when
// MATRIX_FREE_UNIFORM_MESH is not defined: a full inverse Jacobian J^-1[dim][dim] per quadrature point
// (fee_gpu.cuh:235-241 get_gradient: grad_d1 = sum_d2 J[d2][d1] ghat_d2;  :275-281 submit_gradient:
// out_d1 = (sum_d2 J[d1][d2] grad_d2) * JxW;  coefficient in between, laplace_operator_gpu.h:257-260).
// Per quadrature point that is  t = M ghat  with the SYMMETRIC  M = a JxW J J^T  (6 entries in 3D), which
// mfgpu_create folds once (fold_general_kernel): 48 B per point are streamed instead of the reference's
// 72 (J^-1) + 8 (JxW) + 8 (coefficient).
//
// 3D, two-pass scatter mode; hanging nodes through the same constraint passes as apply_batches_x (template HN).  The batch machinery is apply_batches_x's (persistent
// workgroups, XCD-aware batch ranges, aliased source/accumulator array, x-pencil index runs from global
// memory, next batch's loads in flight).  The cell pipeline needs all three reference-gradient components
// at the same point, so they go through LDS: 11 barrier-separated stages per chunk,
//   S_x | S_y | S_z, D_z -> Gz | D_x -> Gx | D_y -> Gy | t = M g (pointwise, linear layout)
//       | D_z^T tz | + D_x^T tx | + D_y^T ty, S_y^T | S_z^T | S_x^T -> accumulator
// with 4 scratch arrays per cell (W/R aliased, Gx, Gy, Gz) -> 2 workgroups per CU at p=4.
// Bound: HBM -- 48 B per quadrature point = 945 MB per vmult on the C2 mesh, against 157 MB of folded
// coefficient on the Cartesian path.
#include <hip/hip_runtime.h>

#include "mfgpu_cell.h"
#include "mfgpu_kernels.h"

namespace mfgpu {

template <typename U>
__device__ __forceinline__ U g_stream_load(const U *p) { return __builtin_nontemporal_load(p); }

template <int n>
__device__ __forceinline__ int gix_at(const uint32_t (&w)[(n + 1) / 2], int i) {
  return (int)((w[i >> 1] >> (16 * (i & 1))) & 0xffffu);
}

template <int n, typename T, bool HN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
apply_batches_g(const ApplyArgs<T> A, const Tables<T, n> tab) {
  constexpr int kBlock = 256;
  constexpr int kGU = (max_batch_dofs(kBlock) + kBlock - 1) / kBlock;
  constexpr int n2 = n * n, nd = n2 * n;
  constexpr int P = n2;
  constexpr int CH = kBlock / P;
  constexpr int CHND = CH * nd;
  constexpr int NW = (n + 1) / 2;
  constexpr int PF = (CHND + kBlock - 1) / kBlock;  // quadrature points per thread in the pointwise stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // gathered source values, then the accumulator: always double (ds_add_f32 is far slower than ds_add_f64 on
  // gfx950, see apply_batches_x)
  double *ua = reinterpret_cast<double *>(smem_raw);
  T *Wb = reinterpret_cast<T *>(ua + A.nb_max);  // w, later the result r (aliased: w is dead after the D_y stage)
  T *Gxb = Wb + CHND;
  T *Gyb = Gxb + CHND;
  T *Gzb = Gyb + CHND;
  T *Wl = Gzb + CHND;  // hanging-node weights (HN only)

  const int tid = threadIdx.x;
  uint32_t b, bstride, bend;
  {  // XCD-aware batch ranges: see apply_batches_x
    const uint32_t nbt = A.batch_end - A.batch0, G = gridDim.x;
    if (G >= 8 && nbt >= G) {
      const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
      const uint32_t q = G >> 3, rem = G & 7u;
      const uint32_t wlo = xcd * q + (xcd < rem ? xcd : rem);
      const uint32_t w = q + (xcd < rem ? 1u : 0u);
      b = A.batch0 + (uint32_t)((uint64_t)nbt * wlo / G) + slot;
      bend = A.batch0 + (uint32_t)((uint64_t)nbt * (wlo + w) / G);
      bstride = w;
    } else {
      b = A.batch0 + blockIdx.x;
      bend = A.batch_end;
      bstride = G;
    }
  }
  if (b >= bend) return;

  const int lc = tid / P;
  const int pen = tid - lc * P;
  const int pa = pen % n;
  const int pb = pen / n;
  const bool lane_on = tid < CH * P;
  const int bx = n * pa + n2 * pb;  // x-pencil (y = pa, z = pb), stride 1
  const int by = pa + n2 * pb;      // y-pencil (x = pa, z = pb), stride n
  const int bz = pa + n * pb;       // z-pencil (x = pa, y = pb), stride n2
  T *Wc = Wb + lc * nd, *Gxc = Gxb + lc * nd, *Gyc = Gyb + lc * nd, *Gzc = Gzb + lc * nd;

  auto lane = [&]() {  // opaque copy of the thread index: keeps hipcc from hoisting tid + j*256 constants
    int l = tid;
    asm volatile("" : "+v"(l));
    return l;
  };
  uint32_t c0, d0, hoff;
  int nb, ncell, nint;
  auto load_meta = [&](uint32_t bb, uint32_t &c0_, int &ncell_, uint32_t &d0_, int &nb_, int &nint_, uint32_t &hoff_) {
    c0_ = A.batch_cell_off[bb];
    ncell_ = (int)(A.batch_cell_off[bb + 1] - c0_);
    d0_ = A.batch_dof_off[bb];
    nb_ = (int)(A.batch_dof_off[bb + 1] - d0_);
    nint_ = (int)A.batch_nint[bb];
    hoff_ = A.halo_off[bb];
  };
  auto load_dofs = [&](uint32_t d0_, int nb_, uint32_t (&g_)[kGU]) {
    const int l = lane();
    const uint32_t *bd = A.bdofs + d0_;
#pragma unroll
    for (int j = 0; j < kGU; ++j) {
      const int t = l + j * kBlock;
      g_[j] = g_stream_load(bd + (t < nb_ ? t : nb_ - 1));
    }
  };
  auto load_src = [&](const uint32_t (&g_)[kGU], T (&sv_)[kGU]) {
#pragma unroll
    for (int j = 0; j < kGU; ++j) sv_[j] = A.src[g_[j] & 0x7fffffffu];
  };
  auto load_ix = [&](uint32_t c0_, int ncell_, uint32_t (&ix_)[kMaxChunks][NW]) {
    const uint32_t *lx = reinterpret_cast<const uint32_t *>(A.lmapx);
#pragma unroll
    for (int k = 0; k < kMaxChunks; ++k) {
      int cell = k * CH + lc;
      cell = cell < ncell_ ? cell : ncell_ - 1;
      const uint32_t *p = lx + ((size_t)(c0_ + cell) * P + (lane_on ? pen : 0)) * NW;
#pragma unroll
      for (int q = 0; q < NW; ++q) ix_[k][q] = g_stream_load(p + q);
    }
  };
  // folded metric M = a JxW J J^T of this thread's points of a chunk, stored [cell][e][q] with
  // e = {00, 01, 02, 11, 12, 22}: for one entry the lanes of a wave read consecutive doubles
  T M[PF][6];
  auto load_metric = [&](uint32_t cell0, int cnt) {
    const T *mg = A.coef + (size_t)cell0 * nd * 6;
    const int l = lane();
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      int i = l + j * kBlock;
      i = i < cnt ? i : cnt - 1;
      const int cl = i / nd, q = i - cl * nd;
      const T *p = mg + (size_t)cl * (6 * nd) + q;
#pragma unroll
      for (int e = 0; e < 6; ++e) M[j][e] = g_stream_load(p + e * nd);
    }
  };
  auto chunk_count = [&](int ncell_, int base_) { return (ncell_ - base_ < CH ? ncell_ - base_ : CH) * nd; };

  if (HN)
    for (int t = tid; t < n2; t += kBlock) Wl[t] = A.hn_weights[t];
  uint32_t G[kGU];
  T SV[kGU];
  uint32_t IX[kMaxChunks][NW];
  load_meta(b, c0, ncell, d0, nb, nint, hoff);
  load_dofs(d0, nb, G);
  load_ix(c0, ncell, IX);
  load_src(G, SV);
  while (true) {
    // ---- 1. gather result -> LDS; bit 31 of a dof entry = constrained row: reads as 0, and the owning batch
    // writes dst = src (constraint_handler_gpu.cu:258-259,286)
    {
      const int l = lane();
      double *ul = ua + l;
#pragma unroll
      for (int j = 0; j < kGU; ++j) {
        const bool con = (G[j] >> 31) != 0;
        if (l < nb - j * kBlock) {
          ul[j * kBlock] = con ? 0.0 : (double)SV[j];
          if (con && l < nint - j * kBlock) {
            T *d = A.dst + (G[j] & 0x7fffffffu);
            *d = A.add ? *d + SV[j] : SV[j];
          }
        }
      }
    }
    const uint32_t bn = b + bstride;
    const bool has_nb = bn < bend;
    uint32_t c0n = c0, d0n = d0, hoffn = hoff;
    int nbn = nb, ncelln = ncell, nintn = nint;
    uint32_t Gn[kGU];
    T SVn[kGU];
    uint32_t IXn[kMaxChunks][NW];
    if (has_nb) {
      load_meta(bn, c0n, ncelln, d0n, nbn, nintn, hoffn);
      load_dofs(d0n, nbn, Gn);
    }
    __syncthreads();
    // ---- 2. source pencils of every chunk -> registers; afterwards the array is the accumulator
    T U[kMaxChunks][n];
#pragma unroll
    for (int k = 0; k < kMaxChunks; ++k)
      if (k * CH < ncell) {
#pragma unroll
        for (int i = 0; i < n; ++i) U[k][i] = (T)ua[gix_at<n>(IX[k], i)];
      }
    __syncthreads();
    {
      const int l = lane();
      double *ul = ua + l;
#pragma unroll
      for (int j = 0; j < kGU; ++j)
        if (l < nb - j * kBlock) ul[j * kBlock] = 0.0;
    }

    // ---- 3. cells
#pragma unroll
    for (int k = 0; k < kMaxChunks; ++k) {
      const int base = k * CH;
      if (base >= ncell) continue;  // uniform
      const bool act = lane_on && (base + lc < ncell);
      const int cnt = chunk_count(ncell, base);
      load_metric(c0 + base, cnt);  // consumed six stages later
      if (k == kMaxChunks - 1 && has_nb) load_src(Gn, SVn);
      if (k == 1 && has_nb) load_ix(c0n, ncelln, IXn);
      T u[n], v[n], w[n], g[n];
      unsigned mask = 0;
      bool any_mask = false;
      if (HN) {
        if (act) mask = A.cmask[(size_t)c0 + base + lc];
        any_mask = __syncthreads_or(mask != 0) != 0;
      }
      if (HN && any_mask) {
        // resolve_hanging_nodes_shmem<NOTRANSPOSE>: x, then y, then z (hanging_nodes.cuh:767-777); only the
        // pencils on a constrained face or edge take the y / z round trips
        bool type;
        if (act) {
          if (mask && hn_flag3<n, 0>(mask, pa, pb, type)) hn_pencil<n, T, false, true>(Wl, type, U[k]);
          lds_store<n>(Wc + bx, 1, U[k]);
        }
        __syncthreads();
        if (act && mask && hn_flag3<n, 1>(mask, pb, pa, type)) {
          lds_load<n>(Wc + by, n, u);
          hn_pencil<n, T, false, true>(Wl, type, u);
          lds_store<n>(Wc + by, n, u);
        }
        __syncthreads();
        if (act && mask && hn_flag3<n, 2>(mask, pa, pb, type)) {
          lds_load<n>(Wc + bz, n2, u);
          hn_pencil<n, T, false, true>(Wl, type, u);
          lds_store<n>(Wc + bz, n2, u);
        }
        __syncthreads();
        if (act) lds_load<n>(Wc + bx, 1, U[k]);
      }
      // P0: interpolate along x
      if (act) {
        mvt<n, 1>(tab.S, U[k], v);
        lds_store<n>(Wc + bx, 1, v);
      }
      __syncthreads();
      // P1: interpolate along y
      if (act) {
        lds_load<n>(Wc + by, n, u);
        mvt<n, 1>(tab.S, u, v);
        lds_store<n>(Wc + by, n, v);
      }
      __syncthreads();
      // P2: interpolate along z -> values at the quadrature points; z-derivative
      if (act) {
        lds_load<n>(Wc + bz, n2, u);
        mvt<n, 1>(tab.S, u, w);
        mv<n, -1>(tab.Dt, w, g);
        lds_store<n>(Wc + bz, n2, w);
        lds_store<n>(Gzc + bz, n2, g);
      }
      __syncthreads();
      // P3: x-derivative
      if (act) {
        lds_load<n>(Wc + bx, 1, w);
        mv<n, -1>(tab.Dt, w, g);
        lds_store<n>(Gxc + bx, 1, g);
      }
      __syncthreads();
      // P4: y-derivative (last read of w: the array becomes the result r)
      if (act) {
        lds_load<n>(Wc + by, n, w);
        mv<n, -1>(tab.Dt, w, g);
        lds_store<n>(Gyc + by, n, g);
      }
      __syncthreads();
      // P5: quadrature-point operation t = M ghat, points in linear order (the chunk's cells are contiguous)
      {
        const int l = lane();
#pragma unroll
        for (int j = 0; j < PF; ++j) {
          if (l < cnt - j * kBlock) {
            const int i = l + j * kBlock;
            const T gx = Gxb[i], gy = Gyb[i], gz = Gzb[i];
            Gxb[i] = fma(M[j][0], gx, fma(M[j][1], gy, M[j][2] * gz));
            Gyb[i] = fma(M[j][1], gx, fma(M[j][3], gy, M[j][4] * gz));
            Gzb[i] = fma(M[j][2], gx, fma(M[j][4], gy, M[j][5] * gz));
          }
        }
      }
      __syncthreads();
      // P6: r = D_z^T tz
      if (act) {
        lds_load<n>(Gzc + bz, n2, g);
        mvt<n, -1>(tab.Dt, g, v);
        lds_store<n>(Wc + bz, n2, v);
      }
      __syncthreads();
      // P7: r += D_x^T tx
      if (act) {
        lds_load<n>(Gxc + bx, 1, g);
        lds_load<n>(Wc + bx, 1, u);
        mvt<n, -1>(tab.Dt, g, v);
#pragma unroll
        for (int s = 0; s < n; ++s) v[s] += u[s];
        lds_store<n>(Wc + bx, 1, v);
      }
      __syncthreads();
      // P8: r += D_y^T ty, then S^T along y
      if (act) {
        lds_load<n>(Gyc + by, n, g);
        lds_load<n>(Wc + by, n, u);
        mvt<n, -1>(tab.Dt, g, w);
#pragma unroll
        for (int s = 0; s < n; ++s) w[s] += u[s];
        mv<n, 1>(tab.S, w, v);
        lds_store<n>(Wc + by, n, v);
      }
      __syncthreads();
      // P9: S^T along z
      if (act) {
        lds_load<n>(Wc + bz, n2, u);
        mv<n, 1>(tab.S, u, v);
        lds_store<n>(Wc + bz, n2, v);
      }
      __syncthreads();
      // P10: S^T along x, add into the batch accumulator (each thread re-uses its own pencil of the array in
      // the next chunk's P0: program order, no barrier needed)
      if (act) {
        lds_load<n>(Wc + bx, 1, u);
        mv<n, 1>(tab.S, u, v);
      }
      if (HN && any_mask) {
        // resolve_hanging_nodes_shmem<TRANSPOSE>: the passes commute; y, z, then x (the index set's pencil)
        bool type;
        if (act) lds_store<n>(Wc + bx, 1, v);
        __syncthreads();
        if (act && mask && hn_flag3<n, 1>(mask, pb, pa, type)) {
          lds_load<n>(Wc + by, n, v);
          hn_pencil<n, T, true, true>(Wl, type, v);
          lds_store<n>(Wc + by, n, v);
        }
        __syncthreads();
        if (act && mask && hn_flag3<n, 2>(mask, pa, pb, type)) {
          lds_load<n>(Wc + bz, n2, v);
          hn_pencil<n, T, true, true>(Wl, type, v);
          lds_store<n>(Wc + bz, n2, v);
        }
        __syncthreads();
        if (act) {
          lds_load<n>(Wc + bx, 1, v);
          if (mask && hn_flag3<n, 0>(mask, pa, pb, type)) hn_pencil<n, T, true, true>(Wl, type, v);
        }
      }
      if (act) {
#pragma unroll
        for (int i = 0; i < n; ++i) lds_add(&ua[gix_at<n>(IX[k], i)], (double)v[i]);
      }
    }
    if (has_nb && ncell <= (kMaxChunks - 1) * CH) {  // short batch (ragged meshes): no overlap
      load_src(Gn, SVn);
      if (ncell <= CH) load_ix(c0n, ncelln, IXn);
    }
    __syncthreads();

    // ---- 4. scatter: interior dofs -> dst, partial sums of shared dofs -> halo (reduce pass sums them)
    {
      const int l = lane();
      const double *ul = ua + l;
      T *hl = A.halo + hoff + l - nint;
      T old[kGU];
      if (A.add) {
#pragma unroll
        for (int j = 0; j < kGU; ++j) old[j] = A.dst[G[j] & 0x7fffffffu];
      }
#pragma unroll
      for (int j = 0; j < kGU; ++j) {
        if (l < nint - j * kBlock) {
          if (!(G[j] >> 31)) A.dst[G[j]] = A.add ? old[j] + (T)ul[j * kBlock] : (T)ul[j * kBlock];
        } else if (l < nb - j * kBlock) {
          hl[j * kBlock] = (T)ul[j * kBlock];
        }
      }
    }
    if (!has_nb) break;
    b = bn;
    c0 = c0n;
    ncell = ncelln;
    d0 = d0n;
    nb = nbn;
    nint = nintn;
    hoff = hoffn;
#pragma unroll
    for (int j = 0; j < kGU; ++j) {
      G[j] = Gn[j];
      SV[j] = SVn[j];
    }
#pragma unroll
    for (int k = 0; k < kMaxChunks; ++k)
#pragma unroll
      for (int q = 0; q < NW; ++q) IX[k][q] = IXn[k][q];
  }
}

// M = a JxW J J^T per quadrature point, plan cell order, [cell][entry][q]; J = inv_jac[cell][q] row-major J[d1][d2]
template <typename T>
__global__ void fold_general_kernel(T *M, const T *coef, const T *jxw, const T *jinv, const uint32_t *order,
                                    uint32_t n_cells, uint32_t nd) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_cells * nd) return;
  const uint32_t cell = (uint32_t)(i / nd), q = (uint32_t)(i - (size_t)cell * nd);
  const size_t s = (size_t)order[cell] * nd + q;
  const T *J = jinv + s * 9;
  const T a = coef[s] * jxw[s];
  T *m = M + (size_t)cell * (6 * nd) + q;  // [cell][e][q]
  int e = 0;
  for (int d1 = 0; d1 < 3; ++d1)
    for (int d2 = d1; d2 < 3; ++d2)
      m[(e++) * nd] = a * (J[3 * d1] * J[3 * d2] + J[3 * d1 + 1] * J[3 * d2 + 1] + J[3 * d1 + 2] * J[3 * d2 + 2]);
}

template <int n, typename T>
static size_t g_lds_bytes(uint32_t nb_max) {
  constexpr int nd = n * n * n;
  constexpr int CH = 256 / (n * n);
  return (size_t)nb_max * sizeof(double) + (size_t)(4 * CH * nd + n * n) * sizeof(T);
}

template <int n, typename T, bool HN>
static hipError_t g_run(const ApplyArgs<T> &a, const double *S, const double *Dt, uint32_t grid, hipStream_t st,
                        bool configure_only, size_t *lds_out, int *occupancy) {
  const size_t lds = g_lds_bytes<n, T>(a.nb_max);
  if (lds_out) *lds_out = lds;
  if (configure_only) {
    hipError_t e = hipFuncSetAttribute((const void *)apply_batches_g<n, T, HN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && occupancy)
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, apply_batches_g<n, T, HN>, 256, lds);
    return e;
  }
  Tables<T, n> tab;
  for (int i = 0; i < ((n + 1) / 2) * n; ++i) {
    tab.S[i] = (T)S[i];
    tab.Dt[i] = (T)Dt[i];
  }
  hipLaunchKernelGGL((apply_batches_g<n, T, HN>), dim3(grid), dim3(256), lds, st, a, tab);
  return hipGetLastError();
}

template <typename T>
hipError_t g_launch(int n, const ApplyArgs<T> &a, const double *S, const double *Dt, bool hn, uint32_t grid,
                    hipStream_t st, bool configure_only, size_t *lds_out, int *occupancy) {
#define G_CASE(N)                                                                                 \
  case N:                                                                                         \
    return hn ? g_run<N, T, true>(a, S, Dt, grid, st, configure_only, lds_out, occupancy)         \
              : g_run<N, T, false>(a, S, Dt, grid, st, configure_only, lds_out, occupancy);
  switch (n) {
    G_CASE(2) G_CASE(3) G_CASE(4) G_CASE(5) G_CASE(6) G_CASE(7)
    default: return hipErrorInvalidValue;
  }
#undef G_CASE
}

template <typename T>
hipError_t fold_general_launch(T *M, const T *coef, const T *jxw, const T *jinv, const uint32_t *order,
                               uint32_t n_cells, uint32_t nd, hipStream_t st) {
  const size_t tot = (size_t)n_cells * nd;
  hipLaunchKernelGGL(fold_general_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, M, coef, jxw, jinv,
                     order, n_cells, nd);
  return hipGetLastError();
}

#define INST(T)                                                                                              \
  template hipError_t g_launch<T>(int, const ApplyArgs<T> &, const double *, const double *, bool, uint32_t,        \
                                  hipStream_t, bool, size_t *, int *);                                                    \
  template hipError_t fold_general_launch<T>(T *, const T *, const T *, const T *, const uint32_t *, uint32_t, \
                                             uint32_t, hipStream_t);
INST(double)
INST(float)

}  // namespace mfgpu
