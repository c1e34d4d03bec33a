// CDNA4 (gfx950) kernels of the matrix-free Laplace apply.
//
// One workgroup (256 threads = 4 waves) owns one BATCH of cells (mfgpu_plan.cpp).  It
//   1. gathers the batch's unique source dofs once into LDS (constrained dofs read as 0:
//      the reference zeroes them in src instead, constraint_handler_gpu.cu:247-261),
//   2. runs the cell kernel for CH cells at a time.  A thread owns one 1D PENCIL of n values in
//      registers; every 1D contraction is an n x n mat-vec in registers with the 1D tables in
//      scalar registers (kernel arguments), and the direction change between two contractions
//      is an in-place transpose through LDS (one barrier).  The reference instead keeps one
//      thread per dof and reads every operand of every contraction from shared memory
//      (tensor_ops.cuh:84-116),
//   3. sums the cell results per batch dof in LDS, and
//   4. writes each dof once: store if this batch is the dof's first toucher in launch order,
//      add otherwise (batches of one launch never share a dof: coloured scatter, no atomics in
//      global memory; reference fee_gpu.cuh:359-362 adds per cell).
//
// Cell kernel algebra (uniform-Jacobian path, fee_gpu.cuh:234,274): with S = nodal->quadrature
// interpolation and D = collocation derivative on the quadrature points,
//   grad_d = D_d (S u),  out = S^T sum_d D_d^T ( c .* grad_d ),  c = a * J0^2 * JxW
// which equals the reference's 2*dim*dim contractions (tensor_ops.cuh:179-261) because
// phi_i'(x_q) = sum_t phi_i(x_t) l_t'(x_q); it needs 4*dim contractions.
#include <hip/hip_runtime.h>

#include "mfgpu_kernels.h"

namespace mfgpu {

// The 1D tables are centro-(anti)symmetric because support and quadrature points are symmetric
// about 1/2:  S[i][q] = S[p-i][p-q],  Dt[q][t] = -Dt[p-q][p-t].  Only rows 0..(n+1)/2-1 are
// passed (kernel arguments live in scalar registers: two full 5x5 double tables would need 100
// of the 102 SGPRs).  sgn = +1 for S, -1 for Dt.
template <int n, int sgn, typename T>
__device__ __forceinline__ T tab_at(const T *__restrict__ M, int r, int c) {
  constexpr int R = (n + 1) / 2, p = n - 1;
  if (r < R) return M[r * n + c];
  return sgn > 0 ? M[(p - r) * n + (p - c)] : -M[(p - r) * n + (p - c)];
}
// out[q] = sum_k M[q][k] in[k]
template <int n, int sgn, typename T>
__device__ __forceinline__ void mv(const T *__restrict__ M, const T (&in)[n], T (&out)[n]) {
#pragma unroll
  for (int q = 0; q < n; ++q) {
    T t = tab_at<n, sgn>(M, q, 0) * in[0];
#pragma unroll
    for (int k = 1; k < n; ++k) t = fma(tab_at<n, sgn>(M, q, k), in[k], t);
    out[q] = t;
  }
}
// out[q] = sum_k M[k][q] in[k]
template <int n, int sgn, typename T>
__device__ __forceinline__ void mvt(const T *__restrict__ M, const T (&in)[n], T (&out)[n]) {
#pragma unroll
  for (int q = 0; q < n; ++q) {
    T t = tab_at<n, sgn>(M, 0, q) * in[0];
#pragma unroll
    for (int k = 1; k < n; ++k) t = fma(tab_at<n, sgn>(M, k, q), in[k], t);
    out[q] = t;
  }
}

template <int n, typename T>
__device__ __forceinline__ void lds_load(const T *p, int stride, T (&v)[n]) {
#pragma unroll
  for (int i = 0; i < n; ++i) v[i] = p[i * stride];
}
template <int n, typename T>
__device__ __forceinline__ void lds_store(T *p, int stride, const T (&v)[n]) {
#pragma unroll
  for (int i = 0; i < n; ++i) p[i * stride] = v[i];
}

// One directional pass of resolve_hanging_nodes_shmem on the pencil a thread owns
// (hanging_nodes.cuh:617-758).  `flag` = the reference's per-thread flag, identical for all
// points of a pencil along `direction`; type = constr & this_type.
template <int n, typename T, bool TR>
__device__ __forceinline__ void hn_pencil(const T *__restrict__ W, bool type, T (&v)[n]) {
  constexpr int p = n - 1;
  T o[n];
#pragma unroll
  for (int q = 0; q < n; ++q) {
    T t = 0;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      // type:  w = TR ? W[i][q] : W[q][i];  !type: mirrored (hanging_nodes.cuh:665-681)
      const T w1 = TR ? W[i * n + q] : W[q * n + i];
      const T w2 = TR ? W[(p - i) * n + (p - q)] : W[(p - q) * n + (p - i)];
      t = fma(type ? w1 : w2, v[i], t);
    }
    o[q] = t;
  }
#pragma unroll
  for (int q = 0; q < n; ++q) v[q] = o[q];
}

// flag of interpolate_boundary_3d for a pencil along `dir` with face1_idx=f1, face2_idx=f2
template <int n, int dir>
__device__ __forceinline__ bool hn_flag3(unsigned constr, int f1, int f2, bool &type) {
  constexpr int p = n - 1;
  constexpr unsigned TYPE[3] = {1u << 0, 1u << 1, 1u << 2};
  constexpr unsigned FACE[3] = {1u << 3, 1u << 4, 1u << 5};
  constexpr unsigned EDGE[3] = {1u << 7, 1u << 8, 1u << 6};  // dir0: YZ, dir1: ZX, dir2: XY
  constexpr int d1 = (dir + 1) % 3, d2 = (dir + 2) % 3;
  const bool on1 = (constr & TYPE[d1]) ? (f1 == 0) : (f1 == p);
  const bool on2 = (constr & TYPE[d2]) ? (f2 == 0) : (f2 == p);
  type = (constr & TYPE[dir]) != 0;
  return ((constr & FACE[d1]) && on1) || ((constr & FACE[d2]) && on2) ||
         ((constr & EDGE[dir]) && on1 && on2);
}
// flag of interpolate_boundary_2d for a pencil along `dir` at other-coordinate o
template <int n, int dir>
__device__ __forceinline__ bool hn_flag2(unsigned constr, int o, bool &type) {
  constexpr int p = n - 1;
  constexpr unsigned TYPE[2] = {1u << 0, 1u << 1};
  constexpr unsigned FACE[2] = {1u << 3, 1u << 4};
  type = (constr & TYPE[dir]) != 0;
  const bool on = (constr & TYPE[1 - dir]) ? (o == 0) : (o == p);
  return (constr & FACE[1 - dir]) && on;
}

template <typename T>
__device__ __forceinline__ void lds_add(T *p, T v) {
  unsafeAtomicAdd(p, v);  // ds_add_f64 / ds_add_f32: no CAS loop on gfx950
}

template <int dim, int n, typename T, bool HN>
__global__ void __launch_bounds__(kBlock)
apply_batches(const ApplyArgs<T> A, const Tables<T, n> tab) {
  constexpr int nd = (dim == 3) ? n * n * n : n * n;
  constexpr int P = nd / n;          // pencils per cell
  constexpr int CH = kBlock / P;     // cells per chunk
  constexpr int n2 = n * n;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *usrc = reinterpret_cast<T *>(smem_raw);
  T *acc = usrc + A.nb_max;
  T *Wb = acc + A.nb_max;
  T *Rb = Wb + CH * nd;
  T *Wl = Rb + CH * nd;  // hanging-node weights (HN only), broadcast reads

  const int tid = threadIdx.x;
  const uint32_t b = A.batch0 + blockIdx.x;
  const uint32_t c0 = A.batch_cell_off[b], c1 = A.batch_cell_off[b + 1];
  const uint32_t d0 = A.batch_dof_off[b];
  const int nb = (int)(A.batch_dof_off[b + 1] - d0);

  if (HN) {
    for (int t = tid; t < n * n; t += kBlock) Wl[t] = A.hn_weights[t];
  }
  // ---- 1. gather (read_dof_values, fee_gpu.cuh:323-331, once per batch dof)
  for (int t = tid; t < nb; t += kBlock) {
    const uint32_t g = A.bdofs[d0 + t];
    const uint8_t f = A.bflags[d0 + t];
    usrc[t] = (f & kFlagConstrained) ? T(0) : A.src[g];
    acc[t] = T(0);
  }
  __syncthreads();

  // ---- 2. cells
  const int lc = tid / P;
  const int pen = tid - lc * P;
  const int pa = (dim == 3) ? pen % n : pen;
  const int pb = (dim == 3) ? pen / n : 0;
  const int ncell = (int)(c1 - c0);
  T *Wc = Wb + lc * nd;
  T *Rc = Rb + lc * nd;
  for (int base = 0; base < ncell; base += CH) {
    const bool act = (tid < CH * P) && (base + lc < ncell);
    const size_t cell = (size_t)c0 + base + lc;
    const uint16_t *lm = A.lmap + cell * nd;
    const T *cf = A.coef + cell * nd;
    unsigned mask = 0;
    bool any_mask = false;
    if (HN) {
      if (act) mask = A.cmask[cell];
      any_mask = __syncthreads_or(mask != 0);
    }
    T u[n], v[n], w[n], g[n], r[n];

    if (dim == 3) {
      const int bx = n * pa + n2 * pb;  // x-pencil (y=pa, z=pb), stride 1
      const int by = pa + n2 * pb;      // y-pencil (x=pa, z=pb), stride n
      const int bz = pa + n * pb;       // z-pencil (x=pa, y=pb), stride n2
      if (act) {
#pragma unroll
        for (int i = 0; i < n; ++i) u[i] = usrc[lm[bx + i]];
      }
      if (HN && any_mask) {
        // resolve_hanging_nodes_shmem<NOTRANSPOSE>: x, then y, then z (hanging_nodes.cuh:767-777)
        bool type;
        if (act) {
          if (mask && hn_flag3<n, 0>(mask, pa, pb, type)) hn_pencil<n, T, false>(Wl, type, u);
          lds_store<n>(Wc + bx, 1, u);
        }
        __syncthreads();
        if (act) {
          lds_load<n>(Wc + by, n, u);
          if (mask && hn_flag3<n, 1>(mask, pb, pa, type)) hn_pencil<n, T, false>(Wl, type, u);
          lds_store<n>(Wc + by, n, u);
        }
        __syncthreads();
        if (act) {
          lds_load<n>(Wc + bz, n2, u);
          if (mask && hn_flag3<n, 2>(mask, pa, pb, type)) hn_pencil<n, T, false>(Wl, type, u);
          lds_store<n>(Wc + bz, n2, u);
        }
        __syncthreads();
        if (act) lds_load<n>(Wc + bx, 1, u);  // P0 rewrites the same pencil in place
      }
      // P0: interpolate along x
      if (act) {
        mvt<n, 1>(tab.S, u, v);
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
      // P2: interpolate along z -> values at quadrature points; z-derivative part
      if (act) {
        lds_load<n>(Wc + bz, n2, u);
        mvt<n, 1>(tab.S, u, w);
        mv<n, -1>(tab.Dt, w, g);
#pragma unroll
        for (int s = 0; s < n; ++s) g[s] *= cf[bz + s * n2];
        mvt<n, -1>(tab.Dt, g, r);
        lds_store<n>(Wc + bz, n2, w);
        lds_store<n>(Rc + bz, n2, r);
      }
      __syncthreads();
      // P3: y-derivative part
      if (act) {
        lds_load<n>(Wc + by, n, w);
        mv<n, -1>(tab.Dt, w, g);
#pragma unroll
        for (int s = 0; s < n; ++s) g[s] *= cf[by + s * n];
        mvt<n, -1>(tab.Dt, g, r);
        lds_load<n>(Rc + by, n, v);
#pragma unroll
        for (int s = 0; s < n; ++s) r[s] += v[s];
        lds_store<n>(Rc + by, n, r);
      }
      __syncthreads();
      // P4: x-derivative part, then S^T along x
      if (act) {
        lds_load<n>(Wc + bx, 1, w);
        mv<n, -1>(tab.Dt, w, g);
#pragma unroll
        for (int s = 0; s < n; ++s) g[s] *= cf[bx + s];
        mvt<n, -1>(tab.Dt, g, r);
        lds_load<n>(Rc + bx, 1, v);
#pragma unroll
        for (int s = 0; s < n; ++s) r[s] += v[s];
        mv<n, 1>(tab.S, r, v);
        lds_store<n>(Rc + bx, 1, v);
      }
      __syncthreads();
      // P5: S^T along y
      if (act) {
        lds_load<n>(Rc + by, n, u);
        mv<n, 1>(tab.S, u, v);
        lds_store<n>(Rc + by, n, v);
      }
      __syncthreads();
      // P6: S^T along z, scatter-add into the batch accumulator
      if (act) {
        lds_load<n>(Rc + bz, n2, u);
        mv<n, 1>(tab.S, u, v);
      }
      if (HN && any_mask) {
        // resolve_hanging_nodes_shmem<TRANSPOSE>; the three passes commute, z is applied first
        // because v is already a z-pencil (reference order x,y,z: hanging_nodes.cuh:767-777)
        bool type;
        if (act) {
          if (mask && hn_flag3<n, 2>(mask, pa, pb, type)) hn_pencil<n, T, true>(Wl, type, v);
          lds_store<n>(Rc + bz, n2, v);
        }
        __syncthreads();
        if (act) {
          lds_load<n>(Rc + by, n, v);
          if (mask && hn_flag3<n, 1>(mask, pb, pa, type)) hn_pencil<n, T, true>(Wl, type, v);
          lds_store<n>(Rc + by, n, v);
        }
        __syncthreads();
        if (act) {
          lds_load<n>(Rc + bx, 1, v);
          if (mask && hn_flag3<n, 0>(mask, pa, pb, type)) hn_pencil<n, T, true>(Wl, type, v);
#pragma unroll
          for (int i = 0; i < n; ++i) lds_add(&acc[lm[bx + i]], v[i]);
        }
      } else if (act) {
#pragma unroll
        for (int k = 0; k < n; ++k) lds_add(&acc[lm[bz + k * n2]], v[k]);
      }
    } else {  // dim == 2
      const int bx = n * pa;  // x-pencil (y=pa), stride 1
      const int by = pa;      // y-pencil (x=pa), stride n
      if (act) {
#pragma unroll
        for (int i = 0; i < n; ++i) u[i] = usrc[lm[bx + i]];
      }
      if (HN && any_mask) {
        bool type;
        if (act) {
          if (mask && hn_flag2<n, 0>(mask, pa, type)) hn_pencil<n, T, false>(Wl, type, u);
          lds_store<n>(Wc + bx, 1, u);
        }
        __syncthreads();
        if (act) {
          lds_load<n>(Wc + by, n, u);
          if (mask && hn_flag2<n, 1>(mask, pa, type)) hn_pencil<n, T, false>(Wl, type, u);
          lds_store<n>(Wc + by, n, u);
        }
        __syncthreads();
        if (act) lds_load<n>(Wc + bx, 1, u);
      }
      // P0: interpolate along x
      if (act) {
        mvt<n, 1>(tab.S, u, v);
        lds_store<n>(Wc + bx, 1, v);
      }
      __syncthreads();
      // P1: interpolate along y; y-derivative part
      if (act) {
        lds_load<n>(Wc + by, n, u);
        mvt<n, 1>(tab.S, u, w);
        mv<n, -1>(tab.Dt, w, g);
#pragma unroll
        for (int s = 0; s < n; ++s) g[s] *= cf[by + s * n];
        mvt<n, -1>(tab.Dt, g, r);
        lds_store<n>(Wc + by, n, w);
        lds_store<n>(Rc + by, n, r);
      }
      __syncthreads();
      // P2: x-derivative part, S^T along x
      if (act) {
        lds_load<n>(Wc + bx, 1, w);
        mv<n, -1>(tab.Dt, w, g);
#pragma unroll
        for (int s = 0; s < n; ++s) g[s] *= cf[bx + s];
        mvt<n, -1>(tab.Dt, g, r);
        lds_load<n>(Rc + bx, 1, v);
#pragma unroll
        for (int s = 0; s < n; ++s) r[s] += v[s];
        mv<n, 1>(tab.S, r, v);
        lds_store<n>(Rc + bx, 1, v);
      }
      __syncthreads();
      // P3: S^T along y, scatter-add
      if (act) {
        lds_load<n>(Rc + by, n, u);
        mv<n, 1>(tab.S, u, v);
      }
      if (HN && any_mask) {
        bool type;
        if (act) {
          if (mask && hn_flag2<n, 1>(mask, pa, type)) hn_pencil<n, T, true>(Wl, type, v);
          lds_store<n>(Rc + by, n, v);
        }
        __syncthreads();
        if (act) {
          lds_load<n>(Rc + bx, 1, v);
          if (mask && hn_flag2<n, 0>(mask, pa, type)) hn_pencil<n, T, true>(Wl, type, v);
#pragma unroll
          for (int i = 0; i < n; ++i) lds_add(&acc[lm[bx + i]], v[i]);
        }
      } else if (act) {
#pragma unroll
        for (int k = 0; k < n; ++k) lds_add(&acc[lm[by + k * n]], v[k]);
      }
    }
    // next chunk's first LDS write to Wc/Rc is separated from this chunk's last reads by the
    // barriers above (Wc last read in P4/P2, Rc last read before the scatter; P0 writes Wc only)
  }
  __syncthreads();

  // ---- 4. scatter (distribute_local_to_global fee_gpu.cuh:346-363 + identity rows
  //         constraint_handler_gpu.cu:276-289), one write per batch dof
  for (int t = tid; t < nb; t += kBlock) {
    const uint32_t gidx = A.bdofs[d0 + t];
    const uint8_t f = A.bflags[d0 + t];
    if (f & kFlagConstrained) {
      if (!(f & kFlagAdd)) {
        const T s = A.src[gidx];
        A.dst[gidx] = A.add ? A.dst[gidx] + s : s;
      }
    } else {
      const T val = acc[t];
      if ((f & kFlagAdd) || A.add)
        A.dst[gidx] += val;
      else
        A.dst[gidx] = val;
    }
  }
}

// dofs no cell touches (e.g. hanging nodes eliminated from loc2glob): vmult gives
// dst = src on constrained rows (identity, laplace_operator_gpu.h:300-302) and 0 elsewhere.
template <typename T>
__global__ void orphan_kernel(T *dst, const T *src, const uint32_t *orph, uint32_t n, int add) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t o = orph[i];
  const uint32_t g = o & 0x7fffffffu;
  const T s = (o >> 31) ? src[g] : T(0);
  dst[g] = add ? dst[g] + s : s;
}

// LocalCoeffOp::eval (laplace_operator_gpu.h:191-203) with one thread per quadrature point
// instead of one per cell (cell_eval_kernel, matrix_free_gpu.h:397-410)
template <typename T, int dim>
__global__ void coefficient_kernel(T *coef, const T *qpts, size_t nq) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  T s = 0;
#pragma unroll
  for (int d = 0; d < dim; ++d) {
    const T x = qpts[i * dim + d];
    s += x * x;
  }
  coef[i] = T(1) / (T(0.05) + T(2) * s);
}

// c[pos*nd+q] = coef * J0^2 * JxW for the cell at plan position pos (uniform-Jacobian path,
// fee_gpu.cuh:234,274 folded at setup)
template <typename T>
__global__ void fold_kernel(T *c, const T *coef, const T *jxw, const T *j0, const uint32_t *order,
                            uint32_t n_cells, uint32_t nd) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_cells * nd) return;
  const uint32_t pos = (uint32_t)(i / nd), q = (uint32_t)(i - (size_t)pos * nd);
  const uint32_t cell = order[pos];
  const T j = j0[cell];
  c[i] = coef[(size_t)cell * nd + q] * j * j * jxw[(size_t)cell * nd + q];
}

template <typename T>
__global__ void fill_kernel(T *v, size_t n, T a) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) v[i] = a;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------

template <int dim, int n, typename T>
static size_t lds_bytes_t(uint32_t nb_max) {
  constexpr int nd = (dim == 3) ? n * n * n : n * n;
  constexpr int CH = kBlock / (nd / n);
  return (size_t)(2 * nb_max + 2 * CH * nd + n * n) * sizeof(T);
}

template <int dim, int n, typename T>
static hipError_t launch_t(const ApplyArgs<T> &a, const double *S, const double *Dt, bool hn,
                           uint32_t nbatches, hipStream_t st) {
  Tables<T, n> tab;
  for (int i = 0; i < ((n + 1) / 2) * n; ++i) {
    tab.S[i] = (T)S[i];
    tab.Dt[i] = (T)Dt[i];
  }

  const size_t lds = lds_bytes_t<dim, n, T>(a.nb_max);
  if (hn)
    hipLaunchKernelGGL((apply_batches<dim, n, T, true>), dim3(nbatches), dim3(kBlock), lds, st, a, tab);
  else
    hipLaunchKernelGGL((apply_batches<dim, n, T, false>), dim3(nbatches), dim3(kBlock), lds, st, a, tab);
  return hipGetLastError();
}

template <int dim, int n, typename T>
static hipError_t configure_t(size_t lds) {
  hipError_t e = hipFuncSetAttribute((const void *)apply_batches<dim, n, T, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void *)apply_batches<dim, n, T, false>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

#define MFGPU_DISPATCH(CALL)                                \
  switch (dim * 10 + n) {                                   \
    case 22: return CALL(2, 2);                             \
    case 23: return CALL(2, 3);                             \
    case 24: return CALL(2, 4);                             \
    case 25: return CALL(2, 5);                             \
    case 26: return CALL(2, 6);                             \
    case 27: return CALL(2, 7);                             \
    case 32: return CALL(3, 2);                             \
    case 33: return CALL(3, 3);                             \
    case 34: return CALL(3, 4);                             \
    case 35: return CALL(3, 5);                             \
    case 36: return CALL(3, 6);                             \
    case 37: return CALL(3, 7);                             \
    default: return hipErrorInvalidValue;                   \
  }

template <typename T>
size_t apply_lds_bytes(int dim, int n, uint32_t nb_max) {
#define CALL(D, N) lds_bytes_t<D, N, T>(nb_max)
  switch (dim * 10 + n) {
    case 22: return CALL(2, 2);
    case 23: return CALL(2, 3);
    case 24: return CALL(2, 4);
    case 25: return CALL(2, 5);
    case 26: return CALL(2, 6);
    case 27: return CALL(2, 7);
    case 32: return CALL(3, 2);
    case 33: return CALL(3, 3);
    case 34: return CALL(3, 4);
    case 35: return CALL(3, 5);
    case 36: return CALL(3, 6);
    case 37: return CALL(3, 7);
    default: return 0;
  }
#undef CALL
}

template <typename T>
hipError_t apply_configure(int dim, int n, size_t lds) {
#define CALL(D, N) configure_t<D, N, T>(lds)
  MFGPU_DISPATCH(CALL)
#undef CALL
}

template <typename T>
hipError_t apply_launch(int dim, int n, const ApplyArgs<T> &a, const double *S, const double *Dt,
                        bool hn, uint32_t nbatches, hipStream_t st) {
#define CALL(D, N) launch_t<D, N, T>(a, S, Dt, hn, nbatches, st)
  MFGPU_DISPATCH(CALL)
#undef CALL
}

template <typename T>
hipError_t orphan_launch(T *dst, const T *src, const uint32_t *orph, uint32_t n, int add,
                         hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(orphan_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, st, dst, src, orph, n, add);
  return hipGetLastError();
}

template <typename T>
hipError_t coefficient_launch(T *coef, const T *qpts, size_t nq, int dim, hipStream_t st) {
  const unsigned grid = (unsigned)((nq + 255) / 256);
  if (dim == 2)
    hipLaunchKernelGGL((coefficient_kernel<T, 2>), dim3(grid), dim3(256), 0, st, coef, qpts, nq);
  else
    hipLaunchKernelGGL((coefficient_kernel<T, 3>), dim3(grid), dim3(256), 0, st, coef, qpts, nq);
  return hipGetLastError();
}

template <typename T>
hipError_t fold_launch(T *c, const T *coef, const T *jxw, const T *j0, const uint32_t *order,
                       uint32_t n_cells, uint32_t nd, hipStream_t st) {
  const size_t tot = (size_t)n_cells * nd;
  hipLaunchKernelGGL(fold_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, c, coef, jxw,
                     j0, order, n_cells, nd);
  return hipGetLastError();
}

template <typename T>
hipError_t fill_launch(T *v, size_t n, T a, hipStream_t st) {
  if (n == 0) return hipSuccess;
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;  // grid-stride (256 CUs x 8 blocks)
  hipLaunchKernelGGL(fill_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, v, n, a);
  return hipGetLastError();
}

#define INST(T)                                                                                         \
  template size_t apply_lds_bytes<T>(int, int, uint32_t);                                               \
  template hipError_t apply_configure<T>(int, int, size_t);                                             \
  template hipError_t apply_launch<T>(int, int, const ApplyArgs<T> &, const double *, const double *,   \
                                      bool, uint32_t, hipStream_t);                                     \
  template hipError_t orphan_launch<T>(T *, const T *, const uint32_t *, uint32_t, int, hipStream_t);   \
  template hipError_t coefficient_launch<T>(T *, const T *, size_t, int, hipStream_t);                  \
  template hipError_t fold_launch<T>(T *, const T *, const T *, const T *, const uint32_t *, uint32_t,  \
                                     uint32_t, hipStream_t);                                            \
  template hipError_t fill_launch<T>(T *, size_t, T, hipStream_t);
INST(double)
INST(float)

}  // namespace mfgpu
