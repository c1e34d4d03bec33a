// Device-side building blocks of the cell kernel, shared by the kernel translation units:
// 1D table access, pencil mat-vecs, hanging-node pencil operations and the cell pipeline.
#ifndef MFGPU_CELL_H
#define MFGPU_CELL_H

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
template <int n, typename T, bool TR, bool LOWREG = false>
__device__ __forceinline__ void hn_pencil(const T *__restrict__ W, bool type, T (&v)[n]) {
  // type:  w = TR ? W[i][q] : W[q][i];  !type: the mirrored entry W[p-i][p-q] / W[p-q][p-i]
  // (hanging_nodes.cuh:665-681).
  constexpr int p = n - 1;
  T o[n];
  if (LOWREG) {
    // apply_batches_x (168-VGPR budget): the mirrored entry sits at n*n - 1 - (index of the unmirrored
    // one), so ONE load per weight through a signed index, and scheduling barriers keep hipcc from
    // fetching all n*n weights up front (2 * n * n VGPRs).  This serialises the weight loads of the
    // flagged pencils and costs apply_batches (256-VGPR budget) 4-6 % on C3, hence opt-in.
    const int sgn = type ? 1 : -1, base = type ? 0 : n * n - 1;
#pragma unroll
    for (int q = 0; q < n; ++q) {
      T t = 0;
#pragma unroll
      for (int i = 0; i < n; ++i) t = fma(W[base + sgn * (TR ? i * n + q : q * n + i)], v[i], t);
      o[q] = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int q = 0; q < n; ++q) {
      T t = 0;
#pragma unroll
      for (int i = 0; i < n; ++i) {
        const T w1 = TR ? W[i * n + q] : W[q * n + i];
        const T w2 = TR ? W[(p - i) * n + (p - q)] : W[(p - q) * n + (p - i)];
        t = fma(type ? w1 : w2, v[i], t);
      }
      o[q] = t;
    }
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

// Diagnostic build only (-DMFGPU_STAMPS, lib/libmfgpu_diag.so): lane 0 of every workgroup records
// s_memtime at phase boundaries into a buffer no kernel reads.  The product build has no stamps.
#ifdef MFGPU_STAMPS
#define STAMP(k)                                                                          \
  do {                                                                                    \
    if (A.stamps && threadIdx.x == 0) {                                                   \
      unsigned long long t_;                                                              \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
      A.stamps[(size_t)b * 16 + (k)] = t_;                          \
    }                                                                                     \
  } while (0)
#define DBG(bit) (A.dbg & (bit))
// constant-rate (100 MHz) counter next to a cycle stamp: in-kernel clock = d(s_memtime) / d(s_memrealtime) * 100 MHz
#define RSTAMP(k)                                                                         \
  do {                                                                                    \
    if (A.stamps && threadIdx.x == 0) {                                                   \
      unsigned long long t_;                                                              \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
      A.stamps[(size_t)b * 16 + (k)] = t_;                                                \
    }                                                                                     \
  } while (0)
// which workgroup ran the batch
#define WGSTAMP(k)                                                                        \
  do {                                                                                    \
    if (A.stamps && threadIdx.x == 0) A.stamps[(size_t)b * 16 + (k)] = blockIdx.x + 1;    \
  } while (0)
#else
#define RSTAMP(k)
#define STAMP(k)
#define WGSTAMP(k)
#define DBG(bit) 0
#endif

// a batch's dof list and source values pass through registers: 9 per thread at 256 threads, 12 at 64
constexpr int max_batch_dofs(int kBlock) { return kBlock == 256 ? 2304 : 768; }
constexpr int kMaxChunks = 3;  // chunks of cells per batch (unrolled in the kernel; planner: mfgpu_plan.cpp)

// Synchronisation granularity of the cell pipeline: the transposes between two contraction stages go
// through LDS; with whole cells owned by ONE wave they need no s_barrier, only program order.
struct WgSync {
  __device__ static __forceinline__ void sync() { __syncthreads(); }
};
struct WaveSync {
  __device__ static __forceinline__ void sync() {
    // LDS instructions of one wave execute in order; this only stops the compiler from moving LDS
    // accesses of other lanes' data across the stage boundary
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
};

// The cell kernel for the cells a thread group holds at once: gather from the batch array `usrc`,
// 4*dim contractions (+ hanging-node passes), add into the batch accumulator `acc`.  A thread owns
// pencil (pa, pb) of its cell; Wc / Rc / cf / lm point to the cell's scratch, coefficient and index map
// in LDS.  stage_next() is called once, at the point where cf / lm are no longer needed.
template <int dim, int n, typename T, bool HN, typename Sync, typename StageNext>
__device__ __forceinline__ void cell_pipeline(const bool act, const int pa, const int pb, const unsigned mask,
                                              const bool any_mask, const T *usrc, double *acc, T *Wc, T *Rc,
                                              const T *cf, const uint16_t *lm, const T *Wl,
                                              const Tables<T, n> &tab, StageNext &&stage_next,
                                              const int dbg = 0, unsigned long long *pst = nullptr) {
  constexpr int n2 = n * n;
#ifdef MFGPU_STAMPS
#define PDBG(bit) (dbg & (bit))
#define PSTAMP(k)                                                                        \
  do {                                                                                   \
    if (pst && threadIdx.x == 0) {                                                       \
      unsigned long long t_;                                                             \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
      pst[k] = t_;                                                                       \
    }                                                                                    \
  } while (0)
#else
#define PDBG(bit) 0
#define PSTAMP(k)
#endif
  PSTAMP(0);
  const int bx = (dim == 3) ? n * pa + n2 * pb : n * pa;  // x-pencil, stride 1
  const int by = (dim == 3) ? pa + n2 * pb : pa;          // y-pencil, stride n
  const int bz = pa + n * pb;                             // z-pencil, stride n2 (3D only)
  constexpr int sl = (dim == 3) ? n2 : n;                 // stride of the last direction
  const int bl = (dim == 3) ? bz : by;                    // pencil base of the last direction
  T u[n], v[n], w[n], g[n], r[n];
  uint16_t ix[n], iz[n];
  if (act) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
      ix[i] = lm[bx + i];
      iz[i] = lm[bl + i * sl];
    }
#pragma unroll
    for (int i = 0; i < n; ++i) u[i] = PDBG(32) ? T(ix[i]) : usrc[ix[i]];
  }

  if (dim == 3) {
    if (HN && any_mask) {
      // resolve_hanging_nodes_shmem<NOTRANSPOSE>: x, then y, then z (hanging_nodes.cuh:767-777)
      bool type;
      if (act) {
        if (mask && hn_flag3<n, 0>(mask, pa, pb, type)) hn_pencil<n, T, false>(Wl, type, u);
        lds_store<n>(Wc + bx, 1, u);
      }
      Sync::sync();
      // only the pencils on a constrained face or edge change: everybody else skips the round trip
      if (act && mask && hn_flag3<n, 1>(mask, pb, pa, type)) {
        lds_load<n>(Wc + by, n, u);
        hn_pencil<n, T, false>(Wl, type, u);
        lds_store<n>(Wc + by, n, u);
      }
      Sync::sync();
      if (act && mask && hn_flag3<n, 2>(mask, pa, pb, type)) {
        lds_load<n>(Wc + bz, n2, u);
        hn_pencil<n, T, false>(Wl, type, u);
        lds_store<n>(Wc + bz, n2, u);
      }
      Sync::sync();
      if (act) lds_load<n>(Wc + bx, 1, u);  // P0 rewrites the same pencil in place
    }
    // P0: interpolate along x
    if (act) {
      mvt<n, 1>(tab.S, u, v);
      lds_store<n>(Wc + bx, 1, v);
    }
    Sync::sync();
    PSTAMP(1);
    // P1: interpolate along y
    if (act) {
      lds_load<n>(Wc + by, n, u);
      mvt<n, 1>(tab.S, u, v);
      lds_store<n>(Wc + by, n, v);
    }
    Sync::sync();
    PSTAMP(2);
    // P2: interpolate along z -> values at quadrature points; z-derivative part
    if (act) {
      lds_load<n>(Wc + bz, n2, u);
      lds_load<n>(cf + bz, n2, v);
      mvt<n, 1>(tab.S, u, w);
      mv<n, -1>(tab.Dt, w, g);
#pragma unroll
      for (int s = 0; s < n; ++s) g[s] *= v[s];
      mvt<n, -1>(tab.Dt, g, r);
      lds_store<n>(Wc + bz, n2, w);
      lds_store<n>(Rc + bz, n2, r);
    }
    Sync::sync();
    PSTAMP(3);
    // P3: y-derivative part
    if (act) {
      lds_load<n>(Wc + by, n, w);
      lds_load<n>(cf + by, n, v);
      mv<n, -1>(tab.Dt, w, g);
#pragma unroll
      for (int s = 0; s < n; ++s) g[s] *= v[s];
      mvt<n, -1>(tab.Dt, g, r);
      lds_load<n>(Rc + by, n, v);
#pragma unroll
      for (int s = 0; s < n; ++s) r[s] += v[s];
      lds_store<n>(Rc + by, n, r);
    }
    Sync::sync();
    PSTAMP(4);
    // P4: x-derivative part, then S^T along x
    if (act) {
      lds_load<n>(Wc + bx, 1, w);
      lds_load<n>(cf + bx, 1, v);
      mv<n, -1>(tab.Dt, w, g);
#pragma unroll
      for (int s = 0; s < n; ++s) g[s] *= v[s];
      mvt<n, -1>(tab.Dt, g, r);
      lds_load<n>(Rc + bx, 1, v);
#pragma unroll
      for (int s = 0; s < n; ++s) r[s] += v[s];
      mv<n, 1>(tab.S, r, v);
      lds_store<n>(Rc + bx, 1, v);
    }
    Sync::sync();
    PSTAMP(5);
    // P5: S^T along y; coefficient / index buffers are free now (last read in P4 / at the chunk
    // start): stage the next chunk
    if (act) {
      lds_load<n>(Rc + by, n, u);
      mv<n, 1>(tab.S, u, v);
      lds_store<n>(Rc + by, n, v);
    }
    stage_next();
    Sync::sync();
    PSTAMP(6);
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
      Sync::sync();
      if (act && mask && hn_flag3<n, 1>(mask, pb, pa, type)) {
        lds_load<n>(Rc + by, n, v);
        hn_pencil<n, T, true>(Wl, type, v);
        lds_store<n>(Rc + by, n, v);
      }
      Sync::sync();
      if (act) {
        lds_load<n>(Rc + bx, 1, v);
        if (mask && hn_flag3<n, 0>(mask, pa, pb, type)) hn_pencil<n, T, true>(Wl, type, v);
#pragma unroll
        for (int i = 0; i < n; ++i) lds_add(&acc[ix[i]], (double)v[i]);
      }
    } else if (act) {
#pragma unroll
      for (int k = 0; k < n; ++k) if (!PDBG(16)) lds_add(&acc[iz[k]], (double)v[k]); else asm volatile("" ::"v"(v[k]), "v"(iz[k]));
    }
    PSTAMP(7);
  } else {  // dim == 2
    if (HN && any_mask) {
      bool type;
      if (act) {
        if (mask && hn_flag2<n, 0>(mask, pa, type)) hn_pencil<n, T, false>(Wl, type, u);
        lds_store<n>(Wc + bx, 1, u);
      }
      Sync::sync();
      if (act && mask && hn_flag2<n, 1>(mask, pa, type)) {
        lds_load<n>(Wc + by, n, u);
        hn_pencil<n, T, false>(Wl, type, u);
        lds_store<n>(Wc + by, n, u);
      }
      Sync::sync();
      if (act) lds_load<n>(Wc + bx, 1, u);
    }
    // P0: interpolate along x
    if (act) {
      mvt<n, 1>(tab.S, u, v);
      lds_store<n>(Wc + bx, 1, v);
    }
    Sync::sync();
    // P1: interpolate along y; y-derivative part
    if (act) {
      lds_load<n>(Wc + by, n, u);
      lds_load<n>(cf + by, n, v);
      mvt<n, 1>(tab.S, u, w);
      mv<n, -1>(tab.Dt, w, g);
#pragma unroll
      for (int s = 0; s < n; ++s) g[s] *= v[s];
      mvt<n, -1>(tab.Dt, g, r);
      lds_store<n>(Wc + by, n, w);
      lds_store<n>(Rc + by, n, r);
    }
    Sync::sync();
    // P2: x-derivative part, S^T along x
    if (act) {
      lds_load<n>(Wc + bx, 1, w);
      lds_load<n>(cf + bx, 1, v);
      mv<n, -1>(tab.Dt, w, g);
#pragma unroll
      for (int s = 0; s < n; ++s) g[s] *= v[s];
      mvt<n, -1>(tab.Dt, g, r);
      lds_load<n>(Rc + bx, 1, v);
#pragma unroll
      for (int s = 0; s < n; ++s) r[s] += v[s];
      mv<n, 1>(tab.S, r, v);
      lds_store<n>(Rc + bx, 1, v);
    }
    Sync::sync();
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
      Sync::sync();
      if (act) {
        lds_load<n>(Rc + bx, 1, v);
        if (mask && hn_flag2<n, 0>(mask, pa, type)) hn_pencil<n, T, true>(Wl, type, v);
#pragma unroll
        for (int i = 0; i < n; ++i) lds_add(&acc[ix[i]], (double)v[i]);
      }
    } else if (act) {
#pragma unroll
      for (int k = 0; k < n; ++k) if (!PDBG(16)) lds_add(&acc[iz[k]], (double)v[k]); else asm volatile("" ::"v"(v[k]), "v"(iz[k]));
    }
    if (stage_next()) Sync::sync();  // Cb last read in P2, Lb at the chunk start
  }
}

}  // namespace mfgpu
#endif
