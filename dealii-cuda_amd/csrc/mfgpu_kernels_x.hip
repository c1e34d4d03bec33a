// 3D cell-loop kernel, two-pass scatter mode, built for THREE workgroups per CU (p <= 4).
//
// apply_batches (mfgpu_kernels.hip) is latency-bound at 2 workgroups = 2 waves per SIMD: VALU, LDS and
// HBM are all below 35 % busy (profiles/r01_notes.md), and LDS capacity (67.6 KB per workgroup at p=4,
// 27-cell batches) is what limits residency.  This variant needs 47.8 KB for the same batch:
//   * the source pencils of ALL chunks of a batch are read from the batch array into registers at the
//     batch start, so the gathered source values and the accumulator share one LDS array;
//   * the contraction order ends in the x-layout in which it started (derivative parts z, x, y; S^T
//     along y, z, x), so the gather and the scatter-add use the same local->batch index set;
//   * that index set is an x-pencil's n contiguous 16-bit entries, padded to 32-bit words, read straight
//     from global memory (12-byte runs per thread at p=4, coalesced) one batch ahead -- no index buffer
//     in LDS.
// Everything else follows apply_batches: persistent workgroups, [interior | shared] batch dofs with the
// deterministic halo reduction (reduce_shared), constrained flag in bit 31 of the dof list, coefficient
// stream staged one chunk ahead, all global loads of the next batch in flight during the current one.
//
// Algebra per cell (uniform-Jacobian path, fee_gpu.cuh:234,274; tensor_ops.cuh:179-261):
//   w = S_z S_y S_x u,   r = sum_d D_d^T (c .* D_d w),   out = S_x^T S_z^T S_y^T r
#include <hip/hip_runtime.h>

#include "mfgpu_cell.h"
#include "mfgpu_kernels.h"

namespace mfgpu {

#ifdef MFGPU_PLAIN_STREAMS
template <typename U>
__device__ __forceinline__ U stream_load(const U *p) { return *p; }
#else
template <typename U>
__device__ __forceinline__ U stream_load(const U *p) { return __builtin_nontemporal_load(p); }
#endif

template <int n>
__device__ __forceinline__ int ix_at(const uint32_t (&w)[(n + 1) / 2], int i) {
  return (int)((w[i >> 1] >> (16 * (i & 1))) & 0xffffu);
}

// One chunk of cells: a thread owns pencil (pa, pb) of its cell.  u: the cell's source values along the
// thread's x-pencil; ixw: batch-local dof ids of that pencil (packed 16-bit).  stage_next() is called
// where the coefficient buffer is dead.
template <int n, typename T, bool HN, typename StageNext>
__device__ __forceinline__ void cell_pipeline_x(const bool act, const bool acty, const bool actz, const int pa, const int pb,
                                                const unsigned mask, const bool any_mask, T (&u)[n],
                                                const uint32_t (&ixw)[(n + 1) / 2], double *acc, T *Wc, T *Rc, const T *cf,
                                                T *Wy, T *Ry, const T *Cy, T *Wz, T *Rz, const T *Cz, const T *Wl,
                                                const Tables<T, n> &tab, StageNext &&stage_next) {

  constexpr int n2 = n * n;
  const int bx = n * pa + n2 * pb;  // x-pencil (y = pa, z = pb), stride 1
  T v[n], w[n], g[n], r[n];
  if (HN && any_mask) {
    // resolve_hanging_nodes_shmem<NOTRANSPOSE>: x, then y, then z (hanging_nodes.cuh:767-777)
    bool type;
    if (act) {
      if (mask && hn_flag3<n, 0>(mask, pa, pb, type)) hn_pencil<n, T, false, true>(Wl, type, u);
      lds_store<n>(Wc + bx, 1, u);
    }
    __syncthreads();
    // only the pencils on a constrained face or edge change: everybody else skips the round trip
    if (acty && mask && hn_flag3<n, 1>(mask, pb, pa, type)) {
      lds_load<n>(Wy, n, u);
      hn_pencil<n, T, false, true>(Wl, type, u);
      lds_store<n>(Wy, n, u);
    }
    __syncthreads();
    if (actz && mask && hn_flag3<n, 2>(mask, pa, pb, type)) {
      lds_load<n>(Wz, n2, u);
      hn_pencil<n, T, false, true>(Wl, type, u);
      lds_store<n>(Wz, n2, u);
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
  // P1: interpolate along y
  if (acty) {
    lds_load<n>(Wy, n, u);
    mvt<n, 1>(tab.S, u, v);
    lds_store<n>(Wy, n, v);
  }
  __syncthreads();
  // P2: interpolate along z -> values at the quadrature points; z-derivative part
  if (actz) {
    lds_load<n>(Wz, n2, u);
    lds_load<n>(Cz, n2, v);
    mvt<n, 1>(tab.S, u, w);
    mv<n, -1>(tab.Dt, w, g);
#pragma unroll
    for (int s = 0; s < n; ++s) g[s] *= v[s];
    mvt<n, -1>(tab.Dt, g, r);
    lds_store<n>(Wz, n2, w);
    lds_store<n>(Rz, n2, r);
  }
  __syncthreads();
  // P3: x-derivative part
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
    lds_store<n>(Rc + bx, 1, r);
  }
  __syncthreads();
  // P4: y-derivative part, then S^T along y
  if (acty) {
    lds_load<n>(Wy, n, w);
    lds_load<n>(Cy, n, v);
    mv<n, -1>(tab.Dt, w, g);
#pragma unroll
    for (int s = 0; s < n; ++s) g[s] *= v[s];
    mvt<n, -1>(tab.Dt, g, r);
    lds_load<n>(Ry, n, v);
#pragma unroll
    for (int s = 0; s < n; ++s) r[s] += v[s];
    mv<n, 1>(tab.S, r, v);
    lds_store<n>(Ry, n, v);
  }
  __syncthreads();
  // P5: S^T along z; the coefficient buffer is free now (last read in P4)
  if (actz) {
    lds_load<n>(Rz, n2, u);
    mv<n, 1>(tab.S, u, v);
    lds_store<n>(Rz, n2, v);
  }
  stage_next();
  __syncthreads();
  // P6: S^T along x, add into the batch accumulator
  if (act) {
    lds_load<n>(Rc + bx, 1, u);
    mv<n, 1>(tab.S, u, v);
  }
  if (HN && any_mask) {
    // resolve_hanging_nodes_shmem<TRANSPOSE>; the three passes commute: y, z, then x, whose pencil is
    // the one the index set belongs to (reference order x,y,z: hanging_nodes.cuh:767-777)
    bool type;
    if (act) lds_store<n>(Rc + bx, 1, v);
    __syncthreads();
    if (acty && mask && hn_flag3<n, 1>(mask, pb, pa, type)) {
      lds_load<n>(Ry, n, v);
      hn_pencil<n, T, true, true>(Wl, type, v);
      lds_store<n>(Ry, n, v);
    }
    __syncthreads();
    if (actz && mask && hn_flag3<n, 2>(mask, pa, pb, type)) {
      lds_load<n>(Rz, n2, v);
      hn_pencil<n, T, true, true>(Wl, type, v);
      lds_store<n>(Rz, n2, v);
    }
    __syncthreads();
    if (act) {
      lds_load<n>(Rc + bx, 1, v);
      if (mask && hn_flag3<n, 0>(mask, pa, pb, type)) hn_pencil<n, T, true, true>(Wl, type, v);
    }
  }
  if (act) {
#pragma unroll
    for (int i = 0; i < n; ++i) lds_add(&acc[ix_at<n>(ixw, i)], (double)v[i]);
  }
}

template <int n>
constexpr int x_waves_per_simd() { return n <= 5 ? 3 : 2; }
// chunks of cells per batch (unrolled; the planner gets the same number from mfgpu_api.hip): p=3 needs a fourth
// chunk of 16 cells to reach the 64-cell batch (13^3 dofs) that p=4 reaches with 27 cells in three chunks
template <int n>
constexpr int x_chunks() { return n == 4 ? 4 : kMaxChunks; }

template <int n, typename T, bool HN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(x_waves_per_simd<n>())))
apply_batches_x(const ApplyArgs<T> A, const Tables<T, n> tab) {
  constexpr int kBlock = 256;
  constexpr int KC = x_chunks<n>();
  constexpr int kGU = (max_batch_dofs(kBlock) + kBlock - 1) / kBlock;
  constexpr int n2 = n * n, nd = n2 * n;
  constexpr int P = n2;
  constexpr int CH = kBlock / P;
  constexpr int CHND = CH * nd;
  constexpr int NW = (n + 1) / 2;  // 32-bit words of a padded x-pencil index run
  constexpr int PF = (CHND + kBlock - 1) / kBlock;
  constexpr int kPark = 4;
  static_assert(CH >= 1, "a cell's pencils must fit into the workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // gathered source values, then the accumulator: always double.  ds_add_f32 is serviced far more slowly
  // than ds_add_f64 on gfx950 (float build: 82 M vs 44 M LDS-active cycles per launch at identical
  // instruction counts, profiles/r01_notes.md), and the float result gains accuracy
  double *ua = reinterpret_cast<double *>(smem_raw);
  T *Wb = reinterpret_cast<T *>(ua + A.nb_max);
  T *Rb = Wb + CHND;
  T *Cb = Rb + CHND;
  T *Wl = Cb + CHND;
  // kPark dof ids per thread wait here between the gather and the scatter of a batch: the registers they
  // free keep the next batch's gather out of scratch memory (a spilled load result costs a vmcnt(0))
  uint32_t *Gp = reinterpret_cast<uint32_t *>(Wl + n2);

  const int tid = threadIdx.x;
  // Persistent workgroups; XCD-aware batch order.  Workgroups are dispatched round-robin over the 8 XCDs
  // (block i runs on XCD i % 8) and every XCD has its own L2.  Consecutive batches are neighbours in the
  // mesh (two-pass plans keep the spatial creation order): they share halo dofs and the 128-byte lines
  // their 13-dof runs of src straddle.  Each XCD therefore gets ONE contiguous range of batches, walked by
  // its workgroups in steps of gridDim/8, so that the batches in flight on one XCD are neighbours and hit
  // in its L2: fabric reads 500 -> 404 MB per launch together with the spatial plan order.  (Handing each
  // XCD the x-th eighth of every ROUND of gridDim batches instead balances better when batch cost varies
  // -- apply_batches does that for hanging-node meshes -- but leaves three XCDs idle in the partial last
  // round of a uniform mesh: 4 % slower here.)
  // (A contiguous run of batches per workgroup instead -- the same lines re-requested one batch later --
  // fetched 5 % more and was 4 % slower: the L2 turns over in less than one batch time.)
  // The read-once streams (dof lists, index runs, coefficients) are loaded non-temporally so that they
  // do not push src lines out of the L2; the result stores are plain: neighbouring batches write
  // adjacent runs of dst at about the same time and the L2 merges them into full lines (non-temporal
  // stores: +24 % written bytes, +3 % time).
  uint32_t b, bstride, bend;
  {
    const uint32_t nbt = A.batch_end - A.batch0, G = gridDim.x;
    if (G >= 8 && nbt >= G) {
      // XCD x runs the blocks i = x, x + 8, ...: w(x) = (G - x + 7) / 8 of them; its batch range is
      // proportional to that count (any grid size, e.g. one that divides the batch count evenly)
      const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
      const uint32_t q = G >> 3, rem = G & 7u;
      const uint32_t wlo = xcd * q + (xcd < rem ? xcd : rem);  // blocks of the XCDs before this one
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
  T *Wc = Wb + lc * nd;
  T *Rc = Rb + lc * nd;
  const T *cf = Cb + lc * nd;
  // Which y-pencil and which z-pencil of the chunk this thread owns.  The x-stage ownership (cell lc, pencil
  // pen) is tied to the gather / scatter index runs and is conflict-free as it is (pencil bases 5 * lane).  In
  // the y- and z-stages the natural ownership costs 2x / 1.5x LDS cycles in bank conflicts, so the host
  // supplies a lane -> pencil map (A.perm, mfgpu_api.hip) under which the 32 lanes of a ds_read_b64 group and
  // the 16 lanes of a ds_write_b64 group hit distinct banks.  Hanging-node kernels keep the natural map:
  // their constraint flags are functions of the x-stage's (pa, pb).
  int cy = lc, oy = lane_on ? lc * nd + pa + n2 * pb : -1;  // y-pencil (x = pa, z = pb), stride n
  int cz = lc, oz = lane_on ? lc * nd + pa + n * pb : -1;   // z-pencil (x = pa, y = pb), stride n2
  if (!HN && A.perm) {
    const int qy = A.perm[tid], qz = A.perm[kBlock + tid];  // pencil id = cell * P + pencil, or 0xffff: idle lane
    cy = qy / P;
    oy = qy == 0xffff ? -1 : cy * nd + (qy % P) % n + n2 * ((qy % P) / n);
    cz = qz / P;
    oz = qz == 0xffff ? -1 : cz * nd + (qz % P) % n + n * ((qz % P) / n);
  }

  // hipcc hoists the per-thread constants tid + j * 256 of every unrolled helper loop out of the batch
  // loop and keeps all of them live; an opaque copy of the thread index makes them temporaries
  auto lane = [&]() {
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
  // all loads unconditional on clamped indices (a predicated load costs a branch and a full wait)
  auto load_dofs = [&](uint32_t d0_, int nb_, uint32_t (&g_)[kGU]) {
    const int l = lane();
    const uint32_t *bd = A.bdofs + d0_;
#pragma unroll
    for (int j = 0; j < kGU; ++j) {
      const int t = l + j * kBlock;
      g_[j] = stream_load(bd + (t < nb_ ? t : nb_ - 1));
    }
  };
  auto load_src = [&](const uint32_t (&g_)[kGU], T (&sv_)[kGU]) {
#pragma unroll
    for (int j = 0; j < kGU; ++j) sv_[j] = A.src[g_[j] & 0x7fffffffu];
  };
  // x-pencil index runs of the thread's cell in every chunk of a batch
  auto load_ix = [&](uint32_t c0_, int ncell_, uint32_t (&ix_)[KC][NW]) {
    const uint32_t *lx = reinterpret_cast<const uint32_t *>(A.lmapx);
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      int cell = k * CH + lc;
      cell = cell < ncell_ ? cell : ncell_ - 1;
      const uint32_t *p = lx + ((size_t)(c0_ + cell) * P + (lane_on ? pen : 0)) * NW;
#pragma unroll
      for (int q = 0; q < NW; ++q) ix_[k][q] = stream_load(p + q);
    }
  };
  T pc[PF];
  auto prefetch = [&](uint32_t cell0, int cnt) {
    const T *cg = A.coef + (size_t)cell0 * nd;
    const int l = lane();
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int i = l + j * kBlock;
      pc[j] = stream_load(cg + (i < cnt ? i : cnt - 1));
    }
  };
  auto stage = [&](int cnt) {
    const int l = lane();
    T *cl = Cb + l;
#pragma unroll
    for (int j = 0; j < PF; ++j)
      if (l < cnt - j * kBlock) cl[j * kBlock] = pc[j];
  };
  auto chunk_count = [&](int ncell_, int base_) { return (ncell_ - base_ < CH ? ncell_ - base_ : CH) * nd; };

  if (HN)
    for (int t = tid; t < n2; t += kBlock) Wl[t] = A.hn_weights[t];

  uint32_t G[kGU];
  T SV[kGU];
  uint32_t IX[KC][NW];
  load_meta(b, c0, ncell, d0, nb, nint, hoff);
  load_dofs(d0, nb, G);
  load_ix(c0, ncell, IX);
  prefetch(c0, chunk_count(ncell, 0));
  load_src(G, SV);
  stage(chunk_count(ncell, 0));
  while (true) {
    // ---- 1. gather result -> LDS (read_dof_values, fee_gpu.cuh:323-331, once per batch dof).  bdofs bit
    // 31 = constrained row: reads as 0 (constraint_handler_gpu.cu:258-259) and, if this batch owns the
    // row, dst = src (identity rows, :286).  Each thread rewrites exactly the entries it read in the
    // previous batch's scatter, so no barrier separates the two.
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
#pragma unroll
    for (int j = 0; j < kPark; ++j) Gp[j * kBlock + tid] = G[kGU - kPark + j];
    // next batch of this workgroup: meta data and dof list (the iteration's first global loads)
    const uint32_t bn = b + bstride;
    const bool has_nb = bn < bend;
    uint32_t c0n = c0, d0n = d0, hoffn = hoff;
    int nbn = nb, ncelln = ncell, nintn = nint;
    uint32_t Gn[kGU];
    T SVn[kGU];
    uint32_t IXn[KC][NW];
    if (has_nb) {
      load_meta(bn, c0n, ncelln, d0n, nbn, nintn, hoffn);
      load_dofs(d0n, nbn, Gn);
    }
    __syncthreads();
    // ---- 2. source pencils of every chunk -> registers; afterwards the array is the accumulator
    T U[KC][n];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      if (k * CH < ncell) {  // uniform
#pragma unroll
        for (int i = 0; i < n; ++i) U[k][i] = (T)ua[ix_at<n>(IX[k], i)];
      }
    }
    __syncthreads();
    {
      const int l = lane();
      double *ul = ua + l;
#pragma unroll
      for (int j = 0; j < kGU; ++j)
        if (l < nb - j * kBlock) ul[j * kBlock] = 0.0;
    }
    // (the first add into the accumulator is at least six barriers away)

    // ---- 3. cells.  The chunk loop is fully unrolled (at most KC chunks, enforced by the planner):
    // in straight-line code hipcc emits counted vmcnt waits and younger loads stay in flight.
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int base = k * CH;
      if (base >= ncell) continue;  // uniform
      const bool act = lane_on && (base + lc < ncell);
      const int nxt = base + CH;
      int cnt_next = 0;
      if (nxt < ncell) {
        cnt_next = chunk_count(ncell, nxt);
        prefetch(c0 + nxt, cnt_next);
      } else if (has_nb) {
        cnt_next = chunk_count(ncelln, 0);
        prefetch(c0n, cnt_next);
      }
      // next batch's gather and index runs: issued in the LAST chunk, where the source pencils of the
      // other chunks are dead (register budget of three waves per SIMD), after this chunk's stream prefetch
      // so that the counted wait for the stream leaves them in flight
      if (k == KC - 1 && has_nb) load_src(Gn, SVn);  // compile-time position: U[k'] are dead
      if (k == 1 && has_nb) load_ix(c0n, ncelln, IXn);
      unsigned mask = 0;
      bool any_mask = false;
      if (HN) {
        if (act) mask = A.cmask[(size_t)c0 + base + lc];
        any_mask = __syncthreads_or(mask != 0) != 0;
      }
      const bool acty = oy >= 0 && base + cy < ncell, actz = oz >= 0 && base + cz < ncell;
      cell_pipeline_x<n, T, HN>(act, acty, actz, pa, pb, mask, any_mask, U[k], IX[k], ua, Wc, Rc, cf, Wb + oy, Rb + oy,
                                Cb + oy, Wb + oz, Rb + oz, Cb + oz, Wl, tab, [&]() {
                                  if (cnt_next > 0) stage(cnt_next);
                                });
    }
    if (has_nb && ncell <= (KC - 1) * CH) {  // short batch (ragged meshes): no overlap
      load_src(Gn, SVn);
      if (ncell <= CH) load_ix(c0n, ncelln, IXn);
    }
    __syncthreads();

    // ---- 4. scatter (distribute_local_to_global fee_gpu.cuh:346-363), one write per batch dof: interior
    // dofs belong to this batch alone and are final; partial sums of shared dofs go to the batch's
    // contiguous halo slots (reduce_shared)
    {
      const int l = lane();
      const double *ul = ua + l;
      T *hl = A.halo + hoff + l - nint;
#pragma unroll
      for (int j = 0; j < kPark; ++j) G[kGU - kPark + j] = Gp[j * kBlock + tid];
      T old[kGU];
      if (A.add) {  // uniform branch
#pragma unroll
        for (int j = 0; j < kGU; ++j) old[j] = A.dst[G[j] & 0x7fffffffu];
      }
#pragma unroll
      for (int j = 0; j < kGU; ++j) {
        if (l < nint - j * kBlock) {
          if (!(G[j] >> 31)) A.dst[G[j]] = A.add ? old[j] + (T)ul[j * kBlock] : (T)ul[j * kBlock];
        } else if (l < nb - j * kBlock) {
          hl[j * kBlock] = (T)ul[j * kBlock];  // constrained shared dofs: value ignored by reduce_shared
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
    for (int k = 0; k < KC; ++k)
#pragma unroll
      for (int q = 0; q < NW; ++q) IX[k][q] = IXn[k][q];
  }
}

template <int n, typename T>
static size_t x_lds_bytes(uint32_t nb_max) {
  constexpr int nd = n * n * n;
  constexpr int CH = 256 / (n * n);
  return (size_t)nb_max * sizeof(double) + (size_t)(3 * CH * nd + n * n) * sizeof(T) + 4 * 256 * sizeof(uint32_t);
}

template <int n, typename T, bool HN>
static hipError_t x_run(const ApplyArgs<T> &a, const double *S, const double *Dt, uint32_t grid, hipStream_t st,
                        bool configure_only, size_t *lds_out, int *occupancy) {
  const size_t lds = x_lds_bytes<n, T>(a.nb_max);
  if (lds_out) *lds_out = lds;
  if (configure_only) {
    hipError_t e = hipFuncSetAttribute((const void *)apply_batches_x<n, T, HN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && occupancy)
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, apply_batches_x<n, T, HN>, 256, lds);
    return e;
  }
  Tables<T, n> tab;
  for (int i = 0; i < ((n + 1) / 2) * n; ++i) {
    tab.S[i] = (T)S[i];
    tab.Dt[i] = (T)Dt[i];
  }
  hipLaunchKernelGGL((apply_batches_x<n, T, HN>), dim3(grid), dim3(256), lds, st, a, tab);
  return hipGetLastError();
}

template <typename T>
hipError_t x_launch(int n, const ApplyArgs<T> &a, const double *S, const double *Dt, bool hn, uint32_t grid,
                    hipStream_t st, bool configure_only, size_t *lds_out, int *occupancy) {
#define X_CASE(N)                                                                                \
  case N:                                                                                        \
    return hn ? x_run<N, T, true>(a, S, Dt, grid, st, configure_only, lds_out, occupancy)        \
              : x_run<N, T, false>(a, S, Dt, grid, st, configure_only, lds_out, occupancy);
  switch (n) {
    X_CASE(2)
    X_CASE(3)
    X_CASE(4)
    X_CASE(5)
    X_CASE(6)
    X_CASE(7)
    default: return hipErrorInvalidValue;
  }
#undef X_CASE
}

template hipError_t x_launch<double>(int, const ApplyArgs<double> &, const double *, const double *, bool, uint32_t,
                                     hipStream_t, bool, size_t *, int *);
template hipError_t x_launch<float>(int, const ApplyArgs<float> &, const double *, const double *, bool, uint32_t,
                                    hipStream_t, bool, size_t *, int *);

}  // namespace mfgpu
