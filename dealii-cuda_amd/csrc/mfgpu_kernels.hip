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

#include "mfgpu_cell.h"
#include "mfgpu_kernels.h"

namespace mfgpu {

// Workgroup = 256 threads = 4 waves; they process CH = 256/P cells at a time and synchronise the transposes with
// s_barrier (250 of 256 lanes busy for p=4).
// The hanging-node variant needs 258 VGPRs uncapped, one more allocation granule than two waves per
// SIMD allow; LDS already limits the kernel to two workgroups per CU, so cap it there.
template <int n>
constexpr int min_waves_per_simd() { return n <= 5 ? 2 : 1; }

template <int dim, int n, typename T, bool HN, bool TWOPASS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(min_waves_per_simd<n>())))
apply_batches(const ApplyArgs<T> A, const Tables<T, n> tab) {
  constexpr int kBlock = 256;
  constexpr int kGU = (max_batch_dofs(kBlock) + kBlock - 1) / kBlock;  // all gather loads of a batch in flight
  constexpr int nd = (dim == 3) ? n * n * n : n * n;
  constexpr int P = nd / n;          // pencils per cell
  constexpr int NG = 1;                        // independent thread groups per workgroup
  constexpr int GT = kBlock / NG;              // threads per group
  constexpr int CH = GT / P;         // cells a group holds at once
  constexpr int n2 = n * n;
  constexpr int CHND = CH * nd;
  static_assert(CH >= 1, "a cell's pencils must fit into one thread group");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // the accumulator is double also in float builds (ds_add_f32 is slow on gfx950, see apply_batches_x)
  double *acc = reinterpret_cast<double *>(smem_raw);
  T *usrc = reinterpret_cast<T *>(acc + A.nb_max);
  const int tid = threadIdx.x;
  const int grp = 0;
  const int gtid = tid;
  T *Wb = usrc + A.nb_max + grp * 3 * CHND;
  T *Rb = Wb + CHND;
  T *Cb = Rb + CHND;     // folded coefficient of the cells in flight
  T *Wl = usrc + A.nb_max + NG * 3 * CHND;  // hanging-node weights (HN only), broadcast reads
  uint16_t *Lb = reinterpret_cast<uint16_t *>(Wl + n2) + grp * CHND;  // local->batch dof map of the cells in flight

  // Workgroup loops over batches b, b + gridDim.x, ... of [batch0, batch_end) (grid = resident
  // workgroups).
  // XCD-aware batch order (see apply_batches_x): block i runs on XCD i % 8; in every round of gridDim.x
  // batches XCD x takes the x-th contiguous eighth, so the batches in flight on it are mesh neighbours
  // and all XCDs advance through the batch list together: hanging-node batches, which cost more, cluster
  // in space, and one contiguous range per XCD (apply_batches_x) would leave that XCD behind.
  const uint32_t bend = A.batch_end, bstride = gridDim.x;
  uint32_t b;
  {
    const uint32_t G = gridDim.x;
    // block i -> position (i % 8) * G/8 + i / 8 of the round: XCD x takes the x-th contiguous eighth
    b = A.batch0 + (((G & 7u) == 0) ? (blockIdx.x & 7u) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x);
  }
  if (b >= bend) return;
  uint32_t c0, d0, hoff;
  int nb, ncell, nint;
  auto load_meta = [&](uint32_t bb, uint32_t &c0_, int &ncell_, uint32_t &d0_, int &nb_, int &nint_,
                       uint32_t &hoff_) {
    c0_ = A.batch_cell_off[bb];
    ncell_ = (int)(A.batch_cell_off[bb + 1] - c0_);
    d0_ = A.batch_dof_off[bb];
    nb_ = (int)(A.batch_dof_off[bb + 1] - d0_);
    nint_ = TWOPASS ? (int)A.batch_nint[bb] : 0;
    hoff_ = TWOPASS ? A.halo_off[bb] : 0u;
  };
  // All loads are unconditional on clamped indices: predicated loads become one branch +
  // s_waitcnt vmcnt(0) EACH (18 serialized memory round trips in an earlier version).
  auto load_dofs = [&](uint32_t d0_, int nb_, uint32_t (&g_)[kGU], uint8_t (&f_)[kGU]) {
#pragma unroll
    for (int j = 0; j < kGU; ++j) {
      const int t = tid + j * kBlock;
      const int tc = t < nb_ ? t : nb_ - 1;
      g_[j] = __builtin_nontemporal_load(A.bdofs + d0_ + tc);  // read-once streams: keep the L2 for src
      f_[j] = TWOPASS ? (uint8_t)0 : __builtin_nontemporal_load(A.bflags + d0_ + tc);
    }
  };
  auto load_src = [&](const uint32_t (&g_)[kGU], T (&sv_)[kGU]) {
#pragma unroll
    for (int j = 0; j < kGU; ++j) sv_[j] = DBG(2) ? T(0) : A.src[g_[j] & 0x7fffffffu];
  };

  const int lc = gtid / P;
  const int pen = gtid - lc * P;
  const int pa = (dim == 3) ? pen % n : pen;
  const int pb = (dim == 3) ? pen / n : 0;

  // Per-chunk streams (folded coefficient, local->batch index map) are contiguous in plan order.  They
  // are fetched ONE CHUNK AHEAD with fully coalesced loads into registers and staged in LDS, from where
  // every pencil layout reads them.  Loading them per layout straight from global memory touched up to
  // 64 distinct cache lines per wave instruction (x-layout: 40-byte lane stride) and made the CU's L1
  // line-access rate the bottleneck (profiles/, round 1).
  constexpr int PF = (CHND + GT - 1) / GT;
  T pc[PF];
  uint16_t pl[PF];
  auto prefetch = [&](uint32_t cell0, int cnt) {
    const T *cg = A.coef + (size_t)cell0 * nd;
    const uint16_t *lg = A.lmap + (size_t)cell0 * nd;
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int i = gtid + j * GT;
      const int ic = i < cnt ? i : cnt - 1;  // clamped, branch-free
      pc[j] = __builtin_nontemporal_load(cg + ic);
      pl[j] = __builtin_nontemporal_load(lg + ic);
    }
  };
  auto stage = [&](int cnt) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int i = gtid + j * GT;
      if (i < cnt) {
        Cb[i] = pc[j];
        Lb[i] = pl[j];
      }
    }
  };
  if (HN) {
    for (int t = tid; t < n2; t += kBlock) Wl[t] = A.hn_weights[t];
  }
  T *Wc = Wb + lc * nd;
  T *Rc = Rb + lc * nd;
  auto chunk_count = [&](int ncell_, int base_) { return (ncell_ - base_ < CH ? ncell_ - base_ : CH) * nd; };

  // ---- software pipeline over batches: while batch b is computed, ALL global loads of the workgroup's
  // next batch are in flight (its dof list from the start, its source values after the first chunk,
  // its first coefficient / index chunk during b's last chunk).  A workgroup that gathers, computes and
  // scatters strictly in turn keeps HBM idle while it computes: memory (80 us alone) and cell (110 us
  // alone) phases ADDED UP in that version (profiles/r01_notes.md).
  // No global load is issued between the loop top and the first use of the prefetched registers: hipcc
  // waits vmcnt(0) after the loop back-edge, which must find only completed loads.
  uint32_t G[kGU];
  uint8_t F[kGU];
  T SV[kGU];
  load_meta(b, c0, ncell, d0, nb, nint, hoff);
  load_dofs(d0, nb, G, F);
  if (!DBG(8)) prefetch(c0, chunk_count(ncell, 0));
  load_src(G, SV);
  stage(chunk_count(ncell, 0));
  while (true) {
    STAMP(0);
    // ---- 1. gather (read_dof_values, fee_gpu.cuh:323-331, once per batch dof): the values are already
    // in registers.  bdofs bit 31 = constrained row: reads as 0 (constraint_handler_gpu.cu:258-259) and,
    // if this batch owns the row, dst = src is written here (identity rows, :286).
#pragma unroll
    for (int j = 0; j < kGU; ++j) {
      const int t = tid + j * kBlock;
      const bool con = (G[j] >> 31) != 0;
      if (t < nb) {
        usrc[t] = con ? T(0) : SV[j];
        acc[t] = 0.0;
        const bool owner = TWOPASS ? (t < nint) : !(F[j] & kFlagAdd);
        if (con && owner) {
          T *d = A.dst + (G[j] & 0x7fffffffu);
          *d = A.add ? *d + SV[j] : SV[j];
        }
      }
    }
    // next batch of this workgroup: meta data and dof list (the iteration's first global loads)
    const uint32_t bn = b + bstride;
    const bool has_nb = bn < bend;
    uint32_t c0n = c0, d0n = d0, hoffn = hoff;
    int nbn = nb, ncelln = ncell, nintn = nint;
    uint32_t Gn[kGU];
    uint8_t Fn[kGU];
    T SVn[kGU];
    if (has_nb) {
      load_meta(bn, c0n, ncelln, d0n, nbn, nintn, hoffn);
      load_dofs(d0n, nbn, Gn, Fn);
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);

    // ---- 2. cells
    const int ncell_eff = DBG(1) ? 0 : ncell;
    // The chunk loop is fully unrolled (a batch has at most kMaxChunks chunks, enforced by the planner):
    // with a run-time trip count hipcc waits vmcnt(0) for every prefetched value used after the
    // back-edge, which also waits for the long-latency gathers of the next batch; in straight-line code
    // it emits counted waits and younger loads stay in flight.
#pragma unroll
    for (int k = 0; k < kMaxChunks; ++k) {
      const int base = k * CH;
      if (base >= ncell_eff) continue;  // uniform
      const bool act = (gtid < CH * P) && (base + lc < ncell);
      const int nxt = base + CH;
      // coefficient / index stream of the next chunk: of this batch, or the first one of the next batch
      int cnt_next = 0;
      if (nxt < ncell) {
        cnt_next = chunk_count(ncell, nxt);
        if (!DBG(8)) prefetch(c0 + nxt, cnt_next);
      } else if (has_nb) {
        cnt_next = chunk_count(ncelln, 0);
        if (!DBG(8)) prefetch(c0n, cnt_next);
      }
      // source values of the next batch: issued AFTER this chunk's stream prefetch so that the counted
      // wait for the stream leaves these gathers in flight; their dof list was requested a chunk ago
      if (k == 1 && has_nb) load_src(Gn, SVn);
      const T *cf = Cb + lc * nd;
      const uint16_t *lm = Lb + lc * nd;
      unsigned mask = 0;
      bool any_mask = false;
      if (HN) {
        if (act) mask = A.cmask[(size_t)c0 + base + lc];
        any_mask = __syncthreads_or(mask != 0) != 0;
      }
      auto stage_next = [&]() {
        if (cnt_next > 0) stage(cnt_next);
        return cnt_next > 0;
      };
      cell_pipeline<dim, n, T, HN, WgSync>(act, pa, pb, mask, any_mask, usrc, acc, Wc, Rc, cf, lm, Wl, tab, stage_next, A.dbg,
#ifdef MFGPU_STAMPS
                                             (A.stamps && k == 1) ? A.stamps + (size_t)(A.batch_end + b) * 16 : nullptr
#else
                                             nullptr
#endif
        );
      // Wc is next written in P0 of the following chunk and was last read in P4 (P2 in 2D); Rc is next
      // written in P2 (P1) and was last read before the scatter-add: both separated by barriers.
      if (k == 0 && ncell_eff <= CH && has_nb) load_src(Gn, SVn);  // single-chunk batch
      STAMP(4 + k);
    }
    if (ncell_eff == 0 && has_nb) {  // diagnostic builds only (cells skipped)
      prefetch(c0n, chunk_count(ncelln, 0));
      load_src(Gn, SVn);
      stage(chunk_count(ncelln, 0));
    }
    __syncthreads();
    STAMP(13);

    // ---- 3. scatter (distribute_local_to_global fee_gpu.cuh:346-363 + identity rows
    //         constraint_handler_gpu.cu:276-289), one write per batch dof
    if (TWOPASS) {
      // batch dofs are ordered [interior | shared]: interior dofs belong to this batch alone and are
      // final; partial sums of shared dofs go to the batch's contiguous halo slots (reduce_shared)
      T *halo = A.halo + hoff;
      if (!DBG(4)) {
        T old[kGU];
        if (A.add) {  // uniform branch
#pragma unroll
          for (int j = 0; j < kGU; ++j) old[j] = A.dst[G[j] & 0x7fffffffu];
        }
#pragma unroll
        for (int j = 0; j < kGU; ++j) {
          const int t = tid + j * kBlock;
          if (t < nint) {
            if (!(G[j] >> 31)) A.dst[G[j]] = A.add ? old[j] + (T)acc[t] : (T)acc[t];
          } else if (t < nb) {
            halo[t - nint] = (T)acc[t];  // constrained shared dofs: value ignored by reduce_shared
          }
        }
      }
    } else if (!DBG(4)) {
      // later colours (and vmult_add) read-modify-write; the read is unconditional to keep the loads
      // free of per-element branches
      T old[kGU];
#pragma unroll
      for (int j = 0; j < kGU; ++j) old[j] = A.dst[G[j] & 0x7fffffffu];
#pragma unroll
      for (int j = 0; j < kGU; ++j) {
        const int t = tid + j * kBlock;
        if (t < nb && !(G[j] >> 31)) A.dst[G[j]] = ((F[j] & kFlagAdd) || A.add) ? old[j] + (T)acc[t] : (T)acc[t];
      }
    }
    STAMP(15);
    if (!has_nb) break;
    __syncthreads();  // usrc / acc are rewritten for the next batch
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
      F[j] = Fn[j];
      SV[j] = SVn[j];
    }
  }  // batch loop
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
  const int CH = 256 / (nd / n);
  return (size_t)nb_max * sizeof(double) + (size_t)(nb_max + 3 * CH * nd + n * n) * sizeof(T) +
         (size_t)CH * nd * sizeof(uint16_t);
}

template <int dim, int n, typename T, bool HN, bool TP>
static hipError_t launch_k(const ApplyArgs<T> &a, const Tables<T, n> &tab, size_t lds, uint32_t grid, hipStream_t st) {
  hipLaunchKernelGGL((apply_batches<dim, n, T, HN, TP>), dim3(grid), dim3(256), lds, st, a, tab);
  return hipGetLastError();
}

// dispatch over the run-time switches (hanging nodes, scatter mode)
#define MFGPU_SWITCH(FN, ...)                                             \
  (hn ? (twopass ? FN<dim, n, T, true, true>(__VA_ARGS__)                  \
                 : FN<dim, n, T, true, false>(__VA_ARGS__))                \
      : (twopass ? FN<dim, n, T, false, true>(__VA_ARGS__)                 \
                 : FN<dim, n, T, false, false>(__VA_ARGS__)))

template <int dim, int n, typename T>
static hipError_t launch_t(const ApplyArgs<T> &a, const double *S, const double *Dt, bool hn,
                           bool twopass, uint32_t grid, hipStream_t st) {
  Tables<T, n> tab;
  for (int i = 0; i < ((n + 1) / 2) * n; ++i) {
    tab.S[i] = (T)S[i];
    tab.Dt[i] = (T)Dt[i];
  }
  const size_t lds = lds_bytes_t<dim, n, T>(a.nb_max);
  return MFGPU_SWITCH(launch_k, a, tab, lds, grid, st);
}

template <int dim, int n, typename T, bool HN, bool TP>
static hipError_t configure_k(size_t lds) {
  return hipFuncSetAttribute((const void *)apply_batches<dim, n, T, HN, TP>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
template <int dim, int n, typename T>
static hipError_t configure_t(size_t lds) {
  hipError_t e = hipSuccess;
  for (int hn_ = 0; hn_ < 2 && e == hipSuccess; ++hn_)
    for (int tp_ = 0; tp_ < 2 && e == hipSuccess; ++tp_) {
      const bool hn = hn_, twopass = tp_;
      e = MFGPU_SWITCH(configure_k, lds);
    }
  return e;
}

template <int dim, int n, typename T, bool HN, bool TP>
static hipError_t occupancy_k(size_t lds, int *blocks) {
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, apply_batches<dim, n, T, HN, TP>, 256, lds);
}
template <int dim, int n, typename T>
static hipError_t occupancy_t(bool hn, bool twopass, size_t lds, int *blocks) {
  return MFGPU_SWITCH(occupancy_k, lds, blocks);
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
hipError_t apply_occupancy(int dim, int n, bool hn, bool twopass, size_t lds, int *blocks) {
#define CALL(D, N) occupancy_t<D, N, T>(hn, twopass, lds, blocks)
  MFGPU_DISPATCH(CALL)
#undef CALL
}

template <typename T>
hipError_t apply_launch(int dim, int n, const ApplyArgs<T> &a, const double *S, const double *Dt,
                        bool hn, bool twopass, uint32_t grid, hipStream_t st) {
#define CALL(D, N) launch_t<D, N, T>(a, S, Dt, hn, twopass, grid, st)
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
  template hipError_t apply_occupancy<T>(int, int, bool, bool, size_t, int *);                          \
  template hipError_t apply_launch<T>(int, int, const ApplyArgs<T> &, const double *, const double *,   \
                                      bool, bool, uint32_t, hipStream_t);                               \
  template hipError_t orphan_launch<T>(T *, const T *, const uint32_t *, uint32_t, int, hipStream_t);   \
  template hipError_t coefficient_launch<T>(T *, const T *, size_t, int, hipStream_t);                  \
  template hipError_t fold_launch<T>(T *, const T *, const T *, const T *, const uint32_t *, uint32_t,  \
                                     uint32_t, hipStream_t);                                            \
  template hipError_t fill_launch<T>(T *, size_t, T, hipStream_t);
INST(double)
INST(float)

}  // namespace mfgpu
