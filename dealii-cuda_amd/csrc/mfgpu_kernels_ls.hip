// Loader / compute specialised cell-loop kernel (two-pass scatter mode).  EXPERIMENTAL (MFGPU_LS=1):
// parity-green, but slower than apply_batches on MI355X (0.285 vs 0.196 ms per launch on the p=4 bench
// workload): two loader waves, capped at 168 VGPRs by the 3 waves/SIMD the layout needs, cannot keep
// enough bytes in flight per CU, so the loaders become the latency-bound stage (profiles/r01_notes.md).
//
// A workgroup has 6 waves: 4 COMPUTE waves run the cell pipeline (mfgpu_cell.cuh) on LDS data only --
// they never issue a global memory instruction and therefore never wait on vmcnt -- and 2 LOADER waves
// own every global load and store of the workgroup's batches:
//   * dof list and source values of batch b+1 (gather, read_dof_values fee_gpu.cuh:323-331) and the
//     coefficient / index stream of the next chunk are in flight while batch b is computed,
//   * the values reach LDS (usrc, Cb, Lb, Mb) at points of the pipeline where the buffers are dead,
//   * after the batch's last cell the loaders move the accumulator to dst / the halo buffer
//     (distribute_local_to_global fee_gpu.cuh:346-363, one write per batch dof) and clear it.
// Every value a loader waits for was requested about one chunk (several microseconds) earlier, so the
// loaders arrive at the pipeline's barriers without stalling the compute waves, and memory latency is
// decoupled from the LDS / VALU work.  In apply_batches (all waves do both) the compiler's in-order
// vmcnt waits made the memory and the cell phase of a workgroup add up (profiles/r01_notes.md).
//
// The loaders execute exactly the barrier sequence of cell_pipeline (loader_shadow below).
#include <hip/hip_runtime.h>

#include "mfgpu_cell.cuh"
#include "mfgpu_kernels.h"

namespace mfgpu {

constexpr int kLsCompute = 256;  // threads of the compute waves
constexpr int kLsLoader = 128;   // threads of the loader waves
constexpr int kLsBlock = kLsCompute + kLsLoader;

// Barrier sequence of cell_pipeline<dim, n, T, HN, WgSync> with an always-true stage_next(), seen from a
// wave that does none of the cell work.  after_first(): usrc, Lb and Mb of the chunk are dead (every
// compute thread read them before its first barrier).  stage(): Cb is dead as well.
template <int dim, bool HN, typename F1, typename F2>
__device__ __forceinline__ void loader_shadow(const bool any_mask, F1 &&after_first, F2 &&stage) {
  if (dim == 3) {
    if (HN && any_mask) {
      __syncthreads();
      after_first();
      __syncthreads();
      __syncthreads();
      __syncthreads();  // after P0
    } else {
      __syncthreads();  // after P0
      after_first();
    }
    __syncthreads();  // after P1
    __syncthreads();  // after P2
    __syncthreads();  // after P3
    __syncthreads();  // after P4: last read of Cb
    stage();
    __syncthreads();  // after P5
    if (HN && any_mask) {
      __syncthreads();
      __syncthreads();
    }
  } else {
    if (HN && any_mask) {
      __syncthreads();
      after_first();
      __syncthreads();
      __syncthreads();  // after P0
    } else {
      __syncthreads();  // after P0
      after_first();
    }
    __syncthreads();  // after P1
    __syncthreads();  // after P2: last read of Cb
    stage();
    if (HN && any_mask) __syncthreads();
    __syncthreads();  // the one guarded by stage_next()
  }
}

template <int n>
constexpr int ls_waves_per_simd() { return n <= 6 ? 3 : 2; }  // 2 workgroups x 6 waves per CU

template <int dim, int n, typename T, bool HN>
__global__ void __launch_bounds__(kLsBlock) __attribute__((amdgpu_waves_per_eu(ls_waves_per_simd<n>())))
apply_batches_ls(const ApplyArgs<T> A, const Tables<T, n> tab) {
  constexpr int nd = (dim == 3) ? n * n * n : n * n;
  constexpr int P = nd / n;
  constexpr int CH = kLsCompute / P;
  constexpr int n2 = n * n;
  constexpr int CHND = CH * nd;
  constexpr int kGL = (max_batch_dofs(256) + kLsLoader - 1) / kLsLoader;
  constexpr int PFL = (CHND + kLsLoader - 1) / kLsLoader;
  static_assert(CH >= 1 && CH <= kLsLoader, "chunk masks are staged by one loader lane per cell");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double *acc = reinterpret_cast<double *>(smem_raw);  // double also in float builds (ds_add_f32 is slow on gfx950)
  T *usrc = reinterpret_cast<T *>(acc + A.nb_max);
  T *Wb = usrc + A.nb_max;
  T *Rb = Wb + CHND;
  T *Cb = Rb + CHND;
  T *Wl = Cb + CHND;
  uint16_t *Lb = reinterpret_cast<uint16_t *>(Wl + n2);
  uint32_t *Mb = reinterpret_cast<uint32_t *>(Lb + ((CHND + 1) & ~1));  // [CH] masks, [2] any-mask per loader wave
  int *meta = reinterpret_cast<int *>(Mb + CH + 2);                      // [0] cells of the batch, [1] another batch follows

  const int tid = threadIdx.x;
  const uint32_t bend = A.batch_end;
  const uint32_t b0 = A.batch0 + blockIdx.x;
  if (b0 >= bend) return;

  if (tid >= kLsCompute) {
    // =========================================================================== loader waves
    const int lt = tid - kLsCompute;
    // hipcc hoists the per-lane constants lt + j * 128 of every unrolled helper loop out of the batch loop
    // and keeps all of them live (18 VGPRs per helper, then spills the dof lists): an opaque copy of the
    // lane index makes them one-instruction temporaries
    auto lane = [&]() {
      int l = lt;
      asm volatile("" : "+v"(l));
      return l;
    };
    struct Meta {
      uint32_t c0, d0, hoff;
      int ncell, nb, nint;
    };
    auto load_meta = [&](uint32_t bb) {
      Meta m;
      m.c0 = A.batch_cell_off[bb];
      m.ncell = (int)(A.batch_cell_off[bb + 1] - m.c0);
      m.d0 = A.batch_dof_off[bb];
      m.nb = (int)(A.batch_dof_off[bb + 1] - m.d0);
      m.nint = (int)A.batch_nint[bb];
      m.hoff = A.halo_off[bb];
      return m;
    };
    // all loads unconditional on clamped indices (a predicated load costs a branch and a full wait)
    auto load_dofs = [&](const Meta &m, uint32_t (&g)[kGL]) {
      const int l = lane();
      const uint32_t *bd = A.bdofs + m.d0;
#pragma unroll
      for (int j = 0; j < kGL; ++j) {
        const int t = l + j * kLsLoader;
        g[j] = bd[t < m.nb ? t : m.nb - 1];
      }
    };
    auto load_src = [&](const uint32_t (&g)[kGL], T (&sv)[kGL]) {
#pragma unroll
      for (int j = 0; j < kGL; ++j) sv[j] = A.src[g[j] & 0x7fffffffu];
    };
    // gather result -> LDS.  bdofs bit 31 = constrained row: reads as 0 (constraint_handler_gpu.cu:258-259)
    // and, if this batch owns the row, dst = src (identity rows, :286).
    auto write_usrc = [&](const uint32_t (&g)[kGL], const T (&sv)[kGL], const Meta &m) {
      const int l = lane();
      T *ul = usrc + l;
#pragma unroll
      for (int j = 0; j < kGL; ++j) {
        const bool con = (g[j] >> 31) != 0;
        if (l < m.nb - j * kLsLoader) {
          ul[j * kLsLoader] = con ? T(0) : sv[j];
          if (con && l < m.nint - j * kLsLoader) {
            T *d = A.dst + (g[j] & 0x7fffffffu);
            *d = A.add ? *d + sv[j] : sv[j];
          }
        }
      }
    };
    // accumulator -> dst (interior dofs: this batch alone touches them) / halo (partial sums of shared
    // dofs, summed by reduce_shared); clears the accumulator for the next batch
    auto scatter = [&](const uint32_t (&g)[kGL], const Meta &m) {
      const int l = lane();
      double *al = acc + l;
      T *hl = A.halo + m.hoff + l - m.nint;
#pragma unroll
      for (int j = 0; j < kGL; ++j) {
        if (l < (int)A.nb_max - j * kLsLoader) {
          const T v = (T)al[j * kLsLoader];
          al[j * kLsLoader] = 0.0;
          if (l < m.nint - j * kLsLoader) {
            if (!(g[j] >> 31)) {
              T *d = A.dst + g[j];
              *d = A.add ? *d + v : v;
            }
          } else if (l < m.nb - j * kLsLoader) {
            hl[j * kLsLoader] = v;  // constrained shared dofs: value ignored by reduce_shared
          }
        }
      }
    };
    T pc[PFL];
    uint16_t pl[PFL];
    uint32_t pm = 0;
    auto prefetch = [&](uint32_t cell0, int cells) {
      const T *cg = A.coef + (size_t)cell0 * nd;
      const uint16_t *lg = A.lmap + (size_t)cell0 * nd;
      const int cnt = cells * nd;
      const int l = lane();
#pragma unroll
      for (int j = 0; j < PFL; ++j) {
        const int i = l + j * kLsLoader;
        const int ic = i < cnt ? i : cnt - 1;
        pc[j] = cg[ic];
        pl[j] = lg[ic];
      }
      if (HN) pm = A.cmask[(size_t)cell0 + (l < cells ? l : cells - 1)];
    };
    auto stage = [&](int cells) {
      const int cnt = cells * nd;
      const int l = lane();
      T *cl = Cb + l;
      uint16_t *ll = Lb + l;
#pragma unroll
      for (int j = 0; j < PFL; ++j) {
        if (l < cnt - j * kLsLoader) {
          cl[j * kLsLoader] = pc[j];
          ll[j * kLsLoader] = pl[j];
        }
      }
      if (HN) {
        const uint32_t mk = lt < cells ? pm : 0u;
        if (lt < CH) Mb[lt] = mk;
        const bool any = __ballot(mk != 0) != 0ull;
        if ((lt & 63) == 0) Mb[CH + (lt >> 6)] = any ? 1u : 0u;
      }
    };
    auto publish = [&](int ncell_, bool more) {
      if (lt == 0) {
        meta[0] = ncell_;
        meta[1] = more ? 1 : 0;
      }
    };

    uint32_t b = b0;
    const uint32_t stride = gridDim.x;
    bool has_nb = b + stride < bend;
    Meta m = load_meta(b);
    Meta mn = has_nb ? load_meta(b + stride) : m;
    uint32_t G[kGL], Gn[kGL];
    T SVn[kGL];
    load_dofs(m, G);
    prefetch(m.c0, m.ncell < CH ? m.ncell : CH);
    load_src(G, SVn);
#pragma unroll
    for (int j = 0; j < kGL; ++j) {
      Gn[j] = 0;
      const int t = lt + j * kLsLoader;
      if (t < (int)A.nb_max) acc[t] = 0.0;
    }
    if (HN)
      for (int t = lt; t < n2; t += kLsLoader) Wl[t] = A.hn_weights[t];
    write_usrc(G, SVn, m);
    stage(m.ncell < CH ? m.ncell : CH);
    publish(m.ncell, has_nb);

    while (true) {
      __syncthreads();  // batch b is staged: the compute waves start
      const bool has_nnb = has_nb && (b + 2 * stride < bend);
      const Meta mnn = has_nnb ? load_meta(b + 2 * stride) : mn;  // scalar loads, used at the rotation
      if (has_nb) load_dofs(mn, Gn);
#pragma unroll
      for (int k = 0; k < kMaxChunks; ++k) {
        const int base = k * CH;
        if (base >= m.ncell) continue;  // uniform
        const int nxt = base + CH;
        const bool last = nxt >= m.ncell;
        int cells_next = 0;
        if (!last) {
          cells_next = m.ncell - nxt < CH ? m.ncell - nxt : CH;
          prefetch(m.c0 + nxt, cells_next);
        } else if (has_nb) {
          cells_next = mn.ncell < CH ? mn.ncell : CH;
          prefetch(mn.c0, cells_next);
        }
        const bool any_mask = HN ? ((Mb[CH] | Mb[CH + 1]) != 0) : false;
        loader_shadow<dim, HN>(
            any_mask, []() {},
            [&]() {
              if (cells_next > 0) stage(cells_next);
              // next batch: gather once its dof list (requested at the top of chunk 0) is here; the values
              // reach LDS a batch's last chunk later (usrc is dead after that chunk's first barrier)
              if (k == 0 && has_nb) load_src(Gn, SVn);
              if (last && has_nb) write_usrc(Gn, SVn, mn);
            });
      }
      __syncthreads();  // accumulator complete
      scatter(G, m);
      if (!has_nb) break;
      publish(mn.ncell, has_nnb);
      b += stride;
      has_nb = has_nnb;
      m = mn;
      mn = mnn;
#pragma unroll
      for (int j = 0; j < kGL; ++j) G[j] = Gn[j];
    }
  } else {
    // =========================================================================== compute waves
    const int lc = tid / P;
    const int pen = tid - lc * P;
    const int pa = (dim == 3) ? pen % n : pen;
    const int pb = (dim == 3) ? pen / n : 0;
    T *Wc = Wb + lc * nd;
    T *Rc = Rb + lc * nd;
    const T *cf = Cb + lc * nd;
    const uint16_t *lm = Lb + lc * nd;
    while (true) {
      __syncthreads();  // batch staged
      const int ncell = meta[0];
      const bool has_nb = meta[1] != 0;
#pragma unroll
      for (int k = 0; k < kMaxChunks; ++k) {
        const int base = k * CH;
        if (base >= ncell) continue;  // uniform
        const bool act = (tid < CH * P) && (base + lc < ncell);
        unsigned mask = 0;
        bool any_mask = false;
        if (HN) {
          if (act) mask = Mb[lc];
          any_mask = (Mb[CH] | Mb[CH + 1]) != 0;
        }
        cell_pipeline<dim, n, T, HN, WgSync>(act, pa, pb, mask, any_mask, usrc, acc, Wc, Rc, cf, lm, Wl, tab,
                                             []() { return true; });
      }
      __syncthreads();  // accumulator complete
      if (!has_nb) break;
    }
  }
}

template <int dim, int n, typename T>
static size_t ls_lds_bytes(uint32_t nb_max) {
  constexpr int nd = (dim == 3) ? n * n * n : n * n;
  constexpr int CH = kLsCompute / (nd / n);
  constexpr int CHND = CH * nd;
  return (size_t)nb_max * sizeof(double) + (size_t)(nb_max + 3 * CHND + n * n) * sizeof(T) +
         (size_t)((CHND + 1) & ~1) * sizeof(uint16_t) +
         (size_t)(CH + 2 + 4) * sizeof(uint32_t);
}

template <int dim, int n, typename T, bool HN>
static hipError_t ls_run(const ApplyArgs<T> &a, const double *S, const double *Dt, uint32_t grid, hipStream_t st,
                         bool configure_only, size_t *lds_out, int *occupancy) {
  const size_t lds = ls_lds_bytes<dim, n, T>(a.nb_max);
  if (lds_out) *lds_out = lds;
  if (configure_only) {
    hipError_t e = hipFuncSetAttribute((const void *)apply_batches_ls<dim, n, T, HN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && occupancy)
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, apply_batches_ls<dim, n, T, HN>, kLsBlock, lds);
    return e;
  }
  Tables<T, n> tab;
  for (int i = 0; i < ((n + 1) / 2) * n; ++i) {
    tab.S[i] = (T)S[i];
    tab.Dt[i] = (T)Dt[i];
  }
  hipLaunchKernelGGL((apply_batches_ls<dim, n, T, HN>), dim3(grid), dim3(kLsBlock), lds, st, a, tab);
  return hipGetLastError();
}

template <typename T>
hipError_t ls_launch(int dim, int n, const ApplyArgs<T> &a, const double *S, const double *Dt, bool hn,
                     uint32_t grid, hipStream_t st, bool configure_only, size_t *lds_out, int *occupancy) {
#define LS_CASE(D, N)                                                                                   \
  case D * 10 + N:                                                                                      \
    return hn ? ls_run<D, N, T, true>(a, S, Dt, grid, st, configure_only, lds_out, occupancy)           \
              : ls_run<D, N, T, false>(a, S, Dt, grid, st, configure_only, lds_out, occupancy);
  switch (dim * 10 + n) {
    LS_CASE(2, 2)
    LS_CASE(2, 3)
    LS_CASE(2, 4)
    LS_CASE(2, 5)
    LS_CASE(2, 6)
    LS_CASE(2, 7)
    LS_CASE(3, 2)
    LS_CASE(3, 3)
    LS_CASE(3, 4)
    LS_CASE(3, 5)
    LS_CASE(3, 6)
    LS_CASE(3, 7)
    default: return hipErrorInvalidValue;
  }
#undef LS_CASE
}

template hipError_t ls_launch<double>(int, int, const ApplyArgs<double> &, const double *, const double *, bool,
                                      uint32_t, hipStream_t, bool, size_t *, int *);
template hipError_t ls_launch<float>(int, int, const ApplyArgs<float> &, const double *, const double *, bool,
                                     uint32_t, hipStream_t, bool, size_t *, int *);

}  // namespace mfgpu
