// 3D cell-loop kernel, two-pass scatter mode, uniform-Jacobian path: a thread owns a 2D PLANE of a cell.
//
// Why (profiles/r01_notes.md, profiles/r02_notes.md): the pencil kernels (apply_batches, apply_batches_x) move every
// cell value through LDS ten times (7 barrier-separated transposes + coefficient staging) and are bound by that
// chain at 2-3 waves per SIMD.  Here a thread holds n x n values in registers, so two of the three tensor
// directions are register mat-vecs and a cell needs TWO LDS transposes of two arrays:
//
//   stage A (xy-plane, nodal z = k):   t = S_y u,  a = S_x t,  b = D_x a                     -> LDS (a, b)
//   stage B (yz-plane, quad  x = k):   gx = S_z b;  v = S_z a;  gy = D_y v;  gz = D_z v
//                                      r  = D_y^T (c gy) + D_z^T (c gz);  r' = S_z^T r;  t' = S_z^T (c gx)
//                                                                                            -> LDS (r', t') in place
//   stage C (xy-plane, nodal z = k):   out = S_y^T S_x^T (r' + D_x^T t')                      -> batch accumulator
//
// i.e. the x-derivative is taken BEFORE the z-interpolation (they commute), so all three gradient components meet
// the coefficient in ONE layout: the coefficient goes straight from HBM into registers (stored per batch as
// [n*n rows][tasks], coalesced), never through LDS.  14 contractions per cell (reference: 18, tensor_ops.cuh:179-261;
// pencil kernels: 12) for 5.5 LDS stores per value instead of 10.
//
// One wave owns CW = 64 / n cells (n lanes per cell); both transposes are cell-local, hence wave-local: no
// s_barrier in the cell phase.  One workgroup = ONE wave = one batch (<= CW cells) in this version: the gather /
// scatter staging is wave-local too and a CU runs four independent workgroups, one per SIMD, each with up to 512
// registers.  Everything a batch needs from HBM (dof list, source values, index runs, coefficient rows) is in
// flight one batch ahead (the dof list two), issued unconditionally at the top of the iteration and consumed
// after ONE wait at its end, before any store of the iteration is issued (vmcnt is in-order and counts stores).
//
// Algebra per cell: fee_gpu.cuh:219-284 (uniform-Jacobian branch), tensor_ops.cuh:179-261;
// gather / scatter: fee_gpu.cuh:323-363; constrained rows: constraint_handler_gpu.cu:247-289.
#include <hip/hip_runtime.h>

#include "mfgpu_cell.h"
#include "mfgpu_kernels.h"
#include "mfgpu_planes.h"

namespace mfgpu {

// Only LDS instructions may cross: pins a global memory operation between two compute steps (the
// scheduler would otherwise issue all of them first), while the next step's LDS reads may still be hoisted
#define MFGPU_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x380)

// HN: the batches of cells WITH a hanging-node mask.  The cells' constrained nodes have private entries behind the
// batch array's dof-list part (mfgpu_api.hip); on them the interpolation passes of resolve_hanging_nodes
// (hanging_nodes.cuh:617-696: along x, then y, then z, one flagged line per lane) run before stage A, and the
// transposed passes in reverse order (the exact adjoint) after stage C, each line as a plain n x n mat-vec with the
// weight matrix W (a line of a cell whose type bit is clear is listed in reverse, which mirrors W).  The cell stages
// themselves do not know about hanging nodes.
template <int n, typename T, bool ADD, bool HN>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
apply_planes3(const ApplyArgs<T> A, const TablesEO<T, n> tab) {
  constexpr int n2 = n * n;
  constexpr int CW = p_cells_per_wave(n);  // cells per wave
  constexpr int NT = CW * n;               // tasks (active lanes) of a full batch
  constexpr int KGU = p_kgu(n);            // 64-lane slots of a batch's dof list:
  constexpr int JI = p_ji(n);              //   JI slots of interior dofs, then HS = KGU - JI slots of pass-2 dofs
  constexpr int HS = KGU - JI;
  constexpr int SA = p_cell_stride(n);     // padded cell stride of the transpose arrays
  constexpr int NIW = (n2 + 1) / 2;        // 32-bit words of a task's packed index run
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int PRIV = HN ? p_priv_max(n) : 0;        // private entries of the hanging-node batches
  double *ua = reinterpret_cast<double *>(smem_raw);  // gathered source values, then the accumulator
  T *Aw = reinterpret_cast<T *>(ua + KGU * 64 + PRIV + (HN ? n2 : 0));  // CW cells + one scratch cell for the idle lanes
  T *Bw = Aw + (CW + 1) * SA;
  double *Wl = ua + KGU * 64 + PRIV;  // HN: the weight matrix W[i * n + k] (hanging_nodes.cuh:580-598)

  const int lane = threadIdx.x;
  if (HN) {
    if (lane < n2) Wl[lane] = (double)A.hn_weights[lane];
  }
  // The idle lanes (NT .. 63) and the tasks of cells a ragged batch does not have run the same instruction stream on
  // harmless data: the idle lanes own a scratch cell of the transpose arrays, a missing task's coefficient rows are
  // zero and its index run points at the batch array's last slot (never a dof: mfgpu_api.hip); only the adds into
  // the accumulator are masked for the idle lanes.  No branch in the cell phase.
  const int lc = lane / n, k = lane - lc * n;  // lc == CW for the idle lanes
  const bool lane_on = lane < NT;
  const int tk = lane_on ? lane : NT - 1;  // idle lanes load a valid entry

  // persistent workgroups, XCD-aware contiguous batch ranges (see apply_batches_x)
  uint32_t b, bstride, bend;
  {
    const uint32_t nbt = A.batch_end - A.batch0 - A.hole_len, G = gridDim.x;
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
      bend = A.batch_end - A.hole_len;
      bstride = G;
    }
  }
  if (b >= bend) return;
  // The ranges above count the launch's batches without its hole [hole0, hole0 + hole_len) (mfgpu_vmult_dist_begin:
  // the batches on the two interface planes of a slab in ONE launch; hole_len = 0 otherwise); b, b1, .. are batch
  // numbers proper.  Scalar arithmetic.
  const uint32_t hole0 = A.hole0, hole_len = A.hole_len;
  auto next_of = [&](uint32_t x) {
    const uint32_t l = (x >= hole0 ? x - hole_len : x) + bstride;
    return l < bend ? (l >= hole0 ? l + hole_len : l) : x;
  };
  if (b >= hole0) b += hole_len;

  // Per-batch records have FIXED sizes and a fixed structure (mfgpu_api.hip): every address below is a uniform base
  // (scalar arithmetic on the batch index) plus a lane offset plus an immediate; nothing is clamped per batch and
  // there is no per-batch metadata at all.  Vectors and the halo buffer are addressed base + 32-bit byte offset
  // (the plan guarantees n_dofs < 2^29; shifting a dof-list entry left by 3 also drops its flag bit 31).
  auto load_dofs = [&](uint32_t bb, uint32_t (&g)[KGU]) {
    const uint32_t *p = A.bdofsp + (size_t)bb * (KGU * 64) + lane;
#pragma unroll
    for (int j = 0; j < KGU; ++j) g[j] = nt_load(p + j * 64);
  };
  auto src_at = [&](uint32_t g) -> T {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(A.src) + (uint32_t)(g * (uint32_t)sizeof(T)));
  };
  auto dst_at = [&](uint32_t g) -> T * {
    return reinterpret_cast<T *>(reinterpret_cast<char *>(A.dst) + (uint32_t)(g * (uint32_t)sizeof(T)));
  };
  auto load_ix = [&](uint32_t bb, uint32_t (&ix)[NIW]) {
    const uint32_t *p = A.idxp + (size_t)bb * (NIW * NT) + tk;
#pragma unroll
    for (int w = 0; w < NIW; ++w) ix[w] = nt_load(p + w * NT);
  };
  // gathered values -> LDS (read_dof_values, fee_gpu.cuh:323-331, once per batch dof).  bdofs bit 31 = constrained
  // row: reads as 0 (constraint_handler_gpu.cu:258-259).  All KGU * 64 slots are written (the padding of the dof
  // list repeats its last entry).  The identity rows themselves (dst = src, :286) are written by pass 2 for
  // EVERY constrained dof (PlanLimits::pass2_owns_constrained): this kernel never stores to a constrained row.
  auto stage_src = [&](const uint32_t (&g)[KGU], const T (&sv)[KGU]) {
#pragma unroll
    for (int j = 0; j < KGU; ++j) {
      double v = (double)sv[j];
      if (j >= JI) {  // constrained dofs sit in the pass-2 slots only
        // value & ~(sign of g): bit arithmetic instead of a select, which hipcc turns into a branch per slot
        const unsigned long long keep = (unsigned long long)(long long)~((int)g[j] >> 31);
        v = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(v) & keep));
      }
      ua[lane + j * 64] = v;
    }
  };
  // byte offset into ua of entry i = x + n * y of this task's index run (stored pre-multiplied by 8)
  auto ixb = [&](const uint32_t (&ix)[NIW], int i) -> uint32_t {
    return (i & 1) ? (ix[i >> 1] >> 16) : (ix[i >> 1] & 0xffffu);
  };
  // scatter (distribute_local_to_global, fee_gpu.cuh:346-363), ONE wave-wide store per 64-lane slot and no case
  // distinction at run time: the first JI slots hold interior dofs -- the batch's alone, final, never constrained --
  // and go to dst (padding lanes store the 0 of their untouched accumulator slot, or old + 0, to a pass-2 dof of the
  // batch, which pass 2 rewrites); the other HS slots are partial sums for pass 2 and go to the batch's HS * 64 halo
  // slots (padding lanes: unused halo slots).
  auto scatter_slot = [&](int j, uint32_t bb, uint32_t g, T r, T oldv) {
    if (j < JI) {
      *dst_at(g) = ADD ? oldv + r : r;
    } else {
      T *const hp = A.halo + (size_t)bb * (HS * 64) + lane;
      hp[(j - JI) * 64] = r;  // (constrained dofs: value ignored by pass 2)
    }
  };

  uint32_t b1 = next_of(b), b2 = next_of(b1);
  uint32_t Gp[KGU], Gc[KGU], Gn[KGU], Gnn[KGU];  // dof lists: previous (its scatter is deferred), current, two ahead
  uint32_t IXc[NIW], IXn[NIW];
  constexpr int CR = PRIV / 64, HROWS = HN ? p_hn_rows(n) : 1;
  uint32_t Hc[HROWS], Hn[HROWS];  // HN: this batch's and the next batch's hanging-node record
  auto load_hn = [&](uint32_t bb, uint32_t (&H)[HROWS]) {
    const uint32_t slot = A.hn_slot[bb];  // uniform
    if (slot != 0xffffffffu) {
      const uint32_t *p = A.hnrec + (size_t)slot * (HROWS * 64) + lane;
#pragma unroll
      for (int w = 0; w < HROWS; ++w) H[w] = nt_load(p + w * 64);
    } else {  // a batch of cells without a mask: no copies, no line operations
#pragma unroll
      for (int w = 0; w < HROWS; ++w) H[w] = 0u;
    }
  };
  T Cc[n2];
  T SVn[KGU], R[KGU], old[KGU];
  load_dofs(b, Gc);
  load_dofs(b1, Gn);
  load_ix(b, IXc);
  if (HN) load_hn(b, Hc);
  // the idle lanes add into the batch array's last slot, which is never a dof
  constexpr uint32_t kDummyIx = 8u * (uint32_t)(KGU * 64 - 1) * 0x10001u;
#pragma unroll
  for (int w = 0; w < NIW; ++w) IXc[w] = lane_on ? IXc[w] : kDummyIx;
  {
    const T *p = A.coefp + (size_t)b * (n2 * NT) + tk;
#pragma unroll
    for (int r = 0; r < n2; ++r) Cc[r] = nt_load(p + r * NT);
  }
#pragma unroll
  for (int j = 0; j < KGU; ++j) {
    SVn[j] = src_at(Gc[j]);
    Gp[j] = Gc[j];
    R[j] = T(0);
    old[j] = T(0);
  }
  stage_src(Gc, SVn);
  WaveSync::sync();

  // LDS bases of this task: xy-plane with z = k (stages A, C) and yz-plane with x = k (stage B)
  const int pxy = lc * SA + n2 * k;  // + x + n * y
  const int pyz = lc * SA + k;       // + n * y + n2 * z
  // the batch whose results wait in R for their deferred scatter.  (First iteration: this batch itself with R = 0 --
  // zeros, or old + 0, go where the real results follow one iteration later; a condition here would be a branch, and
  // hipcc merges the branches of neighbouring steps by moving the contractions between them out of the way.)
  uint32_t bp = b;

  while (true) {
    const bool has_next = b1 != b;
    // ---- hanging-node batches: private copies of the constrained nodes, then the interpolation passes x, y, z.
    // The batch's record (copies, line operations, counts) was requested one batch ahead with the other coalesced
    // loads; lanes beyond a count work on nothing.
    constexpr int kH2 = HN ? HROWS - 2 : 0, kH1 = HN ? HROWS - 1 : 0;  // the record's two rows of counts
    const uint32_t hn_ncopy = HN ? (uint32_t)__builtin_amdgcn_readfirstlane(Hc[kH2] & 0xffffu) : 0u;
    auto hn_count = [&](int dir) -> uint32_t {
      const uint32_t w = dir == 0 ? Hc[kH2] >> 16 : dir == 1 ? Hc[kH1] & 0xffffu : Hc[kH1] >> 16;
      return (uint32_t)__builtin_amdgcn_readfirstlane(w);
    };
    // (Wr: the weight matrix in registers, read from LDS once per group of three passes instead of 25 broadcast reads
    // per line operation, which tripled the LDS instructions of a hanging-node batch)
    auto hn_pass = [&](int dir, bool transposed, const double (&Wr)[n2]) {
      const uint32_t count = hn_count(dir);
#pragma unroll
      for (int r = 0; r < kHnOpRounds; ++r) {
        if ((uint32_t)(r * 64) >= count) break;  // uniform
        const bool on = (uint32_t)(r * 64 + lane) < count;
        const uint32_t *ww = &Hc[CR + (dir * kHnOpRounds + r) * 3];
        double v[n], o[n];
        uint32_t pos[n];
#pragma unroll
        for (int t = 0; t < n; ++t) {
          pos[t] = on ? (ww[t >> 1] >> (16 * (t & 1))) & 0xffffu : (uint32_t)(KGU * 64 + PRIV - 1);
          v[t] = ua[pos[t]];
        }
#pragma unroll
        for (int i = 0; i < n; ++i) {
          double acc = 0.0;
#pragma unroll
          for (int k2 = 0; k2 < n; ++k2) acc = fma(transposed ? Wr[k2 * n + i] : Wr[i * n + k2], v[k2], acc);
          o[i] = acc;
        }
        if (on) {
#pragma unroll
          for (int i = 0; i < n; ++i) ua[pos[i]] = o[i];
        }
      }
    };
    if (HN && hn_ncopy) {  // (uniform; a batch of cells without a mask has no copies)
      double Wr[n2];
#pragma unroll
      for (int i = 0; i < n2; ++i) Wr[i] = Wl[i];
      // all reads, then all writes: rounds beyond the count run with no lane active instead of ending the loop on a
      // uniform branch, which would put an LDS round trip between every two of them
      double cv[CR > 0 ? CR : 1];
#pragma unroll
      for (int r = 0; r < CR; ++r) cv[r] = (uint32_t)(r * 64 + lane) < hn_ncopy ? ua[Hc[r] & 0xffffu] : 0.0;
#pragma unroll
      for (int r = 0; r < CR; ++r)
        if ((uint32_t)(r * 64 + lane) < hn_ncopy) ua[Hc[r] >> 16] = cv[r];
      WaveSync::sync();
      hn_pass(0, false, Wr);
      WaveSync::sync();
      hn_pass(1, false, Wr);
      WaveSync::sync();
      hn_pass(2, false, Wr);
      WaveSync::sync();
    }
    STAMP(0);
    RSTAMP(8);
    WGSTAMP(10);
    // ---- coalesced loads of the coming batches: dof list two ahead, index runs one ahead
    load_dofs(b2, Gnn);
    load_ix(b1, IXn);
    if (HN) load_hn(b1, Hn);
    if (ADD) {
#pragma unroll
      for (int j = 0; j < JI; ++j) old[j] = *dst_at(Gp[j]);
    }
    STAMP(1);
    // The scattered accesses (gather of the next batch's source values, stores of the PREVIOUS batch's results) and
    // the coefficient rows are spread over the compute steps below, one or two per step: a 64-address gather or
    // scatter occupies the memory pipeline for ~150 cycles and a lone wave per SIMD has nothing else to hide it with.
    // Stage B is where the arithmetic is (8 of the 14 contractions, 25 steps): it takes the scatter AND the gather, one
    // or two instructions per step; stage A, short and behind the burst of coalesced loads above, takes none (with the
    // gather in stage A, two per step: +4 us per vmult on C2, profiles/r03_notes.md section 10).
    auto hookA = [&](int) {};
    auto hookB = [&](int s) {  // 5 n steps: the previous batch's scatter, the gather of the next batch
      MFGPU_PIN_VMEM();
#pragma unroll
      for (int j = (KGU * s) / (5 * n); j < (KGU * (s + 1)) / (5 * n); ++j)
        scatter_slot(j, bp, Gp[j], R[j], old[j]);
#pragma unroll
      for (int j = (KGU * s) / (5 * n); j < (KGU * (s + 1)) / (5 * n); ++j)
        SVn[j] = src_at(Gn[j]);
      MFGPU_PIN_VMEM();
    };
    const T *const cnext = A.coefp + (size_t)b1 * (n2 * NT) + tk;
    auto hookC = [&](int s) {  // 2 n steps: the next batch's coefficient rows (stage B is done with this batch's)
      MFGPU_PIN_VMEM();
#pragma unroll
      for (int r = (n2 * s) / (2 * n); r < (n2 * (s + 1)) / (2 * n); ++r)
        Cc[r] = nt_load(cnext + r * NT);
      MFGPU_PIN_VMEM();
    };

    // ---- stage A: gather the plane, S_y, then per line S_x and D_x
    {
      T u[n2];
      // (read in the order the first contraction consumes: line x = 0 first)
#pragma unroll
      for (int x = 0; x < n; ++x)
#pragma unroll
        for (int y = 0; y < n; ++y)
          u[x + n * y] = (T) * reinterpret_cast<const double *>(reinterpret_cast<const char *>(ua) + ixb(IXc, x + n * y));
#pragma unroll
      for (int x = 0; x < n; ++x) {
        T in[n], out[n];
        get_line<n, 1>(u, x, in);
        eo_apply<n, 0>(tab, in, out);
        set_line<n, 1>(u, x, out);
        hookA(x);
      }
#pragma unroll
      for (int y = 0; y < n; ++y) {
        T in[n], a[n], bb[n];
        get_line<n, 0>(u, y, in);
        eo_apply<n, 0>(tab, in, a);
        eo_apply<n, 2>(tab, a, bb);
#pragma unroll
        for (int x = 0; x < n; ++x) {
          Aw[pxy + x + n * y] = a[x];
          Bw[pxy + x + n * y] = bb[x];
        }
        hookA(n + y);
      }
    }
    WaveSync::sync();
    STAMP(2);
    // every gather of the batch is done: the array becomes the accumulator
#pragma unroll
    for (int j = 0; j < KGU + PRIV / 64; ++j) ua[lane + j * 64] = 0.0;

    // ---- stage B: plane (y, z) at quadrature index x = k.  The LDS reads of a line are issued one step ahead of
    // its contractions (a lone wave per SIMD has nothing else to cover the LDS latency with).
    {
      T nx[n];  // the next line to be processed, in flight
#pragma unroll
      for (int z = 0; z < n; ++z) nx[z] = Bw[pyz + n2 * z];
      // x-derivative part: t' = S_z^T (c .* S_z b)
#pragma unroll
      for (int y = 0; y < n; ++y) {
        T in[n], g[n], o[n];
#pragma unroll
        for (int z = 0; z < n; ++z) in[z] = nx[z];
#pragma unroll
        for (int z = 0; z < n; ++z) nx[z] = y + 1 < n ? Bw[pyz + n * (y + 1) + n2 * z] : Aw[pyz + n2 * z];
        eo_apply<n, 0>(tab, in, g);
#pragma unroll
        for (int z = 0; z < n; ++z) g[z] *= Cc[y + n * z];
        eo_apply<n, 1>(tab, g, o);
#pragma unroll
        for (int z = 0; z < n; ++z) Bw[pyz + n * y + n2 * z] = o[z];
        hookB(y);
      }
      // values at the quadrature points
      T v[n2], r[n2];
#pragma unroll
      for (int y = 0; y < n; ++y) {
        T in[n], o[n];
#pragma unroll
        for (int z = 0; z < n; ++z) in[z] = nx[z];
        if (y + 1 < n) {
#pragma unroll
          for (int z = 0; z < n; ++z) nx[z] = Aw[pyz + n * (y + 1) + n2 * z];
        }
        eo_apply<n, 0>(tab, in, o);
        set_line<n, 1>(v, y, o);
        hookB(n + y);
      }
      // z-derivative part
#pragma unroll
      for (int y = 0; y < n; ++y) {
        T in[n], g[n], o[n];
        get_line<n, 1>(v, y, in);
        eo_apply<n, 2>(tab, in, g);
#pragma unroll
        for (int z = 0; z < n; ++z) g[z] *= Cc[y + n * z];
        eo_apply<n, 3>(tab, g, o);
        set_line<n, 1>(r, y, o);
        hookB(2 * n + y);
      }
      // y-derivative part
#pragma unroll
      for (int z = 0; z < n; ++z) {
        T in[n], g[n], o[n];
        get_line<n, 0>(v, z, in);
        eo_apply<n, 2>(tab, in, g);
#pragma unroll
        for (int y = 0; y < n; ++y) g[y] *= Cc[y + n * z];
        eo_apply<n, 3>(tab, g, o);
#pragma unroll
        for (int y = 0; y < n; ++y) r[y + n * z] += o[y];
        hookB(3 * n + z);
      }
      // r' = S_z^T r
#pragma unroll
      for (int y = 0; y < n; ++y) {
        T in[n], o[n];
        get_line<n, 1>(r, y, in);
        eo_apply<n, 1>(tab, in, o);
#pragma unroll
        for (int z = 0; z < n; ++z) Aw[pyz + n * y + n2 * z] = o[z];
        hookB(4 * n + y);
      }
    }
    WaveSync::sync();
    STAMP(3);

    // ---- stage C: plane (x, y) at nodal z = k; add into the batch accumulator
    {
      T o[n2];
      T nr[n], nt[n];  // the next line, in flight
#pragma unroll
      for (int x = 0; x < n; ++x) {
        nr[x] = Aw[pxy + x];
        nt[x] = Bw[pxy + x];
      }
#pragma unroll
      for (int y = 0; y < n; ++y) {
        T rp[n], tp[n], w[n], ol[n];
#pragma unroll
        for (int x = 0; x < n; ++x) {
          rp[x] = nr[x];
          tp[x] = nt[x];
        }
        if (y + 1 < n) {
#pragma unroll
          for (int x = 0; x < n; ++x) {
            nr[x] = Aw[pxy + x + n * (y + 1)];
            nt[x] = Bw[pxy + x + n * (y + 1)];
          }
        }
        eo_apply<n, 3>(tab, tp, w);
#pragma unroll
        for (int x = 0; x < n; ++x) w[x] += rp[x];
        eo_apply<n, 1>(tab, w, ol);
        set_line<n, 0>(o, y, ol);
        hookC(y);
      }
#pragma unroll
      for (int x = 0; x < n; ++x) {
        T in[n], out[n];
        get_line<n, 1>(o, x, in);
        eo_apply<n, 1>(tab, in, out);
        // a finished line goes to the accumulator at once: its LDS atomics run under the next line's contraction
#pragma unroll
        for (int y = 0; y < n; ++y)
          lds_add(reinterpret_cast<double *>(reinterpret_cast<char *>(ua) + ixb(IXc, x + n * y)), (double)out[y]);
        hookC(n + x);
      }
    }
    WaveSync::sync();
    STAMP(4);

    if (HN && hn_ncopy) {
      // the transposed passes in reverse order, then the private entries' sums go to their dofs' entries
      double Wr[n2];
#pragma unroll
      for (int i = 0; i < n2; ++i) Wr[i] = Wl[i];
      hn_pass(2, true, Wr);
      WaveSync::sync();
      hn_pass(1, true, Wr);
      WaveSync::sync();
      hn_pass(0, true, Wr);
      WaveSync::sync();
      double cv[CR > 0 ? CR : 1];
#pragma unroll
      for (int r = 0; r < CR; ++r) cv[r] = (uint32_t)(r * 64 + lane) < hn_ncopy ? ua[Hc[r] >> 16] : 0.0;
#pragma unroll
      for (int r = 0; r < CR; ++r)
        if ((uint32_t)(r * 64 + lane) < hn_ncopy) lds_add(ua + (Hc[r] & 0xffffu), cv[r]);
      WaveSync::sync();
    }
    // ---- batch results -> registers (stored during the next iteration); the next batch's gathered values -> LDS:
    // the iteration's one wait for the gather, issued a stage and a half ago
#pragma unroll
    for (int j = 0; j < KGU; ++j) R[j] = (T)ua[lane + j * 64];
    WaveSync::sync();
    STAMP(5);
    stage_src(Gn, SVn);  // (after the last batch: its own values again, unused)
    STAMP(6);
    bp = b;
#pragma unroll
    for (int j = 0; j < KGU; ++j) Gp[j] = Gc[j];
    STAMP(7);
    RSTAMP(9);
    if (!has_next) break;
    WaveSync::sync();
    b = b1;
    b1 = b2;
    b2 = next_of(b2);
#pragma unroll
    for (int j = 0; j < KGU; ++j) {
      Gc[j] = Gn[j];
      Gn[j] = Gnn[j];
    }
#pragma unroll
    for (int w = 0; w < NIW; ++w) IXc[w] = lane_on ? IXn[w] : kDummyIx;
    if (HN) {
#pragma unroll
      for (int w = 0; w < HROWS; ++w) Hc[w] = Hn[w];
    }
  }
  // the last batch's results
  if (ADD) {
#pragma unroll
    for (int j = 0; j < JI; ++j) old[j] = *dst_at(Gp[j]);
  }
#pragma unroll
  for (int j = 0; j < KGU; ++j) scatter_slot(j, bp, Gp[j], R[j], old[j]);
}

template <int n, typename T>
static size_t p_lds_bytes(bool hn) {
  return ((size_t)p_kgu(n) * 64 + (hn ? (size_t)p_priv_max(n) + n * n : 0)) * sizeof(double) +
         (size_t)2 * (p_cells_per_wave(n) + 1) * p_cell_stride(n) * sizeof(T);
}

template <int n, typename T>
static hipError_t p_run(const ApplyArgs<T> &a, const double *S, const double *Dt, bool hn, uint32_t grid, hipStream_t st,
                        bool configure_only, size_t *lds_out, int *occupancy) {
  const size_t lds = p_lds_bytes<n, T>(hn);
  if (lds_out) *lds_out = lds;
  if (configure_only) {
    const void *f0 = hn ? (const void *)apply_planes3<n, T, false, true> : (const void *)apply_planes3<n, T, false, false>;
    const void *f1 = hn ? (const void *)apply_planes3<n, T, true, true> : (const void *)apply_planes3<n, T, true, false>;
    hipError_t e = hipFuncSetAttribute(f0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(f1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && occupancy)
      e = hn ? hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, apply_planes3<n, T, false, true>, 64, lds)
             : hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, apply_planes3<n, T, false, false>, 64, lds);
    return e;
  }
  const TablesEO<T, n> tab = make_tables_eo<T, n>(S, Dt);
  if (hn) {
    if (a.add)
      hipLaunchKernelGGL((apply_planes3<n, T, true, true>), dim3(grid), dim3(64), lds, st, a, tab);
    else
      hipLaunchKernelGGL((apply_planes3<n, T, false, true>), dim3(grid), dim3(64), lds, st, a, tab);
  } else {
    if (a.add)
      hipLaunchKernelGGL((apply_planes3<n, T, true, false>), dim3(grid), dim3(64), lds, st, a, tab);
    else
      hipLaunchKernelGGL((apply_planes3<n, T, false, false>), dim3(grid), dim3(64), lds, st, a, tab);
  }
  return hipGetLastError();
}

template <typename T>
hipError_t p_launch(int n, const ApplyArgs<T> &a, const double *S, const double *Dt, bool hn, uint32_t grid,
                    hipStream_t st, bool configure_only, size_t *lds_out, int *occupancy) {
  switch (n) {
    case 3: return p_run<3, T>(a, S, Dt, hn, grid, st, configure_only, lds_out, occupancy);
    case 4: return p_run<4, T>(a, S, Dt, hn, grid, st, configure_only, lds_out, occupancy);
    case 5: return p_run<5, T>(a, S, Dt, hn, grid, st, configure_only, lds_out, occupancy);
    default: return hipErrorInvalidValue;
  }
}

template hipError_t p_launch<double>(int, const ApplyArgs<double> &, const double *, const double *, bool, uint32_t,
                                     hipStream_t, bool, size_t *, int *);
template hipError_t p_launch<float>(int, const ApplyArgs<float> &, const double *, const double *, bool, uint32_t,
                                    hipStream_t, bool, size_t *, int *);

// coefficient in plan cell order [cell][q] -> per batch [row r = y + n z][NT tasks = cell_in_batch * n + x]
// (fixed record size n*n*NT per batch; the tasks of a ragged batch's missing cells stay zero)
template <typename T>
__global__ void relayout_coef_kernel(T *out, const T *in, const uint32_t *cell_batch, const uint32_t *cell_pos,
                                     size_t total, int n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int n2 = n * n, nd = n2 * n, NT = (64 / n) * n;
  const size_t cell = i / nd;
  const int q = (int)(i - cell * nd);
  const int x = q % n, r = q / n;  // r = y + n z
  out[((size_t)cell_batch[cell] * n2 + r) * NT + cell_pos[cell] * n + x] = in[i];
}

template <typename T>
hipError_t relayout_coef_launch(T *out, const T *in, const uint32_t *cell_batch, const uint32_t *cell_pos,
                                size_t total, int n, hipStream_t st) {
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(relayout_coef_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, in,
                     cell_batch, cell_pos, total, n);
  return hipGetLastError();
}
template hipError_t relayout_coef_launch<double>(double *, const double *, const uint32_t *, const uint32_t *, size_t,
                                                 int, hipStream_t);
template hipError_t relayout_coef_launch<float>(float *, const float *, const uint32_t *, const uint32_t *, size_t, int,
                                                hipStream_t);

}  // namespace mfgpu
