// 3D cell-loop kernel, two-pass scatter mode, uniform-Jacobian path: a thread owns a 2D PLANE of a cell; ONE LDS
// transpose array (apply_planes4, round 3).  Default at p = 3 in double (two waves per SIMD) and at p = 5, 6 (one wave
// per SIMD, apply_planes4w); on request at p = 2, 4 (MFGPU_KERNEL_PLANES_2W).
//
// apply_planes3 (mfgpu_kernels_p.hip, round 2) runs ONE wave per SIMD: two LDS transpose arrays of 12 cells plus the
// batch array are 36.9 KB per wave and its prefetch state ~330 registers.  Its ablation (profiles/r02_notes.md section 2)
// showed 102.6 us of pure contractions + LDS with every global access removed -- more than the 83 us the 60 % roofline
// target leaves -- because a lone wave serialises its VALU, LDS and memory issue.  This kernel keeps the algebra and
// the per-batch records and halves the resources per wave, so that eight single-wave workgroups share a CU:
//
//   * ONE transpose array T, aliased with the batch array ua (gathered values, later the accumulator): ua is dead
//     from the moment every lane has its xy-plane in registers until the results are added, and that is exactly when T
//     is alive.  17.2 KB per wave at p = 4 (cell stride 165, plane stride 33: both thread layouts conflict-free).
//   * the two operands of the forward transpose (a = S_x S_y u and b = D_x a) pass through T one after the other;
//     what waits stays in the registers of the xy-layout.  The second derivative part of the yz-stage updates the
//     values in place and adds its S_z^T image into T with ds_add, so that stage holds one plane, not two:
//
//   S1 (xy, nodal z = k):  u <- ua;  t = S_y u;  a = S_x t (in place);  b = D_x a            b  -> T
//   S2 (yz, quad  x = k):  t' = S_z^T (c . S_z b)                                             t' -> T (in place)
//   S3 (xy):               w = D_x^T t' (registers);                                          a  -> T
//   S4 (yz):               v = S_z a;  T  = S_z^T D_z^T (c . D_z v);  v <- D_y^T (c . D_y v);  T += S_z^T v
//   S5 (xy):               out = S_y^T S_x^T (w + T);  ua <- 0;  ua += out
//
//     14 contractions per cell as before (reference: 18, tensor_ops.cuh:179-261); 5 LDS stores + 1 LDS add per value.
//   * prefetch state cut to the NEXT batch only: its dof list (requested before S3), source values and index runs
//     (requested inside S5, between the contractions) and coefficient rows (inside S5 as well, when this batch's are
//     dead -- or, with one wave per SIMD, the batch's own rows during its S1); the results are stored at the end of
//     their own iteration.  <= 256 registers at p <= 4.
//
// What it bought (profiles/r03_notes.md): at p = 4 in double NOTHING (equal to apply_planes3: the floor there is FP64
// issue, and two in-order waves running the same phases overlap little); p = 3: 9-12 % per vmult over the pencil kernel
// and apply_planes3; p = 5, 6: a plane kernel at all (apply_planes3's two arrays would be 73 KB per wave): C5 -18 %.
//
// Algebra per cell: fee_gpu.cuh:219-284 (uniform-Jacobian branch), tensor_ops.cuh:179-261;
// gather / scatter: fee_gpu.cuh:323-363; constrained rows: constraint_handler_gpu.cu:247-289;
// hanging nodes: hanging_nodes.cuh:617-778 (as in apply_planes3: private entries behind the dof list).
#include <hip/hip_runtime.h>

#include "mfgpu_cell.h"
#include "mfgpu_kernels.h"
#include "mfgpu_planes.h"

namespace mfgpu {

// Only LDS instructions may cross: pins a global memory operation between two compute steps
#define MFGPU_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x380)

template <int n, typename T, bool ADD, bool HN>
__device__ __forceinline__ void planes4_body(const ApplyArgs<T> &A, const TablesEO<T, n> &tab) {
  constexpr int n2 = n * n;
  constexpr int CW = p_cells_per_wave(n);  // cells per wave
  constexpr int NT = CW * n;               // tasks (active lanes) of a full batch
  constexpr int KGU = p_kgu(n);            // 64-lane slots of a batch's dof list:
  constexpr int JI = p_ji(n);              //   JI slots of interior dofs, then HS = KGU - JI slots of pass-2 dofs
  constexpr int HS = KGU - JI;
  constexpr int SA = q_cell_stride(n);     // cell stride and plane (z) stride of the transpose array
  constexpr int ZS = q_plane_stride(n);
  constexpr int NIW = (n2 + 1) / 2;        // 32-bit words of a task's packed index run
  constexpr int PRIV = HN ? p_priv_max(n) : 0;  // private entries of the hanging-node batches
  constexpr int NUA = KGU * 64 + PRIV;          // entries of the batch array
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double *ua = reinterpret_cast<double *>(smem_raw);  // gathered source values, then the accumulator
  T *Tw = reinterpret_cast<T *>(smem_raw);            // ALIASED: CW cells + one scratch cell for the idle lanes
  double *Wl = reinterpret_cast<double *>(smem_raw) + q_region_doubles<T>(n);  // HN: W[i * n + k], behind both

  const int lane = threadIdx.x;
  if (HN) {
    if (lane < n2) Wl[lane] = (double)A.hn_weights[lane];
  }
  // The idle lanes (NT .. 63) and the tasks of cells a ragged batch does not have run the same instruction stream on
  // harmless data (as in apply_planes3): a scratch cell of T, zero coefficient rows, index runs that point at the batch
  // array's last slot (never a dof).
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

  // Per-batch records: fixed sizes, fixed structure (mfgpu_plan.cpp build_plane_records), no per-batch metadata.
  // Vectors and the halo buffer are addressed base + 32-bit byte offset (n_dofs < 2^29; shifting a dof-list entry
  // left by 3 also drops its flag bit 31).
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
  auto load_coef = [&](uint32_t bb, T (&c)[n2]) {
    const T *p = A.coefp + (size_t)bb * (n2 * NT) + tk;
#pragma unroll
    for (int r = 0; r < n2; ++r) c[r] = nt_load(p + r * NT);
  };
  // gathered values -> LDS (read_dof_values, fee_gpu.cuh:323-331, once per batch dof).  bdofs bit 31 = constrained
  // row: reads as 0 (constraint_handler_gpu.cu:258-259); constrained dofs sit in the pass-2 slots only.  The identity
  // rows themselves (dst = src, :286) are written by pass 2 for EVERY constrained dof: this kernel never stores to one.
  auto stage_src = [&](const uint32_t (&g)[KGU], const T (&sv)[KGU]) {
#pragma unroll
    for (int j = 0; j < KGU; ++j) {
      double v = (double)sv[j];
      if (j >= JI) {
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
  constexpr uint32_t kDummyIx = 8u * (uint32_t)(KGU * 64 - 1) * 0x10001u;  // the idle lanes' runs

  uint32_t b1 = next_of(b);
  uint32_t Gc[KGU], IXc[NIW];
  T Cc[n2], SV[KGU];
  constexpr int CR = PRIV / 64, HROWS = HN ? p_hn_rows(n) : 1;
  uint32_t Hc[HROWS];  // HN: this batch's hanging-node record
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
  load_dofs(b, Gc);
  load_ix(b, IXc);
  if (HN) load_hn(b, Hc);
#pragma unroll
  for (int j = 0; j < KGU; ++j) SV[j] = src_at(Gc[j]);
  // Coefficient rows: requested a batch ahead, inside S5 (two waves per SIMD: the registers are free from S5 on and the
  // sibling wave covers the burst) -- or, with one wave per SIMD (n >= 6), THIS batch's rows during S1, one or two
  // between the contractions: the lone wave is blocked ~150 cycles by every vector-memory instruction it issues right
  // behind another one, and S5 already carries the next batch's gathers and index runs (stamp build: S5 took 25 k of
  // 53 k cycles per batch with all 115 loads in it).  S1 is 5 k cycles long: the rows are there when S2 needs them.
  constexpr bool kCoefInS1 = n >= 6;
  if (!kCoefInS1) load_coef(b, Cc);
#pragma unroll
  for (int w = 0; w < NIW; ++w) IXc[w] = lane_on ? IXc[w] : kDummyIx;

  // LDS bases of this task: xy-plane with z = k (S1, S3, S5) and yz-plane with x = k (S2, S4)
  const int pxy = lc * SA + ZS * k;  // + x + n * y
  const int pyz = lc * SA + k;       // + n * y + ZS * z

  while (true) {
    const bool has_next = b1 != b;
    // ---- S0: the batch's gathered values -> ua (the one wait for the gather, requested during the previous S5)
    stage_src(Gc, SV);
    WaveSync::sync();

    // ---- hanging-node batches: private copies of the constrained nodes, then the interpolation passes x, y, z
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
          pos[t] = on ? (ww[t >> 1] >> (16 * (t & 1))) & 0xffffu : (uint32_t)(NUA - 1);
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

    const T *const cthis = A.coefp + (size_t)b * (n2 * NT) + tk;
    auto hook_coef = [&](int s) {  // step s of the 2 n steps of S1 (kCoefInS1)
      MFGPU_PIN_VMEM();
#pragma unroll
      for (int r = (n2 * s) / (2 * n); r < (n2 * (s + 1)) / (2 * n); ++r) Cc[r] = nt_load(cthis + r * NT);
      MFGPU_PIN_VMEM();
    };
    // ---- S1 (xy): gather the plane, S_y, then per line S_x (kept) and D_x (-> T)
    T u[n2];
#pragma unroll
    for (int i = 0; i < n2; ++i)
      u[i] = (T) * reinterpret_cast<const double *>(reinterpret_cast<const char *>(ua) + ixb(IXc, i));
    WaveSync::sync();  // every lane has its plane: ua is dead, T may be written
#pragma unroll
    for (int x = 0; x < n; ++x) {
      T in[n], out[n];
      get_line<n, 1>(u, x, in);
      eo_apply<n, 0>(tab, in, out);
      set_line<n, 1>(u, x, out);
      if (kCoefInS1) hook_coef(x);
    }
#pragma unroll
    for (int y = 0; y < n; ++y) {
      T in[n], a[n], bb[n];
      get_line<n, 0>(u, y, in);
      eo_apply<n, 0>(tab, in, a);
      eo_apply<n, 2>(tab, a, bb);
      set_line<n, 0>(u, y, a);
#pragma unroll
      for (int x = 0; x < n; ++x) Tw[pxy + x + n * y] = bb[x];
      if (kCoefInS1) hook_coef(n + y);
    }
    WaveSync::sync();

    // ---- S2 (yz): x-derivative part, t' = S_z^T (c .* S_z b), in place
    // p = 5 (one wave per SIMD): the next batch's index runs are requested here, not among the 63 loads of S5 (cell loop
    // 120 -> 116 us at 10^7 dofs); at p = 6 the 25 registers they hold through S4 are 20 spills (147 -> 136 us lost)
    constexpr bool kIxInS2 = n == 6;
    uint32_t IXn[NIW];
    const uint32_t *const ixnext = A.idxp + (size_t)b1 * (NIW * NT) + tk;
#pragma unroll
    for (int y = 0; y < n; ++y) {
      T in[n], g[n], o[n];
#pragma unroll
      for (int z = 0; z < n; ++z) in[z] = Tw[pyz + n * y + ZS * z];
      eo_apply<n, 0>(tab, in, g);
#pragma unroll
      for (int z = 0; z < n; ++z) g[z] *= Cc[y + n * z];
      eo_apply<n, 1>(tab, g, o);
#pragma unroll
      for (int z = 0; z < n; ++z) Tw[pyz + n * y + ZS * z] = o[z];
      if (kIxInS2) {
        MFGPU_PIN_VMEM();
#pragma unroll
        for (int i = (NIW * y) / n; i < (NIW * (y + 1)) / n; ++i) IXn[i] = nt_load(ixnext + i * NT);
        MFGPU_PIN_VMEM();
      }
    }
    WaveSync::sync();

    // the next batch's dof list: the gather addresses must be here when the gather is issued (before S5)
    uint32_t Gn[KGU];
    load_dofs(b1, Gn);

    // ---- S3 (xy): w = D_x^T t' into registers; a -> T (same entries, this lane's own)
    T w[n2];
#pragma unroll
    for (int y = 0; y < n; ++y) {
      T tp[n], wl[n];
#pragma unroll
      for (int x = 0; x < n; ++x) tp[x] = Tw[pxy + x + n * y];
      eo_apply<n, 3>(tab, tp, wl);
      set_line<n, 0>(w, y, wl);
#pragma unroll
      for (int x = 0; x < n; ++x) Tw[pxy + x + n * y] = u[x + n * y];
    }
    WaveSync::sync();

    // ---- S4 (yz): values at the quadrature points, z- and y-derivative parts
    {
      T v[n2];
#pragma unroll
      for (int y = 0; y < n; ++y) {
        T in[n], o[n];
#pragma unroll
        for (int z = 0; z < n; ++z) in[z] = Tw[pyz + n * y + ZS * z];
        eo_apply<n, 0>(tab, in, o);
        set_line<n, 1>(v, y, o);
      }
      // z-derivative part, straight through S_z^T into T
#pragma unroll
      for (int y = 0; y < n; ++y) {
        T in[n], g[n], o[n], o2[n];
        get_line<n, 1>(v, y, in);
        eo_apply<n, 2>(tab, in, g);
#pragma unroll
        for (int z = 0; z < n; ++z) g[z] *= Cc[y + n * z];
        eo_apply<n, 3>(tab, g, o);
        eo_apply<n, 1>(tab, o, o2);
#pragma unroll
        for (int z = 0; z < n; ++z) Tw[pyz + n * y + ZS * z] = o2[z];
      }
      // y-derivative part, in place
#pragma unroll
      for (int z = 0; z < n; ++z) {
        T in[n], g[n], o[n];
        get_line<n, 0>(v, z, in);
        eo_apply<n, 2>(tab, in, g);
#pragma unroll
        for (int y = 0; y < n; ++y) g[y] *= Cc[y + n * z];
        eo_apply<n, 3>(tab, g, o);
        set_line<n, 0>(v, z, o);
      }
      // ... and its S_z^T image added
#pragma unroll
      for (int y = 0; y < n; ++y) {
        T in[n], o[n];
        get_line<n, 1>(v, y, in);
        eo_apply<n, 1>(tab, in, o);
#pragma unroll
        for (int z = 0; z < n; ++z) lds_add(&Tw[pyz + n * y + ZS * z], o[z]);
      }
    }
    WaveSync::sync();

    // ---- everything the NEXT batch needs from HBM: this batch's coefficient rows are dead
    // The requests are spread over the 2 n steps of S5, pinned between its contractions (only LDS instructions may
    // cross the pins): issued in one burst they block the wave for ~7 k cycles (125 cycles per instruction: eight
    // waves share the CU's one address unit), spread out they are issued while the wave computes.
    uint32_t Hn[HROWS];
    if (HN) load_hn(b1, Hn);
    const T *const cnext = A.coefp + (size_t)b1 * (n2 * NT) + tk;
    auto hook = [&](int s) {  // step s of 2 n: gathers first (consumed first), then index runs, then coefficient rows
      constexpr int NIX = kIxInS2 ? 0 : NIW;
      constexpr int NL = KGU + NIX + (kCoefInS1 ? 0 : n2);
      MFGPU_PIN_VMEM();
#pragma unroll
      for (int i = (NL * s) / (2 * n); i < (NL * (s + 1)) / (2 * n); ++i) {
        if (i < KGU) SV[i] = src_at(Gn[i]);
        else if (i < KGU + NIX) IXn[i - KGU] = nt_load(ixnext + (i - KGU) * NT);
        else if (!kCoefInS1) Cc[i - KGU - NIX] = nt_load(cnext + (i - KGU - NIX) * NT);
      }
      MFGPU_PIN_VMEM();
    };

    // ---- S5 (xy): out = S_y^T S_x^T (w + r'), then into the batch accumulator
#pragma unroll
    for (int y = 0; y < n; ++y) {
      T rp[n], s2[n], ol[n];
#pragma unroll
      for (int x = 0; x < n; ++x) rp[x] = Tw[pxy + x + n * y];
#pragma unroll
      for (int x = 0; x < n; ++x) s2[x] = w[x + n * y] + rp[x];
      eo_apply<n, 1>(tab, s2, ol);
      set_line<n, 0>(w, y, ol);
      hook(y);
    }
    WaveSync::sync();  // T is dead: the region becomes the accumulator
#pragma unroll
    for (int j = 0; j < (NUA + 63) / 64; ++j) ua[lane + j * 64] = 0.0;
#pragma unroll
    for (int x = 0; x < n; ++x) {
      T in[n], out[n];
      get_line<n, 1>(w, x, in);
      eo_apply<n, 1>(tab, in, out);
      set_line<n, 1>(w, x, out);
      hook(n + x);
    }
    WaveSync::sync();
#pragma unroll
    for (int i = 0; i < n2; ++i)
      lds_add(reinterpret_cast<double *>(reinterpret_cast<char *>(ua) + ixb(IXc, i)), (double)w[i]);
    WaveSync::sync();

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

    // ---- S6: scatter (distribute_local_to_global, fee_gpu.cuh:346-363), ONE wave-wide store per 64-lane slot and no
    // case distinction at run time: the first JI slots hold interior dofs -- the batch's alone, final, never
    // constrained -- and go to dst (padding lanes store the 0 of their untouched accumulator slot, or old + 0, to a
    // pass-2 dof of the batch, which pass 2 rewrites); the other HS slots are partial sums for pass 2 and go to the
    // batch's HS * 64 halo slots (padding lanes: unused halo slots; constrained dofs: value ignored by pass 2).
    {
      T *const hp = A.halo + (size_t)b * (HS * 64) + lane;
#pragma unroll
      for (int j = 0; j < KGU; ++j) {
        const T r = (T)ua[lane + j * 64];
        if (j < JI) {
          T *const p = dst_at(Gc[j]);
          *p = ADD ? *p + r : r;
        } else {
          hp[(j - JI) * 64] = r;
        }
      }
    }
    WaveSync::sync();
    if (!has_next) break;
    b = b1;
    b1 = next_of(b1);
#pragma unroll
    for (int j = 0; j < KGU; ++j) Gc[j] = Gn[j];
#pragma unroll
    for (int w2 = 0; w2 < NIW; ++w2) IXc[w2] = lane_on ? IXn[w2] : kDummyIx;
    if (HN) {
#pragma unroll
      for (int w2 = 0; w2 < HROWS; ++w2) Hc[w2] = Hn[w2];
    }
  }
}

// p <= 4: two waves per SIMD (<= 256 registers).  p = 5, 6 (n = 6, 7): a plane is 72 / 98 registers and four of them
// are live in S4 -- one wave per SIMD with the whole register file; the single aliased transpose array is what lets
// four such workgroups share a CU's LDS at all (34 / 36 KB each).
template <int n, typename T, bool ADD, bool HN>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
apply_planes4(const ApplyArgs<T> A, const TablesEO<T, n> tab) {
  planes4_body<n, T, ADD, HN>(A, tab);
}
template <int n, typename T, bool ADD>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
apply_planes4w(const ApplyArgs<T> A, const TablesEO<T, n> tab) {
  planes4_body<n, T, ADD, false>(A, tab);
}

template <int n, typename T>
static hipError_t q_run_w(const ApplyArgs<T> &a, const double *S, const double *Dt, uint32_t grid, hipStream_t st,
                          bool configure_only, size_t *lds_out, int *occupancy) {
  const size_t lds = q_lds_bytes<T>(n, false);
  if (lds_out) *lds_out = lds;
  if (configure_only) {
    hipError_t e = hipFuncSetAttribute((const void *)apply_planes4w<n, T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)apply_planes4w<n, T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && occupancy)
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, apply_planes4w<n, T, false>, 64, lds);
    return e;
  }
  const TablesEO<T, n> tab = make_tables_eo<T, n>(S, Dt);
  if (a.add)
    hipLaunchKernelGGL((apply_planes4w<n, T, true>), dim3(grid), dim3(64), lds, st, a, tab);
  else
    hipLaunchKernelGGL((apply_planes4w<n, T, false>), dim3(grid), dim3(64), lds, st, a, tab);
  return hipGetLastError();
}

template <int n, typename T>
static hipError_t q_run(const ApplyArgs<T> &a, const double *S, const double *Dt, bool hn, uint32_t grid, hipStream_t st,
                        bool configure_only, size_t *lds_out, int *occupancy) {
  const size_t lds = q_lds_bytes<T>(n, hn);
  if (lds_out) *lds_out = lds;
  if (configure_only) {
    const void *f0 = hn ? (const void *)apply_planes4<n, T, false, true> : (const void *)apply_planes4<n, T, false, false>;
    const void *f1 = hn ? (const void *)apply_planes4<n, T, true, true> : (const void *)apply_planes4<n, T, true, false>;
    hipError_t e = hipFuncSetAttribute(f0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(f1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && occupancy)
      e = hn ? hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, apply_planes4<n, T, false, true>, 64, lds)
             : hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, apply_planes4<n, T, false, false>, 64, lds);
    return e;
  }
  const TablesEO<T, n> tab = make_tables_eo<T, n>(S, Dt);
  if (hn) {
    if (a.add)
      hipLaunchKernelGGL((apply_planes4<n, T, true, true>), dim3(grid), dim3(64), lds, st, a, tab);
    else
      hipLaunchKernelGGL((apply_planes4<n, T, false, true>), dim3(grid), dim3(64), lds, st, a, tab);
  } else {
    if (a.add)
      hipLaunchKernelGGL((apply_planes4<n, T, true, false>), dim3(grid), dim3(64), lds, st, a, tab);
    else
      hipLaunchKernelGGL((apply_planes4<n, T, false, false>), dim3(grid), dim3(64), lds, st, a, tab);
  }
  return hipGetLastError();
}

template <typename T>
hipError_t q_launch(int n, const ApplyArgs<T> &a, const double *S, const double *Dt, bool hn, uint32_t grid,
                    hipStream_t st, bool configure_only, size_t *lds_out, int *occupancy) {
  switch (n) {
    case 3: return q_run<3, T>(a, S, Dt, hn, grid, st, configure_only, lds_out, occupancy);
    case 4: return q_run<4, T>(a, S, Dt, hn, grid, st, configure_only, lds_out, occupancy);
    case 5: return q_run<5, T>(a, S, Dt, hn, grid, st, configure_only, lds_out, occupancy);
    case 6: return hn ? hipErrorInvalidValue : q_run_w<6, T>(a, S, Dt, grid, st, configure_only, lds_out, occupancy);
    case 7: return hn ? hipErrorInvalidValue : q_run_w<7, T>(a, S, Dt, grid, st, configure_only, lds_out, occupancy);
    default: return hipErrorInvalidValue;
  }
}

template hipError_t q_launch<double>(int, const ApplyArgs<double> &, const double *, const double *, bool, uint32_t,
                                     hipStream_t, bool, size_t *, int *);
template hipError_t q_launch<float>(int, const ApplyArgs<float> &, const double *, const double *, bool, uint32_t,
                                    hipStream_t, bool, size_t *, int *);

}  // namespace mfgpu
