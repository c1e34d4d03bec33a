/* C restatement of the reference operator apply for the CPU (TEST INFRASTRUCTURE / cpu_baseline).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file; the
 * product never links or calls it.  It follows the same algorithm as oracle/mf_oracle.py:
 *   gather (fee_gpu.cuh:323-331) -> grad_at_quad_pts (tensor_ops.cuh:179-217) -> quadrature-point
 *   operation, uniform-Jacobian form (laplace_operator_gpu.h:257-260, fee_gpu.cuh:234,274) ->
 *   quad_int_grad<false> (tensor_ops.cuh:219-261) -> scatter (fee_gpu.cuh:346-363), bracketed by the
 *   constrained-row handling of vmult_add (laplace_operator_gpu.h:286-303), dst = 0 first (:221).
 * It stands in for bmop-cpu.cc / laplace_operator_cpu.cc:122-211, whose arithmetic lives in an
 * external deal.II that is not available: threads over cells of one colour (the reference uses
 * deal.II's TBB partition_color scheme, laplace_operator_cpu.cc:51-52) and, in 3D, SIMD over VL = 8 cells
 * (the reference's VectorizedArray<number> over cells, laplace_operator_cpu.cc:122-143): structure-of-arrays
 * cell data [dof][8 cells], compile-time n, the lane loops vectorised by the compiler (AVX-512 on the GPU
 * box's host); one parallel region per vmult.  The arithmetic per cell is the scalar path's, in the same order.
 * PARITY UNPINNED beyond the reference's two test procedures (see mf_oracle.py header); this file is
 * checked against mf_oracle.py in tests/test_cpu_ref.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXN 7
#define MAXND (MAXN * MAXN * MAXN)

/* out[..q..] = sum_k M[q*n+k] in[..k..] along direction d (0 = x fastest) */
static void contract(int dim, int n, int d, const double *M, const double *in, double *out) {
  int nd = 1, s = 1;
  for (int i = 0; i < dim; ++i) nd *= n;
  for (int i = 0; i < d; ++i) s *= n;
  for (int i = 0; i < nd; ++i) {
    const int q = (i / s) % n;
    const int base = i - q * s;
    double t = 0.0;
    for (int k = 0; k < n; ++k) t += M[q * n + k] * in[base + k * s];
    out[i] = t;
  }
}

static void cell_apply(int dim, int n, const double *Se, const double *Ge, const double *S, const double *G,
                       const double *coef, const double *jxw, double j0, const double *u, double *out) {
  int nd = 1;
  for (int i = 0; i < dim; ++i) nd *= n;
  double g[3][MAXND], t1[MAXND], t2[MAXND];
  for (int d = 0; d < dim; ++d) { /* evaluate: reduce along x, then y, then z */
    const double *cur = u;
    double *bufs[2] = {t1, t2};
    for (int r = 0; r < dim; ++r) {
      double *dstb = (r == dim - 1) ? g[d] : bufs[r & 1];
      contract(dim, n, r, r == d ? Ge : Se, cur, dstb);
      cur = dstb;
    }
  }
  for (int d = 0; d < dim; ++d)
    for (int q = 0; q < nd; ++q) g[d][q] = ((coef[q] * (j0 * g[d][q])) * j0) * jxw[q];
  for (int i = 0; i < nd; ++i) out[i] = 0.0;
  for (int d = 0; d < dim; ++d) { /* integrate */
    const double *cur = g[d];
    double *bufs[2] = {t1, t2};
    for (int r = 0; r < dim; ++r) {
      double *dstb = bufs[r & 1];
      contract(dim, n, r, r == d ? G : S, cur, dstb);
      cur = dstb;
    }
    for (int i = 0; i < nd; ++i) out[i] += cur[i];
  }
}

/* ---- 3D, VL cells at a time (SIMD over cells) ------------------------------------------------------------ */
#define VL 8
#define INL static inline __attribute__((always_inline))

/* out[i][l] = sum_k M[q*n+k] in[base + k*s][l] along direction d; n is a compile-time constant at every call */
INL void contract_v(const int n, const int d, const double *restrict M, const double *restrict in,
                    double *restrict out) {
  const int nd = n * n * n, s = d == 0 ? 1 : d == 1 ? n : n * n;
  for (int i = 0; i < nd; ++i) {
    const int q = (i / s) % n;
    const int base = i - q * s;
    double t[VL];
#pragma omp simd
    for (int l = 0; l < VL; ++l) t[l] = 0.0;
    for (int k = 0; k < n; ++k) {
      const double m = M[q * n + k];
      const double *restrict x = in + (size_t)(base + k * s) * VL;
#pragma omp simd
      for (int l = 0; l < VL; ++l) t[l] += m * x[l];
    }
#pragma omp simd
    for (int l = 0; l < VL; ++l) out[(size_t)i * VL + l] = t[l];
  }
}

/* cells k0 .. k0+cnt-1 (cnt <= VL) of the colour-sorted cell list; lanes >= cnt repeat the last cell and are not scattered */
INL void cells_apply_v(const int n, const double *Se, const double *Ge, const double *S, const double *G,
                       const uint32_t *l2g, const double *coef, const double *jxw, const double *j0,
                       const uint32_t *cell_order, int64_t k0, int cnt, double *dst, const double *src) {
  const int nd = n * n * n;
  double u[MAXND * VL], g[3][MAXND * VL], t1[MAXND * VL], t2[MAXND * VL], c[MAXND * VL];
  uint32_t cell[VL];
  for (int l = 0; l < VL; ++l) cell[l] = cell_order[k0 + (l < cnt ? l : cnt - 1)];
  for (int l = 0; l < VL; ++l) {
    const uint32_t *idx = l2g + (size_t)cell[l] * nd;
    const double *cf = coef + (size_t)cell[l] * nd, *jw = jxw + (size_t)cell[l] * nd;
    const double jj = j0[cell[l]];
    for (int i = 0; i < nd; ++i) {
      u[(size_t)i * VL + l] = src[idx[i]];
      c[(size_t)i * VL + l] = cf[i];
      t1[(size_t)i * VL + l] = jw[i]; /* staged: JxW */
    }
    for (int i = 0; i < nd; ++i) t2[(size_t)i * VL + l] = jj; /* staged: J0 */
  }
  /* evaluate: reduce along x, then y, then z; quadrature-point operation ((coef * (J0 * g)) * J0) * JxW */
  double jwv[MAXND * VL], j0v[MAXND * VL];
  memcpy(jwv, t1, sizeof(double) * nd * VL);
  memcpy(j0v, t2, sizeof(double) * nd * VL);
  for (int d = 0; d < 3; ++d) {
    contract_v(n, 0, d == 0 ? Ge : Se, u, t1);
    contract_v(n, 1, d == 1 ? Ge : Se, t1, t2);
    contract_v(n, 2, d == 2 ? Ge : Se, t2, g[d]);
    for (int i = 0; i < nd * VL; ++i) g[d][i] = ((c[i] * (j0v[i] * g[d][i])) * j0v[i]) * jwv[i];
  }
  /* integrate */
  double out[MAXND * VL];
  for (int i = 0; i < nd * VL; ++i) out[i] = 0.0;
  for (int d = 0; d < 3; ++d) {
    contract_v(n, 0, d == 0 ? G : S, g[d], t1);
    contract_v(n, 1, d == 1 ? G : S, t1, t2);
    contract_v(n, 2, d == 2 ? G : S, t2, t1);
    for (int i = 0; i < nd * VL; ++i) out[i] += t1[i];
  }
  for (int l = 0; l < cnt; ++l) {
    const uint32_t *idx = l2g + (size_t)cell[l] * nd;
    for (int i = 0; i < nd; ++i) dst[idx[i]] += out[(size_t)i * VL + l];
  }
}

/* The library is built in one container and run on another host: no -march=native.  The hot loops are cloned for
 * AVX-512, AVX2 and the baseline ISA and dispatched at load time (GCC function multiversioning). */
#define DEF_COLOUR_LOOP(N)                                                                                    \
  __attribute__((target_clones("avx512f", "avx2", "default")))                                 \
  static void colour_loop_##N(const double *Se, const double *Ge, const double *S, const double *G,             \
                              const uint32_t *l2g, const double *coef, const double *jxw, const double *j0,   \
                              const uint32_t *cell_order, int64_t kb, int64_t ke, double *dst, const double *src) { \
    const int64_t nblk = (ke - kb + VL - 1) / VL;                                                               \
    _Pragma("omp for schedule(static)") for (int64_t blk = 0; blk < nblk; ++blk) {                              \
      const int64_t k0 = kb + blk * VL;                                                                         \
      const int cnt = (int)(ke - k0 < VL ? ke - k0 : VL);                                                       \
      cells_apply_v(N, Se, Ge, S, G, l2g, coef, jxw, j0, cell_order, k0, cnt, dst, src);                        \
    }                                                                                                           \
  }
DEF_COLOUR_LOOP(2)
DEF_COLOUR_LOOP(3)
DEF_COLOUR_LOOP(4)
DEF_COLOUR_LOOP(5)
DEF_COLOUR_LOOP(6)
DEF_COLOUR_LOOP(7)

/* dst = A src.  Cells are visited colour by colour (color_off[ncolors+1] into cell_order); cells of
 * one colour share no dof.  src is modified and restored like the reference does. Returns threads. */
int cpu_ref_vmult(int dim, int n, uint32_t n_dofs, const uint32_t *l2g, const double *coef,
                  const double *jxw, const double *j0, const double *sv, const double *sg,
                  const uint32_t *constrained, uint32_t ncon, const uint32_t *color_off, int ncolors,
                  const uint32_t *cell_order, double *dst, double *src) {
  if (n > MAXN) return -1;
  int nd = 1;
  for (int i = 0; i < dim; ++i) nd *= n;
  double Se[MAXN * MAXN], Ge[MAXN * MAXN];
  for (int i = 0; i < n; ++i)
    for (int q = 0; q < n; ++q) {
      Se[q * n + i] = sv[i * n + q]; /* evaluate: out_q = sum_k T[k*n+q] in_k */
      Ge[q * n + i] = sg[i * n + q];
    }
  double *tmp = (double *)malloc(sizeof(double) * (ncon ? ncon : 1));
  int nthreads = 1;
#pragma omp parallel
  {
#ifdef _OPENMP
#pragma omp single
    nthreads = omp_get_num_threads();
#endif
#pragma omp for schedule(static)
    for (int64_t i = 0; i < (int64_t)n_dofs; ++i) dst[i] = 0.0; /* vmult: dst = 0 */
#pragma omp for schedule(static)
    for (int64_t i = 0; i < (int64_t)ncon; ++i) { /* save_constrained_values */
      tmp[i] = src[constrained[i]];
      src[constrained[i]] = 0.0;
    }
    for (int c = 0; c < ncolors; ++c) {
      if (dim == 3) { /* SIMD over 8 cells (cells of one colour share no dof, so the lanes' scatters are independent) */
        const int64_t kb = color_off[c], ke = color_off[c + 1];
        switch (n) {
          case 2: colour_loop_2(Se, Ge, sv, sg, l2g, coef, jxw, j0, cell_order, kb, ke, dst, src); break;
          case 3: colour_loop_3(Se, Ge, sv, sg, l2g, coef, jxw, j0, cell_order, kb, ke, dst, src); break;
          case 4: colour_loop_4(Se, Ge, sv, sg, l2g, coef, jxw, j0, cell_order, kb, ke, dst, src); break;
          case 5: colour_loop_5(Se, Ge, sv, sg, l2g, coef, jxw, j0, cell_order, kb, ke, dst, src); break;
          case 6: colour_loop_6(Se, Ge, sv, sg, l2g, coef, jxw, j0, cell_order, kb, ke, dst, src); break;
          default: colour_loop_7(Se, Ge, sv, sg, l2g, coef, jxw, j0, cell_order, kb, ke, dst, src); break;
        }
        continue;
      }
#pragma omp for schedule(static)
      for (int64_t k = color_off[c]; k < (int64_t)color_off[c + 1]; ++k) {
        const uint32_t cell = cell_order[k];
        const uint32_t *idx = l2g + (size_t)cell * nd;
        double u[MAXND], out[MAXND];
        for (int i = 0; i < nd; ++i) u[i] = src[idx[i]];
        cell_apply(dim, n, Se, Ge, sv, sg, coef + (size_t)cell * nd, jxw + (size_t)cell * nd, j0[cell], u, out);
        for (int i = 0; i < nd; ++i) dst[idx[i]] += out[i];
      }
    }
#pragma omp for schedule(static)
    for (int64_t i = 0; i < (int64_t)ncon; ++i) { /* load_and_add_constrained_values (tmp_dst = 0) */
      dst[constrained[i]] = 0.0 + tmp[i];
      src[constrained[i]] = tmp[i];
    }
  }
  free(tmp);
  return nthreads;
}

/* number of OpenMP threads of the following calls (0: the runtime's default) */
void cpu_ref_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* bmop-cpu.cc:137-155: dst = init; n_iter x { swap(dst,src); vmult(dst,src) }.  a, b: two vectors;
 * returns 0 if the result is in a, 1 if in b. */
int cpu_ref_bmop(int dim, int n, uint32_t n_dofs, const uint32_t *l2g, const double *coef, const double *jxw,
                 const double *j0, const double *sv, const double *sg, const uint32_t *constrained,
                 uint32_t ncon, const uint32_t *color_off, int ncolors, const uint32_t *cell_order,
                 double *a, double *b, double init, int n_iter) {
  double *dst = a, *src = b;
  for (uint32_t i = 0; i < n_dofs; ++i) dst[i] = init;
  for (int it = 0; it < n_iter; ++it) {
    double *t = dst;
    dst = src;
    src = t;
    cpu_ref_vmult(dim, n, n_dofs, l2g, coef, jxw, j0, sv, sg, constrained, ncon, color_off, ncolors,
                  cell_order, dst, src);
  }
  return dst == a ? 0 : 1;
}
