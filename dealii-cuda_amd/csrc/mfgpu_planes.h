// Device helpers shared by the plane-per-thread cell-loop kernels (mfgpu_kernels_p.hip, mfgpu_kernels_q.hip): register
// planes, and the even-odd form of the 1D contractions (tensor_ops.cuh:25-117 is the reference's plain form).
#ifndef MFGPU_PLANES_H
#define MFGPU_PLANES_H

#include <hip/hip_runtime.h>

namespace mfgpu {

namespace {

template <typename U>
__device__ __forceinline__ U nt_load(const U *p) { return __builtin_nontemporal_load(p); }

// strided view of a register plane: line `o` along direction DIR (0: fast index, 1: slow index)
template <int n, int DIR, typename T>
__device__ __forceinline__ void get_line(const T (&p)[n * n], int o, T (&l)[n]) {
#pragma unroll
  for (int i = 0; i < n; ++i) l[i] = p[DIR == 0 ? i + n * o : o + n * i];
}
template <int n, int DIR, typename T>
__device__ __forceinline__ void set_line(T (&p)[n * n], int o, const T (&l)[n]) {
#pragma unroll
  for (int i = 0; i < n; ++i) p[DIR == 0 ? i + n * o : o + n * i] = l[i];
}


// ---- even-odd form of the 1D contractions ----------------------------------------------------------------
// The 1D tables are centro-symmetric (S[i][q] = S[p-i][p-q]) or centro-antisymmetric (Dt[q][t] = -Dt[p-q][p-t]), so a
// length-n mat-vec splits into an even and an odd half-size one on e[k] = v[k] + v[p-k], o[k] = v[k] - v[p-k] (and
// the middle entry for odd n): 21 / 20 instead of 25 operations at n = 5, and 25 instead of 30 table entries in
// scalar registers.  With h = n / 2, m = (n + 1) / 2:
//   Se[m][m]: k < h, q < h: (S[k][q] + S[p-k][q]) / 2;  row h / column h (odd n): S[h][q], S[k][h]
//   So[h][h]: (S[k][q] - S[p-k][q]) / 2
//   De[m][h]: q < h: (Dt[q][t] - Dt[q][p-t]) / 2;  row h (odd n): Dt[h][t]
//   Do[h][m]: t < h: (Dt[q][t] + Dt[q][p-t]) / 2;  column h (odd n): Dt[q][h]
template <typename T, int n>
struct TablesEO {
  static constexpr int h = n / 2, m = (n + 1) / 2;
  T Se[m * m], So[h * h], De[m * h], Do[h * m];
};

// KIND 0: out[q] = sum_k S[k][q] in[k]   (interpolate, "mvt(S)")      1: out[q] = sum_k S[q][k] in[k]   ("mv(S)")
//      2: out[q] = sum_t Dt[q][t] in[t]  (derivative, "mv(Dt)")       3: out[t] = sum_q Dt[q][t] in[q]  ("mvt(Dt)")
template <int n, int KIND, typename T>
__device__ __forceinline__ void eo_apply(const TablesEO<T, n> &tb, const T (&in)[n], T (&out)[n]) {
  constexpr int h = n / 2, m = (n + 1) / 2, p = n - 1;
  constexpr bool odd = (n & 1) != 0;
  T e[m], o[h > 0 ? h : 1];
#pragma unroll
  for (int k = 0; k < h; ++k) {
    e[k] = in[k] + in[p - k];
    o[k] = in[k] - in[p - k];
  }
  if (odd) e[h] = in[h];
  T E[m], O[h > 0 ? h : 1];
  if (KIND == 0 || KIND == 1) {
    // even part: m x m on (e, mid); odd part: h x h on o
#pragma unroll
    for (int q = 0; q < m; ++q) {
      T t = (KIND == 0 ? tb.Se[0 * m + q] : tb.Se[q * m + 0]) * e[0];
#pragma unroll
      for (int k = 1; k < m; ++k) t = fma(KIND == 0 ? tb.Se[k * m + q] : tb.Se[q * m + k], e[k], t);
      E[q] = t;
    }
#pragma unroll
    for (int q = 0; q < h; ++q) {
      T t = (KIND == 0 ? tb.So[0 * h + q] : tb.So[q * h + 0]) * o[0];
#pragma unroll
      for (int k = 1; k < h; ++k) t = fma(KIND == 0 ? tb.So[k * h + q] : tb.So[q * h + k], o[k], t);
      O[q] = t;
    }
  } else if (KIND == 2) {
    // even outputs (and the middle one) from o through De[m][h]; odd outputs from (e, mid) through Do[h][m]
#pragma unroll
    for (int q = 0; q < m; ++q) {
      T t = tb.De[q * h + 0] * o[0];
#pragma unroll
      for (int k = 1; k < h; ++k) t = fma(tb.De[q * h + k], o[k], t);
      E[q] = t;
    }
#pragma unroll
    for (int q = 0; q < h; ++q) {
      T t = tb.Do[q * m + 0] * e[0];
#pragma unroll
      for (int k = 1; k < m; ++k) t = fma(tb.Do[q * m + k], e[k], t);
      O[q] = t;
    }
  } else {
    // transposed derivative: even outputs (and the middle one) from o through Do^T; odd outputs from (e, mid) through De^T
#pragma unroll
    for (int q = 0; q < m; ++q) {
      T t = tb.Do[0 * m + q] * o[0];
#pragma unroll
      for (int k = 1; k < h; ++k) t = fma(tb.Do[k * m + q], o[k], t);
      E[q] = t;
    }
#pragma unroll
    for (int q = 0; q < h; ++q) {
      T t = tb.De[0 * h + q] * e[0];
#pragma unroll
      for (int k = 1; k < m; ++k) t = fma(tb.De[k * h + q], e[k], t);
      O[q] = t;
    }
  }
#pragma unroll
  for (int q = 0; q < h; ++q) {
    out[q] = E[q] + O[q];
    out[p - q] = E[q] - O[q];
  }
  if (odd) out[h] = E[h];
}


// host: the even-odd tables from S[i*n+q] = phi_i(x_q), Dt[q*n+t] = l_t'(x_q) (full n x n, symmetrised by mfgpu_create)
template <typename T, int n>
inline TablesEO<T, n> make_tables_eo(const double *S, const double *Dt) {
  TablesEO<T, n> tab;
  constexpr int h = n / 2, m = (n + 1) / 2, p = n - 1;
  for (int k = 0; k < m; ++k)
    for (int q = 0; q < m; ++q)
      tab.Se[k * m + q] = (T)((k < h && q < h) ? 0.5 * (S[k * n + q] + S[(p - k) * n + q]) : S[k * n + q]);
  for (int k = 0; k < h; ++k)
    for (int q = 0; q < h; ++q) tab.So[k * h + q] = (T)(0.5 * (S[k * n + q] - S[(p - k) * n + q]));
  for (int q = 0; q < m; ++q)
    for (int t = 0; t < h; ++t)
      tab.De[q * h + t] = (T)(q < h ? 0.5 * (Dt[q * n + t] - Dt[q * n + (p - t)]) : Dt[q * n + t]);
  for (int q = 0; q < h; ++q)
    for (int t = 0; t < m; ++t)
      tab.Do[q * m + t] = (T)(t < h ? 0.5 * (Dt[q * n + t] + Dt[q * n + (p - t)]) : Dt[q * n + t]);
  return tab;
}

}  // namespace

}  // namespace mfgpu
#endif
