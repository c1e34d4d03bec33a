// Kernel argument structs + launcher declarations (implemented in mfgpu_kernels.hip).
#ifndef MFGPU_KERNELS_H
#define MFGPU_KERNELS_H

#include <hip/hip_runtime.h>

#include "mfgpu_internal.h"

namespace mfgpu {

template <typename T>
struct ApplyArgs {
  const uint32_t *batch_cell_off;
  const uint32_t *batch_dof_off;
  const uint32_t *bdofs;
  const uint8_t *bflags;
  const uint16_t *lmap;
  const uint16_t *lmapx;  // apply_batches_x: x-pencil index runs padded to 32-bit words, or nullptr
  const uint16_t *perm;   // apply_batches_x: [2][256] lane -> pencil id of the y- and the z-stage (nullptr: hanging nodes)
  // apply_planes3: fixed-size per-batch records (nullptr otherwise)
  const uint32_t *bdofsp;  // [p_kgu(n) * 64] dof list: p_ji(n) slots of interior dofs, p_hs(n) slots of pass-2 dofs
  const uint32_t *idxp;    // [(n*n+1)/2 words][NT tasks] packed 16-bit byte offsets into the batch array
  const T *coefp;          // [n*n rows][NT tasks] folded coefficient
  // apply_planes3<HN> (batches hn_batch0 ..): fixed-size records of p_hn_rows(n) x 64 words (mfgpu_internal.h)
  const uint32_t *hnrec;
  const uint32_t *hn_slot;  // per plane batch: index of its record in hnrec, or 0xffffffff (cells without a mask)
  const T *coef;          // folded a*J0^2*JxW (apply_batches_g: the 6 entries of a*JxW*J*J^T), plan cell order
  const uint32_t *cmask;  // plan cell order, or nullptr
  const T *hn_weights;    // [n*n] W[i*n+j] (device), or nullptr
  const T *tabS, *tabDt;  // apply_batches_g2: full 1D tables S[i*n+q], Dt[q*n+t] on the device (nullptr otherwise)
  const uint32_t *batch_nint;  // two-pass mode: interior dofs per batch
  const uint32_t *halo_off;    // two-pass mode: first halo slot per batch
  T *halo;                     // two-pass mode: partial sums of shared dofs
  T *dst;
  const T *src;
  uint32_t batch0;     // first batch of this launch (colour)
  uint32_t batch_end;  // one past the last batch of this launch
  uint32_t hole0 = 0xffffffffu, hole_len = 0;  // plane kernels: batches [hole0, hole0 + hole_len) are skipped
  uint32_t nb_max;  // LDS layout: max dofs per batch
  int add;          // vmult_add semantics
  unsigned long long *stamps;  // diagnostic build only (MFGPU_STAMPS), else nullptr
  int dbg;                     // diagnostic build only: ablation bits (1 cells, 2 gather, 4 scatter, 8 prefetch)
};

// 1D tables, passed by value as kernel arguments (=> scalar registers).
template <typename T, int n>
struct Tables {
  T S[((n + 1) / 2) * n];   // S[i*n+q]  = phi_i(x_q), rows i < (n+1)/2 (rest by symmetry)
  T Dt[((n + 1) / 2) * n];  // Dt[q*n+t] = l_t'(x_q), rows q < (n+1)/2 (rest by antisymmetry)
};

template <typename T>
size_t apply_lds_bytes(int dim, int n, uint32_t nb_max);
template <typename T>
hipError_t apply_configure(int dim, int n, size_t lds);
template <typename T>
hipError_t apply_launch(int dim, int n, const ApplyArgs<T> &a, const double *S, const double *Dt,
                        bool hn, bool twopass, uint32_t grid, hipStream_t st);
template <typename T>
hipError_t apply_occupancy(int dim, int n, bool hn, bool twopass, size_t lds, int *blocks);
// pass 2, class-sorted structure-of-arrays form (mfgpu_pass2.hip)
void build_pass2_classes(const std::vector<uint32_t> &sdofs, const std::vector<uint32_t> &s_off,
                         const std::vector<uint32_t> &s_idx, std::vector<uint32_t> &arr, std::vector<uint32_t> &tiles);
template <typename T>
hipError_t reduce_classes_launch(T *dst, const T *src, const T *halo, const uint32_t *arr, const uint32_t *tiles,
                                 uint32_t n_tiles, int add, hipStream_t st);
// 3D cell loop for three workgroups per CU (mfgpu_kernels_x.hip; two-pass mode)
template <typename T>
hipError_t x_launch(int n, const ApplyArgs<T> &a, const double *S, const double *Dt, bool hn, uint32_t grid,
                    hipStream_t st, bool configure_only, size_t *lds_out, int *occupancy);
// plane-per-thread cell loop (mfgpu_kernels_p.hip; 3D, two-pass mode, uniform-Jacobian path) and its setup relayout
template <typename T>
hipError_t p_launch(int n, const ApplyArgs<T> &a, const double *S, const double *Dt, bool hn, uint32_t grid,
                    hipStream_t st, bool configure_only, size_t *lds_out, int *occupancy);
// ... two waves per SIMD (mfgpu_kernels_q.hip): same records, same arguments
template <typename T>
hipError_t q_launch(int n, const ApplyArgs<T> &a, const double *S, const double *Dt, bool hn, uint32_t grid,
                    hipStream_t st, bool configure_only, size_t *lds_out, int *occupancy);
template <typename T>
hipError_t relayout_coef_launch(T *out, const T *in, const uint32_t *cell_batch, const uint32_t *cell_pos,
                                size_t total, int n, hipStream_t st);
// general-Jacobian cell loop (mfgpu_kernels_g.hip; 3D, two-pass mode, conforming meshes) and its setup fold
template <typename T>
hipError_t g_launch(int n, const ApplyArgs<T> &a, const double *S, const double *Dt, bool hn, uint32_t grid,
                    hipStream_t st, bool configure_only, size_t *lds_out, int *occupancy);
template <typename T>
hipError_t fold_general_launch(T *M, const T *coef, const T *jxw, const T *jinv, const uint32_t *order,
                               uint32_t n_cells, uint32_t nd, hipStream_t st);
// general-Jacobian cell loop in 2D (mfgpu_kernels_g2.hip; two-pass mode) and its setup fold (3 metric entries per point)
template <typename T>
hipError_t g2_launch(int n, const ApplyArgs<T> &a, bool hn, uint32_t grid, hipStream_t st, bool configure_only,
                     size_t *lds_out, int *occupancy);
template <typename T>
hipError_t fold_general2_launch(T *M, const T *coef, const T *jxw, const T *jinv, const uint32_t *order,
                                uint32_t n_cells, uint32_t nd, hipStream_t st);
template <typename T>
hipError_t orphan_launch(T *dst, const T *src, const uint32_t *orph, uint32_t n, int add, hipStream_t st);
template <typename T>
hipError_t coefficient_launch(T *coef, const T *qpts, size_t nq, int dim, hipStream_t st);
template <typename T>
hipError_t fold_launch(T *c, const T *coef, const T *jxw, const T *j0, const uint32_t *order,
                       uint32_t n_cells, uint32_t nd, hipStream_t st);
template <typename T>
hipError_t fill_launch(T *v, size_t n, T a, hipStream_t st);

// ---- SURVEY.md 8(f) N1 / N2 (mfgpu_aux.hip)
template <typename T>
hipError_t diag_launch(int dim, int n, T *diag, uint32_t n_batches, const uint32_t *batch_cell_off,
                       const uint32_t *batch_dof_off, const uint32_t *bdofs, const uint16_t *lmap, const T *coef,
                       const uint32_t *cmask, const T *hn_weights, const T *tab2, hipStream_t st);
template <typename T>
hipError_t diag_general_launch(int n, T *diag, uint32_t n_batches, const uint32_t *batch_cell_off,
                               const uint32_t *batch_dof_off, const uint32_t *bdofs, const uint16_t *lmap,
                               const T *metric, const uint32_t *cmask, const T *hn_weights, const T *tab,
                               hipStream_t st);
template <typename T>
hipError_t diag_general2_launch(int n, T *diag, uint32_t n_batches, const uint32_t *batch_cell_off,
                                const uint32_t *batch_dof_off, const uint32_t *bdofs, const uint16_t *lmap,
                                const T *metric, const uint32_t *cmask, const T *hn_weights, const T *tab,
                                hipStream_t st);
template <typename T>
hipError_t set_values_launch(T *v, const uint32_t *idx, uint32_t n, T value, hipStream_t st);
// op: 0 sadd (v = s v + a w), 1 equ (v = a w), 2 scale (v *= w), 3 divide (v /= w), 4 invert, 5 mul (v *= a)
template <typename T>
hipError_t vec_map_launch(int op, T *v, const T *w, T s, T a, size_t n, hipStream_t st);
// op: 0 dot (v . w), 1 add_and_dot (v += a x; v . w), 2 count of non-zeros; blocking, result on the host
template <typename T>
hipError_t vec_reduce_launch(int op, T *v, const T *x, const T *w, T a, size_t n, hipStream_t st, double *out);

}  // namespace mfgpu
#endif
