// Pass 2 of the two-pass scatter: dst = sum of the partial sums the batches left in the halo buffer, for every dof
// on the pass-2 route (shared between batches, or demoted to this route by the plan), in ascending batch order
// (deterministic; the reference adds with atomics in arbitrary order, fee_gpu.cuh:359-362); constrained rows are
// identity rows, dst = src (laplace_operator_gpu.h:300-302, constraint_handler_gpu.cu:276-289).
//
// The dofs are sorted by their number of partial sums k into CLASSES (k = 2: interior of a face between two batches,
// 4: an edge, 8: a vertex, 1: demoted; other values on irregular meshes).  A class is stored structure-of-arrays --
// dof ids, then k arrays of halo slots -- so a thread's k + 1 index loads are independent and coalesced and the whole
// reduction is TWO dependent memory round trips (indices, partial sums) instead of the four of a descriptor ->
// group starts -> partials chain; every thread carries two dofs.  A 512-entry tile per workgroup; a class is padded
// to whole tiles with 0xffffffff.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "mfgpu_kernels.h"

namespace mfgpu {

namespace {

template <typename T, int K>
__device__ __forceinline__ void reduce_two(T *__restrict__ dst, const T *__restrict__ src, const T *__restrict__ halo,
                                           const uint32_t *__restrict__ p, uint32_t cnt, uint32_t i0, uint32_t k,
                                           int add) {
  // K > 0: compile-time number of partial sums; K == 0: run-time k (rare classes)
  const uint32_t i1 = i0 + 256u;
  const uint32_t d0 = p[i0], d1 = p[i1];
  constexpr int KU = K > 0 ? K : 1;
  uint32_t s0[KU], s1[KU];
#pragma unroll
  for (int t = 0; t < KU; ++t) {
    s0[t] = p[(size_t)(1 + t) * cnt + i0];
    s1[t] = p[(size_t)(1 + t) * cnt + i1];
  }
  const bool on0 = d0 != 0xffffffffu, on1 = d1 != 0xffffffffu;
  const uint32_t g0 = d0 & 0x7fffffffu, g1 = d1 & 0x7fffffffu;
  T v0 = T(0), v1 = T(0), o0 = T(0), o1 = T(0);
  if (on0) {
    if (d0 >> 31) {
      v0 = src[g0];
    } else {
      T q[KU];
#pragma unroll
      for (int t = 0; t < KU; ++t) q[t] = halo[s0[t]];
      v0 = q[0];
#pragma unroll
      for (int t = 1; t < KU; ++t) v0 += q[t];
      if (K == 0)
        for (uint32_t t = 1; t < k; ++t) v0 += halo[p[(size_t)(1 + t) * cnt + i0]];
    }
    if (add) o0 = dst[g0];
  }
  if (on1) {
    if (d1 >> 31) {
      v1 = src[g1];
    } else {
      T q[KU];
#pragma unroll
      for (int t = 0; t < KU; ++t) q[t] = halo[s1[t]];
      v1 = q[0];
#pragma unroll
      for (int t = 1; t < KU; ++t) v1 += q[t];
      if (K == 0)
        for (uint32_t t = 1; t < k; ++t) v1 += halo[p[(size_t)(1 + t) * cnt + i1]];
    }
    if (add) o1 = dst[g1];
  }
  if (on0) dst[g0] = add ? o0 + v0 : v0;
  if (on1) dst[g1] = add ? o1 + v1 : v1;
}

template <typename T>
__global__ void __launch_bounds__(256)
reduce_classes(T *__restrict__ dst, const T *__restrict__ src, const T *__restrict__ halo,
               const uint32_t *__restrict__ arr, const uint4 *__restrict__ tiles, int add) {
  const uint4 td = tiles[blockIdx.x];  // {class base in arr, k, entries of the class (padded), first entry of the tile}
  const uint32_t *p = arr + td.x;
  const uint32_t i0 = td.w + threadIdx.x;
  switch (td.y) {  // wave-uniform
    case 1: reduce_two<T, 1>(dst, src, halo, p, td.z, i0, 1, add); break;
    case 2: reduce_two<T, 2>(dst, src, halo, p, td.z, i0, 2, add); break;
    case 3: reduce_two<T, 3>(dst, src, halo, p, td.z, i0, 3, add); break;
    case 4: reduce_two<T, 4>(dst, src, halo, p, td.z, i0, 4, add); break;
    case 8: reduce_two<T, 8>(dst, src, halo, p, td.z, i0, 8, add); break;
    default: reduce_two<T, 0>(dst, src, halo, p, td.z, i0, td.y, add); break;
  }
}

}  // namespace

// Host side: (sdofs, s_off, s_idx) of the plan -> class arrays.  `arr` = for every class [dofs | slots_0 | ... |
// slots_{k-1}], each `entries` long; `tiles` = one uint4 per 512 entries.
void build_pass2_classes(const std::vector<uint32_t> &sdofs, const std::vector<uint32_t> &s_off,
                         const std::vector<uint32_t> &s_idx, std::vector<uint32_t> &arr, std::vector<uint32_t> &tiles) {
  arr.clear();
  tiles.clear();
  const size_t ns = sdofs.size();
  uint32_t kmax = 0;
  for (size_t i = 0; i < ns; ++i) kmax = std::max(kmax, s_off[i + 1] - s_off[i]);
  std::vector<std::vector<uint32_t>> members(kmax + 1);
  for (size_t i = 0; i < ns; ++i) {
    uint32_t k = s_off[i + 1] - s_off[i];
    if (k == 0) k = 1;  // constrained dof listed without a partial sum: its slot entries are never read
    members[k].push_back((uint32_t)i);
  }
  for (uint32_t k = 1; k <= kmax; ++k) {
    const std::vector<uint32_t> &m = members[k];
    if (m.empty()) continue;
    const uint32_t entries = (uint32_t)((m.size() + 511) / 512 * 512);
    const uint32_t base = (uint32_t)arr.size();
    arr.resize(arr.size() + (size_t)(1 + k) * entries, 0u);
    for (uint32_t e = 0; e < entries; ++e) arr[base + e] = e < m.size() ? sdofs[m[e]] : 0xffffffffu;
    for (size_t e = 0; e < m.size(); ++e) {
      const uint32_t i = m[e], cnt = s_off[i + 1] - s_off[i];
      for (uint32_t t = 0; t < k; ++t) arr[base + (size_t)(1 + t) * entries + e] = t < cnt ? s_idx[s_off[i] + t] : 0u;
    }
    for (uint32_t first = 0; first < entries; first += 512) {
      tiles.push_back(base);
      tiles.push_back(k);
      tiles.push_back(entries);
      tiles.push_back(first);
    }
  }
}

template <typename T>
hipError_t reduce_classes_launch(T *dst, const T *src, const T *halo, const uint32_t *arr, const uint32_t *tiles,
                                 uint32_t n_tiles, int add, hipStream_t st) {
  if (n_tiles == 0) return hipSuccess;
  hipLaunchKernelGGL(reduce_classes<T>, dim3(n_tiles), dim3(256), 0, st, dst, src, halo, arr,
                     reinterpret_cast<const uint4 *>(tiles), add);
  return hipGetLastError();
}
template hipError_t reduce_classes_launch<double>(double *, const double *, const double *, const uint32_t *,
                                                  const uint32_t *, uint32_t, int, hipStream_t);
template hipError_t reduce_classes_launch<float>(float *, const float *, const float *, const uint32_t *,
                                                 const uint32_t *, uint32_t, int, hipStream_t);

}  // namespace mfgpu
