// Single-node multi-GPU mode behind the C-ABI: one process per GPU, cells sharded by z-slab, the partial sums on a
// slab's two interface planes exchanged with the z-neighbours over RCCL (xGMI point-to-point links) after the
// local cell loop.  The reference is single-GPU (no MPI / NCCL anywhere); this is new work
// shaped by SURVEY.md 8e:
//
//   mfgpu_vmult_dist_begin   cell loop over the batches that touch an interface plane   (launch stream), event
//                            pass 2 of the INTERFACE dofs, pack the planes,            (side stream, behind the event)
//                            grouped ncclSend / ncclRecv with both neighbours          (side stream)
//                            cell loop over the interior batches, pass 2 of the rest   (launch stream) -- overlap the
//                                                                                        whole exchange
//   mfgpu_vmult_dist_end     wait for the exchange; dst[plane] += received, constrained rows excepted (identity
//                            rows on both sides, laplace_operator_gpu.h:300-302)
//
// (cells touching an interface first, exchange, interior cells overlap the exchange.)  The
// interface batches are found from the plan at attach time: the batches listing an interface dof lie, in plan order,
// at the two ends of the slab's batch sequence (the planner keeps the caller's z-major cell order); the longest run
// of batches that touch no interface dof is the interior.  If the interface batches are more than 60 % of the slab
// (thin slabs), or the operator runs its cell loop in several segments or in coloured mode, the round-2 schedule is
// used instead: whole cell loop, pass 2 of the interface dofs, pack, exchange next to pass 2 of the rest.
//
// A slab shares dofs only with its two neighbours, so the exchange is two point-to-point transfers per rank, not a
// collective over all ranks.  Transports: RCCL (ncclSend / ncclRecv on a communicator created from a unique id), or
// IN-PROCESS (mfgpu_dist_connect_local: the neighbours are objects of the same process and the transfer is a device
// copy) -- the second exists so that the whole path except the two RCCL calls runs in the single-GPU test suite.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "mfgpu_internal.h"

using namespace mfgpu;

#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
      return e_ == hipErrorOutOfMemory ? MFGPU_ENOMEM : MFGPU_EHIP;                      \
    }                                                                                    \
  } while (0)
#define NCCL_TRY(expr)                                                                   \
  do {                                                                                   \
    ncclResult_t r_ = (expr);                                                            \
    if (r_ != ncclSuccess) {                                                             \
      set_error(std::string(#expr) + ": " + ncclGetErrorString(r_));                     \
      return MFGPU_EHIP;                                                                 \
    }                                                                                    \
  } while (0)

struct mfgpu_dist {
  int rank = 0, world = 1, number_type = MFGPU_F64;
  ncclComm_t comm = nullptr;            // RCCL transport, or
  mfgpu_dist *local_peer[2] = {};       // in-process transport: the lower / upper neighbour's object
  uint32_t n_if[2] = {0, 0};            // interface dofs on the lower / upper plane
  std::vector<uint32_t> ids[2];         // their local dof ids (host copy: priority dofs of the operator)
  // device copies: ONE allocation each for both planes ([lower | upper]: one pack and one add launch per apply, a tiny
  // launch costs ~5 us on the stream it sits in); [w] points at plane w's part
  uint32_t *d_ids[2] = {};
  uint8_t *d_free[2] = {};              // 1: summed with the neighbour, 0: constrained (identity row)
  void *d_send[2] = {}, *d_recv[2] = {};
  void *d_base[4] = {};                 // the four allocations (ids, free, send, recv)
  hipStream_t side = nullptr;
  hipEvent_t ev_packed = nullptr, ev_done = nullptr, ev_if = nullptr;
  bool in_flight = false;
  // interface-first schedule (set by mfgpu_dist_attach): batches [0, r1_end) and [r2_begin, n_batches) touch an
  // interface plane, [r1_end, r2_begin) is the interior
  bool interface_first = false;
  uint32_t r1_end = 0, r2_begin = 0, n_batches = 0;
};

namespace {

template <typename T>
__global__ void pack_kernel(T *out, const T *vec, const uint32_t *ids, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = vec[ids[i]];
}
template <typename T>
__global__ void add_kernel(T *vec, const T *in, const uint32_t *ids, const uint8_t *free_, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && free_[i]) vec[ids[i]] += in[i];
}
size_t esize(int nt) { return nt == MFGPU_F32 ? 4 : 8; }

int pack_planes(mfgpu_dist *d, const void *vec, hipStream_t st) {
  const uint32_t n = d->n_if[0] + d->n_if[1];  // both planes in one launch (contiguous [lower | upper])
  if (!n) return 0;
  const unsigned grid = (n + 255) / 256;
  if (d->number_type == MFGPU_F64)
    hipLaunchKernelGGL(pack_kernel<double>, dim3(grid), dim3(256), 0, st, (double *)d->d_base[2], (const double *)vec,
                       (const uint32_t *)d->d_base[0], n);
  else
    hipLaunchKernelGGL(pack_kernel<float>, dim3(grid), dim3(256), 0, st, (float *)d->d_base[2], (const float *)vec,
                       (const uint32_t *)d->d_base[0], n);
  HIP_TRY(hipGetLastError());
  return 0;
}

// grouped send / recv with both neighbours on the side stream, behind the pack (which ran on `st`: the launch stream,
// or the side stream itself)
int start_exchange(mfgpu_dist *d, hipStream_t st) {
  HIP_TRY(hipEventRecord(d->ev_packed, st));
  d->in_flight = true;
  if (!d->comm) return 0;  // in-process transport: the copies happen in finish_exchange, when every peer has packed
  if (st != d->side) HIP_TRY(hipStreamWaitEvent(d->side, d->ev_packed, 0));
  const ncclDataType_t dt = d->number_type == MFGPU_F64 ? ncclDouble : ncclFloat;
  NCCL_TRY(ncclGroupStart());
  // (a failing call must not leave the group open: the group depth is state of the calling thread, and an open group
  // would defer every later RCCL call of this process, the caller's own included)
  ncclResult_t bad = ncclSuccess;
  for (int w = 0; w < 2 && bad == ncclSuccess; ++w) {
    if (!d->n_if[w]) continue;
    const int peer = w == 0 ? d->rank - 1 : d->rank + 1;
    bad = ncclSend(d->d_send[w], d->n_if[w], dt, peer, d->comm, d->side);
    if (bad == ncclSuccess) bad = ncclRecv(d->d_recv[w], d->n_if[w], dt, peer, d->comm, d->side);
  }
  const ncclResult_t end = ncclGroupEnd();
  if (bad != ncclSuccess || end != ncclSuccess) {
    set_error(std::string("grouped ncclSend / ncclRecv with the slab's neighbours: ") +
              ncclGetErrorString(bad != ncclSuccess ? bad : end));
    d->in_flight = false;
    return MFGPU_EHIP;
  }
  HIP_TRY(hipEventRecord(d->ev_done, d->side));
  return 0;
}

int finish_exchange(mfgpu_dist *d, void *vec, hipStream_t st) {
  if (!d->in_flight) {
    set_error("mfgpu_vmult_dist_end without a matching _begin");
    return MFGPU_EINVAL;
  }
  d->in_flight = false;
  if (d->comm) {
    HIP_TRY(hipStreamWaitEvent(st, d->ev_done, 0));
  } else {
    HIP_TRY(hipStreamWaitEvent(st, d->ev_packed, 0));  // this slab's own planes are complete (side stream)
    for (int w = 0; w < 2; ++w) {
      if (!d->n_if[w]) continue;
      mfgpu_dist *p = d->local_peer[w];
      if (!p || p->n_if[1 - w] != d->n_if[w]) {
        set_error("in-process transport: neighbour not connected (mfgpu_dist_connect_local)");
        return MFGPU_EINVAL;
      }
      // my lower plane is the neighbour's upper plane and vice versa
      HIP_TRY(hipStreamWaitEvent(st, p->ev_packed, 0));
      HIP_TRY(hipMemcpyAsync(d->d_recv[w], p->d_send[1 - w], (size_t)d->n_if[w] * esize(d->number_type),
                             hipMemcpyDeviceToDevice, st));
    }
  }
  const uint32_t n = d->n_if[0] + d->n_if[1];  // both planes in one launch (the two planes share no dof)
  if (n) {
    const unsigned grid = (n + 255) / 256;
    if (d->number_type == MFGPU_F64)
      hipLaunchKernelGGL(add_kernel<double>, dim3(grid), dim3(256), 0, st, (double *)vec, (const double *)d->d_base[3],
                         (const uint32_t *)d->d_base[0], (const uint8_t *)d->d_base[1], n);
    else
      hipLaunchKernelGGL(add_kernel<float>, dim3(grid), dim3(256), 0, st, (float *)vec, (const float *)d->d_base[3],
                         (const uint32_t *)d->d_base[0], (const uint8_t *)d->d_base[1], n);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

}  // namespace

extern "C" {

int mfgpu_dist_unique_id(void *id128) {
  if (!id128) return MFGPU_EINVAL;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  std::memcpy(id128, &id, sizeof(id));
  return 0;
}

int mfgpu_dist_create(const void *id128, int rank, int world, const uint32_t *lower_ids, uint32_t n_lower,
                      const uint32_t *upper_ids, uint32_t n_upper, const uint32_t *constrained, uint32_t n_constrained,
                      uint32_t n_dofs, int number_type, mfgpu_dist **out) {
  if (!out || world < 1 || rank < 0 || rank >= world || (n_lower && !lower_ids) || (n_upper && !upper_ids) ||
      (n_constrained && !constrained) || (number_type != MFGPU_F64 && number_type != MFGPU_F32)) {
    set_error("mfgpu_dist_create: bad argument");
    return MFGPU_EINVAL;
  }
  if ((rank == 0 && n_lower) || (rank == world - 1 && n_upper)) {
    set_error("mfgpu_dist_create: the first / last slab has no lower / upper neighbour");
    return MFGPU_EINVAL;
  }
  mfgpu_dist *d = new mfgpu_dist();
  d->rank = rank;
  d->world = world;
  d->number_type = number_type;
  std::vector<uint8_t> con(n_dofs, 0);
  for (uint32_t i = 0; i < n_constrained; ++i)
    if (constrained[i] < n_dofs) con[constrained[i]] = 1;
  const uint32_t *src_ids[2] = {lower_ids, upper_ids};
  const uint32_t cnt[2] = {n_lower, n_upper};
  auto fail = [&](int rc) {
    mfgpu_dist_destroy(d);
    return rc;
  };
  std::vector<uint32_t> ids_all;
  std::vector<uint8_t> free_all;
  for (int w = 0; w < 2; ++w) {
    d->n_if[w] = cnt[w];
    if (!cnt[w]) continue;
    d->ids[w].assign(src_ids[w], src_ids[w] + cnt[w]);
    for (uint32_t i = 0; i < cnt[w]; ++i) {
      if (src_ids[w][i] >= n_dofs) {
        set_error("mfgpu_dist_create: interface dof out of range");
        return fail(MFGPU_EINVAL);
      }
      ids_all.push_back(src_ids[w][i]);
      free_all.push_back(con[src_ids[w][i]] ? 0 : 1);
    }
  }
  const size_t n_all = ids_all.size(), es = esize(number_type);
  if (n_all) {
    if (hipMalloc(&d->d_base[0], n_all * 4) != hipSuccess || hipMalloc(&d->d_base[1], n_all) != hipSuccess ||
        hipMalloc(&d->d_base[2], n_all * es) != hipSuccess || hipMalloc(&d->d_base[3], n_all * es) != hipSuccess ||
        hipMemcpy(d->d_base[0], ids_all.data(), n_all * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d->d_base[1], free_all.data(), n_all, hipMemcpyHostToDevice) != hipSuccess) {
      set_error("mfgpu_dist_create: device allocation failed");
      return fail(MFGPU_ENOMEM);
    }
    for (int w = 0; w < 2; ++w) {
      const size_t off = w ? cnt[0] : 0;
      d->d_ids[w] = (uint32_t *)d->d_base[0] + off;
      d->d_free[w] = (uint8_t *)d->d_base[1] + off;
      d->d_send[w] = (char *)d->d_base[2] + off * es;
      d->d_recv[w] = (char *)d->d_base[3] + off * es;
    }
  }
  // highest priority: the exchange's short kernels (pass 2 of the planes, pack, RCCL's send / recv) should get wave
  // slots ahead of the interior cell loop that runs beside them
  int prio_least = 0, prio_greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_greatest = 0;
  if (hipStreamCreateWithPriority(&d->side, hipStreamNonBlocking, prio_greatest) != hipSuccess ||
      hipEventCreateWithFlags(&d->ev_packed, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&d->ev_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&d->ev_if, hipEventDisableTiming) != hipSuccess) {
    set_error("mfgpu_dist_create: stream / event creation failed");
    return fail(MFGPU_EHIP);
  }
  if (id128 && world > 1) {
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&d->comm, world, id, rank);
    if (r != ncclSuccess) {
      set_error(std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
      d->comm = nullptr;
      return fail(MFGPU_EHIP);
    }
  }
  *out = d;
  return 0;
}

int mfgpu_dist_connect_local(mfgpu_dist *lower_rank, mfgpu_dist *upper_rank) {
  if (!lower_rank || !upper_rank || lower_rank->comm || upper_rank->comm ||
      lower_rank->n_if[1] != upper_rank->n_if[0] || lower_rank->number_type != upper_rank->number_type) {
    set_error("mfgpu_dist_connect_local: not two adjacent in-process slabs");
    return MFGPU_EINVAL;
  }
  lower_rank->local_peer[1] = upper_rank;
  upper_rank->local_peer[0] = lower_rank;
  return 0;
}

int mfgpu_dist_attach(mfgpu_dist *d, mfgpu_handle *h) {
  if (!d || !h || handle_number_type(h) != d->number_type) {
    set_error("mfgpu_dist_attach: null argument or number type mismatch");
    return MFGPU_EINVAL;
  }
  std::vector<uint32_t> all(d->ids[0]);
  all.insert(all.end(), d->ids[1].begin(), d->ids[1].end());
  int rc = handle_set_priority_dofs(h, all.data(), (uint32_t)all.size());
  if (rc) return rc;
  // interface-first schedule: the longest run of batches that touch no interface dof is the interior
  d->interface_first = false;
  d->n_batches = (uint32_t)handle_n_batches(h);
  if (all.empty() || !handle_ranged_ok(h) || d->n_batches < 3) return 0;
  std::vector<uint8_t> touch;
  if ((rc = handle_batches_touching(h, all.data(), (uint32_t)all.size(), touch))) return rc;
  uint32_t best_lo = 0, best_len = 0, run_lo = 0;
  for (uint32_t b = 0; b <= d->n_batches; ++b)
    if (b == d->n_batches || touch[b]) {
      if (b - run_lo > best_len) {
        best_len = b - run_lo;
        best_lo = run_lo;
      }
      run_lo = b + 1;
    }
  if ((uint64_t)best_len * 10 >= (uint64_t)d->n_batches * 4) {  // interior >= 40 % of the slab's batches
    d->interface_first = true;
    d->r1_end = best_lo;
    d->r2_begin = best_lo + best_len;
  }
  return 0;
}

int mfgpu_dist_schedule(const mfgpu_dist *d, uint32_t info[4]) {
  if (!d || !info) return MFGPU_EINVAL;
  info[0] = d->interface_first ? 1u : 0u;
  info[1] = d->r1_end;
  info[2] = d->r2_begin;
  info[3] = d->n_batches;
  return 0;
}

int mfgpu_vmult_dist_begin(mfgpu_handle *h, mfgpu_dist *d, void *dst, const void *src, void *stream) {
  if (!h || !d || !dst || !src || dst == src) {
    set_error("mfgpu_vmult_dist_begin: null or aliasing argument");
    return MFGPU_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  if (d->interface_first && (uint32_t)handle_n_batches(h) == d->n_batches) {
    // SURVEY.md 8e steps 1-3.  The side stream writes dst on the interface planes only (pass 2 of the priority dofs);
    // the interior batches and the rest of pass 2 write every other entry: no two streams touch the same entry.
    int rc = handle_cells_two_ranges(h, 0, d->r1_end, d->r2_begin, d->n_batches, dst, src, stream, 0);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(d->ev_if, st));
    HIP_TRY(hipStreamWaitEvent(d->side, d->ev_if, 0));
    rc = handle_pass2_group(h, 0, dst, src, d->side, 0);  // the interface planes hold the slab's sums
    if (!rc) rc = pack_planes(d, dst, d->side);
    if (!rc) rc = start_exchange(d, d->side);
    if (!rc) rc = handle_cells_range(h, d->r1_end, d->r2_begin, dst, src, stream, 0);  // overlaps the exchange
    if (!rc) rc = handle_pass2_group(h, 1, dst, src, stream, 0);
    return rc;
  }
  int rc = handle_vmult_phase(h, 0, dst, src, stream, 0);
  if (!rc) rc = handle_vmult_phase(h, 1, dst, src, stream, 0);  // the interface planes are complete
  if (!rc) rc = pack_planes(d, dst, st);
  if (!rc) rc = start_exchange(d, st);
  if (!rc) rc = handle_vmult_phase(h, 2, dst, src, stream, 0);  // overlaps the exchange
  return rc;
}

int mfgpu_vmult_dist_end(mfgpu_handle *h, mfgpu_dist *d, void *dst, void *stream) {
  if (!h || !d || !dst) {
    set_error("mfgpu_vmult_dist_end: null argument");
    return MFGPU_EINVAL;
  }
  return finish_exchange(d, dst, (hipStream_t)stream);
}

int mfgpu_vmult_dist(mfgpu_handle *h, mfgpu_dist *d, void *dst, const void *src, void *stream) {
  if (d && !d->comm && (d->n_if[0] || d->n_if[1])) {
    // in-process transport: _end copies the neighbour's packed planes, which exist only once the neighbour's _begin
    // has run -- begin ALL slabs, then end all of them (a wait on a never-recorded event is no wait)
    set_error("mfgpu_vmult_dist on the in-process transport: call mfgpu_vmult_dist_begin for every slab, then _end");
    return MFGPU_EINVAL;
  }
  int rc = mfgpu_vmult_dist_begin(h, d, dst, src, stream);
  if (!rc) rc = mfgpu_vmult_dist_end(h, d, dst, stream);
  return rc;
}

void mfgpu_dist_destroy(mfgpu_dist *d) {
  if (!d) return;
  if (d->comm) ncclCommDestroy(d->comm);
  for (int w = 0; w < 2; ++w)
    if (d->local_peer[w]) d->local_peer[w]->local_peer[1 - w] = nullptr;
  for (void *p : d->d_base) hipFree(p);
  if (d->side) hipStreamDestroy(d->side);
  if (d->ev_packed) hipEventDestroy(d->ev_packed);
  if (d->ev_done) hipEventDestroy(d->ev_done);
  if (d->ev_if) hipEventDestroy(d->ev_if);
  delete d;
}

}  // extern "C"
