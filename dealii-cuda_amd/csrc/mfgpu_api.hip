// C-ABI entry points (include/mfgpu.h): handle life cycle, vmult / vmult_add, GpuVector pieces.
#include <hip/hip_runtime.h>

#include <type_traits>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mfgpu_kernels.h"

using namespace mfgpu;

#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
      return e_ == hipErrorOutOfMemory ? MFGPU_ENOMEM : MFGPU_EHIP;                      \
    }                                                                                    \
  } while (0)

struct mfgpu_handle {
  Plan plan;
  int dim = 0, n = 0, nd = 0, number_type = MFGPU_F64;
  bool hn = false;
  std::vector<double> S, Dt;
  std::vector<double> sv, sg;  // the caller's 1D tables (diagonal)
  // device arrays
  uint32_t *d_batch_cell_off = nullptr, *d_batch_dof_off = nullptr, *d_bdofs = nullptr;
  uint8_t *d_bflags = nullptr;
  uint16_t *d_lmap = nullptr;
  uint16_t *d_lmapx = nullptr;
  uint16_t *d_perm = nullptr;  // apply_batches_x: bank-conflict-free lane -> pencil maps of the y- and z-stage
  // apply_planes3: fixed-size per-batch records (see ApplyArgs)
  uint32_t *d_bdofsp = nullptr, *d_idxp = nullptr;
  uint32_t *d_hnrec = nullptr;  // apply_planes3<HN>: per-batch records of the hanging-node line operations
  uint32_t *d_hn_slot = nullptr;  // ... and per plane batch the index of its record (0xffffffff: none)
  void *d_coefp = nullptr;
  void *d_coef = nullptr;
  uint32_t *d_cmask = nullptr, *d_orphans = nullptr;
  void *d_hnw = nullptr;
  uint32_t *d_constrained = nullptr;  // constrained dof list (set_constrained_values)
  uint32_t n_constrained = 0;
  void *d_tabsd = nullptr;  // apply_batches_g2: [S | Dt] full 1D tables in the operator's number type
  void *d_tab2 = nullptr;  // [2][n*n] squared 1D tables of the diagonal kernel, built on first use
  // two-pass mode
  bool twopass = true;
  uint32_t *d_batch_nint = nullptr, *d_halo_off = nullptr;
  // The cell loop runs in SEGMENTS of consecutive batches, one launch each (seg_end[s] = one past the segment's last
  // batch).  Pass 2, class-sorted form (mfgpu_pass2.hip), in 1 + n_segments groups: [0] the priority dofs (mfgpu_dist:
  // the slab's interface planes, reduced first after the whole cell loop so that their exchange overlaps the rest),
  // [1 + s] the other dofs whose LAST toucher batch lies in segment s.  Group 1 + s needs segments 0..s only, so for
  // s < last it runs on the handle's side stream while the next segment's cells are computed (a latency-bound kernel
  // next to an issue-bound one); the last group and the priority group run on the caller's stream.
  std::vector<uint32_t> seg_end;
  std::vector<uint32_t *> d_p2arr, d_p2tiles;
  std::vector<uint32_t> n_p2tiles;
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> ev_seg;  // [s]: segment s done (recorded on the caller's stream)
  hipEvent_t ev_side = nullptr;    // the side stream's pass-2 launches of this vmult done
  bool side_pending = false;       // the caller's stream has not joined the side stream yet
  void *d_halo = nullptr;
  unsigned long long *d_stamps = nullptr;  // diagnostic build only
  size_t lds = 0, device_bytes = 0;
  uint32_t max_grid = 0;    // resident workgroups of the cell-loop kernel
  uint32_t max_grid_p = 0;  // ... of apply_planes3 (its batches: the first plan.n_plane_batches)
  uint32_t max_grid_ph = 0;  // ... of apply_planes3<HN> (the plane batches of cells with a hanging-node mask)
  bool xk = false;        // 3D two-pass kernel for three workgroups per CU (apply_batches_x)
  bool gk = false;        // general-Jacobian kernel (apply_batches_g)
  bool pk = false;        // plane-per-thread kernel: 3D uniform-Jacobian default for p = 4
  bool qk = false;        // ... apply_planes4 (MFGPU_KERNEL_PLANES_2W) instead of apply_planes3
  // profiling
  bool prof = false;
  std::vector<hipEvent_t> ev;  // start/stop pairs
  size_t ev_used = 0;
  double prof_ms = 0.0;
  uint64_t prof_vmults = 0;
  std::vector<hipEvent_t> ev2;  // start/stop pairs around pass 2 (mfgpu_vmult / mfgpu_vmult_add only)
  size_t ev2_used = 0;
  double prof2_ms = 0.0;
};

namespace {

template <typename T>
hipError_t planes_launch(mfgpu_handle *h, const ApplyArgs<T> &a, bool hn, uint32_t grid, hipStream_t st,
                         bool configure_only, size_t *lds_out, int *occupancy);

template <typename P>
int dev_upload(P **dst, const void *src, size_t bytes, size_t &acct) {
  *dst = nullptr;
  if (bytes == 0) return 0;
  HIP_TRY(hipMalloc((void **)dst, bytes));
  HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  acct += bytes;
  return 0;
}

// symmetry of the 1D tables (see mfgpu_kernels.hip tab_at); also makes mirrored entries bit-equal
int check_symmetrize(int n, std::vector<double> &S, std::vector<double> &Dt) {
  const int p = n - 1;
  double err = 0, mag = 0;
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) {
      const double s1 = S[r * n + c], s2 = S[(p - r) * n + (p - c)];
      const double d1 = Dt[r * n + c], d2 = -Dt[(p - r) * n + (p - c)];
      err = std::fmax(err, std::fmax(std::fabs(s1 - s2), std::fabs(d1 - d2)));
      mag = std::fmax(mag, std::fmax(std::fabs(s1), std::fabs(d1)));
    }
  if (err > 1e-10 * mag) {
    set_error("shape tables are not symmetric about the cell midpoint (unsupported)");
    return MFGPU_EUNSUPPORTED;
  }
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) {
      const int r2 = p - r, c2 = p - c;
      if (r * n + c < r2 * n + c2) {
        const double s = 0.5 * (S[r * n + c] + S[r2 * n + c2]);
        S[r * n + c] = S[r2 * n + c2] = s;
        const double d = 0.5 * (Dt[r * n + c] - Dt[r2 * n + c2]);
        Dt[r * n + c] = d;
        Dt[r2 * n + c2] = -d;
      } else if (r == r2 && c == c2) {
        Dt[r * n + c] = 0.0;
      }
    }
  return 0;
}

// pass-2 arrays from the plan: group 0 = the dofs listed in `priority` (may be empty), group 1 = the rest.  Pass 2 also
// writes the dofs no cell touches (hanging dofs after substitution: dst = 0, or the identity row of a constrained one):
// listed with ONE partial sum, a halo slot behind the batches' that is zero and stays zero.
int upload_pass2(mfgpu_handle *h, const uint32_t *priority, uint32_t n_priority) {
  const Plan &P = h->plan;
  std::vector<uint8_t> prio(P.n_dofs, 0);
  for (uint32_t i = 0; i < n_priority; ++i) {
    if (priority[i] >= P.n_dofs) {
      set_error("priority dof out of range");
      return MFGPU_EINVAL;
    }
    prio[priority[i]] = 1;
  }
  const size_t ng = 1 + h->seg_end.size();
  std::vector<std::vector<uint32_t>> sd(ng), so(ng, std::vector<uint32_t>(1, 0u)), si(ng);
  auto add = [&](uint32_t dof, const uint32_t *slots, uint32_t k, size_t seg) {
    const size_t g = prio[dof & 0x7fffffffu] ? 0 : 1 + seg;
    sd[g].push_back(dof);
    si[g].insert(si[g].end(), slots, slots + k);
    so[g].push_back((uint32_t)si[g].size());
  };
  for (size_t i = 0; i < P.sdofs.size(); ++i) {
    // the slots of a dof are listed in ascending batch order: the last one belongs to its last toucher
    size_t seg = 0;
    if (P.s_off[i + 1] > P.s_off[i]) {
      const uint32_t slot = P.s_idx[P.s_off[i + 1] - 1];
      const uint32_t batch = (uint32_t)(std::upper_bound(P.halo_off.begin(), P.halo_off.end(), slot) - P.halo_off.begin()) - 1;
      seg = (size_t)(std::upper_bound(h->seg_end.begin(), h->seg_end.end(), batch) - h->seg_end.begin());
      if (seg >= h->seg_end.size()) seg = h->seg_end.size() - 1;
    }
    add(P.sdofs[i], P.s_idx.data() + P.s_off[i], P.s_off[i + 1] - P.s_off[i], seg);
  }
  const uint32_t zero_slot = P.halo_off.empty() ? 0u : P.halo_off.back();
  for (uint32_t orph : P.orphans) add(orph, &zero_slot, 1, 0);  // depend on no batch
  for (size_t g = 0; g < h->d_p2arr.size(); ++g) {
    hipFree(h->d_p2arr[g]);
    hipFree(h->d_p2tiles[g]);
  }
  h->d_p2arr.assign(ng, nullptr);
  h->d_p2tiles.assign(ng, nullptr);
  h->n_p2tiles.assign(ng, 0u);
  for (size_t g = 0; g < ng; ++g) {
    std::vector<uint32_t> arr, tiles;
    build_pass2_classes(sd[g], so[g], si[g], arr, tiles);
    h->n_p2tiles[g] = (uint32_t)(tiles.size() / 4);
    // Tile order = the order the workgroups of pass 2 are dispatched in.  The builder's order is ascending in the
    // position of a dof's first toucher; while the halo buffer fits the 256 MB Infinity Cache the REVERSE is faster --
    // the partial sums the cell loop wrote last are read first, from cache: pass 2 42.4 -> 41.3 us on C2, 38.9 -> 35.3
    // on C5, -1 .. -3 % per vmult up to 38 M dofs; beyond (n = 96: 57 M dofs, 340 MB of partial sums) it is slower, at
    // 81 M dofs by 10 % (profiles/r03_notes.md section 13)
    const size_t esz = h->number_type == MFGPU_F64 ? 8 : 4;
    if ((size_t)P.halo_off.back() * esz <= ((size_t)256 << 20)) {
      const size_t nt = tiles.size() / 4;
      for (size_t a = 0, b = nt ? nt - 1 : 0; a < b; ++a, --b)
        for (int w = 0; w < 4; ++w) std::swap(tiles[4 * a + w], tiles[4 * b + w]);
    }
    size_t acct = 0;
    int rc;
    if ((rc = dev_upload(&h->d_p2arr[g], arr.data(), arr.size() * 4, acct))) return rc;
    if ((rc = dev_upload(&h->d_p2tiles[g], tiles.data(), tiles.size() * 4, acct))) return rc;
  }
  return 0;
}

template <typename T>
int create_arrays(mfgpu_handle *h, const mfgpu_desc &d) {
  const Plan &P = h->plan;
  const size_t ncell = P.n_cells, nd = (size_t)P.nd;
  size_t &acct = h->device_bytes;
  int rc;
  if ((rc = dev_upload(&h->d_batch_cell_off, P.batch_cell_off.data(), P.batch_cell_off.size() * 4, acct))) return rc;
  if ((rc = dev_upload(&h->d_batch_dof_off, P.batch_dof_off.data(), P.batch_dof_off.size() * 4, acct))) return rc;
  if ((rc = dev_upload(&h->d_bdofs, P.bdofs.data(), P.bdofs.size() * 4, acct))) return rc;
  if ((rc = dev_upload(&h->d_bflags, P.bflags.data(), P.bflags.size(), acct))) return rc;
  if ((rc = dev_upload(&h->d_lmap, P.lmap.data(), P.lmap.size() * 2, acct))) return rc;
  if (h->xk && !h->hn) {
    // Lane -> pencil maps of the y- and z-stage (see apply_batches_x).  LDS rules (MI355X_MICROARCH.md): a
    // ds_read_b64 is served in 32-lane groups, a double occupies slot (index mod 32); ds_write_b64 / ds_read2_b64
    // in 16-lane groups, slot (index mod 16).  All n elements of a pencil shift its base by the same stride, so
    // only the bases matter: the pencil whose base has residue r mod 32 gets lane 32 k + r (k-th pencil with that
    // residue) -- distinct slots in every 32-lane group, and in each of its 16-lane halves.  Residue classes with
    // more than 8 pencils (4 pencils at p=4 in the y-stage, 2 in the z-stage) overflow into the idle lanes, which
    // all sit in the last group.
    const int n = P.n, n2 = n * n, PP = n2, CH = 256 / PP, ndl = n2 * n;
    std::vector<uint16_t> perm(512, 0xffff);
    for (int layout = 0; layout < 2; ++layout) {
      uint16_t *lanes = perm.data() + 256 * layout;
      std::vector<int> fill(32, 0), overflow;
      for (int q = 0; q < CH * PP; ++q) {
        const int cell = q / PP, pen = q % PP, a = pen % n, b = pen / n;
        const int base = cell * ndl + (layout == 0 ? a + n2 * b : a + n * b);
        const int r = base & 31;
        if (fill[r] < 8) lanes[32 * fill[r]++ + r] = (uint16_t)q;
        else overflow.push_back(q);
      }
      for (int l = 255; l >= 0 && !overflow.empty(); --l)
        if (lanes[l] == 0xffff) {
          lanes[l] = (uint16_t)overflow.back();
          overflow.pop_back();
        }
    }
    if ((rc = dev_upload(&h->d_perm, perm.data(), perm.size() * 2, acct))) return rc;
  }
  if (h->xk || h->gk) {
    // x-pencil index runs (n contiguous entries of lmap) padded to whole 32-bit words
    const size_t n = (size_t)P.n, np = (n + 1) & ~(size_t)1, runs = P.lmap.size() / n;
    std::vector<uint16_t> lx(runs * np, 0);
    for (size_t r = 0; r < runs; ++r)
      for (size_t i = 0; i < n; ++i) lx[r * np + i] = P.lmap[r * n + i];
    if ((rc = dev_upload(&h->d_lmapx, lx.data(), lx.size() * 2, acct))) return rc;
  }
  if (h->pk) {
    if ((rc = build_plane_records(h->plan, d.constraint_mask))) return rc;
    if (!P.pr_hn.empty()) {
      if ((rc = dev_upload(&h->d_hnrec, P.pr_hn.data(), P.pr_hn.size() * 4, acct))) return rc;
      if ((rc = dev_upload(&h->d_hn_slot, P.pr_hn_slot.data(), P.pr_hn_slot.size() * 4, acct))) return rc;
    }
    if ((rc = dev_upload(&h->d_bdofsp, P.pr_dofs.data(), P.pr_dofs.size() * 4, acct))) return rc;
    if ((rc = dev_upload(&h->d_idxp, P.pr_idx.data(), P.pr_idx.size() * 4, acct))) return rc;
  }
  if ((rc = dev_upload(&h->d_orphans, P.orphans.data(), P.orphans.size() * 4, acct))) return rc;
  if (h->twopass) {
    if ((rc = dev_upload(&h->d_batch_nint, P.batch_nint.data(), P.batch_nint.size() * 4, acct))) return rc;
    if ((rc = dev_upload(&h->d_halo_off, P.halo_off.data(), P.halo_off.size() * 4, acct))) return rc;
    if (h->pk && (uint64_t)P.halo_off.back() >= (1ull << 29)) {
      set_error("halo buffer too large for 32-bit byte offsets");
      return MFGPU_EUNSUPPORTED;
    }
    const size_t hb = ((size_t)P.halo_off.back() + 1) * sizeof(T);  // + the always-zero slot of the untouched dofs
    if (hb) {
      HIP_TRY(hipMalloc(&h->d_halo, hb));
      HIP_TRY(hipMemset(h->d_halo, 0, hb));
      acct += hb;
    }
  }
  if (h->hn) {
    std::vector<uint32_t> cm(ncell);
    for (size_t i = 0; i < ncell; ++i) cm[i] = d.constraint_mask[P.cell_order[i]];
    if ((rc = dev_upload(&h->d_cmask, cm.data(), ncell * 4, acct))) return rc;
    std::vector<T> w((size_t)h->n * h->n);
    for (size_t i = 0; i < w.size(); ++i) w[i] = (T)d.constraint_weights[i];
    if ((rc = dev_upload((T **)&h->d_hnw, w.data(), w.size() * sizeof(T), acct))) return rc;
  }
  // coefficient (given, or evaluated on the device from the quadrature points), then folded
  size_t tmp = 0;
  T *t_coef = nullptr, *t_jxw = nullptr, *t_j0 = nullptr, *t_q = nullptr;
  uint32_t *t_order = nullptr;
  auto cleanup = [&]() {
    hipFree(t_coef);
    hipFree(t_jxw);
    hipFree(t_j0);
    hipFree(t_q);
    hipFree(t_order);
  };
  if (d.coefficient) {
    if ((rc = dev_upload(&t_coef, d.coefficient, ncell * nd * sizeof(T), tmp))) { cleanup(); return rc; }
  } else {
    if ((rc = dev_upload(&t_q, d.quadrature_points, ncell * nd * P.dim * sizeof(T), tmp))) { cleanup(); return rc; }
    hipError_t e = hipMalloc((void **)&t_coef, ncell * nd * sizeof(T));
    if (e == hipSuccess) e = coefficient_launch<T>(t_coef, t_q, ncell * nd, P.dim, nullptr);
    if (e != hipSuccess) {
      set_error(std::string("coefficient evaluation: ") + hipGetErrorString(e));
      cleanup();
      return MFGPU_EHIP;
    }
  }
  if ((rc = dev_upload(&t_jxw, d.JxW, ncell * nd * sizeof(T), tmp))) { cleanup(); return rc; }
  const size_t jac_per_cell = h->gk ? nd * (size_t)(P.dim * P.dim) : 1;  // full J^-1 per point, or one scalar per cell
  if ((rc = dev_upload(&t_j0, d.inv_jac, ncell * jac_per_cell * sizeof(T), tmp))) { cleanup(); return rc; }
  if ((rc = dev_upload(&t_order, P.cell_order.data(), ncell * 4, tmp))) { cleanup(); return rc; }
  // symmetric M = a JxW J^-1 J^-T (6 entries in 3D, 3 in 2D), or the scalar a J0^2 JxW
  const size_t coef_per_point = h->gk ? (P.dim == 3 ? 6 : 3) : 1;
  hipError_t e = hipMalloc(&h->d_coef, ncell * nd * coef_per_point * sizeof(T));
  if (e == hipSuccess) {
    acct += ncell * nd * coef_per_point * sizeof(T);
    e = h->gk ? (P.dim == 3 ? fold_general_launch<T>((T *)h->d_coef, t_coef, t_jxw, t_j0, t_order, (uint32_t)ncell, (uint32_t)nd, nullptr)
                            : fold_general2_launch<T>((T *)h->d_coef, t_coef, t_jxw, t_j0, t_order, (uint32_t)ncell, (uint32_t)nd, nullptr))
              : fold_launch<T>((T *)h->d_coef, t_coef, t_jxw, t_j0, t_order, (uint32_t)ncell, (uint32_t)nd, nullptr);
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  cleanup();
  if (e != hipSuccess) {
    set_error(std::string("coefficient fold: ") + hipGetErrorString(e));
    return MFGPU_EHIP;
  }
  if (h->pk) {
    // the folded coefficient again, per batch [row y + n z][task]: the layout of stage B of apply_planes3
    // (d_coef in plan cell order stays: the diagonal kernel reads it)
    const int n = P.n, NT = p_cells_per_wave(n) * n;
    const size_t nbat = P.n_plane_batches, total = nbat * (size_t)(n * n) * NT;
    const size_t ncell_p = P.batch_cell_off[nbat];  // the plane batches' cells come first in plan order
    std::vector<uint32_t> cb(ncell_p), cp(ncell_p);
    for (size_t b = 0; b < nbat; ++b)
      for (uint32_t c = P.batch_cell_off[b]; c < P.batch_cell_off[b + 1]; ++c) {
        cb[c] = (uint32_t)b;
        cp[c] = c - P.batch_cell_off[b];
      }
    uint32_t *t_cb = nullptr, *t_cp = nullptr;
    size_t tmp2 = 0;
    rc = dev_upload(&t_cb, cb.data(), ncell_p * 4, tmp2);
    if (!rc) rc = dev_upload(&t_cp, cp.data(), ncell_p * 4, tmp2);
    if (!rc) {
      hipError_t e2 = hipMalloc(&h->d_coefp, total * sizeof(T));
      if (e2 == hipSuccess) {
        acct += total * sizeof(T);
        e2 = hipMemset(h->d_coefp, 0, total * sizeof(T));
      }
      if (e2 == hipSuccess)
        e2 = relayout_coef_launch<T>((T *)h->d_coefp, (const T *)h->d_coef, t_cb, t_cp, ncell_p * nd, n, nullptr);
      if (e2 == hipSuccess) e2 = hipDeviceSynchronize();
      if (e2 != hipSuccess) {
        set_error(std::string("coefficient relayout: ") + hipGetErrorString(e2));
        rc = e2 == hipErrorOutOfMemory ? MFGPU_ENOMEM : MFGPU_EHIP;
      }
    }
    hipFree(t_cb);
    hipFree(t_cp);
    if (rc) return rc;
    ApplyArgs<T> dummy{};
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    size_t lds_p = 0;
    HIP_TRY(planes_launch<T>(h, dummy, false, 0, nullptr, true, &lds_p, &per_cu));
    if (P.n_plain_plane_batches < P.n_plane_batches) {  // batches of masked cells: apply_planes3<HN>
      int per_cu_h = 0;
      size_t lds_h = 0;
      HIP_TRY(planes_launch<T>(h, dummy, true, 0, nullptr, true, &lds_h, &per_cu_h));
      hipDeviceProp_t prop_h;
      int dev_h = 0;
      HIP_TRY(hipGetDevice(&dev_h));
      HIP_TRY(hipGetDeviceProperties(&prop_h, dev_h));
      h->max_grid_ph = (uint32_t)(per_cu_h < 1 ? 1 : per_cu_h) * (uint32_t)prop_h.multiProcessorCount;
      if (d.max_workgroups && d.max_workgroups < h->max_grid_ph) h->max_grid_ph = d.max_workgroups;
      if (lds_h > lds_p) lds_p = lds_h;
    }
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    h->max_grid_p = (uint32_t)(per_cu < 1 ? 1 : per_cu) * (uint32_t)prop.multiProcessorCount;
    if (d.max_workgroups && d.max_workgroups < h->max_grid_p) h->max_grid_p = d.max_workgroups;
    if (!h->xk) {
      h->lds = lds_p;
      return 0;
    }
  }
  // persistent grid: as many workgroups as fit on the chip (each loops over its batches)
  ApplyArgs<T> dummy{};
  dummy.nb_max = P.max_batch_dofs;
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (h->gk && P.dim == 2) {
    std::vector<T> sd(2 * (size_t)P.n * P.n);
    for (int i = 0; i < P.n * P.n; ++i) {
      sd[i] = (T)h->S[i];
      sd[P.n * P.n + i] = (T)h->Dt[i];
    }
    if ((rc = dev_upload((T **)&h->d_tabsd, sd.data(), sd.size() * sizeof(T), acct))) return rc;
    HIP_TRY(g2_launch<T>(P.n, dummy, h->hn, 0, nullptr, true, &h->lds, nullptr));
  } else if (h->gk) {
    HIP_TRY(g_launch<T>(P.n, dummy, nullptr, nullptr, h->hn, 0, nullptr, true, &h->lds, nullptr));
  } else if (h->xk) {
    HIP_TRY(x_launch<T>(P.n, dummy, nullptr, nullptr, h->hn, 0, nullptr, true, &h->lds, nullptr));
  } else {
    h->lds = apply_lds_bytes<T>(P.dim, P.n, P.max_batch_dofs);
  }
  if (h->lds > 160 * 1024) {
    set_error("batch needs more than 160 KiB of LDS; lower max_dofs_per_batch");
    return MFGPU_EINVAL;
  }
  if (h->gk && P.dim == 2) {
    HIP_TRY(g2_launch<T>(P.n, dummy, h->hn, 0, nullptr, true, &h->lds, &per_cu));
  } else if (h->gk) {
    HIP_TRY(g_launch<T>(P.n, dummy, nullptr, nullptr, h->hn, 0, nullptr, true, &h->lds, &per_cu));
  } else if (h->xk) {
    HIP_TRY(x_launch<T>(P.n, dummy, nullptr, nullptr, h->hn, 0, nullptr, true, &h->lds, &per_cu));
  } else {
    HIP_TRY(apply_configure<T>(P.dim, P.n, h->lds));
    HIP_TRY(apply_occupancy<T>(P.dim, P.n, h->hn, h->twopass, h->lds, &per_cu));
  }
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  h->max_grid = (uint32_t)(per_cu < 1 ? 1 : per_cu) * (uint32_t)prop.multiProcessorCount;
  if (d.max_workgroups && d.max_workgroups < h->max_grid) h->max_grid = d.max_workgroups;
  return 0;
}

// Segments of the cell loop (see mfgpu_handle::seg_end).  request = mfgpu_desc.cell_loop_segments: 0 the library's
// choice, 1 one segment (pass 2 strictly after the cell loop), k > 1 k segments of equal batch counts.  The choice:
// where two kernel families share the mesh (plane batches | pencil batches of the cells with a hanging-node mask) the
// family boundary -- the launch boundary exists anyway (C3 in round 2: 0.381 instead of 0.392 ms per vmult); two halves
// on hanging-node meshes at p = 4 (below); ONE segment otherwise.  Measured on C2 (profiles/r02_notes.md section 6): with the last 4 of 13 grid iterations as a second
// segment the two kernels do run side by side, but the cell loop slows down by what pass 2 takes (58.6 instead of
// 37 us for the segment; both are short of issue slots and memory latency, not of different resources), and the event
// record / cross-stream waits add three pipeline drains of 5-12 us per vmult: 0.171 instead of 0.156 ms.
void choose_segments(mfgpu_handle *h, uint32_t request) {
  const Plan &P = h->plan;
  const uint32_t nb = (uint32_t)(P.batch_cell_off.size() - 1);
  h->seg_end.assign(1, nb);
  if (!h->twopass || nb < 2 || request == 1) return;
  const uint32_t npl = h->pk ? P.n_plane_batches : 0u, nplain = h->pk ? P.n_plain_plane_batches : 0u;
  std::vector<uint32_t> cuts;
  if (npl > 0 && npl < nb) cuts.push_back(npl);
  if (nplain > 0 && nplain < npl) cuts.push_back(nplain);  // plain plane batches | plane batches of masked cells
  if (request > 1) {
    for (uint32_t i = 1; i < request; ++i) cuts.push_back((uint32_t)((uint64_t)nb * i / request));
  }
  // hanging-node meshes at p = 4 (batches of masked and unmasked cells interleaved, one instantiation): pass 2 also
  // writes the identity rows of the eliminated hanging-node dofs and is a third of the cell loop's time; its first
  // half beside the second half of the cell loop measures 0.267-0.269 instead of 0.274-0.278 ms on C3 (three segments:
  // 0.280; at p = 3: no difference) -- profiles/r03_notes.md section 8
  if (request == 0 && h->pk && !P.pr_hn.empty() && nplain == 0 && P.n == 5) cuts.push_back(nb / 2);
  std::sort(cuts.begin(), cuts.end());
  h->seg_end.clear();
  for (uint32_t c : cuts)
    if (c > 0 && c < nb && (h->seg_end.empty() || h->seg_end.back() != c)) h->seg_end.push_back(c);
  h->seg_end.push_back(nb);
}

template <typename T>
int create_typed(mfgpu_handle *h, const mfgpu_desc &d) {
  int rc = create_arrays<T>(h, d);
  if (rc) return rc;
  choose_segments(h, d.cell_loop_segments);
  if (!h->twopass) return 0;
  if (h->seg_end.size() > 1) {
    // lowest priority: when a segment ends, the next segment's workgroups should be placed before the pass-2 waves,
    // which fill the remaining wave slots
    int prio_least = 0, prio_greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    HIP_TRY(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, prio_least));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming));
    h->ev_seg.assign(h->seg_end.size() - 1, nullptr);
    for (hipEvent_t &e : h->ev_seg) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  return upload_pass2(h, nullptr, 0);
}

// one vmult in three phases: the cell loop (with pass 2 of the earlier segments' dofs on the side stream); pass 2 of
// the priority dofs; pass 2 of the last segment's dofs and the join with the side stream.  mfgpu_vmult runs them back
// to back; mfgpu_vmult_dist_begin starts the exchange of the slab's interface planes between the last two.
template <typename T>
int launch_pass2_group(mfgpu_handle *h, size_t group, void *dst, const void *src, hipStream_t st, int add) {
  HIP_TRY(reduce_classes_launch<T>((T *)dst, (const T *)src, (const T *)h->d_halo, h->d_p2arr[group], h->d_p2tiles[group],
                                   h->n_p2tiles[group], add, st));
  return 0;
}

template <typename T>
int vmult_pass2(mfgpu_handle *h, int phase, void *dst, const void *src, hipStream_t st, int add) {
  if (!h->twopass) return 0;
  if (phase == 0) return launch_pass2_group<T>(h, 0, dst, src, st, add);
  int rc = launch_pass2_group<T>(h, h->seg_end.size(), dst, src, st, add);
  if (!rc && h->side_pending) {
    HIP_TRY(hipStreamWaitEvent(st, h->ev_side, 0));
    h->side_pending = false;
  }
  return rc;
}

// the plane kernel of this handle
template <typename T>
hipError_t planes_launch(mfgpu_handle *h, const ApplyArgs<T> &a0, bool hn, uint32_t grid, hipStream_t st,
                         bool configure_only, size_t *lds_out, int *occupancy) {
  const ApplyArgs<T> &a = a0;
  return h->qk ? q_launch<T>(h->plan.n, a, h->S.data(), h->Dt.data(), hn, grid, st, configure_only, lds_out, occupancy)
               : p_launch<T>(h->plan.n, a, h->S.data(), h->Dt.data(), hn, grid, st, configure_only, lds_out, occupancy);
}

// batches [b0, b1) of one scatter pass, each with the kernel family that owns it
template <typename T>
int launch_cells(mfgpu_handle *h, ApplyArgs<T> a, uint32_t b0, uint32_t b1, hipStream_t st) {
  const Plan &P = h->plan;
  const uint32_t npl = h->pk ? P.n_plane_batches : 0u;
  const uint32_t nplain = h->pk ? P.n_plain_plane_batches : 0u;
  if (b0 < nplain) {  // the batches of cells without a hanging-node mask (all batches on conforming meshes)
    a.batch0 = b0;
    a.batch_end = b1 < nplain ? b1 : nplain;
    const uint32_t nbat = a.batch_end - a.batch0;
    HIP_TRY(planes_launch<T>(h, a, false, nbat < h->max_grid_p ? nbat : h->max_grid_p, st, false, nullptr, nullptr));
    b0 = a.batch_end;
  }
  if (b0 < npl && b0 < b1) {  // plane batches of cells WITH a mask
    a.batch0 = b0;
    a.batch_end = b1 < npl ? b1 : npl;
    const uint32_t nbat = a.batch_end - a.batch0;
    HIP_TRY(planes_launch<T>(h, a, true, nbat < h->max_grid_ph ? nbat : h->max_grid_ph, st, false, nullptr, nullptr));
    b0 = a.batch_end;
  }
  if (b0 >= b1) return 0;
  a.batch0 = b0;
  a.batch_end = b1;
  const uint32_t nrest = b1 - b0, grid = nrest < h->max_grid ? nrest : h->max_grid;
  if (h->gk && P.dim == 2)
    HIP_TRY(g2_launch<T>(P.n, a, h->hn, grid, st, false, nullptr, nullptr));
  else if (h->gk)
    HIP_TRY(g_launch<T>(P.n, a, h->S.data(), h->Dt.data(), h->hn, grid, st, false, nullptr, nullptr));
  else if (h->xk)
    HIP_TRY(x_launch<T>(P.n, a, h->S.data(), h->Dt.data(), h->hn, grid, st, false, nullptr, nullptr));
  else
    HIP_TRY(apply_launch<T>(P.dim, P.n, a, h->S.data(), h->Dt.data(), h->hn, h->twopass, grid, st));
  return 0;
}

template <typename T>
ApplyArgs<T> make_args(mfgpu_handle *h, void *dst, const void *src, int add) {
  const Plan &P = h->plan;
  ApplyArgs<T> a;
  a.batch_cell_off = h->d_batch_cell_off;
  a.batch_dof_off = h->d_batch_dof_off;
  a.bdofs = h->d_bdofs;
  a.bflags = h->d_bflags;
  a.lmap = h->d_lmap;
  a.lmapx = h->d_lmapx;
  a.perm = h->d_perm;
  a.bdofsp = h->d_bdofsp;
  a.idxp = h->d_idxp;
  a.hnrec = h->d_hnrec;
  a.hn_slot = h->d_hn_slot;
  a.coefp = (const T *)h->d_coefp;
  a.coef = (const T *)h->d_coef;
  a.cmask = h->d_cmask;
  a.hn_weights = (const T *)h->d_hnw;
  a.tabS = (const T *)h->d_tabsd;
  a.tabDt = h->d_tabsd ? (const T *)h->d_tabsd + (size_t)P.n * P.n : nullptr;
  a.batch_nint = h->d_batch_nint;
  a.halo_off = h->d_halo_off;
  a.halo = (T *)h->d_halo;
  a.dst = (T *)dst;
  a.src = (const T *)src;
  a.nb_max = P.max_batch_dofs;
  a.add = add;
  a.stamps = h->d_stamps;
  a.dbg = 0;
  return a;
}

template <typename T>
int vmult_main(mfgpu_handle *h, void *dst, const void *src, hipStream_t st, int add) {
  const Plan &P = h->plan;
  ApplyArgs<T> a = make_args<T>(h, dst, src, add);
  if (h->prof) {
    if (h->ev_used + 2 > h->ev.size()) {
      hipEvent_t e0, e1;
      HIP_TRY(hipEventCreate(&e0));
      HIP_TRY(hipEventCreate(&e1));
      h->ev.push_back(e0);
      h->ev.push_back(e1);
    }
    HIP_TRY(hipEventRecord(h->ev[h->ev_used], st));
  }
  if (h->twopass) {
    // ONE sweep over all batches (no inter-batch dependency) in segments; after every segment but the last, the
    // shared-dof sums that are complete by then start on the side stream
    const size_t nseg = h->seg_end.size();
    if (h->side_pending) {  // (a caller that ran phase 0 twice without the closing phase)
      HIP_TRY(hipStreamWaitEvent(st, h->ev_side, 0));
      h->side_pending = false;
    }
    for (size_t s = 0; s < nseg; ++s) {
      int rc = launch_cells<T>(h, a, s ? h->seg_end[s - 1] : 0u, h->seg_end[s], st);
      if (rc) return rc;
      if (s + 1 < nseg) {
        HIP_TRY(hipEventRecord(h->ev_seg[s], st));
        HIP_TRY(hipStreamWaitEvent(h->side, h->ev_seg[s], 0));
        if ((rc = launch_pass2_group<T>(h, 1 + s, dst, src, h->side, add))) return rc;
      }
    }
    if (nseg > 1) {
      HIP_TRY(hipEventRecord(h->ev_side, h->side));
      h->side_pending = true;
    }
  } else {
    // coloured mode: one launch per batch colour (first toucher stores, later colours add)
    for (size_t c = 0; c + 1 < P.color_batch_off.size(); ++c) {
      if (P.color_batch_off[c + 1] == P.color_batch_off[c]) continue;
      int rc = launch_cells<T>(h, a, P.color_batch_off[c], P.color_batch_off[c + 1], st);
      if (rc) return rc;
    }
  }
  if (h->prof) {
    HIP_TRY(hipEventRecord(h->ev[h->ev_used + 1], st));
    h->ev_used += 2;
  }
  if (!h->twopass)  // coloured mode has no pass 2: the dofs no cell touches get their own small kernel
    HIP_TRY(orphan_launch<T>((T *)dst, (const T *)src, h->d_orphans, (uint32_t)P.orphans.size(), add, st));
  if (h->prof) h->prof_vmults++;
  return 0;
}

template <typename T>
int vmult_typed(mfgpu_handle *h, void *dst, const void *src, hipStream_t st, int add) {
  int rc = vmult_main<T>(h, dst, src, st, add);
  if (rc) return rc;
  if (h->prof && h->twopass) {
    if (h->ev2_used + 2 > h->ev2.size()) {
      hipEvent_t e0, e1;
      HIP_TRY(hipEventCreate(&e0));
      HIP_TRY(hipEventCreate(&e1));
      h->ev2.push_back(e0);
      h->ev2.push_back(e1);
    }
    HIP_TRY(hipEventRecord(h->ev2[h->ev2_used], st));
  }
  rc = vmult_pass2<T>(h, 0, dst, src, st, add);
  if (!rc) rc = vmult_pass2<T>(h, 1, dst, src, st, add);
  if (!rc && h->prof && h->twopass) {
    HIP_TRY(hipEventRecord(h->ev2[h->ev2_used + 1], st));
    h->ev2_used += 2;
  }
  return rc;
}

}  // namespace

// ---- diagonal, set_constrained_values
namespace {
template <typename T>
int inverse_diagonal_typed(mfgpu_handle *h, void *diag, hipStream_t st) {
  const Plan &P = h->plan;
  if (!h->d_tab2) {  // 1D tables T[2][n*n]: squared [S.^2 | G.^2], or plain [S | G] for the general-geometry path
    const int nn = h->n * h->n;
    std::vector<T> t2(2 * (size_t)nn);
    for (int i = 0; i < nn; ++i) {
      t2[i] = (T)(h->gk ? h->sv[i] : h->sv[i] * h->sv[i]);
      t2[nn + i] = (T)(h->gk ? h->sg[i] : h->sg[i] * h->sg[i]);
    }
    int rc = dev_upload(&h->d_tab2, t2.data(), t2.size() * sizeof(T), h->device_bytes);
    if (rc) return rc;
  }
  // inv_diag.reinit(m()): zero  (laplace_operator_gpu.h:407)
  HIP_TRY(fill_launch<T>((T *)diag, P.n_dofs, T(0), st));
  // data.cell_loop(inv_diag, diag_loc_op)  (:409-410)
  if (h->gk && P.dim == 2)
    HIP_TRY(diag_general2_launch<T>(P.n, (T *)diag, (uint32_t)(P.batch_cell_off.size() - 1), h->d_batch_cell_off,
                                    h->d_batch_dof_off, h->d_bdofs, h->d_lmap, (const T *)h->d_coef, h->d_cmask,
                                    (const T *)h->d_hnw, (const T *)h->d_tab2, st));
  else if (h->gk)
    HIP_TRY(diag_general_launch<T>(P.n, (T *)diag, (uint32_t)(P.batch_cell_off.size() - 1), h->d_batch_cell_off,
                                   h->d_batch_dof_off, h->d_bdofs, h->d_lmap, (const T *)h->d_coef, h->d_cmask,
                                   (const T *)h->d_hnw, (const T *)h->d_tab2, st));
  else
    HIP_TRY(diag_launch<T>(P.dim, P.n, (T *)diag, (uint32_t)(P.batch_cell_off.size() - 1), h->d_batch_cell_off,
                           h->d_batch_dof_off, h->d_bdofs, h->d_lmap, (const T *)h->d_coef, h->d_cmask,
                           (const T *)h->d_hnw, (const T *)h->d_tab2, st));
  // constraint_handler.set_constrained_values(inv_diag, 1.0)  (:412)
  HIP_TRY(set_values_launch<T>((T *)diag, h->d_constrained, h->n_constrained, T(1), st));
  // inv_diag.invert()  (:414)
  HIP_TRY(vec_map_launch<T>(4, (T *)diag, nullptr, T(0), T(0), P.n_dofs, st));
  return 0;
}
}  // namespace


// ---- GpuVector BLAS-1 and reductions
namespace {
int vec_map(int op, void *v, const void *w, double s, double a, size_t n, int nt, void *stream) {
  if ((!v && n) || (!w && n && op <= 3)) {
    set_error("null vector");
    return MFGPU_EINVAL;
  }
  if (nt == MFGPU_F32)
    HIP_TRY(vec_map_launch<float>(op, (float *)v, (const float *)w, (float)s, (float)a, n, (hipStream_t)stream));
  else if (nt == MFGPU_F64)
    HIP_TRY(vec_map_launch<double>(op, (double *)v, (const double *)w, s, a, n, (hipStream_t)stream));
  else
    return MFGPU_EINVAL;
  return 0;
}
int vec_reduce(int op, void *v, const void *x, const void *w, double a, size_t n, int nt, void *stream, double *out) {
  if (!out || (n && (!v || (op != 2 && !w) || (op == 1 && !x)))) {
    set_error("null argument");
    return MFGPU_EINVAL;
  }
  if (nt == MFGPU_F32)
    HIP_TRY(vec_reduce_launch<float>(op, (float *)v, (const float *)x, (const float *)w, (float)a, n, (hipStream_t)stream, out));
  else if (nt == MFGPU_F64)
    HIP_TRY(vec_reduce_launch<double>(op, (double *)v, (const double *)x, (const double *)w, a, n, (hipStream_t)stream, out));
  else
    return MFGPU_EINVAL;
  return 0;
}
}  // namespace


// ---- entry points of mfgpu_dist.hip (multi-GPU slab exchange) into the operator
namespace mfgpu {
int handle_number_type(const mfgpu_handle *h) { return h->number_type; }
int handle_set_priority_dofs(mfgpu_handle *h, const uint32_t *ids, uint32_t n) {
  return h->twopass ? upload_pass2(h, ids, n) : 0;
}
int handle_n_batches(const mfgpu_handle *h) { return (int)h->plan.batch_cell_off.size() - 1; }
bool handle_ranged_ok(const mfgpu_handle *h) { return h->twopass && h->seg_end.size() == 1; }
int handle_batches_touching(const mfgpu_handle *h, const uint32_t *ids, uint32_t n, std::vector<uint8_t> &flags) {
  const Plan &P = h->plan;
  std::vector<uint8_t> mark(P.n_dofs, 0);
  for (uint32_t i = 0; i < n; ++i) {
    if (ids[i] >= P.n_dofs) {
      set_error("interface dof out of range");
      return MFGPU_EINVAL;
    }
    mark[ids[i]] = 1;
  }
  const size_t nb = P.batch_cell_off.size() - 1;
  flags.assign(nb, 0);
  for (size_t b = 0; b < nb; ++b)
    for (uint32_t t = P.batch_dof_off[b]; t < P.batch_dof_off[b + 1]; ++t)
      if (mark[P.bdofs[t] & 0x7fffffffu]) {
        flags[b] = 1;
        break;
      }
  return 0;
}
int handle_cells_range(mfgpu_handle *h, uint32_t b0, uint32_t b1, void *dst, const void *src, void *stream, int add) {
  if (b0 >= b1) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (h->number_type == MFGPU_F64) return launch_cells<double>(h, make_args<double>(h, dst, src, add), b0, b1, st);
  return launch_cells<float>(h, make_args<float>(h, dst, src, add), b0, b1, st);
}
// batches [b0, b1) and [c0, c1), b1 <= c0: one launch with a hole when both lie in the plain plane batches
int handle_cells_two_ranges(mfgpu_handle *h, uint32_t b0, uint32_t b1, uint32_t c0, uint32_t c1, void *dst,
                            const void *src, void *stream, int add) {
  const uint32_t nplain = h->pk ? h->plan.n_plain_plane_batches : 0u, npl = h->pk ? h->plan.n_plane_batches : 0u;
  const bool plain = c1 <= nplain, masked = b0 >= nplain && c1 <= npl;  // both ranges in one instantiation's batches
  if (b0 >= b1 || c0 >= c1 || b1 > c0 || !(plain || masked)) {
    const int rc = handle_cells_range(h, b0, b1, dst, src, stream, add);
    return rc ? rc : handle_cells_range(h, c0, c1, dst, src, stream, add);
  }
  hipStream_t st = (hipStream_t)stream;
  auto run = [&](auto a) -> int {
    a.batch0 = b0;
    a.batch_end = c1;
    a.hole0 = b1;
    a.hole_len = c0 - b1;
    const uint32_t nbat = (b1 - b0) + (c1 - c0);
    using T = typename std::remove_const<typename std::remove_pointer<decltype(a.src)>::type>::type;
    const uint32_t cap = plain ? h->max_grid_p : h->max_grid_ph;
    HIP_TRY(planes_launch<T>(h, a, !plain, nbat < cap ? nbat : cap, st, false, nullptr, nullptr));
    return 0;
  };
  return h->number_type == MFGPU_F64 ? run(make_args<double>(h, dst, src, add)) : run(make_args<float>(h, dst, src, add));
}
int handle_pass2_group(mfgpu_handle *h, int group, void *dst, const void *src, void *stream, int add) {
  if (!h->twopass) return 0;
  hipStream_t st = (hipStream_t)stream;
  return h->number_type == MFGPU_F64 ? launch_pass2_group<double>(h, (size_t)group, dst, src, st, add)
                                     : launch_pass2_group<float>(h, (size_t)group, dst, src, st, add);
}
int handle_vmult_phase(mfgpu_handle *h, int phase, void *dst, const void *src, void *stream, int add) {
  hipStream_t st = (hipStream_t)stream;
  if (h->number_type == MFGPU_F64)
    return phase == 0 ? vmult_main<double>(h, dst, src, st, add) : vmult_pass2<double>(h, phase - 1, dst, src, st, add);
  return phase == 0 ? vmult_main<float>(h, dst, src, st, add) : vmult_pass2<float>(h, phase - 1, dst, src, st, add);
}
}  // namespace mfgpu

extern "C" {

int mfgpu_create(const mfgpu_desc *desc, mfgpu_handle **out) {
  if (!desc || !out) {
    set_error("null argument");
    return MFGPU_EINVAL;
  }
  const mfgpu_desc &d = *desc;
  if (d.number_type != MFGPU_F64 && d.number_type != MFGPU_F32) {
    set_error("number_type must be MFGPU_F64 or MFGPU_F32");
    return MFGPU_EINVAL;
  }
  const bool general = !(d.flags & MFGPU_UNIFORM_J0);
  if (general && (d.flags & MFGPU_COLORED_SCATTER)) {
    set_error("the general-Jacobian path (no MFGPU_UNIFORM_J0) is implemented in two-pass scatter mode only");
    return MFGPU_EUNSUPPORTED;
  }
  if (!d.JxW || !d.inv_jac || !d.shape_values || !d.shape_gradients ||
      (!d.coefficient && !d.quadrature_points)) {
    set_error("JxW, inv_jac, shape tables and coefficient (or quadrature_points) are required");
    return MFGPU_EINVAL;
  }
  const bool hn = (d.flags & MFGPU_HANGING_NODES) != 0;
  if (hn && (!d.constraint_mask || !d.constraint_weights)) {
    set_error("MFGPU_HANGING_NODES needs constraint_mask and constraint_weights");
    return MFGPU_EINVAL;
  }
  mfgpu_handle *h = new mfgpu_handle();
  KernelChoice kc;
  int rc = choose_kernel_and_plan(d, kc, h->plan);
  h->gk = kc.general;
  h->pk = kc.planes;
  // which plane kernel: apply_planes4 on request, at p = 5, 6 (the only one that fits), and by default at p = 3 in
  // double; apply_planes3 otherwise (p = 4: equal in double, faster in float)
  h->qk = d.kernel == MFGPU_KERNEL_PLANES_2W || d.degree >= 5 ||
          (d.kernel == MFGPU_KERNEL_AUTO && d.degree == 3 && d.number_type == MFGPU_F64);
  h->xk = kc.pencils_x;
  if (rc) {
    delete h;
    return rc;
  }
  h->dim = d.dim;
  h->n = d.degree + 1;
  h->nd = h->plan.nd;
  h->number_type = d.number_type;
  h->hn = hn;
  h->twopass = !(d.flags & MFGPU_COLORED_SCATTER);
  const int nn = h->n * h->n;
  std::vector<double> sv(nn), sg(nn);
  for (int i = 0; i < nn; ++i) {
    sv[i] = d.number_type == MFGPU_F64 ? ((const double *)d.shape_values)[i] : ((const float *)d.shape_values)[i];
    sg[i] = d.number_type == MFGPU_F64 ? ((const double *)d.shape_gradients)[i] : ((const float *)d.shape_gradients)[i];
  }
  h->sv = sv;
  h->sg = sg;
  rc = derive_tables(h->n, sv.data(), sg.data(), h->S, h->Dt);
  if (!rc) rc = check_symmetrize(h->n, h->S, h->Dt);
  if (!rc) rc = d.number_type == MFGPU_F64 ? create_typed<double>(h, d) : create_typed<float>(h, d);
  if (!rc && d.n_constrained) {
    h->n_constrained = d.n_constrained;
    rc = dev_upload(&h->d_constrained, d.constrained_dofs, (size_t)d.n_constrained * 4, h->device_bytes);
  }
  if (rc) {
    mfgpu_destroy(h);
    return rc;
  }
  *out = h;
  return 0;
}

void mfgpu_destroy(mfgpu_handle *h) {
  if (!h) return;
  hipFree(h->d_batch_cell_off);
  hipFree(h->d_batch_dof_off);
  hipFree(h->d_bdofs);
  hipFree(h->d_bflags);
  hipFree(h->d_lmap);
  hipFree(h->d_lmapx);
  hipFree(h->d_perm);
  hipFree(h->d_bdofsp);
  hipFree(h->d_idxp);
  hipFree(h->d_hnrec);
  hipFree(h->d_hn_slot);
  hipFree(h->d_coefp);
  hipFree(h->d_constrained);
  hipFree(h->d_tab2);
  hipFree(h->d_tabsd);
  hipFree(h->d_coef);
  hipFree(h->d_cmask);
  hipFree(h->d_orphans);
  hipFree(h->d_hnw);
  hipFree(h->d_batch_nint);
  hipFree(h->d_halo_off);
  for (size_t g = 0; g < h->d_p2arr.size(); ++g) {
    hipFree(h->d_p2arr[g]);
    hipFree(h->d_p2tiles[g]);
  }
  if (h->side) {
    hipStreamSynchronize(h->side);
    hipStreamDestroy(h->side);
  }
  for (hipEvent_t e : h->ev_seg)
    if (e) hipEventDestroy(e);
  if (h->ev_side) hipEventDestroy(h->ev_side);
  hipFree(h->d_halo);
  hipFree(h->d_stamps);
  for (hipEvent_t e : h->ev) hipEventDestroy(e);
  for (hipEvent_t e : h->ev2) hipEventDestroy(e);
  delete h;
}

int mfgpu_vmult(mfgpu_handle *h, void *dst, const void *src, void *stream) {
  if (!h || !dst || !src) {
    set_error("null argument");
    return MFGPU_EINVAL;
  }
  if (dst == src) {
    set_error("vmult: dst and src must not alias");
    return MFGPU_EINVAL;
  }
  return h->number_type == MFGPU_F64 ? vmult_typed<double>(h, dst, src, (hipStream_t)stream, 0)
                                     : vmult_typed<float>(h, dst, src, (hipStream_t)stream, 0);
}

int mfgpu_vmult_add(mfgpu_handle *h, void *dst, const void *src, void *stream) {
  if (!h || !dst || !src) {
    set_error("null argument");
    return MFGPU_EINVAL;
  }
  if (dst == src) {
    set_error("vmult_add: dst and src must not alias");
    return MFGPU_EINVAL;
  }
  return h->number_type == MFGPU_F64 ? vmult_typed<double>(h, dst, src, (hipStream_t)stream, 1)
                                     : vmult_typed<float>(h, dst, src, (hipStream_t)stream, 1);
}

uint32_t mfgpu_n_dofs(const mfgpu_handle *h) { return h ? h->plan.n_dofs : 0; }

size_t mfgpu_memory_consumption(const mfgpu_handle *h) { return h ? h->device_bytes : 0; }

int mfgpu_plan_stats(const mfgpu_handle *h, uint64_t s[8]) {
  if (!h || !s) return MFGPU_EINVAL;
  const Plan &P = h->plan;
  s[0] = P.batch_cell_off.size() - 1;
  s[1] = P.color_batch_off.size() - 1;
  s[2] = P.bdofs.size();
  s[3] = P.max_batch_dofs;
  s[4] = P.max_batch_cells;
  s[5] = P.orphans.size();
  s[6] = P.n_first;
  s[7] = P.n_add;
  if (h->twopass) {  // two-pass mode reports shared dofs / halo partial sums instead
    // cell-loop launches per vmult: per segment one for each kernel instantiation that owns batches of it
    const uint32_t nb = (uint32_t)s[0], npl = h->pk ? P.n_plane_batches : 0u, nplain = h->pk ? P.n_plain_plane_batches : 0u;
    const uint32_t edge[4] = {0u, nplain, npl, nb};
    s[1] = 0;
    for (size_t g = 0; g < h->seg_end.size(); ++g) {
      const uint32_t s0 = g ? h->seg_end[g - 1] : 0u, s1 = h->seg_end[g];
      for (int f = 0; f < 3; ++f)
        if (std::max(s0, edge[f]) < std::min(s1, edge[f + 1])) ++s[1];
    }
    s[6] = P.sdofs.size();
    s[7] = P.halo_off.back();
  }
  return 0;
}

const char *mfgpu_kernel_name(const mfgpu_handle *h) {
  if (!h) return "";
  if (h->pk && h->qk) return h->xk ? "apply_planes4+apply_batches_x" : "apply_planes4";
  return h->pk ? (h->xk ? "apply_planes3+apply_batches_x" : "apply_planes3")
               : h->gk ? (h->dim == 2 ? "apply_batches_g2" : "apply_batches_g") : h->xk ? "apply_batches_x" : "apply_batches";
}

int mfgpu_profile_enable(mfgpu_handle *h, int on) {
  if (!h) return MFGPU_EINVAL;
  h->prof = on != 0;
  h->ev_used = 0;
  h->prof_ms = 0.0;
  h->prof_vmults = 0;
  h->ev2_used = 0;
  h->prof2_ms = 0.0;
  return 0;
}

int mfgpu_profile_read_pass2(mfgpu_handle *h, double *ms) {
  if (!h || !ms) return MFGPU_EINVAL;
  HIP_TRY(hipDeviceSynchronize());
  for (size_t i = 0; i + 1 < h->ev2_used; i += 2) {
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, h->ev2[i], h->ev2[i + 1]));
    h->prof2_ms += t;
  }
  h->ev2_used = 0;
  *ms = h->prof2_ms;
  return 0;
}

int mfgpu_profile_read(mfgpu_handle *h, double *ms, uint64_t *nv) {
  if (!h || !ms || !nv) return MFGPU_EINVAL;
  HIP_TRY(hipDeviceSynchronize());
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, h->ev[i], h->ev[i + 1]));
    h->prof_ms += t;
  }
  h->ev_used = 0;
  *ms = h->prof_ms;
  *nv = h->prof_vmults;
  return 0;
}

#ifdef MFGPU_STAMPS
// diagnostic build only: allocate / read the per-workgroup phase stamps (16 u64 per batch)
int mfgpu_debug_stamps(mfgpu_handle *h, unsigned long long *out, size_t n_batches) {
  if (!h) return MFGPU_EINVAL;
  const size_t nbt = h->plan.batch_cell_off.size() - 1;
  if (!h->d_stamps) {
    HIP_TRY(hipMalloc((void **)&h->d_stamps, 2 * nbt * 16 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(h->d_stamps, 0, 2 * nbt * 16 * sizeof(unsigned long long)));
    return 0;
  }
  if (out && n_batches == nbt) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, h->d_stamps, 2 * nbt * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  }
  return 0;
}
#endif

// ---- GpuVector pieces -----------------------------------------------------------------------

static size_t esize(int nt) { return nt == MFGPU_F32 ? 4 : 8; }

// ---- diagonal, set_constrained_values
int mfgpu_compute_inverse_diagonal(mfgpu_handle *h, void *inv_diag, void *stream) {
  if (!h || !inv_diag) {
    set_error("null argument");
    return MFGPU_EINVAL;
  }
  return h->number_type == MFGPU_F64 ? inverse_diagonal_typed<double>(h, inv_diag, (hipStream_t)stream)
                                     : inverse_diagonal_typed<float>(h, inv_diag, (hipStream_t)stream);
}

int mfgpu_set_constrained_values(mfgpu_handle *h, void *vec, double value, void *stream) {
  if (!h || !vec) {
    set_error("null argument");
    return MFGPU_EINVAL;
  }
  if (h->number_type == MFGPU_F64)
    HIP_TRY(set_values_launch<double>((double *)vec, h->d_constrained, h->n_constrained, value, (hipStream_t)stream));
  else
    HIP_TRY(set_values_launch<float>((float *)vec, h->d_constrained, h->n_constrained, (float)value, (hipStream_t)stream));
  return 0;
}

// ---- GpuVector BLAS-1 and reductions
int mfgpu_vec_sadd(void *v, double s, double a, const void *w, size_t n, int nt, void *stream) {
  return vec_map(0, v, w, s, a, n, nt, stream);
}
int mfgpu_vec_equ(void *v, double a, const void *w, size_t n, int nt, void *stream) {
  return vec_map(1, v, w, 0.0, a, n, nt, stream);
}
int mfgpu_vec_scale(void *v, const void *w, size_t n, int nt, void *stream) {
  return vec_map(2, v, w, 0.0, 0.0, n, nt, stream);
}
int mfgpu_vec_divide(void *v, const void *w, size_t n, int nt, void *stream) {
  return vec_map(3, v, w, 0.0, 0.0, n, nt, stream);
}
int mfgpu_vec_invert(void *v, size_t n, int nt, void *stream) { return vec_map(4, v, nullptr, 0.0, 0.0, n, nt, stream); }
int mfgpu_vec_mul(void *v, double a, size_t n, int nt, void *stream) {
  return vec_map(5, v, nullptr, 0.0, a, n, nt, stream);
}
int mfgpu_vec_dot(const void *v, const void *w, size_t n, int nt, void *stream, double *result) {
  return vec_reduce(0, const_cast<void *>(v), nullptr, w, 0.0, n, nt, stream, result);
}
int mfgpu_vec_l2_norm(const void *v, size_t n, int nt, void *stream, double *result) {
  int rc = vec_reduce(0, const_cast<void *>(v), nullptr, v, 0.0, n, nt, stream, result);
  if (!rc) *result = std::sqrt(*result);
  return rc;
}
int mfgpu_vec_add_and_dot(void *v, double a, const void *x, const void *w, size_t n, int nt, void *stream,
                          double *result) {
  return vec_reduce(1, v, x, w, a, n, nt, stream, result);
}
int mfgpu_vec_all_zero(const void *v, size_t n, int nt, void *stream, int *result) {
  if (!result) return MFGPU_EINVAL;
  double cnt = 0.0;
  int rc = vec_reduce(2, const_cast<void *>(v), nullptr, nullptr, 0.0, n, nt, stream, &cnt);
  if (!rc) *result = cnt == 0.0;
  return rc;
}

int mfgpu_vec_alloc(void **dev, size_t n, int nt) {
  if (!dev) return MFGPU_EINVAL;
  *dev = nullptr;
  if (n == 0) return 0;
  HIP_TRY(hipMalloc(dev, n * esize(nt)));
  HIP_TRY(hipMemset(*dev, 0, n * esize(nt)));
  return 0;
}
int mfgpu_vec_free(void *dev) {
  if (dev) HIP_TRY(hipFree(dev));
  return 0;
}
int mfgpu_vec_fill(void *dev, size_t n, int nt, double value, void *stream) {
  if (!dev && n) return MFGPU_EINVAL;
  if (nt == MFGPU_F32)
    HIP_TRY(fill_launch<float>((float *)dev, n, (float)value, (hipStream_t)stream));
  else
    HIP_TRY(fill_launch<double>((double *)dev, n, value, (hipStream_t)stream));
  return 0;
}
int mfgpu_vec_from_host(void *dev, const void *host, size_t n, int nt) {
  if (n) HIP_TRY(hipMemcpy(dev, host, n * esize(nt), hipMemcpyHostToDevice));
  return 0;
}
int mfgpu_vec_to_host(void *host, const void *dev, size_t n, int nt) {
  if (n) HIP_TRY(hipMemcpy(host, dev, n * esize(nt), hipMemcpyDeviceToHost));
  return 0;
}
int mfgpu_device_synchronize(void) {
  HIP_TRY(hipDeviceSynchronize());
  return 0;
}
int mfgpu_device_memory_info(size_t *free_bytes, size_t *total_bytes) {
  if (!free_bytes || !total_bytes) return MFGPU_EINVAL;
  HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
  return 0;
}

}  // extern "C"
