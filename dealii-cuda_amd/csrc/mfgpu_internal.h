// Internal structures shared by the planner (host only), the API layer and the kernels.
#ifndef MFGPU_INTERNAL_H
#define MFGPU_INTERNAL_H

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mfgpu.h"

namespace mfgpu {

void set_error(const std::string &msg);

inline int ipow(int a, int e) {
  int r = 1;
  for (int i = 0; i < e; ++i) r *= a;
  return r;
}

// ---- apply_planes3 (mfgpu_kernels_p.hip): a wave owns 64 / n cells (n lanes per cell).  A batch's dof list has a
// FIXED structure of 64-lane slots: p_ji(n) slots of interior dofs (stored to dst by the cell loop) followed by
// p_hs(n) slots of pass-2 dofs (partial sums to the halo buffer), sized for the most compact batch of 64 / n cells
// (3x2x2 cells at p = 4: 539 interior, 514 surface dofs).  Cells of the LDS transpose arrays are p_cell_stride(n)
// values apart (stride = n mod 32: the 16-lane store groups and 32-lane load groups of the yz-plane stage hit
// distinct banks).
constexpr int p_cells_per_wave(int n) { return 64 / n; }
// (n = 6, 7, apply_planes4 only: 10 / 9 cells per wave; sized for 3x3x1 slabs and 2x2x2 cubes of cells)
constexpr int p_ji(int n) { return n == 3 ? 2 : n == 4 ? 5 : n == 5 ? 9 : n == 6 ? 14 : 23; }
constexpr int p_hs(int n) { return n == 3 ? 3 : n == 4 ? 6 : n == 5 ? 9 : n == 6 ? 14 : 18; }
constexpr int p_kgu(int n) { return p_ji(n) + p_hs(n); }
constexpr int p_cell_stride(int n) {
  int s = n * n * n;
  while ((s & 31) != (n & 31)) ++s;
  return s;
}

// apply_planes4 (mfgpu_kernels_q.hip): ONE transpose array, aliased with the batch array.  Element (x, y, z) of cell c
// sits at c * q_cell_stride + x + n * y + q_plane_stride * z.  A thread of the xy-layout (lane = n c + z) accesses
// c * SA + ZS * z + const, one of the yz-layout (lane = n c + x) c * SA + x + const: with SA = n and ZS = 1 (mod 32) both
// are lane + const, i.e. distinct banks in every 32-lane group of a ds_read_b64 and every 16-lane group of a
// ds_write_b64 (MI355X_MICROARCH.md, LDS table).  (p = 2: the smallest such strides are 99 / 33.)
constexpr int q_plane_stride(int n) { return n >= 6 ? 65 : n == 5 ? 33 : n == 4 ? 17 : 33; }
constexpr int q_cell_stride(int n) { return n * q_plane_stride(n); }
// apply_planes3 on cells WITH a hanging-node mask: the constrained nodes of a cell get PRIVATE entries behind the
// batch's dof list (p_priv_max(n) of them per batch), a copy of the gathered values, on which the 1D interpolation
// passes of resolve_hanging_nodes (hanging_nodes.cuh:617-696) run line by line before the cell stages, and their
// transposes after them (mfgpu_plan.cpp hn_cell_lines, mfgpu_kernels_p.hip)
constexpr int p_priv_max(int n) { return n >= 5 ? 384 : n == 4 ? 256 : 192; }
// ... and its fixed-size per-batch record of 32-bit words, stored [row][lane] (a wave reads a row with one coalesced
// load, one batch ahead): p_priv_max / 64 rows of copies (private position << 16 | dof-list position), then per
// direction (x, y, z) kHnOpRounds rounds of line operations, 3 rows each (the n <= 5 private positions of lane's line,
// 16 bit each), then 2 rows with the counts (copies | x lines << 16; y lines | z lines << 16) in every lane
constexpr int kHnOpRounds = 2;
constexpr int p_hn_rows(int n) { return p_priv_max(n) / 64 + 3 * kHnOpRounds * 3 + 2; }

// LDS of apply_planes4: max(transpose array incl. the idle lanes' scratch cell, batch array incl. private entries) --
// aliased --, and behind it the hanging-node weight matrix
template <typename T>
constexpr size_t q_region_doubles(int n) {
  const size_t t = ((size_t)(p_cells_per_wave(n) + 1) * q_cell_stride(n) * sizeof(T) + 7) / 8;
  const size_t u = (size_t)p_kgu(n) * 64 + p_priv_max(n);
  return t > u ? t : u;
}
template <typename T>
constexpr size_t q_lds_bytes(int n, bool hn) {
  return (q_region_doubles<T>(n) + (hn ? (size_t)n * n : 0)) * 8;
}

constexpr uint8_t kFlagConstrained = 1;  // batch dof is a constrained row (identity)
constexpr uint8_t kFlagAdd = 2;          // batch is NOT the first toucher: dst += (else dst =)

// Host-side execution plan: replaces deal.II GraphColoring + the per-colour arrays of
// ReinitHelper::init_with_coloring (reference matrix_free_gpu.cu:157-186).  Cells are grouped
// into batches (one workgroup each); batches, not cells, are coloured.
struct Plan {
  int dim = 0, degree = 0, n = 0, nd = 0;
  uint32_t n_dofs = 0, n_cells = 0;
  std::vector<uint32_t> cell_order;       // plan position -> caller's cell index
  std::vector<uint32_t> batch_cell_off;   // [n_batches+1] into plan positions
  std::vector<uint32_t> batch_dof_off;    // [n_batches+1] into bdofs
  std::vector<uint32_t> color_batch_off;  // [n_colors+1]
  std::vector<uint32_t> bdofs;            // global dof ids per batch [interior asc | shared asc]; bit 31 = constrained
  std::vector<uint8_t> bflags;            // per bdofs entry
  std::vector<uint16_t> lmap;             // [n_cells*nd] plan order: local dof -> batch-local id
  std::vector<uint32_t> orphans;          // dofs touched by no cell; bit 31 set = constrained
  // two-pass ("owner gathers") mode: per batch the dofs are ordered [interior | shared]; interior
  // dofs (touched by this batch only) are written straight to dst, partial sums of shared dofs go
  // to a halo buffer slot and are summed per dof by a second kernel in fixed batch order
  std::vector<uint32_t> batch_nint;       // [n_batches] number of interior dofs of the batch
  std::vector<uint32_t> halo_off;         // [n_batches+1] first halo slot of the batch's shared dofs
  std::vector<uint32_t> sdofs;            // [n_shared] global ids grouped by toucher set; bit 31 = constrained
  std::vector<uint32_t> s_off;            // [n_shared+1] CSR into s_idx
  std::vector<uint32_t> s_idx;            // halo slots of the partial sums, ascending batch order
  std::vector<uint32_t> chunks;           // [4 * n_chunks] {sdofs position, count | k << 16, gstarts offset, offset in group}
  std::vector<uint32_t> gstarts;          // per group: halo slot of its first dof in each of its k touchers
  uint32_t max_batch_dofs = 0, max_batch_cells = 0;
  uint32_t n_plane_batches = 0;  // the first n_plane_batches batches run in the plane kernels, the rest in apply_batches_x
  // ... of which the first n_plain_plane_batches hold cells without a mask only (plain instantiation); the others run
  // in the <HN> instantiation.  With PlanLimits::masked_planes the batches of masked and of unmasked cells are NOT
  // sorted apart (one launch, creation order = spatial order): 0 as soon as one batch has a masked cell.
  uint32_t n_plain_plane_batches = 0;
  // plane kernels (build_plane_records): fixed-size per-batch records -- dof lists, index runs, and for the batches
  // of masked cells the hanging-node records; pr_hn_slot[b] = index of batch b's record in pr_hn, or 0xffffffff
  std::vector<uint32_t> pr_dofs, pr_idx, pr_hn, pr_hn_slot;
  uint64_t n_first = 0, n_add = 0;
};

// Batch limits a cell-loop kernel imposes on the planner (0 = derive from the description, see
// default_batch_limits): apply_planes3 processes a batch in one pass of one wave.
struct PlanLimits {
  uint32_t max_cells = 0, max_dofs = 0;
  // > 0 (apply_planes3): at most interior_max dofs of a batch stay on the interior route (stored by the cell loop,
  // never constrained); the rest -- at most shared_max, else the plan fails with MFGPU_EUNSUPPORTED -- takes the
  // pass-2 route, and every batch has at least one pass-2 dof; every batch owns halo_stride halo slots
  uint32_t interior_max = 0, shared_max = 0, halo_stride = 0;
  // meshes with hanging nodes: a batch holds masked cells only or unmasked cells only; the batches of unmasked cells
  // (plane kernel) come first, then the batches of masked cells under the pencil kernel's limits (see build_plan)
  bool segregate_masked = false;
  // ... or, with masked_planes, the masked cells' batches are plane batches too (<HN> instantiation): same slot
  // structure, at most private_max private entries per batch for the cells' constrained nodes, and the two kinds
  // of batches stay interleaved in creation order (ONE launch of the <HN> instantiation walks them all)
  bool masked_planes = false;
  uint32_t private_max = 0;
};

// Build the plan from a description (validates it).  Returns 0 or MFGPU_E*.
// max_chunks: chunks of cells per batch the cell-loop kernel unrolls (3; apply_batches_x at p=3: 4)
int build_plan(const mfgpu_desc &d, Plan &plan, uint32_t max_chunks = 3, const PlanLimits *limits = nullptr);

// The line operations of interpolate_boundary_3d (hanging_nodes.cuh:617-696) for one cell mask: per direction d
// (passes run in the order x, y, z) the flagged lines, each as its n local node ids (x + n y + n^2 z) listed in the
// order in which the PLAIN weight matrix W applies to them -- ascending along d where the mask's type bit of d is set,
// descending otherwise (W reversed in both indices == W on the reversed line).  nodes: every local node on a flagged
// line, ascending.
struct HnLine {
  uint16_t node[8];
};
void hn_cell_lines(unsigned mask, int n, std::vector<HnLine> (&lines)[3], std::vector<uint16_t> &nodes);

// Which cell-loop kernel family serves a description (mfgpu_desc.kernel; 0 = the library's choice), and the plan
// built for it -- shared by mfgpu_create and the host-only mfgpu_plan_create:
//   planes   apply_planes3: 3D, uniform-Jacobian path, two-pass mode, p = 4 by default (p = 2, 3 on request:
//            mfgpu_desc.kernel = MFGPU_KERNEL_PLANES); on meshes with hanging nodes it takes the batches of cells
//            without a constraint mask (Plan::n_plane_batches) and pencils_x is set as well for the others
//   pencils_x apply_batches_x: 3D two-pass otherwise
//   general  apply_batches_g: no MFGPU_UNIFORM_J0
//   none of them: apply_batches (2D, coloured-scatter mode)
struct KernelChoice {
  bool planes = false, pencils_x = false, general = false;
};
int choose_kernel_and_plan(const mfgpu_desc &d, KernelChoice &kc, Plan &plan);
int build_plane_records(Plan &plan, const uint32_t *constraint_mask);

// Derive the kernel's 1D tables from the reference-layout tables T[dof*n+q]:
//   S[i*n+q]  = shape_values (interpolation nodal -> quadrature points)
//   Dt[q*n+t] = l_t'(x_q): collocation derivative on the quadrature points, from
//               shape_gradients = S * Dt^T  (phi_i'(x_q) = sum_t phi_i(x_t) l_t'(x_q))
int derive_tables(int n, const double *shape_values, const double *shape_gradients,
                  std::vector<double> &S, std::vector<double> &Dt);

// entry points of mfgpu_dist.hip into the operator (mfgpu_api.hip): phase 0 = cell loop, 1 = pass 2 of the priority
// dofs, 2 = pass 2 of the rest
int handle_number_type(const mfgpu_handle *h);
int handle_set_priority_dofs(mfgpu_handle *h, const uint32_t *ids, uint32_t n);
int handle_vmult_phase(mfgpu_handle *h, int phase, void *dst, const void *src, void *stream, int add);
// interface-first schedule of mfgpu_dist: the batches that list any of `ids` (flags[b] = 1),
// the cell loop over a batch range, one pass-2 group (0 = the priority dofs, 1 = the rest) on a stream of the caller's
// choice.  handle_ranged_ok: two-pass mode with one cell-loop segment (else the caller keeps the phase schedule).
int handle_n_batches(const mfgpu_handle *h);
bool handle_ranged_ok(const mfgpu_handle *h);
int handle_batches_touching(const mfgpu_handle *h, const uint32_t *ids, uint32_t n, std::vector<uint8_t> &flags);
int handle_cells_range(mfgpu_handle *h, uint32_t b0, uint32_t b1, void *dst, const void *src, void *stream, int add);
int handle_cells_two_ranges(mfgpu_handle *h, uint32_t b0, uint32_t b1, uint32_t c0, uint32_t c1, void *dst,
                            const void *src, void *stream, int add);
int handle_pass2_group(mfgpu_handle *h, int group, void *dst, const void *src, void *stream, int add);

}  // namespace mfgpu

struct mfgpu_plan {
  mfgpu::Plan plan;
};

#endif
