// Host-side planner: groups cells into batches (one workgroup each), colours the batches,
// and decides for every (batch, dof) whether the batch stores (first toucher) or adds.
//
// Replaces, MI355X-first, the reference's per-CELL graph colouring (coloring.cc:8-33,
// matrix_free_gpu.cu:157-186): a batch sums every dof shared by its own cells in LDS, so
// only dofs on batch surfaces are read-modify-written in global memory, and the first
// toucher of a dof stores instead of adding, which removes the separate `dst = 0` pass
// (laplace_operator_gpu.h:221) and the save/load constrained-row kernels
// (constraint_handler_gpu.cu:247-289) from the hot loop.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "mfgpu_internal.h"

namespace mfgpu {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
const char *last_error() { return g_err.c_str(); }

int derive_tables(int n, const double *sv, const double *sg, std::vector<double> &S,
                  std::vector<double> &Dt) {
  // Solve S * X = G for X = Dt^T (n x n), long double Gaussian elimination with pivoting.
  std::vector<long double> A(n * n), B(n * n);
  for (int i = 0; i < n * n; ++i) {
    A[i] = sv[i];
    B[i] = sg[i];
  }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (fabsl(A[r * n + c]) > fabsl(A[piv * n + c])) piv = r;
    if (fabsl(A[piv * n + c]) < 1e-30L) {
      set_error("shape_values table is singular");
      return MFGPU_EINVAL;
    }
    if (piv != c)
      for (int k = 0; k < n; ++k) {
        std::swap(A[c * n + k], A[piv * n + k]);
        std::swap(B[c * n + k], B[piv * n + k]);
      }
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      long double f = A[r * n + c] / A[c * n + c];
      for (int k = 0; k < n; ++k) {
        A[r * n + k] -= f * A[c * n + k];
        B[r * n + k] -= f * B[c * n + k];
      }
    }
  }
  S.assign(sv, sv + n * n);
  Dt.resize(n * n);
  // X[t][q] = B[t][q]/A[t][t];  Dt[q*n+t] = X[t][q]
  for (int t = 0; t < n; ++t)
    for (int q = 0; q < n; ++q) Dt[q * n + t] = (double)(B[t * n + q] / A[t * n + t]);
  return 0;
}

static void default_batch_limits(const mfgpu_desc &d, uint32_t max_chunks, uint32_t &max_cells, uint32_t &max_dofs) {
  const int p = d.degree, dim = d.dim;
  max_dofs = d.max_dofs_per_batch ? d.max_dofs_per_batch : 2304u;
  // the kernel keeps a batch's dof list and source values in registers: kGU * kBlock = 9 * 256 dofs
  if (max_dofs > 2304u) max_dofs = 2304u;
  const uint32_t nd = (uint32_t)ipow(p + 1, dim);
  if (max_dofs < nd) max_dofs = nd;
  if (d.max_cells_per_batch) {
    max_cells = d.max_cells_per_batch;
  } else {
    // the largest near-cubic box of cells (edge lengths m or m + 1) whose dofs fit: 3x3x3 at p=4, 4x4x4 at
    // p=3, 3x2x2 at p=5 (the cube 2x2x2 would leave the dof budget half empty); at low degree the chunk
    // limit below is the binding one
    int m = 1;
    while ((uint64_t)ipow((m + 1) * p + 1, dim) <= max_dofs) ++m;
    uint64_t best = (uint64_t)ipow(m, dim);
    for (int k = 1; k < dim; ++k) {  // k edges of length m + 1, dim - k of length m
      const uint64_t dofs = (uint64_t)ipow((m + 1) * p + 1, k) * (uint64_t)ipow(m * p + 1, dim - k);
      if (dofs <= max_dofs) best = (uint64_t)ipow(m + 1, k) * (uint64_t)ipow(m, dim - k);
    }
    max_cells = (uint32_t)best;
    // keep enough workgroups per launch on small meshes
    const uint32_t cap = std::max<uint32_t>(1u, d.n_cells / 4096u);
    max_cells = std::min(max_cells, cap);
  }
  // the kernel unrolls at most max_chunks chunks of CH = threads / n^(dim-1) cells per batch
  const uint32_t threads = d.max_dofs_per_batch && d.max_dofs_per_batch <= 768 ? 64u : 256u;
  const uint32_t ch = std::max<uint32_t>(1u, threads / (uint32_t)ipow(p + 1, dim - 1));
  max_cells = std::min(max_cells, max_chunks * ch);
  if (max_cells < 1) max_cells = 1;
}

void hn_cell_lines(unsigned mask, int n, std::vector<HnLine> (&lines)[3], std::vector<uint16_t> &nodes) {
  const int p = n - 1;
  const unsigned TYPE[3] = {1u << 0, 1u << 1, 1u << 2}, FACE[3] = {1u << 3, 1u << 4, 1u << 5};
  const unsigned EDGE[3] = {1u << 7, 1u << 8, 1u << 6};  // direction x: edge YZ, y: ZX, z: XY (hanging_nodes.cuh:38-50)
  const int stride[3] = {1, n, n * n};
  std::vector<uint8_t> on(n * n * n, 0);
  for (int d = 0; d < 3; ++d) {
    lines[d].clear();
    const int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
    if (!(mask & (FACE[d1] | FACE[d2] | EDGE[d]))) continue;
    const int i1 = (mask & TYPE[d1]) ? 0 : p, i2 = (mask & TYPE[d2]) ? 0 : p;
    const bool typ = (mask & TYPE[d]) != 0;
    for (int a = 0; a < n; ++a)
      for (int b = 0; b < n; ++b) {
        const bool flag = ((mask & FACE[d1]) && a == i1) || ((mask & FACE[d2]) && b == i2) ||
                          ((mask & EDGE[d]) && a == i1 && b == i2);
        if (!flag) continue;
        HnLine L{};
        for (int t = 0; t < n; ++t) {
          const int node = (typ ? t : p - t) * stride[d] + a * stride[d1] + b * stride[d2];
          L.node[t] = (uint16_t)node;
          on[node] = 1;
        }
        lines[d].push_back(L);
      }
  }
  nodes.clear();
  for (int i = 0; i < n * n * n; ++i)
    if (on[i]) nodes.push_back((uint16_t)i);
}

int build_plan(const mfgpu_desc &d, Plan &P, uint32_t max_chunks, const PlanLimits *limits) {
  if (d.dim != 2 && d.dim != 3) {
    set_error("dim must be 2 or 3");
    return MFGPU_EINVAL;
  }
  if (d.degree < 1 || d.degree > 6) {
    set_error("degree must be in 1..6");
    return MFGPU_EUNSUPPORTED;
  }
  if (!d.loc2glob || d.n_cells == 0 || d.n_dofs == 0) {
    set_error("empty description (n_cells, n_dofs and loc2glob are required)");
    return MFGPU_EINVAL;
  }
  if (d.n_constrained && !d.constrained_dofs) {
    set_error("n_constrained > 0 but constrained_dofs is NULL");
    return MFGPU_EINVAL;
  }
  P.dim = d.dim;
  P.degree = d.degree;
  P.n = d.degree + 1;
  P.nd = ipow(P.n, d.dim);
  P.n_dofs = d.n_dofs;
  P.n_cells = d.n_cells;
  const uint32_t nd = (uint32_t)P.nd, nc = d.n_cells, N = d.n_dofs;
  const uint32_t *l2g = d.loc2glob;

  for (uint64_t i = 0; i < (uint64_t)nc * nd; ++i)
    if (l2g[i] >= N) {
      set_error("loc2glob entry out of range");
      return MFGPU_EINVAL;
    }
  std::vector<uint8_t> constrained(N, 0);
  for (uint32_t i = 0; i < d.n_constrained; ++i) {
    if (d.constrained_dofs[i] >= N) {
      set_error("constrained_dofs entry out of range");
      return MFGPU_EINVAL;
    }
    constrained[d.constrained_dofs[i]] = 1;
  }

  if (N >= 0x80000000u) {
    set_error("n_dofs >= 2^31 is not supported (bit 31 of the dof lists carries the constrained flag)");
    return MFGPU_EUNSUPPORTED;
  }
  // limits->segregate_masked (meshes with hanging nodes under apply_planes3): cells WITHOUT a hanging-node mask are
  // batched under `limits` and run in the plane kernel, which has no constraint stages; cells with a mask get batches of
  // their own under the pencil kernel's default limits (Bmax1, NBmax1) and run in apply_batches_x<HN>.  The plane
  // batches come first in the execution order (P.n_plane_batches of them).
  const bool segregate = limits && limits->segregate_masked && (d.flags & MFGPU_HANGING_NODES) && d.constraint_mask;
  uint32_t Bmax1 = 0, NBmax1 = 0;
  if (segregate) default_batch_limits(d, max_chunks, Bmax1, NBmax1);
  // masked cells in plane batches of their own (apply_planes3<HN>): the plane limits, plus the private entries of the
  // cells' constrained nodes (counted per mask value)
  const bool masked_planes = segregate && limits->masked_planes && limits->max_cells && limits->max_dofs;
  std::vector<uint32_t> priv_of_mask;  // mask value (9 bits) -> private entries of such a cell
  if (masked_planes) {
    priv_of_mask.assign(512, 0u);
    std::vector<HnLine> ln[3];
    std::vector<uint16_t> nodes;
    for (unsigned m = 1; m < 512; ++m) {
      hn_cell_lines(m, P.n, ln, nodes);
      priv_of_mask[m] = (uint32_t)nodes.size();
    }
    for (uint32_t c = 0; c < nc; ++c)
      if (d.constraint_mask[c] >= 512) {
        set_error("constraint_mask has bits beyond the nine of hanging_nodes.cuh:38-50");
        return MFGPU_EINVAL;
      }
  }
  auto priv_of = [&](uint32_t c) { return masked_planes ? priv_of_mask[d.constraint_mask[c]] : 0u; };
  uint32_t Bmax, NBmax;
  if (limits && limits->max_cells && limits->max_dofs) {
    // kernel-imposed limits; the caller's knobs may only tighten them
    Bmax = d.max_cells_per_batch ? std::min(d.max_cells_per_batch, limits->max_cells) : limits->max_cells;
    NBmax = d.max_dofs_per_batch ? std::min(d.max_dofs_per_batch, limits->max_dofs) : limits->max_dofs;
    if (NBmax < (uint32_t)P.nd) {
      set_error("max_dofs_per_batch is smaller than one cell");
      return MFGPU_EINVAL;
    }
  } else {
    default_batch_limits(d, max_chunks, Bmax, NBmax);
  }

  // dof -> cells incidence (CSR)
  std::vector<uint32_t> dc_off(N + 1, 0);
  for (uint64_t i = 0; i < (uint64_t)nc * nd; ++i) dc_off[l2g[i] + 1]++;
  for (uint32_t g = 0; g < N; ++g) dc_off[g + 1] += dc_off[g];
  std::vector<uint32_t> dc(dc_off[N]);
  {
    std::vector<uint32_t> pos(dc_off.begin(), dc_off.end() - 1);
    for (uint32_t c = 0; c < nc; ++c)
      for (uint32_t i = 0; i < nd; ++i) dc[pos[l2g[(uint64_t)c * nd + i]]++] = c;
    // a cell that lists one dof twice (substituted / degenerate loc2glob) counts once: the cells of a dof are
    // in ascending order, so duplicates are adjacent
    uint32_t w = 0;
    for (uint32_t g = 0; g < N; ++g) {
      const uint32_t beg = dc_off[g], end = dc_off[g + 1];
      dc_off[g] = w;
      for (uint32_t k = beg; k < end; ++k)
        if (k == beg || dc[k] != dc[k - 1]) dc[w++] = dc[k];
    }
    dc_off[N] = w;
    dc.resize(w);
  }

  // ---- batching.  A batch starts at the lowest unassigned cell -- as a BOX of cells where the mesh offers one (plane
  // plans; see below) -- and grows greedily: always the candidate that shares most dofs with the batch (ties: earliest
  // discovered), until a limit is hit.  With limits->interior_max (plane kernels) a batch must also keep its SURFACE
  // within the pass-2 slots of the dof list: surface = dofs with an incident cell outside the batch, or constrained,
  // plus the interior dofs beyond interior_max.  The surface is not monotone in the number of cells, so growth runs to
  // the cell / dof limit and the batch is then cut back to the longest prefix of its growth order that satisfied the
  // bound.
  constexpr uint32_t NONE = 0xffffffffu;
  const bool bound_surface_any = limits && limits->interior_max;
  auto masked = [&](uint32_t c) { return segregate && d.constraint_mask[c] != 0; };

  // Face neighbours by direction (plane plans, n >= 3): the cell across face (axis, side) is the other cell of a dof in
  // the interior of that face, provided it lists the dof at the mirrored position (same size, same orientation; a
  // hanging-node face, whose entries were substituted, has no neighbour in this sense).  They let a batch start as a
  // box a x b x c: on a mesh whose extent is no multiple of the natural box (64 cells: 21 boxes of 3 and one cell over)
  // greedy growth alone wraps around the row ends and fills the mesh with irregular 9-11-cell batches whose surface
  // exceeds the dof list (n = 64: 23 340 batches where 21 845 would do; profiles/r03_notes.md section 9).
  std::vector<uint32_t> nbr;
  // (conforming meshes only: on the octree meshes with hanging nodes, cells in Morton order and two kinds of cells,
  // boxes anchored at the lowest unassigned cell leave more single-cell leftovers than they fill batches -- 17 899
  // against 17 777 batches on C3)
  const bool use_boxes = bound_surface_any && !segregate && P.dim == 3 && P.n >= 3 && Bmax >= 4;
  if (use_boxes) {
    nbr.assign((size_t)nc * 6, NONE);
    const uint32_t n_ = (uint32_t)P.n, mid = 1;
    for (uint32_t c = 0; c < nc; ++c)
      for (uint32_t axis = 0; axis < 3; ++axis)
        for (uint32_t side = 0; side < 2; ++side) {
          uint32_t ijk[3] = {mid, mid, mid}, opp[3] = {mid, mid, mid};
          ijk[axis] = side ? n_ - 1 : 0;
          opp[axis] = side ? 0 : n_ - 1;
          const uint32_t li = ijk[0] + n_ * ijk[1] + n_ * n_ * ijk[2], lo = opp[0] + n_ * opp[1] + n_ * n_ * opp[2];
          const uint32_t g = l2g[(uint64_t)c * nd + li];
          if (dc_off[g + 1] - dc_off[g] != 2) continue;
          const uint32_t c2 = dc[dc_off[g]] == c ? dc[dc_off[g] + 1] : dc[dc_off[g]];
          if (c2 != c && l2g[(uint64_t)c2 * nd + lo] == g) nbr[(size_t)c * 6 + 2 * axis + side] = c2;
        }
  }
  // box shapes a x b x c (cells along x, y, z) up to the cell limit: most cells first, then the most compact
  struct Shape {
    uint32_t a, b, c;
  };
  std::vector<Shape> shapes;
  if (use_boxes) {
    for (uint32_t a = 1; a <= Bmax; ++a)
      for (uint32_t b2 = 1; a * b2 <= Bmax; ++b2)
        for (uint32_t c2 = 1; a * b2 * c2 <= Bmax; ++c2)
          if (a * b2 * c2 == Bmax) shapes.push_back({a, b2, c2});  // (smaller boxes fragment irregular meshes: the growth does better)
    auto spread = [](const Shape &s2) { return std::max(s2.a, std::max(s2.b, s2.c)) * 4 + s2.a + s2.b + s2.c; };
    std::stable_sort(shapes.begin(), shapes.end(), [&](const Shape &x, const Shape &y) {
      const uint32_t vx = x.a * x.b * x.c, vy = y.a * y.b * y.c;
      if (vx != vy) return vx > vy;
      if (spread(x) != spread(y)) return spread(x) < spread(y);
      return x.a != y.a ? x.a > y.a : x.b > y.b;  // the long side along x: the dofs of a batch are x-runs
    });
  }

  std::vector<uint8_t> batch_masked;  // per batch: class of its cells (segregate only)
  std::vector<uint32_t> cell_batch(nc, NONE);
  // state of the batch under construction; `stamp` changes with every (re)build, so nothing is ever cleared
  std::vector<uint32_t> dof_stamp(N, NONE);   // stamp of the build that already contains this dof
  std::vector<uint32_t> inc_cnt(bound_surface_any ? N : 0, 0);  // incident cells of the dof inside the current batch
  std::vector<uint32_t> gain(nc, 0), gain_stamp(nc, NONE);
  std::vector<uint32_t> cand, box;
  std::vector<std::vector<uint32_t>> batches;
  uint32_t stamp = 0;
  uint32_t seed = 0;
  while (true) {
    while (seed < nc && cell_batch[seed] != NONE) ++seed;
    if (seed >= nc) break;
    const uint32_t b = (uint32_t)batches.size();
    batches.emplace_back();
    std::vector<uint32_t> &cells = batches.back();
    const bool cls = masked(seed);
    batch_masked.push_back(cls);
    const bool bound_surface = bound_surface_any && (!cls || masked_planes);
    const uint32_t Bmax_b = cls && !masked_planes ? Bmax1 : Bmax, NBmax_b = cls && !masked_planes ? NBmax1 : NBmax;
    uint32_t ndofs = 0, n_enclosed = 0, npriv = 0;
    auto surface_ok = [&]() {
      return !bound_surface || ndofs - std::min(n_enclosed, limits->interior_max) <= limits->shared_max;
    };
    auto begin_build = [&]() {
      for (uint32_t c : cells) cell_batch[c] = NONE;
      cells.clear();
      cand.clear();
      ++stamp;
      ndofs = n_enclosed = npriv = 0;
    };
    auto add_cell = [&](uint32_t c) {
      cell_batch[c] = b;
      cells.push_back(c);
      npriv += priv_of(c);
      for (uint32_t i = 0; i < nd; ++i) {
        const uint32_t g = l2g[(uint64_t)c * nd + i];
        const bool first = dof_stamp[g] != stamp;
        if (bound_surface) {
          // (a cell listing one dof twice counts once: compare with the previous entries of this cell)
          bool dup = false;
          for (uint32_t i2 = 0; i2 < i && !dup; ++i2) dup = l2g[(uint64_t)c * nd + i2] == g;
          if (!dup) {
            if (first) inc_cnt[g] = 0;
            if (++inc_cnt[g] == dc_off[g + 1] - dc_off[g] && !constrained[g]) ++n_enclosed;
          }
        }
        if (!first) continue;
        dof_stamp[g] = stamp;
        ++ndofs;
        if (Bmax_b == 1) continue;
        for (uint32_t k = dc_off[g]; k < dc_off[g + 1]; ++k) {
          const uint32_t c2 = dc[k];
          if (cell_batch[c2] != NONE || masked(c2) != cls) continue;
          if (gain_stamp[c2] != stamp) {
            gain_stamp[c2] = stamp;
            gain[c2] = 0;
            cand.push_back(c2);
          }
          gain[c2]++;
        }
      }
    };

    // ---- the box the batch starts as: the first shape whose cells exist from the seed in +x, +y, +z, are unassigned
    // and of the seed's kind, and which keeps every bound
    bool boxed = false;
    if (use_boxes && bound_surface && Bmax_b == Bmax) {
      for (const Shape &sh : shapes) {
        box.clear();
        bool ok = true;
        uint32_t cz = seed;
        for (uint32_t k = 0; k < sh.c && ok; ++k) {
          uint32_t cy = cz;
          for (uint32_t j = 0; j < sh.b && ok; ++j) {
            uint32_t cx = cy;
            for (uint32_t i = 0; i < sh.a && ok; ++i) {
              if (cx == NONE || cell_batch[cx] != NONE || masked(cx) != cls) {
                ok = false;
                break;
              }
              box.push_back(cx);
              cx = nbr[(size_t)cx * 6 + 1];
            }
            cy = cy == NONE ? NONE : nbr[(size_t)cy * 6 + 3];
            if (cy == NONE && j + 1 < sh.b) ok = false;
          }
          cz = cz == NONE ? NONE : nbr[(size_t)cz * 6 + 5];
          if (cz == NONE && k + 1 < sh.c) ok = false;
        }
        if (!ok) continue;
        begin_build();
        for (uint32_t c : box) add_cell(c);
        if (ndofs <= NBmax_b && surface_ok() && (!(cls && masked_planes) || npriv <= limits->private_max)) {
          boxed = true;
          break;
        }
      }
      if (!boxed) begin_build();
    } else {
      begin_build();
    }

    // ---- greedy growth (from the seed, or on from the box)
    size_t last_ok = boxed ? cells.size() : 0;
    uint32_t next = boxed ? NONE : seed;
    while (true) {
      if (next != NONE) {
        add_cell(next);
        if (surface_ok()) last_ok = cells.size();
      }
      if (cells.size() >= Bmax_b) break;
      // pick best candidate
      uint32_t best = NONE, best_gain = 0;
      size_t w = 0;
      for (size_t k = 0; k < cand.size(); ++k) {
        const uint32_t c2 = cand[k];
        if (cell_batch[c2] != NONE) continue;  // was taken
        cand[w++] = c2;
        if (gain[c2] > best_gain) {
          best_gain = gain[c2];
          best = c2;
        }
      }
      cand.resize(w);
      if (best == NONE) break;
      if (ndofs + (nd - best_gain) > NBmax_b) break;
      if (cls && masked_planes && npriv + priv_of(best) > limits->private_max) break;
      next = best;
    }
    if (last_ok == 0) last_ok = 1;  // (a single cell over the bound is reported by the classification below)
    for (size_t k = last_ok; k < cells.size(); ++k) cell_batch[cells[k]] = NONE;
    cells.resize(last_ok);
    if (seed < nc && cell_batch[seed] != b) seed = 0;  // defensive: the seed is the first cell of its batch
  }
  if (segregate && !masked_planes) {  // plane batches first (stable)
    std::vector<uint32_t> idx(batches.size());
    std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b2) { return batch_masked[a] < batch_masked[b2]; });
    std::vector<std::vector<uint32_t>> sorted_batches(batches.size());
    std::vector<uint8_t> sorted_masked(batches.size());
    for (size_t k = 0; k < idx.size(); ++k) {
      sorted_batches[k] = std::move(batches[idx[k]]);
      sorted_masked[k] = batch_masked[idx[k]];
    }
    batches.swap(sorted_batches);
    batch_masked.swap(sorted_masked);
  } else if (!segregate) {
    batch_masked.assign(batches.size(), 0);
  }
  uint32_t nb = (uint32_t)batches.size();
  // cells with a hanging-node mask first: the kernel takes the extra interpolation stages for a whole
  // chunk of cells as soon as one of them is masked, so masked cells should share chunks
  if ((d.flags & MFGPU_HANGING_NODES) && d.constraint_mask)
    for (auto &cells : batches)
      std::stable_partition(cells.begin(), cells.end(), [&](uint32_t c) { return d.constraint_mask[c] != 0; });

  // ---- per batch: unique dofs, ordered [interior ascending | shared] where interior = touched by this
  // batch only (the shared part is re-ordered by toucher group below)
  //
  // A dof takes the pass-2 route ("shared") if two or more batches touch it.  With limits->interior_max
  // (apply_planes3: fixed slot structure of the batch dof list, no per-lane case distinction in the scatter) more
  // dofs are DEMOTED to that route although only this batch touches them (one partial sum; pass 2 copies it):
  // constrained dofs -- pass 2 writes the identity row of every constrained dof it lists, so the cell loop never
  // stores to one --, the interior dofs beyond interior_max, and one dof if the batch would otherwise have no pass-2
  // dof at all (the padding entries of the interior slots store a zero to a pass-2 dof of the batch).  A batch with
  // more than shared_max pass-2 dofs (the growth above only bounds the total) is split in two and everything is
  // classified again.
  const uint32_t interior_max = limits ? limits->interior_max : 0u;
  std::vector<std::vector<uint32_t>> bd;
  std::vector<uint32_t> ntouch, nint;
  std::vector<uint8_t> shared_flag;
  for (;;) {
    nb = (uint32_t)batches.size();
    bd.assign(nb, {});
    ntouch.assign(N, 0);
    for (uint32_t b = 0; b < nb; ++b) {
      std::vector<uint32_t> &v = bd[b];
      v.reserve(batches[b].size() * nd);
      for (uint32_t c : batches[b])
        for (uint32_t i = 0; i < nd; ++i) v.push_back(l2g[(uint64_t)c * nd + i]);
      std::sort(v.begin(), v.end());
      v.erase(std::unique(v.begin(), v.end()), v.end());
      if (v.size() > (batch_masked[b] && !masked_planes ? NBmax1 : NBmax) || v.size() > 8191u) {
        // (the greedy estimate nd - gain under-counts a cell that lists one dof twice; the kernels hold a batch's
        // dofs in a fixed number of register / LDS slots and byte offsets of batch-local ids in 16 bits)
        set_error("internal: batch exceeds the kernel's dof slots (degenerate loc2glob?)");
        return MFGPU_EINVAL;
      }
      for (uint32_t g : v) ntouch[g]++;
    }
    shared_flag.assign(N, 0);
    for (uint32_t g = 0; g < N; ++g) shared_flag[g] = ntouch[g] >= 2;
    if (interior_max)  // constrained dofs a single PLANE batch touches are demoted (the pencil kernel writes its own)
      for (uint32_t b = 0; b < nb; ++b)
        if (!batch_masked[b] || masked_planes)
          for (uint32_t g : bd[b])
            if (ntouch[g] == 1 && constrained[g]) shared_flag[g] = 1;
    nint.assign(nb, 0);
    std::vector<uint32_t> too_big;
    for (uint32_t b = 0; b < nb; ++b) {
      std::vector<uint32_t> &v = bd[b];
      std::stable_partition(v.begin(), v.end(), [&](uint32_t g) { return !shared_flag[g]; });
      uint32_t k = 0;
      while (k < v.size() && !shared_flag[v[k]]) ++k;
      if (interior_max && (!batch_masked[b] || masked_planes)) {
        uint32_t keep = std::min(k, interior_max);
        if (keep == v.size() && keep > 0) --keep;
        for (uint32_t t = keep; t < k; ++t) shared_flag[v[t]] = 1;
        k = keep;
        if (v.size() - k > limits->shared_max) too_big.push_back(b);
      }
      nint[b] = k;
    }
    if (too_big.empty()) break;
    for (size_t i = too_big.size(); i-- > 0;) {  // back to front: indices stay valid
      const uint32_t b = too_big[i];
      if (batches[b].size() < 2) {
        set_error("one cell has more pass-2 dofs than the plane kernel's dof-list slots hold");
        return MFGPU_EUNSUPPORTED;
      }
      const size_t half = batches[b].size() / 2;
      std::vector<uint32_t> second(batches[b].begin() + half, batches[b].end());
      batches[b].resize(half);
      batches.insert(batches.begin() + b + 1, std::move(second));
      batch_masked.insert(batch_masked.begin() + b + 1, batch_masked[b]);
    }
  }

  // ---- greedy colouring of batches (conflict = shared dof)
  std::vector<uint64_t> dof_colors(N, 0);
  std::vector<uint32_t> bcolor(nb, 0);
  uint32_t ncolors = 0;
  for (uint32_t b = 0; b < nb; ++b) {
    uint64_t forbidden = 0;
    for (uint32_t g : bd[b]) forbidden |= dof_colors[g];
    uint32_t c = 0;
    while (c < 64 && (forbidden >> c) & 1) ++c;
    if (c >= 64) {
      set_error("more than 64 batch colours needed");
      return MFGPU_EUNSUPPORTED;
    }
    bcolor[b] = c;
    ncolors = std::max(ncolors, c + 1);
    for (uint32_t g : bd[b]) dof_colors[g] |= (1ull << c);
  }
  // execution order.  Coloured scatter: colour-major, stable (one launch per colour).  Two-pass scatter
  // has no inter-batch dependency: the batches keep their creation order, which follows the caller's
  // cell order and is therefore spatially coherent -- batches that run at the same time on one XCD are
  // mesh neighbours and share the 128-byte lines of src that their dof runs straddle (a colour-major
  // order puts every neighbour into a different colour, i.e. as far apart in time as possible).
  const bool colored = (d.flags & MFGPU_COLORED_SCATTER) != 0;
  std::vector<uint32_t> order(nb);
  std::iota(order.begin(), order.end(), 0u);
  if (colored) {
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t a, uint32_t b2) { return bcolor[a] < bcolor[b2]; });
    P.color_batch_off.assign(ncolors + 1, 0);
    for (uint32_t b = 0; b < nb; ++b) P.color_batch_off[bcolor[b] + 1]++;
    for (uint32_t c = 0; c < ncolors; ++c) P.color_batch_off[c + 1] += P.color_batch_off[c];
  } else {
    P.color_batch_off.assign({0u, nb});
  }

  // ---- shared dofs, grouped by toucher set.  All dofs that are shared by the same set of batches (the
  // interior of a face between two batches, of an edge between four, a vertex between eight) form a
  // GROUP.  The global shared list (pass 2 walks it) is sorted by (toucher sequence in execution order,
  // global id) and every batch lists its shared dofs in that same order, so a group occupies ONE
  // contiguous run of halo slots in each of its touchers, with identical dof order: pass 2 reads the
  // partial sums of a group as k coalesced runs and needs no per-partial index.
  std::vector<uint32_t> rank_of(nb);
  for (uint32_t k = 0; k < nb; ++k) rank_of[order[k]] = k;
  std::vector<uint32_t> shared_ids;
  for (uint32_t g = 0; g < N; ++g)
    if (shared_flag[g]) shared_ids.push_back(g);
  const size_t ns = shared_ids.size();
  std::vector<uint32_t> sid(N, NONE), t_off(ns + 1, 0);
  for (size_t i = 0; i < ns; ++i) {
    sid[shared_ids[i]] = (uint32_t)i;
    t_off[i + 1] = t_off[i] + ntouch[shared_ids[i]];
  }
  std::vector<uint32_t> touchers(t_off[ns]);
  {
    std::vector<uint32_t> fill(t_off.begin(), t_off.end() - 1);
    for (uint32_t k = 0; k < nb; ++k)  // ascending execution position
      for (size_t t = nint[order[k]]; t < bd[order[k]].size(); ++t) touchers[fill[sid[bd[order[k]][t]]]++] = k;
  }
  auto seq_less = [&](uint32_t a, uint32_t b2) {  // a, b2: indices into shared_ids
    const uint32_t la = t_off[a + 1] - t_off[a], lb = t_off[b2 + 1] - t_off[b2];
    const uint32_t *pa = &touchers[t_off[a]], *pb = &touchers[t_off[b2]];
    for (uint32_t i = 0; i < la && i < lb; ++i)
      if (pa[i] != pb[i]) return pa[i] < pb[i];
    if (la != lb) return la < lb;
    return a < b2;  // same group: ascending global id
  };
  auto same_group = [&](uint32_t a, uint32_t b2) {
    const uint32_t la = t_off[a + 1] - t_off[a], lb = t_off[b2 + 1] - t_off[b2];
    return la == lb && std::equal(&touchers[t_off[a]], &touchers[t_off[a]] + la, &touchers[t_off[b2]]);
  };
  std::vector<uint32_t> sorder(ns);
  std::iota(sorder.begin(), sorder.end(), 0u);
  std::sort(sorder.begin(), sorder.end(), seq_less);
  std::vector<uint32_t> spos(N, NONE);  // global id -> position in the grouped shared list
  for (size_t i = 0; i < ns; ++i) spos[shared_ids[sorder[i]]] = (uint32_t)i;
  for (uint32_t b = 0; b < nb; ++b)
    std::sort(bd[b].begin() + nint[b], bd[b].end(), [&](uint32_t x, uint32_t y) { return spos[x] < spos[y]; });

  // ---- emit arrays in execution order
  P.cell_order.clear();
  P.cell_order.reserve(nc);
  P.batch_cell_off.assign(1, 0);
  P.batch_dof_off.assign(1, 0);
  P.bdofs.clear();
  P.bflags.clear();
  P.lmap.assign((size_t)nc * nd, 0);
  std::vector<uint8_t> touched(N, 0);
  std::vector<uint16_t> pos_in_batch(N, 0);
  P.max_batch_dofs = P.max_batch_cells = 0;
  P.n_first = P.n_add = 0;
  P.batch_nint.clear();
  P.halo_off.assign(1, 0);
  for (uint32_t k = 0; k < nb; ++k) {
    const uint32_t b = order[k];
    const std::vector<uint32_t> &v = bd[b];
    P.batch_nint.push_back(nint[b]);
    // (apply_planes3: every batch owns a fixed number of halo slots, so a slot index follows from the batch index)
    P.halo_off.push_back(P.halo_off.back() +
                         (interior_max && (!batch_masked[b] || masked_planes) ? limits->halo_stride
                                                                              : (uint32_t)(v.size() - nint[b])));
    for (uint32_t g : v) {
      uint8_t f = 0;
      if (constrained[g]) f |= kFlagConstrained;
      if (touched[g]) {
        f |= kFlagAdd;
        ++P.n_add;
      } else {
        touched[g] = 1;
        ++P.n_first;
      }
      P.bdofs.push_back(g | (constrained[g] ? 0x80000000u : 0u));  // bit 31: constrained row
      P.bflags.push_back(f);
    }
    for (size_t t = 0; t < v.size(); ++t) pos_in_batch[v[t]] = (uint16_t)t;  // valid for this batch's dofs only
    for (uint32_t c : batches[b]) {
      const size_t pos = P.cell_order.size();
      P.cell_order.push_back(c);
      for (uint32_t i = 0; i < nd; ++i) P.lmap[pos * nd + i] = pos_in_batch[l2g[(uint64_t)c * nd + i]];
    }
    P.batch_cell_off.push_back((uint32_t)P.cell_order.size());
    P.batch_dof_off.push_back((uint32_t)P.bdofs.size());
    P.max_batch_dofs = std::max<uint32_t>(P.max_batch_dofs, (uint32_t)v.size());
    P.max_batch_cells = std::max<uint32_t>(P.max_batch_cells, (uint32_t)batches[b].size());
  }
  P.n_plane_batches = P.n_plain_plane_batches = 0;
  if (interior_max) {
    for (uint32_t k = 0; k < nb; ++k) {
      P.n_plain_plane_batches += batch_masked[order[k]] ? 0u : 1u;
      P.n_plane_batches += (batch_masked[order[k]] && !masked_planes) ? 0u : 1u;
    }
    // masked_planes: the two kinds are interleaved, every plane batch runs in the <HN> instantiation
    if (masked_planes && P.n_plain_plane_batches < P.n_plane_batches) P.n_plain_plane_batches = 0;
  }
  P.orphans.clear();
  for (uint32_t g = 0; g < N; ++g)
    if (!touched[g]) P.orphans.push_back(g | (constrained[g] ? 0x80000000u : 0u));
  // ---- second pass.  sdofs: the grouped shared list.  (s_off, s_idx): CSR of the halo slots of every
  // shared dof in ascending execution order of the batches -- the generic form, used by the host-side
  // checks and by reduce_shared when the groups are too small to pay (irregular meshes).  (chunks,
  // gstarts): the grouped form -- a chunk is up to 64 consecutive dofs of one group; its k partial sums
  // per dof sit at gstarts[tstart + t] + offset + lane, t = 0..k-1.
  {
    P.sdofs.resize(ns);
    for (size_t i = 0; i < ns; ++i) {
      const uint32_t g = shared_ids[sorder[i]];
      P.sdofs[i] = g | (constrained[g] ? 0x80000000u : 0u);
    }
    P.s_off.assign(ns + 1, 0);
    for (size_t i = 0; i < ns; ++i) P.s_off[i + 1] = P.s_off[i] + ntouch[P.sdofs[i] & 0x7fffffffu];
    P.s_idx.assign(P.s_off[ns], 0);
    std::vector<uint32_t> fill(P.s_off.begin(), P.s_off.end() - 1);
    for (uint32_t k = 0; k < nb; ++k) {
      const uint32_t ni = P.batch_nint[k];
      const uint32_t d0 = P.batch_dof_off[k], d1 = P.batch_dof_off[k + 1];
      for (uint32_t t = d0 + ni; t < d1; ++t) {
        const uint32_t g = P.bdofs[t] & 0x7fffffffu;
        P.s_idx[fill[spos[g]]++] = P.halo_off[k] + (t - d0 - ni);
      }
    }
    P.chunks.clear();
    P.gstarts.clear();
    for (size_t i = 0; i < ns;) {
      size_t j = i + 1;
      while (j < ns && same_group(sorder[i], sorder[j])) ++j;
      const uint32_t k = P.s_off[i + 1] - P.s_off[i], tstart = (uint32_t)P.gstarts.size();
      for (uint32_t t = 0; t < k; ++t) P.gstarts.push_back(P.s_idx[P.s_off[i] + t]);  // first dof's slots
      for (size_t o = i; o < j; o += 64) {
        const uint32_t cnt = (uint32_t)std::min<size_t>(64, j - o);
        P.chunks.push_back((uint32_t)o);
        P.chunks.push_back(cnt | (k << 16));
        P.chunks.push_back(tstart);
        P.chunks.push_back((uint32_t)(o - i));
      }
      i = j;
    }
  }
  return 0;
}

// Fixed-size per-batch records of apply_planes3.  Dof list: p_ji slots of interior dofs, padded with a pass-2 dof
// of the batch (the padding lanes store the zero their never-touched accumulator slot holds, or old + 0, to a dof
// pass 2 rewrites), then p_hs slots of pass-2 dofs, padded with the last one.  Index runs: per task (cell c, plane
// k) the n*n slot numbers of the xy-plane z = k as byte offsets (slot * 8) packed two per word, stored
// [word][task] so that a wave reads consecutive words; the tasks of cells a ragged batch does not have point at
// the list's last slot, which is padding in every batch the planner accepts.
// (constraint_mask: the description's, caller's cell order; nullptr on conforming meshes)
int build_plane_records(Plan &P, const uint32_t *constraint_mask) {
  const int n = P.n, n2 = n * n, NT = p_cells_per_wave(n) * n, NIW = (n2 + 1) / 2;
  const int JI = p_ji(n) * 64, NB = p_kgu(n) * 64;
  const size_t nbat = P.n_plane_batches;
  const uint32_t dummy = 8u * (uint32_t)(NB - 1);
  std::vector<uint32_t> &bd = P.pr_dofs, &ix = P.pr_idx;
  bd.assign((size_t)NB * nbat, 0u);
  ix.assign((size_t)NIW * NT * nbat, dummy | (dummy << 16));
  std::vector<uint32_t> slot_of;  // batch-local id (position in P.bdofs) -> slot
  // batches of cells WITH a hanging-node mask (<HN> instantiation; anywhere behind the first n_plain_plane_batches): the
  // constrained nodes of a cell (those on a line one of the interpolation passes of hanging_nodes.cuh:617-696
  // touches) get PRIVATE positions behind the dof list; the cell's index runs point there.  Per batch: a copy list
  // (private position <- position of the node's dof in the list) and per direction the line operations, each the n
  // private positions of a line in the order the plain weight matrix applies to (hn_cell_lines).
  const int HROWS = p_hn_rows(n), CR = p_priv_max(n) / 64;
  std::vector<uint32_t> &hnrec = P.pr_hn;
  hnrec.clear();
  P.pr_hn_slot.assign(nbat, 0xffffffffu);
  std::vector<HnLine> lines[3];
  std::vector<uint16_t> pnodes;
  std::vector<uint32_t> priv_pos((size_t)P.nd);
  for (size_t b = 0; b < nbat; ++b) {
    const uint32_t c0 = P.batch_cell_off[b], nc = P.batch_cell_off[b + 1] - c0;
    const uint32_t d0 = P.batch_dof_off[b], nbd = P.batch_dof_off[b + 1] - d0, ni = P.batch_nint[b];
    if ((int)nc * n > NT || (int)ni > JI || (int)(nbd - ni) >= NB - JI || nbd == ni) {
      set_error("internal: batch does not fit the plane kernel's dof-list slots");
      return MFGPU_EINVAL;
    }
    slot_of.assign(nbd, 0u);
    for (uint32_t t = 0; t < nbd; ++t) slot_of[t] = t < ni ? t : (uint32_t)JI + (t - ni);
    for (int t = 0; t < NB; ++t) {
      uint32_t src_t;
      if (t < JI) src_t = (uint32_t)t < ni ? (uint32_t)t : ni;  // padding: the first pass-2 dof
      else src_t = std::min<uint32_t>(ni + (uint32_t)(t - JI), nbd - 1);
      bd[b * NB + t] = P.bdofs[d0 + src_t];
    }
    bool hnb = false;  // (a batch holds masked cells only or unmasked cells only)
    if (constraint_mask && b >= P.n_plain_plane_batches)
      for (uint32_t c = 0; c < nc; ++c) hnb = hnb || constraint_mask[P.cell_order[c0 + c]] != 0;
    uint32_t next_priv = (uint32_t)NB;
    std::vector<uint32_t> copies, ops_d[3];
    for (uint32_t c = 0; c < nc; ++c) {
      const unsigned mask = hnb ? constraint_mask[P.cell_order[c0 + c]] : 0u;
      std::fill(priv_pos.begin(), priv_pos.end(), 0xffffffffu);
      if (mask) {
        hn_cell_lines(mask, n, lines, pnodes);
        for (uint16_t node : pnodes) {
          if (next_priv >= (uint32_t)NB + (uint32_t)p_priv_max(n)) {  // (the planner budgets them)
            set_error("internal: batch exceeds the plane kernel's private hanging-node entries");
            return MFGPU_EINVAL;
          }
          priv_pos[node] = next_priv;
          copies.push_back((next_priv << 16) | slot_of[P.lmap[(size_t)(c0 + c) * P.nd + node]]);
          ++next_priv;
        }
        for (int dir = 0; dir < 3; ++dir)
          for (const HnLine &L : lines[dir]) {
            uint32_t w[3] = {0u, 0u, 0u};
            for (int t = 0; t < n; ++t) w[t >> 1] |= priv_pos[L.node[t]] << (16 * (t & 1));
            ops_d[dir].insert(ops_d[dir].end(), w, w + 3);
          }
      }
      for (int k = 0; k < n; ++k)
        for (int i = 0; i < n2; ++i) {
          const int node = i + n2 * k;
          const uint32_t pos = priv_pos[node] != 0xffffffffu ? priv_pos[node] : slot_of[P.lmap[(size_t)(c0 + c) * P.nd + node]];
          const uint32_t off = 8u * pos;
          uint32_t &w = ix[(b * NIW + i / 2) * NT + c * n + k];
          w = (i & 1) ? ((w & 0xffffu) | (off << 16)) : ((w & 0xffff0000u) | off);
        }
    }
    if (hnb) {
      P.pr_hn_slot[b] = (uint32_t)(hnrec.size() / ((size_t)HROWS * 64));
      hnrec.resize(hnrec.size() + (size_t)HROWS * 64, 0u);
      uint32_t *rec = hnrec.data() + (size_t)P.pr_hn_slot[b] * HROWS * 64;
      for (size_t e = 0; e < copies.size(); ++e) rec[e] = copies[e];  // rows 0 .. CR-1, entry e at [e / 64][e % 64]
      for (int dir = 0; dir < 3; ++dir) {
        const size_t nops = ops_d[dir].size() / 3;
        if (nops > (size_t)kHnOpRounds * 64) {
          set_error("internal: more hanging-node lines in a batch than the plane kernel's record holds");
          return MFGPU_EINVAL;
        }
        for (size_t e = 0; e < nops; ++e)
          for (int w = 0; w < 3; ++w)
            rec[(size_t)(CR + (dir * kHnOpRounds + (int)(e / 64)) * 3 + w) * 64 + e % 64] = ops_d[dir][e * 3 + w];
      }
      const uint32_t h0 = (uint32_t)copies.size() | ((uint32_t)(ops_d[0].size() / 3) << 16);
      const uint32_t h1 = (uint32_t)(ops_d[1].size() / 3) | ((uint32_t)(ops_d[2].size() / 3) << 16);
      for (int l = 0; l < 64; ++l) {
        rec[(size_t)(HROWS - 2) * 64 + l] = h0;
        rec[(size_t)(HROWS - 1) * 64 + l] = h1;
      }
    }
  }
  return 0;
}

int choose_kernel_and_plan(const mfgpu_desc &d, KernelChoice &kc, Plan &plan) {
  const bool general = !(d.flags & MFGPU_UNIFORM_J0), hn = (d.flags & MFGPU_HANGING_NODES) != 0;
  const bool colored = (d.flags & MFGPU_COLORED_SCATTER) != 0;
  if (d.kernel > MFGPU_KERNEL_PLANES_2W) {
    set_error("unknown mfgpu_desc.kernel");
    return MFGPU_EINVAL;
  }
  // (with hanging nodes: the cells without a mask run in the plane kernel, the masked ones in apply_batches_x)
  const bool pk_ok = d.dim == 3 && !general && !colored && d.degree >= 2 && d.degree <= 6 &&
                     d.n_dofs < (1u << 29);  // (vectors are addressed base + 32-bit byte offset)
  const bool xk_ok = d.dim == 3 && !general && !colored;
  const bool want_planes = d.kernel == MFGPU_KERNEL_PLANES || d.kernel == MFGPU_KERNEL_PLANES_2W;
  if ((want_planes && !pk_ok) || (d.kernel == MFGPU_KERNEL_PENCILS_X && !xk_ok) ||
      (d.kernel == MFGPU_KERNEL_PENCILS && general)) {
    set_error("mfgpu_desc.kernel: this kernel family does not cover the description (see include/mfgpu.h)");
    return MFGPU_EUNSUPPORTED;
  }
  kc.general = general;
  // by default the plane kernel serves p = 4 only: at p = 2, 3 the pencil kernel measures faster
  // (on meshes with hanging nodes also p = 3: 0.174 instead of 0.256 ms on the bmop ADAPTIVE_GRID mesh, n_ref = 6)
  // p = 5, 6: apply_planes4 with one wave per SIMD (apply_planes3's two transpose arrays do not fit the LDS there)
  // p = 3: apply_planes4 with two waves per SIMD (16 cells per wave) measures 9 % faster than the pencil kernel per
  // vmult (0.222 vs 0.243 ms at 10^7 dofs; profiles/r03_notes.md); p = 2: the pencil kernel stays ahead
  // (in float the pencil kernel is ahead at p = 3 on conforming meshes: 0.152 vs 0.179 ms)
  kc.planes = pk_ok && (want_planes || (d.kernel == MFGPU_KERNEL_AUTO &&
                                        (d.degree >= 4 || (d.degree == 3 && (hn || d.number_type == MFGPU_F64)))));
  kc.pencils_x = xk_ok && !kc.planes && d.kernel != MFGPU_KERNEL_PENCILS;
  PlanLimits lim;
  if (kc.planes) {
    lim.max_cells = (uint32_t)p_cells_per_wave(d.degree + 1);
    lim.max_dofs = (uint32_t)p_kgu(d.degree + 1) * 64u - 1u;
    lim.interior_max = (uint32_t)p_ji(d.degree + 1) * 64u;
    lim.halo_stride = (uint32_t)p_hs(d.degree + 1) * 64u;
    lim.shared_max = lim.halo_stride - 1u;  // the list's last slot stays padding (idle tasks)
    lim.segregate_masked = hn;
    // cells with a hanging-node mask run in the plane kernel too (apply_planes3<HN>), in batches of their own
    lim.masked_planes = hn && d.degree <= 4;  // (p = 5, 6: masked cells stay in the pencil kernel)
    lim.private_max = (uint32_t)p_priv_max(d.degree + 1);
  }
  // apply_batches_x unrolls 4 chunks at p=3 (64-cell batches = 13^3 dofs like p=4); everything else 3
  int rc = build_plan(d, plan, (kc.pencils_x && d.degree == 3) ? 4u : 3u, kc.planes ? &lim : nullptr);
  if (rc == MFGPU_EUNSUPPORTED && kc.planes && d.kernel == MFGPU_KERNEL_AUTO && xk_ok) {
    // a cell with more surface dofs than the plane kernel's dof-list slots hold (cannot happen on conforming
    // hexahedral meshes): the pencil kernel has no such limit
    kc.planes = false;
    kc.pencils_x = true;
    plan = Plan();
    rc = build_plan(d, plan, d.degree == 3 ? 4u : 3u, nullptr);
  }
  if (!rc && kc.planes) {
    const uint32_t nbat = (uint32_t)plan.batch_cell_off.size() - 1;
    kc.pencils_x = plan.n_plane_batches < nbat;  // the masked cells' batches
    if (plan.n_plane_batches == 0) kc.planes = false;
  }
  return rc;
}

}  // namespace mfgpu

extern "C" {

const char *mfgpu_last_error(void) { return mfgpu::last_error(); }

int mfgpu_plan_create(const mfgpu_desc *desc, mfgpu_plan **out) {
  if (!desc || !out) {
    mfgpu::set_error("null argument");
    return MFGPU_EINVAL;
  }
  mfgpu_plan *p = new mfgpu_plan();
  mfgpu::KernelChoice kc;
  int rc = mfgpu::choose_kernel_and_plan(*desc, kc, p->plan);
  if (!rc && kc.planes) rc = mfgpu::build_plane_records(p->plan, desc->constraint_mask);
  if (rc) {
    delete p;
    return rc;
  }
  *out = p;
  return 0;
}

void mfgpu_plan_destroy(mfgpu_plan *p) { delete p; }

int mfgpu_suggest_renumbering(const mfgpu_desc *desc, uint32_t *new_index) {
  if (!desc || !new_index) {
    mfgpu::set_error("null argument");
    return MFGPU_EINVAL;
  }
  mfgpu::Plan P;
  mfgpu::KernelChoice kc;
  int rc = mfgpu::choose_kernel_and_plan(*desc, kc, P);
  if (rc) return rc;
  // batch-major: the dofs a batch owns alone, batch after batch (one contiguous run per batch: coalesced gathers and
  // stores of the cell loop), then the dofs several batches share in the order pass 2 walks them (grouped by the set
  // of batches: whole lines for pass 2 instead of isolated entries of a lexicographic numbering), then the dofs no
  // cell touches
  const uint32_t NONE = 0xffffffffu;
  std::fill(new_index, new_index + desc->n_dofs, NONE);
  uint32_t next = 0;
  const size_t nb = P.batch_cell_off.size() - 1;
  for (size_t b = 0; b < nb; ++b) {
    const uint32_t d0 = P.batch_dof_off[b];
    // coloured mode has no interior / shared split: a dof goes with the first batch that lists it
    const uint32_t ni = P.batch_nint.empty() ? P.batch_dof_off[b + 1] - d0 : P.batch_nint[b];
    for (uint32_t t = 0; t < ni; ++t) {
      const uint32_t g = P.bdofs[d0 + t] & 0x7fffffffu;
      if (new_index[g] == NONE) new_index[g] = next++;
    }
  }
  for (uint32_t e : P.sdofs) {
    const uint32_t g = e & 0x7fffffffu;
    if (new_index[g] == NONE) new_index[g] = next++;
  }
  for (uint32_t g = 0; g < desc->n_dofs; ++g)
    if (new_index[g] == NONE) new_index[g] = next++;
  if (next != desc->n_dofs) {
    mfgpu::set_error("internal: renumbering is not a permutation");
    return MFGPU_EINVAL;
  }
  return 0;
}

int64_t mfgpu_plan_array_u32(const mfgpu_plan *p, int what, const uint32_t **ptr) {
  if (!p || !ptr) return MFGPU_EINVAL;
  const std::vector<uint32_t> *v = nullptr;
  switch (what) {
    case 0: v = &p->plan.batch_cell_off; break;
    case 1: v = &p->plan.batch_dof_off; break;
    case 2: v = &p->plan.color_batch_off; break;
    case 3: v = &p->plan.cell_order; break;
    case 4: v = &p->plan.bdofs; break;
    case 5: v = &p->plan.orphans; break;
    case 6: v = &p->plan.batch_nint; break;
    case 7: v = &p->plan.halo_off; break;
    case 8: v = &p->plan.sdofs; break;
    case 9: v = &p->plan.s_off; break;
    case 10: v = &p->plan.s_idx; break;
    case 11: v = &p->plan.chunks; break;
    case 12: v = &p->plan.gstarts; break;
    case 13: v = &p->plan.pr_dofs; break;
    case 14: v = &p->plan.pr_idx; break;
    case 15: v = &p->plan.pr_hn; break;
    case 16: v = &p->plan.pr_hn_slot; break;
    default: mfgpu::set_error("bad array id"); return MFGPU_EINVAL;
  }
  *ptr = v->data();
  return (int64_t)v->size();
}

int64_t mfgpu_plan_lmap(const mfgpu_plan *p, const uint16_t **ptr) {
  if (!p || !ptr) return MFGPU_EINVAL;
  *ptr = p->plan.lmap.data();
  return (int64_t)p->plan.lmap.size();
}

int64_t mfgpu_plan_bflags(const mfgpu_plan *p, const uint8_t **ptr) {
  if (!p || !ptr) return MFGPU_EINVAL;
  *ptr = p->plan.bflags.data();
  return (int64_t)p->plan.bflags.size();
}

}  // extern "C"
