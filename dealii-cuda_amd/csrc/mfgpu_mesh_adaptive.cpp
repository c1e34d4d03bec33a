// Adaptive mesh stand-in (host only): single-root octree/quadtree on hyper_cube(-1,1) with the
// refinement recipe of bmop's ADAPTIVE_GRID (reference bmop_common.h:49-105), one-irregular
// (2:1 over faces and, in 3D, edges, as deal.II's execute_coarsening_and_refinement enforces),
// FE_Q(p) dofs with separate hanging-node dofs, and the per-cell hanging-node data the GPU path
// needs: constraint mask (hanging_nodes.cuh:23-53) and loc2glob with constrained face / edge entries
// replaced by the coarse neighbour's dofs (HangingNodes::setup_constraints, :209-454).
//
// On a single-root octree every face orientation is standard, so the rotation / flip branches of the
// reference (:285-302, :457-578) never trigger; the mask and substitution logic is the same.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <unordered_set>

#include "mfgpu_mesh.h"

namespace mfgpu {
namespace {

struct Cell {
  int level;
  uint32_t c[3];
};

inline uint64_t ckey(int level, const uint32_t *c) {
  return ((uint64_t)level << 57) | ((uint64_t)c[2] << 38) | ((uint64_t)c[1] << 19) | (uint64_t)c[0];
}

struct Tree {
  int dim;
  std::unordered_set<uint64_t> active;  // leaves
  int max_level = 0;
  bool balance_vertices = false;  // Triangulation::limit_level_difference_at_vertices (poisson_mg.cu:131)

  bool is_active(int l, const uint32_t *c) const { return active.count(ckey(l, c)) != 0; }

  // level of the leaf covering cell (l, c) or any ancestor of it; -1 if (l,c) is refined further
  // (i.e. the covering leaves are finer) or outside the domain.
  int covering_level(int l, const int64_t *ci) const {
    uint32_t c[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d) {
      if (ci[d] < 0 || ci[d] >= ((int64_t)1 << l)) return -2;  // outside
      c[d] = (uint32_t)ci[d];
    }
    for (int ll = l; ll >= 0; --ll) {
      if (is_active(ll, c)) return ll;
      for (int d = 0; d < dim; ++d) c[d] >>= 1;
    }
    return -1;  // finer leaves
  }

  void refine(const Cell &cell) {
    active.erase(ckey(cell.level, cell.c));
    for (int k = 0; k < (1 << dim); ++k) {
      uint32_t c[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) c[d] = 2 * cell.c[d] + ((k >> d) & 1);
      active.insert(ckey(cell.level + 1, c));
    }
    max_level = std::max(max_level, cell.level + 1);
  }

  std::vector<Cell> leaves() const {
    std::vector<Cell> v;
    v.reserve(active.size());
    for (uint64_t k : active) {
      Cell c;
      c.level = (int)(k >> 57);
      c.c[0] = (uint32_t)(k & 0x7ffff);
      c.c[1] = (uint32_t)((k >> 19) & 0x7ffff);
      c.c[2] = (uint32_t)((k >> 38) & 0x7ffff);
      v.push_back(c);
    }
    return v;
  }

  // refine the flagged leaves and whatever else is needed to stay one-irregular over faces and
  // (3D) edges: a leaf next to a flagged leaf of HIGHER level is flagged too, until stable.
  void refine_flagged(std::vector<Cell> flagged) {
    std::unordered_set<uint64_t> flag;
    for (const Cell &c : flagged) flag.insert(ckey(c.level, c.c));
    std::vector<Cell> work = flagged;
    while (!work.empty()) {
      std::vector<Cell> next;
      for (const Cell &c : work) {
        // neighbours over faces and edges at the cell's own level
        for (int ox = -1; ox <= 1; ++ox)
          for (int oy = -1; oy <= 1; ++oy)
            for (int oz = (dim == 3 ? -1 : 0); oz <= (dim == 3 ? 1 : 0); ++oz) {
              const int nz = (ox != 0) + (oy != 0) + (oz != 0);
              if (nz == 0) continue;
              if (!balance_vertices) {
                if (nz > 2) continue;               // faces (1) and edges (2); not vertices
                if (dim == 2 && nz == 2) continue;  // 2D: faces only
              }
              int64_t ci[3] = {(int64_t)c.c[0] + ox, (int64_t)c.c[1] + oy, (int64_t)c.c[2] + oz};
              const int lv = covering_level(c.level, ci);
              if (lv >= 0 && lv < c.level) {
                // the covering leaf is coarser: it must be refined as well
                Cell nb;
                nb.level = lv;
                for (int d = 0; d < 3; ++d) nb.c[d] = d < dim ? (uint32_t)(ci[d] >> (c.level - lv)) : 0u;
                if (flag.insert(ckey(nb.level, nb.c)).second) next.push_back(nb);
              }
            }
      }
      work.swap(next);
    }
    // refine coarse-to-fine so that parents exist as leaves when their turn comes
    std::vector<Cell> all;
    for (uint64_t k : flag) {
      Cell c;
      c.level = (int)(k >> 57);
      c.c[0] = (uint32_t)(k & 0x7ffff);
      c.c[1] = (uint32_t)((k >> 19) & 0x7ffff);
      c.c[2] = (uint32_t)((k >> 38) & 0x7ffff);
      all.push_back(c);
    }
    std::sort(all.begin(), all.end(), [](const Cell &a, const Cell &b) { return a.level < b.level; });
    for (const Cell &c : all)
      if (is_active(c.level, c.c)) refine(c);
  }
};

struct DofKey {
  int level;
  uint32_t k[3];
  uint8_t i[3];
  bool operator==(const DofKey &o) const {
    return level == o.level && k[0] == o.k[0] && k[1] == o.k[1] && k[2] == o.k[2] && i[0] == o.i[0] &&
           i[1] == o.i[1] && i[2] == o.i[2];
  }
};
struct DofKeyHash {
  size_t operator()(const DofKey &a) const {
    uint64_t h = (uint64_t)a.level * 0x9E3779B97F4A7C15ull;
    for (int d = 0; d < 3; ++d) {
      h ^= ((uint64_t)a.k[d] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2));
      h ^= ((uint64_t)a.i[d] + 0x517cc1b727220a95ull + (h << 6) + (h >> 2));
    }
    return (size_t)h;
  }
};

// Structural identity of the node (i0,i1,i2) of a cell: per direction the lattice coordinate of the
// entity at the cell's level (high boundary -> next cell's low boundary); pure vertices are moved to
// the coarsest level on which they are lattice points.  Nodes of a fine edge / face lying on a coarse
// edge / face keep the fine level: they are the separate hanging-node dofs deal.II creates.
DofKey dof_key(int dim, int p, const Cell &c, const int *idx) {
  DofKey key;
  key.level = c.level;
  bool vertex = true;
  for (int d = 0; d < 3; ++d) {
    if (d >= dim) {
      key.k[d] = 0;
      key.i[d] = 0;
      continue;
    }
    if (idx[d] == p) {
      key.k[d] = c.c[d] + 1;
      key.i[d] = 0;
    } else {
      key.k[d] = c.c[d];
      key.i[d] = (uint8_t)idx[d];
      if (idx[d] != 0) vertex = false;
    }
  }
  if (vertex) {
    while (key.level > 0) {
      bool even = true;
      for (int d = 0; d < dim; ++d) even = even && (key.k[d] % 2 == 0);
      if (!even) break;
      for (int d = 0; d < dim; ++d) key.k[d] >>= 1;
      key.level--;
    }
  }
  return key;
}

}  // namespace

int build_adaptive(Mesh &M, int n_ref, bool balance_vertices) {
  const int dim = M.dim, p = M.degree, n = p + 1, nd = ipow(n, dim);
  if (n_ref < 0 || n_ref > 12) {
    set_error("mfgpu_mesh_create_adaptive: n_ref out of range");
    return MFGPU_EINVAL;
  }
  M.init_tables();
  Tree T;
  T.dim = dim;
  T.balance_vertices = balance_vertices;
  {
    uint32_t c0[3] = {0, 0, 0};
    T.active.insert(ckey(0, c0));
  }
  // pseudo_adaptive_refinement (bmop_common.h:49-105), CUBE domain
  const int nglob = std::max(n_ref - 2, 0);
  for (int r = 0; r < nglob; ++r) T.refine_flagged(T.leaves());
  const double reduction = dim == 2 ? 0.005 : 0.015;
  auto center = [&](const Cell &c, double *x) {
    const double h = 2.0 / (double)(1u << c.level);
    for (int d = 0; d < dim; ++d) x[d] = -1.0 + h * (c.c[d] + 0.5);
  };
  auto annulus = [&](double R, double r, const double *ctr) {
    std::vector<Cell> fl;
    for (const Cell &c : T.leaves()) {
      double x[3], s = 0;
      center(c, x);
      for (int d = 0; d < dim; ++d) s += (x[d] - ctr[d]) * (x[d] - ctr[d]);
      const double dist = std::sqrt(s);
      if (dist > r && dist < R) fl.push_back(c);
    }
    T.refine_flagged(fl);
  };
  const double origin[3] = {0, 0, 0};
  double offset[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) offset[d] = -0.1 * (d + 1);
  annulus(0.55 - reduction, 0.0, origin);
  annulus(0.42 - reduction, 0.3 + reduction, origin);
  annulus(0.41 - reduction, 0.32 + reduction, origin);
  annulus(0.33 - reduction, 0.17 + reduction, offset);
  annulus(0.31 - reduction, 0.21 + reduction, offset);
  if (dim == 2) {
    for (int s = 0; s < 4; ++s) {  // mark_cells_on_shell (bmop_common.h:27-47,98-104)
      std::vector<Cell> fl;
      for (const Cell &c : T.leaves()) {
        const double h = 2.0 / (double)(1u << c.level);
        int inside = 0;
        for (int v = 0; v < 4; ++v) {
          double s2 = 0;
          for (int d = 0; d < 2; ++d) {
            const double x = -1.0 + h * (c.c[d] + ((v >> d) & 1)) - offset[d];
            s2 += x * x;
          }
          inside += std::sqrt(s2) < 0.25;
        }
        if (inside != 0 && inside != 4) fl.push_back(c);
      }
      T.refine_flagged(fl);
    }
  }
  std::vector<std::array<uint32_t, 4>> v;
  for (const Cell &c : T.leaves()) v.push_back({(uint32_t)c.level, c.c[0], c.c[1], c.c[2]});
  (void)n;
  (void)nd;
  return build_from_tree_leaves(M, dim, v);
}

// Builds every array of the mesh from a one-irregular set of leaves (level, cx, cy, cz).
int build_from_tree_leaves(Mesh &M, int dim, std::vector<std::array<uint32_t, 4>> leaves_in) {
  const int p = M.degree, n = p + 1, nd = ipow(n, dim);
  if (M.nodes.empty()) M.init_tables();
  Tree T;
  T.dim = dim;
  for (auto &l : leaves_in) {
    uint32_t c[3] = {l[1], l[2], l[3]};
    T.active.insert(ckey((int)l[0], c));
    T.max_level = std::max(T.max_level, (int)l[0]);
  }
  // Morton (depth-first) order: consecutive cells are close in space, which is what the batch
  // planner's locality relies on (deal.II orders children of a cell consecutively as well)
  std::vector<Cell> cells = T.leaves();
  const int L = T.max_level;
  auto morton = [&](const Cell &c) {
    uint64_t m = 0;
    for (int b = L - 1; b >= 0; --b)
      for (int d = dim - 1; d >= 0; --d) {
        const uint32_t cc = c.c[d] << (L - c.level);
        m = (m << 1) | ((cc >> b) & 1);
      }
    return m;
  };
  std::sort(cells.begin(), cells.end(), [&](const Cell &a, const Cell &b) {
    const uint64_t ma = morton(a), mb = morton(b);
    return ma != mb ? ma < mb : a.level < b.level;
  });
  const uint32_t nc = (uint32_t)cells.size();
  std::unordered_map<uint64_t, uint32_t> cell_id;
  for (uint32_t i = 0; i < nc; ++i) cell_id[ckey(cells[i].level, cells[i].c)] = i;

  // ---- dofs (structural keys) and unsubstituted loc2glob
  std::unordered_map<DofKey, uint32_t, DofKeyHash> dof_id;
  std::vector<uint32_t> l2g((size_t)nc * nd);
  std::vector<double> coords;
  std::vector<uint8_t> on_boundary;
  for (uint32_t ci = 0; ci < nc; ++ci) {
    const Cell &c = cells[ci];
    const double h = 2.0 / (double)(1u << c.level);
    for (int i = 0; i < nd; ++i) {
      int idx[3] = {i % n, (i / n) % n, dim == 3 ? i / (n * n) : 0};
      DofKey key = dof_key(dim, p, c, idx);
      auto it = dof_id.find(key);
      uint32_t g;
      if (it == dof_id.end()) {
        g = (uint32_t)dof_id.size();
        dof_id.emplace(key, g);
        bool onb = false;
        for (int d = 0; d < dim; ++d) {
          const double x = -1.0 + h * (c.c[d] + M.nodes[idx[d]]);
          coords.push_back(x);
          const uint32_t last = (1u << c.level) - 1;
          if ((idx[d] == 0 && c.c[d] == 0) || (idx[d] == p && c.c[d] == last)) onb = true;
        }
        on_boundary.push_back(onb);
      } else {
        g = it->second;
      }
      l2g[(size_t)ci * nd + i] = g;
    }
  }
  const uint32_t N = (uint32_t)dof_id.size();

  // ---- hanging-node constraints per cell (HangingNodes::setup_constraints)
  const uint32_t TYPE[3] = {1u << 0, 1u << 1, 1u << 2};
  const uint32_t FACE[3] = {1u << 3, 1u << 4, 1u << 5};
  // edge along direction e: perpendicular directions (e+1)%3, (e+2)%3; mask bit as the kernel's
  // interpolate_boundary_3d expects it for interpolation direction e (hanging_nodes.cuh:636-637)
  const uint32_t EDGE_DIR[3] = {1u << 7 /*YZ: along x*/, 1u << 8 /*ZX: along y*/, 1u << 6 /*XY: along z*/};
  std::vector<uint32_t> sub(l2g);  // substituted loc2glob
  std::vector<uint32_t> mask(nc, 0);
  std::vector<uint8_t> hanging(N, 0);
  bool any_mask = false;
  auto find_cell = [&](int level, const int64_t *ci, const Cell *&out) -> bool {
    uint32_t c[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d) c[d] = (uint32_t)ci[d];
    auto it = cell_id.find(ckey(level, c));
    if (it == cell_id.end()) return false;
    out = &cells[it->second];
    return true;
  };
  for (uint32_t ci = 0; ci < nc; ++ci) {
    const Cell &c = cells[ci];
    if (c.level == 0) continue;
    uint32_t m = 0;
    bool face_con[3] = {false, false, false};
    // faces
    for (int d = 0; d < dim; ++d)
      for (int side = 0; side < 2; ++side) {
        int64_t nbc[3] = {c.c[0], c.c[1], c.c[2]};
        nbc[d] += side ? 1 : -1;
        const int lv = T.covering_level(c.level, nbc);
        if (lv < 0 || lv >= c.level) continue;  // boundary, same level or finer
        // coarse neighbour (exactly one level coarser in a one-irregular mesh)
        int64_t pc[3] = {nbc[0] >> 1, nbc[1] >> 1, nbc[2] >> 1};
        const Cell *nb = nullptr;
        if (lv != c.level - 1 || !find_cell(lv, pc, nb)) {
          set_error("internal: mesh is not one-irregular over faces");
          return MFGPU_EINVAL;
        }
        const uint32_t nbi = cell_id[ckey(nb->level, nb->c)];
        face_con[d] = true;
        m |= FACE[d];
        if (side == 0) m |= TYPE[d];  // constrained face is the low one (hanging_nodes.cuh:307-308)
        // tangential type bits: set when this cell is the first child along that direction
        // (subface index parity, :309-312,319-322,329-332)
        for (int t = 0; t < dim; ++t)
          if (t != d && (c.c[t] % 2 == 0)) m |= TYPE[t];
        // copy the coarse neighbour's dofs of the opposite face onto this face's entries (:336-353)
        const int fi = side ? p : 0;      // this cell's face index along d
        const int nfi = side ? 0 : p;     // neighbour's opposite face
        for (int a = 0; a < n; ++a)
          for (int b2 = 0; b2 < (dim == 3 ? n : 1); ++b2) {
            int idx[3], nidx[3];
            int t1 = (d + 1) % dim, t2 = dim == 3 ? (d + 2) % 3 : -1;
            if (dim == 2) t1 = 1 - d;
            idx[d] = fi;
            nidx[d] = nfi;
            idx[t1] = nidx[t1] = a;
            if (dim == 3) idx[t2] = nidx[t2] = b2;
            const int li = idx[0] + n * idx[1] + (dim == 3 ? n * n * idx[2] : 0);
            const int ni = nidx[0] + n * nidx[1] + (dim == 3 ? n * n * nidx[2] : 0);
            const uint32_t oldg = l2g[(size_t)ci * nd + li];
            const uint32_t newg = l2g[(size_t)nbi * nd + ni];
            sub[(size_t)ci * nd + li] = newg;
            if (oldg != newg) hanging[oldg] = 1;
          }
      }
    // edges (3D): an edge none of whose two faces is constrained, with a coarser cell diagonally
    // across it (:364-453)
    if (dim == 3)
      for (int e = 0; e < 3; ++e) {
        const int d1 = (e + 1) % 3, d2 = (e + 2) % 3;
        if (face_con[d1] || face_con[d2]) {
          // edges on a constrained face are covered by the face; the remaining edges of this cell
          // along e lie on the two unconstrained faces and are checked below
        }
        for (int s1 = 0; s1 < 2; ++s1)
          for (int s2 = 0; s2 < 2; ++s2) {
            // the edge belongs to face d1 (side s1) and face d2 (side s2) of this cell
            const bool f1 = (m & FACE[d1]) && (((m & TYPE[d1]) != 0) == (s1 == 0));
            const bool f2 = (m & FACE[d2]) && (((m & TYPE[d2]) != 0) == (s2 == 0));
            if (f1 || f2) continue;
            int64_t nbc[3] = {c.c[0], c.c[1], c.c[2]};
            nbc[d1] += s1 ? 1 : -1;
            nbc[d2] += s2 ? 1 : -1;
            const int lv = T.covering_level(c.level, nbc);
            if (lv < 0 || lv >= c.level) continue;
            int64_t pc[3] = {nbc[0] >> 1, nbc[1] >> 1, nbc[2] >> 1};
            const Cell *nb = nullptr;
            if (lv != c.level - 1 || !find_cell(lv, pc, nb)) {
              set_error("internal: mesh is not one-irregular over edges");
              return MFGPU_EINVAL;
            }
            const uint32_t nbi = cell_id[ckey(nb->level, nb->c)];
            m |= EDGE_DIR[e];
            if (s1 == 0) m |= TYPE[d1];
            if (s2 == 0) m |= TYPE[d2];
            if (c.c[e] % 2 == 0) m |= TYPE[e];  // first child along the edge
            for (int a = 0; a < n; ++a) {
              int idx[3], nidx[3];
              idx[e] = nidx[e] = a;
              idx[d1] = s1 ? p : 0;
              idx[d2] = s2 ? p : 0;
              nidx[d1] = s1 ? 0 : p;  // the coarse cell's edge facing this cell
              nidx[d2] = s2 ? 0 : p;
              const int li = idx[0] + n * idx[1] + n * n * idx[2];
              const int ni = nidx[0] + n * nidx[1] + n * n * nidx[2];
              const uint32_t oldg = l2g[(size_t)ci * nd + li];
              const uint32_t newg = l2g[(size_t)nbi * nd + ni];
              sub[(size_t)ci * nd + li] = newg;
              if (oldg != newg) hanging[oldg] = 1;
            }
          }
      }
    mask[ci] = m;
    any_mask = any_mask || m != 0;
  }

  // ---- fill the mesh
  M.n_cells = nc;
  M.n_dofs = N;
  M.loc2glob = sub;
  M.constraint_mask.clear();
  if (any_mask) M.constraint_mask = mask;
  M.dof_coords = coords;
  M.constrained.clear();
  for (uint32_t g = 0; g < N; ++g)
    if (on_boundary[g] || hanging[g]) M.constrained.push_back(g);  // Dirichlet AND hanging (bmop.cu:118-124)
  M.JxW.resize((size_t)nc * nd);
  M.inv_jac.resize(nc);
  M.qpoints.resize((size_t)nc * nd * dim);
  for (uint32_t ci = 0; ci < nc; ++ci) {
    const Cell &c = cells[ci];
    const double h = 2.0 / (double)(1u << c.level);
    M.inv_jac[ci] = 1.0 / h;
    for (int q = 0; q < nd; ++q) {
      int qq = q;
      double w = 1.0;
      for (int d = 0; d < dim; ++d) {
        const int qi = qq % n;
        qq /= n;
        w *= M.wq[qi] * h;
        M.qpoints[((size_t)ci * nd + q) * dim + d] = -1.0 + h * (c.c[d] + M.xq[qi]);
      }
      M.JxW[(size_t)ci * nd + q] = w;
    }
  }
  M.cell_levels.resize((size_t)nc * 4);
  for (uint32_t ci = 0; ci < nc; ++ci) {
    M.cell_levels[4 * ci] = (uint32_t)cells[ci].level;
    for (int d = 0; d < 3; ++d) M.cell_levels[4 * ci + 1 + d] = cells[ci].c[d];
  }
  M.iface[0].clear();
  M.iface[1].clear();
  M.finalize_typed();
  return 0;
}

}  // namespace mfgpu

extern "C" {

int mfgpu_mesh_create_adaptive(int dim, int degree, int n_ref, int number_type, mfgpu_mesh **out) {
  if (!out || (dim != 2 && dim != 3) || degree < 1 || degree > 6 ||
      (number_type != MFGPU_F64 && number_type != MFGPU_F32)) {
    mfgpu::set_error("mfgpu_mesh_create_adaptive: bad argument");
    return MFGPU_EINVAL;
  }
  mfgpu_mesh *m = new mfgpu_mesh();
  m->mesh.dim = dim;
  m->mesh.degree = degree;
  m->mesh.number_type = number_type;
  int rc = mfgpu::build_adaptive(m->mesh, n_ref, false);
  if (rc) {
    delete m;
    return rc;
  }
  *out = m;
  return 0;
}

int mfgpu_mesh_create_adaptive_mg(int dim, int degree, int n_ref, int number_type, mfgpu_mesh **out) {
  if (!out || (dim != 2 && dim != 3) || degree < 1 || degree > 6 ||
      (number_type != MFGPU_F64 && number_type != MFGPU_F32)) {
    mfgpu::set_error("mfgpu_mesh_create_adaptive_mg: bad argument");
    return MFGPU_EINVAL;
  }
  mfgpu_mesh *m = new mfgpu_mesh();
  m->mesh.dim = dim;
  m->mesh.degree = degree;
  m->mesh.number_type = number_type;
  int rc = mfgpu::build_adaptive(m->mesh, n_ref, true);
  if (rc) {
    delete m;
    return rc;
  }
  *out = m;
  return 0;
}

int mfgpu_mesh_create_from_leaves(int dim, int degree, const uint32_t *leaves, uint32_t n_leaves,
                                  int number_type, mfgpu_mesh **out) {
  if (!out || !leaves || n_leaves == 0 || (dim != 2 && dim != 3) || degree < 1 || degree > 6) {
    mfgpu::set_error("mfgpu_mesh_create_from_leaves: bad argument");
    return MFGPU_EINVAL;
  }
  mfgpu_mesh *m = new mfgpu_mesh();
  m->mesh.dim = dim;
  m->mesh.degree = degree;
  m->mesh.number_type = number_type;
  std::vector<std::array<uint32_t, 4>> v(n_leaves);
  for (uint32_t i = 0; i < n_leaves; ++i) v[i] = {leaves[4 * i], leaves[4 * i + 1], leaves[4 * i + 2], leaves[4 * i + 3]};
  int rc = mfgpu::build_from_tree_leaves(m->mesh, dim, v);
  if (rc) {
    delete m;
    return rc;
  }
  *out = m;
  return 0;
}

int64_t mfgpu_mesh_cell_levels(const mfgpu_mesh *m, const uint32_t **ptr) {
  if (!m || !ptr) return MFGPU_EINVAL;
  *ptr = m->mesh.cell_levels.data();
  return (int64_t)m->mesh.cell_levels.size();
}

}  // extern "C"
