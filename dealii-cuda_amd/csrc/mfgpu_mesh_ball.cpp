// deal.II stand-in for the BALL domain of bmop / poisson (reference poisson_common.h:65-70, bmop_common.h:108-120 with
// -DBALL_GRID): GridGenerator::hyper_ball (unit ball: a central cube + 2*dim caps, 5 cells in 2D, 7 in 3D), a
// SphericalManifold on the BOUNDARY only, refine_global(n_ref), MappingQ1 (matrix_free_gpu.h:257), FE_Q(p), Dirichlet
// values on the whole boundary (bmop.cu:118-122).  Host only, no HIP.
//
// The mesh is unstructured (three caps meet at each of the coarse outer vertices), cells are multilinear images of
// the unit cube with a full inverse Jacobian per quadrature point: the description has no MFGPU_UNIFORM_J0 and takes
// the general-geometry operator path (fee_gpu.cuh:235-241,275-281).
//
// New vertices of a refinement step, as deal.II's Triangulation::execute_refinement places them with the default
// manifold weights (manifold.cc, get_default_points_and_weights): on the boundary -- lines and quads with all
// vertices on the sphere -- the normalised average of the surrounding points (SphericalManifold with equal radii);
// interior line: midpoint; interior quad / 2D cell: 1/16 x vertices + 3/16 x line midpoints; hex: 1/128 x vertices +
// 7/192 x line midpoints + 1/12 x quad centres.  deal.II itself is not available here, so this geometry is the
// restatement of that recipe, not a checked copy of its output ("Parity").
//
// DoFs are identified topologically -- vertex, (line, position from its lower-numbered end), (quad, position in the
// frame of its lowest-numbered vertex), cell interior -- so the numbering is conforming whatever the relative
// orientation of two cells sharing a face, and numbered in order of first use by the cells in refinement-tree order
// (children of a cell are consecutive: spatially local).
#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <unordered_map>

#include "mfgpu_mesh.h"

namespace mfgpu {

namespace {

using Pt = std::array<double, 3>;

struct BallCell {
  uint32_t v[8];       // vertex ids, lexicographic (x fastest)
  bool bnd[6] = {};    // faces x-, x+, y-, y+, z-, z+ on the boundary
};

struct Key2 {
  uint32_t a, b;
  bool operator==(const Key2 &o) const { return a == o.a && b == o.b; }
};
struct Key2Hash {
  size_t operator()(const Key2 &k) const { return (size_t)k.a * 0x9e3779b97f4a7c15ull ^ (size_t)k.b; }
};
struct Key4 {
  uint32_t v[4];  // sorted
  bool operator==(const Key4 &o) const { return std::equal(v, v + 4, o.v); }
};
struct Key4Hash {
  size_t operator()(const Key4 &k) const {
    size_t h = 1469598103934665603ull;
    for (int i = 0; i < 4; ++i) h = (h ^ k.v[i]) * 1099511628211ull;
    return h;
  }
};

Pt add(const Pt &a, const Pt &b, double wb) { return {a[0] + wb * b[0], a[1] + wb * b[1], a[2] + wb * b[2]}; }
Pt normalised(const Pt &a) {
  const double r = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  return {a[0] / r, a[1] / r, a[2] / r};
}

struct BallMesh {
  int dim;
  std::vector<Pt> vert;
  std::vector<uint8_t> vbnd;  // vertex on the sphere
  std::vector<BallCell> cells;

  uint32_t new_vertex(const Pt &p, bool b) {
    vert.push_back(p);
    vbnd.push_back(b ? 1 : 0);
    return (uint32_t)vert.size() - 1;
  }

  void coarse() {
    const double s = 1.0 / std::sqrt((double)dim), a = s / (1.0 + std::sqrt((double)dim));
    const int nv = 1 << dim;
    // inner vertices 0..nv-1, outer nv..2nv-1, both lexicographic in the signs
    for (int outer = 0; outer < 2; ++outer)
      for (int i = 0; i < nv; ++i) {
        Pt p{0, 0, 0};
        for (int d = 0; d < dim; ++d) p[d] = ((i >> d) & 1 ? 1.0 : -1.0) * (outer ? s : a);
        new_vertex(p, outer != 0);
      }
    BallCell c;
    for (int i = 0; i < nv; ++i) c.v[i] = (uint32_t)i;
    cells.push_back(c);
    // caps: local axes parallel to the global ones; along the cap's direction d the local coordinate runs from the
    // inner to the outer vertex on the + side and from the outer to the inner vertex on the - side
    for (int d = 0; d < dim; ++d)
      for (int side = 0; side < 2; ++side) {
        BallCell k;
        for (int i = 0; i < nv; ++i) {
          const int bit = (i >> d) & 1;
          const bool outer = side ? bit == 1 : bit == 0;
          const int sign_index = side ? (i | (1 << d)) : (i & ~(1 << d));
          k.v[i] = (uint32_t)(sign_index + (outer ? nv : 0));
        }
        k.bnd[2 * d + side] = true;
        cells.push_back(k);
      }
  }

  void refine() {
    std::unordered_map<Key2, uint32_t, Key2Hash> line_mid;
    std::unordered_map<Key4, uint32_t, Key4Hash> quad_mid;
    auto mid_line = [&](uint32_t a, uint32_t b) {
      const Key2 k{std::min(a, b), std::max(a, b)};
      auto it = line_mid.find(k);
      if (it != line_mid.end()) return it->second;
      Pt p = add(vert[a], vert[b], 1.0);
      for (double &x : p) x *= 0.5;
      const bool onb = vbnd[a] && vbnd[b];
      if (onb) p = normalised(p);
      const uint32_t id = new_vertex(p, onb);
      line_mid.emplace(k, id);
      return id;
    };
    // quad with corners q[0..3] lexicographic and line midpoints m[0..3]
    auto mid_quad = [&](const uint32_t *q, const uint32_t *m, bool is_cell_2d) {
      Key4 k{{q[0], q[1], q[2], q[3]}};
      std::sort(k.v, k.v + 4);
      if (!is_cell_2d) {
        auto it = quad_mid.find(k);
        if (it != quad_mid.end()) return it->second;
      }
      const bool onb = !is_cell_2d && vbnd[q[0]] && vbnd[q[1]] && vbnd[q[2]] && vbnd[q[3]];
      Pt p{0, 0, 0};
      for (int i = 0; i < 4; ++i) {
        p = add(p, vert[q[i]], onb ? 0.125 : 1.0 / 16.0);
        p = add(p, vert[m[i]], onb ? 0.125 : 3.0 / 16.0);
      }
      if (onb) p = normalised(p);
      const uint32_t id = new_vertex(p, onb);
      if (!is_cell_2d) quad_mid.emplace(k, id);
      return id;
    };
    std::vector<BallCell> fine;
    fine.reserve(cells.size() << dim);
    for (const BallCell &c : cells) {
      // 3^dim lattice of the refined cell's vertices
      uint32_t lat[27];
      auto at = [&](int i, int j, int k) -> uint32_t & { return lat[i + 3 * j + 9 * k]; };
      const int kz = dim == 3 ? 2 : 1;
      for (int k = 0; k < kz; ++k)
        for (int j = 0; j < 2; ++j)
          for (int i = 0; i < 2; ++i) at(2 * i, 2 * j, 2 * k) = c.v[i + 2 * j + 4 * k];
      // line midpoints
      for (int k = 0; k < (dim == 3 ? 3 : 1); k += 2)
        for (int j = 0; j < 3; j += 2) {
          at(1, j, k) = mid_line(at(0, j, k), at(2, j, k));
          at(j, 1, k) = mid_line(at(j, 0, k), at(j, 2, k));
        }
      if (dim == 3)
        for (int j = 0; j < 3; j += 2)
          for (int i = 0; i < 3; i += 2) at(i, j, 1) = mid_line(at(i, j, 0), at(i, j, 2));
      if (dim == 2) {
        const uint32_t q[4] = {at(0, 0, 0), at(2, 0, 0), at(0, 2, 0), at(2, 2, 0)};
        const uint32_t m[4] = {at(1, 0, 0), at(1, 2, 0), at(0, 1, 0), at(2, 1, 0)};
        at(1, 1, 0) = mid_quad(q, m, true);
      } else {
        for (int f = 0; f < 3; f += 2) {  // faces x = const, y = const, z = const at positions 0 and 2
          {
            const uint32_t q[4] = {at(f, 0, 0), at(f, 2, 0), at(f, 0, 2), at(f, 2, 2)};
            const uint32_t m[4] = {at(f, 1, 0), at(f, 1, 2), at(f, 0, 1), at(f, 2, 1)};
            at(f, 1, 1) = mid_quad(q, m, false);
          }
          {
            const uint32_t q[4] = {at(0, f, 0), at(2, f, 0), at(0, f, 2), at(2, f, 2)};
            const uint32_t m[4] = {at(1, f, 0), at(1, f, 2), at(0, f, 1), at(2, f, 1)};
            at(1, f, 1) = mid_quad(q, m, false);
          }
          {
            const uint32_t q[4] = {at(0, 0, f), at(2, 0, f), at(0, 2, f), at(2, 2, f)};
            const uint32_t m[4] = {at(1, 0, f), at(1, 2, f), at(0, 1, f), at(2, 1, f)};
            at(1, 1, f) = mid_quad(q, m, false);
          }
        }
        Pt p{0, 0, 0};
        for (int k = 0; k < 3; ++k)
          for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {
              const int odd = (i == 1) + (j == 1) + (k == 1);
              if (odd == 3) continue;
              p = add(p, vert[at(i, j, k)], odd == 0 ? 1.0 / 128.0 : odd == 1 ? 7.0 / 192.0 : 1.0 / 12.0);
            }
        at(1, 1, 1) = new_vertex(p, false);
      }
      for (int ck = 0; ck < kz; ++ck)
        for (int cj = 0; cj < 2; ++cj)
          for (int ci = 0; ci < 2; ++ci) {
            BallCell ch;
            for (int k = 0; k < kz; ++k)
              for (int j = 0; j < 2; ++j)
                for (int i = 0; i < 2; ++i) ch.v[i + 2 * j + 4 * k] = at(ci + i, cj + j, ck + k);
            const int cc[3] = {ci, cj, ck};
            for (int d = 0; d < dim; ++d) {
              ch.bnd[2 * d] = c.bnd[2 * d] && cc[d] == 0;
              ch.bnd[2 * d + 1] = c.bnd[2 * d + 1] && cc[d] == 1;
            }
            fine.push_back(ch);
          }
    }
    cells.swap(fine);
  }
};

// multilinear map of the unit cube and its derivative F[a][b] = d x_a / d xi_b
void map_point(const BallMesh &B, const BallCell &c, const double *xi, double *x, double *F) {
  const int dim = B.dim, nv = 1 << dim;
  for (int a = 0; a < dim; ++a) {
    x[a] = 0;
    for (int b = 0; b < dim; ++b) F[a * dim + b] = 0;
  }
  for (int i = 0; i < nv; ++i) {
    double w = 1.0, dw[3];
    for (int d = 0; d < dim; ++d) w *= ((i >> d) & 1) ? xi[d] : 1.0 - xi[d];
    for (int b = 0; b < dim; ++b) {
      dw[b] = ((i >> b) & 1) ? 1.0 : -1.0;
      for (int d = 0; d < dim; ++d)
        if (d != b) dw[b] *= ((i >> d) & 1) ? xi[d] : 1.0 - xi[d];
    }
    const Pt &p = B.vert[c.v[i]];
    for (int a = 0; a < dim; ++a) {
      x[a] += w * p[a];
      for (int b = 0; b < dim; ++b) F[a * dim + b] += dw[b] * p[a];
    }
  }
}

bool invert(int dim, const double *F, double *inv, double &det) {
  if (dim == 2) {
    det = F[0] * F[3] - F[1] * F[2];
    if (!(det > 0)) return false;
    inv[0] = F[3] / det;
    inv[1] = -F[1] / det;
    inv[2] = -F[2] / det;
    inv[3] = F[0] / det;
    return true;
  }
  const double c00 = F[4] * F[8] - F[5] * F[7], c01 = F[5] * F[6] - F[3] * F[8], c02 = F[3] * F[7] - F[4] * F[6];
  det = F[0] * c00 + F[1] * c01 + F[2] * c02;
  if (!(det > 0)) return false;
  inv[0] = c00 / det;
  inv[1] = (F[2] * F[7] - F[1] * F[8]) / det;
  inv[2] = (F[1] * F[5] - F[2] * F[4]) / det;
  inv[3] = c01 / det;
  inv[4] = (F[0] * F[8] - F[2] * F[6]) / det;
  inv[5] = (F[2] * F[3] - F[0] * F[5]) / det;
  inv[6] = c02 / det;
  inv[7] = (F[1] * F[6] - F[0] * F[7]) / det;
  inv[8] = (F[0] * F[4] - F[1] * F[3]) / det;
  return true;
}

}  // namespace

int build_ball(Mesh &M, int n_ref) {
  const int dim = M.dim, p = M.degree, n = p + 1, nd = ipow(n, dim);
  if (n_ref < 0 || (uint64_t)(2 * dim + 1) << (dim * n_ref) > (1ull << 24)) {
    set_error("mfgpu_mesh_create_ball: n_ref out of range");
    return MFGPU_EINVAL;
  }
  BallMesh B;
  B.dim = dim;
  B.coarse();
  for (int r = 0; r < n_ref; ++r) B.refine();
  const size_t nc = B.cells.size();
  M.n_cells = (uint32_t)nc;
  M.general = true;
  M.mg_kind = 1;  // refinement-tree order, children lexicographic
  M.init_tables();
  M.loc2glob.assign(nc * nd, 0u);
  M.constraint_mask.clear();

  // ---- topological dof identification
  std::unordered_map<Key2, uint32_t, Key2Hash> line_id;
  std::unordered_map<Key4, uint32_t, Key4Hash> quad_id;
  std::unordered_map<uint64_t, uint32_t> dof_of;  // entity key -> dof (entity kinds in disjoint ranges of the key)
  std::vector<uint32_t> vertex_dof(B.vert.size(), 0xffffffffu);
  uint32_t next = 0;
  const uint64_t pm1 = (uint64_t)(p > 1 ? p - 1 : 1);
  auto line_of = [&](uint32_t a, uint32_t b) {
    const Key2 k{std::min(a, b), std::max(a, b)};
    auto it = line_id.find(k);
    if (it != line_id.end()) return it->second;
    const uint32_t id = (uint32_t)line_id.size();
    line_id.emplace(k, id);
    return id;
  };
  auto get = [&](uint64_t key) {
    auto it = dof_of.find(key);
    if (it != dof_of.end()) return it->second;
    dof_of.emplace(key, next);
    return next++;
  };
  const uint64_t LINE_BASE = 1ull << 40, QUAD_BASE = 1ull << 50;
  for (size_t c = 0; c < nc; ++c) {
    const BallCell &cell = B.cells[c];
    for (int i = 0; i < nd; ++i) {
      int li[3] = {i % n, (i / n) % n, dim == 3 ? i / (n * n) : 0};
      int nfree = 0, fd[3];  // directions in which the node is strictly inside (0 < li < p)
      for (int d = 0; d < dim; ++d)
        if (li[d] > 0 && li[d] < p) fd[nfree++] = d;
      auto vid = [&](const int *corner) { return cell.v[corner[0] + 2 * corner[1] + 4 * (dim == 3 ? corner[2] : 0)]; };
      int corner[3] = {li[0] == p, li[1] == p, dim == 3 ? li[2] == p : 0};
      uint32_t dof;
      if (nfree == dim) {
        dof = next++;  // cell interior: unique
      } else if (nfree == 0) {
        uint32_t &vd = vertex_dof[vid(corner)];
        if (vd == 0xffffffffu) vd = next++;
        dof = vd;
      } else if (nfree == 1) {
        const int d = fd[0];
        int c0[3] = {corner[0], corner[1], corner[2]}, c1[3] = {corner[0], corner[1], corner[2]};
        c0[d] = 0;
        c1[d] = 1;
        const uint32_t a = vid(c0), b = vid(c1);
        const uint64_t t = a < b ? (uint64_t)li[d] : (uint64_t)(p - li[d]);  // position from the lower-numbered end
        dof = get(LINE_BASE + (uint64_t)line_of(a, b) * pm1 + (t - 1));
      } else {  // nfree == 2 in 3D: a node inside a quad
        const int d0 = fd[0], d1 = fd[1];
        uint32_t q[4];
        for (int b1 = 0; b1 < 2; ++b1)
          for (int b0 = 0; b0 < 2; ++b0) {
            int cc[3] = {corner[0], corner[1], corner[2]};
            cc[d0] = b0;
            cc[d1] = b1;
            q[b0 + 2 * b1] = vid(cc);
          }
        int o = 0;
        for (int k = 1; k < 4; ++k)
          if (q[k] < q[o]) o = k;
        const int a0 = o & 1, b0 = o >> 1;
        const uint64_t s = a0 ? p - li[d0] : li[d0], t = b0 ? p - li[d1] : li[d1];
        const uint32_t ns = q[(1 - a0) + 2 * b0], nt = q[a0 + 2 * (1 - b0)];
        Key4 k{{q[0], q[1], q[2], q[3]}};
        std::sort(k.v, k.v + 4);
        auto it = quad_id.find(k);
        uint32_t qid;
        if (it == quad_id.end()) {
          qid = (uint32_t)quad_id.size();
          quad_id.emplace(k, qid);
        } else {
          qid = it->second;
        }
        const uint64_t u = ns < nt ? s : t, v = ns < nt ? t : s;
        dof = get(QUAD_BASE + ((uint64_t)qid * pm1 + (u - 1)) * pm1 + (v - 1));
      }
      M.loc2glob[c * nd + i] = dof;
    }
  }
  M.n_dofs = next;

  // ---- geometry per quadrature point, dof coordinates, Dirichlet boundary
  M.JxW.assign(nc * nd, 0.0);
  M.inv_jac.assign(nc * nd * dim * dim, 0.0);
  M.qpoints.assign(nc * nd * dim, 0.0);
  M.dof_coords.assign((size_t)M.n_dofs * dim, 0.0);
  std::vector<uint8_t> onb(M.n_dofs, 0);
  for (size_t c = 0; c < nc; ++c) {
    const BallCell &cell = B.cells[c];
    for (int i = 0; i < nd; ++i) {
      const int li[3] = {i % n, (i / n) % n, dim == 3 ? i / (n * n) : 0};
      double xi[3], x[3], F[9], inv[9], det;
      for (int d = 0; d < dim; ++d) xi[d] = M.xq[li[d]];
      map_point(B, cell, xi, x, F);
      if (!invert(dim, F, inv, det)) {
        set_error("ball mesh: cell with non-positive Jacobian");
        return MFGPU_EINVAL;
      }
      double w = det;
      for (int d = 0; d < dim; ++d) w *= M.wq[li[d]];
      M.JxW[c * nd + i] = w;
      for (int k = 0; k < dim * dim; ++k) M.inv_jac[(c * nd + i) * dim * dim + k] = inv[k];
      for (int d = 0; d < dim; ++d) M.qpoints[(c * nd + i) * dim + d] = x[d];
      for (int d = 0; d < dim; ++d) xi[d] = M.nodes[li[d]];
      map_point(B, cell, xi, x, F);
      const uint32_t g = M.loc2glob[c * nd + i];
      for (int d = 0; d < dim; ++d) M.dof_coords[(size_t)g * dim + d] = x[d];
      for (int d = 0; d < dim; ++d)
        if ((cell.bnd[2 * d] && li[d] == 0) || (cell.bnd[2 * d + 1] && li[d] == p)) onb[g] = 1;
    }
  }
  M.constrained.clear();
  for (uint32_t g = 0; g < M.n_dofs; ++g)
    if (onb[g]) M.constrained.push_back(g);
  M.finalize_typed();
  return 0;
}

}  // namespace mfgpu

extern "C" int mfgpu_mesh_create_ball(int dim, int degree, int n_ref, int number_type, mfgpu_mesh **out) {
  if (!out || (dim != 2 && dim != 3) || degree < 1 || degree > 6 ||
      (number_type != MFGPU_F64 && number_type != MFGPU_F32)) {
    mfgpu::set_error("mfgpu_mesh_create_ball: bad argument");
    return MFGPU_EINVAL;
  }
  mfgpu_mesh *m = new mfgpu_mesh();
  m->mesh.dim = dim;
  m->mesh.degree = degree;
  m->mesh.number_type = number_type;
  int rc = mfgpu::build_ball(m->mesh, n_ref);
  if (rc) {
    delete m;
    return rc;
  }
  *out = m;
  return 0;
}
