// deal.II stand-in for the SETUP side of bmop (host only, no HIP): produces the plain arrays that
// MatrixFreeGpu::reinit extracts from Triangulation / DoFHandler / FEValues / ShapeInfo /
// ConstraintMatrix (reference matrix_free_gpu.cu:283-339,502-513; bmop.cu:111-130).
//
//  * 1D tables: FE_Q(p) on Gauss-Lobatto support points, QGauss<1>(p+1) on [0,1]
//  * uniform mesh: hyper_cube(lo,hi) with n cells per direction (bmop_common.h:119 generalised)
//  * adaptive mesh: see mfgpu_mesh_adaptive.cpp
#include <algorithm>
#include <cmath>
#include <cstring>

#include "mfgpu_mesh.h"

namespace mfgpu {

// ---- Legendre helpers (long double) ----
static void legendre(int n, long double x, long double &P, long double &dP) {
  long double p0 = 1.0L, p1 = x;
  if (n == 0) {
    P = 1.0L;
    dP = 0.0L;
    return;
  }
  for (int k = 2; k <= n; ++k) {
    long double pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
    p0 = p1;
    p1 = pk;
  }
  P = p1;
  dP = n * (x * p1 - p0) / (x * x - 1.0L);
}

void gauss_01(int n, std::vector<double> &x, std::vector<double> &w) {
  x.resize(n);
  w.resize(n);
  const long double pi = 3.141592653589793238462643383279502884L;
  for (int i = 0; i < n; ++i) {
    long double z = -cosl(pi * (i + 0.75L) / (n + 0.5L));
    for (int it = 0; it < 100; ++it) {
      long double P, dP;
      legendre(n, z, P, dP);
      long double dz = P / dP;
      z -= dz;
      if (fabsl(dz) < 1e-19L) break;
    }
    long double P, dP;
    legendre(n, z, P, dP);
    x[i] = (double)(0.5L * (z + 1.0L));
    w[i] = (double)(1.0L / ((1.0L - z * z) * dP * dP));
  }
  for (int i = 0; i < n / 2; ++i) {  // exact symmetry
    double a = 0.5 * (x[i] + (1.0 - x[n - 1 - i]));
    x[i] = a;
    x[n - 1 - i] = 1.0 - a;
    double ww = 0.5 * (w[i] + w[n - 1 - i]);
    w[i] = w[n - 1 - i] = ww;
  }
  if (n % 2) x[n / 2] = 0.5;
}

void gll_01(int p, std::vector<double> &x) {
  x.assign(p + 1, 0.0);
  x[p] = 1.0;
  const long double pi = 3.141592653589793238462643383279502884L;
  for (int i = 1; i < p; ++i) {
    // roots of P'_p: Newton on q(z) = P'_p(z), q' from the Legendre ODE
    long double z = -cosl(pi * i / p);
    for (int it = 0; it < 100; ++it) {
      long double P, dP;
      legendre(p, z, P, dP);
      long double d2P = (2 * z * dP - p * (p + 1) * P) / (1.0L - z * z);
      long double dz = dP / d2P;
      z -= dz;
      if (fabsl(dz) < 1e-19L) break;
    }
    x[i] = (double)(0.5L * (z + 1.0L));
  }
  for (int i = 0; i <= p / 2; ++i) {
    double a = 0.5 * (x[i] + (1.0 - x[p - i]));
    x[i] = a;
    x[p - i] = 1.0 - a;
  }
}

void lagrange_eval(const std::vector<double> &nodes, double x, std::vector<double> &val,
                   std::vector<double> &der) {
  const int n = (int)nodes.size();
  val.assign(n, 0.0);
  der.assign(n, 0.0);
  for (int i = 0; i < n; ++i) {
    long double denom = 1.0L, v = 1.0L, dsum = 0.0L;
    for (int j = 0; j < n; ++j)
      if (j != i) {
        denom *= (long double)nodes[i] - nodes[j];
        v *= (long double)x - nodes[j];
      }
    for (int m = 0; m < n; ++m) {
      if (m == i) continue;
      long double t = 1.0L;
      for (int j = 0; j < n; ++j)
        if (j != i && j != m) t *= (long double)x - nodes[j];
      dsum += t;
    }
    val[i] = (double)(v / denom);
    der[i] = (double)(dsum / denom);
  }
}

void Mesh::init_tables() {
  const int n = degree + 1;
  gll_01(degree, nodes);
  gauss_01(n, xq, wq);
  shape_values.assign(n * n, 0.0);
  shape_gradients.assign(n * n, 0.0);
  std::vector<double> v, d;
  for (int q = 0; q < n; ++q) {
    lagrange_eval(nodes, xq[q], v, d);
    for (int i = 0; i < n; ++i) {
      shape_values[i * n + q] = v[i];
      shape_gradients[i * n + q] = d[i];
    }
  }
  // W[i][j] = phi_j(x_i / 2)   (hanging_nodes.cuh:580-598)
  weights.assign(n * n, 0.0);
  for (int i = 0; i < n; ++i) {
    lagrange_eval(nodes, 0.5 * nodes[i], v, d);
    for (int j = 0; j < n; ++j) weights[i * n + j] = v[j];
  }
}

template <typename T>
static void to_typed(const std::vector<double> &src, std::vector<unsigned char> &dst) {
  dst.resize(src.size() * sizeof(T));
  T *p = reinterpret_cast<T *>(dst.data());
  for (size_t i = 0; i < src.size(); ++i) p[i] = (T)src[i];
}

void Mesh::finalize_typed() {
  if (number_type == MFGPU_F32) {
    to_typed<float>(JxW, t_JxW);
    to_typed<float>(inv_jac, t_inv_jac);
    to_typed<float>(qpoints, t_qpoints);
    to_typed<float>(shape_values, t_sv);
    to_typed<float>(shape_gradients, t_sg);
  }
}

void Mesh::fill_desc(mfgpu_desc &d) const {
  std::memset(&d, 0, sizeof(d));
  d.dim = dim;
  d.degree = degree;
  d.number_type = number_type;
  d.flags = (general ? 0u : MFGPU_UNIFORM_J0) | (constraint_mask.empty() ? 0u : MFGPU_HANGING_NODES);
  d.n_dofs = n_dofs;
  d.n_cells = n_cells;
  d.loc2glob = loc2glob.data();
  d.constraint_mask = constraint_mask.empty() ? nullptr : constraint_mask.data();
  const bool f32 = number_type == MFGPU_F32;
  d.JxW = f32 ? (const void *)t_JxW.data() : (const void *)JxW.data();
  d.inv_jac = f32 ? (const void *)t_inv_jac.data() : (const void *)inv_jac.data();
  d.coefficient = nullptr;
  d.quadrature_points = f32 ? (const void *)t_qpoints.data() : (const void *)qpoints.data();
  d.shape_values = f32 ? (const void *)t_sv.data() : (const void *)shape_values.data();
  d.shape_gradients = f32 ? (const void *)t_sg.data() : (const void *)shape_gradients.data();
  d.constraint_weights = weights.data();
  d.constrained_dofs = constrained.data();
  d.n_constrained = (uint32_t)constrained.size();
}

// hyper_cube(lo,hi), nper[d] cells per direction, cells [slab_begin, slab_end) of the last
// direction.  DoFs: lexicographic on the slab-local tensor grid.  Cells: lexicographic.
int build_uniform(Mesh &M, const uint32_t *nper, double lo, double hi, uint32_t sb, uint32_t se) {
  const int dim = M.dim, p = M.degree, n = p + 1;
  const int nd = ipow(n, dim);
  uint32_t nc[3] = {1, 1, 1}, ncl[3] = {1, 1, 1}, ng[3] = {1, 1, 1};
  for (int d = 0; d < dim; ++d) nc[d] = ncl[d] = nper[d];
  if (se <= sb || se > nc[dim - 1]) {
    set_error("bad slab range");
    return MFGPU_EINVAL;
  }
  ncl[dim - 1] = se - sb;
  M.mg_kind = (sb == 0 && se == nc[dim - 1]) ? 0 : -1;
  for (int d = 0; d < 3; ++d) M.nper[d] = nc[d];
  uint64_t ndofs64 = 1, ncells64 = 1;
  for (int d = 0; d < dim; ++d) {
    ng[d] = ncl[d] * p + 1;
    ndofs64 *= ng[d];
    ncells64 *= ncl[d];
  }
  if (ndofs64 >= (1ull << 32) || ncells64 * nd >= (1ull << 40)) {
    set_error("mesh too large for 32-bit dof indices");
    return MFGPU_EINVAL;
  }
  M.n_dofs = (uint32_t)ndofs64;
  M.n_cells = (uint32_t)ncells64;
  const double h = (hi - lo) / nc[0];
  uint32_t off[3] = {0, 0, 0};
  off[dim - 1] = sb;  // global cell offset of the slab

  M.init_tables();
  M.loc2glob.resize((size_t)M.n_cells * nd);
  M.JxW.resize((size_t)M.n_cells * nd);
  M.inv_jac.assign(M.n_cells, 1.0 / h);
  M.qpoints.resize((size_t)M.n_cells * nd * dim);
  std::vector<double> wloc(nd);
  for (int q = 0; q < nd; ++q) {
    int qq = q;
    double w = 1.0;
    for (int d = 0; d < dim; ++d) {
      w *= M.wq[qq % n];
      qq /= n;
    }
    wloc[q] = w * std::pow(h, dim);
  }
  const uint64_t stride[3] = {1, ng[0], (uint64_t)ng[0] * ng[1]};
  uint32_t cell = 0;
  for (uint32_t cz = 0; cz < ncl[2]; ++cz)
    for (uint32_t cy = 0; cy < ncl[1]; ++cy)
      for (uint32_t cx = 0; cx < ncl[0]; ++cx, ++cell) {
        const uint32_t cc[3] = {cx, cy, cz};
        uint32_t *l2g = &M.loc2glob[(size_t)cell * nd];
        double *qp = &M.qpoints[(size_t)cell * nd * dim];
        double *jw = &M.JxW[(size_t)cell * nd];
        for (int i = 0; i < nd; ++i) {
          int ii = i;
          uint64_t g = 0;
          for (int d = 0; d < dim; ++d) {
            const int li = ii % n;
            ii /= n;
            g += (uint64_t)(cc[d] * p + li) * stride[d];
            qp[i * dim + d] = lo + h * ((double)(cc[d] + off[d]) + M.xq[li]);
          }
          l2g[i] = (uint32_t)g;
          jw[i] = wloc[i];
        }
      }
  // dof coordinates + Dirichlet boundary of the GLOBAL cube (bmop.cu:118-122)
  M.dof_coords.resize((size_t)M.n_dofs * dim);
  M.constrained.clear();
  const uint32_t ngl_last = nc[dim - 1] * p + 1;  // global grid size in the slab direction
  for (uint32_t g = 0; g < M.n_dofs; ++g) {
    uint32_t gg = g;
    bool onb = false;
    for (int d = 0; d < dim; ++d) {
      uint32_t gi = gg % ng[d];
      gg /= ng[d];
      const uint32_t gglob = gi + off[d] * p;
      const uint32_t nglob = (d == dim - 1) ? ngl_last : ng[d];
      if (gglob == 0 || gglob == nglob - 1) onb = true;
      uint32_t c = gglob / p;
      if (c >= nc[d]) c = nc[d] - 1;
      M.dof_coords[(size_t)g * dim + d] = lo + h * ((double)c + M.nodes[gglob - c * p]);
    }
    if (onb) M.constrained.push_back(g);
  }
  // slab interface planes (multi-GPU): local z index 0 / last
  M.iface[0].clear();
  M.iface[1].clear();
  uint64_t plane = 1;
  for (int d = 0; d < dim - 1; ++d) plane *= ng[d];
  if (sb > 0)
    for (uint64_t i = 0; i < plane; ++i) M.iface[0].push_back((uint32_t)i);
  if (se < nc[dim - 1])
    for (uint64_t i = 0; i < plane; ++i)
      M.iface[1].push_back((uint32_t)(i + plane * (uint64_t)(ng[dim - 1] - 1)));
  M.finalize_typed();
  return 0;
}

int mesh_transfer_patches(const Mesh &C, const Mesh &F, std::vector<uint32_t> &cd, std::vector<uint32_t> &fd) {
  const int dim = C.dim, p = C.degree, n = p + 1, nf = 2 * p + 1;
  if (F.dim != dim || F.degree != p || C.mg_kind < 0 || C.mg_kind != F.mg_kind) {
    set_error("multigrid transfer: the two meshes are not a level pair of a supported kind");
    return MFGPU_EUNSUPPORTED;
  }
  if (C.mg_kind == 0) {
    for (int d = 0; d < dim; ++d)
      if (F.nper[d] != 2 * C.nper[d]) {
        set_error("multigrid transfer: the fine mesh is not the global refinement of the coarse one");
        return MFGPU_EINVAL;
      }
  } else if ((uint64_t)F.n_cells != ((uint64_t)C.n_cells << dim)) {
    set_error("multigrid transfer: the fine mesh is not the global refinement of the coarse one");
    return MFGPU_EINVAL;
  }
  const int nd = ipow(n, dim), NF = ipow(nf, dim);
  cd = C.loc2glob;
  fd.assign((size_t)C.n_cells * NF, 0u);
  for (uint32_t c = 0; c < C.n_cells; ++c) {
    uint32_t cc[3] = {0, 0, 0};
    if (C.mg_kind == 0) {
      uint32_t r = c;
      for (int d = 0; d < dim; ++d) {
        cc[d] = r % C.nper[d];
        r /= C.nper[d];
      }
    }
    for (int t = 0; t < NF; ++t) {
      int X[3] = {t % nf, (t / nf) % nf, dim == 3 ? t / (nf * nf) : 0};
      int child = 0, local = 0, stride = 1;
      uint64_t fcell = 0, fstride = 1;
      for (int d = 0; d < dim; ++d) {
        const int a = X[d] > p ? 1 : 0;
        child |= a << d;
        local += (X[d] - a * p) * stride;
        stride *= n;
        fcell += (uint64_t)(2 * cc[d] + a) * fstride;
        fstride *= F.nper[d];
      }
      if (C.mg_kind != 0) fcell = ((uint64_t)c << dim) + child;
      fd[(size_t)c * NF + t] = F.loc2glob[fcell * nd + local];
    }
  }
  return 0;
}

}  // namespace mfgpu

extern "C" {

int mfgpu_mesh_create_uniform(int dim, int degree, const uint32_t *n_per_dir, double lo, double hi,
                              uint32_t slab_begin, uint32_t slab_end, int number_type,
                              mfgpu_mesh **out) {
  if (!out || !n_per_dir || (dim != 2 && dim != 3) || degree < 1 || degree > 6 ||
      (number_type != MFGPU_F64 && number_type != MFGPU_F32)) {
    mfgpu::set_error("mfgpu_mesh_create_uniform: bad argument");
    return MFGPU_EINVAL;
  }
  for (int d = 0; d < dim; ++d)
    if (n_per_dir[d] == 0) {
      mfgpu::set_error("mfgpu_mesh_create_uniform: zero cells in a direction");
      return MFGPU_EINVAL;
    }
  mfgpu_mesh *m = new mfgpu_mesh();
  m->mesh.dim = dim;
  m->mesh.degree = degree;
  m->mesh.number_type = number_type;
  if (slab_begin == 0 && slab_end == 0) slab_end = n_per_dir[dim - 1];
  int rc = mfgpu::build_uniform(m->mesh, n_per_dir, lo, hi, slab_begin, slab_end);
  if (rc) {
    delete m;
    return rc;
  }
  *out = m;
  return 0;
}

void mfgpu_mesh_destroy(mfgpu_mesh *m) { delete m; }

int mfgpu_mesh_renumber(mfgpu_mesh *m, const uint32_t *new_index) {
  if (!m || !new_index) {
    mfgpu::set_error("mfgpu_mesh_renumber: null argument");
    return MFGPU_EINVAL;
  }
  mfgpu::Mesh &M = m->mesh;
  const uint32_t N = M.n_dofs;
  std::vector<uint8_t> seen(N, 0);
  for (uint32_t g = 0; g < N; ++g) {
    if (new_index[g] >= N || seen[new_index[g]]) {
      mfgpu::set_error("mfgpu_mesh_renumber: not a permutation of the dofs");
      return MFGPU_EINVAL;
    }
    seen[new_index[g]] = 1;
  }
  for (uint32_t &g : M.loc2glob) g = new_index[g];
  for (uint32_t &g : M.constrained) g = new_index[g];
  std::sort(M.constrained.begin(), M.constrained.end());  // the description wants an ascending list
  for (int w = 0; w < 2; ++w)
    for (uint32_t &g : M.iface[w]) g = new_index[g];  // (order kept: both neighbours agree on it)
  const int dim = M.dim;
  std::vector<double> xc(M.dof_coords.size());
  for (uint32_t g = 0; g < N; ++g)
    for (int d = 0; d < dim; ++d) xc[(size_t)new_index[g] * dim + d] = M.dof_coords[(size_t)g * dim + d];
  M.dof_coords.swap(xc);
  M.mg_kind = -1;  // the level-pair structure survives, but a renumbered level no longer matches its neighbours' patches
  return 0;
}

int64_t mfgpu_mesh_transfer_patches(const mfgpu_mesh *coarse, const mfgpu_mesh *fine, uint32_t *coarse_cell_dofs,
                                    uint32_t *fine_patch_dofs) {
  if (!coarse || !fine || !coarse_cell_dofs || !fine_patch_dofs) {
    mfgpu::set_error("mfgpu_mesh_transfer_patches: null argument");
    return MFGPU_EINVAL;
  }
  std::vector<uint32_t> cd, fd;
  int rc = mfgpu::mesh_transfer_patches(coarse->mesh, fine->mesh, cd, fd);
  if (rc) return rc;
  std::memcpy(coarse_cell_dofs, cd.data(), cd.size() * 4);
  std::memcpy(fine_patch_dofs, fd.data(), fd.size() * 4);
  return (int64_t)coarse->mesh.n_cells;
}

int mfgpu_mesh_desc(const mfgpu_mesh *m, mfgpu_desc *desc) {
  if (!m || !desc) {
    mfgpu::set_error("null argument");
    return MFGPU_EINVAL;
  }
  m->mesh.fill_desc(*desc);
  return 0;
}

int64_t mfgpu_mesh_dof_coords(const mfgpu_mesh *m, const double **ptr) {
  if (!m || !ptr) return MFGPU_EINVAL;
  *ptr = m->mesh.dof_coords.data();
  return (int64_t)m->mesh.dof_coords.size();
}

int64_t mfgpu_mesh_interface_dofs(const mfgpu_mesh *m, int which, const uint32_t **ptr) {
  if (!m || !ptr || which < 0 || which > 1) return MFGPU_EINVAL;
  *ptr = m->mesh.iface[which].data();
  return (int64_t)m->mesh.iface[which].size();
}

}  // extern "C"
