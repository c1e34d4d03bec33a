// Multigrid level hierarchy of an ADAPTIVE stand-in mesh (host only): what
// DoFHandler::distribute_mg_dofs + MGConstrainedDoFs + MGTransferMatrixFreeGpu::build hand to the level operators,
// the transfer and copy_to_mg / copy_from_mg (poisson_mg.cu:152,199-209,325-326; mg_transfer_matrix_free_gpu.cu:150-330,
// 690-760) on a locally refined mesh:
//   level l mesh      = all octree cells of level l (leaves of that level and ancestors of finer leaves), conforming
//   refinement edge   = level dofs on faces between a level-l cell and a region level l does not cover
//                       (MGConstrainedDoFs::get_refinement_edge_indices)
//   transfer l-1 -> l = the level-(l-1) cells that are refined, with the (2p+1)^dim patch of their children
//   copy pairs        = (active dof, level dof) for the dofs of the cells active on level l, off its refinement edge
// The mesh must be 2:1 balanced over vertices too (mfgpu_mesh_create_adaptive_mg), so that a cell of level l + 1
// never touches the boundary of the level-l region.
#include <algorithm>
#include <array>
#include <unordered_map>

#include "mfgpu_mesh.h"

struct mfgpu_mg_hierarchy {
  int dim = 0, degree = 0;
  std::vector<mfgpu_mesh *> levels;
  std::vector<std::vector<uint32_t>> edge, pair_active, pair_level, tr_coarse, tr_fine;
};

namespace {
inline uint64_t key_of(uint32_t cx, uint32_t cy, uint32_t cz) { return ((uint64_t)cz << 42) | ((uint64_t)cy << 21) | cx; }
}  // namespace

extern "C" {

void mfgpu_mg_hierarchy_destroy(mfgpu_mg_hierarchy *h) {
  if (!h) return;
  for (mfgpu_mesh *m : h->levels) delete m;
  delete h;
}

int mfgpu_mg_hierarchy_create(const mfgpu_mesh *adaptive, mfgpu_mg_hierarchy **out) {
  using namespace mfgpu;
  if (!adaptive || !out) {
    set_error("mfgpu_mg_hierarchy_create: null argument");
    return MFGPU_EINVAL;
  }
  const Mesh &A = adaptive->mesh;
  if (A.cell_levels.size() != (size_t)A.n_cells * 4 || A.general) {
    set_error("mfgpu_mg_hierarchy_create: needs an octree stand-in mesh (mfgpu_mesh_create_adaptive_mg / _from_leaves)");
    return MFGPU_EINVAL;
  }
  const int dim = A.dim, p = A.degree, n = p + 1, nd = ipow(n, dim), nf = 2 * p + 1, NF = ipow(nf, dim);
  int Lmax = 0;
  for (uint32_t c = 0; c < A.n_cells; ++c) Lmax = std::max(Lmax, (int)A.cell_levels[4 * c]);
  std::vector<std::vector<std::array<uint32_t, 4>>> S(Lmax + 1);
  {
    std::vector<std::unordered_map<uint64_t, int>> seen(Lmax + 1);
    for (uint32_t c = 0; c < A.n_cells; ++c) {
      const int L = (int)A.cell_levels[4 * c];
      for (int l = 0; l <= L; ++l) {
        const uint32_t cx = A.cell_levels[4 * c + 1] >> (L - l), cy = A.cell_levels[4 * c + 2] >> (L - l),
                       cz = A.cell_levels[4 * c + 3] >> (L - l);
        if (seen[l].emplace(key_of(cx, cy, cz), 1).second) S[l].push_back({(uint32_t)l, cx, cy, cz});
      }
    }
  }
  mfgpu_mg_hierarchy *H = new mfgpu_mg_hierarchy();
  H->dim = dim;
  H->degree = p;
  const int nl = Lmax + 1;
  H->edge.resize(nl);
  H->pair_active.resize(nl);
  H->pair_level.resize(nl);
  H->tr_coarse.resize(nl);
  H->tr_fine.resize(nl);
  std::vector<std::unordered_map<uint64_t, uint32_t>> index(nl);  // level cell -> cell index in the level mesh
  auto fail = [&](int rc) {
    mfgpu_mg_hierarchy_destroy(H);
    return rc;
  };
  for (int l = 0; l < nl; ++l) {
    mfgpu_mesh *m = new mfgpu_mesh();
    m->mesh.dim = dim;
    m->mesh.degree = p;
    m->mesh.number_type = A.number_type;
    H->levels.push_back(m);
    int rc = build_from_tree_leaves(m->mesh, dim, S[l]);
    if (rc) return fail(rc);
    const Mesh &M = m->mesh;
    if (!M.constraint_mask.empty() && std::any_of(M.constraint_mask.begin(), M.constraint_mask.end(), [](uint32_t v) { return v != 0; })) {
      set_error("internal: a level mesh has hanging nodes");
      return fail(MFGPU_EINVAL);
    }
    for (uint32_t c = 0; c < M.n_cells; ++c)
      index[l][key_of(M.cell_levels[4 * c + 1], M.cell_levels[4 * c + 2], M.cell_levels[4 * c + 3])] = c;
    // refinement edge: faces whose neighbour of the same level lies inside the domain but is no cell of the level
    std::vector<uint8_t> is_edge(M.n_dofs, 0);
    for (uint32_t c = 0; c < M.n_cells; ++c) {
      const uint32_t cc[3] = {M.cell_levels[4 * c + 1], M.cell_levels[4 * c + 2], M.cell_levels[4 * c + 3]};
      for (int d = 0; d < dim; ++d)
        for (int side = 0; side < 2; ++side) {
          int64_t nb[3] = {cc[0], cc[1], cc[2]};
          nb[d] += side ? 1 : -1;
          if (nb[d] < 0 || nb[d] >= ((int64_t)1 << l)) continue;
          if (index[l].count(key_of((uint32_t)nb[0], (uint32_t)nb[1], (uint32_t)nb[2]))) continue;
          // (the neighbour may be listed later in the loop over cells: index[l] is complete, it was filled above)
          for (int i = 0; i < nd; ++i) {
            int ii = i, li = 0;
            for (int dd = 0; dd <= d; ++dd) {
              li = ii % n;
              ii /= n;
            }
            if (li == (side ? p : 0)) is_edge[M.loc2glob[(size_t)c * nd + i]] = 1;
          }
        }
    }
    for (uint32_t g = 0; g < M.n_dofs; ++g)
      if (is_edge[g]) H->edge[l].push_back(g);
    // transfer from level l - 1: refined parents and their children's patch
    if (l > 0) {
      const Mesh &C = H->levels[l - 1]->mesh;
      for (uint32_t c = 0; c < C.n_cells; ++c) {
        const uint32_t cc[3] = {C.cell_levels[4 * c + 1], C.cell_levels[4 * c + 2], C.cell_levels[4 * c + 3]};
        if (!index[l].count(key_of(2 * cc[0], 2 * cc[1], dim == 3 ? 2 * cc[2] : 0))) continue;
        H->tr_coarse[l].insert(H->tr_coarse[l].end(), &C.loc2glob[(size_t)c * nd], &C.loc2glob[(size_t)c * nd] + nd);
        for (int t = 0; t < NF; ++t) {
          const int X[3] = {t % nf, (t / nf) % nf, dim == 3 ? t / (nf * nf) : 0};
          uint32_t kid[3] = {0, 0, 0};
          int local = 0, stride = 1;
          for (int d = 0; d < dim; ++d) {
            const int a = X[d] > p ? 1 : 0;
            kid[d] = 2 * cc[d] + (uint32_t)a;
            local += (X[d] - a * p) * stride;
            stride *= n;
          }
          auto it = index[l].find(key_of(kid[0], kid[1], kid[2]));
          if (it == index[l].end()) {
            set_error("internal: a refined cell lacks a child on the next level");
            return fail(MFGPU_EINVAL);
          }
          H->tr_fine[l].push_back(M.loc2glob[(size_t)it->second * nd + local]);
        }
      }
    }
  }
  // copy pairs: the cells of the active mesh on their own level
  for (uint32_t c = 0; c < A.n_cells; ++c) {
    const int L = (int)A.cell_levels[4 * c];
    const Mesh &M = H->levels[L]->mesh;
    auto it = index[L].find(key_of(A.cell_levels[4 * c + 1], A.cell_levels[4 * c + 2], A.cell_levels[4 * c + 3]));
    if (it == index[L].end()) {
      set_error("internal: active cell missing on its level");
      return fail(MFGPU_EINVAL);
    }
    for (int i = 0; i < nd; ++i) {
      const uint32_t j = M.loc2glob[(size_t)it->second * nd + i];
      if (std::binary_search(H->edge[L].begin(), H->edge[L].end(), j)) continue;
      H->pair_active[L].push_back(A.loc2glob[(size_t)c * nd + i]);
      H->pair_level[L].push_back(j);
    }
  }
  *out = H;
  return 0;
}

int mfgpu_mg_n_levels(const mfgpu_mg_hierarchy *h) { return h ? (int)h->levels.size() : 0; }

const mfgpu_mesh *mfgpu_mg_level_mesh(const mfgpu_mg_hierarchy *h, int level) {
  return (h && level >= 0 && level < (int)h->levels.size()) ? h->levels[level] : nullptr;
}

int64_t mfgpu_mg_edge_dofs(const mfgpu_mg_hierarchy *h, int level, const uint32_t **ptr) {
  if (!h || !ptr || level < 0 || level >= (int)h->levels.size()) return MFGPU_EINVAL;
  *ptr = h->edge[level].data();
  return (int64_t)h->edge[level].size();
}

int64_t mfgpu_mg_copy_pairs(const mfgpu_mg_hierarchy *h, int level, const uint32_t **active_dofs, const uint32_t **level_dofs) {
  if (!h || !active_dofs || !level_dofs || level < 0 || level >= (int)h->levels.size()) return MFGPU_EINVAL;
  *active_dofs = h->pair_active[level].data();
  *level_dofs = h->pair_level[level].data();
  return (int64_t)h->pair_active[level].size();
}

int64_t mfgpu_mg_transfer_arrays(const mfgpu_mg_hierarchy *h, int level, const uint32_t **coarse_cell_dofs,
                                 const uint32_t **fine_patch_dofs) {
  if (!h || !coarse_cell_dofs || !fine_patch_dofs || level < 1 || level >= (int)h->levels.size()) return MFGPU_EINVAL;
  *coarse_cell_dofs = h->tr_coarse[level].data();
  *fine_patch_dofs = h->tr_fine[level].data();
  return (int64_t)(h->tr_coarse[level].size() / (size_t)mfgpu::ipow(h->degree + 1, h->dim));
}

}  // extern "C"
