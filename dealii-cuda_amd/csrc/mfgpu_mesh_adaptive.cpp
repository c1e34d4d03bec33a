// Adaptive (octree, hanging-node) mesh stand-in: placeholder until the hanging-node setup lands.
#include "mfgpu_mesh.h"

namespace mfgpu {
int build_adaptive(Mesh &, int) {
  set_error("adaptive mesh setup is not implemented yet");
  return MFGPU_EUNSUPPORTED;
}
}  // namespace mfgpu

extern "C" int mfgpu_mesh_create_adaptive(int dim, int degree, int n_ref, int number_type,
                                          mfgpu_mesh **out) {
  if (!out) return MFGPU_EINVAL;
  mfgpu_mesh *m = new mfgpu_mesh();
  m->mesh.dim = dim;
  m->mesh.degree = degree;
  m->mesh.number_type = number_type;
  int rc = mfgpu::build_adaptive(m->mesh, n_ref);
  if (rc) {
    delete m;
    return rc;
  }
  *out = m;
  return 0;
}
