// Host-side planner and mesh stand-ins under AddressSanitizer + UBSan (CPU build only: the GPU pool has no sanitizer runs):
// every mesh kind x degree -> description -> plan (batching, plane records, hanging-node records), then teardown.
// Built and run by tests/test_host_sanitizers.py.
#include <cstdio>
#include <cstdint>
#include "mfgpu.h"
static int plan_of(mfgpu_mesh *m, int kernel) {
  mfgpu_desc d;
  if (mfgpu_mesh_desc(m, &d)) return 1;
  d.kernel = kernel;
  mfgpu_plan *p = nullptr;
  int rc = mfgpu_plan_create(&d, &p);
  if (rc) { std::printf("plan rc=%d %s\n", rc, mfgpu_last_error()); return rc; }
  const uint32_t *ptr; 
  long n16 = mfgpu_plan_array_u32(p, 16, &ptr), n15 = mfgpu_plan_array_u32(p, 15, &ptr), n0 = mfgpu_plan_array_u32(p, 0, &ptr);
  std::printf("  batches %ld hn_slot %ld hn words %ld\n", n0 - 1, n16, n15);
  mfgpu_plan_destroy(p);
  return 0;
}
int main() {
  int bad = 0;
  for (int p = 2; p <= 6; ++p)
    for (int nref = 3; nref <= 4; ++nref) {
      mfgpu_mesh *m = nullptr;
      if (mfgpu_mesh_create_adaptive(3, p, nref, 0, &m)) { std::printf("mesh fail\n"); return 1; }
      std::printf("adaptive p=%d nref=%d\n", p, nref);
      bad |= plan_of(m, 0);
      mfgpu_mesh_destroy(m);
    }
  for (int p = 3; p <= 6; ++p) {
    uint32_t n[3] = {7, 8, 10};
    mfgpu_mesh *m = nullptr;
    if (mfgpu_mesh_create_uniform(3, p, n, -1, 1, 2, 7, 0, &m)) { std::printf("mesh fail\n"); return 1; }
    std::printf("uniform slab p=%d\n", p);
    bad |= plan_of(m, 0);
    mfgpu_mesh_destroy(m);
  }
  { mfgpu_mesh *m = nullptr; mfgpu_mesh_create_ball(3, 4, 2, 0, &m); std::printf("ball\n"); bad |= plan_of(m, 0); mfgpu_mesh_destroy(m); }
  std::printf("done bad=%d\n", bad);
  return bad;
}
