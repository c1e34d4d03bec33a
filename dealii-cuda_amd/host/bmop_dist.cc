// bmop on N GPUs of one node, host side in C++ over the C-ABI: one process per GPU (started by run_bmop_dist.sh with
// HIP_VISIBLE_DEVICES = rank), the cube cut into z-slabs, interface planes exchanged over RCCL by mfgpu_vmult_dist
// (include/mfgpu.h).  Same protocol and output line as bmop.cu:134-153 (dst = 0.1; 100 x {swap; vmult}),
// printed by rank 0 with the GLOBAL number of dofs and the slowest rank's time.
// Rendezvous without MPI: rank 0 writes the 128-byte RCCL unique id to $MFGPU_ID_FILE, the others wait for it; timing
// is exchanged through small files next to it.
//   env: MFGPU_RANK, MFGPU_WORLD, MFGPU_ID_FILE      argv: cells per direction of the GLOBAL mesh [degree = 4]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mfgpu.h"

#define CHECK(x)                                                         \
  do {                                                                   \
    if ((x) != 0) {                                                      \
      fprintf(stderr, "%s failed: %s\n", #x, mfgpu_last_error());        \
      return 1;                                                          \
    }                                                                    \
  } while (0)

static bool read_file(const std::string &path, void *buf, size_t n) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  const size_t got = fread(buf, 1, n, f);
  fclose(f);
  return got == n;
}
static void write_file(const std::string &path, const void *buf, size_t n) {
  const std::string tmp = path + ".tmp";
  FILE *f = fopen(tmp.c_str(), "wb");
  fwrite(buf, 1, n, f);
  fclose(f);
  rename(tmp.c_str(), path.c_str());
}

int main(int argc, char **argv) {
  const int rank = getenv("MFGPU_RANK") ? atoi(getenv("MFGPU_RANK")) : 0;
  const int world = getenv("MFGPU_WORLD") ? atoi(getenv("MFGPU_WORLD")) : 1;
  const std::string idf = getenv("MFGPU_ID_FILE") ? getenv("MFGPU_ID_FILE") : "/tmp/mfgpu_bmop_dist.id";
  const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 54;
  const int degree = argc > 2 ? atoi(argv[2]) : 4;
  const int n_iterations = 100;

  unsigned char id[128];
  if (world > 1) {
    if (rank == 0) {
      CHECK(mfgpu_dist_unique_id(id));
      write_file(idf, id, sizeof(id));
    } else {
      while (!read_file(idf, id, sizeof(id))) std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }
  // balanced contiguous z-slabs
  const uint32_t base = n / (uint32_t)world, rem = n % (uint32_t)world;
  const uint32_t zb = (uint32_t)rank * base + std::min<uint32_t>((uint32_t)rank, rem), ze = zb + base + ((uint32_t)rank < rem ? 1 : 0);
  const uint32_t nper[3] = {n, n, n};
  mfgpu_mesh *mesh = nullptr;
  CHECK(mfgpu_mesh_create_uniform(3, degree, nper, -1.0, 1.0, zb, ze, MFGPU_F64, &mesh));
  mfgpu_desc desc;
  CHECK(mfgpu_mesh_desc(mesh, &desc));
  mfgpu_handle *op = nullptr;
  CHECK(mfgpu_create(&desc, &op));
  const uint32_t *lo = nullptr, *up = nullptr;
  const int64_t nlo = mfgpu_mesh_interface_dofs(mesh, 0, &lo), nup = mfgpu_mesh_interface_dofs(mesh, 1, &up);
  mfgpu_dist *dist = nullptr;
  CHECK(mfgpu_dist_create(world > 1 ? id : nullptr, rank, world, lo, (uint32_t)nlo, up, (uint32_t)nup,
                          desc.constrained_dofs, desc.n_constrained, desc.n_dofs, MFGPU_F64, &dist));
  CHECK(mfgpu_dist_attach(dist, op));
  void *a = nullptr, *b = nullptr;
  CHECK(mfgpu_vec_alloc(&a, desc.n_dofs, MFGPU_F64));
  CHECK(mfgpu_vec_alloc(&b, desc.n_dofs, MFGPU_F64));
  void *dst = a, *src = b;
  CHECK(mfgpu_vec_fill(dst, desc.n_dofs, MFGPU_F64, 0.1, nullptr));
  for (int i = 0; i < 3; ++i) {  // warm-up (and the first RCCL transfers)
    std::swap(dst, src);
    CHECK(mfgpu_vmult_dist(op, dist, dst, src, nullptr));
  }
  CHECK(mfgpu_vec_fill(dst, desc.n_dofs, MFGPU_F64, 0.1, nullptr));
  CHECK(mfgpu_device_synchronize());
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n_iterations; ++i) {
    std::swap(dst, src);
    CHECK(mfgpu_vmult_dist(op, dist, dst, src, nullptr));
  }
  CHECK(mfgpu_device_synchronize());
  double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  // slowest rank's time, through files
  write_file(idf + ".t" + std::to_string(rank), &wall, sizeof(wall));
  if (rank == 0) {
    for (int r = 1; r < world; ++r) {
      double w = 0;
      while (!read_file(idf + ".t" + std::to_string(r), &w, sizeof(w))) std::this_thread::sleep_for(std::chrono::milliseconds(20));
      wall = std::max(wall, w);
    }
    const double nd = (double)degree * n + 1;
    printf("%d\t%d\t%.0f\t%g\t%d GPU(s)\n", 3, degree, nd * nd * nd, wall / n_iterations, world);
  }
  mfgpu_dist_destroy(dist);
  mfgpu_destroy(op);
  mfgpu_vec_free(a);
  mfgpu_vec_free(b);
  mfgpu_mesh_destroy(mesh);
  return 0;
}
