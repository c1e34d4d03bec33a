// Microbenchmark (round 3): how long the CU's vector-memory front end (TA / L1) is busy with ONE wave instruction of the
// kinds apply_planes4 issues.  1, 2 or 8 single-wave workgroups per CU each issue a stream of 32 loads (or stores) per
// iteration from an L2-resident buffer and wait for them; cycles per instruction = what a wave pays at issue when the
// front end is the bottleneck.
//   0 dword coalesced (256 B)   1 dwordx2 coalesced (512 B)   2 dwordx4 coalesced (1 KB)
//   3 dwordx2 gather, runs of 12 doubles 217 doubles apart (the batch dof pattern)   4 dwordx2 gather, 64 separate lines
//   5 dwordx2 store coalesced   6 dwordx2 scatter store, runs of 12
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_vmem.hip -o tools/bin/ubench_vmem
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

constexpr int kIter = 200, kOps = 32;
typedef double d2_t __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k_vmem(double *buf, size_t words_per_wg, unsigned long long *cyc, double *sink) {
  const int l = threadIdx.x;
  double *base = buf + (size_t)blockIdx.x * words_per_wg;  // each workgroup its own 256 KB window (L2-resident)
  double acc = 0;
  const int run = l / 12, pos = l % 12;
  const unsigned long long t0 = now();
  for (int it = 0; it < kIter; ++it) {
    const int o = (it & 7) * 2048;
    if (MODE == 0) {
      float v[kOps];
#pragma unroll
      for (int i = 0; i < kOps; ++i) v[i] = reinterpret_cast<const volatile float *>(base)[o + i * 64 + l];
#pragma unroll
      for (int i = 0; i < kOps; ++i) acc += v[i];
    } else if (MODE == 1 || MODE == 3 || MODE == 4) {
      double v[kOps];
#pragma unroll
      for (int i = 0; i < kOps; ++i) {
        const int idx = MODE == 1 ? o + i * 64 + l : MODE == 3 ? o + i * 16 + run * 217 + pos : o + i + l * 16 * 32;
        v[i] = reinterpret_cast<const volatile double *>(base)[idx & 32767];
      }
#pragma unroll
      for (int i = 0; i < kOps; ++i) acc += v[i];
    } else if (MODE == 2) {
      d2_t v[kOps / 2];
#pragma unroll
      for (int i = 0; i < kOps / 2; ++i) v[i] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(base) + ((o / 2 + i * 64 + l) & 16383));
#pragma unroll
      for (int i = 0; i < kOps / 2; ++i) acc += v[i].x + v[i].y;
    } else {
#pragma unroll
      for (int i = 0; i < kOps; ++i) {
        const int idx = MODE == 5 ? o + i * 64 + l : o + i * 16 + run * 217 + pos;
        reinterpret_cast<volatile double *>(base)[idx & 32767] = acc + i;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = now();
  if (acc == 12345.678) sink[0] = acc;
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
static int run(const char *name, K kern, int wg_per_cu, int ops) {
  const int blocks = 256 * wg_per_cu;
  const size_t words = 32768;  // 256 KB per workgroup
  double *buf, *sink;
  unsigned long long *cyc;
  CHECK(hipMalloc(&buf, (size_t)blocks * words * 8));
  CHECK(hipMemset(buf, 0, (size_t)blocks * words * 8));
  CHECK(hipMalloc(&sink, 8));
  CHECK(hipMalloc(&cyc, (size_t)blocks * 8));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, buf, words, cyc, sink);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, buf, words, cyc, sink);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(blocks);
  CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
  double avg = 0;
  for (auto c : h) avg += (double)c;
  avg /= blocks;
  printf("{\"bench\": \"%s\", \"workgroups_per_cu\": %d, \"cycles_per_inst_per_wave\": %.1f, \"cu_cycles_per_inst\": %.1f}\n", name,
         wg_per_cu, avg / kIter / ops, avg / kIter / ops / wg_per_cu);
  CHECK(hipFree(buf));
  CHECK(hipFree(sink));
  CHECK(hipFree(cyc));
  return 0;
}

int main() {
  for (int w : {1, 2, 8}) {
    if (run("dword coalesced", k_vmem<0>, w, kOps)) return 1;
    if (run("dwordx2 coalesced", k_vmem<1>, w, kOps)) return 1;
    if (run("dwordx4 coalesced", k_vmem<2>, w, kOps / 2)) return 1;
    if (run("dwordx2 gather runs of 12", k_vmem<3>, w, kOps)) return 1;
    if (run("dwordx2 gather 64 lines", k_vmem<4>, w, kOps)) return 1;
    if (run("dwordx2 store coalesced", k_vmem<5>, w, kOps)) return 1;
    if (run("dwordx2 scatter runs of 12", k_vmem<6>, w, kOps)) return 1;
  }
  return 0;
}
