// Microbenchmark (round 3): cost of LDS f64 atomics (ds_add_f64) against plain LDS stores / loads, per wave, with 1, 2
// waves per SIMD of single-wave workgroups (as apply_planes4).  Each iteration issues 25 instructions of one kind at
// conflict-free addresses (lane-contiguous doubles) and then waits for them (s_waitcnt lgkmcnt(0)).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_lds_atomic.hip -o tools/bin/ubench_lds_atomic
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

constexpr int kIter = 400, kOps = 25;

// MODE 0: ds_write_b64   1: ds_add_f64 (no return)   2: ds_read_b64 + v_add + ds_write_b64 (read-modify-write in
// registers)   3: ds_add_f64 where lanes 5 c + k of a cell hit 5-strided addresses with 2-way collisions (as the kernel)
template <int MODE>
__global__ void k_lds(double *out, unsigned long long *cyc) {
  extern __shared__ double sm[];
  const int l = threadIdx.x;
  double v[kOps];
#pragma unroll
  for (int i = 0; i < kOps; ++i) v[i] = l + i;
  for (int i = l; i < 64 * kOps; i += 64) sm[i] = 0.0;
  __syncthreads();
  const unsigned long long t0 = now();
  for (int it = 0; it < kIter; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < kOps; ++i) sm[i * 64 + l] = v[i];
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < kOps; ++i) unsafeAtomicAdd(&sm[i * 64 + l], v[i]);
    } else if (MODE == 2) {
      double r[kOps];
#pragma unroll
      for (int i = 0; i < kOps; ++i) r[i] = sm[i * 64 + l];
#pragma unroll
      for (int i = 0; i < kOps; ++i) sm[i * 64 + l] = r[i] + v[i];
    } else {
#pragma unroll
      for (int i = 0; i < kOps; ++i) unsafeAtomicAdd(&sm[(i * 61 + (l / 5) * 25 + (l % 5) * 5) % (64 * kOps)], v[i]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = now();
  double s = 0;
#pragma unroll
  for (int i = 0; i < kOps; ++i) s += sm[i * 64 + l];
  out[blockIdx.x * 64 + l] = s;
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
static int run(const char *name, K kern, int waves_per_simd) {
  const int blocks = 256 * 4 * waves_per_simd;
  const size_t lds = 64 * kOps * 8;
  double *out;
  unsigned long long *cyc;
  CHECK(hipMalloc(&out, (size_t)blocks * 64 * 8));
  CHECK(hipMalloc(&cyc, (size_t)blocks * 8));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, out, cyc);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, out, cyc);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(blocks);
  CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
  double avg = 0;
  for (auto c : h) avg += (double)c;
  avg /= blocks;
  printf("{\"bench\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_group_of_%d\": %.0f, \"cycles_per_inst_per_wave\": %.1f}\n", name,
         waves_per_simd, kOps, avg / kIter, avg / kIter / kOps);
  CHECK(hipFree(out));
  CHECK(hipFree(cyc));
  return 0;
}

int main() {
  for (int w : {1, 2}) {
    if (run("ds_write_b64", k_lds<0>, w)) return 1;
    if (run("ds_add_f64", k_lds<1>, w)) return 1;
    if (run("ds_read_b64 + add + ds_write_b64", k_lds<2>, w)) return 1;
    if (run("ds_add_f64, colliding lanes", k_lds<3>, w)) return 1;
  }
  return 0;
}
