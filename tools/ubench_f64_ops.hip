// Microbenchmark (round 3): issue cost of the FP64 VALU instructions the plane kernel is made of, per SIMD, for 1 and 2
// waves per SIMD, every CU busy.  Question: apply_planes4 with all memory and LDS traffic removed still takes ~80 us for
// 23 M wave-instructions (tools/ablate_q.py) -- 7 SIMD cycles per instruction.  v_fma_f64 alone measures 4.3 / 3.1
// cycles (r02_ubench_f64.jsonl): are v_add_f64 / v_mul_f64, SGPR operands, or dependent chains dearer?
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_f64_ops.hip -o gpurun_out/ubench_f64_ops
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
__device__ __forceinline__ unsigned long long now_real() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

constexpr int kIter = 2000;

// MODE 0: v_fma_f64 vgpr operands; 1: v_add_f64; 2: v_mul_f64; 3: v_fma_f64 with an SGPR-pair multiplier;
// 4: v_mul_f64 with SGPR; 5: mix 6 add : 4 mul : 6 fma (the kernel's proportions), SGPR multipliers;
// 6: as 5 but chains of 4 dependent instructions (4 independent chains of length 4)
template <int MODE>
__global__ void k_ops(double *out, unsigned long long *cyc, double a, double b) {
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x + i;
  const double av = a + 1e-9 * threadIdx.x;  // VGPR operand
  __syncthreads();
  const unsigned long long r0 = now_real();
  const unsigned long long t0 = now();
  for (int it = 0; it < kIter; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(av));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(av));
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(av));
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "s"(a), "v"(av));
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[i]) : "s"(a));
    } else if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i % 16 < 6) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(av));
        else if (i % 16 < 10) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[i]) : "s"(a));
        else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "s"(a), "v"(av));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = i / 4;  // chain c: 4 dependent instructions on acc[c]
        if (i % 4 == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(av));
        else if (i % 4 == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[c]) : "s"(a));
        else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[c]) : "s"(a), "v"(av));
      }
    }
  }
  const unsigned long long t1 = now();
  const unsigned long long r1 = now_real();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    cyc[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)] = t1 - t0;
    cyc[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) + 1] = r1 - r0;
  }
}

template <typename K>
static int run(const char *name, K kern, int waves_per_simd) {
  const int threads = 64, blocks = 256 * 4 * waves_per_simd;  // single-wave workgroups, as apply_planes4
  const int nw = blocks;
  double *out;
  unsigned long long *cyc;
  CHECK(hipMalloc(&out, (size_t)blocks * threads * 8));
  CHECK(hipMalloc(&cyc, (size_t)nw * 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0000001, 1e-9);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0000001, 1e-9);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * nw);
  CHECK(hipMemcpy(h.data(), cyc, nw * 16, hipMemcpyDeviceToHost));
  double avg = 0, real = 0;
  for (int i = 0; i < nw; ++i) {
    avg += (double)h[2 * i];
    real += (double)h[2 * i + 1];
  }
  avg /= nw;
  real /= nw;
  const double per_wave = avg / (kIter * 16.0);
  printf("{\"bench\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_inst_per_wave\": %.2f, \"simd_cycles_per_inst\": %.2f, "
         "\"clock_ghz\": %.3f, \"wall_ms\": %.4f, \"wave_inst_per_us_chip\": %.0f}\n",
         name, waves_per_simd, per_wave, per_wave / waves_per_simd, avg / real / 10.0,  // memrealtime: 100 MHz
         ms, (double)nw * kIter * 16 / (ms * 1e3));
  hipFree(out);
  hipFree(cyc);
  return 0;
}

int main() {
  for (int w : {1, 2, 4}) {
    if (run("v_fma_f64 vgpr", k_ops<0>, w)) return 1;
    if (run("v_add_f64", k_ops<1>, w)) return 1;
    if (run("v_mul_f64", k_ops<2>, w)) return 1;
    if (run("v_fma_f64 sgpr", k_ops<3>, w)) return 1;
    if (run("v_mul_f64 sgpr", k_ops<4>, w)) return 1;
    if (run("mix 6add:4mul:6fma", k_ops<5>, w)) return 1;
    if (run("mix, chains of 4", k_ops<6>, w)) return 1;
  }
  return 0;
}
