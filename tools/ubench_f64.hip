// Microbenchmark: FP64 vector FMA vs FP64 MFMA on gfx950, per SIMD, in shader cycles (s_memtime).
//
// Question (VERDICT r01 item 4 / SURVEY row X1): can v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 carry the
// 5x5 / 7x7 1D contractions of the cell kernel faster than v_fma_f64?  Measures, for 1, 2 and 4 waves per SIMD:
//   fma    : 16 independent accumulator chains of v_fma_f64 per lane           -> cycles per wave-instruction
//   mfma16 : 4 independent accumulators of v_mfma_f64_16x16x4_f64 (2048 flop)   -> cycles per instruction
//   mfma4  : 4 independent accumulators of v_mfma_f64_4x4x4_4b_f64 (512 flop)   -> cycles per instruction
//   contract5: a batched 5x5 contraction out[q][col] = sum_k M[q][k] in[k][col] over 64*... columns done
//     (a) with v_fma_f64 (25 FMAs per column per lane) and (b) block-diagonal on mfma16 (3 matrices per tile,
//     4 k-steps): useful flops per cycle of both.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_f64.hip -o gpurun_out/ubench_f64 ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

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

constexpr int kIter = 2000;

__global__ void k_fma(double *out, unsigned long long *cyc, double a, double b) {
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x + i;
  __syncthreads();
  const unsigned long long t0 = now();
  for (int it = 0; it < kIter; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
  }
  const unsigned long long t1 = now();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

__global__ void k_mfma16(double *out, unsigned long long *cyc, double a, double b) {
  d4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  double av = a + threadIdx.x, bv = b - threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = now();
  for (int it = 0; it < kIter; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = now();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

__global__ void k_mfma4(double *out, unsigned long long *cyc, double a, double b) {
  double acc[4] = {0, 0, 0, 0};
  double av = a + threadIdx.x, bv = b - threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = now();
  for (int it = 0; it < kIter; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = now();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// MFMA and VALU from the SAME wave, interleaved: do the two pipes overlap?
__global__ void k_mix(double *out, unsigned long long *cyc, double a, double b) {
  d4 acc[4];
  double f[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i) f[i] = threadIdx.x + i;
  double av = a + threadIdx.x, bv = b - threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = now();
  for (int it = 0; it < kIter; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) f[4 * i + j] = fma(f[4 * i + j], a, b);
    }
  }
  const unsigned long long t1 = now();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// LDS: per-wave ds_write_b64 / ds_read_b64 streams (conflict-free, stride 1 double per lane), 16 per iteration
__global__ void k_lds(double *out, unsigned long long *cyc, int mode) {
  extern __shared__ double sm[];
  double *w = sm + (threadIdx.x / 64) * 64 * 17;
  const int l = threadIdx.x & 63;
  double v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = l + i;
  __syncthreads();
  const unsigned long long t0 = now();
  for (int it = 0; it < kIter; ++it) {
    if (mode == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) w[i * 64 + l] = v[i];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] += w[i * 64 + l];
    }
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  const unsigned long long t1 = now();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i] + w[i * 64 + l];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <typename K, typename... Args>
static int run(const char *name, K kern, int waves_per_simd, double per_iter_insts, double flop_per_inst, size_t lds,
               Args... args) {
  const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;  // 4 SIMDs per CU
  const int blocks = 256 * ((256 * waves_per_simd) / threads);
  const int nw = blocks * threads / 64;
  double *out;
  unsigned long long *cyc;
  CHECK(hipMalloc(&out, (size_t)blocks * threads * 8));
  CHECK(hipMalloc(&cyc, (size_t)nw * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, cyc, args...);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, cyc, args...);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(nw);
  CHECK(hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost));
  double avg = 0;
  for (auto c : h) avg += (double)c;
  avg /= nw;
  const double cyc_per_inst = avg / (kIter * per_iter_insts);  // per wave-instruction, as seen by ONE wave
  const double simd_cyc = cyc_per_inst / waves_per_simd;       // SIMD cycles per instruction (all its waves)
  const double tflops = (double)nw * kIter * per_iter_insts * flop_per_inst / (ms * 1e-3) / 1e12;
  printf("{\"bench\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_inst_per_wave\": %.2f, \"simd_cycles_per_inst\": %.2f, "
         "\"wall_ms\": %.4f, \"tflops\": %.1f}\n",
         name, waves_per_simd, cyc_per_inst, simd_cyc, ms, tflops);
  hipFree(out);
  hipFree(cyc);
  return 0;
}

int main() {
  for (int w : {1, 2, 4}) {
    if (run("v_fma_f64", k_fma, w, 16, 64 * 2, 0, 1.0000001, 1e-9)) return 1;
    if (run("v_mfma_f64_16x16x4", k_mfma16, w, 4, 2048, 0, 1.0000001, 1e-9)) return 1;
    if (run("v_mfma_f64_4x4x4_4b", k_mfma4, w, 4, 512, 0, 1.0000001, 1e-9)) return 1;
    if (run("mix_1mfma16_per_4fma", k_mix, w, 4, 2048 + 4 * 128, 0, 1.0000001, 1e-9)) return 1;
    const size_t lds = (size_t)(256 * w > 1024 ? 1024 : 256 * w) / 64 * 64 * 17 * 8;
    if (run("ds_write_b64", k_lds, w, 16, 0, lds, 0)) return 1;
    if (run("ds_read_b64", k_lds, w, 16, 0, lds, 1)) return 1;
  }
  return 0;
}
