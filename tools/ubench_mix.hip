// Microbenchmark: does a lone wave per SIMD hide LDS instructions behind FP64 FMAs?  (profiles/r02_notes.md)
// One 256-thread block per CU = one wave per SIMD, as apply_planes3 runs.  Per loop iteration:
//   fma      : 50 v_fma_f64 (10 independent chains)
//   wr       : 5 ds_write_b64            wr+fma : the same 5 writes spread between the 50 FMAs
//   rd+fma   : 5 ds_read_b64 consumed one iteration later, between the FMAs
//   add+fma  : 5 ds_add_f64 (no return) between the FMAs
//   gl+fma   : 1 scattered global_load_dwordx2 (64 distinct lines) per iteration between the FMAs, consumed 8 later
// Output: cycles per iteration (s_memtime), so  wr+fma - fma  is what five LDS stores cost a wave that has FMAs to do.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
constexpr int kIter = 1000;

template <int MODE>
__global__ void __launch_bounds__(256) k_mix(double *out, unsigned long long *cyc, const double *gsrc, double a, double b) {
  extern __shared__ double sm[];
  const int l = threadIdx.x;
  double *w = sm + l;  // conflict-free: lane-consecutive doubles, 5 rows of 256
  double acc[10], v[5] = {1, 2, 3, 4, 5}, pend[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = l + i;
  for (int i = 0; i < 5; ++i) w[i * 256] = l;
  const double *gp = gsrc + (size_t)(blockIdx.x * 256 + l) * 16;  // one 128-byte line per lane
  __syncthreads();
  const unsigned long long t0 = now();
  for (int it = 0; it < kIter; ++it) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      if (MODE == 1 || MODE == 2) w[s * 256] = acc[s];
      if (MODE == 3) v[s] = w[s * 256];
      if (MODE == 4) unsafeAtomicAdd(&w[s * 256], acc[s]);
      if (MODE == 5 && s == 0) pend[it & 7] = gp[(it & 15)];
      if (MODE != 1) {
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[i] = fma(acc[i], a, MODE == 3 ? v[s] * 1e-30 + b : b);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 5) acc[0] += pend[(it + 1) & 7] * 1e-30;
  }
  const unsigned long long t1 = now();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) s += acc[i];
  out[blockIdx.x * 256 + l] = s + w[0];
  if ((l & 63) == 0) cyc[blockIdx.x * 4 + l / 64] = t1 - t0;
}

template <int MODE>
static void run(const char *name, double *out, unsigned long long *cyc, const double *g) {
  hipLaunchKernelGGL(k_mix<MODE>, dim3(256), dim3(256), 5 * 256 * 8, 0, out, cyc, g, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k_mix<MODE>, dim3(256), dim3(256), 5 * 256 * 8, 0, out, cyc, g, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1024);
  hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto c : h) avg += (double)c;
  printf("{\"bench\": \"%s\", \"cycles_per_iteration\": %.1f}\n", name, avg / 1024 / kIter);
}

int main() {
  double *out, *g;
  unsigned long long *cyc;
  hipMalloc(&out, 256 * 256 * 8);
  hipMalloc(&cyc, 1024 * 8);
  hipMalloc(&g, (size_t)256 * 256 * 16 * 8);
  hipMemset(g, 0, (size_t)256 * 256 * 16 * 8);
  run<0>("50 fma", out, cyc, g);
  run<1>("5 ds_write_b64", out, cyc, g);
  run<2>("5 ds_write_b64 + 50 fma", out, cyc, g);
  run<3>("5 ds_read_b64 + 50 fma", out, cyc, g);
  run<4>("5 ds_add_f64 + 50 fma", out, cyc, g);
  run<5>("1 scattered global_load_dwordx2 + 50 fma", out, cyc, g);
  return 0;
}
