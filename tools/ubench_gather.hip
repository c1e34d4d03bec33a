// Microbenchmark: what does ONE partly coalesced global load / store instruction cost a lone wave per SIMD that has
// FMAs to do?  (profiles/r02_notes.md)  A wave instruction touches 64 / R runs of R consecutive doubles (runs 1736
// bytes apart, as the x-lines of the C2 mesh are); dwordx2 = one double per lane, dwordx4 = two (half as many lanes'
// worth of runs per double).  Reported: cycles per loop iteration minus the 50-FMA baseline = cost of the one access.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
constexpr int kIter = 2000;
// MODE 0: fma only; 1: load x2; 2: load x4; 3: store x2; 4: store x4
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, unsigned long long *cyc, double *buf, int R, double a, double b) {
  const int l = threadIdx.x & 63, wv = blockIdx.x * 4 + threadIdx.x / 64;
  double acc[10], pend[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = l + i;
  // per wave a private 1 MB window; runs 1736 B apart; the window slides by 64 KB per iteration (no cache reuse)
  char *base = reinterpret_cast<char *>(buf) + (size_t)wv * (1 << 20);
  const int elems = (MODE == 2 || MODE == 4) ? 2 : 1;
  const int run = (l * elems) / R, in_run = (l * elems) % R;
  const size_t off = (size_t)run * 1736 + (size_t)in_run * 8;
  __syncthreads();
  const unsigned long long t0 = now();
  for (int it = 0; it < kIter; ++it) {
    char *p = base + off + (size_t)(it & 7) * (1 << 16) + ((it >> 3) & 7) * 16;
    if (MODE == 1) pend[it & 7] = *reinterpret_cast<double *>(p);
    if (MODE == 2) {
      d2 v = *reinterpret_cast<d2 *>(p);
      pend[it & 7] = v.x + v.y;
    }
    if (MODE == 3) *reinterpret_cast<double *>(p) = acc[it & 7];
    if (MODE == 4) *reinterpret_cast<d2 *>(p) = d2{acc[0], acc[1]};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int i = 0; i < 10; ++i) acc[i] = fma(acc[i], a, b);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1 || MODE == 2) acc[0] += pend[(it + 1) & 7] * 1e-30;
  }
  const unsigned long long t1 = now();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (l == 0) cyc[wv] = t1 - t0;
}
template <int MODE>
static double run(double *out, unsigned long long *cyc, double *buf, int R) {
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, buf, R, 1.0000001, 1e-9);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(1024);
  hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto c : h) avg += (double)c;
  return avg / 1024 / kIter;
}
int main() {
  double *out, *buf;
  unsigned long long *cyc;
  hipMalloc(&out, 256 * 256 * 8);
  hipMalloc(&cyc, 1024 * 8);
  hipMalloc(&buf, (size_t)1024 << 20);
  hipMemset(buf, 0, (size_t)1024 << 20);
  const double base = run<0>(out, cyc, buf, 1);
  printf("{\"bench\": \"50 fma\", \"cycles\": %.1f}\n", base);
  for (int R : {1, 2, 4, 8, 12, 16, 32, 64}) {
    const double l2 = run<1>(out, cyc, buf, R) - base, s2 = run<3>(out, cyc, buf, R) - base;
    double l4 = 0, s4 = 0;
    if (R % 2 == 0) {
      l4 = run<2>(out, cyc, buf, R) - base;
      s4 = run<4>(out, cyc, buf, R) - base;
    }
    printf("{\"run_doubles\": %d, \"load_dwordx2\": %.1f, \"store_dwordx2\": %.1f, \"load_dwordx4\": %.1f, \"store_dwordx4\": %.1f}\n", R,
           l2, s2, l4, s4);
  }
  return 0;
}
