// Micro-benchmark: does a v_mfma_f32_32x32x2_f32 that accumulates into the SAME registers as the one before it issue
// back to back?  One wave per SIMD; patterns over 4 accumulators.   hipcc --offload-arch=gfx950 -O3 -o dep_bench dep_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0)

template <int PAT>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const float a = (float)lane, b = (float)(lane + 1);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (PAT == 0) { MF(0); MF(1); MF(2); MF(3); MF(0); MF(1); MF(2); MF(3); }        // never the same twice in a row
    if (PAT == 1) { MF(0); MF(0); MF(0); MF(0); MF(0); MF(0); MF(0); MF(0); }        // one chain
    if (PAT == 2) { MF(0); MF(0); MF(1); MF(1); MF(2); MF(2); MF(3); MF(3); }        // pairs
    if (PAT == 3) { MF(0); MF(1); MF(1); MF(2); MF(3); MF(3); MF(2); MF(3); }        // the transposed kernel's order (3 repeats of 8... 9)
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int PAT>
static void run(const char* name) {
  const int wgs = 256, iters = 20000;
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, wgs * 256 * 4); (void)hipMalloc(&cyc, wgs * 4 * 8);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<PAT>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize(); }
  std::vector<unsigned long long> h(wgs * 4);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (auto x : h) sum += (double)x;
  printf("%-46s %6.2f ticks per MFMA\n", name, sum / h.size() / (8.0 * iters));
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  run<0>("4 accumulators round robin");
  run<1>("one accumulator (every MFMA depends on the last)");
  run<2>("pairs (A A B B C C D D)");
  run<3>("A B B C D D C D");
  return 0;
}
