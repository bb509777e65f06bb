#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  bf16x8 a[3], b[6][3];
  for (int i = 0; i < 3; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(float)(threadIdx.x + i + e);
  for (int t = 0; t < 6; ++t) for (int i = 0; i < 3; ++i) for (int e = 0; e < 8; ++e) b[t][i][e] = (__bf16)(float)(threadIdx.x * 3 + i + e + t);
  f32x16 acc[6];
  for (int t = 0; t < 6; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {        // tile-major: 6 dependent MFMAs per accumulator
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int p = 0; p < 6; ++p) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[p % 3], b[t][p / 2], acc[t], 0, 0, 0);
    } else {                // product-major: consecutive MFMAs on different accumulators
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[p % 3], b[t][p / 2], acc[t], 0, 0, 0);
    }
  }
  float s = 0;
  for (int t = 0; t < 6; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs : {256, 512}) for (int mode = 0; mode < 2; ++mode) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, out, iters); else hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)wgs * 4 * iters * 36;
    printf("wgs=%d mode=%s: %.3f ms, %.1f cycles/MFMA/SIMD at 2.4 GHz (waves per SIMD %d), %.0f TFLOP/s\n", wgs, mode ? "product-major" : "tile-major(dependent)", ms,
           ms * 1e-3 * 2.4e9 / (mf / 1024.0), wgs / 256, mf * 32768 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
