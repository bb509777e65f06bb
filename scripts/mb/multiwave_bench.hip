// Micro-benchmark: FP32 MFMA rate of the whole chip with 1, 2, 3, 4 waves per SIMD issuing v_mfma_f32_32x32x2_f32 (three
// accumulators round robin, as the stride-1 kernel), (a) nothing else, (b) with the kernel's LDS fragment reads (4 ds_read_b32 per 3
// MFMAs, one k-step ahead), (c) b + a workgroup barrier every 81 MFMAs.  One workgroup of 256 x NW threads per CU (100 KB of LDS
// keeps a second one off the CU).   hipcc --offload-arch=gfx950 -O3 -o multiwave_bench multiwave_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NW, int MODE>
__global__ __launch_bounds__(256 * NW) void k(float* out, int iters) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 25000; i += 256 * NW) lds[i] = (float)(i & 255) * 0.001f;
  __syncthreads();
  f32x16 acc[3];
  for (int t = 0; t < 3; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const float* ab = lds + lane + (threadIdx.x >> 6) * 64;
  float a = ab[0], b0 = ab[1024], b1 = ab[2048], b2 = ab[3072];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 27; ++ks) {
      float na = a, n0 = b0, n1 = b1, n2 = b2;
      if (MODE >= 1) {
        const float* p = ab + ((ks + 1) % 27) * 64;
        na = p[0]; n0 = p[4096 + ks]; n1 = p[8192 + ks]; n2 = p[12288 + ks];
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
      a = na; b0 = n0; b1 = n1; b2 = n2;
    }
    if (MODE >= 2) __syncthreads();
  }
  float s = 0;
  for (int t = 0; t < 3; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 * NW + threadIdx.x] = s;
}

template <int NW, int MODE>
static void run() {
  const int wgs = 256, iters = 600 / NW;
  float* out;
  (void)hipMalloc(&out, wgs * 256 * NW * 4);
  (void)hipFuncSetAttribute((const void*)k<NW, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NW, MODE>), dim3(wgs), dim3(256 * NW), 100 * 1024, 0, out, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double flop = (double)wgs * 4 * NW * iters * 81.0 * 4096.0;
  printf("waves/SIMD %d  mode %d (%s): %.3f ms  %.1f TF/s = %.3f of 157.3\n", NW, MODE,
         MODE == 0 ? "MFMA only" : MODE == 1 ? "+ LDS fragment reads" : "+ reads + barrier per 81", best, flop / best * 1e-9,
         flop / best * 1e-9 / 157.3);
  (void)hipFree(out);
}

int main() {
  run<1, 0>(); run<2, 0>(); run<3, 0>(); run<4, 0>();
  run<1, 1>(); run<2, 1>(); run<3, 1>(); run<4, 1>();
  run<1, 2>(); run<2, 2>(); run<3, 2>(); run<4, 2>();
  return 0;
}
