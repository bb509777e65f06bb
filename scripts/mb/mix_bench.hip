// Micro-benchmark: a wave that interleaves N plain instructions (v_fma_f32, or ds_read_b32) with every FP32 MFMA -- how many
// fit under the 64 cycles of a v_mfma_f32_32x32x2_f32 before the MFMA rate drops?  1, 2 or 3 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mix_bench mix_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int N, int KIND>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ float sh[4096];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = (float)i;
  __syncthreads();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = (float)(lane + i);
  const float a = (float)lane, b = (float)(lane + 1), m = 1.0001f, c = 0.5f;
  int idx = lane;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        if (KIND == 0) v[i % 16] = __builtin_fmaf(v[i % 16], m, c);
        else v[i % 16] += sh[(idx + 64 * (i + 4 * t)) & 4095];
      }
    }
    idx = (idx + 1) & 63;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int N, int KIND>
static void run(int waves_per_simd) {
  const int wgs = 256, threads = 256 * waves_per_simd, iters = 4000;
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)wgs * threads * 4); (void)hipMalloc(&cyc, (size_t)wgs * (threads / 64) * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<N, KIND>), dim3(wgs), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h((size_t)wgs * (threads / 64));
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (auto x : h) sum += (double)x;
  const double per_mfma_simd = sum / h.size() / (4.0 * iters) / waves_per_simd;   // SIMD ticks per MFMA
  printf("%s x%-2d per MFMA, %d wave(s)/SIMD: %6.1f ticks per MFMA per SIMD  (%.0f %% of the MFMA rate)\n", KIND ? "ds_read_b32" : "v_fma_f32  ", N,
         waves_per_simd, per_mfma_simd, 6400.0 / per_mfma_simd);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 3; ++w) {
    run<0, 0>(w); run<1, 0>(w); run<2, 0>(w); run<4, 0>(w); run<8, 0>(w); run<12, 0>(w); run<16, 0>(w);
    run<1, 1>(w); run<2, 1>(w); run<4, 1>(w); run<8, 1>(w);
  }
  return 0;
}
