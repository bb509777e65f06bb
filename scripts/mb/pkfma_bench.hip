// Micro-benchmark: issue rate of v_pk_fma_f32 (two FP32 fmas per lane and instruction) on one wave per SIMD: multiplier from a VGPR
// pair, from an SGPR pair broadcast with op_sel_hi (what conv3d_c1v_kernel's weights compile to), and plain v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 -o pkfma_bench pkfma_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float ws) {
  const int lane = threadIdx.x & 63;
  f32x2 acc[8];
  for (int t = 0; t < 8; ++t) acc[t] = f32x2{(float)t, (float)lane};
  f32x2 in = f32x2{(float)lane * 0.5f, 1.f};
  f32x2 wv = f32x2{ws + lane * 1e-9f, ws + lane * 1e-9f};
  const unsigned long long wss = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(ws)) * 0x100000001ull;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(in), "v"(wv));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[t]) : "v"(in), "s"(wss));
        if (MODE == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[t].x) : "v"(in.x), "v"(wv.x));
        if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(acc[t]) : "v"(in), "v"(wv));
      }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int t = 0; t < 8; ++t) s += acc[t].x + acc[t].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int waves_per_simd) {
  const int wgs = 256 * waves_per_simd, iters = 4000;
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, wgs * 256 * 4); (void)hipMalloc(&cyc, wgs * 4 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters, 1.0001f);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> h(wgs * 4);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (auto x : h) sum += (double)x;
  const double instr = 32.0 * iters;
  printf("%-44s %d waves/SIMD: %6.2f ticks (100 MHz) per instruction and wave; chip: %.1f G wave-instr/s = %.2f cycles at 2.4 GHz per SIMD\n", name,
         waves_per_simd, sum / h.size() / instr, wgs * 4 * instr / ms * 1e-6, 1024 * 2.4e9 / (wgs * 4 * instr / (ms * 1e-3)));
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("v_pk_fma_f32 v, v, v", w);
    run<1>("v_pk_fma_f32 v, v, s (op_sel_hi broadcast)", w);
    run<2>("v_fma_f32", w);
    run<3>("v_pk_mul_f32", w);
  }
  return 0;
}
