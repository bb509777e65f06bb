// How fast does one CU move L2-resident bytes into LDS?  (a) LDS-DMA: buffer_load_dwordx4 ... lds (no registers), (b) the same
// bytes through registers: buffer_load_dwordx4 -> ds_write_b128, (c) LDS-DMA with dword words.  One workgroup of W waves per CU,
// every wave copies 1 KiB per instruction from a 64 KiB window (L2 / L1 resident) in a loop of 16 copies between waits.
//   hipcc --offload-arch=gfx950 -O3 scripts/mb/dma_rate_bench.hip -o scripts/mb/dma_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
template <int MODE>
__global__ __launch_bounds__(1024) void copy_kernel(const float* src, float* sink, int iters, int window_floats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(src + (size_t)blockIdx.x * window_floats, (unsigned)window_floats * 4u);
  float* dst = lds + wave * (16 * 256);   // 16 KiB per wave
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const unsigned base = (unsigned)((it * 7 + wave) % (window_floats / 4096)) * 16384u;   // a 16 KiB slice of the window
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + k * 256), 16, base + k * 1024u + lane * 16u, 0, 0, 0);
      __builtin_amdgcn_s_waitcnt(0x0F70);
    } else if (MODE == 1) {
      u32x4 v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(base + k * 1024u + lane * 16u), 0, 0);
#pragma unroll
      for (int k = 0; k < 16; ++k) *reinterpret_cast<u32x4*>(dst + k * 256 + lane * 4) = v[k];
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + k * 64), 4, base + k * 256u + lane * 4u, 0, 0, 0);
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    acc += dst[(it * 64 + lane) & 4095];
  }
  if (acc == 12345.678f) sink[0] = acc;
}
template <int MODE>
static void run(const char* name, int waves, const float* src, float* sink, int window_floats) {
  const int iters = 2000, grid = 256;
  const size_t lds = (size_t)waves * 16 * 256 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&copy_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(copy_kernel<MODE>, dim3(grid), dim3(64 * waves), lds, 0, src, sink, 50, window_floats);
  hipEventRecord(a);
  hipLaunchKernelGGL(copy_kernel<MODE>, dim3(grid), dim3(64 * waves), lds, 0, src, sink, iters, window_floats);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  const double bytes_per_cu = (double)iters * waves * 16.0 * (MODE == 2 ? 256.0 : 1024.0);
  printf("%-44s %2d waves/CU: %7.3f ms  %6.1f B/clk/CU at 2.4 GHz  (%5.1f clk per 64-lane instruction and CU)\n", name, waves, ms,
         bytes_per_cu / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / ((double)iters * waves * 16.0));
}
int main() {
  const int window_floats = 16384;   // 64 KiB per workgroup
  float *src, *sink;
  hipMalloc(&src, (size_t)256 * window_floats * 4);
  hipMalloc(&sink, 4);
  hipMemset(src, 0, (size_t)256 * window_floats * 4);
  for (int waves : {4, 8, 12, 16}) {
    if (waves * 16 > 160) continue;
    run<0>("LDS-DMA, 16-byte words", waves, src, sink, window_floats);
    run<1>("registers: load b128 -> ds_write_b128", waves, src, sink, window_floats);
    run<2>("LDS-DMA, dwords", waves, src, sink, window_floats);
  }
  return 0;
}
