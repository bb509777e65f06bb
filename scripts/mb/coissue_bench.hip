// Micro-benchmark: how fast does a wave issue plain vector instructions while ANOTHER wave on the same SIMD streams MFMAs?
// One workgroup of 512 threads per CU = 2 waves per SIMD: waves 0-3 run the MFMA stream (4 independent accumulators of
// v_mfma_f32_32x32x2_f32, 64 cycles each), waves 4-7 run a stream of independent v_fma_f32 (or LDS reads) and time
// themselves with s_memtime.  Build: hipcc --offload-arch=gfx950 -O3 -o coissue_bench coissue_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int NOPS>   // KIND 0: v_fma_f32 stream, 1: ds_read_b32 stream with a wait per 8 reads, 2: v_pk_fma_f32 stream;
                                // NOPS: s_nop 15 (16 idle issue cycles each) the MFMA wave inserts after every MFMA
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int mfma_iters, int valu_iters, int mfma_on, int prio) {
  __shared__ float sh[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  sh[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  if (wave < 4) {
    if (!mfma_on) return;
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float a = (float)lane, b = (float)(lane + 1);
    const unsigned long long m0 = __builtin_readcyclecounter();
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NOPS; ++n) asm volatile("s_nop 15");
      }
    }
    const unsigned long long m1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[1024 + blockIdx.x * 4 + wave] = m1 - m0;
  } else {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(lane + i);
    const float m = 1.0001f, c = 0.5f;
    if (prio) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (KIND == 0) {
      for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], m, c);       // 64 independent-ish v_fma per iteration
      }
    } else if (KIND == 1) {
      int idx = lane;
      for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += sh[(idx + 64 * i) & 4095];
        idx = (idx + 1) & 63;
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
      }
    } else {
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p[4];
      for (int i = 0; i < 4; ++i) p[i] = f2{v[2 * i], v[2 * i + 1]};
      for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], f2{m, m}, f2{c, c});   // 64 v_pk_fma per iteration
      }
      for (int i = 0; i < 4; ++i) { v[2 * i] = p[i].x; v[2 * i + 1] = p[i].y; }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave - 4] = t1 - t0;
  }
}

template <int KIND, int NOPS>
static void run(const char* name, int per_iter) {
  float* out; unsigned long long* cyc;
  const int wgs = 256;
  hipMalloc(&out, wgs * 512 * 4); hipMalloc(&cyc, 2 * wgs * 4 * 8);
  const int valu_iters = 2000;
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int on = cfg > 0, prio = cfg == 2;
    // enough MFMAs to outlast the vector stream: 64 cycles each
    const int mfma_iters = 40000;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((k<KIND, NOPS>), dim3(wgs), dim3(512), 0, 0, out, cyc, mfma_iters, valu_iters, on, prio);
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(2 * wgs * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0, msum = 0; for (int i = 0; i < wgs * 4; ++i) { sum += (double)h[i]; msum += (double)h[1024 + i]; }
    printf("%-24s [%d x s_nop 15 after each MFMA] other wave %-26s: %6.1f ticks per instruction", name, NOPS, !on ? "idle" : (prio ? "streams MFMAs, s_setprio 3" : "streams MFMAs"),
           sum / (wgs * 4) / ((double)valu_iters * per_iter));
    if (on) printf("   (MFMA wave: %.1f ticks per MFMA)", msum / (wgs * 4) / (4.0 * mfma_iters));
    printf("\n");
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0>("v_fma_f32", 64);
  run<0, 1>("v_fma_f32", 64);
  run<0, 2>("v_fma_f32", 64);
  run<0, 3>("v_fma_f32", 64);
  run<0, 4>("v_fma_f32", 64);
  run<1, 0>("ds_read_b32 (x8 + wait)", 8);
  run<1, 3>("ds_read_b32 (x8 + wait)", 8);
  return 0;
}
