// Group-wise correlation volume on the matrix cores (the north-star's MFMA use): for one (batch, group, row) the
// per-group inner products of ALL (x, x') pairs in the disparity band are one small FP32 GEMM
//     corr[x][x'] = sum_{c in group} L[c][x] * R[c][x']          (M = x, N = x', K = C/G)
// computed with v_mfma_f32_32x32x2_f32 (exact FP32, the same ascending-c fma chain as the VALU kernel in volume.hip,
// so both forms are bit-identical).  An x tile of 32 positions needs x' in [x0 - 64, x0 + 32): three 32-wide B
// tiles, 3 * (C/G)/2 MFMAs.  The band is then read back along its diagonals through a per-wave LDS scratch (pitch 98
// floats: a diagonal walk has bank stride 99 = 3 mod 32, conflict free) so that every disparity plane receives 32
// consecutive x (128-byte stores): out[b, g, k, y, x] = corr[x][x - d_k] / (C/G).  Columns x' < 0 are zero in LDS (the
// LDS-DMA bounds check), which yields the reference-style zero fill for x < d_k with no branch.
// HBM-bound by design (writes 4 B per 2*(C/G) flops); usable when 0 <= d_k <= 64, otherwise volume.hip's kernel runs.
#include "dmb_common.h"

namespace dmb {

constexpr int GW_RPAD = 64;           // zero columns staged left of the right-feature row
constexpr int GW_LROW = 256;          // staged left-feature row length (x tiles of 32, up to 8 per pass)
constexpr int GW_RROW = GW_RPAD + GW_LROW;
constexpr int GW_CP = 98;             // scratch pitch
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int CG>
__global__ __launch_bounds__(256, 2) void gwc_mfma_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                          float* __restrict__ out, int C, int G, int H, int W, int D,
                                                          DispIdx idx, int out_channels, int och_off) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lrow = lds;                        // [CG][GW_LROW]
  float* rrow = lds + CG * GW_LROW;         // [CG][GW_RROW]
  float* scr = rrow + CG * GW_RROW;         // [4 waves][32][GW_CP]
  const int y = blockIdx.x % H, xpass = blockIdx.x / H;   // one pass = 256 columns
  const int g = blockIdx.y, b = blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const unsigned HW = (unsigned)H * W;
  const int xbase = xpass * GW_LROW;
  const float* Lb = L + ((size_t)b * C + (size_t)g * CG) * HW;
  const float* Rb = R + ((size_t)b * C + (size_t)g * CG) * HW;
  const __amdgpu_buffer_rsrc_t lrs = make_rsrc(Lb, (unsigned)CG * HW * 4u);
  const __amdgpu_buffer_rsrc_t rrs = make_rsrc(Rb, (unsigned)CG * HW * 4u);

  // ---- stage the group's rows: CG left rows (256 floats) + CG right rows (64 zero columns + 256 floats)
  if ((W & 3) == 0 && ((((uintptr_t)L | (uintptr_t)R) & 15) == 0)) {
    // 16-byte words: a left row is one copy instruction (64 lanes x 16 bytes), a right row two (64 + 16 words)
    for (int u = wave; u < CG * 3; u += 4) {
      const int c = u / 3, seg = u - c * 3;   // 0: left row, 1: right row words 0..63, 2: right row words 64..79
      const int word = (seg == 2 ? 64 : 0) + lane;
      const int gx = (seg == 0 ? xbase : xbase - GW_RPAD) + word * 4;
      const bool on = seg != 2 || lane < 16;
      const unsigned voff = (on && gx >= 0 && gx < W) ? (unsigned)gx * 4u : DMA_OOB;
      const unsigned soff = ((unsigned)c * HW + (unsigned)y * W) * 4u;
      float* dst = seg == 0 ? lrow + c * GW_LROW : rrow + c * GW_RROW + (seg == 2 ? 256 : 0);
      if (on) dma16(seg == 0 ? lrs : rrs, voff, soff, dst);
    }
  } else
  for (int u = wave; u < CG * 9; u += 4) {
    const int c = u / 9, seg = u - c * 9;  // segments 0..3: left row, 4..8: right row
    const bool left = seg < 4;
    const int col = (left ? seg : seg - 4) * 64 + lane;             // column inside the staged row
    const int gx = left ? xbase + col : xbase - GW_RPAD + col;
    const unsigned voff = (gx >= 0 && gx < W) ? (unsigned)gx * 4u : DMA_OOB;
    const unsigned soff = ((unsigned)c * HW + (unsigned)y * W) * 4u;
    float* dst = left ? lrow + c * GW_LROW + seg * 64 : rrow + c * GW_RROW + (seg - 4) * 64;
    dma4(left ? lrs : rrs, voff, soff, dst);
  }
  __syncthreads();

  float* myscr = scr + wave * 32 * GW_CP;
  float* o = out + ((size_t)b * out_channels + och_off + g) * (size_t)D * HW + (size_t)y * W;
  for (int xt = wave; xt < GW_LROW / 32; xt += 4) {
    const int x0 = xt * 32;                 // tile start inside the staged row
    if (xbase + x0 >= W) break;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < CG / 2; ++s) {
      const float a = lrow[(2 * s + h) * GW_LROW + x0 + j];                       // A[i = x][k = c]
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const float bv = rrow[(2 * s + h) * GW_RROW + x0 + t * 32 + j];           // x' = x0 - 64 + 32 t + j (+ 64 pad)
        acc[t] = DMB_MFMA(a, bv, acc[t]);
      }
    }
    // scratch[x][x' - (x0 - 64)]
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) myscr[cd_row(r, h) * GW_CP + t * 32 + j] = acc[t][r];
    if ((W & 3) == 0) {
      // diagonals, 16-byte form: lane = (plane p of a group of 8, x quad q): four diagonal reads (rows 4q .. 4q + 3 of the
      // scratch, bank stride 12 q - p: the 32 lanes of an LDS group cover the 32 banks once) -> one float4 -> one store
      // instruction writes 8 planes x 128 bytes.  All reads of a block of planes are issued before the first store.
      const int pq = lane >> 3, q4 = (lane & 7) * 4;
      const int gx = xbase + x0 + q4;
      constexpr int KB = 2;   // groups of 8 planes in flight
      for (int k0 = 0; k0 < D; k0 += 8 * KB) {
        float v[KB][4];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const int k = k0 + kb * 8 + pq;
          const int d = idx.d[k < D ? k : D - 1];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[kb][e] = myscr[(q4 + e) * GW_CP + (q4 + e) + GW_RPAD - d];
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const int k = k0 + kb * 8 + pq;
          if (k < D && gx < W)
            __builtin_nontemporal_store(f32x4_t{v[kb][0] / (float)CG, v[kb][1] / (float)CG, v[kb][2] / (float)CG, v[kb][3] / (float)CG},
                                        reinterpret_cast<f32x4_t*>(o + (size_t)k * HW + gx));
        }
      }
    } else {
      // diagonals: half-wave h takes disparity samples k = 2 q + h
      const int gx = xbase + x0 + j;
      for (int k = h; k < D; k += 2) {
        const int d = idx.d[k];
        const float v = myscr[j * GW_CP + j + GW_RPAD - d];
        if (gx < W) o[(size_t)k * HW + gx] = v / (float)CG;
      }
    }
  }
}

template <int CG>
static int launch_gwc_mfma(const float* L, const float* R, float* out, int B, int C, int G, int H, int W, int D,
                           const DispIdx& idx, int out_channels, int och_off, hipStream_t st) {
  const size_t lds = (size_t)(CG * (GW_LROW + GW_RROW) + 4 * 32 * GW_CP) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gwc_mfma_kernel<CG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    attr_set = true;
  }
  dim3 grid(H * cdiv(W, GW_LROW), G, B);
  hipLaunchKernelGGL((gwc_mfma_kernel<CG>), grid, dim3(256), lds, st, L, R, out, C, G, H, W, D, idx, out_channels, och_off);
  return launch_status("gwc_mfma launch failed");
}

// Returns DMB_EUNSUPPORTED (without setting the error string) when the MFMA form does not apply; the caller then
// falls back to the VALU kernel.
int gwc_mfma_dispatch(const float* L, const float* R, float* out, int B, int C, int G, int H, int W, int D,
                      const DispIdx& idx, int out_channels, int och_off, hipStream_t st) {
  const int CG = C / G;
  for (int k = 0; k < D; ++k)
    if (idx.d[k] < 0 || idx.d[k] > GW_RPAD) return DMB_EUNSUPPORTED;
  if (G > 65535 || B > 65535 || (long long)CG * H * W * 4 >= 0x7fffffffLL) return DMB_EUNSUPPORTED;
  switch (CG) {
    case 2: return launch_gwc_mfma<2>(L, R, out, B, C, G, H, W, D, idx, out_channels, och_off, st);
    case 4: return launch_gwc_mfma<4>(L, R, out, B, C, G, H, W, D, idx, out_channels, och_off, st);
    case 8: return launch_gwc_mfma<8>(L, R, out, B, C, G, H, W, D, idx, out_channels, och_off, st);
    case 16: return launch_gwc_mfma<16>(L, R, out, B, C, G, H, W, D, idx, out_channels, och_off, st);
    default: return DMB_EUNSUPPORTED;
  }
}

}  // namespace dmb
