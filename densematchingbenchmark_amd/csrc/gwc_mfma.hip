// Group-wise correlation volume on the matrix cores (the north-star's MFMA use): for one (batch, group, row) the
// per-group inner products of ALL (x, x') pairs in the disparity band are one small FP32 GEMM
//     corr[x][x'] = sum_{c in group} L[c][x] * R[c][x']          (M = x, N = x', K = C/G)
// computed with v_mfma_f32_32x32x2_f32 (exact FP32, the same ascending-c fma chain as the VALU kernel in volume.hip,
// so both forms are bit-identical).  An x tile of 32 positions needs x' in [x0 - 64, x0 + 32): three 32-wide B
// tiles, 3 * (C/G)/2 MFMAs.  The band 0 <= x - x' <= 64 of the products goes through an LDS scratch indexed [x][d]; the four
// waves' tiles (128 consecutive columns) are then read back together, lane = (plane of a pair, x quad), so that one store
// instruction writes 2 disparity planes x 512 contiguous bytes: out[b, g, k, y, x] = corr[x][x - d_k] / (C/G).  Columns
// x' < 0 are zero in LDS (the LDS-DMA bounds check), which yields the reference-style zero fill for x < d_k with no branch.
// HBM-bound by design (writes 4 B per 2*(C/G) flops); usable when 0 <= d_k <= 64, otherwise volume.hip's kernel runs.
#include "dmb_common.h"

namespace dmb {

constexpr int GW_RPAD = 64;           // zero columns staged left of the right-feature row
constexpr int GW_LROW = 256;          // staged left-feature row length (x tiles of 32, up to 8 per pass)
constexpr int GW_RROW = GW_RPAD + GW_LROW;
constexpr int GW_CP = 66;             // scratch pitch: row x of the tile holds corr[x][x - d] for d = 0..64 (65 values; 66 = 2 mod 16:
                                      // the 8 rows x 8 consecutive disparities a read instruction touches fall in 64 distinct banks)
constexpr int GW_SCR = 32 * GW_CP + 2;   // one wave's scratch region (tile), = 2 (mod 64)
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int CG>
__global__ __launch_bounds__(256, 3) void gwc_mfma_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                          float* __restrict__ out, int C, int G, int H, int W, int D,
                                                          DispIdx idx, int out_channels, int och_off, int rpw, int dmax, int dbg_arg) {
  // dbg: development build only (option 6; the constant 0 in the release build): 1 = no stores, 2 = no staging
  const int dbg = DMB_DBG(dbg_arg);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lrow = lds;                        // [CG][GW_LROW]
  float* rrow = lds + CG * GW_LROW;         // [CG][GW_RROW]
  float* scr = rrow + CG * GW_RROW;         // [4 waves][32][GW_CP]
  int* dlds = reinterpret_cast<int*>(scr + 4 * GW_SCR);   // the disparity list (a per-lane index into the kernel argument
                                                           // would be a global load in the middle of the read-back)
  // A workgroup walks rpw consecutive rows of one (batch, group, 256-column pass); the next row's copies are issued as soon
  // as the current row's products exist and land under its read-back.
  const int nrg = cdiv(H, rpw);
  const int y_first = (blockIdx.x % nrg) * rpw, xpass = blockIdx.x / nrg;   // one pass = 256 columns
  const int y_end = min(y_first + rpw, H);
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
  auto stage = [&](int y) {
  if (dbg & 2) {
  } else if ((W & 3) == 0 && ((((uintptr_t)L | (uintptr_t)R) & 15) == 0)) {
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
  };
  stage(y_first);
  for (int k = threadIdx.x; k < D; k += 256) dlds[k] = idx.d[k];
  float* myscr = scr + wave * GW_SCR;
  const int nround = min(GW_LROW / 128, cdiv(W - xbase, 128));
  for (int y = y_first; y < y_end; ++y) {
  __syncthreads();   // this row's copies have landed (and the previous row's read-back is over)
  float* o = out + ((size_t)b * out_channels + och_off + g) * (size_t)D * HW + (size_t)y * W;
  // One round = 4 x tiles of 32 (one per wave) = 128 consecutive columns.
  for (int round = 0; round < nround; ++round) {
    const int x0 = round * 128 + wave * 32;     // tile start inside the staged row
    if (xbase + x0 < W) {
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
      // scratch[x][d] with d = x - x' = 64 + row - (32 t + j): only the band 0 <= d <= 64 is kept (half of the 32 x 96
      // products; 8.4 KB per wave instead of 12.5, which is what admits a third workgroup per CU).  With row = c_r + 4 h
      // and jj = j - 4 h:  tile 1 (d = 32 + c_r - jj) lies inside the band entirely, tile 0 needs jj >= c_r, tile 2
      // jj <= c_r; rows of tile 0 whose smallest disparity 33 + c_r exceeds the largest sample are skipped wave-wide.
      const int jj = j - 4 * h;
      float* wbase = myscr + (4 * h) * GW_CP + GW_RPAD - jj;      // + c_r * (GW_CP + 1) - 32 t
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cr = (r & 3) + 8 * (r >> 2);
        wbase[cr * (GW_CP + 1) - 32] = acc[1][r];
        if (33 + cr <= dmax && jj >= cr) wbase[cr * (GW_CP + 1)] = acc[0][r];
        if (jj <= cr) wbase[cr * (GW_CP + 1) - 64] = acc[2][r];
      }
    }
    __syncthreads();
    // the staged rows are dead once the last round's products exist: the next row's copies run under this read-back
    if (round == nround - 1 && y + 1 < y_end) stage(y + 1);
    if ((W & 3) == 0) {
      // Read-back across the four waves' tiles: lane = (plane pp of a pair, x quad xq of the round's 128 columns), so one
      // store instruction writes 2 planes x 512 contiguous bytes.  Wave w takes the plane pairs w, w + 4, ...  Banks: tile
      // region pitch = 2 (mod 64), row pitch 66: (xq >> 3) * 2 + (xq & 7) * 8 + pp covers the 64 banks once.
      const int pp = lane >> 5, xq = lane & 31;
      const float* src = scr + (xq >> 3) * GW_SCR + ((xq & 7) * 4) * GW_CP;
      const int gx = xbase + round * 128 + xq * 4;
      constexpr int KB = 3;   // plane pairs in flight per wave
      for (int k0 = 2 * wave; k0 < D; k0 += 8 * KB) {
        float v[KB][4];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const int k = k0 + kb * 8 + pp;
          const int d = dlds[k < D ? k : D - 1];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[kb][e] = src[e * GW_CP + d];
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const int k = k0 + kb * 8 + pp;
          if (k < D && gx < W && !(dbg & 1))
            __builtin_nontemporal_store(f32x4_t{v[kb][0] / (float)CG, v[kb][1] / (float)CG, v[kb][2] / (float)CG, v[kb][3] / (float)CG},
                                        reinterpret_cast<f32x4_t*>(o + (size_t)k * HW + gx));
        }
      }
    } else {
      // half-wave h takes disparity samples k = 2 q + h of this wave's own tile
      const int gx = xbase + x0 + j;
      for (int k = h; k < D; k += 2) {
        const int d = dlds[k];
        const float v = myscr[j * GW_CP + d];
        if (gx < W) o[(size_t)k * HW + gx] = v / (float)CG;
      }
    }
    if (round + 1 < nround) __syncthreads();   // the scratch is rewritten by the next round
  }
  }
}

template <int CG>
static int launch_gwc_mfma(const float* L, const float* R, float* out, int B, int C, int G, int H, int W, int D,
                           const DispIdx& idx, int out_channels, int och_off, hipStream_t st) {
  const size_t lds = (size_t)(CG * (GW_LROW + GW_RROW) + 4 * GW_SCR + DMB_MAX_DISP_SAMPLES) * sizeof(float);
  DMB_ENSURE_LDS((&gwc_mfma_kernel<CG>), (size_t)(lds));
  const int rpw = DMB_OPT(5) > 0 ? DMB_OPT(5) : 2;   // rows per workgroup (DMB_OPT(5): development knob)
  int dmax = 0;
  for (int k = 0; k < D; ++k) dmax = idx.d[k] > dmax ? idx.d[k] : dmax;
  dim3 grid(cdiv(H, rpw) * cdiv(W, GW_LROW), G, B);
  hipLaunchKernelGGL((gwc_mfma_kernel<CG>), grid, dim3(256), lds, st, L, R, out, C, G, H, W, D, idx, out_channels, och_off, rpw,
                     dmax, DMB_OPT(6));
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
