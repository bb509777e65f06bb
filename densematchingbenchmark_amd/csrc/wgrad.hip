// Weight gradients of the 3x3x3 convolutions (training side) as FP32 implicit GEMMs on the matrix cores.
//
//   dW[co, ci, tap] = sum_{b, v} dc[b, co, v] * x[b, ci, S v + tap - 1]      (S = 1 here; M = co, N = ci, K = voxels)
//
// (dc = gradient with respect to the raw convolution output: torch.nn.grad.conv3d_weight.)  The GEMM is K-dominated --
// 25 M voxels against a 32 x 32 x 27 result -- so the result lives in registers for the whole launch and the two input
// tensors stream through LDS exactly once:
//   * a workgroup owns one (32 output channels x 32 input channels) block and walks (4 rows x 24 columns) columns of
//     the volume along z; wave w owns taps 7w .. 7w+6 (112 accumulator registers), every wave reads the same dc
//     fragment (A) and its taps' shifted x fragments (B) -- 8 ds_read_b32 per 7 MFMAs;
//   * x planes sit in a ring of four LDS slots (z - 1, z, z + 1 in use, z + 2 landing by LDS-DMA), so every plane is
//     staged once per column and not three times; dc planes are double buffered;
//   * an MFMA k-step multiplies two x-adjacent voxels (lanes 0-31: even column, lanes 32-63: odd); channel pitches
//     (196 / 100 floats) are = 4 mod 64 so that the 2 x 32 lanes of a fragment read fall 2-way on the banks;
//   * persistent launch (one workgroup per CU): partial results go to a workspace [slot][tap][32][32] and a second
//     kernel adds the slots in a fixed order -- no atomics, bit-reproducible.
//
// Reference semantics: autograd of nn.Conv3d in dmb/modeling/stereo/layers/basic_layers.py:68-83,160-177.
#include "dmb_common.h"

namespace dmb {

template <int TX_>
struct WgCfg {
  static constexpr int TY = 4, TX = TX_, ROWS = TY + 2, XOFF = 3;
  static constexpr int P = (4 + TX + 1 + 3) / 4 * 4;            // staged row: aligned column x0 - 4 .. x0 + TX + 3
  static constexpr int SX = ROWS * P + 4, SD = TY * TX + 4;     // channel pitches (floats), = 4 mod 64
  static constexpr int XPLANE = 32 * SX, DPLANE = 32 * SD;
  static constexpr int NRING = 4;
  static constexpr int LDS_FLOATS = NRING * XPLANE + 2 * DPLANE;
  static constexpr int UX = ROWS * P / 4, UD = TY * TX / 4;     // 16-byte units per channel plane
  static constexpr int KSTEPS = TY * TX / 2;
  static constexpr int NTAPW = 7;                               // taps per wave (4 x 7 = 28 >= 27; the last one is a dummy)
  static_assert(SX % 8 == 4 && SD % 8 == 4 && UX <= 64 && UD <= 64 && LDS_FLOATS * 4 <= 160 * 1024, "tile");
};

// workspace layout: ws[((blk * nslots + slot) * 27 + tap) * 1024 + m * 32 + n], blk = cob * ncib + cib
template <bool V16, int TX_>
__global__ __launch_bounds__(256, 1) void conv3d_wgrad_s1_kernel(const float* __restrict__ x, const float* __restrict__ dc,
                                                                 float* __restrict__ ws, int B, int Ci, int Co, int D, int H,
                                                                 int W, int ntx, int nty, int nzs, int zseg) {
  typedef WgCfg<TX_> C;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xring = lds;
  float* dbuf = lds + C::NRING * C::XPLANE;
  const int ncib = cdiv(Ci, 32);
  const int cib = blockIdx.y % ncib, cob = blockIdx.y / ncib;
  const int slot = xcd_remap(blockIdx.x, gridDim.x), nslots = gridDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n = lane & 31, kk = lane >> 5;
  const unsigned HW = (unsigned)H * W, DHW = (unsigned)D * HW;
  const int items = B * nzs * nty * ntx;
  const int nci = min(32, Ci - cib * 32), nco = min(32, Co - cob * 32);   // real channels of this block

  f32x16 acc[C::NTAPW];
#pragma unroll
  for (int i = 0; i < C::NTAPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // this wave's taps: B-fragment offsets inside a plane and the plane (dz) they read
  int tapoff[C::NTAPW], tapdz[C::NTAPW];
#pragma unroll
  for (int i = 0; i < C::NTAPW; ++i) {
    const int t = min(wave * C::NTAPW + i, 26);
    tapdz[i] = t / 9;
    tapoff[i] = n * C::SX + kk + C::XOFF + ((t / 3) % 3) * C::P + (t % 3);
  }
  const int aoff = n * C::SD + kk;

  for (int it = slot; it < items; it += nslots) {
    int t = it;
    const int tx = t % ntx;
    t /= ntx;
    const int ty = t % nty;
    t /= nty;
    const int zs = t % nzs;
    const int b = t / nzs;
    const int x0 = tx * C::TX, y0 = ty * C::TY, za = zs * zseg, zb = min(D, za + zseg);
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(x + ((size_t)b * Ci + cib * 32) * DHW, (unsigned)nci * DHW * 4u);
    const __amdgpu_buffer_rsrc_t drs = make_rsrc(dc + ((size_t)b * Co + cob * 32) * DHW, (unsigned)nco * DHW * 4u);
    // per-lane (row, unit) of the two staging patterns; the plane and the channel are wave-uniform scalar offsets
    unsigned xvo, dvo;
    {
      const int row = lane / (C::P / 4), un = lane % (C::P / 4);
      const int gy = y0 - 1 + row, gx = x0 - 4 + 4 * un;
      xvo = (lane < C::UX && gy >= 0 && gy < H && gx >= 0 && gx < W) ? ((unsigned)gy * W + (unsigned)gx) * 4u : DMA_OOB;
      const int drow = lane / (C::TX / 4), dun = lane % (C::TX / 4);
      const int dgy = y0 + drow, dgx = x0 + 4 * dun;
      dvo = (lane < C::UD && dgy < H && dgx < W) ? ((unsigned)dgy * W + (unsigned)dgx) * 4u : DMA_OOB;
    }
    // (channels >= nci / nco fall outside the resource and read zeros; a plane outside the volume is written as zeros)
    // One staging step = one channel plane of x or dc (one 16-byte LDS-DMA instruction, or one per row with dword
    // copies).  Steps are dealt out between the k-steps of the multiply loop: a wave that issues its 16 copies in one go
    // sits in the address queue for thousands of cycles while its SIMD's matrix core idles (one wave per SIMD here).
    auto stage_x1 = [&](int gz, int ring, int i) {
      const bool zok = gz >= 0 && gz < D;
      const int c = wave * 8 + i;
      float* dst = xring + ring * C::XPLANE + c * C::SX;
      if constexpr (V16) {
        if (lane < C::UX) dma16(xrs, zok ? xvo : DMA_OOB, zok ? ((unsigned)c * DHW + (unsigned)gz * HW) * 4u : 0u, dst);
      } else {
        const int gxs = x0 - 4 + lane;
        if (lane < C::P) {
#pragma unroll
          for (int row = 0; row < C::ROWS; ++row) {
            const int gy = y0 - 1 + row;
            const bool ok = zok && gy >= 0 && gy < H && gxs >= 0 && gxs < W;
            dma4(xrs, ok ? ((unsigned)gy * W + (unsigned)gxs) * 4u : DMA_OOB, ok ? ((unsigned)c * DHW + (unsigned)gz * HW) * 4u : 0u,
                 dst + row * C::P);
          }
        }
      }
    };
    auto stage_d1 = [&](int gz, int buf, int i) {
      const bool zok = gz >= 0 && gz < zb;
      const int c = wave * 8 + i;
      float* dst = dbuf + buf * C::DPLANE + c * C::SD;
      if constexpr (V16) {
        if (lane < C::UD) dma16(drs, zok ? dvo : DMA_OOB, zok ? ((unsigned)c * DHW + (unsigned)gz * HW) * 4u : 0u, dst);
      } else {
        const int gxs = x0 + lane;
        if (lane < C::TX) {
#pragma unroll
          for (int row = 0; row < C::TY; ++row) {
            const int gy = y0 + row;
            const bool ok = zok && gy < H && gxs < W;
            dma4(drs, ok ? ((unsigned)gy * W + (unsigned)gxs) * 4u : DMA_OOB, ok ? ((unsigned)c * DHW + (unsigned)gz * HW) * 4u : 0u,
                 dst + row * C::TX);
          }
        }
      }
    };
    auto stage_x = [&](int gz, int ring) {
#pragma unroll
      for (int i = 0; i < 8; ++i) stage_x1(gz, ring, i);
    };
    auto stage_d = [&](int gz, int buf) {
#pragma unroll
      for (int i = 0; i < 8; ++i) stage_d1(gz, buf, i);
    };

    // prologue: planes za - 1, za, za + 1 -> ring slots 0, 1, 2; dc plane za -> buffer 0
    __syncthreads();   // the previous column's last plane is still being read by slower waves
    stage_x(za - 1, 0);
    stage_x(za, 1);
    stage_x(za + 1, 2);
    stage_d(za, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    for (int z = za; z < zb; ++z) {
      const int rel = z - za;   // plane z + dz - 1 sits in ring slot (rel + dz) % 4
      const bool more = z + 1 < zb;
      const float* ap = dbuf + (rel & 1) * C::DPLANE + aoff;
      const float* bp[C::NTAPW];
#pragma unroll
      for (int i = 0; i < C::NTAPW; ++i) bp[i] = xring + ((rel + tapdz[i]) & 3) * C::XPLANE + tapoff[i];
      float af[2], bf[2][C::NTAPW];
      auto load_frag = [&](int q, float& a, float (&bq)[C::NTAPW]) {
        const int r = q / (C::TX / 2), c2 = 2 * (q % (C::TX / 2));
        a = ap[r * C::TX + c2];
#pragma unroll
        for (int i = 0; i < C::NTAPW; ++i) bq[i] = bp[i][r * C::P + c2];
      };
      load_frag(0, af[0], bf[0]);
#pragma unroll
      for (int q = 0; q < C::KSTEPS; ++q) {
        if (q + 1 < C::KSTEPS) load_frag(q + 1, af[(q + 1) & 1], bf[(q + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < C::NTAPW; ++i) acc[i] = DMB_MFMA(af[q & 1], bf[q & 1][i], acc[i]);
        if (q % 2 == 0 && q / 2 < 16 && more) {   // the next planes, one copy every other k-step (KSTEPS >= 32)
          if (q / 2 < 8)
            stage_x1(z + 2, (rel + 3) & 3, q / 2);
          else
            stage_d1(z + 1, (rel + 1) & 1, q / 2 - 8);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // the next planes have landed
      __syncthreads();                      // ... and everyone is done with the oldest ones
    }
  }

  // partial result of this slot
  float* wsb = ws + ((size_t)blockIdx.y * nslots + slot) * 27 * 1024;
#pragma unroll
  for (int i = 0; i < C::NTAPW; ++i) {
    const int t = wave * C::NTAPW + i;
    if (t < 27) {
#pragma unroll
      for (int r = 0; r < 16; ++r) wsb[t * 1024 + cd_row(r, kk) * 32 + n] = acc[i][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Stride 2: out[cs, cb, tap] = sum_{b, o} small[b, cs, o] * big[b, cb, 2 o + tap - 1], the weight gradient of both
//   nn.Conv3d(stride 2):      small = dc [Co], big = x [Ci]   -> out = dW  [Co, Ci, 27]
//   nn.ConvTranspose3d(s 2):  small = x [Ci],  big = dy [Co]  -> out = dWt [Ci, Co, 27]
// Same structure as the stride-1 kernel; a small plane z needs big planes 2z - 1, 2z, 2z + 1 (ring of five slots, two
// landing per step); the tile is 2 rows x 12 columns of the small tensor (5 x 28 staged floats of the big one per
// channel); fragment reads of the big tile step 2 floats per k lane (bank pattern 12 n + 2 kk: still 2-way).
// Staging units are linear in LDS across channels (the unit count per channel is made odd: pitch = 4 mod 8 floats), so one LDS-DMA
// instruction moves 64 units whatever the plane size.
// ------------------------------------------------------------------------------------------------------------------
// (round 6) The tile is a parameter: 2 x 12 (84 MFMAs per wave between two barriers) or 4 x 8 (112; and 64-column rows -- the
// training crops -- tile exactly where 12-column tiles compute 72); dmb_conv3d_k3s2_wgrad_f32 picks per launch.
template <int TY_, int TX_>
struct Wg2Cfg {
  static constexpr int TY = TY_, TX = TX_, ROWS = 2 * TY + 1, XOFF = 3;
  static constexpr int P = (4 + 2 * TX + 3) / 4 * 4;            // staged row of the big tile: aligned column 2 x0 - 4 ..
  static constexpr int UBU = ROWS * P / 4, USU = TY * TX / 4;   // 16-byte units per channel plane ...
  static constexpr int UB = UBU | 1, US = USU | 1;              // ... padded to an odd count (pitch = 4 mod 8 floats)
  static constexpr int SB = UB * 4, SS = US * 4;                // channel pitches (floats)
  static constexpr int BPLANE = 32 * SB, SPLANE = 32 * SS;
  static constexpr int NRING = 5;
  static constexpr int LDS_FLOATS = NRING * BPLANE + 2 * SPLANE;
  static constexpr int KSTEPS = TY * TX / 2;
  static constexpr int NTAPW = 7;
  static constexpr int IB = (32 * UB + 255) / 256, IS = (32 * US + 255) / 256;   // copy instructions per wave and plane
  static_assert(SB % 8 == 4 && SS % 8 == 4 && LDS_FLOATS * 4 <= 160 * 1024 && 2 * IB + IS <= KSTEPS && TX % 4 == 0, "tile");
};

template <class C, bool V16>
__global__ __launch_bounds__(256, 1) void conv3d_wgrad_s2_kernel(const float* __restrict__ sm, const float* __restrict__ bg,
                                                                 float* __restrict__ ws, int B, int Cs, int Cb, int Ds, int Hs,
                                                                 int Ws, int Db, int Hb, int Wb, int ntx, int nty, int nzs,
                                                                 int zseg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ring = lds;
  float* sbuf = lds + C::NRING * C::BPLANE;
  const int ncbb = cdiv(Cb, 32);
  const int cbb = blockIdx.y % ncbb, csb = blockIdx.y / ncbb;
  const int slot = xcd_remap(blockIdx.x, gridDim.x), nslots = gridDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n = lane & 31, kk = lane >> 5;
  const unsigned HWs = (unsigned)Hs * Ws, DHWs = (unsigned)Ds * HWs, HWb = (unsigned)Hb * Wb, DHWb = (unsigned)Db * HWb;
  const int items = B * nzs * nty * ntx;
  const int ncs = min(32, Cs - csb * 32), ncb = min(32, Cb - cbb * 32);

  f32x16 acc[C::NTAPW];
#pragma unroll
  for (int i = 0; i < C::NTAPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  int tapoff[C::NTAPW], tapdz[C::NTAPW];
#pragma unroll
  for (int i = 0; i < C::NTAPW; ++i) {
    const int t = min(wave * C::NTAPW + i, 26);
    tapdz[i] = t / 9;
    tapoff[i] = n * C::SB + 2 * kk + C::XOFF + ((t / 3) % 3) * C::P + (t % 3);
  }
  const int aoff = n * C::SS + kk;

  for (int it = slot; it < items; it += nslots) {
    int t = it;
    const int tx = t % ntx;
    t /= ntx;
    const int ty = t % nty;
    t /= nty;
    const int zs = t % nzs;
    const int b = t / nzs;
    const int x0 = tx * C::TX, y0 = ty * C::TY, za = zs * zseg, zb = min(Ds, za + zseg);
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(bg + ((size_t)b * Cb + cbb * 32) * DHWb, (unsigned)ncb * DHWb * 4u);
    const __amdgpu_buffer_rsrc_t srs = make_rsrc(sm + ((size_t)b * Cs + csb * 32) * DHWs, (unsigned)ncs * DHWs * 4u);
    // per-lane source offsets (channel + in-plane part) of this wave's copy instructions; the plane is a scalar offset
    unsigned bvo[C::IB], svo[C::IS];
#pragma unroll
    for (int j = 0; j < C::IB; ++j) {
      const int u = (j * 4 + wave) * 64 + lane, ch = u / C::UB, un = u - ch * C::UB;
      const int row = un / (C::P / 4), cu = un - row * (C::P / 4);
      const int gy = 2 * y0 - 1 + row, gx = 2 * x0 - 4 + 4 * cu;
      const bool ok = ch < 32 && un < C::UBU && gy >= 0 && gy < Hb && gx >= 0 && gx < Wb;
      bvo[j] = ok ? ((unsigned)ch * DHWb + (unsigned)gy * Wb + (unsigned)gx) * 4u : DMA_OOB;
    }
#pragma unroll
    for (int j = 0; j < C::IS; ++j) {
      const int u = (j * 4 + wave) * 64 + lane, ch = u / C::US, un = u - ch * C::US;
      const int row = un / (C::TX / 4), cu = un - row * (C::TX / 4);
      const int gy = y0 + row, gx = x0 + 4 * cu;
      const bool ok = ch < 32 && un < C::USU && gy < Hs && gx < Ws;
      svo[j] = ok ? ((unsigned)ch * DHWs + (unsigned)gy * Ws + (unsigned)gx) * 4u : DMA_OOB;
    }
    auto stage_b1 = [&](int gz, int rslot, int j) {   // copy instruction j of this wave for big plane gz
      const bool zok = gz >= 0 && gz < Db;
      const int first = (j * 4 + wave) * 64;
      if (first + lane < 32 * C::UB) dma16(brs, zok ? bvo[j] : DMA_OOB, zok ? (unsigned)gz * HWb * 4u : 0u, ring + rslot * C::BPLANE + first * 4);
    };
    auto stage_s1 = [&](int gz, int buf, int j) {
      const bool zok = gz >= 0 && gz < zb;
      const int first = (j * 4 + wave) * 64;
      if (first + lane < 32 * C::US) dma16(srs, zok ? svo[j] : DMA_OOB, zok ? (unsigned)gz * HWs * 4u : 0u, sbuf + buf * C::SPLANE + first * 4);
    };
    // widths that are not multiples of 4: whole planes with dword copies, one staged row per instruction (slow path)
    auto stage_b_rows = [&](int gz, int rslot) {
      const bool zok = gz >= 0 && gz < Db;
      const int gx = 2 * x0 - 4 + lane;
      if (lane < C::P) {
        for (int i = 0; i < 8; ++i) {
          const int c = wave * 8 + i;
#pragma unroll
          for (int row = 0; row < C::ROWS; ++row) {
            const int gy = 2 * y0 - 1 + row;
            const bool ok = zok && gy >= 0 && gy < Hb && gx >= 0 && gx < Wb;
            dma4(brs, ok ? ((unsigned)gy * Wb + (unsigned)gx) * 4u : DMA_OOB, ok ? ((unsigned)c * DHWb + (unsigned)gz * HWb) * 4u : 0u,
                 ring + rslot * C::BPLANE + c * C::SB + row * C::P);
          }
        }
      }
    };
    auto stage_s_rows = [&](int gz, int buf) {
      const bool zok = gz >= 0 && gz < zb;
      const int gx = x0 + lane;
      if (lane < C::TX) {
        for (int i = 0; i < 8; ++i) {
          const int c = wave * 8 + i;
#pragma unroll
          for (int row = 0; row < C::TY; ++row) {
            const int gy = y0 + row;
            const bool ok = zok && gy < Hs && gx < Ws;
            dma4(srs, ok ? ((unsigned)gy * Ws + (unsigned)gx) * 4u : DMA_OOB, ok ? ((unsigned)c * DHWs + (unsigned)gz * HWs) * 4u : 0u,
                 sbuf + buf * C::SPLANE + c * C::SS + row * C::TX);
          }
        }
      }
    };
    // ring slot of big plane p: (p - (2 za - 1)) % 5
    __syncthreads();
    if constexpr (V16) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < C::IB; ++j) stage_b1(2 * za - 1 + q, q, j);
#pragma unroll
      for (int j = 0; j < C::IS; ++j) stage_s1(za, 0, j);
    } else {
      for (int q = 0; q < 3; ++q) stage_b_rows(2 * za - 1 + q, q);
      stage_s_rows(za, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    int q0 = 0;   // ring index of plane 2 z - 1
    for (int z = za; z < zb; ++z) {
      const int rel = z - za;
      const bool more = z + 1 < zb;
      const float* ap = sbuf + (rel & 1) * C::SPLANE + aoff;
      const float* bp[C::NTAPW];
#pragma unroll
      for (int i = 0; i < C::NTAPW; ++i) bp[i] = ring + ((q0 + tapdz[i]) % C::NRING) * C::BPLANE + tapoff[i];
      const int s3 = (q0 + 3) % C::NRING, s4 = (q0 + 4) % C::NRING;
      if constexpr (!V16) {
        if (more) {
          stage_b_rows(2 * z + 2, s3);
          stage_b_rows(2 * z + 3, s4);
          stage_s_rows(z + 1, (rel + 1) & 1);
        }
      }
      float af[2], bf[2][C::NTAPW];
      auto load_frag = [&](int q, float& a, float (&bq)[C::NTAPW]) {
        const int r = q / (C::TX / 2), c = q % (C::TX / 2);
        a = ap[r * C::TX + 2 * c];
#pragma unroll
        for (int i = 0; i < C::NTAPW; ++i) bq[i] = bp[i][2 * r * C::P + 4 * c];
      };
      load_frag(0, af[0], bf[0]);
#pragma unroll
      for (int q = 0; q < C::KSTEPS; ++q) {
        if (q + 1 < C::KSTEPS) load_frag(q + 1, af[(q + 1) & 1], bf[(q + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < C::NTAPW; ++i) acc[i] = DMB_MFMA(af[q & 1], bf[q & 1][i], acc[i]);
        if (V16 && more) {   // the next planes, one copy per k-step
          if (q < C::IB)
            stage_b1(2 * z + 2, s3, q);
          else if (q < 2 * C::IB)
            stage_b1(2 * z + 3, s4, q - C::IB);
          else if (q < 2 * C::IB + C::IS)
            stage_s1(z + 1, (rel + 1) & 1, q - 2 * C::IB);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      q0 = (q0 + 2) % C::NRING;
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
  }

  float* wsb = ws + ((size_t)blockIdx.y * nslots + slot) * 27 * 1024;
#pragma unroll
  for (int i = 0; i < C::NTAPW; ++i) {
    const int t = wave * C::NTAPW + i;
    if (t < 27) {
#pragma unroll
      for (int r = 0; r < 16; ++r) wsb[t * 1024 + cd_row(r, kk) * 32 + n] = acc[i][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// One output channel (the classifier heads, nn.Conv3d(C, 1, 3, 1, 1)): dw[ci, tap] = sum_u x[ci, u] * dy[u - tap + 1].
// Round 5: on the matrix cores.  It is a GEMM with M = 32 input channels, N = 27 taps (one 32 x 32 tile) and K = the voxels:
// A[ci][u] = x[ci][u], B[u][tap] = dy[u - tap + 1] -- the 27 shifted views of ONE small dy neighbourhood.  A workgroup walks
// tiles of 4 rows x 64 columns of one (b, z) plane, one row per wave: the x tile (32 channels x 4 x 64, read coalesced, 256 B per
// channel row) and the dy neighbourhood (3 planes x 6 rows x 66 columns) are fetched into registers one tile AHEAD, committed to
// LDS between two barriers, and a wave then issues 32 MFMAs (two voxels per k-step) whose A / B operands are single LDS dwords
// (x pitch 65: conflict-free; taps 27 .. 31 read a zero row).  The round-1 form -- a VALU kernel, 27 address computations and
// cached loads per voxel and channel group -- ran at 0.11 of the HBM rate (0.30 ms per head at the training shape; rocprofv3,
// profiles/r05_train_pmc.csv): 10x the time of its one pass over x.  Partials per workgroup ([ci][tap], the four waves added in
// a fixed order) go to the workspace and are added in a fixed order in FP64 by the reduce kernel: bit-reproducible, no atomics.
// ------------------------------------------------------------------------------------------------------------------
constexpr int C1M_XT = 64, C1M_ROWS = 4, C1M_XP = C1M_XT + 1;        // tile columns, rows (= waves), LDS pitch of an x row
constexpr int C1M_DYP = 68, C1M_DYROWS = 3 * (C1M_ROWS + 2) + 1;     // dy neighbourhood: pitch, rows (+ 1 zero row)
constexpr int C1M_XS = C1M_ROWS * 32 * C1M_XP, C1M_DYS = C1M_DYROWS * C1M_DYP;
constexpr int C1M_XPT = C1M_ROWS * 32 * C1M_XT / 256;                // x words per thread and tile (32)
constexpr int C1M_DYN = 3 * (C1M_ROWS + 2) * (C1M_XT + 2);           // dy words per tile (1188)
constexpr int C1M_DPT = (C1M_DYN + 255) / 256;                       // ... per thread (5)
static_assert(C1M_XS >= 4 * 32 * 33, "the wave reduction reuses the x tile");

__global__ __launch_bounds__(256, 4) void conv3d_c1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 float* __restrict__ ws, int B, int Ci, int D, int H, int W,
                                                                 int ntx, int nty, int ntiles) {
  __shared__ float xs[C1M_XS];
  __shared__ float dys[C1M_DYS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c0 = blockIdx.y * 32;
  const long long HW = (long long)H * W, DHW = (long long)D * HW;
  for (int i = tid; i < C1M_DYP; i += 256) dys[(C1M_DYROWS - 1) * C1M_DYP + i] = 0.f;   // the zero row (never overwritten)

  float xr[C1M_XPT], dr[C1M_DPT];
  auto fetch = [&](int t) {
    const int tx = t % ntx;
    int r = t / ntx;
    const int ty = r % nty;
    r /= nty;
    const int z = r % D, b = r / D;
    const int x0 = tx * C1M_XT, y0 = ty * C1M_ROWS;
    const float* xb = x + ((long long)b * Ci + c0) * DHW + (long long)z * HW;
#pragma unroll
    for (int q = 0; q < C1M_XPT; ++q) {
      const int idx = q * 256 + tid;                  // (row, channel, column): 64 consecutive lanes = one 256-byte run
      const int xx = idx & 63, ci = (idx >> 6) & 31, rr = idx >> 11;
      const int gy = y0 + rr, gx = x0 + xx;
      xr[q] = (c0 + ci < Ci && gy < H && gx < W) ? xb[(long long)ci * DHW + (long long)gy * W + gx] : 0.f;
    }
    const float* db = dy + (long long)b * DHW;
#pragma unroll
    for (int q = 0; q < C1M_DPT; ++q) {
      const int idx = q * 256 + tid;
      const int cc = idx % (C1M_XT + 2), r2 = idx / (C1M_XT + 2), rr = r2 % (C1M_ROWS + 2), zz = r2 / (C1M_ROWS + 2);
      const int gz = z - 1 + zz, gy = y0 - 1 + rr, gx = x0 - 1 + cc;
      dr[q] = (idx < C1M_DYN && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) ? db[(long long)gz * HW + (long long)gy * W + gx] : 0.f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int q = 0; q < C1M_XPT; ++q) {
      const int idx = q * 256 + tid;
      xs[(idx >> 6) * C1M_XP + (idx & 63)] = xr[q];   // row index (rr * 32 + ci) = idx >> 6
    }
#pragma unroll
    for (int q = 0; q < C1M_DPT; ++q) {
      const int idx = q * 256 + tid;
      if (idx < C1M_DYN) dys[(idx / (C1M_XT + 2)) * C1M_DYP + idx % (C1M_XT + 2)] = dr[q];
    }
  };

  // operand addresses of this lane: A = x[channel i][voxel 2 s + k] of the wave's row, B = dy[voxel 2 s + k - tap j + 1]
  const int i = lane & 31, k = lane >> 5;
  const int tz = i / 9, tyy = (i / 3) % 3, txx = i % 3;
  const float* ap = xs + (wave * 32 + i) * C1M_XP + k;
  const float* bp = (i < 27) ? dys + ((2 - tz) * (C1M_ROWS + 2) + wave + 2 - tyy) * C1M_DYP + 2 - txx + k
                             : dys + (C1M_DYROWS - 1) * C1M_DYP + k;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  int t = blockIdx.x;
  if (t < ntiles) fetch(t);
  for (; t < ntiles; t += gridDim.x) {
    __syncthreads();          // every wave is done with the previous tile
    commit();
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);   // lands while this tile is multiplied
#pragma unroll 8
    for (int s = 0; s < C1M_XT / 2; ++s) acc = DMB_MFMA(ap[2 * s], bp[2 * s], acc);
  }
  // the four waves' tiles, added in a fixed order (the x tile's LDS is free now)
  __syncthreads();
  float* red = xs;
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 32 + cd_row(r, k)) * 33 + i] = acc[r];
  __syncthreads();
  for (int e = tid; e < 32 * 27; e += 256) {
    const int ci = e / 27, tp = e - ci * 27;
    if (c0 + ci < Ci)
      ws[((size_t)blockIdx.x * Ci + c0 + ci) * 27 + tp] =
          (red[ci * 33 + tp] + red[(32 + ci) * 33 + tp]) + (red[(64 + ci) * 33 + tp] + red[(96 + ci) * 33 + tp]);
  }
}

// One wave per (channel, tap): the lanes walk the workgroups' partials 64 apart (independent loads; one thread walking all of
// them alone was a chain of 256 .. 1024 dependent-latency loads, 63 us for 864 numbers), each in FP64, then a fixed butterfly.
__global__ __launch_bounds__(256) void conv3d_c1_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Ci, int nchunk) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= Ci * 27) return;
  double s = 0.0;
  for (int c = lane; c < nchunk; c += 64) s += (double)ws[(size_t)c * Ci * 27 + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) dw[i] = (float)s;
}

// dw[co][ci][tap] = sum over slots in FP64, fixed order.  A workgroup owns 64 consecutive elements of one (block, tap) plane and
// its four waves a quarter of the slots each (eight interleaved partial sums per thread: independent loads), combined through LDS
// as (q0 + q1) + (q2 + q3).  (Rounds 1-5: one thread per element walked all 256 slots -- 32 dependent rounds of 8 loads, 9.2 us for
// 28 MB, 25 times per training step; more threads per element, not more bytes, is what it lacked.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Co, int Ci, int nslots, int transposed) {
  __shared__ double sm[4][64];
  const int ncib = cdiv(Ci, 32);
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  long long r = blockIdx.x;
  const int grp = (int)(r & 15);
  r >>= 4;
  const int t = (int)(r % 27);
  const int blk = (int)(r / 27);
  const int m = grp * 2 + (e >> 5), nn = e & 31;
  const int cib = blk % ncib, cob = blk / ncib;
  const int co = cob * 32 + m, ci = cib * 32 + nn;
  const float* p = ws + (size_t)blk * nslots * 27 * 1024 + t * 1024 + m * 32 + nn;
  const int per = (nslots + 3) / 4;
  const int s0 = q * per, s1 = s0 + per < nslots ? s0 + per : nslots;
  double part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int sl = s0;
  for (; sl + 8 <= s1; sl += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) part[u] += (double)p[(size_t)(sl + u) * 27 * 1024];
  }
  for (int u = 0; sl < s1; ++sl, ++u) part[u] += (double)p[(size_t)sl * 27 * 1024];
  sm[q][e] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
  __syncthreads();
  if (q != 0 || co >= Co || ci >= Ci) return;
  const double s = (sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e]);
  if (transposed)
    dw[((size_t)ci * Co + co) * 27 + t] = (float)s;
  else
    dw[((size_t)co * Ci + ci) * 27 + t] = (float)s;
}

static long long cdiv_ll(long long a, long long b) { return (a + b - 1) / b; }
static int wgrad_slots() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

}  // namespace dmb

using namespace dmb;

// slots per (32 x 32) channel block: the CUs are shared between the blocks of one launch
static int wgrad_slots_per_block(int Co, int Ci) {
  const int nblk = cdiv(Co, 32) * cdiv(Ci, 32);
  const int s = wgrad_slots() / nblk;
  return s < 1 ? 1 : s;
}

extern "C" long long dmb_conv3d_wgrad_workspace_floats(int Co, int Ci) {
  if (Co <= 0 || Ci <= 0) return 0;
  return (long long)cdiv(Co, 32) * cdiv(Ci, 32) * wgrad_slots_per_block(Co, Ci) * 27 * 1024;
}

extern "C" int dmb_conv3d_k3_wgrad_f32(const float* x, const float* dc, float* dw, float* workspace, int B, int Ci, int Co,
                                       int D, int H, int W, void* stream) {
  if (!x || !dc || !dw || !workspace || B <= 0 || Ci <= 0 || Co <= 0 || D <= 0 || H <= 0 || W <= 0)
    return fail(DMB_EINVAL, "conv3d_wgrad: bad argument");
  if ((long long)32 * D * H * W * 4 >= 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv3d_wgrad: 32 channels of one batch item must stay below 2 GiB");
  hipStream_t st = (hipStream_t)stream;
  if (Co == 1) {   // classifier heads: the single-output-channel form (M = channels, N = taps, K = voxels on the matrix cores)
    const int ntx = cdiv(W, C1M_XT), nty = cdiv(H, C1M_ROWS);
    const long long nt = (long long)B * D * ntx * nty;
    if (nt > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv3d_wgrad (1 channel): too many tiles");
    // persistent grid: four workgroups per CU, bounded by the tiles and by the caller's workspace ([workgroup][Ci][27] partials)
    long long g = 4LL * wgrad_slots();
    const long long cap = dmb_conv3d_wgrad_workspace_floats(Co, Ci) / ((long long)Ci * 27);
    if (g > nt) g = nt;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(conv3d_c1_wgrad_kernel, dim3((unsigned)g, cdiv(Ci, 32)), dim3(256), 0, st, x, dc, workspace, B, Ci, D, H, W, ntx, nty, (int)nt);
    hipLaunchKernelGGL(conv3d_c1_wgrad_reduce_kernel, dim3(cdiv(Ci * 27, 4)), dim3(256), 0, st, workspace, dw, Ci, (int)g);
    return launch_status("conv3d_wgrad (1 channel) launch failed");
  }
  const int nblk = cdiv(Co, 32) * cdiv(Ci, 32);
  // 24- or 32-column tiles: whichever covers W with fewer computed columns (W = 128 of the training crops: 128 against 144)
  const bool wide = cdiv(W, 32) * 32 < cdiv(W, 24) * 24;
  const int TX = wide ? 32 : 24;
  const int ntx = cdiv(W, TX), nty = cdiv(H, 4);
  // z segments: the split that minimises rounds x (planes per item + prologue); a round = one item on every slot
  const int nslots = wgrad_slots_per_block(Co, Ci);
  int zseg = D;
  {
    double best = 1e30;
    for (int nz = 1; nz <= D; ++nz) {
      const int zs = cdiv(D, nz);
      if (zs < 4 && nz > 1) break;
      const double cost = (double)cdiv_ll((long long)B * ntx * nty * cdiv(D, zs), nslots) * (zs + 0.5);
      if (cost < best - 1e-9) {
        best = cost;
        zseg = zs;
      }
    }
  }
  if (DMB_OPT(5) > 0) zseg = cdiv(D, DMB_OPT(5));   // development knob: number of z segments
  const int nzs = cdiv(D, zseg);
  const bool v16 = W % 4 == 0 && (((uintptr_t)x | (uintptr_t)dc) & 15) == 0;
  DMB_ENSURE_LDS((&conv3d_wgrad_s1_kernel<true, 24>), (size_t)(WgCfg<24>::LDS_FLOATS * 4));
  DMB_ENSURE_LDS((&conv3d_wgrad_s1_kernel<true, 32>), (size_t)(WgCfg<32>::LDS_FLOATS * 4));
  DMB_ENSURE_LDS((&conv3d_wgrad_s1_kernel<false, 24>), (size_t)(WgCfg<24>::LDS_FLOATS * 4));
  const long long items3 = (long long)B * ntx * nty * nzs;
  const int nused = (int)(items3 < nslots ? items3 : nslots);   // small layers: no idle slots to write and add zeros for
  const dim3 grid((unsigned)nused, (unsigned)nblk);
  if (v16 && wide)
    hipLaunchKernelGGL((conv3d_wgrad_s1_kernel<true, 32>), grid, dim3(256), WgCfg<32>::LDS_FLOATS * 4, st, x, dc, workspace, B, Ci, Co, D, H, W, ntx, nty, nzs, zseg);
  else if (v16)
    hipLaunchKernelGGL((conv3d_wgrad_s1_kernel<true, 24>), grid, dim3(256), WgCfg<24>::LDS_FLOATS * 4, st, x, dc, workspace, B, Ci, Co, D, H, W, ntx, nty, nzs, zseg);
  else
    hipLaunchKernelGGL((conv3d_wgrad_s1_kernel<false, 24>), grid, dim3(256), WgCfg<24>::LDS_FLOATS * 4, st, x, dc, workspace, B, Ci, Co, D, H, W, cdiv(W, 24), nty, nzs, zseg);
  int rc = launch_status("conv3d_wgrad launch failed");
  if (rc != DMB_OK) return rc;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblk * 27 * 16), dim3(256), 0, st, workspace, dw, Co, Ci, nused, 0);
  return launch_status("conv3d_wgrad reduce launch failed");
}

constexpr double WG2_EFF_4x8 = 1.1;   // rate of the 4 x 8 tile relative to 2 x 12 on shapes both tile exactly (calibrated below)

extern "C" int dmb_conv3d_k3s2_wgrad_f32(const float* small, const float* big, float* dw, float* workspace, int B, int Cs, int Cb,
                                         int Ds, int Hs, int Ws, int Db, int Hb, int Wb, void* stream) {
  if (!small || !big || !dw || !workspace || B <= 0 || Cs <= 0 || Cb <= 0 || Ds <= 0 || Hs <= 0 || Ws <= 0)
    return fail(DMB_EINVAL, "conv3d_s2_wgrad: bad argument");
  if ((Db != 2 * Ds && Db != 2 * Ds - 1) || (Hb != 2 * Hs && Hb != 2 * Hs - 1) || (Wb != 2 * Ws && Wb != 2 * Ws - 1))
    return fail(DMB_EINVAL, "conv3d_s2_wgrad: the big tensor must be 2n or 2n - 1 per axis");
  if ((long long)32 * Db * Hb * Wb * 4 >= 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv3d_s2_wgrad: 32 channels of one batch item must stay below 2 GiB");
  const bool v16 = Ws % 4 == 0 && Wb % 4 == 0 && (((uintptr_t)small | (uintptr_t)big) & 15) == 0;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = cdiv(Cs, 32) * cdiv(Cb, 32);
  const int nslots = wgrad_slots_per_block(Cs, Cb);
  // z segments and the launch's cost for a tile shape: rounds of items on the slots x (planes per item + prologue) x the tile's
  // k-steps, over what the tile reaches between its barriers (4 x 8: 112 MFMAs per wave and barrier against 84; measured at the
  // training crop, profiles/r06_wgrad_s2_tiles.log)
  auto plan = [&](int TY, int TX, double eff, int& zseg_out) {
    const int ntx_ = cdiv(Ws, TX), nty_ = cdiv(Hs, TY);
    double best = 1e30;
    int zseg_ = Ds;
    for (int nz = 1; nz <= Ds; ++nz) {
      const int zs = cdiv(Ds, nz);
      if (zs < 4 && nz > 1) break;
      const double cost = (double)cdiv_ll((long long)B * ntx_ * nty_ * cdiv(Ds, zs), nslots) * (zs + 1.0);
      if (cost < best - 1e-9) {
        best = cost;
        zseg_ = zs;
      }
    }
    zseg_out = zseg_;
    return best * (TY * TX) / eff;
  };
  int zseg_a, zseg_b;
  const double cost_a = plan(2, 12, 1.0, zseg_a), cost_b = plan(4, 8, WG2_EFF_4x8, zseg_b);
  bool wide = cost_b < cost_a;            // the 4 x 8 tile
  if (DMB_OPT(27) == 1) wide = false;     // (development build: force a tile)
  if (DMB_OPT(27) == 2) wide = true;
  const int zseg = wide ? zseg_b : zseg_a;
  const int ntx = cdiv(Ws, wide ? 8 : 12), nty = cdiv(Hs, wide ? 4 : 2);
  const int nzs = cdiv(Ds, zseg);
  const long long items2 = (long long)B * ntx * nty * nzs;
  const int nused = (int)(items2 < nslots ? items2 : nslots);
#define DMB_WG2(CFG, V)                                                                                                                  \
  do {                                                                                                                                  \
    DMB_ENSURE_LDS((&conv3d_wgrad_s2_kernel<CFG, V>), (size_t)(CFG::LDS_FLOATS * 4));                                                    \
    hipLaunchKernelGGL((conv3d_wgrad_s2_kernel<CFG, V>), dim3((unsigned)nused, (unsigned)nblk), dim3(256), CFG::LDS_FLOATS * 4, st, small, \
                       big, workspace, B, Cs, Cb, Ds, Hs, Ws, Db, Hb, Wb, ntx, nty, nzs, zseg);                                         \
  } while (0)
  typedef Wg2Cfg<2, 12> WgA;
  typedef Wg2Cfg<4, 8> WgB;
  if (wide) {
    if (v16) DMB_WG2(WgB, true); else DMB_WG2(WgB, false);
  } else {
    if (v16) DMB_WG2(WgA, true); else DMB_WG2(WgA, false);
  }
#undef DMB_WG2
  int rc = launch_status("conv3d_s2_wgrad launch failed");
  if (rc != DMB_OK) return rc;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblk * 27 * 16), dim3(256), 0, st, workspace, dw, Cs, Cb, nused, 0);
  return launch_status("conv3d_s2_wgrad reduce launch failed");
}

// ------------------------------------------------------------------------------------------------------------------
// 2-D: dW[co, ci, tap] = sum_{b, p} dc[b, co, p] * x[b, ci, p + tap - 1] for nn.Conv2d(k 3, s 1, p 1) -- AcfNet's confidence
// heads (cmn/cmn.py:21-36: 192 -> 64 over the full-resolution cost volume).  Same idea as the 3-D kernel, one dimension down:
// a workgroup owns a (32 x 32) channel block and walks strips of 64 columns along y (in steps of the dilation: the image is
// `dilation` interleaved row classes, so a tap's rows are neighbours inside the class); x rows sit in a ring of four LDS slots,
// dc rows are double buffered; the four waves split the strip's columns (16 each) and each holds all nine taps (144
// accumulator registers); their partial results are added through LDS at the end, the slots' by the reduction kernel.
// ------------------------------------------------------------------------------------------------------------------
namespace dmb {

template <int KS_, int DIL_>
struct Wg2dCfg {
  static constexpr int KS = KS_, DIL = DIL_, NTAP = KS * KS;
  static constexpr int HALO = DIL * (KS / 2);                  // columns either side
  static constexpr int PAD = HALO <= 4 ? 4 : 8;                // the staged row starts at the aligned column x0 - PAD
  static constexpr int TX = 64, XOFF = PAD - HALO;
  // Rows are walked in steps of DIL (the image is DIL interleaved row classes; taps reach rows y - DIL, y, y + DIL = the
  // neighbours INSIDE the class), so the ring is four slots whatever the dilation: three in use, one landing.
  static constexpr int NR = KS == 1 ? 2 : 4;
  static constexpr int P = (PAD + TX + HALO + 3) / 4 * 4;
  static constexpr int UXU = P / 4, UDU = TX / 4;              // 16-byte units per channel row
  static constexpr int UX = UXU | 1, UD = UDU | 1;             // ... made odd: pitch = 4 mod 8 floats
  static constexpr int SX = UX * 4, SD = UD * 4;
  static constexpr int XROW = 32 * SX, DROW = 32 * SD;
  static constexpr int LDS_FLOATS = NR * XROW + 2 * DROW;
  static constexpr int IX = (32 * UX + 255) / 256, ID = (32 * UD + 255) / 256;   // copy instructions per wave and row
  static constexpr int KSTEPS = TX / 4 / 2;                    // per wave: 16 columns = 8 k-steps
  static_assert(HALO <= 8 && SX % 8 == 4 && SD % 8 == 4 && IX + ID <= KSTEPS && LDS_FLOATS * 4 * 2 <= 160 * 1024, "tile");
};

// nys: segments per row class; yseg: class rows per segment.  An item = (batch item, column strip, row class, segment).
template <int KS_, int DIL_>
__global__ __launch_bounds__(256, 2) void conv2d_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dc,
                                                              float* __restrict__ ws, int B, int Ci, int Co, int H, int W, int ntx,
                                                              int nys, int yseg) {
  typedef Wg2dCfg<KS_, DIL_> C;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xring = lds;
  float* dbuf = lds + C::NR * C::XROW;
  const int ncib = cdiv(Ci, 32);
  const int cib = blockIdx.y % ncib, cob = blockIdx.y / ncib;
  const int slot = xcd_remap(blockIdx.x, gridDim.x), nslots = gridDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n = lane & 31, kk = lane >> 5;
  const unsigned HW = (unsigned)H * W;
  const int items = B * nys * C::DIL * ntx;
  const int nci = min(32, Ci - cib * 32), nco = min(32, Co - cob * 32);
  constexpr int RH = C::KS / 2;   // class rows either side (0 or 1)

  f32x16 acc[C::NTAP];
#pragma unroll
  for (int i = 0; i < C::NTAP; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int aoff = n * C::SD + wave * 16 + kk;
  const int boff = n * C::SX + wave * 16 + kk + C::XOFF;

  for (int it = slot; it < items; it += nslots) {
    int t = it;
    const int tx = t % ntx;
    t /= ntx;
    const int rc = t % C::DIL;        // row class: image rows rc, rc + DIL, rc + 2 DIL, ...
    t /= C::DIL;
    const int ys = t % nys;
    const int b = t / nys;
    const int x0 = tx * C::TX, ia = ys * yseg, ib = ia + yseg;   // class-row range [ia, ib); rows past H read as zeros
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(x + ((size_t)b * Ci + cib * 32) * HW, (unsigned)nci * HW * 4u);
    const __amdgpu_buffer_rsrc_t drs = make_rsrc(dc + ((size_t)b * Co + cob * 32) * HW, (unsigned)nco * HW * 4u);
    unsigned xvo[C::IX], dvo[C::ID];   // channel + column part of this wave's copy instructions; the row is a scalar offset
#pragma unroll
    for (int j = 0; j < C::IX; ++j) {
      const int u = (j * 4 + wave) * 64 + lane, ch = u / C::UX, un = u - ch * C::UX;
      const int gx = x0 - C::PAD + 4 * un;
      xvo[j] = (ch < 32 && un < C::UXU && gx >= 0 && gx < W) ? ((unsigned)ch * HW + (unsigned)gx) * 4u : DMA_OOB;
    }
#pragma unroll
    for (int j = 0; j < C::ID; ++j) {
      const int u = (j * 4 + wave) * 64 + lane, ch = u / C::UD, un = u - ch * C::UD;
      const int gx = x0 + 4 * un;
      dvo[j] = (ch < 32 && un < C::UDU && gx < W) ? ((unsigned)ch * HW + (unsigned)gx) * 4u : DMA_OOB;
    }
    auto stage_x1 = [&](int ci_row, int rslot, int j) {   // class row -> image row rc + DIL * ci_row
      const int gy = rc + C::DIL * ci_row;
      const bool ok = ci_row >= 0 && gy < H;
      const int first = (j * 4 + wave) * 64;
      if (first + lane < 32 * C::UX) dma16(xrs, ok ? xvo[j] : DMA_OOB, ok ? (unsigned)gy * W * 4u : 0u, xring + rslot * C::XROW + first * 4);
    };
    auto stage_d1 = [&](int ci_row, int buf, int j) {
      const int gy = rc + C::DIL * ci_row;
      const bool ok = ci_row < ib && gy < H;
      const int first = (j * 4 + wave) * 64;
      if (first + lane < 32 * C::UD) dma16(drs, ok ? dvo[j] : DMA_OOB, ok ? (unsigned)gy * W * 4u : 0u, dbuf + buf * C::DROW + first * 4);
    };
    // ring slot of class row j: (j - (ia - RH)) % NR
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2 * RH + 1; ++q)
#pragma unroll
      for (int j = 0; j < C::IX; ++j) stage_x1(ia - RH + q, q, j);
#pragma unroll
    for (int j = 0; j < C::ID; ++j) stage_d1(ia, 0, j);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    int r0 = 0;   // ring index of class row i - RH
    for (int i = ia; i < ib; ++i) {
      const int rel = i - ia;
      const bool more = i + 1 < ib;
      const float* ap = dbuf + (rel & 1) * C::DROW + aoff;
      const float* bp[C::KS];
#pragma unroll
      for (int ty = 0; ty < C::KS; ++ty) bp[ty] = xring + ((r0 + ty) % C::NR) * C::XROW + boff;
      const int rnew = (r0 + 2 * RH + 1) % C::NR;
      float af[2], bf[2][C::NTAP];
      auto load_frag = [&](int q, float& a, float (&bq)[C::NTAP]) {
        a = ap[2 * q];
#pragma unroll
        for (int tt = 0; tt < C::NTAP; ++tt) bq[tt] = bp[tt / C::KS][2 * q + (tt % C::KS) * C::DIL];
      };
      load_frag(0, af[0], bf[0]);
#pragma unroll
      for (int q = 0; q < C::KSTEPS; ++q) {
        if (q + 1 < C::KSTEPS) load_frag(q + 1, af[(q + 1) & 1], bf[(q + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < C::NTAP; ++tt) acc[tt] = DMB_MFMA(af[q & 1], bf[q & 1][tt], acc[tt]);
        if (more) {
          if (q < C::IX)
            stage_x1(i + RH + 1, rnew, q);
          else if (q < C::IX + C::ID)
            stage_d1(i + 1, (rel + 1) & 1, q - C::IX);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      r0 = (r0 + 1) % C::NR;
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
  }
  // the four waves' partial results are added through LDS (wave 0 stores, the others add in turn) before one copy per
  // workgroup goes to the workspace
  __syncthreads();
  float* red = lds;   // NTAP x 1024 floats <= the ring
  static_assert(C::NTAP * 1024 <= C::LDS_FLOATS, "reduction scratch");
#pragma unroll
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int tt = 0; tt < C::NTAP; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* q = red + tt * 1024 + cd_row(r, kk) * 32 + n;
          *q = wv == 0 ? acc[tt][r] : *q + acc[tt][r];
        }
    }
    __syncthreads();
  }
  float* wsb = ws + ((size_t)blockIdx.y * nslots + slot) * C::NTAP * 1024;
  for (int i = threadIdx.x; i < C::NTAP * 1024; i += 256) wsb[i] = red[i];
}

__global__ void conv2d_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Co, int Ci, int nparts, int ntap) {
  const int ncib = cdiv(Ci, 32);
  const long long total = (long long)cdiv(Co, 32) * ncib * ntap * 1024;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int nn = (int)(i & 31), m = (int)((i >> 5) & 31);
    long long r = i >> 10;
    const int t = (int)(r % ntap);
    const int blk = (int)(r / ntap);
    const int cib = blk % ncib, cob = blk / ncib;
    const int co = cob * 32 + m, ci = cib * 32 + nn;
    if (co >= Co || ci >= Ci) continue;
    const float* p = ws + (size_t)blk * nparts * ntap * 1024 + t * 1024 + m * 32 + nn;
    double part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int sl = 0;
    for (; sl + 8 <= nparts; sl += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) part[u] += (double)p[(size_t)(sl + u) * ntap * 1024];
    }
    for (; sl < nparts; ++sl) part[sl & 7] += (double)p[(size_t)sl * ntap * 1024];
    dw[((size_t)co * Ci + ci) * ntap + t] = (float)(((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7])));
  }
}

}  // namespace dmb

static int wgrad2d_slots_per_block(int Co, int Ci) {
  const int nblk = cdiv(Co, 32) * cdiv(Ci, 32);
  const int s = 2 * wgrad_slots() / nblk;   // two workgroups per CU
  return s < 1 ? 1 : s;
}

extern "C" long long dmb_conv2d_wgrad_workspace_floats(int Co, int Ci) {
  if (Co <= 0 || Ci <= 0) return 0;
  return (long long)cdiv(Co, 32) * cdiv(Ci, 32) * wgrad2d_slots_per_block(Co, Ci) * 9 * 1024;
}

template <int KS, int DIL>
static int launch_wgrad2d(const float* x, const float* dc, float* dw, float* workspace, int B, int Ci, int Co, int H, int W, hipStream_t st) {
  typedef Wg2dCfg<KS, DIL> C;
  const int nblk = cdiv(Co, 32) * cdiv(Ci, 32), nslots = wgrad2d_slots_per_block(Co, Ci);
  const int ntx = cdiv(W, C::TX);
  const int HC = cdiv(H, DIL);   // rows of one row class
  int yseg = HC;
  {
    double best = 1e30;
    for (int ny = 1; ny <= HC; ++ny) {
      const int ysz = cdiv(HC, ny);
      if (ysz < 8 && ny > 1) break;
      const double cost = (double)cdiv_ll((long long)B * ntx * DIL * cdiv(HC, ysz), nslots) * (ysz + 2.0);
      if (cost < best - 1e-9) {
        best = cost;
        yseg = ysz;
      }
    }
  }
  const int nys = cdiv(HC, yseg);
  DMB_ENSURE_LDS((&conv2d_wgrad_kernel<KS, DIL>), (size_t)(C::LDS_FLOATS * 4));
  const long long items = (long long)B * ntx * DIL * nys;
  const int nused = (int)(items < nslots ? items : nslots);   // small layers: no idle slots to write and add zeros for
  hipLaunchKernelGGL((conv2d_wgrad_kernel<KS, DIL>), dim3((unsigned)nused, (unsigned)nblk), dim3(256), C::LDS_FLOATS * 4, st, x, dc, workspace, B,
                     Ci, Co, H, W, ntx, nys, yseg);
  int rc = launch_status("conv2d_wgrad launch failed");
  if (rc != DMB_OK) return rc;
  hipLaunchKernelGGL(conv2d_wgrad_reduce_kernel, dim3(cdiv(nblk * C::NTAP * 1024, 256)), dim3(256), 0, st, workspace, dw, Co, Ci, nused, C::NTAP);
  return launch_status("conv2d_wgrad reduce launch failed");
}

extern "C" int dmb_conv2d_wgrad_f32(const float* x, const float* dc, float* dw, float* workspace, int B, int Ci, int Co, int H, int W,
                                    int ksize, int dilation, void* stream) {
  if (!x || !dc || !dw || !workspace || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "conv2d_wgrad: bad argument");
  if ((long long)32 * H * W * 4 >= 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv2d_wgrad: 32 channels of one image must stay below 2 GiB");
  if (W % 4 != 0 || (((uintptr_t)x | (uintptr_t)dc) & 15) != 0)
    return fail(DMB_EUNSUPPORTED, "conv2d_wgrad: the width must be a multiple of 4 and the tensors 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (ksize == 3 && dilation == 1) return launch_wgrad2d<3, 1>(x, dc, dw, workspace, B, Ci, Co, H, W, st);
  if (ksize == 3 && dilation == 2) return launch_wgrad2d<3, 2>(x, dc, dw, workspace, B, Ci, Co, H, W, st);
  if (ksize == 3 && dilation == 4) return launch_wgrad2d<3, 4>(x, dc, dw, workspace, B, Ci, Co, H, W, st);
  if (ksize == 3 && dilation == 8) return launch_wgrad2d<3, 8>(x, dc, dw, workspace, B, Ci, Co, H, W, st);
  if (ksize == 1) return launch_wgrad2d<1, 1>(x, dc, dw, workspace, B, Ci, Co, H, W, st);
  return fail(DMB_EUNSUPPORTED, "conv2d_wgrad: kernel 1, or kernel 3 with dilation 1, 2, 4 or 8 (stride 1)");
}
