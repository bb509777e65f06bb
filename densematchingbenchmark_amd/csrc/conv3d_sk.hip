// 3x3x3 convolution (stride 1 or 2) for launches that do NOT fill the chip: the K chain of a tile split over the waves of a
// workgroup (split-K), round 6.
//
// Why.  conv3d_s1_kernel / conv3d_s2_kernel (conv3d.hip) give one wave the whole chain of 27 Ci / 2 MFMAs of its 32 x 32 tiles and
// walk the input channels in chunks of two behind a workgroup barrier each.  On a full grid that is the fastest form (0.9 of the
// FP32 matrix peak); on the deepest levels of the hourglass for ONE small pair (BASELINE configs[0]: [1, 64, 4, 16, 32] = 128 tiles
// for 1024 SIMDs; hourglass.py:62-86 called as apis/inference.py:191-225 does, one pair per call) a launch takes 31-35 us whatever
// its arithmetic (2.9 us at the matrix peak): 32 chunks x (one LDS-DMA round trip + a barrier) with nothing else resident on the
// CU to hide them, on an eighth of the chip.
//
// Here a workgroup of NW (8) waves owns ONE output tile -- MT column tiles of 16 x  x  2 y voxels of one z-slice, NT 32-channel row
// tiles -- and wave w multiplies input channels [w Ci / NW, (w + 1) Ci / NW) only:
//   * every wave stages the haloed input tile of ITS channels with one burst of 16-byte LDS-DMA copies into its own LDS region
//     (whole K range resident: 64 channels x 3 x 4 x 24 floats = 74 KB for a stride-1 tile): ONE memory round trip, one barrier;
//   * no wave shares weights with another one, so the A fragments (prepacked, one 256-byte line per k-step and row tile) come
//     straight from L2 into registers -- all of the wave's (at most 4 channel pairs x 27 taps), requested next to the copies;
//   * the NW partial tiles are summed in LDS in ascending wave order (a fixed order: results are reproducible run to run), then
//     the shared epilogue (BatchNorm affine, skip operand, ReLU in the reference's order: layers/basic_layers.py:68-100) stores
//     16-byte words.
// The sum of a voxel is therefore NW partial fma chains added in order, not one chain: results differ from the full-grid kernels
// in the last bits (and with them batch 1 from batch 4 where this form is picked) -- both are FP32 evaluations of the same
// convolution; the tests bound them against the FP64 yardstick instead of against each other.
//
// Which launches take this form is a cost estimate in dmb_conv3d_k3_f32 (conv3d.hip); conv3d_sk_try returns -1 where the shape
// does not qualify.
#include "dmb_common.h"

namespace dmb {

// KZ = 3: the 3x3x3 convolution; KZ = 1: a 3x3 convolution of [B, C, H, W] maps (D = 1, no z taps) -- the 2-D layers of the backbones
// (layers/basic_layers.py:12-66, backbones/PSMNet.py:8-129) for ONE small image pair, where conv2d.hip's persistent tiles leave most
// of the chip idle in the same way (31 of PSMNet's layers at [1, 64, 64, 128]: 31.7 us each for 3.8 us of MFMAs).  DIL: dilation
// (stride 1 only; 2 = PSMNet's layer4).
template <int S_, int NW_, int NT_, int XS_, int YS_, int NPR_, int KZ_ = 3, int DIL_ = 1>
struct SKCfg {
  static constexpr int S = S_, NW = NW_, NT = NT_, XS = XS_, YS = YS_, KZ = KZ_, DIL = DIL_;
  static constexpr int TAPS = KZ * 9;
  static constexpr int NPR = NPR_;   // channel pairs per wave whose A fragments are resident in registers (Ci <= 2 NW NPR)
  static constexpr int NTHREADS = 64 * NW;
  static constexpr int TXO = 16 * XS, TYO = 2 * YS, MT = XS * YS;   // output tile; 32-voxel column tiles = row pairs of 16 columns
  static constexpr int ROWS = S == 1 ? TYO + 2 * DIL : 2 * TYO + 1;                   // staged input rows
  static constexpr int P = S == 1 ? (4 + TXO + DIL + 3) / 4 * 4 : 2 * TXO + 4;       // staged row: from the 16-byte aligned column S x0 - 4
  static constexpr int XOFF = S == 1 ? 4 - DIL : 3;                                  // staged column of the first tap of output column 0
  static constexpr int ZS = KZ;
  static_assert(DIL == 1 || S == 1, "dilated layers are stride 1");
  static_assert(DIL <= 4, "the staged row starts 4 columns left of the tile");
  static constexpr int PLANE = ROWS * P, CHS = ZS * PLANE;         // one channel of the tile (a whole number of 16-byte units)
  static constexpr int UPR = P / 4, UPC = ZS * ROWS * UPR;         // 16-byte units per row / per channel
  static constexpr int PP = MT * 32 + 4;                           // pitch of a partial tile's rows
  static constexpr int PART = NT * 32 * PP;                        // floats of one wave's partial sums
  static_assert(P % 4 == 0 && CHS % 4 == 0, "16-byte units");
};

// The kernel is PERSISTENT: workgroup g walks work items g, g + G, ... (item = voxel tile x row-tile group; G is a multiple of the
// row-tile groups, so a workgroup keeps ITS row tiles and with them its A fragments: the weights are read once per workgroup, not
// once per tile -- at 1500 tile chains the one-item-per-workgroup form re-read 0.34 GB of weights from L2 and ran at 0.45 of the
// matrix peak).  With fewer items than workgroup slots every workgroup has exactly one item.  Where two input tiles fit the LDS
// the next item's copies are in flight while the current one is multiplied.
template <class C>
__global__ __launch_bounds__(C::NTHREADS) void conv3d_sk_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ res, float* __restrict__ y, int Ci, int D,
                                                                int H, int W, int Do, int Ho, int Wo, int ntx, int nty, int NTT,
                                                                int relu, int items, int buf_floats, int nbuf, long long in_bs,
                                                                long long out_bs, long long res_bs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  const int NTS = NTT / C::NT;
  const int nt0 = (wg % NTS) * C::NT;   // first 32-channel row tile of this workgroup (innermost: the row tiles of a voxel tile share its input in L2)

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const unsigned HW = (unsigned)H * W, DHW = (unsigned)D * HW;
  const int CW = Ci / C::NW;           // input channels of this wave (even: checked on the host)
  const int c0 = wave * CW;

  struct Tile {
    int b, z0, y0, x0;
  };
  auto tile_of = [&](int item) {
    int t = item / NTS;
    Tile tl;
    tl.x0 = (t % ntx) * C::TXO;
    t /= ntx;
    tl.y0 = (t % nty) * C::TYO;
    t /= nty;
    tl.z0 = t % Do;
    tl.b = t / Do;
    return tl;
  };
  // ---- staging: one burst per item; unit u = 4 consecutive floats of a staged row, the units of the wave's channels are linear in
  // its own part of the buffer
  auto stage = [&](const Tile& tl, float* buf) {
    float* region = buf + c0 * C::CHS;
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(x + (size_t)tl.b * in_bs + (size_t)c0 * DHW, (unsigned)CW * DHW * 4u);
    const int NU = CW * C::UPC;
    for (int u0 = 0; u0 < NU; u0 += 64) {
      const int u = u0 + lane;
      const int cl = u / C::UPC, r0 = u - cl * C::UPC;
      const int zz = r0 / (C::ROWS * C::UPR), r1 = r0 - zz * (C::ROWS * C::UPR), yy = r1 / C::UPR, sg = r1 - yy * C::UPR;
      const int gz = C::S * tl.z0 - (C::KZ == 3 ? 1 : 0) + zz, gy = C::S * tl.y0 - (C::S == 1 ? C::DIL : 1) + yy, gx = C::S * tl.x0 - 4 + sg * 4;
      const bool ok = gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
      if (u < NU)
        dma16(xrs, ok ? ((unsigned)cl * DHW + (unsigned)gz * HW + (unsigned)gy * W + (unsigned)gx) * 4u : DMA_OOB, 0u, region + u0 * 4);
    }
  };
  if (wg >= items) return;
  stage(tile_of(wg), lds);

  // ---- A fragments: ALL of the wave's channel pairs (27 taps x NT row tiles each) are requested once, next to the first item's
  // copies: one memory round trip for the workgroup's whole weight set.  (Requested a pair ahead of their MFMAs they arrived late:
  // a pair's 27 MFMAs last 0.7 us, an L2 round trip under load longer -- 16 us instead of 13 for the [1, 64, 4, 16, 32] layer.)
  const int NP = CW / 2;   // <= NPR (checked on the host)
  const float* wbase = wp + ((size_t)(c0 / 2) * C::TAPS * NTT + nt0) * 64 + lane;
  float a[C::NPR][C::TAPS][C::NT];
#pragma unroll
  for (int p = 0; p < C::NPR; ++p)
    if (p < NP) {
      const float* wq = wbase + (size_t)p * C::TAPS * NTT * 64;
#pragma unroll
      for (int tap = 0; tap < C::TAPS; ++tap)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) a[p][tap][nt] = wq[(tap * NTT + nt) * 64];
    }

  const unsigned HWo = (unsigned)Ho * Wo, DHWo = (unsigned)Do * HWo;
  const bool vec = (Wo & 3) == 0;
  // epilogue operands that do not depend on the item (a thread owns the same rows of every output tile): requested here, under the
  // first item's copies, not after its multiply phase (an exposed L2 round trip per launch in the one-item case)
  constexpr int NE4 = C::NT * 32 * C::MT * 8;   // 16-byte words of the workgroup's output tile
  constexpr int NEPT = (NE4 + C::NTHREADS - 1) / C::NTHREADS;
  float scv[NEPT], shv[NEPT];
#pragma unroll
  for (int q = 0; q < NEPT; ++q) {
    const int e = threadIdx.x + q * C::NTHREADS, co = nt0 * 32 + (e < NE4 ? e / (C::MT * 8) : 0);
    scv[q] = scale ? scale[co] : 1.f;
    shv[q] = shift ? shift[co] : 0.f;
  }
  int it = 0;
  for (int item = wg; item < items; item += G, ++it) {
    const Tile tl = tile_of(item);
    float* buf = lds + (nbuf == 2 ? (it & 1) * buf_floats : 0);
    __syncthreads();   // (the compiler drains the copies -- vmcnt(0) -- here): this item's tile is in `buf`; the other buffer is free
    if (nbuf == 2 && item + G < items) stage(tile_of(item + G), lds + ((it + 1) & 1) * buf_floats);   // lands while this item is multiplied
    float* yb = y + (size_t)tl.b * out_bs;
    const float* rb = res ? res + (size_t)tl.b * res_bs : nullptr;
    // this thread's words of the output tile and its skip operand (requested now, consumed after the multiply phase)
    size_t off[NEPT];
    bool live[NEPT];
    float4 rv[NEPT];
#pragma unroll
    for (int q = 0; q < NEPT; ++q) {
      const int e = threadIdx.x + q * C::NTHREADS;
      const int row = e / (C::MT * 8), c4 = e - row * (C::MT * 8), mt = c4 >> 3, jj = (c4 & 7) * 4;
      const int gy = tl.y0 + 2 * (mt / C::XS) + (jj >> 4), gx = tl.x0 + 16 * (mt % C::XS) + (jj & 15);
      live[q] = e < NE4 && gy < Ho && gx < Wo;
      off[q] = (size_t)(nt0 * 32 + row) * DHWo + (size_t)tl.z0 * HWo + (size_t)gy * Wo + gx;
      rv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rb && live[q] && vec) rv[q] = *reinterpret_cast<const float4*>(rb + off[q]);
    }

    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // B fragment of (pair p, tap, column tile): lane (j, h) reads channel 2 p + h at the tap's input voxel of output (j / 16, j % 16)
    const float* bb = buf + c0 * C::CHS + h * C::CHS + C::S * (j >> 4) * C::P + C::S * (j & 15) + C::XOFF;
#pragma unroll
    for (int p = 0; p < C::NPR; ++p)
      if (p < NP) {
        const float* bp = bb + 2 * p * C::CHS;
#pragma unroll
        for (int tap = 0; tap < C::TAPS; ++tap) {
          const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
          float bf[C::MT];
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt)
            bf[mt] = bp[dz * C::PLANE + (dy * C::DIL + C::S * 2 * (mt / C::XS)) * C::P + dx * C::DIL + C::S * 16 * (mt % C::XS)];
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt) acc[mt][nt] = DMB_MFMA(a[p][tap][nt], bf[mt], acc[mt][nt]);
        }
      }

    // ---- partial tiles -> the consumed buffer (dead once every wave has left the loop above), summed in ascending wave order
    __syncthreads();
    {
      float* part = buf + wave * C::PART;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) part[(nt * 32 + cd_row(r, h)) * C::PP + mt * 32 + j] = acc[mt][nt][r];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NEPT; ++q) {
      const int e = threadIdx.x + q * C::NTHREADS;
      if (!live[q]) continue;
      const int row = e / (C::MT * 8), c4 = e - row * (C::MT * 8), mt = c4 >> 3, jj = (c4 & 7) * 4;
      const float* pp = buf + row * C::PP + mt * 32 + jj;
      float4 s = *reinterpret_cast<const float4*>(pp);
#pragma unroll
      for (int w = 1; w < C::NW; ++w) {
        const float4 qq = *reinterpret_cast<const float4*>(pp + w * C::PART);
        s.x += qq.x;
        s.y += qq.y;
        s.z += qq.z;
        s.w += qq.w;
      }
      const float sc = scv[q], sh = shv[q];
      float v[4] = {fmaf(s.x, sc, sh), fmaf(s.y, sc, sh), fmaf(s.z, sc, sh), fmaf(s.w, sc, sh)};
      const size_t o = off[q];
      if (vec) {
        if (relu == 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        if (rb) {
          v[0] += rv[q].x;
          v[1] += rv[q].y;
          v[2] += rv[q].z;
          v[3] += rv[q].w;
        }
        if (relu == 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        *reinterpret_cast<float4*>(yb + o) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        const int gx = tl.x0 + 16 * (mt % C::XS) + (jj & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (gx + i >= Wo) break;
          float qv = v[i];
          if (relu == 2) qv = fmaxf(qv, 0.f);
          if (rb) qv += rb[o + i];
          if (relu == 1) qv = fmaxf(qv, 0.f);
          yb[o + i] = qv;
        }
      }
    }
    if (nbuf == 1 && item + G < items) {
      __syncthreads();   // the partial sums have been read: the buffer may take the next item's tile
      stage(tile_of(item + G), lds);
    }
  }
}

template <class C>
static int launch_sk(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y, int B,
                     int Ci, int Co, int D, int H, int W, int relu, hipStream_t st, long long in_ctot = 0, long long out_ctot = 0,
                     long long res_ctot = 0) {
  const int Do = (D - 1) / C::S + 1, Ho = (H - 1) / C::S + 1, Wo = (W - 1) / C::S + 1;
  // channels a batch item of the three tensors holds (the 2-D entry point reads / writes channel windows of wider tensors)
  const long long in_bs = (in_ctot ? in_ctot : Ci) * (long long)D * H * W, out_bs = (out_ctot ? out_ctot : Co) * (long long)Do * Ho * Wo,
                  res_bs = (res_ctot ? res_ctot : Co) * (long long)Do * Ho * Wo;
  const int ntx = cdiv(Wo, C::TXO), nty = cdiv(Ho, C::TYO), NTT = Co / 32, NTS = NTT / C::NT;
  const long long items = (long long)B * Do * nty * ntx * NTS;
  if (items > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv3d: grid too large");
  const size_t in_floats = (size_t)Ci * C::CHS, part_floats = (size_t)C::NW * C::PART;
  const size_t buf_floats = in_floats > part_floats ? in_floats : part_floats;
  // one workgroup per CU (the A fragments take half of the register file); persistent when there are more items than CUs, then
  // with a second input buffer where it fits
  long long G = num_cus() / NTS * NTS;
  if (G < NTS) G = NTS;
  if (items <= G) G = items;
  const int nbuf = (items > G && 2 * buf_floats * sizeof(float) <= 160 * 1024) ? 2 : 1;
  const size_t lds = nbuf * buf_floats * sizeof(float);
  DMB_ENSURE_LDS((&conv3d_sk_kernel<C>), (size_t)(160 * 1024));
  hipLaunchKernelGGL((conv3d_sk_kernel<C>), dim3((unsigned)G), dim3(C::NTHREADS), lds, st, x, wp, scale, shift, res, y, Ci, D, H, W,
                     Do, Ho, Wo, ntx, nty, NTT, relu, (int)items, (int)buf_floats, nbuf, in_bs, out_bs, res_bs);
  return launch_status("conv3d split-K launch failed");
}

// LDS bytes a variant needs for Ci input channels (the whole K range of a tile is resident)
template <class C>
static constexpr long long sk_lds_bytes(int Ci) {
  const long long a = (long long)Ci * C::CHS, b = (long long)C::NW * C::PART;
  return (a > b ? a : b) * 4;
}

// variant: 1 = 16 x 2 voxel tile, one 32-channel row tile per workgroup; 2 = 32 x 2 voxels, one row tile; 3 = 32 x 2 voxels, both row
// tiles of a 64-channel layer (32 input channels only: the A fragments of a wave must fit its registers).  -1: the shape does not qualify.
int conv3d_sk_try(int variant, const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                  int B, int Ci, int Co, int D, int H, int W, int stride, int relu, hipStream_t st) {
  constexpr int NW = 8;
  if (Ci % (2 * NW) != 0 || (Co != 32 && Co != 64) || W % 4 != 0 || (((uintptr_t)x) & 15) != 0) return -1;
  const int Wo = (W - 1) / stride + 1;
  if (Wo % 4 == 0 && ((((uintptr_t)y) | ((uintptr_t)res)) & 15) != 0) return -1;
  if ((long long)(Ci / NW) * D * H * W * 4 >= 0x7fffffffLL) return -1;   // 32-bit offsets inside a wave's channels
  const int NP = Ci / (2 * NW);
#define DMB_SK(S, NT, XS, YS, NPR)                                                                             \
  do {                                                                                                         \
    using C = SKCfg<S, NW, NT, XS, YS, NPR>;                                                                   \
    if (NP > NPR || sk_lds_bytes<C>(Ci) > 160 * 1024) return -1;                                               \
    return launch_sk<C>(x, wp, scale, shift, res, y, B, Ci, Co, D, H, W, relu, st);                            \
  } while (0)
  // (variant 2 and the stride-1 form of variant 3 are never picked by sk_variant: measured, kept in the development build only)
  if (stride == 1) {
    if (variant == 1) DMB_SK(1, 1, 1, 1, 4);
#ifdef DMB_DEV
    if (variant == 2) DMB_SK(1, 1, 2, 1, 4);
    if (variant == 3 && Co == 64) DMB_SK(1, 2, 2, 1, 2);
#endif
  } else if (stride == 2) {
    if (variant == 1) DMB_SK(2, 1, 1, 1, 4);
#ifdef DMB_DEV
    if (variant == 2) DMB_SK(2, 1, 2, 1, 4);
#endif
    if (variant == 3 && Co == 64) DMB_SK(2, 2, 2, 1, 2);
  }
#undef DMB_SK
  return -1;
}

// The 2-D layers (dmb_conv2d_f32, kernel 3, stride 1, dilation 1 or 2) in the same form: x / y / residual are the channel windows
// the caller has already offset, `*_ctot` the channels a batch item of each tensor holds.  relu: after the skip add (conv2d.hip).
// -1: the shape does not qualify (every row tile of a pixel tile goes to one workgroup when the A fragments fit, else one each).
int conv2d_sk_try(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y, int B, int Ci,
                  int Co, int H, int W, int dilation, int relu, int in_ctot, int out_ctot, int res_ctot, hipStream_t st) {
  constexpr int NW = 8;
  if (Ci % (2 * NW) != 0 || Co % 32 != 0 || Co > 128 || W % 4 != 0 || ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)res)) & 15) != 0) return -1;
  if ((long long)(Ci / NW) * H * W * 4 >= 0x7fffffffLL) return -1;
  const int NP = Ci / (2 * NW), NTT = Co / 32;
#define DMB_SK2(NT, NPR, DL)                                                                                        \
  do {                                                                                                              \
    using C = SKCfg<1, NW, NT, 1, 1, NPR, 1, DL>;                                                                   \
    if (NP > NPR || sk_lds_bytes<C>(Ci) > 160 * 1024) return -1;                                                    \
    return launch_sk<C>(x, wp, scale, shift, res, y, B, Ci, Co, 1, H, W, relu ? 1 : 0, st, in_ctot, out_ctot, res_ctot); \
  } while (0)
  // A fragments per wave = NP x 9 x NT registers: both row tiles of a pair up to 64 input channels, one above (both with 128 input
  // channels = 144 registers: measured, spills and 30.4 against 32.4 us -- not kept)
  if (dilation == 1) {
    if (NTT % 2 == 0 && NP <= 4) DMB_SK2(2, 4, 1);
    DMB_SK2(1, 8, 1);
  } else if (dilation == 2) {
    if (NTT % 2 == 0 && NP <= 4) DMB_SK2(2, 4, 2);
    DMB_SK2(1, 8, 2);
  }
#undef DMB_SK2
  return -1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Transposed convolution k3 s2 p1 op1 (hourglass conv5 / conv6: utils/hourglass.py:53-60,84-86), the same split-K idea.
//   y[2 i - 1 + k] += x[i] w[k] per axis: an even output 2 m sees k = 1 from input m; an odd output 2 m + 1 sees k = 2 from input m
//   and k = 0 from input m + 1.
// A workgroup owns (input-resolution tile of XS x 16 x  x  2 y positions of one z-slice, output z parity, output y parity, one
// 32-channel row tile): both x parities are accumulated side by side so that a thread of the epilogue owns four consecutive
// output columns.  Wave w multiplies its Ci / NW input channels -- the tile's (1 + pz) x (2 + py) x (16 XS + 1) haloed inputs in
// its own LDS region, its (1 + pz)(1 + py) x 3 taps of every channel pair in registers -- then the partial tiles are added in
// ascending wave order.  The four parity classes carry 3 / 6 / 6 / 12 MFMAs per channel pair and column tile.
// ---------------------------------------------------------------------------------------------------------------------------
template <int NW_, int XS_, int NPR_>
struct DSKCfg {
  static constexpr int NW = NW_, XS = XS_, NPR = NPR_, NTHREADS = 64 * NW_;
  static constexpr int TXI = 16 * XS, TYI = 2, MT = XS;
  static constexpr int P = TXI + 4;                   // staged row: columns x0 .. x0 + TXI (+ padding to 16 bytes)
  static constexpr int UPR = P / 4;
  static constexpr int PP = MT * 64 + 4;              // pitch of a partial tile's rows: 2 x 32 output columns per column tile
  static constexpr int PART = 32 * PP;
  static constexpr int IN_MAX = 2 * 3 * P;            // floats per channel of the largest class
};

// Persistent, as the convolution above: a workgroup belongs to ONE (parity class, row tile) -- so its A fragments are loaded once --
// and walks that class's tiles slot, slot + nslots, ...; the classes get workgroups in proportion to their arithmetic (4 : 2 : 2 : 1).
// (One workgroup per (tile, class) re-read the weights from L2 for every tile: 2048 workgroups x 55 KB = 113 MB for the 64 -> 32
// layer of one 256x512 pair -- 41 us for 11.5 us of MFMAs.)
struct DSKGeom {
  int Ci, D, H, W, Wout, ntx, nty, NTT, cvalid, relu, ntiles, buf_floats, nbuf;
};

template <class C, int PZ, int PY>
__device__ __forceinline__ void deconv_sk_body(float* lds, const float* __restrict__ x, const float* __restrict__ wp,
                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                               const float* __restrict__ res, float* __restrict__ y, const DSKGeom& gm, int slot,
                                               int nslots, int nt0) {
  constexpr int ZS = 1 + PZ, ROWS = 2 + PY, PLANE = ROWS * C::P, CHS = ZS * PLANE, UPC = ZS * ROWS * C::UPR;
  constexpr int NA = (1 + PZ) * (1 + PY);   // (kz, ky) pairs an output of the class sees
  const int Ci = gm.Ci, D = gm.D, H = gm.H, W = gm.W, Wout = gm.Wout, NTT = gm.NTT, cvalid = gm.cvalid, relu = gm.relu;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const unsigned HW = (unsigned)H * W, DHW = (unsigned)D * HW;
  const int CW = Ci / C::NW, c0 = wave * CW;
  struct Tile {
    int b, z0, y0, x0;
  };
  auto tile_of = [&](int t) {
    Tile tl;
    tl.x0 = (t % gm.ntx) * C::TXI;
    t /= gm.ntx;
    tl.y0 = (t % gm.nty) * C::TYI;
    t /= gm.nty;
    tl.z0 = t % D;
    tl.b = t / D;
    return tl;
  };
  auto stage = [&](const Tile& tl, float* buf) {
    float* region = buf + c0 * CHS;
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(x + ((size_t)tl.b * Ci + c0) * DHW, (unsigned)CW * DHW * 4u);
    const int NU = CW * UPC;
    for (int u0 = 0; u0 < NU; u0 += 64) {
      const int u = u0 + lane;
      const int cl = u / UPC, r0 = u - cl * UPC;
      const int zz = r0 / (ROWS * C::UPR), r1 = r0 - zz * (ROWS * C::UPR), yy = r1 / C::UPR, sg = r1 - yy * C::UPR;
      const int gz = tl.z0 + zz, gy = tl.y0 + yy, gx = tl.x0 + sg * 4;
      const bool ok = gz < D && gy < H && gx < W;
      if (u < NU)
        dma16(xrs, ok ? ((unsigned)cl * DHW + (unsigned)gz * HW + (unsigned)gy * W + (unsigned)gx) * 4u : DMA_OOB, 0u, region + u0 * 4);
    }
  };
  if (slot >= gm.ntiles) return;
  stage(tile_of(slot), lds);
  const int NP = CW / 2;
  const float* wbase = wp + ((size_t)(c0 / 2) * 27 * NTT + nt0) * 64 + lane;
  float a[C::NPR][NA][3];
#pragma unroll
  for (int p = 0; p < C::NPR; ++p)
    if (p < NP) {
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const int az = q / (1 + PY), ay = q % (1 + PY);
        const int kz = PZ ? (az ? 0 : 2) : 1, ky = PY ? (ay ? 0 : 2) : 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) a[p][q][kx] = wbase[((size_t)(p * 27 + kz * 9 + ky * 3 + kx) * NTT) * 64];
      }
    }
  // epilogue operands that do not depend on the tile
  constexpr int NE4 = 32 * C::MT * 16;
  constexpr int NEPT = (NE4 + C::NTHREADS - 1) / C::NTHREADS;
  float scv[NEPT], shv[NEPT];
#pragma unroll
  for (int q = 0; q < NEPT; ++q) {
    const int e = threadIdx.x + q * C::NTHREADS, co = nt0 * 32 + (e < NE4 ? e / (C::MT * 16) : 0);
    scv[q] = (scale && co < cvalid) ? scale[co] : 1.f;
    shv[q] = (shift && co < cvalid) ? shift[co] : 0.f;
  }
  const int Do = 2 * D, Ho = 2 * H;
  const unsigned HWo = (unsigned)Ho * Wout, DHWo = (unsigned)Do * HWo;
  const int Co = cvalid;
  int it = 0;
  for (int t = slot; t < gm.ntiles; t += nslots, ++it) {
    const Tile tl = tile_of(t);
    float* buf = lds + (gm.nbuf == 2 ? (it & 1) * gm.buf_floats : 0);
    __syncthreads();   // this tile's copies have landed (the compiler drains them here); the other buffer is free
    if (gm.nbuf == 2 && t + nslots < gm.ntiles) stage(tile_of(t + nslots), lds + ((it + 1) & 1) * gm.buf_floats);
    float* yb = y + (size_t)tl.b * Co * DHWo;
    const float* rb = res ? res + (size_t)tl.b * Co * DHWo : nullptr;
    size_t off[NEPT];
    bool live[NEPT];
    float4 rv[NEPT];
#pragma unroll
    for (int q = 0; q < NEPT; ++q) {   // this thread's words of the output tile and its skip operand (consumed after the multiply phase)
      const int e = threadIdx.x + q * C::NTHREADS;
      const int row = e / (C::MT * 16), c4 = e - row * (C::MT * 16), mt = c4 >> 4, jj = (c4 & 15) * 4;
      const int co = nt0 * 32 + row, ly = jj >> 5, ox = jj & 31;
      const int gz = 2 * tl.z0 + PZ, gy = 2 * (tl.y0 + ly) + PY, gx = 2 * (tl.x0 + 16 * mt) + ox;
      live[q] = e < NE4 && co < cvalid && tl.y0 + ly < H && gx < Wout;
      off[q] = (size_t)co * DHWo + (size_t)gz * HWo + (size_t)gy * Wout + gx;
      rv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rb && live[q]) rv[q] = *reinterpret_cast<const float4*>(rb + off[q]);
    }
    f32x16 acc[2][C::MT];   // [x parity][column tile]
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[px][mt][r] = 0.f;
    const float* bb = buf + c0 * CHS + h * CHS + (j >> 4) * C::P + (j & 15);
#pragma unroll
    for (int p = 0; p < C::NPR; ++p)
      if (p < NP) {
        const float* bp = bb + 2 * p * CHS;
#pragma unroll
        for (int q = 0; q < NA; ++q) {
          const int az = q / (1 + PY), ay = q % (1 + PY);
          float b0[C::MT], b1[C::MT];
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) {
            b0[mt] = bp[az * PLANE + ay * C::P + 16 * mt];
            b1[mt] = bp[az * PLANE + ay * C::P + 16 * mt + 1];
          }
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) {
            acc[0][mt] = DMB_MFMA(a[p][q][1], b0[mt], acc[0][mt]);   // even x: kx = 1 from input m
            acc[1][mt] = DMB_MFMA(a[p][q][2], b0[mt], acc[1][mt]);   // odd x: kx = 2 from input m ...
            acc[1][mt] = DMB_MFMA(a[p][q][0], b1[mt], acc[1][mt]);   // ... and kx = 0 from input m + 1
          }
        }
      }
    __syncthreads();
    {
      float* part = buf + wave * C::PART;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          *reinterpret_cast<float2*>(part + cd_row(r, h) * C::PP + mt * 64 + (j >> 4) * 32 + 2 * (j & 15)) = make_float2(acc[0][mt][r], acc[1][mt][r]);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NEPT; ++q) {
      if (!live[q]) continue;
      const int e = threadIdx.x + q * C::NTHREADS;
      const int row = e / (C::MT * 16), c4 = e - row * (C::MT * 16), mt = c4 >> 4, jj = (c4 & 15) * 4;   // jj: column of the 64-wide strip
      const float* pp = buf + row * C::PP + mt * 64 + jj;
      float4 s = *reinterpret_cast<const float4*>(pp);
#pragma unroll
      for (int w = 1; w < C::NW; ++w) {
        const float4 qq = *reinterpret_cast<const float4*>(pp + w * C::PART);
        s.x += qq.x;
        s.y += qq.y;
        s.z += qq.z;
        s.w += qq.w;
      }
      const float sc = scv[q], sh = shv[q];
      float v[4] = {fmaf(s.x, sc, sh), fmaf(s.y, sc, sh), fmaf(s.z, sc, sh), fmaf(s.w, sc, sh)};
      if (relu == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
      }
      if (rb) {
        v[0] += rv[q].x;
        v[1] += rv[q].y;
        v[2] += rv[q].z;
        v[3] += rv[q].w;
      }
      if (relu == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
      }
      *reinterpret_cast<float4*>(yb + off[q]) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (gm.nbuf == 1 && t + nslots < gm.ntiles) {
      __syncthreads();   // the partial sums have been read: the buffer may take the next tile
      stage(tile_of(t + nslots), lds);
    }
  }
}

// Workgroups [0, g1) belong to class (pz, py) = (1, 1), [g1, g2) to (1, 0), [g2, g3) to (0, 1), the rest to (0, 0); inside a class
// workgroup l has row tile l % NTT and walks the tiles l / NTT, l / NTT + slots, ...
template <class C>
__global__ __launch_bounds__(C::NTHREADS) void deconv3d_sk_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const float* __restrict__ res, float* __restrict__ y, const DSKGeom gm,
                                                                  int g1, int g2, int g3) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  const int NTT = gm.NTT;
  if (wg < g1)
    deconv_sk_body<C, 1, 1>(lds, x, wp, scale, shift, res, y, gm, wg / NTT, g1 / NTT, wg % NTT);
  else if (wg < g2)
    deconv_sk_body<C, 1, 0>(lds, x, wp, scale, shift, res, y, gm, (wg - g1) / NTT, (g2 - g1) / NTT, (wg - g1) % NTT);
  else if (wg < g3)
    deconv_sk_body<C, 0, 1>(lds, x, wp, scale, shift, res, y, gm, (wg - g2) / NTT, (g3 - g2) / NTT, (wg - g2) % NTT);
  else
    deconv_sk_body<C, 0, 0>(lds, x, wp, scale, shift, res, y, gm, (wg - g3) / NTT, (G - g3) / NTT, (wg - g3) % NTT);
}

template <class C>
static int launch_dsk(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y, int B,
                      int Ci, int Co, int D, int H, int W, int Wout, int relu, hipStream_t st) {
  const int ntx = cdiv(W, C::TXI), nty = cdiv(H, C::TYI), NTT = cdiv(Co, 32);
  const long long ntiles = (long long)B * D * nty * ntx;
  if (ntiles * 4 * NTT > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "deconv3d: grid too large");
  const size_t in_floats = (size_t)Ci * C::IN_MAX, part_floats = (size_t)C::NW * C::PART;
  const size_t buf_floats = in_floats > part_floats ? in_floats : part_floats;
  if (buf_floats * sizeof(float) > 160 * 1024) return -1;
  // with at most two single-buffered workgroups per CU's worth of items every workgroup takes ONE tile; beyond that the walk is
  // persistent on the chip's workgroup slots: two per CU where two double-buffered workgroups fit the LDS, else one
  const int two_bufs = 2 * buf_floats * sizeof(float) <= 160 * 1024;
  const int per_cu = (two_bufs && 4 * buf_floats * sizeof(float) <= 160 * 1024) ? 2 : 1;
  const long long slots = (long long)per_cu * num_cus();
  long long gc[4];
  int nbuf = 1;
  if (ntiles * 4 * NTT <= (long long)(two_bufs ? 2 : 1) * num_cus()) {
    for (int c = 0; c < 4; ++c) gc[c] = ntiles * NTT;    // one tile per workgroup
  } else {
    static const int weight[4] = {4, 2, 2, 1};           // (kz, ky) pairs of the classes (1, 1), (1, 0), (0, 1), (0, 0)
    for (int c = 0; c < 4; ++c) {
      long long g = slots * weight[c] / 9 / NTT * NTT;
      if (g < NTT) g = NTT;
      if (g > ntiles * NTT) g = ntiles * NTT;
      gc[c] = g;
    }
    nbuf = two_bufs ? 2 : 1;
  }
  DSKGeom gm = {Ci, D, H, W, Wout, ntx, nty, NTT, Co, relu, (int)ntiles, (int)buf_floats, nbuf};
  const size_t lds = nbuf * buf_floats * sizeof(float);
  DMB_ENSURE_LDS((&deconv3d_sk_kernel<C>), (size_t)(160 * 1024));
  hipLaunchKernelGGL((deconv3d_sk_kernel<C>), dim3((unsigned)(gc[0] + gc[1] + gc[2] + gc[3])), dim3(C::NTHREADS), lds, st, x, wp, scale, shift,
                     res, y, gm, (int)gc[0], (int)(gc[0] + gc[1]), (int)(gc[0] + gc[1] + gc[2]));
  return launch_status("deconv3d split-K launch failed");
}

// variant: 1 = 16 x 2 input positions per workgroup, eight waves; 2 = 32 x 2, eight waves; 3 = 16 x 2, four waves; 4 = 32 x 2, four waves.
// -1: the shape does not qualify.
int deconv3d_sk_try(int variant, const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                    int B, int Ci, int Co, int D, int H, int W, int Wout, int relu, hipStream_t st) {
  if (Co > 64 || W % 4 != 0 || Wout % 4 != 0 || ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)res)) & 15) != 0) return -1;
  const int nw = variant <= 2 ? 8 : 4;
  if (Ci % (2 * nw) != 0 || Ci / (2 * nw) > 4 * (variant <= 2 ? 1 : 2)) return -1;
  if ((long long)(Ci / nw) * D * H * W * 4 >= 0x7fffffffLL) return -1;
  switch (variant) {   // (dsk_variant picks 3 only: the others lost every measured shape and exist in the development build)
    case 3: return launch_dsk<DSKCfg<4, 1, 8>>(x, wp, scale, shift, res, y, B, Ci, Co, D, H, W, Wout, relu, st);
#ifdef DMB_DEV
    case 1: return launch_dsk<DSKCfg<8, 1, 4>>(x, wp, scale, shift, res, y, B, Ci, Co, D, H, W, Wout, relu, st);
    case 2: return launch_dsk<DSKCfg<8, 2, 4>>(x, wp, scale, shift, res, y, B, Ci, Co, D, H, W, Wout, relu, st);
    case 4: return launch_dsk<DSKCfg<4, 2, 8>>(x, wp, scale, shift, res, y, B, Ci, Co, D, H, W, Wout, relu, st);
#endif
  }
  return -1;
}

}  // namespace dmb
