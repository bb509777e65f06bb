// 3-D convolution family of the cost aggregators as FP32 implicit GEMMs on the gfx950 matrix cores.
//
//   y[co, p] = sum_{ci, tap} w[co, ci, tap] * x[ci, p + tap]        (M = Cout, N = voxels, K = Cin * 27)
//
// v_mfma_f32_32x32x2_f32 is exact FP32 (bitwise an fmaf chain in k order) and runs at the FP32 vector peak
// (157.3 TFLOP/s), reached from one wave per SIMD -- which a VALU kernel cannot do.  Roles: A = weights
// (row = output channel), B = input voxels (column = voxel), so that every accumulator register holds one
// output channel x 32 consecutive voxels along W and the epilogue stores 128-byte runs into NCDHW.
//
// Data flow per workgroup (256 threads = 4 waves, 2 workgroups per CU so one stages while the other
// computes):  for each chunk of CK input channels: stage the haloed input tile [CK][TZ+2][TY+2][TX+2]
// into LDS (zero padding materialised there) -> every wave runs (CK/2)*27 k-steps; per k-step one
// coalesced 256-B weight-fragment load (L2-resident, prepacked) and MT ds_read_b32 B-fragments feed
// MT*NT MFMAs.  The "voxel" index of a B fragment walks the FLATTENED padded (y, x) plane of the LDS tile,
// so a tap (dz, dy, dx) is a compile-time address offset and 32 lanes always read 32 consecutive floats
// (conflict-free); the 2 halo columns per row are computed and discarded (2/(TX+2) waste).
//
// Reference semantics: dmb/modeling/stereo/layers/basic_layers.py:68-100,160-177 (Conv3d/ConvTranspose3d
// + BatchNorm3d + ReLU factories), cost_processors/utils/hourglass.py:62-86.
#include "dmb_common.h"

namespace dmb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DMB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Row of the 32x32 C/D tile held by accumulator register r of lane-half h (cdna_hip_programming.md s3).
__device__ __forceinline__ int cd_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---------------------------------------------------------------------------------------------------------
// Weight prepack: A fragments in k-step order.
//   wp[((kp * 27 + tap) * NTT + nt) * 64 + lane] = W(co = nt*32 + (lane & 31), ci = 2*kp + (lane >> 5), tap)
// For nn.Conv3d W(co, ci, tap) = w[co][ci][tap]; for nn.ConvTranspose3d W(co, ci, tap) = w[ci][co][tap].
// ---------------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int transposed) {
  const int NTT = Co / 32;
  const long long total = (long long)Ci * 27 * Co;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    long long r = i >> 6;
    const int nt = (int)(r % NTT);
    r /= NTT;
    const int tap = (int)(r % 27);
    const int kp = (int)(r / 27);
    const int co = nt * 32 + (lane & 31);
    const int ci = 2 * kp + (lane >> 5);
    wp[i] = transposed ? w[((size_t)ci * Co + co) * 27 + tap] : w[((size_t)co * Ci + ci) * 27 + tap];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Shared epilogue: v = acc*scale + shift (+ residual) (relu), scattered into NCDHW.
// ---------------------------------------------------------------------------------------------------------
struct Affine {
  const float* scale;
  const float* shift;
};

// ---------------------------------------------------------------------------------------------------------
// Stride-1 kernel.
// ---------------------------------------------------------------------------------------------------------
template <int CIN_, int COUT_, int TY_, int TX_, int CK_, int WN_>
struct S1Cfg {
  static constexpr int CIN = CIN_, COUT = COUT_, TY = TY_, TX = TX_, CK = CK_, WN = WN_;
  static constexpr int WZ = 4 / WN;        // waves along z; one output z-slice per wave
  static constexpr int TZ = WZ;
  static constexpr int P = TX + 2;         // padded row pitch
  static constexpr int ROWS = TY + 2;
  static constexpr int PLANE = ROWS * P;
  static constexpr int ZS = TZ + 2;
  static constexpr int MT = (TY * P + 31) / 32;  // 32-voxel column tiles per wave
  static constexpr int NTT = COUT / 32;          // 32-channel row tiles in total
  static constexpr int NT = NTT / WN;            // ... per wave
  static constexpr int CH_STRIDE = ZS * PLANE + 36;  // + slack read by discarded columns
  static constexpr int LDS_FLOATS = CK * CH_STRIDE;
  static_assert(P <= 64, "one wave stages one tile row per instruction");
  static_assert(CIN % CK == 0 && CK % 2 == 0 && COUT % (32 * WN) == 0, "shape");
};

template <class C>
__global__ __launch_bounds__(256, 2) void conv3d_s1_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ res, float* __restrict__ y, int D,
                                                           int H, int W, int ntx, int nty, int ntz, int relu) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % ntx;
  t /= ntx;
  const int ty = t % nty;
  t /= nty;
  const int tz = t % ntz;
  const int b = t / ntz;
  const int x0 = tx * C::TX, y0 = ty * C::TY, z0 = tz * C::TZ;

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int wz = wave / C::WN, wn = wave % C::WN;
  const size_t HW = (size_t)H * W;
  const float* xb = x + (size_t)b * C::CIN * D * HW;

  f32x16 acc[C::MT][C::NT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const float* bbase = lds + h * C::CH_STRIDE + wz * C::PLANE + j;

  for (int c0 = 0; c0 < C::CIN; c0 += C::CK) {
    __syncthreads();
    // ---- stage [CK][ZS][ROWS][P] ----
    constexpr int NR = C::CK * C::ZS * C::ROWS;
    for (int r = wave; r < NR; r += 4) {
      const int cl = r / (C::ZS * C::ROWS);
      const int rem = r % (C::ZS * C::ROWS);
      const int zz = rem / C::ROWS, yy = rem % C::ROWS;
      const int gz = z0 - 1 + zz, gy = y0 - 1 + yy, gx = x0 - 1 + lane;
      if (lane < C::P) {
        float v = 0.f;
        if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
          v = xb[((size_t)(c0 + cl) * D + gz) * HW + (size_t)gy * W + gx];
        lds[cl * C::CH_STRIDE + zz * C::PLANE + yy * C::P + lane] = v;
      }
    }
    __syncthreads();
    // ---- (CK/2) * 27 k-steps ----
    const float* wpc = wp + ((size_t)(c0 / 2) * 27 * C::NTT + wn * C::NT) * 64 + lane;
#pragma unroll
    for (int cp = 0; cp < C::CK / 2; ++cp) {
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        float a[C::NT];
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) a[nt] = wpc[((size_t)(cp * 27 + tap) * C::NTT + nt) * 64];
        const float* bp = bbase + 2 * cp * C::CH_STRIDE + dz * C::PLANE + dy * C::P + dx;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
          const float bv = bp[mt * 32];
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt) acc[mt][nt] = DMB_MFMA(a[nt], bv, acc[mt][nt]);
        }
      }
    }
  }

  // ---- epilogue ----
  const int gz = z0 + wz;
  if (gz >= D) return;
  float* yb = y + (size_t)b * C::COUT * D * HW;
  const float* rb = res ? res + (size_t)b * C::COUT * D * HW : nullptr;
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt) {
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (wn * C::NT + nt) * 32 + cd_row(r, h);
      sc[r] = scale ? scale[co] : 1.f;
      sh[r] = shift ? shift[co] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
      const int m = mt * 32 + j;
      const int ly = m / C::P, lx = m - ly * C::P;
      const int gy = y0 + ly, gx = x0 + lx;
      if (m < C::TY * C::P && lx < C::TX && gy < H && gx < W) {
        const size_t o = (size_t)gz * HW + (size_t)gy * W + gx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = (wn * C::NT + nt) * 32 + cd_row(r, h);
          float v = fmaf(acc[mt][nt][r], sc[r], sh[r]);
          if (rb) v += rb[(size_t)co * D * HW + o];
          if (relu) v = fmaxf(v, 0.f);
          yb[(size_t)co * D * HW + o] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Stride-2 kernel (k3, pad 1): input voxel = 2*out - 1 + tap.  LDS tile rows are split by y parity so that
// a flattened output index m = ly*(TX+1) + lx maps to LDS offset 2*m + const(tap); the x stride of 2 floats
// is a harmless 2-way bank conflict (LDS has > 4x headroom next to a 64-cycle MFMA).
// ---------------------------------------------------------------------------------------------------------
template <int CIN_, int COUT_, int TY_, int TX_, int CK_, int WN_>
struct S2Cfg {
  static constexpr int CIN = CIN_, COUT = COUT_, TY = TY_, TX = TX_, CK = CK_, WN = WN_;
  static constexpr int WZ = 4 / WN;
  static constexpr int TZ = WZ;
  static constexpr int PO = TX + 1;            // output-position pitch
  static constexpr int R = 2 * PO;             // LDS row pitch (input columns 2*x0-1 .. 2*x0+2*TX-1, padded)
  static constexpr int PYPL = (TY + 1) * R;    // one y-parity plane
  static constexpr int ZPL = 2 * PYPL;         // one input z-slice
  static constexpr int ZS = 2 * TZ + 1;
  static constexpr int INROWS = 2 * TY + 1;
  static constexpr int INCOLS = 2 * TX + 1;
  static constexpr int MT = (TY * PO + 31) / 32;
  static constexpr int NTT = COUT / 32;
  static constexpr int NT = NTT / WN;
  static constexpr int CH_STRIDE = ZS * ZPL + 72;
  static constexpr int LDS_FLOATS = CK * CH_STRIDE;
  static_assert(CIN % CK == 0 && CK % 2 == 0 && COUT % (32 * WN) == 0, "shape");
};

template <class C>
__global__ __launch_bounds__(256, 2) void conv3d_s2_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ res, float* __restrict__ y, int D,
                                                           int H, int W, int Do, int Ho, int Wo, int ntx, int nty,
                                                           int ntz, int relu) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % ntx;
  t /= ntx;
  const int ty = t % nty;
  t /= nty;
  const int tz = t % ntz;
  const int b = t / ntz;
  const int x0 = tx * C::TX, y0 = ty * C::TY, z0 = tz * C::TZ;  // output coordinates

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int wz = wave / C::WN, wn = wave % C::WN;
  const size_t HW = (size_t)H * W;
  const float* xb = x + (size_t)b * C::CIN * D * HW;

  f32x16 acc[C::MT][C::NT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const float* bbase = lds + h * C::CH_STRIDE + (2 * wz) * C::ZPL + 2 * j;

  for (int c0 = 0; c0 < C::CIN; c0 += C::CK) {
    __syncthreads();
    constexpr int NR = C::CK * C::ZS * C::INROWS;
    for (int r = wave; r < NR; r += 4) {
      const int cl = r / (C::ZS * C::INROWS);
      const int rem = r % (C::ZS * C::INROWS);
      const int zz = rem / C::INROWS, ry = rem % C::INROWS;
      const int gz = 2 * z0 - 1 + zz, gy = 2 * y0 - 1 + ry;
      const bool rowok = gz >= 0 && gz < D && gy >= 0 && gy < H;
      float* dst = lds + cl * C::CH_STRIDE + zz * C::ZPL + (ry & 1) * C::PYPL + (ry >> 1) * C::R;
      const float* src = xb + ((size_t)(c0 + cl) * D + gz) * HW + (size_t)gy * W;
      for (int col = lane; col < C::INCOLS; col += 64) {
        const int gx = 2 * x0 - 1 + col;
        dst[col] = (rowok && gx >= 0 && gx < W) ? src[gx] : 0.f;
      }
    }
    __syncthreads();
    const float* wpc = wp + ((size_t)(c0 / 2) * 27 * C::NTT + wn * C::NT) * 64 + lane;
#pragma unroll
    for (int cp = 0; cp < C::CK / 2; ++cp) {
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        float a[C::NT];
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) a[nt] = wpc[((size_t)(cp * 27 + tap) * C::NTT + nt) * 64];
        const float* bp = bbase + 2 * cp * C::CH_STRIDE + dz * C::ZPL + (dy & 1) * C::PYPL + (dy >> 1) * C::R + dx;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
          const float bv = bp[mt * 64];
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt) acc[mt][nt] = DMB_MFMA(a[nt], bv, acc[mt][nt]);
        }
      }
    }
  }

  const int gz = z0 + wz;
  if (gz >= Do) return;
  const size_t HWo = (size_t)Ho * Wo;
  float* yb = y + (size_t)b * C::COUT * Do * HWo;
  const float* rb = res ? res + (size_t)b * C::COUT * Do * HWo : nullptr;
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt) {
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (wn * C::NT + nt) * 32 + cd_row(r, h);
      sc[r] = scale ? scale[co] : 1.f;
      sh[r] = shift ? shift[co] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
      const int m = mt * 32 + j;
      const int ly = m / C::PO, lx = m - ly * C::PO;
      const int gy = y0 + ly, gx = x0 + lx;
      if (m < C::TY * C::PO && lx < C::TX && gy < Ho && gx < Wo) {
        const size_t o = (size_t)gz * HWo + (size_t)gy * Wo + gx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = (wn * C::NT + nt) * 32 + cd_row(r, h);
          float v = fmaf(acc[mt][nt][r], sc[r], sh[r]);
          if (rb) v += rb[(size_t)co * Do * HWo + o];
          if (relu) v = fmaxf(v, 0.f);
          yb[(size_t)co * Do * HWo + o] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Transposed convolution k3 s2 p1 op1:  y[2i - 1 + k] += x[i] * w[k]  per axis.  An output of parity 0 (even)
// sees k=1 from i=o/2; parity 1 (odd) sees k=2 from i=(o-1)/2 and k=0 from i=(o+1)/2.  One workgroup =
// (input-resolution tile, z parity, y parity); both x parities are accumulated by the same wave so that a
// lane owns two adjacent outputs and stores them as one 8-byte word.  No zero-insertion, no wasted MACs.
// ---------------------------------------------------------------------------------------------------------
template <int CIN_, int COUT_, int TY_, int TX_, int CK_, int WN_>
struct DCfg {
  static constexpr int CIN = CIN_, COUT = COUT_, TY = TY_, TX = TX_, CK = CK_, WN = WN_;
  static constexpr int WZ = 4 / WN;
  static constexpr int TZ = WZ;
  static constexpr int P = TX + 1;
  static constexpr int ROWS = TY + 1;
  static constexpr int PLANE = ROWS * P;
  static constexpr int ZS = TZ + 1;
  static constexpr int MT = (TY * P + 31) / 32;
  static constexpr int NTT = COUT / 32;
  static constexpr int NT = NTT / WN;
  static constexpr int CH_STRIDE = ZS * PLANE + 36;
  static constexpr int LDS_FLOATS = CK * CH_STRIDE;
  static_assert(P <= 64, "one wave stages one tile row per instruction");
  static_assert(CIN % CK == 0 && CK % 2 == 0 && COUT % (32 * WN) == 0, "shape");
};

template <class C>
__global__ __launch_bounds__(256, 2) void deconv3d_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ res, float* __restrict__ y, int D,
                                                          int H, int W, int ntx, int nty, int ntz, int relu) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int py = t & 1, pz = (t >> 1) & 1;
  t >>= 2;
  const int tx = t % ntx;
  t /= ntx;
  const int ty = t % nty;
  t /= nty;
  const int tz = t % ntz;
  const int b = t / ntz;
  const int x0 = tx * C::TX, y0 = ty * C::TY, z0 = tz * C::TZ;  // input coordinates

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int wz = wave / C::WN, wn = wave % C::WN;
  const size_t HW = (size_t)H * W;
  const float* xb = x + (size_t)b * C::CIN * D * HW;

  f32x16 acc[2][C::MT][C::NT];
#pragma unroll
  for (int px = 0; px < 2; ++px)
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[px][mt][nt][r] = 0.f;

  const float* bbase = lds + h * C::CH_STRIDE + wz * C::PLANE + j;

  for (int c0 = 0; c0 < C::CIN; c0 += C::CK) {
    __syncthreads();
    constexpr int NR = C::CK * C::ZS * C::ROWS;
    for (int r = wave; r < NR; r += 4) {
      const int cl = r / (C::ZS * C::ROWS);
      const int rem = r % (C::ZS * C::ROWS);
      const int zz = rem / C::ROWS, yy = rem % C::ROWS;
      const int gz = z0 + zz, gy = y0 + yy, gx = x0 + lane;
      if (lane < C::P) {
        float v = 0.f;
        if (gz < D && gy < H && gx < W) v = xb[((size_t)(c0 + cl) * D + gz) * HW + (size_t)gy * W + gx];
        lds[cl * C::CH_STRIDE + zz * C::PLANE + yy * C::P + lane] = v;
      }
    }
    __syncthreads();
    const float* wpc = wp + ((size_t)(c0 / 2) * 27 * C::NTT + wn * C::NT) * 64 + lane;
#pragma unroll 1
    for (int cp = 0; cp < C::CK / 2; ++cp) {
#pragma unroll 1
      for (int az = 0; az <= pz; ++az) {
        const int kz = pz ? (az ? 0 : 2) : 1;  // az = input z offset
#pragma unroll 1
        for (int ay = 0; ay <= py; ++ay) {
          const int ky = py ? (ay ? 0 : 2) : 1;
          const float* wk = wpc + ((size_t)(cp * 27 + kz * 9 + ky * 3) * C::NTT) * 64;
          const float* bp = bbase + 2 * cp * C::CH_STRIDE + az * C::PLANE + ay * C::P;
          float a0[C::NT], a1[C::NT], a2[C::NT];
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt) {
            a0[nt] = wk[((size_t)0 * C::NTT + nt) * 64];
            a1[nt] = wk[((size_t)1 * C::NTT + nt) * 64];
            a2[nt] = wk[((size_t)2 * C::NTT + nt) * 64];
          }
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) {
            const float b0 = bp[mt * 32], b1 = bp[mt * 32 + 1];
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt) {
              acc[0][mt][nt] = DMB_MFMA(a1[nt], b0, acc[0][mt][nt]);  // even x: k=1, i=q
              acc[1][mt][nt] = DMB_MFMA(a2[nt], b0, acc[1][mt][nt]);  // odd x:  k=2, i=q
              acc[1][mt][nt] = DMB_MFMA(a0[nt], b1, acc[1][mt][nt]);  //         k=0, i=q+1
            }
          }
        }
      }
    }
  }

  const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const int gzi = z0 + wz;
  if (gzi >= D) return;
  const int gz = 2 * gzi + pz;
  const size_t HWo = (size_t)Ho * Wo;
  float* yb = y + (size_t)b * C::COUT * Do * HWo;
  const float* rb = res ? res + (size_t)b * C::COUT * Do * HWo : nullptr;
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt) {
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (wn * C::NT + nt) * 32 + cd_row(r, h);
      sc[r] = scale ? scale[co] : 1.f;
      sh[r] = shift ? shift[co] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
      const int m = mt * 32 + j;
      const int ly = m / C::P, lx = m - ly * C::P;
      const int gyi = y0 + ly, gxi = x0 + lx;
      if (m < C::TY * C::P && lx < C::TX && gyi < H && gxi < W) {
        const size_t o = (size_t)gz * HWo + (size_t)(2 * gyi + py) * Wo + 2 * gxi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = (wn * C::NT + nt) * 32 + cd_row(r, h);
          float v0 = fmaf(acc[0][mt][nt][r], sc[r], sh[r]);
          float v1 = fmaf(acc[1][mt][nt][r], sc[r], sh[r]);
          if (rb) {
            const float2 rv = *reinterpret_cast<const float2*>(rb + (size_t)co * Do * HWo + o);
            v0 += rv.x;
            v1 += rv.y;
          }
          if (relu) {
            v0 = fmaxf(v0, 0.f);
            v1 = fmaxf(v1, 0.f);
          }
          *reinterpret_cast<float2*>(yb + (size_t)co * Do * HWo + o) = make_float2(v0, v1);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Single-output-channel 3x3x3 convolution (classifier heads).  N = 1 wastes 31/32 of an MFMA tile, and the
// layer is HBM-bound anyway (reads Ci planes, writes one), so this one is a VALU kernel: weights are wave
// uniform (scalar loads -> SGPR operands of v_fmac_f32), each thread produces 4 consecutive x from 6-float
// LDS rows.  acc order: ci ascending, then (kd, kh, kw) ascending -- the same FP32 fma chain as above.
// ---------------------------------------------------------------------------------------------------------
constexpr int C1_TX = 64, C1_TY = 4, C1_TZ = 4, C1_CK = 4;
constexpr int C1_P = C1_TX + 4;  // 66 used, even pitch keeps 8-byte alignment
constexpr int C1_PLANE = (C1_TY + 2) * C1_P;
constexpr int C1_CH = (C1_TZ + 2) * C1_PLANE;

__global__ __launch_bounds__(256) void conv3d_c1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float bias, const float* __restrict__ res,
                                                        float* __restrict__ y, int Ci, int D, int H, int W, int ntx,
                                                        int nty, int ntz) {
  __shared__ float lds[C1_CK * C1_CH];
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % ntx;
  t /= ntx;
  const int ty = t % nty;
  t /= nty;
  const int tz = t % ntz;
  const int b = t / ntz;
  const int x0 = tx * C1_TX, y0 = ty * C1_TY, z0 = tz * C1_TZ;
  const size_t HW = (size_t)H * W;
  const float* xb = x + (size_t)b * Ci * D * HW;
  const int lxq = threadIdx.x & 15;         // 16 threads x 4 outputs along x
  const int lyz = threadIdx.x >> 4;         // 16 (y, z) rows
  const int ly = lyz & 3, lz = lyz >> 2;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

  for (int c0 = 0; c0 < Ci; c0 += C1_CK) {
    __syncthreads();
    constexpr int NR = C1_CK * (C1_TZ + 2) * (C1_TY + 2);
    for (int r = wave; r < NR; r += 4) {
      const int cl = r / ((C1_TZ + 2) * (C1_TY + 2));
      const int rem = r % ((C1_TZ + 2) * (C1_TY + 2));
      const int zz = rem / (C1_TY + 2), yy = rem % (C1_TY + 2);
      const int gz = z0 - 1 + zz, gy = y0 - 1 + yy;
      const bool rowok = (c0 + cl) < Ci && gz >= 0 && gz < D && gy >= 0 && gy < H;
      for (int col = lane; col < C1_TX + 2; col += 64) {
        const int gx = x0 - 1 + col;
        lds[cl * C1_CH + zz * C1_PLANE + yy * C1_P + col] =
            (rowok && gx >= 0 && gx < W) ? xb[((size_t)(c0 + cl) * D + gz) * HW + (size_t)gy * W + gx] : 0.f;
      }
    }
    __syncthreads();
    const int nc = (Ci - c0) < C1_CK ? (Ci - c0) : C1_CK;
    for (int cl = 0; cl < nc; ++cl) {
      const float* wc = w + (size_t)(c0 + cl) * 27;
#pragma unroll
      for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const float* row = lds + cl * C1_CH + (lz + dz) * C1_PLANE + (ly + dy) * C1_P + lxq * 4;
          const float2 q0 = *reinterpret_cast<const float2*>(row);
          const float2 q1 = *reinterpret_cast<const float2*>(row + 2);
          const float2 q2 = *reinterpret_cast<const float2*>(row + 4);
          const float v[6] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y};
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float wv = wc[dz * 9 + dy * 3 + dx];
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o] = fmaf(v[o + dx], wv, acc[o]);
          }
        }
    }
  }
  const int gz = z0 + lz, gy = y0 + ly, gx = x0 + lxq * 4;
  if (gz < D && gy < H) {
    const size_t o = (size_t)b * D * HW + (size_t)gz * HW + (size_t)gy * W + gx;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (gx + i < W) {
        float v = acc[i] + bias;
        if (res) v += res[o + i];
        y[o + i] = v;
      }
  }
}

// The accumulation order above is (dz, dy) outer, dx inner PER channel, i.e. tap-ascending within a channel and
// channels ascending -- identical to the MFMA kernels' k order up to their channel pairing.

template <class C>
static int launch_s1(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                     float* y, int B, int D, int H, int W, int relu, hipStream_t st) {
  const int ntx = cdiv(W, C::TX), nty = cdiv(H, C::TY), ntz = cdiv(D, C::TZ);
  const long long nblk = (long long)B * ntx * nty * ntz;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv3d: grid too large");
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_s1_kernel<C>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3d_s1_kernel<C>), dim3((unsigned)nblk), dim3(256), lds, st, x, wp, scale, shift, res, y, D, H,
                     W, ntx, nty, ntz, relu);
  return launch_status("conv3d stride-1 launch failed");
}

template <class C>
static int launch_s2(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                     float* y, int B, int D, int H, int W, int relu, hipStream_t st) {
  const int Do = (D - 1) / 2 + 1, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int ntx = cdiv(Wo, C::TX), nty = cdiv(Ho, C::TY), ntz = cdiv(Do, C::TZ);
  const long long nblk = (long long)B * ntx * nty * ntz;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv3d: grid too large");
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_s2_kernel<C>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3d_s2_kernel<C>), dim3((unsigned)nblk), dim3(256), lds, st, x, wp, scale, shift, res, y, D, H,
                     W, Do, Ho, Wo, ntx, nty, ntz, relu);
  return launch_status("conv3d stride-2 launch failed");
}

template <class C>
static int launch_deconv(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                         float* y, int B, int D, int H, int W, int relu, hipStream_t st) {
  const int ntx = cdiv(W, C::TX), nty = cdiv(H, C::TY), ntz = cdiv(D, C::TZ);
  const long long nblk = 4LL * B * ntx * nty * ntz;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "deconv3d: grid too large");
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&deconv3d_kernel<C>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((deconv3d_kernel<C>), dim3((unsigned)nblk), dim3(256), lds, st, x, wp, scale, shift, res, y, D, H,
                     W, ntx, nty, ntz, relu);
  return launch_status("deconv3d launch failed");
}

}  // namespace dmb

using namespace dmb;

extern "C" long long dmb_conv3d_packed_floats(int Co, int Ci) { return (long long)Co * Ci * 27; }
extern "C" long long dmb_deconv3d_packed_floats(int Ci, int Co) { return (long long)Co * Ci * 27; }

static int pack_common(const float* w, float* wp, int Co, int Ci, int transposed, void* stream) {
  if (!w || !wp || Co <= 0 || Ci <= 0 || Co % 32 != 0 || Ci % 2 != 0)
    return fail(DMB_EINVAL, "pack_weights: Co must be a multiple of 32 and Ci even");
  const long long total = (long long)Co * Ci * 27;
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp, Co, Ci, transposed);
  return launch_status("pack_weights launch failed");
}

extern "C" int dmb_conv3d_pack_weights_f32(const float* w, float* wpack, int Co, int Ci, void* stream) {
  return pack_common(w, wpack, Co, Ci, 0, stream);
}
extern "C" int dmb_deconv3d_pack_weights_f32(const float* w, float* wpack, int Ci, int Co, void* stream) {
  return pack_common(w, wpack, Co, Ci, 1, stream);
}

extern "C" int dmb_conv3d_k3_f32(const float* x, const float* wpack, const float* scale, const float* shift,
                                 const float* residual, float* y, int B, int Ci, int Co, int D, int H, int W,
                                 int stride, int relu, void* stream) {
  if (!x || !wpack || !y || B <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "conv3d: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1) {
    if (Ci == 32 && Co == 32) return launch_s1<S1Cfg<32, 32, 4, 60, 8, 1>>(x, wpack, scale, shift, residual, y, B, D, H, W, relu, st);
    if (Ci == 64 && Co == 32) return launch_s1<S1Cfg<64, 32, 4, 60, 8, 1>>(x, wpack, scale, shift, residual, y, B, D, H, W, relu, st);
    if (Ci == 64 && Co == 64) return launch_s1<S1Cfg<64, 64, 4, 60, 8, 2>>(x, wpack, scale, shift, residual, y, B, D, H, W, relu, st);
    if (Ci == 32 && Co == 64) return launch_s1<S1Cfg<32, 64, 4, 60, 8, 2>>(x, wpack, scale, shift, residual, y, B, D, H, W, relu, st);
  } else if (stride == 2) {
    if (Ci == 32 && Co == 64) return launch_s2<S2Cfg<32, 64, 4, 60, 2, 2>>(x, wpack, scale, shift, residual, y, B, D, H, W, relu, st);
    if (Ci == 64 && Co == 64) return launch_s2<S2Cfg<64, 64, 4, 60, 2, 2>>(x, wpack, scale, shift, residual, y, B, D, H, W, relu, st);
  }
  return fail(DMB_EUNSUPPORTED, "conv3d: (Ci, Co, stride) not instantiated");
}

extern "C" int dmb_deconv3d_k3s2_f32(const float* x, const float* wpack, const float* scale, const float* shift,
                                     const float* residual, float* y, int B, int Ci, int Co, int D, int H, int W,
                                     int relu, void* stream) {
  if (!x || !wpack || !y || B <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "deconv3d: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (Ci == 64 && Co == 64) return launch_deconv<DCfg<64, 64, 2, 60, 16, 2>>(x, wpack, scale, shift, residual, y, B, D, H, W, relu, st);
  if (Ci == 64 && Co == 32) return launch_deconv<DCfg<64, 32, 2, 60, 16, 1>>(x, wpack, scale, shift, residual, y, B, D, H, W, relu, st);
  return fail(DMB_EUNSUPPORTED, "deconv3d: (Ci, Co) not instantiated");
}

extern "C" int dmb_conv3d_k3_c1_f32(const float* x, const float* w, float bias, const float* residual, float* y,
                                    int B, int Ci, int D, int H, int W, void* stream) {
  if (!x || !w || !y || B <= 0 || Ci <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "conv3d_c1: bad argument");
  const int ntx = cdiv(W, C1_TX), nty = cdiv(H, C1_TY), ntz = cdiv(D, C1_TZ);
  const long long nblk = (long long)B * ntx * nty * ntz;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv3d_c1: grid too large");
  hipLaunchKernelGGL(conv3d_c1_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x, w, bias, residual, y,
                     Ci, D, H, W, ntx, nty, ntz);
  return launch_status("conv3d_c1 launch failed");
}
