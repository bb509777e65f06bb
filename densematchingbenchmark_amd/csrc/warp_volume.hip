// Sample-based cost-volume builders ("fast_mode"): the target features are WARPED by per-plane or per-pixel disparity
// samples instead of shifted by integers.
//
// Reference semantics: dmb/modeling/stereo/cost_processors/utils/cat_fms.py:51-82 (fast_cat_fms), dif_fms.py:49-86
// (fast_dif_fms), layers/inverse_warp_3d.py:4-52.  The reference reaches F.grid_sample on a 5-D grid normalised with
// (size - 1) but sampled with align_corners=False (the default since torch 1.3), so every output voxel is a tri-linear
// blend around  ix = (x - s) * W / (W - 1) - 0.5,  iy = y * H / (H - 1) - 0.5,  iz = k * D / (D - 1) - 0.5  with zero
// padding -- a misaligned blend that is part of the behaviour a drop-in has to reproduce.  The arithmetic
// below follows the reference's FP32 operations one by one (no contraction into FMAs), and the eight neighbours are
// accumulated in the sampler's order, so the result is bit-identical to the reference on CPU.
//
// HBM-write-bound: one pass over the output [B, 2C | C, D, H, W]; the target features (a few MB) are gathered through
// L2.  One thread per (b, k, y, x): grid coordinates and blend weights once, then a loop over the channels.
#include "dmb_common.h"

namespace dmb {

enum { WARP_CAT = 0, WARP_DIF = 1, WARP_DIF_NORM = 2 };

struct WarpTaps {
  float w[8];      // tnw, tne, tsw, tse, bnw, bne, bsw, bse (top/bottom = plane, north/south = row, west/east = column)
  int off[4];      // nw, ne, sw, se offsets into a feature plane (0 when the tap is out of range)
  int xi[2], yi[2];   // west / east column, north / south row (0 when out of range)
  unsigned valid;  // bit t: tap t is inside the volume
};

// No contraction anywhere in this file: every product and sum below rounds as the reference's does (build.py also
// compiles this file with -ffp-contract=off).
#pragma clang fp contract(off)
__device__ inline WarpTaps warp_taps(float disp, int k, int y, int x, int D, int H, int W) {
  // inverse_warp_3d.py:33-43: mesh + disparity, then (g / (size - 1) * 2) - 1
  const float gd = ((((float)k / (float)(D - 1)) * 2.f) - 1.f);
  const float gh = ((((float)y / (float)(H - 1)) * 2.f) - 1.f);
  const float gw = (((((float)x + disp) / (float)(W - 1)) * 2.f) - 1.f);
  // grid_sample, align_corners=False: ((g + 1) * size - 1) / 2
  const float ix = ((((gw + 1.f) * (float)W) - 1.f) / 2.f);
  const float iy = ((((gh + 1.f) * (float)H) - 1.f) / 2.f);
  const float iz = ((((gd + 1.f) * (float)D) - 1.f) / 2.f);
  const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
  const float x1 = (x0 + 1.f), y1 = (y0 + 1.f), z1 = (z0 + 1.f);
  const float wx0 = (x1 - ix), wx1 = (ix - x0);
  const float wy0 = (y1 - iy), wy1 = (iy - y0);
  const float wz0 = (z1 - iz), wz1 = (iz - z0);
  WarpTaps t;
  t.w[0] = ((wx0 * wy0) * wz0);
  t.w[1] = ((wx1 * wy0) * wz0);
  t.w[2] = ((wx0 * wy1) * wz0);
  t.w[3] = ((wx1 * wy1) * wz0);
  t.w[4] = ((wx0 * wy0) * wz1);
  t.w[5] = ((wx1 * wy0) * wz1);
  t.w[6] = ((wx0 * wy1) * wz1);
  t.w[7] = ((wx1 * wy1) * wz1);
  const bool vx0 = x0 >= 0.f && x0 < (float)W, vx1 = x1 >= 0.f && x1 < (float)W;   // false for NaN samples
  const bool vy0 = y0 >= 0.f && y0 < (float)H, vy1 = y1 >= 0.f && y1 < (float)H;
  const bool vz0 = z0 >= 0.f && z0 < (float)D, vz1 = z1 >= 0.f && z1 < (float)D;
  const int xi0 = vx0 ? (int)x0 : 0, xi1 = vx1 ? (int)x1 : 0, yi0 = vy0 ? (int)y0 : 0, yi1 = vy1 ? (int)y1 : 0;
  t.xi[0] = xi0;
  t.xi[1] = xi1;
  t.yi[0] = yi0;
  t.yi[1] = yi1;
  t.off[0] = yi0 * W + xi0;
  t.off[1] = yi0 * W + xi1;
  t.off[2] = yi1 * W + xi0;
  t.off[3] = yi1 * W + xi1;
  const unsigned q = (vx0 && vy0 ? 1u : 0u) | (vx1 && vy0 ? 2u : 0u) | (vx0 && vy1 ? 4u : 0u) | (vx1 && vy1 ? 8u : 0u);
  t.valid = (vz0 ? q : 0u) | (vz1 ? q << 4 : 0u);
  return t;
}

// the sampler's accumulation: out = 0; out += v * w for each tap inside the volume, in tap order
#pragma clang fp contract(off)
__device__ inline float warp_blend(const WarpTaps& t, const float* __restrict__ plane) {
  float v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = ((t.valid | (t.valid >> 4)) >> q & 1u) ? plane[t.off[q]] : 0.f;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (t.valid >> i & 1u) acc = (acc + (v[i & 3] * t.w[i]));
  return acc;
}

// disp: [B, D, H, W] when per_pixel, else [D] (one sample per plane); the warp uses -disp (cat_fms.py:74)
template <int MODE>
__global__ __launch_bounds__(256) void warp_volume_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                          const float* __restrict__ disp, float* __restrict__ out,
                                                          int C, int D, int H, int W, int per_pixel, float p) {
  const int HW = H * W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y, b = blockIdx.z;
  if (i >= HW) return;
  const int y = i / W, x = i - y * W;
  const float s = per_pixel ? disp[((size_t)b * D + k) * HW + i] : disp[k];
  const WarpTaps t = warp_taps(-s, k, y, x, D, H, W);
  const float* Lp = L + (size_t)b * C * HW + i;
  const float* Rp = R + (size_t)b * C * HW;
  const size_t DHW = (size_t)D * HW;
  if (MODE == WARP_DIF_NORM) {
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
      const float tv = warp_blend(t, Rp + (size_t)c * HW);
      const float v = fabsf(Lp[(size_t)c * HW] * (tv > 0.f ? 1.f : 0.f) - tv);
      acc += p == 1.f ? v : (p == 2.f ? v * v : powf(v, p));
    }
    out[((size_t)b * D + k) * HW + i] = p == 1.f ? acc : (p == 2.f ? sqrtf(acc) : powf(acc, 1.f / p));
    return;
  }
  const int OC = MODE == WARP_CAT ? 2 * C : C;
  float* o = out + ((size_t)b * OC * D + k) * HW + i;
#pragma unroll 4
  for (int c = 0; c < C; ++c) {
    const float tv = warp_blend(t, Rp + (size_t)c * HW);
    const float lv = Lp[(size_t)c * HW] * (tv > 0.f ? 1.f : 0.f);   // reference features masked where the warped target <= 0 (:77)
    if (MODE == WARP_CAT) {
      __builtin_nontemporal_store(lv, o + (size_t)c * DHW);
      __builtin_nontemporal_store(tv, o + (size_t)(C + c) * DHW);
    } else {
      __builtin_nontemporal_store((lv - tv), o + (size_t)c * DHW);
    }
  }
}

template <int MODE>
static int launch_warp(const float* L, const float* R, const float* disp, float* out, int B, int C, int D, int H, int W,
                       int per_pixel, float p, hipStream_t st) {
  if (!L || !R || !disp || !out || B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "fast_fms: bad argument");
  if (D < 2 || H < 2 || W < 2) return fail(DMB_EINVAL, "fast_fms: the reference divides by (size - 1); D, H, W must be >= 2");
  if ((long long)C * H * W >= 0x7fffffffLL || D > 65535 || B > 65535) return fail(DMB_EUNSUPPORTED, "fast_fms: feature map too large");
  hipLaunchKernelGGL((warp_volume_kernel<MODE>), dim3(cdiv(H * W, 256), D, B), dim3(256), 0, st, L, R, disp, out, C, D, H,
                     W, per_pixel, p);
  return launch_status("fast_fms launch failed");
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the sample-based builders (the adjoint of the tri-linear sampler; the reference gets it from autograd through
// F.grid_sample and the expand of inverse_warp_3d.py:19-20).  With T the warped target:
//     cat:  out = [L * (T > 0), T]      dL = sum_k G_ref * (T > 0),     dT = G_tgt
//     dif:  out = L * (T > 0) - T       dL = sum_k G * (T > 0),         dT = -G
//     dR[b, c, yn, xn] = sum over (k, y, x) and the taps of (k, y, x) that land on (yn, xn) of  w_tap * dT[b, c, k, y, x]
// (the target image is the same on every plane, so the two plane taps of a (row, column) add up).  The rows a sample touches
// depend on y only and the columns on the sample: kernel A owns ONE output row y and a group of channels, walks all planes and
// columns and scatters into two LDS rows (north / south source row) with LDS atomics -- the order of the adds is the
// hardware's, as in the reference's own GPU backward (atomicAdd) -- and leaves them as partial rows; kernel B adds, per
// source row, the two or three partial rows that point at it in ascending y.  dL needs no scatter: one register per column.
//     normalised dif (dif_fms.py:82-84):  out = || L * (T > 0) - T ||_p over the channels;  with v_c the channel's difference,
//     d out / d v_c = sgn(v_c) |v_c|^(p-1) / out^(p-1)  (0 where v_c == 0 or out == 0, torch's norm backward), then as dif.
// Per-pixel samples get a gradient too (AnyNet.py:60-73 and DeepPruner.py:192 build them from predicted disparities): the
// sampler's derivative with respect to the column coordinate,
//     d T_c / d ix = sum over the taps inside the volume of  (+1 east | -1 west) * w_row * w_plane * R_c[tap],
// times d ix / d sample = -(W / 2) * (2 / (W - 1))  (grid_sample's un-normalisation, inverse_warp_3d.py:41's normalisation and
// cat_fms.py:74's sign), summed over the channels: kernel C, one thread per sample, no scatter.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WB_CG = 8;   // channels per workgroup (two per wave)

// d ||v||_p / d v_c times the upstream gradient g (FunctionsManual.cpp's norm_backward: sgn(0) = 0, zero norm -> 0)
__device__ inline float norm_grad(float v, float nrm, float g, float p) {
  if (v == 0.f || nrm == 0.f) return 0.f;
  if (p == 1.f) return v > 0.f ? g : -g;
  if (p == 2.f) return g * (v / nrm);
  const float a = powf(fabsf(v), p - 1.f) * (g / powf(nrm, p - 1.f));
  return v > 0.f ? a : -a;
}

// upstream gradients of one voxel's channel c: g_ref (reference half) and g_tgt (the warped target T)
template <int MODE>
__device__ inline void warp_upstream(const float* __restrict__ G, const float* __restrict__ nrm, size_t vox, size_t o, int c, int C,
                                     size_t DHW, float lv, float tv, float p, float& g_ref, float& g_tgt) {
  if (MODE == WARP_DIF_NORM) {
    const float v = (tv > 0.f ? lv : 0.f) - tv;
    g_ref = norm_grad(v, nrm[vox], G[vox], p);
    g_tgt = -g_ref;
  } else {
    g_ref = G[o + (size_t)c * DHW];
    g_tgt = MODE == WARP_CAT ? G[o + (size_t)(C + c) * DHW] : -g_ref;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void warp_volume_bwd_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                              const float* __restrict__ disp, const float* __restrict__ G,
                                                              const float* __restrict__ nrm, float* __restrict__ dL,
                                                              float* __restrict__ part, int C, int D, int H, int W,
                                                              int per_pixel, float p) {
  extern __shared__ float rows[];   // [2][WB_CG][W]
  const int y = blockIdx.x, c0 = blockIdx.y * WB_CG, b = blockIdx.z;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int HW = H * W;
  const size_t DHW = (size_t)D * HW;
  for (int i = threadIdx.x; i < 2 * WB_CG * W; i += 256) rows[i] = 0.f;
  __syncthreads();
  const int OC = MODE == WARP_CAT ? 2 * C : C;
  const float* Rb = R + (size_t)b * C * HW;
  for (int x = lane; x < W; x += 64) {
    float dl[2] = {0.f, 0.f};
    for (int k = 0; k < D; ++k) {
      const float s = per_pixel ? disp[((size_t)b * D + k) * HW + y * W + x] : disp[k];
      const WarpTaps t = warp_taps(-s, k, y, x, D, H, W);
      // weight of (north | south, west | east): the two plane taps of one image point add up
      float wq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) wq[q] = ((t.valid >> q & 1u) ? t.w[q] : 0.f) + ((t.valid >> (q + 4) & 1u) ? t.w[q + 4] : 0.f);
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int cl = wave + 4 * cc, c = c0 + cl;
        if (c >= C) continue;
        const float tv = warp_blend(t, Rb + (size_t)c * HW);
        const size_t o = ((size_t)b * OC * D + k) * HW + (size_t)y * W + x;
        const size_t vox = ((size_t)b * D + k) * HW + (size_t)y * W + x;
        float g_ref, g_tgt;
        warp_upstream<MODE>(G, nrm, vox, o, c, C, DHW, MODE == WARP_DIF_NORM ? L[((size_t)b * C + c) * HW + (size_t)y * W + x] : 0.f,
                            tv, p, g_ref, g_tgt);
        if (tv > 0.f) dl[cc] += g_ref;
        float* r0 = rows + (size_t)cl * W;                 // north source row
        float* r1 = rows + (size_t)(WB_CG + cl) * W;       // south source row
        if (wq[0] != 0.f) atomicAdd(r0 + t.xi[0], wq[0] * g_tgt);
        if (wq[1] != 0.f) atomicAdd(r0 + t.xi[1], wq[1] * g_tgt);
        if (wq[2] != 0.f) atomicAdd(r1 + t.xi[0], wq[2] * g_tgt);
        if (wq[3] != 0.f) atomicAdd(r1 + t.xi[1], wq[3] * g_tgt);
      }
    }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int c = c0 + wave + 4 * cc;
      if (c < C) dL[((size_t)b * C + c) * HW + (size_t)y * W + x] = dl[cc];
    }
  }
  __syncthreads();
  // partial rows: part[b][c][y][ns][W]
  for (int i = threadIdx.x; i < 2 * WB_CG * W; i += 256) {
    const int ns = i / (WB_CG * W), r = i - ns * (WB_CG * W), cl = r / W, x = r - cl * W;
    if (c0 + cl < C) part[((((size_t)b * C + c0 + cl) * H + y) * 2 + ns) * W + x] = rows[i];
  }
}

// dR[b, c, yn, x] = sum, in ascending y, of the partial rows of the output rows y whose north (south) source row is yn
__global__ __launch_bounds__(256) void warp_volume_bwd_rows_kernel(const float* __restrict__ part, float* __restrict__ dR,
                                                                   int C, int H, int W) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int bc = blockIdx.y;
  if (i >= H * W) return;
  const int yn = i / W, x = i - yn * W;
  float acc = 0.f;
  for (int y = max(0, yn - 2); y <= min(H - 1, yn + 2); ++y) {
    const float gh = ((((float)y / (float)(H - 1)) * 2.f) - 1.f);
    const float iy = ((((gh + 1.f) * (float)H) - 1.f) / 2.f);
    const float y0 = floorf(iy), y1 = y0 + 1.f;
    const float* p = part + (((size_t)bc * H + y) * 2) * W + x;
    if (y0 >= 0.f && y0 < (float)H && (int)y0 == yn) acc += p[0];
    if (y1 >= 0.f && y1 < (float)H && (int)y1 == yn) acc += p[W];
  }
  dR[(size_t)bc * H * W + i] = acc;
}

// Kernel C: dS[b, k, y, x] for per-pixel samples.  The taps' x weights are (x1 - ix) west and (ix - x0) east, so the derivative
// of a tap's weight with respect to ix is -/+ (row weight * plane weight); the planes share the image, so both plane taps of
// an image point use the same feature value.  Accumulated over the channels in the order grid_sample's backward does.
template <int MODE>
__global__ __launch_bounds__(256) void warp_volume_bwd_samples_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                                      const float* __restrict__ disp, const float* __restrict__ G,
                                                                      const float* __restrict__ nrm, float* __restrict__ dS,
                                                                      int C, int D, int H, int W, float p) {
  const int HW = H * W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y, b = blockIdx.z;
  if (i >= HW) return;
  const int y = i / W, x = i - y * W;
  const size_t vox = ((size_t)b * D + k) * HW + i;
  const WarpTaps t = warp_taps(-disp[vox], k, y, x, D, H, W);
  // d w_tap / d ix, plane taps of one image point added up; w = wx * wy * wz, so the quotient is taken from the factors again
  const float gd = ((((float)k / (float)(D - 1)) * 2.f) - 1.f), gh = ((((float)y / (float)(H - 1)) * 2.f) - 1.f);
  const float iy = ((((gh + 1.f) * (float)H) - 1.f) / 2.f), iz = ((((gd + 1.f) * (float)D) - 1.f) / 2.f);
  const float wy1 = iy - floorf(iy), wy0 = (floorf(iy) + 1.f) - iy, wz1 = iz - floorf(iz), wz0 = (floorf(iz) + 1.f) - iz;
  const float wyq[4] = {wy0, wy0, wy1, wy1};
  float dq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float wz = ((t.valid >> q & 1u) ? wz0 : 0.f) + ((t.valid >> (q + 4) & 1u) ? wz1 : 0.f);
    dq[q] = (q & 1 ? wyq[q] : -wyq[q]) * wz;
  }
  const int OC = MODE == WARP_CAT ? 2 * C : C;
  const size_t DHW = (size_t)D * HW;
  const size_t o = ((size_t)b * OC * D + k) * HW + i;
  const float* Rb = R + (size_t)b * C * HW;
  const unsigned any = t.valid | (t.valid >> 4);
  float acc = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* plane = Rb + (size_t)c * HW;
    float gx = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (any >> q & 1u) gx += plane[t.off[q]] * dq[q];
    float g_ref, g_tgt;
    const float tv = MODE == WARP_DIF_NORM ? warp_blend(t, plane) : 0.f;
    warp_upstream<MODE>(G, nrm, vox, o, c, C, DHW, MODE == WARP_DIF_NORM ? L[((size_t)b * C + c) * HW + i] : 0.f, tv, p, g_ref, g_tgt);
    acc += g_tgt * gx;
  }
  dS[vox] = -((acc * (0.5f * (float)W)) * 2.f / (float)(W - 1));
}

}  // namespace dmb

using namespace dmb;

extern "C" int dmb_fast_fms_bwd_f32(const float* L, const float* R, const float* disp_sample, const float* dvol,
                                    const float* norm_out, float* dL, float* dR, float* d_samples, float* partial, int B, int C,
                                    int D, int H, int W, int per_pixel, int mode, float p, void* stream) {
  if (!L || !R || !disp_sample || !dvol || !dL || !dR || !partial || B <= 0 || C <= 0 || D < 2 || H < 2 || W < 2 || mode < 0 ||
      mode > 2 || (mode == WARP_DIF_NORM && (!norm_out || !(p > 0.f))) || (d_samples && !per_pixel))
    return fail(DMB_EINVAL, "fast_fms_bwd: bad argument");
  if ((long long)C * H * W >= 0x7fffffffLL || H > 65535 || D > 65535 || B > 65535 || (size_t)2 * WB_CG * W * 4 > 64 * 1024)
    return fail(DMB_EUNSUPPORTED, "fast_fms_bwd: feature map too large");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)2 * WB_CG * W * sizeof(float);
  const dim3 grid(H, cdiv(C, WB_CG), B), sgrid(cdiv(H * W, 256), D, B);
#define DMB_WARP_BWD(MODE)                                                                                                        \
  do {                                                                                                                            \
    hipLaunchKernelGGL((warp_volume_bwd_kernel<MODE>), grid, dim3(256), lds, st, L, R, disp_sample, dvol, norm_out, dL, partial,  \
                       C, D, H, W, per_pixel, p);                                                                                 \
    if (d_samples)                                                                                                                \
      hipLaunchKernelGGL((warp_volume_bwd_samples_kernel<MODE>), sgrid, dim3(256), 0, st, L, R, disp_sample, dvol, norm_out,      \
                         d_samples, C, D, H, W, p);                                                                               \
  } while (0)
  if (mode == WARP_CAT) DMB_WARP_BWD(WARP_CAT);
  else if (mode == WARP_DIF) DMB_WARP_BWD(WARP_DIF);
  else DMB_WARP_BWD(WARP_DIF_NORM);
#undef DMB_WARP_BWD
  hipLaunchKernelGGL(warp_volume_bwd_rows_kernel, dim3(cdiv(H * W, 256), B * C), dim3(256), 0, st, partial, dR, C, H, W);
  return launch_status("fast_fms_bwd launch failed");
}

extern "C" int dmb_fast_cat_fms_f32(const float* L, const float* R, const float* disp_sample, float* out, int B, int C,
                                    int D, int H, int W, int per_pixel, void* stream) {
  return launch_warp<WARP_CAT>(L, R, disp_sample, out, B, C, D, H, W, per_pixel, 0.f, (hipStream_t)stream);
}

extern "C" int dmb_fast_dif_fms_f32(const float* L, const float* R, const float* disp_sample, float* out, int B, int C,
                                    int D, int H, int W, int per_pixel, int normalize, float p, void* stream) {
  if (normalize) {
    if (!(p > 0.f)) return fail(DMB_EINVAL, "fast_dif_fms: norm order must be positive");
    return launch_warp<WARP_DIF_NORM>(L, R, disp_sample, out, B, C, D, H, W, per_pixel, p, (hipStream_t)stream);
  }
  return launch_warp<WARP_DIF>(L, R, disp_sample, out, B, C, D, H, W, per_pixel, 0.f, (hipStream_t)stream);
}
