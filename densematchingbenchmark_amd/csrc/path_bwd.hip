// Backward passes of the HBM-bound ends of the cost path (second part): the cost-volume builders and the
// disparity regression.  Autograd of
//   cat_fms / dif_fms                            dmb/modeling/stereo/cost_processors/utils/{cat_fms.py:7-48, dif_fms.py:7-46}
//   FasterSoftArgmin / SoftArgmin                dmb/modeling/stereo/disp_predictors/{faster_soft_argmin.py:51-75, soft_argmin.py:45-75}
//   F.interpolate(trilinear, align_corners=True) dmb/modeling/stereo/cost_processors/aggregators/PSMNet.py:74-93
// Each kernel reads its large operand once; nothing the size of the [B, D, H, W] cost volume is written.
#include "dmb_common.h"
#include "interp.h"

#pragma clang fp contract(off)

namespace dmb {

// ------------------------------------------------------------------------------------------------------------------
// Cost-volume builders.  Forward: vol[:C, k, y, x] = L[y, x], vol[C:, k, y, x] = R[y, x - d_k] on the valid columns
// x in [max(d_k, 0), min(W, W + d_k)), zero elsewhere.  Backward: dL[y, x] = sum_k [x valid] dvol[c, k, y, x],
// dR[y, x'] = sum_k [x' + d_k valid] dvol[C + c, k, y, x' + d_k]  (difference volume: one tensor, the second term negated).
// One thread = one (b, c, y, x) of dL or dR walking the D planes (a wave reads 256 contiguous bytes per plane).
// ------------------------------------------------------------------------------------------------------------------
template <bool DIF>
__global__ __launch_bounds__(256) void volume_bwd_kernel(const float* __restrict__ dvol, float* __restrict__ dL,
                                                         float* __restrict__ dR, int C, int H, int W, int D, DispIdx idx) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const int y = blockIdx.y % H, side = blockIdx.y / H;   // side 0: dL, 1: dR
  const int c = blockIdx.z % C, b = blockIdx.z / C;
  const size_t HW = (size_t)H * W;
  const int VC = DIF ? C : 2 * C;
  const float* p = dvol + (((size_t)b * VC + ((DIF || side == 0) ? c : C + c)) * D) * HW + (size_t)y * W;
  float s = 0.f;
  for (int k = 0; k < D; ++k) {
    const int d = idx.d[k];
    const int xv = side == 0 ? x : x + d;                 // the volume column this plane contributes from
    const int lo = d > 0 ? d : 0, hi = d < 0 ? W + d : W;
    if (xv >= lo && xv < hi) s += p[(size_t)k * HW + xv];
  }
  float* out = (side == 0 ? dL : dR) + (((size_t)b * C + c) * H + y) * W + x;
  *out = (DIF && side == 1) ? -s : s;
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of the aggregator's FIRST convolution on a concatenation / difference volume of unit disparity step, without the
// volume (the backward twin of csrc/catconv.hip).  With V = cat_fms(L, R) (cat_fms.py:7-48: V[ci, z, y, x] = L[ci, y, x] and
// V[C + ci, z, y, x] = R[ci, y, x - z] for x >= z, zero otherwise) and zero padding around it,
//   dW[co, ci, dz, dy, dx]     = sum_{y, x} L[ci, y + dy - 1, x + dx - 1] * GL_{dz, dx}[co, y, x]
//   dW[co, C + ci, dz, dy, dx] = sum_{y, u} R[ci, y + dy - 1, u]          * GR_{dz, dx}[co, y, u]
//   GL_{dz, dx}[co, y, x] = sum_z dc[co, z, y, x]                  over 0 <= z' = z + dz - 1 < D,  x + dx - 1 >= z'
//   GR_{dz, dx}[co, y, u] = sum_z dc[co, z, y, u + z' - dx + 1]    over 0 <= z' < D,  u + z' < W,  the column inside [0, W)
// i.e. ONE pass over dc that folds z into 2 x 9 maps per output channel (this kernel), then two 2-D weight gradients of the feature
// maps against those maps (dmb_conv2d_wgrad_f32; the caller picks tap dx' = dx of the left result and the centre column of the
// right one).  The 3-D weight gradient of the materialised volume is 64 x 32 x 27 x 25 M multiply-adds (1.22 ms + 0.06 ms for the
// volume at the training crop); this is 0.2 GB of traffic and two small GEMMs.
// maps_*: [B, 9 * Co, H, W], channel (dz * 3 + dx) * Co + co.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cat_wgrad_maps_kernel(const float* __restrict__ dc, float* __restrict__ gl, float* __restrict__ gr,
                                                             int Co, int D, int H, int W) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int HW = H * W;
  if (pix >= HW) return;
  const int y = pix / W, x = pix - y * W, co = blockIdx.y, b = blockIdx.z;
  const float* p = dc + ((size_t)(b * Co + co) * D) * HW + (size_t)y * W;
  float g[3][3], h[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) g[a][c] = h[a][c] = 0.f;
  for (int z = 0; z < D; ++z) {
    const float* row = p + (size_t)z * HW;
    const float v = row[x];
    float d[5];   // the right half's diagonal walk: dc at column x + z' - dx + 1 = x + z + (dz - dx) = x + z + k - 2, 0 outside the row
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int xx = x + z + k - 2;
      d[k] = (xx >= 0 && xx < W) ? row[xx] : 0.f;
    }
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      const int zp = z + dz - 1;
      if (zp < 0 || zp >= D) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        if (x + dx - 1 >= zp) g[dz][dx] += v;
        if (x + zp < W) h[dz][dx] += d[dz - dx + 2];
      }
    }
  }
  const size_t ob = ((size_t)b * 9 * Co + co) * HW + pix;
#pragma unroll
  for (int dz = 0; dz < 3; ++dz)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      gl[ob + (size_t)(dz * 3 + dx) * Co * HW] = g[dz][dx];
      gr[ob + (size_t)(dz * 3 + dx) * Co * HW] = h[dz][dx];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Soft-argmin: disp = sum_k p_k s_k, p = softmax(alpha * cost)  =>  dcost_k = g * alpha * p_k * (s_k - disp).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void soft_argmin_bwd_kernel(const float* __restrict__ cost, const float* __restrict__ disp,
                                                              const float* __restrict__ g, float* __restrict__ dcost, int D,
                                                              long long HW, float alpha, DispVal dv) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= HW) return;
  const int b = blockIdx.y;
  const float* cp = cost + (size_t)b * D * HW + i;
  float m = -INFINITY;
  for (int k = 0; k < D; ++k) m = fmaxf(m, cp[(size_t)k * HW] * alpha);
  double s = 0.0;
  for (int k = 0; k < D; ++k) s += (double)__expf(cp[(size_t)k * HW] * alpha - m);
  const float inv = (float)(1.0 / s);
  const float dsp = disp[(size_t)b * HW + i], ga = g[(size_t)b * HW + i] * alpha;
  float* op = dcost + (size_t)b * D * HW + i;
  for (int k = 0; k < D; ++k) {
    const float pk = __expf(cp[(size_t)k * HW] * alpha - m) * inv;
    op[(size_t)k * HW] = ga * pk * (dv.v[k] - dsp);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Up-sampling + soft-argmin, backward, without the [B, Do, Ho, Wo] volume: pass A (one thread per output pixel)
// re-creates the pixel's up-sampled logits from the low-resolution cost exactly as the forward kernel does, forms
// dcost_k and folds it along z straight into the (H, W)-interpolated planes it came from:
//   t[b, zi, yo, xo] = sum_k wz(zi <- k) * dcost_k            ([B, Di, Ho, Wo]: Do/Di times smaller than the volume)
// pass B contracts t over the (yo, xo) neighbourhood of every low-resolution voxel with the bilinear weights.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample_regress_bwd_z_kernel(const float* __restrict__ x, const float* __restrict__ disp,
                                                                     const float* __restrict__ g, float* __restrict__ t, int Di,
                                                                     int Hi, int Wi, int Do, int Ho, int Wo, float sd, float sh,
                                                                     float sw, float alpha, DispVal dv) {
  // (round 6) the z interpolation of an output plane is the same for every pixel: one table per workgroup instead of a lerp set-up
  // per plane, pixel and walk (2 x 192 x 9 instructions of a thread's ~10 000: the walks are what this kernel's time goes into)
  __shared__ Lerp lzt[DMB_MAX_DISP_SAMPLES];
  for (int k = threadIdx.x; k < Do; k += 256) lzt[k] = lerp_setup(k, Di, sd);
  __syncthreads();
  const int xo = blockIdx.x * 256 + threadIdx.x;
  if (xo >= Wo) return;
  const int yo = blockIdx.y, b = blockIdx.z;
  const Lerp ly = lerp_setup(yo, Hi, sh), lx = lerp_setup(xo, Wi, sw);
  const float* xb = x + (size_t)b * Di * Hi * Wi;
  const size_t plane = (size_t)Hi * Wi;
  const size_t o00 = (size_t)ly.i0 * Wi + lx.i0, o01 = (size_t)ly.i0 * Wi + lx.i1;
  const size_t o10 = (size_t)ly.i1 * Wi + lx.i0, o11 = (size_t)ly.i1 * Wi + lx.i1;
  auto hw = [&](int zi) {
    const float* pz = xb + (size_t)zi * plane;
    const float a0 = lerp2(pz[o00], lx.w0, pz[o01], lx.w1);
    const float a1 = lerp2(pz[o10], lx.w0, pz[o11], lx.w1);
    return lerp2(a0, ly.w0, a1, ly.w1);
  };
  // the pixel's up-sampled logit of output plane zo (two cached input planes, as in the forward kernels)
  int cz0 = -1, cz1 = -1;
  float h0 = 0.f, h1 = 0.f;
  auto logit = [&](int zo, Lerp& lz) {
    lz = lzt[zo];
    if (lz.i0 != cz0) {
      h0 = (lz.i0 == cz1) ? h1 : hw(lz.i0);
      cz0 = lz.i0;
    }
    if (lz.i1 != cz1) {
      h1 = (lz.i1 == cz0) ? h0 : hw(lz.i1);
      cz1 = lz.i1;
    }
    return lerp2(h0, lz.w0, h1, lz.w1) * alpha;
  };
  Lerp lz;
  // (round 5) running maximum and normaliser in ONE walk over the planes (the walk -- a lerp set-up and a blend per plane -- is
  // what this kernel's time goes into, not memory): the sum is rescaled when the maximum moves, which it does a few times per pixel
  float m = -INFINITY;
  double s = 0.0;
  for (int zo = 0; zo < Do; ++zo) {
    const float v = logit(zo, lz);
    if (v > m) {
      s *= (m == -INFINITY) ? 0.0 : (double)__expf(m - v);
      m = v;
    }
    s += (double)__expf(v - m);
  }
  cz0 = cz1 = -1;
  const float inv = (float)(1.0 / s);
  const size_t pix = ((size_t)b * Ho + yo) * Wo + xo;
  const float dsp = disp[pix], ga = g[pix] * alpha;
  // fold along z: input planes are visited in ascending order, so two running sums (planes ia, ia + 1) suffice
  float* tp = t + (size_t)b * Di * Ho * Wo + (size_t)yo * Wo + xo;
  const size_t tstride = (size_t)Ho * Wo;
  int ia = 0;
  float acc_a = 0.f, acc_b = 0.f;
  for (int zo = 0; zo < Do; ++zo) {
    const float pk = __expf(logit(zo, lz) - m) * inv;
    const float dck = ga * pk * (dv.v[zo] - dsp);
    while (ia < lz.i0) {
      tp[(size_t)ia * tstride] = acc_a;
      acc_a = acc_b;
      acc_b = 0.f;
      ++ia;
    }
    acc_a = fmaf(dck, lz.w0, acc_a);
    if (lz.i1 != lz.i0)
      acc_b = fmaf(dck, lz.w1, acc_b);
    else
      acc_a = fmaf(dck, lz.w1, acc_a);
  }
  while (ia < Di) {
    tp[(size_t)ia * tstride] = acc_a;
    acc_a = acc_b;
    acc_b = 0.f;
    ++ia;
  }
}

// The same z fold for a gradient that arrives on the up-sampled volume itself (a loss on the costs, e.g. the focal loss):
// t[b, zi, yo, xo] (+)= sum_k wz(zi <- k) * dcost[b, k, yo, xo]; accumulate = 1 adds to what pass A of the regression wrote.
__global__ __launch_bounds__(256) void upsample_bwd_z_kernel(const float* __restrict__ dcost, float* __restrict__ t, int Di, int Do,
                                                             int Ho, int Wo, float sd, int accumulate) {
  const int xo = blockIdx.x * 256 + threadIdx.x;
  if (xo >= Wo) return;
  const int yo = blockIdx.y, b = blockIdx.z;
  const size_t tstride = (size_t)Ho * Wo;
  const float* dp = dcost + (size_t)b * Do * tstride + (size_t)yo * Wo + xo;
  float* tp = t + (size_t)b * Di * tstride + (size_t)yo * Wo + xo;
  int ia = 0;
  float acc_a = 0.f, acc_b = 0.f;
  auto flush = [&]() {
    tp[(size_t)ia * tstride] = accumulate ? tp[(size_t)ia * tstride] + acc_a : acc_a;
    acc_a = acc_b;
    acc_b = 0.f;
    ++ia;
  };
  for (int zo = 0; zo < Do; ++zo) {
    const Lerp lz = lerp_setup(zo, Di, sd);
    const float dck = dp[(size_t)zo * tstride];
    while (ia < lz.i0) flush();
    acc_a = fmaf(dck, lz.w0, acc_a);
    if (lz.i1 != lz.i0)
      acc_b = fmaf(dck, lz.w1, acc_b);
    else
      acc_a = fmaf(dck, lz.w1, acc_a);
  }
  while (ia < Di) flush();
}

// dlow[b, zi, yl, xl] = sum_{yo, xo} wy(yl <- yo) * wx(xl <- xo) * t[b, zi, yo, xo]; one thread per low-resolution voxel.
// Output rows / columns that blend input index i: src = scale * o in (i - 1, i + 1).
__global__ __launch_bounds__(256) void upsample_regress_bwd_hw_kernel(const float* __restrict__ t, float* __restrict__ dx, int Di,
                                                                      int Hi, int Wi, int Ho, int Wo, float sh, float sw) {
  const int xl = blockIdx.x * 256 + threadIdx.x;
  if (xl >= Wi) return;
  const int yl = blockIdx.y % Hi, zi = blockIdx.y / Hi, b = blockIdx.z;
  auto range = [](int i, float scale, int out, int& lo, int& hi) {
    if (scale <= 0.f) {   // a single output position blends input 0 only
      lo = 0;
      hi = i == 0 ? out - 1 : -1;
      return;
    }
    lo = (int)floorf((float)(i - 1) / scale) - 1;
    hi = (int)ceilf((float)(i + 1) / scale) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > out - 1 ? out - 1 : hi;
  };
  int ylo, yhi, xlo, xhi;
  range(yl, sh, Ho, ylo, yhi);
  range(xl, sw, Wo, xlo, xhi);
  const float* tb = t + ((size_t)b * Di + zi) * Ho * Wo;
  float acc = 0.f;
  // (round 5) The column weights of this thread's window do not depend on the row: they are computed ONCE (the window of a 4x
  // up-sampling is 9 .. 12 columns wide) instead of once per (row, column) -- the kernel spent its time in ~100 lerp set-ups per
  // output, not in its 100 loads (0.137 -> see profiles/r05_train_pmc.csv).  Wider windows (other scale factors) take the old loop.
  constexpr int KMAX = 16;
  if (xhi - xlo < KMAX) {
    float wxs[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int xo = xlo + k;
      float wv = 0.f;
      if (xo <= xhi) {
        const Lerp lx = lerp_setup(xo, Wi, sw);
        wv = (lx.i0 == xl ? lx.w0 : 0.f) + (lx.i1 == xl ? lx.w1 : 0.f);
      }
      wxs[k] = wv;
    }
    for (int yo = ylo; yo <= yhi; ++yo) {
      const Lerp ly = lerp_setup(yo, Hi, sh);
      const float wy = (ly.i0 == yl ? ly.w0 : 0.f) + (ly.i1 == yl ? ly.w1 : 0.f);
      if (wy == 0.f) continue;
      const float* tr = tb + (size_t)yo * Wo + xlo;
      float row = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (wxs[k] != 0.f) row = fmaf(tr[k], wxs[k], row);     // same terms, same ascending order as the loop below
      acc = fmaf(row, wy, acc);
    }
    dx[(((size_t)b * Di + zi) * Hi + yl) * Wi + xl] = acc;
    return;
  }
  for (int yo = ylo; yo <= yhi; ++yo) {
    const Lerp ly = lerp_setup(yo, Hi, sh);
    const float wy = (ly.i0 == yl ? ly.w0 : 0.f) + (ly.i1 == yl ? ly.w1 : 0.f);
    if (wy == 0.f) continue;
    float row = 0.f;
    for (int xo = xlo; xo <= xhi; ++xo) {
      const Lerp lx = lerp_setup(xo, Wi, sw);
      const float wx = (lx.i0 == xl ? lx.w0 : 0.f) + (lx.i1 == xl ? lx.w1 : 0.f);
      if (wx != 0.f) row = fmaf(tb[(size_t)yo * Wo + xo], wx, row);
    }
    acc = fmaf(row, wy, acc);
  }
  dx[(((size_t)b * Di + zi) * Hi + yl) * Wi + xl] = acc;
}

// (round 6) The same contraction for a group of R low-resolution rows per workgroup, in the two steps the sum above already has:
// row[yo][xl] = sum_xo t[yo][xo] wx(xl <- xo) for the high-resolution rows the group touches (into LDS; the 16 loads of a row sum
// are independent and a high-resolution row is contracted ONCE, not once per low-resolution row that blends it), then
// dlow[yl][xl] = sum_yo row[yo][xl] wy(yl <- yo) from LDS.  Same terms in the same order as upsample_regress_bwd_hw_kernel's
// windowed path: bit-identical.  That kernel walks its 9 .. 12 rows one memory round trip after the other (113 us for 100 MB at
// the training crop); here a thread has a row's loads in flight together and a workgroup reads its rows once.
template <int R>
__global__ __launch_bounds__(256) void upsample_bwd_hw_rows_kernel(const float* __restrict__ t, float* __restrict__ dx, int Di, int Hi,
                                                                   int Wi, int Ho, int Wo, float sh, float sw, int nrmax) {
  extern __shared__ float rs[];   // [nrmax][Wi]
  const int y0 = blockIdx.x * R, zi = blockIdx.y, b = blockIdx.z;
  auto range = [](int i, float scale, int out, int& lo, int& hi) {   // (scale > 0: the launcher's condition)
    lo = (int)floorf((float)(i - 1) / scale) - 1;
    hi = (int)ceilf((float)(i + 1) / scale) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > out - 1 ? out - 1 : hi;
  };
  int glo, ghi, tmp;
  range(y0, sh, Ho, glo, tmp);
  range(min(y0 + R - 1, Hi - 1), sh, Ho, tmp, ghi);
  const int nr = min(ghi - glo + 1, nrmax);
  const float* tb = t + ((size_t)b * Di + zi) * Ho * Wo;
  constexpr int KMAX = 16;
  const int RL = Wi >= 256 ? 1 : 256 / Wi;          // row lanes: threads (rl, xl)
  const int XW = Wi >= 256 ? 256 : Wi;
  const int rl = threadIdx.x / XW, xc = threadIdx.x - rl * XW;
  for (int xb = 0; xb < Wi; xb += XW) {
    const int xl = xb + xc;
    if (rl < RL && xl < Wi) {
      int xlo, xhi;
      range(xl, sw, Wo, xlo, xhi);
      float wxs[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        const int xo = xlo + k;
        float wv = 0.f;
        if (xo <= xhi) {
          const Lerp lx = lerp_setup(xo, Wi, sw);
          wv = (lx.i0 == xl ? lx.w0 : 0.f) + (lx.i1 == xl ? lx.w1 : 0.f);
        }
        wxs[k] = wv;
      }
      // two rows' loads in flight per thread (a row's 16 are independent; the rows of a thread were one round trip each)
      for (int r = rl; r < nr; r += 2 * RL) {
        const bool two = r + RL < nr;
        const float* tr0 = tb + (size_t)(glo + r) * Wo + xlo;
        const float* tr1 = tr0 + (size_t)RL * Wo;
        float v0[KMAX], v1[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          v0[k] = wxs[k] != 0.f ? tr0[k] : 0.f;
          v1[k] = (two && wxs[k] != 0.f) ? tr1[k] : 0.f;
        }
        float row0 = 0.f, row1 = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
          if (wxs[k] != 0.f) {
            row0 = fmaf(v0[k], wxs[k], row0);
            row1 = fmaf(v1[k], wxs[k], row1);
          }
        rs[r * Wi + xl] = row0;
        if (two) rs[(r + RL) * Wi + xl] = row1;
      }
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < R * Wi; idx += 256) {
    const int yl = y0 + idx / Wi, xl = idx % Wi;
    if (yl >= Hi) break;
    int ylo, yhi;
    range(yl, sh, Ho, ylo, yhi);
    float acc = 0.f;
    for (int yo = ylo; yo <= yhi; ++yo) {
      const Lerp ly = lerp_setup(yo, Hi, sh);
      const float wy = (ly.i0 == yl ? ly.w0 : 0.f) + (ly.i1 == yl ? ly.w1 : 0.f);
      if (wy == 0.f) continue;
      acc = fmaf(rs[(yo - glo) * Wi + xl], wy, acc);
    }
    dx[(((size_t)b * Di + zi) * Hi + yl) * Wi + xl] = acc;
  }
}

// Launch of the (y, x) contraction: the row-group form where it applies (both scales > 0, column windows of fewer than 16
// outputs, the group's row sums within 64 KB of LDS), else one thread per low-resolution voxel.
static void launch_upsample_bwd_hw(const float* t, float* dx, int B, int planes, int Hi, int Wi, int Ho, int Wo, float sh, float sw,
                                   hipStream_t st) {
  constexpr int R = 8;
  if (sh > 0.f && sw > 0.f && Hi >= 2 && Wi >= 2) {
    const int win = (int)ceilf(2.f / sw) + 5;                 // columns a low-resolution pixel may blend (bound of xhi - xlo + 1)
    const int nrmax = (int)ceilf((float)(R + 1) / sh) + 6;    // rows a group of R may blend (bound of ghi - glo + 1)
    const size_t lds = (size_t)nrmax * Wi * sizeof(float);
    if (win <= 16 && lds <= 64 * 1024 && cdiv(Hi, R) <= 65535 && planes <= 65535) {
      hipLaunchKernelGGL((upsample_bwd_hw_rows_kernel<R>), dim3(cdiv(Hi, R), planes, B), dim3(256), lds, st, t, dx, planes, Hi, Wi, Ho, Wo,
                         sh, sw, nrmax);
      return;
    }
  }
  hipLaunchKernelGGL(upsample_regress_bwd_hw_kernel, dim3(cdiv(Wi, 256), planes * Hi, B), dim3(256), 0, st, t, dx, planes, Hi, Wi, Ho, Wo,
                     sh, sw);
}

// ------------------------------------------------------------------------------------------------------------------
// AcfNet's learned up-sampling, nn.ConvTranspose3d(1, 1, 8, 4, 2) (aggregators/AcfNet.py:55-57): y[o] = sum x[i] w[k],
// o = 4 i - 2 + k per axis.  Backward:
//   dx[i] = sum_k dy[4 i - 2 + k] w[k]                       (a stride-4 convolution of dy with the same 8 x 8 x 8 taps)
//   dw[k] = sum_{b, i} x[b, i] dy[b, 4 i - 2 + k]
// Both walk the 8 x 8 rows of dy that touch an input voxel and read 8 contiguous floats of each; dy is read a few times
// out of the caches (neighbouring voxels share half of every row segment), 3 x 8 = 24 multiply-adds per 32 bytes.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void k8_row(const float* __restrict__ row, int xi, int Wo, float (&d)[8]) {
  const int x0 = 4 * xi - 2;   // even: 8-byte aligned pairs
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int xx = x0 + 2 * p;
    if (xx >= 0 && xx + 1 < Wo) {
      const float2 v = *reinterpret_cast<const float2*>(row + xx);
      d[2 * p] = v.x;
      d[2 * p + 1] = v.y;
    } else {
      d[2 * p] = (xx >= 0 && xx < Wo) ? row[xx] : 0.f;
      d[2 * p + 1] = (xx + 1 >= 0 && xx + 1 < Wo) ? row[xx + 1] : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void deconv_k8s4_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                             float* __restrict__ dx, int D, int H, int W) {
  __shared__ float ws[512];
  for (int t = threadIdx.x; t < 512; t += 256) ws[t] = w[t];
  __syncthreads();
  const int xi = blockIdx.x * 256 + threadIdx.x;
  if (xi >= W) return;
  const int yi = blockIdx.y % H, zi = blockIdx.y / H, b = blockIdx.z;
  const int Do = 4 * D, Ho = 4 * H, Wo = 4 * W;
  const float* dyb = dy + (size_t)b * Do * Ho * Wo;
  float acc = 0.f;
  for (int kz = 0; kz < 8; ++kz) {
    const int zo = 4 * zi - 2 + kz;
    if (zo < 0 || zo >= Do) continue;
    for (int ky = 0; ky < 8; ++ky) {
      const int yo = 4 * yi - 2 + ky;
      if (yo < 0 || yo >= Ho) continue;
      float d[8];
      k8_row(dyb + ((size_t)zo * Ho + yo) * Wo, xi, Wo, d);
      const float* wr = ws + (kz * 8 + ky) * 8;
#pragma unroll
      for (int kx = 0; kx < 8; ++kx) acc = fmaf(d[kx], wr[kx], acc);
    }
  }
  dx[(((size_t)b * D + zi) * H + yi) * W + xi] = acc;
}

// one block = one (kz, ky) and a chunk of input rows: 8 private sums per thread, block reduction, partials per chunk
constexpr int K8_CHUNKS = 32;
__global__ __launch_bounds__(256) void deconv_k8s4_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             double* __restrict__ ws, int B, int D, int H, int W) {
  __shared__ double sm[4][8];
  const int kz = blockIdx.y >> 3, ky = blockIdx.y & 7;
  const int Do = 4 * D, Ho = 4 * H, Wo = 4 * W;
  const long long rows = (long long)B * D * H;
  const long long per = (rows + gridDim.x - 1) / gridDim.x;
  const long long r0 = blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  double tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // threads walk the chunk's (row, column) pairs flattened, so narrow rows still fill the block
  const long long n = (r1 - r0) * W;
  int run = 0;
  for (long long i = threadIdx.x; i < n; i += 256) {
    const long long r = r0 + i / W;
    const int xi = (int)(i % W);
    const int yi = (int)(r % H), zi = (int)((r / H) % D), b = (int)(r / ((long long)H * D));
    const int zo = 4 * zi - 2 + kz, yo = 4 * yi - 2 + ky;
    if (zo >= 0 && zo < Do && yo >= 0 && yo < Ho) {
      float d[8];
      k8_row(dy + (((size_t)b * Do + zo) * Ho + yo) * Wo, xi, Wo, d);
      const float xv = x[(((size_t)b * D + zi) * H + yi) * W + xi];
#pragma unroll
      for (int kx = 0; kx < 8; ++kx) acc[kx] = fmaf(xv, d[kx], acc[kx]);
    }
    if (++run == 64) {   // FP32 runs of 64 terms, FP64 across runs
      run = 0;
#pragma unroll
      for (int kx = 0; kx < 8; ++kx) {
        tot[kx] += (double)acc[kx];
        acc[kx] = 0.f;
      }
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int kx = 0; kx < 8; ++kx) {
    double v = tot[kx] + (double)acc[kx];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) sm[wave][kx] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8)
    ws[((size_t)blockIdx.x * 64 + blockIdx.y) * 8 + threadIdx.x] =
        (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

__global__ void deconv_k8s4_dw_reduce_kernel(const double* __restrict__ ws, float* __restrict__ dw, int nchunk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 512) return;
  double s = 0.0;
  for (int c = 0; c < nchunk; ++c) s += ws[(size_t)c * 512 + i];
  dw[i] = (float)s;
}

// nn.AvgPool2d(k, stride k) backward (the SPP branches, backbones/PSMNet.py:43-58): every input element of a pooled window gets
// dy / k^2, the rows / columns the pooling dropped get zero.
__global__ __launch_bounds__(256) void avgpool2d_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long long total,
                                                            int H, int W, int k, int Ho, int Wo) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W), y = (int)((i / W) % H);
  const long long bc = i / ((long long)W * H);
  const int yo = y / k, xo = x / k;
  dx[i] = (yo < Ho && xo < Wo) ? dy[(bc * Ho + yo) * Wo + xo] / (float)(k * k) : 0.f;
}

// Bilinear (align_corners) adjoint for SMALL inputs (the SPP branches: 1x2 .. 8x16 maps blown up to the feature size): one
// wave per input element gathers its whole footprint in parallel (the one-thread-per-element form above walks up to the
// whole output image serially).
__global__ __launch_bounds__(64) void bilinear_ac_bwd_small_kernel(const float* __restrict__ t, float* __restrict__ dx, int C, int Hi,
                                                                   int Wi, int Ho, int Wo, float sh, float sw) {
  const int xl = blockIdx.x % Wi, yl = blockIdx.x / Wi, c = blockIdx.y, b = blockIdx.z;
  auto range = [](int i, float scale, int out, int& lo, int& hi) {
    if (scale <= 0.f) {
      lo = 0;
      hi = i == 0 ? out - 1 : -1;
      return;
    }
    lo = (int)floorf((float)(i - 1) / scale) - 1;
    hi = (int)ceilf((float)(i + 1) / scale) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > out - 1 ? out - 1 : hi;
  };
  int ylo, yhi, xlo, xhi;
  range(yl, sh, Ho, ylo, yhi);
  range(xl, sw, Wo, xlo, xhi);
  const float* tb = t + ((size_t)b * C + c) * Ho * Wo;
  const int nx = xhi - xlo + 1, total = (yhi - ylo + 1) * (nx > 0 ? nx : 0);
  float acc = 0.f;
  for (int i = threadIdx.x; i < total; i += 64) {
    const int yo = ylo + i / nx, xo = xlo + i % nx;
    const Lerp ly = lerp_setup(yo, Hi, sh), lx = lerp_setup(xo, Wi, sw);
    const float wy = (ly.i0 == yl ? ly.w0 : 0.f) + (ly.i1 == yl ? ly.w1 : 0.f);
    const float wx = (lx.i0 == xl ? lx.w0 : 0.f) + (lx.i1 == xl ? lx.w1 : 0.f);
    acc = fmaf(tb[(size_t)yo * Wo + xo], wy * wx, acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if (threadIdx.x == 0) dx[(((size_t)b * C + c) * Hi + yl) * Wi + xl] = acc;
}

// Adjoint of F.interpolate(mode='bilinear', align_corners=False) * mult (the refinement's disparity up-sampling,
// disp_refinement/utils/edge_aware.py:49-50): one wave per input element gathers its footprint; the weights are recomputed
// with the forward kernel's arithmetic (src = max(scale * (dst + 0.5) - 0.5, 0), conv2d.hip::bilinear_hp_kernel).
__global__ __launch_bounds__(64) void bilinear_hp_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int C, int Hi, int Wi,
                                                             int Ho, int Wo, float sh, float sw, float mult) {
  const int xl = blockIdx.x % Wi, yl = blockIdx.x / Wi, c = blockIdx.y, b = blockIdx.z;
  auto range = [](int i, float scale, int out, int& lo, int& hi) {
    lo = (int)floorf((float)(i - 1) / scale) - 2;
    hi = (int)ceilf((float)(i + 2) / scale) + 2;
    lo = lo < 0 ? 0 : lo;
    hi = hi > out - 1 ? out - 1 : hi;
  };
  auto weight = [](int o, int i, int in, float scale) {   // weight of input index i in output index o
    const float s = fmaxf(scale * ((float)o + 0.5f) - 0.5f, 0.f);
    int i0 = (int)s;
    i0 = i0 > in - 1 ? in - 1 : i0;
    const int i1 = i0 + (i0 < in - 1 ? 1 : 0);
    float l = s - (float)i0;
    l = fminf(fmaxf(l, 0.f), 1.f);
    return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
  };
  int ylo, yhi, xlo, xhi;
  range(yl, sh, Ho, ylo, yhi);
  range(xl, sw, Wo, xlo, xhi);
  const float* tb = dy + ((size_t)b * C + c) * Ho * Wo;
  const int nx = xhi - xlo + 1, total = (yhi - ylo + 1) * (nx > 0 ? nx : 0);
  float acc = 0.f;
  for (int i = threadIdx.x; i < total; i += 64) {
    const int yo = ylo + i / nx, xo = xlo + i % nx;
    const float wy = weight(yo, yl, Hi, sh);
    if (wy == 0.f) continue;
    acc = fmaf(tb[(size_t)yo * Wo + xo], wy * weight(xo, xl, Wi, sw), acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if (threadIdx.x == 0) dx[(((size_t)b * C + c) * Hi + yl) * Wi + xl] = acc * mult;
}

static int fill_idx(const int* host, int D, DispIdx& idx) {
  if (!host || D <= 0 || D > DMB_MAX_DISP_SAMPLES) return fail(DMB_EINVAL, "disparity sample count out of range");
  for (int k = 0; k < D; ++k) idx.d[k] = host[k];
  return DMB_OK;
}
static int fill_val(const float* host, int D, DispVal& dv) {
  if (!host || D <= 0 || D > DMB_MAX_DISP_SAMPLES) return fail(DMB_EINVAL, "disparity sample count out of range");
  for (int k = 0; k < D; ++k) dv.v[k] = host[k];
  return DMB_OK;
}

template <bool DIF>
static int launch_volume_bwd(const float* dvol, float* dL, float* dR, int B, int C, int H, int W, int D, const int* disp_idx_host,
                             void* stream) {
  if (!dvol || !dL || !dR || B <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "volume_bwd: bad argument");
  if ((long long)B * C > 65535 || 2LL * H > 65535) return fail(DMB_EUNSUPPORTED, "volume_bwd: grid too large");
  DispIdx idx;
  if (int e = fill_idx(disp_idx_host, D, idx)) return e;
  hipLaunchKernelGGL(volume_bwd_kernel<DIF>, dim3(cdiv(W, 256), 2 * H, B * C), dim3(256), 0, (hipStream_t)stream, dvol, dL, dR, C, H,
                     W, D, idx);
  return launch_status("volume_bwd launch failed");
}

}  // namespace dmb

using namespace dmb;

extern "C" int dmb_cat_fms_bwd_f32(const float* dvol, float* dL, float* dR, int B, int C, int H, int W, int D,
                                   const int* disp_idx_host, void* stream) {
  return launch_volume_bwd<false>(dvol, dL, dR, B, C, H, W, D, disp_idx_host, stream);
}
extern "C" int dmb_dif_fms_bwd_f32(const float* dvol, float* dL, float* dR, int B, int C, int H, int W, int D,
                                   const int* disp_idx_host, void* stream) {
  return launch_volume_bwd<true>(dvol, dL, dR, B, C, H, W, D, disp_idx_host, stream);
}

extern "C" int dmb_cat_first_wgrad_maps_f32(const float* dc, float* maps_left, float* maps_right, int B, int Co, int D, int H, int W,
                                            void* stream) {
  if (!dc || !maps_left || !maps_right || B <= 0 || Co <= 0 || D <= 0 || H <= 0 || W <= 0)
    return fail(DMB_EINVAL, "cat_first_wgrad_maps: bad argument");
  if (B > 65535 || Co > 65535) return fail(DMB_EUNSUPPORTED, "cat_first_wgrad_maps: grid too large");
  hipLaunchKernelGGL(cat_wgrad_maps_kernel, dim3(cdiv(H * W, 256), Co, B), dim3(256), 0, (hipStream_t)stream, dc, maps_left, maps_right,
                     Co, D, H, W);
  return launch_status("cat_first_wgrad_maps launch failed");
}

extern "C" int dmb_soft_argmin_bwd_f32(const float* cost, const float* disp, const float* grad_disp, float* grad_cost, int B,
                                       int D, int H, int W, float alpha, const float* disp_sample_host, void* stream) {
  if (!cost || !disp || !grad_disp || !grad_cost || B <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "soft_argmin_bwd: bad argument");
  DispVal dv;
  if (int e = fill_val(disp_sample_host, D, dv)) return e;
  const long long HW = (long long)H * W;
  hipLaunchKernelGGL(soft_argmin_bwd_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, cost, disp,
                     grad_disp, grad_cost, D, HW, alpha, dv);
  return launch_status("soft_argmin_bwd launch failed");
}

extern "C" int dmb_trilinear_ac_soft_argmin_bwd_f32(const float* x, const float* disp, const float* grad_disp, const float* grad_y,
                                                    float* scratch, float* grad_x, int B, int Di, int Hi, int Wi, int Do, int Ho,
                                                    int Wo, float alpha, const float* disp_sample_host, void* stream) {
  if (!x || !disp || !grad_disp || !scratch || !grad_x || B <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0)
    return fail(DMB_EINVAL, "trilinear_soft_argmin_bwd: bad argument");
  if (Ho > 65535 || (long long)Di * Hi > 65535 || B > 65535) return fail(DMB_EUNSUPPORTED, "trilinear_soft_argmin_bwd: grid too large");
  DispVal dv;
  if (int e = fill_val(disp_sample_host, Do, dv)) return e;
  hipStream_t st = (hipStream_t)stream;
  const float sd = ac_scale(Di, Do), sh = ac_scale(Hi, Ho), sw = ac_scale(Wi, Wo);
  hipLaunchKernelGGL(upsample_regress_bwd_z_kernel, dim3(cdiv(Wo, 256), Ho, B), dim3(256), 0, st, x, disp, grad_disp, scratch, Di, Hi,
                     Wi, Do, Ho, Wo, sd, sh, sw, alpha, dv);
  if (grad_y)   // a loss on the volume as well: its z fold joins the regression's before the (y, x) contraction
    hipLaunchKernelGGL(upsample_bwd_z_kernel, dim3(cdiv(Wo, 256), Ho, B), dim3(256), 0, st, grad_y, scratch, Di, Do, Ho, Wo, sd, 1);
  launch_upsample_bwd_hw(scratch, grad_x, B, Di, Hi, Wi, Ho, Wo, sh, sw, st);
  return launch_status("trilinear_soft_argmin_bwd launch failed");
}

extern "C" long long dmb_deconv3d_k8s4_bwd_workspace_doubles(void) { return (long long)K8_CHUNKS * 512; }

extern "C" int dmb_deconv3d_k8s4_c1_bwd_f32(const float* x, const float* w, const float* dy, float* dx, float* dw,
                                            double* workspace, int B, int D, int H, int W, void* stream) {
  if (!x || !w || !dy || (!dx && !dw) || (dw && !workspace) || B <= 0 || D <= 0 || H <= 0 || W <= 0)
    return fail(DMB_EINVAL, "deconv_k8s4_bwd: bad argument");
  if ((long long)D * H > 65535 || B > 65535) return fail(DMB_EUNSUPPORTED, "deconv_k8s4_bwd: grid too large");
  hipStream_t st = (hipStream_t)stream;
  if (dx) hipLaunchKernelGGL(deconv_k8s4_dx_kernel, dim3(cdiv(W, 256), D * H, B), dim3(256), 0, st, dy, w, dx, D, H, W);
  if (dw) {
    hipLaunchKernelGGL(deconv_k8s4_dw_kernel, dim3(K8_CHUNKS, 64), dim3(256), 0, st, x, dy, workspace, B, D, H, W);
    hipLaunchKernelGGL(deconv_k8s4_dw_reduce_kernel, dim3(2), dim3(256), 0, st, workspace, dw, K8_CHUNKS);
  }
  return launch_status("deconv_k8s4_bwd launch failed");
}

extern "C" int dmb_trilinear_ac_bwd_f32(const float* grad_y, float* scratch, float* grad_x, int B, int Di, int Hi, int Wi, int Do,
                                        int Ho, int Wo, void* stream) {
  if (!grad_y || !scratch || !grad_x || B <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0)
    return fail(DMB_EINVAL, "trilinear_bwd: bad argument");
  if (Ho > 65535 || (long long)Di * Hi > 65535 || B > 65535) return fail(DMB_EUNSUPPORTED, "trilinear_bwd: grid too large");
  hipStream_t st = (hipStream_t)stream;
  const float sd = ac_scale(Di, Do), sh = ac_scale(Hi, Ho), sw = ac_scale(Wi, Wo);
  hipLaunchKernelGGL(upsample_bwd_z_kernel, dim3(cdiv(Wo, 256), Ho, B), dim3(256), 0, st, grad_y, scratch, Di, Do, Ho, Wo, sd, 0);
  launch_upsample_bwd_hw(scratch, grad_x, B, Di, Hi, Wi, Ho, Wo, sh, sw, st);
  return launch_status("trilinear_bwd launch failed");
}

extern "C" int dmb_avgpool2d_bwd_f32(const float* grad_y, float* grad_x, int B, int C, int H, int W, int k, void* stream) {
  if (!grad_y || !grad_x || B <= 0 || C <= 0 || k <= 0 || H < k || W < k) return fail(DMB_EINVAL, "avgpool2d_bwd: bad argument");
  const long long total = (long long)B * C * H * W;
  hipLaunchKernelGGL(avgpool2d_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_y, grad_x, total,
                     H, W, k, H / k, W / k);
  return launch_status("avgpool2d_bwd launch failed");
}

extern "C" int dmb_bilinear_ac_bwd_f32(const float* grad_y, float* grad_x, int B, int C, int Hi, int Wi, int Ho, int Wo, void* stream) {
  if (!grad_y || !grad_x || B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return fail(DMB_EINVAL, "bilinear_bwd: bad argument");
  if ((long long)C * Hi > 65535 || B > 65535 || C > 65535) return fail(DMB_EUNSUPPORTED, "bilinear_bwd: grid too large");
  if ((long long)Hi * Wi * 64 <= (long long)Ho * Wo) {   // footprints of >= 64 output pixels: one wave per input element
    hipLaunchKernelGGL(bilinear_ac_bwd_small_kernel, dim3(Hi * Wi, C, B), dim3(64), 0, (hipStream_t)stream, grad_y, grad_x, C, Hi, Wi, Ho,
                       Wo, ac_scale(Hi, Ho), ac_scale(Wi, Wo));
    return launch_status("bilinear_bwd launch failed");
  }
  // the (y, x) contraction of the up-sampling backward, one "plane" per channel
  launch_upsample_bwd_hw(grad_y, grad_x, B, C, Hi, Wi, Ho, Wo, ac_scale(Hi, Ho), ac_scale(Wi, Wo), (hipStream_t)stream);
  return launch_status("bilinear_bwd launch failed");
}

extern "C" int dmb_bilinear_scale_bwd_f32(const float* grad_y, float* grad_x, int B, int C, int Hi, int Wi, int Ho, int Wo, float mult,
                                          void* stream) {
  if (!grad_y || !grad_x || B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return fail(DMB_EINVAL, "bilinear_scale_bwd: bad argument");
  if (C > 65535 || B > 65535) return fail(DMB_EUNSUPPORTED, "bilinear_scale_bwd: grid too large");
  hipLaunchKernelGGL(bilinear_hp_bwd_kernel, dim3(Hi * Wi, C, B), dim3(64), 0, (hipStream_t)stream, grad_y, grad_x, C, Hi, Wi, Ho, Wo,
                     (float)Hi / (float)Ho, (float)Wi / (float)Wo, mult);
  return launch_status("bilinear_scale_bwd launch failed");
}
