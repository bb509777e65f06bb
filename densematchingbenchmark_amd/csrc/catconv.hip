// First 3-D convolution of the aggregators applied to a CONCATENATION volume that is never materialised.
//
// The volume cat_fms builds (cost_processors/utils/cat_fms.py:7-48, unit disparity step: d_k = k) is
//     V[c,      z, y, x] = L[c, y, x]       * [x >= z]
//     V[C + c,  z, y, x] = R[c, y, x - z]   * [x >= z]
// i.e. its left half does not depend on z and its right half depends on (z, x) through x - z only.  A 3x3x3 convolution
// (aggregators/PSMNet.py:31-33, AcfNet.py:28-31: dres0[0], 2C -> Co, padding 1) of such a volume collapses into 2-D maps:
//     out[co, z, y, x] = sum_{dz in Z(z)} ( F_{dz, m}[co, y, x]  +  H_dz[co, y, x - (z + dz - 1)] )
//   F_{dz, m}[co, y, x] = sum_{ci, dy, dx >= m} w[co, ci,     dz, dy, dx] * L0[ci, y + dy - 1, x + dx - 1]     (3x3 conv of L)
//   H_dz   [co, y, n]   = sum_{ci, dy, dx}      w[co, C + ci, dz, dy, dx] * R0[ci, y + dy - 1, n + dx - 1]     (3x3 conv of R)
// with L0 / R0 zero outside the image (R0[n < 0] = 0 IS the [x >= z] mask of the right half), Z(z) the taps whose input
// plane z + dz - 1 exists, and m = max(0, dz - (x - z)) the first dx tap the left half's mask lets through (m = 0 for
// x - z >= 2, i.e. everywhere but four columns per row).  At x = W - 1 the dx = 2 tap of the right half falls outside the
// image and H is replaced by the map computed without it.  The 2-D convolutions run on conv2d_kernel (conv2d.hip); this
// file holds what is left: zero-filled window copies, the 2-D summation of the per-dz maps, and the one HBM-bound pass
// that writes the layer's output.  Same FP32 products as the 3-D form; the sums are grouped per dz (each an fma chain
// over ci, dy, dx) instead of one chain over all taps.  The raw volume (1.6 GB at the BASELINE size) is neither written
// nor read, and 2/3 of the layer's multiplications disappear.
#include "dmb_common.h"

namespace dmb {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// dst[r, j] = src[r, j + xs] if 0 <= j + xs < W else 0,  j in [0, Wd)
__global__ __launch_bounds__(256) void copy_window_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          long long rows, int W, int Wd, int xs) {
  const long long total = rows * Wd;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / Wd;
    const int j = (int)(i - r * Wd), g = j + xs;
    dst[i] = (g >= 0 && g < W) ? src[r * W + g] : 0.f;
  }
}

// t[r, x0 .. pitch) = 0 for every row r: the padding columns of a tensor whose rows are padded to a 16-byte multiple
__global__ __launch_bounds__(256) void zero_columns_kernel(float* __restrict__ t, long long rows, int pitch, int x0) {
  const int n = pitch - x0;
  const long long total = rows * n;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / n;
    t[r * pitch + x0 + (int)(i - r * n)] = 0.f;
  }
}

__device__ __forceinline__ bool z_tap_valid(int v, int dz) { return v == 0 ? dz >= 1 : (v == 2 ? dz <= 1 : true); }

// 2-D summations over dz, one workgroup per (b, co, y) row.  Z(z): z = 0 -> taps dz in {1, 2}; interior -> {0, 1, 2};
// z = D - 1 -> {0, 1}.
//   FA [B, CA, H, W]       channel dz*Co + co            F_{dz, 0}
//   FB [B, CB, H, Wc]      channel (m-1)*CB/2+dz*Co+co   F_{dz, m}, m = 1, 2, columns x in [0, Wc)
//   HC [B, CA, H, W + 4]   channel dz*Co + co            H_dz at n = j - 4
//   HD [B, CA, H, Wc]      channel dz*Co + co            H_dz without the dx = 2 tap, at n = j + W - Wc
// ->
//   FM [B, Co, H, W]               interior planes, x - z >= 2:  sum_dz F_{dz, 0}[x]
//   GM [B, Co, H, W + 4]           interior planes, at n = j - 4:  sum_dz H_dz[n - dz + 1]
//   BAND [B, Co, H, D, 4]          the left half next to x == z, plane z, x = z + u, u = -2 .. 1:
//                                  sum_{dz in Z(z), dz - u <= 2} F_{dz, max(0, dz - u)}[x]   (0 where x is outside the image)
//   GB [B, Co, H, D]               the right half at x = W - 1 of plane z:  sum_{dz in Z(z)} HD_dz[W - 1 - (z + dz - 1)]
__global__ __launch_bounds__(256) void catconv_finalize_kernel(const float* __restrict__ FA, const float* __restrict__ FB,
                                                               const float* __restrict__ HC, const float* __restrict__ HD,
                                                               float* __restrict__ FM, float* __restrict__ BAND,
                                                               float* __restrict__ GM, float* __restrict__ GB, int B, int Co,
                                                               int CA, int CB, int D, int H, int W, int Wc) {
  const int Wg = W + 4;
  int r = blockIdx.x;
  const int y = r % H;
  r /= H;
  const int co = r % Co, b = r / Co;
  const float* fa[3];
  const float* fb[2][3];
  const float* hc[3];
  const float* hd[3];
#pragma unroll
  for (int dz = 0; dz < 3; ++dz) {
    fa[dz] = FA + (((size_t)b * CA + dz * Co + co) * H + y) * W;
    hc[dz] = HC + (((size_t)b * CA + dz * Co + co) * H + y) * Wg;
    hd[dz] = HD + (((size_t)b * CA + dz * Co + co) * H + y) * Wc;
#pragma unroll
    for (int m = 0; m < 2; ++m) fb[m][dz] = FB + (((size_t)b * CB + m * (CB / 2) + dz * Co + co) * H + y) * Wc;
  }
  const size_t row = ((size_t)b * Co + co) * H + y;
  const int t = threadIdx.x;
  // the four families are dealt to disjoint thread ranges (W / 4 + (W + 4) + D <= 256 + ...: loops cover any width)
  for (int x4 = t; x4 < W / 4; x4 += 256) {
    const float4 a0 = *reinterpret_cast<const float4*>(fa[0] + x4 * 4), a1 = *reinterpret_cast<const float4*>(fa[1] + x4 * 4),
                 a2 = *reinterpret_cast<const float4*>(fa[2] + x4 * 4);
    *reinterpret_cast<float4*>(FM + row * W + x4 * 4) =
        make_float4(a0.x + a1.x + a2.x, a0.y + a1.y + a2.y, a0.z + a1.z + a2.z, a0.w + a1.w + a2.w);
  }
  for (int j = 255 - t; j < Wg; j += 256) {   // (reversed: these run on the lanes the loop above leaves idle)
    float s = 0.f;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      const int jj = j - dz + 1;   // H_dz at n - dz + 1
      if (jj >= 0 && jj < Wg) s += hc[dz][jj];
    }
    GM[row * Wg + j] = s;
  }
  for (int q = t; q < 4 * D; q += 256) {
    const int z = q >> 2, u = (q & 3) - 2, x = z + u;
    float s = 0.f;
    if (x >= 0 && x < Wc) {
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        const int m = dz - u > 0 ? dz - u : 0;
        if (!z_tap_valid(z == 0 ? 0 : (z == D - 1 ? 2 : 1), dz) || m > 2) continue;
        s += m == 0 ? fa[dz][x] : fb[m - 1][dz][x];
      }
    }
    BAND[row * (4 * D) + q] = s;
  }
  for (int z = 255 - t; z < D; z += 256) {
    const int v = z == 0 ? 0 : (z == D - 1 ? 2 : 1);
    float s = 0.f;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      const int jj = Wc - 1 - (z + dz - 1);   // HD column of n = W - 1 - z'
      if (z_tap_valid(v, dz) && jj >= 0 && jj < Wc) s += hd[dz][jj];
    }
    GB[row * D + z] = s;
  }
}

// The layer's output, one pass: a thread owns 4 consecutive x of one (b, co, y) row and walks z.
//   out[b, co, z, y, x] = act(scale[co] * (f + g) + shift[co])
//   f = sum_{dz in Z(z)} F_{dz,0}[x] for x - z >= 2 (FM on interior planes), BAND[z][x - z + 2] for -2 <= x - z < 2, else 0
//   g = sum_{dz in Z(z)} H_dz[x - z - dz + 1] (GM on interior planes; exactly 0 for x - z <= -3), GB[z] at x = W - 1
// Along z the four g values of a thread slide by one column per plane; the two border planes are summed from the per-dz
// maps on the fly.  Loads and stores retire through ONE in-order counter on this chip (a load issued after a store waits
// for that store's acknowledgement), so everything a block of ZB planes reads -- its g window of ZB + 3 columns, its band
// words, its x = W - 1 values -- is loaded one block AHEAD of the stores.
// NEAR: this thread's columns can be reached by a mask band (x < XS): f is selected per element.
template <bool NEAR>
__device__ __forceinline__ void catconv_combine_body(const float* __restrict__ FA, const float* __restrict__ HC,
                                                     const float* __restrict__ FM, const float* __restrict__ BAND,
                                                     const float* __restrict__ GM, const float* __restrict__ GB,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     float* __restrict__ out, int Co, int CA, int D, int H, int W, int relu,
                                                     long long rowid, int x, int zlo, int zhi) {
  const int Wg = W + 4;
  long long r = rowid;
  const int y = (int)(r % H); r /= H;
  const int co = (int)(r % Co);
  const int b = (int)(r / Co);
  const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
  const float lo = relu ? 0.f : -__builtin_inff();
  const size_t HW = (size_t)H * W, row = ((size_t)b * Co + co) * H + y;
  const float* gb = GB + row * D;
  const float* gm_row = GM + row * Wg + 4;   // indexed by n >= -4; exactly 0 at n <= -3
  const float4* band = reinterpret_cast<const float4*>(BAND + row * (4 * D));
  float* op = out + (((size_t)b * Co + co) * D) * HW + (size_t)y * W + x;
  const bool border = !NEAR && x + 4 == W;
  auto fa4 = [&](int dz) { return *reinterpret_cast<const float4*>(FA + (((size_t)b * CA + dz * Co + co) * H + y) * W + x); };
  auto hc1 = [&](int dz, int n) {   // H_dz at n, 0 left of the stored range (it IS 0 there)
    return n >= -4 ? HC[(((size_t)b * CA + dz * Co + co) * H + y) * Wg + 4 + n] : 0.f;
  };
  // bd: the plane's band word, gbz: the right half at x = W - 1 (one address per row and plane each: broadcasts)
  auto plane = [&](int z, const float4& fvv, const float (&g)[4], const float4& bd, float gbz) {
    float f[4] = {fvv.x, fvv.y, fvv.z, fvv.w};
    if (NEAR) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int u = x + e - z;
        const float t = u == -2 ? bd.x : (u == -1 ? bd.y : (u == 0 ? bd.z : bd.w));
        f[e] = u >= 2 ? f[e] : (u >= -2 ? t : 0.f);
      }
    }
    const float g3 = border ? gbz : g[3];
    float4 o;
    o.x = fmaxf(fmaf(f[0] + g[0], sc, sh), lo);
    o.y = fmaxf(fmaf(f[1] + g[1], sc, sh), lo);
    o.z = fmaxf(fmaf(f[2] + g[2], sc, sh), lo);
    o.w = fmaxf(fmaf(f[3] + g3, sc, sh), lo);
    // streaming store: the 0.8 GB output is far larger than the caches it would otherwise evict the maps from
    __builtin_nontemporal_store(f32x4_t{o.x, o.y, o.z, o.w}, reinterpret_cast<f32x4_t*>(op + (size_t)z * HW));
  };
  if (zlo == 0) {   // plane 0: taps dz = 1, 2 (input planes 0 and 1)
    const float4 a1 = fa4(1), a2 = fa4(2);
    float g[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = hc1(1, x + e) + hc1(2, x + e - 1);
    plane(0, make_float4(a1.x + a2.x, a1.y + a2.y, a1.z + a2.z, a1.w + a2.w), g, band[0], gb[0]);
  }
  const float4 fm = *reinterpret_cast<const float4*>(FM + row * W + x);
  auto gml = [&](int n) { return gm_row[n > -4 ? n : -4]; };
  constexpr int ZB = 4;
  float win[ZB + 3], gbw[ZB];   // block starting at plane z0: win[i] = G[x - z0 + 3 - i]; plane z0 + k uses g[e] = win[3 - e + k]
  float4 bnd[ZB];
  auto load_block = [&](int z0, float (&w)[ZB + 3], float4 (&bq)[ZB], float (&gq)[ZB]) {
#pragma unroll
    for (int i = 0; i < ZB + 3; ++i) w[i] = gml(x - z0 + 3 - i);
#pragma unroll
    for (int k = 0; k < ZB; ++k) {
      const int z = z0 + k < D ? z0 + k : D - 1;
      // the band values matter for x + e - z in [-2, 2) only: lanes further from the diagonal skip the load
      if (NEAR) bq[k] = (x - z >= -5 && x - z <= 1) ? band[z] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (!NEAR) gq[k] = gb[z];
    }
  };
  const int zb = zlo > 1 ? zlo : 1, ze = zhi < D - 1 ? zhi : D - 1;   // interior planes of this thread's range
  load_block(zb, win, bnd, gbw);
  for (int z0 = zb; z0 < ze; z0 += ZB) {
    float nwin[ZB + 3], ngbw[ZB];
    float4 nbnd[ZB];
    if (z0 + ZB < ze) load_block(z0 + ZB, nwin, nbnd, ngbw);
#pragma unroll
    for (int k = 0; k < ZB; ++k) {
      if (z0 + k < ze) {
        const float g[4] = {win[3 + k], win[2 + k], win[1 + k], win[k]};
        plane(z0 + k, fm, g, bnd[k], gbw[k]);
      }
    }
#pragma unroll
    for (int i = 0; i < ZB + 3; ++i) win[i] = nwin[i];
#pragma unroll
    for (int k = 0; k < ZB; ++k) {
      bnd[k] = nbnd[k];
      gbw[k] = ngbw[k];
    }
  }
  if (zhi == D) {   // plane D - 1: taps dz = 0, 1 (input planes D - 2 and D - 1)
    const float4 a0 = fa4(0), a1 = fa4(1);
    const int u0 = x - (D - 1);
    float g[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = hc1(0, u0 + e + 1) + hc1(1, u0 + e);
    plane(D - 1, make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w), g, band[D - 1], gb[D - 1]);
  }
}

// Two launches: NEAR = the columns x < XS (every column a mask band can reach, rounded up to 64 bytes), the other one the
// rest of the rows with no per-element selection at all.  (One launch whose waves split the same way, and one launch with
// the selection on every lane, were both slower: 0.26 / 0.44 ms against 0.11 + 0.11 ms at the BASELINE size.)
template <bool NEAR>
__global__ __launch_bounds__(256) void catconv_combine_kernel(const float* __restrict__ FA, const float* __restrict__ HC,
                                                              const float* __restrict__ FM, const float* __restrict__ BAND,
                                                              const float* __restrict__ GM, const float* __restrict__ GB,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float* __restrict__ out, long long rows, int Co, int CA, int D,
                                                              int H, int W, int XS, int relu) {
  // NEAR: the few columns next to the band give one wave of work per 4 rows when a thread walks all D planes -- a chain of
  // dependent map loads at half occupancy; the planes are independent, so each (row, column quad) is split over ZSPLIT threads.
  constexpr int ZSPLIT = NEAR ? 4 : 1;
  const int W4 = NEAR ? XS >> 2 : (W - XS) >> 2;
  long long idx = blockIdx.x * 256LL + threadIdx.x;
  if (idx >= rows * W4 * ZSPLIT) return;
  const int zc = (int)(idx / (rows * W4));
  idx -= zc * (rows * W4);
  const long long rowid = idx / W4;
  const int dz = D / ZSPLIT;   // D is a multiple of 4
  catconv_combine_body<NEAR>(FA, HC, FM, BAND, GM, GB, scale, shift, out, Co, CA, D, H, W, relu, rowid,
                             (int)(idx - rowid * W4) * 4 + (NEAR ? 0 : XS), zc * dz, (zc + 1) * dz);
}

}  // namespace dmb

using namespace dmb;

extern "C" int dmb_copy_window_f32(const float* src, float* dst, long long rows, int W, int Wd, int xs, void* stream) {
  if (!src || !dst || rows <= 0 || W <= 0 || Wd <= 0) return fail(DMB_EINVAL, "copy_window: bad argument");
  const long long total = rows * Wd;
  const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(copy_window_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, rows, W, Wd, xs);
  return launch_status("copy_window launch failed");
}

extern "C" int dmb_zero_columns_f32(float* t, long long rows, int pitch, int x0, void* stream) {
  if (!t || rows <= 0 || pitch <= 0 || x0 < 0 || x0 > pitch) return fail(DMB_EINVAL, "zero_columns: bad argument");
  if (x0 == pitch) return DMB_OK;
  const long long total = rows * (pitch - x0);
  const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(zero_columns_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, rows, pitch, x0);
  return launch_status("zero_columns launch failed");
}

extern "C" int dmb_catconv_finalize_f32(const float* FA, const float* FB, const float* HC, const float* HD, float* FM,
                                        float* BAND, float* GM, float* GB, int B, int Co, int CA, int CB, int D, int H, int W,
                                        int Wc, void* stream) {
  if (!FA || !FB || !HC || !HD || !FM || !BAND || !GM || !GB || B <= 0 || Co <= 0 || D < 3 || H <= 0 || W <= 0 || (W & 3) ||
      CA < 3 * Co || CB < 6 * Co || (CB & 1) || Wc > W || Wc < D + 2 || (((uintptr_t)FA | (uintptr_t)FM) & 15))
    return fail(DMB_EINVAL, "catconv_finalize: bad argument");
  const long long rows = (long long)B * Co * H;
  if (rows > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "catconv_finalize: grid too large");
  hipLaunchKernelGGL(catconv_finalize_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, FA, FB, HC, HD, FM, BAND,
                     GM, GB, B, Co, CA, CB, D, H, W, Wc);
  return launch_status("catconv_finalize launch failed");
}

extern "C" int dmb_catconv_combine_f32(const float* FA, const float* HC, const float* FM, const float* BAND, const float* GM,
                                       const float* GB, const float* scale, const float* shift, float* out, int B, int Co,
                                       int CA, int D, int H, int W, int relu, void* stream) {
  if (!FA || !HC || !FM || !BAND || !GM || !GB || !out || B <= 0 || Co <= 0 || CA < 3 * Co || D < 3 || (D & 3) || H <= 0 ||
      W <= 0 || (W & 3) || W < D + 8 || (((uintptr_t)FA | (uintptr_t)FM | (uintptr_t)BAND | (uintptr_t)out) & 15))
    return fail(DMB_EINVAL, "catconv_combine: bad argument (D, W multiples of 4, W >= D + 8, tensors 16-byte aligned)");
  int XS = (D + 4 + 15) / 16 * 16;   // first column no mask band reaches, rounded up to 64 bytes
  if (XS > W - 4) XS = W - 4;
  const long long rows = (long long)B * Co * H, tn = rows * (XS / 4) * 4, tf = rows * ((W - XS) / 4);   // near: 4 plane ranges per quad
  if ((tf + 255) / 256 > 0x7fffffffLL || (tn + 255) / 256 > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "catconv_combine: grid too large");
  hipLaunchKernelGGL(catconv_combine_kernel<false>, dim3((unsigned)((tf + 255) / 256)), dim3(256), 0, (hipStream_t)stream, FA,
                     HC, FM, BAND, GM, GB, scale, shift, out, rows, Co, CA, D, H, W, XS, relu);
  hipLaunchKernelGGL(catconv_combine_kernel<true>, dim3((unsigned)((tn + 255) / 256)), dim3(256), 0, (hipStream_t)stream, FA,
                     HC, FM, BAND, GM, GB, scale, shift, out, rows, Co, CA, D, H, W, XS, relu);
  return launch_status("catconv_combine launch failed");
}
