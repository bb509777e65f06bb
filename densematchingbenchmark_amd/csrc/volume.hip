// Cost-volume builders: cat_fms, dif_fms (bit-exact copies / one subtract) and the GwcNet
// group-wise correlation volume.  HBM-write-bound kernels: one pass over the output, features
// staged through LDS so the shifted right-feature reads never touch HBM twice.
//
// Reference semantics: dmb/modeling/stereo/cost_processors/utils/cat_fms.py:7-48,
// dif_fms.py:7-46 (see include/dmb_hip.h for the formulas).
#include "dmb_common.h"

namespace dmb {

enum { MODE_CAT = 0, MODE_DIF = 1 };

// One workgroup = 256 threads x VEC consecutive elements of one (b, c) feature plane, all D disparity
// samples.  The right-feature window [i0 - dpos, i0 + CH + dneg) sits in LDS; every thread keeps its
// left-feature elements in registers and streams D output rows (coalesced 16-B stores along W, the
// disparity axis is the slow loop so consecutive stores of one thread are H*W*4 bytes apart while the
// wave as a whole writes 1 KiB contiguous per instruction).
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void volume_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                     float* __restrict__ out, int C, int H, int W, int D,
                                                     DispIdx idx, int dpos, int dneg, int out_channels,
                                                     int och_off) {
  extern __shared__ float win[];
  constexpr int CH = 256 * VEC;
  const int HW = H * W;
  const int c = blockIdx.y, b = blockIdx.z;
  const int i0 = blockIdx.x * CH;
  const float* Lp = L + ((size_t)b * C + c) * HW;
  const float* Rp = R + ((size_t)b * C + c) * HW;

  // stage the right-feature window (zeros outside the plane; those slots are never selected)
  const int wlen = CH + dpos + dneg;
  for (int t = threadIdx.x; t < wlen; t += 256) {
    const int g = i0 - dpos + t;
    win[t] = (g >= 0 && g < HW) ? Rp[g] : 0.f;
  }

  const int i = i0 + threadIdx.x * VEC;
  float lv[VEC];
  int xs[VEC];
  bool inb[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    inb[j] = (i + j) < HW;
    lv[j] = inb[j] ? Lp[i + j] : 0.f;
    xs[j] = (i + j) % W;
  }
  __syncthreads();

  const size_t DHW = (size_t)D * HW;
  float* oL;
  float* oR = nullptr;
  if (MODE == MODE_CAT) {
    oL = out + ((size_t)b * out_channels + och_off + c) * DHW + i;
    oR = out + ((size_t)b * out_channels + och_off + C + c) * DHW + i;
  } else {
    oL = out + ((size_t)b * out_channels + och_off + c) * DHW + i;
  }
  const int lbase = threadIdx.x * VEC + dpos;
  for (int k = 0; k < D; ++k) {
    const int d = idx.d[k];
    float a[VEC], r[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      // cat_fms.py:36-44: d > 0 keeps x >= d, d < 0 keeps x < W + d, d == 0 keeps everything
      const bool keep = (d > 0) ? (xs[j] >= d) : ((d < 0) ? (xs[j] < W + d) : true);
      const float rv = win[lbase + j - d];
      if (MODE == MODE_CAT) {
        a[j] = keep ? lv[j] : 0.f;
        r[j] = keep ? rv : 0.f;
      } else {
        a[j] = keep ? (lv[j] - rv) : 0.f;
      }
    }
    if constexpr (VEC == 4) {
      if (inb[0]) {  // W % 4 == 0 => the whole float4 is inside the plane
        *reinterpret_cast<float4*>(oL + (size_t)k * HW) = make_float4(a[0], a[1], a[2], a[3]);
        if (MODE == MODE_CAT) *reinterpret_cast<float4*>(oR + (size_t)k * HW) = make_float4(r[0], r[1], r[2], r[3]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        if (inb[j]) {
          oL[(size_t)k * HW + j] = a[j];
          if (MODE == MODE_CAT) oR[(size_t)k * HW + j] = r[j];
        }
    }
  }
}

// Group-wise correlation, VALU form: thread = one x position of one (b, g, y) row, left features of the
// group in registers, right-feature rows of the group in LDS, loop over disparity samples.
// out[b, g, k, y, x] = (1/CG) * sum_c L[c, y, x] * R[c, y, x - d_k]   (FP32 fma chain, ascending c).
template <int CG>
__global__ __launch_bounds__(256) void gwc_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                  float* __restrict__ out, int C, int G, int H, int W, int D,
                                                  DispIdx idx, int dpos, int dneg, int out_channels, int och_off) {
  extern __shared__ float win[];  // [CG][wlen]
  const int HW = H * W;
  const int nchunk = cdiv(HW, 256);
  const int chunk = blockIdx.x % nchunk;
  const int g = blockIdx.y, b = blockIdx.z;
  const int i0 = chunk * 256;
  const int wlen = 256 + dpos + dneg;
  const float* Lp = L + ((size_t)b * C + (size_t)g * CG) * HW;
  const float* Rp = R + ((size_t)b * C + (size_t)g * CG) * HW;
  for (int t = threadIdx.x; t < CG * wlen; t += 256) {
    const int c = t / wlen, o = t % wlen;
    const int gi = i0 - dpos + o;
    win[t] = (gi >= 0 && gi < HW) ? Rp[(size_t)c * HW + gi] : 0.f;
  }
  const int i = i0 + threadIdx.x;
  const bool inb = i < HW;
  const int x = i % W;
  float lv[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) lv[c] = inb ? Lp[(size_t)c * HW + i] : 0.f;
  __syncthreads();
  float* o = out + ((size_t)b * out_channels + och_off + g) * (size_t)D * HW + i;
  const int lbase = threadIdx.x + dpos;
  for (int k = 0; k < D; ++k) {
    const int d = idx.d[k];
    const bool keep = (d > 0) ? (x >= d) : ((d < 0) ? (x < W + d) : true);
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < CG; ++c) acc = fmaf(lv[c], win[c * wlen + lbase - d], acc);
    if (inb) o[(size_t)k * HW] = keep ? acc / (float)CG : 0.f;
  }
}

int gwc_mfma_dispatch(const float* L, const float* R, float* out, int B, int C, int G, int H, int W, int D,
                      const DispIdx& idx, int out_channels, int och_off, hipStream_t st);  // gwc_mfma.hip

static int disp_range(const int* idx, int D, DispIdx& out, int& dpos, int& dneg) {
  if (!idx || D <= 0 || D > DMB_MAX_DISP_SAMPLES) return fail(DMB_EINVAL, "disparity sample count out of range");
  dpos = 0;
  dneg = 0;
  for (int k = 0; k < D; ++k) {
    out.d[k] = idx[k];
    if (idx[k] > dpos) dpos = idx[k];
    if (-idx[k] > dneg) dneg = -idx[k];
  }
  return DMB_OK;
}

template <int MODE>
static int launch_volume(const float* L, const float* R, float* out, int B, int C, int H, int W, int D,
                         const int* disp_idx_host, int out_channels, int och_off, hipStream_t st) {
  if (!L || !R || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "volume: bad argument");
  DispIdx idx;
  int dpos, dneg;
  if (int e = disp_range(disp_idx_host, D, idx, dpos, dneg)) return e;
  const int HW = H * W;
  const bool vec = (W % 4 == 0);
  const int CH = vec ? 1024 : 256;
  const size_t lds = (size_t)(CH + dpos + dneg) * sizeof(float);
  if (lds > 64 * 1024) return fail(DMB_EUNSUPPORTED, "volume: disparity range too wide for the LDS window");
  dim3 grid(cdiv(HW, CH), C, B);
  if (vec)
    hipLaunchKernelGGL((volume_kernel<MODE, 4>), grid, dim3(256), lds, st, L, R, out, C, H, W, D, idx, dpos, dneg,
                       out_channels, och_off);
  else
    hipLaunchKernelGGL((volume_kernel<MODE, 1>), grid, dim3(256), lds, st, L, R, out, C, H, W, D, idx, dpos, dneg,
                       out_channels, och_off);
  return launch_status("volume kernel launch failed");
}

// correlation1d_cost (cost_processors/utils/correlation1d_cost.py:7-27): full-channel correlation along the epipolar
// line, the first max_disp of the sampler's 2*max_disp - 1 patch offsets, leaky ReLU:
//   out[b, j, y, x] = lrelu( sum_c L[b, c, y, x] * R[b, c, y, x + j - (D - 1)] ),  0 <= j < D, R = 0 left of the image
// (channel j <-> disparity D - 1 - j; no 1/C normalisation).  One thread = one pixel x 16 consecutive offsets; the
// channel sum is an FP32 fma chain in ascending c.  The 16 right-feature values of a channel are consecutive addresses.
__global__ __launch_bounds__(256) void correlation1d_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                            float* __restrict__ out, int C, int H, int W, int D, float slope) {
  const int HW = H * W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int j0 = blockIdx.y * 16, b = blockIdx.z;
  if (i >= HW) return;
  const int x = i % W;
  const float* Lp = L + (size_t)b * C * HW + i;
  const float* Rp = R + (size_t)b * C * HW + i;
  float acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0.f;
  for (int c = 0; c < C; ++c) {
    const float l = Lp[(size_t)c * HW];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int off = j0 + t - (D - 1);   // <= 0
      const float r = (x + off >= 0 && j0 + t < D) ? Rp[(size_t)c * HW + off] : 0.f;
      acc[t] = fmaf(l, r, acc[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < 16; ++t)
    if (j0 + t < D) out[((size_t)b * D + j0 + t) * HW + i] = acc[t] > 0.f ? acc[t] : acc[t] * slope;
}

}  // namespace dmb

using namespace dmb;

extern "C" int dmb_correlation1d_f32(const float* L, const float* R, float* out, int B, int C, int H, int W, int D,
                                     float negative_slope, void* stream) {
  if (!L || !R || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || D <= 0) return fail(DMB_EINVAL, "correlation1d: bad argument");
  if ((long long)C * H * W >= 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "correlation1d: feature map too large");
  hipLaunchKernelGGL(correlation1d_kernel, dim3(cdiv(H * W, 256), cdiv(D, 16), B), dim3(256), 0, (hipStream_t)stream, L, R, out,
                     C, H, W, D, negative_slope);
  return launch_status("correlation1d launch failed");
}

extern "C" int dmb_cat_fms_f32(const float* L, const float* R, float* out, int B, int C, int H, int W, int D,
                               const int* disp_idx_host, void* stream) {
  return launch_volume<MODE_CAT>(L, R, out, B, C, H, W, D, disp_idx_host, 2 * C, 0, (hipStream_t)stream);
}

extern "C" int dmb_cat_fms_into_f32(const float* L, const float* R, float* out, int B, int C, int H, int W, int D,
                                    const int* disp_idx_host, int out_channels, int out_ch_offset, void* stream) {
  if (out_ch_offset < 0 || out_ch_offset + 2 * C > out_channels) return fail(DMB_EINVAL, "cat_fms_into: channel window");
  return launch_volume<MODE_CAT>(L, R, out, B, C, H, W, D, disp_idx_host, out_channels, out_ch_offset,
                                 (hipStream_t)stream);
}

extern "C" int dmb_dif_fms_f32(const float* L, const float* R, float* out, int B, int C, int H, int W, int D,
                               const int* disp_idx_host, void* stream) {
  return launch_volume<MODE_DIF>(L, R, out, B, C, H, W, D, disp_idx_host, C, 0, (hipStream_t)stream);
}

extern "C" int dmb_gwc_fms_f32(const float* L, const float* R, float* out, int B, int C, int G, int H, int W, int D,
                               const int* disp_idx_host, int out_channels, int out_ch_offset, void* stream) {
  if (!L || !R || !out || B <= 0 || C <= 0 || G <= 0 || H <= 0 || W <= 0 || C % G != 0)
    return fail(DMB_EINVAL, "gwc: bad argument");
  if (out_ch_offset < 0 || out_ch_offset + G > out_channels) return fail(DMB_EINVAL, "gwc: channel window");
  DispIdx idx;
  int dpos, dneg;
  if (int e = disp_range(disp_idx_host, D, idx, dpos, dneg)) return e;
  const int CG = C / G;
  if (!DMB_OPT(1)) {  // matrix-core form whenever it applies (0 <= d_k <= 64, even channels per group)
    const int rc = gwc_mfma_dispatch(L, R, out, B, C, G, H, W, D, idx, out_channels, out_ch_offset, (hipStream_t)stream);
    if (rc != DMB_EUNSUPPORTED) return rc;
  }
  const size_t lds = (size_t)CG * (256 + dpos + dneg) * sizeof(float);
  if (lds > 64 * 1024) return fail(DMB_EUNSUPPORTED, "gwc: disparity range too wide for the LDS window");
  dim3 grid(cdiv(H * W, 256), G, B);
  hipStream_t st = (hipStream_t)stream;
#define DMB_GWC_CASE(N)                                                                                          \
  case N:                                                                                                        \
    hipLaunchKernelGGL((gwc_kernel<N>), grid, dim3(256), lds, st, L, R, out, C, G, H, W, D, idx, dpos, dneg,     \
                       out_channels, out_ch_offset);                                                             \
    break;
  switch (CG) {
    DMB_GWC_CASE(1)
    DMB_GWC_CASE(2)
    DMB_GWC_CASE(4)
    DMB_GWC_CASE(8)
    DMB_GWC_CASE(16)
    DMB_GWC_CASE(32)
    default:
      return fail(DMB_EUNSUPPORTED, "gwc: channels per group must be one of 1,2,4,8,16,32");
  }
#undef DMB_GWC_CASE
  return launch_status("gwc kernel launch failed");
}
