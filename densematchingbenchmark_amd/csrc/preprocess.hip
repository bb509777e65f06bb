// The data-side conventions in front of the path (dmb/data/transforms/stereo_trans.py:78-119, order
// dmb/data/datasets/stereo/builder.py:22-28 and dmb/apis/inference.py:120-129): take a window of the decoded image
// (CenterCrop), zero-pad it on the TOP and on the RIGHT to the network's input shape (StereoPad) and normalise per channel,
// (x - mean[c]) / std[c] (Normalize = torchvision's sub_ then div_) -- padding BEFORE normalisation, so a padded pixel
// holds (0 - mean[c]) / std[c], not 0.  One pass, HBM-bound (4 bytes written per output element, 1 or 4 read): the source is
// either a planar FP32 tensor [B, Cs, sh, sw] or the decoder's interleaved bytes [B, sh, sw, Cs] (imread's layout; the first
// C of Cs channels are taken, stereo/scene_flow/base.py:17-23), whose uint8 -> float conversion is exact.
//
// Arithmetic: one FP32 subtraction and one correctly rounded FP32 division per element (hipcc's default; no reciprocal
// multiply), i.e. bit for bit what the reference's transform computes on the host.
#include "dmb_common.h"

namespace dmb {

struct ChanAffine {
  float mean[4];
  float std[4];
};

template <bool U8>
__global__ __launch_bounds__(256) void pad_normalize_kernel(const void* __restrict__ src_, float* __restrict__ dst, int C, int Cs,
                                                            int sh, int sw, int y0, int x0, int h, int w, int th, int tw,
                                                            int normalize, ChanAffine aff, long long total4) {
  // one thread = 4 consecutive columns of one output row (tw % 4 == 0 is checked on the host: 16-byte stores)
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total4) return;
  const int q = tw >> 2;
  const int xq = (int)(t % q);
  long long r = t / q;
  const int y = (int)(r % th);
  r /= th;
  const int c = (int)(r % C);
  const int b = (int)(r / C);
  const int pad_top = th - h;
  const int ys = y - pad_top;     // row inside the window; negative = padding
  const float m = aff.mean[c], s = aff.std[c];
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int x = xq * 4 + j;
    float a = 0.f;
    if (ys >= 0 && x < w) {
      const long long sy = ys + y0, sx = x + x0;
      if (U8)
        a = (float)static_cast<const unsigned char*>(src_)[(((long long)b * sh + sy) * sw + sx) * Cs + c];
      else
        a = static_cast<const float*>(src_)[(((long long)b * Cs + c) * sh + sy) * sw + sx];
    }
    v[j] = normalize ? (a - m) / s : a;
  }
  *reinterpret_cast<float4*>(dst + (((long long)b * C + c) * th + y) * tw + xq * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

static int launch(bool u8, const void* src, float* dst, int B, int C, int Cs, int sh, int sw, int y0, int x0, int h, int w, int th,
                  int tw, const float* mean_host, const float* std_host, hipStream_t st) {
  if (!src || !dst) return fail(DMB_EINVAL, "stereo_pad_normalize: NULL pointer");
  if (B <= 0 || C <= 0 || C > 4 || Cs < C || sh <= 0 || sw <= 0 || h <= 0 || w <= 0 || th < h || tw < w)
    return fail(DMB_EINVAL, "stereo_pad_normalize: sizes (1..4 channels, target >= window)");
  if (y0 < 0 || x0 < 0 || y0 + h > sh || x0 + w > sw) return fail(DMB_EINVAL, "stereo_pad_normalize: window outside the source image");
  if (tw % 4) return fail(DMB_EUNSUPPORTED, "stereo_pad_normalize: target width must be a multiple of 4 (16-byte rows)");
  if ((mean_host == nullptr) != (std_host == nullptr)) return fail(DMB_EINVAL, "stereo_pad_normalize: mean and std come together");
  ChanAffine aff;
  for (int c = 0; c < 4; ++c) {
    aff.mean[c] = (mean_host && c < C) ? mean_host[c] : 0.f;
    aff.std[c] = (std_host && c < C) ? std_host[c] : 1.f;
  }
  const long long total4 = (long long)B * C * th * (tw / 4);
  const long long blocks = (total4 + 255) / 256;
  if (blocks > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "stereo_pad_normalize: too many elements for one launch");
  if (u8)
    pad_normalize_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(src, dst, C, Cs, sh, sw, y0, x0, h, w, th, tw,
                                                                            mean_host != nullptr, aff, total4);
  else
    pad_normalize_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(src, dst, C, Cs, sh, sw, y0, x0, h, w, th, tw,
                                                                             mean_host != nullptr, aff, total4);
  return launch_status("stereo_pad_normalize launch failed");
}

}  // namespace dmb

extern "C" int dmb_stereo_pad_normalize_f32(const float* src, float* dst, int B, int C, int Cs, int sh, int sw, int y0, int x0, int h,
                                            int w, int th, int tw, const float* mean_host, const float* std_host, void* stream) {
  return dmb::launch(false, src, dst, B, C, Cs, sh, sw, y0, x0, h, w, th, tw, mean_host, std_host, (hipStream_t)stream);
}

extern "C" int dmb_stereo_pad_normalize_u8(const unsigned char* src_hwc, float* dst, int B, int C, int Cs, int sh, int sw, int y0,
                                           int x0, int h, int w, int th, int tw, const float* mean_host, const float* std_host,
                                           void* stream) {
  return dmb::launch(true, src_hwc, dst, B, C, Cs, sh, sw, y0, x0, h, w, th, tw, mean_host, std_host, (hipStream_t)stream);
}
