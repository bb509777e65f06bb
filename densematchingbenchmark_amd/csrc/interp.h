// align_corners=True linear interpolation indices / weights, exactly as the CPU code path the reference's
// F.interpolate call reaches (area_pixel_compute_scale / compute_indices_weights): FP32 scale (in-1)/(out-1),
// src = scale*dst, i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1.  Include from translation units
// src must round BEFORE the subtraction (an fma there moves lambda by an ulp of src, ~6e-5 at src ~ 900, and the
// up-sampled costs with it), so this header switches fma contraction off for the rest of the translation unit.
#pragma once
#include "dmb_common.h"

#pragma clang fp contract(off)

namespace dmb {

struct Lerp {
  int i0, i1;
  float w0, w1;
};
__device__ inline Lerp lerp_setup(int dst, int in, float scale) {
  const float src = scale * (float)dst;
  Lerp l;
  l.i0 = (int)src;
  l.i1 = l.i0 + ((l.i0 < in - 1) ? 1 : 0);
  float l1 = src - (float)l.i0;
  l1 = fminf(fmaxf(l1, 0.f), 1.f);
  l.w1 = l1;
  l.w0 = 1.f - l1;
  return l;
}
__host__ __device__ inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

__device__ inline float lerp2(float a, float wa, float b, float wb) { return fmaf(b, wb, a * wa); }

}  // namespace dmb
