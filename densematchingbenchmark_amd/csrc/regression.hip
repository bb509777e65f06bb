// Disparity regression (soft-argmin family), cost up-sampling (trilinear align_corners=True and the
// AcfNet k8/s4 transposed convolution) and the EPE accumulator.  All HBM-bound: each kernel reads its
// input volume exactly once with 16-byte loads coalesced along W; reductions over the disparity axis
// walk planes H*W apart so that every wave instruction still covers 1 KiB of contiguous memory.
//
// Reference semantics: dmb/modeling/stereo/disp_predictors/{soft_argmin.py:45-75,
// faster_soft_argmin.py:51-75, local_soft_argmin.py:48-105}, cost_processors/aggregators/PSMNet.py:74-93,
// AcfNet.py:55-57,81-83, data/datasets/evaluation/stereo/pixel_error.py:6-73.
#include "dmb_common.h"
#include "interp.h"

// No implicit fma contraction in this file: the interpolation index/weight arithmetic must round exactly like the
// reference's (ATen CPU) float code -- src = scale * dst rounded, THEN lambda = src - floor(src) -- or lambda moves
// by an ulp of src (~6e-5 at src ~ 900) and the up-sampled costs with it.  Explicit fmaf() calls are unaffected.
#pragma clang fp contract(off)

namespace dmb {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------
// Online soft-argmin state for one pixel.  exp in FP32 (__expf), sum(e) and sum(e*d) in
// FP64 so that the result is the correctly rounded quotient; the reference's own FP32 evaluation sits up to
// ~1e-4 from this at D=192 on flat distributions.
// ---------------------------------------------------------------------------------------------------------
struct SoftState {
  // Lazy running maximum: the reference point m moves only when a block maximum exceeds it by more than
  // SOFT_SLACK, so a pixel sees at most (cost range / SOFT_SLACK) + 1 rescales instead of up to D on monotone cost
  // profiles (each rescale multiplies by an inexact exp); terms may exceed 1 (<= e^SOFT_SLACK), sums are FP64.
  static constexpr float SOFT_SLACK = 11.0f;
  float m;
  double s, t;
  __device__ void init() {
    m = -INFINITY;
    s = 0.0;
    t = 0.0;
  }
  // fold a block of N logits with sample values dv[]
  template <int N>
  __device__ void fold(const float (&v)[N], const float (&dv)[N]) {
    float bm = v[0];
#pragma unroll
    for (int i = 1; i < N; ++i) bm = fmaxf(bm, v[i]);
    if (bm > m + SOFT_SLACK) {
      const double f = (m == -INFINITY) ? 0.0 : (double)__expf(m - bm);
      s *= f;
      t *= f;
      m = bm;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float e = __expf(v[i] - m);
      s += (double)e;
      t = fma((double)e, (double)dv[i], t);
    }
  }
  __device__ float result() const { return (float)(t / s); }
};

constexpr int SA_BLK = 8;  // disparity planes folded per max-rescale step

// cost [B, D, H, W] -> disp [B, 1, H, W]; VEC pixels per thread.
template <int VEC, bool SAMPLED>
__global__ __launch_bounds__(256) void soft_argmin_kernel(const float* __restrict__ cost,
                                                          const float* __restrict__ sample,
                                                          float* __restrict__ disp, int D, int HW, float alpha,
                                                          int normalize, DispVal dv) {
  const int b = blockIdx.y;
  const int p = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (p >= HW) return;
  const float* cp = cost + (size_t)b * D * HW + p;
  const float* sp = SAMPLED ? sample + (size_t)b * D * HW + p : nullptr;
  float out[VEC];
  if (normalize) {
    SoftState st[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) st[j].init();
    int k = 0;
    for (; k + SA_BLK <= D; k += SA_BLK) {
      float v[VEC][SA_BLK], d[VEC][SA_BLK];
#pragma unroll
      for (int i = 0; i < SA_BLK; ++i) {
        if constexpr (VEC == 4) {
          const float4 q = *reinterpret_cast<const float4*>(cp + (size_t)(k + i) * HW);
          v[0][i] = q.x * alpha; v[1][i] = q.y * alpha; v[2][i] = q.z * alpha; v[3][i] = q.w * alpha;
          if (SAMPLED) {
            const float4 s4 = *reinterpret_cast<const float4*>(sp + (size_t)(k + i) * HW);
            d[0][i] = s4.x; d[1][i] = s4.y; d[2][i] = s4.z; d[3][i] = s4.w;
          }
        } else {
          v[0][i] = cp[(size_t)(k + i) * HW] * alpha;
          if (SAMPLED) d[0][i] = sp[(size_t)(k + i) * HW];
        }
        if (!SAMPLED) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) d[j][i] = dv.v[k + i];
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) st[j].template fold<SA_BLK>(v[j], d[j]);
    }
    for (; k < D; ++k) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float v1[1] = {cp[(size_t)k * HW + j] * alpha};
        float d1[1] = {SAMPLED ? sp[(size_t)k * HW + j] : dv.v[k]};
        st[j].template fold<1>(v1, d1);
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[j] = st[j].result();
  } else {
    // normalize=False: disp = sum_k (alpha * c_k) * d_k   (soft_argmin.py:56-59,73)
    double acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.0;
    for (int k = 0; k < D; ++k) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float c = cp[(size_t)k * HW + j] * alpha;
        const float d = SAMPLED ? sp[(size_t)k * HW + j] : dv.v[k];
        acc[j] = fma((double)c, (double)d, acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[j] = (float)acc[j];
  }
  float* dp = disp + (size_t)b * HW + p;
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(dp) = make_float4(out[0], out[1], out[2], out[3]);
  } else {
    dp[0] = out[0];
  }
}

// ---------------------------------------------------------------------------------------------------------
// LocalSoftArgmin: arg-max over D (first maximal index, exact FP32 compare on the un-scaled cost,
// local_soft_argmin.py:65), then a (2R+1)-tap masked softmax around it (:69-103).
// ---------------------------------------------------------------------------------------------------------
constexpr int LSA_MAX_TAPS = 33;

__global__ __launch_bounds__(256) void local_soft_argmin_kernel(const float* __restrict__ cost,
                                                                float* __restrict__ disp,
                                                                long long* __restrict__ argidx, int D, int HW,
                                                                int radius, int rdil, int start, int dil,
                                                                float alpha) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const float* cp = cost + (size_t)b * D * HW + p;
  float best = cp[0];
  int bi = 0;
  // torch.argmax: NaN counts as maximal and the first occurrence wins
  bool best_nan = best != best;
  for (int k = 1; k < D; ++k) {
    const float v = cp[(size_t)k * HW];
    if (!best_nan && (v > best || v != v)) {
      best = v;
      bi = k;
      best_nan = v != v;
    }
  }
  if (argidx) argidx[(size_t)b * HW + p] = (long long)bi;

  const int taps = 2 * radius + 1;
  const float fill = -10000.0f * alpha;
  float logit[LSA_MAX_TAPS], samp[LSA_MAX_TAPS];
  float mx = -INFINITY;
  for (int i = 0; i < taps; ++i) {
    const int raw = bi + (i - radius) * rdil;
    const float mask = (raw >= 0 && raw <= D - 1) ? 1.f : 0.f;
    const int ci = raw < 0 ? 0 : (raw > D - 1 ? D - 1 : raw);
    const float g = cp[(size_t)ci * HW] * alpha;
    // gathered * mask + (1 - mask) * (-10000 * alpha)   (local_soft_argmin.py:100)
    const float l = g * mask + (1.f - mask) * fill;
    logit[i] = l;
    samp[i] = (float)start + (float)ci * (float)dil;
    mx = fmaxf(mx, l);
  }
  double s = 0.0, t = 0.0;
  for (int i = 0; i < taps; ++i) {
    const float e = __expf(logit[i] - mx);
    s += (double)e;
    t = fma((double)e, (double)samp[i], t);
  }
  disp[(size_t)b * HW + p] = (float)(t / s);
}

// ---------------------------------------------------------------------------------------------------------
// Trilinear, align_corners=True.  Index/weight arithmetic follows ATen's CPU path that the reference's
// F.interpolate call reaches (area_pixel_compute_scale / compute_indices_weights): FP32 scale
// (in-1)/(out-1), src = scale*dst, i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1;
// evaluation order W, then H, then D.
// ---------------------------------------------------------------------------------------------------------
// one thread = 4 consecutive output x of one (b, zo, yo) row
__global__ __launch_bounds__(256) void trilinear_kernel(const float* __restrict__ x, float* __restrict__ y, int Di,
                                                        int Hi, int Wi, int Do, int Ho, int Wo, float sd, float sh,
                                                        float sw) {
  const int nxb = cdiv(cdiv(Wo, 4), 256);
  const int xq = (blockIdx.x % nxb) * 256 + threadIdx.x;  // float4 index along Wo
  const int xo = xq * 4;
  if (xo >= Wo) return;
  const int row = blockIdx.x / nxb;  // (zo, yo) flattened: grid.y is limited to 65535
  const int yo = row % Ho;
  const int zo = row / Ho;
  const int b = blockIdx.y;
  const Lerp lz = lerp_setup(zo, Di, sd), ly = lerp_setup(yo, Hi, sh);
  const float* xb = x + (size_t)b * Di * Hi * Wi;
  const float* r00 = xb + ((size_t)lz.i0 * Hi + ly.i0) * Wi;
  const float* r01 = xb + ((size_t)lz.i0 * Hi + ly.i1) * Wi;
  const float* r10 = xb + ((size_t)lz.i1 * Hi + ly.i0) * Wi;
  const float* r11 = xb + ((size_t)lz.i1 * Hi + ly.i1) * Wi;
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int xx = xo + j < Wo ? xo + j : Wo - 1;
    const Lerp lx = lerp_setup(xx, Wi, sw);
    const float a00 = lerp2(r00[lx.i0], lx.w0, r00[lx.i1], lx.w1);
    const float a01 = lerp2(r01[lx.i0], lx.w0, r01[lx.i1], lx.w1);
    const float a10 = lerp2(r10[lx.i0], lx.w0, r10[lx.i1], lx.w1);
    const float a11 = lerp2(r11[lx.i0], lx.w0, r11[lx.i1], lx.w1);
    const float h0 = lerp2(a00, ly.w0, a01, ly.w1);
    const float h1 = lerp2(a10, ly.w0, a11, ly.w1);
    o[j] = lerp2(h0, lz.w0, h1, lz.w1);
  }
  float* yp = y + (((size_t)b * Do + zo) * Ho + yo) * Wo + xo;
  if ((Wo & 3) == 0) {
    *reinterpret_cast<float4*>(yp) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (xo + j < Wo) yp[j] = o[j];
  }
}

// Column-streaming variant (the one used for real volumes): one thread owns 4 consecutive output x of one output row
// and walks ALL output planes.  The (H, W)-interpolated value of an input plane is computed once per thread and
// reused by every output plane that blends it, so an output costs ~0.4 L1/L2 reads instead of 8; no LDS, no barrier,
// stores are 16 bytes per lane and 1 KiB contiguous per wave.  Same arithmetic and order as trilinear_kernel
// (W, then H, then D), hence bit-identical results.
// WITH_DISP (zsplit == 1): the thread also folds every up-sampled logit of its 4 pixels into the soft-argmin, in the
// same blocks of SA_BLK planes and the same order as soft_argmin_kernel, and writes the disparity: the regression then
// costs no second pass over the [B, Do, Ho, Wo] volume (the volume is still written: the reference returns it).
// NX: consecutive output columns per thread: 4 (16-byte stores) -- or 1 for outputs so small that four columns per thread leave the chip
// with less than one wave per SIMD (round 6: one 256x512 map, 32768 threads walking 64 planes each, 20 us; 131072 threads: see
// dmb_trilinear_ac_soft_argmin_f32).  The arithmetic of a pixel does not depend on NX: bit-identical.
template <bool WITH_DISP, int NX = 4>
__global__ __launch_bounds__(256) void trilinear_zcol_kernel(const float* __restrict__ x, float* __restrict__ y, int Di,
                                                             int Hi, int Wi, int Do, int Ho, int Wo, float sd, float sh,
                                                             float sw, int zsplit, float* __restrict__ disp, float alpha,
                                                             DispVal dv, int flat) {
  // flat (Wo % 4 == 0): the (row, 4-column word) pairs of an output plane are ONE linear range -- word q sits at float offset
  // 4 q -- dealt to the threads in order, so every lane of every workgroup but the last has a word (a grid of rows x 256-word
  // blocks leaves 200 of the second block's 256 lanes idle on a 1248-column row: 0.40 -> 0.31 ms at the KITTI shape).
  // Otherwise blockIdx.y is the row and a row's words are dealt in blocks of 256.
  const int nxq = cdiv(Wo, NX);
  const int nxb = flat ? (int)(((long long)Ho * nxq + 255) / 256) : cdiv(nxq, 256);
  const int idx = (blockIdx.x % nxb) * 256 + threadIdx.x;
  const int yo = flat ? idx / nxq : (int)blockIdx.y;
  const int xq = flat ? idx - yo * nxq : idx;
  if (xq >= nxq || yo >= Ho) return;
  const int zpart = blockIdx.x / nxb;   // this thread walks output planes [zbeg, zend)
  const int zbeg = (int)((long long)Do * zpart / zsplit), zend = (int)((long long)Do * (zpart + 1) / zsplit);
  const int xo = xq * NX, b = blockIdx.z;
  const Lerp ly = lerp_setup(yo, Hi, sh);
  Lerp lx[NX];
#pragma unroll
  for (int j = 0; j < NX; ++j) lx[j] = lerp_setup(min(xo + j, Wo - 1), Wi, sw);
  const float* xb = x + (size_t)b * Di * Hi * Wi;
  const size_t plane = (size_t)Hi * Wi;
  const float* r0 = xb + (size_t)ly.i0 * Wi;
  const float* r1 = xb + (size_t)ly.i1 * Wi;
  auto hw = [&](int zi, float (&o)[NX]) {
    const float* p0 = r0 + (size_t)zi * plane;
    const float* p1 = r1 + (size_t)zi * plane;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const float a0 = lerp2(p0[lx[j].i0], lx[j].w0, p0[lx[j].i1], lx[j].w1);
      const float a1 = lerp2(p1[lx[j].i0], lx[j].w0, p1[lx[j].i1], lx[j].w1);
      o[j] = lerp2(a0, ly.w0, a1, ly.w1);
    }
  };
  // The input plane after the newest cached one is fetched AHEAD as raw values (16 registers) and only blended when it is
  // promoted, about four output planes later: loads and stores retire through one in-order counter on this chip, so a load
  // issued and awaited between two stores drains every store before it (one full write round trip per input plane).
  float raw[4 * NX];
  int pz = -1;
  auto prefetch = [&](int zi) {
    const float* p0 = r0 + (size_t)zi * plane;
    const float* p1 = r1 + (size_t)zi * plane;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      raw[4 * j + 0] = p0[lx[j].i0];
      raw[4 * j + 1] = p0[lx[j].i1];
      raw[4 * j + 2] = p1[lx[j].i0];
      raw[4 * j + 3] = p1[lx[j].i1];
    }
    pz = zi;
  };
  auto from_raw = [&](float (&o)[NX]) {
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const float a0 = lerp2(raw[4 * j + 0], lx[j].w0, raw[4 * j + 1], lx[j].w1);
      const float a1 = lerp2(raw[4 * j + 2], lx[j].w0, raw[4 * j + 3], lx[j].w1);
      o[j] = lerp2(a0, ly.w0, a1, ly.w1);
    }
  };
  float h0[NX], h1[NX];
  int cz0 = -1, cz1 = -1;
  const size_t ostride = (size_t)Ho * Wo;
  float* yp = y + ((size_t)b * Do * Ho + yo) * Wo + xo + (size_t)zbeg * ostride;
  const bool vec = NX == 4 && (Wo & 3) == 0;
  SoftState st[WITH_DISP ? NX : 1];
  float vb[WITH_DISP ? NX : 1][SA_BLK];
  if constexpr (WITH_DISP) {
#pragma unroll
    for (int j = 0; j < NX; ++j) st[j].init();
  }
  // one output plane: blend the two cached input planes, store the 4 columns, return them
  auto out_plane = [&](int zo, float (&o)[NX]) {
    const Lerp lz = lerp_setup(zo, Di, sd);
    if (lz.i0 != cz0) {
      if (lz.i0 == cz1) {
#pragma unroll
        for (int j = 0; j < NX; ++j) h0[j] = h1[j];
      } else {
        hw(lz.i0, h0);
      }
      cz0 = lz.i0;
    }
    if (lz.i1 != cz1) {
      if (lz.i1 == cz0) {
#pragma unroll
        for (int j = 0; j < NX; ++j) h1[j] = h0[j];
      } else if (lz.i1 == pz) {
        from_raw(h1);
      } else {
        hw(lz.i1, h1);
      }
      cz1 = lz.i1;
      if (cz1 + 1 < Di) prefetch(cz1 + 1);
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) o[j] = lerp2(h0[j], lz.w0, h1[j], lz.w1);
    if (vec) {
      // streaming store: 1.6 GB per launch that nothing re-reads soon must not wash the L2
      if constexpr (NX == 4) __builtin_nontemporal_store(f32x4_t{o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4_t*>(yp));
    } else {
#pragma unroll
      for (int j = 0; j < NX; ++j)
        if (xo + j < Wo) yp[j] = o[j];
    }
    yp += ostride;
  };
  if constexpr (WITH_DISP) {
    // planes [0, full) fold in blocks of SA_BLK, the tail one by one -- exactly as soft_argmin_kernel does
    const int full = Do - Do % SA_BLK;
    for (int zb = 0; zb < full; zb += SA_BLK) {
      float d[SA_BLK];
#pragma unroll
      for (int i = 0; i < SA_BLK; ++i) {
        float o[NX];
        out_plane(zb + i, o);
#pragma unroll
        for (int j = 0; j < NX; ++j) vb[j][i] = o[j] * alpha;
        d[i] = dv.v[zb + i];
      }
#pragma unroll
      for (int j = 0; j < NX; ++j) st[j].template fold<SA_BLK>(vb[j], d);
    }
    for (int zo = full; zo < Do; ++zo) {
      float o[NX];
      out_plane(zo, o);
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        float v1[1] = {o[j] * alpha};
        float d1[1] = {dv.v[zo]};
        st[j].template fold<1>(v1, d1);
      }
    }
  } else {
    for (int zo = zbeg; zo < zend; ++zo) {
      float o[NX];
      out_plane(zo, o);
    }
  }
  if constexpr (WITH_DISP) {
    float* dp = disp + ((size_t)b * Ho + yo) * Wo + xo;
#pragma unroll
    for (int j = 0; j < NX; ++j)
      if (xo + j < Wo) dp[j] = st[j].result();
  }
}

// Fused trilinear + soft-argmin: one thread per output pixel walks the Do output planes; the (H,W)-lerped
// value of each INPUT plane is computed once and reused by the output planes that blend it, with the same
// arithmetic (and therefore bit-identical logits) as trilinear_kernel.
__global__ __launch_bounds__(256) void trilinear_soft_argmin_kernel(const float* __restrict__ x,
                                                                    float* __restrict__ disp, int Di, int Hi, int Wi,
                                                                    int Do, int Ho, int Wo, float sd, float sh,
                                                                    float sw, float alpha, DispVal dv) {
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
  SoftState st;
  st.init();
  int cz0 = -1, cz1 = -1;
  float h0 = 0.f, h1 = 0.f;
  for (int zo = 0; zo < Do; ++zo) {
    const Lerp lz = lerp_setup(zo, Di, sd);
    if (lz.i0 != cz0) {
      if (lz.i0 == cz1) {
        h0 = h1;
      } else {
        h0 = hw(lz.i0);
      }
      cz0 = lz.i0;
    }
    if (lz.i1 != cz1) {
      h1 = (lz.i1 == cz0) ? h0 : hw(lz.i1);
      cz1 = lz.i1;
    }
    float v[1] = {lerp2(h0, lz.w0, h1, lz.w1) * alpha};
    float d[1] = {dv.v[zo]};
    st.template fold<1>(v, d);
  }
  disp[((size_t)b * Ho + yo) * Wo + xo] = st.result();
}

// ---------------------------------------------------------------------------------------------------------
// ConvTranspose3d(1, 1, k=8, s=4, p=2), output exactly 4x: y[o] = sum x[i] * w[k], o = 4i - 2 + k.
// Per axis an output o sees k in {(o+2)%4, (o+2)%4 + 4} from i = (o+2)/4 and (o+2)/4 - 1.
// One thread = 4 consecutive output x; FP32 fma chain in ascending (kd, kh, kw).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void deconv_k8s4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, int D, int H, int W) {
  __shared__ float ws[512];
  for (int t = threadIdx.x; t < 512; t += 256) ws[t] = w[t];
  __syncthreads();
  const int Wo = 4 * W, Ho = 4 * H, Do = 4 * D;
  const int nxb = cdiv(W, 256);
  const int q = (blockIdx.x % nxb) * 256 + threadIdx.x;  // input-resolution column: outputs 4q .. 4q+3
  if (q >= W) return;
  const int row = blockIdx.x / nxb;
  const int yo = row % Ho, zo = row / Ho, b = blockIdx.y;
  const int pz = (zo + 2) & 3, iz = (zo + 2) >> 2;
  const int py = (yo + 2) & 3, iy = (yo + 2) >> 2;
  const float* xb = x + (size_t)b * D * H * W;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // ascending kd: kd = pz uses input iz, kd = pz + 4 uses iz - 1
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int kd = pz + 4 * a, zi = iz - a;
    if (zi < 0 || zi >= D) continue;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int kh = py + 4 * c, yi = iy - c;
      if (yi < 0 || yi >= H) continue;
      const float* row = xb + ((size_t)zi * H + yi) * W;
      const float* wr = ws + (kd * 8 + kh) * 8;
      const float xm = q > 0 ? row[q - 1] : 0.f;
      const float x0 = row[q];
      const float xp = q + 1 < W ? row[q + 1] : 0.f;
      // output 4q+j: (o+2) = 4q + j + 2 -> j=0: phase 2, i=q ; j=1: phase 3, i=q ; j=2: phase 0, i=q+1 ; j=3: phase 1, i=q+1
      // taps in ascending kw: kw = phase uses i, kw = phase + 4 uses i - 1
      acc[0] = fmaf(x0, wr[2], acc[0]);
      acc[0] = fmaf(xm, wr[6], acc[0]);
      acc[1] = fmaf(x0, wr[3], acc[1]);
      acc[1] = fmaf(xm, wr[7], acc[1]);
      acc[2] = fmaf(xp, wr[0], acc[2]);
      acc[2] = fmaf(x0, wr[4], acc[2]);
      acc[3] = fmaf(xp, wr[1], acc[3]);
      acc[3] = fmaf(x0, wr[5], acc[3]);
    }
  }
  float* yp = y + (((size_t)b * Do + zo) * Ho + yo) * Wo + 4 * q;
  __builtin_nontemporal_store(f32x4_t{acc[0], acc[1], acc[2], acc[3]}, reinterpret_cast<f32x4_t*>(yp));   // streaming (1.6 GB)
}

// ---------------------------------------------------------------------------------------------------------
// EPE accumulator.  Pass 1: EPE_SLICES workgroups per image reduce a slice each and write six partial sums to the slice's
// workspace row; pass 2: one thread per image adds the slices in ascending order, turns the sums into the image's means and
// adds those to acc (one wave: see epe_finalize_kernel).
// ---------------------------------------------------------------------------------------------------------
constexpr int EPE_SLICES = 64;

// Several estimates against ONE ground truth in a launch (the three disparity maps of a PSMNet / AcfNet forward): blockIdx.z picks
// the estimate and its block of B workspace rows.
constexpr int EPE_MAXMAPS = 4;
struct EpeMaps {
  const float* est[EPE_MAXMAPS];
};

__global__ __launch_bounds__(256) void epe_partial_kernel(EpeMaps maps, const float* __restrict__ gt,
                                                          double* __restrict__ ws, int Hp, int Wp, int H0, int W0,
                                                          float lb, float ub) {
  const int b = blockIdx.y;
  const float* __restrict__ est = maps.est[blockIdx.z];
  ws += (size_t)blockIdx.z * gridDim.y * EPE_SLICES * 6;
  // eval.py:24-29: pad_top = Hp - H0; the crop [pad_top:, :W0] is applied only when pad_top >= 0
  const bool crop = Hp - H0 >= 0;
  const int top = crop ? Hp - H0 : 0;
  const int rows = Hp - top;
  const int cols = crop ? (W0 < Wp ? W0 : Wp) : Wp;
  const float* e = est + (size_t)b * Hp * Wp;
  const float* g = gt + (size_t)b * Hp * Wp;
  double sum = 0.0;
  unsigned cnt = 0, n1 = 0, n2 = 0, n3 = 0, n5 = 0;
  const int total = rows * cols;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += EPE_SLICES * 256) {
    const int r = i / cols, c = i - r * cols;
    const size_t o = (size_t)(top + r) * Wp + c;
    const float gv = g[o];
    if (gv > lb && gv < ub) {
      const float a = fabsf(gv - e[o]);
      sum += (double)a;
      cnt++;
      n1 += a > 1.f;
      n2 += a > 2.f;
      n3 += a > 3.f;
      n5 += a > 5.f;
    }
  }
  __shared__ double sh[6][4];
  double v[6] = {(double)cnt, sum, (double)n1, (double)n2, (double)n3, (double)n5};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double t = v[k];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = t;
  }
  __syncthreads();
  if (threadIdx.x < 6) {   // this slice's six sums, in its own workspace row: no atomics, no clearing, a fixed summation order
    const int k = threadIdx.x;
    ws[((size_t)b * EPE_SLICES + blockIdx.x) * 6 + k] = (sh[k][0] + sh[k][1]) + (sh[k][2] + sh[k][3]);
  }
}

// One 64-lane wave per estimate.  Per image the 64 slices are read by the 64 lanes at once and combined by a fixed butterfly
// (round 6: one lane walking 64 x 6 dependent FP64 adds took 13 us -- as long as a convolution layer of one small pair); the
// images are folded in ascending order and lane 0 adds the six totals to the accumulator row -- no atomics anywhere: the metrics
// of a run are reproducible bit for bit, whatever the dispatch order.  (The counts are integers below 2^53: exact in any order.)
__global__ __launch_bounds__(64) void epe_finalize_kernel(const double* __restrict__ ws, double* __restrict__ acc, int B) {
  static_assert(EPE_SLICES == 64, "one slice per lane");
  const double* maps = ws + (size_t)blockIdx.x * B * EPE_SLICES * 6;   // blockIdx.x: the estimate; its accumulator row follows
  acc += blockIdx.x * 6;
  double tot[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int b = 0; b < B; ++b) {
    const double* slice = maps + ((size_t)b * EPE_SLICES + threadIdx.x) * 6;
    double r[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double t = slice[k];
      for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);   // every lane ends with the same total
      r[k] = t;
    }
    tot[0] += 1.0;
    if (r[0] >= 1.0) {  // pixel_error.py:48: an empty mask yields all-zero errors for this image
      tot[1] += r[1] / r[0];
#pragma unroll
      for (int k = 2; k < 6; ++k) tot[k] += 100.0 * r[k] / r[0];
    }
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] += tot[k];
  }
}

static int fill_samples(const float* host, int D, DispVal& dv) {
  if (!host || D <= 0 || D > DMB_MAX_DISP_SAMPLES) return fail(DMB_EINVAL, "disparity sample count out of range");
  for (int k = 0; k < D; ++k) dv.v[k] = host[k];
  return DMB_OK;
}

}  // namespace dmb

using namespace dmb;

extern "C" int dmb_soft_argmin_f32(const float* cost, float* disp, int B, int D, int H, int W, float alpha,
                                   int normalize, const float* disp_sample_host, void* stream) {
  if (!cost || !disp || B <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "soft_argmin: bad argument");
  DispVal dv;
  if (int e = fill_samples(disp_sample_host, D, dv)) return e;
  const int HW = H * W;
  hipStream_t st = (hipStream_t)stream;
  if (HW % 4 == 0) {
    dim3 grid(cdiv(HW / 4, 256), B);
    hipLaunchKernelGGL((soft_argmin_kernel<4, false>), grid, dim3(256), 0, st, cost, (const float*)nullptr, disp, D,
                       HW, alpha, normalize, dv);
  } else {
    dim3 grid(cdiv(HW, 256), B);
    hipLaunchKernelGGL((soft_argmin_kernel<1, false>), grid, dim3(256), 0, st, cost, (const float*)nullptr, disp, D,
                       HW, alpha, normalize, dv);
  }
  return launch_status("soft_argmin launch failed");
}

extern "C" int dmb_soft_argmin_sampled_f32(const float* cost, const float* sample, float* disp, int B, int D, int H,
                                           int W, float alpha, int normalize, void* stream) {
  if (!cost || !sample || !disp || B <= 0 || D <= 0 || H <= 0 || W <= 0)
    return fail(DMB_EINVAL, "soft_argmin_sampled: bad argument");
  DispVal dv;  // unused in the sampled instantiation
  dv.v[0] = 0.f;
  const int HW = H * W;
  hipStream_t st = (hipStream_t)stream;
  if (HW % 4 == 0) {
    dim3 grid(cdiv(HW / 4, 256), B);
    hipLaunchKernelGGL((soft_argmin_kernel<4, true>), grid, dim3(256), 0, st, cost, sample, disp, D, HW, alpha,
                       normalize, dv);
  } else {
    dim3 grid(cdiv(HW, 256), B);
    hipLaunchKernelGGL((soft_argmin_kernel<1, true>), grid, dim3(256), 0, st, cost, sample, disp, D, HW, alpha,
                       normalize, dv);
  }
  return launch_status("soft_argmin_sampled launch failed");
}

extern "C" int dmb_local_soft_argmin_f32(const float* cost, float* disp, long long* argidx, int B, int D, int H,
                                         int W, int radius, int radius_dilation, int start_disp, int dilation,
                                         float alpha, void* stream) {
  if (!cost || !disp || B <= 0 || D <= 0 || H <= 0 || W <= 0 || radius < 0)
    return fail(DMB_EINVAL, "local_soft_argmin: bad argument");
  if (2 * radius + 1 > LSA_MAX_TAPS) return fail(DMB_EUNSUPPORTED, "local_soft_argmin: radius > 16");
  const int HW = H * W;
  dim3 grid(cdiv(HW, 256), B);
  hipLaunchKernelGGL(local_soft_argmin_kernel, grid, dim3(256), 0, (hipStream_t)stream, cost, disp, argidx, D, HW,
                     radius, radius_dilation, start_disp, dilation, alpha);
  return launch_status("local_soft_argmin launch failed");
}

// Row form or flat form of trilinear_zcol_kernel (see the kernel): flat wherever rows are 16-byte multiples.  Measured inside
// the PSMNet step, forms alternated in one process (scripts/ab_step.py, development option 22): 544 x 960 (240 of a row block's 256
// lanes busy) 27.05 ms with the row form against 26.94 flat; 384 x 1248 (312 of 512) 26.60 against 26.24.
static int trilinear_flat(int Ho, int Wo) {
  if ((Wo & 3) != 0 || (long long)Ho * (Wo / 4) >= 0x7fffff00LL) return 0;
  if (DMB_OPT(22)) return DMB_OPT(22) == 2;   // (development option 22: 1 = row form, 2 = flat form)
  return 1;
}

extern "C" int dmb_trilinear_ac_f32(const float* x, float* y, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                    void* stream) {
  if (!x || !y || B <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0)
    return fail(DMB_EINVAL, "trilinear: bad argument");
  if ((long long)cdiv(cdiv(Wo, 4), 256) * Do * Ho > 0x7fffffffLL || B > 65535)
    return fail(DMB_EUNSUPPORTED, "trilinear: grid too large");
  hipStream_t st = (hipStream_t)stream;
  if (Ho <= 65535 && B <= 65535) {
    const int flat = trilinear_flat(Ho, Wo);
    const long long nxb = flat ? ((long long)Ho * (Wo / 4) + 255) / 256 : cdiv(cdiv(Wo, 4), 256);
    // split the plane walk only when the (row, column-block) grid alone cannot fill the chip
    const long long nblk0 = nxb * (flat ? 1 : Ho) * B;
    int zsplit = 1;
    while (nblk0 * zsplit < 2048 && zsplit * 2 <= Do && zsplit < 16) zsplit *= 2;
    hipLaunchKernelGGL(trilinear_zcol_kernel<false>, dim3((unsigned)(nxb * zsplit), flat ? 1 : Ho, B), dim3(256), 0, st, x, y, Di, Hi,
                       Wi, Do, Ho, Wo, ac_scale(Di, Do), ac_scale(Hi, Ho), ac_scale(Wi, Wo), zsplit, (float*)nullptr, 1.f,
                       DispVal{}, flat);
  } else {
    dim3 grid(cdiv(cdiv(Wo, 4), 256) * Do * Ho, B);
    hipLaunchKernelGGL(trilinear_kernel, grid, dim3(256), 0, st, x, y, Di, Hi, Wi, Do, Ho, Wo,
                       ac_scale(Di, Do), ac_scale(Hi, Ho), ac_scale(Wi, Wo));
  }
  return launch_status("trilinear launch failed");
}

extern "C" int dmb_trilinear_ac_soft_argmin_f32(const float* x, float* y, float* disp, int B, int Di, int Hi, int Wi,
                                                int Do, int Ho, int Wo, float alpha, const float* disp_sample_host,
                                                void* stream) {
  if (!x || !y || !disp || B <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0)
    return fail(DMB_EINVAL, "trilinear_ac_soft_argmin: bad argument");
  if (Ho > 65535 || B > 65535) return fail(DMB_EUNSUPPORTED, "trilinear_ac_soft_argmin: grid too large");
  DispVal dv;
  if (int e = fill_samples(disp_sample_host, Do, dv)) return e;
  const int flat = trilinear_flat(Ho, Wo);
  // (round 6) fewer than one wave per SIMD at four columns per thread: one column per thread -- four times the waves, each with a
  // quarter of the per-plane arithmetic of its serial walk over the Do planes (one 256x512 map: 20 -> 12 us); same bits
  if ((long long)B * Ho * cdiv(Wo, 4) < 4LL * 64 * num_cus() && (long long)Ho * Wo < 0x7fffff00LL) {
    const long long nxb1 = flat ? ((long long)Ho * Wo + 255) / 256 : cdiv(Wo, 256);
    hipLaunchKernelGGL((trilinear_zcol_kernel<true, 1>), dim3((unsigned)nxb1, flat ? 1 : Ho, B), dim3(256), 0, (hipStream_t)stream, x, y,
                       Di, Hi, Wi, Do, Ho, Wo, ac_scale(Di, Do), ac_scale(Hi, Ho), ac_scale(Wi, Wo), 1, disp, alpha, dv, flat);
    return launch_status("trilinear_ac_soft_argmin launch failed");
  }
  const long long nxb = flat ? ((long long)Ho * (Wo / 4) + 255) / 256 : cdiv(cdiv(Wo, 4), 256);
  hipLaunchKernelGGL(trilinear_zcol_kernel<true>, dim3((unsigned)nxb, flat ? 1 : Ho, B), dim3(256), 0, (hipStream_t)stream, x, y,
                     Di, Hi, Wi, Do, Ho, Wo, ac_scale(Di, Do), ac_scale(Hi, Ho), ac_scale(Wi, Wo), 1, disp, alpha, dv, flat);
  return launch_status("trilinear_ac_soft_argmin launch failed");
}

extern "C" int dmb_trilinear_soft_argmin_f32(const float* x, float* disp, int B, int Di, int Hi, int Wi, int Do,
                                             int Ho, int Wo, float alpha, const float* disp_sample_host,
                                             void* stream) {
  if (!x || !disp || B <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0)
    return fail(DMB_EINVAL, "trilinear_soft_argmin: bad argument");
  DispVal dv;
  if (int e = fill_samples(disp_sample_host, Do, dv)) return e;
  dim3 grid(cdiv(Wo, 256), Ho, B);
  hipLaunchKernelGGL(trilinear_soft_argmin_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, disp, Di, Hi, Wi, Do,
                     Ho, Wo, ac_scale(Di, Do), ac_scale(Hi, Ho), ac_scale(Wi, Wo), alpha, dv);
  return launch_status("trilinear_soft_argmin launch failed");
}

extern "C" int dmb_deconv3d_k8s4_c1_f32(const float* x, const float* w, float* y, int B, int D, int H, int W,
                                        void* stream) {
  if (!x || !w || !y || B <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "deconv_k8s4: bad argument");
  if ((long long)cdiv(W, 256) * 16 * D * H > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "deconv_k8s4: grid too large");
  dim3 grid(cdiv(W, 256) * 16 * D * H, B);
  hipLaunchKernelGGL(deconv_k8s4_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, w, y, D, H, W);
  return launch_status("deconv_k8s4 launch failed");
}

namespace dmb {

// The same learned up-sampling as deconv_k8s4_kernel, as a z-column walk (the shape of trilinear_zcol_kernel): a thread owns
// the four outputs 4q .. 4q + 3 of one output row and walks the Do = 4 D output planes.  The row's y phase -- hence the two
// kh taps -- is uniform over the workgroup and the plane's two kd taps over the loop iteration, so the 32 weights of an output
// plane are scalar loads; the six input values of a (plane, row) pair are kept for the two input planes in use, and the next
// input plane's are fetched one step AHEAD (a load awaited between two stores would drain every store before it).  With
// WITH_DISP the logits are folded into the soft-argmin exactly as soft_argmin_kernel folds them (same blocks, same order):
// the regression costs no second pass over the 1.6 GB volume.  Per-output accumulation order = deconv_k8s4_kernel's.
template <bool WITH_DISP>
__global__ __launch_bounds__(256) void deconv_k8s4_zcol_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               float* __restrict__ y, float* __restrict__ disp, int D, int H,
                                                               int W, float alpha, DispVal dv) {
  const int Wo = 4 * W, Ho = 4 * H, Do = 4 * D;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;   // (workgroups of 64 .. 256 threads: the host evens a row's threads out over its workgroups)
  if (q >= W) return;
  const int yo = blockIdx.y, b = blockIdx.z;
  const int py = (yo + 2) & 3, iy = (yo + 2) >> 2;   // kh = py uses input row iy, kh = py + 4 uses iy - 1
  const float* xb = x + (size_t)b * D * H * W;
  const bool rowok[2] = {iy < H, iy - 1 >= 0};
  // v[c][3]: input values (q - 1, q, q + 1) of row iy - c of ONE input plane
  auto fetch = [&](int zi, float (&v)[2][3]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bool ok = zi >= 0 && zi < D && rowok[c];
      const float* row = xb + ((size_t)(ok ? zi : 0) * H + (ok ? iy - c : 0)) * W;
      v[c][0] = (ok && q > 0) ? row[q - 1] : 0.f;
      v[c][1] = ok ? row[q] : 0.f;
      v[c][2] = (ok && q + 1 < W) ? row[q + 1] : 0.f;
    }
  };
  float cur[2][3], prev[2][3], nxt[2][3];   // input planes iz, iz - 1 and the prefetched iz + 1
  int iz_cur = 0;                           // plane zo = 0: iz = (0 + 2) >> 2 = 0
  fetch(0, cur);
  fetch(-1, prev);
  fetch(1, nxt);
  SoftState st[4];
  float vb[4][SA_BLK];
  if constexpr (WITH_DISP) {
#pragma unroll
    for (int j = 0; j < 4; ++j) st[j].init();
  }
  float* yp = y + (((size_t)b * Do) * Ho + yo) * Wo + 4 * q;
  const size_t ostride = (size_t)Ho * Wo;
  auto out_plane = [&](int zo, float (&o)[4]) {
    const int pz = (zo + 2) & 3, iz = (zo + 2) >> 2;
    if (iz != iz_cur) {   // advances by one every four planes (uniform)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          prev[c][k] = cur[c][k];
          cur[c][k] = nxt[c][k];
        }
      iz_cur = iz;
      fetch(iz + 1, nxt);
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a) {   // ascending kd: kd = pz uses input iz, kd = pz + 4 uses iz - 1
      const int kd = pz + 4 * a, zi = iz - a;
      if (zi < 0 || zi >= D) continue;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (!rowok[c]) continue;
        const float* wr = w + (kd * 8 + py + 4 * c) * 8;   // uniform: scalar loads
        const float xm = a ? prev[c][0] : cur[c][0], x0 = a ? prev[c][1] : cur[c][1], xp = a ? prev[c][2] : cur[c][2];
        acc[0] = fmaf(x0, wr[2], acc[0]);
        acc[0] = fmaf(xm, wr[6], acc[0]);
        acc[1] = fmaf(x0, wr[3], acc[1]);
        acc[1] = fmaf(xm, wr[7], acc[1]);
        acc[2] = fmaf(xp, wr[0], acc[2]);
        acc[2] = fmaf(x0, wr[4], acc[2]);
        acc[3] = fmaf(xp, wr[1], acc[3]);
        acc[3] = fmaf(x0, wr[5], acc[3]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = acc[j];
    __builtin_nontemporal_store(f32x4_t{acc[0], acc[1], acc[2], acc[3]}, reinterpret_cast<f32x4_t*>(yp));   // streaming (1.6 GB)
    yp += ostride;
  };
  if constexpr (WITH_DISP) {
    const int full = Do - Do % SA_BLK;
    for (int zb = 0; zb < full; zb += SA_BLK) {
      float d[SA_BLK];
#pragma unroll
      for (int i = 0; i < SA_BLK; ++i) {
        float o[4];
        out_plane(zb + i, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) vb[j][i] = o[j] * alpha;
        d[i] = dv.v[zb + i];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) st[j].template fold<SA_BLK>(vb[j], d);
    }
    for (int zo = full; zo < Do; ++zo) {
      float o[4];
      out_plane(zo, o);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v1[1] = {o[j] * alpha};
        float d1[1] = {dv.v[zo]};
        st[j].template fold<1>(v1, d1);
      }
    }
    float* dp = disp + ((size_t)b * Ho + yo) * Wo + 4 * q;
    *reinterpret_cast<float4*>(dp) = make_float4(st[0].result(), st[1].result(), st[2].result(), st[3].result());
  } else {
    for (int zo = 0; zo < Do; ++zo) {
      float o[4];
      out_plane(zo, o);
    }
  }
}

}  // namespace dmb

using namespace dmb;

extern "C" int dmb_deconv3d_k8s4_c1_soft_argmin_f32(const float* x, const float* w, float* y, float* disp, int B, int D, int H,
                                                    int W, float alpha, const float* disp_sample_host, void* stream) {
  if (!x || !w || !y || B <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "deconv_k8s4_soft_argmin: bad argument");
  if (4 * H > 65535 || B > 65535) return fail(DMB_EUNSUPPORTED, "deconv_k8s4_soft_argmin: grid too large");
  // a row's W threads in cdiv(W, 256) workgroups of equal size, rounded up to whole waves (312 input columns: 2 x 192 threads
  // instead of 256 + 56 of 256)
  const int nwg = cdiv(W, 256), wgt = cdiv(cdiv(W, nwg), 64) * 64;
  dim3 grid(nwg, 4 * H, B);
  if (disp) {
    DispVal dv;
    if (int e = fill_samples(disp_sample_host, 4 * D, dv)) return e;
    hipLaunchKernelGGL(deconv_k8s4_zcol_kernel<true>, grid, dim3(wgt), 0, (hipStream_t)stream, x, w, y, disp, D, H, W, alpha, dv);
  } else {
    DispVal dv = {};
    hipLaunchKernelGGL(deconv_k8s4_zcol_kernel<false>, grid, dim3(wgt), 0, (hipStream_t)stream, x, w, y, disp, D, H, W, alpha, dv);
  }
  return launch_status("deconv_k8s4_soft_argmin launch failed");
}

extern "C" int dmb_epe_accum_f64(const float* est, const float* gt, double* acc, double* workspace, int B, int Hp,
                                 int Wp, int H0, int W0, float lb, float ub, void* stream) {
  if (!est || !gt || !acc || !workspace || B <= 0 || B > 65535 || Hp <= 0 || Wp <= 0 || H0 <= 0 || W0 <= 0)
    return fail(DMB_EINVAL, "epe_accum: bad argument");
  hipStream_t st = (hipStream_t)stream;
  EpeMaps maps{};
  maps.est[0] = est;
  hipLaunchKernelGGL(epe_partial_kernel, dim3(EPE_SLICES, B), dim3(256), 0, st, maps, gt, workspace, Hp, Wp, H0, W0, lb, ub);
  hipLaunchKernelGGL(epe_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, acc, B);
  return launch_status("epe_accum launch failed");
}

extern "C" int dmb_epe_accum_multi_f64(int nmaps, const float* const* est, const float* gt, double* acc, double* workspace, int B,
                                       int Hp, int Wp, int H0, int W0, float lb, float ub, void* stream) {
  if (nmaps <= 0 || nmaps > EPE_MAXMAPS || !est || !gt || !acc || !workspace || B <= 0 || B > 65535 || Hp <= 0 || Wp <= 0 ||
      H0 <= 0 || W0 <= 0)
    return fail(DMB_EINVAL, "epe_accum_multi: bad argument (1 .. 4 estimates)");
  EpeMaps maps{};
  for (int i = 0; i < nmaps; ++i) {
    if (!est[i]) return fail(DMB_EINVAL, "epe_accum_multi: null estimate");
    maps.est[i] = est[i];
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(epe_partial_kernel, dim3(EPE_SLICES, B, nmaps), dim3(256), 0, st, maps, gt, workspace, Hp, Wp, H0, W0, lb, ub);
  hipLaunchKernelGGL(epe_finalize_kernel, dim3(nmaps), dim3(64), 0, st, workspace, acc, B);
  return launch_status("epe_accum_multi launch failed");
}
