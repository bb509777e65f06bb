// Training-side "cost filtering" of AcfNet (first part): the three loss terms of the reference,
// each as ONE forward pass and ONE backward pass over the tensors they read.
//   StereoFocalLoss + LaplaceDisp2Prob   dmb/modeling/stereo/losses/stereo_focal_loss.py:63-101,
//                                        dmb/modeling/stereo/losses/utils/disp2prob.py:107-173
//   ConfidenceNllLoss                    dmb/modeling/stereo/losses/conf_nll_loss.py:35-56
//   DispSmoothL1Loss                     dmb/modeling/stereo/losses/smooth_l1_loss.py:36-58
// The reference materialises five [B, D, H, W] temporaries per cost level (disparity samples, target probability,
// log-softmax, focal weight, product); here the target distribution is recomputed per pixel in registers (it depends
// only on the ground-truth disparity and the variance), the cost volume is read once per pass, and the only
// full-size tensor written is the gradient.  All reductions are FP64, per block then one finalising block
// (deterministic: no floating-point atomics).  HBM-bound: forward reads 4 B per cost element, backward reads 4 B and
// writes 4 B.
#include "dmb_common.h"

#pragma clang fp contract(off)

namespace dmb {

constexpr int LOSS_BLOCK = 256;

__device__ __forceinline__ double block_sum(double v, double* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

// Per-pixel target distribution of LaplaceDisp2Prob: p_d = softmax_d(-|s_d - g| / v) * m2 + eps, with
// g = gt * m1 * m2 (the two masks of stereo_focal_loss.py:80 and disp2prob.py:127-129).  The distribution depends only
// on (g, v): its normaliser (and, for the variance gradient, E = sum_k p_k |s_k - g| / v^2) is computed once per pixel
// from registers; the passes over the cost column then cost one exp per plane for it.
struct Target {
  float g, m1, m2, inv_v, inv_v2, tmax, inv_tsum, e;
  __device__ void setup(float gt, float var, float lb, float ub, float start, float end, int D, const DispVal& dv,
                        bool want_e) {
    m1 = (gt > lb && gt < ub) ? 1.f : 0.f;
    const float gg = gt * m1;
    m2 = (gg > start && gg < end) ? 1.f : 0.f;
    g = gg * m2;
    inv_v = 1.f / var;
    inv_v2 = inv_v * inv_v;
    float amin = INFINITY;   // the largest logit belongs to the sample nearest to g
    for (int d = 0; d < D; ++d) amin = fminf(amin, fabsf(dv.v[d] - g));
    tmax = -amin * inv_v;
    float tsum = 0.f, es = 0.f;
    for (int d = 0; d < D; ++d) {
      const float ad = fabsf(dv.v[d] - g);
      const float ex = __expf(-ad * inv_v - tmax);
      tsum += ex;
      if (want_e) es = fmaf(ex, ad, es);
    }
    inv_tsum = 1.f / tsum;
    e = es * inv_tsum * inv_v2;
  }
  __device__ float p(float s) const { return __expf(-fabsf(s - g) * inv_v - tmax) * inv_tsum; }   // plain softmax term
};

// a(P) = P * (1 - P)^(-fc); pw = (1 - P)^(-fc) is returned for the derivative a'(P) = pw + fc * P * pw / (1 - P).
__device__ __forceinline__ float focal_pow(float P, float fc) { return fc == 0.f ? 1.f : __expf(-fc * __logf(1.f - P)); }

// Forward: one thread per pixel.  stats[pixel] = (logsumexp_d cost, sum_d a_d) for the backward pass.
// partial[block] = (sum of -sum_d a_d * log q_d * m1, number of valid pixels).
__global__ __launch_bounds__(LOSS_BLOCK) void focal_fwd_kernel(const float* __restrict__ cost, const float* __restrict__ gt,
                                                               const float* __restrict__ var, float var_scalar,
                                                               float2* __restrict__ stats, double* __restrict__ partial,
                                                               int D, int HW, long long npix, DispVal dv, float lb,
                                                               float ub, float start, float end, float fc, float eps) {
  __shared__ double sm[4];
  const long long pix = blockIdx.x * (long long)LOSS_BLOCK + threadIdx.x;
  double loss = 0.0, cnt = 0.0;
  if (pix < npix) {
    const long long b = pix / HW, r = pix - b * HW;
    Target tg;
    tg.setup(gt[pix], var ? var[pix] : var_scalar, lb, ub, start, end, D, dv, false);
    const float* cp = cost + b * (long long)D * HW + r;
    // online logsumexp of the cost column, sum_d a_d * c_d and sum_d a_d in the same pass
    float m = -INFINITY;
    double se = 0.0, sac = 0.0, sa = 0.0;
    for (int d = 0; d < D; ++d) {
      const float c = cp[(long long)d * HW];
      if (c > m) {
        se *= (m == -INFINITY) ? 0.0 : (double)__expf(m - c);
        m = c;
      }
      se += (double)__expf(c - m);
      const float P = tg.p(dv.v[d]) * tg.m2 + eps;
      const float a = P * focal_pow(P, fc);
      sac = fma((double)a, (double)c, sac);
      sa += (double)a;
    }
    const double lse = (double)m + log(se);
    stats[pix] = make_float2((float)lse, (float)sa);
    loss = -(sac - lse * sa) * (double)tg.m1;
    cnt = (double)tg.m1;
  }
  const double bl = block_sum(loss, sm), bc = block_sum(cnt, sm);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = bl;
    partial[2 * blockIdx.x + 1] = bc;
  }
}

// out[0] = loss = sum / max(count, 1) (FP32, what the reference returns), out[1] = max(count, 1).
__global__ void loss_finalize_kernel(const double* __restrict__ partial, int nblk, float* __restrict__ out, int mean_over_all,
                                     long long n_all) {
  __shared__ double sm[4];
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nblk; i += LOSS_BLOCK) {
    s += partial[2 * i];
    c += partial[2 * i + 1];
  }
  s = block_sum(s, sm);
  c = block_sum(c, sm);
  if (threadIdx.x == 0) {
    const double n = mean_over_all ? (double)n_all : (c < 1.0 ? 1.0 : c);
    out[0] = (float)(s / n);
    out[1] = (float)n;
  }
}

// Backward: grad_cost[b, d, p] = go * m1 / N * (A * softmax(cost)_d - a_d);
// grad_var[p] = -go * m1 / N * sum_d log q_d * a'(P_d) * m2 * p_d * (dt_d - sum_k p_k dt_k), dt_d = |s_d - g| / v^2.
__global__ __launch_bounds__(LOSS_BLOCK) void focal_bwd_kernel(const float* __restrict__ cost, const float* __restrict__ gt,
                                                               const float* __restrict__ var, float var_scalar,
                                                               const float2* __restrict__ stats,
                                                               const float* __restrict__ loss_out,
                                                               const float* __restrict__ grad_out, float grad_scale,
                                                               float* __restrict__ grad_cost, float* __restrict__ grad_var,
                                                               int D, int HW, long long npix, DispVal dv, float lb,
                                                               float ub, float start, float end, float fc, float eps) {
  const long long pix = blockIdx.x * (long long)LOSS_BLOCK + threadIdx.x;
  if (pix >= npix) return;
  const long long b = pix / HW, r = pix - b * HW;
  Target tg;
  tg.setup(gt[pix], var ? var[pix] : var_scalar, lb, ub, start, end, D, dv, grad_var != nullptr);
  const float go = (grad_out ? grad_out[0] : 1.f) * grad_scale;
  const float k = go * tg.m1 / loss_out[1];
  const float2 st = stats[pix];
  const float* cp = cost + b * (long long)D * HW + r;
  float* gp = grad_cost + b * (long long)D * HW + r;
  double s1 = 0.0;
  for (int d = 0; d < D; ++d) {
    const float c = cp[(long long)d * HW];
    const float pd = tg.p(dv.v[d]);
    const float P = pd * tg.m2 + eps;
    const float pw = focal_pow(P, fc);
    __builtin_nontemporal_store(k * (st.y * __expf(c - st.x) - P * pw), gp + (long long)d * HW);   // streaming gradient
    if (grad_var) {
      const float da = fc == 0.f ? 1.f : pw + fc * P * pw / (1.f - P);
      const float dt = fabsf(dv.v[d] - tg.g) * tg.inv_v2;
      s1 += (double)((c - st.x) * da * pd) * (double)(dt - tg.e);
    }
  }
  if (grad_var) grad_var[pix] = (float)(-(double)k * (double)tg.m2 * s1);
}

// ConfidenceNllLoss (logits in, conf_nll_loss.py:52) and DispSmoothL1Loss (beta = 1): masked means over [B, 1, H, W].
// mode 0: -logsigmoid(x) * mask;  mode 1: smooth_l1(x - gt) * mask.
__device__ __forceinline__ float softplus_neg(float x) {   // -logsigmoid(x) = softplus(-x), stable
  return fmaxf(-x, 0.f) + log1pf(__expf(-fabsf(x)));
}
__global__ __launch_bounds__(LOSS_BLOCK) void map_loss_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gt,
                                                                  double* __restrict__ partial, long long n, float lb,
                                                                  float ub, int mode) {
  __shared__ double sm[4];
  const long long i = blockIdx.x * (long long)LOSS_BLOCK + threadIdx.x;
  double loss = 0.0, cnt = 0.0;
  if (i < n) {
    const float g = gt[i];
    if (g > lb && g < ub) {
      cnt = 1.0;
      if (mode == 0) {
        loss = (double)softplus_neg(x[i]);
      } else {
        const float d = fabsf(x[i] - g);
        loss = (double)(d < 1.f ? 0.5f * d * d : d - 0.5f);
      }
    }
  }
  const double bl = block_sum(loss, sm), bc = block_sum(cnt, sm);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = bl;
    partial[2 * blockIdx.x + 1] = bc;
  }
}
__global__ __launch_bounds__(LOSS_BLOCK) void map_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gt,
                                                                  const float* __restrict__ loss_out,
                                                                  const float* __restrict__ grad_out, float grad_scale,
                                                                  float* __restrict__ grad_x, long long n, float lb, float ub,
                                                                  int mode) {
  const long long i = blockIdx.x * (long long)LOSS_BLOCK + threadIdx.x;
  if (i >= n) return;
  const float g = gt[i];
  float gr = 0.f;
  if (g > lb && g < ub) {
    const float k = (grad_out ? grad_out[0] : 1.f) * grad_scale / loss_out[1];
    if (mode == 0) {
      gr = -k / (1.f + __expf(x[i]));                 // d/dx softplus(-x) = -sigmoid(-x)
    } else {
      const float d = x[i] - g;
      gr = k * (fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f));
    }
  }
  grad_x[i] = gr;
}

static int fill_dv(const float* host, int D, DispVal& dv) {
  if (!host || D <= 0 || D > DMB_MAX_DISP_SAMPLES) return fail(DMB_EINVAL, "loss: 1..256 disparity samples required");
  for (int i = 0; i < D; ++i) dv.v[i] = host[i];
  return DMB_OK;
}

}  // namespace dmb

using namespace dmb;

extern "C" long long dmb_loss_workspace_doubles(long long n_elements) { return 2 * ((n_elements + LOSS_BLOCK - 1) / LOSS_BLOCK); }

extern "C" int dmb_stereo_focal_loss_fwd_f32(const float* cost, const float* gt, const float* variance,
                                             float variance_scalar, const float* disp_sample_host, float* stats,
                                             double* workspace, float* loss_out, int B, int D, int H, int W, float lower,
                                             float upper, float start_disp, float end_disp, float focal_coefficient,
                                             void* stream) {
  if (!cost || !gt || !stats || !workspace || !loss_out || B <= 0 || H <= 0 || W <= 0)
    return fail(DMB_EINVAL, "stereo_focal_loss_fwd: bad argument");
  DispVal dv;
  if (int e = fill_dv(disp_sample_host, D, dv)) return e;
  const long long npix = (long long)B * H * W;
  const long long nblk = (npix + LOSS_BLOCK - 1) / LOSS_BLOCK;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "stereo_focal_loss_fwd: grid too large");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(focal_fwd_kernel, dim3((unsigned)nblk), dim3(LOSS_BLOCK), 0, st, cost, gt, variance, variance_scalar,
                     reinterpret_cast<float2*>(stats), workspace, D, H * W, npix, dv, lower, upper, start_disp, end_disp,
                     focal_coefficient, 1e-40f);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(LOSS_BLOCK), 0, st, workspace, (int)nblk, loss_out, 0, npix);
  return launch_status("stereo_focal_loss_fwd launch failed");
}

extern "C" int dmb_stereo_focal_loss_bwd_f32(const float* cost, const float* gt, const float* variance,
                                             float variance_scalar, const float* disp_sample_host, const float* stats,
                                             const float* loss_out, const float* grad_out, float grad_scale,
                                             float* grad_cost, float* grad_variance, int B, int D, int H, int W,
                                             float lower, float upper, float start_disp, float end_disp,
                                             float focal_coefficient, void* stream) {
  if (!cost || !gt || !stats || !loss_out || !grad_cost || B <= 0 || H <= 0 || W <= 0)
    return fail(DMB_EINVAL, "stereo_focal_loss_bwd: bad argument");
  if (grad_variance && !variance) return fail(DMB_EINVAL, "stereo_focal_loss_bwd: grad_variance needs a variance map");
  DispVal dv;
  if (int e = fill_dv(disp_sample_host, D, dv)) return e;
  const long long npix = (long long)B * H * W;
  const long long nblk = (npix + LOSS_BLOCK - 1) / LOSS_BLOCK;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "stereo_focal_loss_bwd: grid too large");
  hipLaunchKernelGGL(focal_bwd_kernel, dim3((unsigned)nblk), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, cost, gt, variance,
                     variance_scalar, reinterpret_cast<const float2*>(stats), loss_out, grad_out, grad_scale, grad_cost,
                     grad_variance, D, H * W, npix, dv, lower, upper, start_disp, end_disp, focal_coefficient, 1e-40f);
  return launch_status("stereo_focal_loss_bwd launch failed");
}

extern "C" int dmb_map_loss_fwd_f32(const float* x, const float* gt, double* workspace, float* loss_out, long long n,
                                    float lower, float upper, int mode, void* stream) {
  if (!x || !gt || !workspace || !loss_out || n <= 0 || (mode != 0 && mode != 1)) return fail(DMB_EINVAL, "map_loss_fwd: bad argument");
  const long long nblk = (n + LOSS_BLOCK - 1) / LOSS_BLOCK;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "map_loss_fwd: grid too large");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(map_loss_fwd_kernel, dim3((unsigned)nblk), dim3(LOSS_BLOCK), 0, st, x, gt, workspace, n, lower, upper, mode);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(LOSS_BLOCK), 0, st, workspace, (int)nblk, loss_out, 0, n);
  return launch_status("map_loss_fwd launch failed");
}

extern "C" int dmb_map_loss_bwd_f32(const float* x, const float* gt, const float* loss_out, const float* grad_out,
                                    float grad_scale, float* grad_x, long long n, float lower, float upper, int mode,
                                    void* stream) {
  if (!x || !gt || !loss_out || !grad_x || n <= 0 || (mode != 0 && mode != 1)) return fail(DMB_EINVAL, "map_loss_bwd: bad argument");
  const long long nblk = (n + LOSS_BLOCK - 1) / LOSS_BLOCK;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "map_loss_bwd: grid too large");
  hipLaunchKernelGGL(map_loss_bwd_kernel, dim3((unsigned)nblk), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, x, gt, loss_out,
                     grad_out, grad_scale, grad_x, n, lower, upper, mode);
  return launch_status("map_loss_bwd launch failed");
}
