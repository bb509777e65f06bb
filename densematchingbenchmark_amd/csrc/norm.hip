// Training-side BatchNorm(+residual, +ReLU) of the convolution units (second part): batch statistics,
// the normalising pass, and the two passes of the backward (channel sums, then the gradient w.r.t. the raw convolution
// output).  nn.BatchNorm3d / nn.BatchNorm2d in training mode as the reference's factories build them
// (dmb/modeling/stereo/layers/basic_layers.py:68-100,160-177; the skip adds of cost_processors/utils/hourglass.py:62-86).
//
//   forward   mean_c, var_c (biased) over (B, voxels);  invstd = 1/sqrt(var + eps);  scale = gamma*invstd,
//             shift = beta - mean*scale;  y = act(c*scale + shift (+ residual))      (same epilogue as the conv kernels)
//             running_mean += momentum*(mean - running_mean), running_var likewise with the unbiased variance
//   backward  dpre = dy * [unit output > 0];  dbeta = sum dpre;  dgamma = sum dpre * xhat,  xhat = (c - mean)*invstd
//             dc = scale * (dpre - dbeta/N - xhat*dgamma/N)          (eval mode: dc = scale * dpre)
//
// All four are HBM-bound single passes with 16-byte accesses; sums are FP64 per block, finished by one block in a fixed
// order (deterministic, no atomics).  Layout [B, C, S] with S = voxels (or pixels) per channel.
#include "dmb_common.h"

#pragma clang fp contract(off)

namespace dmb {

constexpr int NB = 256;   // threads per block

__device__ __forceinline__ double norm_block_sum(double v, double* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

// Slice s of channel ch: elements [lo, hi) of every batch item's channel plane.  Visit them as float4 where aligned.
template <class F4, class F1>
__device__ __forceinline__ void for_slice(long long S, int B, int C, int ch, int s, int nsplit, bool vec, F4 f4, F1 f1) {
  const long long per = ((S + nsplit - 1) / nsplit + 3) / 4 * 4;
  const long long lo = (long long)s * per, hi = lo + per < S ? lo + per : S;
  for (int b = 0; b < B; ++b) {
    const long long base = ((long long)b * C + ch) * S;
    if (vec) {
      for (long long i = lo + 4LL * threadIdx.x; i < hi; i += 4LL * NB) {
        if (i + 4 <= hi)
          f4(base + i);
        else
          for (long long j = i; j < hi; ++j) f1(base + j);
      }
    } else {
      for (long long i = lo + threadIdx.x; i < hi; i += NB) f1(base + i);
    }
  }
}

// ws[(ch * nsplit + s) * 2 + {0, 1}] = sum, sum of squares (about the channel's first element, for conditioning)
__global__ __launch_bounds__(NB) void bn_stats_kernel(const float* __restrict__ c, double* __restrict__ ws, int B, int C,
                                                      long long S, int nsplit, int vec) {
  __shared__ double sm[4];
  const int ch = blockIdx.y, s = blockIdx.x;
  const float pivot = c[(long long)ch * S];
  // FP32 lane partials over short runs, FP64 across runs
  double sum = 0.0, sq = 0.0;
  float ps = 0.f, pq = 0.f;
  int run = 0;
  auto add = [&](float v) {
    const float d = v - pivot;
    ps += d;
    pq = fmaf(d, d, pq);
  };
  auto flush = [&]() {
    sum += (double)ps;
    sq += (double)pq;
    ps = pq = 0.f;
    run = 0;
  };
  for_slice(S, B, C, ch, s, nsplit, vec != 0,
            [&](long long o) {
              const float4 v = *reinterpret_cast<const float4*>(c + o);
              add(v.x), add(v.y), add(v.z), add(v.w);
              if (++run == 64) flush();
            },
            [&](long long o) {
              add(c[o]);
              if (++run == 256) flush();
            });
  flush();
  sum = norm_block_sum(sum, sm);
  sq = norm_block_sum(sq, sm);
  if (threadIdx.x == 0) {
    ws[((long long)ch * nsplit + s) * 2] = sum;
    ws[((long long)ch * nsplit + s) * 2 + 1] = sq;
  }
}

__global__ void bn_stats_finalize_kernel(const float* __restrict__ c, const double* __restrict__ ws, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, float* __restrict__ running_mean,
                                         float* __restrict__ running_var, float momentum, float eps, float* __restrict__ mean_out,
                                         float* __restrict__ invstd_out, float* __restrict__ scale_out, float* __restrict__ shift_out,
                                         int B, int C, long long S, int nsplit) {
  // one wave per channel: the lanes walk the slices 64 apart (independent loads) and a fixed butterfly adds them -- one thread
  // walking up to 128 slices alone was a chain of dependent-latency loads, 10 us for a kernel that does nothing else
  const int ch = blockIdx.x, lane = threadIdx.x;
  double sum = 0.0, sq = 0.0;
  for (int s = lane; s < nsplit; s += 64) {
    sum += ws[((long long)ch * nsplit + s) * 2];
    sq += ws[((long long)ch * nsplit + s) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_down(sum, o, 64);
    sq += __shfl_down(sq, o, 64);
  }
  if (lane != 0) return;
  const double n = (double)B * (double)S;
  const double dm = sum / n;                      // mean - pivot
  double var = sq / n - dm * dm;
  if (var < 0.0) var = 0.0;
  const double mean = (double)c[(long long)ch * S] + dm;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[ch] : 1.f, bta = beta ? beta[ch] : 0.f;
  const float sc = g * invstd;
  mean_out[ch] = (float)mean;
  invstd_out[ch] = invstd;
  scale_out[ch] = sc;
  shift_out[ch] = bta - (float)mean * sc;
  if (running_mean) running_mean[ch] = running_mean[ch] + momentum * ((float)mean - running_mean[ch]);
  if (running_var) {
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    running_var[ch] = running_var[ch] + momentum * ((float)unbiased - running_var[ch]);
  }
}

// y = act(c*scale + shift (+ residual));  relu: 0 none, 1 after the residual add, 2 before it
__global__ __launch_bounds__(NB) void bn_act_kernel(const float* __restrict__ c, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, const float* __restrict__ res,
                                                    float* __restrict__ y, int C, long long S, long long total, int relu, int vec) {
  const float lo1 = relu == 1 ? 0.f : -INFINITY, lo2 = relu == 2 ? 0.f : -INFINITY;
  auto one = [&](float v, float r, float sc, float sh) { return fmaxf(fmaxf(fmaf(v, sc, sh), lo2) + r, lo1); };
  if (vec) {
    for (long long i = (blockIdx.x * (long long)NB + threadIdx.x) * 4; i < total; i += (long long)gridDim.x * NB * 4) {
      const int ch = (int)((i / S) % C);
      const float sc = scale[ch], sh = shift[ch];
      const float4 v = *reinterpret_cast<const float4*>(c + i);
      float4 r = {0.f, 0.f, 0.f, 0.f};
      if (res) r = *reinterpret_cast<const float4*>(res + i);
      float4 o = {one(v.x, r.x, sc, sh), one(v.y, r.y, sc, sh), one(v.z, r.z, sc, sh), one(v.w, r.w, sc, sh)};
      *reinterpret_cast<float4*>(y + i) = o;
    }
  } else {
    for (long long i = blockIdx.x * (long long)NB + threadIdx.x; i < total; i += (long long)gridDim.x * NB) {
      const int ch = (int)((i / S) % C);
      y[i] = one(c[i], res ? res[i] : 0.f, scale[ch], shift[ch]);
    }
  }
}

// Channel totals of the per-slice partial sums, in the order bn_stats_finalize_kernel adds them (lanes 64 slices apart, then the
// butterfly): a consuming kernel whose workgroups each re-create the totals of their channel from the
// same <= 2 KB of partials gets the same doubles everywhere -- and the finalising launch between the two passes goes away
// (round 6: 25 + 25 launches of 4.7 us in a PSMNet training step).
__device__ __forceinline__ void channel_totals(const double* __restrict__ ws, int ch, int nsplit, double* sm2, double& a, double& b) {
  if (nsplit > 256) {
    // thousands of partials (one per workgroup of a convolution whose epilogue summed its own output): all four waves walk them,
    // 256 apart (one wave alone was a chain of 48 dependent rounds: 16 us in front of a 66 us pass), combined in wave order
    __shared__ double wsum[4][2];
    double s0 = 0.0, s1 = 0.0;
    for (int s = threadIdx.x; s < nsplit; s += NB) {
      s0 += ws[((long long)ch * nsplit + s) * 2];
      s1 += ws[((long long)ch * nsplit + s) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s0 += __shfl_down(s0, o, 64);
      s1 += __shfl_down(s1, o, 64);
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6][0] = s0, wsum[threadIdx.x >> 6][1] = s1;
    __syncthreads();
    a = ((wsum[0][0] + wsum[1][0]) + wsum[2][0]) + wsum[3][0];
    b = ((wsum[0][1] + wsum[1][1]) + wsum[2][1]) + wsum[3][1];
    return;
  }
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int s = lane; s < nsplit; s += 64) {
      s0 += ws[((long long)ch * nsplit + s) * 2];
      s1 += ws[((long long)ch * nsplit + s) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s0 += __shfl_down(s0, o, 64);
      s1 += __shfl_down(s1, o, 64);
    }
    if (lane == 0) sm2[0] = s0, sm2[1] = s1;
  }
  __syncthreads();
  a = sm2[0];
  b = sm2[1];
}

// bn_stats_finalize_kernel + bn_act_kernel in one launch: workgroup (s, ch) finishes channel ch's statistics itself (the
// arithmetic of bn_stats_finalize_kernel, operation for operation) and normalises slice s of the channel; workgroup (0, ch)
// also writes the per-channel outputs and updates the running buffers, workgroup (0, 0) counts the batch.
__global__ __launch_bounds__(NB) void bn_act_train_kernel(const float* __restrict__ c, const double* __restrict__ ws,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          long long* __restrict__ num_batches_tracked, float momentum, float eps,
                                                          float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                          float* __restrict__ scale_out, float* __restrict__ shift_out,
                                                          const float* __restrict__ res, float* __restrict__ y, int B, int C,
                                                          long long S, int nsplit, int nact, int relu, int vec, int pivot_free) {
  __shared__ double tot[2];
  const int ch = blockIdx.y, s = blockIdx.x;
  double sum, sq;
  channel_totals(ws, ch, nsplit, tot, sum, sq);
  const double n = (double)B * (double)S;
  const double dm = sum / n;                      // mean - pivot
  double var = sq / n - dm * dm;
  if (var < 0.0) var = 0.0;
  // pivot_free: the partial sums come from a convolution's epilogue (conv3d_s1_kernel, STATS: FP64 sums of the raw values and
  // their squares, no pivot -- E[x^2] - mean^2 in FP64 loses nothing an FP32 tensor could show until |mean| / std ~ 1e4)
  const double mean = (pivot_free ? 0.0 : (double)c[(long long)ch * S]) + dm;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[ch] : 1.f, bta = beta ? beta[ch] : 0.f;
  const float sc = g * invstd;
  const float sh = bta - (float)mean * sc;
  if (s == 0 && threadIdx.x == 0) {
    mean_out[ch] = (float)mean;
    invstd_out[ch] = invstd;
    scale_out[ch] = sc;
    shift_out[ch] = sh;
    if (running_mean) running_mean[ch] = running_mean[ch] + momentum * ((float)mean - running_mean[ch]);
    if (running_var) {
      const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
      running_var[ch] = running_var[ch] + momentum * ((float)unbiased - running_var[ch]);
    }
    if (ch == 0 && num_batches_tracked) *num_batches_tracked += 1;
  }
  const float lo1 = relu == 1 ? 0.f : -INFINITY, lo2 = relu == 2 ? 0.f : -INFINITY;
  auto one = [&](float v, float r) { return fmaxf(fmaxf(fmaf(v, sc, sh), lo2) + r, lo1); };
  for_slice(S, B, C, ch, s, nact, vec != 0,
            [&](long long o) {
              const float4 v = *reinterpret_cast<const float4*>(c + o);
              float4 r = {0.f, 0.f, 0.f, 0.f};
              if (res) r = *reinterpret_cast<const float4*>(res + o);
              float4 q = {one(v.x, r.x), one(v.y, r.y), one(v.z, r.z), one(v.w, r.w)};
              *reinterpret_cast<float4*>(y + o) = q;
            },
            [&](long long o) { y[o] = one(c[o], res ? res[o] : 0.f); });
}

// gradient entering the normalisation: dy masked by the unit's ReLU.
//   relu 1 (activation after the skip add): the mask is the unit's output y > 0
//   relu 2 (activation before the skip add): the mask is the normalised value c*scale + shift > 0
__device__ __forceinline__ float dpre_of(float dy, float cv, float yv, float sc, float sh, int relu) {
  if (relu == 1) return yv > 0.f ? dy : 0.f;
  if (relu == 2) return fmaf(cv, sc, sh) > 0.f ? dy : 0.f;
  return dy;
}

// ws[(ch * nsplit + s) * 2 + {0, 1}] = sum dpre, sum dpre * xhat
__global__ __launch_bounds__(NB) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ c,
                                                           const float* __restrict__ y, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, double* __restrict__ ws, int B, int C,
                                                           long long S, int nsplit, int relu, int vec) {
  __shared__ double sm[4];
  const int ch = blockIdx.y, s = blockIdx.x;
  const float sc = scale[ch], sh = shift[ch], mu = mean[ch], is = invstd[ch];
  double s1 = 0.0, s2 = 0.0;
  float p1 = 0.f, p2 = 0.f;
  int run = 0;
  auto add = [&](float g, float cv, float yv) {
    const float d = dpre_of(g, cv, yv, sc, sh, relu);
    p1 += d;
    p2 = fmaf(d, (cv - mu) * is, p2);
  };
  auto flush = [&]() {
    s1 += (double)p1;
    s2 += (double)p2;
    p1 = p2 = 0.f;
    run = 0;
  };
  for_slice(S, B, C, ch, s, nsplit, vec != 0,
            [&](long long o) {
              const float4 g = *reinterpret_cast<const float4*>(dy + o);
              const float4 cv = *reinterpret_cast<const float4*>(c + o);
              float4 yv = {0.f, 0.f, 0.f, 0.f};
              if (relu == 1) yv = *reinterpret_cast<const float4*>(y + o);
              add(g.x, cv.x, yv.x), add(g.y, cv.y, yv.y), add(g.z, cv.z, yv.z), add(g.w, cv.w, yv.w);
              if (++run == 64) flush();
            },
            [&](long long o) {
              add(dy[o], c[o], relu == 1 ? y[o] : 0.f);
              if (++run == 256) flush();
            });
  flush();
  s1 = norm_block_sum(s1, sm);
  s2 = norm_block_sum(s2, sm);
  if (threadIdx.x == 0) {
    ws[((long long)ch * nsplit + s) * 2] = s1;
    ws[((long long)ch * nsplit + s) * 2 + 1] = s2;
  }
}

// dc = scale*(dpre - dbeta/N - xhat*dgamma/N) (training) or scale*dpre (eval).  The channel sums are finished by every workgroup
// itself (channel_totals: rounds 1-5 had a finalising launch between the two passes), and the skip branch's gradient is an
// ACCUMULATING output: dres = (what flows into the skip branch) + dres_acc, where dres_acc is the gradient the skip operand has
// already collected from its other consumers -- the addition torch.autograd would launch on its own (three tensor passes) costs
// one extra read here.  Workgroup (0, ch) writes dgamma / dbeta.
__global__ __launch_bounds__(NB) void bn_bwd_apply_fin_kernel(const float* __restrict__ dy, const float* __restrict__ c,
                                                              const float* __restrict__ y, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const double* __restrict__ ws,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ dc, float* __restrict__ dres,
                                                              const float* __restrict__ dres_acc, int B, int C, long long S,
                                                              int nsplit, int nact, float inv_n, int relu, int training, int vec) {
  __shared__ double tot[2];
  const int ch = blockIdx.y, s = blockIdx.x;
  double s1, s2;
  channel_totals(ws, ch, nsplit, tot, s1, s2);
  const float db = (float)s1, dg = (float)s2;
  if (s == 0 && threadIdx.x == 0) {
    dbeta[ch] = db;
    dgamma[ch] = dg;
  }
  const float sc = scale[ch], sh = shift[ch], mu = mean[ch], is = invstd[ch];
  auto one = [&](float g, float cv, float yv, float av, float& dr) {
    const float d = dpre_of(g, cv, yv, sc, sh, relu);
    dr = (relu == 1 ? d : g) + av;   // what flows into the skip branch (+ what it already holds)
    if (!training) return sc * d;
    const float xh = (cv - mu) * is;
    return sc * (d - db * inv_n - xh * (dg * inv_n));
  };
  for_slice(S, B, C, ch, s, nact, vec != 0,
            [&](long long o) {
              const float4 g = *reinterpret_cast<const float4*>(dy + o);
              const float4 cv = *reinterpret_cast<const float4*>(c + o);
              float4 yv = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
              if (relu == 1) yv = *reinterpret_cast<const float4*>(y + o);
              if (dres_acc) av = *reinterpret_cast<const float4*>(dres_acc + o);
              float4 q, r;
              q.x = one(g.x, cv.x, yv.x, av.x, r.x);
              q.y = one(g.y, cv.y, yv.y, av.y, r.y);
              q.z = one(g.z, cv.z, yv.z, av.z, r.z);
              q.w = one(g.w, cv.w, yv.w, av.w, r.w);
              *reinterpret_cast<float4*>(dc + o) = q;
              if (dres) *reinterpret_cast<float4*>(dres + o) = r;
            },
            [&](long long o) {
              float r;
              dc[o] = one(dy[o], c[o], relu == 1 ? y[o] : 0.f, dres_acc ? dres_acc[o] : 0.f, r);
              if (dres) dres[o] = r;
            });
}

// out[c] = sum_{b, s} a[b, c, s] * g[b, 0, s]: the weight gradient of a 1x1 convolution with one output channel
// (the second layer of AcfNet's confidence heads, cmn/cmn.py:30).  ws[ch * nsplit + s] partial sums (FP64).
__global__ __launch_bounds__(NB) void channel_dot_kernel(const float* __restrict__ a, const float* __restrict__ g,
                                                         double* __restrict__ ws, int B, int C, long long S, int nsplit, int vec) {
  __shared__ double sm[4];
  const int ch = blockIdx.y, s = blockIdx.x;
  double tot = 0.0;
  float p = 0.f;
  int run = 0;
  auto flush = [&]() {
    tot += (double)p;
    p = 0.f;
    run = 0;
  };
  for_slice(S, B, C, ch, s, nsplit, vec != 0,
            [&](long long o) {
              // o indexes a ([B, C, S]); the matching element of g ([B, 1, S]) is (b, s) = (o / (C S), o % S)
              const long long b = o / ((long long)C * S), sp = o % S;
              const float4 av = *reinterpret_cast<const float4*>(a + o);
              const float4 gv = *reinterpret_cast<const float4*>(g + b * S + sp);
              p = fmaf(av.x, gv.x, p), p = fmaf(av.y, gv.y, p), p = fmaf(av.z, gv.z, p), p = fmaf(av.w, gv.w, p);
              if (++run == 64) flush();
            },
            [&](long long o) {
              const long long b = o / ((long long)C * S), sp = o % S;
              p = fmaf(a[o], g[b * S + sp], p);
              if (++run == 256) flush();
            });
  flush();
  tot = norm_block_sum(tot, sm);
  if (threadIdx.x == 0) ws[(long long)ch * nsplit + s] = tot;
}

__global__ void channel_dot_finalize_kernel(const double* __restrict__ ws, float* __restrict__ out, int C, int nsplit) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= C) return;
  double s = 0.0;
  for (int i = 0; i < nsplit; ++i) s += ws[(long long)ch * nsplit + i];
  out[ch] = (float)s;
}

static int bn_nsplit(int C, long long S) {
  long long n = 4096 / (C > 0 ? C : 1);
  if (n < 1) n = 1;
  const long long most = (S + 1023) / 1024;   // at least ~1024 elements per slice and batch item
  if (n > most) n = most;
  return (int)(n < 1 ? 1 : n);
}

// slices per channel of the consuming (normalising) kernels: ~1024 workgroups, at least ~2048 elements per slice and batch item
static int bn_nact(int C, long long S) {
  long long n = 1024 / (C > 0 ? C : 1);
  if (n < 1) n = 1;
  const long long most = (S + 2047) / 2048;
  if (n > most) n = most;
  return (int)(n < 1 ? 1 : n);
}

static int elementwise_blocks(long long total, int per_thread) {
  long long b = (total + (long long)NB * per_thread - 1) / ((long long)NB * per_thread);
  if (b > 16384) b = 16384;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace dmb

using namespace dmb;

extern "C" long long dmb_bn_workspace_doubles(int C, long long S) {
  if (C <= 0 || S <= 0) return 0;
  return 2LL * C * bn_nsplit(C, S);
}

extern "C" int dmb_bn_train_stats_f32(const float* c, const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, float momentum, float eps, float* mean_out, float* invstd_out,
                                      float* scale_out, float* shift_out, double* workspace, int B, int C, long long S,
                                      void* stream) {
  if (!c || !mean_out || !invstd_out || !scale_out || !shift_out || !workspace || B <= 0 || C <= 0 || S <= 0)
    return fail(DMB_EINVAL, "bn_train_stats: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int nsplit = bn_nsplit(C, S);
  const int vec = S % 4 == 0 && ((uintptr_t)c & 15) == 0;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nsplit, C), dim3(NB), 0, st, c, workspace, B, C, S, nsplit, vec);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C), dim3(64), 0, st, c, workspace, gamma, beta, running_mean,
                     running_var, momentum, eps, mean_out, invstd_out, scale_out, shift_out, B, C, S, nsplit);
  return launch_status("bn_train_stats launch failed");
}

extern "C" int dmb_bn_train_fwd_f32(const float* c, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, long long* num_batches_tracked, float momentum, float eps,
                                    float* mean_out, float* invstd_out, float* scale_out, float* shift_out, const float* residual,
                                    float* y, double* workspace, int B, int C, long long S, int relu, void* stream) {
  if (!c || !mean_out || !invstd_out || !scale_out || !shift_out || !y || !workspace || B <= 0 || C <= 0 || S <= 0 || relu < 0 || relu > 2)
    return fail(DMB_EINVAL, "bn_train_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int nsplit = bn_nsplit(C, S), nact = bn_nact(C, S);
  const int vec = S % 4 == 0 && (((uintptr_t)c | (uintptr_t)y | (uintptr_t)residual) & 15) == 0;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nsplit, C), dim3(NB), 0, st, c, workspace, B, C, S, nsplit, vec);
  hipLaunchKernelGGL(bn_act_train_kernel, dim3(nact, C), dim3(NB), 0, st, c, workspace, gamma, beta, running_mean, running_var,
                     num_batches_tracked, momentum, eps, mean_out, invstd_out, scale_out, shift_out, residual, y, B, C, S, nsplit, nact,
                     relu, vec, 0);
  return launch_status("bn_train_fwd launch failed");
}

extern "C" int dmb_bn_train_act_f32(const float* c, const double* partials, int nparts, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                                    float* mean_out, float* invstd_out, float* scale_out, float* shift_out, const float* residual,
                                    float* y, int B, int C, long long S, int relu, void* stream) {
  if (!c || !partials || nparts <= 0 || !mean_out || !invstd_out || !scale_out || !shift_out || !y || B <= 0 || C <= 0 || S <= 0 ||
      relu < 0 || relu > 2)
    return fail(DMB_EINVAL, "bn_train_act: bad argument");
  const int nact = bn_nact(C, S);
  const int vec = S % 4 == 0 && (((uintptr_t)c | (uintptr_t)y | (uintptr_t)residual) & 15) == 0;
  hipLaunchKernelGGL(bn_act_train_kernel, dim3(nact, C), dim3(NB), 0, (hipStream_t)stream, c, partials, gamma, beta, running_mean,
                     running_var, num_batches_tracked, momentum, eps, mean_out, invstd_out, scale_out, shift_out, residual, y, B, C, S,
                     nparts, nact, relu, vec, 1);
  return launch_status("bn_train_act launch failed");
}

extern "C" int dmb_bn_act_f32(const float* c, const float* scale, const float* shift, const float* residual, float* y, int B,
                              int C, long long S, int relu, void* stream) {
  if (!c || !scale || !shift || !y || B <= 0 || C <= 0 || S <= 0 || relu < 0 || relu > 2) return fail(DMB_EINVAL, "bn_act: bad argument");
  const long long total = (long long)B * C * S;
  const int vec = S % 4 == 0 && (((uintptr_t)c | (uintptr_t)y | (uintptr_t)residual) & 15) == 0;
  hipLaunchKernelGGL(bn_act_kernel, dim3(elementwise_blocks(total, vec ? 8 : 2)), dim3(NB), 0, (hipStream_t)stream, c, scale, shift,
                     residual, y, C, S, total, relu, vec);
  return launch_status("bn_act launch failed");
}

extern "C" int dmb_bn_act_bwd_f32(const float* dy, const float* c, const float* y, const float* scale, const float* shift,
                                  const float* mean, const float* invstd, double* workspace, float* dgamma, float* dbeta,
                                  float* dc, float* dres, const float* dres_acc, int B, int C, long long S, int relu, int training,
                                  void* stream) {
  if (!dy || !c || !scale || !shift || !mean || !invstd || !workspace || !dgamma || !dbeta || !dc || B <= 0 || C <= 0 || S <= 0 ||
      relu < 0 || relu > 2 || (relu == 1 && !y) || (dres_acc && !dres))
    return fail(DMB_EINVAL, "bn_act_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int nsplit = bn_nsplit(C, S), nact = bn_nact(C, S);
  const int vec = S % 4 == 0 && (((uintptr_t)dy | (uintptr_t)c | (uintptr_t)y | (uintptr_t)dc | (uintptr_t)dres | (uintptr_t)dres_acc) & 15) == 0;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nsplit, C), dim3(NB), 0, st, dy, c, y, scale, shift, mean, invstd, workspace, B, C, S,
                     nsplit, relu, vec);
  hipLaunchKernelGGL(bn_bwd_apply_fin_kernel, dim3(nact, C), dim3(NB), 0, st, dy, c, y, scale, shift, mean, invstd, workspace, dgamma,
                     dbeta, dc, dres, dres_acc, B, C, S, nsplit, nact, (float)(1.0 / ((double)B * (double)S)), relu, training, vec);
  return launch_status("bn_act_bwd launch failed");
}

extern "C" int dmb_channel_dot_f32(const float* a, const float* g, double* workspace, float* out, int B, int C, long long S,
                                   void* stream) {
  if (!a || !g || !workspace || !out || B <= 0 || C <= 0 || S <= 0) return fail(DMB_EINVAL, "channel_dot: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int nsplit = bn_nsplit(C, S);
  const int vec = S % 4 == 0 && (((uintptr_t)a | (uintptr_t)g) & 15) == 0;
  hipLaunchKernelGGL(channel_dot_kernel, dim3(nsplit, C), dim3(NB), 0, st, a, g, workspace, B, C, S, nsplit, vec);
  hipLaunchKernelGGL(channel_dot_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, workspace, out, C, nsplit);
  return launch_status("channel_dot launch failed");
}
