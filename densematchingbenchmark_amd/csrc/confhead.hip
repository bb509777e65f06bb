// AcfNet confidence head (dmb/modeling/stereo/cmn/cmn.py:10-36,57-69) as one fused kernel:
//   h    = relu(BN(conv2d_3x3(cost)))      cost: [B, D, H, W] (the disparity axis is the channel axis), Cm maps
//   conf = sigmoid(sum_m h_m * w2[m])      1x1 conv + sigmoid, fused into the epilogue (h never leaves registers)
// The 3x3 convolution is an FP32 implicit GEMM on the matrix cores (M = Cm, N = pixels, K = D * 9) with the same
// structure as conv3d.hip: LDS-DMA staged, double-buffered chunks of CH_CK disparity planes (input tile + the
// chunk's prepacked weight fragments), B fragments read along the flattened padded (y, x) plane, A/B fragments
// register double-buffered one k-step ahead of the MFMAs.  346.7 GFLOP/pair for AcfNet's three heads: MFMA-bound.
#include "dmb_common.h"

namespace dmb {

constexpr int CH_CK = 8;  // disparity planes (K channels) staged per chunk

// wp[((kp * 9 + tap) * NTT + nt) * 64 + lane] = w1[m = nt*32 + (lane & 31)][d = 2*kp + (lane >> 5)][tap], zero padded
__global__ void pack_conf_kernel(const float* __restrict__ w1, float* __restrict__ wp, int Cm, int D, int NTT, int Dpad) {
  const long long total = (long long)(Dpad / 2) * 9 * NTT * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    long long r = i >> 6;
    const int nt = (int)(r % NTT);
    r /= NTT;
    const int tap = (int)(r % 9);
    const int kp = (int)(r / 9);
    const int m = nt * 32 + (lane & 31);
    const int d = 2 * kp + (lane >> 5);
    wp[i] = (m < Cm && d < D) ? w1[((size_t)m * D + d) * 9 + tap] : 0.f;
  }
}

template <int NTT>
struct CHCfg {
  static constexpr int WN = NTT;           // one 32-map row tile per wave column
  static constexpr int WY = 4 / WN;
  static constexpr int RY = 4;             // output rows per wave
  static constexpr int TY = RY * WY;
  static constexpr int TX = 60;
  static constexpr int P = TX + 2;
  static constexpr int ROWS = TY + 2;
  static constexpr int MT = (RY * P + 31) / 32;
  static constexpr int CH_STRIDE = ROWS * P + 36;
  static constexpr int NK = (CH_CK / 2) * 9;
  static constexpr int IN_FLOATS = CH_CK * CH_STRIDE;
  static constexpr int W_FLOATS = NK * NTT * 64;
  static constexpr int BUF_FLOATS = IN_FLOATS + W_FLOATS;
  static constexpr int RED_FLOATS = 2 * TY * P;                      // cross-wave reduction scratch
  static constexpr int LDS_FLOATS = 2 * BUF_FLOATS;                  // the reduction scratch aliases buffer 0
  static_assert(RED_FLOATS <= BUF_FLOATS, "reduction scratch fits one buffer");
  static_assert(IN_FLOATS % 4 == 0 && W_FLOATS % 4 == 0, "16-byte aligned weight region");
  static_assert((CH_CK * ROWS) % 4 == 0, "rows are dealt evenly to the 4 waves");
};

template <int NTT>
__global__ __launch_bounds__(256, 2) void conf_head_kernel(const float* __restrict__ cost, const float* __restrict__ wp,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ w2, float* __restrict__ conf, int D,
                                                           int Cm, int H, int W, int ntx, int nty) {
  using C = CHCfg<NTT>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* red = lds;  // used only after the last chunk's barrier, when the tile buffers are dead
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % ntx;
  t /= ntx;
  const int ty = t % nty;
  const int b = t / nty;
  const int x0 = tx * C::TX, y0 = ty * C::TY;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int wy = wave / C::WN, wn = wave % C::WN;
  const unsigned HW = (unsigned)H * W;
  const float* cb = cost + (size_t)b * D * HW;
  const int Dpad = cdiv(D, CH_CK) * CH_CK;

  f32x16 acc[C::MT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  // ---- staging (LDS-DMA): rows of the haloed tile dealt to the waves in contiguous runs; planes past D and the
  // zero padding come from the buffer bounds check.
  constexpr int RPW = CH_CK * C::ROWS / 4;  // rows per wave per chunk
  constexpr int WV4 = (C::W_FLOATS / 4 + 255) / 256;
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(cb, (unsigned)D * HW * 4u);
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(wp, (unsigned)((Dpad / 2) * 9 * NTT * 64) * 4u);
  const int gx = x0 - 1 + lane;
  const unsigned xvoff = (lane < C::P && gx >= 0 && gx < W) ? (unsigned)gx * 4u : DMA_OOB;
  auto stage = [&](int c0, float* buf) {
    if (lane < C::P) {
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const int rid = wave * RPW + q, cl = rid / C::ROWS, yy = rid - cl * C::ROWS;
        const int gy = y0 - 1 + yy;
        const bool ok = c0 + cl < D && gy >= 0 && gy < H;
        dma4(xrs, ok ? xvoff : DMA_OOB, ok ? ((unsigned)(c0 + cl) * HW + (unsigned)gy * W) * 4u : 0u,
             buf + cl * C::CH_STRIDE + yy * C::P);
      }
    }
#pragma unroll
    for (int i = 0; i < WV4; ++i) {
      const int q4 = i * 256 + threadIdx.x;
      if (q4 < C::W_FLOATS / 4)
        dma16(wrs, (unsigned)q4 * 16u, (unsigned)(c0 / 2) * (9 * NTT * 64 * 4), buf + C::IN_FLOATS + (i * 256 + wave * 64) * 4);
    }
  };

  const int NC = Dpad / CH_CK;
  stage(0, lds);
  __syncthreads();
  for (int ci = 0; ci < NC; ++ci) {
    const float* cur = lds + (ci & 1) * C::BUF_FLOATS;
    if (ci + 1 < NC) stage((ci + 1) * CH_CK, lds + ((ci + 1) & 1) * C::BUF_FLOATS);
    const float* abase = cur + C::IN_FLOATS + wn * 64 + lane;
    const float* bbase = cur + h * C::CH_STRIDE + (wy * C::RY) * C::P + j;
    float af[2], bf[2][C::MT];
    auto load_frag = [&](int ks, float& a, float (&bq)[C::MT]) {
      const int cp = ks / 9, tap = ks % 9;
      const int dy = tap / 3, dx = tap % 3;
      a = abase[ks * NTT * 64];
      const float* bp = bbase + 2 * cp * C::CH_STRIDE + dy * C::P + dx;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) bq[mt] = bp[mt * 32];
    };
    load_frag(0, af[0], bf[0]);
#pragma unroll
    for (int ks = 0; ks < C::NK; ++ks) {
      if (ks + 1 < C::NK) load_frag(ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) acc[mt] = DMB_MFMA(af[ks & 1], bf[ks & 1][mt], acc[mt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // epilogue: BN + ReLU per map, dot with w2 over the maps held by this lane, then reduce over lane halves
  // (h) with a cross-lane swap and over wave columns (wn) through LDS.
  float sc[16], sh[16], wv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = wn * 32 + cd_row(r, h);
    const bool ok = m < Cm;
    sc[r] = ok ? scale[m] : 0.f;
    sh[r] = ok ? shift[m] : 0.f;
    wv[r] = ok ? w2[m] : 0.f;
  }
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt) {
    float part = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) part = fmaf(fmaxf(fmaf(acc[mt][r], sc[r], sh[r]), 0.f), wv[r], part);
    part += __shfl_xor(part, 32, 64);
    const int m = mt * 32 + j;
    if (h == 0 && m < C::RY * C::P) red[(wn * C::TY + wy * C::RY) * C::P + m] = part;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C::TY * C::P; i += 256) {
    const int ly = i / C::P, lx = i % C::P;
    const int gy = y0 + ly, gxo = x0 + lx;
    if (lx < C::TX && gy < H && gxo < W) {
      float v = red[i];
      if (C::WN == 2) v += red[C::TY * C::P + i];
      conf[(size_t)b * HW + (size_t)gy * W + gxo] = 1.f / (1.f + __expf(-v));
    }
  }
}

}  // namespace dmb

using namespace dmb;

namespace dmb {

// ------------------------------------------------------------------------------------------------------------------
// The confidence head on a cost volume that is the learned 4x up-sampling of a quarter-resolution volume
// (aggregators/AcfNet.py:55-57,81-83: ConvTranspose3d(1, 1, 8, 4, 2) applied to c [B, Dq, Hq, Wq]).  The head's 3x3
// convolution over the up-sampled volume (D = 4 Dq channels, cmn.py:21-27) composed with that up-sampling is, for each of the
// 16 output phases (Y mod 4, X mod 4), a 3x3 convolution of c itself with Dq input channels -- 4x fewer multiplications,
// on the 2-D convolution kernel (conv2d.hip) with BatchNorm + ReLU in its epilogue:
//     hq[(phase, m), Y', X'] = relu(bn(sum_{z, ty, tx} K[(phase, m), z, ty, tx] * c[z, Y' + ty - 1, X' + tx - 1]))
// The two kernels here finish the job: conf_gather = 1x1 convolution + sigmoid per full-resolution pixel; conf_ring =
// the outermost pixel ring evaluated directly on the up-sampled volume (there the head's zero padding of that volume is not
// what the composed form sees: it continues the up-sampling past the image border).
// ------------------------------------------------------------------------------------------------------------------
// hq: [B, 16 * M, Hq, Wq] (channel = (by * 4 + bx) * M + m) -> conf [B, 1, 4 Hq, 4 Wq]
__global__ __launch_bounds__(256) void conf_gather_kernel(const float* __restrict__ hq, const float* __restrict__ w2,
                                                          float* __restrict__ conf, int B, int M, int Hq, int Wq) {
  const long long total = (long long)B * 16 * Hq * Wq;
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= total) return;
  const int xq = (int)(i % Wq);
  long long r = i / Wq;
  const int yq = (int)(r % Hq); r /= Hq;
  const int ph = (int)(r % 16), b = (int)(r / 16);
  const size_t plane = (size_t)Hq * Wq;
  const float* h = hq + ((size_t)b * 16 + ph) * M * plane + (size_t)yq * Wq + xq;
  float acc = 0.f;
  for (int m = 0; m < M; ++m) acc = fmaf(h[(size_t)m * plane], w2[m], acc);
  const int Y = 4 * yq + (ph >> 2), X = 4 * xq + (ph & 3);
  conf[((size_t)b * 4 * Hq + Y) * (4 * Wq) + X] = 1.f / (1.f + __expf(-acc));
}

// Ring pixels straight from the up-sampled volume.  Lane = ring pixel (64 per workgroup), the four waves split the D
// planes; a lane keeps all M = 64 hidden sums in registers and the tap's 64 weights arrive as wave-uniform (scalar) loads
// from w1t [D, 3, 3, 64] -- one vector load feeds 64 fmas.  Partial sums meet in LDS in a fixed order (reproducible).
constexpr int RING_M = 64;
constexpr int RING_NW = 16;  // waves per workgroup = slices of the D planes: the ring is only ~3000 pixels per image, i.e. one
                             // wave per SIMD at best; a lane's chain of D * 9 * 64 fmas is what the kernel's time is made of
__global__ __launch_bounds__(64 * RING_NW) void conf_ring_kernel(const float* __restrict__ cost, const float* __restrict__ w1t,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ w2, float* __restrict__ conf, int B,
                                                                 int D, int H, int W) {
  extern __shared__ float ring_lds[];   // part[RING_NW][RING_M / 2][64]
  float (*part)[RING_M / 2][64] = reinterpret_cast<float (*)[RING_M / 2][64]>(ring_lds);
  const int ring = 2 * W + 2 * (H - 2);
  const int nblk = cdiv(ring, 64);
  const int b = blockIdx.x / nblk;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  int p = (blockIdx.x % nblk) * 64 + lane;
  const bool live = p < ring;
  if (!live) p = ring - 1;
  int py, px;
  if (p < W) { py = 0; px = p; }
  else if (p < 2 * W) { py = H - 1; px = p - W; }
  else if (p < 2 * W + (H - 2)) { py = p - 2 * W + 1; px = 0; }
  else { py = p - 2 * W - (H - 2) + 1; px = W - 1; }
  float acc[RING_M];
#pragma unroll
  for (int m = 0; m < RING_M; ++m) acc[m] = 0.f;
  const size_t HW = (size_t)H * W;
  const float* cb = cost + (size_t)b * D * HW;
  const int d0 = wave * D / RING_NW, d1 = (wave + 1) * D / RING_NW;
  for (int d = d0; d < d1; ++d)
#pragma unroll 1   // (one tap's 64 weights fill the scalar registers; unrolled, 9 x 64 of them spilled)
    for (int t = 0; t < 9; ++t) {
      const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
      const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? cb[(size_t)d * HW + (size_t)yy * W + xx] : 0.f;
      const float* wv = w1t + ((size_t)d * 9 + t) * RING_M;   // wave-uniform: scalar loads
#pragma unroll
      for (int m = 0; m < RING_M; ++m) acc[m] = fmaf(v, wv[m], acc[m]);
    }
  // the partial sums meet in LDS, half of the hidden channels at a time (16 waves x 64 channels x 64 lanes would not fit)
  float logit = 0.f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();
#pragma unroll
    for (int m = 0; m < RING_M / 2; ++m) part[wave][m][lane] = acc[half * (RING_M / 2) + m];
    __syncthreads();
    if (wave == 0) {
      for (int m = 0; m < RING_M / 2; ++m) {
        float s = part[0][m][lane];
#pragma unroll
        for (int w = 1; w < RING_NW; ++w) s += part[w][m][lane];   // fixed order: reproducible
        const int mm = half * (RING_M / 2) + m;
        logit = fmaf(fmaxf(fmaf(s, scale[mm], shift[mm]), 0.f), w2[mm], logit);
      }
    }
  }
  if (wave == 0 && live) conf[((size_t)b * H + py) * W + px] = 1.f / (1.f + __expf(-logit));
}

}  // namespace dmb

using namespace dmb;

extern "C" int dmb_conf_gather_f32(const float* hq, const float* w2, float* conf, int B, int M, int Hq, int Wq, void* stream) {
  if (!hq || !w2 || !conf || B <= 0 || M <= 0 || Hq <= 0 || Wq <= 0) return fail(DMB_EINVAL, "conf_gather: bad argument");
  const long long total = (long long)B * 16 * Hq * Wq;
  hipLaunchKernelGGL(conf_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, hq, w2, conf, B,
                     M, Hq, Wq);
  return launch_status("conf_gather launch failed");
}

extern "C" int dmb_conf_ring_f32(const float* cost, const float* w1t, const float* scale, const float* shift, const float* w2,
                                 float* conf, int B, int D, int M, int H, int W, void* stream) {
  if (!cost || !w1t || !scale || !shift || !w2 || !conf || B <= 0 || D <= 0 || M != RING_M || H < 3 || W < 3)
    return fail(DMB_EINVAL, "conf_ring: bad argument (64 hidden channels)");
  const int ring = 2 * W + 2 * (H - 2);
  const size_t lds = (size_t)RING_NW * (RING_M / 2) * 64 * sizeof(float);
  DMB_ENSURE_LDS((&conf_ring_kernel), (size_t)(lds));
  hipLaunchKernelGGL(conf_ring_kernel, dim3((unsigned)(B * cdiv(ring, 64))), dim3(64 * RING_NW), lds, (hipStream_t)stream, cost,
                     w1t, scale, shift, w2, conf, B, D, H, W);
  return launch_status("conf_ring launch failed");
}

extern "C" long long dmb_conf_head_packed_floats(int Cm, int D) {
  if (Cm <= 0 || D <= 0) return 0;
  const int NTT = cdiv(Cm, 32), Dpad = cdiv(D, CH_CK) * CH_CK;
  return (long long)(Dpad / 2) * 9 * NTT * 64;
}

extern "C" int dmb_conf_head_pack_weights_f32(const float* w1, float* w1pack, int Cm, int D, void* stream) {
  if (!w1 || !w1pack || Cm <= 0 || D <= 0) return fail(DMB_EINVAL, "conf_head_pack: bad argument");
  const int NTT = cdiv(Cm, 32), Dpad = cdiv(D, CH_CK) * CH_CK;
  if (NTT > 2) return fail(DMB_EUNSUPPORTED, "conf_head: more than 64 intermediate maps");
  hipLaunchKernelGGL(pack_conf_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, w1, w1pack, Cm, D, NTT, Dpad);
  return launch_status("conf_head_pack launch failed");
}

template <int NTT>
static int launch_conf(const float* cost, const float* wp, const float* scale, const float* shift, const float* w2,
                       float* conf, int B, int D, int Cm, int H, int W, hipStream_t st) {
  using C = CHCfg<NTT>;
  const int ntx = cdiv(W, C::TX), nty = cdiv(H, C::TY);
  const long long nblk = (long long)B * ntx * nty;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conf_head: grid too large");
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  DMB_ENSURE_LDS((&conf_head_kernel<NTT>), (size_t)(lds));
  hipLaunchKernelGGL((conf_head_kernel<NTT>), dim3((unsigned)nblk), dim3(256), lds, st, cost, wp, scale, shift, w2, conf,
                     D, Cm, H, W, ntx, nty);
  return launch_status("conf_head launch failed");
}

extern "C" int dmb_conf_head_f32(const float* cost, const float* w1pack, const float* scale, const float* shift,
                                 const float* w2, float* conf, int B, int D, int Cm, int H, int W, void* stream) {
  if (!cost || !w1pack || !scale || !shift || !w2 || !conf || B <= 0 || D <= 0 || Cm <= 0 || H <= 0 || W <= 0)
    return fail(DMB_EINVAL, "conf_head: bad argument");
  if ((long long)D * H * W * 4 >= 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conf_head: one batch item must stay below 2 GiB");
  const int NTT = cdiv(Cm, 32);
  hipStream_t st = (hipStream_t)stream;
  if (NTT == 1) return launch_conf<1>(cost, w1pack, scale, shift, w2, conf, B, D, Cm, H, W, st);
  if (NTT == 2) return launch_conf<2>(cost, w1pack, scale, shift, w2, conf, B, D, Cm, H, W, st);
  return fail(DMB_EUNSUPPORTED, "conf_head: more than 64 intermediate maps");
}
