// 2-D convolution family of the PSMNet feature backbone (the first "next" row):
// dmb/modeling/stereo/backbones/PSMNet.py:8-129 + layers/basic_layers.py:31-46,105-123,219-243 (conv_bn,
// conv_bn_relu, BasicBlock).  Same FP32 implicit-GEMM design as conv3d.hip / confhead.hip:
//   y[co, p] = sum_{ci, tap} w[co, ci, tap] * x[ci, p * stride + (tap - KS/2) * DIL]    M = Cout, N = pixels, K = Cin*KS*KS
// v_mfma_f32_32x32x2_f32, A = prepacked weight fragments, B = row-pair tiles of the haloed LDS tile (lanes 0-15: 16
// pixels of row r, lanes 16-31: the same columns of row r + 1), LDS-DMA double-buffered chunks of CK input channels
// (input rows + the chunk's weight fragments), fragments register double-buffered one k-step ahead.
// One kernel covers kernel size 1 / 3, dilation 1 / 2 / 4 / 8 (compile time; 4 and 8 are StereoNet's edge-aware
// refinement, disp_refinement/utils/edge_aware.py:33-40), kernel size 5 (StereoNet's down-sampling heads,
// backbones/StereoNet.py:26-27) and stride 1 / 2 (compile time: a strided tile stages (TX - 1) * 2 + 1 + 2 * HALO input
// columns and its B fragments walk LDS with stride 2).
// Input and output may be channel windows of wider tensors (the 320-channel SPP concat is written in place).
#include <type_traits>

#include "dmb_common.h"

// Workgroups per CU the kernel is built for (build-time experiment knob, build.py DMB_BUILD_DEFS): LDS budget per workgroup,
// register cap and persistent grid follow it.
#ifndef DMB_C2_WPE
#define DMB_C2_WPE 2
#endif

namespace dmb {

constexpr int C2_WPE = DMB_C2_WPE;
constexpr long long C2_SK_MAX_UNITS_PER_CU = 5;   // split-K form of small 3x3 launches (dmb_conv2d_f32): calibrated in profiles/r06_c2d_sk_probe.log
constexpr int C2_LDS_BUDGET = 160 * 1024 / C2_WPE;
constexpr int C2_CK = 8;  // packed weight streams are zero-padded to a multiple of this many input channels

// wp[((kp * KK + tap) * NTT + nt) * 64 + lane] = w[co = nt*32 + (lane & 31)][ci = 2*kp + (lane >> 5)][tap]; zero padded
__global__ void pack_conv2d_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int Cipad, int KK,
                                   int NTT) {
  const long long total = (long long)(Cipad / 2) * KK * NTT * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    long long r = i >> 6;
    const int nt = (int)(r % NTT);
    r /= NTT;
    const int tap = (int)(r % KK);
    const int kp = (int)(r / KK);
    const int co = nt * 32 + (lane & 31);
    const int ci = 2 * kp + (lane >> 5);
    wp[i] = (co < Co && ci < Ci) ? w[((size_t)co * Ci + ci) * KK + tap] : 0.f;
  }
}

// V16 = the vector path, taken when rows are 16-byte aligned (W, Wo multiples of 4, aligned base pointers): the input
// tile is staged with 16-byte LDS-DMA words (a tile row starts at the aligned column x0 * S - LP) and the epilogue
// transposes each 32 x 32 accumulator tile through LDS so that a lane stores 4 consecutive pixels of one channel.
// Both cut the vector-memory INSTRUCTION count 4x: measured on MI355X these layers are bound by the rate at which a CU
// issues 64-lane memory instructions (a 3x3 layer has 3x less arithmetic per staged byte than the 3x3x3 ones), not by
// bytes: 32->32 at 384x1248 ran 1.65 ms with dword copies / stores against 0.97 ms with neither.
// RYO: output rows per wave, 0 = the default (4 at stride 1).  The persistent grid is two workgroups per CU and a launch of the
// backbone is a few tile rounds deep, so the tile HEIGHT is picked per launch by rounds x rows (launch_conv2d_auto): 8 images of
// 136 x 240 at 64 channels are 680 tiles of 8 rows = 1.33 rounds on 512 slots (the chip works for 2) or 1360 tiles of 4 rows
// = 2.66 rounds (it works for 3 half-sized ones): 0.195 -> 0.16 ms for each of PSMNet's 31 layer2 convolutions.
template <int NTT_, int KS_, int DIL_, int S_ = 1, bool V16_ = true, bool DOT_ = false, int RYO_ = 0>
struct C2Cfg {
  static constexpr int NTT = NTT_, KS = KS_, DIL = DIL_, S = S_;
  static constexpr bool V16 = V16_;
  // DOT (128 output channels = two groups of 64): instead of storing the activations, every pixel's ReLU(BN(.)) vector of a
  // 64-channel group is reduced against a weight vector and the sigmoid of that sum is scattered into a 4x finer map --
  // AcfNet's confidence head composed with its learned up-sampling (ops.conf_head_from_source): the [B, 1024, H/4, W/4]
  // hidden tensor (0.5 GB at the BASELINE size) is never written.  `res` = the 64 weights, `y` = the confidence map
  // [B, 1, 4 H, 4 W], `res_ctot` = the number of weight sets laid out one after the other in `wp`: ONE launch walks
  // (spatial tile, weight set) work items, set p = phases 2 p and 2 p + 1 (phase = 4 * (y mod 4) + (x mod 4)).
  static constexpr bool DOT = DOT_;
  static constexpr int WN = NTT;               // one 32-channel row tile per wave column (1, 2 or 4)
  static constexpr int WY = 4 / WN;            // waves stacked along y
  static constexpr int RY = RYO_ > 0 ? RYO_ : ((S == 1) ? 4 : 2);  // output rows per wave (row pairs); strided tiles read 2x2 the input
  static constexpr int TY = RY * WY, TX = 48;
  static constexpr int HALO = (KS / 2) * DIL;
  static constexpr int LP = V16 ? (HALO + 3) / 4 * 4 : HALO;           // staged columns left of the tile origin
  static constexpr int PMIN = LP + (TX - 1) * S + 1 + HALO;
  static constexpr int P = V16 ? (PMIN + 3) / 4 * 4 : PMIN;            // staged input row length (LDS row pitch)
  static constexpr int ROWS = (TY - 1) * S + 1 + 2 * HALO;
  static constexpr int SEG = (P + 63) / 64;    // scalar path: one wave stages 64 floats of a tile row per instruction
  static constexpr int XS = TX / 16;
  static constexpr int MT = (RY / 2) * XS;     // row-pair tiles per wave (6, or 3 when strided)
  static constexpr int KK = KS * KS;
  static constexpr int CH_STRIDE = V16 ? ROWS * P : ROWS * P + 4;      // vector path: channels are contiguous 16-byte units
  // input channels per chunk: the largest of 8 / 4 / 2 whose double-buffered chunk (input rows + weight fragments)
  // lets two workgroups share one CU's 160 KB of LDS
  static constexpr int lds_bytes(int ck) { return 2 * ((ck * CH_STRIDE + 3) / 4 * 4 + (ck / 2) * KK * NTT * 64) * 4; }
  static constexpr int CK = lds_bytes(8) <= C2_LDS_BUDGET ? 8 : (lds_bytes(4) <= C2_LDS_BUDGET ? 4 : 2);
  static constexpr int NK = (CK / 2) * KK;  // k-steps per chunk
  static constexpr int IN_FLOATS = (CK * CH_STRIDE + 3) / 4 * 4;
  static constexpr int W_FLOATS = NK * NTT * 64;
  static constexpr int TR_PITCH = 36;                        // transposition scratch: [32 channels][32 pixels + 4]
  static constexpr int TR_FLOATS = 4 * 32 * TR_PITCH;        // one accumulator tile per wave
  // (a chunk buffer a few floats short of the scratch is rounded up to it: cheaper than a third region behind the two buffers)
  static constexpr int BUF_RAW = IN_FLOATS + W_FLOATS;
  static constexpr int BUF_FLOATS = (V16_ && BUF_RAW < TR_FLOATS && TR_FLOATS - BUF_RAW <= 256) ? TR_FLOATS : BUF_RAW;
  // the scratch lives in the chunk buffer that was consumed last (free until the next copy lands in it) when that
  // buffer is large enough, else (1x1 layers) behind the two buffers
  static constexpr bool TR_OWN = V16 && BUF_FLOATS < TR_FLOATS;
  static constexpr int LDS_FLOATS = 2 * BUF_FLOATS + (TR_OWN ? TR_FLOATS : 0);
  static constexpr int UNITS = CK * ROWS * SEG;              // scalar path: 64-float row segments per chunk
  static constexpr int UPR = P / 4, UPC = ROWS * UPR;        // vector path: 16-byte units per row / per channel
  static constexpr int VUNITS = CK * UPC;
  static_assert(C2_CK % CK == 0, "chunks tile the padded channel count");
  static_assert(W_FLOATS % 16 == 0, "weights are copied with 16-byte words, evenly over 4 waves");
  static_assert(LDS_FLOATS * 4 <= C2_LDS_BUDGET, "C2_WPE workgroups per CU");
};

template <class F>
__device__ __forceinline__ void noalias_step(const float* __restrict__ cur, float* __restrict__ nxt, F&& f) {
  f(cur, nxt);
}

// Persistent workgroups: the grid is two workgroups per CU, workgroup g walks tiles g, g + G, g + 2G, ... (ids remapped so
// that one XCD owns a contiguous tile range).  The chunk pipeline runs ACROSS tiles: the first chunk of the next tile
// is in flight while the last chunk of the current one is multiplied, and a tile's stores drain while the next tile's
// first chunk is multiplied.
// MULTI: one launch walks the tiles of several independent convolutions ("jobs") that share the layer shape (Ci, Co, H, kernel)
// and differ in input, weights, output and image WIDTH: the five 2-D convolutions of the volume-free first layer
// (csrc/catconv.hip), three of which are 52-column border maps of 272 tiles each -- one after the other they take a round of
// the chip apiece for a tenth of a round's arithmetic.
constexpr int C2_MAXJOBS = 6;
struct C2Jobs {
  const float* x[C2_MAXJOBS];
  const float* wp[C2_MAXJOBS];
  float* y[C2_MAXJOBS];
  int W[C2_MAXJOBS], out_ctot[C2_MAXJOBS], ntx[C2_MAXJOBS], tile_begin[C2_MAXJOBS];
  int njobs;
};

template <class C, bool MULTI = false>
__global__ __launch_bounds__(256, C2_WPE) void conv2d_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const float* __restrict__ res, float* __restrict__ y, int Ci,
                                                        int Co, int H, int W, int relu, int in_ctot, int out_ctot,
                                                        int res_ctot, int ntx, int nty, int ntiles, const C2Jobs jobs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int G = gridDim.x;
  if ((int)blockIdx.x >= ntiles) return;
  const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int wy = wave / C::WN, wn = wave % C::WN;
  const int Cipad = cdiv(Ci, C2_CK) * C2_CK;
  const int Ho = (H - 1) / C::S + 1;

  struct Tile {
    int b, x0, y0;   // batch item, output coordinates of the tile origin
    int p;           // DOT: which of the launch's weight sets (pairs of phases); 0 otherwise
    int job;         // MULTI: which convolution of the launch
  };
  // per-tile view of what MULTI makes job-dependent (wave-uniform: scalar loads from the kernel arguments)
  auto W_of = [&](const Tile& tl) { return MULTI ? jobs.W[tl.job] : W; };
  auto x_of = [&](const Tile& tl) { return MULTI ? jobs.x[tl.job] : x; };
  auto y_of = [&](const Tile& tl) { return MULTI ? jobs.y[tl.job] : y; };
  auto wp_of = [&](const Tile& tl) { return MULTI ? jobs.wp[tl.job] : wp; };
  auto octot_of = [&](const Tile& tl) { return MULTI ? jobs.out_ctot[tl.job] : out_ctot; };
  auto tile_at = [&](int it) {
    int t = xcd_remap((int)blockIdx.x + it * G, ntiles);
    Tile tl;
    tl.p = 0;
    tl.job = 0;
    int ntx_ = ntx;
    if constexpr (MULTI) {
#pragma unroll
      for (int q = 1; q < C2_MAXJOBS; ++q)
        if (q < jobs.njobs && t >= jobs.tile_begin[q]) tl.job = q;
      t -= jobs.tile_begin[tl.job];
      ntx_ = jobs.ntx[tl.job];
    }
    if constexpr (C::DOT) {   // res_ctot = number of weight sets; the set is the fastest index: 8 consecutive work items
      tl.p = t % res_ctot;    // share one input tile
      t /= res_ctot;
    }
    tl.x0 = (t % ntx_) * C::TX;
    t /= ntx_;
    tl.y0 = (t % nty) * C::TY;
    tl.b = t / nty;
    return tl;
  };

  constexpr int WPW = C::W_FLOATS / 16;      // 16-byte weight words per wave
  constexpr int WI = (WPW + 63) / 64;
  // Vector path: the per-lane source offsets of a chunk's copies depend on the TILE only (the chunk's channels are selected by the
  // resource base), so they are computed once per tile, not once per chunk: the unit -> (channel, row, 16-byte column) decode, the
  // bounds tests and the row multiplies were 50 - 70 vector instructions per chunk (five of them quarter-rate multiplies) next to
  // 108 - 216 MFMAs, in a loop where vector time adds to matrix time.
  constexpr int IPW = C::V16 ? (C::VUNITS + 255) / 256 : 1;   // copy instructions per wave and chunk
  auto tile_offsets = [&](const Tile& tl, unsigned (&off)[IPW]) {
    if constexpr (C::V16) {
      const int W = W_of(tl);
      const unsigned HW = (unsigned)H * W;
      const int gxb = tl.x0 * C::S - C::LP, gyb = tl.y0 * C::S - C::HALO;
#pragma unroll
      for (int q = 0; q < IPW; ++q) {
        // unit u = 4 consecutive floats of the staged chunk, linear in LDS: u -> (channel, tile row, 16-byte column)
        const int u = (wave * IPW + q) * 64 + lane;
        const int cl = u / C::UPC, rr = u - cl * C::UPC, yy = rr / C::UPR, sg = rr - yy * C::UPR;
        const int gy = gyb + yy, gx = gxb + sg * 4;
        const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
        off[q] = ok ? ((unsigned)cl * HW + (unsigned)gy * W + (unsigned)gx) * 4u : DMA_OOB;
      }
    }
  };
  auto stage = [&](const Tile& tl, int c0, float* buf, const unsigned (&off)[IPW]) {
    const int W = W_of(tl);   // (shadows the launch's W: everything below is per tile)
    const unsigned HW = (unsigned)H * W;
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(wp_of(tl), (unsigned)((C::DOT ? res_ctot : 1) * (Cipad / 2) * C::KK * C::NTT * 64) * 4u);
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(x_of(tl) + (size_t)tl.b * in_ctot * HW, (unsigned)Ci * HW * 4u);
    if constexpr (C::V16) {
      // the chunk's first channel is one uniform add per copy; a channel past Ci lies beyond the resource and reads zeros, and an
      // out-of-image unit (DMA_OOB) stays out of range (c0 * HW * 4 < 2^31: checked on the host)
      const unsigned cbase = (unsigned)c0 * HW * 4u;
#pragma unroll
      for (int q = 0; q < IPW; ++q)
        if ((wave * IPW + q) * 64 + lane < C::VUNITS)   // inactive lanes write nothing: the words behind the last unit belong to the weight fragments
          dma16(xrs, off[q] + cbase, 0u, buf + (wave * IPW + q) * 256);
    } else {
      constexpr int RPW = (C::UNITS + 3) / 4;    // row segments per wave per chunk
      const int gx0 = tl.x0 * C::S - C::HALO + lane;
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const int rid = wave * RPW + q;
        if (C::UNITS % 4 == 0 || rid < C::UNITS) {
          const int seg = rid % C::SEG, rr = rid / C::SEG, cl = rr / C::ROWS, yy = rr - cl * C::ROWS;
          const int gy = tl.y0 * C::S - C::HALO + yy, gx = gx0 + seg * 64;
          const bool rowok = c0 + cl < Ci && gy >= 0 && gy < H;               // wave-uniform: the scalar offset stays an SGPR
          const bool ok = rowok && seg * 64 + lane < C::P && gx >= 0 && gx < W;
          if (C::P % 64 == 0 || seg * 64 + lane < C::P)
            dma4(xrs, ok ? (unsigned)gx * 4u : DMA_OOB, rowok ? ((unsigned)(c0 + cl) * HW + (unsigned)gy * W) * 4u : 0u,
                 buf + cl * C::CH_STRIDE + yy * C::P + seg * 64);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int q4 = wave * WPW + i * 64 + lane;
      if (i * 64 + lane < WPW)
        dma16(wrs, (unsigned)q4 * 16u, ((unsigned)(c0 / 2) + (unsigned)tl.p * (unsigned)(Cipad / 2)) * (C::KK * C::NTT * 64 * 4),
              buf + C::IN_FLOATS + (wave * WPW + i * 64) * 4);
    }
  };

  f32x16 acc[C::MT];
  auto clear = [&]() {
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  };

  // Per-channel affine of this lane's output channels, loaded ONCE: a load inside the tile loop that some path leaves
  // unconsumed stays "pending" in the compiler's wait-count model across the back edge, and the first instruction that
  // reuses its register then drains vmcnt(0) -- including the chunk copy that was just put in flight.
  constexpr int NAFF = (C::V16 && !C::DOT) ? 4 : 16;
  float sc[NAFF], sh[NAFF];
#pragma unroll
  for (int k = 0; k < NAFF; ++k) {
    const int co = wn * 32 + ((C::V16 && !C::DOT) ? k * 8 + (lane >> 3) : cd_row(k, h));
    sc[k] = (scale && co < Co) ? scale[co] : 1.f;
    sh[k] = (shift && co < Co) ? shift[co] : 0.f;
  }

  // epilogue: BN scale/shift -> + residual -> ReLU (basic_layers.py:219-243 adds the skip AFTER conv2's BN, no ReLU)
  auto epilogue = [&](auto has_res, const Tile& tl, float* scratch) {
    constexpr bool HAS_RES = decltype(has_res)::value;
    const int Wo = (W_of(tl) - 1) / C::S + 1, out_ctot = octot_of(tl);   // (shadow the launch's: per tile)
    const unsigned HWo = (unsigned)Ho * Wo;
    float* yb = y_of(tl) + (size_t)tl.b * out_ctot * HWo;
    const float* rb = res ? res + (size_t)tl.b * res_ctot * HWo : nullptr;
    if constexpr (C::V16) {
      // accumulator tile -> scratch[channel][pixel] -> a lane owns 4 consecutive pixels of one channel.  Loads and
      // stores go through buffer resources: an out-of-tile lane gets an out-of-range offset (reads 0 / store dropped),
      // so the epilogue has no divergent branches and the compiler batches the residual loads under counted waits.
      float* my = scratch + wave * (32 * C::TR_PITCH);
      const __amdgpu_buffer_rsrc_t yrs = make_rsrc(yb, (unsigned)out_ctot * HWo * 4u);
      const __amdgpu_buffer_rsrc_t rrs = make_rsrc(rb ? rb : yb, (unsigned)(rb ? res_ctot : out_ctot) * HWo * 4u);
      const int px = (lane & 7) * 4;                       // 0..28: pixels 0-15 = first row of the pair, 16-31 = second
      const float lo = relu ? 0.f : -__builtin_inff();     // branch-free optional ReLU
      auto offsets = [&](int mt, unsigned (&off)[4]) {
        const int gy = tl.y0 + wy * C::RY + 2 * (mt / C::XS) + (px >> 4);
        const int gxo = tl.x0 + (mt % C::XS) * 16 + (px & 15);
        const bool inb = gy < Ho && gxo < Wo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int co = wn * 32 + k * 8 + (lane >> 3);
          off[k] = (inb && co < Co) ? ((unsigned)co * HWo + (unsigned)gy * Wo + (unsigned)gxo) * 4u : DMA_OOB;
        }
      };
      // residual loads run one accumulator tile ahead of the stores: the counter a load is waited on (vmcnt, in issue
      // order) then never covers a store
      unsigned off[2][4];
      u32x4 rv[2][4];
      offsets(0, off[0]);
      if constexpr (HAS_RES) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rv[0][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)off[0][k], 0, 0);
      }
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
        if (mt + 1 < C::MT) {
          offsets(mt + 1, off[(mt + 1) & 1]);
          if constexpr (HAS_RES) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              rv[(mt + 1) & 1][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)off[(mt + 1) & 1][k], 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) my[cd_row(r, h) * C::TR_PITCH + j] = acc[mt][r];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float4 v = *reinterpret_cast<const float4*>(my + (k * 8 + (lane >> 3)) * C::TR_PITCH + px);
          v.x = fmaf(v.x, sc[k], sh[k]);
          v.y = fmaf(v.y, sc[k], sh[k]);
          v.z = fmaf(v.z, sc[k], sh[k]);
          v.w = fmaf(v.w, sc[k], sh[k]);
          if constexpr (HAS_RES) {   // (not __builtin_bit_cast on a vector element: this clang reads element 0 for every index)
            v.x += __uint_as_float(rv[mt & 1][k].x);
            v.y += __uint_as_float(rv[mt & 1][k].y);
            v.z += __uint_as_float(rv[mt & 1][k].z);
            v.w += __uint_as_float(rv[mt & 1][k].w);
          }
          u32x4 o;
          o.x = __float_as_uint(fmaxf(v.x, lo));
          o.y = __float_as_uint(fmaxf(v.y, lo));
          o.z = __float_as_uint(fmaxf(v.z, lo));
          o.w = __float_as_uint(fmaxf(v.w, lo));
          __builtin_amdgcn_raw_buffer_store_b128(o, yrs, (int)off[mt & 1][k], 0, 0);
        }
      }
    } else {
      bool cok[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) cok[r] = wn * 32 + cd_row(r, h) < Co;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
        const int gy = tl.y0 + wy * C::RY + 2 * (mt / C::XS) + (j >> 4);
        const int gxo = tl.x0 + (mt % C::XS) * 16 + (j & 15);
        if (gy < Ho && gxo < Wo) {
          const unsigned o = (unsigned)gy * Wo + (unsigned)gxo;
          float rv[16];
          if (rb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = cok[r] ? rb[(size_t)(wn * 32 + cd_row(r, h)) * HWo + o] : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = fmaf(acc[mt][r], sc[r], sh[r]);
            if (rb) v += rv[r];
            if (relu) v = fmaxf(v, 0.f);
            if (cok[r]) yb[(size_t)(wn * 32 + cd_row(r, h)) * HWo + o] = v;
          }
        }
      }
    }
  };

  // DOT epilogue: lane (j, h) of wave wn holds, for pixel j of each row-pair tile, channels wn * 32 + cd_row(r, h)
  auto dot_epilogue = [&](const Tile& tl, float* scratch) {
    const int Wo = (W - 1) / C::S + 1;
    const unsigned HWo = (unsigned)Ho * Wo;
    float w2r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) w2r[r] = res[(wn & 1) * 32 + cd_row(r, h)];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
      float p = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) p = fmaf(fmaxf(fmaf(acc[mt][r], sc[r], sh[r]), 0.f), w2r[r], p);
      scratch[((wave * 2 + h) * C::MT + mt) * 32 + j] = p;
    }
    __syncthreads();
    constexpr int NOUT = 2 * C::MT * 32;   // (phase of the launch, tile, pixel)
    float* cb = y + (size_t)tl.b * 16 * HWo;
    for (int o = threadIdx.x; o < NOUT; o += 256) {
      const int phl = o / (C::MT * 32), rem = o - phl * (C::MT * 32), mt = rem >> 5, jj = rem & 31;
      const float* q0 = scratch + ((2 * phl) * 2 * C::MT + mt) * 32 + jj;           // wave 2 phl: halves h = 0, 1
      const float* q1 = scratch + ((2 * phl + 1) * 2 * C::MT + mt) * 32 + jj;       // wave 2 phl + 1
      const float s = ((q0[0] + q0[C::MT * 32]) + q1[0]) + q1[C::MT * 32];         // fixed order: reproducible
      const int gy = tl.y0 + 2 * (mt / C::XS) + (jj >> 4), gx = tl.x0 + (mt % C::XS) * 16 + (jj & 15);
      const int ph = 2 * tl.p + phl;
      if (gy < Ho && gx < Wo) cb[(size_t)(4 * gy + (ph >> 2)) * (4 * Wo) + 4 * gx + (ph & 3)] = 1.f / (1.f + __expf(-s));
    }
  };

  const int NC = Cipad / C::CK;
  Tile cur_t = tile_at(0);
  unsigned coff[IPW], noff[IPW];   // copy offsets of the current / the next tile
  tile_offsets(cur_t, coff);
#pragma unroll
  for (int q = 0; q < IPW; ++q) noff[q] = coff[q];
  stage(cur_t, 0, lds, coff);
  __syncthreads();
  int g = 0;  // chunks consumed by this workgroup so far: selects the LDS buffer
  for (int it = 0; it < my_tiles; ++it) {
    const bool has_next = it + 1 < my_tiles;
    Tile next_t = cur_t;
    if (has_next) {
      next_t = tile_at(it + 1);
      tile_offsets(next_t, noff);
    }
    clear();
    for (int ci = 0; ci < NC; ++ci, ++g) {
      // The copy into `nxt` must stay in flight while `cur` is multiplied.  The compiler orders an LDS read after every
      // outstanding LDS-DMA write it cannot prove disjoint (s_waitcnt vmcnt(0) in front of the first ds_read, which
      // would serialise copy and arithmetic); the __restrict__ scope of noalias_step() is that proof.  The explicit
      // vmcnt(0) below is what makes the data visible to the other waves before the barrier.
      noalias_step(lds + (g & 1) * C::BUF_FLOATS, lds + ((g + 1) & 1) * C::BUF_FLOATS,
                   [&](const float* cur, float* nxt) __attribute__((always_inline)) {
        {  // ONE inlined copy of the staging code: what to fetch next is data, not control flow
          const bool same = ci + 1 < NC;
          Tile st_t;
          st_t.b = same ? cur_t.b : next_t.b;
          st_t.x0 = same ? cur_t.x0 : next_t.x0;
          st_t.y0 = same ? cur_t.y0 : next_t.y0;
          st_t.p = same ? cur_t.p : next_t.p;
          st_t.job = same ? cur_t.job : next_t.job;
          unsigned st_off[IPW];
#pragma unroll
          for (int q = 0; q < IPW; ++q) st_off[q] = same ? coff[q] : noff[q];
          if (same || has_next) stage(st_t, same ? (ci + 1) * C::CK : 0, nxt, st_off);
        }
        const float* abase = cur + C::IN_FLOATS + wn * 64 + lane;
        const float* bbase = cur + h * C::CH_STRIDE + ((wy * C::RY + (j >> 4)) * C::P + (j & 15)) * C::S + (C::LP - C::HALO);
        float af[2], bf[2][C::MT];
        auto load_frag = [&](int ks, float& a, float (&bq)[C::MT]) {
          const int cp = ks / C::KK, tap = ks % C::KK;
          const int dy = tap / C::KS, dx = tap % C::KS;
          a = abase[ks * C::NTT * 64];
          const float* bp = bbase + 2 * cp * C::CH_STRIDE + dy * C::DIL * C::P + dx * C::DIL;
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) bq[mt] = bp[(2 * (mt / C::XS) * C::P + (mt % C::XS) * 16) * C::S];
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
      });
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's copies have landed in LDS
      __syncthreads();
    }
    // the chunk buffer just consumed is free until the next copy lands in it: it doubles as transposition scratch
    float* scratch = C::TR_OWN ? lds + 2 * C::BUF_FLOATS : lds + ((g + 1) & 1) * C::BUF_FLOATS;
    if constexpr (C::DOT) {
      dot_epilogue(cur_t, scratch);
    } else {
      if (res)
        epilogue(std::true_type{}, cur_t, scratch);
      else
        epilogue(std::false_type{}, cur_t, scratch);
    }
    if constexpr ((C::V16 && !C::TR_OWN) || C::DOT) __syncthreads();
    cur_t = next_t;
#pragma unroll
    for (int q = 0; q < IPW; ++q) coff[q] = noff[q];
  }
}

// nn.AvgPool2d(k, stride=k) (PSMNet.py:43-58): one WAVE per output element -- lanes take the window's elements round robin
// (coalesced along the window rows), FP32 partial sums, one wave reduction, / k^2.  (One thread per element walked the 64 x 64
// window of the largest branch serially: 0.58 ms for 48 outputs.)
__global__ __launch_bounds__(256) void avgpool2d_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C,
                                                        int H, int W, int k, int Ho, int Wo, int in_ctot, int in_coff) {
  const long long i = blockIdx.x * 4LL + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= (long long)B * C * Ho * Wo) return;
  const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho);
  const long long bc = i / ((long long)Wo * Ho);
  const long long b = bc / C, c = bc - b * C;
  const float* p = x + (((b * in_ctot + in_coff + c) * H) + (long long)yo * k) * W + (long long)xo * k;
  float s = 0.f;
  for (int e = lane; e < k * k; e += 64) s += p[(long long)(e / k) * W + e % k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) y[i] = s / (float)(k * k);
}

// F.interpolate(mode='bilinear', align_corners=True) (PSMNet.py:95-117) into a channel window of a wider tensor.
// Same contraction-free index arithmetic as the trilinear kernel (regression.hip).
__global__ __launch_bounds__(256) void bilinear_ac_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int Hi,
                                                          int Wi, int Ho, int Wo, float sh, float sw, int out_ctot,
                                                          int out_coff) {
#pragma clang fp contract(off)
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= (long long)C * Ho * Wo) return;
  const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho), c = (int)(i / ((long long)Wo * Ho));
  const float sy = sh * (float)yo, sx = sw * (float)xo;
  int y0 = (int)sy, x0 = (int)sx;
  y0 = y0 > Hi - 1 ? Hi - 1 : y0;
  x0 = x0 > Wi - 1 ? Wi - 1 : x0;
  const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
  float ly = sy - (float)y0, lx = sx - (float)x0;
  ly = fminf(fmaxf(ly, 0.f), 1.f);
  lx = fminf(fmaxf(lx, 0.f), 1.f);
  const float* p = x + ((size_t)b * C + c) * Hi * Wi;
  const float a0 = fmaf(p[(size_t)y0 * Wi + x1], lx, p[(size_t)y0 * Wi + x0] * (1.f - lx));
  const float a1 = fmaf(p[(size_t)y1 * Wi + x1], lx, p[(size_t)y1 * Wi + x0] * (1.f - lx));
  y[(((size_t)b * out_ctot + out_coff + c) * Ho + yo) * Wo + xo] = fmaf(a1, ly, a0 * (1.f - ly));
}

// F.interpolate(mode='bilinear', align_corners=False) * mult (disp_refinement/StereoNet.py:49-50, edge_aware.py:49-50:
// the coarse disparity map up-sampled to image size and rescaled by the resolution ratio).  ATen's source index:
// src = max(scale * (dst + 0.5) - 0.5, 0), scale = in / out in FP32.
__global__ __launch_bounds__(256) void bilinear_hp_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int Hi,
                                                          int Wi, int Ho, int Wo, float sh, float sw, float mult,
                                                          int out_ctot, int out_coff) {
#pragma clang fp contract(off)
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= (long long)C * Ho * Wo) return;
  const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho), c = (int)(i / ((long long)Wo * Ho));
  const float sy = fmaxf(sh * ((float)yo + 0.5f) - 0.5f, 0.f), sx = fmaxf(sw * ((float)xo + 0.5f) - 0.5f, 0.f);
  int y0 = (int)sy, x0 = (int)sx;
  y0 = y0 > Hi - 1 ? Hi - 1 : y0;
  x0 = x0 > Wi - 1 ? Wi - 1 : x0;
  const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
  float ly = sy - (float)y0, lx = sx - (float)x0;
  ly = fminf(fmaxf(ly, 0.f), 1.f);
  lx = fminf(fmaxf(lx, 0.f), 1.f);
  const float* p = x + ((size_t)b * C + c) * Hi * Wi;
  const float a0 = fmaf(p[(size_t)y0 * Wi + x1], lx, p[(size_t)y0 * Wi + x0] * (1.f - lx));
  const float a1 = fmaf(p[(size_t)y1 * Wi + x1], lx, p[(size_t)y1 * Wi + x0] * (1.f - lx));
  y[(((size_t)b * out_ctot + out_coff + c) * Ho + yo) * Wo + xo] = fmaf(a1, ly, a0 * (1.f - ly)) * mult;
}

template <class C>
static int launch_conv2d(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                         float* y, int B, int Ci, int Co, int H, int W, int relu, int in_ctot, int out_ctot,
                         int res_ctot, hipStream_t st) {
  const int Ho = (H - 1) / C::S + 1, Wo = (W - 1) / C::S + 1;
  const int ntx = cdiv(Wo, C::TX), nty = cdiv(Ho, C::TY);
  const long long ntiles = (long long)B * ntx * nty * (C::DOT ? res_ctot : 1);
  if (ntiles > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv2d: grid too large");
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  DMB_ENSURE_LDS((&conv2d_kernel<C>), (size_t)(lds));
  const long long slots = (long long)C2_WPE * num_cus();   // C2_WPE workgroups per CU, a multiple of the 8 XCDs
  const unsigned grid = (unsigned)(ntiles < slots ? ntiles : slots);
  hipLaunchKernelGGL((conv2d_kernel<C>), dim3(grid), dim3(256), lds, st, x, wp, scale, shift, res, y, Ci, Co, H, W, relu,
                     in_ctot, out_ctot, res_ctot, ntx, nty, (int)ntiles, C2Jobs{});
  return launch_status("conv2d launch failed");
}

// Tile height by rounds x rows (see C2Cfg): the default 4 rows per wave, or 2 where that takes less of the chip's time.
template <int N, int K, int DL, bool V16>
static int launch_conv2d_auto(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                              float* y, int B, int Ci, int Co, int H, int W, int relu, int in_ctot, int out_ctot,
                              int res_ctot, hipStream_t st) {
  using C4 = C2Cfg<N, K, DL, 1, V16, false, 0>;
  using C2 = C2Cfg<N, K, DL, 1, V16, false, 2>;
  const long long slots = (long long)C2_WPE * num_cus();
  const long long t4 = (long long)B * cdiv(W, C4::TX) * cdiv(H, C4::TY), t2 = (long long)B * cdiv(W, C2::TX) * cdiv(H, C2::TY);
  const long long cost4 = ((t4 + slots - 1) / slots) * C4::RY, cost2 = ((t2 + slots - 1) / slots) * C2::RY;
  // (round 6, profiles/r06_kbench2d_tile_height.log: at equal cost the two-row tiles win for 32 and 128 output channels -- one launch of
  // 8 images: 128 -> 128 0.611 -> 0.579 ms, 320 -> 128 1.505 -> 1.409, 32 -> 32 0.194 -> 0.185 -- and tie for 64.  Under the backbone's
  // two view streams the tile height does not matter: the other view's launch fills a launch's tail, 14.8 ms either way.)
  const bool two = cost2 < cost4 || (cost2 == cost4 && N != 2);
  if ((two && !DMB_OPT(18)) || DMB_OPT(18) == 2)   // (development option 18: 1 = always the default height, 2 = always 2 rows per wave)
    return launch_conv2d<C2>(x, wp, scale, shift, res, y, B, Ci, Co, H, W, relu, in_ctot, out_ctot, res_ctot, st);
  return launch_conv2d<C4>(x, wp, scale, shift, res, y, B, Ci, Co, H, W, relu, in_ctot, out_ctot, res_ctot, st);
}

// Several convolutions of one layer shape in ONE launch (see C2Jobs): no affine, no residual, no ReLU, stride 1.
template <class C>
static int launch_conv2d_multi(const C2Jobs& jobs_in, int B, int Ci, int Co, int H, hipStream_t st) {
  static_assert(C::S == 1 && !C::DOT, "multi-job launches are stride-1 convolutions");
  C2Jobs jobs = jobs_in;
  const int nty = cdiv(H, C::TY);
  long long ntiles = 0;
  for (int q = 0; q < jobs.njobs; ++q) {
    jobs.ntx[q] = cdiv(jobs.W[q], C::TX);
    jobs.tile_begin[q] = (int)ntiles;
    ntiles += (long long)B * jobs.ntx[q] * nty;
    if (ntiles > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv2d_multi: grid too large");
  }
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  DMB_ENSURE_LDS((&conv2d_kernel<C, true>), (size_t)(lds));
  const long long slots = (long long)C2_WPE * num_cus();
  const unsigned grid = (unsigned)(ntiles < slots ? ntiles : slots);
  hipLaunchKernelGGL((conv2d_kernel<C, true>), dim3(grid), dim3(256), lds, st, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, Ci, Co, H, 0, 0, Ci,
                     0, 0, 0, nty, (int)ntiles, jobs);
  return launch_status("conv2d_multi launch failed");
}

}  // namespace dmb

using namespace dmb;

static int c2_cipad(int Ci) { return cdiv(Ci, C2_CK) * C2_CK; }

extern "C" long long dmb_conv2d_packed_floats(int Co, int Ci, int ksize) {
  if (Co <= 0 || Ci <= 0 || (ksize != 1 && ksize != 3 && ksize != 5)) return 0;
  return (long long)(c2_cipad(Ci) / 2) * ksize * ksize * cdiv(Co, 32) * 64;
}

extern "C" int dmb_conv2d_pack_weights_f32(const float* w, float* wpack, int Co, int Ci, int ksize, void* stream) {
  if (!w || !wpack || Co <= 0 || Ci <= 0 || (ksize != 1 && ksize != 3 && ksize != 5))
    return fail(DMB_EINVAL, "conv2d_pack: bad argument");
  const int NTT = cdiv(Co, 32);
  if (NTT != 1 && NTT != 2 && NTT != 4) return fail(DMB_EUNSUPPORTED, "conv2d: output channels must fit 32, 64 or 128");
  hipLaunchKernelGGL(pack_conv2d_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, w, wpack, Co, Ci, c2_cipad(Ci),
                     ksize * ksize, NTT);
  return launch_status("conv2d_pack launch failed");
}

// The five conv2d weight packs of the first layer's 2-D form (csrc/catconv.hip) from the 3-D weight in ONE launch: blockIdx.y =
// pack (A, B1, B2 = left half with the dx taps from 0 / 1 / 2; HC, HD = right half with the dx taps up to 2 / 1), each the
// pack_conv2d_kernel layout of the virtual [CA, C, 3, 3] tensor k[dz * Co + co][ci][dy][dx] = half[co][ci][dz][dy][dx] (rows
// from 3 Co on and the taps outside the range: zero).  sign = -1 on the right half of a difference volume (exact).
__global__ void catconv_pack_kernel(const float* __restrict__ w, float* __restrict__ packs, long long pack_floats, int Co, int C,
                                    int Cw, int right_off, float right_sign, int Cipad, int NTT) {
  const int p = blockIdx.y;
  const int dx_from = p == 1 ? 1 : (p == 2 ? 2 : 0), dx_to = p == 4 ? 2 : 3;
  const bool right = p >= 3;
  const int choff = right ? right_off : 0;
  const float sgn = right ? right_sign : 1.f;
  float* wp = packs + (long long)p * pack_floats;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < pack_floats; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    long long r = i >> 6;
    const int nt = (int)(r % NTT);
    r /= NTT;
    const int tap = (int)(r % 9);
    const int kp = (int)(r / 9);
    const int row = nt * 32 + (lane & 31), ci = 2 * kp + (lane >> 5);
    const int dz = row / Co, co = row - dz * Co, dy = tap / 3, dx = tap - dy * 3;
    float v = 0.f;
    if (dz < 3 && ci < C && dx >= dx_from && dx < dx_to) v = sgn * w[((size_t)co * Cw + choff + ci) * 27 + dz * 9 + dy * 3 + dx];
    wp[i] = v;
  }
}

extern "C" int dmb_catconv_pack_weights_f32(const float* w, float* packs, int Co, int C, int CA, int dif, void* stream) {
  if (!w || !packs || Co <= 0 || C <= 0 || 3 * Co > CA || (CA != 32 && CA != 64 && CA != 128))
    return fail(DMB_EINVAL, "catconv_pack: bad argument");
  const int NTT = CA / 32, Cipad = c2_cipad(C);
  const long long pack_floats = (long long)(Cipad / 2) * 9 * NTT * 64;
  hipLaunchKernelGGL(catconv_pack_kernel, dim3(64, 5), dim3(256), 0, (hipStream_t)stream, w, packs, pack_floats, Co, C,
                     dif ? C : 2 * C, dif ? 0 : C, dif ? -1.f : 1.f, Cipad, NTT);
  return launch_status("catconv_pack launch failed");
}

extern "C" int dmb_conv2d_f32(const float* x, const float* wpack, const float* scale, const float* shift,
                              const float* residual, float* y, int B, int Ci, int Co, int H, int W, int ksize, int stride,
                              int dilation, int relu, int in_channels_total, int out_channels_total,
                              int res_channels_total, void* stream) {
  if (!x || !wpack || !y || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "conv2d: bad argument");
  if (stride != 1 && stride != 2) return fail(DMB_EUNSUPPORTED, "conv2d: stride must be 1 or 2");
  if (in_channels_total < Ci || out_channels_total < Co || (residual && res_channels_total < Co))
    return fail(DMB_EINVAL, "conv2d: channel window");
  // (+ 8: the staging adds the chunk's first channel, < Ci rounded up to the chunk size, to per-lane offsets of which DMA_OOB =
  // 2^31 marks "outside the image": the sum must not wrap past 2^32 back into the tensor)
  if ((long long)in_channels_total * H * W * 4 >= 0x7fffffffLL || (long long)out_channels_total * H * W * 4 >= 0x7fffffffLL ||
      (long long)res_channels_total * H * W * 4 >= 0x7fffffffLL || (long long)(Ci + 8) * H * W * 4 >= 0x7fffffffLL)
    return fail(DMB_EUNSUPPORTED, "conv2d: one batch item must stay below 2 GiB");
  const int NTT = cdiv(Co, 32);
  hipStream_t st = (hipStream_t)stream;
  const bool single_chain = (relu & DMB_CONV_SINGLE_CHAIN) != 0;
  relu &= 0xff;
  // vector path: every row of x, y and residual starts on a 16-byte boundary
  const int Wo_ = (W - 1) / stride + 1;
  const bool v16 = W % 4 == 0 && Wo_ % 4 == 0 && !DMB_OPT(3) &&
                   (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) == 0;
  // (round 6) a 3x3 layer that would leave most of the chip idle (one small image per view: backbones/PSMNet.py:8-129 as
  // dmb/apis/inference.py:191-225 calls it) takes the split-K form of csrc/conv3d_sk.hip: units = 16 x 2 pixel tiles x 32-channel
  // row tiles.  DMB_OPT(26) (development build): 1 = never, 2 = always.
  if (stride == 1 && ksize == 3 && (dilation == 1 || dilation == 2) && v16) {
    const long long units = (long long)B * cdiv(H, 2) * cdiv(W, 16) * NTT;
    if (((!single_chain && units <= C2_SK_MAX_UNITS_PER_CU * num_cus()) && DMB_OPT(26) != 1) || DMB_OPT(26) == 2) {
      const int rc = conv2d_sk_try(x, wpack, scale, shift, residual, y, B, Ci, Co, H, W, dilation, relu, in_channels_total,
                                   out_channels_total, res_channels_total, st);
      if (rc != -1) return rc;
    }
  }
#define DMB_C2(N, K, DL, S)                                                                                           \
  return v16 ? launch_conv2d<C2Cfg<N, K, DL, S, true>>(x, wpack, scale, shift, residual, y, B, Ci, Co, H, W, relu,     \
                                                       in_channels_total, out_channels_total, res_channels_total, st)  \
             : launch_conv2d<C2Cfg<N, K, DL, S, false>>(x, wpack, scale, shift, residual, y, B, Ci, Co, H, W, relu,    \
                                                        in_channels_total, out_channels_total, res_channels_total, st)
  if (stride == 1) {
    if (ksize == 3 && dilation == 1) {
      // the tile height follows the launch's size (launch_conv2d_auto; round 6: for every output width, not only 64 channels)
      if (NTT == 1) {
        if (v16) return launch_conv2d_auto<1, 3, 1, true>(x, wpack, scale, shift, residual, y, B, Ci, Co, H, W, relu, in_channels_total,
                                                          out_channels_total, res_channels_total, st);
        DMB_C2(1, 3, 1, 1);
      }
      if (NTT == 2) {   // PSMNet layer2 / StereoNet trunk widths
        if (v16) return launch_conv2d_auto<2, 3, 1, true>(x, wpack, scale, shift, residual, y, B, Ci, Co, H, W, relu, in_channels_total,
                                                          out_channels_total, res_channels_total, st);
        DMB_C2(2, 3, 1, 1);
      }
      if (NTT == 4) {
        if (v16) return launch_conv2d_auto<4, 3, 1, true>(x, wpack, scale, shift, residual, y, B, Ci, Co, H, W, relu, in_channels_total,
                                                          out_channels_total, res_channels_total, st);
        DMB_C2(4, 3, 1, 1);
      }
    } else if (ksize == 3 && dilation == 2) {
      if (NTT == 1) DMB_C2(1, 3, 2, 1);
      if (NTT == 2) DMB_C2(2, 3, 2, 1);
      if (NTT == 4) {
        if (v16) return launch_conv2d_auto<4, 3, 2, true>(x, wpack, scale, shift, residual, y, B, Ci, Co, H, W, relu, in_channels_total,
                                                          out_channels_total, res_channels_total, st);
        DMB_C2(4, 3, 2, 1);
      }
    } else if (ksize == 3 && dilation == 4) {
      if (NTT == 1) DMB_C2(1, 3, 4, 1);
    } else if (ksize == 3 && dilation == 8) {
      if (NTT == 1) DMB_C2(1, 3, 8, 1);
    } else if (ksize == 1) {
      if (NTT == 1) DMB_C2(1, 1, 1, 1);
      if (NTT == 2) DMB_C2(2, 1, 1, 1);
      if (NTT == 4) DMB_C2(4, 1, 1, 1);
    }
  } else if (dilation == 1) {
    if (ksize == 3) {
      if (NTT == 1) DMB_C2(1, 3, 1, 2);
      if (NTT == 2) DMB_C2(2, 3, 1, 2);
    } else if (ksize == 1) {
      if (NTT == 1) DMB_C2(1, 1, 1, 2);
      if (NTT == 2) DMB_C2(2, 1, 1, 2);
    } else if (ksize == 5) {
      if (NTT == 1) DMB_C2(1, 5, 1, 2);
    }
  }
#undef DMB_C2
  return fail(DMB_EUNSUPPORTED, "conv2d: stride 1 with kernel 1 | 3, dilation 1 | 2 (4 | 8 up to 32 output channels), output channels <= 128; "
                                "stride 2 with kernel 1 | 3 (<= 64 output channels) or 5 (<= 32), dilation 1");
}

extern "C" int dmb_conv2d_k3_multi_f32(int njobs, const float* const* x, const float* const* wpack, float* const* y, const int* W,
                                       const int* out_channels_total, int B, int Ci, int Co, int H, void* stream) {
  if (njobs <= 0 || njobs > C2_MAXJOBS || !x || !wpack || !y || !W || !out_channels_total || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0)
    return fail(DMB_EINVAL, "conv2d_multi: bad argument");
  C2Jobs jobs = {};
  jobs.njobs = njobs;
  for (int q = 0; q < njobs; ++q) {
    if (!x[q] || !wpack[q] || !y[q] || W[q] <= 0 || out_channels_total[q] < Co) return fail(DMB_EINVAL, "conv2d_multi: bad job");
    if (W[q] % 4 != 0 || (((uintptr_t)x[q] | (uintptr_t)y[q]) & 15) != 0)
      return fail(DMB_EUNSUPPORTED, "conv2d_multi: rows must be 16-byte aligned (W % 4 == 0)");
    if ((long long)Ci * H * W[q] * 4 >= 0x7fffffffLL || (long long)out_channels_total[q] * H * W[q] * 4 >= 0x7fffffffLL)
      return fail(DMB_EUNSUPPORTED, "conv2d_multi: one batch item must stay below 2 GiB");
    jobs.x[q] = x[q];
    jobs.wp[q] = wpack[q];
    jobs.y[q] = y[q];
    jobs.W[q] = W[q];
    jobs.out_ctot[q] = out_channels_total[q];
  }
  const int NTT = cdiv(Co, 32);
  hipStream_t st = (hipStream_t)stream;
  if (NTT == 1) return launch_conv2d_multi<C2Cfg<1, 3, 1, 1, true>>(jobs, B, Ci, Co, H, st);
  if (NTT == 2) return launch_conv2d_multi<C2Cfg<2, 3, 1, 1, true>>(jobs, B, Ci, Co, H, st);
  if (NTT == 4) return launch_conv2d_multi<C2Cfg<4, 3, 1, 1, true>>(jobs, B, Ci, Co, H, st);
  return fail(DMB_EUNSUPPORTED, "conv2d_multi: output channels <= 64 or 97 .. 128");
}

extern "C" int dmb_conf_phase_conv2d_f32(const float* c, const float* wpack, const float* scale, const float* shift,
                                         const float* w2, float* conf, int B, int Ci, int Hq, int Wq, int nsets, void* stream) {
  if (!c || !wpack || !scale || !shift || !w2 || !conf || B <= 0 || Ci <= 0 || Hq <= 0 || Wq <= 0 || nsets <= 0 || nsets > 8)
    return fail(DMB_EINVAL, "conf_phase_conv2d: bad argument");
  if ((long long)Ci * Hq * Wq * 4 >= 0x7fffffffLL || (long long)16 * Hq * Wq * 4 >= 0x7fffffffLL)
    return fail(DMB_EUNSUPPORTED, "conf_phase_conv2d: one batch item must stay below 2 GiB");
  hipStream_t st = (hipStream_t)stream;
  const bool v16 = Wq % 4 == 0 && !DMB_OPT(3) && (((uintptr_t)c) & 15) == 0;
  // (`res` carries the 64 dot weights, `y` the confidence map, `res_ctot` the number of weight sets: see C2Cfg::DOT)
  return v16 ? launch_conv2d<C2Cfg<4, 3, 1, 1, true, true>>(c, wpack, scale, shift, w2, conf, B, Ci, 128, Hq, Wq, 1, Ci, 0, nsets, st)
             : launch_conv2d<C2Cfg<4, 3, 1, 1, false, true>>(c, wpack, scale, shift, w2, conf, B, Ci, 128, Hq, Wq, 1, Ci, 0, nsets, st);
}

extern "C" int dmb_avgpool2d_f32(const float* x, float* y, int B, int C, int H, int W, int k, int in_channels_total,
                                 int in_ch_offset, void* stream) {
  if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || H / k <= 0 || W / k <= 0 || in_ch_offset < 0 ||
      in_ch_offset + C > in_channels_total)
    return fail(DMB_EINVAL, "avgpool2d: bad argument");
  const int Ho = H / k, Wo = W / k;
  const long long n = (long long)B * C * Ho * Wo;
  hipLaunchKernelGGL(avgpool2d_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, B, C, H, W, k,
                     Ho, Wo, in_channels_total, in_ch_offset);
  return launch_status("avgpool2d launch failed");
}

extern "C" int dmb_bilinear_ac_f32(const float* x, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                   int out_channels_total, int out_ch_offset, void* stream) {
  if (!x || !y || B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || out_ch_offset < 0 ||
      out_ch_offset + C > out_channels_total || B > 65535)
    return fail(DMB_EINVAL, "bilinear: bad argument");
  const long long n = (long long)C * Ho * Wo;
  const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  hipLaunchKernelGGL(bilinear_ac_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, x, y, C, Hi,
                     Wi, Ho, Wo, sh, sw, out_channels_total, out_ch_offset);
  return launch_status("bilinear launch failed");
}

extern "C" int dmb_bilinear_scale_f32(const float* x, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo, float mult,
                                      int out_channels_total, int out_ch_offset, void* stream) {
  if (!x || !y || B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || out_ch_offset < 0 ||
      out_ch_offset + C > out_channels_total || B > 65535)
    return fail(DMB_EINVAL, "bilinear_scale: bad argument");
  const long long n = (long long)C * Ho * Wo;
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  hipLaunchKernelGGL(bilinear_hp_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, x, y, C, Hi,
                     Wi, Ho, Wo, sh, sw, mult, out_channels_total, out_ch_offset);
  return launch_status("bilinear_scale launch failed");
}
