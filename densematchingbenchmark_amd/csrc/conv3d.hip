// 3-D convolution family of the cost aggregators as FP32 implicit GEMMs on the gfx950 matrix cores.
//
//   y[co, p] = sum_{ci, tap} w[co, ci, tap] * x[ci, p + tap]        (M = Cout, N = voxels, K = Cin * 27)
//
// v_mfma_f32_32x32x2_f32 is exact FP32 (bitwise an fmaf chain in k order) and runs at the FP32 vector peak
// (157.3 TFLOP/s), reached from one wave per SIMD -- which a VALU kernel cannot do.  Roles: A = weights
// (row = output channel), B = input voxels (column = voxel), so that every accumulator register holds one
// output channel x 32 consecutive voxels along W and the epilogue stores 128-byte runs into NCDHW.
//
// Data flow per workgroup (256 threads = 4 waves, 2 workgroups per CU so one stages while the other
// computes):  for each chunk of CK input channels: stage the haloed input tile [CK][TZ+2][TY+2][TX+2]
// into LDS (zero padding materialised there) -> every wave runs (CK/2)*27 k-steps; per k-step one
// coalesced 256-B weight-fragment load (L2-resident, prepacked) and MT ds_read_b32 B-fragments feed
// MT*NT MFMAs.  The "voxel" index of a B fragment walks the FLATTENED padded (y, x) plane of the LDS tile,
// so a tap (dz, dy, dx) is a compile-time address offset and 32 lanes always read 32 consecutive floats
// (conflict-free); the 2 halo columns per row are computed and discarded (2/(TX+2) waste).
//
// Reference semantics: dmb/modeling/stereo/layers/basic_layers.py:68-100,160-177 (Conv3d/ConvTranspose3d
// + BatchNorm3d + ReLU factories), cost_processors/utils/hourglass.py:62-86.
#include <type_traits>

#include "dmb_common.h"

// Build-time experiment knob (build.py DMB_BUILD_DEFS): 0 = natural row pairs (r, r + 1) in the stride-1 kernel (see S1Cfg::GSTR).
#ifndef DMB_S1_GSTR
#define DMB_S1_GSTR 1
#endif

#ifndef DMB_EPI_LD
#define DMB_EPI_LD 0   // buffer cache policy of the transposed kernel's epilogue: bit 1 = nt (streaming)
#endif
#ifndef DMB_EPI_ST
#define DMB_EPI_ST 0
#endif

namespace dmb {

// ---------------------------------------------------------------------------------------------------------
// Weight prepack: A fragments in k-step order.
//   wp[((kp * 27 + tap) * NTT + nt) * 64 + lane] = W(co = nt*32 + (lane & 31), ci = 2*kp + (lane >> 5), tap)
// For nn.Conv3d W(co, ci, tap) = w[co][ci][tap]; for nn.ConvTranspose3d W(co, ci, tap) = w[ci][co][tap]; for the data
// gradient of a stride-1 nn.Conv3d W(co, ci, tap) = w[ci][co][26 - tap].
// ---------------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int Cipad,
                                    int transposed) {
  const int NTT = cdiv(Co, 32);   // output channels beyond Co are zero rows (1-channel transposed head of GC-Net)
  const long long total = (long long)Cipad * 27 * NTT * 32;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    long long r = i >> 6;
    const int nt = (int)(r % NTT);
    r /= NTT;
    const int tap = (int)(r % 27);
    const int kp = (int)(r / 27);
    const int co = nt * 32 + (lane & 31);
    const int ci = 2 * kp + (lane >> 5);
    float v = 0.f;  // channels >= Ci are zero padding (the kernels consume Ci rounded up to their chunk size)
    // transposed == 2: data gradient of a stride-1 convolution = the same convolution with the channel roles exchanged and
    // the taps mirrored
    if (ci < Ci && co < Co)
      v = transposed == 2 ? w[((size_t)ci * Co + co) * 27 + 26 - tap]
                          : (transposed ? w[((size_t)ci * Co + co) * 27 + tap] : w[((size_t)co * Ci + ci) * 27 + tap]);
    wp[i] = v;
  }
}

// Many weight tensors in ONE launch (a training step re-packs every unit's forward and data-gradient weights after each optimizer
// update: 52 launches of 4.8 us in a PSMNet step): blockIdx.y = job, the table lives in device memory (dmb_pack_job, dmb_hip.h).
struct PackJob {
  const float* w;
  float* wp;
  int Co, Ci, mode, pad;
};
__global__ void pack_weights_multi_kernel(const PackJob* __restrict__ jobs) {
  const PackJob jb = jobs[blockIdx.y];
  const int Co = jb.Co, Ci = jb.Ci, transposed = jb.mode;
  const int Cipad = (Ci + 7) / 8 * 8;
  const int NTT = cdiv(Co, 32);
  const long long total = (long long)Cipad * 27 * NTT * 32;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    long long r = i >> 6;
    const int nt = (int)(r % NTT);
    r /= NTT;
    const int tap = (int)(r % 27);
    const int kp = (int)(r / 27);
    const int co = nt * 32 + (lane & 31);
    const int ci = 2 * kp + (lane >> 5);
    float v = 0.f;
    if (ci < Ci && co < Co)
      v = transposed == 2 ? jb.w[((size_t)ci * Co + co) * 27 + 26 - tap]
                          : (transposed ? jb.w[((size_t)ci * Co + co) * 27 + tap] : jb.w[((size_t)co * Ci + ci) * 27 + tap]);
    jb.wp[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Shared epilogue: v = acc*scale + shift (+ residual) (relu), scattered into NCDHW.
// ---------------------------------------------------------------------------------------------------------
struct Affine {
  const float* scale;
  const float* shift;
};

// ---------------------------------------------------------------------------------------------------------
// Stride-1 kernel.
// ---------------------------------------------------------------------------------------------------------
// LIN (with G = 16, TX = W, TY = 3): the wave's 32-voxel column tiles are 32 CONSECUTIVE voxels of the (y, x) plane in memory
// order, a workgroup = TZ planes x 64 consecutive voxels (two column tiles per wave).  For the quarter-resolution layer of the
// hourglass ([4, 64, 12, 34, 60]: 97 920 voxels = 382.5 per CU) no box tiling comes near a multiple of 256 equal workgroups (216
// boxes of 4 x 4 x 60 = 84 % of the chip, one wave per SIMD); 64-voxel runs give 4 x 6 x 32 = 768 equal workgroups = exactly three
// per CU, 99.6 % of their column tiles real.  The staged tile is the 3 + 2 rows a 64-voxel run can touch (W >= 32), full width.
template <int CIN_, int COUT_, int TY_, int TX_, int CK_, int WN_, int SCHED_ = 1, int G_ = 0, int PMIN_ = 0, bool LIN_ = false>
struct S1Cfg {
  static constexpr int CIN = CIN_, COUT = COUT_, TY = TY_, TX = TX_, CK = CK_, WN = WN_, SCHED = SCHED_;
  static constexpr bool LIN = LIN_;
  static constexpr int G = G_;                  // row-group width: 0 = flattened tiles, 16 = row pairs, 8 = row quads
  static constexpr bool ROWPAIR = G_ != 0;     // any row-group mapping (takes the 16-byte vector path)
  static constexpr int GR = G_ ? 32 / G_ : 1;  // rows per group
  static constexpr int WZ = 4 / WN;        // waves along z; one output z-slice per wave
  static constexpr int TZ = WZ;
  // Row-pair tiles are staged with 16-byte LDS-DMA words: a staged row starts at the 16-byte aligned column x0 - 4
  // (XOFF = 3 unused floats in front of the halo column) and is a whole number of words long.  A CU retires LDS-DMA
  // at about one lane per clock whatever the word size, so this is 4x fewer TA cycles for the same bytes.
  static constexpr int XOFF = ROWPAIR ? 3 : 0;
  static constexpr int PV = (4 + TX + 1 + 3) / 4 * 4;
  static constexpr int P = ROWPAIR ? (PV > PMIN_ ? PV : PMIN_) : TX + 2;   // padded row pitch (PMIN: bank-friendly pitch)
  static constexpr int ROWS = TY + 2;
  static constexpr int PLANE = ROWS * P;
  static constexpr int ZS = TZ + 2;
  static constexpr int UPR = P / 4, UPC = ZS * ROWS * UPR;   // row-pair path: 16-byte units per row / per channel
  static constexpr int IPC = (UPC + 63) / 64;                // copy instructions per channel
  static constexpr int TR_PITCH = 36;                        // epilogue transposition scratch [32 channels][32 + 4]
  // How the 32 columns (voxels) of an MFMA B tile map onto the LDS tile:
  //  - flattened (default): 32 consecutive positions of the padded (y, x) plane; works for any TX, but the 2 halo
  //    columns per row and the last partial tile are computed and discarded (6 % at TX = 60, TY = 4);
  //  - row group (TX % G == 0, TY % (32 / G) == 0): G columns of 32 / G consecutive rows -- row pairs of 16 (lanes 0-15 =
  //    row r, 16-31 = row r + 1) or row quads of 8: every computed voxel is a real output.  Used when W % TX == 0.
  static constexpr int XS = ROWPAIR ? TX / (G_ ? G_ : 1) : 1;
  static constexpr int MT = LIN ? 2 : (ROWPAIR ? (TY / GR) * XS : (TY * P + 31) / 32);  // 32-voxel column tiles per wave
  // Which rows a row PAIR takes.  A B-fragment read (ds_read_b32) is served 32 lanes at a time, bank = dword address mod 32: lanes
  // 0-15 cover 16 banks of one row, lanes 16-31 the same columns of the pair's other row, GSTR * P floats on.  With the natural
  // pair (rows r, r + 1) and P = 56 (48-column tiles) or 40 (32-column tiles) the second row starts 24 / 8 banks on: eight lanes
  // collide and every such read takes two LDS cycles (counters of the dominant layer: 45 % of its LDS cycles were conflicts).
  // Pairing rows (r, r + 2) of a four-row tile instead -- tile q of a column = rows q and q + 2 -- puts the second row
  // 2 P = 16 (mod 32) banks on: conflict-free.  Same MFMAs on the same operands; only which accumulator holds which row changes
  // (bit-identical: scripts/attic/kbench_s1_ab.py prints a checksum).  What it buys is LDS cycles, not time: 2.389 / 2.396 against
  // 2.386 / 2.391 ms at [4, 32, 48, 136, 240] (noise), 2.236 / 2.240 against 2.244 / 2.243 ms at [4, 32, 48, 96, 312] (-0.3 %) --
  // the LDS pipe is a quarter busy either way and the fragment reads run a k-step ahead of their MFMAs.
  static constexpr int GSTR = (G_ == 16 && TY_ == 4 && !LIN_ && (2 * P) % 32 == 16 && P % 32 != 16 && DMB_S1_GSTR) ? 2 : 1;
  __device__ static constexpr int row_base(int q) { return GSTR == 2 ? q : GR * q; }   // first row of the q-th group of a column
  __device__ static constexpr int lane_off(int j) { return ROWPAIR ? (j / (G_ ? G_ : 1)) * GSTR * P + (j % (G_ ? G_ : 1)) : j; }
  __device__ static constexpr int tile_off(int mt) { return ROWPAIR ? row_base(mt / XS) * P + (mt % XS) * G_ : mt * 32; }
  __device__ static void decode(int mt, int j, int& ly, int& lx, bool& valid) {
    if (ROWPAIR) {
      ly = row_base(mt / XS) + GSTR * (j / (G_ ? G_ : 1));
      lx = (mt % XS) * G_ + j % (G_ ? G_ : 1);
      valid = true;
    } else {
      const int m = mt * 32 + j;
      ly = m / P;
      lx = m - ly * P;
      valid = m < TY * P && lx < TX;
    }
  }
  static constexpr int NTT = COUT / 32;          // 32-channel row tiles in total
  static constexpr int NT = NTT / WN;            // ... per wave
  static constexpr int CH_STRIDE = ZS * PLANE + (ROWPAIR ? 4 : 36);  // + slack read by discarded columns
  static constexpr int NK = (CK / 2) * 27;           // k-steps per chunk
  static constexpr int IN_FLOATS = CK * CH_STRIDE;   // input tile of one chunk
  static constexpr int BUF_FLOATS = IN_FLOATS + NK * NTT * 64;  // + the chunk's weight fragments
  static constexpr int LDS_FLOATS = 2 * BUF_FLOATS;  // double buffered
  // workgroups per CU: three where LDS admits them (row-group tiles: 43-49 KB; the compiler then keeps the kernel inside
  // 168 registers): 32 -> 32 at full resolution 142 -> 146 TFLOP/s
  static constexpr int WPE = (ROWPAIR && LDS_FLOATS * 4 * 3 <= 160 * 1024) ? 3 : 2;
  static_assert(ROWPAIR || P <= 64, "one wave stages one tile row per instruction");
  static_assert(CK % 2 == 0 && COUT % (32 * WN) == 0 && IN_FLOATS % 4 == 0, "shape");
  static_assert(!ROWPAIR || LIN || (G_ % 4 == 0 && TX % (G_ ? G_ : 1) == 0 && TY % GR == 0), "row-group tiles need TX % G == 0 and TY % (32 / G) == 0");
  static_assert(!LIN || (ROWPAIR && TY == 3 && TX % 4 == 0 && TX >= 32), "linear tiles: vector path, 3 + 2 staged rows, full-width rows");
  static_assert(!ROWPAIR || ((CK * IPC) % 4 == 0 && NT == 1 && LDS_FLOATS >= 4 * 32 * TR_PITCH), "row-pair staging / scratch");
};

// Epilogue shared by the three MFMA kernels: v = acc*scale + shift (+ residual) (relu) for the 16 accumulator
// registers of one 32x32 tile; `o` is the voxel offset inside one channel plane, `cstride` the channel stride.
// Residual loads are issued as one batch before the stores (one latency per tile instead of sixteen).
// relu: 0 none, 1 after the residual add (hourglass.py:67-81), 2 before it (GC-Net adds the skip to the activated
// output, aggregators/GCNet.py:108-116).  cvalid: channels >= cvalid are padding rows and are neither read nor written.
template <int VEC>
__device__ __forceinline__ void store_tile(const f32x16 (&a)[VEC], const float (&sc)[16], const float (&sh)[16],
                                           const float* __restrict__ rb, float* __restrict__ yb, int co0, int h,
                                           unsigned cstride, unsigned o, int relu, int cvalid = 1 << 30) {
  float rv[16][VEC];
  if (rb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (co0 + cd_row(r, h) >= cvalid) {
        rv[r][0] = 0.f;
        rv[r][VEC - 1] = 0.f;
        continue;
      }
      const float* rp = rb + (size_t)(co0 + cd_row(r, h)) * cstride + o;
      if (VEC == 2) {
        const float2 t = *reinterpret_cast<const float2*>(rp);
        rv[r][0] = t.x;
        rv[r][VEC - 1] = t.y;
      } else {
        rv[r][0] = *rp;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      v[e] = fmaf(a[e][r], sc[r], sh[r]);
      if (relu == 2) v[e] = fmaxf(v[e], 0.f);
      if (rb) v[e] += rv[r][e];
      if (relu == 1) v[e] = fmaxf(v[e], 0.f);
    }
    if (co0 + cd_row(r, h) >= cvalid) continue;
    float* yp = yb + (size_t)(co0 + cd_row(r, h)) * cstride + o;
    if (VEC == 2)
      *reinterpret_cast<float2*>(yp) = make_float2(v[0], v[VEC - 1]);
    else
      *yp = v[0];
  }
}

// WITH_RES: the residual operand is a compile-time property of the launch (two instantiations per tile shape).  The vector
// epilogue with its residual ring needs 3 - 5 registers more than the 168 that three workgroups per CU leave; as a run-time
// branch that meant spills -- a scratch segment -- for EVERY launch of the kernel, and a kernel with a scratch segment costs
// about 6 us of dispatch gap on either side of each launch (five of the six 32 -> 32 launches of a PSMNet step carry no
// residual).
// STATS (round 6, row-pair tiles of the training path): the epilogue also sums the RAW convolution outputs of its tile and their
// squares per channel -- FP64 per lane, the eight lanes of a channel by a fixed butterfly, the four waves in ascending order -- and
// writes one partial pair per workgroup and channel to stats[(ch * gridDim.x + blockIdx.x) * 2 + {0, 1}]: the batch statistics of
// the BatchNorm that follows (layers/basic_layers.py:68-83 under train()) without the pass over the output that
// dmb_bn_train_stats_f32 makes.  A separate instantiation: the inference kernels do not change.
template <class C, bool WITH_RES, bool STATS = false>
__global__ __launch_bounds__(256, C::WPE) void conv3d_s1_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ res, float* __restrict__ y, int Ci,
                                                           int D, int H, int W, int ntx, int nty, int ntz, int relu,
                                                           double* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // development build only (DMB_DBG is the constant 0 in the release build: these branches do not exist there) -- diagnostics
  // (option 6): 1 = no stores, 2 = no staging after the first chunk, 32 = no barrier in the chunk loop; option 14: start-up stagger
  const int dbg = DMB_DBG((relu >> 8) & 0xff);
  const int stg = DMB_DBG(relu >> 16);
  relu &= 0xff;
  if (stg > 0 && blockIdx.x < 256u * C::WPE) {
    // the first round's workgroups of a CU start together: delay them by their slot on the CU (HW_ID.TG_ID) x stg x 3.4 us so that
    // set-up and epilogue of one fall under the matrix work of the others (scripts/attic/s1_stagger_probe.py: -0.3 % at best)
    const int n = (int)(__builtin_amdgcn_s_getreg((3 << 11) | (16 << 6) | 4) & 15u) * stg;   // HW_REG_HW_ID bits 19:16
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % ntx;     // LIN: ntx = 64-voxel runs per plane, nty = 1
  t /= ntx;
  const int ty = t % nty;
  t /= nty;
  const int tz = t % ntz;
  const int b = t / ntz;
  const int p0 = C::LIN ? tx * 64 : 0;                                   // LIN: first voxel of the run in its plane
  const int x0 = C::LIN ? 0 : tx * C::TX, y0 = C::LIN ? p0 / W : ty * C::TY, z0 = tz * C::TZ;

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int wz = wave / C::WN, wn = wave % C::WN;
  const unsigned HW = (unsigned)H * W, DHW = (unsigned)D * HW;
  const float* xb = x + (size_t)b * Ci * DHW;

  f32x16 acc[C::MT][C::NT];   // started by the first chunk's first k-step (see `chunk`)

  // ---- staging (LDS-DMA).  Work unit = one (channel, z) plane of the haloed tile = ROWS row copies; the CK*ZS
  // planes of a chunk are dealt to the 4 waves in contiguous runs.  All copies of chunk i+1 (input rows + the
  // chunk's weight fragments) are issued before the MFMAs of chunk i and land while they run.
  constexpr int NPL = C::CK * C::ZS;            // planes per chunk
  static_assert(NPL % 4 == 0, "planes are dealt evenly to the 4 waves");
  constexpr int PPW = NPL / 4;                  // planes per wave
  constexpr int WCH = C::NK * C::NTT * 64;      // weight floats per chunk
  static_assert(WCH % 16 == 0, "weights are staged with 16-byte copies, evenly over 4 waves");
  constexpr int WPW = WCH / 16;                 // 16-byte words per wave
  constexpr int WI = (WPW + 63) / 64;           // copy instructions per wave
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(wp, (unsigned)(cdiv(Ci, C::CK) * C::CK * 27 * C::COUT) * 4u);
  const int gx = x0 - 1 + lane;
  const unsigned xvoff = (lane < C::P && gx >= 0 && gx < W) ? (unsigned)gx * 4u : DMA_OOB;
  auto stage = [&](int c0, float* buf) {
    // the resource covers only this chunk's channels: 32-bit offsets then never limit the tensor size
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(xb + (size_t)c0 * DHW, (unsigned)min(C::CK, Ci - c0) * DHW * 4u);
    if constexpr (C::ROWPAIR) {
      // unit = 4 consecutive floats of a staged row; the units of one channel are linear in LDS
      constexpr int IPW = C::CK * C::IPC / 4;   // instructions per wave
#pragma unroll
      for (int q = 0; q < IPW; ++q) {
        const int id = wave * IPW + q, cl = id / C::IPC, qi = id - cl * C::IPC;
        const int u = qi * 64 + lane;
        const int zz = u / (C::ROWS * C::UPR), rr = u - zz * (C::ROWS * C::UPR), yy = rr / C::UPR, sg = rr - yy * C::UPR;
        const int gz = z0 - 1 + zz, gy = y0 - 1 + yy, gxs = x0 - 4 + sg * 4;
        const bool ok = c0 + cl < Ci && gz >= 0 && gz < D && gy >= 0 && gy < H && gxs >= 0 && gxs < W;
        if (u < C::UPC)
          dma16(xrs, ok ? ((unsigned)cl * DHW + (unsigned)gz * HW + (unsigned)gy * W + (unsigned)gxs) * 4u : DMA_OOB,
                0u, buf + cl * C::CH_STRIDE + qi * 256);
      }
    } else if (lane < C::P) {
#pragma unroll
      for (int q = 0; q < PPW; ++q) {
        const int pl = wave * PPW + q, cl = pl / C::ZS, zz = pl - cl * C::ZS;
        const int gz = z0 - 1 + zz;
        const bool zok = gz >= 0 && gz < D && c0 + cl < Ci;
        const unsigned zoff = ((unsigned)cl * DHW + (unsigned)max(gz, 0) * HW) * 4u;
        float* dpl = buf + cl * C::CH_STRIDE + zz * C::PLANE;
#pragma unroll
        for (int yy = 0; yy < C::ROWS; ++yy) {
          const int gy = y0 - 1 + yy;
          const bool ok = zok && gy >= 0 && gy < H;
          dma4(xrs, ok ? xvoff : DMA_OOB, ok ? zoff + (unsigned)gy * W * 4u : 0u, dpl + yy * C::P);
        }
      }
    }
    // weights: WCH/4 16-byte words split evenly over the 4 waves (WPW each, in WI instructions per wave), so that
    // every wave issues the same number of copies and a counted s_waitcnt vmcnt(N) means the same thing in all of them
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int q4 = wave * WPW + i * 64 + lane;
      if (i * 64 + lane < WPW)
        dma16(wrs, (unsigned)q4 * 16u, (unsigned)(c0 / 2) * (27 * C::NTT * 64 * 4), buf + C::IN_FLOATS + (wave * WPW + i * 64) * 4);
    }
  };

  // LIN: where this lane's voxel of each column tile sits in the staged tile (rows y0 - 1 .. y0 + 3, full width)
  int lin_off[C::MT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt) {
    const int pos = p0 + mt * 32 + j, py = pos / W;
    lin_off[mt] = C::LIN ? (py - y0) * C::P + (pos - py * W) : 0;
  }
  const int NC = cdiv(Ci, C::CK);  // a partial last chunk reads zeros (bounds check) against zero-padded weights
  stage(0, lds);
  __syncthreads();
  // One chunk: FIRST = the launch's first chunk, whose first k-step starts the accumulators from the constant 0 operand of the
  // MFMA instead of from registers that would have to be cleared first (96 v_mov per wave, twice with this compiler's loop
  // lowering: 0.5 % of the shared ALU's time).
  auto chunk = [&](auto first_tag, int ci) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const float* cur = lds + (ci & 1) * C::BUF_FLOATS;
    if (ci + 1 < NC && !(dbg & 2)) stage((ci + 1) * C::CK, lds + ((ci + 1) & 1) * C::BUF_FLOATS);  // lands while we compute
    // ---- NK k-steps on the current buffer; A and B fragments register double-buffered one k-step ahead ----
    const float* abase = cur + C::IN_FLOATS + (wn * C::NT) * 64 + lane;
    const float* bbase = cur + h * C::CH_STRIDE + wz * C::PLANE + (C::LIN ? 0 : C::lane_off(j)) + C::XOFF;
    float af[2][C::NT], bf[2][C::MT];
    auto load_frag = [&](int ks, float (&a)[C::NT], float (&bq)[C::MT]) {
      const int cp = ks / 27, tap = ks % 27;
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) a[nt] = abase[(ks * C::NTT + nt) * 64];
      const float* bp = bbase + 2 * cp * C::CH_STRIDE + dz * C::PLANE + dy * C::P + dx;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) bq[mt] = bp[C::LIN ? lin_off[mt] : C::tile_off(mt)];
    };
    load_frag(0, af[0], bf[0]);
#pragma unroll
    for (int ks = 0; ks < C::NK; ++ks) {
      if (ks + 1 < C::NK) load_frag(ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
      if (C::SCHED != 0) __builtin_amdgcn_sched_barrier(0);  // keep the next step's LDS reads ahead of this step's MFMAs
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) {
          if (FIRST && ks == 0) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[mt][nt] = DMB_MFMA(af[0][nt], bf[0][mt], zero);
          } else {
            acc[mt][nt] = DMB_MFMA(af[ks & 1][nt], bf[ks & 1][mt], acc[mt][nt]);
          }
        }
      if (C::SCHED != 0) __builtin_amdgcn_sched_barrier(0);
    }
    if (!(dbg & 32)) __syncthreads();  // the compiler drains the DMA (vmcnt(0)) here: next buffer complete, current one free
  };
  chunk(std::true_type{}, 0);
  for (int ci = 1; ci < NC; ++ci) chunk(std::false_type{}, ci);

  // ---- epilogue ----
  if (dbg & 1) return;
  const int gz = z0 + wz;
  float* yb = y + (size_t)b * C::COUT * DHW;
  const float* rb = res ? res + (size_t)b * C::COUT * DHW : nullptr;
  if constexpr (C::ROWPAIR) {
    // Each 32 x 32 accumulator tile goes through a per-wave LDS scratch (the chunk buffers are free after the last
    // barrier) so that a lane owns 4 consecutive x of one channel: 16-byte stores / residual loads, 4x fewer
    // vector-memory instructions.  Buffer addressing with an out-of-range offset for lanes outside the volume keeps
    // the code branch-free; residual loads run one tile ahead of the stores.
    float* my = lds + wave * (32 * C::TR_PITCH);
    const __amdgpu_buffer_rsrc_t yrs = make_rsrc(yb, (unsigned)C::COUT * DHW * 4u);
    const __amdgpu_buffer_rsrc_t rrs = make_rsrc(rb ? rb : yb, (unsigned)C::COUT * DHW * 4u);
    const int px = (lane & 7) * 4;
    const float lo = relu == 1 ? 0.f : -__builtin_inff();    // ReLU after the residual add
    const float lo2 = relu == 2 ? 0.f : -__builtin_inff();   // ReLU before it (GC-Net)
    float sc4[4], sh4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int co = wn * 32 + k * 8 + (lane >> 3);
      sc4[k] = scale ? scale[co] : 1.f;
      sh4[k] = shift ? shift[co] : 0.f;
    }
    // a tile's four 16-byte words per lane are 8 channels apart: one per-lane offset per tile (out of range for lanes outside
    // the volume; adding the channel steps keeps it out of range) instead of four -- the eight offset registers of two tiles in
    // flight were what pushed this epilogue past 168 registers (3 spills = a scratch segment for the kernel, and a kernel with
    // a scratch segment costs ~6 us of dispatch gap on either side of every launch)
    const unsigned kstep = 8u * DHW * 4u;
    auto offset0 = [&](int mt) {
      if constexpr (C::LIN) {   // 4 consecutive voxels of the run (H * W % 4 == 0: a word never straddles the end of the plane)
        const unsigned pos = (unsigned)(p0 + mt * 32 + px);
        return (gz < D && pos < HW) ? ((unsigned)(wn * 32 + (lane >> 3)) * DHW + (unsigned)gz * HW + pos) * 4u : DMA_OOB;
      }
      const int gy = y0 + C::row_base(mt / C::XS) + C::GSTR * (px / C::G), gxo = x0 + (mt % C::XS) * C::G + px % C::G;
      const bool inb = gz < D && gy < H && gxo < W;
      return inb ? ((unsigned)(wn * 32 + (lane >> 3)) * DHW + (unsigned)gz * HW + (unsigned)gy * W + (unsigned)gxo) * 4u : DMA_OOB;
    };
    auto run = [&](auto has_res) {
      constexpr bool HAS_RES = decltype(has_res)::value;
      // Skip-operand loads run ahead of the stores in a ring of three register sets: tile 1 is requested when tile 0 is
      // processed, from then on TWO tiles ahead -- the accumulator registers of the tiles already written out pay for the third
      // set, so the register peak stays where the first tile puts it (96 accumulators + 2 sets).
      unsigned off0[C::MT];
      u32x4 rv[3][4];
      double ssum[STATS ? 4 : 1], ssq[STATS ? 4 : 1];   // this lane's channels k * 8 + (lane >> 3)
      (void)ssum, (void)ssq;
      if constexpr (STATS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ssum[k] = ssq[k] = 0.0;
      }
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) off0[mt] = 0;
      off0[0] = offset0(0);
      auto request = [&](int t) {
        off0[t] = offset0(t);
        if constexpr (HAS_RES) {
#pragma unroll
          for (int k = 0; k < 4; ++k) rv[t % 3][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)(off0[t] + k * kstep), 0, 0);
        }
      };
      request(0);
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
        if (mt <= 1 && mt + 1 < C::MT) request(mt + 1);   // tiles 1 and 2: one ahead (at mt = 0 all six accumulator tiles are live)
        if (mt >= 1 && mt + 2 < C::MT) request(mt + 2);   // from tile 3 on: two ahead
#pragma unroll
        for (int r = 0; r < 16; ++r) my[cd_row(r, h) * C::TR_PITCH + j] = acc[mt][0][r];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float4 v = *reinterpret_cast<const float4*>(my + (k * 8 + (lane >> 3)) * C::TR_PITCH + px);
          if constexpr (STATS) {   // four voxels of one channel: FP32 inside the word, FP64 across words
            if (off0[mt] != DMA_OOB) {
              ssum[k] += (double)((v.x + v.y) + (v.z + v.w));
              ssq[k] += (double)fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, v.w * v.w)));
            }
          }
          v.x = fmaxf(fmaf(v.x, sc4[k], sh4[k]), lo2);
          v.y = fmaxf(fmaf(v.y, sc4[k], sh4[k]), lo2);
          v.z = fmaxf(fmaf(v.z, sc4[k], sh4[k]), lo2);
          v.w = fmaxf(fmaf(v.w, sc4[k], sh4[k]), lo2);
          if constexpr (HAS_RES) {   // (not __builtin_bit_cast on a vector element: this clang reads element 0 for every index)
            v.x += __uint_as_float(rv[mt % 3][k].x);
            v.y += __uint_as_float(rv[mt % 3][k].y);
            v.z += __uint_as_float(rv[mt % 3][k].z);
            v.w += __uint_as_float(rv[mt % 3][k].w);
          }
          u32x4 o;
          o.x = __float_as_uint(fmaxf(v.x, lo));
          o.y = __float_as_uint(fmaxf(v.y, lo));
          o.z = __float_as_uint(fmaxf(v.z, lo));
          o.w = __float_as_uint(fmaxf(v.w, lo));
          __builtin_amdgcn_raw_buffer_store_b128(o, yrs, (int)(off0[mt] + k * kstep), 0, 0);   // (streaming stores: step +0.3 %)
        }
      }
      if constexpr (STATS) {
        // the eight lanes of a channel (lane & 7), then the four waves (z-slices) through LDS behind the transposition scratch
        static_assert(C::NT == 1 && C::LDS_FLOATS >= 4 * 32 * C::TR_PITCH + 4 * 32 * 4, "statistics scratch");
        double* sred = reinterpret_cast<double*>(lds + 4 * 32 * C::TR_PITCH);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int o = 1; o < 8; o <<= 1) {
            ssum[k] += __shfl_xor(ssum[k], o, 64);
            ssq[k] += __shfl_xor(ssq[k], o, 64);
          }
          if ((lane & 7) == 0) {
            sred[(wave * 32 + k * 8 + (lane >> 3)) * 2] = ssum[k];
            sred[(wave * 32 + k * 8 + (lane >> 3)) * 2 + 1] = ssq[k];
          }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
          const int ch = threadIdx.x >> 1, st = threadIdx.x & 1;
          const double tot = ((sred[(0 * 32 + ch) * 2 + st] + sred[(1 * 32 + ch) * 2 + st]) + sred[(2 * 32 + ch) * 2 + st]) + sred[(3 * 32 + ch) * 2 + st];
          stats[((size_t)(wn * 32 + ch) * gridDim.x + blockIdx.x) * 2 + st] = tot;
        }
      }
    };
    run(std::integral_constant<bool, WITH_RES>{});
    return;
  }
  if (gz >= D) return;
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt) {
    const int co0 = (wn * C::NT + nt) * 32;
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + cd_row(r, h);
      sc[r] = scale ? scale[co] : 1.f;
      sh[r] = shift ? shift[co] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
      int ly, lx;
      bool valid;
      C::decode(mt, j, ly, lx, valid);
      const int gy = y0 + ly, gxo = x0 + lx;
      if (valid && gy < H && gxo < W) {
        const f32x16 a1[1] = {acc[mt][nt]};
        store_tile<1>(a1, sc, sh, rb, yb, co0, h, DHW, (unsigned)gz * HW + (unsigned)gy * W + gxo, relu);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Stride-2 kernel (k3, pad 1): input voxel = 2*out - 1 + tap.  LDS tile rows are split by y parity so that
// a flattened output index m = ly*(TX+1) + lx maps to LDS offset 2*m + const(tap); the x stride of 2 floats
// is a harmless 2-way bank conflict (LDS has > 4x headroom next to a 64-cycle MFMA).
// ---------------------------------------------------------------------------------------------------------
// V16 (W % 4 == 0, aligned base): rows are staged with 16-byte LDS-DMA words from the aligned column 2*x0 - 4 (XOFF = 3
// unused floats in front of the halo column), row pitch 64 floats = output-position pitch 32: with dword copies a
// stride-2 tile (4x the input per output) cost more TA cycles than its MFMAs cost matrix-core cycles.
// WM: wave groups along the output positions.  WM = 2 makes a workgroup of EIGHT waves on the same LDS tile, each with half of
// the tile's 32-position column tiles (32 instead of 64 accumulator registers): two workgroups per CU (the LDS bound) then
// put four waves on every SIMD instead of two, so that one wave's staging waits and epilogue hide behind three others.
// VEP (with V16): the epilogue goes through a per-wave LDS scratch so that a lane owns TWO adjacent output columns of one channel
// and stores them as one 8-byte word (16 bytes would need 4-column alignment, and a 30-column tile starts on an odd pair in every
// second tile): a CU retires vector-memory lanes at about one per clock whatever their width, and the dword epilogue's 50 M lane
// stores of the 32 -> 64 layer were 0.08 of its 0.72 ms (diagnostic switch 1).
template <int CIN_, int COUT_, int TY_, int TX_, int CK_, int WN_, bool V16_ = false, int WM_ = 1, bool VEP_ = false>
struct S2Cfg {
  static constexpr int CIN = CIN_, COUT = COUT_, TY = TY_, TX = TX_, CK = CK_, WN = WN_;
  static constexpr bool V16 = V16_, VEP = V16_ && VEP_;
  static constexpr int TR_PITCH = 40;          // epilogue transposition scratch [32 channels][32 + 8]: = 8 (mod 16), see the kernel
  static constexpr int WM = WM_, NWAVES = 4 * WM_, NTHREADS = 64 * NWAVES;
  static constexpr int WZ = 4 / WN;
  static constexpr int TZ = WZ;
  static constexpr int XOFF = V16 ? 3 : 0;
  static constexpr int PO = V16 ? (XOFF + 2 * TX + 1 + 7) / 8 * 4 : TX + 1;   // output-position pitch
  static constexpr int R = 2 * PO;             // LDS row pitch (input columns 2*x0-1 .. 2*x0+2*TX-1, padded)
  static constexpr int PYPL = (TY + 1) * R;    // one y-parity plane
  static constexpr int ZPL = 2 * PYPL;         // one input z-slice
  static constexpr int ZS = 2 * TZ + 1;
  static constexpr int INROWS = 2 * TY + 1;
  static constexpr int INCOLS = 2 * TX + 1;
  static constexpr int MT = (TY * PO + 31) / 32;
  static constexpr int MTW = MT / WM;          // column tiles per wave
  static constexpr int NTT = COUT / 32;
  static constexpr int NT = NTT / WN;
  static constexpr int CH_STRIDE = ZS * ZPL + 72;
  static constexpr int UPR = R / 4, UPC = ZS * ZPL / 4;      // vector path: 16-byte units per row / per channel
  static constexpr int IPC = (UPC + 63) / 64;                // copy instructions per channel
  static constexpr int NK = (CK / 2) * 27;
  static constexpr int IN_FLOATS = CK * CH_STRIDE;
  static constexpr int BUF_FLOATS = IN_FLOATS + NK * NTT * 64;
  static constexpr int LDS_FLOATS = 2 * BUF_FLOATS;  // double buffered
  static constexpr int WPE = (LDS_FLOATS * 4 * 2 <= 160 * 1024) ? 2 : 1;  // workgroups per CU the LDS admits
  static_assert(CK % 2 == 0 && COUT % (32 * WN) == 0 && IN_FLOATS % 4 == 0, "shape");
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
  static_assert(MT % WM == 0, "the column tiles are dealt evenly to the wave groups");
  static_assert(!VEP || (PO % 2 == 0 && TX % 2 == 0 && LDS_FLOATS >= NWAVES * 32 * TR_PITCH), "pair epilogue: even tiles, scratch");
};

template <class C>
// (HIP: the second launch-bounds figure is the minimum number of WAVES PER SIMD, which for four-wave workgroups equals the
// workgroups per CU)
__global__ __launch_bounds__(C::NTHREADS, C::WPE * C::WM) void conv3d_s2_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ res, float* __restrict__ y, int Ci,
                                                           int D, int H, int W, int Do, int Ho, int Wo, int ntx, int nty,
                                                           int ntz, int relu) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % ntx;
  t /= ntx;
  const int ty = t % nty;
  t /= nty;
  const int tz = t % ntz;
  const int b = t / ntz;
  const int x0 = tx * C::TX, y0 = ty * C::TY, z0 = tz * C::TZ;  // output coordinates
  const int dbg = DMB_DBG(relu >> 8);   // development build only (option 6): 1 = no stores, 2 = no staging after the first chunk
  relu &= 0xff;

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int wm = wave / 4, w4 = wave % 4;      // wave group along the positions; (z, channel tile) inside the group
  const int wz = w4 / C::WN, wn = w4 % C::WN;
  const unsigned HW = (unsigned)H * W, DHW = (unsigned)D * HW;
  const float* xb = x + (size_t)b * Ci * DHW;

  f32x16 acc[C::MTW][C::NT];
#pragma unroll
  for (int mt = 0; mt < C::MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // ---- staging (LDS-DMA): unit = (channel, input z-slice, 64-column pass) = INROWS row copies; dealt to the
  // waves in contiguous runs (see conv3d_s1_kernel).  Rows go to the y-parity plane they belong to.
  constexpr int NPASS = (C::INCOLS + 63) / 64;
  constexpr int NUNIT = C::CK * C::ZS * NPASS;
  constexpr int UPW = (NUNIT + C::NWAVES - 1) / C::NWAVES;
  constexpr int WCH = C::NK * C::NTT * 64;
  constexpr int WV4 = (WCH / 4 + C::NTHREADS - 1) / C::NTHREADS;
  static_assert(WCH % 4 == 0, "weights are staged with 16-byte copies");
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(wp, (unsigned)(cdiv(Ci, C::CK) * C::CK * 27 * C::COUT) * 4u);
  // Vector path: the per-lane source offsets of a chunk's copies depend on the tile only, not on the chunk (the chunk's
  // channels are selected by the resource base), so they are computed ONCE here.  Recomputing the unit -> (z, parity, row,
  // column) decode and the bounds tests for every copy of every chunk cost about two vector instructions per MFMA.
  constexpr int S_NI = C::CK * C::IPC, S_IPW = (S_NI + C::NWAVES - 1) / C::NWAVES;
  unsigned xoff[C::V16 ? S_IPW : 1];
  if constexpr (C::V16) {
#pragma unroll
    for (int q = 0; q < S_IPW; ++q) {
      const int id = wave * S_IPW + q;
      const int cl = id / C::IPC, qi = id - cl * C::IPC;
      const int u = qi * 64 + lane;
      const int zz = u / (C::ZPL / 4), r1 = u - zz * (C::ZPL / 4), pp = r1 / (C::PYPL / 4), r2 = r1 - pp * (C::PYPL / 4);
      const int row = r2 / C::UPR, sg = r2 - row * C::UPR;
      const int ry = 2 * row + pp;
      const int gz = 2 * z0 - 1 + zz, gy = 2 * y0 - 1 + ry, gxs = 2 * x0 - 4 + sg * 4;
      const bool ok = ry < C::INROWS && gz >= 0 && gz < D && gy >= 0 && gy < H && gxs >= 0 && gxs < W;
      // (a channel past Ci lies beyond the chunk's resource: the bounds check returns zeros for it)
      xoff[q] = ok ? ((unsigned)cl * DHW + (unsigned)gz * HW + (unsigned)gy * W + (unsigned)gxs) * 4u : DMA_OOB;
    }
  }
  auto stage = [&](int c0, float* buf) {
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(xb + (size_t)c0 * DHW, (unsigned)min(C::CK, Ci - c0) * DHW * 4u);
    if constexpr (C::V16) {
      // unit = 4 consecutive floats of a staged row; the units of one channel ([z][y parity][row][64]) are linear in LDS
#pragma unroll
      for (int q = 0; q < S_IPW; ++q) {
        const int id = wave * S_IPW + q;
        if (S_NI % C::NWAVES == 0 || id < S_NI) {
          const int cl = id / C::IPC, qi = id - cl * C::IPC;
          if (qi * 64 + lane < C::UPC) dma16(xrs, xoff[q], 0u, buf + cl * C::CH_STRIDE + qi * 256);
        }
      }
    } else {
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
      const int uid = wave * UPW + q, pl = uid / NPASS, pass = uid - pl * NPASS;
      if (uid >= NUNIT) continue;
      const int cl = pl / C::ZS, zz = pl - cl * C::ZS;
      const int gz = 2 * z0 - 1 + zz, col = pass * 64 + lane, gx = 2 * x0 - 1 + col;
      const bool zok = gz >= 0 && gz < D && c0 + cl < Ci;
      const unsigned xvoff = (col < C::INCOLS && gx >= 0 && gx < W) ? (unsigned)gx * 4u : DMA_OOB;
      const unsigned zoff = ((unsigned)cl * DHW + (unsigned)max(gz, 0) * HW) * 4u;
      float* dpl = buf + cl * C::CH_STRIDE + zz * C::ZPL + pass * 64;
      if (col < C::R) {
#pragma unroll
        for (int ry = 0; ry < C::INROWS; ++ry) {
          const int gy = 2 * y0 - 1 + ry;
          const bool ok = zok && gy >= 0 && gy < H;
          dma4(xrs, ok ? xvoff : DMA_OOB, ok ? zoff + (unsigned)gy * W * 4u : 0u,
               dpl + (ry & 1) * C::PYPL + (ry >> 1) * C::R);
        }
      }
    }
    }
#pragma unroll
    for (int i = 0; i < WV4; ++i) {
      const int q4 = i * C::NTHREADS + threadIdx.x;
      if (q4 < WCH / 4)
        dma16(wrs, (unsigned)q4 * 16u, (unsigned)(c0 / 2) * (27 * C::NTT * 64 * 4), buf + C::IN_FLOATS + (i * C::NTHREADS + wave * 64) * 4);
    }
  };

  const int NC = cdiv(Ci, C::CK);  // a partial last chunk reads zeros (bounds check) against zero-padded weights
  stage(0, lds);
  __syncthreads();
  for (int ci = 0; ci < NC; ++ci) {
    const float* cur = lds + (ci & 1) * C::BUF_FLOATS;
    if (ci + 1 < NC && !(dbg & 2)) stage((ci + 1) * C::CK, lds + ((ci + 1) & 1) * C::BUF_FLOATS);
    const float* abase = cur + C::IN_FLOATS + (wn * C::NT) * 64 + lane;
    const float* bbase = cur + h * C::CH_STRIDE + (2 * wz) * C::ZPL + 2 * j + C::XOFF + wm * C::MTW * 64;
    float af[2][C::NT], bf[2][C::MTW];
    auto load_frag = [&](int ks, float (&a)[C::NT], float (&bq)[C::MTW]) {
      const int cp = ks / 27, tap = ks % 27;
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) a[nt] = abase[(ks * C::NTT + nt) * 64];
      const float* bp = bbase + 2 * cp * C::CH_STRIDE + dz * C::ZPL + (dy & 1) * C::PYPL + (dy >> 1) * C::R + dx;
#pragma unroll
      for (int mt = 0; mt < C::MTW; ++mt) bq[mt] = bp[mt * 64];
    };
    load_frag(0, af[0], bf[0]);
#pragma unroll
    for (int ks = 0; ks < C::NK; ++ks) {
      if (ks + 1 < C::NK) load_frag(ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < C::MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) acc[mt][nt] = DMB_MFMA(af[ks & 1][nt], bf[ks & 1][mt], acc[mt][nt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  const int gz = z0 + wz;
  if (gz >= Do || (dbg & 1)) return;
  const unsigned HWo = (unsigned)Ho * Wo, DHWo = (unsigned)Do * HWo;
  float* yb = y + (size_t)b * C::COUT * DHWo;
  const float* rb = res ? res + (size_t)b * C::COUT * DHWo : nullptr;
  if constexpr (C::VEP) {
    // Each 32 x 32 accumulator tile (32 positions of one output row x 32 channels) goes through this wave's scratch (the chunk
    // buffers are free after the last barrier): lane = (channel c4 of a group of 4, position pair p2), 8 groups = 8 words of 8
    // bytes.  Scratch pitch 40: the two lane halves of the write (channel rows 4 apart) and the four channel rows of a read land
    // on disjoint bank groups.  Branch-free: lanes outside the tile / volume get an out-of-range buffer offset.
    float* my = lds + wave * (32 * C::TR_PITCH);
    const __amdgpu_buffer_rsrc_t yrs = make_rsrc(yb, (unsigned)C::COUT * DHWo * 4u);
    const __amdgpu_buffer_rsrc_t rrs = make_rsrc(rb ? rb : yb, (unsigned)C::COUT * DHWo * 4u);
    const int p2 = (lane & 15) * 2, c4 = lane >> 4;
    const float lo = relu == 1 ? 0.f : -__builtin_inff();    // ReLU after the residual add
    const float lo2 = relu == 2 ? 0.f : -__builtin_inff();   // ReLU before it (GC-Net)
    const unsigned kstep = 4u * DHWo * 4u;                   // the 8 words of a lane are 4 channels apart
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) {
      const int co0 = (wn * C::NT + nt) * 32;
      float sc8[8], sh8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sc8[k] = scale ? scale[co0 + k * 4 + c4] : 1.f;
        sh8[k] = shift ? shift[co0 + k * 4 + c4] : 0.f;
      }
#pragma unroll
      for (int mt = 0; mt < C::MTW; ++mt) {
        const int m = (wm * C::MTW + mt) * 32 + p2;
        const int ly = m / C::PO, lx = m - ly * C::PO;
        const int gy = y0 + ly, gxo = x0 + lx;
        const bool ok = m < C::TY * C::PO && lx < C::TX && gy < Ho && gxo < Wo;   // (PO, TX, Wo even: a pair is in or out as a whole)
        const unsigned off = ok ? ((unsigned)(co0 + c4) * DHWo + (unsigned)gz * HWo + (unsigned)gy * Wo + (unsigned)gxo) * 4u : DMA_OOB;
        u32x2 rv[8];
        if (rb) {
#pragma unroll
          for (int k = 0; k < 8; ++k) rv[k] = __builtin_amdgcn_raw_buffer_load_b64(rrs, (int)(off + k * kstep), 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) my[cd_row(r, h) * C::TR_PITCH + j] = acc[mt][nt][r];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float2 v = *reinterpret_cast<const float2*>(my + (k * 4 + c4) * C::TR_PITCH + p2);
          v.x = fmaxf(fmaf(v.x, sc8[k], sh8[k]), lo2);
          v.y = fmaxf(fmaf(v.y, sc8[k], sh8[k]), lo2);
          if (rb) {
            v.x += __uint_as_float(rv[k].x);
            v.y += __uint_as_float(rv[k].y);
          }
          u32x2 o;
          o.x = __float_as_uint(fmaxf(v.x, lo));
          o.y = __float_as_uint(fmaxf(v.y, lo));
          __builtin_amdgcn_raw_buffer_store_b64(o, yrs, (int)(off + k * kstep), 0, 0);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt) {
    const int co0 = (wn * C::NT + nt) * 32;
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + cd_row(r, h);
      sc[r] = scale ? scale[co] : 1.f;
      sh[r] = shift ? shift[co] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < C::MTW; ++mt) {
      const int m = (wm * C::MTW + mt) * 32 + j;
      const int ly = m / C::PO, lx = m - ly * C::PO;
      const int gy = y0 + ly, gxo = x0 + lx;
      if (m < C::TY * C::PO && lx < C::TX && gy < Ho && gxo < Wo) {
        const f32x16 a1[1] = {acc[mt][nt]};
        store_tile<1>(a1, sc, sh, rb, yb, co0, h, DHWo, (unsigned)gz * HWo + (unsigned)gy * Wo + gxo, relu);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Transposed convolution k3 s2 p1 op1:  y[2i - 1 + k] += x[i] * w[k]  per axis.  An output of parity 0 (even)
// sees k=1 from i=o/2; parity 1 (odd) sees k=2 from i=(o-1)/2 and k=0 from i=(o+1)/2.  One workgroup =
// (input-resolution tile, z parity, y parity); both x parities are accumulated by the same wave so that a
// lane owns two adjacent outputs and stores them as one 8-byte word.  No zero-insertion, no wasted MACs.
// ---------------------------------------------------------------------------------------------------------
// V16: rows are 16-byte aligned (W % 4 == 0, aligned base): the tile is staged with 16-byte LDS-DMA words (row pitch
// rounded up to 64 floats).  A CU retires LDS-DMA at about one LANE per clock whatever the word size, and with dword
// copies this kernel spent more TA cycles on staging than matrix-core cycles on arithmetic.
// VEPI (with V16): the 16-byte epilogue through the LDS scratch; a separate instantiation, not a run-time switch -- with both
// epilogues in one kernel the register allocation of the one not taken spilled into the other (the scalar form ran 1.6x
// slower once the vector form was added next to it).
template <int CIN_, int COUT_, int TY_, int TX_, int CK_, int WN_, bool V16_ = false, bool VEPI_ = false>
struct DCfg {
  static constexpr int CIN = CIN_, COUT = COUT_, TY = TY_, TX = TX_, CK = CK_, WN = WN_;
  static constexpr bool V16 = V16_, VEPI = V16_ && VEPI_;
  static constexpr int WZ = 4 / WN;
  static constexpr int TZ = WZ;
  static constexpr int P = V16 ? (TX + 1 + 3) / 4 * 4 : TX + 1;
  static constexpr int ROWS = TY + 1;
  static constexpr int PLANE = ROWS * P;
  static constexpr int ZS = TZ + 1;
  static constexpr int MT = (TY * P + 31) / 32;
  static constexpr int NTT = COUT / 32;
  static constexpr int NT = NTT / WN;
  static constexpr int CH_STRIDE = ZS * PLANE + (V16 ? 4 : 36);   // = 4 (mod 32): the two lane halves hit disjoint banks
  static constexpr int IN_FLOATS = CK * CH_STRIDE;
  static constexpr int UPR = P / 4, UPC = ZS * ROWS * UPR;    // vector path: 16-byte units per row / per channel
  static constexpr int IPC = (UPC + 63) / 64;                 // copy instructions per channel
  static constexpr int RUN = 9 * NTT * 64;                    // weight floats of one (channel pair, kz)
  static constexpr int W_FLOATS = (CK / 2) * 2 * RUN;         // worst case: two kz taps (odd output z)
  static constexpr int BUF_FLOATS = IN_FLOATS + W_FLOATS;
  static constexpr int LDS_FLOATS = 2 * BUF_FLOATS;           // double buffered
  static constexpr int AFF_FLOATS = 2 * COUT;                 // scale / shift table behind the chunk buffers
  // Vector epilogue (V16): PCH channels x 64 output columns of one output row go through a per-wave LDS scratch so that
  // a lane stores 4 consecutive x.  The scratch is the part of the just-consumed chunk buffer that only this wave's own
  // copies write (its CK/4 channels), or a dedicated region behind the affine table where that part is too small.
  static constexpr int SCR_PITCH = 68;
  static constexpr int PRIV_FLOATS = (CK / 4) * CH_STRIDE;
  static constexpr int PCH = PRIV_FLOATS >= 16 * SCR_PITCH ? 16 : 8;
  static constexpr bool SCR_PRIVATE = PRIV_FLOATS >= PCH * SCR_PITCH;
  static constexpr int SCR_FLOATS = (VEPI && !SCR_PRIVATE) ? 4 * PCH * SCR_PITCH : 0;
  static constexpr int WPE = ((LDS_FLOATS + AFF_FLOATS + SCR_FLOATS) * 4 * 2 <= 160 * 1024 && MT * NT <= 2) ? 2 : 1;  // workgroups per CU
  static_assert(P <= 64, "one wave stages one tile row per instruction");
  static_assert(CK % 2 == 0 && COUT % (32 * WN) == 0 && IN_FLOATS % 4 == 0 && (!V16 || (CH_STRIDE % 4 == 0 && TX % 4 == 0)), "shape");
  static_assert((LDS_FLOATS + AFF_FLOATS + SCR_FLOATS) * 4 <= 160 * 1024, "LDS budget");
  static_assert(!V16 || P == 32 || P == 64, "a 32-position tile of the vector epilogue is part of one input row");
};

// Body for one z parity PZ.  A work item = (input-resolution tile, z parity): the outputs of z parity PZ for BOTH y
// parities and BOTH x parities (4 accumulator sets), so the staged input tile is used by every tap that can touch it:
// per (channel pair, kz) unit 9*MT*NT MFMAs against 4*MT B reads and 9*NT A reads.
// The kernel is PERSISTENT: a workgroup walks tiles first, first + stride, ... and the chunk pipeline runs across
// tiles (the first chunk of the next tile is copied while the last chunk of this one is multiplied).  A work item
// here lasts only 40-75 k cycles, so a per-item prologue (DMA latency with idle matrix cores) and a workgroup
// re-dispatch per item would cost 15-25 %.
template <class C, int PZ>
__device__ __forceinline__ void deconv_body(float* lds, const float* __restrict__ x, const float* __restrict__ wp,
                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                            const float* __restrict__ res, float* __restrict__ y, int Ci, int D, int H,
                                            int W, int ntx, int nty, int ntz, int first, int stride, int ntiles,
                                            int relu, int cvalid) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int wz = wave / C::WN, wn = wave % C::WN;
  const unsigned HW = (unsigned)H * W, DHW = (unsigned)D * HW;
  if (first >= ntiles) return;
  const int my_tiles = (ntiles - first + stride - 1) / stride;
  const int dbg = DMB_DBG(relu >> 8);   // development build only (option 6)
  relu &= 0xff;

  struct Tile {
    int b, x0, y0, z0;
  };
  auto tile_at = [&](int it) {
    int t = first + it * stride;
    Tile tl;
    tl.x0 = (t % ntx) * C::TX;
    t /= ntx;
    tl.y0 = (t % nty) * C::TY;
    t /= nty;
    tl.z0 = (t % ntz) * C::TZ;
    tl.b = t / ntz;
    return tl;
  };

  // ---- staging: whole channels per wave; weights: per channel pair the 9 (ky, kx) taps of each needed kz, laid
  // out [cp][az][ky][kx][nt][64] with az = 0 -> kz = (PZ ? 2 : 1), az = 1 -> kz = 0.
  constexpr int NAZ = 1 + PZ;
  static_assert(C::CK % 4 == 0, "each wave stages whole channels");
  constexpr int CPW = C::CK / 4;                        // channels per wave per chunk
  constexpr int WCH4 = (C::CK / 2) * NAZ * C::RUN / 4;  // 16-byte copies per chunk
  constexpr int WV4 = (WCH4 + 255) / 256;
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(wp, (unsigned)(cdiv(Ci, C::CK) * C::CK * 27 * C::COUT) * 4u);
  // Vector path: a copy's per-lane source offset depends on the tile only (the chunk's channels are selected by the
  // resource base), so the unit -> (z, row, column) decode and the bounds tests run once per tile (tile_offsets), not once
  // per copy and chunk: they cost more vector-instruction issue slots than the chunk's MFMAs left free.
  unsigned woff[WV4];   // weight copies: per-lane source offset inside a chunk's block (the chunk goes into the scalar offset)
#pragma unroll
  for (int i = 0; i < WV4; ++i) {
    const int q4 = i * 256 + (int)threadIdx.x;
    const int run = q4 / (C::RUN / 4), off = q4 - run * (C::RUN / 4);
    const int cp = run / NAZ, az = run - cp * NAZ;
    const int kz = PZ ? (az ? 0 : 2) : 1;
    woff[i] = (unsigned)(((cp * 27 + kz * 9) * C::NTT * 64) * 4 + off * 16);
  }
  constexpr int NQ = C::V16 ? CPW * C::IPC : 1;
  auto tile_offsets = [&](const Tile& tl, unsigned (&o)[NQ]) {
    if constexpr (C::V16) {
#pragma unroll
      for (int cc = 0; cc < CPW; ++cc) {
        const int cl = wave * CPW + cc;
#pragma unroll
        for (int q = 0; q < C::IPC; ++q) {
          const int u = q * 64 + lane;
          const int zz = u / (C::ROWS * C::UPR), rr = u - zz * (C::ROWS * C::UPR), yy = rr / C::UPR, sg = rr - yy * C::UPR;
          const int gz = tl.z0 + zz, gy = tl.y0 + yy, gx = tl.x0 + sg * 4;
          // (a channel past Ci lies beyond the chunk's resource: the bounds check returns zeros for it)
          o[cc * C::IPC + q] = (gz < D && gy < H && gx < W)
                                   ? ((unsigned)cl * DHW + (unsigned)gz * HW + (unsigned)gy * W + (unsigned)gx) * 4u : DMA_OOB;
        }
      }
    }
  };
  // A chunk's copies are numbered: pieces [0, NQ) = the input tile (vector path), [NQ, NQ + WV4) = the weights; stage()
  // issues pieces [lo, hi) so that the main loop can deal them out between its MFMA groups (a wave that queues a dozen
  // 1-KiB copies back to back sits in the address queue for a few thousand cycles with its matrix pipe idle).
  constexpr int NPIECE = NQ + WV4;
  auto stage = [&](const Tile& tl, const unsigned (&toff)[NQ], int c0, float* buf, int lo = 0, int hi = 1 << 20) {
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(x + ((size_t)tl.b * Ci + c0) * DHW, (unsigned)min(C::CK, Ci - c0) * DHW * 4u);
    if constexpr (C::V16) {
      // unit = 4 consecutive floats of a staged row; the units of one channel are linear in LDS
#pragma unroll
      for (int cc = 0; cc < CPW; ++cc) {
        const int cl = wave * CPW + cc;
#pragma unroll
        for (int q = 0; q < C::IPC; ++q)
          if (cc * C::IPC + q >= lo && cc * C::IPC + q < hi && q * 64 + lane < C::UPC)
            dma16(xrs, toff[cc * C::IPC + q], 0u, buf + cl * C::CH_STRIDE + q * 256);
      }
    } else if (lo == 0) {
    const int gx = tl.x0 + lane;
    const unsigned xvoff = (lane < C::P && gx < W) ? (unsigned)gx * 4u : DMA_OOB;
    if (lane < C::P) {
#pragma unroll
      for (int cc = 0; cc < CPW; ++cc) {
        const int cl = wave * CPW + cc;
        float* dc = buf + cl * C::CH_STRIDE;
#pragma unroll
        for (int zz = 0; zz < C::ZS; ++zz) {
          const bool zok = tl.z0 + zz < D && c0 + cl < Ci;
          const unsigned zoff = ((unsigned)cl * DHW + (unsigned)(tl.z0 + zz) * HW) * 4u;
#pragma unroll
          for (int yy = 0; yy < C::ROWS; ++yy) {
            const bool ok = zok && tl.y0 + yy < H;
            dma4(xrs, ok ? xvoff : DMA_OOB, ok ? zoff + (unsigned)(tl.y0 + yy) * W * 4u : 0u,
                 dc + zz * C::PLANE + yy * C::P);
          }
        }
      }
    }
    }
#pragma unroll
    for (int i = 0; i < WV4; ++i)
      if (NQ + i >= lo && NQ + i < hi && i * 256 + (int)threadIdx.x < WCH4)
        dma16(wrs, woff[i], (unsigned)(c0 / 2) * (27 * C::NTT * 64 * 4), buf + C::IN_FLOATS + (i * 256 + wave * 64) * 4);
  };

  // Per-channel affine, staged ONCE into LDS behind the chunk buffers.  (Global loads of scale / shift inside the tile
  // loop stay "pending" in the compiler's wait-count model on the paths that do not consume them; the first reuse of
  // their registers in the next tile then drains vmcnt(0) and stalls on the chunk copy that was just put in flight.
  // LDS reads count on lgkmcnt instead and cost no registers across the loop.)
  float* aff = lds + C::LDS_FLOATS;
  if (threadIdx.x < C::COUT) {
    const bool real = (int)threadIdx.x < cvalid;   // rows >= cvalid are padding (the 1-channel head uses a 32-row tile)
    aff[threadIdx.x] = (scale && real) ? scale[threadIdx.x] : 1.f;
    aff[C::COUT + threadIdx.x] = (shift && real) ? shift[threadIdx.x] : 0.f;
  }

  const int NC = cdiv(Ci, C::CK);  // a partial last chunk reads zeros (bounds check) against zero-padded weights
  constexpr int NU = (C::CK / 2) * NAZ;  // (channel pair, az) units per chunk
  const int Ho = 2 * H, Wo = 2 * W;
  const unsigned HWo = (unsigned)Ho * Wo, DHWo = 2u * D * HWo;
  Tile cur_t = tile_at(0);
  unsigned coff[NQ], noff[NQ];
  tile_offsets(cur_t, coff);
  stage(cur_t, coff, 0, lds);
  __syncthreads();
  int g = 0;  // chunks consumed by this workgroup so far: selects the LDS buffer
  for (int it = 0; it < my_tiles; ++it) {
    const bool has_next = it + 1 < my_tiles;
    Tile next_t = cur_t;
    if (has_next) {
      next_t = tile_at(it + 1);
      tile_offsets(next_t, noff);
    }

    f32x16 acc[2][2][C::MT][C::NT];  // [py][px]
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[py][px][mt][nt][r] = 0.f;

    for (int ci = 0; ci < NC; ++ci, ++g) {
      const float* cur = lds + (g & 1) * C::BUF_FLOATS;
      float* nxt = lds + ((g + 1) & 1) * C::BUF_FLOATS;
      // the next chunk's copies (of this tile, or the first chunk of the next one) are dealt out over the first SU units
      constexpr int SU = C::V16 ? (NU >= 6 ? NU - 2 : (NU > 1 ? NU - 1 : 1)) : 1, PPU = (NPIECE + SU - 1) / SU;
      const int smode = (dbg & 2) ? 0 : (ci + 1 < NC ? 1 : (has_next ? 2 : 0));
      auto deal = [&](int u) {
        if (smode == 1)
          stage(cur_t, coff, (ci + 1) * C::CK, nxt, u * PPU, (u + 1) * PPU);
        else if (smode == 2)
          stage(next_t, noff, 0, nxt, u * PPU, (u + 1) * PPU);
      };
      if (!C::V16) deal(0);
      const float* abase = cur + C::IN_FLOATS + (wn * C::NT) * 64 + lane;
      const float* bbase = cur + h * C::CH_STRIDE + wz * C::PLANE + j;
      float af[2][9][C::NT], bf[2][2][2][C::MT];  // bf[buf][ay][ox][mt]
      auto load_frag = [&](int u, float (&a)[9][C::NT], float (&bq)[2][2][C::MT]) {
        const int cp = u / NAZ, az = u % NAZ;
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt) a[k][nt] = abase[(u * 9 + k) * C::NTT * 64 + nt * 64];
        const float* bp = bbase + 2 * cp * C::CH_STRIDE + az * C::PLANE;
#pragma unroll
        for (int ay = 0; ay < 2; ++ay)
#pragma unroll
          for (int ox = 0; ox < 2; ++ox)
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt) bq[ay][ox][mt] = bp[ay * C::P + ox + mt * 32];
      };
      load_frag(0, af[0], bf[0]);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (u + 1 < NU) load_frag(u + 1, af[(u + 1) & 1], bf[(u + 1) & 1]);
        if (C::V16 && u < SU) deal(u);
        __builtin_amdgcn_sched_barrier(0);
        const auto& a = af[u & 1];
        const auto& bq = bf[u & 1];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt) {
            // even y (py = 0): ky = 1 from input row ay = 0
            acc[0][0][mt][nt] = DMB_MFMA(a[3 + 1][nt], bq[0][0][mt], acc[0][0][mt][nt]);  // even x: kx = 1, ox = 0
            acc[0][1][mt][nt] = DMB_MFMA(a[3 + 2][nt], bq[0][0][mt], acc[0][1][mt][nt]);  // odd x:  kx = 2, ox = 0
            acc[0][1][mt][nt] = DMB_MFMA(a[3 + 0][nt], bq[0][1][mt], acc[0][1][mt][nt]);  //         kx = 0, ox = 1
            // odd y (py = 1): ky = 2 from ay = 0, ky = 0 from ay = 1
            acc[1][0][mt][nt] = DMB_MFMA(a[6 + 1][nt], bq[0][0][mt], acc[1][0][mt][nt]);
            acc[1][1][mt][nt] = DMB_MFMA(a[6 + 2][nt], bq[0][0][mt], acc[1][1][mt][nt]);
            acc[1][1][mt][nt] = DMB_MFMA(a[6 + 0][nt], bq[0][1][mt], acc[1][1][mt][nt]);
            acc[1][0][mt][nt] = DMB_MFMA(a[0 + 1][nt], bq[1][0][mt], acc[1][0][mt][nt]);
            acc[1][1][mt][nt] = DMB_MFMA(a[0 + 2][nt], bq[1][0][mt], acc[1][1][mt][nt]);
            acc[1][1][mt][nt] = DMB_MFMA(a[0 + 0][nt], bq[1][1][mt], acc[1][1][mt][nt]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }

    // ---- epilogue of cur_t ----
    const int gzi = cur_t.z0 + wz;
    if constexpr (C::VEPI) {
      if (!(dbg & 1)) {
      // ---- vector epilogue: per (channel tile, 32-position tile, y parity) the two x-parity accumulator tiles are
      // interleaved in LDS ([PCH channels][64 output columns], 8-byte writes), read back as 4 consecutive x of one
      // channel and stored / residual-loaded as 16-byte words through buffer resources (lanes outside the volume get an
      // out-of-range offset: branch-free).  Residual loads run one pass ahead of the stores.
      // (g has been advanced past the last chunk: buffer (g & 1) is being filled for the next tile, ((g - 1) & 1) is
      // the one just consumed; every wave's reads of it completed before the barrier that ended the chunk loop)
      float* scr = C::SCR_PRIVATE ? lds + ((g - 1) & 1) * C::BUF_FLOATS + wave * C::PRIV_FLOATS
                                  : lds + C::LDS_FLOATS + C::AFF_FLOATS + wave * (C::PCH * C::SCR_PITCH);
      const unsigned gz = 2 * gzi + PZ;
      float* yb = y + (size_t)cur_t.b * C::COUT * DHWo;
      const __amdgpu_buffer_rsrc_t yrs = make_rsrc(yb, (unsigned)C::COUT * DHWo * 4u);
      const __amdgpu_buffer_rsrc_t rrs = make_rsrc(res ? res + (size_t)cur_t.b * C::COUT * DHWo : yb, (unsigned)C::COUT * DHWo * 4u);
      const int rl = lane >> 4, x4 = (lane & 15) * 4;
      const float lo = relu == 1 ? 0.f : -__builtin_inff();    // ReLU after the residual add
      const float lo2 = relu == 2 ? 0.f : -__builtin_inff();   // ReLU before it (GC-Net)
      constexpr int QP = 32 / C::PCH, KP = C::PCH / 4;        // passes per 32-channel tile, 16-byte words per lane and pass
      constexpr int NPASS = C::NT * C::MT * 2 * QP;
      auto pass_off = [&](int t, unsigned (&off)[KP], int (&ch)[KP]) {
        const int q = t % QP, py = (t / QP) % 2, mt = (t / (2 * QP)) % C::MT, nt = t / (2 * QP * C::MT);
        const int ly = (mt * 32) / C::P, lx0 = (mt * 32) % C::P;
        const int gyi = cur_t.y0 + ly, gxi = cur_t.x0 + lx0 + x4 / 2;
        const bool ok = gzi < D && gyi < H && lx0 + x4 / 2 < C::TX && gxi < W && !(dbg & 4);
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          ch[k] = (wn * C::NT + nt) * 32 + q * C::PCH + 4 * k + rl;
          off[k] = ok ? ((unsigned)ch[k] * DHWo + gz * HWo + (unsigned)(2 * gyi + py) * Wo + 2u * (unsigned)(cur_t.x0 + lx0) + (unsigned)x4) * 4u
                      : DMA_OOB;
        }
      };
      auto run = [&](auto has_res) {
        constexpr bool HAS_RES = decltype(has_res)::value;
        // A wave has no other work to hide a load behind (the accumulators take half of its registers), so the residual
        // loads run in a ring RD passes deep (16 words of 16 bytes in flight per lane): one exposed latency per item
        // instead of one per pass.
        constexpr int RD = HAS_RES ? (16 / KP < NPASS ? 16 / KP : NPASS) : 1;
        u32x4 rv[RD][KP];   // (offsets are recomputed at use: only the loaded words stay live across passes)
        if constexpr (HAS_RES) {
#pragma unroll
          for (int t = 0; t < RD; ++t) {
            unsigned o0[KP];
            int c0[KP];
            pass_off(t, o0, c0);
#pragma unroll
            for (int k = 0; k < KP; ++k) rv[t][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)o0[k], 0, DMB_EPI_LD);
          }
        }
#pragma unroll
        for (int t = 0; t < NPASS; ++t) {
          const int q = t % QP, py = (t / QP) % 2, mt = (t / (2 * QP)) % C::MT, nt = t / (2 * QP * C::MT);
          const int sl = t % RD;
          unsigned off[KP];
          int ch[KP];
          pass_off(t, off, ch);
#pragma unroll
          for (int rr = 0; rr < C::PCH / 2; ++rr) {
            const int r = q * (C::PCH / 2) + rr;
            const int row = (r & 3) + 4 * h + (C::PCH == 16 ? 8 * ((r >> 2) & 1) : 0);
            *reinterpret_cast<float2*>(scr + row * C::SCR_PITCH + 2 * j) =
                make_float2(acc[py][0][mt][nt][r], acc[py][1][mt][nt][r]);
          }
#pragma unroll
          for (int k = 0; k < KP; ++k) {
            float4 v = *reinterpret_cast<const float4*>(scr + (4 * k + rl) * C::SCR_PITCH + x4);
            const float sc = aff[ch[k]], sh = aff[C::COUT + ch[k]];
            v.x = fmaxf(fmaf(v.x, sc, sh), lo2);
            v.y = fmaxf(fmaf(v.y, sc, sh), lo2);
            v.z = fmaxf(fmaf(v.z, sc, sh), lo2);
            v.w = fmaxf(fmaf(v.w, sc, sh), lo2);
            if constexpr (HAS_RES) {
              v.x += __uint_as_float(rv[sl][k].x);
              v.y += __uint_as_float(rv[sl][k].y);
              v.z += __uint_as_float(rv[sl][k].z);
              v.w += __uint_as_float(rv[sl][k].w);
            }
            u32x4 o;
            o.x = __float_as_uint(fmaxf(v.x, lo));
            o.y = __float_as_uint(fmaxf(v.y, lo));
            o.z = __float_as_uint(fmaxf(v.z, lo));
            o.w = __float_as_uint(fmaxf(v.w, lo));
            __builtin_amdgcn_raw_buffer_store_b128(o, yrs, (int)off[k], 0, DMB_EPI_ST);
          }
          if constexpr (HAS_RES) {
            if (t + RD < NPASS) {   // refill the slot just consumed
              pass_off(t + RD, off, ch);
#pragma unroll
              for (int k = 0; k < KP; ++k) rv[sl][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)off[k], 0, DMB_EPI_LD);
            }
          }
        }
      };
      if (res)
        run(std::true_type{});
      else
        run(std::false_type{});
      }
    } else if (gzi < D && !(dbg & 1)) {
      const unsigned gz = 2 * gzi + PZ;
      const int cout = cvalid < C::COUT ? cvalid : C::COUT;   // channels the output tensor really has
      float* yb = y + (size_t)cur_t.b * cout * DHWo;
      const float* rb = res ? res + (size_t)cur_t.b * cout * DHWo : nullptr;
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) {
        const int co0 = (wn * C::NT + nt) * 32;
        float sc[16], sh[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sc[r] = aff[co0 + cd_row(r, h)];
          sh[r] = aff[C::COUT + co0 + cd_row(r, h)];
        }
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
          const int m = mt * 32 + j;
          const int ly = m / C::P, lx = m - ly * C::P;
          const int gyi = cur_t.y0 + ly, gxi = cur_t.x0 + lx;
          if (m < C::TY * C::P && lx < C::TX && gyi < H && gxi < W) {
#pragma unroll
            for (int py = 0; py < 2; ++py) {
              const f32x16 a2[2] = {acc[py][0][mt][nt], acc[py][1][mt][nt]};
              store_tile<2>(a2, sc, sh, rb, yb, co0, h, DHWo, gz * HWo + (unsigned)(2 * gyi + py) * Wo + 2u * gxi, relu, cvalid);
            }
          }
        }
      }
    }
    cur_t = next_t;
#pragma unroll
    for (int q = 0; q < NQ; ++q) coff[q] = noff[q];
  }
}

// Workgroups [0, g0) take the even-z work items, the rest the odd-z ones (which carry twice the MFMAs).
template <class C>
__global__ __launch_bounds__(256, C::WPE) void deconv3d_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ res, float* __restrict__ y,
                                                               int Ci, int D, int H, int W, int ntx, int nty, int ntz,
                                                               int ntiles, int g0, int relu, int cvalid) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x < g0)
    deconv_body<C, 0>(lds, x, wp, scale, shift, res, y, Ci, D, H, W, ntx, nty, ntz, blockIdx.x, g0, ntiles, relu, cvalid);
  else
    deconv_body<C, 1>(lds, x, wp, scale, shift, res, y, Ci, D, H, W, ntx, nty, ntz, blockIdx.x - g0, gridDim.x - g0,
                      ntiles, relu, cvalid);
}

// ---------------------------------------------------------------------------------------------------------
// Single-output-channel 3x3x3 convolution (classifier heads).  N = 1 wastes 31/32 of an MFMA tile, and the
// layer is HBM-bound anyway (reads Ci planes, writes one), so this one is a VALU kernel: weights are wave
// uniform (scalar loads -> SGPR operands of v_fmac_f32), each thread produces 4 consecutive x from 6-float
// LDS rows.  acc order: ci ascending, then (kd, kh, kw) ascending -- the same FP32 fma chain as above.
// ---------------------------------------------------------------------------------------------------------
constexpr int C1_TX = 60, C1_TY = 4, C1_TZ = 8, C1_CK = 2;
constexpr int C1_P = C1_TX + 2;                     // 62: one LDS-DMA row per wave instruction, even pitch (8-byte reads)
constexpr int C1_ROWS = C1_TY + 2, C1_ZS = C1_TZ + 2;
constexpr int C1_PLANE = C1_ROWS * C1_P + 2;        // 374 = 2 (mod 4): z-neighbour lanes hit disjoint banks
constexpr int C1_CH = C1_ZS * C1_PLANE + 2;
constexpr int C1_BUF = C1_CK * C1_CH;

__global__ __launch_bounds__(256, 2) void conv3d_c1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float bias, const float* __restrict__ res,
                                                           float* __restrict__ y, int Ci, int D, int H, int W, int ntx,
                                                           int nty, int ntz) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 * C1_BUF floats
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % ntx;
  t /= ntx;
  const int ty = t % nty;
  t /= nty;
  const int tz = t % ntz;
  const int b = t / ntz;
  const int x0 = tx * C1_TX, y0 = ty * C1_TY, z0 = tz * C1_TZ;
  const unsigned HW = (unsigned)H * W, DHW = (unsigned)D * HW;
  const float* xb = x + (size_t)b * Ci * DHW;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // compute mapping: a thread owns 4 consecutive x of TWO adjacent rows of one z-slice (8 independent accumulators,
  // every LDS row it loads feeds up to 2 x 3 x 4 fmas).  16 lanes per row pair (15 active); lanes 16-31 of a 32-lane
  // LDS access group take the next z-slice, whose plane offset is 2 mod 4 floats -> disjoint banks, no conflicts.
  const int lxq = threadIdx.x & 15;
  const int lz = ((threadIdx.x >> 6) << 1) | ((threadIdx.x >> 4) & 1);   // 0..7
  const int lyp = (threadIdx.x >> 5) & 1;                                // row pair: rows 2*lyp, 2*lyp + 1
  const bool worker = lxq < 15;
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

  const int gx = x0 - 1 + lane;
  const unsigned xvoff = (lane < C1_P && gx >= 0 && gx < W) ? (unsigned)gx * 4u : DMA_OOB;
  constexpr int PPW = C1_CK * C1_ZS / 4;  // 5 planes per wave per chunk
  static_assert((C1_CK * C1_ZS) % 4 == 0, "planes are dealt evenly to the 4 waves");
  auto stage = [&](int c0, float* buf) {
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(xb + (size_t)c0 * DHW, (unsigned)min(C1_CK, Ci - c0) * DHW * 4u);
    if (lane < C1_P) {
#pragma unroll
      for (int q = 0; q < PPW; ++q) {
        const int pl = wave * PPW + q, cl = pl / C1_ZS, zz = pl - cl * C1_ZS;
        const int gz = z0 - 1 + zz;
        const bool zok = gz >= 0 && gz < D && c0 + cl < Ci;
        const unsigned zoff = ((unsigned)cl * DHW + (unsigned)max(gz, 0) * HW) * 4u;
        float* dpl = buf + cl * C1_CH + zz * C1_PLANE;
#pragma unroll
        for (int yy = 0; yy < C1_ROWS; ++yy) {
          const int gy = y0 - 1 + yy;
          const bool ok = zok && gy >= 0 && gy < H;
          dma4(xrs, ok ? xvoff : DMA_OOB, ok ? zoff + (unsigned)gy * W * 4u : 0u, dpl + yy * C1_P);
        }
      }
    }
  };

  const int NC = (Ci + C1_CK - 1) / C1_CK;
  stage(0, lds);
  __syncthreads();
  for (int ci = 0; ci < NC; ++ci) {
    const float* cur = lds + (ci & 1) * C1_BUF;
    if (ci + 1 < NC) stage((ci + 1) * C1_CK, lds + ((ci + 1) & 1) * C1_BUF);
    if (worker) {
#pragma unroll
      for (int cl = 0; cl < C1_CK; ++cl) {
        const int c = ci * C1_CK + cl;
        const float* wc = w + (size_t)min(c, Ci - 1) * 27;  // channels past Ci were staged as zeros
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
          float v[4][6];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float* row = cur + cl * C1_CH + (lz + dz) * C1_PLANE + (2 * lyp + r) * C1_P + lxq * 4;
            const float2 q0 = *reinterpret_cast<const float2*>(row);
            const float2 q1 = *reinterpret_cast<const float2*>(row + 2);
            const float2 q2 = *reinterpret_cast<const float2*>(row + 4);
            v[r][0] = q0.x; v[r][1] = q0.y; v[r][2] = q1.x; v[r][3] = q1.y; v[r][4] = q2.x; v[r][5] = q2.y;
          }
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const float wv = wc[dz * 9 + dy * 3 + dx];
#pragma unroll
              for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[oy][o] = fmaf(v[oy + dy][o + dx], wv, acc[oy][o]);
            }
        }
      }
    }
    __syncthreads();
  }
  const int gz = z0 + lz, gxo = x0 + lxq * 4;
  if (worker && gz < D && gxo < W) {
#pragma unroll
    for (int oy = 0; oy < 2; ++oy) {
      const int gy = y0 + 2 * lyp + oy;
      if (gy >= H) continue;
      const size_t o = (size_t)b * DHW + (size_t)gz * HW + (size_t)gy * W + gxo;
      if ((W & 3) == 0) {  // gxo % 4 == 0 and W % 4 == 0: the four outputs are one aligned 16-byte word
        float4 v = make_float4(acc[oy][0] + bias, acc[oy][1] + bias, acc[oy][2] + bias, acc[oy][3] + bias);
        if (res) {
          const float4 r = *reinterpret_cast<const float4*>(res + o);
          v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        *reinterpret_cast<float4*>(y + o) = v;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (gxo + i < W) {
            float v = acc[oy][i] + bias;
            if (res) v += res[o + i];
            y[o + i] = v;
          }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same layer for 16-byte aligned rows (W % 4 == 0): what bounded the kernel above was not HBM but (a) the dword
// LDS-DMA rate (one lane per clock whatever the word size: 1.6 GB of haloed rows as dwords) and (b) LDS reads (every staged
// value was read six times as 8-byte words).  Here
//   * a tile is 8 z x 8 y x 60 x, ONE input channel per step; its haloed rows are fetched as 16-byte words from the aligned
//     column x0 - 4 with ordinary buffer loads (zero padding = the bounds check) one step AHEAD into registers and written
//     to LDS with ds_write_b128 after the step's arithmetic: a quarter of the vector-memory lanes, single LDS buffer;
//   * a thread owns 4 x  x  2 y  x  2 z outputs (16 accumulators); per input row it reads ONE aligned 16-byte word --
//     the four columns under its outputs -- and takes the two halo columns from its x neighbours' registers with DPP lane
//     shifts inside the 16-lane row (lane 0 of a row reads the left halo word, lane 15 fetches the single right halo
//     column): LDS bytes per row = the row;
//   * accumulation order per output: channel ascending, then (dz, dy, dx) ascending -- bit-identical to the kernel above.
// ---------------------------------------------------------------------------------------------------------
constexpr int C1V_TX = 60, C1V_TY = 8, C1V_TZ = 8;
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int C1V_RW = 17;                                 // staged row: columns x0 - 4 .. x0 + 63 = 17 words of 16 bytes
// LDS row pitch 96 floats: ds_read_b128 is served in four groups of 16 NON-contiguous lanes ({0-3, 12-15, 20-27}, ...,
// MI355X_MICROARCH.md, LDS), so the two 16-lane halves of a 32-lane block -- two output row pairs, 2 rows apart -- must start
// on the same 256-byte bank row for a group to cover each bank once: 2 * pitch = 0 (mod 64 floats).  (At pitch 68 the
// counters showed 57 % of the LDS cycles lost to conflicts.)
constexpr int C1V_P = 96;
constexpr int C1V_ROWS = C1V_TY + 2, C1V_ZS = C1V_TZ + 2;
constexpr int C1V_UNITS = C1V_ZS * C1V_ROWS * C1V_RW;      // 1700 words per channel
constexpr int C1V_UPT = (C1V_UNITS + 255) / 256;            // 7 per thread

__device__ __forceinline__ float dpp_row_shr1(float v) {   // lane i <- lane i - 1 within a row of 16 lanes (lane 0 keeps v)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_row_shl1(float v) {   // lane i <- lane i + 1 within a row of 16 lanes (lane 15 keeps v)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));
}

#ifndef DMB_C1V_WPE
#define DMB_C1V_WPE 2   // workgroups per CU the register allocation is held to (build-time experiment knob)
#endif
__global__ __launch_bounds__(256, DMB_C1V_WPE) void conv3d_c1v_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            float bias, const float* __restrict__ res,
                                                            float* __restrict__ y, int Ci, int D, int H, int W, int ntx,
                                                            int nty, int ntz) {
  __shared__ __attribute__((aligned(16))) float tile[C1V_ZS * C1V_ROWS * C1V_P];
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % ntx;
  t /= ntx;
  const int ty = t % nty;
  t /= nty;
  const int tz = t % ntz;
  const int b = t / ntz;
  const int x0 = tx * C1V_TX, y0 = ty * C1V_TY, z0 = tz * C1V_TZ;
  const unsigned HW = (unsigned)H * W, DHW = (unsigned)D * HW;
  const float* xb = x + (size_t)b * Ci * DHW;
  const int tid = threadIdx.x;
  const int lq = tid & 15;            // word of the row: columns 4 lq .. 4 lq + 3 of the staged row = x0 - 4 + 4 lq ...
  const int lyp = (tid >> 4) & 3;     // output rows y0 + 2 lyp, + 1
  const int lzp = tid >> 6;           // output planes z0 + 2 lzp, + 1 (one wave per plane pair)

  // staging offsets of this thread's words (tile constants; the channel is selected by the resource base)
  unsigned soff[C1V_UPT];
#pragma unroll
  for (int q = 0; q < C1V_UPT; ++q) {
    const int u = q * 256 + tid;
    const int zz = u / (C1V_ROWS * 17), rr = u - zz * (C1V_ROWS * 17), yy = rr / 17, sg = rr - yy * 17;
    const int gz = z0 - 1 + zz, gy = y0 - 1 + yy, gx = x0 - 4 + sg * 4;
    const bool ok = u < C1V_UNITS && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
    soff[q] = ok ? ((unsigned)gz * HW + (unsigned)gy * W + (unsigned)gx) * 4u : DMA_OOB;
  }
  u32x4 stg[C1V_UPT];
  auto fetch = [&](int c) {
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(xb + (size_t)c * DHW, DHW * 4u);
#pragma unroll
    for (int q = 0; q < C1V_UPT; ++q) stg[q] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)soff[q], 0, 0);
  };
  int doff[C1V_UPT];   // LDS float offset of each word
#pragma unroll
  for (int q = 0; q < C1V_UPT; ++q) {
    const int u = q * 256 + tid, rw = u / C1V_RW;
    doff[q] = rw * C1V_P + (u - rw * C1V_RW) * 4;
  }
  auto commit = [&]() {
#pragma unroll
    for (int q = 0; q < C1V_UPT; ++q)
      if (q * 256 + tid < C1V_UNITS) *reinterpret_cast<u32x4*>(tile + doff[q]) = stg[q];
  };

  // accumulators as x pairs (outputs 0,1 and 2,3 of a row): the arithmetic below is written on 2-vectors so that it compiles
  // to v_pk_fma_f32 -- two independent FP32 fmas per instruction, same rounding -- which is what makes the FP32 vector rate
  // reachable at all (a plain v_fma_f32 stream tops out at half of it, and this kernel was bound by exactly that).
  f32x2 acc[2][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int o = 0; o < 2; ++o) acc[a][c2][o] = f32x2{0.f, 0.f};

  fetch(0);
  for (int c = 0; c < Ci; ++c) {
    __syncthreads();          // every wave is done reading the previous channel's tile
    commit();
    __syncthreads();
    if (c + 1 < Ci) fetch(c + 1);   // lands while this channel is multiplied
    const float* wc = w + (size_t)c * 27;
#pragma unroll
    for (int p = 0; p < 4; ++p) {   // input plane z0 - 1 + 2 lzp + p
      // per input row the six columns under / next to this thread's outputs, as even pairs (0,1) (2,3) (4,5) and odd pairs
      // (1,2) (3,4): tap dx of output pair o reads ev[o] (dx 0), od[o] (dx 1), ev[o + 1] (dx 2)
      f32x2 ev[4][3], od[4][2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // input row y0 - 1 + 2 lyp + r
        const float* row = tile + ((2 * lzp + p) * C1V_ROWS + 2 * lyp + r) * C1V_P;
        const float4 m = *reinterpret_cast<const float4*>(row + lq * 4);
        const float v0 = dpp_row_shr1(m.w);                 // left neighbour's last column
        float v5 = dpp_row_shl1(m.x);                       // right neighbour's first column
        if (lq == 15) v5 = row[64];                         // the last worker fetches the right halo column itself
        ev[r][0] = f32x2{v0, m.x};
        ev[r][1] = f32x2{m.y, m.z};
        ev[r][2] = f32x2{m.w, v5};
        od[r][0] = f32x2{m.x, m.y};
        od[r][1] = f32x2{m.z, m.w};
      }
#pragma unroll
      for (int oz = 0; oz < 2; ++oz) {
        const int dz = p - oz;
        if (dz < 0 || dz > 2) continue;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float wv = wc[dz * 9 + dy * 3 + dx];
            const f32x2 w2 = f32x2{wv, wv};
#pragma unroll
            for (int oy = 0; oy < 2; ++oy)
#pragma unroll
              for (int o = 0; o < 2; ++o) {
                const f32x2 in = dx == 0 ? ev[oy + dy][o] : (dx == 1 ? od[oy + dy][o] : ev[oy + dy][o + 1]);
                acc[oz][oy][o] = __builtin_elementwise_fma(in, w2, acc[oz][oy][o]);
              }
          }
      }
    }
  }
  // lane lq's outputs are the columns of its word: x0 - 4 + 4 lq .. + 3 -- lane 0 holds the left halo word, lane 16's would
  // be the right one: workers are lanes 1 .. 15
  const int gx = x0 - 4 + lq * 4;
  if (lq == 0 || gx >= W) return;
#pragma unroll
  for (int oz = 0; oz < 2; ++oz) {
    const int gz = z0 + 2 * lzp + oz;
    if (gz >= D) continue;
#pragma unroll
    for (int oy = 0; oy < 2; ++oy) {
      const int gy = y0 + 2 * lyp + oy;
      if (gy >= H) continue;
      const size_t o = (size_t)b * DHW + (size_t)gz * HW + (size_t)gy * W + gx;
      float4 r4 = make_float4(acc[oz][oy][0].x + bias, acc[oz][oy][0].y + bias, acc[oz][oy][1].x + bias, acc[oz][oy][1].y + bias);
      if (res) {
        const float4 r = *reinterpret_cast<const float4*>(res + o);
        r4.x += r.x; r4.y += r.y; r4.z += r.z; r4.w += r.w;
      }
      *reinterpret_cast<float4*>(y + o) = r4;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same layer for launches that do not fill the chip (round 6: one small pair -- [1, 32, 16, 64, 128] is 48 tiles of
// conv3d_c1v_kernel, each walking its 32 channels one after the other behind two barriers: 32 x (a memory round trip) = 60 us for
// 17 MB).  Here a tile is 2 z x 8 y x 60 x and the FOUR WAVES of a workgroup split the input channels: wave w stages and multiplies
// channels [w Ci / 4, (w + 1) Ci / 4) in its own LDS tile (no workgroup barrier in the channel loop, the fetches of the next two
// channels in flight), then the four partial sums of an output are added in ascending wave order (fixed order: reproducible run to
// run; not the single ascending chain of the kernels above, so the last bits differ from them).  Thread mapping, row reads and the
// packed-fma arithmetic are conv3d_c1v_kernel's.
// ---------------------------------------------------------------------------------------------------------
constexpr int C1S_TZ = 2, C1S_ZS = C1S_TZ + 2;
constexpr int C1S_UNITS = C1S_ZS * C1V_ROWS * C1V_RW;      // 680 words per channel
constexpr int C1S_UPT = (C1S_UNITS + 63) / 64;              // 11 per lane
constexpr int C1S_TILE = C1S_ZS * C1V_ROWS * C1V_P;         // floats of one wave's tile

__global__ __launch_bounds__(256, 1) void conv3d_c1s_kernel(const float* __restrict__ x, const float* __restrict__ w, float bias,
                                                            const float* __restrict__ res, float* __restrict__ y, int Ci, int D,
                                                            int H, int W, int ntx, int nty, int ntz) {
  __shared__ __attribute__((aligned(16))) float tiles[4 * C1S_TILE];
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % ntx;
  t /= ntx;
  const int ty = t % nty;
  t /= nty;
  const int tz = t % ntz;
  const int b = t / ntz;
  const int x0 = tx * C1V_TX, y0 = ty * C1V_TY, z0 = tz * C1S_TZ;
  const unsigned HW = (unsigned)H * W, DHW = (unsigned)D * HW;
  const float* xb = x + (size_t)b * Ci * DHW;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int lq = lane & 15, lyp = lane >> 4;
  float* tile = tiles + wave * C1S_TILE;
  const int CW = (Ci + 3) / 4, cb = wave * CW, ce = min(Ci, cb + CW);

  unsigned soff[C1S_UPT];
#pragma unroll
  for (int q = 0; q < C1S_UPT; ++q) {
    const int u = q * 64 + lane;
    const int zz = u / (C1V_ROWS * C1V_RW), rr = u - zz * (C1V_ROWS * C1V_RW), yy = rr / C1V_RW, sg = rr - yy * C1V_RW;
    const int gz = z0 - 1 + zz, gy = y0 - 1 + yy, gx = x0 - 4 + sg * 4;
    const bool ok = u < C1S_UNITS && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
    soff[q] = ok ? ((unsigned)gz * HW + (unsigned)gy * W + (unsigned)gx) * 4u : DMA_OOB;
  }
  // LDS float offset of word q * 64 + lane = doff0 + what q adds: 64 = 3 x 17 + 13 words on -> 3 rows and 13 words, one more row when
  // the word index wraps past 17 (recomputed per commit: eleven registers fewer than a table)
  const int rw0 = lane / C1V_RW, sg0 = lane - rw0 * C1V_RW;
  auto fetch = [&](u32x4 (&stg)[C1S_UPT], int c) {
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(xb + (size_t)c * DHW, DHW * 4u);
#pragma unroll
    for (int q = 0; q < C1S_UPT; ++q) stg[q] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)soff[q], 0, 0);
  };
  auto commit = [&](const u32x4 (&stg)[C1S_UPT]) {
#pragma unroll
    for (int q = 0; q < C1S_UPT; ++q) {
      const int sg = sg0 + (q * 64) % C1V_RW, wrap = sg >= C1V_RW ? 1 : 0;
      const int rw = rw0 + (q * 64) / C1V_RW + wrap;
      if (q * 64 + lane < C1S_UNITS) *reinterpret_cast<u32x4*>(tile + rw * C1V_P + (sg - wrap * C1V_RW) * 4) = stg[q];
    }
  };
  f32x2 acc[2][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int o = 0; o < 2; ++o) acc[a][c2][o] = f32x2{0.f, 0.f};
  auto multiply = [&](int c) {
    const float* wc = w + (size_t)c * 27;
#pragma unroll
    for (int p = 0; p < 4; ++p) {   // input plane z0 - 1 + p
      f32x2 ev[4][3], od[4][2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // input row y0 - 1 + 2 lyp + r
        const float* row = tile + (p * C1V_ROWS + 2 * lyp + r) * C1V_P;
        const float4 m = *reinterpret_cast<const float4*>(row + lq * 4);
        const float v0 = dpp_row_shr1(m.w);
        float v5 = dpp_row_shl1(m.x);
        if (lq == 15) v5 = row[64];
        ev[r][0] = f32x2{v0, m.x};
        ev[r][1] = f32x2{m.y, m.z};
        ev[r][2] = f32x2{m.w, v5};
        od[r][0] = f32x2{m.x, m.y};
        od[r][1] = f32x2{m.z, m.w};
      }
#pragma unroll
      for (int oz = 0; oz < 2; ++oz) {
        const int dz = p - oz;
        if (dz < 0 || dz > 2) continue;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float wv = wc[dz * 9 + dy * 3 + dx];
            const f32x2 w2 = f32x2{wv, wv};
#pragma unroll
            for (int oy = 0; oy < 2; ++oy)
#pragma unroll
              for (int o = 0; o < 2; ++o) {
                const f32x2 in = dx == 0 ? ev[oy + dy][o] : (dx == 1 ? od[oy + dy][o] : ev[oy + dy][o + 1]);
                acc[oz][oy][o] = __builtin_elementwise_fma(in, w2, acc[oz][oy][o]);
              }
          }
      }
    }
  };
  // the wave's channels, two fetches in flight; the tile is this wave's own: ordering inside a wave is program order (the fences
  // only keep the compiler from moving LDS accesses across the lane-crossing hand-over)
  u32x4 sa[C1S_UPT], sb[C1S_UPT];
  if (cb < ce) fetch(sa, cb);
  if (cb + 1 < ce) fetch(sb, cb + 1);
  for (int c = cb; c < ce; c += 2) {
    commit(sa);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (c + 2 < ce) fetch(sa, c + 2);
    multiply(c);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (c + 1 < ce) {
      commit(sb);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (c + 3 < ce) fetch(sb, c + 3);
      multiply(c + 1);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
  // partial sums [wave][16 values][64 lanes] into the wave's own (now dead) tile, then thread (lane, q) adds the four partials of
  // output row q = (oz, oy) of its lane in ascending wave order
#pragma unroll
  for (int oz = 0; oz < 2; ++oz)
#pragma unroll
    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        tile[((oz * 2 + oy) * 4 + o * 2) * 64 + lane] = acc[oz][oy][o].x;
        tile[((oz * 2 + oy) * 4 + o * 2 + 1) * 64 + lane] = acc[oz][oy][o].y;
      }
  __syncthreads();
  const int q = wave, oz = q >> 1, oy = q & 1;
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float s = tiles[(q * 4 + i) * 64 + lane];
#pragma unroll
    for (int ww = 1; ww < 4; ++ww) s += tiles[ww * C1S_TILE + (q * 4 + i) * 64 + lane];
    v[i] = s + bias;
  }
  const int gx = x0 - 4 + lq * 4, gz = z0 + oz, gy = y0 + 2 * lyp + oy;
  if (lq == 0 || gx >= W || gz >= D || gy >= H) return;
  const size_t o = (size_t)b * DHW + (size_t)gz * HW + (size_t)gy * W + gx;
  float4 r4 = make_float4(v[0], v[1], v[2], v[3]);
  if (res) {
    const float4 r = *reinterpret_cast<const float4*>(res + o);
    r4.x += r.x; r4.y += r.y; r4.z += r.z; r4.w += r.w;
  }
  *reinterpret_cast<float4*>(y + o) = r4;
}

// The accumulation order above is (dz, dy) outer, dx inner PER channel, i.e. tap-ascending within a channel and
// channels ascending -- identical to the MFMA kernels' k order up to their channel pairing.

static long long cdiv_ll(long long a, long long b) { return (a + b - 1) / b; }
template <class C>
static int launch_s1(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                     float* y, int B, int Ci, int D, int H, int W, int relu, hipStream_t st) {
  const int ntx = C::LIN ? cdiv(H * W, 64) : cdiv(W, C::TX), nty = C::LIN ? 1 : cdiv(H, C::TY), ntz = cdiv(D, C::TZ);
  const long long nblk = (long long)B * ntx * nty * ntz;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv3d: grid too large");
  // (development option 17: extra KB of LDS per workgroup -- fewer workgroups per CU, for occupancy experiments)
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float) + (size_t)DMB_OPT(17) * 1024;
  if (DMB_OPT(17)) {   // the experiment raises the cap itself (DMB_ENSURE_LDS sets it once, to the kernel's own need)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_s1_kernel<C, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_s1_kernel<C, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (res) {
    DMB_ENSURE_LDS((&conv3d_s1_kernel<C, true>), (size_t)(lds));
    hipLaunchKernelGGL((conv3d_s1_kernel<C, true>), dim3((unsigned)nblk), dim3(256), lds, st, x, wp, scale, shift, res, y, Ci, D,
                       H, W, ntx, nty, ntz, relu, (double*)nullptr);
  } else {
    DMB_ENSURE_LDS((&conv3d_s1_kernel<C, false>), (size_t)(lds));
    hipLaunchKernelGGL((conv3d_s1_kernel<C, false>), dim3((unsigned)nblk), dim3(256), lds, st, x, wp, scale, shift, res, y, Ci, D,
                       H, W, ntx, nty, ntz, relu, (double*)nullptr);
  }
  return launch_status("conv3d stride-1 launch failed");
}

// The raw convolution (no affine, skip or ReLU) with the per-workgroup channel sums of its output (conv3d_s1_kernel, STATS).
template <class C>
static int launch_s1_stats(const float* x, const float* wp, float* y, double* stats, int B, int Ci, int D, int H, int W, hipStream_t st) {
  const int ntx = cdiv(W, C::TX), nty = cdiv(H, C::TY), ntz = cdiv(D, C::TZ);
  const long long nblk = (long long)B * ntx * nty * ntz;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv3d: grid too large");
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  DMB_ENSURE_LDS((&conv3d_s1_kernel<C, false, true>), (size_t)(lds));
  hipLaunchKernelGGL((conv3d_s1_kernel<C, false, true>), dim3((unsigned)nblk), dim3(256), lds, st, x, wp, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, y, Ci, D, H, W, ntx, nty, ntz, 0, stats);
  return launch_status("conv3d stride-1 (+ statistics) launch failed");
}

template <class C>
static int launch_s2(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                     float* y, int B, int Ci, int D, int H, int W, int relu, hipStream_t st) {
  const int Do = (D - 1) / 2 + 1, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int ntx = cdiv(Wo, C::TX), nty = cdiv(Ho, C::TY), ntz = cdiv(Do, C::TZ);
  const long long nblk = (long long)B * ntx * nty * ntz;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv3d: grid too large");
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  DMB_ENSURE_LDS((&conv3d_s2_kernel<C>), (size_t)(lds));
  hipLaunchKernelGGL((conv3d_s2_kernel<C>), dim3((unsigned)nblk), dim3(C::NTHREADS), lds, st, x, wp, scale, shift, res, y, Ci, D,
                     H, W, Do, Ho, Wo, ntx, nty, ntz, relu);
  return launch_status("conv3d stride-2 launch failed");
}

template <class C>
static int launch_deconv(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                         float* y, int B, int Ci, int Co, int D, int H, int W, int relu, hipStream_t st) {
  const int ntx = cdiv(W, C::TX), nty = cdiv(H, C::TY), ntz = cdiv(D, C::TZ);
  const long long ntiles = (long long)B * ntx * nty * ntz;
  if (ntiles > 0x3fffffffLL) return fail(DMB_EUNSUPPORTED, "deconv3d: grid too large");
  const size_t lds = (size_t)(C::LDS_FLOATS + C::AFF_FLOATS + C::SCR_FLOATS) * sizeof(float);
  DMB_ENSURE_LDS((&deconv3d_kernel<C>), (size_t)(lds));
  const int ncu = num_cus();
  // persistent grid: C::WPE workgroups per CU; a third of them walk the even-z items, two thirds the odd-z items
  // (twice the work each).  With fewer items than slots every workgroup gets exactly one item.
  const long long slots = (long long)C::WPE * ncu;
  long long g0, g1;
  if (2 * ntiles <= slots) {
    g0 = g1 = ntiles;
  } else {
    // twice as many workgroups as resident slots: each walks half as many items, and the dispatcher's dynamic placement
    // evens out what a static walk cannot (measured 64 -> 32 at half resolution: 0.98 -> 0.82 ms)
    const long long total = 2 * slots;
    g0 = total / 3 > 0 ? total / 3 : 1;
    g1 = total - g0;
    if (g0 > ntiles) g0 = ntiles;
    if (g1 > ntiles) g1 = ntiles;
  }
  hipLaunchKernelGGL((deconv3d_kernel<C>), dim3((unsigned)(g0 + g1)), dim3(256), lds, st, x, wp, scale, shift, res, y, Ci,
                     D, H, W, ntx, nty, ntz, (int)ntiles, (int)g0, relu, Co);
  return launch_status("deconv3d launch failed");
}

}  // namespace dmb

using namespace dmb;

static int ci_padded(int Ci) { return (Ci + 7) / 8 * 8; }  // covers every kernel's channel-chunk size
static int co_padded(int Co) { return cdiv(Co, 32) * 32; }
extern "C" long long dmb_conv3d_packed_floats(int Co, int Ci) { return (long long)co_padded(Co) * ci_padded(Ci) * 27; }
extern "C" long long dmb_deconv3d_packed_floats(int Ci, int Co) { return (long long)co_padded(Co) * ci_padded(Ci) * 27; }

static int pack_common(const float* w, float* wp, int Co, int Ci, int transposed, void* stream) {
  if (!w || !wp || Co <= 0 || Ci <= 0) return fail(DMB_EINVAL, "pack_weights: bad argument");
  const long long total = (long long)co_padded(Co) * ci_padded(Ci) * 27;
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp, Co, Ci, ci_padded(Ci),
                     transposed);
  return launch_status("pack_weights launch failed");
}

extern "C" int dmb_conv3d_pack_weights_f32(const float* w, float* wpack, int Co, int Ci, void* stream) {
  return pack_common(w, wpack, Co, Ci, 0, stream);
}
extern "C" int dmb_deconv3d_pack_weights_f32(const float* w, float* wpack, int Ci, int Co, void* stream) {
  return pack_common(w, wpack, Co, Ci, 1, stream);
}
extern "C" int dmb_conv3d_pack_dgrad_weights_f32(const float* w, float* wpack, int Co, int Ci, void* stream) {
  return pack_common(w, wpack, Ci, Co, 2, stream);   // a convolution Co -> Ci
}
extern "C" int dmb_conv3d_pack_weights_multi_f32(const void* jobs_device, int njobs, void* stream) {
  static_assert(sizeof(PackJob) == sizeof(dmb_pack_job), "dmb_pack_job layout");
  if (!jobs_device || njobs <= 0 || njobs > 65535) return fail(DMB_EINVAL, "pack_weights_multi: bad argument");
  hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(64, njobs), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const PackJob*>(jobs_device));
  return launch_status("pack_weights_multi launch failed");
}

// ---- Tile choice for the stride-1 kernel ---------------------------------------------------------------------------
// Every row-group tile shape works for ANY volume with 16-byte rows (W % 4 == 0, aligned bases): partial tiles are zero-filled
// by the staging's bounds checks and masked in the epilogue.  Which one runs is a cost estimate over the instantiated shapes --
// rounds of workgroups the launch needs on this chip x resident workgroups per CU x the matrix work of one workgroup -- not a
// table of image widths: e.g. 240 columns -> 48-column row pairs (5 tiles, nothing discarded), 312 columns (KITTI, 1248 / 4) ->
// 24-column row quads of 8 rows (13 tiles, nothing discarded), 156 -> 40-column row quads (4 tiles, 2.5 % discarded).
// Without 16-byte rows: the flattened mapping (dword staging) with the TX (60 or 52) that wastes fewer columns.
constexpr double S1_EFF_32x2 = 1.0;    // 32 x 2 row pairs (two column tiles per wave): measured rate relative to the estimate, profiles/r05_kbench_small_32.log
struct S1Tile {
  int tx, ty, tz, mt, wpe;   // tile extent, 32-voxel column tiles per wave, workgroups per CU
  bool lin;                  // 64-voxel runs of the (y, x) plane instead of boxes
  double eff;                // measured rate of the shape on volumes it tiles exactly, relative to the 48 x 4 row pairs
};
static double s1_cost(const S1Tile& t, int B, int D, int H, int W) {
  const long long n = t.lin ? (long long)B * cdiv(D, t.tz) * cdiv(H * W, 64)
                            : (long long)B * cdiv(D, t.tz) * cdiv(H, t.ty) * cdiv(W, t.tx);
  // What a CU holds is ceil(n / CUs) workgroups, each with one wave per SIMD running `mt` column tiles -- serial chains of
  // 27 Ci / 2 MFMAs -- one after the other; + 0.1: set-up and epilogue of a workgroup in units of one column tile's arithmetic.
  // (Round 4 counted rounds of `wpe` co-resident workgroups instead: the same ranking on full grids, but a step function at
  // n = wpe x CUs that misjudged launches around one round -- batch 1, small images.)  Checked against every candidate measured
  // so far: profiles/r04_kbench_hg*.log (batch 4: 240 / 120 / 60 and 312 / 156 columns) and profiles/r05_kbench_small.log (batch 1
  // at 256x512 / D 64, 544x960 and 384x1248) -- the estimate ranks them as measured.
  return (double)cdiv_ll(n, num_cus()) * (t.mt + 0.1) / t.eff;
}
// index of the cheapest candidate (ties: the earlier one); DMB_OPT(19) = k > 0 forces candidate k - 1 (development build)
static int s1_pick(const S1Tile* cand, const bool* ok, int n, int B, int D, int H, int W) {
  if (DMB_OPT(19) > 0 && DMB_OPT(19) <= n && ok[DMB_OPT(19) - 1]) return DMB_OPT(19) - 1;
  int best = -1;
  double bc = 0.0;
  for (int i = 0; i < n; ++i) {
    if (!ok[i]) continue;
    const double c = s1_cost(cand[i], B, D, H, W);
    if (best < 0 || c < bc) best = i, bc = c;
  }
  return best;
}
static int flat_tx(int W) {
  const int w60 = cdiv(W, 60) * 60, w52 = cdiv(W, 52) * 52;
  // useful fraction = W / padded width * (TX / (TX + 2)) * (TY*P / (MT*32))
  const double e60 = (double)W / w60 * (60.0 / 62.0) * (248.0 / 256.0);
  const double e52 = (double)W / w52 * (52.0 / 54.0) * (216.0 / 224.0);
  return e52 > e60 ? 52 : 60;
}

// The 32-channel row-pair / row-quad tiles of the vector path: 0 = 48 x 4, 1 = 24 x 8, 2 = 32 x 4, 3 = 32 x 2
static int s1_pick32(int B, int D, int H, int W) {
  // row pairs (16 columns x 2 rows per 32-voxel column tile) of 48 or 32 columns x 4 rows, row quads (8 x 4) of 24 columns x 8
  // rows; one z-slice per wave, three workgroups per CU each
  // (8-row quads stage a quarter more halo rows per byte of output and store 32-byte instead of 64-byte runs: 0.972)
  static const S1Tile cand[4] = {{48, 4, 4, 6, 3, false, 1.0}, {24, 8, 4, 6, 3, false, 0.972}, {32, 4, 4, 4, 3, false, 1.0},
                                 {32, 2, 4, 2, 3, false, S1_EFF_32x2}};
  const bool ok[4] = {true, true, true, true};
  int pick = s1_pick(cand, ok, 4, B, D, H, W);
  // (round 6) a launch of at most ONE 32 x 4 workgroup per CU: two 32 x 2 workgroups per CU instead, so that one's prologue and
  // epilogue fall under the other's multiply phase ([1, 32, 16, 64, 128]: 57.6 -> 55.7 us, profiles/r06_sk_probe_midsizes.log)
  if (pick == 2 && DMB_OPT(19) == 0 && (long long)B * cdiv(D, 4) * cdiv(H, 4) * cdiv(W, 32) <= num_cus()) pick = 3;
  return pick;
}
static long long s1_blocks32(int pick, int B, int D, int H, int W) {
  static const int tx[4] = {48, 24, 32, 32}, ty[4] = {4, 8, 4, 2};
  return (long long)B * cdiv(W, tx[pick]) * cdiv(H, ty[pick]) * cdiv(D, 4);
}

// ---- Which launches take the split-K form (csrc/conv3d_sk.hip): 0 = none, else its variant --------------------------------
// A full-grid kernel gives one wave the whole chain of 27 Ci / 2 MFMAs of a tile, in chunks of two channels behind a barrier each:
// with fewer tile chains than SIMDs a launch costs ~1 us per chunk whatever its arithmetic (31-35 us for 64 channels).  Split-K
// puts eight waves on a tile.  Units: 32-voxel column tiles (16 x 2 row pairs) x 32-channel row tiles.
static int sk_variant(int B, int Ci, int Co, int D, int H, int W, int stride) {
  const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const long long tiles = (long long)B * Do * cdiv(Ho, 2) * cdiv(Wo, 16);
  const int ntt = Co / 32, ncu = num_cus();
  // measured (profiles/r06_sk_probe.log, one MI355X): [1, 64, 4, 16, 32] 64 -> 64: stride 1 32.4 -> 12.8 us, stride 2 (from
  // [1, 64, 8, 32, 64]) 33.8 -> 14.9 us with one row tile per workgroup (128 workgroups); [1, 32, 16, 64, 128] -> 64 channels at
  // stride 2: 31.6 -> 22.0 us with 32 x 2 voxels x both row tiles (256 workgroups; 32 input channels = 2 pairs per wave keep the
  // A fragments of both row tiles in registers).  At 1500 tile chains and more the full-grid kernels win (each split-K workgroup
  // re-reads its share of the weights from L2: 0.45 of the matrix peak at best).
  if (tiles * ntt <= ncu + ncu / 2) return 1;
  if (stride == 2 && Ci == 32 && Co == 64 && tiles * ntt <= 4LL * ncu) return 3;
  return 0;
}

extern "C" int dmb_conv3d_k3_f32(const float* x, const float* wpack, const float* scale, const float* shift,
                                 const float* residual, float* y, int B, int Ci, int Co, int D, int H, int W,
                                 int stride, int relu, void* stream) {
  if (!x || !wpack || !y || B <= 0 || Ci <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "conv3d: bad argument");
  if ((long long)8 * D * H * W * 4 >= 0x7fffffffLL)
    return fail(DMB_EUNSUPPORTED, "conv3d: 8 channels of one batch item must stay below 2 GiB (32-bit buffer offsets)");
  const bool out_small = (long long)Co * D * H * W * 4 < 0x7fffffffLL;   // the vector epilogue addresses the whole output item
  hipStream_t st = (hipStream_t)stream;
  const bool single_chain = (relu & DMB_CONV_SINGLE_CHAIN) != 0;
  relu &= 0xff;
  relu |= (DMB_OPT(6) & 0xff) << 8;                    // (development build: diagnostics, see the kernels)
  const int relu_s1 = relu | (DMB_OPT(14) << 16);      // (development build: start-up stagger unit of the stride-1 kernels)
#define DMB_S1(CO, TY, TX, WN, G, PM) launch_s1<S1Cfg<0, CO, TY, TX, 2, WN, 1, G, PM>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu_s1, st)
  if ((stride == 1 || stride == 2) && (Co == 32 || Co == 64)) {
    // (round 6) launches that leave most of the chip idle: the K chain of a tile split over the eight waves of a workgroup
    // (csrc/conv3d_sk.hip).  DMB_OPT(23) (development build): 1 = never, k >= 2 = force variant k - 1.
    int v = single_chain ? 0 : sk_variant(B, Ci, Co, D, H, W, stride);
    if (DMB_OPT(23) == 1) v = 0;
    if (DMB_OPT(23) >= 2) v = DMB_OPT(23) - 1;
    if (v > 0) {
      const int rc = conv3d_sk_try(v, x, wpack, scale, shift, residual, y, B, Ci, Co, D, H, W, stride, relu & 0xff, st);
      if (rc != -1) return rc;
    }
  }
  if (stride == 1) {
    // the vector path: 16-byte rows for staging, skip operand and stores
    const bool vec = W % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) == 0 && out_small && DMB_OPT(2) == 0;
    const int tx = flat_tx(W);
    if (Co == 32) {
#ifdef DMB_DEV
      if (DMB_OPT(0) == 1) return launch_s1<S1Cfg<0, 32, 4, 60, 2, 1, 0, 0>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu_s1, st);
#endif
#ifdef DMB_DEV
      // (round 6 experiment) chunks of 4 input channels for the 32-column tiles
      if (vec && DMB_OPT(19) == 8) return launch_s1<S1Cfg<0, 32, 4, 32, 4, 1, 1, 16, 0>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu_s1, st);
      if (vec && DMB_OPT(19) == 9) return launch_s1<S1Cfg<0, 32, 2, 32, 4, 1, 1, 16, 0>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu_s1, st);
#endif
      if (vec) {
        switch (s1_pick32(B, D, H, W)) {
          case 0: return DMB_S1(32, 4, 48, 1, 16, 0);
          case 1: return DMB_S1(32, 8, 24, 1, 8, 40);
          case 2: return DMB_S1(32, 4, 32, 1, 16, 0);
          default: return DMB_S1(32, 2, 32, 1, 16, 0);
        }
      }
      return tx == 52 ? DMB_S1(32, 4, 52, 1, 0, 0) : DMB_S1(32, 4, 60, 1, 0, 0);
    }
    if (Co == 64) {
      if (vec) {
        // 64-voxel runs of the plane in memory order for rows of up to 64 voxels (planes no box tiling fills the chip with: the
        // deepest hourglass level), row quads of 40 / 24 / 32 columns x 4 rows otherwise; two z-slices x two channel tiles per
        // workgroup.  A launch here is only a few rounds of workgroups deep, so the estimate decides per launch.
        // (round 5) + row pairs of 16 columns x 2 rows: ONE column tile per wave, for launches that do not fill the chip (batch 1,
        // small images: 62 -> 37 us at [1, 64, 8, 32, 64], 59 -> 33 us at [1, 64, 4, 16, 32]); twice the halo per output of the quads
        static const S1Tile cand[5] = {{64, 3, 2, 2, 3, true, 1.0}, {40, 4, 2, 5, 3, false, 1.0}, {24, 4, 2, 3, 3, false, 1.0}, {32, 4, 2, 4, 3, false, 1.0},
                                       {16, 2, 2, 1, 2, false, 0.95}};
        const bool ok[5] = {W >= 32 && W <= 64 && (H * W) % 4 == 0 && DMB_OPT(13) == 0, true, true, true, true};
#ifdef DMB_DEV
        // (round 6 experiment) the 16 x 2 tile with chunks of 2 / 8 input channels (the library's: 4)
        if (DMB_OPT(19) == 6) return DMB_S1(64, 2, 16, 2, 16, 0);   // (the round-5 form of the tile: chunks of 2)
        if (DMB_OPT(19) == 7) return launch_s1<S1Cfg<0, 64, 2, 16, 8, 2, 1, 16, 0>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu_s1, st);
#endif
        switch (s1_pick(cand, ok, 5, B, D, H, W)) {
          case 0: return launch_s1<S1Cfg<0, 64, 3, 64, 2, 2, 1, 16, 0, true>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu_s1, st);
          case 1: return DMB_S1(64, 4, 40, 2, 8, 56);
          case 2: return DMB_S1(64, 4, 24, 2, 8, 40);
          case 3: return DMB_S1(64, 4, 32, 2, 8, 40);
          // (round 6: chunks of four input channels for this tile -- a chunk of two is 27 MFMAs per wave between two barriers, less than
          // the copies' round trip: [1, 64, 8, 32, 64] 33.5 -> 31.6 us, [1, 64, 12, 34, 60] 59.6 -> 56.0 us; two workgroups per CU)
          default: return launch_s1<S1Cfg<0, 64, 2, 16, 4, 2, 1, 16, 0>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu_s1, st);
        }
      }
      return tx == 52 ? DMB_S1(64, 4, 52, 2, 0, 0) : DMB_S1(64, 4, 60, 2, 0, 0);
    }
    if (Co == 128)   // GC-Net's deepest level: 2 waves x 2 row tiles, 2-row tiles keep the accumulators at 160 registers
      return launch_s1<S1Cfg<0, 128, 2, 60, 2, 2, 1, 0, 0>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu_s1, st);
  } else if (stride == 2) {
    const bool v16 = W % 4 == 0 && ((uintptr_t)x & 15) == 0 && !DMB_OPT(3);   // 16-byte aligned input rows
    if (Co == 64 && v16) {
      // output positions computed per tile row: 32 with 30-column tiles, 24 with 22-column ones; the narrower tile wins at
      // the training-crop widths (Wo = 64: 3 x 96 against 3 x 128 positions per 4 rows, Wo = 32: 2 x 96 against 2 x 128)
      const int Wo = (W - 1) / 2 + 1, Ho = (H - 1) / 2 + 1, Do = (D - 1) / 2 + 1;
      // 8-byte epilogue: W % 4 == 0 makes Wo even; the output (and skip operand) base must be 8-byte aligned and one batch item
      // of the output addressable with 32-bit byte offsets
      const bool pair_ok = ((((uintptr_t)y | (uintptr_t)residual) & 7) == 0) && (long long)Co * Do * Ho * Wo * 4 < 0x7fffffffLL;
      const bool narrow = cdiv(Wo, 22) * 96 < cdiv(Wo, 30) * 128;
      // (round 5) Tile height by estimate.  What a CU holds is ceil(workgroups / CUs) workgroups, and one SIMD runs a workgroup's
      // column tiles one after the other -- 4 with the 4 x 30 eight-wave tiles, 3 with 4 x 22, 2 with two-row tiles, 1 with one-row
      // tiles -- so finer tiles shorten the chain of a launch that leaves CUs idle (batch 1, small images: 112 -> 59 us at
      // [1, 64, 24, 68, 120]) and even out a grid that does not divide over the chip (432 four-row workgroups on 256 CUs: the
      // 64 -> 64 layer at batch 4, 201 -> 188 us on 1632 one-row ones), at the price of more halo rows per output: efficiencies
      // fitted to profiles/r05_s2_tile_probe.log and r05_kbench_small.log (same results bit for bit whatever the tile).
      const long long nz = (long long)B * cdiv(Do, 2);
      const int ncu = num_cus();
      const double c_old = (double)cdiv_ll(nz * (narrow ? cdiv(Wo, 22) : cdiv(Wo, 30)) * cdiv(Ho, 4), ncu) * (narrow ? 3.0 / 0.85 : 4.0);
      const double c_two = (double)cdiv_ll(nz * cdiv(Wo, 30) * cdiv(Ho, 2), ncu) * 2.0 / 0.97;
      const double c_one = (double)cdiv_ll(nz * cdiv(Wo, 30) * Ho, ncu) * 1.0 / 0.94;
#ifdef DMB_DEV
      if (pair_ok && DMB_OPT(10) == 3) return launch_s2<S2Cfg<0, 64, 1, 30, 2, 2, true, 1, true>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
      if (pair_ok && DMB_OPT(10) == 4) return launch_s2<S2Cfg<0, 64, 2, 30, 2, 2, true, 2, true>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
      // (round 6 experiment) chunks of 4 input channels
      if (pair_ok && DMB_OPT(10) == 5) return launch_s2<S2Cfg<0, 64, 1, 30, 4, 2, true, 1, true>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
      if (pair_ok && DMB_OPT(10) == 6) return launch_s2<S2Cfg<0, 64, 2, 30, 4, 2, true, 2, true>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
#endif
      if (pair_ok && DMB_OPT(10) == 0) {
        if (c_one < c_two && c_one < c_old)
          return launch_s2<S2Cfg<0, 64, 1, 30, 2, 2, true, 1, true>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
        if (c_two < c_old)
          return launch_s2<S2Cfg<0, 64, 2, 30, 2, 2, true, 2, true>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
      }
      // output positions computed per tile row: 32 with 30-column tiles, 24 with 22-column ones; the narrower tile wins at
      // the training-crop widths (Wo = 64: 3 x 96 against 3 x 128 positions per 4 rows, Wo = 32: 2 x 96 against 2 x 128)
      if (narrow)
        return launch_s2<S2Cfg<0, 64, 4, 22, 2, 2, true>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
#ifdef DMB_DEV
      if (DMB_OPT(10) == 1)   // A/B: four-wave workgroups
        return launch_s2<S2Cfg<0, 64, 4, 30, 2, 2, true>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
#endif
      if (pair_ok && DMB_OPT(10) != 2)   // (A/B: 10 = 2 keeps the dword epilogue)
        return launch_s2<S2Cfg<0, 64, 4, 30, 2, 2, true, 2, true>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
      return launch_s2<S2Cfg<0, 64, 4, 30, 2, 2, true, 2>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
    }
    if (Co == 64) return launch_s2<S2Cfg<0, 64, 4, 30, 2, 2>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
    if (Co == 32) return launch_s2<S2Cfg<0, 32, 4, 30, 2, 1>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
    if (Co == 128) return launch_s2<S2Cfg<0, 128, 4, 30, 2, 4>>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
  }
#undef DMB_S1
  return fail(DMB_EUNSUPPORTED, "conv3d: output channels must be 32 or 64 (or 1: dmb_conv3d_k3_c1_f32), stride 1 or 2");
}

// Raw stride-1 convolution Ci -> 32 + the batch statistics' partial sums (training path).  Applies to the vector path only.
static bool s1_stats_applicable(const float* x, const float* y, int B, int Ci, int Co, int D, int H, int W) {
  return Co == 32 && Ci > 0 && W % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && (long long)Co * D * H * W * 4 < 0x7fffffffLL &&
         (long long)8 * D * H * W * 4 < 0x7fffffffLL && DMB_OPT(2) == 0;
}
extern "C" long long dmb_conv3d_k3_bnstats_partials(int B, int Ci, int Co, int D, int H, int W) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || Co != 32 || Ci <= 0 || W % 4 != 0) return 0;
  if ((long long)Co * D * H * W * 4 >= 0x7fffffffLL || (long long)8 * D * H * W * 4 >= 0x7fffffffLL) return 0;
  return s1_blocks32(s1_pick32(B, D, H, W), B, D, H, W);
}
extern "C" int dmb_conv3d_k3_bnstats_f32(const float* x, const float* wpack, float* y, double* stats, int B, int Ci, int Co, int D, int H,
                                         int W, void* stream) {
  if (!x || !wpack || !y || !stats || B <= 0 || Ci <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "conv3d_bnstats: bad argument");
  if (!s1_stats_applicable(x, y, B, Ci, Co, D, H, W)) return fail(DMB_EUNSUPPORTED, "conv3d_bnstats: 32 output channels, 16-byte rows");
  hipStream_t st = (hipStream_t)stream;
  switch (s1_pick32(B, D, H, W)) {
    case 0: return launch_s1_stats<S1Cfg<0, 32, 4, 48, 2, 1, 1, 16, 0>>(x, wpack, y, stats, B, Ci, D, H, W, st);
    case 1: return launch_s1_stats<S1Cfg<0, 32, 8, 24, 2, 1, 1, 8, 40>>(x, wpack, y, stats, B, Ci, D, H, W, st);
    case 2: return launch_s1_stats<S1Cfg<0, 32, 4, 32, 2, 1, 1, 16, 0>>(x, wpack, y, stats, B, Ci, D, H, W, st);
    default: return launch_s1_stats<S1Cfg<0, 32, 2, 32, 2, 1, 1, 16, 0>>(x, wpack, y, stats, B, Ci, D, H, W, st);
  }
}

// Split-K form of the transposed convolution (csrc/conv3d_sk.hip): 0 = no, else its variant.  Units: input-resolution tiles.
static int dsk_variant(int B, int Ci, int Co, int D, int H, int W) {
  const long long tiles = (long long)B * D * cdiv(H, 2) * cdiv(W, 16);
  const int ncu = num_cus();
  // measured (profiles/r06_sk_probe_deconv.log, persistent form, four waves per workgroup; units = tiles x row tiles):
  // 128 units ([1, 64, 4, 16, 32] -> 64 channels) 37.3 -> 14.2 us, 192 units 38.0 -> 16.7, 360 units 37.8 -> 26.7, 576 units 59.3 -> 40.3,
  // 1632 units 99 against 103 (a tie); 512 units ([1, 64, 8, 32, 64] -> 32 channels) 50.2 -> 36.9, 768 units 69.2 -> 52.5, 1200 units
  // 80.6 -> 76.6, 2304 units 136 against 143: from there on the work-queue kernel (deconv3d_zy.hip) wins
  const long long units = tiles * cdiv(Co, 32);
  if (Ci % 8 == 0 && Ci <= 64 && units <= 5LL * ncu) return 3;   // 16 x 2 input positions per workgroup, four waves
  return 0;
}

extern "C" int dmb_deconv3d_k3s2_f32(const float* x, const float* wpack, const float* scale, const float* shift,
                                     const float* residual, float* y, int B, int Ci, int Co, int D, int H, int W, int Wout,
                                     int relu, void* workspace, void* stream) {
  if (!x || !wpack || !y || B <= 0 || Ci <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "deconv3d: bad argument");
  if (Wout <= 0 || Wout > 2 * W || Wout <= 2 * W - 8 || (Wout & 1)) return fail(DMB_EINVAL, "deconv3d: Wout must be 2 W, or 2 W minus the (at most 3) doubled padding columns");
  if ((long long)8 * D * H * W * 4 >= 0x7fffffffLL)
    return fail(DMB_EUNSUPPORTED, "deconv3d: 8 input channels of one batch item must stay below 2 GiB (32-bit buffer offsets)");
  hipStream_t st = (hipStream_t)stream;
  const bool single_chain = (relu & DMB_CONV_SINGLE_CHAIN) != 0;
  relu &= 0xff;
  relu |= DMB_OPT(6) << 8;   // (development build: diagnostics)
  {
    // (round 6) launches that leave most of the chip idle: split-K over the waves of a workgroup (csrc/conv3d_sk.hip).
    // DMB_OPT(25) (development build): 1 = never, k >= 2 = force variant k - 1.
    int v = single_chain ? 0 : dsk_variant(B, Ci, Co, D, H, W);
    if (DMB_OPT(25) == 1) v = 0;
    if (DMB_OPT(25) >= 2) v = DMB_OPT(25) - 1;
    if (v > 0) {
      const int rc = deconv3d_sk_try(v, x, wpack, scale, shift, residual, y, B, Ci, Co, D, H, W, Wout, relu & 0xff, st);
      if (rc != -1) return rc;
    }
  }
  if (workspace && !DMB_OPT(3) && !DMB_OPT(7)) {   // three workgroups per CU where the shape admits it (csrc/deconv3d_zy.hip)
    const int rc = deconv3d_zy_try(x, wpack, scale, shift, residual, y, B, Ci, Co, D, H, W, Wout, relu, static_cast<int*>(workspace), st);
    if (rc != -1) return rc;
  }
  if (Wout != 2 * W) return fail(DMB_EUNSUPPORTED, "deconv3d: a row-padded input (Wout < 2 W) needs the workspace form: Co 32 or 64, Ci % 16 == 0, Wout % 4 == 0, aligned operands");
  const bool v16 = W % 4 == 0 && ((uintptr_t)x & 15) == 0 && !DMB_OPT(3);   // 16-byte aligned rows
  // 16-byte epilogue: aligned output / residual rows, every channel real, one batch item of the output below 2 GiB
  const bool vepi = v16 && Co % 32 == 0 && ((((uintptr_t)y | (uintptr_t)residual) & 15) == 0) &&
                    (long long)Co * 8 * D * H * W * 4 < 0x7fffffffLL && !DMB_OPT(7);
  // 2 rows x 28 columns per item instead of 1 x 60 where that computes fewer positions (input W = 64: 3 x 32 against
  // 2 x 64 per row; the tile width stays a multiple of 4 for the 16-byte staging)
  const bool narrow = v16 && cdiv(W, 28) * 32 < cdiv(W, 60) * 64;
#define DMB_DC(CO, TY, TX, CK, WN, V, E) launch_deconv<DCfg<0, CO, TY, TX, CK, WN, V, E>>(x, wpack, scale, shift, residual, y, B, Ci, Co, D, H, W, relu, st)
  if (Co == 64 && narrow) return vepi ? DMB_DC(64, 2, 28, 4, 2, true, true) : DMB_DC(64, 2, 28, 4, 2, true, false);
  if (Co <= 32 && narrow) return vepi ? DMB_DC(32, 2, 28, 8, 1, true, true) : DMB_DC(32, 2, 28, 8, 1, true, false);
  if (Co == 64) return !v16 ? DMB_DC(64, 1, 60, 4, 2, false, false) : (vepi ? DMB_DC(64, 1, 60, 4, 2, true, true) : DMB_DC(64, 1, 60, 4, 2, true, false));
  if (Co <= 32)   // fewer than 32 channels (GC-Net's 1-channel head): zero-padded weight rows, masked scalar epilogue
    return !v16 ? DMB_DC(32, 1, 60, 8, 1, false, false) : (vepi ? DMB_DC(32, 1, 60, 8, 1, true, true) : DMB_DC(32, 1, 60, 8, 1, true, false));
#undef DMB_DC
  return fail(DMB_EUNSUPPORTED, "deconv3d: output channels must be <= 32 or 64");
}

extern "C" int dmb_conv3d_k3_c1_f32(const float* x, const float* w, float bias, const float* residual, float* y,
                                    int B, int Ci, int D, int H, int W, int flags, void* stream) {
  if (!x || !w || !y || B <= 0 || Ci <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "conv3d_c1: bad argument");
  const int ntx = cdiv(W, C1_TX), nty = cdiv(H, C1_TY), ntz = cdiv(D, C1_TZ);
  const long long nblk = (long long)B * ntx * nty * ntz;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv3d_c1: grid too large");
  const size_t lds = (size_t)2 * C1_BUF * sizeof(float);
  DMB_ENSURE_LDS((&conv3d_c1_kernel), (size_t)(lds));
  if ((long long)2 * D * H * W * 4 >= 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv3d_c1: tensor too large for 32-bit offsets");
  if (W % 4 == 0 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) == 0) && !DMB_OPT(3)) {   // 16-byte rows
    const int vx = cdiv(W, C1V_TX), vy = cdiv(H, C1V_TY), vz = cdiv(D, C1V_TZ);
    // (round 6) fewer tiles than one round of two workgroups per CU: the channels split over the waves of a workgroup
    // (conv3d_c1s_kernel).  DMB_OPT(24) (development build): 1 = never, 2 = always.
    // (tiles of the single-chain kernel: 48 ([1, 32, 16, 64, 128]) 52 -> 18 us, 72 - 90: 56 - 60 -> 33 us, 192: 62 -> 48 us, 408: 70 against 107 us)
    const bool small = !(flags & DMB_CONV_SINGLE_CHAIN) && (long long)B * vx * vy * vz <= num_cus();
    if ((small && DMB_OPT(24) != 1) || DMB_OPT(24) == 2) {
      const int sz = cdiv(D, C1S_TZ);
      hipLaunchKernelGGL(conv3d_c1s_kernel, dim3((unsigned)((long long)B * vx * vy * sz)), dim3(256), 0, (hipStream_t)stream, x, w,
                         bias, residual, y, Ci, D, H, W, vx, vy, sz);
      return launch_status("conv3d_c1 launch failed");
    }
    hipLaunchKernelGGL(conv3d_c1v_kernel, dim3((unsigned)((long long)B * vx * vy * vz)), dim3(256), 0, (hipStream_t)stream, x, w,
                       bias, residual, y, Ci, D, H, W, vx, vy, vz);
    return launch_status("conv3d_c1 launch failed");
  }
  hipLaunchKernelGGL(conv3d_c1_kernel, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, x, w, bias, residual, y,
                     Ci, D, H, W, ntx, nty, ntz);
  return launch_status("conv3d_c1 launch failed");
}
