// Transposed 3-D convolution k3 s2 p1 op1, 64 -> 32 channels at half -> full resolution (hourglass conv6:
// utils/hourglass.py:53-60,84-86 through layers/basic_layers.py:160-177), fourth form: ONE sixteen-wave workgroup per CU
// computes all eight output parity classes of an input tile from ONE staged copy of the tile.
//
//   y[2i - 1 + k] += x[i] * w[k]  per axis:  an even output o = 2i sees k = 1 from input i; an odd output o = 2i + 1 sees
//   k = 2 from input i and k = 0 from input i + 1.
//
// What the other forms leave on the table for this layer (measured, scripts/kbench_hg.py diag):
//   * deconv3d_zy_kernel (items = (tile, z parity, y parity), three four-wave workgroups per CU) multiplies at 0.82 of the
//     matrix peak on its own (48 MFMAs between barriers) and stages every input tile FOUR times (2.1 GB of L2 misses for a
//     0.2 GB input); with the layer's 0.8 GB of output and 0.8 GB of skip operand on top the fabric carries 4.2 TB/s on
//     average, the chunk copies land late and 0.14 ms of the 0.86 are barrier waits on them (confining the epilogue's
//     traffic to an L2-resident window -- same instructions, no DRAM -- gives 0.73 ms);
//   * deconv3d_kernel (both y parities per item) stages a tile twice and holds 128 accumulators: two workgroups per CU.
// Here a workgroup is 4 classes x 4 input planes = 16 waves on one tile of 4 planes x 1 row x 60 columns:
//   * wave = (class c = (z parity, y parity), plane wz); waves are numbered c * 4 + wz, so each of the CU's four SIMDs hosts
//     one wave of every class: the classes' 1 : 2 : 2 : 4 arithmetic is balanced PER SIMD, and the light waves simply reach
//     the chunk barrier early while the heavy one keeps the matrix pipe busy;
//   * the tile (5 planes x 2 rows x 64 columns with its halo) and the chunk's weights (ALL 27 taps: 27 KB per 8 input
//     channels) are staged once per chunk for all classes: 0.53 GB of input fetches instead of 2.1; a chunk is 216 MFMAs per
//     SIMD (13.8 k cycles) between barriers instead of 48, so a copy has 5.8 us to land;
//   * 64 accumulators per wave (two x parities x two 32-column tiles), 128 registers: four waves per SIMD;
//   * items (tiles) come from an atomic counter as in deconv3d_zy_kernel; the chunk pipeline runs across items.
// Same FP32 products and the same ascending (channel, kz, ky, kx) fma chain per output as the other forms: bit-identical.
// STATUS: an experiment kept behind development option 11 (tested bit-identical; see deconv3d_w16_try for the measurement).
#include <type_traits>

#include <mutex>

#include "dmb_common.h"

namespace dmb {

struct W16Cfg {
  static constexpr int COUT = 32, TZ = 4, TX = 60, P = 64, MT = 2;
  static constexpr int NWAVES = 16, NTHREADS = 64 * NWAVES;
  static constexpr int CK = 8;                        // input channels per chunk
  static constexpr int ZS = TZ + 1, ROWS = 2, PLANE = ROWS * P;
  // no padding between channels: a chunk's input units are linear in LDS (unit u at float 4 u), and the two lane halves of
  // a B read (channels 2 cp and 2 cp + 1) are served in different LDS cycles anyway (ds_read_b32: lanes 0-31, then 32-63)
  static constexpr int CH_STRIDE = ZS * PLANE;
  static constexpr int IN_FLOATS = CK * CH_STRIDE;
  static constexpr int W_FLOATS = (CK / 2) * 27 * 64; // all taps of the chunk's channel pairs, in the packed (global) order
  static constexpr int BUF_FLOATS = IN_FLOATS + W_FLOATS;
  static constexpr int UPC = ZS * ROWS * (P / 4);     // 16-byte units per channel
  static constexpr int IN_UNITS = CK * UPC, W_UNITS = W_FLOATS / 4;
  static constexpr int IN_PIECES = (IN_UNITS + NTHREADS - 1) / NTHREADS, W_PIECES = (W_UNITS + NTHREADS - 1) / NTHREADS;
  static constexpr int NPIECE = IN_PIECES + W_PIECES;
  static constexpr int SCR_PITCH = 68, PCH = 8;       // epilogue scratch: 8 channels x 64 output columns per pass and wave
  static constexpr int AFF_FLOATS = 2 * COUT;
  static constexpr int SCR_FLOATS = NWAVES * PCH * SCR_PITCH;
  static constexpr int LDS_FLOATS = 2 * BUF_FLOATS + AFF_FLOATS + SCR_FLOATS;
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "one workgroup per CU");
  static_assert(IN_FLOATS % 4 == 0 && W_FLOATS % 4 == 0, "16-byte copies");
};

struct W16Args {
  const float* x;
  const float* wp;
  const float* res;
  float* y;
  int* counter;
  unsigned base;   // value of the counter when this launch starts
  int Ci, D, H, W, ntx, nty, ntz, ntiles, relu, dbg;
  int stagger;   // start-up delay unit (see the kernel)
};

// The whole life of one wave: class (PZ, PY) is fixed, items arrive through `slot`.
template <int PZ, int PY>
__device__ __forceinline__ void w16_body(float* lds, int* slot, const W16Args& a, int item, int tid, int wz) {
  using C = W16Cfg;
  constexpr int NAZ = 1 + PZ, NAY = 1 + PY, NA = NAZ * NAY;   // (kz, ky) pairs an output of the class sees
  constexpr int NU = (C::CK / 2) * NA;                        // (channel pair, kz, ky) units per chunk: 4 / 8 / 8 / 16
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  const int D = a.D, H = a.H, W = a.W, Ci = a.Ci;
  const unsigned HW = (unsigned)H * W, DHW = (unsigned)D * HW;
  const int Ho = 2 * H, Wo = 2 * W;
  const unsigned HWo = (unsigned)Ho * Wo, DHWo = 2u * D * HWo;
  float* aff = lds + 2 * C::BUF_FLOATS;
  float* scr = aff + C::AFF_FLOATS + wave * (C::PCH * C::SCR_PITCH);

  struct Tile {
    int b, x0, y0, z0;
  };
  auto tile_at = [&](int t) {
    Tile tl;
    tl.x0 = (t % a.ntx) * C::TX;
    t /= a.ntx;
    tl.y0 = t % a.nty;
    t /= a.nty;
    tl.z0 = (t % a.ntz) * C::TZ;
    tl.b = t / a.ntz;
    return tl;
  };

  // ---- copies.  Input: unit = 4 consecutive floats of a staged row; the units of a chunk ([channel][plane][row][16 units],
  // channels CH_STRIDE apart) are dealt to the 1024 threads in IN_PIECES rounds; a unit's source offset inside the chunk
  // depends on the tile only.  Weights: the chunk's block of the packed tensor is copied as it lies.
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.wp, (unsigned)(Ci * 27 * C::COUT) * 4u);
  auto tile_offsets = [&](const Tile& tl, unsigned (&o)[C::IN_PIECES]) {
#pragma unroll
    for (int i = 0; i < C::IN_PIECES; ++i) {
      const int u = i * C::NTHREADS + tid;
      const int cl = u / C::UPC, r = u - cl * C::UPC;
      const int zz = r / (C::ROWS * 16), rr = r - zz * (C::ROWS * 16), yy = rr / 16, sg = rr - yy * 16;
      const int gz = tl.z0 + zz, gy = tl.y0 + yy, gx = tl.x0 + sg * 4;
      o[i] = (u < C::IN_UNITS && gz < D && gy < H && gx < W)
                 ? ((unsigned)cl * DHW + (unsigned)gz * HW + (unsigned)gy * W + (unsigned)gx) * 4u : DMA_OOB;
    }
  };
  auto in_rsrc = [&](const Tile& tl, int c0) {
    return make_rsrc(a.x + ((size_t)tl.b * Ci + c0) * DHW, (unsigned)C::CK * DHW * 4u);
  };
  auto stage = [&](const __amdgpu_buffer_rsrc_t xrs, const unsigned (&toff)[C::IN_PIECES], int c0, float* buf, int lo, int hi2) {
#pragma unroll
    for (int i = 0; i < C::IN_PIECES; ++i) {
      if (i < lo || i >= hi2) continue;
      if (C::IN_UNITS % C::NTHREADS == 0 || i * C::NTHREADS + tid < C::IN_UNITS)
        dma16(xrs, toff[i], 0u, buf + (i * C::NTHREADS + wave * 64) * 4);
    }
#pragma unroll
    for (int i = 0; i < C::W_PIECES; ++i) {
      if (C::IN_PIECES + i < lo || C::IN_PIECES + i >= hi2) continue;
      const int q4 = i * C::NTHREADS + tid;
      if (C::W_UNITS % C::NTHREADS == 0 || q4 < C::W_UNITS)
        dma16(wrs, (unsigned)q4 * 16u, (unsigned)(c0 / 2) * (27 * 64 * 4), buf + C::IN_FLOATS + (i * C::NTHREADS + wave * 64) * 4);
    }
  };

  const int NC = (a.dbg & 64) ? 2 : Ci / C::CK;
  Tile cur_t = tile_at(item);
  unsigned coff[C::IN_PIECES], noff[C::IN_PIECES];
  tile_offsets(cur_t, coff);
  stage(in_rsrc(cur_t, 0), coff, 0, lds, 0, C::NPIECE);
  __syncthreads();
  int g = 0;   // chunks consumed so far: selects the LDS buffer
  for (;;) {
    int next = a.ntiles;
    bool has_next = false;
    Tile next_t = cur_t;
    int fetched = 0;

    f32x16 acc[2][C::MT];   // [x parity][column tile]
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[px][mt][r] = 0.f;

#pragma clang loop unroll(disable)
    for (int ci = 0; ci < NC; ++ci, ++g) {
      const float* cur = lds + (g & 1) * C::BUF_FLOATS;
      float* nxt = lds + ((g + 1) & 1) * C::BUF_FLOATS;
      if (ci == 0 && tid == 0) fetched = (int)((unsigned)atomicAdd(a.counter, 1) - a.base);
      // the next chunk's copies (of this item, or the first chunk of the next one) are dealt out over the first units
      const bool more = ci + 1 < NC;
      const bool staging = !(a.dbg & 2) && (more || has_next);
      const int st_c0 = more ? (ci + 1) * C::CK : 0;
      const __amdgpu_buffer_rsrc_t st_rs = in_rsrc(more ? cur_t : next_t, st_c0);
      unsigned st_off[C::IN_PIECES];
#pragma unroll
      for (int q = 0; q < C::IN_PIECES; ++q) st_off[q] = more ? coff[q] : noff[q];
      constexpr int SU = NU >= C::NPIECE ? C::NPIECE : NU, PPU = (C::NPIECE + SU - 1) / SU;
      auto deal = [&](int u) {
        if (staging) stage(st_rs, st_off, st_c0, nxt, u * PPU, (u + 1) * PPU);
      };
      const float* abase = cur + C::IN_FLOATS + lane;
      const float* bbase = cur + h * C::CH_STRIDE + wz * C::PLANE + j;
      float af[2][3], bf[2][2][C::MT];   // af[buf][kx], bf[buf][ox][mt]
      auto load_frag = [&](int u, float (&fa)[3], float (&fb)[2][C::MT]) {
        const int cp = u / NA, ta = u % NA, az = ta / NAY, ay = ta % NAY;
        const int kz = PZ ? (az ? 0 : 2) : 1, ky = PY ? (ay ? 0 : 2) : 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) fa[k] = abase[(cp * 27 + kz * 9 + ky * 3 + k) * 64];
        const float* bp = bbase + 2 * cp * C::CH_STRIDE + az * C::PLANE + ay * C::P;
#pragma unroll
        for (int ox = 0; ox < 2; ++ox)
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) fb[ox][mt] = bp[ox + mt * 32];
      };
      load_frag(0, af[0], bf[0]);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (u + 1 < NU) load_frag(u + 1, af[(u + 1) & 1], bf[(u + 1) & 1]);
        if (u < SU) deal(u);
        __builtin_amdgcn_sched_barrier(0);
        const auto& fa = af[u & 1];
        const auto& fb = bf[u & 1];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
          acc[0][mt] = DMB_MFMA(fa[1], fb[0][mt], acc[0][mt]);   // even x: kx = 1 from input x
          acc[1][mt] = DMB_MFMA(fa[2], fb[0][mt], acc[1][mt]);   // odd x:  kx = 2 from input x
          acc[1][mt] = DMB_MFMA(fa[0], fb[1][mt], acc[1][mt]);   //         kx = 0 from input x + 1
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ci == 0 && tid == 0) __atomic_store_n(slot, fetched, __ATOMIC_RELAXED);
      __syncthreads();
      if (ci == 0) {
        next = __builtin_amdgcn_readfirstlane(__atomic_load_n(slot, __ATOMIC_RELAXED));
        has_next = next < a.ntiles && !(a.dbg & 16);
        if (has_next) {
          next_t = tile_at(next);
          tile_offsets(next_t, noff);
        }
      }
    }

    // ---- epilogue of cur_t (after the item's last barrier: the wave's accumulators are final, the chunk buffers belong to
    // the next item; the scratch is this wave's own).  Per (column tile, 8-channel group) the two x-parity accumulator tiles
    // are interleaved in LDS ([8 channels][64 output columns]), read back as 4 consecutive x of one channel and stored /
    // residual-loaded as 16-byte words (lanes outside the volume get an out-of-range offset: branch-free).
    if (!(a.dbg & 1)) {
      const int gzi = cur_t.z0 + wz;
      float* yb = a.y + (size_t)cur_t.b * C::COUT * DHWo;
      const __amdgpu_buffer_rsrc_t yrs = make_rsrc(yb, (unsigned)C::COUT * DHWo * 4u);
      const __amdgpu_buffer_rsrc_t rrs = make_rsrc(a.res ? a.res + (size_t)cur_t.b * C::COUT * DHWo : yb, (unsigned)C::COUT * DHWo * 4u);
      const int rl = lane >> 4, x4 = (lane & 15) * 4;
      const float lo = a.relu == 1 ? 0.f : -__builtin_inff();    // ReLU after the residual add
      const float lo2 = a.relu == 2 ? 0.f : -__builtin_inff();   // ReLU before it (GC-Net)
      constexpr int QP = 32 / C::PCH, KP = C::PCH / 4;          // passes per 32-channel tile, 16-byte words per lane and pass
      constexpr int NPASS = C::MT * QP;
      unsigned voff[C::MT];
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
        const int lx = mt * 32 + x4 / 2;
        const bool ok = gzi < D && lx < C::TX && cur_t.x0 + lx < W && !(a.dbg & 4);
        voff[mt] = ok ? ((a.dbg & 8) ? (unsigned)lane * 16u : ((unsigned)rl * DHWo + (unsigned)(2 * mt * 32 + x4)) * 4u) : DMA_OOB;
      }
      const unsigned sbase = ((unsigned)(2 * gzi + PZ) * HWo + (unsigned)(2 * cur_t.y0 + PY) * Wo + 2u * (unsigned)cur_t.x0) * 4u;
      const unsigned sstep = 4u * DHWo * 4u;                      // four channels on
      auto soff = [&](int t, int k) {
        const unsigned o = sbase + (unsigned)((t % QP) * KP + k) * sstep;
        return (a.dbg & 8) ? (o & 0x1fff80u) : o;   // development: epilogue traffic confined to a 2 MiB window
      };
      const float* affl = aff + rl;                               // + 8 q + 4 k: immediate offsets
      float* swr = scr + 4 * h * C::SCR_PITCH + 2 * j;           // + rr * pitch
      const float* srd = scr + rl * C::SCR_PITCH + x4;           // + 4 k * pitch
      auto run = [&](auto has_res) {
        constexpr bool HAS_RES = decltype(has_res)::value;
        constexpr int RD = HAS_RES ? 2 : 1;
        u32x4 rv[RD][KP];
        if constexpr (HAS_RES) {
#pragma unroll
          for (int t = 0; t < RD; ++t)
#pragma unroll
            for (int k = 0; k < KP; ++k) rv[t][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)voff[t / QP], (int)soff(t, k), 0);
        }
#pragma unroll
        for (int t = 0; t < NPASS; ++t) {
          const int q = t % QP, mt = t / QP;
          const int sl = t % RD;
#pragma unroll
          for (int rr = 0; rr < C::PCH / 2; ++rr) {
            const int r = q * (C::PCH / 2) + rr;   // accumulator register r of lane half h = channel 8 q + rr + 4 h of the tile
            *reinterpret_cast<float2*>(swr + rr * C::SCR_PITCH) = make_float2(acc[0][mt][r], acc[1][mt][r]);
          }
#pragma unroll
          for (int k = 0; k < KP; ++k) {
            float4 v = *reinterpret_cast<const float4*>(srd + 4 * k * C::SCR_PITCH);
            const float sc = affl[q * C::PCH + 4 * k], sh = affl[C::COUT + q * C::PCH + 4 * k];
            v.x = fmaxf(fmaf(v.x, sc, sh), lo2);
            v.y = fmaxf(fmaf(v.y, sc, sh), lo2);
            v.z = fmaxf(fmaf(v.z, sc, sh), lo2);
            v.w = fmaxf(fmaf(v.w, sc, sh), lo2);
            if constexpr (HAS_RES) {   // (not __builtin_bit_cast on a vector element: this clang reads element 0 for every index)
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
            // (uniform part in the VECTOR offset: a 16-byte store with an SGPR soffset lets the next VALU instruction
            // overwrite its data registers too early on this chip -- see deconv3d_zy.hip)
            __builtin_amdgcn_raw_buffer_store_b128(o, yrs, (int)(voff[mt] + soff(t, k)), 0, 0);
          }
          if constexpr (HAS_RES) {
            if (t + RD < NPASS) {   // refill the slot just consumed
#pragma unroll
              for (int k = 0; k < KP; ++k)
                rv[sl][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)voff[(t + RD) / QP], (int)soff(t + RD, k), 0);
            }
          }
        }
      };
      if (a.res)
        run(std::true_type{});
      else
        run(std::false_type{});
    }
    if (!has_next) {
      // drain: the items this workgroup will never process still advance nothing; every wave leaves together
      return;
    }
    cur_t = next_t;
#pragma unroll
    for (int q = 0; q < C::IN_PIECES; ++q) coff[q] = noff[q];
  }
}

__global__ __launch_bounds__(W16Cfg::NTHREADS, 4) void deconv3d_w16_kernel(W16Args a, const float* __restrict__ scale,
                                                                           const float* __restrict__ shift) {
  using C = W16Cfg;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* aff = lds + 2 * C::BUF_FLOATS;
  __shared__ int slot[1];   // the next item, published by thread 0
  // De-phase the workgroups across the chip (see deconv3d_zy_kernel): without it all 256 workgroups reach their epilogues
  // together and the layer's output / skip traffic comes as chip-wide bursts.
  if (a.stagger > 0) {
    const int n = (int)((blockIdx.x * 7u) & 15u) * a.stagger;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }
  if (threadIdx.x < C::COUT) {
    aff[threadIdx.x] = scale ? scale[threadIdx.x] : 1.f;
    aff[C::COUT + threadIdx.x] = shift ? shift[threadIdx.x] : 0.f;
  }
  if (threadIdx.x == 0) __atomic_store_n(slot, (int)((unsigned)atomicAdd(a.counter, 1) - a.base), __ATOMIC_RELAXED);
  __syncthreads();
  const int item = __builtin_amdgcn_readfirstlane(__atomic_load_n(slot, __ATOMIC_RELAXED));
  if (item >= a.ntiles) return;
  __syncthreads();   // (the slot is rewritten during the first item's first chunk: everyone has read it)
  int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave = class * 4 + plane: waves w, w + 4, w + 8, w + 12 (one of each class) share SIMD w % 4
  const int cls = wave >> 2, wz = wave & 3;
  if (cls == 0)
    w16_body<1, 1>(lds, slot, a, item, tid, wz);
  else if (cls == 1)
    w16_body<1, 0>(lds, slot, a, item, tid, wz);
  else if (cls == 2)
    w16_body<0, 1>(lds, slot, a, item, tid, wz);
  else
    w16_body<0, 0>(lds, slot, a, item, tid, wz);
}

struct W16Ticket {
  int* counter;
  unsigned base;
  bool* dirty;
};
static W16Ticket w16_ticket(unsigned advance, hipStream_t st) {
  constexpr int RING = 256;
  static int* ring[64] = {};
  static unsigned basev[64][RING] = {};
  static bool dirty[64][RING] = {};
  static unsigned seq = 0;
  static std::mutex mu;   // (see zy_ticket)
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return {nullptr, 0u, nullptr};
  if (!ring[dev]) {
    if (hipMalloc(reinterpret_cast<void**>(&ring[dev]), RING * sizeof(int)) != hipSuccess ||
        hipMemset(ring[dev], 0, RING * sizeof(int)) != hipSuccess) {
      ring[dev] = nullptr;
      return {nullptr, 0u, nullptr};
    }
  }
  const unsigned s = seq++ % RING;
  if (dirty[dev][s]) {   // a launch on this slot failed: zero it before it is trusted again
    if (hipMemsetAsync(ring[dev] + s, 0, sizeof(int), st) != hipSuccess) return {nullptr, 0u, nullptr};
    basev[dev][s] = 0;
    dirty[dev][s] = false;
  }
  W16Ticket t{ring[dev] + s, basev[dev][s], &dirty[dev][s]};
  basev[dev][s] += advance;
  return t;
}

// Entry for dmb_deconv3d_k3s2_f32 (conv3d.hip): -1 when this form does not apply or is not asked for (see below).
int deconv3d_w16_try(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                     int B, int Ci, int Co, int D, int H, int W, int relu, hipStream_t st) {
  using C = W16Cfg;
  if (g_dev_opts[4] != 0) return -1;   // 1: deconv3d_kernel, 2: deconv3d_zy_kernel
  if (Co != 32 || Ci % C::CK != 0 || Ci < 2 * C::CK || W % 4 != 0) return -1;
  if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) != 0) return -1;
  if ((long long)C::CK * D * H * W * 4 >= 0x7fffffffLL || (long long)Co * 8 * D * H * W * 4 >= 0x7fffffffLL) return -1;
  if (cdiv(W, 28) * 32 < cdiv(W, 60) * 64) return -1;
  const int ntx = cdiv(W, C::TX), nty = H, ntz = cdiv(D, C::TZ);
  const long long ntiles = (long long)B * ntx * nty * ntz;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  // OPT-IN (development option 11 = 1: wherever the shape admits it; = 2: only where a launch is >= 6 tiles per CU deep).
  // Measured on the layer it was written for (conv6 of the PSMNet hourglass, batch 4): 0.737 ms without and 0.894 ms with
  // the skip operand against 0.770 / 0.882 ms for deconv3d_zy_kernel; inside the whole step the two are equal
  // (26.997 vs 26.984 ms, scripts/ab_step.py) -- with one workgroup per CU nothing multiplies while the sixteen waves wait
  // for their skip-operand loads, which costs what the shorter staging wins.  Not selected by default.
  if (g_dev_opts[11] == 0) return -1;
  if (ntiles < 6LL * ncu && g_dev_opts[11] == 2) return -1;
  if (ntiles > 0x3fffffffLL) return -1;
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  DMB_ENSURE_LDS((&deconv3d_w16_kernel), lds);
  long long grid = ntiles < ncu ? ntiles : ncu;
  if (g_dev_opts[9] > 0 && g_dev_opts[9] < grid) grid = g_dev_opts[9];
  const W16Ticket tk = w16_ticket((unsigned)(ntiles + grid), st);
  if (!tk.counter) return fail(DMB_EINVAL, "deconv3d: could not set up the work-item counter");
  W16Args a{x, wp, res, y, tk.counter, tk.base, Ci, D, H, W, ntx, nty, ntz, (int)ntiles, relu & 0xff, relu >> 8, g_dev_opts[12]};
  hipLaunchKernelGGL(deconv3d_w16_kernel, dim3((unsigned)grid), dim3(C::NTHREADS), lds, st, a, scale, shift);
  const int rc = launch_status("deconv3d (sixteen-wave workgroups) launch failed");
  if (rc != DMB_OK) *tk.dirty = true;
  return rc;
}

}  // namespace dmb
