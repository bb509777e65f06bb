// Transposed 3-D convolution k3 s2 p1 op1 (hourglass conv5 / conv6: utils/hourglass.py:53-60,84-86 through
// layers/basic_layers.py:160-177), third form: one work item = (input-resolution tile, output z parity, output y parity).
//
//   y[2i - 1 + k] += x[i] * w[k]  per axis:  an even output o = 2i sees k = 1 from input i; an odd output o = 2i + 1 sees
//   k = 2 from input i and k = 0 from input i + 1.
//
// Why a third form.  deconv3d_kernel (conv3d.hip) gives a work item both y parities: 2 (y) x 2 (x) x 2 column tiles x 16 =
// 128 accumulator registers, so two workgroups per CU is all the register file admits -- and with two waves per SIMD the
// time both sit in their (vector-issue bound) epilogues at once is lost to the matrix cores (45 % /
// 44 % / 10.6 % of the time two / one / no wave of a SIMD multiplies).  Here an item owns ONE y parity: 2 (x) x 2 x 16 = 64
// accumulators, the kernel fits 168 registers and 46 KB of LDS, THREE workgroups share a CU, and the chance that every wave
// of a SIMD is outside its multiply phase drops from p^2 to p^3.
//
// The four (z parity, y parity) classes carry 1 : 2 : 2 : 4 of the arithmetic (number of (kz, ky) pairs an output of the
// class sees).  Each class is its own instantiation of the body -- its own input halo ((TZ + pz) planes x (1 + py) rows),
// its own channel chunk (16 / 8 / 8 / 4 input channels, so that every class runs 48 MFMAs per wave between two barriers and
// double-buffers within the same LDS budget).  Items are handed out through atomic counters: a persistent workgroup takes the
// next item while it multiplies the current one, and the chunk pipeline (LDS-DMA of chunk i + 1 under the MFMAs of chunk i)
// runs across consecutive items of a class.
//
// Item order (round 4).  The four classes of a tile stage (almost) the same input voxels.  In one class-major list over the
// whole layer (round 3) they ran a whole pass over the input apart: every input byte came from HBM four times (counters: 2.7 GB
// fetched for 1.0 GB of operands).  Now the tiles are split into eight contiguous ranges, one per XCD (a workgroup reads
// HW_REG_XCC_ID and draws from ITS range's counter first: speed only -- when that range is exhausted it draws from the others,
// so the result does not depend on where the hardware places workgroups), and inside a range the list goes group by group:
// RUN = 2^k consecutive tiles x the four classes, heaviest class first.  The four stagings of a tile then happen on one XCD within
// a few items of each other and three of them are served by that XCD's L2.
//
// The counters live in a caller-provided workspace (include/dmb_hip.h: DMB_DECONV3D_WORKSPACE_BYTES, zeroed once by the
// caller).  The last workgroup to leave resets them, so every launch finds and leaves zeros: nothing is predicted on the
// host, a launch captured in a HIP graph replays correctly, and the library allocates nothing.
//
// Everything else is the design of deconv3d_kernel: A = weights (rows = output channels), B = input voxels, both x
// parities in one wave so that a lane pair of accumulators is two adjacent outputs; 16-byte LDS-DMA staging; the epilogue
// interleaves the two x parities through a per-wave LDS scratch and leaves as 16-byte stores (BatchNorm affine, skip
// operand, ReLU in the reference's order).  Same FP32 products, same ascending (channel, tap) fma chain per output as the
// other two forms: bit-identical results.
#include <type_traits>

#include "dmb_common.h"

// Cache policy of the epilogue's stores / skip-operand loads (the aux operand of the buffer instructions on gfx950: 1 = sc0,
// 2 = nt, 16 = sc1; build-time knobs, build.py: DMB_BUILD_DEFS).  The output is 8x the input and is not read again by this
// launch: written `nt sc1` it does not stay in the XCD's L2, where the four classes of a tile find each other's input tiles.
// Measured (profiles/r04_zy_policy.log, min of 3 passes): quarter -> half resolution 0.265 -> 0.238 ms, half -> full 0.778 ->
// 0.767 ms, with the skip operand 0.856 either way; `nt` on the skip operand's loads changes nothing.
#ifndef DMB_ZY_ST
#define DMB_ZY_ST 18
#endif
#ifndef DMB_ZY_LD
#define DMB_ZY_LD 0
#endif

namespace dmb {

// MT_: 32-column MFMA tiles per input row of a work item: 2 (60 real positions + the halo column of the odd outputs in 64 staged
// columns) or 1 (28 real positions in 32) -- the narrower tile computes fewer discarded columns where the image width is
// awkward for 60 (80 columns: 3 x 32 computed instead of 2 x 64); deconv3d_zy_try picks per launch.
// FULL_ (round 6): all 32 MT positions of a staged row are real -- the row is staged 4 columns longer (pitch 32 MT + 4) instead of the
// tile being 4 positions narrower than what the matrix cores compute.  For widths that are multiples of 32 (the training crops: 64
// and 32 input columns) nothing computed is discarded, where the 28- / 60-column tiles compute 96 for 64 and 64 for 32.
template <int COUT_, int MT_ = 2, bool FULL_ = false>
struct ZYCfg {
  static constexpr int COUT = COUT_;
  static constexpr int NTT = COUT / 32;       // 32-channel row tiles
  static constexpr int WN = NTT;              // waves along the output channels: one row tile per wave
  static constexpr int WZ = 4 / WN, TZ = WZ;  // waves (= input planes) along z
  static constexpr int MT = MT_, P = FULL_ ? 32 * MT_ + 4 : 32 * MT_, TX = FULL_ ? 32 * MT_ : P - 4;   // one input row: P staged columns, TX real positions, MT 32-column MFMA tiles
  static constexpr int RUN = 3 * NTT * 64;    // weight floats of one (channel pair, kz, ky): its three kx taps
  static constexpr int SCR_PITCH = 68, PCH = 8;   // epilogue scratch: 8 channels x 64 output columns per pass and wave
  static constexpr int AFF_FLOATS = 2 * COUT;     // scale / shift table

  template <int PZ, int PY>
  struct Cls {
    static constexpr int NAZ = 1 + PZ, NAY = 1 + PY, NA = NAZ * NAY;   // (kz, ky) pairs an output of the class sees
    static constexpr int CK = 16 / NA;                                  // input channels per chunk
    static constexpr int ZS = TZ + PZ, ROWS = 1 + PY, PLANE = ROWS * P;
    static constexpr int CH_STRIDE = ZS * PLANE + 4;                    // = 4 (mod 32): the lane halves hit disjoint banks
    static constexpr int IN_FLOATS = CK * CH_STRIDE;
    static constexpr int W_FLOATS = (CK / 2) * NA * RUN;
    static constexpr int BUF_FLOATS = IN_FLOATS + W_FLOATS;
    static constexpr int CPW = CK / 4;                                  // channels staged by one wave
    static constexpr int UPC = ZS * ROWS * (P / 4);                     // 16-byte units per channel
    static constexpr int IPC = (UPC + 63) / 64;                         // copy instructions per channel
    static constexpr int WCH4 = W_FLOATS / 4, WV4 = (WCH4 + 255) / 256; // weight copies per chunk / per thread
    static constexpr int NU = (CK / 2) * NA;                            // (channel pair, kz, ky) units per chunk
    static constexpr int NPIECE = CPW * IPC + WV4;                      // copy instructions of one wave per chunk
    // the epilogue scratch of a wave lies in the part of the consumed chunk buffer only this wave's own copies write
    static constexpr bool SCR_PRIVATE = CPW * CH_STRIDE >= PCH * SCR_PITCH;
    static_assert(CK % 4 == 0 && IN_FLOATS % 4 == 0 && W_FLOATS % 4 == 0, "shape");
  };
  static constexpr int cmax(int a, int b) { return a > b ? a : b; }
  static constexpr int BUF_MAX = cmax(cmax(Cls<0, 0>::BUF_FLOATS, Cls<0, 1>::BUF_FLOATS),
                                      cmax(Cls<1, 0>::BUF_FLOATS, Cls<1, 1>::BUF_FLOATS));
  static constexpr bool ALL_PRIVATE = Cls<0, 0>::SCR_PRIVATE && Cls<0, 1>::SCR_PRIVATE && Cls<1, 0>::SCR_PRIVATE && Cls<1, 1>::SCR_PRIVATE;
  static constexpr int SCR_FLOATS = ALL_PRIVATE ? 0 : 4 * PCH * SCR_PITCH;
  static constexpr int LDS_FLOATS = 2 * BUF_MAX + AFF_FLOATS + SCR_FLOATS;
  static_assert(LDS_FLOATS * 4 * 3 <= 160 * 1024, "three workgroups per CU");
};

constexpr int ZY_RUN_LOG2 = 1;   // tiles per group of the item order = 2 (DMB_OPT(16) overrides the log2 in the development build)
constexpr int ZY_PARTS = 8;      // tile ranges = item counters (one per XCD)
constexpr int ZY_STRIDE = 32;    // ints between two counters: each on its own 128-byte line (eight XCDs hammer eight lines, not one)
// workspace (ints): counter k at [k * ZY_STRIDE], workgroups-done counter at [ZY_PARTS * ZY_STRIDE]
static_assert((ZY_PARTS * ZY_STRIDE + 1) * 4 <= DMB_DECONV3D_WORKSPACE_BYTES, "workspace layout");

struct ZYArgs {
  const float* x;
  const float* wp;
  const float* res;
  float* y;
  int* ws;       // the caller's workspace: zeros on entry, zeros again when the last workgroup has left
  int Ci, D, H, W, ntx, nty, ntz, ntiles, relu, dbg;
  int Wout;      // output row length: 2 W, or less when the input's last columns are zero padding (multiple of 4)
  int lrun;      // log2 of the tiles per group of the item order
  int nparts;    // tile ranges in use (ZY_PARTS; 1 in the development build's single-list mode)
  int q, r;      // ntiles = q * nparts + r: range k holds q + (k < r) tiles from k q + min(k, r)
  int stagger;   // development build: start-up delay unit (see the kernel)
};

// ---- drawing work items ---------------------------------------------------------------------------------------------------
// Range k's counter value t names group t >> (lrun + 2) of RUN = 1 << lrun tiles and, inside the group, class-major (heaviest
// class first): class (t mod 4 RUN) / (tiles in the group), tile (t mod 4 RUN) mod (tiles in the group).
// A draw is split in three so that NOTHING waits on the atomic's round trip: thread 0 ISSUES the fetch-add before the item's
// first multiply loop (zy_issue), turns the returned ticket into a published (range, ticket) pair after that loop (zy_finish;
// only when its range is exhausted -- the tail of the launch -- does it draw from the other ranges there, synchronously), and
// every wave DECODES the pair into (class, tile) after the barrier, in scalar arithmetic (zy_decode).
__device__ __forceinline__ int zy_range(const ZYArgs& a, int part, int skip) {
  const int k = part + skip;
  return k >= a.nparts ? k - a.nparts : k;
}
__device__ __forceinline__ unsigned zy_issue(const ZYArgs& a, int part, int skip) {
  if (skip >= a.nparts) return 0xffffffffu;
  return (unsigned)__hip_atomic_fetch_add(a.ws + zy_range(a, part, skip) * ZY_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// -> (range << 27) | ticket, or -1 when every range is exhausted (tickets stay below 2^27: checked on the host)
__device__ __forceinline__ int zy_finish(const ZYArgs& a, int part, int& skip, unsigned t) {
  while (skip < a.nparts) {
    const int k = zy_range(a, part, skip);
    const unsigned nk = (unsigned)(a.q + (k < a.r ? 1 : 0));
    if (t < 4u * nk) return (k << 27) | (int)t;
    ++skip;   // this range is exhausted for good (its counter only grows)
    t = zy_issue(a, part, skip);
  }
  return -1;
}
// -> item code cls * ntiles + tile (4 * ntiles = no item)
__device__ __forceinline__ int zy_decode(const ZYArgs& a, int code) {
  if (code < 0) return 4 * a.ntiles;
  const int k = code >> 27, t = code & ((1 << 27) - 1);
  const int nk = a.q + (k < a.r ? 1 : 0), base = k * a.q + min(k, a.r);
  const int run = 1 << a.lrun, first = (t >> (a.lrun + 2)) << a.lrun, rr = t & (4 * run - 1);
  const int ng = min(run, nk - first);
  const int cls = ng == run ? rr >> a.lrun : rr / ng;   // (the division: last, partial group of a range only)
  return cls * a.ntiles + base + first + (rr - cls * ng);
}

// One class: processes `item` and every following item the counter hands out as long as it belongs to the same class;
// returns the first item that does not (>= 4 * ntiles: the list is exhausted).
template <class C, int PZ, int PY>
__device__ __forceinline__ int zy_body(float* lds, int* slot, const ZYArgs& a, int item, int part, int& skip) {
  using K = typename C::template Cls<PZ, PY>;
  // The four class bodies are inlined into one loop over items: without this barrier the compiler hoists every body's
  // per-lane constants (copy offsets, epilogue lane coordinates) to the kernel entry and keeps all four sets live -- 190
  // registers for a kernel whose largest body needs 120.  Everything per-lane below derives from this opaque copy.
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  const int wz = wave / C::WN, wn = wave % C::WN;
  const int D = a.D, H = a.H, W = a.W, Ci = a.Ci;
  const unsigned HW = (unsigned)H * W, DHW = (unsigned)D * HW;
  const int Ho = 2 * H, Wo = a.Wout;
  const unsigned HWo = (unsigned)Ho * Wo, DHWo = 2u * D * HWo;
  constexpr int CLS = 3 - (2 * PZ + PY);   // class number of the item code cls * ntiles + tile: heaviest first
  const int lo_item = CLS * a.ntiles, hi = (CLS + 1) * a.ntiles;
  float* aff = lds + 2 * C::BUF_MAX;

  struct Tile {
    int b, x0, y0, z0;
  };
  auto tile_at = [&](int it) {
    int t = it - CLS * a.ntiles;
    Tile tl;
    tl.x0 = (t % a.ntx) * C::TX;
    t /= a.ntx;
    tl.y0 = t % a.nty;
    t /= a.nty;
    tl.z0 = (t % a.ntz) * C::TZ;
    tl.b = t / a.ntz;
    return tl;
  };

  // ---- weight copies: per-lane source offset inside a chunk's block (the chunk goes into the scalar offset); LDS layout
  // [channel pair][az][ay][kx][row tile][64]: az = 0 -> kz = (PZ ? 2 : 1), az = 1 -> kz = 0; ay likewise
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.wp, (unsigned)(Ci * 27 * C::COUT) * 4u);
  unsigned woff[K::WV4];
#pragma unroll
  for (int i = 0; i < K::WV4; ++i) {
    const int q4 = i * 256 + tid;
    const int run = q4 / (C::RUN / 4), off = q4 - run * (C::RUN / 4);
    const int cp = run / K::NA, ta = run - cp * K::NA, az = ta / K::NAY, ay = ta - az * K::NAY;
    const int kz = PZ ? (az ? 0 : 2) : 1, ky = PY ? (ay ? 0 : 2) : 1;
    woff[i] = q4 < K::WCH4 ? (unsigned)(((cp * 27 + kz * 9 + ky * 3) * C::NTT * 64) * 4 + off * 16) : DMA_OOB;
  }
  // ---- input copies: unit = 4 consecutive floats of a staged row, the units of one channel ([plane][row][16 units]) are
  // linear in LDS; a unit's per-lane source offset depends on the tile only (the channel is added at issue time)
  auto tile_offsets = [&](const Tile& tl, unsigned (&o)[K::IPC]) {
#pragma unroll
    for (int q = 0; q < K::IPC; ++q) {
      const int u = q * 64 + lane;
      constexpr int UPR = C::P / 4;   // 16-byte units per staged row
      const int zz = u / (K::ROWS * UPR), rr = u - zz * (K::ROWS * UPR), yy = rr / UPR, sg = rr - yy * UPR;
      const int gz = tl.z0 + zz, gy = tl.y0 + yy, gx = tl.x0 + sg * 4;
      o[q] = (u < K::UPC && gz < D && gy < H && gx < W) ? ((unsigned)gz * HW + (unsigned)gy * W + (unsigned)gx) * 4u : DMA_OOB;
    }
  };
  // a wave's copies of one chunk are numbered [0, NPIECE): input pieces first, then its share of the weights; stage()
  // issues pieces [lo, hi) so that the multiply loop can deal them out between its MFMA groups
  auto in_rsrc = [&](const Tile& tl, int c0) {
    return make_rsrc(a.x + ((size_t)tl.b * Ci + c0) * DHW, (unsigned)K::CK * DHW * 4u);
  };
  auto stage = [&](const __amdgpu_buffer_rsrc_t xrs, const unsigned (&toff)[K::IPC], int c0, float* buf, int lo, int hi2) {
#pragma unroll
    for (int cc = 0; cc < K::CPW; ++cc) {
      const int cl = wave * K::CPW + cc;
#pragma unroll
      for (int q = 0; q < K::IPC; ++q)
        if (cc * K::IPC + q >= lo && cc * K::IPC + q < hi2 && (K::UPC % 64 == 0 || q * 64 + lane < K::UPC))
          dma16(xrs, toff[q], (unsigned)cl * DHW * 4u, buf + cl * K::CH_STRIDE + q * 256);   // (whole chunks only: Ci % 16 == 0)
    }
#pragma unroll
    for (int i = 0; i < K::WV4; ++i)
      if (K::CPW * K::IPC + i >= lo && K::CPW * K::IPC + i < hi2 && (K::WCH4 % 256 == 0 || i * 256 + tid < K::WCH4))
        dma16(wrs, woff[i], (unsigned)(c0 / 2) * (27 * C::NTT * 64 * 4), buf + K::IN_FLOATS + (i * 256 + wave * 64) * 4);
  };

  const int NC = DMB_DBG(a.dbg & 64) ? 2 : Ci / K::CK;   // (development: two chunks per item -- the memory side of an item almost alone)
  Tile cur_t = tile_at(item);
  unsigned coff[K::IPC], noff[K::IPC];
  tile_offsets(cur_t, coff);
  __syncthreads();   // a class switch: every wave has left the previous class's buffers (and its epilogue scratch)
  stage(in_rsrc(cur_t, 0), coff, 0, lds, 0, K::NPIECE);
  __syncthreads();
  int g = 0;          // chunks consumed so far: selects the LDS buffer
  for (;;) {
    int next = 4 * a.ntiles;    // published after the first chunk's barrier
    bool has_next = false;
    Tile next_t = cur_t;
    unsigned ticket = 0;

    f32x16 acc[2][C::MT];   // [x parity][column tile]; started by the item's first unit (constant 0 operand, see below)

#pragma clang loop unroll(disable)   // (and no peeling of the first chunk: one copy of the multiply loop per class)
    for (int ci = 0; ci < NC; ++ci, ++g) {
      const float* cur = lds + (g & 1) * C::BUF_MAX;
      float* nxt = lds + ((g + 1) & 1) * C::BUF_MAX;
      if (ci == 0 && tid == 0) ticket = zy_issue(a, part, skip);   // in flight during this chunk's multiply loop
      // the next chunk's copies (of this item, or the first chunk of the next one) are dealt out over the first SU units
      constexpr int SU = K::NU - 2, PPU = (K::NPIECE + SU - 1) / SU;
      const bool more = ci + 1 < NC;
      const bool staging = !DMB_DBG(a.dbg & 2) && (more || has_next);
      const int st_c0 = more ? (ci + 1) * K::CK : 0;
      const __amdgpu_buffer_rsrc_t st_rs = in_rsrc(more ? cur_t : next_t, st_c0);
      unsigned st_off[K::IPC];
#pragma unroll
      for (int q = 0; q < K::IPC; ++q) st_off[q] = more ? coff[q] : noff[q];
      auto deal = [&](int u) {
        if (staging && !DMB_DBG(a.dbg & 32)) stage(st_rs, st_off, st_c0, nxt, u * PPU, (u + 1) * PPU);
        if (staging && DMB_DBG(a.dbg & 32) && u == 0) stage(st_rs, st_off, st_c0, nxt, 0, K::NPIECE);
      };
      const float* abase = cur + K::IN_FLOATS + wn * 64 + lane;
      const float* bbase = cur + h * K::CH_STRIDE + wz * K::PLANE + j;
      float af[2][3], bf[2][2][C::MT];   // af[buf][kx], bf[buf][ox][mt]
      auto load_frag = [&](int u, float (&fa)[3], float (&fb)[2][C::MT]) {
        const int cp = u / K::NA, ta = u % K::NA, az = ta / K::NAY, ay = ta % K::NAY;
        // (volatile: one ds_read_b32 per fragment with an immediate offset from ONE base register.  Left to itself the compiler
        // pairs the reads into ds_read2_b32, whose 8-bit offsets need a fresh base -- a v_add -- per pair: 16 vector instructions
        // per 48 MFMAs in a loop where vector time adds to matrix time)
#ifdef DMB_ZY_PLAIN_READS   // (build-time A/B knob: the compiler's paired reads)
        typedef const __attribute__((address_space(3))) float* lds_cvf;
#else
        typedef const volatile __attribute__((address_space(3))) float* lds_cvf;   // (LDS address space, kept explicit under volatile)
#endif
        lds_cvf av = (lds_cvf)abase;
#pragma unroll
        for (int k = 0; k < 3; ++k) fa[k] = av[(u * 3 + k) * C::NTT * 64];
        lds_cvf bp = (lds_cvf)(bbase + 2 * cp * K::CH_STRIDE + az * K::PLANE + ay * C::P);
#pragma unroll
        for (int ox = 0; ox < 2; ++ox)
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) fb[ox][mt] = bp[ox + mt * 32];
      };
      load_frag(0, af[0], bf[0]);
#pragma unroll
      for (int u = 0; u < K::NU; ++u) {
        if (u + 1 < K::NU) load_frag(u + 1, af[(u + 1) & 1], bf[(u + 1) & 1]);
        if (u < SU) deal(u);
        __builtin_amdgcn_sched_barrier(0);
        const auto& fa = af[u & 1];
        const auto& fb = bf[u & 1];
        if (u == 0 && ci == 0) {
          // the item's first products start the accumulators from the MFMA's constant 0 operand: no 64 v_mov per item to clear
          // them (a wave-uniform branch around the first unit only; the loop itself stays one copy per class)
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) {
            acc[0][mt] = DMB_MFMA(fa[1], fb[0][mt], zero);
            acc[1][mt] = DMB_MFMA(fa[2], fb[0][mt], zero);
            acc[1][mt] = DMB_MFMA(fa[0], fb[1][mt], acc[1][mt]);
          }
        } else {
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) {
            acc[0][mt] = DMB_MFMA(fa[1], fb[0][mt], acc[0][mt]);   // even x: kx = 1 from input x
            acc[1][mt] = DMB_MFMA(fa[2], fb[0][mt], acc[1][mt]);   // odd x:  kx = 2 from input x
            acc[1][mt] = DMB_MFMA(fa[0], fb[1][mt], acc[1][mt]);   //         kx = 0 from input x + 1
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ci == 0 && tid == 0) __atomic_store_n(slot, zy_finish(a, part, skip, ticket), __ATOMIC_RELAXED);
      __syncthreads();
      if (ci == 0) {
        // wave-uniform: tiles, resources and branches derived from it stay scalar
        next = zy_decode(a, __builtin_amdgcn_readfirstlane(__atomic_load_n(slot, __ATOMIC_RELAXED)));
        has_next = next >= lo_item && next < hi && !DMB_DBG(a.dbg & 16);   // the next item continues this class: pipeline across it
        if (has_next) {
          next_t = tile_at(next);
          tile_offsets(next_t, noff);
        }
      }
    }

    // ---- epilogue of cur_t: per (column tile, 8-channel group) the two x-parity accumulator tiles are interleaved in the
    // wave's LDS scratch ([8 channels][64 output columns], 8-byte writes), read back as 4 consecutive x of one channel and
    // stored / residual-loaded as 16-byte words through buffer resources (lanes outside the volume get an out-of-range
    // offset: branch-free).  Residual loads run RD passes ahead of the stores.
    // (g has been advanced past the last chunk: buffer (g & 1) is being filled for the next item, ((g - 1) & 1) is the one
    // just consumed; every wave's reads of it completed before the barrier that ended the chunk loop)
    if (!DMB_DBG(a.dbg & 1)) {
      float* scr = K::SCR_PRIVATE ? lds + ((g - 1) & 1) * C::BUF_MAX + wave * (K::CPW * K::CH_STRIDE)
                                  : aff + C::AFF_FLOATS + wave * (C::PCH * C::SCR_PITCH);
      const int gzi = cur_t.z0 + wz;
      float* yb = a.y + (size_t)cur_t.b * C::COUT * DHWo;
      const __amdgpu_buffer_rsrc_t yrs = make_rsrc(yb, (unsigned)C::COUT * DHWo * 4u);
      const __amdgpu_buffer_rsrc_t rrs = make_rsrc(a.res ? a.res + (size_t)cur_t.b * C::COUT * DHWo : yb, (unsigned)C::COUT * DHWo * 4u);
      const int rl = lane >> 4, x4 = (lane & 15) * 4;
      const float lo = a.relu == 1 ? 0.f : -__builtin_inff();    // ReLU after the residual add
      const bool pre_relu = a.relu == 2;                          // ReLU before it (GC-Net)
      constexpr int QP = 32 / C::PCH, KP = C::PCH / 4;          // passes per 32-channel tile, 16-byte words per lane and pass
      constexpr int NPASS = C::MT * QP;
      // Addressing: a word's byte offset = lane part (channel row rl of the pass, 4 consecutive x: ONE register per column
      // tile, out of range for lanes outside the volume) + a wave-uniform part (item, pass, word) that travels in the scalar
      // offset of the buffer instruction.  Per-lane offsets for every (pass, word) would be 16 item-invariant registers the
      // compiler keeps live across the multiply loop -- the difference between 168 registers and spilling.
      unsigned voff[C::MT];
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
        const int lx = mt * 32 + x4 / 2;
        // (a lane's word = output columns 2 (x0 + lx) .. + 3; Wo % 4 == 0: inside or outside as a whole)
        const bool ok = gzi < D && lx < C::TX && 2 * (cur_t.x0 + lx) < Wo && !DMB_DBG(a.dbg & 4);
        voff[mt] = ok ? (DMB_DBG(a.dbg & 8) ? (unsigned)lane * 16u : ((unsigned)rl * DHWo + (unsigned)(2 * mt * 32 + x4)) * 4u) : DMA_OOB;
      }
      const unsigned sbase = ((unsigned)(wn * 32) * DHWo + (unsigned)(2 * gzi + PZ) * HWo + (unsigned)(2 * cur_t.y0 + PY) * Wo +
                              2u * (unsigned)cur_t.x0) * 4u;
      const unsigned sstep = 4u * DHWo * 4u;                      // four channels on
      auto soff = [&](int t, int k) {
        const unsigned o = sbase + (unsigned)((t % QP) * KP + k) * sstep;
        return DMB_DBG(a.dbg & 8) ? (o & 0x1fff80u) : o;   // development: epilogue traffic confined to a 2 MiB window
      };
      const float* affl = aff + wn * 32 + rl;                     // + 8 q + 4 k: immediate offsets
      float* swr = scr + 4 * h * C::SCR_PITCH + 2 * j;           // + rr * pitch
      const unsigned swr_addr = (unsigned)(uintptr_t)swr;          // LDS byte address (the low 32 bits of the generic pointer)
      const float* srd = scr + rl * C::SCR_PITCH + x4;           // + 4 k * pitch
      auto run = [&](auto has_res) {
        constexpr bool HAS_RES = decltype(has_res)::value;
        constexpr int RD = HAS_RES ? 4 : 1;
        u32x4 rv[RD][KP];
        if constexpr (HAS_RES) {
#pragma unroll
          for (int t = 0; t < RD; ++t)
#pragma unroll
            for (int k = 0; k < KP; ++k) rv[t][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)voff[t / QP], (int)soff(t, k), DMB_ZY_LD);
        }
#pragma unroll
        for (int t = 0; t < NPASS; ++t) {
          const int q = t % QP, mt = t / QP;
          const int sl = t % RD;
#pragma unroll
          for (int rr = 0; rr < C::PCH / 2; ++rr) {
            const int r = q * (C::PCH / 2) + rr;   // accumulator register r of lane half h = channel 8 q + rr + 4 h of the tile
            // ds_write2_b32 by hand: its two data registers need not be adjacent.  Written as an 8-byte (or two adjacent dword)
            // stores the compiler first copies the pair into adjacent registers -- 128 v_mov per work item, and vector
            // instructions cost this kernel a tenth of its matrix time (profiles/r03_pmc_hg.log: SQ_INSTS_VALU counts the MFMAs too).  LDS operations of a wave
            // complete in order, so the float4 reads below see these writes; "memory" keeps the compiler from moving them.
            static_assert((C::PCH / 2 - 1) * C::SCR_PITCH + 1 <= 255, "ds_write2_b32 offsets are 8 bits of dwords");
            asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4"
                         :
                         : "v"(swr_addr), "v"(acc[0][mt][r]), "v"(acc[1][mt][r]), "n"(rr * C::SCR_PITCH), "n"(rr * C::SCR_PITCH + 1)
                         : "memory");
          }
#pragma unroll
          for (int k = 0; k < KP; ++k) {
            float4 v = *reinterpret_cast<const float4*>(srd + 4 * k * C::SCR_PITCH);
            const float sc = affl[q * C::PCH + 4 * k], sh = affl[C::COUT + q * C::PCH + 4 * k];
            v.x = fmaf(v.x, sc, sh);
            v.y = fmaf(v.y, sc, sh);
            v.z = fmaf(v.z, sc, sh);
            v.w = fmaf(v.w, sc, sh);
            if (pre_relu) {   // wave-uniform (GC-Net's ReLU before the skip add): a scalar branch instead of 64 v_max per work item
              v.x = fmaxf(v.x, 0.f);
              v.y = fmaxf(v.y, 0.f);
              v.z = fmaxf(v.z, 0.f);
              v.w = fmaxf(v.w, 0.f);
            }
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
            // The uniform part goes into the VECTOR offset for stores (one v_add each), not into the scalar-offset operand:
            // with an SGPR soffset the compiler's hazard model lets the next VALU instruction overwrite the store's data
            // registers at once, and on this chip a 16-byte store still reads the data of its last lanes then -- under load
            // (three workgroups per CU) lanes 12..15 of each 16 stored the NEXT pass's accumulator values (found with
            // scripts/attic/zy_debug.py: one wrong float in 10^3, only with more than one workgroup per CU).  With a literal
            // soffset the compiler inserts the wait states itself.
            __builtin_amdgcn_raw_buffer_store_b128(o, yrs, (int)(voff[mt] + soff(t, k)), 0, DMB_ZY_ST);
          }
          if constexpr (HAS_RES) {
            if (t + RD < NPASS) {   // refill the slot just consumed
#pragma unroll
              for (int k = 0; k < KP; ++k)
                rv[sl][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)voff[(t + RD) / QP], (int)soff(t + RD, k), DMB_ZY_LD);
            }
          }
        }
      };
      if (a.res)
        run(std::true_type{});
      else
        run(std::false_type{});
    }
    if (!has_next) return next;
    cur_t = next_t;
#pragma unroll
    for (int q = 0; q < K::IPC; ++q) coff[q] = noff[q];
  }
}

template <class C>
__global__ __launch_bounds__(256, 3) void deconv3d_zy_kernel(ZYArgs a, const float* __restrict__ scale,
                                                             const float* __restrict__ shift) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* aff = lds + 2 * C::BUF_MAX;
  __shared__ int slot[1];   // the next item, published by thread 0
#ifdef DMB_DEV
  // (development option 12) a start-up delay of (workgroup index mod 16) x stagger x 3.4 us: measured, no gain
  if (a.stagger > 0) {
    const int n = (int)((blockIdx.x * 7u) & 15u) * a.stagger;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  // per-channel affine, staged once (LDS reads count on lgkmcnt and cost no registers across the item loop; see
  // deconv3d_kernel)
  if (threadIdx.x < C::COUT) {
    aff[threadIdx.x] = scale ? scale[threadIdx.x] : 1.f;
    aff[C::COUT + threadIdx.x] = shift ? shift[threadIdx.x] : 0.f;
  }
  // the tile range this workgroup draws from first: its XCD's (HW_REG_XCC_ID, bits 3:0 of hardware register 20) -- locality
  // only; zy_finish moves on to the other ranges when this one is exhausted, so any placement computes every item exactly once
  const int part = a.nparts > 1 ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u) % a.nparts : 0;
  int skip = 0;   // (state of thread 0, the only caller of zy_issue / zy_finish)
  if (threadIdx.x == 0) __atomic_store_n(slot, zy_finish(a, part, skip, zy_issue(a, part, skip)), __ATOMIC_RELAXED);
  __syncthreads();
  int item = zy_decode(a, __builtin_amdgcn_readfirstlane(__atomic_load_n(slot, __ATOMIC_RELAXED)));
  const int total = 4 * a.ntiles;
  while (item >= 0 && item < total) {
    const int cls = item / a.ntiles;   // 0 = (odd z, odd y): four (kz, ky) pairs ... 3 = (even z, even y): one
    if (cls == 0)
      item = zy_body<C, 1, 1>(lds, slot, a, item, part, skip);
    else if (cls == 1)
      item = zy_body<C, 1, 0>(lds, slot, a, item, part, skip);
    else if (cls == 2)
      item = zy_body<C, 0, 1>(lds, slot, a, item, part, skip);
    else
      item = zy_body<C, 0, 0>(lds, slot, a, item, part, skip);
  }
  // Leave the workspace as it was found.  A workgroup draws nothing after this point; the last one to arrive knows that every
  // other one has drawn its last ticket (release / acquire through the `done` counter) and zeroes counters and `done`.
  if (threadIdx.x == 0) {
    int* done = a.ws + ZY_PARTS * ZY_STRIDE;
    if (__hip_atomic_fetch_add(done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
      for (int k = 0; k < ZY_PARTS; ++k) __hip_atomic_store(a.ws + k * ZY_STRIDE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(done, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <class C>
static int launch_zy(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                     int B, int Ci, int D, int H, int W, int Wout, int relu, int* ws, hipStream_t st) {
  const int ntx = cdiv(W, C::TX), nty = H, ntz = cdiv(D, C::TZ);
  const long long ntiles = (long long)B * ntx * nty * ntz;
  if (4 * ntiles > 0x3fffffffLL) return fail(DMB_EUNSUPPORTED, "deconv3d: grid too large");
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  DMB_ENSURE_LDS((&deconv3d_zy_kernel<C>), lds);
  const long long slots = 3LL * num_cus() * (DMB_OPT(8) > 0 ? DMB_OPT(8) : 1);
  long long grid = 4 * ntiles < slots ? 4 * ntiles : slots;
  if (DMB_OPT(9) > 0 && DMB_OPT(9) < grid) grid = DMB_OPT(9);   // development: few workgroups walk many items
  if (4 * ntiles >= (1LL << 27)) return -1;   // (ticket field of the published draw; 33 M tiles: far beyond any volume here)
  const bool sel = DMB_OPT(21) == 0 || DMB_OPT(21) == C::COUT;   // (development option 21: options 16 / 20 for one width only)
  int lrun = sel && DMB_OPT(16) > 0 ? DMB_OPT(16) - 1 : ZY_RUN_LOG2;   // (development option 16 = log2(tiles per group) + 1)
  int nparts = ZY_PARTS;
  if (sel && DMB_OPT(20)) {   // development: ONE class-major list over the whole layer (the round-3 order)
    nparts = 1;
    lrun = 0;
    while ((1LL << lrun) < ntiles) ++lrun;
  }
  ZYArgs a{x, wp, res, y, ws, Ci, D, H, W, ntx, nty, ntz, (int)ntiles, relu & 0xff, relu >> 8, Wout, lrun, nparts,
           (int)(ntiles / nparts), (int)(ntiles % nparts), DMB_OPT(12)};
  hipLaunchKernelGGL((deconv3d_zy_kernel<C>), dim3((unsigned)grid), dim3(256), lds, st, a, scale, shift);
  return launch_status("deconv3d (z/y-parity items) launch failed");
}

// Entry for dmb_deconv3d_k3s2_f32 (conv3d.hip): returns -1 when this form does not apply (the caller falls back to
// deconv3d_kernel), otherwise the launch status.  Requirements: 16-byte aligned rows (W % 4 == 0, aligned bases), all 32 or
// 64 output channels real, whole 16-channel chunks and at least two of them, 16 input channels of one batch item and one batch item of the output
// below 2 GiB (32-bit buffer offsets).
int deconv3d_zy_try(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                    int B, int Ci, int Co, int D, int H, int W, int Wout, int relu, int* ws, hipStream_t st) {
  if (DMB_OPT(4) == 1 || !ws) return -1;   // (development option 4 = 1: deconv3d_kernel)
  if ((((uintptr_t)ws) & 3) != 0) return fail(DMB_EINVAL, "deconv3d: misaligned workspace");
  if (!(Co == 32 || Co == 64) || Ci % 16 != 0 || Ci < 32 || W % 4 != 0 || Wout % 4 != 0) return -1;   // (>= 2 chunks in every class)
  if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) != 0) return -1;
  if ((long long)16 * D * H * W * 4 >= 0x7fffffffLL || (long long)Co * 8 * D * H * W * 4 >= 0x7fffffffLL) return -1;
  // 28-column tiles where they compute fewer columns than 60-column ones (80 input columns: 96 against 128); (round 6) tiles of 32 /
  // 64 REAL columns (a longer staged row) where those compute fewer still: 64 input columns 64 against 96, 32 columns 32 against
  // 64, 156 (KITTI) 160 against 192.  Same arithmetic per output whatever the tile.
  const bool narrow = cdiv(W, 28) * 32 < cdiv(W, 60) * 64;
  const int c_old = narrow ? cdiv(W, 28) * 32 : cdiv(W, 60) * 64, c_f32 = cdiv(W, 32) * 32, c_f64 = cdiv(W, 64) * 64;
  // (a tie between 60- and 64-column tiles -- 120 input columns: 2 x 64 computed either way -- goes to the 64-column form, whose tiles
  // start on 512-byte boundaries of the output rows: 849-865 -> 829 us for conv6 at [4, 64, 24, 68, 120] alone
  // (scripts/attic/zy_b1_tile_probe.py), equal within the noise inside the step: 807 against 803 us under rocprofv3)
  int full = c_f64 <= c_old && c_f64 <= c_f32 ? 2 : (c_f32 < c_old ? 1 : 0);
  if (DMB_OPT(29) == 1) full = 0;                      // (development option 29: 1 = never, 2 / 3 = force the 32- / 64-column form)
  if (DMB_OPT(29) >= 2) full = DMB_OPT(29) - 1;
  // (round 6) A 64-channel launch of about one item per workgroup slot (conv5 of one 544x960 pair: 816 items of weight 1 .. 4 on
  // 768 slots) is faster on deconv3d_kernel's items with both y parities: 0.099 -> 0.069 ms at [1, 64, 12, 34, 60]; equal at the
  // KITTI shape and from two pairs on (scripts/kbench_hg.py, KB_B = 1 / 2).  Same arithmetic, bit-identical results.
  if (Co == 64 && Wout == 2 * W && !DMB_OPT(28)) {   // (development option 28: keep the work-queue form)
    const long long items = 4LL * B * cdiv(W, narrow ? 28 : 60) * H * cdiv(D, 2);
    if (4 * items <= 5LL * 3 * num_cus()) return -1;
  }
  if (Co == 32) {
    if (full == 2) return launch_zy<ZYCfg<32, 2, true>>(x, wp, scale, shift, res, y, B, Ci, D, H, W, Wout, relu, ws, st);
    if (full == 1) return launch_zy<ZYCfg<32, 1, true>>(x, wp, scale, shift, res, y, B, Ci, D, H, W, Wout, relu, ws, st);
    return narrow ? launch_zy<ZYCfg<32, 1>>(x, wp, scale, shift, res, y, B, Ci, D, H, W, Wout, relu, ws, st)
                  : launch_zy<ZYCfg<32, 2>>(x, wp, scale, shift, res, y, B, Ci, D, H, W, Wout, relu, ws, st);
  }
  if (full == 2) return launch_zy<ZYCfg<64, 2, true>>(x, wp, scale, shift, res, y, B, Ci, D, H, W, Wout, relu, ws, st);
  if (full == 1) return launch_zy<ZYCfg<64, 1, true>>(x, wp, scale, shift, res, y, B, Ci, D, H, W, Wout, relu, ws, st);
  return narrow ? launch_zy<ZYCfg<64, 1>>(x, wp, scale, shift, res, y, B, Ci, D, H, W, Wout, relu, ws, st)
                : launch_zy<ZYCfg<64, 2>>(x, wp, scale, shift, res, y, B, Ci, D, H, W, Wout, relu, ws, st);
}

}  // namespace dmb
