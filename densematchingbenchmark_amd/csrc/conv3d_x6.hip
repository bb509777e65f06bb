// EXPERIMENTAL, OPT-IN: 3x3x3 stride-1 convolution (32 or 64 output channels) with FP32 operands split exactly into three
// bf16 pieces and the six largest cross products issued on v_mfma_f32_32x32x16_bf16 (FP32 accumulate).
//
//   x = x0 + x1 + x2,  w = w0 + w1 + w2  (each piece a bf16, the sum exact: 3 x 8 = 24 significand bits)
//   x * w ~= x0 w0 + x0 w1 + x1 w0 + x0 w2 + x2 w0 + x1 w1          (dropped terms <= 2^-24 |x w|)
//
// Every bf16 x bf16 product is exact in FP32 and a k-step accumulates 16 of them at once, so the result is at least as
// close to the real-number convolution as the FP32 fma chain of conv3d_s1_kernel (measured in tests against FP64; K =
// 864: max error 2.0e-6 vs 3.8e-6) -- it is NOT bit-identical to it, which is why this path is never taken silently:
// the caller must ask for it (densematchingbenchmark_amd.ops.set_conv3d_mode("bf16x6")).
// Cost per 16 input channels x tap x (32 voxels x 32 channels): 6 x 32 = 192 matrix-core cycles against 8 x 64 = 512.
//
// Layout: same workgroup tile as the exact kernel (4 z-slices x 4 rows x 48 columns, row-pair B tiles, one z-slice per
// wave).  K is consumed in chunks of 4 input channels; one MFMA k-step = 4 taps x 4 channels (lanes 0-31: taps 4s,
// 4s+1, lanes 32-63: taps 4s+2, 4s+3; 27 taps padded to 28 with zero weights).  The tile is loaded as FP32 with
// 16-byte buffer loads one chunk ahead (registers), split into bf16 triples by the VALU (v_cvt_pk_bf16_f32 + exact
// residuals) and stored to LDS as [piece][voxel][4 channels] (8 bytes per voxel): a B fragment is two ds_read_b64.
// Weights are pre-split on the device once per load_state_dict into fragment order and streamed by LDS-DMA.
// LDS is single-buffered (48 KB activations + 21 KB weights): two workgroups per CU alternate split and multiply.
#include <type_traits>

#include "dmb_common.h"

namespace dmb {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// NT_ 32-channel row tiles per wave (output channels = 32 NT_); G_ = 16: B tiles are row pairs (rows r, r + 2) of 16 columns,
// G_ = 8: row quads (4 consecutive rows) of 8 columns; P_ = LDS row pitch in voxels, chosen so that the rows of one tile
// start 32 (pairs) or 16 (quads) banks apart: a ds_read_b64 lane group then covers all 64 banks exactly once.
template <int NT_, int G_, int TX_, int P_>
struct X6Cfg {
  static constexpr int NT = NT_, G = G_, TX = TX_, P = P_;
  static constexpr int TY = 4, TZ = 4, CK = 4, NSTEP = 7;
  static constexpr int ROWS = TY + 2, ZS = TZ + 2, PLANE = ROWS * P, NVOX = ZS * PLANE;
  static constexpr int XOFF = 3;                           // the staged row starts at the aligned column x0 - 4
  static constexpr int PIECE_BYTES = NVOX * 8;             // [voxel][4 channels] bf16
  static constexpr int ACT_BYTES = 3 * PIECE_BYTES;
  static constexpr int WSTEP_BYTES = 3 * NT * 64 * 16;     // one k-step: 3 pieces x NT row tiles x 64 lanes x 8 bf16
  static constexpr int WGT_BYTES = NSTEP * WSTEP_BYTES;
  static constexpr int LDS_BYTES = ACT_BYTES + WGT_BYTES;
  static constexpr int ULOAD = (4 + TX + 1 + 3) / 4;       // 16-byte units loaded per staged row (columns x0 - 4 .. x0 + TX)
  static constexpr int NUNIT = ZS * ROWS * ULOAD;          // (z, y, 4 columns) units per chunk
  static constexpr int XS = TX / G, GR = 32 / G;           // tiles per row group, rows per group
  static constexpr int MT = (G == 16 ? TY / 2 : TY / 4) * XS;
  static constexpr int TR_PITCH = 36;
  static constexpr bool A_AHEAD = NT == 1;                 // weight fragments one k-step ahead (register budget)
  // lane j of a B tile -> (row, column) inside the tile; tile mt -> (row, column) origin
  __device__ static constexpr int lane_row(int j) { return G == 16 ? (j >> 4) * 2 : (j >> 3); }
  __device__ static constexpr int lane_col(int j) { return j & (G - 1); }
  __device__ static constexpr int tile_row(int mt) { return G == 16 ? mt / XS : 4 * (mt / XS); }
  __device__ static constexpr int tile_col(int mt) { return (mt % XS) * G; }
  __host__ __device__ static constexpr int tapoff(int t) {   // voxel offset of tap t (t = 27: the zero-weight padding tap)
    return t >= 27 ? 0 : (t / 9) * PLANE + ((t / 3) % 3) * P + (t % 3);
  }
  static_assert(ULOAD * 4 <= P && NUNIT <= 512 && LDS_BYTES <= 80 * 1024 && ACT_BYTES >= 4 * 32 * TR_PITCH * 4, "tile");
  static_assert(G == 16 ? (2 * P * 2) % 64 == 32 : (P * 2) % 64 == 16, "rows of a tile must interleave the 64 LDS banks");
};
typedef X6Cfg<1, 16, 48, 56> X6C32;   // 32 output channels, W % 48 == 0 (full-resolution layers)
typedef X6Cfg<2, 8, 24, 40> X6C64;    // 64 output channels, W % 24 == 0 (half-resolution layers)
constexpr int X6_CK = 4, X6_NSTEP = 7;

// x = hi + mid + lo exactly, each a bf16 (round-to-nearest pieces, exact residuals); two values at a time so that
// every conversion is one v_cvt_pk_bf16_f32 and the results are already packed pairs.
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  f32x2 v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  f32x2 r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  f32x2 r2 = {r.x - __uint_as_float(mid << 16), r.y - __uint_as_float(mid & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

// wp[((((chunk * 7 + step) * 3 + piece) * NT + nt) * 64 + lane) * 8 + e] = piece(W(co = 32 nt + (lane & 31),
// ci = 4 chunk + (e & 3), tap = 4 step + 2 (lane >> 5) + (e >> 2))), zero for tap >= 27 or ci >= Ci.
__global__ void pack_x6_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Co, int Ci, int nchunk, int NT) {
  const long long total = (long long)nchunk * X6_NSTEP * NT * 64 * 8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    long long r = i >> 9;
    const int nt = (int)(r % NT);
    r /= NT;
    const int step = (int)(r % X6_NSTEP), chunk = (int)(r / X6_NSTEP);
    const int co = 32 * nt + (lane & 31), ci = 4 * chunk + (e & 3), tap = 4 * step + 2 * (lane >> 5) + (e >> 2);
    const float v = (co < Co && ci < Ci && tap < 27) ? w[((size_t)co * Ci + ci) * 27 + tap] : 0.f;
    unsigned h, m, l;
    split_pair(v, 0.f, h, m, l);
    const size_t base = ((((size_t)(chunk * X6_NSTEP + step) * 3) * NT + nt) * 64 + lane) * 8 + e;
    const size_t pstride = (size_t)NT * 64 * 8;
    wp[base] = (unsigned short)(h & 0xffffu);
    wp[base + pstride] = (unsigned short)(m & 0xffffu);
    wp[base + 2 * pstride] = (unsigned short)(l & 0xffffu);
  }
}

template <class C>
__global__ __launch_bounds__(256, 2) void conv3d_s1_x6_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wp,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ res, float* __restrict__ y, int Ci,
                                                              int D, int H, int W, int ntx, int nty, int ntz, int relu) {
  constexpr int CK = C::CK, P = C::P, ROWS = C::ROWS, PLANE = C::PLANE, MT = C::MT, NT = C::NT, COUT = 32 * C::NT;
  constexpr int PIECE_BYTES = C::PIECE_BYTES, WGT_BYTES = C::WGT_BYTES, TR_PITCH = C::TR_PITCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
  unsigned char* act = lds8;
  unsigned char* wgt = lds8 + C::ACT_BYTES;
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % ntx;
  t /= ntx;
  const int ty = t % nty;
  t /= nty;
  const int tz = t % ntz;
  const int b = t / ntz;
  const int x0 = tx * C::TX, y0 = ty * C::TY, z0 = tz * C::TZ;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const unsigned HW = (unsigned)H * W, DHW = (unsigned)D * HW;
  const float* xb = x + (size_t)b * Ci * DHW;
  const int nchunk = cdiv(Ci, CK);

  // ---- staging: unit u = (zz, yy, 4 columns); a thread owns units tid and tid + 256 and loads their 4 channels
  int uoff[2];       // byte offset of the unit inside one channel volume (or out of range)
  int uvox[2];       // first voxel index of the unit in the LDS tile
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int u = (int)threadIdx.x + q * 256;
    const int zz = u / (ROWS * C::ULOAD), rr = u - zz * (ROWS * C::ULOAD), yy = rr / C::ULOAD, sg = rr - yy * C::ULOAD;
    const int gz = z0 - 1 + zz, gy = y0 - 1 + yy, gx = x0 - 4 + sg * 4;
    const bool ok = u < C::NUNIT && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
    uoff[q] = ok ? (int)(((unsigned)gz * HW + (unsigned)gy * W + (unsigned)gx) * 4u) : (int)DMA_OOB;
    uvox[q] = (zz * ROWS + yy) * P + sg * 4;
  }
  u32x4 pre[2][CK];
  auto fetch = [&](int c0) {
    // the resource covers only this chunk's channels: a channel >= Ci is out of range and reads zeros
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(xb + (size_t)c0 * DHW, (unsigned)min(CK, Ci - c0) * DHW * 4u);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 1 && C::NUNIT <= 256) break;
#pragma unroll
      for (int c = 0; c < CK; ++c)
        pre[q][c] = __builtin_amdgcn_raw_buffer_load_b128(xrs, uoff[q] == (int)DMA_OOB ? (int)DMA_OOB : uoff[q] + (int)((unsigned)c * DHW * 4u), 0, 0);
    }
  };
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(wp, (unsigned)nchunk * WGT_BYTES);
  auto split_store = [&](int chunk) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 1 && (C::NUNIT <= 256 || (int)threadIdx.x + 256 >= C::NUNIT)) break;
      unsigned vals[CK][4];
#pragma unroll
      for (int c = 0; c < CK; ++c) {
        vals[c][0] = pre[q][c].x;
        vals[c][1] = pre[q][c].y;
        vals[c][2] = pre[q][c].z;
        vals[c][3] = pre[q][c].w;
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        unsigned h01, m01, l01, h23, m23, l23;
        split_pair(__uint_as_float(vals[0][v]), __uint_as_float(vals[1][v]), h01, m01, l01);
        split_pair(__uint_as_float(vals[2][v]), __uint_as_float(vals[3][v]), h23, m23, l23);
        unsigned char* dst = act + (uvox[q] + v) * 8;
        *reinterpret_cast<u32x2*>(dst) = u32x2{h01, h23};
        *reinterpret_cast<u32x2*>(dst + PIECE_BYTES) = u32x2{m01, m23};
        *reinterpret_cast<u32x2*>(dst + 2 * PIECE_BYTES) = u32x2{l01, l23};
      }
    }
    // this chunk's weight fragments (21 KB per 32 output channels), LDS-DMA in 16-byte words
#pragma unroll
    for (int i = 0; i < (WGT_BYTES / 16 + 255) / 256; ++i) {
      const int q4 = i * 256 + wave * 64 + lane;
      if (i * 256 + wave * 64 < WGT_BYTES / 16 && q4 < WGT_BYTES / 16)
        dma16(wrs, (unsigned)q4 * 16u, (unsigned)chunk * WGT_BYTES, reinterpret_cast<float*>(wgt + (i * 256 + wave * 64) * 16));
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int lane_vox = wave * PLANE + C::lane_row(j) * P + C::lane_col(j) + C::XOFF;
  fetch(0);
  for (int ci = 0; ci < nchunk; ++ci) {
    split_store(ci);                      // waits for the prefetched registers (vmcnt) as it consumes them
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the weight copy has landed
    __syncthreads();
    if (ci + 1 < nchunk) fetch((ci + 1) * CK);   // in flight under the MFMAs below
    // One flat sequence of (k-step, tile) iterations: lanes 0-31 multiply taps 4s, 4s+1, lanes 32-63 taps 4s+2, 4s+3.
    // B fragments travel from LDS one tile ahead (also across k-steps), A fragments one k-step ahead.
    auto load_a = [&](int s, bf16x8 (&d)[NT][3]) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          d[nt][p] = *reinterpret_cast<const bf16x8*>(wgt + (((s * 3 + p) * NT + nt) * 64 + lane) * 16);
    };
    auto load_b = [&](int s, int mt, bf16x8 (&d)[3]) {
      const int offA = (lane_vox + (h ? C::tapoff(4 * s + 2) : C::tapoff(4 * s))) * 8;
      const int offB = (lane_vox + (h ? C::tapoff(4 * s + 3) : C::tapoff(4 * s + 1))) * 8;
      const int to = (C::tile_row(mt) * P + C::tile_col(mt)) * 8;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const u32x2 lo = *reinterpret_cast<const u32x2*>(act + p * PIECE_BYTES + offA + to);
        const u32x2 hi = *reinterpret_cast<const u32x2*>(act + p * PIECE_BYTES + offB + to);
        const u32x4 q = {lo.x, lo.y, hi.x, hi.y};
        d[p] = __builtin_bit_cast(bf16x8, q);
      }
    };
    constexpr int ADEPTH = C::A_AHEAD ? 2 : 1;
    bf16x8 a[ADEPTH][NT][3], bq[2][3];
    load_a(0, a[0]);
    load_b(0, 0, bq[0]);
#pragma unroll
    for (int it = 0; it < C::NSTEP * MT; ++it) {
      const int s = it / MT, mt = it - s * MT;
      if (C::A_AHEAD && mt == 0 && s + 1 < C::NSTEP) load_a(s + 1, a[(s + 1) & 1]);
      if (!C::A_AHEAD && mt == 0 && s > 0) load_a(s, a[0]);
      if (it + 1 < C::NSTEP * MT) load_b((it + 1) / MT, (it + 1) % MT, bq[(it + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const auto& aa = a[C::A_AHEAD ? (s & 1) : 0];
      const auto& bb = bq[it & 1];
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // smallest cross terms first
#pragma unroll
      for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[nt][PA[t6]], bb[PB[t6]], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // everyone is done reading this chunk: the next split may overwrite it
  }

  // ---- epilogue: as the exact row-group kernel (transposition through LDS, 16-byte stores, residual one tile ahead)
  float* my = reinterpret_cast<float*>(act) + wave * (32 * TR_PITCH);
  float* yb = y + (size_t)b * COUT * DHW;
  const __amdgpu_buffer_rsrc_t yrs = make_rsrc(yb, (unsigned)COUT * DHW * 4u);
  const __amdgpu_buffer_rsrc_t rrs = make_rsrc(res ? res + (size_t)b * COUT * DHW : yb, (unsigned)COUT * DHW * 4u);
  const int gz = z0 + wave;
  const int px = (lane & 7) * 4;
  const float lo1 = relu == 1 ? 0.f : -__builtin_inff(), lo2 = relu == 2 ? 0.f : -__builtin_inff();
  // tile u of the epilogue = (mt, nt); a lane owns 4 consecutive columns of channel 32 nt + 8 k + lane / 8
  auto offsets = [&](int u, unsigned (&off)[4]) {
    const int mt = u / NT, nt = u - mt * NT;
    const int gy = y0 + C::tile_row(mt) + C::lane_row(px), gxo = x0 + C::tile_col(mt) + C::lane_col(px);
    const bool inb = gz < D && gy < H && gxo < W;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      off[k] = inb ? ((unsigned)(nt * 32 + k * 8 + (lane >> 3)) * DHW + (unsigned)gz * HW + (unsigned)gy * W + (unsigned)gxo) * 4u : DMA_OOB;
  };
  auto run = [&](auto has_res) {
    constexpr bool HAS_RES = decltype(has_res)::value;
    unsigned off[2][4];
    u32x4 rv[2][4];
    offsets(0, off[0]);
    if constexpr (HAS_RES) {
#pragma unroll
      for (int k = 0; k < 4; ++k) rv[0][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)off[0][k], 0, 0);
    }
#pragma unroll
    for (int u = 0; u < MT * NT; ++u) {
      const int mt = u / NT, nt = u - mt * NT;
      if (u + 1 < MT * NT) {
        offsets(u + 1, off[(u + 1) & 1]);
        if constexpr (HAS_RES) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            rv[(u + 1) & 1][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)off[(u + 1) & 1][k], 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) my[cd_row(r, h) * TR_PITCH + j] = acc[mt][nt][r];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int co = nt * 32 + k * 8 + (lane >> 3);
        const float sck = scale ? scale[co] : 1.f, shk = shift ? shift[co] : 0.f;
        float4 v = *reinterpret_cast<const float4*>(my + (k * 8 + (lane >> 3)) * TR_PITCH + px);
        v.x = fmaxf(fmaf(v.x, sck, shk), lo2);
        v.y = fmaxf(fmaf(v.y, sck, shk), lo2);
        v.z = fmaxf(fmaf(v.z, sck, shk), lo2);
        v.w = fmaxf(fmaf(v.w, sck, shk), lo2);
        if constexpr (HAS_RES) {
          v.x += __uint_as_float(rv[u & 1][k].x);
          v.y += __uint_as_float(rv[u & 1][k].y);
          v.z += __uint_as_float(rv[u & 1][k].z);
          v.w += __uint_as_float(rv[u & 1][k].w);
        }
        u32x4 o;
        o.x = __float_as_uint(fmaxf(v.x, lo1));
        o.y = __float_as_uint(fmaxf(v.y, lo1));
        o.z = __float_as_uint(fmaxf(v.z, lo1));
        o.w = __float_as_uint(fmaxf(v.w, lo1));
        __builtin_amdgcn_raw_buffer_store_b128(o, yrs, (int)off[u & 1][k], 0, 0);
      }
    }
  };
  if (res)
    run(std::true_type{});
  else
    run(std::false_type{});
}

template <class C>
static int launch_x6(const float* x, const void* wpack, const float* scale, const float* shift, const float* residual,
                     float* y, int B, int Ci, int D, int H, int W, int relu, hipStream_t st) {
  const int ntx = W / C::TX, nty = cdiv(H, C::TY), ntz = cdiv(D, C::TZ);
  const long long nblk = (long long)B * ntx * nty * ntz;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv3d_x6: grid too large");
  DMB_ENSURE_LDS((&conv3d_s1_x6_kernel<C>), (size_t)(C::LDS_BYTES));
  hipLaunchKernelGGL((conv3d_s1_x6_kernel<C>), dim3((unsigned)nblk), dim3(256), C::LDS_BYTES, st, x,
                     (const unsigned short*)wpack, scale, shift, residual, y, Ci, D, H, W, ntx, nty, ntz, relu);
  return launch_status("conv3d_x6 launch failed");
}

}  // namespace dmb

using namespace dmb;

extern "C" long long dmb_conv3d_x6_packed_bytes(int Co, int Ci) {
  if ((Co != 32 && Co != 64) || Ci <= 0) return 0;
  return (long long)cdiv(Ci, X6_CK) * X6_NSTEP * 3 * (Co / 32) * 64 * 16;
}

extern "C" int dmb_conv3d_x6_pack_weights_f32(const float* w, void* wpack, int Co, int Ci, void* stream) {
  if (!w || !wpack || (Co != 32 && Co != 64) || Ci <= 0) return fail(DMB_EINVAL, "conv3d_x6_pack: 32 or 64 output channels");
  hipLaunchKernelGGL(pack_x6_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)wpack, Co, Ci,
                     cdiv(Ci, X6_CK), Co / 32);
  return launch_status("conv3d_x6_pack launch failed");
}

extern "C" int dmb_conv3d_k3_x6_f32(const float* x, const void* wpack, const float* scale, const float* shift,
                                    const float* residual, float* y, int B, int Ci, int Co, int D, int H, int W,
                                    int relu, void* stream) {
  if (!x || !wpack || !y || B <= 0 || Ci <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "conv3d_x6: bad argument");
  const bool ok32 = Co == 32 && W % X6C32::TX == 0, ok64 = Co == 64 && W % X6C64::TX == 0;
  if ((!ok32 && !ok64) || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) != 0 ||
      (long long)Co * D * H * W * 4 >= 0x7fffffffLL)
    return fail(DMB_EUNSUPPORTED, "conv3d_x6: 32 output channels with W % 48 == 0 or 64 with W % 24 == 0, 16-byte aligned "
                                  "tensors, output item < 2 GiB");
  hipStream_t st = (hipStream_t)stream;
  if (ok32) return launch_x6<X6C32>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
  return launch_x6<X6C64>(x, wpack, scale, shift, residual, y, B, Ci, D, H, W, relu, st);
}
