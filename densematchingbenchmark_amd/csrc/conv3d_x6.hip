// EXPERIMENTAL, OPT-IN: 3x3x3 stride-1 convolution (32 output channels) with FP32 operands split exactly into three
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

namespace x6 {
constexpr int TX = 48, TY = 4, TZ = 4, CK = 4, NSTEP = 7;
constexpr int P = 56, ROWS = TY + 2, ZS = TZ + 2, PLANE = ROWS * P, NVOX = ZS * PLANE;   // 2016 staged voxels
constexpr int XOFF = 3;                                  // the staged row starts at the aligned column x0 - 4
constexpr int PIECE_BYTES = NVOX * 8;                    // [voxel][4 channels] bf16
constexpr int ACT_BYTES = 3 * PIECE_BYTES;               // 48384
constexpr int WSTEP_BYTES = 3 * 64 * 16;                 // one k-step: 3 pieces x 64 lanes x 8 bf16
constexpr int WGT_BYTES = NSTEP * WSTEP_BYTES;           // 21504
constexpr int LDS_BYTES = ACT_BYTES + WGT_BYTES;         // 69888: two workgroups per CU
constexpr int UPR = P / 4, NUNIT = ZS * ROWS * UPR;      // 504 (z, y, 4 columns) units per chunk
constexpr int XS = TX / 16, MT = (TY / 2) * XS;          // 6 row-pair tiles per wave
constexpr int TR_PITCH = 36;
__host__ __device__ constexpr int tapoff(int t) {        // voxel offset of tap t (t = 27 is the zero-weight padding tap)
  return t >= 27 ? 0 : (t / 9) * PLANE + ((t / 3) % 3) * P + (t % 3);
}
}  // namespace x6

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

// wp[(((chunk * 7 + step) * 3 + piece) * 64 + lane) * 8 + e] = piece(W(co = lane & 31, ci = 4 chunk + (e & 3),
// tap = 4 step + 2 (lane >> 5) + (e >> 2))), zero for tap >= 27 or ci >= Ci.
__global__ void pack_x6_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Co, int Ci, int nchunk) {
  const long long total = (long long)nchunk * x6::NSTEP * 64 * 8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long long cs = i >> 9;
    const int step = (int)(cs % x6::NSTEP), chunk = (int)(cs / x6::NSTEP);
    const int co = lane & 31, ci = 4 * chunk + (e & 3), tap = 4 * step + 2 * (lane >> 5) + (e >> 2);
    const float v = (co < Co && ci < Ci && tap < 27) ? w[((size_t)co * Ci + ci) * 27 + tap] : 0.f;
    unsigned h, m, l;
    split_pair(v, 0.f, h, m, l);
    const size_t base = ((size_t)(chunk * x6::NSTEP + step) * 3 * 64 + lane) * 8 + e;
    wp[base] = (unsigned short)(h & 0xffffu);
    wp[base + 64 * 8] = (unsigned short)(m & 0xffffu);
    wp[base + 2 * 64 * 8] = (unsigned short)(l & 0xffffu);
  }
}

__global__ __launch_bounds__(256, 2) void conv3d_s1_x6_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wp,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ res, float* __restrict__ y, int Ci,
                                                              int D, int H, int W, int ntx, int nty, int ntz, int relu) {
  using namespace x6;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
  unsigned char* act = lds8;
  unsigned char* wgt = lds8 + ACT_BYTES;
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % ntx;
  t /= ntx;
  const int ty = t % nty;
  t /= nty;
  const int tz = t % ntz;
  const int b = t / ntz;
  const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
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
    const int zz = u / (ROWS * UPR), rr = u - zz * (ROWS * UPR), yy = rr / UPR, sg = rr - yy * UPR;
    const int gz = z0 - 1 + zz, gy = y0 - 1 + yy, gx = x0 - 4 + sg * 4;
    const bool ok = u < NUNIT && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
    uoff[q] = ok ? (int)(((unsigned)gz * HW + (unsigned)gy * W + (unsigned)gx) * 4u) : (int)DMA_OOB;
    uvox[q] = (zz * ROWS + yy) * P + sg * 4;
  }
  u32x4 pre[2][CK];
  auto fetch = [&](int c0) {
    // the resource covers only this chunk's channels: a channel >= Ci is out of range and reads zeros
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(xb + (size_t)c0 * DHW, (unsigned)min(CK, Ci - c0) * DHW * 4u);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int c = 0; c < CK; ++c)
        pre[q][c] = __builtin_amdgcn_raw_buffer_load_b128(xrs, uoff[q] == (int)DMA_OOB ? (int)DMA_OOB : uoff[q] + (int)((unsigned)c * DHW * 4u), 0, 0);
  };
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(wp, (unsigned)nchunk * WGT_BYTES);
  auto split_store = [&](int chunk) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 1 && (int)threadIdx.x + 256 >= NUNIT) break;
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
    // this chunk's weight fragments: 21 KB = 1344 16-byte words, LDS-DMA
#pragma unroll
    for (int i = 0; i < (WGT_BYTES / 16 + 255) / 256; ++i) {
      const int q4 = i * 256 + wave * 64 + lane;
      if (i * 256 + wave * 64 < WGT_BYTES / 16 && q4 < WGT_BYTES / 16)
        dma16(wrs, (unsigned)q4 * 16u, (unsigned)chunk * WGT_BYTES, reinterpret_cast<float*>(wgt + (i * 256 + wave * 64) * 16));
    }
  };

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  // per-channel affine of this lane's output channels (4 of them after the epilogue's transposition), loaded once
  float sc4[4], sh4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int co = k * 8 + (lane >> 3);
    sc4[k] = scale ? scale[co] : 1.f;
    sh4[k] = shift ? shift[co] : 0.f;
  }

  // a B tile pairs rows r and r + 2 (not r + 1): 2 * P voxels = 224 dwords = 32 (mod 64 banks), so the two 16-lane halves
  // of a ds_read_b64 group cover all 64 banks exactly once (rows r, r + 1 would collide on 16 banks: 2 cycles per group)
  const int lane_vox = wave * PLANE + (j >> 4) * 2 * P + (j & 15) + XOFF;
  fetch(0);
  for (int ci = 0; ci < nchunk; ++ci) {
    split_store(ci);                 // waits for the prefetched registers (vmcnt) as it consumes them
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the weight copy has landed
    __syncthreads();
    if (ci + 1 < nchunk) fetch((ci + 1) * CK);   // in flight under the MFMAs below
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      // lanes 0-31 multiply taps 4s, 4s+1, lanes 32-63 taps 4s+2, 4s+3
      const int offA = (lane_vox + (h ? tapoff(4 * s + 2) : tapoff(4 * s))) * 8;
      const int offB = (lane_vox + (h ? tapoff(4 * s + 3) : tapoff(4 * s + 1))) * 8;
      bf16x8 a[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(wgt + ((s * 3 + p) * 64 + lane) * 16);
      // one tile ahead: the next tile's fragments travel from LDS while this tile's six MFMAs run
      bf16x8 bq[2][3];
      auto load_b = [&](int mt, bf16x8 (&d)[3]) {
        const int to = ((mt / XS) * P + (mt % XS) * 16) * 8;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const u32x2 lo = *reinterpret_cast<const u32x2*>(act + p * PIECE_BYTES + offA + to);
          const u32x2 hi = *reinterpret_cast<const u32x2*>(act + p * PIECE_BYTES + offB + to);
          const u32x4 q = {lo.x, lo.y, hi.x, hi.y};
          d[p] = __builtin_bit_cast(bf16x8, q);
        }
      };
      load_b(0, bq[0]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (mt + 1 < MT) load_b(mt + 1, bq[(mt + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const auto& bb = bq[mt & 1];
        // smallest cross terms first
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bb[0], acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bb[2], acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bb[1], acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bb[0], acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bb[1], acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bb[0], acc[mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();   // everyone is done reading this chunk: the next split may overwrite it
  }

  // ---- epilogue: as the exact row-pair kernel (transposition through LDS, 16-byte stores, residual one tile ahead)
  float* my = reinterpret_cast<float*>(act) + wave * (32 * TR_PITCH);
  float* yb = y + (size_t)b * 32 * DHW;
  const __amdgpu_buffer_rsrc_t yrs = make_rsrc(yb, 32u * DHW * 4u);
  const __amdgpu_buffer_rsrc_t rrs = make_rsrc(res ? res + (size_t)b * 32 * DHW : yb, 32u * DHW * 4u);
  const int gz = z0 + wave;
  const int px = (lane & 7) * 4;
  const float lo1 = relu == 1 ? 0.f : -__builtin_inff(), lo2 = relu == 2 ? 0.f : -__builtin_inff();
  auto offsets = [&](int mt, unsigned (&off)[4]) {
    const int gy = y0 + (mt / XS) + 2 * (px >> 4), gxo = x0 + (mt % XS) * 16 + (px & 15);
    const bool inb = gz < D && gy < H && gxo < W;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      off[k] = inb ? ((unsigned)(k * 8 + (lane >> 3)) * DHW + (unsigned)gz * HW + (unsigned)gy * W + (unsigned)gxo) * 4u : DMA_OOB;
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
    for (int mt = 0; mt < MT; ++mt) {
      if (mt + 1 < MT) {
        offsets(mt + 1, off[(mt + 1) & 1]);
        if constexpr (HAS_RES) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            rv[(mt + 1) & 1][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)off[(mt + 1) & 1][k], 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) my[cd_row(r, h) * TR_PITCH + j] = acc[mt][r];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float4 v = *reinterpret_cast<const float4*>(my + (k * 8 + (lane >> 3)) * TR_PITCH + px);
        v.x = fmaxf(fmaf(v.x, sc4[k], sh4[k]), lo2);
        v.y = fmaxf(fmaf(v.y, sc4[k], sh4[k]), lo2);
        v.z = fmaxf(fmaf(v.z, sc4[k], sh4[k]), lo2);
        v.w = fmaxf(fmaf(v.w, sc4[k], sh4[k]), lo2);
        if constexpr (HAS_RES) {
          v.x += __uint_as_float(rv[mt & 1][k].x);
          v.y += __uint_as_float(rv[mt & 1][k].y);
          v.z += __uint_as_float(rv[mt & 1][k].z);
          v.w += __uint_as_float(rv[mt & 1][k].w);
        }
        u32x4 o;
        o.x = __float_as_uint(fmaxf(v.x, lo1));
        o.y = __float_as_uint(fmaxf(v.y, lo1));
        o.z = __float_as_uint(fmaxf(v.z, lo1));
        o.w = __float_as_uint(fmaxf(v.w, lo1));
        __builtin_amdgcn_raw_buffer_store_b128(o, yrs, (int)off[mt & 1][k], 0, 0);
      }
    }
  };
  if (res)
    run(std::true_type{});
  else
    run(std::false_type{});
}

}  // namespace dmb

using namespace dmb;

extern "C" long long dmb_conv3d_x6_packed_bytes(int Co, int Ci) {
  if (Co <= 0 || Co > 32 || Ci <= 0) return 0;
  return (long long)cdiv(Ci, x6::CK) * x6::WGT_BYTES;
}

extern "C" int dmb_conv3d_x6_pack_weights_f32(const float* w, void* wpack, int Co, int Ci, void* stream) {
  if (!w || !wpack || Co <= 0 || Co > 32 || Ci <= 0) return fail(DMB_EINVAL, "conv3d_x6_pack: 1..32 output channels");
  hipLaunchKernelGGL(pack_x6_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)wpack, Co, Ci,
                     cdiv(Ci, x6::CK));
  return launch_status("conv3d_x6_pack launch failed");
}

extern "C" int dmb_conv3d_k3_x6_f32(const float* x, const void* wpack, const float* scale, const float* shift,
                                    const float* residual, float* y, int B, int Ci, int Co, int D, int H, int W,
                                    int relu, void* stream) {
  if (!x || !wpack || !y || B <= 0 || Ci <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "conv3d_x6: bad argument");
  if (Co != 32 || W % x6::TX != 0 || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) != 0 ||
      (long long)32 * D * H * W * 4 >= 0x7fffffffLL)
    return fail(DMB_EUNSUPPORTED, "conv3d_x6: 32 output channels, W a multiple of 48, 16-byte aligned tensors, output item < 2 GiB");
  const int ntx = W / x6::TX, nty = cdiv(H, x6::TY), ntz = cdiv(D, x6::TZ);
  const long long nblk = (long long)B * ntx * nty * ntz;
  if (nblk > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "conv3d_x6: grid too large");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_s1_x6_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              x6::LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(conv3d_s1_x6_kernel, dim3((unsigned)nblk), dim3(256), x6::LDS_BYTES, (hipStream_t)stream, x,
                     (const unsigned short*)wpack, scale, shift, residual, y, Ci, D, H, W, ntx, nty, ntz, relu);
  return launch_status("conv3d_x6 launch failed");
}
