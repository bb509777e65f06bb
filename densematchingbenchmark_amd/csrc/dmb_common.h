// Internal helpers shared by the HIP translation units of libdmb_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/dmb_hip.h"

namespace dmb {

void set_last_error(const char* msg);

// Development knobs exist only in the DEVELOPMENT build of the library (-DDMB_DEV: lib/libdmb_hip_dev.so, loaded by scripts/
// through DMB_LIB=dev, never by the package, the tests or bench.py).  In the release build DMB_OPT(k) is the constant 0 and
// DMB_DBG(expr) the constant 0: no global option table, no diagnostic branch in any kernel, dmb_dev_set_option not exported.
#ifdef DMB_DEV
extern int g_dev_opts[32];
#define DMB_OPT(k) (::dmb::g_dev_opts[(k)])
#define DMB_DBG(expr) (expr)
#else
#define DMB_OPT(k) 0
#define DMB_DBG(expr) 0
#endif

// Compute units of the CURRENT device (cached per device ordinal; a process may drive several GPUs).
int num_cus();

inline int fail(int code, const char* msg) {
  set_last_error(msg);
  return code;
}

// Launch check: launches are asynchronous, so this only catches configuration errors, which is
// what the reference's native op checks too (gaterecurrent2dnoind_kernel.cu:544-549) -- but we
// return the code instead of calling exit().
inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error(what);
    return (int)e;
  }
  return DMB_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: `*seen` (one static per kernel
// instantiation) remembers the device ordinals it has been raised on, so a process that moves to a second GPU
// (torch.cuda.device(1)) sets it there too.  Returns DMB_OK or the failure (never ignored: the launch would fail opaquely).
inline int ensure_dynamic_lds(const void* kernel, size_t bytes, unsigned long long* seen) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (*seen & bit) return DMB_OK;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return fail((int)e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
  *seen |= bit;
  return DMB_OK;
}
#define DMB_ENSURE_LDS(kernel, bytes)                                                                       \
  do {                                                                                                      \
    static unsigned long long dmb_seen_ = 0;                                                                \
    const int dmb_rc_ = dmb::ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (bytes), &dmb_seen_); \
    if (dmb_rc_ != DMB_OK) return dmb_rc_;                                                                  \
  } while (0)

// Disparity sample indices travel as a by-value kernel argument (<= 1 KiB of kernarg).
struct DispIdx {
  int d[DMB_MAX_DISP_SAMPLES];
};
struct DispVal {
  float v[DMB_MAX_DISP_SAMPLES];
};

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Workgroup -> tile remap so that consecutive tile ids land on the same XCD (workgroup b is placed on
// XCD b % 8; each XCD has a private 4 MiB L2, and neighbouring tiles share halo voxels).  Bijective for
// any grid size (cdna_hip_programming.md, "XCD swizzle must be bijective").  Speed only, never correctness.
__device__ inline int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}


// ------------------------------------------------------------------------------------------------------------------
// MFMA / LDS-DMA helpers shared by the implicit-GEMM kernels (conv3d.hip, confhead.hip, gwc_mfma.hip)
// ------------------------------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define DMB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Asynchronous global -> LDS copies (LDS-DMA, buffer_load ... lds): no VGPR round trip, no ds_write.  The LDS
// destination is the wave-uniform `dst` + lane * size; the global source is base(rsrc) + soffset (scalar) + voffset
// (per lane).  Zero padding comes for free from the buffer bounds check: a lane whose voffset is DMA_OOB reads 0.
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr unsigned DMA_OOB = 0x80000000u;  // >= any num_records we create (host side checks sizes < 2 GiB)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void dma4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, float* dst_uniform) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst_uniform, 4, voff, soff, 0, 0);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, float* dst_uniform) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst_uniform, 16, voff, soff, 0, 0);
}

// Row of the 32x32 C/D tile held by accumulator register r of lane-half h (cdna_hip_programming.md s3).
__device__ __forceinline__ int cd_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

#endif  // __HIPCC__

// csrc/deconv3d_zy.hip: the (tile, z parity, y parity) form of the transposed convolution; -1 = does not apply.
// `workspace`: DMB_DECONV3D_WORKSPACE_BYTES of device memory holding zeros (see include/dmb_hip.h); the launch leaves it zeroed.
int deconv3d_zy_try(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                    int B, int Ci, int Co, int D, int H, int W, int Wout, int relu, int* workspace, hipStream_t st);

// csrc/conv3d_sk.hip: split-K form of the stride-1 / stride-2 convolution for launches that do not fill the chip; -1 = does not apply.
int conv3d_sk_try(int variant, const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                  int B, int Ci, int Co, int D, int H, int W, int stride, int relu, hipStream_t st);

int conv2d_sk_try(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y, int B, int Ci,
                  int Co, int H, int W, int dilation, int relu, int in_ctot, int out_ctot, int res_ctot, hipStream_t st);
int deconv3d_sk_try(int variant, const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                    int B, int Ci, int Co, int D, int H, int W, int Wout, int relu, hipStream_t st);

}  // namespace dmb
