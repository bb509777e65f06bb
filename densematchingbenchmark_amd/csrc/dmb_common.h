// Internal helpers shared by the HIP translation units of libdmb_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/dmb_hip.h"

namespace dmb {

void set_last_error(const char* msg);

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

}  // namespace dmb
