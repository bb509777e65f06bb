// libdmb_hip.so bookkeeping: ABI version, the per-thread last-error string, the per-device CU count.
#include <atomic>

#include "dmb_common.h"

namespace dmb {
static thread_local const char* g_last_error = "";
void set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }

// multiProcessorCount per device ordinal.  Read-mostly cache: a racing first use queries twice and stores the same value.
int num_cus() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n > 0) return n;
  hipDeviceProp_t prop;
  n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  cache[dev].store(n, std::memory_order_relaxed);
  return n;
}
}  // namespace dmb

#ifdef DMB_DEV
// Development knobs of the DEVELOPMENT build only (lib/libdmb_hip_dev.so; see dmb_common.h).  All 0 by default.  They select kernel
// variants for A/B measurements inside one process (scripts/ab_step.py, scripts/kbench_hg.py):
//    0  = 1: the round-1 flat stride-1 kernel for 32 output channels    1  = 1: VALU form of the group-wise correlation
//    2  = 1: flattened conv3d tiles (no row pairs / groups / runs)      3  = 1: scalar (dword) staging / store paths
//    4  transposed conv: 1 = deconv3d_kernel (round 2)                  5  rows per workgroup (gwc) / z segments (wgrad)
//    6  diagnostic bits of the conv3d kernels (no stores, no staging)   7  = 1: no zy / vector transposed paths
//    8  persistent grid multiplier (zy)    9  persistent grid override  10  stride 2: 1 = four-wave workgroups, 2 = dword epilogue
//   12  start-up stagger unit of the zy kernel                          13  = 1: box tiles instead of linear runs (stride-1, 64 ch)
//   14  start-up stagger unit of the stride-1 kernels                   16  zy item order: tiles per class run (0 = default)
//   17  extra KB of LDS per stride-1 workgroup (occupancy)              18  = 1: default tile height for the 64-channel conv2d layers
//   19  stride-1 tile override (see dmb_conv3d_k3_f32)                   20  = 1: zy items from ONE counter instead of one per XCD
//   21  = 32 / 64: options 16 and 20 for that output width only         22  up-sampling: 1 = row form, 2 = flat form
//   23  split-K conv3d: 1 = never, k >= 2 = force variant k - 1          24  32 -> 1 head: 1 = never the split-channel form, 2 = always
//   25  split-K transposed conv: 1 = never, k >= 2 = force variant k - 1    26  split-K 3x3 conv2d: 1 = never, 2 = always
namespace dmb {
int g_dev_opts[32] = {0};
}
extern "C" void dmb_dev_set_option(int key, int value) {
  if (key >= 0 && key < 32) dmb::g_dev_opts[key] = value;
}
#endif

extern "C" int dmb_abi_version(void) { return 8; }
extern "C" const char* dmb_last_error(void) { return dmb::g_last_error; }
