// libdmb_hip.so bookkeeping: ABI version and the per-thread last-error string.
#include "dmb_common.h"

namespace dmb {
static thread_local const char* g_last_error = "";
void set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }
}  // namespace dmb

namespace dmb {
int g_dev_opts[32] = {1};  // development knobs (kernel variant selection in micro-benchmarks)
}
// Development knobs, NOT part of the ABI (absent from include/dmb_hip.h); all 0 by default except key 0 = 1.  They select kernel
// variants for A/B measurements (scripts/ab_step.py, scripts/kbench_hg.py) and for the bit-identity tests between variants:
//    0  conv scheduling variant (0 = the round-1 flat stride-1 kernel)      1  = 1: VALU form of the group-wise correlation
//    2  = 1: flattened conv3d tiles (no row pairs / groups / runs)          3  = 1: scalar (dword) staging / store paths
//    4  transposed conv: 1 = deconv3d_kernel (round 2), 2 = force zy        5  rows per workgroup (gwc) / z segments (wgrad)
//    6  diagnostic bits of the conv3d kernels (no stores, no staging, ...)  7  = 1: no zy / vector transposed paths
//    8  persistent grid multiplier (zy)    9  persistent grid override     10  stride 2: 1 = four-wave workgroups, 2 = dword epilogue
//   11  sixteen-wave transposed conv (1 always, 2 only >= 6 tiles per CU)   12  start-up stagger unit of the zy kernel
//   13  = 1: box tiles for the quarter-resolution stride-1 layer            14  start-up stagger unit of the stride-1 kernels
//   15  = 1: one launch per head in dmb_conv3d_k3_c1_multi_f32              17  extra KB of LDS per stride-1 workgroup (occupancy)
//   18  = 1: default tile height for the 64-channel conv2d layers
extern "C" void dmb_dev_set_option(int key, int value) {
  if (key >= 0 && key < 32) dmb::g_dev_opts[key] = value;
}

extern "C" int dmb_abi_version(void) { return 3; }
extern "C" const char* dmb_last_error(void) { return dmb::g_last_error; }
