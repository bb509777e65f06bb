// libdmb_hip.so bookkeeping: ABI version and the per-thread last-error string.
#include "dmb_common.h"

namespace dmb {
static thread_local const char* g_last_error = "";
void set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }
}  // namespace dmb

namespace dmb {
int g_dev_opts[32] = {1};  // development knobs (kernel variant selection in micro-benchmarks)
}
// Development knob, NOT part of the ABI (absent from include/dmb_hip.h): key 0 = conv scheduling variant,
// key 1 = 1 forces the VALU form of the group-wise correlation, key 2 = 1 forces flattened conv3d tiles,
// key 3 = 1 forces the scalar (dword) staging / store paths (conv2d, stride-2 and transposed conv3d).
extern "C" void dmb_dev_set_option(int key, int value) {
  if (key >= 0 && key < 32) dmb::g_dev_opts[key] = value;
}

extern "C" int dmb_abi_version(void) { return 3; }
extern "C" const char* dmb_last_error(void) { return dmb::g_last_error; }
