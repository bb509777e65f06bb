// libdmb_hip.so bookkeeping: ABI version and the per-thread last-error string.
#include "dmb_common.h"

namespace dmb {
static thread_local const char* g_last_error = "";
void set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }
}  // namespace dmb

extern "C" int dmb_abi_version(void) { return 1; }
extern "C" const char* dmb_last_error(void) { return dmb::g_last_error; }
