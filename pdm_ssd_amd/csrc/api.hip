// Error plumbing shared by every entry point of libpdmssd_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace pdm {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Launch-time errors only (bad configuration, missing code object ...); asynchronous faults surface
// at the caller's next synchronisation, as with any HIP launch.  Never exits the process.
int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
}

}  // namespace pdm

extern "C" int pdm_abi_version(void) { return PDM_ABI_VERSION; }
extern "C" const char *pdm_last_error(void) { return pdm::g_err; }
