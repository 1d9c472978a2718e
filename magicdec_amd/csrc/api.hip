// Host-side plumbing of libmagicdec_hip.so: error strings, ABI version.
#include "md_common.h"
#include <string.h>

namespace {
thread_local char g_err[512] = "";
}

void md_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int md_abi_version(void) { return 10; }
extern "C" const char* md_last_error_string(void) { return g_err; }
// returns AND clears the calling thread's sticky HIP error (hipGetLastError): after a hipGraph capture was invalidated
// the runtime keeps reporting that error at the next launch check until somebody has read it (Engine/graph.py fall-back)
extern "C" int md_clear_last_hip_error(void) { return (int)hipGetLastError(); }
