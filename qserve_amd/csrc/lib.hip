// lib.hip -- library identification and error reporting for libqserve_amd.so
#include "common.h"

static thread_local char g_err[512] = "";

void qs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int qs_version(void) { return 1; }
extern "C" const char* qs_arch(void) { return "gfx950"; }
extern "C" const char* qs_last_error(void) { return g_err; }
