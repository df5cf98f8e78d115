// Library-wide pieces of the C ABI: version and last-error text.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mplattice.h"

static thread_local char g_err[512] = "";

void mp_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int mp_version(void) { return MP_VERSION; }
extern "C" const char *mp_last_error(void) { return g_err; }
