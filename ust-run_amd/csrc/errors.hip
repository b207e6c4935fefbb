#include "common.h"
#include <stdarg.h>
namespace ustrun {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
const char* get_error() { return g_err; }
}
extern "C" int ustrun_version(void) { return USTRUN_VERSION; }
extern "C" const char* ustrun_last_error(void) { return ustrun::get_error(); }
