// Error channel of the C ABI: functions return negative codes and never throw;
// the message of the most recent failure on the calling thread is kept here.
#include "common.h"
#include <cstdarg>

namespace dic {
static thread_local char g_err[1024] = "";

void set_last_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }
}  // namespace dic
