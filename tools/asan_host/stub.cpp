// Host-only harness: the plan layout and greedy tree code with ASan/UBSan (no HIP runtime calls).
#include <cstdarg>
#include <cstdio>
#include "asp_common.hpp"
namespace asp {
ErrorState &error_state() { static thread_local ErrorState s; return s; }
int set_error(int code, const char *fmt, ...) { ErrorState &s = error_state(); s.code = code; va_list a; va_start(a, fmt); vsnprintf(s.message, sizeof s.message, fmt, a); va_end(a); return code; }
int pool_alloc(size_t, void **) { return -1; }
void pool_free(void *) {}
}
extern "C" void asp_clear_error(void) { asp::error_state().code = 0; }
extern "C" const char *asp_last_error(void) { return asp::error_state().message; }
