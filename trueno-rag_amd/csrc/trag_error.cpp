// Storage of the per-thread error message behind trag_last_error().
#include "trag_error.h"

namespace {
thread_local std::string g_last_error;
}

namespace trag {
trag_status set_error(trag_status st, const std::string& msg) {
    g_last_error = msg;
    return st;
}
}  // namespace trag

extern "C" const char* trag_last_error(void) { return g_last_error.c_str(); }
