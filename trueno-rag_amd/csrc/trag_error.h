// The calling thread's last error message (trag_last_error in include/trag.h). Host-only, no HIP:
// shared by the device-facing files and by the pure host ones (fusion, MockEmbedder).
#pragma once
#include <string>
#include "trag.h"

namespace trag {
// Records the message for trag_last_error() on this thread and returns st.
trag_status set_error(trag_status st, const std::string& msg);
}  // namespace trag
