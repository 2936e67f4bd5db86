// C-ABI housekeeping for libclibd_hip.so: error text, ABI version.
#include "host_util.h"
#include "../../include/clibd_hip.h"

namespace clibd {
char* last_error_buf() {
    static thread_local char buf[kErrBufLen] = {0};
    return buf;
}
}  // namespace clibd

extern "C" const char* clibd_last_error(void) { return clibd::last_error_buf(); }
extern "C" int clibd_abi_version(void) { return 1; }
