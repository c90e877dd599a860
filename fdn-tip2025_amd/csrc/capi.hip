// ABI bookkeeping for libfdn_hip.so.
#include "common.hpp"

extern "C" int fdn_abi_version(void) { return 5; }

extern "C" const char* fdn_error_string(int code) {
    switch (code) {
        case FDN_OK: return "ok";
        case FDN_ERR_ARG: return "invalid argument (null pointer, bad shape or unsupported size)";
        case FDN_ERR_LAUNCH: return "HIP kernel launch failed";
        case FDN_ERR_WORKSPACE: return "workspace too small";
        case FDN_ERR_UNSUPPORTED: return "configuration not supported by this build";
        default: return "unknown error";
    }
}
