// ABI bookkeeping entry points.
#include "re_common.h"

extern "C" int re_abi_version(void) { return 1; }

extern "C" const char* re_error_string(int code) {
    switch (code) {
        case RE_OK: return "ok";
        case RE_EINVAL: return "invalid argument";
        case RE_EWORKSPACE: return "workspace too small";
        case RE_ELAUNCH: return "kernel launch failed";
        case RE_EUNSUPPORTED: return "unsupported shape/alignment";
        default: return "unknown error";
    }
}
