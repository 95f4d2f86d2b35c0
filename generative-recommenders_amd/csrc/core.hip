// ABI version and error strings of libhmx.so.
#include "hmx_common.h"

extern "C" int hmx_abi_version(void) { return 1; }

extern "C" const char* hmx_error_string(int code) {
    switch (code) {
        case HMX_OK: return "ok";
        case HMX_EINVAL: return "invalid argument (shape, null pointer or unsupported size)";
        case HMX_EWORKSPACE: return "workspace too small";
        case HMX_ELAUNCH: return "HIP launch error";
        case HMX_ELIMIT: return "input exceeds the fixed scratch size of this no-allocation entry point";
        default: return "unknown hmx error code";
    }
}
