// ABI bookkeeping: version and error strings for include/dsmnet_hip.h.
#include "common.hpp"

extern "C" int dsm_abi_version(void) { return DSM_ABI_VERSION; }

extern "C" const char* dsm_strerror(int code) {
  switch (code) {
    case DSM_OK: return "ok";
    case DSM_ERR_ARG: return "invalid argument (null pointer, non-positive size or bad enum)";
    case DSM_ERR_UNSUPPORTED: return "request not supported by this build";
    case DSM_ERR_LAUNCH: return "HIP kernel launch failed";
    case DSM_ERR_ALIGN: return "pointer not 16-byte aligned";
    default: return "unknown dsmnet_hip error code";
  }
}
