// Library-level entry points of libstage_hip.so (see include/stage_hip.h).
#include "common.h"
#include "../../include/stage_hip.h"

extern "C" int stage_hip_abi_version(void) { return 1; }

extern "C" const char* stage_hip_error_string(int code) {
    if (code == 0) return "success";
    if (code == STAGE_ERR_SHAPE) return "stage_hip: unsupported shape";
    if (code == STAGE_ERR_WORKSPACE) return "stage_hip: workspace too small";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "stage_hip: unknown error";
}
