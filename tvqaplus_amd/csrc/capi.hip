// Library-level entry points of libstage_hip.so (see include/stage_hip.h).
#include "common.h"
#include "../../include/stage_hip.h"

extern "C" int stage_hip_abi_version(void) { return STAGE_HIP_ABI_VERSION; }   // 2: round 4 (ragged token rows: stage_rag_*, *_fc, *_rag); 3: balanced work table (wtab) of stage_cat3_dx_ln_bwd_rag, T[5]; 4: round 5 (stage_ts_loss na_total, stage_mha_core_qkv_*); 5: round 6 (stage_cat3_bwd_dw*: the [a,b,a*b] backward with the Linear's gradients inside, z optional in stage_cat3_ln_gemm_fwd*)

extern "C" const char* stage_hip_error_string(int code) {
    if (code == 0) return "success";
    if (code == STAGE_ERR_SHAPE) return "stage_hip: unsupported shape";
    if (code == STAGE_ERR_WORKSPACE) return "stage_hip: workspace too small";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "stage_hip: unknown error";
}

// event helpers for hosts without a HIP binding of their own (bench.py times kernels on the launch stream with them)
extern "C" void* stage_timer_create(void) {
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? (void*)e : nullptr;
}
extern "C" void stage_timer_destroy(void* e) { if (e) (void)hipEventDestroy((hipEvent_t)e); }
extern "C" float stage_timer_elapsed_ms(void* start, void* stop) {
    float ms = -1.f;
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess || hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return -1.f;
    return ms;
}
