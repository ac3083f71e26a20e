// A non-Python host of the C ABI (include/stage_hip.h): StructuredAttention forward (model/context_query_attention.py:35-101)
// on deterministic inputs, straight from C++ with the HIP runtime -- no torch anywhere.
//
//   hipcc --offload-arch=gfx950 -O2 -I include examples/k1_forward_host.cpp -L tvqaplus_amd -lstage_hip \
//         -Wl,-rpath,$PWD/tvqaplus_amd -o build/k1_forward_host && build/k1_forward_host
//
// Prints "A <sum> S_norm <sum> rows <n>"; tests/test_abi.py builds it, tests/test_hip_ops.py runs it on the GPU and compares the
// sums with the Python binding on the same inputs.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "stage_hip.h"

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); return 2; } } while (0)

int main() {
    const int N = 2, NA = 5, Li = 6, Lqa = 12, Lr = 20, D = 128;
    const long nC = (long)N * NA * Lqa * D, nQ = (long)N * Li * Lr * D, U = (long)N * NA * Li * Lqa;
    std::vector<float> C(nC), Q(nQ), cm((long)N * NA * Lqa), qm((long)N * Li * Lr);
    for (long i = 0; i < nC; i++) C[i] = std::sin(0.37f * (float)i) + 0.1f;          // the Python side regenerates these
    for (long i = 0; i < nQ; i++) Q[i] = std::cos(0.11f * (float)i) * 1.5f;
    for (long i = 0; i < (long)cm.size(); i++) cm[i] = (i % Lqa) < 9 ? 1.f : 0.f;    // three padded QA words per candidate
    for (long i = 0; i < (long)qm.size(); i++) qm[i] = (i % Lr) < 17 ? 1.f : 0.f;     // three padded regions per frame
    float *dC, *dCn, *dQ, *dcm, *dqm, *dA, *dS, *dSn;
    CK(hipMalloc(&dC, nC * 4)); CK(hipMalloc(&dCn, nC * 4)); CK(hipMalloc(&dQ, nQ * 4));
    CK(hipMalloc(&dcm, cm.size() * 4)); CK(hipMalloc(&dqm, qm.size() * 4));
    CK(hipMalloc(&dA, U * D * 4)); CK(hipMalloc(&dS, U * Lr * 4)); CK(hipMalloc(&dSn, U * Lr * 4));
    CK(hipMemcpy(dC, C.data(), nC * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dQ, Q.data(), nQ * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcm, cm.data(), cm.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dqm, qm.data(), qm.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    if (stage_hip_abi_version() != STAGE_HIP_ABI_VERSION) { std::fprintf(stderr, "ABI version\n"); return 3; }
    int rc = stage_l2norm_fwd(dC, dCn, nullptr, (long long)N * NA * Lqa, D, 1e-12f, 0.f, 0ull, st);     // the context side, normalised
    if (rc == 0) rc = stage_str_attn_fwd(dCn, dQ, dcm, dqm, dA, dS, dSn, N, NA, Li, Lqa, Lr, D, 10.0f, 0.f, 0ull, st);
    if (rc != 0) { std::fprintf(stderr, "stage_hip: %s\n", stage_hip_error_string(rc)); return 4; }
    CK(hipStreamSynchronize(st));
    std::vector<float> A(U * D), Sn(U * Lr);
    CK(hipMemcpy(A.data(), dA, A.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(Sn.data(), dSn, Sn.size() * 4, hipMemcpyDeviceToHost));
    double sa = 0, ss = 0;
    for (float v : A) sa += v;
    for (float v : Sn) ss += v;
    std::printf("A %.6f S_norm %.6f rows %ld\n", sa, ss, U);
    return 0;
}
