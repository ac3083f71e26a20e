// Ablation harness for the register-resident K1 forward kernel: compile with -DK1_ABL=<bits> and compare timings.
#include "../../tvqaplus_amd/csrc/str_attn_fwd_reg.hip"
#include <stdio.h>
#include <vector>
int main() {
    const int N = 16, NA = 5, Li = 300, Lqa = 40, Lr = 20, D = 128;
    const size_t nC = (size_t)N * NA * Lqa * D, nQ = (size_t)N * Li * Lr * D, U = (size_t)N * NA * Li * Lqa;
    std::vector<float> hC(nC), hQ(nQ), hcm((size_t)N * NA * Lqa, 1.f), hqm((size_t)N * Li * Lr, 1.f);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : hC) v = rnd() * 0.2f;
    for (auto& v : hQ) v = rnd();
    float *C, *Q, *cm, *qm, *A, *S, *Sn;
    hipMalloc(&C, nC * 4); hipMalloc(&Q, nQ * 4); hipMalloc(&cm, hcm.size() * 4); hipMalloc(&qm, hqm.size() * 4);
    hipMalloc(&A, U * D * 4); hipMalloc(&S, U * Lr * 4); hipMalloc(&Sn, U * Lr * 4);
    hipMemcpy(C, hC.data(), nC * 4, hipMemcpyHostToDevice); hipMemcpy(Q, hQ.data(), nQ * 4, hipMemcpyHostToDevice);
    hipMemcpy(cm, hcm.data(), hcm.size() * 4, hipMemcpyHostToDevice); hipMemcpy(qm, hqm.data(), hqm.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; r++) stage_str_attn_fwd_reg(C, Q, cm, qm, A, S, Sn, N, NA, Li, Lqa, Lr, D, 10.f, 0.f, 0, 0);
    hipEventRecord(e0);
    for (int r = 0; r < 20; r++) stage_str_attn_fwd_reg(C, Q, cm, qm, A, S, Sn, N, NA, Li, Lqa, Lr, D, 10.f, 0.f, 0, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("K1_ABL=%d  %.1f us\n", K1_ABL, ms * 50);
    return 0;
}
