// Standalone GPU probe (no torch): checks lav_gemm_bf16 in its three layouts against a CPU fp32 reference and
// times a few LAVENDER shapes.  Build: see Makefile target `probe`.  Run on the GPU box only.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/lavender_hip.h"

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }
static float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

static int check(int layout, int M, int N, int K, int splits) {
    // A: layout 0/1 [M,K]; layout 2 [K,M].  B: layout 0 [N,K]; 1/2 [K,N]
    long lda = layout == 2 ? (M + 7) / 8 * 8 : (K + 7) / 8 * 8, ldb = layout == 0 ? (K + 7) / 8 * 8 : (N + 7) / 8 * 8;
    long arows = layout == 2 ? K : M, brows = layout == 0 ? N : K;
    std::vector<uint16_t> A(arows * lda, 0), B(brows * ldb, 0);
    std::vector<float> Af(arows * lda, 0.f), Bf(brows * ldb, 0.f);
    long acols = layout == 2 ? M : K, bcols = layout == 0 ? K : N;
    for (long r = 0; r < arows; ++r) for (long c = 0; c < acols; ++c) { A[r * lda + c] = f2bf(frand()); Af[r * lda + c] = bf2f(A[r * lda + c]); }
    for (long r = 0; r < brows; ++r) for (long c = 0; c < bcols; ++c) { B[r * ldb + c] = f2bf(frand()); Bf[r * ldb + c] = bf2f(B[r * ldb + c]); }
    long ldc = (N + 7) / 8 * 8;
    std::vector<float> ref((long)M * N, 0.f);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < K; ++k) {
                float a = layout == 2 ? Af[(long)k * lda + m] : Af[(long)m * lda + k];
                float b = layout == 0 ? Bf[(long)n * ldb + k] : Bf[(long)k * ldb + n];
                s += (double)a * b;
            }
            ref[(long)m * N + n] = (float)s;
        }
    void *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, (long)M * ldc * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dC, 0, (long)M * ldc * 4);
    lav_gemm_epilogue e; memset(&e, 0, sizeof(e)); e.alpha = 1.f; e.out_mode = splits > 1 ? 2 : 1;
    int rc = lav_gemm_bf16(nullptr, layout, M, N, K, dA, lda, dB, ldb, dC, ldc, &e, splits);
    if (rc) { printf("  gemm rc=%d %s\n", rc, lav_last_error()); return 1; }
    hipError_t he = hipDeviceSynchronize();
    if (he != hipSuccess) { printf("  sync error %s\n", hipGetErrorString(he)); return 1; }
    std::vector<float> C((long)M * ldc);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double maxe = 0; long bad = 0;
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
        double d = fabs(C[(long)m * ldc + n] - ref[(long)m * N + n]);
        if (d > maxe) maxe = d;
        if (d > 1e-2 * sqrt((double)K)) ++bad;
    }
    printf("layout %d M=%d N=%d K=%d splits=%d: max|err| %.3e bad=%ld  %s\n", layout, M, N, K, splits, maxe, bad, bad ? "FAIL" : "ok");
    hipFree(dA); hipFree(dB); hipFree(dC);
    return bad != 0;
}

static void bench(int layout, int M, int N, int K, int splits, const char* name, int force_mode = -1) {
    long lda = layout == 2 ? M : K, ldb = layout == 0 ? K : N, arows = layout == 2 ? K : M, brows = layout == 0 ? N : K;
    long ldc = (N + 7) / 8 * 8;
    void *dA, *dB, *dC;
    hipMalloc(&dA, arows * lda * 2); hipMalloc(&dB, brows * ldb * 2); hipMalloc(&dC, (long)M * ldc * 4);
    std::vector<uint16_t> h(arows * lda); for (auto& v : h) v = f2bf(frand());
    hipMemcpy(dA, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    h.resize(brows * ldb); for (auto& v : h) v = f2bf(frand());
    hipMemcpy(dB, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dC, 0, (long)M * ldc * 4);
    lav_gemm_epilogue e; memset(&e, 0, sizeof(e)); e.alpha = 1.f; e.out_mode = splits > 1 ? 2 : 0;
    if (force_mode >= 0) e.out_mode = force_mode;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) lav_gemm_bf16(nullptr, layout, M, N, K, dA, lda, dB, ldb, dC, ldc, &e, splits);
    hipEventRecord(a, nullptr);
    const int it = 10;
    for (int i = 0; i < it; ++i) lav_gemm_bf16(nullptr, layout, M, N, K, dA, lda, dB, ldb, dC, ldc, &e, splits);
    hipEventRecord(b, nullptr); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
    printf("%-34s layout %d M=%-7d N=%-6d K=%-7d splits=%-3d %8.3f ms  %7.1f TFLOP/s\n", name, layout, M, N, K, splits, ms, 2.0 * M * N * K / ms / 1e9);
    hipFree(dA); hipFree(dB); hipFree(dC);
}

int main(int argc, char** argv) {
    int fails = 0;
    int shapes[][3] = {{128, 128, 64}, {256, 128, 128}, {200, 136, 96}, {77, 40, 160}, {384, 30522 / 64, 768}, {130, 260, 32}};
    for (auto& s : shapes) for (int l = 0; l < 3; ++l) fails += check(l, s[0], s[1], s[2], 1);
    fails += check(2, 96, 128, 1000, 4);
    fails += check(2, 384, 128, 4096, 7);
    printf("correctness failures: %d\n", fails);
    if (argc > 1) {
        bench(0, 501760, 384, 128, 1, "swin s1 qkv fwd");
        bench(0, 501760, 512, 128, 1, "swin s1 fc1 fwd");
        bench(0, 31360, 1536, 512, 1, "swin s3 qkv fwd");
        bench(0, 31360, 2048, 512, 1, "swin s3 fc1 fwd");
        bench(0, 31360, 512, 2048, 1, "swin s3 fc2 fwd");
        bench(0, 36096, 2304, 768, 1, "bert qkv fwd (vtm)");
        bench(0, 36096, 3072, 768, 1, "bert ffn1 fwd (vtm)");
        bench(0, 36096, 768, 3072, 1, "bert ffn2 fwd (vtm)");
        bench(0, 4096, 30522, 768, 1, "decoder fwd (vtm)");
        bench(0, 8192, 8192, 8192, 1, "square 8k");
        bench(1, 31360, 512, 2048, 1, "swin s3 fc1 dX");
        bench(1, 36096, 768, 3072, 1, "bert ffn1 dX");
        bench(2, 2048, 512, 31360, 8, "swin s3 fc1 dW");
        bench(2, 3072, 768, 36096, 8, "bert ffn1 dW");
        bench(2, 384, 128, 501760, 64, "swin s1 qkv dW");
        bench(2, 3072, 768, 36096, 1, "bert ffn1 dW store f32", 1);
        bench(2, 3072, 768, 36096, 1, "bert ffn1 dW atomic", 2);
        bench(2, 3072, 768, 36096, 3, "bert ffn1 dW atomic s3", 2);
        bench(0, 3072, 768, 36096, 1, "same shape NT store f32", 1);
        bench(0, 3072, 768, 36096, 1, "same shape NT bf16", 0);
        bench(1, 3072, 768, 36096, 1, "same shape NN store f32", 1);
        bench(2, 384, 128, 501760, 167, "swin s1 qkv dW s167", 2);
        bench(2, 384, 128, 501760, 32, "swin s1 qkv dW s32", 2);
        bench(2, 512, 128, 501760, 128, "swin s1 fc1 dW s128", 2);
        bench(2, 2048, 512, 31360, 4, "swin s3 fc1 dW s4", 2);
        bench(2, 2048, 512, 31360, 1, "swin s3 fc1 dW s1 store", 1);
    }
    return fails != 0;
}
