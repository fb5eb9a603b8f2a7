// Runs one GEMM shape a few times (for rocprofv3 --pmc runs).  usage: gemm_one layout M N K [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/lavender_hip.h"
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }
int main(int argc, char** argv) {
    int layout = atoi(argv[1]), M = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]), it = argc > 5 ? atoi(argv[5]) : 5;
    long lda = layout == 2 ? M : K, ldb = layout == 0 ? K : N, arows = layout == 2 ? K : M, brows = layout == 0 ? N : K;
    void *dA, *dB, *dC;
    hipMalloc(&dA, arows * lda * 2); hipMalloc(&dB, brows * ldb * 2); hipMalloc(&dC, (long)M * N * 4);
    std::vector<uint16_t> h(arows * lda); for (auto& v : h) v = f2bf((float)rand() / RAND_MAX * 2 - 1);
    hipMemcpy(dA, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    h.resize(brows * ldb); for (auto& v : h) v = f2bf((float)rand() / RAND_MAX * 2 - 1);
    hipMemcpy(dB, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    lav_gemm_epilogue e; memset(&e, 0, sizeof(e)); e.alpha = 1.f;
    for (int i = 0; i < it; ++i) lav_gemm_bf16(nullptr, layout, M, N, K, dA, lda, dB, ldb, dC, N, &e, 1);
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
